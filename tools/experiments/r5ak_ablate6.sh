# tap_gemm6 ablation builds (-DT6_ABL_*, wrong results by construction; lib_old = shipped): per-layer tap-GEMM times with one ingredient of a stage removed
mkdir -p gpurun_out/r5ak
for n in wavtokenizer encodec; do for l in old abl6_NOMFMA abl6_NOBLOAD abl6_NOALOAD abl6_NOSTORE old; do AUDIOCODECS_AMD_LIB=$PWD/tools/experiments/lib_$l.so timeout 900 python tools/experiments/r5l_lib_ab.py $l $n 2>&1 | grep -E "^(old|abl6)" ; done; done > gpurun_out/r5ak/ablate6.txt
cat gpurun_out/r5ak/ablate6.txt
