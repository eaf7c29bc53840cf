# tap_gemm8 ablation builds (wrong results by construction): per-layer tap-GEMM times of Mimi / EnCodec with one ingredient of a stage removed
mkdir -p gpurun_out/r5z
for n in mimi encodec; do for l in old abl_NOMFMA abl_NODMA abl_NOALOAD abl_NOSTORE old; do AUDIOCODECS_AMD_LIB=$PWD/tools/experiments/lib_$l.so python tools/experiments/r5l_lib_ab.py $l $n 2>&1 | grep -E "^(old|abl)" ; done; done > gpurun_out/r5z/ablate.txt
cat gpurun_out/r5z/ablate.txt
