# (measures tap_gemm9, which is NOT in the tree: apply profiles/r5ab_tap9_experiment.diff first -- profiles/r5_tapgemm8_notes.md section 11)
mkdir -p gpurun_out/r5ab
for n in encodec mimi; do
  for l in new abl9_NOSTORE abl9_NOMFMA; do AC_TAP9=1 AUDIOCODECS_AMD_LIB=$PWD/tools/experiments/lib_$l.so timeout 600 python tools/experiments/r5l_lib_ab.py $l $n 2>&1 | grep -E "^(new|abl9)|rror|fault" | head -3; done
done > gpurun_out/r5ab/ab.txt 2>&1
cat gpurun_out/r5ab/ab.txt
