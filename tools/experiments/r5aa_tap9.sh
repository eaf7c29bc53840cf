# (measures tap_gemm9, which is NOT in the tree: apply profiles/r5ab_tap9_experiment.diff first -- profiles/r5_tapgemm8_notes.md section 11)
mkdir -p gpurun_out/r5aa
L=$PWD/tools/experiments/lib_new.so
for n in encodec mimi; do
  AUDIOCODECS_AMD_LIB=$L timeout 600 python tools/experiments/r5l_lib_ab.py tap8 $n 2>&1 | grep -E "^(tap8|tap9)|rror|fault" | head -5
  AC_TAP9=1 AUDIOCODECS_AMD_LIB=$L timeout 600 python tools/experiments/r5l_lib_ab.py tap9_1 $n 2>&1 | grep -E "^(tap8|tap9)|rror|fault" | head -5
  AC_TAP9=2 AUDIOCODECS_AMD_LIB=$L timeout 600 python tools/experiments/r5l_lib_ab.py tap9_2 $n 2>&1 | grep -E "^(tap8|tap9)|rror|fault" | head -5
done > gpurun_out/r5aa/ab.txt 2>&1
cat gpurun_out/r5aa/ab.txt
