# (compares hand-made builds: lib_old = shipped (48 frames per wave), lib_new = -DRVQ_MS_EXP=2 (32), lib_ms1 = -DRVQ_MS_EXP=1 (16))
mkdir -p gpurun_out/r5ai
for i in 1 2 3; do for l in old new ms1; do AUDIOCODECS_AMD_LIB=$PWD/tools/experiments/lib_$l.so python tools/experiments/r5u_kernel_ab.py $l encodec rvq 2>&1 | grep -E "^(old|new|ms1) "; done; done > gpurun_out/r5ai/ab.txt
cat gpurun_out/r5ai/ab.txt
