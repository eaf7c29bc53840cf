mkdir -p gpurun_out/r5u
for i in 1 2 3; do for l in old new; do AUDIOCODECS_AMD_LIB=$PWD/tools/experiments/lib_$l.so python tools/experiments/r5u_kernel_ab.py $l mimi rb_fused6 rb128 2>&1 | grep -E "^(old|new) "; done; done > gpurun_out/r5u/ab.txt
cat gpurun_out/r5u/ab.txt
