# (compares tools/experiments/lib_old.so and lib_new.so, two builds made by hand)
mkdir -p gpurun_out/r5ah
for i in 1 2 3; do for n in encodec mimi wavtokenizer; do for l in old new; do AUDIOCODECS_AMD_LIB=$PWD/tools/experiments/lib_$l.so python tools/experiments/r5u_kernel_ab.py $l $n rb_fused6 rb128 2>&1 | grep -E "^(old|new) "; done; done; done > gpurun_out/r5ah/ab.txt
cat gpurun_out/r5ah/ab.txt
