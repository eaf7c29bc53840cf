# (compares tools/experiments/lib_old.so and lib_new.so, two builds made by hand: AC_OUT=... AC_OBJ=... bash audiocodecs_amd/csrc/build.sh [-D...])
mkdir -p gpurun_out/r5ad
for i in 1 2; do for n in encodec mimi wavtokenizer; do for l in old new; do AUDIOCODECS_AMD_LIB=$PWD/tools/experiments/lib_$l.so python tools/experiments/r5l_lib_ab.py $l $n 2>&1 | grep -E "^(old|new) "; done; done; done > gpurun_out/r5ad/ab.txt
cut -c1-110 gpurun_out/r5ad/ab.txt
