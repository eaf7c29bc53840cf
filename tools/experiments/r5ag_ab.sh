# (compares tools/experiments/lib_old.so and lib_new.so, two builds made by hand)
mkdir -p gpurun_out/r5ag
for i in 1 2 3; do for n in mimi wavtokenizer encodec; do for l in old new; do AUDIOCODECS_AMD_LIB=$PWD/tools/experiments/lib_$l.so python tools/experiments/r5l_lib_ab.py $l $n 2>&1 | grep -E "^(old|new) "; done; done; done > gpurun_out/r5ag/ab.txt
for l in old new; do AUDIOCODECS_AMD_LIB=$PWD/tools/experiments/lib_$l.so python tools/experiments/r5l_lib_ab.py $l dac 2>&1 | grep -E "^(old|new) "; done >> gpurun_out/r5ag/ab.txt
cut -c1-100 gpurun_out/r5ag/ab.txt
