mkdir -p gpurun_out/r5t
for i in 1 2 3; do for n in mimi wavtokenizer encodec; do for l in old new; do AUDIOCODECS_AMD_LIB=$PWD/tools/experiments/lib_$l.so python tools/experiments/r5l_lib_ab.py $l $n 2>&1 | grep -E "^(old|new) " ; done; done; done > gpurun_out/r5t/ab.txt
for l in old new; do AUDIOCODECS_AMD_LIB=$PWD/tools/experiments/lib_$l.so python tools/experiments/r5l_lib_ab.py $l dac 2>&1 | grep -E "^(old|new) "; done >> gpurun_out/r5t/ab.txt
AUDIOCODECS_AMD_LIB=$PWD/tools/experiments/lib_trace.so python tools/experiments/r5r_trace_mimi.py > gpurun_out/r5t/trace_mimi.txt 2>&1
cat gpurun_out/r5t/ab.txt
