mkdir -p gpurun_out/r5v
AUDIOCODECS_AMD_LIB=$PWD/tools/experiments/lib_new.so python -m pytest tests/test_tap_gemm8_gpu.py tests/test_tap_gemm6_gpu.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error" > gpurun_out/r5v/pytest_tap.txt
cat gpurun_out/r5v/pytest_tap.txt
for i in 1 2 3; do for n in encodec mimi wavtokenizer; do for l in old new; do AUDIOCODECS_AMD_LIB=$PWD/tools/experiments/lib_$l.so python tools/experiments/r5l_lib_ab.py $l $n 2>&1 | grep -E "^(old|new) " ; done; done; done > gpurun_out/r5v/ab.txt
for l in old new; do AUDIOCODECS_AMD_LIB=$PWD/tools/experiments/lib_$l.so python tools/experiments/r5l_lib_ab.py $l dac 2>&1 | grep -E "^(old|new) "; done >> gpurun_out/r5v/ab.txt
cut -c1-90 gpurun_out/r5v/ab.txt
