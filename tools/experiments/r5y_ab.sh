mkdir -p gpurun_out/r5y
AUDIOCODECS_AMD_LIB=$PWD/tools/experiments/lib_new.so python -m pytest tests/test_tap_gemm8_gpu.py tests/test_mimi_gpu_parity.py tests/test_wavtok_gpu_parity.py -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -3 > gpurun_out/r5y/pytest.txt
cat gpurun_out/r5y/pytest.txt
for i in 1 2 3; do for n in mimi wavtokenizer encodec; do for l in old new; do AUDIOCODECS_AMD_LIB=$PWD/tools/experiments/lib_$l.so python tools/experiments/r5l_lib_ab.py $l $n 2>&1 | grep -E "^(old|new) "; done; done; done > gpurun_out/r5y/ab.txt
for l in old new; do AUDIOCODECS_AMD_LIB=$PWD/tools/experiments/lib_$l.so python tools/experiments/r5l_lib_ab.py $l dac 2>&1 | grep -E "^(old|new) "; done >> gpurun_out/r5y/ab.txt
cut -c1-100 gpurun_out/r5y/ab.txt
