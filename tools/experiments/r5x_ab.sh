mkdir -p gpurun_out/r5x
AUDIOCODECS_AMD_LIB=$PWD/tools/experiments/lib_new.so python -m pytest tests/test_gpu_parity.py tests/test_graph_mode_gpu.py tests/test_wavtok_gpu_parity.py tests/test_workspace_contract_gpu.py tests/test_integration_stub_gpu.py -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -3 > gpurun_out/r5x/pytest.txt
cat gpurun_out/r5x/pytest.txt
for i in 1 2 3; do for n in encodec wavtokenizer; do for l in old new; do AUDIOCODECS_AMD_LIB=$PWD/tools/experiments/lib_$l.so python tools/experiments/r5l_lib_ab.py $l $n 2>&1 | grep -E "^(old|new) " | cut -c1-80; done; done; done > gpurun_out/r5x/ab.txt
cat gpurun_out/r5x/ab.txt
