cd $GRAFT_REPO_ROOT
python -m pytest tests/test_workspace_contract_gpu.py tests/test_fused_chains_gpu.py tests/test_mimi_gpu_parity.py tests/test_wavtok_gpu_parity.py tests/test_dac_gpu_parity.py -x -q 2>&1 | tail -15
