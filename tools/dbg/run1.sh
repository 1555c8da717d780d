cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_native_sharded.py -x -q 2>&1 | grep -E "passed|failed" | tail -3
timeout 900 python tests/fuzz_parity.py 80 41 2>&1 | tail -2
