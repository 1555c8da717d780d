cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q 2>&1 | grep -E "passed|failed" | tail -3
timeout 900 python tests/fuzz_parity.py 60 21 2>&1 | tail -3
export ABARGS="--extras 0 --spread-steps 0"; tools/dbg/ab_hard.sh ab3 base nohq base nohq
