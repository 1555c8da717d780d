cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "smin_prefilter or yfcc_shape" 2>&1 | tail -4
timeout 600 python tests/fuzz_parity.py 30 15 2>&1 | tail -3
export YFCC_ARGS="--w 2 64 --parity 0"; tools/dbg/ab_yfcc.sh aby6 base np32 base
