cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "smin_prefilter or yfcc_shape" 2>&1 | tail -4
timeout 600 python tests/fuzz_parity.py 40 31 2>&1 | tail -3
export YFCC_ARGS="--w 64 --parity 0"; tools/dbg/ab_yfcc.sh aby7 base
export YFCC_ARGS="--w 64 --parity 0 --opt smin_bf16=0"; tools/dbg/ab_yfcc.sh aby8 base
