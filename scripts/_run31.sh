cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04ae
NO_EXTRAS=1 CGH_TIMING=1 python scripts/session_leg.py 22 6 2>&1 | grep -E "entry:|each" | tail -7 | cut -c1-200
NO_EXTRAS=1 CGH_TIMING=1 python scripts/session_leg.py 16 6 2>&1 | grep -E "entry:|each" | tail -5 | cut -c1-200
timeout 1500 python -m pytest tests/test_rep3_party_abi.py tests/test_synthetic_scale.py tests/test_contexts_and_blocks.py tests/test_chacha_rand.py -m gpu -x -q 2>&1 | tail -2
