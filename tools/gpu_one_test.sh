cd "$GRAFT_REPO_ROOT"; timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "variants_agree" 2>&1 | tail -15
