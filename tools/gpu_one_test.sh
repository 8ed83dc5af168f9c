cd "$GRAFT_REPO_ROOT"; timeout 900 python -m pytest tests/test_bench_contract.py -m gpu -x -q -k "alone_on_one_device" 2>&1 | tail -8
