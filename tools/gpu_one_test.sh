cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_bench_contract.py -m gpu -x -q -k "strong" 2>&1 | tail -8
