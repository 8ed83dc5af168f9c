cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_bench_contract.py tests/test_distributed.py -m gpu -x -q -k "relative_row_groups or config5 or contract or overlap or rccl" 2>&1 | tail -8
