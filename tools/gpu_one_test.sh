cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "row_forms_beyond" 2>&1 | tail -12
