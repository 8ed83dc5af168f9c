cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
PFEM_CG_GRAPH=2 timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py -m gpu -q 2>&1 | tail -6
PFEM_CG_GRAPH=0 PFEM_CG_CHUNK=1 timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "solve or single_reduction or indefinite or edge" 2>&1 | tail -4
