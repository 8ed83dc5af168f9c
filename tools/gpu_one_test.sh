cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
PFEM_CG_SINGLE_REDUCTION=1 timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py tests/test_golden_drivers.py tests/test_fortran_boundary.py -m gpu -q 2>&1 | tail -25
