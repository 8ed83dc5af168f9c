cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -5
timeout 600 python bench.py --steps 2 --warmup 1 --single-reduction --no-cpu-baseline --no-parity-step 2>&1 | grep '^{' | tail -1 | cut -c1-300
