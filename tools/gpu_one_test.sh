cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --cells 80 --strong --steps 2 --warmup 1 --backend gloo --same-device --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print({k:d[k] for k in ('value','n_gpus','scaling','iterations','ms_per_step')}, d['config']['workload'][:80], d['config']['free_dofs'])
"
timeout 300 python bench.py --cells 80 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print({k:d[k] for k in ('value','n_gpus','scaling','iterations','ms_per_step')}, d['config']['free_dofs'])
"
