cd $GRAFT_REPO_ROOT
# the entry point with N > 1 on a one-GPU box (both ranks on cuda:0, gloo for the barrier): spawn path and torchrun path
DET6D_BENCH_BACKEND=gloo timeout 600 python3 bench.py --gpus 2 --steps 20 --warmup 5 2>/dev/null | grep '^{' | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print('spawn', d['n_gpus'], d['value'], d['ranks_seen'], d['per_rank_scenes_per_s'], d['selfcheck'])"
DET6D_BENCH_BACKEND=gloo timeout 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 20 --warmup 5 2>/dev/null | grep '^{' | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print('torchrun', d['n_gpus'], d['value'], d['ranks_seen'], d['per_rank_scenes_per_s'], d['selfcheck'])"
