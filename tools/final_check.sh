set -e
python __graft_entry__.py --smoke 2>&1 | tail -3
HSA_ENABLE_IPC_MODE_LEGACY=0 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 2 --warmup 1 --pairs 8000000 --backend gloo --same-gpu 2>/dev/null | tail -1 | cut -c1-400
timeout -k 10 200 python bench.py --pairs 16000000 --steps 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin); print('1 rank 16M counters', d['counters'])"
