#!/bin/bash
# Pre-round-end checks on the GPU box (repo root): smoke, the default bench line, and a 2-rank rehearsal on one GPU
set -e
python __graft_entry__.py --smoke 2>&1 | tail -1
timeout -k 10 500 python bench.py > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err; python -c "
import json; d=json.load(open('gpurun_out/final_bench.json'))
print('value', d['value']/1e6, 'ms/step', d['ms_per_step'], 'frac', d['roofline']['frac'], 'traffic', d['roofline']['traffic'])
print('dedup', d.get('dedup')); print('plain', d.get('sam2pairs_only')); print('cpu', d['cpu_baseline']); print(d['counters'])"
HSA_ENABLE_IPC_MODE_LEGACY=0 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 2 --warmup 1 --pairs 8000000 --backend gloo --same-gpu 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.load(sys.stdin); print('2 ranks', d['n_gpus'], d['value']/1e6, d['counters'])"
timeout -k 10 200 python bench.py --pairs 16000000 --steps 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('1 rank 16M', d['value']/1e6, d['counters'])"
timeout -k 10 200 python bench.py --pairs 16000000 --steps 2 --no-cpu-baseline --dedup no 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('1 rank 16M plain', d['value']/1e6, d['metric'])"
