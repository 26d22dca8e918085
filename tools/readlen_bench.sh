#!/bin/bash
# throughput for other read lengths, forced 32 KB tiles vs geometry chosen per input (GPU box, repo root)
for rl in 50 75 100 150; do for tl in fast auto; do
  timeout -k 10 200 python bench.py --pairs 16000000 --steps 3 --no-cpu-baseline --read-len $rl --tiles $tl 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin); r=d['roofline']
print('read_len $rl tiles $tl: %.1f M pairs/s, k_fast %.0f GB/s, tiles %d, left to generic %d, %.0f B/pair' % (d['value']/1e6, r['achieved'], r['tiles'], r['tiles_left_to_generic_kernel'], d['config']['bytes_per_pair_in']))"
done; done
