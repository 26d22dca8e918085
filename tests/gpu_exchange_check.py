"""Run by tests/test_gpu_ext.py::test_hash_partitioned_exchange_on_device in a process of its own (torch first, see there)."""
import os
import sys
import threading

import torch                       # BEFORE microcket_amd: one HIP runtime in the process (torch's)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import microcket_amd as m          # noqa: E402
import util                        # noqa: E402
from microcket_amd import shard    # noqa: E402
from test_gpu_ext import _dup_heavy  # noqa: E402

lanes = len(sys.argv) > 1 and sys.argv[1] == "1"
assert torch.cuda.is_available()
ext = m.EXT_KEYS | (m.EXT_LANES if lanes else 0)
text = _dup_heavy(5000, lanes=4) + util.synth("unc", 71, 3000, 100, "mm10", 4)
with m.Context("unc", 0.5, 10, False, 4, device=0, extensions=ext, ordered=True) as c:
    c.run_bytes(text)
    total, dups, want = c.ext_dedup(True)
po = util.oracle_run(text, "unc", 4, 0.5, 10, False)[0]
assert want == util.expected_dups(po, lanes)
assert 0 < dups < total
for world in (2, 3, 4):      # 4 ranks, mm10, 100 bp, 4 lanes: the shape of BASELINE.json configs[4] (C5)
    cuts = shard.cut_points(text, world, min_mapq=10)
    ctxs = [m.Context("unc", 0.5, 10, False, 4, device=0, extensions=ext) for _ in range(world)]
    for r in reversed(range(world)):      # the name tables fill in different orders
        ctxs[r].submit(text[cuts[r]:cuts[r + 1]], last=True)
    counts = [c.group_count() for c in ctxs]
    for r, c in enumerate(ctxs):
        c.finish(drop_last=(r == world - 1), group_offset=sum(counts[:r]), total_groups=sum(counts))
    fd = util.FakeDist(world)
    res = [None] * world
    err = []

    def run(r):
        try:
            res[r] = shard.dedup_exchange(ctxs[r], r, world, r == world - 1, fd.rank(r), torch, "cuda:0")
        except Exception as ex:          # a dead rank must not leave the others at a barrier
            err.append(repr(ex))
            fd.bar.abort()

    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for c in ctxs:
        c.close()
    assert not err, err
    assert b"".join(res[r][0] for r in range(world)) == want, world
    assert sum(res[r][1] for r in range(world)) == dups and all(res[r][2] == dups for r in range(world))
print("exchange ok")
