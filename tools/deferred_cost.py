"""Diagnostic: what tiles left to the generic kernel cost.  A synthetic 150 bp set of ~1 GB with K lines whose CIGAR has more than 32
bytes (the lean parser refuses them: their tiles are deferred), K from argv; five passes on the resident path.  Run it under
rocprofv3 --kernel-trace --stats and read k_tiles' average duration:  rocprofv3 ... -- python3 tools/deferred_cost.py 8"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import microcket_amd as m

K = int(sys.argv[1]) if len(sys.argv) > 1 else 1
with m.Context("unc", 0.5, 10, False, 8, device=0) as c:
    ds = c.dataset(20260105, 0, 1_000_000, 1 << 20)
    (p, n, g) = ds.blocks[0]
    host = bytearray(c.copy_to_host(p, n))
    ds.close()
lines = host.split(b"\n")
step = max(1, (len(lines) - 2) // (K + 1))
odd = 0
for k in range(K):
    i = (k + 1) * step
    f = lines[i].split(b"\t")
    if len(f) > 9 and f[5] != b"*":
        f[5] = b"10M1I10M1D10M1I10M1D10M1I10M1D10M1I10M1D" + f[5]       # 40 bytes in front: the classification changes, the oracle is not asked
        lines[i] = b"\t".join(f)
        odd += 1
text = b"\n".join(lines)
with m.Context("unc", 0.5, 10, False, 8, device=0) as c:
    d = c.device_text(bytes(text))
    for _ in range(5):
        c.submit_device(d, len(text))
        c.sync()
    t = c.timing()
    print(f"odd lines {odd}: tiles {t.tiles} deferred {t.deferred_tiles} over {t.tile_launches} launches, k_fast {t.tile_kernel_ms / t.tile_launches:.4f} ms per launch", flush=True)
