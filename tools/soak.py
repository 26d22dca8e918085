"""Soak check on the GPU box (test tooling; uses the oracle as the checker): bigger inputs than the test-suite's, every
profile x mode x sam, through the streaming path with the geometry chosen per input; the outputs' line multisets and the
.log must equal the oracle's.   python tools/soak.py [groups]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import microcket_amd as m
import util
from test_gpu_parity import _line_multiset_checksum

groups = int(sys.argv[1]) if len(sys.argv) > 1 else 1_500_000
bad = 0
for prof, pid, mode in (("unc", 0, "unc"), ("flash", 1, "flash"), ("stress", 2, "unc"), ("stress", 2, "flash")):
    for read_len in (150, 60):
        with m.Context(mode, 0.5, 10, True, 8, device=0) as c:
            ds = c.dataset(4242 + pid, pid, groups, 1 << 18, read_len=read_len, tail_group=True)
            host = b"".join(c.copy_to_host(p, nb) for (p, nb, g) in ds.blocks)
            ds.close()
        t0 = time.time()
        po, so, lo, ost = util.oracle_run(host, mode, 8, 0.5, 10, True)
        t1 = time.time()
        for sam in (True, False):
            for block in (0, 48 << 20):
                with m.Context(mode, 0.5, 10, sam, 8, device=0, block_bytes=block) as c:
                    p, s, st, log = c.run_bytes(host, chunk=64 << 20)
                    tm = c.timing()
                ok = log == lo and len(p) == len(po) and _line_multiset_checksum(p) == _line_multiset_checksum(po)
                if sam:
                    ok = ok and len(s) == len(so) and _line_multiset_checksum(s) == _line_multiset_checksum(so)
                print(f"{prof:6s} mode={mode:5s} read_len={read_len:3d} sam={int(sam)} block={block >> 20:3d}M  {len(host) / 1e6:7.1f} MB  pairs {st.pairs:8d}  "
                      f"tiles {tm.tiles} deferred {tm.deferred_tiles}  {'OK' if ok else 'MISMATCH'}", flush=True)
                bad += 0 if ok else 1
        print(f"   (oracle {t1 - t0:.1f} s)", flush=True)
print("soak: bad =", bad)
sys.exit(1 if bad else 0)
