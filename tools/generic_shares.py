"""Diagnostic: where the GENERIC tile kernel (k_tiles, MKT_NO_LEAN=1: every tile) spends a tile -- shader-clock stamps of lane 0 (stamp build).
    MKT_NO_LEAN=1 MKT_LIB=microcket_amd/libmkt_hip_stamps.so python tools/generic_shares.py [pairs]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import microcket_amd as m

LABEL = {1: "window -> LDS, masks", 2: "line table", 3: "parse", 4: "group starts", 5: "-", 6: "groups + sums", 7: "claim", 8: "emit"}
pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000
ctx = m.Context("unc", 0.5, 10, False, 8, device=0, tiles=m.TILES_FAST)
ds = ctx.dataset(20260105, 0, pairs, 1 << 19)
ctx.L.mkt_debug_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
out = (C.c_ulonglong * 16)()
ctx.reset_timing()
for (p, n, g) in ds.blocks:
    ctx.submit_device(p, n)
ctx.sync()
ctx.L.mkt_debug_stamps(ctx.h, out)
t = ctx.timing()
tot = sum(out[k] for k in range(1, 9))
print(f"pairs {ds.total_groups} bytes {ds.total_bytes} tiles {t.tiles} kernel_ms {t.tile_kernel_ms:.2f}")
for k in range(1, 9):
    print(f"{LABEL[k]:24s} {out[k]:16d} {100.0 * out[k] / max(tot, 1):6.2f} %")
