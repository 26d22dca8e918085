"""Diagnostic: per-phase shader-clock shares of the fused tile kernel (stamp build, never shipped).

    python -c "from microcket_amd import build; build.build_stamps_lib()"
    MKT_LIB=microcket_amd/libmkt_hip_stamps.so python tools/phase_shares.py [pairs] [yes|no]
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import microcket_amd as m

# STAMP(k) adds the time since STAMP(k-1)
LABEL = {1: "load + bitmaps", 2: "line table", 3: "parse", 4: "group start", 5: "group walk + classify", 6: "tile sums", 7: "look-back", 8: "emit"}

pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
sam = len(sys.argv) > 2 and sys.argv[2] == "yes"
ctx = m.Context("unc", 0.5, 10, sam, 8, device=0, tiles=m.TILES_FAST)
ds = ctx.dataset(20260105, 0, pairs, 1 << 19)
ctx.L.mkt_debug_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
out = (C.c_ulonglong * 16)()
for _ in range(2):
    ctx.reset_timing()
    for (p, n, g) in ds.blocks:
        ctx.submit_device(p, n)
    ctx.sync()
    ctx.L.mkt_debug_stamps(ctx.h, out)
tot = sum(out[1:9])
t = ctx.timing()
print(f"pairs {ds.total_groups} bytes {ds.total_bytes} kernel_ms(1 pass) {t.tile_kernel_ms:.2f}  -> {ds.total_bytes / t.tile_kernel_ms / 1e6:.1f} GB/s")
for k in range(1, 9):
    print(f"{LABEL[k]:24s} {out[k]:16d} {100.0 * out[k] / tot:6.2f} %")
