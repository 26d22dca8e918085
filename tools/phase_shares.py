"""Diagnostic: per-phase shader-clock shares of the fused tile kernel (stamp build, never shipped).

    python -c "from microcket_amd import build; build.build_stamps_lib()"
    MKT_LIB=microcket_amd/libmkt_hip_stamps.so python tools/phase_shares.py [pairs] [yes|no] [read_len]
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import microcket_amd as m

# STAMP(k) adds the time since STAMP(k-1)
LABEL = {1: "scan (loads + bitmaps)", 2: "line table", 9: "heads", 3: "parse", 10: "group starts", 4: "groups (classify)", 5: "-", 6: "sums + claim", 7: "-", 11: "account + emit", 8: "tail (.sam copy, end barrier)"}
ORDER = [1, 2, 9, 3, 10, 4, 6, 11, 8]

pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
sam = len(sys.argv) > 2 and sys.argv[2] == "yes"
ctx = m.Context("unc", 0.5, 10, sam, 8, device=0, tiles=m.TILES_AUTO if os.environ.get("MKT_TILES") == "auto" else m.TILES_FAST, extensions=m.EXT_KEYS if os.environ.get("MKT_PROBE_EXT") else 0)
read_len = int(sys.argv[3]) if len(sys.argv) > 3 else 150
ds = ctx.dataset(20260105, 0, pairs, 1 << 21, read_len=read_len)
ctx.L.mkt_debug_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
out = (C.c_ulonglong * 16)()
for _ in range(2):
    ctx.reset_timing()
    for (p, n, g) in ds.blocks:
        ctx.submit_device(p, n)
    ctx.sync()
    ctx.L.mkt_debug_stamps(ctx.h, out)
tot = sum(out[k] for k in ORDER)
t = ctx.timing()
print(f"pairs {ds.total_groups} bytes {ds.total_bytes} kernel_ms(1 pass) {t.tile_kernel_ms:.2f}  -> {ds.total_bytes / t.tile_kernel_ms / 1e6:.1f} GB/s")
for k in ORDER:
    print(f"{LABEL[k]:24s} {out[k]:16d} {100.0 * out[k] / tot:6.2f} %")

names = "LCAP LONG TAB PREV_WS PREV_HEAD NO_PREV OPEN_GROUP LAST_LINE GCAP PAIR_BYTES SHORT_LINE".split()
why = {names[4 * q + k]: (out[12 + q] >> (16 * k)) & 0xFFFF for q in range(3) for k in range(4) if 4 * q + k < len(names)}
print("deferred tiles by reason (2 passes):", {k: v for k, v in why.items() if v}, " of", t.tiles, "tiles per pass; deferred per pass", t.deferred_tiles, [hex(out[k]) for k in (12, 13, 14)])
