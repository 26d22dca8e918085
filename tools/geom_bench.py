"""Diagnostic: k_fast launch time per tile geometry on the same resident data set, per read length.
    python tools/geom_bench.py [--pairs N] [--read-lens 150,100,60] [--tiles fast,auto] [--sam]
Prints ms per launch, GB/s of text, M lines/s-equivalent pairs/s, deferred tiles; checks that every geometry gives the same
statistics and .log (the outputs themselves are pinned by tests/test_gpu_parity.py)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import microcket_amd as m  # noqa: E402
from microcket_amd import capi  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--pairs", type=int, default=8_000_000)
ap.add_argument("--read-lens", default="150,100,60")
ap.add_argument("--tiles", default="fast,auto")
ap.add_argument("--sam", action="store_true")
ap.add_argument("--mode", default="unc")
ap.add_argument("--passes", type=int, default=5)
args = ap.parse_args()
TILES = {"fast": capi.TILES_FAST, "auto": capi.TILES_AUTO}
for rl in [int(x) for x in args.read_lens.split(",")]:
    ref = None
    for tn in args.tiles.split(","):
        ctx = m.Context(args.mode, 0.5, 10, args.sam, 8, device=0, tiles=TILES[tn])
        ds = ctx.dataset(20260105, 0 if args.mode == "unc" else 1, args.pairs, 1 << 21, read_len=rl)
        for _ in range(args.passes):
            ctx.reset_timing()
            for (p, n, g) in ds.blocks:
                ctx.submit_device(p, n)
            ctx.sync()
        t = ctx.timing()
        st = ctx.finish(True)
        log = ctx.format_log(st)
        key = (st.pairs, st.pair_bytes, st.groups, log)
        ok = "" if ref is None or ref == key else "   STATS DIFFER FROM THE FIRST GEOMETRY"
        if ref is None:
            ref = key
        ms = t.tile_kernel_ms / t.tile_launches
        print(f"read_len {rl:3d} sam={int(args.sam)} tiles {tn:5s}: k_fast {ms:.4f} ms/launch  {ds.total_bytes / t.tile_kernel_ms / 1e6:7.0f} GB/s of text  "
              f"{ds.total_groups / t.tile_kernel_ms / 1e3:7.1f} M pairs/s  tiles {t.tiles} deferred {t.deferred_tiles}  pairs {st.pairs}{ok}", flush=True)
        ds.close()
        ctx.close()
