"""Timing of the extensions on a resident data set (not part of bench.py's metric).
    python tools/ext_bench.py [pairs]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import microcket_amd as m

pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
bg = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 21
for ext in (0, m.EXT_KEYS):
    ctx = m.Context("unc", 0.5, 10, False, 8, device=0, tiles=m.TILES_AUTO, extensions=ext)
    ds = ctx.dataset(20260105, 0, pairs, bg, tail_group=True)
    for rep in range(2):
        ctx.reset(); ctx.reset_timing()
        t0 = time.perf_counter()
        for (p, n, g) in ds.blocks: ctx.submit_device(p, n)
        ctx.sync()
        dt = time.perf_counter() - t0
    st = ctx.finish(True)
    print(f"ext={ext} pass {dt*1e3:.1f} ms  {ds.total_groups/dt/1e6:.1f} M pairs/s  kernel {ctx.timing().tile_kernel_ms:.1f} ms  pairs {st.pairs}", flush=True)
    if ext:
        t0 = time.perf_counter(); total, dups, _ = ctx.ext_dedup(True, want_flags=False); t1 = time.perf_counter()
        cs = ctx.ext_chrstat(True); t2 = time.perf_counter()
        print(f"dedup: {total} keys, {dups} dups in {(t1-t0)*1e3:.1f} ms ({total/(t1-t0)/1e6:.1f} M keys/s); chrstat {len(cs.splitlines())} rows in {(t2-t1)*1e3:.1f} ms", flush=True)
    ds.close(); ctx.close()
