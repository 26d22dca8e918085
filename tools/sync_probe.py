import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import microcket_amd as m
ctx = m.Context("unc", 0.5, 10, False, 8, device=0, tiles=m.TILES_FAST)
ds = ctx.dataset(20260105, 0, 16000000, 1 << 21)
for rep in range(3):
    for passes in (1, 4, 16):
        t0 = time.perf_counter()
        for _ in range(passes):
            for (p, n, g) in ds.blocks: ctx.submit_device(p, n)
        t1 = time.perf_counter()
        import ctypes
        ctx.L.hipDeviceSynchronize if False else None
        ctx.sync()
        t2 = time.perf_counter()
        t3 = time.perf_counter(); ctx.sync(); t4 = time.perf_counter()
        print(f"passes {passes}: enqueue {1e3*(t1-t0):.2f} ms, sync {1e3*(t2-t1):.2f} ms (gpu work ~{passes*8*2.25:.1f} ms), empty sync {1e3*(t4-t3):.3f} ms", flush=True)
