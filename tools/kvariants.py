"""Diagnostic: k_fast launch time of several library builds (MKT_LIB) on the same resident data set.
    python tools/kvariants.py libA.so libB.so ...        (names relative to microcket_amd/)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r"""
import os, sys
sys.path.insert(0, %r)
import microcket_amd as m
for sam, mode, prof in ((False, "unc", 0), (True, "unc", 0), (False, "flash", 1)):
    ctx = m.Context(mode, 0.5, 10, sam, 8, device=0)
    ds = ctx.dataset(20260105, prof, 8000000, (1 << 21) if prof == 0 else (1 << 20))
    for _ in range(5):
        ctx.reset_timing()
        for (p, n, g) in ds.blocks: ctx.submit_device(p, n)
        ctx.sync()
    t = ctx.timing()
    st = ctx.finish(True)
    print("   %%-5s sam=%%d  k_fast %%.4f ms/launch  %%.0f GB/s of text  deferred %%d  pairs %%d" %% (mode, sam, t.tile_kernel_ms / t.tile_launches, ds.total_bytes / t.tile_kernel_ms / 1e6, t.deferred_tiles, st.pairs), flush=True)
    ds.close(); ctx.close()
""" % ROOT
for lib in sys.argv[1:]:
    print(lib, flush=True)
    env = dict(os.environ, MKT_LIB=os.path.join(ROOT, "microcket_amd", lib))
    r = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    print(r.stdout.decode().rstrip(), flush=True)
