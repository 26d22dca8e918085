"""Diagnostic: kernel time of k_fast when every tile stops after phase k (stamp build, MKT_NO_STAMPS=1).
1 load+bitmaps, 2 line table, 9 heads, 3 parse, 10 group starts (+ a 32-byte record per line when MKT_LADDER_SAM=1: the front half of a
two-kernel split), 4 group, 6 sums, 7 claim, 0 everything.  Outputs are wrong for k != 0.  Stops 1 .. 3 also lose the scan pipeline
(the next window is scanned at the top of its tile), stop 10 keeps it."""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
code = r'''
import os, sys
sys.path.insert(0, %r)
import microcket_amd as m
sam = bool(os.environ.get("MKT_LADDER_SAM"))
ctx = m.Context("unc", 0.5, 10, sam, 8, device=0, tiles=m.TILES_AUTO, extensions=m.EXT_KEYS if os.environ.get("MKT_LADDER_EXT") else 0)
ds = ctx.dataset(20260105, 0, 8000000, 1 << 21)
for _ in range(2):
    ctx.reset_timing()
    for (p, n, g) in ds.blocks: ctx.submit_device(p, n)
    try: ctx.sync()
    except Exception as e: pass
t = ctx.timing()
print("stop=%%s sam=%%d k_fast_ms_per_block %%.4f  GB/s %%.1f" %% (os.environ.get("MKT_DEBUG_STOP","0"), sam, t.tile_kernel_ms / t.tile_launches, ds.total_bytes / t.tile_kernel_ms / 1e6))
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = sys.argv[1] if len(sys.argv) > 1 else "libmkt_hip_stamps.so"
stops = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else (1, 2, 9, 3, 10, 4, 6, 7, 0)
print("lib", lib, flush=True)
for k in stops:
    env = dict(os.environ, MKT_DEBUG_STOP=str(k), MKT_NO_STAMPS="1", MKT_LIB=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "microcket_amd", lib))
    print(subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT).stdout.decode().strip().splitlines()[-1], flush=True)
