"""Diagnostic: where the sam=yes wall time of the drop-in executable goes (verdict r02 item 5a).  GPU box, repo root:
    python tools/samyes_probe.py [pairs]
Same 150 bp file in /dev/shm; wall clock and the executable's own MKT_VERBOSE marks for: sam=no; sam=yes; sam=yes with the .sam going to
/dev/null (a symlink: everything but the file write); sam=yes with 1 / 2 / 4 / 16 writer threads; fewer reader threads."""
import os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import microcket_amd as m

pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 8_000_000
d = tempfile.mkdtemp(prefix="mkt_probe_", dir="/dev/shm")
path = os.path.join(d, "in.sam")
with m.Context("unc", device=0) as c:
    ds = c.dataset(20260105, 0, pairs, 1 << 19, tail_group=True)
    with open(path, "wb") as f:
        for (p, n, g) in ds.blocks:
            f.write(c.copy_to_host(p, n))
    ds.close()
size = os.path.getsize(path)
print(f"file {size / 1e9:.2f} GB, {pairs} pairs, host cpus {os.cpu_count()}, affinity {len(os.sched_getaffinity(0))}", flush=True)
try:
    print("cpu.max:", open("/sys/fs/cgroup/cpu.max").read().strip(), flush=True)
except OSError:
    pass


def run(name, sam, env=None, null_sam=False, reps=2):
    best = None
    for _ in range(reps):
        for fn in ("out.unc.sam",):
            try:
                os.unlink(os.path.join(d, fn))
            except OSError:
                pass
        if null_sam:
            os.symlink("/dev/null", os.path.join(d, "out.unc.sam"))
        e = dict(os.environ, MKT_VERBOSE="1")
        if env:
            e.update(env)
        t0 = time.time()
        with open(os.devnull, "wb") as o:
            p = subprocess.run([m.exe_path(), path, "unc", os.path.join(d, "out"), "8", "0.5", "10", sam], stdout=o, stderr=subprocess.PIPE, env=e)
        dt = time.time() - t0
        if best is None or dt < best[0]:
            best = (dt, p.returncode, p.stderr.decode())
    dt, rc, err = best
    marks = " | ".join(l[6:].strip() for l in err.splitlines() if l.startswith("[mkt]"))
    print(f"{name:40s} rc={rc} {dt:6.3f} s {size / dt / 1e9:6.2f} GB/s   {marks}", flush=True)


run("sam=no", "no")
run("sam=yes", "yes")
run("sam=yes, .sam -> /dev/null", "yes", null_sam=True)
for w in (1, 2, 8):
    run(f"sam=yes W_THREADS={w}", "yes", {"MKT_W_THREADS": str(w)})
for r, w in ((8, 8), (4, 4), (12, 4)):
    run(f"sam=yes IO_THREADS={r} W_THREADS={w}", "yes", {"MKT_IO_THREADS": str(r), "MKT_W_THREADS": str(w)})
run("sam=yes MKT_BLOCK_MB=32", "yes", {"MKT_BLOCK_MB": "32"})
run("sam=yes MKT_BLOCK_MB=128", "yes", {"MKT_BLOCK_MB": "128"})
for fn in os.listdir(d):
    os.unlink(os.path.join(d, fn))
os.rmdir(d)
