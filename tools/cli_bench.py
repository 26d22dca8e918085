"""End-to-end (PCIe- and file-I/O-inclusive) timing of the drop-in executable against the reference binary
on the same SAM file in /dev/shm, wall clock including process start (SURVEY.md 8(d) "Metric"; never bench.py's `value`).
    python tools/cli_bench.py [pairs] [--sweep]"""
import os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import microcket_amd as m

args = [a for a in sys.argv[1:] if not a.startswith("--")]
pairs = int(args[0]) if args else 4_000_000
sweep = "--sweep" in sys.argv
d = tempfile.mkdtemp(prefix="mkt_cli_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
path = os.path.join(d, "in.sam")
with m.Context("unc", device=0) as c:
    ds = c.dataset(20260105, 0, pairs, 1 << 19, tail_group=True)
    with open(path, "wb") as f:
        for (p, n, g) in ds.blocks:
            f.write(c.copy_to_host(p, n))
    ds.close()
size = os.path.getsize(path)
print(f"file {size/1e9:.2f} GB, {pairs} pairs, host cpus {os.cpu_count()}", flush=True)
ref = os.path.join(ROOT, "oracle", "_ref", "sam2pairs.ref")


def run(name, exe, sam, env=None, reps=1, out="null"):
    if not os.path.exists(exe):
        return
    best = None
    for _ in range(reps):
        e = dict(os.environ)
        if env:
            e.update(env)
        t0 = time.time()
        with open(os.devnull if out == "null" else os.path.join(d, "out.pairs"), "wb") as o:
            p = subprocess.run([exe, path, "unc", os.path.join(d, "out"), "8", "0.5", "10", sam], stdout=o, stderr=subprocess.PIPE, env=e)
        dt = time.time() - t0
        if best is None or dt < best[0]:
            best = (dt, p.returncode, p.stderr.decode())
    dt, rc, err = best
    print(f"{name:44s} rc={rc} {dt:7.3f} s  {pairs/dt/1e6:7.2f} M pairs/s  {size/dt/1e9:6.2f} GB/s", flush=True)
    if env and env.get("MKT_VERBOSE"):
        print("".join("    " + l + "\n" for l in err.splitlines() if l.startswith("[mkt]")), end="", flush=True)
    return dt


run("mi355x sam=no (first run)", m.exe_path(), "no", {"MKT_VERBOSE": "1"})
a = run("mi355x sam=no", m.exe_path(), "no", {"MKT_VERBOSE": "1"}, reps=2)
run("mi355x sam=no, stdout to a /dev/shm file", m.exe_path(), "no", reps=2, out="file")
b = run("mi355x sam=yes", m.exe_path(), "yes", {"MKT_VERBOSE": "1"}, reps=2)
if sweep:
    for mb in (16, 32, 128, 256):
        run(f"mi355x sam=no MKT_BLOCK_MB={mb}", m.exe_path(), "no", {"MKT_BLOCK_MB": str(mb)}, reps=2)
    for th in (2, 4, 16, 32):
        run(f"mi355x sam=no MKT_IO_THREADS={th}", m.exe_path(), "no", {"MKT_IO_THREADS": str(th)}, reps=2)
        run(f"mi355x sam=yes MKT_IO_THREADS={th}", m.exe_path(), "yes", {"MKT_IO_THREADS": str(th)}, reps=2)
r1 = run("reference sam=no thread=8", ref, "no")
r2 = run("reference sam=yes thread=8", ref, "yes")
if a and r1:
    print(f"speedup sam=no {r1 / a:.1f}x   sam=yes {r2 / b:.1f}x", flush=True)
for fn in os.listdir(d):
    os.unlink(os.path.join(d, fn))
os.rmdir(d)
