"""End-to-end (PCIe- and file-I/O-inclusive) timing of the drop-in executable against the reference binary
on the same SAM file (never bench.py's `value`; DESIGN.md quotes it as the PCIe-inclusive note).
    python tools/cli_bench.py [pairs]"""
import os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import microcket_amd as m

pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
d = tempfile.mkdtemp(prefix="mkt_cli_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
path = os.path.join(d, "in.sam")
with m.Context("unc", device=0) as c:
    ds = c.dataset(20260105, 0, pairs, 1 << 19, tail_group=True)
    with open(path, "wb") as f:
        for (p, n, g) in ds.blocks:
            f.write(c.copy_to_host(p, n))
    ds.close()
size = os.path.getsize(path)
print(f"file {size/1e9:.2f} GB, {pairs} pairs", flush=True)
ref = os.path.join(ROOT, "oracle", "_ref", "sam2pairs.ref")
for name, exe, sam in (("mi355x sam=no", m.exe_path(), "no"), ("mi355x sam=yes", m.exe_path(), "yes"), ("reference sam=no thread=8", ref, "no"), ("reference sam=yes thread=8", ref, "yes")):
    if not os.path.exists(exe):
        continue
    t0 = time.time()
    with open(os.devnull, "wb") as null:
        rc = subprocess.run([exe, path, "unc", os.path.join(d, "out"), "8", "0.5", "10", sam], stdout=null, stderr=subprocess.PIPE).returncode
    dt = time.time() - t0
    print(f"{name:28s} rc={rc} {dt:7.2f} s  {pairs/dt/1e6:7.2f} M pairs/s  {size/dt/1e9:6.2f} GB/s", flush=True)
for fn in os.listdir(d):
    os.unlink(os.path.join(d, fn))
os.rmdir(d)
