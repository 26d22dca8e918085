"""Diagnostic: fixed costs of the drop-in executable (process start, HIP runtime, context, first pinned window, exit) on a tiny input."""
import os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import microcket_amd as m
import util
d = tempfile.mkdtemp(prefix="mkt_fix_", dir="/dev/shm")
p = os.path.join(d, "in.sam")
open(p, "wb").write(util.synth("unc", 1, 2000))
for k in range(4):
    t0 = time.time()
    r = subprocess.run([m.exe_path(), p, "unc", os.path.join(d, "o"), "8", "0.5", "10", "no"], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=dict(os.environ, MKT_VERBOSE="1"))
    dt = time.time() - t0
    print(f"run {k}: total {dt * 1e3:.1f} ms rc={r.returncode}")
    print("".join("    " + l + "\n" for l in r.stderr.decode().splitlines() if l.startswith("[mkt]")), end="")
t0 = time.time(); subprocess.run(["/bin/true"]); print(f"/bin/true: {(time.time() - t0) * 1e3:.1f} ms")
t0 = time.time(); subprocess.run([os.path.join(ROOT, "oracle", "_ref", "sam2pairs.ref"), p, "unc", os.path.join(d, "r"), "8", "0.5", "10", "no"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL); print(f"reference on the same tiny file: {(time.time() - t0) * 1e3:.1f} ms")
for fn in os.listdir(d): os.unlink(os.path.join(d, fn))
os.rmdir(d)
