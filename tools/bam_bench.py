"""sam2bam on a synthetic .sam of N read groups: wall time of the executable, phase marks of the library (MKT_VERBOSE), sizes.
    python tools/bam_bench.py [groups=2000000] [level=1]"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import util  # noqa: E402
import test_gpu_bam as T  # noqa: E402
import microcket_amd.build as b  # noqa: E402

groups = int(sys.argv[1]) if len(sys.argv) > 1 else 2000000
level = sys.argv[2] if len(sys.argv) > 2 else "1"
d = os.environ.get("TMPDIR", "/tmp")
t0 = time.time()
body = util.synth("unc", 21, groups)
hdr = "".join(f"@SQ\tSN:{c}\tLN:250000000\n" for c in reversed(T.CHROMS)).encode()
sam = os.path.join(d, "bench.sam")
with open(sam, "wb") as f:
    f.write(hdr); f.write(body)
print(f"synthetic .sam: {len(body) / 1e9:.2f} GB, {body.count(10)} lines ({time.time() - t0:.1f} s to make)", flush=True)
del body
out = os.path.join(d, "bench.bam")
for rep in range(2):
    t0 = time.time()
    r = subprocess.run([b.SAM2BAM, "-l", level, "-o", out, sam], env=dict(os.environ, MKT_VERBOSE="1"), capture_output=True)
    dt = time.time() - t0
    if r.returncode:
        print(r.stderr.decode()); sys.exit(1)
    print(f"run {rep}: rc {r.returncode}  wall {dt:.2f} s  bam {os.path.getsize(out) / 1e9:.3f} GB  bai {os.path.getsize(out + '.bai') / 1e6:.2f} MB", flush=True)
    print(r.stderr.decode(), flush=True)
