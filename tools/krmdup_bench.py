"""Diagnostic: bin/krmdup.pipe on a 1.3 GB FASTQ in /dev/shm, two segment sizes; output compared with the oracle (GPU box)."""
import os, sys, time, subprocess
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import util, microcket_amd as m
parts = [util.synth_fastq(80 + k, 250000, 100, dup_rate=0.3) for k in range(4)]
text = b"".join(parts * 3)           # 3 M pairs, two thirds of them repeats of the first million
open("/dev/shm/krm_big.fq", "wb").write(text)
print("pairs", text.count(b"\n") // 8, "bytes", len(text), flush=True)
t = time.time(); o1, o2, ol = util.krmdup_oracle(text); print("oracle %.2f s" % (time.time() - t), ol.decode().replace("\n", " "), flush=True)
exe = os.path.join(os.path.dirname(m.exe_path()), "krmdup.pipe")
for seg in ("256", "64"):
    t = time.time()
    p = subprocess.run([exe, "-i", "/dev/shm/krm_big.fq", "-o", "/dev/shm/krm_big_out"], stdout=subprocess.PIPE, env=dict(os.environ, MKT_RMDUP_SEGMENT_MB=seg))
    dt = time.time() - t
    l1, l2 = o1.split(b"\n")[:-1], o2.split(b"\n")[:-1]
    want = b"".join(b"\n".join(l1[i:i + 4] + l2[i:i + 4]) + b"\n" for i in range(0, len(l1), 4))
    print("segment %s MB: rc %d  %.2f s  %.2f GB/s  output equal to the oracle: %s" % (seg, p.returncode, dt, len(text) / dt / 1e9, p.stdout == want), flush=True)
    os.remove("/dev/shm/krm_big_out.log")
os.remove("/dev/shm/krm_big.fq")
