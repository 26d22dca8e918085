"""CLI edge cases on the GPU box (test tooling): the drop-in executable against the reference binary on tiny / odd inputs,
file and pipe input.   python tools/cli_edge.py"""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import microcket_amd as m
import util
ref = os.path.join(ROOT, "oracle", "_ref", "sam2pairs.ref")
exe = m.exe_path()
base = util.synth("unc", 5, 300, tail=1)
cases = {"empty": b"", "newline": b"\n", "header_only": b"@HD\tVN:1.6\n@SQ\tSN:chr1\tLN:1000\n", "one_group": util.synth("unc", 9, 1, tail=0),
         "no_final_newline": base[:-1], "normal": base, "big": util.synth("unc", 6, 60000, tail=1)}
bad = 0
d = tempfile.mkdtemp(prefix="mkt_edge_")
for name, text in cases.items():
    path = os.path.join(d, name + ".sam")
    open(path, "wb").write(text)
    outs = {}
    for tag, binary in (("ref", ref), ("gpu", exe)):
        for how in ("file", "pipe"):
            prefix = os.path.join(d, f"{name}.{tag}.{how}")
            if how == "file":
                r = subprocess.run([binary, path, "unc", prefix, "4", "0.5", "10", "yes"], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            else:
                r = subprocess.run([binary, "/dev/stdin", "unc", prefix, "4", "0.5", "10", "yes"], input=text, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            log = open(prefix + ".unc2pairs.log", "rb").read() if os.path.exists(prefix + ".unc2pairs.log") else None
            sam = open(prefix + ".unc.sam", "rb").read() if os.path.exists(prefix + ".unc.sam") else None
            outs[(tag, how)] = (r.returncode, util.canon(r.stdout), log, util.canon(sam) if sam is not None else None)
    want = outs[("ref", "file")]
    for k, v in outs.items():
        ok = v == want
        if not ok:
            bad += 1
            print("MISMATCH", name, k, v[0], want[0], v[2], want[2], len(v[1]), len(want[1]))
    print(name, "rc", want[0], "pairs bytes", len(want[1]), "ok" if all(v == want for v in outs.values()) else "BAD", flush=True)
print("cli edge: bad =", bad)
sys.exit(1 if bad else 0)
