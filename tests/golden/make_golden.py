"""Generates tests/golden/golden.json by running THE REFERENCE ITSELF on the fixture inputs.

The reference is compiled unmodified from /root/reference/src/sam2pairs/ into oracle/_ref/ by
oracle/Makefile (nothing from it is copied into this repository).  For every (input, mode, threads,
ratio, mapQ, sam) case the vector holds what the reference produced:
  * the 8-line <prefix>.<mode>2pairs.log, verbatim;
  * stdout (.pairs lines) and <prefix>.<mode>.sam in canonical form (LANG=C sort, the form the
    driver itself produces at microcket:480): full text for the hand-written edge inputs,
    SHA-256 + line count for the seeded synthetic inputs (regenerated from the seed by tests/util.synth).
Run here (needs /root/reference for the build):   python tests/golden/make_golden.py
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import util  # noqa: E402

GRID = [  # (threads, ratio, mapq, sam)
    (4, 0.5, 10, True), (2, 0.5, 10, True), (8, 0.5, 10, False), (4, 0.8, 10, True), (4, 0.5, 30, True), (3, 0.5, 0, True),
]
SYNTH = [  # (name, profile, seed, groups, read_len, genome, lanes)
    ("unc150", "unc", 20260104, 5000, 150, "hg38", 1),
    ("flash", "flash", 20260105, 5000, 150, "hg38", 1),
    ("stress", "stress", 20260106, 12000, 150, "hg38", 1),
    ("unc100_mm10_4lanes", "unc", 20260108, 5000, 100, "mm10", 4),
    # BASELINE configs C4 / C5 at sizes that leave the 48 KiB geometry (several blocks, line tables of the 32 / 16 KiB lean
    # kernels): the reference's own outputs pin those kernels.  One case each (big inputs: seconds of reference time).
    ("unc100_hg38_300k", "unc", 20260109, 300000, 100, "hg38", 1),
    ("unc100_mm10_4lanes_300k", "unc", 20260110, 300000, 100, "mm10", 4),
    ("unc60_hg38_300k", "unc", 20260111, 300000, 60, "hg38", 1),
    ("stress100_hg38_200k", "stress", 20260112, 200000, 100, "hg38", 1),
]
BIG_GRID = [(8, 0.5, 10, True)]


def case(text, mode, T, ratio, mapq, sam, full):
    rc, pairs, s, log, err = util.ref_run(text, mode, T, ratio, mapq, sam)
    assert rc == 0, err
    cp, cs = util.canon(pairs), util.canon(s)
    d = {"mode": mode, "threads": T, "ratio": ratio, "mapq": mapq, "sam": sam, "log": log.decode(),
         "pairs_lines": cp.count(b"\n"), "pairs_sha256": util.sha(cp), "sam_lines": cs.count(b"\n"), "sam_sha256": util.sha(cs)}
    if full:
        d["pairs_sorted"] = cp.decode()
    return d


def main():
    assert util.have_ref(), "oracle/_ref/sam2pairs.ref missing: run make -C oracle (needs /root/reference)"
    out = {"generator": "tests/golden/make_golden.py", "reference_build": "g++ -std=c++11 -O3 -fopenmp (makefile:3,13-15)", "inputs": []}
    for name in ("edge_unc.sam", "edge_flash.sam", "edge_aba.sam"):
        text = open(os.path.join(HERE, name), "rb").read()
        ent = {"name": name, "kind": "file", "sha256": util.sha(text), "cases": []}
        for mode in ("unc", "flash"):
            for (T, ratio, mapq, sam) in GRID:
                ent["cases"].append(case(text, mode, T, ratio, mapq, sam, True))
        out["inputs"].append(ent)
    for (name, prof, seed, groups, rl, genome, lanes) in SYNTH:
        text = util.synth(prof, seed, groups, rl, genome, lanes, 1)
        ent = {"name": name, "kind": "synth", "profile": prof, "seed": seed, "groups": groups, "read_len": rl, "genome": genome, "lanes": lanes,
               "sha256": util.sha(text), "bytes": len(text), "cases": []}
        modes = ("flash",) if prof == "flash" else ("unc",) if prof == "unc" else ("unc", "flash")
        for mode in modes:
            for (T, ratio, mapq, sam) in (BIG_GRID if groups > 100000 else GRID):
                ent["cases"].append(case(text, mode, T, ratio, mapq, sam, False))
        out["inputs"].append(ent)
    with open(os.path.join(HERE, "golden.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("cases:", sum(len(e["cases"]) for e in out["inputs"]))


if __name__ == "__main__":
    main()
