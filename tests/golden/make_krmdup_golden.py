"""Generates tests/golden/krmdup_golden.json by running THE REFERENCE's krmdup / krmdup.pipe (oracle/_ref, built unmodified from
/root/reference/src/preprocess/ by oracle/Makefile) on seeded synthetic FASTQ (tests/util.synth_fastq).  Per case: the log
text, SHA-256 of <prefix>.read1.fq / .read2.fq (krmdup: byte order is defined) and of the sorted 8-line records of
krmdup.pipe's stdout (its four bucket threads write concurrently).   python tests/golden/make_krmdup_golden.py"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import util  # noqa: E402

CASES = [  # (name, seed, pairs, read_len, args)
    ("small", 31, 3000, 100, ()),
    ("two_batches", 32, 70000, 60, ()),            # > 2^16 pairs: the A,C,G,T output order restarts per batch
    ("keys_12_20", 33, 20000, 100, ("-k", "3", "-K", "7", "-s", "12", "-S", "20")),
    ("dup_heavy", 34, 40000, 50, ()),
]


def main():
    assert os.path.exists(util.KRMDUP_REF), "oracle/_ref/krmdup.ref missing: make -C oracle"
    out = {"generator": "tests/golden/make_krmdup_golden.py", "cases": []}
    for name, seed, pairs, rl, args in CASES:
        text = util.synth_fastq(seed, pairs, rl, dup_rate=0.6 if name == "dup_heavy" else 0.3)
        rc, r1, r2, log, err = util.krmdup_run_cli(util.KRMDUP_REF, text, False, args)
        assert rc == 0, err
        rcp, so, _, logp, errp = util.krmdup_run_cli(util.KRMDUP_PIPE_REF, text, True, args)
        assert rcp == 0 and logp == log, errp
        out["cases"].append({"name": name, "seed": seed, "pairs": pairs, "read_len": rl, "args": list(args), "input_sha256": util.sha(text),
                             "log": log.decode(), "read1_sha256": util.sha(r1), "read2_sha256": util.sha(r2),
                             "pipe_records_sha256": util.sha(b"\n".join(util.fastq_records(so)))})
    with open(os.path.join(HERE, "krmdup_golden.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("cases:", len(out["cases"]))


if __name__ == "__main__":
    main()
