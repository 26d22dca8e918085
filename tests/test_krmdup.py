"""SURVEY.md 8(f) N2 -- the reference's own duplicate removal (krmdup / krmdup.pipe, FASTQ keys).  CPU part: the restatement
oracle/krmdup_oracle.c is pinned against golden vectors made by the reference itself and against oracle/_ref directly."""
import json
import os

import pytest

import util


def _golden():
    with open(os.path.join(util.GOLDEN, "krmdup_golden.json")) as f:
        return json.load(f)


def _case_input(c):
    text = util.synth_fastq(c["seed"], c["pairs"], c["read_len"], dup_rate=0.6 if c["name"] == "dup_heavy" else 0.3)
    assert util.sha(text) == c["input_sha256"], c["name"]
    return text


def _kw(args):
    m = {"-k": "hskip1", "-K": "hskip2", "-s": "keylen1", "-S": "keylen2"}
    return {m[args[i]]: int(args[i + 1]) for i in range(0, len(args), 2)}


def test_krmdup_oracle_matches_golden():
    for c in _golden()["cases"]:
        r1, r2, log = util.krmdup_oracle(_case_input(c), **_kw(c["args"]))
        assert log.decode() == c["log"], c["name"]
        assert util.sha(r1) == c["read1_sha256"] and util.sha(r2) == c["read2_sha256"], c["name"]
        # the interleaved form (krmdup.pipe): same records, any bucket order inside a batch
        l1, l2 = r1.split(b"\n")[:-1], r2.split(b"\n")[:-1]
        inter = b"".join(b"\n".join(l1[i:i + 4] + l2[i:i + 4]) + b"\n" for i in range(0, len(l1), 4))
        assert util.sha(b"\n".join(util.fastq_records(inter))) == c["pipe_records_sha256"], c["name"]


@pytest.mark.skipif(not os.path.exists(util.KRMDUP_REF), reason="oracle/_ref/krmdup.ref not built (needs /root/reference)")
def test_krmdup_oracle_matches_reference_build():
    for seed, pairs, rl, args in ((41, 5000, 80, ()), (42, 66000, 40, ("-k", "0", "-s", "8", "-S", "8")), (43, 100, 30, ())):
        text = util.synth_fastq(seed, pairs, rl)
        rc, r1, r2, log, err = util.krmdup_run_cli(util.KRMDUP_REF, text, False, args)
        assert rc == 0, err
        o1, o2, ol = util.krmdup_oracle(text, **_kw(args))
        assert (o1, o2, ol) == (r1, r2, log), (seed, args)
