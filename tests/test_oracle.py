"""The oracle (oracle/sam2pairs_oracle.c, a CPU restatement) is pinned:
  * against the committed golden vectors, which hold outputs of the REFERENCE ITSELF
    (tests/golden/golden.json, made by tests/golden/make_golden.py from oracle/_ref);
  * against oracle/_ref directly whenever that build is present (this container; it also travels
    to the GPU box with the snapshot)."""
import os

import pytest

import util


def _input_bytes(ent):
    if ent["kind"] == "file":
        return open(os.path.join(util.GOLDEN, ent["name"]), "rb").read()
    return util.synth(ent["profile"], ent["seed"], ent["groups"], ent["read_len"], ent["genome"], ent["lanes"], 1)


def test_golden_inputs_reproducible(golden):
    """The seeded generator regenerates the exact inputs the golden vectors were made from."""
    for ent in golden["inputs"]:
        assert util.sha(_input_bytes(ent)) == ent["sha256"], ent["name"]


def test_oracle_matches_golden(golden):
    n = 0
    for ent in golden["inputs"]:
        text = _input_bytes(ent)
        for c in ent["cases"]:
            pairs, sam, log, st = util.oracle_run(text, c["mode"], c["threads"], c["ratio"], c["mapq"], c["sam"])
            cp, cs = util.canon(pairs), util.canon(sam)
            tag = (ent["name"], c["mode"], c["threads"], c["ratio"], c["mapq"], c["sam"])
            assert log.decode() == c["log"], tag
            assert cp.count(b"\n") == c["pairs_lines"], tag
            assert util.sha(cp) == c["pairs_sha256"], tag
            assert util.sha(cs) == c["sam_sha256"], tag
            if "pairs_sorted" in c:
                assert cp.decode() == c["pairs_sorted"], tag
            n += 1
    assert n >= 50


@pytest.mark.skipif(not util.have_ref(), reason="oracle/_ref not built (needs /root/reference)")
@pytest.mark.parametrize("profile,seed,groups,modes", [
    ("unc", 101, 4000, ("unc",)), ("flash", 102, 4000, ("flash",)), ("stress", 103, 9000, ("unc", "flash")),
])
def test_oracle_matches_reference_build(profile, seed, groups, modes):
    text = util.synth(profile, seed, groups)
    for mode in modes:
        for (T, ratio, mapq, sam) in ((2, 0.5, 10, True), (4, 0.5, 10, True), (8, 0.5, 10, True), (4, 0.8, 30, False), (5, 0.5, 0, True)):
            rc, rp, rs, rl, err = util.ref_run(text, mode, T, ratio, mapq, sam)
            assert rc == 0
            op, osam, ol, st = util.oracle_run(text, mode, T, ratio, mapq, sam)
            assert util.canon(rp) == util.canon(op), (profile, mode, T, ratio, mapq)
            assert util.canon(rs) == util.canon(osam), (profile, mode, T, ratio, mapq)
            assert rl == ol, (profile, mode, T, ratio, mapq, rl, ol)


@pytest.mark.skipif(not util.have_ref(), reason="oracle/_ref not built (needs /root/reference)")
def test_quirk_q2_across_batches():
    """K just above 2^18 surviving groups with a LARGE remainder (keeps the reference clear of its
    own `loaded` race): the logged selfCircle depends on the thread count exactly as the closed form says."""
    n = (1 << 18) + (1 << 17)
    text = util.synth("unc", 77, n, read_len=50)
    for T in (2, 4, 8):
        rc, rp, rs, rl, err = util.ref_run(text, "unc", T, 0.5, 0, False)
        assert rc == 0
        op, osam, ol, st = util.oracle_run(text, "unc", T, 0.5, 0, False)
        assert st.groups % (1 << 18) >= (1 << 16)
        assert rl == ol, (T, rl, ol)
        assert util.sha(util.canon(rp)) == util.sha(util.canon(op))


def test_oracle_cli_exit_codes(tmp_path):
    import subprocess
    exe = util.ORACLE_EXE
    sam = tmp_path / "x.sam"
    sam.write_bytes(util.synth("unc", 1, 10))
    assert subprocess.run([exe], stderr=subprocess.PIPE).returncode == 2
    assert subprocess.run([exe, str(sam), "unc", str(tmp_path / "o"), "1"], stderr=subprocess.PIPE).returncode == 5
    assert subprocess.run([exe, str(sam), "bad", str(tmp_path / "o"), "4"], stderr=subprocess.PIPE).returncode == 6
    assert subprocess.run([exe, str(tmp_path / "missing.sam"), "unc", str(tmp_path / "o"), "4"], stderr=subprocess.PIPE).returncode == 10
