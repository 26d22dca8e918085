"""bin/makestat, the native stand-in for the reference's bin/make.stat.pl, against golden vectors printed by that
script itself (tests/golden/makestat_golden.json) and, when /root/reference is present, against the script run on the same logs."""
import json
import os
import subprocess

import pytest

import util

EXE = os.path.join(util.ROOT, "microcket_amd", "bin", "makestat")
SCRIPT = "/root/reference/bin/make.stat.pl"


def _build():
    import sys
    sys.path.insert(0, util.ROOT)
    from microcket_amd import build
    build.build_makestat()


def test_makestat_matches_golden(tmp_path):
    _build()
    g = json.load(open(os.path.join(util.GOLDEN, "makestat_golden.json")))
    for c in g["cases"]:
        for fn, txt in c["files"].items():
            (tmp_path / ("s." + fn)).write_text(txt)
        got = subprocess.run([EXE, "s", c["concat"]], cwd=tmp_path, stdout=subprocess.PIPE, check=True).stdout.decode()
        assert got == c["expected"], (c["seed"], c["concat"])
        if os.path.exists(SCRIPT):
            ref = subprocess.run(["perl", SCRIPT, "s", c["concat"]], cwd=tmp_path, stdout=subprocess.PIPE, check=True).stdout.decode()
            assert got == ref
        for fn in c["files"]:
            (tmp_path / ("s." + fn)).unlink()
    assert subprocess.run([EXE], stderr=subprocess.PIPE).returncode == 2


def test_makestat_chrstat_extension_keeps_the_legacy_lines(tmp_path):
    _build()
    c = json.load(open(os.path.join(util.GOLDEN, "makestat_golden.json")))["cases"][1]
    for fn, txt in c["files"].items():
        (tmp_path / ("s." + fn)).write_text(txt)
    (tmp_path / "s.unc.chrstat").write_text("chr1\tchr1\t1234567\nchr1\tchr10\t89\n")
    got = subprocess.run([EXE, "s", c["concat"], "--chrstat"], cwd=tmp_path, stdout=subprocess.PIPE, check=True).stdout.decode()
    assert got.startswith(c["expected"]) and got.endswith("## Contacts per chromosome pair\nchr1\tchr1\t1,234,567\nchr1\tchr10\t89\n")
