"""CPU checks of the product's host-side logic and of the tile algorithm's phase functions.

tests/host/tile_emul.cpp runs the SAME per-tile phase functions the HIP kernel runs
(microcket_amd/csrc/mkt_tile.h, mkt_core.h), serially, for several tile geometries -- including
tiny tiles and halos that force every out-of-window path -- plus the block cutting and Q1/Q2
bookkeeping of mkt_host.h.  It is a test tool, not a product path: the library has no CPU fallback
(test_abi.py checks that)."""
import os

import pytest

import util

CFGS = {0: "production 48K tile", 5: "production 32K tile", 15: "lean + generic, 32K tile", 1: "small 256 B tile", 2: "mid 1K tile", 3: "2K tile, 16 B halos", 4: "production 16K tile (short lines)",
        10: "lean path + generic for deferred tiles, 48K tile", 12: "lean + generic, 1K tile", 13: "lean + generic, 2K tile / 16 B halos",
        14: "lean + generic, 16K tile"}


def _check(text, mode, T, ratio, mapq, sam, cfg, block):
    po, so, lo, st = util.oracle_run(text, mode, T, ratio, mapq, sam)
    pe, se, le, es = util.emul_run(text, mode, T, ratio, mapq, sam, cfg, block)
    tag = (mode, T, ratio, mapq, sam, CFGS[cfg], block)
    assert es["err"] == 0, tag
    assert pe == po, tag           # the emulation keeps input order: bytes are identical
    assert se == so, tag
    assert le == lo, tag
    assert es["groups"] == st.groups, tag


@pytest.mark.parametrize("cfg", [0, 1, 2, 3, 4, 5, 10, 12, 13, 14, 15])
@pytest.mark.parametrize("name", ["edge_unc.sam", "edge_flash.sam"])
def test_tile_phases_edge_fixtures(name, cfg):
    text = open(os.path.join(util.GOLDEN, name), "rb").read()
    for mode in ("unc", "flash"):
        for (T, ratio, mapq, sam) in ((4, 0.5, 10, True), (2, 0.8, 10, True), (8, 0.5, 0, False)):
            for block in (0, 3000):
                _check(text, mode, T, ratio, mapq, sam, cfg, block)


@pytest.mark.parametrize("cfg", [0, 1, 2, 3, 4, 5, 10, 12, 13, 14, 15])
@pytest.mark.parametrize("profile,seed,groups,modes", [
    ("unc", 11, 1500, ("unc",)), ("flash", 12, 1500, ("flash",)), ("stress", 13, 4000, ("unc", "flash")),
])
def test_tile_phases_synthetic(profile, seed, groups, modes, cfg):
    text = util.synth(profile, seed, groups)
    for mode in modes:
        for (T, ratio, mapq, sam) in ((4, 0.5, 10, True), (3, 0.8, 30, False)):
            for block in (0, 100000, 5000):
                _check(text, mode, T, ratio, mapq, sam, cfg, block)


@pytest.mark.parametrize("cfg", [0, 1, 3, 10, 13])
def test_filtered_stranger_inside_a_group_never_splits_it(cfg):
    """A kept, B filtered, A kept (tests/golden/edge_aba.sam): the reference groups surviving lines only, so the host cut
    must not fall between A and B whatever the block size."""
    text = open(os.path.join(util.GOLDEN, "edge_aba.sam"), "rb").read()
    for (T, ratio, mapq, sam) in ((4, 0.5, 10, True), (8, 0.5, 1, False)):
        for block in (0, 700, 1500, 4096, 20000):
            _check(text, "unc", T, ratio, mapq, sam, cfg, block)
    _check(text, "flash", 4, 0.5, 10, True, cfg, 1500)


def test_host_cut_on_surviving_lines():
    from microcket_amd import shard
    text = open(os.path.join(util.GOLDEN, "edge_aba.sam"), "rb").read()
    po, so, lo, st = util.oracle_run(text, "unc", 4, 0.5, 10, True)
    for parts in (2, 3, 7):
        cuts = shard.cut_points(text, parts, min_mapq=10)
        pairs = b""
        groups = 0
        for r in range(parts):
            piece = text[cuts[r]:cuts[r + 1]]
            # every piece is a whole number of groups: its own oracle run + a sacrificial last group = the same lines
            p, s_, l_, st_ = util.oracle_run(piece + b"zz\t65\tchr1\t1\t60\t1M\nzz\t129\tchr1\t1\t60\t1M\n", "unc", 4, 0.5, 10, True)
            pairs += p
            groups += st_.groups - 1
        assert groups == st.groups, (parts, groups, st.groups)
        assert util.canon(pairs).count(b"\n") >= util.canon(po).count(b"\n")      # the whole input's last group is dropped (Q1), the pieces' are not
        assert set(po.splitlines()) <= set(pairs.splitlines())


def test_ragged_and_empty_inputs():
    for text in (b"", b"\n", b"\n\n\n", b"@HD\tVN:1.6\n", b"no newline at all", b"a\tb\n",
                 b"r1\t65\tchr1\t100\t60\t50M\t=\t1\t0\tAC\tFF\nr1\t129\tchr1\t5000\t60\t50M\t=\t1\t0\tAC\tFF\nr2\t65\tchr1\t1\t60\t5M"):
        for mode in ("unc", "flash"):
            for cfg in (0, 1, 10, 13):
                _check(text, mode, 4, 0.5, 10, True, cfg, 0)


def test_last_line_without_newline_and_crlf():
    base = util.synth("unc", 5, 50, tail=1)
    _check(base[:-1], "unc", 4, 0.5, 10, True, 1, 0)                  # no trailing newline
    _check(base[:-1], "unc", 4, 0.5, 10, True, 10, 0)
    _check(base.replace(b"\n", b"\r\n"), "unc", 4, 0.5, 10, True, 12, 0)
    _check(base.replace(b"\n", b"\r\n"), "unc", 4, 0.5, 10, True, 1, 0)   # CR stays part of the last field / the .sam line


def test_long_fields_take_the_generic_parser():
    q = b"Q" * 300
    text = (q + b"\t65\tchr1\t1000\t60\t150M\t=\t1\t0\tA\tF\n" + q + b"\t129\tchrUn_" + b"x" * 200 + b"\t9000\t60\t150M\t=\t1\t0\tA\tF\n"
            + b"z\t65\tchr1\t1\t60\t1M\t=\t1\t0\tA\tF\nz\t129\tchr1\t1\t60\t1M\t=\t1\t0\tA\tF\n")
    for cfg in (0, 1, 2, 3, 4, 5, 10, 12, 13, 14, 15):
        _check(text, "unc", 4, 0.5, 10, True, cfg, 0)
