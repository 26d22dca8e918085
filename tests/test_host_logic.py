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


def test_cigar_shapes_lean_and_generic():
    """CIGAR shapes the lean parser decodes per operation (counts of up to nine digits, tokens of up to 32 bytes) and those it
    hands to the generic kernel (longer tokens, ten-digit counts), next to odd ones both must treat like the reference:
    unknown operations, `*`, digits behind the last operation, an operation without a count, clips in the middle."""
    cigars = [b"150M", b"60M90S", b"60H90M", b"70M1I79M", b"30M1D90M30S", b"50M1234N100M", b"50M123456789N100M", b"20S30M54321N40M10D40M20S",
              b"10M2I10M2D10M2I10M2D10M2I10M90S", b"10M1I10M1D10M1I10M1D10M1I10M1D10M1I10M1D80S", b"50M1234567890N100M", b"*", b"150M7", b"M", b"50M5S95M",
              b"75X75M", b"0M150M", b"100M50N", b"1000M", b"12345M", b"50M3N3N3N50M", b"5S5H140M", b"150", b"9999S1M"]
    lines = []
    for k, cg in enumerate(cigars):
        for mate, (flag, pos, chrom) in enumerate(((65 + (k % 2) * 16, 1000 + 37 * k, b"chr1"), (129 + ((k // 2) % 2) * 16, 1400 + 41 * k, b"chr1" if k % 3 else b"chr10"))):
            lines.append(b"rd%05d\t%d\t%s\t%d\t60\t%s\t=\t1\t0\tACGT\tFFFF\tNM:i:0" % (k, flag, chrom, pos, cg if mate == k % 2 else b"150M"))
    # the same CIGARs as the split read of a 1 + 2 group
    for k, cg in enumerate(cigars):
        lines.append(b"sp%05d\t65\tchr2\t%d\t60\t150M\t=\t1\t0\tACGT\tFFFF" % (k, 5000 + k))
        lines.append(b"sp%05d\t129\tchr2\t%d\t60\t%s\t=\t1\t0\tACGT\tFFFF" % (k, 5300 + k, cg))
        lines.append(b"sp%05d\t2177\tchr3\t%d\t60\t80H70M\t=\t1\t0\tACGT\tFFFF" % (k, 900 + k))
    lines += [b"zz\t65\tchr1\t1\t60\t1M\t=\t1\t0\tA\tF", b"zz\t129\tchr1\t1\t60\t1M\t=\t1\t0\tA\tF"]
    text = b"\n".join(lines) + b"\n"
    # without the two shapes the lean parser refuses (a token of more than 32 bytes, a ten-digit count): the lean path must take the tile
    keep = [ln for ln in lines if b"10M1I10M1D10M1I10M1D10M1I10M1D10M1I10M1D80S" not in ln and b"1234567890N" not in ln]
    lean_text = b"\n".join(keep) + b"\n"
    for mode in ("unc", "flash"):
        for cfg in (0, 1, 3, 10, 12, 14):
            for (T, ratio, mapq, sam) in ((4, 0.5, 10, True), (2, 0.8, 0, False)):
                _check(text, mode, T, ratio, mapq, sam, cfg, 0)
                _check(lean_text, mode, T, ratio, mapq, sam, cfg, 0)
        es = util.emul_run(lean_text, mode, 4, 0.5, 10, True, 10, 0)[3]
        assert es["tiles"] == 1 and es["lean_tiles"] == 1, es
        es = util.emul_run(text, mode, 4, 0.5, 10, True, 10, 0)[3]
        assert es["lean_tiles"] == 0, es


def test_chromosome_name_order_short_and_long_names():
    """chr1.compare(chr2) is bytewise (flash2pairs.h:110, unc2pairs.h:315).  The lean path compares the first eight bytes as one
    number and only looks further when they agree: names of 1..22 bytes with shared prefixes, both orders, both modes."""
    names = [b"chr1", b"chr10", b"chr2", b"chrX", b"chrUn_KI270742v1", b"chrUn_KI270743v1", b"chrUn_KI2", b"chrUn_KI", b"chrUn_K", b"chr1_KI270706v1_random",
             b"chr1_KI270707v1_random", b"chr1_KI270706v1_randon", b"1", b"10", b"MT", b"chrEBV", b"chr22_KI270731v1_random", b"c"]
    lines = []
    k = 0
    for a in names:
        for b in names:
            q = b"pr%05d" % k
            lines.append(q + b"\t65\t" + a + b"\t%d\t60\t100M\t=\t1\t0\t" % (1000 + k) + b"A" * 100 + b"\t" + b"F" * 100)
            lines.append(q + b"\t145\t" + b + b"\t%d\t60\t100M\t=\t1\t0\t" % (90000 - k) + b"A" * 100 + b"\t" + b"F" * 100)
            k += 1
    lines += [b"zz\t65\tchr1\t1\t60\t1M\t=\t1\t0\tA\tF", b"zz\t129\tchr1\t1\t60\t1M\t=\t1\t0\tA\tF"]
    text = b"\n".join(lines) + b"\n"
    for mode in ("unc", "flash"):
        for cfg in (0, 1, 10, 14, 15):
            _check(text, mode, 4, 0.5, 10, True, cfg, 0)
        es = util.emul_run(text, mode, 4, 0.5, 10, True, 15, 0)[3]       # (250-byte lines: the 32 KiB lean geometry holds them)
        assert es["lean_tiles"] == es["tiles"], es


def test_tile_geometry_follows_the_line_length():
    """mkt_fast.h: lean_dims -- the window holds about 126 lines whatever the line length (the lane-per-line phases run two full waves),
    every part is a multiple of 16 bytes inside the kernel's capacities, the window a multiple of 1 KiB where the caps leave room (the
    scan's full-window path), halos of 7.5 / 8.75 lines; monotone in the line length."""
    import ctypes as C
    util.ensure_built()
    lib = C.CDLL(util.EMUL_SO)
    lib.emul_lean_dims.argtypes = [C.c_double, C.POINTER(C.c_uint32)]
    prev = 0
    for avg in (20, 70, 120, 150, 235, 305, 330, 400, 409, 520, 700, 2000):      # (below 48 bytes per line the chooser clamps)
        o = (C.c_uint32 * 6)()
        lib.emul_lean_dims(float(avg), o)
        tile, hb, hf, mt, mb, mf = list(o)
        assert tile % 16 == 0 and hb % 16 == 0 and hf % 16 == 0 and 16 <= tile <= mt and hb <= mb and hf <= mf, (avg, list(o))
        w = tile + hb + hf
        assert w >= prev, (avg, w, prev)
        prev = w
        if tile < mt and hb < mb and hf < mf and tile > 2048 and hb > 256 and hf > 512:      # nothing capped: ~126 lines, 1 KiB granularity
            assert w % 1024 == 0 and abs(w - 126 * avg) <= 512 + 48, (avg, w, w / avg)      # (nearest KiB)
            assert abs(hb - 7.5 * avg) <= 16 and abs(hf - 8.75 * avg) <= 16, (avg, hb, hf)
    o = (C.c_uint32 * 6)()
    lib.emul_lean_dims(409.0, o)                                         # SURVEY's 150 bp bwa line
    assert 40000 <= o[0] <= 49152
