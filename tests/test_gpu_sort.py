"""the .pairs sorter on the GPU against the system's GNU sort, run exactly as the driver runs it
(microcket:480,514): LANG=C sort -k2,2d -k4,4d -k3,3n -k5,5n.  Byte-identical output is the bar."""
import os
import subprocess

import pytest

import microcket_amd as m
import util

pytestmark = pytest.mark.gpu


def _gnu_sort(data: bytes, tmp_path) -> bytes:
    p = tmp_path / "in.pairs"
    p.write_bytes(data)
    e = dict(os.environ, LANG="C", LC_ALL="C")
    return subprocess.run(["sort", "-k2,2d", "-k4,4d", "-k3,3n", "-k5,5n", str(p)], stdout=subprocess.PIPE, env=e, check=True).stdout


def _pairs(profile, seed, groups, mode, **kw):
    text = util.synth(profile, seed, groups, **kw)
    return util.oracle_run(text, mode, 4, 0.5, 10, False)[0]


def test_sorted_pairs_equal_gnu_sort(tmp_path):
    if m.device_count() < 1:
        pytest.fail("no HIP device")
    sets = [
        _pairs("stress", 901, 20000, "unc"),                       # chr10 / chr2 / chrUn_... names: dictionary order drops the '_'
        _pairs("unc", 902, 60000, "unc", genome="mm10", read_len=100, lanes=4),
        _pairs("flash", 903, 30000, "flash"),
    ]
    # heavy ties: the same contact under many read names (whole-line comparison decides), runs longer than the in-thread limit
    tie = b"".join(b"r%05d\tchr2\t1000\tchr10\t5000\t+\t-\n" % ((7919 * k) % 3000) for k in range(3000))
    tie += b"".join(b"q%03d\tchr_2\t1000\tchr10\t5000\t-\t-\n" % k for k in range(40))      # "chr_2" and "chr2" are the same key under -d
    tie += b"x\tchr2\t999\tchr10\t5000\t+\t+\nx\tchr2\t1000\tchr1\t70\t+\t+\ny\tchr2\t1000\tchr10\t4999\t+\t+\n"
    sets.append(tie)
    sets.append(sets[0] + sets[2] + tie)                           # pooling the modes = sort -m of the two sorted files (microcket:514)
    sets.append(b"only\tchr1\t5\tchr1\t9\t+\t-")               # one line without its newline
    for data in sets:
        with m.PairsSorter(0) as s:
            for k in range(0, len(data), 1 << 20):                 # fed in pieces
                s.add(data[k:k + (1 << 20)])
            got = s.sort()
        want = _gnu_sort(data if data.endswith(b"\n") else data + b"\n", tmp_path)
        assert got == want, (len(data), got[:300], want[:300])
    with m.PairsSorter(0) as s:
        assert s.sort() == b""


def test_huge_runs_of_equal_keys_are_slow_but_never_an_error(tmp_path):
    """A pile-up locus: 70,000 lines with the same (chr1, chr2, pos1, pos2) -- more than one workgroup ranks -- plus a second run of
    5,000 and ordinary lines around them.  GNU sort has no limit here (its last-resort comparison is the whole line); neither has
    the sorter: runs beyond 2,048 lines are ranked by the whole GPU (k_tie_huge)."""
    if m.device_count() < 1:
        pytest.fail("no HIP device")
    big = b"".join(b"A00123:45:HXXXXXXXX:%d:%d:%d:%d\tchr3\t777777\tchr3\t888888\t%s\t%s\n" % (1 + k % 4, 1101 + (k * 7919) % 1500, (k * 104729) % 30000, (k * 1299709) % 20000,
                                                                                             b"+-"[k % 2:k % 2 + 1], b"-+"[(k // 2) % 2:(k // 2) % 2 + 1]) for k in range(70000))
    mid = b"".join(b"m%06d\tchr3\t777777\tchr3\t888889\t+\t-\n" % ((k * 31337) % 5000) for k in range(5000))
    rest = _pairs("unc", 905, 20000, "unc")
    rl, bl = rest.splitlines(keepends=True), big.splitlines(keepends=True)
    data = b"".join(rl[:len(rl) // 2] + bl[:len(bl) // 2]) + mid + b"".join(rl[len(rl) // 2:] + bl[len(bl) // 2:])
    with m.PairsSorter(0) as s:
        for k in range(0, len(data), 1 << 20):
            s.add(data[k:k + (1 << 20)])
        got = s.sort()
    assert got == _gnu_sort(data, tmp_path)


def test_sorter_rejects_what_is_not_pairs_text():
    if m.device_count() < 1:
        pytest.fail("no HIP device")
    with m.PairsSorter(0) as s:
        s.add(b"no tabs here\n")
        with pytest.raises(m.MktError):
            s.sort()


def test_executable_sorted_mode_and_pairsort(tmp_path):
    """MKT_SORTED=1: the drop-in executable prints its pairs already in the driver's order (so that the `sort` behind it has
    nothing left to do), optionally behind the 4DN header; bin/pairsort pools two modes' files like `sort -m` (microcket:514)."""
    if m.device_count() < 1:
        pytest.fail("no HIP device")
    exe = m.exe_path()
    pairsort = os.path.join(os.path.dirname(exe), "pairsort")
    hdr = tmp_path / "hdr"
    hdr.write_bytes(b"## pairs format v1.0\n#columns: readID chr1 position1 chr2 position2 strand1 strand2\n")
    outs = {}
    for prof, mode in (("stress", "unc"), ("flash", "flash")):
        text = util.synth(prof, 911, 15000)
        inp = tmp_path / f"{mode}.sam"
        inp.write_bytes(text)
        po, so, lo, st = util.oracle_run(text, mode, 4, 0.5, 10, False)
        e = dict(os.environ, MKT_SORTED="1", MKT_BLOCK_MB="1")
        p = subprocess.run([exe, str(inp), mode, str(tmp_path / "o"), "4", "0.5", "10", "no"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=e)
        assert p.returncode == 0, p.stderr
        want = _gnu_sort(po, tmp_path)
        assert p.stdout == want
        assert (tmp_path / f"o.{mode}2pairs.log").read_bytes() == lo
        c = subprocess.run(["sort", "-c", "-k2,2d", "-k4,4d", "-k3,3n", "-k5,5n"], input=p.stdout, env=dict(os.environ, LANG="C", LC_ALL="C"))
        assert c.returncode == 0                                # what the driver's own sort would find: already in order
        (tmp_path / f"{mode}.pairs").write_bytes(p.stdout)
        outs[mode] = po
        if mode == "unc":
            p2 = subprocess.run([exe, str(inp), mode, str(tmp_path / "h"), "4", "0.5", "10", "no"], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                                env=dict(e, MKT_HEADER=str(hdr)))
            assert p2.returncode == 0 and p2.stdout == hdr.read_bytes() + want
    p = subprocess.run([pairsort, "-H", str(hdr), str(tmp_path / "flash.pairs"), str(tmp_path / "unc.pairs")], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 0, p.stderr
    assert p.stdout == hdr.read_bytes() + _gnu_sort(outs["flash"] + outs["unc"], tmp_path)
