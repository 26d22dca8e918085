"""Extensions: pairs-level duplicate marking and per-chromosome contact counts.

NOT part of the reference's sam2pairs: parity is UNPINNED by the reference.  The checker here is a direct Python
restatement of the definition over the ORACLE's .pairs output (input order):
  duplicate  = an earlier reported pair has the same (chr1, pos1, chr2, pos2, strand1, strand2);
  chrstat    = count of reported pairs per (chr1, chr2)  ==  `cut -f2,4 | sort | uniq -c` of the oracle's stdout.
Also checked: enabling the extensions changes nothing in stdout / .sam / .log."""
import collections
import os

import pytest

import microcket_amd as m
import util

pytestmark = pytest.mark.gpu


def _expected(pairs_bytes):
    seen = set()
    flags = bytearray()
    stat = collections.Counter()
    for line in pairs_bytes.split(b"\n")[:-1]:
        f = line.split(b"\t")
        key = tuple(f[1:7])
        flags.append(1 if key in seen else 0)
        seen.add(key)
        stat[(f[1], f[3])] += 1
    txt = b"".join(a + b"\t" + b + b"\t" + str(c).encode() + b"\n" for (a, b), c in sorted(stat.items()))
    return bytes(flags), txt


def _dup_heavy(n_groups, lanes=0):
    """name-grouped SAM with many exact duplicate contacts (PCR-duplicate like) and chr10/chr2/chrX names; lanes > 0: Illumina
    style read names spread over that many lanes (QNAME field 4)"""
    rows = []
    for g in range(n_groups):
        k = g % 37 if g % 3 else g            # two thirds of the pairs repeat one of 37 contacts
        c1 = ("chr1", "chr10", "chr2", "chrX")[k % 4]
        c2 = ("chr1", "chr10", "chr2", "chrX")[(k // 4) % 4] if k % 5 == 0 else c1
        p1, p2 = 10000 + 97 * k, 50000 + 131 * k
        f1, f2 = (65, 145) if k % 2 else (81, 129)
        q = f"M01:7:FCX:{g * 7 % lanes + 1}:1101:{g}:{3 * g + 1}" if lanes else f"r{g}"
        rows.append(f"{q}\t{f1}\t{c1}\t{p1}\t60\t50M\t=\t1\t0\t{'ACGT' * 25}\t{'F' * 100}\n")
        rows.append(f"{q}\t{f2}\t{c2}\t{p2}\t60\t50M\t=\t1\t0\t{'ACGT' * 25}\t{'F' * 100}\n")
    rows.append("zz\t65\tchr1\t1\t60\t50M\t=\t1\t0\tA\tF\nzz\t129\tchr1\t5000\t60\t50M\t=\t1\t0\tA\tF\n")
    return "".join(rows).encode()


@pytest.mark.parametrize("ordered,tiles", [(False, m.TILES_FAST), (True, m.TILES_FAST), (False, m.TILES_SMALL)])
def test_dedup_and_chrstat_vs_definition(ordered, tiles):
    if m.device_count() < 1:
        pytest.fail("no HIP device")
    for text, mode in ((_dup_heavy(6000), "unc"), (util.synth("unc", 61, 8000), "unc"), (util.synth("flash", 62, 6000), "flash"),
                       (util.synth("stress", 63, 9000), "unc")):
        po, so, lo, st = util.oracle_run(text, mode, 4, 0.5, 10, True)
        want_flags, want_stat = _expected(po)
        with m.Context(mode, 0.5, 10, True, 4, device=0, block_bytes=1 << 18, tiles=tiles, ordered=ordered, extensions=m.EXT_KEYS) as c:
            p, s, stats, log = c.run_bytes(text, chunk=1 << 20)
            total, dups, flags = c.ext_dedup(True)
            chrstat = c.ext_chrstat(True)
        # the extension leaves the reference outputs alone
        assert log == lo and util.canon(p) == util.canon(po) and util.canon(s) == util.canon(so)
        assert total == len(want_flags) == st.pairs
        assert flags == want_flags
        assert dups == sum(want_flags)
        assert chrstat == want_stat


def test_executable_side_files(tmp_path):
    if m.device_count() < 1:
        pytest.fail("no HIP device")
    text = _dup_heavy(3000)
    rc, out, s, log, err = util.cli_run(m.exe_path(), text, "unc", 4, 0.5, 10, True, env={"MKT_EXT": "1", "MKT_BLOCK_MB": "1"})
    assert rc == 0, err
    po, so, lo, st = util.oracle_run(text, "unc", 4, 0.5, 10, True)
    assert log == lo and util.canon(out) == util.canon(po)
    # side files live next to the log (util.cli_run uses a temp dir; run again in tmp_path to read them)
    import subprocess
    inp = tmp_path / "in.sam"
    inp.write_bytes(text)
    e = dict(os.environ, MKT_EXT="1")
    p = subprocess.run([m.exe_path(), str(inp), "unc", str(tmp_path / "o"), "4", "0.5", "10", "no"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=e)
    assert p.returncode == 0, p.stderr
    want_flags, want_stat = _expected(po)
    assert (tmp_path / "o.unc.chrstat").read_bytes() == want_stat
    assert (tmp_path / "o.unc.dedup.stat").read_text() == f"Total\t{len(want_flags)}\nUniq\t{len(want_flags) - sum(want_flags)}\nDup\t{sum(want_flags)}\n"
    assert (tmp_path / "o.unc.dups").read_text() == "".join(f"{k}\n" for k, f in enumerate(want_flags) if f)
    # <prefix>.<mode>.dedup.pairs: the reported pairs in input order without the duplicates (first in input order stays)
    lines = po.split(b"\n")[:-1]
    assert p.stdout == po                                    # MKT_EXT=1 runs in input order
    assert (tmp_path / "o.unc.dedup.pairs").read_bytes() == b"".join(l + b"\n" for l, f in zip(lines, want_flags) if not f)
    assert not (tmp_path / "o.unc.dedup.pairs.tmp").exists()


@pytest.mark.parametrize("devices", ["0,0", "0,0,0"])
def test_executable_on_several_devices_equals_one(tmp_path, devices):
    """bin/sam2pairs with MKT_DEVICES (here the same GPU twice / three times = that many contexts on it): one contiguous shard of the
    input file per context, cut on surviving-line group boundaries by the C++ host code; group offsets (quirk Q2), the input's end
    on the last shard (quirk Q1), one .log.  With MKT_EXT=1 the duplicate keys are exchanged device to device inside the library
    (mkt_ext_dedup_multi): .dups / .dedup.stat / .dedup.pairs / .chrstat and stdout (input order) equal the single-device run."""
    if m.device_count() < 1:
        pytest.fail("no HIP device")
    import subprocess
    texts = {"stress": (util.synth("stress", 63, 9000), "unc"), "dup": (_dup_heavy(6000, lanes=4) + util.synth("unc", 71, 3000, 100, "mm10", 4), "unc"),
             "flash": (util.synth("flash", 64, 5000), "flash")}
    for name, (text, mode) in texts.items():
        inp = tmp_path / f"{name}.sam"
        inp.write_bytes(text)
        po, so, lo, st = util.oracle_run(text, mode, 8, 0.5, 10, True)
        for ext in (False, True):
            outs = {}
            for tag, env in (("one", {}), ("multi", {"MKT_DEVICES": devices})):
                e = dict(os.environ, MKT_BLOCK_MB="1", **env)
                if ext:
                    e.update(MKT_EXT="1", MKT_EXT_LANES="1")
                pre = tmp_path / f"{name}_{tag}_{int(ext)}"
                p = subprocess.run([m.exe_path(), str(inp), mode, str(pre), "8", "0.5", "10", "yes"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=e)
                assert p.returncode == 0, (name, tag, p.stderr)
                side = {sfx: (tmp_path / f"{name}_{tag}_{int(ext)}.{mode}{sfx}").read_bytes() for sfx in
                        ([".sam", "2pairs.log"] + ([".dups", ".dedup.stat", ".dedup.pairs", ".chrstat"] if ext else []))}
                outs[tag] = (p.stdout, side)
            (o1, s1), (o2, s2) = outs["one"], outs["multi"]
            assert s2["2pairs.log"] == s1["2pairs.log"] == lo, (name, ext)
            assert util.canon(o2) == util.canon(o1) == util.canon(po), (name, ext)
            assert util.canon(s2[".sam"]) == util.canon(s1[".sam"]) == util.canon(so), (name, ext)
            if ext:
                assert o2 == o1 == po                       # MKT_EXT=1: input order, over the shards as well
                for sfx in (".dups", ".dedup.stat", ".dedup.pairs", ".chrstat"):
                    assert s2[sfx] == s1[sfx], (name, sfx)
                assert not list(tmp_path.glob("*.tmp*"))


def test_sharded_dedup_equals_single_context():
    """Two contexts = two shards on one GPU; the key-space exchange is emulated with Python lists (bench / production
    pass RCCL all-gathers).  Flags must equal the single-context result; chromosome slots differ between the contexts
    (insertion / probing order), which the name-table exchange has to undo."""
    if m.device_count() < 1:
        pytest.fail("no HIP device")
    from microcket_amd import shard
    text = _dup_heavy(5000) + util.synth("unc", 71, 3000)
    # make sure the final shard holds the input's last group, as in a real run
    with m.Context("unc", 0.5, 10, False, 4, device=0, extensions=m.EXT_KEYS, ordered=True) as c:
        c.run_bytes(text)
        total, dups, want = c.ext_dedup(True)
    cuts = shard.cut_points(text, 2)
    ctxs = [m.Context("unc", 0.5, 10, False, 4, device=0, extensions=m.EXT_KEYS) for _ in range(2)]
    # feed the second shard first into its context so the two name tables fill in different orders
    ctxs[1].submit(text[cuts[1]:cuts[2]], last=True)
    ctxs[0].submit(text[cuts[0]:cuts[1]], last=True)
    counts = [c.group_count() for c in ctxs]
    for r, c in enumerate(ctxs):
        c.finish(drop_last=(r == 1), group_offset=sum(counts[:r]), total_groups=sum(counts))
    names = [c.ext_chr_names() for c in ctxs]
    keys = []
    # first "gather": every rank's remapped keys (dedup_sharded remaps before gathering; emulate the two-step exchange)
    import numpy as np
    got = []
    stash = {}

    def run(rank):
        def ag_obj(x):
            return names
        def ag_keys(k):
            stash[rank] = k
            return [stash.get(0, np.zeros((0, 3), np.uint64)), stash.get(1, np.zeros((0, 3), np.uint64))]
        return shard.dedup_sharded(ctxs[rank], rank, 2, rank == 1, ag_obj, ag_keys)
    run(0)                       # fills stash[0] (its own result is incomplete: rank 1 not gathered yet)
    f1, d1, tot1 = run(1)        # both shards present
    f0, d0, tot0 = run(0)
    for c in ctxs:
        c.close()
    assert bytes(f0) + bytes(f1) == want
    assert d0 + d1 == dups == tot0 == tot1


def test_lane_scoped_duplicates_like_the_drivers_b_switch():
    """MKT_EXT_LANES: the read's lane (QNAME field 4) joins the key, so equal contacts in different lanes are both kept -- the
    driver's -b runs one krmdup per lane (microcket:421-451).  Checked against the definition with and without the lane."""
    if m.device_count() < 1:
        pytest.fail("no HIP device")
    for text in (_dup_heavy(6000, lanes=4), util.synth("unc", 81, 8000, 100, "mm10", 4)):
        po, so, lo, st = util.oracle_run(text, "unc", 4, 0.5, 10, False)
        for ext, lanes in ((m.EXT_KEYS, False), (m.EXT_KEYS | m.EXT_LANES, True)):
            for tiles in (m.TILES_FAST, m.TILES_SMALL):
                with m.Context("unc", 0.5, 10, False, 4, device=0, block_bytes=1 << 18, tiles=tiles, extensions=ext) as c:
                    p, s, stats, log = c.run_bytes(text, chunk=1 << 20)
                    total, dups, flags = c.ext_dedup(True)
                assert log == lo and util.canon(p) == util.canon(po)
                assert flags == util.expected_dups(po, lanes), (lanes, tiles)
    assert util.expected_dups(util.oracle_run(_dup_heavy(6000, lanes=4), "unc", 4, 0.5, 10, False)[0], True) != \
        util.expected_dups(util.oracle_run(_dup_heavy(6000, lanes=4), "unc", 4, 0.5, 10, False)[0], False)


@pytest.mark.parametrize("lanes", [False, True])
def test_hash_partitioned_exchange_on_device(lanes):
    """The xGMI design of the sharded duplicate marking with its REAL device side: world contexts = world shards on one GPU, one
    thread per rank, microcket_amd.shard.dedup_exchange over an in-process stand-in for torch.distributed (the collectives
    move torch CUDA tensors; bench.py passes torch.distributed with the nccl = RCCL backend).  Partition, remap, marking
    of the received records and the way back all run as HIP kernels; the flags must equal the single-context result.
    Runs in a process of its own (tests/gpu_exchange_check.py): PyTorch has to be imported BEFORE libmkt_hip.so is loaded --
    torch/lib/libamdhip64.so carries the soname the library asks for, so the library then shares torch's HIP runtime; the
    other way round the process would hold two runtimes and torch would see no GPU."""
    if m.device_count() < 1:
        pytest.fail("no HIP device")
    import subprocess
    import sys
    p = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "gpu_exchange_check.py"), "1" if lanes else "0"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert p.returncode == 0 and b"exchange ok" in p.stdout, p.stdout.decode()[-3000:]
