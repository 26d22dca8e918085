"""GPU parity tests (run on the MI355X box: python -m pytest tests -m gpu).

Everything goes through the C ABI of include/mkt.h (ctypes) or through the drop-in executable;
the oracle / the reference build are only the CHECKER.  Bit-exact bar: .log byte-identical;
stdout and .sam byte-identical in input order (ordered=True) or after the driver's own canonical
sort (default any-order mode -- the reference's order is thread-schedule dependent too)."""
import os

import numpy as np
import pytest

import microcket_amd as m
import util

pytestmark = pytest.mark.gpu

GRID = [(4, 0.5, 10, True), (2, 0.8, 30, True), (8, 0.5, 0, False), (3, 0.5, 10, True)]


def _need_gpu():
    if m.device_count() < 1:
        pytest.fail("no HIP device: the HIP path is the only path (no CPU fallback to test)")


def _run(text, mode, T, ratio, mapq, sam, tiles=m.TILES_AUTO, ordered=False, block=0, chunk=1 << 20):
    with m.Context(mode, ratio, mapq, sam, T, device=0, block_bytes=block, tiles=tiles, ordered=ordered) as c:
        return c.run_bytes(text, chunk=chunk)


def _check(text, mode, T, ratio, mapq, sam, **kw):
    po, so, lo, st = util.oracle_run(text, mode, T, ratio, mapq, sam)
    p, s, stats, log = _run(text, mode, T, ratio, mapq, sam, **kw)
    tag = (mode, T, ratio, mapq, sam, kw)
    assert log == lo, tag
    assert stats.groups == st.groups and stats.pairs == st.pairs, tag
    if kw.get("ordered"):
        assert p == po and s == so, tag
    else:
        assert util.canon(p) == util.canon(po) and util.canon(s) == util.canon(so), tag


def test_golden_vectors_through_the_c_abi(golden):
    """Every golden vector (outputs of the reference itself) reproduced by the HIP path.  The big entries (BASELINE configs
    C4 / C5 shapes: 100 bp hg38, mm10 with 4 lanes, 60 bp; >= 2^18 groups) run in 24 MiB blocks so that MKT_TILES_AUTO
    leaves the 48 KiB geometry: the reference's own outputs pin the 32 KiB and 16 KiB lean kernels byte for byte."""
    _need_gpu()
    for ent in golden["inputs"]:
        if ent["kind"] == "file":
            text = open(os.path.join(util.GOLDEN, ent["name"]), "rb").read()
        else:
            text = util.synth(ent["profile"], ent["seed"], ent["groups"], ent["read_len"], ent["genome"], ent["lanes"], 1)
        assert util.sha(text) == ent["sha256"]
        big = ent["kind"] == "synth" and ent["groups"] > 100000
        for c in ent["cases"]:
            with m.Context(c["mode"], c["ratio"], c["mapq"], c["sam"], c["threads"], device=0, block_bytes=(24 << 20) if big else 0) as ctx:
                p, s, stats, log = ctx.run_bytes(text, chunk=(16 << 20) if big else (1 << 20))
                tm = ctx.timing()
            cp, cs = util.canon(p), util.canon(s)
            tag = (ent["name"], c["mode"], c["threads"], c["ratio"], c["mapq"], c["sam"])
            assert log.decode() == c["log"], tag
            assert util.sha(cp) == c["pairs_sha256"] and cp.count(b"\n") == c["pairs_lines"], tag
            assert util.sha(cs) == c["sam_sha256"], tag
            if big:     # the lean kernel did the work, the first block included (tile bytes from the line length, not from a failed block)
                assert tm.tiles > 0 and tm.deferred_tiles * 100 < tm.tiles, (tag, tm.tiles, tm.deferred_tiles)


@pytest.mark.parametrize("read_len", [100, 60])
@pytest.mark.parametrize("profile,pid,modes", [("unc", 0, ("unc",)), ("stress", 2, ("unc", "flash")), ("flash", 1, ("flash",))])
def test_lean_geometries_byte_parity(profile, pid, modes, read_len):
    """100 bp and 60 bp reads (tile bytes chosen from the line length before the first launch): >= 2^18 groups through the
    streaming path in 48 MiB blocks under MKT_TILES_AUTO; .log byte-identical, .pairs and .sam equal to the oracle's as line
    multisets (same length + order-independent 64-bit checksum of the lines), few tiles left to the generic kernel."""
    _need_gpu()
    groups = (1 << 18) + (1 << 16) + 4321
    with m.Context("unc", device=0) as c:
        ds = c.dataset(777 + pid, pid, groups, 1 << 17, read_len=read_len, genome=(pid == 2), lanes=1, tail_group=True)
        host = b"".join(c.copy_to_host(p, nb) for (p, nb, g) in ds.blocks)
        ds.close()
    for mode in modes:
        for sam in (True, False):
            po, so, lo, ost = util.oracle_run(host, mode, 8, 0.5, 10, sam)
            with m.Context(mode, 0.5, 10, sam, 8, device=0, block_bytes=48 << 20) as c:
                p, s, st, log = c.run_bytes(host, chunk=24 << 20)
                tm = c.timing()
            tag = (profile, mode, read_len, sam, tm.tiles, tm.deferred_tiles)
            assert log == lo, tag
            assert st.groups == ost.groups and st.pairs == ost.pairs, tag
            assert len(p) == len(po) and _line_multiset_checksum(p) == _line_multiset_checksum(po), tag
            assert len(s) == len(so) and _line_multiset_checksum(s) == _line_multiset_checksum(so), tag
            assert tm.deferred_tiles * 100 < tm.tiles, tag          # < 1 % of the tiles left to the generic kernel, block 0 included


@pytest.mark.parametrize("tiles,ordered", [(m.TILES_FAST, True), (m.TILES_FAST, False), (m.TILES_SMALL, True)])
@pytest.mark.parametrize("profile,seed,groups,modes", [
    ("unc", 21, 6000, ("unc",)), ("flash", 22, 6000, ("flash",)), ("stress", 23, 12000, ("unc", "flash")),
])
def test_synthetic_sets_vs_oracle(profile, seed, groups, modes, tiles, ordered):
    _need_gpu()
    text = util.synth(profile, seed, groups)
    for mode in modes:
        for (T, ratio, mapq, sam) in GRID:
            for block in (0, 1 << 17):
                _check(text, mode, T, ratio, mapq, sam, tiles=tiles, ordered=ordered, block=block, chunk=70001)


@pytest.mark.parametrize("name", ["edge_unc.sam", "edge_flash.sam", "edge_aba.sam"])
def test_edge_fixtures_vs_oracle(name):
    _need_gpu()
    text = open(os.path.join(util.GOLDEN, name), "rb").read()
    for mode in ("unc", "flash"):
        for (T, ratio, mapq, sam) in GRID:
            for tiles in (m.TILES_FAST, m.TILES_SMALL):
                _check(text, mode, T, ratio, mapq, sam, tiles=tiles, ordered=True, block=4096, chunk=777)


def test_ragged_inputs():
    _need_gpu()
    base = util.synth("unc", 5, 200, tail=1)
    for text in (b"", b"\n\n", b"@HD\tVN:1.6\n", b"junk line without tabs\n", base[:-1], base.replace(b"\n", b"\r\n"),
                 base + b"@late\theader\n", b"\n" + base):
        _check(text, "unc", 4, 0.5, 10, True, ordered=True)
        _check(text, "flash", 4, 0.5, 10, True, tiles=m.TILES_SMALL, ordered=True)


def test_input_window_equals_submit():
    """mkt_input_window / mkt_submit_window (what the executable uses) against mkt_submit on the same bytes, odd piece sizes."""
    _need_gpu()
    text = util.synth("unc", 11, 4000, tail=1)
    po, so, lo, ost = util.oracle_run(text, "unc", 4, 0.5, 10, True)
    for block, piece in ((0, 0), (1 << 16, 0), (1 << 16, 12345), (5000, 777)):
        with m.Context("unc", 0.5, 10, True, 4, device=0, block_bytes=block, ordered=True) as c:
            p, s, st, log = c.run_bytes_window(text, piece)
        assert log == lo and p == po and s == so, (block, piece)


def test_tiny_lines_overflow_the_line_table_and_fall_back():
    """Lines of ~35 bytes put more line starts into a window than the fast config's table holds: AUTO re-runs the
    block with the small-tile config; forcing the fast config fails loudly instead of returning wrong results."""
    _need_gpu()
    rows = []
    for g in range(4000):
        rows.append(f"r{g}\t65\tchr1\t{1000 + g}\t60\t5M\t=\t1\t0\tA\tF\n")
        rows.append(f"r{g}\t129\tchr1\t{9000 + 3 * g}\t60\t5M\t=\t1\t0\tA\tF\n")
    text = "".join(rows).encode()
    _check(text, "unc", 4, 0.5, 10, True, tiles=m.TILES_AUTO)
    with pytest.raises(m.MktError):
        _run(text, "unc", 4, 0.5, 10, True, tiles=m.TILES_FAST)


def test_executable_is_a_drop_in(tmp_path):
    """Same argv, stdout, side files and stderr messages as the reference binary (when present) / the oracle."""
    _need_gpu()
    exe = m.exe_path()
    for profile, mode in (("unc", "unc"), ("flash", "flash"), ("stress", "unc")):
        text = util.synth(profile, 41, 5000)
        for (T, ratio, mapq, sam) in ((4, 0.5, 10, True), (8, 0.8, 20, False)):
            rc, out, s, log, err = util.cli_run(exe, text, mode, T, ratio, mapq, sam, env={"MKT_BLOCK_MB": "1"})
            assert rc == 0, err
            if util.have_ref():
                rrc, rout, rs, rlog, rerr = util.ref_run(text, mode, T, ratio, mapq, sam)
                assert rerr == err                       # INFO / WARN lines, byte for byte
            else:
                rout, rs, rlog, _ = util.oracle_run(text, mode, T, ratio, mapq, sam)
            assert log == rlog
            assert util.canon(out) == util.canon(rout)
            assert util.canon(s) == util.canon(rs)
    # stdin, as the driver calls it (microcket:479): bwa | sam2pairs /dev/stdin ...
    import subprocess
    text = util.synth("unc", 43, 3000)
    p = subprocess.run([exe, "/dev/stdin", "unc", str(tmp_path / "o"), "4", "0.5", "10", "no"], input=text, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    po, so, lo, st = util.oracle_run(text, "unc", 4, 0.5, 10, False)
    assert p.returncode == 0 and util.canon(p.stdout) == util.canon(po)
    assert (tmp_path / "o.unc2pairs.log").read_bytes() == lo and not (tmp_path / "o.unc.sam").exists()


def test_device_generator_equals_host_generator():
    _need_gpu()
    with m.Context("unc", device=0) as c:
        for prof, pid in (("unc", 0), ("flash", 1), ("stress", 2)):
            d, n = c.synth_device(77, pid, 3000, first_group=123, tail_group=True)
            assert c.copy_to_host(d, n) == util.synth(prof, 77, 3000, tail=1, first=123), prof


def test_resident_path_multi_block_and_q2_across_batches():
    """K > 2^18 surviving groups, device-resident blocks: the logged selfCircle must follow the closed form (quirk Q2)."""
    _need_gpu()
    n = (1 << 18) + (1 << 17) + 12345
    for T in (2, 8):
        with m.Context("unc", 0.5, 0, False, T, device=0) as c:
            ds = c.dataset(99, 0, n, 1 << 16, read_len=50, tail_group=True)
            host = b"".join(c.copy_to_host(p, nb) for (p, nb, g) in ds.blocks)
            for (p, nb, g) in ds.blocks:
                c.submit_device(p, nb)
            st = c.finish(True)
            tm = c.timing()
            ds.close()
        po, so, lo, ost = util.oracle_run(host, "unc", T, 0.5, 0, False)
        assert c.format_log(st) == lo, (T, c.format_log(st), lo)
        assert st.groups == ost.groups and st.pairs == ost.pairs
        # 50 bp reads (~190-byte lines): the tile bytes follow the line length counted BEFORE the first launch, so even the
        # first block stays on the lean kernel (round 2 learned the geometry by failing: block 0 went to the generic kernel)
        assert tm.deferred_tiles * 100 < tm.tiles, (tm.tiles, tm.deferred_tiles)


def test_resident_path_replays_a_block_when_the_lines_get_shorter():
    """Concatenated inputs of very different line length, queued on the resident path without a sync in between: two blocks of
    150 bp reads (48 KiB tiles), then a block of ~35-byte lines that overflows even the generic kernel's line table.  The
    failing block -- not the probe -- is re-run with smaller geometries by mkt_sync; outputs must equal the oracle's."""
    _need_gpu()
    a1, a2 = util.synth("unc", 1234, 4000, tail=0), util.synth("unc", 1235, 4000, tail=0, first=4000)
    rows = []
    for g in range(6000):
        rows.append(f"s{g}\t65\tchr1\t{1000 + g}\t60\t5M\t=\t1\t0\tA\tF\n")
        rows.append(f"s{g}\t129\tchr1\t{9000 + 3 * g}\t60\t5M\t=\t1\t0\tA\tF\n")
    short = "".join(rows).encode()
    host = a1 + a2 + short
    po, so, lo, ost = util.oracle_run(host, "unc", 4, 0.5, 10, True)
    with m.Context("unc", 0.5, 10, True, 4, device=0, ordered=False) as c:
        for part in (a1, a2, short):
            c.submit_device(c.device_text(part), len(part))
        st = c.finish(True)
        pairs, sam = c.fetch_last_block()
    assert c.format_log(st) == lo
    assert st.groups == ost.groups and st.pairs == ost.pairs and st.pair_bytes == len(po)
    # the last block's own outputs: the oracle's for that text plus the line of its final group (dropped by quirk Q1 in a whole run)
    pshort = util.oracle_run(short, "unc", 4, 0.5, 10, True)[0]
    assert util.canon(pairs)[:len(util.canon(pshort))] == util.canon(pshort) or set(pshort.splitlines()) <= set(pairs.splitlines())


def test_long_groups_make_the_host_widen_the_halos():
    """Read names with nine alignment lines each among ordinary pairs: a group that starts in the last lines of a tile does not end
    inside the default forward halo (8.75 lines), the tile goes to the generic kernel (counted apart: BlockResult::pad2), and the host
    widens the halos of the following blocks (mkt_capi.cpp: adapt_geometry / widen_halos).  The geometry changes in mid-stream: the
    outputs must equal the oracle's all the same, and over the whole input fewer tiles are deferred than the first block alone defers."""
    _need_gpu()
    import random
    rnd = random.Random(99)
    seq, qual = "A" * 150, "F" * 150
    rows = []
    for g in range(330000):
        n = 9 if rnd.random() < 0.3 else 2
        for k in range(n):
            flag = 65 if k % 2 == 0 else 129
            rows.append(f"A00123:45:HXXXXXXXX:1:{1101 + g % 50}:{g}:{1000 + g % 977}\t{flag}\tchr{1 + g % 22}\t{100000 + 37 * g + 400 * k}\t60\t150M\t=\t{100000 + 37 * g}\t0\t{seq}\t{qual}\tNM:i:0\n")
    host = "".join(rows).encode()
    assert len(host) > 3 * (64 << 20)
    po, so, lo, ost = util.oracle_run(host, "unc", 4, 0.5, 10, False)

    def run(text):
        with m.Context("unc", 0.5, 10, False, 4, device=0, block_bytes=64 << 20) as c:
            p, s, st, log = c.run_bytes(text, chunk=16 << 20)
            tm = c.timing()
        return p, log, tm.tiles, tm.deferred_tiles

    p, log, tiles, deferred = run(host)
    assert log == lo and len(p) == len(po) and _line_multiset_checksum(p) == _line_multiset_checksum(po)
    first = host[:host.rfind(b"\n", 0, 60 << 20) + 1]            # (about the first block)
    _p1, _l1, tiles1, deferred1 = run(first)
    assert deferred1 * 500 > tiles1, (tiles1, deferred1)          # the default halos are too narrow for this input ...
    assert deferred * tiles1 < 0.7 * deferred1 * tiles, (tiles, deferred, tiles1, deferred1)      # ... and what follows the first block does better


def test_two_contexts_as_two_shards():
    """The sharded bookkeeping (group offsets, Q1 on the last shard only) with real kernels: 2 contexts on one GPU."""
    _need_gpu()
    from microcket_amd import shard
    text = util.synth("stress", 51, 8000)
    cuts = shard.cut_points(text, 2)
    po, so, lo, ost = util.oracle_run(text, "unc", 4, 0.5, 10, True)
    ctxs = [m.Context("unc", 0.5, 10, True, 4, device=0, block_bytes=1 << 18, ordered=True) for _ in range(2)]
    outs = []
    for r, c in enumerate(ctxs):
        c.submit(text[cuts[r]:cuts[r + 1]], last=True)
    counts = [c.group_count() for c in ctxs]
    total = sum(counts)
    sums = [0] * 8
    pairs, sam = b"", b""
    for r, c in enumerate(ctxs):
        a, b = c.drain()
        st = c.finish(drop_last=(r == 1), group_offset=sum(counts[:r]), total_groups=total)
        a2, b2 = c.drain()
        pairs += a + a2
        sam += b + b2
        for k, name in enumerate(shard.COUNTERS):
            sums[k] += getattr(st, name)
        c.close()
    assert pairs == po and sam == so
    assert shard.format_log(dict(zip(shard.COUNTERS, sums))) == lo


def test_executable_clean_exit_switch(tmp_path):
    """MKT_CLEAN_EXIT=1: the executable tears its context down (mkt_destroy) instead of leaving through _exit(0); same bytes, same
    files, exit code 0 -- what a leak checker runs, and the only path on which a destructor-time error could ever surface."""
    _need_gpu()
    text = util.synth("stress", 77, 9000)
    po, so, lo, st = util.oracle_run(text, "unc", 4, 0.5, 10, True)
    for env in ({"MKT_CLEAN_EXIT": "1"}, {"MKT_CLEAN_EXIT": "1", "MKT_BLOCK_MB": "1"}, {}):
        rc, out, sam, log, err = util.cli_run(m.exe_path(), text, "unc", 4, 0.5, 10, True, env=env)
        assert rc == 0, err
        assert util.canon(out) == util.canon(po) and util.canon(sam) == util.canon(so) and log == lo, env


def _line_multiset_checksum(buf: bytes) -> int:
    """Order-independent 64-bit checksum of the lines of buf (vectorised polynomial hash per line, summed)."""
    a = np.frombuffer(buf, dtype=np.uint8)
    if a.size == 0:
        return 0
    nl = np.flatnonzero(a == 10)
    starts = np.concatenate(([0], nl[:-1] + 1)) if nl.size else np.array([0])
    line_id = np.zeros(a.size, dtype=np.int64)
    line_id[starts[1:]] = 1
    line_id = np.cumsum(line_id)
    pos = np.arange(a.size, dtype=np.int64) - starts[line_id]
    P = np.empty(int(pos.max()) + 1, dtype=np.uint64)
    P[0] = 1
    for i in range(1, P.size):
        P[i] = (int(P[i - 1]) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    with np.errstate(over="ignore"):
        w = (a.astype(np.uint64) + np.uint64(1)) * P[pos]
        h = np.add.reduceat(w, starts)
        h ^= h >> np.uint64(29)
        h *= np.uint64(0x9E3779B97F4A7C15)
        return int(h.sum(dtype=np.uint64))


def test_full_size_properties():
    """BASELINE.json configs[1] size (100 M pairs, ~92 GB resident), EVERY block against the CPU oracle.
      * each of the 48 resident blocks (2^21 pairs, 1.9 GB of text; the last one holds the input's tail group, quirk Q1): the
        block's .pairs bytes as they sit in HBM after its pass == the oracle's output for that block's text as line multisets
        (same byte count, same line count, same order-independent 64-bit checksum); the oracle runs per block in a pool of host
        threads (ctypes releases the GIL), nothing on the GPU is re-run for the comparison;
      * the whole run's statistics == the blocks' oracle counters added up, with the logged selfCircle (quirk Q2) evaluated at
        the global group indices: the .log is byte-identical to what the reference would write for the 94 GB input;
      * pairs == trans + cis10K + cis1K + cis0; idempotence (a second pass gives identical statistics); block-cut independence
        (1.9 GB and 0.47 GB blocks of a 16 M-pair data set give the same statistics);
      * a 4 M-pair slice through the STREAMING path (host bytes, 64 MiB blocks) against the oracle as well."""
    _need_gpu()
    import concurrent.futures as cf
    pairs = int(os.environ.get("MKT_TEST_FULL_PAIRS", 100_000_000))
    T = 8
    workers = max(2, min(16, (os.cpu_count() or 4) - 2))
    with m.Context("unc", 0.5, 10, False, T, device=0, tiles=m.TILES_AUTO) as c:
        ds = c.dataset(20260105, 0, pairs, 1 << 21, tail_group=True)      # bench.py's blocks: 1.9 GB of text each
        nb_ = ds.n_blocks
        results = [None] * nb_

        def check_block(b, text, out_pairs, last):
            st, (hp, pl, pn), _sam, scl = util.oracle_shard_summary(text, "unc", T, 0.5, 10, False, drop_last=False)
            gh, gl = util.lines_checksum(out_pairs)
            return b, st, (hp, pl, pn), (gh, gl, int(out_pairs.size)), scl

        c.reset()
        pend = []
        with cf.ThreadPoolExecutor(workers) as ex:
            for b, (p, nb, g) in enumerate(ds.blocks):
                c.submit_device(p, nb)
                c.sync()
                out_pairs, _ = c.fetch_last_block_np()                 # every group of the block, its last one included
                text = c.copy_to_host_np(p, nb)
                pend.append(ex.submit(check_block, b, text, out_pairs, b == nb_ - 1))
                del text, out_pairs
                while len(pend) >= workers + 2:                         # bounded: at most workers + 2 blocks (2 GB each) on the host
                    r = pend.pop(0).result()
                    results[r[0]] = r[1:]
            for f in pend:
                r = f.result()
                results[r[0]] = r[1:]
        a = c.finish(True)
        log_gpu = c.format_log(a)
        # ---- per block: line multisets
        for b, (st, want, got, scl) in enumerate(results):
            assert got == want, (b, got, want)
        # ---- the run: counters added up, quirk Q1 (the input's last group) and Q2 (global indices) applied by the checker
        K = sum(int(r[0].groups) for r in results)
        assert a.groups == K
        tot = dict(lowMap=0, manyHits=0, unpaired=0, selfCircle=0, trans=0, cis10K=0, cis1K=0, cis0=0)
        off = 0
        for b, (st, want, got, scl) in enumerate(results):
            for k in ("lowMap", "manyHits", "unpaired", "trans", "cis10K", "cis1K", "cis0"):
                tot[k] += int(getattr(st, k))
            for gi in scl:
                if off + int(gi) != K - 1 and util.selfcircle_logged(off + int(gi), K, T):
                    tot["selfCircle"] += 1
            off += int(st.groups)
        # the tail group (the generator's dummy last pair, dropped by Q1) was counted by the per-block oracle: take it out again
        tail_st, _p, _s, tail_sc = util.oracle_shard_summary(c.copy_to_host_np(ds.blocks[-1][0], ds.blocks[-1][1]), "unc", T, 0.5, 10, False, drop_last=True)
        last_st = results[-1][0]
        for k in ("lowMap", "manyHits", "unpaired", "trans", "cis10K", "cis1K", "cis0"):
            tot[k] -= int(getattr(last_st, k)) - int(getattr(tail_st, k))
        want_log = "".join(f"{k}\t{tot[k] & 0xFFFFFFFF}\n" for k in ("lowMap", "manyHits", "unpaired", "selfCircle", "trans", "cis10K", "cis1K", "cis0")).encode()
        assert log_gpu == want_log, (log_gpu, want_log)
        assert a.pairs == a.trans + a.cis10K + a.cis1K + a.cis0
        assert a.bytes_in == ds.total_bytes
        # ---- idempotence
        c.reset()
        for (p, nb, g) in ds.blocks:
            c.submit_device(p, nb)
        b2 = c.finish(True)
        assert a.counters() == b2.counters() and a.pairs == b2.pairs and a.pair_bytes == b2.pair_bytes and a.groups == b2.groups
        nblk = min(2, ds.n_blocks)
        host = b"".join(c.copy_to_host(p, nb) for (p, nb, g) in ds.blocks[:nblk])
        ds.close()
        # the same data cut into 1.9 GB and 0.47 GB blocks: identical statistics
        sub = min(pairs, 16_000_000)
        cut = []
        for bg in (1 << 21, 1 << 19):
            c.reset()
            d2 = c.dataset(20260105, 0, sub, bg, tail_group=True)
            for (p, nb, g) in d2.blocks:
                c.submit_device(p, nb)
            st2 = c.finish(True)
            cut.append((st2.counters(), st2.pairs, st2.pair_bytes, st2.groups))
            d2.close()
        assert cut[0] == cut[1]
    # ---- a slice through the streaming path (host bytes in, 64 MiB blocks and one big block)
    st, (hp, pl, pn), _sam, _scl = util.oracle_shard_summary(host, "unc", T, 0.5, 10, False, drop_last=True)
    po, so, lo, ost = util.oracle_run(host, "unc", T, 0.5, 10, False)
    assert util.lines_checksum(po)[0] == hp
    for tiles, block in ((m.TILES_AUTO, 0), (m.TILES_FAST, 64 << 20)):
        with m.Context("unc", 0.5, 10, False, T, device=0, block_bytes=block, tiles=tiles) as c:
            p, s, st, log = c.run_bytes(host, chunk=256 << 20)
        assert log == lo
        assert len(p) == len(po) and util.lines_checksum(p) == (hp, pl)
