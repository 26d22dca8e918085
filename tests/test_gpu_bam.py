"""SURVEY.md 8(f) N3 on the GPU: SAM text -> coordinate-sorted BGZF BAM + BAI (mkt_bam_*, bin/sam2bam).

The checker is tests/bamio.py (an independent reader written from the SAM/BAM specification; Python's zlib inflates every
block and verifies its CRC-32 / ISIZE).  samtools ships with the reference only as a prebuilt binary that is never run, so
parity with samtools' exact bytes is UNPINNED; what is pinned: the decoded records equal the input lines, the order is
(reference id, position, strand) with ties in input order, the header carries the @SQ dictionary and SO:coordinate, integer
tags use the smallest type, `bin` is reg2bin of the alignment, and every query through the .bai finds exactly the records a
scan finds."""
import os
import random
import subprocess

import pytest

import bamio
import util

pytestmark = pytest.mark.gpu

CHROMS = ["chr1", "chr10", "chr11", "chr12", "chr13", "chr14", "chr15", "chr16", "chr17", "chr18", "chr19", "chr2", "chr20", "chr21", "chr22",
          "chr3", "chr4", "chr5", "chr6", "chr7", "chr8", "chr9", "chrM", "chrX", "chrY"]


def header_for(sam: bytes, extra=(), hd=None):
    """@SQ lines for every reference name in the text (in a fixed, non-alphabetical order), 250 Mb each"""
    names = set()
    for ln in sam.split(b"\n"):
        f = ln.split(b"\t")
        if len(f) > 6:
            for x in (f[2], f[6]):
                if x not in (b"*", b"="):
                    names.add(x.decode())
    order = [c for c in reversed(CHROMS) if c in names] + sorted(names - set(CHROMS)) + list(extra)
    h = "" if hd is None else hd
    for c in order:
        h += f"@SQ\tSN:{c}\tLN:250000000\n"
    h += "@PG\tID:bwa\tPN:bwa\tVN:0.7.17\tCL:bwa mem -5 -S -P\n"
    return h.encode(), order


def check_bam(sam_header: bytes, body: bytes, bam_bytes: bytes, bai_bytes: bytes, nrec, sorted_=True, order=None):
    bam = bamio.Bam(bam_bytes)
    lines = [ln for ln in body.decode().split("\n") if ln]
    assert nrec == len(lines) == len(bam.records)
    ids = {n: i for i, n in enumerate(order)}
    assert [r[0] for r in bam.refs] == order
    assert all(r[1] == 250000000 for r in bam.refs)
    want_text = sam_header.decode()
    if sorted_:
        assert bam.text.split("\n")[0].startswith("@HD") and "SO:coordinate" in bam.text.split("\n")[0]
        assert [ln for ln in bam.text.split("\n") if not ln.startswith("@HD")] == [ln for ln in want_text.split("\n") if not ln.startswith("@HD")]
        lines = sorted(lines, key=lambda ln: bamio.sam_sort_key(ln, ids))          # Python's sort is stable: ties in input order
    else:
        assert bam.text == want_text
    got = [bam.sam_line(r[2]) for r in bam.records]
    assert got == [bamio.normalise_sam_line(ln) for ln in lines]
    for _, _, r in bam.records:
        b, e = bamio.ref_span(r)
        assert r["bin"] == bamio.reg2bin(b, e)
        for tag, ty, v in r["tags"]:
            if ty in "cCsSiI":
                assert ty == bamio.expected_int_type(v), (tag, ty, v)
    if not sorted_:
        assert bai_bytes == b""
        return bam
    bai = bamio.Bai(bai_bytes)
    assert len(bai.refs) == len(order)
    assert bai.n_no_coor == sum(1 for _, _, r in bam.records if r["tid"] < 0)
    by_tid = {}
    for v0, v1, r in bam.records:
        if r["tid"] >= 0:
            by_tid.setdefault(r["tid"], []).append((v0, v1, r))
    rng = random.Random(7)
    for tid, (bins, lin, meta) in enumerate(bai.refs):
        recs = by_tid.get(tid, [])
        if not recs:
            assert not bins and meta is None
            continue
        assert meta[0][0] == recs[0][0] and meta[0][1] >= recs[-1][0]
        assert meta[1] == (sum(1 for x in recs if not x[2]["flag"] & 4), sum(1 for x in recs if x[2]["flag"] & 4))
        # every record sits in a chunk of its own bin
        for v0, v1, r in recs:
            b, e = bamio.ref_span(r)
            assert any(c0 <= v0 < c1 for c0, c1 in bins[bamio.reg2bin(b, e)])
        # region queries: what the index lets a reader visit contains everything a scan finds
        spans = [bamio.ref_span(x[2]) for x in recs]
        hi = max(e for _, e in spans)
        for _ in range(40):
            qb = rng.randrange(0, hi + 1)
            qe = qb + rng.choice([1, 100, 5000, 20000, 200000, 5000000])
            chunks = bai.query_chunks(tid, qb, qe)
            for (v0, v1, r), (b, e) in zip(recs, spans):
                if b < qe and e > qb:
                    assert any(c0 <= v0 < c1 for c0, c1 in chunks), (tid, qb, qe, r["qname"], b, e)
        # linear index: no window's offset lies behind a record that overlaps it
        for (v0, v1, r), (b, e) in zip(recs, spans):
            for w in range(max(b, 0) >> 14, ((e - 1) >> 14) + 1):
                assert w < len(lin) and lin[w] <= v0
    return bam


EDGE_BODY = "\n".join([
    "r001\t99\tchr2\t7\t30\t8M2I4M1D3M\t=\t37\t39\tTTAGATAAAGGATACTG\t*\tNM:i:0\tXA:A:q\tXf:f:3.25\tXh:H:1AE301\tXb:B:c,-1,2,-128\tXs:B:S,0,65535\tXF:B:f,1.5,-2,1e3",
    "r002\t0\tchr2\t9\t30\t3S6M1P1I4M\t*\t0\t0\tAAAAGATAAGGATA\t" + "I" * 14 + "\tXi:i:-1\tXj:i:-129\tXk:i:-32769\tXl:i:255\tXm:i:256\tXn:i:65536\tXo:i:4294967295\tXp:i:-2147483648",
    "r003\t4\t*\t0\t0\t*\t*\t0\t0\tNNNRYK\t!!!~~~",
    "r004\t16\tchr10\t250\t255\t6H5=1X2N10M\tchr2\t7\t-100\tacgtnACGTNacgtnACG\t" + "5" * 18 + "\tRG:Z:grp 1\tCO:Z:",
    "r005\t73\tchrM\t16384\t0\t20M\t=\t16384\t0\tACGTACGTACGTACGTACGT\t" + "F" * 20,
    "r006\t133\tchrM\t16384\t0\t*\t=\t16384\t0\tACGTACGTACGTACGTACG\t" + "F" * 19,
    "r007\t0\tchr10\t250\t3\t1M\t*\t0\t0\tA\t*",
    "r008\t16\tchr10\t250\t3\t5M\t*\t0\t0\t*\t*",
    "r009\t0\tchr10\t250\t3\t70000M\t*\t0\t0\t*\t*\tXz:Z:" + "z" * 300,
    "r011\t256\tchr10\t1\t0\t3M\t*\t0\t0\tAC=\t#+5",
]) + "\n"


@pytest.mark.parametrize("level", [0, 1, 2])
def test_edge_records(level):
    import microcket_amd as m
    body = EDGE_BODY.encode()
    hdr, order = header_for(body, hd="@HD\tVN:1.5\tSO:unsorted\tGO:query\n")
    bam_b, bai_b, n = m.sam_to_bam(hdr + body, sorted=True, level=level)
    bam = check_bam(hdr, body, bam_b, bai_b, n, True, order)
    by = {r["qname"]: r for _, _, r in bam.records}
    assert by["r001"]["tags"][2][0] == "Xf" and abs(by["r001"]["tags"][2][2] - 3.25) == 0
    assert by["r001"]["tags"][6] == ("XF", "Bf", [1.5, -2.0, 1000.0])
    assert by["r006"]["flag"] == 133                                   # already unmapped
    assert by["r004"]["seq"] == "ACGTNACGTNACGTNACG"                    # lower case reads as upper case
    assert by["r007"]["qual"] == b"\xff"
    # input order (samtools view -b), no index
    bam_b, bai_b, n = m.sam_to_bam(hdr + body, sorted=False, level=level)
    check_bam(hdr, body, bam_b, bai_b, n, False, order)


@pytest.mark.parametrize("profile,groups,level", [("unc", 3000, 0), ("unc", 3000, 1), ("unc", 3000, 2), ("flash", 2500, 2), ("stress", 2000, 2)])
def test_synthetic_sam_round_trip(profile, groups, level):
    import microcket_amd as m
    body = util.synth(profile, 11, groups)
    if profile == "stress":       # the stress profile holds lines that are not SAM (fewer fields, bad CIGARs): keep the well-formed ones
        keep = []
        for ln in body.split(b"\n"):
            f = ln.split(b"\t")
            if len(f) >= 11 and f[5] != b"*" and all(c in b"0123456789MIDNSHP=X" for c in f[5]) and f[5][-1:] not in b"0123456789" \
                    and f[5][:1] in b"0123456789" and f[1].isdigit() and f[3].isdigit() and f[4].isdigit() and len(f[9]) == len(f[10]) and int(f[4]) < 256:
                keep.append(ln)
        body = b"\n".join(keep) + b"\n"
    hdr, order = header_for(body)
    bam_b, bai_b, n = m.sam_to_bam(hdr + body, sorted=True, level=level, piece=1 << 20)
    bam = check_bam(hdr, body, bam_b, bai_b, n, True, order)
    assert bam.nblocks > 3
    if level:
        assert bam.compressed < 0.8 * bam.raw_len          # (fixed Huffman codes + matches: the synthetic text has long runs of F)


def test_deflate_on_repetitive_and_random_data():
    """match lengths around 255..258, distances 1, 2, 300 and none at all; every block must inflate to the stored block's bytes"""
    import microcket_amd as m
    rng = random.Random(3)
    rnd = lambda k: "".join(rng.choice("ACGTNacgtn0123456789:;<=>?@") for _ in range(k))
    lines = []
    blk = rnd(300)
    for i, z in enumerate(["A" * 1000, "AB" * 600, blk * 4, rnd(2000), "Q" * 255, "Q" * 256 + "x", "R" * 257 + "y", "S" * 258 + "z", "T" * 259, "U" * 262 + "ab" * 131, rnd(7) * 200]):
        for rep in range(40):
            q = "".join(chr(33 + rng.randrange(0, 41)) for _ in range(50))
            sq = "".join(rng.choice("ACGT") for _ in range(50))
            lines.append(f"read{i}_{rep}\t0\tchr1\t{1 + rng.randrange(10 ** 6)}\t60\t50M\t*\t0\t0\t{sq}\t{q}\tXZ:Z:{z}\tXr:Z:{rnd(rng.randrange(0, 40))}")
    body = ("\n".join(lines) + "\n").encode()
    hdr = b"@SQ\tSN:chr1\tLN:250000000\n"
    for sorted_ in (False, True):
        b0, _, n0 = m.sam_to_bam(hdr + body, sorted=sorted_, level=0)
        raw0 = b"".join(r for _, _, r in bamio.bgzf_blocks(b0))
        sizes = []
        for level in (1, 2):
            b1, i1, n1 = m.sam_to_bam(hdr + body, sorted=sorted_, level=level)
            raw1 = b"".join(r for _, _, r in bamio.bgzf_blocks(b1))
            assert raw0 == raw1 and n0 == n1 == len(lines)
            sizes.append(len(b1))
        assert sizes[0] < 0.5 * len(b0) and sizes[1] < sizes[0]
    check_bam(hdr + b"@PG\tID:bwa\tPN:bwa\tVN:0.7.17\tCL:bwa mem -5 -S -P\n", body, *m.sam_to_bam(hdr + b"@PG\tID:bwa\tPN:bwa\tVN:0.7.17\tCL:bwa mem -5 -S -P\n" + body), True, ["chr1"])


def test_many_blocks_and_chunked_input():
    import microcket_amd as m
    body = util.synth("unc", 23, 40000)
    hdr, order = header_for(body)
    bam_b, bai_b, n = m.sam_to_bam(hdr + body, sorted=True, level=2, piece=(1 << 20) + 12345)
    bam = check_bam(hdr, body, bam_b, bai_b, n, True, order)
    assert bam.nblocks > 300


def test_errors_are_reported():
    import microcket_amd as m
    hdr = b"@SQ\tSN:chr1\tLN:1000\n"
    ok = b"r\t0\tchr1\t1\t0\t1M\t*\t0\t0\tA\tI\n"
    for bad in (b"r\t0\tchr9\t1\t0\t1M\t*\t0\t0\tA\tI\n",            # reference not in the header
                b"r\t0\tchr1\t1\t0\t1M\t*\t0\t0\tA\n",                # ten fields
                b"r\t0\tchr1\t1\t0\t1Q\t*\t0\t0\tA\tI\n",             # CIGAR
                b"r\t0\tchr1\t1\t0\t1M\t*\t0\t0\tAC\tI\n",            # SEQ / QUAL lengths
                b"r\t0\tchr1\t1\t0\t1M\t*\t0\t0\tA\tI\tNM:i:x\n",     # tag value
                b"r\t0\tchr1\tx\t0\t1M\t*\t0\t0\tA\tI\n"):            # POS
        with pytest.raises(m.MktError):
            m.sam_to_bam(hdr + ok + bad)
    bam_b, bai_b, n = m.sam_to_bam(hdr + ok)
    assert n == 1
    big = b"@SQ\tSN:chrBig\tLN:600000000\n"                          # beyond the BAI format: a BAM, but no index
    bam_b, bai_b, n = m.sam_to_bam(big + b"r\t0\tchrBig\t550000000\t0\t1M\t*\t0\t0\tA\tI\n")
    assert n == 1 and bai_b == b"" and bamio.Bam(bam_b).records[0][2]["pos"] == 549999999
    # a POSITION beyond the BAI format, or beyond what the linear index of its (short) reference holds: still a BAM, no index, the
    # reason noted
    for pos in (600000000, (1 << 29) + 1, 40000):
        notes = []
        bam_b, bai_b, n = m.sam_to_bam(hdr + ok + b"q\t0\tchr1\t%d\t0\t1M\t*\t0\t0\tA\tI\n" % pos, notes=notes)
        assert n == 2 and bai_b == b"" and "no index" in notes[0], (pos, notes)
        assert sorted(r[2]["pos"] for r in bamio.Bam(bam_b).records) == [0, pos - 1]
    for pos in (1000, 5000):                                          # the last base of the reference / past LN but inside its last index window
        notes = []
        bam_b, bai_b, n = m.sam_to_bam(hdr + ok + b"q\t0\tchr1\t%d\t0\t1M\t*\t0\t0\tA\tI\n" % pos, notes=notes)
        assert n == 2 and bai_b != b"" and notes == [""]
    # integer B arrays: every element inside its subtype's range
    for tag, good in ((b"XB:B:c,-128,127", True), (b"XB:B:c,300", False), (b"XB:B:C,255", True), (b"XB:B:C,-1", False), (b"XB:B:s,-32768,32767", True),
                      (b"XB:B:S,65536", False), (b"XB:B:i,-2147483648", True), (b"XB:B:I,4294967295", True), (b"XB:B:I,4294967296", False)):
        line = b"r\t0\tchr1\t1\t0\t1M\t*\t0\t0\tA\tI\t" + tag + b"\n"
        if good:
            assert m.sam_to_bam(hdr + line)[2] == 1
        else:
            with pytest.raises(m.MktError):
                m.sam_to_bam(hdr + line)
    # "mapped query cannot have zero coordinate; treated as unmapped" (htslib): POS 0 or RNAME * without FLAG 4
    bam_b, bai_b, n = m.sam_to_bam(hdr + b"z\t0\tchr1\t0\t0\t1M\t*\t0\t0\tA\tI\n" + b"y\t16\t*\t0\t0\t*\t*\t0\t0\tA\tI\n")
    assert sorted(r[2]["flag"] for r in bamio.Bam(bam_b).records) == [4, 20]
    bam_b, bai_b, n = m.sam_to_bam(hdr)                               # header only
    assert n == 0 and bamio.Bam(bam_b).refs == [("chr1", 1000)]
    bam_b, bai_b, n = m.sam_to_bam(b"")
    assert n == 0 and bamio.Bam(bam_b).refs == []


def test_executable_replaces_view_sort_index(tmp_path):
    """bin/sam2bam: `cat header a.sam b.sam | samtools view -b | samtools sort -o x.bam; samtools index x.bam` (microcket:533-540)"""
    import microcket_amd.build as b
    exe = b.SAM2BAM
    assert os.path.exists(exe)
    a = util.synth("flash", 5, 600)
    c = util.synth("unc", 6, 700)
    hdr, order = header_for(a + c)
    (tmp_path / "h.sam").write_bytes(hdr)
    (tmp_path / "a.sam").write_bytes(a)
    (tmp_path / "c.sam").write_bytes(c)
    out = tmp_path / "x.valid.bam"
    r = subprocess.run([exe, "-@", "4", "-o", str(out), str(tmp_path / "h.sam"), str(tmp_path / "a.sam"), str(tmp_path / "c.sam")], capture_output=True)
    assert r.returncode == 0, r.stderr
    check_bam(hdr, a + c, out.read_bytes(), (tmp_path / "x.valid.bam.bai").read_bytes(), len((a + c).splitlines()), True, order)
    # the same through a pipe, unsorted
    r = subprocess.run([exe, "-u", "-o", "-", "-"], input=hdr + a + c, capture_output=True)
    assert r.returncode == 0, r.stderr
    # stdout appended to a regular file (>>) and handed over at an offset: the BAM follows what is there
    for mode in ("ab", "r+b"):
        tgt = tmp_path / ("app_" + mode[0] + ".bin")
        tgt.write_bytes(b"PREFIX--")
        with open(tgt, mode) as fo:
            if mode == "r+b":
                fo.seek(8)
            rr = subprocess.run([exe, "-o", "-", str(tmp_path / "h.sam"), str(tmp_path / "a.sam"), str(tmp_path / "c.sam")], stdout=fo, stderr=subprocess.PIPE)
        assert rr.returncode == 0, rr.stderr
        got = tgt.read_bytes()
        assert got[:8] == b"PREFIX--" and got[8:] == out.read_bytes(), mode
    # an index that cannot be written is announced
    (tmp_path / "far.sam").write_bytes(b"far\t0\t" + hdr.split(b"SN:")[1].split(b"\t")[0] + b"\t600000000\t60\t1M\t*\t0\t0\tA\tI\n")
    rr = subprocess.run([exe, "-o", str(tmp_path / "far.bam"), str(tmp_path / "h.sam"), str(tmp_path / "far.sam")], capture_output=True)
    assert rr.returncode == 0 and b"WARN" in rr.stderr and b"no index" in rr.stderr and not (tmp_path / "far.bam.bai").exists(), rr.stderr
    check_bam(hdr, a + c, r.stdout, b"", len((a + c).splitlines()), False, order)
    r = subprocess.run([exe, "-o", str(out), str(tmp_path / "nothing.sam")], capture_output=True)
    assert r.returncode != 0
