"""TEST INFRASTRUCTURE: an independent BAM / BGZF / BAI reader and a SAM-side model, written from the published
specification (hts-specs SAMv1: 4.1 BGZF, 4.2 BAM, 5.2 BAI, 5.3 reg2bin) with Python's zlib doing the inflating
(zlib checks every block's CRC-32 and ISIZE).  It is the checker of microcket_amd's GPU BAM writer (SURVEY.md 8(f)
N3); samtools itself ships with the reference only as a prebuilt binary, which is never run: parity with samtools is
UNPINNED, this reader pins the output to the specification instead.  Product code never imports this file."""
import struct
import zlib

SEQ_CODES = "=ACMGRSVTWYHKDBN"
CIGAR_OPS = "MIDNSHP=X"


def bgzf_blocks(data: bytes):
    """Yields (compressed offset, compressed size, raw bytes) of every BGZF block; checks the framing of 4.1."""
    p = 0
    while p < len(data):
        assert data[p:p + 4] == b"\x1f\x8b\x08\x04", f"no BGZF block at {p}"
        xlen = struct.unpack_from("<H", data, p + 10)[0]
        extra = data[p + 12:p + 12 + xlen]
        bsize = None
        q = 0
        while q < len(extra):
            si1, si2, slen = extra[q], extra[q + 1], struct.unpack_from("<H", extra, q + 2)[0]
            if si1 == 66 and si2 == 67:
                assert slen == 2
                bsize = struct.unpack_from("<H", extra, q + 4)[0] + 1
            q += 4 + slen
        assert bsize is not None, "BGZF block without BC field"
        block = data[p:p + bsize]
        raw = zlib.decompress(block, wbits=31)          # gzip member: CRC-32 and ISIZE verified by zlib
        assert len(raw) <= 65536
        yield p, bsize, raw
        p += bsize


EOF_BLOCK = bytes([0x1f, 0x8b, 0x08, 0x04, 0, 0, 0, 0, 0, 0xff, 0x06, 0, 0x42, 0x43, 0x02, 0, 0x1b, 0, 0x03, 0, 0, 0, 0, 0, 0, 0, 0, 0])


class Bam:
    """header text, references [(name, length)], records [(virtual offset, end virtual offset, dict)]"""

    def __init__(self, data: bytes):
        assert data.endswith(EOF_BLOCK), "no BGZF end-of-file marker block"
        blocks = list(bgzf_blocks(data))
        self.nblocks = len(blocks)
        self.compressed = len(data)
        raw = b"".join(b[2] for b in blocks)
        self.raw_len = len(raw)
        # uncompressed offset -> virtual offset
        starts, u = [], 0
        for coff, _, r in blocks:
            starts.append((u, coff, len(r)))
            u += len(r)
        self._starts = starts

        def voff(uo):
            # the block that holds byte uo (the first non-empty one starting at or before it); the end of the data maps to the EOF block
            lo, hi = 0, len(starts) - 1
            while lo < hi:
                mid = (lo + hi + 1) // 2
                if starts[mid][0] <= uo:
                    lo = mid
                else:
                    hi = mid - 1
            return (starts[lo][1] << 16) | (uo - starts[lo][0])
        self.voff = voff
        assert raw[:4] == b"BAM\x01"
        l_text = struct.unpack_from("<i", raw, 4)[0]
        self.text = raw[8:8 + l_text].decode()
        p = 8 + l_text
        n_ref = struct.unpack_from("<i", raw, p)[0]
        p += 4
        self.refs = []
        for _ in range(n_ref):
            l_name = struct.unpack_from("<i", raw, p)[0]
            name = raw[p + 4:p + 4 + l_name]
            assert name.endswith(b"\0")
            l_ref = struct.unpack_from("<i", raw, p + 4 + l_name)[0]
            self.refs.append((name[:-1].decode(), l_ref))
            p += 8 + l_name
        self.records = []
        while p < len(raw):
            bs = struct.unpack_from("<i", raw, p)[0]
            rec = self._record(raw[p + 4:p + 4 + bs])
            assert len(raw) >= p + 4 + bs
            self.records.append((voff(p), voff(p + 4 + bs), rec))
            p += 4 + bs
        assert p == len(raw)

    def _record(self, b):
        (tid, pos, l_rn, mapq, bin_, n_cig, flag, l_seq, mtid, mpos, tlen) = struct.unpack_from("<iiBBHHHiiii", b, 0)
        p = 32
        qname = b[p:p + l_rn]
        assert qname.endswith(b"\0") and b"\0" not in qname[:-1]
        p += l_rn
        cig = struct.unpack_from("<%dI" % n_cig, b, p)
        p += 4 * n_cig
        seqb = b[p:p + (l_seq + 1) // 2]
        p += (l_seq + 1) // 2
        qual = b[p:p + l_seq]
        p += l_seq
        seq = "".join(SEQ_CODES[(seqb[i >> 1] >> (4 if i % 2 == 0 else 0)) & 15] for i in range(l_seq))
        if l_seq % 2:
            assert seqb[-1] & 15 == 0
        tags = []
        while p < len(b):
            tag = b[p:p + 2].decode()
            ty = chr(b[p + 2])
            p += 3
            if ty == "A":
                tags.append((tag, "A", chr(b[p]))); p += 1
            elif ty in "cCsSiI":
                fmt = {"c": "<b", "C": "<B", "s": "<h", "S": "<H", "i": "<i", "I": "<I"}[ty]
                v = struct.unpack_from(fmt, b, p)[0]
                p += struct.calcsize(fmt)
                tags.append((tag, ty, v))
            elif ty == "f":
                tags.append((tag, "f", struct.unpack_from("<f", b, p)[0])); p += 4
            elif ty in "ZH":
                e = b.index(b"\0", p)
                tags.append((tag, ty, b[p:e].decode())); p = e + 1
            elif ty == "B":
                sub = chr(b[p])
                cnt = struct.unpack_from("<i", b, p + 1)[0]
                fmt = {"c": "b", "C": "B", "s": "h", "S": "H", "i": "i", "I": "I", "f": "f"}[sub]
                vals = struct.unpack_from("<%d%s" % (cnt, fmt), b, p + 5)
                p += 5 + cnt * struct.calcsize(fmt)
                tags.append((tag, "B" + sub, list(vals)))
            else:
                raise AssertionError(f"tag type {ty!r}")
        assert p == len(b)
        return dict(tid=tid, pos=pos, mapq=mapq, bin=bin_, flag=flag, mtid=mtid, mpos=mpos, tlen=tlen, qname=qname[:-1].decode(),
                    cigar=[(c >> 4, c & 15) for c in cig], seq=seq, qual=bytes(qual), tags=tags)

    def sam_line(self, rec):
        """The record as SAM text (integer tags print as i, whatever their binary width)."""
        rn = lambda t: "*" if t < 0 else self.refs[t][0]
        cig = "".join(f"{l}{CIGAR_OPS[o]}" for l, o in rec["cigar"]) or "*"
        seq = rec["seq"] or "*"
        if not rec["qual"] or all(q == 0xFF for q in rec["qual"]):
            qual = "*"
        else:
            qual = "".join(chr(q + 33) for q in rec["qual"])
        rnext = "=" if rec["mtid"] >= 0 and rec["mtid"] == rec["tid"] else rn(rec["mtid"])
        f = [rec["qname"], str(rec["flag"]), rn(rec["tid"]), str(rec["pos"] + 1), str(rec["mapq"]), cig, rnext, str(rec["mpos"] + 1), str(rec["tlen"]), seq, qual]
        for tag, ty, v in rec["tags"]:
            if ty in ("c", "C", "s", "S", "i", "I"):
                f.append(f"{tag}:i:{v}")
            elif ty == "f":
                f.append(f"{tag}:f:{v:g}")
            elif ty.startswith("B"):
                f.append(f"{tag}:B:{ty[1]}" + "".join("," + (f"{x:g}" if ty[1] == "f" else str(x)) for x in v))
            else:
                f.append(f"{tag}:{ty}:{v}")
        return "\t".join(f)


def reg2bin(beg, end):
    end -= 1
    if beg >> 14 == end >> 14: return ((1 << 15) - 1) // 7 + (beg >> 14)
    if beg >> 17 == end >> 17: return ((1 << 12) - 1) // 7 + (beg >> 17)
    if beg >> 20 == end >> 20: return ((1 << 9) - 1) // 7 + (beg >> 20)
    if beg >> 23 == end >> 23: return ((1 << 6) - 1) // 7 + (beg >> 23)
    if beg >> 26 == end >> 26: return ((1 << 3) - 1) // 7 + (beg >> 26)
    return 0


def reg2bins(beg, end):
    """5.3: the bins that may hold records overlapping [beg, end)"""
    end -= 1
    out = [0]
    for sh, base in ((26, 1), (23, 9), (20, 73), (17, 585), (14, 4681)):
        out.extend(range(base + (beg >> sh), base + (end >> sh) + 1))
    return out


def ref_span(rec):
    """[pos, end) on the reference as an indexer sees the record: CIGAR M/D/N/=/X, one base for unmapped reads or empty CIGARs"""
    rlen = sum(l for l, o in rec["cigar"] if o in (0, 2, 3, 7, 8))
    if rec["flag"] & 4 or rlen == 0:
        rlen = 1
    return rec["pos"], rec["pos"] + rlen


class Bai:
    def __init__(self, data: bytes):
        assert data[:4] == b"BAI\x01"
        n_ref = struct.unpack_from("<i", data, 4)[0]
        p = 8
        self.refs = []
        for _ in range(n_ref):
            n_bin = struct.unpack_from("<i", data, p)[0]
            p += 4
            bins, meta = {}, None
            for _ in range(n_bin):
                b, n_chunk = struct.unpack_from("<Ii", data, p)
                p += 8
                chunks = [struct.unpack_from("<QQ", data, p + 16 * k) for k in range(n_chunk)]
                p += 16 * n_chunk
                if b == 37450:
                    assert n_chunk == 2
                    meta = chunks
                else:
                    assert b < 37450 and b not in bins
                    bins[b] = chunks
            n_intv = struct.unpack_from("<i", data, p)[0]
            p += 4
            lin = list(struct.unpack_from("<%dQ" % n_intv, data, p))
            p += 8 * n_intv
            self.refs.append((bins, lin, meta))
        self.n_no_coor = None
        if p + 8 <= len(data):
            self.n_no_coor = struct.unpack_from("<Q", data, p)[0]
            p += 8
        assert p == len(data)

    def query_chunks(self, tid, beg, end):
        """The file ranges an index-driven reader visits for [beg, end): candidate bins, cut by the linear index (5.1.3)."""
        bins, lin, _ = self.refs[tid]
        w = beg >> 14
        min_off = lin[w] if w < len(lin) else (lin[-1] if lin else 0)
        if w >= len(lin):
            min_off = 0 if not lin else min_off      # beyond the last window: nothing can be skipped safely except via bins
        out = []
        for b in reg2bins(beg, end):
            for c0, c1 in bins.get(b, ()):
                if c1 > min_off:
                    out.append((c0, c1))
        return sorted(out)


def sam_sort_key(line: str, ref_ids: dict):
    """(reference id with unmapped last, position, reverse strand) of a SAM line: samtools sort's coordinate order"""
    f = line.split("\t")
    tid = ref_ids[f[2]] if f[2] != "*" else len(ref_ids)
    return (tid, int(f[3]), 1 if int(f[1]) & 16 else 0)


def expected_int_type(v):
    """htslib's choice for a SAM :i: value"""
    if v < 0:
        return "c" if v >= -128 else ("s" if v >= -32768 else "i")
    return "C" if v <= 255 else ("S" if v <= 65535 else "I")


def normalise_sam_line(line: str) -> str:
    """A SAM line as Bam.sam_line prints its record: SEQ in upper case, float values in %g form"""
    f = line.split("\t")
    f[9] = f[9].upper()
    for k in range(11, len(f)):
        tag, ty, v = f[k].split(":", 2)
        if ty == "f":
            f[k] = f"{tag}:f:{float(v):g}"
        elif ty == "B" and v[:1] == "f":
            f[k] = f"{tag}:B:f" + "".join("," + f"{float(x):g}" for x in v.split(",")[1:])
    return "\t".join(f)
