"""The BAM / BGZF / BAI checker of tests/test_gpu_bam.py, checked on its own (no GPU): a BAM assembled here with struct + zlib by
the rules of the specification must read back, and the helper functions must agree with the specification's worked values."""
import struct
import zlib

import pytest

import bamio


def bgzf(raw: bytes, level=6) -> bytes:
    c = zlib.compressobj(level, zlib.DEFLATED, -15)
    d = c.compress(raw) + c.flush()
    return (b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", len(d) + 25) + d + struct.pack("<II", zlib.crc32(raw), len(raw)))


def test_reader_on_a_hand_made_bam():
    text = "@HD\tVN:1.6\tSO:coordinate\n@SQ\tSN:ref\tLN:45\n"
    raw = b"BAM\1" + struct.pack("<i", len(text)) + text.encode() + struct.pack("<i", 1) + struct.pack("<i", 4) + b"ref\0" + struct.pack("<i", 45)
    # the specification's example read r001 (SAMv1 1.1): 99 ref 7 30 8M2I4M1D3M = 37 39 TTAGATAAAGGATACTG *
    cigar = [(8, 0), (2, 1), (4, 0), (1, 2), (3, 0)]
    seq = "TTAGATAAAGGATACTG"
    packed = bytearray()
    for i in range(0, len(seq), 2):
        hi = bamio.SEQ_CODES.index(seq[i])
        lo = bamio.SEQ_CODES.index(seq[i + 1]) if i + 1 < len(seq) else 0
        packed.append(hi << 4 | lo)
    body = struct.pack("<iiBBHHHiiii", 0, 6, 5, 30, bamio.reg2bin(6, 22), len(cigar), 99, len(seq), 0, 36, 39) + b"r001\0"
    body += b"".join(struct.pack("<I", l << 4 | o) for l, o in cigar) + bytes(packed) + b"\xff" * len(seq)
    body += b"NMC\x01" + b"XSs" + struct.pack("<h", -300) + b"RGZgrp\0" + b"XBBS" + struct.pack("<iHH", 2, 7, 9)
    raw += struct.pack("<i", len(body)) + body
    data = bgzf(raw[:50]) + bgzf(raw[50:]) + bamio.EOF_BLOCK
    bam = bamio.Bam(data)
    assert bam.refs == [("ref", 45)] and bam.text == text and len(bam.records) == 1
    v0, v1, r = bam.records[0]
    assert bam.sam_line(r) == "r001\t99\tref\t7\t30\t8M2I4M1D3M\t=\t37\t39\tTTAGATAAAGGATACTG\t*\tNM:i:1\tXS:i:-300\tRG:Z:grp\tXB:B:S,7,9"
    assert bamio.ref_span(r) == (6, 22)
    # the record starts in the second block
    first_len = len(bgzf(raw[:50]))
    assert v0 >> 16 == first_len and v0 & 0xFFFF == len(raw) - 50 - 4 - len(body)
    with pytest.raises(zlib.error):                                     # a flipped payload bit: zlib's CRC check
        bad = bytearray(data)
        bad[first_len + 30] ^= 1
        bamio.Bam(bytes(bad))
    with pytest.raises(AssertionError):
        bamio.Bam(data[:-28])                                           # no end-of-file block


def test_bins_and_types():
    # SAMv1 5.3: bin numbers of the five levels
    assert bamio.reg2bin(0, 1) == 4681 and bamio.reg2bin(0, 1 << 14) == 4681 and bamio.reg2bin(0, (1 << 14) + 1) == 585
    assert bamio.reg2bin((1 << 26) - 1, (1 << 26) + 1) == 0 and bamio.reg2bin(1 << 26, (1 << 26) + 1) == 4681 + (1 << 12)
    assert bamio.reg2bin(-1, 0) == 4680                                # unplaced: POS 0
    for beg, end in ((0, 1), (16383, 16385), (100000, 5000000), (1 << 28, (1 << 28) + 70000)):
        assert bamio.reg2bin(beg, end) in bamio.reg2bins(beg, end)
    assert [bamio.expected_int_type(v) for v in (0, 255, 256, 65535, 65536, -1, -128, -129, -32768, -32769)] == list("CCSSIccssi")
    assert bamio.normalise_sam_line("q\t0\tc\t1\t0\t1M\t*\t0\t0\tacgt\tIIII\tXf:f:1e3\tXB:B:f,1.50,2") == "q\t0\tc\t1\t0\t1M\t*\t0\t0\tACGT\tIIII\tXf:f:1000\tXB:B:f,1.5,2"


def test_bai_reader_round_trip():
    chunks = {4681: [(100 << 16, 200 << 16)], 585: [(200 << 16 | 5, 300 << 16)]}
    b = b"BAI\1" + struct.pack("<i", 2)
    b += struct.pack("<i", 3)
    for k, cs in chunks.items():
        b += struct.pack("<Ii", k, len(cs)) + b"".join(struct.pack("<QQ", *c) for c in cs)
    b += struct.pack("<Ii", 37450, 2) + struct.pack("<QQQQ", 100 << 16, 300 << 16, 7, 1)
    b += struct.pack("<i", 2) + struct.pack("<QQ", 100 << 16, 200 << 16 | 5)
    b += struct.pack("<i", 0) + struct.pack("<i", 0)
    b += struct.pack("<Q", 3)
    bai = bamio.Bai(b)
    assert bai.n_no_coor == 3 and bai.refs[0][2][1] == (7, 1) and bai.refs[1] == ({}, [], None)
    assert bai.query_chunks(0, 0, 10) == [(100 << 16, 200 << 16), (200 << 16 | 5, 300 << 16)]
    assert bai.query_chunks(0, 16384, 16400) == [(200 << 16 | 5, 300 << 16)]          # the second 16 kb window: bin 4681 is no candidate
