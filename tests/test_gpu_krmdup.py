"""SURVEY.md 8(f) N2 on the GPU: mkt_rmdup_* and the bin/krmdup / bin/krmdup.pipe drop-ins against the golden vectors made by
the reference itself (tests/golden/krmdup_golden.json), against the oracle restatement and -- when the build travelled -- against
oracle/_ref/krmdup.ref run on the same input.  Bit-exact: read files byte for byte, log byte for byte."""
import json
import os

import pytest

import microcket_amd as m
import util
from test_krmdup import _case_input, _golden, _kw

pytestmark = pytest.mark.gpu


def test_rmdup_matches_reference_goldens():
    if m.device_count() < 1:
        pytest.fail("no HIP device")
    for c in _golden()["cases"]:
        text = _case_input(c)
        kw = _kw(c["args"])
        r1, r2, st = m.rmdup(text, **kw)
        assert util.krmdup_log(*st).decode() == c["log"], c["name"]
        assert util.sha(r1) == c["read1_sha256"] and util.sha(r2) == c["read2_sha256"], c["name"]
        inter, _none, st2 = m.rmdup(text, interleaved=True, **kw)
        assert st2 == st and _none == b""
        assert util.sha(b"\n".join(util.fastq_records(inter))) == c["pipe_records_sha256"], c["name"]


def test_rmdup_edge_inputs_vs_oracle():
    if m.device_count() < 1:
        pytest.fail("no HIP device")
    base = util.synth_fastq(51, 300, 40)
    for text in (b"", base, base[:-1], util.synth_fastq(52, 70000, 30, dup_rate=0.8), util.synth_fastq(53, 2000, 21, short_rate=0.5),
                 b"@a\nNNNNNNNNNNNNNNNNNNNNNNNNNNNNNN\n+\nFFFFFFFFFFFFFFFFFFFFFFFFFFFFFF\n@a\nACGTACGTACGTACGTACGTACGTACGTAC\n+\nFFFFFFFFFFFFFFFFFFFFFFFFFFFFFF\n"):
        o1, o2, ol = util.krmdup_oracle(text)
        r1, r2, st = m.rmdup(text, piece=100003)
        assert (r1, r2, util.krmdup_log(*st)) == (o1, o2, ol), len(text)


def test_rmdup_streaming_segments_equal_the_oracle(monkeypatch):
    """begin / push / push(final) with segments far smaller than the input (MKT_RMDUP_SEGMENT_MB=8: a few 2^16-pair batches each, the
    key set carried from segment to segment, growing and rehashing on the way) == the oracle == one resident run; pieces that cut
    lines and records anywhere; duplicates whose first occurrence lies segments back; an all-G key (every bit set) and the same bases
    in lower case (same key, other bucket: both stay, krmdup.cpp:105-108 / 171-176)."""
    if m.device_count() < 1:
        pytest.fail("no HIP device")
    monkeypatch.setenv("MKT_RMDUP_SEGMENT_MB", "8")
    g = "G" * 40
    odd = (f"@g1 1\n{g}\n+\n{'F' * 40}\n@g1 2\n{g}\n+\n{'F' * 40}\n"
           f"@g2 1\n{g.lower()}\n+\n{'F' * 40}\n@g2 2\n{g}\n+\n{'F' * 40}\n").encode()
    a = util.synth_fastq(71, 150000, 36, dup_rate=0.4)
    b = util.synth_fastq(72, 120000, 36, dup_rate=0.2)
    la = a.split(b"\n")
    half = b"\n".join(la[:len(la) // 16 * 8]) + b"\n"    # (whole pairs)
    text = odd + a + b + half + odd                      # the fourth part repeats pairs seen 270 000 pairs earlier
    assert text.count(b"\n") // 8 > 4 * 65536
    o1, o2, ol = util.krmdup_oracle(text)
    for piece in (1 << 22, 999983):
        r1, r2, st = m.rmdup(text, piece=piece, stream=True)
        assert util.krmdup_log(*st) == ol
        assert (util.sha(r1), util.sha(r2)) == (util.sha(o1), util.sha(o2)), piece
    q1, q2, st2 = m.rmdup(text)                           # resident form: one segment
    assert (q1, q2, util.krmdup_log(*st2)) == (o1, o2, ol)
    inter, _none, st3 = m.rmdup(text, interleaved=True, piece=1 << 22, stream=True)
    l1, l2 = o1.split(b"\n")[:-1], o2.split(b"\n")[:-1]
    assert st3 == st2 and inter == b"".join(b"\n".join(l1[i:i + 4] + l2[i:i + 4]) + b"\n" for i in range(0, len(l1), 4))


def test_krmdup_executables_are_drop_ins():
    if m.device_count() < 1:
        pytest.fail("no HIP device")
    exe = os.path.join(os.path.dirname(m.exe_path()), "krmdup")
    text = util.synth_fastq(61, 8000, 70)
    for args in ((), ("-k", "2", "-K", "4", "-s", "10", "-S", "14")):
        o1, o2, ol = util.krmdup_oracle(text, **_kw(args))
        rc, r1, r2, log, err = util.krmdup_run_cli(exe, text, False, args)
        assert rc == 0, err
        assert (r1, r2, log) == (o1, o2, ol)
        rc, so, _, log, err = util.krmdup_run_cli(exe + ".pipe", text, True, args)
        assert rc == 0 and log == ol, err
        l1, l2 = o1.split(b"\n")[:-1], o2.split(b"\n")[:-1]
        assert so == b"".join(b"\n".join(l1[i:i + 4] + l2[i:i + 4]) + b"\n" for i in range(0, len(l1), 4))
        if os.path.exists(util.KRMDUP_REF):
            rrc, q1, q2, qlog, qerr = util.krmdup_run_cli(util.KRMDUP_REF, text, False, args)
            assert (rrc, q1, q2, qlog) == (0, r1, r2, log)
    # exit codes of the reference: usage 2, key sizes 1
    import subprocess
    assert subprocess.run([exe], stderr=subprocess.PIPE).returncode == 2
    assert subprocess.run([exe, "-i", "x", "-o", "y", "-s", "3", "-S", "3"], stderr=subprocess.PIPE).returncode == 1


def test_krmdup_pipe_streams_in_the_drivers_pipe():
    """microcket:405-408: `ktrim ... | krmdup.pipe -i /dev/stdin -o prefix | flash ...`.  The executable reads a pipe that delivers the
    FASTQ in small writes, works it off in segments (MKT_RMDUP_SEGMENT_MB=8) and -- the point of streaming -- its first reads leave
    while input is still arriving: the reader below waits for output before it sends the second half."""
    if m.device_count() < 1:
        pytest.fail("no HIP device")
    import subprocess
    import tempfile
    import threading
    exe = os.path.join(os.path.dirname(m.exe_path()), "krmdup.pipe")
    text = util.synth_fastq(91, 300000, 36, dup_rate=0.3)
    o1, o2, ol = util.krmdup_oracle(text)
    l1, l2 = o1.split(b"\n")[:-1], o2.split(b"\n")[:-1]
    want = b"".join(b"\n".join(l1[i:i + 4] + l2[i:i + 4]) + b"\n" for i in range(0, len(l1), 4))
    with tempfile.TemporaryDirectory(prefix="krmp_") as d:
        pre = os.path.join(d, "o")
        p = subprocess.Popen([exe, "-i", "/dev/stdin", "-o", pre], stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                             env=dict(os.environ, MKT_RMDUP_SEGMENT_MB="8"))
        got = []
        first_out = threading.Event()

        def reader():
            while True:
                b = p.stdout.read(1 << 20)
                if not b:
                    break
                got.append(b)
                first_out.set()

        th = threading.Thread(target=reader)
        th.start()
        half = len(text) // 2
        for k in range(0, half, 1 << 16):
            p.stdin.write(text[k:min(k + (1 << 16), half)])
        p.stdin.flush()
        early = first_out.wait(60)                       # output of the first segments before the input is complete
        for k in range(half, len(text), 1 << 16):
            p.stdin.write(text[k:k + (1 << 16)])
        p.stdin.close()
        th.join()
        rc = p.wait()
        err = p.stderr.read()
        assert rc == 0, err
        assert early, "no read left the executable before its input ended"
        assert b"".join(got) == want
        assert open(pre + ".log", "rb").read() == ol
