"""Shared helpers for the test-suite: oracle binding (CHECKER only), reference runner, generators."""
import ctypes as C
import hashlib
import os
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_SO = os.path.join(ROOT, "oracle", "_build", "liboracle.so")
ORACLE_EXE = os.path.join(ROOT, "oracle", "_build", "sam2pairs_oracle")
REF_EXE = os.path.join(ROOT, "oracle", "_ref", "sam2pairs.ref")
EMUL_SO = os.path.join(ROOT, "tests", "host", "_build", "libmkt_emul.so")
SYNTH_EXE = os.path.join(ROOT, "tools", "_build", "synth_sam")
GOLDEN = os.path.join(ROOT, "tests", "golden")

MODES = {"flash": 0, "unc": 1}


_built_once = False


def ensure_built():
    """Builds the CPU-side test tools (oracle, emulation, generator) if missing."""
    global _built_once
    if _built_once:
        return
    import sys
    sys.path.insert(0, ROOT)
    from microcket_amd import build
    # mtime-checked (make / build._newer): an edited header rebuilds the emulation and the oracle instead of testing stale code
    build.build_oracle()
    build.build_test_tools()
    for p in (ORACLE_SO, ORACLE_EXE, EMUL_SO, SYNTH_EXE):
        assert os.path.exists(p), p
    _built_once = True


class _OP(C.Structure):
    _fields_ = [("mode", C.c_int), ("threads", C.c_int), ("ratio", C.c_float), ("min_mapq", C.c_int), ("write_sam", C.c_int)]


class _OB(C.Structure):
    _fields_ = [("p", C.c_void_p), ("n", C.c_size_t), ("cap", C.c_size_t)]


class OStats(C.Structure):
    _fields_ = [(k, C.c_uint32) for k in "lowMap manyHits unpaired selfCircle trans cis10K cis1K cis0 selfCircle_all".split()] + \
               [(k, C.c_uint64) for k in "lines records groups pairs".split()]


_oracle = None


def oracle_run(text: bytes, mode, threads=4, ratio=0.5, mapq=10, sam=True):
    """The CPU restatement (oracle/): returns (pairs, sam, log, stats). Input order output."""
    global _oracle
    ensure_built()
    if _oracle is None:
        _oracle = C.CDLL(ORACLE_SO)
    if isinstance(mode, str):
        mode = MODES[mode]
    p = _OP(mode, threads, ratio, mapq, 1 if sam else 0)
    a, b, st = _OB(), _OB(), OStats()
    rc = _oracle.orc_run(text, C.c_size_t(len(text)), C.byref(p), C.byref(a), C.byref(b), C.byref(st))
    assert rc == 0
    pairs = C.string_at(a.p, a.n) if a.n else b""
    s = C.string_at(b.p, b.n) if b.n else b""
    log = C.create_string_buffer(512)
    _oracle.orc_format_log(C.byref(st), log, 512)
    _oracle.orc_buf_free(C.byref(a))
    _oracle.orc_buf_free(C.byref(b))
    return pairs, s, log.value, st


def _oracle_lib():
    global _oracle
    ensure_built()
    if _oracle is None:
        _oracle = C.CDLL(ORACLE_SO)
    _oracle.orc_lines_checksum.restype = C.c_uint64
    _oracle.orc_lines_checksum.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_uint64)]
    _oracle.orc_run_shard.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_int, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    return _oracle


def lines_checksum(buf):
    """(order-independent 64-bit checksum of the lines, number of lines) of bytes or a numpy uint8 array (oracle/: checker only)."""
    L = _oracle_lib()
    n = C.c_uint64()
    if isinstance(buf, (bytes, bytearray)):
        h = L.orc_lines_checksum(C.c_char_p(bytes(buf)), len(buf), C.byref(n))
    else:
        h = L.orc_lines_checksum(C.c_void_p(buf.ctypes.data), buf.size, C.byref(n))
    return int(h), int(n.value)


def oracle_shard_summary(text, mode, threads=4, ratio=0.5, mapq=10, sam=False, drop_last=False):
    """One shard (whole QNAME groups; numpy uint8 array or bytes) through the CPU restatement without keeping its outputs:
    returns (stats, (pairs checksum, lines, bytes), (sam checksum, lines, bytes), local indices of the self-circle groups).
    The calls release the GIL: run several shards in a thread pool."""
    import numpy as np
    L = _oracle_lib()
    if isinstance(mode, str):
        mode = MODES[mode]
    p = _OP(mode, threads, ratio, mapq, 1 if sam else 0)
    a, b, sc, st = _OB(), _OB(), _OB(), OStats()
    if isinstance(text, (bytes, bytearray)):
        ptr, n = C.cast(C.c_char_p(bytes(text)), C.c_void_p), len(text)
    else:
        ptr, n = C.c_void_p(text.ctypes.data), text.size
    rc = L.orc_run_shard(ptr, n, C.byref(p), 1 if drop_last else 0, 0, 1 << 62, C.byref(a), C.byref(b), C.byref(sc), C.byref(st))
    assert rc == 0
    ln = C.c_uint64()
    hp = L.orc_lines_checksum(C.c_void_p(a.p), a.n, C.byref(ln)); pl = int(ln.value)
    hs = L.orc_lines_checksum(C.c_void_p(b.p), b.n, C.byref(ln)); sl = int(ln.value)
    scl = np.frombuffer(C.string_at(sc.p, sc.n), dtype=np.uint64).copy() if sc.n else np.zeros(0, dtype=np.uint64)
    out = (st, (int(hp), pl, int(a.n)), (int(hs), sl, int(b.n)), scl)
    for x in (a, b, sc):
        L.orc_buf_free(C.byref(x))
    return out


def selfcircle_logged(g, K, threads):
    return bool(_oracle_lib().orc_selfcircle_logged(C.c_uint64(int(g)), C.c_uint64(int(K)), C.c_int(threads)))


def have_ref():
    return os.path.exists(REF_EXE)


def _run_cli(exe, text, mode, threads, ratio, mapq, sam, env=None):
    with tempfile.TemporaryDirectory(prefix="mkt_") as d:
        inp = os.path.join(d, "in.sam")
        with open(inp, "wb") as f:
            f.write(text)
        prefix = os.path.join(d, "out")
        e = dict(os.environ)
        if env:
            e.update(env)
        p = subprocess.run([exe, inp, mode, prefix, str(threads), repr(ratio), str(mapq), "yes" if sam else "no"],
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=e)
        log = b""
        lp = f"{prefix}.{mode}2pairs.log"
        if os.path.exists(lp):
            log = open(lp, "rb").read()
        s = b""
        sp = f"{prefix}.{mode}.sam"
        if os.path.exists(sp):
            s = open(sp, "rb").read()
        return p.returncode, p.stdout, s, log, p.stderr


def ref_run(text, mode, threads=4, ratio=0.5, mapq=10, sam=True):
    """The reference itself (oracle/_ref, compiled from /root/reference by oracle/Makefile)."""
    return _run_cli(REF_EXE, text, mode, threads, ratio, mapq, sam)


def cli_run(exe, text, mode, threads=4, ratio=0.5, mapq=10, sam=True, env=None):
    return _run_cli(exe, text, mode, threads, ratio, mapq, sam, env)


def canon(b: bytes) -> bytes:
    """LANG=C sort of lines (the driver's canonical form, microcket:480)."""
    lines = b.split(b"\n")
    if lines and lines[-1] == b"":
        lines.pop()
    lines.sort()
    return b"\n".join(lines) + (b"\n" if lines else b"")


def sha(b: bytes) -> str:
    return hashlib.sha256(b).hexdigest()


def synth(profile, seed, n_groups, read_len=150, genome="hg38", lanes=1, tail=1, first=0) -> bytes:
    ensure_built()
    return subprocess.run([SYNTH_EXE, profile, str(seed), str(n_groups), str(read_len), genome, str(lanes), str(tail), str(first)],
                          stdout=subprocess.PIPE, check=True).stdout


# ---- host emulation of the tile algorithm (TEST TOOL; see tests/host/tile_emul.cpp)
_emul = None


def emul_run(text, mode, threads=4, ratio=0.5, mapq=10, sam=True, cfg=0, block=0):
    global _emul
    ensure_built()
    if _emul is None:
        _emul = C.CDLL(EMUL_SO)
    if isinstance(mode, str):
        mode = MODES[mode]
    op, osam = C.c_void_p(), C.c_void_p()
    npairs, nsam = C.c_size_t(), C.c_size_t()
    log = C.create_string_buffer(256)
    st = (C.c_uint64 * 4)()
    _emul.emul_run(text, C.c_size_t(len(text)), mode, C.c_float(ratio), mapq, 1 if sam else 0, threads, cfg, C.c_size_t(block),
                   C.byref(op), C.byref(npairs), C.byref(osam), C.byref(nsam), log, st)
    pairs = C.string_at(op, npairs.value)
    s = C.string_at(osam, nsam.value)
    _emul.emul_free(op)
    _emul.emul_free(osam)
    return pairs, s, log.value, {"groups": st[0], "pairs": st[1], "err": st[2], "blocks": st[3] & 0xFFFFF, "lean_tiles": (st[3] >> 20) & 0x3FFFFF,
                                 "tiles": st[3] >> 42}


class EmulShard:
    """Test stand-in for microcket_amd.Context on a machine without a GPU: same tile phases, run on the CPU."""

    class St:
        pass

    def __init__(self, mode, ratio=0.5, mapq=10, sam=True, threads=4, cfg=0):
        global _emul
        ensure_built()
        if _emul is None:
            _emul = C.CDLL(EMUL_SO)
        _emul.emul_open.restype = C.c_void_p
        _emul.emul_groups.restype = C.c_uint64
        _emul.emul_groups.argtypes = [C.c_void_p]
        _emul.emul_feed_bytes.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_size_t]
        _emul.emul_finish.argtypes = [C.c_void_p, C.c_int, C.c_uint64, C.c_uint64, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t),
                                      C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)]
        _emul.emul_close.argtypes = [C.c_void_p]
        if isinstance(mode, str):
            mode = MODES[mode]
        self.h = C.c_void_p(_emul.emul_open(mode, C.c_float(ratio), mapq, 1 if sam else 0, threads, cfg))

    def feed(self, text, block=0):
        _emul.emul_feed_bytes(self.h, text, len(text), block)

    def group_count(self):
        return int(_emul.emul_groups(self.h))

    def finish(self, drop_last=True, group_offset=0, total_groups=0):
        op, osam = C.c_void_p(), C.c_void_p()
        npairs, nsam = C.c_size_t(), C.c_size_t()
        c8 = (C.c_uint32 * 8)()
        st = (C.c_uint64 * 4)()
        _emul.emul_finish(self.h, 1 if drop_last else 0, group_offset, total_groups, C.byref(op), C.byref(npairs), C.byref(osam), C.byref(nsam), c8, st)
        r = EmulShard.St()
        r.pairs_bytes = C.string_at(op, npairs.value)
        r.sam_bytes = C.string_at(osam, nsam.value)
        _emul.emul_free(op)
        _emul.emul_free(osam)
        for k, name in enumerate(("lowMap", "manyHits", "unpaired", "selfCircle", "trans", "cis10K", "cis1K", "cis0")):
            setattr(r, name, int(c8[k]))
        r.groups, r.pairs, r.err = int(st[0]), int(st[1]), int(st[2])
        return r

    def close(self):
        if self.h:
            _emul.emul_close(self.h)
            self.h = None


# ---- sharded duplicate marking: CPU stand-ins for the exchange of microcket_amd.shard.dedup_exchange -----------------------
def _mix64_np(x):
    import numpy as np
    x = x.copy()
    with np.errstate(over="ignore"):
        x ^= x >> np.uint64(33); x *= np.uint64(0xff51afd7ed558ccd); x ^= x >> np.uint64(33); x *= np.uint64(0xc4ceb9fe1a85ec53); x ^= x >> np.uint64(33)
    return x


def qname_lane(q: bytes) -> int:
    f = q.split(b":")
    if len(f) < 4:
        return 0
    d = b""
    for ch in f[3]:
        if 48 <= ch <= 57:
            d += bytes([ch])
        else:
            break
    return min(int(d), 0xFFFF) if d else 0


def expected_dups(pairs_bytes, lanes=False):
    """The definition, over .pairs lines in input order: a pair is a duplicate when an EARLIER pair has the same
    (chr1, pos1, chr2, pos2, strand1, strand2) [and lane, with lanes=True]."""
    seen = set()
    flags = bytearray()
    for line in pairs_bytes.split(b"\n")[:-1]:
        f = line.split(b"\t")
        key = tuple(f[1:7]) + ((qname_lane(f[0]),) if lanes else ())
        flags.append(1 if key in seen else 0)
        seen.add(key)
    return bytes(flags)


class KeyEngine:
    """TEST stand-in for microcket_amd.Context in the key exchange: key records built from .pairs lines (numpy), the same
    record layout (mkt_core.h KeyRec), per-rank chromosome slots in a rank-dependent order."""
    MASK1 = 0xFFFFFFFFC000FFFF

    def __init__(self, pairs_lines, rank, lanes=False):
        import numpy as np
        names = sorted({f for l in pairs_lines for f in (l.split(b"\t")[1], l.split(b"\t")[3])}, reverse=(rank % 2 == 1))
        self.slot = {nm: (7 * k + 3 + rank) % 8192 for k, nm in enumerate(names)}
        k = np.zeros((len(pairs_lines), 3), dtype=np.uint64)
        for j, l in enumerate(pairs_lines):
            f = l.split(b"\t")
            k[j, 0] = (self.slot[f[1]] << 45) | (self.slot[f[3]] << 32) | int(f[2])
            k[j, 1] = (int(f[4]) << 32) | ((1 if f[5] == b"-" else 0) << 31) | ((1 if f[6] == b"-" else 0) << 30) | (qname_lane(f[0]) if lanes else 0)
            k[j, 2] = j
        self.keys = k

    def ext_chr_names(self):
        return {s: nm for nm, s in self.slot.items()}

    def ext_partition(self, drop_last, lut, world, torch, device):
        import numpy as np
        k = self.keys.copy()
        if k.shape[0]:
            k0 = k[:, 0]
            a = lut[((k0 >> np.uint64(45)) & np.uint64(8191)).astype(np.int64)].astype(np.uint64)
            b = lut[((k0 >> np.uint64(32)) & np.uint64(8191)).astype(np.int64)].astype(np.uint64)
            k[:, 0] = (a << np.uint64(45)) | (b << np.uint64(32)) | (k0 & np.uint64(0xFFFFFFFF))
        dest = (_mix64_np(k[:, 0] ^ _mix64_np(k[:, 1] & np.uint64(self.MASK1))) % np.uint64(world)).astype(np.int64)
        order = np.argsort(dest, kind="stable")
        self.perm = np.empty(k.shape[0], dtype=np.int64)
        self.perm[order] = np.arange(k.shape[0])
        send = torch.from_numpy(np.ascontiguousarray(k[order]).view(np.uint8).reshape(-1).copy())
        return send, [int((dest == r).sum()) for r in range(world)]

    def ext_dedup_tensor(self, recv, torch):
        import numpy as np
        k = recv.numpy().view(np.uint64).reshape(-1, 3)
        seen = set()
        flags = np.zeros(k.shape[0], dtype=np.uint8)
        for j in range(k.shape[0]):
            key = (int(k[j, 0]), int(k[j, 1]) & self.MASK1)
            if key in seen:
                flags[j] = 1
            seen.add(key)
        return torch.from_numpy(flags), int(flags.sum())

    def ext_unpartition(self, back, want_flags=True):
        f = back.numpy()[self.perm] if back.numel() else back.numpy()
        return bytes(f.tolist()), int(f.sum())


class FakeDist:
    """In-process stand-in for torch.distributed (one THREAD per rank): the collectives dedup_exchange uses, on torch tensors of any
    device.  Lets two contexts on ONE GPU run the real device-side exchange in the GPU tests."""

    def __init__(self, world):
        import threading
        self.world = world
        self.bar = threading.Barrier(world)
        self.slots = [None] * world

    def rank(self, r):
        return _FakeRank(self, r)


class _FakeRank:
    def __init__(self, fd, r):
        self.fd, self.r = fd, r

    def _swap(self, x):
        self.fd.slots[self.r] = x
        self.fd.bar.wait()
        got = list(self.fd.slots)
        self.fd.bar.wait()
        return got

    def all_gather_object(self, out, x):
        out[:] = self._swap(x)

    def all_to_all_single(self, out, inp, output_split_sizes=None, input_split_sizes=None):
        w = self.fd.world
        if input_split_sizes is None:
            input_split_sizes = [inp.numel() // w] * w
        got = self._swap((inp, list(input_split_sizes)))
        pos = 0
        for src in range(w):
            t, sp = got[src]
            off = sum(sp[:self.r])
            n = sp[self.r]
            out[pos:pos + n] = t[off:off + n]
            pos += n
        self.fd.bar.wait()

    def all_reduce(self, t):
        got = self._swap(t.clone())
        tot = got[0].clone()
        for x in got[1:]:
            tot += x
        t.copy_(tot)
        self.fd.bar.wait()


# ---- krmdup: FASTQ generator, oracle binding, reference runner ----------------------------------------
KRMDUP_REF = os.path.join(ROOT, "oracle", "_ref", "krmdup.ref")
KRMDUP_PIPE_REF = os.path.join(ROOT, "oracle", "_ref", "krmdup.pipe.ref")


def synth_fastq(seed, pairs, read_len=100, dup_rate=0.3, n_rate=0.02, short_rate=0.02, lower_rate=0.01) -> bytes:
    """Seeded interleaved paired-end FASTQ (8 lines per pair): PCR-duplicate-like repeats of earlier fragments (same leading
    bases, different read names and tails), some N bases inside and outside the key window, reads shorter than the key
    window, a few lower-case bases (the reference folds case in the key but not in the bucket letter)."""
    import random
    rnd = random.Random(seed)
    frags = []
    out = []
    for i in range(pairs):
        if frags and rnd.random() < dup_rate:
            a, b = frags[rnd.randrange(len(frags))]
            a = a[:40] + "".join(rnd.choice("ACGT") for _ in range(len(a) - 40)) if len(a) > 40 else a      # same key window, other tail
        else:
            la = read_len if rnd.random() > short_rate else rnd.randrange(3, 24)
            lb = read_len if rnd.random() > short_rate else rnd.randrange(3, 24)
            a = "".join(rnd.choice("ACGT") for _ in range(la))
            b = "".join(rnd.choice("ACGT") for _ in range(lb))
            if rnd.random() < n_rate:
                k = rnd.randrange(len(a)); a = a[:k] + "N" + a[k + 1:]
            if rnd.random() < n_rate:
                k = rnd.randrange(min(len(b), 30)); b = b[:k] + "N" + b[k + 1:]
            if rnd.random() < lower_rate:
                k = rnd.randrange(min(len(a), 22)); a = a[:k] + a[k].lower() + a[k + 1:]
            frags.append((a, b))
        out.append(f"@R{seed}.{i} 1:N:0\n{a}\n+{'' if i % 3 else 'R' + str(i)}\n{'F' * len(a)}\n@R{seed}.{i} 2:N:0\n{b}\n+\n{'#' * len(b)}\n")
    return "".join(out).encode()


class KStats(C.Structure):
    _fields_ = [(k, C.c_uint64) for k in ("total", "uniq", "dup", "discard")]


class _KP(C.Structure):
    _fields_ = [(k, C.c_uint32) for k in ("hskip1", "keylen1", "hskip2", "keylen2")]


def krmdup_oracle(text: bytes, hskip1=5, keylen1=16, hskip2=5, keylen2=16):
    """The CPU restatement (oracle/krmdup_oracle.c): (read1 bytes, read2 bytes, log text)."""
    global _oracle
    ensure_built()
    if _oracle is None:
        _oracle = C.CDLL(ORACLE_SO)
    a, b, st = _OB(), _OB(), KStats()
    kp = _KP(hskip1, keylen1, hskip2, keylen2)
    rc = _oracle.krm_run(text, C.c_size_t(len(text)), C.byref(kp), C.byref(a), C.byref(b), C.byref(st))
    assert rc == 0
    r1 = C.string_at(a.p, a.n) if a.n else b""
    r2 = C.string_at(b.p, b.n) if b.n else b""
    _oracle.krm_buf_free(C.byref(a))
    _oracle.krm_buf_free(C.byref(b))
    return r1, r2, krmdup_log(st.total, st.uniq, st.dup, st.discard)


def krmdup_log(total, uniq, dup, discard) -> bytes:
    return f"Total\t{total}\nUniq\t{uniq}\nDup\t{dup}\nDiscard\t{discard}\n".encode()


def krmdup_run_cli(exe, text, pipe=False, args=()):
    """Runs a krmdup executable (the reference build or the GPU one): (rc, read1, read2 | stdout, log, stderr)."""
    with tempfile.TemporaryDirectory(prefix="krm_") as d:
        inp = os.path.join(d, "in.fq")
        with open(inp, "wb") as f:
            f.write(text)
        pre = os.path.join(d, "o")
        p = subprocess.run([exe, "-i", inp, "-o", pre, *args], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        rd = lambda fn: open(fn, "rb").read() if os.path.exists(fn) else b""
        if pipe:
            return p.returncode, p.stdout, b"", rd(pre + ".log"), p.stderr
        return p.returncode, rd(pre + ".read1.fq"), rd(pre + ".read2.fq"), rd(pre + ".log"), p.stderr


def fastq_records(b: bytes):
    """sorted list of 4-line records (canonical form of krmdup.pipe's stdout, whose bucket order is schedule dependent)"""
    L = b.split(b"\n")
    if L and L[-1] == b"":
        L.pop()
    return sorted(b"\n".join(L[i:i + 8]) for i in range(0, len(L), 8))
