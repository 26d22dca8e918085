"""ctypes binding of include/mkt.h (libmkt_hip.so).  Plumbing only: no record is ever touched here."""
import ctypes as C
import os
import sys
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
MODE_FLASH, MODE_UNC = 0, 1
TILES_AUTO, TILES_FAST, TILES_SMALL = 0, 1, 2
EXT_KEYS = 1
EXT_LANES = 2
KEY_BYTES = 24
EXPORTS = [
    "mkt_abi_version", "mkt_strerror", "mkt_last_error", "mkt_device_count", "mkt_create", "mkt_destroy",
    "mkt_input_window", "mkt_submit_window",
    "mkt_submit", "mkt_drain", "mkt_drain_wait", "mkt_submit_device", "mkt_sync", "mkt_fetch_last_block", "mkt_finish",
    "mkt_format_log", "mkt_get_timing", "mkt_reset_timing", "mkt_synth_device", "mkt_copy_to_host", "mkt_device_text",
    "mkt_reset", "mkt_ext_dedup", "mkt_ext_chrstat", "mkt_ext_chr_names", "mkt_ext_keys_fetch", "mkt_ext_dedup_keys", "mkt_ext_keys_device", "mkt_ext_partition", "mkt_ext_dedup_device", "mkt_ext_unpartition", "mkt_ext_dedup_multi", "mkt_dataset_create", "mkt_dataset_info", "mkt_dataset_block", "mkt_dataset_destroy", "mkt_group_count",
    "mkt_sorter_create", "mkt_sorter_destroy", "mkt_sorter_error", "mkt_sorter_add", "mkt_sorter_add_device", "mkt_sorter_sort", "mkt_sorter_fetch",
    "mkt_rmdup_create", "mkt_rmdup_destroy", "mkt_rmdup_error", "mkt_rmdup_reserve", "mkt_rmdup_add", "mkt_rmdup_run", "mkt_rmdup_fetch",
    "mkt_rmdup_begin", "mkt_rmdup_push", "mkt_rmdup_stats",
    "mkt_bam_create", "mkt_bam_destroy", "mkt_bam_error", "mkt_bam_note", "mkt_bam_add", "mkt_bam_add_device", "mkt_bam_run", "mkt_bam_fetch",
    "mkt_bam_reserve", "mkt_bam_window", "mkt_bam_commit", "mkt_bam_read",
]


class MktError(RuntimeError):
    pass


class Params(C.Structure):
    _fields_ = [("mode", C.c_int32), ("min_mapped_ratio", C.c_float), ("min_mapq", C.c_int32), ("write_sam", C.c_int32),
                ("ref_threads", C.c_int32), ("device", C.c_int32), ("block_bytes", C.c_uint64), ("tiles", C.c_int32),
                ("ordered", C.c_int32), ("extensions", C.c_uint32), ("reserved2", C.c_uint32)]


class Stats(C.Structure):
    _fields_ = [(k, C.c_uint32) for k in ("lowMap", "manyHits", "unpaired", "selfCircle", "trans", "cis10K", "cis1K", "cis0",
                                           "selfCircle_all", "reserved")] + \
               [(k, C.c_uint64) for k in ("groups", "pairs", "pair_bytes", "sam_bytes", "lines_in", "bytes_in", "blocks")]

    def counters(self):
        return {k: getattr(self, k) for k in ("lowMap", "manyHits", "unpaired", "selfCircle", "trans", "cis10K", "cis1K", "cis0")}


class Out(C.Structure):
    _fields_ = [("pairs", C.c_void_p), ("pairs_len", C.c_size_t), ("sam", C.c_void_p), ("sam_len", C.c_size_t)]


class Timing(C.Structure):
    _fields_ = [("tile_kernel_ms", C.c_double), ("tile_launches", C.c_uint64), ("tile_bytes", C.c_uint64), ("other_ms", C.c_double),
                ("tiles", C.c_uint64), ("deferred_tiles", C.c_uint64)]


def lib_path():
    # MKT_LIB selects a diagnostic build (e.g. the phase-stamp build); never needed in production
    return os.environ.get("MKT_LIB") or os.path.join(HERE, "libmkt_hip.so")


def exe_path():
    return os.path.join(HERE, "bin", "sam2pairs")


_lib = None


def hip_runtimes():
    """The HIP runtime libraries mapped into this process (paths of libamdhip64.so*)."""
    found = set()
    try:
        with open("/proc/self/maps") as f:
            for line in f:
                if "libamdhip64" in line:
                    found.add(os.path.realpath(line.split()[-1]))
    except OSError:
        pass
    return sorted(found)


def check_single_hip_runtime():
    """One HIP runtime per process.  PyTorch bundles its own libamdhip64.so and loads it by file name; libmkt_hip.so asks for the
    soname.  torch imported FIRST: its copy satisfies the library, one runtime.  The library loaded first: it binds /opt/rocm's
    copy, torch later maps its own as a SECOND runtime, and one of the two then sees no GPU ("No HIP GPUs are available",
    hipErrorNoDevice from mkt_create).  Say so instead."""
    rts = hip_runtimes()
    if len(rts) > 1:
        raise MktError("two HIP runtimes are mapped into this process (" + ", ".join(rts) + "): import torch BEFORE microcket_amd "
                       "(or not at all), so that libmkt_hip.so shares torch's runtime; with two, one of them sees no GPU")


def load_library():
    """Loads libmkt_hip.so (raises if it has not been built: there is no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise MktError(f"{path} is missing: run `python -m microcket_amd.build` (hipcc, gfx950). No CPU path exists.")
    L = C.CDLL(path)
    if "torch" in sys.modules:
        check_single_hip_runtime()           # torch came first and the library still brought a second runtime: say so now
    L.mkt_strerror.restype = C.c_char_p
    L.mkt_last_error.restype = C.c_char_p
    L.mkt_last_error.argtypes = [C.c_void_p]
    L.mkt_create.argtypes = [C.POINTER(Params), C.POINTER(C.c_void_p)]
    L.mkt_destroy.argtypes = [C.c_void_p]
    L.mkt_destroy.restype = None
    L.mkt_submit.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_int]
    L.mkt_drain.argtypes = [C.c_void_p, C.POINTER(Out)]
    L.mkt_drain_wait.argtypes = [C.c_void_p, C.POINTER(Out), C.POINTER(C.c_int)]
    L.mkt_input_window.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
    L.mkt_submit_window.argtypes = [C.c_void_p, C.c_size_t, C.c_int]
    L.mkt_submit_device.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.mkt_sync.argtypes = [C.c_void_p]
    L.mkt_fetch_last_block.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.mkt_finish.argtypes = [C.c_void_p, C.c_int, C.c_uint64, C.c_uint64, C.POINTER(Stats)]
    L.mkt_format_log.argtypes = [C.POINTER(Stats), C.c_char_p, C.c_size_t]
    L.mkt_get_timing.argtypes = [C.c_void_p, C.POINTER(Timing)]
    L.mkt_reset_timing.argtypes = [C.c_void_p]
    L.mkt_synth_device.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_uint64, C.c_int,
                                   C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
    L.mkt_copy_to_host.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    L.mkt_device_text.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.POINTER(C.c_void_p)]
    L.mkt_dataset_create.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int,
                                     C.POINTER(C.c_void_p)]
    L.mkt_dataset_info.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.mkt_dataset_block.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_uint64)]
    L.mkt_dataset_destroy.argtypes = [C.c_void_p]
    L.mkt_dataset_destroy.restype = None
    L.mkt_reset.argtypes = [C.c_void_p]
    L.mkt_ext_dedup.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_void_p, C.c_size_t]
    L.mkt_ext_chrstat.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.mkt_ext_chr_names.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.mkt_ext_keys_fetch.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_uint64)]
    L.mkt_ext_dedup_keys.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.POINTER(C.c_uint64)]
    L.mkt_group_count.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    L.mkt_ext_keys_device.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
    L.mkt_ext_partition.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_uint32, C.c_void_p, C.POINTER(C.c_uint64)]
    L.mkt_ext_dedup_device.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.POINTER(C.c_uint64)]
    L.mkt_ext_unpartition.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_uint64)]
    L.mkt_sorter_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
    L.mkt_sorter_destroy.argtypes = [C.c_void_p]
    L.mkt_sorter_destroy.restype = None
    L.mkt_sorter_error.argtypes = [C.c_void_p]
    L.mkt_sorter_error.restype = C.c_char_p
    L.mkt_sorter_add.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
    L.mkt_sorter_add_device.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.mkt_sorter_sort.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.mkt_sorter_fetch.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_size_t]
    L.mkt_rmdup_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
    L.mkt_rmdup_destroy.argtypes = [C.c_void_p]
    L.mkt_rmdup_destroy.restype = None
    L.mkt_rmdup_error.argtypes = [C.c_void_p]
    L.mkt_rmdup_error.restype = C.c_char_p
    L.mkt_rmdup_add.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
    L.mkt_rmdup_run.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.mkt_rmdup_fetch.argtypes = [C.c_void_p, C.c_int, C.c_uint64, C.c_void_p, C.c_size_t]
    L.mkt_rmdup_begin.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int]
    L.mkt_rmdup_push.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_int, C.POINTER(C.c_uint64)]
    L.mkt_rmdup_stats.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    L.mkt_bam_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
    L.mkt_bam_destroy.argtypes = [C.c_void_p]
    L.mkt_bam_destroy.restype = None
    L.mkt_bam_error.argtypes = [C.c_void_p]
    L.mkt_bam_error.restype = C.c_char_p
    L.mkt_bam_note.argtypes = [C.c_void_p]
    L.mkt_bam_note.restype = C.c_char_p
    L.mkt_bam_add.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
    L.mkt_bam_add_device.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.mkt_bam_run.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.mkt_bam_fetch.argtypes = [C.c_void_p, C.c_int, C.c_uint64, C.c_void_p, C.c_size_t]
    L.mkt_bam_reserve.argtypes = [C.c_void_p, C.c_size_t]
    L.mkt_bam_window.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
    L.mkt_bam_commit.argtypes = [C.c_void_p, C.c_size_t]
    L.mkt_bam_read.argtypes = [C.c_void_p, C.c_int, C.c_uint64, C.c_size_t, C.POINTER(C.c_void_p)]
    _lib = L
    return L


def device_count():
    return int(load_library().mkt_device_count())


class Context:
    """One GPU context = one input stream (mirrors one bin/sam2pairs process of the reference)."""

    def __init__(self, mode, ratio=0.5, min_mapq=10, write_sam=True, ref_threads=4, device=0, block_bytes=0, tiles=TILES_AUTO,
                 ordered=False, extensions=0):
        self.L = load_library()
        if isinstance(mode, str):
            mode = {"flash": MODE_FLASH, "unc": MODE_UNC}[mode]
        self.params = Params(mode, ratio, min_mapq, 1 if write_sam else 0, ref_threads, device, block_bytes, tiles, 1 if ordered else 0, extensions, 0)
        self.h = C.c_void_p()
        rc = self.L.mkt_create(C.byref(self.params), C.byref(self.h))
        if rc != 0:
            hint = ""
            if len(hip_runtimes()) > 1:      # (this package loaded before torch: the usual reason for "no device" on a GPU box)
                hint = " [two HIP runtimes are mapped into this process (" + ", ".join(hip_runtimes()) + "): import torch BEFORE microcket_amd]"
            raise MktError(f"mkt_create: {self.L.mkt_strerror(rc).decode()}: {self.L.mkt_last_error(None).decode()}{hint}")

    def _chk(self, rc, what):
        if rc != 0:
            raise MktError(f"{what}: {self.L.mkt_strerror(rc).decode()}: {self.L.mkt_last_error(self.h).decode()}")

    def close(self):
        if self.h:
            self.L.mkt_destroy(self.h)
            self.h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # ---- streaming path
    def submit(self, data: bytes, last=False):
        self._chk(self.L.mkt_submit(self.h, data, len(data), 1 if last else 0), "mkt_submit")

    def drain(self):
        o = Out()
        self._chk(self.L.mkt_drain(self.h, C.byref(o)), "mkt_drain")
        pairs = C.string_at(o.pairs, o.pairs_len) if o.pairs_len else b""
        sam = C.string_at(o.sam, o.sam_len) if o.sam_len else b""
        return pairs, sam

    def finish(self, drop_last=True, group_offset=0, total_groups=0):
        st = Stats()
        self._chk(self.L.mkt_finish(self.h, 1 if drop_last else 0, group_offset, total_groups, C.byref(st)), "mkt_finish")
        return st

    def format_log(self, st):
        buf = C.create_string_buffer(512)
        self.L.mkt_format_log(C.byref(st), buf, 512)
        return buf.value

    def run_bytes(self, text: bytes, chunk=0):
        """Whole input -> (pairs, sam, stats, log)."""
        pairs, sam = [], []
        if chunk <= 0:
            chunk = max(len(text), 1)
        pos = 0
        while True:
            part = text[pos:pos + chunk]
            pos += len(part)
            last = pos >= len(text)
            self.submit(part, last)
            a, b = self.drain()
            pairs.append(a)
            sam.append(b)
            if last:
                break
        st = self.finish(True)
        a, b = self.drain()
        pairs.append(a)
        sam.append(b)
        return b"".join(pairs), b"".join(sam), st, self.format_log(st)

    def run_bytes_window(self, text: bytes, piece=0):
        """The same through the zero-copy input window (mkt_input_window / mkt_submit_window), `piece` bytes per commit."""
        pairs, sam = [], []
        pos = 0
        while True:
            buf, cap = C.c_void_p(), C.c_size_t()
            self._chk(self.L.mkt_input_window(self.h, C.byref(buf), C.byref(cap)), "mkt_input_window")
            take = min(cap.value, len(text) - pos, piece if piece > 0 else cap.value)
            if take:
                C.memmove(buf.value, text[pos:pos + take], take)
            pos += take
            last = pos >= len(text)
            self._chk(self.L.mkt_submit_window(self.h, C.c_size_t(take), 1 if last else 0), "mkt_submit_window")
            a, b = self.drain()
            pairs.append(a)
            sam.append(b)
            if last:
                break
        st = self.finish(True)
        a, b = self.drain()
        pairs.append(a)
        sam.append(b)
        return b"".join(pairs), b"".join(sam), st, self.format_log(st)

    # ---- resident path
    def synth_device(self, seed, profile, n_groups, first_group=0, genome=0, read_len=150, lanes=1, tail_group=False):
        p = C.c_void_p()
        n = C.c_size_t()
        self._chk(self.L.mkt_synth_device(self.h, seed, profile, genome, read_len, lanes, first_group, n_groups, 1 if tail_group else 0,
                                          C.byref(p), C.byref(n)), "mkt_synth_device")
        return p.value, n.value

    def submit_device(self, d_ptr, n):
        self._chk(self.L.mkt_submit_device(self.h, C.c_void_p(d_ptr), n), "mkt_submit_device")

    def sync(self):
        self._chk(self.L.mkt_sync(self.h), "mkt_sync")

    def fetch_last_block(self):
        np_, ns_ = C.c_size_t(), C.c_size_t()
        self._chk(self.L.mkt_fetch_last_block(self.h, None, 0, C.byref(np_), None, 0, C.byref(ns_)), "mkt_fetch_last_block")
        pb = C.create_string_buffer(max(np_.value, 1))
        sb = C.create_string_buffer(max(ns_.value, 1))
        self._chk(self.L.mkt_fetch_last_block(self.h, pb, np_.value, C.byref(np_), sb, ns_.value, C.byref(ns_)), "mkt_fetch_last_block")
        return pb.raw[:np_.value], sb.raw[:ns_.value]

    def fetch_last_block_np(self):
        """the same as numpy uint8 arrays (no extra copies: the blocks of the bench are 100 MB of .pairs)"""
        import numpy as np
        np_, ns_ = C.c_size_t(), C.c_size_t()
        self._chk(self.L.mkt_fetch_last_block(self.h, None, 0, C.byref(np_), None, 0, C.byref(ns_)), "mkt_fetch_last_block")
        pb = np.empty(max(np_.value, 1), dtype=np.uint8)
        sb = np.empty(max(ns_.value, 1), dtype=np.uint8)
        self._chk(self.L.mkt_fetch_last_block(self.h, pb.ctypes.data_as(C.c_void_p), np_.value, C.byref(np_), sb.ctypes.data_as(C.c_void_p), ns_.value, C.byref(ns_)),
                  "mkt_fetch_last_block")
        return pb[:np_.value], sb[:ns_.value]

    def copy_to_host_np(self, d_ptr, n):
        import numpy as np
        buf = np.empty(max(n, 1), dtype=np.uint8)
        self._chk(self.L.mkt_copy_to_host(self.h, C.c_void_p(d_ptr), buf.ctypes.data_as(C.c_void_p), n), "mkt_copy_to_host")
        return buf[:n]

    def copy_to_host(self, d_ptr, n):
        buf = C.create_string_buffer(max(n, 1))
        self._chk(self.L.mkt_copy_to_host(self.h, C.c_void_p(d_ptr), buf, n), "mkt_copy_to_host")
        return buf.raw[:n]

    def device_text(self, data: bytes):
        """host text -> device buffer owned by the context; returns the device pointer (a block for submit_device)"""
        p = C.c_void_p()
        self._chk(self.L.mkt_device_text(self.h, data, len(data), C.byref(p)), "mkt_device_text")
        return p.value

    def reset(self):
        self._chk(self.L.mkt_reset(self.h), "mkt_reset")

    def ext_dedup(self, drop_last=True, want_flags=True):
        """(total reported pairs, duplicates, flags bytes in input order)"""
        tot, dup = C.c_uint64(), C.c_uint64()
        self._chk(self.L.mkt_ext_dedup(self.h, 1 if drop_last else 0, C.byref(tot), C.byref(dup), None, 0), "mkt_ext_dedup")
        if not want_flags or tot.value == 0:
            return tot.value, dup.value, b""
        buf = C.create_string_buffer(tot.value)
        self._chk(self.L.mkt_ext_dedup(self.h, 1 if drop_last else 0, C.byref(tot), C.byref(dup), buf, tot.value), "mkt_ext_dedup")
        return tot.value, dup.value, buf.raw[:tot.value]

    def ext_chr_names(self):
        """{slot: name} of this context's chromosome-name table"""
        n = C.c_size_t()
        self._chk(self.L.mkt_ext_chr_names(self.h, None, 0, C.byref(n)), "mkt_ext_chr_names")
        buf = C.create_string_buffer(max(n.value, 1))
        self._chk(self.L.mkt_ext_chr_names(self.h, buf, n.value, C.byref(n)), "mkt_ext_chr_names")
        out = {}
        for line in buf.raw[:n.value].split(b"\n"):
            if line:
                s, name = line.split(b"\t", 1)
                out[int(s)] = name
        return out

    def ext_keys_fetch(self, drop_last=True):
        """key records as a numpy uint64 array of shape (n, 3): k0, k1, ordinal (input order)"""
        import numpy as np
        n = C.c_uint64()
        self._chk(self.L.mkt_ext_keys_fetch(self.h, 1 if drop_last else 0, None, 0, C.byref(n)), "mkt_ext_keys_fetch")
        arr = np.zeros((n.value, 3), dtype=np.uint64)
        if n.value:
            self._chk(self.L.mkt_ext_keys_fetch(self.h, 1 if drop_last else 0, arr.ctypes.data_as(C.c_void_p), arr.nbytes, C.byref(n)), "mkt_ext_keys_fetch")
        return arr

    def ext_dedup_keys(self, keys):
        """duplicate flags (numpy uint8) for a (n, 3) uint64 key array in input order; runs on this context's GPU"""
        import numpy as np
        keys = np.ascontiguousarray(keys, dtype=np.uint64)
        n = keys.shape[0]
        flags = np.zeros(n, dtype=np.uint8)
        dups = C.c_uint64()
        if n:
            self._chk(self.L.mkt_ext_dedup_keys(self.h, keys.ctypes.data_as(C.c_void_p), n, flags.ctypes.data_as(C.c_void_p), C.byref(dups)), "mkt_ext_dedup_keys")
        return flags, dups.value

    # ---- sharded duplicate marking: the device side of microcket_amd.shard.dedup_exchange (torch tensors carry the buffers)
    def ext_key_count(self, drop_last=True):
        p, n = C.c_void_p(), C.c_uint64()
        self._chk(self.L.mkt_ext_keys_device(self.h, 1 if drop_last else 0, C.byref(p), C.byref(n)), "mkt_ext_keys_device")
        return n.value

    def ext_partition(self, drop_last, lut, world, torch, device):
        """Key records grouped by destination rank (stable) in a uint8 device tensor of 24 n bytes; returns (tensor, counts)."""
        import numpy as np
        n = self.ext_key_count(drop_last)
        send = torch.empty(max(n, 1) * KEY_BYTES, dtype=torch.uint8, device=device)
        counts = (C.c_uint64 * 16)()
        lut = None if lut is None else np.ascontiguousarray(lut, dtype=np.uint16)
        assert lut is None or lut.shape[0] == 8192
        self._chk(self.L.mkt_ext_partition(self.h, 1 if drop_last else 0, None if lut is None else lut.ctypes.data_as(C.c_void_p), world,
                                           C.c_void_p(send.data_ptr()), counts), "mkt_ext_partition")
        return send[:n * KEY_BYTES], [int(counts[r]) for r in range(world)]

    def ext_dedup_tensor(self, recv, torch):
        """Duplicate flags (uint8 device tensor) for the key records lying in the uint8 device tensor recv, first in buffer order wins."""
        n = recv.numel() // KEY_BYTES
        flags = torch.zeros(max(n, 1), dtype=torch.uint8, device=recv.device)
        dups = C.c_uint64()
        if n:
            self._chk(self.L.mkt_ext_dedup_device(self.h, C.c_void_p(recv.data_ptr()), n, C.c_void_p(flags.data_ptr()), C.byref(dups)), "mkt_ext_dedup_device")
        return flags[:n], dups.value

    def ext_unpartition(self, flags_part, want_flags=True):
        """Flags that came back in ext_partition's order -> input order (bytes) and the number of duplicates among this context's pairs."""
        n = flags_part.numel()
        dups = C.c_uint64()
        buf = C.create_string_buffer(max(n, 1)) if want_flags else None
        self._chk(self.L.mkt_ext_unpartition(self.h, C.c_void_p(flags_part.data_ptr()) if n else None, buf, n if want_flags else 0, C.byref(dups)), "mkt_ext_unpartition")
        return (buf.raw[:n] if want_flags else b""), dups.value

    def ext_chrstat(self, drop_last=True):
        n = C.c_size_t()
        cap = 1 << 20                          # one call in the usual case (the table is a few KB)
        buf = C.create_string_buffer(cap)
        rc = self.L.mkt_ext_chrstat(self.h, 1 if drop_last else 0, buf, cap, C.byref(n))
        if rc != 0 and n.value > cap:          # too small: *len holds the size needed
            buf = C.create_string_buffer(n.value)
            rc = self.L.mkt_ext_chrstat(self.h, 1 if drop_last else 0, buf, n.value, C.byref(n))
        self._chk(rc, "mkt_ext_chrstat")
        return buf.raw[:n.value]

    def group_count(self):
        g = C.c_uint64()
        self._chk(self.L.mkt_group_count(self.h, C.byref(g)), "mkt_group_count")
        return g.value

    def dataset(self, seed, profile, n_groups, groups_per_block, first_group=0, genome=0, read_len=150, lanes=1, tail_group=False):
        return Dataset(self, seed, profile, n_groups, groups_per_block, first_group, genome, read_len, lanes, tail_group)

    def timing(self):
        t = Timing()
        self._chk(self.L.mkt_get_timing(self.h, C.byref(t)), "mkt_get_timing")
        return t

    def reset_timing(self):
        self._chk(self.L.mkt_reset_timing(self.h), "mkt_reset_timing")


class Dataset:
    """Synthetic SAM resident in HBM, cut into group-aligned blocks (see mkt_dataset_create)."""

    def __init__(self, ctx, seed, profile, n_groups, groups_per_block, first_group, genome, read_len, lanes, tail_group):
        self.ctx = ctx
        self.h = C.c_void_p()
        ctx._chk(ctx.L.mkt_dataset_create(ctx.h, seed, profile, genome, read_len, lanes, first_group, n_groups, groups_per_block,
                                          1 if tail_group else 0, C.byref(self.h)), "mkt_dataset_create")
        nb, tb, tg = C.c_uint64(), C.c_uint64(), C.c_uint64()
        ctx.L.mkt_dataset_info(self.h, C.byref(nb), C.byref(tb), C.byref(tg))
        self.n_blocks, self.total_bytes, self.total_groups = nb.value, tb.value, tg.value
        self.blocks = []
        for i in range(self.n_blocks):
            p, n, g = C.c_void_p(), C.c_size_t(), C.c_uint64()
            ctx.L.mkt_dataset_block(self.h, i, C.byref(p), C.byref(n), C.byref(g))
            self.blocks.append((p.value, n.value, g.value))

    def close(self):
        if self.h:
            self.ctx.L.mkt_dataset_destroy(self.h)
            self.h = C.c_void_p()


class PairsSorter:
    """.pairs text in the driver's order (LANG=C sort -k2,2d -k4,4d -k3,3n -k5,5n) on the GPU: see mkt_sorter_* in include/mkt.h."""

    def __init__(self, device=0):
        self.L = load_library()
        self.h = C.c_void_p()
        rc = self.L.mkt_sorter_create(device, C.byref(self.h))
        if rc != 0:
            raise MktError(f"mkt_sorter_create: {self.L.mkt_strerror(rc).decode()}")

    def _chk(self, rc, what):
        if rc != 0:
            raise MktError(f"{what}: {self.L.mkt_strerror(rc).decode()}: {self.L.mkt_sorter_error(self.h).decode()}")

    def add(self, data: bytes):
        self._chk(self.L.mkt_sorter_add(self.h, data, len(data)), "mkt_sorter_add")

    def add_device(self, d_ptr, n):
        self._chk(self.L.mkt_sorter_add_device(self.h, C.c_void_p(d_ptr), n), "mkt_sorter_add_device")

    def sort(self):
        """Sorts what was added; returns the sorted text."""
        lines, nbytes = C.c_uint64(), C.c_uint64()
        self._chk(self.L.mkt_sorter_sort(self.h, C.byref(lines), C.byref(nbytes)), "mkt_sorter_sort")
        self.lines = lines.value
        buf = C.create_string_buffer(max(nbytes.value, 1))
        self._chk(self.L.mkt_sorter_fetch(self.h, 0, buf, nbytes.value), "mkt_sorter_fetch")
        return buf.raw[:nbytes.value]

    def close(self):
        if self.h:
            self.L.mkt_sorter_destroy(self.h)
            self.h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


def rmdup(text: bytes, hskip1=5, keylen1=16, hskip2=5, keylen2=16, interleaved=False, device=0, piece=1 << 24, stream=False):
    """The reference's krmdup on the GPU (mkt_rmdup_*): returns (read1 | interleaved bytes, read2 bytes, (total, uniq, dup, discard)).
    stream=False: everything added, then ONE run (the resident form).  stream=True: begin / push piece by piece / push(final), the
    outputs of every segment taken as they come (MKT_RMDUP_SEGMENT_MB sets the segment size; what bin/krmdup does)."""
    L = load_library()
    h = C.c_void_p()
    rc = L.mkt_rmdup_create(device, C.byref(h))
    if rc != 0:
        raise MktError(f"mkt_rmdup_create: {L.mkt_strerror(rc).decode()}")

    def fetch(ob, outs):
        for which in (0, 1):
            if ob[which]:
                buf = C.create_string_buffer(ob[which])
                rc = L.mkt_rmdup_fetch(h, which, 0, buf, ob[which])
                if rc != 0:
                    raise MktError(f"mkt_rmdup_fetch: {L.mkt_strerror(rc).decode()}: {L.mkt_rmdup_error(h).decode()}")
                outs[which].append(buf.raw[:ob[which]])

    try:
        st = (C.c_uint64 * 4)()
        ob = (C.c_uint64 * 2)()
        outs = ([], [])
        if stream:
            rc = L.mkt_rmdup_begin(h, hskip1, keylen1, hskip2, keylen2, 1 if interleaved else 0)
            if rc != 0:
                raise MktError(f"mkt_rmdup_begin: {L.mkt_strerror(rc).decode()}: {L.mkt_rmdup_error(h).decode()}")
            for k in list(range(0, len(text), piece)) + [None]:
                part = b"" if k is None else text[k:k + piece]
                rc = L.mkt_rmdup_push(h, part, len(part), 1 if k is None else 0, ob)
                if rc != 0:
                    raise MktError(f"mkt_rmdup_push: {L.mkt_strerror(rc).decode()}: {L.mkt_rmdup_error(h).decode()}")
                fetch(ob, outs)
            L.mkt_rmdup_stats(h, st)
        else:
            for k in range(0, len(text), piece):
                part = text[k:k + piece]
                rc = L.mkt_rmdup_add(h, part, len(part))
                if rc != 0:
                    raise MktError(f"mkt_rmdup_add: {L.mkt_strerror(rc).decode()}: {L.mkt_rmdup_error(h).decode()}")
            rc = L.mkt_rmdup_run(h, hskip1, keylen1, hskip2, keylen2, 1 if interleaved else 0, st, ob)
            if rc != 0:
                raise MktError(f"mkt_rmdup_run: {L.mkt_strerror(rc).decode()}: {L.mkt_rmdup_error(h).decode()}")
            fetch(ob, outs)
        return b"".join(outs[0]), b"".join(outs[1]), tuple(int(x) for x in st)
    finally:
        L.mkt_rmdup_destroy(h)


def run_sam2pairs(in_sam, mode, prefix, threads=4, ratio=0.5, mapq=10, sam="yes", env=None, exe=None):
    """Runs the drop-in executable with the reference's argv (microcket:479,483,501,505).  Returns (rc, stdout, stderr)."""
    exe = exe or exe_path()
    if not os.path.exists(exe):
        raise MktError(f"{exe} is missing: run `python -m microcket_amd.build`")
    e = dict(os.environ)
    if env:
        e.update(env)
    p = subprocess.run([exe, in_sam, mode, prefix, str(threads), str(ratio), str(mapq), sam], stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, env=e)
    return p.returncode, p.stdout, p.stderr


def sam_to_bam(sam: bytes, sorted=True, level=2, device=0, piece=1 << 24, notes=None):
    """SAM text (header lines + alignment lines) -> (BAM bytes, BAI bytes or b"", records) on the GPU: mkt_bam_* in include/mkt.h.
    notes: a list that receives mkt_bam_note() (why no index was made), if given."""
    L = load_library()
    h = C.c_void_p()
    rc = L.mkt_bam_create(device, C.byref(h))
    if rc != 0:
        raise MktError(f"mkt_bam_create: {L.mkt_strerror(rc).decode()}")
    try:
        for k in range(0, len(sam), piece):
            part = sam[k:k + piece]
            rc = L.mkt_bam_add(h, part, len(part))
            if rc != 0:
                raise MktError(f"mkt_bam_add: {L.mkt_strerror(rc).decode()}: {L.mkt_bam_error(h).decode()}")
        nrec, nbam, nbai = C.c_uint64(), C.c_uint64(), C.c_uint64()
        rc = L.mkt_bam_run(h, 1 if sorted else 0, level, C.byref(nrec), C.byref(nbam), C.byref(nbai))
        if rc != 0:
            raise MktError(f"mkt_bam_run: {L.mkt_strerror(rc).decode()}: {L.mkt_bam_error(h).decode()}")
        outs = []
        for which, n in ((0, nbam.value), (1, nbai.value)):
            buf = C.create_string_buffer(max(n, 1))
            rc = L.mkt_bam_fetch(h, which, 0, buf, n)
            if rc != 0:
                raise MktError(f"mkt_bam_fetch: {L.mkt_strerror(rc).decode()}: {L.mkt_bam_error(h).decode()}")
            outs.append(buf.raw[:n])
        if notes is not None:
            notes.append(L.mkt_bam_note(h).decode())
        return outs[0], outs[1], nrec.value
    finally:
        L.mkt_bam_destroy(h)
