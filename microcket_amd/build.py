"""Build every native artefact of the package in-tree (no JIT caches, nothing in site-packages).

    python -m microcket_amd.build            # library + executable
    python -m microcket_amd.build --all      # + oracle, reference build, test tools

libmkt_hip.so   hipcc --offload-arch=gfx950, hand-written HIP kernels + the C ABI (include/mkt.h)
bin/sam2pairs   the drop-in executable (same argv as the reference's bin/sam2pairs)
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmkt_hip.so")
EXE = os.path.join(HERE, "bin", "sam2pairs")
ARCH = "gfx950"


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _run(cmd, **kw):
    print("+", " ".join(cmd), flush=True)
    subprocess.check_call(cmd, **kw)


def _headers():
    """Every header a native artefact may include: csrc/*.h (all of them, so that a new header can never be forgotten) + the ABI."""
    import glob
    return sorted(glob.glob(os.path.join(CSRC, "*.h"))) + sorted(glob.glob(os.path.join(ROOT, "include", "*.h")))


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: the HIP library cannot be built (there is no CPU build of this package)")


def _lib_sources():
    return [os.path.join(CSRC, f) for f in ("mkt_kernels.hip", "mkt_sort.hip", "mkt_bam.hip", "mkt_capi.cpp")]


OBJ = os.path.join(HERE, "_build", "obj")


def _compile_objects(srcs, tag, flags):
    """One object per source (rebuilt only when it or a header is newer): a change in one file does not recompile the kernels."""
    os.makedirs(OBJ, exist_ok=True)
    hdrs = _headers()
    objs, procs = [], []
    for s in srcs:
        o = os.path.join(OBJ, tag + os.path.basename(s) + ".o")
        objs.append(o)
        if _newer(o, [s] + hdrs):
            cmd = [hipcc(), "-c", "-fPIC", f"--offload-arch={ARCH}", "-O3", "-std=c++17", *flags, "-Wno-unused-function", "-I" + CSRC, s, "-o", o]
            print("+", " ".join(cmd), flush=True)
            procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, pr in procs:
        if pr.wait() != 0:
            raise subprocess.CalledProcessError(pr.returncode, cmd)
    return objs


def build_lib(force=False):
    srcs = _lib_sources()
    deps = srcs + _headers()
    if force or _newer(LIB, deps):
        objs = _compile_objects(srcs, "", ("-Wall",))
        _run([hipcc(), "-shared", "-fPIC", f"--offload-arch={ARCH}", "-Wl,-rpath,/opt/rocm/lib", *objs, "-o", LIB])
    return LIB


def build_stamps_lib(name="libmkt_hip_stamps.so", defines=("-DMKT_STAMPS",)):
    """Diagnostic builds (phase stamps / phase ladder / tile-geometry experiments); never loaded by default."""
    out = os.path.join(HERE, name)
    srcs = _lib_sources()
    _run([hipcc(), "-shared", "-fPIC", f"--offload-arch={ARCH}", "-O3", "-std=c++17", *defines, "-Wno-unused-function",
          "-Wl,-rpath,/opt/rocm/lib", *srcs, "-o", out])
    return out


def build_variant(name, kernel_src=None, defines=()):
    """Experiment builds: another kernel source file and / or extra -D flags -> microcket_amd/<name> (select it with MKT_LIB)."""
    out = os.path.join(HERE, name)
    srcs = [kernel_src or os.path.join(CSRC, "mkt_kernels.hip")] + _lib_sources()[1:]
    _run([hipcc(), "-shared", "-fPIC", f"--offload-arch={ARCH}", "-O3", "-std=c++17", *defines, "-Wno-unused-function", "-I" + CSRC,
          "-Wl,-rpath,/opt/rocm/lib", *srcs, "-o", out])
    return out


def build_exe(force=False):
    src = os.path.join(CSRC, "sam2pairs_main.cpp")
    if force or _newer(EXE, [src, LIB] + _headers()):
        os.makedirs(os.path.dirname(EXE), exist_ok=True)
        _run(["g++", "-O2", "-std=c++17", "-Wall", "-pthread", src, "-o", EXE, "-L" + HERE, "-lmkt_hip",
              "-Wl,-rpath,$ORIGIN/..", "-Wl,-rpath,/opt/rocm/lib", "-Wl,-rpath-link,/opt/rocm/lib"])
    return EXE


PAIRSORT = os.path.join(HERE, "bin", "pairsort")


def build_pairsort(force=False):
    """bin/pairsort: the driver's sort / sort -m step on the GPU."""
    src = os.path.join(CSRC, "pairsort_main.cpp")
    if force or _newer(PAIRSORT, [src, LIB] + _headers()):
        os.makedirs(os.path.dirname(PAIRSORT), exist_ok=True)
        _run(["g++", "-O2", "-std=c++17", "-Wall", src, "-o", PAIRSORT, "-L" + HERE, "-lmkt_hip",
              "-Wl,-rpath,$ORIGIN/..", "-Wl,-rpath,/opt/rocm/lib", "-Wl,-rpath-link,/opt/rocm/lib"])
    return PAIRSORT


KRMDUP = os.path.join(HERE, "bin", "krmdup")
KRMDUP_PIPE = os.path.join(HERE, "bin", "krmdup.pipe")


def build_krmdup(force=False):
    """bin/krmdup and bin/krmdup.pipe: drop-ins for the reference's FASTQ duplicate removal; one program,
    the interleaved-stdout form is chosen by its name."""
    src = os.path.join(CSRC, "krmdup_main.cpp")
    if force or _newer(KRMDUP, [src, LIB] + _headers()):
        os.makedirs(os.path.dirname(KRMDUP), exist_ok=True)
        _run(["g++", "-O2", "-std=c++17", "-Wall", src, "-o", KRMDUP, "-L" + HERE, "-lmkt_hip",
              "-Wl,-rpath,$ORIGIN/..", "-Wl,-rpath,/opt/rocm/lib", "-Wl,-rpath-link,/opt/rocm/lib"])
    if force or _newer(KRMDUP_PIPE, [KRMDUP]):
        shutil.copy2(KRMDUP, KRMDUP_PIPE)
    return KRMDUP


SAM2BAM = os.path.join(HERE, "bin", "sam2bam")


def build_sam2bam(force=False):
    """bin/sam2bam: the driver's `samtools view -b | samtools sort; samtools index` tail on the GPU."""
    src = os.path.join(CSRC, "sam2bam_main.cpp")
    if force or _newer(SAM2BAM, [src, LIB] + _headers()):
        os.makedirs(os.path.dirname(SAM2BAM), exist_ok=True)
        _run(["g++", "-O2", "-std=c++17", "-Wall", src, "-o", SAM2BAM, "-L" + HERE, "-lmkt_hip",
              "-Wl,-rpath,$ORIGIN/..", "-Wl,-rpath,/opt/rocm/lib", "-Wl,-rpath-link,/opt/rocm/lib"])
    return SAM2BAM


MAKESTAT = os.path.join(HERE, "bin", "makestat")


def build_makestat(force=False):
    """bin/makestat: native stand-in for the reference's bin/make.stat.pl; host-only, no GPU library."""
    src = os.path.join(CSRC, "makestat_main.cpp")
    if force or _newer(MAKESTAT, [src]):
        os.makedirs(os.path.dirname(MAKESTAT), exist_ok=True)
        _run(["g++", "-O2", "-std=c++17", "-Wall", src, "-o", MAKESTAT])
    return MAKESTAT


def build_oracle():
    """Test infrastructure: CPU restatement (+ the reference itself when /root/reference is present)."""
    _run(["make", "-s", "-C", os.path.join(ROOT, "oracle")])


def build_test_tools():
    out = os.path.join(ROOT, "tests", "host", "_build")
    os.makedirs(out, exist_ok=True)
    emul = os.path.join(out, "libmkt_emul.so")
    src = os.path.join(ROOT, "tests", "host", "tile_emul.cpp")
    if _newer(emul, [src] + _headers()):
        _run(["g++", "-O1", "-g", "-std=c++17", "-Wall", "-fPIC", "-shared", "-o", emul, src])
    tout = os.path.join(ROOT, "tools", "_build")
    os.makedirs(tout, exist_ok=True)
    synth = os.path.join(tout, "synth_sam")
    ssrc = os.path.join(ROOT, "tools", "synth_sam.cpp")
    if _newer(synth, [ssrc, os.path.join(CSRC, "mkt_synth.h")]):
        _run(["g++", "-O2", "-std=c++17", "-Wall", "-o", synth, ssrc])


def build_all(force=False, extras=True):
    build_lib(force)
    build_exe(force)
    build_pairsort(force)
    build_krmdup(force)
    build_sam2bam(force)
    build_makestat(force)
    if extras:
        build_oracle()
        build_test_tools()


if __name__ == "__main__":
    build_all(force="--force" in sys.argv, extras="--all" in sys.argv or True)
