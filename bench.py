#!/usr/bin/env python3
"""bench.py -- read-pairs/s through sam2pairs+dedup on MI355X (BASELINE.json's metric).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload (config.workload): BASELINE.json configs[1] -- 100 M synthetic 150 bp PE read pairs with
hg38 chromosome names (seeded generator), unstitched mode, sam=no (the driver's -x),
~92 GB of SAM text resident in HBM per GPU before the timed region starts.  One STEP = one pass
of the hot path over the whole resident data set (all blocks) followed by the duplicate marking of the
reported pairs and the per-chromosome-pair counts (BASELINE.json: "sam2pairs+dedup"; both are extensions
that the reference's sam2pairs does not have, so the same steps WITHOUT them -- exactly the reference's
behaviour and outputs -- are timed too and reported as `sam2pairs_only`; `--dedup no` makes that the value).  N > 1: every rank holds its own
100 M-pair shard (weak scaling; shard r = groups [r*P, (r+1)*P) of one seeded data set); the only
exchange is the group-count all_gather + 8-counter all_reduce at the end (quirks Q1/Q2).

`value` = pairs processed by all ranks / wall time of K steps (max over ranks), inputs resident.
`roofline`: the fused tile kernel (k_fast), algorithmic bytes (SAM bytes in + .pairs bytes out)
per launch over its HIP-event launch duration, against 8 TB/s HBM.
`cpu_baseline`: the reference itself (oracle/_ref/sam2pairs.ref, built from /root/reference by
oracle/Makefile) timed on this host on a bounded sample of the same data (first blocks).
`end_to_end` (N = 1): SURVEY.md 8(d)'s metric -- the drop-in executable against the reference on that same sample FILE,
process start to exit, sam=no and sam=yes; `speedup_vs_cpu_baseline` is that like-for-like ratio.
`--config C2|C3|C4|C5` picks BASELINE.json's configs[1..4]: pairs per GPU, read length, genome names, lanes and the lane-scoped
duplicate key (the driver's -b); C2 is the default, C3 / C5 are the shapes of the 8- and 4-GPU configs (per-GPU share).
`c4_unc100`, `c5_mm10_4lanes` (N = 1): the C4 / C5 shapes as resident legs with their own k_fast roofline.
`sam_yes`, `flash` (N = 1): the same resident path with the .sam pass-through on / in stitched mode, each with its own
k_fast roofline.  `roofline.traffic` is imported from the committed rocprofv3 --pmc passes (profiles/), not measured in
this process.
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def cpu_baseline(ctx, ds, sample_groups, threads, mode="unc"):
    """Times the reference (or, if absent, the C restatement) on the first blocks of the data set, and -- on the SAME file in
    /dev/shm -- the drop-in executable (wall clock including process start, file reads, PCIe both ways and all writes):
    SURVEY.md 8(d)'s end-to-end metric, sam=no and sam=yes, from a regular file and through a pipe (`cat file | exe /dev/stdin`,
    the driver's real input surface, microcket:479-506).  Returns (cpu_baseline, end_to_end)."""
    ref = os.path.join(ROOT, "oracle", "_ref", "sam2pairs.ref")
    port = os.path.join(ROOT, "oracle", "_build", "sam2pairs_oracle")
    exe, kind = (ref, "reference") if os.path.exists(ref) else (port, "port")
    if not os.path.exists(exe):
        return None, None
    import microcket_amd as m
    tmpdir = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else tempfile.gettempdir()
    d = tempfile.mkdtemp(prefix="mkt_bench_", dir=tmpdir)
    path = os.path.join(d, "sample.sam")
    groups = 0
    nbytes = 0
    try:
        with open(path, "wb") as f:
            for (p, n, g) in ds.blocks:
                f.write(ctx.copy_to_host(p, n))
                groups += g
                nbytes += n
                if groups >= sample_groups:
                    break

        def run(binary, nthreads, sam, pipe=False):
            args = [binary, "/dev/stdin" if pipe else path, mode, os.path.join(d, "out"), str(nthreads), "0.5", "10", sam]
            t0 = time.time()
            with open(os.devnull, "wb") as o:
                if pipe:
                    cat = subprocess.Popen(["cat", path], stdout=subprocess.PIPE)
                    rc = subprocess.run(args, stdin=cat.stdout, stdout=o, stderr=subprocess.PIPE).returncode
                    cat.stdout.close()
                    rc = rc or cat.wait()
                else:
                    rc = subprocess.run(args, stdout=o, stderr=subprocess.PIPE).returncode
            return rc, time.time() - t0

        def times(binary, nthreads, sam, pipe, reps):
            ts = []
            for _ in range(reps):
                rc, dt = run(binary, nthreads, sam, pipe)
                if rc != 0:
                    return None
                ts.append(dt)
            return ts

        tref = times(exe, threads, "no", False, 2)
        if not tref:
            return None, None
        dt = min(tref)
        cpu = {"value": groups / dt, "unit": "read-pairs/s", "cores": threads if kind == "reference" else 1, "kind": kind,
               "sample": f"first {groups} pairs ({nbytes / 1e9:.2f} GB SAM) of the same data set, file input in /dev/shm, sam=no, thread={threads}, "
                         f"best of {len(tref)} runs ({', '.join('%.1f' % t for t in tref)} s)",
               "host_cpus": os.cpu_count()}
        many = min(os.cpu_count() or 1, 64)
        if kind == "reference" and many > threads:              # also at thread = min(nproc, 64)
            t2 = times(exe, many, "no", False, 1)
            if t2:
                cpu["more_threads"] = {"cores": many, "value": groups / t2[0], "seconds": t2[0]}
        e2e = None
        mine = m.exe_path()
        if os.path.exists(mine):
            e2e = {"unit": "read-pairs/s", "what": "the sam2pairs executables on the same input: wall clock from process start to exit, "
                   "input from a file in /dev/shm (sam_*) or through `cat file | exe /dev/stdin` (pipe_*), stdout to /dev/null, side files to "
                   "/dev/shm; best run of each side, every run listed; never `value`",
                   "sample_pairs": groups, "sample_bytes": nbytes}
            for key, sam, pipe, reps_ref in (("sam_no", "no", False, 0), ("sam_yes", "yes", False, 1), ("pipe_no", "no", True, 1), ("pipe_yes", "yes", True, 1)):
                tm = times(mine, threads, sam, pipe, 3 if not pipe else 2)
                tr = tref if key == "sam_no" else times(exe, threads, sam, pipe, reps_ref)
                if tm:
                    best, dref = min(tm), (min(tr) if tr else None)
                    e2e[key] = {"mi355x": groups / best, "mi355x_seconds": best, "mi355x_GBps": nbytes / best / 1e9, "mi355x_runs_s": tm,
                                "cpu_" + kind: (groups / dref) if dref else None, "cpu_seconds": dref, "cpu_runs_s": tr,
                                "speedup": (dref / best) if dref else None,
                                "speedup_min_max": [min(tr) / max(tm), max(tr) / min(tm)] if tr else None}
        return cpu, e2e
    finally:
        try:
            for fn in os.listdir(d):
                os.unlink(os.path.join(d, fn))
            os.rmdir(d)
        except OSError:
            pass


def resident_leg(m, local, mode, sam, pairs, block_groups, read_len, steps, warmup, seed, tiles, barrier, genome=0, lanes=1, dedup=False, what=""):
    """One more resident workload on its own context + data set (sam=yes, flash mode, the C4 / C5 shapes): pairs/s and the k_fast
    roofline.  dedup: every step also marks duplicates (lane-scoped when lanes > 1: the driver's -b) and counts chromosome pairs."""
    ext = (m.EXT_KEYS | (m.EXT_LANES if lanes > 1 else 0)) if dedup else 0
    ctx = m.Context(mode, 0.5, 10, sam, 8, device=local, tiles=tiles, extensions=ext)
    ds = ctx.dataset(seed, 0 if mode == "unc" else 1, pairs, block_groups, genome=genome, read_len=read_len, lanes=lanes, tail_group=True)
    extra = {}

    def one_step():
        if dedup:
            ctx.reset()
        for (p, n, _g) in ds.blocks:
            ctx.submit_device(p, n)
        if dedup:
            tot, dups, _ = ctx.ext_dedup(True, want_flags=False)
            rows = ctx.ext_chrstat(True)
            extra.update(reported_pairs=tot, duplicates=dups, chrstat_rows=len(rows.splitlines()))
    try:
        for _ in range(max(warmup, 1)):
            one_step()
        ctx.sync(); ctx.reset(); ctx.reset_timing()
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            one_step()
        ctx.sync()
        el = time.perf_counter() - t0
        tm = ctx.timing()
        ctx.reset()
        for (p, n, _g) in ds.blocks:
            ctx.submit_device(p, n)
        st = ctx.finish(True)
        out_b = st.pair_bytes + (st.sam_bytes if sam else 0)
        algo = (ds.total_bytes + out_b) * steps
        ach = algo / (tm.tile_kernel_ms / 1e3) / 1e9 if tm.tile_kernel_ms > 0 else 0.0
        return {"value": ds.total_groups * steps / el, "unit": "read-pairs/s", "ms_per_step": el / steps * 1e3, "pairs": ds.total_groups,
                "workload": f"{what}{ds.total_groups} synthetic {read_len} bp pairs, {'mm10' if genome else 'hg38'} names, {lanes} lane(s), {mode} mode, "
                            f"sam={'yes' if sam else 'no'}, {ds.total_bytes / 1e9:.1f} GB resident" + (", every step with duplicate marking + chromosome-pair counts" if dedup else ""),
                **extra,
                "bytes_per_pair_in": ds.total_bytes / ds.total_groups, "bytes_per_pair_out": out_b / ds.total_groups,
                "roofline": {"bound": "hbm", "kernel": "k_fast", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                             "avg_launch_ms": tm.tile_kernel_ms / tm.tile_launches if tm.tile_launches else None,
                             "tiles_left_to_generic_kernel": tm.deferred_tiles}}
    finally:
        ds.close()
        ctx.close()


# BASELINE.json configs[1..4]: what one GPU holds.  (configs[0] is the reference's own CPU plumbing case: a parity-test size, no bench line.)
PRESETS = {
    "C2": dict(pairs=100_000_000, read_len=150, genome=0, lanes=1, what="C2: 100 M 150 bp PE Micro-C read pairs, hg38, one MI355X"),
    "C3": dict(pairs=125_000_000, read_len=150, genome=0, lanes=1, what="C3: 1 B 150 bp PE Hi-C read pairs, hg38, sharded across 8 MI355X (125 M per GPU), cross-shard duplicate marking"),
    "C4": dict(pairs=100_000_000, read_len=100, genome=0, lanes=1, what="C4: unstitched mode, split-read chimeric alignments, 2 x 100 bp PE, hg38, one MI355X (100 M pairs)"),
    "C5": dict(pairs=100_000_000, read_len=100, genome=1, lanes=4, what="C5: mm10 100 bp PE Hi-C, 4 MI355X (100 M pairs per GPU), 4 lanes, inter-lane duplicates kept (-b)"),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="C2", choices=sorted(PRESETS), help="BASELINE.json config: pairs per GPU, read length, genome names, lanes (-b)")
    ap.add_argument("--pairs", type=int, default=None, help="read pairs per GPU (default: the config's; MKT_BENCH_PAIRS overrides)")
    ap.add_argument("--block-groups", type=int, default=1 << 21, help="read groups per block (one kernel pass); 2^21 groups = 1.9 GB of SAM text")
    ap.add_argument("--sam", default="no", choices=["no", "yes"])
    ap.add_argument("--read-len", type=int, default=None, help="read length of the synthetic data (default: the config's)")
    ap.add_argument("--tiles", default="auto", choices=["fast", "auto"], help="tile geometry: chosen per input after a probe block (the library default), or the 48 KiB lean tiles forced")
    ap.add_argument("--mode", default="unc", choices=["unc", "flash"])
    ap.add_argument("--dedup", default="yes", choices=["yes", "no"],
                    help="yes (BASELINE.json's metric, sam2pairs+dedup): every step also marks duplicate pairs and counts chromosome pairs")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-legs", action="store_true", help="skip the sam=yes and flash resident legs")
    ap.add_argument("--leg-pairs", type=int, default=32_000_000, help="read pairs of the sam=yes / flash legs")
    ap.add_argument("--cpu-sample-pairs", type=int, default=12_000_000, help="pairs of the file the CPU reference and the end-to-end executables are timed on (~16 s of reference time per run)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="torch.distributed backend for N > 1 (nccl = RCCL over xGMI)")
    ap.add_argument("--same-gpu", action="store_true", help="rehearsal only: every rank uses GPU 0 (needs --backend gloo)")
    ap.add_argument("--force-dist", action="store_true", help="rehearsal only: run the N > 1 code path (process group, RCCL all_to_all of the key space) with whatever WORLD_SIZE is, even 1")
    args = ap.parse_args()
    preset = PRESETS[args.config]
    if args.pairs is None:
        args.pairs = int(os.environ.get("MKT_BENCH_PAIRS", preset["pairs"]))
    if args.read_len is None:
        args.read_len = preset["read_len"]
    genome, lanes = preset["genome"], preset["lanes"]

    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if world != args.gpus:
        log(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE")
    dist = None
    torch = None
    if world > 1 or args.force_dist:
        import torch  # noqa: F811
        import torch.distributed as dist  # noqa: F811
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if args.same_gpu:
            local = 0
        if args.backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend="gloo")
    # (N = 1 never imports torch: ctx.sync() brackets the timed region.  At N > 1 torch comes first on purpose: its bundled
    #  libamdhip64.so carries the soname libmkt_hip.so asks for, so the library binds to torch's HIP runtime and the process
    #  holds ONE runtime; the other order would load two, and the second one sees no GPU.)

    import microcket_amd as m
    from microcket_amd import shard

    def barrier():
        if dist is not None:
            dist.barrier()

    tdev = "cuda" if (dist is None or args.backend == "nccl") else "cpu"

    profile = 0 if args.mode == "unc" else 1
    seed = 20260104 + int(args.config[1]) - 1  # seeds 20260104 + config index (BASELINE.json configs[1] = C2)
    tiles = m.TILES_FAST if args.tiles == "fast" else m.TILES_AUTO
    dedup = args.dedup == "yes"
    ext = (m.EXT_KEYS | (m.EXT_LANES if lanes > 1 else 0)) if dedup else 0      # lanes > 1: the lane joins the duplicate key (the driver's -b)
    ctx = m.Context(args.mode, 0.5, 10, args.sam == "yes", 8, device=local, tiles=tiles, extensions=ext)
    t0 = time.time()
    ds = ctx.dataset(seed, profile, args.pairs, args.block_groups, first_group=rank * args.pairs, genome=genome, read_len=args.read_len, lanes=lanes,
                     tail_group=(rank == world - 1))
    log(f"[rank {rank}] data set: {ds.total_groups} pairs, {ds.total_bytes / 1e9:.2f} GB in {ds.n_blocks} blocks, generated in {time.time() - t0:.1f} s")
    drop_last = rank == world - 1          # every rank holds pairs; the input's end is on the last one (quirk Q1)
    global_dedup = dedup and dist is not None and args.backend == "nccl"
    ext_out = {}

    def run_steps(c, steps, with_dedup):
        """`steps` passes of the hot path over the resident data set.  With dedup every step is a whole input: sam2pairs over all
        blocks, then duplicate marking of the reported pairs and the per-chromosome-pair counts (this rank's shard)."""
        for _ in range(steps):
            if with_dedup:
                c.reset()
            for (p, n, _g) in ds.blocks:
                c.submit_device(p, n)
            if with_dedup:
                if global_dedup:
                    # N > 1: the key space is hash-partitioned across the ranks (all_to_all over RCCL / xGMI, device buffers),
                    # marked where the equal keys meet, flags returned: duplicates are found across shards, inside the step
                    _f, dups, dups_all = shard.dedup_exchange(c, rank, world, drop_last, dist, torch, torch.device("cuda", local), want_flags=False)
                    tot = c.ext_key_count(drop_last)
                    ext_out.update(duplicates_all_ranks=dups_all)
                else:
                    tot, dups, _ = c.ext_dedup(drop_last, want_flags=False)
                rows = c.ext_chrstat(drop_last)
                ext_out.update(reported_pairs=tot, duplicates=dups, chrstat_rows=len(rows.splitlines()))
        c.sync()

    def timed(c, with_dedup):
        run_steps(c, max(args.warmup, 0), with_dedup)
        if not with_dedup:
            c.reset()
        c.reset_timing()
        # ---- timed region: exactly K steps
        if torch is not None and tdev == "cuda" and torch.cuda.is_available():
            torch.cuda.synchronize()
        c.sync(); barrier()
        t_start = time.perf_counter()
        run_steps(c, args.steps, with_dedup)
        if torch is not None and tdev == "cuda" and torch.cuda.is_available():
            torch.cuda.synchronize()
        c.sync(); barrier()
        el = time.perf_counter() - t_start
        if dist is not None:
            tt = torch.tensor([el], dtype=torch.float64, device=tdev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = float(tt.item())
        return el, c.timing()

    elapsed, tmg = timed(ctx, dedup)
    tm_ms, tm_launch, tm_bytes = tmg.tile_kernel_ms, tmg.tile_launches, tmg.tile_bytes
    tiles_seen, tiles_deferred = tmg.tiles, tmg.deferred_tiles

    # the same passes without the extensions (exactly the reference's behaviour and outputs), on a second context
    plain = None
    if dedup:
        ctx2 = m.Context(args.mode, 0.5, 10, args.sam == "yes", 8, device=local, tiles=tiles)      # (no extensions)
        el2, tmg2 = timed(ctx2, False)
        plain = (el2, tmg2.tile_kernel_ms, tmg2.tile_launches)
        ctx2.close()

    # ---- one more (untimed) pass for the end-of-input bookkeeping and the counters of ONE pass
    ctx.reset()
    for (p, n, _g) in ds.blocks:
        ctx.submit_device(p, n)
    my_groups = ctx.group_count()
    if dist is not None:
        gl = [None] * world
        dist.all_gather_object(gl, my_groups)
        offset = sum(gl[:rank])
        total = sum(gl)
        last_nonempty = max(i for i, g in enumerate(gl) if g > 0)
        st = ctx.finish(drop_last=(rank == last_nonempty), group_offset=offset, total_groups=total)
        cnt = torch.tensor([st.lowMap, st.manyHits, st.unpaired, st.selfCircle, st.trans, st.cis10K, st.cis1K, st.cis0, st.pairs, st.pair_bytes,
                            ext_out.get("duplicates", 0)], dtype=torch.int64, device=tdev)
        dist.all_reduce(cnt)
        counters = [int(x) for x in cnt.tolist()]
    else:
        st = ctx.finish(True)
        counters = [st.lowMap, st.manyHits, st.unpaired, st.selfCircle, st.trans, st.cis10K, st.cis1K, st.cis0, st.pairs, st.pair_bytes,
                    ext_out.get("duplicates", 0)]

    pairs_per_step = ds.total_groups
    out_bytes_step = st.pair_bytes + (st.sam_bytes if args.sam == "yes" else 0)
    algo_bytes_step = ds.total_bytes + out_bytes_step
    total_pairs = pairs_per_step * args.steps * world
    value = total_pairs / elapsed
    result = None
    if rank == 0:
        achieved = (algo_bytes_step * args.steps) / (tm_ms / 1e3) / 1e9 if tm_ms > 0 else 0.0
        # HBM traffic of k_fast per launch: rocprofv3 cannot run inside this process, so the figure comes from the committed
        # PMC passes of this same command (tools/round_profiles.sh: separate --pmc FETCH_SIZE / WRITE_SIZE runs, FETCH_SIZE
        # doubled as MI355X_MICROARCH.md prescribes for 16-byte-per-lane streams), as measured traffic / algorithmic bytes,
        # applied to this run's algorithmic bytes per launch.  Only for the profiled workload shape; otherwise null.
        traffic, traffic_src = None, None
        try:
            import glob
            cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")))
            if cands and tm_launch and args.config == "C2" and args.read_len == 150 and args.sam == "no" and args.mode == "unc":
                tj = json.loads(open(cands[-1]).read())
                if tj.get("block_groups") == args.block_groups and ("duplicate marking" in tj.get("workload", "")) == dedup:
                    traffic = tj["traffic_over_algorithmic"] * algo_bytes_step * args.steps / tm_launch
                    traffic_src = f"profiles/{os.path.basename(cands[-1])}: traffic/algorithmic = {tj['traffic_over_algorithmic']:.3f} (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE)"
        except Exception:
            traffic, traffic_src = None, None
        cpu, e2e = None, None
        if world == 1 and not args.no_cpu_baseline:
            cpu, e2e = cpu_baseline(ctx, ds, args.cpu_sample_pairs, 8, args.mode)
        result = {
            "metric": ("read-pairs/sec through sam2pairs+dedup" if dedup else "read-pairs/sec through sam2pairs")
                      + f" ({args.read_len} bp PE, {'mm10' if genome else 'hg38'} names, SAM text resident in HBM)",
            "value": value,
            "unit": "read-pairs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "config": {
                "workload": f"{preset['what']} -- {args.pairs} synthetic {args.read_len} bp PE read pairs per GPU, {'mm10' if genome else 'hg38'} chromosome names, "
                            f"{lanes} lane(s), {args.mode} mode, sam={args.sam}, seed {seed}, one step = one pass over the whole resident data set"
                            + (" + duplicate marking of the reported pairs on (chr1,pos1,chr2,pos2,strands) + per-chromosome-pair counts" if dedup else ""),
                "preset": args.config,
                "pairs_per_gpu": ds.total_groups,
                "sam_bytes_per_gpu": ds.total_bytes,
                "bytes_per_pair_in": ds.total_bytes / ds.total_groups,
                "bytes_per_pair_out": out_bytes_step / ds.total_groups,
                "block_groups": args.block_groups,
                "blocks": ds.n_blocks,
                "parallelism": f"shard{world}: independent read-group shards, no data-path collective",
            },
            "roofline": {
                "bound": "hbm",
                "kernel": "k_fast (fused newline scan / parse / classify / format)",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "traffic_unit": "bytes/launch",
                "traffic_measured_in_this_run": False,
                "traffic_source": traffic_src,
                "launches": tm_launch,
                "avg_launch_ms": tm_ms / tm_launch if tm_launch else None,
                "algorithmic_bytes_per_launch": algo_bytes_step * args.steps / tm_launch if tm_launch else None,
                "tiles": tiles_seen,
                "tiles_left_to_generic_kernel": tiles_deferred,
            },
            "cpu_baseline": cpu,
            "counters": dict(zip(["lowMap", "manyHits", "unpaired", "selfCircle", "trans", "cis10K", "cis1K", "cis0", "pairs", "pair_bytes", "duplicates"], counters)),
        }
        if dedup:
            result["dedup"] = dict(ext_out, scope=("global: hash-partitioned all_to_all of the key space over RCCL inside every step (microcket_amd.shard.dedup_exchange)"
                                                   if global_dedup else "one GPU holds the whole input" if world == 1 else "per GPU shard (gloo rehearsal)"),
                                   note="extension, not in the reference's sam2pairs: default off in the library, never changes stdout/.sam/.log")
            el2, ms2, ln2 = plain
            ach2 = (algo_bytes_step * args.steps) / (ms2 / 1e3) / 1e9 if ms2 > 0 else 0.0
            result["sam2pairs_only"] = {"value": total_pairs / el2, "unit": "read-pairs/s", "ms_per_step": el2 / args.steps * 1e3,
                                        "note": "the same steps without the extensions: exactly the reference's behaviour and outputs",
                                        "roofline": {"achieved": ach2, "unit": "GB/s", "frac": ach2 / HBM_PEAK_GBS, "avg_launch_ms": ms2 / ln2 if ln2 else None}}
        if e2e:
            result["end_to_end"] = e2e
            if e2e.get("sam_no", {}).get("speedup"):
                result["speedup_vs_cpu_baseline"] = e2e["sam_no"]["speedup"]
                result["speedup_note"] = "like for like: both executables on the same file, process start to exit (end_to_end.sam_no)"
        if cpu:
            result["kernel_only_ratio_vs_cpu_baseline"] = value / cpu["value"]
            result["kernel_only_ratio_note"] = "`value` has its input resident in HBM and leaves its output there; the CPU run reads a file: NOT like for like"
    ds.close()
    ctx.close()
    if world == 1 and not args.no_extra_legs and result is not None:
        # the other shapes of the same path, each with its own k_fast roofline (the 94 GB data set above is released first)
        legs_steps = max(1, min(args.steps, 5))
        try:
            if args.sam == "no":
                result["sam_yes"] = resident_leg(m, local, args.mode, True, min(args.pairs, args.leg_pairs), args.block_groups, args.read_len, legs_steps,
                                                 1, seed, tiles, barrier, genome=genome, lanes=lanes)
            if args.mode == "unc":
                result["flash"] = resident_leg(m, local, "flash", args.sam == "yes", min(args.pairs, args.leg_pairs), args.block_groups // 2, args.read_len,
                                               legs_steps, 1, seed + 100, tiles, barrier, genome=genome, lanes=lanes)
            if args.config == "C2" and args.mode == "unc" and args.sam == "no":
                # the single-GPU share of BASELINE.json configs[3] and [4], same seeds as `--config C4` / `--config C5`
                result["c4_unc100"] = resident_leg(m, local, "unc", False, min(args.pairs, args.leg_pairs), args.block_groups, PRESETS["C4"]["read_len"], legs_steps,
                                                   1, 20260104 + 3, tiles, barrier, genome=0, lanes=1, dedup=dedup, what="C4 shape: ")
                result["c5_mm10_4lanes"] = resident_leg(m, local, "unc", False, min(args.pairs, args.leg_pairs), args.block_groups, PRESETS["C5"]["read_len"], legs_steps,
                                                        1, 20260104 + 4, tiles, barrier, genome=1, lanes=4, dedup=dedup, what="C5 shape (one GPU's share): ")
        except Exception as ex:      # an auxiliary leg never takes the headline down with it
            result["extra_legs_error"] = repr(ex)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result), flush=True)


if __name__ == "__main__":
    main()
