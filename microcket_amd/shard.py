"""Sharding the sam2pairs path across ranks (one process per GPU).

QNAME groups are independent, so the input is cut into contiguous byte ranges on group boundaries
and every rank runs the whole path on its range; there is NO data-path collective.  Two reference
quirks are global and need a few integers exchanged at the end:
  Q1  only the input's very last surviving group is dropped  -> the last non-empty rank drops it;
  Q2  the logged selfCircle depends on each group's GLOBAL index and the total K
      -> all_gather of the per-rank group counts, then a SUM all_reduce of the 8 counters.
`engine` is anything with group_count() and finish(drop_last, group_offset, total_groups) returning
an object with the 8 counter attributes (microcket_amd.Context, or the test-suite's emulation).
"""
COUNTERS = ("lowMap", "manyHits", "unpaired", "selfCircle", "trans", "cis10K", "cis1K", "cis0")


def _first_token(buf, ls):
    e = ls
    n = len(buf)
    while e < n and buf[e] not in b" \t\r\n\v\f":
        e += 1
    return buf[ls:e]


def _survives(buf, ls, min_mapq):
    """The per-line filter of pairutil.h:157-161 on the line starting at ls: six tokens, decimal FLAG / POS / MAPQ,
    not FLAG & 0x700, MAPQ >= min_mapq (a header line never survives)."""
    nl = buf.find(b"\n", ls)
    tok = buf[ls:nl if nl >= 0 else len(buf)].split(None, 6)
    if len(tok) < 6 or tok[0][:1] == b"@":
        return False
    try:
        if not (tok[1].isdigit() and tok[3].isdigit() and tok[4].isdigit()):
            return False
        flag, pos, mapq = int(tok[1]), int(tok[3]), int(tok[4])
    except ValueError:
        return False
    if flag > 0xFFFFFFFF or pos > 0xFFFFFFFF or mapq > 0xFFFFFFFF:
        return False
    return not (flag & 0x700) and mapq >= min_mapq


def cut_points(buf: bytes, parts: int, min_mapq=None):
    """Byte offsets [0, c1, ..., len] cutting `buf` into `parts` ranges, each starting on a line that opens a QNAME group.

    The reference groups SURVIVING lines only (pairutil.h:157-163 filters before it compares names), so a cut is valid
    when the last surviving line before it and the first surviving line after it carry different names.  With
    min_mapq=None the raw names of adjacent lines are compared instead, which is the same thing for name-grouped
    aligner output (lines of one read contiguous)."""
    n = len(buf)

    def line_after(p):
        nl = buf.find(b"\n", p)
        return n if nl < 0 else nl + 1

    def valid(p):
        if p <= 0 or p >= n:
            return True
        if min_mapq is None:
            prev_ls = buf.rfind(b"\n", 0, p - 1) + 1
            return _first_token(buf, prev_ls) != _first_token(buf, p)
        q = p                                           # last surviving line before p
        prev = None
        while q > 0:
            ls = buf.rfind(b"\n", 0, q - 1) + 1
            if _survives(buf, ls, min_mapq):
                prev = _first_token(buf, ls)
                break
            q = ls
        if prev is None:
            return True
        q = p                                           # first surviving line at or after p
        while q < n:
            if _survives(buf, q, min_mapq):
                return _first_token(buf, q) != prev
            q = line_after(q)
        return True

    cuts = [0]
    for k in range(1, parts):
        p = max(cuts[-1], n * k // parts)
        p = line_after(p) if p > 0 else 0
        while p < n and not valid(p):
            p = line_after(p)
        cuts.append(min(p, n))
    cuts.append(n)
    return cuts


def finish_sharded(engine, rank, world, all_gather_int, all_reduce_sum):
    """Ends a sharded run.  all_gather_int(x) -> list of every rank's x; all_reduce_sum(list) -> summed list.
    Returns (local_stats, global_counters_dict, total_groups)."""
    mine = int(engine.group_count())
    counts = [int(c) for c in all_gather_int(mine)]
    offset = sum(counts[:rank])
    total = sum(counts)
    nonempty = [i for i, c in enumerate(counts) if c > 0]
    drop = bool(nonempty) and rank == nonempty[-1]
    st = engine.finish(drop_last=drop, group_offset=offset, total_groups=total)
    summed = all_reduce_sum([int(getattr(st, k)) for k in COUNTERS])
    return st, {k: int(v) & 0xFFFFFFFF for k, v in zip(COUNTERS, summed)}, total


def format_log(counters):
    return "".join(f"{k}\t{counters[k]}\n" for k in COUNTERS).encode()


def dedup_sharded(ctx, rank, world, drop_last, all_gather_obj, all_gather_keys):
    """Pairs-level duplicate marking across shards (extension): an all-gather of the dedup key space.

    Shards are contiguous ranges of the input in rank order, so the concatenation of the ranks' key lists IS the
    input order.  Chromosome slots are per context: the name tables are gathered first and every slot is rewritten to
    the rank of its name in the sorted union, which is the same on every rank.
      all_gather_obj(x)   -> list of every rank's small Python object (name tables)
      all_gather_keys(a)  -> list of every rank's (n_r, 3) uint64 array (RCCL all_gather of padded device tensors in
                             bench / production; plain lists in the single-GPU tests)
    Returns (flags of this rank's reported pairs, duplicates on this rank, total duplicates)."""
    import numpy as np
    names = ctx.ext_chr_names()
    union = sorted(set().union(*[set(t.values()) for t in all_gather_obj(names)]))
    gid = {nm: i for i, nm in enumerate(union)}
    lut = np.zeros(8192, dtype=np.uint64)
    for slot, nm in names.items():
        lut[slot] = gid[nm]
    keys = ctx.ext_keys_fetch(drop_last)
    if keys.shape[0]:
        k0 = keys[:, 0]
        a = lut[((k0 >> np.uint64(45)) & np.uint64(8191)).astype(np.int64)]
        b = lut[((k0 >> np.uint64(32)) & np.uint64(8191)).astype(np.int64)]
        keys[:, 0] = (a << np.uint64(45)) | (b << np.uint64(32)) | (k0 & np.uint64(0xFFFFFFFF))
    parts = all_gather_keys(keys)
    start = sum(p.shape[0] for p in parts[:rank])
    allk = np.concatenate(parts, axis=0) if parts else keys
    flags, total_dups = ctx.ext_dedup_keys(allk)
    mine = flags[start:start + keys.shape[0]]
    return mine, int(mine.sum()), int(total_dups)


def dedup_exchange(engine, rank, world, drop_last, dist, torch, device, want_flags=True):
    """Pairs-level duplicate marking across shards as an xGMI design (extension): the key space is
    hash-partitioned, not gathered.  Every key record travels to rank mix64(key) % world in ONE all_to_all_single of device
    buffers (RCCL over xGMI: each GPU sends 1/world of its keys over each link, nothing goes through the host), is marked
    there together with the equal keys of all other ranks, and one byte per record travels back the same way.

    "First in input order wins" survives the exchange because shards are contiguous ranges of the input in rank order:
    the partition is stable, all_to_all delivers the segments in source-rank order, so every rank marks its share in
    global input order.  Chromosome slots are per context: the (tiny) name tables are gathered first and every slot is
    rewritten to the rank of its name in the sorted union.

    engine: microcket_amd.Context (device kernels) or anything with ext_chr_names / ext_partition / ext_dedup_tensor /
    ext_unpartition (the CPU stand-in of tests/test_dist.py).  dist: torch.distributed (nccl on GPUs, gloo in the CPU tests).
    Returns (flags of this rank's reported pairs in input order, duplicates among them, duplicates of the whole run)."""
    import numpy as np
    names = engine.ext_chr_names()
    tables = [None] * world
    dist.all_gather_object(tables, names)
    union = sorted(set().union(*[set(t.values()) for t in tables]))
    gid = {nm: i for i, nm in enumerate(union)}
    lut = np.zeros(8192, dtype=np.uint16)
    for slot, nm in names.items():
        lut[slot] = gid[nm]
    send, counts = engine.ext_partition(drop_last, lut, world, torch, device)
    cs = torch.tensor(counts, dtype=torch.int64, device=device)
    cr = torch.empty_like(cs)
    dist.all_to_all_single(cr, cs)
    rcounts = [int(x) for x in cr.tolist()]
    recv = torch.empty(sum(rcounts) * 24, dtype=torch.uint8, device=device)
    dist.all_to_all_single(recv, send, output_split_sizes=[c * 24 for c in rcounts], input_split_sizes=[c * 24 for c in counts])
    if str(device).startswith("cuda"):
        torch.cuda.synchronize()
    flags_recv, _ = engine.ext_dedup_tensor(recv, torch)
    back = torch.empty(sum(counts), dtype=torch.uint8, device=device)
    dist.all_to_all_single(back, flags_recv, output_split_sizes=counts, input_split_sizes=rcounts)
    if str(device).startswith("cuda"):
        torch.cuda.synchronize()
    mine, dups = engine.ext_unpartition(back, want_flags)
    tot = torch.tensor([dups], dtype=torch.int64, device=device)
    dist.all_reduce(tot)
    return mine, int(dups), int(tot.item())


def torch_gatherers(dist, torch, device, world):
    """(all_gather_obj, all_gather_keys) over torch.distributed for dedup_sharded: the key arrays travel as padded int64
    tensors on `device` (backend nccl = RCCL over xGMI when device is a GPU; gloo with device "cpu" in the CPU tests)."""
    import numpy as np

    def ag_obj(x):
        out = [None] * world
        dist.all_gather_object(out, x)
        return out

    def ag_keys(k):
        n = torch.tensor([k.shape[0]], dtype=torch.int64, device=device)
        ns = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
        dist.all_gather(ns, n)
        ns = [int(t.item()) for t in ns]
        mx = max(max(ns), 1)
        pad = np.zeros((mx, 3), dtype=np.int64)
        pad[:k.shape[0]] = k.view(np.int64)
        t = torch.from_numpy(pad).to(device)
        outs = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(outs, t)
        return [o[:ns[r]].cpu().numpy().view(np.uint64).reshape(ns[r], 3) for r, o in enumerate(outs)]

    return ag_obj, ag_keys
