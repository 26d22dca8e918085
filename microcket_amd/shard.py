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


def cut_points(buf: bytes, parts: int):
    """Byte offsets [0, c1, ..., len] cutting `buf` into `parts` ranges that start on a line whose
    QNAME differs from the previous line's (lines of one read are contiguous in name-grouped SAM)."""
    n = len(buf)
    cuts = [0]
    for k in range(1, parts):
        p = max(cuts[-1], n * k // parts)
        nl = buf.find(b"\n", p)
        if nl < 0:
            p = n
        else:
            p = nl + 1
            prev_ls = buf.rfind(b"\n", 0, nl) + 1
            prev = _first_token(buf, prev_ls)
            while p < n:
                cur = _first_token(buf, p)
                if cur != prev:
                    break
                nl = buf.find(b"\n", p)
                if nl < 0:
                    p = n
                    break
                prev = cur
                p = nl + 1
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
