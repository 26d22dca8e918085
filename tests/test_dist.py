"""The N > 1 path: contiguous group-aligned shards, no data-path collective, and the two global
quirks (Q1 last-group drop, Q2 thread-0 selfCircle share) reconciled with one all_gather + one
all_reduce.  Runs here with world_size 2 and 3 over gloo; the per-rank engine is the CPU emulation
of the tile phases (tests/host), since this container has no GPU."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import util
from microcket_amd import shard


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, path, mode, threads, outdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    text = open(path, "rb").read()
    cuts = shard.cut_points(text, world)
    part = text[cuts[rank]:cuts[rank + 1]]
    eng = util.EmulShard(mode, 0.5, 10, True, threads, cfg=2)
    eng.feed(part, block=50000)

    def all_gather_int(x):
        out = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(out, torch.tensor([x], dtype=torch.int64))
        return [int(t.item()) for t in out]

    def all_reduce_sum(v):
        t = torch.tensor(v, dtype=torch.int64)
        dist.all_reduce(t)
        return t.tolist()

    st, counters, total = shard.finish_sharded(eng, rank, world, all_gather_int, all_reduce_sum)
    open(os.path.join(outdir, f"pairs.{rank}"), "wb").write(st.pairs_bytes)
    open(os.path.join(outdir, f"sam.{rank}"), "wb").write(st.sam_bytes)
    if rank == 0:
        open(os.path.join(outdir, "log"), "wb").write(shard.format_log(counters))
        open(os.path.join(outdir, "total"), "w").write(str(total))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("profile,mode,groups", [("stress", "unc", 6000), ("flash", "flash", 3000)])
def test_sharded_equals_single_stream(tmp_path, world, profile, mode, groups):
    text = util.synth(profile, 4242, groups)
    path = tmp_path / "in.sam"
    path.write_bytes(text)
    threads = 4
    mp.spawn(_worker, args=(world, _free_port(), str(path), mode, threads, str(tmp_path)), nprocs=world, join=True)
    po, so, lo, st = util.oracle_run(text, mode, threads, 0.5, 10, True)
    pairs = b"".join((tmp_path / f"pairs.{r}").read_bytes() for r in range(world))
    sam = b"".join((tmp_path / f"sam.{r}").read_bytes() for r in range(world))
    assert pairs == po            # contiguous shards in rank order reproduce the input order
    assert sam == so
    assert (tmp_path / "log").read_bytes() == lo
    assert int((tmp_path / "total").read_text()) == st.groups


def test_cut_points_are_group_boundaries():
    text = util.synth("unc", 9, 500)
    for parts in (2, 3, 8, 17):
        cuts = shard.cut_points(text, parts)
        assert cuts[0] == 0 and cuts[-1] == len(text) and cuts == sorted(cuts)
        for c in cuts[1:-1]:
            assert text[c - 1:c] == b"\n"
            prev_ls = text.rfind(b"\n", 0, c - 1) + 1
            assert text[prev_ls:].split(b"\t", 1)[0] != text[c:].split(b"\t", 1)[0]


def test_more_ranks_than_groups():
    text = util.synth("unc", 3, 2, tail=1)      # 3 groups
    cuts = shard.cut_points(text, 8)
    assert cuts[-1] == len(text)
    assert b"".join(text[cuts[i]:cuts[i + 1]] for i in range(8)) == text


def _gather_worker(rank, world, port, outdir):
    import numpy as np
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ag_obj, ag_keys = shard.torch_gatherers(dist, torch, "cpu", world)
    k = (np.arange((rank + 1) * 5 * 3, dtype=np.uint64).reshape(-1, 3) + np.uint64(rank) * np.uint64(1 << 60))
    parts = ag_keys(k)
    objs = ag_obj({rank: b"chr%d" % rank})
    ok = len(parts) == world and all(p.shape == ((r + 1) * 5, 3) and int(p[0, 0] >> np.uint64(60)) == r for r, p in enumerate(parts))
    ok = ok and objs == [{r: b"chr%d" % r} for r in range(world)]
    open(os.path.join(outdir, f"ok.{rank}"), "w").write("1" if ok else "0")
    dist.barrier()
    dist.destroy_process_group()


def test_key_space_gatherers_over_gloo(tmp_path):
    """The exchange primitives of the sharded duplicate marking (shard.torch_gatherers), world_size 2 on CPU."""
    mp.spawn(_gather_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok.0").read_text() == "1" and (tmp_path / "ok.1").read_text() == "1"


def _exchange_worker(rank, world, port, path, lanes, outdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lines = open(path, "rb").read().split(b"\n")[:-1]
    cuts = [len(lines) * r // world for r in range(world + 1)]
    if world == 3:
        cuts[1] = 1                      # a nearly empty shard
    eng = util.KeyEngine(lines[cuts[rank]:cuts[rank + 1]], rank, lanes)
    flags, dups, total = shard.dedup_exchange(eng, rank, world, rank == world - 1, dist, torch, "cpu")
    open(os.path.join(outdir, f"flags.{rank}"), "wb").write(flags)
    open(os.path.join(outdir, f"tot.{rank}"), "w").write(f"{dups} {total}")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("lanes", [False, True])
def test_hash_partitioned_dedup_exchange_over_gloo(tmp_path, world, lanes):
    """shard.dedup_exchange (name-table gather, all_to_all of the key space by key hash, flags back, all_reduce) with world_size
    2 and 3 over gloo; the per-rank engine is a numpy stand-in that uses the same key record layout.  The flags of all
    ranks, concatenated, must equal the single-stream definition -- also with the lane in the key (the driver's -b)."""
    text = util.synth("unc", 515, 4000, 100, "mm10", 4)
    po, so, lo, st = util.oracle_run(text, "unc", 4, 0.5, 10, False)
    lines = po.split(b"\n")[:-1]
    # make duplicates: repeat every 5th contact under another read name, in the same lane and in another one
    extra = []
    for j, l in enumerate(lines[::5]):
        f = l.split(b"\t")
        q = f[0].split(b":")
        q[-1] = b"9%d" % j
        extra.append(b"\t".join([b":".join(q)] + f[1:]))
        q[3] = b"%d" % (int(q[3]) % 4 + 1)
        extra.append(b"\t".join([b":".join(q)] + f[1:]))
    allp = b"\n".join(lines + extra) + b"\n"
    path = tmp_path / "pairs"
    path.write_bytes(allp)
    want = util.expected_dups(allp, lanes)
    assert 0 < sum(want) < len(want)
    mp.spawn(_exchange_worker, args=(world, _free_port(), str(path), lanes, str(tmp_path)), nprocs=world, join=True)
    got = b"".join((tmp_path / f"flags.{r}").read_bytes() for r in range(world))
    assert got == want
    parts = [(tmp_path / f"tot.{r}").read_text().split() for r in range(world)]
    assert sum(int(p[0]) for p in parts) == sum(want) and all(int(p[1]) == sum(want) for p in parts)
