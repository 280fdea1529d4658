"""Multi-GPU sharding of independent ZPAQ blocks.

Blocks are independent units (every block resets model, coder and VM state:
Decompresser.cs:128-134, Decoder.cs:100-105), so the path shards with NO
data-path collective.  torch.distributed (RCCL over xGMI on the GPU box, gloo in
the CPU tests) carries only the block work table and the per-block results —
KiB-scale traffic:

    all_gather(block weights)  ->  every rank computes the same LPT assignment
    ... each rank decodes its own blocks, no communication ...
    all_gather(per-block {status, out_len, checksum})  ->  rank 0 reports

`ShardedJob` is the product entry point: it binds the plan to
Context.decode_blocks_device(ids = shard) on every rank (bench.py --gpus N and
the 2-rank single-GPU rehearsal in tests/ run exactly this).  `sharded_decode`
takes the decode as a function so that the table / plan / gather logic can also
be exercised on CPU-only ranks (gloo) in the `-m "not gpu"` tests.
"""
from __future__ import annotations

from typing import Callable, List, Sequence, Tuple

import numpy as np


def lpt_assign(weights: Sequence[int], world: int) -> List[List[int]]:
    """Longest-processing-time-first assignment of blocks to ranks.
    Deterministic (ties broken by block index) so every rank derives the same plan."""
    order = sorted(range(len(weights)), key=lambda i: (-int(weights[i]), i))
    load = [0] * world
    shards: List[List[int]] = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        shards[r].append(i)
        load[r] += int(weights[i])
    return shards


def interleave_assign(n_blocks: int, world: int) -> List[List[int]]:
    """Static interleave b -> rank b mod world (SURVEY.md §8e 'Partitioning')."""
    return [list(range(r, n_blocks, world)) for r in range(world)]


def all_gather_table(local: np.ndarray, dist, device=None) -> np.ndarray:
    """all_gather of a small int64 table with equal shape on every rank -> [world, ...]."""
    import torch
    t = torch.from_numpy(np.ascontiguousarray(local.astype(np.int64)))
    if device is not None:
        t = t.to(device)
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return np.stack([o.cpu().numpy() for o in out])


def broadcast_table(table: np.ndarray, dist, src: int = 0, device=None) -> np.ndarray:
    """Broadcast of the block work table (int64) from `src`; shape must be known to all ranks."""
    import torch
    t = torch.from_numpy(np.ascontiguousarray(table.astype(np.int64)))
    if device is not None:
        t = t.to(device)
    dist.broadcast(t, src=src)
    return t.cpu().numpy()


def sharded_decode(weights: Sequence[int], decode_fn: Callable[[List[int]], np.ndarray], dist, device=None,
                   policy: str = "lpt") -> Tuple[np.ndarray, List[List[int]]]:
    """Every rank holds the full stream and the full block table (`weights[b]` =
    coded bytes of block b).  Rank r decodes shard r with `decode_fn(ids)`, which
    returns an int64 array [len(ids), k] of per-block results.  Returns the
    gathered result table [n_blocks, k] (identical on all ranks) and the plan."""
    world, rank = dist.get_world_size(), dist.get_rank()
    n = len(weights)
    shards = lpt_assign(weights, world) if policy == "lpt" else interleave_assign(n, world)
    mine = shards[rank]
    res = np.asarray(decode_fn(mine), dtype=np.int64).reshape(len(mine), -1)
    k = res.shape[1] if len(mine) else 0
    kk = all_gather_table(np.array([k]), dist, device).max()
    width = max(len(s) for s in shards)
    pad = np.full((width, int(kk)), -1, np.int64)
    if len(mine):
        pad[:len(mine)] = res
    allres = all_gather_table(pad, dist, device)
    table = np.full((n, int(kk)), -1, np.int64)
    for r, ids in enumerate(shards):
        for j, b in enumerate(ids):
            table[b] = allres[r, j]
    return table, shards


def _bcast_bytes(payload, dist, src, device):
    """Broadcast a byte string whose length only `src` knows."""
    import torch
    n = torch.tensor([len(payload) if payload is not None else 0], dtype=torch.int64)
    if device is not None:
        n = n.to(device)
    dist.broadcast(n, src=src)
    n = int(n.item())
    t = torch.frombuffer(bytearray(payload), dtype=torch.uint8).clone() if payload is not None else torch.empty(n, dtype=torch.uint8)
    if device is not None:
        t = t.to(device)
    dist.broadcast(t, src=src)
    return t.cpu().numpy().tobytes()


class ShardedJob:
    """One shared compressed stream, one block table, N ranks (one per GPU): BASELINE configs[3], SURVEY.md §8e.

    Every rank holds the whole stream in its HBM (a multi-GiB archive is a few hundred MB per GiB of plaintext) and the
    same table; rank r decodes `plan[r]` — the longest-first assignment every rank derives for itself from the table's
    coded sizes — through zpaqhip_decode_blocks_device(ids = plan[r]).  `decode()` returns the per-block result table
    {status, out_len} of ALL blocks, all-gathered; the plaintext of a rank's shard stays in that rank's output buffer
    (block plan[r][j] at out_off[j])."""

    def __init__(self, ctx, h_stream: np.ndarray, d_in, sc, plan, rank, dist=None, coll_dev=None):
        self.ctx, self.h_stream, self.d_in, self.sc = ctx, h_stream, d_in, sc
        self.stream_len = int(h_stream.size)
        self.plan, self.rank, self.dist, self.coll_dev = plan, rank, dist, coll_dev
        self.shard = plan[rank]

    @staticmethod
    def _weights(sc):
        return [sum(int(sc.segments[b.first_seg + s].data_len) for s in range(b.n_seg)) for b in sc.blocks]

    @classmethod
    def single(cls, ctx, stream: np.ndarray, dev):
        import torch
        from . import api
        sc = api.scan(stream)
        d_in = torch.from_numpy(np.concatenate([stream, np.zeros(16, np.uint8)]))
        if dev is not None:
            d_in = d_in.to(dev)
        return cls(ctx, stream, d_in, sc, [list(range(sc.n_blocks))], 0)

    @classmethod
    def from_parts(cls, ctx, part: np.ndarray, dist, dev, coll_dev=None):
        """`part` = the consecutive piece of the shared stream this rank holds (whole blocks).  The pieces are
        all-gathered (RCCL on the GPU box), rank 0 scans the result and broadcasts the table."""
        import ctypes as C

        import torch
        from . import _lib, api
        world, rank = dist.get_world_size(), dist.get_rank()
        lens = all_gather_table(np.array([part.size]), dist, coll_dev).reshape(-1)
        width = int(lens.max())
        t = torch.zeros(width, dtype=torch.uint8)
        t[:part.size] = torch.from_numpy(part)
        if coll_dev is not None:
            t = t.to(coll_dev)
        outs = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(outs, t)
        h_stream = np.concatenate([o[:int(n)].cpu().numpy() for o, n in zip(outs, lens)])
        del outs, t
        # the block work table: scanned once, on rank 0, then broadcast (zpaqhip_block / zpaqhip_segment records)
        if rank == 0:
            sc0 = api.scan(h_stream)
            payload = (np.array([sc0.n_blocks, sc0.n_segments], np.int64).tobytes()
                       + bytes(sc0.blocks) + bytes(sc0.segments))
        else:
            payload = None
        raw = _bcast_bytes(payload, dist, 0, coll_dev)
        nb, ns = (int(x) for x in np.frombuffer(raw[:16], np.int64))
        blocks = (_lib.Block * max(1, nb)).from_buffer_copy(raw[16:16 + nb * C.sizeof(_lib.Block)].ljust(C.sizeof(_lib.Block), b"\0"))
        o = 16 + nb * C.sizeof(_lib.Block)
        segs = (_lib.Segment * max(1, ns)).from_buffer_copy(raw[o:o + ns * C.sizeof(_lib.Segment)].ljust(C.sizeof(_lib.Segment), b"\0"))
        sc = api.ScanResult((_lib.Block * nb).from_buffer(blocks), (_lib.Segment * ns).from_buffer(segs))
        plan = lpt_assign(cls._weights(sc), world)
        d_in = torch.from_numpy(np.concatenate([h_stream, np.zeros(16, np.uint8)]))
        if dev is not None:                                  # dev None: CPU-only ranks (tests of the table / plan / gather logic)
            d_in = d_in.to(dev)
        return cls(ctx, h_stream, d_in, sc, plan, rank, dist, coll_dev)

    def decode(self, d_out, out_off, out_cap, decode_fn=None, **opt) -> np.ndarray:
        """Decode this rank's shard into `d_out` (a device tensor); returns the [n_blocks, 2] table {status, out_len}
        of every block of the stream, gathered from all ranks.  `decode_fn(ids) -> [len(ids), 2]` replaces the HIP decode
        on ranks without a GPU (tests only)."""
        mine = self.shard
        local = np.zeros((len(mine), 2), np.int64)
        if mine and decode_fn is not None:
            local[:] = np.asarray(decode_fn(mine), np.int64).reshape(len(mine), 2)
        elif mine:
            rc, res = self.ctx.decode_blocks_device(self.d_in.data_ptr(), self.stream_len, self.sc, d_out.data_ptr(), out_off, out_cap,
                                                    ids=mine, h_in=self.h_stream, raise_on_error=False, **opt)
            for j, b in enumerate(mine):
                blk = self.sc.blocks[b]
                segs = [res[blk.first_seg + s] for s in range(blk.n_seg)]
                local[j, 0] = next((int(r.status) for r in segs if r.status != 0), 0)
                local[j, 1] = sum(int(r.out_len) for r in segs)
        n = self.sc.n_blocks
        if self.dist is None:
            table = np.zeros((n, 2), np.int64)
            table[mine] = local
            return table
        width = max(len(s) for s in self.plan)
        pad = np.full((width, 2), -1, np.int64)
        pad[:len(mine)] = local
        allres = all_gather_table(pad, self.dist, self.coll_dev)
        table = np.full((n, 2), -1, np.int64)
        for r, ids in enumerate(self.plan):
            for j, b in enumerate(ids):
                table[b] = allres[r, j]
        return table
