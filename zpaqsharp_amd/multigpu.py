"""Multi-GPU sharding of independent ZPAQ blocks.

Blocks are independent units (every block resets model, coder and VM state:
Decompresser.cs:128-134, Decoder.cs:100-105), so the path shards with NO
data-path collective.  torch.distributed (RCCL over xGMI on the GPU box, gloo in
the CPU tests) carries only the block work table and the per-block results —
KiB-scale traffic:

    all_gather(block weights)  ->  every rank computes the same LPT assignment
    ... each rank decodes its own blocks, no communication ...
    all_gather(per-block {status, out_len, checksum})  ->  rank 0 reports

The decode itself is injected (`decode_fn`) so that the sharding logic can be
exercised on CPU ranks in the tests; the product binds it to
Context.decode_blocks_device.
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
