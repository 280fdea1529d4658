"""Multi-GPU sharding of independent ZPAQ blocks.

Blocks are independent units (every block resets model, coder and VM state:
Decompresser.cs:128-134, Decoder.cs:100-105), so the path shards with NO
data-path collective.  torch.distributed (RCCL over xGMI on the GPU box, gloo in
the CPU tests) carries only the block work table and the per-block results —
KiB-scale traffic:

    broadcast(block table)  ->  every rank derives the same cost-ordered queue
    ... each rank decodes blocks, no payload communication ...
    all_gather(per-block {status, out_len})  ->  every rank holds the result table

Who decodes what is either STATIC — longest-processing-time-first over the estimated block
costs (zpaqhip_block_costs: plaintext bytes x cycles per byte of the block's kernel) — or
DYNAMIC: the cost-ordered blocks are dealt into chunks of `queue_blocks` (default_queue_blocks: 256, one block per CU of
the GPU that takes the chunk), chunk k = every K-th block of that order, and every rank pulls the
next chunk from one shared counter whenever it has finished one (`WorkQueue`; the counter is an
atomic add on the job's rendezvous store, a few bytes per pull), so a GPU that is faster, or
whose chunks were cheaper than estimated, takes more of them.  Placement does not depend on who
decodes: the caller gives every block its output offset.

`ShardedJob` is the product entry point: it binds the plan to
Context.decode_blocks_device(ids = shard) on every rank (bench.py --gpus N and
the 2-rank single-GPU rehearsal in tests/ run exactly this).  `sharded_decode`
takes the decode as a function so that the table / plan / gather logic can also
be exercised on CPU-only ranks (gloo) in the `-m "not gpu"` tests.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence, Tuple

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


def default_queue_blocks(n_blocks: int, world: int, all_single_cm: bool = False) -> int:
    """Blocks per pull when the caller does not say: a chunk that fills a GPU (256: one block per CU; 512 where every block is a
    single-CM one, which run two per CU) — but never so large that a rank gets fewer than four pulls: 2 048 blocks on 8 ranks
    in chunks of 256 would be one chunk per rank, and the queue could rebalance nothing."""
    full = 512 if all_single_cm else 256
    if world <= 1:
        return full                                       # one taker: nothing to balance
    per4 = (n_blocks + 4 * world - 1) // (4 * world)
    return max(1, min(full, per4))


def queue_chunks(costs: Sequence[int], queue_blocks: int = 256) -> List[List[int]]:
    """The chunks of the dynamic queue (== zpaqhip_decompress_multi): blocks sorted by cost, descending, ties by index;
    K = ceil(n / queue_blocks) chunks; chunk k = every K-th block of that order starting at k, in stream order.
    Every chunk is a cross-section of the cost distribution, and the chunks' total costs differ by at most one
    block's cost per stride."""
    n = len(costs)
    if n == 0:
        return []
    order = sorted(range(n), key=lambda i: (-int(costs[i]), i))
    k_chunks = (n + queue_blocks - 1) // queue_blocks
    return [sorted(order[k::k_chunks]) for k in range(k_chunks)]


class LocalCounter:
    """fetch-and-add for ranks that are threads of one process (tests; zpaqhip_decompress_multi uses std::atomic)."""

    def __init__(self):
        import threading
        self._v, self._l = 0, threading.Lock()

    def fetch_add(self, n: int = 1) -> int:
        with self._l:
            v = self._v
            self._v += n
            return v


class StoreCounter:
    """fetch-and-add on the torch.distributed rendezvous store (TCPStore.add is atomic on the server): the shared head
    of the block work queue for one-process-per-GPU jobs.  `key` must be new for every pass over the queue."""

    def __init__(self, store, key: str):
        self.store, self.key = store, key

    def fetch_add(self, n: int = 1) -> int:
        return int(self.store.add(self.key, n)) - n


class WorkQueue:
    """Chunks of a cost-ordered block list behind one shared counter; `pull()` returns the next chunk's block ids or
    None when the queue is empty.  Every rank builds the same chunks from the same table, so only the counter is shared."""

    def __init__(self, costs: Sequence[int], counter, queue_blocks: int = 256):
        self.chunks = queue_chunks(costs, queue_blocks)
        self.counter = counter

    def pull(self) -> Optional[List[int]]:
        k = self.counter.fetch_add(1)
        return self.chunks[k] if k < len(self.chunks) else None


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

    def __init__(self, ctx, h_stream: np.ndarray, d_in, sc, plan, rank, dist=None, coll_dev=None, store=None):
        self.ctx, self.h_stream, self.d_in, self.sc = ctx, h_stream, d_in, sc
        self.stream_len = int(h_stream.size)
        self.plan, self.rank, self.dist, self.coll_dev = plan, rank, dist, coll_dev
        self.shard = plan[rank]
        self.costs = None                                   # filled on first use by decode_dynamic
        self._passes = 0
        self.store = store                                  # key-value store the ranks share (default: the process group's)
        self._job_id = None                                 # drawn by rank 0 on the first dynamic pass, broadcast to all
        self.pulled: List[List[int]] = []                   # chunks this rank took in the last dynamic pass
        self.kernel_ms = 0.0                                # kernel time of this rank's last pass (HIP events, summed over launches)

    @staticmethod
    def _weights(sc, h_stream=None):
        """Estimated decode cost per block (zpaqhip_block_costs); without the stream (table-only callers): plaintext
        size from the comment, else 4 x coded bytes."""
        if h_stream is not None:
            from . import api
            return [int(x) for x in api.block_costs(h_stream, sc)]
        out = []
        for b in sc.blocks:
            coded = sum(int(sc.segments[b.first_seg + s].data_len) for s in range(b.n_seg))
            out.append(int(b.usize_hint) if b.usize_hint != (1 << 64) - 1 and b.usize_hint <= 1 << 40 else 4 * coded)
        return out

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
    def from_parts(cls, ctx, part: np.ndarray, dist, dev, coll_dev=None, store=None):
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
        plan = lpt_assign(cls._weights(sc, h_stream), world)
        d_in = torch.from_numpy(np.concatenate([h_stream, np.zeros(16, np.uint8)]))
        if dev is not None:                                  # dev None: CPU-only ranks (tests of the table / plan / gather logic)
            d_in = d_in.to(dev)
        return cls(ctx, h_stream, d_in, sc, plan, rank, dist, coll_dev, store)

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
        if decode_fn is None and mine:
            self.kernel_ms = float(self.ctx.stats().kernel_ms)
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

    # ---- dynamic form: the ranks pull chunks from one shared queue instead of decoding a fixed shard
    def _decode_ids(self, ids, d_out, out_off, out_cap, decode_fn, opt):
        """Decode blocks `ids` to d_out + out_off[b] (offsets indexed by GLOBAL block id) -> [len(ids), 2]."""
        local = np.zeros((len(ids), 2), np.int64)
        if decode_fn is not None:
            local[:] = np.asarray(decode_fn(ids), np.int64).reshape(len(ids), 2)
            return local
        rc, res = self.ctx.decode_blocks_device(self.d_in.data_ptr(), self.stream_len, self.sc, d_out.data_ptr(),
                                                [out_off[b] for b in ids], [out_cap[b] for b in ids],
                                                ids=ids, h_in=self.h_stream, raise_on_error=False, **opt)
        self.kernel_ms += float(self.ctx.stats().kernel_ms)
        for j, b in enumerate(ids):
            blk = self.sc.blocks[b]
            segs = [res[blk.first_seg + s] for s in range(blk.n_seg)]
            local[j, 0] = next((int(r.status) for r in segs if r.status != 0), 0)
            local[j, 1] = sum(int(r.out_len) for r in segs)
        return local

    def _queue_store(self):
        if self.store is None:
            from torch.distributed import distributed_c10d as c10d
            self.store = c10d._get_default_store()          # (torch has no public accessor for the default group's store)
        return self.store

    def _queue_key(self) -> str:
        """Key of this pass's shared counter: unique per job AND per pass.  Two ShardedJobs on one process group must not
        share a counter (the second would start at the first one's final value and hand out no chunk at all): rank 0
        draws a job number from the store the first time, every rank receives it by broadcast."""
        if self._job_id is None:
            jid = np.zeros(1, np.int64)
            if self.rank == 0:
                jid[0] = int(self._queue_store().add("zpaqhip/jobs", 1))
            self._job_id = int(broadcast_table(jid, self.dist, 0, self.coll_dev)[0])
        return f"zpaqhip/queue/{self._job_id}/{self._passes}"

    def decode_dynamic(self, d_out, out_off, out_cap, counter=None, queue_blocks: int = 0, decode_fn=None, **opt) -> np.ndarray:
        """One pass over the whole stream with the ranks pulling chunks from the shared queue (module docstring).
        `out_off` / `out_cap` are indexed by GLOBAL block id and must be the same on every rank — a block lands at
        the same offset of whichever rank's `d_out` decodes it.  Returns the [n_blocks, 3] table {status, out_len,
        rank that decoded it}, gathered from all ranks.  `counter`: fetch_add provider shared by the ranks (default: a
        StoreCounter on the job's store under a key that is new for every job and pass, deleted after the gather)."""
        if self.costs is None:
            self.costs = self._weights(self.sc, self.h_stream)
        self._passes += 1
        own_key = None
        if counter is None:
            if self.dist is None:
                counter = LocalCounter()
            else:
                own_key = self._queue_key()
                counter = StoreCounter(self._queue_store(), own_key)
        n = self.sc.n_blocks
        if not queue_blocks:                                  # 0: default_queue_blocks (>= 4 pulls per rank)
            world = self.dist.get_world_size() if self.dist is not None else 1
            queue_blocks = default_queue_blocks(n, world, all(int(b.n_comp) == 1 for b in self.sc.blocks))
        self.queue_blocks = queue_blocks
        q = WorkQueue(self.costs, counter, queue_blocks)
        mine = np.full((n, 3), -1, np.int64)
        self.pulled, self.kernel_ms = [], 0.0
        while True:
            ids = q.pull()
            if ids is None:
                break
            self.pulled.append(ids)
            mine[ids, :2] = self._decode_ids(ids, d_out, out_off, out_cap, decode_fn, opt)
            mine[ids, 2] = self.rank
        if self.dist is None:
            return mine
        allres = all_gather_table(mine, self.dist, self.coll_dev)          # [world, n, 3]: each block filled by exactly one rank
        if own_key is not None and self.rank == 0:                          # every rank has left the queue: the counter can go
            try:
                self._queue_store().delete_key(own_key)
            except Exception:                                               # a store without delete_key keeps a few bytes
                pass
        owner = allres[:, :, 2].argmax(axis=0)                              # the rank whose row is not -1
        table = allres[owner, np.arange(n)]
        return table
