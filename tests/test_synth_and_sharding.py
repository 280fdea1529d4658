"""Product CPU stream writer (libzpaqgen) and the multi-rank sharding logic (gloo, CPU)."""
import hashlib
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle
from tests import util
from zpaqsharp_amd import multigpu, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("model", ["l1", "min", "mid", "max", "max+e8e9"])
def test_two_independent_encoders_agree_byte_for_byte(model):
    data = util.x86ish(6000) if "e8e9" in model else util.text(6000)
    a = synth.compress_block(model, data)          # zpaqsharp_amd/gen/zpaqgen.cpp
    b = util.block(model, data)                    # oracle/zpaq_oracle.c encoder mirror
    assert a == b and oracle.decompress(a) == data


def test_generators_are_deterministic_and_distinct():
    for kind in "TXR":
        a, b = synth.plain(kind, 5, 10000), synth.plain(kind, 5, 10000)
        assert np.array_equal(a, b) and not np.array_equal(a, synth.plain(kind, 6, 10000))
    assert np.array_equal(synth.plain("T", 5, 4000), synth.plain("T", 5, 10000)[:4000])
    x = synth.plain("X", 0, 1 << 16)
    assert ((x == 0xE8) | (x == 0xE9)).sum() > 1000


def test_stream_of_blocks_round_trips_on_the_oracle():
    s, offs = synth.stream("l1", "T", nblocks=5, block_size=20000, first_block=3, threads=3)
    assert len(offs) == 6 and offs[-1] == s.size
    want = np.concatenate([synth.plain("T", 3 + i, 20000) for i in range(5)]).tobytes()
    assert oracle.decompress(s.tobytes()) == want
    one = synth.stream("l1", "T", nblocks=1, block_size=20000, first_block=4, threads=1)[0]
    assert np.array_equal(one, s[int(offs[1]):int(offs[2])])       # independent of thread count / position


def test_forward_e8e9_matches_oracle():
    x = synth.plain("X", 1, 50000)
    assert synth.e8e9(x).tobytes() == oracle.e8e9(x.tobytes())


def test_lpt_assignment_is_balanced_and_deterministic():
    w = [100, 90, 80, 10, 10, 10, 10, 10, 5, 5]
    sh = multigpu.lpt_assign(w, 3)
    assert sorted(sum(sh, [])) == list(range(10))
    loads = [sum(w[i] for i in s) for s in sh]
    assert max(loads) - min(loads) <= 20 and sh == multigpu.lpt_assign(w, 3)
    assert multigpu.interleave_assign(7, 3) == [[0, 3, 6], [1, 4], [2, 5]]


def test_queue_chunks_are_cross_sections_and_the_queue_hands_every_block_out_once():
    import random
    import threading
    rnd = random.Random(4)
    costs = [rnd.choice((170, 1600, 3100)) * rnd.randint(1000, 5000) for _ in range(1000)]
    chunks = multigpu.queue_chunks(costs, 256)
    assert len(chunks) == 4 and sorted(sum(chunks, [])) == list(range(1000)) and all(c == sorted(c) for c in chunks)
    tot = [sum(costs[i] for i in c) for c in chunks]
    assert max(tot) - min(tot) <= max(costs)                 # chunk totals differ by less than one block
    assert multigpu.queue_chunks([], 256) == [] and multigpu.queue_chunks([5], 256) == [[0]]
    # many pullers, one counter: every chunk goes to exactly one of them
    q = multigpu.WorkQueue(costs, multigpu.LocalCounter(), 10)
    got = [[] for _ in range(8)]

    def pull(i):
        while (c := q.pull()) is not None:
            got[i].append(c)
    th = [threading.Thread(target=pull, args=(i,)) for i in range(8)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert sorted(b for g in got for c in g for b in c) == list(range(1000))
    assert sum(len(g) for g in got) == 100


def test_block_costs_weigh_plaintext_size_and_model_depth():
    import zpaqsharp_amd as z
    from tests import util
    d = util.text(30000, seed=3)
    s = (util.block("l1", d) + util.block("mid", d) + util.block("max", d) + util.block("mid", d[:10000])
         + util.block("mid", d, comment=b"no size here"))
    sc = z.scan(s)
    c = [int(x) for x in z.block_costs(s, sc)]
    assert c[0] < c[1] < c[2]                                # same plaintext: deeper model, higher cost
    assert c[1] == 3 * c[3]                                  # same model: cost follows the plaintext size
    assert c[0] == 30000 * 1100 and c[1] == 30000 * 9200 and c[2] == 30000 * 17300
    coded = int(sc.segments[sc.blocks[4].first_seg].data_len)
    assert c[4] == 4 * coded * 9200                          # no size in the comment: 4 x coded bytes
    plan = multigpu.lpt_assign(c, 2)
    loads = [sum(c[i] for i in sh) for sh in plan]
    assert max(loads) <= 0.6 * sum(c)                        # (by coded bytes alone the max block would be paired with mid)


_WORKER = r'''
import hashlib, os, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
import torch.distributed as dist
import oracle
import zpaqsharp_amd as z
from zpaqsharp_amd import multigpu, synth
dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
stream, offs = synth.stream("l1", "T", nblocks=7, block_size=3000, threads=1)
sc = z.scan(stream)
weights = [sc.segments[b.first_seg].data_len for b in sc.blocks]
def decode(ids):      # CPU stand-in for Context.decode_blocks_device (tests only)
    out = []
    for b in ids:
        plain = oracle.decompress(stream[int(offs[b]):int(offs[b + 1])].tobytes())
        out.append([0, len(plain), int.from_bytes(hashlib.sha1(plain).digest()[:7], "big")])
    return np.array(out, dtype=np.int64).reshape(len(ids), 3)
table, shards = multigpu.sharded_decode(weights, decode, dist)
want = [int.from_bytes(hashlib.sha1(synth.plain("T", b, 3000).tobytes()).digest()[:7], "big") for b in range(7)]
assert table.shape == (7, 3) and list(table[:, 0]) == [0] * 7 and list(table[:, 1]) == [3000] * 7
assert list(table[:, 2]) == want
assert sorted(sum(shards, [])) == list(range(7)) and all(len(s) >= 3 for s in shards)
bt = multigpu.broadcast_table(np.array(weights if dist.get_rank() == 0 else [0] * 7), dist)
assert list(bt) == weights
# the product's job object: each rank contributes its piece of the shared stream, rank 0 scans, the table is broadcast
rank = dist.get_rank()
cut = int(offs[4])
part = stream[:cut] if rank == 0 else stream[cut:]
job = multigpu.ShardedJob.from_parts(None, part, dist, None)
assert job.stream_len == stream.size and np.array_equal(job.h_stream, stream)
assert job.sc.n_blocks == 7 and [int(b.tag_off) for b in job.sc.blocks] == [int(b.tag_off) for b in sc.blocks]
costs = [int(x) for x in z.block_costs(stream, sc)]
assert costs == [3000 * 1100] * 7
assert job.plan == multigpu.lpt_assign(costs, 2) and job.shard == job.plan[rank]
def fake(ids):        # CPU stand-in for the HIP decode of a shard
    return [[0, len(oracle.decompress(stream[int(offs[b]):int(offs[b + 1])].tobytes()))] for b in ids]
t2 = job.decode(None, None, None, decode_fn=fake)
assert t2.shape == (7, 2) and list(t2[:, 0]) == [0] * 7 and list(t2[:, 1]) == [3000] * 7
# the dynamic form: both ranks pull chunks of 2 blocks from the queue behind the job's store
for _ in range(2):                                          # (a second pass uses a fresh counter key)
    t3 = job.decode_dynamic(None, None, None, queue_blocks=2, decode_fn=fake)
    assert t3.shape == (7, 3) and list(t3[:, 0]) == [0] * 7 and list(t3[:, 1]) == [3000] * 7
    took = multigpu.all_gather_table(np.array([len(job.pulled)]), dist).reshape(-1)
    assert int(took.sum()) == 4 and set(t3[:, 2]) <= {0, 1}
    for ids in job.pulled:
        assert all(int(t3[b, 2]) == rank for b in ids)
dist.destroy_process_group()
print("rank ok")
'''


def test_two_rank_sharded_decode_over_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29531", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script), ROOT], env=dict(env, RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=300)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert all("rank ok" in o for o in outs)
