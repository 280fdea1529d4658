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
assert job.plan == multigpu.lpt_assign(weights, 2) and job.shard == job.plan[rank]
def fake(ids):        # CPU stand-in for the HIP decode of a shard
    return [[0, len(oracle.decompress(stream[int(offs[b]):int(offs[b + 1])].tobytes()))] for b in ids]
t2 = job.decode(None, None, None, decode_fn=fake)
assert t2.shape == (7, 2) and list(t2[:, 0]) == [0] * 7 and list(t2[:, 1]) == [3000] * 7
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
