"""All-core CPU figure asked for by SURVEY.md §8d (ii): the C oracle, one block per thread, on the bench host.
Not collected by pytest.  Usage: python tests/cpu_allcores.py [threads] [blocks]"""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle  # noqa: E402
from zpaqsharp_amd import models, synth  # noqa: E402

threads = int(sys.argv[1]) if len(sys.argv) > 1 else (os.cpu_count() or 1)
blocks = int(sys.argv[2]) if len(sys.argv) > 2 else threads
bs = 4 << 20
stream, offs = synth.stream(models.get("l1"), "T", blocks, bs, threads=min(32, threads))
pieces = [stream[int(offs[b]):int(offs[b + 1])].tobytes() for b in range(blocks)]
done = [0] * threads


def work(t):
    for b in range(t, blocks, threads):
        out = oracle.decompress(pieces[b], cap=bs + 16)
        assert len(out) == bs
        done[t] += 1


t0 = time.perf_counter()
th = [threading.Thread(target=work, args=(t,)) for t in range(threads)]
[x.start() for x in th]
[x.join() for x in th]
dt = time.perf_counter() - t0
print(f"{threads} threads, {blocks} x 4 MiB L1 blocks: {blocks * bs / dt / 1e6:.1f} MB/s ({dt:.2f} s), host has {os.cpu_count()} logical cores")
