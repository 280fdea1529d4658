#!/bin/bash
# two blocks per workgroup: wavefront order A0 A1 B0 B1 C0 C1 (H1) against A0 B0 C0 A1 B1 C1 (H0), 1 024 x 1 MiB text / x86-like, same box
mkdir -p gpurun_out/r04
cp build/ab/libH1.so zpaqsharp_amd/libzpaqhip.so
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "two_wave_cm or more_single_cm" > gpurun_out/r04/ab30_tests.log 2>&1; tail -1 gpurun_out/r04/ab30_tests.log
grep -q passed gpurun_out/r04/ab30_tests.log && ! grep -q failed gpurun_out/r04/ab30_tests.log || exit 1
timeout -k 10 200 python tests/fuzz_l1.py 16 32 6 | tail -1
for v in H0 H1 H0 H1; do
  cp build/ab/lib$v.so zpaqsharp_amd/libzpaqhip.so
  python3 - $v <<'PY'
import os, sys, time, numpy as np
sys.path.insert(0, os.getcwd())
import zpaqsharp_amd as z
from zpaqsharp_amd import synth
ctx = z.Context(0)
for kind in "TX":
    nb, bs = 1024, 1 << 20
    s, _ = synth.stream("l1", kind, nb, bs)
    ctx.decompress(s, out_cap=nb * bs, kernel=0)
    out = ctx.decompress(s, out_cap=nb * bs, kernel=0); st = ctx.stats()
    ok = all(np.array_equal(out[b * bs:(b + 1) * bs], synth.plain(kind, b, bs)) for b in range(0, nb, 16))
    print(sys.argv[1], kind, "1024 x 1 MiB two per workgroup: kernel %.1f ms = %.1f MB/s, exact=%s" % (st.kernel_ms, nb * bs / st.kernel_ms / 1e3, ok), flush=True)
PY
done | tee gpurun_out/r04/ab30.txt
cp build/ab/libH1.so zpaqsharp_amd/libzpaqhip.so
