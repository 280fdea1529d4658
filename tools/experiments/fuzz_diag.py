"""For every stream under tools/experiments/fuzz_fail/ (copy gpurun_out/fuzz_fail_*.bin there): which block decodes differently by the host's kernel choice (kernel=0) and by
the lane-per-component kernel (kernel=4), and where the outputs first differ."""
import glob, hashlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import zpaqsharp_amd as z
ctx = z.Context(0)
for f in sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "fuzz_fail", "*.bin"))):
    s = np.fromfile(f, np.uint8)
    sc = z.scan(s)
    print(os.path.basename(f), sc.n_blocks, "blocks", flush=True)
    for i in range(sc.n_blocks):
        b = sc.blocks[i]
        one = s[b.tag_off:b.end_off].copy()
        seg = sc.segments[b.first_seg]
        res = {}
        for kern in (0, 4):
            try:
                out = ctx.decompress(one, kernel=kern, out_cap=int(b.usize_hint) + 64).tobytes()
                res[kern] = (hashlib.sha1(out).digest() == bytes(seg.sha1), out)
            except Exception as e:
                res[kern] = (False, repr(e).encode())
        if not (res[0][0] and res[4][0]):
            a, c = res[0][1], res[4][1]
            n = min(len(a), len(c))
            first = next((k for k in range(n) if a[k] != c[k]), n)
            print(f"  block {i}: n_comp {b.n_comp} hh {b.hh} hm {b.hm} ph {b.ph} pm {b.pm} size {b.usize_hint}: kernel0 ok={res[0][0]} len {len(a)}, kernel4 ok={res[4][0]} len {len(c)}, first diff {first}", flush=True)
            lo = max(0, first - 24)
            print("     k0:", a[lo:first + 24]); print("     k4:", c[lo:first + 24], flush=True)
            # again in the same stream position? (alone it may pass: then it is an interaction between blocks of one launch)
print("done")
