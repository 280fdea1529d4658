#!/usr/bin/env python3
"""bench.py — decompress MB/s (bit-exact) on a multi-block ZPAQ stream, 1..8 MI355X.

A "step" is one full decode pass (the hot path: arithmetic decode + component
chain predict/update + ZPAQL HCOMP/PCOMP) over the rank's resident blocks.  The
compressed stream is already in HBM when the timed region starts and the
plaintext stays in HBM (PCIe-inclusive rates are in DESIGN.md, never `value`).

Workload (BASELINE.json configs[1], the configuration the metric is quoted on):
256 x 4 MiB independent blocks per GPU, level-1 model (single direct order-1 CM),
text-like synthetic plaintext "T".  Weak scaling: N GPUs decode N x 256 blocks;
blocks are independent, so the only communication is the block work table and
the per-block results (RCCL all_gather of a few KiB).

Launch:  python bench.py [--gpus N --steps K --warmup W]
  N>1:   python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
             --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# Algorithmic HBM bytes per decoded plaintext byte, excluding stream I/O
# (derivation repeated in DESIGN.md §5).
B_ALG = {"l1": 64, "min": 128, "mid": 842, "max": 3114}
HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
HBM_COPY_GBPS = 6290.0          # same guide: measured copy peak (SURVEY.md §8d asks for both fractions)


def pmc_traffic(model, nb, bs):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/<round>/*_pmc_{FETCH,WRITE}_SIZE.csv, produced by tools/profile_bench.sh for this
    exact workload): 2 x FETCH_SIZE (gfx950 counts wide coalesced reads at half) + WRITE_SIZE, KB -> B.
    None when no matching profile is committed."""
    import csv
    import glob
    tag = f"{model}_{nb}x{bs >> 20}MiB"
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*", f"*_{tag}_pmc_FETCH_SIZE.csv"))):
        w = f.replace("FETCH_SIZE", "WRITE_SIZE")
        if not os.path.exists(w):
            continue
        try:
            fe = float(next(csv.DictReader(open(f)))["value"])
            wr = float(next(csv.DictReader(open(w)))["value"])
        except (StopIteration, KeyError, ValueError):
            continue
        best = {"bytes": (2 * fe + wr) * 1024.0, "source": os.path.relpath(f, ROOT)}
    return best


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--model", default="l1", help="l1 | min | mid | max | max+e8e9 (bench line = l1)")
    ap.add_argument("--kind", default=None, help="plaintext generator T | X | R (default T, X for +e8e9)")
    ap.add_argument("--blocks", type=int, default=256, help="blocks per GPU")
    ap.add_argument("--block-bytes", type=int, default=4 << 20)
    ap.add_argument("--cpu-sample-blocks", type=int, default=None,
                    help="blocks decoded by the CPU oracle for cpu_baseline (default: ~15 s of work)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--gen-threads", type=int, default=None)
    ap.add_argument("--kernel", type=int, default=0, help="zpaqhip_opts.kernel (0 auto; 5 = round-1 chain kernels for min/mid/max)")
    ap.add_argument("--cache-dir", default=None, help="keep generated streams here and reuse them (profiling runs)")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (RCCL; the real thing) | gloo (rehearsal of the N>1 path)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="rehearsal only: ranks share the visible GPUs round-robin (needs --dist-backend gloo)")
    return ap.parse_args()


def main():
    args = parse_args()
    import torch
    import torch.distributed as dist

    import zpaqsharp_amd as z
    from zpaqsharp_amd import models, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the ZPAQ decode path has no CPU fallback")
    if args.share_gpu:
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    coll_dev = dev if args.dist_backend == "nccl" else None     # gloo moves CPU tensors
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.dist_backend, rank=rank, world_size=world)

    model_name = args.model
    base = model_name.split("+")[0]
    kind = args.kind or ("X" if "+e8e9" in model_name else "T")
    nb, bs = args.blocks, args.block_bytes
    model = models.get(model_name)

    # ---- this rank's shard of the job: global blocks [rank*nb, (rank+1)*nb)
    ncpu = os.cpu_count() or 1
    gen_threads = args.gen_threads or max(1, min(32, ncpu // max(1, min(world, 8))))
    t0 = time.time()
    cache = os.path.join(args.cache_dir, f"{model_name}_{kind}_{nb}x{bs}_r{rank}.npz") if args.cache_dir else None
    if cache and os.path.exists(cache):
        with np.load(cache) as f:
            stream, offs = f["stream"], f["offs"]
    else:
        stream, offs = synth.stream(model, kind, nb, bs, first_block=rank * nb, threads=gen_threads)
        if cache:
            os.makedirs(args.cache_dir, exist_ok=True)
            np.savez(cache, stream=stream, offs=offs)
    gen_s = time.time() - t0
    sc = z.scan(stream)
    assert sc.n_blocks == nb, (sc.n_blocks, nb)
    plain_bytes = nb * bs
    coded_bytes = int(sum(s.data_len for s in sc.segments))
    rho = coded_bytes / plain_bytes

    ctx = z.Context(local_rank)
    d_in = torch.from_numpy(stream).to(dev)
    d_out = torch.zeros(plain_bytes, dtype=torch.uint8, device=dev)
    out_off = [i * bs for i in range(nb)]
    out_cap = [bs] * nb

    # ---- work table over RCCL: every rank learns every block's weight; the plan is
    # static (a block is decoded where it is resident), so no payload moves.
    weights = np.array([sc.segments[b.first_seg].data_len for b in sc.blocks], dtype=np.int64)
    if world > 1:
        from zpaqsharp_amd import multigpu
        all_w = multigpu.all_gather_table(weights, dist, coll_dev)
        assert all_w.shape == (world, nb)

    def step():
        rc, res = ctx.decode_blocks_device(d_in.data_ptr(), stream.size, sc, d_out.data_ptr(), out_off, out_cap,
                                           h_in=stream, kernel=args.kernel)
        return res

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    kernel_ms = []
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
        kernel_ms.append(ctx.stats().kernel_ms)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev or "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    st = ctx.stats()

    # ---- bit-exact check of the last step's output against the generator's plaintext
    ok = all(r.status == 0 and r.out_len == bs for r in res)
    if not args.no_verify:
        got = d_out.cpu().numpy()
        for b in range(nb):
            exp = synth.plain(kind, rank * nb + b, bs)
            if not np.array_equal(got[b * bs:(b + 1) * bs], exp):
                ok = False
                break
    okv = np.array([int(ok), plain_bytes, coded_bytes], dtype=np.int64)
    if world > 1:
        from zpaqsharp_amd import multigpu
        allv = multigpu.all_gather_table(okv, dist, coll_dev)
        ok = bool(allv[:, 0].all())
        total_plain = int(allv[:, 1].sum())
    else:
        total_plain = plain_bytes

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    ms_per_step = elapsed / args.steps * 1e3
    value = total_plain * args.steps / elapsed / 1e6
    kms = float(np.mean(kernel_ms))
    b_alg = B_ALG[base] + 1 + rho
    achieved = b_alg * plain_bytes / (kms * 1e-3) / 1e9

    cpu = None
    if not args.no_cpu_baseline:
        import oracle
        # bounded sample of the SAME stream: first S blocks, one host thread.
        per_block_guess = bs / 9e6 if base == "l1" else bs / 1.5e6
        S = args.cpu_sample_blocks or max(1, min(nb, int(15.0 / per_block_guess)))
        sample = stream[:int(offs[S])].tobytes()
        t0 = time.perf_counter()
        out = oracle.decompress(sample, cap=S * bs + 16)
        dt = time.perf_counter() - t0
        assert len(out) == S * bs
        cpu = {"value": S * bs / dt / 1e6, "unit": "MB/s", "cores": 1, "kind": "port",
               "sample": f"first {S} of {nb} blocks ({S * bs >> 20} MiB plaintext) of the same stream, "
                         f"oracle/zpaq_oracle.c -O2, 1 thread, host has {ncpu} logical cores"}

    line = {
        "metric": "decompress MB/s (bit-exact) on 1 GiB multi-block stream",
        "value": value, "unit": "MB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u32", "data": "synthetic",
        "bit_exact": bool(ok),
        "config": {
            "workload": f"BASELINE configs[1]: {nb} x {bs >> 20} MiB independent blocks per GPU, "
                        f"model {model_name} ({model.n} component(s)), plaintext generator {kind}",
            "zpaq_model": model_name, "blocks_per_gpu": nb, "block_bytes": bs, "plaintext": kind,
            "coded_over_plain": round(rho, 4), "parallelism": f"blocks x{world} (no data-path collective)",
            "kernel_kind": int(st.kernel_kind), "blocks_in_flight": int(st.concurrent),
        },
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBPS,
                     "frac_of_measured_copy_peak": achieved / HBM_COPY_GBPS,
                     "traffic": (pmc_traffic(model_name, nb, bs) or {}).get("bytes"),
                     "traffic_source": (pmc_traffic(model_name, nb, bs) or {}).get("source"),
                     "algorithmic_bytes_per_launch": b_alg * plain_bytes,
                     "kernel_ms": kms, "alg_bytes_per_plain_byte": b_alg,
                     "note": "bit-serial chain bound by single-wave instruction issue, not by HBM; see DESIGN.md §4"},
        "cpu_baseline": cpu,
        "gen_seconds": round(gen_s, 1),
    }
    if not ok:
        line["value"] = 0.0
        line["error"] = "GPU output is not bit-exact"
    print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
