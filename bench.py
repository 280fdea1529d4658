#!/usr/bin/env python3
"""bench.py — decompress MB/s (bit-exact) on a multi-block ZPAQ stream, 1..8 MI355X.

A "step" is one full decode pass (the hot path: arithmetic decode + component
chain predict/update + ZPAQL HCOMP/PCOMP) over the rank's shard of the blocks.
The compressed stream is already in HBM when the timed region starts and the
plaintext stays in HBM; the host-to-host (PCIe-inclusive) rate is reported
beside it as `host_to_host`, never as `value`.

Workload (BASELINE.json configs[1], the configuration the metric is quoted on):
256 x 4 MiB independent blocks per GPU, level-1 model (single direct order-1 CM),
text-like synthetic plaintext "T".  Weak scaling: N GPUs decode N x 256 blocks.

N > 1 (BASELINE configs[3] layout): ONE shared stream and ONE block table.  Every
rank writes 1/N of the blocks; the pieces are all-gathered into the shared stream
(every rank holds it in HBM, as every rank would read the same archive), rank 0
scans it and broadcasts the block/segment table, every rank derives the same
longest-first plan over the estimated block costs (multigpu.lpt_assign over
zpaqhip_block_costs) and decodes its shard with
zpaqhip_decode_blocks_device(ids = shard); per-block results are all-gathered
inside the timed region.  RCCL carries the table and the results only — no
payload moves during decode (blocks are independent units).  --schedule queue
replaces the fixed shards by the dynamic work queue (multigpu.WorkQueue: ranks
pull chunks of --queue-blocks blocks from one counter on the job's store).

Launch:  python bench.py [--gpus N --steps K --warmup W]
  N>1:   python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
             --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
Prints ONE JSON line on rank 0.
"""
import argparse
import glob
import hashlib
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# Algorithmic HBM bytes per decoded plaintext byte, excluding stream I/O
# (derivation repeated in DESIGN.md §4).
B_ALG = {"l1": 64, "min": 128, "mid": 842, "max": 3114}
HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
HBM_COPY_GBPS = 6290.0          # same guide: measured copy peak (SURVEY.md §8d asks for both fractions)
CLOCK_GHZ = 2.4                 # max shader clock (same guide)


# The device sources each hot kernel is compiled from (the .hip file and every header it includes): a PMC summary or an
# instruction count belongs to a kernel, and stays valid while THAT kernel's sources do
KERNEL_SOURCES = {
    "l1": ("zh_cm.hip", "zh_cm_fast.h", "zh_core.h", "zh_dev.h", "zh_model.h"),
    "chain2": ("zh_chain2.hip", "zh_c2_common.h", "zh_zpaql_native.h", "zh_core.h", "zh_dev.h", "zh_model.h"),
    "nibble": ("zh_nibble.hip", "zh_nb_fast.h", "zh_nb_fast_mid.h", "zh_c2_common.h", "zh_zpaql_native.h", "zh_core.h", "zh_dev.h", "zh_model.h"),
}


def source_hash(model=None):
    """Identifies the kernel a PMC summary or an instruction count was taken with: SHA-1 over the device sources of the
    decode kernel of `model` (l1 -> zh_cm.hip and its headers; min / mid -> zh_nibble.hip, max[+e8e9] -> zh_chain2.hip and theirs);
    model None: every device source of libzpaqhip (zpaqsharp_amd/csrc/*.hip, *.h)."""
    h = hashlib.sha1()
    d = os.path.join(ROOT, "zpaqsharp_amd", "csrc")
    if model is None:
        files = sorted(glob.glob(os.path.join(d, "*.h")) + glob.glob(os.path.join(d, "*.hip")))
    else:
        base = model.replace("_", "+").split("+")[0]
        files = [os.path.join(d, f) for f in sorted(KERNEL_SOURCES["l1" if base == "l1" else "nibble" if base in ("min", "mid") else "chain2"])]
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic(model, nb, bs):
    """HBM bytes per launch of the decode kernel from a committed rocprofv3 PMC summary
    (profiles/<round>/pmc_<model>_<nb>x<KiB>KiB.json, written by tools/profile_bench.sh): FETCH_SIZE + WRITE_SIZE as the
    counters report them (KB).  Only a summary taken with THESE sources counts; otherwise (None, reason)."""
    tag = f"{model.replace('+', '_')}_{nb}x{bs >> 10}KiB"
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "*", f"pmc_{tag}.json")))
    if not cands:
        return None, "no PMC summary committed for this workload"
    cur = source_hash(model)
    for f in reversed(cands):
        try:
            d = json.load(open(f))
        except (OSError, ValueError):
            continue
        if d.get("src_hash") == cur:
            return {"bytes": (float(d["FETCH_SIZE_KB"]) + float(d["WRITE_SIZE_KB"])) * 1024.0,
                    "source": os.path.relpath(f, ROOT)}, None
    return None, f"the committed PMC summaries were taken with other kernel sources than {cur}"


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
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the extra figures of the N=1 line (host_to_host, all-core CPU, configs[2] / [4] shaped runs)")
    ap.add_argument("--extras-block-bytes", type=int, default=4 << 20,
                    help="block size of the configs[2] / configs[4] shaped extra runs (256 blocks each; BASELINE size = 4194304)")
    ap.add_argument("--extras-distinct", type=int, default=64,
                    help="distinct blocks written for each extra run; the 256 blocks are these, repeated (blocks decode "
                         "independently, each in its own arena slot, so a repeated block costs what a distinct one costs; "
                         "writing 1 GiB with the max model takes the host cores minutes)")
    ap.add_argument("--schedule", default="lpt", help="N > 1: lpt (fixed shards, longest-first over estimated block costs) | "
                                                      "queue (ranks pull chunks from the shared work queue)")
    ap.add_argument("--queue-blocks", type=int, default=0, help="--schedule queue: blocks per pull (0: multigpu.default_queue_blocks)")
    ap.add_argument("--gen-threads", type=int, default=None)
    ap.add_argument("--kernel", type=int, default=0, help="zpaqhip_opts.kernel (0 auto; 5 = round-1 chain kernels for min/mid/max)")
    ap.add_argument("--cache-dir", default=None, help="keep generated streams here and reuse them (profiling runs)")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (RCCL; the real thing) | gloo (rehearsal of the N>1 path)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="rehearsal only: ranks share the visible GPUs round-robin (needs --dist-backend gloo)")
    return ap.parse_args()


def make_stream(synth, model, model_name, kind, nb, bs, first_block, threads, cache_dir):
    cache = os.path.join(cache_dir, f"{model_name}_{kind}_{nb}x{bs}_b{first_block}.npz") if cache_dir else None
    if cache and os.path.exists(cache):
        with np.load(cache) as f:
            return f["stream"], f["offs"]
    stream, offs = synth.stream(model, kind, nb, bs, first_block=first_block, threads=threads)
    if cache:
        os.makedirs(cache_dir, exist_ok=True)
        np.savez(cache, stream=stream, offs=offs)
    return stream, offs


def roofline(base_model, kms, plain_bytes, rho, nb, bs, model_tag):
    b_alg = B_ALG[base_model] + 1 + rho
    achieved = b_alg * plain_bytes / (kms * 1e-3) / 1e9
    tr, why = pmc_traffic(model_tag, nb, bs)
    waves = int(min(nb, 256))
    return {"bound": "issue",                 # one bit-serial wave (two for a single CM) per block: see "issue" and DESIGN.md §4
            "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
            "frac_of_measured_copy_peak": achieved / HBM_COPY_GBPS,
            "traffic": tr["bytes"] if tr else None, "traffic_source": tr["source"] if tr else None,
            "traffic_note": why or "FETCH_SIZE + WRITE_SIZE as the counters report them (KB); gfx950 may count 16-B/lane reads at half",
            "measured_hbm_gbps": (tr["bytes"] / (kms * 1e-3) / 1e9) if tr else None,
            "algorithmic_bytes_per_launch": b_alg * plain_bytes, "alg_bytes_per_plain_byte": b_alg,
            "kernel_ms": kms,
            "issue": issue_bound(base_model, kms, plain_bytes, waves)}


# Instructions the decoder wave of a block executes per plaintext byte: static counts of the gfx950 code objects taken by
# tools/count_instr.py (llvm-objdump -d of libzpaqhip.so; the byte loop from the EOS flag to the loop's backward branch),
# committed as profiles/<round>/instr_<kernel>.json with the hash of the sources they were counted on.  A count taken on
# other sources than the ones this run uses is refused, like a stale PMC summary.  A lone wavefront issues at most one
# instruction per 4 cycles (tools/ubench/salu_bench), which is the ceiling the measured cycles per byte are set against.
INSTR_KERNEL = {"l1": "zh_decode_cm", "min": "nb_fast_min", "mid": "nb_fast_mid", "max": "zh_decode_c2_max"}
ISSUE_CYCLES_PER_INSTR = 4


def decoder_instr_per_byte(base_model):
    """(count, source file) or (None, reason)."""
    sym = INSTR_KERNEL.get(base_model)
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "*", f"instr_{sym}.json"))) if sym else []
    if not cands:
        return None, "no instruction count committed for this kernel (tools/count_instr.py)"
    cur = source_hash(base_model)
    for f in reversed(cands):
        try:
            d = json.load(open(f))
        except (OSError, ValueError):
            continue
        if d.get("src_hash") == cur:
            return int(d["instr_per_byte_static"]), os.path.relpath(f, ROOT)
    return None, f"the committed instruction counts were taken on other kernel sources than {cur}"


def issue_bound(base_model, kms, plain_bytes, waves):
    cyc = kms * 1e-3 * CLOCK_GHZ * 1e9 / (plain_bytes / waves)
    n, src = decoder_instr_per_byte(base_model)
    return {"blocks_in_flight": waves,
            "cycles_per_plain_byte_per_block": cyc,
            "decoder_wave_instr_per_plain_byte": n, "instr_source": src,
            "ceiling_cycles_per_instr": ISSUE_CYCLES_PER_INSTR,
            "frac_of_issue_ceiling": (n * ISSUE_CYCLES_PER_INSTR / cyc) if n else None,
            "note": "bound = one wave's instruction chain per block: DESIGN.md 4"}


def method_streams(z, synth, ctx, kib=4096, blocks=256, threads=None):
    """The widened row: streams written with the reference's method strings whose whole cost is the
    post-processor — lazy2 / lzpre (LZ77, LibZPAQ.cs:427-639) and bwtrle (inverse BWT, :642-795) on unmodelled blocks:
    the wave-wide kernels of zh_store.hip.  256 DISTINCT blocks of 4 MiB, pre-processed by zpaqgen in the LZBuffer.cs:96-115
    formats; host buffer to host buffer, stored SHA-1 verified on the device, every block compared with its plaintext."""
    from tools import methods
    bs = kib << 10
    out = []
    # ... and (round 5) the MODELLED method strings real archives carry (LibZPAQ.compressBlock's choices, LibZPAQ.cs:196-260).  The
    # BWT model of level 3 (`ci1`) and the level-4 models (`ci1,1,1,1,2am`, text: `...2awm`) have min's / mid's component lists:
    # and so has level 3's LZ77 + CM model (`...,1c0,0,511i2`): zh_nibble.hip decodes them (256 blocks of 256 KiB here).  The level-5
    # recipe runs on the lane-per-component kernel of round 1 (zh_chain.hip) — 256 blocks of 64 KiB: at 3 MB/s the default run
    # must stay within minutes
    for mt, what, mkib in (("x2,1,4,0,3,22", "lazy2: bit-packed LZ77, no model", 0), ("x2,2,12,0,7,22", "lzpre: byte-aligned LZ77, no model", 0),
                           ("x3,3", "bwtrle: BWT, no model", 0),
                           ("x0,2,12,0,7,16,1c0,0,511i2", "level 3: lzpre + icm/isse over the parse state", 256),
                           ("x0,3ci1", "level 3 / 4 on text: BWT + icm/isse", 256),
                           ("x0,0ci1,1,1,1,2awm", "level 4 on text: icm/isse chain + match + word icm + mix", 256),
                           ("x0,4ci1,1,1,1,2am", "level 4 on binaries: E8E9 + icm/isse chain + match + mix", 256),
                           ("x0,0c0,0,255w1i1c256ci1,1,1,1,1,1,2ac0,2,0,255i1c0,3,0,0,255i1c0,4,0,0,0,255i1mm16ts19t0", "level 5's recipe", 64)):
        progress(f"method stream {mt[:24]}")
        model, margs = methods.model_of(mt)
        modelled = model.n > 0
        bs = (mkib << 10) if modelled else (kib << 10)
        s, _ = synth.method_stream(model, margs, "T", blocks, bs, threads=threads)
        got = ctx.decompress(s, out_cap=bs * blocks, verify_sha1=True)          # warm-up (arena allocation) + check
        ok = got.size == bs * blocks and all(np.array_equal(got[b * bs:(b + 1) * bs], synth.plain("T", b, bs)) for b in range(blocks))
        del got
        t0 = time.time()
        ctx.decompress(s, out_cap=bs * blocks)
        dt = time.time() - t0
        kms = float(ctx.stats().kernel_ms)
        # which decode kernel the host picks for this model, read off zpaqhip_block_costs (cycles per plaintext byte of the block's family)
        per_byte = int(z.block_costs(s, z.scan(s))[0]) // bs
        dev_kernel = {30: "zh_decode_store", 120: "zh_decode_store", 6800: "zh_decode_nb_mid", 7000: "zh_decode_nb_mid8", 3800: "zh_decode_nb_min",
                      5300: "zh_decode_nb_min", 8300: "zh_decode_nb_mid", 8500: "zh_decode_nb_mid8"}.get(per_byte, "zh_decode_chain")
        out.append({"method": mt[:24], "what": what, "kernel": int(ctx.stats().kernel_kind), "device_kernel": dev_kernel, "block_KiB": bs >> 10,
                    "workload": f"{blocks} x {bs >> 10} KiB distinct blocks, text-like plaintext, coded {s.size / 1e6:.0f} MB",
                    "value": (bs * blocks / dt / 1e6) if ok else 0.0, "unit": "MB/s (host to host)",
                    "kernel_ms": kms, "kernel_MBps": bs * blocks / (kms * 1e-3) / 1e6 if kms else None, "bit_exact": bool(ok)})
    return out


def resident_run(z, synth, torch, ctx, dev, model_name, kind, nb, bs, threads, cache_dir, distinct=None):
    """One GPU, one decode pass over a resident stream: (MB/s, kernel_ms, rho, bit_exact, stats, stream, scan).  `distinct` < nb: only
    that many blocks are written and the stream repeats them (see --extras-distinct)."""
    from zpaqsharp_amd import models
    nd = min(nb, distinct or nb)
    part, _ = make_stream(synth, models.get(model_name), model_name, kind, nd, bs, 0, threads, cache_dir)
    stream = np.concatenate([part] * (nb // nd) + [part[:0]]) if nd < nb and nb % nd == 0 else part
    if stream is part and nd < nb:
        raise SystemExit("--extras-distinct must divide the block count")
    sc = z.scan(stream)
    assert sc.n_blocks == nb
    d_in = torch.from_numpy(np.concatenate([stream, np.zeros(16, np.uint8)])).to(dev)
    d_out = torch.zeros(nb * bs, dtype=torch.uint8, device=dev)
    off, cap = [i * bs for i in range(nb)], [bs] * nb
    # untimed: as many blocks of 1 KiB with the same model — the call allocates the arena (tens of GB for the max model)
    # and loads the code object
    warm, _ = make_stream(synth, models.get(model_name), model_name, kind, nb, 1024, 0, threads, cache_dir)
    ctx.decompress(warm)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rc, res = ctx.decode_blocks_device(d_in.data_ptr(), stream.size, sc, d_out.data_ptr(), off, cap, h_in=stream)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st = ctx.stats()
    ok = all(r.status == 0 and r.out_len == bs for r in res)
    if ok:
        got = d_out.cpu().numpy()
        ok = all(np.array_equal(got[b * bs:(b + 1) * bs], synth.plain(kind, b % nd, bs)) for b in range(nb))
    rho = sum(s.data_len for s in sc.segments) / (nb * bs)
    return nb * bs / dt / 1e6, float(st.kernel_ms), rho, ok, st, stream, sc


def cpu_all_cores(oracle, h_stream, block_start, n_glob, bs, per, threads):
    """The C oracle on `threads` host threads, `per` consecutive blocks each (one oracle instance per thread; ctypes
    releases the GIL): the all-core CPU figure SURVEY.md 8d asks for beside the 1-thread baseline."""
    pieces = []
    for i in range(threads):
        b0 = (i * per) % max(1, n_glob - per + 1)
        pieces.append(h_stream[block_start[b0]:block_start[b0 + per]].tobytes())
    done = [0] * threads

    def work(i):
        done[i] = len(oracle.decompress(pieces[i], cap=per * bs + 16))
    th = [threading.Thread(target=work, args=(i,)) for i in range(threads)]
    t0 = time.perf_counter()
    [t.start() for t in th]
    [t.join() for t in th]
    dt = time.perf_counter() - t0
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = os.cpu_count() or 1
    return {"value": sum(done) / dt / 1e6, "unit": "MB/s", "threads": threads, "host_logical_cores": os.cpu_count(),
            "cores_this_process_may_use": usable,
            "sample": f"{threads} threads x {per} block(s) of the same stream ({sum(done) >> 20} MiB plaintext), one oracle instance per thread"}


def cpu_share():
    """What the host lets this process use: logical cores, affinity, and the cgroup CPU quota (cores' worth) if one is set —
    a thread sweep past that number measures the quota, not the oracle."""
    out = {"host_logical_cores": os.cpu_count()}
    try:
        out["affinity_cores"] = len(os.sched_getaffinity(0))
    except AttributeError:
        out["affinity_cores"] = os.cpu_count()
    out["cgroup_cpu_quota_cores"] = None
    for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(f).read().split()
            if f.endswith("cpu.max"):
                if txt[0] != "max":
                    out["cgroup_cpu_quota_cores"] = round(int(txt[0]) / int(txt[1]), 2)
            else:
                q = int(txt[0])
                if q > 0:
                    out["cgroup_cpu_quota_cores"] = round(q / int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read()), 2)
            break
        except (OSError, ValueError, IndexError):
            continue
    return out


def cpu_sweep(oracle, pieces, plain_each, counts):
    """The C oracle on T host threads for every T in `counts`, thread i decoding pieces[i % len(pieces)] (whole blocks;
    one oracle instance per thread, ctypes releases the GIL) -> [{threads, value MB/s, seconds}].  VERDICT r03: "does
    not scale further" has to be a measurement."""
    pts = []
    for T in counts:
        done = [0] * T

        def work(i):
            done[i] = len(oracle.decompress(pieces[i % len(pieces)], cap=plain_each + 16))
        th = [threading.Thread(target=work, args=(i,)) for i in range(T)]
        t0 = time.perf_counter()
        [t.start() for t in th]
        [t.join() for t in th]
        dt = time.perf_counter() - t0
        pts.append({"threads": T, "value": round(sum(done) / dt / 1e6, 2), "unit": "MB/s", "seconds": round(dt, 2),
                    "bytes_per_thread": plain_each})
    return pts


def sweep_counts(share):
    """8 threads up to every core the process may run on — or, where a cgroup CPU quota is set, up to four times the
    quota (past that the sweep only measures the quota: the r04 boxes give 16 cores' worth of 256, L1 peaks at 32
    threads and mid at 16, and 256 threads of the mid model take a minute to say so)."""
    top = share["affinity_cores"] or share["host_logical_cores"] or 1
    q = share.get("cgroup_cpu_quota_cores")
    if q:
        top = min(top, max(32, int(4 * q)))
    return sorted({c for c in (8, 16, 32, 64, 128, top) if c <= top})


_T0 = time.time()


def progress(msg):
    """one line on stderr per stage (a run that says nothing for minutes looks hung to whoever watches it)"""
    print(f"[bench {time.time() - _T0:6.1f} s] {msg}", file=sys.stderr, flush=True)


def _r(x, n=2):
    return round(x, n) if isinstance(x, float) else x


def compact_line(line):
    """The ONE line bench.py prints, kept under 8 KB (the driver records the last 8 KB of the output): the contract's keys in full,
    roofline and cpu_baseline without their prose, and every other measured configuration as one short record — the
    per-config summary comes LAST.  The unabridged line is written to gpurun_out/bench_full.json."""
    def roof(r):
        iss = r.get("issue") or {}
        return {"bound": r["bound"], "achieved": _r(r["achieved"], 3), "peak": r["peak"], "unit": r["unit"], "frac": _r(r["frac"], 5),
                "traffic": r.get("traffic"), "kernel_ms": _r(r["kernel_ms"], 1),
                "cycles_per_byte": _r(iss.get("cycles_per_plain_byte_per_block"), 0), "instr_per_byte": iss.get("decoder_wave_instr_per_plain_byte"),
                "frac_of_issue_ceiling": _r(iss.get("frac_of_issue_ceiling"), 3)}

    def cpu(c):
        return None if not c else {"value": _r(c["value"]), "unit": c["unit"], "cores": c.get("cores", c.get("threads")), "kind": c.get("kind", "port"),
                                   "sample": c["sample"][:110]}
    out = {k: line[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                                "dtype", "data", "bit_exact") if k in line}
    if "error" in line:
        out["error"] = line["error"]
    cfg = dict(line["config"])
    cfg["workload"] = cfg["workload"][:200]
    cfg["parallelism"] = cfg["parallelism"][:120]
    out["config"] = cfg
    out["roofline"] = roof(line["roofline"])
    out["roofline"]["traffic_note"] = (line["roofline"].get("traffic_source") or line["roofline"].get("traffic_note") or "")[:90]
    out["cpu_baseline"] = cpu(line.get("cpu_baseline"))
    if "value_host_to_host" in line:
        out["value_host_to_host"] = _r(line["value_host_to_host"])
        h = line["host_to_host"]
        out["host_to_host"] = {k: _r(h[k]) for k in ("value", "h2d_ms", "kernel_ms", "d2h_ms", "bit_exact")}
    if "cpu_all_cores" in line:
        out["cpu_all_cores"] = {"value": _r(line["cpu_all_cores"]["value"]), "threads": line["cpu_all_cores"]["threads"]}
        out["cpu_all_cores_sweep"] = [[p["threads"], p["value"]] for p in line.get("cpu_all_cores_sweep", [])]
        sh = line.get("cpu_share", {})
        out["cpu_share"] = [sh.get("host_logical_cores"), sh.get("affinity_cores"), sh.get("cgroup_cpu_quota_cores")]
    if "method_streams" in line:
        out["method_streams"] = [{"method": m["method"], "kernel": m.get("device_kernel", m.get("kernel")), "KiB": m.get("block_KiB"), "value": _r(m["value"], 1),
                                  "kernel_MBps": _r(m["kernel_MBps"], 1), "bit_exact": m["bit_exact"]} for m in line["method_streams"]]
    if "other_configs" in line:                                  # last: one record per BASELINE-shaped configuration
        oc = []
        for r in line["other_configs"]:
            rec = {"config": r["config"][:40], "workload": r["workload"][:60], "value": _r(r["value"]), "unit": "MB/s", "bit_exact": r["bit_exact"],
                   "roofline": roof(r["roofline"])}
            if r.get("cpu_baseline"):
                rec["cpu_baseline"] = {"value": _r(r["cpu_baseline"]["value"]), "cores": 1, "kind": "port"}
            if r.get("cpu_all_cores"):
                rec["cpu_all_cores"] = [_r(r["cpu_all_cores"]["value"]), r["cpu_all_cores"]["threads"]]
            if r.get("cpu_all_cores_sweep"):
                rec["cpu_sweep"] = [[p["threads"], p["value"]] for p in r["cpu_all_cores_sweep"]]
            oc.append(rec)
        out["other_configs"] = oc
    return out


def main():
    args = parse_args()
    import torch
    import torch.distributed as dist

    import zpaqsharp_amd as z
    from zpaqsharp_amd import models, multigpu, synth

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
    ncpu = os.cpu_count() or 1
    try:
        usable_cores = len(os.sched_getaffinity(0))
    except AttributeError:
        usable_cores = ncpu
    gen_threads = args.gen_threads or max(1, min(32, ncpu // max(1, min(world, 8))))

    # ---- the shared stream: this rank writes global blocks [rank*nb, (rank+1)*nb)
    t0 = time.time()
    part, _ = make_stream(synth, model, model_name, kind, nb, bs, rank * nb, gen_threads, args.cache_dir)
    gen_s = time.time() - t0
    ctx = z.Context(local_rank)
    if world > 1:
        job = multigpu.ShardedJob.from_parts(ctx, part, dist, dev, coll_dev)   # all-gather pieces, scan on rank 0, broadcast table, plan
    else:
        job = multigpu.ShardedJob.single(ctx, part, dev)
    n_glob = job.sc.n_blocks
    assert n_glob == nb * world, (n_glob, nb, world)
    mine = job.shard
    dynamic = world > 1 and args.schedule == "queue"
    if dynamic:
        # any rank may decode any block: every block has its place (global id x block size) in every rank's buffer
        d_out = torch.zeros(n_glob * bs, dtype=torch.uint8, device=dev)
        out_off = [i * bs for i in range(n_glob)]
        out_cap = [bs] * n_glob
    else:
        d_out = torch.zeros(max(1, len(mine)) * bs, dtype=torch.uint8, device=dev)
        out_off = [i * bs for i in range(len(mine))]
        out_cap = [bs] * len(mine)

    def step():
        if dynamic:                                          # ranks pull chunks from the shared queue (+ all_gather of the results)
            return job.decode_dynamic(d_out, out_off, out_cap, queue_blocks=args.queue_blocks, kernel=args.kernel)
        return job.decode(d_out, out_off, out_cap, kernel=args.kernel)         # HIP decode of the shard (+ all_gather of the results)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    progress(f"stream written ({gen_s:.1f} s), job planned; warm-up")
    for _ in range(args.warmup):
        step()
    progress("timed steps")
    kernel_ms = []
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        table = step()
        kernel_ms.append(job.kernel_ms if dynamic else ctx.stats().kernel_ms)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev or "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    st = ctx.stats()
    if dynamic and not args.no_verify:
        # every pass of the queue writes the same global offsets, so bytes of an earlier pass could stand in for a block the
        # last pass wrote wrongly or not at all: verify ONE FRESH pass into a zeroed buffer (untimed, after the timed region)
        d_out.zero_()
        table = step()
        barrier()

    # ---- bit-exact check of the last (queue: the fresh) pass's output against the generator's plaintext (global block ids)
    ok = bool((table[:, 0] == 0).all() and (table[:, 1] == bs).all())
    if not args.no_verify:
        got = d_out.cpu().numpy()
        if dynamic:
            mine = [b for ids in job.pulled for b in ids]    # what this rank decoded in the last pass, at its global place
            where = {b: b for b in mine}
        else:
            where = {b: j for j, b in enumerate(mine)}
        for b in mine:
            j = where[b]
            if not np.array_equal(got[j * bs:(j + 1) * bs], synth.plain(kind, b, bs)):
                ok = False
                break
    coded_bytes = int(sum(s.data_len for s in job.sc.segments))
    total_plain = n_glob * bs
    if world > 1:
        allv = multigpu.all_gather_table(np.array([int(ok)], dtype=np.int64), dist, coll_dev)
        ok = bool(allv.all())
    rho = coded_bytes / total_plain
    rank_kernel_ms = None
    if world > 1:
        kk = multigpu.all_gather_table(np.array([int(round(float(np.mean(kernel_ms)) * 1000)), len(mine)], dtype=np.int64), dist, coll_dev)
        rank_kernel_ms = [float(x) / 1000.0 for x in kk[:, 0]]
        rank_blocks = [int(x) for x in kk[:, 1]]

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    ms_per_step = elapsed / args.steps * 1e3
    value = total_plain * args.steps / elapsed / 1e6
    kms = float(np.mean(kernel_ms))

    cpu = None
    extras = {}
    block_start = [int(b.tag_off) for b in job.sc.blocks] + [job.stream_len]
    progress(f"timed region done: {value:.1f} MB/s; cpu baseline")
    if not args.no_cpu_baseline:
        import oracle
        # bounded sample of the SAME stream: first S blocks, one host thread.
        per_block_guess = bs / 9e6 if base == "l1" else bs / 1.5e6
        S = args.cpu_sample_blocks or max(1, min(nb, int(15.0 / per_block_guess)))
        sample = job.h_stream[:block_start[S]].tobytes()
        t0 = time.perf_counter()
        out = oracle.decompress(sample, cap=S * bs + 16)
        dt = time.perf_counter() - t0
        assert len(out) == S * bs
        cpu = {"value": S * bs / dt / 1e6, "unit": "MB/s", "cores": 1, "kind": "port",
               "sample": f"first {S} of {n_glob} blocks ({S * bs >> 20} MiB plaintext) of the same stream, "
                         f"oracle/zpaq_oracle.c -O2, 1 thread, host has {ncpu} logical cores"}
        if world == 1 and not args.no_extras:
            # every host core this process may use (SURVEY 8d ii): blocks spread over threads
            progress("cpu all cores + sweep")
            extras["cpu_all_cores"] = cpu_all_cores(oracle, job.h_stream, block_start, n_glob, bs, max(1, S // 8), max(1, min(usable_cores, 64)))
            # ... and the sweep behind "it does not scale further": one block per thread, 8 threads up to every core the
            # process may use (distinct blocks of the same stream)
            share = cpu_share()
            pieces = [job.h_stream[block_start[b]:block_start[b + 1]].tobytes() for b in range(min(n_glob, 64))]
            extras["cpu_share"] = share
            extras["cpu_all_cores_sweep"] = cpu_sweep(oracle, pieces, bs, sweep_counts(share))

    if world == 1 and not args.no_extras:
        # host buffer -> host buffer through zpaqhip_decompress (scan + H2D + kernel + D2H), pinned memory
        progress("host to host")
        h_in = torch.from_numpy(job.h_stream).pin_memory()
        h_out = torch.empty(total_plain, dtype=torch.uint8).pin_memory()
        ctx.decompress_into(h_in.numpy(), h_out.numpy())                     # warm-up (allocations)
        t0 = time.perf_counter()
        n = ctx.decompress_into(h_in.numpy(), h_out.numpy())
        dt = time.perf_counter() - t0
        s2 = ctx.stats()
        extras["host_to_host"] = {"value": n / dt / 1e6, "unit": "MB/s", "memory": "pinned",
                                  "fraction_of_resident_rate": (n / dt / 1e6) / value if value else None,
                                  "h2d_ms": s2.h2d_ms, "kernel_ms": s2.kernel_ms, "d2h_ms": s2.d2h_ms, "launches": int(s2.launches),
                                  "bit_exact": bool(n == total_plain and all(
                                      np.array_equal(h_out.numpy()[b * bs:(b + 1) * bs], synth.plain(kind, b, bs)) for b in range(n_glob)))}
        extras["value_host_to_host"] = extras["host_to_host"]["value"] if extras["host_to_host"]["bit_exact"] else 0.0   # pinned host -> pinned host
        del h_in, h_out
        # BASELINE configs[2] (mid) and configs[4] (max + the reference's E8E9 PCOMP) shaped runs, 256 blocks each
        if model_name == "l1":
            ebs, nd = args.extras_block_bytes, args.extras_distinct
            small = min(ebs, 1 << 20)
            runs = (("mid", "T", ebs, nd, "BASELINE configs[2]"), ("max+e8e9", "X", ebs, nd, "BASELINE configs[4]"),
                    ("min", "T", ebs, nd, "BASELINE configs[1], L1' (built-in min: icm + isse, SURVEY 8d config 2)"),
                    ("l1", "X", small, 256, "BASELINE configs[1] on generator X (x86-like: ~20 % window misses)"),
                    ("l1", "R", small, 256, "BASELINE configs[1] on generator R (uniform random: ~75 % window misses)"))
            runs = tuple((a, b, c, d, e, 256) for a, b, c, d, e in runs) + (
                    ("l1", "T", 1 << 20, 256, "configs[1] > 256 blocks (two per CU)", 1024), ("mid", "T", 1 << 20, 64, "configs[2] > 256 blocks (one per CU)", 1024))
            for mname, mkind, ebs, nd, cfg, xnb in runs:
                progress(f"other config: {xnb} x {ebs >> 10} KiB {mname} {mkind}")
                v, k, r, okx, sx, xs, xsc = resident_run(z, synth, torch, ctx, dev, mname, mkind, xnb, ebs, gen_threads, args.cache_dir, nd)
                rec = {
                    "config": cfg,
                    "workload": f"{xnb} x {ebs >> 10} KiB blocks, model {mname}, plaintext {mkind}"
                                + (f" ({nd} distinct blocks, repeated {xnb // nd} x: each block decodes independently in its own arena slot)" if nd < xnb else ""),
                    "value": v if okx else 0.0, "unit": "MB/s", "bit_exact": bool(okx), "kernel_kind": int(sx.kernel_kind),
                    "roofline": roofline(mname.split("+")[0], k, xnb * ebs, r, xnb, ebs, mname)}
                if not args.no_cpu_baseline and xnb == 256:
                    xstart = [int(b.tag_off) for b in xsc.blocks] + [int(xs.size)]
                    t0 = time.perf_counter()
                    out = oracle.decompress(xs[:xstart[1]].tobytes(), cap=ebs + 16)
                    dt = time.perf_counter() - t0
                    rec["cpu_baseline"] = {"value": len(out) / dt / 1e6, "unit": "MB/s", "cores": 1, "kind": "port",
                                           "sample": f"first block ({ebs >> 20} MiB plaintext) of the same stream, oracle/zpaq_oracle.c -O2, 1 thread"}
                    rec["cpu_all_cores"] = cpu_all_cores(oracle, xs, xstart, xnb, ebs, 1, max(1, min(usable_cores, 32)))
                    if mname == "mid":                           # the sweep for the deep chain: 1 MiB of a block per thread would not
                        mid_small, _ = make_stream(synth, models.get("mid"), "mid", "T", 8, 1 << 20, 0, gen_threads, args.cache_dir)   # end a run: 8 distinct 1 MiB blocks
                        msc = z.scan(mid_small)
                        mst = [int(b.tag_off) for b in msc.blocks] + [int(mid_small.size)]
                        rec["cpu_all_cores_sweep"] = cpu_sweep(oracle, [mid_small[mst[i]:mst[i + 1]].tobytes() for i in range(8)], 1 << 20,
                                                               sweep_counts(cpu_share()))
                del xs
                extras.setdefault("other_configs", []).append(rec)
            progress("method streams")
            extras["method_streams"] = method_streams(z, synth, ctx, threads=gen_threads)
            progress("done")

    line = {
        "metric": "decompress MB/s (bit-exact) on 1 GiB multi-block stream",
        "value": value, "unit": "MB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u32", "data": "synthetic",
        "bit_exact": bool(ok),
        "config": {
            "workload": f"BASELINE configs[{1 if base in ('l1', 'min') else 2 if base == 'mid' else 4}]"
                        + (f" x{world} GPUs (configs[3] layout: one shared stream + block table)" if world > 1 else "")
                        + f": {nb} x {bs >> 20} MiB independent blocks per GPU, "
                        f"model {model_name} ({model.n} component(s)), plaintext generator {kind}",
            "zpaq_model": model_name, "blocks_per_gpu": nb, "block_bytes": bs, "plaintext": kind,
            "coded_over_plain": round(rho, 4),
            "parallelism": ((f"blocks x{world}: shared stream, broadcast table, work queue (chunks of {getattr(job, 'queue_blocks', args.queue_blocks)} cost-ordered blocks, "
                             f"one counter on the job's store), all_gather of results" if dynamic else
                             f"blocks x{world}: shared stream, broadcast table, LPT plan over estimated block costs, ids-sharded HIP decode, "
                             f"all_gather of results") if world > 1 else "blocks x1"),
            "kernel_kind": int(st.kernel_kind), "blocks_in_flight": int(st.concurrent),
            "shard_blocks": rank_blocks if world > 1 else [len(s) for s in job.plan],
            "rank_kernel_ms": rank_kernel_ms,
        },
        "roofline": roofline(base, kms, len(mine) * bs, rho, len(mine), bs, model_name),
        "cpu_baseline": cpu,
        "gen_seconds": round(gen_s, 1),
    }
    line.update(extras)
    if not ok:
        line["value"] = 0.0
        line["error"] = "GPU output is not bit-exact"
    try:                                                    # everything, for the builder: profiles/ keeps copies of this file
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "bench_full.json"), "w") as f:
            json.dump(line, f)
    except OSError:
        pass
    print(json.dumps(compact_line(line), separators=(",", ":")))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
