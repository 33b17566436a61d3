#!/usr/bin/env python3
"""Headline benchmark: images/sec of 160x160 Inception-ResNet-v1 triplet training (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One step = the whole hot path on one batch of synthetic input already resident in HBM:
  mining forward over a P x K = 45 x 4 = 180 image pool (BN-folded f16 inference path) -> [180,180] distance
  matrix -> online triplet selection (alpha 0.2) -> gather of the 90-image (30 triplets) train batch ->
  forward(training=True) -> l2_normalize -> triplet loss -> backward -> Keras Adam(eps=0.1) + L2 -> weight packs.
Nothing is skipped inside the timed region.  `value` counts the 90 TRAINED images per GPU per step
(BASELINE.json configs[1]); `value_train_only` times the same step without the mining forward.

Weak scaling: every rank trains its own 90-image batch (per-replica BatchNorm, local mining); gradients are summed
by RCCL all-reduce in backward-ordered buckets overlapped with backward.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch

FWD_GFLOP_PER_IMAGE = 2.802       # SURVEY.md 8d / BASELINE.md section 2 (E=128)
MFMA_PEAK_TFLOPS = 2500.0         # dense bf16/f16, MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0


def conv_flops(d, op):
    """Algorithmic FLOPs of one launch: 2 * MACs of the layer (SURVEY.md shape table); dgrad/wgrad = the same MACs."""
    cout = d.Cout + (d.Cout2 if d.dy2 else 0) + (d.Cout3 if d.dy3 else 0)        # sibling sources of a merged 1x1 data gradient
    return 2.0 * d.N * d.OH * d.OW * cout * d.KH * d.KW * d.Cin


def conv_bytes(d, op):
    """Algorithmic HBM bytes of one launch: every operand once (activations 2 B, fp32 dW 4 B)."""
    x = 2.0 * d.N * d.H * d.W * d.Cin
    y = (4.0 if d.out_f32 and op == 0 else 2.0) * d.N * d.OH * d.OW * d.Cout
    w = (4.0 if op == 2 else 2.0) * d.Cout * d.KH * d.KW * d.Cin
    return x + y + w


def event_pair_overhead_ms(n=200):
    """What two back-to-back event records measure with nothing between them: subtracted from every per-launch timing so
    that the figures are kernel durations (what rocprofv3 --kernel-trace reports), not duration + event bookkeeping."""
    evs = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        b.record()
        evs.append((a, b))
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in evs]))


def time_ops_individually(ops, stream, lib, reps=3, burst=4):
    """HIP events around a short burst of back-to-back launches of every op (on the stream the kernels run on); returns the
    per-launch MEAN in milliseconds, net of the empty event-pair overhead (amortised over the burst).  A burst, not a single
    launch: the event bookkeeping is several microseconds, comparable to the small kernels themselves, and this is what
    makes the figures agree with rocprofv3's per-kernel averages.  What repeated launches accumulate (BN statistics, dW) is
    re-zeroed by the next real step."""
    from facenet_amd import _lib
    sums = [0.0] * len(ops)
    for _ in range(reps):
        evs = []
        for op in ops:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(burst):
                rc = op.fn(*op.args, stream)
                if rc:
                    _lib.check(rc, op.name)
            b.record()
            evs.append((a, b))
        torch.cuda.synchronize()
        for i, (a, b) in enumerate(evs):
            sums[i] += a.elapsed_time(b)
    ovh = event_pair_overhead_ms()
    return [max((t / reps - ovh) / burst, 1e-4) for t in sums]


def time_ops_insitu(ops, stream, lib, reps=5, prepare=None):
    """IN-STEP per-launch durations: the whole step is replayed eagerly in program order with ONE event pair around every
    launch (events on the stream the kernels run on), `reps` times; returns the per-launch MEDIAN in milliseconds, net of the
    empty event-pair overhead.  Operands are as cold / warm as the preceding launches of the step leave them and every launch
    waits for its real predecessor -- this is the duration rocprofv3 --kernel-trace reports for the same launch, unlike a
    burst of identical back-to-back launches (time_ops_individually), which runs L2-warm."""
    from facenet_amd import _lib
    per = [[] for _ in ops]
    for _ in range(reps):
        if prepare is not None:
            prepare()
        evs = []
        for op in ops:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            rc = op.fn(*op.args, stream)
            if rc:
                _lib.check(rc, op.name)
            b.record()
            evs.append((a, b))
        torch.cuda.synchronize()
        for i, (a, b) in enumerate(evs):
            per[i].append(a.elapsed_time(b))
    ovh = event_pair_overhead_ms()
    return [max(float(np.median(t)) - ovh, 1e-4) for t in per]


def _mangle_hint(demangled):
    """rocprofv3 reports demangled names: rebuild the template-argument part in mangled spelling for one matcher."""
    import re
    m = re.match(r"void fn::(\w+)<(__bf16|_Float16), (.*)>\(", demangled)
    if not m:
        return ""
    args = "".join((f"Lb{int(a == 'true')}E" if a in ("true", "false") else f"Li{a}E") for a in m.group(3).split(", "))
    return f"{m.group(1)}I{'DF16b' if m.group(2) == '__bf16' else 'DF16_'}{args}E"


def latest_profile(suffix):
    """profiles/rNN_<suffix> of the highest round that has one (the run's tile choices and PMC passes belong together)."""
    import glob
    found = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r[0-9][0-9]_{suffix}")))
    return found[-1] if found else os.path.join(ROOT, "profiles", f"r00_{suffix}")


PMC_PROFILE = latest_profile("pmc_hbm_traffic.json")


def tile_signature(*tile_dicts):
    """Identifies the kernel variants a run launches (the autotuner's choices): PMC counters cannot be read inside this
    process, so `roofline.traffic` is only quoted from the committed rocprofv3 passes when THAT run used the same tiles."""
    import hashlib
    h = hashlib.sha1()
    for d in tile_dicts:
        for k in sorted(d):
            h.update(f"{k}={d[k]};".encode())
    return h.hexdigest()[:16]


def pmc_traffic(kernel_key, signature):
    """HBM bytes per launch of one conv instantiation from the committed rocprofv3 PMC passes (separate --pmc passes,
    FETCH_SIZE x2 + WRITE_SIZE as MI355X_MICROARCH.md prescribes; tools/pmc_dump.py); None when the profile is absent or was
    taken with other tile choices than this run's."""
    import re
    path = PMC_PROFILE
    if os.path.exists(path) and json.load(open(path)).get("tile_signature") != signature:
        return None
    if not os.path.exists(path):
        return None
    tn = {"__bf16": "DF16b", "_Float16": "DF16_"}
    m = re.match(r"(conv_igemm_kernel|conv_igemm_grouped_kernel|conv_wgrad_kernel|conv_wgrad_grouped_kernel)"
                 r"<(__bf16|_Float16),(\d+),(\d+)(?:,ks=(\d))?(?:,1x1=(\d))?>", kernel_key)
    f = re.match(r"(block17_infer_kernel|block35_infer_kernel|block8_infer_kernel|conv_wgrad_taps_kernel|conv_halo_kernel|bn_relu_fwd_kernel|"
                 r"bn_relu_bwd_apply_kernel|adam_keras_kernel)<(__bf16|_Float16)", kernel_key)
    if m:
        tname = tn[m.group(2)]
        if "wgrad" in m.group(1):      # <T, BMW, BNW, NORM>
            pat = re.compile(rf"{m.group(1)}I{tname}Li{m.group(3)}ELi{m.group(4)}ELb0EE")
        else:                          # <T, BM, BN, WM, WN, DEPTH, KS, PLAIN, MODE>
            pat = re.compile(rf"{m.group(1)}I{tname}Li{m.group(3)}ELi{m.group(4)}ELi\d+ELi\d+ELi\d+ELi{m.group(5) or 1}ELb{m.group(6)}ELi0EE")
    elif f:                            # one instantiation per dtype: mangled <kernel>I<type>..., demangled <kernel><<type>...
        pat = re.compile(rf"{f.group(1)}(I{tn[f.group(2)]}|<{f.group(2)})")
    else:
        return None
    n = tot = 0.0
    for name, v in json.load(open(path))["kernels"].items():
        if pat.search(name) or pat.search(_mangle_hint(name)):
            n += v["launches"]
            tot += v["launches"] * v["hbm_bytes_per_launch"]
    return round(tot / n) if n else None


def other_op_model(op, kind, net):
    """(kernel key, algorithmic FLOPs, algorithmic HBM bytes) of the launches that are not plain convolutions.
    Fused inference blocks (csrc/block_fused.hip): 2 * MACs of the block's layers (SURVEY.md shape table: Block35 22.2, Block17 44.0,
    Block8 14.4 M MAC per image); bytes = trunk in + out once, weights once.  Streaming kernels: the bytes they must touch."""
    tn = lambda dt: "__bf16" if dt == 0 else "_Float16"
    a = op.args
    if kind in ("block17_fused", "block35_fused", "block8_fused"):
        pre = op.name.split(":", 1)[1]
        Ls = [L for n, L in net.layers.items() if n.startswith(pre + "/")]
        hw = {"block17_fused": 64, "block35_fused": 289, "block8_fused": 9}[kind]
        N = a[2]
        macs = sum(L.numel for L in Ls) * hw * N
        chans = Ls[-1].cout                     # `up` restores the trunk width
        by = 2.0 * N * hw * chans * 2 + 2.0 * sum(L.numel for L in Ls)
        return f"{kind.replace('_fused', '_infer_kernel')}<{tn(a[-1])}>", 2.0 * macs, by
    if kind == "bn_relu_fwd":                   # (raw, ld, act, ld, M, C, ...): read y, write z
        return f"bn_relu_fwd_kernel<{tn(a[-1])}>", 0.0, 4.0 * a[4] * a[5]
    if kind == "bn_relu_bwd":                   # (dz, ld, y, ld, M, C, ...): read dz and y, write dy in place
        return f"bn_relu_bwd_apply_kernel<{tn(a[-1])}>", 0.0, 6.0 * a[4] * a[5]
    if kind == "adam_keras":                    # w, g, m, v read; w, m, v written; low-precision pack written
        return f"adam_keras_kernel<{tn(a[-1])}>", 0.0, 28.0 * a[6] + 2.0 * a[5]
    if kind == "conv_wgrad_reduce":
        ws = op.keep[1]
        return "wgrad_reduce_kernel", 0.0, 4.0 * sum(w.numel() for w in ws) + 4.0 * net.n_kernel
    return kind, 0.0, 0.0


def kernel_roofline(trainer, miner, lib, dump=None, signature=None):
    """Attribute event-timed launches to kernel instantiations; report the dominant one against its roofline."""
    import ctypes as C
    net = trainer.net
    st = net.stream()
    groups = {}
    all_ops = list(miner.ops) + [op for op in trainer.step_ops if not getattr(op.fn, "_torch_op", False)]
    trainer._zero()
    torch.cuda.synchronize()
    t_burst = time_ops_individually(all_ops, st, lib)
    t = time_ops_insitu(all_ops, st, lib, prepare=trainer._zero)
    burst_of = {}
    per_op = []
    for op, ms, msb in zip(all_ops, t, t_burst):
        kind = op.name.split(":")[0]
        if kind in ("conv_fwd_grouped", "conv_dgrad_grouped"):
            descs = op.keep[0]
            opi = 0 if kind == "conv_fwd_grouped" else 1
            tname = "__bf16" if descs[0].dtype == 0 else "_Float16"
            d0 = descs[0]
            plain = int(d0.KH == 1 and d0.KW == 1 and d0.stride == 1 and d0.pad_h == 0 and d0.pad_w == 0)
            key = f"conv_igemm_grouped_kernel<{tname},{op.name.split(':')[1].replace('x', ',').replace('k', ',ks=')},1x1={plain}>"
            fl = sum(conv_flops(d, opi) for d in descs)
            g = groups.setdefault(key, dict(ms=0.0, flops=0.0, bytes=0.0, launches=0, bound="mfma"))
            g["flops"] += fl
            g["bytes"] += sum(conv_bytes(d, opi) for d in descs)
            per_op.append(dict(op=op.name, kernel=key, us=round(ms * 1e3, 2), gflop=round(fl / 1e9, 3), layers=len(descs),
                               tflops=round(fl / (ms * 1e-3) / 1e12, 1) if ms > 0 else 0))
        elif kind in ("conv_wgrad_grouped", "conv_wgrad_taps"):
            descs = op.keep[0]
            tname = "__bf16" if descs[0].dtype == 0 else "_Float16"
            key = f"{kind}_kernel<{tname},{op.name.split(':')[1].replace('x', ',')}>"
            fl = sum(conv_flops(d, 2) for d in descs)
            g = groups.setdefault(key, dict(ms=0.0, flops=0.0, bytes=0.0, launches=0, bound="mfma"))
            g["flops"] += fl
            g["bytes"] += sum(conv_bytes(d, 2) for d in descs)
            per_op.append(dict(op=op.name, kernel=key, us=round(ms * 1e3, 2), gflop=round(fl / 1e9, 3), layers=len(descs),
                               tflops=round(fl / (ms * 1e-3) / 1e12, 1) if ms > 0 else 0))
        elif kind in ("conv_fwd", "conv_dgrad", "conv_wgrad") and op.keep:
            d = op.keep[0]
            opi = {"conv_fwd": 0, "conv_dgrad": 1, "conv_wgrad": 2}[kind]
            v = lib.fn_conv2d_variant(C.byref(d), opi)
            tname = "__bf16" if d.dtype == 0 else "_Float16"
            plain = int(d.KH == 1 and d.KW == 1 and d.stride == 1 and d.pad_h == 0 and d.pad_w == 0)
            if opi == 2:
                key = f"conv_wgrad_kernel<{tname},{v // 1000},{v % 1000}>"
            elif v >= 9000000:
                key = f"conv_halo_kernel<{tname},BN={v % 1000},3x3>"
            else:
                ks = f",ks={v // 1000000}" if v >= 1000000 else ""
                key = f"conv_igemm_kernel<{tname},{v % 1000000 // 1000},{v % 1000}{ks},1x1={plain}>"
            g = groups.setdefault(key, dict(ms=0.0, flops=0.0, bytes=0.0, launches=0, bound="mfma"))
            g["flops"] += conv_flops(d, opi)
            g["bytes"] += conv_bytes(d, opi)
            per_op.append(dict(op=op.name, kernel=key, us=round(ms * 1e3, 2), gflop=round(conv_flops(d, opi) / 1e9, 3),
                               tflops=round(conv_flops(d, opi) / (ms * 1e-3) / 1e12, 1) if ms > 0 else 0,
                               shape=f"N{d.N} {d.H}x{d.W}x{d.Cin}->{d.OH}x{d.OW}x{d.Cout} k{d.KH}x{d.KW}s{d.stride}"))
        else:
            key, fl, by = other_op_model(op, kind, net)
            g = groups.setdefault(key, dict(ms=0.0, flops=0.0, bytes=0.0, launches=0, bound="mfma" if fl else "hbm"))
            g["flops"] += fl
            g["bytes"] += by
            per_op.append(dict(op=op.name, kernel=key, us=round(ms * 1e3, 2), **({"gflop": round(fl / 1e9, 3), "tflops": round(fl / (ms * 1e-3) / 1e12, 1)} if fl else {}),
                               **({"gbs": round(by / (ms * 1e-3) / 1e9, 1)} if by and ms > 0 else {})))
        g["ms"] += ms
        g["ms_burst"] = g.get("ms_burst", 0.0) + msb
        g["launches"] += 1
        per_op[-1]["us_burst"] = round(msb * 1e3, 2)
    if dump:
        with open(dump, "w") as f:
            json.dump(per_op, f, indent=0)
    total_ms = sum(g["ms"] for g in groups.values())
    top = sorted(groups.items(), key=lambda kv: -kv[1]["ms"])
    # the dominant instantiation by summed in-step time -- convolution, fused block or streaming kernel alike (every group that can
    # top the list carries its algorithmic flops / bytes: other_op_model)
    name, g = next((kv for kv in top if kv[1]["flops"] > 0 or kv[1]["bytes"] > 0), top[0])
    # the roofline that binds the dominant kernel: the larger of (algorithmic flops / MFMA peak) and (algorithmic bytes /
    # HBM peak) is the time it cannot beat; the other one is reported under "other_roofline"
    sec = g["ms"] * 1e-3
    tf = g["flops"] / sec / 1e12 if sec > 0 else 0.0
    gbs = g["bytes"] / sec / 1e9 if sec > 0 else 0.0
    mfma = {"bound": "mfma", "achieved": round(tf, 2), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(tf / MFMA_PEAK_TFLOPS, 4)}
    hbm = {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4)}
    first, other = (hbm, mfma) if hbm["frac"] > mfma["frac"] else (mfma, hbm)
    roof = {
        **first, "kernel": name, "traffic": pmc_traffic(name, signature),
        "timing": "in-step: one HIP-event pair per launch over an eager replay of the whole step, median of 5 passes",
        "launches_per_step": g["launches"], "avg_launch_us": round(1e3 * g["ms"] / g["launches"], 2),
        "avg_launch_us_warm_burst": round(1e3 * g.get("ms_burst", 0.0) / g["launches"], 2),
        "share_of_step_kernel_time": round(g["ms"] / total_ms, 3),
        "flop_per_launch_avg": round(g["flops"] / g["launches"], 0), "algorithmic_bytes_per_launch_avg": round(g["bytes"] / g["launches"], 0),
        "other_roofline": other,
    }
    breakdown = [{"kernel": k, "ms": round(v["ms"], 3), "launches": v["launches"],
                  **({"tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 1)} if v["flops"] > 0 and v["ms"] > 0 else {}),
                  **({"gbs": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1)} if v["bytes"] > 0 and v["ms"] > 0 else {})}
                 for k, v in top[:12]]
    return roof, breakdown, total_ms


def cpu_baseline(pool_n=180, batch=90, E=128, reps=3):
    """The oracle (a port: PyTorch-CPU fp32 restatement) timed on this box's host cores on the SAME workload as the GPU step
    (BASELINE.json configs[1]): mining forward over the 45x4 = 180 image pool + distance matrix + selection + one 90-image
    (30 triplets) train step incl. Keras Adam; 1 warm-up + `reps` timed steps."""
    from oracle import facenet_oracle as fo
    params, trainable, regularized = fo.build_params(E, seed=0)
    rng = np.random.default_rng(0)
    pool = rng.integers(0, 256, (pool_n, 160, 160, 3), dtype=np.uint8)
    labels = np.repeat(np.arange(pool_n // 4), 4)
    opt = fo.AdamKeras(trainable, params, lr=0.05)

    def one():
        with torch.no_grad():
            emb = fo.Oracle(params).forward(pool, training=False).numpy()
        dist = fo.squared_distance_matrix(emb)
        trip = fo.select_triplets(dist, labels, 0.2, batch // 3, seed=0)
        x = pool[trip.reshape(-1)]
        _, _, grads, stats, _ = fo.train_step_grads(params, trainable, regularized, x, "triplet", alpha=0.2)
        opt.step(params, grads)
        for k, v in stats.items():
            params[k].copy_(v)
    one()  # warm-up
    t0 = time.perf_counter()
    for _ in range(reps):
        one()
    dt = (time.perf_counter() - t0) / reps
    return {"value": round(batch / dt, 2), "unit": "images/sec", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"1 warm-up + {reps} timed steps of the full workload (mining forward {pool_n} images + selection + triplet train step on "
                      f"{batch} images + Keras Adam), fp32 PyTorch-CPU oracle, {dt:.2f} s/step, host has {os.cpu_count()} logical cores"}


def embedding_l2_err(dev):
    """BASELINE.json metric, second half ("LFW embedding L2 err"; no LFW images exist here): max row L2 distance
    between the GPU path's unit-norm embeddings and the CPU oracle's on BASELINE.json configs[0] -- 16 seeded random uint8
    images, E=128, fresh and perturbed BatchNorm statistics -- for both storage types.  north_star bound: 1e-3."""
    from facenet_amd.engine import Network
    from oracle import facenet_oracle as fo
    x = np.random.default_rng(0).integers(0, 256, (16, 160, 160, 3), dtype=np.uint8)
    out = {}
    for variant in ("fresh", "perturbed"):
        params, _, _ = fo.build_params(128, seed=0)
        if variant == "perturbed":
            fo.perturb_bn_stats(params, seed=1)
        ref = fo.Oracle(params).forward(x, training=False)
        for name, dt in (("f16", torch.float16), ("bf16", torch.bfloat16)):
            net = Network(embedding_size=128, device=str(dev), infer_dtype=dt)
            net.load_keras_params(params)
            plan = net.plan(16, training=False)
            plan.images.copy_(torch.from_numpy(x))
            plan.run_forward()
            torch.cuda.synchronize()
            emb = fo.l2_normalize(plan.embedding.buf.act.view(16, 128).float().cpu())
            out[f"{name}_{variant}"] = float(f"{(emb - ref).norm(dim=1).max().item():.3e}")
    out["note"] = ("max row L2 distance to the fp32 CPU oracle, 16 random 160x160 images (BASELINE.json configs[0]); the inference / mining "
                   "path stores activations in f16 (bound 1e-3 met), bf16 storage is the training dtype and is shown for reference")
    return out


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: run `python -m torch.distributed.run --nnodes=1 --nproc-per-node N
    --master-addr 127.0.0.1 --master-port <free> bench.py <same arguments>` as a child process (one rank per GPU, RCCL), let
    it print rank 0's JSON line on our stdout, and return its exit code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def dry_run(args):
    """--dry-run: everything around the HIP work -- rendezvous (FACENET_DIST_BACKEND, default nccl = RCCL), barrier, MAX-over-ranks
    timing reduction, one JSON line from rank 0 -- with a sleep instead of the step.  Lets the N > 1 launch path be tested on a box
    without GPUs (tests/test_bench_launcher.py)."""
    import torch.distributed as dist
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(os.environ.get("FACENET_DIST_BACKEND", "nccl"))
        dist.barrier()
    t0 = time.perf_counter()
    time.sleep(0.001 * args.steps)
    el = time.perf_counter() - t0
    ranks_seen = 1
    if world > 1:
        t = torch.tensor([el], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
        ranks_seen = dist.get_world_size()
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"metric": "images/sec (160x160 triplet train)", "value": None, "unit": "images/sec", "n_gpus": world,
                          "dist_world_size": ranks_seen, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * el / args.steps, 3),
                          "dry_run": True, "data": "none"}))
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)      # 20 warm-up + 100 timed steps
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-graph", action="store_true", help="replay the launch list eagerly instead of through HIP graphs")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the CPU legs (oracle timing and embedding error)")
    ap.add_argument("--cpu-sample", action="store_true", help="time the CPU oracle on a third of the workload (60-image pool + 30-image step)")
    ap.add_argument("--batch", type=int, default=90)
    ap.add_argument("--pool", type=int, default=180)
    ap.add_argument("--streams", type=int, default=1, help="HIP streams the dependency scheduler may use (1 = serial)")
    ap.add_argument("--force-segments", action="store_true", help="1 GPU: run the data-parallel segment structure (6 buckets) without the all-reduce")
    ap.add_argument("--exchange-self", action="store_true",
                    help="1 GPU: the whole data-parallel step INCLUDING the bucket all-reduces, through a ONE-rank communicator of "
                         "FACENET_DIST_BACKEND (default nccl = RCCL): the exchange is the identity, the backend calls, the "
                         "communication stream and its ordering against the captured segments are the real ones")
    ap.add_argument("--dump-ops", default=None, help="write per-launch HIP-event timings (eager) to this JSON file")
    ap.add_argument("--dry-run", action="store_true", help="launcher / rendezvous / reporting path only: no HIP work (CPU test of --gpus N)")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.streams > 2 and not args.no_graph:
        raise SystemExit("--streams > 2 needs --no-graph: captured schedules span at most 2 streams (DESIGN.md section 5)")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N`: start the N ranks ourselves, as a CHILD torch.distributed.run (never exec: this
        # process must not be replaced once anything has touched the GPU, and nothing has yet -- no HIP call above this line)
        sys.exit(launch_ranks(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus} (or unset WORLD_SIZE and let "
                         f"bench.py start the ranks)")
    if args.dry_run:
        return dry_run(args)

    # the tile choices of the committed profiles (profiles/rNN_tile_cache.json of the latest round: one autotune run on an MI355X) are reused when
    # present: the run then launches the kernel variants the rocprofv3 passes under profiles/ measured (roofline.traffic), starts
    # faster and is reproducible.  FACENET_TUNE_CACHE= (empty) re-tunes from scratch.
    if "FACENET_TUNE_CACHE" not in os.environ and os.path.exists(latest_profile("tile_cache.json")):
        os.environ["FACENET_TUNE_CACHE"] = latest_profile("tile_cache.json")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # one process per GPU; FACENET_DIST_BACKEND=gloo lets the whole data-parallel path be rehearsed with several ranks on
    # ONE GPU (RCCL refuses two ranks on a device), which is how it is tested on the single-GPU development box
    backend = os.environ.get("FACENET_DIST_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    dev_index = local_rank if backend == "nccl" else local_rank % max(1, ndev)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    pg = None
    if world == 1 and args.exchange_self:
        import socket
        import torch.distributed as dist
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        kw = {"device_id": dev} if backend == "nccl" else {}
        dist.init_process_group(backend, init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, **kw)
        pg = dist.group.WORLD
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)   # backend "nccl" is RCCL on ROCm
        else:
            dist.init_process_group(backend)
        pg = dist.group.WORLD

    from facenet_amd import _lib
    from facenet_amd.engine import Network
    from facenet_amd.train import Trainer, TripletMiner
    lib = _lib.load()

    E, B, POOL = 128, args.batch, args.pool
    net = Network(embedding_size=E, device=str(dev), train_dtype=torch.bfloat16, infer_dtype=torch.float16, seed=0)
    trainer = Trainer(net, batch=B, loss="triplet", alpha=0.2, lr=0.05, world_size=world, process_group=pg, n_streams=args.streams,
                      force_segments=args.force_segments)
    labels = np.repeat(np.arange(POOL // 4), 4)
    miner = TripletMiner(net, POOL, labels, B // 3, alpha=0.2, seed=1000 * rank, n_streams=args.streams)
    miner.build(trainer.plan.images)
    # synthetic pools resident in HBM before the timed region (seeds offset by rank)
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    pools = [torch.randint(0, 256, (POOL, 160, 160, 3), dtype=torch.uint8, device=dev, generator=g) for _ in range(4)]

    # ---- capture ---------------------------------------------------------------------------------
    miner.plan.images.copy_(pools[0])
    miner.run()
    trainer.step_eager()
    torch.cuda.synchronize()
    mine_graph = None
    if not args.no_graph:
        from facenet_amd.train import GraphRunner
        from facenet_amd.schedule import make_events
        mine_events = make_events(miner.sched)
        mine_graph = GraphRunner(dev).capture(lambda: miner.run(mine_events))
        trainer.capture()

    def step(i, with_mining=True):
        if with_mining:
            miner.plan.images.copy_(pools[i % len(pools)])
            if mine_graph is not None:
                mine_graph.replay()
            else:
                miner.run()
        trainer.step()

    def timed(n, with_mining):
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            step(i, with_mining)
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        el = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([el], dtype=torch.float64, device=dev)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            el = float(t.item())
        return el

    for i in range(args.warmup):
        step(i)
    loss_warm = trainer.loss_value()
    elapsed = timed(args.steps, True)
    loss = trainer.loss_value()            # last mined step (the train-only replays below re-fit one batch: not a loss to report)
    n2 = max(5, args.steps // 2)
    elapsed_train = timed(n2, False)

    out = None
    if rank == 0:
        ms = 1e3 * elapsed / args.steps
        value = B * world * args.steps / elapsed
        out = {
            "metric": "images/sec (160x160 triplet train)", "value": round(value, 1), "unit": "images/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "Inception-ResNet-v1 triplet-loss train, batch=90 (30 triplets), bf16, per GPU; "
                                   "mining forward over a 45x4=180 image pool (f16 inference path) + on-device selection inside the step",
                       "global_batch": B * world, "image": "160x160x3 uint8", "embedding": E, "alpha": 0.2,
                       "optimizer": "Keras Adam eps=0.1 + L2 5e-4", "parallelism": f"dp{world}",
                       "hip_graph": not args.no_graph, "streams": args.streams},
            "dist_world_size": torch.distributed.get_world_size() if pg is not None else 1,
            "dist_backend": (backend + (" (RCCL)" if backend == "nccl" else "")) if pg is not None else None,
            "value_train_only": round(B * world * n2 / elapsed_train, 1),
            "ms_per_step_train_only": round(1e3 * elapsed_train / n2, 3),
            "loss_after_warmup": round(float(loss_warm), 5), "final_loss": round(loss, 5),
            "model_tflops": round((3 * B + POOL) * FWD_GFLOP_PER_IMAGE * 1e-3 / (ms * 1e-3), 1),
        }
    if rank == 0 and world == 1:
        sig = tile_signature(trainer.tiles, miner.tiles)
        roof, breakdown, kernel_ms = kernel_roofline(trainer, miner, lib, args.dump_ops, sig)
        out["roofline"] = roof
        out["kernel_breakdown"] = breakdown
        out["sum_kernel_ms_insitu"] = round(kernel_ms, 3)
        out["launches_per_step"] = len(miner.ops) + len([op for op in trainer.step_ops])
        out["tile_signature"] = sig
        # whole-step rooflines (SURVEY.md 8d / BASELINE.md section 2): algorithmic work of one step against the chip's peaks
        step_flop = (3 * B + POOL) * FWD_GFLOP_PER_IMAGE * 1e9
        step_bytes = POOL * 12.4e6 + B * 30.8e6 + 0.82e9
        out["step_roofline"] = {
            "flop": step_flop, "lower_bound_bytes": step_bytes,
            "mfma_frac": round(step_flop / (ms * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS, 4),
            "hbm_frac": round(step_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "lower_bound_ms": round(1e3 * max(step_flop / (MFMA_PEAK_TFLOPS * 1e12), step_bytes / (HBM_PEAK_GBS * 1e9)), 3),
            "note": "forward 2.802 GFLOP/img x (180 mined + 3 x 90 trained); bytes = 12.4 MB/img mining + 30.8 MB/img training + 0.82 GB parameters"}
        if not args.no_cpu_baseline:
            out["embedding_l2_err"] = embedding_l2_err(dev)
            out["cpu_baseline"] = cpu_baseline(60, 30, reps=2) if args.cpu_sample else cpu_baseline()
    if pg is not None:
        # gradient exchange: per-bucket all-reduce time on the communication stream and how much of it backward hides
        # (every rank runs the profiled steps -- they are collective; rank 0 reports its own view)
        xp = trainer.exchange_profile(steps=3)
        if rank == 0:
            out["gradient_exchange"] = {**xp, "n_buckets": len(trainer.buckets), "dtype": "fp32", "op": "all_reduce(SUM), 1/world inside the optimiser",
                                        "note": "HIP events on the communication stream; overlapped = share of all-reduce time before the compute stream finished backward"}
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    sys.exit(main())
