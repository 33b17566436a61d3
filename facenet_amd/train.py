"""Training steps on the static-plan engine.

* ``Trainer(loss='softmax')`` is the reference's train step: ``Sequential([model, Dense(C)])`` +
  ``SparseCategoricalCrossentropy(from_logits=True)`` + ``Adam(epsilon=0.1)`` + Keras L2(5e-4)
  (apps/train_softmax.py:49-104; embedding NOT normalised in training, inception_resnet_v1.py:491).
* ``Trainer(loss='triplet')`` is the north-star path the reference lacks (build-defined
  from arXiv 1503.03832): forward(training=True) -> l2_normalize -> triplet loss over rows laid out
  (a0,p0,n0,a1,...).  ``TripletMiner`` does the online selection in a PxK pool on device.
* Data parallelism restates ``tf.distribute.MirroredStrategy()`` (apps/train_softmax_tf2_gpus.py:49):
  one process per GPU, per-replica BatchNorm, gradients summed by RCCL all-reduce in backward-ordered
  buckets on a side stream (overlapped with the rest of backward), divided by the replica count
  inside the fused optimiser.
"""
from __future__ import annotations

import ctypes as C
import json
import os
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib, parallel
from .engine import BN_EPS, L2_WEIGHT, Lowering, Network, _pad8, _ptr
from .schedule import Op, Schedule, StreamSet, levelize, make_events, region, run_schedule, torch_op


class GraphRunner:
    """Capture launch schedules into HIP graphs (one per segment) and replay them.

    Segments exist so that collectives issued between them stay outside the captured graphs."""

    def __init__(self, device: torch.device):
        self.device = device
        self.graphs: List[torch.cuda.CUDAGraph] = []
        self.pool = None

    def capture(self, fn: Callable[[], None]) -> torch.cuda.CUDAGraph:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, pool=self.pool):
            fn()
        if self.pool is None:
            self.pool = g.pool()
        self.graphs.append(g)
        return g


def group_wgrads(ops: List[Op], net: Network) -> List[Op]:
    """Weight gradients have no consumer before the optimiser (or the bucket all-reduce): pull every ``conv_wgrad`` launch
    out of ``ops`` and append ONE grouped launch per tile variant at the end (fn_conv2d_wgrad_grouped), planned once on the
    host.  Thousands of workgroups per launch instead of ~130 launches that each fill a fraction of the 256 CUs."""
    lib = net.lib
    singles = [op for op in ops if op.name.startswith("conv_wgrad:") and op.keep]
    if len(singles) < 2:
        return list(ops)
    rest = [op for op in ops if not (op.name.startswith("conv_wgrad:") and op.keep)]
    groups = {}
    for op in singles:
        d = op.keep[0]
        v = lib.fn_conv2d_variant(C.byref(d), 2)
        groups.setdefault((v + (1000000 if d.nrm_stats and v < 5000000 else 0), d.dtype), []).append(op)
    nbytes = lib.fn_conv2d_wgrad_arg_bytes()
    out = list(rest)
    split_tables, split_keep, split_writes = [], [], []
    for (variant, dt), members in sorted(groups.items()):
        n = len(members)
        descs = (_lib.ConvDesc * n)(*[m.keep[0] for m in members])
        host_args = (C.c_uint8 * (nbytes * n))()
        host_prefix = (C.c_int32 * (n + 1))()
        ws_elems = C.c_int64(0)
        total = lib.fn_conv2d_wgrad_group_build(descs, n, variant, host_args, host_prefix, None, C.byref(ws_elems))   # sizing call
        if total < 0:
            _lib.check(total, "wgrad_group_build")
        # split layers write one fp32 slab per pixel split, summed in order by fn_conv2d_wgrad_reduce: no atomics, same bits every run
        ws = torch.empty(max(1, ws_elems.value), dtype=torch.float32, device=net.device)
        total = lib.fn_conv2d_wgrad_group_build(descs, n, variant, host_args, host_prefix, _ptr(ws), C.byref(ws_elems))
        if total < 0:
            _lib.check(total, "wgrad_group_build")
        dev_args = torch.frombuffer(bytearray(host_args), dtype=torch.uint8).to(net.device)
        dev_prefix = torch.tensor(list(host_prefix), dtype=torch.int32, device=net.device)
        reads, writes = [], []
        for m in members:
            reads.extend(m.reads)
            writes.extend(m.writes)
        if variant >= 5000000:      # tap-sharing kernel: 5000000 + BMW*1000 + taps*10 + (stride 2)
            code = variant - 5000000
            name = f"conv_wgrad_taps:{code // 1000}x{code % 1000 // 10}" + ("s2" if code % 10 else "")
        else:
            name = f"conv_wgrad_grouped:{variant % 1000000 // 1000}x{variant % 1000}" + (":norm" if variant >= 1000000 else "")
        out.append(Op(name, lib.fn_conv2d_wgrad_grouped,
                      (_ptr(dev_args), _ptr(dev_prefix), n, total, variant, dt), keep=(descs, dev_args, dev_prefix, members, ws),
                      reads=tuple(reads), writes=tuple(writes) + (region(ws),)))
        if ws_elems.value > 0:
            split_tables.append(dev_args)
            split_keep.append(ws)
            split_writes.extend(writes)
    if split_tables:      # ONE ordered slab reduction for the split layers of every group (records of both kernels share a layout)
        table = torch.cat(split_tables)
        out.append(Op("conv_wgrad_reduce", lib.fn_conv2d_wgrad_reduce, (_ptr(table), table.numel() // nbytes), keep=(table, split_keep),
                      reads=tuple(region(w) for w in split_keep), writes=tuple(split_writes)))
    return out


TILE_CANDIDATES = tuple((bm, bn) for bm in (128, 64, 32) for bn in (128, 64, 32))


def autotune_convs(ops: Sequence[Op], net: Network, launches: int = 8, rounds: int = 2) -> Dict[str, int]:
    """Measure, don't guess: time every forward / data-gradient convolution of a plan with each tile variant (a burst of
    back-to-back launches between two HIP events, best of `rounds`) and write the winner into the descriptor
    (fn_conv_desc.tile_fwd / tile_dgrad).  The library heuristic stays the fallback (FACENET_AUTOTUNE=0) and the tie
    breaker: a candidate must beat it by 3 % to replace it.  Runs once per plan, before grouping and graph capture; what
    the launches write while being timed is overwritten or re-zeroed by the first real step."""
    if os.environ.get("FACENET_AUTOTUNE", "1") == "0":
        return {}
    lib, st = net.lib, net.stream()
    chosen: Dict[str, int] = {}
    # FACENET_TUNE_CACHE=<file>: reuse the tiles of an earlier run (same shapes) instead of timing again -- reproducible
    # plans, and profiles of a tuned run that do not contain the tuning bursts
    cache_path = os.environ.get("FACENET_TUNE_CACHE")
    cache: Dict[str, int] = {}
    if cache_path and os.path.exists(cache_path):
        with open(cache_path) as fh:
            cache = json.load(fh)
    dirty = False

    def burst(op):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(launches):
            rc = op.fn(*op.args, st)
            if rc:
                return float("inf")
        b.record()
        b.synchronize()
        return a.elapsed_time(b)

    for op in ops:
        kind = op.name.split(":")[0]
        if kind not in ("conv_fwd", "conv_dgrad") or not op.keep or not isinstance(op.keep[0], _lib.ConvDesc):
            continue
        d = op.keep[0]
        field = "tile_fwd" if kind == "conv_fwd" else "tile_dgrad"
        nout = d.Cout if kind == "conv_fwd" else d.Cin
        key = f"{op.name}|N{d.N}|{d.H}x{d.W}x{d.Cin}|dt{d.dtype}|nrm{int(bool(d.nrm_stats))}"
        if key in cache:
            setattr(d, field, int(cache[key]))
            chosen[op.name] = int(cache[key])
            continue
        setattr(d, field, 0)
        base_code = lib.fn_conv2d_variant(C.byref(d), 0 if kind == "conv_fwd" else 1)
        base = base_code % 1000000
        timings = {}
        if base_code >= 9000000:                         # the library's own choice is the halo-tile kernel: it competes as tile 0
            base = 0
            burst(op)
            timings[0] = min(burst(op) for _ in range(rounds))
        for bm, bn in TILE_CANDIDATES:
            if bn > 32 and bn // 2 >= nout:              # a tile twice as wide as the layer only multiplies zeros
                continue
            setattr(d, field, bm * 1000 + bn)
            burst(op)                                    # warm-up (code object, L2)
            timings[bm * 1000 + bn] = min(burst(op) for _ in range(rounds))
        best = min(timings, key=timings.get)
        if base in timings and timings[best] > 0.97 * timings[base]:
            best = base
        setattr(d, field, best)
        chosen[op.name] = best
        cache[key] = best
        dirty = True
    torch.cuda.synchronize()
    if cache_path and dirty:
        tmp = f"{cache_path}.{os.getpid()}.tmp"        # several ranks may share the file: replace it atomically
        with open(tmp, "w") as fh:
            json.dump(cache, fh, indent=0)
        os.replace(tmp, cache_path)
    return chosen


def group_convs(ops: List[Op], net: Network) -> List[Op]:
    """Order the launch list by dependency level (a valid topological order) and fuse same-level forward / data-gradient
    convolutions that share a tile variant into ONE grouped launch (fn_conv2d_grouped): sibling inception towers run as one
    kernel with 2-3x the workgroups instead of 2-3 under-occupied launches."""
    lib = net.lib
    level = levelize(ops)
    order = sorted(range(len(ops)), key=lambda i: (level[i], i))
    nbytes = lib.fn_conv2d_arg_bytes()
    buckets = {}
    for i in order:
        op = ops[i]
        kind = op.name.split(":")[0]
        if kind in ("conv_fwd", "conv_dgrad") and op.keep and isinstance(op.keep[0], _lib.ConvDesc) and not op.keep[0].dy2:
            d = op.keep[0]
            opi = 0 if kind == "conv_fwd" else 1
            if lib.fn_conv2d_variant(C.byref(d), opi) >= 9000000:
                continue                             # halo-tile kernel: a launch of its own
            plain = int(d.KH == 1 and d.KW == 1 and d.stride == 1 and d.pad_h == 0 and d.pad_w == 0)
            if opi == 0 and d.nrm_stats:
                plain |= 2                           # normalise-on-load members form their own groups
            buckets.setdefault((level[i], opi, lib.fn_conv2d_variant(C.byref(d), opi), plain, d.dtype), []).append(i)
    fused_at, skip = {}, set()
    for (lv, opi, variant, plain, dt), idxs in buckets.items():
        for c0 in range(0, len(idxs), 8):            # at most 8 layers per launch (linear scan in the kernel)
            chunk = idxs[c0:c0 + 8]
            if len(chunk) < 2:
                continue
            n = len(chunk)
            descs = (_lib.ConvDesc * n)(*[ops[i].keep[0] for i in chunk])
            host_args = (C.c_uint8 * (nbytes * n))()
            host_prefix = (C.c_int32 * (n + 1))()
            smem = C.c_int32(0)
            total = lib.fn_conv2d_group_build(descs, n, opi, variant, host_args, host_prefix, C.byref(smem))
            if total < 0:
                _lib.check(total, "conv_group_build")
            dev_args = torch.frombuffer(bytearray(host_args), dtype=torch.uint8).to(net.device)
            dev_prefix = torch.tensor(list(host_prefix), dtype=torch.int32, device=net.device)
            reads, writes = [], []
            for i in chunk:
                reads.extend(ops[i].reads)
                writes.extend(ops[i].writes)
            kname = "conv_fwd_grouped" if opi == 0 else "conv_dgrad_grouped"
            vname = f"{variant % 1000000 // 1000}x{variant % 1000}" + (f"k{variant // 1000000}" if variant >= 1000000 else "")
            fused_at[chunk[0]] = Op(f"{kname}:{vname}:" + "+".join(ops[i].name.split(":", 1)[1] for i in chunk),
                                   lib.fn_conv2d_grouped, (_ptr(dev_args), _ptr(dev_prefix), n, total, variant, plain, smem.value, dt),
                                   keep=(descs, dev_args, dev_prefix, [ops[i] for i in chunk]), reads=tuple(reads), writes=tuple(writes))
            skip.update(chunk[1:])
    out = []
    for i in order:
        if i in skip:
            continue
        out.append(fused_at.get(i, ops[i]))
    return out


def adam_beta_powers(t: int, beta1: float, beta2: float) -> Tuple[float, float]:
    """(beta1^t, beta2^t) as fn_adam_tick derives them from the integer step count: the betas arrive on the device as fp32, the
    power is taken in double and rounded to fp32 once."""
    return float(np.float32(np.float64(np.float32(beta1)) ** t)), float(np.float32(np.float64(np.float32(beta2)) ** t))


def _streams_for(net: Network, n_streams: int) -> StreamSet:
    ss = getattr(net, "_stream_set", None)
    if ss is None or len(ss.side) < n_streams - 1:
        ss = StreamSet(net.device, n_streams)
        net._stream_set = ss
    return ss


class Trainer:
    def __init__(self, net: Network, batch: int, loss: str = "triplet", alpha: float = 0.2, lr: float = 0.05, beta1: float = 0.9,
                 beta2: float = 0.999, epsilon: float = 0.1, l2: float = L2_WEIGHT, world_size: int = 1, process_group=None,
                 n_buckets: int = 6, n_streams: int = 1, group_wgrad: bool = True, force_segments: bool = False):
        self.group_wgrad = group_wgrad
        # force_segments: a single replica runs the data-parallel step structure (backward cut at the bucket boundaries, one graph
        # per segment, per-segment grouped weight gradients) with the all-reduce left out: what the segmentation alone costs
        # A process group given together with world_size == 1 still EXCHANGES: the bucket all-reduces run through that one-rank
        # communicator on the communication stream (identity on the data, the real backend calls and stream ordering) -- how
        # the RCCL path is exercised on a one-GPU box (bench.py --exchange-self, tests/test_gpu_dp.py).
        self.exchange = world_size > 1 or process_group is not None
        self.segmented = self.exchange or force_segments
        if loss not in ("triplet", "softmax"):
            raise ValueError(f"unknown loss {loss!r}")
        if loss == "triplet" and batch % 3:
            raise ValueError("triplet batches are laid out (a,p,n,...): batch must be a multiple of 3")
        if loss == "softmax" and net.nrof_classes is None:
            raise ValueError("softmax training needs Network(nrof_classes=...)")
        self.net, self.N, self.loss_kind, self.alpha = net, batch, loss, alpha
        self.beta1, self.beta2, self.eps, self.l2 = beta1, beta2, epsilon, l2
        self.world, self.pg = world_size, process_group
        self.n_streams = n_streams
        dev, E, lib = net.device, net.E, net.lib
        self.lib = lib
        self.G = net.alloc_grads()       # + net.Gacc: fixed-point accumulators of the bias gradients (engine.Network.alloc_grads)
        self.M = torch.zeros_like(self.G)
        self.V = torch.zeros_like(self.G)
        # hyper = {lr, beta1^t, beta2^t, grad_scale, t (int32 bits), 3 spare words}; lives on device so HIP-graph replays see
        # LR changes and advance Adam's step count themselves (fn_adam_tick)
        self.hyper = torch.tensor([lr, 1.0, 1.0, 1.0 / world_size, 0.0, 0.0, 0.0, 0.0], dtype=torch.float32, device=dev)
        self.loss = torch.zeros(4, dtype=torch.float32, device=dev)     # [0] the loss; [1..3] the launch's flag + fixed-point accumulator
        if world_size > 1:
            # MirroredStrategy creates every replica from the SAME variables (apps/train_softmax_tf2_gpus.py:49-67): rank 0's
            # parameters and moving statistics win, whatever seed or file the other ranks were built from
            parallel.broadcast_parameters([net.P, net.S_mean, net.S_var], src=0, group=process_group)
            net.folded_valid = False
            net.refresh_packs()
        self.plan: Lowering = net.plan(batch, training=True)
        self.demb = torch.zeros(batch, E, dtype=torch.float32, device=dev)
        self.dt = _lib.dtype_code(net.train_dtype)
        ebuf = self.plan.embedding.buf
        emb = ebuf.act
        r_emb = region(emb)
        self.emb = emb.view(batch, E)
        self.pre_ops: List[Op] = [
            Op("zero_grads", torch_op(lambda: (self.G.zero_(), net.Gacc.zero_())), (), writes=(region(self.G), region(net.Gacc))),
            Op("zero_bn_workspace", torch_op(lambda: (self.plan.ws.zero_(), self.plan.ws_b.zero_())), (),
               writes=(region(self.plan.ws), (self.plan.ws.data_ptr() + 1, 0, net.CB), region(self.plan.ws_b),
                       (self.plan.ws_b.data_ptr() + 1, 0, net.CB))),
        ]
        self.loss_ops: List[Op] = []
        if loss == "triplet":
            self.embn = torch.zeros(batch, E, dtype=torch.float32, device=dev)
            self.dembn = torch.zeros(batch, E, dtype=torch.float32, device=dev)
            self._op(self.loss_ops, "l2norm_fwd", lib.fn_l2norm_fwd, _ptr(emb), _ptr(self.embn), batch, E, 1e-10,
                     r=[r_emb], w=[region(self.embn)])
            self._op(self.loss_ops, "triplet_loss", lib.fn_triplet_loss_fwd_bwd, _ptr(self.embn), _ptr(self.dembn), _ptr(self.loss),
                     batch // 3, E, alpha, r=[region(self.embn)], w=[region(self.dembn), region(self.loss)])
            self._op(self.loss_ops, "l2norm_bwd", lib.fn_l2norm_bwd, _ptr(emb), _ptr(self.dembn), _ptr(self.demb), batch, E, 1e-10,
                     r=[r_emb, region(self.dembn)], w=[region(self.demb)])
        else:
            L = net.layers["classifier/logits"]
            Cp, Cr = L.cout, L.cout_real
            self.labels = torch.zeros(batch, dtype=torch.int32, device=dev)
            self.emb_lp = torch.zeros(batch, E, dtype=net.train_dtype, device=dev)
            self.logits = torch.zeros(batch, Cp, dtype=torch.float32, device=dev)
            self.dlogits = torch.zeros(batch, Cp, dtype=net.train_dtype, device=dev)
            rw = region(net.W_train, L.w_off, L.w_off + L.numel)
            rwt = region(net.Wt_train, L.w_off, L.w_off + L.numel)
            rgw = region(self.G, L.w_off, L.w_off + L.numel)
            rgb = region(net.Gacc, L.bias_off - net.bias_lo, L.bias_off - net.bias_lo + L.cout)
            d = self._cls_desc(L)
            d.x, d.w, d.y, d.bias, d.out_f32 = _ptr(self.emb_lp), _ptr(net.W_train, L.w_off), _ptr(self.logits), _ptr(net.P, L.bias_off), 1
            self._op(self.loss_ops, "cast_emb", lib.fn_cast_f32_to_lp, _ptr(emb), _ptr(self.emb_lp), batch * E, self.dt,
                     r=[r_emb], w=[region(self.emb_lp)])
            self._op(self.loss_ops, "conv_fwd:classifier", lib.fn_conv2d_fwd, C.byref(d), keep=(d,),
                     r=[region(self.emb_lp), rw, region(net.P, L.bias_off, L.bias_off + L.cout)], w=[region(self.logits)])
            self._op(self.loss_ops, "softmax_xent", lib.fn_softmax_xent_fwd_bwd, _ptr(self.logits), Cp, _ptr(self.labels), _ptr(self.loss),
                     _ptr(self.dlogits), Cp, _ptr(net.Gacc, L.bias_off - net.bias_lo), batch, Cr, 1.0 / batch, self.dt,
                     r=[region(self.logits), region(self.labels)], w=[region(self.loss), region(self.dlogits), rgb])
            w = self._cls_desc(L)
            w.x, w.y, w.dw = _ptr(self.emb_lp), _ptr(self.dlogits), _ptr(self.G, L.w_off)
            self._op(self.loss_ops, "conv_wgrad:classifier", lib.fn_conv2d_wgrad, C.byref(w), keep=(w,),
                     r=[region(self.emb_lp), region(self.dlogits)], w=[rgw])
            g = self._cls_desc(L)
            g.y, g.w, g.dx, g.out_f32 = _ptr(self.dlogits), _ptr(net.Wt_train, L.w_off), _ptr(self.demb), 1
            self._op(self.loss_ops, "conv_dgrad:classifier", lib.fn_conv2d_dgrad, C.byref(g), keep=(g,),
                     r=[region(self.dlogits), rwt], w=[region(self.demb)])
        self.plan.build_backward(self.demb)
        self.opt_ops: List[Op] = []
        self._op(self.opt_ops, "adam_tick", lib.fn_adam_tick, _ptr(self.hyper), beta1, beta2, w=[region(self.hyper)])
        self._op(self.opt_ops, "adam_keras", lib.fn_adam_keras, _ptr(net.P), _ptr(self.G), _ptr(self.M), _ptr(self.V), _ptr(net.W_train),
                 net.n_kernel, net.n_params, net.n_decay, _ptr(self.hyper), beta1, beta2, epsilon, l2, self.dt,
                 r=[region(self.G), region(self.hyper)], w=[region(net.P), region(self.M), region(self.V), region(net.W_train)])
        self._op(self.opt_ops, "pack_transpose", lib.fn_pack_transpose, _ptr(net.W_train), _ptr(net.Wt_train), _ptr(net.table),
                 len(net.layers), net.max_layer_elems, self.dt, r=[region(net.W_train)], w=[region(net.Wt_train)])
        n_buckets = int(os.environ.get("FACENET_DP_BUCKETS", n_buckets))      # tuning aid: gradient buckets of the data-parallel step
        self.buckets = self._make_buckets(n_buckets) if self.segmented else []
        self.comm_stream = torch.cuda.Stream(device=dev) if self.exchange else None
        self.streams = _streams_for(net, n_streams)
        self.tiles = autotune_convs(self.plan.fwd + self.loss_ops + self.plan.bwd, net)
        self._build_segments()
        self._graph = None
        self._exchange_events: Optional[dict] = None     # set by exchange_profile() around a step

    def _op(self, lst, name, fn, *args, keep=(), r=(), w=()):
        lst.append(Op(name, fn, args, tuple(keep), tuple(r), tuple(w)))

    def _cls_desc(self, L):
        d = _lib.ConvDesc()
        d.N, d.H, d.W, d.Cin, d.OH, d.OW, d.Cout = self.N, 1, 1, L.cin, 1, 1, L.cout
        d.KH = d.KW = d.stride = 1
        d.dtype, d.ld_x, d.ld_y, d.scale = self.dt, L.cin, L.cout, 1.0
        return d

    # ---- data-parallel buckets -----------------------------------------------------------------
    def _make_buckets(self, n_buckets: int) -> List[Tuple[int, int, int]]:
        """[(bwd_op_index_after_which_ready, lo, hi)] over the flat gradient buffer, in backward order."""
        net = self.net
        layers = list(net.layers.values())
        done_at = {}
        for op_idx, li in self.plan.bwd_marks:
            done_at[li] = max(done_at.get(li, 0), op_idx)
        # the classifier (softmax) finishes inside the loss ops, i.e. before backward starts: done_at defaults to 0
        tail = (net.n_decay, net.n_params)
        buckets = parallel.make_buckets([L.w_off for L in layers], [L.numel for L in layers], done_at, net.n_kernel, tail,
                                        len(self.plan.bwd), n_buckets)
        parallel.check_buckets(buckets, net.n_kernel, tail)
        return buckets

    def _allreduce(self, lo: int, hi: int):
        parallel.allreduce_bucket(self.G, lo, hi, self.pg)

    def _build_segments(self):
        """world 1: one schedule for the whole step.  world > 1: the backward is cut where a gradient bucket becomes
        complete; every segment is its own multi-stream schedule (all streams joined at its end), and the bucket's
        all-reduce is issued on the communication stream while the next segment computes."""
        head = self.pre_ops + self.plan.fwd + self.loss_ops
        grp = (lambda ops: group_wgrads(group_convs(ops, self.net), self.net)) if self.group_wgrad else (lambda ops: list(ops))
        self.segments: List[Tuple[Optional[Schedule], Optional[Tuple[int, int]]]] = []
        if not self.segmented:
            chunks = int(os.environ.get("FACENET_WGRAD_CHUNKS", "0"))
            if chunks > 0 and self.group_wgrad:
                # experiment: the weight gradients leave the critical path -- the backward is cut into `chunks` pieces, each piece's
                # weight gradients become grouped launches pinned to a SIDE stream (they only feed the optimiser), everything else
                # stays on the main stream; the dgrad chain is latency-bound at a few % of the wave slots, the side stream fills them
                marks = [i for i, op in enumerate(self.plan.bwd) if op.name.startswith("conv_wgrad:")]
                cuts = [0] + [marks[len(marks) * k // chunks] for k in range(1, chunks)] + [len(self.plan.bwd)]
                ops: List[Op] = []
                for k in range(chunks):
                    ops += grp((head if k == 0 else []) + self.plan.bwd[cuts[k]:cuts[k + 1]])
                ops += self.opt_ops
                for op in ops:
                    op.stream_hint = 1 if op.name.startswith("conv_wgrad") else 0
                self.n_streams = max(2, self.n_streams)
                self.streams = _streams_for(self.net, self.n_streams)
                self.segments.append((Schedule(ops, self.n_streams), None))
                return
            self.segments.append((Schedule(grp(head + self.plan.bwd) + self.opt_ops, self.n_streams), None))
            return
        pos, first = 0, True
        for (ready, lo, hi) in self.buckets:
            a, b = pos, max(pos, ready)
            ops = grp((head if first else []) + self.plan.bwd[a:b])   # a bucket's weight gradients stay inside its segment
            self.segments.append((Schedule(ops, self.n_streams) if ops else None, (lo, hi)))
            pos, first = b, False
        assert pos == len(self.plan.bwd)
        self.segments.append((Schedule(self.opt_ops, 1), None))

    @property
    def step_ops(self) -> List[Op]:
        """The launches one step actually issues, in program order (after weight-gradient grouping)."""
        out: List[Op] = []
        for sched, _ in self.segments:
            if sched is not None:
                out.extend(sched.ops)
        return out

    # ---- one step ------------------------------------------------------------------------------
    def _zero(self):
        self.G.zero_()
        self.net.Gacc.zero_()
        self.plan.ws.zero_()
        self.plan.ws_b.zero_()

    def _run_segments(self, launch: Callable[[int], None]):
        self.net.folded_valid = False        # parameters and moving statistics are about to change
        if not self.segmented:
            launch(0)
            return
        if not self.exchange:                 # force_segments: the segment structure without the exchange
            for i, (sched, _) in enumerate(self.segments):
                if sched is not None:
                    launch(i)
            return
        cur = torch.cuda.current_stream(self.net.device)
        prof = self._exchange_events
        for i, (sched, rng) in enumerate(self.segments[:-1]):
            if sched is not None:
                launch(i)
            ev = torch.cuda.Event()
            ev.record(cur)
            self.comm_stream.wait_event(ev)
            with torch.cuda.stream(self.comm_stream):
                if prof is not None:
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record(self.comm_stream)
                self._allreduce(*rng)
                if prof is not None:
                    b.record(self.comm_stream)
                    prof["buckets"].append((rng, a, b))
        if prof is not None:           # the compute stream has issued the whole backward: what is still exchanging now is exposed
            prof["bwd_end"] = torch.cuda.Event(enable_timing=True)
            prof["bwd_end"].record(cur)
        cur.wait_stream(self.comm_stream)
        launch(len(self.segments) - 1)

    def exchange_profile(self, steps: int = 3) -> dict:
        """Data-parallel runs: per-bucket all-reduce durations (HIP events on the communication stream) and the fraction of
        the exchange that ran while the compute stream was still inside backward (hidden) -- the rest delays the optimiser.
        Runs `steps` ordinary steps (they count as training steps) and reports their mean."""
        if not self.exchange:
            return {"buckets": [], "allreduce_ms": 0.0, "overlapped_frac": None}
        per_bucket, hidden, total = None, 0.0, 0.0
        for _ in range(steps):
            t0 = torch.cuda.Event(enable_timing=True)
            t0.record(torch.cuda.current_stream(self.net.device))
            self._exchange_events = {"buckets": []}
            try:
                self.step()
            finally:
                prof, self._exchange_events = self._exchange_events, None
            torch.cuda.synchronize(self.net.device)
            t_end = t0.elapsed_time(prof["bwd_end"])
            ms = []
            for (rng, a, b) in prof["buckets"]:
                ta, tb = t0.elapsed_time(a), t0.elapsed_time(b)
                ms.append(tb - ta)
                total += tb - ta
                hidden += max(0.0, min(tb, t_end) - min(ta, t_end))
            per_bucket = ms if per_bucket is None else [x + y for x, y in zip(per_bucket, ms)]
        return {"buckets": [{"elements": int(hi - lo), "mbytes": round(4e-6 * (hi - lo), 2), "allreduce_ms": round(m / steps, 4)}
                            for (_, lo, hi), m in zip(self.buckets, per_bucket)],
                "allreduce_ms": round(total / steps, 4), "overlapped_frac": round(hidden / total, 4) if total > 0 else None}

    def step_eager(self):
        """zero -> forward -> loss -> backward (+ bucketed all-reduce) -> Adam; the loss stays on device."""
        self._run_segments(lambda i: run_schedule(self.segments[i][0], self.streams))

    def capture(self):
        """Capture every segment into a HIP graph (multi-stream edges become graph dependencies); with world_size > 1
        the all-reduces are issued between graph launches on the communication stream.

        Captured schedules span at most 2 streams.  Ending a capture whose fork / join pattern spans 3 or more streams
        segfaults inside hipStreamEndCapture under torch.cuda.graph on ROCm 7.0/7.2 (MI355X; 2 streams capture and replay
        correctly, and a single in-order stream replays fastest anyway: DESIGN.md section 5), so a wider trainer is refused
        here instead of being offered a way to crash a process that has initialised the GPU.  Eager replay
        (``step_eager``) keeps the requested width."""
        if self.n_streams > 2:
            raise ValueError(f"captured schedules span at most 2 streams (got n_streams={self.n_streams}); build the Trainer with "
                             f"n_streams <= 2 or replay eagerly (step_eager)")
        # The warm-up below is a full training step on whatever the image buffer holds.  Training state is snapshotted and
        # restored around it, so capture() followed by n steps equals n eager steps (Adam's t, the moving statistics and the
        # parameters are untouched; the reference's fit() has no uncounted step either).
        net = self.net
        saved = [t.clone() for t in (net.P, net.S_mean, net.S_var, self.M, self.V, self.hyper)]
        self.step_eager()           # warm-up: first-call attribute set-up, allocator
        torch.cuda.synchronize(net.device)
        for t, s in zip((net.P, net.S_mean, net.S_var, self.M, self.V, self.hyper), saved):
            t.copy_(s)
        net.folded_valid = False
        net.refresh_packs()
        torch.cuda.synchronize(net.device)
        runner = GraphRunner(self.net.device)
        graphs, keep = [], []
        for (sched, _) in self.segments:
            if sched is None:
                graphs.append(None)
                continue
            evs = make_events(sched)     # events owned by this capture only
            keep.append(evs)
            graphs.append(runner.capture(lambda sched=sched, evs=evs: run_schedule(sched, self.streams, evs)))
        self._graph = (runner, graphs, keep)

    def step(self):
        if self._graph is None:
            return self.step_eager()
        graphs = self._graph[1]
        self._run_segments(lambda i: graphs[i].replay())

    # ---- host conveniences -----------------------------------------------------------------------
    def set_images(self, images: torch.Tensor, labels: Optional[torch.Tensor] = None):
        self.plan.images.copy_(images.to(self.net.device, non_blocking=True))
        if labels is not None:
            if self.loss_kind != "softmax":
                raise ValueError("labels are only used by the softmax loss")
            labels = torch.as_tensor(labels)
            nc = self.net.layers["classifier/logits"].cout_real
            if labels.numel() != self.N or int(labels.min()) < 0 or int(labels.max()) >= nc:
                # TF's sparse softmax cross-entropy rejects out-of-range class indices; the kernel indexes logits[label]
                raise ValueError(f"labels must be {self.N} class indices in [0, {nc}), got range [{int(labels.min())}, {int(labels.max())}]")
            self.labels.copy_(labels.to(device=self.net.device, dtype=torch.int32))

    # ---- checkpoints (apps/train_softmax.py:68-78,105; SURVEY.md section 5: optimiser state) -----------------------------
    def averaged_moving_stats(self) -> Tuple[torch.Tensor, torch.Tensor]:
        """BatchNorm moving statistics as MirroredStrategy reads them: every replica keeps its own (per-replica batch
        statistics), a read or save aggregates them with MEAN (apps/train_softmax_tf2_gpus.py:49).  Collective:
        every rank must call it."""
        mean, var = self.net.S_mean.clone(), self.net.S_var.clone()
        if self.world > 1:
            import torch.distributed as dist
            for t in (mean, var):
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.pg)
                t.mul_(1.0 / self.world)
        return mean, var

    def state_dict(self, epoch: int = 0) -> "Dict[str, np.ndarray]":
        """Model variables under their Keras names (replica-averaged moving statistics) + the Keras-Adam slots ``Adam/<var>/m``,
        ``Adam/<var>/v``, ``Adam/iter`` and the schedule position: everything ``fit`` needs to resume."""
        from . import keras_names
        net = self.net
        rep = int(net.cfg["block8_1"]["repeat"])
        out = {k: v.numpy() for k, v in net.keras_variables(self.averaged_moving_stats()).items()}
        table = dict((i, k) for k, i in keras_names.keras_variable_table(net.layers, rep))
        for slot, buf in (("m", self.M), ("v", self.V)):
            for key, t in net.export_keras_grads(buf).items():
                out[keras_names.optimizer_slot_names(table[key])[0 if slot == "m" else 1]] = t.numpy()
        out["Adam/iter:0"] = np.asarray(self.iterations, dtype=np.int64)
        out["Adam/learning_rate:0"] = np.asarray(self.hyper[0].item(), dtype=np.float32)
        out["epoch"] = np.asarray(int(epoch), dtype=np.int64)
        return out

    def save_checkpoint(self, path, epoch: int = 0):
        sd = self.state_dict(epoch)          # collective when world > 1; rank 0 writes
        if self.world == 1 or int(os.environ.get("RANK", "0")) == 0:
            np.savez(path, **sd)

    def load_checkpoint(self, path) -> int:
        """Restore parameters, moving statistics and optimiser state; returns the stored epoch."""
        from . import keras_names
        net = self.net
        with np.load(path, allow_pickle=False) as z:
            sd = {k: z[k] for k in z.files}
        net.load_keras_params({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items() if not k.startswith("Adam/") and k != "epoch"})
        if "Adam/iter:0" in sd:
            rep = int(net.cfg["block8_1"]["repeat"])
            for slot, buf in ((0, self.M), (1, self.V)):
                tmp = {}
                for k, i in keras_names.keras_variable_table(net.layers, rep):
                    if i.endswith(("/moving_mean", "/moving_variance")):
                        continue
                    tmp[i] = torch.from_numpy(sd[keras_names.optimizer_slot_names(k)[slot]])
                buf.copy_(net.flat_from_keras(tmp))
            self.hyper[0:1].fill_(float(sd["Adam/learning_rate:0"]))
            self.iterations = int(sd["Adam/iter:0"])
        return int(sd.get("epoch", 0))

    @property
    def iterations(self) -> int:
        """Keras' ``optimizer.iterations``: optimiser steps taken so far (an int32 word on the device, bumped by fn_adam_tick)."""
        return int(self.hyper.view(torch.int32)[4].item())

    @iterations.setter
    def iterations(self, t: int):
        """Sets the step count and the beta powers that belong to it (what the NEXT tick will overwrite with t + 1)."""
        if t < 0:
            raise ValueError(f"Adam iteration count must be >= 0, got {t}")
        self.hyper.view(torch.int32)[4:5].fill_(int(t))
        self.hyper[1:3].copy_(torch.tensor(adam_beta_powers(t, self.beta1, self.beta2)))

    def reset_optimizer(self, lr: Optional[float] = None):
        """Adam as freshly constructed: zero moments, t = 0."""
        self.M.zero_()
        self.V.zero_()
        self.iterations = 0
        if lr is not None:
            self.set_learning_rate(lr)

    def set_learning_rate(self, lr: float):
        self.hyper[0:1].fill_(float(lr))     # device write: visible to the next graph replay

    def loss_value(self) -> float:
        return float(self.loss[0].item())


class TripletMiner:
    """Embeds a PxK pool with the inference path, selects triplets on device and assembles the train batch."""

    def __init__(self, net: Network, pool_size: int, labels: Sequence[int], nrof_triplets: int, alpha: float = 0.2, seed: int = 0,
                 semi_hard: bool = False, n_streams: int = 1, group: bool = True):
        self.group = group
        self.net, self.n, self.T, self.alpha, self.seed, self.semi_hard = net, pool_size, nrof_triplets, alpha, seed, semi_hard
        dev = net.device
        self.n_streams = n_streams
        lab = np.asarray(list(labels))
        if lab.shape != (pool_size,):
            raise ValueError(f"labels must have one entry per pool image ({pool_size}), got {lab.shape}")
        _, counts = np.unique(lab, return_counts=True)
        pairs = int((counts * (counts - 1) // 2).sum())
        # every selected triplet needs its own anchor-positive pair and a negative of another identity: a pool that cannot
        # supply them would leave triplet slots unwritten (the gather would reuse stale indices)
        if len(counts) < 2 or pairs < nrof_triplets:
            raise ValueError(f"the pool holds {len(counts)} identities and {pairs} anchor-positive pairs; {nrof_triplets} triplets need "
                             f">= 2 identities and >= {nrof_triplets} pairs")
        if pairs > 1 << 15:
            raise ValueError(f"{pairs} anchor-positive pairs exceed the 32768 the on-device ranking handles; use more identities with fewer images each")
        self.plan = net.plan(pool_size, training=False)
        E = net.E
        self.emb = self.plan.embedding.buf.act.view(pool_size, E)
        self.embn = torch.zeros(pool_size, E, dtype=torch.float32, device=dev)
        self.dist = torch.zeros(pool_size, pool_size, dtype=torch.float32, device=dev)
        self.labels = torch.as_tensor(list(labels), dtype=torch.int32).to(dev)
        self.triplets = torch.zeros(nrof_triplets, 3, dtype=torch.int32, device=dev)
        qmax = pool_size * (pool_size - 1) // 2
        self.info = torch.zeros(8 + 5 * qmax, dtype=torch.int32, device=dev)
        self.ops: List[Op] = []
        self.sched: Optional[Schedule] = None
        self.streams = _streams_for(net, n_streams)

    def build(self, train_images: torch.Tensor):
        """train_images: the uint8 [3T,H,W,3] input buffer of the training plan (filled by the gather)."""
        net, lib, n, E = self.net, self.net.lib, self.n, self.net.E
        o = self.ops
        bytes_per = train_images[0].numel()
        o.append(Op("fold_bn", lambda st: (net.refresh_folded(st), 0)[1], (),
                    reads=(region(net.P), region(net.S_mean), region(net.S_var)), writes=(region(net.W_infer), region(net.fold_bias))))
        net.refresh_folded(net.stream())          # the inference pack must exist before launches are timed
        self.tiles = autotune_convs(self.plan.fwd, net)
        o.extend(self.plan.fwd)
        o.append(Op("l2norm_fwd", lib.fn_l2norm_fwd, (_ptr(self.emb), _ptr(self.embn), n, E, 1e-10),
                    reads=(region(self.plan.embedding.buf.act),), writes=(region(self.embn),)))
        o.append(Op("pairwise_sqdist", lib.fn_pairwise_sqdist, (_ptr(self.embn), _ptr(self.embn), _ptr(self.dist), None, n, n, E, 2),
                    reads=(region(self.embn),), writes=(region(self.dist),)))
        o.append(Op("select_triplets", lib.fn_select_triplets, (_ptr(self.dist), _ptr(self.labels), n, self.alpha, self.T, self.seed,
                                                                 1 if self.semi_hard else 0, _ptr(self.triplets), _ptr(self.info)),
                    reads=(region(self.dist), region(self.labels)), writes=(region(self.triplets), region(self.info))))
        o.append(Op("gather_images", lib.fn_gather_images, (_ptr(self.plan.images), _ptr(self.triplets), _ptr(train_images), 3 * self.T,
                                                             bytes_per),
                    reads=(region(self.plan.images), region(self.triplets)), writes=(region(train_images),)))
        self.ops = group_convs(o, net) if self.group else o
        self.sched = Schedule(self.ops, self.n_streams)

    def run(self, events=None):
        run_schedule(self.sched, self.streams, events)
