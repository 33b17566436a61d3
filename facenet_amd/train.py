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
from typing import Callable, List, Optional, Sequence, Tuple

import torch

from . import _lib
from .engine import BN_EPS, L2_WEIGHT, Lowering, Network, Op, _pad8, _ptr


class GraphRunner:
    """Capture a launch list into HIP graphs (one per segment) and replay them.

    Segments exist so that collectives issued between them stay outside the captured graphs."""

    def __init__(self, device: torch.device):
        self.device = device
        self.graphs: List[torch.cuda.CUDAGraph] = []
        self.pool = None

    def capture(self, fn: Callable[[int], None]) -> torch.cuda.CUDAGraph:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, pool=self.pool):
            fn(torch.cuda.current_stream(self.device).cuda_stream)
        if self.pool is None:
            self.pool = g.pool()
        self.graphs.append(g)
        return g


class Trainer:
    def __init__(self, net: Network, batch: int, loss: str = "triplet", alpha: float = 0.2, lr: float = 0.05, beta1: float = 0.9,
                 beta2: float = 0.999, epsilon: float = 0.1, l2: float = L2_WEIGHT, world_size: int = 1, process_group=None,
                 n_buckets: int = 6):
        if loss not in ("triplet", "softmax"):
            raise ValueError(f"unknown loss {loss!r}")
        if loss == "triplet" and batch % 3:
            raise ValueError("triplet batches are laid out (a,p,n,...): batch must be a multiple of 3")
        if loss == "softmax" and net.nrof_classes is None:
            raise ValueError("softmax training needs Network(nrof_classes=...)")
        self.net, self.N, self.loss_kind, self.alpha = net, batch, loss, alpha
        self.beta1, self.beta2, self.eps, self.l2 = beta1, beta2, epsilon, l2
        self.world, self.pg = world_size, process_group
        dev, E, lib = net.device, net.E, net.lib
        self.lib = lib
        if net.G is None:
            net.G = torch.zeros(net.n_params, dtype=torch.float32, device=dev)
        self.G = net.G
        self.M = torch.zeros_like(self.G)
        self.V = torch.zeros_like(self.G)
        # hyper = {lr, beta1^t, beta2^t, grad_scale}; lives on device so HIP-graph replays see LR changes
        self.hyper = torch.tensor([lr, 1.0, 1.0, 1.0 / world_size], dtype=torch.float32, device=dev)
        self.loss = torch.zeros(1, dtype=torch.float32, device=dev)
        self.plan: Lowering = net.plan(batch, training=True)
        self.demb = torch.zeros(batch, E, dtype=torch.float32, device=dev)
        self.dt = _lib.dtype_code(net.train_dtype)
        emb = self.plan.embedding.buf.act
        self.emb = emb.view(batch, E)
        self.loss_ops: List[Op] = []
        self.post_bwd_ops: List[Op] = []
        if loss == "triplet":
            self.embn = torch.zeros(batch, E, dtype=torch.float32, device=dev)
            self.dembn = torch.zeros(batch, E, dtype=torch.float32, device=dev)
            self._op(self.loss_ops, "l2norm_fwd", lib.fn_l2norm_fwd, _ptr(emb), _ptr(self.embn), batch, E, 1e-10)
            self._op(self.loss_ops, "triplet_loss", lib.fn_triplet_loss_fwd_bwd, _ptr(self.embn), _ptr(self.dembn), _ptr(self.loss),
                     batch // 3, E, alpha)
            self._op(self.loss_ops, "l2norm_bwd", lib.fn_l2norm_bwd, _ptr(emb), _ptr(self.dembn), _ptr(self.demb), batch, E, 1e-10)
        else:
            L = net.layers["classifier/logits"]
            Cp, Cr = L.cout, L.cout_real
            self.labels = torch.zeros(batch, dtype=torch.int32, device=dev)
            self.emb_lp = torch.zeros(batch, E, dtype=net.train_dtype, device=dev)
            self.logits = torch.zeros(batch, Cp, dtype=torch.float32, device=dev)
            self.dlogits = torch.zeros(batch, Cp, dtype=net.train_dtype, device=dev)
            d = self._cls_desc(L)
            d.x, d.w, d.y, d.bias, d.out_f32 = _ptr(self.emb_lp), _ptr(net.W_train, L.w_off), _ptr(self.logits), _ptr(net.P, L.bias_off), 1
            self._op(self.loss_ops, "cast_emb", lib.fn_cast_f32_to_lp, _ptr(emb), _ptr(self.emb_lp), batch * E, self.dt)
            self._op(self.loss_ops, "conv_fwd:classifier", lib.fn_conv2d_fwd, C.byref(d), keep=(d,))
            self._op(self.loss_ops, "softmax_xent", lib.fn_softmax_xent_fwd_bwd, _ptr(self.logits), Cp, _ptr(self.labels), _ptr(self.loss),
                     _ptr(self.dlogits), Cp, _ptr(self.G, L.bias_off), batch, Cr, 1.0 / batch, self.dt)
            w = self._cls_desc(L)
            w.x, w.y, w.dw = _ptr(self.emb_lp), _ptr(self.dlogits), _ptr(self.G, L.w_off)
            self._op(self.loss_ops, "conv_wgrad:classifier", lib.fn_conv2d_wgrad, C.byref(w), keep=(w,))
            g = self._cls_desc(L)
            g.y, g.w, g.dx, g.out_f32 = _ptr(self.dlogits), _ptr(net.Wt_train, L.w_off), _ptr(self.demb), 1
            self._op(self.loss_ops, "conv_dgrad:classifier", lib.fn_conv2d_dgrad, C.byref(g), keep=(g,))
        self.plan.build_backward(self.demb)
        self.opt_ops: List[Op] = []
        self._op(self.opt_ops, "adam_tick", lib.fn_adam_tick, _ptr(self.hyper), beta1, beta2)
        self._op(self.opt_ops, "adam_keras", lib.fn_adam_keras, _ptr(net.P), _ptr(self.G), _ptr(self.M), _ptr(self.V), _ptr(net.W_train),
                 net.n_kernel, net.n_params, net.n_decay, _ptr(self.hyper), beta1, beta2, epsilon, l2, self.dt)
        self._op(self.opt_ops, "pack_transpose", lib.fn_pack_transpose, _ptr(net.W_train), _ptr(net.Wt_train), _ptr(net.table),
                 len(net.layers), net.max_layer_elems, self.dt)
        self.buckets = self._make_buckets(n_buckets) if world_size > 1 else []
        self.comm_stream = torch.cuda.Stream(device=dev) if world_size > 1 else None
        self._graph = None

    def _op(self, lst, name, fn, *args, keep=()):
        lst.append(Op(name, fn, args, tuple(keep)))

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
        for L in layers:   # classifier (softmax) finishes inside the loss ops, i.e. before backward starts
            done_at.setdefault(L.index, 0)
        total = net.n_kernel
        target = total / n_buckets
        buckets, hi, acc, ready = [], net.n_kernel, 0, 0
        for L in reversed(layers):
            acc += L.numel
            ready = max(ready, done_at[L.index])
            if acc >= target or L.index == 0:
                buckets.append((ready, L.w_off, hi))
                hi, acc = L.w_off, 0
        # everything a bucket needs must be finished: make readiness monotone in issue order
        out, r = [], 0
        for (rd, lo, h) in buckets:
            r = max(r, rd)
            out.append((r, lo, h))
        out.append((len(self.plan.bwd), net.n_decay, net.n_params))   # betas + biases, after the whole backward
        return out

    def _allreduce(self, lo: int, hi: int):
        import torch.distributed as dist
        dist.all_reduce(self.G[lo:hi], op=dist.ReduceOp.SUM, group=self.pg)

    # ---- one step ------------------------------------------------------------------------------
    def _zero(self):
        self.G.zero_()
        self.plan.ws.zero_()

    def _run_compute(self, stream: int, lo: int, hi: int):
        Lowering.run_ops(self.plan.bwd, stream, lo, hi)

    def step_eager(self):
        """zero -> forward -> loss -> backward (+ bucketed all-reduce) -> Adam; returns nothing (loss stays on device)."""
        net = self.net
        st = net.stream()
        self._zero()
        Lowering.run_ops(self.plan.fwd, st)
        Lowering.run_ops(self.loss_ops, st)
        if self.world == 1:
            Lowering.run_ops(self.plan.bwd, st)
        else:
            cur = torch.cuda.current_stream(net.device)
            pos = 0
            for (ready, lo, hi) in self.buckets:
                if ready > pos:
                    Lowering.run_ops(self.plan.bwd, st, pos, ready)
                    pos = ready
                ev = torch.cuda.Event()
                ev.record(cur)
                self.comm_stream.wait_event(ev)
                with torch.cuda.stream(self.comm_stream):
                    self._allreduce(lo, hi)
            if pos < len(self.plan.bwd):
                Lowering.run_ops(self.plan.bwd, st, pos, None)
            cur.wait_stream(self.comm_stream)
        Lowering.run_ops(self.opt_ops, st)

    def capture(self):
        """Capture the step into HIP graph(s).  With world_size > 1 the backward is cut at bucket boundaries and
        the all-reduces are issued between graph launches on the communication stream."""
        net = self.net
        self.step_eager()           # warm-up: first-call attribute set-up, allocator
        torch.cuda.synchronize(net.device)
        runner = GraphRunner(net.device)
        segs: List[Tuple[torch.cuda.CUDAGraph, Optional[Tuple[int, int]]]] = []
        if self.world == 1:
            def whole(st):
                self._zero()
                Lowering.run_ops(self.plan.fwd, st)
                Lowering.run_ops(self.loss_ops, st)
                Lowering.run_ops(self.plan.bwd, st)
                Lowering.run_ops(self.opt_ops, st)
            segs.append((runner.capture(whole), None))
        else:
            pos = 0
            first = True
            for (ready, lo, hi) in self.buckets:
                a, b = pos, max(pos, ready)

                def seg(st, a=a, b=b, first=first):
                    if first:
                        self._zero()
                        Lowering.run_ops(self.plan.fwd, st)
                        Lowering.run_ops(self.loss_ops, st)
                    Lowering.run_ops(self.plan.bwd, st, a, b)
                if first or b > a:
                    segs.append((runner.capture(seg), (lo, hi)))
                else:
                    segs.append((None, (lo, hi)))
                pos, first = b, False
            segs.append((runner.capture(lambda st: Lowering.run_ops(self.opt_ops, st)), None))
        self._graph = (runner, segs)

    def step(self):
        if self._graph is None:
            return self.step_eager()
        _, segs = self._graph
        if self.world == 1:
            segs[0][0].replay()
            return
        cur = torch.cuda.current_stream(self.net.device)
        for g, rng in segs[:-1]:
            if g is not None:
                g.replay()
            ev = torch.cuda.Event()
            ev.record(cur)
            self.comm_stream.wait_event(ev)
            with torch.cuda.stream(self.comm_stream):
                self._allreduce(*rng)
        cur.wait_stream(self.comm_stream)
        segs[-1][0].replay()

    # ---- host conveniences -----------------------------------------------------------------------
    def set_images(self, images: torch.Tensor, labels: Optional[torch.Tensor] = None):
        self.plan.images.copy_(images.to(self.net.device, non_blocking=True))
        if labels is not None:
            self.labels.copy_(labels.to(device=self.net.device, dtype=torch.int32))

    def set_learning_rate(self, lr: float):
        self.hyper[0:1].fill_(float(lr))     # device write: visible to the next graph replay

    def loss_value(self) -> float:
        return float(self.loss.item())


class TripletMiner:
    """Embeds a PxK pool with the inference path, selects triplets on device and assembles the train batch."""

    def __init__(self, net: Network, pool_size: int, labels: Sequence[int], nrof_triplets: int, alpha: float = 0.2, seed: int = 0,
                 semi_hard: bool = False):
        self.net, self.n, self.T, self.alpha, self.seed, self.semi_hard = net, pool_size, nrof_triplets, alpha, seed, semi_hard
        dev, lib = net.device, net.lib
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

    def build(self, train_images: torch.Tensor):
        """train_images: the uint8 [3T,H,W,3] input buffer of the training plan (filled by the gather)."""
        net, lib, n, E = self.net, self.net.lib, self.n, self.net.E
        o = self.ops
        bytes_per = train_images[0].numel()
        o.append(Op("fold_bn", lambda st: (net.refresh_folded(st), 0)[1], ()))
        o.extend(self.plan.fwd)
        o.append(Op("l2norm_fwd", lib.fn_l2norm_fwd, (_ptr(self.emb), _ptr(self.embn), n, E, 1e-10)))
        o.append(Op("pairwise_sqdist", lib.fn_pairwise_sqdist, (_ptr(self.embn), _ptr(self.embn), _ptr(self.dist), None, n, n, E, 2)))
        o.append(Op("select_triplets", lib.fn_select_triplets, (_ptr(self.dist), _ptr(self.labels), n, self.alpha, self.T, self.seed,
                                                                 1 if self.semi_hard else 0, _ptr(self.triplets), _ptr(self.info))))
        o.append(Op("gather_images", lib.fn_gather_images, (_ptr(self.plan.images), _ptr(self.triplets), _ptr(train_images), 3 * self.T,
                                                             bytes_per)))

    def run(self, stream: Optional[int] = None):
        Lowering.run_ops(self.ops, self.net.stream() if stream is None else stream)
