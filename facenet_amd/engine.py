"""Static-plan execution engine for the Inception-ResNet-v1 hot path on MI355X.

The reference runs this path through Keras/TensorFlow's graph runtime
(facenet/models/inception_resnet_v1.py:380-494 built by ``self(input_shape)``, trained by
``network.fit`` at apps/train_softmax.py:95-104).  Here the network is lowered ONCE, for a fixed
batch size, into a flat list of C-ABI kernel launches over pre-allocated HBM buffers:

  * every activation tensor lives for the whole step (288 GB of HBM: nothing is recomputed or
    re-allocated); towers write straight into channel slices of their concat buffer;
  * parameters, gradients and Adam state are single flat fp32 buffers (one fused optimiser launch,
    contiguous all-reduce buckets); the MFMA kernels read low-precision packs of the same layout;
  * the launch list is replayed eagerly or captured into one HIP graph (engine.GraphRunner) -
    HIP streams and graphs instead of a tracing compiler;
  * backward is derived here, op by op, in reverse order of the forward records (no autograd).

PyTorch supplies device memory, streams and torch.distributed only.
"""
from __future__ import annotations

import ctypes as C
import math
import os
from collections import OrderedDict
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib
from ._lib import ConvDesc
from .schedule import Op, Region, region

BN_EPS = 1e-3        # Keras default (inception_resnet_v1.py:57-58 commented out)
BN_MOMENTUM = 0.99   # Keras default
L2_WEIGHT = 5e-4     # inception_resnet_v1.py:65
STAT_REPLICAS = 16   # max BN-statistic accumulator replicas (engine-internal)

DEFAULT_CONFIG = {   # inception_resnet_v1.py:13-43
    "reduction_a": {"filters": [[384], [192, 192, 256]]},
    "reduction_b": {"filters": [[256, 384], [256, 256], [256, 256, 256]]},
    "block35": {"repeat": 5, "scale": 0.17, "activation": "relu"},
    "block17": {"repeat": 10, "scale": 0.10, "activation": "relu"},
    "block8_1": {"repeat": 5, "scale": 0.2, "activation": "relu"},
    "block8_2": {"scale": 1.0, "activation": None},
    "output": {"size": 512},
}


# tower declarations (name, filters, kernel[, stride, padding]) of the residual and reduction blocks
# (inception_resnet_v1.py:83-259 and :262-377); shared by the full network and by BlockNetwork
BLOCK_TOWERS = {
    "block35": ([[("Conv2d_1x1", 32, (1, 1))],
                 [("Conv2d_0a_1x1", 32, (1, 1)), ("Conv2d_0b_3x3", 32, (3, 3))],
                 [("Conv2d_0a_1x1", 32, (1, 1)), ("Conv2d_0b_3x3", 32, (3, 3)), ("Conv2d_0c_3x3", 32, (3, 3))]], 256),
    "block17": ([[("Conv2d_1x1", 128, (1, 1))],
                 [("Conv2d_0a_1x1", 128, (1, 1)), ("Conv2d_0b_1x7", 128, (1, 7)), ("Conv2d_0c_7x1", 128, (7, 1))]], 896),
    "block8": ([[("Conv2d_1x1", 192, (1, 1))],
                [("Conv2d_0a_1x1", 192, (1, 1)), ("Conv2d_0b_1x3", 192, (1, 3)), ("Conv2d_0c_3x1", 192, (3, 1))]], 1792),
}


def reduction_towers(kind: str, filters):
    if kind == "reduction_a":
        fa = filters
        return [[("Conv2d_1a_3x3", fa[0][0], (3, 3), 2, "valid")],
                [("Conv2d_0a_1x1", fa[1][0], (1, 1), 1, "same"), ("Conv2d_0b_3x3", fa[1][1], (3, 3), 1, "same"),
                 ("Conv2d_1a_3x3", fa[1][2], (3, 3), 2, "valid")]]
    fb = filters
    return [[("Conv2d_0a_1x1", fb[0][0], (1, 1), 1, "same"), ("Conv2d_1a_3x3", fb[0][1], (3, 3), 2, "valid")],
            [("Conv2d_0a_1x1", fb[1][0], (1, 1), 1, "same"), ("Conv2d_1a_3x3", fb[1][1], (3, 3), 2, "valid")],
            [("Conv2d_0a_1x1", fb[2][0], (1, 1), 1, "same"), ("Conv2d_0b_3x3", fb[2][1], (3, 3), 1, "same"),
             ("Conv2d_1a_3x3", fb[2][2], (3, 3), 2, "valid")]]



def _pad8(c: int) -> int:
    return (c + 7) // 8 * 8


# ------------------------------------------------------------------------------------------------
# declarations
# ------------------------------------------------------------------------------------------------
@dataclass
class Layer:
    name: str
    cin: int          # padded to a multiple of 8
    cin_real: int
    cout: int
    kh: int
    kw: int
    stride: int
    pad_h: int
    pad_w: int
    has_bn: bool
    has_bias: bool
    dense: bool = False
    cout_real: int = -1   # un-padded output channels (classifier only differs)
    w_off: int = -1       # element offset of [cout][kh][kw][cin] in the flat parameter buffer
    bias_off: int = -1    # element offset of the bias in the flat parameter buffer
    bn_off: int = -1      # offset in the global BatchNorm channel space
    index: int = -1

    @property
    def ktot(self) -> int:
        return self.kh * self.kw * self.cin

    @property
    def numel(self) -> int:
        return self.cout * self.ktot


class Buf:
    """One NHWC activation tensor: raw conv output, activated output and (training) gradient."""

    def __init__(self, name: str, N: int, H: int, W: int, Cc: int, bn_off: Optional[int] = None):
        self.name, self.N, self.H, self.W, self.C = name, N, H, W, Cc
        self.bn_off = bn_off
        self.act: Optional[torch.Tensor] = None
        self.raw: Optional[torch.Tensor] = None
        self.grad: Optional[torch.Tensor] = None
        self.grad_ranges: List[Tuple[int, int]] = []
        self.f32 = False

    @property
    def M(self) -> int:
        return self.N * self.H * self.W

    def full(self) -> "Slice":
        return Slice(self, 0, self.C)

    def sl(self, c0: int, c: int) -> "Slice":
        return Slice(self, c0, c)


@dataclass
class Slice:
    buf: Buf
    c0: int
    C: int


@dataclass
class Rec:
    kind: str
    layer: Optional[Layer] = None
    x: Optional[Slice] = None
    y: Optional[Slice] = None
    extra: dict = field(default_factory=dict)


def _ptr(t: torch.Tensor, elem_off: int = 0) -> int:
    return t.data_ptr() + elem_off * t.element_size()


# ------------------------------------------------------------------------------------------------
# the network
# ------------------------------------------------------------------------------------------------
class Network:
    """Parameters + topology of Inception-ResNet-v1; ``plan()`` lowers it for one batch size."""

    def __init__(self, embedding_size: int = 512, config: Optional[dict] = None, image_size: int = 160,
                 normalization: int = 0, nrof_classes: Optional[int] = None, device: str = "cuda",
                 train_dtype: torch.dtype = torch.bfloat16, infer_dtype: torch.dtype = torch.float16, seed: int = 0,
                 allocate: bool = True):
        self.cfg = {k: (dict(v) if isinstance(v, dict) else v) for k, v in DEFAULT_CONFIG.items()}
        if config:
            for k, v in config.items():
                self.cfg[k] = v
        self.E = int(embedding_size)
        self.image_size = int(image_size)
        self.normalization = int(normalization)
        self.nrof_classes = nrof_classes
        self.device = torch.device(device)
        self.train_dtype, self.infer_dtype = train_dtype, infer_dtype
        # allocate=False: host-side description only (layer table, flat layout, variable counts) -- nothing is computed
        if allocate and self.device.type != "cuda":
            raise _lib.FacenetHipError("facenet_amd runs on a HIP device only (no CPU fallback)")
        self.lib = _lib.load() if allocate else None

        self.layers: "OrderedDict[str, Layer]" = OrderedDict()
        self.buf_bn: Dict[str, int] = {}     # buffer name -> BN channel offset
        self.CB = 0                          # size of the global BN channel space
        self.G = None
        self._declare()
        self._layout()
        if allocate:
            self._alloc_params(seed)

    # ---- topology (written from inception_resnet_v1.py; independent of oracle/) ------------------
    def _topology(self, g: "Lowering"):
        cfg = self.cfg
        s = self.image_size
        x = g.input(s, s)
        x = g.cbr("conv2d/Conv2d_1a_3x3", x, 32, (3, 3), 2, "valid", cin_real=3)       # :388
        x = g.cbr("conv2d/Conv2d_2a_3x3", x, 32, (3, 3), 1, "valid")                   # :395
        x = g.cbr("conv2d/Conv2d_2b_3x3", x, 64, (3, 3), 1, "valid")                   # :402 ('valid' in this fork)
        x = g.maxpool("conv2d/MaxPool_3a_3x3", x)                                      # :409
        x = g.cbr("conv2d/Conv2d_3b_1x1", x, 80, (1, 1), 1, "valid")                   # :410
        x = g.cbr("conv2d/Conv2d_4a_3x3", x, 192, (3, 3), 1, "valid")                  # :417
        x = g.cbr("conv2d/Conv2d_4b_3x3", x, 256, (3, 3), 2, "valid")                  # :424
        b35, up35 = BLOCK_TOWERS["block35"]
        for i in range(cfg["block35"]["repeat"]):                                      # :433-435 ; relu hard-coded :88
            x = g.block(f"block35/{i}", x, b35, up35, cfg["block35"]["scale"], True)
        x = g.reduction("reduction_a", x, reduction_towers("reduction_a", cfg["reduction_a"]["filters"]))   # :262-307
        b17, up17 = BLOCK_TOWERS["block17"]
        for i in range(cfg["block17"]["repeat"]):                                      # :441-443 ; relu hard-coded :158
            x = g.block(f"block17/{i}", x, b17, up17, cfg["block17"]["scale"], True)
        x = g.reduction("reduction_b", x, reduction_towers("reduction_b", cfg["reduction_b"]["filters"]))   # :310-377
        b8, up8 = BLOCK_TOWERS["block8"]
        for i in range(cfg["block8_1"]["repeat"]):                                     # :449-451 ; activation from config :213
            x = g.block(f"block8/{i}", x, b8, up8, cfg["block8_1"]["scale"], bool(cfg["block8_1"]["activation"]))
        x = g.block("block8_2", x, b8, up8, cfg["block8_2"]["scale"], bool(cfg["block8_2"]["activation"]))  # :453
        return g.head(x, self.E)                                                         # :459-468

    def _declare(self):
        g = Lowering(self, N=1, training=False, declare=True)
        self._topology(g)
        if self.nrof_classes is not None:   # apps/train_softmax.py:57-63
            L = self._declare_layer("classifier/logits", self.E, self.E, _pad8(self.nrof_classes), 1, 1, 1, 0, 0, has_bn=False,
                                    has_bias=True, dense=True)
            L.cout_real = self.nrof_classes     # padded rows stay exactly zero (zero init, zero gradient)

    def _declare_layer(self, name, cin, cin_real, cout, kh, kw, stride, pad_h, pad_w, has_bn, has_bias, dense=False) -> Layer:
        if name in self.layers:
            return self.layers[name]
        L = Layer(name, cin, cin_real, cout, kh, kw, stride, pad_h, pad_w, has_bn, has_bias, dense, cout_real=cout, index=len(self.layers))
        self.layers[name] = L
        return L

    def _layout(self):
        off = 0
        for L in self.layers.values():
            L.w_off = off
            off += L.numel
            assert L.numel % 8 == 0
        self.n_kernel = off
        self.n_decay = (off + 3) // 4 * 4            # coupled-L2 region of the flat buffer
        self.beta_base = self.n_decay
        off = self.beta_base + self.CB
        for L in self.layers.values():
            if L.has_bias:
                L.bias_off = off
                off += L.cout
        self.n_params = (off + 3) // 4 * 4
        self.bias_lo = self.beta_base + self.CB          # [bias_lo, n_params): the biases
        self.max_layer_elems = max(L.numel for L in self.layers.values())

    def _alloc_params(self, seed: int):
        dev = self.device
        self.P = torch.zeros(self.n_params, dtype=torch.float32, device=dev)
        self.S_mean = torch.zeros(self.CB, dtype=torch.float32, device=dev)
        self.S_var = torch.ones(self.CB, dtype=torch.float32, device=dev)
        self.W_train = torch.zeros(self.n_kernel, dtype=self.train_dtype, device=dev)
        self.Wt_train = torch.zeros(self.n_kernel, dtype=self.train_dtype, device=dev)
        self.W_infer = torch.zeros(self.n_kernel, dtype=self.infer_dtype, device=dev)
        self.fold_bias = torch.zeros(self.CB, dtype=torch.float32, device=dev)
        tab = np.zeros((len(self.layers), 8), dtype=np.int32)
        for i, L in enumerate(self.layers.values()):
            tab[i] = [L.w_off, L.cout, L.ktot, L.kh * L.kw, L.cin, L.bn_off if L.has_bn else -1,
                      L.bn_off if L.has_bn else -1, 0]
        self.table = torch.from_numpy(tab).to(dev)
        self.G = None  # gradient / optimiser state are created by the Trainer
        self.load_keras_params(self.init_keras_params(seed))

    # ---- Keras-layout import / export (HWIO kernels, [in,out] dense; apps/train_softmax.py:68-78) ----
    def init_keras_params(self, seed: int = 0) -> "OrderedDict[str, torch.Tensor]":
        """Glorot-uniform kernels (inception_resnet_v1.py:66), zero biases / beta, moving stats (0, 1),
        drawn in declaration order from torch.Generator(seed) on the CPU."""
        gen = torch.Generator().manual_seed(seed)
        out: "OrderedDict[str, torch.Tensor]" = OrderedDict()
        for L in self.layers.values():
            if L.dense:
                w = torch.empty(L.cin_real, L.cout_real)
                lim = math.sqrt(6.0 / (L.cin_real + L.cout_real))
            else:
                w = torch.empty(L.kh, L.kw, L.cin_real, L.cout)
                lim = math.sqrt(6.0 / (L.kh * L.kw * (L.cin_real + L.cout)))
            w.uniform_(-lim, lim, generator=gen)
            out[L.name + "/kernel"] = w
            if L.has_bias:
                out[L.name + "/bias"] = torch.zeros(L.cout_real)
            if L.has_bn:
                pre = self._bn_prefix(L)
                out[pre + "/beta"] = torch.zeros(L.cout)
                out[pre + "/moving_mean"] = torch.zeros(L.cout)
                out[pre + "/moving_variance"] = torch.ones(L.cout)
        return out

    @staticmethod
    def _bn_prefix(L: Layer) -> str:
        return "features/bn" if L.name == "features/logits" else L.name + "/bn"

    def _flat(self, params: Dict[str, torch.Tensor], with_stats: bool):
        """Engine-keyed Keras-layout tensors -> the flat fp32 layout [kernels OHWI | pad | betas | biases] (+ moving statistics)."""
        P = torch.zeros(self.n_params, dtype=torch.float32)
        mean = torch.zeros(self.CB)
        var = torch.ones(self.CB)
        for L in self.layers.values():
            w = torch.as_tensor(params[L.name + "/kernel"]).to(torch.float32)
            if L.dense:
                w = w.t().reshape(L.cout_real, 1, 1, L.cin_real)
                if L.cout != L.cout_real:
                    w = torch.cat([w, torch.zeros(L.cout - L.cout_real, 1, 1, L.cin_real)], 0)
            else:
                w = w.permute(3, 0, 1, 2)                      # HWIO -> O,H,W,I
            if L.cin != L.cin_real:
                w = torch.nn.functional.pad(w, (0, L.cin - L.cin_real))
            P[L.w_off:L.w_off + L.numel] = w.reshape(-1)
            if L.has_bias:
                P[L.bias_off:L.bias_off + L.cout_real] = torch.as_tensor(params[L.name + "/bias"]).to(torch.float32)
            if L.has_bn:
                pre = self._bn_prefix(L)
                P[self.beta_base + L.bn_off:self.beta_base + L.bn_off + L.cout] = torch.as_tensor(params[pre + "/beta"])
                if with_stats:
                    mean[L.bn_off:L.bn_off + L.cout] = torch.as_tensor(params[pre + "/moving_mean"])
                    var[L.bn_off:L.bn_off + L.cout] = torch.as_tensor(params[pre + "/moving_variance"])
        return P, mean, var

    def flat_from_keras(self, params: Dict[str, torch.Tensor]) -> torch.Tensor:
        """Per-variable tensors (engine keys, Keras layouts; no moving statistics) -> one flat buffer laid out like ``P``
        (gradients, Adam slots)."""
        return self._flat(params, with_stats=False)[0].to(self.device)

    def load_keras_params(self, params: Dict[str, torch.Tensor]):
        """``params``: Keras-layout tensors keyed by Keras variable names (keras_names.py: the names and order the reference's
        declaration produces) or by the engine's ``<layer>/kernel`` keys; matched by name."""
        P, mean, var = self._flat(self._engine_keys(params), with_stats=True)
        self.P.copy_(P)
        self.S_mean.copy_(mean)
        self.S_var.copy_(var)
        self.folded_valid = False
        self.refresh_packs()

    def _engine_keys(self, params: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        from . import keras_names
        return keras_names.from_keras(params, self.layers, int(self.cfg["block8_1"]["repeat"]))

    def keras_variables(self, moving_stats: Optional[Tuple[torch.Tensor, torch.Tensor]] = None) -> "OrderedDict[str, torch.Tensor]":
        """``model.weights`` of the reference model: Keras variable names, Keras layouts, Keras order (keras_names.py)."""
        from . import keras_names
        return keras_names.to_keras(self.export_keras_params(moving_stats), self.layers, int(self.cfg["block8_1"]["repeat"]))

    def export_folded_params(self) -> "OrderedDict[str, torch.Tensor]":
        """BN-folded inference weights as facenet/tfutils.py:229-258 (export_h5) writes them: per layer ``<name>/weights`` =
        kernel * 1/sqrt(moving_variance + eps) (HWIO) and ``<name>/biases`` = beta - moving_mean * scale, or the layer's own
        bias.  Read back from the device packs the inference kernels use (fn_fold_bn), so this is what actually runs."""
        self.refresh_folded(self.stream(), force=True)
        torch.cuda.synchronize(self.device)
        W = self.W_infer.float().cpu()
        fb = self.fold_bias.cpu()
        P = self.P.detach().cpu()
        out: "OrderedDict[str, torch.Tensor]" = OrderedDict()
        for L in self.layers.values():
            w = W[L.w_off:L.w_off + L.numel].reshape(L.cout, L.kh, L.kw, L.cin)[:L.cout_real, ..., :L.cin_real]
            out[L.name + "/weights"] = (w.reshape(L.cout_real, L.cin_real).t() if L.dense else w.permute(1, 2, 3, 0)).contiguous()
            if L.has_bn:
                out[L.name + "/biases"] = fb[L.bn_off:L.bn_off + L.cout].clone()
            elif L.has_bias:
                out[L.name + "/biases"] = P[L.bias_off:L.bias_off + L.cout_real].clone()
        return out

    def export_keras_params(self, moving_stats: Optional[Tuple[torch.Tensor, torch.Tensor]] = None) -> "OrderedDict[str, torch.Tensor]":
        """Keras-layout tensors under the engine's keys.  ``moving_stats``: (mean, var) to export instead of this replica's
        (data parallelism: the cross-replica average, Trainer.averaged_moving_stats)."""
        P = self.P.detach().cpu()
        mean, var = (self.S_mean.cpu(), self.S_var.cpu()) if moving_stats is None else (moving_stats[0].cpu(), moving_stats[1].cpu())
        out: "OrderedDict[str, torch.Tensor]" = OrderedDict()
        for L in self.layers.values():
            w = P[L.w_off:L.w_off + L.numel].reshape(L.cout, L.kh, L.kw, L.cin)[:L.cout_real, ..., :L.cin_real]
            out[L.name + "/kernel"] = (w.reshape(L.cout_real, L.cin_real).t() if L.dense else w.permute(1, 2, 3, 0)).contiguous()
            if L.has_bias:
                out[L.name + "/bias"] = P[L.bias_off:L.bias_off + L.cout_real].clone()
            if L.has_bn:
                pre = self._bn_prefix(L)
                out[pre + "/beta"] = P[self.beta_base + L.bn_off:self.beta_base + L.bn_off + L.cout].clone()
                out[pre + "/moving_mean"] = mean[L.bn_off:L.bn_off + L.cout].clone()
                out[pre + "/moving_variance"] = var[L.bn_off:L.bn_off + L.cout].clone()
        return out

    def export_keras_grads(self, G: torch.Tensor) -> Dict[str, torch.Tensor]:
        Gc = G.detach().cpu()
        out = {}
        for L in self.layers.values():
            w = Gc[L.w_off:L.w_off + L.numel].reshape(L.cout, L.kh, L.kw, L.cin)[:L.cout_real, ..., :L.cin_real]
            out[L.name + "/kernel"] = (w.reshape(L.cout_real, L.cin_real).t() if L.dense else w.permute(1, 2, 3, 0)).contiguous()
            if L.has_bias:
                out[L.name + "/bias"] = Gc[L.bias_off:L.bias_off + L.cout_real].clone()
            if L.has_bn:
                out[self._bn_prefix(L) + "/beta"] = Gc[self.beta_base + L.bn_off:self.beta_base + L.bn_off + L.cout].clone()
        return out

    # ---- weight packs ---------------------------------------------------------------------------
    def stream(self) -> int:
        return torch.cuda.current_stream(self.device).cuda_stream

    def refresh_packs(self, stream: Optional[int] = None):
        """fp32 master -> training pack, transposed (dgrad) pack, BN-folded inference pack."""
        st = self.stream() if stream is None else stream
        self.W_train.copy_(self.P[:self.n_kernel])
        self.refresh_transposed(st)
        self.refresh_folded(st)

    def refresh_transposed(self, st: int):
        _lib.check(self.lib.fn_pack_transpose(_ptr(self.W_train), _ptr(self.Wt_train), _ptr(self.table), len(self.layers),
                                              self.max_layer_elems, _lib.dtype_code(self.train_dtype), st), "pack_transpose")

    def refresh_folded(self, st: int, force: bool = True):
        """BN-folded inference pack from the fp32 masters.  `force=False` skips the launch while nothing has changed the
        parameters or the moving statistics since the last fold (`folded_valid`; cleared by load_keras_params, by every
        trainer step and by training-mode forwards)."""
        if not force and getattr(self, "folded_valid", False):
            return
        self.folded_valid = True
        _lib.check(self.lib.fn_fold_bn(_ptr(self.P), _ptr(self.W_infer), _ptr(self.fold_bias), _ptr(self.P, self.beta_base),
                                       _ptr(self.S_mean), _ptr(self.S_var), _ptr(self.table), len(self.layers),
                                       self.max_layer_elems, BN_EPS, _lib.dtype_code(self.infer_dtype), st), "fold_bn")

    def alloc_grads(self) -> torch.Tensor:
        """The flat fp32 gradient buffer G (laid out like P) and Gacc, the fixed-point (fn_acc_t) accumulators of the bias
        gradients: bias gradients are sums over many workgroups, added as integers (order-independent) and converted into
        G[bias_lo:] by the plan's `grad_finalize` launch.  Both are zeroed before every backward pass."""
        if self.G is None:
            self.G = torch.zeros(self.n_params, dtype=torch.float32, device=self.device)
            self.Gacc = torch.zeros(max(1, self.n_params - self.bias_lo), dtype=torch.int64, device=self.device)
        return self.G

    def count_variables(self) -> Tuple[int, int]:
        """(total, trainable) counted on the UN-padded Keras shapes (SURVEY.md shape table)."""
        tot = tr = 0
        for L in self.layers.values():
            k = L.cout_real * L.kh * L.kw * L.cin_real
            tr += k + (L.cout_real if L.has_bias else 0) + (L.cout if L.has_bn else 0)
            tot += k + (L.cout_real if L.has_bias else 0) + (3 * L.cout if L.has_bn else 0)
        return tot, tr

    def plan(self, N: int, training: bool, loss: Optional[str] = None) -> "Lowering":
        g = Lowering(self, N=N, training=training, declare=False, loss=loss)
        g.embedding = self._topology(g)
        g.finish()
        return g


class BlockNetwork(Network):
    """ONE residual or reduction block (Block35 / Block17 / Block8 / ReductionA / ReductionB, inception_resnet_v1.py:83-377) on
    a [N,H,W,C] low-precision feature map, lowered by the same engine code as the full network.  Used by the per-block
    forward + backward parity tests ("block" level): ``plan.bufs['trunk']`` is the input (``.act`` in,
    ``.grad`` out), ``plan.embedding`` the block output (``.act`` out, ``.grad`` in)."""

    def __init__(self, kind: str, H: int, W: int, C: int, scale: float = 0.17, relu: bool = True, filters=None, repeat: int = 1, **kw):
        if kind not in BLOCK_TOWERS and kind not in ("reduction_a", "reduction_b"):
            raise ValueError(f"unknown block kind {kind!r}")
        self._blk = (kind, H, W, C, float(scale), bool(relu), filters, int(repeat))
        super().__init__(embedding_size=8, **kw)

    def _topology(self, g: "Lowering"):
        kind, H, W, C, scale, relu, filters, repeat = self._blk
        x = g.feature_input(H, W, C)
        if kind in BLOCK_TOWERS:                 # `repeat` chained blocks "<kind>/<i>", like the repeated stages of the model
            towers, up = BLOCK_TOWERS[kind]
            if up != C:
                raise ValueError(f"{kind} expects {up} input channels, got {C}")
            for i in range(repeat):
                x = g.block(f"{kind}/{i}", x, towers, up, scale, relu)
            return x
        return g.reduction(kind, x, reduction_towers(kind, filters or DEFAULT_CONFIG[kind]["filters"]))

    def _bn_prefix(self, L: Layer) -> str:
        return L.name + "/bn"

    def _engine_keys(self, params):
        return params          # a lone block has no place in the reference model's variable naming: engine keys only


# ------------------------------------------------------------------------------------------------
# lowering: topology -> buffers + forward records -> launch lists
# ------------------------------------------------------------------------------------------------
class Lowering:
    def __init__(self, net: Network, N: int, training: bool, declare: bool, loss: Optional[str] = None):
        self.net, self.N, self.training, self.declare, self.loss = net, N, training, declare, loss
        self.fuse_bn_bwd = True     # BN-backward reduction inside the producing dgrad's epilogue where it is the sole producer
        # Optional: BN+ReLU outputs that only convolutions read are never written; the consumers (forward and weight
        # gradient) normalise the raw tensor while staging their operand tile (fn_conv_desc.nrm_*) and fn_bn_finalize
        # publishes scale / shift for the backward pass and the moving statistics in one launch.  Saves 78 of 82 bn_relu_fwd
        # launches and the activated copies, but every tap and every N tile repeats the per-element affine+ReLU, which makes
        # the staging VALU-bound: measured 9.52 ms vs 9.11 ms per step on MI355X -> off by default.
        self.norm_on_load = bool(int(os.environ.get("FACENET_NORM_ON_LOAD", "0")))
        # Optional (FACENET_LAZY_BN_MAXHW=17; default 0 = off): on maps up to that size a BN+ReLU output with exactly ONE reader, a
        # stride-1 convolution whose output map has the input's size, is materialised BY that reader -- it normalises the raw
        # tensor while staging its operand and the workgroups of its first column tile write the activated tensor once, at the
        # centre tap (fn_conv_desc.nrm_z); the weight gradient and the backward pass read the same buffers as before.  68 of the
        # 82 fn_bn_relu_train_fwd launches disappear, but the step does not get faster (MI355X, batch 90: 7.58-7.60 ms with,
        # 7.54 ms without): normalise-on-load costs each reader 2.5-4.5 us (tools/dev_normcost.py: a dependent statistics round
        # trip in the prologue plus ~40 VALU instructions per 16-byte chunk on the load -> LDS path of a latency-bound k loop,
        # repeated per tap and per column tile), which is what the removed launch and its boundary cost.  Kept as an option.
        self.lazy_bn_maxhw = int(os.environ.get("FACENET_LAZY_BN_MAXHW", "0"))
        self.lazy_bn_kmax = int(os.environ.get("FACENET_LAZY_BN_KMAX", "1024"))      # k x k readers only up to this many K columns
        self.lazy: Dict[str, List[Tuple[int, int]]] = {}        # buffer -> [(c0, C)] BN ranges materialised by their reader
        self.merge_sibling_dgrads = bool(int(os.environ.get("FACENET_MERGE_SIBLINGS", "1")))
        # The residual backward of block i (ReLU mask, scaled copy for its `up` branch, bias gradient, pass-through to its
        # trunk) runs in the epilogue of the launch that completes the gradient of block i's output: the merged sibling data
        # gradient of block i+1 (fn_conv_desc.rb_*).  18 of the 21 fn_residual_bwd launches and one read-modify-write pass over
        # every block output's gradient disappear.
        self.fuse_residual_bwd = bool(int(os.environ.get("FACENET_FUSE_RESIDUAL_BWD", "1")))
        # Inference / mining plans: a whole Block17 (five convolution launches) runs as ONE launch with its tower activations in
        # LDS (fn_block17_infer, csrc/block_fused.hip).  BatchNorm is folded there, so nothing couples the images of a batch.
        # One workgroup per image: below ~32 images the fused kernels leave most of the chip idle and the layer-wise launches win
        # (tools/bench_inference.py, batch 1 / 8 / 32: 0.91 / 0.97 / 1.09 ms fused against 0.80 / 0.86 / 1.06 ms layer-wise);
        # FACENET_FUSE_BLOCKS = 0 never, 1 from FACENET_FUSE_BLOCKS_MIN_BATCH (32) images up.
        self.fuse_blocks = bool(int(os.environ.get("FACENET_FUSE_BLOCKS", "1"))) and \
            (declare or N >= int(os.environ.get("FACENET_FUSE_BLOCKS_MIN_BATCH", "32")))
        self.virtual: Dict[str, List[Tuple[int, int]]] = {}     # buffer -> [(c0, C)] BN ranges that are not materialised
        self.dtype = net.train_dtype if training else net.infer_dtype
        self.dt = None if declare else _lib.dtype_code(self.dtype)
        self.bufs: "OrderedDict[str, Buf]" = OrderedDict()
        self.recs: List[Rec] = []
        self.fwd: List[Op] = []
        self.bwd: List[Op] = []
        self.bwd_marks: List[Tuple[int, int]] = []   # (index into bwd after which..., lowest finished w_off)
        self.readers: Dict[str, int] = {}            # forward consumers per buffer
        self.bn_ranges: Dict[str, List[Tuple[int, int, bool]]] = {}
        self.bn_reduced: Dict[Tuple[str, int, int], int] = {}   # BN slices whose backward reduction a dgrad epilogue performs
        self.embedding = None

    # ---- buffers -------------------------------------------------------------------------------
    def buf(self, name: str, H: int, W: int, Cc: int, bn_channels: int = 0, need_raw: bool = False, f32: bool = False) -> Buf:
        net = self.net
        if self.declare:
            bn_off = None
            if bn_channels:
                bn_off = net.CB
                net.buf_bn[name] = bn_off
                net.CB += bn_channels
            b = Buf(name, self.N, H, W, Cc, bn_off)
        else:
            b = Buf(name, self.N, H, W, Cc, net.buf_bn.get(name))
            dt = torch.float32 if f32 else self.dtype
            b.f32 = f32
            b.act = torch.zeros(self.N, H, W, Cc, dtype=dt, device=net.device)
            if self.training:
                if need_raw:
                    b.raw = torch.zeros(self.N, H, W, Cc, dtype=dt, device=net.device)
                b.grad = torch.zeros(self.N, H, W, Cc, dtype=dt, device=net.device)
        self.bufs[name] = b
        return b

    def input(self, H: int, W: int) -> Slice:
        self.images = None if self.declare else torch.zeros(self.N, H, W, 3, dtype=torch.uint8, device=self.net.device)
        self.norm_work = None if self.declare else torch.zeros(8 * self.N, dtype=torch.float32, device=self.net.device)
        b = self.buf("input", H, W, 8)
        return b.full()

    def feature_input(self, H: int, W: int, Cc: int) -> Slice:
        """A low-precision NHWC feature map as the plan's input (BlockNetwork): no image normalisation, and -- unlike the image
        input -- it receives a data gradient."""
        self.images = None
        b = self.buf("trunk", H, W, Cc)
        return b.full()

    @staticmethod
    def _geom(H, W, k, stride, padding):
        kh, kw = k
        ph, pw = ((kh // 2, kw // 2) if padding == "same" else (0, 0))
        if padding == "same":
            assert stride == 1   # hazard 1: SAME only with stride 1 in v1
        return kh, kw, ph, pw, (H + 2 * ph - kh) // stride + 1, (W + 2 * pw - kw) // stride + 1

    def conv(self, name: str, x: Slice, cout: int, k, stride: int, padding: str, out: Optional[Slice] = None, has_bn=True,
             has_bias=False, cin_real: Optional[int] = None, kind: str = "bn", **extra) -> Slice:
        kh, kw, ph, pw, OH, OW = self._geom(x.buf.H, x.buf.W, k, stride, padding)
        L = self.net._declare_layer(name, x.C, cin_real or x.C, cout, kh, kw, stride, ph, pw, has_bn, has_bias) \
            if self.declare else self.net.layers[name]
        if out is None:
            ob = self.buf(name, OH, OW, cout, bn_channels=cout if has_bn else 0, need_raw=has_bn)
            out = ob.full()
        if self.declare and has_bn:
            L.bn_off = out.buf.bn_off + out.c0
        assert out.buf.H == OH and out.buf.W == OW and out.C == cout, name
        self.readers[x.buf.name] = self.readers.get(x.buf.name, 0) + 1
        if "trunk" in extra:
            t = extra["trunk"]
            self.readers[t.buf.name] = self.readers.get(t.buf.name, 0) + 1
        self.recs.append(Rec("conv", L, x, out, dict(kind=kind, **extra)))
        return out

    def bn_apply(self, b: Buf, c0: int, Cc: int, relu: bool = True):
        self.bn_ranges.setdefault(b.name, []).append((c0, Cc, relu))
        self.recs.append(Rec("bn", None, None, Slice(b, c0, Cc), dict(relu=relu)))

    def cbr(self, name, x, cout, k, stride, padding, cin_real=None) -> Slice:
        y = self.conv(name, x, cout, k, stride, padding, cin_real=cin_real)
        self.bn_apply(y.buf, 0, cout)
        return y

    def maxpool(self, name: str, x: Slice, out: Optional[Slice] = None) -> Slice:
        OH, OW = (x.buf.H - 3) // 2 + 1, (x.buf.W - 3) // 2 + 1
        if out is None:
            out = self.buf(name, OH, OW, x.C).full()
        self.readers[x.buf.name] = self.readers.get(x.buf.name, 0) + 1
        self.recs.append(Rec("maxpool", None, x, out))
        return out

    def _tower(self, prefix: str, x: Slice, tower, last_out: Slice) -> None:
        for j, spec in enumerate(tower):
            nm, cout, k = spec[0], spec[1], spec[2]
            stride = spec[3] if len(spec) > 3 else 1
            padding = spec[4] if len(spec) > 4 else "same"
            name = f"{prefix}/{nm}"
            if j == len(tower) - 1:
                self.conv(name, x, cout, k, stride, padding, out=last_out)
            else:
                x = self.cbr(name, x, cout, k, stride, padding)

    def block(self, prefix: str, trunk: Slice, towers, up: int, scale: float, relu: bool) -> Slice:
        """Block35/17/8 (:83-259): towers -> concat -> up 1x1 (+bias) -> act(trunk + scale*up)."""
        H, W = trunk.buf.H, trunk.buf.W
        cm = sum(t[-1][1] for t in towers)
        if (not self.declare and not self.training and self.fuse_blocks and (H, W, up) == (8, 8, 896) and towers == BLOCK_TOWERS["block17"][0]
                and trunk.c0 == 0 and trunk.C == trunk.buf.C == 896):
            out = self.buf(prefix + "/out", H, W, up)
            self.readers[trunk.buf.name] = self.readers.get(trunk.buf.name, 0) + 1
            self.recs.append(Rec("block17", None, trunk, out.full(), dict(prefix=prefix, scale=float(scale), relu=bool(relu))))
            return out.full()
        if (not self.declare and not self.training and self.fuse_blocks and (H, W, up) == (17, 17, 256) and towers == BLOCK_TOWERS["block35"][0]
                and trunk.c0 == 0 and trunk.C == trunk.buf.C == 256):
            out = self.buf(prefix + "/out", H, W, up)
            self.readers[trunk.buf.name] = self.readers.get(trunk.buf.name, 0) + 1
            self.recs.append(Rec("block35", None, trunk, out.full(), dict(prefix=prefix, scale=float(scale), relu=bool(relu))))
            return out.full()
        mixed = self.buf(prefix + "/mixed", H, W, cm, bn_channels=cm, need_raw=True)
        c0 = 0
        for i, t in enumerate(towers):
            self._tower(f"{prefix}/tower_conv{i}", trunk, t, mixed.sl(c0, t[-1][1]))
            c0 += t[-1][1]
        self.bn_apply(mixed, 0, cm)                      # one pass over the whole concat buffer
        out = self.buf(prefix + "/out", H, W, up)
        self.conv(prefix + "/up", mixed.full(), up, (1, 1), 1, "same", out=out.full(), has_bn=False, has_bias=True,
                  kind="resid", trunk=trunk, scale=float(scale), relu=bool(relu))
        return out.full()

    def reduction(self, prefix: str, trunk: Slice, towers) -> Slice:
        """ReductionA/B (:262-377): strided towers + MaxPool, concatenated."""
        H, W = trunk.buf.H, trunk.buf.W
        OH, OW = (H - 3) // 2 + 1, (W - 3) // 2 + 1
        cbn = sum(t[-1][1] for t in towers)
        out = self.buf(prefix + "/out", OH, OW, cbn + trunk.C, bn_channels=cbn, need_raw=True)
        c0 = 0
        for i, t in enumerate(towers):
            self._tower(f"{prefix}/tower_conv{i}", trunk, t, out.sl(c0, t[-1][1]))
            c0 += t[-1][1]
        self.bn_apply(out, 0, cbn)
        self.maxpool(prefix + "/MaxPool_1a_3x3", trunk, out=out.sl(cbn, trunk.C))
        return out.full()

    def head(self, x: Slice, E: int) -> Slice:
        """features (:459-468): AvgPool2D([3,3]) valid (stride = pool) -> Flatten -> Dense(no bias) -> BN."""
        H, W = x.buf.H, x.buf.W
        if (H, W) != (3, 3):
            # AvgPool2D([3,3], 'valid') pools the top-left 3x3 window only and Flatten of a larger pooled map has a layout of
            # its own (hazard 11): the whole-map average of fn_avgpool_* is the reference's result for 3x3 maps exactly
            raise ValueError(f"head expects a 3x3 final map (image sizes 139..170; the reference uses 160), got {H}x{W}")
        pooled = self.buf("features/avgpool", 1, 1, x.C)
        self.readers[x.buf.name] = self.readers.get(x.buf.name, 0) + 1
        self.recs.append(Rec("avgpool", None, x, pooled.full()))
        yh = self.buf("features/logits", 1, 1, E, bn_channels=E, f32=True)
        L = self.net._declare_layer("features/logits", x.C, x.C, E, 1, 1, 1, 0, 0, True, False, dense=True) \
            if self.declare else self.net.layers["features/logits"]
        if self.declare:
            L.bn_off = yh.bn_off
        self.recs.append(Rec("conv", L, pooled.full(), yh.full(), dict(kind="f32")))
        emb = self.buf("features/bn", 1, 1, E, f32=True)
        self.recs.append(Rec("head_bn", L, yh.full(), emb.full()))
        return emb.full()

    # ---- emission ------------------------------------------------------------------------------
    # Every launch declares the regions it reads / writes (schedule.Region) so that schedule.Schedule can overlap
    # independent launches.  Activation regions are channel intervals of an NHWC buffer.
    @staticmethod
    def _ra(s: Slice) -> Region:
        return (s.buf.act.data_ptr(), s.c0, s.c0 + s.C)

    @staticmethod
    def _rr(s: Slice) -> Region:
        return (s.buf.raw.data_ptr(), s.c0, s.c0 + s.C)

    @staticmethod
    def _rg(s: Slice) -> Region:
        return (s.buf.grad.data_ptr(), s.c0, s.c0 + s.C)

    def _desc(self, L: Layer, x: Slice, y: Slice) -> ConvDesc:
        d = ConvDesc()
        d.N, d.H, d.W, d.Cin = self.N, x.buf.H, x.buf.W, L.cin
        d.OH, d.OW, d.Cout = y.buf.H, y.buf.W, L.cout
        d.KH, d.KW, d.stride, d.pad_h, d.pad_w = L.kh, L.kw, L.stride, L.pad_h, L.pad_w
        d.dtype = self.dt
        d.ld_x, d.ld_y = x.buf.C, y.buf.C
        d.scale = 1.0
        return d

    @staticmethod
    def _replicas(M: int) -> int:
        """Accumulator replicas for the conv-epilogue BN statistics: thousands of row tiles adding into one address
        serialise at the memory side (MI355X_MICROARCH.md 'Global float atomics')."""
        r = 1
        while r < STAT_REPLICAS and M // (64 * r) > 32:   # ~<= 32 row tiles add into one replica
            r *= 2
        return r

    def _emit(self, lst: List[Op], name: str, fn, *args, keep=(), r=(), w=()):
        lst.append(Op(name, fn, args, tuple(keep), tuple(r), tuple(w)))

    def _grad_mode(self, s: Slice) -> int:
        """0 = first writer of this channel range (overwrite), 1 = accumulate."""
        rng = (s.c0, s.c0 + s.C)
        for (a, b) in s.buf.grad_ranges:
            if a <= rng[0] and rng[1] <= b:
                return 1
        s.buf.grad_ranges.append(rng)
        return 0

    def finish(self):
        net, lib = self.net, self.net.lib
        dev = net.device
        CB = net.CB
        N = self.N
        if self.training:
            # BN workspaces: STAT_REPLICAS x (sum | sumsq) accumulator replicas of fixed-point fn_acc_t (int64): integer atomics,
            # so the totals -- and with them every training step -- have the same bits whatever order the workgroups arrive in
            self.ws = torch.zeros(2 * STAT_REPLICAS * CB, dtype=torch.int64, device=dev)
            self.ws_b = torch.zeros(2 * STAT_REPLICAS * CB, dtype=torch.int64, device=dev)   # BN-backward sums (replicated)
            self.save_scale = torch.zeros(CB, dtype=torch.float32, device=dev)
            self.save_shift = torch.zeros(CB, dtype=torch.float32, device=dev)
            self.head_mean = torch.zeros(net.E, dtype=torch.float32, device=dev)
            self.head_rstd = torch.zeros(net.E, dtype=torch.float32, device=dev)
            self._find_virtual()
            self._find_lazy()
            self.fin_reps = torch.zeros(CB, dtype=torch.int32)
            self.fin_count = torch.ones(CB, dtype=torch.int32)
        inp = self.bufs.get("input")
        if inp is not None:
            self._emit(self.fwd, "image_normalize", lib.fn_image_normalize, _ptr(self.images), _ptr(inp.act), _ptr(self.norm_work),
                       N, self.images.shape[1] * self.images.shape[2], net.normalization, self.dt,
                       r=[region(self.images)], w=[self._ra(inp.full()), region(self.norm_work)])
        for r in self.recs:
            getattr(self, "_fwd_" + r.kind)(r)
        if self.training and (self.virtual or self.lazy):
            self.fin_reps, self.fin_count = self.fin_reps.to(dev), self.fin_count.to(dev)
            self._emit(self.fwd, "bn_finalize", lib.fn_bn_finalize, _ptr(self.ws), CB, 2 * CB, _ptr(self.fin_reps), _ptr(self.fin_count),
                       _ptr(net.P, net.beta_base), _ptr(self.save_scale), _ptr(self.save_shift), _ptr(net.S_mean), _ptr(net.S_var),
                       BN_MOMENTUM, BN_EPS, CB,
                       r=[(self.ws.data_ptr() + 1, 0, CB), region(net.P, net.beta_base, net.beta_base + CB)],
                       w=[region(self.save_scale), region(self.save_shift), region(net.S_mean), region(net.S_var)])

    def _find_virtual(self):
        """A BN(+ReLU) range is virtual when every reader is the x operand of a convolution that can normalise on load."""
        if not self.norm_on_load:
            return
        for rec in self.recs:
            if rec.kind != "bn" or not rec.extra["relu"]:
                continue
            b, c0, Cc = rec.y.buf, rec.y.c0, rec.y.C
            ok = b.raw is not None
            for r2 in self.recs:
                if r2.kind == "conv":
                    t = r2.extra.get("trunk")
                    if t is not None and t.buf is b and t.c0 < c0 + Cc and c0 < t.c0 + t.C:
                        ok = False                      # residual operand of an `up` convolution
                    if r2.x.buf is b and r2.x.c0 < c0 + Cc and c0 < r2.x.c0 + r2.x.C:
                        inside = c0 <= r2.x.c0 and r2.x.c0 + r2.x.C <= c0 + Cc
                        ok = ok and inside and r2.layer.cin <= 512
                elif r2.kind != "bn" and r2.x is not None and r2.x.buf is b and r2.x.c0 < c0 + Cc and c0 < r2.x.c0 + r2.x.C:
                    ok = False                          # pools and the head read the activated tensor
            if ok:
                self.virtual.setdefault(b.name, []).append((c0, Cc))

    def _find_lazy(self):
        """BN(+ReLU) ranges that their single reader materialises (see __init__)."""
        if self.lazy_bn_maxhw <= 0:
            return
        for rec in self.recs:
            if rec.kind != "bn" or not rec.extra["relu"]:
                continue
            b, c0, Cc = rec.y.buf, rec.y.c0, rec.y.C
            if b.raw is None or max(b.H, b.W) > self.lazy_bn_maxhw or self._is_virtual(rec.y):
                continue
            readers = []
            for r2 in self.recs:
                if r2.kind == "bn":
                    continue
                t = r2.extra.get("trunk") if r2.kind == "conv" else None
                touches = (r2.x is not None and r2.x.buf is b and r2.x.c0 < c0 + Cc and c0 < r2.x.c0 + r2.x.C) or \
                          (t is not None and t.buf is b and t.c0 < c0 + Cc and c0 < t.c0 + t.C)
                if touches:
                    readers.append(r2)
            if len(readers) != 1:
                continue
            r2 = readers[0]
            L = r2.layer
            if (r2.kind == "conv" and r2.x.buf is b and r2.x.c0 == c0 and r2.x.C == Cc and L.stride == 1 and L.cin <= 512
                    and r2.y.buf.H == b.H and r2.y.buf.W == b.W and (L.kh * L.kw == 1 or L.ktot <= self.lazy_bn_kmax)):
                self.lazy.setdefault(b.name, []).append((c0, Cc))

    def _is_lazy(self, s: Slice) -> bool:
        return any(c0 == s.c0 and Cc == s.C for (c0, Cc) in self.lazy.get(s.buf.name, []))

    def _is_virtual(self, s: Slice) -> bool:
        return any(c0 <= s.c0 and s.c0 + s.C <= c0 + Cc for (c0, Cc) in self.virtual.get(s.buf.name, []))

    def _norm_operand(self, d: ConvDesc, x: Slice, reads: list):
        """x is a virtual BN output: point the descriptor at the raw tensor and describe its statistics."""
        net = self.net
        o = x.buf.bn_off + x.c0
        d.x = _ptr(x.buf.raw, x.c0)
        d.nrm_stats = _ptr(self.ws, o)
        d.nrm_beta = _ptr(net.P, net.beta_base + o)
        d.nrm_sq_off, d.nrm_replicas, d.nrm_rep_stride = net.CB, self._replicas(x.buf.M), 2 * net.CB
        d.nrm_count, d.nrm_eps = x.buf.M, BN_EPS
        reads += [self._rr(x), (self.ws.data_ptr() + 1, o, o + x.C), region(net.P, net.beta_base + o, net.beta_base + o + x.C)]

    # forward emitters
    def _fwd_conv(self, r: Rec):
        net, lib, L = self.net, self.net.lib, r.layer
        kind = r.extra["kind"]
        d = self._desc(L, r.x, r.y)
        if self.training and self._is_virtual(r.x):
            reads, writes = [], []
            self._norm_operand(d, r.x, reads)
        elif self.training and self._is_lazy(r.x):
            reads, writes = [], [self._ra(r.x)]
            self._norm_operand(d, r.x, reads)
            d.nrm_z = _ptr(r.x.buf.act, r.x.c0)        # this launch writes the activated tensor it normalises
        else:
            d.x = _ptr(r.x.buf.act, r.x.c0)
            reads, writes = [self._ra(r.x)], []
        if self.training:
            d.w = _ptr(net.W_train, L.w_off)
            reads.append(region(net.W_train, L.w_off, L.w_off + L.numel))
            if kind == "bn":
                d.y = _ptr(r.y.buf.raw, r.y.c0)
                d.stats = _ptr(self.ws, L.bn_off)
                d.stats_sq_off = net.CB
                d.stats_replicas = self._replicas(r.y.buf.M)
                d.stats_rep_stride = 2 * net.CB
                writes += [self._rr(r.y), (self.ws.data_ptr() + 1, L.bn_off, L.bn_off + L.cout)]
        else:
            d.w = _ptr(net.W_infer, L.w_off)
            reads.append(region(net.W_infer, L.w_off, L.w_off + L.numel))
            if kind == "bn":
                d.y = _ptr(r.y.buf.act, r.y.c0)
                d.bias = _ptr(net.fold_bias, L.bn_off)
                d.relu = 1
                reads.append(region(net.fold_bias, L.bn_off, L.bn_off + L.cout))
                writes.append(self._ra(r.y))
        if kind == "resid":
            t: Slice = r.extra["trunk"]
            d.y = _ptr(r.y.buf.act, r.y.c0)
            d.bias = _ptr(net.P, L.bias_off)
            d.resid = _ptr(t.buf.act, t.c0)
            d.ld_res = t.buf.C
            d.scale = r.extra["scale"]
            d.relu = 1 if r.extra["relu"] else 0
            reads += [self._ra(t), region(net.P, L.bias_off, L.bias_off + L.cout)]
            writes.append(self._ra(r.y))
        elif kind == "f32":
            tgt = r.y.buf
            if L.has_bn and not self.training:       # inference: BN folded, write the embedding buffer directly
                tgt = self.bufs["features/bn"]
                d.bias = _ptr(net.fold_bias, L.bn_off)
                reads.append(region(net.fold_bias, L.bn_off, L.bn_off + L.cout))
            d.y = _ptr(tgt.act, r.y.c0)
            d.out_f32 = 1
            if L.has_bias:
                d.bias = _ptr(net.P, L.bias_off)
            writes.append((tgt.act.data_ptr(), r.y.c0, r.y.c0 + r.y.C))
        self._emit(self.fwd, "conv_fwd:" + L.name, lib.fn_conv2d_fwd, C.byref(d), keep=(d,), r=reads, w=writes)

    def _fwd_block35(self, r: Rec):
        net, pre = self.net, r.extra["prefix"]
        L1 = [net.layers[f"{pre}/{n}"] for n in ("tower_conv0/Conv2d_1x1", "tower_conv1/Conv2d_0a_1x1", "tower_conv2/Conv2d_0a_1x1")]
        L3 = [net.layers[f"{pre}/{n}"] for n in ("tower_conv1/Conv2d_0b_3x3", "tower_conv2/Conv2d_0b_3x3", "tower_conv2/Conv2d_0c_3x3")]
        Lu = net.layers[f"{pre}/up"]
        arr = lambda ptrs: (C.c_void_p * 3)(*ptrs)
        w1, w3 = arr([_ptr(net.W_infer, L.w_off) for L in L1]), arr([_ptr(net.W_infer, L.w_off) for L in L3])
        b1, b3 = arr([_ptr(net.fold_bias, L.bn_off) for L in L1]), arr([_ptr(net.fold_bias, L.bn_off) for L in L3])
        reads = [self._ra(r.x)] + [region(net.W_infer, L.w_off, L.w_off + L.numel) for L in L1 + L3 + [Lu]] + \
                [region(net.fold_bias, L.bn_off, L.bn_off + L.cout) for L in L1 + L3] + [region(net.P, Lu.bias_off, Lu.bias_off + Lu.cout)]
        warm, warm_bytes = self._warm_next_block(pre, ("tower_conv0/Conv2d_1x1", "tower_conv1/Conv2d_0a_1x1", "tower_conv2/Conv2d_0a_1x1",
                                                       "tower_conv1/Conv2d_0b_3x3", "tower_conv2/Conv2d_0b_3x3", "tower_conv2/Conv2d_0c_3x3", "up"), reads)
        self._emit(self.fwd, "block35_fused:" + pre, net.lib.fn_block35_infer_warm, _ptr(r.x.buf.act), _ptr(r.y.buf.act), self.N, w1, w3,
                   _ptr(net.W_infer, Lu.w_off), b1, b3, _ptr(net.P, Lu.bias_off), r.extra["scale"], 1 if r.extra["relu"] else 0,
                   warm, warm_bytes, self.dt, keep=(w1, w3, b1, b3), r=reads, w=[self._ra(r.y)])

    def _warm_next_block(self, pre: str, names, reads: list):
        """Warm-ahead range of a fused block launch: the inference weight packs of the NEXT block of the same kind (the launch's spare
        workgroups read them into every XCD's L2: fn_block17_infer_warm).  (None, 0) for the last block or FACENET_WARM_AHEAD=0."""
        net = self.net
        head, _, idx = pre.rpartition("/")
        if not (idx.isdigit() and f"{head}/{int(idx) + 1}/up" in net.layers and int(os.environ.get("FACENET_WARM_AHEAD", "1"))):
            return None, 0
        nxt = [net.layers[f"{head}/{int(idx) + 1}/{n}"] for n in names]
        lo, hi = min(L.w_off for L in nxt), max(L.w_off + L.numel for L in nxt)
        if hi - lo > 2 * sum(L.numel for L in nxt):          # the packs do not sit (nearly) back to back in the inference buffer
            return None, 0
        esz = net.W_infer.element_size()
        lo -= lo % (16 // esz)
        nbytes = min(((hi - lo) * esz + 15) // 16 * 16, (net.W_infer.numel() - lo) * esz // 16 * 16)
        reads.append(region(net.W_infer, lo, hi))
        return _ptr(net.W_infer, lo), nbytes

    def _fwd_block17(self, r: Rec):
        net, pre = self.net, r.extra["prefix"]
        Ls = [net.layers[f"{pre}/{n}"] for n in ("tower_conv0/Conv2d_1x1", "tower_conv1/Conv2d_0a_1x1", "tower_conv1/Conv2d_0b_1x7",
                                                 "tower_conv1/Conv2d_0c_7x1", "up")]
        ws = [_ptr(net.W_infer, L.w_off) for L in Ls]
        bs = [_ptr(net.fold_bias, L.bn_off) for L in Ls[:4]] + [_ptr(net.P, Ls[4].bias_off)]
        reads = [self._ra(r.x)] + [region(net.W_infer, L.w_off, L.w_off + L.numel) for L in Ls] + \
                [region(net.fold_bias, L.bn_off, L.bn_off + L.cout) for L in Ls[:4]] + [region(net.P, Ls[4].bias_off, Ls[4].bias_off + Ls[4].cout)]
        # warm-ahead: the launch's spare workgroups (180 images on 256 CUs) read the NEXT block's weight packs into every XCD's L2
        # (fn_block17_infer_warm): from memory the next block's weight stream costs 50 us per launch, from a warm L2 ~40
        warm, warm_bytes = self._warm_next_block(pre, ("tower_conv0/Conv2d_1x1", "tower_conv1/Conv2d_0a_1x1", "tower_conv1/Conv2d_0b_1x7",
                                                       "tower_conv1/Conv2d_0c_7x1", "up"), reads)
        self._emit(self.fwd, "block17_fused:" + pre, net.lib.fn_block17_infer_warm, _ptr(r.x.buf.act), _ptr(r.y.buf.act), self.N, *ws, *bs,
                   r.extra["scale"], 1 if r.extra["relu"] else 0, warm, warm_bytes, self.dt, r=reads, w=[self._ra(r.y)])

    def _fwd_bn(self, r: Rec):
        if not self.training:
            return  # folded into the convolution epilogue (facenet/tfutils.py:244-250)
        net, lib = self.net, self.net.lib
        b, c0, Cc = r.y.buf, r.y.c0, r.y.C
        o = b.bn_off + c0
        if self._is_virtual(r.y) or self._is_lazy(r.y):
            self.fin_reps[o:o + Cc] = self._replicas(b.M)
            self.fin_count[o:o + Cc] = b.M
            return
        self._emit(self.fwd, "bn_relu_fwd:" + b.name, lib.fn_bn_relu_train_fwd, _ptr(b.raw, c0), b.C, _ptr(b.act, c0), b.C, b.M, Cc,
                   _ptr(self.ws, o), net.CB, self._replicas(b.M), 2 * net.CB, _ptr(net.P, net.beta_base + o), _ptr(self.save_scale, o),
                   _ptr(self.save_shift, o), _ptr(net.S_mean, o), _ptr(net.S_var, o), BN_MOMENTUM, BN_EPS, 1 if r.extra["relu"] else 0, self.dt,
                   r=[self._rr(r.y), (self.ws.data_ptr() + 1, o, o + Cc), region(net.P, net.beta_base + o, net.beta_base + o + Cc)],
                   w=[self._ra(r.y), region(self.save_scale, o, o + Cc), region(self.save_shift, o, o + Cc),
                      region(net.S_mean, o, o + Cc), region(net.S_var, o, o + Cc)])

    def _fwd_maxpool(self, r: Rec):
        x, y = r.x, r.y
        am = None
        if self.training and x.buf.name != "input":   # 1-byte argmax map for the backward (first maximum of every window)
            am = torch.zeros(self.N, y.buf.H, y.buf.W, x.C, dtype=torch.uint8, device=self.net.device)
            r.extra["argmax"] = am
        self._emit(self.fwd, "maxpool_fwd", self.net.lib.fn_maxpool3x3s2_fwd, _ptr(x.buf.act, x.c0), x.buf.C, _ptr(y.buf.act, y.c0), y.buf.C,
                   self.N, x.buf.H, x.buf.W, x.C, _ptr(am) if am is not None else None, self.dt,
                   r=[self._ra(x)], w=[self._ra(y)] + ([region(am)] if am is not None else []))

    def _fwd_avgpool(self, r: Rec):
        x, y = r.x, r.y
        self._emit(self.fwd, "avgpool_fwd", self.net.lib.fn_avgpool_fwd, _ptr(x.buf.act), _ptr(y.buf.act), self.N, x.buf.H * x.buf.W, x.C, self.dt,
                   r=[self._ra(x)], w=[self._ra(y)])

    def _fwd_head_bn(self, r: Rec):
        if not self.training:
            return  # folded into the Dense epilogue
        net, L = self.net, r.layer
        o = L.bn_off
        self._emit(self.fwd, "head_bn_fwd", net.lib.fn_head_bn_fwd, _ptr(r.x.buf.act), _ptr(r.y.buf.act), self.N, net.E,
                   _ptr(net.P, net.beta_base + o), _ptr(net.S_mean, o), _ptr(net.S_var, o), _ptr(self.head_mean), _ptr(self.head_rstd), 1,
                   BN_MOMENTUM, BN_EPS,
                   r=[self._ra(r.x), region(net.P, net.beta_base + o, net.beta_base + o + net.E)],
                   w=[self._ra(r.y), region(self.head_mean), region(self.head_rstd), region(net.S_mean, o, o + net.E),
                      region(net.S_var, o, o + net.E)])

    # ---- backward (training plans only); demb = fp32 gradient wrt the un-normalised embedding ----
    def build_backward(self, demb: torch.Tensor):
        assert self.training and not self.bwd
        for b in self.bufs.values():
            b.grad_ranges = []
        self._demb = demb
        self._dup: Dict[str, torch.Tensor] = {}
        # sibling 1x1 stride-1 layers reading the same trunk slice (the first convolutions of inception towers): their data
        # gradients all accumulate into that slice -- emitted as ONE multi-source launch when the last of them is reached
        self._siblings: Dict[Tuple[str, int, int], List[Rec]] = {}
        self._sib_pending: Dict[Tuple[str, int, int], list] = {}
        if self.merge_sibling_dgrads:
            for r in self.recs:
                L = r.layer
                if (r.kind == "conv" and r.extra.get("kind") == "bn" and L.kh == 1 and L.kw == 1 and L.stride == 1 and L.pad_h == 0
                        and L.pad_w == 0 and r.x.buf.name != "input"):
                    self._siblings.setdefault((r.x.buf.name, r.x.c0, r.x.C), []).append(r)
            self._siblings = {k: v for k, v in self._siblings.items() if 2 <= len(v) <= 3}
        self._resid_of: Dict[str, Rec] = {r.y.buf.name: r for r in self.recs if r.kind == "conv" and r.extra.get("kind") == "resid"}
        for r in self.recs:
            r.extra.pop("rb_fused", None)
        for r in reversed(self.recs):
            getattr(self, "_bwd_" + r.kind)(r)
        self.emit_grad_finalize(self.bwd)

    def emit_grad_finalize(self, lst: List[Op]):
        """Bias gradients leave their fixed-point accumulators (net.Gacc) for the fp32 gradient buffer: one launch at the end of
        backward, before the optimiser / the all-reduce of the bias bucket."""
        net = self.net
        n = net.n_params - net.bias_lo
        if n > 0:
            self._emit(lst, "grad_finalize", net.lib.fn_acc_to_float, _ptr(net.Gacc), _ptr(net.G, net.bias_lo), n, 40,
                       r=[region(net.Gacc)], w=[region(net.G, net.bias_lo, net.n_params)])

    def _mark(self, L: Layer):
        self.bwd_marks.append((len(self.bwd), L.index))

    def _bwd_head_bn(self, r: Rec):
        net, L = self.net, r.layer
        self.head_dy = torch.zeros(self.N, net.E, dtype=self.dtype, device=net.device)
        gb = net.beta_base + L.bn_off
        self._emit(self.bwd, "head_bn_bwd", net.lib.fn_head_bn_bwd, _ptr(self._demb), _ptr(r.x.buf.act), _ptr(self.head_mean),
                   _ptr(self.head_rstd), _ptr(net.G, gb), _ptr(self.head_dy), self.N, net.E, self.dt,
                   r=[region(self._demb), self._ra(r.x), region(self.head_mean), region(self.head_rstd)],
                   w=[region(self.head_dy), region(net.G, gb, gb + net.E)])

    def _bias_acc(self, L: Layer) -> Tuple[int, Region]:
        """(pointer, region) of the fixed-point accumulator that receives the bias gradient of layer L (net.Gacc mirrors the
        bias part of the flat gradient buffer; `grad_finalize` at the end of backward converts it into G)."""
        net = self.net
        o = L.bias_off - net.bias_lo
        return _ptr(net.Gacc, o), region(net.Gacc, o, o + L.cout)

    def _bwd_conv(self, r: Rec):
        net, lib, L = self.net, self.net.lib, r.layer
        kind = r.extra["kind"]
        x, y = r.x, r.y
        if kind == "resid":
            t: Slice = r.extra["trunk"]
            # one scratch per block: the weight gradient of this block may still be reading it while the next block's
            # residual backward runs on another stream
            if r.extra.get("rb_fused"):      # done by the epilogue of the launch that completed y's gradient (see _fuse_residual)
                dup = self._dup[L.name]
            else:
                dup = torch.zeros(y.buf.M, y.buf.C, dtype=self.dtype, device=net.device)
                self._dup[L.name] = dup
                acc = self._grad_mode(t)
                bptr, breg = self._bias_acc(L)
                self._emit(self.bwd, "residual_bwd:" + L.name, lib.fn_residual_bwd, _ptr(y.buf.grad), _ptr(y.buf.act), _ptr(t.buf.grad),
                           _ptr(dup), bptr, y.buf.M, y.buf.C, r.extra["scale"], 1 if r.extra["relu"] else 0, acc, self.dt,
                           r=[self._rg(y), self._ra(y)], w=[self._rg(t), region(dup), breg])
            dy_ptr, ld_dy, dy_reg = _ptr(dup), y.buf.C, region(dup)
        elif kind == "f32":
            dy_ptr, ld_dy, dy_reg = _ptr(self.head_dy), net.E, region(self.head_dy)
        else:
            dy_ptr, ld_dy, dy_reg = _ptr(y.buf.grad, y.c0), y.buf.C, self._rg(y)
        d = self._desc(L, x, y)
        d.ld_y = ld_dy
        wreads = [dy_reg]
        if self._is_virtual(x):
            self._norm_operand(d, x, wreads)
        else:
            d.x = _ptr(x.buf.act, x.c0)
            wreads.append(self._ra(x))
        d.y = dy_ptr
        d.dw = _ptr(net.G, L.w_off)
        self._emit(self.bwd, "conv_wgrad:" + L.name, lib.fn_conv2d_wgrad, C.byref(d), keep=(d,),
                   r=wreads, w=[region(net.G, L.w_off, L.w_off + L.numel)])
        sib_key = (x.buf.name, x.c0, x.C)
        if sib_key in self._siblings and any(r is m for m in self._siblings[sib_key]):
            pend = self._sib_pending.setdefault(sib_key, [])
            pend.append((L, dy_ptr, ld_dy, dy_reg))
            if len(pend) == len(self._siblings[sib_key]):
                (L0, p0, ld0, reg0), rest = pend[0], pend[1:]
                g = self._desc(L0, x, y)
                g.Cout, g.ld_y, g.y = L0.cout, ld0, p0
                g.w = _ptr(net.Wt_train, L0.w_off)
                g.dx = _ptr(x.buf.grad, x.c0)
                g.accumulate = self._grad_mode(x)
                rd = [reg0, region(net.Wt_train, L0.w_off, L0.w_off + L0.numel)]
                wr = [self._rg(x)]
                self._fuse_residual(g, x, rd, wr)
                for i, (Li, pi, ldi, regi) in enumerate(rest):
                    setattr(g, ("dy2", "dy3")[i], pi)
                    setattr(g, ("w2", "w3")[i], _ptr(net.Wt_train, Li.w_off))
                    setattr(g, ("Cout2", "Cout3")[i], Li.cout)
                    setattr(g, ("ld_y2", "ld_y3")[i], ldi)
                    rd += [regi, region(net.Wt_train, Li.w_off, Li.w_off + Li.numel)]
                self._emit(self.bwd, "conv_dgrad:" + "+".join(p[0].name for p in pend), lib.fn_conv2d_dgrad, C.byref(g), keep=(g,),
                           r=rd, w=wr)
        elif x.buf.name != "input":
            g = self._desc(L, x, y)
            g.ld_y = ld_dy
            g.y = dy_ptr
            g.w = _ptr(net.Wt_train, L.w_off)
            g.dx = _ptr(x.buf.grad, x.c0)
            g.accumulate = self._grad_mode(x)
            rd = [dy_reg, region(net.Wt_train, L.w_off, L.w_off + L.numel)]
            wr = [self._rg(x)]
            bnr = [r_ for r_ in self.bn_ranges.get(x.buf.name, []) if r_[0] == x.c0 and r_[1] == x.C]
            if self.fuse_bn_bwd and bnr and g.accumulate == 0 and self.readers.get(x.buf.name, 0) == 1 and x.buf.raw is not None:
                o = x.buf.bn_off + x.c0            # this dgrad is the only producer of d(BN output): reduce in its epilogue
                reps = self._replicas(x.buf.M)
                g.bn_y = _ptr(x.buf.raw, x.c0)
                g.ld_bn_y = x.buf.C
                g.bn_scale, g.bn_shift = _ptr(self.save_scale, o), _ptr(self.save_shift, o)
                g.bn_beta = _ptr(net.P, net.beta_base + o)
                g.bn_acc, g.bn_sq_off, g.bn_replicas, g.bn_rep_stride = _ptr(self.ws_b, o), net.CB, reps, 2 * net.CB
                g.bn_relu = 1 if bnr[0][2] else 0
                self.bn_reduced[(x.buf.name, x.c0, x.C)] = reps
                rd += [self._rr(x), region(self.save_scale, o, o + x.C), region(self.save_shift, o, o + x.C)]
                wr.append((self.ws_b.data_ptr() + 1, o, o + x.C))
            self._emit(self.bwd, "conv_dgrad:" + L.name, lib.fn_conv2d_dgrad, C.byref(g), keep=(g,), r=rd, w=wr)
        self._mark(L)

    def _fuse_residual(self, g: ConvDesc, x: Slice, rd: list, wr: list):
        """`g` is the launch that completes the gradient of x.  When x is the whole output of a residual block and `g` only adds
        to what the next block's residual backward already left there, the block's own residual backward moves into g's epilogue
        (fn_conv_desc.rb_*): nothing else ever reads the completed gradient of x, so it is not even written."""
        net = self.net
        prev = self._resid_of.get(x.buf.name)
        if not (self.fuse_residual_bwd and prev is not None and g.accumulate == 1 and x.c0 == 0 and x.C == x.buf.C):
            return
        Lp, t = prev.layer, prev.extra["trunk"]
        if not (t.c0 == 0 and t.C == t.buf.C == x.buf.C):
            return
        dup = torch.zeros(x.buf.M, x.buf.C, dtype=self.dtype, device=net.device)
        self._dup[Lp.name] = dup
        prev.extra["rb_fused"] = True
        g.accumulate = 0
        g.rb_prev = _ptr(x.buf.grad)
        g.rb_out = _ptr(x.buf.act) if prev.extra["relu"] else None
        g.rb_dtrunk = _ptr(t.buf.grad)
        g.rb_accumulate = self._grad_mode(t)
        g.rb_dup = _ptr(dup)
        g.rb_dbias, breg = self._bias_acc(Lp)
        g.rb_scale = float(prev.extra["scale"])
        rd += [self._rg(x), self._ra(x)]
        wr[:] = [self._rg(t), region(dup), breg]

    def _bwd_bn(self, r: Rec):
        net = self.net
        b, c0, Cc = r.y.buf, r.y.c0, r.y.C
        o = b.bn_off + c0
        gb = net.beta_base + o
        reps = self.bn_reduced.get((b.name, c0, Cc), 0)
        self._emit(self.bwd, "bn_relu_bwd:" + b.name, net.lib.fn_bn_relu_train_bwd, _ptr(b.grad, c0), b.C, _ptr(b.raw, c0), b.C, b.M, Cc,
                   _ptr(net.P, gb), _ptr(self.save_scale, o), _ptr(self.save_shift, o), _ptr(net.G, gb),
                   _ptr(self.ws_b, o), net.CB, max(1, reps), 2 * net.CB, 1 if reps else 0, 1 if r.extra["relu"] else 0, self.dt,
                   r=[self._rr(r.y), region(net.P, gb, gb + Cc), region(self.save_scale, o, o + Cc), region(self.save_shift, o, o + Cc)],
                   w=[self._rg(r.y), region(net.G, gb, gb + Cc), (self.ws_b.data_ptr() + 1, o, o + Cc)])

    def _bwd_maxpool(self, r: Rec):
        x, y = r.x, r.y
        if x.buf.name == "input":
            return
        acc = self._grad_mode(x)
        am = r.extra.get("argmax")
        self._emit(self.bwd, "maxpool_bwd", self.net.lib.fn_maxpool3x3s2_bwd, _ptr(x.buf.act, x.c0), x.buf.C, _ptr(y.buf.grad, y.c0), y.buf.C,
                   _ptr(x.buf.grad, x.c0), x.buf.C, self.N, x.buf.H, x.buf.W, x.C, _ptr(am) if am is not None else None, acc, self.dt,
                   r=[self._ra(x), self._rg(y)] + ([region(am)] if am is not None else []), w=[self._rg(x)])

    def _bwd_avgpool(self, r: Rec):
        x, y = r.x, r.y
        assert self._grad_mode(x) == 0
        self._emit(self.bwd, "avgpool_bwd", self.net.lib.fn_avgpool_bwd, _ptr(y.buf.grad), _ptr(x.buf.grad), self.N, x.buf.H * x.buf.W, x.C, self.dt,
                   r=[self._rg(y)], w=[self._rg(x)])

    # ---- execution -----------------------------------------------------------------------------
    @staticmethod
    def run_ops(ops: Sequence[Op], stream: int, lo: int = 0, hi: Optional[int] = None):
        """Single-stream, program-order replay (reference semantics for the scheduled replay)."""
        for op in ops[lo:hi]:
            if getattr(op.fn, "_torch_op", False):
                op.fn(*op.args)
                continue
            rc = op.fn(*op.args, stream)
            if rc:
                _lib.check(rc, op.name)

    def run_forward(self, stream: Optional[int] = None):
        self.run_ops(self.fwd, self.net.stream() if stream is None else stream)
