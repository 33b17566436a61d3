"""``InceptionResnetV1`` with the reference's call contract (facenet/models/inception_resnet_v1.py:380-501):
``InceptionResnetV1(input_shape, image_processing, config=None)``; ``model(inputs, training=False) -> float32 [N,E]``
(L2-normalised when ``training`` is False, :491-492); ``.config``, ``.image_processing``, ``.custom_layers``, ``.summary()``.
Every launch is a hand-written gfx950 kernel behind libfacenet_hip.so; plans are cached per batch size."""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np
import torch

from .. import _lib
from ..config import Config
from ..engine import DEFAULT_CONFIG, Lowering, Network, _ptr

default_config = DEFAULT_CONFIG


def check_input_config(cfg=None):
    return Config(default_config) if cfg is None else cfg


class InceptionResnetV1:
    def __init__(self, input_shape, image_processing, config=None, device: str = "cuda", seed: int = 0,
                 infer_dtype: torch.dtype = torch.float16, train_dtype: torch.dtype = torch.bfloat16, nrof_classes=None):
        self.config = check_input_config(config)
        self.image_processing = image_processing
        cfg = self.config.as_dict
        size = int(input_shape[0]) if input_shape is not None else image_processing.config.size
        norm = image_processing.config.normalization if image_processing is not None else 0
        self.network = Network(embedding_size=cfg["output"]["size"], config=cfg, image_size=size, normalization=int(norm),
                               nrof_classes=nrof_classes, device=device, train_dtype=train_dtype, infer_dtype=infer_dtype, seed=seed)
        self.custom_layers = ("image_processing", "conv2d", "block35", "reduction_a", "block17", "reduction_b", "block8",
                              "block8_2", "features")          # :470-480
        self._plans: Dict[tuple, Lowering] = {}
        self._f32_in: Dict[int, torch.Tensor] = {}
        self._graphs: Dict[int, tuple] = {}      # inference: batch size -> (HIP graph of the plan + l2_normalize, output buffer)

    @property
    def embedding_size(self) -> int:
        return self.network.E

    def _plan(self, n: int, training: bool) -> Lowering:
        key = (n, training)
        if key not in self._plans:
            plan = self.network.plan(n, training=training)
            if n >= 16:     # batches worth the ~0.3 s: time the tile variants of every convolution once (train.autotune_convs)
                from ..train import autotune_convs
                if not training:
                    self.network.refresh_folded(self.network.stream())
                autotune_convs(plan.fwd, self.network)
            self._plans[key] = plan
        return self._plans[key]

    MAX_PLAN_BATCH = 256
    use_graphs = True       # inference calls on uint8 images replay a captured HIP graph per batch size

    def __call__(self, inputs, training: bool = False, **kwargs) -> torch.Tensor:
        net = self.network
        x = torch.as_tensor(np.asarray(inputs)) if not torch.is_tensor(inputs) else inputs
        if x.dim() != 4 or x.shape[3] != 3:
            raise ValueError(f"expected NHWC images [N,{net.image_size},{net.image_size},3], got {tuple(x.shape)}")
        n = x.shape[0]
        if not training and n > self.MAX_PLAN_BATCH:
            # the kernels address an activation tensor with 32-bit byte offsets (< 1 GiB: 388 images at the 147x147x64 stem
            # layer); inference has no cross-image coupling (BatchNorm folded), so larger batches run in slices
            return torch.cat([self(x[i:i + self.MAX_PLAN_BATCH], training=False) for i in range(0, n, self.MAX_PLAN_BATCH)])
        if training and n > 380:
            raise ValueError(f"training batch {n} exceeds the 380 images one plan can address (per-GPU batch; use data parallelism)")
        if x.shape[1] != net.image_size or x.shape[2] != net.image_size:     # tf.image.resize, facenet.py:70
            src = x.to(net.device).contiguous()
            if src.dtype not in (torch.uint8, torch.float32):
                src = src.to(torch.float32)
            x = torch.empty(n, net.image_size, net.image_size, 3, dtype=torch.float32, device=net.device)
            _lib.check(net.lib.fn_image_resize_bilinear(_ptr(src), 1 if src.dtype == torch.float32 else 0, _ptr(x), n, src.shape[1], src.shape[2],
                                                        net.image_size, net.image_size, net.stream()), "image_resize")
        plan = self._plan(n, training)
        st = net.stream()
        if training:
            net.folded_valid = False       # the forward below updates the moving statistics
        else:
            net.refresh_folded(st, force=False)
        if x.dtype == torch.uint8 and not training and self.use_graphs:
            # serving path: the whole plan + l2_normalize replayed as ONE HIP graph (107 launches; at batch 1 the eager
            # Python dispatch costs more than the kernels)
            plan.images.copy_(x.to(net.device))
            if n not in self._graphs:
                out = torch.empty(n, net.E, dtype=torch.float32, device=net.device)
                emb = plan.embedding.buf.act.view(n, net.E)

                def run():
                    s_ = net.stream()
                    Lowering.run_ops(plan.fwd, s_)
                    _lib.check(net.lib.fn_l2norm_fwd(_ptr(emb), _ptr(out), n, net.E, 1e-10, s_), "l2norm")
                run()                                  # warm-up outside the capture
                torch.cuda.synchronize(net.device)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    run()
                self._graphs[n] = (g, out)
            g, out = self._graphs[n]
            g.replay()
            return out.clone()
        if x.dtype == torch.uint8:
            plan.images.copy_(x.to(net.device))
            Lowering.run_ops(plan.fwd, st)
        else:   # float images: facenet.py:69 casts to float32 anyway
            xf = x.to(device=net.device, dtype=torch.float32).contiguous()
            _lib.check(net.lib.fn_image_normalize_f32(_ptr(xf), _ptr(plan.bufs["input"].act), _ptr(plan.norm_work), n,
                                                      net.image_size * net.image_size, net.normalization, plan.dt, st), "image_normalize")
            Lowering.run_ops(plan.fwd[1:], st)
        emb = plan.embedding.buf.act.view(n, net.E)
        if training:
            return emb.clone()                 # un-normalised in training (:491)
        out = torch.empty_like(emb)
        _lib.check(net.lib.fn_l2norm_fwd(_ptr(emb), _ptr(out), n, net.E, 1e-10, st), "l2norm")   # :492
        return out

    def summary(self, line_length=None, positions=None, print_fn=None):
        pr = print_fn or print
        for L in self.network.layers.values():
            pr(f"{L.name:60s} {L.kh}x{L.kw}/s{L.stride} {L.cin_real:5d} -> {L.cout_real:5d}" + ("  +BN" if L.has_bn else "") + ("  +bias" if L.has_bias else ""))
        tot, tr = self.network.count_variables()
        pr(f"Total variables: {tot}")
        pr(f"Trainable variables: {tr}")
        pr(f"Non-trainable variables: {tot - tr}")

    # weights in Keras layout (HWIO / [in,out]); apps/train_softmax.py:68-78 load_weights / ModelCheckpoint
    def get_weights(self):
        return self.network.export_keras_params()

    def set_weights(self, params):
        self.network.load_keras_params(params)

    def save_weights(self, path):
        """``.npz`` keyed by the reference model's Keras variable names, in ``model.weights`` order (keras_names.py)."""
        np.savez(path, **{k: v.numpy() for k, v in self.network.keras_variables().items()})

    def load_weights(self, path):
        with np.load(path, allow_pickle=False) as z:
            self.network.load_keras_params({k: torch.from_numpy(z[k]) for k in z.files})
