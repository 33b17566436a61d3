"""Developer aid: where does a block's backward leave the rounded-autograd oracle?  Compares every intermediate gradient of one
residual block (raw conv outputs = post-BN-backward gradients, block input) between the engine and tests/quant_oracle.py.
    python tools/dev_blockgrad.py [block17|block35|block8] [f16|bf16] [N]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("FACENET_AUTOTUNE", "0")

import torch
import torch.nn.functional as F

from facenet_amd.engine import BLOCK_TOWERS, BlockNetwork
from oracle import facenet_oracle as fo
from tests.quant_oracle import QuantOracle, _q

kind = sys.argv[1] if len(sys.argv) > 1 else "block17"
dt = torch.float16 if (len(sys.argv) > 2 and sys.argv[2] == "f16") else torch.bfloat16
N = int(sys.argv[3]) if len(sys.argv) > 3 else 32
H, C, scale = {"block35": (17, 256, 0.17), "block17": (8, 896, 0.10), "block8": (3, 1792, 0.2)}[kind]


def rel(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))


class Tap(QuantOracle):
    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.raw = {}

    def _conv(self, x, prefix, spec, bias=False):
        y = super()._conv(x, prefix, spec, bias)
        y.retain_grad()
        self.raw[prefix] = y
        return y


net = BlockNetwork(kind, H, H, C, scale=scale, relu=True, device="cuda:0", train_dtype=dt, seed=11)
g = torch.Generator().manual_seed(5)
params = net.export_keras_params()
for k in params:
    if k.endswith("/beta") or k.endswith("/bias"):
        params[k] = 0.1 * torch.randn(params[k].shape, generator=g)
net.load_keras_params(params)
plan = net.plan(N, training=True)
net.G = torch.zeros(net.n_params, dtype=torch.float32, device=net.device)
x = torch.relu(torch.randn(N, H, H, C, generator=g) + 0.3).to(dt)
trunk, out = plan.bufs["trunk"], plan.embedding.buf
trunk.act.copy_(x)
st = net.stream()
plan.ws.zero_(); plan.ws_b.zero_()
plan.run_ops(plan.fwd, st)
dout = (0.05 * torch.randn(N, out.H, out.W, out.C, generator=g)).to(dt)
out.grad.copy_(dout)
plan.build_backward(None)
plan.run_ops(plan.bwd, st)
torch.cuda.synchronize()

for k in params:
    if not k.endswith(("/moving_mean", "/moving_variance")):
        params[k].requires_grad_(True)
xr = x.float().permute(0, 3, 1, 2).contiguous().requires_grad_(True)
o = Tap(params, dt)
blk = {"block35": fo.BLOCK35, "block17": fo.BLOCK17, "block8": fo.BLOCK8}[kind]
y = o._block(xr, f"{kind}/0", blk, scale, "relu", True)
y.backward(dout.float().permute(0, 3, 1, 2))
print(f"{kind} {dt} N={N}")
print(f"  out fwd        {rel(out.act.float().cpu(), y.detach().permute(0, 2, 3, 1)):.2e}")
for name, b in plan.bufs.items():
    if b.grad is None:
        continue
    print(f"  buffer {name:45s} |grad| {float(b.grad.float().norm()):.3e}")
# gradient w.r.t. raw conv outputs (= what the BN backward leaves in the grad buffers)
for prefix, t in o.raw.items():
    L = net.layers[prefix]
    gref = t.grad.permute(0, 2, 3, 1)
    if prefix.endswith("/up"):
        got = plan._dup[prefix].view(N, H, H, C).float().cpu()
    else:
        rec = next(r for r in plan.recs if r.kind == "conv" and r.layer is L)
        got = rec.y.buf.grad[..., rec.y.c0:rec.y.c0 + rec.y.C].float().cpu()
    print(f"  d(raw {prefix:42s}) rel err {rel(got, gref):.2e}   |ref| {float(gref.norm()):.3e}")
print(f"  dX             {rel(trunk.grad.float().cpu(), xr.grad.permute(0, 2, 3, 1)):.2e}")
mine = net.export_keras_grads(net.G)
for k, v in mine.items():
    if params[k].grad is not None:
        print(f"  dW {k:50s} {rel(v, params[k].grad):.2e}")
