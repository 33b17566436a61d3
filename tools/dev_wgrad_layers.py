"""Per-layer weight-gradient times (ungrouped launches, batch 90): where the grouped launch's time goes."""
import sys, json, torch
sys.path.insert(0, '.')
import bench
from facenet_amd.engine import Network
from facenet_amd.train import Trainer
net = Network(embedding_size=128, device="cuda:0")
tr = Trainer(net, batch=90, loss="triplet", group_wgrad=False)
x = torch.randint(0, 256, (90, 160, 160, 3), dtype=torch.uint8)
tr.set_images(x)
tr.step_eager(); torch.cuda.synchronize()
ops = [op for op in tr.step_ops if op.name.startswith("conv_wgrad:")]
tr._zero()
t = bench.time_ops_individually(ops, net.stream(), net.lib)
rows = []
for op, ms in zip(ops, t):
    d = op.keep[0]
    fl = 2.0 * d.N * d.OH * d.OW * d.Cout * d.KH * d.KW * d.Cin
    rows.append((ms * 1e3, op.name.split(":", 1)[1], f"{d.H}x{d.W}x{d.Cin}->{d.Cout} k{d.KH}x{d.KW}s{d.stride}", fl / (ms * 1e-3) / 1e12))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print(f"{len(rows)} layers, {tot:.0f} us ungrouped")
for r in rows[:25]:
    print(f"{r[0]:8.1f} us {r[3]:7.1f} TF/s  {r[1]:45s} {r[2]}")
