import sys, torch
sys.path.insert(0, '.')
from facenet_amd.engine import Network, Lowering
from oracle import facenet_oracle as fo
from tests.util import structured_images
params, _, _ = fo.build_params(128, seed=0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 6
x = torch.from_numpy(structured_images(N, seed=7))
net = Network(embedding_size=128, device="cuda:0"); net.load_keras_params(params)
for training in (True, False):
    plan = net.plan(N, training=training)
    plan.images.copy_(x)
    snaps = []
    for rep in range(3):
        if training: plan.ws.zero_()
        Lowering.run_ops(plan.fwd, net.stream()); torch.cuda.synchronize()
        snaps.append({k: (b.act.float().clone(), None if b.raw is None else b.raw.float().clone()) for k, b in plan.bufs.items()})
    print('== training', training)
    shown = 0
    for k in plan.bufs:
        for which in (1, 0):
            a, b, c = snaps[0][k][which], snaps[1][k][which], snaps[2][k][which]
            if a is None: continue
            d1 = ((a - b).norm() / (a.norm() + 1e-20)).item(); d2 = ((a - c).norm() / (a.norm() + 1e-20)).item()
            if (d1 > 1e-6 or d2 > 1e-6) and shown < 12:
                nz = ((a - b).abs() > 0).float().mean().item()
                print(f"{k:45s} {'raw' if which else 'act'} rel diff {d1:.3e} {d2:.3e} frac-elems-differ {nz:.4f} shape {tuple(a.shape)}")
                shown += 1
