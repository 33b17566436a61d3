"""Developer aid: does a captured TripletMiner replay reproduce the eager run?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("FACENET_AUTOTUNE", "0")
import numpy as np, torch
from facenet_amd.engine import Network
from facenet_amd.train import GraphRunner, TripletMiner
from facenet_amd.schedule import make_events
from tests.util_data import structured_images

P, K, T = 12, 4, 10
n = P * K
net = Network(embedding_size=128, device="cuda:0")
miner = TripletMiner(net, n, np.repeat(np.arange(P), K), T, seed=7, group=("nogroup" not in sys.argv))
train_images = torch.zeros(3 * T, 160, 160, 3, dtype=torch.uint8, device="cuda:0")
miner.build(train_images)
pools = [torch.from_numpy(structured_images(n, seed=30 + k)) for k in range(2)]
eager = []
for k in range(2):
    miner.plan.images.copy_(pools[k]); miner.run(); torch.cuda.synchronize()
    eager.append((miner.embn.clone(), miner.plan.bufs["input"].act.clone(), miner.plan.bufs["conv2d/Conv2d_1a_3x3"].act.clone()))
print("eager rows differ:", float((eager[0][0][0] - eager[0][0][1]).abs().max()), "pools differ:", float((eager[0][0] - eager[1][0]).abs().max()))
ev = make_events(miner.sched)
runner = GraphRunner(net.device).capture(lambda: miner.run(ev))
for k in range(2):
    miner.plan.images.copy_(pools[k]); runner.replay(); torch.cuda.synchronize()
    e, i, c = miner.embn, miner.plan.bufs["input"].act, miner.plan.bufs["conv2d/Conv2d_1a_3x3"].act
    print(f"graph pool {k}: emb vs eager {float((e - eager[k][0]).abs().max()):.3e}  input {float((i.float() - eager[k][1].float()).abs().max()):.3e} "
          f"conv1a {float((c.float() - eager[k][2].float()).abs().max()):.3e}  rows differ {float((e[0] - e[1]).abs().max()):.3e}")
for i, op in enumerate(miner.ops[:6]):
    print(i, op.name)
