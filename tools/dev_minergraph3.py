import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("FACENET_AUTOTUNE", "0")
import numpy as np, torch
from facenet_amd.engine import Network
from facenet_amd.train import GraphRunner, TripletMiner
from facenet_amd.schedule import make_events
from oracle import facenet_oracle as fo
from tests.util_data import structured_images

def run(graph, variant):
    P, K, T, alpha, seed = 12, 4, 10, 0.2, 7
    n = P * K
    labels = np.repeat(np.arange(P), K)
    net = Network(embedding_size=128, device="cuda:0")
    params, _, _ = fo.build_params(128, seed=0)
    fo.perturb_bn_stats(params, seed=1)
    net.load_keras_params(params)
    miner = TripletMiner(net, n, labels, T, alpha=alpha, seed=seed)
    train_images = torch.zeros(3 * T, 160, 160, 3, dtype=torch.uint8, device="cuda:0")
    miner.build(train_images)
    pools = [torch.from_numpy(structured_images(n, seed=30 + k)) for k in range(3)]
    runner = None
    if graph:
        miner.plan.images.copy_(pools[0])
        miner.run()
        torch.cuda.synchronize()
        e0 = miner.embn.clone()
        if "noinfo" not in variant:
            miner.info.zero_()
        ev = make_events(miner.sched)
        runner = GraphRunner(net.device).capture(lambda: miner.run(ev))
        if "noinfo" not in variant:
            miner.info.zero_()
    for k in range(3):
        miner.plan.images.copy_(pools[k])
        runner.replay() if runner is not None else miner.run()
        torch.cuda.synchronize()
        emb = miner.embn.cpu()
        ref = fo.Oracle(params).forward(pools[k].numpy(), training=False)
        print(f"graph={graph} {variant} k={k}: emb-ref {(emb - ref).norm(dim=1).max().item():.3e}", (f"emb-eager0 {float((miner.embn - e0).norm(dim=1).max()):.3e}" if graph else ""))

for v in sys.argv[1:]:
    g, variant = v.split(":")
    run(g == "1", variant)
