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

P, K, T = 12, 4, 10
n = P * K
params, _, _ = fo.build_params(128, seed=0)
fo.perturb_bn_stats(params, seed=1)
pools = [torch.from_numpy(structured_images(n, seed=30 + k)) for k in range(2)]

def scenario(graph):
    net = Network(embedding_size=128, device="cuda:0")
    net.load_keras_params(params)
    miner = TripletMiner(net, n, np.repeat(np.arange(P), K), T, seed=7)
    train_images = torch.zeros(3 * T, 160, 160, 3, dtype=torch.uint8, device="cuda:0")
    miner.build(train_images)
    miner.plan.images.copy_(pools[0]); miner.run(); torch.cuda.synchronize()
    ref = (miner.embn.clone(), net.W_infer.float().abs().sum().item(), net.fold_bias.abs().sum().item())
    print("graph" if graph else "eager", "eager-run: rows differ", float((ref[0][0] - ref[0][1]).abs().max()), "W_infer", ref[1], "fold_bias", ref[2])
    if graph:
        ev = make_events(miner.sched)
        runner = GraphRunner(net.device).capture(lambda: miner.run(ev))
        print("  after capture: W_infer", net.W_infer.float().abs().sum().item(), "P", net.P.abs().sum().item(), "S_var", net.S_var.sum().item())
        miner.plan.images.copy_(pools[0]); runner.replay(); torch.cuda.synchronize()
        print("  after replay: emb vs eager", float((miner.embn - ref[0]).abs().max()), "W_infer", net.W_infer.float().abs().sum().item(),
              "fold_bias", net.fold_bias.abs().sum().item(), "P", net.P.abs().sum().item(), "S_var", net.S_var.sum().item(), "table", net.table.sum().item())
    return None

scenario(False)
scenario(True)
