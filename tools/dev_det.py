import sys, torch
sys.path.insert(0, '.')
from facenet_amd.engine import Network
from facenet_amd.train import Trainer
from oracle import facenet_oracle as fo
from tests.util import structured_images
params, _, _ = fo.build_params(128, seed=0)
x = torch.from_numpy(structured_images(6, seed=7))
def run(mode, S=2):
    net = Network(embedding_size=128, device="cuda:0"); net.load_keras_params(params)
    tr = Trainer(net, batch=6, loss="triplet", alpha=0.2, lr=0.01, n_streams=S); tr.set_images(x)
    if mode == 'serial':
        st = net.stream()
        for ops in (tr.pre_ops, tr.plan.fwd, tr.loss_ops, tr.plan.bwd, tr.opt_ops): tr.plan.run_ops(ops, st)
    else:
        tr.step_eager()
    torch.cuda.synchronize()
    return tr.loss_value(), tr.emb.clone(), tr.G.clone()
base = run('serial')
for mode, S in (('serial', 1), ('serial', 1), ('streams', 1), ('streams', 2), ('streams', 2), ('streams', 4)):
    l, e, g = run(mode, S)
    print(mode, S, 'loss', l, 'dloss', l - base[0], 'emb rel', ((e - base[1]).norm() / base[1].norm()).item(), 'G rel', ((g - base[2]).norm() / base[2].norm()).item())
