import sys, numpy as np, torch
sys.path.insert(0, '.')
from facenet_amd.engine import Network
from facenet_amd.train import Trainer
from oracle import facenet_oracle as fo
from tests.quant_oracle import quant_train_step_grads
E, N = 128, 9
params, trainable, regularized = fo.build_params(E, seed=0)
x = np.random.default_rng(3).integers(0, 256, (N, 160, 160, 3), dtype=np.uint8)
loss_ref, _, grads_ref, new_stats, emb_ref = fo.train_step_grads(params, trainable, [], x, "triplet", alpha=0.2)
for dt in (torch.float16, torch.bfloat16):
    lq, gq, eq = quant_train_step_grads(params, trainable, x, "triplet", dt)
    net = Network(embedding_size=E, device="cuda:0", train_dtype=dt)
    net.load_keras_params(params)
    tr = Trainer(net, batch=N, loss="triplet", alpha=0.2, l2=0.0)
    tr.set_images(torch.from_numpy(x))
    st = net.stream()
    tr._zero(); tr.plan.run_ops(tr.plan.fwd, st); tr.plan.run_ops(tr.loss_ops, st); tr.plan.run_ops(tr.plan.bwd, st)
    torch.cuda.synchronize()
    emb = tr.emb.float().cpu()
    print(dt, 'emb rel vs fp32', ((emb-emb_ref).norm()/emb_ref.norm()).item(), 'vs quant', ((emb-eq).norm()/eq.norm()).item(), 'quant vs fp32', ((eq-emb_ref).norm()/emb_ref.norm()).item())
    print('  loss', tr.loss_value(), 'fp32', loss_ref, 'quant', lq)
    mine = net.export_keras_grads(tr.G)
    for name, ref in (('fp32', grads_ref), ('quant', gq)):
        errs = sorted([((mine[k]-g).norm().item()/(g.norm().item()+1e-12), k) for k, g in ref.items() if g.norm().item() > 1e-4], reverse=True)
        print('  vs', name, 'worst', errs[:3], 'median', errs[len(errs)//2])
    errs = sorted([((gq[k]-g).norm().item()/(g.norm().item()+1e-12), k) for k, g in grads_ref.items() if g.norm().item() > 1e-4], reverse=True)
    print('  quant vs fp32: worst', errs[:2], 'median', errs[len(errs)//2])
