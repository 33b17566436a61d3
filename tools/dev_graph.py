"""Bisect HIP-graph capture of multi-stream schedules."""
import sys, torch, faulthandler
faulthandler.enable()
sys.path.insert(0, '.')
from facenet_amd.engine import Network
from facenet_amd.train import Trainer
from tests.util import structured_images
S = int(sys.argv[1]) if len(sys.argv) > 1 else 4
net = Network(embedding_size=128, device="cuda:0")
tr = Trainer(net, batch=6, loss="triplet", n_streams=S)
tr.set_images(torch.from_numpy(structured_images(6, 1)))
print('stats', tr.segments[0][0].stats(), flush=True)
tr.step_eager(); torch.cuda.synchronize(); print('eager ok', tr.loss_value(), flush=True)
tr.capture(); torch.cuda.synchronize(); print('capture ok', flush=True)
tr.step(); torch.cuda.synchronize(); print('replay ok', tr.loss_value(), flush=True)
