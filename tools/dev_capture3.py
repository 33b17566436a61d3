"""Developer aid for DESIGN.md section 5: capture a 3-stream training schedule (FACENET_CAPTURE_MAX_STREAMS=3) under faulthandler."""
import faulthandler, os, sys
faulthandler.enable()
sys.path.insert(0, '.')
os.environ.setdefault("FACENET_AUTOTUNE", "0")
os.environ["FACENET_CAPTURE_MAX_STREAMS"] = sys.argv[1] if len(sys.argv) > 1 else "3"
import torch
from facenet_amd.engine import Network
from facenet_amd.train import Trainer
from tests.util_data import structured_images
n = int(os.environ["FACENET_CAPTURE_MAX_STREAMS"])
net = Network(embedding_size=128, device="cuda:0")
tr = Trainer(net, batch=6, loss="triplet", n_streams=n)
tr.set_images(torch.from_numpy(structured_images(6, seed=1)))
sched = tr.segments[0][0]
print("schedule:", sched.stats(), flush=True)
waits = [(s, x) for (k, s, x) in sched.steps if k == "wait"]
recs = {x: s for (k, s, x) in sched.steps if k == "record"}
print("cross-side waits:", sum(1 for (s, x) in waits if s > 0 and recs.get(x, 0) > 0), "waits on main:", sum(1 for (s, x) in waits if s == 0), flush=True)
with open("gpurun_out/steps3.txt", "w") as fh:
    fh.write(f"{sched.n_streams} {sched.n_events}\n")
    for (k, s_, x) in sched.steps:
        fh.write(f"{ {'wait': 'w', 'run': 'r', 'record': 'e'}[k] } {s_} {x}\n")
if len(sys.argv) > 2 and sys.argv[2] == "dump":
    sys.exit(0)
tr.capture()
tr.step(); torch.cuda.synchronize()
print("captured and replayed with", n, "streams; loss", tr.loss_value())
