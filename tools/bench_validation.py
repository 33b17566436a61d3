"""Times one face-to-face validation pass at the reference's scale (530 classes / 26 489 embeddings of size 512,
models/20200724-231357/logs/report.txt:13-22; 693-1 547 s per pass in the reference's logs, :47,647) on the GPU, and the
NumPy restatement on a bounded sample."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from facenet_amd.config import Config
from facenet_amd.statistics import FaceToFaceValidation
from oracle import statistics_oracle as so

def pool(C, per, E, seed=0):
    rng = np.random.default_rng(seed)
    cen = rng.normal(size=(C, 1, E)).astype(np.float32)
    sizes = rng.integers(per[0], per[1] + 1, C)
    emb = np.concatenate([cen[c] * 0.6 + rng.normal(size=(sizes[c], E)).astype(np.float32) * 0.7 for c in range(C)])
    emb /= np.linalg.norm(emb, axis=1, keepdims=True)
    return emb.astype(np.float32), np.repeat(np.arange(C), sizes)

emb, labels = pool(530, (49, 51), 512)
print("embeddings", emb.shape, "classes", 530)
cfg = Config({"metric": 0, "nrof_folds": 10, "far_target": 1e-3})
FaceToFaceValidation(emb[:2000], labels[:2000], cfg)                       # warm-up
torch.cuda.synchronize(); t0 = time.perf_counter()
v = FaceToFaceValidation(emb, labels, cfg)
torch.cuda.synchronize(); gpu_s = time.perf_counter() - t0
print(f"GPU 10-fold validation: {gpu_s:.2f} s; accuracy {v.dict['MaximumAccuracy']['accuracy']:.5f}")
e2, l2 = pool(40, (49, 51), 512, seed=1)
t0 = time.perf_counter(); so.face_to_face_validation(e2, l2, 0, nrof_folds=10); cpu_s = time.perf_counter() - t0
scale = (530 / 40) ** 2
print(f"NumPy restatement on 40 classes / {len(l2)} embeddings: {cpu_s:.1f} s  (x{scale:.0f} class pairs at full scale ~ {cpu_s * scale:.0f} s)")
