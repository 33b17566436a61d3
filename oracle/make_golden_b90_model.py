#!/usr/bin/env python3
"""Adds the STORAGE-ROUNDING-MODEL trajectories to tests/golden/train_trajectory_b90.npz: the same five batch-90 triplet steps as
`make_golden.py b90`, fp32 arithmetic, but with weights and forward activations rounded to bf16 / f16 where the device stores them
(tests/quant_oracle.py).  They show how far ANY 16-bit-storage implementation drifts from the fp32 trajectory at lr 0.05 (the
trajectory is chaotic: 20 % in the loss after two steps for bf16, 10-15 % for f16), which is what the GPU test's bounds are set by.

    python oracle/make_golden_b90_model.py bf16 && python oracle/make_golden_b90_model.py f16      # ~10 min each on 8 cores
"""
import sys, json, numpy as np, torch
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import facenet_oracle as fo
from oracle.make_golden import B90_LAYERS
from tests.util_data import b90_batch
from tests.quant_oracle import quant_train_step_grads
torch.set_num_threads(min(8, torch.get_num_threads()))
dt = {"bf16": torch.bfloat16, "f16": torch.float16}[sys.argv[1]]
params, trainable, regularized = fo.build_params(128, seed=0)
opt = fo.AdamKeras(trainable, params, lr=0.05)
out = {"losses": [], "norms": []}
for step in range(5):
    data, grads, emb = quant_train_step_grads(params, trainable, b90_batch(step), "triplet", dt, alpha=0.2)
    out["losses"].append(data)
    out["norms"].append([grads[k].double().norm().item() for k in B90_LAYERS])
    for k in regularized:
        grads[k] = grads[k] + 2.0 * fo.L2_WEIGHT * params[k]
    opt.step(params, grads)
    # moving statistics are not used by training forwards
    print(sys.argv[1], step, data, out["norms"][-1], flush=True)
path = os.path.join(ROOT, "tests", "golden", "train_trajectory_b90.npz")
z = dict(np.load(path))
z[f"losses_{sys.argv[1]}_storage_model"] = np.array(out["losses"])
z[f"grad_norms_{sys.argv[1]}_storage_model"] = np.array(out["norms"])
np.savez_compressed(path, **z)
