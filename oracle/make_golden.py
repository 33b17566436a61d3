#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the CPU oracle (oracle/facenet_oracle.py).

SELF-GENERATED, NOT REFERENCE-DERIVED: sMedX/FaceNet ships no fixtures, golden vectors or weights and
TensorFlow is not installable here, so these vectors pin the ORACLE (against drift) and give the GPU tests committed
expected outputs; they cannot pin the oracle to the reference ("parity unpinned").

    python oracle/make_golden.py                      # rewrites every fixture under tests/golden/
    python oracle/make_golden.py triplets b90         # only the named sections: c1 image triplets trajectory b90
Inputs are re-created from seeds by the tests (numpy default_rng / torch.Generator), only outputs are stored.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import facenet_oracle as fo  # noqa: E402
from tests.util_data import b90_batch, c1_images, structured_images, triplet_pool  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


B90_LAYERS = ("conv2d/Conv2d_1a_3x3/kernel", "conv2d/Conv2d_4b_3x3/kernel", "block35/2/up/kernel",
              "reduction_a/tower_conv1/Conv2d_0b_3x3/kernel", "block17/4/tower_conv1/Conv2d_0b_1x7/kernel", "block17/9/up/kernel",
              "block8/2/up/kernel", "features/logits/kernel")


def section_c1():
    # ---- C1: forward embeddings (BASELINE.json configs[0]) --------------------------------
    x = c1_images()
    emb = {}
    for E in (128, 512):
        for variant in ("fresh", "perturbed"):
            params, _, _ = fo.build_params(E, seed=0)
            if variant == "perturbed":
                fo.perturb_bn_stats(params, seed=1)
            o = fo.Oracle(params)
            if E == 128 and variant == "perturbed":
                o.taps = {}
            e = o.forward(x, training=False)
            emb[f"emb_{E}_{variant}"] = e.numpy().astype(np.float32)
            if o.taps is not None:
                taps = {}
                for k, v in o.taps.items():
                    v = v.detach()
                    flat = v.reshape(v.shape[0], -1)
                    taps[k.replace("/", "__")] = np.array([v.mean().item(), v.std().item(), v.abs().max().item(),
                                                           flat[0, :8].sum().item(), flat[-1, -8:].sum().item()], dtype=np.float64)
                np.savez_compressed(os.path.join(OUT, "c1_taps_128_perturbed.npz"), **taps)
    np.savez_compressed(os.path.join(OUT, "c1_embeddings.npz"), **emb)


def section_image():
    # ---- image processing (facenet.py:67-86) ---------------------------------------------------------------
    xi = c1_images(3)
    xi[1] = 77
    ip = {f"mode{m}": fo.image_processing(xi, m).numpy()[:, ::16, ::16, :].astype(np.float32) for m in (0, 1)}
    np.savez_compressed(os.path.join(OUT, "image_processing.npz"), **ip)


def section_triplets():
    # ---- triplet selection (build-defined, A13) ------------------------------------------------------------
    embp, labels = triplet_pool()
    dist = fo.squared_distance_matrix(embp)          # device summation order: bit-identical to fn_pairwise_sqdist
    import hashlib
    sel = {"dist_checksum": np.array([dist.sum(dtype=np.float64), (dist ** 2).sum(dtype=np.float64)]),
           "dist_sha1": np.frombuffer(hashlib.sha1(np.ascontiguousarray(dist).tobytes()).digest(), dtype=np.uint8)}
    for seed in (0, 7):
        for semi in (0, 1):
            sel[f"triplets_seed{seed}_semi{semi}"] = fo.select_triplets(dist, labels, 0.2, 30, seed, semi_hard=bool(semi))
    sims = fo.pairwise_similarities(embp[:40], None, 0)
    sel["pairwise_metric0_triu_first40"] = sims.astype(np.float32)
    np.savez_compressed(os.path.join(OUT, "triplets.npz"), **sel)


def section_trajectory():
    # ---- 3-step training trajectories (loss + parameter checksums) -----------------------------------------
    traj = {}
    for kind in ("triplet", "softmax"):
        N = 9 if kind == "triplet" else 8
        ncls = None if kind == "triplet" else 37
        params, trainable, regularized = fo.build_params(128, seed=0, nrof_classes=ncls)
        xs = structured_images(N, seed=3 if kind == "triplet" else 4)
        labels_s = np.random.default_rng(5).integers(0, 37, N)
        opt = fo.AdamKeras(trainable, params, lr=0.01)
        losses, sums = [], []
        for step in range(3):
            data, total, grads, stats, _ = fo.train_step_grads(params, trainable, regularized, xs, kind, labels=labels_s, alpha=0.2)
            opt.step(params, grads)
            for k, v in stats.items():
                params[k].copy_(v)
            losses.append([data, total])
            sums.append([params[k].double().sum().item() for k in ("conv2d/Conv2d_1a_3x3/kernel", "block17/4/up/kernel",
                                                                   "features/logits/kernel", "features/bn/beta",
                                                                   "conv2d/Conv2d_4b_3x3/bn/moving_variance")])
        traj[f"{kind}_losses"] = np.array(losses)
        traj[f"{kind}_param_sums"] = np.array(sums)
    np.savez_compressed(os.path.join(OUT, "train_trajectory.npz"), **traj)


def section_b90():
    """Training fidelity at the REAL configuration (BASELINE.json configs[1]): 5 triplet steps at batch 90 (30 triplets), E = 128,
    Keras Adam(eps 0.1) lr 0.05, L2 5e-4, fp32 oracle.  Per step: data loss, total loss, and the L2 norm of the DATA gradient
    (without the coupled L2 term, which the device adds inside the optimiser) of eight kernels spread over the depth."""
    params, trainable, regularized = fo.build_params(128, seed=0)
    opt = fo.AdamKeras(trainable, params, lr=0.05)
    losses, norms, emb_norm = [], [], []
    for step in range(5):
        data, total, grads, stats, emb = fo.train_step_grads(params, trainable, regularized, b90_batch(step), "triplet", alpha=0.2)
        norms.append([(grads[k] - 2.0 * fo.L2_WEIGHT * params[k]).double().norm().item() for k in B90_LAYERS])
        emb_norm.append(emb.double().norm(dim=1).mean().item())
        opt.step(params, grads)
        for k, v in stats.items():
            params[k].copy_(v)
        losses.append([data, total])
        print(f"b90 step {step}: loss {data:.5f} total {total:.5f} |g| {norms[-1]}", flush=True)
    np.savez_compressed(os.path.join(OUT, "train_trajectory_b90.npz"), losses=np.array(losses), grad_norms=np.array(norms),
                        emb_norm=np.array(emb_norm), layers=np.array(B90_LAYERS))


SECTIONS = {"c1": section_c1, "image": section_image, "triplets": section_triplets, "trajectory": section_trajectory, "b90": section_b90}


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(min(8, torch.get_num_threads()))
    want = sys.argv[1:] or list(SECTIONS)
    for name in want:
        if name not in SECTIONS:
            raise SystemExit(f"unknown section {name!r}; choose from {list(SECTIONS)}")
    for name in want:
        SECTIONS[name]()
    print("golden fixtures written to", OUT, want)


if __name__ == "__main__":
    main()
