"""Seeded synthetic inputs shared by oracle/make_golden.py and the tests (CPU only, no GPU imports)."""
import numpy as np
import torch


def c1_images(n=16):
    """np.random.default_rng(0).integers(0, 256, (16,160,160,3), uint8)."""
    return np.random.default_rng(0).integers(0, 256, (n, 160, 160, 3), dtype=np.uint8)


def structured_images(n, seed):
    """Synthetic uint8 images with low-frequency structure, texture and per-image brightness / contrast, so that
    different images produce different activations (pure white noise makes every image look alike to a CNN)."""
    g = torch.Generator().manual_seed(seed)
    out = []
    for _ in range(n):
        low = torch.rand(1, 3, 6, 6, generator=g)
        img = torch.nn.functional.interpolate(low, size=(160, 160), mode="bicubic", align_corners=False)
        fine = torch.rand(1, 3, 40, 40, generator=g)
        img = img + 0.25 * torch.nn.functional.interpolate(fine, size=(160, 160), mode="bilinear", align_corners=False)
        lo, hi = torch.rand(2, generator=g)
        img = (img - img.min()) / (img.max() - img.min())
        img = (0.5 * lo + (0.5 + 0.5 * hi - 0.5 * lo) * img) * 255
        out.append(img[0].permute(1, 2, 0).clamp(0, 255).to(torch.uint8))
    return torch.stack(out).numpy()


def triplet_pool(P=45, K=4, E=128, seed=0):
    """A P x K pool of unit-norm embeddings with overlapping classes and its labels: about a third
    of the anchor-positive pairs have margin-violating negatives, the rest exercise the top-up path."""
    rng = np.random.default_rng(seed)
    centers = rng.normal(size=(P, 1, E)).astype(np.float32)
    emb = (centers.repeat(K, 1) * 0.04 + rng.normal(size=(P, K, E)).astype(np.float32) * 0.05).reshape(P * K, E)
    emb /= np.linalg.norm(emb, axis=1, keepdims=True)
    return emb.astype(np.float32), np.repeat(np.arange(P), K)


def b90_batch(step: int):
    """The 90-image (30 triplets, rows a,p,n) batch of step `step` of the batch-90 training trajectory
    (tests/golden/train_trajectory_b90.npz): seeded structured images, a fresh batch per step."""
    return structured_images(90, seed=900 + step)
