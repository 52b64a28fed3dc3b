"""Deterministic synthetic eye images: low-frequency sinusoid gradient + 45/135 degree
hard edges + +-4/255 uniform noise + a constant block, alpha = 255; and a uniform-random variant.
Seed convention: 0x5EED0000 + 2*pair + eye."""
import numpy as np


def seed_for(pair, eye):
    return 0x5EED0000 + 2 * pair + eye


def structured_u8(w, h, seed):
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w].astype(np.float32)
    ph = rng.uniform(0, 6.28, 3).astype(np.float32)
    img = np.ones((h, w, 4), np.float32)
    for c in range(3):
        img[..., c] = 0.5 + 0.35 * np.sin(x * (0.011 + 0.004 * c) + ph[c]) * np.cos(y * (0.008 + 0.003 * c) - ph[c])
    period = max(16, min(w, h) // 6)
    d45 = ((x + y) % period) < (period / 2)
    d135 = ((x - y) % (period * 1.5)) < (period * 0.5)
    img[..., 0] = np.where(d45, img[..., 0] * 0.35, img[..., 0])
    img[..., 1] = np.where(d135, 1.0 - img[..., 1] * 0.5, img[..., 1])
    img[..., 2] = np.where(d45 & d135, 0.95, img[..., 2])
    noise = rng.integers(-4, 5, size=(h, w, 3)).astype(np.float32) / 255.0
    img[..., :3] += noise
    by, bx = h // 3, w // 3
    img[by:by + max(4, h // 8), bx:bx + max(4, w // 8), :3] = np.array([0.25, 0.5, 0.75], np.float32)
    out = np.clip(np.floor(img * 255.0 + 0.5), 0, 255).astype(np.uint8)
    out[..., 3] = 255
    return out


def random_u8(w, h, seed):
    rng = np.random.default_rng(seed)
    out = rng.integers(0, 256, size=(h, w, 4), dtype=np.uint8)
    out[..., 3] = 255
    return out


def extremes_u8(w, h, seed):
    """Only 0 / 255 / a few mid values: exercises RCAS's 0*inf NaN paths and EASU's zero-gradient guard."""
    rng = np.random.default_rng(seed)
    vals = np.array([0, 0, 255, 255, 1, 254, 128], np.uint8)
    out = vals[rng.integers(0, len(vals), size=(h, w, 4))]
    blk = max(2, min(w, h) // 4)
    out[:blk, :blk, :3] = 0
    out[-blk:, -blk:, :3] = 255
    out[..., 3] = 255
    return out
