"""Drop-in for the reference's src/data.py, plus a seeded synthetic stand-in for tiny_nerf_data.npz
(the dataset blob is not shipped with the reference and cannot be downloaded here)."""
import os
from typing import Any, Dict

import numpy as np


def load_tiny_nerf_npz(path: str = "data/tiny_nerf_data.npz") -> Dict[str, Any]:
    """npz -> dict with 'images' (N,H,W,3), 'poses' (N,4,4), 'focal'; float64 arrays become float32.
    [reference src/data.py:4-13]"""
    with np.load(path) as z:
        return {k: (v.astype(np.float32) if v.dtype == np.float64 else v) for k, v in ((k, z[k]) for k in z.files)}


# ---------------------------------------------------------------------------- synthetic scene
_BOXES = (  # centre, half-size, colour: a blocky "bulldozer" on a base plate
    ((0.0, 0.0, -0.35), (1.1, 0.75, 0.12), (0.55, 0.55, 0.58)),
    ((0.15, 0.0, 0.05), (0.6, 0.45, 0.28), (0.95, 0.75, 0.05)),
    ((-0.25, 0.0, 0.55), (0.3, 0.32, 0.22), (0.85, 0.15, 0.1)),
    ((0.95, 0.0, -0.05), (0.12, 0.7, 0.2), (0.2, 0.25, 0.3)),
)
_DENSITY = 40.0


def _look_at(eye):
    fwd = -eye / np.linalg.norm(eye)
    right = np.cross(fwd, np.array([0.0, 0.0, 1.0])); right /= np.linalg.norm(right)
    up = np.cross(right, fwd)
    m = np.eye(4)
    m[:3, 0], m[:3, 1], m[:3, 2], m[:3, 3] = right, up, -fwd, eye
    return m


def _render_boxes(o, d):
    """Exact emission-absorption integral through constant-density boxes on a white background."""
    R = d.shape[0]
    t_in = np.full((R, len(_BOXES)), np.inf); seg = np.zeros((R, len(_BOXES)))
    inv = 1.0 / np.where(np.abs(d) < 1e-12, 1e-12, d)
    for k, (c, hs, _) in enumerate(_BOXES):
        c, hs = np.asarray(c), np.asarray(hs)
        t0 = (c - hs - o) * inv; t1 = (c + hs - o) * inv
        lo = np.minimum(t0, t1).max(-1); hi = np.maximum(t0, t1).min(-1)
        hit = hi > np.maximum(lo, 0.0)
        t_in[:, k] = np.where(hit, np.maximum(lo, 0.0), np.inf)
        seg[:, k] = np.where(hit, hi - np.maximum(lo, 0.0), 0.0)
    order = np.argsort(t_in, axis=1)
    cols = np.asarray([b[2] for b in _BOXES])
    T = np.ones(R); out = np.zeros((R, 3))
    for j in range(len(_BOXES)):
        k = order[:, j]
        a = 1.0 - np.exp(-_DENSITY * np.take_along_axis(seg, k[:, None], 1)[:, 0])
        out += (T * a)[:, None] * cols[k]
        T *= 1.0 - a
    return out + T[:, None]


def make_synthetic_scene(n_images: int = 106, H: int = 100, W: int = 100, focal: float = 138.88887889922103,
                         radius: float = 4.03, seed: int = 0) -> Dict[str, Any]:
    """Same schema as tiny_nerf_data.npz: cameras on the upper hemisphere looking at the origin
    (-z forward, +y up, no half-pixel offset: the conventions of src/rays.py:21-25)."""
    rng = np.random.RandomState(seed)
    poses, images = [], []
    jj, ii = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    cam = np.stack([(ii - W * 0.5) / focal, -(jj - H * 0.5) / focal, -np.ones_like(ii, dtype=np.float64)], -1).reshape(-1, 3)
    for _ in range(n_images):
        th, ph = rng.uniform(0, 2 * np.pi), rng.uniform(np.deg2rad(12), np.deg2rad(65))
        eye = radius * np.array([np.cos(ph) * np.cos(th), np.cos(ph) * np.sin(th), np.sin(ph)])
        m = _look_at(eye)
        d = cam @ m[:3, :3].T
        d /= np.linalg.norm(d, axis=-1, keepdims=True)
        images.append(_render_boxes(m[:3, 3][None], d).reshape(H, W, 3))
        poses.append(m)
    return {"images": np.stack(images).astype(np.float32), "poses": np.stack(poses).astype(np.float32),
            "focal": np.float32(focal)}


SYNTHETIC = "synthetic"


def load_scene(path: str = "data/tiny_nerf_data.npz", **synthetic_kwargs) -> Dict[str, Any]:
    """load_tiny_nerf_npz(path) — a missing file raises FileNotFoundError exactly like the reference (src/data.py:9,
    src/train.py:71) — unless `path` is the explicit sentinel "synthetic": then the seeded stand-in scene is built
    (tagged 'synthetic': True).  There is no silent fallback: a typo in --data-path must not train on fake data."""
    if path == SYNTHETIC:
        d = make_synthetic_scene(**synthetic_kwargs); d["synthetic"] = True
        return d
    d = load_tiny_nerf_npz(path); d["synthetic"] = False
    return d
