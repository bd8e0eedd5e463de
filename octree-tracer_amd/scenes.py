"""Deterministic scene generators and camera poses for the benchmark configs.
No reference counterpart: the reference's procedural generator (procedual.wgsl) is racy."""
import ctypes as C

import numpy as np

from ._lib import TerrainParams, lib


def _gen(fn, params, max_words):
    buf = np.empty(max_words, dtype=np.uint32)  # pages are committed only as they are written
    n = fn(C.byref(params), buf.ctypes.data, max_words)
    return buf[:n]


def _params(seed, max_depth, cam, lod_c, min_depth, max_words):
    p = TerrainParams()
    p.seed, p.max_depth, p.lod_c, p.min_depth, p.max_words = seed, max_depth, lod_c, min_depth, max_words
    p.cam[:] = list(cam)
    return p


def terrain(seed=0, max_depth=16, cam=(0.0, 0.0, 0.0), lod_c=3000.0, min_depth=6, max_words=120_000_000):
    return _gen(lib().svo_gen_terrain, _params(seed, max_depth, cam, lod_c, min_depth, max_words), max_words)


def fractal(seed=0, max_depth=20, cam=(0.0, 0.0, 0.0), lod_c=3000.0, min_depth=5, max_words=120_000_000):
    return _gen(lib().svo_gen_fractal, _params(seed, max_depth, cam, lod_c, min_depth, max_words), max_words)


def random_tree(seed=0, max_depth=8, p_split=0.5, p_solid=0.3, max_words=1 << 20):
    buf = np.empty(max_words, dtype=np.uint32)
    n = lib().svo_gen_random(seed, max_depth, p_split, p_solid, max_words, buf.ctypes.data, max_words)
    return buf[:n].copy()


def terrain_height_world(seed, max_depth, x, z):
    """world-space y of the terrain surface above world (x, z)"""
    R = 1 << max_depth
    X = min(R - 1, max(0, int((x + 1.0) * 0.5 * R)))
    Z = min(R - 1, max(0, int((z + 1.0) * 0.5 * R)))
    h = lib().svo_gen_terrain_height(seed, max_depth, X, Z)
    return -1.0 + 2.0 * h / R


def terrain_camera(seed=0, max_depth=16, x=0.05, z=-0.6, eye_height=0.004, look=(0.15, -0.25, 1.0)):
    """A pose standing just above the terrain (inside the cube, so the finest LOD is in view)."""
    y = terrain_height_world(seed, max_depth, x, z) + eye_height
    return (x, y, z), tuple(look)


def max_depth(words):
    words = np.ascontiguousarray(words, dtype=np.uint32)
    return lib().svo_nodes_max_depth(words.ctypes.data, words.size)


def relayout(words, block_level=10, with_perm=False):
    """svo_nodes_relayout: the same tree, levels 1 .. block_level breadth-first, every subtree below as one contiguous block."""
    words = np.ascontiguousarray(words, dtype=np.uint32)
    out = np.empty(words.size, dtype=np.uint32)
    perm = np.empty(words.size, dtype=np.uint32) if with_perm else None
    n = lib().svo_nodes_relayout(words.ctypes.data, words.size, block_level, out.ctypes.data, perm.ctypes.data if with_perm else None)
    if n == 0:
        raise ValueError("svo_nodes_relayout: malformed tree")
    return (out[:n], perm[:n]) if with_perm else out[:n]
