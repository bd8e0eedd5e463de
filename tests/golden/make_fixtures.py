"""Regenerates the committed fixtures under tests/golden/.

Inputs: the reference's .vox ASSETS (data files, /root/reference/files/*.vox), decoded to voxel
lists in file order (the insertion order defines the node array).  Expected outputs: produced by
the CPU oracle (oracle/svo_oracle.c).  The reference has no tests or golden vectors of its own and
cannot be run here, so these pin the oracle against regressions and let the GPU box check the
HIP path without /root/reference; they are not reference outputs.

Run from the repo root:  python tests/golden/make_fixtures.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/files"


def main():
    # (monu10, defualt, phantom_mansion: their voxel lists pin the loader restatement through the node counts of SURVEY.md 8c KAT 3)
    for name in ("small", "monu9", "monu10", "defualt", "phantom_mansion"):
        data = open(os.path.join(REF, name + ".vox"), "rb").read()
        size, xyzi, pal = O.vox_parse(data)
        words = O.Tree.from_vox(data).to_octree()
        np.savez_compressed(os.path.join(HERE, f"{name}_vox.npz"), size=np.array(size, dtype=np.uint32), xyzi=xyzi,
                            palette=pal, n_words=np.array([words.size], dtype=np.uint64),
                            words_crc=np.array([int(np.bitwise_xor.reduce(words * np.arange(1, words.size + 1, dtype=np.uint32)))],
                                               dtype=np.uint64))
    # the eight 16^3 block models the reference's World instances (world.rs:19-58), decoded like the others
    blocks = {}
    for name in ("stone", "dirt", "grass", "wood", "leaf", "slate", "crystal", "glass"):
        size, xyzi, pal = O.vox_parse(open(os.path.join(os.path.dirname(REF), "blocks", name + ".vox"), "rb").read())
        assert size == (16, 16, 16)
        blocks[name + "_xyzi"] = xyzi
        blocks[name + "_palette"] = pal
    np.savez_compressed(os.path.join(HERE, "blocks_vox.npz"), **blocks)
    # config 1: small.vox, 256x256, default camera (main.rs:131-132), fov 90, static tree
    data = open(os.path.join(REF, "small.vox"), "rb").read()
    words = O.Tree.from_vox(data).to_octree()
    u = O.make_uniforms(width=256, height=256, flags=O.F_PAUSE_ADAPTIVE)
    hits, stats = O.trace_frame(words, u, stats=True, threads=4)
    np.savez_compressed(os.path.join(HERE, "config1_small_256.npz"), hits=hits.reshape(-1).view(np.uint32).reshape(-1, 4),
                        stats=stats.reshape(-1, 2), camera=np.array(u.camera, dtype=np.float32),
                        camera_inverse=np.array(u.camera_inverse, dtype=np.float32))
    # monu9: 4096 sampled pixels of the 1920x1080 frame, default camera
    data = open(os.path.join(REF, "monu9.vox"), "rb").read()
    words = O.Tree.from_vox(data).to_octree()
    u = O.make_uniforms(width=1920, height=1080, flags=O.F_PAUSE_ADAPTIVE)
    rng = np.random.default_rng(9)
    px = rng.integers(0, 1920, 4096)
    py = rng.integers(0, 1080, 4096)
    rec = np.empty((4096, 4), dtype=np.uint32)
    for i, (x, y) in enumerate(zip(px, py)):
        rec[i] = O.trace_frame(words, u, tile=(int(x), int(y), 1, 1)).reshape(-1).view(np.uint32)
    np.savez_compressed(os.path.join(HERE, "config2_monu9_samples.npz"), px=px.astype(np.uint32), py=py.astype(np.uint32),
                        hits=rec, camera=np.array(u.camera, dtype=np.float32),
                        camera_inverse=np.array(u.camera_inverse, dtype=np.float32))
    print("fixtures written to", HERE)


if __name__ == "__main__":
    main()
