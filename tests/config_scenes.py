"""The scenes and camera poses of BASELINE.json's configs 2..5 at their stated sizes, shared by the full-size parity
tests (tests/test_configs_full.py) and the developer probes under tools/.  Inputs are asset data decoded into
tests/golden/*.npz (voxel lists of the reference's .vox files) or seeded generators; nothing reads /root/reference."""
import math
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def config2(pkg):
    """files/monu9.vox -> 65 184 words, 1920x1080; default camera (main.rs:131-132) + 3 orbit poses."""
    z = np.load(os.path.join(GOLDEN, "monu9_vox.npz"))
    words = pkg.CpuOctree.from_voxels(int(z["size"][0]), z["xyzi"], z["palette"]).to_octree_words()
    poses = [((0.1, 0.2, -1.5), (0.0, 0.0, 1.5))]
    for k in range(3):  # orbit at radius 1.6, height 0.5, looking at the centre
        ang = 2 * math.pi * (k + 0.37) / 3
        p = (1.6 * math.sin(ang), 0.5, -1.6 * math.cos(ang))
        poses.append((p, (-p[0], -p[1], -p[2])))
    return words, poses, (1920, 1080)


def config3(pkg):
    """Stand-in for files/statuette.rsvo (absent from the checkout): a depth-6 .rsvo-format spherical shell whose leaves
    instance the reference's eight 16^3 block models (cpu_octree.rs:37, world.rs:19-58), fully expanded: 27.5 M words,
    depth 10.  Poses: outside the shell, and inside it (95 % hits)."""
    z = np.load(os.path.join(GOLDEN, "blocks_vox.npz"))
    world = pkg.World.new("")
    for i, name in enumerate(pkg.world.BLOCK_NAMES):
        world.insert(i + 1, pkg.CpuOctree.from_voxels(16, z[name + "_xyzi"], z[name + "_palette"]))
        world.generate_mip_tree(i + 1)
    depth = 6
    tree = pkg.CpuOctree.new(0)
    n = 1 << depth
    ax = (np.arange(n) + 0.5) / n * 2 - 1
    X, Y, Z = np.meshgrid(ax, ax, ax, indexing="ij")
    r = np.sqrt(X * X + Y * Y + Z * Z)
    for i, j, k in np.argwhere(np.abs(r - 0.75) < 1.0 / n):
        tree.put_in_voxel((float(ax[i]), float(ax[j]), float(ax[k])), pkg.Voxel(1, 1, 1), depth)
    world.insert(0, pkg.CpuOctree.load_octree(tree.to_rsvo(), depth))
    world.generate_mip_tree(0)
    octree = world.root_octree()
    world.expand(octree, max_depth=depth + 4)
    words = octree.raw_data()
    poses = [((0.3, 0.4, -1.6), (-0.2, -0.3, 1.5)), ((0.0, 0.0, -0.2), (0.3, 0.1, 1.0))]
    return words, poses, (1920, 1080)


def config4(pkg):
    """Procedural terrain, depth 16, 3840x2160 (the bench workload's tree and pose at 4K)."""
    cam, look = pkg.scenes.terrain_camera(0, 16)
    words = pkg.scenes.terrain(seed=0, max_depth=16, cam=cam, lod_c=1500.0, max_words=125_000_000)
    return words, [(cam, look)], (3840, 2160)


def config5(pkg):
    """Procedural fractal, depth 20 (87 M words), 3840x2160 with 4 secondary rays per hit pixel; needs SVO_OPT_TREE_DEPTH 20."""
    words = pkg.scenes.fractal(seed=0, max_depth=20, cam=(-0.9999, -0.9999, -0.9999), lod_c=3000.0, min_depth=4,
                               max_words=120_000_000)
    poses = [((-0.9990, -0.9985, -0.9980), (-1.0, -1.2, -0.9))]
    return words, poses, (3840, 2160)
