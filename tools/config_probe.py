"""Developer probe: BASELINE.json configs 2, 3 and 5 at their stated sizes on ONE GPU (the multi-GPU configs
shard these same frames by tiles).  Each scene is checked against the oracle on a sub-rectangle before timing.
usage: python tools/config_probe.py [--configs 2,3,5] [--reps 20]"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402


def timed(fn, reps, torch):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ms = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        e1.synchronize()
        ms.append(e0.elapsed_time(e1))
    return float(np.median(ms)), float(min(ms))


def check(pkg, O, render, words, rect, what, n_secondary=0):
    u = O.Uniforms()
    for f in ("camera", "camera_inverse", "dimensions", "sun_dir"):
        getattr(u, f)[:] = list(getattr(render.uniforms, f))
    u.flags, u.misc_value = render.uniforms.flags, render.uniforms.misc_value
    x0, y0, w, h = rect
    if n_secondary:
        prim, sec = render.render_secondary(n_secondary, tile=rect)
        render.gpu.sync()
        oprim, osec = O.secondary_frame(words, u, n_secondary, tile=rect, threads=os.cpu_count())
        ok = np.array_equal(pkg.render.hits_to_numpy(prim).view(np.uint32), oprim.reshape(-1).view(np.uint32)) and \
            np.array_equal(pkg.render.hits_to_numpy(sec).view(np.uint32), osec.reshape(-1).view(np.uint32))
    else:
        got = pkg.render.hits_to_numpy(render.render(tile=rect))
        render.gpu.sync()
        ok = np.array_equal(got.view(np.uint32), O.trace_frame(words, u, tile=rect, threads=os.cpu_count()).reshape(-1).view(np.uint32))
    assert ok, f"{what}: GPU records differ from the oracle on {rect}"
    return True


def frame_stats(pkg, hits):
    h = pkg.render.hits_to_numpy(hits)
    return {"steps_mean": round(float((h["info"] & 0xFF).mean()), 2), "hit_frac": round(float(((h["info"] >> 16) & 1).mean()), 4)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="2,3,5")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--variant", type=int, default=1, help="0 RESTART, 1 STACK (svo_hip.h)")
    a = ap.parse_args()
    pkg, O = entry.load_package(), entry.load_oracle()
    import torch
    gpu = pkg.Gpu(0)
    gpu.set_option(pkg.gpu.OPT_VARIANT, a.variant)
    want = [int(c) for c in a.configs.split(",")]
    golden = os.path.join(ROOT, "tests", "golden")

    if 2 in want:  # monu9.vox, 1920x1080, default camera + 3 orbit poses
        z = np.load(os.path.join(golden, "monu9_vox.npz"))
        words = pkg.CpuOctree.from_voxels(int(z["size"][0]), z["xyzi"], z["palette"]).to_octree_words()
        W, H = 1920, 1080
        render = pkg.Render(gpu, (W, H), words, capacity=words.size)
        render.set_flags(pause_adaptive=True, shadows=False)
        poses = [((0.1, 0.2, -1.5), (0.0, 0.0, 1.5))]
        for k in range(3):  # orbit at radius 1.6, height 0.5, looking at the centre
            ang = 2 * math.pi * (k + 0.37) / 3
            p = (1.6 * math.sin(ang), 0.5, -1.6 * math.cos(ang))
            poses.append((p, (-p[0], -p[1], -p[2])))
        for i, (pos, look) in enumerate(poses):
            render.update(pkg.Settings(), pkg.Character(pos, look))
            check(pkg, O, render, words, (700, 400, 512, 256), f"config 2 pose {i}")
            hits = render.alloc_hits(W * H)
            med, best = timed(lambda: render.render(hits=hits), a.reps, torch)
            print(json.dumps({"config": 2, "scene": "monu9.vox", "words": int(words.size), "pose": i, "rays": W * H, "ms": round(med, 4),
                              "ms_min": round(best, 4), "mrays_s": round(W * H / med / 1e3, 1), **frame_stats(pkg, hits)}), flush=True)

    if 3 in want:  # .rsvo shell (depth 6) whose leaves instance the 16^3 blocks -> depth 10, fully expanded
        t0 = time.time()
        z = np.load(os.path.join(golden, "blocks_vox.npz"))
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
        print(f"config 3 scene: {words.size} words, depth {pkg.scenes.max_depth(words)}, built in {time.time() - t0:.1f}s", flush=True)
        W, H = 1920, 1080
        render = pkg.Render(gpu, (W, H), words, capacity=words.size)
        render.set_flags(pause_adaptive=True, shadows=False)
        for i, (pos, look) in enumerate((((0.3, 0.4, -1.6), (-0.2, -0.3, 1.5)), ((0.0, 0.0, -0.2), (0.3, 0.1, 1.0)))):  # outside / inside the shell
            render.update(pkg.Settings(), pkg.Character(pos, look))
            check(pkg, O, render, words, (700, 400, 512, 256), f"config 3 pose {i}")
            hits = render.alloc_hits(W * H)
            med, best = timed(lambda: render.render(hits=hits), a.reps, torch)
            print(json.dumps({"config": 3, "scene": "rsvo shell d6 + 16^3 blocks (synthetic stand-in)", "words": int(words.size), "pose": i,
                              "rays": W * H, "ms": round(med, 4), "ms_min": round(best, 4), "mrays_s": round(W * H / med / 1e3, 1),
                              **frame_stats(pkg, hits)}), flush=True)

    if 5 in want:  # depth-20 fractal, 3840x2160, 4 secondary rays per hit pixel
        t0 = time.time()
        words = pkg.scenes.fractal(seed=0, max_depth=20, cam=(-0.9999, -0.9999, -0.9999), lod_c=3000.0, min_depth=4,
                                   max_words=120_000_000)
        print(f"config 5 scene: {words.size} words, depth {pkg.scenes.max_depth(words)}, built in {time.time() - t0:.1f}s", flush=True)
        W, H = 3840, 2160
        gpu.set_option(pkg.gpu.OPT_TREE_DEPTH, 20)
        render = pkg.Render(gpu, (W, H), words, capacity=words.size)
        render.set_flags(pause_adaptive=True, shadows=True)
        render.update(pkg.Settings(), pkg.Character((-0.9990, -0.9985, -0.9980), (-1.0, -1.2, -0.9)))
        check(pkg, O, render, words, (1700, 900, 384, 192), "config 5", n_secondary=4)
        prim = render.alloc_hits(W * H)
        sec = render.alloc_hits(4 * W * H)
        med1, best1 = timed(lambda: render.render(hits=prim), a.reps, torch)
        med, best = timed(lambda: render.render_secondary(4, hits=prim, secondary=sec), a.reps, torch)
        st = frame_stats(pkg, prim)
        n_hit = int(round(st["hit_frac"] * W * H))
        print(json.dumps({"config": 5, "scene": "fractal depth 20", "words": int(words.size), "rays_nominal": 5 * W * H,
                          "rays_effective": W * H + 4 * n_hit, "ms_primary_only": round(med1, 4), "ms": round(med, 4), "ms_min": round(best, 4),
                          "mrays_s_nominal": round(5 * W * H / med / 1e3, 1), "mrays_s_effective": round((W * H + 4 * n_hit) / med / 1e3, 1),
                          **st}), flush=True)
        gpu.set_option(pkg.gpu.OPT_TREE_DEPTH, 16)


if __name__ == "__main__":
    main()
