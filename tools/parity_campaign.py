"""Randomised parity campaign: many camera poses (inside and outside the cube, grazing, looking at corners) on several
scenes, STACK kernel against the oracle, bit-exact, both tie-break modes.  Not part of the test suite (minutes).
usage: python tools/parity_campaign.py [--poses 200] [--w 640 --h 360]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--poses", type=int, default=200)
    ap.add_argument("--w", type=int, default=640)
    ap.add_argument("--h", type=int, default=360)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--count", action="store_true", help="adaptive mode: also compare the hit counters after each frame (slower: the oracle counts on one thread)")
    ap.add_argument("--secondary", action="store_true", help="svo_render_secondary with 4 rays per hit pixel, random sun directions, ray 0 fused into the primary launch (SVO_OPT_FUSED_SHADOWS = 1): primary and secondary records against the oracle")
    ap.add_argument("--cull", type=int, default=2, help="SVO_OPT_CULL (1: cull whenever the camera is outside the cube); every 8th pose then stands far away")
    ap.add_argument("--variant", type=int, default=1, help="kernel variant (svo_hip.h): 0 RESTART, 1 STACK")
    ap.add_argument("--deep", action="store_true", help="deep trees instead (depth-20 fractal, depth-19 terrain): the 19-level stack instantiations, SVO_OPT_TREE_DEPTH = 20")
    a = ap.parse_args()
    pkg, O = entry.load_package(), entry.load_oracle()
    gpu = pkg.Gpu(0)
    gpu.set_option(pkg.gpu.OPT_VARIANT, a.variant)
    gpu.set_option(pkg.gpu.OPT_CULL, a.cull)
    if a.secondary:
        gpu.set_option(pkg.gpu.OPT_FUSED_SHADOWS, 1)
    cam, look = pkg.scenes.terrain_camera(0, 16)
    z = np.load(os.path.join(ROOT, "tests", "golden", "monu9_vox.npz"))
    scenes = {
        "terrain16": pkg.scenes.terrain(seed=0, max_depth=16, cam=cam, lod_c=600.0, max_words=40_000_000),
        "monu9": pkg.CpuOctree.from_voxels(int(z["size"][0]), z["xyzi"], z["palette"]).to_octree_words(),
        "random9": pkg.scenes.random_tree(seed=5, max_depth=9, p_split=0.55, p_solid=0.25, max_words=1 << 21),
        "fractal14": pkg.scenes.fractal(seed=1, max_depth=14, cam=(-0.9, -0.9, -0.9), lod_c=200.0, min_depth=4, max_words=8_000_000),
    }
    if a.deep:
        gpu.set_option(pkg.gpu.OPT_TREE_DEPTH, 20)
        scenes = {
            "fractal20": pkg.scenes.fractal(seed=0, max_depth=20, cam=(-0.9999, -0.9999, -0.9999), lod_c=1200.0, min_depth=4, max_words=30_000_000),
            "terrain19": pkg.scenes.terrain(seed=2, max_depth=19, cam=pkg.scenes.terrain_camera(2, 19)[0], lod_c=250.0, max_words=30_000_000),
        }
        for name, words in scenes.items():
            print(f"{name}: {words.size} words, depth {pkg.scenes.max_depth(words)}", flush=True)
    rng = np.random.default_rng(a.seed)
    threads = os.cpu_count() or 8
    total = bad = 0
    t0 = time.time()
    for name, words in scenes.items():
        render = pkg.Render(gpu, (a.w, a.h), words, capacity=words.size)
        for k in range(a.poses):
            kind = k % 4
            if kind == 0:    # inside the cube
                pos = rng.uniform(-0.95, 0.95, 3)
            elif kind == 1:  # outside, looking roughly at the cube
                pos = rng.normal(size=3); pos = pos / np.linalg.norm(pos) * (rng.uniform(3.0, 12.0) if k % 8 == 1 else rng.uniform(1.2, 3.0))
            elif kind == 2:  # on or near cell boundaries
                pos = np.round(rng.uniform(-1, 1, 3) * 16) / 16 + rng.choice([0.0, 1e-7, -1e-7, 1e-3], 3)
            else:            # near a face of the cube
                pos = rng.uniform(-0.99, 0.99, 3); pos[rng.integers(0, 3)] = rng.choice([-1.0, 1.0]) * rng.uniform(0.98, 1.02)
            target = rng.uniform(-0.8, 0.8, 3) if kind != 1 else rng.uniform(-0.3, 0.3, 3)
            if a.deep and k % 2 == 0:  # next to the finest geometry (the generators refine towards their LOD camera)
                base = np.array([-0.999, -0.9985, -0.998]) if name.startswith("fractal") else np.array(pkg.scenes.terrain_camera(2, 19)[0])
                pos = base + rng.normal(size=3) * 10.0 ** rng.uniform(-4.0, -1.5)
                target = base + rng.normal(size=3) * 0.01 - np.array([0.0, 0.02, 0.0])
            lookv = target - pos
            if np.linalg.norm(lookv) < 1e-3 or abs(lookv[0]) + abs(lookv[2]) < 1e-4:
                lookv = np.array([0.3, -0.2, 0.9])
            flags = (0 if a.count else O.F_PAUSE_ADAPTIVE) | (O.F_MISC_BOOL if (k // 4) % 2 else 0)
            u = O.make_uniforms(pos=tuple(float(x) for x in pos), look=tuple(float(x) for x in lookv), fov=float(rng.choice([60, 90, 120])),
                                width=a.w, height=a.h, flags=flags)
            if a.secondary:
                sun = rng.normal(size=3)
                if k % 5 == 0:
                    sun[rng.integers(0, 3)] = 0.0  # an axis-parallel component: octree_ray biases it
                u.sun_dir[:3] = [float(x) for x in sun]
                u.flags |= O.F_SHADOWS
            for f in ("camera", "camera_inverse", "dimensions", "sun_dir"):
                getattr(render.uniforms, f)[:] = list(getattr(u, f))
            render.uniforms.flags, render.uniforms.misc_value = u.flags, u.misc_value
            render.upload_uniforms()
            if a.count:
                render.write_nodes(words)  # counters back to zero
            buf = render.alloc_hits(a.w * a.h)
            buf.fill_(-1)
            total += 1
            if a.secondary:
                ns = 1 if a.count else 4  # with live counters: fs_main's own ray set (primary + shadow), which count_frame models
                sbuf = render.alloc_hits(ns * a.w * a.h)
                sbuf.fill_(-1)
                render.render_secondary(ns, hits=buf, secondary=sbuf)
                gpu.sync()
                oprim, osec = O.secondary_frame(words, u, ns, threads=threads)
                got = pkg.render.hits_to_numpy(buf).view(np.uint32)
                want = oprim.reshape(-1).view(np.uint32)
                gsec = pkg.render.hits_to_numpy(sbuf).view(np.uint32).reshape(ns, -1, 4)
                wsec = osec.reshape(ns, -1).view(np.uint32).reshape(ns, -1, 4)
                if not np.array_equal(gsec, wsec):
                    bad += 1
                    per_set = [int((gsec[j] != wsec[j]).any(axis=1).sum()) for j in range(ns)]
                    print(f"SECONDARY MISMATCH scene {name} pose {k} pos {pos.tolist()} look {lookv.tolist()} sun {list(u.sun_dir)[:3]} flags {u.flags}: differing records per set {per_set}", flush=True)
            else:
                got = pkg.render.hits_to_numpy(render.render(hits=buf)).view(np.uint32)
                gpu.sync()
                want = O.trace_frame(words, u, threads=threads).reshape(-1).view(np.uint32)
            if a.count and not np.array_equal(render.read_nodes(words.size), O.count_frame(words, u)):
                bad += 1
                print(f"COUNTER MISMATCH scene {name} pose {k} pos {pos.tolist()} look {lookv.tolist()} flags {flags}", flush=True)
            if not np.array_equal(got.reshape(-1), want):
                bad += 1
                diff = np.flatnonzero((got.reshape(-1, 4) != want.reshape(-1, 4)).any(axis=1))
                print(f"MISMATCH scene {name} pose {k} pos {pos.tolist()} look {lookv.tolist()} flags {flags}: {diff.size} rays, first {diff[:5].tolist()}", flush=True)
        print(f"{name}: {a.poses} poses done ({time.time() - t0:.0f} s), mismatching frames so far {bad}", flush=True)
    print(f"frames {total}, mismatching {bad}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
