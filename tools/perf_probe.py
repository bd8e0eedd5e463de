"""Developer probe: time kernel variants / launch options on one scene in one process.
usage: python tools/perf_probe.py [--lod 1500] [--w 1920 --h 1080] [--scene terrain|monu9] [--reps 10]"""
import argparse
import itertools
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lod", type=float, default=1500.0)
    ap.add_argument("--w", type=int, default=1920)
    ap.add_argument("--h", type=int, default=1080)
    ap.add_argument("--scene", default="terrain")
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--max-words", type=int, default=125_000_000)
    ap.add_argument("--variants", default="0,1")
    ap.add_argument("--refill", default="16")
    ap.add_argument("--strip", default="64")
    ap.add_argument("--dynamic", default="1")
    ap.add_argument("--grid", default="0")
    ap.add_argument("--prio", default="0")
    ap.add_argument("--schedule", default="8")
    ap.add_argument("--block", default="3")
    ap.add_argument("--cull", default="2")
    ap.add_argument("--claim-at", default="64")
    ap.add_argument("--tail", default="0")
    ap.add_argument("--pose", type=int, default=0)
    ap.add_argument("--relayout", type=int, default=0, help="re-linearise the node array first (svo_nodes_relayout): subtrees below this level as contiguous blocks")
    ap.add_argument("--motion", type=float, default=0.0, help="degrees of yaw added to the camera every frame")
    a = ap.parse_args()
    pkg = entry.load_package()
    import torch
    t0 = time.time()
    if a.scene == "terrain":
        cam, look = pkg.scenes.terrain_camera(0, 16)
        words = pkg.scenes.terrain(seed=0, max_depth=16, cam=cam, lod_c=a.lod, max_words=a.max_words)
    else:
        z = np.load(os.path.join(ROOT, "tests/golden/monu9_vox.npz"))
        words = pkg.CpuOctree.from_voxels(int(z["size"][0]), z["xyzi"], z["palette"]).to_octree_words()
        cam, look = (0.1, 0.2, -1.5), (0.0, 0.0, 1.5)
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import config_scenes as cs
        if a.scene.startswith("config3"):
            words, poses, _ = cs.config3(pkg)
            cam, look = poses[1 if a.scene == "config3b" else 0]
        else:
            cam, look = cs.config2(pkg)[1][a.pose]
    if a.relayout:
        words = pkg.scenes.relayout(words, a.relayout)
    print(f"scene {a.scene}: {words.size} words ({words.size * 4 / 1e6:.1f} MB) built in {time.time() - t0:.1f}s", flush=True)
    gpu = pkg.Gpu(0)
    render = pkg.Render(gpu, (a.w, a.h), words, capacity=words.size)
    render.set_flags(pause_adaptive=True, shadows=False)
    render.update(pkg.Settings(), pkg.Character(cam, look))
    gpu.set_option(pkg.gpu.OPT_TIMING, 1)
    hits = render.alloc_hits(a.w * a.h)
    n = a.w * a.h
    def set_opt(opt, val):  # (older library builds under A/B lack the newer options)
        try:
            gpu.set_option(opt, val)
        except Exception:
            pass

    ref = None
    for tail, claim_at, cull, pairs, blockw, variant, refill, strip, dyn, grid, prio, sched in itertools.product(
            [int(x) for x in a.tail.split(",")], [int(x) for x in a.claim_at.split(",")], [int(x) for x in a.cull.split(",")], [0], [int(x) for x in a.block.split(",")],
            [int(x) for x in a.variants.split(",")], [int(x) for x in a.refill.split(",")],
            [int(x) for x in a.strip.split(",")], [int(x) for x in a.dynamic.split(",")],
            [int(x) for x in a.grid.split(",")], [int(x) for x in a.prio.split(",")],
            [int(x) for x in a.schedule.split(",")]):
        if variant == 0 and (refill, strip, dyn) != (int(a.refill.split(",")[0]), int(a.strip.split(",")[0]), int(a.dynamic.split(",")[0])):
            continue
        set_opt(pkg.gpu.OPT_VARIANT, variant)
        set_opt(pkg.gpu.OPT_REFILL_MIN, refill)
        set_opt(pkg.gpu.OPT_STRIP_ITEMS, strip)
        set_opt(pkg.gpu.OPT_DYNAMIC_STRIPS, dyn)
        set_opt(pkg.gpu.OPT_GRID_BLOCKS, grid)
        set_opt(pkg.gpu.OPT_PRIO_STEPS, prio)
        set_opt(pkg.gpu.OPT_SCHEDULE, sched)
        set_opt(pkg.gpu.OPT_BLOCK_SHAPE, blockw)
        set_opt(pkg.gpu.OPT_CULL, cull)
        if hasattr(pkg.gpu, "OPT_LATE_TAIL"):
            set_opt(pkg.gpu.OPT_LATE_TAIL, tail)
        if hasattr(pkg.gpu, "OPT_CLAIM_AT"):
            set_opt(pkg.gpu.OPT_CLAIM_AT, claim_at)  # (experimental builds only)
        ms = []
        for i in range(a.reps + 2):
            if a.motion:
                import math
                ang = math.radians(a.motion) * (i + 1)
                lx, ly, lz = look
                rl = (lx * math.cos(ang) + lz * math.sin(ang), ly, -lx * math.sin(ang) + lz * math.cos(ang))
                render.update(pkg.Settings(), pkg.Character(cam, rl))
            render.render(hits=hits)
            t = gpu.last_render_ms()
            if i >= 2:
                ms.append(t)
        gpu.sync()
        h = hits.cpu().numpy().view(np.uint32)
        sig = int(np.bitwise_xor.reduce(h.reshape(-1)))
        if ref is None:
            ref = h.copy()
        same = bool(np.array_equal(ref, h)) if not a.motion else None
        med = float(np.median(ms))
        print(json.dumps({"tail": tail, "claim_at": claim_at, "cull": cull, "variant": variant, "refill": refill, "strip": strip, "dynamic": dyn, "grid": grid, "prio": prio, "schedule": sched, "block": blockw,
                          "ms_med": round(med, 4), "ms_min": round(min(ms), 4), "mrays_s": round(n / med / 1e3, 1),
                          "sig": sig, "same_as_first": same}), flush=True)
    steps = (ref[:, 2] & 0xFF)
    print("steps mean", float(steps.mean()), "max", int(steps.max()), "hit frac", float(((ref[:, 2] >> 16) & 1).mean()),
          "cap frac", float((ref[:, 0] == 0xFF000000).mean()))


if __name__ == "__main__":
    main()
