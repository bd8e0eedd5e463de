"""Diagnostic: per-wave counters of the two-rays-per-lane kernel (trace_dual_kernel's debug instantiation) on the bench workload
or one of the BASELINE configs: loop iterations, live rays per iteration, DDA steps and loads per iteration, cycles per turn.
usage: python tools/dual_timeline.py [--scene terrain|config2|config3a|config3b] [--w 1920 --h 1080] [--json out.json]"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as entry  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="terrain")
    ap.add_argument("--w", type=int, default=1920)
    ap.add_argument("--h", type=int, default=1080)
    ap.add_argument("--json", default=None)
    ap.add_argument("--opt", action="append", default=[], help="NAME=VALUE for gpu.set_option (e.g. REFILL_MIN=8)")
    a = ap.parse_args()
    pkg = entry.load_package()
    import torch
    import config_scenes as cs
    W, H = a.w, a.h
    if a.scene == "terrain":
        words, poses, _ = cs.config4(pkg)
        pose = poses[0]
    elif a.scene == "config2":
        words, poses, _ = cs.config2(pkg)
        pose = poses[1]
    elif a.scene in ("config3a", "config3b"):
        words, poses, _ = cs.config3(pkg)
        pose = poses[0 if a.scene == "config3a" else 1]
    else:
        raise SystemExit("unknown scene")
    gpu = pkg.Gpu(0)
    render = pkg.Render(gpu, (W, H), words, capacity=words.size)
    render.set_flags(pause_adaptive=True, shadows=False)
    render.update(pkg.Settings(), pkg.Character(*pose))
    for o in a.opt:
        k, v = o.split("=")
        gpu.set_option(getattr(pkg.gpu, "OPT_" + k), int(v))
    dbg = torch.zeros((16384, 16), dtype=torch.int32, device="cuda")
    hits = render.alloc_hits(W * H)
    gpu.set_option(pkg.gpu.OPT_TIMING, 8)
    for _ in range(4):
        render.render(hits=hits)
    ms_plain = gpu.last_render_ms()
    gpu.sync()
    gpu.set_option(pkg.gpu.OPT_DEBUG_BUFFER, dbg.data_ptr())
    render.render(hits=hits)
    ms = gpu.last_render_ms()
    gpu.sync()
    gpu.set_option(pkg.gpu.OPT_DEBUG_BUFFER, 0)
    h = pkg.render.hits_to_numpy(hits)
    d = dbg.cpu().numpy().view(np.uint32)
    d = d[d[:, 2] != 0].astype(np.int64)
    t0 = d[:, 0].min()
    start, dry, end = (d[:, 0] - t0) * 0.01, (d[:, 1] - t0) * 0.01, (d[:, 2] - t0) * 0.01  # us
    pct = [0, 1, 5, 25, 50, 75, 95, 99, 100]
    it = max(int(d[:, 3].sum()), 1)
    life = d[:, 8] + d[:, 9] + d[:, 10]
    out = {
        "scene": a.scene, "w": W, "h": H, "words": int(words.size), "steps_mean": round(float((h["info"] & 0xFF).mean()), 2),
        "kernel_us_plain_build": round(ms_plain * 1e3, 1), "kernel_us_timeline_build": round(ms * 1e3, 1), "waves": int(len(d)),
        "pct": pct, "start_us": np.percentile(start, pct).round(1).tolist(), "dry_us": np.percentile(dry, pct).round(1).tolist(),
        "end_us": np.percentile(end, pct).round(1).tolist(),
        "iterations_per_wave": np.percentile(d[:, 3], pct).round(0).tolist(), "total_iterations": it,
        "live_rays_per_iteration_of_128": round(float(d[:, 4].sum() / it), 2),
        "steps_per_iteration": round(float(d[:, 12].sum() / it), 2), "loads_per_iteration": round(float(d[:, 13].sum() / it), 2),
        "loads_per_step": round(float(d[:, 13].sum() / max(d[:, 12].sum(), 1)), 3),
        "turns_per_step_of_a_live_ray": round(float(d[:, 4].sum() / max(d[:, 12].sum(), 1)), 3),
        "cycles_per_iteration": {"refill_incl_gen": round(float(d[:, 8].sum() / it), 1), "gen": round(float(d[:, 11].sum() / it), 1),
                                 "turn_a": round(float(d[:, 9].sum() / it), 1), "turn_b": round(float(d[:, 10].sum() / it), 1)},
        "claim_wait_cycles_per_wave": int(d[:, 15].mean()), "refills_per_iteration": round(float(d[:, 6].sum() / it), 3),
        "strips_generated": int(d[:, 7].sum()),
        "shader_clock_ghz_in_kernel": round(float(np.median(life / np.maximum((d[:, 2] - d[:, 0]) * 10.0, 1))), 3),
        "last_ray_steps_by_end_decile": [int(np.median(d[np.argsort(end)][i * len(d) // 10:(i + 1) * len(d) // 10, 5])) for i in range(10)],
    }
    print(json.dumps(out, indent=1))
    if a.json:
        with open(a.json, "w") as f:
            json.dump(out, f, indent=1)
    gpu.close()


if __name__ == "__main__":
    main()
