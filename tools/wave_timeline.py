"""Diagnostic: per-wave timeline of the STACK kernel (start / work-queue-dry / end), shader cycles per phase (refill, ray
generation, descent, step) and the shape of the descent loop, on the bench workload or one of the BASELINE configs.
usage: python tools/wave_timeline.py [--scene terrain|config2|config3a|config3b] [--w 1920 --h 1080] [--json out.json]"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as entry  # noqa: E402


def progress(ev, d, t0):
    gens = ev[:, 1:15]
    gens = (gens[gens != 0] - t0) * 0.01
    enter, dry, end = (ev[:, 0] - t0) * 0.01, (d[:, 1] - t0) * 0.01, (d[:, 2] - t0) * 0.01
    rows = []
    for b in range(int(end.max() // 10) + 1):
        lo, hi = b * 10.0, b * 10.0 + 10.0
        rows.append({"t_us": lo, "strips_generated": int(((gens >= lo) & (gens < hi)).sum()), "waves_in_loop": int(((enter < hi) & (end >= lo)).sum()),
                     "waves_not_dry": int(((enter < hi) & (dry >= lo)).sum())})
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="terrain")
    ap.add_argument("--w", type=int, default=1920)
    ap.add_argument("--h", type=int, default=1080)
    ap.add_argument("--json", default=None)
    ap.add_argument("--cold", action="store_true", help="the measured frame is the first of a layout: its strip schedule is dropped just before (static tree only)")
    ap.add_argument("--raw", default=None, help="save the per-wave words (timeline, events) as .npz")
    ap.add_argument("--count", action="store_true", help="hit counters live, cleared before the frame (the reference's default mode)")
    ap.add_argument("--carry", action="store_true", help="with --count: the counters carry over from frame to frame (no scan in between)")
    ap.add_argument("--frames", type=int, default=0, help="frames before the measured one (default: 24, with --carry 8)")
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
    render.set_flags(pause_adaptive=not a.count, shadows=False)
    compute = pkg.Compute.new(gpu, render) if a.count else None
    if a.count:
        gpu.set_option(pkg.gpu.OPT_SCAN_CLEARS_COUNTERS, 1)

    def clear():
        if a.count and not a.carry:
            compute.update(int(words.size)); compute.read_lists()
    render.update(pkg.Settings(), pkg.Character(*pose))
    for o in a.opt:
        k, v = o.split("=")
        gpu.set_option(getattr(pkg.gpu, "OPT_" + k), int(v))
    dbg = torch.zeros((2 * 16384, 16), dtype=torch.int32, device="cuda")  # second half: per-wave event stamps (round 5)
    hits = render.alloc_hits(W * H)
    gpu.set_option(pkg.gpu.OPT_TIMING, 8)
    for _ in range(a.frames or (8 if a.carry else 24)):
        clear()
        render.render(hits=hits)
    ms_plain = gpu.last_render_ms()
    gpu.sync()
    clear()
    gpu.set_option(pkg.gpu.OPT_DEBUG_BUFFER, dbg.data_ptr())
    for _ in range(2):  # (the first launch of the timeline instantiation starts seven of the eight XCDs 100 us late: its scratch memory is set up then)
        dbg.zero_()
        clear()  # (--count without --carry: every frame starts from cleared counters, the measured one too)
        if a.cold:
            gpu.set_option(pkg.gpu.OPT_SCHEDULE, 2)  # (setting the period forgets the schedule)
        render.render(hits=hits)
        ms = gpu.last_render_ms()
        gpu.sync()
    gpu.set_option(pkg.gpu.OPT_DEBUG_BUFFER, 0)
    h = pkg.render.hits_to_numpy(hits)
    steps_mean = float((h["info"] & 0xFF).mean())
    hit_frac = float(((h["info"] >> 16) & 1).mean())
    d_all = dbg.cpu().numpy().view(np.uint32)
    live = d_all[:16384, 2] != 0
    ev = d_all[16384:][live].astype(np.int64)
    d = d_all[:16384][live].astype(np.int64)
    t0 = d[:, 0].min()
    if a.raw:
        np.savez_compressed(a.raw, d=d, ev=ev, wave=np.nonzero(live)[0])
    start, dry, end = (d[:, 0] - t0) * 0.01, (d[:, 1] - t0) * 0.01, (d[:, 2] - t0) * 0.01  # us
    pct = [0, 1, 5, 25, 50, 75, 95, 99, 100]
    rounds = d[:, 3].sum()
    life_cyc = d[:, 8] + d[:, 9] + d[:, 10]
    out = {
        "scene": a.scene, "w": W, "h": H, "words": int(words.size), "steps_mean": round(steps_mean, 2), "hit_frac": round(hit_frac, 4),
        "kernel_us_plain_build": round(ms_plain * 1e3, 1), "kernel_us_timeline_build": round(ms * 1e3, 1), "waves": int(len(d)),
        "pct": pct,
        "start_us": np.percentile(start, pct).round(1).tolist(), "dry_us": np.percentile(dry, pct).round(1).tolist(),
        "end_us": np.percentile(end, pct).round(1).tolist(), "drain_us": np.percentile(end - dry, pct).round(1).tolist(),
        "rounds_per_wave": np.percentile(d[:, 3], pct).round(0).tolist(),
        "us_per_round": np.percentile((end - start) / np.maximum(d[:, 3], 1), pct).round(2).tolist(),
        "total_rounds": int(rounds), "active_lanes_per_round": round(float(d[:, 4].sum() / max(rounds, 1)), 2),
        "refills_per_round": round(float(d[:, 6].sum() / max(rounds, 1)), 3), "strips_generated": int(d[:, 7].sum()),
        # phase clocks: mean shader cycles per wave, and per round
        "cycles_per_wave": {"refill_incl_gen": int(d[:, 8].mean()), "gen": int(d[:, 11].mean()), "descent": int(d[:, 9].mean()),
                            "step": int(d[:, 10].mean()), "sum": int(life_cyc.mean())},
        "cycles_per_round": {"refill_incl_gen": round(float(d[:, 8].sum() / max(rounds, 1)), 1), "gen": round(float(d[:, 11].sum() / max(rounds, 1)), 1),
                             "descent": round(float(d[:, 9].sum() / max(rounds, 1)), 1), "step": round(float(d[:, 10].sum() / max(rounds, 1)), 1)},
        "counting": {"cycles_per_round": round(float(d[:, 15].sum() / max(rounds, 1)), 1), "level_loop_iters_per_round": round(float(d[:, 12].sum() / max(rounds, 1)), 2),
                     "flush_cycles_per_round": round(float(d[:, 13].sum() / max(rounds, 1)), 1), "flushes_per_wave": round(float(d[:, 14].mean()), 2)} if a.count else None,
        "claim_wait_cycles_per_wave": int(d[:, 15].mean()) if not a.count else None,
        # (light build: slot 14 = cycles until the claim's answer is there, slot 15 = the same plus the schedule-list look-ups)
        "counter_probes_per_wave_pct": np.percentile(d[:, 12], pct).round(0).tolist(), "counter_probes_total": int(d[:, 12].sum()),  # (light build only)
        "claim_answer_cycles_pct": np.percentile(d[:, 14], pct).round(0).tolist(), "claim_total_cycles_pct": np.percentile(d[:, 15], pct).round(0).tolist(),
        "cycles_per_generated_strip": round(float(d[:, 11].sum() / max(d[:, 7].sum(), 1)), 1),
        "shader_clock_ghz_in_kernel": round(float(np.median(life_cyc / np.maximum((d[:, 2] - d[:, 0]) * 10.0, 1))), 3),
        # descent shape: wave-level iterations (dependent loads) per round vs the mean over lanes
        "descent": {"wave_iters_per_round": round(float(d[:, 12].sum() / max(rounds, 1)), 2),
                    "wave_iters_per_descending_round": round(float(d[:, 12].sum() / max(d[:, 14].sum(), 1)), 2),
                    "lanes_in_loop_per_iter": round(float(d[:, 13].sum() / max(d[:, 12].sum(), 1)), 2),
                    "cycles_per_wave_iter": round(float(d[:, 9].sum() / max(d[:, 12].sum(), 1)), 1)},
        # round 5: the frame's progress curve.  Per 10 us bin: strips generated (first 15 generations of every wave), waves that have
        # entered their main loop and not ended, waves that do not know yet that the lists are empty
        "loop_entry_us": np.percentile((ev[:, 0] - t0) * 0.01, pct).round(1).tolist(),
        "camera_walk_us": np.percentile((ev[:, 0] - d[:, 0]) * 0.01, pct).round(2).tolist(),
        "progress_10us": progress(ev, d, t0),
        "last_ray_steps_by_end_decile": [int(np.median(d[np.argsort(end)][i * len(d) // 10:(i + 1) * len(d) // 10, 5])) for i in range(10)],
    }
    print(json.dumps(out, indent=1))
    if a.json:
        with open(a.json, "w") as f:
            json.dump(out, f, indent=1)
    gpu.close()


if __name__ == "__main__":
    main()
