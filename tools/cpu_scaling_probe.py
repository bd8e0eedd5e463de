"""Developer probe: how the CPU baseline (oracle/svo_oracle.c) scales with threads on this host, and what the host offers
(affinity, cgroup CPU quota).  usage: python tools/cpu_scaling_probe.py [--threads 8,16,32,64,128,256]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402


def cgroup_cpus():
    for p in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            t = open(p).read().split()
            if p.endswith("cpu.max"):
                return None if t[0] == "max" else float(t[0]) / float(t[1])
            q = float(t[0])
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            return None if q < 0 else q / per
        except OSError:
            continue
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", default="8,16,32,64,128,256")
    ap.add_argument("--frac", type=int, default=1)
    a = ap.parse_args()
    print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), "cgroup quota (cpus)", cgroup_cpus(), flush=True)
    pkg = entry.load_package()
    O = entry.load_oracle()
    cam, look = pkg.scenes.terrain_camera(0, 16)
    words = pkg.scenes.terrain(seed=0, max_depth=16, cam=cam, lod_c=1500.0, max_words=125_000_000)
    W, H = 1920, 1080
    u = O.make_uniforms(pos=cam, look=look, width=W, height=H, flags=O.F_PAUSE_ADAPTIVE)
    rows = H // a.frac
    O.trace_frame(words, u, tile=(0, 0, W, rows), threads=16)
    for t in [int(x) for x in a.threads.split(",")]:
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            O.trace_frame(words, u, tile=(0, 0, W, rows), threads=t)
            best = min(best, time.perf_counter() - t0)
        print(f"threads {t:4d}: {W * rows / best / 1e6:7.2f} Mrays/s (best of 3, {best * 1e3:.0f} ms per frame)", flush=True)


if __name__ == "__main__":
    main()
