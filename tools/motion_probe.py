"""Developer probe: the strip schedule under camera motion.
--pairs: per step of yaw, frame A is traced with the schedule built from the previous pose's frame (stale), frame B -- the
same pose again -- with the schedule built from A's own step counts (exact): what staleness costs.
--hold: pose B traced with its own schedule, kept, while another pose is traced in between: is it the schedule or the caches?
--classes: how well one frame's strip cost classes predict the next frame's (in place and shifted by the best whole-pixel
shift), and a map of where the strips with step-limit rays are.
default: continuous motion, one frame per pose, kernel time per frame for several yaw rates and schedule periods; also a
diagonal motion (yaw + pitch) and a translation.  usage: python tools/motion_probe.py [--deg 0.25,1,3] [--steps 40] [--pairs]"""
import argparse, math, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
ap = argparse.ArgumentParser()
ap.add_argument("--deg", default="0.25,1.0,3.0")
ap.add_argument("--steps", type=int, default=30)
ap.add_argument("--period", default="1,2")
ap.add_argument("--pairs", action="store_true")
ap.add_argument("--classes", action="store_true", help="how well do a frame's strip cost classes predict the next frame's?")
ap.add_argument("--floor", default="0", help="SVO_OPT_SCHEDULE_MOTION values to compare (continuous mode), e.g. 0,0x2205")
ap.add_argument("--hold", action="store_true", help="exact schedule held while another pose is traced in between: schedule or caches?")
a = ap.parse_args()
pkg = entry.load_package()
cam, look = pkg.scenes.terrain_camera(0, 16)
words = pkg.scenes.terrain(seed=0, max_depth=16, cam=cam, lod_c=1500.0, max_words=125_000_000)
gpu = pkg.Gpu(0)
W, H = 1920, 1080
render = pkg.Render(gpu, (W, H), words, capacity=words.size)
render.set_flags(pause_adaptive=True, shadows=False)
gpu.set_option(pkg.gpu.OPT_TIMING, 4)
hits = render.alloc_hits(W * H)
lx, ly, lz = look
import itertools


def pose(kind, deg, i):
    ang = math.radians(deg) * (i + 1)
    if kind == "yaw":
        return cam, (lx * math.cos(ang) + lz * math.sin(ang), ly, -lx * math.sin(ang) + lz * math.cos(ang))
    if kind == "yaw+pitch":
        return cam, (lx * math.cos(ang) + lz * math.sin(ang), ly + 0.5 * math.sin(ang), -lx * math.sin(ang) + lz * math.cos(ang))
    return (cam[0] + 0.0004 * deg * (i + 1), cam[1], cam[2] + 0.0008 * deg * (i + 1)), look  # walking


if a.classes:
    def steps_of(p):
        render.update(pkg.Settings(), pkg.Character(*p)); render.render(hits=hits); gpu.sync()
        return (hits[:, 2] & 0xFF).reshape(H, W).cpu().numpy().astype(np.int32)

    def classes(st):  # 8x8 strips, class = min(max steps / 8, 15)
        return np.minimum(st.reshape(H // 8, 8, W // 8, 8).max(axis=(1, 3)) >> 3, 15)

    for deg in [float(x) for x in a.deg.split(",")]:
        s0, s1 = steps_of(pose("yaw", deg, 0)), steps_of(pose("yaw", deg, 1))
        c0, c1 = classes(s0), classes(s1)
        print(f"yaw {deg} deg: steps mean {s1.mean():.2f}, max {s1.max()}, rays >= 64 steps: {(s1 >= 64).sum()}, >= 96: {(s1 >= 96).sum()}")
        print("  class histogram (next frame):", np.bincount(c1.ravel(), minlength=16).tolist())
        if deg == float(a.deg.split(",")[0]):  # where the long strips are: share of class >= 12 strips per 2x2 strips, '.' none .. '4' all
            m = (c1 >= 12)[: H // 16 * 2, : W // 16 * 2].reshape(H // 16, 2, W // 16, 2).sum(axis=(1, 3))
            for row in m[::2]:
                print("  " + "".join(".1234"[v] for v in row))
        best = None
        for sx in range(-40, 41):
            sh = np.roll(s0, sx, axis=1)
            err = np.abs(classes(sh) - c1).sum()
            if best is None or err < best[0]:
                best = (err, sx)
        cp = classes(np.roll(s0, best[1], axis=1))
        for name, c in (("in place", c0), (f"shifted by {best[1]} px (best)", cp)):
            hi = c1 >= 8
            print(f"  {name}: strips of class >= 8 next frame: {hi.sum()}, of them predicted >= 8: {(c[hi] >= 8).sum()}, predicted < 4: {(c[hi] < 4).sum()};"
                  f" mean |class error| {np.abs(c - c1).mean():.3f}; strips under-predicted by >= 4 classes: {(c1 - c >= 4).sum()}")
elif a.hold:
    for deg in [float(x) for x in a.deg.split(",")]:
        gpu.set_option(pkg.gpu.OPT_SCHEDULE, 1024)  # (drops the schedule; the next frame builds one, kept for 1024 frames)
        pb, pa = pose("yaw", deg, 0), pose("yaw", deg, 1)
        render.update(pkg.Settings(), pkg.Character(*pb)); render.render(hits=hits); gpu.sync()
        same, alt, other = [], [], []
        for i in range(a.steps):
            render.render(hits=hits); gpu.sync(); same.append(gpu.last_render_ms())
        for i in range(a.steps):
            render.update(pkg.Settings(), pkg.Character(*pa)); render.render(hits=hits); gpu.sync(); other.append(gpu.last_render_ms())
            render.update(pkg.Settings(), pkg.Character(*pb)); render.render(hits=hits); gpu.sync(); alt.append(gpu.last_render_ms())
        print(f"hold, poses {deg} deg apart: pose B with its own schedule, repeated {np.median(same):.4f} ms; alternating with pose A "
              f"{np.median(alt):.4f} ms; pose A with B's schedule {np.median(other):.4f} ms", flush=True)
elif a.pairs:
    gpu.set_option(pkg.gpu.OPT_SCHEDULE, 1)
    for deg in [float(x) for x in a.deg.split(",")]:
        ta, tb = [], []
        for i in range(a.steps + 3):
            render.update(pkg.Settings(), pkg.Character(*pose("yaw", deg, i)))
            render.render(hits=hits); gpu.sync(); t1 = gpu.last_render_ms()
            render.render(hits=hits); gpu.sync(); t2 = gpu.last_render_ms()
            if i >= 3:
                ta.append(t1); tb.append(t2)
        print(f"yaw {deg} deg/step: stale schedule {np.median(ta):.4f} ms, exact schedule {np.median(tb):.4f} ms (kernel, median of {a.steps})", flush=True)
else:
    for kind, period, deg, fl in itertools.product(["yaw", "yaw+pitch", "walk"], [int(x) for x in a.period.split(",")],
                                                   [float(x) for x in a.deg.split(",")], [int(x, 0) for x in a.floor.split(",")]):
        gpu.set_option(pkg.gpu.OPT_SCHEDULE, period)
        gpu.set_option(pkg.gpu.OPT_SCHEDULE_MOTION, fl)
        ts = []
        for i in range(a.steps + 4):
            render.update(pkg.Settings(), pkg.Character(*pose(kind, deg, i)))
            render.render(hits=hits); gpu.sync()
            if i >= 4:
                ts.append(gpu.last_render_ms())
        print(f"{kind} {deg}/step, schedule period {period}, motion floor {fl:#x}: {np.median(ts):.4f} ms kernel (median of {a.steps} frames)", flush=True)
