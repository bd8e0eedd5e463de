"""Oracle-side statistics of a bench workload: algorithmic words and bytes per ray (B_ray = 4 W_ray + 16),
the reference algorithm's words per ray (W_restart), steps, hit fraction.  Runs on the CPU (the oracle); no GPU needed.
usage: python tools/ray_stats.py [terrain16_1080p|terrain16_4k]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "terrain16_1080p"
W, H = {"terrain16_1080p": (1920, 1080), "terrain16_4k": (3840, 2160)}[name]
pkg, O = entry.load_package(), entry.load_oracle()
cam, look = pkg.scenes.terrain_camera(0, 16)
words = pkg.scenes.terrain(seed=0, max_depth=16, cam=cam, lod_c=1500.0, max_words=125_000_000)
camm, inv = pkg.camera.camera_matrices(cam, look, 90.0, W, H)
u = O.Uniforms()
u.camera[:] = camm.tolist()
u.camera_inverse[:] = inv.tolist()
u.dimensions[:] = [float(W), float(H), 0.0, 0.0]
u.sun_dir[:] = [-1.7, -1.0, 0.8, 0.0]
u.flags = O.F_PAUSE_ADAPTIVE
rec, st = O.trace_frame(words, u, stats=True, threads=os.cpu_count() or 8)
st = st.reshape(-1, 2).astype(np.float64)
info = rec.reshape(-1)["info"]
print(json.dumps({"workload": name, "rays": W * H, "node_words": int(words.size), "w_ray_words": round(float(st[:, 1].mean()), 4),
                  "algo_bytes_per_ray": round(float(4.0 * st[:, 1].mean() + 16.0), 4), "w_restart_words": round(float(st[:, 0].mean()), 3),
                  "steps_mean": round(float((info & 0xFF).mean()), 3), "hit_frac": round(float(((info >> 16) & 1).mean()), 4),
                  "step_cap_frac": round(float((rec.reshape(-1)["value"] == 0xFF000000).mean()), 5)}))
