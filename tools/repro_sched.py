"""repro: first (unscheduled) vs second (scheduled) frame, and both against the oracle"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as entry
pkg = entry.load_package(); O = entry.load_oracle()
import torch
from conftest import set_uniforms_from_oracle
z = np.load(os.path.join(ROOT, "tests/golden/monu9_vox.npz"))
words = pkg.CpuOctree.from_voxels(int(z["size"][0]), z["xyzi"], z["palette"]).to_octree_words()
gpu = pkg.Gpu(0)
for opts in ({"PAIR_TABLE": 0, "CULL": 0}, {"PAIR_TABLE": 1, "CULL": 0}, {"PAIR_TABLE": 0, "CULL": 2}):
    for k, v in opts.items():
        try:
            gpu.set_option(getattr(pkg.gpu, "OPT_" + k), v)
        except Exception as e:
            print("option", k, "not supported:", e)
    for pose in [((0.1, 0.2, -1.5), (0.0, 0.0, 1.5)), ((1.3, 0.9, 1.2), (-1.0, -0.6, -1.0)), ((0.02, 0.31, 0.05), (0.3, -0.2, 1.0)), ((-1.6, 0.1, 0.2), (1.0, 0.0, 0.0))]:
        u = O.make_uniforms(pos=pose[0], look=pose[1], width=480, height=270, flags=O.F_PAUSE_ADAPTIVE)
        want = O.trace_frame(words, u, threads=8).reshape(-1)
        for rep in range(3):
            render = pkg.Render(gpu, (480, 270), words, capacity=words.size)
            set_uniforms_from_oracle(render, u)
            frames = []
            for f in range(3):
                buf = render.alloc_hits(480 * 270); buf.fill_(-1)
                h = render.render(hits=buf); gpu.sync()
                frames.append(pkg.render.hits_to_numpy(h))
            bad = [int((fr.view(np.uint32).reshape(-1, 4) != want.view(np.uint32).reshape(-1, 4)).any(axis=1).sum()) for fr in frames]
            if any(bad):
                fr = frames[[i for i, b in enumerate(bad) if b][0]]
                idx = np.flatnonzero((fr.view(np.uint32).reshape(-1, 4) != want.view(np.uint32).reshape(-1, 4)).any(axis=1))
                print(opts, pose[0], "rep", rep, "bad rays per frame", bad, "first", idx[:6], "got", fr[idx[:3]], "want", want[idx[:3]])
            else:
                print(opts, pose[0], "rep", rep, "ok")
