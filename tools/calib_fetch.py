"""Calibrates rocprofv3's FETCH_SIZE for the trace kernels' access shape (single-dword gathers).
Run under:  rocprofv3 --pmc FETCH_SIZE --output-format csv -d <dir> -- python3 tools/calib_fetch.py
Two launches of diag_gather_kernel over a 1 GiB buffer (4x the Infinity Cache): 4 Mi loads at a 128-byte
stride (4 Mi distinct 128-B lines) and 4 Mi loads at a 64-byte stride (2 Mi distinct 128-B lines, 4 Mi
distinct 64-B sectors).  Expected FETCH_SIZE [KB] if it counts 64 B per fetched sector: 262144 both;
if it counts 64 B per 128-B line: 262144 and 131072."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
pkg = entry.load_package()
gpu = pkg.Gpu(0)
words = np.zeros(8, dtype=np.uint32)
render = pkg.Render(gpu, (8, 8), words, capacity=1 << 28 >> 1)  # 2^27 words = 512 MiB (layout cap)
n = 1 << 21
for stride in (128, 64, 128, 64):
    gpu.check(pkg._lib.lib().svo_diag_gather(gpu._h, stride, n if stride == 128 else n))
    gpu.sync()
print("done: loads per launch", n, "buffer MiB", (1 << 27) * 4 >> 20)
