"""Counter-scan dispatch: reference src/compute.rs (Compute::{new, update}) and the read-back
half of src/adaptive.rs."""
import ctypes as C

import numpy as np

from ._lib import lib

MAX_SUBDIVISIONS_PER_FRAME = 1024000    # adaptive.rs:3
MAX_UNSUBDIVISIONS_PER_FRAME = 1024000  # adaptive.rs:4


class Compute:
    def __init__(self, gpu, render):
        self.gpu, self.render = gpu, render

    @classmethod
    def new(cls, gpu, render):
        return cls(gpu, render)

    def update(self, octree_or_length):
        """compute.rs:99-127: scan nodes[0 .. octree.nodes.len())"""
        n = octree_or_length if isinstance(octree_or_length, int) else len(octree_or_length)
        self.gpu.check(lib().svo_scan_dispatch(self.gpu._h, n))

    def read_lists(self):
        """adaptive.rs:12-23,76-87: map both lists, clamp the count, reset the counters.
        Returns (subdivide_indices, unsubdivide_indices)."""
        cap = MAX_SUBDIVISIONS_PER_FRAME
        sub = np.empty(cap, dtype=np.uint32)
        unsub = np.empty(cap, dtype=np.uint32)
        ns, nu = C.c_uint32(), C.c_uint32()
        self.gpu.check(lib().svo_scan_read(self.gpu._h, sub.ctypes.data, C.byref(ns), unsub.ctypes.data,
                                           C.byref(nu), cap))
        return sub[1:1 + ns.value].copy(), unsub[1:1 + nu.value].copy()
