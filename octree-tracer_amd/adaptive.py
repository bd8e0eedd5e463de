"""View-dependent LOD streaming on top of the device scan: reference src/adaptive.rs
(process_subdivision :6-68, process_unsubdivision :70-126) and the frame update of src/app.rs:94-118.
The list processing itself is native (svo_adaptive_subdivide / svo_adaptive_unsubdivide)."""
import numpy as np

from ._lib import lib
from .gpu import OPT_SCAN_CLEARS_COUNTERS
from .world import World as _World


def World(chunk0=None, path=""):
    """A world around one root chunk (App::new, app.rs:33-36): insert as chunk 0 and build its mip colours."""
    w = _World(path)
    if chunk0 is not None:
        w.insert(0, chunk0)
        w.top_mip = w.generate_mip_tree(0)
    return w


def _list(nodes):
    return np.ascontiguousarray(nodes, dtype=np.uint32)


def process_subdivision(compute_lists, octree, world):
    """adaptive.rs:29-61 over the subdivide list: hot leaves get their 8 children from the CPU world (the
    root group of the referenced chunk at a block leaf).  Returns the number of subdivisions."""
    nodes = _list(compute_lists)
    done = lib().svo_adaptive_subdivide(world._h, octree._h, nodes.ctypes.data, nodes.size, None)
    return world._check(done)


def process_unsubdivision(compute_lists, octree, world):
    """adaptive.rs:93-121 over the unsubdivide list: cold interior nodes collapse to their mip colour."""
    nodes = _list(compute_lists)
    done = lib().svo_adaptive_unsubdivide(world._h, octree._h, nodes.ctypes.data, nodes.size)
    return world._check(done)


class AdaptiveLoop:
    """App::update (app.rs:94-118): uniforms -> trace (counters live) -> scan -> CPU (un)subdivide -> re-upload.

    incremental=True replaces the reference's re-upload of the WHOLE array (which is also what resets the hit
    counters: host words carry counter 0) by its device-side equivalent: the scan zeroes the counters it has read
    (SVO_OPT_SCAN_CLEARS_COUNTERS) and only the words the list processing changed are sent (svo_nodes_scatter).
    The device array after a frame is the same either way."""

    def __init__(self, gpu, render, compute, octree, world, incremental=False):
        self.gpu, self.render, self.compute, self.octree, self.world = gpu, render, compute, octree, world
        self.incremental = incremental
        gpu.set_option(OPT_SCAN_CLEARS_COUNTERS, 1 if incremental else 0)
        octree.take_dirty()  # the device already holds the octree as it is now

    def frame(self, settings, character, deterministic=False):
        self.render.update(settings, character)
        hits = self.render.render()
        if self.render.uniforms.flags & 1:  # pause_adaptive (app.rs:97)
            return hits, 0, 0
        self.compute.update(len(self.octree))
        sub, unsub = self.compute.read_lists()
        if deterministic:  # the device appends in no particular order; sorted lists make runs repeatable
            sub.sort()
            unsub.sort()
        n_sub = process_subdivision(sub, self.octree, self.world)
        n_unsub = process_unsubdivision(unsub, self.octree, self.world)
        if self.incremental:
            idx, val = self.octree.take_dirty()
            self.render.scatter_nodes(idx, val, node_length=len(self.octree))
        else:
            # app.rs:113-118: the whole array goes back (host words carry counter 0, which also clears the counters)
            self.render.write_nodes(self.octree.raw_data())
        return hits, n_sub, n_unsub
