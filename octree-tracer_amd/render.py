"""Trace dispatch: reference src/render.rs (Render::{new, update, render, resize}, Uniforms)."""
import ctypes as C

import numpy as np
import torch

from ._lib import Uniforms, lib
from .camera import camera_matrices

F_PAUSE_ADAPTIVE, F_SHOW_STEPS, F_SHOW_HITS, F_SHADOWS, F_MISC_BOOL = 1, 2, 4, 8, 16
HIT_DTYPE = np.dtype([("value", "<u4"), ("t", "<f4"), ("info", "<u4"), ("normal_bits", "<u4")])
DEFAULT_NODE_CAPACITY = 10_000_000  # render.rs:53


class Render:
    """Owns the device node buffer and the uniforms, like the reference's Render.

    Render.new(gpu, size, octree)  render.rs:18-180 (node buffer = octree.expanded(capacity))
    update(settings, character)    render.rs:191-215
    render(...)                    render.rs:217-284: one pass over every pixel
    resize(size)                   render.rs:182-189
    """

    def __init__(self, gpu, size, octree_words, capacity=None):
        self.gpu = gpu
        self.size = (int(size[0]), int(size[1]))
        words = np.ascontiguousarray(octree_words, dtype=np.uint32)
        cap = capacity if capacity is not None else max(DEFAULT_NODE_CAPACITY, words.size)
        cap = max(cap, words.size, 8)
        gpu.check(lib().svo_nodes_alloc(gpu._h, cap))
        self.capacity = cap
        self.node_length = 0
        self.uniforms = Uniforms()
        self.uniforms.sun_dir[:] = [-1.7, -1.0, 0.8, 0.0]  # render.rs:312
        self.uniforms.flags = F_SHADOWS                     # render.rs:313-316 defaults
        self.write_nodes(words)

    @classmethod
    def share_nodes(cls, gpu, other):
        """A second Render on another context (its own stream) over the SAME device node buffer (zero copy):
        svo_nodes_share.  Used to keep more than one frame in flight.  The contexts share the buffer's generation counter:
        a write through either one (write_nodes, scatter_nodes, the adaptive loop) makes every one of them rebuild its top
        table and schedule before its next trace, which also waits on the device for that write.  A write does not wait
        for frames other lanes still have in flight -- sync them first."""
        self = cls.__new__(cls)
        self.gpu, self.size = gpu, other.size
        gpu.check(lib().svo_nodes_share(gpu._h, other.gpu._h))
        self.capacity, self.node_length = other.capacity, other.node_length
        self.uniforms = Uniforms()
        C.memmove(C.byref(self.uniforms), C.byref(other.uniforms), C.sizeof(Uniforms))
        self.upload_uniforms()
        return self

    @classmethod
    def new(cls, gpu, size, octree, capacity=None):
        words = octree.raw_data() if hasattr(octree, "raw_data") else octree
        return cls(gpu, size, words, capacity)

    def write_nodes(self, words, offset=0):
        """queue.write_buffer(&node_buffer, 0, nodes) (app.rs:113-118)"""
        words = np.ascontiguousarray(words, dtype=np.uint32)
        self.gpu.check(lib().svo_nodes_write(self.gpu._h, offset, words.ctypes.data, words.size))
        self.gpu.sync()  # the host array may be released by the caller
        self.node_length = max(self.node_length, offset + words.size)

    def scatter_nodes(self, indices, words, node_length=None):
        """Incremental upload: words[i] -> node buffer [indices[i]] (svo_nodes_scatter)."""
        indices = np.ascontiguousarray(indices, dtype=np.uint32)
        words = np.ascontiguousarray(words, dtype=np.uint32)
        self.gpu.check(lib().svo_nodes_scatter(self.gpu._h, indices.ctypes.data, words.ctypes.data, indices.size))
        self.gpu.sync()  # the host arrays may be released by the caller
        if indices.size:
            self.node_length = max(self.node_length, int(indices.max()) + 1)
        if node_length is not None:
            self.node_length = max(self.node_length, node_length)

    def read_nodes(self, n=None, offset=0):
        n = self.node_length if n is None else n
        out = np.empty(n, dtype=np.uint32)
        self.gpu.check(lib().svo_nodes_read(self.gpu._h, offset, out.ctypes.data, n))
        return out

    def resize(self, new_size):
        if new_size[0] > 0 and new_size[1] > 0:  # render.rs:183
            self.size = (int(new_size[0]), int(new_size[1]))

    def set_flags(self, pause_adaptive=None, show_steps=None, show_hits=None, shadows=None, misc_bool=None):
        f = self.uniforms.flags
        for bit, v in ((F_PAUSE_ADAPTIVE, pause_adaptive), (F_SHOW_STEPS, show_steps), (F_SHOW_HITS, show_hits),
                       (F_SHADOWS, shadows), (F_MISC_BOOL, misc_bool)):
            if v is not None:
                f = (f | bit) if v else (f & ~bit)
        self.uniforms.flags = f

    def update(self, settings, character):
        """render.rs:191-215: camera = proj * look_at_rh; upload the uniforms."""
        w, h = self.size
        # Settings.octree_depth (app.rs:24) is the depth the adaptive loop refines to: a tree may reach it, so it is declared to the
        # library when it exceeds the default kernel's 16 levels (SVO_OPT_TREE_DEPTH; never lowered here -- ADVICE r4: a deeper
        # scene used to fail only at svo_sync)
        depth = int(getattr(settings, "octree_depth", 0) or 0)
        if depth > 16 and depth > getattr(self, "_declared_depth", 16):
            from .gpu import OPT_TREE_DEPTH
            self.gpu.set_option(OPT_TREE_DEPTH, depth)
            self._declared_depth = depth
        cam, inv = camera_matrices(character.pos, character.look, settings.fov, w, h)
        self.uniforms.camera[:] = cam.tolist()
        self.uniforms.camera_inverse[:] = inv.tolist()
        self.uniforms.dimensions[:] = [float(w), float(h), 0.0, 0.0]
        self.upload_uniforms()

    def upload_uniforms(self):
        self.gpu.check(lib().svo_set_uniforms(self.gpu._h, C.byref(self.uniforms)))

    def alloc_hits(self, n_pixels, device=None):
        dev = device if device is not None else f"cuda:{self.gpu.device}"
        return torch.empty((n_pixels, 4), dtype=torch.int32, device=dev)

    def render(self, hits=None, tile=None, rgba=None):
        """Trace every pixel of the frame (or of tile=(x0, y0, w, h)); asynchronous.  Returns the
        device tensor of hit records (n_pixels x 4 int32 = svo_hit)."""
        w, h = self.size
        x0, y0, tw, th = tile if tile is not None else (0, 0, w, h)
        if hits is None:
            hits = self.alloc_hits(tw * th)
        self.gpu.check(lib().svo_render(self.gpu._h, w, h, x0, y0, tw, th, hits.data_ptr(),
                                        rgba.data_ptr() if rgba is not None else None))
        return hits

    def render_tiles(self, tile_w, tile_h, first_tile, tile_stride, hits=None, rgba=None):
        """This rank's tiles (first_tile, first_tile + tile_stride, ...), contiguous in that order."""
        w, h = self.size
        n_tiles_total = (w // tile_w) * (h // tile_h)
        n_mine = max(0, (n_tiles_total - first_tile + tile_stride - 1) // tile_stride)
        if hits is None:
            hits = self.alloc_hits(n_mine * tile_w * tile_h)
        self.gpu.check(lib().svo_render_tiles(self.gpu._h, w, h, tile_w, tile_h, first_tile, tile_stride,
                                              hits.data_ptr(), rgba.data_ptr() if rgba is not None else None))
        return hits

    def render_secondary(self, n_secondary, tile=None, hits=None, secondary=None):
        """Primary rays plus n_secondary rays per hit pixel (svo_render_secondary): returns (primary records,
        secondary records [n_secondary * n_pixels, ray-major]) as device tensors; asynchronous."""
        w, h = self.size
        x0, y0, tw, th = tile if tile is not None else (0, 0, w, h)
        if hits is None:
            hits = self.alloc_hits(tw * th)
        if secondary is None:
            secondary = self.alloc_hits(n_secondary * tw * th)
        self.gpu.check(lib().svo_render_secondary(self.gpu._h, w, h, x0, y0, tw, th, n_secondary, hits.data_ptr(),
                                                  secondary.data_ptr()))
        return hits, secondary

    def render_tiles_secondary(self, tile_w, tile_h, first_tile, tile_stride, n_secondary, hits=None, secondary=None):
        """render_tiles with secondary rays: this rank's tiles, contiguous; secondary ray-major over them."""
        w, h = self.size
        n_tiles_total = (w // tile_w) * (h // tile_h)
        n_mine = max(0, (n_tiles_total - first_tile + tile_stride - 1) // tile_stride)
        if hits is None:
            hits = self.alloc_hits(n_mine * tile_w * tile_h)
        if secondary is None:
            secondary = self.alloc_hits(n_secondary * n_mine * tile_w * tile_h)
        self.gpu.check(lib().svo_render_tiles_secondary(self.gpu._h, w, h, tile_w, tile_h, first_tile, tile_stride,
                                                        n_secondary, hits.data_ptr(), secondary.data_ptr()))
        return hits, secondary

    def assemble_tiles(self, gathered, tile_w, tile_h, out=None):
        """Rank 0: un-permute a gathered frame ([world, n_pad, tile_h * tile_w, 4] int32, device; last dimension 3 for wire
        records) into the row-major frame
        ([H, W, 4]) with one kernel on this context's stream (svo_assemble_tiles)."""
        w, h = self.size
        world, n_pad = int(gathered.shape[0]), int(gathered.shape[1])
        if out is None:
            out = torch.empty((h, w, 4), dtype=torch.int32, device=gathered.device)
        fn = lib().svo_assemble_tiles_packed if gathered.shape[-1] == 3 else lib().svo_assemble_tiles
        self.gpu.check(fn(self.gpu._h, gathered.data_ptr(), world, n_pad, w, h, tile_w, tile_h, out.data_ptr()))
        return out

    def assemble_tiles_rgba(self, gathered, tile_w, tile_h, out=None):
        """Rank 0: un-permute a gathered COLOUR frame ([world, n_pad, tile_h * tile_w] int32 RGBA8, device) into [H, W]
        (svo_assemble_tiles_rgba)."""
        w, h = self.size
        world, n_pad = int(gathered.shape[0]), int(gathered.shape[1])
        if out is None:
            out = torch.empty((h, w), dtype=torch.int32, device=gathered.device)
        self.gpu.check(lib().svo_assemble_tiles_rgba(self.gpu._h, gathered.data_ptr(), world, n_pad, w, h, tile_w, tile_h, out.data_ptr()))
        return out

    def pack_records(self, records, wire):
        """[..., 4] int32 hit records -> [..., 3] int32 wire records (svo_pack_records), on this context's stream."""
        n = records.numel() // 4
        self.gpu.check(lib().svo_pack_records(self.gpu._h, records.data_ptr(), n, wire.data_ptr()))
        return wire

    def render_host(self, tile=None, rgba=False):
        """Blocking variant with host results: hit records (numpy structured array, tile-local row-major)
        and, with rgba=True, also the shaded RGBA8 image of fs_main (uint8 [h, w, 4])."""
        w, h = self.size
        x0, y0, tw, th = tile if tile is not None else (0, 0, w, h)
        out = np.empty((th, tw), dtype=HIT_DTYPE)
        img = np.empty((th, tw), dtype=np.uint32) if rgba else None
        self.gpu.check(lib().svo_render_host(self.gpu._h, w, h, x0, y0, tw, th, out.ctypes.data,
                                             img.ctypes.data if rgba else None))
        return (out, img.view(np.uint8).reshape(th, tw, 4)) if rgba else out

    def alloc_rgba(self, n_pixels, device=None):
        dev = device if device is not None else f"cuda:{self.gpu.device}"
        return torch.empty((n_pixels,), dtype=torch.int32, device=dev)

    def trace_rays(self, rays, hits=None):
        """octree_ray over explicit rays (device float32 tensor n x 6)."""
        n = rays.shape[0]
        if hits is None:
            hits = self.alloc_hits(n, rays.device)
        self.gpu.check(lib().svo_trace_rays(self.gpu._h, rays.data_ptr(), n, hits.data_ptr()))
        return hits


def hits_to_numpy(hits):
    """device hit tensor -> numpy structured array (value, t, info, normal_bits)"""
    return hits.cpu().numpy().view(np.uint32).reshape(-1, 4).copy().view(HIT_DTYPE).reshape(-1)
