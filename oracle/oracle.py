"""ctypes binding of the CPU oracle (oracle/libsvo_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package never imports this module.
Parity status: unpinned by the reference (see svo_oracle.h).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libsvo_oracle.so")

VOXEL_OFFSET = 134217728
CHUNK_OFFSET = 2147483648
F_PAUSE_ADAPTIVE, F_SHOW_STEPS, F_SHOW_HITS, F_SHADOWS, F_MISC_BOOL = 1, 2, 4, 8, 16

HIT_DTYPE = np.dtype([("value", "<u4"), ("t", "<f4"), ("info", "<u4"), ("normal_bits", "<u4")])


class Uniforms(C.Structure):
    _fields_ = [
        ("camera", C.c_float * 16),
        ("camera_inverse", C.c_float * 16),
        ("dimensions", C.c_float * 4),
        ("sun_dir", C.c_float * 4),
        ("flags", C.c_uint32),
        ("misc_value", C.c_float),
    ]


def build(force=False):
    """Compile the oracle with gcc (strict IEEE flags live in oracle/Makefile).  SVO_ORACLE_LIB names a prebuilt
    library instead (the sanitizer build of tools/sanitize_cpu.sh)."""
    if os.environ.get("SVO_ORACLE_LIB"):
        return os.environ["SVO_ORACLE_LIB"]
    src = os.path.join(_HERE, "svo_oracle.c")
    if (not force and os.path.exists(_LIB_PATH)
            and os.path.getmtime(_LIB_PATH) >= max(os.path.getmtime(src),
                                                   os.path.getmtime(os.path.join(_HERE, "svo_oracle.h")))):
        return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-B", "libsvo_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    L = C.CDLL(build())
    vp, u8p, u32p, fp = C.c_void_p, C.POINTER(C.c_uint8), C.POINTER(C.c_uint32), C.POINTER(C.c_float)
    L.oracle_tree_new.restype = vp
    L.oracle_tree_new.argtypes = [C.c_uint8]
    L.oracle_tree_free.argtypes = [vp]
    L.oracle_tree_len.restype = C.c_size_t
    L.oracle_tree_len.argtypes = [vp]
    L.oracle_tree_put_in_voxel.argtypes = [vp, C.c_float, C.c_float, C.c_float, C.c_uint8, C.c_uint8, C.c_uint8, C.c_uint32]
    L.oracle_tree_find_voxel.argtypes = [vp, C.c_float, C.c_float, C.c_float, C.c_int64, C.POINTER(C.c_uint64), u32p, fp]
    L.oracle_tree_from_vox.restype = vp
    L.oracle_tree_from_vox.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t]
    L.oracle_tree_from_voxels.restype = vp
    L.oracle_tree_from_voxels.argtypes = [C.c_uint32, vp, C.c_size_t, vp]
    L.oracle_tree_from_rsvo.restype = vp
    L.oracle_tree_from_rsvo.argtypes = [C.c_char_p, C.c_size_t, C.c_uint32, C.c_char_p, C.c_size_t]
    L.oracle_tree_to_octree.argtypes = [vp, vp]
    L.oracle_tree_raw.argtypes = [vp, vp, vp]
    L.oracle_tree_generate_mips.argtypes = [vp, u8p]
    L.oracle_vox_parse.restype = C.c_int64
    L.oracle_vox_parse.argtypes = [C.c_char_p, C.c_size_t, u32p, vp, C.c_size_t, vp, C.c_char_p, C.c_size_t]
    L.oracle_camera.argtypes = [fp, fp, C.c_float, C.c_float, C.c_float, fp, fp]
    L.oracle_find_voxel.argtypes = [vp, C.c_size_t, fp, C.c_int, u32p, fp, u32p]
    L.oracle_trace_rays.argtypes = [vp, C.c_size_t, C.c_uint32, vp, C.c_size_t, vp, vp, C.c_int]
    L.oracle_trace_frame.argtypes = [vp, C.c_size_t, C.POINTER(Uniforms), C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, C.c_int]
    L.oracle_shade_frame.argtypes = [vp, C.c_size_t, C.POINTER(Uniforms), C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int]
    L.oracle_secondary_frame.argtypes = [vp, C.c_size_t, C.POINTER(Uniforms), C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint32,
                                         vp, vp, C.c_int]
    L.oracle_count_frame.argtypes = [vp, C.c_size_t, C.POINTER(Uniforms), C.c_int, C.c_int, C.c_int, C.c_int]
    L.oracle_scan.argtypes = [vp, C.c_size_t, C.c_uint32, vp, vp, C.c_size_t]
    _lib = L
    return L


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class Tree:
    """CpuOctree (src/cpu_octree.rs)."""

    def __init__(self, handle):
        if not handle:
            raise ValueError("oracle tree construction failed")
        self._h = handle

    def __del__(self):
        if getattr(self, "_h", None):
            lib().oracle_tree_free(self._h)
            self._h = None

    @classmethod
    def new(cls, mask=0):
        return cls(lib().oracle_tree_new(mask))

    @classmethod
    def from_vox(cls, data: bytes):
        err = C.create_string_buffer(256)
        h = lib().oracle_tree_from_vox(data, len(data), err, 256)
        if not h:
            raise ValueError(err.value.decode())
        return cls(h)

    @classmethod
    def from_voxels(cls, size, xyzi, palette):
        xyzi = np.ascontiguousarray(xyzi, dtype=np.uint8)
        palette = np.ascontiguousarray(palette, dtype=np.uint32)
        assert palette.size == 256 and xyzi.ndim == 2 and xyzi.shape[1] == 4
        h = lib().oracle_tree_from_voxels(size, _ptr(xyzi), xyzi.shape[0], _ptr(palette))
        if not h:
            raise ValueError("Voxel model size is not a power of 2!")
        return cls(h)

    @classmethod
    def from_rsvo(cls, data: bytes, octree_depth: int):
        err = C.create_string_buffer(256)
        h = lib().oracle_tree_from_rsvo(data, len(data), octree_depth, err, 256)
        if not h:
            raise ValueError(err.value.decode())
        return cls(h)

    def __len__(self):
        return lib().oracle_tree_len(self._h)

    def put_in_voxel(self, pos, rgb, depth):
        lib().oracle_tree_put_in_voxel(self._h, pos[0], pos[1], pos[2], rgb[0], rgb[1], rgb[2], depth)

    def find_voxel(self, pos, max_depth=None):
        idx, d = C.c_uint64(), C.c_uint32()
        p = (C.c_float * 3)()
        lib().oracle_tree_find_voxel(self._h, pos[0], pos[1], pos[2], -1 if max_depth is None else max_depth,
                                     C.byref(idx), C.byref(d), p)
        return idx.value, d.value, tuple(p)

    def to_octree(self):
        out = np.empty(len(self), dtype=np.uint32)
        lib().oracle_tree_to_octree(self._h, _ptr(out))
        return out

    def raw(self):
        ptrs = np.empty(len(self), dtype=np.uint32)
        rgb = np.empty((len(self), 3), dtype=np.uint8)
        lib().oracle_tree_raw(self._h, _ptr(ptrs), _ptr(rgb))
        return ptrs, rgb

    def generate_mips(self):
        top = (C.c_uint8 * 3)()
        lib().oracle_tree_generate_mips(self._h, top)
        return tuple(top)


def vox_parse(data: bytes):
    size = (C.c_uint32 * 3)()
    err = C.create_string_buffer(256)
    n = lib().oracle_vox_parse(data, len(data), size, None, 0, None, err, 256)
    if n < 0:
        raise ValueError(err.value.decode())
    xyzi = np.empty((n, 4), dtype=np.uint8)
    pal = np.empty(256, dtype=np.uint32)
    lib().oracle_vox_parse(data, len(data), size, _ptr(xyzi), xyzi.nbytes, _ptr(pal), err, 256)
    return tuple(size), xyzi, pal


def camera(pos, look, fov_deg, width, height):
    p = (C.c_float * 3)(*pos)
    l = (C.c_float * 3)(*look)
    cam, inv = (C.c_float * 16)(), (C.c_float * 16)()
    lib().oracle_camera(p, l, fov_deg, width, height, cam, inv)
    return np.array(cam, dtype=np.float32), np.array(inv, dtype=np.float32)


def make_uniforms(pos=(0.1, 0.2, -1.5), look=(0.0, 0.0, 1.5), fov=90.0, width=256, height=256,
                  flags=F_PAUSE_ADAPTIVE, sun_dir=(-1.7, -1.0, 0.8, 0.0), misc_value=0.0):
    """Uniforms as Render::update builds them (render.rs:191-206); default pose main.rs:128-137."""
    u = Uniforms()
    cam, inv = camera(pos, look, fov, float(width), float(height))
    u.camera[:] = cam.tolist()
    u.camera_inverse[:] = inv.tolist()
    u.dimensions[:] = [float(width), float(height), 0.0, 0.0]
    u.sun_dir[:] = list(sun_dir)
    u.flags = flags
    u.misc_value = misc_value
    return u


def find_voxel(nodes, pos, misc_bool=False):
    nodes = np.ascontiguousarray(nodes, dtype=np.uint32)
    p = (C.c_float * 3)(*pos)
    value, depth = C.c_uint32(), C.c_uint32()
    vpos = (C.c_float * 3)()
    lib().oracle_find_voxel(_ptr(nodes), nodes.size, p, int(misc_bool), C.byref(value), vpos, C.byref(depth))
    return value.value, tuple(vpos), depth.value


def trace_rays(nodes, rays, flags=F_PAUSE_ADAPTIVE, stats=False, threads=1):
    nodes = np.ascontiguousarray(nodes, dtype=np.uint32)
    rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 6)
    out = np.empty(rays.shape[0], dtype=HIT_DTYPE)
    st = np.empty((rays.shape[0], 2), dtype=np.uint32) if stats else None
    lib().oracle_trace_rays(_ptr(nodes), nodes.size, flags, _ptr(rays), rays.shape[0], _ptr(out), _ptr(st), threads)
    return (out, st) if stats else out


def trace_frame(nodes, u, tile=None, stats=False, threads=1):
    nodes = np.ascontiguousarray(nodes, dtype=np.uint32)
    W, H = int(u.dimensions[0]), int(u.dimensions[1])
    x0, y0, w, h = tile if tile else (0, 0, W, H)
    out = np.empty((h, w), dtype=HIT_DTYPE)
    st = np.empty((h, w, 2), dtype=np.uint32) if stats else None
    lib().oracle_trace_frame(_ptr(nodes), nodes.size, C.byref(u), x0, y0, w, h, _ptr(out), _ptr(st), threads)
    return (out, st) if stats else out


def shade_frame(nodes, u, tile=None, threads=1):
    nodes = np.ascontiguousarray(nodes, dtype=np.uint32)
    W, H = int(u.dimensions[0]), int(u.dimensions[1])
    x0, y0, w, h = tile if tile else (0, 0, W, H)
    rgba = np.empty((h, w, 4), dtype=np.float32)
    lib().oracle_shade_frame(_ptr(nodes), nodes.size, C.byref(u), x0, y0, w, h, _ptr(rgba), threads)
    return rgba


def secondary_frame(nodes, u, n_secondary, tile=None, threads=1):
    """Primary records [h, w] and secondary records [n_secondary, h, w] (svo_render_secondary's definition)."""
    nodes = np.ascontiguousarray(nodes, dtype=np.uint32)
    W, H = int(u.dimensions[0]), int(u.dimensions[1])
    x0, y0, w, h = tile if tile else (0, 0, W, H)
    prim = np.empty((h, w), dtype=HIT_DTYPE)
    sec = np.empty((n_secondary, h, w), dtype=HIT_DTYPE)
    lib().oracle_secondary_frame(_ptr(nodes), nodes.size, C.byref(u), x0, y0, w, h, n_secondary, _ptr(prim), _ptr(sec), threads)
    return prim, sec


def count_frame(nodes, u, tile=None):
    """Returns a copy of nodes with the hit-counter side effect of one frame applied."""
    nodes = np.array(nodes, dtype=np.uint32, copy=True)
    W, H = int(u.dimensions[0]), int(u.dimensions[1])
    x0, y0, w, h = tile if tile else (0, 0, W, H)
    lib().oracle_count_frame(_ptr(nodes), nodes.size, C.byref(u), x0, y0, w, h)
    return nodes


def scan(nodes, node_length=None, capacity=1024000):
    nodes = np.ascontiguousarray(nodes, dtype=np.uint32)
    sub = np.zeros(capacity, dtype=np.uint32)
    unsub = np.zeros(capacity, dtype=np.uint32)
    lib().oracle_scan(_ptr(nodes), nodes.size, nodes.size if node_length is None else node_length,
                      _ptr(sub), _ptr(unsub), capacity)
    return sub, unsub


def unpack_info(info):
    info = np.asarray(info)
    return {"steps": info & 0xFF, "depth": (info >> 8) & 0xFF, "hit": (info >> 16) & 1, "normal": (info >> 17) & 0x3F}
