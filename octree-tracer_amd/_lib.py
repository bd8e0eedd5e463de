"""ctypes loader for libsvo_hip.so (the C-ABI drop-in, include/svo_hip.h + include/svo_host.h).

There is no CPU fallback: if the library is missing this module raises, and every device
call goes through the HIP kernels in csrc/.
"""
import ctypes as C
import os

# torch first: it ships its own libamdhip64.so.7; loading it before our library makes both share
# one HIP runtime (same SONAME), so torch tensors/streams and our kernels live in one context.
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SVO_HIP_LIB") or os.path.join(_HERE, "libsvo_hip.so")  # override: A/B builds in experiments


class Uniforms(C.Structure):
    """svo_uniforms: the reference's Uniforms (render.rs:287-322) with explicit u32 flags."""
    _fields_ = [
        ("camera", C.c_float * 16),
        ("camera_inverse", C.c_float * 16),
        ("dimensions", C.c_float * 4),
        ("sun_dir", C.c_float * 4),
        ("flags", C.c_uint32),
        ("misc_value", C.c_float),
    ]


class TerrainParams(C.Structure):
    _fields_ = [
        ("seed", C.c_uint32),
        ("max_depth", C.c_uint32),
        ("cam", C.c_float * 3),
        ("lod_c", C.c_float),
        ("min_depth", C.c_uint32),
        ("max_words", C.c_uint64),
    ]


DEVICE_SYMBOLS = [
    "svo_device_count", "svo_buffer_alloc", "svo_buffer_free", "svo_buffer_read", "svo_ctx_create", "svo_ctx_destroy", "svo_ctx_set_stream", "svo_set_option", "svo_last_error", "svo_sync",
    "svo_nodes_alloc", "svo_nodes_bind_device", "svo_nodes_write", "svo_nodes_scatter", "svo_nodes_read", "svo_nodes_device_ptr", "svo_nodes_share", "svo_nodes_invalidate",
    "svo_comm_unique_id", "svo_comm_init_rank", "svo_comm_init_all", "svo_comm_destroy", "svo_gather_frame", "svo_gather_frame_all", "svo_gather_wait",
    "svo_set_uniforms", "svo_render", "svo_render_host", "svo_render_tiles", "svo_render_secondary", "svo_render_tiles_secondary", "svo_assemble_tiles", "svo_assemble_tiles_packed", "svo_assemble_tiles_rgba", "svo_pack_records", "svo_trace_rays",
    "svo_last_render_ms", "svo_timing_collect", "svo_diag_gather", "svo_diag_strip_classes", "svo_scan_dispatch", "svo_scan_read",
]
HOST_SYMBOLS = [
    "svo_cpu_octree_new", "svo_cpu_octree_free", "svo_cpu_octree_len", "svo_cpu_octree_load_file",
    "svo_cpu_octree_load_vox", "svo_cpu_octree_load_rsvo", "svo_cpu_octree_from_voxels",
    "svo_cpu_octree_put_in_voxel", "svo_cpu_octree_put_in_block", "svo_cpu_octree_find_voxel",
    "svo_cpu_octree_get_node_mask", "svo_cpu_octree_to_octree", "svo_cpu_octree_raw",
    "svo_cpu_octree_generate_mips", "svo_vox_parse", "svo_vox_write", "svo_rsvo_write",
    "svo_octree_new", "svo_octree_from_words", "svo_octree_free", "svo_octree_len", "svo_octree_raw_data",
    "svo_octree_get_node", "svo_octree_subdivide", "svo_octree_unsubdivide", "svo_octree_find_voxel",
    "svo_octree_expanded", "svo_octree_pos_offset", "svo_octree_holes", "svo_octree_set_node", "svo_octree_position", "svo_octree_take_dirty", "svo_camera_matrices",
    "svo_world_new", "svo_world_free", "svo_world_last_error", "svo_world_insert", "svo_world_remove",
    "svo_world_chunk", "svo_world_chunk_ids", "svo_world_find_voxel", "svo_world_generate_mip_tree",
    "svo_world_save_chunk", "svo_world_load_chunk", "svo_world_load", "svo_cpu_octree_bin", "svo_cpu_octree_from_bin",
    "svo_adaptive_subdivide", "svo_adaptive_unsubdivide", "svo_world_expand",
    "svo_gen_terrain", "svo_gen_terrain_height", "svo_gen_fractal", "svo_gen_random", "svo_nodes_max_depth", "svo_nodes_relayout",
]

_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `make -C octree-tracer_amd/csrc` "
            "(or __graft_entry__.build()); there is no CPU fallback")
    L = C.CDLL(LIB_PATH)
    vp, sz, u32, u64, i64, f32 = C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint64, C.c_int64, C.c_float
    cp = C.c_char_p
    fp = C.POINTER(C.c_float)

    def sig(name, res, *args):
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = list(args)

    # device boundary (include/svo_hip.h)
    sig("svo_device_count", C.c_int)
    sig("svo_buffer_alloc", C.c_int, vp, sz, C.POINTER(vp))
    sig("svo_buffer_free", C.c_int, vp, vp)
    sig("svo_buffer_read", C.c_int, vp, vp, vp, sz)
    sig("svo_ctx_create", C.c_int, C.c_int, C.POINTER(vp))
    sig("svo_ctx_destroy", C.c_int, vp)
    sig("svo_ctx_set_stream", C.c_int, vp, vp, C.c_int)
    sig("svo_set_option", C.c_int, vp, C.c_int, i64)
    sig("svo_last_error", cp, vp)
    sig("svo_sync", C.c_int, vp)
    sig("svo_nodes_alloc", C.c_int, vp, sz)
    sig("svo_nodes_bind_device", C.c_int, vp, vp, sz)
    sig("svo_nodes_write", C.c_int, vp, sz, vp, sz)
    sig("svo_nodes_scatter", C.c_int, vp, vp, vp, sz)
    sig("svo_nodes_read", C.c_int, vp, sz, vp, sz)
    sig("svo_nodes_device_ptr", C.c_int, vp, C.POINTER(vp), C.POINTER(sz))
    sig("svo_nodes_share", C.c_int, vp, vp)
    sig("svo_nodes_invalidate", C.c_int, vp)
    sig("svo_comm_unique_id", C.c_int, vp)
    sig("svo_comm_init_rank", C.c_int, vp, vp, C.c_int, C.c_int)
    sig("svo_comm_init_all", C.c_int, C.c_int, C.POINTER(vp))
    sig("svo_comm_destroy", C.c_int, vp)
    sig("svo_gather_frame", C.c_int, vp, vp, sz, vp, C.c_int)
    sig("svo_gather_wait", C.c_int, vp)
    sig("svo_gather_frame_all", C.c_int, C.c_int, C.POINTER(vp), C.POINTER(vp), sz, vp, C.c_int)
    sig("svo_set_uniforms", C.c_int, vp, C.POINTER(Uniforms))
    sig("svo_render", C.c_int, vp, u32, u32, u32, u32, u32, u32, vp, vp)
    sig("svo_render_host", C.c_int, vp, u32, u32, u32, u32, u32, u32, vp, vp)
    sig("svo_render_tiles", C.c_int, vp, u32, u32, u32, u32, u32, u32, vp, vp)
    sig("svo_render_secondary", C.c_int, vp, u32, u32, u32, u32, u32, u32, u32, vp, vp)
    sig("svo_render_tiles_secondary", C.c_int, vp, u32, u32, u32, u32, u32, u32, u32, vp, vp)
    sig("svo_assemble_tiles", C.c_int, vp, vp, u32, u32, u32, u32, u32, u32, vp)
    sig("svo_assemble_tiles_packed", C.c_int, vp, vp, u32, u32, u32, u32, u32, u32, vp)
    sig("svo_assemble_tiles_rgba", C.c_int, vp, vp, u32, u32, u32, u32, u32, u32, vp)
    sig("svo_pack_records", C.c_int, vp, vp, sz, vp)
    sig("svo_trace_rays", C.c_int, vp, vp, sz, vp)
    sig("svo_last_render_ms", C.c_int, vp, fp)
    sig("svo_timing_collect", C.c_int, vp, fp, sz, C.POINTER(sz))
    sig("svo_diag_gather", C.c_int, vp, u32, u32)
    sig("svo_diag_strip_classes", C.c_int, vp, vp, sz)
    sig("svo_scan_dispatch", C.c_int, vp, u32)
    sig("svo_scan_read", C.c_int, vp, vp, C.POINTER(u32), vp, C.POINTER(u32), sz)
    # host data model (include/svo_host.h)
    sig("svo_cpu_octree_new", vp, C.c_uint8)
    sig("svo_cpu_octree_free", None, vp)
    sig("svo_cpu_octree_len", sz, vp)
    sig("svo_cpu_octree_load_file", vp, cp, u32, cp, sz)
    sig("svo_cpu_octree_load_vox", vp, cp, sz, cp, sz)
    sig("svo_cpu_octree_load_rsvo", vp, cp, sz, u32, cp, sz)
    sig("svo_cpu_octree_from_voxels", vp, u32, vp, sz, vp, cp, sz)
    sig("svo_cpu_octree_put_in_voxel", None, vp, fp, C.POINTER(C.c_uint8), u32)
    sig("svo_cpu_octree_put_in_block", None, vp, fp, u32, u32)
    sig("svo_cpu_octree_find_voxel", None, vp, fp, i64, C.POINTER(u64), C.POINTER(u32), fp)
    sig("svo_cpu_octree_get_node_mask", None, vp, sz, C.POINTER(C.c_uint8))
    sig("svo_cpu_octree_to_octree", None, vp, vp)
    sig("svo_cpu_octree_raw", None, vp, vp, vp)
    sig("svo_cpu_octree_generate_mips", None, vp, C.POINTER(C.c_uint8))
    sig("svo_vox_parse", i64, cp, sz, C.POINTER(u32), vp, sz, vp, cp, sz)
    sig("svo_vox_write", sz, u32, vp, sz, vp, vp, sz)
    sig("svo_rsvo_write", sz, vp, vp, sz)
    sig("svo_octree_new", vp, C.POINTER(C.c_uint8))
    sig("svo_octree_from_words", vp, vp, sz)
    sig("svo_octree_free", None, vp)
    sig("svo_octree_len", sz, vp)
    sig("svo_octree_raw_data", vp, vp)
    sig("svo_octree_get_node", u32, vp, sz)
    sig("svo_octree_subdivide", C.c_int, vp, sz, C.POINTER(C.c_uint8), u32)
    sig("svo_octree_unsubdivide", C.c_int, vp, sz)
    sig("svo_octree_find_voxel", None, vp, fp, i64, C.POINTER(u64), C.POINTER(u32), fp)
    sig("svo_octree_expanded", C.c_int, vp, sz, vp)
    sig("svo_octree_pos_offset", None, u32, u32, fp)
    sig("svo_octree_holes", sz, vp)
    sig("svo_octree_set_node", None, vp, sz, u32)
    sig("svo_octree_position", None, vp, sz, fp)
    sig("svo_octree_take_dirty", sz, vp, vp, vp, sz)
    sig("svo_world_new", vp, cp)
    sig("svo_world_free", None, vp)
    sig("svo_world_last_error", cp, vp)
    sig("svo_world_insert", C.c_int, vp, u32, vp)
    sig("svo_world_remove", C.c_int, vp, u32)
    sig("svo_world_chunk", vp, vp, u32)
    sig("svo_world_chunk_ids", sz, vp, vp, sz)
    sig("svo_world_find_voxel", C.c_int, vp, fp, i64, C.POINTER(u32), C.POINTER(u64), C.POINTER(u32), fp)
    sig("svo_world_generate_mip_tree", C.c_int, vp, u32, C.POINTER(C.c_uint8))
    sig("svo_world_save_chunk", C.c_int, vp, u32)
    sig("svo_world_load_chunk", C.c_int, vp, u32)
    sig("svo_world_load", vp, cp, cp, sz)
    sig("svo_cpu_octree_bin", sz, vp, vp, sz)
    sig("svo_cpu_octree_from_bin", vp, cp, sz, cp, sz)
    sig("svo_adaptive_subdivide", i64, vp, vp, vp, sz, C.POINTER(u64))
    sig("svo_adaptive_unsubdivide", i64, vp, vp, vp, sz)
    sig("svo_world_expand", u64, vp, vp, u32, fp, f32, u64)
    sig("svo_camera_matrices", None, fp, fp, f32, f32, f32, fp, fp)
    sig("svo_gen_terrain", u64, C.POINTER(TerrainParams), vp, u64)
    sig("svo_gen_terrain_height", C.c_int32, u32, u32, u32, u32)
    sig("svo_gen_fractal", u64, C.POINTER(TerrainParams), vp, u64)
    sig("svo_gen_random", u64, u32, u32, f32, f32, u64, vp, u64)
    sig("svo_nodes_max_depth", u32, vp, u64)
    sig("svo_nodes_relayout", u64, vp, u64, u32, vp, vp)
    _lib = L
    return L


class SvoError(RuntimeError):
    pass
