"""octree-tracer_amd: MI355X-native drop-in for the GPU path of ria8651/octree-tracer.

Host-side mirror of the reference's dispatch API (same names and argument meaning) over the
C ABI of libsvo_hip.so:
    gpu.Gpu            <- src/gpu.rs      Gpu
    render.Render      <- src/render.rs   Render {new, update, render, resize}, Uniforms
    compute.Compute    <- src/compute.rs  Compute {new, update}
    octree.Octree      <- src/octree.rs   Octree, Voxel, VOXEL_OFFSET
    cpu_octree.CpuOctree <- src/cpu_octree.rs CpuOctree
    camera.{Character, Settings} <- src/main.rs
    world.World        <- src/world.rs    World {chunks, find_voxel, generate_mip_tree, save/load_chunk, load_world}
    adaptive           <- src/adaptive.rs  process_subdivision / process_unsubdivision
    scenes             deterministic benchmark scene generators (no reference counterpart)
The package directory name contains a hyphen; import it through __graft_entry__.load_package(),
which registers it as module `octree_tracer_amd`.
"""
from . import _lib
from ._lib import SvoError, Uniforms
from .octree import VOXEL_OFFSET, Octree, Voxel, create_node
from .cpu_octree import CHUNK_OFFSET, CpuOctree
from .camera import Character, Settings, camera_matrices
from .gpu import Gpu
from .render import Render, HIT_DTYPE, F_PAUSE_ADAPTIVE, F_SHOW_STEPS, F_SHOW_HITS, F_SHADOWS, F_MISC_BOOL
from .compute import Compute
from .world import World
from . import scenes
from . import sharding
from . import adaptive

__all__ = ["Gpu", "Render", "Compute", "Octree", "CpuOctree", "Voxel", "World", "Uniforms", "Character", "Settings",
           "SvoError", "VOXEL_OFFSET", "CHUNK_OFFSET", "HIT_DTYPE", "create_node", "camera_matrices", "scenes", "sharding", "adaptive"]
