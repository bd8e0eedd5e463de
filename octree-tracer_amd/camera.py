"""Camera state and matrices: reference src/main.rs:116-162 (Settings, Character,
create_proj_matrix) and src/render.rs:191-206."""
import ctypes as C

import numpy as np

from ._lib import lib


class Settings:
    """main.rs:116-120; defaults app.rs:23-27"""

    def __init__(self, octree_depth=12, fov=90.0, sensitivity=0.00005):
        self.octree_depth, self.fov, self.sensitivity = octree_depth, fov, sensitivity


class Character:
    """main.rs:122-137"""

    def __init__(self, pos=(0.1, 0.2, -1.5), look=(0.0, 0.0, 1.5)):
        self.pos, self.look = tuple(pos), tuple(look)
        self.cursour_grabbed, self.speed = True, -5.0


def camera_matrices(pos, look, fov, width, height):
    """camera = proj * view, camera_inverse (column-major float32[16] each)."""
    cam, inv = (C.c_float * 16)(), (C.c_float * 16)()
    lib().svo_camera_matrices((C.c_float * 3)(*pos), (C.c_float * 3)(*look), fov, float(width), float(height), cam, inv)
    return np.array(cam, dtype=np.float32), np.array(inv, dtype=np.float32)
