"""Device context: replaces reference src/gpu.rs (wgpu instance/adapter/device/queue) with a
HIP context handle from the C ABI."""
import ctypes as C

from ._lib import SvoError, lib

OPT_VARIANT, OPT_TIMING, OPT_GRID_BLOCKS, OPT_REFILL_MIN, OPT_STRIP_ITEMS, OPT_DYNAMIC_STRIPS, OPT_PRIO_STEPS, OPT_DEBUG_BUFFER, OPT_SCHEDULE, OPT_TREE_DEPTH, OPT_BLOCK_SHAPE, OPT_SCAN_CLEARS_COUNTERS, OPT_FUSED_SHADOWS, OPT_PAIR_TABLE, OPT_CULL, OPT_CAMERA_SHORTCUT, OPT_SCHEDULE_MOTION = range(17)
VARIANT_RESTART, VARIANT_STACK = 0, 1


class Gpu:
    def __init__(self, device=0, stream=None):
        h = C.c_void_p()
        rc = lib().svo_ctx_create(device, C.byref(h))
        if rc != 0:
            # the reference unwraps (gpu.rs:24,39): fail loudly, there is no fallback device
            raise SvoError(f"svo_ctx_create(device={device}) failed with status {rc} (no usable HIP device?)")
        self._h = h
        self.device = device
        # torch allocates and copies on its current stream: launch there so tensors handed to the
        # C ABI are ordered with the kernels that read or write them
        import torch
        self.set_stream(stream if stream is not None else torch.cuda.current_stream(device).cuda_stream)

    @classmethod
    def new(cls, device=0):
        return cls(device)

    def check(self, rc):
        if rc != 0:
            raise SvoError(f"status {rc}: {lib().svo_last_error(self._h).decode()}")

    def set_stream(self, hip_stream):
        """hip_stream: a hipStream_t handle as int (0 = HIP's default stream)"""
        self.check(lib().svo_ctx_set_stream(self._h, C.c_void_p(hip_stream), 0))

    def use_own_stream(self):
        self.check(lib().svo_ctx_set_stream(self._h, None, 1))

    def set_option(self, option, value):
        self.check(lib().svo_set_option(self._h, option, int(value)))

    def sync(self):
        """device.poll(Maintain::Wait)"""
        self.check(lib().svo_sync(self._h))

    def last_render_ms(self):
        ms = C.c_float()
        self.check(lib().svo_last_render_ms(self._h, C.byref(ms)))
        return ms.value

    def strip_classes(self, n_strips):
        """class byte per 64-pixel block of the last pixel frame that ran the culling pass (0xFF = culled); diagnostics"""
        import numpy as np
        out = np.zeros(n_strips, dtype=np.uint8)
        self.check(lib().svo_diag_strip_classes(self._h, out.ctypes.data_as(C.c_void_p), n_strips))
        return out

    def timing_collect(self, cap=65536):
        """durations (ms) of the trace launches recorded since the last collect"""
        import numpy as np
        buf = np.empty(cap, dtype=np.float32)
        n = C.c_size_t()
        self.check(lib().svo_timing_collect(self._h, buf.ctypes.data_as(C.POINTER(C.c_float)), cap, C.byref(n)))
        return buf[:n.value].copy()

    # ---- multi-GPU frame end: RCCL behind the C ABI (svo_comm_*, svo_gather_frame) ----
    @staticmethod
    def comm_unique_id():
        """128 bytes rank 0 hands to the other ranks (ncclGetUniqueId)"""
        buf = (C.c_uint8 * 128)()
        rc = lib().svo_comm_unique_id(buf)
        if rc != 0:
            raise SvoError(f"svo_comm_unique_id failed with status {rc} (librccl missing?)")
        return bytes(buf)

    def comm_init_rank(self, unique_id, world, rank):
        """collective: every rank's context, with rank 0's id (one process per GPU)"""
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        self.check(lib().svo_comm_init_rank(self._h, buf, world, rank))

    @staticmethod
    def comm_init_all(gpus):
        """one process driving several devices: gpus[r] becomes rank r (ncclCommInitAll)"""
        arr = (C.c_void_p * len(gpus))(*[g._h for g in gpus])
        rc = lib().svo_comm_init_all(len(gpus), arr)
        if rc != 0:
            raise SvoError(f"status {rc}: {lib().svo_last_error(gpus[0]._h).decode()}")

    def comm_destroy(self):
        self.check(lib().svo_comm_destroy(self._h))

    def gather_frame(self, send, recv_on_root=None, root=0):
        """this rank's part of the frame-end gather, on the context's stream: tensor `send` -> recv_on_root[rank] on root"""
        self.check(lib().svo_gather_frame(self._h, send.data_ptr(), send.numel() * send.element_size(),
                                          recv_on_root.data_ptr() if recv_on_root is not None else None, root))

    def gather_wait(self):
        """the context's stream waits for the gathers issued so far"""
        self.check(lib().svo_gather_wait(self._h))

    def close(self):
        if getattr(self, "_h", None):
            lib().svo_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        if lib is not None:  # (module globals are gone at interpreter exit)
            self.close()
