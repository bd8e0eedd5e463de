"""Device context: replaces reference src/gpu.rs (wgpu instance/adapter/device/queue) with a
HIP context handle from the C ABI."""
import ctypes as C

from ._lib import SvoError, lib

OPT_VARIANT, OPT_TIMING, OPT_GRID_BLOCKS, OPT_REFILL_MIN, OPT_STRIP_ITEMS, OPT_DYNAMIC_STRIPS, OPT_PRIO_STEPS, OPT_DEBUG_BUFFER, OPT_SCHEDULE, OPT_TREE_DEPTH, OPT_BLOCK_SHAPE, OPT_SCAN_CLEARS_COUNTERS, OPT_FUSED_SHADOWS = range(13)
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

    def timing_collect(self, cap=65536):
        """durations (ms) of the trace launches recorded since the last collect"""
        import numpy as np
        buf = np.empty(cap, dtype=np.float32)
        n = C.c_size_t()
        self.check(lib().svo_timing_collect(self._h, buf.ctypes.data_as(C.POINTER(C.c_float)), cap, C.byref(n)))
        return buf[:n.value].copy()

    def close(self):
        if getattr(self, "_h", None):
            lib().svo_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        if lib is not None:  # (module globals are gone at interpreter exit)
            self.close()
