// Does gfx950 drop LDS writes (and zero LDS reads) beyond a workgroup's allocation?  Every workgroup fills its 1 KiB of dynamic LDS
// with its own id, writes a poison word at byte offsets 1 KiB .. 64 KiB (beyond its allocation: if the hardware did not clamp, that
// is other workgroups' LDS on the same CU), then all check their own words and what an out-of-range read returns.
// build: hipcc -O2 --offload-arch=gfx950 -o build/lds_oob_probe tools/experiments/lds_oob_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((address_space(3))) unsigned lds_u32;
__global__ __launch_bounds__(256) void k(unsigned *bad, unsigned *oob_read_nonzero, int rounds) {
    extern __shared__ unsigned lds[];
    const unsigned id = blockIdx.x + 1u;
    lds[threadIdx.x] = id;
    __syncthreads();
    for (int r = 0; r < rounds; r++) {
        for (unsigned off = 1024u + threadIdx.x * 4u; off < 65536u; off += 1024u) {
            *(volatile lds_u32 *)(uintptr_t)off = 0xDEAD0000u | id;
            const unsigned v = *(volatile lds_u32 *)(uintptr_t)off;
            if (v != 0u) atomicAdd(oob_read_nonzero, 1u);
        }
        __syncthreads();
        if (lds[threadIdx.x] != id) atomicAdd(bad, 1u);
        __syncthreads();
    }
}
int main() {
    unsigned *d;
    hipMalloc(&d, 8);
    hipMemset(d, 0, 8);
    hipLaunchKernelGGL(k, dim3(256 * 16), dim3(256), 1024, 0, d, d + 1, 50);
    hipDeviceSynchronize();
    unsigned h[2];
    hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
    printf("words of a workgroup's own LDS changed by other workgroups' out-of-range writes: %u; out-of-range reads that returned non-zero: %u\n", h[0], h[1]);
    return 0;
}
