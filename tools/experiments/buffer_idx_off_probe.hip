// Does a gfx950 buffer load with BOTH idxen and offen (address = base + index * stride + offset) accept an offset beyond the stride,
// and how is it range-checked?  Descriptor: stride 4, num_records = n (records), raw dword format.
// build: hipcc -O2 --offload-arch=gfx950 -o build/buffer_idx_off_probe tools/experiments/buffer_idx_off_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__global__ void k(const unsigned *buf, unsigned n, unsigned stride, unsigned numrec, const unsigned *idx, const unsigned *off, unsigned *out, int cnt) {
    const u32x4 rs = {(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)buf),
                      ((unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((size_t)buf >> 32)) & 0xFFFFu) | (stride << 16),
                      (unsigned)__builtin_amdgcn_readfirstlane((int)numrec), 0x00020000u};
    int i = threadIdx.x;
    if (i < cnt) {
        u32x2 a = {idx[i], off[i]};
        unsigned v;
        asm volatile("buffer_load_dword %0, %1, %2, 0 idxen offen\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(a), "s"(rs) : "memory");
        out[i] = v;
    }
}
int main() {
    const unsigned n = 1024;
    std::vector<unsigned> h(n);
    for (unsigned i = 0; i < n; i++) h[i] = 0xAB000000u | i;
    unsigned *d, *di, *dof, *dout;
    hipMalloc(&d, n * 4); hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
    unsigned idx[8] = {0, 10, 10, 10, 1000, 1023, 1024, 1016}, off[8] = {0, 0, 4, 28, 28, 0, 0, 28};
    hipMalloc(&di, 32); hipMalloc(&dof, 32); hipMalloc(&dout, 32);
    hipMemcpy(di, idx, 32, hipMemcpyHostToDevice); hipMemcpy(dof, off, 32, hipMemcpyHostToDevice);
    for (unsigned numrec : {n, n * 4}) {
        hipMemset(dout, 0xFF, 32);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, n, 4u, numrec, di, dof, dout, 8);
        hipDeviceSynchronize();
        unsigned o[8]; hipMemcpy(o, dout, 32, hipMemcpyDeviceToHost);
        printf("stride 4, num_records %u:\n", numrec);
        for (int i = 0; i < 8; i++) printf("  index %4u offset %2u -> %08x (word %u would be %08x)\n", idx[i], off[i], o[i], idx[i] + off[i] / 4, idx[i] + off[i] / 4 < n ? h[idx[i] + off[i] / 4] : 0u);
    }
    return 0;
}
