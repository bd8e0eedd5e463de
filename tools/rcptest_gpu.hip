// rcptest_gpu.hip -- proof by exhaustion (round 4): for every f32 d with 2^-20 <= |d| <= 2^64 (the range of a clean ray's direction in grid
// units, with margin), v_rcp_f32 followed by one Newton step,  y = fma(fma(-d, r, 1), r, r),  equals the correctly rounded 1 / d.
// The pick-up of a pooled ray needs RN(1 / Dr) three times; the IEEE division sequence costs eleven instructions each.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o build/rcptest tools/rcptest_gpu.hip ; run: build/rcptest
#include <hip/hip_runtime.h>

#include <cstdio>

__global__ void rcp_check(unsigned long long *bad, unsigned *first_bad) {
    const unsigned e = 127u - 20u + blockIdx.y;  // biased exponent
    unsigned long long local = 0;
    for (unsigned m = blockIdx.x * blockDim.x + threadIdx.x; m < (1u << 23); m += gridDim.x * blockDim.x) {
        for (unsigned s = 0; s < 2; s++) {
            const unsigned bits = (s << 31) | (e << 23) | m;
            const float d = __uint_as_float(bits);
            const float want = 1.0f / d;  // correctly rounded (hipcc default: -fhip-fp32-correctly-rounded-divide-sqrt)
            const float r = __builtin_amdgcn_rcpf(d);
            const float y = __builtin_fmaf(__builtin_fmaf(-d, r, 1.0f), r, r);
            if (__float_as_uint(y) != __float_as_uint(want)) {
                local++;
                atomicMin(first_bad, bits & 0x7FFFFFFFu);
            }
        }
    }
    if (local) atomicAdd(bad, local);
}

int main() {
    unsigned long long *bad, h = 0;
    unsigned *fb, hf = 0xFFFFFFFFu;
    if (hipMalloc(&bad, 8) != hipSuccess || hipMalloc(&fb, 4) != hipSuccess) { printf("no device\n"); return 1; }
    hipMemcpy(bad, &h, 8, hipMemcpyHostToDevice);
    hipMemcpy(fb, &hf, 4, hipMemcpyHostToDevice);
    const int n_exp = 20 + 64 + 1;
    hipLaunchKernelGGL(rcp_check, dim3(1024, n_exp), dim3(256), 0, 0, bad, fb);
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
    hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost);
    hipMemcpy(&hf, fb, 4, hipMemcpyDeviceToHost);
    printf("v_rcp_f32 + one Newton step vs IEEE 1/d: %d binades x 2^23 significands x 2 signs = %llu values, %llu mismatches", n_exp,
           (unsigned long long)n_exp << 24, h);
    if (h) printf(" (smallest |d| that differs: 0x%08x)", hf);
    printf("\n");
    return h ? 2 : 0;
}
