// Exhaustive check of the reciprocal-based division used by trace_stack_kernel's stepping arithmetic:
// for EVERY pair of f32 significands (a, d in [1, 2), 2^23 x 2^23 pairs) compare IEEE a / d with
//   one correction : q0 = a*y;  q1 = fma(fma(-d, q0, a), y, q0)                      y = RN(1 / d)
//   two corrections: q2 = fma(fma(-d, q1, a), y, q1)
// Both forms are invariant under scaling a and d by powers of two as long as nothing over/underflows (the
// clean-ray ranges guarantee that), and under sign changes, so the significand pairs cover every
// operand the kernel can see.
// build: hipcc -O2 --offload-arch=gfx950 -ffp-contract=off -o divtest_gpu tools/divtest_gpu.hip
// run:   ./divtest_gpu [slices=64] [first_slice=0] [n_slices=all]     (about a minute for everything)
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ __launch_bounds__(256) void sweep(uint32_t ma_lo, uint32_t ma_hi, unsigned long long *bad1, unsigned long long *bad2,
                                             uint8_t *bad_d, uint32_t *examples) {
    const uint32_t md = blockIdx.x * 256u + threadIdx.x;  // grid covers 2^23 divisors
    const float d = __uint_as_float(0x3F800000u | md);
    const float y = 1.0f / d;
    uint32_t n0 = 0, n1 = 0, n2 = 0;
    for (uint32_t ma = ma_lo; ma < ma_hi; ma++) {
        const float a = __uint_as_float(0x3F800000u | ma);
        const float want = a / d;
        const float q0 = a * y;
        const float q1 = __builtin_fmaf(__builtin_fmaf(-d, q0, a), y, q0);
        const float q2 = __builtin_fmaf(__builtin_fmaf(-d, q1, a), y, q1);
        if (__float_as_uint(q1) != __float_as_uint(want)) {
            if (n1 == 0) {
                const unsigned long long slot = atomicAdd(&bad1[1], 1ull);
                if (slot < 64) { examples[2 * slot] = ma; examples[2 * slot + 1] = md; }
            }
            n1++;
        }
        n2 += __float_as_uint(q2) != __float_as_uint(want);
        n0 += __float_as_uint(q0) != __float_as_uint(want);  // control: the uncorrected product must fail often
    }
    if (n1) { atomicAdd(&bad1[0], (unsigned long long)n1); bad_d[md] = 1; }
    if (n2) atomicAdd(&bad2[0], (unsigned long long)n2);
    if (n0) atomicAdd(&bad1[2], (unsigned long long)n0);
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

int main(int argc, char **argv) {
    const uint32_t slices = argc > 1 ? (uint32_t)atoi(argv[1]) : 64u;
    const uint32_t first = argc > 2 ? (uint32_t)atoi(argv[2]) : 0u;
    const uint32_t count = argc > 3 ? (uint32_t)atoi(argv[3]) : slices - first;
    const uint32_t N = 1u << 23;
    unsigned long long *bad1, *bad2;
    uint8_t *bad_d;
    uint32_t *examples;
    CK(hipMalloc((void **)&bad1, 24));
    CK(hipMalloc((void **)&bad2, 8));
    CK(hipMalloc((void **)&bad_d, N));
    CK(hipMalloc((void **)&examples, 128 * sizeof(uint32_t)));
    CK(hipMemset(bad1, 0, 24));
    CK(hipMemset(bad2, 0, 8));
    CK(hipMemset(bad_d, 0, N));
    CK(hipMemset(examples, 0, 128 * sizeof(uint32_t)));
    const uint32_t per = N / slices;
    for (uint32_t s = first; s < first + count && s < slices; s++) {
        hipLaunchKernelGGL(sweep, dim3(N / 256), dim3(256), 0, 0, s * per, s == slices - 1 ? N : (s + 1) * per, bad1, bad2, bad_d, examples);
        CK(hipDeviceSynchronize());
        unsigned long long h1[3], h2;
        CK(hipMemcpy(h1, bad1, 24, hipMemcpyDeviceToHost));
        CK(hipMemcpy(&h2, bad2, 8, hipMemcpyDeviceToHost));
        printf("slice %u/%u: numerators [%u, %u) x all 2^23 divisors: mismatches so far: no correction %llu (control), one %llu, two %llu\n",
               s + 1, slices, s * per, s == slices - 1 ? N : (s + 1) * per, h1[2], h1[0], h2);
        fflush(stdout);
    }
    std::vector<uint8_t> flags(N);
    uint32_t ex[128];
    unsigned long long h1[3], h2;
    CK(hipMemcpy(flags.data(), bad_d, N, hipMemcpyDeviceToHost));
    CK(hipMemcpy(ex, examples, sizeof ex, hipMemcpyDeviceToHost));
    CK(hipMemcpy(h1, bad1, 24, hipMemcpyDeviceToHost));
    CK(hipMemcpy(&h2, bad2, 8, hipMemcpyDeviceToHost));
    uint32_t n_bad_d = 0;
    for (uint32_t i = 0; i < N; i++) n_bad_d += flags[i];
    printf("pairs checked: %llu\n", (unsigned long long)count * per * N);
    printf("no correction (control): %llu mismatches\n", h1[2]);
    printf("two corrections: %llu mismatches\n", h2);
    printf("one correction : %llu mismatches over %u distinct divisor significands\n", h1[0], n_bad_d);
    for (uint32_t i = 0; i < 64 && i < h1[1]; i++) printf("  example: a = 0x%08x  d = 0x%08x\n", 0x3F800000u | ex[2 * i], 0x3F800000u | ex[2 * i + 1]);
    if (n_bad_d && n_bad_d <= 64)
        for (uint32_t i = 0; i < N; i++)
            if (flags[i]) printf("  divisor significand with a failing numerator: 0x%06x\n", i);
    return h2 ? 1 : 0;
}
