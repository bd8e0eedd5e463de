// valu_rate.hip -- diagnostic: how many cycles does one wave64 vector instruction cost a gfx950 SIMD, as a function of
// the number of waves resident on it?  Decides how to read SQ_ACTIVE_INST_VALU (quad-cycles per wave) for the trace
// kernel: 4 cycles per instruction exclusive (the VALU would be ~88 % busy) or 2 cycles with two waves overlapped
// (~44 % busy: the kernel is latency-bound).  Build: hipcc --offload-arch=gfx950 -O3 -o build_ab/valu_rate tools/valu_rate.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

template <int KIND>
__global__ __launch_bounds__(256) void rate_kernel(float *out, unsigned long long *cyc, int iters) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const float m = 1.0000001f, c = 0.5f;
    unsigned int u0 = threadIdx.x, u1 = u0 * 3u, u2 = u0 * 5u, u3 = u0 * 7u;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
        if (KIND == 0) {  // 8 independent v_fma_f32
            asm volatile("v_fma_f32 %0, %0, %8, %9\n\tv_fma_f32 %1, %1, %8, %9\n\tv_fma_f32 %2, %2, %8, %9\n\tv_fma_f32 %3, %3, %8, %9\n\t"
                         "v_fma_f32 %4, %4, %8, %9\n\tv_fma_f32 %5, %5, %8, %9\n\tv_fma_f32 %6, %6, %8, %9\n\tv_fma_f32 %7, %7, %8, %9"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                         : "v"(m), "v"(c));
        } else if (KIND == 1) {  // 8 dependent v_fma_f32 (one chain)
            asm volatile("v_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\t"
                         "v_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2"
                         : "+v"(a0)
                         : "v"(m), "v"(c));
        } else if (KIND == 2) {  // 8 independent integer ops (v_add_u32 / v_xor / v_lshl_or / v_bfe mix)
            asm volatile("v_add_u32 %0, %0, %4\n\tv_xor_b32 %1, %1, %4\n\tv_lshl_or_b32 %2, %2, 1, %4\n\tv_bfe_u32 %3, %3, 1, 30\n\t"
                         "v_add_u32 %0, %0, %4\n\tv_xor_b32 %1, %1, %4\n\tv_lshl_or_b32 %2, %2, 1, %4\n\tv_bfe_u32 %3, %3, 1, 30"
                         : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3)
                         : "v"(u0 | 1u));
        } else if (KIND == 3) {  // 4 v_pk_fma_f32 (8 lanes-worth of f32 fma) on register pairs
            typedef float float2v __attribute__((ext_vector_type(2)));
            float2v p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, mm = {m, m}, cc = {c, c};
            asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n\tv_pk_fma_f32 %1, %1, %4, %5\n\tv_pk_fma_f32 %2, %2, %4, %5\n\tv_pk_fma_f32 %3, %3, %4, %5"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3)
                         : "v"(mm), "v"(cc));
            a0 = p0.x; a1 = p0.y; a2 = p1.x; a3 = p1.y; a4 = p2.x; a5 = p2.y; a6 = p3.x; a7 = p3.y;
        } else if (KIND == 4) {  // 4 VALU + 4 SALU interleaved
            unsigned int s = 0;
            asm volatile("v_fma_f32 %0, %0, %5, %6\n\ts_add_u32 %4, %4, 1\n\tv_fma_f32 %1, %1, %5, %6\n\ts_add_u32 %4, %4, 1\n\t"
                         "v_fma_f32 %2, %2, %5, %6\n\ts_add_u32 %4, %4, 1\n\tv_fma_f32 %3, %3, %5, %6\n\ts_add_u32 %4, %4, 1"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+s"(s)
                         : "v"(m), "v"(c)
                         : "scc");
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(u0 ^ u1 ^ u2 ^ u3);
    if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * 256 + threadIdx.x) >> 6] = t1 - t0;
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    const int iters = 20000;
    float *out;
    unsigned long long *cyc;
    hipMalloc(&out, (size_t)cus * 8 * 256 * sizeof(float));
    hipMalloc(&cyc, (size_t)cus * 8 * 4 * sizeof(unsigned long long));
    const char *names[5] = {"8 indep v_fma_f32", "8 dep v_fma_f32", "8 indep int VALU", "4 v_pk_fma_f32", "4 v_fma + 4 s_add"};
    const int per_iter[5] = {8, 8, 8, 4, 4};
    printf("CUs %d, clock %d kHz\n", cus, prop.clockRate);
    if (hipGetLastError() != hipSuccess) { printf("hip error at start\n"); return 1; }
    for (int kind = 0; kind < 5; kind++) {
        for (int wps : {1, 2, 3, 4, 6, 8}) {
            const int blocks = cus * wps;  // 256-thread blocks: one wave per SIMD each
            hipEvent_t e0, e1;
            hipEventCreate(&e0);
            hipEventCreate(&e1);
            auto launch = [&]() {
                switch (kind) {
                    case 0: hipLaunchKernelGGL(rate_kernel<0>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters); break;
                    case 1: hipLaunchKernelGGL(rate_kernel<1>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters); break;
                    case 2: hipLaunchKernelGGL(rate_kernel<2>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters); break;
                    case 3: hipLaunchKernelGGL(rate_kernel<3>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters); break;
                    default: hipLaunchKernelGGL(rate_kernel<4>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters); break;
                }
            };
            launch();
            if (hipDeviceSynchronize() != hipSuccess) { printf("sync failed: %s\n", hipGetErrorString(hipGetLastError())); return 1; }
            hipEventRecord(e0);
            launch();
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            std::vector<unsigned long long> h((size_t)blocks * 4);
            hipMemcpy(h.data(), cyc, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
            double mean = 0;
            for (auto v : h) mean += (double)v;
            mean /= (double)h.size();
            const double instr_per_wave = (double)iters * per_iter[kind];
            // cycles the SIMD spends per wave-instruction = wave lifetime / (instructions per wave * waves per SIMD)
            printf("%-20s waves/SIMD %d: %.3f ms, wave lifetime %.0f cyc, %.2f cyc per instr per wave, %.2f cyc of SIMD time per wave-instr\n",
                   names[kind], wps, ms, mean, mean / instr_per_wave, mean / instr_per_wave / wps);
        }
    }
    return 0;
}
