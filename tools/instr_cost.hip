// instr_cost.hip -- diagnostic: issue cost of the instruction kinds the trace kernel is made of, on a gfx950 SIMD that
// holds 6 (and 8) waves, the trace kernel's occupancy.  Each test is a long unrolled block (64 instructions, 8 independent
// register chains) inside a loop, so loop overhead is ~5 %; cost = wave lifetime / instructions / waves per SIMD = SIMD
// cycles per wave-instruction.  Build: hipcc --offload-arch=gfx950 -O3 -o build_ab/instr_cost tools/instr_cost.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define BLOCK8x8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X)

template <int KIND>
__global__ __launch_bounds__(256) void cost_kernel(unsigned *out, unsigned long long *cyc, int iters) {
    unsigned r0 = threadIdx.x * 2654435761u, r1 = r0 ^ 0x1234567u, r2 = r0 + 77u, r3 = r0 * 3u, r4 = r0 >> 3, r5 = ~r0, r6 = r0 + 5u, r7 = r0 ^ 99u;
    unsigned a = threadIdx.x | 1u, b = (threadIdx.x & 15u) + 1u, sgl = 0;
    extern __shared__ unsigned lds[];
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    const unsigned laddr = threadIdx.x * 4u;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
#define OPS(TXT) asm volatile(TXT : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7), "+s"(sgl) : "v"(a), "v"(b), "v"(laddr) : "vcc", "scc", "memory")
#define R(n) "%" #n
        if (KIND == 0) {
#define X(n) "v_fma_f32 " R(n) ", " R(n) ", %9, %10\n\t"
            OPS(BLOCK8x8(X));
#undef X
        } else if (KIND == 1) {
#define X(n) "v_mul_f32 " R(n) ", " R(n) ", %9\n\t"
            OPS(BLOCK8x8(X));
#undef X
        } else if (KIND == 2) {
#define X(n) "v_add_u32 " R(n) ", " R(n) ", %9\n\t"
            OPS(BLOCK8x8(X));
#undef X
        } else if (KIND == 3) {
#define X(n) "v_bfe_u32 " R(n) ", " R(n) ", %10, 7\n\t"
            OPS(BLOCK8x8(X));
#undef X
        } else if (KIND == 4) {
#define X(n) "v_lshl_or_b32 " R(n) ", " R(n) ", 1, %9\n\t"
            OPS(BLOCK8x8(X));
#undef X
        } else if (KIND == 5) {
#define X(n) "v_and_or_b32 " R(n) ", " R(n) ", %9, %10\n\t"
            OPS(BLOCK8x8(X));
#undef X
        } else if (KIND == 6) {
#define X(n) "v_xor_b32 " R(n) ", " R(n) ", %9\n\t"
            OPS(BLOCK8x8(X));
#undef X
        } else if (KIND == 7) {
#define X(n) "v_cvt_f32_i32 " R(n) ", " R(n) "\n\t"
            OPS(BLOCK8x8(X));
#undef X
        } else if (KIND == 8) {
#define X(n) "v_cvt_flr_i32_f32 " R(n) ", " R(n) "\n\t"
            OPS(BLOCK8x8(X));
#undef X
        } else if (KIND == 9) {
#define X(n) "v_min3_f32 " R(n) ", " R(n) ", %9, %10\n\t"
            OPS(BLOCK8x8(X));
#undef X
        } else if (KIND == 10) {
#define X(n) "v_cmp_lt_u32 vcc, " R(n) ", %9\n\tv_cndmask_b32 " R(n) ", " R(n) ", %10, vcc\n\t"
            OPS(REP8(X) REP8(X) REP8(X) REP8(X));  // 32 pairs = 64 instructions
#undef X
        } else if (KIND == 11) {
#define X(n) "v_bfi_b32 " R(n) ", " R(n) ", %9, %10\n\t"
            OPS(BLOCK8x8(X));
#undef X
        } else if (KIND == 12) {
#define X(n) "v_ffbh_u32 " R(n) ", " R(n) "\n\t"
            OPS(BLOCK8x8(X));
#undef X
        } else if (KIND == 13) {
#define X(n) "v_readlane_b32 %8, " R(n) ", 3\n\t"
            OPS(BLOCK8x8(X));
#undef X
        } else if (KIND == 14) {
#define X(n) "s_add_u32 %8, %8, 1\n\t"
            OPS(BLOCK8x8(X));
#undef X
        } else if (KIND == 15) {
#define X(n) "ds_write_b32 %11, " R(n) "\n\t"
            OPS(BLOCK8x8(X) "s_waitcnt lgkmcnt(0)\n\t");
#undef X
        } else if (KIND == 16) {
#define X(n) "ds_read_b32 " R(n) ", %11\n\t"
            OPS(BLOCK8x8(X) "s_waitcnt lgkmcnt(0)\n\t");
#undef X
        } else if (KIND == 17) {
#define X(n) "v_mov_b32 " R(n) ", %9\n\t"
            OPS(BLOCK8x8(X));
#undef X
        } else if (KIND == 18) {
#define X(n) "v_add_lshl_u32 " R(n) ", " R(n) ", %9, 2\n\t"
            OPS(BLOCK8x8(X));
#undef X
        } else if (KIND == 19) {  // a step-like mix: 2 fma, mul, sub, bfe, lshl_or, and_or, cndmask pair
#define X(n) "v_fma_f32 " R(n) ", " R(n) ", %9, %10\n\tv_mul_f32 " R(n) ", " R(n) ", %9\n\tv_bfe_u32 " R(n) ", " R(n) ", %10, 9\n\tv_lshl_or_b32 " R(n) ", " R(n) ", 1, %9\n\tv_sub_f32 " R(n) ", " R(n) ", %9\n\tv_and_or_b32 " R(n) ", " R(n) ", %9, %10\n\tv_xor_b32 " R(n) ", " R(n) ", %9\n\tv_add_u32 " R(n) ", " R(n) ", %10\n\t"
            OPS(REP8(X));
#undef X
        } else if (KIND == 20) {  // alternating VALU / SALU
#define X(n) "v_add_u32 " R(n) ", " R(n) ", %9\n\ts_add_u32 %8, %8, 1\n\t"
            OPS(REP8(X) REP8(X) REP8(X) REP8(X));
#undef X
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 256 + threadIdx.x] = r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7 ^ sgl;
    if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * 256 + threadIdx.x) >> 6] = t1 - t0;
}

template <int K>
void run_one(const char *name, int cus, unsigned *out, unsigned long long *cyc) {
    const int iters = 2000;
    for (int wps : {1, 6, 8}) {
        const int blocks = cus * wps;
        hipLaunchKernelGGL(cost_kernel<K>, dim3(blocks), dim3(256), 1024, 0, out, cyc, iters);
        if (hipDeviceSynchronize() != hipSuccess) { printf("%s failed\n", name); return; }
        hipLaunchKernelGGL(cost_kernel<K>, dim3(blocks), dim3(256), 1024, 0, out, cyc, iters);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h((size_t)blocks * 4);
        hipMemcpy(h.data(), cyc, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        double mean = 0;
        for (auto v : h) mean += (double)v;
        mean /= (double)h.size();
        printf("%-28s waves/SIMD %d: %6.2f cycles per instruction per wave, %5.2f SIMD cycles per wave-instruction\n", name, wps,
               mean / (iters * 64.0), mean / (iters * 64.0) / wps);
    }
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    unsigned *out;
    unsigned long long *cyc;
    hipMalloc(&out, (size_t)cus * 8 * 256 * sizeof(unsigned));
    hipMalloc(&cyc, (size_t)cus * 8 * 4 * sizeof(unsigned long long));
    run_one<0>("v_fma_f32", cus, out, cyc);
    run_one<1>("v_mul_f32", cus, out, cyc);
    run_one<2>("v_add_u32", cus, out, cyc);
    run_one<3>("v_bfe_u32", cus, out, cyc);
    run_one<4>("v_lshl_or_b32", cus, out, cyc);
    run_one<5>("v_and_or_b32", cus, out, cyc);
    run_one<6>("v_xor_b32", cus, out, cyc);
    run_one<7>("v_cvt_f32_i32", cus, out, cyc);
    run_one<8>("v_cvt_flr_i32_f32", cus, out, cyc);
    run_one<9>("v_min3_f32", cus, out, cyc);
    run_one<10>("v_cmp + v_cndmask (pair)", cus, out, cyc);
    run_one<11>("v_bfi_b32", cus, out, cyc);
    run_one<12>("v_ffbh_u32", cus, out, cyc);
    run_one<13>("v_readlane_b32", cus, out, cyc);
    run_one<14>("s_add_u32", cus, out, cyc);
    run_one<15>("ds_write_b32", cus, out, cyc);
    run_one<16>("ds_read_b32", cus, out, cyc);
    run_one<17>("v_mov_b32", cus, out, cyc);
    run_one<18>("v_add_lshl_u32", cus, out, cyc);
    run_one<19>("mixed float/int VALU", cus, out, cyc);
    run_one<20>("v_add_u32 / s_add_u32 pairs", cus, out, cyc);
    return 0;
}
