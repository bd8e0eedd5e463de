// issue_rate.hip -- diagnostic (round 4): what one wave64 instruction costs a gfx950 SIMD, by WALL CLOCK.
//
// Round 2 left two microbenchmarks that disagree: tools/instr_cost.hip reads 1.15-1.4 "SIMD cycles" per
// full-rate vector instruction at 6 waves per SIMD, tools/valu_rate.hip 1.7-2.0.  Both divide s_memtime deltas; neither looks
// at the wall clock, so neither knows what clock the chip held.  This one times every kind with HIP events over a launch of
// >= 2 ms, reads s_memtime AND s_memrealtime (constant 100 MHz) in every wave to get the shader clock actually held, and
// reports, per kind: wave-instructions per SIMD per microsecond, SIMD cycles per wave-instruction at the measured clock and
// at the nominal 2.4 GHz.  Every test is one unrolled block of 64 instructions on 8 independent register chains inside a
// loop (loop overhead: 3 scalar instructions per 64).
// Build: hipcc --offload-arch=gfx950 -O3 -o build_ab/issue_rate tools/issue_rate.hip ; run: build_ab/issue_rate [waves_per_simd ...]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define BLOCK8x8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X)
#define R(n) "%" #n

// operands: %0..%7 chains (VGPR), %8..%15 second halves for 64-bit (packed) chains, %16 scalar chain, %17 %18 vector inputs,
// %19 LDS address, %20:%21 a scalar pair (mask), %22 scalar input
#define OPS(TXT)                                                                                                            \
    asm volatile(TXT : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7), "+v"(p0), "+v"(p1),  \
                 "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7), "+s"(sgl)                                      \
                 : "v"(a), "v"(b), "v"(laddr), "s"(smask), "s"(sin)                                                         \
                 : "vcc", "scc", "memory")

typedef float float2v __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ __launch_bounds__(256) void rate_kernel(unsigned *out, unsigned long long *clk, int iters) {
    unsigned r0 = threadIdx.x * 2654435761u, r1 = r0 ^ 0x1234567u, r2 = r0 + 77u, r3 = r0 * 3u, r4 = r0 >> 3, r5 = ~r0, r6 = r0 + 5u, r7 = r0 ^ 99u;
    float2v p0 = {1.0f, 2.0f}, p1 = {3.0f, 4.0f}, p2 = {5.0f, 6.0f}, p3 = {7.0f, 8.0f}, p4 = {9.0f, 1.5f}, p5 = {2.5f, 3.5f}, p6 = {4.5f, 5.5f}, p7 = {6.5f, 7.5f};
    unsigned a = threadIdx.x | 1u, b = (threadIdx.x & 15u) + 1u, sgl = 0, sin = 3;
    unsigned long long smask = 0x5555555555555555ull;
    extern __shared__ unsigned lds[];
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    const unsigned laddr = threadIdx.x * 4u;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), q0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; i++) {
        if (KIND == 0) {
#define X(n) "v_fma_f32 " R(n) ", " R(n) ", %17, %18\n\t"
            OPS(BLOCK8x8(X));
#undef X
        } else if (KIND == 1) {
#define X(n) "v_mul_f32 " R(n) ", " R(n) ", %17\n\t"
            OPS(BLOCK8x8(X));
#undef X
        } else if (KIND == 2) {
#define X(n) "v_add_f32 " R(n) ", " R(n) ", %17\n\t"
            OPS(BLOCK8x8(X));
#undef X
        } else if (KIND == 3) {
#define X(n) "v_add_u32 " R(n) ", " R(n) ", %17\n\t"
            OPS(BLOCK8x8(X));
#undef X
        } else if (KIND == 4) {
#define X(n) "v_xor_b32 " R(n) ", " R(n) ", %17\n\t"
            OPS(BLOCK8x8(X));
#undef X
        } else if (KIND == 5) {
#define X(n) "v_lshlrev_b32 " R(n) ", %18, " R(n) "\n\t"
            OPS(BLOCK8x8(X));
#undef X
        } else if (KIND == 6) {
#define X(n) "v_lshrrev_b32 " R(n) ", 3, " R(n) "\n\t"
            OPS(BLOCK8x8(X));
#undef X
        } else if (KIND == 7) {
#define X(n) "v_bfe_u32 " R(n) ", " R(n) ", %18, 7\n\t"
            OPS(BLOCK8x8(X));
#undef X
        } else if (KIND == 8) {
#define X(n) "v_lshl_or_b32 " R(n) ", " R(n) ", 1, %17\n\t"
            OPS(BLOCK8x8(X));
#undef X
        } else if (KIND == 9) {
#define X(n) "v_and_or_b32 " R(n) ", " R(n) ", %17, %18\n\t"
            OPS(BLOCK8x8(X));
#undef X
        } else if (KIND == 10) {
#define X(n) "v_bfi_b32 " R(n) ", " R(n) ", %17, %18\n\t"
            OPS(BLOCK8x8(X));
#undef X
        } else if (KIND == 11) {
#define X(n) "v_add_lshl_u32 " R(n) ", " R(n) ", %17, 2\n\t"
            OPS(BLOCK8x8(X));
#undef X
        } else if (KIND == 12) {
#define X(n) "v_lshl_add_u32 " R(n) ", " R(n) ", 2, %17\n\t"
            OPS(BLOCK8x8(X));
#undef X
        } else if (KIND == 13) {
#define X(n) "v_or3_b32 " R(n) ", " R(n) ", %17, %18\n\t"
            OPS(BLOCK8x8(X));
#undef X
        } else if (KIND == 14) {
#define X(n) "v_min3_f32 " R(n) ", " R(n) ", %17, %18\n\t"
            OPS(BLOCK8x8(X));
#undef X
        } else if (KIND == 15) {
#define X(n) "v_min_f32 " R(n) ", " R(n) ", %17\n\t"
            OPS(BLOCK8x8(X));
#undef X
        } else if (KIND == 16) {
#define X(n) "v_cvt_f32_i32 " R(n) ", " R(n) "\n\t"
            OPS(BLOCK8x8(X));
#undef X
        } else if (KIND == 17) {
#define X(n) "v_cvt_flr_i32_f32 " R(n) ", " R(n) "\n\t"
            OPS(BLOCK8x8(X));
#undef X
        } else if (KIND == 18) {
#define X(n) "v_ffbh_u32 " R(n) ", " R(n) "\n\t"
            OPS(BLOCK8x8(X));
#undef X
        } else if (KIND == 19) {  // VOP2 select on vcc (vcc written once in front)
#define X(n) "v_cndmask_b32 " R(n) ", " R(n) ", %17, vcc\n\t"
            OPS("v_cmp_lt_u32 vcc, %17, %18\n\t" BLOCK8x8(X));
#undef X
        } else if (KIND == 20) {  // VOP3 select on a scalar pair
#define X(n) "v_cndmask_b32 " R(n) ", " R(n) ", %17, %20\n\t"
            OPS(BLOCK8x8(X));
#undef X
        } else if (KIND == 21) {  // compare into vcc (VOPC)
#define X(n) "v_cmp_eq_f32 vcc, " R(n) ", %17\n\t"
            OPS(BLOCK8x8(X));
#undef X
        } else if (KIND == 22) {  // compare into a scalar pair (VOP3)
#define X(n) "v_cmp_eq_f32 s[10:11], " R(n) ", %17\n\t"
            asm volatile(BLOCK8x8(X) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7), "+v"(p0), "+v"(p1),
                         "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7), "+s"(sgl)
                         : "v"(a), "v"(b), "v"(laddr), "s"(smask), "s"(sin)
                         : "vcc", "scc", "memory", "s10", "s11");
#undef X
        } else if (KIND == 23) {
#define X(n) "v_mov_b32 " R(n) ", %17\n\t"
            OPS(BLOCK8x8(X));
#undef X
        } else if (KIND == 24) {  // packed f32: one instruction, two lanes-worth
            asm volatile("v_pk_fma_f32 %8, %8, %9, %10\n\tv_pk_fma_f32 %11, %11, %9, %10\n\tv_pk_fma_f32 %12, %12, %9, %10\n\tv_pk_fma_f32 %13, %13, %9, %10\n\t"
                         "v_pk_fma_f32 %14, %14, %9, %10\n\tv_pk_fma_f32 %15, %15, %9, %10\n\tv_pk_fma_f32 %8, %8, %9, %10\n\tv_pk_fma_f32 %11, %11, %9, %10\n\t"
                         "v_pk_fma_f32 %12, %12, %9, %10\n\tv_pk_fma_f32 %13, %13, %9, %10\n\tv_pk_fma_f32 %14, %14, %9, %10\n\tv_pk_fma_f32 %15, %15, %9, %10\n\t"
                         "v_pk_fma_f32 %8, %8, %9, %10\n\tv_pk_fma_f32 %11, %11, %9, %10\n\tv_pk_fma_f32 %12, %12, %9, %10\n\tv_pk_fma_f32 %13, %13, %9, %10\n\t"
                         "v_pk_fma_f32 %14, %14, %9, %10\n\tv_pk_fma_f32 %15, %15, %9, %10\n\tv_pk_fma_f32 %8, %8, %9, %10\n\tv_pk_fma_f32 %11, %11, %9, %10\n\t"
                         "v_pk_fma_f32 %12, %12, %9, %10\n\tv_pk_fma_f32 %13, %13, %9, %10\n\tv_pk_fma_f32 %14, %14, %9, %10\n\tv_pk_fma_f32 %15, %15, %9, %10\n\t"
                         "v_pk_fma_f32 %8, %8, %9, %10\n\tv_pk_fma_f32 %11, %11, %9, %10\n\tv_pk_fma_f32 %12, %12, %9, %10\n\tv_pk_fma_f32 %13, %13, %9, %10\n\t"
                         "v_pk_fma_f32 %14, %14, %9, %10\n\tv_pk_fma_f32 %15, %15, %9, %10\n\tv_pk_fma_f32 %8, %8, %9, %10\n\tv_pk_fma_f32 %11, %11, %9, %10\n\t"
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7), "+v"(p0), "+v"(p1), "+v"(p2),
                           "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7)
                         :
                         : "memory");
        } else if (KIND == 25) {
            asm volatile("v_pk_add_f32 %8, %8, %9\n\tv_pk_add_f32 %11, %11, %9\n\tv_pk_add_f32 %12, %12, %9\n\tv_pk_add_f32 %13, %13, %9\n\t"
                         "v_pk_add_f32 %14, %14, %9\n\tv_pk_add_f32 %15, %15, %9\n\tv_pk_add_f32 %8, %8, %9\n\tv_pk_add_f32 %11, %11, %9\n\t"
                         "v_pk_add_f32 %12, %12, %9\n\tv_pk_add_f32 %13, %13, %9\n\tv_pk_add_f32 %14, %14, %9\n\tv_pk_add_f32 %15, %15, %9\n\t"
                         "v_pk_add_f32 %8, %8, %9\n\tv_pk_add_f32 %11, %11, %9\n\tv_pk_add_f32 %12, %12, %9\n\tv_pk_add_f32 %13, %13, %9\n\t"
                         "v_pk_add_f32 %14, %14, %9\n\tv_pk_add_f32 %15, %15, %9\n\tv_pk_add_f32 %8, %8, %9\n\tv_pk_add_f32 %11, %11, %9\n\t"
                         "v_pk_add_f32 %12, %12, %9\n\tv_pk_add_f32 %13, %13, %9\n\tv_pk_add_f32 %14, %14, %9\n\tv_pk_add_f32 %15, %15, %9\n\t"
                         "v_pk_add_f32 %8, %8, %9\n\tv_pk_add_f32 %11, %11, %9\n\tv_pk_add_f32 %12, %12, %9\n\tv_pk_add_f32 %13, %13, %9\n\t"
                         "v_pk_add_f32 %14, %14, %9\n\tv_pk_add_f32 %15, %15, %9\n\tv_pk_add_f32 %8, %8, %9\n\tv_pk_add_f32 %11, %11, %9\n\t"
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7), "+v"(p0), "+v"(p1), "+v"(p2),
                           "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7)
                         :
                         : "memory");
        } else if (KIND == 26) {
#define X(n) "v_mul_u32_u24 " R(n) ", " R(n) ", %17\n\t"
            OPS(BLOCK8x8(X));
#undef X
        } else if (KIND == 27) {
#define X(n) "v_mad_u32_u24 " R(n) ", " R(n) ", %17, %18\n\t"
            OPS(BLOCK8x8(X));
#undef X
        } else if (KIND == 28) {  // a VOP2 instruction with a 32-bit literal (8 bytes of code)
#define X(n) "v_add_u32 " R(n) ", 0x12345678, " R(n) "\n\t"
            OPS(BLOCK8x8(X));
#undef X
        } else if (KIND == 29) {  // a two-operand instruction in its VOP3 encoding (8 bytes of code)
#define X(n) "v_add_f32_e64 " R(n) ", " R(n) ", -%17\n\t"
            OPS(BLOCK8x8(X));
#undef X
        } else if (KIND == 30) {
#define X(n) "v_addc_co_u32 " R(n) ", vcc, 0, " R(n) ", vcc\n\t"
            OPS(BLOCK8x8(X));
#undef X
        } else if (KIND == 31) {
#define X(n) "s_add_u32 %16, %16, 1\n\t"
            OPS(BLOCK8x8(X));
#undef X
        } else if (KIND == 32) {
#define X(n) "s_and_b64 vcc, vcc, %20\n\t"
            OPS(BLOCK8x8(X));
#undef X
        } else if (KIND == 33) {
#define X(n) "v_readlane_b32 %16, " R(n) ", 3\n\t"
            OPS(BLOCK8x8(X));
#undef X
        } else if (KIND == 34) {
#define X(n) "ds_write_b32 %19, " R(n) "\n\t"
            OPS(BLOCK8x8(X) "s_waitcnt lgkmcnt(0)\n\t");
#undef X
        } else if (KIND == 35) {
#define X(n) "ds_read_b32 " R(n) ", %19\n\t"
            OPS(BLOCK8x8(X) "s_waitcnt lgkmcnt(0)\n\t");
#undef X
        } else if (KIND == 36) {  // half VALU, half SALU, alternating
#define X(n) "v_add_u32 " R(n) ", " R(n) ", %17\n\ts_add_u32 %16, %16, 1\n\t"
            OPS(REP8(X) REP8(X) REP8(X) REP8(X));
#undef X
        } else if (KIND == 37) {  // 3 VALU : 1 SALU
#define X(n) "v_add_u32 " R(n) ", " R(n) ", %17\n\tv_xor_b32 " R(n) ", " R(n) ", %18\n\tv_add_u32 " R(n) ", " R(n) ", %18\n\ts_add_u32 %16, %16, 1\n\t"
            OPS(REP8(X) REP8(X));
#undef X
        } else if (KIND == 38) {
#define X(n) "v_sub_f32 " R(n) ", " R(n) ", %17\n\t"
            OPS(BLOCK8x8(X));
#undef X
        } else if (KIND == 39) {
#define X(n) "v_max3_f32 " R(n) ", " R(n) ", %17, %18\n\t"
            OPS(BLOCK8x8(X));
#undef X
        } else if (KIND == 40) {
#define X(n) "v_min_i32 " R(n) ", " R(n) ", %17\n\t"
            OPS(BLOCK8x8(X));
#undef X
        } else if (KIND == 41) {
#define X(n) "v_not_b32 " R(n) ", " R(n) "\n\t"
            OPS(BLOCK8x8(X));
#undef X
        } else if (KIND == 42) {  // fma with a negated source (the division's residual)
#define X(n) "v_fma_f32 " R(n) ", -" R(n) ", %17, %18\n\t"
            OPS(BLOCK8x8(X));
#undef X
        } else if (KIND == 43) {  // VOP2 fmac
#define X(n) "v_fmac_f32 " R(n) ", %17, %18\n\t"
            OPS(BLOCK8x8(X));
#undef X
        } else if (KIND == 44) {
#define X(n) "v_alignbit_b32 " R(n) ", " R(n) ", %17, 31\n\t"
            OPS(BLOCK8x8(X));
#undef X
        } else if (KIND == 45) {
#define X(n) "v_perm_b32 " R(n) ", " R(n) ", %17, %18\n\t"
            OPS(BLOCK8x8(X));
#undef X
        } else if (KIND == 46) {
#define X(n) "v_cvt_f32_u32 " R(n) ", " R(n) "\n\t"
            OPS(BLOCK8x8(X));
#undef X
        } else if (KIND == 47) {
#define X(n) "v_xad_u32 " R(n) ", " R(n) ", %17, %18\n\t"
            OPS(BLOCK8x8(X));
#undef X
        } else if (KIND == 48) {
#define X(n) "v_add3_u32 " R(n) ", " R(n) ", %17, %18\n\t"
            OPS(BLOCK8x8(X));
#undef X
        } else if (KIND == 49) {  // SDWA form of a two-operand instruction
#define X(n) "v_and_b32_sdwa " R(n) ", " R(n) ", %17 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n\t"
            OPS(BLOCK8x8(X));
#undef X
        } else if (KIND == 50) {  // DPP form
#define X(n) "v_mov_b32_dpp " R(n) ", " R(n) " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
            OPS(BLOCK8x8(X));
#undef X
        } else if (KIND == 51) {
#define X(n) "v_cndmask_b32_e64 " R(n) ", " R(n) ", %17, vcc\n\t"
            OPS("v_cmp_lt_u32 vcc, %17, %18\n\t" BLOCK8x8(X));
#undef X
        } else if (KIND == 52) {
#define X(n) "v_cndmask_b32 " R(n) ", " R(n) ", %17, vcc\n\tv_add_u32 " R(n) ", " R(n) ", %18\n\t"
            OPS("v_cmp_lt_u32 vcc, %17, %18\n\t" REP8(X) REP8(X) REP8(X) REP8(X));
#undef X
        } else if (KIND == 53) {
#define X(n) "v_cndmask_b32 " R(n) ", " R(n) ", %17, vcc\n\tv_add_u32 " R(n) ", " R(n) ", %18\n\tv_xor_b32 " R(n) ", " R(n) ", %18\n\tv_add_u32 " R(n) ", " R(n) ", %17\n\t"
            OPS("v_cmp_lt_u32 vcc, %17, %18\n\t" REP8(X) REP8(X));
#undef X
        } else if (KIND == 54) {
#define X(n) "v_cndmask_b32 " R(n) ", " R(n) ", %17, %20\n\tv_add_u32 " R(n) ", " R(n) ", %18\n\t"
            OPS(REP8(X) REP8(X) REP8(X) REP8(X));
#undef X
        } else if (KIND == 55) {
#define X(n) "v_bfe_u32 " R(n) ", " R(n) ", %18, 7\n\tv_add_u32 " R(n) ", " R(n) ", %18\n\t"
            OPS(REP8(X) REP8(X) REP8(X) REP8(X));
#undef X
        } else if (KIND == 56) {
#define X(n) "v_bfe_u32 " R(n) ", " R(n) ", %18, 7\n\tv_add_u32 " R(n) ", " R(n) ", %18\n\tv_xor_b32 " R(n) ", " R(n) ", %18\n\tv_add_u32 " R(n) ", " R(n) ", %17\n\t"
            OPS(REP8(X) REP8(X));
#undef X
        } else if (KIND == 57) {
#define X(n) "v_lshlrev_b32 " R(n) ", 3, " R(n) "\n\t"
            OPS(BLOCK8x8(X));
#undef X
        } else if (KIND == 58) {
#define X(n) "v_lshrrev_b32 " R(n) ", %18, " R(n) "\n\t"
            OPS(BLOCK8x8(X));
#undef X
        } else if (KIND == 59) {
#define X(n) "v_ashrrev_i32 " R(n) ", 3, " R(n) "\n\t"
            OPS(BLOCK8x8(X));
#undef X
        } else if (KIND == 60) {
#define X(n) "v_and_b32 " R(n) ", " R(n) ", %17\n\t"
            OPS(BLOCK8x8(X));
#undef X
        } else if (KIND == 61) {
#define X(n) "v_or_b32 " R(n) ", " R(n) ", %17\n\t"
            OPS(BLOCK8x8(X));
#undef X
        } else if (KIND == 62) {
#define X(n) "v_sub_u32 " R(n) ", " R(n) ", %17\n\t"
            OPS(BLOCK8x8(X));
#undef X
        } else if (KIND == 63) {
#define X(n) "v_max_f32 " R(n) ", " R(n) ", %17\n\t"
            OPS(BLOCK8x8(X));
#undef X
        } else if (KIND == 64) {
#define X(n) "v_max_u32 " R(n) ", " R(n) ", %17\n\t"
            OPS(BLOCK8x8(X));
#undef X
        } else if (KIND == 65) {
#define X(n) "v_med3_f32 " R(n) ", " R(n) ", %17, %18\n\t"
            OPS(BLOCK8x8(X));
#undef X
        } else if (KIND == 66) {
#define X(n) "v_cmp_lt_u32 vcc, " R(n) ", %17\n\t"
            OPS(BLOCK8x8(X));
#undef X
        } else if (KIND == 67) {
#define X(n) "v_cmp_class_f32 vcc, " R(n) ", %17\n\t"
            OPS(BLOCK8x8(X));
#undef X
        } else if (KIND == 68) {
#define X(n) "v_mul_lo_u32 " R(n) ", " R(n) ", %17\n\t"
            OPS(BLOCK8x8(X));
#undef X
        } else if (KIND == 69) {
#define X(n) "v_and_b32 " R(n) ", 0x12345678, " R(n) "\n\t"
            OPS(BLOCK8x8(X));
#undef X
        } else if (KIND == 70) {
#define X(n) "v_add_f32 " R(n) ", %22, " R(n) "\n\t"
            OPS(BLOCK8x8(X));
#undef X
        } else if (KIND == 71) {
#define X(n) "v_mul_f32 " R(n) ", 2.0, " R(n) "\n\t"
            OPS(BLOCK8x8(X));
#undef X
        } else if (KIND == 72) {
#define X(n) "v_xor_b32 " R(n) ", %22, " R(n) "\n\t"
            OPS(BLOCK8x8(X));
#undef X
        } else if (KIND == 73) {
#define X(n) "v_sub_f32_e64 " R(n) ", |" R(n) "|, -%17\n\t"
            OPS(BLOCK8x8(X));
#undef X
        } else if (KIND == 74) {
#define X(n) "v_mul_f32 " R(n) ", " R(n) ", %17\n\tv_bfe_u32 " R(n) ", " R(n) ", %18, 9\n\tv_add_f32 " R(n) ", " R(n) ", %17\n\tv_cmp_eq_f32 vcc, " R(n) ", %17\n\t"
            OPS(REP8(X) REP8(X));
#undef X
        } else if (KIND == 75) {
#define X(n) "v_cvt_f32_ubyte0 " R(n) ", " R(n) "\n\t"
            OPS(BLOCK8x8(X));
#undef X
        } else if (KIND == 76) {
#define X(n) "v_mbcnt_lo_u32_b32 " R(n) ", %17, " R(n) "\n\t"
            OPS(BLOCK8x8(X));
#undef X
        } else if (KIND == 77) {
#define X(n) "v_subrev_u32 " R(n) ", %17, " R(n) "\n\t"
            OPS(BLOCK8x8(X));
#undef X
        } else if (KIND == 78) {
#define X(n) "v_lshlrev_b32 " R(n) ", %22, " R(n) "\n\t"
            OPS(BLOCK8x8(X));
#undef X
        } else if (KIND == 79) {
#define X(n) "v_mul_f32 " R(n) ", %17, %18\n\t"
            OPS(BLOCK8x8(X));
#undef X
        } else if (KIND == 80) {
            OPS("v_bfe_u32 %0, %0, %18, 7\n\tv_add_u32 %1, %1, %18\n\tv_bfe_u32 %2, %2, %18, 7\n\tv_add_u32 %3, %3, %18\n\tv_bfe_u32 %4, %4, %18, 7\n\tv_add_u32 %5, %5, %18\n\tv_bfe_u32 %6, %6, %18, 7\n\tv_add_u32 %7, %7, %18\n\tv_bfe_u32 %0, %0, %18, 7\n\tv_add_u32 %1, %1, %18\n\tv_bfe_u32 %2, %2, %18, 7\n\tv_add_u32 %3, %3, %18\n\tv_bfe_u32 %4, %4, %18, 7\n\tv_add_u32 %5, %5, %18\n\tv_bfe_u32 %6, %6, %18, 7\n\tv_add_u32 %7, %7, %18\n\tv_bfe_u32 %0, %0, %18, 7\n\tv_add_u32 %1, %1, %18\n\tv_bfe_u32 %2, %2, %18, 7\n\tv_add_u32 %3, %3, %18\n\tv_bfe_u32 %4, %4, %18, 7\n\tv_add_u32 %5, %5, %18\n\tv_bfe_u32 %6, %6, %18, 7\n\tv_add_u32 %7, %7, %18\n\tv_bfe_u32 %0, %0, %18, 7\n\tv_add_u32 %1, %1, %18\n\tv_bfe_u32 %2, %2, %18, 7\n\tv_add_u32 %3, %3, %18\n\tv_bfe_u32 %4, %4, %18, 7\n\tv_add_u32 %5, %5, %18\n\tv_bfe_u32 %6, %6, %18, 7\n\tv_add_u32 %7, %7, %18\n\tv_bfe_u32 %0, %0, %18, 7\n\tv_add_u32 %1, %1, %18\n\tv_bfe_u32 %2, %2, %18, 7\n\tv_add_u32 %3, %3, %18\n\tv_bfe_u32 %4, %4, %18, 7\n\tv_add_u32 %5, %5, %18\n\tv_bfe_u32 %6, %6, %18, 7\n\tv_add_u32 %7, %7, %18\n\tv_bfe_u32 %0, %0, %18, 7\n\tv_add_u32 %1, %1, %18\n\tv_bfe_u32 %2, %2, %18, 7\n\tv_add_u32 %3, %3, %18\n\tv_bfe_u32 %4, %4, %18, 7\n\tv_add_u32 %5, %5, %18\n\tv_bfe_u32 %6, %6, %18, 7\n\tv_add_u32 %7, %7, %18\n\tv_bfe_u32 %0, %0, %18, 7\n\tv_add_u32 %1, %1, %18\n\tv_bfe_u32 %2, %2, %18, 7\n\tv_add_u32 %3, %3, %18\n\tv_bfe_u32 %4, %4, %18, 7\n\tv_add_u32 %5, %5, %18\n\tv_bfe_u32 %6, %6, %18, 7\n\tv_add_u32 %7, %7, %18\n\tv_bfe_u32 %0, %0, %18, 7\n\tv_add_u32 %1, %1, %18\n\tv_bfe_u32 %2, %2, %18, 7\n\tv_add_u32 %3, %3, %18\n\tv_bfe_u32 %4, %4, %18, 7\n\tv_add_u32 %5, %5, %18\n\tv_bfe_u32 %6, %6, %18, 7\n\tv_add_u32 %7, %7, %18\n\t");
        } else if (KIND == 81) {
            OPS("v_bfe_u32 %0, %0, %18, 7\n\tv_add_u32 %1, %1, %18\n\tv_xor_b32 %2, %2, %18\n\tv_add_u32 %3, %3, %18\n\tv_bfe_u32 %4, %4, %18, 7\n\tv_add_u32 %5, %5, %18\n\tv_xor_b32 %6, %6, %18\n\tv_add_u32 %7, %7, %18\n\tv_bfe_u32 %0, %0, %18, 7\n\tv_add_u32 %1, %1, %18\n\tv_xor_b32 %2, %2, %18\n\tv_add_u32 %3, %3, %18\n\tv_bfe_u32 %4, %4, %18, 7\n\tv_add_u32 %5, %5, %18\n\tv_xor_b32 %6, %6, %18\n\tv_add_u32 %7, %7, %18\n\tv_bfe_u32 %0, %0, %18, 7\n\tv_add_u32 %1, %1, %18\n\tv_xor_b32 %2, %2, %18\n\tv_add_u32 %3, %3, %18\n\tv_bfe_u32 %4, %4, %18, 7\n\tv_add_u32 %5, %5, %18\n\tv_xor_b32 %6, %6, %18\n\tv_add_u32 %7, %7, %18\n\tv_bfe_u32 %0, %0, %18, 7\n\tv_add_u32 %1, %1, %18\n\tv_xor_b32 %2, %2, %18\n\tv_add_u32 %3, %3, %18\n\tv_bfe_u32 %4, %4, %18, 7\n\tv_add_u32 %5, %5, %18\n\tv_xor_b32 %6, %6, %18\n\tv_add_u32 %7, %7, %18\n\tv_bfe_u32 %0, %0, %18, 7\n\tv_add_u32 %1, %1, %18\n\tv_xor_b32 %2, %2, %18\n\tv_add_u32 %3, %3, %18\n\tv_bfe_u32 %4, %4, %18, 7\n\tv_add_u32 %5, %5, %18\n\tv_xor_b32 %6, %6, %18\n\tv_add_u32 %7, %7, %18\n\tv_bfe_u32 %0, %0, %18, 7\n\tv_add_u32 %1, %1, %18\n\tv_xor_b32 %2, %2, %18\n\tv_add_u32 %3, %3, %18\n\tv_bfe_u32 %4, %4, %18, 7\n\tv_add_u32 %5, %5, %18\n\tv_xor_b32 %6, %6, %18\n\tv_add_u32 %7, %7, %18\n\tv_bfe_u32 %0, %0, %18, 7\n\tv_add_u32 %1, %1, %18\n\tv_xor_b32 %2, %2, %18\n\tv_add_u32 %3, %3, %18\n\tv_bfe_u32 %4, %4, %18, 7\n\tv_add_u32 %5, %5, %18\n\tv_xor_b32 %6, %6, %18\n\tv_add_u32 %7, %7, %18\n\tv_bfe_u32 %0, %0, %18, 7\n\tv_add_u32 %1, %1, %18\n\tv_xor_b32 %2, %2, %18\n\tv_add_u32 %3, %3, %18\n\tv_bfe_u32 %4, %4, %18, 7\n\tv_add_u32 %5, %5, %18\n\tv_xor_b32 %6, %6, %18\n\tv_add_u32 %7, %7, %18\n\t");
        } else if (KIND == 82) {
            OPS("v_bfe_u32 %0, %0, %18, 7\n\tv_bfe_u32 %1, %1, %18, 7\n\tv_add_u32 %2, %2, %18\n\tv_add_u32 %3, %3, %18\n\tv_bfe_u32 %4, %4, %18, 7\n\tv_bfe_u32 %5, %5, %18, 7\n\tv_add_u32 %6, %6, %18\n\tv_add_u32 %7, %7, %18\n\tv_bfe_u32 %0, %0, %18, 7\n\tv_bfe_u32 %1, %1, %18, 7\n\tv_add_u32 %2, %2, %18\n\tv_add_u32 %3, %3, %18\n\tv_bfe_u32 %4, %4, %18, 7\n\tv_bfe_u32 %5, %5, %18, 7\n\tv_add_u32 %6, %6, %18\n\tv_add_u32 %7, %7, %18\n\tv_bfe_u32 %0, %0, %18, 7\n\tv_bfe_u32 %1, %1, %18, 7\n\tv_add_u32 %2, %2, %18\n\tv_add_u32 %3, %3, %18\n\tv_bfe_u32 %4, %4, %18, 7\n\tv_bfe_u32 %5, %5, %18, 7\n\tv_add_u32 %6, %6, %18\n\tv_add_u32 %7, %7, %18\n\tv_bfe_u32 %0, %0, %18, 7\n\tv_bfe_u32 %1, %1, %18, 7\n\tv_add_u32 %2, %2, %18\n\tv_add_u32 %3, %3, %18\n\tv_bfe_u32 %4, %4, %18, 7\n\tv_bfe_u32 %5, %5, %18, 7\n\tv_add_u32 %6, %6, %18\n\tv_add_u32 %7, %7, %18\n\tv_bfe_u32 %0, %0, %18, 7\n\tv_bfe_u32 %1, %1, %18, 7\n\tv_add_u32 %2, %2, %18\n\tv_add_u32 %3, %3, %18\n\tv_bfe_u32 %4, %4, %18, 7\n\tv_bfe_u32 %5, %5, %18, 7\n\tv_add_u32 %6, %6, %18\n\tv_add_u32 %7, %7, %18\n\tv_bfe_u32 %0, %0, %18, 7\n\tv_bfe_u32 %1, %1, %18, 7\n\tv_add_u32 %2, %2, %18\n\tv_add_u32 %3, %3, %18\n\tv_bfe_u32 %4, %4, %18, 7\n\tv_bfe_u32 %5, %5, %18, 7\n\tv_add_u32 %6, %6, %18\n\tv_add_u32 %7, %7, %18\n\tv_bfe_u32 %0, %0, %18, 7\n\tv_bfe_u32 %1, %1, %18, 7\n\tv_add_u32 %2, %2, %18\n\tv_add_u32 %3, %3, %18\n\tv_bfe_u32 %4, %4, %18, 7\n\tv_bfe_u32 %5, %5, %18, 7\n\tv_add_u32 %6, %6, %18\n\tv_add_u32 %7, %7, %18\n\tv_bfe_u32 %0, %0, %18, 7\n\tv_bfe_u32 %1, %1, %18, 7\n\tv_add_u32 %2, %2, %18\n\tv_add_u32 %3, %3, %18\n\tv_bfe_u32 %4, %4, %18, 7\n\tv_bfe_u32 %5, %5, %18, 7\n\tv_add_u32 %6, %6, %18\n\tv_add_u32 %7, %7, %18\n\t");
        } else if (KIND == 83) {
            OPS("v_bfe_u32 %0, %0, %18, 7\n\tv_add_f32 %1, %1, %17\n\tv_bfe_u32 %2, %2, %18, 7\n\tv_add_f32 %3, %3, %17\n\tv_bfe_u32 %4, %4, %18, 7\n\tv_add_f32 %5, %5, %17\n\tv_bfe_u32 %6, %6, %18, 7\n\tv_add_f32 %7, %7, %17\n\tv_bfe_u32 %0, %0, %18, 7\n\tv_add_f32 %1, %1, %17\n\tv_bfe_u32 %2, %2, %18, 7\n\tv_add_f32 %3, %3, %17\n\tv_bfe_u32 %4, %4, %18, 7\n\tv_add_f32 %5, %5, %17\n\tv_bfe_u32 %6, %6, %18, 7\n\tv_add_f32 %7, %7, %17\n\tv_bfe_u32 %0, %0, %18, 7\n\tv_add_f32 %1, %1, %17\n\tv_bfe_u32 %2, %2, %18, 7\n\tv_add_f32 %3, %3, %17\n\tv_bfe_u32 %4, %4, %18, 7\n\tv_add_f32 %5, %5, %17\n\tv_bfe_u32 %6, %6, %18, 7\n\tv_add_f32 %7, %7, %17\n\tv_bfe_u32 %0, %0, %18, 7\n\tv_add_f32 %1, %1, %17\n\tv_bfe_u32 %2, %2, %18, 7\n\tv_add_f32 %3, %3, %17\n\tv_bfe_u32 %4, %4, %18, 7\n\tv_add_f32 %5, %5, %17\n\tv_bfe_u32 %6, %6, %18, 7\n\tv_add_f32 %7, %7, %17\n\tv_bfe_u32 %0, %0, %18, 7\n\tv_add_f32 %1, %1, %17\n\tv_bfe_u32 %2, %2, %18, 7\n\tv_add_f32 %3, %3, %17\n\tv_bfe_u32 %4, %4, %18, 7\n\tv_add_f32 %5, %5, %17\n\tv_bfe_u32 %6, %6, %18, 7\n\tv_add_f32 %7, %7, %17\n\tv_bfe_u32 %0, %0, %18, 7\n\tv_add_f32 %1, %1, %17\n\tv_bfe_u32 %2, %2, %18, 7\n\tv_add_f32 %3, %3, %17\n\tv_bfe_u32 %4, %4, %18, 7\n\tv_add_f32 %5, %5, %17\n\tv_bfe_u32 %6, %6, %18, 7\n\tv_add_f32 %7, %7, %17\n\tv_bfe_u32 %0, %0, %18, 7\n\tv_add_f32 %1, %1, %17\n\tv_bfe_u32 %2, %2, %18, 7\n\tv_add_f32 %3, %3, %17\n\tv_bfe_u32 %4, %4, %18, 7\n\tv_add_f32 %5, %5, %17\n\tv_bfe_u32 %6, %6, %18, 7\n\tv_add_f32 %7, %7, %17\n\tv_bfe_u32 %0, %0, %18, 7\n\tv_add_f32 %1, %1, %17\n\tv_bfe_u32 %2, %2, %18, 7\n\tv_add_f32 %3, %3, %17\n\tv_bfe_u32 %4, %4, %18, 7\n\tv_add_f32 %5, %5, %17\n\tv_bfe_u32 %6, %6, %18, 7\n\tv_add_f32 %7, %7, %17\n\t");
        } else if (KIND == 84) {
            OPS("v_bfe_u32 %0, %0, %18, 7\n\tv_mul_f32 %1, %1, %17\n\tv_add_f32 %2, %2, %17\n\tv_fma_f32 %3, %3, %17, %18\n\tv_bfe_u32 %4, %4, %18, 7\n\tv_mul_f32 %5, %5, %17\n\tv_add_f32 %6, %6, %17\n\tv_fma_f32 %7, %7, %17, %18\n\tv_bfe_u32 %0, %0, %18, 7\n\tv_mul_f32 %1, %1, %17\n\tv_add_f32 %2, %2, %17\n\tv_fma_f32 %3, %3, %17, %18\n\tv_bfe_u32 %4, %4, %18, 7\n\tv_mul_f32 %5, %5, %17\n\tv_add_f32 %6, %6, %17\n\tv_fma_f32 %7, %7, %17, %18\n\tv_bfe_u32 %0, %0, %18, 7\n\tv_mul_f32 %1, %1, %17\n\tv_add_f32 %2, %2, %17\n\tv_fma_f32 %3, %3, %17, %18\n\tv_bfe_u32 %4, %4, %18, 7\n\tv_mul_f32 %5, %5, %17\n\tv_add_f32 %6, %6, %17\n\tv_fma_f32 %7, %7, %17, %18\n\tv_bfe_u32 %0, %0, %18, 7\n\tv_mul_f32 %1, %1, %17\n\tv_add_f32 %2, %2, %17\n\tv_fma_f32 %3, %3, %17, %18\n\tv_bfe_u32 %4, %4, %18, 7\n\tv_mul_f32 %5, %5, %17\n\tv_add_f32 %6, %6, %17\n\tv_fma_f32 %7, %7, %17, %18\n\tv_bfe_u32 %0, %0, %18, 7\n\tv_mul_f32 %1, %1, %17\n\tv_add_f32 %2, %2, %17\n\tv_fma_f32 %3, %3, %17, %18\n\tv_bfe_u32 %4, %4, %18, 7\n\tv_mul_f32 %5, %5, %17\n\tv_add_f32 %6, %6, %17\n\tv_fma_f32 %7, %7, %17, %18\n\tv_bfe_u32 %0, %0, %18, 7\n\tv_mul_f32 %1, %1, %17\n\tv_add_f32 %2, %2, %17\n\tv_fma_f32 %3, %3, %17, %18\n\tv_bfe_u32 %4, %4, %18, 7\n\tv_mul_f32 %5, %5, %17\n\tv_add_f32 %6, %6, %17\n\tv_fma_f32 %7, %7, %17, %18\n\tv_bfe_u32 %0, %0, %18, 7\n\tv_mul_f32 %1, %1, %17\n\tv_add_f32 %2, %2, %17\n\tv_fma_f32 %3, %3, %17, %18\n\tv_bfe_u32 %4, %4, %18, 7\n\tv_mul_f32 %5, %5, %17\n\tv_add_f32 %6, %6, %17\n\tv_fma_f32 %7, %7, %17, %18\n\tv_bfe_u32 %0, %0, %18, 7\n\tv_mul_f32 %1, %1, %17\n\tv_add_f32 %2, %2, %17\n\tv_fma_f32 %3, %3, %17, %18\n\tv_bfe_u32 %4, %4, %18, 7\n\tv_mul_f32 %5, %5, %17\n\tv_add_f32 %6, %6, %17\n\tv_fma_f32 %7, %7, %17, %18\n\t");
        } else if (KIND == 85) {
            OPS("v_mul_f32 %0, %0, %17\n\tv_bfe_u32 %1, %1, %18, 7\n\tv_add_f32 %2, %2, %17\n\tv_cmp_eq_f32 vcc, %3, %17\n\tv_mul_f32 %4, %4, %17\n\tv_bfe_u32 %5, %5, %18, 7\n\tv_add_f32 %6, %6, %17\n\tv_cmp_eq_f32 vcc, %7, %17\n\tv_mul_f32 %0, %0, %17\n\tv_bfe_u32 %1, %1, %18, 7\n\tv_add_f32 %2, %2, %17\n\tv_cmp_eq_f32 vcc, %3, %17\n\tv_mul_f32 %4, %4, %17\n\tv_bfe_u32 %5, %5, %18, 7\n\tv_add_f32 %6, %6, %17\n\tv_cmp_eq_f32 vcc, %7, %17\n\tv_mul_f32 %0, %0, %17\n\tv_bfe_u32 %1, %1, %18, 7\n\tv_add_f32 %2, %2, %17\n\tv_cmp_eq_f32 vcc, %3, %17\n\tv_mul_f32 %4, %4, %17\n\tv_bfe_u32 %5, %5, %18, 7\n\tv_add_f32 %6, %6, %17\n\tv_cmp_eq_f32 vcc, %7, %17\n\tv_mul_f32 %0, %0, %17\n\tv_bfe_u32 %1, %1, %18, 7\n\tv_add_f32 %2, %2, %17\n\tv_cmp_eq_f32 vcc, %3, %17\n\tv_mul_f32 %4, %4, %17\n\tv_bfe_u32 %5, %5, %18, 7\n\tv_add_f32 %6, %6, %17\n\tv_cmp_eq_f32 vcc, %7, %17\n\tv_mul_f32 %0, %0, %17\n\tv_bfe_u32 %1, %1, %18, 7\n\tv_add_f32 %2, %2, %17\n\tv_cmp_eq_f32 vcc, %3, %17\n\tv_mul_f32 %4, %4, %17\n\tv_bfe_u32 %5, %5, %18, 7\n\tv_add_f32 %6, %6, %17\n\tv_cmp_eq_f32 vcc, %7, %17\n\tv_mul_f32 %0, %0, %17\n\tv_bfe_u32 %1, %1, %18, 7\n\tv_add_f32 %2, %2, %17\n\tv_cmp_eq_f32 vcc, %3, %17\n\tv_mul_f32 %4, %4, %17\n\tv_bfe_u32 %5, %5, %18, 7\n\tv_add_f32 %6, %6, %17\n\tv_cmp_eq_f32 vcc, %7, %17\n\tv_mul_f32 %0, %0, %17\n\tv_bfe_u32 %1, %1, %18, 7\n\tv_add_f32 %2, %2, %17\n\tv_cmp_eq_f32 vcc, %3, %17\n\tv_mul_f32 %4, %4, %17\n\tv_bfe_u32 %5, %5, %18, 7\n\tv_add_f32 %6, %6, %17\n\tv_cmp_eq_f32 vcc, %7, %17\n\tv_mul_f32 %0, %0, %17\n\tv_bfe_u32 %1, %1, %18, 7\n\tv_add_f32 %2, %2, %17\n\tv_cmp_eq_f32 vcc, %3, %17\n\tv_mul_f32 %4, %4, %17\n\tv_bfe_u32 %5, %5, %18, 7\n\tv_add_f32 %6, %6, %17\n\tv_cmp_eq_f32 vcc, %7, %17\n\t");
        } else if (KIND == 86) {
            OPS("v_cmp_eq_f32 vcc, %0, %17\n\tv_add_f32 %1, %1, %17\n\tv_cmp_eq_f32 vcc, %2, %17\n\tv_add_f32 %3, %3, %17\n\tv_cmp_eq_f32 vcc, %4, %17\n\tv_add_f32 %5, %5, %17\n\tv_cmp_eq_f32 vcc, %6, %17\n\tv_add_f32 %7, %7, %17\n\tv_cmp_eq_f32 vcc, %0, %17\n\tv_add_f32 %1, %1, %17\n\tv_cmp_eq_f32 vcc, %2, %17\n\tv_add_f32 %3, %3, %17\n\tv_cmp_eq_f32 vcc, %4, %17\n\tv_add_f32 %5, %5, %17\n\tv_cmp_eq_f32 vcc, %6, %17\n\tv_add_f32 %7, %7, %17\n\tv_cmp_eq_f32 vcc, %0, %17\n\tv_add_f32 %1, %1, %17\n\tv_cmp_eq_f32 vcc, %2, %17\n\tv_add_f32 %3, %3, %17\n\tv_cmp_eq_f32 vcc, %4, %17\n\tv_add_f32 %5, %5, %17\n\tv_cmp_eq_f32 vcc, %6, %17\n\tv_add_f32 %7, %7, %17\n\tv_cmp_eq_f32 vcc, %0, %17\n\tv_add_f32 %1, %1, %17\n\tv_cmp_eq_f32 vcc, %2, %17\n\tv_add_f32 %3, %3, %17\n\tv_cmp_eq_f32 vcc, %4, %17\n\tv_add_f32 %5, %5, %17\n\tv_cmp_eq_f32 vcc, %6, %17\n\tv_add_f32 %7, %7, %17\n\tv_cmp_eq_f32 vcc, %0, %17\n\tv_add_f32 %1, %1, %17\n\tv_cmp_eq_f32 vcc, %2, %17\n\tv_add_f32 %3, %3, %17\n\tv_cmp_eq_f32 vcc, %4, %17\n\tv_add_f32 %5, %5, %17\n\tv_cmp_eq_f32 vcc, %6, %17\n\tv_add_f32 %7, %7, %17\n\tv_cmp_eq_f32 vcc, %0, %17\n\tv_add_f32 %1, %1, %17\n\tv_cmp_eq_f32 vcc, %2, %17\n\tv_add_f32 %3, %3, %17\n\tv_cmp_eq_f32 vcc, %4, %17\n\tv_add_f32 %5, %5, %17\n\tv_cmp_eq_f32 vcc, %6, %17\n\tv_add_f32 %7, %7, %17\n\tv_cmp_eq_f32 vcc, %0, %17\n\tv_add_f32 %1, %1, %17\n\tv_cmp_eq_f32 vcc, %2, %17\n\tv_add_f32 %3, %3, %17\n\tv_cmp_eq_f32 vcc, %4, %17\n\tv_add_f32 %5, %5, %17\n\tv_cmp_eq_f32 vcc, %6, %17\n\tv_add_f32 %7, %7, %17\n\tv_cmp_eq_f32 vcc, %0, %17\n\tv_add_f32 %1, %1, %17\n\tv_cmp_eq_f32 vcc, %2, %17\n\tv_add_f32 %3, %3, %17\n\tv_cmp_eq_f32 vcc, %4, %17\n\tv_add_f32 %5, %5, %17\n\tv_cmp_eq_f32 vcc, %6, %17\n\tv_add_f32 %7, %7, %17\n\t");
        } else if (KIND == 87) {
            OPS("v_cmp_eq_f32 vcc, %0, %17\n\tv_add_u32 %1, %1, %18\n\tv_cmp_eq_f32 vcc, %2, %17\n\tv_add_u32 %3, %3, %18\n\tv_cmp_eq_f32 vcc, %4, %17\n\tv_add_u32 %5, %5, %18\n\tv_cmp_eq_f32 vcc, %6, %17\n\tv_add_u32 %7, %7, %18\n\tv_cmp_eq_f32 vcc, %0, %17\n\tv_add_u32 %1, %1, %18\n\tv_cmp_eq_f32 vcc, %2, %17\n\tv_add_u32 %3, %3, %18\n\tv_cmp_eq_f32 vcc, %4, %17\n\tv_add_u32 %5, %5, %18\n\tv_cmp_eq_f32 vcc, %6, %17\n\tv_add_u32 %7, %7, %18\n\tv_cmp_eq_f32 vcc, %0, %17\n\tv_add_u32 %1, %1, %18\n\tv_cmp_eq_f32 vcc, %2, %17\n\tv_add_u32 %3, %3, %18\n\tv_cmp_eq_f32 vcc, %4, %17\n\tv_add_u32 %5, %5, %18\n\tv_cmp_eq_f32 vcc, %6, %17\n\tv_add_u32 %7, %7, %18\n\tv_cmp_eq_f32 vcc, %0, %17\n\tv_add_u32 %1, %1, %18\n\tv_cmp_eq_f32 vcc, %2, %17\n\tv_add_u32 %3, %3, %18\n\tv_cmp_eq_f32 vcc, %4, %17\n\tv_add_u32 %5, %5, %18\n\tv_cmp_eq_f32 vcc, %6, %17\n\tv_add_u32 %7, %7, %18\n\tv_cmp_eq_f32 vcc, %0, %17\n\tv_add_u32 %1, %1, %18\n\tv_cmp_eq_f32 vcc, %2, %17\n\tv_add_u32 %3, %3, %18\n\tv_cmp_eq_f32 vcc, %4, %17\n\tv_add_u32 %5, %5, %18\n\tv_cmp_eq_f32 vcc, %6, %17\n\tv_add_u32 %7, %7, %18\n\tv_cmp_eq_f32 vcc, %0, %17\n\tv_add_u32 %1, %1, %18\n\tv_cmp_eq_f32 vcc, %2, %17\n\tv_add_u32 %3, %3, %18\n\tv_cmp_eq_f32 vcc, %4, %17\n\tv_add_u32 %5, %5, %18\n\tv_cmp_eq_f32 vcc, %6, %17\n\tv_add_u32 %7, %7, %18\n\tv_cmp_eq_f32 vcc, %0, %17\n\tv_add_u32 %1, %1, %18\n\tv_cmp_eq_f32 vcc, %2, %17\n\tv_add_u32 %3, %3, %18\n\tv_cmp_eq_f32 vcc, %4, %17\n\tv_add_u32 %5, %5, %18\n\tv_cmp_eq_f32 vcc, %6, %17\n\tv_add_u32 %7, %7, %18\n\tv_cmp_eq_f32 vcc, %0, %17\n\tv_add_u32 %1, %1, %18\n\tv_cmp_eq_f32 vcc, %2, %17\n\tv_add_u32 %3, %3, %18\n\tv_cmp_eq_f32 vcc, %4, %17\n\tv_add_u32 %5, %5, %18\n\tv_cmp_eq_f32 vcc, %6, %17\n\tv_add_u32 %7, %7, %18\n\t");
        } else if (KIND == 88) {
            OPS("v_cvt_f32_i32 %0, %0\n\tv_add_f32 %1, %1, %17\n\tv_cvt_f32_i32 %2, %2\n\tv_add_f32 %3, %3, %17\n\tv_cvt_f32_i32 %4, %4\n\tv_add_f32 %5, %5, %17\n\tv_cvt_f32_i32 %6, %6\n\tv_add_f32 %7, %7, %17\n\tv_cvt_f32_i32 %0, %0\n\tv_add_f32 %1, %1, %17\n\tv_cvt_f32_i32 %2, %2\n\tv_add_f32 %3, %3, %17\n\tv_cvt_f32_i32 %4, %4\n\tv_add_f32 %5, %5, %17\n\tv_cvt_f32_i32 %6, %6\n\tv_add_f32 %7, %7, %17\n\tv_cvt_f32_i32 %0, %0\n\tv_add_f32 %1, %1, %17\n\tv_cvt_f32_i32 %2, %2\n\tv_add_f32 %3, %3, %17\n\tv_cvt_f32_i32 %4, %4\n\tv_add_f32 %5, %5, %17\n\tv_cvt_f32_i32 %6, %6\n\tv_add_f32 %7, %7, %17\n\tv_cvt_f32_i32 %0, %0\n\tv_add_f32 %1, %1, %17\n\tv_cvt_f32_i32 %2, %2\n\tv_add_f32 %3, %3, %17\n\tv_cvt_f32_i32 %4, %4\n\tv_add_f32 %5, %5, %17\n\tv_cvt_f32_i32 %6, %6\n\tv_add_f32 %7, %7, %17\n\tv_cvt_f32_i32 %0, %0\n\tv_add_f32 %1, %1, %17\n\tv_cvt_f32_i32 %2, %2\n\tv_add_f32 %3, %3, %17\n\tv_cvt_f32_i32 %4, %4\n\tv_add_f32 %5, %5, %17\n\tv_cvt_f32_i32 %6, %6\n\tv_add_f32 %7, %7, %17\n\tv_cvt_f32_i32 %0, %0\n\tv_add_f32 %1, %1, %17\n\tv_cvt_f32_i32 %2, %2\n\tv_add_f32 %3, %3, %17\n\tv_cvt_f32_i32 %4, %4\n\tv_add_f32 %5, %5, %17\n\tv_cvt_f32_i32 %6, %6\n\tv_add_f32 %7, %7, %17\n\tv_cvt_f32_i32 %0, %0\n\tv_add_f32 %1, %1, %17\n\tv_cvt_f32_i32 %2, %2\n\tv_add_f32 %3, %3, %17\n\tv_cvt_f32_i32 %4, %4\n\tv_add_f32 %5, %5, %17\n\tv_cvt_f32_i32 %6, %6\n\tv_add_f32 %7, %7, %17\n\tv_cvt_f32_i32 %0, %0\n\tv_add_f32 %1, %1, %17\n\tv_cvt_f32_i32 %2, %2\n\tv_add_f32 %3, %3, %17\n\tv_cvt_f32_i32 %4, %4\n\tv_add_f32 %5, %5, %17\n\tv_cvt_f32_i32 %6, %6\n\tv_add_f32 %7, %7, %17\n\t");
        } else if (KIND == 89) {
            OPS("v_min_f32 %0, %0, %17\n\tv_add_f32 %1, %1, %17\n\tv_min_f32 %2, %2, %17\n\tv_add_f32 %3, %3, %17\n\tv_min_f32 %4, %4, %17\n\tv_add_f32 %5, %5, %17\n\tv_min_f32 %6, %6, %17\n\tv_add_f32 %7, %7, %17\n\tv_min_f32 %0, %0, %17\n\tv_add_f32 %1, %1, %17\n\tv_min_f32 %2, %2, %17\n\tv_add_f32 %3, %3, %17\n\tv_min_f32 %4, %4, %17\n\tv_add_f32 %5, %5, %17\n\tv_min_f32 %6, %6, %17\n\tv_add_f32 %7, %7, %17\n\tv_min_f32 %0, %0, %17\n\tv_add_f32 %1, %1, %17\n\tv_min_f32 %2, %2, %17\n\tv_add_f32 %3, %3, %17\n\tv_min_f32 %4, %4, %17\n\tv_add_f32 %5, %5, %17\n\tv_min_f32 %6, %6, %17\n\tv_add_f32 %7, %7, %17\n\tv_min_f32 %0, %0, %17\n\tv_add_f32 %1, %1, %17\n\tv_min_f32 %2, %2, %17\n\tv_add_f32 %3, %3, %17\n\tv_min_f32 %4, %4, %17\n\tv_add_f32 %5, %5, %17\n\tv_min_f32 %6, %6, %17\n\tv_add_f32 %7, %7, %17\n\tv_min_f32 %0, %0, %17\n\tv_add_f32 %1, %1, %17\n\tv_min_f32 %2, %2, %17\n\tv_add_f32 %3, %3, %17\n\tv_min_f32 %4, %4, %17\n\tv_add_f32 %5, %5, %17\n\tv_min_f32 %6, %6, %17\n\tv_add_f32 %7, %7, %17\n\tv_min_f32 %0, %0, %17\n\tv_add_f32 %1, %1, %17\n\tv_min_f32 %2, %2, %17\n\tv_add_f32 %3, %3, %17\n\tv_min_f32 %4, %4, %17\n\tv_add_f32 %5, %5, %17\n\tv_min_f32 %6, %6, %17\n\tv_add_f32 %7, %7, %17\n\tv_min_f32 %0, %0, %17\n\tv_add_f32 %1, %1, %17\n\tv_min_f32 %2, %2, %17\n\tv_add_f32 %3, %3, %17\n\tv_min_f32 %4, %4, %17\n\tv_add_f32 %5, %5, %17\n\tv_min_f32 %6, %6, %17\n\tv_add_f32 %7, %7, %17\n\tv_min_f32 %0, %0, %17\n\tv_add_f32 %1, %1, %17\n\tv_min_f32 %2, %2, %17\n\tv_add_f32 %3, %3, %17\n\tv_min_f32 %4, %4, %17\n\tv_add_f32 %5, %5, %17\n\tv_min_f32 %6, %6, %17\n\tv_add_f32 %7, %7, %17\n\t");
        } else if (KIND == 90) {
            OPS("v_lshl_or_b32 %0, %0, 1, %17\n\tv_add_u32 %1, %1, %18\n\tv_add_u32 %2, %2, %18\n\tv_lshl_or_b32 %3, %3, 1, %17\n\tv_add_u32 %4, %4, %18\n\tv_add_u32 %5, %5, %18\n\tv_lshl_or_b32 %6, %6, 1, %17\n\tv_add_u32 %7, %7, %18\n\tv_add_u32 %0, %0, %18\n\tv_lshl_or_b32 %1, %1, 1, %17\n\tv_add_u32 %2, %2, %18\n\tv_add_u32 %3, %3, %18\n\tv_lshl_or_b32 %4, %4, 1, %17\n\tv_add_u32 %5, %5, %18\n\tv_add_u32 %6, %6, %18\n\tv_lshl_or_b32 %7, %7, 1, %17\n\tv_add_u32 %0, %0, %18\n\tv_add_u32 %1, %1, %18\n\tv_lshl_or_b32 %2, %2, 1, %17\n\tv_add_u32 %3, %3, %18\n\tv_add_u32 %4, %4, %18\n\tv_lshl_or_b32 %5, %5, 1, %17\n\tv_add_u32 %6, %6, %18\n\tv_add_u32 %7, %7, %18\n\tv_lshl_or_b32 %0, %0, 1, %17\n\tv_add_u32 %1, %1, %18\n\tv_add_u32 %2, %2, %18\n\tv_lshl_or_b32 %3, %3, 1, %17\n\tv_add_u32 %4, %4, %18\n\tv_add_u32 %5, %5, %18\n\tv_lshl_or_b32 %6, %6, 1, %17\n\tv_add_u32 %7, %7, %18\n\tv_add_u32 %0, %0, %18\n\tv_lshl_or_b32 %1, %1, 1, %17\n\tv_add_u32 %2, %2, %18\n\tv_add_u32 %3, %3, %18\n\tv_lshl_or_b32 %4, %4, 1, %17\n\tv_add_u32 %5, %5, %18\n\tv_add_u32 %6, %6, %18\n\tv_lshl_or_b32 %7, %7, 1, %17\n\tv_add_u32 %0, %0, %18\n\tv_add_u32 %1, %1, %18\n\tv_lshl_or_b32 %2, %2, 1, %17\n\tv_add_u32 %3, %3, %18\n\tv_add_u32 %4, %4, %18\n\tv_lshl_or_b32 %5, %5, 1, %17\n\tv_add_u32 %6, %6, %18\n\tv_add_u32 %7, %7, %18\n\tv_lshl_or_b32 %0, %0, 1, %17\n\tv_add_u32 %1, %1, %18\n\tv_add_u32 %2, %2, %18\n\tv_lshl_or_b32 %3, %3, 1, %17\n\tv_add_u32 %4, %4, %18\n\tv_add_u32 %5, %5, %18\n\tv_lshl_or_b32 %6, %6, 1, %17\n\tv_add_u32 %7, %7, %18\n\tv_add_u32 %0, %0, %18\n\tv_lshl_or_b32 %1, %1, 1, %17\n\tv_add_u32 %2, %2, %18\n\tv_add_u32 %3, %3, %18\n\tv_lshl_or_b32 %4, %4, 1, %17\n\tv_add_u32 %5, %5, %18\n\tv_add_u32 %6, %6, %18\n\tv_lshl_or_b32 %7, %7, 1, %17\n\t");
        } else if (KIND == 91) {
            OPS("v_add_u32 %0, %0, %18\n\tv_add_f32 %1, %1, %17\n\tv_add_u32 %2, %2, %18\n\tv_add_f32 %3, %3, %17\n\tv_add_u32 %4, %4, %18\n\tv_add_f32 %5, %5, %17\n\tv_add_u32 %6, %6, %18\n\tv_add_f32 %7, %7, %17\n\tv_add_u32 %0, %0, %18\n\tv_add_f32 %1, %1, %17\n\tv_add_u32 %2, %2, %18\n\tv_add_f32 %3, %3, %17\n\tv_add_u32 %4, %4, %18\n\tv_add_f32 %5, %5, %17\n\tv_add_u32 %6, %6, %18\n\tv_add_f32 %7, %7, %17\n\tv_add_u32 %0, %0, %18\n\tv_add_f32 %1, %1, %17\n\tv_add_u32 %2, %2, %18\n\tv_add_f32 %3, %3, %17\n\tv_add_u32 %4, %4, %18\n\tv_add_f32 %5, %5, %17\n\tv_add_u32 %6, %6, %18\n\tv_add_f32 %7, %7, %17\n\tv_add_u32 %0, %0, %18\n\tv_add_f32 %1, %1, %17\n\tv_add_u32 %2, %2, %18\n\tv_add_f32 %3, %3, %17\n\tv_add_u32 %4, %4, %18\n\tv_add_f32 %5, %5, %17\n\tv_add_u32 %6, %6, %18\n\tv_add_f32 %7, %7, %17\n\tv_add_u32 %0, %0, %18\n\tv_add_f32 %1, %1, %17\n\tv_add_u32 %2, %2, %18\n\tv_add_f32 %3, %3, %17\n\tv_add_u32 %4, %4, %18\n\tv_add_f32 %5, %5, %17\n\tv_add_u32 %6, %6, %18\n\tv_add_f32 %7, %7, %17\n\tv_add_u32 %0, %0, %18\n\tv_add_f32 %1, %1, %17\n\tv_add_u32 %2, %2, %18\n\tv_add_f32 %3, %3, %17\n\tv_add_u32 %4, %4, %18\n\tv_add_f32 %5, %5, %17\n\tv_add_u32 %6, %6, %18\n\tv_add_f32 %7, %7, %17\n\tv_add_u32 %0, %0, %18\n\tv_add_f32 %1, %1, %17\n\tv_add_u32 %2, %2, %18\n\tv_add_f32 %3, %3, %17\n\tv_add_u32 %4, %4, %18\n\tv_add_f32 %5, %5, %17\n\tv_add_u32 %6, %6, %18\n\tv_add_f32 %7, %7, %17\n\tv_add_u32 %0, %0, %18\n\tv_add_f32 %1, %1, %17\n\tv_add_u32 %2, %2, %18\n\tv_add_f32 %3, %3, %17\n\tv_add_u32 %4, %4, %18\n\tv_add_f32 %5, %5, %17\n\tv_add_u32 %6, %6, %18\n\tv_add_f32 %7, %7, %17\n\t");
        } else if (KIND == 92) {
            OPS("v_add_f32 %0, %0, %17\n\tv_add_f32 %1, %1, %17\n\tv_add_f32 %2, %2, %17\n\tv_add_f32 %3, %3, %17\n\tv_add_f32 %4, %4, %17\n\tv_add_f32 %5, %5, %17\n\tv_add_f32 %6, %6, %17\n\tv_bfe_u32 %7, %7, %18, 7\n\tv_add_f32 %0, %0, %17\n\tv_add_f32 %1, %1, %17\n\tv_add_f32 %2, %2, %17\n\tv_add_f32 %3, %3, %17\n\tv_add_f32 %4, %4, %17\n\tv_add_f32 %5, %5, %17\n\tv_add_f32 %6, %6, %17\n\tv_bfe_u32 %7, %7, %18, 7\n\tv_add_f32 %0, %0, %17\n\tv_add_f32 %1, %1, %17\n\tv_add_f32 %2, %2, %17\n\tv_add_f32 %3, %3, %17\n\tv_add_f32 %4, %4, %17\n\tv_add_f32 %5, %5, %17\n\tv_add_f32 %6, %6, %17\n\tv_bfe_u32 %7, %7, %18, 7\n\tv_add_f32 %0, %0, %17\n\tv_add_f32 %1, %1, %17\n\tv_add_f32 %2, %2, %17\n\tv_add_f32 %3, %3, %17\n\tv_add_f32 %4, %4, %17\n\tv_add_f32 %5, %5, %17\n\tv_add_f32 %6, %6, %17\n\tv_bfe_u32 %7, %7, %18, 7\n\tv_add_f32 %0, %0, %17\n\tv_add_f32 %1, %1, %17\n\tv_add_f32 %2, %2, %17\n\tv_add_f32 %3, %3, %17\n\tv_add_f32 %4, %4, %17\n\tv_add_f32 %5, %5, %17\n\tv_add_f32 %6, %6, %17\n\tv_bfe_u32 %7, %7, %18, 7\n\tv_add_f32 %0, %0, %17\n\tv_add_f32 %1, %1, %17\n\tv_add_f32 %2, %2, %17\n\tv_add_f32 %3, %3, %17\n\tv_add_f32 %4, %4, %17\n\tv_add_f32 %5, %5, %17\n\tv_add_f32 %6, %6, %17\n\tv_bfe_u32 %7, %7, %18, 7\n\tv_add_f32 %0, %0, %17\n\tv_add_f32 %1, %1, %17\n\tv_add_f32 %2, %2, %17\n\tv_add_f32 %3, %3, %17\n\tv_add_f32 %4, %4, %17\n\tv_add_f32 %5, %5, %17\n\tv_add_f32 %6, %6, %17\n\tv_bfe_u32 %7, %7, %18, 7\n\tv_add_f32 %0, %0, %17\n\tv_add_f32 %1, %1, %17\n\tv_add_f32 %2, %2, %17\n\tv_add_f32 %3, %3, %17\n\tv_add_f32 %4, %4, %17\n\tv_add_f32 %5, %5, %17\n\tv_add_f32 %6, %6, %17\n\tv_bfe_u32 %7, %7, %18, 7\n\t");
        } else if (KIND == 93) {
            OPS("v_add_f32 %0, %0, %17\n\tv_add_f32 %1, %1, %17\n\tv_add_f32 %2, %2, %17\n\tv_add_f32 %3, %3, %17\n\tv_bfe_u32 %4, %4, %18, 7\n\tv_bfe_u32 %5, %5, %18, 7\n\tv_bfe_u32 %6, %6, %18, 7\n\tv_bfe_u32 %7, %7, %18, 7\n\tv_add_f32 %0, %0, %17\n\tv_add_f32 %1, %1, %17\n\tv_add_f32 %2, %2, %17\n\tv_add_f32 %3, %3, %17\n\tv_bfe_u32 %4, %4, %18, 7\n\tv_bfe_u32 %5, %5, %18, 7\n\tv_bfe_u32 %6, %6, %18, 7\n\tv_bfe_u32 %7, %7, %18, 7\n\tv_add_f32 %0, %0, %17\n\tv_add_f32 %1, %1, %17\n\tv_add_f32 %2, %2, %17\n\tv_add_f32 %3, %3, %17\n\tv_bfe_u32 %4, %4, %18, 7\n\tv_bfe_u32 %5, %5, %18, 7\n\tv_bfe_u32 %6, %6, %18, 7\n\tv_bfe_u32 %7, %7, %18, 7\n\tv_add_f32 %0, %0, %17\n\tv_add_f32 %1, %1, %17\n\tv_add_f32 %2, %2, %17\n\tv_add_f32 %3, %3, %17\n\tv_bfe_u32 %4, %4, %18, 7\n\tv_bfe_u32 %5, %5, %18, 7\n\tv_bfe_u32 %6, %6, %18, 7\n\tv_bfe_u32 %7, %7, %18, 7\n\tv_add_f32 %0, %0, %17\n\tv_add_f32 %1, %1, %17\n\tv_add_f32 %2, %2, %17\n\tv_add_f32 %3, %3, %17\n\tv_bfe_u32 %4, %4, %18, 7\n\tv_bfe_u32 %5, %5, %18, 7\n\tv_bfe_u32 %6, %6, %18, 7\n\tv_bfe_u32 %7, %7, %18, 7\n\tv_add_f32 %0, %0, %17\n\tv_add_f32 %1, %1, %17\n\tv_add_f32 %2, %2, %17\n\tv_add_f32 %3, %3, %17\n\tv_bfe_u32 %4, %4, %18, 7\n\tv_bfe_u32 %5, %5, %18, 7\n\tv_bfe_u32 %6, %6, %18, 7\n\tv_bfe_u32 %7, %7, %18, 7\n\tv_add_f32 %0, %0, %17\n\tv_add_f32 %1, %1, %17\n\tv_add_f32 %2, %2, %17\n\tv_add_f32 %3, %3, %17\n\tv_bfe_u32 %4, %4, %18, 7\n\tv_bfe_u32 %5, %5, %18, 7\n\tv_bfe_u32 %6, %6, %18, 7\n\tv_bfe_u32 %7, %7, %18, 7\n\tv_add_f32 %0, %0, %17\n\tv_add_f32 %1, %1, %17\n\tv_add_f32 %2, %2, %17\n\tv_add_f32 %3, %3, %17\n\tv_bfe_u32 %4, %4, %18, 7\n\tv_bfe_u32 %5, %5, %18, 7\n\tv_bfe_u32 %6, %6, %18, 7\n\tv_bfe_u32 %7, %7, %18, 7\n\t");
        } else if (KIND == 94) {
            OPS("v_add_f32 %0, %0, %17\n\tv_add_f32 %1, %1, %17\n\tv_add_f32 %2, %2, %17\n\tv_add_f32 %3, %3, %17\n\tv_add_f32 %4, %4, %17\n\tv_add_f32 %5, %5, %17\n\tv_add_f32 %6, %6, %17\n\tv_add_f32 %7, %7, %17\n\tv_add_f32 %0, %0, %17\n\tv_add_f32 %1, %1, %17\n\tv_add_f32 %2, %2, %17\n\tv_add_f32 %3, %3, %17\n\tv_add_f32 %4, %4, %17\n\tv_add_f32 %5, %5, %17\n\tv_add_f32 %6, %6, %17\n\tv_add_f32 %7, %7, %17\n\tv_bfe_u32 %0, %0, %18, 7\n\tv_bfe_u32 %1, %1, %18, 7\n\tv_bfe_u32 %2, %2, %18, 7\n\tv_bfe_u32 %3, %3, %18, 7\n\tv_bfe_u32 %4, %4, %18, 7\n\tv_bfe_u32 %5, %5, %18, 7\n\tv_bfe_u32 %6, %6, %18, 7\n\tv_bfe_u32 %7, %7, %18, 7\n\tv_bfe_u32 %0, %0, %18, 7\n\tv_bfe_u32 %1, %1, %18, 7\n\tv_bfe_u32 %2, %2, %18, 7\n\tv_bfe_u32 %3, %3, %18, 7\n\tv_bfe_u32 %4, %4, %18, 7\n\tv_bfe_u32 %5, %5, %18, 7\n\tv_bfe_u32 %6, %6, %18, 7\n\tv_bfe_u32 %7, %7, %18, 7\n\tv_add_f32 %0, %0, %17\n\tv_add_f32 %1, %1, %17\n\tv_add_f32 %2, %2, %17\n\tv_add_f32 %3, %3, %17\n\tv_add_f32 %4, %4, %17\n\tv_add_f32 %5, %5, %17\n\tv_add_f32 %6, %6, %17\n\tv_add_f32 %7, %7, %17\n\tv_add_f32 %0, %0, %17\n\tv_add_f32 %1, %1, %17\n\tv_add_f32 %2, %2, %17\n\tv_add_f32 %3, %3, %17\n\tv_add_f32 %4, %4, %17\n\tv_add_f32 %5, %5, %17\n\tv_add_f32 %6, %6, %17\n\tv_add_f32 %7, %7, %17\n\tv_bfe_u32 %0, %0, %18, 7\n\tv_bfe_u32 %1, %1, %18, 7\n\tv_bfe_u32 %2, %2, %18, 7\n\tv_bfe_u32 %3, %3, %18, 7\n\tv_bfe_u32 %4, %4, %18, 7\n\tv_bfe_u32 %5, %5, %18, 7\n\tv_bfe_u32 %6, %6, %18, 7\n\tv_bfe_u32 %7, %7, %18, 7\n\tv_bfe_u32 %0, %0, %18, 7\n\tv_bfe_u32 %1, %1, %18, 7\n\tv_bfe_u32 %2, %2, %18, 7\n\tv_bfe_u32 %3, %3, %18, 7\n\tv_bfe_u32 %4, %4, %18, 7\n\tv_bfe_u32 %5, %5, %18, 7\n\tv_bfe_u32 %6, %6, %18, 7\n\tv_bfe_u32 %7, %7, %18, 7\n\t");
        } else if (KIND == 95) {
            OPS("v_floor_f32 %0, %0\n\tv_floor_f32 %1, %1\n\tv_floor_f32 %2, %2\n\tv_floor_f32 %3, %3\n\tv_floor_f32 %4, %4\n\tv_floor_f32 %5, %5\n\tv_floor_f32 %6, %6\n\tv_floor_f32 %7, %7\n\tv_floor_f32 %0, %0\n\tv_floor_f32 %1, %1\n\tv_floor_f32 %2, %2\n\tv_floor_f32 %3, %3\n\tv_floor_f32 %4, %4\n\tv_floor_f32 %5, %5\n\tv_floor_f32 %6, %6\n\tv_floor_f32 %7, %7\n\tv_floor_f32 %0, %0\n\tv_floor_f32 %1, %1\n\tv_floor_f32 %2, %2\n\tv_floor_f32 %3, %3\n\tv_floor_f32 %4, %4\n\tv_floor_f32 %5, %5\n\tv_floor_f32 %6, %6\n\tv_floor_f32 %7, %7\n\tv_floor_f32 %0, %0\n\tv_floor_f32 %1, %1\n\tv_floor_f32 %2, %2\n\tv_floor_f32 %3, %3\n\tv_floor_f32 %4, %4\n\tv_floor_f32 %5, %5\n\tv_floor_f32 %6, %6\n\tv_floor_f32 %7, %7\n\tv_floor_f32 %0, %0\n\tv_floor_f32 %1, %1\n\tv_floor_f32 %2, %2\n\tv_floor_f32 %3, %3\n\tv_floor_f32 %4, %4\n\tv_floor_f32 %5, %5\n\tv_floor_f32 %6, %6\n\tv_floor_f32 %7, %7\n\tv_floor_f32 %0, %0\n\tv_floor_f32 %1, %1\n\tv_floor_f32 %2, %2\n\tv_floor_f32 %3, %3\n\tv_floor_f32 %4, %4\n\tv_floor_f32 %5, %5\n\tv_floor_f32 %6, %6\n\tv_floor_f32 %7, %7\n\tv_floor_f32 %0, %0\n\tv_floor_f32 %1, %1\n\tv_floor_f32 %2, %2\n\tv_floor_f32 %3, %3\n\tv_floor_f32 %4, %4\n\tv_floor_f32 %5, %5\n\tv_floor_f32 %6, %6\n\tv_floor_f32 %7, %7\n\tv_floor_f32 %0, %0\n\tv_floor_f32 %1, %1\n\tv_floor_f32 %2, %2\n\tv_floor_f32 %3, %3\n\tv_floor_f32 %4, %4\n\tv_floor_f32 %5, %5\n\tv_floor_f32 %6, %6\n\tv_floor_f32 %7, %7\n\t");
        } else if (KIND == 96) {
            OPS("v_ldexp_f32 %0, %0, %18\n\tv_ldexp_f32 %1, %1, %18\n\tv_ldexp_f32 %2, %2, %18\n\tv_ldexp_f32 %3, %3, %18\n\tv_ldexp_f32 %4, %4, %18\n\tv_ldexp_f32 %5, %5, %18\n\tv_ldexp_f32 %6, %6, %18\n\tv_ldexp_f32 %7, %7, %18\n\tv_ldexp_f32 %0, %0, %18\n\tv_ldexp_f32 %1, %1, %18\n\tv_ldexp_f32 %2, %2, %18\n\tv_ldexp_f32 %3, %3, %18\n\tv_ldexp_f32 %4, %4, %18\n\tv_ldexp_f32 %5, %5, %18\n\tv_ldexp_f32 %6, %6, %18\n\tv_ldexp_f32 %7, %7, %18\n\tv_ldexp_f32 %0, %0, %18\n\tv_ldexp_f32 %1, %1, %18\n\tv_ldexp_f32 %2, %2, %18\n\tv_ldexp_f32 %3, %3, %18\n\tv_ldexp_f32 %4, %4, %18\n\tv_ldexp_f32 %5, %5, %18\n\tv_ldexp_f32 %6, %6, %18\n\tv_ldexp_f32 %7, %7, %18\n\tv_ldexp_f32 %0, %0, %18\n\tv_ldexp_f32 %1, %1, %18\n\tv_ldexp_f32 %2, %2, %18\n\tv_ldexp_f32 %3, %3, %18\n\tv_ldexp_f32 %4, %4, %18\n\tv_ldexp_f32 %5, %5, %18\n\tv_ldexp_f32 %6, %6, %18\n\tv_ldexp_f32 %7, %7, %18\n\tv_ldexp_f32 %0, %0, %18\n\tv_ldexp_f32 %1, %1, %18\n\tv_ldexp_f32 %2, %2, %18\n\tv_ldexp_f32 %3, %3, %18\n\tv_ldexp_f32 %4, %4, %18\n\tv_ldexp_f32 %5, %5, %18\n\tv_ldexp_f32 %6, %6, %18\n\tv_ldexp_f32 %7, %7, %18\n\tv_ldexp_f32 %0, %0, %18\n\tv_ldexp_f32 %1, %1, %18\n\tv_ldexp_f32 %2, %2, %18\n\tv_ldexp_f32 %3, %3, %18\n\tv_ldexp_f32 %4, %4, %18\n\tv_ldexp_f32 %5, %5, %18\n\tv_ldexp_f32 %6, %6, %18\n\tv_ldexp_f32 %7, %7, %18\n\tv_ldexp_f32 %0, %0, %18\n\tv_ldexp_f32 %1, %1, %18\n\tv_ldexp_f32 %2, %2, %18\n\tv_ldexp_f32 %3, %3, %18\n\tv_ldexp_f32 %4, %4, %18\n\tv_ldexp_f32 %5, %5, %18\n\tv_ldexp_f32 %6, %6, %18\n\tv_ldexp_f32 %7, %7, %18\n\tv_ldexp_f32 %0, %0, %18\n\tv_ldexp_f32 %1, %1, %18\n\tv_ldexp_f32 %2, %2, %18\n\tv_ldexp_f32 %3, %3, %18\n\tv_ldexp_f32 %4, %4, %18\n\tv_ldexp_f32 %5, %5, %18\n\tv_ldexp_f32 %6, %6, %18\n\tv_ldexp_f32 %7, %7, %18\n\t");
        } else if (KIND == 97) {
            OPS("v_fract_f32 %0, %0\n\tv_fract_f32 %1, %1\n\tv_fract_f32 %2, %2\n\tv_fract_f32 %3, %3\n\tv_fract_f32 %4, %4\n\tv_fract_f32 %5, %5\n\tv_fract_f32 %6, %6\n\tv_fract_f32 %7, %7\n\tv_fract_f32 %0, %0\n\tv_fract_f32 %1, %1\n\tv_fract_f32 %2, %2\n\tv_fract_f32 %3, %3\n\tv_fract_f32 %4, %4\n\tv_fract_f32 %5, %5\n\tv_fract_f32 %6, %6\n\tv_fract_f32 %7, %7\n\tv_fract_f32 %0, %0\n\tv_fract_f32 %1, %1\n\tv_fract_f32 %2, %2\n\tv_fract_f32 %3, %3\n\tv_fract_f32 %4, %4\n\tv_fract_f32 %5, %5\n\tv_fract_f32 %6, %6\n\tv_fract_f32 %7, %7\n\tv_fract_f32 %0, %0\n\tv_fract_f32 %1, %1\n\tv_fract_f32 %2, %2\n\tv_fract_f32 %3, %3\n\tv_fract_f32 %4, %4\n\tv_fract_f32 %5, %5\n\tv_fract_f32 %6, %6\n\tv_fract_f32 %7, %7\n\tv_fract_f32 %0, %0\n\tv_fract_f32 %1, %1\n\tv_fract_f32 %2, %2\n\tv_fract_f32 %3, %3\n\tv_fract_f32 %4, %4\n\tv_fract_f32 %5, %5\n\tv_fract_f32 %6, %6\n\tv_fract_f32 %7, %7\n\tv_fract_f32 %0, %0\n\tv_fract_f32 %1, %1\n\tv_fract_f32 %2, %2\n\tv_fract_f32 %3, %3\n\tv_fract_f32 %4, %4\n\tv_fract_f32 %5, %5\n\tv_fract_f32 %6, %6\n\tv_fract_f32 %7, %7\n\tv_fract_f32 %0, %0\n\tv_fract_f32 %1, %1\n\tv_fract_f32 %2, %2\n\tv_fract_f32 %3, %3\n\tv_fract_f32 %4, %4\n\tv_fract_f32 %5, %5\n\tv_fract_f32 %6, %6\n\tv_fract_f32 %7, %7\n\tv_fract_f32 %0, %0\n\tv_fract_f32 %1, %1\n\tv_fract_f32 %2, %2\n\tv_fract_f32 %3, %3\n\tv_fract_f32 %4, %4\n\tv_fract_f32 %5, %5\n\tv_fract_f32 %6, %6\n\tv_fract_f32 %7, %7\n\t");
        } else if (KIND == 98) {
            OPS("v_trunc_f32 %0, %0\n\tv_trunc_f32 %1, %1\n\tv_trunc_f32 %2, %2\n\tv_trunc_f32 %3, %3\n\tv_trunc_f32 %4, %4\n\tv_trunc_f32 %5, %5\n\tv_trunc_f32 %6, %6\n\tv_trunc_f32 %7, %7\n\tv_trunc_f32 %0, %0\n\tv_trunc_f32 %1, %1\n\tv_trunc_f32 %2, %2\n\tv_trunc_f32 %3, %3\n\tv_trunc_f32 %4, %4\n\tv_trunc_f32 %5, %5\n\tv_trunc_f32 %6, %6\n\tv_trunc_f32 %7, %7\n\tv_trunc_f32 %0, %0\n\tv_trunc_f32 %1, %1\n\tv_trunc_f32 %2, %2\n\tv_trunc_f32 %3, %3\n\tv_trunc_f32 %4, %4\n\tv_trunc_f32 %5, %5\n\tv_trunc_f32 %6, %6\n\tv_trunc_f32 %7, %7\n\tv_trunc_f32 %0, %0\n\tv_trunc_f32 %1, %1\n\tv_trunc_f32 %2, %2\n\tv_trunc_f32 %3, %3\n\tv_trunc_f32 %4, %4\n\tv_trunc_f32 %5, %5\n\tv_trunc_f32 %6, %6\n\tv_trunc_f32 %7, %7\n\tv_trunc_f32 %0, %0\n\tv_trunc_f32 %1, %1\n\tv_trunc_f32 %2, %2\n\tv_trunc_f32 %3, %3\n\tv_trunc_f32 %4, %4\n\tv_trunc_f32 %5, %5\n\tv_trunc_f32 %6, %6\n\tv_trunc_f32 %7, %7\n\tv_trunc_f32 %0, %0\n\tv_trunc_f32 %1, %1\n\tv_trunc_f32 %2, %2\n\tv_trunc_f32 %3, %3\n\tv_trunc_f32 %4, %4\n\tv_trunc_f32 %5, %5\n\tv_trunc_f32 %6, %6\n\tv_trunc_f32 %7, %7\n\tv_trunc_f32 %0, %0\n\tv_trunc_f32 %1, %1\n\tv_trunc_f32 %2, %2\n\tv_trunc_f32 %3, %3\n\tv_trunc_f32 %4, %4\n\tv_trunc_f32 %5, %5\n\tv_trunc_f32 %6, %6\n\tv_trunc_f32 %7, %7\n\tv_trunc_f32 %0, %0\n\tv_trunc_f32 %1, %1\n\tv_trunc_f32 %2, %2\n\tv_trunc_f32 %3, %3\n\tv_trunc_f32 %4, %4\n\tv_trunc_f32 %5, %5\n\tv_trunc_f32 %6, %6\n\tv_trunc_f32 %7, %7\n\t");
        } else if (KIND == 99) {
            OPS("v_rndne_f32 %0, %0\n\tv_rndne_f32 %1, %1\n\tv_rndne_f32 %2, %2\n\tv_rndne_f32 %3, %3\n\tv_rndne_f32 %4, %4\n\tv_rndne_f32 %5, %5\n\tv_rndne_f32 %6, %6\n\tv_rndne_f32 %7, %7\n\tv_rndne_f32 %0, %0\n\tv_rndne_f32 %1, %1\n\tv_rndne_f32 %2, %2\n\tv_rndne_f32 %3, %3\n\tv_rndne_f32 %4, %4\n\tv_rndne_f32 %5, %5\n\tv_rndne_f32 %6, %6\n\tv_rndne_f32 %7, %7\n\tv_rndne_f32 %0, %0\n\tv_rndne_f32 %1, %1\n\tv_rndne_f32 %2, %2\n\tv_rndne_f32 %3, %3\n\tv_rndne_f32 %4, %4\n\tv_rndne_f32 %5, %5\n\tv_rndne_f32 %6, %6\n\tv_rndne_f32 %7, %7\n\tv_rndne_f32 %0, %0\n\tv_rndne_f32 %1, %1\n\tv_rndne_f32 %2, %2\n\tv_rndne_f32 %3, %3\n\tv_rndne_f32 %4, %4\n\tv_rndne_f32 %5, %5\n\tv_rndne_f32 %6, %6\n\tv_rndne_f32 %7, %7\n\tv_rndne_f32 %0, %0\n\tv_rndne_f32 %1, %1\n\tv_rndne_f32 %2, %2\n\tv_rndne_f32 %3, %3\n\tv_rndne_f32 %4, %4\n\tv_rndne_f32 %5, %5\n\tv_rndne_f32 %6, %6\n\tv_rndne_f32 %7, %7\n\tv_rndne_f32 %0, %0\n\tv_rndne_f32 %1, %1\n\tv_rndne_f32 %2, %2\n\tv_rndne_f32 %3, %3\n\tv_rndne_f32 %4, %4\n\tv_rndne_f32 %5, %5\n\tv_rndne_f32 %6, %6\n\tv_rndne_f32 %7, %7\n\tv_rndne_f32 %0, %0\n\tv_rndne_f32 %1, %1\n\tv_rndne_f32 %2, %2\n\tv_rndne_f32 %3, %3\n\tv_rndne_f32 %4, %4\n\tv_rndne_f32 %5, %5\n\tv_rndne_f32 %6, %6\n\tv_rndne_f32 %7, %7\n\tv_rndne_f32 %0, %0\n\tv_rndne_f32 %1, %1\n\tv_rndne_f32 %2, %2\n\tv_rndne_f32 %3, %3\n\tv_rndne_f32 %4, %4\n\tv_rndne_f32 %5, %5\n\tv_rndne_f32 %6, %6\n\tv_rndne_f32 %7, %7\n\t");
        } else if (KIND == 100) {
            OPS("v_mul_f32_e64 %0, %0, %17 clamp\n\tv_mul_f32_e64 %1, %1, %17 clamp\n\tv_mul_f32_e64 %2, %2, %17 clamp\n\tv_mul_f32_e64 %3, %3, %17 clamp\n\tv_mul_f32_e64 %4, %4, %17 clamp\n\tv_mul_f32_e64 %5, %5, %17 clamp\n\tv_mul_f32_e64 %6, %6, %17 clamp\n\tv_mul_f32_e64 %7, %7, %17 clamp\n\tv_mul_f32_e64 %0, %0, %17 clamp\n\tv_mul_f32_e64 %1, %1, %17 clamp\n\tv_mul_f32_e64 %2, %2, %17 clamp\n\tv_mul_f32_e64 %3, %3, %17 clamp\n\tv_mul_f32_e64 %4, %4, %17 clamp\n\tv_mul_f32_e64 %5, %5, %17 clamp\n\tv_mul_f32_e64 %6, %6, %17 clamp\n\tv_mul_f32_e64 %7, %7, %17 clamp\n\tv_mul_f32_e64 %0, %0, %17 clamp\n\tv_mul_f32_e64 %1, %1, %17 clamp\n\tv_mul_f32_e64 %2, %2, %17 clamp\n\tv_mul_f32_e64 %3, %3, %17 clamp\n\tv_mul_f32_e64 %4, %4, %17 clamp\n\tv_mul_f32_e64 %5, %5, %17 clamp\n\tv_mul_f32_e64 %6, %6, %17 clamp\n\tv_mul_f32_e64 %7, %7, %17 clamp\n\tv_mul_f32_e64 %0, %0, %17 clamp\n\tv_mul_f32_e64 %1, %1, %17 clamp\n\tv_mul_f32_e64 %2, %2, %17 clamp\n\tv_mul_f32_e64 %3, %3, %17 clamp\n\tv_mul_f32_e64 %4, %4, %17 clamp\n\tv_mul_f32_e64 %5, %5, %17 clamp\n\tv_mul_f32_e64 %6, %6, %17 clamp\n\tv_mul_f32_e64 %7, %7, %17 clamp\n\tv_mul_f32_e64 %0, %0, %17 clamp\n\tv_mul_f32_e64 %1, %1, %17 clamp\n\tv_mul_f32_e64 %2, %2, %17 clamp\n\tv_mul_f32_e64 %3, %3, %17 clamp\n\tv_mul_f32_e64 %4, %4, %17 clamp\n\tv_mul_f32_e64 %5, %5, %17 clamp\n\tv_mul_f32_e64 %6, %6, %17 clamp\n\tv_mul_f32_e64 %7, %7, %17 clamp\n\tv_mul_f32_e64 %0, %0, %17 clamp\n\tv_mul_f32_e64 %1, %1, %17 clamp\n\tv_mul_f32_e64 %2, %2, %17 clamp\n\tv_mul_f32_e64 %3, %3, %17 clamp\n\tv_mul_f32_e64 %4, %4, %17 clamp\n\tv_mul_f32_e64 %5, %5, %17 clamp\n\tv_mul_f32_e64 %6, %6, %17 clamp\n\tv_mul_f32_e64 %7, %7, %17 clamp\n\tv_mul_f32_e64 %0, %0, %17 clamp\n\tv_mul_f32_e64 %1, %1, %17 clamp\n\tv_mul_f32_e64 %2, %2, %17 clamp\n\tv_mul_f32_e64 %3, %3, %17 clamp\n\tv_mul_f32_e64 %4, %4, %17 clamp\n\tv_mul_f32_e64 %5, %5, %17 clamp\n\tv_mul_f32_e64 %6, %6, %17 clamp\n\tv_mul_f32_e64 %7, %7, %17 clamp\n\tv_mul_f32_e64 %0, %0, %17 clamp\n\tv_mul_f32_e64 %1, %1, %17 clamp\n\tv_mul_f32_e64 %2, %2, %17 clamp\n\tv_mul_f32_e64 %3, %3, %17 clamp\n\tv_mul_f32_e64 %4, %4, %17 clamp\n\tv_mul_f32_e64 %5, %5, %17 clamp\n\tv_mul_f32_e64 %6, %6, %17 clamp\n\tv_mul_f32_e64 %7, %7, %17 clamp\n\t");
        } else if (KIND == 101) {
            OPS("v_fma_f32 %0, %0, %17, %18 clamp\n\tv_fma_f32 %1, %1, %17, %18 clamp\n\tv_fma_f32 %2, %2, %17, %18 clamp\n\tv_fma_f32 %3, %3, %17, %18 clamp\n\tv_fma_f32 %4, %4, %17, %18 clamp\n\tv_fma_f32 %5, %5, %17, %18 clamp\n\tv_fma_f32 %6, %6, %17, %18 clamp\n\tv_fma_f32 %7, %7, %17, %18 clamp\n\tv_fma_f32 %0, %0, %17, %18 clamp\n\tv_fma_f32 %1, %1, %17, %18 clamp\n\tv_fma_f32 %2, %2, %17, %18 clamp\n\tv_fma_f32 %3, %3, %17, %18 clamp\n\tv_fma_f32 %4, %4, %17, %18 clamp\n\tv_fma_f32 %5, %5, %17, %18 clamp\n\tv_fma_f32 %6, %6, %17, %18 clamp\n\tv_fma_f32 %7, %7, %17, %18 clamp\n\tv_fma_f32 %0, %0, %17, %18 clamp\n\tv_fma_f32 %1, %1, %17, %18 clamp\n\tv_fma_f32 %2, %2, %17, %18 clamp\n\tv_fma_f32 %3, %3, %17, %18 clamp\n\tv_fma_f32 %4, %4, %17, %18 clamp\n\tv_fma_f32 %5, %5, %17, %18 clamp\n\tv_fma_f32 %6, %6, %17, %18 clamp\n\tv_fma_f32 %7, %7, %17, %18 clamp\n\tv_fma_f32 %0, %0, %17, %18 clamp\n\tv_fma_f32 %1, %1, %17, %18 clamp\n\tv_fma_f32 %2, %2, %17, %18 clamp\n\tv_fma_f32 %3, %3, %17, %18 clamp\n\tv_fma_f32 %4, %4, %17, %18 clamp\n\tv_fma_f32 %5, %5, %17, %18 clamp\n\tv_fma_f32 %6, %6, %17, %18 clamp\n\tv_fma_f32 %7, %7, %17, %18 clamp\n\tv_fma_f32 %0, %0, %17, %18 clamp\n\tv_fma_f32 %1, %1, %17, %18 clamp\n\tv_fma_f32 %2, %2, %17, %18 clamp\n\tv_fma_f32 %3, %3, %17, %18 clamp\n\tv_fma_f32 %4, %4, %17, %18 clamp\n\tv_fma_f32 %5, %5, %17, %18 clamp\n\tv_fma_f32 %6, %6, %17, %18 clamp\n\tv_fma_f32 %7, %7, %17, %18 clamp\n\tv_fma_f32 %0, %0, %17, %18 clamp\n\tv_fma_f32 %1, %1, %17, %18 clamp\n\tv_fma_f32 %2, %2, %17, %18 clamp\n\tv_fma_f32 %3, %3, %17, %18 clamp\n\tv_fma_f32 %4, %4, %17, %18 clamp\n\tv_fma_f32 %5, %5, %17, %18 clamp\n\tv_fma_f32 %6, %6, %17, %18 clamp\n\tv_fma_f32 %7, %7, %17, %18 clamp\n\tv_fma_f32 %0, %0, %17, %18 clamp\n\tv_fma_f32 %1, %1, %17, %18 clamp\n\tv_fma_f32 %2, %2, %17, %18 clamp\n\tv_fma_f32 %3, %3, %17, %18 clamp\n\tv_fma_f32 %4, %4, %17, %18 clamp\n\tv_fma_f32 %5, %5, %17, %18 clamp\n\tv_fma_f32 %6, %6, %17, %18 clamp\n\tv_fma_f32 %7, %7, %17, %18 clamp\n\tv_fma_f32 %0, %0, %17, %18 clamp\n\tv_fma_f32 %1, %1, %17, %18 clamp\n\tv_fma_f32 %2, %2, %17, %18 clamp\n\tv_fma_f32 %3, %3, %17, %18 clamp\n\tv_fma_f32 %4, %4, %17, %18 clamp\n\tv_fma_f32 %5, %5, %17, %18 clamp\n\tv_fma_f32 %6, %6, %17, %18 clamp\n\tv_fma_f32 %7, %7, %17, %18 clamp\n\t");
        } else if (KIND == 102) {
            OPS("v_frexp_exp_i32_f32 %0, %0\n\tv_frexp_exp_i32_f32 %1, %1\n\tv_frexp_exp_i32_f32 %2, %2\n\tv_frexp_exp_i32_f32 %3, %3\n\tv_frexp_exp_i32_f32 %4, %4\n\tv_frexp_exp_i32_f32 %5, %5\n\tv_frexp_exp_i32_f32 %6, %6\n\tv_frexp_exp_i32_f32 %7, %7\n\tv_frexp_exp_i32_f32 %0, %0\n\tv_frexp_exp_i32_f32 %1, %1\n\tv_frexp_exp_i32_f32 %2, %2\n\tv_frexp_exp_i32_f32 %3, %3\n\tv_frexp_exp_i32_f32 %4, %4\n\tv_frexp_exp_i32_f32 %5, %5\n\tv_frexp_exp_i32_f32 %6, %6\n\tv_frexp_exp_i32_f32 %7, %7\n\tv_frexp_exp_i32_f32 %0, %0\n\tv_frexp_exp_i32_f32 %1, %1\n\tv_frexp_exp_i32_f32 %2, %2\n\tv_frexp_exp_i32_f32 %3, %3\n\tv_frexp_exp_i32_f32 %4, %4\n\tv_frexp_exp_i32_f32 %5, %5\n\tv_frexp_exp_i32_f32 %6, %6\n\tv_frexp_exp_i32_f32 %7, %7\n\tv_frexp_exp_i32_f32 %0, %0\n\tv_frexp_exp_i32_f32 %1, %1\n\tv_frexp_exp_i32_f32 %2, %2\n\tv_frexp_exp_i32_f32 %3, %3\n\tv_frexp_exp_i32_f32 %4, %4\n\tv_frexp_exp_i32_f32 %5, %5\n\tv_frexp_exp_i32_f32 %6, %6\n\tv_frexp_exp_i32_f32 %7, %7\n\tv_frexp_exp_i32_f32 %0, %0\n\tv_frexp_exp_i32_f32 %1, %1\n\tv_frexp_exp_i32_f32 %2, %2\n\tv_frexp_exp_i32_f32 %3, %3\n\tv_frexp_exp_i32_f32 %4, %4\n\tv_frexp_exp_i32_f32 %5, %5\n\tv_frexp_exp_i32_f32 %6, %6\n\tv_frexp_exp_i32_f32 %7, %7\n\tv_frexp_exp_i32_f32 %0, %0\n\tv_frexp_exp_i32_f32 %1, %1\n\tv_frexp_exp_i32_f32 %2, %2\n\tv_frexp_exp_i32_f32 %3, %3\n\tv_frexp_exp_i32_f32 %4, %4\n\tv_frexp_exp_i32_f32 %5, %5\n\tv_frexp_exp_i32_f32 %6, %6\n\tv_frexp_exp_i32_f32 %7, %7\n\tv_frexp_exp_i32_f32 %0, %0\n\tv_frexp_exp_i32_f32 %1, %1\n\tv_frexp_exp_i32_f32 %2, %2\n\tv_frexp_exp_i32_f32 %3, %3\n\tv_frexp_exp_i32_f32 %4, %4\n\tv_frexp_exp_i32_f32 %5, %5\n\tv_frexp_exp_i32_f32 %6, %6\n\tv_frexp_exp_i32_f32 %7, %7\n\tv_frexp_exp_i32_f32 %0, %0\n\tv_frexp_exp_i32_f32 %1, %1\n\tv_frexp_exp_i32_f32 %2, %2\n\tv_frexp_exp_i32_f32 %3, %3\n\tv_frexp_exp_i32_f32 %4, %4\n\tv_frexp_exp_i32_f32 %5, %5\n\tv_frexp_exp_i32_f32 %6, %6\n\tv_frexp_exp_i32_f32 %7, %7\n\t");
        } else if (KIND == 103) {
            OPS("v_cvt_u32_f32 %0, %0\n\tv_cvt_u32_f32 %1, %1\n\tv_cvt_u32_f32 %2, %2\n\tv_cvt_u32_f32 %3, %3\n\tv_cvt_u32_f32 %4, %4\n\tv_cvt_u32_f32 %5, %5\n\tv_cvt_u32_f32 %6, %6\n\tv_cvt_u32_f32 %7, %7\n\tv_cvt_u32_f32 %0, %0\n\tv_cvt_u32_f32 %1, %1\n\tv_cvt_u32_f32 %2, %2\n\tv_cvt_u32_f32 %3, %3\n\tv_cvt_u32_f32 %4, %4\n\tv_cvt_u32_f32 %5, %5\n\tv_cvt_u32_f32 %6, %6\n\tv_cvt_u32_f32 %7, %7\n\tv_cvt_u32_f32 %0, %0\n\tv_cvt_u32_f32 %1, %1\n\tv_cvt_u32_f32 %2, %2\n\tv_cvt_u32_f32 %3, %3\n\tv_cvt_u32_f32 %4, %4\n\tv_cvt_u32_f32 %5, %5\n\tv_cvt_u32_f32 %6, %6\n\tv_cvt_u32_f32 %7, %7\n\tv_cvt_u32_f32 %0, %0\n\tv_cvt_u32_f32 %1, %1\n\tv_cvt_u32_f32 %2, %2\n\tv_cvt_u32_f32 %3, %3\n\tv_cvt_u32_f32 %4, %4\n\tv_cvt_u32_f32 %5, %5\n\tv_cvt_u32_f32 %6, %6\n\tv_cvt_u32_f32 %7, %7\n\tv_cvt_u32_f32 %0, %0\n\tv_cvt_u32_f32 %1, %1\n\tv_cvt_u32_f32 %2, %2\n\tv_cvt_u32_f32 %3, %3\n\tv_cvt_u32_f32 %4, %4\n\tv_cvt_u32_f32 %5, %5\n\tv_cvt_u32_f32 %6, %6\n\tv_cvt_u32_f32 %7, %7\n\tv_cvt_u32_f32 %0, %0\n\tv_cvt_u32_f32 %1, %1\n\tv_cvt_u32_f32 %2, %2\n\tv_cvt_u32_f32 %3, %3\n\tv_cvt_u32_f32 %4, %4\n\tv_cvt_u32_f32 %5, %5\n\tv_cvt_u32_f32 %6, %6\n\tv_cvt_u32_f32 %7, %7\n\tv_cvt_u32_f32 %0, %0\n\tv_cvt_u32_f32 %1, %1\n\tv_cvt_u32_f32 %2, %2\n\tv_cvt_u32_f32 %3, %3\n\tv_cvt_u32_f32 %4, %4\n\tv_cvt_u32_f32 %5, %5\n\tv_cvt_u32_f32 %6, %6\n\tv_cvt_u32_f32 %7, %7\n\tv_cvt_u32_f32 %0, %0\n\tv_cvt_u32_f32 %1, %1\n\tv_cvt_u32_f32 %2, %2\n\tv_cvt_u32_f32 %3, %3\n\tv_cvt_u32_f32 %4, %4\n\tv_cvt_u32_f32 %5, %5\n\tv_cvt_u32_f32 %6, %6\n\tv_cvt_u32_f32 %7, %7\n\t");
        } else if (KIND == 104) {
            OPS("v_mul_f32 %0, 0x7e800000, %0\n\tv_mul_f32 %1, 0x7e800000, %1\n\tv_mul_f32 %2, 0x7e800000, %2\n\tv_mul_f32 %3, 0x7e800000, %3\n\tv_mul_f32 %4, 0x7e800000, %4\n\tv_mul_f32 %5, 0x7e800000, %5\n\tv_mul_f32 %6, 0x7e800000, %6\n\tv_mul_f32 %7, 0x7e800000, %7\n\tv_mul_f32 %0, 0x7e800000, %0\n\tv_mul_f32 %1, 0x7e800000, %1\n\tv_mul_f32 %2, 0x7e800000, %2\n\tv_mul_f32 %3, 0x7e800000, %3\n\tv_mul_f32 %4, 0x7e800000, %4\n\tv_mul_f32 %5, 0x7e800000, %5\n\tv_mul_f32 %6, 0x7e800000, %6\n\tv_mul_f32 %7, 0x7e800000, %7\n\tv_mul_f32 %0, 0x7e800000, %0\n\tv_mul_f32 %1, 0x7e800000, %1\n\tv_mul_f32 %2, 0x7e800000, %2\n\tv_mul_f32 %3, 0x7e800000, %3\n\tv_mul_f32 %4, 0x7e800000, %4\n\tv_mul_f32 %5, 0x7e800000, %5\n\tv_mul_f32 %6, 0x7e800000, %6\n\tv_mul_f32 %7, 0x7e800000, %7\n\tv_mul_f32 %0, 0x7e800000, %0\n\tv_mul_f32 %1, 0x7e800000, %1\n\tv_mul_f32 %2, 0x7e800000, %2\n\tv_mul_f32 %3, 0x7e800000, %3\n\tv_mul_f32 %4, 0x7e800000, %4\n\tv_mul_f32 %5, 0x7e800000, %5\n\tv_mul_f32 %6, 0x7e800000, %6\n\tv_mul_f32 %7, 0x7e800000, %7\n\tv_mul_f32 %0, 0x7e800000, %0\n\tv_mul_f32 %1, 0x7e800000, %1\n\tv_mul_f32 %2, 0x7e800000, %2\n\tv_mul_f32 %3, 0x7e800000, %3\n\tv_mul_f32 %4, 0x7e800000, %4\n\tv_mul_f32 %5, 0x7e800000, %5\n\tv_mul_f32 %6, 0x7e800000, %6\n\tv_mul_f32 %7, 0x7e800000, %7\n\tv_mul_f32 %0, 0x7e800000, %0\n\tv_mul_f32 %1, 0x7e800000, %1\n\tv_mul_f32 %2, 0x7e800000, %2\n\tv_mul_f32 %3, 0x7e800000, %3\n\tv_mul_f32 %4, 0x7e800000, %4\n\tv_mul_f32 %5, 0x7e800000, %5\n\tv_mul_f32 %6, 0x7e800000, %6\n\tv_mul_f32 %7, 0x7e800000, %7\n\tv_mul_f32 %0, 0x7e800000, %0\n\tv_mul_f32 %1, 0x7e800000, %1\n\tv_mul_f32 %2, 0x7e800000, %2\n\tv_mul_f32 %3, 0x7e800000, %3\n\tv_mul_f32 %4, 0x7e800000, %4\n\tv_mul_f32 %5, 0x7e800000, %5\n\tv_mul_f32 %6, 0x7e800000, %6\n\tv_mul_f32 %7, 0x7e800000, %7\n\tv_mul_f32 %0, 0x7e800000, %0\n\tv_mul_f32 %1, 0x7e800000, %1\n\tv_mul_f32 %2, 0x7e800000, %2\n\tv_mul_f32 %3, 0x7e800000, %3\n\tv_mul_f32 %4, 0x7e800000, %4\n\tv_mul_f32 %5, 0x7e800000, %5\n\tv_mul_f32 %6, 0x7e800000, %6\n\tv_mul_f32 %7, 0x7e800000, %7\n\t");
        } else if (KIND == 105) {
            OPS("v_floor_f32 %0, %0\n\tv_add_f32 %1, %1, %17\n\tv_mul_f32 %2, %2, %17\n\tv_fma_f32 %3, %3, %17, %18\n\tv_floor_f32 %4, %4\n\tv_add_f32 %5, %5, %17\n\tv_mul_f32 %6, %6, %17\n\tv_fma_f32 %7, %7, %17, %18\n\tv_floor_f32 %0, %0\n\tv_add_f32 %1, %1, %17\n\tv_mul_f32 %2, %2, %17\n\tv_fma_f32 %3, %3, %17, %18\n\tv_floor_f32 %4, %4\n\tv_add_f32 %5, %5, %17\n\tv_mul_f32 %6, %6, %17\n\tv_fma_f32 %7, %7, %17, %18\n\tv_floor_f32 %0, %0\n\tv_add_f32 %1, %1, %17\n\tv_mul_f32 %2, %2, %17\n\tv_fma_f32 %3, %3, %17, %18\n\tv_floor_f32 %4, %4\n\tv_add_f32 %5, %5, %17\n\tv_mul_f32 %6, %6, %17\n\tv_fma_f32 %7, %7, %17, %18\n\tv_floor_f32 %0, %0\n\tv_add_f32 %1, %1, %17\n\tv_mul_f32 %2, %2, %17\n\tv_fma_f32 %3, %3, %17, %18\n\tv_floor_f32 %4, %4\n\tv_add_f32 %5, %5, %17\n\tv_mul_f32 %6, %6, %17\n\tv_fma_f32 %7, %7, %17, %18\n\tv_floor_f32 %0, %0\n\tv_add_f32 %1, %1, %17\n\tv_mul_f32 %2, %2, %17\n\tv_fma_f32 %3, %3, %17, %18\n\tv_floor_f32 %4, %4\n\tv_add_f32 %5, %5, %17\n\tv_mul_f32 %6, %6, %17\n\tv_fma_f32 %7, %7, %17, %18\n\tv_floor_f32 %0, %0\n\tv_add_f32 %1, %1, %17\n\tv_mul_f32 %2, %2, %17\n\tv_fma_f32 %3, %3, %17, %18\n\tv_floor_f32 %4, %4\n\tv_add_f32 %5, %5, %17\n\tv_mul_f32 %6, %6, %17\n\tv_fma_f32 %7, %7, %17, %18\n\tv_floor_f32 %0, %0\n\tv_add_f32 %1, %1, %17\n\tv_mul_f32 %2, %2, %17\n\tv_fma_f32 %3, %3, %17, %18\n\tv_floor_f32 %4, %4\n\tv_add_f32 %5, %5, %17\n\tv_mul_f32 %6, %6, %17\n\tv_fma_f32 %7, %7, %17, %18\n\tv_floor_f32 %0, %0\n\tv_add_f32 %1, %1, %17\n\tv_mul_f32 %2, %2, %17\n\tv_fma_f32 %3, %3, %17, %18\n\tv_floor_f32 %4, %4\n\tv_add_f32 %5, %5, %17\n\tv_mul_f32 %6, %6, %17\n\tv_fma_f32 %7, %7, %17, %18\n\t");
        } else if (KIND == 106) {
            OPS("v_bfe_u32 %0, %0, %18, 7\n\tv_add_f32 %1, %1, %17\n\tv_add_f32 %2, %2, %17\n\tv_bfe_u32 %3, %3, %18, 7\n\tv_add_f32 %4, %4, %17\n\tv_add_f32 %5, %5, %17\n\tv_bfe_u32 %6, %6, %18, 7\n\tv_add_f32 %7, %7, %17\n\tv_add_f32 %0, %0, %17\n\tv_bfe_u32 %1, %1, %18, 7\n\tv_add_f32 %2, %2, %17\n\tv_add_f32 %3, %3, %17\n\tv_bfe_u32 %4, %4, %18, 7\n\tv_add_f32 %5, %5, %17\n\tv_add_f32 %6, %6, %17\n\tv_bfe_u32 %7, %7, %18, 7\n\tv_add_f32 %0, %0, %17\n\tv_add_f32 %1, %1, %17\n\tv_bfe_u32 %2, %2, %18, 7\n\tv_add_f32 %3, %3, %17\n\tv_add_f32 %4, %4, %17\n\tv_bfe_u32 %5, %5, %18, 7\n\tv_add_f32 %6, %6, %17\n\tv_add_f32 %7, %7, %17\n\tv_bfe_u32 %0, %0, %18, 7\n\tv_add_f32 %1, %1, %17\n\tv_add_f32 %2, %2, %17\n\tv_bfe_u32 %3, %3, %18, 7\n\tv_add_f32 %4, %4, %17\n\tv_add_f32 %5, %5, %17\n\tv_bfe_u32 %6, %6, %18, 7\n\tv_add_f32 %7, %7, %17\n\tv_add_f32 %0, %0, %17\n\tv_bfe_u32 %1, %1, %18, 7\n\tv_add_f32 %2, %2, %17\n\tv_add_f32 %3, %3, %17\n\tv_bfe_u32 %4, %4, %18, 7\n\tv_add_f32 %5, %5, %17\n\tv_add_f32 %6, %6, %17\n\tv_bfe_u32 %7, %7, %18, 7\n\tv_add_f32 %0, %0, %17\n\tv_add_f32 %1, %1, %17\n\tv_bfe_u32 %2, %2, %18, 7\n\tv_add_f32 %3, %3, %17\n\tv_add_f32 %4, %4, %17\n\tv_bfe_u32 %5, %5, %18, 7\n\tv_add_f32 %6, %6, %17\n\tv_add_f32 %7, %7, %17\n\tv_bfe_u32 %0, %0, %18, 7\n\tv_add_f32 %1, %1, %17\n\tv_add_f32 %2, %2, %17\n\tv_bfe_u32 %3, %3, %18, 7\n\tv_add_f32 %4, %4, %17\n\tv_add_f32 %5, %5, %17\n\tv_bfe_u32 %6, %6, %18, 7\n\tv_add_f32 %7, %7, %17\n\tv_add_f32 %0, %0, %17\n\tv_bfe_u32 %1, %1, %18, 7\n\tv_add_f32 %2, %2, %17\n\tv_add_f32 %3, %3, %17\n\tv_bfe_u32 %4, %4, %18, 7\n\tv_add_f32 %5, %5, %17\n\tv_add_f32 %6, %6, %17\n\tv_bfe_u32 %7, %7, %18, 7\n\t");
        } else if (KIND == 107) {
            OPS("v_bfe_u32 %0, %0, %18, 7\n\tv_fma_f32 %1, %1, %17, %18 clamp\n\tv_bfe_u32 %2, %2, %18, 7\n\tv_fma_f32 %3, %3, %17, %18 clamp\n\tv_bfe_u32 %4, %4, %18, 7\n\tv_fma_f32 %5, %5, %17, %18 clamp\n\tv_bfe_u32 %6, %6, %18, 7\n\tv_fma_f32 %7, %7, %17, %18 clamp\n\tv_bfe_u32 %0, %0, %18, 7\n\tv_fma_f32 %1, %1, %17, %18 clamp\n\tv_bfe_u32 %2, %2, %18, 7\n\tv_fma_f32 %3, %3, %17, %18 clamp\n\tv_bfe_u32 %4, %4, %18, 7\n\tv_fma_f32 %5, %5, %17, %18 clamp\n\tv_bfe_u32 %6, %6, %18, 7\n\tv_fma_f32 %7, %7, %17, %18 clamp\n\tv_bfe_u32 %0, %0, %18, 7\n\tv_fma_f32 %1, %1, %17, %18 clamp\n\tv_bfe_u32 %2, %2, %18, 7\n\tv_fma_f32 %3, %3, %17, %18 clamp\n\tv_bfe_u32 %4, %4, %18, 7\n\tv_fma_f32 %5, %5, %17, %18 clamp\n\tv_bfe_u32 %6, %6, %18, 7\n\tv_fma_f32 %7, %7, %17, %18 clamp\n\tv_bfe_u32 %0, %0, %18, 7\n\tv_fma_f32 %1, %1, %17, %18 clamp\n\tv_bfe_u32 %2, %2, %18, 7\n\tv_fma_f32 %3, %3, %17, %18 clamp\n\tv_bfe_u32 %4, %4, %18, 7\n\tv_fma_f32 %5, %5, %17, %18 clamp\n\tv_bfe_u32 %6, %6, %18, 7\n\tv_fma_f32 %7, %7, %17, %18 clamp\n\tv_bfe_u32 %0, %0, %18, 7\n\tv_fma_f32 %1, %1, %17, %18 clamp\n\tv_bfe_u32 %2, %2, %18, 7\n\tv_fma_f32 %3, %3, %17, %18 clamp\n\tv_bfe_u32 %4, %4, %18, 7\n\tv_fma_f32 %5, %5, %17, %18 clamp\n\tv_bfe_u32 %6, %6, %18, 7\n\tv_fma_f32 %7, %7, %17, %18 clamp\n\tv_bfe_u32 %0, %0, %18, 7\n\tv_fma_f32 %1, %1, %17, %18 clamp\n\tv_bfe_u32 %2, %2, %18, 7\n\tv_fma_f32 %3, %3, %17, %18 clamp\n\tv_bfe_u32 %4, %4, %18, 7\n\tv_fma_f32 %5, %5, %17, %18 clamp\n\tv_bfe_u32 %6, %6, %18, 7\n\tv_fma_f32 %7, %7, %17, %18 clamp\n\tv_bfe_u32 %0, %0, %18, 7\n\tv_fma_f32 %1, %1, %17, %18 clamp\n\tv_bfe_u32 %2, %2, %18, 7\n\tv_fma_f32 %3, %3, %17, %18 clamp\n\tv_bfe_u32 %4, %4, %18, 7\n\tv_fma_f32 %5, %5, %17, %18 clamp\n\tv_bfe_u32 %6, %6, %18, 7\n\tv_fma_f32 %7, %7, %17, %18 clamp\n\tv_bfe_u32 %0, %0, %18, 7\n\tv_fma_f32 %1, %1, %17, %18 clamp\n\tv_bfe_u32 %2, %2, %18, 7\n\tv_fma_f32 %3, %3, %17, %18 clamp\n\tv_bfe_u32 %4, %4, %18, 7\n\tv_fma_f32 %5, %5, %17, %18 clamp\n\tv_bfe_u32 %6, %6, %18, 7\n\tv_fma_f32 %7, %7, %17, %18 clamp\n\t");
        } else if (KIND == 108) {
            OPS("v_bfe_u32 %0, %0, %18, 7\n\tv_bfe_u32 %1, %1, %18, 7\n\tv_add_f32 %2, %2, %17\n\tv_bfe_u32 %3, %3, %18, 7\n\tv_bfe_u32 %4, %4, %18, 7\n\tv_add_f32 %5, %5, %17\n\tv_bfe_u32 %6, %6, %18, 7\n\tv_bfe_u32 %7, %7, %18, 7\n\tv_add_f32 %0, %0, %17\n\tv_bfe_u32 %1, %1, %18, 7\n\tv_bfe_u32 %2, %2, %18, 7\n\tv_add_f32 %3, %3, %17\n\tv_bfe_u32 %4, %4, %18, 7\n\tv_bfe_u32 %5, %5, %18, 7\n\tv_add_f32 %6, %6, %17\n\tv_bfe_u32 %7, %7, %18, 7\n\tv_bfe_u32 %0, %0, %18, 7\n\tv_add_f32 %1, %1, %17\n\tv_bfe_u32 %2, %2, %18, 7\n\tv_bfe_u32 %3, %3, %18, 7\n\tv_add_f32 %4, %4, %17\n\tv_bfe_u32 %5, %5, %18, 7\n\tv_bfe_u32 %6, %6, %18, 7\n\tv_add_f32 %7, %7, %17\n\tv_bfe_u32 %0, %0, %18, 7\n\tv_bfe_u32 %1, %1, %18, 7\n\tv_add_f32 %2, %2, %17\n\tv_bfe_u32 %3, %3, %18, 7\n\tv_bfe_u32 %4, %4, %18, 7\n\tv_add_f32 %5, %5, %17\n\tv_bfe_u32 %6, %6, %18, 7\n\tv_bfe_u32 %7, %7, %18, 7\n\tv_add_f32 %0, %0, %17\n\tv_bfe_u32 %1, %1, %18, 7\n\tv_bfe_u32 %2, %2, %18, 7\n\tv_add_f32 %3, %3, %17\n\tv_bfe_u32 %4, %4, %18, 7\n\tv_bfe_u32 %5, %5, %18, 7\n\tv_add_f32 %6, %6, %17\n\tv_bfe_u32 %7, %7, %18, 7\n\tv_bfe_u32 %0, %0, %18, 7\n\tv_add_f32 %1, %1, %17\n\tv_bfe_u32 %2, %2, %18, 7\n\tv_bfe_u32 %3, %3, %18, 7\n\tv_add_f32 %4, %4, %17\n\tv_bfe_u32 %5, %5, %18, 7\n\tv_bfe_u32 %6, %6, %18, 7\n\tv_add_f32 %7, %7, %17\n\tv_bfe_u32 %0, %0, %18, 7\n\tv_bfe_u32 %1, %1, %18, 7\n\tv_add_f32 %2, %2, %17\n\tv_bfe_u32 %3, %3, %18, 7\n\tv_bfe_u32 %4, %4, %18, 7\n\tv_add_f32 %5, %5, %17\n\tv_bfe_u32 %6, %6, %18, 7\n\tv_bfe_u32 %7, %7, %18, 7\n\tv_add_f32 %0, %0, %17\n\tv_bfe_u32 %1, %1, %18, 7\n\tv_bfe_u32 %2, %2, %18, 7\n\tv_add_f32 %3, %3, %17\n\tv_bfe_u32 %4, %4, %18, 7\n\tv_bfe_u32 %5, %5, %18, 7\n\tv_add_f32 %6, %6, %17\n\tv_bfe_u32 %7, %7, %18, 7\n\t");
        } else if (KIND == 109) {
            OPS("v_bfe_u32 %0, %0, %18, 7\n\tv_add_u32 %1, %1, %18\n\tv_add_f32 %2, %2, %17\n\tv_bfe_u32 %3, %3, %18, 7\n\tv_add_u32 %4, %4, %18\n\tv_add_f32 %5, %5, %17\n\tv_bfe_u32 %6, %6, %18, 7\n\tv_add_u32 %7, %7, %18\n\tv_add_f32 %0, %0, %17\n\tv_bfe_u32 %1, %1, %18, 7\n\tv_add_u32 %2, %2, %18\n\tv_add_f32 %3, %3, %17\n\tv_bfe_u32 %4, %4, %18, 7\n\tv_add_u32 %5, %5, %18\n\tv_add_f32 %6, %6, %17\n\tv_bfe_u32 %7, %7, %18, 7\n\tv_add_u32 %0, %0, %18\n\tv_add_f32 %1, %1, %17\n\tv_bfe_u32 %2, %2, %18, 7\n\tv_add_u32 %3, %3, %18\n\tv_add_f32 %4, %4, %17\n\tv_bfe_u32 %5, %5, %18, 7\n\tv_add_u32 %6, %6, %18\n\tv_add_f32 %7, %7, %17\n\tv_bfe_u32 %0, %0, %18, 7\n\tv_add_u32 %1, %1, %18\n\tv_add_f32 %2, %2, %17\n\tv_bfe_u32 %3, %3, %18, 7\n\tv_add_u32 %4, %4, %18\n\tv_add_f32 %5, %5, %17\n\tv_bfe_u32 %6, %6, %18, 7\n\tv_add_u32 %7, %7, %18\n\tv_add_f32 %0, %0, %17\n\tv_bfe_u32 %1, %1, %18, 7\n\tv_add_u32 %2, %2, %18\n\tv_add_f32 %3, %3, %17\n\tv_bfe_u32 %4, %4, %18, 7\n\tv_add_u32 %5, %5, %18\n\tv_add_f32 %6, %6, %17\n\tv_bfe_u32 %7, %7, %18, 7\n\tv_add_u32 %0, %0, %18\n\tv_add_f32 %1, %1, %17\n\tv_bfe_u32 %2, %2, %18, 7\n\tv_add_u32 %3, %3, %18\n\tv_add_f32 %4, %4, %17\n\tv_bfe_u32 %5, %5, %18, 7\n\tv_add_u32 %6, %6, %18\n\tv_add_f32 %7, %7, %17\n\tv_bfe_u32 %0, %0, %18, 7\n\tv_add_u32 %1, %1, %18\n\tv_add_f32 %2, %2, %17\n\tv_bfe_u32 %3, %3, %18, 7\n\tv_add_u32 %4, %4, %18\n\tv_add_f32 %5, %5, %17\n\tv_bfe_u32 %6, %6, %18, 7\n\tv_add_u32 %7, %7, %18\n\tv_add_f32 %0, %0, %17\n\tv_bfe_u32 %1, %1, %18, 7\n\tv_add_u32 %2, %2, %18\n\tv_add_f32 %3, %3, %17\n\tv_bfe_u32 %4, %4, %18, 7\n\tv_add_u32 %5, %5, %18\n\tv_add_f32 %6, %6, %17\n\tv_bfe_u32 %7, %7, %18, 7\n\t");
        } else if (KIND == 110) {
            OPS("v_add_u32 %0, %0, %18\n\tv_add_f32 %1, %1, %17\n\tv_add_f32 %2, %2, %17\n\tv_add_u32 %3, %3, %18\n\tv_add_f32 %4, %4, %17\n\tv_add_f32 %5, %5, %17\n\tv_add_u32 %6, %6, %18\n\tv_add_f32 %7, %7, %17\n\tv_add_f32 %0, %0, %17\n\tv_add_u32 %1, %1, %18\n\tv_add_f32 %2, %2, %17\n\tv_add_f32 %3, %3, %17\n\tv_add_u32 %4, %4, %18\n\tv_add_f32 %5, %5, %17\n\tv_add_f32 %6, %6, %17\n\tv_add_u32 %7, %7, %18\n\tv_add_f32 %0, %0, %17\n\tv_add_f32 %1, %1, %17\n\tv_add_u32 %2, %2, %18\n\tv_add_f32 %3, %3, %17\n\tv_add_f32 %4, %4, %17\n\tv_add_u32 %5, %5, %18\n\tv_add_f32 %6, %6, %17\n\tv_add_f32 %7, %7, %17\n\tv_add_u32 %0, %0, %18\n\tv_add_f32 %1, %1, %17\n\tv_add_f32 %2, %2, %17\n\tv_add_u32 %3, %3, %18\n\tv_add_f32 %4, %4, %17\n\tv_add_f32 %5, %5, %17\n\tv_add_u32 %6, %6, %18\n\tv_add_f32 %7, %7, %17\n\tv_add_f32 %0, %0, %17\n\tv_add_u32 %1, %1, %18\n\tv_add_f32 %2, %2, %17\n\tv_add_f32 %3, %3, %17\n\tv_add_u32 %4, %4, %18\n\tv_add_f32 %5, %5, %17\n\tv_add_f32 %6, %6, %17\n\tv_add_u32 %7, %7, %18\n\tv_add_f32 %0, %0, %17\n\tv_add_f32 %1, %1, %17\n\tv_add_u32 %2, %2, %18\n\tv_add_f32 %3, %3, %17\n\tv_add_f32 %4, %4, %17\n\tv_add_u32 %5, %5, %18\n\tv_add_f32 %6, %6, %17\n\tv_add_f32 %7, %7, %17\n\tv_add_u32 %0, %0, %18\n\tv_add_f32 %1, %1, %17\n\tv_add_f32 %2, %2, %17\n\tv_add_u32 %3, %3, %18\n\tv_add_f32 %4, %4, %17\n\tv_add_f32 %5, %5, %17\n\tv_add_u32 %6, %6, %18\n\tv_add_f32 %7, %7, %17\n\tv_add_f32 %0, %0, %17\n\tv_add_u32 %1, %1, %18\n\tv_add_f32 %2, %2, %17\n\tv_add_f32 %3, %3, %17\n\tv_add_u32 %4, %4, %18\n\tv_add_f32 %5, %5, %17\n\tv_add_f32 %6, %6, %17\n\tv_add_u32 %7, %7, %18\n\t");
        } else if (KIND == 111) {
            OPS("v_xor_b32 %0, %0, %18\n\tv_bfe_u32 %1, %1, %18, 7\n\tv_add_f32 %2, %2, %17\n\tv_add_f32 %3, %3, %17\n\tv_xor_b32 %4, %4, %18\n\tv_bfe_u32 %5, %5, %18, 7\n\tv_add_f32 %6, %6, %17\n\tv_add_f32 %7, %7, %17\n\tv_xor_b32 %0, %0, %18\n\tv_bfe_u32 %1, %1, %18, 7\n\tv_add_f32 %2, %2, %17\n\tv_add_f32 %3, %3, %17\n\tv_xor_b32 %4, %4, %18\n\tv_bfe_u32 %5, %5, %18, 7\n\tv_add_f32 %6, %6, %17\n\tv_add_f32 %7, %7, %17\n\tv_xor_b32 %0, %0, %18\n\tv_bfe_u32 %1, %1, %18, 7\n\tv_add_f32 %2, %2, %17\n\tv_add_f32 %3, %3, %17\n\tv_xor_b32 %4, %4, %18\n\tv_bfe_u32 %5, %5, %18, 7\n\tv_add_f32 %6, %6, %17\n\tv_add_f32 %7, %7, %17\n\tv_xor_b32 %0, %0, %18\n\tv_bfe_u32 %1, %1, %18, 7\n\tv_add_f32 %2, %2, %17\n\tv_add_f32 %3, %3, %17\n\tv_xor_b32 %4, %4, %18\n\tv_bfe_u32 %5, %5, %18, 7\n\tv_add_f32 %6, %6, %17\n\tv_add_f32 %7, %7, %17\n\tv_xor_b32 %0, %0, %18\n\tv_bfe_u32 %1, %1, %18, 7\n\tv_add_f32 %2, %2, %17\n\tv_add_f32 %3, %3, %17\n\tv_xor_b32 %4, %4, %18\n\tv_bfe_u32 %5, %5, %18, 7\n\tv_add_f32 %6, %6, %17\n\tv_add_f32 %7, %7, %17\n\tv_xor_b32 %0, %0, %18\n\tv_bfe_u32 %1, %1, %18, 7\n\tv_add_f32 %2, %2, %17\n\tv_add_f32 %3, %3, %17\n\tv_xor_b32 %4, %4, %18\n\tv_bfe_u32 %5, %5, %18, 7\n\tv_add_f32 %6, %6, %17\n\tv_add_f32 %7, %7, %17\n\tv_xor_b32 %0, %0, %18\n\tv_bfe_u32 %1, %1, %18, 7\n\tv_add_f32 %2, %2, %17\n\tv_add_f32 %3, %3, %17\n\tv_xor_b32 %4, %4, %18\n\tv_bfe_u32 %5, %5, %18, 7\n\tv_add_f32 %6, %6, %17\n\tv_add_f32 %7, %7, %17\n\tv_xor_b32 %0, %0, %18\n\tv_bfe_u32 %1, %1, %18, 7\n\tv_add_f32 %2, %2, %17\n\tv_add_f32 %3, %3, %17\n\tv_xor_b32 %4, %4, %18\n\tv_bfe_u32 %5, %5, %18, 7\n\tv_add_f32 %6, %6, %17\n\tv_add_f32 %7, %7, %17\n\t");
        } else if (KIND == 112) {
            OPS("v_mov_b32 %0, %17\n\tv_bfe_u32 %1, %1, %18, 7\n\tv_mov_b32 %2, %17\n\tv_bfe_u32 %3, %3, %18, 7\n\tv_mov_b32 %4, %17\n\tv_bfe_u32 %5, %5, %18, 7\n\tv_mov_b32 %6, %17\n\tv_bfe_u32 %7, %7, %18, 7\n\tv_mov_b32 %0, %17\n\tv_bfe_u32 %1, %1, %18, 7\n\tv_mov_b32 %2, %17\n\tv_bfe_u32 %3, %3, %18, 7\n\tv_mov_b32 %4, %17\n\tv_bfe_u32 %5, %5, %18, 7\n\tv_mov_b32 %6, %17\n\tv_bfe_u32 %7, %7, %18, 7\n\tv_mov_b32 %0, %17\n\tv_bfe_u32 %1, %1, %18, 7\n\tv_mov_b32 %2, %17\n\tv_bfe_u32 %3, %3, %18, 7\n\tv_mov_b32 %4, %17\n\tv_bfe_u32 %5, %5, %18, 7\n\tv_mov_b32 %6, %17\n\tv_bfe_u32 %7, %7, %18, 7\n\tv_mov_b32 %0, %17\n\tv_bfe_u32 %1, %1, %18, 7\n\tv_mov_b32 %2, %17\n\tv_bfe_u32 %3, %3, %18, 7\n\tv_mov_b32 %4, %17\n\tv_bfe_u32 %5, %5, %18, 7\n\tv_mov_b32 %6, %17\n\tv_bfe_u32 %7, %7, %18, 7\n\tv_mov_b32 %0, %17\n\tv_bfe_u32 %1, %1, %18, 7\n\tv_mov_b32 %2, %17\n\tv_bfe_u32 %3, %3, %18, 7\n\tv_mov_b32 %4, %17\n\tv_bfe_u32 %5, %5, %18, 7\n\tv_mov_b32 %6, %17\n\tv_bfe_u32 %7, %7, %18, 7\n\tv_mov_b32 %0, %17\n\tv_bfe_u32 %1, %1, %18, 7\n\tv_mov_b32 %2, %17\n\tv_bfe_u32 %3, %3, %18, 7\n\tv_mov_b32 %4, %17\n\tv_bfe_u32 %5, %5, %18, 7\n\tv_mov_b32 %6, %17\n\tv_bfe_u32 %7, %7, %18, 7\n\tv_mov_b32 %0, %17\n\tv_bfe_u32 %1, %1, %18, 7\n\tv_mov_b32 %2, %17\n\tv_bfe_u32 %3, %3, %18, 7\n\tv_mov_b32 %4, %17\n\tv_bfe_u32 %5, %5, %18, 7\n\tv_mov_b32 %6, %17\n\tv_bfe_u32 %7, %7, %18, 7\n\tv_mov_b32 %0, %17\n\tv_bfe_u32 %1, %1, %18, 7\n\tv_mov_b32 %2, %17\n\tv_bfe_u32 %3, %3, %18, 7\n\tv_mov_b32 %4, %17\n\tv_bfe_u32 %5, %5, %18, 7\n\tv_mov_b32 %6, %17\n\tv_bfe_u32 %7, %7, %18, 7\n\t");
        } else if (KIND == 113) {
            OPS("v_lshrrev_b32 %0, 3, %0\n\tv_bfe_u32 %1, %1, %18, 7\n\tv_lshrrev_b32 %2, 3, %2\n\tv_bfe_u32 %3, %3, %18, 7\n\tv_lshrrev_b32 %4, 3, %4\n\tv_bfe_u32 %5, %5, %18, 7\n\tv_lshrrev_b32 %6, 3, %6\n\tv_bfe_u32 %7, %7, %18, 7\n\tv_lshrrev_b32 %0, 3, %0\n\tv_bfe_u32 %1, %1, %18, 7\n\tv_lshrrev_b32 %2, 3, %2\n\tv_bfe_u32 %3, %3, %18, 7\n\tv_lshrrev_b32 %4, 3, %4\n\tv_bfe_u32 %5, %5, %18, 7\n\tv_lshrrev_b32 %6, 3, %6\n\tv_bfe_u32 %7, %7, %18, 7\n\tv_lshrrev_b32 %0, 3, %0\n\tv_bfe_u32 %1, %1, %18, 7\n\tv_lshrrev_b32 %2, 3, %2\n\tv_bfe_u32 %3, %3, %18, 7\n\tv_lshrrev_b32 %4, 3, %4\n\tv_bfe_u32 %5, %5, %18, 7\n\tv_lshrrev_b32 %6, 3, %6\n\tv_bfe_u32 %7, %7, %18, 7\n\tv_lshrrev_b32 %0, 3, %0\n\tv_bfe_u32 %1, %1, %18, 7\n\tv_lshrrev_b32 %2, 3, %2\n\tv_bfe_u32 %3, %3, %18, 7\n\tv_lshrrev_b32 %4, 3, %4\n\tv_bfe_u32 %5, %5, %18, 7\n\tv_lshrrev_b32 %6, 3, %6\n\tv_bfe_u32 %7, %7, %18, 7\n\tv_lshrrev_b32 %0, 3, %0\n\tv_bfe_u32 %1, %1, %18, 7\n\tv_lshrrev_b32 %2, 3, %2\n\tv_bfe_u32 %3, %3, %18, 7\n\tv_lshrrev_b32 %4, 3, %4\n\tv_bfe_u32 %5, %5, %18, 7\n\tv_lshrrev_b32 %6, 3, %6\n\tv_bfe_u32 %7, %7, %18, 7\n\tv_lshrrev_b32 %0, 3, %0\n\tv_bfe_u32 %1, %1, %18, 7\n\tv_lshrrev_b32 %2, 3, %2\n\tv_bfe_u32 %3, %3, %18, 7\n\tv_lshrrev_b32 %4, 3, %4\n\tv_bfe_u32 %5, %5, %18, 7\n\tv_lshrrev_b32 %6, 3, %6\n\tv_bfe_u32 %7, %7, %18, 7\n\tv_lshrrev_b32 %0, 3, %0\n\tv_bfe_u32 %1, %1, %18, 7\n\tv_lshrrev_b32 %2, 3, %2\n\tv_bfe_u32 %3, %3, %18, 7\n\tv_lshrrev_b32 %4, 3, %4\n\tv_bfe_u32 %5, %5, %18, 7\n\tv_lshrrev_b32 %6, 3, %6\n\tv_bfe_u32 %7, %7, %18, 7\n\tv_lshrrev_b32 %0, 3, %0\n\tv_bfe_u32 %1, %1, %18, 7\n\tv_lshrrev_b32 %2, 3, %2\n\tv_bfe_u32 %3, %3, %18, 7\n\tv_lshrrev_b32 %4, 3, %4\n\tv_bfe_u32 %5, %5, %18, 7\n\tv_lshrrev_b32 %6, 3, %6\n\tv_bfe_u32 %7, %7, %18, 7\n\t");
        } else if (KIND == 114) {
            OPS("v_and_b32 %0, %0, %17\n\tv_bfe_u32 %1, %1, %18, 7\n\tv_and_b32 %2, %2, %17\n\tv_bfe_u32 %3, %3, %18, 7\n\tv_and_b32 %4, %4, %17\n\tv_bfe_u32 %5, %5, %18, 7\n\tv_and_b32 %6, %6, %17\n\tv_bfe_u32 %7, %7, %18, 7\n\tv_and_b32 %0, %0, %17\n\tv_bfe_u32 %1, %1, %18, 7\n\tv_and_b32 %2, %2, %17\n\tv_bfe_u32 %3, %3, %18, 7\n\tv_and_b32 %4, %4, %17\n\tv_bfe_u32 %5, %5, %18, 7\n\tv_and_b32 %6, %6, %17\n\tv_bfe_u32 %7, %7, %18, 7\n\tv_and_b32 %0, %0, %17\n\tv_bfe_u32 %1, %1, %18, 7\n\tv_and_b32 %2, %2, %17\n\tv_bfe_u32 %3, %3, %18, 7\n\tv_and_b32 %4, %4, %17\n\tv_bfe_u32 %5, %5, %18, 7\n\tv_and_b32 %6, %6, %17\n\tv_bfe_u32 %7, %7, %18, 7\n\tv_and_b32 %0, %0, %17\n\tv_bfe_u32 %1, %1, %18, 7\n\tv_and_b32 %2, %2, %17\n\tv_bfe_u32 %3, %3, %18, 7\n\tv_and_b32 %4, %4, %17\n\tv_bfe_u32 %5, %5, %18, 7\n\tv_and_b32 %6, %6, %17\n\tv_bfe_u32 %7, %7, %18, 7\n\tv_and_b32 %0, %0, %17\n\tv_bfe_u32 %1, %1, %18, 7\n\tv_and_b32 %2, %2, %17\n\tv_bfe_u32 %3, %3, %18, 7\n\tv_and_b32 %4, %4, %17\n\tv_bfe_u32 %5, %5, %18, 7\n\tv_and_b32 %6, %6, %17\n\tv_bfe_u32 %7, %7, %18, 7\n\tv_and_b32 %0, %0, %17\n\tv_bfe_u32 %1, %1, %18, 7\n\tv_and_b32 %2, %2, %17\n\tv_bfe_u32 %3, %3, %18, 7\n\tv_and_b32 %4, %4, %17\n\tv_bfe_u32 %5, %5, %18, 7\n\tv_and_b32 %6, %6, %17\n\tv_bfe_u32 %7, %7, %18, 7\n\tv_and_b32 %0, %0, %17\n\tv_bfe_u32 %1, %1, %18, 7\n\tv_and_b32 %2, %2, %17\n\tv_bfe_u32 %3, %3, %18, 7\n\tv_and_b32 %4, %4, %17\n\tv_bfe_u32 %5, %5, %18, 7\n\tv_and_b32 %6, %6, %17\n\tv_bfe_u32 %7, %7, %18, 7\n\tv_and_b32 %0, %0, %17\n\tv_bfe_u32 %1, %1, %18, 7\n\tv_and_b32 %2, %2, %17\n\tv_bfe_u32 %3, %3, %18, 7\n\tv_and_b32 %4, %4, %17\n\tv_bfe_u32 %5, %5, %18, 7\n\tv_and_b32 %6, %6, %17\n\tv_bfe_u32 %7, %7, %18, 7\n\t");
        } else if (KIND == 115) {
            unsigned long long sv;
            asm volatile("s_mov_b64 %0, exec" : "=s"(sv));
#define X(n) "v_cmpx_ge_u32 vcc, " R(n) ", %17\n\t"
            OPS(BLOCK8x8(X));
#undef X
            asm volatile("s_mov_b64 exec, %0" : : "s"(sv));
        } else if (KIND == 116) {
            OPS("v_bfe_u32 %0, %0, %18, 7\n\tv_bfe_u32 %1, %1, %18, 7\n\tv_bfe_u32 %2, %2, %18, 7\n\tv_bfe_u32 %3, %3, %18, 7\n\tv_add_u32 %4, %4, %18\n\tv_add_u32 %5, %5, %18\n\tv_add_u32 %6, %6, %18\n\tv_add_u32 %7, %7, %18\n\tv_bfe_u32 %0, %0, %18, 7\n\tv_bfe_u32 %1, %1, %18, 7\n\tv_bfe_u32 %2, %2, %18, 7\n\tv_bfe_u32 %3, %3, %18, 7\n\tv_add_u32 %4, %4, %18\n\tv_add_u32 %5, %5, %18\n\tv_add_u32 %6, %6, %18\n\tv_add_u32 %7, %7, %18\n\tv_bfe_u32 %0, %0, %18, 7\n\tv_bfe_u32 %1, %1, %18, 7\n\tv_bfe_u32 %2, %2, %18, 7\n\tv_bfe_u32 %3, %3, %18, 7\n\tv_add_u32 %4, %4, %18\n\tv_add_u32 %5, %5, %18\n\tv_add_u32 %6, %6, %18\n\tv_add_u32 %7, %7, %18\n\tv_bfe_u32 %0, %0, %18, 7\n\tv_bfe_u32 %1, %1, %18, 7\n\tv_bfe_u32 %2, %2, %18, 7\n\tv_bfe_u32 %3, %3, %18, 7\n\tv_add_u32 %4, %4, %18\n\tv_add_u32 %5, %5, %18\n\tv_add_u32 %6, %6, %18\n\tv_add_u32 %7, %7, %18\n\tv_bfe_u32 %0, %0, %18, 7\n\tv_bfe_u32 %1, %1, %18, 7\n\tv_bfe_u32 %2, %2, %18, 7\n\tv_bfe_u32 %3, %3, %18, 7\n\tv_add_u32 %4, %4, %18\n\tv_add_u32 %5, %5, %18\n\tv_add_u32 %6, %6, %18\n\tv_add_u32 %7, %7, %18\n\tv_bfe_u32 %0, %0, %18, 7\n\tv_bfe_u32 %1, %1, %18, 7\n\tv_bfe_u32 %2, %2, %18, 7\n\tv_bfe_u32 %3, %3, %18, 7\n\tv_add_u32 %4, %4, %18\n\tv_add_u32 %5, %5, %18\n\tv_add_u32 %6, %6, %18\n\tv_add_u32 %7, %7, %18\n\tv_bfe_u32 %0, %0, %18, 7\n\tv_bfe_u32 %1, %1, %18, 7\n\tv_bfe_u32 %2, %2, %18, 7\n\tv_bfe_u32 %3, %3, %18, 7\n\tv_add_u32 %4, %4, %18\n\tv_add_u32 %5, %5, %18\n\tv_add_u32 %6, %6, %18\n\tv_add_u32 %7, %7, %18\n\tv_bfe_u32 %0, %0, %18, 7\n\tv_bfe_u32 %1, %1, %18, 7\n\tv_bfe_u32 %2, %2, %18, 7\n\tv_bfe_u32 %3, %3, %18, 7\n\tv_add_u32 %4, %4, %18\n\tv_add_u32 %5, %5, %18\n\tv_add_u32 %6, %6, %18\n\tv_add_u32 %7, %7, %18\n\t");
        } else if (KIND == 117) {
            OPS("v_bfe_u32 %0, %0, %18, 7\n\tv_bfe_u32 %1, %1, %18, 7\n\tv_add_u32 %2, %2, %18\n\tv_add_u32 %3, %3, %18\n\tv_bfe_u32 %4, %4, %18, 7\n\tv_bfe_u32 %5, %5, %18, 7\n\tv_add_u32 %6, %6, %18\n\tv_add_u32 %7, %7, %18\n\tv_bfe_u32 %0, %0, %18, 7\n\tv_bfe_u32 %1, %1, %18, 7\n\tv_add_u32 %2, %2, %18\n\tv_add_u32 %3, %3, %18\n\tv_bfe_u32 %4, %4, %18, 7\n\tv_bfe_u32 %5, %5, %18, 7\n\tv_add_u32 %6, %6, %18\n\tv_add_u32 %7, %7, %18\n\tv_bfe_u32 %0, %0, %18, 7\n\tv_bfe_u32 %1, %1, %18, 7\n\tv_add_u32 %2, %2, %18\n\tv_add_u32 %3, %3, %18\n\tv_bfe_u32 %4, %4, %18, 7\n\tv_bfe_u32 %5, %5, %18, 7\n\tv_add_u32 %6, %6, %18\n\tv_add_u32 %7, %7, %18\n\tv_bfe_u32 %0, %0, %18, 7\n\tv_bfe_u32 %1, %1, %18, 7\n\tv_add_u32 %2, %2, %18\n\tv_add_u32 %3, %3, %18\n\tv_bfe_u32 %4, %4, %18, 7\n\tv_bfe_u32 %5, %5, %18, 7\n\tv_add_u32 %6, %6, %18\n\tv_add_u32 %7, %7, %18\n\tv_bfe_u32 %0, %0, %18, 7\n\tv_bfe_u32 %1, %1, %18, 7\n\tv_add_u32 %2, %2, %18\n\tv_add_u32 %3, %3, %18\n\tv_bfe_u32 %4, %4, %18, 7\n\tv_bfe_u32 %5, %5, %18, 7\n\tv_add_u32 %6, %6, %18\n\tv_add_u32 %7, %7, %18\n\tv_bfe_u32 %0, %0, %18, 7\n\tv_bfe_u32 %1, %1, %18, 7\n\tv_add_u32 %2, %2, %18\n\tv_add_u32 %3, %3, %18\n\tv_bfe_u32 %4, %4, %18, 7\n\tv_bfe_u32 %5, %5, %18, 7\n\tv_add_u32 %6, %6, %18\n\tv_add_u32 %7, %7, %18\n\tv_bfe_u32 %0, %0, %18, 7\n\tv_bfe_u32 %1, %1, %18, 7\n\tv_add_u32 %2, %2, %18\n\tv_add_u32 %3, %3, %18\n\tv_bfe_u32 %4, %4, %18, 7\n\tv_bfe_u32 %5, %5, %18, 7\n\tv_add_u32 %6, %6, %18\n\tv_add_u32 %7, %7, %18\n\tv_bfe_u32 %0, %0, %18, 7\n\tv_bfe_u32 %1, %1, %18, 7\n\tv_add_u32 %2, %2, %18\n\tv_add_u32 %3, %3, %18\n\tv_bfe_u32 %4, %4, %18, 7\n\tv_bfe_u32 %5, %5, %18, 7\n\tv_add_u32 %6, %6, %18\n\tv_add_u32 %7, %7, %18\n\t");
        } else if (KIND == 118) {
            OPS("v_bfe_u32 %0, %0, %18, 7\n\tv_add_f32 %1, %1, %17\n\tv_add_u32 %2, %2, %18\n\tv_add_f32 %3, %3, %17\n\tv_bfe_u32 %4, %4, %18, 7\n\tv_add_f32 %5, %5, %17\n\tv_add_u32 %6, %6, %18\n\tv_add_f32 %7, %7, %17\n\tv_bfe_u32 %0, %0, %18, 7\n\tv_add_f32 %1, %1, %17\n\tv_add_u32 %2, %2, %18\n\tv_add_f32 %3, %3, %17\n\tv_bfe_u32 %4, %4, %18, 7\n\tv_add_f32 %5, %5, %17\n\tv_add_u32 %6, %6, %18\n\tv_add_f32 %7, %7, %17\n\tv_bfe_u32 %0, %0, %18, 7\n\tv_add_f32 %1, %1, %17\n\tv_add_u32 %2, %2, %18\n\tv_add_f32 %3, %3, %17\n\tv_bfe_u32 %4, %4, %18, 7\n\tv_add_f32 %5, %5, %17\n\tv_add_u32 %6, %6, %18\n\tv_add_f32 %7, %7, %17\n\tv_bfe_u32 %0, %0, %18, 7\n\tv_add_f32 %1, %1, %17\n\tv_add_u32 %2, %2, %18\n\tv_add_f32 %3, %3, %17\n\tv_bfe_u32 %4, %4, %18, 7\n\tv_add_f32 %5, %5, %17\n\tv_add_u32 %6, %6, %18\n\tv_add_f32 %7, %7, %17\n\tv_bfe_u32 %0, %0, %18, 7\n\tv_add_f32 %1, %1, %17\n\tv_add_u32 %2, %2, %18\n\tv_add_f32 %3, %3, %17\n\tv_bfe_u32 %4, %4, %18, 7\n\tv_add_f32 %5, %5, %17\n\tv_add_u32 %6, %6, %18\n\tv_add_f32 %7, %7, %17\n\tv_bfe_u32 %0, %0, %18, 7\n\tv_add_f32 %1, %1, %17\n\tv_add_u32 %2, %2, %18\n\tv_add_f32 %3, %3, %17\n\tv_bfe_u32 %4, %4, %18, 7\n\tv_add_f32 %5, %5, %17\n\tv_add_u32 %6, %6, %18\n\tv_add_f32 %7, %7, %17\n\tv_bfe_u32 %0, %0, %18, 7\n\tv_add_f32 %1, %1, %17\n\tv_add_u32 %2, %2, %18\n\tv_add_f32 %3, %3, %17\n\tv_bfe_u32 %4, %4, %18, 7\n\tv_add_f32 %5, %5, %17\n\tv_add_u32 %6, %6, %18\n\tv_add_f32 %7, %7, %17\n\tv_bfe_u32 %0, %0, %18, 7\n\tv_add_f32 %1, %1, %17\n\tv_add_u32 %2, %2, %18\n\tv_add_f32 %3, %3, %17\n\tv_bfe_u32 %4, %4, %18, 7\n\tv_add_f32 %5, %5, %17\n\tv_add_u32 %6, %6, %18\n\tv_add_f32 %7, %7, %17\n\t");
        } else if (KIND == 119) {
            OPS("v_bfe_u32 %0, %0, %18, 7\n\tv_add_u32 %1, %1, %18\n\tv_add_f32 %2, %2, %17\n\tv_add_f32 %3, %3, %17\n\tv_bfe_u32 %4, %4, %18, 7\n\tv_add_u32 %5, %5, %18\n\tv_add_f32 %6, %6, %17\n\tv_add_f32 %7, %7, %17\n\tv_bfe_u32 %0, %0, %18, 7\n\tv_add_u32 %1, %1, %18\n\tv_add_f32 %2, %2, %17\n\tv_add_f32 %3, %3, %17\n\tv_bfe_u32 %4, %4, %18, 7\n\tv_add_u32 %5, %5, %18\n\tv_add_f32 %6, %6, %17\n\tv_add_f32 %7, %7, %17\n\tv_bfe_u32 %0, %0, %18, 7\n\tv_add_u32 %1, %1, %18\n\tv_add_f32 %2, %2, %17\n\tv_add_f32 %3, %3, %17\n\tv_bfe_u32 %4, %4, %18, 7\n\tv_add_u32 %5, %5, %18\n\tv_add_f32 %6, %6, %17\n\tv_add_f32 %7, %7, %17\n\tv_bfe_u32 %0, %0, %18, 7\n\tv_add_u32 %1, %1, %18\n\tv_add_f32 %2, %2, %17\n\tv_add_f32 %3, %3, %17\n\tv_bfe_u32 %4, %4, %18, 7\n\tv_add_u32 %5, %5, %18\n\tv_add_f32 %6, %6, %17\n\tv_add_f32 %7, %7, %17\n\tv_bfe_u32 %0, %0, %18, 7\n\tv_add_u32 %1, %1, %18\n\tv_add_f32 %2, %2, %17\n\tv_add_f32 %3, %3, %17\n\tv_bfe_u32 %4, %4, %18, 7\n\tv_add_u32 %5, %5, %18\n\tv_add_f32 %6, %6, %17\n\tv_add_f32 %7, %7, %17\n\tv_bfe_u32 %0, %0, %18, 7\n\tv_add_u32 %1, %1, %18\n\tv_add_f32 %2, %2, %17\n\tv_add_f32 %3, %3, %17\n\tv_bfe_u32 %4, %4, %18, 7\n\tv_add_u32 %5, %5, %18\n\tv_add_f32 %6, %6, %17\n\tv_add_f32 %7, %7, %17\n\tv_bfe_u32 %0, %0, %18, 7\n\tv_add_u32 %1, %1, %18\n\tv_add_f32 %2, %2, %17\n\tv_add_f32 %3, %3, %17\n\tv_bfe_u32 %4, %4, %18, 7\n\tv_add_u32 %5, %5, %18\n\tv_add_f32 %6, %6, %17\n\tv_add_f32 %7, %7, %17\n\tv_bfe_u32 %0, %0, %18, 7\n\tv_add_u32 %1, %1, %18\n\tv_add_f32 %2, %2, %17\n\tv_add_f32 %3, %3, %17\n\tv_bfe_u32 %4, %4, %18, 7\n\tv_add_u32 %5, %5, %18\n\tv_add_f32 %6, %6, %17\n\tv_add_f32 %7, %7, %17\n\t");
        } else if (KIND == 120) {
            OPS("v_bfe_u32 %0, %0, %18, 7\n\tv_add_f32 %1, %1, %17\n\tv_bfe_u32 %2, %2, %18, 7\n\tv_add_f32 %3, %3, %17\n\tv_bfe_u32 %4, %4, %18, 7\n\tv_add_f32 %5, %5, %17\n\tv_bfe_u32 %6, %6, %18, 7\n\tv_add_f32 %7, %7, %17\n\tv_add_u32 %0, %0, %18\n\tv_add_u32 %1, %1, %18\n\tv_add_u32 %2, %2, %18\n\tv_add_u32 %3, %3, %18\n\tv_add_f32 %4, %4, %17\n\tv_add_f32 %5, %5, %17\n\tv_add_f32 %6, %6, %17\n\tv_add_f32 %7, %7, %17\n\tv_bfe_u32 %0, %0, %18, 7\n\tv_add_f32 %1, %1, %17\n\tv_bfe_u32 %2, %2, %18, 7\n\tv_add_f32 %3, %3, %17\n\tv_bfe_u32 %4, %4, %18, 7\n\tv_add_f32 %5, %5, %17\n\tv_bfe_u32 %6, %6, %18, 7\n\tv_add_f32 %7, %7, %17\n\tv_add_u32 %0, %0, %18\n\tv_add_u32 %1, %1, %18\n\tv_add_u32 %2, %2, %18\n\tv_add_u32 %3, %3, %18\n\tv_add_f32 %4, %4, %17\n\tv_add_f32 %5, %5, %17\n\tv_add_f32 %6, %6, %17\n\tv_add_f32 %7, %7, %17\n\tv_bfe_u32 %0, %0, %18, 7\n\tv_add_f32 %1, %1, %17\n\tv_bfe_u32 %2, %2, %18, 7\n\tv_add_f32 %3, %3, %17\n\tv_bfe_u32 %4, %4, %18, 7\n\tv_add_f32 %5, %5, %17\n\tv_bfe_u32 %6, %6, %18, 7\n\tv_add_f32 %7, %7, %17\n\tv_add_u32 %0, %0, %18\n\tv_add_u32 %1, %1, %18\n\tv_add_u32 %2, %2, %18\n\tv_add_u32 %3, %3, %18\n\tv_add_f32 %4, %4, %17\n\tv_add_f32 %5, %5, %17\n\tv_add_f32 %6, %6, %17\n\tv_add_f32 %7, %7, %17\n\tv_bfe_u32 %0, %0, %18, 7\n\tv_add_f32 %1, %1, %17\n\tv_bfe_u32 %2, %2, %18, 7\n\tv_add_f32 %3, %3, %17\n\tv_bfe_u32 %4, %4, %18, 7\n\tv_add_f32 %5, %5, %17\n\tv_bfe_u32 %6, %6, %18, 7\n\tv_add_f32 %7, %7, %17\n\tv_add_u32 %0, %0, %18\n\tv_add_u32 %1, %1, %18\n\tv_add_u32 %2, %2, %18\n\tv_add_u32 %3, %3, %18\n\tv_add_f32 %4, %4, %17\n\tv_add_f32 %5, %5, %17\n\tv_add_f32 %6, %6, %17\n\tv_add_f32 %7, %7, %17\n\t");
        } else if (KIND == 121) {
            OPS("v_bfe_u32 %0, %0, %18, 7\n\tv_add_u32 %1, %1, %18\n\tv_add_u32 %2, %2, %18\n\tv_add_u32 %3, %3, %18\n\tv_bfe_u32 %4, %4, %18, 7\n\tv_add_u32 %5, %5, %18\n\tv_add_u32 %6, %6, %18\n\tv_add_u32 %7, %7, %18\n\tv_bfe_u32 %0, %0, %18, 7\n\tv_add_u32 %1, %1, %18\n\tv_add_u32 %2, %2, %18\n\tv_add_u32 %3, %3, %18\n\tv_bfe_u32 %4, %4, %18, 7\n\tv_add_u32 %5, %5, %18\n\tv_add_u32 %6, %6, %18\n\tv_add_u32 %7, %7, %18\n\tv_bfe_u32 %0, %0, %18, 7\n\tv_add_u32 %1, %1, %18\n\tv_add_u32 %2, %2, %18\n\tv_add_u32 %3, %3, %18\n\tv_bfe_u32 %4, %4, %18, 7\n\tv_add_u32 %5, %5, %18\n\tv_add_u32 %6, %6, %18\n\tv_add_u32 %7, %7, %18\n\tv_bfe_u32 %0, %0, %18, 7\n\tv_add_u32 %1, %1, %18\n\tv_add_u32 %2, %2, %18\n\tv_add_u32 %3, %3, %18\n\tv_bfe_u32 %4, %4, %18, 7\n\tv_add_u32 %5, %5, %18\n\tv_add_u32 %6, %6, %18\n\tv_add_u32 %7, %7, %18\n\tv_bfe_u32 %0, %0, %18, 7\n\tv_add_u32 %1, %1, %18\n\tv_add_u32 %2, %2, %18\n\tv_add_u32 %3, %3, %18\n\tv_bfe_u32 %4, %4, %18, 7\n\tv_add_u32 %5, %5, %18\n\tv_add_u32 %6, %6, %18\n\tv_add_u32 %7, %7, %18\n\tv_bfe_u32 %0, %0, %18, 7\n\tv_add_u32 %1, %1, %18\n\tv_add_u32 %2, %2, %18\n\tv_add_u32 %3, %3, %18\n\tv_bfe_u32 %4, %4, %18, 7\n\tv_add_u32 %5, %5, %18\n\tv_add_u32 %6, %6, %18\n\tv_add_u32 %7, %7, %18\n\tv_bfe_u32 %0, %0, %18, 7\n\tv_add_u32 %1, %1, %18\n\tv_add_u32 %2, %2, %18\n\tv_add_u32 %3, %3, %18\n\tv_bfe_u32 %4, %4, %18, 7\n\tv_add_u32 %5, %5, %18\n\tv_add_u32 %6, %6, %18\n\tv_add_u32 %7, %7, %18\n\tv_bfe_u32 %0, %0, %18, 7\n\tv_add_u32 %1, %1, %18\n\tv_add_u32 %2, %2, %18\n\tv_add_u32 %3, %3, %18\n\tv_bfe_u32 %4, %4, %18, 7\n\tv_add_u32 %5, %5, %18\n\tv_add_u32 %6, %6, %18\n\tv_add_u32 %7, %7, %18\n\t");
        } else if (KIND == 122) {
            OPS("v_bfe_u32 %0, %0, %18, 7\n\tv_add_u32 %1, %1, %18\n\tv_add_u32 %2, %2, %18\n\tv_add_u32 %3, %3, %18\n\tv_add_f32 %4, %4, %17\n\tv_add_f32 %5, %5, %17\n\tv_add_f32 %6, %6, %17\n\tv_add_f32 %7, %7, %17\n\tv_bfe_u32 %0, %0, %18, 7\n\tv_add_u32 %1, %1, %18\n\tv_add_u32 %2, %2, %18\n\tv_add_u32 %3, %3, %18\n\tv_add_f32 %4, %4, %17\n\tv_add_f32 %5, %5, %17\n\tv_add_f32 %6, %6, %17\n\tv_add_f32 %7, %7, %17\n\tv_bfe_u32 %0, %0, %18, 7\n\tv_add_u32 %1, %1, %18\n\tv_add_u32 %2, %2, %18\n\tv_add_u32 %3, %3, %18\n\tv_add_f32 %4, %4, %17\n\tv_add_f32 %5, %5, %17\n\tv_add_f32 %6, %6, %17\n\tv_add_f32 %7, %7, %17\n\tv_bfe_u32 %0, %0, %18, 7\n\tv_add_u32 %1, %1, %18\n\tv_add_u32 %2, %2, %18\n\tv_add_u32 %3, %3, %18\n\tv_add_f32 %4, %4, %17\n\tv_add_f32 %5, %5, %17\n\tv_add_f32 %6, %6, %17\n\tv_add_f32 %7, %7, %17\n\tv_bfe_u32 %0, %0, %18, 7\n\tv_add_u32 %1, %1, %18\n\tv_add_u32 %2, %2, %18\n\tv_add_u32 %3, %3, %18\n\tv_add_f32 %4, %4, %17\n\tv_add_f32 %5, %5, %17\n\tv_add_f32 %6, %6, %17\n\tv_add_f32 %7, %7, %17\n\tv_bfe_u32 %0, %0, %18, 7\n\tv_add_u32 %1, %1, %18\n\tv_add_u32 %2, %2, %18\n\tv_add_u32 %3, %3, %18\n\tv_add_f32 %4, %4, %17\n\tv_add_f32 %5, %5, %17\n\tv_add_f32 %6, %6, %17\n\tv_add_f32 %7, %7, %17\n\tv_bfe_u32 %0, %0, %18, 7\n\tv_add_u32 %1, %1, %18\n\tv_add_u32 %2, %2, %18\n\tv_add_u32 %3, %3, %18\n\tv_add_f32 %4, %4, %17\n\tv_add_f32 %5, %5, %17\n\tv_add_f32 %6, %6, %17\n\tv_add_f32 %7, %7, %17\n\tv_bfe_u32 %0, %0, %18, 7\n\tv_add_u32 %1, %1, %18\n\tv_add_u32 %2, %2, %18\n\tv_add_u32 %3, %3, %18\n\tv_add_f32 %4, %4, %17\n\tv_add_f32 %5, %5, %17\n\tv_add_f32 %6, %6, %17\n\tv_add_f32 %7, %7, %17\n\t");
        } else if (KIND == 123) {
            OPS("v_bfe_u32 %0, %0, %18, 7\n\tv_bfe_u32 %1, %1, %18, 7\n\tv_add_u32 %2, %2, %18\n\tv_add_u32 %3, %3, %18\n\tv_add_u32 %4, %4, %18\n\tv_add_u32 %5, %5, %18\n\tv_bfe_u32 %6, %6, %18, 7\n\tv_bfe_u32 %7, %7, %18, 7\n\tv_bfe_u32 %0, %0, %18, 7\n\tv_bfe_u32 %1, %1, %18, 7\n\tv_bfe_u32 %2, %2, %18, 7\n\tv_add_u32 %3, %3, %18\n\tv_add_u32 %4, %4, %18\n\tv_bfe_u32 %5, %5, %18, 7\n\tv_bfe_u32 %6, %6, %18, 7\n\tv_add_u32 %7, %7, %18\n\tv_bfe_u32 %0, %0, %18, 7\n\tv_bfe_u32 %1, %1, %18, 7\n\tv_add_u32 %2, %2, %18\n\tv_add_u32 %3, %3, %18\n\tv_add_u32 %4, %4, %18\n\tv_add_u32 %5, %5, %18\n\tv_bfe_u32 %6, %6, %18, 7\n\tv_bfe_u32 %7, %7, %18, 7\n\tv_bfe_u32 %0, %0, %18, 7\n\tv_bfe_u32 %1, %1, %18, 7\n\tv_bfe_u32 %2, %2, %18, 7\n\tv_add_u32 %3, %3, %18\n\tv_add_u32 %4, %4, %18\n\tv_bfe_u32 %5, %5, %18, 7\n\tv_bfe_u32 %6, %6, %18, 7\n\tv_add_u32 %7, %7, %18\n\tv_bfe_u32 %0, %0, %18, 7\n\tv_bfe_u32 %1, %1, %18, 7\n\tv_add_u32 %2, %2, %18\n\tv_add_u32 %3, %3, %18\n\tv_add_u32 %4, %4, %18\n\tv_add_u32 %5, %5, %18\n\tv_bfe_u32 %6, %6, %18, 7\n\tv_bfe_u32 %7, %7, %18, 7\n\tv_bfe_u32 %0, %0, %18, 7\n\tv_bfe_u32 %1, %1, %18, 7\n\tv_bfe_u32 %2, %2, %18, 7\n\tv_add_u32 %3, %3, %18\n\tv_add_u32 %4, %4, %18\n\tv_bfe_u32 %5, %5, %18, 7\n\tv_bfe_u32 %6, %6, %18, 7\n\tv_add_u32 %7, %7, %18\n\tv_bfe_u32 %0, %0, %18, 7\n\tv_bfe_u32 %1, %1, %18, 7\n\tv_add_u32 %2, %2, %18\n\tv_add_u32 %3, %3, %18\n\tv_add_u32 %4, %4, %18\n\tv_add_u32 %5, %5, %18\n\tv_bfe_u32 %6, %6, %18, 7\n\tv_bfe_u32 %7, %7, %18, 7\n\tv_bfe_u32 %0, %0, %18, 7\n\tv_bfe_u32 %1, %1, %18, 7\n\tv_bfe_u32 %2, %2, %18, 7\n\tv_add_u32 %3, %3, %18\n\tv_add_u32 %4, %4, %18\n\tv_bfe_u32 %5, %5, %18, 7\n\tv_bfe_u32 %6, %6, %18, 7\n\tv_add_u32 %7, %7, %18\n\t");
        } else if (KIND == 124) {
            OPS("v_bfe_u32 %0, %0, %18, 7\n\tv_bfe_u32 %1, %1, %18, 7\n\tv_bfe_u32 %2, %2, %18, 7\n\tv_bfe_u32 %3, %3, %18, 7\n\tv_bfe_u32 %4, %4, %18, 7\n\tv_bfe_u32 %5, %5, %18, 7\n\tv_bfe_u32 %6, %6, %18, 7\n\tv_bfe_u32 %7, %7, %18, 7\n\tv_bfe_u32 %0, %0, %18, 7\n\tv_add_u32 %1, %1, %18\n\tv_add_u32 %2, %2, %18\n\tv_add_u32 %3, %3, %18\n\tv_add_u32 %4, %4, %18\n\tv_add_u32 %5, %5, %18\n\tv_add_u32 %6, %6, %18\n\tv_add_u32 %7, %7, %18\n\tv_bfe_u32 %0, %0, %18, 7\n\tv_bfe_u32 %1, %1, %18, 7\n\tv_bfe_u32 %2, %2, %18, 7\n\tv_bfe_u32 %3, %3, %18, 7\n\tv_bfe_u32 %4, %4, %18, 7\n\tv_bfe_u32 %5, %5, %18, 7\n\tv_bfe_u32 %6, %6, %18, 7\n\tv_bfe_u32 %7, %7, %18, 7\n\tv_bfe_u32 %0, %0, %18, 7\n\tv_add_u32 %1, %1, %18\n\tv_add_u32 %2, %2, %18\n\tv_add_u32 %3, %3, %18\n\tv_add_u32 %4, %4, %18\n\tv_add_u32 %5, %5, %18\n\tv_add_u32 %6, %6, %18\n\tv_add_u32 %7, %7, %18\n\tv_bfe_u32 %0, %0, %18, 7\n\tv_bfe_u32 %1, %1, %18, 7\n\tv_bfe_u32 %2, %2, %18, 7\n\tv_bfe_u32 %3, %3, %18, 7\n\tv_bfe_u32 %4, %4, %18, 7\n\tv_bfe_u32 %5, %5, %18, 7\n\tv_bfe_u32 %6, %6, %18, 7\n\tv_bfe_u32 %7, %7, %18, 7\n\tv_bfe_u32 %0, %0, %18, 7\n\tv_add_u32 %1, %1, %18\n\tv_add_u32 %2, %2, %18\n\tv_add_u32 %3, %3, %18\n\tv_add_u32 %4, %4, %18\n\tv_add_u32 %5, %5, %18\n\tv_add_u32 %6, %6, %18\n\tv_add_u32 %7, %7, %18\n\tv_bfe_u32 %0, %0, %18, 7\n\tv_bfe_u32 %1, %1, %18, 7\n\tv_bfe_u32 %2, %2, %18, 7\n\tv_bfe_u32 %3, %3, %18, 7\n\tv_bfe_u32 %4, %4, %18, 7\n\tv_bfe_u32 %5, %5, %18, 7\n\tv_bfe_u32 %6, %6, %18, 7\n\tv_bfe_u32 %7, %7, %18, 7\n\tv_bfe_u32 %0, %0, %18, 7\n\tv_add_u32 %1, %1, %18\n\tv_add_u32 %2, %2, %18\n\tv_add_u32 %3, %3, %18\n\tv_add_u32 %4, %4, %18\n\tv_add_u32 %5, %5, %18\n\tv_add_u32 %6, %6, %18\n\tv_add_u32 %7, %7, %18\n\t");
        } else if (KIND == 125) {
            OPS("v_bfe_u32 %0, %0, %18, 7\n\tv_add_f32 %1, %1, %17\n\tv_bfe_u32 %2, %2, %18, 7\n\tv_add_f32 %3, %3, %17\n\tv_add_u32 %4, %4, %18\n\tv_add_u32 %5, %5, %18\n\tv_add_u32 %6, %6, %18\n\tv_add_u32 %7, %7, %18\n\tv_bfe_u32 %0, %0, %18, 7\n\tv_add_f32 %1, %1, %17\n\tv_bfe_u32 %2, %2, %18, 7\n\tv_add_f32 %3, %3, %17\n\tv_bfe_u32 %4, %4, %18, 7\n\tv_add_f32 %5, %5, %17\n\tv_bfe_u32 %6, %6, %18, 7\n\tv_add_f32 %7, %7, %17\n\tv_bfe_u32 %0, %0, %18, 7\n\tv_add_f32 %1, %1, %17\n\tv_add_u32 %2, %2, %18\n\tv_add_u32 %3, %3, %18\n\tv_bfe_u32 %4, %4, %18, 7\n\tv_add_f32 %5, %5, %17\n\tv_bfe_u32 %6, %6, %18, 7\n\tv_add_f32 %7, %7, %17\n\tv_add_u32 %0, %0, %18\n\tv_add_f32 %1, %1, %17\n\tv_add_f32 %2, %2, %17\n\tv_add_f32 %3, %3, %17\n\tv_add_f32 %4, %4, %17\n\tv_add_f32 %5, %5, %17\n\tv_add_f32 %6, %6, %17\n\tv_add_f32 %7, %7, %17\n\tv_bfe_u32 %0, %0, %18, 7\n\tv_add_f32 %1, %1, %17\n\tv_bfe_u32 %2, %2, %18, 7\n\tv_add_f32 %3, %3, %17\n\tv_add_u32 %4, %4, %18\n\tv_add_u32 %5, %5, %18\n\tv_add_u32 %6, %6, %18\n\tv_add_u32 %7, %7, %18\n\tv_bfe_u32 %0, %0, %18, 7\n\tv_add_f32 %1, %1, %17\n\tv_bfe_u32 %2, %2, %18, 7\n\tv_add_f32 %3, %3, %17\n\tv_bfe_u32 %4, %4, %18, 7\n\tv_add_f32 %5, %5, %17\n\tv_bfe_u32 %6, %6, %18, 7\n\tv_add_f32 %7, %7, %17\n\tv_bfe_u32 %0, %0, %18, 7\n\tv_add_f32 %1, %1, %17\n\tv_add_u32 %2, %2, %18\n\tv_add_u32 %3, %3, %18\n\tv_bfe_u32 %4, %4, %18, 7\n\tv_add_f32 %5, %5, %17\n\tv_bfe_u32 %6, %6, %18, 7\n\tv_add_f32 %7, %7, %17\n\tv_add_u32 %0, %0, %18\n\tv_add_f32 %1, %1, %17\n\tv_add_f32 %2, %2, %17\n\tv_add_f32 %3, %3, %17\n\tv_add_f32 %4, %4, %17\n\tv_add_f32 %5, %5, %17\n\tv_add_f32 %6, %6, %17\n\tv_add_f32 %7, %7, %17\n\t");
        } else if (KIND == 126) {
            OPS("v_add_u32 %0, %0, %18\n\tv_add_u32 %1, %1, %18\n\tv_add_u32 %2, %2, %18\n\tv_add_u32 %3, %3, %18\n\tv_add_u32 %4, %4, %18\n\tv_add_u32 %5, %5, %18\n\tv_add_u32 %6, %6, %18\n\tv_add_u32 %7, %7, %18\n\tv_add_u32 %0, %0, %18\n\tv_add_u32 %1, %1, %18\n\tv_add_u32 %2, %2, %18\n\tv_add_u32 %3, %3, %18\n\tv_add_u32 %4, %4, %18\n\tv_add_u32 %5, %5, %18\n\tv_add_u32 %6, %6, %18\n\tv_add_u32 %7, %7, %18\n\tv_add_u32 %0, %0, %18\n\tv_add_u32 %1, %1, %18\n\tv_add_u32 %2, %2, %18\n\tv_add_u32 %3, %3, %18\n\tv_add_u32 %4, %4, %18\n\tv_add_u32 %5, %5, %18\n\tv_add_u32 %6, %6, %18\n\tv_add_u32 %7, %7, %18\n\tv_add_u32 %0, %0, %18\n\tv_add_u32 %1, %1, %18\n\tv_add_u32 %2, %2, %18\n\tv_add_u32 %3, %3, %18\n\tv_add_u32 %4, %4, %18\n\tv_add_u32 %5, %5, %18\n\tv_add_u32 %6, %6, %18\n\tv_add_u32 %7, %7, %18\n\tv_add_u32 %0, %0, %18\n\tv_add_u32 %1, %1, %18\n\tv_add_u32 %2, %2, %18\n\tv_add_u32 %3, %3, %18\n\tv_add_u32 %4, %4, %18\n\tv_add_u32 %5, %5, %18\n\tv_add_u32 %6, %6, %18\n\tv_add_u32 %7, %7, %18\n\tv_add_u32 %0, %0, %18\n\tv_add_u32 %1, %1, %18\n\tv_add_u32 %2, %2, %18\n\tv_add_u32 %3, %3, %18\n\tv_add_u32 %4, %4, %18\n\tv_add_u32 %5, %5, %18\n\tv_add_u32 %6, %6, %18\n\tv_add_u32 %7, %7, %18\n\tv_add_u32 %0, %0, %18\n\tv_add_u32 %1, %1, %18\n\tv_add_u32 %2, %2, %18\n\tv_add_u32 %3, %3, %18\n\tv_add_u32 %4, %4, %18\n\tv_add_u32 %5, %5, %18\n\tv_add_u32 %6, %6, %18\n\tv_add_u32 %7, %7, %18\n\tv_add_u32 %0, %0, %18\n\tv_add_u32 %1, %1, %18\n\tv_add_u32 %2, %2, %18\n\tv_add_u32 %3, %3, %18\n\tv_add_u32 %4, %4, %18\n\tv_add_u32 %5, %5, %18\n\tv_add_u32 %6, %6, %18\n\tv_add_u32 %7, %7, %18\n\t");
        } else if (KIND == 127) {
            OPS("v_bfe_u32 %0, %0, %18, 7\n\tv_add_u32 %1, %1, %18\n\tv_bfe_u32 %2, %2, %18, 7\n\tv_add_f32 %3, %3, %17\n\tv_bfe_u32 %4, %4, %18, 7\n\tv_add_u32 %5, %5, %18\n\tv_bfe_u32 %6, %6, %18, 7\n\tv_add_f32 %7, %7, %17\n\tv_bfe_u32 %0, %0, %18, 7\n\tv_add_u32 %1, %1, %18\n\tv_bfe_u32 %2, %2, %18, 7\n\tv_add_f32 %3, %3, %17\n\tv_bfe_u32 %4, %4, %18, 7\n\tv_add_u32 %5, %5, %18\n\tv_bfe_u32 %6, %6, %18, 7\n\tv_add_f32 %7, %7, %17\n\tv_bfe_u32 %0, %0, %18, 7\n\tv_add_u32 %1, %1, %18\n\tv_bfe_u32 %2, %2, %18, 7\n\tv_add_f32 %3, %3, %17\n\tv_bfe_u32 %4, %4, %18, 7\n\tv_add_u32 %5, %5, %18\n\tv_bfe_u32 %6, %6, %18, 7\n\tv_add_f32 %7, %7, %17\n\tv_bfe_u32 %0, %0, %18, 7\n\tv_add_u32 %1, %1, %18\n\tv_bfe_u32 %2, %2, %18, 7\n\tv_add_f32 %3, %3, %17\n\tv_bfe_u32 %4, %4, %18, 7\n\tv_add_u32 %5, %5, %18\n\tv_bfe_u32 %6, %6, %18, 7\n\tv_add_f32 %7, %7, %17\n\tv_bfe_u32 %0, %0, %18, 7\n\tv_add_u32 %1, %1, %18\n\tv_bfe_u32 %2, %2, %18, 7\n\tv_add_f32 %3, %3, %17\n\tv_bfe_u32 %4, %4, %18, 7\n\tv_add_u32 %5, %5, %18\n\tv_bfe_u32 %6, %6, %18, 7\n\tv_add_f32 %7, %7, %17\n\tv_bfe_u32 %0, %0, %18, 7\n\tv_add_u32 %1, %1, %18\n\tv_bfe_u32 %2, %2, %18, 7\n\tv_add_f32 %3, %3, %17\n\tv_bfe_u32 %4, %4, %18, 7\n\tv_add_u32 %5, %5, %18\n\tv_bfe_u32 %6, %6, %18, 7\n\tv_add_f32 %7, %7, %17\n\tv_bfe_u32 %0, %0, %18, 7\n\tv_add_u32 %1, %1, %18\n\tv_bfe_u32 %2, %2, %18, 7\n\tv_add_f32 %3, %3, %17\n\tv_bfe_u32 %4, %4, %18, 7\n\tv_add_u32 %5, %5, %18\n\tv_bfe_u32 %6, %6, %18, 7\n\tv_add_f32 %7, %7, %17\n\tv_bfe_u32 %0, %0, %18, 7\n\tv_add_u32 %1, %1, %18\n\tv_bfe_u32 %2, %2, %18, 7\n\tv_add_f32 %3, %3, %17\n\tv_bfe_u32 %4, %4, %18, 7\n\tv_add_u32 %5, %5, %18\n\tv_bfe_u32 %6, %6, %18, 7\n\tv_add_f32 %7, %7, %17\n\t");
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), q1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * 256 + threadIdx.x] = r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7 ^ sgl ^ (unsigned)(p0.x + p1.x + p2.x + p3.x + p4.y + p5.y + p6.y + p7.y);
    if (threadIdx.x == 0) {
        clk[2 * blockIdx.x] = t1 - t0;
        clk[2 * blockIdx.x + 1] = q1 - q0;
    }
}

static const char *kNames[] = {
    "v_fma_f32", "v_mul_f32", "v_add_f32", "v_add_u32", "v_xor_b32", "v_lshlrev_b32 (reg)", "v_lshrrev_b32 (imm)", "v_bfe_u32",
    "v_lshl_or_b32", "v_and_or_b32", "v_bfi_b32", "v_add_lshl_u32", "v_lshl_add_u32", "v_or3_b32", "v_min3_f32", "v_min_f32",
    "v_cvt_f32_i32", "v_cvt_flr_i32_f32", "v_ffbh_u32", "v_cndmask_b32 (vcc, VOP2)", "v_cndmask_b32 (sgpr pair, VOP3)",
    "v_cmp_eq_f32 -> vcc", "v_cmp_eq_f32 -> sgpr pair", "v_mov_b32", "v_pk_fma_f32", "v_pk_add_f32", "v_mul_u32_u24",
    "v_mad_u32_u24", "v_add_u32 + literal", "v_add_f32_e64 (VOP3, neg)", "v_addc_co_u32", "s_add_u32", "s_and_b64",
    "v_readlane_b32", "ds_write_b32", "ds_read_b32", "v_add_u32 / s_add_u32 1:1", "VALU / SALU 3:1", "v_sub_f32", "v_max3_f32",
    "v_min_i32", "v_not_b32", "v_fma_f32 (neg src)", "v_fmac_f32 (VOP2)", "v_alignbit_b32", "v_perm_b32", "v_cvt_f32_u32",
    "v_xad_u32", "v_add3_u32", "v_and_b32_sdwa", "v_mov_b32_dpp",
    "v_cndmask_b32_e64 (vcc as mask)", "v_cndmask_e32 vcc : v_add_u32 1:1", "v_cndmask_e32 vcc : v_add_u32 1:3", "v_cndmask sgpr : v_add_u32 1:1", "v_bfe_u32 : v_add_u32 1:1", "v_bfe_u32 : v_add_u32 1:3", "v_lshlrev_b32 (imm)", "v_lshrrev_b32 (reg)", "v_ashrrev_i32 (imm)", "v_and_b32", "v_or_b32", "v_sub_u32", "v_max_f32", "v_max_u32", "v_med3_f32", "v_cmp_lt_u32 -> vcc", "v_cmp_class_f32 -> vcc", "v_mul_lo_u32", "v_and_b32 + literal", "v_add_f32 (sgpr src)", "v_mul_f32 (inline const)", "v_xor_b32 (sgpr src)", "v_sub_f32_e64 abs/neg", "v_mul_f32 : v_bfe : v_cmp mix", "v_cvt_f32_ubyte0", "v_mbcnt_lo_u32_b32", "v_subrev_u32", "v_lshlrev_b32 (sgpr amount)", "v_mul_f32 (2 vgpr, distinct)", 
    "indep: bfe,addu (S F)", "indep: bfe,addu,xor,addu (S F F F)", "indep: bfe,bfe,addu,addu (S S F F)", "indep: bfe,addf (S Ff)", "indep: bfe,mulf,addf,fma (S Ff Ff Ff)", "indep: mulf,bfe,addf,cmp", "indep: cmp,addf (S Ff)", "indep: cmp,addu (S F)", "indep: cvt,addf (S Ff)", "indep: minf,addf (S Ff)", "indep: lshl_or,addu,addu (S F F)", "indep: addu,addf (F Ff)", "indep: 7F 1S (addf x7, bfe)", "indep: 4F 4S blocks", "indep: 16F 16S blocks", 
    "v_floor_f32", "v_ldexp_f32", "v_fract_f32", "v_trunc_f32", "v_rndne_f32", "v_mul_f32_e64 clamp", "v_fma_f32 clamp", "v_frexp_exp_i32_f32", "v_cvt_u32_f32", "v_mul_f32 literal", "indep: floor,addf,mulf,fma (S Ff Ff Ff)", "indep: bfe,addf,addf (S Ff Ff)", "indep: bfe,fma clamp (S Fc)", "indep: bfe,bfe,addf (S S Ff)", "indep: bfe,addu,addf (S I Ff)", "indep: addu,addf,addf (I Ff Ff)", "indep: xor,bfe,addf,addf (I S Ff Ff)", "indep: mov,bfe (I S)", "indep: lshrrev,bfe (I S)", "indep: and,bfe (I S)", "v_cmpx_ge_u32 (always true)",
    "indep: 4S 4I blocks", "indep: 2S 2I blocks", "indep: S F I F", "indep: S I F F", "indep: SFSFSFSF IIII FFFF", "indep: S I I I", "indep: S I I I F F F F", "walk-like: SSIIII SSSSS II SS I (9S 7I)", "walk-like grouped: 9S then 7I", "walk-like + 16 F beside: 9S 7I 16F", "v_add_u32 x8 (check)", "indep: S I S F"};
constexpr int kKinds = 128;

template <int K>
void launch(int blocks, unsigned *out, unsigned long long *clk, int iters) {
    hipLaunchKernelGGL(rate_kernel<K>, dim3(blocks), dim3(256), 1024, 0, out, clk, iters);
}

template <int K>
struct Table {
    static void fill(void (**t)(int, unsigned *, unsigned long long *, int)) {
        t[K] = &launch<K>;
        Table<K - 1>::fill(t);
    }
};
template <>
struct Table<-1> {
    static void fill(void (**)(int, unsigned *, unsigned long long *, int)) {}
};

int main(int argc, char **argv) {
    setvbuf(stdout, nullptr, _IONBF, 0);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) != hipSuccess) { printf("no device\n"); return 1; }
    const int cus = prop.multiProcessorCount;
    std::vector<int> wps_list;
    for (int i = 1; i < argc; i++) wps_list.push_back(atoi(argv[i]));
    if (wps_list.empty()) wps_list = {6, 1};
    unsigned *out;
    unsigned long long *clk;
    hipMalloc(&out, (size_t)cus * 8 * 256 * sizeof(unsigned));
    hipMalloc(&clk, (size_t)cus * 8 * 2 * sizeof(unsigned long long));
    void (*table[kKinds])(int, unsigned *, unsigned long long *, int);
    Table<kKinds - 1>::fill(table);
    printf("CUs %d, nominal clock %d kHz; one line per instruction kind and occupancy\n", cus, prop.clockRate);
    printf("%-34s %5s %9s %12s %12s %14s %14s\n", "kind", "w/SIMD", "wall ms", "clock GHz", "Ginstr/s/SIMD", "cyc@measured", "cyc@2.4GHz");
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int first = getenv("ISSUE_FIRST") ? atoi(getenv("ISSUE_FIRST")) : 0;
    for (int kind = first; kind < kKinds; kind++) {
        for (int wps : wps_list) {
            const int blocks = cus * wps;  // 256-thread workgroups: one wave per SIMD each
            const int iters = wps >= 4 ? 6000 : 12000;
            table[kind](blocks, out, clk, 200);  // warm-up
            if (hipDeviceSynchronize() != hipSuccess) { printf("sync failed: %s\n", hipGetErrorString(hipGetLastError())); return 1; }
            hipEventRecord(e0);
            table[kind](blocks, out, clk, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            std::vector<unsigned long long> h((size_t)blocks * 2);
            hipMemcpy(h.data(), clk, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
            double cyc = 0, real = 0;
            for (int i = 0; i < blocks; i++) { cyc += (double)h[2 * i]; real += (double)h[2 * i + 1]; }
            const double ghz = cyc / (real * 10.0);  // s_memrealtime: 100 MHz = 10 ns per tick
            const double per_block = (kind == 24 || kind == 25) ? 32.0 : 64.0;  // (the packed kinds: 32 instructions per block)
            const double instr_per_simd = (double)iters * per_block * wps;  // wave-instructions one SIMD executed
            const double per_us = instr_per_simd / (ms * 1000.0);
            printf("%-34s %5d %9.3f %12.3f %12.4f %14.3f %14.3f\n", kNames[kind], wps, ms, ghz, per_us / 1000.0,
                   ghz * 1000.0 / per_us, 2400.0 / per_us);
        }
    }
    return 0;
}
