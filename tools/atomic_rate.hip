// atomic_rate.hip -- diagnostic: throughput of the read-modify-writes the counting trace is made of, by scope and kind, on
// scattered words of a buffer the size of the benchmark tree (428 MB), from a grid shaped like the trace kernel's
// (256 CUs x 6 workgroups x 256 threads).  Every thread walks its own pseudo-random sequence of word indices; a "run" of R
// adjacent lanes shares a word group (R = 1: 64 distinct lines per wave-instruction; R = 8: 8 lanes in one 32-byte sector).
// Build: hipcc --offload-arch=gfx950 -O3 -o build_ab/atomic_rate tools/atomic_rate.hip
#include <hip/hip_runtime.h>
#pragma clang diagnostic ignored "-Wunused-value"
#pragma clang diagnostic ignored "-Wunused-result"

#include <cstdio>
#include <cstdlib>

enum { LOAD = 0, CAS_AGENT, CAS_WG, ADD_AGENT_NORET, ADD_WG_NORET, ADD_AGENT_RET, ADD_WG_RET, STORE, CAS_WG_PRIVATE, N_KINDS };
static const char *kNames[N_KINDS] = {"load", "cas agent (returning)", "cas workgroup scope", "add agent, no return", "add workgroup, no return",
                                      "add agent, returning", "add workgroup, returning", "store", "cas workgroup scope, per-XCD region"};

template <int KIND>
__global__ __launch_bounds__(256) void rate_kernel(uint32_t *buf, uint32_t n_words, int iters, int run_log2, int dependent, uint32_t *sink, int active = 64) {
    if ((int)(threadIdx.x & 63u) >= active) return;
    uint32_t x = (blockIdx.x * 256u + threadIdx.x) >> run_log2;
    x = x * 2654435761u + 12345u;
    uint32_t acc = 0;
    uint32_t base = 0, span = n_words;
    if (KIND == CAS_WG_PRIVATE) {  // XCD = blockIdx % 8 (round-robin dispatch): one eighth of the buffer each
        span = n_words / 8u;
        base = (blockIdx.x & 7u) * span;
    }
    for (int i = 0; i < iters; i++) {
        x = x * 1664525u + 1013904223u;
        const uint32_t p = base + (uint32_t)(((uint64_t)(x ^ (dependent ? (acc & 1u) : 0u)) * span) >> 32);
        uint32_t r = 0;
        if (KIND == LOAD) r = __builtin_nontemporal_load(buf + p);
        if (KIND == CAS_AGENT) { uint32_t e = 0; __hip_atomic_compare_exchange_strong(buf + p, &e, 1u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); r = e; }
        if (KIND == CAS_WG || KIND == CAS_WG_PRIVATE) { uint32_t e = 0; __hip_atomic_compare_exchange_strong(buf + p, &e, 1u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); r = e; }
        if (KIND == ADD_AGENT_NORET) (void)__hip_atomic_fetch_add(buf + p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (KIND == ADD_WG_NORET) (void)__hip_atomic_fetch_add(buf + p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (KIND == ADD_AGENT_RET) r = __hip_atomic_fetch_add(buf + p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (KIND == ADD_WG_RET) r = __hip_atomic_fetch_add(buf + p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (KIND == STORE) buf[p] = x;
        acc += r;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int KIND>
void run(uint32_t *buf, uint32_t n_words, uint32_t *sink, int cus) {
    for (int dependent = 0; dependent < 2; dependent++) {
        if (dependent && (KIND == ADD_AGENT_NORET || KIND == ADD_WG_NORET || KIND == STORE)) continue;
        for (int run_log2 : {0, 3}) {
            const int iters = 200, blocks = cus * 6;
            hipMemset(buf, 0, (size_t)n_words * 4);
            hipEvent_t e0, e1;
            hipEventCreate(&e0);
            hipEventCreate(&e1);
            hipLaunchKernelGGL(rate_kernel<KIND>, dim3(blocks), dim3(256), 0, 0, buf, n_words, 20, run_log2, dependent, sink);
            hipEventRecord(e0);
            hipLaunchKernelGGL(rate_kernel<KIND>, dim3(blocks), dim3(256), 0, 0, buf, n_words, iters, run_log2, dependent, sink);
            hipEventRecord(e1);
            if (hipEventSynchronize(e1) != hipSuccess) { printf("%s failed\n", kNames[KIND]); exit(1); }
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            const double ops = (double)blocks * 256 * iters;
            printf("%-38s %s lanes per word-run %d: %8.3f ms, %7.2f G lane-ops/s, %6.2f G wave-instructions/s\n", kNames[KIND],
                   dependent ? "dependent  " : "independent", 1 << run_log2, ms, ops / ms * 1e-6, ops / 64 / ms * 1e-6);
        }
    }
}

// sparse wave-instructions (the counting trace issues its compare-and-swaps from a few lanes of a wave at a time) and the
// latency of one dependent chain
template <int KIND>
void run_sparse(uint32_t *buf, uint32_t n_words, uint32_t *sink, int cus) {
    for (int active : {1, 4, 16, 64}) {
        const int iters = 200, blocks = cus * 6;
        hipMemset(buf, 0, (size_t)n_words * 4);
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipLaunchKernelGGL(rate_kernel<KIND>, dim3(blocks), dim3(256), 0, 0, buf, n_words, 20, 0, 1, sink, active);
        hipEventRecord(e0);
        hipLaunchKernelGGL(rate_kernel<KIND>, dim3(blocks), dim3(256), 0, 0, buf, n_words, iters, 0, 1, sink, active);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        const double wi = (double)blocks * 4 * iters;
        printf("%-38s dependent, %2d active lanes per wave: %8.3f ms, %7.2f G lane-ops/s, %6.2f G wave-instructions/s, %6.0f ns per dependent op\n", kNames[KIND], active, ms,
               wi * active / ms * 1e-6, wi / ms * 1e-6, ms * 1e6 / iters);
    }
    {  // one wave alone: latency
        const int iters = 2000;
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipLaunchKernelGGL(rate_kernel<KIND>, dim3(1), dim3(64), 0, 0, buf, n_words, 20, 0, 1, sink, 1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(rate_kernel<KIND>, dim3(1), dim3(64), 0, 0, buf, n_words, iters, 0, 1, sink, 1);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        printf("%-38s one lane of one wave, dependent chain: %6.0f ns per op\n", kNames[KIND], ms * 1e6 / iters);
    }
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const uint32_t n_words = 107u << 20;  // 428 MB
    uint32_t *buf, *sink;
    hipMalloc(&buf, (size_t)n_words * 4);
    hipMalloc(&sink, 64);
    const int cus = prop.multiProcessorCount;
    run<LOAD>(buf, n_words, sink, cus);
    run<STORE>(buf, n_words, sink, cus);
    run<CAS_AGENT>(buf, n_words, sink, cus);
    run<CAS_WG>(buf, n_words, sink, cus);
    run<CAS_WG_PRIVATE>(buf, n_words, sink, cus);
    run<ADD_AGENT_NORET>(buf, n_words, sink, cus);
    run<ADD_WG_NORET>(buf, n_words, sink, cus);
    run<ADD_AGENT_RET>(buf, n_words, sink, cus);
    run<ADD_WG_RET>(buf, n_words, sink, cus);
    run_sparse<LOAD>(buf, n_words, sink, cus);
    run_sparse<CAS_AGENT>(buf, n_words, sink, cus);
    run_sparse<ADD_AGENT_RET>(buf, n_words, sink, cus);
    return 0;
}
