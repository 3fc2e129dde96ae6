// tools/microbench/fetch_rate.hip -- how fast ONE wave runs straight-line code of a given footprint on MI355X: a loop whose body
// is n copies of a 4-byte VALU (or SALU) instruction, iterated until ~256K instructions have run; s_memtime ticks per
// instruction.  A body that fits the instruction cache is fetched from it on every trip but the first; issue_latency.hip's
// 256-copy cases (each executed twice) measured 4.17 ticks per 4-byte instruction, which is what this tool puts in context:
// is that the machine's issue rate for one wave, or the instruction FETCH of code that is not resident?
//   hipcc --offload-arch=gfx950 -O3 -o fetch_rate fetch_rate.hip && ./fetch_rate [waves_per_simd]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

__device__ __forceinline__ uint64_t tick() {
    uint64_t t;
    asm volatile("s_waitcnt lgkmcnt(0) vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}

#define BODY(n, text) asm volatile(".rept " #n "\n\t" text "\n\t.endr" : "+v"(v), "+v"(v2), "+s"(s) : : "scc")

#define CASE(idx, n, text)                                              \
    {                                                                   \
        const int trips = (256 * 1024) / (n);                           \
        _Pragma("unroll 1") for (int warm = 0; warm < 2; warm++) {      \
            const uint64_t t0 = tick();                                 \
            _Pragma("unroll 1") for (int i = 0; i < trips; i++) BODY(n, text); \
            const uint64_t t1 = tick();                                 \
            if (threadIdx.x == 0 && blockIdx.x == 0) out[idx] = t1 - t0; \
        }                                                               \
    }

template <int kCase>
__global__ void k(uint64_t *out, uint32_t seed) {
    uint32_t v = threadIdx.x, v2 = seed, s = seed;
    constexpr int idx = kCase;
    if (kCase == 0) CASE(idx, 16, "v_add_u32 %0, 1, %0")
    if (kCase == 1) CASE(idx, 64, "v_add_u32 %0, 1, %0")
    if (kCase == 2) CASE(idx, 256, "v_add_u32 %0, 1, %0")
    if (kCase == 3) CASE(idx, 1024, "v_add_u32 %0, 1, %0")
    if (kCase == 4) CASE(idx, 4096, "v_add_u32 %0, 1, %0")
    if (kCase == 5) CASE(idx, 12288, "v_add_u32 %0, 1, %0")
    if (kCase == 6) CASE(idx, 256, "s_add_u32 %2, %2, 1")
    if (kCase == 7) CASE(idx, 4096, "s_add_u32 %2, %2, 1")
    if (kCase == 8) CASE(idx, 256, "v_add_u32 %0, 1, %0\n\tv_add_u32 %1, 1, %1")
    if (kCase == 9) CASE(idx, 2048, "v_add_u32 %0, 1, %0\n\tv_add_u32 %1, 1, %1")
    if (kCase == 10) CASE(idx, 256, "v_add_u32 %0, 1, %0\n\ts_add_u32 %2, %2, 1")
    if (kCase == 11) CASE(idx, 2048, "v_add_u32 %0, 1, %0\n\ts_add_u32 %2, %2, 1")
    if (threadIdx.x == 0 && blockIdx.x == 0) out[15] = v + v2 + s;
}

template <int kCase>
void launch_all(uint64_t *d, int waves) {
    k<kCase><<<1, 64 * waves>>>(d, 5);  // waves > 1: that many waves of one workgroup, spread over the CU's four SIMDs
    if constexpr (kCase < 11) launch_all<kCase + 1>(d, waves);
}

int main(int argc, char **argv) {
    const int waves = argc > 1 ? atoi(argv[1]) : 1;
    uint64_t *d, h[16];
    (void)hipMalloc(&d, sizeof h);
    launch_all<0>(d, waves);
    (void)hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    const char *names[12] = {"v_add x 16 (64 B body)", "v_add x 64 (256 B)", "v_add x 256 (1 KiB)", "v_add x 1024 (4 KiB)", "v_add x 4096 (16 KiB)",
                             "v_add x 12288 (48 KiB)", "s_add x 256 (1 KiB)", "s_add x 4096 (16 KiB)", "2 independent v_add x 256 (2 KiB)",
                             "2 independent v_add x 2048 (16 KiB)", "v_add + s_add x 256 (2 KiB)", "v_add + s_add x 2048 (16 KiB)"};
    const int per[12] = {1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2};
    printf("%d wave(s) in the workgroup\n", waves);
    for (int i = 0; i < 12; i++) printf("%-40s %6.2f ticks per instruction\n", names[i], (double)h[i] / (256.0 * 1024.0 * per[i]));
    return 0;
}
