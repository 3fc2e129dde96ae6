// tools/microbench/issue_latency.hip -- what ONE wavefront pays per instruction on MI355X (the progressive stream kernel is one
// wave per stream, so its speed is the issue latency of dependent scalar / vector / cross-unit sequences, not throughput).
// Every case is a loop of 256 copies of a short sequence, timed with s_memtime on a single wave; printed: cycles per copy.
//   hipcc --offload-arch=gfx950 -O3 -o issue_latency issue_latency.hip && ./issue_latency
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)
#define REP256(x) REP64(x) REP64(x) REP64(x) REP64(x)

#define CASE(idx, seq, ...)                                                                         \
    {                                                                                               \
        uint64_t t0, t1;                                                                            \
        asm volatile("s_waitcnt lgkmcnt(0) vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");   \
        asm volatile(REP256(seq) : __VA_ARGS__);                                                    \
        asm volatile("s_waitcnt lgkmcnt(0) vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");   \
        if (threadIdx.x == 0) out[idx] = t1 - t0;                                                   \
    }

__global__ void k(uint64_t *out, uint32_t seed) {
    __shared__ uint32_t lds[256];
    lds[threadIdx.x] = threadIdx.x * 4 % 252;
    __syncthreads();
    uint32_t s = seed & 7, s2 = 1, v = threadIdx.x, v2 = threadIdx.x ^ seed;
    uint64_t m = 0x0123456789abcdefull ^ seed;
    uint32_t a = (uint32_t)(uintptr_t)lds + (threadIdx.x * 4 % 252);
    for (int warm = 0; warm < 2; warm++) {
        CASE(0, "s_add_u32 %0, %0, 1\n\t", "+s"(s) : : "scc")                                                  // dependent SALU
        CASE(1, "s_add_u32 %0, %0, 1\n\ts_add_u32 %1, %1, 1\n\t", "+s"(s), "+s"(s2) : : "scc")                  // two independent SALU
        CASE(2, "v_add_u32 %0, 1, %0\n\t", "+v"(v))                                                            // dependent VALU
        CASE(3, "v_add_u32 %0, 1, %0\n\tv_add_u32 %1, 1, %1\n\t", "+v"(v), "+v"(v2))                           // two independent VALU
        CASE(4, "v_readlane_b32 %0, %1, %0\n\ts_and_b32 %0, %0, 63\n\t", "+s"(s) : "v"(v) : "scc")              // SALU -> readlane -> SALU chain
        CASE(5, "v_cmp_eq_u32_e32 vcc, %0, %2\n\ts_and_b64 %1, vcc, %1\n\ts_ff1_i32_b64 %0, %1\n\ts_and_b32 %0, %0, 63\n\t", "+s"(s), "+s"(m) : "v"(v) : "vcc", "scc")  // SALU -> v_cmp -> SALU
        CASE(6, "s_mov_b32 m0, %1\n\tv_writelane_b32 %0, %1, m0\n\t", "+v"(v) : "s"(s) : "m0")                  // writelane via M0
        CASE(7, "s_cmp_eq_u32 %0, 99\n\ts_cbranch_scc1 1f\n\t1:\n\t", : "s"(s) : "scc")                        // compare + branch not taken
        CASE(8, "s_branch 1f\n\ts_nop 0\n\t1:\n\t", : : )                                                      // taken branch (short)
        CASE(9, "ds_read_b32 %0, %0\n\ts_waitcnt lgkmcnt(0)\n\t", "+v"(a))                                     // dependent LDS read
        CASE(10, "ds_bpermute_b32 %0, %1, %0\n\ts_waitcnt lgkmcnt(0)\n\t", "+v"(v) : "v"(a))                    // dependent bpermute
        CASE(11, "s_mov_b64 exec, %1\n\tv_mov_b32 %0, %2\n\ts_mov_b64 exec, -1\n\t", "+v"(v) : "s"(m | 1), "s"(s))  // exec-masked move
        CASE(12, "v_readlane_b32 %0, %1, %0\n\ts_and_b32 %0, %0, 63\n\ts_add_u32 %2, %2, 1\n\ts_add_u32 %2, %2, 1\n\ts_add_u32 %2, %2, 1\n\ts_add_u32 %2, %2, 1\n\t", "+s"(s), "+v"(v), "+s"(s2) : : "scc")  // readlane chain + 4 SALU
        CASE(13, "s_bfe_u32 %1, %0, 0x40008\n\ts_add_u32 %0, %0, %1\n\t", "+s"(s), "+s"(s2) : : "scc")           // bfe + add chain
        CASE(14, "v_readfirstlane_b32 %0, %1\n\tv_add_u32 %1, %0, %1\n\t", "+s"(s), "+v"(v))                   // VALU -> SGPR -> VALU chain
        CASE(15, "s_nop 0\n\t", : : )
    }
    if (threadIdx.x == 0) out[16] = s + s2 + v + v2 + (uint32_t)m + a;
    if (threadIdx.x == 1) out[17] = v + v2 + a;
}

int main() {
    uint64_t *d, h[18];
    hipMalloc(&d, sizeof h);
    k<<<1, 64>>>(d, 5);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    const char *names[16] = {"dependent s_add", "2 independent s_add", "dependent v_add", "2 independent v_add", "v_readlane(sgpr lane) + s_and chain",
                             "v_cmp + s_and_b64 + s_ff1 + s_and chain", "s_mov m0 + v_writelane", "s_cmp + branch not taken", "s_branch taken",
                             "ds_read_b32 + wait (dependent)", "ds_bpermute + wait (dependent)", "s_mov exec + v_mov + s_mov exec", "readlane chain + 4 s_add",
                             "s_bfe + s_add chain", "v_readfirstlane + v_add chain", "s_nop 0"};
    for (int i = 0; i < 16; i++) printf("%-44s %7.2f cycles per copy (s_memtime ticks)\n", names[i], (double)h[i] / 256.0);
    return 0;
}
