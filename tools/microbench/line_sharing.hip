// tools/microbench/line_sharing.hip -- what happens to a 128-byte line two XCDs write different bytes of?
// The pipelined progressive launch lets scans on different XCDs store into the same coefficient block (= one line) at about
// the same time: the DC refinement reads coefficient 0 and stores it back, an AC scan stores coefficient 1..63.  The eight
// L2s are not coherent with each other; whether that is safe depends on how a line that was READ and then partly WRITTEN is
// written back: under a byte mask (only the bytes this XCD stored), or whole (with the other bytes as they were when read).
//
// Workgroup A (XCD 0) and workgroup B (another XCD: workgroups go round the XCDs) share N lines, lane i of each the i-th:
//   1. A loads 2 bytes at offset 0 of its line                      (the line is now valid in A's L2)
//   2. B stores 0xBBBB at offset `off` of the same line, releases (buffer_wbl2 + wait), says so
//   3. A stores 0xAAAA at offset 0                                  (the line is dirty in A's L2), releases, says so
//   4. the host reads the lines: offset `off` should hold 0xBBBB; 0 means A's write-back took B's bytes with it
// for off = 2 (same dword), 4 (same 32-byte sector), 32 (next sector), 64 (other half of the line); and the same with step 1
// left out (A only stores: a partial line).
//   hipcc --offload-arch=gfx950 -O3 -o line_sharing line_sharing.hip && ./line_sharing
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

// plain cached accesses, as the kernels make them (a `volatile` access would be sc0 sc1: system scope, past the caches)
__device__ __forceinline__ uint32_t ld16(const void *p) {
    uint32_t v;
    asm volatile("global_load_ushort %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ void st16(void *p, uint32_t v) { asm volatile("global_store_short %0, %1, off" : : "v"(p), "v"(v) : "memory"); }

__device__ __forceinline__ void wait_for(uint32_t *flag, uint32_t v) {
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < v) __builtin_amdgcn_s_sleep(8);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
__device__ __forceinline__ void say(uint32_t *flag, uint32_t v) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(flag, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// And the READER's side (kernel r): does A, after its acquire (buffer_inv), see what B stored into a line A holds
//   R1 clean (A only read it before),  R2 dirty (A has stored offset 0 into it and not released),  R3 dirty, released, then acquired?
__global__ __launch_bounds__(256) void r(uint8_t *lines, uint32_t *flags, uint32_t off, int variant, uint32_t b_wg, uint32_t *seen_out) {
    uint8_t *mine0 = lines + (size_t)threadIdx.x * 128;
    uint8_t *mine_off = lines + (size_t)threadIdx.x * 128 + off;
    if (blockIdx.x == 0) {
        const uint32_t first = ld16(mine_off);  // the line is valid in A's L2 (and B's bytes are still zero)
        if (variant >= 2) st16(mine0, 0xAAAAu);
        say(&flags[0], 1);  // (a release: writes A's dirty bytes back -- B has not stored yet)
        if (variant == 2) st16(mine0, 0xA5A5u);  // dirty again, and not released before the read below
        wait_for(&flags[1], 1);  // B has stored and released; wait_for ends with the acquire
        if (variant == 3) {
            st16(mine0, 0xA5A5u);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        seen_out[threadIdx.x] = ld16(mine_off) | (first << 16);
    } else if (blockIdx.x == b_wg) {
        wait_for(&flags[0], 1);
        st16(mine_off, 0xBBBBu);
        say(&flags[1], 1);
    }
}

// grid: 16 workgroups of 256; workgroup 0 is A, workgroup `b_wg` is B, the others leave at once
__global__ __launch_bounds__(256) void k(uint8_t *lines, uint32_t *flags, uint32_t off, int a_loads, uint32_t b_wg, uint32_t *sink) {
    uint8_t *mine0 = lines + (size_t)threadIdx.x * 128;
    uint8_t *mine_off = lines + (size_t)threadIdx.x * 128 + off;
    if (blockIdx.x == 0) {
        uint32_t seen = 0;
        if (a_loads) seen = ld16(mine0);
        say(&flags[0], 1);
        wait_for(&flags[1], 1);
        st16(mine0, 0xAAAAu | seen);
        say(&flags[2], 1);
        if (seen == 0x1234u) sink[0] = 1;
    } else if (blockIdx.x == b_wg) {
        wait_for(&flags[0], 1);
        st16(mine_off, 0xBBBBu);
        say(&flags[1], 1);
    }
}

int main() {
    const int n = 256;
    uint8_t *d_lines;
    uint32_t *d_flags, *d_sink;
    (void)hipMalloc(&d_lines, n * 128);
    (void)hipMalloc(&d_flags, 64);
    (void)hipMalloc(&d_sink, 64);
    static uint8_t h[n * 128];
    printf("offset of B's store, A reads the line first?, B's workgroup: lines (of %d) in which B's bytes survived / A's bytes arrived\n", n);
    for (int a_loads = 1; a_loads >= 0; a_loads--)
        for (uint32_t off : {2u, 4u, 32u, 64u})
            for (uint32_t b_wg : {1u, 3u, 8u}) {  // 8: the same XCD as workgroup 0 if workgroups go round eight XCDs
                (void)hipMemset(d_lines, 0, n * 128);
                (void)hipMemset(d_flags, 0, 64);
                (void)hipDeviceSynchronize();
                k<<<16, 256>>>(d_lines, d_flags, off, a_loads, b_wg, d_sink);
                (void)hipDeviceSynchronize();
                (void)hipMemcpy(h, d_lines, n * 128, hipMemcpyDeviceToHost);
                int b_ok = 0, a_ok = 0;
                for (int i = 0; i < n; i++) {
                    uint16_t vb, va;
                    memcpy(&vb, h + i * 128 + off, 2);
                    memcpy(&va, h + i * 128, 2);
                    b_ok += vb == 0xBBBB;
                    a_ok += va == 0xAAAA;
                }
                printf("off %2u  A %s  B = workgroup %u: B's bytes %3d, A's bytes %3d\n", off, a_loads ? "loads, then stores" : "only stores      ", b_wg, b_ok, a_ok);
            }
    uint32_t *d_seen, h_seen[256];
    (void)hipMalloc(&d_seen, sizeof h_seen);
    printf("reader's side: lanes (of %d) of A that read B's bytes after the acquire\n", n);
    for (int variant = 1; variant <= 3; variant++)
        for (uint32_t off : {2u, 64u})
            for (uint32_t b_wg : {1u, 3u, 8u}) {
                (void)hipMemset(d_lines, 0, n * 128);
                (void)hipMemset(d_flags, 0, 64);
                (void)hipDeviceSynchronize();
                r<<<16, 256>>>(d_lines, d_flags, off, variant, b_wg, d_seen);
                (void)hipDeviceSynchronize();
                (void)hipMemcpy(h_seen, d_seen, sizeof h_seen, hipMemcpyDeviceToHost);
                int ok = 0;
                for (int i = 0; i < n; i++) ok += (h_seen[i] & 0xFFFFu) == 0xBBBBu;
                printf("R%d (%s) off %2u  B = workgroup %u: %3d\n", variant,
                       variant == 1 ? "line clean in A's L2" : (variant == 2 ? "line dirty in A's L2, acquire only" : "dirty, release + acquire"), off, b_wg, ok);
            }
    return 0;
}
