// tools/microbench/simultaneous_writeback.hip -- two XCDs hold DIFFERENT bytes of the same 32-byte sector dirty and write their L2s
// back at the same moment: does the memory side merge both?  (line_sharing.hip orders the two write-backs; the progressive
// launch does not: the DC scans store coefficient 0 of a block while an AC scan, on another XCD, stores coefficients 1..63.)
// Workgroup 0 stores 0xA000 | iteration into bytes 0-1 of each of 4096 lines, workgroup `b_wg` 0xB000 | iteration into bytes
// `off`..`off`+1; both wait for each other, release together, then workgroup 0 reads every line back (agent scope) and counts
// halves that are not there.  1000 iterations per setting.
//   hipcc --offload-arch=gfx950 -O3 -o simultaneous_writeback simultaneous_writeback.hip && ./simultaneous_writeback
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

constexpr int kLines = 4096, kIters = 1000;

__device__ __forceinline__ void st16(void *p, uint32_t v) { asm volatile("global_store_short %0, %1, off" : : "v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void release() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
__device__ __forceinline__ void arrive_and_wait(uint32_t *counter, uint32_t target) {  // both workgroups, all lanes
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void k(uint8_t *lines, uint32_t *sync, uint32_t off, uint32_t b_wg, uint32_t *lost) {
    const bool is_a = blockIdx.x == 0, is_b = blockIdx.x == b_wg;
    if (!is_a && !is_b) return;
    uint32_t lost_a = 0, lost_b = 0;
    for (uint32_t it = 1; it <= (uint32_t)kIters; it++) {
        for (uint32_t l = threadIdx.x; l < (uint32_t)kLines; l += 256) st16(lines + (size_t)l * 128 + (is_a ? 0u : off), (is_a ? 0xA000u : 0xB000u) | (it & 0xFFFu));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        arrive_and_wait(&sync[0], 2 * (3 * it - 2));  // both have stored (their lines are dirty in their own L2)
        release();                                    // ... and write back at the same moment
        arrive_and_wait(&sync[0], 2 * (3 * it - 1));
        if (is_a) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            for (uint32_t l = threadIdx.x; l < (uint32_t)kLines; l += 256) {
                const uint32_t va = __hip_atomic_load(reinterpret_cast<uint16_t *>(lines + (size_t)l * 128), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const uint32_t vb = __hip_atomic_load(reinterpret_cast<uint16_t *>(lines + (size_t)l * 128 + off), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                lost_a += va != (0xA000u | (it & 0xFFFu));
                lost_b += vb != (0xB000u | (it & 0xFFFu));
            }
        }
        arrive_and_wait(&sync[0], 2 * (3 * it));
    }
    if (is_a) {
        atomicAdd(&lost[0], lost_a);
        atomicAdd(&lost[1], lost_b);
    }
}

int main() {
    uint8_t *d_lines;
    uint32_t *d_sync, *d_lost, h_lost[2];
    (void)hipMalloc(&d_lines, (size_t)kLines * 128);
    (void)hipMalloc(&d_sync, 64);
    (void)hipMalloc(&d_lost, 64);
    printf("%d iterations x %d lines; halves missing after both XCDs released at the same moment\n", kIters, kLines);
    for (uint32_t off : {2u, 4u, 30u, 64u})
        for (uint32_t b_wg : {1u, 3u, 5u, 8u}) {
            (void)hipMemset(d_lines, 0, (size_t)kLines * 128);
            (void)hipMemset(d_sync, 0, 64);
            (void)hipMemset(d_lost, 0, 64);
            (void)hipDeviceSynchronize();
            k<<<16, 256>>>(d_lines, d_sync, off, b_wg, d_lost);
            (void)hipDeviceSynchronize();
            (void)hipMemcpy(h_lost, d_lost, 8, hipMemcpyDeviceToHost);
            printf("offset %2u, B = workgroup %u: A's halves missing %u, B's halves missing %u\n", off, b_wg, h_lost[0], h_lost[1]);
        }
    return 0;
}
