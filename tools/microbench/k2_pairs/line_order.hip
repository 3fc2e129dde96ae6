// tools/microbench/k2_pairs/line_order.hip -- is K2 slowed down by the ORDER in which it writes its whole 128-byte lines?
// (round 6: with the pair lookup the wave's symbol steps fell by 37 % and huffman_decode_kernel did not get faster: per-wave
// cycle counters put 36 % of a wave's life into the flush and 12 % into the ring top-up, whose wait also covers the stores.)
// K2's order: a wave owns 64 consecutive restart intervals = one contiguous span of 64 x 24 blocks x 128 B = 192 KB, and every
// block step writes ONE line in each of the 64 intervals -- 64 lines at a stride of 3 KB; the neighbours of a line arrive one block
// step (~6 us) later each, 24 steps to fill the span.  Variants, all writing the same bytes with the same wave / workgroup shape
// (11 waves, a workgroup per CU at a time because of LDS) and the same stand-in for the decode between two steps:
//   k2        the order above
//   stream    every step writes 64 CONSECUTIVE lines (8 KB) of the wave's span
//   pairs     every step writes 2 consecutive lines in each of 32 intervals (what staging two blocks per lane would give)
// Build: hipcc --offload-arch=gfx950 -O3 line_order.hip -o line_order ; run: ./line_order [delay] [n_images]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

constexpr int kIntervals = 8100, kBlocksPerInterval = 24, kWaves = 11;

template <int MODE>
__global__ __launch_bounds__(64 * kWaves) void writer(uint8_t *out, int n_images, int delay, int lds_bytes) {
    extern __shared__ uint8_t smem[];
    const int wg_per_img = (kIntervals + 64 * kWaves - 1) / (64 * kWaves);
    const int img = blockIdx.x / wg_per_img;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int first = ((blockIdx.x % wg_per_img) * kWaves + wave) * 64;
    if (first >= kIntervals) return;
    if (lds_bytes > 64 && threadIdx.x == 0) smem[lds_bytes - 1] = 1;  // (touch: the allocation is what limits the CU to one workgroup)
    uint8_t *span = out + ((size_t)img * kIntervals + first) * kBlocksPerInterval * 128;
    const int n_own = kIntervals - first < 64 ? kIntervals - first : 64;
    uint4 v = {threadIdx.x, blockIdx.x, 3, 4};
    for (int step = 0; step < kBlocksPerInterval; step++) {
        float f = (float)v.x;
        for (int i = 0; i < delay; i++) f = f * 1.0001f + 0.5f;  // stand-in for decoding the block
        v.w = (uint32_t)f;
#pragma unroll
        for (int it = 0; it < 8; it++) {
            const int idx = it * 8 + (lane >> 3), chunk = lane & 7;  // idx = 0..63: which of the step's 64 lines
            size_t line;
            if (MODE == 0) line = (size_t)idx * kBlocksPerInterval + step;                                    // one line per interval
            else if (MODE == 1) line = (size_t)step * 64 + idx;                                               // 64 consecutive lines
            else line = (size_t)((step & 1) * 32 + (idx >> 1)) * kBlocksPerInterval + (step >> 1) * 2 + (idx & 1);  // 2 lines in 32 intervals
            if (line < (size_t)n_own * kBlocksPerInterval) *reinterpret_cast<uint4 *>(span + line * 128 + chunk * 16) = v;
        }
    }
}

int main(int argc, char **argv) {
    const int delay = argc > 1 ? atoi(argv[1]) : 400;
    const int n_images = argc > 2 ? atoi(argv[2]) : 512;
    const size_t bytes = (size_t)n_images * kIntervals * kBlocksPerInterval * 128;
    uint8_t *d;
    if (hipMalloc(&d, bytes + 4096) != hipSuccess) return 1;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int wg_per_img = (kIntervals + 64 * kWaves - 1) / (64 * kWaves);
    const int lds = argc > 3 ? atoi(argv[3]) * 1024 : 150 * 1024;  // (150 KB: one workgroup per CU, K2's occupancy)
    hipFuncSetAttribute(reinterpret_cast<const void *>(&writer<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void *>(&writer<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void *>(&writer<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const char *names[3] = {"k2 (64 lines, 3 KB apart)", "stream (64 consecutive lines)", "pairs (2 lines x 32 intervals)"};
    for (int mode = 0; mode < 3; mode++)
        for (int rep = 0; rep < 3; rep++) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(writer<0>, dim3(n_images * wg_per_img), dim3(64 * kWaves), lds, 0, d, n_images, delay, lds);
            if (mode == 1) hipLaunchKernelGGL(writer<1>, dim3(n_images * wg_per_img), dim3(64 * kWaves), lds, 0, d, n_images, delay, lds);
            if (mode == 2) hipLaunchKernelGGL(writer<2>, dim3(n_images * wg_per_img), dim3(64 * kWaves), lds, 0, d, n_images, delay, lds);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (rep == 2) printf("delay %5d  %-32s %8.3f ms  %7.1f GB/s\n", delay, names[mode], ms, bytes / ms / 1e6);
        }
    return 0;
}
