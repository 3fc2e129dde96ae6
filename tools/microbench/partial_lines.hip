// tools/microbench/partial_lines.hip -- what do PARTIAL 128-byte lines cost on MI355X?  The coefficient hand-off between the
// Huffman kernel (K2) and the IDCT kernel (K3) is one 128-byte line per block, mostly zeros behind the block's last non-zero
// coefficient.  If K2 wrote and K3 read only the first `ext` 32-byte sectors of every line (ext per block in a byte array),
// would HBM charge for the sectors or for the lines?
//   k3like: tiles of 256 consecutive blocks; lane (block, 16-byte piece) reads its piece when piece < 2 * ext[block] and
//           always writes 16 bytes of a dense output (K3's traffic mix: a sparse read + a dense write of the same size)
//   k2like: a wave owns 64 blocks at a stride of 24 blocks (one lane per restart interval of 4 MCUs x 6 blocks) and writes the
//           first ext sectors of each, 8 lanes per block, as K2's flush does
// Build: hipcc --offload-arch=gfx950 -O3 partial_lines.hip -o partial_lines ; run: ./partial_lines [blocks_in_millions]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

__global__ __launch_bounds__(256) void k3like(const uint4 *__restrict__ src, const uint8_t *__restrict__ ext, uint4 *__restrict__ dst,
                                              size_t n_blocks) {
    const size_t tile = (size_t)blockIdx.x * 256;
    const uint32_t tid = threadIdx.x;
    uint4 v[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const size_t b = tile + k * 32 + (tid >> 3);
        const uint32_t piece = tid & 7;
        v[k] = uint4{0, 0, 0, 0};
        if (b < n_blocks && piece < 2u * ext[b]) v[k] = src[b * 8 + piece];
    }
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const size_t b = tile + k * 32 + (tid >> 3);
        if (b < n_blocks) dst[b * 8 + (tid & 7)] = v[k];
    }
}

// one wave = 64 lanes = 64 intervals; interval i owns blocks [i * 24, i * 24 + 24); step j of the wave flushes block j of
// every lane's interval: 8 passes of (8 blocks x 8 pieces)
__global__ __launch_bounds__(256) void k2like(uint4 *__restrict__ dst, const uint8_t *__restrict__ ext, size_t n_blocks, int delay) {
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t first_interval = ((size_t)blockIdx.x * 4 + wave) * 64;
    uint4 v = {threadIdx.x, blockIdx.x, 3, 4};
    for (int j = 0; j < 24; j++) {
#pragma unroll
        for (int it = 0; it < 8; it++) {
            const uint32_t blk = it * 8 + (lane >> 3), piece = lane & 7;
            const size_t b = (first_interval + blk) * 24 + j;
            if (b < n_blocks && piece < 2u * ext[b]) dst[b * 8 + piece] = v;
        }
        float f = (float)v.x;  // stand-in for decoding the next block
        for (int i = 0; i < delay; i++) f = f * 1.0001f + 0.5f;
        v.w = (uint32_t)f;
    }
}

int main(int argc, char **argv) {
    const size_t n_blocks = (size_t)(argc > 1 ? atoi(argv[1]) : 100) * 1000000;
    const int delay = argc > 2 ? atoi(argv[2]) : 300;
    uint4 *src, *dst;
    uint8_t *ext;
    if (hipMalloc(&src, n_blocks * 128) != hipSuccess || hipMalloc(&dst, n_blocks * 128) != hipSuccess || hipMalloc(&ext, n_blocks) != hipSuccess) return 1;
    hipMemset(src, 1, n_blocks * 128);
    std::vector<uint8_t> h(n_blocks);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    // extent mixes: all 4 sectors (today), the headline batch's histogram (41 % 1, 56 % 2, 3 % 3), all 2, all 1
    const char *names[] = {"all 4 sectors (128 B)", "4K Q75 mix (mean 1.62 sectors)", "all 2 sectors (64 B)", "all 1 sector (32 B)", "1080p Q90 mix (mean 2.8)"};
    for (int mix = 0; mix < 5; mix++) {
        uint64_t s = 88172645463325252ull;
        double mean = 0;
        for (size_t i = 0; i < n_blocks; i++) {
            s ^= s << 13, s ^= s >> 7, s ^= s << 17;
            const uint32_t r = (uint32_t)(s >> 33) % 1000;
            uint8_t e = 4;
            if (mix == 1) e = r < 411 ? 1 : (r < 968 ? 2 : (r < 997 ? 3 : 4));
            if (mix == 2) e = 2;
            if (mix == 3) e = 1;
            if (mix == 4) e = r < 331 ? 1 : (r < 343 ? 2 : (r < 527 ? 3 : 4));
            h[i] = e;
            mean += e;
        }
        mean /= (double)n_blocks;
        hipMemcpy(ext, h.data(), n_blocks, hipMemcpyHostToDevice);
        float best3 = 1e9f, best2 = 1e9f;
        for (int rep = 0; rep < 4; rep++) {
            float ms;
            hipEventRecord(e0);
            hipLaunchKernelGGL(k3like, dim3((unsigned)((n_blocks + 255) / 256)), dim3(256), 0, 0, src, ext, dst, n_blocks);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
            if (ms < best3) best3 = ms;
            hipEventRecord(e0);
            hipLaunchKernelGGL(k2like, dim3((unsigned)((n_blocks / 24 + 255) / 256)), dim3(256), 0, 0, src, ext, n_blocks, delay);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
            if (ms < best2) best2 = ms;
        }
        const double dense = (double)n_blocks * 128, sparse = (double)n_blocks * 32 * mean + (double)n_blocks;
        printf("%-34s k3like %.3f ms (%.2f TB/s of sector bytes, %.2f of line bytes)   k2like %.3f ms (%.2f TB/s sector bytes, %.2f line bytes)\n",
               names[mix], best3, (dense + sparse) / best3 / 1e9, 2 * dense / best3 / 1e9, best2, sparse / best2 / 1e9, dense / best2 / 1e9);
    }
    return 0;
}
