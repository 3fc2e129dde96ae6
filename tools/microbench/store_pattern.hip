// tools/microbench/store_pattern.hip -- how fast can a wave write interleaved 4:2:0 output if every lane owns the MCUs of
// one restart interval (48-byte pieces at 192-byte stride, the neighbouring pieces arriving ~tens of microseconds later)?
// Build: hipcc --offload-arch=gfx950 -O3 store_pattern.hip -o store_pattern ; run: ./store_pattern [delay_iters]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

constexpr int W = 3840, H = 2160, MX = 240, MY = 135, MCUS = MX * MY, DRI = 4, INTERVALS = MCUS / DRI;

__global__ __launch_bounds__(640) void scattered(uint8_t *out, int n_images, int delay) {
    const int wg_per_img = (INTERVALS + 639) / 640;
    const int img = blockIdx.x / wg_per_img;
    const int interval = (blockIdx.x % wg_per_img) * 640 + threadIdx.x;
    if (interval >= INTERVALS) return;
    uint8_t *base = out + (size_t)img * W * H * 3;
    uint4 v = {threadIdx.x, blockIdx.x, 3, 4};
    for (int m = 0; m < DRI; m++) {
        const int mcu = interval * DRI + m;
        const int my = mcu / MX, mx = mcu % MX;
        for (int r = 0; r < 16; r++) {
            uint8_t *p = base + ((size_t)(my * 16 + r) * W + mx * 16) * 3;
#pragma unroll
            for (int q = 0; q < 3; q++) *reinterpret_cast<uint4 *>(p + q * 16) = v;
        }
        // stand-in for decoding the next MCU
        float f = (float)v.x;
        for (int i = 0; i < delay; i++) f = f * 1.0001f + 0.5f;
        v.w = (uint32_t)f;
    }
}

// reference pattern: every store instruction of a wave covers whole 128-byte lines (what idct_output_kernel does)
__global__ __launch_bounds__(256) void coalesced(uint8_t *out, size_t total16) {
    size_t i = (size_t)blockIdx.x * 256 * 8 + threadIdx.x;
    uint4 v = {threadIdx.x, blockIdx.x, 3, 4};
    for (int k = 0; k < 8; k++, i += 256)
        if (i < total16) reinterpret_cast<uint4 *>(out)[i] = v;
}

// the traffic mix of idct_output_kernel: every byte read once, every byte written once
__global__ __launch_bounds__(256) void copy16(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t total16) {
    size_t i = (size_t)blockIdx.x * 256 * 8 + threadIdx.x;
    uint4 v[8];
#pragma unroll
    for (int k = 0; k < 8; k++) v[k] = (i + k * 256 < total16) ? src[i + k * 256] : uint4{0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 8; k++)
        if (i + k * 256 < total16) dst[i + k * 256] = v[k];
}

int main(int argc, char **argv) {
    const int delay = argc > 1 ? atoi(argv[1]) : 2000;
    const int n_images = argc > 2 ? atoi(argv[2]) : 256;
    const size_t bytes = (size_t)n_images * W * H * 3;
    uint8_t *d;
    if (hipMalloc(&d, bytes) != hipSuccess) return 1;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int wg_per_img = (INTERVALS + 639) / 640;
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(scattered, dim3(n_images * wg_per_img), dim3(640), 0, 0, d, n_images, delay);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("scattered delay=%d: %.3f ms  %.1f GB/s\n", delay, ms, bytes / ms / 1e6);
    }
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        const size_t total16 = bytes / 16;
        hipLaunchKernelGGL(coalesced, dim3((total16 + 2047) / 2048), dim3(256), 0, 0, d, total16);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("coalesced: %.3f ms  %.1f GB/s\n", ms, bytes / ms / 1e6);
    }
    {
        uint8_t *d2;
        const size_t half = bytes / 2;
        if (hipMalloc(&d2, half) != hipSuccess) return 1;
        for (int rep = 0; rep < 3; rep++) {
            hipEventRecord(e0);
            const size_t total16 = half / 16;
            hipLaunchKernelGGL(copy16, dim3((total16 + 2047) / 2048), dim3(256), 0, 0, (const uint4 *)d, (uint4 *)d2, total16);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            printf("copy (read %zu MB + write %zu MB): %.3f ms  %.1f GB/s total\n", half >> 20, half >> 20, ms, 2.0 * half / ms / 1e6);
        }
    }
    return 0;
}
