// jpeglibrary_amd/csrc/k3_idct.hip -- K3: dequantise + float32 IDCT + level shift + block output in the requested layout; the literal Dispose() pass
//
// MUST be compiled with -ffp-contract=off: the reference's Vector4 arithmetic never fuses a*b+c
// (FastFloatingPointDCT.cs:79-185).  No fast-math.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>

#include "common.h"
#include "kernels.h"
#include "encode_kernels.h"
#include "kernels_device.h"

namespace jpgpu {

// ------------------------------------------------------------------------------------------------
// K3: dequantise + IDCT + level shift + block output.  One lane per block.
// ------------------------------------------------------------------------------------------------

// ref: JpegZigZag.cs:27-38
__device__ constexpr uint8_t kNat[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                         41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                         30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

// The 8-point butterfly of IDCT8x4_LeftPart/RightPart (ref: FastFloatingPointDCT.cs:79-127), one column.
// Operation order and parenthesisation are normative.
#define JPGPU_IDCT8(y0, y1, y2, y3, y4, y5, y6, y7)                     \
    {                                                                   \
        float mz0 = y1 + y7;                                            \
        float mz2 = y3 + y7;                                            \
        float mz1 = y3 + y5;                                            \
        float mz3 = y1 + y5;                                            \
        float mz4 = (mz0 + mz1) * 1.175875602f;                         \
        mz2 = (mz2 * -1.961570560f) + mz4;                              \
        mz3 = (mz3 * -0.390180644f) + mz4;                              \
        mz0 = mz0 * -0.899976223f;                                      \
        mz1 = mz1 * -2.562915447f;                                      \
        const float mb3 = ((y7 * 0.298631336f) + mz0) + mz2;            \
        const float mb2 = ((y5 * 2.053119869f) + mz1) + mz3;            \
        const float mb1 = ((y3 * 3.072711026f) + mz1) + mz2;            \
        const float mb0 = ((y1 * 1.501321110f) + mz0) + mz3;            \
        mz4 = (y2 + y6) * 0.541196100f;                                 \
        mz0 = y0 + y4;                                                  \
        mz1 = y0 - y4;                                                  \
        mz2 = mz4 + (y6 * -1.847759065f);                               \
        mz3 = mz4 + (y2 * 0.765366865f);                                \
        const float a0 = mz0 + mz3;                                     \
        const float a3 = mz0 - mz3;                                     \
        const float a1 = mz1 + mz2;                                     \
        const float a2 = mz1 - mz2;                                     \
        y0 = a0 + mb0;                                                  \
        y7 = a0 - mb0;                                                  \
        y1 = a1 + mb1;                                                  \
        y6 = a1 - mb1;                                                  \
        y2 = a2 + mb2;                                                  \
        y5 = a2 - mb2;                                                  \
        y3 = a3 + mb3;                                                  \
        y4 = a3 - mb3;                                                  \
    }

// DequantizeBlockAndUnZigZag for one block (ref: ScanDecoder/JpegScanDecoder.cs:50-62).
// c_lds: this lane's 64 int16 coefficients (zig-zag) in the swizzled LDS staging (8 chunks of 16 B, chunk p at
// c_lds + ((p ^ swz) * 16)); q_lds: 64 uint16 quantisers (zig-zag) of the block's component.
__device__ __forceinline__ void block_dequant(const uint8_t *c_lds, uint32_t swz, const uint16_t *q_lds, float (&f)[64]) {
#pragma unroll
    for (int piece = 0; piece < 8; piece++) {
        const uint4 cv = *reinterpret_cast<const uint4 *>(c_lds + ((piece ^ swz) * 16));
        const uint4 qv = reinterpret_cast<const uint4 *>(q_lds)[piece];
        const uint32_t cw[4] = {cv.x, cv.y, cv.z, cv.w};
        const uint32_t qw[4] = {qv.x, qv.y, qv.z, qv.w};
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int k = piece * 8 + j;
            const uint32_t w = cw[j >> 1], q = qw[j >> 1];
            const int32_t c = (j & 1) ? ((int32_t)w >> 16) : (int32_t)(int16_t)(w & 0xFFFF);
            const int32_t qq = (j & 1) ? (int32_t)(q >> 16) : (int32_t)(q & 0xFFFF);
            f[kNat[k]] = (float)(qq * c);  // ushort * short -> int -> float
        }
    }
}

typedef float float2v __attribute__((ext_vector_type(2)));
typedef short short2v __attribute__((ext_vector_type(2)));

// The 8-point butterfly of IDCT8x4_LeftPart/RightPart (ref: FastFloatingPointDCT.cs:79-127) on T = float or a pair of
// floats (two independent columns at once: v_pk_add_f32 / v_pk_mul_f32, each component one IEEE operation, never fused).
// Operation order and parenthesisation are normative.
template <typename T>
__device__ __forceinline__ void idct8(T &y0, T &y1, T &y2, T &y3, T &y4, T &y5, T &y6, T &y7) {
    T mz0 = y1 + y7;
    T mz2 = y3 + y7;
    T mz1 = y3 + y5;
    T mz3 = y1 + y5;
    T mz4 = (mz0 + mz1) * 1.175875602f;
    mz2 = (mz2 * -1.961570560f) + mz4;
    mz3 = (mz3 * -0.390180644f) + mz4;
    mz0 = mz0 * -0.899976223f;
    mz1 = mz1 * -2.562915447f;
    const T mb3 = ((y7 * 0.298631336f) + mz0) + mz2;
    const T mb2 = ((y5 * 2.053119869f) + mz1) + mz3;
    const T mb1 = ((y3 * 3.072711026f) + mz1) + mz2;
    const T mb0 = ((y1 * 1.501321110f) + mz0) + mz3;
    mz4 = (y2 + y6) * 0.541196100f;
    mz0 = y0 + y4;
    mz1 = y0 - y4;
    mz2 = mz4 + (y6 * -1.847759065f);
    mz3 = mz4 + (y2 * 0.765366865f);
    const T a0 = mz0 + mz3;
    const T a3 = mz0 - mz3;
    const T a1 = mz1 + mz2;
    const T a2 = mz1 - mz2;
    y0 = a0 + mb0;
    y7 = a0 - mb0;
    y1 = a1 + mb1;
    y6 = a1 - mb1;
    y2 = a2 + mb2;
    y5 = a2 - mb2;
    y3 = a3 + mb3;
    y4 = a3 - mb3;
}

// TransformIDCT + ShiftDataLevel (ref: FastFloatingPointDCT.cs:54-70, ScanDecoder/JpegScanDecoder.cs:64-73) on a
// dequantised block held in registers, two lanes of the butterfly per instruction.
// out[r * 4 + c2] = samples (r, 2*c2) | (r, 2*c2 + 1) << 16 as int16: (short)(Round(v) + levelShift), unclamped.
__device__ __forceinline__ void block_idct(const float (&f)[64], int32_t level_shift, uint32_t (&out)[32]) {
#if defined(JPGPU_K3_PRICE)
    // (pricing build only, tools/trace/ab_k3_price.sh: the transform left out -- wrong samples, the time of everything else)
#pragma unroll
    for (int i = 0; i < 32; i++) out[i] = (__builtin_bit_cast(uint32_t, f[2 * i]) >> 16) | (__builtin_bit_cast(uint32_t, f[2 * i + 1]) & 0xFFFF0000u) | (uint32_t)level_shift;
    return;
#endif
    // pass 1: 1-D IDCT along each ROW (the reference transposes, runs the column butterfly, transposes back);
    // a[r2][c] holds rows 2*r2 and 2*r2+1 of column c
    float2v a[4][8];
#pragma unroll
    for (int r2 = 0; r2 < 4; r2++)
#pragma unroll
        for (int c = 0; c < 8; c++) a[r2][c] = float2v{f[(2 * r2) * 8 + c], f[(2 * r2 + 1) * 8 + c]};
#pragma unroll
    for (int r2 = 0; r2 < 4; r2++) idct8(a[r2][0], a[r2][1], a[r2][2], a[r2][3], a[r2][4], a[r2][5], a[r2][6], a[r2][7]);
    // pass 2: along each COLUMN; b[r][c2] holds columns 2*c2 and 2*c2+1 of row r
    float2v b[8][4];
#pragma unroll
    for (int r2 = 0; r2 < 4; r2++)
#pragma unroll
        for (int c2 = 0; c2 < 4; c2++) {
            b[2 * r2][c2] = float2v{a[r2][2 * c2].x, a[r2][2 * c2 + 1].x};
            b[2 * r2 + 1][c2] = float2v{a[r2][2 * c2].y, a[r2][2 * c2 + 1].y};
        }
#pragma unroll
    for (int c2 = 0; c2 < 4; c2++) idct8(b[0][c2], b[1][c2], b[2][c2], b[3][c2], b[4][c2], b[5][c2], b[6][c2], b[7][c2]);
    const uint32_t shift2 = ((uint32_t)level_shift & 0xFFFFu) * 0x00010001u;
#pragma unroll
    for (int r = 0; r < 8; r++)
#pragma unroll
        for (int c2 = 0; c2 < 4; c2++) {
            const float2v v = b[r][c2] * 0.1250f;                  // MultiplyInplace(C_0_125)
            const int32_t x = (int32_t)__builtin_rintf(v.x);       // MathF.Round: half to even (v_rndne_f32)
            const int32_t y = (int32_t)__builtin_rintf(v.y);
            const uint32_t pk = ((uint32_t)x & 0xFFFFu) | ((uint32_t)y << 16);
            // (short)(Round + levelShift): 16-bit wrap-around add on both halves (v_pk_add_u16)
            const short2v sum = __builtin_bit_cast(short2v, pk) + __builtin_bit_cast(short2v, shift2);
            out[r * 4 + c2] = __builtin_bit_cast(uint32_t, sum);
        }
}

// signed clamp of two int16 samples to [0, 255] (JpegBufferOutputWriter8Bit.ClampTo8Bit): v_pk_max_i16 + v_pk_min_i16
__device__ __forceinline__ uint32_t clamp2_u8(uint32_t pk) {
    short2v v = __builtin_bit_cast(short2v, pk);
    v = __builtin_elementwise_max(v, short2v{0, 0});
    v = __builtin_elementwise_min(v, short2v{255, 255});
    return __builtin_bit_cast(uint32_t, v);
}
// four clamped samples (two packed pairs) -> four bytes
__device__ __forceinline__ uint32_t pack4_u8(uint32_t pk01, uint32_t pk23) {
    return __builtin_amdgcn_perm(clamp2_u8(pk23), clamp2_u8(pk01), 0x06040200u);  // bytes 0,2 of pk01 then 0,2 of pk23
}
// byte gather from the 8 bytes {lo (indices 0-3), hi (indices 4-7)}: one v_perm_b32
__device__ __forceinline__ uint32_t pick4(uint32_t lo, uint32_t hi, uint32_t sel) { return __builtin_amdgcn_perm(hi, lo, sel); }
#define JPGPU_SEL(a, b, c, d) ((uint32_t)(a) | ((uint32_t)(b) << 8) | ((uint32_t)(c) << 16) | ((uint32_t)(d) << 24))

constexpr int kIdctThreads = 256;
constexpr uint32_t kPxRowStride = kIdctThreads * 8;  // bytes between sample rows in the LDS pixel tile

// Output layout classes of the INTERLEAVED_U8 format (chosen per scan on the host, see idct_layout_class()).
enum IdctLayout : int { kLayGeneric = 0, kLayYccH1V1 = 1, kLayYccH2V1 = 2, kLayYccH2V2 = 3, kLayGray = 4, kNumIdctLayouts = 5 };

// ---- YCbCr -> RGB(A) (ref: apps/JpegDecode/JpegYCbCrToRgbConverter.cs:134-206).  The reference looks the terms up in
// tables built by Init (:66-118); with ReferenceBlackWhite = {0,255,128,255,128,255} the tables are exactly
//   yTable[i] = i, crRTable[i] = (cr_r * (i-128) + half) >> 16, cbBTable[i] = (cb_b * (i-128) + half) >> 16,
//   crGTable[i] = cr_g * (i-128), cbGTable[i] = cb_g * (i-128) + half,  and the clamp table is a clamp to [0, 255],
// so the terms are computed instead of fetched (the factors come from the host, derived like Init derives them).
struct ChromaTerms {
    int32_t r, g, b;
};
__device__ __forceinline__ ChromaTerms chroma_terms(uint32_t cb_sample, uint32_t cr_sample, const YccRgbFactors &k) {
    const int32_t cb = (int32_t)cb_sample - 128, cr = (int32_t)cr_sample - 128;
    ChromaTerms t;
    t.r = (k.cr_r * cr + 32768) >> 16;
    t.b = (k.cb_b * cb + 32768) >> 16;
    t.g = (k.cb_g * cb + 32768 + k.cr_g * cr) >> 16;
    return t;
}
__device__ __forceinline__ uint32_t clamp_u8_i32(int32_t v) { return (uint32_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }
// one pixel: R | G << 8 | B << 16
__device__ __forceinline__ uint32_t rgb_pixel(uint32_t y, const ChromaTerms &t) {
    return clamp_u8_i32((int32_t)y + t.r) | (clamp_u8_i32((int32_t)y + t.g) << 8) | (clamp_u8_i32((int32_t)y + t.b) << 16);
}
__device__ __forceinline__ uint32_t byte_of(uint32_t lo, uint32_t hi, int i) { return ((i < 4 ? lo : hi) >> (8 * (i & 3))) & 0xFFu; }
// N pixels (R | G << 8 | B << 16 each) -> interleaved bytes at dst (16-byte aligned for N = 16, 8-byte aligned for N = 8)
template <int N, int BPP>
__device__ __forceinline__ void store_rgb_pixels(uint8_t *dst, const uint32_t (&p)[N]) {
    if (BPP == 4) {
#pragma unroll
        for (int i = 0; i < N; i += 4) {
            const uint4 v = {p[i] | 0xFF000000u, p[i + 1] | 0xFF000000u, p[i + 2] | 0xFF000000u, p[i + 3] | 0xFF000000u};
            *reinterpret_cast<uint4 *>(dst + i * 4) = v;
        }
    } else {
        uint32_t w[N * 3 / 4];
#pragma unroll
        for (int i = 0; i < N; i += 4) {  // four pixels -> three dwords
            w[i * 3 / 4 + 0] = p[i] | (p[i + 1] << 24);
            w[i * 3 / 4 + 1] = (p[i + 1] >> 8) | (p[i + 2] << 16);
            w[i * 3 / 4 + 2] = (p[i + 2] >> 16) | (p[i + 3] << 8);
        }
        if (N == 16) {
#pragma unroll
            for (int i = 0; i < 3; i++) {
                const uint4 v = {w[i * 4], w[i * 4 + 1], w[i * 4 + 2], w[i * 4 + 3]};
                *reinterpret_cast<uint4 *>(dst + i * 16) = v;
            }
        } else {
#pragma unroll
            for (int i = 0; i < N * 3 / 8; i++) {
                const uint2 v = {w[i * 2], w[i * 2 + 1]};
                *reinterpret_cast<uint2 *>(dst + i * 8) = v;
            }
        }
    }
}

// Stand-alone conversion of an interleaved u8 image (C = 3: Y,Cb,Cr; C = 1: Y with Cb = Cr = 128 like
// apps/JpegDecode/DecodeAction.cs:57-65) for the layouts the writer kernel has no fused path for.
__global__ __launch_bounds__(256) void ycc_to_rgb_kernel(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, uint64_t n_pixels, int comps,
                                                         int bpp, YccRgbFactors k) {
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n_pixels; i += (uint64_t)gridDim.x * 256) {
        const uint32_t y = src[i * comps];
        const uint32_t cb = comps == 3 ? src[i * 3 + 1] : 128u, cr = comps == 3 ? src[i * 3 + 2] : 128u;
        const uint32_t px = rgb_pixel(y, chroma_terms(cb, cr, k));
        uint8_t *d = dst + i * bpp;
        d[0] = (uint8_t)px;
        d[1] = (uint8_t)(px >> 8);
        d[2] = (uint8_t)(px >> 16);
        if (bpp == 4) d[3] = 255;
    }
}

// "O3", the xunit tests' sink (ref: tests/JpegLibrary.Tests/Utils/JpegExtendingOutputWriter.cs:30-112) as a device format:
// out[(y * W + x) * 4 + c] (componentCount = 4, the way every test constructs it), uint16.  Input: the PLANAR_I16 planes K3
// wrote = WriteBlock's arguments before chroma expansion.  WriteBlockSlow replicates with shifts (:238-268), so pixel (x, y)
// of component c is plane_c[y >> vshift][x >> hshift] -- for a sampling factor that is the frame's maximum or 1.  Any other
// factor (round 6: H = 2 under a maximum of 4 ...) meets the decoder's `(offsetX + x) * 8` placement (...BaselineScanDecoder.cs:104,
// 134): block x of the MCU lands 8 pixels behind block x - 1 and is hs * 8 wide, so the blocks of one MCU overlap, the later call
// winning, and the MCU's last (hs - 1) * (h - 1) * 8 columns are never written (they keep the buffer's zero); the same downwards.
// The writer then takes (ushort)sample -- a negative sample becomes a large value -- clamps to 2^P - 1 and spreads the P bits
// over 16 (FastExpandBits for P >= 8, ExpandBits below).
// One launch for the whole batch: blockIdx.y = image (its descriptor in HBM), blockIdx.x strides over the image's pixels.
__global__ __launch_bounds__(256) void extend_u16_kernel(const uint8_t *__restrict__ planes, uint8_t *__restrict__ out_base,
                                                         const ExtendPlanes *__restrict__ images) {
    const ExtendPlanes g = images[blockIdx.y];
    uint16_t *out = reinterpret_cast<uint16_t *>(out_base + g.out_off);
    for (uint64_t px = (uint64_t)blockIdx.x * 256 + threadIdx.x; px < (uint64_t)g.width * g.height; px += (uint64_t)gridDim.x * 256) {
    const uint32_t y = (uint32_t)(px / g.width), x = (uint32_t)(px - (uint64_t)y * g.width);
    const uint32_t p = g.precision, mx = (1u << p) - 1u;
    uint16_t v4[4] = {0, 0, 0, 0};
#pragma unroll  // (compile-time component index: the descriptor's arrays stay in registers, no scratch)
    for (uint32_t c = 0; c < 4u; c++) {
        if (c >= g.ncomp) continue;
        const int16_t *pl = reinterpret_cast<const int16_t *>(planes + g.plane_off[c]);
        // the LAST block of the MCU whose replicated samples cover the pixel: x* = min(px >> 3, h - 1), covered while px < 8 x* + 8 hs
        const uint32_t mw = 8u * (g.max_h | (g.max_h == 0)), mh = 8u * (g.max_v | (g.max_v == 0));
        const uint32_t mcx = x / mw, pxm = x - mcx * mw, mcy = y / mh, pym = y - mcy * mh;
        const uint32_t bx = (pxm >> 3) < g.hcnt[c] - 1u ? (pxm >> 3) : g.hcnt[c] - 1u, by = (pym >> 3) < g.vcnt[c] - 1u ? (pym >> 3) : g.vcnt[c] - 1u;
        const uint32_t dx = pxm - 8u * bx, dy = pym - 8u * by;
        if (g.max_h != 0 && (dx >= (8u << g.hshift[c]) || dy >= (8u << g.vshift[c]))) continue;  // never written: the fresh buffer's zero
        const uint32_t s = g.max_h == 0 ? (uint32_t)(uint16_t)pl[(uint64_t)(y >> g.vshift[c]) * g.pitch[c] + (x >> g.hshift[c])]  // (a progressive frame: the allocator's Flush)
                                        : (uint32_t)(uint16_t)pl[(uint64_t)((mcy * g.vcnt[c] + by) * 8u + (dy >> g.vshift[c])) * g.pitch[c] + (mcx * g.hcnt[c] + bx) * 8u + (dx >> g.hshift[c])];
        uint32_t bits = s < mx ? s : mx;  // Clamp((ushort)sample, max)
        if (p >= 8u) {
            const uint32_t rem = 16u - p;
            bits = (bits << rem) | (bits & ((1u << rem) - 1u));  // FastExpandBits, as written
        } else {
            uint32_t cur = p;
            while (cur < 16u) {
                bits = (bits << p) | bits;
                cur += p;
            }
            if (cur > 16u) {
                bits >>= p;
                cur -= p;
                const uint32_t rem = 16u - cur;
                bits = (bits << rem) | (bits & ((1u << rem) - 1u));
            }
        }
        v4[c] = (uint16_t)bits;
    }
    // channels the frame does not have keep what the caller's (fresh, zeroed) buffer held: the batch owns the buffer, so zero
    *reinterpret_cast<uint2 *>(out + px * 4) = uint2{(uint32_t)v4[0] | ((uint32_t)v4[1] << 16), (uint32_t)v4[2] | ((uint32_t)v4[3] << 16)};
    }
}

// Output assembly of the INTERLEAVED_U8 format from the LDS sample tile [8 rows][256 blocks][8 B] (phase C).
// CONV: 0 = the samples as they are (Y,Cb,Cr), 3 / 4 = converted to R,G,B / R,G,B,A bytes (fast layouts only).
template <int LAY, int CONV>
__device__ __forceinline__ void interleaved_output_from_tile(const uint8_t *sh_px, const DevScan &s, uint32_t tile_first, uint32_t n_mcu,
                                                             uint32_t tid, bool have_block, const DevScanComponent &comp, uint32_t mcu_x,
                                                             uint32_t mcu_y, uint32_t b, uint8_t *out, const YccRgbFactors &kf, bool reached,
                                                             uint32_t mcu, uint32_t fail_block) {
    const uint32_t W = s.width, H = s.height, C = s.frame_components;
    uint8_t *img = out + s.out_off;

    if (LAY == kLayGeneric) {
        // any component count / sampling: bytewise stores with WriteBlockSlow's replication
        // (ref: ScanDecoder/JpegHuffmanBaselineScanDecoder.cs:238-268) and the sink's clipping (x < W, y < H)
        if (have_block) {
            const uint32_t hs = comp.hs, vs = comp.vs;
            // (a progressive frame's Dispose(): JpegBlockAllocator.Flush places block (col, row) of the component's own grid at
            // (col * hs * 8, row * vs * 8), JpegBlockAllocator.cs:120-149 -- the same place for a factor that is the maximum or 1)
            const bool flush = s.kind == kScanFrameOnly;
            const uint32_t x0 = flush ? (mcu_x * comp.h + s.blk_x[b]) * hs * 8 : (mcu_x * s.max_h + s.blk_x[b]) * 8;
            const uint32_t y0 = flush ? (mcu_y * comp.v + s.blk_y[b]) * vs * 8 : (mcu_y * s.max_v + s.blk_y[b]) * 8;
            const uint32_t hshift = 31 - __builtin_clz(hs | 1), vshift = 31 - __builtin_clz(vs | 1);
            // A sampling factor that is neither the frame's maximum nor 1 (round 6): the decoder places block x of the MCU at
            // (offsetX + x) * 8 (:104, 134) whatever the block's replicated width hs * 8 is, so the blocks of one MCU OVERLAP and the
            // later WriteBlock wins: a block owns a replicated pixel unless the next block that covers it -- block x + 1 of its
            // row from column 8 on, else from row 8 on the first covering block of the row below -- has reached the writer too.
            // (What no block covers keeps the buffer's content: the host clears the outputs of such frames, plan_work.)
            const uint32_t bx = s.blk_x[b], by = s.blk_y[b];
            const bool overlap_h = !flush && comp.h > 1 && hs > 1, overlap_v = !flush && comp.v > 1 && vs > 1;
            const uint32_t base = b - (by * comp.h + bx);  // the component's first block in the MCU
            if ((overlap_h || overlap_v) && !reached) return;
            for (uint32_t v = 0; v < vs; v++)
                for (uint32_t i = 0; i < 8; i++) {
                    const uint32_t y = y0 + 8 * v + i;
                    if (y >= H) continue;
                    const uint8_t *srow = sh_px + ((8 * v + i) >> vshift) * kPxRowStride + tid * 8;
                    for (uint32_t h = 0; h < hs; h++) {
                        if (overlap_h || overlap_v) {
                            uint32_t later = 0xFFFFFFFFu;  // the earliest later block that covers these eight pixels
                            if (overlap_h && bx + 1 < comp.h && h >= 1) later = base + by * comp.h + bx + 1;
                            else if (overlap_v && by + 1 < comp.v && v >= 1) {
                                const int32_t xf = (int32_t)bx + (int32_t)h - (int32_t)hs + 1;
                                later = base + (by + 1) * comp.h + (uint32_t)(xf < 0 ? 0 : xf);
                            }
                            if (later != 0xFFFFFFFFu && (uint64_t)mcu * s.blocks_per_mcu + later < fail_block) continue;
                        }
                        for (uint32_t j = 0; j < 8; j++) {
                            const uint32_t x = x0 + 8 * h + j;
                            if (x < W) img[((size_t)y * W + x) * C + comp.component_index] = srow[(8 * h + j) >> hshift];
                        }
                    }
                }
        }
    } else {
    // YCbCr fast paths: one task = one pixel row of one MCU (8*max_h pixels); consecutive lanes take consecutive MCUs of
    // the same row, so a wave writes one contiguous run of the output row per store instruction group.
    constexpr uint32_t max_h = (LAY == kLayYccH1V1) ? 1 : 2;
    constexpr uint32_t max_v = (LAY == kLayYccH2V2) ? 2 : 1;
    constexpr uint32_t rows_per_mcu = 8 * max_v;
    constexpr uint32_t vshift = max_v >> 1;
    constexpr uint32_t kbpm = max_h * max_v + 2;
    const uint32_t n_tasks = rows_per_mcu * n_mcu;
    for (uint32_t t = tid; t < n_tasks; t += kIdctThreads) {
        const uint32_t row = t / n_mcu, m = t - row * n_mcu;
        const uint32_t gm = tile_first + m;
        const uint32_t gx = gm % s.mcus_per_line, gy = gm / s.mcus_per_line;
        const uint32_t y = gy * rows_per_mcu + row;
        if (y >= H) continue;
        const uint8_t *yrow = sh_px + (row & 7) * kPxRowStride + (m * kbpm + (row >> 3) * max_h) * 8;
        const uint8_t *crow = sh_px + (row >> vshift) * kPxRowStride + (m * kbpm + max_h * max_v) * 8;
        if (max_h == 2 && CONV != 0) {
            const uint4 yv = *reinterpret_cast<const uint4 *>(yrow);  // 16 luma samples (two adjacent blocks)
            const uint4 cv = *reinterpret_cast<const uint4 *>(crow);  // 8 Cb (x,y) + 8 Cr (z,w)
            uint32_t px[16];
#pragma unroll
            for (int j = 0; j < 8; j++) {  // one chroma sample pair covers two pixels
                const ChromaTerms t = chroma_terms(byte_of(cv.x, cv.y, j), byte_of(cv.z, cv.w, j), kf);
                px[2 * j] = rgb_pixel(byte_of(j < 4 ? yv.x : yv.z, j < 4 ? yv.y : yv.w, (2 * j) & 7), t);
                px[2 * j + 1] = rgb_pixel(byte_of(j < 4 ? yv.x : yv.z, j < 4 ? yv.y : yv.w, (2 * j + 1) & 7), t);
            }
            store_rgb_pixels<16, (CONV == 4 ? 4 : 3)>(img + ((size_t)y * W + gx * 16) * (CONV == 4 ? 4 : 3), px);
        } else if (CONV != 0) {
            const uint2 yv = *reinterpret_cast<const uint2 *>(yrow);
            const uint2 bv = *reinterpret_cast<const uint2 *>(crow);
            const uint2 rv = *reinterpret_cast<const uint2 *>(crow + 8);
            uint32_t px[8];
#pragma unroll
            for (int j = 0; j < 8; j++) px[j] = rgb_pixel(byte_of(yv.x, yv.y, j), chroma_terms(byte_of(bv.x, bv.y, j), byte_of(rv.x, rv.y, j), kf));
            store_rgb_pixels<8, (CONV == 4 ? 4 : 3)>(img + ((size_t)y * W + gx * 8) * (CONV == 4 ? 4 : 3), px);
        } else if (max_h == 2) {
            const uint4 yv = *reinterpret_cast<const uint4 *>(yrow);  // 16 luma samples (two adjacent blocks)
            const uint4 cv = *reinterpret_cast<const uint4 *>(crow);  // 8 Cb (x,y) + 8 Cr (z,w)
            const uint32_t cc0 = pick4(cv.x, cv.z, JPGPU_SEL(0, 4, 1, 5)), cc1 = pick4(cv.x, cv.z, JPGPU_SEL(2, 6, 3, 7));
            const uint32_t cc2 = pick4(cv.y, cv.w, JPGPU_SEL(0, 4, 1, 5)), cc3 = pick4(cv.y, cv.w, JPGPU_SEL(2, 6, 3, 7));
            uint4 o0, o1, o2;
            o0.x = pick4(yv.x, cc0, JPGPU_SEL(0, 4, 5, 1));
            o0.y = pick4(yv.x, cc0, JPGPU_SEL(4, 5, 2, 6));
            o0.z = pick4(yv.x, cc0, JPGPU_SEL(7, 3, 6, 7));
            o0.w = pick4(yv.y, cc1, JPGPU_SEL(0, 4, 5, 1));
            o1.x = pick4(yv.y, cc1, JPGPU_SEL(4, 5, 2, 6));
            o1.y = pick4(yv.y, cc1, JPGPU_SEL(7, 3, 6, 7));
            o1.z = pick4(yv.z, cc2, JPGPU_SEL(0, 4, 5, 1));
            o1.w = pick4(yv.z, cc2, JPGPU_SEL(4, 5, 2, 6));
            o2.x = pick4(yv.z, cc2, JPGPU_SEL(7, 3, 6, 7));
            o2.y = pick4(yv.w, cc3, JPGPU_SEL(0, 4, 5, 1));
            o2.z = pick4(yv.w, cc3, JPGPU_SEL(4, 5, 2, 6));
            o2.w = pick4(yv.w, cc3, JPGPU_SEL(7, 3, 6, 7));
            uint4 *dst = reinterpret_cast<uint4 *>(img + ((size_t)y * W + gx * 16) * 3);
            dst[0] = o0;
            dst[1] = o1;
            dst[2] = o2;
        } else {
            const uint2 yv = *reinterpret_cast<const uint2 *>(yrow);
            const uint2 bv = *reinterpret_cast<const uint2 *>(crow);
            const uint2 rv = *reinterpret_cast<const uint2 *>(crow + 8);
            uint2 o0, o1, o2;
            {
                const uint32_t lo = pick4(bv.x, rv.x, JPGPU_SEL(0, 4, 1, 5)), hi = pick4(bv.x, rv.x, JPGPU_SEL(2, 6, 3, 7));
                const uint32_t mid = pick4(lo, hi, JPGPU_SEL(2, 3, 4, 5));
                o0.x = pick4(yv.x, lo, JPGPU_SEL(0, 4, 5, 1));
                o0.y = pick4(yv.x, mid, JPGPU_SEL(4, 5, 2, 6));
                o1.x = pick4(yv.x, hi, JPGPU_SEL(5, 3, 6, 7));
            }
            {
                const uint32_t lo = pick4(bv.y, rv.y, JPGPU_SEL(0, 4, 1, 5)), hi = pick4(bv.y, rv.y, JPGPU_SEL(2, 6, 3, 7));
                const uint32_t mid = pick4(lo, hi, JPGPU_SEL(2, 3, 4, 5));
                o1.y = pick4(yv.y, lo, JPGPU_SEL(0, 4, 5, 1));
                o2.x = pick4(yv.y, mid, JPGPU_SEL(4, 5, 2, 6));
                o2.y = pick4(yv.y, hi, JPGPU_SEL(5, 3, 6, 7));
            }
            uint2 *dst = reinterpret_cast<uint2 *>(img + ((size_t)y * W + gx * 8) * 3);
            dst[0] = o0;
            dst[1] = o1;
            dst[2] = o2;
        }
    }
    }  // YCbCr fast paths
}

// Each workgroup walks a run of consecutive tiles (kIdctThreads / blocks_per_mcu MCUs each) of one scan.
// Pipeline per tile:  lanes dequantise their block out of the LDS staging into registers -> barrier -> the staging is
// refilled for tile i+1 by LDS-DMA (global_load_lds_dwordx4: no VGPRs, asynchronous; the XOR swizzle is applied to the
// per-lane SOURCE address because the LDS side of the DMA is lane-linear) -> IDCT in registers while the DMA is in
// flight -> clamped samples to the LDS tile -> wait for the DMA -> barrier -> output assembly + global stores.
// All LDS lives in one array: staging 32 KiB | u8 sample tile [8 rows][256 blocks][8 B] 16 KiB | quant tables 512 B.
typedef __attribute__((address_space(3))) void jpgpu_lds_void;
typedef const __attribute__((address_space(1))) void jpgpu_gbl_void;

// PRE: the store already holds SAMPLES (the generic Dispose() pass of a progressive frame whose component slots do not map one
// to one onto its components, dispose_pass_kernel below): no dequantisation, no transform -- the block goes to the writer as it lies
template <int FMT, int LAY, bool PRE>
__device__ __forceinline__ void idct_output_body(
    const int16_t *__restrict__ coefs, const DevScan *__restrict__ scans, const IdctWork *__restrict__ work,
    const DevScanStatus *__restrict__ status, const DevQuantTable *__restrict__ quant_pool, uint8_t *__restrict__ out, YccRgbFactors kf) {
    constexpr int CONV = FMT == kFmtRgbU8 ? 3 : (FMT == kFmtRgbaU8 ? 4 : 0);  // fused YCbCr -> RGB(A), fast layouts and gray only
    __shared__ __attribute__((aligned(16))) uint8_t sh_all[kIdctThreads * 128 + kIdctThreads * 64 + kMaxScanComponents * 128];
    uint8_t *sh = sh_all;
    uint8_t *sh_px = sh_all + kIdctThreads * 128;
    uint16_t(*sh_q)[64] = reinterpret_cast<uint16_t(*)[64]>(sh_all + kIdctThreads * 128 + kIdctThreads * 64);

    const IdctWork wk = work[blockIdx.x];
    const DevScan &s = scans[wk.scan];
    const uint32_t tid = threadIdx.x;
    const uint32_t wave = tid >> 6;
    const uint32_t bpm = s.blocks_per_mcu;
    const uint32_t mcus_per_tile = wk.mcus_per_tile;
    // MCUs the scan never reached (EOI met in a restart check, :144-150): the reference leaves their samples as the caller's
    // buffer held them -- zero in the buffer the batch owns -- so they go through the same output code with zero samples
    uint32_t decoded = status ? status[wk.scan].decoded_mcus : s.total_mcus;
    if (decoded > s.total_mcus) decoded = s.total_mcus;
    // ... and the same BLOCK by block for a scan that fails (round 5): the reference has called WriteBlock for every block in
    // front of the one it threw in and for none behind it (:99-134, 153; JpegHuffmanScanDecoder.cs:103-110), and a scan behind a
    // failed scan of the image was never started (Decode() left with the exception).
    uint32_t fail_block = 0xFFFFFFFFu;
    if (status != nullptr && s.kind == kScanSequential) {
        const uint32_t fb = status[wk.scan].pad[1];
        if (fb != 0) fail_block = kFailBlockBase - fb;
        for (uint32_t j = s.first_scan; j < wk.scan; j++)
            if (scans[j].image_index == s.image_index && status[j].first_error != kNoError) fail_block = 0;
    }
    // keep: the caller's canvas (jpgpu_decode_scan), or a scan ordered behind another scan of its image that wrote the same
    // component -- what this scan does not reach is not touched.  The YCbCr fast layouts assemble whole pixels of whole MCUs:
    // there the MCU the scan failed in is left to a launch of the bytewise form (first_mcu == kIdctPartialMcu).
    const bool keep = (s.shadow_mask & kKeepUnreachedMcus) != 0;
    constexpr bool kPerMcuLayout = fmt_is_interleaved(FMT) && (LAY == kLayYccH1V1 || LAY == kLayYccH2V1 || LAY == kLayYccH2V2);
    uint32_t first_mcu = wk.first_mcu;
    uint32_t range_end = wk.first_mcu + wk.n_mcus;
    if (first_mcu == kIdctPartialMcu) {
        if (fail_block == 0xFFFFFFFFu || fail_block % bpm == 0 || fail_block / bpm >= decoded) return;
        first_mcu = fail_block / bpm;
        range_end = first_mcu + 1;
    } else if (keep) {
        const uint32_t reach = kPerMcuLayout ? fail_block / bpm : (fail_block == 0xFFFFFFFFu ? fail_block : (fail_block + bpm - 1) / bpm);
        if (range_end > decoded) range_end = decoded;
        if (range_end > reach) range_end = reach;
        if (first_mcu >= range_end) return;
    }

    // quantisation tables of the scan components
    if (tid < (uint32_t)s.scan_components * 32) {
        const uint32_t c = tid >> 5, i = tid & 31;
        reinterpret_cast<uint32_t *>(sh_q[c])[i] =
            reinterpret_cast<const uint32_t *>(quant_pool[s.quant_pool[s.comp[c].quant_slot]].q)[i];
    }

    const uint32_t mcu_local = tid / bpm;
    const uint32_t b = tid - mcu_local * bpm;
    const uint32_t ci_early = s.blk_comp[b < kMaxBlocksPerMcu ? b : 0];
    const uint8_t *coef_bytes = reinterpret_cast<const uint8_t *>(coefs + s.coef_off * 64);

    auto tile_mcus = [&](uint32_t first) { return (range_end - first) < mcus_per_tile ? (range_end - first) : mcus_per_tile; };
    // LDS-DMA of one tile: linear 16-byte slot c = k * 256 + tid (block c >> 3, slot c & 7) receives piece
    // (slot ^ swizzle(block)); the swizzle term ((block >> 1) & 7) does not depend on k, so every lane's source is
    // one fixed offset plus k * 4096.  Always a full tile: the coefficient buffer has a tile of slack behind it.
    const uint32_t tile_blocks = mcus_per_tile * bpm;
    auto dma_tile = [&](uint32_t tile_first) {
        // (recomputed per tile, three instructions, rather than kept in a register across the transform)
        uint32_t t_ = tid;
        asm volatile("" : "+v"(t_));
        const uint32_t dma_lane_off = (t_ >> 3) * 128 + (((t_ & 7) ^ ((t_ >> 4) & 7)) * 16);
        const uint8_t *src = coef_bytes + (uint64_t)tile_first * bpm * 128;  // wave-uniform
#pragma unroll
        for (int k = 0; k < 8; k++)
            if (k < 6 || (uint32_t)k * 32 + (tid >> 3) < tile_blocks)  // blocks behind the tile's last MCU are not fetched
                __builtin_amdgcn_global_load_lds((jpgpu_gbl_void *)(src + (uint32_t)(k * 4096) + dma_lane_off),
                                                 (jpgpu_lds_void *)(sh + ((uint32_t)k * kIdctThreads + wave * 64) * 16), 16, 0, 0);
    };

    dma_tile(first_mcu);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

  for (uint32_t tile_first = first_mcu; tile_first < range_end; tile_first += mcus_per_tile) {
    const uint32_t n_mcu = tile_mcus(tile_first);
    const uint32_t n_blk = n_mcu * bpm;
    const uint32_t next_first = tile_first + mcus_per_tile;
    const bool have_next = next_first < range_end;

    const uint32_t mcu = tile_first + mcu_local;
    const bool have_block = tid < n_blk;
    // a scan component whose frame component a LATER scan component also resolves to: the reference writes its blocks first
    // and the later component's over them (WriteBlock by ComponentIndex, :118-134), so only the later ones reach the output
    bool writes = have_block && ((s.shadow_mask >> ci_early) & 1u) == 0;

    // phase B1: dequantise this lane's block out of the staging into registers
    float f[64];
    uint32_t px[32];  // int16 sample pairs
    if (PRE) {
        if (have_block) {
#pragma unroll
            for (int piece = 0; piece < 8; piece++) {
                const uint4 cv = *reinterpret_cast<const uint4 *>(sh + tid * 128 + ((piece ^ ((tid >> 1) & 7)) * 16));
                px[piece * 4] = cv.x;
                px[piece * 4 + 1] = cv.y;
                px[piece * 4 + 2] = cv.z;
                px[piece * 4 + 3] = cv.w;
            }
        }
    } else {
        // The lane's eight swizzled staging addresses do not change from tile to tile; hipcc computes them in front of the tile
        // loop -- and in the three variants with the most state in their output assembly spills five of them to scratch for
        // the length of the transform.  There they are derived again in every tile (sixteen instructions) from a copy of the
        // lane id the compiler cannot see through.
        constexpr bool kPerTile = (FMT == kFmtRgbU8 && (LAY == kLayYccH2V1 || LAY == kLayYccH2V2)) || (FMT == kFmtInterleavedU8 && LAY == kLayGeneric);
        uint32_t t_ = tid;
        if (kPerTile) asm volatile("" : "+v"(t_));
        if (have_block) block_dequant(sh + t_ * 128, (t_ >> 1) & 7, sh_q[ci_early], f);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // every lane holds its coefficients: the staging can be refilled
    if (have_next) dma_tile(next_first);  // in flight during the whole transform below

    // phase B2: IDCT entirely in registers
    if (!PRE && have_block) block_idct(f, (int32_t)s.level_shift, px);
    // (a frame has fewer than 2^32 blocks: 32-bit arithmetic; computed behind the transform, nothing more alive across it)
    const bool reached = mcu < decoded && mcu * bpm + b < fail_block;
    if ((s.shadow_mask & 0xFu) == 0) {
        writes = writes && (reached || !keep);
    } else {
        // A scan that names a frame component twice (a corrupted selector; uniform branch): the reference writes the blocks in scan
        // order, so at every place the LAST block that reached the writer stands -- the later duplicate's wherever the scan got to
        // it, the earlier one's in the MCU the scan failed in BETWEEN the two (:118-134), zeros (in the batch's own buffer) where
        // neither got.  Exactly one lane writes each place: a reached block unless the later duplicate's block at its place is
        // reached too; an unreached block only if it is the last duplicate and no earlier duplicate's block is there either.
        const uint32_t bb = b < kMaxBlocksPerMcu ? b : 0u, me = s.blk_comp[bb], fc = s.comp[me & 3u].component_index;
        uint32_t later = 0xFFu, earlier = 0xFFu;  // scan components: the nearest duplicates of this block's
        for (uint32_t cc = 0; cc < s.scan_components && cc < (uint32_t)kMaxScanComponents; cc++) {
            if (cc == me || s.comp[cc].component_index != fc) continue;
            if (cc > me && later == 0xFFu) later = cc;
            if (cc < me) earlier = cc;
        }
        uint32_t first_me = 0xFFu, first_later = 0xFFu, first_earlier = 0xFFu;  // their first blocks in the MCU
        for (uint32_t k = 0; k < bpm && k < (uint32_t)kMaxBlocksPerMcu; k++) {
            const uint32_t ck = s.blk_comp[k];
            if (ck == me && first_me == 0xFFu) first_me = k;
            if (ck == later && first_later == 0xFFu) first_later = k;
            if (ck == earlier && first_earlier == 0xFFu) first_earlier = k;
        }
        const bool later_reached = later != 0xFFu && mcu < decoded && mcu * bpm + (b - first_me + first_later) < fail_block;
        const bool earlier_reached = earlier != 0xFFu && mcu < decoded && mcu * bpm + (b - first_me + first_earlier) < fail_block;
        writes = have_block && (reached ? !later_reached : (!keep && later == 0xFFu && !earlier_reached));
    }
    if (!reached) {
#pragma unroll
        for (int i = 0; i < 32; i++) px[i] = 0;
    }
    bool synced = false;
    // the MCU's place in the image is only needed from here on: computed behind the transform (an empty asm the compiler may not
    // move across keeps it from being hoisted in front of it), two registers fewer are alive while the 64 + 32 of the
    // transform are -- what four of the sixteen variants spilled (profiles/r03_kernel_resources.txt)
    uint32_t mcu_late = mcu, b_late = b;
    asm volatile("" : "+v"(mcu_late), "+v"(b_late));
    const uint32_t mcu_y = mcu_late / s.mcus_per_line, mcu_x = mcu_late - mcu_y * s.mcus_per_line;
    const uint32_t ci = s.blk_comp[b_late < kMaxBlocksPerMcu ? b_late : 0];  // (again: one byte from the L1-resident descriptor)
    const DevScanComponent comp = s.comp[ci];

    if (FMT == kFmtPlanarI16) {
        // "O1": unclamped int16 at component-native resolution, planes padded to whole MCUs
        if (writes) {
            int16_t *plane = reinterpret_cast<int16_t *>(out + s.out_off + s.plane_off[ci]);
            const uint32_t pitch = s.plane_pitch[ci];
            const uint32_t x0 = (mcu_x * comp.h + s.blk_x[b]) * 8, y0 = (mcu_y * comp.v + s.blk_y[b]) * 8;
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const uint4 v = {px[r * 4 + 0], px[r * 4 + 1], px[r * 4 + 2], px[r * 4 + 3]};
                *reinterpret_cast<uint4 *>(plane + (size_t)(y0 + r) * pitch + x0) = v;
            }
        }
    } else {
    // u8 formats: clamp (signed, like JpegBufferOutputWriter8Bit.ClampTo8Bit) and pack 8 samples per row
    uint2 rows[8];
#pragma unroll
    for (int r = 0; r < 8; r++) {
        rows[r].x = pack4_u8(px[r * 4 + 0], px[r * 4 + 1]);
        rows[r].y = pack4_u8(px[r * 4 + 2], px[r * 4 + 3]);
    }

    if (CONV != 0 && LAY == kLayGray) {
        // a single-component image as R = G = B = Y (Cb = Cr = 128 contribute nothing, DecodeAction.cs:57-65)
        if (writes) {
            const uint32_t x0 = (mcu_x * comp.h + s.blk_x[b]) * 8, y0 = (mcu_y * comp.v + s.blk_y[b]) * 8;
#pragma unroll
            for (int r = 0; r < 8; r++) {
                if (y0 + r >= s.height) continue;
                uint32_t px[8];
#pragma unroll
                for (int j = 0; j < 8; j++) px[j] = byte_of(rows[r].x, rows[r].y, j) * 0x010101u;
                store_rgb_pixels<8, (CONV == 4 ? 4 : 3)>(out + s.out_off + ((size_t)(y0 + r) * s.width + x0) * (CONV == 4 ? 4 : 3), px);
            }
        }
    } else if (FMT == kFmtPlanarU8 || (FMT == kFmtInterleavedU8 && LAY == kLayGray)) {
        // planar u8 (planes padded to whole MCUs), or a single-component interleaved image (same addressing,
        // pitch = W, clipped at the bottom; the host only picks kLayGray when W is a multiple of 8)
        if (writes) {
            const bool gray = (FMT == kFmtInterleavedU8);
            uint8_t *plane = out + s.out_off + (gray ? 0 : s.plane_off[ci]);
            const uint32_t pitch = gray ? s.width : s.plane_pitch[ci];
            const uint32_t x0 = (mcu_x * comp.h + s.blk_x[b]) * 8, y0 = (mcu_y * comp.v + s.blk_y[b]) * 8;
#pragma unroll
            for (int r = 0; r < 8; r++)
                if (!gray || y0 + r < s.height) *reinterpret_cast<uint2 *>(plane + (size_t)(y0 + r) * pitch + x0) = rows[r];
        }
    } else {
    // ---- interleaved u8 ("O2", JpegBufferOutputWriter8Bit semantics): stage the clamped samples in LDS, tile[r][block]
    if (have_block) {
#pragma unroll
        for (int r = 0; r < 8; r++) *reinterpret_cast<uint2 *>(sh_px + r * kPxRowStride + tid * 8) = rows[r];
    }
    // The DMA of the next tile has had the whole transform to land.  Wait for it BEFORE this tile's global stores are
    // issued (vmcnt retires in order: waiting later would also wait for those stores to drain), then one barrier
    // publishes both the sample tile and the refilled staging.
    asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    synced = true;

    interleaved_output_from_tile<LAY, CONV>(sh_px, s, tile_first, n_mcu, tid, writes, comp, mcu_x, mcu_y, b, out, kf, reached, mcu, fail_block);
    }  // interleaved
    }  // u8 formats

    if (!synced) {  // planar / gray paths: publish the refilled staging
        asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
  }  // tile loop
}


hipError_t launch_extend_u16(hipStream_t stream, const uint8_t *planes, uint8_t *out_base, const ExtendPlanes *images, int n_images,
                             uint32_t max_pixels) {
    if (n_images <= 0 || max_pixels == 0) return hipSuccess;
    const uint32_t bx = (uint32_t)std::min<uint64_t>(((uint64_t)max_pixels + 255) / 256, 4096);
    for (int base = 0; base < n_images; base += 65535) {  // grid.y limit
        const int n = n_images - base < 65535 ? n_images - base : 65535;
        hipLaunchKernelGGL(extend_u16_kernel, dim3(bx, (uint32_t)n), dim3(256), 0, stream, planes, out_base, images + base);
    }
    return hipGetLastError();
}


template <int FMT, int LAY>
__global__ __launch_bounds__(kIdctThreads, (FMT == kFmtPlanarI16 ? 2 : 3)) void idct_output_kernel(
    const int16_t *__restrict__ coefs, const DevScan *__restrict__ scans, const IdctWork *__restrict__ work,
    const DevScanStatus *__restrict__ status, const DevQuantTable *__restrict__ quant_pool, uint8_t *__restrict__ out, YccRgbFactors kf) {
    idct_output_body<FMT, LAY, false>(coefs, scans, work, status, quant_pool, out, kf);
}
template <int FMT>
__global__ __launch_bounds__(kIdctThreads, 2) void flush_output_kernel(
    const int16_t *__restrict__ coefs, const DevScan *__restrict__ scans, const IdctWork *__restrict__ work,
    const DevScanStatus *__restrict__ status, const DevQuantTable *__restrict__ quant_pool, uint8_t *__restrict__ out, YccRgbFactors kf) {
    idct_output_body<FMT, kLayGeneric, true>(coefs, scans, work, status, quant_pool, out, kf);
}

// The reference's Dispose() as it is written (ScanDecoder/JpegHuffmanProgressiveScanDecoder.cs:421-470): every component SLOT of
// the scan decoder, as the last scans left it, dequantises + transforms + level-shifts the blocks of its component IN PLACE.  For
// files in the usual scan order that is one transform per component and K3 does it on the way to the writer.  When the slots
// do not map one to one onto the components (a file whose last scan of slot 0 is not the first component: e.g. slots
// {Cb, Cb, Cr}) a component is transformed twice -- the second time reading its own samples as zig-zag coefficients -- and
// another never (its quantised coefficients reach the writer as samples); a file without any scan flushes zeros.  This kernel
// does literally that to the frame's store, one lane per block, `n` transforms with the slots' tables in slot order;
// flush_output_kernel then writes the store out.  (Both also serve the partial flush of a progressive file that failed.)
__global__ __launch_bounds__(64) void dispose_pass_kernel(int16_t *__restrict__ coefs, const DisposeJob *__restrict__ jobs,
                                                          const DevQuantTable *__restrict__ quant_pool) {
    const DisposeJob &j = jobs[blockIdx.y];
    const uint32_t g = blockIdx.x * 64u + threadIdx.x;
    if (g >= j.n_blocks) return;
    const uint32_t c = j.blk_comp[g % j.bpm];
    const uint32_t n = j.n[c];
    if (n == 0) return;
    int16_t *blk = coefs + (j.coef_off + g) * 64;
    uint32_t w[32];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const uint4 v = reinterpret_cast<const uint4 *>(blk)[i];
        w[i * 4] = v.x;
        w[i * 4 + 1] = v.y;
        w[i * 4 + 2] = v.z;
        w[i * 4 + 3] = v.w;
    }
    for (uint32_t t = 0; t < n; t++) {
        const uint16_t *q = quant_pool[j.quant[c][t]].q;
        float f[64];
#pragma unroll
        for (int k = 0; k < 64; k++) {
            const int32_t cv = (k & 1) ? ((int32_t)w[k >> 1] >> 16) : (int32_t)(int16_t)(w[k >> 1] & 0xFFFFu);
            f[kNat[k]] = (float)((int32_t)q[k] * cv);  // ushort * short -> int -> float (DequantizeBlockAndUnZigZag)
        }
        block_idct(f, (int32_t)j.level_shift, w);
    }
#pragma unroll
    for (int i = 0; i < 8; i++) reinterpret_cast<uint4 *>(blk)[i] = make_uint4(w[i * 4], w[i * 4 + 1], w[i * 4 + 2], w[i * 4 + 3]);
}

template <int FMT, int LAY>
static void launch_idct_one(hipStream_t stream, const int16_t *coefs, const DevScan *scans, const IdctWork *work, int n_work,
                            const DevScanStatus *status, const DevQuantTable *quant_pool, uint8_t *out,
                            YccRgbFactors kf = YccRgbFactors{0, 0, 0, 0}) {
    hipLaunchKernelGGL((idct_output_kernel<FMT, LAY>), dim3(n_work), dim3(kIdctThreads), 0, stream, coefs, scans, work, status,
                       quant_pool, out, kf);
}

// work is sorted by layout class; class_begin[c]..class_begin[c+1] are the workgroups of class c.
hipError_t launch_idct(hipStream_t stream, const int16_t *coefs, const DevScan *scans, const IdctWork *work,
                       const int class_begin[kNumIdctLayoutClasses + 1], const DevScanStatus *status,
                       const DevQuantTable *quant_pool, uint8_t *out, int format, const YccRgbFactors &kf, uint8_t *generic_out) {
    for (int c = 0; c < kNumIdctLayoutClasses; c++) {
        const int n = class_begin[c + 1] - class_begin[c];
        if (n <= 0) continue;
        const IdctWork *w = work + class_begin[c];
        if (c == kIdctClassStoreHoldsSamples) {  // the generic Dispose() pass has run: the store goes to the writer as it is
            if (format == kFmtPlanarI16) hipLaunchKernelGGL((flush_output_kernel<kFmtPlanarI16>), dim3(n), dim3(kIdctThreads), 0, stream, coefs, scans, w, status, quant_pool, out, kf);
            else if (format == kFmtPlanarU8) hipLaunchKernelGGL((flush_output_kernel<kFmtPlanarU8>), dim3(n), dim3(kIdctThreads), 0, stream, coefs, scans, w, status, quant_pool, out, kf);
            else hipLaunchKernelGGL((flush_output_kernel<kFmtInterleavedU8>), dim3(n), dim3(kIdctThreads), 0, stream, coefs, scans, w, status, quant_pool,
                                    (format == kFmtRgbU8 || format == kFmtRgbaU8) ? generic_out : out, kf);
        } else if (format == kFmtPlanarI16) {
            launch_idct_one<kFmtPlanarI16, kLayGeneric>(stream, coefs, scans, w, n, status, quant_pool, out);
        } else if (format == kFmtPlanarU8) {
            launch_idct_one<kFmtPlanarU8, kLayGeneric>(stream, coefs, scans, w, n, status, quant_pool, out);
        } else if (format == kFmtRgbU8) {
            switch (c) {
            case kLayYccH1V1: launch_idct_one<kFmtRgbU8, kLayYccH1V1>(stream, coefs, scans, w, n, status, quant_pool, out, kf); break;
            case kLayYccH2V1: launch_idct_one<kFmtRgbU8, kLayYccH2V1>(stream, coefs, scans, w, n, status, quant_pool, out, kf); break;
            case kLayYccH2V2: launch_idct_one<kFmtRgbU8, kLayYccH2V2>(stream, coefs, scans, w, n, status, quant_pool, out, kf); break;
            case kLayGray: launch_idct_one<kFmtRgbU8, kLayGray>(stream, coefs, scans, w, n, status, quant_pool, out, kf); break;
            // no fused path: the samples go to `generic_out` as INTERLEAVED_U8 and are converted by launch_ycc_to_rgb
            default: launch_idct_one<kFmtInterleavedU8, kLayGeneric>(stream, coefs, scans, w, n, status, quant_pool, generic_out); break;
            }
        } else if (format == kFmtRgbaU8) {
            switch (c) {
            case kLayYccH1V1: launch_idct_one<kFmtRgbaU8, kLayYccH1V1>(stream, coefs, scans, w, n, status, quant_pool, out, kf); break;
            case kLayYccH2V1: launch_idct_one<kFmtRgbaU8, kLayYccH2V1>(stream, coefs, scans, w, n, status, quant_pool, out, kf); break;
            case kLayYccH2V2: launch_idct_one<kFmtRgbaU8, kLayYccH2V2>(stream, coefs, scans, w, n, status, quant_pool, out, kf); break;
            case kLayGray: launch_idct_one<kFmtRgbaU8, kLayGray>(stream, coefs, scans, w, n, status, quant_pool, out, kf); break;
            default: launch_idct_one<kFmtInterleavedU8, kLayGeneric>(stream, coefs, scans, w, n, status, quant_pool, generic_out); break;
            }
        } else {
            switch (c) {
            case kLayYccH1V1: launch_idct_one<kFmtInterleavedU8, kLayYccH1V1>(stream, coefs, scans, w, n, status, quant_pool, out); break;
            case kLayYccH2V1: launch_idct_one<kFmtInterleavedU8, kLayYccH2V1>(stream, coefs, scans, w, n, status, quant_pool, out); break;
            case kLayYccH2V2: launch_idct_one<kFmtInterleavedU8, kLayYccH2V2>(stream, coefs, scans, w, n, status, quant_pool, out); break;
            case kLayGray: launch_idct_one<kFmtInterleavedU8, kLayGray>(stream, coefs, scans, w, n, status, quant_pool, out); break;
            default: launch_idct_one<kFmtInterleavedU8, kLayGeneric>(stream, coefs, scans, w, n, status, quant_pool, out); break;
            }
        }
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

// INTERLEAVED_U8 image (comps = 1 or 3) -> RGB / RGBA, for the layouts without a fused path
hipError_t launch_ycc_to_rgb(hipStream_t stream, const uint8_t *src, uint8_t *dst, uint64_t n_pixels, int comps, int bpp, const YccRgbFactors &kf) {
    if (n_pixels == 0) return hipSuccess;
    const uint64_t want = (n_pixels + 255) / 256;
    const int grid = (int)(want < 65536 ? want : 65536);
    hipLaunchKernelGGL(ycc_to_rgb_kernel, dim3(grid), dim3(256), 0, stream, src, dst, n_pixels, comps, bpp, kf);
    return hipGetLastError();
}


hipError_t launch_dispose_pass(hipStream_t stream, int16_t *coefs, const DisposeJob *jobs, int n_jobs, uint32_t max_blocks, const DevQuantTable *quant_pool) {
    if (n_jobs <= 0 || max_blocks == 0) return hipSuccess;
    hipLaunchKernelGGL(dispose_pass_kernel, dim3((max_blocks + 63u) / 64u, n_jobs), dim3(64), 0, stream, coefs, jobs, quant_pool);
    return hipGetLastError();
}

// Layout class of a scan for the INTERLEAVED_U8 format (0 = generic bytewise path).
int idct_layout_class(const DevScan &s) {
    const uint32_t W = s.width;
    if (s.frame_components == 1 && s.scan_components == 1 && s.comp[0].h == 1 && s.comp[0].v == 1 && (W % 8) == 0 && (s.out_off % 8) == 0)
        return kLayGray;
    const bool ycc = s.frame_components == 3 && s.scan_components == 3 && s.comp[0].component_index == 0 &&
                     s.comp[1].component_index == 1 && s.comp[2].component_index == 2 && s.comp[0].hs == 1 && s.comp[0].vs == 1 &&
                     s.comp[1].h == 1 && s.comp[1].v == 1 && s.comp[2].h == 1 && s.comp[2].v == 1;
    if (!ycc) return kLayGeneric;
    if (s.max_h == 1 && s.max_v == 1 && (W % 8) == 0 && (s.out_off % 8) == 0) return kLayYccH1V1;
    if (s.max_h == 2 && (W % 16) == 0 && (s.out_off % 16) == 0) {
        if (s.max_v == 1) return kLayYccH2V1;
        if (s.max_v == 2) return kLayYccH2V2;
    }
    return kLayGeneric;
}

}  // namespace jpgpu
