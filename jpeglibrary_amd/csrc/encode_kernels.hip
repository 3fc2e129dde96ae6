// jpeglibrary_amd/csrc/encode_kernels.hip -- the baseline ENCODER's hot path on gfx950.
//
// Reference (paths relative to /root/reference/src/JpegLibrary): JpegEncoder.WriteScanData (JpegEncoder.cs:662-741) is a
// serial loop per block: ReadBlock (+ 2x2 box sub-sampling) -> ShiftDataLevel -> FastFloatingPointDCT.TransformFDCT ->
// ZigZagAndQuantizeBlock -> EncodeBlock (Huffman emit through JpegWriter.WriteBits, FF byte stuffing on flush).
// Here it becomes four device stages over a batch of images:
//   E1 fdct_quant_kernel    one lane per MCU: gather / sub-sample the samples, float32 FDCT in the reference's operation
//                           order (no FMA), quantise -> int16 zig-zag blocks in MCU order (the decoder's layout)
//   E2 block_bits_kernel    one lane per block: bits its Huffman codes + magnitudes will take; prefix sums give every
//                           block its bit offset in the scan (encoding is embarrassingly parallel once lengths are known)
//   E3 emit_kernel          one lane per block: the codes again, ORed into the raw bit stream at the block's offset
//   E4 stuff_*_kernel       FF -> FF 00 (JpegWriter.FlushRegister), the final one-bits padding, EOI behind the data
// MUST be compiled with -ffp-contract=off (the reference's Vector4 arithmetic never fuses a*b+c).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <stdint.h>

#include "encode_kernels.h"

namespace jpgpu {

// ref: JpegZigZag.cs:27-38 (zig-zag index -> natural index)
__device__ constexpr uint8_t kEncNat[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                            41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                            30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

// ------------------------------------------------------------------------------------------------ E1

// FDCT8x4_LeftPart / RightPart for one column of eight values (ref: FastFloatingPointDCT.cs:194-311).
// Operation order and parenthesisation are normative.
// T = float, or two floats side by side (EncF2: the same row or column of two blocks, one v_pk_* instruction for both --
// every component one IEEE operation, never fused).
typedef float EncF2 __attribute__((ext_vector_type(2)));
template <typename T>
__device__ __forceinline__ void fdct8(T &s0, T &s1, T &s2, T &s3, T &s4, T &s5, T &s6, T &s7) {
    T c0 = s0, c1 = s7;
    const T t0 = c0 + c1, t7 = c0 - c1;
    c1 = s6;
    c0 = s1;
    const T t1 = c0 + c1, t6 = c0 - c1;
    c1 = s5;
    c0 = s2;
    const T t2 = c0 + c1, t5 = c0 - c1;
    c0 = s3;
    c1 = s4;
    const T t3 = c0 + c1, t4 = c0 - c1;
    c0 = t0 + t3;
    T c3 = t0 - t3;
    c1 = t1 + t2;
    T c2 = t1 - t2;
    s0 = c0 + c1;
    s4 = c0 - c1;
    float w0 = 0.541196f, w1 = 1.306563f;
    s2 = (w0 * c2) + (w1 * c3);
    s6 = (w0 * c3) - (w1 * c2);
    w0 = 1.175876f;
    w1 = 0.785695f;
    c3 = (w0 * t4) + (w1 * t7);
    c0 = (w0 * t7) - (w1 * t4);
    w0 = 1.387040f;
    w1 = 0.275899f;
    c2 = (w0 * t5) + (w1 * t6);
    c1 = (w0 * t6) - (w1 * t5);
    s3 = c0 - c2;
    s5 = c3 - c1;
    const float invsqrt2 = 0.707107f;
    c0 = (c0 + c2) * invsqrt2;
    c3 = (c3 + c1) * invsqrt2;
    s1 = c0 + c3;
    s7 = c0 - c3;
}

// TransformFDCT (ref: :343-362): the reference transposes, runs the column butterfly, transposes, runs it again; in
// element terms pass 1 transforms every ROW of the block (over x), pass 2 every COLUMN of the result, then x 0.125.
__device__ __forceinline__ void block_fdct(float (&f)[64]) {
#pragma unroll
    for (int r = 0; r < 8; r++) fdct8(f[r * 8 + 0], f[r * 8 + 1], f[r * 8 + 2], f[r * 8 + 3], f[r * 8 + 4], f[r * 8 + 5], f[r * 8 + 6], f[r * 8 + 7]);
#pragma unroll
    for (int c = 0; c < 8; c++) fdct8(f[0 * 8 + c], f[1 * 8 + c], f[2 * 8 + c], f[3 * 8 + c], f[4 * 8 + c], f[5 * 8 + c], f[6 * 8 + c], f[7 * 8 + c]);
#pragma unroll
    for (int i = 0; i < 64; i++) f[i] = f[i] * 0.1250f;
}

// The shapes fdct_fused_kernel takes (E1 as one kernel: three components from three-sample pixels, luma 2 x 2, 2 x 1 or
// 1 x 1 -> instance 1, 2, 3; from R,G,B,A pixels -> 4, 5, 6); every other shape (0) goes through E1a + E1b.
__host__ __device__ inline int enc_fused_shape(const DevEncImage &im) {
    // (four-byte pixels: R, G, B, A with the alpha ignored -- ConvertRgba32ToYCbCr8, instances 4, 5, 6)
    if (im.components != 3 || !(im.in_components == 3 || (im.in_components == 4 && im.input_rgb != 0))) return 0;
    const int wide = im.in_components == 4 ? 3 : 0;
    if (im.luma_h == 2 && im.luma_v == 2) return 1 + wide;
    if (im.luma_h == 2 && im.luma_v == 1) return 2 + wide;
    if (im.luma_h == 1 && im.luma_v == 1) return 3 + wide;
    return 0;
}
__host__ __device__ inline bool enc_fused_ok(const DevEncImage &im) { return enc_fused_shape(im) != 0; }

// What the sample reader needs from the image descriptor, held in registers (the descriptor is read once).
struct EncSrc {
    const uint8_t *px;
    uint32_t width, height, comps;
    bool rgb;
    int32_t k[8];  // Fix() factors of the RGB -> YCbCr tables
};

// apps/JpegEncode/JpegRgbToYCbCrConverter.cs:64-96: the tables are i * Fix(x) (+ rounding / offset terms), so the terms
// are computed (24-bit multiplies: factors < 2^17, samples < 2^8)
__device__ __forceinline__ int32_t enc_convert(const EncSrc &s, uint32_t c, int32_t b0, int32_t b1, int32_t b2) {
    if (!s.rgb) return c == 0 ? b0 : (c == 1 ? b1 : b2);
    if (c == 0) return (__mul24(s.k[0], b0) + __mul24(s.k[1], b1) + __mul24(s.k[2], b2) + 32768) >> 16;
    if (c == 1) return (__mul24(s.k[5], b2) - __mul24(s.k[3], b0) - __mul24(s.k[4], b1) + (128 << 16) + 32767) >> 16;
    return (__mul24(s.k[5], b0) - __mul24(s.k[6], b1) - __mul24(s.k[7], b2) + (128 << 16) + 32767) >> 16;
}

// One input sample of component c at pixel (x, y); outside the image the reader leaves zeros
// (ref: apps/JpegEncode/JpegBufferInputReader.cs:27-52).
__device__ __forceinline__ int32_t enc_sample(const EncSrc &s, uint32_t c, uint32_t x, uint32_t y) {
    if (x >= s.width || y >= s.height) return 0;
    const uint8_t *p = s.px + ((size_t)y * s.width + x) * s.comps;
    if (s.comps < 3) return p[0];
    return enc_convert(s, c, p[0], p[1], p[2]);  // (a fourth byte per pixel is alpha: ConvertRgba32ToYCbCr8 steps over it)
}

// Component c of P consecutive pixels of one row, all inside the image, the row address dword aligned.
template <int P>
__device__ __forceinline__ void enc_row(const uint8_t *rowp, const EncSrc &s, uint32_t c, int32_t (&val)[P]) {
    if (s.comps == 3) {
        uint32_t w[P * 3 / 4];
        __builtin_memcpy(w, __builtin_assume_aligned(rowp, 4), sizeof w);
#pragma unroll
        for (int i = 0; i < P; i++) {
            const int32_t b0 = (w[(3 * i) >> 2] >> (8 * ((3 * i) & 3))) & 0xFF;
            const int32_t b1 = (w[(3 * i + 1) >> 2] >> (8 * ((3 * i + 1) & 3))) & 0xFF;
            const int32_t b2 = (w[(3 * i + 2) >> 2] >> (8 * ((3 * i + 2) & 3))) & 0xFF;
            val[i] = enc_convert(s, c, b0, b1, b2);
        }
    } else {
        uint32_t w[P / 4];
        __builtin_memcpy(w, __builtin_assume_aligned(rowp, 4), sizeof w);
#pragma unroll
        for (int i = 0; i < P; i++) val[i] = (w[i >> 2] >> (8 * (i & 3))) & 0xFF;
    }
}

// ShiftDataLevel + TransformFDCT + ZigZagAndQuantizeBlock (ref: JpegEncoder.cs:801-826): q[i] = (short)MathF.Round(F / Q)
__device__ __forceinline__ void fdct_quantize(const int32_t (&smp)[64], const uint16_t *quant, int32_t (&q)[64]) {
    float f[64];
#pragma unroll
    for (int i = 0; i < 64; i++) f[i] = (float)(smp[i] - 128);
    block_fdct(f);
#pragma unroll
    for (int i = 0; i < 64; i++) {
        const float v = f[kEncNat[i]] / (float)quant[i];
        q[i] = (int32_t)(int16_t)(int32_t)__builtin_rintf(v);  // half to even; (short) wraps
    }
}

// F / Q as the hardware's own IEEE division computes it, with the part that depends on Q alone taken out of the block.
// hipcc expands a / b into v_div_scale x 2, v_rcp, one Newton step on the reciprocal (fma, fma), q0 = a * r, two
// residual corrections (fma, fma, fma, then v_div_fmas = one more fma) and v_div_fixup -- eleven dependent instructions,
// and because the second v_div_scale hands a flag to v_div_fmas through VCC it does not interleave two divisions: 64 of them
// were a third of the kernel's instructions and most of its stalls.  For a divisor in [1, 65535] and a dividend that is zero
// or not smaller than ~1e-6 in magnitude the scaling steps are the identity and the fix-up returns its first operand, so
// the quotient is exactly  r = rcp(b) refined once;  q0 = a r;  q1 = fma(fma(-b, q0, a), r, q0);  q = fma(fma(-b, q1, a), r, q1)
// -- the same instructions on the same operands, bit for bit (byte-exact encoder tests, tools/stress_parity.py).  b and r
// come from LDS (quant_pair: once per workgroup), the five that remain are independent from coefficient to coefficient.
struct QuantPair {
    float d, r;
};
__device__ __forceinline__ QuantPair quant_pair(uint32_t q) {
    QuantPair p;
    p.d = (float)q;
    float r0;
    asm("v_rcp_f32 %0, %1" : "=v"(r0) : "v"(p.d));  // (the builtin becomes v_rcp_iflag_f32 behind an integer conversion)
    p.r = __builtin_fmaf(__builtin_fmaf(-p.d, r0, 1.0f), r0, r0);
    return p;
}
__device__ __forceinline__ float quant_divide(float a, const QuantPair &p) {
    const float q0 = a * p.r;
    const float q1 = __builtin_fmaf(__builtin_fmaf(-p.d, q0, a), p.r, q0);
    return __builtin_fmaf(__builtin_fmaf(-p.d, q1, a), p.r, q1);
}

// ShiftDataLevel + TransformFDCT + ZigZagAndQuantizeBlock, one row of eight quantised coefficients at a time (zig-zag
// positions 8r .. 8r + 7), handed to `row(r, values)` as soon as they exist: the 64 results never stand in registers together.
template <typename Row>
__device__ __forceinline__ void fdct_quantize_rows(const int32_t (&smp)[64], const QuantPair *quant, Row row) {
    float f[64];
#pragma unroll
    for (int i = 0; i < 64; i++) f[i] = (float)(smp[i] - 128);
    block_fdct(f);
#pragma unroll
    for (int r = 0; r < 8; r++) {
        int32_t q[8];
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const float v = quant_divide(f[kEncNat[r * 8 + i]], quant[r * 8 + i]);
            q[i] = (int32_t)(int16_t)(int32_t)__builtin_rintf(v);  // half to even; (short) wraps
        }
        row(r, q);
    }
}

__device__ __forceinline__ uint4 pack8_i16(const int32_t (&v)[8]) {
    return uint4{((uint32_t)v[0] & 0xFFFFu) | ((uint32_t)v[1] << 16), ((uint32_t)v[2] & 0xFFFFu) | ((uint32_t)v[3] << 16),
                 ((uint32_t)v[4] & 0xFFFFu) | ((uint32_t)v[5] << 16), ((uint32_t)v[6] & 0xFFFFu) | ((uint32_t)v[7] << 16)};
}
__device__ __forceinline__ void unpack8_i16(const uint4 &p, int32_t (&v)[8]) {
    const uint32_t w[4] = {p.x, p.y, p.z, p.w};
#pragma unroll
    for (int i = 0; i < 8; i++) v[i] = (i & 1) ? ((int32_t)w[i >> 1] >> 16) : (int32_t)(int16_t)(w[i >> 1] & 0xFFFFu);
}

// All three components of P consecutive pixels of one row, all inside the image, the row address dword aligned.
template <int P>
__device__ __forceinline__ void enc_row3(const uint8_t *rowp, const EncSrc &s, int32_t (&c0)[P], int32_t (&c1)[P], int32_t (&c2)[P]) {
    uint32_t w[P * 3 / 4];
    __builtin_memcpy(w, __builtin_assume_aligned(rowp, 4), sizeof w);
#pragma unroll
    for (int i = 0; i < P; i++) {
        const int32_t b0 = (w[(3 * i) >> 2] >> (8 * ((3 * i) & 3))) & 0xFF;
        const int32_t b1 = (w[(3 * i + 1) >> 2] >> (8 * ((3 * i + 1) & 3))) & 0xFF;
        const int32_t b2 = (w[(3 * i + 2) >> 2] >> (8 * ((3 * i + 2) & 3))) & 0xFF;
        c0[i] = enc_convert(s, 0, b0, b1, b2);
        c1[i] = enc_convert(s, 1, b0, b1, b2);
        c2[i] = enc_convert(s, 2, b0, b1, b2);
    }
}

// Samples of one MCU as E1b reads them (enc_sample_stride bytes): luma_h x luma_v luma blocks of 64 bytes (the block's 64
// samples, row-major), then for a 3-component image two blocks of 64 int16: per chroma sample the SUM of the converted
// values of the luma_h x luma_v pixels it covers (what ReadBlockWithSubsample adds up, JpegEncoder.cs:788-799; the rounding
// shift -- and, without optimizeCoding, the previous block's coefficients it is added to -- come in E1b).
__host__ __device__ constexpr uint32_t enc_sample_stride(uint32_t luma_h, uint32_t luma_v, uint32_t components) {
    return luma_h * luma_v * 64u + (components > 1 ? (components - 1) * 128u : 0u);
}

// The luma_v pixel rows behind chroma row k of one MCU, HS = luma_h blocks of 8 pixels each: luma samples to their blocks,
// chroma values summed per chroma sample (8 / HS of them per block of 8 pixels: with HS fixed the sums stay in registers).
template <int HS, typename OutPtr>
__device__ __forceinline__ void enc_gather_rows(const EncSrc &src, uint32_t x0, uint32_t ymcu, uint32_t k, uint32_t V, uint32_t components,
                                                bool rows_aligned, OutPtr out, int32_t (&sum1)[8], int32_t (&sum2)[8]) {
    // The usual case -- RGB rows, the lane's pixels all inside the image, at most two pixel rows of at most two blocks --
    // loads everything first (up to 96 bytes per lane in flight instead of 24) and converts afterwards.
    if (HS <= 2 && V <= 2 && src.comps == 3 && components == 3 && rows_aligned && x0 + 8 * HS <= src.width && ymcu + k * V + V <= src.height) {
        uint32_t w[2][HS][6];
#pragma unroll
        for (int dy = 0; dy < 2; dy++)
            if ((uint32_t)dy < V) {
                const uint8_t *rowp = src.px + ((size_t)(ymcu + k * V + dy) * src.width + x0) * 3;
#pragma unroll
                for (int bx = 0; bx < HS; bx++) __builtin_memcpy(w[dy][bx], __builtin_assume_aligned(rowp + bx * 24, 4), 24);
            }
#pragma unroll
        for (int dy = 0; dy < 2; dy++)
            if ((uint32_t)dy < V) {
                const uint32_t ry = k * V + dy, by = ry >> 3, r = ry & 7u;
#pragma unroll
                for (int bx = 0; bx < HS; bx++) {
                    int32_t c0[8], c1[8], c2[8];
#pragma unroll
                    for (int i = 0; i < 8; i++) {
                        const int32_t b0 = (w[dy][bx][(3 * i) >> 2] >> (8 * ((3 * i) & 3))) & 0xFF;
                        const int32_t b1 = (w[dy][bx][(3 * i + 1) >> 2] >> (8 * ((3 * i + 1) & 3))) & 0xFF;
                        const int32_t b2 = (w[dy][bx][(3 * i + 2) >> 2] >> (8 * ((3 * i + 2) & 3))) & 0xFF;
                        c0[i] = enc_convert(src, 0, b0, b1, b2);
                        c1[i] = enc_convert(src, 1, b0, b1, b2);
                        c2[i] = enc_convert(src, 2, b0, b1, b2);
                    }
                    uint2 pk;
                    pk.x = (uint32_t)c0[0] | ((uint32_t)c0[1] << 8) | ((uint32_t)c0[2] << 16) | ((uint32_t)c0[3] << 24);
                    pk.y = (uint32_t)c0[4] | ((uint32_t)c0[5] << 8) | ((uint32_t)c0[6] << 16) | ((uint32_t)c0[7] << 24);
                    *reinterpret_cast<uint2 *>(out + (by * HS + (uint32_t)bx) * 64 + r * 8) = pk;
#pragma unroll
                    for (int i = 0; i < 8; i++) {
                        constexpr int kShift = HS == 1 ? 0 : (HS == 2 ? 1 : 2);
                        sum1[(bx * 8 + i) >> kShift] += c1[i];
                        sum2[(bx * 8 + i) >> kShift] += c2[i];
                    }
                }
            }
        return;
    }
    for (uint32_t dy = 0; dy < V; dy++) {
        const uint32_t ry = k * V + dy;  // pixel row inside the MCU
        const uint32_t y = ymcu + ry;
        const uint32_t by = ry >> 3, r = ry & 7u;
#pragma unroll
        for (int bx = 0; bx < HS; bx++) {
            const uint32_t x = x0 + (uint32_t)bx * 8;
            int32_t c0[8], c1[8], c2[8];
            if (x + 8 <= src.width && y < src.height && rows_aligned && src.comps == 3) {
                enc_row3<8>(src.px + ((size_t)y * src.width + x) * 3, src, c0, c1, c2);
            } else {
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    c0[i] = enc_sample(src, 0, x + i, y);
                    c1[i] = components > 1 ? enc_sample(src, 1, x + i, y) : 0;
                    c2[i] = components > 1 ? enc_sample(src, 2, x + i, y) : 0;
                }
            }
            uint2 pk;
            pk.x = (uint32_t)c0[0] | ((uint32_t)c0[1] << 8) | ((uint32_t)c0[2] << 16) | ((uint32_t)c0[3] << 24);
            pk.y = (uint32_t)c0[4] | ((uint32_t)c0[5] << 8) | ((uint32_t)c0[6] << 16) | ((uint32_t)c0[7] << 24);
            *reinterpret_cast<uint2 *>(out + (by * HS + (uint32_t)bx) * 64 + r * 8) = pk;
            if (components > 1) {
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    constexpr int kShift = HS == 1 ? 0 : (HS == 2 ? 1 : 2);
                    sum1[(bx * 8 + i) >> kShift] += c1[i];  // compile-time index
                    sum2[(bx * 8 + i) >> kShift] += c2[i];
                }
            }
        }
    }
}

// E1a: the pixel pass.  One lane per (MCU, chroma row k of the MCU): the luma_v pixel rows behind that chroma row, 8 x luma_h
// pixels each -- converted once (RGB -> YCbCr when asked to), luma samples to their blocks, chroma values summed.  Lanes of
// a wave are consecutive MCUs of one k, so a wave reads whole contiguous stretches of a pixel row.
// STAGE: the workgroup's 128 MCU records (enc_sample_stride bytes each, contiguous in `samples`) are put together in LDS and
// go out as one contiguous stretch, 16 bytes per lane.  Written where they are produced they are 8- and 16-byte pieces 512
// bytes apart, eight waves contributing to every line at eight different times: WRITE_SIZE showed 2.5 GB leaving the L2
// for the 1.06 GB of samples of 64 images (tools/trace/encoder_pmc.sh).  Not STAGE: records too large for 64 KB of LDS.
template <bool STAGE>
__global__ __launch_bounds__(8 * kEncMcusPerWg) void enc_gather_kernel(const uint8_t *__restrict__ pixels, const DevEncImage *__restrict__ images,
                                                                       const EncWork *__restrict__ work, uint8_t *__restrict__ samples, bool skip_fused) {
    extern __shared__ __attribute__((aligned(16))) uint8_t sh_records[];
    const EncWork wk = work[blockIdx.x];
    const DevEncImage &im = images[wk.image];
    if (skip_fused && enc_fused_ok(im)) return;  // (uniform: fdct_fused_kernel takes the image)
    const uint32_t m = threadIdx.x % kEncMcusPerWg, k = threadIdx.x / kEncMcusPerWg;
    const uint32_t mcus_per_line = im.mcus_per_line;
    const uint32_t n_mcus = mcus_per_line * im.mcus_per_column;
    const bool active = wk.first + m < n_mcus;
    if (!STAGE && !active) return;
    const uint32_t mcu = active ? wk.first + m : n_mcus - 1;
    const uint32_t H = im.luma_h, V = im.luma_v, components = im.components;
    const uint32_t stride = enc_sample_stride(H, V, components);
    uint8_t *global_base = samples + (uint64_t)im.smp_off_256 * 256u;
    if (active) {
        EncSrc src;
        src.px = pixels + im.px_off;
        src.width = im.width;
        src.height = im.height;
        src.comps = im.in_components;
        src.rgb = im.input_rgb != 0;
#pragma unroll
        for (int i = 0; i < 8; i++) src.k[i] = im.r2y[i];
        const uint32_t mx = mcu % mcus_per_line, my = mcu / mcus_per_line;
        const uint32_t x0 = mx * 8 * H;
        const bool rows_aligned = ((src.width * src.comps) & 3u) == 0;
        int32_t sum1[8] = {0, 0, 0, 0, 0, 0, 0, 0}, sum2[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        auto rows = [&](auto *out) {
            if (H == 1) enc_gather_rows<1>(src, x0, my * 8 * V, k, V, components, rows_aligned, out, sum1, sum2);
            else if (H == 2) enc_gather_rows<2>(src, x0, my * 8 * V, k, V, components, rows_aligned, out, sum1, sum2);
            else enc_gather_rows<4>(src, x0, my * 8 * V, k, V, components, rows_aligned, out, sum1, sum2);
            if (components > 1) {
                auto *cb = out + H * V * 64 + k * 16;
                *reinterpret_cast<uint4 *>(cb) = pack8_i16(sum1);
                *reinterpret_cast<uint4 *>(cb + 128) = pack8_i16(sum2);
            }
        };
        if (STAGE) rows(sh_records + m * stride);
        else rows(global_base + (uint64_t)mcu * stride);
    }
    if (STAGE) {
        __syncthreads();
        const uint32_t n_here = n_mcus - wk.first < (uint32_t)kEncMcusPerWg ? n_mcus - wk.first : (uint32_t)kEncMcusPerWg;
        const uint32_t bytes = n_here * stride;  // a multiple of 64
        uint8_t *dst = global_base + (uint64_t)wk.first * stride;
        for (uint32_t o = threadIdx.x * 16u; o < bytes; o += 8u * kEncMcusPerWg * 16u)
            *reinterpret_cast<uint4 *>(dst + o) = *reinterpret_cast<const uint4 *>(sh_records + o);
    }
}

// E1b: one lane per MCU, kEncMcusPerWg lanes per workgroup: the gathered samples through ShiftDataLevel + TransformFDCT +
// ZigZagAndQuantizeBlock.  Blocks of an MCU are produced in encoding order because the reference's sub-sampling reader
// accumulates into the ONE block buffer WriteScanData reuses (:712, :788-799): a sub-sampled component's block starts from
// the previous block's quantised coefficients (kept here in registers, packed).  With optimizeCoding every block has its
// own zeroed allocator slot (TransformBlocks :414-485) and nothing carries over.
// (First version: one kernel, the gather inside the block loop row by row through LDS -- 211 VGPRs and 33 KB of LDS kept
// it at 2 waves per SIMD with a load-use wait per pixel row: 2.4 ms per 64 x 4K, 47 % of the wave cycles waiting.)
#ifndef JPGPU_E1B_WAVES
#define JPGPU_E1B_WAVES 1
#endif
__global__ __launch_bounds__(kEncMcusPerWg, JPGPU_E1B_WAVES) void fdct_quant_kernel(const uint8_t *__restrict__ samples, const DevEncImage *__restrict__ images,
                                                                   const EncWork *__restrict__ work, int16_t *__restrict__ coefs, bool skip_fused) {
    __shared__ QuantPair sh_q[2][64];  // the two quantisation tables as (divisor, refined reciprocal)
    // a wave's 64 blocks (one per lane, 128 bytes each) on their way out: every lane's eight 16-byte rows, then eight lanes per
    // block write it as ONE 128-byte line -- a lane storing its own block touches 64 lines with every store instruction and
    // each of them eight times
    __shared__ __attribute__((aligned(16))) uint8_t sh_blk[kEncMcusPerWg * 128];
    const EncWork wk = work[blockIdx.x];
    const DevEncImage &im = images[wk.image];
    if (skip_fused && enc_fused_ok(im)) return;
    const uint32_t lane = threadIdx.x;
    if (lane < 64) {
        sh_q[0][lane] = quant_pair(im.quant[0][lane]);
        sh_q[1][lane] = quant_pair(im.quant[1][lane]);
    }
    __syncthreads();
    const uint32_t n_mcus = im.mcus_per_line * im.mcus_per_column;
    if (wk.first + (lane & ~63u) >= n_mcus) return;  // (whole waves only: the lanes of a wave write each other's blocks)
    const uint32_t mcu = wk.first + lane < n_mcus ? wk.first + lane : n_mcus - 1;  // lanes behind the last MCU redo it, and store nothing
    const uint32_t components = im.components, bpm = im.bpm;
    const uint32_t H = im.luma_h, V = im.luma_v, ny = H * V;
    const uint32_t total = (31 - __builtin_clz(H)) + (31 - __builtin_clz(V));  // rounding shift of a chroma sample
    const bool own_blocks = im.table_base != 0;
    const uint8_t *in = samples + (uint64_t)im.smp_off_256 * 256u + (uint64_t)mcu * enc_sample_stride(H, V, components);
    uint4 prev[8];
#pragma unroll
    for (int r = 0; r < 8; r++) prev[r] = uint4{0, 0, 0, 0};
    // the samples of block b + 1 are fetched while block b is transformed (8 x 16 bytes; a luma block uses the first four)
    uint4 raw[8];
    auto fetch = [&](uint32_t bb, uint4 (&dst)[8]) {
        const uint4 *p = reinterpret_cast<const uint4 *>(bb < ny ? in + bb * 64 : in + ny * 64 + (bb - ny) * 128);
#pragma unroll
        for (int v4 = 0; v4 < 4; v4++) dst[v4] = p[v4];
        if (bb >= ny) {
#pragma unroll
            for (int v4 = 4; v4 < 8; v4++) dst[v4] = p[v4];
        }
    };
#ifndef JPGPU_E1B_NO_PREFETCH
    fetch(0, raw);
#endif
#pragma unroll 1
    for (uint32_t b = 0; b < bpm; b++) {
        int32_t smp[64];
#ifdef JPGPU_E1B_NO_PREFETCH
        fetch(b, raw);
#endif
        if (b < ny) {
#pragma unroll
            for (int v4 = 0; v4 < 4; v4++) {
                const uint32_t ww[4] = {raw[v4].x, raw[v4].y, raw[v4].z, raw[v4].w};
#pragma unroll
                for (int i = 0; i < 16; i++) smp[v4 * 16 + i] = (int32_t)((ww[i >> 2] >> (8 * (i & 3))) & 0xFFu);
            }
        } else {
#pragma unroll
            for (int r = 0; r < 8; r++) {
                int32_t sum[8], base[8];
                unpack8_i16(raw[r], sum);
                unpack8_i16(prev[r], base);
                // (branch-free: as `if (total == 0) ... else ...` this was a scalar branch per sample, 128 of them per chroma block)
                // (and `total == 0 ? a : b` on a uniform condition becomes a branch again: without sub-sampling the block does not
                // start from the previous one, which is the same as a previous block of zeros)
                const int32_t half = total == 0 ? 0 : 1 << (total - 1), keep = total == 0 ? 0 : -1;
#pragma unroll
                for (int i = 0; i < 8; i++)
                    smp[r * 8 + i] = (int32_t)(int16_t)(((int32_t)(int16_t)((base[i] & keep) + sum[i]) + half) >> total);
            }
        }
#ifndef JPGPU_E1B_NO_PREFETCH
        if (b + 1 < bpm) fetch(b + 1, raw);
#endif
        uint8_t *mine = sh_blk + lane * 128u;
        fdct_quantize_rows(smp, sh_q[b < ny ? 0 : 1], [&](int r, const int32_t (&row)[8]) {
            const uint4 pk = pack8_i16(row);
            *reinterpret_cast<uint4 *>(mine + (((uint32_t)r ^ (lane & 7u)) * 16u)) = pk;  // (swizzled: eight lanes, eight columns)
            if (!own_blocks) prev[r] = pk;  // ZigZagAndQuantizeBlock writes into the buffer the next ReadBlock starts from
        });
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the wave's own LDS writes (no other wave touches its 8 KB)
        {
            const uint32_t w0 = lane & ~63u, sub = (lane & 63u) >> 3, piece = lane & 7u;
#pragma unroll
            for (uint32_t k = 0; k < 8; k++) {
                const uint32_t owner = w0 + k * 8u + sub;  // the lane whose block this is
                const uint4 v = *reinterpret_cast<const uint4 *>(sh_blk + owner * 128u + ((piece ^ (owner & 7u)) * 16u));
                if (wk.first + owner < n_mcus)
                    *reinterpret_cast<uint4 *>(coefs + (im.coef_off + (uint64_t)(wk.first + owner) * bpm + b) * 64 + piece * 8u) = v;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // read before the next block's rows overwrite it
    }
}

// ------------------------------------------------------------------------------------------------ E1 fused
//
// E1 as ONE kernel for the usual shapes (three components from three-sample pixels, luma 2 x 2, 2 x 1 or 1 x 1:
// enc_fused_shape; described for 2 x 2, the others differ in how many MCUs a gather round takes): pixels in,
// quantised blocks out, nothing in between leaves the CU.  E1a + E1b move the gathered samples through HBM (4.2 GB out and
// in again per 256 x 4K) and E1b holds a whole block per lane (245 registers, two waves per SIMD).  Here a WAVE owns 16
// consecutive MCUs and every step runs on all 64 lanes, two rows / columns per lane as one packed float pair:
//   gather (4 rounds of 4 MCUs)  lane = (MCU, pixel row): 16 pixels converted once; chroma: pairs summed in the lane, the row
//                                below added from the neighbour lane (DPP), to LDS; the lane's luma row of the left and the
//                                right block goes through pass 1 (rows) as one packed pair -> transpose buffer
//   luma pass 2 (1 per gather)   lane = (block, columns 2c and 2c + 1): pass 2, x 0.125, quantise, (short) -> the block's
//                                zig-zag positions in the staging buffer; 16 finished blocks leave as whole 128-byte lines
//   Cb, then Cr                  pass 1: lane = (row, MCUs 2m and 2m + 1), the sample = box average started from the PREVIOUS
//                                block's quantised coefficients (ReadBlockWithSubsample adds into the one buffer
//                                WriteScanData reuses: Cb from Y3, Cr from Cb; not with optimizeCoding); pass 2 as for luma
// A wave never waits for another one (LDS operations of one wave complete in order): no barrier in the kernel.
// Arithmetic is E1a's and E1b's, operation for operation (the FDCT butterflies as v_pk_add_f32 / v_pk_mul_f32, the quotient
// of quant_divide as v_pk_mul_f32 + 4 v_pk_fma_f32 on the same operands).

// natural index (8 r + c) -> zig-zag position: the inverse of kEncNat
__device__ constexpr uint8_t kEncZig[64] = {0,  1,  5,  6,  14, 15, 27, 28, 2,  4,  7,  13, 16, 26, 29, 42, 3,  8,  12, 17, 25, 30,
                                            41, 43, 9,  11, 18, 24, 31, 40, 44, 53, 10, 19, 23, 32, 39, 45, 52, 54, 20, 22, 33, 38,
                                            46, 51, 55, 60, 21, 34, 37, 47, 50, 56, 59, 61, 35, 36, 48, 49, 57, 58, 62, 63};

constexpr int kEfMcus = 16;                 // MCUs per wave (= workgroup)
constexpr uint32_t kEfS = 72u;              // transpose buffer: dwords from block to block (72: the column pairs of eight
                                            // blocks, four lanes each, fall on 64 different banks)
constexpr uint32_t kEfQuant = 0;            // the table in use as {d(2c), d(2c+1), r(2c), r(2c+1)} per (row, column pair)
constexpr uint32_t kEfT = kEfQuant + 512u;  // 16 blocks after pass 1, row-major
constexpr uint32_t kEfSums = kEfT + 16u * kEfS * 4u;          // chroma 2 x 2 sums: [MCU][Cb, Cr][row] x 8 int16
constexpr uint32_t kEfStage = kEfSums + kEfMcus * 2u * 128u;  // 16 finished blocks (zig-zag int16) on their way out,
constexpr uint32_t kEfStageStride = 144u;                     // 144 bytes apart: a coefficient position of eight blocks on eight banks
constexpr uint32_t kEfCarry = kEfStage + 16u * kEfStageStride;  // block Y3 of every MCU: what its Cb block starts from
constexpr uint32_t kEfLdsBytes = kEfCarry + kEfMcus * 128u;   // 13 568: twelve waves per CU

// the lane's entry of the quantisation table in use (lane = natural index): divisor and refined reciprocal
__device__ __forceinline__ void ef_quant_table(uint8_t *sh, const uint16_t *quant_zigzag, uint32_t lane) {
    const QuantPair a = quant_pair(quant_zigzag[kEncZig[lane]]);
    float *e = reinterpret_cast<float *>(sh + kEfQuant + (lane >> 1) * 16u) + (lane & 1u);
    e[0] = a.d;
    e[2] = a.r;
}

// x + 1.5 * 2^23 for |x| < 2^22: the sum lies where floats are 1 apart, so the addition itself rounds x to the nearest integer,
// ties to even (MathF.Round), and the low 16 bits of the result are that integer as a (short) -- rint + convert in one addition
constexpr float kEfRound = 12582912.0f;
__device__ __forceinline__ EncF2 ef_floor(EncF2 a) { return EncF2{__builtin_floorf(a.x), __builtin_floorf(a.y)}; }

// pass 2 of the 16 blocks in the transpose buffer: lane (q, c2) takes columns 2 c2 and 2 c2 + 1 of block q
__device__ __forceinline__ void ef_pass2(uint8_t *sh, uint32_t q, uint32_t c2, const uint32_t (&za)[8], const uint32_t (&zb)[8]) {
    const EncF2 *t = reinterpret_cast<const EncF2 *>(sh + kEfT + (q * kEfS + 2u * c2) * 4u);
    EncF2 v[8];
#pragma unroll
    for (int r = 0; r < 8; r++) v[r] = t[r * 4];
    fdct8(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
    const float4 *qt = reinterpret_cast<const float4 *>(sh + kEfQuant + c2 * 16u);
    float4 dr[8];
#pragma unroll
    for (int r = 0; r < 8; r++) dr[r] = qt[r * 4];
#pragma unroll
    for (int r = 0; r < 8; r += 2) {
        // quant_divide on both halves; rows r and r + 1 statement by statement side by side (a packed operation waits a cycle
        // for the one before it when it needs its result)
        const EncF2 a0 = v[r] * 0.1250f, a1 = v[r + 1] * 0.1250f;
        const EncF2 D0 = EncF2{dr[r].x, dr[r].y}, R0 = EncF2{dr[r].z, dr[r].w}, D1 = EncF2{dr[r + 1].x, dr[r + 1].y}, R1 = EncF2{dr[r + 1].z, dr[r + 1].w};
        const EncF2 q00 = a0 * R0, q01 = a1 * R1;
        const EncF2 e00 = __builtin_elementwise_fma(-D0, q00, a0), e01 = __builtin_elementwise_fma(-D1, q01, a1);
        const EncF2 q10 = __builtin_elementwise_fma(e00, R0, q00), q11 = __builtin_elementwise_fma(e01, R1, q01);
        const EncF2 e10 = __builtin_elementwise_fma(-D0, q10, a0), e11 = __builtin_elementwise_fma(-D1, q11, a1);
        const EncF2 qq0 = __builtin_elementwise_fma(e10, R0, q10), qq1 = __builtin_elementwise_fma(e11, R1, q11);
        // (short)MathF.Round(.)  (scalar copies: __builtin_bit_cast of a vector ELEMENT reads the vector's first one)
        const EncF2 r0 = qq0 + EncF2{kEfRound, kEfRound}, r1 = qq1 + EncF2{kEfRound, kEfRound};
        const float x0 = r0.x, y0 = r0.y, x1 = r1.x, y1 = r1.y;
        *reinterpret_cast<uint16_t *>(sh + za[r]) = (uint16_t)(__builtin_bit_cast(uint32_t, x0) & 0xFFFFu);
        *reinterpret_cast<uint16_t *>(sh + zb[r]) = (uint16_t)(__builtin_bit_cast(uint32_t, y0) & 0xFFFFu);
        *reinterpret_cast<uint16_t *>(sh + za[r + 1]) = (uint16_t)(__builtin_bit_cast(uint32_t, x1) & 0xFFFFu);
        *reinterpret_cast<uint16_t *>(sh + zb[r + 1]) = (uint16_t)(__builtin_bit_cast(uint32_t, y1) & 0xFFFFu);
    }
}

// JpegRgbToYCbCrConverter in float: the tables are i * Fix(x) (+ rounding / offset terms) >> 16 with Fix(x) < 2^17 and i < 2^8.
// Scaled by 2^-16 every factor, product and partial sum (taken in the order below: below 256 in magnitude, 16 fractional bits)
// is exact in a float's 24 bits, so floor(.) is the table lookup's integer -- and v_cvt_f32_ubyteN takes the byte out of its
// dword for free.  The luma offset carries ShiftDataLevel's -128.  Pixels that are Y, Cb, Cr already: factors 1 / 0.
struct EfConvert {
    float y[3], b[3], r[3], oy, oc;  // luma: R G B; Cb: R G B; Cr: R G B (the negative ones first in the chain)
};

// One pixel row of an MCU (8 H pixels in w): luma row through pass 1 into the transpose buffer (block tblk, and tblk + 1 for
// the right half when H = 2), chroma sums into the sums buffer -- pairs summed in the lane when H = 2, the row below (above)
// added from the neighbour lane when V = 2.  EDGE: pixels outside the image (`inside`: one bit per pixel).
template <int H, int V, int BPP, bool EDGE>
__device__ __forceinline__ void ef_row(uint8_t *sh, const uint32_t (&w)[2 * BPP * H], uint32_t inside, const EfConvert &cv, uint32_t mloc, uint32_t gk,
                                       uint32_t tblk, uint32_t gdy, uint32_t ry) {
    float sb[8], sr[8];
    EncF2 v[8];  // H = 2: left block in .x, right block in .y; H = 1: .x
#pragma unroll
    for (int i = 0; i < 8 * H; i += 2) {
        EncF2 c[3];
#pragma unroll
        for (int ch = 0; ch < 3; ch++) {
            const int b0 = BPP * i + ch, b1 = BPP * (i + 1) + ch;
            c[ch] = EncF2{(float)((w[b0 >> 2] >> (8 * (b0 & 3))) & 0xFFu), (float)((w[b1 >> 2] >> (8 * (b1 & 3))) & 0xFFu)};
        }
        const EncF2 yv = ef_floor(__builtin_elementwise_fma(EncF2{cv.y[0], cv.y[0]}, c[0],
                             __builtin_elementwise_fma(EncF2{cv.y[1], cv.y[1]}, c[1],
                             __builtin_elementwise_fma(EncF2{cv.y[2], cv.y[2]}, c[2], EncF2{cv.oy, cv.oy}))));
        const EncF2 bv = ef_floor(__builtin_elementwise_fma(EncF2{cv.b[2], cv.b[2]}, c[2],
                             __builtin_elementwise_fma(EncF2{cv.b[1], cv.b[1]}, c[1],
                             __builtin_elementwise_fma(EncF2{cv.b[0], cv.b[0]}, c[0], EncF2{cv.oc, cv.oc}))));
        const EncF2 rv = ef_floor(__builtin_elementwise_fma(EncF2{cv.r[0], cv.r[0]}, c[0],
                             __builtin_elementwise_fma(EncF2{cv.r[2], cv.r[2]}, c[2],
                             __builtin_elementwise_fma(EncF2{cv.r[1], cv.r[1]}, c[1], EncF2{cv.oc, cv.oc}))));
        if (H == 2) {
            if (i < 8) v[i & 7].x = yv.x, v[(i + 1) & 7].x = yv.y;
            else v[i & 7].y = yv.x, v[(i + 1) & 7].y = yv.y;
            sb[(i >> 1) & 7] = bv.x + bv.y;
            sr[(i >> 1) & 7] = rv.x + rv.y;
        } else {
            v[i & 7].x = yv.x, v[(i + 1) & 7].x = yv.y;
            sb[i & 7] = bv.x, sb[(i + 1) & 7] = bv.y;
            sr[i & 7] = rv.x, sr[(i + 1) & 7] = rv.y;
        }
    }
    if (EDGE) {
        // a pixel outside the image is the SAMPLE zero in every component, not the conversion of a black pixel: the zero
        // bytes gave luma 0 (- 128) but chroma floor(oc) (128 from RGB) -- taken out of the sums again
        const float off = __builtin_floorf(cv.oc);
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const float n_out = H == 2 ? (float)(2 - (int32_t)__builtin_popcount((inside >> (2 * j)) & 3u)) : (float)(1u - ((inside >> j) & 1u));
            sb[j] -= n_out * off;
            sr[j] -= n_out * off;
        }
    }
    // sums -> int16 (x + 2^23: the low bits are the small, non-negative integer.  Scalar copies: __builtin_bit_cast of a vector
    // ELEMENT reads the vector's first one)
    auto pack2 = [](float a, float b) {
        const float lo = a + 8388608.0f, hi = b + 8388608.0f;
        return __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, hi), __builtin_bit_cast(uint32_t, lo), 0x05040100u);
    };
    if (V == 2) {
        // the pixel row below (above) is the neighbour lane's: the even lane keeps Cb, the odd lane Cr
        uint32_t both[4];
#pragma unroll
        for (int j = 0; j < 8; j += 2) {
            float tot[2];
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const float mine = gdy ? sr[j + u] : sb[j + u], give = gdy ? sb[j + u] : sr[j + u];
                tot[u] = mine + __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, give), 0xB1, 0xF, 0xF, true));  // quad_perm [1,0,3,2]
            }
            both[j >> 1] = pack2(tot[0], tot[1]);
        }
        *reinterpret_cast<uint4 *>(sh + kEfSums + ((mloc * 2u + gdy) * 8u + gk) * 16u) = uint4{both[0], both[1], both[2], both[3]};
    } else {
        *reinterpret_cast<uint4 *>(sh + kEfSums + ((mloc * 2u + 0u) * 8u + gk) * 16u) =
            uint4{pack2(sb[0], sb[1]), pack2(sb[2], sb[3]), pack2(sb[4], sb[5]), pack2(sb[6], sb[7])};
        *reinterpret_cast<uint4 *>(sh + kEfSums + ((mloc * 2u + 1u) * 8u + gk) * 16u) =
            uint4{pack2(sr[0], sr[1]), pack2(sr[2], sr[3]), pack2(sr[4], sr[5]), pack2(sr[6], sr[7])};
    }
    // pass 1 of the lane's luma row (H = 2: left and right block side by side)
    float *t = reinterpret_cast<float *>(sh + kEfT) + tblk * kEfS + (ry & 7u) * 8u;
    if (H == 2) {
        fdct8(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
#pragma unroll
        for (int i = 0; i < 8; i++) {
            t[i] = v[i].x;
            t[kEfS + i] = v[i].y;
        }
    } else {
        float f[8];
#pragma unroll
        for (int i = 0; i < 8; i++) f[i] = v[i].x;
        fdct8(f[0], f[1], f[2], f[3], f[4], f[5], f[6], f[7]);
#pragma unroll
        for (int i = 0; i < 8; i++) t[i] = f[i];
    }
}

// H x V = the luma sampling factors: 2 x 2 (4:2:0), 2 x 1 (4:2:2), 1 x 1 (4:4:4); enc_fused_shape() names the instance.
// BPP = bytes per input pixel: 3, or 4 = R, G, B and an alpha byte nobody reads (the reference's EncoderBenchmark hands Rgba32
// pixels to ConvertRgba32ToYCbCr8, tests/JpegLibrary.Benchmarks/ColorConverters/JpegRgbToYCbCrConverter.cs:95-124).
template <int H, int V, int BPP>
__global__ __launch_bounds__(64) void fdct_fused_kernel(const uint8_t *__restrict__ pixels, const DevEncImage *__restrict__ images,
                                                        const EncWork *__restrict__ work, int16_t *__restrict__ coefs) {
    constexpr uint32_t kRows = 8 * V;                    // pixel rows of an MCU
    constexpr uint32_t kPerRound = 64 / kRows;           // MCUs a gather round takes (one lane per pixel row)
    constexpr uint32_t kRounds = kEfMcus / kPerRound;
    constexpr uint32_t kNY = H * V, kBpm = kNY + 2;      // luma blocks / blocks per MCU
    constexpr uint32_t kRoundsPerPass = 16 / (kPerRound * kNY);  // gather rounds that fill the transpose buffer's 16 blocks
    constexpr uint32_t kMcuW = 8 * H, kMcuH = 8 * V, kWords = 2 * BPP * H;
    constexpr int kShift = (H == 2 ? 1 : 0) + (V == 2 ? 1 : 0);  // pixels per chroma sample = 1 << kShift
    __shared__ __attribute__((aligned(16))) uint8_t sh[kEfLdsBytes];
    constexpr uint32_t kPerItem = kEncMcusPerWg / kEfMcus;
    const EncWork wk = work[blockIdx.x / kPerItem];
    const DevEncImage &im = images[wk.image];
    if (enc_fused_shape(im) != (H == 2 ? (V == 2 ? 1 : 2) : 3) + (BPP == 4 ? 3 : 0)) return;  // (another instance, or E1a + E1b, takes the image)
    const uint32_t mcus_per_line = im.mcus_per_line;
    const uint32_t n_mcus = mcus_per_line * im.mcus_per_column;
    const uint32_t base = wk.first + (blockIdx.x % kPerItem) * kEfMcus;
    if (base >= n_mcus) return;
    const uint32_t lane = threadIdx.x;
    const uint32_t width = im.width, height = im.height;
    const uint8_t *px = pixels + im.px_off;
    const bool own_blocks = im.table_base != 0;
    int16_t *out = coefs + im.coef_off * 64;

    ef_quant_table(sh, im.quant[0], lane);
    // pass 2: the lane's block and column pair, and where its 16 results go in the staging buffer
    const uint32_t p2q = lane >> 2, p2c = lane & 3u;
    uint32_t za[8], zb[8];
#pragma unroll
    for (int r = 0; r < 8; r++) {
        za[r] = kEfStage + p2q * kEfStageStride + 2u * kEncZig[r * 8 + 2 * p2c];
        zb[r] = kEfStage + p2q * kEfStageStride + 2u * kEncZig[r * 8 + 2 * p2c + 1];
    }
    EfConvert cv;
    if (im.input_rgb != 0) {
        constexpr float k = 1.0f / 65536.0f;
        cv.y[0] = (float)im.r2y[0] * k, cv.y[1] = (float)im.r2y[1] * k, cv.y[2] = (float)im.r2y[2] * k;
        cv.b[0] = (float)-im.r2y[3] * k, cv.b[1] = (float)-im.r2y[4] * k, cv.b[2] = (float)im.r2y[5] * k;
        cv.r[0] = (float)im.r2y[5] * k, cv.r[1] = (float)-im.r2y[6] * k, cv.r[2] = (float)-im.r2y[7] * k;
        cv.oy = (float)(32768 - (128 << 16)) * k, cv.oc = (float)((128 << 16) + 32767) * k;
    } else {
        cv.y[0] = 1.0f, cv.y[1] = 0.0f, cv.y[2] = 0.0f;
        cv.b[0] = 0.0f, cv.b[1] = 1.0f, cv.b[2] = 0.0f;
        cv.r[0] = 0.0f, cv.r[1] = 0.0f, cv.r[2] = 1.0f;
        cv.oy = -128.0f, cv.oc = 0.0f;
    }
    const uint32_t row_bytes = width * (uint32_t)BPP;
    const bool rows4 = (row_bytes & 3u) == 0;

    // ---- luma: rounds of gather + pass 1; pass 2 and out whenever the transpose buffer holds 16 blocks
    // gather: lane = (MCU of the round, pixel row inside the MCU); chroma row, row of the pair.  (Eight neighbouring lanes put
    // eight different rows of one block column into the transpose buffer: two lanes per bank group instead of four.)
    const uint32_t gm = lane / kRows, ry = lane % kRows, gk = ry >> (V - 1), gdy = V == 2 ? (lane & 1u) : 0u;
    // The usual case -- pixel rows of an MCU that start on 16 (8 for H = 1) bytes -- fetches the pixels of round g + 1 while
    // round g goes through pass 2: loads from an address clamped into the image, so that they can be issued before anybody
    // knows whether the round touches the image's edge (then they are dropped and the edge variant reads byte by byte).
    const bool ahead = (row_bytes & (H == 2 ? 15u : 7u)) == 0 && width >= kMcuW && height >= 1;
    uint32_t w[kWords];
#pragma unroll
    for (uint32_t j = 0; j < kWords; j++) w[j] = 0;
    auto fetch = [&](uint32_t g) {
        const uint32_t mloc = g * kPerRound + gm;
        const uint32_t mcu = base + mloc < n_mcus ? base + mloc : n_mcus - 1;
        uint32_t x0 = (mcu % mcus_per_line) * kMcuW, y = (mcu / mcus_per_line) * kMcuH + ry;
        x0 = x0 + kMcuW <= width ? x0 : width - kMcuW;
        y = y < height ? y : height - 1;
        const uint8_t *rowp = px + ((size_t)y * width + x0) * BPP;
        if constexpr (H == 2) {
#pragma unroll
            for (int j = 0; j < BPP; j++) {
                const uint4 t = reinterpret_cast<const uint4 *>(rowp)[j];
                w[4 * j] = t.x, w[4 * j + 1] = t.y, w[4 * j + 2] = t.z, w[4 * j + 3] = t.w;
            }
        } else {
#pragma unroll
            for (int j = 0; j < BPP; j++) {
                const uint2 t = reinterpret_cast<const uint2 *>(rowp)[j];
                w[2 * j] = t.x, w[2 * j + 1] = t.y;
            }
        }
    };
    if (ahead) fetch(0);
#pragma unroll 1
    for (uint32_t g = 0; g < kRounds; g++) {
        {
            const uint32_t mloc = g * kPerRound + gm;
            const uint32_t mcu = base + mloc < n_mcus ? base + mloc : n_mcus - 1;  // lanes behind the last MCU redo it, and store nothing
            const uint32_t x0 = (mcu % mcus_per_line) * kMcuW, y = (mcu / mcus_per_line) * kMcuH + ry;
            const uint32_t tblk = (g % kRoundsPerPass) * (kPerRound * kNY) + gm * kNY + (V == 2 ? (ry >> 3) * H : 0u);
            const bool whole = x0 + kMcuW <= width && y < height;
            if (ahead && __builtin_amdgcn_ballot_w64(!whole) == 0) {
                ef_row<H, V, BPP, false>(sh, w, 0xFFFFu, cv, mloc, gk, tblk, gdy, ry);
            } else {
                // the edge of the image (outside it the reader leaves zeros, JpegBufferInputReader.cs:27-52), or rows that do
                // not start where the wide loads want them
                const uint8_t *rowp = px + ((size_t)y * width + x0) * BPP;
                uint32_t we[kWords];
                uint32_t inside = 0xFFFFu;  // the row's pixels that lie inside the image
                if (whole && rows4) {
                    __builtin_memcpy(we, __builtin_assume_aligned(rowp, 4), kWords * 4);
                } else {
#pragma unroll
                    for (uint32_t j = 0; j < kWords; j++) we[j] = 0;
                    inside = 0;
                    if (y < height) {
                        const uint32_t np = x0 >= width ? 0u : (width - x0 < kMcuW ? width - x0 : kMcuW), nb = np * (uint32_t)BPP;
                        inside = (1u << np) - 1u;
                        for (uint32_t j = 0; j < nb; j++) {
                            const uint32_t bv = (uint32_t)rowp[j] << (8u * (j & 3u));
#pragma unroll
                            for (uint32_t q = 0; q < kWords; q++)
                                if ((j >> 2) == q) we[q] |= bv;
                        }
                    }
                }
                ef_row<H, V, BPP, true>(sh, we, inside, cv, mloc, gk, tblk, gdy, ry);
            }
            if (ahead && g + 1 < kRounds) fetch(g + 1);
        }
        if ((g + 1) % kRoundsPerPass != 0) continue;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        ef_pass2(sh, p2q, p2c, za, zb);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const uint32_t pass_first = (g + 1 - kRoundsPerPass) * kPerRound;  // first MCU (of the wave's 16) in the transpose buffer
#pragma unroll
        for (uint32_t j = 0; j < 2; j++) {
            const uint32_t i = lane + 64u * j, q = i >> 3, piece = i & 7u;
            const uint32_t mloc = pass_first + q / kNY, yb = q % kNY;
            const uint4 val = *reinterpret_cast<const uint4 *>(sh + kEfStage + q * kEfStageStride + piece * 16u);
            if (base + mloc < n_mcus) *reinterpret_cast<uint4 *>(out + ((uint64_t)(base + mloc) * kBpm + yb) * 64u + piece * 8u) = val;
            if (kShift > 0 && yb == kNY - 1) *reinterpret_cast<uint4 *>(sh + kEfCarry + mloc * 128u + piece * 16u) = val;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    // ---- chroma: Cb of the 16 MCUs, then Cr
    ef_quant_table(sh, im.quant[1], lane);
    // a sub-sampled component's block starts from the previous block's coefficients (not with optimizeCoding); a component
    // that is not sub-sampled is read like a luma block (ReadBlock: nothing carried over, nothing to round)
    const uint32_t keep = (own_blocks || kShift == 0) ? 0u : 0xFFFFFFFFu;
    constexpr float kScale = 1.0f / (float)(1 << kShift), kOffset = (kShift > 0 ? 0.5f : 0.0f) - 128.0f;
    const uint32_t cr = lane & 7u, cm = (lane >> 3) * 2u;  // pass 1: row, first MCU of the pair
#pragma unroll 1
    for (uint32_t comp = 0; comp < 2; comp++) {
        {
            const uint8_t *from = sh + (comp == 0 ? kEfCarry : kEfStage);
            const uint32_t from_stride = comp == 0 ? 128u : kEfStageStride;
            EncF2 v[8];
#pragma unroll
            for (uint32_t side = 0; side < 2; side++) {
                const uint32_t m = cm + side;
                const uint4 sum = *reinterpret_cast<const uint4 *>(sh + kEfSums + ((m * 2u + comp) * 8u + cr) * 16u);
                const uint4 prev = *reinterpret_cast<const uint4 *>(from + m * from_stride + cr * 16u);
                const uint32_t sw[4] = {sum.x, sum.y, sum.z, sum.w}, pw[4] = {prev.x, prev.y, prev.z, prev.w};
#pragma unroll
                for (int i = 0; i < 8; i += 2) {
                    // (short)(previous + sum): 16-bit wrap-around add on both halves; then ((int)t + half) >> shift, - 128, as
                    // floor(t / 2^shift + 0.5 - 128) -- exact in float
                    typedef short S2 __attribute__((ext_vector_type(2)));
                    typedef unsigned short U2 __attribute__((ext_vector_type(2)));
                    const S2 t = __builtin_bit_cast(S2, __builtin_bit_cast(U2, pw[i >> 1] & keep) + __builtin_bit_cast(U2, sw[i >> 1]));
                    const EncF2 f = ef_floor(__builtin_elementwise_fma(EncF2{(float)t.x, (float)t.y}, EncF2{kScale, kScale}, EncF2{kOffset, kOffset}));
                    if (side == 0) v[i].x = f.x, v[i + 1].x = f.y;
                    else v[i].y = f.x, v[i + 1].y = f.y;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            fdct8(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
            float *t = reinterpret_cast<float *>(sh + kEfT) + cm * kEfS + cr * 8u;
#pragma unroll
            for (int i = 0; i < 8; i++) {
                t[i] = v[i].x;
                t[kEfS + i] = v[i].y;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        ef_pass2(sh, p2q, p2c, za, zb);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (uint32_t j = 0; j < 2; j++) {
            const uint32_t i = lane + 64u * j, m = i >> 3, piece = i & 7u;
            const uint4 val = *reinterpret_cast<const uint4 *>(sh + kEfStage + m * kEfStageStride + piece * 16u);
            if (base + m < n_mcus) *reinterpret_cast<uint4 *>(out + ((uint64_t)(base + m) * kBpm + kNY + comp) * 64u + piece * 8u) = val;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
}

// ------------------------------------------------------------------------------------------------ E2 / E3

// bits needed for |a| (ref: BitCountTable, JpegEncoder.cs:938-956)
__device__ __forceinline__ uint32_t enc_bit_count(uint32_t a) { return a ? 32u - (uint32_t)__builtin_clz(a) : 0u; }

// optimizeCoding keeps the blocks in a JpegBlockAllocator: blocks of the MCU grid outside the component's own grid of
// ceil(ceil(W / 8) / hs) x ceil(ceil(H / 8) / vs) blocks all alias its ONE dummy block (JpegBlockAllocator.cs:93-114), so
// the statistics and the scan see, for each of them, what the LAST such block of TransformBlocks left there.  Only the
// luma component can leave its grid (the others are 1 x 1 per MCU), and whenever some block does, the last luma block of
// the last MCU does: that block's coefficients stand in for all of them.
__device__ __forceinline__ uint32_t enc_source_block(const DevEncImage &im, uint32_t blk) {
    if (im.table_base == 0) return blk;
    const uint32_t ny = im.luma_h * im.luma_v;
    const uint32_t mcu = blk / im.bpm, b = blk - mcu * im.bpm;
    if (b >= ny) return blk;
    const uint32_t mx = mcu % im.mcus_per_line, my = mcu / im.mcus_per_line;
    const uint32_t bx = mx * im.luma_h + b % im.luma_h, by = my * im.luma_v + b / im.luma_h;
    if (bx < (im.width + 7) / 8 && by < (im.height + 7) / 8) return blk;
    return (im.mcus_per_line * im.mcus_per_column - 1) * im.bpm + ny - 1;
}

// scan position of a block: component, and the DC value it is predicted from (EncodeBlock :835-838)
__device__ __forceinline__ int32_t enc_dc_predictor(const DevEncImage &im, const int16_t *img_coefs, uint32_t mcu, uint32_t b, uint32_t &comp) {
    const uint32_t ny = im.luma_h * im.luma_v;
    comp = b < ny ? 0u : b - ny + 1u;
    if (comp == 0 && b > 0) return img_coefs[(size_t)enc_source_block(im, mcu * im.bpm + b - 1) * 64];
    if (mcu == 0) return 0;
    if (im.restart_interval != 0 && mcu % im.restart_interval == 0) return 0;  // first MCU of a restart interval (extension)
    const uint32_t pb = comp == 0 ? ny - 1 : b;
    return img_coefs[(size_t)enc_source_block(im, (mcu - 1) * im.bpm + pb) * 64];
}

// Walks the symbols of one block in EncodeBlock order (:828-870) and hands (code, length) pairs to `put`.
template <typename Put>
__device__ __forceinline__ void enc_block_symbols(const uint4 (&cv)[8], int32_t dc_pred, const EncHuffTable &dc, const EncHuffTable &ac, Put put) {
    uint32_t w[32] = {cv[0].x, cv[0].y, cv[0].z, cv[0].w, cv[1].x, cv[1].y, cv[1].z, cv[1].w, cv[2].x, cv[2].y, cv[2].z,
                      cv[2].w, cv[3].x, cv[3].y, cv[3].z, cv[3].w, cv[4].x, cv[4].y, cv[4].z, cv[4].w, cv[5].x, cv[5].y,
                      cv[5].z, cv[5].w, cv[6].x, cv[6].y, cv[6].z, cv[6].w, cv[7].x, cv[7].y, cv[7].z, cv[7].w};
    auto coef = [&](int i) -> int32_t { return (i & 1) ? ((int32_t)w[i >> 1] >> 16) : (int32_t)(int16_t)(w[i >> 1] & 0xFFFFu); };
    // EncodeRunLength (:893-918)
    auto run_length = [&](const EncHuffTable &t, uint32_t run, int32_t value) {
        int32_t a = value, bb = value;
        if (a < 0) {
            a = -value;
            bb = value - 1;
        }
        const uint32_t bits = enc_bit_count((uint32_t)a & 0xFFFFu);
        const uint32_t sym = ((run << 4) | bits) & 0xFFu;
        // the code and the magnitude bits behind it as ONE field (16 + 15 bits at most): one call of the bit writer per symbol
        put((t.code[sym] << bits) | ((uint32_t)bb & ((1u << bits) - 1u)), t.len[sym] + bits);
    };
    const int32_t dcv = coef(0);
    run_length(dc, 0, dcv - dc_pred);
    uint32_t run = 0;
    auto ac_coefficient = [&](int32_t t) {
        if (t == 0) {
            run++;
        } else {
            while (run > 15) {
                put(ac.code[0xF0], ac.len[0xF0]);
                run -= 16;
            }
            run_length(ac, run, t);
            run = 0;
        }
    };
    // Two coefficients (one register) at a time, every register named at compile time: most of a block is zero, and a wave
    // steps over a register in three instructions when it is zero in all of its lanes.  (One coefficient per trip of a
    // loop with a run-time register index cost ~21 instructions per coefficient, zero or not.)
    ac_coefficient((int32_t)w[0] >> 16);
#pragma unroll
    for (int k = 1; k < 32; k++) {
        if (w[k] == 0) {
            run += 2;
        } else {
            ac_coefficient((int32_t)(int16_t)(w[k] & 0xFFFFu));
            ac_coefficient((int32_t)w[k] >> 16);
        }
    }
    if (run > 0) put(ac.code[0], ac.len[0]);
}

// The four Huffman encoding tables of an image (DC0, AC0, DC1, AC1: 3 KB) staged in LDS: every symbol of every lane looks two of
// their entries up.
__device__ __forceinline__ void enc_stage_tables(const EncHuffTable *__restrict__ tables, uint32_t table_base, EncHuffTable *sh_tab) {
    const uint32_t *src = reinterpret_cast<const uint32_t *>(tables + table_base);
    uint32_t *dst = reinterpret_cast<uint32_t *>(sh_tab);
    for (uint32_t i = threadIdx.x; i < 4u * (uint32_t)sizeof(EncHuffTable) / 4u; i += blockDim.x) dst[i] = src[i];
    __syncthreads();
}

// E2: bits of every block (EncodeBlock with a counting writer).  bits[] receives the block's offset inside its workgroup
// of 256 blocks, wg_bits[] the workgroup's total: the image-wide scan (block_offsets_kernel) then runs over one entry per
// workgroup instead of one per block.
__global__ __launch_bounds__(256) void block_bits_kernel(const DevEncImage *__restrict__ images, const EncWork *__restrict__ work,
                                                         const EncHuffTable *__restrict__ tables, const int16_t *__restrict__ coefs,
                                                         uint32_t *__restrict__ bits, uint32_t *__restrict__ wg_bits) {
    __shared__ EncHuffTable sh_tab[4];
    const EncWork wk = work[blockIdx.x];
    const DevEncImage &im = images[wk.image];
    enc_stage_tables(tables, im.table_base, sh_tab);
    const uint32_t blk = wk.first + threadIdx.x;  // block, or restart interval
    uint32_t n = 0;
    const int16_t *img_coefs = coefs + im.coef_off * 64;
    if (im.restart_interval != 0) {
        // one lane per restart interval: the bits of all its blocks, rounded up to whole bytes (the interval ends with one-bit
        // padding and the next one starts on a byte boundary: its marker is not part of the bit stream)
        if (blk < im.n_units) {
            const uint32_t first_mcu = blk * im.restart_interval, total_mcus = im.mcus_per_line * im.mcus_per_column;
            const uint32_t end_mcu = first_mcu + im.restart_interval < total_mcus ? first_mcu + im.restart_interval : total_mcus;
            for (uint32_t mcu = first_mcu; mcu < end_mcu; mcu++)
                for (uint32_t b = 0; b < im.bpm; b++) {
                    uint32_t comp;
                    const int32_t pred = enc_dc_predictor(im, img_coefs, mcu, b, comp);
                    const uint4 *src = reinterpret_cast<const uint4 *>(img_coefs + (size_t)enc_source_block(im, mcu * im.bpm + b) * 64);
                    const uint4 cv[8] = {src[0], src[1], src[2], src[3], src[4], src[5], src[6], src[7]};
                    enc_block_symbols(cv, pred, sh_tab[comp == 0 ? 0 : 2], sh_tab[comp == 0 ? 1 : 3],
                                      [&](uint32_t, uint32_t len) { n += len; });
                }
            n = (n + 7u) & ~7u;
        }
    } else if (blk < im.total_blocks) {
    const uint32_t mcu = blk / im.bpm, b = blk - mcu * im.bpm;
    uint32_t comp;
    const int32_t pred = enc_dc_predictor(im, img_coefs, mcu, b, comp);
    const uint4 *src = reinterpret_cast<const uint4 *>(img_coefs + (size_t)enc_source_block(im, blk) * 64);
    const uint4 cv[8] = {src[0], src[1], src[2], src[3], src[4], src[5], src[6], src[7]};
    enc_block_symbols(cv, pred, sh_tab[comp == 0 ? 0 : 2], sh_tab[comp == 0 ? 1 : 3], [&](uint32_t, uint32_t len) { n += len; });
    }
    // exclusive scan of n over the workgroup
    __shared__ uint32_t sh_wave[4];
    uint32_t incl = n;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(incl, o, 64);
        if ((threadIdx.x & 63) >= (uint32_t)o) incl += t;
    }
    if ((threadIdx.x & 63) == 63) sh_wave[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint32_t before = 0, total = 0;
#pragma unroll
    for (uint32_t k = 0; k < 4; k++) {
        if (k < (threadIdx.x >> 6)) before += sh_wave[k];
        total += sh_wave[k];
    }
    if (blk < im.n_units) bits[im.coef_off + blk] = before + incl - n;
    if (threadIdx.x == 0) wg_bits[blockIdx.x] = total;
}

// optimizeCoding: GatherBlockStatistics (:552-597) for the 256 blocks of a workgroup: LDS histograms of the four tables
// (DC0, AC0, DC1, AC1), merged into the image's counters.
// Round 6: the block is read as E2 reads it -- eight 16-byte loads, the symbol walk two coefficients at a time over registers named
// at compile time (enc_block_symbols' walk with a counting `put`) -- where a loop of 63 two-byte loads per lane walked it before
// (1.9 ms per 8192 x 8192 canvas against E1's 0.09: the whole cost of optimizeCoding), and every WAVE counts into a histogram of
// its own (most symbols of most blocks are the same handful: 256 lanes on one LDS word).
__global__ __launch_bounds__(256) void block_stats_kernel(const DevEncImage *__restrict__ images, const EncWork *__restrict__ work,
                                                          const int16_t *__restrict__ coefs, uint32_t *__restrict__ hist) {
    __shared__ uint32_t lh[4][4 * 256];  // [wave][table][symbol]
    const EncWork wk = work[blockIdx.x];
    const DevEncImage &im = images[wk.image];
    for (uint32_t i = threadIdx.x; i < 4u * 4u * 256u; i += 256u) (&lh[0][0])[i] = 0;
    __syncthreads();
    const uint32_t blk = wk.first + threadIdx.x;
    if (im.table_base != 0 && blk < im.total_blocks) {
        const int16_t *img_coefs = coefs + im.coef_off * 64;
        const uint32_t mcu = blk / im.bpm, b = blk - mcu * im.bpm;
        uint32_t comp;
        const int32_t pred = enc_dc_predictor(im, img_coefs, mcu, b, comp);
        const uint4 *src = reinterpret_cast<const uint4 *>(img_coefs + (size_t)enc_source_block(im, blk) * 64);
        const uint4 cv[8] = {src[0], src[1], src[2], src[3], src[4], src[5], src[6], src[7]};
        uint32_t *mine = lh[threadIdx.x >> 6];
        uint32_t *hdc = mine + (comp == 0 ? 0 : 2) * 256, *hac = mine + (comp == 0 ? 1 : 3) * 256;
        const uint32_t w[32] = {cv[0].x, cv[0].y, cv[0].z, cv[0].w, cv[1].x, cv[1].y, cv[1].z, cv[1].w, cv[2].x, cv[2].y, cv[2].z,
                                cv[2].w, cv[3].x, cv[3].y, cv[3].z, cv[3].w, cv[4].x, cv[4].y, cv[4].z, cv[4].w, cv[5].x, cv[5].y,
                                cv[5].z, cv[5].w, cv[6].x, cv[6].y, cv[6].z, cv[6].w, cv[7].x, cv[7].y, cv[7].z, cv[7].w};
        auto symbol = [](uint32_t run, int32_t value) -> uint32_t {  // GatherRunLengthCodeStatistics (:872-891)
            const uint32_t a = (uint32_t)(value < 0 ? -value : value);
            return ((run << 4) | enc_bit_count(a & 0xFFFFu)) & 0xFFu;
        };
        atomicAdd(&hdc[symbol(0, (int32_t)(int16_t)(w[0] & 0xFFFFu) - pred)], 1u);
        uint32_t run = 0;
        auto ac_coefficient = [&](int32_t t) {
            if (t == 0) {
                run++;
            } else {
                while (run > 15) {
                    atomicAdd(&hac[0xF0], 1u);
                    run -= 16;
                }
                atomicAdd(&hac[symbol(run, t)], 1u);
                run = 0;
            }
        };
        ac_coefficient((int32_t)w[0] >> 16);
#pragma unroll
        for (int k = 1; k < 32; k++) {
            if (w[k] == 0) {
                run += 2;
            } else {
                ac_coefficient((int32_t)(int16_t)(w[k] & 0xFFFFu));
                ac_coefficient((int32_t)w[k] >> 16);
            }
        }
        if (run > 0) atomicAdd(&hac[0], 1u);
    }
    __syncthreads();
    uint32_t *gh = hist + (size_t)wk.image * 4 * 256;
    for (uint32_t i = threadIdx.x; i < 4u * 256u; i += 256u) {
        const uint32_t v = lh[0][i] + lh[1][i] + lh[2][i] + lh[3][i];
        if (v != 0) atomicAdd(&gh[i], v);
    }
}

// Exclusive prefix sums of the workgroup bit totals of one image (one workgroup per image); total -> images' raw_bits.
__global__ __launch_bounds__(1024) void block_offsets_kernel(const DevEncImage *__restrict__ images, const uint32_t *__restrict__ wg_bits,
                                                             uint64_t *__restrict__ wg_base, uint64_t *__restrict__ raw_bits) {
    const DevEncImage &im = images[blockIdx.x];
    __shared__ uint64_t sh[1024];
    const uint32_t tid = threadIdx.x;
    const uint32_t n = (im.n_units + 255) / 256;  // workgroups of block_bits_kernel / emit_kernel for this image
    const uint32_t per = (n + 1023) / 1024;
    const uint32_t lo = tid * per < n ? tid * per : n, hi = lo + per < n ? lo + per : n;
    const uint32_t *src = wg_bits + im.work_first;
    uint64_t sum = 0;
    for (uint32_t i = lo; i < hi; i++) sum += src[i];
    sh[tid] = sum;
    __syncthreads();
    for (uint32_t o = 1; o < 1024; o <<= 1) {
        const uint64_t v = tid >= o ? sh[tid - o] : 0;
        __syncthreads();
        sh[tid] += v;
        __syncthreads();
    }
    uint64_t run = sh[tid] - sum;
    for (uint32_t i = lo; i < hi; i++) {
        wg_base[im.work_first + i] = run;
        run += src[i];
    }
    if (tid == 1023) raw_bits[blockIdx.x] = sh[1023];
}

// One lane per block: its codes ORed into the raw bit stream (MSB-first) at its bit offset.  The stream is kept as
// big-endian 32-bit words, i.e. every word is stored byte-swapped.
// Round 3: the 256 blocks (or restart intervals) of a workgroup own ONE contiguous stretch of the stream -- from the
// workgroup's base to the next one's -- so the lanes assemble it in LDS (ds_or on words the workgroup zeroed; ~3 per block)
// and the workgroup writes it out with plain coalesced stores; only its first and last word can be shared with a
// neighbouring workgroup and go out as atomicOr on the zeroed buffer.  Before, every completed word of every lane was an
// atomicOr in HBM (150 M of them per 256 x 4K: emit_kernel 5.05 ms against 2.2 ms for the same walk in block_bits_kernel).
// A stretch longer than the LDS buffer falls back to the global atomics.  The buffer is sized per launch (lds_words, dynamic
// LDS): the host has the images' bit totals by now and asks for twice the batch's largest AVERAGE stretch, between 8 and
// 32 KB -- with the 32 KB it used to hold always, four workgroups fitted a CU and the kernel ran at 4 waves per SIMD:
// 2.74 ms per 256 x 4K at Q75, 2.04 ms with 16 KB, 1.94 with the 8 KB the rule picks there (the walk alone, block_bits_kernel: 1.6).
__global__ __launch_bounds__(256) void emit_kernel(const DevEncImage *__restrict__ images, const EncWork *__restrict__ work,
                                                   const EncHuffTable *__restrict__ tables, const int16_t *__restrict__ coefs,
                                                   const uint32_t *__restrict__ bits, const uint64_t *__restrict__ wg_base,
                                                   const uint64_t *__restrict__ raw_bits, uint8_t *__restrict__ raw,
                                                   uint32_t *__restrict__ marks, uint32_t lds_words) {
    extern __shared__ uint32_t sh_words[];
    __shared__ EncHuffTable sh_tab[4];
    const EncWork wk = work[blockIdx.x];
    const DevEncImage &im = images[wk.image];
    enc_stage_tables(tables, im.table_base, sh_tab);
    const uint32_t blk = wk.first + threadIdx.x;  // block, or restart interval
    const bool active = blk < im.n_units;
    const int16_t *img_coefs = coefs + im.coef_off * 64;
    uint32_t *words = reinterpret_cast<uint32_t *>(raw + im.raw_off);
    // the workgroup's stretch of the stream, in whole words
    const uint64_t wg_start = wg_base[blockIdx.x];
    const bool last_wg = wk.first + 256u >= im.n_units;
    const uint64_t wg_end = last_wg ? ((raw_bits[wk.image] + 7ull) & ~7ull) : wg_base[blockIdx.x + 1];
    const uint64_t first_word = wg_start >> 5;
    const uint64_t span_bits = wg_end - (first_word << 5);
    const bool use_lds = span_bits <= (uint64_t)lds_words * 32u;  // the same for every lane
    const uint32_t n_words = use_lds ? (uint32_t)((span_bits + 31u) >> 5) : 0u;
    for (uint32_t w = threadIdx.x; w < n_words; w += 256u) sh_words[w] = 0;
    __syncthreads();
    const uint64_t start = wg_start + (active ? bits[im.coef_off + blk] : 0u);
    uint32_t wi = (uint32_t)((start >> 5) - first_word);  // current word, counted from the workgroup's first
    uint32_t fill = (uint32_t)(start & 31);               // bits already taken in it (by earlier blocks): always < 32 between calls
    uint32_t cur = 0;                                     // the current word, MSB first
    auto flush_word = [&](uint32_t w) {
        if (use_lds) atomicOr(&sh_words[wi], w);
        else atomicOr(&words[first_word + wi], __builtin_bswap32(w));  // (plain stores for the words in between: measured slower)
    };
    auto flush = [&]() { flush_word(cur); };
    auto put = [&](uint32_t code, uint32_t len) {  // len <= 32 (31 from enc_block_symbols), code < 2^len
        if (len == 0) return;  // a symbol the table has no code for (and the reference writes nothing for)
        const uint32_t left = code << (32u - len);  // the field with its first bit at bit 31
        cur |= left >> fill;
        const uint32_t before = fill;
        fill += len;
        if (fill >= 32u) {
            flush();
            wi++;
            cur = __builtin_amdgcn_alignbit(left, 0u, before);  // what did not fit: left << (32 - before), 0 when before == 0
            fill -= 32u;
        }
    };
    if (active && im.restart_interval != 0) {
        // one lane per restart interval: its blocks, one-bit padding to the byte boundary (ExitBitMode), and a mark on its
        // last byte when a restart marker follows (the stuffing pass writes RSTm there)
        const uint32_t first_mcu = blk * im.restart_interval, total_mcus = im.mcus_per_line * im.mcus_per_column;
        const uint32_t end_mcu = first_mcu + im.restart_interval < total_mcus ? first_mcu + im.restart_interval : total_mcus;
        uint32_t nbits = 0;
        auto put_counted = [&](uint32_t code, uint32_t len) {
            nbits += len;
            put(code, len);
        };
        for (uint32_t mcu = first_mcu; mcu < end_mcu; mcu++)
            for (uint32_t b = 0; b < im.bpm; b++) {
                uint32_t comp;
                const int32_t pred = enc_dc_predictor(im, img_coefs, mcu, b, comp);
                const uint4 *src = reinterpret_cast<const uint4 *>(img_coefs + (size_t)enc_source_block(im, mcu * im.bpm + b) * 64);
                const uint4 cv[8] = {src[0], src[1], src[2], src[3], src[4], src[5], src[6], src[7]};
                enc_block_symbols(cv, pred, sh_tab[comp == 0 ? 0 : 2], sh_tab[comp == 0 ? 1 : 3], put_counted);
            }
        const uint32_t rem = (8u - (nbits & 7u)) & 7u;
        if (rem) put((1u << rem) - 1u, rem);
        if (fill) flush();
        if (blk + 1 < im.n_units) {
            const uint64_t last_byte = (start >> 3) + ((nbits + 7u) >> 3) - 1u;
            atomicOr(&marks[(im.raw_off >> 5) + (last_byte >> 5)], 1u << (uint32_t)(last_byte & 31u));
        }
    } else if (active) {
        const uint32_t mcu = blk / im.bpm, b = blk - mcu * im.bpm;
        uint32_t comp;
        const int32_t pred = enc_dc_predictor(im, img_coefs, mcu, b, comp);
        const uint4 *src = reinterpret_cast<const uint4 *>(img_coefs + (size_t)enc_source_block(im, blk) * 64);
        const uint4 cv[8] = {src[0], src[1], src[2], src[3], src[4], src[5], src[6], src[7]};
        enc_block_symbols(cv, pred, sh_tab[comp == 0 ? 0 : 2], sh_tab[comp == 0 ? 1 : 3], put);
        if (blk == im.total_blocks - 1) {
            // ExitBitMode (ref: JpegWriter.cs:123-147): pad the last byte with one-bits
            const uint32_t rem = (uint32_t)((8u - (raw_bits[wk.image] & 7u)) & 7u);
            if (rem) put((1u << rem) - 1u, rem);
        }
        if (fill) flush();
    }
    if (n_words != 0) {
        __syncthreads();
        for (uint32_t w = threadIdx.x; w < n_words; w += 256u) {
            const uint32_t v = __builtin_bswap32(sh_words[w]);
            if (w == 0 || w + 1 == n_words) {
                if (v) atomicOr(&words[first_word + w], v);  // may be shared with the neighbouring workgroup's stretch
            } else {
                words[first_word + w] = v;
            }
        }
    }
}

// E2 + E3 as ONE pass over the blocks (round 5; restart-free images -- everything the reference's encoder can write): a workgroup
// counts the bits of its 256 blocks, assembles its stretch of the stream in LDS at offsets RELATIVE to its own first bit (nothing
// but the workgroup's own scan is needed for that), learns its base from its predecessors through a decoupled look-back and
// writes the stretch out shifted by the base's bit phase -- one read of the quantised blocks instead of two, no per-block offsets
// in HBM, no image-wide scan kernel, and no zeroed raw buffer: every stream word has ONE writer (a word that straddles two
// workgroups is written by the later one, which takes the earlier one's last bits from the chain record), so the raw streams can
// be placed before their sizes are known (a worst-case slot per image that is never touched beyond the stream's end).
//   chain[t] = {head, tail}: head = flag << 62 | bits (1: this workgroup's own total; 2: the image's total up to and including
//   it), tail = the last 32 bits of its stretch.  A workgroup with a successor holds >= 256 blocks x >= 2 bits, i.e. its own tail
//   IS the stream's tail in front of the successor.  Tickets (ctl[0]) give the workgroups their place in the work list in the
//   order they START, so whatever a workgroup waits for belongs to a workgroup that is running or done; the wait is bounded and
//   a stretch that does not fit the LDS buffer is not emitted: either sets ctl[1] and the host issues the two-kernel path.
struct EncChain {
    unsigned long long head, tail;
};
constexpr unsigned long long kChainFlagShift = 62, kChainValueMask = (1ull << 62) - 1ull;
__global__ __launch_bounds__(256) void bits_emit_kernel(const DevEncImage *__restrict__ images, const EncWork *__restrict__ work,
                                                        const EncHuffTable *__restrict__ tables, const int16_t *__restrict__ coefs,
                                                        EncChain *__restrict__ chain, uint32_t *__restrict__ ctl, uint8_t *__restrict__ raw,
                                                        uint64_t *__restrict__ raw_bits, uint32_t lds_words, const uint32_t *__restrict__ order) {
    extern __shared__ uint32_t sh_words[];
    __shared__ EncHuffTable sh_tab[4];
    __shared__ uint32_t sh_wave[4], sh_prev_tail;
    __shared__ unsigned long long sh_base;
    const uint32_t tid = threadIdx.x;
#if defined(JPGPU_ENC_TICKETS)
    // (round 5: a ticket -- the place in the list in the order the workgroups START, whatever the dispatcher does)
    __shared__ uint32_t sh_ticket;
    if (tid == 0) sh_ticket = atomicAdd(&ctl[0], 1u);
    __syncthreads();
    const uint32_t place = sh_ticket;
#else
    // Round 6, as in K1 (k1_markers.hip, marker_onepass_kernel): the workgroup's own index.  Workgroups are handed out in index order,
    // so the lowest unfinished one is always resident; the wait stays bounded and ctl[1] still sends the batch to the two-kernel
    // path if it ever ran out -- slow then, never wrong.  Saves a device-scope atomic on one address and a barrier per workgroup.
    const uint32_t place = blockIdx.x;
#endif
    // order[]: the work list's entries by (place inside the image, image) -- the workgroups that run side by side then belong to
    // as many images as the batch has, and an image's chain is a handful of records long at any time.  (In work-list order all
    // ~1 500 resident workgroups sat in ONE image's chain and each walked back over all of them: 24 ms per 256 x 4K.)  An image's
    // workgroups still start in the image's order, and their records (indexed by work-list position) are consecutive.
    const uint32_t t = order[place];
    const EncWork wk = work[t];
    const DevEncImage &im = images[wk.image];
    enc_stage_tables(tables, im.table_base, sh_tab);
    const uint32_t blk = wk.first + tid;
    const bool active = blk < im.total_blocks;
    const int16_t *img_coefs = coefs + im.coef_off * 64;
    uint4 cv[8];
    int32_t pred = 0;
    uint32_t comp = 0, n = 0;
    if (active) {
        const uint32_t mcu = blk / im.bpm, b = blk - mcu * im.bpm;
        pred = enc_dc_predictor(im, img_coefs, mcu, b, comp);
        const uint4 *src = reinterpret_cast<const uint4 *>(img_coefs + (size_t)enc_source_block(im, blk) * 64);
#pragma unroll
        for (int i = 0; i < 8; i++) cv[i] = src[i];
        enc_block_symbols(cv, pred, sh_tab[comp == 0 ? 0 : 2], sh_tab[comp == 0 ? 1 : 3], [&](uint32_t, uint32_t len) { n += len; });
    }
    // the block's offset inside the workgroup's stretch, the stretch's length
    uint32_t incl = n;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t v = __shfl_up(incl, o, 64);
        if ((tid & 63) >= (uint32_t)o) incl += v;
    }
    if ((tid & 63) == 63) sh_wave[tid >> 6] = incl;
    __syncthreads();
    uint32_t before = 0, total = 0;
#pragma unroll
    for (uint32_t k = 0; k < 4; k++) {
        if (k < (tid >> 6)) before += sh_wave[k];
        total += sh_wave[k];
    }
    const uint32_t n_rel_words = (total + 31u) >> 5;
    const bool fits = n_rel_words + 1u <= lds_words;  // (one word of slack: the shifted read of the last, partial word)
    if (fits)
        for (uint32_t w = tid; w <= n_rel_words; w += 256u) sh_words[w] = 0;
    __syncthreads();
    if (fits && active) {
        const uint32_t start = before + incl - n;
        uint32_t wi = start >> 5, fill = start & 31u, cur = 0;
        auto put = [&](uint32_t code, uint32_t len) {  // as emit_kernel's
            if (len == 0) return;
            const uint32_t left = code << (32u - len);
            cur |= left >> fill;
            const uint32_t was = fill;
            fill += len;
            if (fill >= 32u) {
                atomicOr(&sh_words[wi], cur);
                wi++;
                cur = __builtin_amdgcn_alignbit(left, 0u, was);
                fill -= 32u;
            }
        };
        enc_block_symbols(cv, pred, sh_tab[comp == 0 ? 0 : 2], sh_tab[comp == 0 ? 1 : 3], put);
        if (fill) atomicOr(&sh_words[wi], cur);
    }
    __syncthreads();
    const bool first_wg = wk.first == 0, last_wg = wk.first + 256u >= im.total_blocks;
    if (tid < 64u) {  // wave 0: publish, look back 64 records at a time, publish again
        const uint32_t lane = tid;
        if (lane == 0) {
            uint32_t tail = 0;
            if (fits && total != 0) {
                const uint32_t w = (total - 1u) >> 5, r = total & 31u;
                tail = r == 0 ? sh_words[w] : ((w != 0 ? sh_words[w - 1u] << r : 0u) | (sh_words[w] >> (32u - r)));
            }
            if (!fits) atomicOr(&ctl[1], 1u);
            // Relaxed device-scope atomics throughout: they are performed where all XCDs see them, and NOTHING else is handed from
            // one workgroup to another -- a release / acquire pair would write back / invalidate the whole L2 of the XCD (every
            // other workgroup's stream words) twice per workgroup and once per poll.  The tail is in memory before the head says so.
            __hip_atomic_store(&chain[t].tail, (unsigned long long)tail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (!first_wg) __hip_atomic_store(&chain[t].head, (1ull << kChainFlagShift) | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        unsigned long long base = 0;
        if (!first_wg) {
            // the image's workgroups hold consecutive tickets' records (the work list is in image order): the first one's is the
            // lowest this walk may read, and it is an inclusive one
            const int64_t lowest = (int64_t)t - (int64_t)(wk.first >> 8);
            int64_t pos = (int64_t)t - 1;
            uint32_t polls = 0;
            for (;;) {
                const int64_t j = pos - (int64_t)lane;
                const bool valid = j >= lowest;
                unsigned long long v = 0;
                if (valid) v = __hip_atomic_load(&chain[j].head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const uint32_t flag = (uint32_t)(v >> kChainFlagShift);
                const uint64_t inclusive = __ballot(valid && flag == 2u), empty = __ballot(valid && flag == 0u);
                uint64_t take;  // the records this step adds up: up to the nearest inclusive one, or all 64
                bool done;
                if (inclusive != 0) {
                    const uint32_t k = (uint32_t)__builtin_ctzll(inclusive);
                    take = k == 63u ? ~0ull : ((2ull << k) - 1ull);
                    done = true;
                } else {
                    take = ~0ull;
                    done = false;
                }
                if ((empty & take) != 0) {  // a record the sum needs has not been published yet
                    if (++polls > (1u << 18)) {  // ~a second: something is wrong; the host issues the two-kernel path
                        if (lane == 0) atomicOr(&ctl[1], 2u);
                        break;
                    }
                    __builtin_amdgcn_s_sleep(2);
                    continue;
                }
                unsigned long long part = (valid && ((take >> lane) & 1ull)) ? (v & kChainValueMask) : 0ull;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
                base += part;
                if (done) break;
                pos -= 64;
            }
        }
        if (lane == 0) {
            uint32_t prev_tail = 0;
            asm volatile("" ::: "memory");  // (behind the walk that has seen record t - 1 published: its tail was in memory before its head)
            if (!first_wg) prev_tail = (uint32_t)__hip_atomic_load(&chain[t - 1u].tail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&chain[t].head, (2ull << kChainFlagShift) | (base + total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            sh_base = base;
            sh_prev_tail = prev_tail;
            if (last_wg) raw_bits[wk.image] = base + total;
        }
    }
    __syncthreads();
    if (!fits) return;
    const unsigned long long base = sh_base, end = base + total;
    const uint32_t p = (uint32_t)(base & 31ull), prev_tail = sh_prev_tail;
    uint32_t *words = reinterpret_cast<uint32_t *>(raw + im.raw_off) + (base >> 5);
    auto abs_word = [&](uint32_t w) -> uint32_t {  // stream word (base >> 5) + w out of the relative stretch
        if (p == 0) return sh_words[w];
        const uint32_t hi = (w == 0 ? prev_tail : sh_words[w - 1u]) << (32u - p);
        return hi | (sh_words[w] >> p);
    };
    const uint32_t n_full = (uint32_t)((end >> 5) - (base >> 5));  // words this workgroup completes
    for (uint32_t w = tid; w < n_full; w += 256u) words[w] = __builtin_bswap32(abs_word(w));
    if (last_wg && tid == 0) {
        // the stream's last, partial word: the bits left over, ExitBitMode's one-bits up to the byte boundary (ref: JpegWriter.cs:123-147)
        const uint32_t r = (uint32_t)(end & 31ull), rem = (8u - (uint32_t)(end & 7ull)) & 7u;
        if (r != 0) {
            uint32_t a = abs_word(n_full);
            if (rem) a |= ((1u << rem) - 1u) << (32u - r - rem);
            words[n_full] = __builtin_bswap32(a);
        }
    }
}

// ------------------------------------------------------------------------------------------------ E4: byte stuffing

constexpr uint32_t kStuffChunk = 4096;  // raw bytes per workgroup (256 lanes x 16)

// The 16 mark bits of the raw bytes [first, first + 16) of an image (first is a multiple of 16): a restart marker follows
// the marked byte.
__device__ __forceinline__ uint32_t enc_marks16(const DevEncImage &im, const uint32_t *__restrict__ marks, uint64_t first) {
    if (im.restart_interval == 0) return 0;
    return (marks[(im.raw_off >> 5) + (first >> 5)] >> (uint32_t)(first & 31u)) & 0xFFFFu;
}

// chunk_ff: FF bytes of the chunk (the bytes stuffing adds) in the low half, marked bytes (two marker bytes each) in the high half
__global__ __launch_bounds__(256) void stuff_count_kernel(const DevEncImage *__restrict__ images, const EncWork *__restrict__ work,
                                                          const uint64_t *__restrict__ raw_bits, const uint8_t *__restrict__ raw,
                                                          const uint32_t *__restrict__ marks, uint32_t *__restrict__ chunk_ff) {
    const EncWork wk = work[blockIdx.x];
    const DevEncImage &im = images[wk.image];
    const uint64_t raw_len = (raw_bits[wk.image] + 7) >> 3;
    const uint64_t first = (uint64_t)wk.first * kStuffChunk + (uint64_t)threadIdx.x * 16;
    uint32_t n = 0;
    if (first < raw_len) {
        const uint4 v = *reinterpret_cast<const uint4 *>(raw + im.raw_off + first);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 16; j++)
            if (first + j < raw_len && ((w[j >> 2] >> ((j & 3) * 8)) & 0xFFu) == 0xFFu) n++;
        n += (uint32_t)__builtin_popcount(enc_marks16(im, marks, first)) << 16;
    }
    __shared__ uint32_t sh;
    if (threadIdx.x == 0) sh = 0;
    __syncthreads();
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) n += __shfl_xor(n, o, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(&sh, n);
    __syncthreads();
    if (threadIdx.x == 0) chunk_ff[im.chunk_off + wk.first] = sh;
}

__global__ __launch_bounds__(256) void stuff_write_kernel(const DevEncImage *__restrict__ images, const EncWork *__restrict__ work,
                                                          const uint64_t *__restrict__ raw_bits, const uint8_t *__restrict__ raw,
                                                          const uint32_t *__restrict__ marks, const uint32_t *__restrict__ chunk_ff,
                                                          uint8_t *__restrict__ out, uint64_t *__restrict__ out_len) {
    const EncWork wk = work[blockIdx.x];
    const DevEncImage &im = images[wk.image];
    const uint64_t raw_len = (raw_bits[wk.image] + 7) >> 3;
    const uint32_t tid = threadIdx.x;
    // FF bytes and marked bytes in the chunks before this one (an image's chunks: a few hundred at most per workgroup)
    __shared__ uint32_t sh_ff, sh_mk, sh_wave[4];
    if (tid == 0) sh_ff = 0, sh_mk = 0;
    __syncthreads();
    {
        uint32_t sf = 0, sm = 0;
        for (uint32_t i = tid; i < wk.first; i += 256) {
            const uint32_t c = chunk_ff[im.chunk_off + i];
            sf += c & 0xFFFFu;
            sm += c >> 16;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            sf += __shfl_xor(sf, o, 64);
            sm += __shfl_xor(sm, o, 64);
        }
        if ((tid & 63) == 0 && sf) atomicAdd(&sh_ff, sf);
        if ((tid & 63) == 0 && sm) atomicAdd(&sh_mk, sm);
    }
    __syncthreads();
    const uint64_t first = (uint64_t)wk.first * kStuffChunk + (uint64_t)tid * 16;
    uint32_t w[4] = {0, 0, 0, 0}, mine = 0, valid = 0, mk = 0;
    if (first < raw_len) {
        const uint4 v = *reinterpret_cast<const uint4 *>(raw + im.raw_off + first);
        w[0] = v.x;
        w[1] = v.y;
        w[2] = v.z;
        w[3] = v.w;
        valid = raw_len - first < 16 ? (uint32_t)(raw_len - first) : 16u;
        for (uint32_t j = 0; j < valid; j++)
            if (((w[j >> 2] >> ((j & 3) * 8)) & 0xFFu) == 0xFFu) mine++;
        mk = enc_marks16(im, marks, first);
        mine |= (uint32_t)__builtin_popcount(mk) << 16;
    }
    // exclusive scan of `mine` (both halves at once: a chunk holds at most 4096 of either) over the workgroup
    uint32_t incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(incl, o, 64);
        if ((tid & 63) >= (uint32_t)o) incl += t;
    }
    if ((tid & 63) == 63) sh_wave[tid >> 6] = incl;
    __syncthreads();
    uint32_t in_chunk = incl - mine;
    for (uint32_t k = 0; k < (tid >> 6); k++) in_chunk += sh_wave[k];
    const uint32_t before = sh_ff + (in_chunk & 0xFFFFu);  // bytes stuffing added in front of this lane
    uint32_t markers = sh_mk + (in_chunk >> 16);            // restart markers written in front of this lane
    uint8_t *dst = out + im.out_off + im.header_len + first + before + 2ull * markers;
    for (uint32_t j = 0; j < valid; j++) {
        const uint8_t bch = (uint8_t)((w[j >> 2] >> ((j & 3) * 8)) & 0xFFu);
        *dst++ = bch;
        if (bch == 0xFF) *dst++ = 0;
        if ((mk >> j) & 1u) {  // the interval ends here: RSTm, m counting modulo 8 (never stuffed)
            *dst++ = 0xFF;
            *dst++ = (uint8_t)(0xD0u + (markers & 7u));
            markers++;
        }
    }
    // the lane holding the last raw byte closes the stream: EOI (WriteEndOfImage :930-933) and the total length
    if (valid && first + valid == raw_len) {
        dst[0] = 0xFF;
        dst[1] = 0xD9;
        out_len[wk.image] = im.header_len + raw_len + before + (mine & 0xFFFFu) + 2ull * markers + 2;
    }
    if (raw_len == 0 && wk.first == 0 && tid == 0) {
        uint8_t *d0 = out + im.out_off + im.header_len;
        d0[0] = 0xFF;
        d0[1] = 0xD9;
        out_len[wk.image] = im.header_len + 2;
    }
}

// SOI .. SOS of every image in front of its stream (where that is depends on the sizes block_bits_kernel found: one upload of
// all headers + this kernel instead of one small copy per image, which cost ~5 us each on the stream)
__global__ __launch_bounds__(256) void place_headers_kernel(const DevEncImage *__restrict__ images, const uint8_t *__restrict__ headers,
                                                            uint8_t *__restrict__ out) {
    const DevEncImage &im = images[blockIdx.x];
    const uint8_t *src = headers + im.hdr_off;
    uint8_t *dst = out + im.out_off;
    for (uint32_t j = threadIdx.x; j < im.header_len; j += 256) dst[j] = src[j];
}

// ------------------------------------------------------------------------------------------------ launch wrappers

size_t enc_sample_bytes_per_mcu(uint32_t luma_h, uint32_t luma_v, uint32_t components) { return enc_sample_stride(luma_h, luma_v, components); }

int enc_image_fused_shape(const DevEncImage &im) { return enc_fused_shape(im); }

hipError_t launch_fdct_quant(hipStream_t stream, const uint8_t *pixels, const DevEncImage *images, const EncWork *work, int n_work,
                             uint8_t *samples, int16_t *coefs, size_t max_record_bytes, uint32_t fused_shapes, bool any_other) {
    if (n_work <= 0) return hipSuccess;
    const dim3 grid(n_work * (kEncMcusPerWg / kEfMcus));
    if (fused_shapes & 2u) hipLaunchKernelGGL((fdct_fused_kernel<2, 2, 3>), grid, dim3(64), 0, stream, pixels, images, work, coefs);
    if (fused_shapes & 4u) hipLaunchKernelGGL((fdct_fused_kernel<2, 1, 3>), grid, dim3(64), 0, stream, pixels, images, work, coefs);
    if (fused_shapes & 8u) hipLaunchKernelGGL((fdct_fused_kernel<1, 1, 3>), grid, dim3(64), 0, stream, pixels, images, work, coefs);
    if (fused_shapes & 16u) hipLaunchKernelGGL((fdct_fused_kernel<2, 2, 4>), grid, dim3(64), 0, stream, pixels, images, work, coefs);
    if (fused_shapes & 32u) hipLaunchKernelGGL((fdct_fused_kernel<2, 1, 4>), grid, dim3(64), 0, stream, pixels, images, work, coefs);
    if (fused_shapes & 64u) hipLaunchKernelGGL((fdct_fused_kernel<1, 1, 4>), grid, dim3(64), 0, stream, pixels, images, work, coefs);
    if (any_other) {
        const bool skip = fused_shapes != 0;
        const size_t lds = max_record_bytes * (size_t)kEncMcusPerWg;
        if (lds <= 64 * 1024)
            hipLaunchKernelGGL(enc_gather_kernel<true>, dim3(n_work), dim3(8 * kEncMcusPerWg), lds, stream, pixels, images, work, samples, skip);
        else hipLaunchKernelGGL(enc_gather_kernel<false>, dim3(n_work), dim3(8 * kEncMcusPerWg), 0, stream, pixels, images, work, samples, skip);
        hipLaunchKernelGGL(fdct_quant_kernel, dim3(n_work), dim3(kEncMcusPerWg), 0, stream, samples, images, work, coefs, skip);
    }
    return hipGetLastError();
}
hipError_t launch_block_bits(hipStream_t stream, const DevEncImage *images, const EncWork *work, int n_work, const EncHuffTable *tables,
                             const int16_t *coefs, uint32_t *bits, int n_images, uint32_t *wg_bits, uint64_t *wg_base, uint64_t *raw_bits) {
    if (n_work <= 0) return hipSuccess;
    hipLaunchKernelGGL(block_bits_kernel, dim3(n_work), dim3(256), 0, stream, images, work, tables, coefs, bits, wg_bits);
    hipLaunchKernelGGL(block_offsets_kernel, dim3(n_images), dim3(1024), 0, stream, images, wg_bits, wg_base, raw_bits);
    return hipGetLastError();
}
hipError_t launch_block_stats(hipStream_t stream, const DevEncImage *images, const EncWork *work, int n_work, const int16_t *coefs,
                              uint32_t *hist) {
    if (n_work <= 0) return hipSuccess;
    hipLaunchKernelGGL(block_stats_kernel, dim3(n_work), dim3(256), 0, stream, images, work, coefs, hist);
    return hipGetLastError();
}
hipError_t launch_emit(hipStream_t stream, const DevEncImage *images, const EncWork *work, int n_work, const EncHuffTable *tables,
                       const int16_t *coefs, const uint32_t *bits, const uint64_t *wg_base, const uint64_t *raw_bits, uint8_t *raw,
                       uint32_t *marks, uint32_t lds_words) {
    if (n_work <= 0) return hipSuccess;
    if (lds_words != 0) lds_words = std::min(std::max(lds_words, kEmitLdsWordsMin), kEmitLdsWordsMax);  // 0: no workgroup's stretch would fit
    hipLaunchKernelGGL(emit_kernel, dim3(n_work), dim3(256), (size_t)lds_words * 4, stream, images, work, tables, coefs, bits, wg_base, raw_bits, raw,
                       marks, lds_words);
    return hipGetLastError();
}
// E2 + E3 in one pass (restart-free batches): chain = n_work x 16 bytes and ctl = 2 words, both zeroed by the caller; ctl[1] != 0
// afterwards = not every workgroup emitted (a stretch beyond lds_words, a wait that ran out): issue launch_block_bits + launch_emit
size_t enc_chain_bytes(int n_work) { return (size_t)n_work * sizeof(EncChain); }
hipError_t launch_bits_emit(hipStream_t stream, const DevEncImage *images, const EncWork *work, int n_work, const EncHuffTable *tables,
                            const int16_t *coefs, void *chain, uint32_t *ctl, uint8_t *raw, uint64_t *raw_bits, uint32_t lds_words, const uint32_t *order) {
    if (n_work <= 0) return hipSuccess;
    lds_words = std::min(std::max(lds_words, kEmitLdsWordsMin), kEmitLdsWordsMax);
    hipLaunchKernelGGL(bits_emit_kernel, dim3(n_work), dim3(256), (size_t)lds_words * 4, stream, images, work, tables, coefs, (EncChain *)chain, ctl, raw,
                       raw_bits, lds_words, order);
    return hipGetLastError();
}
hipError_t launch_place_headers(hipStream_t stream, const DevEncImage *images, int n_images, const uint8_t *headers, uint8_t *out) {
    if (n_images <= 0) return hipSuccess;
    hipLaunchKernelGGL(place_headers_kernel, dim3(n_images), dim3(256), 0, stream, images, headers, out);
    return hipGetLastError();
}
hipError_t launch_stuff(hipStream_t stream, const DevEncImage *images, const EncWork *work, int n_work, const uint64_t *raw_bits,
                        const uint8_t *raw, const uint32_t *marks, uint32_t *chunk_ff, uint8_t *out, uint64_t *out_len) {
    if (n_work <= 0) return hipSuccess;
    hipLaunchKernelGGL(stuff_count_kernel, dim3(n_work), dim3(256), 0, stream, images, work, raw_bits, raw, marks, chunk_ff);
    hipLaunchKernelGGL(stuff_write_kernel, dim3(n_work), dim3(256), 0, stream, images, work, raw_bits, raw, marks, chunk_ff, out, out_len);
    return hipGetLastError();
}

}  // namespace jpgpu
