/* oracle/jpegenc.c -- TEST INFRASTRUCTURE ONLY: scalar C restatement of the reference's baseline ENCODER
 * (the step on the other side of the wire format).
 *
 * Follows (all paths relative to /root/reference/src/JpegLibrary unless noted):
 *   JpegEncoder.cs           Encode :255-291, WriteQuantizationTables :305-335, WriteHuffmanTables :336-352,
 *                            WriteStartOfFrame :353-386, WriteStartOfScan :387-413, WriteScanData :662-741,
 *                            ReadBlock / ReadBlockWithSubsample / CopySubsampleBlock :743-799, ShiftDataLevel :801-810,
 *                            ZigZagAndQuantizeBlock :812-826, EncodeBlock :828-870, EncodeRunLength :893-918
 *   FastFloatingPointDCT.cs  TransformFDCT :343-362, FDCT8x4_LeftPart / RightPart :194-311
 *   JpegWriter.cs            WriteBits / FlushRegister / ExitBitMode / WriteMarker / WriteLength
 *   JpegStandardQuantizationTable.cs :12-87, JpegStandardHuffmanEncodingTable.cs :14-131 (BuildCanonicalCode)
 *   apps/JpegEncode/EncodeAction.cs :17-71 (the canonical call sequence), apps/JpegEncode/JpegBufferInputReader.cs :27-52
 *
 * PARITY UNPINNED: the reference's tests hold no encoder golden vectors and the reference cannot run here (no .NET), so
 * this restatement is pinned only by construction and by the round trip through the (golden-pinned) decoder oracle.
 *
 * One behaviour restated on purpose: in WriteScanData the block buffer `inputBuffer` is ONE stack slot reused for every
 * block (an `out` local in the loop body, :712); ReadBlockWithSubsample ACCUMULATES into it (:788-799) without clearing,
 * so a sub-sampled component's block starts from the previous block's quantised zig-zag coefficients (ZigZagAndQuantize
 * wrote them into the same buffer, :721).  Full-resolution blocks overwrite the buffer and are unaffected.  The
 * optimised-table path (TransformBlocks, :414-485) reads into freshly cleared allocator blocks instead; it is not
 * restated here.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "jpegref.h"

/* ---------------------------------------------------------------------------------------------- tables */

/* ref: JpegStandardQuantizationTable.cs:12-34 (zig-zag order) */
static const uint16_t k_std_lum[64] = {16, 11, 12, 14, 12, 10, 16, 14, 13, 14, 18, 17, 16, 19, 24, 40, 26, 24, 22, 22, 24, 49,
                                       35, 37, 29, 40, 58, 51, 61, 60, 57, 51, 56, 55, 64, 72, 92, 78, 64, 68, 87, 69, 55, 56,
                                       80, 109, 81, 87, 95, 98, 103, 104, 103, 62, 77, 113, 121, 112, 100, 120, 92, 101, 103, 99};
static const uint16_t k_std_chr[64] = {17, 18, 18, 24, 21, 24, 47, 26, 26, 47, 99, 66, 56, 66, 99, 99, 99, 99, 99, 99, 99, 99,
                                       99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99,
                                       99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99};

/* ref: JpegStandardQuantizationTable.cs:64-87 */
void jref_scale_quant_table(const uint16_t *src, int quality, uint16_t *dst) {
    int scale = quality < 50 ? 5000 / quality : 200 - (quality * 2);
    for (int i = 0; i < 64; i++) {
        int x = src[i];
        x = ((x * scale) + 50) / 100;
        dst[i] = (uint16_t)(x < 1 ? 1 : (x > 255 ? 255 : x));
    }
}

/* ref: JpegStandardHuffmanEncodingTable.cs:14-83 */
static const uint8_t k_dc_lum_len[16] = {0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0};
static const uint8_t k_dc_chr_len[16] = {0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0};
static const uint8_t k_dc_val[12] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11};
static const uint8_t k_ac_lum_len[16] = {0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 125};
static const uint8_t k_ac_lum_val[162] = {
    0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71, 0x14, 0x32, 0x81,
    0x91, 0xa1, 0x08, 0x23, 0x42, 0xb1, 0xc1, 0x15, 0x52, 0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72, 0x82, 0x09, 0x0a, 0x16, 0x17, 0x18,
    0x19, 0x1a, 0x25, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x34, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48,
    0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75,
    0x76, 0x77, 0x78, 0x79, 0x7a, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99,
    0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3,
    0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2, 0xe3, 0xe4, 0xe5,
    0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};
static const uint8_t k_ac_chr_len[16] = {0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 119};
static const uint8_t k_ac_chr_val[162] = {
    0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71, 0x13, 0x22, 0x32, 0x81, 0x08,
    0x14, 0x42, 0x91, 0xa1, 0xb1, 0xc1, 0x09, 0x23, 0x33, 0x52, 0xf0, 0x15, 0x62, 0x72, 0xd1, 0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25,
    0xf1, 0x17, 0x18, 0x19, 0x1a, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47,
    0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74,
    0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x82, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97,
    0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba,
    0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe2, 0xe3, 0xe4,
    0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};

typedef struct {
    uint8_t lengths[16];
    const uint8_t *values;
    int count;
    uint16_t code[256];   /* by symbol */
    uint8_t code_len[256]; /* by symbol, 0 = no code */
} enc_table;

/* ref: JpegStandardHuffmanEncodingTable.cs:85-131 (BuildCanonicalCode) + JpegHuffmanEncodingTable.GetCode :94-100 */
static void enc_table_build(enc_table *t, const uint8_t lengths[16], const uint8_t *values, int count) {
    memcpy(t->lengths, lengths, 16);
    t->values = values;
    t->count = count;
    memset(t->code, 0, sizeof t->code);
    memset(t->code_len, 0, sizeof t->code_len);
    uint8_t remaining[16];
    memcpy(remaining, lengths, 16);
    uint8_t len_of[256];
    int pos = 0, current = 1;
    for (int i = 0; i < count; i++) {
        while (remaining[pos] == 0) {
            pos++;
            current++;
        }
        remaining[pos]--;
        len_of[i] = (uint8_t)current;
    }
    uint16_t bit_code = 0;
    int bit_count = len_of[0];
    t->code[values[0]] = 0;
    t->code_len[values[0]] = len_of[0];
    for (int i = 1; i < count; i++) {
        if (len_of[i] > bit_count) {
            bit_code++;
            bit_code = (uint16_t)(bit_code << (len_of[i] - bit_count));
            bit_count = len_of[i];
        } else {
            ++bit_code;
        }
        t->code[values[i]] = bit_code;
        t->code_len[values[i]] = len_of[i];
    }
}

int jref_std_huffman_code(int table /* 0 DC lum, 1 AC lum, 2 DC chr, 3 AC chr */, int symbol, int *length) {
    static enc_table tabs[4];
    static int ready = 0;
    if (!ready) {
        enc_table_build(&tabs[0], k_dc_lum_len, k_dc_val, 12);
        enc_table_build(&tabs[1], k_ac_lum_len, k_ac_lum_val, 162);
        enc_table_build(&tabs[2], k_dc_chr_len, k_dc_val, 12);
        enc_table_build(&tabs[3], k_ac_chr_len, k_ac_chr_val, 162);
        ready = 1;
    }
    *length = tabs[table & 3].code_len[symbol & 255];
    return tabs[table & 3].code[symbol & 255];
}

/* ---------------------------------------------------------------------------------------------- writer */

typedef struct {
    uint8_t *p;
    size_t n, cap;
    int overflow;
    uint64_t reg; /* left-justified bit buffer (ref: JpegWriter.cs:17) */
    int bits;
} writer;

static void w_byte(writer *w, uint8_t b) {
    if (w->n < w->cap) w->p[w->n] = b;
    else w->overflow = 1;
    w->n++;
}
static void w_marker(writer *w, uint8_t m) { /* ref: JpegWriter.cs:289-303 */
    w_byte(w, 0xFF);
    w_byte(w, m);
}
static void w_length(writer *w, uint16_t length) { /* ref: :309-321 -- length + 2, big endian */
    uint16_t v = (uint16_t)(length + 2);
    w_byte(w, (uint8_t)(v >> 8));
    w_byte(w, (uint8_t)v);
}
static void w_flush_register(writer *w) { /* ref: :93-116 */
    while (w->bits >= 8) {
        uint8_t b = (uint8_t)(w->reg >> 56);
        w->reg <<= 8;
        w->bits -= 8;
        w_byte(w, b);
        if (b == 0xFF) w_byte(w, 0);
    }
}
static void w_bits(writer *w, uint32_t bits, int length) { /* ref: :184-204 */
    if (w->bits > 32) w_flush_register(w);
    if (length == 0) return; /* (bits << 64) is not a C shift; the reference's shift count wraps to a no-op OR of zero bits */
    w->reg |= ((uint64_t)bits) << (64 - w->bits - length);
    w->bits += length;
}
static void w_exit_bit_mode(writer *w) { /* ref: :123-147 */
    w_flush_register(w);
    if (w->bits > 0) {
        w->reg |= ((((uint64_t)1) << (8 - w->bits)) - 1) << 56;
        w->bits = 8;
        w_flush_register(w);
    }
}

/* ---------------------------------------------------------------------------------------------- block math */

static const uint8_t k_zz_to_natural[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                            41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                            30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

/* FDCT8x4_LeftPart + RightPart over all 8 columns: rows of s -> rows of d (ref: FastFloatingPointDCT.cs:194-311) */
static void fdct_columns(const float *s, float *d) {
    for (int j = 0; j < 8; j++) {
        float c0 = s[0 * 8 + j];
        float c1 = s[7 * 8 + j];
        float t0 = c0 + c1;
        float t7 = c0 - c1;
        c1 = s[6 * 8 + j];
        c0 = s[1 * 8 + j];
        float t1 = c0 + c1;
        float t6 = c0 - c1;
        c1 = s[5 * 8 + j];
        c0 = s[2 * 8 + j];
        float t2 = c0 + c1;
        float t5 = c0 - c1;
        c0 = s[3 * 8 + j];
        c1 = s[4 * 8 + j];
        float t3 = c0 + c1;
        float t4 = c0 - c1;
        c0 = t0 + t3;
        float c3 = t0 - t3;
        c1 = t1 + t2;
        float c2 = t1 - t2;
        d[0 * 8 + j] = c0 + c1;
        d[4 * 8 + j] = c0 - c1;
        float w0 = 0.541196f;
        float w1 = 1.306563f;
        d[2 * 8 + j] = (w0 * c2) + (w1 * c3);
        d[6 * 8 + j] = (w0 * c3) - (w1 * c2);
        w0 = 1.175876f;
        w1 = 0.785695f;
        c3 = (w0 * t4) + (w1 * t7);
        c0 = (w0 * t7) - (w1 * t4);
        w0 = 1.387040f;
        w1 = 0.275899f;
        c2 = (w0 * t5) + (w1 * t6);
        c1 = (w0 * t6) - (w1 * t5);
        d[3 * 8 + j] = c0 - c2;
        d[5 * 8 + j] = c3 - c1;
        const float invsqrt2 = 0.707107f;
        c0 = (c0 + c2) * invsqrt2;
        c3 = (c3 + c1) * invsqrt2;
        d[1 * 8 + j] = c0 + c3;
        d[7 * 8 + j] = c0 - c3;
    }
}
static void transpose8(const float *s, float *d) {
    for (int r = 0; r < 8; r++)
        for (int c = 0; c < 8; c++) d[c * 8 + r] = s[r * 8 + c];
}
/* ref: TransformFDCT :343-362 */
static void transform_fdct(const float *src, float *dest, float *temp) {
    transpose8(src, temp);
    fdct_columns(temp, dest);
    transpose8(dest, temp);
    fdct_columns(temp, dest);
    for (int i = 0; i < 64; i++) dest[i] = dest[i] * 0.1250f;
}

/* ShiftDataLevel + TransformFDCT + ZigZagAndQuantizeBlock on one block of samples (ref: JpegEncoder.cs:801-826):
 * out[i] = (short)MathF.Round(F[natural(i)] / q[i]), i in zig-zag order. */
void jref_fdct_quantize_block(const int16_t *samples, const uint16_t *quant_zigzag, int16_t *out_zigzag) {
    float in[64], f[64], tmp[64];
    for (int i = 0; i < 64; i++) in[i] = (float)(samples[i] - 128);
    transform_fdct(in, f, tmp);
    for (int i = 0; i < 64; i++) {
        float coefficient = f[k_zz_to_natural[i]];
        float q = coefficient / (float)quant_zigzag[i];
        out_zigzag[i] = (int16_t)(int32_t)rintf(q); /* MathF.Round: half to even; (short) conversion wraps */
    }
}

/* ---------------------------------------------------------------------------------------------- input reader */

typedef struct {
    const uint8_t *buf;
    int width, height, component_count;
} buffer_reader;

/* ref: apps/JpegEncode/JpegBufferInputReader.cs:27-52 */
static void reader_read_block(const buffer_reader *r, int16_t *block, int component_index, int x, int y) {
    int bw = r->width - x < 8 ? r->width - x : 8;
    int bh = r->height - y < 8 ? r->height - y : 8;
    if (bw != 8 || bh != 8) memset(block, 0, 64 * sizeof(int16_t));
    for (int oy = 0; oy < bh; oy++) {
        int row = (y + oy) * r->width + x;
        for (int ox = 0; ox < bw; ox++) block[oy * 8 + ox] = r->buf[(size_t)(row + ox) * r->component_count + component_index];
    }
}

static int log2i(int v) {
    int r = 0;
    while (v > 1) {
        v >>= 1;
        r++;
    }
    return r;
}

/* ref: JpegEncoder.cs:743-799.  `block` is NOT cleared: the sub-sampling branch accumulates into what is there. */
static void enc_read_block(const buffer_reader *r, int16_t *block, int component_index, int x, int y, int hs, int vs) {
    if (hs == 1 && vs == 1) {
        reader_read_block(r, block, component_index, x, y);
        return;
    }
    int16_t temp[64];
    int h_shift = log2i(hs), v_shift = log2i(vs);
    int h_block_shift = 3 - h_shift, v_block_shift = 3 - v_shift;
    for (int v = 0; v < vs; v++)
        for (int h = 0; h < hs; h++) {
            reader_read_block(r, temp, component_index, x + 8 * h, y + 8 * v);
            int box = h << h_block_shift, boy = v << v_block_shift;
            for (int yy = 0; yy < 8; yy++)
                for (int xx = 0; xx < 8; xx++) {
                    int16_t *d = &block[(boy + (yy >> v_shift)) * 8 + box + (xx >> h_shift)];
                    *d = (int16_t)(*d + temp[yy * 8 + xx]);
                }
        }
    int total = h_shift + v_shift;
    if (total > 0) {
        int delta = 1 << (total - 1);
        for (int i = 0; i < 64; i++) block[i] = (int16_t)((block[i] + delta) >> total);
    }
}

/* ---------------------------------------------------------------------------------------------- entropy coding */

static int bit_count(int a) { /* BitCountTable (:938-956): bits needed for a, a < 0x10000 */
    int n = 0;
    while (a) {
        n++;
        a >>= 1;
    }
    return n;
}

/* ref: EncodeRunLength :893-918 */
static void encode_run_length(writer *w, const enc_table *t, int run, int value) {
    int a = value, b = value;
    if (a < 0) {
        a = -value;
        b = value - 1;
    }
    int bits = a < 0x100 ? bit_count(a) : 8 + bit_count(a >> 8);
    int sym = (run << 4) | bits;
    w_bits(w, t->code[sym & 255], t->code_len[sym & 255]);
    if (bits > 0) w_bits(w, (uint32_t)b & (uint32_t)((1 << bits) - 1), bits);
}

typedef struct {
    int component_index; /* index into the input buffer's interleaved components == JpegHuffmanEncodingComponent.Index */
    int identifier;      /* ComponentIndex argument of AddComponent: the identifier written to SOF / SOS */
    int h, v, hs, vs;
    const uint16_t *quant;
    int quant_id, dc_id, ac_id;
    const enc_table *dc, *ac;
    int dc_predictor;
} enc_component;

/* ref: EncodeBlock :828-870 */
static void encode_block(writer *w, enc_component *c, const int16_t *block) {
    int value = block[0];
    int t = value - c->dc_predictor;
    c->dc_predictor = value;
    encode_run_length(w, c->dc, 0, t);
    int run = 0;
    for (int i = 1; i < 64; i++) {
        t = block[i];
        if (t == 0) {
            run++;
        } else {
            while (run > 15) {
                w_bits(w, c->ac->code[0xF0], c->ac->code_len[0xF0]);
                run -= 16;
            }
            encode_run_length(w, c->ac, run, t);
            run = 0;
        }
    }
    if (run > 0) w_bits(w, c->ac->code[0], c->ac->code_len[0]);
}

/* ---------------------------------------------------------------------------------------------- Encode */

/* The call sequence of apps/JpegEncode/EncodeAction.cs:38-63 with optimizeCoding = false, generalised to the luma
 * sampling (h, v) and to 1 or 3 components: standard tables scaled by `quality`, standard Huffman tables,
 * AddComponent(1, 0, 0, 0, h, v) [, AddComponent(2, 1, 1, 1, 1, 1), AddComponent(3, 1, 1, 1, 1, 1)].
 * `pixels` = interleaved 8-bit samples (JpegBufferInputReader(width, height, components, buffer)).
 * coef_tap (optional): receives every block's quantised zig-zag coefficients in encoding order.
 * Returns 0 and the byte count in *out_len; 1 when `cap` is too small (out_len still holds the size needed). */
/* GatherRunLengthCodeStatistics :872-891 */
static void gather_run_length(uint32_t *freq, int run, int value) {
    int a = value < 0 ? -value : value;
    int bits = a < 0x100 ? bit_count(a) : 8 + bit_count(a >> 8);
    freq[((run << 4) | bits) & 255]++;
}
/* GatherBlockStatistics :552-597 */
static void gather_block(enc_component *c, const int16_t *block, uint32_t *dc_freq, uint32_t *ac_freq) {
    int value = block[0];
    int t = value - c->dc_predictor;
    c->dc_predictor = value;
    gather_run_length(dc_freq, 0, t);
    int run = 0;
    for (int i = 1; i < 64; i++) {
        t = block[i];
        if (t == 0) {
            run++;
        } else {
            while (run > 15) {
                ac_freq[0xF0]++;
                run -= 16;
            }
            gather_run_length(ac_freq, run, t);
            run = 0;
        }
    }
    if (run > 0) ac_freq[0]++;
}


int jref_encode_8bit(const uint8_t *pixels, int width, int height, int components, int luma_h, int luma_v, int quality,
                     uint8_t *out, size_t cap, size_t *out_len, int16_t *coef_tap) {
    return jref_encode_8bit_ex(pixels, width, height, components, luma_h, luma_v, quality, 0, out, cap, out_len, coef_tap);
}

/* optimize_coding != 0: EncodeAction's other branch (apps/JpegEncode/EncodeAction.cs:40-46): SetHuffmanTable(isDc, id) with no
 * table for DC0, AC0, DC1, AC1, which makes Encode() (JpegEncoder.cs:255-291) run TransformBlocks (:414-485) into a
 * JpegBlockAllocator, BuildHuffmanTables (:491-550; JpegHuffmanEncodingTableBuilder.Build(false), restated in jpegopt.inc),
 * WriteHuffmanTables with the built tables and WritePreparedScanData (:604-655).  Two things differ from the other branch
 * and are restated as they are: every block is transformed in its OWN (zeroed) allocator block, so the sub-sampling reader
 * no longer accumulates onto the previous block's coefficients; and blocks of the MCU grid that lie outside a component's
 * own block grid all alias the allocator's ONE dummy block (JpegBlockAllocator.cs:93-114), so they are encoded with whatever
 * the last of them left there.  Returns 2 when a table has no symbol ("No symbol is recorded.": a single-component image,
 * whose chrominance builders stay empty). */
int jref_encode_8bit_ex(const uint8_t *pixels, int width, int height, int components, int luma_h, int luma_v, int quality,
                        int optimize_coding, uint8_t *out, size_t cap, size_t *out_len, int16_t *coef_tap) {
    return jref_encode_8bit_dri(pixels, width, height, components, luma_h, luma_v, quality, optimize_coding, 0, out, cap, out_len, coef_tap);
}

/* restart_interval != 0 is an EXTENSION (SURVEY 8f N3 "+ DRI emission"): the reference's encoder writes no restart markers
 * at all, so there is no reference behaviour to restate.  The definition is T.81's (B.2.4.4, E.1.4; what libjpeg writes): a DRI
 * segment -- placed in front of SOF0, where the reference's decoder sees it even without Identify() -- and, in
 * front of every MCU whose index is a non-zero multiple of the interval, the bit buffer padded with one-bits (ExitBitMode),
 * RSTm with m counting modulo 8, and every DC predictor back at zero.  The statistics pass of optimizeCoding resets its
 * predictors at the same MCUs.  What pins it: the golden-pinned decoder restatement decodes such a stream to the coefficients
 * that went in, and so does libjpeg-turbo. */
int jref_encode_8bit_dri(const uint8_t *pixels, int width, int height, int components, int luma_h, int luma_v, int quality,
                         int optimize_coding, int restart_interval, uint8_t *out, size_t cap, size_t *out_len, int16_t *coef_tap) {
    return jref_encode_8bit_tables(pixels, width, height, components, luma_h, luma_v, quality, NULL, NULL, optimize_coding, restart_interval, out,
                                   cap, out_len, coef_tap);
}

/* quant_lum / quant_chr != NULL: the tables the caller handed to SetQuantizationTable (identifiers 0 and 1, zig-zag order, element
 * precision 0: JpegEncoder.cs:102-126) instead of the standard ones scaled by `quality`. */
int jref_encode_8bit_tables(const uint8_t *pixels, int width, int height, int components, int luma_h, int luma_v, int quality,
                            const uint16_t *quant_lum, const uint16_t *quant_chr, int optimize_coding, int restart_interval, uint8_t *out,
                            size_t cap, size_t *out_len, int16_t *coef_tap) {
    uint16_t q_lum[64], q_chr[64];
    jref_scale_quant_table(k_std_lum, quality > 0 ? quality : 50, q_lum);
    jref_scale_quant_table(k_std_chr, quality > 0 ? quality : 50, q_chr);
    if (quant_lum) memcpy(q_lum, quant_lum, sizeof q_lum);
    if (quant_chr) memcpy(q_chr, quant_chr, sizeof q_chr);
    enc_table t_dc_lum, t_ac_lum, t_dc_chr, t_ac_chr;
    enc_table_build(&t_dc_lum, k_dc_lum_len, k_dc_val, 12);
    enc_table_build(&t_ac_lum, k_ac_lum_len, k_ac_lum_val, 162);
    enc_table_build(&t_dc_chr, k_dc_chr_len, k_dc_val, 12);
    enc_table_build(&t_ac_chr, k_ac_chr_len, k_ac_chr_val, 162);

    enc_component comps[3];
    int ncomp = components == 1 ? 1 : 3;
    comps[0] = (enc_component){0, 1, luma_h, luma_v, 1, 1, q_lum, 0, 0, 0, &t_dc_lum, &t_ac_lum, 0};
    comps[1] = (enc_component){1, 2, 1, 1, 1, 1, q_chr, 1, 1, 1, &t_dc_chr, &t_ac_chr, 0};
    comps[2] = (enc_component){2, 3, 1, 1, 1, 1, q_chr, 1, 1, 1, &t_dc_chr, &t_ac_chr, 0};
    if (ncomp == 1) {
        comps[0].h = luma_h;
        comps[0].v = luma_v;
    }

    writer w = {out, 0, cap, 0, 0, 0};
    w_marker(&w, 0xD8); /* WriteStartOfImage */
    /* WriteQuantizationTables :305-335: one DQT segment, tables in SetQuantizationTable order */
    {
        int ntab = 2;
        w_marker(&w, 0xDB);
        w_length(&w, (uint16_t)(ntab * 65));
        const uint16_t *tabs[2] = {q_lum, q_chr};
        for (int t = 0; t < ntab; t++) {
            w_byte(&w, (uint8_t)((0 << 4) | t));
            for (int i = 0; i < 64; i++) w_byte(&w, (uint8_t)tabs[t][i]);
        }
    }
    if (restart_interval > 0) { /* extension: DRI, Lr = 4 */
        w_marker(&w, 0xDD);
        w_length(&w, 2);
        w_byte(&w, (uint8_t)(restart_interval >> 8));
        w_byte(&w, (uint8_t)restart_interval);
    }
    /* WriteStartOfFrame :353-386 */
    {
        w_marker(&w, 0xC0);
        w_length(&w, (uint16_t)(6 + 3 * ncomp));
        w_byte(&w, 8);
        w_byte(&w, (uint8_t)(height >> 8));
        w_byte(&w, (uint8_t)height);
        w_byte(&w, (uint8_t)(width >> 8));
        w_byte(&w, (uint8_t)width);
        w_byte(&w, (uint8_t)ncomp);
        for (int i = 0; i < ncomp; i++) {
            w_byte(&w, (uint8_t)comps[i].identifier);
            w_byte(&w, (uint8_t)((comps[i].h << 4) | (comps[i].v & 0xF)));
            w_byte(&w, (uint8_t)comps[i].quant_id);
        }
    }
    /* ---- optimizeCoding: TransformBlocks + BuildHuffmanTables, then the built tables replace the standard ones */
    int16_t *store = NULL; /* JpegBlockAllocator: block 0 = the dummy, then the components' own grids */
    int grid_w[3] = {0, 0, 0}, grid_h[3] = {0, 0, 0}, grid_off[3] = {0, 0, 0};
    enc_table o_tab[4];
    uint8_t o_bits[4][16], o_vals[4][256];
    int o_n[4] = {0, 0, 0, 0};
    if (optimize_coding) {
        int max_h = 1, max_v = 1;
        for (int i = 0; i < ncomp; i++) {
            comps[i].dc_predictor = 0;
            if (comps[i].h > max_h) max_h = comps[i].h;
            if (comps[i].v > max_v) max_v = comps[i].v;
        }
        int hb = (width + 7) / 8, vb = (height + 7) / 8, index = 1;
        for (int i = 0; i < ncomp; i++) {
            comps[i].hs = max_h / comps[i].h;
            comps[i].vs = max_v / comps[i].v;
            grid_w[i] = (hb + comps[i].hs - 1) / comps[i].hs;
            grid_h[i] = (vb + comps[i].vs - 1) / comps[i].vs;
            grid_off[i] = index;
            index += grid_w[i] * grid_h[i];
        }
        store = (int16_t *)calloc((size_t)index * 64, sizeof(int16_t));
#define BLOCK_REF(ci, bx, by) (store + 64 * (size_t)(((bx) >= grid_w[ci] || (by) >= grid_h[ci]) ? 0 : grid_off[ci] + (by)*grid_w[ci] + (bx)))
        int mcus_per_line = (width + 8 * max_h - 1) / (8 * max_h);
        int mcus_per_column = (height + 8 * max_v - 1) / (8 * max_v);
        buffer_reader rd = {pixels, width, height, components};
        for (int row_mcu = 0; row_mcu < mcus_per_column; row_mcu++)
            for (int col_mcu = 0; col_mcu < mcus_per_line; col_mcu++)
                for (int ci = 0; ci < ncomp; ci++) {
                    enc_component *c = &comps[ci];
                    int offset_x = col_mcu * c->h, offset_y = row_mcu * c->v;
                    for (int y = 0; y < c->v; y++)
                        for (int x = 0; x < c->h; x++) {
                            int16_t *block = BLOCK_REF(ci, offset_x + x, offset_y + y);
                            enc_read_block(&rd, block, c->component_index, (offset_x + x) * 8 * c->hs, (offset_y + y) * 8 * c->vs, c->hs, c->vs);
                            int16_t q[64];
                            jref_fdct_quantize_block(block, c->quant, q);
                            memcpy(block, q, sizeof q);
                        }
                }
        /* BuildHuffmanTables */
        uint32_t freq[4][256];
        memset(freq, 0, sizeof freq);
        for (int i = 0; i < ncomp; i++) comps[i].dc_predictor = 0;
        for (int row_mcu = 0; row_mcu < mcus_per_column; row_mcu++)
            for (int col_mcu = 0; col_mcu < mcus_per_line; col_mcu++) {
                const int mcu_index = row_mcu * mcus_per_line + col_mcu;
                if (restart_interval > 0 && mcu_index > 0 && mcu_index % restart_interval == 0)
                    for (int i = 0; i < ncomp; i++) comps[i].dc_predictor = 0;
                for (int ci = 0; ci < ncomp; ci++) {
                    enc_component *c = &comps[ci];
                    for (int y = 0; y < c->v; y++)
                        for (int x = 0; x < c->h; x++)
                            gather_block(c, BLOCK_REF(ci, col_mcu * c->h + x, row_mcu * c->v + y), freq[2 * c->dc_id], freq[2 * c->ac_id + 1]);
                }
            }
        for (int t = 0; t < 4; t++) { /* _huffmanTables.BuildTables: every builder of the collection, in SetHuffmanTable order */
            uint16_t code[256];
            uint8_t len[256];
            if (jref_build_optimal_table_ex(freq[t], optimize_coding == 2 /* MostOptimalCoding */, o_bits[t], o_vals[t], &o_n[t], code, len) != 0) {
                free(store);
                *out_len = 0;
                return 2;
            }
            memset(&o_tab[t], 0, sizeof o_tab[t]);
            for (int sym = 0; sym < 256; sym++) {
                o_tab[t].code[sym] = code[sym];
                o_tab[t].code_len[sym] = len[sym];
            }
        }
        comps[0].dc = &o_tab[0];
        comps[0].ac = &o_tab[1];
        for (int i = 1; i < ncomp; i++) {
            comps[i].dc = &o_tab[2];
            comps[i].ac = &o_tab[3];
        }
        const int cls[4] = {0, 1, 0, 1}, ids[4] = {0, 0, 1, 1};
        int total = 0;
        for (int t = 0; t < 4; t++) total += 1 + 16 + o_n[t];
        w_marker(&w, 0xC4);
        w_length(&w, (uint16_t)total);
        for (int t = 0; t < 4; t++) {
            w_byte(&w, (uint8_t)((cls[t] << 4) | ids[t]));
            for (int l = 0; l < 16; l++) w_byte(&w, o_bits[t][l]);
            for (int i = 0; i < o_n[t]; i++) w_byte(&w, o_vals[t][i]);
        }
    }
    /* WriteHuffmanTables :336-352: one DHT segment, tables in SetHuffmanTable order (DC0, AC0, DC1, AC1) */
    if (!optimize_coding) {
        const enc_table *tabs[4] = {&t_dc_lum, &t_ac_lum, &t_dc_chr, &t_ac_chr};
        const int cls[4] = {0, 1, 0, 1}, ids[4] = {0, 0, 1, 1};
        int total = 0;
        for (int t = 0; t < 4; t++) total += 1 + 16 + tabs[t]->count;
        w_marker(&w, 0xC4);
        w_length(&w, (uint16_t)total);
        for (int t = 0; t < 4; t++) {
            w_byte(&w, (uint8_t)((cls[t] << 4) | ids[t]));
            for (int l = 0; l < 16; l++) w_byte(&w, tabs[t]->lengths[l]);
            for (int i = 0; i < tabs[t]->count; i++) w_byte(&w, tabs[t]->values[i]);
        }
    }
    /* WriteStartOfScan :387-413 */
    {
        w_marker(&w, 0xDA);
        w_length(&w, (uint16_t)(1 + 2 * ncomp + 3));
        w_byte(&w, (uint8_t)ncomp);
        for (int i = 0; i < ncomp; i++) {
            w_byte(&w, (uint8_t)comps[i].identifier);
            w_byte(&w, (uint8_t)((comps[i].dc_id << 4) | (comps[i].ac_id & 0xF)));
        }
        w_byte(&w, 0);
        w_byte(&w, 63);
        w_byte(&w, 0);
    }
    if (optimize_coding) { /* WritePreparedScanData :604-655 */
        int max_h = 1, max_v = 1;
        for (int i = 0; i < ncomp; i++) {
            comps[i].dc_predictor = 0;
            if (comps[i].h > max_h) max_h = comps[i].h;
            if (comps[i].v > max_v) max_v = comps[i].v;
        }
        int mcus_per_line = (width + 8 * max_h - 1) / (8 * max_h);
        int mcus_per_column = (height + 8 * max_v - 1) / (8 * max_v);
        size_t nblock = 0;
        int rst = 0;
        for (int row_mcu = 0; row_mcu < mcus_per_column; row_mcu++)
            for (int col_mcu = 0; col_mcu < mcus_per_line; col_mcu++) {
                const int mcu_index = row_mcu * mcus_per_line + col_mcu;
                if (restart_interval > 0 && mcu_index > 0 && mcu_index % restart_interval == 0) {
                    w_exit_bit_mode(&w);
                    w_marker(&w, (uint8_t)(0xD0 + (rst++ & 7)));
                    for (int i = 0; i < ncomp; i++) comps[i].dc_predictor = 0;
                }
                for (int ci = 0; ci < ncomp; ci++) {
                    enc_component *c = &comps[ci];
                    for (int y = 0; y < c->v; y++)
                        for (int x = 0; x < c->h; x++) {
                            const int16_t *block = BLOCK_REF(ci, col_mcu * c->h + x, row_mcu * c->v + y);
                            if (coef_tap) memcpy(coef_tap + nblock * 64, block, 64 * sizeof(int16_t));
                            nblock++;
                            encode_block(&w, c, block);
                        }
                }
            }
        w_exit_bit_mode(&w);
        free(store);
#undef BLOCK_REF
    } else
    /* WriteScanData :662-741 */
    {
        int max_h = 1, max_v = 1;
        for (int i = 0; i < ncomp; i++) {
            comps[i].dc_predictor = 0;
            if (comps[i].h > max_h) max_h = comps[i].h;
            if (comps[i].v > max_v) max_v = comps[i].v;
        }
        for (int i = 0; i < ncomp; i++) {
            comps[i].hs = max_h / comps[i].h;
            comps[i].vs = max_v / comps[i].v;
        }
        int mcus_per_line = (width + 8 * max_h - 1) / (8 * max_h);
        int mcus_per_column = (height + 8 * max_v - 1) / (8 * max_v);
        buffer_reader rd = {pixels, width, height, components};
        int16_t input_buffer[64]; /* ONE buffer for every block (see the header comment) */
        memset(input_buffer, 0, sizeof input_buffer);
        size_t nblock = 0;
        int rst = 0;
        for (int row_mcu = 0; row_mcu < mcus_per_column; row_mcu++) {
            int offset_y = row_mcu * max_v;
            for (int col_mcu = 0; col_mcu < mcus_per_line; col_mcu++) {
                int offset_x = col_mcu * max_h;
                const int mcu_index = row_mcu * mcus_per_line + col_mcu;
                if (restart_interval > 0 && mcu_index > 0 && mcu_index % restart_interval == 0) {
                    w_exit_bit_mode(&w);
                    w_marker(&w, (uint8_t)(0xD0 + (rst++ & 7)));
                    for (int i = 0; i < ncomp; i++) comps[i].dc_predictor = 0;
                }
                for (int ci = 0; ci < ncomp; ci++) {
                    enc_component *c = &comps[ci];
                    for (int y = 0; y < c->v; y++) {
                        int block_offset_y = (offset_y + y) * 8;
                        for (int x = 0; x < c->h; x++) {
                            enc_read_block(&rd, input_buffer, c->component_index, (offset_x + x) * 8, block_offset_y, c->hs, c->vs);
                            int16_t q[64];
                            jref_fdct_quantize_block(input_buffer, c->quant, q);
                            memcpy(input_buffer, q, sizeof q); /* ZigZagAndQuantizeBlock writes into the same buffer */
                            if (coef_tap) memcpy(coef_tap + nblock * 64, q, sizeof q);
                            nblock++;
                            encode_block(&w, c, input_buffer);
                        }
                    }
                }
            }
        }
        w_exit_bit_mode(&w);
    }
    w_marker(&w, 0xD9);
    *out_len = w.n;
    return w.overflow ? 1 : 0;
}

/* ---------------------------------------------------------------------------------------------- RGB -> YCbCr
 * ref: apps/JpegEncode/JpegRgbToYCbCrConverter.cs (the caller's step before the encoder), restated in jref_rgb_to_ycbcr8. */

#define R2Y_SCALE_BITS 16
static int r2y_fix(float x) { return (int)((x * (float)(1L << R2Y_SCALE_BITS)) + 0.5F); } /* ref: :59-62 */

/* stride = bytes per source pixel: 3 = ConvertRgb24ToYCbCr8 (:64-96); 4 = ConvertRgba32ToYCbCr8, the benchmark project's copy of the
 * converter (tests/JpegLibrary.Benchmarks/ColorConverters/JpegRgbToYCbCrConverter.cs:95-124): the same tables, the source stepping
 * over a fourth byte per pixel. */
static void r2y_convert(const uint8_t *rgb, uint8_t *ycbcr, size_t count, int stride) {
    static int y_r[256], y_g[256], y_b[256], cb_r[256], cb_g[256], cb_b[256], cr_g[256], cr_b[256];
    static int ready = 0;
    if (!ready) { /* constructor :26-57 */
        const int cbcr_offset = 128 << R2Y_SCALE_BITS, half = 1 << (R2Y_SCALE_BITS - 1);
        for (int i = 0; i < 256; i++) {
            y_r[i] = r2y_fix(0.299F) * i;
            y_g[i] = r2y_fix(0.587F) * i;
            y_b[i] = (r2y_fix(0.114F) * i) + half;
            cb_r[i] = (-r2y_fix(0.168735892F)) * i;
            cb_g[i] = (-r2y_fix(0.331264108F)) * i;
            cb_b[i] = (r2y_fix(0.5F) * i) + cbcr_offset + half - 1;
            cr_g[i] = (-r2y_fix(0.418687589F)) * i;
            cr_b[i] = (-r2y_fix(0.081312411F)) * i;
        }
        ready = 1;
    }
    for (size_t i = 0; i < count; i++) { /* ConvertRgb24ToYCbCr8 :64-96 */
        uint8_t r = rgb[0], g = rgb[1], b = rgb[2];
        ycbcr[0] = (uint8_t)((y_r[r] + y_g[g] + y_b[b]) >> R2Y_SCALE_BITS);
        ycbcr[1] = (uint8_t)((cb_r[r] + cb_g[g] + cb_b[b]) >> R2Y_SCALE_BITS);
        ycbcr[2] = (uint8_t)((cb_b[r] + cr_g[g] + cr_b[b]) >> R2Y_SCALE_BITS);
        rgb += stride;
        ycbcr += 3;
    }
}
void jref_rgb_to_ycbcr8(const uint8_t *rgb, uint8_t *ycbcr, size_t count) { r2y_convert(rgb, ycbcr, count, 3); }
void jref_rgba_to_ycbcr8(const uint8_t *rgba, uint8_t *ycbcr, size_t count) { r2y_convert(rgba, ycbcr, count, 4); }
