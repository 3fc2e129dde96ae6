/*
 * tools/jpegsynth.c -- deterministic synthetic baseline-JPEG generator (bench / test input only).
 *
 * The reference's encoder (JpegEncoder.cs) has no restart-interval support, so it cannot produce the
 * benchmark's DRI=4 inputs.  This is the build's own minimal baseline encoder:
 *   synthetic RGB (SURVEY 8d recipe) -> JFIF YCbCr -> 4:4:4 / 4:2:2 / 4:2:0 / gray, or any per-component sampling factors
 *   (H, V) in 1..4 (round 6: 4:1:1, 4:4:0, 1x4, luma BELOW chroma, factors that are neither the maximum nor 1 ...) -> float AAN
 *   FDCT -> Annex-K tables scaled libjpeg-style -> standard Huffman tables (one 418-byte DHT) -> DRI/RSTn -> EOI.
 * It is NOT on the decode path and NOT an oracle; files it writes are ordinary JFIF files (Pillow decodes them).
 * All segment lengths have a low byte >= 2, so the reference's length quirk (JpegReader.cs:174) is never hit.
 */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct jsynth_params {
    int width, height;
    int subsampling;      /* 0 = 4:4:4, 1 = 4:2:2, 2 = 4:2:0, 3 = grayscale */
    int quality;          /* 1..100, libjpeg scaling */
    int restart_interval; /* MCUs, 0 = none */
    uint64_t seed;
    int noninterleaved;   /* three single-component scans instead of one interleaved scan.  The blocks of a scan go out in the
                             order the REFERENCE reads them -- h x v blocks of the component per frame MCU
                             (JpegHuffmanBaselineScanDecoder.cs:99-134 treats every scan as interleaved) -- which is T.81's order
                             only for 4:4:4; noninterleaved = 2: T.81's order (one block per MCU, the component's own block
                             raster), which the reference reads as something else */
    int samp[3][2];       /* per component {H, V}; all zero = take `subsampling`.  A component is the box average over
                             (Hmax / H) x (Vmax / V) pixels (ratios must be whole) */
    int progressive;      /* SOF2 with a fixed script (round 6: the layouts above through the progressive scan decoder): DC of all
                             components interleaved with Al = 1, the DC refinement, then per component the AC bands 1..5 and 6..63
                             (Al = 0; EOB per block, no EOB runs: the standard tables carry no EOBn symbols) in T.81's
                             non-interleaved order = the order JpegHuffmanProgressiveScanDecoder reads
                             (DecodeProgressiveDataNonInterleaved, :140-194).  restart_interval counts each scan's own MCUs */
} jsynth_params;

static const uint8_t k_zigzag_to_natural[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,
                                                12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6,  7,  14, 21, 28,
                                                35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51,
                                                58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

/* ITU-T T.81 Annex K.1, natural (row-major) order */
static const uint8_t k_std_lum[64] = {16, 11, 10, 16, 24,  40,  51,  61,  12, 12, 14, 19, 26,  58,  60,  55,
                                      14, 13, 16, 24, 40,  57,  69,  56,  14, 17, 22, 29, 51,  87,  80,  62,
                                      18, 22, 37, 56, 68,  109, 103, 77,  24, 35, 55, 64, 81,  104, 113, 92,
                                      49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99};
static const uint8_t k_std_chr[64] = {17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99,
                                      24, 26, 56, 99, 99, 99, 99, 99, 47, 66, 99, 99, 99, 99, 99, 99,
                                      99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99,
                                      99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99};

/* ITU-T T.81 Annex K.3 */
static const uint8_t k_dc_lum_bits[16] = {0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0};
static const uint8_t k_dc_chr_bits[16] = {0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0};
static const uint8_t k_dc_vals[12] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11};
static const uint8_t k_ac_lum_bits[16] = {0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 125};
static const uint8_t k_ac_lum_vals[162] = {
    0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71,
    0x14, 0x32, 0x81, 0x91, 0xa1, 0x08, 0x23, 0x42, 0xb1, 0xc1, 0x15, 0x52, 0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72,
    0x82, 0x09, 0x0a, 0x16, 0x17, 0x18, 0x19, 0x1a, 0x25, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x34, 0x35, 0x36, 0x37,
    0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59,
    0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x83,
    0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3,
    0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3,
    0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2,
    0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};
static const uint8_t k_ac_chr_bits[16] = {0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 119};
static const uint8_t k_ac_chr_vals[162] = {
    0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71, 0x13, 0x22,
    0x32, 0x81, 0x08, 0x14, 0x42, 0x91, 0xa1, 0xb1, 0xc1, 0x09, 0x23, 0x33, 0x52, 0xf0, 0x15, 0x62, 0x72, 0xd1,
    0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25, 0xf1, 0x17, 0x18, 0x19, 0x1a, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x35, 0x36,
    0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58,
    0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a,
    0x82, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a,
    0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba,
    0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda,
    0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};

typedef struct enc_table {
    uint16_t code[256];
    uint8_t size[256];
} enc_table;

static void build_enc_table(enc_table *t, const uint8_t bits[16], const uint8_t *vals) {
    memset(t, 0, sizeof *t);
    int code = 0, k = 0;
    for (int l = 1; l <= 16; l++) {
        for (int i = 0; i < bits[l - 1]; i++, k++) {
            t->code[vals[k]] = (uint16_t)code++;
            t->size[vals[k]] = (uint8_t)l;
        }
        code <<= 1;
    }
}

typedef struct bitwriter {
    uint8_t *p, *end;
    uint64_t acc;
    int nbits;
    int overflow;
} bitwriter;

static inline void bw_emit_byte(bitwriter *w, uint8_t b) {
    if (w->p + 2 > w->end) {
        w->overflow = 1;
        return;
    }
    *w->p++ = b;
    if (b == 0xFF) *w->p++ = 0x00;
}

static inline void bw_put(bitwriter *w, uint32_t code, int size) {
    w->acc = (w->acc << size) | (code & ((1u << size) - 1));
    w->nbits += size;
    while (w->nbits >= 8) {
        w->nbits -= 8;
        bw_emit_byte(w, (uint8_t)(w->acc >> w->nbits));
    }
}

static inline void bw_flush_ones(bitwriter *w) {
    if (w->nbits > 0) {
        int pad = 8 - w->nbits;
        bw_put(w, (1u << pad) - 1, pad);
    }
    w->acc = 0;
    w->nbits = 0;
}

/* float AAN forward DCT (same factorisation as IJG jfdctflt); output is scaled by the AAN factors. */
static void fdct_aan(float *d) {
    float *p = d;
    for (int i = 0; i < 8; i++, p += 8) {
        float t0 = p[0] + p[7], t7 = p[0] - p[7], t1 = p[1] + p[6], t6 = p[1] - p[6];
        float t2 = p[2] + p[5], t5 = p[2] - p[5], t3 = p[3] + p[4], t4 = p[3] - p[4];
        float t10 = t0 + t3, t13 = t0 - t3, t11 = t1 + t2, t12 = t1 - t2;
        p[0] = t10 + t11;
        p[4] = t10 - t11;
        float z1 = (t12 + t13) * 0.707106781f;
        p[2] = t13 + z1;
        p[6] = t13 - z1;
        t10 = t4 + t5;
        t11 = t5 + t6;
        t12 = t6 + t7;
        float z5 = (t10 - t12) * 0.382683433f;
        float z2 = 0.541196100f * t10 + z5, z4 = 1.306562965f * t12 + z5, z3 = t11 * 0.707106781f;
        float z11 = t7 + z3, z13 = t7 - z3;
        p[5] = z13 + z2;
        p[3] = z13 - z2;
        p[1] = z11 + z4;
        p[7] = z11 - z4;
    }
    p = d;
    for (int i = 0; i < 8; i++, p++) {
        float t0 = p[0] + p[56], t7 = p[0] - p[56], t1 = p[8] + p[48], t6 = p[8] - p[48];
        float t2 = p[16] + p[40], t5 = p[16] - p[40], t3 = p[24] + p[32], t4 = p[24] - p[32];
        float t10 = t0 + t3, t13 = t0 - t3, t11 = t1 + t2, t12 = t1 - t2;
        p[0] = t10 + t11;
        p[32] = t10 - t11;
        float z1 = (t12 + t13) * 0.707106781f;
        p[16] = t13 + z1;
        p[48] = t13 - z1;
        t10 = t4 + t5;
        t11 = t5 + t6;
        t12 = t6 + t7;
        float z5 = (t10 - t12) * 0.382683433f;
        float z2 = 0.541196100f * t10 + z5, z4 = 1.306562965f * t12 + z5, z3 = t11 * 0.707106781f;
        float z11 = t7 + z3, z13 = t7 - z3;
        p[40] = z13 + z2;
        p[24] = z13 - z2;
        p[8] = z11 + z4;
        p[56] = z11 - z4;
    }
}

static inline uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

typedef struct synth_ctx {
    int width, height;
    uint64_t seed;
    /* separable trig tables of the SURVEY 8d recipe */
    float *sx0, *cy1;               /* R = 128 + 70 sin(x/37+p0) cos(y/53+p1) */
    float *gcx, *gsx, *gcy, *gsy;   /* G = 128 + 60 cos(x/91 + y/29 + p2)     */
    float *bsx, *bcx, *bsy, *bcy;   /* B = 128 + 90 sin((x+y)/67 + p3)        */
} synth_ctx;

static void synth_init(synth_ctx *s, int w, int h, uint64_t seed) {
    s->width = w;
    s->height = h;
    s->seed = seed;
    double ph[4];
    for (int i = 0; i < 4; i++) ph[i] = (double)(splitmix64(seed * 4 + i) >> 11) * (6.283185307179586 / 9007199254740992.0);
    size_t n = (size_t)(w > h ? w : h) + 16;
    float *buf = (float *)malloc(sizeof(float) * n * 10);
    s->sx0 = buf;
    s->cy1 = buf + n;
    s->gcx = buf + 2 * n;
    s->gsx = buf + 3 * n;
    s->gcy = buf + 4 * n;
    s->gsy = buf + 5 * n;
    s->bsx = buf + 6 * n;
    s->bcx = buf + 7 * n;
    s->bsy = buf + 8 * n;
    s->bcy = buf + 9 * n;
    for (int x = 0; x < w + 16; x++) {
        s->sx0[x] = (float)sin(x / 37.0 + ph[0]);
        s->gcx[x] = (float)cos(x / 91.0 + ph[2]);
        s->gsx[x] = (float)sin(x / 91.0 + ph[2]);
        s->bsx[x] = (float)sin(x / 67.0 + ph[3]);
        s->bcx[x] = (float)cos(x / 67.0 + ph[3]);
    }
    for (int y = 0; y < h + 16; y++) {
        s->cy1[y] = (float)cos(y / 53.0 + ph[1]);
        s->gcy[y] = (float)cos(y / 29.0);
        s->gsy[y] = (float)sin(y / 29.0);
        s->bsy[y] = (float)sin(y / 67.0);
        s->bcy[y] = (float)cos(y / 67.0);
    }
}

static void synth_free(synth_ctx *s) { free(s->sx0); }

static inline float clip255(float v) { return v < 0.f ? 0.f : (v > 255.f ? 255.f : v); }

/* One row of Y/Cb/Cr (float, level-shifted by -128 for Y only later). Edge pixels replicate (x>=w -> w-1). */
static void synth_row(const synth_ctx *s, int y, int wpad, float *Y, float *Cb, float *Cr) {
    int yy = y < s->height ? y : s->height - 1;
    float cy1 = s->cy1[yy], gcy = s->gcy[yy], gsy = s->gsy[yy], bsy = s->bsy[yy], bcy = s->bcy[yy];
    uint64_t rowkey = s->seed * 0x100000001B3ull + (uint64_t)yy * 0x1000003ull;
    for (int x = 0; x < wpad; x++) {
        int xx = x < s->width ? x : s->width - 1;
        uint64_t h = splitmix64(rowkey + (uint64_t)xx);
        /* Irwin-Hall(3) of 7-bit uniforms: sigma = 64, scaled to sigma = 8 */
        float n0 = (float)((int)((h & 127) + ((h >> 7) & 127) + ((h >> 14) & 127)) * 2 - 381) * (1.0f / 16.0f);
        float n1 = (float)((int)(((h >> 21) & 127) + ((h >> 28) & 127) + ((h >> 35) & 127)) * 2 - 381) * (1.0f / 16.0f);
        float n2 = (float)((int)(((h >> 42) & 127) + ((h >> 49) & 127) + ((h >> 56) & 127)) * 2 - 381) * (1.0f / 16.0f);
        float r = 128.f + 70.f * s->sx0[xx] * cy1 + n0;
        float g = 128.f + 60.f * (s->gcx[xx] * gcy - s->gsx[xx] * gsy) + n1;
        float b = 128.f + 90.f * (s->bsx[xx] * bcy + s->bcx[xx] * bsy) + n2;
        r = floorf(clip255(r) + 0.5f);
        g = floorf(clip255(g) + 0.5f);
        b = floorf(clip255(b) + 0.5f);
        Y[x] = floorf(0.299f * r + 0.587f * g + 0.114f * b + 0.5f);
        Cb[x] = floorf(clip255(-0.168736f * r - 0.331264f * g + 0.5f * b + 128.f) + 0.5f);
        Cr[x] = floorf(clip255(0.5f * r - 0.418688f * g - 0.081312f * b + 128.f) + 0.5f);
    }
}

static void scale_quant(const uint8_t *base, int quality, uint8_t out_nat[64]) {
    if (quality < 1) quality = 1;
    if (quality > 100) quality = 100;
    int scale = quality < 50 ? 5000 / quality : 200 - quality * 2;
    for (int i = 0; i < 64; i++) {
        int v = (base[i] * scale + 50) / 100;
        out_nat[i] = (uint8_t)(v < 1 ? 1 : (v > 255 ? 255 : v));
    }
}

static inline int bit_size(int v) {
    int a = v < 0 ? -v : v, n = 0;
    while (a) {
        n++;
        a >>= 1;
    }
    return n;
}

static void encode_block(bitwriter *w, const float *samples /*64, level-shifted*/, const float *qdiv /*natural*/,
                         const enc_table *dc, const enc_table *ac, int *pred) {
    float d[64];
    memcpy(d, samples, sizeof d);
    fdct_aan(d);
    int q[64];
    for (int i = 0; i < 64; i++) q[i] = (int)lrintf(d[i] * qdiv[i]);
    int diff = q[0] - *pred;
    *pred = q[0];
    int s = bit_size(diff);
    bw_put(w, dc->code[s], dc->size[s]);
    if (s) bw_put(w, (uint32_t)(diff < 0 ? diff - 1 : diff), s);
    int run = 0;
    for (int k = 1; k < 64; k++) {
        int v = q[k_zigzag_to_natural[k]];
        if (v == 0) {
            run++;
            continue;
        }
        while (run > 15) {
            bw_put(w, ac->code[0xF0], ac->size[0xF0]);
            run -= 16;
        }
        s = bit_size(v);
        int sym = (run << 4) | s;
        bw_put(w, ac->code[sym], ac->size[sym]);
        bw_put(w, (uint32_t)(v < 0 ? v - 1 : v), s);
        run = 0;
    }
    if (run > 0) bw_put(w, ac->code[0], ac->size[0]);
}

/* FDCT + quantisation alone: zig-zag order */
static void quantize_block(const float *samples, const float *qdiv, int16_t zz[64]) {
    float d[64];
    memcpy(d, samples, sizeof d);
    fdct_aan(d);
    for (int k = 0; k < 64; k++) zz[k] = (int16_t)lrintf(d[k_zigzag_to_natural[k]] * qdiv[k_zigzag_to_natural[k]]);
}

/* restart bookkeeping shared by the progressive scans: called after every MCU of the scan */
static void prog_restart(bitwriter *bw, int dri, int *before, int *rst_index, int last, int *pred) {
    if (dri <= 0 || --*before != 0) return;
    if (!last) {
        bw_flush_ones(bw);
        if (bw->p + 2 <= bw->end) {
            *bw->p++ = 0xFF;
            *bw->p++ = (uint8_t)(0xD0 + (*rst_index & 7));
        } else
            bw->overflow = 1;
        (*rst_index)++;
        pred[0] = pred[1] = pred[2] = 0;
    }
    *before = dri;
}

static uint8_t *put_marker_seg(uint8_t *p, int marker, const uint8_t *payload, int n) {
    *p++ = 0xFF;
    *p++ = (uint8_t)marker;
    *p++ = (uint8_t)((n + 2) >> 8);
    *p++ = (uint8_t)((n + 2) & 0xFF);
    memcpy(p, payload, (size_t)n);
    return p + n;
}

/* Core encoder over a row source. get_rows fills `rows` MCU-row-worth of Y/Cb/Cr (wpad wide). */
long jsynth_encode(const jsynth_params *prm, uint8_t *out, size_t cap) {
    int w = prm->width, h = prm->height, ss = prm->subsampling;
    if (w <= 0 || h <= 0 || w > 65535 || h > 65535 || ss < 0 || ss > 3) return -1;
    int hmax = (ss == 1 || ss == 2) ? 2 : 1, vmax = (ss == 2) ? 2 : 1;
    int ncomp = ss == 3 ? 1 : 3;
    int ch[3] = {hmax, 1, 1}, cv[3] = {vmax, 1, 1};
    if (prm->samp[0][0] != 0) {
        hmax = vmax = 1;
        for (int c = 0; c < ncomp; c++) {
            ch[c] = prm->samp[c][0];
            cv[c] = prm->samp[c][1];
            if (ch[c] < 1 || ch[c] > 4 || cv[c] < 1 || cv[c] > 4) return -1;
            if (ch[c] > hmax) hmax = ch[c];
            if (cv[c] > vmax) vmax = cv[c];
        }
        for (int c = 0; c < ncomp; c++)
            if (hmax % ch[c] != 0 || vmax % cv[c] != 0) return -1;
    }
    int mcuw = 8 * hmax, mcuh = 8 * vmax;
    int mcus_x = (w + mcuw - 1) / mcuw, mcus_y = (h + mcuh - 1) / mcuh;
    int wpad = mcus_x * mcuw;
    if (cap < 1024) return -2;

    uint8_t qlum[64], qchr[64];
    scale_quant(k_std_lum, prm->quality, qlum);
    scale_quant(k_std_chr, prm->quality, qchr);
    static const double aan[8] = {1.0, 1.387039845, 1.306562965, 1.175875602, 1.0, 0.785694958, 0.541196100, 0.275899379};
    float qdiv_l[64], qdiv_c[64];
    for (int i = 0; i < 64; i++) {
        double sc = aan[i >> 3] * aan[i & 7] * 8.0;
        qdiv_l[i] = (float)(1.0 / (qlum[i] * sc));
        qdiv_c[i] = (float)(1.0 / (qchr[i] * sc));
    }
    enc_table dcl, dcc, acl, acc;
    build_enc_table(&dcl, k_dc_lum_bits, k_dc_vals);
    build_enc_table(&dcc, k_dc_chr_bits, k_dc_vals);
    build_enc_table(&acl, k_ac_lum_bits, k_ac_lum_vals);
    build_enc_table(&acc, k_ac_chr_bits, k_ac_chr_vals);

    uint8_t *p = out;
    *p++ = 0xFF;
    *p++ = 0xD8;
    static const uint8_t jfif[14] = {'J', 'F', 'I', 'F', 0, 1, 1, 0, 0, 1, 0, 1, 0, 0};
    p = put_marker_seg(p, 0xE0, jfif, 14);
    uint8_t seg[512];
    /* DQT: one segment per table (length 67 = 0x43) */
    seg[0] = 0x00;
    for (int i = 0; i < 64; i++) seg[1 + i] = qlum[k_zigzag_to_natural[i]];
    p = put_marker_seg(p, 0xDB, seg, 65);
    if (ncomp == 3) {
        seg[0] = 0x01;
        for (int i = 0; i < 64; i++) seg[1 + i] = qchr[k_zigzag_to_natural[i]];
        p = put_marker_seg(p, 0xDB, seg, 65);
    }
    /* SOF0 */
    seg[0] = 8;
    seg[1] = (uint8_t)(h >> 8);
    seg[2] = (uint8_t)h;
    seg[3] = (uint8_t)(w >> 8);
    seg[4] = (uint8_t)w;
    seg[5] = (uint8_t)ncomp;
    seg[6] = 1;
    seg[7] = (uint8_t)((ch[0] << 4) | cv[0]);
    seg[8] = 0;
    if (ncomp == 3) {
        seg[9] = 2;
        seg[10] = (uint8_t)((ch[1] << 4) | cv[1]);
        seg[11] = 1;
        seg[12] = 3;
        seg[13] = (uint8_t)((ch[2] << 4) | cv[2]);
        seg[14] = 1;
    }
    p = put_marker_seg(p, prm->progressive ? 0xC2 : 0xC0, seg, 6 + 3 * ncomp);
    /* DHT: all tables in one segment (4 tables -> length 418 = 0x01A2; gray: 2 tables -> 210 = 0xD2) */
    {
        uint8_t *q = seg;
        *q++ = 0x00;
        memcpy(q, k_dc_lum_bits, 16);
        q += 16;
        memcpy(q, k_dc_vals, 12);
        q += 12;
        *q++ = 0x10;
        memcpy(q, k_ac_lum_bits, 16);
        q += 16;
        memcpy(q, k_ac_lum_vals, 162);
        q += 162;
        if (ncomp == 3) {
            *q++ = 0x01;
            memcpy(q, k_dc_chr_bits, 16);
            q += 16;
            memcpy(q, k_dc_vals, 12);
            q += 12;
            *q++ = 0x11;
            memcpy(q, k_ac_chr_bits, 16);
            q += 16;
            memcpy(q, k_ac_chr_vals, 162);
            q += 162;
        }
        p = put_marker_seg(p, 0xC4, seg, (int)(q - seg));
    }
    /* DRI after SOF, like libjpeg-family encoders */
    if (prm->restart_interval > 0) {
        seg[0] = (uint8_t)(prm->restart_interval >> 8);
        seg[1] = (uint8_t)prm->restart_interval;
        p = put_marker_seg(p, 0xDD, seg, 2);
    }
    const int nscans = (prm->noninterleaved && ncomp == 3) ? 3 : 1;
    const int t81_order = prm->noninterleaved == 2 && nscans == 3;
    synth_ctx sc;
    synth_init(&sc, w, h, prm->seed);
    size_t rowsz = (size_t)wpad;
    float *Y = (float *)malloc(sizeof(float) * rowsz * mcuh * 3);
    float *Cb = Y + rowsz * mcuh, *Cr = Cb + rowsz * mcuh;
    bitwriter bw = {p, out + cap - 4, 0, 0, 0};
    float blk[64];
    if (prm->progressive) {
        /* quantised blocks of the whole frame, per component on its MCU-padded grid */
        int gw[3], gh[3];
        int16_t *store[3] = {NULL, NULL, NULL};
        for (int c = 0; c < ncomp; c++) {
            gw[c] = mcus_x * ch[c];
            gh[c] = mcus_y * cv[c];
            store[c] = (int16_t *)malloc((size_t)gw[c] * gh[c] * 64 * sizeof(int16_t));
        }
        for (int my = 0; my < mcus_y; my++) {
            for (int r = 0; r < mcuh; r++) synth_row(&sc, my * mcuh + r, wpad, Y + r * rowsz, Cb + r * rowsz, Cr + r * rowsz);
            for (int mx = 0; mx < mcus_x; mx++)
                for (int c = 0; c < ncomp; c++) {
                    const float *pl = c == 0 ? Y : (c == 1 ? Cb : Cr);
                    const int fx = hmax / ch[c], fy = vmax / cv[c];
                    for (int by = 0; by < cv[c]; by++)
                        for (int bx = 0; bx < ch[c]; bx++) {
                            const float *src = pl + (size_t)(by * 8 * fy) * rowsz + (size_t)mx * mcuw + bx * 8 * fx;
                            for (int i = 0; i < 8; i++)
                                for (int j = 0; j < 8; j++) {
                                    float acc_ = 0.f;
                                    for (int vy = 0; vy < fy; vy++)
                                        for (int vx = 0; vx < fx; vx++) acc_ += src[(size_t)(i * fy + vy) * rowsz + j * fx + vx];
                                    blk[i * 8 + j] = floorf(acc_ / (float)(fx * fy) + 0.5f) - 128.f;
                                }
                            quantize_block(blk, c == 0 ? qdiv_l : qdiv_c, store[c] + ((size_t)(my * cv[c] + by) * gw[c] + mx * ch[c] + bx) * 64);
                        }
                }
        }
        const int dri = prm->restart_interval;
        /* scans 0, 1: DC first (Al = 1) and DC refinement, every component interleaved */
        for (int pass = 0; pass < 2 && !bw.overflow; pass++) {
            uint8_t *hp = bw.p;
            seg[0] = (uint8_t)ncomp;
            for (int c = 0; c < ncomp; c++) {
                seg[1 + 2 * c] = (uint8_t)(c + 1);
                seg[2 + 2 * c] = c == 0 ? 0x00 : 0x11;
            }
            seg[1 + 2 * ncomp] = 0;
            seg[2 + 2 * ncomp] = 0;
            seg[3 + 2 * ncomp] = pass == 0 ? 0x01 : 0x10;
            bw.p = put_marker_seg(hp, 0xDA, seg, 4 + 2 * ncomp);
            bw.acc = 0;
            bw.nbits = 0;
            int pred[3] = {0, 0, 0}, before = dri, rst_index = 0;
            for (int my = 0; my < mcus_y; my++)
                for (int mx = 0; mx < mcus_x; mx++) {
                    for (int c = 0; c < ncomp; c++)
                        for (int by = 0; by < cv[c]; by++)
                            for (int bx = 0; bx < ch[c]; bx++) {
                                const int dcv = store[c][((size_t)(my * cv[c] + by) * gw[c] + mx * ch[c] + bx) * 64];
                                if (pass == 0) {
                                    const int t = dcv >> 1, diff = t - pred[c];  /* point transform: arithmetic shift (T.81 G.1.2.1) */
                                    pred[c] = t;
                                    const int sz = bit_size(diff);
                                    const enc_table *tdc = c == 0 ? &dcl : &dcc;
                                    bw_put(&bw, tdc->code[sz], tdc->size[sz]);
                                    if (sz) bw_put(&bw, (uint32_t)(diff < 0 ? diff - 1 : diff), sz);
                                } else
                                    bw_put(&bw, (uint32_t)(dcv & 1), 1);
                            }
                    prog_restart(&bw, dri, &before, &rst_index, my == mcus_y - 1 && mx == mcus_x - 1, pred);
                }
            bw_flush_ones(&bw);
        }
        /* AC bands per component, the component's own block raster: ceil(W / (8 hs)) x ceil(H / (8 vs)) blocks */
        static const int band[2][2] = {{1, 5}, {6, 63}};
        /* (the luma LAST: the reference's Dispose() transforms the components its decoder slots name when the file ends, and a
           single-component scan always takes slot 0 -- SURVEY 3.4-11; libjpeg's script ends with the luma too) */
        for (int cc = 0; cc < ncomp && !bw.overflow; cc++)
            for (int bnd = 0; bnd < 2 && !bw.overflow; bnd++) {
                const int c = (cc + 1) % ncomp;
                const int hsf = hmax / ch[c], vsf = vmax / cv[c];
                const int nbx = (w + 8 * hsf - 1) / (8 * hsf), nby = (h + 8 * vsf - 1) / (8 * vsf);
                uint8_t *hp = bw.p;
                seg[0] = 1;
                seg[1] = (uint8_t)(c + 1);
                seg[2] = c == 0 ? 0x00 : 0x11;
                seg[3] = (uint8_t)band[bnd][0];
                seg[4] = (uint8_t)band[bnd][1];
                seg[5] = 0;
                bw.p = put_marker_seg(hp, 0xDA, seg, 6);
                bw.acc = 0;
                bw.nbits = 0;
                const enc_table *tac = c == 0 ? &acl : &acc;
                int pred[3] = {0, 0, 0}, before = dri, rst_index = 0;
                for (int by = 0; by < nby; by++)
                    for (int bx = 0; bx < nbx; bx++) {
                        const int16_t *zz = store[c] + ((size_t)by * gw[c] + bx) * 64;
                        int run = 0;
                        for (int k = band[bnd][0]; k <= band[bnd][1]; k++) {
                            const int v = zz[k];
                            if (v == 0) {
                                run++;
                                continue;
                            }
                            while (run > 15) {
                                bw_put(&bw, tac->code[0xF0], tac->size[0xF0]);
                                run -= 16;
                            }
                            const int sz = bit_size(v);
                            bw_put(&bw, tac->code[(run << 4) | sz], tac->size[(run << 4) | sz]);
                            bw_put(&bw, (uint32_t)(v < 0 ? v - 1 : v), sz);
                            run = 0;
                        }
                        if (run > 0) bw_put(&bw, tac->code[0], tac->size[0]);
                        prog_restart(&bw, dri, &before, &rst_index, by == nby - 1 && bx == nbx - 1, pred);
                    }
                bw_flush_ones(&bw);
            }
        for (int c = 0; c < ncomp; c++) free(store[c]);
        free(Y);
        synth_free(&sc);
        if (bw.overflow) return -2;
        p = bw.p;
        *p++ = 0xFF;
        *p++ = 0xD9;
        return (long)(p - out);
    }
    for (int scan = 0; scan < nscans && !bw.overflow; scan++) {
        /* SOS */
        uint8_t *hp = bw.p;
        if (nscans == 1) {
            seg[0] = (uint8_t)ncomp;
            seg[1] = 1;
            seg[2] = 0x00;
            if (ncomp == 3) {
                seg[3] = 2;
                seg[4] = 0x11;
                seg[5] = 3;
                seg[6] = 0x11;
            }
            seg[1 + 2 * ncomp] = 0;
            seg[2 + 2 * ncomp] = 63;
            seg[3 + 2 * ncomp] = 0;
            hp = put_marker_seg(hp, 0xDA, seg, 4 + 2 * ncomp);
        } else {
            seg[0] = 1;
            seg[1] = (uint8_t)(scan + 1);
            seg[2] = scan == 0 ? 0x00 : 0x11;
            seg[3] = 0;
            seg[4] = 63;
            seg[5] = 0;
            hp = put_marker_seg(hp, 0xDA, seg, 6);
        }
        bw.p = hp;
        bw.acc = 0;
        bw.nbits = 0;
        int pred[3] = {0, 0, 0};
        int mcus_before_restart = prm->restart_interval, rst_index = 0;
        /* T.81's non-interleaved order: an MCU is ONE block, the scan walks the component's own block raster (A.2.3) */
        const int sc_c = nscans == 3 ? scan : 0;
        const int t81_bx = t81_order ? ((w * ch[sc_c] + hmax - 1) / hmax + 7) / 8 : 0;
        const int t81_by = t81_order ? ((h * cv[sc_c] + vmax - 1) / vmax + 7) / 8 : 0;
        const long n_units = t81_order ? (long)t81_bx * t81_by : (long)mcus_x * mcus_y;
        long unit = 0;
        for (int my = 0; my < mcus_y && !bw.overflow; my++) {
            for (int r = 0; r < mcuh; r++) synth_row(&sc, my * mcuh + r, wpad, Y + r * rowsz, Cb + r * rowsz, Cr + r * rowsz);
            for (int mx = 0; mx < mcus_x; mx++) {
                for (int c = 0; c < ncomp; c++) {
                    if (nscans == 3 && scan != c) continue;
                    const float *pl = c == 0 ? Y : (c == 1 ? Cb : Cr);
                    const int fx = hmax / ch[c], fy = vmax / cv[c];
                    const float *ql = c == 0 ? qdiv_l : qdiv_c;
                    const enc_table *tdc = c == 0 ? &dcl : &dcc, *tac = c == 0 ? &acl : &acc;
                    for (int by = 0; by < cv[c]; by++)
                        for (int bx = 0; bx < ch[c]; bx++) {
                            if (t81_order && (mx * ch[c] + bx >= t81_bx || my * cv[c] + by >= t81_by)) continue;
                            const float *src = pl + (size_t)(by * 8 * fy) * rowsz + (size_t)mx * mcuw + bx * 8 * fx;
                            for (int i = 0; i < 8; i++)
                                for (int j = 0; j < 8; j++) {
                                    if (fx == 1 && fy == 1) {
                                        blk[i * 8 + j] = src[i * rowsz + j] - 128.f;
                                        continue;
                                    }
                                    float acc_ = 0.f;
                                    for (int vy = 0; vy < fy; vy++)
                                        for (int vx = 0; vx < fx; vx++) acc_ += src[(size_t)(i * fy + vy) * rowsz + j * fx + vx];
                                    float v = acc_ / (float)(fx * fy);
                                    blk[i * 8 + j] = floorf(v + 0.5f) - 128.f;
                                }
                            /* (T.81 order with more than one block row per MCU row would need the blocks of a whole MCU row
                               reordered: block rows of one MCU row are emitted MCU by MCU here, which is T.81's order only
                               for V = 1 -- good enough for a file the reference misreads anyway) */
                            encode_block(&bw, blk, ql, tdc, tac, &pred[c]);
                            if (t81_order) {
                                unit++;
                                if (prm->restart_interval > 0 && --mcus_before_restart == 0) {
                                    if (unit != n_units) {
                                        bw_flush_ones(&bw);
                                        if (bw.p + 2 <= bw.end) {
                                            *bw.p++ = 0xFF;
                                            *bw.p++ = (uint8_t)(0xD0 + (rst_index & 7));
                                        } else
                                            bw.overflow = 1;
                                        rst_index++;
                                        pred[0] = pred[1] = pred[2] = 0;
                                    }
                                    mcus_before_restart = prm->restart_interval;
                                }
                            }
                        }
                }
                if (!t81_order && prm->restart_interval > 0 && --mcus_before_restart == 0) {
                    int last = (my == mcus_y - 1 && mx == mcus_x - 1);
                    if (!last) {
                        bw_flush_ones(&bw);
                        if (bw.p + 2 <= bw.end) {
                            *bw.p++ = 0xFF;
                            *bw.p++ = (uint8_t)(0xD0 + (rst_index & 7));
                        } else
                            bw.overflow = 1;
                        rst_index++;
                        pred[0] = pred[1] = pred[2] = 0;
                    }
                    mcus_before_restart = prm->restart_interval;
                }
            }
        }
        bw_flush_ones(&bw);
    }
    free(Y);
    synth_free(&sc);
    if (bw.overflow) return -2;
    p = bw.p;
    *p++ = 0xFF;
    *p++ = 0xD9;
    return (long)(p - out);
}

typedef struct batch_job {
    const jsynth_params *prm;
    int n;
    uint8_t *out;
    size_t stride;
    long *sizes;
    volatile int *next;
} batch_job;

static void *batch_worker(void *arg) {
    batch_job *j = (batch_job *)arg;
    for (;;) {
        int i = __sync_fetch_and_add(j->next, 1);
        if (i >= j->n) break;
        j->sizes[i] = jsynth_encode(&j->prm[i], j->out + (size_t)i * j->stride, j->stride);
    }
    return NULL;
}

/* Encodes n images into out[i*stride ...]; sizes[i] = bytes or <0. Returns 0 when all succeeded. */
int jsynth_encode_batch(const jsynth_params *prm, int n, uint8_t *out, size_t stride, long *sizes, int nthreads) {
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 256) nthreads = 256;
    volatile int next = 0;
    batch_job job = {prm, n, out, stride, sizes, &next};
    pthread_t th[256];
    for (int t = 0; t < nthreads; t++) pthread_create(&th[t], NULL, batch_worker, &job);
    for (int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
    for (int i = 0; i < n; i++)
        if (sizes[i] < 0) return -1;
    return 0;
}
