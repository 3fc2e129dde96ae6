/*
 * oracle/jpegref.c -- CPU restatement ("oracle") of yigolden/JpegLibrary's Huffman-DCT decode path.
 *
 * TEST INFRASTRUCTURE ONLY (see jpegref.h).  Scalar C99, one block at a time, like the reference.
 * Build with -O2 -ffp-contract=off -fno-fast-math: every float op below must be one IEEE-754 binary32
 * operation (the reference's System.Numerics.Vector4 lanes never fuse multiply-add).
 *
 * Parity pin: 0 mismatching samples against the reference's golden PNG dumps for cramps.jpg, lake.jpg,
 * testorig12.jpg, progress.jpg and yellowcat_progressive_restart.jpg (tests/test_oracle_golden.py).
 * Unpinned by reference tests (no asset): baseline with DRI, 4:4:4 / 4:2:2 baseline, malformed streams.
 *
 * Citations "ref:" are relative to /root/reference/src/JpegLibrary.
 */
#include "jpegref.h"

#include <math.h>
#include <setjmp.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------------
 * small helpers
 * ---------------------------------------------------------------------------------------------- */

/* C# int shifts mask the count to 5 bits. */
#define SHL32(v, n) ((int32_t)((uint32_t)(v) << ((n)&31)))
#define SAR32(v, n) ((int32_t)(v) >> ((n)&31))

/* ref: JpegZigZag.cs:27-38 (s_bufferToBlock): zig-zag index -> natural (row-major) index. */
static const uint8_t k_buffer_to_block[64] = {
    0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
    41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
    30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

typedef struct huff_table {
    int present;
    uint8_t table_class, identifier;
    uint8_t values[256];
    uint16_t maxcode[18];
    uint8_t valoffset[19];
    uint8_t la_size[256], la_symbol[256];
} huff_table;

typedef struct quant_table {
    int present; /* !IsEmpty */
    uint8_t precision, identifier;
    uint16_t elements[64]; /* zig-zag order, ref: JpegQuantizationTable.cs:47 */
} quant_table;

typedef struct frame_header {
    int present;
    uint8_t precision;
    uint16_t lines, samples_per_line;
    uint8_t ncomp;
    int has_components;
    jref_component comp[256];
} frame_header;

typedef struct scan_component {
    uint8_t selector, td, ta;
} scan_component;

typedef struct scan_header {
    uint8_t ncomp;
    scan_component comp[256];
    uint8_t ss, se, ah, al;
} scan_header;

/* ref: JpegReader.cs -- a cursor over the input. */
typedef struct reader {
    const uint8_t *p;
    size_t n;
    size_t initial;
} reader;

/* ref: JpegHuffmanDecodingComponent.cs:5-15 */
typedef struct decoding_component {
    int component_index;
    uint8_t h, v;
    int dc_predictor;
    const huff_table *dc_table, *ac_table;
    const quant_table *quant; /* NULL == IsEmpty */
    int hs, vs;
} decoding_component;

/* ref: JpegBlockAllocator.cs:204-211 */
typedef struct component_allocation {
    int hs, vs, hblocks, vblocks, offset;
} component_allocation;

enum { SCAN_NONE = 0, SCAN_BASELINE, SCAN_PROGRESSIVE, SCAN_UNSUPPORTED };

struct jref_decoder {
    /* JpegDecoder fields, ref: JpegDecoder.cs:22-43 */
    const uint8_t *input;
    size_t input_len;
    frame_header frame;
    int restart_interval;
    int start_of_frame;
    /* table registries keep insertion order with replace-by-key, ref: JpegDecoder.cs:793-861 */
    quant_table qt[16];
    int nqt;
    huff_table ht[32];
    int nht;
    jref_write_block_fn writer;
    void *writer_user;
    jref_coef_tap_fn tap;
    void *tap_user;
    jref_progressive_tap_fn ptap;
    void *ptap_user;

    /* scan decoder state (baseline + progressive), ref: ScanDecoder/ *.cs */
    int scan_kind;
    frame_header sd_frame;
    int sd_max_h, sd_max_v;
    int sd_restart_interval; /* baseline: latched in ctor, ref: JpegHuffmanBaselineScanDecoder.cs:38 */
    int sd_mcus_per_line, sd_mcus_per_column, sd_level_shift;
    decoding_component sd_components[256];
    int sd_ncomponents_alloc;
    long sd_block_counter;
    /* progressive */
    int pg_restart_interval, pg_mcus_before_restart, pg_eobrun;
    component_allocation *pg_alloc;
    int pg_nalloc;
    int16_t *pg_blocks; /* [nblocks][64] */
    jref_write_block_fn pg_writer;
    void *pg_writer_user;

    /* exception emulation */
    jmp_buf jb;
    int jb_armed;
    int err_code;
    char err[256];
};

static void throw_err(jref_decoder *d, int code, const char *fmt, ...)
#if defined(__GNUC__)
    __attribute__((noreturn, format(printf, 3, 4)))
#endif
    ;

#include <stdarg.h>
static void throw_err(jref_decoder *d, int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(d->err, sizeof d->err, fmt, ap);
    va_end(ap);
    d->err_code = code;
    if (!d->jb_armed) {
        fprintf(stderr, "jpegref: unhandled error: %s\n", d->err);
        abort();
    }
    longjmp(d->jb, 1);
}

/* ref: ScanDecoder/JpegScanDecoder.cs:39-42 */
#define THROW_SCAN_INVALID(d, msg) throw_err((d), JREF_INVALID_DATA, "Failed to decode JPEG data. %s", (msg))
/* ref: JpegDecoder.cs:371-375 */
#define THROW_AT(d, off, msg) \
    throw_err((d), JREF_INVALID_DATA, "Failed to decode JPEG data at offset %d. %s", (int)(off), (msg))

/* ------------------------------------------------------------------------------------------------
 * JpegReader (ref: JpegReader.cs)
 * ---------------------------------------------------------------------------------------------- */

static int rd_consumed(const reader *r) { return (int)(r->initial - r->n); } /* ref: :51 */

/* ref: JpegReader.cs:98-112 */
static int rd_try_read_soi(reader *r) {
    if (r->n < 2) return 0;
    if (r->p[0] == 0xFF && r->p[1] == 0xD8) {
        r->p += 2;
        r->n -= 2;
        return 1;
    }
    return 0;
}

/* ref: JpegReader.cs:120-158 */
static int rd_try_read_marker(reader *r, int *marker) {
    while (r->n >= 2) {
        uint8_t b1 = r->p[0], b2 = r->p[1];
        if (b1 == 0xFF) {
            if (b2 == 0xFF) {
                r->p += 1;
                r->n -= 1;
                continue;
            } else if (b2 == 0) {
                r->p += 2;
                r->n -= 2;
                continue;
            }
            r->p += 2;
            r->n -= 2;
            *marker = b2;
            return 1;
        }
        const uint8_t *q = (const uint8_t *)memchr(r->p, 0xFF, r->n);
        if (!q) {
            r->p += r->n;
            r->n = 0;
            *marker = 0;
            return 0;
        }
        r->n -= (size_t)(q - r->p);
        r->p = q;
    }
    *marker = 0;
    return 0;
}

/* ref: JpegReader.cs:166-177.  NOTE the reference's expression `(ushort)(b0 << 8 | b1 - 2)` binds as
 * (b0 << 8) | (b1 - 2): it is only the true payload length when the low length byte is >= 2. Restated as is. */
static int rd_try_read_length(reader *r, uint16_t *length) {
    if (r->n < 2) {
        *length = 0;
        return 0;
    }
    int32_t v = ((int32_t)r->p[0] << 8) | ((int32_t)r->p[1] - 2);
    *length = (uint16_t)v;
    r->p += 2;
    r->n -= 2;
    return 1;
}

/* ref: JpegReader.cs:203-214 */
static int rd_try_read_bytes(reader *r, int length, const uint8_t **bytes) {
    if (r->n < (size_t)length) return 0;
    *bytes = r->p;
    r->p += length;
    r->n -= (size_t)length;
    return 1;
}

/* ref: JpegReader.cs:239-247 */
static int rd_try_advance(reader *r, int length) {
    if (length < 0) { /* ReadOnlySequence.Slice(negative) throws; treat as failure */
        return 0;
    }
    if (r->n < (size_t)length) return 0;
    r->p += length;
    r->n -= (size_t)length;
    return 1;
}

/* ------------------------------------------------------------------------------------------------
 * Huffman decoding table (ref: JpegHuffmanDecodingTable.cs)
 * ---------------------------------------------------------------------------------------------- */

/* ref: :293-309 */
static int gen_size_table(const uint8_t bits[16], uint8_t huffsize[257]) {
    int k = 0;
    for (int i = 1; i <= 16; i++) {
        int j = 1;
        while (j++ <= bits[i - 1]) huffsize[k++] = (uint8_t)i;
    }
    huffsize[k] = 0;
    return k;
}

/* ref: :311-337 */
static void gen_code_table(const uint8_t huffsize[257], uint16_t huffcode[257]) {
    int k = 0, code = 0, si = huffsize[0];
    if (si == 0) return; /* empty table: the reference reads uninitialised stack here; fenced off */
    for (;;) {
        do {
            huffcode[k] = (uint16_t)code;
            code++;
            k++;
        } while (huffsize[k] == si);
        if (huffsize[k] == 0) return;
        do {
            code <<= 1;
            si++;
        } while (huffsize[k] != si);
    }
}

/* ref: :339-390 (Configure + FillByteLookupTable) */
static void huff_configure(huff_table *t, const uint8_t bits[16], const uint16_t huffcode[257], const uint8_t *values,
                           int nvalues) {
    memset(t->values, 0, sizeof t->values);
    memset(t->maxcode, 0, sizeof t->maxcode);
    memset(t->valoffset, 0, sizeof t->valoffset);
    memset(t->la_size, 0, sizeof t->la_size);
    memset(t->la_symbol, 0, sizeof t->la_symbol);
    memcpy(t->values, values, (size_t)nvalues);

    int p = 0;
    for (int l = 1; l <= 16; l++) {
        if (bits[l - 1] != 0) {
            int offset = p - huffcode[p];
            t->valoffset[l] = (uint8_t)offset;
            p += bits[l - 1];
            t->maxcode[l] = huffcode[p - 1];
            t->maxcode[l] = (uint16_t)(t->maxcode[l] << (16 - l));
            t->maxcode[l] = (uint16_t)(t->maxcode[l] | (uint32_t)((1 << (16 - l)) - 1));
        } else {
            t->maxcode[l] = 0;
        }
    }
    t->valoffset[18] = 0;
    t->maxcode[17] = 0xFFFF;

    p = 0;
    for (int l = 1; l <= 8; l++) {
        for (int i = 0; i < bits[l - 1]; i++, p++) {
            int free_bits = 8 - l;
            int code = (uint8_t)(huffcode[p] << free_bits);
            for (int j = 0; j < (1 << free_bits); j++) {
                if (code + j > 255) break; /* IndexOutOfRange in the reference; oversubscribed table */
                t->la_size[code + j] = (uint8_t)l;
                t->la_symbol[code + j] = t->values[p];
            }
        }
    }
}

/* ref: :249-291.  buffer excludes the Tc/Th byte. */
static int huff_try_parse(huff_table *t, uint8_t table_class, uint8_t identifier, const uint8_t *buf, size_t len,
                          int *bytes_consumed) {
    if (len < 16) return 0;
    int code_count = 0;
    for (int i = 15; i >= 0; i--) code_count += buf[i];
    if (code_count > 256) return 0;
    uint8_t huffsize[257];
    uint16_t huffcode[257];
    memset(huffsize, 0, sizeof huffsize);
    memset(huffcode, 0, sizeof huffcode);
    gen_size_table(buf, huffsize);
    *bytes_consumed += 16;
    if (len - 16 < (size_t)code_count) return 0;
    gen_code_table(huffsize, huffcode);
    *bytes_consumed += code_count;
    t->present = 1;
    t->table_class = table_class;
    t->identifier = identifier;
    huff_configure(t, buf, huffcode, buf + 16, code_count);
    return 1;
}

int jref_build_huffman(const uint8_t bits[16], const uint8_t *values, int nvalues, uint8_t lookahead_size[256],
                       uint8_t lookahead_symbol[256], uint16_t maxcode[18], uint8_t valoffset[19],
                       uint8_t values_out[256]) {
    uint8_t buf[16 + 256];
    if (nvalues < 0 || nvalues > 256) return 0;
    memcpy(buf, bits, 16);
    memcpy(buf + 16, values, (size_t)nvalues);
    huff_table t;
    int consumed = 0;
    if (!huff_try_parse(&t, 0, 0, buf, (size_t)(16 + nvalues), &consumed)) return 0;
    memcpy(lookahead_size, t.la_size, 256);
    memcpy(lookahead_symbol, t.la_symbol, 256);
    memcpy(maxcode, t.maxcode, sizeof t.maxcode);
    memcpy(valoffset, t.valoffset, sizeof t.valoffset);
    memcpy(values_out, t.values, 256);
    return 1;
}

/* ref: :73-113 (Lookup + LookupSlow) */
static void huff_lookup(jref_decoder *d, const huff_table *t, int code16, int *size, int *symbol) {
    int high8 = code16 >> 8;
    if (t->la_size[high8] != 0) {
        *size = t->la_size[high8];
        *symbol = t->la_symbol[high8];
        return;
    }
    int s = 9;
    while (code16 > t->maxcode[s]) s++;
    if (s > 16) throw_err(d, JREF_INVALID_DATA, "Invalid Huffman code encountered.");
    code16 >>= (16 - s);
    *size = s;
    *symbol = t->values[(t->valoffset[s] + code16) & 0xFF];
}

/* ------------------------------------------------------------------------------------------------
 * JpegBitReader (ref: JpegBitReader.cs)
 * ---------------------------------------------------------------------------------------------- */

typedef struct bit_reader {
    const uint8_t *p;
    size_t n;
    uint64_t buffer; /* right-justified */
    uint8_t bits_in_buffer;
    int next_marker;
} bit_reader;

static void br_init(bit_reader *b, const uint8_t *p, size_t n) {
    b->p = p;
    b->n = n;
    b->buffer = 0;
    b->bits_in_buffer = 0;
    b->next_marker = 0;
}

static int br_remaining_bits(const bit_reader *b) { return 8 * (int)b->n + b->bits_in_buffer; } /* ref: :27 */

/* ref: :95-138 */
static int br_fill_buffer(bit_reader *b) {
    while (b->bits_in_buffer < 32) {
        if (b->next_marker != 0) return b->bits_in_buffer;
        if (b->n == 0) break;
        uint8_t byte = *b->p++;
        b->n--;
        if (byte == 0xFF) {
            if (b->n == 0) break; /* the stream ended prematurely */
            byte = *b->p;         /* peek */
            if (byte == 0xFF) continue; /* padding byte */
            b->p++;
            b->n--;
            if (byte != 0) {
                b->next_marker = byte;
                break;
            }
            byte = 0xFF; /* stuffed byte */
        }
        b->buffer = (b->buffer << 8) | byte;
        b->bits_in_buffer = (uint8_t)(b->bits_in_buffer + 8);
    }
    return b->bits_in_buffer;
}

/* ref: :29-33 */
static void br_advance_align_byte(bit_reader *b) {
    b->bits_in_buffer = (uint8_t)(b->bits_in_buffer - (b->bits_in_buffer % 8));
    br_fill_buffer(b);
}

/* ref: :140-149 */
static int br_try_read_marker(bit_reader *b) {
    if (b->bits_in_buffer == 0) {
        int m = b->next_marker;
        b->next_marker = 0;
        return m;
    }
    return 0;
}

/* ref: :151-154 */
static int br_try_peek_marker(const bit_reader *b) { return b->bits_in_buffer == 0 ? b->next_marker : 0; }

/* ref: :157-172 */
static int br_peek_bits(bit_reader *b, int length, int *bits_peeked) {
    int bits_in_buffer = b->bits_in_buffer;
    if (bits_in_buffer < length) {
        bits_in_buffer = br_fill_buffer(b);
        if (bits_in_buffer < length) {
            *bits_peeked = bits_in_buffer;
            return (SHL32((int32_t)(uint32_t)b->buffer, length - bits_in_buffer) & (SHL32(1, length) - 1)) |
                   (SHL32(1, length - bits_in_buffer) - 1);
        }
    }
    int remaining = bits_in_buffer - length;
    *bits_peeked = length;
    return (int32_t)(uint32_t)(b->buffer >> (remaining & 63)) & (SHL32(1, length) - 1);
}

/* ref: :207-218 */
static int br_try_load_bits(bit_reader *b, int length, int *is_marker) {
    int bits = br_fill_buffer(b);
    if (bits < length) {
        *is_marker = (bits == 0 && b->next_marker != 0);
        return 0;
    }
    *is_marker = 0;
    return 1;
}

/* ref: :175-187 */
static int br_try_advance_bits(bit_reader *b, int length, int *is_marker) {
    if (b->bits_in_buffer < length) {
        if (!br_try_load_bits(b, length, is_marker)) return 0;
    }
    b->bits_in_buffer = (uint8_t)(b->bits_in_buffer - length);
    *is_marker = 0;
    return 1;
}

/* ref: :190-204 */
static int br_try_read_bits(bit_reader *b, int length, int *bits, int *is_marker) {
    if (b->bits_in_buffer < length) {
        if (!br_try_load_bits(b, length, is_marker)) {
            *bits = 0;
            return 0;
        }
    }
    b->bits_in_buffer = (uint8_t)(b->bits_in_buffer - length);
    *bits = (int32_t)(uint32_t)(b->buffer >> (b->bits_in_buffer & 63)) & (SHL32(1, length) - 1);
    *is_marker = 0;
    return 1;
}

/* ------------------------------------------------------------------------------------------------
 * JpegHuffmanScanDecoder helpers (ref: ScanDecoder/JpegHuffmanScanDecoder.cs)
 * ---------------------------------------------------------------------------------------------- */

/* ref: :81-88 */
static int decode_huffman_code(jref_decoder *d, bit_reader *b, const huff_table *t) {
    int bits_read;
    int bits = br_peek_bits(b, 16, &bits_read);
    int size, symbol;
    huff_lookup(d, t, bits, &size, &symbol);
    if (size < bits_read) bits_read = size;
    int dummy;
    (void)br_try_advance_bits(b, bits_read, &dummy);
    return symbol;
}

/* ref: :100-115 */
static int receive_and_extend(jref_decoder *d, bit_reader *b, int length) {
    int value, is_marker;
    if (!br_try_read_bits(b, length, &value, &is_marker)) {
        if (is_marker) THROW_SCAN_INVALID(d, "Expect raw data from bit stream. Yet a marker is encountered.");
        THROW_SCAN_INVALID(d, "The bit stream ended prematurely.");
    }
    /* Extend(v, nbits) */
    return value - ((SAR32(value + value, length) - 1) & (SHL32(1, length) - 1));
}

/* ------------------------------------------------------------------------------------------------
 * Block math (ref: ScanDecoder/JpegScanDecoder.cs:50-73, FastFloatingPointDCT.cs:19-185,
 *             JpegBlock8x8F.cs:163-181,228-301, JpegMathHelper.cs:13-20)
 * ---------------------------------------------------------------------------------------------- */

/* ref: ScanDecoder/JpegScanDecoder.cs:50-62 */
static void dequantize_and_unzigzag(const uint16_t *q, const int16_t *in, float *out) {
    for (int i = 0; i < 64; i++) {
        int32_t prod = (int32_t)q[i] * (int32_t)in[i];
        out[k_buffer_to_block[i]] = (float)prod;
    }
}

/* ref: JpegBlock8x8F.cs:228-301 */
static void transpose_into(const float *s, float *d) {
    for (int r = 0; r < 8; r++)
        for (int c = 0; c < 8; c++) d[c * 8 + r] = s[r * 8 + c];
}

/* ref: FastFloatingPointDCT.cs:79-127 / :137-185: the butterfly, applied to `lanes` columns starting at col0.
 * s and d are 8x8 row-major; V{k}L = row k cols 0..3, V{k}R = row k cols 4..7. */
static void idct8x4(const float *s, float *d, int col0) {
    static const float C_1_175876 = 1.175875602f, C_1_961571 = -1.961570560f, C_0_390181 = -0.390180644f,
                       C_0_899976 = -0.899976223f, C_2_562915 = -2.562915447f, C_0_298631 = 0.298631336f,
                       C_2_053120 = 2.053119869f, C_3_072711 = 3.072711026f, C_1_501321 = 1.501321110f,
                       C_0_541196 = 0.541196100f, C_1_847759 = -1.847759065f, C_0_765367 = 0.765366865f;
    for (int c = col0; c < col0 + 4; c++) {
        float my1 = s[1 * 8 + c];
        float my7 = s[7 * 8 + c];
        float mz0 = my1 + my7;

        float my3 = s[3 * 8 + c];
        float mz2 = my3 + my7;
        float my5 = s[5 * 8 + c];
        float mz1 = my3 + my5;
        float mz3 = my1 + my5;

        float mz4 = (mz0 + mz1) * C_1_175876;

        mz2 = (mz2 * C_1_961571) + mz4;
        mz3 = (mz3 * C_0_390181) + mz4;
        mz0 = mz0 * C_0_899976;
        mz1 = mz1 * C_2_562915;

        float mb3 = ((my7 * C_0_298631) + mz0) + mz2;
        float mb2 = ((my5 * C_2_053120) + mz1) + mz3;
        float mb1 = ((my3 * C_3_072711) + mz1) + mz2;
        float mb0 = ((my1 * C_1_501321) + mz0) + mz3;

        float my2 = s[2 * 8 + c];
        float my6 = s[6 * 8 + c];
        mz4 = (my2 + my6) * C_0_541196;
        float my0 = s[0 * 8 + c];
        float my4 = s[4 * 8 + c];
        mz0 = my0 + my4;
        mz1 = my0 - my4;

        mz2 = mz4 + (my6 * C_1_847759);
        mz3 = mz4 + (my2 * C_0_765367);

        my0 = mz0 + mz3;
        my3 = mz0 - mz3;
        my1 = mz1 + mz2;
        my2 = mz1 - mz2;

        d[0 * 8 + c] = my0 + mb0;
        d[7 * 8 + c] = my0 - mb0;
        d[1 * 8 + c] = my1 + mb1;
        d[6 * 8 + c] = my1 - mb1;
        d[2 * 8 + c] = my2 + mb2;
        d[5 * 8 + c] = my2 - mb2;
        d[3 * 8 + c] = my3 + mb3;
        d[4 * 8 + c] = my3 - mb3;
    }
}

/* ref: FastFloatingPointDCT.cs:54-70 */
static void transform_idct(const float *src, float *dest, float *temp) {
    transpose_into(src, temp);
    idct8x4(temp, dest, 0);
    idct8x4(temp, dest, 4);
    transpose_into(dest, temp);
    idct8x4(temp, dest, 0);
    idct8x4(temp, dest, 4);
    for (int i = 0; i < 64; i++) dest[i] = dest[i] * 0.1250f;
}

/* ref: ScanDecoder/JpegScanDecoder.cs:64-73 + JpegMathHelper.cs:13-20: MathF.Round = round-half-to-even. */
static void shift_data_level(const float *src, int16_t *dst, int level_shift) {
    for (int i = 0; i < 64; i++) {
        int32_t r = (int32_t)rintf(src[i]); /* default rounding mode: to nearest, ties to even */
        dst[i] = (int16_t)(r + level_shift);
    }
}

void jref_block_dequant_idct_shift(const int16_t *zigzag_coefs, const uint16_t *quant_zigzag, int level_shift,
                                   int16_t *out64) {
    float f[64], o[64], t[64];
    dequantize_and_unzigzag(quant_zigzag, zigzag_coefs, f);
    transform_idct(f, o, t);
    shift_data_level(o, out64, level_shift);
}

static int log2u(unsigned v) { /* ref: JpegMathHelper.cs:54-61 (BitOperations.Log2; 0 -> 0) */
    int r = 0;
    while (v >>= 1) r++;
    return r;
}

/* ref: JpegHuffmanBaselineScanDecoder.cs:225-268 and JpegBlockAllocator.cs:151-190 (identical bodies) */
static void write_block_expanded(jref_write_block_fn fn, void *user, const int16_t *block, int component_index, int x,
                                 int y, int hs, int vs) {
    if (hs == 1 && vs == 1) {
        fn(user, block, component_index, x, y);
        return;
    }
    int16_t temp[64];
    int hshift = log2u((unsigned)hs), vshift = log2u((unsigned)vs);
    for (int v = 0; v < vs; v++) {
        for (int h = 0; h < hs; h++) {
            int vblock = 8 * v, hblock = 8 * h;
            for (int i = 0; i < 8; i++) {
                const int16_t *row = block + ((vblock + i) >> vshift) * 8;
                for (int j = 0; j < 8; j++) temp[8 * i + j] = row[(hblock + j) >> hshift];
            }
            fn(user, temp, component_index, x + 8 * h, y + 8 * v);
        }
    }
}

/* ------------------------------------------------------------------------------------------------
 * Table registry (ref: JpegDecoder.cs:793-925)
 * ---------------------------------------------------------------------------------------------- */

static void set_huffman_table(jref_decoder *d, const huff_table *t) {
    for (int i = 0; i < d->nht; i++) {
        if (d->ht[i].table_class == t->table_class && d->ht[i].identifier == t->identifier) {
            d->ht[i] = *t;
            return;
        }
    }
    if (d->nht < 32) d->ht[d->nht++] = *t;
}

static const huff_table *get_huffman_table(const jref_decoder *d, int is_dc, uint8_t identifier) {
    int table_class = is_dc ? 0 : 1;
    for (int i = 0; i < d->nht; i++)
        if (d->ht[i].table_class == table_class && d->ht[i].identifier == identifier) return &d->ht[i];
    return NULL;
}

static void set_quant_table(jref_decoder *d, const quant_table *t) {
    for (int i = 0; i < d->nqt; i++) {
        if (d->qt[i].identifier == t->identifier) {
            d->qt[i] = *t;
            return;
        }
    }
    if (d->nqt < 16) d->qt[d->nqt++] = *t;
}

static const quant_table *get_quant_table(const jref_decoder *d, uint8_t identifier) {
    for (int i = 0; i < d->nqt; i++)
        if (d->qt[i].identifier == identifier) return &d->qt[i];
    return NULL;
}

/* ------------------------------------------------------------------------------------------------
 * Segment processing (ref: JpegDecoder.cs:251-311, 635-763)
 * ---------------------------------------------------------------------------------------------- */

static void process_other_marker(jref_decoder *d, reader *r) { /* ref: :251-263 */
    uint16_t length;
    if (!rd_try_read_length(r, &length))
        THROW_AT(d, rd_consumed(r), "Unexpected end of input data when reading segment length.");
    if (!rd_try_advance(r, length)) THROW_AT(d, rd_consumed(r), "Unexpected end of input data reached.");
}

/* ref: JpegFrameHeader.cs:137-182 */
static int frame_try_parse(const uint8_t *buf, size_t len, int metadata_only, frame_header *fh, int *consumed) {
    *consumed = 0;
    if (len < 6) return 0;
    uint8_t ncomp = buf[5];
    uint16_t spl = (uint16_t)(buf[4] | (buf[3] << 8));
    uint16_t lines = (uint16_t)(buf[2] | (buf[1] << 8));
    uint8_t precision = buf[0];
    buf += 6;
    len -= 6;
    *consumed += 6;
    if (len < (size_t)(3 * ncomp)) return 0;
    memset(fh, 0, sizeof *fh);
    fh->present = 1;
    fh->precision = precision;
    fh->lines = lines;
    fh->samples_per_line = spl;
    fh->ncomp = ncomp;
    if (metadata_only) {
        *consumed += 3 * ncomp;
        fh->has_components = 0;
        return 1;
    }
    fh->has_components = 1;
    for (int i = 0; i < ncomp; i++) {
        fh->comp[i].identifier = buf[0];
        fh->comp[i].h = (uint8_t)(buf[1] >> 4);
        fh->comp[i].v = (uint8_t)(buf[1] & 0xf);
        fh->comp[i].tq = buf[2];
        buf += 3;
        *consumed += 3;
    }
    return 1;
}

static void process_frame_header(jref_decoder *d, reader *r, int metadata_only, int override_allowed) { /* ref: :265-289 */
    uint16_t length;
    const uint8_t *buf;
    if (!rd_try_read_length(r, &length))
        THROW_AT(d, rd_consumed(r), "Unexpected end of input data when reading segment length.");
    if (!rd_try_read_bytes(r, length, &buf))
        THROW_AT(d, rd_consumed(r), "Unexpected end of input data when reading segment content.");
    frame_header fh;
    int consumed;
    if (!frame_try_parse(buf, length, metadata_only, &fh, &consumed))
        THROW_AT(d, rd_consumed(r) - length + consumed, "Failed to parse frame header.");
    if (!override_allowed && d->frame.present) THROW_AT(d, rd_consumed(r), "Multiple frame is not supported.");
    d->frame = fh;
}

/* ref: JpegScanHeader.cs:157-205 */
static int scan_try_parse(const uint8_t *buf, size_t len, int metadata_only, scan_header *sh, int *consumed) {
    *consumed = 0;
    if (len == 0) return 0;
    uint8_t ncomp = buf[0];
    buf++;
    len--;
    (*consumed)++;
    if (len < (size_t)(2 * ncomp + 3)) return 0;
    memset(sh, 0, sizeof *sh);
    sh->ncomp = ncomp;
    if (!metadata_only) {
        for (int i = 0; i < ncomp; i++) {
            sh->comp[i].selector = buf[0];
            sh->comp[i].td = (uint8_t)(buf[1] >> 4);
            sh->comp[i].ta = (uint8_t)(buf[1] & 0xf);
            buf += 2;
            *consumed += 2;
        }
    } else {
        buf += 2 * ncomp;
        *consumed += 2 * ncomp;
    }
    sh->ss = buf[0];
    sh->se = buf[1];
    sh->ah = (uint8_t)(buf[2] >> 4);
    sh->al = (uint8_t)(buf[2] & 0xf);
    *consumed += 3;
    return 1;
}

static void process_scan_header(jref_decoder *d, reader *r, int metadata_only, scan_header *sh) { /* ref: :291-307 */
    uint16_t length;
    const uint8_t *buf;
    if (!rd_try_read_length(r, &length))
        THROW_AT(d, rd_consumed(r), "Unexpected end of input data when reading segment length.");
    if (!rd_try_read_bytes(r, length, &buf))
        THROW_AT(d, rd_consumed(r), "Unexpected end of input data when reading segment content.");
    int consumed;
    if (!scan_try_parse(buf, length, metadata_only, sh, &consumed))
        THROW_AT(d, rd_consumed(r) - length + consumed, "Failed to parse scan header.");
}

static void process_dri(jref_decoder *d, reader *r) { /* ref: :635-650 */
    uint16_t length;
    const uint8_t *buf;
    if (!rd_try_read_length(r, &length))
        THROW_AT(d, rd_consumed(r), "Unexpected end of input data when reading segment length.");
    if (!rd_try_read_bytes(r, length, &buf) || length < 2)
        THROW_AT(d, rd_consumed(r), "Unexpected end of input data when reading segment content.");
    d->restart_interval = (buf[0] << 8) | buf[1];
}

static void process_dht(jref_decoder *d, reader *r) { /* ref: :672-700 */
    uint16_t length;
    const uint8_t *buf;
    if (!rd_try_read_length(r, &length))
        THROW_AT(d, rd_consumed(r), "Unexpected end of input data when reading segment length.");
    if (!rd_try_read_bytes(r, length, &buf))
        THROW_AT(d, rd_consumed(r), "Unexpected end of input data when reading segment content.");
    int offset = rd_consumed(r) - length;
    size_t len = length;
    while (len != 0) {
        huff_table t;
        int consumed = 1; /* Tc/Th byte, ref: JpegHuffmanDecodingTable.cs:152-166 */
        uint8_t tcth = buf[0];
        if (!huff_try_parse(&t, (uint8_t)(tcth >> 4), (uint8_t)(tcth & 0xf), buf + 1, len - 1, &consumed))
            THROW_AT(d, offset, "Failed to parse Huffman table.");
        buf += consumed;
        len -= (size_t)consumed;
        offset += consumed;
        set_huffman_table(d, &t);
    }
}

static void process_dqt(jref_decoder *d, reader *r, int load) { /* ref: :732-763 */
    uint16_t length;
    const uint8_t *buf;
    if (!rd_try_read_length(r, &length))
        THROW_AT(d, rd_consumed(r), "Unexpected end of input data when reading segment length.");
    if (!rd_try_read_bytes(r, length, &buf))
        THROW_AT(d, rd_consumed(r), "Unexpected end of input data when reading segment content.");
    if (!load) return;
    int offset = rd_consumed(r) - length;
    size_t len = length;
    while (len != 0) {
        /* ref: JpegQuantizationTable.cs:99-113, 192-232 */
        quant_table t;
        memset(&t, 0, sizeof t);
        uint8_t b = buf[0];
        int consumed = 1;
        t.precision = (uint8_t)(b >> 4);
        t.identifier = (uint8_t)(b & 0xf);
        const uint8_t *e = buf + 1;
        size_t elen = len - 1;
        int ok = 1;
        if (t.precision == 0) {
            if (elen < 64)
                ok = 0;
            else {
                for (int i = 0; i < 64; i++) t.elements[i] = e[i];
                consumed += 64;
            }
        } else if (t.precision == 1) {
            if (elen < 128)
                ok = 0;
            else {
                for (int i = 0; i < 64; i++) t.elements[i] = (uint16_t)(e[2 * i] << 8 | e[2 * i + 1]);
                consumed += 128;
            }
        } else {
            ok = 0;
        }
        if (!ok) THROW_AT(d, offset, "Failed to parse quantization table.");
        t.present = 1;
        buf += consumed;
        len -= (size_t)consumed;
        offset += consumed;
        set_quant_table(d, &t);
    }
}

/* ------------------------------------------------------------------------------------------------
 * Scan decoders
 * ---------------------------------------------------------------------------------------------- */

/* ref: ScanDecoder/JpegHuffmanScanDecoder.cs:17-72 */
static int init_decode_components(jref_decoder *d, const frame_header *fh, const scan_header *sh,
                                  decoding_component *components, int ncomponents_alloc) {
    int max_h = 1, max_v = 1;
    for (int i = 0; i < fh->ncomp; i++) {
        if (fh->comp[i].h > max_h) max_h = fh->comp[i].h;
        if (fh->comp[i].v > max_v) max_v = fh->comp[i].v;
    }
    if (ncomponents_alloc < sh->ncomp) throw_err(d, JREF_INVALID_OPERATION, "Operation is not valid.");
    for (int i = 0; i < sh->ncomp; i++) {
        const scan_component *sc = &sh->comp[i];
        int component_index = 0;
        const jref_component *fc = NULL;
        for (int j = 0; j < fh->ncomp; j++) {
            if (sc->selector == fh->comp[j].identifier) {
                component_index = j;
                fc = &fh->comp[j];
            }
        }
        if (!fc) THROW_SCAN_INVALID(d, "The specified component is missing.");
        decoding_component *c = &components[i];
        c->component_index = component_index;
        c->h = fc->h;
        c->v = fc->v;
        c->dc_table = get_huffman_table(d, 1, sc->td);
        c->ac_table = get_huffman_table(d, 0, sc->ta);
        c->quant = get_quant_table(d, fc->tq);
        if (c->h == 0 || c->v == 0) THROW_SCAN_INVALID(d, "Attempted to divide by zero."); /* DivideByZeroException */
        c->hs = max_h / c->h;
        c->vs = max_v / c->v;
        /* not restated: for a ratio that is not a power of two WriteBlockSlow's shifted indices leave the block's 64 samples
         * (unsafe reads in the reference, :247-266); the checker refuses instead of guessing what lies behind them */
        if ((c->hs & (c->hs - 1)) != 0 || (c->vs & (c->vs - 1)) != 0)
            throw_err(d, JREF_NOT_SUPPORTED, "Sampling factor ratios that are not powers of two are outside the restated envelope.");
        c->dc_predictor = 0;
    }
    return sh->ncomp;
}

/* ref: JpegHuffmanBaselineScanDecoder.cs:23-49 and JpegHuffmanProgressiveScanDecoder.cs:23-55 (ctor) */
static void scan_decoder_create(jref_decoder *d, int marker) {
    const frame_header *fh = &d->frame;
    d->scan_kind = SCAN_NONE;
    if (marker == 0xC3 || marker == 0xC9 || marker == 0xCA) {
        d->scan_kind = SCAN_UNSUPPORTED; /* lossless / arithmetic: out of scope for this oracle */
        return;
    }
    int max_h = 1, max_v = 1;
    for (int i = 0; i < fh->ncomp; i++) {
        if (fh->comp[i].h > max_h) max_h = fh->comp[i].h;
        if (fh->comp[i].v > max_v) max_v = fh->comp[i].v;
    }
    d->sd_frame = *fh;
    d->sd_max_h = max_h;
    d->sd_max_v = max_v;
    d->sd_mcus_per_line = (fh->samples_per_line + 8 * max_h - 1) / (8 * max_h);
    d->sd_mcus_per_column = (fh->lines + 8 * max_v - 1) / (8 * max_v);
    d->sd_level_shift = SHL32(1, fh->precision - 1);
    d->sd_ncomponents_alloc = fh->ncomp;
    d->sd_block_counter = 0;
    memset(d->sd_components, 0, sizeof d->sd_components);
    if (marker == 0xC0 || marker == 0xC1) {
        d->sd_restart_interval = d->restart_interval; /* latched here (F4) */
        d->scan_kind = SCAN_BASELINE;
        return;
    }
    /* progressive */
    if (!d->writer) THROW_SCAN_INVALID(d, "Output writer is not set.");
    d->pg_writer = d->writer;
    d->pg_writer_user = d->writer_user;
    /* ref: JpegBlockAllocator.cs:35-84 */
    int hblocks = (fh->samples_per_line + 7) / 8, vblocks = (fh->lines + 7) / 8;
    free(d->pg_alloc);
    free(d->pg_blocks);
    d->pg_alloc = (component_allocation *)calloc(fh->ncomp ? fh->ncomp : 1, sizeof(component_allocation));
    d->pg_nalloc = fh->ncomp;
    int index = 1; /* block 0 is the dummy sink */
    for (int i = 0; i < fh->ncomp; i++) {
        component_allocation *a = &d->pg_alloc[i];
        if (fh->comp[i].h == 0 || fh->comp[i].v == 0) THROW_SCAN_INVALID(d, "Attempted to divide by zero.");
        a->hs = max_h / fh->comp[i].h;
        a->vs = max_v / fh->comp[i].v;
        if ((a->hs & (a->hs - 1)) != 0 || (a->vs & (a->vs - 1)) != 0) /* see init_decode_components */
            throw_err(d, JREF_NOT_SUPPORTED, "Sampling factor ratios that are not powers of two are outside the restated envelope.");
        a->hblocks = (hblocks + a->hs - 1) / a->hs;
        a->vblocks = (vblocks + a->vs - 1) / a->vs;
        a->offset = index;
        index += a->hblocks * a->vblocks;
    }
    d->pg_blocks = (int16_t *)calloc((size_t)index * 64, sizeof(int16_t));
    d->scan_kind = SCAN_PROGRESSIVE;
}

/* ref: JpegHuffmanBaselineScanDecoder.cs:179-222 */
static void read_block_baseline(jref_decoder *d, bit_reader *b, decoding_component *c, int16_t *dst) {
    int t = decode_huffman_code(d, b, c->dc_table);
    if (t != 0) t = receive_and_extend(d, b, t);
    t += c->dc_predictor;
    c->dc_predictor = t;
    dst[0] = (int16_t)t;

    const huff_table *ac = c->ac_table;
    for (int i = 1; i < 64;) {
        int s = decode_huffman_code(d, b, ac);
        int r = s >> 4;
        s &= 15;
        if (s != 0) {
            i += r;
            s = receive_and_extend(d, b, s);
            int idx = i++;
            dst[idx < 63 ? idx : 63] = (int16_t)s;
        } else {
            if (r == 0) break;
            i += 16;
        }
    }
}

/* ref: JpegHuffmanBaselineScanDecoder.cs:51-177 */
static void baseline_process_scan(jref_decoder *d, reader *r, const scan_header *sh) {
    const frame_header *fh = &d->sd_frame;
    if (!d->writer) throw_err(d, JREF_INVALID_OPERATION, "Output writer is not specified.");

    int ncomp = init_decode_components(d, fh, sh, d->sd_components, d->sd_ncomponents_alloc);
    decoding_component *components = d->sd_components;
    for (int i = 0; i < ncomp; i++) {
        char msg[96];
        if (!components[i].dc_table || !components[i].ac_table) {
            snprintf(msg, sizeof msg, "Huffman table of component %d is not defined.", components[i].component_index);
            THROW_SCAN_INVALID(d, msg);
        }
        if (!components[i].quant) {
            snprintf(msg, sizeof msg, "Quantization table of component %d is not defined.",
                     components[i].component_index);
            THROW_SCAN_INVALID(d, msg);
        }
    }

    int max_h = d->sd_max_h, max_v = d->sd_max_v;
    int restart_interval = d->sd_restart_interval;
    int mcus_before_restart = restart_interval;
    int mcus_per_line = d->sd_mcus_per_line, mcus_per_column = d->sd_mcus_per_column;
    int level_shift = d->sd_level_shift;
    bit_reader br;
    br_init(&br, r->p, r->n);

    float block_f[64], output_f[64], temp_f[64];
    int16_t output[64];

    for (int row_mcu = 0; row_mcu < mcus_per_column; row_mcu++) {
        int offset_y = row_mcu * max_v;
        for (int col_mcu = 0; col_mcu < mcus_per_line; col_mcu++) {
            int offset_x = col_mcu * max_h;
            for (int ci = 0; ci < ncomp; ci++) {
                decoding_component *c = &components[ci];
                int index = c->component_index;
                for (int y = 0; y < c->v; y++) {
                    int block_offset_y = (offset_y + y) * 8;
                    for (int x = 0; x < c->h; x++) {
                        memset(output, 0, sizeof output);
                        read_block_baseline(d, &br, c, output);
                        if (d->tap) d->tap(d->tap_user, output, index, d->sd_block_counter);
                        d->sd_block_counter++;
                        dequantize_and_unzigzag(c->quant->elements, output, block_f);
                        transform_idct(block_f, output_f, temp_f);
                        shift_data_level(output_f, output, level_shift);
                        write_block_expanded(d->writer, d->writer_user, output, index, (offset_x + x) * 8,
                                             block_offset_y, c->hs, c->vs);
                    }
                }
            }
            /* restart, ref: :139-163 */
            if (restart_interval > 0 && (--mcus_before_restart) == 0) {
                br_advance_align_byte(&br);
                int marker = br_try_read_marker(&br);
                if (marker == 0xD9) {
                    int consumed_eoi = (int)r->n - br_remaining_bits(&br) / 8;
                    rd_try_advance(r, consumed_eoi - 2);
                    return;
                }
                if (!(marker >= 0xD0 && marker <= 0xD7))
                    throw_err(d, JREF_INVALID_OPERATION, "Expect restart marker.");
                mcus_before_restart = restart_interval;
                for (int ci = 0; ci < ncomp; ci++) components[ci].dc_predictor = 0;
            }
        }
    }

    br_advance_align_byte(&br);
    int consumed = (int)r->n - br_remaining_bits(&br) / 8;
    int pm = br_try_peek_marker(&br);
    if (pm != 0) {
        if (!(pm >= 0xD0 && pm <= 0xD7)) consumed -= 2;
    }
    rd_try_advance(r, consumed);
}

/* ref: JpegBlockAllocator.cs:93-114 */
static int16_t *pg_block_ref(jref_decoder *d, int component_index, int bx, int by) {
    if ((unsigned)component_index >= (unsigned)d->pg_nalloc)
        throw_err(d, JREF_ARGUMENT, "Specified argument was out of the range of valid values. (Parameter 'componentIndex')");
    const component_allocation *a = &d->pg_alloc[component_index];
    if (bx >= a->hblocks || by >= a->vblocks) return d->pg_blocks; /* dummy block 0 */
    return d->pg_blocks + (size_t)(a->offset + by * a->hblocks + bx) * 64;
}

/* ref: JpegHuffmanProgressiveScanDecoder.cs:196-224.  Returns 0 when EOI ended the scan. */
static int pg_handle_restart(jref_decoder *d, bit_reader *b, reader *r) {
    if (d->pg_restart_interval > 0 && (--d->pg_mcus_before_restart) == 0) {
        br_advance_align_byte(b);
        int marker = br_try_read_marker(b);
        if (marker == 0xD9) {
            int consumed_eoi = (int)r->n - br_remaining_bits(b) / 8;
            rd_try_advance(r, consumed_eoi - 2);
            return 0;
        }
        if (!(marker >= 0xD0 && marker <= 0xD7)) throw_err(d, JREF_INVALID_OPERATION, "Expect restart marker.");
        d->pg_mcus_before_restart = d->pg_restart_interval;
        d->pg_eobrun = 0;
        for (int i = 0; i < d->sd_ncomponents_alloc; i++) d->sd_components[i].dc_predictor = 0;
    }
    return 1;
}

/* ref: :227-253 */
static void pg_read_block_dc(jref_decoder *d, bit_reader *b, decoding_component *c, const scan_header *sh,
                             int16_t *blk) {
    if (sh->ah == 0) {
        int s = decode_huffman_code(d, b, c->dc_table);
        if (s != 0) s = receive_and_extend(d, b, s);
        s += c->dc_predictor;
        c->dc_predictor = s;
        blk[0] = (int16_t)SHL32(s, sh->al);
    } else {
        int bits, m;
        if (!br_try_read_bits(b, 1, &bits, &m)) THROW_SCAN_INVALID(d, "Unexpected end of JPEG data stream.");
        blk[0] = (int16_t)(blk[0] | (int16_t)SHL32(bits, sh->al));
    }
}

/* ref: :313-419 */
static void pg_read_block_ac_refined(jref_decoder *d, bit_reader *b, const huff_table *ac, const scan_header *sh,
                                     int *eobrun, int16_t *blk) {
    int start = sh->ss, end = sh->se;
    int p1 = SHL32(1, sh->al);
    int m1 = SHL32(-1, sh->al);
    int k = start;
    int bits, m;

    if (*eobrun == 0) {
        for (; k <= end; k++) {
            int s = decode_huffman_code(d, b, ac);
            int r = s >> 4;
            s &= 15;
            if (s != 0) {
                if (!br_try_read_bits(b, 1, &bits, &m)) THROW_SCAN_INVALID(d, "Unexpected end of JPEG data stream.");
                s = bits != 0 ? p1 : m1;
            } else {
                if (r != 15) {
                    *eobrun = SHL32(1, r);
                    if (r != 0) {
                        if (!br_try_read_bits(b, r, &bits, &m))
                            THROW_SCAN_INVALID(d, "Unexpected end of JPEG data stream.");
                        *eobrun += bits;
                    }
                    break;
                }
            }
            do {
                int16_t *coef = &blk[k];
                if (*coef != 0) {
                    if (!br_try_read_bits(b, 1, &bits, &m))
                        THROW_SCAN_INVALID(d, "Unexpected end of JPEG data stream.");
                    if (bits != 0) {
                        if ((*coef & p1) == 0) *coef = (int16_t)(*coef + (int16_t)(*coef >= 0 ? p1 : m1));
                    }
                } else {
                    if (--r < 0) break;
                }
                k++;
            } while (k <= end);

            if ((s != 0) && (k < 64)) blk[k] = (int16_t)s;
        }
    }

    if (*eobrun > 0) {
        for (; k <= end; k++) {
            int16_t *coef = &blk[k];
            if (*coef != 0) {
                if (!br_try_read_bits(b, 1, &bits, &m)) THROW_SCAN_INVALID(d, "Unexpected end of JPEG data stream.");
                if (bits != 0) {
                    if ((*coef & p1) == 0) *coef = (int16_t)(*coef + (int16_t)(*coef > 0 ? p1 : m1));
                }
            }
        }
        --*eobrun;
    }
}

/* ref: :255-311 */
static void pg_read_block_ac(jref_decoder *d, bit_reader *b, const huff_table *ac, const scan_header *sh, int *eobrun,
                             int16_t *blk) {
    if (sh->ah == 0) {
        if (*eobrun != 0) {
            --*eobrun;
            return;
        }
        int start = sh->ss, end = sh->se, low = sh->al;
        for (int i = start; i <= end; i++) {
            int s = decode_huffman_code(d, b, ac);
            int r = s >> 4;
            s &= 15;
            i += r;
            if (s != 0) {
                s = receive_and_extend(d, b, s);
                blk[i < 63 ? i : 63] = (int16_t)SHL32(s, low);
            } else {
                if (r != 15) {
                    *eobrun = SHL32(1, r);
                    if (r != 0) {
                        int bits, m;
                        if (!br_try_read_bits(b, r, &bits, &m))
                            THROW_SCAN_INVALID(d, "Unexpected end of JPEG data stream.");
                        *eobrun += bits;
                    }
                    --*eobrun;
                    break;
                }
            }
        }
    } else {
        pg_read_block_ac_refined(d, b, ac, sh, eobrun, blk);
    }
}

/* ref: JpegHuffmanProgressiveScanDecoder.cs:57-194 */
static void progressive_process_scan(jref_decoder *d, reader *r, const scan_header *sh) {
    if (!d->writer) throw_err(d, JREF_INVALID_OPERATION, "Operation is not valid.");
    /* not restated: a spectral selection that ends behind coefficient 63 makes the reference's block readers walk into the
     * NEXT blocks of its store (Unsafe.Add without a bound, :322-413); the checker refuses instead of following it there */
    if (sh->se > 63 || sh->ss > 63)
        throw_err(d, JREF_NOT_SUPPORTED, "A spectral selection beyond coefficient 63 is outside the restated envelope.");
    const frame_header *fh = &d->sd_frame;
    int ncomp = init_decode_components(d, fh, sh, d->sd_components, d->sd_ncomponents_alloc);
    decoding_component *components = d->sd_components;
    char msg[96];
    for (int i = 0; i < ncomp; i++) {
        if (!components[i].quant) {
            snprintf(msg, sizeof msg, "Quantization table of component %d is not defined.",
                     components[i].component_index);
            THROW_SCAN_INVALID(d, msg);
        }
    }
    d->pg_restart_interval = d->restart_interval; /* re-read per scan, ref: :78 */
    d->pg_mcus_before_restart = d->pg_restart_interval;
    d->pg_eobrun = 0;

    bit_reader br;
    if (ncomp == 1) {
        /* ref: :140-194 */
        decoding_component *c = &components[0];
        br_init(&br, r->p, r->n);
        int ci = c->component_index;
        int hcount = (fh->samples_per_line + 8 * c->hs - 1) / (8 * c->hs);
        int vcount = (fh->lines + 8 * c->vs - 1) / (8 * c->vs);
        if (sh->ss == 0) {
            if (!c->dc_table) {
                snprintf(msg, sizeof msg, "Huffman table of component %d is not defined.", ci);
                THROW_SCAN_INVALID(d, msg);
            }
            for (int by = 0; by < vcount; by++)
                for (int bx = 0; bx < hcount; bx++) {
                    pg_read_block_dc(d, &br, c, sh, pg_block_ref(d, ci, bx, by));
                    if (!pg_handle_restart(d, &br, r)) return;
                }
        } else {
            if (!c->ac_table) {
                snprintf(msg, sizeof msg, "Huffman table of component %d is not defined.", ci);
                THROW_SCAN_INVALID(d, msg);
            }
            for (int by = 0; by < vcount; by++)
                for (int bx = 0; bx < hcount; bx++) {
                    pg_read_block_ac(d, &br, c->ac_table, sh, &d->pg_eobrun, pg_block_ref(d, ci, bx, by));
                    if (!pg_handle_restart(d, &br, r)) return;
                }
        }
    } else {
        /* ref: :92-138 */
        for (int i = 0; i < ncomp; i++) {
            if (!components[i].dc_table) {
                snprintf(msg, sizeof msg, "Huffman table of component %d is not defined.",
                         components[i].component_index);
                THROW_SCAN_INVALID(d, msg);
            }
        }
        br_init(&br, r->p, r->n);
        for (int row_mcu = 0; row_mcu < d->sd_mcus_per_column; row_mcu++) {
            for (int col_mcu = 0; col_mcu < d->sd_mcus_per_line; col_mcu++) {
                for (int i = 0; i < ncomp; i++) {
                    decoding_component *c = &components[i];
                    int offset_x = col_mcu * c->h, offset_y = row_mcu * c->v;
                    for (int y = 0; y < c->v; y++)
                        for (int x = 0; x < c->h; x++)
                            pg_read_block_dc(d, &br, c, sh, pg_block_ref(d, c->component_index, offset_x + x, offset_y + y));
                }
                if (!pg_handle_restart(d, &br, r)) return;
            }
        }
    }
    /* NB: the progressive ProcessScan never advances the outer reader. */
}

/* ref: JpegHuffmanProgressiveScanDecoder.cs:421-470 + JpegBlockAllocator.cs:120-149 */
static void progressive_dispose(jref_decoder *d) {
    float block_f[64], output_f[64], temp_f[64];
    decoding_component *components = d->sd_components;
    if (d->ptap) {
        for (int i = 0; i < d->sd_ncomponents_alloc; i++) {
            const decoding_component *c = &components[i];
            if (!c->quant || c->component_index >= d->pg_nalloc) continue;
            const component_allocation *a = &d->pg_alloc[c->component_index];
            for (int by = 0; by < a->vblocks; by++)
                for (int bx = 0; bx < a->hblocks; bx++)
                    d->ptap(d->ptap_user, d->pg_blocks + (size_t)(a->offset + by * a->hblocks + bx) * 64, c->component_index, bx, by,
                            c->quant->elements);
        }
    }
    for (int row_mcu = 0; row_mcu < d->sd_mcus_per_column; row_mcu++) {
        for (int col_mcu = 0; col_mcu < d->sd_mcus_per_line; col_mcu++) {
            /* iterates ALL pre-allocated components as left by the last scans */
            for (int i = 0; i < d->sd_ncomponents_alloc; i++) {
                decoding_component *c = &components[i];
                int offset_x = col_mcu * c->h, offset_y = row_mcu * c->v;
                for (int y = 0; y < c->v; y++) {
                    for (int x = 0; x < c->h; x++) {
                        int16_t *blk = pg_block_ref(d, c->component_index, offset_x + x, offset_y + y);
                        if (!c->quant) continue; /* Debug.Assert in the reference; unreachable for valid files */
                        dequantize_and_unzigzag(c->quant->elements, blk, block_f);
                        transform_idct(block_f, output_f, temp_f);
                        shift_data_level(output_f, blk, d->sd_level_shift);
                    }
                }
            }
        }
    }
    for (int i = 0; i < d->pg_nalloc; i++) {
        const component_allocation *a = &d->pg_alloc[i];
        for (int row = 0; row < a->vblocks; row++)
            for (int col = 0; col < a->hblocks; col++)
                write_block_expanded(d->pg_writer, d->pg_writer_user,
                                     d->pg_blocks + (size_t)(a->offset + row * a->hblocks + col) * 64, i,
                                     col * a->hs * 8, row * a->vs * 8, a->hs, a->vs);
    }
    free(d->pg_alloc);
    free(d->pg_blocks);
    d->pg_alloc = NULL;
    d->pg_blocks = NULL;
    d->pg_nalloc = 0;
}

static void scan_decoder_dispose(jref_decoder *d) {
    if (d->scan_kind == SCAN_PROGRESSIVE && d->pg_blocks) progressive_dispose(d);
    d->scan_kind = SCAN_NONE;
}

/* ------------------------------------------------------------------------------------------------
 * JpegDecoder public surface
 * ---------------------------------------------------------------------------------------------- */

jref_decoder *jref_create(void) { return (jref_decoder *)calloc(1, sizeof(jref_decoder)); }

void jref_destroy(jref_decoder *d) {
    if (!d) return;
    free(d->pg_alloc);
    free(d->pg_blocks);
    free(d);
}

const char *jref_last_error(const jref_decoder *d) { return d->err; }

void jref_set_input(jref_decoder *d, const uint8_t *data, size_t len) {
    d->input = data;
    d->input_len = len;
    d->frame.present = 0;
    d->restart_interval = 0;
}

void jref_set_output_writer(jref_decoder *d, jref_write_block_fn fn, void *user) {
    d->writer = fn;
    d->writer_user = user;
}

void jref_set_coef_tap(jref_decoder *d, jref_coef_tap_fn fn, void *user) {
    d->tap = fn;
    d->tap_user = user;
}

void jref_set_progressive_tap(jref_decoder *d, jref_progressive_tap_fn fn, void *user) {
    d->ptap = fn;
    d->ptap_user = user;
}

int jref_get_restart_interval(const jref_decoder *d) { return d->restart_interval; }

static int is_sof(int m) {
    return m == 0xC0 || m == 0xC1 || m == 0xC2 || m == 0xC3 || m == 0xC5 || m == 0xC6 || m == 0xC7 || m == 0xC9 ||
           m == 0xCA || m == 0xCB || m == 0xCD || m == 0xCE || m == 0xCF;
}

/* ref: JpegDecoder.cs:114-162 */
static int process_marker_for_identification(jref_decoder *d, int marker, reader *r, int load_qt) {
    if (marker == 0xD8) {
    } else if (is_sof(marker)) {
        d->start_of_frame = marker;
        process_frame_header(d, r, 0, 0);
    } else if (marker == 0xDA) {
        scan_header sh;
        process_scan_header(d, r, 1, &sh);
    } else if (marker == 0xDD) {
        process_dri(d, r);
    } else if (marker == 0xDB) {
        process_dqt(d, r, load_qt);
    } else if (marker >= 0xD0 && marker <= 0xD7) {
    } else if (marker == 0xD9) {
        return 0;
    } else {
        process_other_marker(d, r);
    }
    return 1;
}

int jref_identify(jref_decoder *d, int load_quantization_tables, jref_info *info) {
    d->err[0] = 0;
    d->err_code = JREF_OK;
    if (d->input_len == 0 || !d->input) {
        snprintf(d->err, sizeof d->err, "Input buffer is not specified.");
        return d->err_code = JREF_INVALID_OPERATION;
    }
    d->jb_armed = 1;
    if (setjmp(d->jb)) {
        d->jb_armed = 0;
        return d->err_code;
    }
    reader r = {d->input, d->input_len, d->input_len};
    d->frame.present = 0;
    int to_continue = 1;
    while (to_continue && r.n != 0) {
        int marker;
        if (!rd_try_read_marker(&r, &marker)) THROW_AT(d, rd_consumed(&r), "No marker found.");
        to_continue = process_marker_for_identification(d, marker, &r, load_quantization_tables);
    }
    if (!d->frame.present) throw_err(d, JREF_INVALID_OPERATION, "Frame header was not found.");
    d->jb_armed = 0;
    if (info) {
        memset(info, 0, sizeof *info);
        info->width = d->frame.samples_per_line;
        info->height = d->frame.lines;
        info->precision = d->frame.precision;
        info->ncomp = d->frame.ncomp;
        info->sof = d->start_of_frame;
        info->restart_interval = d->restart_interval;
        info->consumed = rd_consumed(&r);
        for (int i = 0; i < d->frame.ncomp && i < 4; i++) info->comp[i] = d->frame.comp[i];
    }
    return JREF_OK;
}

/* ref: JpegStandardQuantizationTable.cs:12-34 (Annex K tables, zig-zag order as the reference stores them) */
static const uint16_t k_std_lum_zz[64] = {16, 11, 12, 14, 12, 10, 16, 14, 13, 14, 18, 17, 16, 19, 24, 40,
                                          26, 24, 22, 22, 24, 49, 35, 37, 29, 40, 58, 51, 61, 60, 57, 51,
                                          56, 55, 64, 72, 92, 78, 64, 68, 87, 69, 55, 56, 80, 109, 81, 87,
                                          95, 98, 103, 104, 103, 62, 77, 113, 121, 112, 100, 120, 92, 101, 103, 99};
static const uint16_t k_std_chr_zz[64] = {17, 18, 18, 24, 21, 24, 47, 26, 26, 47, 99, 66, 56, 66, 99, 99,
                                          99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99,
                                          99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99,
                                          99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99};

/* ref: JpegDecoder.cs:198-249 */
static float estimate_quality(const uint16_t *q, const uint16_t *std) {
    int all_ones = 1;
    double sum = 0, sumsq = 0;
    for (int i = 0; i < 64; i++) {
        double pct;
        if (q[i] == 0)
            pct = 999.99;
        else
            pct = 100.0 * q[i] / std[i];
        sum += pct;
        sumsq += pct * pct;
        if (q[i] != 1) all_ones = 0;
    }
    sum /= 64.0;
    if (all_ones) return 100.0f;
    if (sum <= 100.0) return (float)((200.0 - sum) / 2.0);
    return (float)(5000.0 / sum);
}

int jref_try_estimate_quality(jref_decoder *d, float *quality) { /* ref: :169-196 */
    const quant_table *q0 = get_quant_table(d, 0);
    if (d->nqt == 0 || !q0) {
        *quality = 0;
        return 0;
    }
    float q = estimate_quality(q0->elements, k_std_lum_zz);
    const quant_table *q1 = get_quant_table(d, 1);
    if (q1) {
        float q2 = estimate_quality(q1->elements, k_std_chr_zz);
        if (q2 < q) q = q2;
    }
    if (q < 0.f) q = 0.f;
    if (q > 100.f) q = 100.f;
    *quality = q;
    return 1;
}

/* ref: JpegDecoder.cs:558-617 */
static int process_marker_for_decode(jref_decoder *d, int marker, reader *r) {
    switch (marker) {
    case 0xC0: case 0xC1: case 0xC2: case 0xC3: case 0xC9: case 0xCA:
        process_frame_header(d, r, 0, 1);
        scan_decoder_create(d, marker);
        break;
    case 0xC5: case 0xC6: case 0xC7: case 0xCB: case 0xCD: case 0xCE: case 0xCF:
        THROW_AT(d, rd_consumed(r), "This type of JPEG stream is not supported.");
        break;
    case 0xC4:
        process_dht(d, r);
        break;
    case 0xCC:
        process_other_marker(d, r); /* DAC: arithmetic conditioning, parsed-and-kept in the reference; unused here */
        break;
    case 0xDB:
        process_dqt(d, r, 1);
        break;
    case 0xDD:
        process_dri(d, r);
        break;
    case 0xDA: {
        if (d->scan_kind == SCAN_NONE) THROW_AT(d, rd_consumed(r), "Scan header appears before frame header.");
        scan_header sh;
        process_scan_header(d, r, 0, &sh);
        if (d->scan_kind == SCAN_BASELINE)
            baseline_process_scan(d, r, &sh);
        else if (d->scan_kind == SCAN_PROGRESSIVE)
            progressive_process_scan(d, r, &sh);
        else
            throw_err(d, JREF_NOT_SUPPORTED, "Lossless / arithmetic scans are outside this oracle's scope.");
        break;
    }
    case 0xD0: case 0xD1: case 0xD2: case 0xD3: case 0xD4: case 0xD5: case 0xD6: case 0xD7:
        break;
    case 0xD9:
        return 0;
    default:
        process_other_marker(d, r);
        break;
    }
    return 1;
}

int jref_decode(jref_decoder *d) { /* ref: JpegDecoder.cs:509-550 */
    d->err[0] = 0;
    d->err_code = JREF_OK;
    if (d->input_len == 0 || !d->input) {
        snprintf(d->err, sizeof d->err, "Input buffer is not specified.");
        return d->err_code = JREF_INVALID_OPERATION;
    }
    if (!d->writer) {
        snprintf(d->err, sizeof d->err, "The output buffer is not specified.");
        return d->err_code = JREF_INVALID_OPERATION;
    }
    reader r = {d->input, d->input_len, d->input_len};
    d->scan_kind = SCAN_NONE;
    d->jb_armed = 1;
    if (setjmp(d->jb)) {
        /* finally { _scanDecoder?.Dispose(); } -- runs even on failure */
        int code = d->err_code;
        char saved[sizeof d->err];
        memcpy(saved, d->err, sizeof saved);
        d->jb_armed = 0;
        scan_decoder_dispose(d);
        memcpy(d->err, saved, sizeof saved);
        return d->err_code = code;
    }
    if (!rd_try_read_soi(&r)) THROW_AT(d, rd_consumed(&r), "Marker StartOfImage not found.");
    int to_continue = 1;
    while (to_continue && r.n != 0) {
        int marker;
        if (!rd_try_read_marker(&r, &marker)) THROW_AT(d, rd_consumed(&r), "No marker found.");
        to_continue = process_marker_for_decode(d, marker, &r);
    }
    d->jb_armed = 0;
    scan_decoder_dispose(d);
    return JREF_OK;
}

/* ------------------------------------------------------------------------------------------------
 * Sinks
 * ---------------------------------------------------------------------------------------------- */

/* ref: apps/JpegDecode/JpegBufferOutputWriter8Bit.cs:28-60 */
void jref_sink8_write(void *sink, const int16_t *block, int component_index, int x, int y) {
    jref_sink8 *s = (jref_sink8 *)sink;
    int cc = s->component_count, width = s->width, height = s->height;
    if (x > width || y > height) return;
    int ww = width - x < 8 ? width - x : 8;
    int wh = height - y < 8 ? height - y : 8;
    uint8_t *dst = s->out + (size_t)y * width * cc + (size_t)x * cc + component_index;
    for (int dy = 0; dy < wh; dy++) {
        uint8_t *row = dst + (size_t)dy * width * cc;
        for (int dx = 0; dx < ww; dx++) {
            int16_t v = block[dx];
            row[dx * cc] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
        }
        block += 8;
    }
}

/* ref: tests/JpegLibrary.Tests/Utils/JpegExtendingOutputWriter.cs:77-112 */
static uint32_t fast_expand_bits(uint32_t bits, int bit_count) {
    int remaining = 16 - bit_count;
    return (bits << remaining) | (bits & ((uint32_t)(1 << remaining) - 1));
}
static uint32_t expand_bits(uint32_t bits, int bit_count) {
    int cur = bit_count;
    while (cur < 16) {
        bits = (bits << bit_count) | bits;
        cur += bit_count;
    }
    if (cur > 16) {
        bits = bits >> bit_count;
        cur -= bit_count;
        bits = fast_expand_bits(bits, cur);
    }
    return bits;
}

/* ref: tests/JpegLibrary.Tests/Utils/JpegExtendingOutputWriter.cs:30-75 */
void jref_sink16_write(void *sink, const int16_t *block, int component_index, int x, int y) {
    jref_sink16 *s = (jref_sink16 *)sink;
    int precision = s->precision;
    uint16_t max = (uint16_t)((1 << precision) - 1);
    int cc = s->component_count, width = s->width, height = s->height;
    if (x > width || y > height) return;
    int ww = width - x < 8 ? width - x : 8;
    int wh = height - y < 8 ? height - y : 8;
    uint16_t *dst = s->out + (size_t)y * width * cc + (size_t)x * cc + component_index;
    for (int dy = 0; dy < wh; dy++) {
        uint16_t *row = dst + (size_t)dy * width * cc;
        for (int dx = 0; dx < ww; dx++) {
            uint16_t u = (uint16_t)block[dx];
            uint32_t v = u > max ? max : u;
            row[dx * cc] = (uint16_t)(precision >= 8 ? fast_expand_bits(v, precision) : expand_bits(v, precision));
        }
        block += 8;
    }
}

void jref_sink_raw_write(void *sink, const int16_t *block, int component_index, int x, int y) {
    jref_sink_raw *s = (jref_sink_raw *)sink;
    if (component_index < 0 || component_index >= s->component_count) return;
    int16_t *plane = s->out + (size_t)component_index * s->padded_width * s->padded_height;
    for (int dy = 0; dy < 8; dy++) {
        int yy = y + dy;
        if (yy < 0 || yy >= s->padded_height) continue;
        for (int dx = 0; dx < 8; dx++) {
            int xx = x + dx;
            if (xx < 0 || xx >= s->padded_width) continue;
            plane[(size_t)yy * s->padded_width + xx] = block[dy * 8 + dx];
        }
    }
}

/* ------------------------------------------------------------------------------------------------
 * One-call helpers: Identify + SetOutputWriter + Decode, the call pattern of every reference caller
 * (apps/JpegDecode/DecodeAction.cs:26-56, tests/.../HuffmanSequentialDecodeTests.cs:30-38,
 *  tests/JpegLibrary.Benchmarks/DecoderBenchmark.cs:54-65)
 * ---------------------------------------------------------------------------------------------- */

int jref_decode_to_8bit(const uint8_t *data, size_t len, int component_count, uint8_t *out, size_t out_cap,
                        jref_info *info, char *err, size_t errcap) {
    jref_decoder *d = jref_create();
    jref_info local;
    if (!info) info = &local;
    jref_set_input(d, data, len);
    int rc = jref_identify(d, 0, info);
    if (rc == JREF_OK) {
        if ((size_t)info->width * info->height * component_count > out_cap) {
            snprintf(d->err, sizeof d->err, "Destination buffer is too small.");
            rc = JREF_ARGUMENT;
        } else {
            jref_sink8 sink = {info->width, info->height, component_count, out};
            jref_set_output_writer(d, jref_sink8_write, &sink);
            rc = jref_decode(d);
        }
    }
    if (err && errcap) snprintf(err, errcap, "%s", d->err);
    jref_destroy(d);
    return rc;
}

/* ---- bench.py's CPU baseline: one independent decoder per host thread (see jpegref.h) ---- */
#include <pthread.h>
#include <time.h>

typedef struct jref_mt_shared {
    const uint8_t *const *files;
    const size_t *lens;
    int n, component_count, threads, warm;
    int rgba; /* DecoderBenchmark.cs:67: ConvertYCbCr8ToRgba32 behind every Decode() */
    pthread_mutex_t lock;
    pthread_cond_t cv;
    int ready, go, abort_run;
    int first_error; /* status of the first failing image, 0 = none */
    char err[256];
} jref_mt_shared;
typedef struct jref_mt_thread {
    jref_mt_shared *sh;
    int tid;
    uint64_t pixels;
    struct timespec t_done;
} jref_mt_thread;

static void *jref_mt_main(void *arg) {
    jref_mt_thread *me = (jref_mt_thread *)arg;
    jref_mt_shared *sh = me->sh;
    uint8_t *out = NULL, *out_rgba = NULL;
    size_t cap = 0;
    char err[256];
    jref_info info;
    /* size and touch this thread's output buffer before the clock starts (Identify of its images only) */
    for (int i = me->tid; i < sh->n; i += sh->threads) {
        jref_decoder *d = jref_create();
        jref_set_input(d, sh->files[i], sh->lens[i]);
        if (jref_identify(d, 0, &info) == JREF_OK) {
            const size_t need = (size_t)info.width * info.height * sh->component_count;
            if (need > cap) cap = need;
        }
        jref_destroy(d);
    }
    if (cap) {
        out = (uint8_t *)malloc(cap);
        if (out) memset(out, 0, cap);
        if (sh->rgba && sh->component_count == 3) { /* (the reference allocates this one inside the timed call: not charged here) */
            out_rgba = (uint8_t *)malloc(cap / 3 * 4);
            if (out_rgba) memset(out_rgba, 0, cap / 3 * 4);
        }
    }
    if (sh->warm && out && me->tid < sh->n) {
        if (jref_decode_to_8bit(sh->files[me->tid], sh->lens[me->tid], sh->component_count, out, cap, &info, err, sizeof err) == JREF_OK && out_rgba)
            jref_ycbcr8_to_rgb(out, out_rgba, (size_t)info.width * info.height, 4);
    }
    pthread_mutex_lock(&sh->lock);
    sh->ready++;
    pthread_cond_broadcast(&sh->cv);
    while (!sh->go) pthread_cond_wait(&sh->cv, &sh->lock);
    const int aborted = sh->abort_run;
    pthread_mutex_unlock(&sh->lock);
    for (int i = me->tid; i < sh->n && out && !aborted; i += sh->threads) {
        const int rc = jref_decode_to_8bit(sh->files[i], sh->lens[i], sh->component_count, out, cap, &info, err, sizeof err);
        if (rc != JREF_OK) {
            pthread_mutex_lock(&sh->lock);
            if (!sh->first_error) {
                sh->first_error = rc;
                snprintf(sh->err, sizeof sh->err, "image %d: %.200s", i, err);
            }
            pthread_mutex_unlock(&sh->lock);
            break;
        }
        if (out_rgba) jref_ycbcr8_to_rgb(out, out_rgba, (size_t)info.width * info.height, 4);
        me->pixels += (uint64_t)info.width * (uint64_t)info.height;
    }
    clock_gettime(CLOCK_MONOTONIC, &me->t_done);
    free(out);
    free(out_rgba);
    return NULL;
}

int jref_decode_batch_mt(const uint8_t *const *files, const size_t *lens, int n, int component_count, int threads, int warm,
                         double *seconds, uint64_t *pixels, char *err, size_t errcap) {
    return jref_decode_batch_mt_ex(files, lens, n, component_count, threads, warm, 0, seconds, pixels, err, errcap);
}

int jref_decode_batch_mt_ex(const uint8_t *const *files, const size_t *lens, int n, int component_count, int threads, int warm, int rgba,
                            double *seconds, uint64_t *pixels, char *err, size_t errcap) {
    if (seconds) *seconds = 0;
    if (pixels) *pixels = 0;
    if (n <= 0 || threads <= 0 || !files || !lens) return JREF_ARGUMENT;
    if (threads > n) threads = n;
    jref_mt_shared sh;
    memset(&sh, 0, sizeof sh);
    sh.files = files;
    sh.lens = lens;
    sh.n = n;
    sh.component_count = component_count;
    sh.threads = threads;
    sh.warm = warm;
    sh.rgba = rgba;
    pthread_mutex_init(&sh.lock, NULL);
    pthread_cond_init(&sh.cv, NULL);
    jref_mt_thread *th = (jref_mt_thread *)calloc((size_t)threads, sizeof *th);
    pthread_t *ids = (pthread_t *)calloc((size_t)threads, sizeof *ids);
    if (!th || !ids) {
        free(th);
        free(ids);
        return JREF_INVALID_OPERATION;
    }
    int started = 0;
    for (int t = 0; t < threads; t++) {
        th[t].sh = &sh;
        th[t].tid = t;
        if (pthread_create(&ids[t], NULL, jref_mt_main, &th[t]) != 0) break;
        started++;
    }
    /* the clock starts when every thread has its buffer and is parked */
    struct timespec t_start;
    pthread_mutex_lock(&sh.lock);
    while (sh.ready < started) pthread_cond_wait(&sh.cv, &sh.lock);
    if (started != threads) {
        sh.abort_run = 1;
        sh.first_error = JREF_INVALID_OPERATION;
        snprintf(sh.err, sizeof sh.err, "pthread_create failed after %d of %d threads", started, threads);
    }
    clock_gettime(CLOCK_MONOTONIC, &t_start);
    sh.go = 1;
    pthread_cond_broadcast(&sh.cv);
    pthread_mutex_unlock(&sh.lock);
    uint64_t px = 0;
    struct timespec t_end = t_start;
    for (int t = 0; t < started; t++) {
        pthread_join(ids[t], NULL);
        px += th[t].pixels;
        if (th[t].t_done.tv_sec > t_end.tv_sec || (th[t].t_done.tv_sec == t_end.tv_sec && th[t].t_done.tv_nsec > t_end.tv_nsec)) t_end = th[t].t_done;
    }
    if (seconds) *seconds = (double)(t_end.tv_sec - t_start.tv_sec) + (double)(t_end.tv_nsec - t_start.tv_nsec) * 1e-9;
    if (pixels) *pixels = px;
    if (err && errcap) snprintf(err, errcap, "%s", sh.err);
    pthread_cond_destroy(&sh.cv);
    pthread_mutex_destroy(&sh.lock);
    free(th);
    free(ids);
    return sh.first_error;
}

int jref_decode_to_16bit(const uint8_t *data, size_t len, int component_count, uint16_t *out, size_t out_cap,
                         jref_info *info, char *err, size_t errcap) {
    jref_decoder *d = jref_create();
    jref_info local;
    if (!info) info = &local;
    jref_set_input(d, data, len);
    int rc = jref_identify(d, 0, info);
    if (rc == JREF_OK) {
        if ((size_t)info->width * info->height * component_count > out_cap) {
            snprintf(d->err, sizeof d->err, "Destination buffer is too small.");
            rc = JREF_ARGUMENT;
        } else {
            jref_sink16 sink = {info->width, info->height, component_count, info->precision, out};
            jref_set_output_writer(d, jref_sink16_write, &sink);
            rc = jref_decode(d);
        }
    }
    if (err && errcap) snprintf(err, errcap, "%s", d->err);
    jref_destroy(d);
    return rc;
}

/* ------------------------------------------------------------------------------------------------
 * YCbCr -> RGB(A), the step the reference's callers run right after the decode
 * (ref: apps/JpegDecode/JpegYCbCrToRgbConverter.cs; callers apps/JpegDecode/DecodeAction.cs:71-74,
 * tests/JpegLibrary.Benchmarks/DecoderBenchmark.cs:67).  Table-driven, restated literally.
 * ---------------------------------------------------------------------------------------------- */

#define YCC_SHIFT 16
#define YCC_ONE_HALF (1 << (YCC_SHIFT - 1))
#define YCC_CLAMP_OFFSET 256

static uint8_t ycc_clamp[4 * 256];
static int32_t ycc_cr_r[256], ycc_cb_b[256], ycc_cr_g[256], ycc_cb_g[256], ycc_y[256];
static int ycc_ready = 0;

/* ref: :121-124 */
static int ycc_fix(float x) { return (int)((double)(x * (float)(1L << YCC_SHIFT)) + 0.5); }

/* ref: :126-129 */
static int ycc_code2v(int c, float rb, float rw, float cr) {
    return (int)((((float)(c - (int)rb)) * cr) / ((int)(rw - rb) != 0 ? (rw - rb) : 1.0f));
}

/* ref: :66-118 (constructor :24-48) */
static void ycc_init(void) {
    float luma[3] = {299 / 1000.0f, 587 / 1000.0f, 114 / 1000.0f};
    float rbw[6] = {0.0f, 255.0f, 128.0f, 255.0f, 128.0f, 255.0f};
    memset(ycc_clamp, 0, sizeof ycc_clamp);
    for (int i = 0; i < 256; i++) ycc_clamp[YCC_CLAMP_OFFSET + i] = (uint8_t)i;
    for (int i = YCC_CLAMP_OFFSET + 256; i < YCC_CLAMP_OFFSET + 256 + 2 * 256; i++) ycc_clamp[i] = 255;
    float luma_red = luma[0], luma_green = luma[1], luma_blue = luma[2];
    float f1 = 2 - 2 * luma_red;
    int d1 = ycc_fix(f1);
    float f2 = luma_red * f1 / luma_green;
    int d2 = -ycc_fix(f2);
    float f3 = 2 - 2 * luma_blue;
    int d3 = ycc_fix(f3);
    float f4 = luma_blue * f3 / luma_green;
    int d4 = -ycc_fix(f4);
    for (int i = 0, x = -128; i < 256; i++, x++) {
        int cr = ycc_code2v(x, rbw[4] - 128.0f, rbw[5] - 128.0f, 127);
        int cb = ycc_code2v(x, rbw[2] - 128.0f, rbw[3] - 128.0f, 127);
        ycc_cr_r[i] = (d1 * cr + YCC_ONE_HALF) >> YCC_SHIFT;
        ycc_cb_b[i] = (d3 * cb + YCC_ONE_HALF) >> YCC_SHIFT;
        ycc_cr_g[i] = d2 * cr;
        ycc_cb_g[i] = d4 * cb + YCC_ONE_HALF;
        ycc_y[i] = ycc_code2v(x + 128, rbw[0], rbw[1], 255);
    }
    ycc_ready = 1;
}

/* ConvertYCbCr8ToRgb24 (:171-206) when bytes_per_pixel == 3, ConvertYCbCr8ToRgba32 (:134-169) when 4 */
void jref_ycbcr8_to_rgb(const uint8_t *ycbcr, uint8_t *out, size_t count, int bytes_per_pixel) {
    if (!ycc_ready) ycc_init();
    for (size_t i = 0; i < count; i++) {
        uint8_t y = ycbcr[0], cb = ycbcr[1], cr = ycbcr[2];
        int yv = ycc_y[y];
        out[0] = ycc_clamp[YCC_CLAMP_OFFSET + yv + ycc_cr_r[cr]];
        out[1] = ycc_clamp[YCC_CLAMP_OFFSET + yv + ((ycc_cb_g[cb] + ycc_cr_g[cr]) >> YCC_SHIFT)];
        out[2] = ycc_clamp[YCC_CLAMP_OFFSET + yv + ycc_cb_b[cb]];
        if (bytes_per_pixel == 4) out[3] = 255;
        ycbcr += 3;
        out += bytes_per_pixel;
    }
}

/* ------------------------------------------------------------------------------------------------
 * JpegOptimizer (ref: JpegOptimizer.cs, JpegHuffmanEncodingTableBuilder.cs): shares the reader, bit reader and decoding
 * tables above.
 * ---------------------------------------------------------------------------------------------- */
#include "jpegopt.inc"
