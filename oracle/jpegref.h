/*
 * oracle/jpegref.h -- CPU restatement ("oracle") of yigolden/JpegLibrary's Huffman-DCT decode path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only as
 * the checker / CPU baseline.  The product path (jpeglibrary_amd/, libjpgpu.so) never links it.
 *
 * Parity pin: this restatement reproduces the reference's own golden PNG dumps
 * (tests/golden/X.high.png, X.low-diff.png: copied data files of the reference's xunit tests)
 * with 0 mismatching samples -- see tests/test_oracle_golden.py.
 *
 * All "ref:" citations are relative to /root/reference/src/JpegLibrary unless stated.
 */
#ifndef JPEGREF_H
#define JPEGREF_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Exception classes of the reference, as status codes. */
enum {
    JREF_OK = 0,
    JREF_INVALID_DATA = 1,      /* System.IO.InvalidDataException           */
    JREF_INVALID_OPERATION = 2, /* System.InvalidOperationException         */
    JREF_NOT_SUPPORTED = 3,     /* System.NotSupportedException             */
    JREF_ARGUMENT = 4           /* System.ArgumentException family          */
};

typedef struct jref_component {
    uint8_t identifier, h, v, tq;
} jref_component;

typedef struct jref_info {
    int width, height, precision, ncomp;
    int sof;              /* marker byte of the frame header (0xC0..0xCF)              */
    int restart_interval; /* value latched by Identify (LAST DRI in the file)          */
    int consumed;         /* Identify() return value = bytes consumed up to and incl. EOI */
    jref_component comp[4];
} jref_info;

/*
 * The JpegBlockOutputWriter.WriteBlock callback (ref: JpegBlockOutputWriter.cs:17):
 * 64 int16 row-major, unclamped, at full-resolution pixel coordinates (x, y).
 */
typedef void (*jref_write_block_fn)(void *user, const int16_t *block, int component_index, int x, int y);

/* Optional tap: every entropy-decoded baseline block, in decode order, before dequantisation
 * (64 int16 in zig-zag order).  block_index counts blocks in scan order from 0. */
typedef void (*jref_coef_tap_fn)(void *user, const int16_t *zigzag_coefs, int component_index, long block_index);

/* Optional tap (progressive only): every block of the coefficient store right before Dispose() dequantises it
 * (ref: ScanDecoder/JpegHuffmanProgressiveScanDecoder.cs:421-470): 64 int16 in zig-zag order, block (bx, by) of component
 * `component_index`, and the quantisation table (zig-zag order) Dispose will use for it. */
typedef void (*jref_progressive_tap_fn)(void *user, const int16_t *zigzag_coefs, int component_index, int bx, int by,
                                        const uint16_t *quant_zigzag);

typedef struct jref_decoder jref_decoder;

jref_decoder *jref_create(void);
void jref_destroy(jref_decoder *d);
const char *jref_last_error(const jref_decoder *d);

/* ref: JpegDecoder.SetInput (JpegDecoder.cs:56-62): resets frame header and restart interval. */
void jref_set_input(jref_decoder *d, const uint8_t *data, size_t len);
/* ref: JpegDecoder.Identify(bool) (JpegDecoder.cs:75-105).  Returns status; info->consumed = return value. */
int jref_identify(jref_decoder *d, int load_quantization_tables, jref_info *info);
/* ref: JpegDecoder.TryEstimateQuanlity (JpegDecoder.cs:169-249).  Returns 1 and *quality on success. */
int jref_try_estimate_quality(jref_decoder *d, float *quality);
/* ref: JpegDecoder.SetOutputWriter (JpegDecoder.cs:501). */
void jref_set_output_writer(jref_decoder *d, jref_write_block_fn fn, void *user);
void jref_set_coef_tap(jref_decoder *d, jref_coef_tap_fn fn, void *user);
void jref_set_progressive_tap(jref_decoder *d, jref_progressive_tap_fn fn, void *user);
/* ref: JpegDecoder.Decode (JpegDecoder.cs:509-550). */
int jref_decode(jref_decoder *d);
/* ref: JpegDecoder.Get/SetRestartInterval (JpegDecoder.cs:656-670), ResetTables etc. */
int jref_get_restart_interval(const jref_decoder *d);

/* ---- concrete sinks restated from the reference's callers ---- */

/* ref: apps/JpegDecode/JpegBufferOutputWriter8Bit.cs:28-60 ("O2"): out[(y*W+x)*C+comp], signed clamp to [0,255]. */
typedef struct jref_sink8 {
    int width, height, component_count;
    uint8_t *out;
} jref_sink8;
void jref_sink8_write(void *sink, const int16_t *block, int component_index, int x, int y);

/* ref: tests/JpegLibrary.Tests/Utils/JpegExtendingOutputWriter.cs:30-112 ("O3"): (ushort) clamp to 2^P-1,
 * bit-replicated to 16 bits. */
typedef struct jref_sink16 {
    int width, height, component_count, precision;
    uint16_t *out;
} jref_sink16;
void jref_sink16_write(void *sink, const int16_t *block, int component_index, int x, int y);

/* Raw sink ("O1x"): full-resolution int16 planes, one per component, plane stride = padded width;
 * stores exactly what WriteBlock receives (unclamped, chroma already replicated), clipped to the padded plane. */
typedef struct jref_sink_raw {
    int padded_width, padded_height, component_count;
    int16_t *out; /* [component][padded_height][padded_width] */
} jref_sink_raw;
void jref_sink_raw_write(void *sink, const int16_t *block, int component_index, int x, int y);

/* ---- one-call helpers used by tests / bench (Identify + Decode, like every caller in the reference) ---- */
int jref_decode_to_8bit(const uint8_t *data, size_t len, int component_count, uint8_t *out, size_t out_cap,
                        jref_info *info, char *err, size_t errcap);
int jref_decode_to_16bit(const uint8_t *data, size_t len, int component_count, uint16_t *out, size_t out_cap,
                         jref_info *info, char *err, size_t errcap);

/* ---- CPU baseline of bench.py: mirrors tests/JpegLibrary.Benchmarks/DecoderBenchmark.cs:51-73 (TestJpegLibrary: new
 * JpegDecoder, SetInput, Identify, SetOutputWriter(JpegBufferOutputWriter over a YCbCr8 buffer), Decode) with one
 * independent decoder per host thread, the way a caller of the single-threaded reference fills a many-core box.
 * `threads` native threads (pthreads); thread t decodes images t, t + threads, ...; each owns one output buffer of
 * width * height * component_count bytes, allocated and touched before the clock starts; with `warm` != 0 every thread
 * decodes its first image once before the clock starts.  *seconds = wall time from the common start until the last thread
 * is done; *pixels = pixels decoded inside that time.  Returns 0, or the status of the first failing image. */
int jref_decode_batch_mt(const uint8_t *const *files, const size_t *lens, int n, int component_count, int threads, int warm,
                         double *seconds, uint64_t *pixels, char *err, size_t errcap);
/* rgba != 0: the benchmark's whole sequence -- ConvertYCbCr8ToRgba32 (DecoderBenchmark.cs:67) into a second per-thread buffer
 * behind every Decode() (component_count 3 only) */
int jref_decode_batch_mt_ex(const uint8_t *const *files, const size_t *lens, int n, int component_count, int threads, int warm, int rgba,
                            double *seconds, uint64_t *pixels, char *err, size_t errcap);

/* ---- primitives exported for unit parity tests ---- */
/* ref: ScanDecoder/JpegScanDecoder.cs:50-73 + FastFloatingPointDCT.cs:54-70: one block, zig-zag int16 in,
 * spatial int16 out (row-major), unclamped. quant in zig-zag order. */
void jref_block_dequant_idct_shift(const int16_t *zigzag_coefs, const uint16_t *quant_zigzag, int level_shift,
                                   int16_t *out64);
/* ref: JpegHuffmanDecodingTable.cs:249-390: build the decode table from BITS/HUFFVAL. Returns 1 on success.
 * lookahead: 256 x {size,symbol}; maxcode[18]; valoffset[19]; values[256]. */
int jref_build_huffman(const uint8_t bits[16], const uint8_t *values, int nvalues, uint8_t lookahead_size[256],
                       uint8_t lookahead_symbol[256], uint16_t maxcode[18], uint8_t valoffset[19],
                       uint8_t values_out[256]);

/* ---- JpegOptimizer restatement (oracle/jpegopt.inc; ref: JpegOptimizer.cs).  PARITY UNPINNED, see the file header. */
/* SetInput + Scan + SetOutput + Optimize(strip) on a fresh optimizer; *out is malloc'ed (jref_free). */
int jref_optimize(const uint8_t *in, size_t len, int strip, uint8_t **out, size_t *out_len, char *err, size_t err_cap);
/* Scan() alone: per table (builder-creation order) class, identifier and the 256 symbol counts; freq is [8][256]. */
int jref_optimizer_statistics(const uint8_t *in, size_t len, uint8_t table_class[8], uint8_t identifier[8], uint32_t *freq,
                              int *ntables, char *err, size_t err_cap);
/* JpegHuffmanEncodingTableBuilder.Build(false) for one table: DHT counts / values and GetCode() for all 256 symbols.
 * Returns 0, -1 ("No symbol is recorded."), -2 (a code size beyond the reference's 60-entry array). */
/* Test hook: the order the restated Array.Sort leaves n int32 keys in (perm[i] = original index of the element at position i). */
void jref_net_sort_permutation(const int32_t *keys, int n, int32_t *perm);
int jref_build_optimal_table(const uint32_t freq[256], uint8_t bits_out[16], uint8_t values_out[256], int *nvalues,
                             uint16_t code_out[256], uint8_t length_out[256]);
void jref_free(void *p);
/* the same with JpegOptimizer.MostOptimalCoding / Build(optimal) = the package-merge builder (:289-497) */
int jref_optimize_ex(const uint8_t *in, size_t len, int strip, int most_optimal, uint8_t **out, size_t *out_len, char *err, size_t err_cap);
int jref_build_optimal_table_ex(const uint32_t freq[256], int most_optimal, uint8_t bits_out[16], uint8_t values_out[256], int *nvalues,
                                uint16_t code_out[256], uint8_t length_out[256]);

#ifdef __cplusplus
}
#endif
/* JpegYCbCrToRgbConverter.ConvertYCbCr8ToRgb24 / ConvertYCbCr8ToRgba32 (ref: apps/JpegDecode/JpegYCbCrToRgbConverter.cs:134-206) */
void jref_ycbcr8_to_rgb(const uint8_t *ycbcr, uint8_t *out, size_t count, int bytes_per_pixel);

/* ---- encoder restatement (oracle/jpegenc.c; PARITY UNPINNED: the reference holds no encoder vectors) */
/* JpegStandardQuantizationTable.ScaleByQuality (ref: JpegStandardQuantizationTable.cs:64-87) */
void jref_scale_quant_table(const uint16_t *src_zigzag, int quality, uint16_t *dst_zigzag);
/* code and length of `symbol` in a standard Huffman table: 0 DC lum, 1 AC lum, 2 DC chr, 3 AC chr */
int jref_std_huffman_code(int table, int symbol, int *length);
/* ShiftDataLevel + TransformFDCT + ZigZagAndQuantizeBlock (ref: JpegEncoder.cs:801-826, FastFloatingPointDCT.cs:194-362) */
void jref_fdct_quantize_block(const int16_t *samples, const uint16_t *quant_zigzag, int16_t *out_zigzag);
/* apps/JpegEncode/EncodeAction.cs call sequence (standard tables, optimizeCoding = false) on an interleaved 8-bit buffer */
int jref_encode_8bit(const uint8_t *pixels, int width, int height, int components, int luma_h, int luma_v, int quality,
                     uint8_t *out, size_t cap, size_t *out_len, int16_t *coef_tap);
/* the same with EncodeAction's optimizeCoding switch (tables built from the image's own statistics); 2 = "No symbol is recorded." */
int jref_encode_8bit_tables(const uint8_t *pixels, int width, int height, int components, int luma_h, int luma_v, int quality,
                            const uint16_t *quant_lum, const uint16_t *quant_chr, int optimize_coding, int restart_interval, uint8_t *out,
                            size_t cap, size_t *out_len, int16_t *coef_tap);
int jref_encode_8bit_dri(const uint8_t *pixels, int width, int height, int components, int luma_h, int luma_v, int quality,
                         int optimize_coding, int restart_interval, uint8_t *out, size_t cap, size_t *out_len, int16_t *coef_tap);
int jref_encode_8bit_ex(const uint8_t *pixels, int width, int height, int components, int luma_h, int luma_v, int quality,
                        int optimize_coding, uint8_t *out, size_t cap, size_t *out_len, int16_t *coef_tap);
/* apps/JpegEncode/JpegRgbToYCbCrConverter.ConvertRgb24ToYCbCr8 */
void jref_rgb_to_ycbcr8(const uint8_t *rgb, uint8_t *ycbcr, size_t count);
void jref_rgba_to_ycbcr8(const uint8_t *rgba, uint8_t *ycbcr, size_t count); /* Rgba32 source, three bytes per pixel out */

#endif
