/*
 * include/jpgpu.h -- C ABI of the MI355X-native baseline-JPEG decode path (libjpgpu.so).
 *
 * This is the drop-in boundary for yigolden/JpegLibrary's hot path
 *     JpegDecoder.Decode() -> JpegHuffmanBaselineScanDecoder.ProcessScan -> JpegBlockOutputWriter.WriteBlock
 * Every entry point names the reference interface it replaces ("ref:", paths relative to
 * /root/reference/src/JpegLibrary).  Plain pointers and sizes only; no C++/torch types.
 * INTEGRATION.md shows the C# P/Invoke stubs that bind these symbols.
 *
 * Three levels:
 *   (1) jpgpu_batch_*    whole-file batch decode with device-resident inputs/outputs (throughput path, bench)
 *   (2) jpgpu_decode_scan  one scan with pre-parsed tables (what a C# JpegScanDecoder replacement P/Invokes)
 *   (3) jpgpu_decoder_*  handle-based mirror of the public JpegDecoder API (Identify / SetOutputWriter / Decode)
 *
 * Threading: a ctx owns one device + one HIP stream; calls on one ctx (and objects made from it) must be
 * serialised by the caller; different ctxs are independent (one per GPU / per thread).
 */
#ifndef JPGPU_H
#define JPGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 101: jpgpu_image_result gained error_block (28 bytes, 24 before) and the library writes all of it: a binding compiled against
 * an older header must refuse to run -- compare jpgpu_version() / jpgpu_sizeof_image_result() with its own at load time. */
#define JPGPU_VERSION 101

/* ---- status codes.  1..4 mirror the reference's exception classes so a shim can rethrow them. ---- */
typedef enum jpgpu_status {
    JPGPU_OK = 0,
    JPGPU_ERR_INVALID_DATA = 1,      /* System.IO.InvalidDataException   (ref: ScanDecoder/JpegScanDecoder.cs:39-48) */
    JPGPU_ERR_INVALID_OPERATION = 2, /* System.InvalidOperationException (ref: JpegDecoder.cs:513-518)               */
    JPGPU_ERR_NOT_SUPPORTED = 3,     /* System.NotSupportedException     (ref: JpegDecoder.cs:629)                   */
    JPGPU_ERR_ARGUMENT = 4,          /* System.ArgumentException family                                              */
    JPGPU_ERR_DEVICE = 5,            /* HIP runtime failure (message in jpgpu_last_error)                            */
    JPGPU_ERR_NO_DEVICE = 6,         /* no MI355X visible: the product has NO CPU fallback                           */
    JPGPU_ERR_OUT_OF_MEMORY = 7
} jpgpu_status;

/* Per-image detail codes reported by the device kernels (jpgpu_image_result.detail). */
typedef enum jpgpu_detail {
    JPGPU_DETAIL_NONE = 0,
    JPGPU_DETAIL_INVALID_HUFFMAN_CODE = 1, /* "Invalid Huffman code encountered."  ref: JpegHuffmanDecodingTable.cs:104 */
    JPGPU_DETAIL_MARKER_IN_DATA = 2,       /* "Expect raw data from bit stream. Yet a marker is encountered." ref: ScanDecoder/JpegHuffmanScanDecoder.cs:107 */
    JPGPU_DETAIL_STREAM_ENDED = 3,         /* "The bit stream ended prematurely."  ref: ScanDecoder/JpegHuffmanScanDecoder.cs:109 */
    JPGPU_DETAIL_EXPECT_RESTART = 4,       /* "Expect restart marker."  ref: ScanDecoder/JpegHuffmanBaselineScanDecoder.cs:153 */
    JPGPU_DETAIL_MISSING_TABLE = 5,        /* Huffman / quantization table not defined. ref: ...BaselineScanDecoder.cs:72-81 */
    JPGPU_DETAIL_UNSUPPORTED_FRAME = 6,    /* SOF other than SOF0/SOF1 on this path */
    JPGPU_DETAIL_BAD_HEADER = 7,           /* marker/segment parse failure before the scan */
    JPGPU_DETAIL_EARLY_EOI = 8,            /* not an error: EOI met at a restart boundary, image partially decoded (ref: ...BaselineScanDecoder.cs:145-150) */
    JPGPU_DETAIL_UNEXPECTED_END = 9,       /* "Unexpected end of JPEG data stream."  ref: ScanDecoder/JpegHuffmanProgressiveScanDecoder.cs:249,295,336,350,366,404 */
    JPGPU_DETAIL_NULL_TABLE = 10           /* optimizer: a block needs a Huffman table that was never defined (the reference dereferences null, JpegOptimizer.cs:468-480) */
} jpgpu_detail;

/* Output layouts. */
typedef enum jpgpu_format {
    /* "O2": interleaved u8, out[(y*W+x)*C + c], signed clamp to [0,255], sub-sampled components replicated,
     * clipped to W x H.  Byte-identical to apps/JpegDecode/JpegBufferOutputWriter8Bit.cs:28-60 with C = components. */
    JPGPU_FMT_INTERLEAVED_U8 = 0,
    /* planar u8 at component-native resolution, planes padded to whole MCUs, signed clamp to [0,255] */
    JPGPU_FMT_PLANAR_U8 = 1,
    /* "O1": planar int16 at component-native resolution, planes padded to whole MCUs, UNCLAMPED level-shifted
     * samples == the blocks JpegBlockOutputWriter.WriteBlock receives before chroma expansion
     * (ref: ScanDecoder/JpegHuffmanBaselineScanDecoder.cs:131-134) */
    JPGPU_FMT_PLANAR_I16 = 2,
    /* interleaved 8-bit R,G,B and R,G,B,A (A = 255): the step every caller of the reference runs right after the
     * decode, fused into the writer -- JpegYCbCrToRgbConverter.ConvertYCbCr8ToRgb24 / ConvertYCbCr8ToRgba32
     * (ref: apps/JpegDecode/JpegYCbCrToRgbConverter.cs:134-206, tables :66-118) applied to the "O2" samples; callers:
     * apps/JpegDecode/DecodeAction.cs:71-74, tests/JpegLibrary.Benchmarks/DecoderBenchmark.cs:67.  3-component frames;
     * 1-component frames are converted with Cb = Cr = 128 like DecodeAction.cs:57-65 does.  8-bit precision only. */
    JPGPU_FMT_RGB_U8 = 3,
    JPGPU_FMT_RGBA_U8 = 4,
    /* "O3": the sink of the reference's own decode tests, tests/JpegLibrary.Tests/Utils/JpegExtendingOutputWriter.cs:30-112
     * constructed with componentCount = 4 as HuffmanSequentialDecodeTests.cs:23-43 does: uint16 out[(y*W+x)*4 + c],
     * (ushort)sample clamped to 2^P - 1 (a negative sample becomes the maximum), the P bits spread over 16
     * (FastExpandBits / ExpandBits :86-111), sub-sampled components replicated like WriteBlockSlow, clipped to W x H;
     * channels >= the frame's component count stay zero.  What the golden PNG pairs of the reference hold. */
    JPGPU_FMT_EXTENDED_U16 = 5
} jpgpu_format;

typedef struct jpgpu_ctx jpgpu_ctx;
typedef struct jpgpu_batch jpgpu_batch;
typedef struct jpgpu_decoder jpgpu_decoder;

/* ------------------------------------------------------------------------------------------------ context */

int jpgpu_version(void);
/* sizeof(jpgpu_image_result) as THIS library writes it (the struct every *_result call fills in whole). */
size_t jpgpu_sizeof_image_result(void);
/* Number of visible HIP devices (0 when none; never initialises a context). */
int jpgpu_device_count(void);
/* Creates a context on `device`.  Fails with JPGPU_ERR_NO_DEVICE when no GPU is present. */
int jpgpu_create(int device, jpgpu_ctx **out);
void jpgpu_destroy(jpgpu_ctx *ctx);
/* Last error text of this context (thread-compatible, not thread-safe). ctx may be NULL for create failures. */
const char *jpgpu_last_error(const jpgpu_ctx *ctx);
/* Host threads jpgpu_batch_upload may use for this context (header parsing of the files, copies into the pinned staging
 * ring, full marker walks of the files that need one).  0 = default: min(CPUs granted to the process, 16), or JPGPU_HOST_THREADS.
 * The reference is single-threaded per decoder ("one decoder per thread"); a batch is where the host fans out. */
int jpgpu_set_host_threads(jpgpu_ctx *ctx, int threads);
/* Page-locked host memory for the zero-copy ingest (jpgpu_batch_upload_segments with JPGPU_UPLOAD_PINNED).  The reference's
 * callers own their input (SetInput takes a ReadOnlyMemory / ReadOnlySequence<byte>, JpegDecoder.cs:49-62, filled e.g. by
 * apps/JpegDecode/MemoryPoolBufferWriter.cs); a C# shim either reads its files into jpgpu_host_alloc'd memory or pins its
 * own array (GCHandle) and registers it, so that the DMA engine reads the bytes where they lie.  Registration costs a few
 * microseconds per page: do it once per buffer, not per decode. */
int jpgpu_host_alloc(jpgpu_ctx *ctx, size_t bytes, void **out);
int jpgpu_host_free(jpgpu_ctx *ctx, void *p);
int jpgpu_host_register(jpgpu_ctx *ctx, void *p, size_t bytes);
int jpgpu_host_unregister(jpgpu_ctx *ctx, void *p);
/* Image-per-GPU sharding: of n_items, rank `rank` of `world` takes items first, first + stride, ... (count of
 * them): image i -> GPU i mod G.  Independent contexts, no exchange between them; any out pointer may be NULL. */
void jpgpu_shard(int n_items, int rank, int world, int *first, int *stride, int *count);
const char *jpgpu_status_string(int status);
const char *jpgpu_detail_string(int detail);

/* ------------------------------------------------------------------------------------------------ headers */

/* ref: JpegFrameHeader.cs (SOF payload) */
typedef struct jpgpu_frame_component {
    uint8_t identifier, h, v, tq;
} jpgpu_frame_component;
typedef struct jpgpu_frame {
    uint16_t width, height; /* SamplesPerLine, NumberOfLines */
    uint8_t precision, num_components;
    uint8_t sof; /* marker byte: 0xC0 / 0xC1 */
    uint8_t reserved;
    jpgpu_frame_component comp[4];
} jpgpu_frame;

/* ref: JpegScanHeader.cs (SOS payload) */
typedef struct jpgpu_scan_component {
    uint8_t selector, td, ta, reserved;
} jpgpu_scan_component;
typedef struct jpgpu_scan {
    uint8_t num_components, ss, se, ah, al;
    uint8_t reserved[3];
    jpgpu_scan_component comp[4];
} jpgpu_scan;

/* ref: JpegHuffmanDecodingTable.TryParse input: BITS[16] + HUFFVAL (DHT payload without the Tc/Th byte) */
typedef struct jpgpu_dht {
    uint8_t present;
    uint8_t bits[16];
    uint8_t num_values_minus_0; /* unused, kept for alignment */
    uint16_t num_values;
    uint8_t values[256];
} jpgpu_dht;

/* Geometry of one decoded image inside a batch's output buffer. */
typedef struct jpgpu_plane_info {
    uint64_t offset; /* bytes from the image's out_offset */
    uint32_t width, height, pitch; /* samples; pitch in samples */
} jpgpu_plane_info;
typedef struct jpgpu_image_info {
    int32_t status; /* host-side parse status (jpgpu_status) */
    int32_t detail;
    uint16_t width, height;
    uint8_t precision, num_components, sof, reserved;
    uint32_t restart_interval;
    uint32_t mcus_per_line, mcus_per_column, blocks_per_mcu;
    uint64_t total_blocks;
    uint64_t out_offset, out_bytes;   /* in the batch output buffer */
    uint64_t coef_offset;             /* first block index in the batch coefficient buffer */
    jpgpu_plane_info plane[4];        /* planar formats only */
} jpgpu_image_info;

typedef struct jpgpu_image_result {
    int32_t status; /* jpgpu_status */
    int32_t detail; /* jpgpu_detail */
    uint32_t error_interval;  /* restart interval index of the first failure */
    uint32_t decoded_mcus;    /* MCUs actually decoded (== total unless EARLY_EOI) */
    uint32_t bytes_consumed;  /* entropy bytes up to the terminating marker; same meaning as the reader advance in
                                 ref: ScanDecoder/JpegHuffmanBaselineScanDecoder.cs:167-176 for well-formed streams */
    uint32_t terminator;      /* marker byte that ended the entropy segment (0xD9 = EOI), 0 if the data ran out */
    uint32_t error_block;     /* a sequential scan that failed: index, in scan order, of the block the reference threw in -- it has
                                 called WriteBlock for every block in front of it and for none behind it
                                 (JpegHuffmanBaselineScanDecoder.cs:99-134, 153); 0xFFFFFFFF otherwise */
} jpgpu_image_result;

/* ------------------------------------------------------------------------------------------------ (1) batch
 * Replaces, for a whole set of files at once, the canonical call sequence of every reference caller
 *   new JpegDecoder(); SetInput; Identify; SetOutputWriter(JpegBufferOutputWriter8Bit); Decode()
 * (ref: apps/JpegDecode/DecodeAction.cs:26-56, tests/JpegLibrary.Benchmarks/DecoderBenchmark.cs:51-73).
 */
int jpgpu_batch_create(jpgpu_ctx *ctx, jpgpu_batch **out);
void jpgpu_batch_destroy(jpgpu_batch *b);

/* Host side of SetInput+Identify+the marker loop of Decode up to SOS (ref: JpegDecoder.cs:75-162, 509-617):
 * parses every file's headers/tables, builds device descriptors and uploads the compressed bytes to HBM.
 * `format` is a jpgpu_format.  Images whose headers fail to parse get a per-image status and are skipped.
 * Returns JPGPU_OK if the batch is usable (even if some images failed). */
int jpgpu_batch_upload(jpgpu_batch *b, const uint8_t *const *jpeg, const size_t *len, int n, int format);
/* The same for input that is not one contiguous span per file: file i consists of the next segments_per_file[i] entries of
 * `segments`, in order -- the ReadOnlySequence<byte> of JpegDecoder.SetInput (JpegDecoder.cs:56-62), which the reference reads
 * in place segment by segment (apps/JpegDecode/MemoryPoolBufferWriter.cs:166-174 builds such a sequence,
 * apps/JpegDecode/DecodeAction.cs:81-98 hands it over).  The host looks at the headers only (the first 64 KiB of a
 * multi-segment file are gathered for that; a file whose first scan starts later, or that needs the full marker walks, is
 * linearised on the host as a whole); the entropy-coded bytes travel segment by segment.
 * flags: JPGPU_UPLOAD_PINNED = every segment lies in page-locked memory (jpgpu_host_alloc / jpgpu_host_register): the
 * segments are DMA'd to HBM from where they lie -- no staging copy, no host thread touches the entropy bytes.  Without the
 * flag the segments go through the pinned staging ring like jpgpu_batch_upload's files.
 * Ordering: an upload is ordered behind device work this batch has issued and not yet synchronised (it will not overwrite
 * inputs a running decode of the SAME batch still reads); results of that earlier work are lost with the upload. */
typedef struct jpgpu_segment {
    const uint8_t *data;
    size_t len;
} jpgpu_segment;
#define JPGPU_UPLOAD_PINNED 1u
/* All segments lie inside ONE page-locked allocation (a read buffer the caller fills file after file), every file contiguous,
 * and the bytes BETWEEN the files belong to that allocation too (they are read, never interpreted): the device copy mirrors
 * the arena's layout and the whole span travels as a few large DMAs instead of one per file.  Falls back to one DMA per segment
 * when the files are not contiguous or the span is mostly gaps (more than twice the payload). */
#define JPGPU_UPLOAD_PINNED_ARENA 2u
int jpgpu_batch_upload_segments(jpgpu_batch *b, const jpgpu_segment *segments, const int *segments_per_file, int n, int format,
                                unsigned flags);
/* What the last jpgpu_batch_upload did.  The host reads headers only: both marker loops stop behind the first SOS header and
 * the file is planned as one sequential scan closed by EOI; the bytes behind the header are looked at by the device, which
 * reports the first marker that is not RSTn (what Identify's walk, JpegDecoder.cs:75-162 / JpegReader.cs:120-158, would
 * meet next).  n_header_only = files whose plan was confirmed that way; n_full_walk = files that took the full host walks
 * of Identify and Decode (several scans, progressive, segments or garbage behind the scan, truncated data).  The bytes
 * travel through a pinned staging ring on the context's upload stream, beside whatever runs on its decode stream. */
typedef struct jpgpu_ingest_stats {
    int32_t threads;        /* host threads used */
    int32_t n_header_only;
    int32_t n_full_walk;
    float parse_ms;         /* header-only plans */
    float copy_ms;          /* staging copies + H2D + device verdict */
    float full_walk_ms;     /* full marker walks (+ the reader-position verdicts of the confirmed plans) */
    float layout_ms;        /* descriptors, work lists, device allocations */
    float total_ms;
    int32_t n_pinned_dma;   /* segments sent by DMA straight from the caller's page-locked memory (JPGPU_UPLOAD_PINNED) */
    int32_t n_linearised;   /* multi-segment files the host had to gather as a whole (full marker walks) */
} jpgpu_ingest_stats;
int jpgpu_batch_ingest_stats(const jpgpu_batch *b, jpgpu_ingest_stats *stats);

/* Coefficient hand-off for multi-scan (progressive, SOF2) images -- BASELINE config 5's "coefficient accumulate then single
 * IDCT pass": the caller's progressive entropy decoder accumulates the coefficient store, the GPU runs what
 * JpegHuffmanProgressiveScanDecoder.Dispose (ScanDecoder/JpegHuffmanProgressiveScanDecoder.cs:421-470: dequantise, IDCT,
 * level shift over the MCU grid) and JpegBlockAllocator.Flush (JpegBlockAllocator.cs:120-190) do.  Images are described
 * by their frame header and quantisation tables only (qt[i][tq][64], zig-zag order); then
 * jpgpu_batch_upload_coefficients(i, blocks in MCU scan order: MCU raster, component order, block raster in the MCU)
 * and jpgpu_batch_run_idct. */
int jpgpu_batch_upload_frames(jpgpu_batch *b, const jpgpu_frame *frames, const uint16_t *qt, int n, int format);

/* Launches the device pipeline on the ctx stream (asynchronous):
 *   marker index -> Huffman MCU decode (ref: ...BaselineScanDecoder.cs:51-222) ->
 *   dequantise + float32 IDCT + level shift (ref: ScanDecoder/JpegScanDecoder.cs:50-73, FastFloatingPointDCT.cs:54-185) ->
 *   block output in the batch's format (ref: ...BaselineScanDecoder.cs:225-268 + the sink).
 * With JPGPU_OVERLAP=1 large batches (>= 4 Mi blocks of restart-interval scans) are issued as two halves of images on two
 * streams, the Huffman stage of the second half beside the output stage of the first; the first call after an upload or a
 * jpgpu_batch_stage_ms query and every 8th after it still run serially so that stage times exist.  Off by default: it
 * stopped paying once both stages became HBM-heavy.  Results do not depend on the issue order. */
int jpgpu_batch_decode(jpgpu_batch *b);
/* A progressive (SOF2) file that FAILS still leaves what the reference leaves in the writer's buffer: Decode()'s finally disposes the
 * scan decoder, which transforms and flushes the partial store (JpegDecoder.cs:545-549).  jpgpu_batch_result reports the failure,
 * jpgpu_batch_download_output of that image returns the partial picture (the step is issued a second time for such a batch, once
 * per upload; sequential files write nothing on failure that the reference would not). */
/* Individual stages, for stage-level parity tests and profiling. */
int jpgpu_batch_run_entropy(jpgpu_batch *b); /* marker index + Huffman -> coefficient buffer */
int jpgpu_batch_run_idct(jpgpu_batch *b);    /* coefficient buffer -> output */
int jpgpu_batch_sync(jpgpu_batch *b);

int jpgpu_batch_size(const jpgpu_batch *b);
int jpgpu_batch_image_info(const jpgpu_batch *b, int i, jpgpu_image_info *info);
/* Valid after jpgpu_batch_sync. */
int jpgpu_batch_result(jpgpu_batch *b, int i, jpgpu_image_result *res);

/* Device pointers (HBM) of the whole-batch buffers; outputs stay resident for downstream GPU consumers.
 * Samples the reference would leave as the caller's buffer held them -- MCUs behind an early EOI
 * (JpegHuffmanBaselineScanDecoder.cs:144-150), components no scan writes, images without any scan -- read as zero here:
 * the batch owns the buffer, and zero is what a freshly allocated managed array holds.  (jpgpu_decode_scan, which
 * decodes over the caller's own samples, leaves them alone.) */
void *jpgpu_batch_output_device(const jpgpu_batch *b, uint64_t *total_bytes);
void *jpgpu_batch_coefficients_device(const jpgpu_batch *b, uint64_t *total_blocks);
/* Copies one image's output / coefficient blocks (int16[blocks][64], zig-zag order, MCU scan order) to the host.  (A progressive
 * frame whose Dispose() is taken literally -- component slots that do not cover every component once, the partial flush of a
 * failed file -- is transformed IN its store: behind the output stage its "coefficients" are samples; the pass runs once per
 * entropy stage, a second jpgpu_batch_run_idct / jpgpu_progressive_dispose flushes the same samples again.) */
int jpgpu_batch_download_output(jpgpu_batch *b, int i, void *dst, size_t cap);
int jpgpu_batch_download_coefficients(jpgpu_batch *b, int i, int16_t *dst, size_t cap_blocks);
/* Overwrites one image's coefficient blocks from the host (IDCT-stage parity tests; config-5 style accumulate-then-IDCT). */
int jpgpu_batch_upload_coefficients(jpgpu_batch *b, int i, const int16_t *src, size_t nblocks);

/* hipEvent timings (ms) on the ctx stream over the jpgpu_batch_decode calls issued since the previous query (synchronises):
 * ms[0] marker index, ms[1] Huffman, ms[2] IDCT+output, averaged over the calls that ran serially (see jpgpu_batch_decode);
 * ms[3] whole pipeline, averaged over all calls. */
int jpgpu_batch_stage_ms(jpgpu_batch *b, float ms[4]);
/* Synchronisation rounds the self-synchronising DRI = 0 decoder needed in the most recent decode (0 = not used).  The rounds
 * are enqueued ahead and checked on the device (jpgpu_batch_decode does not wait for them): valid after jpgpu_batch_sync /
 * _result / _download_*. */
int jpgpu_batch_subseq_rounds(const jpgpu_batch *b);
/* The partial flush of FAILING progressive files (the reference's Decode() disposes the scan decoder in its `finally`, which
 * transforms and flushes the store as the failing scan left it, JpegDecoder.cs:545-549): on by default -- the first call that
 * needs a status or an output after a decode in which a progressive frame failed issues those frames once more, scan by scan
 * (the other images of the batch are not touched).  on = 0: callers that only want the statuses and the outputs of the images
 * that decoded leave it out (also: environment JPGPU_NO_PARTIAL_FLUSH); the failed frames' outputs are then unspecified. */
int jpgpu_batch_set_partial_flush(jpgpu_batch *b, int on);
/* Partial-flush replays issued for this batch so far (0: no progressive frame has failed, or the replay is switched off). */
int jpgpu_batch_progressive_replays(const jpgpu_batch *b);
/* Waits of this batch's jpgpu_batch_sync calls behind which a workgroup of the one-pass marker index (a workgroup per 16 KiB of
 * entropy data that finds the running sums of the data in front of it by looking back at what the workgroups there have published)
 * had run out of patience and counted the data in front of it itself.  Not expected to be anything but 0 -- the wait is bounded so
 * that nothing can hang -- and of no consequence for the results.  Valid after jpgpu_batch_sync. */
int jpgpu_batch_marker_fallbacks(const jpgpu_batch *b);
/* Times the enqueued rounds did not reach the fixed point (the synchronising call then issued the step again with the host
 * reading the counts between rounds, and the upload's later decodes stay that way).  Valid after jpgpu_batch_sync. */
int jpgpu_batch_subseq_fallbacks(const jpgpu_batch *b);
/* Times the single-launch progressive path gave up waiting inside the kernel (its scans follow each other's progress, which
 * relies on workgroups being dispatched in list order) and the step was re-issued scan level by scan level.  Valid after
 * jpgpu_batch_result. */
int jpgpu_batch_progressive_fallbacks(const jpgpu_batch *b);
/* Total entropy-segment bytes / blocks / pixels of the successfully parsed images. */
int jpgpu_batch_totals(const jpgpu_batch *b, uint64_t *compressed_bytes, uint64_t *blocks, uint64_t *pixels,
                       uint64_t *output_bytes);

/* ------------------------------------------------------------------------------------------------ (1b) several devices
 * The multi-GPU driver of SURVEY.md 7 step 6 / 8e inside the library: one context, one batch and one host thread per listed
 * device, image i of a call on device slot i mod G (jpgpu_shard's rule), no exchange between the devices.  Replaces the
 * caller-side loop "one JpegDecoder per thread" (Threading) for a whole file list.  A device may be listed more than
 * once (two independent contexts on it).  jpgpu_multi_decode uploads and decodes every shard concurrently and returns when
 * all of them are done; results and outputs are then read through the shard's batch (jpgpu_batch_result,
 * jpgpu_batch_output_device / _download_output, ...) at the local index jpgpu_multi_locate gives. */
typedef struct jpgpu_multi jpgpu_multi;
int jpgpu_multi_create(const int *devices, int n_devices, jpgpu_multi **out);
void jpgpu_multi_destroy(jpgpu_multi *m);
int jpgpu_multi_devices(const jpgpu_multi *m);
const char *jpgpu_multi_last_error(const jpgpu_multi *m);
/* Decodes n files in `format`; returns the first non-OK call status of any shard (per-image failures are per-image results, as
 * with jpgpu_batch_*).  upload_ms / decode_ms (may be NULL): the slowest shard's time in each phase. */
int jpgpu_multi_decode(jpgpu_multi *m, const uint8_t *const *jpeg, const size_t *len, int n, int format, double *upload_ms,
                       double *decode_ms);
/* The same in two halves, for callers that feed the devices continuously.  Every slot owns TWO batches: jpgpu_multi_submit
 * uploads call k's shards into the idle one of each slot, launches their decode and returns while the devices work;
 * jpgpu_multi_wait(ticket) blocks until that call's decodes are done.  Submitting call k+1 before waiting for call k puts its
 * host parse + H2D beside call k's decode (upload stream / decode stream of each context); at most two calls may be in
 * flight, and a ticket's outputs stay valid until the second submit after it.  flags: JPGPU_UPLOAD_PINNED as in
 * jpgpu_batch_upload_segments (every file one page-locked segment).  The host crew of slot s is (CPUs granted to the
 * process) / G threads -- G slots share the machine -- unless jpgpu_set_host_threads / JPGPU_HOST_THREADS says otherwise.
 * A submit that fails (any slot's upload or launch) takes no ticket (*ticket = -1): the slots that did launch are waited
 * for before it returns, nothing stays in flight and the next submit may use the same batches. */
int jpgpu_multi_submit(jpgpu_multi *m, const uint8_t *const *jpeg, const size_t *len, int n, int format, unsigned flags, int *ticket);
int jpgpu_multi_wait(jpgpu_multi *m, int ticket, double *upload_ms, double *decode_ms);
jpgpu_batch *jpgpu_multi_batch_of(jpgpu_multi *m, int ticket, int slot);
/* Where image i of the last jpgpu_multi_decode went: the device slot (index into `devices`) and its index in that slot's batch. */
int jpgpu_multi_locate(const jpgpu_multi *m, int i, int *slot, int *local_index);
jpgpu_batch *jpgpu_multi_batch(jpgpu_multi *m, int slot); /* of the most recent jpgpu_multi_decode / _submit */
jpgpu_ctx *jpgpu_multi_context(jpgpu_multi *m, int slot);

/* ------------------------------------------------------------------------------------------------ (2) per scan
 * Replaces JpegScanDecoder.ProcessScan(ref JpegReader, JpegScanHeader) for SOF0/SOF1
 * (ref: ScanDecoder/JpegScanDecoder.cs:12-36, ScanDecoder/JpegHuffmanBaselineScanDecoder.cs:51-177).
 * The scan decoder's pulled inputs become arguments: GetRestartInterval (JpegDecoder.cs:656),
 * GetHuffmanTable (:869), GetQuantizationTable (:910).  qt is in zig-zag order like
 * JpegQuantizationTable.Elements (JpegQuantizationTable.cs:47); qt_present[i] = !IsEmpty.
 * dht[0][id] are DC tables, dht[1][id] AC tables.  `entropy` points just after the SOS segment
 * (reader.RemainingBytes); *bytes_consumed is what the reference advances the reader by.
 * Output is written to host memory `out` (cap bytes) in `format`; what `out` holds on entry is the canvas: samples the
 * scan does not write (other components of a non-interleaved scan, MCUs behind an early EOI) keep their values, as
 * with the reference's writers (JpegBufferOutputWriter8Bit.cs:28-60).
 */
int jpgpu_decode_scan(jpgpu_ctx *ctx, const jpgpu_frame *frame, const jpgpu_scan *scan, const uint16_t qt[4][64],
                      const uint8_t qt_present[4], const jpgpu_dht dht[2][4], uint16_t restart_interval,
                      const uint8_t *entropy, size_t len, int format, void *out, size_t cap,
                      jpgpu_image_result *result, size_t *bytes_consumed);

/* ------------------------------------------------------------------------------------------------ (2b) per scan, SOF2
 * Replaces JpegHuffmanProgressiveScanDecoder behind JpegScanDecoder.Create(SOF2, ...) (ScanDecoder/JpegScanDecoder.cs:18-36):
 * what JpegDecoder.ProcessFrameHeader / ProcessScanHeader / Decode's finally call on it (JpegDecoder.cs:562-570, 592-599,
 * 545-549), so that a C# JpegDecoder keeps its own marker loop for progressive files too.
 *   jpgpu_progressive_begin    the constructor (ScanDecoder/JpegHuffmanProgressiveScanDecoder.cs:23-55): MCU geometry, and
 *                              JpegBlockAllocator.Allocate (JpegBlockAllocator.cs:35-84) -- the zeroed coefficient store, in HBM,
 *                              where it stays between the calls.
 *   jpgpu_progressive_scan     ProcessScan (:57-90): resolves the scan's components against the tables passed (the decoder's
 *                              registry at this SOS: GetHuffmanTable / GetQuantizationTable, JpegDecoder.cs:869, 910) and the
 *                              restart interval in force NOW (GetRestartInterval read per scan, :78 -- it may change between
 *                              scans), then decodes the scan into the store on the GPU before it returns: the status is this
 *                              scan's (reference exception classes and messages, as jpgpu_decode_scan).  `entropy` = the bytes
 *                              behind the SOS header; the segment ends at the first marker that is not RSTn.  *bytes_consumed is
 *                              always 0: the reference's progressive ProcessScan never advances the outer reader, Decode's
 *                              TryReadMarker finds the next marker by itself (JpegReader.cs:120-158).
 *   jpgpu_progressive_dispose  Dispose (:421-470): dequantise + IDCT + level shift over the MCU grid with the quantisation tables
 *                              the LAST scans left in the decoder's component slots, then
 *                              JpegBlockAllocator.Flush (JpegBlockAllocator.cs:120-190) into `out` in `format`
 *                              (jpgpu_progressive_output_size says how large), or, _to_writer, as WriteBlock calls in Flush's order.
 *                              Scan orders whose slots do not cover every component once are disposed the way the reference does it
 *                              (a component transformed twice, another never: dispose_pass_kernel); a session without any scan
 *                              flushes the zeroed store.  After a FAILING jpgpu_progressive_scan the store holds what the
 *                              reference's ProcessScan left when it threw -- every coefficient and correction bit in front of the
 *                              throw, nothing behind it: the failing scan is issued once more from a device copy of the store taken
 *                              in front of it, coefficient by coefficient up to the failing restart interval -- so a dispose behind
 *                              it flushes what Decode()'s `finally` flushes (JpegDecoder.cs:545-549), like the batch entry points.
 */
typedef struct jpgpu_progressive jpgpu_progressive;
int jpgpu_progressive_begin(jpgpu_ctx *ctx, const jpgpu_frame *frame, jpgpu_progressive **out);
int jpgpu_progressive_scan(jpgpu_progressive *p, const jpgpu_scan *scan, const uint16_t qt[4][64], const uint8_t qt_present[4],
                           const jpgpu_dht dht[2][4], uint16_t restart_interval, const uint8_t *entropy, size_t len,
                           jpgpu_image_result *result, size_t *bytes_consumed);
int jpgpu_progressive_output_size(jpgpu_progressive *p, int format, size_t *bytes);
int jpgpu_progressive_dispose(jpgpu_progressive *p, int format, void *out, size_t cap);
void jpgpu_progressive_destroy(jpgpu_progressive *p);

/* ------------------------------------------------------------------------------------------------ (3) decoder
 * Handle-based mirror of the public JpegDecoder surface (ref: JpegDecoder.cs).  Names follow the reference.
 */
/* ref: JpegBlockOutputWriter.WriteBlock (JpegBlockOutputWriter.cs:17): 64 int16 row-major, unclamped,
 * full-resolution pixel coordinates; same call order as the reference (MCU raster, scan-component order,
 * block raster in MCU, sub-block raster for expanded chroma). */
typedef void (*jpgpu_write_block_fn)(void *user, const int16_t *block, int component_index, int x, int y);

/* jpgpu_progressive_dispose for an arbitrary JpegBlockOutputWriter: the IDCT pass on the GPU, then the WriteBlock calls of
 * JpegBlockAllocator.Flush (component, block row, block column; sub-sampled components expanded) replayed on the host. */
int jpgpu_progressive_dispose_to_writer(jpgpu_progressive *p, jpgpu_write_block_fn fn, void *user);

/* ctx may be NULL for host-only use (Identify / metadata / tables); Decode then returns JPGPU_ERR_NO_DEVICE. */
int jpgpu_decoder_create(jpgpu_ctx *ctx, jpgpu_decoder **out);          /* new JpegDecoder()                */
void jpgpu_decoder_destroy(jpgpu_decoder *d);
const char *jpgpu_decoder_last_error(const jpgpu_decoder *d);
int jpgpu_decoder_set_input(jpgpu_decoder *d, const uint8_t *data, size_t len); /* SetInput  :49-62    */
int jpgpu_decoder_identify(jpgpu_decoder *d, int load_quantization_tables, int *stream_length); /* Identify :75-105 */
int jpgpu_decoder_try_estimate_quality(jpgpu_decoder *d, float *quality);       /* TryEstimateQuanlity :169-196 */
int jpgpu_decoder_width(const jpgpu_decoder *d);                                 /* Width  :383 (-1: no frame header) */
int jpgpu_decoder_height(const jpgpu_decoder *d);                                /* Height :388 */
int jpgpu_decoder_precision(const jpgpu_decoder *d);                             /* Precision :393 */
int jpgpu_decoder_number_of_components(const jpgpu_decoder *d);                  /* NumberOfComponents :398 */
int jpgpu_decoder_start_of_frame(const jpgpu_decoder *d);                        /* StartOfFrame :43 */
int jpgpu_decoder_get_maximum_horizontal_sampling(jpgpu_decoder *d);             /* :413 */
int jpgpu_decoder_get_maximum_vertical_sampling(jpgpu_decoder *d);               /* :436 */
int jpgpu_decoder_get_horizontal_sampling(jpgpu_decoder *d, int component_index); /* :462 */
int jpgpu_decoder_get_vertical_sampling(jpgpu_decoder *d, int component_index);   /* :481 */
int jpgpu_decoder_get_restart_interval(const jpgpu_decoder *d);                  /* GetRestartInterval :656 */
int jpgpu_decoder_set_restart_interval(jpgpu_decoder *d, int restart_interval);  /* SetRestartInterval :662 */
int jpgpu_decoder_load_tables(jpgpu_decoder *d, const uint8_t *data, size_t len); /* LoadTables :313-363 */
/* SetOutputWriter :501 -- generic writer: blocks are decoded on the GPU (PLANAR_I16) and replayed on the host. */
int jpgpu_decoder_set_output_writer(jpgpu_decoder *d, jpgpu_write_block_fn fn, void *user);
/* SetOutputWriter with the reference's stock 8-bit sink (apps/JpegDecode/JpegBufferOutputWriter8Bit.cs):
 * the interleaved buffer is produced directly on the GPU (INTERLEAVED_U8) and copied into `out`. */
int jpgpu_decoder_set_output_buffer8(jpgpu_decoder *d, int width, int height, int component_count, uint8_t *out,
                                     size_t cap);
int jpgpu_decoder_decode(jpgpu_decoder *d);                                      /* Decode :509-550 */
/* The TIFF-style surface (JPEG-in-TIFF keeps tables, frame header and strips apart; SURVEY 5 "checkpoint / resume" row): the
 * caller sets the pieces itself instead of letting Decode()'s marker loop find them, then hands over one scan's entropy data. */
int jpgpu_decoder_set_start_of_frame(jpgpu_decoder *d, int marker);              /* StartOfFrame { set; } :43 (0xC0 / 0xC1 / 0xC2) */
int jpgpu_decoder_set_frame_header(jpgpu_decoder *d, const jpgpu_frame *frame);  /* SetFrameHeader :404-407 (frame->sof is NOT applied) */
/* SetHuffmanTable(JpegHuffmanDecodingTable) :793-815: replaces the entry with the same class (0 = DC, 1 = AC) and identifier */
int jpgpu_decoder_set_huffman_table(jpgpu_decoder *d, int table_class, int identifier, const uint8_t bits[16], const uint8_t *values,
                                    int num_values);
/* SetQuantizationTable(JpegQuantizationTable) :840-861: 64 elements in zig-zag order (JpegQuantizationTable.cs:47) */
int jpgpu_decoder_set_quantization_table(jpgpu_decoder *d, int element_precision, int identifier, const uint16_t *zigzag64);
int jpgpu_decoder_clear_huffman_table(jpgpu_decoder *d);                         /* ClearHuffmanTable :768-771 */
int jpgpu_decoder_clear_quantization_table(jpgpu_decoder *d);                    /* ClearQuantizationTable :784-787 */
/* ProcessScan(ref JpegReader, JpegScanHeader) :624-632: JpegScanDecoder.Create(StartOfFrame, this, GetFrameHeader()) -- the restart
 * interval is latched at this moment -- one ProcessScan over `entropy` (reader.RemainingBytes), Dispose.  The output
 * goes to the writer set with jpgpu_decoder_set_output_writer / _set_output_buffer8; *bytes_consumed = the reader's advance. */
int jpgpu_decoder_process_scan(jpgpu_decoder *d, const jpgpu_scan *scan, const uint8_t *entropy, size_t len, size_t *bytes_consumed);
void jpgpu_decoder_reset(jpgpu_decoder *d);                                      /* Reset :930 */
void jpgpu_decoder_reset_input(jpgpu_decoder *d);                                /* ResetInput :941 */
void jpgpu_decoder_reset_header(jpgpu_decoder *d);                               /* ResetHeader :949 */
void jpgpu_decoder_reset_tables(jpgpu_decoder *d);                               /* ResetTables :960 */
void jpgpu_decoder_reset_output_writer(jpgpu_decoder *d);                        /* ResetOutputWriter :975 */

/* ------------------------------------------------------------------------------------------------ (4) encoder
 * The step on the other side of the wire format: JpegEncoder.Encode() (ref: JpegEncoder.cs:255-291)
 * for a batch of images, with the call sequence of apps/JpegEncode/EncodeAction.cs:38-63 (optimizeCoding = false):
 *   SetQuantizationTable(ScaleByQuality(luminance, 0, quality)), SetQuantizationTable(ScaleByQuality(chrominance, 1, quality)),
 *   SetHuffmanTable(DC/AC, 0/1, standard tables), AddComponent(1, 0, 0, 0, luma_h, luma_v) [, AddComponent(2 | 3, 1, 1, 1, 1, 1)],
 *   SetInputReader(JpegBufferInputReader(width, height, components, pixels)), Encode().
 * The output is the byte stream the reference writes (SOI, DQT, SOF0, DHT, SOS, entropy data, EOI).
 * input_rgb == 1: pixels are R,G,B and JpegRgbToYCbCrConverter.ConvertRgb24ToYCbCr8 (apps/JpegEncode/
 * JpegRgbToYCbCrConverter.cs:64-96) is applied first, like EncodeAction.cs:31-36 does.  input_rgb == 2: pixels are Rgba32
 * (four bytes each, the alpha byte stepped over) and ConvertRgba32ToYCbCr8 is applied, like the reference's EncoderBenchmark does
 * (tests/JpegLibrary.Benchmarks/EncoderBenchmark.cs:93, ColorConverters/JpegRgbToYCbCrConverter.cs:95-124): same tables. */
typedef struct jpgpu_encode_params {
    int32_t width, height;
    int32_t components;      /* encoded components: 1 or 3 == samples per pixel of the input buffer (input_rgb == 2: four bytes per pixel) */
    int32_t luma_h, luma_v;  /* sampling factors of the first component (1, 2 or 4); the others are 1 x 1 */
    int32_t quality;         /* 1..100, JpegStandardQuantizationTable.ScaleByQuality */
    int32_t input_rgb;       /* 0, 1 or 2, see above */
    int32_t optimize_coding; /* 1 = EncodeAction's optimizeCoding (EncodeAction.cs:40-46): Huffman tables built from the image's own
                                statistics (TransformBlocks / BuildHuffmanTables / WritePreparedScanData, JpegEncoder.cs:264-274);
                                2 = the same with JpegEncoder.MostOptimalCoding (:43) */
    int32_t restart_interval; /* 0 = none = everything the reference's encoder can write.  n > 0 is an EXTENSION (SURVEY.md 8f N3,
                                "+ DRI emission"; JpegEncoder has no restart support): a DRI segment in front of SOF0 and, in front
                                of every MCU whose index is a non-zero multiple of n, one-bit padding to a byte boundary, RSTm
                                (m modulo 8) and the DC predictors back at zero (T.81 B.2.4.4 / E.1.4: what libjpeg writes, and what
                                JpegDecoder reads back: ScanDecoder/JpegHuffmanBaselineScanDecoder.cs:139-163).  1..65535 */
} jpgpu_encode_params;
typedef struct jpgpu_encoder jpgpu_encoder;

int jpgpu_encoder_create(jpgpu_ctx *ctx, jpgpu_encoder **out);
void jpgpu_encoder_destroy(jpgpu_encoder *e);
/* SetInputReader for n images (host parse of nothing: pixels go to HBM as they are) */
int jpgpu_encoder_upload(jpgpu_encoder *e, const uint8_t *const *pixels, const jpgpu_encode_params *params, int n);
/* SetQuantizationTable(new JpegQuantizationTable(0, identifier, elements)) for image i (JpegEncoder.cs:102-126): the caller's own
 * table instead of the standard one scaled by `quality`.  identifier 0 = the first component's table, 1 = the other components';
 * 64 elements in zig-zag order, element precision 0, so 1..255 each (the segment stores bytes; 0 would divide by zero in
 * ZigZagAndQuantizeBlock).  Between jpgpu_encoder_upload and jpgpu_encoder_encode. */
int jpgpu_encoder_set_quantization_table(jpgpu_encoder *e, int i, int identifier, const uint16_t *zigzag64);
/* Encode(): FDCT + quantise, Huffman code lengths, bit emission, byte stuffing -- all on the device */
int jpgpu_encoder_encode(jpgpu_encoder *e);
/* Device time (ms, HIP events) of the last jpgpu_encoder_encode by stage: ms[0] pixels -> quantised zig-zag blocks (ReadBlock, FDCT,
 * ZigZagAndQuantizeBlock: JpegEncoder.cs:662-741, 812-826; with optimize_coding also GatherBlockStatistics :552-597), ms[1] code lengths
 * and bit offsets, ms[2] bit emission (EncodeBlock :828-925), ms[3] byte stuffing (JpegWriter.cs:133-232), ms[4] their sum. */
int jpgpu_encoder_stage_ms(jpgpu_encoder *e, float ms[5]);
/* How the entropy stage of this encoder's jpgpu_encoder_encode calls ran: *one_pass = calls that counted and emitted the bits in ONE
 * pass over the blocks (uploads of two or more images without restart intervals: EncodeBlock, JpegEncoder.cs:828-925, run once per
 * block instead of twice), *fell_back = those of them that had to issue the two-kernel form behind it (a workgroup's stretch of the
 * stream beyond the on-chip buffer: noise at quality 100).  The streams are the same bytes either way.  Either pointer may be NULL. */
int jpgpu_encoder_emit_passes(const jpgpu_encoder *e, int *one_pass, int *fell_back);
int jpgpu_encoder_encoded_size(const jpgpu_encoder *e, int i, size_t *bytes);
int jpgpu_encoder_download(jpgpu_encoder *e, int i, void *dst, size_t cap);                       /* the IBufferWriter's content */
void *jpgpu_encoder_output_device(const jpgpu_encoder *e, int i, size_t *bytes);                  /* stream i, resident in HBM */
/* quantised zig-zag blocks in encoding order (what ZigZagAndQuantizeBlock produced, JpegEncoder.cs:812-826) */
int jpgpu_encoder_download_coefficients(jpgpu_encoder *e, int i, int16_t *dst, size_t cap_blocks);

/* ------------------------------------------------------------------------------------------------
 * (5) Optimizer -- replaces JpegOptimizer (ref: JpegOptimizer.cs) for single-scan baseline files:
 *     SetInput (:57-63) + Scan() (:66-153) + SetOutput (:523-526) + Optimize(strip) (:540-648).  The marker walks run on
 *     the host, the two symbol passes (ProcessScanBaseline :360-463, CopyScanBaseline :719-829) on the device, the new
 *     Huffman tables are JpegHuffmanEncodingTableBuilder.Build(false) (JpegHuffmanEncodingTableBuilder.cs:68-175).
 *     Not supported (JPGPU_ERR_NOT_SUPPORTED): more than one scan, progressive frames.
 * ---------------------------------------------------------------------------------------------- */
typedef struct jpgpu_optimizer jpgpu_optimizer;
int jpgpu_optimizer_create(jpgpu_ctx *ctx, jpgpu_optimizer **out);
void jpgpu_optimizer_destroy(jpgpu_optimizer *o);
/* SetInput for n files (host marker walks + H2D); strip = Optimize(strip)'s argument */
int jpgpu_optimizer_upload(jpgpu_optimizer *o, const uint8_t *const *jpeg, const size_t *len, int n, int strip);
/* JpegOptimizer.MostOptimalCoding (:38): Build(optimal = true), the package-merge builder */
int jpgpu_optimizer_set_most_optimal_coding(jpgpu_optimizer *o, int on);
/* Scan() + Optimize(strip) for every file */
int jpgpu_optimizer_run(jpgpu_optimizer *o);
/* status of file i with the reference's exception classes; out_len = bytes Optimize() wrote */
int jpgpu_optimizer_result(jpgpu_optimizer *o, int i, jpgpu_image_result *res, size_t *out_len);
int jpgpu_optimizer_download(jpgpu_optimizer *o, int i, void *dst, size_t cap);  /* the IBufferWriter's content */
/* what Scan() counted for table `table` of file i, in the order the reference creates its table builders (:381-413) */
int jpgpu_optimizer_statistics(const jpgpu_optimizer *o, int i, int table, uint8_t *table_class, uint8_t *identifier, uint32_t *counts);
int jpgpu_optimizer_last_ms(const jpgpu_optimizer *o, float *ms);  /* device time of the last run (HIP events) */
/* JpegHuffmanEncodingTableBuilder.Build(false) for one table: DHT counts and values, and GetCode() for all 256 symbols */
/* The order .NET's Array.Sort(T[], Comparison<T>) / List<T>.Sort(Comparison<T>) leaves n elements in when the comparison looks at
 * an int32 key alone (ascending): perm[i] = original index of the element that ends up at position i.  That sort is not stable,
 * and the reference orders equal-length symbols (JpegHuffmanEncodingTableBuilder.cs:171) and its package-merge nodes (:352,
 * :374, :395) with it, so the DHT bytes the optimizer and the encoder write depend on the runtime's algorithm (restated from
 * ArraySortHelper<T>.IntrospectiveSort).  Exposed so that the restatement can be held against independent ones. */
int jpgpu_net_sort_permutation(const int32_t *keys, int n, int32_t *perm);
int jpgpu_build_optimal_huffman_table(const uint32_t *counts, int most_optimal, uint8_t *bits, uint8_t *values, int *num_values,
                                      uint16_t *code, uint8_t *length);

#ifdef __cplusplus
}
#endif
#endif /* JPGPU_H */
