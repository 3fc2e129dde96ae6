// jpeglibrary_amd/csrc/encode_kernels.h -- device structures and launch wrappers of the encoder kernels (encode_kernels.hip)
#pragma once
#include <hip/hip_runtime_api.h>
#include <stddef.h>
#include <stdint.h>

namespace jpgpu {

// One image to encode: the reference's JpegEncoder state after the EncodeAction call sequence
// (ref: apps/JpegEncode/EncodeAction.cs:38-63): luma component with sampling (luma_h, luma_v) and table 0, two chroma
// components 1x1 with table 1 (or a single component).
struct alignas(16) DevEncImage {
    uint64_t px_off;    // input pixels, byte offset into the batch's pixel buffer (interleaved, in_components per pixel)
    uint64_t coef_off;  // first block in the coefficient buffer (blocks of 64 int16, zig-zag, MCU order)
    uint64_t raw_off;   // raw (unstuffed) entropy bytes, byte offset into the raw buffer (256-byte aligned)
    uint64_t out_off;   // finished stream, byte offset into the output buffer
    uint32_t width, height;
    uint32_t in_components;  // samples per input pixel
    uint32_t components;     // 1 or 3 encoded components
    uint32_t luma_h, luma_v;
    uint32_t mcus_per_line, mcus_per_column, bpm, total_blocks;
    uint32_t header_len;  // bytes of SOI..SOS already placed at out_off
    uint32_t chunk_off;   // first entry in the per-chunk FF counters (stuffing)
    uint32_t input_rgb;   // 1: pixels are R,G,B and are converted like JpegRgbToYCbCrConverter.ConvertRgb24ToYCbCr8
    uint32_t table_base;  // first of the image's 4 tables (DC0, AC0, DC1, AC1) in the table array; 0 = the standard tables.
                          // != 0 = optimizeCoding (tables built from the image): the TransformBlocks / allocator semantics apply
    uint32_t work_first;  // first workgroup of this image in the per-256-blocks work list (block_bits / emit)
    uint32_t smp_off_256; // gathered samples of the image (E1a -> E1b), offset into the sample buffer in units of 256 bytes
    uint32_t restart_interval;  // MCUs per restart interval, 0 = none (see jpgpu_encode_params)
    uint32_t n_units;     // lanes of block_bits / emit for this image: its blocks, or its restart intervals
    uint32_t hdr_off;     // the image's SOI..SOS bytes in the batch's header buffer (place_headers_kernel puts them at out_off)
    uint32_t reserved0;
    int32_t r2y[8];       // Fix() factors of the RGB -> YCbCr tables (host: rgb_ycc_factors)
    uint16_t quant[2][64];  // zig-zag quantisation tables: luma, chroma
};
static_assert(sizeof(DevEncImage) % 16 == 0, "DevEncImage must be a multiple of 16 bytes");

// JpegHuffmanEncodingTable.GetCode by symbol (ref: JpegHuffmanEncodingTable.cs:94-100); len 0 = no code
struct EncHuffTable {
    uint16_t code[256];
    uint8_t len[256];
};

constexpr int kEncMcusPerWg = 128;  // MCUs (lanes) per workgroup of fdct_quant_kernel

struct EncWork {
    uint32_t image;
    uint32_t first;  // first MCU (kEncMcusPerWg per workgroup) / block (256) / stuffing chunk index
};

// E1: fdct_fused_kernel<H, V> for the images enc_image_fused_shape() gives a shape (1: 2 x 2, 2: 2 x 1, 3: 1 x 1; fused_shapes:
// bit s set = the batch holds images of shape s), for the others (shape 0; any_other) E1a (pixel pass into `samples`,
// enc_sample_stride bytes per MCU) + E1b (FDCT + quantisation)
hipError_t launch_fdct_quant(hipStream_t stream, const uint8_t *pixels, const DevEncImage *images, const EncWork *work, int n_work,
                             uint8_t *samples, int16_t *coefs, size_t max_record_bytes,  // (the largest enc_sample_bytes_per_mcu of the batch)
                             uint32_t fused_shapes, bool any_other);
int enc_image_fused_shape(const DevEncImage &im);
size_t enc_sample_bytes_per_mcu(uint32_t luma_h, uint32_t luma_v, uint32_t components);
hipError_t launch_block_bits(hipStream_t stream, const DevEncImage *images, const EncWork *work, int n_work, const EncHuffTable *tables,
                             const int16_t *coefs, uint32_t *bits, int n_images, uint32_t *wg_bits, uint64_t *wg_base, uint64_t *raw_bits);
// optimizeCoding: GatherBlockStatistics for every block (ref: JpegEncoder.cs:552-597) -> hist[image][4][256]
hipError_t launch_block_stats(hipStream_t stream, const DevEncImage *images, const EncWork *work, int n_work, const int16_t *coefs,
                              uint32_t *hist);
hipError_t launch_emit(hipStream_t stream, const DevEncImage *images, const EncWork *work, int n_work, const EncHuffTable *tables,
                       const int16_t *coefs, const uint32_t *bits, const uint64_t *wg_base, const uint64_t *raw_bits, uint8_t *raw,
                       uint32_t *marks, uint32_t lds_words);
// E2 + E3 as one pass over the blocks (round 5; images without restart intervals): see bits_emit_kernel
size_t enc_chain_bytes(int n_work);
hipError_t launch_bits_emit(hipStream_t stream, const DevEncImage *images, const EncWork *work, int n_work, const EncHuffTable *tables,
                            const int16_t *coefs, void *chain, uint32_t *ctl, uint8_t *raw, uint64_t *raw_bits, uint32_t lds_words,
                            const uint32_t *order /* the work list's entries by (place inside the image, image) */);
// the LDS buffer a workgroup of emit_kernel assembles its stretch of the stream in, in 32-bit words (launch_emit clamps to these)
constexpr uint32_t kEmitLdsWordsMin = 2048, kEmitLdsWordsMax = 8192;
constexpr uint32_t kEncStuffChunk = 4096;
// the headers (one upload for the batch) to the front of every image's stream
hipError_t launch_place_headers(hipStream_t stream, const DevEncImage *images, int n_images, const uint8_t *headers, uint8_t *out);
// marks: one bit per raw byte (word (raw_off >> 5) + (j >> 5), bit j & 31 for byte j of the image): a restart marker follows it
hipError_t launch_stuff(hipStream_t stream, const DevEncImage *images, const EncWork *work, int n_work, const uint64_t *raw_bits,
                        const uint8_t *raw, const uint32_t *marks, uint32_t *chunk_ff, uint8_t *out, uint64_t *out_len);

}  // namespace jpgpu
