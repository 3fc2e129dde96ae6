// jpeglibrary_amd/csrc/common.h -- structures shared by the host library and the HIP kernels.
//
// Vocabulary follows the reference: frame, scan, MCU, restart interval, block (8x8), zig-zag order.
// A "scan job" is one SOS of one image: the unit JpegScanDecoder.ProcessScan works on
// (ref: ScanDecoder/JpegHuffmanBaselineScanDecoder.cs:51-177).
#pragma once
#include <stdint.h>

namespace jpgpu {

constexpr int kMaxScanComponents = 4;
constexpr int kMaxBlocksPerMcu = 16;  // T.81 allows 10; the reference does not check, we cap at 16
constexpr int kHuffLutBits = 9;       // first-level lookup width (reference: 8-bit lookahead + maxcode search)
constexpr int kHuffLutSize = 1 << kHuffLutBits;
constexpr int kMaxHuffSlots = 8;      // tables staged in LDS per scan job: <=4 DC + <=4 AC

// Device image of JpegHuffmanDecodingTable (ref: JpegHuffmanDecodingTable.cs:33-49, 339-376).
// lut[i] for the next kHuffLutBits bits: (code_size << 8) | symbol, 0 = take the maxcode search.
// maxcode / valoffset / values are the reference's arrays verbatim.
struct alignas(16) DevHuffTable {
    uint16_t lut[kHuffLutSize];
    uint16_t maxcode[18];
    uint8_t valoffset[20];
    uint8_t values[256];
    uint8_t pad[8];
};
static_assert(sizeof(DevHuffTable) % 16 == 0, "DevHuffTable must be a multiple of 16 bytes");

// Quantization table, zig-zag order like JpegQuantizationTable.Elements (ref: JpegQuantizationTable.cs:47).
struct alignas(16) DevQuantTable {
    uint16_t q[64];
};

struct DevScanComponent {
    uint8_t component_index;  // index into the frame's component list (WriteBlock's componentIndex)
    uint8_t h, v;             // sampling factors
    uint8_t hs, vs;           // subsampling = max / own (ref: ScanDecoder/JpegHuffmanScanDecoder.cs:66-67)
    uint8_t quant_slot;       // index into DevScan::quant_pool
    uint8_t dc_slot, ac_slot; // index into DevScan::huff_pool (LDS slot)
};

enum OutputFormat : int32_t { kFmtInterleavedU8 = 0, kFmtPlanarU8 = 1, kFmtPlanarI16 = 2, kFmtRgbU8 = 3, kFmtRgbaU8 = 4, kFmtExtendedU16 = 5 };
constexpr int kNumOutputFormats = 6;
constexpr bool fmt_is_interleaved(int f) { return f == kFmtInterleavedU8 || f == kFmtRgbU8 || f == kFmtRgbaU8; }
constexpr int fmt_bytes_per_pixel_rgb(int f) { return f == kFmtRgbaU8 ? 4 : 3; }

// Fixed-point factors of JpegYCbCrToRgbConverter.Init (ref: apps/JpegDecode/JpegYCbCrToRgbConverter.cs:66-118):
// R = Y + (cr_r * Cr + half >> 16), B = Y + (cb_b * Cb + half >> 16), G = Y + (cb_g * Cb + half + cr_g * Cr >> 16)
// with Cb, Cr = sample - 128 and the results clamped to [0, 255] (the reference's clamp table).
struct YccRgbFactors {
    int32_t cr_r, cr_g, cb_b, cb_g;
};

// One scan job.
struct alignas(16) DevScan {
    uint64_t data_off;   // entropy segment start, byte offset into the batch input buffer
    uint64_t coef_off;   // first block of this scan in the coefficient buffer (block units, 64 x int16 each)
    uint64_t out_off;    // image base in the output buffer (bytes)
    uint64_t plane_off[kMaxScanComponents];  // planar formats: byte offset of each scan component's plane from out_off
    uint32_t plane_pitch[kMaxScanComponents];  // planar formats: pitch in samples
    uint32_t data_len;   // bytes from data_off to the end of the file
    uint32_t dri;        // restart interval in MCUs as latched by the scan decoder (0 = none)
    uint32_t total_mcus, mcus_per_line, mcus_per_column;
    uint32_t n_intervals;  // dri ? ceil(total_mcus / dri) : 1
    uint32_t ends_off;     // index of this scan's first slot in the interval-end array (n_intervals slots)
    uint32_t image_index;
    uint32_t level_shift;  // 1 << (precision - 1)
    uint16_t width, height;  // frame SamplesPerLine / NumberOfLines
    uint8_t precision, frame_components, scan_components, max_h, max_v, blocks_per_mcu;
    uint8_t restart_check_at_end;  // dri > 0 && total_mcus % dri == 0: the reference runs the restart check after the last MCU
    uint8_t shadow_mask;  // bit c (c < 4): a later scan component resolves to the same frame component; c's blocks never reach
                          // the output.  kKeepUnreachedMcus: MCUs behind an early EOI are left alone instead of zeroed
    DevScanComponent comp[kMaxScanComponents];
    uint8_t blk_comp[kMaxBlocksPerMcu];  // block-in-MCU -> scan component slot
    uint8_t blk_x[kMaxBlocksPerMcu];     // block-in-MCU -> x, y inside the component's MCU footprint
    uint8_t blk_y[kMaxBlocksPerMcu];
    uint16_t huff_pool[kMaxHuffSlots];   // pool indices of the tables this scan stages (0xFFFF = unused)
    uint16_t quant_pool[kMaxScanComponents];
    uint64_t reserved0;  // bit 0 (kScanStoreHoldsSamples): the frame's store holds samples by the time K3 runs (generic Dispose() pass)
    uint32_t chunk_off;  // first entry of this scan in the chunk-summary array (K1)
    uint32_t n_chunks;   // 4 KiB chunks covering the entropy segment (from its 16-byte aligned base)
    uint32_t sub_off;    // DRI = 0 scans: first slot of this scan in the subsequence state arrays (K2S)
    uint32_t n_subs;     // DRI = 0 scans: number of 1024-bit subsequences covering the segment (0 = interval decoder)
    // ---- progressive entropy scans (kind == kScanProgressive; ref: ScanDecoder/JpegHuffmanProgressiveScanDecoder.cs).
    // total_mcus counts the scan's restart UNITS: MCUs for an interleaved scan, blocks of the component otherwise
    // (:140-194 calls HandleRestart per block); coef_off / mcus_per_line are the FRAME's: every scan of the frame
    // accumulates into one coefficient store laid out in MCU scan order (what the IDCT pass reads).
    uint8_t kind;              // ScanKind
    uint8_t ss, se, ah, al;    // spectral selection / successive approximation of the scan header
    uint8_t frame_bpm;         // blocks per MCU of the frame (all components)
    uint8_t fblk_base[kMaxScanComponents];  // first block of the scan component inside the frame's MCU
    uint8_t publishes;         // pipelined launch: other scans follow this one's progress (DevScanStatus::progress)
    uint8_t sub_shift;         // DRI = 0 scans: log2 of the subsequence length in bits (10..12), see K2S
    uint8_t wave_next;         // pipelined launch: the wave that has finished this scan goes on with scan job (this + wave_next); 0 = none
    uint8_t debug_delay_ms;    // tests only (JPGPU_DEBUG_DELAY_SCAN): the stream kernel idles this long before the scan and after each progress word
    uint8_t pad1[2];
    uint16_t hblocks[kMaxScanComponents];   // the component's own block grid (ref: JpegBlockAllocator.cs:35-84);
    uint16_t vblocks[kMaxScanComponents];   // blocks outside it go to the allocator's dummy block, i.e. nowhere
    uint32_t units_per_line;   // non-interleaved scan: blocks per line walked by the scan (:146-147)
    uint32_t dep[3];           // pipelined launch: scan jobs this scan follows (kNoDep = none), see progressive_stream_kernel
    uint32_t last_interval;    // progressive_scan_kernel: restart intervals behind this one are not decoded (0xFFFFFFFF: all; the
                               // replay of a failed file stops where the reference threw)
    uint32_t first_scan;       // sequential scans: the image's first scan job (K3: a scan behind a failed one was never started)
    uint32_t sr_set;           // DRI = 0 scans: index of the scan's set of round-kernel lookups (built once per upload: sr_lut_build_kernel)
    uint32_t pad2;
};
constexpr uint32_t kNoDep = 0xFFFFFFFFu;
// Sequential scans: DevScanStatus::pad[1] = kFailBlockBase - (index, in scan order, of the block the reference threw in), 0 = the
// scan did not fail.  Written with atomicMax by the Huffman kernels (the LOWEST failing block wins), read by K3: the reference
// has called WriteBlock for every block in front of it and for none behind it (JpegHuffmanBaselineScanDecoder.cs:99-134, 153).
constexpr uint32_t kFailBlockBase = 0xFFFFFFFFu;
constexpr uint32_t kIdctPartialMcu = 0xFFFFFFFFu;  // IdctWork::first_mcu: "the MCU the scan failed in" (the caller's canvas under a fast layout)
constexpr uint64_t kScanStoreHoldsSamples = 1;
// The reference's Dispose() taken literally for one progressive frame (dispose_pass_kernel): component c of the frame is
// transformed n[c] times in place, with the quantisation tables of the decoder's component slots that point at it, in slot order.
struct alignas(16) DisposeJob {
    uint64_t coef_off;   // the frame's first block
    uint32_t n_blocks;   // total MCUs * blocks per MCU
    uint32_t bpm, level_shift;
    uint8_t blk_comp[kMaxBlocksPerMcu];       // block in MCU -> frame component
    uint8_t n[kMaxScanComponents];            // transforms per component
    uint16_t quant[kMaxScanComponents][kMaxScanComponents];  // pool indices of their tables, in slot order
    uint32_t pad[3];
};
constexpr uint8_t kKeepUnreachedMcus = 0x80;
enum ScanKind : uint8_t { kScanSequential = 0, kScanFrameOnly = 1, kScanProgressive = 2 };
static_assert(sizeof(DevScan) % 16 == 0, "DevScan must be a multiple of 16 bytes");

// Device-side result of one scan job.
struct alignas(16) DevScanStatus {
    uint32_t n_ends;        // interval ends found by the marker index (<= n_intervals)
    uint32_t terminator;    // marker byte closing the last indexed interval (RSTn, other), 0 = data ran out
    uint32_t first_error;   // (interval << 8) | detail of the lowest failing interval, 0xFFFFFFFF = none
    uint32_t decoded_mcus;  // MCUs decoded (limits the IDCT pass when EOI came early)
    uint32_t end_pos;       // byte offset (from data_off) of the terminating marker / end of data
    uint32_t pad[3];        // [0] unstuffed length; [1] progressive scans: restart units completed (0xFFFFFFFF = finished); sequential
                            // scans: kFailBlockBase - the failing block, 0 = none;
                            // [2] optimizer walk: bits of the stream left unread behind the scan's last block
};

// Work lists: one entry per workgroup.
struct HuffWork {
    uint32_t scan;            // scan job index
    uint32_t first_interval;  // first restart interval handled by this workgroup
};
struct ChunkWork {
    uint32_t scan;
    uint32_t chunk;
};
// one piece of a page-locked input segment for gather_pinned_kernel: `len` bytes at host address `src` -> input buffer + dst_off
struct GatherPiece {
    uint64_t src;
    uint64_t dst_off;
    uint32_t len;
    uint32_t pad;
};
struct ChunkSum {
    uint32_t rst_cnt;     // RSTn markers whose FF lies in the chunk
    uint32_t keep_cnt;    // bytes the chunk contributes to the unstuffed copy (markers count 2)
    uint32_t first_term;  // raw offset of the first non-RST marker in the chunk, 0xFFFFFFFF = none
    uint32_t pad;
};
struct IdctWork {
    uint32_t scan;
    uint32_t first_mcu;
    uint32_t n_mcus;  // consecutive MCUs handled by this workgroup (a run of tiles)
    uint32_t mcus_per_tile;  // MCUs one pass of the workgroup transforms (<= 256 / blocks_per_mcu, see idct_mcus_per_tile)
};

// EXTENDED_U16 ("O3"): one image of the batch for extend_u16_kernel
struct ExtendPlanes {
    uint64_t plane_off[4];  // byte offsets of the int16 planes in the planes buffer
    uint64_t out_off;       // byte offset of the image's uint16 x 4 output
    uint32_t pitch[4];      // samples
    uint32_t hshift[4], vshift[4];
    uint32_t width, height, ncomp, precision;
    uint32_t hcnt[4], vcnt[4];  // the components' sampling factors H, V (blocks per MCU in x / y) ...
    uint32_t max_h, max_v;      // ... and the frame's maxima: the (offsetX + x) * 8 placement of a factor that is neither (round 6)
};

constexpr uint32_t kNoError = 0xFFFFFFFFu;

// detail codes (mirror jpgpu_detail in include/jpgpu.h)
enum Detail : uint32_t {
    kDetailNone = 0,
    kDetailInvalidHuffmanCode = 1,
    kDetailMarkerInData = 2,
    kDetailStreamEnded = 3,
    kDetailExpectRestart = 4,
    kDetailMissingTable = 5,
    kDetailUnsupportedFrame = 6,
    kDetailBadHeader = 7,
    kDetailEarlyEoi = 8,
    kDetailUnexpectedEnd = 9,
    kDetailNullTable = 10,  // optimizer walk: a block needs a Huffman table that was never defined (the reference's null reference)
    kDetailSpinTimeout = 11  // internal, never reported: a progressive scan of the pipelined launch gave up waiting for its producers
};

}  // namespace jpgpu
