// jpeglibrary_amd/csrc/kernels.h -- launch wrappers of the HIP kernels (internal C++ API of libjpgpu.so)
#pragma once
#include <hip/hip_runtime_api.h>
#include <stddef.h>
#include <stdint.h>

#include "common.h"

namespace jpgpu {

constexpr int kHuffWaves = 11;                // wavefronts per Huffman workgroup at most (what the standard four tables leave room for)
// LDS of the K2 family (k2_huffman.hip, the K2S final pass): the scan's staged tables -- u32 first levels that carry values and
// pairs, round 6: an AC table 2^11 entries, a DC table 2^9, + second level, header and the reference's arrays each -- then
// kK2WaveLdsBytes per wave (64 staged blocks + 64 stream rings) and the block info.
constexpr uint32_t kK2AcTabBytes = 8192 + 512 + 16 + 320, kK2DcTabBytes = 2048 + 512 + 16 + 320, kK2WaveLdsBytes = 8192 + 64 * 68;
constexpr uint32_t kK2LdsBudget = 160 * 1024 - 64;
inline uint32_t k2_scan_tab_bytes(const DevScan &s) {  // a DC table = one that some component decodes its DC symbols with
    uint32_t bytes = 0;
    for (int sl = 0; sl < kMaxHuffSlots; sl++) {
        if (s.huff_pool[sl] == 0xFFFF) continue;
        bool is_dc = false;
        for (int c = 0; c < s.scan_components; c++) is_dc |= s.comp[c].dc_slot == sl;
        bytes += is_dc ? kK2DcTabBytes : kK2AcTabBytes;
    }
    return bytes;
}
// waves per K2 workgroup for a batch whose largest table set takes tab_bytes (eight AC tables still leave seven)
constexpr int huffman_waves(uint32_t tab_bytes) {
    return (int)((kK2LdsBudget - tab_bytes) / kK2WaveLdsBytes) < kHuffWaves ? (int)((kK2LdsBudget - tab_bytes) / kK2WaveLdsBytes) : kHuffWaves;
}
// restart intervals per Huffman workgroup (one per lane): 64 * huffman_waves(table slots of the batch)
constexpr int kIdctBlocksPerWg = 256;         // 8x8 blocks per IDCT tile (one per lane)
constexpr int kIdctTilesPerWg = 16;           // consecutive tiles walked by one IDCT workgroup (prefetch pipeline)


constexpr uint32_t kMarkerChunkBytes = 4096;  // K1 chunk size (256 lanes x 16 bytes)
// consecutive chunks of one scan handled by one K1 workgroup (one work-list entry).  4 was measured in round 2: K1 1.53 ms
// instead of 1.27-1.35 per 1024 x 4K -- the quarter of a million small workgroups are not what K1 waits for, and chunks taken
// one after the other inside a workgroup hide less latency than the same chunks in separate workgroups
constexpr uint32_t kMarkerChunksPerWg = 1;  // marker_count_kernel takes several work entries per workgroup and relies on 1 here
hipError_t launch_marker_index(hipStream_t stream, const uint8_t *data, const DevScan *scans, int n_scans, const ChunkWork *work,
                               int n_chunks, ChunkSum *sums, uint32_t *ends, DevScanStatus *status, uint8_t *udata, uint32_t *ends_u);
// K1 in one pass (k1_markers.hip): desc = kMarkerDescBytes per chunk (cleared when allocated; a group uses its first chunk's),
// tickets[0] = the ticket counter (cleared once per upload; read by -DJPGPU_K1_TICKETS builds only since round 6: a group takes its place
// in `order` from its workgroup index), epoch = decodes of this upload issued before this one, tag = a non-zero
// number no earlier launch over `desc` has used; spin_budget = polls a workgroup may spend waiting for a predecessor before it counts
// the chunks in front of its group itself -- *host_giveup (page-locked host memory) != 0 afterwards says that happened (the results
// are complete either way).
constexpr size_t kMarkerDescBytes = 64;
#ifndef JPGPU_K1_GROUP
#define JPGPU_K1_GROUP 4
#endif
constexpr uint32_t kMarkerGroupChunks = JPGPU_K1_GROUP;  // chunks a workgroup of the one-pass index takes; `order`: (scan, first chunk) per group, by (group, scan)
hipError_t launch_marker_onepass(hipStream_t stream, const uint8_t *data, const DevScan *scans, const ChunkWork *order, int n_groups,
                                 void *desc, uint32_t *tickets, uint32_t epoch, uint32_t tag, uint32_t spin_budget,
                                 uint32_t *host_giveup, uint32_t *ends, DevScanStatus *status, uint8_t *udata, uint32_t *ends_u);
hipError_t launch_huffman(hipStream_t stream, const uint8_t *data, const DevScan *scans, const HuffWork *work, int n_work,
                          const uint32_t *ends, DevScanStatus *status, const DevHuffTable *huff_pool, int16_t *coefs,
                          int n_slots, const uint8_t *lut_pool, uint32_t tab_bytes);
// One pooled run of K2 (huffman_pool_kernel): `work` = its n_chunks entries (scan, first interval) of 64 intervals each, all staging the
// same tables; `groups` workgroups of huffman_waves(tab_bytes) waves each draw from *counter, which is never cleared: the launch takes
// n_chunks + groups * waves tickets, the first of them ticket_base (the sum of the earlier launches' draws).
constexpr int kK2MaxPools = 8;
hipError_t launch_huffman_pool(hipStream_t stream, const uint8_t *data, const DevScan *scans, const HuffWork *work, int n_chunks, uint32_t *counter,
                               uint32_t ticket_base, int groups, const uint32_t *ends, DevScanStatus *status, int16_t *coefs, int n_slots,
                               const uint8_t *lut_pool, uint32_t tab_bytes);
// lut_pool: kLutPoolBytesPerTable per pool table, filled by launch_lut_pool (K2 and the K2S round kernel copy from it)
hipError_t launch_lut_pool(hipStream_t stream, const DevHuffTable *huff_pool, int n_tables, uint8_t *lut_pool);
constexpr int kNumIdctLayoutClasses = 6;
constexpr int kIdctClassStoreHoldsSamples = 5;  // frames whose generic Dispose() pass has run (any format): flush_output_kernel
// Output layout class of a scan for INTERLEAVED_U8 (0 = generic bytewise path, else a specialised kernel).
int idct_layout_class(const DevScan &s);
// work is sorted by layout class; class_begin[c]..class_begin[c+1] are the workgroups of class c.
// RGB / RGBA formats: classes with a fused conversion write `out`; the generic class writes INTERLEAVED_U8 samples into
// `generic_out` (same offsets), to be converted by launch_ycc_to_rgb.
hipError_t launch_idct(hipStream_t stream, const int16_t *coefs, const DevScan *scans, const IdctWork *work,
                       const int class_begin[kNumIdctLayoutClasses + 1], const DevScanStatus *status,
                       const DevQuantTable *quant_pool, uint8_t *out, int format, const YccRgbFactors &kf, uint8_t *generic_out);
// the reference's Dispose() taken literally (frames whose component slots do not map one to one onto their components)
hipError_t launch_dispose_pass(hipStream_t stream, int16_t *coefs, const DisposeJob *jobs, int n_jobs, uint32_t max_blocks, const DevQuantTable *quant_pool);
hipError_t launch_ycc_to_rgb(hipStream_t stream, const uint8_t *src, uint8_t *dst, uint64_t n_pixels, int comps, int bpp, const YccRgbFactors &kf);

// Waves per workgroup of the POOLED form of the K2S final pass (k2s_subseq.hip: runs of scans that stage the same tables; one workgroup per CU, every wave takes the
// next 64 lanes from a counter): 10 waves + 4 tables fill a CU's LDS.  Runs shorter than kSubFinalPoolMinChunks waves, and runs
// beyond the kSubFinalMaxPools-th, take the plain form.
constexpr int kSubFinalPoolWaves = 11, kSubFinalPoolMinChunks = 40, kSubFinalMaxPools = 8;
constexpr int kSubseqCtlPoolCounter = 72;  // changed_dev words [72, 72 + kSubFinalMaxPools): the pools' counters (cleared by every launch)
struct SubseqPool {
    int first, count;  // entries of the pooled work list (one per wave of 64 lanes)
};
// DRI = 0 scans (K2S): self-synchronising subsequence decode into the coefficient buffer (device_rounds: see launch_subseq_sync).
constexpr uint32_t kSubseqBits = 1024;  // smallest subsequence (DevScan::sub_shift = 10); large batches use 2048 / 4096 bits
hipError_t launch_subseq_decode(hipStream_t stream, const uint8_t *udata, const DevScan *scans, const HuffWork *work, int n_work,
                                const uint32_t *scan_ids, int n_scans, const uint32_t *ends_u, DevScanStatus *status,
                                const DevHuffTable *huff_pool, uint32_t *exit_a, uint32_t *exit_b, uint32_t *nblk, uint32_t *first_block,
                                uint32_t *entry_used, void *dcsum, void *dc_entry, uint32_t *changed_dev, int16_t *coefs, int n_slots,
                                int max_rounds, int *rounds_used, const uint8_t *lut_pool, const HuffWork *final_work, int n_final_work,
                                uint32_t *same_dist, bool *same_valid, int device_rounds, const HuffWork *gather_work, int n_gather,
                                const HuffWork *pool_work, const SubseqPool *pools, int n_pools, int num_cus, uint32_t *lane_perm, int subs_per_lane,
                                uint32_t tab_bytes, const uint8_t *sr_luts);
#ifndef JPGPU_SR_LB
#define JPGPU_SR_LB 10
#endif
constexpr int kSrLutBits = JPGPU_SR_LB;  // lookup width of the K2S round kernel (a 10-bit prefix decides codes of up to 10 bits)
constexpr size_t kSrLutSetBytes = (size_t)kMaxHuffSlots << (kSrLutBits + 2);
// sr_luts: the round kernel's lookups, kMaxHuffSlots << (kSrLutBits + 2) bytes per distinct set of tables among the DRI = 0 scans
// (DevScan::sr_set), built once per upload behind launch_lut_pool; set_scan[k] = a scan that stages set k
hipError_t launch_sr_luts(hipStream_t stream, const DevScan *scans, const uint32_t *set_scan, int n_sets, const uint8_t *lut_pool, uint8_t *sr_luts);
// subs_per_lane: subsequences a lane of the final pass takes (the work lists are built for it): kSubFinalSubsPerLane for batches that
// fill the machine, 1 below kSubFinalFewSubs subsequences in the batch (a lone 67-Mpixel canvas: twice the waves, half as long each)
constexpr uint64_t kSubFinalFewSubs = 1u << 20;
// lane_perm: one uint32 per subsequence -- the final pass's lanes of every scan ordered by the MCUs they own (subseq_order_kernel)
// changed_dev: kSubseqCtlWords uint32 (exits changed per round, the device-driven rounds' state; word kSubseqCtlSameDone is
// cleared once per upload, the rest by every launch).  device_rounds > 0: that many rounds enqueued, nothing read back, the
// caller checks changed_dev when it next waits for the stream (k2s_subseq.hip); 0: the host checks between rounds.
constexpr uint32_t kSubseqGatherSpan = 1024;
constexpr int kSubseqCtlWords = 128, kSubseqCtlSameDone = 96, kSubseqFirstBudget = 16;
constexpr int kSubseqMaxDeviceRounds = 61;  // rounds launch_subseq_decode enqueues at most without the host looking (control words [0, 62))
// waves (of 64 subsequences) per workgroup of the K2S final pass.  4 = two workgroups per CU; one workgroup of 10 waves (K2's
// shape, 25 % more waves per CU) was measured slower: 21.1 vs 19.6 ms K2S per 1024 x 4K -- a workgroup waits for its slowest wave
#ifndef JPGPU_SF_WAVES
#define JPGPU_SF_WAVES 4
#endif
constexpr int subseq_final_waves() { return JPGPU_SF_WAVES; }
// the pooled form (eleven waves per workgroup since round 6) fits while the staged tables leave room for them
constexpr uint32_t kSfWaveLdsBytes = kK2WaveLdsBytes;
constexpr bool subseq_pool_fits(uint32_t tab_bytes) { return tab_bytes + (uint32_t)kSubFinalPoolWaves * kSfWaveLdsBytes <= kK2LdsBudget; }
// subsequences per lane of the K2S final pass (1: 8.1 ms per 1024 x 4K; 2: see DESIGN.md)
#ifndef JPGPU_SF_SUBS
#define JPGPU_SF_SUBS 2
#endif
constexpr int kSubFinalSubsPerLane = JPGPU_SF_SUBS;
// per pool table (lut_pool_kernel): the u16 images as an AC and as a DC table (2^11 + 256 u16 entries, header, the reference's arrays:
// what the K2S round kernel derives its lookups from), then the u32 images of round 6 (2^11 / 2^9 entries that carry values and pairs: K2, K2S final pass)
constexpr size_t kLutPoolBytesPerTable = 2 * (4096 + 512 + 16 + 320) + (8192 + 512 + 16 + 320) + (2048 + 512 + 16 + 320);

// progressive frames (K2P): the scans of one ordinal (position inside their frame) of every progressive frame in the batch
hipError_t launch_progressive(hipStream_t stream, const uint8_t *udata, const DevScan *scans, const HuffWork *work, int n_work,
                              const uint32_t *ends_u, DevScanStatus *status, const DevHuffTable *huff_pool, int16_t *coefs, int n_slots);
hipError_t launch_progressive_streams(hipStream_t stream, const uint8_t *udata, const DevScan *scans, const HuffWork *work, int n_work,
                                      const uint32_t *ends_u, DevScanStatus *status, const DevHuffTable *huff_pool, int16_t *coefs,
                                      int n_slots, int pipelined, uint32_t spin_budget, uint32_t *started);
size_t progressive_stream_lds_bytes(int n_slots);  // LDS of one stream workgroup (residency estimate of the pipelined launch)
int progressive_stream_blocks_per_cu(int n_slots);  // stream workgroups a CU holds at once by the runtime's occupancy query (0 = unknown)

// "O3": PLANAR_I16 planes -> the test writer's uint16 x 4 form (extend_u16_kernel)
hipError_t launch_extend_u16(hipStream_t stream, const uint8_t *planes, uint8_t *out_base, const ExtendPlanes *images, int n_images,
                             uint32_t max_pixels);

// K0 (ingest verification): offset of the first non-RST marker in each segment {offset lo, length} (+ offset hi), 0xFFFFFFFF = none
// JPGPU_UPLOAD_PINNED, segments scattered in page-locked host memory: the device reads them over the host link itself
hipError_t launch_gather_pinned(hipStream_t stream, const GatherPiece *pieces, int n_pieces, uint8_t *dst);
hipError_t launch_first_marker(hipStream_t stream, const uint8_t *data, const void *segs, const uint32_t *seg_hi, int n_segs,
                               uint32_t max_len, uint32_t *first);

// KT: symbol-level Huffman transcode of baseline scans (JpegOptimizer): mode 0 count, 1 measure, 2 emit (kt_transcode.hip)
struct EncHuffTable;
hipError_t launch_transcode(hipStream_t stream, int mode, const uint8_t *udata, const uint8_t *input, const DevScan *scans,
                            const HuffWork *work, int n_work, const uint32_t *ends_u, const uint32_t *ends_raw, DevScanStatus *status,
                            const DevHuffTable *huff_pool, uint32_t *hist, const EncHuffTable *enc, uint32_t *sizes,
                            const uint64_t *offsets, uint8_t *out, int n_slots);
hipError_t launch_transcode_offsets(hipStream_t stream, const DevScan *scans, const uint32_t *scan_ids, int n_scans, const uint32_t *sizes,
                                    const uint64_t *base, uint64_t *offsets, uint64_t *totals);
// K2S synchronisation alone + KTS: transcode of DRI = 0 scans by subsequence (kt_transcode.hip)
hipError_t launch_subseq_sync(hipStream_t stream, const uint8_t *udata, const DevScan *scans, const HuffWork *work, int n_work,
                              const uint32_t *scan_ids, int n_scans, const uint32_t *ends_u, DevScanStatus *status,
                              const DevHuffTable *huff_pool, uint32_t *exit_a, uint32_t *exit_b, uint32_t *nblk, uint32_t *first_block,
                              uint32_t *entry_used, void *dcsum, void *dc_entry, uint32_t *changed_dev, int n_slots, int max_rounds,
                              int *rounds_used, const uint8_t *lut_pool, const uint32_t **final_state_out, uint32_t *same_dist, bool *same_valid,
                              int device_rounds, const HuffWork *gather_work, int n_gather, const uint8_t *sr_luts);
// gather_work: (scan, first subsequence) per kSubseqGatherSpan subsequences -- the work list of the rounds behind round 1
// (same_dist: one uint32 per subsequence, *same_valid: "filled for this upload" -- the flat-region twins, k2s_subseq.hip)
hipError_t launch_subseq_transcode(hipStream_t stream, int mode, const uint8_t *udata, const DevScan *scans, const HuffWork *work, int n_work,
                                   const uint32_t *ends_u, DevScanStatus *status, const DevHuffTable *huff_pool, const uint32_t *exit_state,
                                   const uint32_t *first_block, uint32_t *hist, const EncHuffTable *enc, uint32_t *sub_bits,
                                   const uint64_t *sub_bitoff, const uint64_t *scan_raw_off, uint8_t *raw, int n_slots);
hipError_t launch_subseq_bit_offsets(hipStream_t stream, const DevScan *scans, const uint32_t *scan_ids, int n_scans, const uint32_t *sub_bits,
                                     uint64_t *bitoff, uint64_t *totals);
}  // namespace jpgpu
