// jpeglibrary_amd/csrc/device_batch.h -- device-resident batch of scan jobs: HBM layout, uploads, kernel launches.
#pragma once
#include <hip/hip_runtime_api.h>

#include <map>
#include <string>
#include <vector>

#include "../../include/jpgpu.h"
#include "common.h"
#include "host.h"
#include "kernels.h"

namespace jpgpu {
// Crew size when the caller did not choose one: the CPUs this process may really use (affinity mask, cgroup v2 / v1 quota),
// capped at 16; JPGPU_HOST_THREADS overrides.  granted_host_cpus() is the same figure without the cap and the override.
int default_host_threads();
int granted_host_cpus();
}  // namespace jpgpu

// Pinned staging ring of a context: the caller's (pageable) file bytes are copied into it by the host crew and leave it
// as few large DMAs on the context's upload stream (ref input contract: caller-owned, possibly multi-segment
// ReadOnlySequence<byte>, apps/JpegDecode/MemoryPoolBufferWriter.cs:166-174 -- "the shim linearises into pinned memory").
struct StagingRing {
    static constexpr int kMaxSlots = 16;
    int n_slots = 8;                  // JPGPU_STAGING_SLOTS (8 x 32 MiB: profiles/r02_ingest_sweep.jsonl)
    size_t slot_bytes = 32u << 20;    // JPGPU_STAGING_SLOT_MB; fixed once the first slot exists
    uint8_t *slot[kMaxSlots] = {};
    hipEvent_t drained[kMaxSlots] = {};  // recorded behind the DMA that reads the slot
    uint32_t *verdict = nullptr;      // pinned: per-file results of the ingest verification kernel
    size_t verdict_cap = 0;
};

struct jpgpu_ctx {
    int device = 0;
    hipStream_t stream = nullptr;         // kernels of jpgpu_batch_decode and friends
    hipStream_t stream2 = nullptr;        // second half of a large batch in jpgpu_batch_decode (DeviceBatch::decode)
    hipStream_t upload_stream = nullptr;  // H2D of jpgpu_batch_upload: runs beside another batch's decode on `stream`
    // progressive batches too large for one resident launch: the DC scans and every component's AC scans form chains that never
    // touch the same coefficients; each chain's launches go to a stream of its own (DeviceBatch::run_progressive)
    static constexpr int kProgChains = 5;
    hipStream_t prog_stream[kProgChains] = {};
    hipEvent_t prog_ev[kProgChains + 1] = {};
    StagingRing staging;
    int host_threads = 0;                 // crew size of jpgpu_batch_upload; 0 = min(CPUs granted to the process, 16)
    std::string last_error;
    int num_cus = 0;
    uint64_t device_bytes = 0;            // the device's memory: a frame whose output alone is beyond it fails by itself (status 7)
};

namespace jpgpu {

// grow-only device allocation
struct DevBuffer {
    void *ptr = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t bytes);
    void release();
};

// What the last jpgpu_batch_upload did (jpgpu_batch_ingest_stats).
struct IngestStats {
    int threads = 0;
    int n_header_only = 0;  // files whose plan came from their headers alone, confirmed by the device
    int n_full_walk = 0;    // files that took the full Identify + Decode marker walks on the host
    float parse_ms = 0, copy_ms = 0, full_walk_ms = 0, layout_ms = 0, total_ms = 0;
    int n_pinned_dma = 0;   // segments DMA'd straight from the caller's page-locked memory
    int n_linearised = 0;   // multi-segment files gathered as a whole on the host
};

struct ImagePlan {
    int status = JPGPU_OK;
    int detail = 0;
    std::string error;
    // a failure of the marker walk BEHIND a scan that was already handed to the scan decoder: the reference only meets it
    // if that scan itself decodes (Decode() runs ProcessScan before it reads the next marker), so it is reported after the
    // device-side status of the scans recorded before it
    int late_status = JPGPU_OK, late_detail = 0;
    bool replay_skip = false;  // replay of the failed progressive frames (replay_failed_progressive): this image is left as it was decoded
    std::string late_error;
    // what the walk ends in instead when scan job `swallow_job` (the file's last sequential scan) leaves exactly one whole
    // byte unread: the reference's reader then resumes one byte INTO the terminating marker (plan_swallowed_terminator)
    int swallow_job = -1, swallow_status = JPGPU_OK, swallow_detail = 0;
    std::string swallow_error;
    uint16_t width = 0, height = 0;
    uint8_t precision = 0, num_components = 0, sof = 0;
    uint32_t restart_interval = 0;
    uint32_t mcus_per_line = 0, mcus_per_column = 0, blocks_per_mcu = 0;
    uint64_t total_blocks = 0;
    uint64_t out_offset = 0, out_bytes = 0;
    uint64_t planes_offset = 0, planes_bytes = 0;  // EXTENDED_U16: the int16 planes K3 writes, in DeviceBatch::d_planes_
    uint64_t coef_offset = 0;
    jpgpu_plane_info plane[4] = {};
    std::vector<int> jobs;     // indices into DeviceBatch::jobs_
    uint64_t file_offset = 0;  // position of the file inside the device input buffer
    size_t file_len = 0;
};

class WorkCrew;

class DeviceBatch {
  public:
    const IngestStats &ingest_stats() const { return ingest_; }
    explicit DeviceBatch(jpgpu_ctx *ctx) : ctx_(ctx) {}
    ~DeviceBatch();

    // Whole files: host parse (Identify + Decode's marker loop) -> plans/jobs -> HBM.
    int upload_files(const uint8_t *const *jpeg, const size_t *len, int n, int format);
    // (the planner of a one-file batch: sequential scans, by ordinal, that leave their reader one byte into the terminating marker)
    void set_forced_swallow(const std::vector<int> &ordinals) { forced_swallow_ = ordinals; }
    // The same for files handed over as lists of segments (the ReadOnlySequence<byte> a reference caller passes to SetInput,
    // JpegDecoder.cs:56-62; multi-segment: apps/JpegDecode/MemoryPoolBufferWriter.cs:166-174): file i is the next
    // segments_per_file[i] entries of `segments`.  flags & JPGPU_UPLOAD_PINNED: every segment lies in page-locked memory
    // (jpgpu_host_alloc / jpgpu_host_register) and is DMA'd to HBM from where it lies -- no staging copy, no crew time.
    int upload_segments(const jpgpu_segment *segments, const int *segments_per_file, int n, int format, unsigned flags);
    // One progressive frame collected by the JpegDecoder mirror: its entropy scans + the Dispose() pass.
    int upload_progressive_frame(const ProgressiveFrame &frame, const uint8_t *file, size_t file_len, int sof, int format);
    // The per-scan boundary (jpgpu_progressive_*): ONE entropy scan of the frame (index into frame.scans()) accumulating into
    // the frame's coefficient store, which lives in d_coefs_ and persists from call to call (same frame, same layout, the
    // buffers only grow); first_scan clears it like JpegBlockAllocator.Allocate does.  Then run_marker_index + run_huffman.
    int upload_progressive_scan(const ProgressiveFrame &frame, int scan_index, bool first_scan);
    // The store as it is now, kept aside (a device copy): what rerun_failed_progressive_scan() starts again from.
    int snapshot_progressive_store();
    // The scan of the last upload_progressive_scan() failed on the device.  The reference's ProcessScan threw at one coefficient
    // and left the store exactly there (JpegHuffmanProgressiveScanDecoder.cs:92-419; Decode()'s finally then disposes it,
    // JpegDecoder.cs:545-549): the store is put back (the snapshot; zero behind a first scan) and the scan issued once more on
    // the kernel that walks and stores coefficient by coefficient, up to the restart interval it failed in.
    int rerun_failed_progressive_scan(bool first_scan);
    // ... and the Dispose() pass alone over that store: the frame job without entropy scans; then run_idct.
    int upload_progressive_dispose(const ProgressiveFrame &frame, int format);
    // One pre-built scan job whose entropy bytes are `entropy` (level-2 API and the JpegDecoder mirror).
    int upload_single_job(const ScanJob &job, int format, const void *initial_output, size_t initial_output_bytes);
    // Coefficient hand-off (progressive / config 5): images described by frame geometry + quantisation tables only.
    int upload_frames(const jpgpu_frame *frames, const uint16_t *qt /*[n][4][64]*/, int n, int format);

    int run_marker_index();
    int run_huffman();
    int run_subseq_sync(const uint32_t **final_state, const uint32_t **first_block);  // K2S rounds + prefix sums only
    int run_idct();
    int run_progressive();     // entropy scans of progressive frames (K2P), ordinal by ordinal
    int decode();  // marker index + the selected pipeline, with stage events
    int sync();    // waits for THIS batch's device work (an event behind its last launch), not for the whole stream

    int size() const { return (int)images_.size(); }
    const ImagePlan *image(int i) const { return (i >= 0 && i < (int)images_.size()) ? &images_[i] : nullptr; }
    int result(int i, jpgpu_image_result *res);
    void *output_device(uint64_t *total) const {
        if (total) *total = out_bytes_;
        return d_out_.ptr;
    }
    void *coefs_device(uint64_t *total_blocks) const {
        if (total_blocks) *total_blocks = total_blocks_;
        return d_coefs_.ptr;
    }
    int download_output(int i, void *dst, size_t cap);
    int download_coefficients(int i, int16_t *dst, size_t cap_blocks);
    int upload_coefficients(int i, const int16_t *src, size_t nblocks);
    int stage_ms(float ms[4]);
    void totals(uint64_t *compressed, uint64_t *blocks, uint64_t *pixels, uint64_t *out_bytes) const;
    int format() const { return format_; }
    void note_entropy_only_request() { in_decode_request_ = false; }
    int last_subseq_rounds() const { return last_subseq_rounds_; }
    int progressive_fallbacks() const { return prog_fallbacks_; }
    int subseq_fallbacks() const { return k2s_fallbacks_; }
    int marker_fallbacks() const { return k1_fallbacks_; }
    int progressive_replays() const { return prog_replays_; }
    void set_partial_flush(bool on) { partial_flush_ = on; }
    // the optimizer path only needs the files, the scan descriptors and the marker index: no coefficient / sample buffers
    void set_entropy_only(bool on) { entropy_only_ = on; }
    void set_preset_restart_intervals(std::vector<int> v) { preset_dri_ = std::move(v); }
    void set_preset_no_subseq(std::vector<uint8_t> v) { preset_no_subseq_ = std::move(v); }

  private:
    friend class OptimizeBatch;
    bool entropy_only_ = false;
    std::vector<int> preset_dri_;  // entropy-only mode: restart interval in force at each file's scan
    std::vector<uint8_t> preset_no_subseq_;  // entropy-only mode: files whose DRI = 0 scan stays on the interval kernel
    int fail(int status, const std::string &msg);
    int hip_fail(hipError_t e, const char *what);
    int layout_and_upload(const std::vector<const uint8_t *> &file_ptr, const std::vector<size_t> &file_len);
    void plan_image_geometry(ImagePlan &img, const BaselineGeometry &geo) const;
    // ingest (jpgpu_batch_upload), see device_batch.cpp
    struct FilePlan;
    void plan_file_headers(const uint8_t *file, size_t len, FilePlan &fp) const;
    void plan_file_full(const uint8_t *file, size_t len, int index, FilePlan &fp) const;
    void plan_swallowed_terminator(FilePlan &fp, const uint8_t *file, size_t len, bool identify_is_clean) const;
    struct FileSegs;
    int stage_files(WorkCrew &crew, const std::vector<FileSegs> &files, const std::vector<FilePlan> &plans, bool pinned);
    // device work of this batch issued since its last sync(): uploads are ordered behind it (an upload right
    // after an un-synced decode must not overwrite what that decode's kernels still read)
    int mark_work();
    int order_upload_behind_work();
    std::pair<const uint8_t *, size_t> arena_span_ = {nullptr, 0};  // JPGPU_UPLOAD_PINNED_ARENA: the span the device copy mirrors
    hipEvent_t done_ev_ = nullptr;
    bool work_in_flight_ = false;
    int verify_plans(const std::vector<FilePlan> &plans, const std::vector<int> &spec, std::vector<uint32_t> &first);
    bool whole_files_ = false;     // the last upload was whole files (upload_files / upload_segments): d_input_ holds them as they came
    bool files_resident_ = false;  // layout_and_upload: the files are in d_input_ already (staged by upload_files)
    IngestStats ingest_;
    DevBuffer d_verify_;
    int fetch_status();
    int clear_partial_outputs();
    bool keep_progressive_store_ = false;  // run_progressive: the store holds earlier scans' coefficients (per-scan boundary)
    bool defer_refusal_ = false;           // result(): a frame job's `refuse` is the Dispose() pass's business, not a scan's
    bool keep_canvas_ = false;  // layout of a single scan job over the caller's samples: nothing the scan does not write is touched
    struct OutClear {
        uint64_t first, second;              // (offset, bytes) in the output buffer
        uint64_t planes_first, planes_bytes; // EXTENDED_U16: the image's int16 planes
    };
    std::vector<OutClear> out_clear_;  // images whose scans leave frame components unwritten

    jpgpu_ctx *ctx_;
    int format_ = JPGPU_FMT_INTERLEAVED_U8;
    std::vector<ImagePlan> images_;
    std::vector<ScanJob> jobs_;
    std::vector<int> job_image_;
    std::vector<uint64_t> job_entropy_off_;  // offset of the job's entropy segment inside its file
    std::vector<DevScan> h_scans_;
    std::vector<DevScanStatus> h_status_;
    bool status_valid_ = false;
    std::vector<DevHuffTable> huff_pool_;
    std::vector<DevQuantTable> quant_pool_;
    int n_huff_slots_ = 1;
    // A MIDDLE sequential scan that left one whole byte unread (round 6): the reference resumes its marker walk one byte into the
    // terminating marker, so WHICH scans exist behind it depends on the decode.  The file is planned again with that knowledge and decoded
    // by a batch of its own; its output and result replace the image's (redo_swallowed; cached per image).
    int redo_swallowed(int i, int job, jpgpu_image_result *res);
    std::vector<int> forced_swallow_;
    struct Redo {
        jpgpu_image_result res;
        std::string error;
    };
    std::map<int, Redo> redo_;
    std::vector<SubseqPool> k2_pools_;          // pooled runs of K2 (round 6): entries of d_huff_work_ behind the first n_huff_work_
    uint32_t k2_ticket_base_[kK2MaxPools] = {};  // tickets the earlier launches of this upload drew from each pool's counter (d_k2_tickets_)
    uint32_t k2_tab_bytes_ = 0;  // LDS of the largest table set a sequential scan of the upload stages (K2, the K2S final pass)
    int n_huff_work_ = 0, n_idct_work_ = 0;
    int idct_class_begin_[kNumIdctLayoutClasses + 1] = {};
    std::vector<int> idct_later_begin_;  // d_idct_work_ behind the classes: [k], [k + 1]) = the k-th ordered launch (run_idct)
    uint64_t total_blocks_ = 0, out_bytes_ = 0, planes_bytes_ = 0, input_bytes_ = 0, compressed_bytes_ = 0, total_pixels_ = 0;
    DevBuffer d_planes_;  // EXTENDED_U16: K3's PLANAR_I16 output, converted by extend_u16_kernel
    DevBuffer d_extend_desc_;
    DevBuffer d_gather_;  // JPGPU_UPLOAD_PINNED: the piece list of gather_pinned_kernel
    bool in_decode_request_ = false;  // the last entropy stage was issued by decode() (fetch_status's fallback re-issues the same thing)
    uint32_t total_ends_ = 0;

    // DRI = 0 scans: self-synchronising subsequence decode (K2S)
    DevBuffer d_lut_pool_;  // fused lookups of every pool table (K2S round kernel)
    DevBuffer d_sub_work_, d_sub_scan_ids_, d_sub_exit_a_, d_sub_exit_b_, d_sub_nblk_, d_sub_first_, d_sub_entry_, d_sub_dcsum_, d_sub_dcentry_, d_sub_changed_, d_sub_same_, d_sub_perm_;
    DevBuffer d_prog_snapshot_;  // per-scan session: the frame's store in front of the current scan
    uint64_t prog_snapshot_blocks_ = 0;
    DevBuffer d_dispose_;  // DisposeJob per frame that takes the generic Dispose() pass
    std::vector<DisposeJob> dispose_jobs_;
    uint32_t dispose_max_blocks_ = 0;
    int run_dispose_passes(hipStream_t stream);
    // dispose_pass_kernel transforms the stores of the generic-Dispose frames IN PLACE: it runs once per entropy stage / store
    // clear / coefficient upload, a second jpgpu_batch_run_idct (or Dispose of a session) only flushes again
    bool dispose_done_ = false;
    int replay_failed_progressive();
    bool replay_done_ = false, replay_possible_ = false, prog_by_scan_ = false;
    bool replay_layout_active_ = false, in_replay_ = false;  // the work lists are the replay's (failed frames only): the next decode() restores the batch's own
    bool partial_flush_ = true;                               // JPGPU_NO_PARTIAL_FLUSH / jpgpu_batch_set_partial_flush: throughput callers may leave the replay out
    std::vector<std::pair<size_t, ScanJob>> replay_saved_jobs_;  // the jobs the replay rewrote, as they were
    int restore_after_replay();
    int prog_replays_ = 0;
  public:
    int clear_progressive_stores();
  private:
    // K1 in one pass (launch_marker_onepass): the look-back descriptors (tagged, never cleared between decodes), a ticket counter
    // per scan job + the device's give-up word behind them, the host's give-up word in page-locked memory
    DevBuffer d_k1_desc_, d_k1_tickets_, d_k1_order_;
    int n_k1_groups_ = 0;
    uint32_t *h_k1_giveup_ = nullptr;
    uint32_t k1_epoch_ = 0, k1_tag_ = 0;
    bool k1_onepass_ = true;
    int k1_fallbacks_ = 0;
    bool sub_same_valid_ = false;  // d_sub_same_ holds the twins of this upload's subsequences (subseq_same_kernel)
    int n_sub_work_ = 0, n_sub_gather_ = 0, n_sub_scans_ = 0, n_sub_final_work_ = 0, sub_final_spl_ = 2;
    DevBuffer d_sub_final_work_;
    std::vector<SubseqPool> sub_pools_;  // pooled runs of the final pass: entries of d_sub_final_work_ behind the first n_sub_final_work_
    uint32_t total_subs_ = 0, max_subs_per_scan_ = 0;
    std::vector<uint32_t> sub_scan_ids_;
    int last_subseq_rounds_ = 0;
    // Device-driven K2S rounds (launch_subseq_sync): the host enqueues a budget of rounds and reads the counts when it next
    // waits for the batch (sync()).  k2s_budget_: rounds the last checked decode of this upload used (0 = not known yet: the
    // first decode takes kSubseqFirstBudget); k2s_unchecked_: such a decode is in flight; k2s_idct_behind_: the output stage
    // was issued behind it (the fallback repeats it too); k2s_host_checked_: the rounds did not suffice once (or
    // JPGPU_SUBSEQ_HOST_CHECK): this upload's decodes take the host-checked loop.
    int k2s_budget_ = 0, k2s_issued_ = 0, k2s_fallbacks_ = 0;
    bool k2s_unchecked_ = false, k2s_idct_behind_ = false, k2s_host_checked_ = false;
    int check_subseq_rounds();
    // progressive frames: work of scan ordinal k is prog_work[prog_begin_[k] .. prog_begin_[k + 1])
    DevBuffer d_prog_work_, d_prog_sync_;  // d_prog_sync_: workgroups of the pipelined launch that have started
    std::vector<int> prog_begin_;
    // ... and the scans with few, long intervals (one wave per interval): prog_work[prog_stream_begin_[k] .. [k + 1])
    std::vector<int> prog_stream_begin_;
    // ... and the pipelined launch's own list: one entry per WAVE (a scan that only follows one cheap scan runs behind it in
    // that scan's wave, DevScan::wave_next): prog_work[prog_pipe_begin_ .. prog_pipe_begin_ + prog_pipe_count_)
    int prog_pipe_begin_ = 0, prog_pipe_count_ = 0;
    // ... and the same stream work once more, grouped by chain (0 = DC scans, 1 + c = the AC scans of frame component c) and by
    // the scan's ordinal inside its frame's chain: prog_work[prog_chain_begin_[x][j] .. [x][j + 1])
    std::vector<int> prog_chain_begin_[jpgpu_ctx::kProgChains];
    bool prog_chains_ok_ = false;
    bool prog_pipelined_ = false;  // all progressive scans are single streams with <= 3 direct dependencies: one launch
    uint32_t prog_spin_budget_ = 1u << 22;  // polls a follower scan may spend in the pipelined launch (JPGPU_PROG_SPIN_BUDGET)
    int prog_fallbacks_ = 0;                // times the pipelined launch timed out and the step was re-issued level by level
    std::vector<std::pair<uint64_t, uint64_t>> prog_clear_;  // (first block, blocks) of every progressive frame's store
    // RGB / RGBA output for layouts without a fused conversion: INTERLEAVED_U8 samples in a scratch image first
    struct RgbConvert {
        uint32_t image;
        uint64_t out_offset, pixels;
        int components;
    };
    std::vector<RgbConvert> rgb_convert_;
    DevBuffer d_rgb_scratch_;
    DevBuffer d_chunk_work_, d_chunk_sums_;
    int n_chunk_work_ = 0;
    DevBuffer d_unstuffed_, d_ends_u_;  // K1 output: entropy data as the bit reader sees it + interval ends in it
    DevBuffer d_input_, d_scans_, d_status_, d_ends_, d_huff_pool_, d_quant_pool_, d_huff_work_, d_idct_work_, d_coefs_, d_out_, d_k2_tickets_;
    DevBuffer d_sr_luts_, d_sr_set_scan_;  // K2S round kernel: its lookups per distinct set of tables (kSrLutSetBytes each), and a scan that stages each set
    // stage events of every decode() since the last stage_ms() query (4 events per decode)
    std::vector<hipEvent_t> ev_pool_;
    size_t ev_used_ = 0;
    std::vector<bool> ev_serial_;  // per decode(): issued serially with stage events (true) or as two overlapped halves
    // decode() in two halves (K2 of the second beside K3 of the first)
    static constexpr int kSerialEvery = 8;
    bool overlap_ok_ = false;
    int decodes_since_query_ = 0;
    int huff_split_ = 0;
    int idct_split_begin_[2][kNumIdctLayoutClasses + 1] = {};
    DevBuffer d_idct_work_split_;
};

}  // namespace jpgpu
