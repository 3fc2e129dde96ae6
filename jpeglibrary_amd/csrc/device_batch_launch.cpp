// jpeglibrary_amd/csrc/device_batch_launch.cpp -- DeviceBatch: the kernel launches of a decode (K1, the K2 family, K3), their issue order,
// the wait, the device-driven K2S rounds' verdict and the stage times.
#include "device_batch.h"

#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <thread>
#include <vector>

#include "kernels.h"

namespace jpgpu {

int DeviceBatch::run_marker_index() {
    status_valid_ = false;
    if (k1_onepass_) {
        // one pass with a decoupled look-back (k1_markers.hip: groups of four chunks, handed out scan-interleaved, classified once);
        // a group that runs out of patience counts its predecessors itself and says so in *h_k1_giveup_ (a count for the tests)
        const char *bev = getenv("JPGPU_K1_SPIN_BUDGET");  // (read per call: the tests force the give-up with 0)
        const uint32_t budget = bev ? (uint32_t)strtoul(bev, nullptr, 10) : (1u << 20);
        uint32_t *tickets = (uint32_t *)d_k1_tickets_.ptr;
        hipError_t e1 = launch_marker_onepass(ctx_->stream, (const uint8_t *)d_input_.ptr, (const DevScan *)d_scans_.ptr, (const ChunkWork *)d_k1_order_.ptr,
                                              n_k1_groups_, d_k1_desc_.ptr, tickets, k1_epoch_, ++k1_tag_ ? k1_tag_ : ++k1_tag_, budget,
                                              h_k1_giveup_, (uint32_t *)d_ends_.ptr, (DevScanStatus *)d_status_.ptr,
                                              (uint8_t *)d_unstuffed_.ptr, (uint32_t *)d_ends_u_.ptr);
        // (the device's ticket counter has advanced by one launch's worth only if the launch happened: ADVICE r5)
        if (e1 != hipSuccess) return hip_fail(e1, "marker_onepass_kernel");
        k1_epoch_++;
        return mark_work();
    }
    hipError_t e = launch_marker_index(ctx_->stream, (const uint8_t *)d_input_.ptr, (const DevScan *)d_scans_.ptr, (int)h_scans_.size(),
                                       (const ChunkWork *)d_chunk_work_.ptr, n_chunk_work_, (ChunkSum *)d_chunk_sums_.ptr,
                                       (uint32_t *)d_ends_.ptr, (DevScanStatus *)d_status_.ptr, (uint8_t *)d_unstuffed_.ptr,
                                       (uint32_t *)d_ends_u_.ptr);
    return e == hipSuccess ? mark_work() : hip_fail(e, "marker_index_kernel");
}
int DeviceBatch::run_huffman() {
    status_valid_ = false;
    dispose_done_ = false;  // (the stores hold coefficients again)
    hipError_t e = launch_huffman(ctx_->stream, (const uint8_t *)d_unstuffed_.ptr, (const DevScan *)d_scans_.ptr, (const HuffWork *)d_huff_work_.ptr,
                                  n_huff_work_, (const uint32_t *)d_ends_u_.ptr, (DevScanStatus *)d_status_.ptr,
                                  (const DevHuffTable *)d_huff_pool_.ptr, (int16_t *)d_coefs_.ptr, n_huff_slots_, (const uint8_t *)d_lut_pool_.ptr, k2_tab_bytes_);
    if (e != hipSuccess) return hip_fail(e, "huffman_decode_kernel");
    for (size_t p = 0; p < k2_pools_.size(); p++) {
        // pooled runs: one workgroup per CU at most, every wave takes 64 intervals at a time from the run's counter
        const int waves = huffman_waves(k2_tab_bytes_);
        const int groups = std::min(ctx_->num_cus > 0 ? ctx_->num_cus : 256, (k2_pools_[p].count + waves - 1) / waves);
        e = launch_huffman_pool(ctx_->stream, (const uint8_t *)d_unstuffed_.ptr, (const DevScan *)d_scans_.ptr,
                                (const HuffWork *)d_huff_work_.ptr + n_huff_work_ + k2_pools_[p].first, k2_pools_[p].count, (uint32_t *)d_k2_tickets_.ptr + p,
                                k2_ticket_base_[p], groups, (const uint32_t *)d_ends_u_.ptr, (DevScanStatus *)d_status_.ptr, (int16_t *)d_coefs_.ptr,
                                n_huff_slots_, (const uint8_t *)d_lut_pool_.ptr, k2_tab_bytes_);
        if (e != hipSuccess) return hip_fail(e, "huffman_pool_kernel");
        k2_ticket_base_[p] += (uint32_t)k2_pools_[p].count + (uint32_t)(groups * waves);
    }
    if (n_sub_work_ > 0) {
        // DRI = 0 scans: the final pass writes every block of the scan as whole lines (nothing to clear).  The rounds are
        // enqueued ahead, as many as the last decode of this upload used (16 the first time); sync() reads whether they sufficed.
        const int max_rounds = (int)max_subs_per_scan_ + 2;
        const int first_budget = getenv("JPGPU_SUBSEQ_BUDGET") ? std::max(2, atoi(getenv("JPGPU_SUBSEQ_BUDGET"))) : kSubseqFirstBudget;  // (tests: force the fallback)
        const int device_rounds = k2s_host_checked_ ? 0 : std::min(max_rounds, k2s_budget_ > 0 ? k2s_budget_ : first_budget);
        k2s_issued_ = std::min(device_rounds, kSubseqMaxDeviceRounds);  // (what launch_subseq_decode really enqueues: ADVICE r5)
        k2s_unchecked_ = device_rounds > 0;
        k2s_idct_behind_ = false;
        e = launch_subseq_decode(ctx_->stream, (const uint8_t *)d_unstuffed_.ptr, (const DevScan *)d_scans_.ptr, (const HuffWork *)d_sub_work_.ptr,
                                 n_sub_work_, (const uint32_t *)d_sub_scan_ids_.ptr, n_sub_scans_, (const uint32_t *)d_ends_u_.ptr,
                                 (DevScanStatus *)d_status_.ptr, (const DevHuffTable *)d_huff_pool_.ptr, (uint32_t *)d_sub_exit_a_.ptr,
                                 (uint32_t *)d_sub_exit_b_.ptr, (uint32_t *)d_sub_nblk_.ptr, (uint32_t *)d_sub_first_.ptr,
                                 (uint32_t *)d_sub_entry_.ptr, d_sub_dcsum_.ptr, d_sub_dcentry_.ptr, (uint32_t *)d_sub_changed_.ptr, (int16_t *)d_coefs_.ptr, n_huff_slots_, max_rounds,
                                 &last_subseq_rounds_, (const uint8_t *)d_lut_pool_.ptr, (const HuffWork *)d_sub_final_work_.ptr, n_sub_final_work_,
                                 (uint32_t *)d_sub_same_.ptr, &sub_same_valid_, device_rounds, (const HuffWork *)d_sub_work_.ptr + n_sub_work_, n_sub_gather_,
                                 (const HuffWork *)d_sub_final_work_.ptr + n_sub_final_work_, sub_pools_.data(), (int)sub_pools_.size(), ctx_->num_cus,
                                 (uint32_t *)d_sub_perm_.ptr, sub_final_spl_, k2_tab_bytes_, (const uint8_t *)d_sr_luts_.ptr);
        if (e != hipSuccess) return hip_fail(e, "subsequence decode");
    }
    const int rc = run_progressive();
    return rc != JPGPU_OK ? rc : mark_work();
}
// The synchronisation of the DRI = 0 scans alone (optimizer path): converged exit states + first block of every subsequence.
int DeviceBatch::run_subseq_sync(const uint32_t **final_state, const uint32_t **first_block) {
    *final_state = (const uint32_t *)d_sub_exit_a_.ptr;
    *first_block = (const uint32_t *)d_sub_first_.ptr;
    if (n_sub_work_ <= 0) return JPGPU_OK;
    status_valid_ = false;
    hipError_t e = launch_subseq_sync(ctx_->stream, (const uint8_t *)d_unstuffed_.ptr, (const DevScan *)d_scans_.ptr, (const HuffWork *)d_sub_work_.ptr,
                                      n_sub_work_, (const uint32_t *)d_sub_scan_ids_.ptr, n_sub_scans_, (const uint32_t *)d_ends_u_.ptr,
                                      (DevScanStatus *)d_status_.ptr, (const DevHuffTable *)d_huff_pool_.ptr, (uint32_t *)d_sub_exit_a_.ptr,
                                      (uint32_t *)d_sub_exit_b_.ptr, (uint32_t *)d_sub_nblk_.ptr, (uint32_t *)d_sub_first_.ptr,
                                      (uint32_t *)d_sub_entry_.ptr, d_sub_dcsum_.ptr, d_sub_dcentry_.ptr, (uint32_t *)d_sub_changed_.ptr,
                                      n_huff_slots_, (int)max_subs_per_scan_ + 2, &last_subseq_rounds_, (const uint8_t *)d_lut_pool_.ptr,
                                      final_state, (uint32_t *)d_sub_same_.ptr, &sub_same_valid_, 0 /* host-checked: the optimizer waits for the host's table build anyway */,
                                      (const HuffWork *)d_sub_work_.ptr + n_sub_work_, n_sub_gather_, (const uint8_t *)d_sr_luts_.ptr);
    if (e != hipSuccess) return hip_fail(e, "subsequence synchronisation");
    return mark_work();
}
int DeviceBatch::run_progressive() {
    if (prog_begin_.size() <= 1 && prog_clear_.empty()) return JPGPU_OK;
    status_valid_ = false;
    // every frame's store starts from zero (JpegBlockAllocator.Allocate clears it, JpegBlockAllocator.cs:81-83)
    // (the stores of consecutive frames lie back to back: one fill per run of them, not one per frame -- 257 fills, 3.8 ms of
    // a 180 ms step of 256 frames, in round 4)
    for (size_t k = 0; k < prog_clear_.size() && !keep_progressive_store_;) {  // (per-scan boundary: the store holds the scans of earlier calls)
        uint64_t first = prog_clear_[k].first, blocks = prog_clear_[k].second;
        for (k++; k < prog_clear_.size() && prog_clear_[k].first == first + blocks; k++) blocks += prog_clear_[k].second;
        hipError_t e = hipMemsetAsync((int16_t *)d_coefs_.ptr + first * 64, 0, (size_t)blocks * 128, ctx_->stream);
        if (e != hipSuccess) return hip_fail(e, "hipMemsetAsync(progressive store)");
    }
    if (prog_begin_.size() <= 1) return JPGPU_OK;  // (frames without a single scan to run: their stores are zero now, that is all)
    const char *dbg_max = getenv("JPGPU_DEBUG_MAX_PROGRESSIVE_SCANS");  // debugging aid: stop after N scans per frame
    // One pipelined launch, or one launch per level?  One wave per workgroup, LDS bounds them per CU.
    //  - up to three quarters of what the CUs hold: pipelined with the count-in gate (every workgroup resident, see the
    //    residency rule in progressive_stream_kernel; the kernel itself checks);
    //  - up to one and a half times what the CUs hold: still pipelined, no gate -- the work list is ordered by level and MI355X starts
    //    workgroups in list order, so a follower never holds a slot its producer still needs; should that ever not hold, the
    //    bounded spin gives up and fetch_status() falls back to level by level.  Measured (4K 4:2:0 progressive, ms per batch,
    //    level by level vs pipelined): 448 frames 651 / 402, 640: 716 / 588, 896: 791 / 727;
    //  - beyond: level by level (1024 frames: 832 vs 790-885 pipelined, run to run; 2048 frames: 1295 vs 1426 -- by then
    //    every level fills the machine on its own).
    const int n_streams = prog_stream_begin_.back() - prog_stream_begin_.front();
    const int n_waves = prog_pipe_count_;  // of the pipelined launch (<= n_streams)
    const size_t lds_per_wg = (progressive_stream_lds_bytes(n_huff_slots_) + 1023) / 1024 * 1024;
    const int cus = ctx_->num_cus > 0 ? ctx_->num_cus : 256;
    int per_cu = (int)std::min<size_t>(32, (160u << 10) / lds_per_wg);
    {
        // what the runtime says a CU holds of this kernel (registers as well as LDS), asked once per table-slot count
        static std::atomic<int> cached[kMaxHuffSlots + 1];
        const int slot_key = std::min(std::max(n_huff_slots_, 0), kMaxHuffSlots);
        int occ = cached[slot_key].load(std::memory_order_relaxed);
        if (occ == 0) {
            occ = progressive_stream_blocks_per_cu(n_huff_slots_);
            cached[slot_key].store(occ > 0 ? occ : -1, std::memory_order_relaxed);
        }
        if (occ > 0) per_cu = std::min(per_cu, occ);
    }
    const bool force = getenv("JPGPU_PROG_FORCE_PIPELINE") != nullptr;  // experiments: pipelined without the gate, any size
    const bool resident = n_waves <= per_cu * 3 / 4 * cus;
    // (round 3, ADVICE r2: the ungated pipelined launch of grids up to 1.5 x what the CUs hold relied on workgroups starting in
    // list order; it is opt-in now -- JPGPU_PROG_FORCE_PIPELINE -- and larger batches take the chain launches below)
    const bool fits = resident || force;
    const int launch_mode = resident && !force ? 1 : 2;
    const bool no_chains = getenv("JPGPU_PROG_NO_CHAINS") != nullptr;  // A/B switch: level-by-level launches instead
    if (!(prog_pipelined_ && fits) && prog_chains_ok_ && !no_chains && !dbg_max && getenv("JPGPU_PROG_BY_SCAN") == nullptr && !prog_by_scan_ && n_streams > 0) {
        // Batches that do not fit one resident launch.  Scans of different chains -- the DC scans; the AC scans of component 0,
        // 1, 2, 3 -- never touch the same coefficients, scans of one chain follow each other in file order: every chain gets a
        // stream of its own and one launch per ordinal (the j-th scan of the chain in every frame).  No waiting inside a
        // kernel, nothing assumed about dispatch order; the chains fill each other's idle SIMDs (a launch of n one-wave
        // workgroups keeps n / 1024 waves per SIMD busy, and a lone wave issues an instruction every ~5 cycles at best), and
        // the step takes as long as its longest chain instead of the sum over dependency levels.
        jpgpu_ctx *cx = ctx_;
        for (int x = 0; x < jpgpu_ctx::kProgChains; x++) {
            if (!cx->prog_stream[x]) {
                hipError_t e = hipStreamCreateWithFlags(&cx->prog_stream[x], hipStreamNonBlocking);
                if (e != hipSuccess) return hip_fail(e, "hipStreamCreate(progressive chain)");
            }
        }
        for (int x = 0; x <= jpgpu_ctx::kProgChains; x++) {
            if (!cx->prog_ev[x]) {
                hipError_t e = hipEventCreateWithFlags(&cx->prog_ev[x], hipEventDisableTiming);
                if (e != hipSuccess) return hip_fail(e, "hipEventCreate(progressive chain)");
            }
        }
        hipError_t e = hipEventRecord(cx->prog_ev[jpgpu_ctx::kProgChains], cx->stream);  // K1 and the cleared stores are behind this
        if (e != hipSuccess) return hip_fail(e, "hipEventRecord");
        for (int x = 0; x < jpgpu_ctx::kProgChains; x++) {
            if (prog_chain_begin_[x].size() <= 1) continue;
            hipStream_t st = cx->prog_stream[x];
            if ((e = hipStreamWaitEvent(st, cx->prog_ev[jpgpu_ctx::kProgChains], 0)) != hipSuccess) return hip_fail(e, "hipStreamWaitEvent");
            for (size_t jx = 0; jx + 1 < prog_chain_begin_[x].size(); jx++) {
                e = launch_progressive_streams(st, (const uint8_t *)d_unstuffed_.ptr, (const DevScan *)d_scans_.ptr,
                                               (const HuffWork *)d_prog_work_.ptr + prog_chain_begin_[x][jx],
                                               prog_chain_begin_[x][jx + 1] - prog_chain_begin_[x][jx], (const uint32_t *)d_ends_u_.ptr,
                                               (DevScanStatus *)d_status_.ptr, (const DevHuffTable *)d_huff_pool_.ptr, (int16_t *)d_coefs_.ptr,
                                               n_huff_slots_, 0, 0, nullptr);
                if (e != hipSuccess) return hip_fail(e, "progressive_stream_kernel");
            }
            if ((e = hipEventRecord(cx->prog_ev[x], st)) != hipSuccess) return hip_fail(e, "hipEventRecord");
            if ((e = hipStreamWaitEvent(cx->stream, cx->prog_ev[x], 0)) != hipSuccess) return hip_fail(e, "hipStreamWaitEvent");
        }
        return JPGPU_OK;
    }
    if (prog_pipelined_ && fits && !dbg_max) {
        // every scan is one stream: one launch, the work list ordered by level; dependent scans follow their producers' progress
        hipError_t e0 = hipMemsetAsync(d_prog_sync_.ptr, 0, 256, ctx_->stream);
        if (e0 != hipSuccess) return hip_fail(e0, "hipMemsetAsync(progressive sync)");
        const int n = n_waves;
        hipError_t e = launch_progressive_streams(ctx_->stream, (const uint8_t *)d_unstuffed_.ptr, (const DevScan *)d_scans_.ptr,
                                                  (const HuffWork *)d_prog_work_.ptr + prog_pipe_begin_, n,
                                                  (const uint32_t *)d_ends_u_.ptr, (DevScanStatus *)d_status_.ptr,
                                                  (const DevHuffTable *)d_huff_pool_.ptr, (int16_t *)d_coefs_.ptr, n_huff_slots_, launch_mode, prog_spin_budget_, (uint32_t *)d_prog_sync_.ptr);
        if (e != hipSuccess) return hip_fail(e, "progressive_stream_kernel");
        return JPGPU_OK;
    }
    for (size_t k = 0; k + 1 < prog_begin_.size(); k++) {
        if (dbg_max && (int)k >= atoi(dbg_max)) break;
        hipError_t e = launch_progressive(ctx_->stream, (const uint8_t *)d_unstuffed_.ptr, (const DevScan *)d_scans_.ptr,
                                          (const HuffWork *)d_prog_work_.ptr + prog_begin_[k], prog_begin_[k + 1] - prog_begin_[k],
                                          (const uint32_t *)d_ends_u_.ptr, (DevScanStatus *)d_status_.ptr,
                                          (const DevHuffTable *)d_huff_pool_.ptr, (int16_t *)d_coefs_.ptr, n_huff_slots_);
        if (e != hipSuccess) return hip_fail(e, "progressive_scan_kernel");
        e = launch_progressive_streams(ctx_->stream, (const uint8_t *)d_unstuffed_.ptr, (const DevScan *)d_scans_.ptr,
                                       (const HuffWork *)d_prog_work_.ptr + prog_stream_begin_[k],
                                       prog_stream_begin_[k + 1] - prog_stream_begin_[k], (const uint32_t *)d_ends_u_.ptr,
                                       (DevScanStatus *)d_status_.ptr, (const DevHuffTable *)d_huff_pool_.ptr, (int16_t *)d_coefs_.ptr,
                                       n_huff_slots_, 0, 0, nullptr);
        if (e != hipSuccess) return hip_fail(e, "progressive_stream_kernel");
    }
    return JPGPU_OK;
}
int DeviceBatch::clear_partial_outputs() {
    for (const auto &c : out_clear_) {
        hipError_t e = hipMemsetAsync((uint8_t *)d_out_.ptr + c.first, 0, c.second, ctx_->stream);
        if (e == hipSuccess && format_ == JPGPU_FMT_EXTENDED_U16 && c.planes_bytes)
            e = hipMemsetAsync((uint8_t *)d_planes_.ptr + c.planes_first, 0, c.planes_bytes, ctx_->stream);
        if (e == hipSuccess && d_rgb_scratch_.ptr && c.first + c.second <= d_rgb_scratch_.cap)
            e = hipMemsetAsync((uint8_t *)d_rgb_scratch_.ptr + c.first, 0, c.second, ctx_->stream);
        if (e != hipSuccess) return hip_fail(e, "hipMemsetAsync(output)");
    }
    return JPGPU_OK;
}

// the frames' coefficient stores back to zero (JpegBlockAllocator.Allocate clears it): a Dispose() without any scan before it
int DeviceBatch::clear_progressive_stores() {
    dispose_done_ = false;
    for (const auto &c : prog_clear_) {
        hipError_t e = hipMemsetAsync((int16_t *)d_coefs_.ptr + c.first * 64, 0, (size_t)c.second * 128, ctx_->stream);
        if (e != hipSuccess) return hip_fail(e, "hipMemsetAsync(progressive store)");
    }
    return JPGPU_OK;
}

int DeviceBatch::run_dispose_passes(hipStream_t stream) {
    if (dispose_jobs_.empty() || dispose_done_) return JPGPU_OK;
    dispose_done_ = true;
    const hipError_t e = launch_dispose_pass(stream, (int16_t *)d_coefs_.ptr, (const DisposeJob *)d_dispose_.ptr, (int)dispose_jobs_.size(), dispose_max_blocks_,
                                             (const DevQuantTable *)d_quant_pool_.ptr);
    return e == hipSuccess ? JPGPU_OK : hip_fail(e, "dispose_pass_kernel");
}

int DeviceBatch::run_idct() {
    if (k2s_unchecked_) k2s_idct_behind_ = true;
    const YccRgbFactors kf = ycc_rgb_factors();
    int rc0 = clear_partial_outputs();
    if (rc0 != JPGPU_OK) return rc0;
    if ((rc0 = run_dispose_passes(ctx_->stream)) != JPGPU_OK) return rc0;
    const bool extended = format_ == JPGPU_FMT_EXTENDED_U16;
    hipError_t e = launch_idct(ctx_->stream, (const int16_t *)d_coefs_.ptr, (const DevScan *)d_scans_.ptr, (const IdctWork *)d_idct_work_.ptr,
                               idct_class_begin_, (const DevScanStatus *)d_status_.ptr, (const DevQuantTable *)d_quant_pool_.ptr,
                               extended ? (uint8_t *)d_planes_.ptr : (uint8_t *)d_out_.ptr, extended ? (int)JPGPU_FMT_PLANAR_I16 : format_, kf,
                               (uint8_t *)d_rgb_scratch_.ptr);
    if (e != hipSuccess) return hip_fail(e, "idct_output_kernel");
    // scans ordered behind earlier scans of their image (and the failing MCU of a caller's canvas): one bytewise launch per level
    for (size_t lv = 0; lv + 1 < idct_later_begin_.size(); lv++) {
        int cb[kNumIdctLayoutClasses + 1];
        cb[0] = idct_later_begin_[lv];
        for (int c = 1; c <= kNumIdctLayoutClasses; c++) cb[c] = idct_later_begin_[lv + 1];
        e = launch_idct(ctx_->stream, (const int16_t *)d_coefs_.ptr, (const DevScan *)d_scans_.ptr, (const IdctWork *)d_idct_work_.ptr, cb,
                        (const DevScanStatus *)d_status_.ptr, (const DevQuantTable *)d_quant_pool_.ptr,
                        extended ? (uint8_t *)d_planes_.ptr : (uint8_t *)d_out_.ptr, extended ? (int)JPGPU_FMT_PLANAR_I16 : format_, kf,
                        (uint8_t *)d_rgb_scratch_.ptr);
        if (e != hipSuccess) return hip_fail(e, "idct_output_kernel (ordered scans)");
    }
    if (extended) {
        // "O3": the int16 planes (WriteBlock's arguments) through the test writer's clamp + bit expansion: ONE launch for the
        // batch, a descriptor per image (it was a launch per image)
        std::vector<ExtendPlanes> desc;
        uint32_t max_pixels = 0;
        for (const ImagePlan &img : images_) {
            if (img.status != JPGPU_OK || img.jobs.empty() || img.out_bytes == 0) continue;
            ExtendPlanes g;
            memset(&g, 0, sizeof g);
            const BaselineGeometry &geo = jobs_[img.jobs[0]].geo;
            for (int c = 0; c < 4; c++) g.pitch[c] = 1;
            for (int c = 0; c < img.num_components && c < 4; c++) {
                g.plane_off[c] = img.planes_offset + img.plane[c].offset;
                g.pitch[c] = img.plane[c].pitch;
                const int hs = geo.max_h / std::max<int>(1, geo.frame.components[c].h), vs = geo.max_v / std::max<int>(1, geo.frame.components[c].v);
                while ((1 << (g.hshift[c] + 1)) <= hs) g.hshift[c]++;
                while ((1 << (g.vshift[c] + 1)) <= vs) g.vshift[c]++;
                g.hcnt[c] = std::max<int>(1, geo.frame.components[c].h);
                g.vcnt[c] = std::max<int>(1, geo.frame.components[c].v);
            }
            // (0, 0: a progressive frame -- the allocator's Flush places the replicated blocks side by side)
            bool flush = false;
            for (int j : img.jobs) flush |= jobs_[j].kind != kScanSequential;
            g.max_h = flush ? 0u : (uint32_t)geo.max_h;
            g.max_v = flush ? 0u : (uint32_t)geo.max_v;
            g.out_off = img.out_offset;
            g.width = img.width;
            g.height = img.height;
            g.ncomp = img.num_components;
            g.precision = img.precision;
            max_pixels = std::max<uint64_t>(max_pixels, std::min<uint64_t>((uint64_t)img.width * img.height, 0xFFFFFFFFu));
            desc.push_back(g);
        }
        if (!desc.empty()) {
            e = d_extend_desc_.reserve(desc.size() * sizeof(ExtendPlanes));
            if (e != hipSuccess) return hip_fail(e, "hipMalloc(extend descriptors)");
            e = hipMemcpyAsync(d_extend_desc_.ptr, desc.data(), desc.size() * sizeof(ExtendPlanes), hipMemcpyHostToDevice, ctx_->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(ctx_->stream);  // `desc` is a local (pageable) vector; a few KB
            if (e != hipSuccess) return hip_fail(e, "hipMemcpyAsync(extend descriptors)");
            e = launch_extend_u16(ctx_->stream, (const uint8_t *)d_planes_.ptr, (uint8_t *)d_out_.ptr, (const ExtendPlanes *)d_extend_desc_.ptr,
                                  (int)desc.size(), max_pixels);
            if (e != hipSuccess) return hip_fail(e, "extend_u16_kernel");
        }
    }
    for (const RgbConvert &rc : rgb_convert_) {
        e = launch_ycc_to_rgb(ctx_->stream, (const uint8_t *)d_rgb_scratch_.ptr + rc.out_offset, (uint8_t *)d_out_.ptr + rc.out_offset, rc.pixels,
                              rc.components, format_ == JPGPU_FMT_RGBA_U8 ? 4 : 3, kf);
        if (e != hipSuccess) return hip_fail(e, "ycc_to_rgb_kernel");
    }
    return mark_work();
}

// One pass of the device pipeline over the batch.
//
// Issue order.  K2 (Huffman) is bound by the latency of its serial symbol chains and leaves most of the HBM bandwidth idle;
// K3 (IDCT + output) is bound by HBM and leaves most of the issue slots idle.  For batches large enough to matter the batch
// is cut in two halves of images and issued on the context's two streams so that the second half's K2 runs beside the first
// half's K3:
//     stream : K1(all)  K2(A) ------ K3(A) ----------------- [join] 
//     stream2:                 wait  K2(B) ------ K3(B) ------/
// (16.1 vs 17.2 ms per 1024 x 4K measured with two contexts in round 1, with K3 at 10.5 ms; with round 2's kernels it no
// longer pays -- see the numbers at `wanted` in layout_and_upload -- so the mode is opt-in: JPGPU_OVERLAP=1.)
// Kernels that share the machine have no duration of their own, and bench.py's per-kernel roofline is computed from
// exactly that: the first decode() after an upload or a jpgpu_batch_stage_ms query, and every 8th after it, is issued
// serially on one stream with an event between the stages.  jpgpu_batch_stage_ms reports the stage times from those serial
// passes and the whole-pipeline time over all passes.
int DeviceBatch::decode() {
    hipError_t e = hipSetDevice(ctx_->device);
    if (e != hipSuccess) return hip_fail(e, "hipSetDevice");
    if (replay_layout_active_ && !in_replay_) {  // a decode behind a partial-flush replay: a whole pass again, from the batch's own work lists
        const int rr = restore_after_replay();
        if (rr != JPGPU_OK) return rr;
    }
    if (ev_used_ + 4 > 4 * 256) {  // bound the pool: keep the most recent decodes only
        ev_used_ = 0;
        ev_serial_.clear();
    }
    if (ev_pool_.size() < ev_used_ + 4) {  // sixteen decodes' worth at a time: no event is created on the way of a later call
        for (int k = 0; k < 64; k++) {
            hipEvent_t ev = nullptr;
            e = hipEventCreate(&ev);
            if (e != hipSuccess) return hip_fail(e, "hipEventCreate");
            ev_pool_.push_back(ev);
        }
    }
    hipEvent_t *ev = &ev_pool_[ev_used_];
    redo_.clear();  // (a re-planned image's output is this batch's own plan's again behind this decode)
    in_decode_request_ = true;
    const bool serial = !overlap_ok_ || (decodes_since_query_ % kSerialEvery) == 0;
    decodes_since_query_++;
    int rc;
    hipStream_t s1 = ctx_->stream, s2 = ctx_->stream2;
    // (a stage event that could not be recorded would turn jpgpu_batch_stage_ms -- bench.py's roofline -- into a silent lie: checked)
    auto mark = [&](int k, hipStream_t st) { return hipEventRecord(ev[k], st) == hipSuccess ? JPGPU_OK : hip_fail(hipGetLastError(), "hipEventRecord(stage)"); };
    if ((rc = mark(0, s1)) != JPGPU_OK) return rc;
    // One image per call (the reference's callers): an event between two stages is a barrier packet, ~10 us of an idle machine each
    // -- a tenth of such a call.  Batches of up to four images record them on the first decode behind an upload or a
    // jpgpu_batch_stage_ms query and on every 8th after it (the rule of the overlapped mode); the stage times come from those.
    const bool staged = serial && (images_.size() > 4 || ((decodes_since_query_ - 1) % kSerialEvery) == 0);
    if (serial) {
        if ((rc = run_marker_index()) != JPGPU_OK) return rc;
        if (staged && (rc = mark(1, s1)) != JPGPU_OK) return rc;
        if ((rc = run_huffman()) != JPGPU_OK) return rc;
        if (staged && (rc = mark(2, s1)) != JPGPU_OK) return rc;
        if ((rc = run_idct()) != JPGPU_OK) return rc;
        if ((rc = mark(3, s1)) != JPGPU_OK) return rc;
    } else {
        status_valid_ = false;
        const YccRgbFactors kf = ycc_rgb_factors();
        if ((rc = run_marker_index()) != JPGPU_OK) return rc;
        if ((rc = clear_partial_outputs()) != JPGPU_OK) return rc;
        auto k2 = [&](hipStream_t st, int first, int n) {
            return launch_huffman(st, (const uint8_t *)d_unstuffed_.ptr, (const DevScan *)d_scans_.ptr, (const HuffWork *)d_huff_work_.ptr + first, n,
                                  (const uint32_t *)d_ends_u_.ptr, (DevScanStatus *)d_status_.ptr, (const DevHuffTable *)d_huff_pool_.ptr,
                                  (int16_t *)d_coefs_.ptr, n_huff_slots_, (const uint8_t *)d_lut_pool_.ptr, k2_tab_bytes_);
        };
        auto k3 = [&](hipStream_t st, int half) {
            return launch_idct(st, (const int16_t *)d_coefs_.ptr, (const DevScan *)d_scans_.ptr, (const IdctWork *)d_idct_work_split_.ptr,
                               idct_split_begin_[half], (const DevScanStatus *)d_status_.ptr, (const DevQuantTable *)d_quant_pool_.ptr,
                               (uint8_t *)d_out_.ptr, format_, kf, (uint8_t *)d_rgb_scratch_.ptr);
        };
        if ((e = k2(s1, 0, huff_split_)) != hipSuccess) return hip_fail(e, "huffman_decode_kernel");
        if ((rc = mark(1, s1)) != JPGPU_OK) return rc;  // K1 and K2(A) are done: the second half may start
        if ((e = hipStreamWaitEvent(s2, ev[1], 0)) != hipSuccess) return hip_fail(e, "hipStreamWaitEvent");
        if ((e = k2(s2, huff_split_, n_huff_work_ - huff_split_)) != hipSuccess) return hip_fail(e, "huffman_decode_kernel");
        if ((e = k3(s1, 0)) != hipSuccess) return hip_fail(e, "idct_output_kernel");
        if ((e = k3(s2, 1)) != hipSuccess) return hip_fail(e, "idct_output_kernel");
        if ((rc = mark(2, s2)) != JPGPU_OK) return rc;
        if ((e = hipStreamWaitEvent(s1, ev[2], 0)) != hipSuccess) return hip_fail(e, "hipStreamWaitEvent");  // join
        if ((rc = mark(3, s1)) != JPGPU_OK) return rc;
    }
    ev_serial_.push_back(staged);
    ev_used_ += 4;
    return mark_work();
}

// Waits for the device work THIS batch has issued (the event behind its last launch): another batch of the context may be
// decoding on the same stream -- jpgpu_multi_wait waits for call k while call k + 1 runs -- and is not waited for.
int DeviceBatch::sync() {
    hipError_t e = work_in_flight_ && done_ev_ ? hipEventSynchronize(done_ev_) : hipStreamSynchronize(ctx_->stream);
    if (e != hipSuccess) return hip_fail(e, "hipStreamSynchronize");
    work_in_flight_ = false;
    if (k1_onepass_ && h_k1_giveup_ && *h_k1_giveup_ != 0) {  // a group ran out of patience and counted its predecessors itself (tests: budget 0)
        *h_k1_giveup_ = 0;
        k1_fallbacks_++;
    }
    return k2s_unchecked_ ? check_subseq_rounds() : JPGPU_OK;
}

// The device-driven K2S rounds of the decode that has just been waited for: did the enqueued rounds reach the fixed point?
// (a round other than round 0 that changed no exit state).  Yes: the next decode of this upload enqueues exactly as many.
// No (a flat region longer than the budget's walks resolve, a pathological stream): the entropy stage -- and the output stage
// when it was issued behind it -- is issued again with the host reading the counts between rounds, and this upload stays that way.
int DeviceBatch::check_subseq_rounds() {
    k2s_unchecked_ = false;
    uint32_t ctl[64];
    hipError_t e = hipMemcpy(ctl, d_sub_changed_.ptr, sizeof ctl, hipMemcpyDeviceToHost);
    if (e != hipSuccess) return hip_fail(e, "hipMemcpy(K2S control)");
    int used = 0;
    for (int r = 1; r < k2s_issued_ && r < 62; r++)
        if (ctl[r] == 0) {
            used = r + 1;
            break;
        }
    static const bool trace = getenv("JPGPU_SUBSEQ_TRACE") != nullptr;
    if (trace) {
        for (int r = 0; r < k2s_issued_ && r < 62; r++) fprintf(stderr, "K2S round %d: %u changed\n", r, ctl[r]);
        fprintf(stderr, "K2S walks copied: %u; rounds issued %d, used %d\n", ctl[63], k2s_issued_, used);
    }
    if (used != 0 || k2s_issued_ >= (int)max_subs_per_scan_ + 2) {  // (n + 2 rounds always suffice: every round fixes one more subsequence)
        last_subseq_rounds_ = used != 0 ? used : k2s_issued_;
        k2s_budget_ = last_subseq_rounds_;
        return JPGPU_OK;
    }
    k2s_host_checked_ = true;
    k2s_fallbacks_++;
    const bool with_idct = k2s_idct_behind_;
    int rc = run_marker_index();
    if (rc == JPGPU_OK) rc = run_huffman();
    if (rc == JPGPU_OK && with_idct) rc = run_idct();
    if (rc != JPGPU_OK) return rc;
    e = work_in_flight_ && done_ev_ ? hipEventSynchronize(done_ev_) : hipStreamSynchronize(ctx_->stream);
    if (e != hipSuccess) return hip_fail(e, "hipStreamSynchronize");
    work_in_flight_ = false;
    return JPGPU_OK;
}

// Average device time (HIP events on the decode stream) over the decode() calls issued since the previous query: the three
// stages from the passes that were issued serially (see decode()), the whole pipeline over all passes.
int DeviceBatch::stage_ms(float ms[4]) {
    if (ev_used_ == 0) return fail(JPGPU_ERR_INVALID_OPERATION, "jpgpu_batch_stage_ms: no decode has run since the last query");
    int rc = sync();
    if (rc != JPGPU_OK) return rc;
    double acc[4] = {0, 0, 0, 0};
    const size_t n = ev_used_ / 4;
    size_t n_serial = 0;
    for (size_t k = 0; k < n; k++) {
        hipEvent_t *ev = &ev_pool_[k * 4];
        float t;
        if (k < ev_serial_.size() && ev_serial_[k]) {
            for (int i = 0; i < 3; i++) {
                if (hipEventElapsedTime(&t, ev[i], ev[i + 1]) != hipSuccess) return fail(JPGPU_ERR_DEVICE, "hipEventElapsedTime failed");
                acc[i] += t;
            }
            n_serial++;
        }
        if (hipEventElapsedTime(&t, ev[0], ev[3]) != hipSuccess) return fail(JPGPU_ERR_DEVICE, "hipEventElapsedTime failed");
        acc[3] += t;
    }
    for (int i = 0; i < 3; i++) ms[i] = n_serial ? (float)(acc[i] / (double)n_serial) : 0.0f;
    ms[3] = (float)(acc[3] / (double)n);
    ev_used_ = 0;
    ev_serial_.clear();
    decodes_since_query_ = 0;
    return JPGPU_OK;
}

}  // namespace jpgpu
