// jpeglibrary_amd/csrc/device_batch_layout.cpp -- DeviceBatch::layout_and_upload: the HBM layout of an upload (device_batch.cpp has the
// layout's description), the scan descriptors, the table pools and every kernel's work list; the device copies.  (One file with
// device_batch.cpp until round 6; the launches are in device_batch_launch.cpp, results and the partial-flush replay in
// device_batch_result.cpp.)
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

static inline uint64_t align_up(uint64_t v, uint64_t a) { return (v + a - 1) / a * a; }

namespace {
// Workgroup i of a launch runs on XCD i % 8 (round-robin dispatch), each XCD behind its own L2.  Work lists are built in
// memory order; interleaving them gives every XCD one contiguous run, so that neighbouring work items -- which share the
// cache lines at their common boundary -- meet in one L2 instead of writing two halves of a line from two.
template <typename T>
void xcd_interleave(std::vector<T> &w, int xcds) {
    if (xcds <= 1 || w.size() < (size_t)xcds * 8) return;
    const size_t n = w.size(), per = (n + xcds - 1) / xcds;
    std::vector<T> m;
    m.reserve(n);
    for (size_t i = 0; m.size() < n; i++) {
        const size_t src = (i % xcds) * per + i / xcds;
        if (src < n) m.push_back(w[src]);
    }
    w.swap(m);
}
}  // namespace

int DeviceBatch::layout_and_upload(const std::vector<const uint8_t *> &file_ptr, const std::vector<size_t> &file_len) {
    dispose_done_ = false;
    if (!in_replay_) {
        // every upload path starts from the batch's own launch modes (the replay's flags used to be reset by
        // upload_segments alone); upload_segments says afterwards whether a partial flush can become necessary
        replay_done_ = false;
        replay_possible_ = false;
        prog_by_scan_ = false;
        replay_layout_active_ = false;
        replay_saved_jobs_.clear();
        for (ImagePlan &img : images_) img.replay_skip = false;
    }
    hipError_t e = hipSetDevice(ctx_->device);
    if (e != hipSuccess) return hip_fail(e, "hipSetDevice");
    {
        const int rc0 = order_upload_behind_work();
        if (rc0 != JPGPU_OK) return rc0;
    }
    status_valid_ = false;
    ev_used_ = 0;
    keep_progressive_store_ = false;
    defer_refusal_ = false;

    // ---- input layout (jpgpu_batch_upload has laid the files out and sent them already: files_resident_)
    if (!files_resident_) {
        uint64_t in_off = 256;
        for (size_t i = 0; i < images_.size(); i++) {
            images_[i].file_offset = in_off;
            if (images_[i].status == JPGPU_OK && !images_[i].jobs.empty()) in_off = align_up(in_off + file_len[i], 256);
        }
        input_bytes_ = in_off + 256;
    }
    hipStream_t up = ctx_->upload_stream;  // everything an upload does stays off the decode stream

    // debugging switches, read once per upload (not per scan job; not once per process: tests set them between calls)
    int dbg_delay_scan = -1, dbg_delay_ms = 0;
    if (const char *dd = getenv("JPGPU_DEBUG_DELAY_SCAN")) {
        if (sscanf(dd, "%d:%d", &dbg_delay_scan, &dbg_delay_ms) != 2) dbg_delay_scan = -1;
    }
    const bool dbg_status = getenv("JPGPU_DEBUG_STATUS") != nullptr;
    // ---- per-image output / coefficient layout, scan descriptors, pools, work lists
    huff_pool_.clear();
    quant_pool_.clear();
    h_scans_.assign(jobs_.size(), DevScan());
    std::vector<HuffWork> huff_work;
    std::vector<ChunkWork> chunk_work;
    std::vector<ChunkWork> k1_order;  // the one-pass marker index: (scan, first chunk) per group of kMarkerGroupChunks chunks, by (group, scan)
    std::vector<HuffWork> sub_work;
    // DRI = 0 scans are cut into subsequences of 1024 bits; when that gives more lanes than the machine can use anyway the
    // subsequences grow (2048, 4096 bits): a longer one re-synchronises inside itself more often, so fewer rounds
    uint32_t subseq_shift = 10;
    {
        uint64_t dri0_bits = 0;
        for (size_t j = 0; j < jobs_.size(); j++)
            if (jobs_[j].kind == kScanSequential && jobs_[j].geo.restart_interval == 0) dri0_bits += (uint64_t)jobs_[j].entropy_len * 8;
        while (subseq_shift < 12 && (dri0_bits >> subseq_shift) >= 500000u) subseq_shift++;
        // ... and shrink when the batch is ONE small image (round 6: the reference's callers decode one image per call): a round and the
        // final pass last as long as one lane's subsequence, and 1024-bit lanes of a 48 KB scan fill six waves of a 256-CU machine.
        // (512 bits, not 256: the 6-block MCU phase of a 4:2:0 stream re-synchronises over a few hundred bits, and every subsequence
        // it spans is another round -- a 640 x 368 image took 27 rounds of 256-bit lanes, 9 of 512-bit ones)
        while (subseq_shift > 9 && (dri0_bits >> subseq_shift) < 4096u) subseq_shift--;
        if (const char *ev = getenv("JPGPU_SUBSEQ_SHIFT")) subseq_shift = (uint32_t)std::min(14, std::max(8, atoi(ev)));
    }
    // K3 walks runs of consecutive tiles per workgroup (the next tile's coefficients fetched under the current one's transform).
    // A batch of few tiles -- one image per call -- is cut into shorter runs: every CU gets work at once (round 6).
    uint32_t idct_tiles_per_wg = (uint32_t)kIdctTilesPerWg;
    {
        uint64_t blocks = 0;
        for (const ScanJob &job : jobs_)
            if (job.kind != kScanProgressive) blocks += (uint64_t)job.geo.mcus_per_line * job.geo.mcus_per_column * (uint64_t)std::max(1, job.blocks_per_mcu);
        const uint64_t tiles = blocks / (uint64_t)kIdctBlocksPerWg + 1, want_wgs = 3ull * (uint64_t)(ctx_->num_cus > 0 ? ctx_->num_cus : 256);
        while (idct_tiles_per_wg > 1 && tiles / idct_tiles_per_wg < want_wgs) idct_tiles_per_wg /= 2;
    }
    std::vector<std::vector<HuffWork>> prog_work_by_ordinal, prog_streams_by_ordinal;
    std::vector<std::vector<HuffWork>> prog_chain_work[jpgpu_ctx::kProgChains];  // [chain][ordinal in the frame's chain]
    prog_chains_ok_ = true;
    // a scan with fewer restart intervals than this gets one WAVE per interval (progressive_stream_kernel)
    const uint32_t stream_max_intervals = getenv("JPGPU_PROG_STREAM_MAX_INTERVALS") ? (uint32_t)atoi(getenv("JPGPU_PROG_STREAM_MAX_INTERVALS")) : 16u;
    prog_clear_.clear();
    prog_pipelined_ = getenv("JPGPU_PROG_NO_PIPELINE") == nullptr;
    // polls (~2-3 us each) a follower scan of the pipelined launch may spend before it gives up: ~0.3 s by default (a fully
    // resident grid -- the only kind that takes the pipelined launch by default -- makes progress within microseconds)
    prog_spin_budget_ = getenv("JPGPU_PROG_SPIN_BUDGET") ? (uint32_t)strtoul(getenv("JPGPU_PROG_SPIN_BUDGET"), nullptr, 10) : (1u << 17);
    prog_fallbacks_ = 0;
    rgb_convert_.clear();
    sub_scan_ids_.clear();
    total_subs_ = 0;
    max_subs_per_scan_ = 0;
    sub_same_valid_ = false;
    k2s_budget_ = 0;
    k2s_unchecked_ = k2s_idct_behind_ = false;
    // (A/B switch: the host reads the counts between rounds.  Always so over a CALLER'S CANVAS: the device-driven rounds let the
    // output stage run before anybody knows whether they sufficed, and what that stage writes from unconverged states a second,
    // correct pass does not take back where it leaves the canvas alone -- tests/golden/stress/baseline_failing_422_canvas_54.jpg)
    k2s_host_checked_ = getenv("JPGPU_SUBSEQ_HOST_CHECK") != nullptr || keep_canvas_;
    dispose_jobs_.clear();
    dispose_max_blocks_ = 0;
    std::vector<IdctWork> idct_work;
    std::vector<IdctWork> idct_work_by_class[kNumIdctLayoutClasses];
    std::vector<std::vector<IdctWork>> idct_later_levels;  // scans ordered behind earlier scans of their image: one launch per level
    std::vector<IdctWork> idct_partial;                     // "the MCU the scan failed in", bytewise (the caller's canvas)
    std::vector<int> scan_level(jobs_.size(), 0);
    const bool tile_align = !(getenv("JPGPU_TILE_ALIGN") && atoi(getenv("JPGPU_TILE_ALIGN")) == 0);  // A/B switch, default on
    uint64_t out_off = 0, coef_off = 0, planes_off = 0;
    uint32_t ends_off = 0, total_chunks = 0;
    compressed_bytes_ = 0;
    total_pixels_ = 0;
    n_huff_slots_ = 1;
    // (known before any work list is cut: the Huffman workgroup's size -- intervals per work entry -- follows from it)
    for (const ScanJob &job : jobs_) n_huff_slots_ = std::max(n_huff_slots_, job.n_huff);
    // LDS the K2 family stages its tables in: the largest set among the sequential scans (an AC table 9 040 bytes, a DC table
    // 2 896: kernels.h); the waves per workgroup are what it leaves room for
    k2_tab_bytes_ = kK2AcTabBytes + kK2DcTabBytes;
    for (const ScanJob &job : jobs_) {
        if (job.kind != kScanSequential) continue;
        uint32_t bytes = 0;
        for (int k = 0; k < job.n_huff; k++) {
            bool is_dc = false;
            for (int c = 0; c < job.scan_components; c++) is_dc |= job.dc_slot[c] == k;
            bytes += is_dc ? kK2DcTabBytes : kK2AcTabBytes;
        }
        k2_tab_bytes_ = std::max(k2_tab_bytes_, bytes);
    }
    const uint32_t huff_intervals_per_wg = 64u * (uint32_t)huffman_waves(k2_tab_bytes_);
    // the device image of a table depends on BITS / HUFFVAL alone: look those up before building it (a batch of
    // camera files carries the same four tables a thousand times)
    struct HuffKey {
        uint8_t bits[16];
        uint16_t num_values;
        uint8_t values[256];
    };
    std::vector<HuffKey> huff_keys;
    auto huff_index = [&](const HuffTable &t) -> uint16_t {
        for (size_t i = 0; i < huff_keys.size(); i++)
            if (huff_keys[i].num_values == t.num_values && memcmp(huff_keys[i].bits, t.bits, 16) == 0 &&
                memcmp(huff_keys[i].values, t.values, t.num_values) == 0)
                return (uint16_t)i;
        DevHuffTable d;
        t.to_device(&d);
        HuffKey k;
        memcpy(k.bits, t.bits, 16);
        k.num_values = t.num_values;
        memcpy(k.values, t.values, sizeof k.values);
        huff_keys.push_back(k);
        huff_pool_.push_back(d);
        return (uint16_t)(huff_pool_.size() - 1);
    };
    auto quant_index = [&](const QuantTable &t) -> uint16_t {
        DevQuantTable d;
        memcpy(d.q, t.elements, sizeof d.q);
        for (size_t i = 0; i < quant_pool_.size(); i++)
            if (memcmp(&quant_pool_[i], &d, sizeof d) == 0) return (uint16_t)i;
        quant_pool_.push_back(d);
        return (uint16_t)(quant_pool_.size() - 1);
    };

    for (size_t ii = 0; ii < images_.size(); ii++) {
        ImagePlan &img = images_[ii];
        img.out_offset = out_off;
        img.planes_offset = planes_off;
        img.coef_offset = coef_off;
        img.total_blocks = 0;
        if (img.status != JPGPU_OK) continue;
        out_off = align_up(out_off + img.out_bytes, 256);
        planes_off = align_up(planes_off + img.planes_bytes, 256);
        if (!img.jobs.empty()) total_pixels_ += (uint64_t)img.width * img.height;
        // sequential scans of this image that write a component an earlier scan of it has written (ordered launches, see below)
        bool overlapping_scans = false;
        {
            uint32_t seen = 0;
            for (int j : img.jobs) {
                const ScanJob &job = jobs_[j];
                if (job.kind != kScanSequential || job.disabled) continue;
                uint32_t mine = 0;
                for (int c = 0; c < job.scan_components; c++) mine |= 1u << (job.comp[c].component_index & 31);
                overlapping_scans |= (mine & seen) != 0;
                seen |= mine;
            }
        }
        for (int j : img.jobs) {
            const ScanJob &job = jobs_[j];
            DevScan &s = h_scans_[j];
            memset(&s, 0, sizeof s);
            const BaselineGeometry &g = job.geo;
            s.kind = (uint8_t)job.kind;
            s.last_interval = job.last_interval;
            s.data_off = img.file_offset + job_entropy_off_[j];
            s.data_len = (uint32_t)(file_len[ii] - job_entropy_off_[j]);
            if (job.kind == kScanFrameOnly) s.data_len = 0;
            if (job.kind == kScanProgressive) {
                // the segment ends at the next marker that is not RSTn; K1 only has to see that marker
                const size_t end = find_scan_end(job.entropy, job.entropy_len);
                s.data_len = (uint32_t)std::min<size_t>(end + 2, job.entropy_len);
            }
            // progressive entropy scans accumulate into their frame's store (the frame job precedes them)
            s.coef_off = job.kind == kScanProgressive ? h_scans_[img.jobs[0]].coef_off : coef_off;
            s.out_off = format_ == JPGPU_FMT_EXTENDED_U16 ? img.planes_offset : img.out_offset;
            s.dri = job.kind == kScanProgressive ? job.scan_dri : g.restart_interval;
            s.mcus_per_line = (uint32_t)g.mcus_per_line;
            s.mcus_per_column = (uint32_t)g.mcus_per_column;
            s.total_mcus = job.kind == kScanProgressive ? job.total_units : s.mcus_per_line * s.mcus_per_column;
            s.n_intervals = s.dri ? (s.total_mcus + s.dri - 1) / s.dri : 1;
            if (s.total_mcus == 0 || job.kind == kScanFrameOnly) s.n_intervals = 0;
            s.ss = job.ss;
            s.se = job.se;
            s.ah = job.ah;
            s.al = job.al;
            s.frame_bpm = job.frame_bpm;
            s.units_per_line = job.units_per_line;
            for (int c = 0; c < kMaxScanComponents; c++) {
                s.fblk_base[c] = job.fblk_base[c];
                s.hblocks[c] = job.hblocks[c];
                s.vblocks[c] = job.vblocks[c];
            }
            s.ends_off = ends_off;
            ends_off += s.n_intervals;
            s.chunk_off = total_chunks;
            s.n_chunks = (uint32_t)(((uint64_t)s.data_len + (s.data_off & 15u) + kMarkerChunkBytes - 1) / kMarkerChunkBytes);
            if (s.n_chunks == 0) s.n_chunks = 1;
            if (job.kind == kScanFrameOnly) s.n_chunks = 0;  // no entropy data: K1 / K2 skip the job
            // (replay_skip: an image the replay of failed progressive frames leaves alone keeps its place in every buffer and gets no work)
            for (uint32_t c = 0; c < s.n_chunks && !img.replay_skip; c += kMarkerChunksPerWg) chunk_work.push_back({(uint32_t)j, c});
            for (uint32_t c = 0; c < s.n_chunks && !img.replay_skip; c += kMarkerGroupChunks) k1_order.push_back({(uint32_t)j, c});
            total_chunks += s.n_chunks;
            s.image_index = (uint32_t)ii;
            s.first_scan = (uint32_t)img.jobs.front();
            s.level_shift = (uint32_t)g.level_shift;
            s.width = g.frame.samples_per_line;
            s.height = g.frame.lines;
            s.precision = g.frame.precision;
            s.frame_components = g.frame.num_components;
            s.scan_components = (uint8_t)job.scan_components;
            s.max_h = (uint8_t)g.max_h;
            s.max_v = (uint8_t)g.max_v;
            s.blocks_per_mcu = (uint8_t)job.blocks_per_mcu;
            s.restart_check_at_end = (s.dri != 0 && s.total_mcus % s.dri == 0) ? 1 : 0;
            for (int k = 0; k < kMaxHuffSlots; k++) s.huff_pool[k] = 0xFFFF;
            for (int k = 0; k < job.n_huff; k++) s.huff_pool[k] = huff_index(job.huff_copy[k]);
            n_huff_slots_ = std::max(n_huff_slots_, job.n_huff);
            for (int c = 0; c < job.scan_components; c++) {
                DevScanComponent &dc = s.comp[c];
                dc.component_index = (uint8_t)job.comp[c].component_index;
                dc.h = job.comp[c].h;
                dc.v = job.comp[c].v;
                dc.hs = (uint8_t)job.comp[c].hs;
                dc.vs = (uint8_t)job.comp[c].vs;
                dc.quant_slot = (uint8_t)c;
                dc.dc_slot = job.dc_slot[c];
                dc.ac_slot = job.ac_slot[c];
                s.quant_pool[c] = quant_index(job.quant_copy[c]);
                const int fc = job.comp[c].component_index;
                if (fc < 4) {
                    s.plane_off[c] = img.plane[fc].offset;
                    s.plane_pitch[c] = img.plane[fc].pitch;
                }
            }
            s.shadow_mask = keep_canvas_ ? kKeepUnreachedMcus : 0;
            for (int c = 0; c < job.scan_components; c++)
                for (int d = c + 1; d < job.scan_components; d++)
                    if (job.comp[c].component_index == job.comp[d].component_index) s.shadow_mask |= (uint8_t)(1u << c);
            memcpy(s.blk_comp, job.blk_comp, sizeof s.blk_comp);
            memcpy(s.blk_x, job.blk_x, sizeof s.blk_x);
            memcpy(s.blk_y, job.blk_y, sizeof s.blk_y);
            if (job.kind == kScanFrameOnly && job.dispose_generic) {
                // Dispose() as the reference runs it (dispose_pass_kernel), then the store goes to the writer as it is
                s.reserved0 |= kScanStoreHoldsSamples;
                DisposeJob dj;
                memset(&dj, 0, sizeof dj);
                dj.bpm = (uint32_t)job.blocks_per_mcu;
                dj.level_shift = 1u << (g.frame.precision - 1);
                memcpy(dj.blk_comp, job.blk_comp, sizeof dj.blk_comp);
                for (int c = 0; c < kMaxScanComponents; c++) {
                    dj.n[c] = job.dispose_n[c];
                    for (int t = 0; t < job.dispose_n[c]; t++) dj.quant[c][t] = quant_index(job.dispose_q[c][t]);
                }
                dj.n_blocks = 0xFFFFFFFFu;  // (marks "scan index in coef_off": resolved below, once the store's place is known)
                dj.coef_off = (uint64_t)j;
                if (!img.replay_skip) dispose_jobs_.push_back(dj);
            }

            if (job.kind == kScanProgressive) {
                // launch groups: dependency levels; JPGPU_PROG_BY_SCAN=1 (profiling aid): scan k of every frame in a launch of its
                // own, in file order -- one kernel duration per scan kind of the script (tools/trace/progressive_by_scan.sh)
                static const bool by_scan_env = getenv("JPGPU_PROG_BY_SCAN") != nullptr;
                const bool by_scan = by_scan_env || prog_by_scan_;
                if (job.disabled || img.replay_skip) continue;  // (replay of a failed file: the reference never got to this scan; K1 still indexes it)
                const int ordinal = by_scan ? j - img.jobs[0] - 1 : job.ordinal;
                if (by_scan) prog_pipelined_ = false;
                if ((size_t)ordinal >= prog_work_by_ordinal.size()) {
                    prog_work_by_ordinal.resize((size_t)ordinal + 1);
                    prog_streams_by_ordinal.resize((size_t)ordinal + 1);
                }
                // the frame job precedes its scans in the job list: scan k of the frame is job img.jobs[0] + 1 + k
                for (int d = 0; d < 3; d++) s.dep[d] = d < job.n_deps && job.deps[d] >= 0 ? (uint32_t)(img.jobs[0] + 1 + job.deps[d]) : kNoDep;
                s.publishes = job.has_consumers ? 1 : 0;
                // test hook: "k:ms" makes scan k (0-based, in file order) of every progressive frame slow: it idles ms at its start and
                // after every progress word -- a producer its followers catch up with, which the launches only produce by chance
                if (dbg_delay_scan >= 0 && dbg_delay_scan == j - img.jobs[0] - 1) s.debug_delay_ms = (uint8_t)std::min(255, std::max(0, dbg_delay_ms));
                // an AC scan whose band is empty (Ss > Se: a corrupted header; the reference's loops over the band then never run and
                // the scan reads no bit) goes to the lane kernel, whose loops are the reference's: the stream kernel's block decoders
                // are written for a band of at least one coefficient (tests/golden/stress/progressive_se_below_ss_122.jpg)
                const bool empty_band = job.ss != 0 && job.ss > job.se;
                if (job.n_deps > 3 || s.n_intervals != 1 || s.n_intervals > stream_max_intervals || empty_band || job.force_lane) prog_pipelined_ = false;
                if (s.n_intervals <= stream_max_intervals && !job.force_lane && !empty_band) {
                    for (uint32_t i = 0; i < s.n_intervals; i++) prog_streams_by_ordinal[ordinal].push_back({(uint32_t)j, i});
                    // chain of the scan: DC scans (interleaved, or Ss = 0) touch coefficient 0 only, an AC scan the band of ONE
                    // component (what it may write beyond its header stays inside that component's AC coefficients)
                    const int chain = (job.scan_components != 1 || job.ss == 0) ? 0 : 1 + std::min(3, job.comp[0].component_index);
                    int in_chain = 0;
                    for (int jj = img.jobs[0] + 1; jj < j; jj++) {
                        const ScanJob &o = jobs_[jj];
                        in_chain += ((o.scan_components != 1 || o.ss == 0) ? 0 : 1 + std::min(3, o.comp[0].component_index)) == chain;
                    }
                    if (prog_chain_work[chain].size() <= (size_t)in_chain) prog_chain_work[chain].resize((size_t)in_chain + 1);
                    for (uint32_t i = 0; i < s.n_intervals; i++) prog_chain_work[chain][(size_t)in_chain].push_back({(uint32_t)j, i});
                } else {
                    prog_chains_ok_ = false;  // a scan of many restart intervals (lane kernel): level by level
                    for (uint32_t first = 0; first < s.n_intervals; first += 256) prog_work_by_ordinal[ordinal].push_back({(uint32_t)j, first});
                }
                compressed_bytes_ += s.data_len;
                continue;  // no store of its own, no IDCT work
            }
            const uint64_t nblocks = (uint64_t)s.mcus_per_line * s.mcus_per_column * s.blocks_per_mcu;
            if (job.kind == kScanFrameOnly && !img.replay_skip) prog_clear_.push_back({coef_off, nblocks});
            coef_off += nblocks;
            img.total_blocks += nblocks;
            compressed_bytes_ += s.data_len;
            // scans without restart intervals are decoded by the self-synchronising subsequence decoder (K2S)
            bool null_table = false;
            for (int c = 0; c < job.scan_components; c++) null_table |= job.dc_slot[c] == kNullHuffSlot || job.ac_slot[c] == kNullHuffSlot;
            const bool use_subseq = s.dri == 0 && s.total_mcus > 0 && s.data_len >= 512 && !null_table && getenv("JPGPU_NO_SUBSEQ") == nullptr &&
                                    !(entropy_only_ && (size_t)ii < preset_no_subseq_.size() && preset_no_subseq_[ii]);
            if (job.kind == kScanFrameOnly) {
                // the Dispose() pass: IDCT work only
            } else if (use_subseq) {
                s.sub_shift = (uint8_t)subseq_shift;
                s.n_subs = (uint32_t)((((uint64_t)s.data_len * 8) + ((1u << subseq_shift) - 1)) >> subseq_shift);
                s.sub_off = total_subs_;
                total_subs_ += s.n_subs;
                max_subs_per_scan_ = std::max(max_subs_per_scan_, s.n_subs);
                if (!img.replay_skip) {
                    sub_scan_ids_.push_back((uint32_t)j);
                    for (uint32_t first = 0; first < s.n_subs; first += 256) sub_work.push_back({(uint32_t)j, first});
                }
            } else {
                for (uint32_t first = 0; first < s.n_intervals && !img.replay_skip; first += huff_intervals_per_wg) huff_work.push_back({(uint32_t)j, first});
            }
            if (img.replay_skip) continue;  // (its samples are in the output buffer already)
            if (s.blocks_per_mcu == 0) continue;  // cannot happen for a resolved scan (sampling factors are checked); no blocks, no work
            uint32_t mcus_per_wg = (uint32_t)kIdctBlocksPerWg / s.blocks_per_mcu;
            if (tile_align) {
                // a tile whose pixel rows are whole 128-byte lines: neighbouring tiles (other workgroups, other XCDs, other
                // L2s) never write two halves of one line
                uint32_t row_bytes = 8u * s.max_h;
                if (format_ == JPGPU_FMT_INTERLEAVED_U8) row_bytes *= s.frame_components;
                else if (format_ == JPGPU_FMT_RGB_U8) row_bytes *= 3;
                else if (format_ == JPGPU_FMT_RGBA_U8) row_bytes *= 4;
                else if (format_ == JPGPU_FMT_PLANAR_I16 || format_ == JPGPU_FMT_EXTENDED_U16) row_bytes *= 2;
                for (uint32_t t = mcus_per_wg; t * 4 >= mcus_per_wg * 3 && t > 0; t--)
                    if ((t * row_bytes) % 128 == 0) {
                        mcus_per_wg = t;
                        break;
                    }
            }
            const bool holds_samples = (s.reserved0 & kScanStoreHoldsSamples) != 0;
            int cls = holds_samples ? kIdctClassStoreHoldsSamples : (fmt_is_interleaved(format_) ? idct_layout_class(s) : 0);
            // RGB / RGBA of a frame with overlapping scans (below): every scan's samples go to the scratch image in file order and
            // are converted at the end -- a fused conversion of one scan would be overwritten by the scratch image's
            if (!holds_samples && overlapping_scans && (format_ == JPGPU_FMT_RGB_U8 || format_ == JPGPU_FMT_RGBA_U8)) cls = 0;
            if ((cls == 0 || holds_samples) && (format_ == JPGPU_FMT_RGB_U8 || format_ == JPGPU_FMT_RGBA_U8)) {
                // no fused conversion for this layout: samples go to the scratch image, then ycc_to_rgb_kernel
                bool listed = false;
                for (const RgbConvert &rc : rgb_convert_) listed |= rc.image == (uint32_t)ii;
                if (!listed) rgb_convert_.push_back({(uint32_t)ii, img.out_offset, (uint64_t)img.width * img.height, img.num_components});
            }
            // A baseline frame whose LATER scans write a component this one writes too (a corrupted selector: one component
            // scanned twice, another never): in the reference the later WriteBlock wins -- where the later scan GOT to; a later
            // scan that fails or ends early leaves this scan's samples behind its last block -- and two scans' tiles in one
            // launch have no order.  Such a later scan is ORDERED behind the scans it overlaps: its transform goes to a later
            // launch (level = 1 + the highest level among the earlier scans sharing a component with it), on the bytewise form,
            // touching nothing it did not reach (kKeepUnreachedMcus).  Round 4 left the earlier scan's transform out when the
            // later scans covered all its components (wrong when they were truncated) and did not order partial overlaps.
            int level = 0;
            if (job.kind == kScanSequential && img.jobs.size() > 1) {
                uint32_t mine = 0;
                for (int c = 0; c < job.scan_components; c++) mine |= 1u << (job.comp[c].component_index & 31);
                for (int j2 : img.jobs) {
                    if (j2 == j) break;
                    const ScanJob &o = jobs_[(size_t)j2];
                    if (o.kind != kScanSequential || o.disabled) continue;
                    uint32_t theirs = 0;
                    for (int c = 0; c < o.scan_components; c++) theirs |= 1u << (o.comp[c].component_index & 31);
                    if (mine & theirs) level = std::max(level, scan_level[(size_t)j2] + 1);
                }
            }
            scan_level[(size_t)j] = level;
            const uint32_t run = mcus_per_wg * idct_tiles_per_wg;
            if (level > 0) {
                h_scans_[j].shadow_mask |= kKeepUnreachedMcus;
                if ((size_t)level > idct_later_levels.size()) idct_later_levels.resize((size_t)level);
                const uint32_t generic_per_wg = ((uint32_t)kIdctBlocksPerWg / s.blocks_per_mcu);
                const uint32_t grun = generic_per_wg * idct_tiles_per_wg;
                for (uint32_t first = 0; first < s.total_mcus; first += grun)
                    idct_later_levels[(size_t)level - 1].push_back({(uint32_t)j, first, std::min(grun, s.total_mcus - first), generic_per_wg});
                if (format_ == JPGPU_FMT_RGB_U8 || format_ == JPGPU_FMT_RGBA_U8) {  // (bytewise form: samples to the scratch image, then ycc_to_rgb_kernel)
                    bool listed = false;
                    for (const RgbConvert &rc : rgb_convert_) listed |= rc.image == (uint32_t)ii;
                    if (!listed) rgb_convert_.push_back({(uint32_t)ii, img.out_offset, (uint64_t)img.width * img.height, img.num_components});
                }
            } else {
                for (uint32_t first = 0; first < s.total_mcus; first += run)
                    idct_work_by_class[cls].push_back({(uint32_t)j, first, std::min(run, s.total_mcus - first), mcus_per_wg});
                // the caller's canvas under a whole-pixel layout: the MCU a failing scan stops in is written block by block
                if (keep_canvas_ && job.kind == kScanSequential && format_ == JPGPU_FMT_INTERLEAVED_U8 &&
                    cls >= 1 && cls <= 3)  // (kLayYccH1V1 / H2V1 / H2V2, k3_idct.hip)
                    idct_partial.push_back({(uint32_t)j, kIdctPartialMcu, 1, ((uint32_t)kIdctBlocksPerWg / s.blocks_per_mcu)});
            }
        }
    }
    for (DisposeJob &dj : dispose_jobs_) {  // the frames' stores have their places now
        const DevScan &fs = h_scans_[(size_t)dj.coef_off];
        dj.coef_off = fs.coef_off;
        dj.n_blocks = fs.total_mcus * (uint32_t)fs.blocks_per_mcu;
        dispose_max_blocks_ = std::max(dispose_max_blocks_, dj.n_blocks);
    }
    total_blocks_ = coef_off;
    out_bytes_ = out_off;
    planes_bytes_ = planes_off;
    total_ends_ = ends_off;
    // K2's work.  Runs of consecutive scans that stage the same tables in the same slots (all of a batch of files from one encoder,
    // typically) are POOLED (round 6; the K2S final pass below has had it since round 5): one entry per WAVE of 64 restart intervals,
    // taken from a counter by the waves of one workgroup per CU (huffman_pool_kernel).  Everything else keeps a workgroup per
    // huffman_waves() * 64 intervals of one scan.  (JPGPU_OVERLAP, the opt-in two-stream issue order, splits the plain list: no pools then.)
    auto same_tables = [&](const DevScan &a, const DevScan &b) {
        return memcmp(a.huff_pool, b.huff_pool, sizeof a.huff_pool) == 0 && a.scan_components == b.scan_components &&
               a.blocks_per_mcu == b.blocks_per_mcu && memcmp(a.blk_comp, b.blk_comp, sizeof a.blk_comp) == 0 &&
               memcmp(a.comp, b.comp, sizeof a.comp) == 0;
    };
    k2_pools_.clear();
    {
        std::vector<HuffWork> plain, pooled;
#ifdef JPGPU_K2_NO_POOL  // (A/B build variant: tools/trace/ab_build.sh "-DJPGPU_K2_NO_POOL" ...)
        const bool no_pool = true;
#else
        const bool no_pool = getenv("JPGPU_OVERLAP") != nullptr && atoi(getenv("JPGPU_OVERLAP")) != 0;
#endif
        size_t i = 0;
        while (i < huff_work.size() && !no_pool) {
            size_t k = i;
            uint64_t chunks = 0;
            const DevScan &a = h_scans_[huff_work[i].scan];
            while (k < huff_work.size() && same_tables(a, h_scans_[huff_work[k].scan])) {
                if (huff_work[k].first_interval == 0) chunks += (h_scans_[huff_work[k].scan].n_intervals + 63u) / 64u;
                k++;
            }
            const bool pool = (int)k2_pools_.size() < kK2MaxPools && chunks >= 2 && chunks < 0x7FFFFFFFu;
            if (pool) k2_pools_.push_back({(int)pooled.size(), (int)chunks});
            for (size_t q = i; q < k; q++) {
                if (!pool) plain.push_back(huff_work[q]);
                else if (huff_work[q].first_interval == 0)
                    for (uint32_t first = 0; first < h_scans_[huff_work[q].scan].n_intervals; first += 64u) pooled.push_back({huff_work[q].scan, first});
            }
            i = k;
        }
        if (!no_pool) huff_work.swap(plain);
        n_huff_work_ = (int)huff_work.size();
        huff_work.insert(huff_work.end(), pooled.begin(), pooled.end());  // (one buffer: the pooled list behind the plain one)
    }
    for (uint32_t &b : k2_ticket_base_) b = 0;
    // The K2S round kernel's lookups: one set per distinct set of tables among the DRI = 0 scans (one for a batch from one encoder),
    // built once per upload (sr_lut_build_kernel) -- the workgroups of every round built them themselves until round 6.
    std::vector<uint32_t> sr_set_scan;
    for (uint32_t j : sub_scan_ids_) {
        uint32_t k = 0;
        while (k < sr_set_scan.size() && !same_tables(h_scans_[sr_set_scan[k]], h_scans_[j])) k++;
        if (k == sr_set_scan.size()) sr_set_scan.push_back(j);
        h_scans_[j].sr_set = k;
    }
    n_chunk_work_ = (int)chunk_work.size();
    std::stable_sort(k1_order.begin(), k1_order.end(), [](const ChunkWork &a, const ChunkWork &b) { return a.chunk < b.chunk; });
    n_k1_groups_ = (int)k1_order.size();
    n_sub_work_ = (int)sub_work.size();
    // ... and the list of the rounds behind round 1 (a workgroup per kSubseqGatherSpan subsequences), behind it in the same buffer
    for (uint32_t j : sub_scan_ids_)
        for (uint32_t first = 0; first < h_scans_[j].n_subs; first += kSubseqGatherSpan) sub_work.push_back({j, first});
    n_sub_gather_ = (int)sub_work.size() - n_sub_work_;
    // The final pass.  Runs of consecutive scans that stage the same tables in the same slots (all of a batch of files from one
    // encoder, typically) are POOLED: one entry per wave of 64 lanes, taken from a counter by the waves of one workgroup per CU
    // (k2s_subseq.hip).  Everything else takes a workgroup per subseq_final_waves() waves of one scan.
    std::vector<HuffWork> sub_final_work;
    sub_pools_.clear();
    {
#ifdef JPGPU_SF_NO_POOL  // (A/B build variants since round 6: tools/trace/ab_build.sh "-DJPGPU_SF_NO_POOL" ...)
        const bool no_pool = true;
#else
        const bool no_pool = !subseq_pool_fits(k2_tab_bytes_);  // (ten waves + the tables must fit a CU)
#endif
        // (a batch that cannot fill the machine with two subsequences per lane takes one: twice the waves, half as long each)
#ifdef JPGPU_SF_ONE_SUB  // (A/B build variant)
        sub_final_spl_ = 1;
#else
        sub_final_spl_ = (kSubFinalSubsPerLane >= 2 && total_subs_ >= kSubFinalFewSubs) ? 2 : 1;
#endif
        const uint32_t lanes_subs = 64u * (uint32_t)sub_final_spl_;
        std::vector<HuffWork> pooled;
        size_t i = 0;
        while (i < sub_scan_ids_.size()) {
            size_t k = i + 1;
            uint64_t waves = (h_scans_[sub_scan_ids_[i]].n_subs + lanes_subs - 1) / lanes_subs;
            while (k < sub_scan_ids_.size() && same_tables(h_scans_[sub_scan_ids_[i]], h_scans_[sub_scan_ids_[k]])) {
                waves += (h_scans_[sub_scan_ids_[k]].n_subs + lanes_subs - 1) / lanes_subs;
                k++;
            }
            const bool pool = !no_pool && waves >= (uint64_t)kSubFinalPoolMinChunks && (int)sub_pools_.size() < kSubFinalMaxPools && waves < 0x7FFFFFFFu;
            if (pool) sub_pools_.push_back({(int)pooled.size(), (int)waves});
            for (size_t q = i; q < k; q++) {
                const uint32_t j = sub_scan_ids_[q];
                const uint32_t step = pool ? lanes_subs : lanes_subs * (uint32_t)subseq_final_waves();
                for (uint32_t first = 0; first < h_scans_[j].n_subs; first += step) (pool ? pooled : sub_final_work).push_back({j, first});
            }
            i = k;
        }
        n_sub_final_work_ = (int)sub_final_work.size();
        sub_final_work.insert(sub_final_work.end(), pooled.begin(), pooled.end());  // (one buffer: the pooled list behind the plain one)
    }
    n_sub_scans_ = (int)sub_scan_ids_.size();
    std::vector<HuffWork> prog_work;
    prog_begin_.assign(1, 0);
    for (const std::vector<HuffWork> &w : prog_work_by_ordinal) {
        prog_work.insert(prog_work.end(), w.begin(), w.end());
        prog_begin_.push_back((int)prog_work.size());
    }
    prog_stream_begin_.assign(1, (int)prog_work.size());
    for (const std::vector<HuffWork> &w : prog_streams_by_ordinal) {
        prog_work.insert(prog_work.end(), w.begin(), w.end());
        prog_stream_begin_.push_back((int)prog_work.size());
    }
    // The pipelined launch's list.  Fewer waves per frame let more frames share one resident launch (8 instead of 10: 512 frames
    // instead of 409 with two Huffman tables staged; at 256 frames the launch is as fast either way -- what made it slower than
    // a 32-frame launch was the release fences, see progressive_stream_kernel): a scan whose ONLY producer is a cheap scan, and
    // that is cheap itself, runs behind its producer in the same wave -- it could not have overtaken it anyway.  "Cheap": the wave's entropy bytes stay under half of the frame's largest scan (the
    // long pole is left alone) and under 1 MiB, nothing follows the chained scan, and it is an AC scan.  libjpeg's script: Cr first ->
    // Cr refinement, Cb first -> Cb refinement: 8 waves per frame instead of 10.
    prog_pipe_begin_ = (int)prog_work.size();
    {
        static const bool no_wave_chains = getenv("JPGPU_PROG_NO_WAVE_CHAINS") != nullptr;
        std::vector<uint8_t> is_tail(h_scans_.size(), 0);
        for (const ImagePlan &img : images_) {
            if (img.jobs.size() < 2 || no_wave_chains) continue;
            const int j0 = img.jobs[0] + 1, n = (int)img.jobs.size() - 1;  // the frame job precedes its scans
            uint64_t largest = 0;
            // (a scan's own entropy bytes: DevScan::data_len; ScanJob::entropy_len runs to the end of the file)
            for (int k = 0; k < n; k++)
                if (jobs_[(size_t)(j0 + k)].kind == kScanProgressive) largest = std::max<uint64_t>(largest, h_scans_[(size_t)(j0 + k)].data_len);
            std::vector<uint64_t> wave_bytes((size_t)n, 0);  // of the wave that ENDS with scan k
            std::vector<uint8_t> has_next((size_t)n, 0);
            for (int k = 0; k < n; k++) {
                const ScanJob &job = jobs_[(size_t)(j0 + k)];
                if (job.kind != kScanProgressive) continue;
                const uint64_t own = h_scans_[(size_t)(j0 + k)].data_len;
                wave_bytes[(size_t)k] = own;
                // (only a scan nobody follows: behind its producer it starts later than it would beside it, and whatever waited
                // for it would start later too -- Y AC 1-5 -> Y AC 6-63 -> Y refinement in one wave put 60 ms in front of the
                // last luma refinement)
                if (job.n_deps != 1 || job.deps[0] < 0 || job.deps[0] >= k || job.has_consumers) continue;
                // (not a DC refinement: 7 waves per frame instead of 8 only helps batches of 513-585 frames and has not been
                // measured since the wrong frame once seen with it turned out to be the end-of-band skip in the stream kernel)
                if (job.scan_components != 1 || job.ss == 0) continue;
                const int a = job.deps[0];
                if (jobs_[(size_t)(j0 + a)].kind != kScanProgressive || has_next[(size_t)a] || k - a > 255) continue;
                const uint64_t together = wave_bytes[(size_t)a] + own;
                if (together > largest / 2 || together > (1u << 20)) continue;
                if (dbg_status && &img == &images_[0])
                    fprintf(stderr, "jpgpu: wave chain: scan %d behind scan %d (%llu + %llu bytes, largest scan %llu)\n", k, a,
                            (unsigned long long)wave_bytes[(size_t)a], (unsigned long long)own, (unsigned long long)largest);
                h_scans_[(size_t)(j0 + a)].wave_next = (uint8_t)(k - a);
                has_next[(size_t)a] = 1;
                is_tail[(size_t)(j0 + k)] = 1;
                wave_bytes[(size_t)k] = together;
            }
        }
        for (const std::vector<HuffWork> &w : prog_streams_by_ordinal)
            for (const HuffWork &hw : w)
                if (!is_tail[hw.scan]) prog_work.push_back(hw);
    }
    prog_pipe_count_ = (int)prog_work.size() - prog_pipe_begin_;
    for (int x = 0; x < jpgpu_ctx::kProgChains; x++) {
        prog_chain_begin_[x].assign(1, (int)prog_work.size());
        for (const std::vector<HuffWork> &w : prog_chain_work[x]) {
            prog_work.insert(prog_work.end(), w.begin(), w.end());
            prog_chain_begin_[x].push_back((int)prog_work.size());
        }
    }
    // Images whose scans do not cover every frame component (a scan header that names one component twice and another never;
    // no scan at all): the reference leaves those samples of the caller's buffer alone.  The batch owns the output buffer,
    // so "alone" is defined as zero, what a freshly allocated managed array holds.
    out_clear_.clear();
    for (const ImagePlan &img : images_) {
        if (img.status != JPGPU_OK || img.out_bytes == 0) continue;
        // frame components are 0..254 (a baseline frame may carry up to 255; scans are capped at 4): a 256-bit set
        uint64_t covered[4] = {0, 0, 0, 0};
        for (int j : img.jobs) {
            const DevScan &s = h_scans_[j];
            if (jobs_[j].kind == kScanProgressive) continue;
            for (int c = 0; c < s.scan_components; c++) covered[s.comp[c].component_index >> 6] |= 1ull << (s.comp[c].component_index & 63);
        }
        bool all = true;
        // ... and a component whose sampling factor is neither the frame's maximum nor 1 (round 6): the decoder's (offsetX + x) * 8
        // placement leaves the last columns / rows of every MCU unwritten (k3_idct.hip, interleaved_output_from_tile)
        for (int j : img.jobs) {
            const DevScan &s = h_scans_[j];
            if (jobs_[j].kind != kScanSequential) continue;  // (a progressive frame's Flush places the blocks side by side)
            for (int c = 0; c < s.scan_components; c++) all &= !((s.comp[c].h > 1 && s.comp[c].hs > 1) || (s.comp[c].v > 1 && s.comp[c].vs > 1));
        }
        for (int c = 0; c < img.num_components; c++) all &= ((covered[c >> 6] >> (c & 63)) & 1ull) != 0;
        if (!all && !img.replay_skip) out_clear_.push_back({img.out_offset, img.out_bytes, img.planes_offset, img.planes_bytes});
        // RGB / RGBA = the callers' converter applied to the YCbCr8 buffer (DecodeAction.cs:71-74): an image without any scan
        // leaves that buffer as it was (zero here), and the converter still runs over it
        if (img.jobs.empty() && !img.replay_skip && (format_ == JPGPU_FMT_RGB_U8 || format_ == JPGPU_FMT_RGBA_U8))
            rgb_convert_.push_back({(uint32_t)(&img - images_.data()), img.out_offset, (uint64_t)img.width * img.height, img.num_components});
    }
    idct_class_begin_[0] = 0;
    const int xcds = getenv("JPGPU_XCD_MAP") ? atoi(getenv("JPGPU_XCD_MAP")) : 8;  // MI355X: 8 XCDs; 0 / 1 = memory order (A/B switch)
    for (int c = 0; c < kNumIdctLayoutClasses; c++) {
        std::vector<IdctWork> &w = idct_work_by_class[c];
        xcd_interleave(w, xcds);
        idct_work.insert(idct_work.end(), w.begin(), w.end());
        idct_class_begin_[c + 1] = (int)idct_work.size();
    }
    n_idct_work_ = (int)idct_work.size();
    idct_later_begin_.assign(1, n_idct_work_);
    for (const std::vector<IdctWork> &w : idct_later_levels) {
        idct_work.insert(idct_work.end(), w.begin(), w.end());
        idct_later_begin_.push_back((int)idct_work.size());
    }
    if (!idct_partial.empty()) {
        idct_work.insert(idct_work.end(), idct_partial.begin(), idct_partial.end());
        idct_later_begin_.push_back((int)idct_work.size());
    }

    // ---- two halves for decode()'s overlapped issue order (see decode()): images [0, split) and [split, n), balanced by
    // blocks; the Huffman work list is in image order already (one index splits it), the IDCT work gets a second list with
    // the classes of each half interleaved over the XCDs on their own
    overlap_ok_ = false;
    std::vector<IdctWork> idct_work_split;
    {
        // OFF by default: with the round-2 kernels both stages are HBM-heavy (K2 writes the coefficient buffer at 4.7 TB/s)
        // and cannot share a CU (K2 takes 159 KB of its 160 KB LDS), so the halves time-slice instead of overlapping:
        // 16.39-16.52 ms overlapped vs 16.16-16.37 ms serial per 1024 x 4K (gpurun r02e, both issue orders below)
        const char *ev = getenv("JPGPU_OVERLAP");
        const bool wanted = ev && atoi(ev) != 0 && dispose_jobs_.empty();  // (the generic Dispose() pass is issued by run_idct alone)
        uint32_t split_image = 0;
        uint64_t acc = 0;
        for (size_t ii = 0; ii < images_.size() && acc * 2 < total_blocks_; ii++) {
            acc += images_[ii].total_blocks;
            split_image = (uint32_t)ii + 1;
        }
        huff_split_ = 0;
        while (huff_split_ < n_huff_work_ && h_scans_[huff_work[huff_split_].scan].image_index < split_image) huff_split_++;
        // worth it for batches that keep the machine busy for milliseconds: the split costs four launches and two events
        if (wanted && !entropy_only_ && sub_work.empty() && prog_work.empty() && total_blocks_ >= (4u << 20) && huff_split_ > 0 &&
            huff_split_ < n_huff_work_ && rgb_convert_.empty() && format_ != JPGPU_FMT_EXTENDED_U16 && idct_later_begin_.size() == 1) {
            for (int half = 0; half < 2; half++) {
                idct_split_begin_[half][0] = (int)idct_work_split.size();
                for (int c = 0; c < kNumIdctLayoutClasses; c++) {
                    std::vector<IdctWork> w;
                    // idct_work_by_class[c] was interleaved above: take the entries back in memory order
                    for (const IdctWork &x : idct_work_by_class[c])
                        if ((h_scans_[x.scan].image_index < split_image) == (half == 0)) w.push_back(x);
                    std::sort(w.begin(), w.end(), [](const IdctWork &a, const IdctWork &b) { return a.scan != b.scan ? a.scan < b.scan : a.first_mcu < b.first_mcu; });
                    xcd_interleave(w, xcds);
                    idct_work_split.insert(idct_work_split.end(), w.begin(), w.end());
                    idct_split_begin_[half][c + 1] = (int)idct_work_split.size();
                }
            }
            overlap_ok_ = true;
        }
    }
    decodes_since_query_ = 0;

    h_status_.assign(jobs_.size(), DevScanStatus());
    for (size_t j = 0; j < jobs_.size(); j++) {
        DevScanStatus &st = h_status_[j];
        memset(&st, 0, sizeof st);
        st.first_error = kNoError;
        st.decoded_mcus = h_scans_[j].total_mcus;
    }

    // ---- allocate + H2D
    struct Up {
        DevBuffer *buf;
        const void *src;
        size_t bytes;
        size_t reserve;
    };
    const Up ups[] = {
        {&d_scans_, h_scans_.data(), h_scans_.size() * sizeof(DevScan), 0},
        {&d_status_, h_status_.data(), h_status_.size() * sizeof(DevScanStatus), 0},
        {&d_huff_pool_, huff_pool_.data(), huff_pool_.size() * sizeof(DevHuffTable), 0},
        {&d_quant_pool_, quant_pool_.data(), quant_pool_.size() * sizeof(DevQuantTable), 0},
        {&d_huff_work_, huff_work.data(), huff_work.size() * sizeof(HuffWork), 0},
        {&d_chunk_work_, chunk_work.data(), chunk_work.size() * sizeof(ChunkWork), 0},
        {&d_k1_order_, k1_order.data(), k1_order.size() * sizeof(ChunkWork), 16},
        {&d_sub_work_, sub_work.data(), sub_work.size() * sizeof(HuffWork), 0},
        {&d_sub_final_work_, sub_final_work.data(), sub_final_work.size() * sizeof(HuffWork), 0},
        {&d_prog_work_, prog_work.data(), prog_work.size() * sizeof(HuffWork), 0},
        {&d_prog_sync_, nullptr, 0, (size_t)(prog_work.empty() ? 0 : 256)},
        {&d_sub_scan_ids_, sub_scan_ids_.data(), sub_scan_ids_.size() * sizeof(uint32_t), 0},
        {&d_sub_exit_a_, nullptr, 0, (size_t)total_subs_ * sizeof(uint32_t) + 256},
        {&d_sub_exit_b_, nullptr, 0, (size_t)total_subs_ * sizeof(uint32_t) + 256},
        {&d_sub_nblk_, nullptr, 0, (size_t)total_subs_ * sizeof(uint32_t) + 256},
        {&d_sub_first_, nullptr, 0, (size_t)total_subs_ * sizeof(uint32_t) + 256},
        {&d_sub_entry_, nullptr, 0, (size_t)total_subs_ * sizeof(uint32_t) + 256},
        {&d_sub_dcsum_, nullptr, 0, (size_t)total_subs_ * 16 + 256},
        {&d_sub_dcentry_, nullptr, 0, (size_t)total_subs_ * 16 + 256},
        {&d_sub_changed_, nullptr, 0, kSubseqCtlWords * sizeof(uint32_t)},
        {&d_dispose_, dispose_jobs_.data(), dispose_jobs_.size() * sizeof(DisposeJob), 0},
        {&d_sub_same_, nullptr, 0, (size_t)total_subs_ * sizeof(uint32_t) + 256},
        {&d_sub_perm_, nullptr, 0, (size_t)total_subs_ * sizeof(uint32_t) + 256},
        {&d_lut_pool_, nullptr, 0, huff_pool_.size() * kLutPoolBytesPerTable},
        {&d_chunk_sums_, nullptr, 0, (size_t)total_chunks * sizeof(ChunkSum) + 256},
        {&d_k1_tickets_, nullptr, 0, 256},  // [0] the ticket counter, [1] the device's give-up word
        {&d_k2_tickets_, nullptr, 0, 256},
        {&d_sr_set_scan_, sr_set_scan.data(), sr_set_scan.size() * sizeof(uint32_t), 16},
        {&d_sr_luts_, nullptr, 0, sr_set_scan.size() * kSrLutSetBytes + 16},  // K2's pooled runs: a ticket counter each (cleared per upload, never between decodes)
        {&d_idct_work_, idct_work.data(), idct_work.size() * sizeof(IdctWork), 0},
        {&d_idct_work_split_, idct_work_split.data(), idct_work_split.size() * sizeof(IdctWork), 0},
        {&d_ends_, nullptr, 0, (size_t)total_ends_ * sizeof(uint32_t) + 256},
        {&d_ends_u_, nullptr, 0, (size_t)total_ends_ * sizeof(uint32_t) + 256},
        {&d_unstuffed_, nullptr, 0, (size_t)input_bytes_},
        {&d_coefs_, nullptr, 0, entropy_only_ ? 256 : (size_t)total_blocks_ * 128 + (size_t)kIdctBlocksPerWg * 128 + 256},  // + one tile of slack (IDCT DMA reads whole tiles)
        {&d_out_, nullptr, 0, entropy_only_ ? 256 : (size_t)out_bytes_ + 256},
        {&d_rgb_scratch_, nullptr, 0, rgb_convert_.empty() ? 0 : (size_t)out_bytes_ + 256},
        {&d_planes_, nullptr, 0, format_ == JPGPU_FMT_EXTENDED_U16 && !entropy_only_ ? (size_t)planes_bytes_ + 256 : 0},
        {&d_input_, nullptr, 0, (size_t)input_bytes_},
    };
    for (const Up &u : ups) {
        e = u.buf->reserve(std::max(u.bytes, u.reserve));
        if (e != hipSuccess) return hip_fail(e, "hipMalloc");
        if (u.bytes) {
            e = hipMemcpyAsync(u.buf->ptr, u.src, u.bytes, hipMemcpyHostToDevice, up);
            if (e != hipSuccess) return hip_fail(e, "hipMemcpyAsync(descriptors)");
        }
    }
    e = hipMemsetAsync(d_k2_tickets_.ptr, 0, kK2MaxPools * sizeof(uint32_t), up);
    if (e != hipSuccess) return hip_fail(e, "hipMemsetAsync(K2 tickets)");
    {
        // K1 in one pass: descriptors cleared when (re)allocated only (their tags are never reused), tickets + the give-up word per upload
        const size_t want = (size_t)total_chunks * kMarkerDescBytes + 256;
        if (want > d_k1_desc_.cap) {
            e = d_k1_desc_.reserve(want);
            if (e != hipSuccess) return hip_fail(e, "hipMalloc(K1 descriptors)");
            e = hipMemsetAsync(d_k1_desc_.ptr, 0, d_k1_desc_.cap, up);
            if (e != hipSuccess) return hip_fail(e, "hipMemsetAsync(K1 descriptors)");
        }
        e = hipMemsetAsync(d_k1_tickets_.ptr, 0, 2 * sizeof(uint32_t), up);
        if (e != hipSuccess) return hip_fail(e, "hipMemsetAsync(K1 tickets)");
        if (!h_k1_giveup_) {
            e = hipHostMalloc((void **)&h_k1_giveup_, 64, hipHostMallocMapped);
            if (e != hipSuccess) return hip_fail(e, "hipHostMalloc(K1 give-up word)");
        }
        *h_k1_giveup_ = 0;
        k1_epoch_ = 0;
        static const bool three_pass = getenv("JPGPU_K1_THREE_PASS") != nullptr;  // A/B switch: count + prefix + write kernels
        k1_onepass_ = !three_pass;
    }
    e = hipMemsetAsync(d_sub_changed_.ptr, 0, kSubseqCtlWords * sizeof(uint32_t), up);  // (word kSubseqCtlSameDone: no twins found for this upload yet)
    if (e != hipSuccess) return hip_fail(e, "hipMemsetAsync(K2S control)");
    e = launch_lut_pool(up, (const DevHuffTable *)d_huff_pool_.ptr, (int)huff_pool_.size(), (uint8_t *)d_lut_pool_.ptr);
    if (e != hipSuccess) return hip_fail(e, "lut_pool_kernel");
    e = launch_sr_luts(up, (const DevScan *)d_scans_.ptr, (const uint32_t *)d_sr_set_scan_.ptr, (int)sr_set_scan.size(), (const uint8_t *)d_lut_pool_.ptr,
                       (uint8_t *)d_sr_luts_.ptr);
    if (e != hipSuccess) return hip_fail(e, "sr_lut_build_kernel");
    if (!files_resident_) {
        // single scan jobs / frames handed over by the decoder mirror: small, copied as they are
        // slack before the first file and after the last one is read by the kernels' wide loads: keep it defined
        e = hipMemsetAsync(d_input_.ptr, 0, 256, up);
        if (e != hipSuccess) return hip_fail(e, "hipMemsetAsync");
        for (size_t i = 0; i < images_.size(); i++) {
            const ImagePlan &img = images_[i];
            if (img.status != JPGPU_OK || img.jobs.empty() || !file_ptr[i] || file_len[i] == 0) continue;
            e = hipMemcpyAsync((uint8_t *)d_input_.ptr + img.file_offset, file_ptr[i], file_len[i], hipMemcpyHostToDevice, up);
            if (e != hipSuccess) return hip_fail(e, "hipMemcpyAsync(input)");
            const uint64_t tail = img.file_offset + file_len[i];
            if (align_up(tail, 256) > tail) {
                e = hipMemsetAsync((uint8_t *)d_input_.ptr + tail, 0, (size_t)(align_up(tail, 256) - tail), up);
                if (e != hipSuccess) return hip_fail(e, "hipMemsetAsync(tail)");
            }
        }
        e = hipMemsetAsync((uint8_t *)d_input_.ptr + input_bytes_ - 256, 0, 256, up);
        if (e != hipSuccess) return hip_fail(e, "hipMemsetAsync(slack)");
    }
    e = hipStreamSynchronize(up);  // the caller's buffers may be released after upload returns
    if (e != hipSuccess) return hip_fail(e, "hipStreamSynchronize(upload)");
    work_in_flight_ = false;  // the upload stream waited for this batch's earlier device work
    return JPGPU_OK;
}

}  // namespace jpgpu
