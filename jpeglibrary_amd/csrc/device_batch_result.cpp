// jpeglibrary_amd/csrc/device_batch_result.cpp -- DeviceBatch: per-image results, the partial flush of failed progressive frames (the replay),
// downloads and the coefficient tap.
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

int DeviceBatch::fetch_status() {
    if (status_valid_) return JPGPU_OK;
    int rc = sync();
    if (rc != JPGPU_OK) return rc;
    if (!h_status_.empty()) {
        hipError_t e = hipMemcpy(h_status_.data(), d_status_.ptr, h_status_.size() * sizeof(DevScanStatus), hipMemcpyDeviceToHost);
        if (e != hipSuccess) return hip_fail(e, "hipMemcpy(status)");
    }
    if (prog_pipelined_) {
        // A scan of the pipelined progressive launch ran out of polls waiting for its producers: the launch relies on
        // workgroups starting in list order, which the dispatcher does but HIP does not promise.  The whole step is issued
        // again with the frames' scans level by level (fresh launches, no waiting inside a kernel); the batch stays that way.
        bool timed_out = false;
        for (const DevScanStatus &st : h_status_) timed_out |= st.first_error != kNoError && (st.first_error & 0xFFu) == kDetailSpinTimeout;
        if (timed_out) {
            // re-issue what the caller had asked for -- the entropy stage alone (jpgpu_batch_run_entropy; coefficients a caller
            // uploaded for the IDCT stage are then left alone by the output stage that is NOT run) or the whole pipeline
            prog_pipelined_ = false;
            prog_fallbacks_++;
            if (in_decode_request_) {
                rc = decode();
            } else {
                rc = run_marker_index();
                if (rc == JPGPU_OK) rc = run_huffman();
            }
            if (rc != JPGPU_OK) return rc;
            return fetch_status();
        }
    }
    status_valid_ = true;
    static const bool no_partial_flush = getenv("JPGPU_NO_PARTIAL_FLUSH") != nullptr;
    if (replay_possible_ && !replay_done_ && in_decode_request_ && partial_flush_ && !no_partial_flush) return replay_failed_progressive();
    return JPGPU_OK;
}

// The partial flush.  A progressive file that fails in the reference still reaches the writer: Decode()'s `finally` runs the scan
// decoder's Dispose() (JpegDecoder.cs:545-549) over whatever the store holds at that moment -- the scans before the failing one
// complete, the failing one up to where it threw, the later ones never -- with the component slots as the failing scan's
// InitDecodeComponents left them (ScanDecoder/JpegHuffmanProgressiveScanDecoder.cs:60, 421-470), and only then the exception leaves.
// The batch decodes the scans of a frame side by side, so after a failure its store is not that store.  Once per upload, when
// a frame has failed, the step is issued again for the whole batch with: the scans of every frame one after the other in file
// order; in a failed frame the scans behind the failing one left out and the failing one on the kernel that walks and stores
// coefficient by coefficient like the reference (progressive_scan_kernel); the failed frames' Dispose() taken literally
// (dispose_pass_kernel, slots as of the failing scan).  The status of the images does not change; their output is the partial flush.
int DeviceBatch::replay_failed_progressive() {
    struct Failed {
        size_t image;
        int failing;  // index of the failing scan in file order; = number of scans when the marker walk failed behind all of them
    };
    std::vector<Failed> failed;
    for (size_t ii = 0; ii < images_.size(); ii++) {
        const ImagePlan &img = images_[ii];
        if (img.status != JPGPU_OK || img.jobs.size() < 1 || jobs_[img.jobs[0]].kind != kScanFrameOnly) continue;
        int failing = -1;
        for (size_t k = 1; k < img.jobs.size(); k++)
            if (h_status_[(size_t)img.jobs[k]].first_error != kNoError) {
                failing = (int)k - 1;
                break;
            }
        if (failing < 0 && img.late_status != JPGPU_OK) failing = (int)img.jobs.size() - 1;
        if (failing >= 0) failed.push_back({ii, failing});
    }
    replay_done_ = true;
    if (failed.empty()) return JPGPU_OK;
    // Only the failed frames are issued again: every other image keeps its place in the buffers and what the first pass wrote
    // there (one corrupt file in a batch of 1024 used to repeat the whole batch, scan by scan).  Their statuses are
    // the first pass's; the jobs this function rewrites are kept as they were for restore_after_replay().
    const std::vector<DevScanStatus> first_pass = h_status_;
    std::vector<uint8_t> is_failed(images_.size(), 0);
    for (const Failed &f : failed) is_failed[f.image] = 1;
    for (size_t ii = 0; ii < images_.size(); ii++) images_[ii].replay_skip = !is_failed[ii];
    replay_saved_jobs_.clear();
    for (const Failed &f : failed)
        for (int j : images_[f.image].jobs) replay_saved_jobs_.emplace_back((size_t)j, jobs_[(size_t)j]);
    for (const Failed &f : failed) {
        const ImagePlan &img = images_[f.image];
        const int n_scans = (int)img.jobs.size() - 1;
        ScanJob &frame = jobs_[(size_t)img.jobs[0]];
        // the decoder's component slots as of the failing scan (a slot keeps what the last scan with that many components put there)
        int slot_comp[kMaxScanComponents];
        QuantTable slot_q[kMaxScanComponents];
        for (int i = 0; i < kMaxScanComponents; i++) slot_comp[i] = -1;
        for (int k = 0; k < n_scans; k++) {
            ScanJob &job = jobs_[(size_t)img.jobs[(size_t)k + 1]];
            job.disabled = k > f.failing;
            job.force_lane = k == f.failing;
            if (k == f.failing) job.last_interval = h_status_[(size_t)img.jobs[(size_t)k + 1]].first_error >> 8;  // (the lowest failing interval)
            if (k > f.failing) continue;
            for (int i = 0; i < job.scan_components && i < kMaxScanComponents; i++) {
                slot_comp[i] = job.comp[i].component_index;
                slot_q[i] = job.quant_copy[i];
            }
        }
        frame.dispose_generic = true;
        frame.refuse.clear();
        if (f.failing >= n_scans) continue;  // the marker walk failed behind every recorded scan: the slots are as the walk left them (make_frame_job)
        for (int c = 0; c < kMaxScanComponents; c++) frame.dispose_n[c] = 0;
        for (int i = 0; i < frame.geo.frame.num_components && i < kMaxScanComponents; i++) {
            const int c = slot_comp[i];
            if (c < 0 || c >= kMaxScanComponents || frame.dispose_n[c] >= kMaxScanComponents) continue;
            frame.dispose_q[c][frame.dispose_n[c]++] = slot_q[i];
        }
    }
    prog_by_scan_ = true;
    prog_replays_++;
    std::vector<const uint8_t *> fp(images_.size(), nullptr);
    std::vector<size_t> fl(images_.size(), 0);
    for (size_t ii = 0; ii < images_.size(); ii++) fl[ii] = images_[ii].file_len;
    in_replay_ = true;
    files_resident_ = true;
    int rc = layout_and_upload(fp, fl);
    files_resident_ = false;
    if (rc == JPGPU_OK) rc = decode();
    if (rc == JPGPU_OK) rc = fetch_status();
    in_replay_ = false;
    replay_layout_active_ = true;  // the work lists are the failed frames' alone: the next decode() puts the batch's own back
    for (size_t ii = 0; ii < images_.size(); ii++)
        if (!is_failed[ii])
            for (int j : images_[ii].jobs) h_status_[(size_t)j] = first_pass[(size_t)j];
    return rc;
}

// The batch as it was uploaded: the jobs the replay rewrote as they were, every image with work again, the fast launch modes.
int DeviceBatch::restore_after_replay() {
    for (auto &kv : replay_saved_jobs_) jobs_[kv.first] = kv.second;
    std::vector<const uint8_t *> fp(images_.size(), nullptr);
    std::vector<size_t> fl(images_.size(), 0);
    for (size_t ii = 0; ii < images_.size(); ii++) fl[ii] = images_[ii].file_len;
    files_resident_ = true;
    const int rc = layout_and_upload(fp, fl);  // (resets the replay's flags and the images' replay_skip)
    files_resident_ = false;
    replay_possible_ = rc == JPGPU_OK && !entropy_only_;
    return rc;
}

int DeviceBatch::result(int i, jpgpu_image_result *res) {
    const ImagePlan *img = image(i);
    if (!img || !res) return fail(JPGPU_ERR_ARGUMENT, "jpgpu_batch_result: bad index");
    memset(res, 0, sizeof *res);
    if (img->status != JPGPU_OK) {
        res->status = img->status;
        res->detail = img->detail;
        ctx_->last_error = img->error;
        return JPGPU_OK;
    }
    int rc = fetch_status();
    if (rc != JPGPU_OK) return rc;
    res->status = JPGPU_OK;
    res->error_block = 0xFFFFFFFFu;
    bool swallowed = false;
    for (int j : img->jobs) {
        const DevScanStatus &st = h_status_[j];
        res->decoded_mcus = st.decoded_mcus;
        if (jobs_[j].kind == kScanSequential && st.first_error != kNoError && st.pad[1] != 0) res->error_block = kFailBlockBase - st.pad[1];
        res->terminator = st.terminator;
        res->bytes_consumed = st.end_pos;
        if (st.first_error != kNoError && getenv("JPGPU_DEBUG_STATUS"))
            fprintf(stderr, "[jpgpu] image %d job %d kind %d Ss %d Se %d Ah %d Al %d comps %d: first_error %08x decoded %u of %u end_pos %u fail_block %u (bpm %u) shadow %02x\n", i, j,
                    (int)jobs_[j].kind, jobs_[j].ss, jobs_[j].se, jobs_[j].ah, jobs_[j].al, jobs_[j].scan_components, st.first_error, st.decoded_mcus,
                    h_scans_[j].total_mcus, st.end_pos, st.pad[1] ? kFailBlockBase - st.pad[1] : 0xFFFFFFFFu, (unsigned)h_scans_[j].blocks_per_mcu, (unsigned)h_scans_[j].shadow_mask);
        if (st.first_error != kNoError) {
            const uint32_t detail = st.first_error & 0xFF;
            res->detail = (int32_t)detail;
            res->error_interval = st.first_error >> 8;
            // exception class thrown by the reference for each failure
            res->status = (detail == kDetailExpectRestart || detail == kDetailNullTable) ? JPGPU_ERR_INVALID_OPERATION : JPGPU_ERR_INVALID_DATA;
            break;
        }
        if (st.decoded_mcus < h_scans_[j].total_mcus) res->detail = kDetailEarlyEoi;
        if (!entropy_only_ && jobs_[j].kind == kScanSequential && st.terminator != 0 && (st.terminator & 0xF8u) != 0xD0u && (st.pad[2] >> 3) == 1 && st.decoded_mcus >= h_scans_[j].total_mcus) {
            // one whole byte left behind the last block: the reference resumes its walk inside the terminating marker
            if (jobs_[j].forced_swallow) continue;  // (this batch IS the re-plan: the walk behind the scan started there)
            if (j != img->swallow_job) {
                // a middle scan: which scans exist behind it depends on this.  A batch of whole files plans the file again; the
                // per-scan entries (one job, no file: the CALLER walks the markers) still refuse
                if (whole_files_ && img->file_len != 0) return redo_swallowed(i, j, res);
                // (one scan handed over by a caller that walks the markers itself -- jpgpu_decode_scan, the decoder mirror: the reader
                // advance says it, one byte into the terminating marker, and the caller's walk goes on from there like the reference's)
                res->bytes_consumed = st.end_pos + 1;
                continue;
            }
            swallowed = true;
        }
    }
    if (res->status == JPGPU_OK && swallowed) {
        res->status = img->swallow_status;
        if (res->status != JPGPU_OK) res->detail = img->swallow_detail;
        ctx_->last_error = img->swallow_error;
        return JPGPU_OK;
    }
    if (res->status == JPGPU_OK && img->late_status != JPGPU_OK) {
        res->status = img->late_status;
        res->detail = img->late_detail;
        ctx_->last_error = img->late_error;
        return JPGPU_OK;
    }
    if (res->status == JPGPU_OK && !defer_refusal_) {
        for (int j : img->jobs)
            if (!jobs_[j].refuse.empty()) {
                res->status = JPGPU_ERR_NOT_SUPPORTED;
                res->detail = kDetailUnsupportedFrame;
                ctx_->last_error = jobs_[j].refuse;
            }
    }
    return JPGPU_OK;
}

// A middle sequential scan of image i (scan job `job`) left one whole byte unread in front of its terminating marker: the reference's
// reader resumes ONE BYTE INTO that marker (ScanDecoder/JpegHuffmanBaselineScanDecoder.cs:167-176: the marker sits in the bit reader,
// TryPeekMarker() shows it only when the buffer is empty, the two bytes are not given back), typically steps over the next scan's
// header and entropy data as fill, and goes on with whatever marker follows.  The plan this batch was laid out from does not hold
// then.  The file comes back from HBM, is planned again with the scan marked (PlanHandler: the walk continues where the reference's
// does) and decoded by a batch of its own -- which may find the same of a later scan and do the same --; its output replaces the
// image's, its result is the image's.  Rare (a corrupted multi-scan baseline file), and decided once per image.
int DeviceBatch::redo_swallowed(int i, int job, jpgpu_image_result *res) {
    auto hit = redo_.find(i);
    if (hit == redo_.end()) {
        const ImagePlan &img = images_[(size_t)i];
        if (forced_swallow_.size() >= 64) {  // (a re-plan of a re-plan of ...: one level per scan of the file; bounded all the same)
            res->status = JPGPU_ERR_NOT_SUPPORTED;
            res->detail = kDetailUnsupportedFrame;
            ctx_->last_error = "More than 64 scans of one file leave a byte unread in front of their terminating markers.";
            return JPGPU_OK;
        }
        int ordinal = 0;
        for (int j : img.jobs) {
            if (j == job) break;
            ordinal += jobs_[j].kind == kScanSequential ? 1 : 0;
        }
        std::vector<uint8_t> file(img.file_len);
        hipError_t e = hipMemcpy(file.data(), (const uint8_t *)d_input_.ptr + img.file_offset, img.file_len, hipMemcpyDeviceToHost);
        if (e != hipSuccess) return hip_fail(e, "hipMemcpy(file of a re-planned image)");
        Redo r;
        memset(&r.res, 0, sizeof r.res);
        {
            DeviceBatch sub(ctx_);
            std::vector<int> forced = forced_swallow_;
            forced.push_back(ordinal);
            sub.set_forced_swallow(forced);
            const uint8_t *fptr = file.data();
            const size_t flen = file.size();
            int rc = sub.upload_files(&fptr, &flen, 1, format_);
            if (rc == JPGPU_OK) rc = sub.decode();
            if (rc == JPGPU_OK) rc = sub.result(0, &r.res);
            if (rc != JPGPU_OK) return rc;  // (ctx_->last_error says what)
            r.error = ctx_->last_error;
            const ImagePlan *si = sub.image(0);
            if (si && si->status == JPGPU_OK && si->out_bytes == img.out_bytes && img.out_bytes != 0) {
                e = hipMemcpy((uint8_t *)d_out_.ptr + img.out_offset, (const uint8_t *)sub.d_out_.ptr + si->out_offset, img.out_bytes, hipMemcpyDeviceToDevice);
                if (e != hipSuccess) return hip_fail(e, "hipMemcpy(output of a re-planned image)");
            } else if (si && si->status != JPGPU_OK) {  // the re-plan's own host-side verdict (a walk failure in front of any scan cannot happen: scan 0 was planned)
                r.res.status = si->status;
                r.res.detail = si->detail;
                r.error = si->error;
            }
        }
        hit = redo_.emplace(i, std::move(r)).first;
    }
    *res = hit->second.res;
    ctx_->last_error = hit->second.error;
    return JPGPU_OK;
}

int DeviceBatch::download_output(int i, void *dst, size_t cap) {
    const ImagePlan *img = image(i);
    if (!img || !dst) return fail(JPGPU_ERR_ARGUMENT, "jpgpu_batch_download_output: bad argument");
    if (img->status != JPGPU_OK) return fail(img->status, img->error);
    if (cap < img->out_bytes) return fail(JPGPU_ERR_ARGUMENT, "Destination buffer is too small.");
    if (img->jobs.size() > 1 && !entropy_only_ && redo_.find(i) == redo_.end()) {
        // (several scans: the image's result decides whether its plan held -- redo_swallowed -- whatever the caller asks for first)
        jpgpu_image_result tmp;
        const int rr = result(i, &tmp);
        if (rr != JPGPU_OK) return rr;
    }
    // (a batch with a failed progressive frame owes its caller the partial flush whatever is asked for first: ADVICE r4)
    int rc = (replay_possible_ && !replay_done_) ? fetch_status() : sync();
    if (rc != JPGPU_OK) return rc;
    hipError_t e = hipMemcpy(dst, (const uint8_t *)d_out_.ptr + img->out_offset, img->out_bytes, hipMemcpyDeviceToHost);
    return e == hipSuccess ? JPGPU_OK : hip_fail(e, "hipMemcpy(output)");
}

int DeviceBatch::download_coefficients(int i, int16_t *dst, size_t cap_blocks) {
    const ImagePlan *img = image(i);
    if (!img || !dst) return fail(JPGPU_ERR_ARGUMENT, "jpgpu_batch_download_coefficients: bad argument");
    if (img->status != JPGPU_OK) return fail(img->status, img->error);
    if (cap_blocks < img->total_blocks) return fail(JPGPU_ERR_ARGUMENT, "Destination buffer is too small.");
    int rc = (replay_possible_ && !replay_done_) ? fetch_status() : sync();
    if (rc != JPGPU_OK) return rc;
    hipError_t e = hipMemcpy(dst, (const int16_t *)d_coefs_.ptr + img->coef_offset * 64, img->total_blocks * 128, hipMemcpyDeviceToHost);
    return e == hipSuccess ? JPGPU_OK : hip_fail(e, "hipMemcpy(coefficients)");
}

int DeviceBatch::upload_coefficients(int i, const int16_t *src, size_t nblocks) {
    const ImagePlan *img = image(i);
    if (!img || !src) return fail(JPGPU_ERR_ARGUMENT, "jpgpu_batch_upload_coefficients: bad argument");
    if (img->status != JPGPU_OK) return fail(img->status, img->error);
    if (nblocks != img->total_blocks) return fail(JPGPU_ERR_ARGUMENT, "Block count does not match the image.");
    dispose_done_ = false;
    int rc = sync();
    if (rc != JPGPU_OK) return rc;
    hipError_t e = hipMemcpy((int16_t *)d_coefs_.ptr + img->coef_offset * 64, src, nblocks * 128, hipMemcpyHostToDevice);
    return e == hipSuccess ? JPGPU_OK : hip_fail(e, "hipMemcpy(coefficients)");
}

void DeviceBatch::totals(uint64_t *compressed, uint64_t *blocks, uint64_t *pixels, uint64_t *out_bytes) const {
    if (compressed) *compressed = compressed_bytes_;
    if (blocks) *blocks = total_blocks_;
    if (pixels) *pixels = total_pixels_;
    if (out_bytes) {
        uint64_t s = 0;
        for (const ImagePlan &img : images_)
            if (img.status == JPGPU_OK && !img.jobs.empty()) s += img.out_bytes;
        *out_bytes = s;
    }
}

}  // namespace jpgpu
