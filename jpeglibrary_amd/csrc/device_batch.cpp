// jpeglibrary_amd/csrc/device_batch.cpp -- a batch of scan jobs: the host-side plans (header-only / full marker walks), the upload
// paths and the staging ring.  (Round 6 split the file: device_batch_layout.cpp = layout_and_upload, device_batch_launch.cpp = the
// launches of a decode, device_batch_result.cpp = results, the partial-flush replay, downloads.)
//
// HBM layout (all offsets 256-byte aligned unless noted):
//   input   : the files' bytes back to back (256-byte slots, 256 bytes of slack before the first and after the last)
//   ends    : uint32 per restart interval: offset of the FF that closes it (written by K1, read by K2)
//   coefs   : int16[total_blocks][64], zig-zag order, blocks in MCU scan order per scan job (K2 -> K3)
//   out     : per image, in the batch's format (INTERLEAVED_U8: W*H*C bytes; PLANAR_*: padded planes)
#include "device_batch.h"

#include <hip/hip_runtime.h>
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <thread>

#include "host_pool.h"
#include "kernels.h"

namespace jpgpu {

static inline uint64_t align_up(uint64_t v, uint64_t a) { return (v + a - 1) / a * a; }

hipError_t DevBuffer::reserve(size_t bytes) {
    if (bytes <= cap && ptr) return hipSuccess;
    if (ptr) {
        hipError_t e = hipFree(ptr);
        ptr = nullptr;
        cap = 0;
        if (e != hipSuccess) return e;
    }
    if (bytes == 0) bytes = 256;
    hipError_t e = hipMalloc(&ptr, bytes);
    if (e == hipSuccess) cap = bytes;
    return e;
}
void DevBuffer::release() {
    if (ptr) (void)hipFree(ptr);
    ptr = nullptr;
    cap = 0;
}

DeviceBatch::~DeviceBatch() {
    if (ctx_) (void)hipSetDevice(ctx_->device);
    for (DevBuffer *b : {&d_sub_work_, &d_sub_final_work_, &d_sub_scan_ids_, &d_sub_exit_a_, &d_sub_exit_b_, &d_sub_nblk_, &d_sub_first_, &d_sub_entry_, &d_sub_dcsum_, &d_sub_dcentry_, &d_sub_changed_, &d_sub_same_, &d_sub_perm_, &d_k1_desc_, &d_k1_tickets_, &d_k2_tickets_, &d_sr_luts_, &d_sr_set_scan_, &d_k1_order_, &d_prog_snapshot_, &d_dispose_, &d_verify_, &d_lut_pool_, &d_prog_work_, &d_prog_sync_, &d_planes_, &d_extend_desc_, &d_gather_, &d_rgb_scratch_, &d_chunk_work_, &d_chunk_sums_, &d_unstuffed_, &d_ends_u_, &d_input_, &d_scans_, &d_status_, &d_ends_, &d_huff_pool_, &d_quant_pool_, &d_huff_work_, &d_idct_work_, &d_idct_work_split_, &d_coefs_, &d_out_})
        b->release();
    for (hipEvent_t &e : ev_pool_)
        if (e) (void)hipEventDestroy(e);
    if (done_ev_) (void)hipEventDestroy(done_ev_);
    if (h_k1_giveup_) (void)hipHostFree(h_k1_giveup_);
}

int DeviceBatch::fail(int status, const std::string &msg) {
    ctx_->last_error = msg;
    return status;
}
int DeviceBatch::hip_fail(hipError_t e, const char *what) {
    return fail(e == hipErrorOutOfMemory ? JPGPU_ERR_OUT_OF_MEMORY : JPGPU_ERR_DEVICE, std::string(what) + ": " + hipGetErrorString(e));
}

// Planner: what JpegScanDecoder.Create / ProcessScan become while the batch is being laid out.
namespace {

class PlanHandler final : public ScanHandler {
  public:
    explicit PlanHandler(std::vector<ScanJob> *jobs, bool first_scan_only = false, const std::vector<int> *forced = nullptr)
        : jobs_(jobs), first_scan_only_(first_scan_only), forced_(forced) {}
    // Second walk of a file whose sequential scans are already planned (`ends` = where each one's data stops): nothing is
    // recorded, and scan number `swallow` leaves the reader ONE byte into its terminating marker.  That is where the
    // reference's reader stands when exactly one whole byte was left in the bit reader behind the last block: the marker
    // has been pulled into the bit reader, TryPeekMarker() only shows it once the buffer is empty, so the two bytes are
    // not given back (ScanDecoder/JpegHuffmanBaselineScanDecoder.cs:167-176, JpegBitReader.cs:152-155).
    PlanHandler(const std::vector<size_t> *ends, int swallow) : jobs_(nullptr), replay_ends_(ends), swallow_(swallow) {}
    const std::vector<size_t> &sequential_ends() const { return ends_; }
    void on_frame(HostDecoder &dec, int sof) override {
        sof_ = sof;
        baseline_ = false;
        flush_progressive();  // a second SOF replaces the scan decoder: the old one is disposed first (JpegDecoder.cs:568)
        if (sof == kSOF0 || sof == kSOF1) {
            geo_ = BaselineGeometry::latch(dec, dec.frame_header());  // DRI latched at SOF time
            baseline_ = true;
        } else if (sof == kSOF2) {
            prog_.begin(dec, dec.frame_header());
        }
    }
    void on_scan(HostDecoder &dec, MarkerReader &reader, const ScanHeader &scan) override {
        const uint8_t *entropy = reader.remaining_bytes();
        const size_t len = (size_t)reader.remaining_byte_count();
        if (prog_.active()) {
            prog_.add_scan(dec, scan, entropy, len);  // the reference leaves the outer reader where it is
            return;
        }
        if (!baseline_)
            throw DecodeError(JPGPU_ERR_NOT_SUPPORTED, "Only Huffman DCT frames (SOF0, SOF1, SOF2) run on this path.", kDetailUnsupportedFrame);
        if (scan.num_components == 0) {
            // A scan header that names no component: ProcessScan walks the MCUs without reading a bit (:99-136).  With a
            // restart interval the first restart check finds the bit buffer full and no marker (:139-154); without one
            // the reader is left where it is and the outer walk skips the entropy data as fill.
            const uint64_t mcus = (uint64_t)geo_.mcus_per_line * (uint64_t)geo_.mcus_per_column;
            if (geo_.restart_interval != 0 && mcus >= geo_.restart_interval && len != 0 &&
                !(len >= 2 && entropy[0] == 0xFF && entropy[1] != 0x00 && entropy[1] != 0xFF))
                throw DecodeError(JPGPU_ERR_INVALID_OPERATION, "Expect restart marker.", kDetailExpectRestart);
            if (first_scan_only_)
                throw DecodeError(JPGPU_ERR_NOT_SUPPORTED, "A scan without components is not supported by the optimizer path.", kDetailUnsupportedFrame);
            if (geo_.restart_interval != 0 && mcus >= geo_.restart_interval)
                throw DecodeError(JPGPU_ERR_NOT_SUPPORTED, "A scan without components in front of restart markers is not supported.", kDetailUnsupportedFrame);
            reader.try_advance((int)find_scan_end(entropy, len));
            return;
        }
        if (replay_ends_) {
            const int k = replayed_++;
            if (k > swallow_ || k >= (int)replay_ends_->size())
                throw DecodeError(JPGPU_ERR_NOT_SUPPORTED, "A scan behind a scan that left one byte unread is not supported.", kDetailUnsupportedFrame);
            reader.try_advance((int)(*replay_ends_)[k] + (k == swallow_ ? 1 : 0));
            return;
        }
        jobs_->push_back(make_scan_job(dec, geo_, scan, entropy, len, first_scan_only_));
        if (!first_scan_only_) {
            // (a scan KNOWN to leave one whole byte unread -- the device said so in the batch this plan is made for -- hands the
            // reader back one byte into its terminating marker, the way the reference's does: DeviceBatch::redo_swallowed)
            const bool forced = forced_ != nullptr && std::find(forced_->begin(), forced_->end(), (int)ends_.size()) != forced_->end();
            ends_.push_back(find_scan_end(entropy, len));
            jobs_->back().forced_swallow = forced && ends_.back() < len;
            reader.try_advance((int)ends_.back() + (jobs_->back().forced_swallow ? 1 : 0));
            return;
        }
        // leave the reader just before the next non-RST marker, like ProcessScan does (:167-176); the optimizer path only
        // wants the scan resolved (what follows it is its own marker walk's business): nothing is left to read
        reader.try_advance(first_scan_only_ ? (int)len : (int)find_scan_end(entropy, len));
    }
    void on_dispose(HostDecoder &) override { flush_progressive(); }
    const BaselineGeometry &geo() const { return prog_geo_valid_ ? prog_geo_ : geo_; }
    int sof() const { return sof_; }

  private:
    // Dispose() of the progressive scan decoder: the frame's IDCT pass, then its entropy scans in file order
    void flush_progressive() {
        if (!prog_.active()) return;
        if (jobs_) {  // (also without a single recorded scan: Dispose() still flushes the allocator's blocks)
            jobs_->push_back(prog_.make_frame_job());
            for (ScanJob &j : prog_.scans()) jobs_->push_back(std::move(j));
            prog_geo_ = prog_.geo();
            prog_geo_valid_ = true;
        }
        prog_.reset();
    }
    std::vector<ScanJob> *jobs_;
    std::vector<size_t> ends_;
    const std::vector<size_t> *replay_ends_ = nullptr;
    int swallow_ = -1, replayed_ = 0;
    bool first_scan_only_ = false;
    const std::vector<int> *forced_ = nullptr;  // sequential scans (by ordinal) known to leave one byte unread: DeviceBatch::redo_swallowed
    BaselineGeometry geo_, prog_geo_;
    bool prog_geo_valid_ = false;
    ProgressiveFrame prog_;
    bool baseline_ = false;
    int sof_ = 0;
};
}  // namespace

void DeviceBatch::plan_image_geometry(ImagePlan &img, const BaselineGeometry &geo) const {
    const FrameHeader &fh = geo.frame;
    img.width = fh.samples_per_line;
    img.height = fh.lines;
    img.precision = fh.precision;
    img.num_components = fh.num_components;
    img.restart_interval = geo.restart_interval;
    img.mcus_per_line = (uint32_t)geo.mcus_per_line;
    img.mcus_per_column = (uint32_t)geo.mcus_per_column;
    if (format_ == JPGPU_FMT_INTERLEAVED_U8) {
        img.out_bytes = (uint64_t)img.width * img.height * img.num_components;
    } else if (format_ == JPGPU_FMT_RGB_U8 || format_ == JPGPU_FMT_RGBA_U8) {
        if (fh.num_components != 1 && fh.num_components != 3)  // apps/JpegDecode/DecodeAction.cs:29-33
            throw DecodeError(JPGPU_ERR_NOT_SUPPORTED, "This color space is not supported", kDetailUnsupportedFrame);
        if (fh.precision != 8)
            throw DecodeError(JPGPU_ERR_NOT_SUPPORTED, "RGB output is defined for 8-bit precision only (the reference converter assumes 8-bit samples).",
                              kDetailUnsupportedFrame);
        img.out_bytes = (uint64_t)img.width * img.height * (format_ == JPGPU_FMT_RGBA_U8 ? 4 : 3);
    } else {
        if (fh.num_components > 4)  // jpgpu_plane_info describes four planes; a fifth component would land on plane 0
            throw DecodeError(JPGPU_ERR_NOT_SUPPORTED, "The planar output formats describe at most 4 components.", kDetailUnsupportedFrame);
        // EXTENDED_U16 is produced from int16 planes (K3's PLANAR_I16 output in a scratch buffer, see run_idct)
        const uint64_t sample_bytes = (format_ == JPGPU_FMT_PLANAR_I16 || format_ == JPGPU_FMT_EXTENDED_U16) ? 2 : 1;
        uint64_t off = 0;
        for (int c = 0; c < fh.num_components && c < 4; c++) {
            jpgpu_plane_info &p = img.plane[c];
            p.width = (uint32_t)geo.mcus_per_line * fh.components[c].h * 8;
            p.height = (uint32_t)geo.mcus_per_column * fh.components[c].v * 8;
            p.pitch = p.width;
            p.offset = off;
            off = align_up(off + (uint64_t)p.pitch * p.height * sample_bytes, 256);
        }
        img.out_bytes = off;
        if (format_ == JPGPU_FMT_EXTENDED_U16) {
            img.planes_bytes = off;
            img.out_bytes = (uint64_t)img.width * img.height * 4 * sizeof(uint16_t);
        }
    }
    // A frame header is two 16-bit sizes and a component count: a corrupted one can ask for hundreds of gigabytes (the
    // reference's caller would fail allocating the writer's buffer).  Such an image fails BY ITSELF instead of taking the
    // batch's allocation, and every other image, with it (tools/stress_parity.py STRESS_HEADER=1, seed 460).
    if (ctx_ && ctx_->device_bytes != 0 && img.out_bytes + img.planes_bytes > ctx_->device_bytes)
        throw DecodeError(JPGPU_ERR_OUT_OF_MEMORY, "The frame's output (" + std::to_string(img.out_bytes + img.planes_bytes) +
                                                        " bytes) is larger than the device's memory.", kDetailUnsupportedFrame);
}

// ---------------------------------------------------------------------------------------------------------------- ingest
//
// jpgpu_batch_upload = SetInput + Identify + Decode's marker loop for n files (ref: JpegDecoder.cs:75-162, 509-617), without
// the host ever walking entropy-coded bytes in the common case:
//   1. header-only plan, one file per crew thread: Identify's walk up to the first SOS header, then Decode's walk up to the
//      same point; the file is planned as "this one sequential scan, its data closed by EOI" (FastPlanHandler);
//   2. the files go to HBM through the context's pinned staging ring: the crew copies the caller's bytes into 32 MiB slots,
//      every full slot leaves as one DMA on the upload stream while the next ones are being filled;
//   3. the device reads the bytes behind each SOS header once (first_marker_kernel) and reports the first marker that is
//      not RSTn: where that is EOI, Identify and Decode would have seen nothing else either (neither looks behind EOI) and
//      the plan stands -- including Identify's "last DRI in the file" (every DRI lay in front of the SOS);
//   4. every other file (several scans, progressive, tables or garbage behind the scan, truncated data, a Decode-walk
//      failure that a later Identify failure would pre-empt) takes the full walk of both loops, also on the crew.

namespace {
struct NeedFullWalk {};  // the header-only planner met something that is not "headers, one sequential scan"

class FastPlanHandler final : public ScanHandler {
  public:
    explicit FastPlanHandler(std::vector<ScanJob> *jobs) : jobs_(jobs) {}
    void on_frame(HostDecoder &dec, int sof) override {
        if (sof != kSOF0 && sof != kSOF1) throw NeedFullWalk{};
        geo_ = BaselineGeometry::latch(dec, dec.frame_header());  // DRI latched at SOF time
    }
    void on_scan(HostDecoder &dec, MarkerReader &reader, const ScanHeader &scan) override {
        if (scan.num_components == 0) throw NeedFullWalk{};
        const uint8_t *entropy = reader.remaining_bytes();
        const size_t len = (size_t)reader.remaining_byte_count();
        jobs_->push_back(make_scan_job(dec, geo_, scan, entropy, len, false));
        reader.try_advance((int)len);  // the plan: nothing but this scan's data and an EOI follow (checked on the device)
    }
    void on_dispose(HostDecoder &) override {}

  private:
    std::vector<ScanJob> *jobs_;
    BaselineGeometry geo_;
};
}  // namespace

struct DeviceBatch::FilePlan {
    ImagePlan img;
    std::vector<ScanJob> jobs;
    std::vector<size_t> seq_ends;  // where each sequential scan's data stops (offset from the scan's first entropy byte)
    bool speculative = false;      // header-only plan, waiting for the device's verdict
    bool need_full = false;
    size_t scan_data_pos = 0;      // offset of the first entropy byte in the file (speculative plans)
};

// Identify + Decode's marker loop over the whole file (both walk the entropy bytes): the general path.
void DeviceBatch::plan_file_full(const uint8_t *file, size_t len, int index, FilePlan &fp) const {
    fp.jobs.clear();
    fp.seq_ends.clear();
    fp.speculative = false;
    fp.img = ImagePlan();
    ImagePlan &img = fp.img;
    img.file_len = len;
    bool decoding = false;  // Identify() is over, Decode()'s marker loop is running
    HostDecoder dec;
    PlanHandler handler(&fp.jobs, entropy_only_, forced_swallow_.empty() ? nullptr : &forced_swallow_);
    try {
        if (len > 0x7FFFFFF0u) throw DecodeError(JPGPU_ERR_NOT_SUPPORTED, "JPEG streams of 2 GiB or more are not supported.");
        dec.set_input(file, len);
        if (entropy_only_) {
            // optimizer path: JpegOptimizer.Scan() runs no Identify(); the restart interval is the one in force at the
            // scan (OptimizeBatch::plan_file found it) unless a DRI segment in front of the frame header says otherwise
            if ((size_t)index < preset_dri_.size()) dec.set_restart_interval(preset_dri_[index]);
        } else {
            dec.identify(false);  // every reference caller runs Identify before Decode; it latches the LAST DRI (F4)
        }
        img.sof = (uint8_t)dec.start_of_frame();
        decoding = true;
        try {
            dec.decode(handler, true);
        } catch (...) {
            fp.seq_ends = handler.sequential_ends();
            throw;
        }
        decoding = false;
        if (entropy_only_) img.sof = (uint8_t)dec.start_of_frame();
        if (fp.jobs.empty()) {
            // no scan: Decode() succeeds without writing anything; keep the frame geometry for the caller
            if (img.sof == kSOF0 || img.sof == kSOF1 || img.sof == kSOF2) plan_image_geometry(img, BaselineGeometry::latch(dec, dec.frame_header()));
        } else {
            plan_image_geometry(img, fp.jobs[0].geo);
            img.blocks_per_mcu = (uint32_t)fp.jobs[0].blocks_per_mcu;
        }
        fp.seq_ends = handler.sequential_ends();
    } catch (const DecodeError &e) {
        // the scans of a progressive frame recorded before the walk failed ran in the reference too (each ProcessScan
        // decodes its scan on the spot): they are kept so that their own failures come first
        if (decoding && e.status != JPGPU_ERR_NOT_SUPPORTED) {
            try {
                handler.on_dispose(dec);
            } catch (const DecodeError &) {
            }
        }
        const bool keep = decoding && !fp.jobs.empty() && e.status != JPGPU_ERR_NOT_SUPPORTED;
        if (keep) {
            // scans handed to the scan decoder before the walk failed: they run, the failure is reported behind them
            img.late_status = e.status;
            img.late_detail = e.detail;
            img.late_error = e.what();
            try {
                plan_image_geometry(img, fp.jobs[0].geo);
                img.blocks_per_mcu = (uint32_t)fp.jobs[0].blocks_per_mcu;
            } catch (const DecodeError &e2) {
                fp.jobs.clear();
                img.status = e2.status;
                img.detail = e2.detail;
                img.error = e2.what();
            }
        } else {
            fp.jobs.clear();
            img.status = e.status;
            img.detail = e.detail;
            img.error = e.what();
        }
    }
    if (!entropy_only_ && img.status == JPGPU_OK) plan_swallowed_terminator(fp, file, len, false);
}

// Headers only: both marker loops up to the first SOS header, the scan planned as the file's only one.
void DeviceBatch::plan_file_headers(const uint8_t *file, size_t len, FilePlan &fp) const {
    fp.jobs.clear();
    fp.seq_ends.clear();
    fp.speculative = fp.need_full = false;
    fp.img = ImagePlan();
    ImagePlan &img = fp.img;
    img.file_len = len;
    if (entropy_only_ || len > 0x7FFFFFF0u || !forced_swallow_.empty()) {
        fp.need_full = true;  // optimizer walks have rules of their own; oversize files are refused by the full path; a re-plan walks the file
        return;
    }
    HostDecoder dec;
    try {
        dec.set_input(file, len);
        if (!dec.identify_until_scan(false, &fp.scan_data_pos)) {
            fp.need_full = true;  // no scan in the file: the walk just done WAS the whole Identify; let the general path plan it
            return;
        }
        if (!dec.has_frame_header()) {
            fp.need_full = true;  // SOS in front of any SOF: Identify's verdict depends on what follows the scan
            return;
        }
    } catch (const DecodeError &e) {
        // Identify fails in front of the first scan: that is what the caller sees, whatever follows
        img.status = e.status;
        img.detail = e.detail;
        img.error = e.what();
        return;
    }
    img.sof = (uint8_t)dec.start_of_frame();
    try {
        FastPlanHandler handler(&fp.jobs);
        dec.decode(handler, true);
        if (fp.jobs.size() != 1) throw NeedFullWalk{};
        plan_image_geometry(img, fp.jobs[0].geo);
        img.blocks_per_mcu = (uint32_t)fp.jobs[0].blocks_per_mcu;
        fp.speculative = true;
    } catch (const NeedFullWalk &) {
        fp.need_full = true;
    } catch (const DecodeError &) {
        // Decode's loop fails before the scan is planned -- but Identify walks the WHOLE file first, and a failure of
        // its own behind the scan would be the one the caller sees: only the full walk can tell
        fp.need_full = true;
    }
    if (fp.need_full) fp.jobs.clear();
}

// What Decode() ends in when the LAST sequential scan of the file leaves its reader one byte into the terminating marker.
// identify_is_clean: the header-only path already knows that Identify() succeeds (and what it latched lies in front of
// the first SOS): its walk over the entropy data is not repeated.
void DeviceBatch::plan_swallowed_terminator(FilePlan &fp, const uint8_t *file, size_t len, bool identify_is_clean) const {
    ImagePlan &img = fp.img;
    img.swallow_status = JPGPU_OK;
    img.swallow_detail = 0;
    img.swallow_error.clear();
    img.swallow_job = -1;
    const std::vector<size_t> &ends = fp.seq_ends;
    if (ends.empty() || fp.jobs.empty()) return;
    int last = -1, n_seq = 0;
    for (size_t j = 0; j < fp.jobs.size(); j++)
        if (fp.jobs[j].kind == kScanSequential) {
            last = (int)j;
            n_seq++;
        }
    if (last < 0 || n_seq != (int)ends.size() || ends.back() >= fp.jobs[last].entropy_len) return;  // no marker behind it
    img.swallow_job = last;  // index into fp.jobs; upload_files turns it into a batch job index
    if (identify_is_clean && n_seq == 1 && (size_t)(fp.jobs[last].entropy - file) + ends.back() + 2 == len) {
        // The EOI closes the file (every clean file): the replayed walk would step over the scan to one byte into the
        // marker, find a single byte left and fail in TryReadMarker (JpegDecoder.cs:533-537).  Written down directly: a
        // thousand exceptions thrown from a crew of threads serialise on the unwinder's lock.
        img.swallow_status = JPGPU_ERR_INVALID_DATA;
        img.swallow_detail = kDetailBadHeader;
        img.swallow_error = "Failed to decode JPEG data at offset " + std::to_string(len - 1) + ". No marker found.";
        return;
    }
    try {
        HostDecoder dec;
        dec.set_input(file, len);
        if (identify_is_clean) {
            size_t pos;
            (void)dec.identify_until_scan(false, &pos);
        } else {
            dec.identify(false);
        }
        PlanHandler replay(&ends, n_seq - 1);
        dec.decode(replay, true);
    } catch (const DecodeError &e) {
        img.swallow_status = e.status;
        img.swallow_detail = e.detail;
        img.swallow_error = e.what();
    }
}

// Crew size when the caller did not choose one: the CPUs this process may really use -- the affinity mask and the cgroup
// CPU quota (v2 cpu.max, v1 cpu.cfs_quota_us) both bound it (a container often reports the machine's 256 threads and is
// granted 16) -- capped at 16.
int granted_host_cpus() {
    unsigned cpus = std::max(1u, std::thread::hardware_concurrency());
    cpu_set_t set;
    CPU_ZERO(&set);
    if (sched_getaffinity(0, sizeof set, &set) == 0 && CPU_COUNT(&set) > 0) cpus = std::min(cpus, (unsigned)CPU_COUNT(&set));
    bool have_quota = false;
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {  // cgroup v2: "<quota> <period>" or "max <period>"
        char q[32] = {0};
        long period = 0;
        if (fscanf(f, "%31s %ld", q, &period) == 2 && period > 0) {
            have_quota = true;
            if (strcmp(q, "max") != 0) {
                const long quota = atol(q);
                if (quota > 0) cpus = std::min(cpus, (unsigned)std::max(1L, (quota + period - 1) / period));
            }
        }
        fclose(f);
    }
    if (!have_quota) {  // cgroup v1: cpu.cfs_quota_us (-1 = unlimited) / cpu.cfs_period_us
        long quota = -1, period = 0;
        if (FILE *f = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
            if (fscanf(f, "%ld", &quota) != 1) quota = -1;
            fclose(f);
        }
        if (FILE *f = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
            if (fscanf(f, "%ld", &period) != 1) period = 0;
            fclose(f);
        }
        if (quota > 0 && period > 0) cpus = std::min(cpus, (unsigned)std::max(1L, (quota + period - 1) / period));
    }
    return (int)cpus;
}
int default_host_threads() {
    if (const char *ev = getenv("JPGPU_HOST_THREADS")) return std::max(1, atoi(ev));
    return std::min(16, granted_host_cpus());  // a handful of threads already keep the host link busy (profiles/r02_ingest_sweep.jsonl)
}

// One input file as the caller handed it over: a list of segments (one for jpgpu_batch_upload), and the contiguous bytes the
// host parser reads -- the file itself, or what was gathered of a multi-segment file (its head for the header-only plan, all
// of it for the full marker walks).
struct DeviceBatch::FileSegs {
    const jpgpu_segment *seg = nullptr;
    int n = 0;
    size_t len = 0;
    const uint8_t *base = nullptr;
    size_t base_len = 0;
    std::vector<uint8_t> gathered;
    static constexpr size_t kHeadBytes = 64u << 10;
    bool whole() const { return base_len == len; }
    void gather(size_t want) {
        want = std::min(want, len);
        gathered.resize(want);
        size_t pos = 0;
        for (int k = 0; k < n && pos < want; k++) {
            const size_t m = std::min(seg[k].len, want - pos);
            if (m) memcpy(gathered.data() + pos, seg[k].data, m);
            pos += m;
        }
        base = gathered.data();
        base_len = want;
    }
    uint8_t at(size_t off) const {
        for (int k = 0; k < n; k++) {
            if (off < seg[k].len) return seg[k].data[off];
            off -= seg[k].len;
        }
        return 0;
    }
};

int DeviceBatch::mark_work() {
    if (!done_ev_) {
        hipError_t e = hipEventCreateWithFlags(&done_ev_, hipEventDisableTiming);
        if (e != hipSuccess) return hip_fail(e, "hipEventCreate(done)");
    }
    hipError_t e = hipEventRecord(done_ev_, ctx_->stream);
    if (e != hipSuccess) return hip_fail(e, "hipEventRecord(done)");
    work_in_flight_ = true;
    return JPGPU_OK;
}

// An upload rewrites the batch's inputs, descriptors and work lists on the upload stream; kernels this batch launched on the
// decode stream and nobody waited for may still be reading them.  The upload stream waits for them on the device (the host
// does not block, another batch's decode is not waited for).
int DeviceBatch::order_upload_behind_work() {
    if (!work_in_flight_ || !done_ev_) return JPGPU_OK;
    hipError_t e = hipStreamWaitEvent(ctx_->upload_stream, done_ev_, 0);
    if (e != hipSuccess) return hip_fail(e, "hipStreamWaitEvent(upload behind decode)");
    return JPGPU_OK;
}

int DeviceBatch::upload_files(const uint8_t *const *jpeg, const size_t *len, int n, int format) {
    if (n < 0 || (n > 0 && (!jpeg || !len))) return fail(JPGPU_ERR_ARGUMENT, "jpgpu_batch_upload: null argument");
    std::vector<jpgpu_segment> segs((size_t)n);
    std::vector<int> per((size_t)n, 1);
    for (int i = 0; i < n; i++) segs[(size_t)i] = {jpeg[i], len[i]};
    return upload_segments(segs.data(), per.data(), n, format, 0);
}

int DeviceBatch::upload_segments(const jpgpu_segment *segments, const int *segments_per_file, int n, int format, unsigned flags) {
    if (n < 0 || (n > 0 && (!segments || !segments_per_file))) return fail(JPGPU_ERR_ARGUMENT, "jpgpu_batch_upload: null argument");
    if (format < 0 || format >= kNumOutputFormats) return fail(JPGPU_ERR_ARGUMENT, "jpgpu_batch_upload: unknown format");
    if (flags & ~(JPGPU_UPLOAD_PINNED | JPGPU_UPLOAD_PINNED_ARENA)) return fail(JPGPU_ERR_ARGUMENT, "jpgpu_batch_upload_segments: unknown flag");
    const bool pinned = (flags & (JPGPU_UPLOAD_PINNED | JPGPU_UPLOAD_PINNED_ARENA)) != 0;
    using clk = std::chrono::steady_clock;
    auto ms_since = [](clk::time_point t0) { return (float)std::chrono::duration<double, std::milli>(clk::now() - t0).count(); };
    const clk::time_point t_begin = clk::now();
    ingest_ = IngestStats();
    hipError_t e = hipSetDevice(ctx_->device);
    if (e != hipSuccess) return hip_fail(e, "hipSetDevice");
    int rc = order_upload_behind_work();
    if (rc != JPGPU_OK) return rc;
    format_ = format;
    images_.assign((size_t)n, ImagePlan());
    jobs_.clear();
    job_image_.clear();
    job_entropy_off_.clear();
    constexpr size_t kMaxFile = 0x7FFFFFF0u;
    std::vector<FileSegs> files((size_t)n);
    uint64_t total_bytes = 0;
    {
        const jpgpu_segment *sp = segments;
        for (int i = 0; i < n; i++) {
            FileSegs &f = files[(size_t)i];
            if (segments_per_file[i] < 0) return fail(JPGPU_ERR_ARGUMENT, "jpgpu_batch_upload_segments: negative segment count");
            f.seg = sp;
            f.n = segments_per_file[i];
            sp += f.n;
            for (int k = 0; k < f.n; k++) {
                if (f.seg[k].len && !f.seg[k].data) return fail(JPGPU_ERR_ARGUMENT, "jpgpu_batch_upload: null segment");
                f.len += f.seg[k].len;
            }
            if (f.n == 1) {
                f.base = f.seg[0].data;
                f.base_len = f.len;
            }
            total_bytes += f.len;
        }
    }
    std::vector<FilePlan> plans((size_t)n);

    const int want = ctx_->host_threads > 0 ? ctx_->host_threads : default_host_threads();
    // no more threads than there is work for: one per 8 files or per 2 MiB, whichever asks for more
    const int useful = (int)std::max<uint64_t>((uint64_t)(n + 7) / 8, total_bytes >> 21);
    WorkCrew crew(std::max(1, std::min(want, useful)));
    ingest_.threads = crew.threads();

    // ---- 1. header-only plans (a multi-segment file: over its first 64 KiB, gathered; should its first scan start behind
    //         them, or the plan not be "headers + one sequential scan", the whole file is gathered for the full walks)
    clk::time_point t0 = clk::now();
    std::atomic<int> n_linearised{0};
    crew.run((size_t)n, [&](size_t i, int) {
        FileSegs &f = files[i];
        FilePlan &fp = plans[i];
        if (f.len > kMaxFile) {
            fp.img.file_len = f.len;
            fp.need_full = true;  // refused by the full path before it reads a byte
            return;
        }
        if (f.n > 1) f.gather(FileSegs::kHeadBytes);
        plan_file_headers(f.base, f.base_len, fp);
        if (!f.whole()) {
            if (fp.speculative) {
                fp.jobs[0].entropy_len = f.len - fp.scan_data_pos;
            } else {
                f.gather(f.len);
                n_linearised.fetch_add(1, std::memory_order_relaxed);
                plan_file_headers(f.base, f.base_len, fp);
            }
        }
        fp.img.file_len = f.len;
    });
    ingest_.parse_ms = ms_since(t0);

    // ---- 2. the files -> HBM (every file gets its slot, whatever became of its plan: the layout does not wait for plans)
    t0 = clk::now();
    // One page-locked arena (JPGPU_UPLOAD_PINNED_ARENA): the device copy keeps the arena's own layout -- file i lies where it
    // lies in the arena, relative to the lowest address -- so the whole span travels as a few large DMAs instead of one per
    // file (1 MiB copies reach ~36 GB/s on this link, 32 MiB ones 56).  Needs every file contiguous in memory and a span
    // that is mostly payload; otherwise the files go one DMA per segment.
    arena_span_ = {nullptr, 0};
    if ((flags & JPGPU_UPLOAD_PINNED_ARENA) && n > 0) {
        const uint8_t *lo = nullptr, *hi = nullptr;
        bool contiguous = true;
        for (int i = 0; i < n && contiguous; i++) {
            const FileSegs &f = files[(size_t)i];
            if (f.len == 0) continue;
            if (f.len > kMaxFile) contiguous = false;
            const uint8_t *expect = nullptr;
            for (int k = 0; k < f.n; k++) {
                if (!f.seg[k].len) continue;
                if (expect && f.seg[k].data != expect) contiguous = false;
                if (!lo || f.seg[k].data < lo) lo = f.seg[k].data;
                if (!hi || f.seg[k].data + f.seg[k].len > hi) hi = f.seg[k].data + f.seg[k].len;
                expect = f.seg[k].data + f.seg[k].len;
            }
        }
        if (contiguous && lo && (uint64_t)(hi - lo) <= 2 * total_bytes + (1u << 20)) arena_span_ = {lo, (size_t)(hi - lo)};
    }
    uint64_t in_off = 256;
    if (arena_span_.first) {
        for (int i = 0; i < n; i++) {
            const FileSegs &f = files[(size_t)i];
            const uint8_t *first = nullptr;
            for (int k = 0; k < f.n && !first; k++)
                if (f.seg[k].len) first = f.seg[k].data;
            plans[i].img.file_offset = 256 + (first ? (uint64_t)(first - arena_span_.first) : 0u);
        }
        in_off = align_up(256 + arena_span_.second, 256);
    } else {
        for (int i = 0; i < n; i++) {
            plans[i].img.file_offset = in_off;
            if (files[(size_t)i].len <= kMaxFile) in_off = align_up(in_off + files[(size_t)i].len, 256);
        }
    }
    input_bytes_ = in_off + 256;
    rc = stage_files(crew, files, plans, pinned);
    if (rc != JPGPU_OK) return rc;

    // ---- 3. the device's verdict on the header-only plans
    std::vector<int> spec;
    for (int i = 0; i < n; i++)
        if (plans[i].speculative) spec.push_back(i);
    if (!spec.empty()) {
        std::vector<uint32_t> first;
        rc = verify_plans(plans, spec, first);
        if (rc != JPGPU_OK) return rc;
        for (size_t k = 0; k < spec.size(); k++) {
            FilePlan &fp = plans[spec[k]];
            const FileSegs &f = files[(size_t)spec[k]];
            const size_t dlen = f.len - fp.scan_data_pos;
            const uint32_t pos = first[k];
            // classify16 only calls FF xx a marker when xx exists: pos + 1 < dlen
            if (pos != 0xFFFFFFFFu && (size_t)pos + 1 < dlen && f.at(fp.scan_data_pos + pos) == 0xFF && f.at(fp.scan_data_pos + pos + 1) == kEOI) {
                fp.seq_ends.assign(1, (size_t)pos);
            } else {
                fp.speculative = false;
                fp.need_full = true;
                fp.jobs.clear();
            }
        }
    } else {
        e = hipStreamSynchronize(ctx_->upload_stream);  // the caller's buffers may be released after upload returns
        if (e != hipSuccess) return hip_fail(e, "hipStreamSynchronize(upload)");
    }
    work_in_flight_ = false;  // the upload stream waited for this batch's earlier device work, and has been drained
    ingest_.copy_ms = ms_since(t0);

    // ---- 4. the rest: "one byte into the terminator" verdicts of the confirmed plans, full walks of everything else
    t0 = clk::now();
    crew.run((size_t)n, [&](size_t i, int) {
        FilePlan &fp = plans[i];
        FileSegs &f = files[i];
        const uint64_t off = fp.img.file_offset;
        // a multi-segment file planned from its head: the direct verdict needs nothing but offsets when the EOI closes the
        // file; anything else replays the walk over the whole file, i.e. takes the general path
        if (fp.speculative && !f.whole() && fp.scan_data_pos + fp.seq_ends[0] + 2 != f.len) {
            fp.speculative = false;
            fp.need_full = true;
        }
        if (fp.speculative) {
            plan_swallowed_terminator(fp, f.base, f.len, true);
        } else if (fp.need_full) {
            if (!f.whole() && f.len <= kMaxFile) {
                f.gather(f.len);
                n_linearised.fetch_add(1, std::memory_order_relaxed);
            }
            plan_file_full(f.base, f.len, (int)i, fp);
        }
        fp.img.file_offset = off;
    });
    ingest_.full_walk_ms = ms_since(t0);
    ingest_.n_linearised = n_linearised.load();

    // ---- 5. merge into the batch's job list (file order)
    t0 = clk::now();
    std::vector<const uint8_t *> file_ptr((size_t)n);
    std::vector<size_t> file_len((size_t)n);
    for (int i = 0; i < n; i++) {
        FilePlan &fp = plans[i];
        const FileSegs &f = files[(size_t)i];
        file_ptr[(size_t)i] = f.base;
        file_len[(size_t)i] = f.len;
        const size_t first_job = jobs_.size();
        if (fp.speculative) ingest_.n_header_only++;
        else if (fp.need_full) ingest_.n_full_walk++;
        images_[i] = std::move(fp.img);
        ImagePlan &img = images_[i];
        img.file_len = f.len;
        img.jobs.clear();
        if (img.status != JPGPU_OK) continue;
        for (size_t j = 0; j < fp.jobs.size(); j++) {
            img.jobs.push_back((int)(first_job + j));
            job_image_.push_back(i);
            job_entropy_off_.push_back(fp.jobs[j].entropy ? (uint64_t)(fp.jobs[j].entropy - f.base) : 0u);
            jobs_.push_back(std::move(fp.jobs[j]));
        }
        if (img.swallow_job >= 0) img.swallow_job += (int)first_job;
    }
    plans.clear();
    files_resident_ = true;
    rc = layout_and_upload(file_ptr, file_len);
    files_resident_ = false;
    replay_possible_ = rc == JPGPU_OK && !entropy_only_;
    whole_files_ = rc == JPGPU_OK;  // (redo_swallowed: the files are in d_input_ as they came)
    ingest_.layout_ms = ms_since(t0);
    ingest_.total_ms = ms_since(t_begin);
    return rc;
}

// Step 2 of the ingest: caller memory -> HBM.
//  - pageable input: through the pinned staging ring.  The input buffer is cut into pieces that never cross a 32 MiB slot;
//    the crew copies pieces in buffer order, whoever completes a slot sends it off (one DMA per slot) and records the event
//    that frees the slot for the chunk n_slots later;
//  - page-locked input (JPGPU_UPLOAD_PINNED): one DMA per segment from where the caller's bytes lie; the slack between the
//    files is zeroed by one fill of the whole input buffer in front of the copies (~0.3 ms per GB, on the device).
int DeviceBatch::stage_files(WorkCrew &crew, const std::vector<FileSegs> &files, const std::vector<FilePlan> &plans, bool pinned) {
    hipError_t e = d_input_.reserve((size_t)input_bytes_);
    if (e != hipSuccess) return hip_fail(e, "hipMalloc(input)");
    constexpr size_t kMaxFile = 0x7FFFFFF0u;
    hipStream_t up = ctx_->upload_stream;
    uint8_t *d_in = (uint8_t *)d_input_.ptr;
    if (pinned && arena_span_.first) {
        // the slack in front of the span and behind it is read by the kernels' wide loads: defined (zero); what lies between
        // the files inside the span are the arena's own bytes (nothing a result depends on: every read is bounded by a length)
        e = hipMemsetAsync(d_in, 0, 256, up);
        const uint64_t tail = 256 + arena_span_.second;
        if (e == hipSuccess) e = hipMemsetAsync(d_in + tail, 0, (size_t)(input_bytes_ - tail), up);
        if (e != hipSuccess) return hip_fail(e, "hipMemsetAsync(input slack)");
        constexpr size_t kChunk = 32u << 20;
        for (size_t off = 0; off < arena_span_.second; off += kChunk) {
            const size_t m = std::min(kChunk, arena_span_.second - off);
            e = hipMemcpyAsync(d_in + 256 + off, arena_span_.first + off, m, hipMemcpyHostToDevice, up);
            if (e != hipSuccess) return hip_fail(e, "hipMemcpyAsync(pinned arena)");
            ingest_.n_pinned_dma++;
        }
        return JPGPU_OK;
    }
    if (pinned) {
        // segments scattered in page-locked memory: the device pulls them itself (gather_pinned_kernel, 32 KiB pieces) -- one
        // launch instead of one hipMemcpyAsync per segment (JPGPU_PINNED_MEMCPY=1: the copy engine, per segment, as before)
        e = hipMemsetAsync(d_in, 0, (size_t)input_bytes_, up);
        if (e != hipSuccess) return hip_fail(e, "hipMemsetAsync(input)");
        const bool by_memcpy = getenv("JPGPU_PINNED_MEMCPY") != nullptr;
        constexpr uint32_t kPiece = 32u << 10;
        std::vector<GatherPiece> gp;
        for (size_t i = 0; i < plans.size(); i++) {
            const FileSegs &f = files[i];
            if (f.len > kMaxFile || f.len == 0) continue;
            uint64_t off = plans[i].img.file_offset;
            for (int k = 0; k < f.n; k++) {
                if (!f.seg[k].len) continue;
                if (by_memcpy) {
                    e = hipMemcpyAsync(d_in + off, f.seg[k].data, f.seg[k].len, hipMemcpyHostToDevice, up);
                    if (e != hipSuccess) return hip_fail(e, "hipMemcpyAsync(pinned segment)");
                } else {
                    for (size_t at = 0; at < f.seg[k].len; at += kPiece)
                        gp.push_back({(uint64_t)(uintptr_t)(f.seg[k].data + at), off + at, (uint32_t)std::min<size_t>(kPiece, f.seg[k].len - at), 0u});
                }
                off += f.seg[k].len;
                ingest_.n_pinned_dma++;
            }
        }
        if (!gp.empty()) {
            e = d_gather_.reserve(gp.size() * sizeof(GatherPiece));
            if (e != hipSuccess) return hip_fail(e, "hipMalloc(gather list)");
            e = hipMemcpyAsync(d_gather_.ptr, gp.data(), gp.size() * sizeof(GatherPiece), hipMemcpyHostToDevice, up);
            if (e == hipSuccess) e = hipStreamSynchronize(up);  // `gp` is a local (pageable) vector
            if (e == hipSuccess) e = launch_gather_pinned(up, (const GatherPiece *)d_gather_.ptr, (int)gp.size(), d_in);
            if (e != hipSuccess) return hip_fail(e, "gather_pinned_kernel");
        }
        return JPGPU_OK;
    }
    struct Piece {
        const uint8_t *src;  // nullptr: zero fill (slack in front of the first file, behind the last, between files)
        uint64_t dst;
        uint32_t n;
    };
    StagingRing &ring = ctx_->staging;
    const uint64_t kSlot = ring.slot_bytes;
    const size_t n_slots = (size_t)ring.n_slots;
    constexpr uint32_t kPieceMax = 2u << 20;
    std::vector<Piece> pieces;
    auto add = [&](const uint8_t *src, uint64_t dst, uint64_t n) {
        while (n) {
            const uint64_t room = kSlot - dst % kSlot;
            const uint32_t m = (uint32_t)std::min<uint64_t>(std::min<uint64_t>(n, room), kPieceMax);
            pieces.push_back({src, dst, m});
            if (src) src += m;
            dst += m;
            n -= m;
        }
    };
    uint64_t pos = 0;
    for (size_t i = 0; i < plans.size(); i++) {
        const FileSegs &f = files[i];
        uint64_t off = plans[i].img.file_offset;
        if (f.len > kMaxFile || f.len == 0) continue;
        if (off > pos) add(nullptr, pos, off - pos);
        for (int k = 0; k < f.n; k++) {
            if (!f.seg[k].len) continue;
            add(f.seg[k].data, off, f.seg[k].len);
            off += f.seg[k].len;
        }
        pos = off;
    }
    if (input_bytes_ > pos) add(nullptr, pos, input_bytes_ - pos);

    const size_t n_chunks = (size_t)((input_bytes_ + kSlot - 1) / kSlot);
    std::vector<std::atomic<int>> remaining(n_chunks);
    std::vector<std::atomic<int>> state(n_chunks);  // 0 = being filled, 1 = DMA issued (event recorded), 2 = slot known drained
    for (size_t c = 0; c < n_chunks; c++) {
        remaining[c].store(0, std::memory_order_relaxed);
        state[c].store(0, std::memory_order_relaxed);
    }
    for (const Piece &p : pieces) remaining[p.dst / kSlot].fetch_add(1, std::memory_order_relaxed);
    for (size_t c = 0; c < std::min<size_t>(n_chunks, n_slots); c++) {
        if (!ring.slot[c]) {
            e = hipHostMalloc((void **)&ring.slot[c], kSlot, hipHostMallocDefault);
            if (e != hipSuccess) return hip_fail(e, "hipHostMalloc(staging)");
        }
        if (!ring.drained[c]) {
            e = hipEventCreateWithFlags(&ring.drained[c], hipEventDisableTiming | hipEventBlockingSync);
            if (e != hipSuccess) return hip_fail(e, "hipEventCreate(staging)");
        }
    }
    std::atomic<int> hip_error{(int)hipSuccess};
    const int device = ctx_->device;
    const uint64_t total = input_bytes_;
    crew.run(pieces.size(), [&](size_t k, int) {
        const Piece &p = pieces[k];
        const size_t c = (size_t)(p.dst / kSlot);
        const int slot = (int)(c % n_slots);
        if (hip_error.load(std::memory_order_relaxed) != (int)hipSuccess) return;
        (void)hipSetDevice(device);
        if (c >= n_slots) {
            // the slot still holds chunk c - n_slots until that chunk's DMA has read it
            std::atomic<int> &prev = state[c - n_slots];
            while (prev.load(std::memory_order_acquire) == 0) {
                if (hip_error.load(std::memory_order_relaxed) != (int)hipSuccess) return;
                std::this_thread::yield();
            }
            if (prev.load(std::memory_order_acquire) == 1) {
                const hipError_t es = hipEventSynchronize(ring.drained[slot]);
                if (es != hipSuccess) {
                    hip_error.store((int)es);
                    return;
                }
                prev.store(2, std::memory_order_release);
            }
        }
        uint8_t *dst = ring.slot[slot] + (p.dst - (uint64_t)c * kSlot);
        if (p.src) memcpy(dst, p.src, p.n);
        else memset(dst, 0, p.n);
        if (remaining[c].fetch_sub(1, std::memory_order_acq_rel) == 1) {
            const uint64_t base = (uint64_t)c * kSlot;
            const size_t bytes = (size_t)std::min<uint64_t>(kSlot, total - base);
            hipError_t ec = hipMemcpyAsync(d_in + base, ring.slot[slot], bytes, hipMemcpyHostToDevice, up);
            if (ec == hipSuccess) ec = hipEventRecord(ring.drained[slot], up);
            if (ec != hipSuccess) hip_error.store((int)ec);
            state[c].store(1, std::memory_order_release);
        }
    });
    if (hip_error.load() != (int)hipSuccess) {
        (void)hipStreamSynchronize(up);
        return hip_fail((hipError_t)hip_error.load(), "staged H2D");
    }
    return JPGPU_OK;
}

// Step 3 of the ingest: first marker that is not RSTn behind every planned SOS header (first_marker_kernel); synchronises
// the upload stream, so the caller's buffers are free once this returns.
int DeviceBatch::verify_plans(const std::vector<FilePlan> &plans, const std::vector<int> &spec, std::vector<uint32_t> &first) {
    const size_t n = spec.size();
    std::vector<uint32_t> host(3 * n);  // {offset lo, length} pairs, then offset hi
    uint32_t max_len = 0;
    for (size_t k = 0; k < n; k++) {
        const FilePlan &fp = plans[spec[k]];
        const uint64_t off = fp.img.file_offset + fp.scan_data_pos;
        const uint32_t dlen = (uint32_t)(fp.img.file_len - fp.scan_data_pos);
        host[2 * k] = (uint32_t)off;
        host[2 * k + 1] = dlen;
        host[2 * n + k] = (uint32_t)(off >> 32);
        max_len = std::max(max_len, dlen);
    }
    hipError_t e = d_verify_.reserve(4 * n * sizeof(uint32_t) + 256);
    if (e != hipSuccess) return hip_fail(e, "hipMalloc(verify)");
    StagingRing &ring = ctx_->staging;
    if (ring.verdict_cap < n) {
        if (ring.verdict) (void)hipHostFree(ring.verdict);
        ring.verdict = nullptr;
        ring.verdict_cap = 0;
        const size_t cap = std::max<size_t>(n, 4096);
        e = hipHostMalloc((void **)&ring.verdict, cap * sizeof(uint32_t), hipHostMallocDefault);
        if (e != hipSuccess) return hip_fail(e, "hipHostMalloc(verdict)");
        ring.verdict_cap = cap;
    }
    hipStream_t up = ctx_->upload_stream;
    uint32_t *d = (uint32_t *)d_verify_.ptr;
    e = hipMemcpyAsync(d, host.data(), 3 * n * sizeof(uint32_t), hipMemcpyHostToDevice, up);
    if (e == hipSuccess) e = hipMemsetAsync(d + 3 * n, 0xFF, n * sizeof(uint32_t), up);
    if (e == hipSuccess) e = launch_first_marker(up, (const uint8_t *)d_input_.ptr, d, d + 2 * n, (int)n, max_len, d + 3 * n);
    if (e == hipSuccess) e = hipMemcpyAsync(ring.verdict, d + 3 * n, n * sizeof(uint32_t), hipMemcpyDeviceToHost, up);
    if (e == hipSuccess) e = hipStreamSynchronize(up);
    if (e != hipSuccess) return hip_fail(e, "ingest verification");
    first.assign(ring.verdict, ring.verdict + n);
    return JPGPU_OK;
}

int DeviceBatch::upload_single_job(const ScanJob &job, int format, const void *initial_output, size_t initial_output_bytes) {
    whole_files_ = false;
    if (format < 0 || format >= kNumOutputFormats) return fail(JPGPU_ERR_ARGUMENT, "unknown format");
    format_ = format;
    images_.assign(1, ImagePlan());
    jobs_.assign(1, job);
    job_image_.assign(1, 0);
    job_entropy_off_.assign(1, 0);
    ImagePlan &img = images_[0];
    img.sof = kSOF0;
    img.file_len = job.entropy_len;
    plan_image_geometry(img, job.geo);
    img.blocks_per_mcu = (uint32_t)job.blocks_per_mcu;
    img.jobs.push_back(0);
    std::vector<const uint8_t *> fp(1, job.entropy);
    std::vector<size_t> fl(1, job.entropy_len);
    keep_canvas_ = initial_output && initial_output_bytes;
    int rc = layout_and_upload(fp, fl);
    keep_canvas_ = false;
    if (rc != JPGPU_OK) return rc;
    if (initial_output && initial_output_bytes) {
        out_clear_.clear();  // the caller's buffer is the canvas: what this scan does not write keeps the caller's samples
        const size_t nbytes = std::min<size_t>(initial_output_bytes, img.out_bytes);
        hipError_t e = hipMemcpyAsync((uint8_t *)d_out_.ptr + img.out_offset, initial_output, nbytes, hipMemcpyHostToDevice, ctx_->upload_stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx_->upload_stream);
        if (e != hipSuccess) return hip_fail(e, "hipMemcpyAsync(initial output)");
    }
    return JPGPU_OK;
}

int DeviceBatch::upload_progressive_frame(const ProgressiveFrame &frame, const uint8_t *file, size_t file_len, int sof, int format) {
    whole_files_ = false;
    if (format < 0 || format >= kNumOutputFormats) return fail(JPGPU_ERR_ARGUMENT, "unknown format");
    format_ = format;
    images_.assign(1, ImagePlan());
    jobs_.clear();
    job_image_.clear();
    job_entropy_off_.clear();
    jobs_.push_back(frame.make_frame_job());  // throws DecodeError for scan orders the reference mangles
    for (const ScanJob &j : frame.scans()) jobs_.push_back(j);
    ImagePlan &img = images_[0];
    img.sof = (uint8_t)sof;
    img.file_len = file_len;
    plan_image_geometry(img, jobs_[0].geo);
    img.blocks_per_mcu = (uint32_t)jobs_[0].blocks_per_mcu;
    for (size_t j = 0; j < jobs_.size(); j++) {
        img.jobs.push_back((int)j);
        job_image_.push_back(0);
        job_entropy_off_.push_back(jobs_[j].entropy ? (uint64_t)(jobs_[j].entropy - file) : 0u);
    }
    std::vector<const uint8_t *> fp(1, file);
    std::vector<size_t> fl(1, file_len);
    const int rc = layout_and_upload(fp, fl);
    replay_possible_ = rc == JPGPU_OK;  // (a scan that fails: the frame is issued again in file order up to the throw, like a file of a batch)
    return rc;
}

int DeviceBatch::upload_progressive_scan(const ProgressiveFrame &frame, int scan_index, bool first_scan) {
    whole_files_ = false;
    if (scan_index < 0 || scan_index >= (int)frame.scans().size()) return fail(JPGPU_ERR_ARGUMENT, "progressive scan index out of range");
    format_ = JPGPU_FMT_INTERLEAVED_U8;  // no samples are produced by a scan; the smallest output layout
    images_.assign(1, ImagePlan());
    jobs_.clear();
    job_image_.clear();
    job_entropy_off_.clear();
    jobs_.push_back(frame.make_frame_job());  // the store's geometry (and position: block 0 of the coefficient buffer)
    jobs_.push_back(frame.scans()[(size_t)scan_index]);
    ScanJob &scan = jobs_.back();
    // the scans this one depends on ran in earlier calls: nothing to wait for inside the launch
    scan.n_deps = 0;
    scan.deps[0] = scan.deps[1] = scan.deps[2] = -1;
    scan.ordinal = 0;
    scan.has_consumers = false;
    ImagePlan &img = images_[0];
    img.sof = kSOF2;
    img.file_len = scan.entropy_len;
    plan_image_geometry(img, jobs_[0].geo);
    img.blocks_per_mcu = (uint32_t)jobs_[0].blocks_per_mcu;
    for (size_t j = 0; j < jobs_.size(); j++) {
        img.jobs.push_back((int)j);
        job_image_.push_back(0);
        job_entropy_off_.push_back(0);
    }
    std::vector<const uint8_t *> fp(1, scan.entropy);
    std::vector<size_t> fl(1, scan.entropy_len);
    const int rc = layout_and_upload(fp, fl);
    keep_progressive_store_ = !first_scan;
    defer_refusal_ = true;
    return rc;
}

int DeviceBatch::snapshot_progressive_store() {
    prog_snapshot_blocks_ = 0;
    if (prog_clear_.empty() || !d_coefs_.ptr) return JPGPU_OK;
    const uint64_t first = prog_clear_[0].first, blocks = prog_clear_[0].second;
    hipError_t e = d_prog_snapshot_.reserve((size_t)blocks * 128 + 256);
    if (e != hipSuccess) return hip_fail(e, "hipMalloc(store snapshot)");
    e = hipMemcpyAsync(d_prog_snapshot_.ptr, (const int16_t *)d_coefs_.ptr + first * 64, (size_t)blocks * 128, hipMemcpyDeviceToDevice, ctx_->stream);
    if (e != hipSuccess) return hip_fail(e, "hipMemcpyAsync(store snapshot)");
    prog_snapshot_blocks_ = blocks;
    return JPGPU_OK;
}

int DeviceBatch::rerun_failed_progressive_scan(bool first_scan) {
    if (jobs_.size() != 2 || jobs_[1].kind != kScanProgressive || h_status_.size() != 2 || h_status_[1].first_error == kNoError) return JPGPU_OK;
    if (jobs_[1].force_lane) return JPGPU_OK;  // (this WAS the exact kernel)
    jobs_[1].force_lane = true;
    jobs_[1].last_interval = h_status_[1].first_error >> 8;  // the lowest failing restart interval: the reference never got behind it
    std::vector<const uint8_t *> fp(1, nullptr);
    std::vector<size_t> fl(1, images_[0].file_len);
    files_resident_ = true;  // (the scan's bytes are in HBM already)
    int rc = layout_and_upload(fp, fl);
    files_resident_ = false;
    if (rc != JPGPU_OK) return rc;
    if (first_scan || prog_snapshot_blocks_ == 0 || prog_clear_.empty()) {
        keep_progressive_store_ = false;  // run_progressive() clears the store like JpegBlockAllocator.Allocate
    } else {
        keep_progressive_store_ = true;
        const uint64_t blocks = std::min<uint64_t>(prog_snapshot_blocks_, prog_clear_[0].second);
        hipError_t e = hipMemcpyAsync((int16_t *)d_coefs_.ptr + prog_clear_[0].first * 64, d_prog_snapshot_.ptr, (size_t)blocks * 128, hipMemcpyDeviceToDevice, ctx_->stream);
        if (e != hipSuccess) return hip_fail(e, "hipMemcpyAsync(store restore)");
    }
    defer_refusal_ = true;
    if ((rc = run_marker_index()) != JPGPU_OK) return rc;
    if ((rc = run_huffman()) != JPGPU_OK) return rc;
    rc = sync();
    keep_progressive_store_ = true;  // whatever comes next works on this store
    return rc;
}

int DeviceBatch::upload_progressive_dispose(const ProgressiveFrame &frame, int format) {
    whole_files_ = false;
    if (format < 0 || format >= kNumOutputFormats) return fail(JPGPU_ERR_ARGUMENT, "unknown format");
    format_ = format;
    images_.assign(1, ImagePlan());
    jobs_.clear();
    job_image_.clear();
    job_entropy_off_.clear();
    jobs_.push_back(frame.make_frame_job());
    ImagePlan &img = images_[0];
    img.sof = kSOF2;
    plan_image_geometry(img, jobs_[0].geo);
    img.blocks_per_mcu = (uint32_t)jobs_[0].blocks_per_mcu;
    img.jobs.push_back(0);
    job_image_.push_back(0);
    job_entropy_off_.push_back(0);
    std::vector<const uint8_t *> fp(1, nullptr);
    std::vector<size_t> fl(1, 0);
    const bool store_holds_samples = dispose_done_;  // an earlier Dispose() of this session has transformed the store in place
    const int rc = layout_and_upload(fp, fl);
    dispose_done_ = store_holds_samples && !dispose_jobs_.empty();
    keep_progressive_store_ = true;
    defer_refusal_ = false;
    return rc;
}

int DeviceBatch::upload_frames(const jpgpu_frame *frames, const uint16_t *qt, int n, int format) {
    whole_files_ = false;
    if (n < 0 || (n > 0 && (!frames || !qt))) return fail(JPGPU_ERR_ARGUMENT, "jpgpu_batch_upload_frames: null argument");
    if (format < 0 || format >= kNumOutputFormats) return fail(JPGPU_ERR_ARGUMENT, "jpgpu_batch_upload_frames: unknown format");
    format_ = format;
    images_.assign((size_t)n, ImagePlan());
    jobs_.clear();
    job_image_.clear();
    job_entropy_off_.clear();
    std::vector<const uint8_t *> fp((size_t)n, nullptr);
    std::vector<size_t> fl((size_t)n, 0);
    for (int i = 0; i < n; i++) {
        ImagePlan &img = images_[i];
        try {
            const jpgpu_frame &f = frames[i];
            if (f.num_components == 0 || f.num_components > 4) throw DecodeError(JPGPU_ERR_NOT_SUPPORTED, "1..4 components are supported.", kDetailUnsupportedFrame);
            HostDecoder dec;
            FrameHeader fh;
            fh.precision = f.precision;
            fh.lines = f.height;
            fh.samples_per_line = f.width;
            fh.num_components = f.num_components;
            ScanHeader sh;
            sh.num_components = f.num_components;
            sh.se = 63;
            for (int c = 0; c < f.num_components; c++) {
                fh.components.push_back({f.comp[c].identifier, f.comp[c].h, f.comp[c].v, f.comp[c].tq});
                sh.components.push_back({f.comp[c].identifier, 0, 0});
                if (f.comp[c].tq > 3) throw DecodeError(JPGPU_ERR_ARGUMENT, "quantisation table selector out of range");
                QuantTable q;
                q.identifier = f.comp[c].tq;
                memcpy(q.elements, qt + ((size_t)i * 4 + f.comp[c].tq) * 64, sizeof q.elements);
                dec.set_quantization_table(q);
            }
            // the IDCT stage needs no Huffman tables; a placeholder keeps the scan-job builder's checks satisfied
            HuffTable dummy;
            const uint8_t bits[16] = {0, 1};
            const uint8_t vals[1] = {0};
            HuffTable::from_bits_values(0, 0, bits, vals, 1, &dummy);
            dec.set_huffman_table(dummy);
            dummy.table_class = 1;
            dec.set_huffman_table(dummy);
            dec.set_frame_header(fh);
            img.sof = f.sof;
            const BaselineGeometry geo = BaselineGeometry::latch(dec, fh);
            jobs_.push_back(make_scan_job(dec, geo, sh, nullptr, 0));
            plan_image_geometry(img, geo);
            img.blocks_per_mcu = (uint32_t)jobs_.back().blocks_per_mcu;
            img.jobs.push_back((int)jobs_.size() - 1);
            job_image_.push_back(i);
            job_entropy_off_.push_back(0);
        } catch (const DecodeError &e) {
            img.status = e.status;
            img.detail = e.detail;
            img.error = e.what();
        }
    }
    return layout_and_upload(fp, fl);
}

}  // namespace jpgpu
