// jpeglibrary_amd/csrc/device_batch.cpp -- HBM layout, uploads and kernel launches for a batch of scan jobs.
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
    for (DevBuffer *b : {&d_sub_work_, &d_sub_final_work_, &d_sub_scan_ids_, &d_sub_exit_a_, &d_sub_exit_b_, &d_sub_nblk_, &d_sub_first_, &d_sub_entry_, &d_sub_dcsum_, &d_sub_dcentry_, &d_sub_changed_, &d_sub_same_, &d_sub_perm_, &d_k1_desc_, &d_k1_tickets_, &d_k1_order_, &d_prog_snapshot_, &d_dispose_, &d_verify_, &d_lut_pool_, &d_prog_work_, &d_prog_sync_, &d_planes_, &d_extend_desc_, &d_gather_, &d_rgb_scratch_, &d_chunk_work_, &d_chunk_sums_, &d_unstuffed_, &d_ends_u_, &d_input_, &d_scans_, &d_status_, &d_ends_, &d_huff_pool_, &d_quant_pool_, &d_huff_work_, &d_idct_work_, &d_idct_work_split_, &d_coefs_, &d_out_})
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

class PlanHandler final : public ScanHandler {
  public:
    explicit PlanHandler(std::vector<ScanJob> *jobs, bool first_scan_only = false) : jobs_(jobs), first_scan_only_(first_scan_only) {}
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
            ends_.push_back(find_scan_end(entropy, len));
            reader.try_advance((int)ends_.back());
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
    PlanHandler handler(&fp.jobs, entropy_only_);
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
    if (entropy_only_ || len > 0x7FFFFFF0u) {
        fp.need_full = true;  // optimizer walks have rules of their own; oversize files are refused by the full path
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
    n_huff_work_ = (int)huff_work.size();
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
        const bool no_pool = getenv("JPGPU_SF_NO_POOL") != nullptr || !subseq_pool_fits(k2_tab_bytes_);  // (A/B switch; ten waves + the tables must fit a CU)
        auto same_tables = [&](const DevScan &a, const DevScan &b) {
            return memcmp(a.huff_pool, b.huff_pool, sizeof a.huff_pool) == 0 && a.scan_components == b.scan_components &&
                   a.blocks_per_mcu == b.blocks_per_mcu && memcmp(a.blk_comp, b.blk_comp, sizeof a.blk_comp) == 0 &&
                   memcmp(a.comp, b.comp, sizeof a.comp) == 0;
        };
        // (a batch that cannot fill the machine with two subsequences per lane takes one: twice the waves, half as long each)
        sub_final_spl_ = (kSubFinalSubsPerLane >= 2 && total_subs_ >= kSubFinalFewSubs && getenv("JPGPU_SF_ONE_SUB") == nullptr) ? 2 : 1;
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
                                 (uint32_t *)d_sub_perm_.ptr, sub_final_spl_, k2_tab_bytes_);
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
                                      (const HuffWork *)d_sub_work_.ptr + n_sub_work_, n_sub_gather_);
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
            if (j != img->swallow_job) {
                res->status = JPGPU_ERR_NOT_SUPPORTED;
                res->detail = kDetailUnsupportedFrame;
                ctx_->last_error = "A scan that leaves one byte unread in front of its terminating marker is only supported as the last scan.";
                return JPGPU_OK;
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

int DeviceBatch::download_output(int i, void *dst, size_t cap) {
    const ImagePlan *img = image(i);
    if (!img || !dst) return fail(JPGPU_ERR_ARGUMENT, "jpgpu_batch_download_output: bad argument");
    if (img->status != JPGPU_OK) return fail(img->status, img->error);
    if (cap < img->out_bytes) return fail(JPGPU_ERR_ARGUMENT, "Destination buffer is too small.");
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
