// jpeglibrary_amd/csrc/host.h -- host side of the drop-in: marker/table parsing and the JpegDecoder state machine.
//
// Mirrors the reference's host-only logic (all paths relative to /root/reference/src/JpegLibrary):
//   JpegReader.cs (marker sync, lengths), JpegFrameHeader.cs / JpegScanHeader.cs (SOF / SOS payloads),
//   JpegQuantizationTable.cs / JpegHuffmanDecodingTable.cs (DQT / DHT parse + canonical code build),
//   JpegDecoder.cs (Identify / Decode marker loops, table registry, DRI latch).
// The per-block arithmetic is NOT here: it runs in the HIP kernels (k1_markers.hip ... k3_idct.hip).
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <optional>
#include <stdexcept>
#include <string>
#include <vector>

#include "common.h"

namespace jpgpu {

// Exception carrying one of the reference's exception classes as a jpgpu_status value.
struct DecodeError : std::runtime_error {
    int status;
    int detail;
    DecodeError(int status_, const std::string &msg, int detail_ = 0) : std::runtime_error(msg), status(status_), detail(detail_) {}
};
[[noreturn]] void throw_invalid_data(const std::string &msg, int detail = 0);                 // InvalidDataException
[[noreturn]] void throw_invalid_data_at(int offset, const std::string &msg, int detail = 0);  // "... at offset N. ..."
[[noreturn]] void throw_invalid_operation(const std::string &msg, int detail = 0);            // InvalidOperationException

// ref: JpegMarker.cs
enum Marker : uint8_t {
    kSOF0 = 0xC0, kSOF1 = 0xC1, kSOF2 = 0xC2, kSOF3 = 0xC3, kDHT = 0xC4, kSOF5 = 0xC5, kSOF6 = 0xC6, kSOF7 = 0xC7,
    kSOF9 = 0xC9, kSOF10 = 0xCA, kSOF11 = 0xCB, kDAC = 0xCC, kSOF13 = 0xCD, kSOF14 = 0xCE, kSOF15 = 0xCF,
    kRST0 = 0xD0, kRST7 = 0xD7, kSOI = 0xD8, kEOI = 0xD9, kSOS = 0xDA, kDQT = 0xDB, kDRI = 0xDD, kPadding = 0xFF
};
inline bool is_restart_marker(int m) { return m >= kRST0 && m <= kRST7; }  // ref: JpegMarkerHelper.cs:7-10

// ref: JpegReader.cs -- cursor over the caller's bytes (the caller keeps ownership; nothing is copied).
class MarkerReader {
  public:
    MarkerReader() = default;
    MarkerReader(const uint8_t *data, size_t len) : p_(data), n_(len), initial_(len) {}
    bool is_empty() const { return n_ == 0; }
    int remaining_byte_count() const { return (int)n_; }
    int consumed_byte_count() const { return (int)(initial_ - n_); }
    const uint8_t *remaining_bytes() const { return p_; }
    bool try_read_start_of_image();             // :98-112
    bool try_read_marker(int *marker);          // :120-158
    bool try_read_length(uint16_t *length);     // :166-177 (incl. the (b0<<8)|(b1-2) quirk)
    bool try_read_bytes(int length, const uint8_t **bytes);  // :203-214
    bool try_advance(int length);               // :239-247

  private:
    const uint8_t *p_ = nullptr;
    size_t n_ = 0, initial_ = 0;
};

// ref: JpegFrameHeader.cs
struct FrameComponent {
    uint8_t identifier = 0, h = 0, v = 0, tq = 0;
};
struct FrameHeader {
    uint8_t precision = 0;
    uint16_t lines = 0, samples_per_line = 0;
    uint8_t num_components = 0;
    std::vector<FrameComponent> components;  // empty when parsed metadata-only
    static bool try_parse(const uint8_t *buf, size_t len, bool metadata_only, FrameHeader *out, int *consumed);  // :137-182
};

// ref: JpegScanHeader.cs
struct ScanComponent {
    uint8_t selector = 0, td = 0, ta = 0;
};
struct ScanHeader {
    uint8_t num_components = 0, ss = 0, se = 0, ah = 0, al = 0;
    std::vector<ScanComponent> components;
    static bool try_parse(const uint8_t *buf, size_t len, bool metadata_only, ScanHeader *out, int *consumed);  // :157-205
};

// ref: JpegQuantizationTable.cs -- elements stay in zig-zag order (:47)
struct QuantTable {
    uint8_t precision = 0, identifier = 0;
    uint16_t elements[64] = {};
    static bool try_parse(const uint8_t *buf, size_t len, QuantTable *out, int *consumed);  // :99-113, :192-232
};

// ref: JpegHuffmanDecodingTable.cs -- BITS/HUFFVAL plus the reference's derived arrays
struct HuffTable {
    uint8_t table_class = 0, identifier = 0;
    uint8_t bits[16] = {};
    uint16_t num_values = 0;
    uint8_t values[256] = {};
    uint16_t maxcode[18] = {};
    uint8_t valoffset[19] = {};
    uint8_t la_size[256] = {}, la_symbol[256] = {};
    static bool try_parse(const uint8_t *buf, size_t len, HuffTable *out, int *consumed);  // :152-166, :249-291
    static bool from_bits_values(uint8_t table_class, uint8_t identifier, const uint8_t bits[16], const uint8_t *values,
                                 int num_values, HuffTable *out);
    // Reference Lookup()/LookupSlow() (:73-113): returns false for "Invalid Huffman code encountered."
    bool lookup(int code16, int *size, int *symbol) const;
    // Device image: kHuffLutBits-wide first level derived by evaluating lookup() for every prefix.
    void to_device(DevHuffTable *out) const;
};

// Everything the scan decoder pulls from the decoder at ProcessScan time
// (ref: ScanDecoder/JpegHuffmanScanDecoder.cs:17-72 InitDecodeComponents).
struct ResolvedScanComponent {
    int component_index = 0;
    uint8_t h = 0, v = 0;
    int hs = 1, vs = 1;
    const HuffTable *dc = nullptr, *ac = nullptr;
    const QuantTable *quant = nullptr;
};

class HostDecoder;

// What happens at SOS.  Implementations: the batch planner (records a scan job) and the immediate GPU scan runner.
class ScanHandler {
  public:
    virtual ~ScanHandler() = default;
    // ref: JpegScanDecoder.Create at SOF (JpegDecoder.cs:568-569); sof = marker byte.
    virtual void on_frame(HostDecoder &dec, int sof) = 0;
    // ref: JpegScanDecoder.ProcessScan (JpegDecoder.cs:597-598).  Must leave `reader` where the reference would.
    virtual void on_scan(HostDecoder &dec, MarkerReader &reader, const ScanHeader &scan) = 0;
    // ref: _scanDecoder.Dispose() in Decode's finally (JpegDecoder.cs:545-549)
    virtual void on_dispose(HostDecoder &dec) = 0;
    // A handler that DEFERS its scans to on_dispose (the decoder mirror collects a progressive frame's scans during the walk and
    // decodes them in one device pass): true when what on_dispose threw is the failure of such a scan.  The reference decoded that
    // scan at its SOS, in front of whatever ended the walk behind it, so that failure is the one the caller sees (round 6).
    virtual bool dispose_failure_is_a_deferred_scan() const { return false; }
};

// ref: JpegDecoder.cs -- the public decoder's state (input, frame header, DRI, table registry).
class HostDecoder {
  public:
    void set_input(const uint8_t *data, size_t len);  // :56-62
    int identify(bool load_quantization_tables);      // :75-105, returns consumed byte count
    // The same walk, stopped behind the header of the first SOS (the batch ingest's header-only parse: what lies behind
    // that header is looked at by the device, DeviceBatch::upload_files).  Returns true when it stopped there
    // (*scan_data_pos = offset of the first entropy byte), false when the walk reached EOI / the end of the data first --
    // then it WAS the whole Identify(), "Frame header was not found." included.
    bool identify_until_scan(bool load_quantization_tables, size_t *scan_data_pos);
    bool try_estimate_quality(float *quality) const;  // :169-196
    void decode(ScanHandler &handler, bool have_output_writer);  // :509-550
    void load_tables(const uint8_t *data, size_t len);           // :319-363

    const FrameHeader &frame_header() const;  // throws "Call Identify() before this operation." (:378)
    bool has_frame_header() const { return frame_.has_value(); }
    void set_frame_header(const FrameHeader &fh) { frame_ = fh; }
    int start_of_frame() const { return start_of_frame_; }
    void set_start_of_frame(int m) { start_of_frame_ = m; }
    uint16_t restart_interval() const { return restart_interval_; }
    void set_restart_interval(int v);  // :662-670
    int maximum_horizontal_sampling() const;
    int maximum_vertical_sampling() const;

    const HuffTable *huffman_table(bool is_dc, uint8_t identifier) const;  // :869-886
    const QuantTable *quantization_table(uint8_t identifier) const;        // :910-925 (nullptr == IsEmpty)
    void set_huffman_table(const HuffTable &t);                            // :793-815
    void set_quantization_table(const QuantTable &t);                      // :840-861

    void reset_input() { input_ = nullptr; input_len_ = 0; }                            // :941
    void reset_header() { frame_.reset(); restart_interval_ = 0; }                      // :949
    void reset_tables() { huff_.clear(); quant_.clear(); }                              // :960
    void clear_huffman_tables() { huff_.clear(); }                                      // ClearHuffmanTable :768-771
    void clear_quantization_tables() { quant_.clear(); }                                // ClearQuantizationTable :784-787
    const uint8_t *input() const { return input_; }
    size_t input_len() const { return input_len_; }

    // ref: InitDecodeComponents (ScanDecoder/JpegHuffmanScanDecoder.cs:17-72)
    int resolve_scan(const FrameHeader &fh, const ScanHeader &sh, ResolvedScanComponent out[kMaxScanComponents], bool optimizer_rules = false) const;

  private:
    bool process_marker_for_identification(int marker, MarkerReader &r, bool load_qt);  // :114-162
    bool process_marker_for_decode(int marker, MarkerReader &r, ScanHandler &h);        // :558-617
    void process_other_marker(MarkerReader &r);                                         // :251-263
    void process_frame_header(MarkerReader &r, bool metadata_only, bool override_allowed);  // :265-289
    ScanHeader process_scan_header(MarkerReader &r, bool metadata_only);                // :291-307
    void process_dri(MarkerReader &r);                                                  // :635-650
    void process_dht(MarkerReader &r);                                                  // :672-700
    void process_dqt(MarkerReader &r, bool load);                                       // :732-763

    const uint8_t *input_ = nullptr;
    size_t input_len_ = 0;
    std::optional<FrameHeader> frame_;
    uint16_t restart_interval_ = 0;
    int start_of_frame_ = 0;
    std::vector<HuffTable> huff_;
    std::vector<QuantTable> quant_;
    bool scan_decoder_created_ = false;
};

// Geometry of a baseline scan decoder instance, latched at SOF time
// (ref: ScanDecoder/JpegHuffmanBaselineScanDecoder.cs:23-49; DRI is latched HERE -- SURVEY F4).
struct BaselineGeometry {
    FrameHeader frame;
    int max_h = 1, max_v = 1;
    uint16_t restart_interval = 0;
    int mcus_per_line = 0, mcus_per_column = 0, level_shift = 0;
    static BaselineGeometry latch(const HostDecoder &dec, const FrameHeader &fh);
};

// Host-side description of one scan job before it is laid out in a device batch.
struct ScanJob {
    BaselineGeometry geo;
    int scan_components = 0;
    ResolvedScanComponent comp[kMaxScanComponents];
    HuffTable huff_copy[kMaxHuffSlots];  // tables are snapshotted: later DHTs may replace registry entries
    QuantTable quant_copy[kMaxScanComponents];
    int n_huff = 0;
    uint8_t dc_slot[kMaxScanComponents] = {}, ac_slot[kMaxScanComponents] = {};
    const uint8_t *entropy = nullptr;  // reader.RemainingBytes at ProcessScan entry
    size_t entropy_len = 0;
    int blocks_per_mcu = 0;
    uint8_t blk_comp[kMaxBlocksPerMcu] = {}, blk_x[kMaxBlocksPerMcu] = {}, blk_y[kMaxBlocksPerMcu] = {};
    // progressive frames (ScanKind): kScanFrameOnly = the Dispose() pass (dequantise + IDCT + Flush) over the frame's
    // coefficient store, kScanProgressive = one entropy scan accumulating into that store
    int kind = kScanSequential;
    uint8_t ss = 0, se = 63, ah = 0, al = 0;
    int ordinal = 0;              // dependency level inside the frame: scans of one level may run together
    // direct dependencies (indices into the frame's scan list; transitive ones dropped): with them the scans of a frame
    // can run as ONE launch, a dependent scan following its producers MCU row by MCU row (device_batch.cpp)
    int deps[3] = {-1, -1, -1};
    int n_deps = 0;           // > 3: the frame's scans run level by level
    uint64_t dep_closure = 0; // every scan this one transitively depends on (bit = index in the frame's scan list)
    bool has_consumers = false;
    // the partial flush of a progressive file that failed (DeviceBatch::replay_failed_progressive): scans behind the failing one
    // never ran in the reference (disabled); the failing one runs on the kernel that walks and stores the way the reference does
    bool disabled = false, force_lane = false;
    // planned as a scan that leaves one whole byte unread in front of its terminating marker (DeviceBatch::redo_swallowed, round 6): the
    // walk behind it started one byte into that marker
    bool forced_swallow = false;
    uint32_t last_interval = 0xFFFFFFFFu;  // ... and not behind the restart interval in which it threw
    uint16_t scan_dri = 0;        // DRI as read at ProcessScan time (ref: ...ProgressiveScanDecoder.cs:78), not at SOF
    uint8_t frame_bpm = 0;
    uint8_t fblk_base[kMaxScanComponents] = {};
    uint16_t hblocks[kMaxScanComponents] = {}, vblocks[kMaxScanComponents] = {};
    uint32_t units_per_line = 0, total_units = 0;
    std::string refuse;  // frame job: non-empty = report NotSupported once every scan of the frame decoded cleanly
    // frame job: the decoder's component slots do not map one to one onto the frame's components (or no scan was processed):
    // Dispose() as the reference runs it -- component c is transformed dispose_n[c] times with these tables, in slot order
    bool dispose_generic = false;
    uint8_t dispose_n[kMaxScanComponents] = {};
    QuantTable dispose_q[kMaxScanComponents][kMaxScanComponents];
};
// Builds a ScanJob (validates tables like ProcessScan :69-82). Throws DecodeError with the reference's messages.
// optimizer_rules: JpegOptimizer's view of the same scan (JpegOptimizer.cs:381-413): quantisation tables play no part, and a
// Huffman table that is not defined is only met (as a null reference) when a block that needs it is reached -- the slot
// is kNullHuffSlot and the device walk reports kDetailNullTable there.
ScanJob make_scan_job(const HostDecoder &dec, const BaselineGeometry &geo, const ScanHeader &scan, const uint8_t *entropy,
                      size_t entropy_len, bool optimizer_rules = false);
constexpr uint8_t kNullHuffSlot = 0xFF;

// Host side of JpegHuffmanProgressiveScanDecoder (ref: ScanDecoder/JpegHuffmanProgressiveScanDecoder.cs): collects the
// scans of one SOF2 frame; the entropy decoding and the Dispose() pass run on the GPU.
class ProgressiveFrame {
  public:
    // constructor of the scan decoder (:36-55) + JpegBlockAllocator.Allocate (JpegBlockAllocator.cs:35-84)
    void begin(const HostDecoder &dec, const FrameHeader &fh);
    // ProcessScan (:57-90): validates the tables the scan needs, snapshots them, records the job.
    // The reference never advances the outer reader here; neither do we.
    void add_scan(const HostDecoder &dec, const ScanHeader &scan, const uint8_t *entropy, size_t entropy_len);
    // The job of the Dispose() pass (:421-470): components and quantisation tables as the LAST scans left them in
    // the decoder's component slots.  Scan orders whose slots do not cover every frame component exactly
    // once (the reference then transforms some component twice and another never) get ScanJob::refuse set.
    ScanJob make_frame_job() const;
    const BaselineGeometry &geo() const { return geo_; }
    std::vector<ScanJob> &scans() { return scans_; }
    const std::vector<ScanJob> &scans() const { return scans_; }
    bool active() const { return active_; }
    void reset() { active_ = false; scans_.clear(); }

  private:
    bool active_ = false;
    BaselineGeometry geo_;
    int slots_alloc_ = 0;
    bool slot_set_[kMaxScanComponents] = {};
    bool slot_noquant_[kMaxScanComponents] = {};  // the slot's component has no quantisation table (ProcessScan throws right behind InitDecodeComponents)
    ResolvedScanComponent slot_[kMaxScanComponents];
    QuantTable slot_quant_[kMaxScanComponents];
    uint16_t hblocks_[kMaxScanComponents] = {}, vblocks_[kMaxScanComponents] = {};
    uint8_t fblk_base_[kMaxScanComponents] = {};
    uint8_t frame_bpm_ = 0;
    std::vector<ScanJob> scans_;
};

// Fixed-point factors of the reference's YCbCr -> RGB tables, derived the way JpegYCbCrToRgbConverter's constructor and
// Init derive them (ref: apps/JpegDecode/JpegYCbCrToRgbConverter.cs:24-48, :87-99, Fix :121-124): float32 arithmetic.
YccRgbFactors ycc_rgb_factors();

// Scans forward over entropy-coded data to the next marker that is not RSTn, the way the reference's readers end up
// (ref: JpegReader.cs:120-158 + ScanDecoder/JpegHuffmanBaselineScanDecoder.cs:167-176).
// Returns the offset of the marker's FF byte, or len when the data runs out.
size_t find_scan_end(const uint8_t *data, size_t len, std::string *restart_markers = nullptr);  // optionally: the RSTn bytes in front of it

}  // namespace jpgpu
