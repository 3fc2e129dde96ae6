// jpeglibrary_amd/csrc/device_optimize.h -- device-resident batch of JpegOptimizer runs (see device_optimize.cpp)
#pragma once
#include <string>
#include <vector>

#include "../../include/jpgpu.h"
#include "device_batch.h"
#include "encode_kernels.h"

namespace jpgpu {

// JpegHuffmanEncodingTableBuilder.Build(false) (ref: JpegHuffmanEncodingTableBuilder.cs:68-175, 237-283): canonical
// codes in DHT order from the 256 symbol counts.  false = "No symbol is recorded." / a code size beyond the
// reference's 60-entry array.
struct OptimalCode {
    uint16_t code;
    uint8_t symbol, length;
};
// perm[i] = original index of the key the restated Array.Sort(T[], Comparison<T>) leaves at position i (ascending by key)
void net_sort_permutation(const int32_t *keys, int n, int32_t *perm);
bool build_optimal_table(const uint32_t freq[256], std::vector<OptimalCode> *codes, bool most_optimal = false);  // Build(optimal)

class OptimizeBatch {
  public:
    explicit OptimizeBatch(jpgpu_ctx *ctx) : ctx_(ctx), batch_(ctx) {}
    ~OptimizeBatch();
    // SetInput x n: host marker walks (Scan()'s and Optimize()'s) + H2D of the files
    int upload(const uint8_t *const *jpeg, const size_t *len, int n, int strip);
    // Scan() + Optimize(strip) x n: statistics, tables, transcode
    int run();
    int size() const { return (int)plans_.size(); }
    int result(int i, jpgpu_image_result *res, size_t *out_len);
    int download(int i, void *dst, size_t cap);
    // the statistics Scan() collected for table `t` of image i (builder-creation order); false past the last table
    bool statistics(int i, int t, uint8_t *table_class, uint8_t *identifier, uint32_t counts[256]) const;
    float last_ms() const { return last_ms_; }
    void set_most_optimal_coding(bool on) { most_optimal_ = on; }  // JpegOptimizer.MostOptimalCoding

  private:
    // Optimize()'s output is a fixed sequence of pieces: bytes known on the host, the new DHT segment, the scan's data
    struct Piece {
        enum Kind { kBytes, kHuffmanTables, kEntropy } kind;
        std::string bytes;
    };
    struct Plan {
        int status = JPGPU_OK, detail = 0;
        std::string error;
        // failures of the marker walks BEHIND the scan: the reference meets them only once the scan itself went through
        // (Scan() throws from ProcessScanBaseline first), so they are reported after the device-side status
        int late_status = JPGPU_OK, late_detail = 0;
        bool build_failed = false;  // BuildTables threw (it runs at the end of the scan: in front of anything a later marker throws)
        std::string late_error;
        // ... and what the walks end in instead when the scan leaves exactly one whole byte unread: the reference's reader
        // then resumes one byte INTO the terminating marker (DeviceBatch::plan_swallowed_terminator has the mechanism)
        int swallow_status = JPGPU_OK, swallow_detail = 0;
        std::string swallow_error;
        int dri_at_scan = 0;
        size_t rsts_in_scan = 0;  // RSTn markers between the scan header and the scan's terminating marker
        std::vector<Piece> pieces;
        int job = -1;  // scan job inside batch_
        std::string dht;  // the rewritten DHT segment (marker, length, tables)
        uint64_t entropy_off = 0, entropy_len = 0;
        bool by_subsequence = false;  // DRI = 0 scan: transcoded by subsequence (KTS), bytes in d_sout_
        uint64_t out_len = 0;
    };
    int fail(int status, const std::string &msg);
    int hip_fail(hipError_t e, const char *what);
    void plan_file(Plan &p, const uint8_t *data, size_t len, bool strip, bool swallow_terminator = false);
    size_t scan_end(const uint8_t *entropy, size_t len);
    const uint8_t *end_key_ = nullptr;
    size_t end_len_ = 0, end_val_ = 0;
    std::string end_rsts_;

    jpgpu_ctx *ctx_;
    DeviceBatch batch_;
    std::vector<Plan> plans_;
    std::vector<uint32_t> scan_ids_;   // jobs that are transcoded, one lane per restart interval ...
    std::vector<HuffWork> work_;
    std::vector<uint32_t> sub_scan_ids_;  // ... and the DRI = 0 jobs transcoded by subsequence
    std::vector<HuffWork> sub_work_;
    std::vector<uint32_t> h_hist_;     // [jobs][8][256]
    bool ran_ = false;
    bool most_optimal_ = false;
    float last_ms_ = 0;
    hipEvent_t ev0_ = nullptr, ev1_ = nullptr;
    DevBuffer d_work_, d_scan_ids_, d_hist_, d_enc_, d_sizes_, d_offsets_, d_base_, d_totals_, d_out_;
    // subsequence path: bit counts / offsets per subsequence, raw (unstuffed) bits, the encoder's stuffing stage
    DevBuffer d_sub_work_, d_sub_scan_ids_, d_sub_bits_, d_sub_bitoff_, d_sub_totals_, d_scan_raw_off_, d_raw_, d_simages_, d_swork_chunk_,
        d_chunk_ff_, d_sout_, d_sout_len_;
};

}  // namespace jpgpu
