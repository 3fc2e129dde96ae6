// jpeglibrary_amd/csrc/device_encode.h -- device-resident batch encode (see device_encode.cpp)
#pragma once
#include <string>
#include <vector>

#include "../../include/jpgpu.h"
#include "device_batch.h"
#include "encode_kernels.h"

namespace jpgpu {

void rgb_ycc_factors(int32_t out[8]);

class EncodeBatch {
  public:
    explicit EncodeBatch(jpgpu_ctx *ctx) : ctx_(ctx) {}
    ~EncodeBatch();
    int upload(const uint8_t *const *pixels, const jpgpu_encode_params *params, int n);  // SetInputReader x n (+ H2D)
    int set_quantization_table(int i, int identifier, const uint16_t *zigzag64);           // SetQuantizationTable of image i
    int encode();                                                                        // JpegEncoder.Encode() x n
    // device time of the last encode() by stage (HIP events on the context's stream): E1 pixels -> quantised blocks, E2 bit counts
    // and offsets, E3 bit emission, E4 byte stuffing; ms[4] = the four together (the host round trips between them excluded)
    int stage_ms(float ms[5]);
    int size() const { return (int)images_.size(); }
    int encoded_size(int i, size_t *bytes) const;
    int image_status(int i) const { return (i >= 0 && i < (int)status_.size()) ? status_[i] : JPGPU_ERR_ARGUMENT; }
    int download(int i, void *dst, size_t cap);
    int download_coefficients(int i, int16_t *dst, size_t cap_blocks);
    uint32_t total_blocks(int i) const { return (i >= 0 && i < (int)images_.size()) ? images_[i].total_blocks : 0; }
    void *output_device(int i, size_t *bytes) const {
        if (i < 0 || i >= (int)images_.size() || !encoded_) return nullptr;
        if (bytes) *bytes = (size_t)out_len_[i];
        return (uint8_t *)d_out_.ptr + images_[i].out_off;
    }

  private:
    int fail(int status, const std::string &msg);
    int hip_fail(hipError_t e, const char *what);
    jpgpu_ctx *ctx_;
    std::vector<DevEncImage> images_;
    std::vector<std::vector<uint8_t>> headers_;        // SOI .. SOS as Encode() writes them (optimizeCoding: rebuilt per encode())
    std::vector<std::vector<uint8_t>> headers_pre_;    // optimizeCoding: SOI, DQT, SOF0 ...
    std::vector<std::vector<uint8_t>> headers_post_;   // ... and SOS; the DHT between them depends on the statistics
    std::vector<int> optimized_;                       // images with optimizeCoding
    std::vector<uint8_t> most_optimal_;                // ... and MostOptimalCoding (package merge)
    std::vector<int> status_;                          // per image: JPGPU_OK or the reference's failure ("No symbol is recorded.")
    DevBuffer d_hist_;
    std::vector<uint64_t> out_len_;
    bool encoded_ = false;
    hipEvent_t ev_[6] = {};
    uint64_t total_blocks_ = 0, out_cap_ = 0;
    int n_work_mcu_ = 0, n_work_blk_ = 0, n_work_stat_ = 0, n_work_chunk_ = 0;
    DevBuffer d_samples_;  // E1a -> E1b: gathered samples, enc_sample_bytes_per_mcu per MCU
    DevBuffer d_pixels_, d_images_, d_tables_, d_work_mcu_, d_work_blk_, d_work_stat_, d_work_chunk_, d_coefs_, d_bits_, d_bit_off_, d_raw_bits_, d_raw_, d_marks_,
        d_chunk_ff_, d_out_, d_out_len_, d_headers_;
    std::vector<uint8_t> header_bytes_;  // every image's SOI..SOS, concatenated (one upload per encode)
    // E2 + E3 as one pass (bits_emit_kernel): restart-free uploads only; the LDS stretch buffer is sized from the previous encode()
    // of this upload (0: not known yet; 0xFFFFFFFF: a stretch did not fit or a wait ran out -- the two-kernel path from then on)
    DevBuffer d_chain_, d_work_order_;
    bool any_restart_ = false;
    uint32_t emit_hint_ = 0;
    int one_pass_emits_ = 0, fused_emit_fallbacks_ = 0;  // encode() calls that issued bits_emit_kernel / of those, the ones that had to issue the two kernels after it

  public:
    void emit_counters(int *one_pass, int *fell_back) const {
        if (one_pass) *one_pass = one_pass_emits_;
        if (fell_back) *fell_back = fused_emit_fallbacks_;
    }
};

}  // namespace jpgpu
