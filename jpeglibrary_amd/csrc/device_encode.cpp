// jpeglibrary_amd/csrc/device_encode.cpp -- device-resident batch ENCODE: what JpegEncoder.Encode() does for a set of images
// (ref: JpegEncoder.cs:255-291 with the call sequence of apps/JpegEncode/EncodeAction.cs:38-63, optimizeCoding = false).
// The host writes the marker segments (SOI, DQT, SOF0, DHT, SOS -- the reference's JpegWriter calls, byte for byte);
// the per-block arithmetic and the entropy coding run in encode_kernels.hip.
#include "device_encode.h"
#include "device_optimize.h"

#include <string.h>

#include <algorithm>

namespace jpgpu {

// JPGPU_ENC_NO_FUSED (measurement): E1 as its two kernels for every image
static bool enc_fused_enabled() {
    static const bool on = getenv("JPGPU_ENC_NO_FUSED") == nullptr;
    return on;
}
// JPGPU_ENC_TWO_PASS_EMIT (measurement): E2 and E3 as their two kernels for every upload (bits_emit_kernel never used)
static bool enc_fused_emit_enabled() {
    static const bool on = getenv("JPGPU_ENC_TWO_PASS_EMIT") == nullptr;
    return on;
}


namespace {
uint64_t align_up64(uint64_t v, uint64_t a) { return (v + a - 1) / a * a; }

// ref: JpegStandardQuantizationTable.cs:12-34 (zig-zag order)
const uint16_t kStdLum[64] = {16, 11, 12, 14, 12, 10, 16, 14, 13, 14, 18, 17, 16, 19, 24, 40, 26, 24, 22, 22, 24, 49,
                              35, 37, 29, 40, 58, 51, 61, 60, 57, 51, 56, 55, 64, 72, 92, 78, 64, 68, 87, 69, 55, 56,
                              80, 109, 81, 87, 95, 98, 103, 104, 103, 62, 77, 113, 121, 112, 100, 120, 92, 101, 103, 99};
const uint16_t kStdChr[64] = {17, 18, 18, 24, 21, 24, 47, 26, 26, 47, 99, 66, 56, 66, 99, 99, 99, 99, 99, 99, 99, 99,
                              99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99,
                              99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99};
// ref: JpegStandardHuffmanEncodingTable.cs:14-83
const uint8_t kDcLumLen[16] = {0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0};
const uint8_t kDcChrLen[16] = {0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0};
const uint8_t kDcVal[12] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11};
const uint8_t kAcLumLen[16] = {0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 125};
const uint8_t kAcLumVal[162] = {
    0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71, 0x14, 0x32, 0x81,
    0x91, 0xa1, 0x08, 0x23, 0x42, 0xb1, 0xc1, 0x15, 0x52, 0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72, 0x82, 0x09, 0x0a, 0x16, 0x17, 0x18,
    0x19, 0x1a, 0x25, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x34, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48,
    0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75,
    0x76, 0x77, 0x78, 0x79, 0x7a, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99,
    0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3,
    0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2, 0xe3, 0xe4, 0xe5,
    0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};
const uint8_t kAcChrLen[16] = {0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 119};
const uint8_t kAcChrVal[162] = {
    0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71, 0x13, 0x22, 0x32, 0x81, 0x08,
    0x14, 0x42, 0x91, 0xa1, 0xb1, 0xc1, 0x09, 0x23, 0x33, 0x52, 0xf0, 0x15, 0x62, 0x72, 0xd1, 0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25,
    0xf1, 0x17, 0x18, 0x19, 0x1a, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47,
    0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74,
    0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x82, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97,
    0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba,
    0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe2, 0xe3, 0xe4,
    0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};

struct StdTable {
    const uint8_t *lengths, *values;
    int count;
};
const StdTable kStd[4] = {{kDcLumLen, kDcVal, 12}, {kAcLumLen, kAcLumVal, 162}, {kDcChrLen, kDcVal, 12}, {kAcChrLen, kAcChrVal, 162}};

// ref: JpegStandardHuffmanEncodingTable.BuildCanonicalCode (:85-131) + JpegHuffmanEncodingTable.GetCode (:94-100)
void build_enc_table(const StdTable &s, EncHuffTable *t) {
    memset(t, 0, sizeof *t);
    uint8_t remaining[16], len_of[256];
    memcpy(remaining, s.lengths, 16);
    int pos = 0, current = 1;
    for (int i = 0; i < s.count; i++) {
        while (remaining[pos] == 0) {
            pos++;
            current++;
        }
        remaining[pos]--;
        len_of[i] = (uint8_t)current;
    }
    uint16_t bit_code = 0;
    int bit_count = len_of[0];
    t->code[s.values[0]] = 0;
    t->len[s.values[0]] = len_of[0];
    for (int i = 1; i < s.count; i++) {
        if (len_of[i] > bit_count) {
            bit_code++;
            bit_code = (uint16_t)(bit_code << (len_of[i] - bit_count));
            bit_count = len_of[i];
        } else {
            ++bit_code;
        }
        t->code[s.values[i]] = bit_code;
        t->len[s.values[i]] = len_of[i];
    }
}

// ref: JpegStandardQuantizationTable.ScaleByQuality (:64-87)
void scale_by_quality(const uint16_t *src, int quality, uint16_t *dst) {
    const int scale = quality < 50 ? 5000 / quality : 200 - (quality * 2);
    for (int i = 0; i < 64; i++) {
        int x = src[i];
        x = ((x * scale) + 50) / 100;
        dst[i] = (uint16_t)std::min(std::max(x, 1), 255);
    }
}

void put_marker(std::vector<uint8_t> &o, uint8_t m) {  // JpegWriter.WriteMarker (JpegWriter.cs:289-303)
    o.push_back(0xFF);
    o.push_back(m);
}
void put_length(std::vector<uint8_t> &o, uint16_t len) {  // JpegWriter.WriteLength (:309-321): length + 2
    const uint16_t v = (uint16_t)(len + 2);
    o.push_back((uint8_t)(v >> 8));
    o.push_back((uint8_t)v);
}
}  // namespace

// Fix() factors of apps/JpegEncode/JpegRgbToYCbCrConverter.cs:38-56 (float32 arithmetic like the reference's Fix, :59-62)
void rgb_ycc_factors(int32_t out[8]) {
    auto fix = [](float x) { return (int32_t)((x * (float)(1L << 16)) + 0.5F); };
    out[0] = fix(0.299F);
    out[1] = fix(0.587F);
    out[2] = fix(0.114F);
    out[3] = fix(0.168735892F);
    out[4] = fix(0.331264108F);
    out[5] = fix(0.5F);
    out[6] = fix(0.418687589F);
    out[7] = fix(0.081312411F);
}

EncodeBatch::~EncodeBatch() {
    for (hipEvent_t ev : ev_)
        if (ev) (void)hipEventDestroy(ev);
    for (DevBuffer *b : {&d_samples_, &d_pixels_, &d_images_, &d_tables_, &d_work_mcu_, &d_work_blk_, &d_work_stat_, &d_work_chunk_, &d_coefs_, &d_bits_, &d_bit_off_,
                         &d_raw_bits_, &d_raw_, &d_marks_, &d_chunk_ff_, &d_out_, &d_out_len_, &d_hist_, &d_headers_, &d_chain_, &d_work_order_})
        b->release();
}

int EncodeBatch::fail(int status, const std::string &msg) {
    ctx_->last_error = msg;
    return status;
}
int EncodeBatch::hip_fail(hipError_t e, const char *what) {
    ctx_->last_error = std::string(what) + ": " + hipGetErrorString(e);
    return e == hipErrorOutOfMemory ? JPGPU_ERR_OUT_OF_MEMORY : JPGPU_ERR_DEVICE;
}

int EncodeBatch::upload(const uint8_t *const *pixels, const jpgpu_encode_params *params, int n) {
    if (n < 0 || (n > 0 && (!pixels || !params))) return fail(JPGPU_ERR_ARGUMENT, "jpgpu_encode_upload: null argument");
    hipError_t e = hipSetDevice(ctx_->device);
    if (e != hipSuccess) return hip_fail(e, "hipSetDevice");
    images_.assign((size_t)n, DevEncImage());
    headers_.assign((size_t)n, std::vector<uint8_t>());
    headers_pre_.assign((size_t)n, std::vector<uint8_t>());
    headers_post_.assign((size_t)n, std::vector<uint8_t>());
    optimized_.clear();
    most_optimal_.assign((size_t)n, 0);
    status_.assign((size_t)n, JPGPU_OK);
    encoded_ = false;
    any_restart_ = false;
    emit_hint_ = 0;
    std::vector<EncWork> work_mcu, work_blk, work_stat;  // per 128 MCUs | per 256 units (blocks or restart intervals) | per 256 blocks
    uint64_t px_off = 0, coef_off = 0, smp_off = 0;
    for (int i = 0; i < n; i++) {
        const jpgpu_encode_params &p = params[i];
        // argument checks of the reference's setters (JpegEncoder.cs:175-184, EncodeAction.cs:19-22)
        if (p.quality <= 0 || p.quality > 100) return fail(JPGPU_ERR_ARGUMENT, "Specified argument was out of the range of valid values. (Parameter 'quality')");
        if ((p.luma_h != 1 && p.luma_h != 2 && p.luma_h != 4) || (p.luma_v != 1 && p.luma_v != 2 && p.luma_v != 4))
            return fail(JPGPU_ERR_ARGUMENT, "Subsampling factor can only be 1, 2 or 4.");
        if (p.width <= 0 || p.height <= 0 || p.width > 65535 || p.height > 65535) return fail(JPGPU_ERR_ARGUMENT, "image dimensions out of range");
        if (p.components != 1 && p.components != 3) return fail(JPGPU_ERR_NOT_SUPPORTED, "1 or 3 components are supported.");
        if (p.input_rgb && p.components != 3) return fail(JPGPU_ERR_ARGUMENT, "RGB input needs 3 components.");
        if (p.input_rgb < 0 || p.input_rgb > 2) return fail(JPGPU_ERR_ARGUMENT, "input_rgb is 0 (samples), 1 (R,G,B) or 2 (R,G,B,A).");
        if (p.restart_interval < 0 || p.restart_interval > 65535) return fail(JPGPU_ERR_ARGUMENT, "restart interval out of range (0..65535)");
        DevEncImage &im = images_[i];
        memset(&im, 0, sizeof im);
        im.px_off = px_off;
        im.coef_off = coef_off;
        im.width = (uint32_t)p.width;
        im.height = (uint32_t)p.height;
        im.in_components = p.input_rgb == 2 ? 4u : (uint32_t)p.components;  // (2: Rgba32 pixels, the alpha byte stepped over)
        im.components = (uint32_t)p.components;
        im.luma_h = (uint32_t)p.luma_h;
        im.luma_v = (uint32_t)p.luma_v;
        im.mcus_per_line = (im.width + 8 * im.luma_h - 1) / (8 * im.luma_h);
        im.mcus_per_column = (im.height + 8 * im.luma_v - 1) / (8 * im.luma_v);
        im.bpm = im.luma_h * im.luma_v + (im.components == 3 ? 2 : 0);
        im.total_blocks = im.mcus_per_line * im.mcus_per_column * im.bpm;
        im.input_rgb = p.input_rgb ? 1 : 0;
        // restart intervals (an extension, see jpgpu_encode_params): block_bits / emit then take one lane per INTERVAL
        im.restart_interval = (uint32_t)p.restart_interval;
        any_restart_ = any_restart_ || p.restart_interval != 0;
        im.n_units = im.restart_interval ? (im.mcus_per_line * im.mcus_per_column + im.restart_interval - 1) / im.restart_interval : im.total_blocks;
        if (p.optimize_coding) {
            most_optimal_[i] = p.optimize_coding == 2;  // JpegEncoder.MostOptimalCoding
            optimized_.push_back(i);
            im.table_base = 4u * (uint32_t)optimized_.size();  // tables 0..3 are the standard ones
        }
        rgb_ycc_factors(im.r2y);
        scale_by_quality(kStdLum, p.quality, im.quant[0]);
        scale_by_quality(kStdChr, p.quality, im.quant[1]);
        px_off = align_up64(px_off + (uint64_t)im.width * im.height * im.in_components, 256);
        coef_off += im.total_blocks;
        const uint32_t total_mcus = im.mcus_per_line * im.mcus_per_column;
        if ((smp_off >> 8) > 0xFFFFFFFFull) return fail(JPGPU_ERR_OUT_OF_MEMORY, "sample buffer beyond 1 TiB");
        im.smp_off_256 = (uint32_t)(smp_off >> 8);
        // (an image fdct_fused_kernel takes has no gathered samples: 4.2 GB less per 256 x 4K)
        if (!enc_fused_enabled() || enc_image_fused_shape(im) == 0)
            smp_off = align_up64(smp_off + (uint64_t)total_mcus * enc_sample_bytes_per_mcu(im.luma_h, im.luma_v, im.components), 256);
        for (uint32_t f = 0; f < total_mcus; f += kEncMcusPerWg) work_mcu.push_back({(uint32_t)i, f});
        im.work_first = (uint32_t)work_blk.size();
        for (uint32_t f = 0; f < im.n_units; f += 256) work_blk.push_back({(uint32_t)i, f});
        if (p.optimize_coding)
            for (uint32_t f = 0; f < im.total_blocks; f += 256) work_stat.push_back({(uint32_t)i, f});

        // ---- marker segments, in the order Encode() writes them (JpegEncoder.cs:261-280)
        std::vector<uint8_t> &h = headers_[i];
        const int ncomp = (int)im.components;
        put_marker(h, 0xD8);
        put_marker(h, 0xDB);  // WriteQuantizationTables (:305-335): ONE segment, tables in SetQuantizationTable order
        put_length(h, (uint16_t)(2 * 65));
        for (int t = 0; t < 2; t++) {
            h.push_back((uint8_t)t);  // precision 0 << 4 | identifier
            for (int k = 0; k < 64; k++) h.push_back((uint8_t)im.quant[t][k]);
        }
        if (im.restart_interval) {  // DRI in front of SOF0: the reference's decoder latches it there even without Identify()
            put_marker(h, 0xDD);
            put_length(h, 2);
            h.push_back((uint8_t)(im.restart_interval >> 8));
            h.push_back((uint8_t)im.restart_interval);
        }
        put_marker(h, 0xC0);  // WriteStartOfFrame (:353-386)
        put_length(h, (uint16_t)(6 + 3 * ncomp));
        h.push_back(8);
        h.push_back((uint8_t)(im.height >> 8));
        h.push_back((uint8_t)im.height);
        h.push_back((uint8_t)(im.width >> 8));
        h.push_back((uint8_t)im.width);
        h.push_back((uint8_t)ncomp);
        for (int c = 0; c < ncomp; c++) {
            h.push_back((uint8_t)(c + 1));
            h.push_back(c == 0 ? (uint8_t)((im.luma_h << 4) | im.luma_v) : (uint8_t)0x11);
            h.push_back(c == 0 ? 0 : 1);
        }
        headers_pre_[i] = h;
        put_marker(h, 0xC4);  // WriteHuffmanTables (:336-352): ONE segment, tables in SetHuffmanTable order
        int total = 0;
        for (int t = 0; t < 4; t++) total += 1 + 16 + kStd[t].count;
        put_length(h, (uint16_t)total);
        for (int t = 0; t < 4; t++) {
            h.push_back((uint8_t)(((t & 1) << 4) | (t >> 1)));
            for (int l = 0; l < 16; l++) h.push_back(kStd[t].lengths[l]);
            for (int k = 0; k < kStd[t].count; k++) h.push_back(kStd[t].values[k]);
        }
        const size_t sos_at = h.size();
        put_marker(h, 0xDA);  // WriteStartOfScan (:387-413)
        put_length(h, (uint16_t)(1 + 2 * ncomp + 3));
        h.push_back((uint8_t)ncomp);
        for (int c = 0; c < ncomp; c++) {
            h.push_back((uint8_t)(c + 1));
            h.push_back(c == 0 ? 0x00 : 0x11);
        }
        h.push_back(0);
        h.push_back(63);
        h.push_back(0);
        headers_post_[i].assign(h.begin() + (long)sos_at, h.end());
        im.header_len = (uint32_t)h.size();
        // optimizeCoding: the DHT is written once the statistics are in; reserve the largest it can be (4 x 256 symbols)
        if (im.table_base != 0) im.header_len = (uint32_t)(headers_pre_[i].size() + 4 + 4 * (1 + 16 + 256) + headers_post_[i].size());
    }
    total_blocks_ = coef_off;
    n_work_mcu_ = (int)work_mcu.size();
    n_work_blk_ = (int)work_blk.size();
    n_work_stat_ = (int)work_stat.size();

    // bits_emit_kernel's issue order: the work list's entries by (place inside the image, image)
    std::vector<uint32_t> work_order(work_blk.size());
    for (size_t k = 0; k < work_order.size(); k++) work_order[k] = (uint32_t)k;
    std::stable_sort(work_order.begin(), work_order.end(), [&](uint32_t a, uint32_t b) { return work_blk[a].first < work_blk[b].first; });

    std::vector<EncHuffTable> tables(4 + 4 * optimized_.size());
    memset(tables.data(), 0, tables.size() * sizeof(EncHuffTable));
    for (int t = 0; t < 4; t++) build_enc_table(kStd[t], &tables[t]);

    struct Up {
        DevBuffer *buf;
        const void *src;
        size_t bytes, reserve;
    };
    const Up ups[] = {
        {&d_tables_, tables.data(), tables.size() * sizeof(EncHuffTable), 0},
        {&d_hist_, nullptr, 0, (size_t)n * 4 * 256 * sizeof(uint32_t) + 256},
        {&d_work_mcu_, work_mcu.data(), work_mcu.size() * sizeof(EncWork), 0},
        {&d_work_blk_, work_blk.data(), work_blk.size() * sizeof(EncWork), 0},
        {&d_work_order_, work_order.data(), work_order.size() * sizeof(uint32_t), 16},
        {&d_work_stat_, work_stat.data(), work_stat.size() * sizeof(EncWork), 16},
        {&d_pixels_, nullptr, 0, (size_t)px_off + 256},
        {&d_coefs_, nullptr, 0, (size_t)total_blocks_ * 128 + 256},
        {&d_samples_, nullptr, 0, (size_t)smp_off + 256},
        {&d_bits_, nullptr, 0, (size_t)total_blocks_ * sizeof(uint32_t) + 256},
        {&d_bit_off_, nullptr, 0, work_blk.size() * (sizeof(uint64_t) + sizeof(uint32_t)) + 512},  // per workgroup: base (u64) | bits (u32)
        {&d_raw_bits_, nullptr, 0, (size_t)n * sizeof(uint64_t) + 256},
        {&d_out_len_, nullptr, 0, (size_t)n * sizeof(uint64_t) + 256},
        {&d_images_, nullptr, 0, (size_t)n * sizeof(DevEncImage) + 256},
    };
    for (const Up &u : ups) {
        e = u.buf->reserve(std::max(u.bytes, u.reserve));
        if (e != hipSuccess) return hip_fail(e, "hipMalloc");
        if (u.bytes) {
            e = hipMemcpyAsync(u.buf->ptr, u.src, u.bytes, hipMemcpyHostToDevice, ctx_->stream);
            if (e != hipSuccess) return hip_fail(e, "hipMemcpyAsync");
        }
    }
    for (int i = 0; i < n; i++) {
        const DevEncImage &im = images_[i];
        e = hipMemcpyAsync((uint8_t *)d_pixels_.ptr + im.px_off, pixels[i], (size_t)im.width * im.height * im.in_components, hipMemcpyHostToDevice,
                           ctx_->stream);
        if (e != hipSuccess) return hip_fail(e, "hipMemcpyAsync(pixels)");
    }
    e = hipStreamSynchronize(ctx_->stream);
    return e == hipSuccess ? JPGPU_OK : hip_fail(e, "hipStreamSynchronize(upload)");
}

int EncodeBatch::set_quantization_table(int i, int identifier, const uint16_t *zigzag64) {
    if (i < 0 || i >= (int)images_.size() || !zigzag64) return fail(JPGPU_ERR_ARGUMENT, "jpgpu_encoder_set_quantization_table: bad argument");
    if (identifier != 0 && identifier != 1) return fail(JPGPU_ERR_NOT_SUPPORTED, "Quantization table identifiers 0 and 1 are supported.");
    for (int k = 0; k < 64; k++)
        if (zigzag64[k] == 0 || zigzag64[k] > 255) return fail(JPGPU_ERR_ARGUMENT, "Quantization table elements must be in 1..255 (element precision 0).");
    DevEncImage &im = images_[(size_t)i];
    // the DQT segment upload() wrote: SOI, FF DB, length, then per table one identifier byte + 64 elements
    const size_t at = 2 + 2 + 2 + (size_t)identifier * 65 + 1;
    for (int k = 0; k < 64; k++) {
        im.quant[identifier][k] = zigzag64[k];
        headers_[(size_t)i][at + (size_t)k] = (uint8_t)zigzag64[k];
        headers_pre_[(size_t)i][at + (size_t)k] = (uint8_t)zigzag64[k];
    }
    encoded_ = false;
    return JPGPU_OK;
}

int EncodeBatch::encode() {
    hipError_t e = hipSetDevice(ctx_->device);
    if (e != hipSuccess) return hip_fail(e, "hipSetDevice");
    const int n = (int)images_.size();
    if (n == 0) return JPGPU_OK;
    for (hipEvent_t &ev : ev_)
        if (!ev && (e = hipEventCreate(&ev)) != hipSuccess) return hip_fail(e, "hipEventCreate");
    // image descriptors (raw / output offsets are filled in once the bit counts are known)
    e = hipMemcpyAsync(d_images_.ptr, images_.data(), (size_t)n * sizeof(DevEncImage), hipMemcpyHostToDevice, ctx_->stream);
    if (e != hipSuccess) return hip_fail(e, "hipMemcpyAsync(images)");
    size_t max_record = 0;
    uint32_t fused_shapes = 0;
    bool any_other = false;
    const bool no_fused = !enc_fused_enabled();
    for (const DevEncImage &im : images_) {
        max_record = std::max(max_record, enc_sample_bytes_per_mcu(im.luma_h, im.luma_v, im.components));
        const int shape = no_fused ? 0 : enc_image_fused_shape(im);
        if (shape != 0) fused_shapes |= 1u << shape;
        else any_other = true;
    }
    (void)hipEventRecord(ev_[0], ctx_->stream);
    e = launch_fdct_quant(ctx_->stream, (const uint8_t *)d_pixels_.ptr, (const DevEncImage *)d_images_.ptr, (const EncWork *)d_work_mcu_.ptr, n_work_mcu_,
                          (uint8_t *)d_samples_.ptr, (int16_t *)d_coefs_.ptr, max_record, fused_shapes, any_other);
    if (e != hipSuccess) return hip_fail(e, "fdct_quant_kernel");
    if (!optimized_.empty()) {
        // optimizeCoding: BuildHuffmanTables (:491-550) -- statistics on the device, JpegHuffmanEncodingTableBuilder.Build on the host
        e = hipMemsetAsync(d_hist_.ptr, 0, (size_t)n * 4 * 256 * sizeof(uint32_t), ctx_->stream);
        if (e != hipSuccess) return hip_fail(e, "hipMemsetAsync(statistics)");
        e = launch_block_stats(ctx_->stream, (const DevEncImage *)d_images_.ptr, (const EncWork *)d_work_stat_.ptr, n_work_stat_,
                               (const int16_t *)d_coefs_.ptr, (uint32_t *)d_hist_.ptr);
        if (e != hipSuccess) return hip_fail(e, "block_stats_kernel");
        std::vector<uint32_t> hist((size_t)n * 4 * 256);
        e = hipMemcpyAsync(hist.data(), d_hist_.ptr, hist.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx_->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx_->stream);
        if (e != hipSuccess) return hip_fail(e, "hipMemcpy(statistics)");
        std::vector<EncHuffTable> built(4 * optimized_.size());
        memset(built.data(), 0, built.size() * sizeof(EncHuffTable));
        for (size_t k = 0; k < optimized_.size(); k++) {
            const int i = optimized_[k];
            std::vector<uint8_t> &h = headers_[i];
            h = headers_pre_[i];
            std::vector<uint8_t> body;
            status_[i] = JPGPU_OK;
            for (int t = 0; t < 4; t++) {  // every builder of the collection, in SetHuffmanTable order: DC0, AC0, DC1, AC1
                std::vector<OptimalCode> codes;
                if (!build_optimal_table(&hist[((size_t)i * 4 + t) * 256], &codes, most_optimal_[i] != 0)) {
                    status_[i] = JPGPU_ERR_INVALID_OPERATION;  // "No symbol is recorded." (a single-component image: empty chrominance builders)
                    break;
                }
                EncHuffTable &et = built[4 * k + t];
                for (int sym = 0; sym < 256; sym++) {  // GetCode of a symbol without a code: the table's first code
                    et.code[sym] = codes[0].code;
                    et.len[sym] = codes[0].length;
                }
                for (const OptimalCode &c : codes) {
                    et.code[c.symbol] = c.code;
                    et.len[c.symbol] = c.length;
                }
                body.push_back((uint8_t)(((t & 1) << 4) | (t >> 1)));
                for (int l = 1; l <= 16; l++) {
                    int count = 0;
                    for (const OptimalCode &c : codes) count += c.length == l;
                    body.push_back((uint8_t)count);
                }
                for (const OptimalCode &c : codes) body.push_back(c.symbol);
            }
            put_marker(h, 0xC4);
            put_length(h, (uint16_t)body.size());
            h.insert(h.end(), body.begin(), body.end());
            h.insert(h.end(), headers_post_[i].begin(), headers_post_[i].end());
            images_[i].header_len = (uint32_t)h.size();
        }
        e = hipMemcpyAsync((EncHuffTable *)d_tables_.ptr + 4, built.data(), built.size() * sizeof(EncHuffTable), hipMemcpyHostToDevice, ctx_->stream);
        if (e != hipSuccess) return hip_fail(e, "hipMemcpyAsync(built tables)");
    }
    (void)hipEventRecord(ev_[1], ctx_->stream);
    std::vector<uint64_t> raw_bits((size_t)n);
    // E2 + E3 as one pass (bits_emit_kernel): the raw streams are placed BEFORE their sizes are known -- a slot of the worst case
    // per image (every coefficient a 32-bit field: 256 bytes per block), of which only the stream itself is ever touched
    // The chains are per image: a batch of ONE image is one chain, as long as the image has workgroups, and runs at the speed the
    // look-back travels (one 8192 x 8192 canvas: 0.13 ms against the two kernels' 0.107); from two images on the one pass wins.
    bool fused_emit = false;
    bool try_fused = enc_fused_emit_enabled() && !any_restart_ && emit_hint_ != 0xFFFFFFFFu && n >= 2;
    uint64_t slot = 0;
    const size_t chain_bytes = 256 + enc_chain_bytes(n_work_blk_);  // control words | one record per workgroup
    if (try_fused) {
        for (DevEncImage &im : images_) {
            im.raw_off = slot;
            slot = align_up64(slot + (uint64_t)im.total_blocks * 256 + 64, 256);
        }
        // The worst case is ~50 x the streams themselves (51 GB per 1024 x 4K 4:2:0 where the two kernels size ~1 GB from the counts)
        // and d_raw_ never shrinks: the slots may take a third of what the device has free, no more -- the finished streams, the
        // chunk counts and the next upload's buffers are reserved behind them, and a batch that used to encode must not fail there
        // because this one grabbed the rest.  Beyond that the two kernels run.
        size_t free_b = 0, total_b = 0;
        e = (size_t)slot + 256 <= d_raw_.cap ? hipSuccess : hipMemGetInfo(&free_b, &total_b);
        if (e == hipSuccess && (size_t)slot + 256 > d_raw_.cap && (size_t)slot + 256 > free_b / 3) e = hipErrorOutOfMemory;
        if (e == hipSuccess) e = d_raw_.reserve((size_t)slot + 256);
        if (e == hipSuccess) e = d_chain_.reserve(chain_bytes + 256);
        if (e != hipSuccess) {  // no room for the worst-case slots: the two kernels size the raw streams from the counts
            (void)hipGetLastError();
            emit_hint_ = 0xFFFFFFFFu;
            try_fused = false;
        }
    }
    if (try_fused) {
        one_pass_emits_++;
        e = hipMemsetAsync(d_chain_.ptr, 0, chain_bytes, ctx_->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(d_images_.ptr, images_.data(), (size_t)n * sizeof(DevEncImage), hipMemcpyHostToDevice, ctx_->stream);
        if (e != hipSuccess) return hip_fail(e, "hipMemcpyAsync(images)");
        e = launch_bits_emit(ctx_->stream, (const DevEncImage *)d_images_.ptr, (const EncWork *)d_work_blk_.ptr, n_work_blk_, (const EncHuffTable *)d_tables_.ptr,
                             (const int16_t *)d_coefs_.ptr, (uint8_t *)d_chain_.ptr + 256, (uint32_t *)d_chain_.ptr, (uint8_t *)d_raw_.ptr, (uint64_t *)d_raw_bits_.ptr,
                             emit_hint_ ? emit_hint_ : 4096u, (const uint32_t *)d_work_order_.ptr);
        if (e != hipSuccess) return hip_fail(e, "bits_emit_kernel");
        (void)hipEventRecord(ev_[2], ctx_->stream);
        uint32_t ctl[2] = {0, 0};
        e = hipMemcpyAsync(raw_bits.data(), d_raw_bits_.ptr, (size_t)n * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx_->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(ctl, d_chain_.ptr, sizeof ctl, hipMemcpyDeviceToHost, ctx_->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx_->stream);
        if (e != hipSuccess) return hip_fail(e, "hipMemcpy(bit counts)");
        fused_emit = ctl[1] == 0;
        if (!fused_emit) {  // a stretch beyond the LDS buffer, or a wait that ran out: this upload takes the two kernels from here on
            emit_hint_ = 0xFFFFFFFFu;
            fused_emit_fallbacks_++;
        }
    }
    if (!fused_emit) {
    uint32_t *wg_bits = reinterpret_cast<uint32_t *>((uint8_t *)d_bit_off_.ptr + (((size_t)n_work_blk_ * sizeof(uint64_t) + 255) & ~(size_t)255));
    e = launch_block_bits(ctx_->stream, (const DevEncImage *)d_images_.ptr, (const EncWork *)d_work_blk_.ptr, n_work_blk_,
                          (const EncHuffTable *)d_tables_.ptr, (const int16_t *)d_coefs_.ptr, (uint32_t *)d_bits_.ptr, n, wg_bits, (uint64_t *)d_bit_off_.ptr,
                          (uint64_t *)d_raw_bits_.ptr);
    if (e != hipSuccess) return hip_fail(e, "block_bits_kernel");
    (void)hipEventRecord(ev_[2], ctx_->stream);
    // the sizes of the raw and finished streams depend on the data: one host round trip
    e = hipMemcpyAsync(raw_bits.data(), d_raw_bits_.ptr, (size_t)n * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx_->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx_->stream);
    if (e != hipSuccess) return hip_fail(e, "hipMemcpy(bit counts)");
    }
    std::vector<EncWork> work_chunk;
    uint64_t raw_off = 0, out_off = 0;
    uint32_t chunk_off = 0;
    // E3's LDS buffer: twice the largest average stretch a workgroup (256 blocks or restart intervals) of any image writes,
    // in steps of 4 KB -- the smaller it is the more workgroups share a CU; a workgroup whose stretch does not fit falls back
    // to atomics in HBM (launch_emit clamps to 8-32 KB).  Images whose AVERAGE stretch is beyond 32 KB (restart intervals of
    // several MCUs: every lane then writes a stretch of its own and the atomics do not collide) do not count; a batch of
    // nothing else asks for no LDS at all.
    uint64_t emit_words = 0;
    for (int i = 0; i < n; i++) {
        DevEncImage &im = images_[i];
        const uint64_t raw_len = (raw_bits[i] + 7) / 8;
        const uint64_t wgs = ((uint64_t)im.n_units + 255) / 256;
        const uint64_t avg_words = raw_bits[i] / 32 / std::max<uint64_t>(wgs, 1) + 1;
        if (avg_words <= kEmitLdsWordsMax) emit_words = std::max<uint64_t>(emit_words, 2 * avg_words);
        if (!fused_emit) im.raw_off = raw_off;  // (the one-pass form has placed and written the raw streams already)
        im.out_off = out_off;
        im.chunk_off = chunk_off;
        const uint32_t chunks = (uint32_t)((raw_len + kEncStuffChunk - 1) / kEncStuffChunk);
        for (uint32_t c = 0; c < std::max(chunks, 1u); c++) work_chunk.push_back({(uint32_t)i, c});
        chunk_off += std::max(chunks, 1u);
        raw_off = align_up64(raw_off + raw_len + 64, 256);
        // every byte may need stuffing; two marker bytes per restart interval
        out_off = align_up64(out_off + im.header_len + 2 * raw_len + 2 + 64 + (im.restart_interval ? 2 * (uint64_t)im.n_units : 0), 256);
    }
    // the next encode() of this upload sizes bits_emit_kernel's LDS buffer like emit_kernel's (no stretch would fit: never again)
    if (emit_hint_ != 0xFFFFFFFFu) emit_hint_ = emit_words == 0 ? 0xFFFFFFFFu : (uint32_t)std::min<uint64_t>((emit_words + 1023) & ~1023ull, kEmitLdsWordsMax);
    n_work_chunk_ = (int)work_chunk.size();
    out_cap_ = out_off;
    const struct {
        DevBuffer *buf;
        size_t bytes;
    } grow[] = {{&d_raw_, fused_emit ? 0 : (size_t)raw_off + 256}, {&d_marks_, fused_emit ? 0 : (size_t)raw_off / 8 + 256}, {&d_out_, (size_t)out_off + 256},
                {&d_chunk_ff_, (size_t)chunk_off * sizeof(uint32_t) + 256}, {&d_work_chunk_, work_chunk.size() * sizeof(EncWork) + 16}};
    for (const auto &g : grow) {
        e = g.buf->reserve(g.bytes);
        if (e != hipSuccess) return hip_fail(e, "hipMalloc");
    }
    if (!fused_emit) {  // (emit_kernel ORs into zeroed words; the one-pass form has one writer per word and no restart marks)
        e = hipMemsetAsync(d_raw_.ptr, 0, (size_t)raw_off + 256, ctx_->stream);
        if (e == hipSuccess) e = hipMemsetAsync(d_marks_.ptr, 0, (size_t)raw_off / 8 + 256, ctx_->stream);
        if (e != hipSuccess) return hip_fail(e, "hipMemsetAsync(raw)");
    }
    // the headers: one upload for the batch, put in front of the streams on the device
    header_bytes_.clear();
    for (int i = 0; i < n; i++) {
        images_[i].hdr_off = (uint32_t)header_bytes_.size();
        header_bytes_.insert(header_bytes_.end(), headers_[i].begin(), headers_[i].end());
    }
    e = d_headers_.reserve(header_bytes_.size() + 16);
    if (e != hipSuccess) return hip_fail(e, "hipMalloc");
    e = hipMemcpyAsync(d_work_chunk_.ptr, work_chunk.data(), work_chunk.size() * sizeof(EncWork), hipMemcpyHostToDevice, ctx_->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_images_.ptr, images_.data(), (size_t)n * sizeof(DevEncImage), hipMemcpyHostToDevice, ctx_->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_headers_.ptr, header_bytes_.data(), header_bytes_.size(), hipMemcpyHostToDevice, ctx_->stream);
    if (e == hipSuccess) e = launch_place_headers(ctx_->stream, (const DevEncImage *)d_images_.ptr, n, (const uint8_t *)d_headers_.ptr, (uint8_t *)d_out_.ptr);
    if (e != hipSuccess) return hip_fail(e, "hipMemcpyAsync(headers)");
    (void)hipEventRecord(ev_[3], ctx_->stream);
    if (!fused_emit) {
        e = launch_emit(ctx_->stream, (const DevEncImage *)d_images_.ptr, (const EncWork *)d_work_blk_.ptr, n_work_blk_, (const EncHuffTable *)d_tables_.ptr,
                        (const int16_t *)d_coefs_.ptr, (const uint32_t *)d_bits_.ptr, (const uint64_t *)d_bit_off_.ptr, (const uint64_t *)d_raw_bits_.ptr,
                        (uint8_t *)d_raw_.ptr, (uint32_t *)d_marks_.ptr, (uint32_t)std::min<uint64_t>((emit_words + 1023) & ~1023ull, kEmitLdsWordsMax));
        if (e != hipSuccess) return hip_fail(e, "emit_kernel");
    }
    (void)hipEventRecord(ev_[4], ctx_->stream);
    e = launch_stuff(ctx_->stream, (const DevEncImage *)d_images_.ptr, (const EncWork *)d_work_chunk_.ptr, n_work_chunk_, (const uint64_t *)d_raw_bits_.ptr,
                     (const uint8_t *)d_raw_.ptr, (const uint32_t *)d_marks_.ptr, (uint32_t *)d_chunk_ff_.ptr, (uint8_t *)d_out_.ptr,
                     (uint64_t *)d_out_len_.ptr);
    if (e != hipSuccess) return hip_fail(e, "stuff kernels");
    (void)hipEventRecord(ev_[5], ctx_->stream);
    out_len_.assign((size_t)n, 0);
    e = hipMemcpyAsync(out_len_.data(), d_out_len_.ptr, (size_t)n * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx_->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx_->stream);
    if (e != hipSuccess) return hip_fail(e, "hipMemcpy(stream lengths)");
    encoded_ = true;
    return JPGPU_OK;
}

int EncodeBatch::stage_ms(float ms[5]) {
    if (!encoded_ || !ev_[5]) return fail(JPGPU_ERR_INVALID_OPERATION, "jpgpu_encoder_stage_ms: no encode has run");
    const int pairs[4][2] = {{0, 1}, {1, 2}, {3, 4}, {4, 5}};  // (the statistics pass of optimizeCoding counts as E1)
    ms[4] = 0;
    for (int k = 0; k < 4; k++) {
        if (hipEventElapsedTime(&ms[k], ev_[pairs[k][0]], ev_[pairs[k][1]]) != hipSuccess) return fail(JPGPU_ERR_DEVICE, "hipEventElapsedTime failed");
        ms[4] += ms[k];
    }
    return JPGPU_OK;
}

int EncodeBatch::encoded_size(int i, size_t *bytes) const {
    if (i < 0 || i >= (int)images_.size() || !bytes || !encoded_) return JPGPU_ERR_ARGUMENT;
    if (status_[i] != JPGPU_OK) {
        ctx_->last_error = "No symbol is recorded.";
        return status_[i];
    }
    *bytes = (size_t)out_len_[i];
    return JPGPU_OK;
}

int EncodeBatch::download(int i, void *dst, size_t cap) {
    if (i < 0 || i >= (int)images_.size() || !dst) return fail(JPGPU_ERR_ARGUMENT, "jpgpu_encode_download: bad argument");
    if (!encoded_) return fail(JPGPU_ERR_INVALID_OPERATION, "Nothing has been encoded yet.");
    if (status_[i] != JPGPU_OK) return fail(status_[i], "No symbol is recorded.");
    if (cap < out_len_[i]) return fail(JPGPU_ERR_ARGUMENT, "Destination buffer is too small.");
    hipError_t e = hipMemcpy(dst, (const uint8_t *)d_out_.ptr + images_[i].out_off, (size_t)out_len_[i], hipMemcpyDeviceToHost);
    return e == hipSuccess ? JPGPU_OK : hip_fail(e, "hipMemcpy(encoded stream)");
}

int EncodeBatch::download_coefficients(int i, int16_t *dst, size_t cap_blocks) {
    if (i < 0 || i >= (int)images_.size() || !dst) return fail(JPGPU_ERR_ARGUMENT, "bad argument");
    if (!encoded_) return fail(JPGPU_ERR_INVALID_OPERATION, "Nothing has been encoded yet.");
    if (cap_blocks < images_[i].total_blocks) return fail(JPGPU_ERR_ARGUMENT, "Destination buffer is too small.");
    const DevEncImage &im = images_[i];
    hipError_t e = hipMemcpy(dst, (const int16_t *)d_coefs_.ptr + im.coef_off * 64, (size_t)im.total_blocks * 128, hipMemcpyDeviceToHost);
    if (e != hipSuccess) return hip_fail(e, "hipMemcpy(coefficients)");
    if (im.table_base != 0) {
        // optimizeCoding: the blocks as BuildHuffmanTables / WritePreparedScanData see them -- luma blocks outside the
        // component's own grid alias the allocator's dummy block (enc_source_block in encode_kernels.hip)
        const uint32_t ny = im.luma_h * im.luma_v, last = (im.mcus_per_line * im.mcus_per_column - 1) * im.bpm + ny - 1;
        for (uint32_t blk = 0; blk < im.total_blocks; blk++) {
            const uint32_t mcu = blk / im.bpm, b = blk - mcu * im.bpm;
            if (b >= ny) continue;
            const uint32_t bx = (mcu % im.mcus_per_line) * im.luma_h + b % im.luma_h, by = (mcu / im.mcus_per_line) * im.luma_v + b / im.luma_h;
            if (bx < (im.width + 7) / 8 && by < (im.height + 7) / 8) continue;
            if (blk != last) memcpy(dst + (size_t)blk * 64, dst + (size_t)last * 64, 128);
        }
    }
    return JPGPU_OK;
}

}  // namespace jpgpu
