// jpeglibrary_amd/csrc/host_parser.cpp -- host-side marker / table parsing and JpegDecoder state machine.
// See host.h for the map of reference files this mirrors.  No per-block arithmetic happens here.
#include <string.h>

#include <algorithm>

#include "../../include/jpgpu.h"
#include "host.h"

namespace jpgpu {

void throw_invalid_data(const std::string &msg, int detail) { throw DecodeError(JPGPU_ERR_INVALID_DATA, msg, detail); }
void throw_invalid_data_at(int offset, const std::string &msg, int detail) {
    // ref: JpegDecoder.cs:371-375
    throw DecodeError(JPGPU_ERR_INVALID_DATA, "Failed to decode JPEG data at offset " + std::to_string(offset) + ". " + msg, detail);
}
void throw_invalid_operation(const std::string &msg, int detail) { throw DecodeError(JPGPU_ERR_INVALID_OPERATION, msg, detail); }

// ------------------------------------------------------------------------------------------------ MarkerReader

bool MarkerReader::try_read_start_of_image() {
    if (n_ < 2) return false;
    if (p_[0] == kPadding && p_[1] == kSOI) {
        p_ += 2;
        n_ -= 2;
        return true;
    }
    return false;
}

bool MarkerReader::try_read_marker(int *marker) {
    while (n_ >= 2) {
        const uint8_t b1 = p_[0], b2 = p_[1];
        if (b1 == kPadding) {
            if (b2 == kPadding) {  // fill byte
                p_ += 1;
                n_ -= 1;
                continue;
            }
            if (b2 == 0) {  // stuffed FF inside entropy data
                p_ += 2;
                n_ -= 2;
                continue;
            }
            p_ += 2;
            n_ -= 2;
            *marker = b2;
            return true;
        }
        const uint8_t *q = static_cast<const uint8_t *>(memchr(p_, kPadding, n_));
        if (!q) {
            p_ += n_;
            n_ = 0;
            *marker = 0;
            return false;
        }
        n_ -= (size_t)(q - p_);
        p_ = q;
    }
    *marker = 0;
    return false;
}

bool MarkerReader::try_read_length(uint16_t *length) {
    if (n_ < 2) {
        *length = 0;
        return false;
    }
    // The reference computes (ushort)(b0 << 8 | b1 - 2), which binds as (b0 << 8) | (b1 - 2) (JpegReader.cs:174):
    // the true payload length only when the low length byte is >= 2.  Kept bit-for-bit so that files the
    // reference rejects are rejected here too (drop-in behaviour); SURVEY 3.4-10.
    const int32_t v = ((int32_t)p_[0] << 8) | ((int32_t)p_[1] - 2);
    *length = (uint16_t)v;
    p_ += 2;
    n_ -= 2;
    return true;
}

bool MarkerReader::try_read_bytes(int length, const uint8_t **bytes) {
    if (n_ < (size_t)length) return false;
    *bytes = p_;
    p_ += length;
    n_ -= (size_t)length;
    return true;
}

bool MarkerReader::try_advance(int length) {
    if (length < 0 || n_ < (size_t)length) return false;
    p_ += length;
    n_ -= (size_t)length;
    return true;
}

// ------------------------------------------------------------------------------------------------ headers

bool FrameHeader::try_parse(const uint8_t *buf, size_t len, bool metadata_only, FrameHeader *out, int *consumed) {
    *consumed = 0;
    if (len < 6) return false;
    FrameHeader fh;
    fh.num_components = buf[5];
    fh.samples_per_line = (uint16_t)(buf[4] | (buf[3] << 8));
    fh.lines = (uint16_t)(buf[2] | (buf[1] << 8));
    fh.precision = buf[0];
    buf += 6;
    len -= 6;
    *consumed += 6;
    if (len < (size_t)(3 * fh.num_components)) return false;
    if (metadata_only) {
        *consumed += 3 * fh.num_components;
        *out = fh;
        return true;
    }
    fh.components.resize(fh.num_components);
    for (int i = 0; i < fh.num_components; i++) {
        FrameComponent &c = fh.components[i];
        c.identifier = buf[0];
        c.h = (uint8_t)(buf[1] >> 4);
        c.v = (uint8_t)(buf[1] & 0xF);
        c.tq = buf[2];
        buf += 3;
        *consumed += 3;
    }
    *out = fh;
    return true;
}

bool ScanHeader::try_parse(const uint8_t *buf, size_t len, bool metadata_only, ScanHeader *out, int *consumed) {
    *consumed = 0;
    if (len == 0) return false;
    ScanHeader sh;
    sh.num_components = buf[0];
    buf++;
    len--;
    (*consumed)++;
    if (len < (size_t)(2 * sh.num_components + 3)) return false;
    if (!metadata_only) {
        sh.components.resize(sh.num_components);
        for (int i = 0; i < sh.num_components; i++) {
            sh.components[i].selector = buf[0];
            sh.components[i].td = (uint8_t)(buf[1] >> 4);
            sh.components[i].ta = (uint8_t)(buf[1] & 0xF);
            buf += 2;
            *consumed += 2;
        }
    } else {
        buf += 2 * sh.num_components;
        *consumed += 2 * sh.num_components;
    }
    sh.ss = buf[0];
    sh.se = buf[1];
    sh.ah = (uint8_t)(buf[2] >> 4);
    sh.al = (uint8_t)(buf[2] & 0xF);
    *consumed += 3;
    *out = sh;
    return true;
}

bool QuantTable::try_parse(const uint8_t *buf, size_t len, QuantTable *out, int *consumed) {
    *consumed = 0;
    if (len == 0) return false;
    QuantTable t;
    t.precision = (uint8_t)(buf[0] >> 4);
    t.identifier = (uint8_t)(buf[0] & 0xF);
    (*consumed)++;
    buf++;
    len--;
    if (t.precision == 0) {
        if (len < 64) return false;
        for (int i = 0; i < 64; i++) t.elements[i] = buf[i];
        *consumed += 64;
    } else if (t.precision == 1) {
        if (len < 128) return false;
        for (int i = 0; i < 64; i++) t.elements[i] = (uint16_t)(buf[2 * i] << 8 | buf[2 * i + 1]);
        *consumed += 128;
    } else {
        return false;
    }
    *out = t;
    return true;
}

// ------------------------------------------------------------------------------------------------ Huffman tables

bool HuffTable::from_bits_values(uint8_t table_class, uint8_t identifier, const uint8_t bits[16], const uint8_t *values,
                                 int num_values, HuffTable *out) {
    int code_count = 0;
    for (int i = 15; i >= 0; i--) code_count += bits[i];
    if (code_count > 256 || num_values < code_count) return false;

    HuffTable t;
    t.table_class = table_class;
    t.identifier = identifier;
    memcpy(t.bits, bits, 16);
    t.num_values = (uint16_t)code_count;
    memcpy(t.values, values, (size_t)code_count);

    // GenerateSizeTable / GenerateCodeTable (ref: JpegHuffmanDecodingTable.cs:293-337)
    uint8_t huffsize[257] = {};
    uint16_t huffcode[257] = {};
    int k = 0;
    for (int len = 1; len <= 16; len++)
        for (int j = 0; j < bits[len - 1]; j++) huffsize[k++] = (uint8_t)len;
    huffsize[k] = 0;
    if (huffsize[0] != 0) {
        int code = 0, si = huffsize[0];
        k = 0;
        for (;;) {
            do {
                huffcode[k] = (uint16_t)code;
                code++;
                k++;
            } while (huffsize[k] == si);
            if (huffsize[k] == 0) break;
            do {
                code <<= 1;
                si++;
            } while (huffsize[k] != si);
        }
    }
    // Configure (ref: :339-376)
    int p = 0;
    for (int l = 1; l <= 16; l++) {
        if (bits[l - 1] != 0) {
            t.valoffset[l] = (uint8_t)(p - huffcode[p]);
            p += bits[l - 1];
            uint16_t mc = huffcode[p - 1];
            mc = (uint16_t)(mc << (16 - l));
            t.maxcode[l] = (uint16_t)(mc | (uint32_t)((1 << (16 - l)) - 1));
        } else {
            t.maxcode[l] = 0;
        }
    }
    t.valoffset[18] = 0;
    t.maxcode[17] = 0xFFFF;
    // FillByteLookupTable (ref: :378-390)
    p = 0;
    for (int l = 1; l <= 8; l++) {
        for (int i = 0; i < bits[l - 1]; i++, p++) {
            const int free_bits = 8 - l;
            const int code = (uint8_t)(huffcode[p] << free_bits);
            for (int j = 0; j < (1 << free_bits) && code + j < 256; j++) {
                t.la_size[code + j] = (uint8_t)l;
                t.la_symbol[code + j] = t.values[p];
            }
        }
    }
    *out = t;
    return true;
}

bool HuffTable::try_parse(const uint8_t *buf, size_t len, HuffTable *out, int *consumed) {
    *consumed = 0;
    if (len == 0) return false;
    const uint8_t tcth = buf[0];
    (*consumed)++;
    buf++;
    len--;
    if (len < 16) return false;
    int code_count = 0;
    for (int i = 15; i >= 0; i--) code_count += buf[i];
    if (code_count > 256) return false;
    *consumed += 16;
    if (len - 16 < (size_t)code_count) return false;
    *consumed += code_count;
    return from_bits_values((uint8_t)(tcth >> 4), (uint8_t)(tcth & 0xF), buf, buf + 16, code_count, out);
}

bool HuffTable::lookup(int code16, int *size, int *symbol) const {
    const int high8 = code16 >> 8;
    if (la_size[high8] != 0) {
        *size = la_size[high8];
        *symbol = la_symbol[high8];
        return true;
    }
    int s = 9;
    while (code16 > maxcode[s]) s++;
    if (s > 16) return false;
    *size = s;
    *symbol = values[(valoffset[s] + (code16 >> (16 - s))) & 0xFF];
    return true;
}

void HuffTable::to_device(DevHuffTable *out) const {
    memset(out, 0, sizeof *out);
    // A code of length <= kHuffLutBits is decided by the top kHuffLutBits bits alone (maxcode[l] has its low 16-l
    // bits set), so evaluating the reference's Lookup() with the remaining bits all ones gives the entry.
    for (int i = 0; i < kHuffLutSize; i++) {
        const int code16 = (i << (16 - kHuffLutBits)) | ((1 << (16 - kHuffLutBits)) - 1);
        int size = 0, symbol = 0;
        if (lookup(code16, &size, &symbol) && size <= kHuffLutBits) out->lut[i] = (uint16_t)((size << 8) | symbol);
    }
    memcpy(out->maxcode, maxcode, sizeof maxcode);
    memcpy(out->valoffset, valoffset, sizeof valoffset);
    memcpy(out->values, values, sizeof values);
}

// ------------------------------------------------------------------------------------------------ HostDecoder

void HostDecoder::set_input(const uint8_t *data, size_t len) {
    input_ = data;
    input_len_ = len;
    frame_.reset();
    restart_interval_ = 0;
}

const FrameHeader &HostDecoder::frame_header() const {
    if (!frame_) throw_invalid_operation("Call Identify() before this operation.");
    return *frame_;
}

void HostDecoder::set_restart_interval(int v) {
    if ((unsigned)v > 0xFFFFu) throw DecodeError(JPGPU_ERR_ARGUMENT, "Specified argument was out of the range of valid values. (Parameter 'restartInterval')");
    restart_interval_ = (uint16_t)v;
}

int HostDecoder::maximum_horizontal_sampling() const {
    const FrameHeader &fh = frame_header();
    if (fh.components.empty() && fh.num_components) throw_invalid_operation("Operation is not valid due to the current state of the object.");
    int m = 1;
    for (const FrameComponent &c : fh.components) m = std::max(m, (int)c.h);
    return m;
}
int HostDecoder::maximum_vertical_sampling() const {
    const FrameHeader &fh = frame_header();
    if (fh.components.empty() && fh.num_components) throw_invalid_operation("Operation is not valid due to the current state of the object.");
    int m = 1;
    for (const FrameComponent &c : fh.components) m = std::max(m, (int)c.v);
    return m;
}

const HuffTable *HostDecoder::huffman_table(bool is_dc, uint8_t identifier) const {
    const int cls = is_dc ? 0 : 1;
    for (const HuffTable &t : huff_)
        if (t.table_class == cls && t.identifier == identifier) return &t;
    return nullptr;
}
const QuantTable *HostDecoder::quantization_table(uint8_t identifier) const {
    for (const QuantTable &t : quant_)
        if (t.identifier == identifier) return &t;
    return nullptr;
}
void HostDecoder::set_huffman_table(const HuffTable &t) {
    for (HuffTable &e : huff_)
        if (e.table_class == t.table_class && e.identifier == t.identifier) {
            e = t;
            return;
        }
    huff_.push_back(t);
}
void HostDecoder::set_quantization_table(const QuantTable &t) {
    for (QuantTable &e : quant_)
        if (e.identifier == t.identifier) {
            e = t;
            return;
        }
    quant_.push_back(t);
}

void HostDecoder::process_other_marker(MarkerReader &r) {
    uint16_t length;
    if (!r.try_read_length(&length)) throw_invalid_data_at(r.consumed_byte_count(), "Unexpected end of input data when reading segment length.", kDetailBadHeader);
    if (!r.try_advance(length)) throw_invalid_data_at(r.consumed_byte_count(), "Unexpected end of input data reached.", kDetailBadHeader);
}

void HostDecoder::process_frame_header(MarkerReader &r, bool metadata_only, bool override_allowed) {
    uint16_t length;
    const uint8_t *buf;
    if (!r.try_read_length(&length)) throw_invalid_data_at(r.consumed_byte_count(), "Unexpected end of input data when reading segment length.", kDetailBadHeader);
    if (!r.try_read_bytes(length, &buf)) throw_invalid_data_at(r.consumed_byte_count(), "Unexpected end of input data when reading segment content.", kDetailBadHeader);
    FrameHeader fh;
    int consumed;
    if (!FrameHeader::try_parse(buf, length, metadata_only, &fh, &consumed))
        throw_invalid_data_at(r.consumed_byte_count() - length + consumed, "Failed to parse frame header.", kDetailBadHeader);
    if (!override_allowed && frame_) throw_invalid_data_at(r.consumed_byte_count(), "Multiple frame is not supported.", kDetailBadHeader);
    frame_ = fh;
}

ScanHeader HostDecoder::process_scan_header(MarkerReader &r, bool metadata_only) {
    uint16_t length;
    const uint8_t *buf;
    if (!r.try_read_length(&length)) throw_invalid_data_at(r.consumed_byte_count(), "Unexpected end of input data when reading segment length.", kDetailBadHeader);
    if (!r.try_read_bytes(length, &buf)) throw_invalid_data_at(r.consumed_byte_count(), "Unexpected end of input data when reading segment content.", kDetailBadHeader);
    ScanHeader sh;
    int consumed;
    if (!ScanHeader::try_parse(buf, length, metadata_only, &sh, &consumed))
        throw_invalid_data_at(r.consumed_byte_count() - length + consumed, "Failed to parse scan header.", kDetailBadHeader);
    return sh;
}

void HostDecoder::process_dri(MarkerReader &r) {
    uint16_t length;
    const uint8_t *buf;
    if (!r.try_read_length(&length)) throw_invalid_data_at(r.consumed_byte_count(), "Unexpected end of input data when reading segment length.", kDetailBadHeader);
    if (!r.try_read_bytes(length, &buf) || length < 2)
        throw_invalid_data_at(r.consumed_byte_count(), "Unexpected end of input data when reading segment content.", kDetailBadHeader);
    restart_interval_ = (uint16_t)((buf[0] << 8) | buf[1]);
}

void HostDecoder::process_dht(MarkerReader &r) {
    uint16_t length;
    const uint8_t *buf;
    if (!r.try_read_length(&length)) throw_invalid_data_at(r.consumed_byte_count(), "Unexpected end of input data when reading segment length.", kDetailBadHeader);
    if (!r.try_read_bytes(length, &buf)) throw_invalid_data_at(r.consumed_byte_count(), "Unexpected end of input data when reading segment content.", kDetailBadHeader);
    int offset = r.consumed_byte_count() - length;
    size_t len = length;
    while (len != 0) {
        HuffTable t;
        int consumed;
        if (!HuffTable::try_parse(buf, len, &t, &consumed)) throw_invalid_data_at(offset, "Failed to parse Huffman table.", kDetailBadHeader);
        buf += consumed;
        len -= (size_t)consumed;
        offset += consumed;
        set_huffman_table(t);
    }
}

void HostDecoder::process_dqt(MarkerReader &r, bool load) {
    uint16_t length;
    const uint8_t *buf;
    if (!r.try_read_length(&length)) throw_invalid_data_at(r.consumed_byte_count(), "Unexpected end of input data when reading segment length.", kDetailBadHeader);
    if (!r.try_read_bytes(length, &buf)) throw_invalid_data_at(r.consumed_byte_count(), "Unexpected end of input data when reading segment content.", kDetailBadHeader);
    if (!load) return;
    int offset = r.consumed_byte_count() - length;
    size_t len = length;
    while (len != 0) {
        QuantTable t;
        int consumed;
        if (!QuantTable::try_parse(buf, len, &t, &consumed)) throw_invalid_data_at(offset, "Failed to parse quantization table.", kDetailBadHeader);
        buf += consumed;
        len -= (size_t)consumed;
        offset += consumed;
        set_quantization_table(t);
    }
}

static bool is_any_sof(int m) {
    switch (m) {
    case kSOF0: case kSOF1: case kSOF2: case kSOF3: case kSOF5: case kSOF6: case kSOF7:
    case kSOF9: case kSOF10: case kSOF11: case kSOF13: case kSOF14: case kSOF15:
        return true;
    default:
        return false;
    }
}

bool HostDecoder::process_marker_for_identification(int marker, MarkerReader &r, bool load_qt) {
    if (marker == kSOI) {
    } else if (is_any_sof(marker)) {
        start_of_frame_ = marker;
        process_frame_header(r, false, false);
    } else if (marker == kSOS) {
        process_scan_header(r, true);
    } else if (marker == kDRI) {
        process_dri(r);
    } else if (marker == kDQT) {
        process_dqt(r, load_qt);
    } else if (is_restart_marker(marker)) {
    } else if (marker == kEOI) {
        return false;
    } else {
        process_other_marker(r);
    }
    return true;
}

int HostDecoder::identify(bool load_quantization_tables) {
    if (!input_ || input_len_ == 0) throw_invalid_operation("Input buffer is not specified.");
    MarkerReader r(input_, input_len_);
    frame_.reset();
    bool to_continue = true;
    while (to_continue && !r.is_empty()) {
        int marker;
        if (!r.try_read_marker(&marker)) throw_invalid_data_at(r.consumed_byte_count(), "No marker found.", kDetailBadHeader);
        to_continue = process_marker_for_identification(marker, r, load_quantization_tables);
    }
    if (!frame_) throw_invalid_operation("Frame header was not found.");
    return r.consumed_byte_count();
}

bool HostDecoder::identify_until_scan(bool load_quantization_tables, size_t *scan_data_pos) {
    if (!input_ || input_len_ == 0) throw_invalid_operation("Input buffer is not specified.");
    MarkerReader r(input_, input_len_);
    frame_.reset();
    bool to_continue = true;
    while (to_continue && !r.is_empty()) {
        int marker;
        if (!r.try_read_marker(&marker)) throw_invalid_data_at(r.consumed_byte_count(), "No marker found.", kDetailBadHeader);
        to_continue = process_marker_for_identification(marker, r, load_quantization_tables);
        if (marker == kSOS) {
            *scan_data_pos = (size_t)r.consumed_byte_count();
            return true;
        }
    }
    if (!frame_) throw_invalid_operation("Frame header was not found.");
    return false;
}

// ref: JpegStandardQuantizationTable.cs:12-34 (zig-zag order as stored by the reference)
static const uint16_t kStdLum[64] = {16, 11, 12, 14, 12, 10, 16, 14, 13, 14, 18, 17, 16, 19, 24, 40, 26, 24, 22, 22, 24, 49,
                                     35, 37, 29, 40, 58, 51, 61, 60, 57, 51, 56, 55, 64, 72, 92, 78, 64, 68, 87, 69, 55, 56,
                                     80, 109, 81, 87, 95, 98, 103, 104, 103, 62, 77, 113, 121, 112, 100, 120, 92, 101, 103, 99};
static const uint16_t kStdChr[64] = {17, 18, 18, 24, 21, 24, 47, 26, 26, 47, 99, 66, 56, 66, 99, 99, 99, 99, 99, 99, 99, 99,
                                     99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99,
                                     99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99};

static float estimate_quality_one(const uint16_t *q, const uint16_t *std_table) {  // ref: JpegDecoder.cs:198-249
    bool all_ones = true;
    double sum = 0;
    for (int i = 0; i < 64; i++) {
        const double pct = q[i] == 0 ? 999.99 : 100.0 * q[i] / std_table[i];
        sum += pct;
        if (q[i] != 1) all_ones = false;
    }
    sum /= 64.0;
    if (all_ones) return 100.0f;
    if (sum <= 100.0) return (float)((200.0 - sum) / 2.0);
    return (float)(5000.0 / sum);
}

bool HostDecoder::try_estimate_quality(float *quality) const {
    const QuantTable *q0 = quantization_table(0);
    if (quant_.empty() || !q0) {
        *quality = 0;
        return false;
    }
    float q = estimate_quality_one(q0->elements, kStdLum);
    if (const QuantTable *q1 = quantization_table(1)) q = std::min(q, estimate_quality_one(q1->elements, kStdChr));
    *quality = std::min(std::max(q, 0.0f), 100.0f);
    return true;
}

bool HostDecoder::process_marker_for_decode(int marker, MarkerReader &r, ScanHandler &h) {
    switch (marker) {
    case kSOF0: case kSOF1: case kSOF2: case kSOF3: case kSOF9: case kSOF10:
        process_frame_header(r, false, true);
        h.on_frame(*this, marker);
        scan_decoder_created_ = true;
        break;
    case kSOF5: case kSOF6: case kSOF7: case kSOF11: case kSOF13: case kSOF14: case kSOF15:
        throw_invalid_data_at(r.consumed_byte_count(), "This type of JPEG stream is not supported.", kDetailUnsupportedFrame);
    case kDHT:
        process_dht(r);
        break;
    case kDAC:
        process_other_marker(r);  // arithmetic conditioning: not used by the Huffman path
        break;
    case kDQT:
        process_dqt(r, true);
        break;
    case kDRI:
        process_dri(r);
        break;
    case kSOS: {
        if (!scan_decoder_created_) throw_invalid_data_at(r.consumed_byte_count(), "Scan header appears before frame header.", kDetailBadHeader);
        const ScanHeader sh = process_scan_header(r, false);
        h.on_scan(*this, r, sh);
        break;
    }
    case kEOI:
        return false;
    default:
        if (is_restart_marker(marker)) break;
        process_other_marker(r);
        break;
    }
    return true;
}

void HostDecoder::decode(ScanHandler &handler, bool have_output_writer) {
    if (!input_ || input_len_ == 0) throw_invalid_operation("Input buffer is not specified.");
    if (!have_output_writer) throw_invalid_operation("The output buffer is not specified.");
    MarkerReader r(input_, input_len_);
    scan_decoder_created_ = false;
    if (!r.try_read_start_of_image()) throw_invalid_data_at(r.consumed_byte_count(), "Marker StartOfImage not found.", kDetailBadHeader);
    try {
        bool to_continue = true;
        while (to_continue && !r.is_empty()) {
            int marker;
            if (!r.try_read_marker(&marker)) throw_invalid_data_at(r.consumed_byte_count(), "No marker found.", kDetailBadHeader);
            to_continue = process_marker_for_decode(marker, r, handler);
        }
    } catch (...) {
        try {
            handler.on_dispose(*this);  // finally { _scanDecoder?.Dispose(); }
        } catch (const DecodeError &) {
            // the failure that ended the decode is the one the caller sees -- unless the handler decodes its scans HERE and one of
            // them failed: the reference ran that scan at its SOS, in front of the marker the walk failed at (JpegDecoder.cs:592-599)
            if (handler.dispose_failure_is_a_deferred_scan()) {
                scan_decoder_created_ = false;
                throw;
            }
        } catch (...) {
        }
        scan_decoder_created_ = false;
        throw;
    }
    handler.on_dispose(*this);
    scan_decoder_created_ = false;
}

void HostDecoder::load_tables(const uint8_t *data, size_t len) {
    MarkerReader r(data, len);
    while (!r.is_empty()) {
        int marker;
        if (!r.try_read_marker(&marker)) return;
        if (marker == kSOI || is_restart_marker(marker)) continue;
        if (marker == kDHT) process_dht(r);
        else if (marker == kDAC) process_other_marker(r);
        else if (marker == kDQT) process_dqt(r, true);
        else if (marker == kDRI) process_dri(r);
        else if (marker == kEOI) return;
        else process_other_marker(r);
    }
}

int HostDecoder::resolve_scan(const FrameHeader &fh, const ScanHeader &sh, ResolvedScanComponent out[kMaxScanComponents], bool optimizer_rules) const {
    int max_h = 1, max_v = 1;
    for (const FrameComponent &c : fh.components) {
        max_h = std::max(max_h, (int)c.h);
        max_v = std::max(max_v, (int)c.v);
    }
    // JpegOptimizer's own resolution (JpegOptimizer.cs:375-393) has no such check: it only fails on a selector no frame component carries
    if (!optimizer_rules && (int)fh.num_components < (int)sh.num_components)
        throw_invalid_operation("Operation is not valid due to the current state of the object.");
    if (sh.num_components > kMaxScanComponents) throw DecodeError(JPGPU_ERR_NOT_SUPPORTED, "More than 4 components in a scan are not supported.", kDetailUnsupportedFrame);
    for (int i = 0; i < sh.num_components; i++) {
        const ScanComponent &sc = sh.components[i];
        int component_index = 0;
        const FrameComponent *fc = nullptr;
        for (int j = 0; j < fh.num_components; j++)
            if (sc.selector == fh.components[j].identifier) {  // last match wins (no break in the reference)
                component_index = j;
                fc = &fh.components[j];
            }
        if (!fc) throw_invalid_data(optimizer_rules ? "Found invalid data while decoding." : "Failed to decode JPEG data. The specified component is missing.", kDetailBadHeader);
        // JpegOptimizer never divides by the sampling factors (:395-413): a zero factor just gives the component no blocks
        if (!optimizer_rules && (fc->h == 0 || fc->v == 0)) throw_invalid_data("Failed to decode JPEG data. Attempted to divide by zero.", kDetailBadHeader);
        ResolvedScanComponent &c = out[i];
        c.component_index = component_index;
        c.h = fc->h;
        c.v = fc->v;
        c.dc = huffman_table(true, sc.td);
        c.ac = huffman_table(false, sc.ta);
        c.quant = quantization_table(fc->tq);
        c.hs = fc->h ? max_h / fc->h : 1;
        c.vs = fc->v ? max_v / fc->v : 1;
        // WriteBlockSlow replicates with SHIFTS (log2 of the ratio, ...BaselineScanDecoder.cs:238-268): for a ratio that is not a
        // power of two it indexes past the 8 samples of a row.  Not reproduced.
        if (!optimizer_rules && ((c.hs & (c.hs - 1)) != 0 || (c.vs & (c.vs - 1)) != 0 || max_h % fc->h != 0 || max_v % fc->v != 0))
            throw DecodeError(JPGPU_ERR_NOT_SUPPORTED, "Sampling factor ratios that are not powers of two are not supported.", kDetailUnsupportedFrame);
    }
    return sh.num_components;
}

// ------------------------------------------------------------------------------------------------ scan jobs

BaselineGeometry BaselineGeometry::latch(const HostDecoder &dec, const FrameHeader &fh) {
    BaselineGeometry g;
    g.frame = fh;
    for (const FrameComponent &c : fh.components) {
        g.max_h = std::max(g.max_h, (int)c.h);
        g.max_v = std::max(g.max_v, (int)c.v);
    }
    g.restart_interval = dec.restart_interval();  // latched at construction
    g.mcus_per_line = (fh.samples_per_line + 8 * g.max_h - 1) / (8 * g.max_h);
    g.mcus_per_column = (fh.lines + 8 * g.max_v - 1) / (8 * g.max_v);
    g.level_shift = 1 << ((fh.precision - 1) & 31);
    return g;
}

ScanJob make_scan_job(const HostDecoder &dec, const BaselineGeometry &geo, const ScanHeader &scan, const uint8_t *entropy,
                      size_t entropy_len, bool optimizer_rules) {
    ScanJob job;
    job.geo = geo;
    job.entropy = entropy;
    job.entropy_len = entropy_len;
    if (geo.frame.components.empty() && geo.frame.num_components)
        throw_invalid_data("Failed to decode JPEG data. Component parameters are missing in JPEG frame header.", kDetailBadHeader);
    job.scan_components = dec.resolve_scan(geo.frame, scan, job.comp, optimizer_rules);
    for (int i = 0; i < job.scan_components; i++) {
        ResolvedScanComponent &c = job.comp[i];
        if (optimizer_rules) continue;
        if (!c.dc || !c.ac)
            throw_invalid_data("Failed to decode JPEG data. Huffman table of component " + std::to_string(c.component_index) + " is not defined.", kDetailMissingTable);
        if (!c.quant)
            throw_invalid_data("Failed to decode JPEG data. Quantization table of component " + std::to_string(c.component_index) + " is not defined.", kDetailMissingTable);
    }
    // snapshot tables; dedupe by registry pointer
    const HuffTable *seen[kMaxHuffSlots] = {};
    auto slot_of = [&](const HuffTable *t) -> uint8_t {
        for (int i = 0; i < job.n_huff; i++)
            if (seen[i] == t) return (uint8_t)i;
        seen[job.n_huff] = t;
        job.huff_copy[job.n_huff] = *t;
        return (uint8_t)job.n_huff++;
    };
    int nblk = 0;
    for (int i = 0; i < job.scan_components; i++) {
        ResolvedScanComponent &c = job.comp[i];
        job.dc_slot[i] = c.dc ? slot_of(c.dc) : kNullHuffSlot;
        job.ac_slot[i] = c.ac ? slot_of(c.ac) : kNullHuffSlot;
        if (c.quant) job.quant_copy[i] = *c.quant;
        // block order inside an MCU: scan-component order, then y < v, then x < h (ref: ...BaselineScanDecoder.cs:107-118)
        for (int y = 0; y < c.v; y++)
            for (int x = 0; x < c.h; x++) {
                if (nblk >= kMaxBlocksPerMcu)
                    throw DecodeError(JPGPU_ERR_NOT_SUPPORTED, "More than 16 blocks per MCU are not supported.", kDetailUnsupportedFrame);
                job.blk_comp[nblk] = (uint8_t)i;
                job.blk_x[nblk] = (uint8_t)x;
                job.blk_y[nblk] = (uint8_t)y;
                nblk++;
            }
    }
    if (nblk == 0) throw DecodeError(JPGPU_ERR_NOT_SUPPORTED, "A scan without blocks is not supported.", kDetailUnsupportedFrame);
    job.blocks_per_mcu = nblk;
    // the job owns snapshots (huff_copy / quant_copy, addressed by slot); drop the registry pointers so nothing
    // dangles when the job is moved or the registry changes
    for (int i = 0; i < job.scan_components; i++) job.comp[i].dc = job.comp[i].ac = nullptr, job.comp[i].quant = nullptr;
    return job;
}

// ------------------------------------------------------------------------------------------------ progressive frames

void ProgressiveFrame::begin(const HostDecoder &dec, const FrameHeader &fh) {
    reset();
    if (fh.components.empty() && fh.num_components)
        throw_invalid_data("Failed to decode JPEG data. Component parameters are missing in JPEG frame header.", kDetailBadHeader);
    if (fh.num_components > kMaxScanComponents)
        throw DecodeError(JPGPU_ERR_NOT_SUPPORTED, "More than 4 components are not supported.", kDetailUnsupportedFrame);
    geo_ = BaselineGeometry::latch(dec, fh);
    geo_.restart_interval = 0;  // progressive scans read DRI per scan
    slots_alloc_ = fh.num_components;
    for (int i = 0; i < kMaxScanComponents; i++) slot_set_[i] = slot_noquant_[i] = false;
    // JpegBlockAllocator.Allocate: block grid of every component (ref: JpegBlockAllocator.cs:35-84)
    const int hb = (fh.samples_per_line + 7) / 8, vb = (fh.lines + 7) / 8;
    int base = 0;
    for (int i = 0; i < fh.num_components; i++) {
        const FrameComponent &c = fh.components[i];
        if (c.h == 0 || c.v == 0) throw_invalid_data("Failed to decode JPEG data. Attempted to divide by zero.", kDetailBadHeader);
        const int hs = geo_.max_h / c.h, vs = geo_.max_v / c.v;
        if (hs == 0 || vs == 0) throw_invalid_data("Failed to decode JPEG data. Attempted to divide by zero.", kDetailBadHeader);
        // the same fence as resolve_scan's, where the checker has it for a progressive frame (its allocator): in front of whatever
        // the frame's scans would run into (tools/stress_parity.py STRESS_HEADER=1, seed 403)
        if ((hs & (hs - 1)) != 0 || (vs & (vs - 1)) != 0)
            throw DecodeError(JPGPU_ERR_NOT_SUPPORTED, "Sampling factor ratios that are not powers of two are not supported.", kDetailUnsupportedFrame);
        hblocks_[i] = (uint16_t)((hb + hs - 1) / hs);
        vblocks_[i] = (uint16_t)((vb + vs - 1) / vs);
        fblk_base_[i] = (uint8_t)base;
        base += c.h * c.v;
    }
    if (base > kMaxBlocksPerMcu) throw DecodeError(JPGPU_ERR_NOT_SUPPORTED, "More than 16 blocks per MCU are not supported.", kDetailUnsupportedFrame);
    frame_bpm_ = (uint8_t)base;
    active_ = true;
}

void ProgressiveFrame::add_scan(const HostDecoder &dec, const ScanHeader &scan, const uint8_t *entropy, size_t entropy_len) {
    const FrameHeader &fh = geo_.frame;
    ScanJob job;
    job.kind = kScanProgressive;
    job.geo = geo_;
    job.entropy = entropy;
    job.entropy_len = entropy_len;
    job.ss = scan.ss;
    job.se = scan.se;
    job.ah = scan.ah;
    job.al = scan.al;
    // Scans run in dependency LEVELS, not strictly one after the other: a scan waits for the earlier scans that touch
    // the same coefficients of the same component (every kernel write is an exact 2-byte store or a 32-bit atomic OR on
    // the DC word, so disjoint bands of the same blocks can be in flight together).  File order inside a level is
    // irrelevant for the result; errors are still reported for the first failing scan in file order.
    job.ordinal = 0;
    job.scan_dri = dec.restart_interval();
    job.frame_bpm = frame_bpm_;
    if (scan.num_components == 0)
        throw DecodeError(JPGPU_ERR_NOT_SUPPORTED, "A progressive scan without components is not supported.", kDetailUnsupportedFrame);
    if (scan.se > 63 || scan.ss > 63)  // the reference's block readers then walk into the next blocks of its store
        throw DecodeError(JPGPU_ERR_NOT_SUPPORTED, "A spectral selection beyond coefficient 63 is not supported.", kDetailUnsupportedFrame);
    // InitDecodeComponents (:60) fills the decoder's component slots one by one, BEFORE any of ProcessScan's own checks: whatever
    // fails afterwards (or half-way through it), Dispose() -- which Decode()'s `finally` still runs -- iterates the slots as they
    // stand then
    for (int i = 0; i < kMaxScanComponents; i++) job.comp[i].component_index = -1;
    auto record_slots = [&] {
        for (int i = 0; i < kMaxScanComponents; i++) {
            const ResolvedScanComponent &c = job.comp[i];
            if (c.component_index < 0) continue;
            slot_[i] = c;
            slot_noquant_[i] = c.quant == nullptr;
            if (c.quant) slot_quant_[i] = *c.quant;
            slot_set_[i] = true;
        }
    };
    try {
        job.scan_components = dec.resolve_scan(fh, scan, job.comp);
    } catch (...) {
        record_slots();
        throw;
    }
    record_slots();
    for (int i = 0; i < job.scan_components; i++) {  // :63-69
        if (!job.comp[i].quant)
            throw_invalid_data("Failed to decode JPEG data. Quantization table of component " + std::to_string(job.comp[i].component_index) + " is not defined.",
                               kDetailMissingTable);
    }
    const bool interleaved = job.scan_components != 1;
    const bool needs_dc = interleaved || scan.ss == 0;
    for (int i = 0; i < job.scan_components; i++) {  // :97-103, :150-153, :171-174
        if (needs_dc ? !job.comp[i].dc : !job.comp[i].ac)
            throw_invalid_data("Failed to decode JPEG data. Huffman table of component " + std::to_string(job.comp[i].component_index) + " is not defined.",
                               kDetailMissingTable);
    }
    // the reference walks whatever band the header names; bands outside the block are outside what we reproduce
    // (an AC refinement band that leaves the block makes it read its neighbours' coefficients; first passes clamp the index)
    if (!interleaved && scan.ss != 0 && scan.ah != 0 && scan.se > 63)
        throw DecodeError(JPGPU_ERR_NOT_SUPPORTED, "AC refinement scans whose spectral band leaves the block are not supported.",
                          kDetailUnsupportedFrame);
    // snapshot the tables the scan decodes with; dedupe by registry pointer
    const HuffTable *seen[kMaxHuffSlots] = {};
    auto slot_of = [&](const HuffTable *t) -> uint8_t {
        for (int i = 0; i < job.n_huff; i++)
            if (seen[i] == t) return (uint8_t)i;
        seen[job.n_huff] = t;
        job.huff_copy[job.n_huff] = *t;
        return (uint8_t)job.n_huff++;
    };
    for (int i = 0; i < job.scan_components; i++) {
        ResolvedScanComponent &c = job.comp[i];
        job.dc_slot[i] = job.ac_slot[i] = 0;
        if (needs_dc) job.dc_slot[i] = slot_of(c.dc);
        else job.ac_slot[i] = slot_of(c.ac);
        job.quant_copy[i] = *c.quant;
        job.fblk_base[i] = fblk_base_[c.component_index];
        job.hblocks[i] = hblocks_[c.component_index];
        job.vblocks[i] = vblocks_[c.component_index];
    }
    if (interleaved) {
        job.units_per_line = (uint32_t)geo_.mcus_per_line;
        job.total_units = (uint32_t)(geo_.mcus_per_line * geo_.mcus_per_column);
    } else {
        const ResolvedScanComponent &c = job.comp[0];  // :144-147
        job.units_per_line = (uint32_t)((fh.samples_per_line + 8 * c.hs - 1) / (8 * c.hs));
        job.total_units = job.units_per_line * (uint32_t)((fh.lines + 8 * c.vs - 1) / (8 * c.vs));
    }
    std::vector<int> sharing;
    for (size_t ei = 0; ei < scans_.size(); ei++) {
        const ScanJob &e = scans_[ei];
        const bool e_interleaved = e.scan_components != 1;
        // The band a scan may WRITE is wider than its header says: the reference stores a first-pass coefficient at
        // min(k, 63) after k += run without looking at Se (:262-275), and a refinement pass its new coefficient at the index
        // the zero-history walk stopped at, which may be Se + 1 (:330-345).  Well-formed streams never do that, corrupted
        // ones do, and then file order decides what a coefficient ends up as: such scans must not share a level.
        auto written_hi = [](bool il, int ss_, int se_, int ah_) { return il || ss_ == 0 ? se_ : std::min(63, se_ + (ah_ == 0 ? 15 : 1)); };
        const int e_lo = e_interleaved ? 0 : e.ss, e_hi = e_interleaved ? 0 : written_hi(false, e.ss, e.se, e.ah);  // interleaved scans are DC scans
        const int lo = interleaved ? 0 : scan.ss, hi = interleaved ? 0 : written_hi(false, scan.ss, scan.se, scan.ah);
        if (lo > e_hi || e_lo > hi) continue;
        bool shares = false;
        for (int a = 0; a < job.scan_components; a++)
            for (int b = 0; b < e.scan_components; b++) shares |= job.comp[a].component_index == e.comp[b].component_index;
        if (!shares) continue;
        job.ordinal = std::max(job.ordinal, e.ordinal + 1);
        sharing.push_back((int)ei);
        if (ei < 64) job.dep_closure |= e.dep_closure | (1ull << ei);
    }
    // direct dependencies: sharing scans that no other sharing scan already waits for
    if (scans_.size() >= 64) job.n_deps = 4;  // outside the bookkeeping: level by level
    for (int ei : sharing) {
        bool implied = false;
        for (int fi : sharing) implied |= fi != ei && fi < 64 && ei < 64 && ((scans_[fi].dep_closure >> ei) & 1ull) != 0;
        if (implied) continue;
        if (job.n_deps < 3) job.deps[job.n_deps] = ei;
        job.n_deps++;
        scans_[ei].has_consumers = true;
    }
    for (int i = 0; i < job.scan_components; i++) {
        job.comp[i].dc = job.comp[i].ac = nullptr;
        job.comp[i].quant = nullptr;
    }
    scans_.push_back(std::move(job));
}

ScanJob ProgressiveFrame::make_frame_job() const {
    const FrameHeader &fh = geo_.frame;
    // which slot transforms which frame component (Dispose iterates the slots, :425-447)
    int slot_of_component[kMaxScanComponents];
    for (int i = 0; i < kMaxScanComponents; i++) slot_of_component[i] = -1;
    bool ok = true;
    for (int i = 0; i < slots_alloc_; i++) {
        if (!slot_set_[i] || slot_noquant_[i]) ok = false;
        else if (slot_of_component[slot_[i].component_index] >= 0) ok = false;
        else slot_of_component[slot_[i].component_index] = i;
    }
    // Slots that do not map one to one onto the components: the reference transforms a component once per slot that points at
    // it, in slot order and in place -- twice, or never (ScanDecoder/JpegHuffmanProgressiveScanDecoder.cs:425-447; a slot no
    // scan ever filled has sampling factors 0 and does nothing).  Reproduced literally by dispose_pass_kernel.
    const bool generic = !ok;
    // a sequential-style job over ALL frame components in frame order with the slots' quantisation tables
    HostDecoder tmp;
    FrameHeader f2 = fh;
    ScanHeader sh;
    sh.num_components = fh.num_components;
    sh.se = 63;
    for (int c = 0; c < fh.num_components; c++) {
        f2.components[c].tq = (uint8_t)c;
        QuantTable q;
        if (slot_of_component[c] >= 0 && slot_set_[slot_of_component[c]]) q = slot_quant_[slot_of_component[c]];
        else for (uint16_t &e : q.elements) e = 1;  // (generic Dispose(): K3 does not transform, the table is not looked at)
        q.identifier = (uint8_t)c;
        tmp.set_quantization_table(q);
        sh.components.push_back({fh.components[c].identifier, 0, 0});
    }
    HuffTable dummy;  // the IDCT pass needs no Huffman tables; a placeholder satisfies the job builder's checks
    const uint8_t bits[16] = {0, 1};
    const uint8_t vals[1] = {0};
    HuffTable::from_bits_values(0, 0, bits, vals, 1, &dummy);
    tmp.set_huffman_table(dummy);
    dummy.table_class = 1;
    tmp.set_huffman_table(dummy);
    tmp.set_frame_header(f2);
    BaselineGeometry g = BaselineGeometry::latch(tmp, f2);
    ScanJob job = make_scan_job(tmp, g, sh, nullptr, 0);
    job.kind = kScanFrameOnly;
    job.dispose_generic = generic;
    {   // (always written down: the partial flush of a file that fails later takes the slots as the host's walk left them)
        for (int i = 0; i < slots_alloc_; i++) {
            if (!slot_set_[i] || slot_noquant_[i]) continue;  // (a slot without a table: the reference's Debug.Assert; the checker skips it)
            const int c = slot_[i].component_index;
            if (c < 0 || c >= kMaxScanComponents || job.dispose_n[c] >= kMaxScanComponents) continue;
            job.dispose_q[c][job.dispose_n[c]++] = slot_quant_[i];
        }
    }
    return job;
}

YccRgbFactors ycc_rgb_factors() {
    auto fix = [](float x) { return (int32_t)((double)(x * (float)(1L << 16)) + 0.5); };  // Fix (:121-124)
    const float luma_red = 299 / 1000.0f, luma_green = 587 / 1000.0f, luma_blue = 114 / 1000.0f;  // :34-37
    const float f1 = 2 - 2 * luma_red;
    const float f2 = luma_red * f1 / luma_green;
    const float f3 = 2 - 2 * luma_blue;
    const float f4 = luma_blue * f3 / luma_green;
    YccRgbFactors k;
    k.cr_r = fix(f1);    // D1
    k.cr_g = -fix(f2);   // D2
    k.cb_b = fix(f3);    // D3
    k.cb_g = -fix(f4);   // D4
    return k;
}

size_t find_scan_end(const uint8_t *data, size_t len, std::string *restart_markers) {
    size_t pos = 0;
    while (pos + 1 < len) {
        const uint8_t *q = static_cast<const uint8_t *>(memchr(data + pos, 0xFF, len - 1 - pos));
        if (!q) return len;
        pos = (size_t)(q - data);
        const uint8_t nb = data[pos + 1];
        if (nb == 0x00) {
            pos += 2;
        } else if (nb == 0xFF) {
            pos += 1;
        } else if (is_restart_marker(nb)) {
            if (restart_markers) restart_markers->push_back((char)nb);
            pos += 2;
        } else {
            return pos;
        }
    }
    return len;
}

}  // namespace jpgpu
