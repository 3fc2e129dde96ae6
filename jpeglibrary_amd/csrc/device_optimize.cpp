// jpeglibrary_amd/csrc/device_optimize.cpp -- JpegOptimizer (ref: JpegOptimizer.cs) for a batch of baseline files.
//
// The reference runs two passes over one file: Scan() decodes every symbol of the scan and counts it per Huffman table
// (:360-463), builds new tables from the counts (JpegHuffmanEncodingTableBuilder.Build), and Optimize() walks the file
// again, copying the segments it keeps, writing the new DHT in place of the first one and re-writing the scan symbol by
// symbol with the new codes (:540-880).  Here the marker walks run on the host (they touch a few hundred bytes), the
// symbol passes on the GPU: K1 (marker index + unstuffing, shared with the decoder) then KT count -> host table build ->
// KT measure -> exclusive scan -> KT emit (kt_transcode.hip).  The output is assembled from the host pieces and the
// device-resident scan data at download time.
//
// Fences (reported as JPGPU_ERR_NOT_SUPPORTED): files with more than one scan (the reference builds its tables from the
// LAST scan only, :462, and then fails or writes garbage for the others), progressive frames (:580-582), a restart interval
// that changes between the frame header and the scan or after the scan.  Parity of the bytes is unpinned (the reference holds no golden bytes, and
// .NET's unstable sort decides the order of equal-length symbols in the DHT).
#include "device_optimize.h"

#include <algorithm>
#include <cstring>

namespace jpgpu {

// ------------------------------------------------------------------------------------------------ table builder
namespace {
struct Sym {
    int64_t frequency;
    int16_t value;
    uint16_t code_size;
    int16_t others;
};

// Array.Sort(T[], Comparison<T>) / List<T>.Sort(Comparison<T>) of .NET Core 3.0+ / .NET 5+ (ArraySortHelper<T>.IntrospectiveSort:
// insertion sort up to 16 elements, median-of-three pivot parked at hi - 1, heapsort below depth 2 * (log2(n) + 1)).  The
// reference orders its symbols and its package-merge nodes with it, and the sort is not stable: what it leaves depends on
// the algorithm, so the algorithm is restated.  It only moves whole elements and looks at them through the comparison,
// which makes an array of pointers to the elements take exactly the same path.
template <typename T, typename Cmp>
struct NetSort {
    const T **k;
    Cmp cmp;
    void swap_if_greater(int i, int j) {
        if (i != j && cmp(*k[i], *k[j]) > 0) std::swap(k[i], k[j]);
    }
    void insertion(int lo, int n) {
        for (int i = 0; i < n - 1; i++) {
            const T *t = k[lo + i + 1];
            int j = i;
            while (j >= 0 && cmp(*t, *k[lo + j]) < 0) {
                k[lo + j + 1] = k[lo + j];
                j--;
            }
            k[lo + j + 1] = t;
        }
    }
    void down_heap(int lo, int i, int n) {
        const T *d = k[lo + i - 1];
        while (i <= n >> 1) {
            int child = 2 * i;
            if (child < n && cmp(*k[lo + child - 1], *k[lo + child]) < 0) child++;
            if (!(cmp(*d, *k[lo + child - 1]) < 0)) break;
            k[lo + i - 1] = k[lo + child - 1];
            i = child;
        }
        k[lo + i - 1] = d;
    }
    void heap(int lo, int n) {
        for (int i = n >> 1; i >= 1; i--) down_heap(lo, i, n);
        for (int i = n; i > 1; i--) {
            std::swap(k[lo], k[lo + i - 1]);
            down_heap(lo, 1, i - 1);
        }
    }
    int partition(int lo, int n) {
        const int hi = n - 1, middle = hi >> 1;
        swap_if_greater(lo, lo + middle);
        swap_if_greater(lo, lo + hi);
        swap_if_greater(lo + middle, lo + hi);
        const T *pivot = k[lo + middle];
        std::swap(k[lo + middle], k[lo + hi - 1]);
        int left = 0, right = hi - 1;
        while (left < right) {
            while (cmp(*k[lo + ++left], *pivot) < 0) {
            }
            while (cmp(*pivot, *k[lo + --right]) < 0) {
            }
            if (left >= right) break;
            std::swap(k[lo + left], k[lo + right]);
        }
        if (left != hi - 1) std::swap(k[lo + left], k[lo + hi - 1]);
        return left;
    }
    void intro(int lo, int n, int depth) {
        while (n > 1) {
            if (n <= 16) {
                if (n == 2) {
                    swap_if_greater(lo, lo + 1);
                } else if (n == 3) {
                    swap_if_greater(lo, lo + 1);
                    swap_if_greater(lo, lo + 2);
                    swap_if_greater(lo + 1, lo + 2);
                } else {
                    insertion(lo, n);
                }
                return;
            }
            if (depth == 0) {
                heap(lo, n);
                return;
            }
            depth--;
            const int p = partition(lo, n);
            intro(lo + p + 1, n - (p + 1), depth);
            n = p;
        }
    }
};
template <typename T, typename Cmp>
void net_sort(std::vector<const T *> &ptrs, Cmp cmp) {
    const int n = (int)ptrs.size();
    if (n < 2) return;
    int log2 = 0;
    for (unsigned v = (unsigned)n; v > 1; v >>= 1) log2++;
    NetSort<T, Cmp> s{ptrs.data(), cmp};
    s.intro(0, n, 2 * (log2 + 1));
}
template <typename Cmp>
void net_sort_symbols(Sym *s, int n, Cmp cmp) {
    std::vector<Sym> copy(s, s + n);
    std::vector<const Sym *> ptrs((size_t)n);
    for (int i = 0; i < n; i++) ptrs[i] = &copy[i];
    net_sort<Sym>(ptrs, cmp);
    for (int i = 0; i < n; i++) s[i] = *ptrs[i];
}

// BuildCanonicalCode: code words for lengths already in DHT order (:237-283, :470-497)
void assign_canonical_codes(std::vector<OptimalCode> &codes) {
    uint16_t code = 0;
    int count = codes[0].length;
    codes[0].code = 0;
    for (size_t i = 1; i < codes.size(); i++) {
        OptimalCode &c = codes[i];
        if (c.length > count) {
            code++;
            code = (uint16_t)(code << (c.length - count));
            c.code = code;
            count = c.length;
        } else {
            c.code = ++code;
        }
    }
}

// MostOptimalCoding: BuildUsingPackageMerge + RunPackageMerge (JpegHuffmanEncodingTableBuilder.cs:289-428)
struct PmNode {
    int64_t frequency = 0;
    int16_t index = 0;
    const PmNode *left = nullptr, *right = nullptr;
};
void pm_traverse(const PmNode *node, Sym *symbols) {
    if (!node) return;
    if (!node->left) {
        symbols[node->index].code_size++;
    } else {
        pm_traverse(node->left, symbols);
        pm_traverse(node->right, symbols);
    }
}
bool build_package_merge(const uint32_t freq[256], std::vector<OptimalCode> *codes) {
    int code_count = 0;
    for (int i = 0; i < 256; i++) code_count += freq[i] != 0;
    if (code_count == 0) return false;
    Sym s[257];
    int n = 0;
    for (int i = 0; i < 256; i++)
        if (freq[i] != 0) s[n++] = {(int64_t)freq[i], (int16_t)i, 0, 0};
    s[n++] = {0, -1, 0, 0};
    net_sort_symbols(s, n, [](const Sym &x, const Sym &y) { return y.frequency < x.frequency ? -1 : (y.frequency > x.frequency ? 1 : 0); });
    std::vector<PmNode> pool;
    pool.reserve((size_t)n * 40);
    std::vector<const PmNode *> levels[16];
    for (int l = 15; l >= 0; l--)
        for (int i = 0; i < n; i++) {
            pool.push_back(PmNode{s[i].frequency, (int16_t)i, nullptr, nullptr});
            levels[l].push_back(&pool.back());
        }
    auto desc = [](const PmNode &x, const PmNode &y) { return y.frequency < x.frequency ? -1 : (y.frequency > x.frequency ? 1 : 0); };
    auto asc = [](const PmNode &x, const PmNode &y) { return x.frequency < y.frequency ? -1 : (x.frequency > y.frequency ? 1 : 0); };
    for (int l = 15; l > 0; l--) {
        std::vector<const PmNode *> &nodes = levels[l];
        net_sort<PmNode>(nodes, desc);
        while (nodes.size() >= 2) {
            const PmNode *n1 = nodes[nodes.size() - 1], *n2 = nodes[nodes.size() - 2];
            nodes.resize(nodes.size() - 2);
            pool.push_back(PmNode{n1->frequency + n2->frequency, 0, n1, n2});
            levels[l - 1].push_back(&pool.back());
        }
    }
    net_sort<PmNode>(levels[0], asc);
    const int select = std::max(1, 2 * (n - 1));
    for (int i = 0; i < select; i++) pm_traverse(levels[0][(size_t)i], s);
    net_sort_symbols(s, n, [](const Sym &x, const Sym &y) {  // SymbolComparer (:430-453)
        if (x.code_size > y.code_size) return 1;
        if (x.code_size < y.code_size) return -1;
        if (x.frequency > y.frequency) return -1;
        if (x.frequency < y.frequency) return 1;
        return 0;
    });
    int index = 0;
    for (int i = n - 1; i >= 0; i--)
        if (s[i].value == -1) {
            index = i;
            break;
        }
    for (int i = index; i < n - 1; i++) s[i] = s[i + 1];
    codes->assign((size_t)code_count, OptimalCode{0, 0, 0});
    for (int i = 0; i < code_count; i++) {
        if (s[i].code_size > 16) return false;
        (*codes)[i].symbol = (uint8_t)s[i].value;
        (*codes)[i].length = (uint8_t)s[i].code_size;
    }
    assign_canonical_codes(*codes);
    return true;
}
}  // namespace

void net_sort_permutation(const int32_t *keys, int n, int32_t *perm) {
    std::vector<const int32_t *> ptrs((size_t)std::max(0, n));
    for (int i = 0; i < n; i++) ptrs[(size_t)i] = keys + i;
    net_sort<int32_t>(ptrs, [](const int32_t &x, const int32_t &y) { return x < y ? -1 : (x > y ? 1 : 0); });
    for (int i = 0; i < n; i++) perm[i] = (int32_t)(ptrs[(size_t)i] - keys);
}

bool build_optimal_table(const uint32_t freq[256], std::vector<OptimalCode> *codes, bool most_optimal) {
    if (most_optimal) return build_package_merge(freq, codes);
    int code_count = 0;
    for (int i = 0; i < 256; i++) code_count += freq[i] != 0;
    if (code_count == 0) return false;
    Sym s[257];
    int n = 0;
    for (int i = 0; i < 256; i++)
        if (freq[i] != 0) s[n++] = {(int64_t)freq[i], (int16_t)i, 0, -1};
    s[n++] = {1, -1, 0, -1};  // the reserved symbol that keeps the all-ones code unused
    // Figure K.1 as the reference runs it (:178-235): least frequency, FIRST index among equals
    for (;;) {
        int v1 = -1, v2 = -1;
        int64_t f1 = -1, f2 = -1;
        for (int i = 0; i < n; i++)
            if (s[i].frequency >= 0 && (v1 == -1 || s[i].frequency < f1)) {
                v1 = i;
                f1 = s[i].frequency;
            }
        for (int i = 0; i < n; i++)
            if (s[i].frequency >= 0 && i != v1 && (v2 == -1 || s[i].frequency < f2)) {
                v2 = i;
                f2 = s[i].frequency;
            }
        if (v2 == -1) break;
        s[v1].frequency += s[v2].frequency;
        s[v2].frequency = -1;
        s[v1].code_size++;
        while (s[v1].others != -1) {
            v1 = s[v1].others;
            s[v1].code_size++;
        }
        s[v1].others = (int16_t)v2;
        s[v2].code_size++;
        while (s[v2].others != -1) {
            v2 = s[v2].others;
            s[v2].code_size++;
        }
    }
    // Figures K.2 / K.3 on the reference's 0-based 60-entry byte array (:117-158)
    uint8_t bits[60] = {};
    int index = 32;
    for (int i = 0; i < n; i++) {
        const int cs = s[i].code_size;
        if (cs > 0) {
            index = std::max(index, cs);
            if (index >= 60) return false;
            bits[cs - 1]++;
        }
    }
    // The counters are BYTES in the reference (Span<byte> bits, :118): 256 codes of one length -- 255 symbols of nearly equal
    // frequency plus the reserved one, a perfect tree of depth 8 -- wrap to zero, and the searches below then walk off the
    // front of the span: the reference throws IndexOutOfRangeException.  Same here: failure, not a walk through the stack.
    for (;;) {
        while (bits[index] > 0) {
            int j = index - 1;
            do {
                j -= 1;
                if (j < 0) return false;
            } while (bits[j] == 0);
            bits[index] = (uint8_t)(bits[index] - 2);
            bits[index - 1] = (uint8_t)(bits[index - 1] + 1);
            bits[j + 1] = (uint8_t)(bits[j + 1] + 2);
            bits[j] = (uint8_t)(bits[j] - 1);
        }
        index -= 1;
        if (index != 15) continue;
        while (bits[index] == 0) {
            index--;
            if (index < 0) return false;
        }
        bits[index]--;
        break;
    }
    for (int i = 0; i < n; i++)
        if (s[i].value == -1) s[i].code_size = 0xFFFF;
    net_sort_symbols(s, n, [](const Sym &x, const Sym &y) { return x.code_size < y.code_size ? -1 : (x.code_size > y.code_size ? 1 : 0); });
    // BuildCanonicalCode (:237-283)
    codes->assign((size_t)code_count, OptimalCode{0, 0, 0});
    int length = 1, at = 0;
    uint8_t left = bits[0];
    for (int i = 0; i < code_count; i++) {
        while (left == 0) {
            left = bits[++at];
            length++;
        }
        left--;
        (*codes)[i].symbol = (uint8_t)s[i].value;
        (*codes)[i].length = (uint8_t)length;
    }
    assign_canonical_codes(*codes);
    return true;
}

// ------------------------------------------------------------------------------------------------ batch
OptimizeBatch::~OptimizeBatch() {
    for (DevBuffer *b : {&d_work_, &d_scan_ids_, &d_hist_, &d_enc_, &d_sizes_, &d_offsets_, &d_base_, &d_totals_, &d_out_, &d_sub_work_,
                         &d_sub_scan_ids_, &d_sub_bits_, &d_sub_bitoff_, &d_sub_totals_, &d_scan_raw_off_, &d_raw_, &d_simages_, &d_swork_chunk_,
                         &d_chunk_ff_, &d_sout_, &d_sout_len_})
        b->release();
    if (ev0_) (void)hipEventDestroy(ev0_);
    if (ev1_) (void)hipEventDestroy(ev1_);
}
int OptimizeBatch::fail(int status, const std::string &msg) {
    ctx_->last_error = msg;
    return status;
}
int OptimizeBatch::hip_fail(hipError_t e, const char *what) {
    return fail(JPGPU_ERR_DEVICE, std::string(what) + ": " + hipGetErrorString(e));
}

namespace {
void put_marker(std::string &o, int m) {
    o.push_back((char)0xFF);
    o.push_back((char)m);
}
void put_length(std::string &o, uint16_t length) {  // JpegWriter.WriteLength: length + 2, big endian (:309-321)
    const uint16_t v = (uint16_t)(length + 2);
    o.push_back((char)(v >> 8));
    o.push_back((char)v);
}
struct Refuse {
    int status, detail;
    std::string msg;
};
[[noreturn]] void refuse(int status, const std::string &msg, int detail = 0) { throw Refuse{status, detail, msg}; }
std::string at_offset(int off, const char *msg) { return "Failed to decode JPEG data at offset " + std::to_string(off) + ". " + msg; }
}  // namespace

// find_scan_end of the scan the walks of one file keep coming back to (four walks, one search)
size_t OptimizeBatch::scan_end(const uint8_t *entropy, size_t len) {
    if (entropy != end_key_ || len != end_len_) {
        end_key_ = entropy;
        end_len_ = len;
        end_rsts_.clear();
        end_val_ = find_scan_end(entropy, len, &end_rsts_);
    }
    return end_val_;
}

// Scan()'s and Optimize()'s marker walks (JpegOptimizer.cs:66-153, :540-648) on the host.  Leaves the pieces of the
// output in p.pieces; the scan itself is located for the device passes.
void OptimizeBatch::plan_file(Plan &p, const uint8_t *data, size_t len, bool strip, bool swallow_terminator) {
    // ---- Scan(): tables, frame, restart interval as of the scan; exactly one scan
    std::vector<QuantTable> quant;  // _quantizationTables: replace by identifier, else append (:319-338)
    bool after_scan = false;  // from here on a failure of the walks is only met once the scan itself went through
    try {
    bool have_frame = false;
    FrameHeader frame;
    int n_scans = 0;
    uint16_t dri = 0, dri_at_scan = 0;
    {
        if (len == 0) refuse(JPGPU_ERR_INVALID_OPERATION, "Input buffer is not specified.");
        MarkerReader r(data, len);
        bool eoi = false;
        while (!eoi && !r.is_empty()) {
            int marker;
            if (!r.try_read_marker(&marker)) refuse(JPGPU_ERR_INVALID_DATA, at_offset(r.consumed_byte_count(), "No marker found."));
            uint16_t length;
            const uint8_t *buf;
            auto segment = [&]() {
                if (!r.try_read_length(&length))
                    refuse(JPGPU_ERR_INVALID_DATA, at_offset(r.consumed_byte_count(), "Unexpected end of input data when reading segment length."));
                if (!r.try_read_bytes(length, &buf))
                    refuse(JPGPU_ERR_INVALID_DATA, at_offset(r.consumed_byte_count(), "Unexpected end of input data when reading segment content."));
            };
            switch (marker) {
            case 0xD8: break;
            case 0xC0:
            case 0xC1: {
                segment();
                FrameHeader fh;
                int consumed = 0;
                if (!FrameHeader::try_parse(buf, length, false, &fh, &consumed))
                    refuse(JPGPU_ERR_INVALID_DATA, at_offset(r.consumed_byte_count() - length + consumed, "Failed to parse frame header."));
                if (have_frame) refuse(JPGPU_ERR_INVALID_DATA, at_offset(r.consumed_byte_count(), "Multiple frame is not supported."));
                have_frame = true;
                frame = fh;
                break;
            }
            // StartOfFrame2 is not among Scan()'s cases (:95-146): a progressive frame header is skipped like an APPn segment and
            // the first SOS then finds no frame header (the null reference below)
            case 0xC3: case 0xC5: case 0xC6: case 0xC7: case 0xC9: case 0xCA: case 0xCB: case 0xCD: case 0xCE: case 0xCF:
                refuse(JPGPU_ERR_INVALID_DATA,
                       at_offset(r.consumed_byte_count(), ("This type of JPEG stream is not supported (StartOfFrame" + std::to_string(marker - 0xC0) + ").").c_str()),
                       kDetailUnsupportedFrame);
            case 0xC4: {
                // The decoding tables themselves are the decoder-side parser's business (batch_), but a table that does not parse
                // ends Scan() HERE (JpegOptimizer.cs ProcessDefineHuffmanTable), in front of whatever a later marker would have
                // thrown -- e.g. the missing frame header at SOS when a bit flip made the SOF marker a DHT
                // (tests/golden/stress/optimizer_dht_in_place_of_sof_*.jpg)
                segment();
                const uint8_t *tb = buf;
                size_t rem = length;
                int off = r.consumed_byte_count() - length;
                while (rem != 0) {
                    HuffTable t;
                    int consumed = 0;
                    if (!HuffTable::try_parse(tb, rem, &t, &consumed)) refuse(JPGPU_ERR_INVALID_DATA, at_offset(off, "Failed to parse Huffman table."));
                    tb += consumed;
                    rem -= (size_t)consumed;
                    off += consumed;
                }
                break;
            }
            case 0xDB: {
                segment();
                const int base = r.consumed_byte_count() - length;
                int off = 0;
                while (off < (int)length) {
                    QuantTable t;
                    int consumed = 0;
                    if (!QuantTable::try_parse(buf + off, (size_t)length - off, &t, &consumed))
                        refuse(JPGPU_ERR_INVALID_DATA, at_offset(base + off, "Failed to parse quantization table."));
                    off += consumed;
                    bool replaced = false;
                    for (QuantTable &q : quant)
                        if (q.identifier == t.identifier) {
                            q = t;
                            replaced = true;
                        }
                    if (!replaced) quant.push_back(t);
                }
                break;
            }
            case 0xDD:
                segment();
                if (length < 2)
                    refuse(JPGPU_ERR_INVALID_DATA, at_offset(r.consumed_byte_count(), "Unexpected end of input data when reading segment content."));
                dri = (uint16_t)(buf[0] << 8 | buf[1]);
                break;
            case 0xDA: {
                segment();
                if (!have_frame) refuse(JPGPU_ERR_INVALID_OPERATION, "Object reference not set to an instance of an object.");
                if (++n_scans > 1)
                    refuse(JPGPU_ERR_NOT_SUPPORTED, "Files with more than one scan are not supported by the optimizer path.", kDetailUnsupportedFrame);
                dri_at_scan = dri;
                p.dri_at_scan = dri;
                after_scan = true;
                const size_t end = scan_end(r.remaining_bytes(), (size_t)r.remaining_byte_count());
                r.try_advance((int)end + (swallow_terminator && end < (size_t)r.remaining_byte_count() ? 1 : 0));
                break;
            }
            case 0xD0: case 0xD1: case 0xD2: case 0xD3: case 0xD4: case 0xD5: case 0xD6: case 0xD7: break;
            case 0xD9: eoi = true; break;
            default:
                if (!r.try_read_length(&length))
                    refuse(JPGPU_ERR_INVALID_DATA, at_offset(r.consumed_byte_count(), "Unexpected end of input data when reading segment length."));
                if (!r.try_advance(length))
                    refuse(JPGPU_ERR_INVALID_DATA, at_offset(r.consumed_byte_count(), "Unexpected end of input data reached."));
                break;
            }
        }
        if (n_scans == 0) refuse(JPGPU_ERR_INVALID_DATA, at_offset(r.consumed_byte_count(), "No image data is read."));
        if (dri_at_scan != dri)
            refuse(JPGPU_ERR_NOT_SUPPORTED, "A restart interval that changes after the scan is not supported by the optimizer path.", kDetailUnsupportedFrame);
    }
    // ---- Optimize(strip)
    MarkerReader r(data, len);
    std::string cur;
    auto flush = [&]() {
        if (!cur.empty()) p.pieces.push_back({Piece::kBytes, cur});
        cur.clear();
    };
    auto copy_segment = [&](const uint8_t **bytes, uint16_t *n) {  // CopyMarkerData (:661-675)
        uint16_t length;
        const uint8_t *buf;
        if (!r.try_read_length(&length))
            refuse(JPGPU_ERR_INVALID_DATA, at_offset(r.consumed_byte_count(), "Unexpected end of input data when reading segment length."));
        if (!r.try_read_bytes(length, &buf))
            refuse(JPGPU_ERR_INVALID_DATA, at_offset(r.consumed_byte_count(), "Unexpected end of input data when reading segment content."));
        put_length(cur, length);
        cur.append((const char *)buf, length);
        if (bytes) *bytes = buf;
        if (n) *n = length;
    };
    bool eoi = false, dht_written = false, dqt_written = false;
    while (!eoi && !r.is_empty()) {
        int marker;
        if (!r.try_read_marker(&marker)) refuse(JPGPU_ERR_INVALID_DATA, at_offset(r.consumed_byte_count(), "No marker found."));
        switch (marker) {
        case 0xD8: put_marker(cur, marker); break;
        case 0xE0:
        case 0xC0:
        case 0xC1:
            put_marker(cur, marker);
            copy_segment(nullptr, nullptr);
            break;
        case 0xC2: {  // ProcessFrameHeader, then "not supported" (:580-582).  Reached on baseline files too: the walk does not
                      // skip the DQT / DHT payloads, and a quantisation table may hold the bytes FF C2 (low qualities)
            uint16_t length;
            const uint8_t *buf;
            if (!r.try_read_length(&length))
                refuse(JPGPU_ERR_INVALID_DATA, at_offset(r.consumed_byte_count(), "Unexpected end of input data when reading segment length."));
            if (!r.try_read_bytes(length, &buf))
                refuse(JPGPU_ERR_INVALID_DATA, at_offset(r.consumed_byte_count(), "Unexpected end of input data when reading segment content."));
            FrameHeader fh;
            int consumed = 0;
            if (!FrameHeader::try_parse(buf, length, false, &fh, &consumed))
                refuse(JPGPU_ERR_INVALID_DATA, at_offset(r.consumed_byte_count() - length + consumed, "Failed to parse frame header."));
            refuse(JPGPU_ERR_INVALID_DATA, "Progressive JPEG is not supported currently.");
        }
        case 0xC3: case 0xC5: case 0xC6: case 0xC7: case 0xC9: case 0xCA: case 0xCB: case 0xCD: case 0xCE: case 0xCF:
            refuse(JPGPU_ERR_INVALID_DATA,
                   at_offset(r.consumed_byte_count(), ("This type of JPEG stream is not supported (StartOfFrame" + std::to_string(marker - 0xC0) + ").").c_str()));
        case 0xC4:  // the segment's own bytes are not skipped: the marker search runs through them (:596-602)
            if (!dht_written) {
                flush();
                p.pieces.push_back({Piece::kHuffmanTables, std::string()});
                dht_written = true;
            }
            break;
        case 0xDB:
            if (!dqt_written) {
                put_marker(cur, 0xDB);
                uint16_t total = 0;
                for (const QuantTable &q : quant) total = (uint16_t)(total + (uint8_t)(q.precision == 0 ? 65 : 129));
                put_length(cur, total);
                for (const QuantTable &q : quant) {
                    cur.push_back((char)(q.precision << 4 | (q.identifier & 0xF)));
                    for (int k = 0; k < 64; k++) {
                        if (q.precision != 0) cur.push_back((char)(q.elements[k] >> 8));
                        cur.push_back((char)q.elements[k]);
                    }
                }
                dqt_written = true;
            }
            break;
        case 0xDA: {
            put_marker(cur, marker);
            copy_segment(nullptr, nullptr);
            flush();
            p.pieces.push_back({Piece::kEntropy, std::string()});
            const size_t end = scan_end(r.remaining_bytes(), (size_t)r.remaining_byte_count());
            {
                // More RSTn markers than the frame's intervals need: the scan stops behind its last MCU and the walk meets
                // the others as markers of their own, copied one by one (:603-612).  The first of them depends on where
                // the bit reader stood (transcode_kernel decides it), the rest are certain.
                int max_h = 1, max_v = 1;
                for (const FrameComponent &c : frame.components) {
                    max_h = std::max(max_h, (int)c.h);
                    max_v = std::max(max_v, (int)c.v);
                }
                const uint64_t mcus = (uint64_t)((frame.samples_per_line + 8 * max_h - 1) / (8 * max_h)) * ((frame.lines + 8 * max_v - 1) / (8 * max_v));
                const uint64_t intervals = dri_at_scan ? (mcus + dri_at_scan - 1) / dri_at_scan : 1;
                p.rsts_in_scan = end_rsts_.size();
                for (size_t k = (size_t)intervals; k < end_rsts_.size(); k++) put_marker(cur, (uint8_t)end_rsts_[k]);
            }
            r.try_advance((int)end + (swallow_terminator && end < (size_t)r.remaining_byte_count() ? 1 : 0));
            break;
        }
        case 0xD0: case 0xD1: case 0xD2: case 0xD3: case 0xD4: case 0xD5: case 0xD6: case 0xD7: put_marker(cur, marker); break;
        case 0xD9:
            put_marker(cur, 0xD9);
            eoi = true;
            break;
        default:
            if (strip) {
                uint16_t length;
                if (!r.try_read_length(&length))
                    refuse(JPGPU_ERR_INVALID_DATA, at_offset(r.consumed_byte_count(), "Unexpected end of input data when reading segment length."));
                if (!r.try_advance(length))
                    refuse(JPGPU_ERR_INVALID_DATA, at_offset(r.consumed_byte_count(), "Unexpected end of input data when reading segment content."));
            } else {
                put_marker(cur, marker);
                copy_segment(nullptr, nullptr);
            }
            break;
        }
    }
    flush();
    } catch (const Refuse &e) {
        if (!after_scan || e.status == JPGPU_ERR_NOT_SUPPORTED) throw;
        p.late_status = e.status;
        p.late_detail = e.detail;
        p.late_error = e.msg;
        p.pieces.clear();
    }
}

int OptimizeBatch::upload(const uint8_t *const *jpeg, const size_t *len, int n, int strip) {
    if (n < 0 || (n > 0 && (!jpeg || !len))) return fail(JPGPU_ERR_ARGUMENT, "jpgpu_optimizer_upload: bad arguments");
    plans_.assign((size_t)n, Plan());
    ran_ = false;
    for (int i = 0; i < n; i++) {
        try {
            plan_file(plans_[i], jpeg[i], len[i], strip != 0);
        } catch (const Refuse &e) {
            plans_[i].status = e.status;
            plans_[i].detail = e.detail;
            plans_[i].error = e.msg;
            plans_[i].pieces.clear();
        }
        if (plans_[i].status == JPGPU_OK) {
            // the same walks with the reader resuming inside the scan's terminating marker: only the verdict is kept (a
            // walk that still goes through would write a different file: refused)
            Plan &p = plans_[i];
            Plan alt;
            p.swallow_status = JPGPU_ERR_NOT_SUPPORTED;
            p.swallow_detail = kDetailUnsupportedFrame;
            p.swallow_error = "A scan that leaves one byte unread in front of its terminating marker is not supported by the optimizer path.";
            try {
                plan_file(alt, jpeg[i], len[i], strip != 0, true);
                if (alt.late_status != JPGPU_OK) {
                    p.swallow_status = alt.late_status;
                    p.swallow_detail = alt.late_detail;
                    p.swallow_error = alt.late_error;
                }
            } catch (const Refuse &e) {
                if (e.status != JPGPU_ERR_NOT_SUPPORTED) {
                    p.swallow_status = e.status;
                    p.swallow_detail = e.detail;
                    p.swallow_error = e.msg;
                }
            }
        }
    }
    // the decoder-side parser resolves the scan (tables, geometry, restart interval) and lays the files out in HBM
    batch_.set_entropy_only(true);
    {
        std::vector<int> dri((size_t)n, 0);
        std::vector<uint8_t> no_subseq((size_t)n, 0);
        for (int i = 0; i < n; i++) {
            dri[i] = plans_[i].dri_at_scan;
            no_subseq[i] = plans_[i].rsts_in_scan != 0;  // RSTn behind the last block: the interval kernel knows what to do
        }
        batch_.set_preset_restart_intervals(std::move(dri));
        batch_.set_preset_no_subseq(std::move(no_subseq));
    }
    int rc = batch_.upload_files(jpeg, len, n, JPGPU_FMT_PLANAR_U8);
    if (rc != JPGPU_OK) return rc;
    scan_ids_.clear();
    work_.clear();
    sub_scan_ids_.clear();
    sub_work_.clear();
    for (int i = 0; i < n; i++) {
        Plan &p = plans_[i];
        if (p.status != JPGPU_OK) continue;
        const ImagePlan *img = batch_.image(i);
        if (img->status != JPGPU_OK) {
            p.status = img->status;
            p.detail = img->detail;
            p.error = img->error;
            continue;
        }
        if (img->jobs.size() != 1 || batch_.jobs_[img->jobs[0]].kind != kScanSequential) {
            p.status = JPGPU_ERR_NOT_SUPPORTED;
            p.detail = kDetailUnsupportedFrame;
            p.error = "Only single-scan baseline files are supported by the optimizer path.";
            continue;
        }
        p.job = img->jobs[0];
        const DevScan &s = batch_.h_scans_[p.job];
        if ((int)s.dri != p.dri_at_scan) {  // a DRI segment in front of the frame header that a later one overrides
            p.status = JPGPU_ERR_NOT_SUPPORTED;
            p.detail = kDetailUnsupportedFrame;
            p.error = "A restart interval that changes between the frame header and the scan is not supported by the optimizer path.";
            p.job = -1;
            continue;
        }
        p.by_subsequence = s.n_subs != 0;  // the batch marks DRI = 0 scans for the self-synchronising subsequence machinery
                                           // (not those with RSTn markers in the data: upload() asks for the interval path)
        if (p.by_subsequence) {
            sub_scan_ids_.push_back((uint32_t)p.job);
            for (uint32_t first = 0; first < s.n_subs; first += 256) sub_work_.push_back({(uint32_t)p.job, first});
        } else {
            scan_ids_.push_back((uint32_t)p.job);
            for (uint32_t first = 0; first < s.n_intervals; first += 256) work_.push_back({(uint32_t)p.job, first});
        }
    }
    struct Up {
        DevBuffer *buf;
        const void *src;
        size_t bytes, reserve;
    };
    const size_t n_jobs = batch_.h_scans_.size();
    const Up ups[] = {
        {&d_work_, work_.data(), work_.size() * sizeof(HuffWork), 0},
        {&d_scan_ids_, scan_ids_.data(), scan_ids_.size() * sizeof(uint32_t), 0},
        {&d_hist_, nullptr, 0, n_jobs * kMaxHuffSlots * 256 * sizeof(uint32_t) + 256},
        {&d_enc_, nullptr, 0, n_jobs * kMaxHuffSlots * sizeof(EncHuffTable) + 256},
        {&d_sizes_, nullptr, 0, (size_t)batch_.total_ends_ * sizeof(uint32_t) + 256},
        {&d_offsets_, nullptr, 0, (size_t)batch_.total_ends_ * sizeof(uint64_t) + 256},
        {&d_base_, nullptr, 0, scan_ids_.size() * sizeof(uint64_t) + 256},
        {&d_totals_, nullptr, 0, scan_ids_.size() * sizeof(uint64_t) + 256},
        {&d_sub_work_, sub_work_.data(), sub_work_.size() * sizeof(HuffWork), 0},
        {&d_sub_scan_ids_, sub_scan_ids_.data(), sub_scan_ids_.size() * sizeof(uint32_t), 0},
        {&d_sub_bits_, nullptr, 0, (size_t)batch_.total_subs_ * sizeof(uint32_t) + 256},
        {&d_sub_bitoff_, nullptr, 0, (size_t)batch_.total_subs_ * sizeof(uint64_t) + 256},
        {&d_sub_totals_, nullptr, 0, sub_scan_ids_.size() * sizeof(uint64_t) + 256},
        {&d_scan_raw_off_, nullptr, 0, n_jobs * sizeof(uint64_t) + 256},
        {&d_simages_, nullptr, 0, sub_scan_ids_.size() * sizeof(DevEncImage) + 256},
        {&d_sout_len_, nullptr, 0, sub_scan_ids_.size() * sizeof(uint64_t) + 256},
    };
    for (const Up &u : ups) {
        hipError_t e = u.buf->reserve(std::max(u.bytes, u.reserve));
        if (e != hipSuccess) return hip_fail(e, "hipMalloc");
        if (u.bytes) {
            e = hipMemcpyAsync(u.buf->ptr, u.src, u.bytes, hipMemcpyHostToDevice, ctx_->stream);
            if (e != hipSuccess) return hip_fail(e, "hipMemcpyAsync(optimizer work)");
        }
    }
    return JPGPU_OK;
}

int OptimizeBatch::run() {
    ran_ = false;
    if (!ev0_) {
        if (hipEventCreate(&ev0_) != hipSuccess || hipEventCreate(&ev1_) != hipSuccess) return fail(JPGPU_ERR_DEVICE, "hipEventCreate");
    }
    (void)hipEventRecord(ev0_, ctx_->stream);
    int rc = batch_.run_marker_index();
    if (rc != JPGPU_OK) return rc;
    const size_t n_jobs = batch_.h_scans_.size();
    const int n_work = (int)work_.size(), n_scans = (int)scan_ids_.size();
    const uint8_t *udata = (const uint8_t *)batch_.d_unstuffed_.ptr, *input = (const uint8_t *)batch_.d_input_.ptr;
    const DevScan *scans = (const DevScan *)batch_.d_scans_.ptr;
    DevScanStatus *status = (DevScanStatus *)batch_.d_status_.ptr;
    const DevHuffTable *pool = (const DevHuffTable *)batch_.d_huff_pool_.ptr;
    const uint32_t *ends_u = (const uint32_t *)batch_.d_ends_u_.ptr, *ends_raw = (const uint32_t *)batch_.d_ends_.ptr;
    const int n_slots = batch_.n_huff_slots_;
    // ---- Scan(): IncrementCodeCount for every symbol
    hipError_t e = hipMemsetAsync(d_hist_.ptr, 0, n_jobs * kMaxHuffSlots * 256 * sizeof(uint32_t), ctx_->stream);
    if (e != hipSuccess) return hip_fail(e, "hipMemsetAsync(histograms)");
    e = launch_transcode(ctx_->stream, 0, udata, input, scans, (const HuffWork *)d_work_.ptr, n_work, ends_u, ends_raw, status, pool,
                         (uint32_t *)d_hist_.ptr, nullptr, nullptr, nullptr, nullptr, n_slots);
    if (e != hipSuccess) return hip_fail(e, "transcode_kernel<count>");
    // scans without restart intervals: the decoder's self-synchronising subsequences, then the same three passes per subsequence
    const int n_sub_work = (int)sub_work_.size(), n_sub_scans = (int)sub_scan_ids_.size();
    const uint32_t *exit_state = nullptr, *first_block = nullptr;
    if (n_sub_scans) {
        rc = batch_.run_subseq_sync(&exit_state, &first_block);
        if (rc != JPGPU_OK) return rc;
        e = launch_subseq_transcode(ctx_->stream, 0, udata, scans, (const HuffWork *)d_sub_work_.ptr, n_sub_work, ends_u, status, pool, exit_state,
                                    first_block, (uint32_t *)d_hist_.ptr, nullptr, nullptr, nullptr, nullptr, nullptr, n_slots);
        if (e != hipSuccess) return hip_fail(e, "subseq_transcode_kernel<count>");
    }
    h_hist_.assign(n_jobs * kMaxHuffSlots * 256, 0);
    e = hipMemcpyAsync(h_hist_.data(), d_hist_.ptr, h_hist_.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx_->stream);
    if (e != hipSuccess) return hip_fail(e, "hipMemcpyAsync(histograms)");
    e = hipStreamSynchronize(ctx_->stream);
    if (e != hipSuccess) return hip_fail(e, "hipStreamSynchronize");
    // ---- BuildTables (:462): one table per (class, identifier) in builder-creation order == the job's slot order
    std::vector<EncHuffTable> enc(n_jobs * kMaxHuffSlots);
    memset(enc.data(), 0, enc.size() * sizeof(EncHuffTable));
    for (Plan &p : plans_) {
        if (p.status != JPGPU_OK || p.job < 0) continue;
        const ScanJob &job = batch_.jobs_[p.job];
        std::string body;
        for (int t = 0; t < job.n_huff; t++) {
            std::vector<OptimalCode> codes;
            const uint32_t *freq = &h_hist_[((size_t)p.job * kMaxHuffSlots + t) * 256];
            if (!build_optimal_table(freq, &codes, most_optimal_)) {
                // No symbol counted: a failing scan (reported from the device status below, which comes first), or a component
                // without blocks -- a sampling factor of zero, which JpegOptimizer never divides by -- whose builders stay empty:
                // Build() throws "No symbol is recorded." (tests/golden/stress/optimizer_component_without_blocks_421.jpg).
                // Symbols counted and no table: the reference's byte counters overflowed (build_optimal_table) -- its Build() dies
                // with IndexOutOfRangeException.
                bool any = false;
                for (int sym = 0; sym < 256; sym++) any |= freq[sym] != 0;
                if (!p.build_failed) {  // behind the scan's own failures (BuildTables runs after ProcessScanBaseline, :462), in front of
                                        // what the markers behind the scan throw (late_status as the walk left it); first builder wins
                    p.build_failed = true;
                    p.late_status = JPGPU_ERR_INVALID_OPERATION;
                    p.late_detail = 0;
                    p.late_error = any ? "Index was outside the bounds of the array." : "No symbol is recorded.";
                }
                continue;
            }
            EncHuffTable &et = enc[(size_t)p.job * kMaxHuffSlots + t];
            // GetCode(symbol) = codes[_symbolMap[symbol]], and _symbolMap is 0 for symbols without a code (JpegHuffmanEncodingTable.cs:18-33, 90-96)
            for (int sym = 0; sym < 256; sym++) {
                et.code[sym] = codes[0].code;
                et.len[sym] = codes[0].length;
            }
            for (const OptimalCode &c : codes) {
                et.code[c.symbol] = c.code;
                et.len[c.symbol] = c.length;
            }
            // JpegHuffmanEncodingTableCollection.Write (:172-190) + JpegHuffmanEncodingTable.TryWrite (:37-79)
            body.push_back((char)(job.huff_copy[t].table_class << 4 | (job.huff_copy[t].identifier & 0xF)));
            for (int l = 1; l <= 16; l++) {
                int count = 0;
                for (const OptimalCode &c : codes) count += c.length == l;
                body.push_back((char)count);
            }
            for (const OptimalCode &c : codes) body.push_back((char)c.symbol);
        }
        p.dht.clear();
        put_marker(p.dht, 0xC4);
        put_length(p.dht, (uint16_t)body.size());
        p.dht += body;
    }
    e = hipMemcpyAsync(d_enc_.ptr, enc.data(), enc.size() * sizeof(EncHuffTable), hipMemcpyHostToDevice, ctx_->stream);
    if (e != hipSuccess) return hip_fail(e, "hipMemcpyAsync(encoding tables)");
    // ---- Optimize(): size of every interval, offsets, bytes
    e = launch_transcode(ctx_->stream, 1, udata, input, scans, (const HuffWork *)d_work_.ptr, n_work, ends_u, ends_raw, status, pool, nullptr,
                         (const EncHuffTable *)d_enc_.ptr, (uint32_t *)d_sizes_.ptr, nullptr, nullptr, n_slots);
    if (e != hipSuccess) return hip_fail(e, "transcode_kernel<measure>");
    e = launch_transcode_offsets(ctx_->stream, scans, (const uint32_t *)d_scan_ids_.ptr, n_scans, (const uint32_t *)d_sizes_.ptr, nullptr,
                                 nullptr, (uint64_t *)d_totals_.ptr);
    if (e != hipSuccess) return hip_fail(e, "transcode_offsets_kernel");
    std::vector<uint64_t> totals((size_t)n_scans), base((size_t)n_scans);
    if (n_scans) {
        e = hipMemcpyAsync(totals.data(), d_totals_.ptr, totals.size() * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx_->stream);
        if (e != hipSuccess) return hip_fail(e, "hipMemcpyAsync(totals)");
    }
    std::vector<uint64_t> sub_total_bits((size_t)n_sub_scans);
    if (n_sub_scans) {
        e = launch_subseq_transcode(ctx_->stream, 1, udata, scans, (const HuffWork *)d_sub_work_.ptr, n_sub_work, ends_u, status, pool, exit_state,
                                    first_block, nullptr, (const EncHuffTable *)d_enc_.ptr, (uint32_t *)d_sub_bits_.ptr, nullptr, nullptr, nullptr,
                                    n_slots);
        if (e != hipSuccess) return hip_fail(e, "subseq_transcode_kernel<measure>");
        e = launch_subseq_bit_offsets(ctx_->stream, scans, (const uint32_t *)d_sub_scan_ids_.ptr, n_sub_scans, (const uint32_t *)d_sub_bits_.ptr,
                                      (uint64_t *)d_sub_bitoff_.ptr, (uint64_t *)d_sub_totals_.ptr);
        if (e != hipSuccess) return hip_fail(e, "subseq_bit_offsets_kernel");
        e = hipMemcpyAsync(sub_total_bits.data(), d_sub_totals_.ptr, sub_total_bits.size() * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx_->stream);
        if (e != hipSuccess) return hip_fail(e, "hipMemcpyAsync(bit totals)");
    }
    e = hipStreamSynchronize(ctx_->stream);
    if (e != hipSuccess) return hip_fail(e, "hipStreamSynchronize");
    uint64_t out_bytes = 0;
    for (int k = 0; k < n_scans; k++) {
        base[k] = out_bytes;
        out_bytes = (out_bytes + totals[k] + 255) & ~(uint64_t)255;
    }
    e = d_out_.reserve((size_t)out_bytes + 256);
    if (e != hipSuccess) return hip_fail(e, "hipMalloc(optimizer output)");
    if (n_scans) {
        e = hipMemcpyAsync(d_base_.ptr, base.data(), base.size() * sizeof(uint64_t), hipMemcpyHostToDevice, ctx_->stream);
        if (e != hipSuccess) return hip_fail(e, "hipMemcpyAsync(bases)");
    }
    e = launch_transcode_offsets(ctx_->stream, scans, (const uint32_t *)d_scan_ids_.ptr, n_scans, (const uint32_t *)d_sizes_.ptr,
                                 (const uint64_t *)d_base_.ptr, (uint64_t *)d_offsets_.ptr, nullptr);
    if (e != hipSuccess) return hip_fail(e, "transcode_offsets_kernel");
    e = launch_transcode(ctx_->stream, 2, udata, input, scans, (const HuffWork *)d_work_.ptr, n_work, ends_u, ends_raw, status, pool, nullptr,
                         (const EncHuffTable *)d_enc_.ptr, (uint32_t *)d_sizes_.ptr, (const uint64_t *)d_offsets_.ptr, (uint8_t *)d_out_.ptr, n_slots);
    if (e != hipSuccess) return hip_fail(e, "transcode_kernel<emit>");
    std::vector<DevEncImage> simages((size_t)n_sub_scans);
    std::vector<uint64_t> sout_len((size_t)n_sub_scans);
    if (n_sub_scans) {
        // raw (unstuffed) bit buffers + the encoder's stuffing stage (E4): descriptors in the encoder's image form
        std::vector<uint64_t> scan_raw_off(n_jobs, 0);
        std::vector<EncWork> work_chunk;
        uint64_t raw_off = 0, out_off = 0;
        uint32_t chunk_off = 0;
        memset(simages.data(), 0, simages.size() * sizeof(DevEncImage));
        for (int k2 = 0; k2 < n_sub_scans; k2++) {
            const uint64_t raw_len = (sub_total_bits[k2] + 7) / 8;
            DevEncImage &im = simages[k2];
            im.raw_off = raw_off;
            im.out_off = out_off;
            im.chunk_off = chunk_off;
            im.header_len = 0;
            scan_raw_off[sub_scan_ids_[k2]] = raw_off;
            const uint32_t chunks = std::max<uint32_t>(1u, (uint32_t)((raw_len + kEncStuffChunk - 1) / kEncStuffChunk));
            for (uint32_t c = 0; c < chunks; c++) work_chunk.push_back({(uint32_t)k2, c});
            chunk_off += chunks;
            raw_off = (raw_off + raw_len + 64 + 255) & ~(uint64_t)255;
            out_off = (out_off + 2 * raw_len + 2 + 64 + 255) & ~(uint64_t)255;  // every byte may need stuffing
        }
        const struct {
            DevBuffer *buf;
            size_t bytes;
        } grow[] = {{&d_raw_, (size_t)raw_off + 256}, {&d_sout_, (size_t)out_off + 256}, {&d_chunk_ff_, (size_t)chunk_off * sizeof(uint32_t) + 256},
                    {&d_swork_chunk_, work_chunk.size() * sizeof(EncWork) + 16}};
        for (const auto &g : grow) {
            e = g.buf->reserve(g.bytes);
            if (e != hipSuccess) return hip_fail(e, "hipMalloc(optimizer raw buffers)");
        }
        e = hipMemsetAsync(d_raw_.ptr, 0, (size_t)raw_off + 256, ctx_->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(d_scan_raw_off_.ptr, scan_raw_off.data(), scan_raw_off.size() * sizeof(uint64_t), hipMemcpyHostToDevice, ctx_->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(d_simages_.ptr, simages.data(), simages.size() * sizeof(DevEncImage), hipMemcpyHostToDevice, ctx_->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(d_swork_chunk_.ptr, work_chunk.data(), work_chunk.size() * sizeof(EncWork), hipMemcpyHostToDevice, ctx_->stream);
        if (e != hipSuccess) return hip_fail(e, "hipMemcpyAsync(optimizer raw descriptors)");
        e = launch_subseq_transcode(ctx_->stream, 2, udata, scans, (const HuffWork *)d_sub_work_.ptr, n_sub_work, ends_u, status, pool, exit_state,
                                    first_block, nullptr, (const EncHuffTable *)d_enc_.ptr, (uint32_t *)d_sub_bits_.ptr,
                                    (const uint64_t *)d_sub_bitoff_.ptr, (const uint64_t *)d_scan_raw_off_.ptr, (uint8_t *)d_raw_.ptr, n_slots);
        if (e != hipSuccess) return hip_fail(e, "subseq_transcode_kernel<emit>");
        e = launch_stuff(ctx_->stream, (const DevEncImage *)d_simages_.ptr, (const EncWork *)d_swork_chunk_.ptr, (int)work_chunk.size(),
                         (const uint64_t *)d_sub_totals_.ptr, (const uint8_t *)d_raw_.ptr, nullptr /* no restart marks: restart_interval is 0 */,
                         (uint32_t *)d_chunk_ff_.ptr, (uint8_t *)d_sout_.ptr,
                         (uint64_t *)d_sout_len_.ptr);
        if (e != hipSuccess) return hip_fail(e, "stuff kernels");
        e = hipMemcpyAsync(sout_len.data(), d_sout_len_.ptr, sout_len.size() * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx_->stream);
        if (e != hipSuccess) return hip_fail(e, "hipMemcpyAsync(stuffed lengths)");
    }
    (void)hipEventRecord(ev1_, ctx_->stream);
    e = hipStreamSynchronize(ctx_->stream);
    if (e != hipSuccess) return hip_fail(e, "hipStreamSynchronize");
    (void)hipEventElapsedTime(&last_ms_, ev0_, ev1_);
    int k = 0, ks = 0;
    for (Plan &p : plans_) {
        if (p.status != JPGPU_OK || p.job < 0) continue;
        if (p.by_subsequence) {
            p.entropy_off = simages[ks].out_off;
            p.entropy_len = sout_len[ks] >= 2 ? sout_len[ks] - 2 : 0;  // the stuffing stage closes with EOI: that belongs to the host pieces here
            ks++;
        } else {
            p.entropy_off = base[k];
            p.entropy_len = totals[k];
            k++;
        }
        p.out_len = 0;
        for (const Piece &pc : p.pieces)
            p.out_len += pc.kind == Piece::kBytes ? pc.bytes.size() : (pc.kind == Piece::kHuffmanTables ? p.dht.size() : p.entropy_len);
    }
    ran_ = true;
    return JPGPU_OK;
}

int OptimizeBatch::result(int i, jpgpu_image_result *res, size_t *out_len) {
    if (i < 0 || i >= (int)plans_.size() || !res) return fail(JPGPU_ERR_ARGUMENT, "jpgpu_optimizer_result: bad index");
    memset(res, 0, sizeof *res);
    if (out_len) *out_len = 0;
    const Plan &p = plans_[i];
    if (p.status != JPGPU_OK) {
        res->status = p.status;
        res->detail = p.detail;
        ctx_->last_error = p.error;
        return JPGPU_OK;
    }
    if (!ran_) return fail(JPGPU_ERR_INVALID_OPERATION, "jpgpu_optimizer_result: run() has not completed");
    int rc = batch_.result(i, res);  // device-side scan errors, with the reference's exception classes
    if (rc != JPGPU_OK) return rc;
    if (res->status == JPGPU_OK) {
        // a restart check that meets EOI (early EOI, or an MCU count that is a multiple of DRI) makes Scan() return before
        // it builds its tables (:437-442); Optimize() then throws InvalidOperationException (:542-545)
        const DevScan &s = batch_.h_scans_[p.job];
        const DevScanStatus &st = batch_.h_status_[p.job];
        const bool check_saw_eoi = s.dri != 0 && st.terminator == 0xD9 &&
                                   (st.n_ends < s.n_intervals || (st.n_ends == s.n_intervals && s.restart_check_at_end != 0));
        if (check_saw_eoi) {
            res->status = JPGPU_ERR_INVALID_OPERATION;
            ctx_->last_error = "Operation is not valid due to the current state of the object.";
        } else if (st.terminator == 0 && st.pad[2] >= 8) {
            // the data ran out behind the scan with whole bytes left and no marker among them: Scan()'s marker loop fails (:82-86)
            res->status = JPGPU_ERR_INVALID_DATA;
            res->detail = kDetailBadHeader;
            ctx_->last_error = "Failed to decode JPEG data at offset " + std::to_string(batch_.image(i)->file_len) + ". No marker found.";
        } else if (st.terminator != 0 && (st.terminator & 0xF8u) != 0xD0u && (st.pad[2] >> 3) == 1) {
            res->status = p.swallow_status;
            res->detail = p.swallow_detail;
            ctx_->last_error = p.swallow_error;
        } else if (p.late_status != JPGPU_OK) {
            res->status = p.late_status;
            res->detail = p.late_detail;
            ctx_->last_error = p.late_error;
        }
    }
    if (res->status == JPGPU_OK && out_len) *out_len = (size_t)p.out_len;
    return JPGPU_OK;
}

int OptimizeBatch::download(int i, void *dst, size_t cap) {
    jpgpu_image_result res;
    size_t n = 0;
    int rc = result(i, &res, &n);
    if (rc != JPGPU_OK) return rc;
    if (res.status != JPGPU_OK) return fail(res.status, ctx_->last_error);
    if (!dst || cap < n) return fail(JPGPU_ERR_ARGUMENT, "jpgpu_optimizer_download: buffer too small");
    const Plan &p = plans_[i];
    uint8_t *o = (uint8_t *)dst;
    for (const Piece &pc : p.pieces) {
        if (pc.kind == Piece::kBytes) {
            memcpy(o, pc.bytes.data(), pc.bytes.size());
            o += pc.bytes.size();
        } else if (pc.kind == Piece::kHuffmanTables) {
            memcpy(o, p.dht.data(), p.dht.size());
            o += p.dht.size();
        } else if (p.entropy_len) {
            const uint8_t *src = (const uint8_t *)(p.by_subsequence ? d_sout_.ptr : d_out_.ptr) + p.entropy_off;
            hipError_t e = hipMemcpy(o, src, (size_t)p.entropy_len, hipMemcpyDeviceToHost);
            if (e != hipSuccess) return hip_fail(e, "hipMemcpy(optimizer output)");
            o += p.entropy_len;
        }
    }
    return JPGPU_OK;
}

bool OptimizeBatch::statistics(int i, int t, uint8_t *table_class, uint8_t *identifier, uint32_t counts[256]) const {
    if (i < 0 || i >= (int)plans_.size() || !ran_) return false;
    const Plan &p = plans_[i];
    if (p.status != JPGPU_OK || p.job < 0) return false;
    const ScanJob &job = batch_.jobs_[p.job];
    if (t < 0 || t >= job.n_huff) return false;
    *table_class = job.huff_copy[t].table_class;
    *identifier = job.huff_copy[t].identifier;
    memcpy(counts, &h_hist_[((size_t)p.job * kMaxHuffSlots + t) * 256], 256 * sizeof(uint32_t));
    return true;
}

}  // namespace jpgpu
