// jpeglibrary_amd/csrc/kernels_device.h -- device-side helpers shared by the kernel translation units (k1_markers.hip,
// k2_huffman.hip, k2s_subseq.hip, k2p_progressive.hip, k3_idct.hip, kt_transcode.hip): wave primitives, the unstuffed bit reader,
// the K2 family's LDS ring / lookup primitives, the subsequence state word.  Everything here is __forceinline__.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

#include "common.h"
#include "kernels.h"

namespace jpgpu {

// ------------------------------------------------------------------------------------------------
// helpers
// ------------------------------------------------------------------------------------------------

__device__ __forceinline__ uint32_t lane_id() { return __lane_id(); }

__device__ __forceinline__ uint32_t wave_reduce_min(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        uint32_t t = __shfl_xor(v, o, 64);
        v = t < v ? t : v;
    }
    return v;
}
__device__ __forceinline__ uint32_t wave_reduce_max_i(int32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        int32_t t = __shfl_xor(v, o, 64);
        v = t > v ? t : v;
    }
    return (uint32_t)v;
}
// DPP forms of the wave-wide sums (all 64 lanes must be active): data-parallel-primitive operands move values between lanes
// inside the VALU, no LDS crossbar round trip per step as with ds_bpermute (__shfl_*).  Control codes: row_shr:n = 0x110 + n,
// row_bcast:15 = 0x142, row_bcast:31 = 0x143, wave_shl:1 = 0x130, wave_shr:1 = 0x138; lanes without a source receive 0.
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ uint32_t dpp0(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xF, false);
}
__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v) {
    v += dpp0<0x111>(v);       // inside each row of 16 lanes
    v += dpp0<0x112>(v);
    v += dpp0<0x114>(v);
    v += dpp0<0x118>(v);
    v += dpp0<0x142, 0xA>(v);  // rows 1 and 3 take the total of the row before them
    v += dpp0<0x143, 0xC>(v);  // rows 2 and 3 take the total of rows 0-1
    return v;
}
// sum over the wave, the same value in every lane (as a scalar)
__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_readlane((int)wave_inclusive_scan(v), 63);
}

__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ uint32_t lane_get(uint32_t v, uint32_t l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)l); }
__device__ __forceinline__ uint32_t mbcnt64(uint64_t m) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}
__device__ __forceinline__ uint32_t lane_put(uint32_t old, uint32_t v, uint32_t l) {
    // (no clang builtin for v_writelane in ROCm 7.2, and two scalar operands need M0 on gfx9; this runs once per long code:
    // a compare and a select instead of hand-written M0 traffic)
    return __lane_id() == l ? v : old;
}


// Bit source of one lane == a fresh JpegBitReader positioned at the start of its restart interval (ref: JpegBitReader.cs),
// reading the UNSTUFFED copy written by K1: interval bytes [ustart, uend) followed by at least 16 one-bits.
//   hi   : the next 32 bits of the stream (bit 31 first), always fully valid after ub_consume
//   lo   : the bits after them, left aligned, lcnt of them valid (low bits zero)
//   rem  : real data bits left in the interval == the reference's "bits available"; reads past them see the ones
//          padding, which is exactly PeekBits(16)'s padding (JpegBitReader.cs:163-167); once rem is 0 every peek is 0xFFFF
// Words come from a register queue of 2 x 16 bytes (qw current, nx prefetched a whole chunk ahead: memory latency is off
// the critical path).  All arithmetic is 32-bit.
struct UBits {
    const uint8_t *p;  // address of the next 16-byte chunk to prefetch
    uint4 qw, nx;
    uint32_t qn;
    uint32_t hi, lo;
    int32_t lcnt;
    int32_t rem;
};

__device__ __forceinline__ uint32_t ub_next_word(UBits &r) {
    const uint32_t w = r.qw.x;
    r.qw.x = r.qw.y;
    r.qw.y = r.qw.z;
    r.qw.z = r.qw.w;
    r.qn--;
    if (r.qn == 0) {
        r.qw = r.nx;
        __builtin_memcpy(&r.nx, r.p, 16);  // 4-byte aligned 16-byte load; buffers are padded
        r.p += 16;
        r.qn = 4;
    }
    return __builtin_bswap32(w);
}

// consume n bits, 1 <= n <= 32
__device__ __forceinline__ void ub_consume(UBits &r, uint32_t n) {
    r.hi = __builtin_amdgcn_alignbit(r.hi, r.lo, (32u - n) & 31u);  // n == 32 -> lo
    r.lo = n >= 32u ? 0u : (r.lo << n);
    r.lcnt -= (int32_t)n;
    if (r.lcnt < 0) {
        const uint32_t d = (uint32_t)(-r.lcnt);  // 1..32 low bits of hi are missing
        const uint32_t w = ub_next_word(r);
        r.hi |= w >> ((32u - d) & 31u);  // d == 32 -> w
        r.lo = d >= 32u ? 0u : (w << d);
        r.lcnt = 32 - (int32_t)d;
    }
}

__device__ __forceinline__ void ub_init(UBits &r, const uint8_t *ubase, uint32_t ustart, uint32_t uend) {
    const uint32_t a = ustart & ~3u;
    __builtin_memcpy(&r.qw, ubase + a, 16);
    __builtin_memcpy(&r.nx, ubase + a + 16, 16);
    r.p = ubase + a + 32;
    r.qn = 4;
    r.hi = ub_next_word(r);
    r.lo = ub_next_word(r);
    r.lcnt = 32;
    const uint32_t skip = (ustart & 3u) * 8;
    if (skip) ub_consume(r, skip);
    r.rem = (int32_t)((uend - ustart) * 8u);
}

// LDS image of a staged DevHuffTable
struct LdsHuff {
    const uint16_t *lut;
    const uint16_t *maxcode;
    const uint8_t *valoffset;
    const uint8_t *values;
};

__device__ __forceinline__ LdsHuff lds_huff(const uint8_t *tabs, uint32_t slot) {
    const uint8_t *t = tabs + slot * sizeof(DevHuffTable);
    LdsHuff h;
    h.lut = reinterpret_cast<const uint16_t *>(t);
    h.maxcode = reinterpret_cast<const uint16_t *>(t + offsetof(DevHuffTable, maxcode));
    h.valoffset = t + offsetof(DevHuffTable, valoffset);
    h.values = t + offsetof(DevHuffTable, values);
    return h;
}

__device__ __forceinline__ LdsHuff lds_huff16(const uint8_t *tabs, uint32_t off16) {
    const uint8_t *t = tabs + off16 * 16;
    LdsHuff h;
    h.lut = reinterpret_cast<const uint16_t *>(t);
    h.maxcode = reinterpret_cast<const uint16_t *>(t + offsetof(DevHuffTable, maxcode));
    h.valoffset = t + offsetof(DevHuffTable, valoffset);
    h.values = t + offsetof(DevHuffTable, values);
    return h;
}

// One Huffman symbol and its magnitude bits:
//   DecodeHuffmanCode (ref: ScanDecoder/JpegHuffmanScanDecoder.cs:81-88, JpegHuffmanDecodingTable.cs:73-113) followed by
//   ReceiveAndExtend (ref: ScanDecoder/JpegHuffmanScanDecoder.cs:100-115) when the category s is non-zero
//   (s = sym for a DC symbol, sym & 15 for an AC symbol).
// Returns 0, or the failure detail.  value = extended magnitude (0 when s == 0).
template <class R>
__device__ __forceinline__ uint32_t ub_symbol(R &r, const LdsHuff &h, bool is_dc, bool closed_by_marker, uint32_t &sym_out,
                                              int32_t &value) {
    const uint32_t code16 = r.rem > 0 ? (r.hi >> 16) : 0xFFFFu;
    const uint32_t e = h.lut[code16 >> (16 - kHuffLutBits)];
    uint32_t size = e >> 8, sym = e & 0xFF;
    if (size == 0) {
        size = kHuffLutBits + 1;
        while (code16 > h.maxcode[size]) size++;  // maxcode[17] = 0xFFFF terminates
        if (size > 16) return kDetailInvalidHuffmanCode;
        sym = h.values[(h.valoffset[size] + (code16 >> (16 - size))) & 0xFF];
    }
    sym_out = sym;
    const uint32_t s = is_dc ? sym : (sym & 15u);
    // advance Math.Min(entry.CodeSize, bitsRead)
    r.rem = r.rem > (int32_t)size ? r.rem - (int32_t)size : 0;
    value = 0;
    if (s != 0) {
        if (s > 16u) return kDetailInvalidHuffmanCode;  // categories above 16 are outside the verified envelope
        if ((int32_t)s > r.rem) return (r.rem == 0 && closed_by_marker) ? kDetailMarkerInData : kDetailStreamEnded;
        const int32_t v = (int32_t)__builtin_amdgcn_ubfe(r.hi, 32u - size - s, s);
        value = v - ((((v + v) >> s) - 1) & ((1 << s) - 1));  // Extend(v, nbits)
        r.rem -= (int32_t)s;
    }
    ub_consume(r, size + s);
    return 0;
}

// LDS staging of one wave: 64 blocks x 128 B, 16-byte chunks XOR-swizzled so that both the per-lane
// scattered 2-byte stores and the block-major 16-byte flush reads are (nearly) bank-conflict free.
__device__ __forceinline__ uint32_t stage_addr(uint32_t blk, uint32_t coef_index) {
    const uint32_t chunk = (coef_index >> 3) ^ ((blk >> 1) & 7);
    return blk * 128 + chunk * 16 + (coef_index & 7) * 2;
}

// DevScanStatus::pad[1] of a sequential scan that fails in block `block` (scan order); see common.h
__device__ __forceinline__ uint32_t fail_block_word(uint64_t block) { return kFailBlockBase - (uint32_t)(block < 0xFFFFFFF0ull ? block : 0xFFFFFFF0ull); }

// Restart check after an interval (ref: ScanDecoder/JpegHuffmanBaselineScanDecoder.cs:139-163): runs after every completed
// interval except a final partial one.  AdvanceAlignByte + TryReadMarker: no whole byte may be left before the closing
// marker, and the marker must be RSTn (continue) or EOI (return early).  Returns the error code word or kNoError.
__device__ __forceinline__ uint32_t restart_check(const DevScan &s, const DevScanStatus &st, DevScanStatus *status_out, uint32_t interval,
                                                  uint32_t n_ends, uint32_t n_intervals, uint32_t dri_eff, int32_t rem, uint32_t err) {
    if (err != 0) return (interval << 8) | err;
    // bits left behind the scan's last block when the scan's terminating marker closes that interval: the host needs the
    // whole bytes among them for the reader position the reference resumes its marker walk from (:167-176)
    if (interval == n_intervals - 1) status_out->pad[2] = rem > 0 ? (uint32_t)rem : 0u;
    const bool needs_check = s.dri != 0 && (interval < n_intervals - 1 || s.restart_check_at_end);
    if (!needs_check) return kNoError;
    uint32_t closing = 0xD0;  // entries before the last indexed one are RSTn by construction
    if (interval == n_ends - 1) closing = st.terminator;
    // "Expect restart marker." is thrown behind the interval's last MCU: every block of the interval has reached the writer
    const uint32_t behind = fail_block_word((uint64_t)(interval + 1) * dri_eff * s.blocks_per_mcu);
    const bool sequential = s.kind == kScanSequential;  // (a progressive scan's pad[1] is its progress word)
    if (rem >= 8) {
        if (sequential) atomicMax(&status_out->pad[1], behind);
        return (interval << 8) | kDetailExpectRestart;
    }
    if (closing == 0xD9) {
        if (interval < n_intervals - 1) atomicMin(&status_out->decoded_mcus, (interval + 1) * dri_eff);
        return kNoError;
    }
    if ((closing & 0xF8) != 0xD0) {
        if (sequential) atomicMax(&status_out->pad[1], behind);
        return (interval << 8) | kDetailExpectRestart;
    }
    return kNoError;
}

// K2: lanes of a wave decode block b of their MCU in lock-step into a shared LDS staging that is flushed per block as
// whole 128-byte lines of the coefficient buffer (zig-zag int16, MCU scan order).
//
// With 64 lanes in lock-step anything a lane does "rarely" happens in almost every iteration of the wave, so the symbol
// loop is written without per-lane state machines:
//  * the lane's bit source is a bit POSITION into a private 64-byte LDS ring of its unstuffed stream (17 words per lane:
//    16 ring words stored MSB-first + a mirror of word 0, so two consecutive words are always one ds_read2; stride 17
//    words keeps lanes on distinct banks).  Peeking 32 bits is bfe + address + ds_read2 + v_alignbit; consuming n bits is
//    one add.  The ring is topped up once per BLOCK (16 bytes, prefetched one block ahead into registers).
//  * one lookup of the next kK2LutBits bits returns total length (code + magnitude), code length, category and the
//    zig-zag advance in one 32-bit word; EOB and ZRL are ordinary entries whose "coefficient" is a zero stored where
//    nothing has been written yet, so the AC loop has no run/EOB branches.
//  * everything else -- codes longer than the lookup, the last bits of the interval where the reference's
//    "bits available" rules matter (JpegBitReader.cs:157-204), a ring that ran dry inside one block -- takes one
//    exec-masked exact path (k2_slow_symbol), the same decisions as ub_symbol.
constexpr int kK2RingStride = 68;                         // bytes per lane: 16 words + mirror of word 0
constexpr int kK2WaveBytes = 8192 + 64 * kK2RingStride;   // coefficient staging + rings
constexpr int kK2SmallBytes = 320;                        // maxcode[18] + valoffset[20] + values[256] + pad (round kernel's exact path)

// Lookups of the K2 family (K2, the K2S final pass), built once per upload for every table of the pool, as a DC and as an AC
// table (lut_pool_kernel), copied to LDS as they are:
//   L1  2^11 x u16, the next 11 bits:  AC  total bits | zig-zag advance << 6 | category << 12   (advance in coefficients: r + 1;
//                                          16 for any r != 0 with category 0; 63 = EOB, past the end from any position)
//                                      DC  total bits | category << 6
//                                      0 = not decided by 11 bits (a longer code); DC: 0x8000 = a category above 16
//   L2  256 x u16, the LONG codes:     entry j = the reference's maxcode walk on the 16 bits t16 + j, same format, 0 = no code.
//                                      Codes longer than the first level are the numerically largest ones of a canonical
//                                      table: for the standard tables the last 192 of the 65 536 16-bit values hold them all.
//                                      t16 = max(first value L1 does not decide, 65536 - 256).
//   header  t16
//   the reference's maxcode / valoffset / values (the exact walk: invalid codes, tables whose long codes leave the second level)
// Round 4: 32-bit entries took 34 KB of LDS for four tables and a long code cost the maxcode walk (six dependent LDS reads
// with 63 lanes waiting); now 19.8 KB, one more lookup for a long code, and the eleventh wave per workgroup.
// (Everything the symbol loop may touch stays in LDS: with the walk's arrays in global memory hipcc put a `s_waitcnt vmcnt(0)`
// at the head of the symbol loop -- every symbol waited for the previous block's coefficient stores: K2S final pass 7.8 -> 8.8 ms.)
constexpr int kK2LutBits = 11;
constexpr uint32_t kK2L1Bytes = 2u << kK2LutBits;
constexpr uint32_t kK2L2Entries = 256;
constexpr uint32_t kK2BadCat = 0x8000u;
constexpr uint32_t kK2TailBytes = 2u * kK2L2Entries + 16u + kK2SmallBytes;  // L2 | header | the reference's arrays
constexpr uint32_t kK2TabBytes = kK2L1Bytes + kK2TailBytes;  // the u16 image above (round 4-5 K2; still what the K2S ROUND kernel derives its lookups from)

// Round 6: the first level K2 and the K2S final pass keep in LDS carries the VALUES, and two symbols where two fit.
//   AC: 2^11 x u32, the next 11 bits;  DC: 2^9 x u32, the next 9 bits (the standard luminance DC codes end at 9 bits).
//   fast entry    n | nA << 4 | advA << 8 | advB << 14 | valA << 20 | valB << 26
//                 n = bits of everything the entry decodes (1..11), nA = bits of its FIRST symbol (code + magnitude), advA / advB =
//                 zig-zag advance of the first / second symbol in coefficients (r + 1; 16 for ZRL; 63 for EOB; advB = 0: the entry
//                 holds ONE symbol and valB = valA), valA / valB = the Extend()ed coefficients, six signed bits (categories 0..5:
//                 with the standard tables no longer symbol fits the index anyway).  A second symbol is only ever an AC symbol of the
//                 same table behind a first symbol that is not EOB; the lane checks that the first did not end the block.
//   medium entry  n = 0:  advA << 8 | category << 14 | (code + magnitude bits) << 19 -- the code fits the index, the magnitude does
//                 not (or is wider than six bits): one exec-masked stretch takes the value from the stream.
//   0x80000000    DC: a category above 16 (the exact path reports it);   0 = not decided by the index (a longer code: L2 / the walk).
// Image in the pool and in LDS: L1 | L2 (256 x u16, the old format: long codes) | header | the reference's arrays.
// Measured on the CPU before it was built (tools/microbench/k2_pairs/price_pairs_r06.txt): the wave's steps per block -- the
// longest of its 64 lanes' -- fall from 12.7 to 8.0 (Q75) and from 24.7 to 15.5 (Q90).
constexpr int kK2AcBits = 11, kK2DcBits = 9;
static_assert(kK2AcTabBytes == (4u << kK2AcBits) + kK2TailBytes && kK2DcTabBytes == (4u << kK2DcBits) + kK2TailBytes, "kernels.h holds the sizes for the host");
static_assert(kK2WaveLdsBytes == (uint32_t)kK2WaveBytes, "kernels.h holds the size for the host");
// pool, per table: the u16 images as an AC and as a DC table (round kernel), then the u32 images
constexpr uint32_t kPoolNewAc = 2u * kK2TabBytes, kPoolNewDc = kPoolNewAc + kK2AcTabBytes;
static_assert(kPoolNewDc + kK2DcTabBytes == kLutPoolBytesPerTable, "kernels.h sizes the pool");
constexpr uint32_t kK2MediumMask = 0xFFu;  // n | nA of an entry: zero = not a fast entry

struct K2Tab {
    const uint32_t *lut;
    const uint16_t *l2;
    const uint32_t *hdr;  // {t16, 0, 0, 0}
    const uint8_t *small;  // maxcode[18] | valoffset[20] | values[256]
    uint32_t shift;        // 32 - index bits
};

// off16: the table's LDS offset in 16-byte units (blk_info carries it: the images of a scan differ in size)
__device__ __forceinline__ K2Tab k2_tab_at(const uint8_t *tabs, uint32_t off16, bool is_dc) {
    const uint8_t *t = tabs + off16 * 16u;
    const uint32_t l1 = is_dc ? (4u << kK2DcBits) : (4u << kK2AcBits);
    K2Tab h;
    h.lut = reinterpret_cast<const uint32_t *>(t);
    h.l2 = reinterpret_cast<const uint16_t *>(t + l1);
    h.hdr = reinterpret_cast<const uint32_t *>(t + l1 + 2u * kK2L2Entries);
    h.small = t + l1 + 2u * kK2L2Entries + 16u;
    h.shift = is_dc ? 32u - kK2DcBits : 32u - kK2AcBits;
    return h;
}
// blk_info word: scan component | DC table offset << 8 | AC table offset << 20 (offsets in 16-byte units, 12 bits each)
__device__ __forceinline__ K2Tab k2_tab_dc(const uint8_t *tabs, uint32_t bi) { return k2_tab_at(tabs, (bi >> 8) & 0xFFFu, true); }
__device__ __forceinline__ K2Tab k2_tab_ac(const uint8_t *tabs, uint32_t bi) { return k2_tab_at(tabs, bi >> 20, false); }

// zig-zag advance (in int16 BYTES, i.e. 2 x coefficients) of an AC symbol: r + 1 coefficients for a non-zero category,
// 16 for ANY r != 0 with category 0 (ref: ...BaselineScanDecoder.cs:212-220), and "past the end" for EOB.
__device__ __forceinline__ uint32_t k2_ac_advance(uint32_t sym) {
    const uint32_t rr = sym >> 4;
    return (sym & 15u) ? 2u * (rr + 1u) : (rr ? 32u : 127u);
}

// next 32 bits of the lane's stream at bit position pm1 + 1
__device__ __forceinline__ uint32_t k2_window(const uint8_t *ring, int32_t pm1) {
    const uint32_t t = __builtin_amdgcn_ubfe((uint32_t)pm1, 5, 4);
    const uint32_t *p = reinterpret_cast<const uint32_t *>(ring + t * 4);
    return __builtin_amdgcn_alignbit(p[0], p[1], ~(uint32_t)pm1);
}

struct K2Feed {
    const uint8_t *gp;  // global address of the chunk after nx
    uint4 nx;           // chunk number `wr`, already loaded
    uint32_t wr;        // 16-byte chunks written to the ring so far (the ring holds chunks wr-4 .. wr-1)
};

__device__ __forceinline__ void k2_ring_write(uint8_t *ring, uint32_t slot, const uint4 &v) {
    uint32_t *rp = reinterpret_cast<uint32_t *>(ring + slot * 16);
    const uint32_t w0 = __builtin_bswap32(v.x);
    rp[0] = w0;
    rp[1] = __builtin_bswap32(v.y);
    rp[2] = __builtin_bswap32(v.z);
    rp[3] = __builtin_bswap32(v.w);
    if (slot == 0) reinterpret_cast<uint32_t *>(ring)[16] = w0;
}

// moves the prefetched chunk into the ring when the slot it replaces is no longer needed (the word before the current
// position must stay readable: k2_window reads it) and prefetches the next one
__device__ __forceinline__ void k2_topup(uint8_t *ring, K2Feed &f, int32_t pm1) {
    const int32_t rdc = (pm1 < 0 ? 0 : pm1) >> 7;
    if ((int32_t)f.wr < rdc + 4) {
        k2_ring_write(ring, f.wr & 3u, f.nx);
        f.wr++;
        __builtin_memcpy(&f.nx, f.gp, 16);
        f.gp += 16;
    }
}

// fast-path limit: a symbol of n bits at position pos may take the fast path when pos + n <= lim, which guarantees both
// "n real bits are available" and "the words the NEXT symbol reads (k2_symbol: up to 3 words past the one holding the bit
// before its position) are inside the ring"
__device__ __forceinline__ int32_t k2_limit(int32_t endpos, uint32_t wr) {
    const int32_t loaded = (int32_t)(wr * 128u) - 128;
    return endpos < loaded ? endpos : loaded;
}

// Exact symbol decode: DecodeHuffmanCode + ReceiveAndExtend with the reference's "bits available" rules (same decisions
// as ub_symbol).  ONE symbol, whatever the first-level entry holds.  Returns 0 or the failure detail; n = bits consumed, value,
// adv = zig-zag advance (AC, in coefficients).
__device__ __forceinline__ uint32_t k2_slow_symbol(uint8_t *ring, K2Feed &f, int32_t pm1, int32_t endpos, const K2Tab &h, bool is_dc,
                                                bool closed_by_marker, uint32_t &n, int32_t &value, uint32_t &adv) {
    const int32_t pos = pm1 + 1;
    while ((int32_t)(f.wr * 128u) < pos + 160) k2_topup(ring, f, pm1);  // always has room here
    const uint32_t hi = k2_window(ring, pm1);
    int32_t rem = endpos - pos;
    if (rem < 0) rem = 0;
    const uint32_t code16 = rem > 0 ? (hi >> 16) : 0xFFFFu;
    const uint32_t e = h.lut[code16 >> (h.shift - 16u)];
    uint32_t size, s;
    adv = 0;
    if ((e & 15u) != 0) {
        // a fast entry: its FIRST symbol.  The category of an Extend()ed value is the length of its magnitude.
        const int32_t va = (int32_t)__builtin_amdgcn_sbfe(e, 20, 6);
        s = va == 0 ? 0u : 32u - (uint32_t)__builtin_clz((uint32_t)(va < 0 ? -va : va));
        size = ((e >> 4) & 15u) - s;
        adv = (e >> 8) & 63u;
    } else if (e != 0 && (e >> 31) == 0) {  // the code fits the index, the magnitude does not
        s = (e >> 14) & 31u;
        size = ((e >> 19) & 63u) - s;
        adv = (e >> 8) & 63u;
    } else if (e != 0) {
        return kDetailInvalidHuffmanCode;  // DC categories above 16 are outside the verified envelope
    } else {
        uint32_t e2 = 0;
        const uint32_t t16 = h.hdr[0];
        if (code16 >= t16) {  // a long code: the second level (t16 >= 65536 - 256)
            e2 = h.l2[code16 - t16];
            if (is_dc && (e2 & kK2BadCat) != 0) return kDetailInvalidHuffmanCode;
        }
        if (e2 == 0) {
            // longer than the first level decides and not in the second: the reference's walk (an entry of the first level is empty
            // exactly when the code has more bits than the index, so the walk may start there)
            const uint16_t *maxcode = reinterpret_cast<const uint16_t *>(h.small);
            size = 33u - h.shift;
            while (code16 > maxcode[size]) size++;  // maxcode[17] = 0xFFFF terminates
            if (size > 16) return kDetailInvalidHuffmanCode;
            const uint32_t sym = h.small[56 + ((h.small[36 + size] + (code16 >> (16 - size))) & 0xFF)];
            s = is_dc ? sym : (sym & 15u);
            adv = (sym & 15u) ? (sym >> 4) + 1u : ((sym >> 4) ? 16u : 63u);
            if (s > 16u) return kDetailInvalidHuffmanCode;
        } else {
            s = is_dc ? ((e2 >> 6) & 31u) : (e2 >> 12);
            size = (e2 & 63u) - s;
            adv = (e2 >> 6) & 63u;
        }
    }
    rem = rem > (int32_t)size ? rem - (int32_t)size : 0;  // advance Math.Min(entry.CodeSize, bitsRead)
    value = 0;
    if (s != 0) {
        if ((int32_t)s > rem) return (rem == 0 && closed_by_marker) ? kDetailMarkerInData : kDetailStreamEnded;
        const int32_t v = (int32_t)__builtin_amdgcn_ubfe(hi, 32u - size - s, s);
        value = v - ((((v + v) >> s) - 1) & ((1 << s) - 1));  // Extend(v, nbits)
    }
    n = size + s;
    return 0;
}

// The lane's position and the three stream words around it: w0 holds the bit BEFORE the position (word q = pm1 >> 5),
// w1 and w2 follow.  A symbol is at most 32 bits, so q advances by at most one word per symbol; the word that would then
// be missing (q + 3) is read from the ring at the START of the step, off the dependency chain.
struct K2Pos {
    int32_t pm1;  // bit position - 1, relative to the lane's 4-byte aligned origin
    uint32_t w0, w1, w2;
};

__device__ __forceinline__ void k2_pos_init(K2Pos &p, const uint8_t *ring, int32_t pm1) {
    const uint32_t *r = reinterpret_cast<const uint32_t *>(ring);
    const int32_t q = pm1 >> 5;
    p.pm1 = pm1;
    p.w0 = r[q & 15];
    p.w1 = r[(q + 1) & 15];
    p.w2 = r[(q + 2) & 15];
}

// One STEP of a lane -- one symbol, or the two symbols of a pair entry --, fast path for every lane, then ONE branch the wave skips
// unless some lane needs more: a magnitude the index does not hold (medium), a long code (second level), the last bits of the
// interval, a ring that ran dry, a pair whose first symbol ended the block (exact path, one symbol); advances the position.
// i2 = 2 x zig-zag index of the next coefficient.  Out: ia = i2 + 2 x the first symbol's advance (the position behind it, in int16
// BYTES; EOB: past the end from any AC position), adv_b = the second symbol's advance in coefficients (0 and vb = va: one symbol).
// On failure the lane gets ia = i2 + 254 (leaves the AC loop), n = 0, values 0 and the detail code is ORed into `err` (touched on
// the exact path only: the fast path carries no instruction for it).
template <bool IS_DC>
__device__ __forceinline__ void k2_symbol(uint8_t *ring, K2Feed &f, K2Pos &p, int32_t endpos, int32_t &lim, const K2Tab &h, bool closed_by_marker,
                                          uint32_t i2, int32_t &va, int32_t &vb, uint32_t &ia, uint32_t &adv_b, uint32_t &err) {
    const uint32_t nxt = *reinterpret_cast<const uint32_t *>(ring + __builtin_amdgcn_ubfe((uint32_t)(p.pm1 + 96), 5, 4) * 4);
    const uint32_t hi = __builtin_amdgcn_alignbit(p.w0, p.w1, ~(uint32_t)p.pm1);
    // (the index as a shift by the table kind's constant and one shift-add for the address: hipcc's own form -- shift, mask, add --
    // is one instruction longer; the empty asm keeps it from folding the two shifts)
    uint32_t idx = hi >> (IS_DC ? 32u - kK2DcBits : 32u - kK2AcBits);
    asm volatile("" : "+v"(idx));
    const uint32_t e = *reinterpret_cast<const uint32_t *>(reinterpret_cast<const uint8_t *>(h.lut) + idx * 4u);
    uint32_t n = e & 15u;
    va = (int32_t)__builtin_amdgcn_sbfe(e, 20, 6);
    vb = (int32_t)e >> 26;
    ia = IS_DC ? i2 : i2 + 2u * __builtin_amdgcn_ubfe(e, 8, 6);
    adv_b = IS_DC ? 0u : __builtin_amdgcn_ubfe(e, 14, 6);
    // slow: not a fast entry (n - 1 is negative), the step does not fit below the limit, or a pair whose first symbol ends the
    // block (ia >= 128 with a second symbol behind it: that symbol is the next block's) -- one signed test
    const int32_t room = lim - (p.pm1 + 1) - (int32_t)n;
    const uint32_t pair_end = IS_DC ? 0u : ((127u - ia) & (0u - adv_b));
    const bool slow = (int32_t)((n - 1u) | (uint32_t)room | pair_end) < 0;
    if (slow) {  // exec-masked; the wave skips it when no lane is flagged
        const uint32_t ntot = (e >> 19) & 63u;
        if (n == 0 && (int32_t)e > 0 && lim - (p.pm1 + 1) - (int32_t)ntot >= 0) {
            // medium: the code fits the index, the magnitude comes from the stream (Extend(v, nbits)); its advance is where a fast entry's is
            const uint32_t cat = (e >> 14) & 31u;
            const int32_t raw = (int32_t)__builtin_amdgcn_ubfe(hi, 32u - ntot, cat);
            va = vb = raw - ((((raw + raw) >> cat) - 1) & ((1 << cat) - 1));
            n = ntot;
            adv_b = 0;
        } else {
            uint32_t adv = 0;
            const uint32_t e1 = k2_slow_symbol(ring, f, p.pm1, endpos, h, IS_DC, closed_by_marker, n, va, adv);
            vb = va;
            ia = i2 + adv * 2u;
            adv_b = 0;
            lim = k2_limit(endpos, f.wr);
            if (e1 != 0) {
                err |= e1;
                n = 0;
                va = vb = 0;
                ia = i2 + 254u;
            }
        }
    }
    const int32_t np = p.pm1 + (int32_t)n;
    const bool step = ((uint32_t)(np ^ p.pm1) >> 5) != 0;
    p.pm1 = np;
    p.w0 = step ? p.w1 : p.w0;
    p.w1 = step ? p.w2 : p.w1;
    p.w2 = step ? nxt : p.w2;
}
// (ONE symbol, the table kind known only per lane: the K2S final pass on its way to its first MCU)
__device__ __forceinline__ uint32_t k2_symbol_any(uint8_t *ring, K2Feed &f, K2Pos &p, int32_t endpos, int32_t &lim, const K2Tab &h, bool is_dc,
                                                  bool closed_by_marker, int32_t &value, uint32_t &adv2) {
    const uint32_t nxt = *reinterpret_cast<const uint32_t *>(ring + __builtin_amdgcn_ubfe((uint32_t)(p.pm1 + 96), 5, 4) * 4);
    const uint32_t hi = __builtin_amdgcn_alignbit(p.w0, p.w1, ~(uint32_t)p.pm1);
    const uint32_t e = h.lut[hi >> h.shift];
    uint32_t n = (e & 15u) != 0 ? ((e >> 4) & 15u) : 0u;  // the entry's first symbol
    value = (int32_t)__builtin_amdgcn_sbfe(e, 20, 6);
    adv2 = is_dc ? 0u : ((e >> 7) & 0x7Eu);
    uint32_t err = 0;
    const bool slow = (int32_t)((n - 1u) | (uint32_t)(lim - (p.pm1 + 1) - (int32_t)n)) < 0;
    if (slow) {
        uint32_t adv = 0;
        err = k2_slow_symbol(ring, f, p.pm1, endpos, h, is_dc, closed_by_marker, n, value, adv);
        adv2 = is_dc ? 0u : adv * 2u;
        lim = k2_limit(endpos, f.wr);
        if (err != 0) {
            n = 0;
            value = 0;
            adv2 = 254;
        }
    }
    const int32_t np = p.pm1 + (int32_t)n;
    const bool step = ((uint32_t)(np ^ p.pm1) >> 5) != 0;
    p.pm1 = np;
    p.w0 = step ? p.w1 : p.w0;
    p.w1 = step ? p.w2 : p.w1;
    p.w2 = step ? nxt : p.w2;
    return err;
}

// The K2 family's lookups (format above) for every table of the pool, as a DC table (odd blocks) and as an AC table.
// Image (table * 2 + is_dc) * kK2TabBytes: L1 | L2 | header.
constexpr int kLutPoolBits = kK2LutBits;
__device__ __forceinline__ uint32_t k2_entry_of(const DevHuffTable &h, uint32_t code16, bool is_dc, uint32_t max_size) {
    // the reference's Lookup on these 16 bits (JpegHuffmanDecodingTable.cs:73-113): first-level table, then the maxcode walk
    const uint32_t e9 = h.lut[code16 >> (16 - kHuffLutBits)];
    uint32_t size = e9 >> 8, sym = e9 & 0xFFu;
    if (size == 0) {
        size = kHuffLutBits + 1;
        while (code16 > h.maxcode[size]) size++;  // maxcode[17] = 0xFFFF terminates
        if (size > 16) return 0;
        sym = h.values[(h.valoffset[size] + (code16 >> (16 - size))) & 0xFF];
    }
    if (size > max_size) return 0;
    const uint32_t cat = is_dc ? sym : (sym & 15u);
    if (cat > 16u) return kK2BadCat;  // (the exact path reports it)
    if (is_dc) return (size + cat) | (cat << 6);
    const uint32_t adv = (sym & 15u) ? (sym >> 4) + 1u : ((sym >> 4) ? 16u : 63u);
    return (size + cat) | (adv << 6) | (cat << 12);
}
// First-level entry `idx` of the u32 lookup (format at kK2AcBits): the reference's Lookup on the index bits with ones behind them,
// then -- AC, first symbol not EOB -- once more on what the first symbol left of the index.
__device__ __forceinline__ uint32_t k2_fast_entry(const DevHuffTable &h, uint32_t idx, bool is_dc, uint32_t lb) {
    const uint32_t ones = (1u << (16u - lb)) - 1u;
    const uint32_t a = k2_entry_of(h, (idx << (16u - lb)) | ones, is_dc, lb);
    if (a == 0) return 0;
    if (is_dc && (a & kK2BadCat) != 0) return 0x80000000u;
    const uint32_t na = a & 63u, cat_a = is_dc ? ((a >> 6) & 31u) : (a >> 12), adv_a = is_dc ? 0u : ((a >> 6) & 63u);
    if (na > lb || cat_a > 5u) return (adv_a << 8) | (cat_a << 14) | (na << 19);  // medium
    const int32_t raw_a = (int32_t)((idx >> (lb - na)) & ((1u << cat_a) - 1u));
    const int32_t va = cat_a ? raw_a - ((((raw_a + raw_a) >> cat_a) - 1) & ((1 << cat_a) - 1)) : 0;
    uint32_t n = na, adv_b = 0;
    int32_t vb = va;
    if (!is_dc && adv_a != 63u && na < lb) {
        const uint32_t rem = lb - na;
        const uint32_t idx_b = ((idx << na) | ((1u << na) - 1u)) & ((1u << lb) - 1u);
        const uint32_t b = k2_entry_of(h, (idx_b << (16u - lb)) | ones, false, rem);
        if (b != 0) {
            const uint32_t nb = b & 63u, cat_b = b >> 12;
            if (nb <= rem && cat_b <= 5u) {
                const int32_t raw_b = (int32_t)((idx_b >> (lb - nb)) & ((1u << cat_b) - 1u));
                vb = cat_b ? raw_b - ((((raw_b + raw_b) >> cat_b) - 1) & ((1 << cat_b) - 1)) : 0;
                adv_b = (b >> 6) & 63u;
                n = na + nb;
            }
        }
    }
    return n | (na << 4) | (adv_a << 8) | (adv_b << 14) | (((uint32_t)va & 63u) << 20) | (((uint32_t)vb & 63u) << 26);
}
// the scan's tables as the K2 family keeps them in LDS: the pooled u32 images (lut_pool_kernel) copied as they are, one behind the other
__device__ __forceinline__ void k2_stage_scan_tables(const DevScan &s, const uint8_t *lut_pool, uint8_t *tabs, uint32_t *blk_info, int n_slots,
                                                     uint32_t nthreads) {
    const uint32_t tid = threadIdx.x;
    uint32_t off = 0, off16[kMaxHuffSlots];
#pragma unroll
    for (int sl = 0; sl < kMaxHuffSlots; sl++) {
        off16[sl] = off >> 4;
        if (sl >= n_slots) continue;
        const uint32_t pi = s.huff_pool[sl];
        if (pi == 0xFFFF) continue;
        bool is_dc = false;
        for (int c = 0; c < s.scan_components; c++) is_dc |= s.comp[c].dc_slot == sl;
        const uint32_t bytes = is_dc ? kK2DcTabBytes : kK2AcTabBytes;
        const uint4 *src = reinterpret_cast<const uint4 *>(lut_pool + (size_t)pi * kLutPoolBytesPerTable + (is_dc ? kPoolNewDc : kPoolNewAc));
        uint4 *dst = reinterpret_cast<uint4 *>(tabs + off);
        for (uint32_t i = tid; i < bytes / 16; i += nthreads) dst[i] = src[i];
        off += bytes;
    }
    // per block-in-MCU: scan component | DC table offset << 8 | AC table offset << 20 (kept in LDS: the block loop must not touch
    // global memory for it, a vector load there would wait for the coefficient stores of the previous block)
    if (tid < kMaxBlocksPerMcu) {
        const uint32_t ci = s.blk_comp[tid];
        uint32_t dc = 0, ac = 0;
#pragma unroll
        for (int sl = 0; sl < kMaxHuffSlots; sl++) {
            dc = s.comp[ci].dc_slot == sl ? off16[sl] : dc;
            ac = s.comp[ci].ac_slot == sl ? off16[sl] : ac;
        }
        blk_info[tid] = ci | (dc << 8) | (ac << 20);
    }
}

// DecodeHuffmanCode alone (ref: ScanDecoder/JpegHuffmanScanDecoder.cs:81-88): the symbol, no magnitude bits.
template <class R>
__device__ __forceinline__ uint32_t ub_huff(R &r, const LdsHuff &h, uint32_t &sym_out) {
    const uint32_t code16 = r.rem > 0 ? (r.hi >> 16) : 0xFFFFu;
    const uint32_t e = h.lut[code16 >> (16 - kHuffLutBits)];
    uint32_t size = e >> 8, sym = e & 0xFF;
    if (size == 0) {
        size = kHuffLutBits + 1;
        while (code16 > h.maxcode[size]) size++;  // maxcode[17] = 0xFFFF terminates
        if (size > 16) return kDetailInvalidHuffmanCode;
        sym = h.values[(h.valoffset[size] + (code16 >> (16 - size))) & 0xFF];
    }
    sym_out = sym;
    r.rem = r.rem > (int32_t)size ? r.rem - (int32_t)size : 0;  // advance Math.Min(entry.CodeSize, bitsRead)
    ub_consume(r, size);
    return 0;
}

// TryReadBits(n), 1 <= n <= 16 (ref: JpegBitReader.cs:190-204): false when fewer than n bits are left
template <class R>
__device__ __forceinline__ bool ub_try_read_bits(R &r, uint32_t n, uint32_t &bits) {
    if ((int32_t)n > r.rem) return false;
    bits = r.hi >> (32u - n);
    r.rem -= (int32_t)n;
    ub_consume(r, n);
    return true;
}

// subsequence length: (1 << DevScan::sub_shift) bits -- 1024 for small batches (more lanes), up to 4096 for large ones (a
// longer subsequence re-synchronises more often inside itself: fewer rounds until the exit states stop changing)
constexpr uint32_t kSubBad = 0x80000000u;  // the lane hit an invalid code / ran out of data under its entry state

// exit state word: overshoot (bits past the nominal end, 0..63) | b << 6 | k << 11 | kSubBad
__device__ __forceinline__ uint32_t sub_pack(uint32_t overshoot, uint32_t b, uint32_t k) { return overshoot | (b << 6) | (k << 11); }

__device__ __forceinline__ void sub_stage_tables(const DevScan &s, const DevHuffTable *huff_pool, uint8_t *tabs, uint32_t *blk_info, int n_slots,
                                                 uint32_t nthreads) {
    const uint32_t tid = threadIdx.x;
    for (int slot = 0; slot < kMaxHuffSlots && slot < n_slots; slot++) {
        const uint32_t pi = s.huff_pool[slot];
        if (pi == 0xFFFF) continue;
        const uint4 *src = reinterpret_cast<const uint4 *>(&huff_pool[pi]);
        uint4 *dst = reinterpret_cast<uint4 *>(tabs + slot * sizeof(DevHuffTable));
        for (uint32_t i = tid; i < sizeof(DevHuffTable) / 16; i += nthreads) dst[i] = src[i];
    }
    if (tid < kMaxBlocksPerMcu) {
        const uint32_t ci = s.blk_comp[tid];
        const uint32_t dc_off = s.comp[ci].dc_slot * (uint32_t)(sizeof(DevHuffTable) / 16);
        const uint32_t ac_off = s.comp[ci].ac_slot * (uint32_t)(sizeof(DevHuffTable) / 16);
        blk_info[tid] = dc_off | (ac_off << 12) | (ci << 24);
    }
    __syncthreads();
}

// A lane's stream positioned at bit `start_bit` of the scan's unstuffed data: K2's ring + feed + position (64 bytes staged,
// the next 16 prefetched).  Returns pm1 of the start; *endpos = position of the first bit behind the data.
__device__ __forceinline__ int32_t k2_open_at_bit(const uint8_t *ubase, uint32_t start_bit, uint32_t total_bits, uint8_t *ring, K2Feed &feed,
                                                  K2Pos &pos, int32_t *endpos) {
    const uint32_t u0 = start_bit >> 3;
    const int32_t pm1_0 = (int32_t)((u0 & 3u) * 8u + (start_bit & 7u)) - 1;
    *endpos = pm1_0 + 1 + (int32_t)(total_bits - start_bit);
    const uint8_t *g = ubase + (u0 & ~3u);  // 4-byte aligned 16-byte loads; buffers are padded
    uint4 c0, c1, c2, c3;
    __builtin_memcpy(&c0, g, 16);
    __builtin_memcpy(&c1, g + 16, 16);
    __builtin_memcpy(&c2, g + 32, 16);
    __builtin_memcpy(&c3, g + 48, 16);
    __builtin_memcpy(&feed.nx, g + 64, 16);
    k2_ring_write(ring, 0, c0);
    k2_ring_write(ring, 1, c1);
    k2_ring_write(ring, 2, c2);
    k2_ring_write(ring, 3, c3);
    feed.wr = 4;
    feed.gp = g + 80;
    k2_pos_init(pos, ring, pm1_0);
    return pm1_0;
}

// More than 64 KB of dynamic LDS has to be allowed per kernel -- and per DEVICE: a process that drives several devices (one
// jpgpu_ctx each) must do it on each of them.  done: one bit per device ordinal.
static inline hipError_t allow_dynamic_lds(const void *kernel, int bytes, std::atomic<uint64_t> &done) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const uint64_t bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return hipSuccess;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return e;
    done.fetch_or(bit, std::memory_order_release);
    return hipSuccess;
}

}  // namespace jpgpu
