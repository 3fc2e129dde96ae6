// jpeglibrary_amd/csrc/k2s_subseq.hip -- K2S: self-synchronising subsequence decode of scans without restart intervals (DRI = 0)
//
// MUST be compiled with -ffp-contract=off: the reference's Vector4 arithmetic never fuses a*b+c
// (FastFloatingPointDCT.cs:79-185).  No fast-math.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>

#include "common.h"
#include "kernels.h"
#include "encode_kernels.h"
#include "kernels_device.h"

namespace jpgpu {

// ------------------------------------------------------------------------------------------------
// K2S: self-synchronising subsequence decode for scans WITHOUT restart intervals (DRI = 0).
//
// One restart interval = one lane does not scale when the whole scan is a single interval.  The unstuffed stream is cut
// into subsequences of 1 << sub_shift bits; lane i decodes subsequence i.  Its entry state (bit position, block-in-MCU,
// zig-zag position) is unknown a priori, so round 0 guesses (start of a block of component 0) and every later round
// restarts lane i from the exit state lane i-1 reached in the previous round.  Huffman streams self-synchronise, so the
// exit states stop changing after a few rounds; lane 0 is exact from the start, hence a fixed point reached from it is
// the serial decoder's own sequence of states (Klein/Wiseman, Weissenberger/Schmidt).  After convergence the block
// counts are prefix-summed and a final pass writes the coefficients (DC as DIFFERENCE) into the zeroed coefficient
// buffer; a per-component prefix sum then turns the DC differences into the reference's predictor chain
// (ref: ScanDecoder/JpegHuffmanBaselineScanDecoder.cs:186-195).
// ------------------------------------------------------------------------------------------------


// One synchronisation round.  exit_in/exit_out are double-buffered per-subsequence state words (index sub_off + sub).
// A round only has to follow the symbol structure: code and magnitude LENGTHS, zig-zag advance, block and MCU phase, and
// the DC differences (their per-component sums feed the predictor prefix); AC magnitudes are skipped, not extracted.
//
// Round 4 form.  64 lanes stand at 64 different places of their blocks, so whatever a lane does "sometimes" the wave does
// in every step; the step is therefore ONE straight line for every lane, and what cannot be decided by one lookup is not
// branched to but PARKED:
//  * bit source = a bit position into a private 8-word LDS ring of the lane's unstuffed stream (9 words per lane: the ring
//    stored MSB-first + a mirror of word 0, so the two words around the position are always one ds_read2; stride 9 keeps
//    lanes on distinct banks); the ring is topped up (16 bytes, prefetched a burst earlier) between bursts of kSrBurst steps;
//  * one lookup of the next kSrLutBits bits gives `total bits | zig-zag advance << 6 | DC difference << 16`: a DC symbol
//    whose magnitude lies inside the looked-up prefix carries its EXTENDED value in the entry (an AC entry carries zero), so
//    the step adds the entry's upper half to the lane's component sum (one ds_add into 16 bytes of LDS per lane) whatever
//    the symbol is; DC entries advance the zig-zag position by one, EOB by 64: no DC / AC distinction but the table choice;
//  * an entry WITHOUT BITS means "not here": no bits, no advance, no difference -- the lane stands where it stood.  That is what a
//    prefix the lookup cannot decide holds (a code longer than the lookup, a DC magnitude that leaves the prefix, a bad
//    category: the reason sits in bits 13-14, which the step does not look at), and it is what a lane reads whose position is
//    beyond its limit -- the end of its subsequence, 64 bits in front
//    of what the ring holds, 32 bits in front of the end of the data: its lookup address is replaced by the address of a
//    zero word BEFORE the lookup, so the step has no test behind it.  A lane that stands repeats its step until the burst is
//    over; then the standing lanes take the exact path (sr_service: the reference's maxcode walk and "bits available" rules,
//    the ring's top-up, the end of the subsequence) under one branch per burst.
// Before: ~55 vector + ~25 scalar instructions and five branches per symbol step (the word reader's refill branches and the
// exact path, taken by some lane in nearly every step); the first round-4 form (three stream words in registers, a flagged
// entry and a limit test behind the lookup) ~42 vector instructions, 1.56 -> 1.20 ms per round; the step below ~27.
constexpr int kSrRingStride = 36;  // bytes per lane: 8 stream words + the mirror of word 0
#ifndef JPGPU_SR_BURST
#define JPGPU_SR_BURST 16
#endif
constexpr int kSrBurst = JPGPU_SR_BURST;  // fast steps between two service / top-up points
constexpr int32_t kSrParked = -0x40000000;         // limit of a lane that is finished: nothing commits any more
// entries without bits (the lane stands), by reason:
constexpr uint32_t kSrStandMiss = 0x2000u;    // a code longer than the lookup: the reference's walk, from there
constexpr uint32_t kSrStandBadCat = 0x4000u;  // a DC category above 16
constexpr uint32_t kSrStandDcWide = 0x6000u;  // a DC symbol whose magnitude leaves the prefix (code and category: the pooled first level)


struct SrLane {
    int32_t pm1;    // bit position - 1, relative to the lane's 4-byte aligned origin
    uint32_t k;     // zig-zag index of the next coefficient; 0 = the block's DC symbol comes next
    uint32_t ip;    // LDS offset of the block's entry in the block-info table ({DC lookup, AC lookup, DC sum offset, 0} per block of the MCU)
    uint32_t nblk;  // blocks completed
    uint32_t tabdc, tabac, dcaddr;  // LDS byte offsets: the block's two lookups, its component's DC sum
    int32_t slim;                   // a symbol may be looked up while pm1 < slim (kSrParked: never again)
    uint32_t wrw;                   // stream words written to the ring so far (it holds words wrw-8 .. wrw-1)
    uint4 nx;                       // the next 16 bytes of the stream, loaded a burst ago
    const uint8_t *gp;              // address of the 16 bytes behind them
};

// LDS by absolute 32-bit address (the low half of the flat address of a __shared__ object is its LDS address): every address
// the step selects between -- lookups, the zero word, DC sums, block info -- is kept ready-made, so that no base has to be
// added behind a select
typedef __attribute__((address_space(3))) uint32_t sr_lds_u32;
typedef __attribute__((address_space(3))) int32_t sr_lds_i32;
typedef uint32_t sr_u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) sr_u32x4 sr_lds_u128;
__device__ __forceinline__ uint32_t sr_lds_addr(const void *p) { return (uint32_t)reinterpret_cast<uintptr_t>(p); }
__device__ __forceinline__ uint32_t sr_ld32(uint32_t a) { return *reinterpret_cast<const sr_lds_u32 *>((uintptr_t)a); }
__device__ __forceinline__ uint4 sr_ld128(uint32_t a) {
    const sr_u32x4 v = *reinterpret_cast<const sr_lds_u128 *>((uintptr_t)a);
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void sr_st128(uint32_t a, uint32_t x, uint32_t y, uint32_t z, uint32_t w) {
    *reinterpret_cast<sr_lds_u128 *>((uintptr_t)a) = sr_u32x4{x, y, z, w};
}
__device__ __forceinline__ void sr_st32(uint32_t a, uint32_t v) { *reinterpret_cast<sr_lds_u32 *>((uintptr_t)a) = v; }

// what a symbol does to the lane: position, zig-zag index, block end (next block's lookups and DC sum), DC sum
__device__ __forceinline__ void sr_commit(SrLane &L, uint32_t info_off, uint32_t info_end, uint32_t dc_lane, uint32_t n, uint32_t adv, int32_t v) {
    __hip_atomic_fetch_add(reinterpret_cast<sr_lds_i32 *>((uintptr_t)L.dcaddr), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    L.pm1 += (int32_t)n;
    const uint32_t k = L.k + adv;
    const bool end = k >= 64u;
    L.nblk += end ? 1u : 0u;
    L.k = end ? 0u : k;
    uint32_t ip = L.ip + (end ? 16u : 0u);
    ip = ip == info_end ? info_off : ip;
    L.ip = ip;
    const uint4 inf = sr_ld128(ip);
    L.tabdc = end ? inf.x : L.tabdc;
    L.tabac = end ? inf.y : L.tabac;
    L.dcaddr = end ? dc_lane + inf.z : L.dcaddr;
}

__device__ __forceinline__ uint32_t sr_peek(const SrLane &L, uint32_t ring_off) {
    const sr_lds_u32 *p = reinterpret_cast<const sr_lds_u32 *>((uintptr_t)(ring_off + __builtin_amdgcn_ubfe((uint32_t)L.pm1, 5, 3) * 4u));
    return __builtin_amdgcn_alignbit(p[0], p[1], ~(uint32_t)L.pm1);
}

// One fast step for every lane.  Returns the entry it committed (0: the lane stands where it stood).
template <int LB>
__device__ __forceinline__ uint32_t sr_step(SrLane &L, uint32_t ring_off, uint32_t info_off, uint32_t info_end, uint32_t dc_lane, uint32_t zero_addr) {
    const uint32_t hi = sr_peek(L, ring_off);
    const uint32_t tab = L.k == 0 ? L.tabdc : L.tabac;
    uint32_t la = tab + (hi >> (32 - LB)) * 4u;
    la = L.pm1 < L.slim ? la : zero_addr;
    const uint32_t e = sr_ld32(la);
    sr_commit(L, info_off, info_end, dc_lane, e & 63u, __builtin_amdgcn_ubfe(e, 6, 7), (int32_t)e >> 16);
    return e;
}

__device__ __forceinline__ int32_t sr_limit(uint32_t wrw, int32_t endsub, int32_t endpos) {
    // pos = pm1 + 1 <= min(endsub - 1, ring - 64, data - 32): the symbol starts inside the subsequence; whatever it is (at most
    // 32 bits) it ends inside the data, and the 32 bits behind it -- the next step's peek -- are inside the ring
    int32_t s = endsub - 1;
    const int32_t r = (int32_t)(wrw * 32u) - 64, d = endpos - 32;
    s = s < r ? s : r;
    return s < d ? s : d;
}

// the prefetched 16 bytes go to the ring (slot wrw & 7: the words they replace lie in front of the lane's position), the next
// ones are requested
__device__ __forceinline__ void sr_topup(SrLane &L, uint32_t ring_off) {
    const uint32_t w0 = __builtin_bswap32(L.nx.x);
    const uint32_t at = ring_off + (L.wrw & 4u) * 4u;
    sr_st32(at, w0);
    sr_st32(at + 4, __builtin_bswap32(L.nx.y));
    sr_st32(at + 8, __builtin_bswap32(L.nx.z));
    sr_st32(at + 12, __builtin_bswap32(L.nx.w));
    if ((L.wrw & 4u) == 0) sr_st32(ring_off + 32, w0);
    L.wrw += 4;
    __builtin_memcpy(&L.nx, L.gp, 16);
    L.gp += 16;
}

// Exact path for a lane that stands: the end of its subsequence or of the data (finished), a ring that wants its top-up, or a
// symbol the lookup does not decide -- DecodeHuffmanCode + ReceiveAndExtend lengths with the reference's "bits available"
// rules, the same decisions as ub_symbol (ref: JpegHuffmanDecodingTable.cs:73-113, ScanDecoder/JpegHuffmanScanDecoder.cs:81-115).
// Commits the symbol, or finishes the lane (bad = 1: invalid code / the data ends inside the symbol).
template <int LB>
__device__ __forceinline__ void sr_service(SrLane &L, const uint8_t *smem, uint32_t lut0, uint32_t ring_off, uint32_t info_off, uint32_t info_end,
                                           uint32_t dc_lane, uint32_t small_off, const uint8_t *lut_pool, const uint32_t *pool_off, int32_t endsub,
                                           int32_t endpos, uint32_t &bad) {
    const int32_t pos = L.pm1 + 1;
    if (pos >= endsub || pos >= endpos) {  // the loop's two exits: the subsequence's end passed / no data bit left (not a failure)
        L.slim = kSrParked;
        return;
    }
    const int32_t q = L.pm1 >> 5;
    while ((int32_t)L.wrw <= q + 4) sr_topup(L, ring_off);  // the ring as full as it gets
    L.slim = sr_limit(L.wrw, endsub, endpos);
    const uint32_t hi = sr_peek(L, ring_off);
    const bool is_dc = L.k == 0;
    const uint32_t tab = is_dc ? L.tabdc : L.tabac;
    uint32_t e = sr_ld32(tab + (hi >> (32 - LB)) * 4u);
    const int32_t rem = endpos - pos;
    if ((e & 63u) != 0 && L.pm1 < L.slim) return;  // it was the ring: the next fast step takes it
    uint32_t n, adv;
    int32_t v;
    if ((e & 63u) != 0 && !(is_dc && __builtin_amdgcn_ubfe(e, 6, 7) != 1u)) {  // decided by the lookup, but inside the last 32 bits of the data
        n = e & 63u;
        adv = __builtin_amdgcn_ubfe(e, 6, 7);
        v = (int32_t)e >> 16;
    } else {
        const uint32_t sl = (tab - lut0) >> (LB + 2);
        const uint32_t code16 = hi >> 16;
        uint32_t size, cat;
        if ((e & 63u) != 0) e = kSrStandDcWide;  // a DC entry that took the first AC symbol along: the DC symbol alone, from the pooled first level
        if (e == kSrStandBadCat) {  // categories above 16 are outside the verified envelope
            bad = 1;
            L.slim = kSrParked;
            return;
        }
        adv = 1;
        if (e == kSrStandDcWide) {
            // the code has at most LB bits: its length and category are in the pooled first level (global memory: a DC difference
            // of 2^(LB - code length) and more, once in a while)
            const uint32_t e11 = reinterpret_cast<const uint16_t *>(lut_pool + pool_off[sl])[code16 >> (16 - kLutPoolBits)];
            cat = (e11 >> 6) & 31u;
            size = (e11 & 63u) - cat;
        } else {
            const uint8_t *sm = smem + small_off + sl * kK2SmallBytes;
            const uint16_t *maxcode = reinterpret_cast<const uint16_t *>(sm);
            size = LB + 1;
            while (code16 > maxcode[size]) size++;  // maxcode[17] = 0xFFFF terminates
            if (size > 16) {
                bad = 1;
                L.slim = kSrParked;
                return;
            }
            const uint32_t sym = sm[56 + ((sm[36 + size] + (code16 >> (16 - size))) & 0xFF)];
            cat = is_dc ? sym : (sym & 15u);
            if (cat > 16u) {
                bad = 1;
                L.slim = kSrParked;
                return;
            }
            if (!is_dc) adv = (sym & 15u) ? (sym >> 4) + 1u : ((sym >> 4) ? 16u : 64u);
        }
        n = size + cat;
        v = 0;
        if (is_dc && (int32_t)n <= rem) {
            const int32_t raw = (int32_t)__builtin_amdgcn_ubfe(hi, 32u - n, cat);
            v = raw - ((((raw + raw) >> cat) - 1) & ((1 << cat) - 1));  // Extend(v, nbits)
        }
    }
    if ((int32_t)n > rem) {  // the data ends inside the symbol: nothing after it can be right
        bad = 1;
        L.slim = kSrParked;
        return;
    }
    sr_commit(L, info_off, info_end, dc_lane, n, adv, v);
}

// lookups of the round kernel out of the pooled 11-bit ones (lut_pool_kernel): entry i of an LB-bit lookup covers pooled
// prefixes i << (11 - LB) ..; it is decided when the code has at most LB bits (and, for a DC symbol, the magnitude fits too)
// l1ac (DC tables): the pooled AC lookup of the table every component that decodes its DC symbols with this table decodes its AC
// symbols with, or null -- the DC entry then takes the block's FIRST AC symbol along when both fit the index (round 6: a block
// of a flat region is "DC difference 0, EOB", six bits under the standard tables: one step instead of two; no lane can be at a
// position where that is wrong -- a DC symbol never ends a block)
template <int LB>
__device__ __forceinline__ uint32_t sr_entry(const uint16_t *l1, uint32_t i, bool is_dc, const uint16_t *l1ac = nullptr) {
    const uint32_t e = l1[i << (kLutPoolBits - LB)];
    if (e == 0) return kSrStandMiss;
    if (is_dc && (e & kK2BadCat) != 0) {  // (the pooled entry does not say how long the code is: every pooled prefix under i must agree)
        for (uint32_t j = 1; j < (1u << (kLutPoolBits - LB)); j++)
            if ((l1[(i << (kLutPoolBits - LB)) + j] & kK2BadCat) == 0) return kSrStandMiss;
        return kSrStandBadCat;
    }
    const uint32_t n = e & 63u, cat = is_dc ? ((e >> 6) & 31u) : (e >> 12);
    if (n - cat > (uint32_t)LB) return kSrStandMiss;
    if (is_dc) {
        if (n > (uint32_t)LB) return kSrStandDcWide;
        const int32_t raw = (int32_t)((i >> (LB - n)) & ((1u << cat) - 1u));
        const int32_t v = raw - ((((raw + raw) >> cat) - 1) & ((1 << cat) - 1));
        if (l1ac != nullptr) {
            // ... and every AC symbol behind it that still fits: behind a DC symbol the zig-zag position is 1, ten bits of AC symbols
            // cannot take it to 64 unless the last of them is EOB, so no lane can be where the chain is wrong
            uint32_t nn = n, adv = 1u;
            while (nn < (uint32_t)LB && adv < 64u) {
                const uint32_t rem = (uint32_t)LB - nn;
                const uint32_t ib = ((i << nn) | ((1u << nn) - 1u)) & ((1u << LB) - 1u);
                const uint32_t eb = l1ac[(ib << (kLutPoolBits - LB)) | ((1u << (kLutPoolBits - LB)) - 1u)];
                if (eb == 0 || (eb & 63u) > rem) break;
                const uint32_t advb = (eb >> 6) & 63u;
                nn += eb & 63u;
                adv += advb == 63u ? 64u : advb;
            }
            return nn | ((adv < 127u ? adv : 127u) << 6) | ((uint32_t)v << 16);
        }
        return n | (1u << 6) | ((uint32_t)v << 16);
    }
    const uint32_t adv = (e >> 6) & 63u;
    // (round 6: AC + AC pairs as in K2's lookup were built and measured here too -- the blocked-pair test costs the step five
    // instructions, 27 -> 32: 4k_dri0 rounds -0.15 ms per 1024 images, the reference's benchmark canvas, three quarters flat, +0.09 ms
    // per canvas: not kept; patch and table in tools/microbench/k2_pairs/)
    return n | ((adv == 63u ? 64u : adv) << 6);
}

// One lane's round: subsequence `sub` of scan `s` decoded (structure only) from `entry`; exit state, block count and DC sums out.
template <int LB>
__device__ __forceinline__ void sr_decode_lane(const uint8_t *__restrict__ udata, const DevScan &s, const uint8_t *smem, uint32_t lut0, uint32_t small_off,
                                               uint32_t info_off, uint32_t rings_off, uint32_t dc_lane, const uint8_t *__restrict__ lut_pool,
                                               const uint32_t *pool_off, uint32_t tid, uint32_t sub, uint32_t slot, uint32_t entry, bool warm,
                                               uint32_t warm_bits, uint32_t total_bits, int round, const uint32_t *__restrict__ exit_in,
                                               uint32_t *__restrict__ exit_out, uint32_t *__restrict__ nblk_out, int4 *__restrict__ dcsum_out,
                                               uint32_t *__restrict__ changed) {
    const uint32_t bpm = s.blocks_per_mcu;
    const uint32_t end_bit = (sub + 1) << s.sub_shift;
    const uint32_t start_bit = warm ? end_bit - warm_bits : (sub << s.sub_shift) + (entry & 63u);
    uint32_t ex;
    int4 dcs = make_int4(0, 0, 0, 0);
    uint32_t nblk = 0;
    if (start_bit >= total_bits) {
        ex = sub_pack(0, (entry >> 6) & 31u, (entry >> 11) & 127u) | kSubBad;
    } else {
        SrLane L;
        const uint32_t ring_off = rings_off + tid * kSrRingStride;
        const uint32_t info_end = info_off + bpm * 16u;
        const uint32_t zero_addr = info_off + 12u;  // (the fourth word of a block-info entry)
        const uint32_t u0 = start_bit >> 3;
        const int32_t pm1_0 = (int32_t)((u0 & 3u) * 8u + (start_bit & 7u)) - 1;
        // (positions are relative to the lane's own start: the distance to the end of the data is only ever compared, so a
        // stream longer than 2^30 bits behind the lane may as well end there)
        const uint32_t left = total_bits - start_bit;
        const int32_t endpos = pm1_0 + 1 + (int32_t)(left < 0x3FFFFFFFu ? left : 0x3FFFFFFFu);
        const int32_t endsub = pm1_0 + 1 + (int32_t)(end_bit - start_bit);
        {
            const uint8_t *g = udata + s.data_off + (u0 & ~3u);  // 4-byte aligned 16-byte loads; buffers are padded
            uint4 c0, c1;
            __builtin_memcpy(&c0, g, 16);
            __builtin_memcpy(&c1, g + 16, 16);
            __builtin_memcpy(&L.nx, g + 32, 16);
            L.gp = g + 48;
            const uint32_t w0 = __builtin_bswap32(c0.x);
            sr_st32(ring_off, w0);
            sr_st32(ring_off + 4, __builtin_bswap32(c0.y));
            sr_st32(ring_off + 8, __builtin_bswap32(c0.z));
            sr_st32(ring_off + 12, __builtin_bswap32(c0.w));
            sr_st32(ring_off + 16, __builtin_bswap32(c1.x));
            sr_st32(ring_off + 20, __builtin_bswap32(c1.y));
            sr_st32(ring_off + 24, __builtin_bswap32(c1.z));
            sr_st32(ring_off + 28, __builtin_bswap32(c1.w));
            sr_st32(ring_off + 32, w0);
            L.wrw = 8;
        }
        L.pm1 = pm1_0;
        L.k = (entry >> 11) & 127u;
        L.ip = info_off + ((entry >> 6) & 31u) * 16u;
        L.nblk = 0;
        L.slim = sr_limit(L.wrw, endsub, endpos);
        {
            const uint4 inf = sr_ld128(L.ip);
            L.tabdc = inf.x;
            L.tabac = inf.y;
            L.dcaddr = dc_lane + inf.z;
        }
        uint32_t bad = 0;
        for (;;) {
            uint32_t e = 0;
#pragma unroll
            for (int t = 0; t < kSrBurst; t++) e = sr_step<LB>(L, ring_off, info_off, info_end, dc_lane, zero_addr);
            // a lane that stands repeats its step, so the last step of the burst says who stands; finished lanes are not served
            if ((e & 63u) == 0 && L.slim != kSrParked)
                sr_service<LB>(L, smem, lut0, ring_off, info_off, info_end, dc_lane, small_off, lut_pool, pool_off, endsub, endpos, bad);
            if (L.slim != kSrParked && (int32_t)L.wrw <= (L.pm1 >> 5) + 4) {
                sr_topup(L, ring_off);
                L.slim = sr_limit(L.wrw, endsub, endpos);
            }
            if (__ballot(L.slim != kSrParked) == 0) break;
        }
        const int32_t over = L.pm1 + 1 - endsub;  // bits past the nominal end
        const uint32_t b_exit = (L.ip - info_off) >> 4;
        ex = sub_pack(over > 0 ? (over < 63 ? (uint32_t)over : 63u) : 0u, b_exit, L.k);
        if (bad) ex = sub_pack(0, b_exit, L.k) | kSubBad;
        nblk = L.nblk;
        const uint4 dsum = sr_ld128(dc_lane);
        dcs = make_int4((int)dsum.x, (int)dsum.y, (int)dsum.z, (int)dsum.w);
    }
    {
        // how many exits this round changed (the convergence test; one atomic per wave that changed anything)
        const bool ch = round == 0 || ex != exit_in[slot];
        const uint64_t m = __ballot(ch);
        if (ch && (uint32_t)__builtin_ctzll(m) == (threadIdx.x & 63u)) atomicAdd(changed, (uint32_t)__builtin_popcountll(m));
    }
    exit_out[slot] = ex;
    nblk_out[slot] = nblk;
    dcsum_out[slot] = dcs;
}

// GATHER (rounds behind round 1, round 5): a late round re-decodes the few subsequences whose entry state changed -- a tenth
// of them in round 2, a few per thousand in round 3 -- and cost nearly as much as a full one: they sat one here, one there in
// the waves of every workgroup, and a wave runs as long as its slowest lane whatever the other 63 do.  A workgroup now looks at
// kSrGatherSpan subsequences (four per lane), copies the exits of the ones that stand, GATHERS the others into its lowest
// lanes and decodes those: a tenth of the waves, dense ones, and waves without a lane leave.  (Per-scan work LISTS built by
// the round before, tried in round 4, were slower: the lists' atomics and a second launch shape; here nothing leaves the
// workgroup.)  1.46 + 0.73 + 0.20 + 0.14 ms per 1024 x 4K for rounds 2-5 before.
// The round kernel's lookups of ONE set of tables (a DRI = 0 scan's: `set_scan[blockIdx.y]` is a scan that stages the set), slot
// blockIdx.x: kSrLutBits-bit u32 entries out of the pooled 11-bit ones (sr_entry), once per upload.
__global__ __launch_bounds__(256) void sr_lut_build_kernel(const DevScan *__restrict__ scans, const uint32_t *__restrict__ set_scan,
                                                           const uint8_t *__restrict__ lut_pool, uint8_t *__restrict__ sr_luts) {
    constexpr int LB = kSrLutBits;
    const DevScan &s = scans[set_scan[blockIdx.y]];
    const int sl = (int)blockIdx.x;
    uint32_t *dst = reinterpret_cast<uint32_t *>(sr_luts + (((size_t)blockIdx.y * kMaxHuffSlots + sl) << (LB + 2)));
    const uint32_t pi = s.huff_pool[sl];
    if (pi == 0xFFFF) return;
    bool is_dc = false;
    for (int c = 0; c < s.scan_components; c++) is_dc |= s.comp[c].dc_slot == sl;
    // (the pool's u16 images: AC at the table's base, DC behind it; the u32 images of K2 / the final pass follow)
    const uint16_t *src = reinterpret_cast<const uint16_t *>(lut_pool + (size_t)pi * kLutPoolBytesPerTable + (is_dc ? kK2TabBytes : 0u));
    // (a DC table's partner: the one AC table of every component that uses it)
    const uint16_t *src_ac = nullptr;
    if (is_dc) {
        uint32_t ac_pi = 0xFFFFFFFFu;
        bool one = true;
        for (int c = 0; c < s.scan_components; c++)
            if (s.comp[c].dc_slot == sl) {
                const uint32_t a = s.comp[c].ac_slot < kMaxHuffSlots ? s.huff_pool[s.comp[c].ac_slot] : 0xFFFFu;
                one &= ac_pi == 0xFFFFFFFFu || ac_pi == a;
                ac_pi = a;
            }
        if (one && ac_pi < 0xFFFFu) src_ac = reinterpret_cast<const uint16_t *>(lut_pool + (size_t)ac_pi * kLutPoolBytesPerTable);
    }
    for (uint32_t i = threadIdx.x; i < (1u << LB); i += 256) dst[i] = sr_entry<LB>(src, i, is_dc, src_ac);
}
hipError_t launch_sr_luts(hipStream_t stream, const DevScan *scans, const uint32_t *set_scan, int n_sets, const uint8_t *lut_pool, uint8_t *sr_luts) {
    if (n_sets <= 0) return hipSuccess;
    hipLaunchKernelGGL(sr_lut_build_kernel, dim3(kMaxHuffSlots, n_sets), dim3(256), 0, stream, scans, set_scan, lut_pool, sr_luts);
    return hipGetLastError();
}

constexpr uint32_t kSrGatherSpan = kSubseqGatherSpan;
template <bool GATHER>
__global__ __launch_bounds__(256) void subseq_round_kernel(const uint8_t *__restrict__ udata, const DevScan *__restrict__ scans,
                                                            const HuffWork *__restrict__ work, const uint32_t *__restrict__ ends_u,
                                                            const DevScanStatus *__restrict__ status, const DevHuffTable *__restrict__ huff_pool,
                                                            const uint8_t *__restrict__ lut_pool,
                                                            const uint32_t *__restrict__ exit_in, uint32_t *__restrict__ exit_out,
                                                            uint32_t *__restrict__ nblk_out, uint32_t *__restrict__ entry_used,
                                                            int4 *__restrict__ dcsum_out, uint32_t *__restrict__ changed, int round,
                                                            int n_slots, uint32_t warm_bits, const uint32_t *__restrict__ prev_changed,
                                                            const uint8_t *__restrict__ sr_luts) {
    constexpr int LB = kSrLutBits;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    // Device-side convergence (round 5): the rounds are enqueued ahead without the host looking at anything; once a round
    // (other than round 0) has changed no exit state, every later round is a no-op -- and leaves at once.  Both exit buffers
    // hold the same states then (a round writes every slot, changed or not), so whoever reads "the last one" may read either.
    if (prev_changed != nullptr && *prev_changed == 0) return;
    // LDS: lookups (n_slots << (LB + 2), at offset 0) | small arrays | block info [kMaxBlocksPerMcu] x 16 | rings | DC sums
    // (small_off is relative to smem; info_off, rings_off, dcs_off and what derives from them are absolute LDS addresses)
    const uint32_t lut0 = sr_lds_addr(smem);
    const uint32_t small_off = (uint32_t)n_slots << (LB + 2);
    const uint32_t info_off = lut0 + small_off + (uint32_t)n_slots * kK2SmallBytes;
    const uint32_t rings_off = info_off + kMaxBlocksPerMcu * 16u;
    const uint32_t dcs_off = rings_off + 256u * kSrRingStride;
    uint32_t *pool_off = reinterpret_cast<uint32_t *>(smem + (dcs_off - lut0) + 256u * 16u);  // [kMaxHuffSlots]: byte offset of the slot's pooled image
    const HuffWork wk = work[blockIdx.x];  // first_interval holds the first subsequence of this workgroup
    const DevScan &s = scans[wk.scan];
    const DevScanStatus st = status[wk.scan];
    if (st.n_ends == 0) return;
    const uint32_t tid = threadIdx.x;
    const uint32_t ulen = ends_u[s.ends_off];  // DRI = 0: interval 0 starts at 0
    const uint32_t total_bits = ulen * 8;
    // what a lane decodes from: the exit its predecessor reached in the round before (round 0, subsequence 0 and behind a
    // failed predecessor: the start of a block of the first component, no overshoot)
    auto entry_of = [&](uint32_t sub, uint32_t slot) -> uint32_t {
        if (sub == 0 || round == 0) return 0u;
        const uint32_t prev = exit_in[slot - 1];
        return (prev & kSubBad) ? 0u : prev;
    };
    uint32_t sub, slot, entry = 0;
    bool need;
    uint32_t n_gathered = 0;  // GATHER: subsequences this workgroup decodes (uniform)
    __shared__ uint32_t gathered[GATHER ? kSrGatherSpan : 1];
    __shared__ uint32_t wave_cnt[4];
    if (!GATHER) {
        sub = wk.first_interval + tid;
        const bool in_range = sub < s.n_subs;
        slot = s.sub_off + (in_range ? sub : 0);
        if (in_range) entry = entry_of(sub, slot);
        // a lane whose entry state did not change since it last decoded keeps its exit state (and block count);
        // a workgroup with no lane left to decode leaves before staging anything (most workgroups after round 1)
        need = in_range && !(round > 0 && (sub == 0 || entry_used[slot] == entry));
        if (in_range && !need) exit_out[slot] = exit_in[slot];
        if (!__syncthreads_or(need ? 1 : 0)) return;
    } else {
        const uint32_t lane = tid & 63u, wave = tid >> 6;
#pragma unroll 1
        for (uint32_t j = 0; j < kSrGatherSpan / 256u; j++) {
            const uint32_t sub_j = wk.first_interval + j * 256u + tid;
            const bool in_range = sub_j < s.n_subs;
            const uint32_t slot_j = s.sub_off + (in_range ? sub_j : 0);
            const bool need_j = in_range && sub_j != 0 && entry_used[slot_j] != entry_of(sub_j, slot_j);
            if (in_range && !need_j) exit_out[slot_j] = exit_in[slot_j];
            const uint64_t m = __ballot(need_j);
            if (lane == 0) wave_cnt[wave] = (uint32_t)__builtin_popcountll(m);
            __syncthreads();
            uint32_t at = n_gathered;
            for (uint32_t w = 0; w < wave; w++) at += wave_cnt[w];
            if (need_j) gathered[at + mbcnt64(m)] = sub_j;
            n_gathered += wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
            __syncthreads();
        }
        if (n_gathered == 0) return;
        sub = 0;
        slot = s.sub_off;
        need = false;
    }

    // ---- stage: lookups (round 6: built once per upload for every distinct set of tables -- sr_lut_build_kernel -- and copied here; the
    // workgroups of a round built them themselves before, 4 096 entries each, a chain of dependent loads per DC entry: a fifth to two
    // fifths of a round kernel's time), the reference's small arrays (long codes), block info
    {
        const uint4 *set = reinterpret_cast<const uint4 *>(sr_luts + ((size_t)s.sr_set * kMaxHuffSlots << (LB + 2)));
        uint4 *dst = reinterpret_cast<uint4 *>(smem);
        for (uint32_t i = tid; i < ((uint32_t)n_slots << (LB + 2)) / 16u; i += 256) dst[i] = set[i];
    }
    for (int sl = 0; sl < kMaxHuffSlots && sl < n_slots; sl++) {
        const uint32_t pi = s.huff_pool[sl];
        if (pi == 0xFFFF) continue;
        bool is_dc = false;
        for (int c = 0; c < s.scan_components; c++) is_dc |= s.comp[c].dc_slot == sl;
        if (tid == 0) pool_off[sl] = (uint32_t)((size_t)pi * kLutPoolBytesPerTable + (is_dc ? kK2TabBytes : 0u));
        const uint4 *ssrc = reinterpret_cast<const uint4 *>(reinterpret_cast<const uint8_t *>(&huff_pool[pi]) + offsetof(DevHuffTable, maxcode));
        uint4 *sdst = reinterpret_cast<uint4 *>(smem + small_off + sl * kK2SmallBytes);
        if (tid < kK2SmallBytes / 16) sdst[tid] = ssrc[tid];
    }
    if (tid < kMaxBlocksPerMcu) {
        const uint32_t ci = s.blk_comp[tid];
        sr_st128(info_off + tid * 16u, lut0 + ((uint32_t)s.comp[ci].dc_slot << (LB + 2)), lut0 + ((uint32_t)s.comp[ci].ac_slot << (LB + 2)), (ci & 3u) * 4u, 0u);
    }
    const uint32_t dc_lane = dcs_off + tid * 16u;
    sr_st128(dc_lane, 0, 0, 0, 0);
    __syncthreads();
    if (!GATHER) {
        if (!need) return;
        // ROUND 0 IS A WARM-UP.  Its entry state is a guess for every lane but the first, so all it can deliver is a plausible
        // exit state (right when the decode re-synchronises before the subsequence ends) -- and every such lane is decoded again
        // in round 1 anyway, from its predecessor's exit.  It therefore only decodes the LAST warm_bits bits of the subsequence:
        // less work in round 0, more lanes to redo in rounds 2-3 (the block phase is what converges slowly: the total number of
        // re-decodes is set by how far the nearest upstream synchronisation point is, not by round 0) -- a small net gain,
        // 19.6 -> 19.2-19.4 ms per 1024 x 4K.  entry_used is poisoned so that round 1 decodes the lane whatever its entry turns out to be.
        const bool warm = round == 0 && sub > 0 && warm_bits != 0 && warm_bits < (1u << s.sub_shift);
        entry_used[slot] = warm ? 0xFFFFFFFFu : entry;
        sr_decode_lane<LB>(udata, s, smem, lut0, small_off, info_off, rings_off, dc_lane, lut_pool, pool_off, tid, sub, slot, entry, warm, warm_bits,
                           total_bits, round, exit_in, exit_out, nblk_out, dcsum_out, changed);
    } else {
#pragma unroll 1
        for (uint32_t base = 0; base < n_gathered; base += 256u) {
            if (base + (tid & ~63u) >= n_gathered) break;  // (waves without a lane: nothing behind this needs them)
            if (base + tid < n_gathered) {
                sub = gathered[base + tid];
                slot = s.sub_off + sub;
                entry = entry_of(sub, slot);
                entry_used[slot] = entry;
                if (base != 0) sr_st128(dc_lane, 0, 0, 0, 0);
                sr_decode_lane<LB>(udata, s, smem, lut0, small_off, info_off, rings_off, dc_lane, lut_pool, pool_off, tid, sub, slot, entry, false,
                                   warm_bits, total_bits, round, exit_in, exit_out, nblk_out, dcsum_out, changed);
            }
        }
    }
}
// (Earlier forms: the word reader with the symbol step as per-lane branches, 1.9 ms per round on average at 1024 x 4K; K2's
// 68-byte ring with a top-up per block, round 2: 2.45 ms -- the rings halved the occupancy; the word reader with its refill load
// issued by hand, round 3: 1.55 ms.)
// ---- Flat regions.  Self-synchronisation lives on the randomness of the data: a constant region of the image is the same
// few bits over and over (a black 4:2:0 MCU under the standard tables is 32 bits: 00 1010 x 4, 00 00 x 2), a decoder that enters
// it with the wrong state parses it in a wrong but self-consistent way for ever, and the right state only advances one
// subsequence per round from the region's left edge.  The reference's OWN benchmark input (DecoderBenchmark.cs: three quarters
// of an 8192 x 8192 canvas are black) took 4 107 rounds, 281 ms per image.
// What such a region offers instead: subsequence i has exactly the bits of subsequence i - m (m * length = a multiple of the
// period), and a decoder is a function of (bits, entry state) -- so once subsequence i - m has been decoded from state s, the
// answer for subsequence i entered in state s is known without decoding: exit, block count and DC sums are those of i - m.
//   subseq_same_kernel       (once per upload, when a batch has not converged after a few rounds) finds for every subsequence
//                            the smallest m <= 64 with identical bits (the subsequence itself + the 128 bits behind it that a
//                            decode of it can look at), by comparison, not by hash;
//   subseq_propagate_kernel  walks a scan's subsequences once, in order, one wave per scan: where the predecessor's exit is
//                            not the state a subsequence was last decoded from, but IS the state its twin i - m was decoded
//                            from, the twin's results are copied.  Every statement it writes down is a true statement about the
//                            decoder ("entered like this, it leaves like that"), so the rounds and it can alternate freely;
//                            in a flat region the states repeat with period m, and the whole region resolves in one walk
//                            once its first m subsequences have been decoded from the right state.
// Control words of the device-driven rounds (d_sub_changed_): [0, 62) exits changed by round r, [63] copied by the walks, then
// the state of the check below; kSubCtlSameDone outlives a decode (the twins are found once per upload).
enum : uint32_t { kSubCtlArmed = 64, kSubCtlLast = 65, kSubCtlBefore = 66, kSubCtlNextCheck = 67, kSubCtlChecked = 68, kSubCtlConverged = 69,
                  kSubCtlPerDecode = 96, kSubCtlSameDone = 96, kSubCtlWords = 128 };
static_assert(kSubCtlSameDone == kSubseqCtlSameDone && kSubCtlWords == kSubseqCtlWords, "kernels.h describes this layout to the host");
// What the host's loop decides between two rounds, decided on the device (one lane): after `n` rounds, is this one of the
// points where the host would have looked at the counts (every third round; every round once a round changed only a few
// exits), has the batch converged, and does the flat-region walk (subseq_same / subseq_propagate_kernel, which return at
// once unless armed) join in -- not converged, not before round `propagate_from`, and on a PLATEAU: ordinary synchronisation
// dies out geometrically (a round changes a third to a tenth of what the one before it changed), a flat region changes as
// many exits round after round.
__global__ __launch_bounds__(64) void subseq_check_kernel(uint32_t *__restrict__ ctl, uint32_t n, uint32_t few_changes, uint32_t propagate_from) {
    if (threadIdx.x != 0) return;
    ctl[kSubCtlArmed] = 0;
    if (ctl[kSubCtlConverged] != 0) return;
    const uint32_t next = ctl[kSubCtlNextCheck];
    if (next != 0 && n < next) return;
    const uint32_t checked = ctl[kSubCtlChecked];
    uint32_t last = checked != 0 ? ctl[kSubCtlLast] : 0xFFFFFFFFu, before = checked != 0 ? ctl[kSubCtlBefore] : 0xFFFFFFFFu;
    bool converged = false;
    for (uint32_t r = checked; r < n; r++) {
        const uint32_t c = ctl[r];
        converged |= r > 0 && c == 0;
        before = last;
        last = c;
    }
    ctl[kSubCtlLast] = last;
    ctl[kSubCtlBefore] = before;
    ctl[kSubCtlChecked] = n;
    ctl[kSubCtlConverged] = converged ? 1u : 0u;
    ctl[kSubCtlNextCheck] = n + (last <= few_changes ? 1u : 3u);
    const bool plateau = before == 0xFFFFFFFFu || (uint64_t)last * 2 > before;
    ctl[kSubCtlArmed] = (!converged && n >= propagate_from && plateau) ? 1u : 0u;
}

__global__ __launch_bounds__(256) void subseq_same_kernel(const uint8_t *__restrict__ udata, const DevScan *__restrict__ scans,
                                                          const HuffWork *__restrict__ work, const uint32_t *__restrict__ ends_u,
                                                          const DevScanStatus *__restrict__ status, uint32_t *__restrict__ same_dist,
                                                          const uint32_t *__restrict__ ctl) {
    if (ctl != nullptr && (ctl[kSubCtlArmed] == 0 || ctl[kSubCtlSameDone] != 0)) return;  // (device-driven rounds: not now / done for this upload)
    const HuffWork wk = work[blockIdx.x];  // (the rounds' work list: 256 subsequences per entry)
    const DevScan &s = scans[wk.scan];
    if (status[wk.scan].n_ends == 0) return;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t total_bits = ends_u[s.ends_off] * 8u;
    const uint32_t sub_bytes = (1u << s.sub_shift) >> 3, n_words = sub_bytes / 4u + 4u;
    const uint8_t *base = udata + s.data_off;
    for (uint32_t q = wave; q < 256u; q += 4u) {
        const uint32_t sub = wk.first_interval + q;
        if (sub >= s.n_subs) break;
        uint32_t result = 0;
        // (a subsequence near the end of the data also depends on where the data ends: no twin for it)
        if ((uint64_t)(sub + 1u) * sub_bytes * 8u + 128u <= total_bits) {
            const uint8_t *mine = base + (size_t)sub * sub_bytes;
            uint32_t w0, wc = 0;
            __builtin_memcpy(&w0, mine, 4);
            const uint32_t m_lane = lane + 1u;
            if (m_lane <= sub) __builtin_memcpy(&wc, mine - (size_t)m_lane * sub_bytes, 4);
            uint64_t cand = __ballot(m_lane <= sub && wc == w0);
            while (cand != 0) {
                const uint32_t m = (uint32_t)__builtin_ctzll(cand) + 1u;
                const uint8_t *twin = mine - (size_t)m * sub_bytes;
                bool diff = false;
                for (uint32_t t = lane; t < n_words; t += 64u) {
                    uint32_t a, b;
                    __builtin_memcpy(&a, mine + t * 4u, 4);
                    __builtin_memcpy(&b, twin + t * 4u, 4);
                    diff |= a != b;
                }
                if (__ballot(diff) == 0) {
                    result = m;
                    break;
                }
                cand &= cand - 1;
            }
        }
        if (lane == 0) same_dist[s.sub_off + sub] = result;
    }
}

// one wave per scan; exit = the buffer the last round wrote
__global__ __launch_bounds__(64) void subseq_propagate_kernel(const DevScan *__restrict__ scans, const uint32_t *__restrict__ scan_ids,
                                                              const uint32_t *__restrict__ same_dist, uint32_t *__restrict__ exit_state,
                                                              uint32_t *__restrict__ entry_used, uint32_t *__restrict__ nblk,
                                                              int4 *__restrict__ dcsum, uint32_t *__restrict__ n_copied, uint32_t *__restrict__ ctl) {
    if (ctl != nullptr) {  // device-driven rounds: the walk runs where subseq_check_kernel armed it
        if (ctl[kSubCtlArmed] == 0) return;
        if (blockIdx.x == 0 && threadIdx.x == 0) ctl[kSubCtlSameDone] = 1;  // (the twins were found in front of this launch)
    }
    const DevScan &s = scans[scan_ids[blockIdx.x]];
    const uint32_t lane = threadIdx.x;
    uint32_t e_prev = 0xFFFFFFFFu, x_prev = 0;  // the chunk before (lane l = subsequence base - 64 + l)
    uint32_t copied = 0;
    // the next chunk's three words are asked for while this one is looked at (a chunk only ever stores its OWN entries, and most
    // chunks are consistent already: the walk is a chain of load latencies otherwise -- 483 chunks for one benchmark canvas)
    uint32_t e_next = 0xFFFFFFFFu, x_next = 0, d_next = 0;
    auto fetch = [&](uint32_t base) {
        const uint32_t sub = base + lane;
        const bool in = sub < s.n_subs;
        const uint32_t slot = s.sub_off + (in ? sub : 0u);
        e_next = in ? entry_used[slot] : 0xFFFFFFFFu;
        x_next = in ? exit_state[slot] : 0u;
        d_next = in ? same_dist[slot] : 0u;
    };
    fetch(0);
    for (uint32_t base = 0; base < s.n_subs; base += 64u) {
        const uint32_t sub = base + lane;
        const bool in = sub < s.n_subs;
        const uint32_t slot = s.sub_off + (in ? sub : 0u);
        uint32_t e = e_next, x = x_next;
        const uint32_t d = d_next;
        if (base + 64u < s.n_subs) fetch(base + 64u);
        uint32_t src = 0;  // lane l: the twin whose results it takes (distance), 0 = none
        // what the rounds give a subsequence as its entry: the predecessor's exit, or the start state behind a failed one
        auto expected = [](uint32_t prev_exit) { return (prev_exit & kSubBad) ? 0u : prev_exit; };
        // whole chunk consistent already?  (lane 0 against the previous chunk's last)
        const uint32_t xl = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(((lane + 63u) & 63u) * 4u), (int)x);
        const uint32_t before = lane == 0 ? lane_get(x_prev, 63) : xl;
        const bool first = sub == 0;
        if (__ballot(in && !first && e != expected(before)) != 0) {
            const uint32_t n_here = s.n_subs - base < 64u ? s.n_subs - base : 64u;
            // A chunk INSIDE a periodic run (every one of its 64 subsequences has its twin at the same distance m: the middle of a
            // flat region): what the lane-by-lane walk below would arrive at is the last period of the chunk before, repeated --
            // subsequence l takes the states of the one m * (l / m + 1) places back.  Proposed for all 64 lanes at once and
            // CHECKED the way the walk checks (the entry the twin was decoded from is the exit of the proposed predecessor); one
            // step instead of 64 dependent ones.  The reference's benchmark canvas (one 67-Mpixel scan, 550 chunks, most of them
            // black): the walk took 0.90 of the 2.7 ms a single canvas takes.
            bool periodic = false;
            const uint32_t m0 = lane_get(d, 0);
            if (base != 0 && n_here == 64u && m0 != 0 && __ballot(d != m0) == 0) {
                const uint32_t back = m0 * (lane / m0 + 1u);
                const uint32_t pl = 64u + lane - back;  // the source: a lane of the last period of the chunk before
                const uint32_t pe = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(pl * 4u), (int)e_prev);
                const uint32_t px = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(pl * 4u), (int)x_prev);
                const uint32_t pxb = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(((lane + 63u) & 63u) * 4u), (int)px);
                const uint32_t before_p = lane == 0 ? lane_get(x_prev, 63) : pxb;
                if (__ballot(pe != expected(before_p)) == 0) {
                    periodic = true;
                    const bool took = e != pe || x != px;
                    if (took) {
                        entry_used[slot] = pe;
                        exit_state[slot] = px;
                        nblk[slot] = nblk[slot - back];
                        dcsum[slot] = dcsum[slot - back];
                    }
                    e = pe;
                    x = px;
                    if (__ballot(took) != 0) {
                        copied += (uint32_t)__builtin_popcountll(__ballot(took));
                        __threadfence();  // the next chunks read what this one stored
                    }
                }
            }
            // Otherwise lane by lane, in order -- but only the lanes that can take anything: the ones not entered with their
            // predecessor's exit now, and the successor of every lane whose exit this walk changes (a chunk of the benchmark
            // canvas's upper half holds one flat stretch of eight subsequences: eight visits instead of 64).
            uint64_t todo = periodic ? 0ull : __ballot(in && !first && e != expected(before));
            while (todo != 0) {
                const uint32_t l = (uint32_t)__builtin_ctzll(todo);
                todo &= todo - 1;
                const uint32_t want = expected(l == 0 ? lane_get(x_prev, 63) : lane_get(x, l - 1u));
                if (lane_get(e, l) == want) continue;
                const uint32_t m = lane_get(d, l);
                if (m == 0) continue;
                uint32_t te, tx;
                if (m <= l) {
                    te = lane_get(e, l - m);
                    tx = lane_get(x, l - m);
                } else {
                    if (base == 0) continue;
                    te = lane_get(e_prev, 64u + l - m);
                    tx = lane_get(x_prev, 64u + l - m);
                }
                if (te != want) continue;
                // a twin that took ITS results from a twin in this very chunk has not stored them yet: go to where they lie
                const uint32_t via = m <= l ? lane_get(src, l - m) : 0u;
                if (lane == l) {  // (uniform values, one lane's registers)
                    e = want;
                    x = tx;
                    src = m + via;
                }
                // the successor is entered with this lane's new exit from now on
                if (l + 1u < n_here) {
                    if (lane_get(e, l + 1u) != expected(tx)) todo |= 1ull << (l + 1u);
                    else todo &= ~(1ull << (l + 1u));
                }
            }
            const bool took = src != 0;
            if (took) {
                entry_used[slot] = e;
                exit_state[slot] = x;
                nblk[slot] = nblk[slot - src];
                dcsum[slot] = dcsum[slot - src];
            }
            if (__ballot(took) != 0) {
                copied += (uint32_t)__builtin_popcountll(__ballot(took));
                __threadfence();  // the next chunks read what this one stored
            }
        }
        e_prev = e;
        x_prev = x;
    }
    if (lane == 0 && copied != 0) atomicAdd(n_copied, copied);
}

// Exclusive prefix sums over a scan's subsequences: first block and DC predictors at the entry of every subsequence.
// One workgroup per scan.
__global__ __launch_bounds__(1024) void subseq_scan_kernel(const DevScan *__restrict__ scans, const uint32_t *__restrict__ scan_ids,
                                                            const uint32_t *__restrict__ nblk, uint32_t *__restrict__ first_block,
                                                            const int4 *__restrict__ dcsum, int4 *__restrict__ dc_entry) {
    const DevScan &s = scans[scan_ids[blockIdx.x]];
    __shared__ int32_t sh[5][1024];
    const uint32_t tid = threadIdx.x;
    const uint32_t per = (s.n_subs + 1023) / 1024;
    const uint32_t lo = tid * per, hi = (lo + per) < s.n_subs ? (lo + per) : s.n_subs;
    int32_t sum[5] = {0, 0, 0, 0, 0};
    // (independent loads, a stride of `per` entries between neighbouring lanes: unrolled so that several are in flight -- one
    // workgroup per scan, and a single 67-Mpixel scan has 35 entries per lane: 0.124 ms of a lone canvas's 2.3 before)
#pragma unroll 8
    for (uint32_t i = lo; i < hi; i++) {
        const int4 d = dcsum[s.sub_off + i];
        sum[0] += (int32_t)nblk[s.sub_off + i];
        sum[1] += d.x;
        sum[2] += d.y;
        sum[3] += d.z;
        sum[4] += d.w;
    }
#pragma unroll
    for (int c = 0; c < 5; c++) sh[c][tid] = sum[c];
    __syncthreads();
    for (uint32_t o = 1; o < 1024; o <<= 1) {  // Hillis-Steele inclusive scan
        int32_t v[5];
#pragma unroll
        for (int c = 0; c < 5; c++) v[c] = tid >= o ? sh[c][tid - o] : 0;
        __syncthreads();
#pragma unroll
        for (int c = 0; c < 5; c++) sh[c][tid] += v[c];
        __syncthreads();
    }
    int32_t run[5];
#pragma unroll
    for (int c = 0; c < 5; c++) run[c] = sh[c][tid] - sum[c];
#pragma unroll 8
    for (uint32_t i = lo; i < hi; i++) {
        first_block[s.sub_off + i] = (uint32_t)run[0];
        dc_entry[s.sub_off + i] = make_int4(run[1], run[2], run[3], run[4]);
        const int4 d = dcsum[s.sub_off + i];
        run[0] += (int32_t)nblk[s.sub_off + i];
        run[1] += d.x;
        run[2] += d.y;
        run[3] += d.z;
        run[4] += d.w;
    }
}

// Final pass.  The converged entry states say where every subsequence's first block begins: lane i decodes the whole MCUs
// that START inside subsequence i (it first parses, without storing, the blocks between its entry and that MCU -- the rest
// of an MCU the previous lane owns -- and runs past its own end to finish its last MCU), so every MCU has exactly one owner
// and EVERY LANE OF A WAVE STANDS AT THE SAME BLOCK OF ITS MCU: from there on this is K2 -- block b of every lane decoded
// in lock-step with wave-uniform tables into the wave's LDS staging and flushed as whole 128-byte lines; the coefficient
// buffer needs no clearing.  The DC predictor chain starts from the prefix sums of subseq_scan_kernel plus the DC
// differences of the blocks parsed on the way to the first MCU.
// (First version: the generic word reader and symbol decoder, 11.0 ms per 1024 x 4K against K2's 5.7.  Second, rounds 2-3:
// K2's ring and symbol step with BLOCK-aligned ownership -- lanes at different blocks of their MCUs, so the tables were
// picked per lane and a lock-step iteration mixed long luma with short chroma blocks: 8.3 ms.)
// What a lane of the final pass owns: the whole MCUs that START in its kSubFinalSubsPerLane subsequences (from `sub`).
struct SfRange {
    bool live;
    uint32_t entry, my_first, my_end, skip;
};
__device__ __forceinline__ SfRange sf_lane_range(const DevScan &s, const uint32_t *__restrict__ exit_state, const uint32_t *__restrict__ first_block,
                                                 uint32_t sub, uint32_t spl) {
    const uint32_t slot = s.sub_off + (sub < s.n_subs ? sub : 0);
    const uint32_t total_mcus = s.total_mcus, bpm = s.blocks_per_mcu;
    SfRange r;
    r.live = sub < s.n_subs;
    r.entry = 0;
    if (r.live && sub > 0) {
        const uint32_t prev = exit_state[slot - 1];
        if (prev & kSubBad) r.live = false;  // the stream ended or failed in an earlier subsequence: reported by that lane
        else r.entry = prev;
    }
    const uint32_t i2 = ((r.entry >> 11) & 127u) * 2u;  // 2 x zig-zag position inside the block in progress at the entry
    // first_block = blocks completed before the entry = index of the block in progress (i2 != 0: the previous lane's) or
    // about to start there.  This lane owns the MCUs from the first one that starts at or behind its entry ...
    r.my_first = total_mcus;
    r.my_end = total_mcus;
    r.skip = 0;
    if (r.live) {
        const uint32_t at = first_block[slot];
        r.my_first = (at + (i2 != 0 ? 1u : 0u) + bpm - 1) / bpm;
        r.skip = r.my_first * bpm - at;  // block ends between the entry and that MCU (the first of them may be half a block away)
        // ... up to the first one that starts at or behind the next lane's entry (a stream that failed or ran out inside this
        // lane's subsequences leaves it everything that remains)
        const uint32_t n_mine = s.n_subs - sub < spl ? s.n_subs - sub : spl;
        bool open_end = sub + n_mine >= s.n_subs;
        uint32_t ex = 0;
        for (uint32_t q = 0; q < n_mine; q++) {
            ex = exit_state[slot + q];
            open_end |= (ex & kSubBad) != 0;
        }
        if (!open_end) r.my_end = (first_block[slot + n_mine] + ((((ex >> 11) & 127u) != 0) ? 1u : 0u) + bpm - 1) / bpm;
        if (r.my_end > total_mcus) r.my_end = total_mcus;  // the reference stops after the last MCU
        if (r.my_first > r.my_end) r.my_first = r.my_end;
    }
    return r;
}

// The lanes of a scan ORDERED by the number of MCUs they own (round 5).  A wave of the final pass iterates to the largest count
// among its 64 lanes -- 35 where the mean is 31 on the benchmark's 4K frames: a tenth of the pass spent on lanes that have
// finished.  Per group of kSubseqGatherSpan subsequences (512 lanes, one workgroup; the rounds' gather list) the lanes are
// counting-sorted by their count: perm[sub_off + k] = the lane that takes place k.  Ownership does not change, only which 64
// lanes share a wave; a lane's stream and its coefficient range were never its neighbours' anyway.
__global__ __launch_bounds__(1024) void subseq_order_kernel(const DevScan *__restrict__ scans, const HuffWork *__restrict__ work,
                                                            const DevScanStatus *__restrict__ status, const uint32_t *__restrict__ exit_state,
                                                            const uint32_t *__restrict__ first_block, uint32_t *__restrict__ perm, uint32_t spl) {
    constexpr uint32_t kKeys = 256;  // (launched with kSubseqGatherSpan / spl threads: one per lane of the group)
    __shared__ uint32_t hist[kKeys];
    const HuffWork wk = work[blockIdx.x];
    const DevScan &s = scans[wk.scan];
    if (status[wk.scan].n_ends == 0) return;
    const uint32_t tid = threadIdx.x;
    if (tid < kKeys) hist[tid] = 0;
    __syncthreads();
    const uint32_t lane_index = wk.first_interval / spl + tid;
    const uint32_t n_lanes = (s.n_subs + spl - 1) / spl;
    const bool in_range = lane_index < n_lanes;
    uint32_t key = 0;
    if (in_range) {
        const SfRange r = sf_lane_range(s, exit_state, first_block, lane_index * spl, spl);
        const uint32_t count = r.my_end - r.my_first;
        key = count < kKeys - 1 ? count : kKeys - 1;
    }
    uint32_t rank_in_key = 0;
    if (in_range) rank_in_key = atomicAdd(&hist[key], 1u);
    __syncthreads();
    // exclusive prefix over the keys (256 entries: one wave's worth of work, done by every lane's own walk over LDS would be 256
    // reads; four threads per ... keep it simple: thread k sums the keys below k)
    __shared__ uint32_t base[kKeys];
    if (tid < kKeys) {
        uint32_t acc = 0;
        for (uint32_t k = 0; k < tid; k++) acc += hist[k];
        base[tid] = acc;
    }
    __syncthreads();
    if (in_range) perm[s.sub_off + wk.first_interval / spl + base[key] + rank_in_key] = lane_index;
}

#ifdef JPGPU_K2_PROFILE  // per-wave cycle counters of the final pass (tools/trace/k2_phases.py dri0): [0] waves, [1] open + skip, [2] decode, [3] top-up, [4] flush, [5] total, [6] block steps, [7] lane-blocks
__device__ unsigned long long sf_prof[8];
#define SF_TICK() __builtin_readcyclecounter()
#define SF_PROF_ADD(i, v) do { if (lane == 0) atomicAdd(&sf_prof[i], (unsigned long long)(v)); } while (0)
extern "C" int jpgpu_debug_sf_profile(unsigned long long *out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(sf_prof), sizeof(sf_prof)) != hipSuccess) return 1;
    if (reset) { unsigned long long z[8] = {}; if (hipMemcpyToSymbol(HIP_SYMBOL(sf_prof), z, sizeof z) != hipSuccess) return 1; }
    return 0;
}
#else
#define SF_TICK() 0ull
#define SF_PROF_ADD(i, v) do { (void)(v); } while (0)
#endif
constexpr int kSubFinalMaxWaves = JPGPU_SF_WAVES > kSubFinalPoolWaves ? JPGPU_SF_WAVES : kSubFinalPoolWaves;  // the launch picks subseq_final_waves(n_slots) / the pool's
constexpr int kSfWaveBytes = kK2WaveBytes;
static_assert(kSfWaveBytes == (int)kSfWaveLdsBytes && kSubFinalPoolWaves == 11, "kernels.h: subseq_pool_fits");  // K2's staging + rings
// One wave's 64 lanes: subsequences wk.first_interval + lane * kSubFinalSubsPerLane .. of scan wk.scan (tables staged, the wave's
// coefficient staging zero on entry and on exit).
__device__ __forceinline__ void sf_wave(const uint8_t *__restrict__ udata, const DevScan &s, const DevScanStatus &st, const HuffWork wk,
                                        const uint32_t *__restrict__ ends_u, DevScanStatus *__restrict__ status,
                                        const uint32_t *__restrict__ exit_state, const uint32_t *__restrict__ first_block,
                                        const int4 *__restrict__ dc_entry, int16_t *__restrict__ coefs, const uint8_t *tabs,
                                        const uint32_t *blk_info, uint8_t *stage, uint8_t *ring, uint32_t lane,
                                        const uint32_t *__restrict__ perm, uint32_t spl) {
    const unsigned long long sf_t0 = SF_TICK();
    unsigned long long sf_dec = 0, sf_top = 0, sf_fl = 0;
    const uint32_t ulen = ends_u[s.ends_off];
    const uint32_t total_bits = ulen * 8;
    // a lane takes kSubFinalSubsPerLane consecutive subsequences: half the lanes, but half the parsed-not-stored blocks and a
    // narrower spread of MCU counts inside a wave (the wave iterates to its largest)
    // (perm: the lanes of a scan in the order subseq_order_kernel put them in -- waves of lanes that own about as many MCUs)
    const uint32_t lane_index = wk.first_interval / spl + lane;
    const uint32_t n_lanes = (s.n_subs + spl - 1) / spl;
    const uint32_t sub = (perm != nullptr && lane_index < n_lanes ? perm[s.sub_off + lane_index] : lane_index) * spl;
    const uint32_t slot = s.sub_off + (sub < s.n_subs ? sub : 0);
    const uint32_t total_mcus = s.total_mcus;
    const uint32_t bpm = s.blocks_per_mcu;
    const uint64_t coef_off = s.coef_off;
    const bool closed_by_marker = st.terminator != 0;
    const SfRange rg = sf_lane_range(s, exit_state, first_block, sub, spl);
    const bool live = rg.live;
    const uint32_t entry = rg.entry, my_first = rg.my_first, my_end = rg.my_end, skip = rg.skip;
    uint32_t b_in_mcu = (entry >> 6) & 31u;
    uint32_t i2 = ((entry >> 11) & 127u) * 2u;  // 2 x zig-zag position inside the block in progress at the entry
    const uint32_t count = my_end - my_first;
    const uint32_t wave_count = wave_reduce_max_i((int32_t)count);

    const bool decodes = live && count != 0;
    const uint32_t start_bit = decodes ? (sub << s.sub_shift) + (entry & 63u) : 0u;
    K2Feed feed;
    K2Pos pos;
    int32_t endpos = 0;
    // A lane that owns MCUs but starts behind the data (the stream ran out at a symbol boundary in an earlier subsequence)
    // decodes them the way the reference does: from the all-ones padding, i.e. with no data bits at all (its loads still
    // have to stay inside the buffer: it opens the stream at bit 0 and sees it as empty).
    const bool behind_data = start_bit >= total_bits;
    const int32_t pm1_0 = k2_open_at_bit(udata + s.data_off, behind_data ? 0u : start_bit, total_bits, ring, feed, pos, &endpos);
    if (behind_data) endpos = pm1_0 + 1;
    int32_t lim = k2_limit(endpos, feed.wr);
    uint32_t err = 0;
    int32_t pred0 = 0, pred1 = 0, pred2 = 0, pred3 = 0;
    if (decodes) {
        const int4 de = dc_entry[slot];
        pred0 = de.x;
        pred1 = de.y;
        pred2 = de.z;
        pred3 = de.w;
    }
    // the blocks in front of the lane's first MCU (the previous lane's): parsed, not stored; their DC differences count.
    // The only stretch where the lanes of a wave stand at different blocks of their MCUs (tables picked per lane).
    {
        uint32_t left = decodes ? skip : 0u;
        uint32_t info = blk_info[b_in_mcu];
        uint32_t it = 0;
        while (__ballot(left != 0) != 0) {
            if (left != 0) {
                const bool is_dc = i2 == 0;
                int32_t v;
                uint32_t adv;
                err = k2_symbol_any(ring, feed, pos, endpos, lim, k2_tab_at(tabs, is_dc ? ((info >> 8) & 0xFFFu) : (info >> 20), is_dc), is_dc,
                                    closed_by_marker, v, adv);
                if (is_dc) {
                    const uint32_t ci = info & 0xFFu;
                    if (ci == 0) pred0 += v;
                    else if (ci == 1) pred1 += v;
                    else if (ci == 2) pred2 += v;
                    else pred3 += v;
                    adv = 2;
                }
                i2 += adv;
                if (err != 0 || i2 >= 128u) {
                    left = err != 0 ? 0u : left - 1;
                    i2 = 0;
                    b_in_mcu = (b_in_mcu + 1 == bpm) ? 0u : b_in_mcu + 1;
                    info = blk_info[b_in_mcu];
                }
            }
            if ((++it & 3u) == 0) {
                k2_topup(ring, feed, pos.pm1);
                lim = k2_limit(endpos, feed.wr);
            }
        }
    }
    k2_topup(ring, feed, pos.pm1);
    uint8_t *my_stage = stage + lane * 128;
    const uint32_t swz16 = ((lane >> 1) & 7u) << 4;
    const unsigned long long sf_t1 = SF_TICK();

    for (uint32_t j = 0; j < wave_count; j++) {
        for (uint32_t b = 0; b < bpm; b++) {
            const unsigned long long sf_a = SF_TICK();
            const uint32_t bi = __builtin_amdgcn_readfirstlane(blk_info[b]);  // wave-uniform
            const uint32_t ci = bi & 0xFFu;
            const K2Tab hdc = k2_tab_dc(tabs, bi);
            const K2Tab hac = k2_tab_ac(tabs, bi);
            lim = k2_limit(endpos, feed.wr);
            if (j < count && err == 0) {
                // ReadBlockBaseline (ref: ScanDecoder/JpegHuffmanBaselineScanDecoder.cs:179-222)
                int32_t v, vb;
                uint32_t ia = 0, adv_b = 0;
                k2_symbol<true>(ring, feed, pos, endpos, lim, hdc, closed_by_marker, 0u, v, vb, ia, adv_b, err);
                const int32_t pred = ci == 0 ? pred0 : (ci == 1 ? pred1 : (ci == 2 ? pred2 : pred3));
                v += pred;
                if (ci == 0) pred0 = v;
                else if (ci == 1) pred1 = v;
                else if (ci == 2) pred2 = v;
                else pred3 = v;
                *reinterpret_cast<int16_t *>(my_stage + swz16) = (int16_t)v;  // zig-zag index 0
                uint32_t k2i = err == 0 ? 2u : 128u;
                while (k2i < 128u) {
                    // one step = one symbol or the two of a pair entry (see K2): a step of one symbol stores it twice
                    k2_symbol<false>(ring, feed, pos, endpos, lim, hac, closed_by_marker, k2i, v, vb, ia, adv_b, err);
                    const uint32_t at = ia - 2u < 126u ? ia - 2u : 126u;
                    *reinterpret_cast<int16_t *>(my_stage + (at ^ swz16)) = (int16_t)v;
                    k2i = ia + 2u * adv_b;
                    const uint32_t at_b = k2i - 2u < 126u ? k2i - 2u : 126u;
                    *reinterpret_cast<int16_t *>(my_stage + (at_b ^ swz16)) = (int16_t)vb;
                }
                // the block the reference throws in (see K2): in front of it every block has reached the writer
                if (err != 0)
                    atomicMax(&status[wk.scan].pad[1], fail_block_word(((uint64_t)my_first + j) * bpm + b));
            }
            const unsigned long long sf_b = SF_TICK();
            k2_topup(ring, feed, pos.pm1);
            const unsigned long long sf_c = SF_TICK();
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int it = 0; it < 8; it++) {
                const uint32_t blk = it * 8 + (lane >> 3);
                const uint32_t chunk = lane & 7;
                uint4 *src = reinterpret_cast<uint4 *>(stage + blk * 128 + ((chunk ^ ((blk >> 1) & 7)) * 16));
                const uint4 v = *src;
                const uint4 z = {0, 0, 0, 0};
                *src = z;
                // (round 6: the owner's first MCU and count come through the lane crossbar -- ds_bpermute, no storage -- where an LDS table of
                // 512 bytes per wave held them: the eleventh wave of the pooled workgroup fits)
                const uint32_t owner_first = (uint32_t)__shfl((int)my_first, (int)blk, 64), owner_count = (uint32_t)__shfl((int)count, (int)blk, 64);
                if (j < owner_count) *reinterpret_cast<uint4 *>(coefs + (coef_off + ((uint64_t)owner_first + j) * bpm + b) * 64 + chunk * 8) = v;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const unsigned long long sf_d = SF_TICK();
            sf_dec += sf_b - sf_a;
            sf_top += sf_c - sf_b;
            sf_fl += sf_d - sf_c;
        }
    }
    SF_PROF_ADD(0, 1);
    SF_PROF_ADD(1, sf_t1 - sf_t0);
    SF_PROF_ADD(2, sf_dec);
    SF_PROF_ADD(3, sf_top);
    SF_PROF_ADD(4, sf_fl);
    SF_PROF_ADD(5, SF_TICK() - sf_t0);
    SF_PROF_ADD(6, (unsigned long long)wave_count * bpm);
    const uint32_t sf_lane_blocks = wave_sum(count) * bpm;
    SF_PROF_ADD(7, sf_lane_blocks);
    if (live && err != 0) {
        // failure on the true path: same detail codes as the interval decoder; "interval" field carries the subsequence
        atomicMin(&status[wk.scan].first_error, (sub << 8) | err);
    }
    // bits left behind the scan's last block (see restart_check)
    if (live && err == 0 && count != 0 && my_end == total_mcus) {
        const int32_t rem = endpos - (pos.pm1 + 1);
        status[wk.scan].pad[2] = rem > 0 ? (uint32_t)rem : 0u;
    }
}

// POOL (round 5).  The pass is K2 with fewer waves per CU: a workgroup of the plain form is four waves of one scan (two
// workgroups, eight waves per CU; K2 has eleven), because a workgroup waits for its slowest wave and the waves of this pass
// own different numbers of MCUs -- one workgroup of ten waves was measured slower (21.1 vs 19.6 ms K2S per 1024 x 4K).  Where many
// consecutive scans stage the same tables (every batch of files from one encoder with the standard tables), the waves need not
// be tied to a scan: one workgroup of kSubFinalPoolWaves waves per CU stages the tables once and every WAVE takes the next 64
// lanes of the run from a counter until there are none -- no wave waits for another, and nothing is restaged.
template <bool POOL>
__global__ __launch_bounds__(64 * kSubFinalMaxWaves) void subseq_final_kernel(const uint8_t *__restrict__ udata, const DevScan *__restrict__ scans,
                                                                           const HuffWork *__restrict__ work, const uint32_t *__restrict__ ends_u,
                                                                           DevScanStatus *__restrict__ status,
                                                                           const DevHuffTable *__restrict__ huff_pool,
                                                                           const uint8_t *__restrict__ lut_pool,
                                                                           const uint32_t *__restrict__ exit_state,
                                                                           const uint32_t *__restrict__ first_block,
                                                                           const int4 *__restrict__ dc_entry, int16_t *__restrict__ coefs,
                                                                           int n_slots, uint32_t n_chunks, uint32_t *__restrict__ counter,
                                                                           const uint32_t *__restrict__ perm, uint32_t spl, uint32_t tab_bytes) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t n_waves = blockDim.x >> 6;
    uint8_t *tabs = smem;                                                    // tab_bytes: the batch's largest set of staged tables
    uint8_t *wave_all = smem + tab_bytes;                                    // n_waves * kSfWaveBytes
    uint32_t *blk_info = reinterpret_cast<uint32_t *>(wave_all + n_waves * kSfWaveBytes);  // [kMaxBlocksPerMcu]
    const HuffWork wk0 = work[POOL ? 0u : blockIdx.x];  // POOL: `work` lists the run's waves (scan, first subsequence), all with one set of tables
    if (!POOL && status[wk0.scan].n_ends == 0) return;
    k2_stage_scan_tables(scans[wk0.scan], lut_pool, tabs, blk_info, n_slots, blockDim.x);
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint8_t *stage = wave_all + wave * kSfWaveBytes;
    uint8_t *ring = stage + 8192 + lane * kK2RingStride;
    {
        const uint4 z = {0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < 8; i++) reinterpret_cast<uint4 *>(stage)[i * 64 + lane] = z;
    }
    __syncthreads();
    if (!POOL) {
        const HuffWork wk = {wk0.scan, wk0.first_interval + wave * 64u * spl};
        sf_wave(udata, scans[wk.scan], status[wk.scan], wk, ends_u, status, exit_state, first_block, dc_entry, coefs, tabs, blk_info, stage, ring, lane, perm, spl);
    } else {
#if !defined(JPGPU_K2_TICKET_AHEAD)
        for (;;) {
            uint32_t c = 0;
            if (lane == 0) c = atomicAdd(counter, 1u);
            c = (uint32_t)__builtin_amdgcn_readfirstlane((int)c);
            if (c >= n_chunks) break;
            const HuffWork wk = work[c];
            const DevScanStatus st = status[wk.scan];
            if (st.n_ends == 0) continue;
            sf_wave(udata, scans[wk.scan], st, wk, ends_u, status, exit_state, first_block, dc_entry, coefs, tabs, blk_info, stage, ring, lane, perm, spl);
        }
#else
        // (-DJPGPU_K2_TICKET_AHEAD: measured and not adopted, see huffman_pool_kernel in k2_huffman.hip)
        uint32_t c = 0;
        if (lane == 0) c = atomicAdd(counter, 1u);
        c = (uint32_t)__builtin_amdgcn_readfirstlane((int)c);
        while (c < n_chunks) {
            uint32_t ahead = 0;
            if (lane == 0) ahead = atomicAdd(counter, 1u);
            const HuffWork wk = work[c];
            const DevScanStatus st = status[wk.scan];
            if (st.n_ends != 0)
                sf_wave(udata, scans[wk.scan], st, wk, ends_u, status, exit_state, first_block, dc_entry, coefs, tabs, blk_info, stage, ring, lane, perm, spl);
            c = (uint32_t)__builtin_amdgcn_readfirstlane((int)ahead);
        }
#endif
    }
}


// DRI = 0 scans: self-synchronising subsequence decode.  `work` lists (scan, first subsequence) per workgroup of 256 lanes;
// `scan_ids` the scans concerned.  The synchronisation part alone (rounds until the exit states stop changing, then the block /
// DC prefix sums); leaves the converged exit states in *final_state.  Shared by the decoder (subseq_final_kernel) and the
// optimizer (subseq_transcode_kernel).
//
// Two ways of knowing when to stop:
//  * device_rounds > 0 (the decoder, round 5): exactly that many rounds are enqueued and the host looks at NOTHING -- no copy,
//    no stream synchronisation.  A round behind the one that changed no exit leaves at once (subseq_round_kernel), the
//    flat-region walk is armed or not by subseq_check_kernel; both exit buffers hold the converged states.  Whether the rounds
//    sufficed is the caller's to read from changed_dev when it next waits for the stream anyway (DeviceBatch::sync): the
//    first r >= 1 with changed_dev[r] == 0 says rounds r + 1 were used; none -> not converged, the step is issued again the
//    host-checked way.  *rounds_used is left alone.
//  * device_rounds == 0 (the optimizer, and the decoder's fallback): the host reads the counts every few rounds and stops.
hipError_t launch_subseq_sync(hipStream_t stream, const uint8_t *udata, const DevScan *scans, const HuffWork *work, int n_work,
                              const uint32_t *scan_ids, int n_scans, const uint32_t *ends_u, DevScanStatus *status,
                              const DevHuffTable *huff_pool, uint32_t *exit_a, uint32_t *exit_b, uint32_t *nblk, uint32_t *first_block,
                              uint32_t *entry_used, void *dcsum, void *dc_entry, uint32_t *changed_dev, int n_slots, int max_rounds,
                              int *rounds_used, const uint8_t *lut_pool, const uint32_t **final_state_out, uint32_t *same_dist,
                              bool *same_valid, int device_rounds, const HuffWork *gather_work, int n_gather, const uint8_t *sr_luts) {
    *final_state_out = exit_a;
    if (n_work <= 0 || n_scans <= 0) return hipSuccess;
    const size_t lds_round = ((size_t)n_slots << (kSrLutBits + 2)) + (size_t)n_slots * kK2SmallBytes + kMaxBlocksPerMcu * 16 + 256 * (kSrRingStride + 16) +
                             kMaxHuffSlots * sizeof(uint32_t);

    static const uint32_t warm_bits = [] {  // bits of a subsequence round 0 decodes (0 = all of it); see subseq_round_kernel
        const char *ev = getenv("JPGPU_SUBSEQ_WARM_BITS");
        return ev ? (uint32_t)atoi(ev) : 2048u;
    }();
    static const int propagate_from = [] {  // rounds without convergence before the flat-region walk joins in (0 = never)
        const char *ev = getenv("JPGPU_SUBSEQ_PROPAGATE_FROM");
        return ev ? atoi(ev) : 6;  // (6: behind the second check -- a batch that converges in six rounds never pays for it)
    }();
    uint32_t *bufs[2] = {exit_a, exit_b};
    // changed_dev[r] = exits round r changed; the host looks at the counts only every kCheckEvery rounds (one sync per check),
    // after every round once a round has changed no more than a few exits per scan (the end is then a round or two away);
    // changed_dev[63] counts what the flat-region walks copied
    constexpr int kCheckEvery = 3;
    const uint32_t few_changes = 8u * (uint32_t)n_scans;
    hipError_t e = hipMemsetAsync(changed_dev, 0, (device_rounds > 0 ? (size_t)kSubCtlPerDecode : 64) * sizeof(uint32_t), stream);
    if (e != hipSuccess) return e;
    // round r reads the exits round r - 1 wrote; rounds behind round 1 gather their few lanes (subseq_round_kernel<true>)
#ifdef JPGPU_SUBSEQ_NO_GATHER  // (A/B build variants since round 6: tools/trace/ab_build.sh "-DJPGPU_SUBSEQ_NO_GATHER" ...)
    constexpr bool no_gather = true;
#else
    constexpr bool no_gather = false;
#endif
    auto launch_round = [&](int r, uint32_t *count, const uint32_t *prev) {
        if (r >= 2 && gather_work != nullptr && n_gather > 0 && !no_gather)
            hipLaunchKernelGGL(subseq_round_kernel<true>, dim3(n_gather), dim3(256), lds_round, stream, udata, scans, gather_work, ends_u, status, huff_pool,
                               lut_pool, bufs[(r + 1) & 1], bufs[r & 1], nblk, entry_used, (int4 *)dcsum, count, r, n_slots, warm_bits, prev, sr_luts);
        else
            hipLaunchKernelGGL(subseq_round_kernel<false>, dim3(n_work), dim3(256), lds_round, stream, udata, scans, work, ends_u, status, huff_pool,
                               lut_pool, bufs[(r + 1) & 1], bufs[r & 1], nblk, entry_used, (int4 *)dcsum, count, r, n_slots, warm_bits, prev, sr_luts);
    };
    int round = 0;
    if (device_rounds > 0) {
        const int n_rounds = device_rounds < kSubseqMaxDeviceRounds ? device_rounds : kSubseqMaxDeviceRounds;
        for (; round < n_rounds; round++) {
            launch_round(round, changed_dev + round, round >= 2 ? changed_dev + (round - 1) : nullptr);
            const int n = round + 1;  // rounds issued so far
            if (n < n_rounds && propagate_from > 0 && n >= propagate_from && same_dist != nullptr) {
                // where the host's loop would have checked (and perhaps walked): decided on the device, the walk gated by it
                hipLaunchKernelGGL(subseq_check_kernel, dim3(1), dim3(64), 0, stream, changed_dev, (uint32_t)n, few_changes, (uint32_t)propagate_from);
                hipLaunchKernelGGL(subseq_same_kernel, dim3(n_work), dim3(256), 0, stream, udata, scans, work, ends_u, status, same_dist, changed_dev);
                hipLaunchKernelGGL(subseq_propagate_kernel, dim3(n_scans), dim3(64), 0, stream, scans, scan_ids, same_dist, bufs[(n + 1) & 1], entry_used,
                                   nblk, (int4 *)dcsum, changed_dev + 63, changed_dev);
            }
        }
    }
    bool converged = device_rounds > 0;
    uint32_t last_count = 0xFFFFFFFFu, count_before = 0xFFFFFFFFu;  // exits the last two rounds changed
    while (!converged && round < max_rounds) {
        const int batch_first = round;
        const int batch = last_count <= few_changes ? 1 : kCheckEvery;
        for (int i = 0; i < batch && round < max_rounds; i++, round++) {
            launch_round(round, changed_dev + (round % 62), nullptr);
        }
        uint32_t flags[64];
        e = hipMemcpyAsync(flags, changed_dev, sizeof flags, hipMemcpyDeviceToHost, stream);
        if (e != hipSuccess) return e;
        e = hipStreamSynchronize(stream);
        if (e != hipSuccess) return e;
        // converged as soon as one round (other than round 0) changed nothing: later rounds are then no-ops
        for (int r2 = batch_first; r2 < round; r2++) {
            if (r2 > 0 && flags[r2 % 62] == 0) converged = true;
            count_before = last_count;
            last_count = flags[r2 % 62];
        }
        static const bool trace = getenv("JPGPU_SUBSEQ_TRACE") != nullptr;  // subsequences whose exit changed, per round
        if (trace) {
            for (int r2 = batch_first; r2 < round; r2++) fprintf(stderr, "K2S round %d: %u changed\n", r2, flags[r2 % 62]);
            fprintf(stderr, "K2S walks copied so far: %u\n", flags[63]);
        }
        if (!converged) {
            e = hipMemsetAsync(changed_dev, 0, 64 * sizeof(uint32_t), stream);
            if (e != hipSuccess) return e;
            // Ordinary synchronisation dies out geometrically (a round changes a third to a tenth of what the one before it
            // changed); a flat region changes as many exits round after round (one per run).  The walk is for the second kind.
            const bool plateau = count_before == 0xFFFFFFFFu || (uint64_t)last_count * 2 > count_before;
            if (propagate_from > 0 && round >= propagate_from && same_dist != nullptr && round < max_rounds && plateau) {
                // still not converged (this check said so), and not about to: flat regions?  Twins once per upload, then a walk
                // (it patches the buffer the last round wrote: the next round reads that one).  Decided AFTER the check: a batch
                // that has just converged, or is converging, does not pay for the walk (0.66 ms per 16 benchmark canvases)
                if (!*same_valid) {
                    hipLaunchKernelGGL(subseq_same_kernel, dim3(n_work), dim3(256), 0, stream, udata, scans, work, ends_u, status, same_dist,
                                       (const uint32_t *)nullptr);
                    *same_valid = true;
                }
                hipLaunchKernelGGL(subseq_propagate_kernel, dim3(n_scans), dim3(64), 0, stream, scans, scan_ids, same_dist, bufs[(round + 1) & 1], entry_used,
                                   nblk, (int4 *)dcsum, changed_dev + 63, (uint32_t *)nullptr);
            }
        }
    }
    if (rounds_used && device_rounds <= 0) *rounds_used = round;
    *final_state_out = bufs[(round + 1) & 1];  // buffer written by the last round executed (device-driven: both hold the converged states)
    hipLaunchKernelGGL(subseq_scan_kernel, dim3(n_scans), dim3(1024), 0, stream, scans, scan_ids, nblk, first_block, (const int4 *)dcsum,
                       (int4 *)dc_entry);
    return hipGetLastError();
}

hipError_t launch_subseq_decode(hipStream_t stream, const uint8_t *udata, const DevScan *scans, const HuffWork *work, int n_work,
                                const uint32_t *scan_ids, int n_scans, const uint32_t *ends_u, DevScanStatus *status,
                                const DevHuffTable *huff_pool, uint32_t *exit_a, uint32_t *exit_b, uint32_t *nblk, uint32_t *first_block,
                                uint32_t *entry_used, void *dcsum, void *dc_entry, uint32_t *changed_dev, int16_t *coefs, int n_slots,
                                int max_rounds, int *rounds_used, const uint8_t *lut_pool, const HuffWork *final_work, int n_final_work,
                                uint32_t *same_dist, bool *same_valid, int device_rounds, const HuffWork *gather_work, int n_gather,
                                const HuffWork *pool_work, const SubseqPool *pools, int n_pools, int num_cus, uint32_t *lane_perm, int subs_per_lane,
                                uint32_t tab_bytes, const uint8_t *sr_luts) {
    if (n_work <= 0 || n_scans <= 0) return hipSuccess;
    const uint32_t spl = subs_per_lane >= 2 ? 2u : 1u;
    const uint32_t *final_state = nullptr;
    hipError_t e = launch_subseq_sync(stream, udata, scans, work, n_work, scan_ids, n_scans, ends_u, status, huff_pool, exit_a, exit_b, nblk,
                                      first_block, entry_used, dcsum, dc_entry, changed_dev, n_slots, max_rounds, rounds_used, lut_pool,
                                      &final_state, same_dist, same_valid, device_rounds, gather_work, n_gather, sr_luts);
    if (e != hipSuccess) return e;
    static std::atomic<uint64_t> configured{0}, configured_pool{0};
#ifdef JPGPU_SF_NO_ORDER  // (A/B build variant)
    constexpr bool no_order = true;
#else
    constexpr bool no_order = false;
#endif
    const uint32_t *perm = nullptr;
    if (lane_perm != nullptr && gather_work != nullptr && n_gather > 0 && !no_order) {
        hipLaunchKernelGGL(subseq_order_kernel, dim3(n_gather), dim3(kSubseqGatherSpan / spl), 0, stream, scans, gather_work, status, final_state, first_block,
                           lane_perm, spl);
        perm = lane_perm;
    }
    if (n_final_work > 0) {
        const int waves = subseq_final_waves();
        const size_t lds_final = (size_t)tab_bytes + (size_t)waves * kSfWaveBytes + kMaxBlocksPerMcu * sizeof(uint32_t);
        const hipError_t ea = allow_dynamic_lds(reinterpret_cast<const void *>(&subseq_final_kernel<false>), 160 * 1024, configured);
        if (ea != hipSuccess) return ea;
        // final_work: (scan, first subsequence) per workgroup of waves * 64 lanes (the rounds' work list is per 256)
        hipLaunchKernelGGL(subseq_final_kernel<false>, dim3(n_final_work), dim3(64 * waves), lds_final, stream, udata, scans, final_work, ends_u, status,
                           huff_pool, lut_pool, final_state, first_block, (const int4 *)dc_entry, coefs, n_slots, 0u, (uint32_t *)nullptr, perm, spl, tab_bytes);
    }
    if (n_pools > 0) {
        // (the counters live in the control words every launch_subseq_sync clears; a host-checked launch clears the first 64 only)
        if (device_rounds <= 0) {
            e = hipMemsetAsync(changed_dev + kSubseqCtlPoolCounter, 0, kSubFinalMaxPools * sizeof(uint32_t), stream);
            if (e != hipSuccess) return e;
        }
        const size_t lds_pool = (size_t)tab_bytes + (size_t)kSubFinalPoolWaves * kSfWaveBytes + kMaxBlocksPerMcu * sizeof(uint32_t);
        const hipError_t ea = allow_dynamic_lds(reinterpret_cast<const void *>(&subseq_final_kernel<true>), 160 * 1024, configured_pool);
        if (ea != hipSuccess) return ea;
        for (int p = 0; p < n_pools; p++) {
            const int groups = std::min(num_cus > 0 ? num_cus : 256, (pools[p].count + kSubFinalPoolWaves - 1) / kSubFinalPoolWaves);
            hipLaunchKernelGGL(subseq_final_kernel<true>, dim3(groups), dim3(64 * kSubFinalPoolWaves), lds_pool, stream, udata, scans,
                               pool_work + pools[p].first, ends_u, status, huff_pool, lut_pool, final_state, first_block, (const int4 *)dc_entry, coefs,
                               n_slots, (uint32_t)pools[p].count, changed_dev + kSubseqCtlPoolCounter + p, perm, spl, tab_bytes);
        }
    }
    return hipGetLastError();
}

}  // namespace jpgpu
