// jpeglibrary_amd/csrc/k2p_progressive.hip -- K2P: entropy scans of progressive frames (lane per interval; wave per stream)
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
// K2P: one entropy scan of a progressive frame (ref: ScanDecoder/JpegHuffmanProgressiveScanDecoder.cs).
//
// One lane per restart interval of the scan (one lane for the whole scan when DRI = 0); a workgroup belongs to one
// scan, so the scan kind (DC / AC, first / refinement) is uniform in it.  Scans of a frame are launched in file order
// (refinements depend on what earlier scans stored); all frames of a batch advance together, scan ordinal by ordinal.
// Coefficients accumulate in the frame's MCU-ordered store, the layout the IDCT pass (K3) reads:
//   * DC first / AC first write single coefficients (nothing has to be read: first passes only touch zeros),
//   * DC refinement ORs one bit into the stored DC (fire-and-forget atomic),
//   * AC refinement stages the block in LDS (it branches on every stored coefficient), the next block's 128 bytes are
//     prefetched into registers meanwhile.
// Blocks outside a component's own grid go to the reference's dummy block (JpegBlockAllocator.cs:93-114): decoded,
// not stored.
// ------------------------------------------------------------------------------------------------


typedef uint32_t __attribute__((may_alias)) aliasing_u32;  // word view of a staged int16 block (type-punned on purpose)
constexpr int kProgThreads = 256;
constexpr int kProgBlockStride = 132;  // bytes per lane in the AC-refinement staging: 33 words keep lanes on distinct banks

// Where the blocks of one scan component live in the frame's MCU-ordered coefficient store, held in registers (the
// scan descriptor is read once: nothing in the block loops touches it again).
struct ProgComp {
    uint32_t h, v, hblocks, vblocks, base;
};
struct ProgFrame {
    uint64_t coef_off;
    uint32_t mcus_per_line, bpm;
};
__device__ __forceinline__ ProgComp prog_comp(const DevScan &s, uint32_t c) {
    ProgComp p;
    p.h = s.comp[c].h;
    p.v = s.comp[c].v;
    p.hblocks = s.hblocks[c];
    p.vblocks = s.vblocks[c];
    p.base = s.fblk_base[c];
    return p;
}
// block (bx, by) of a component; false = the allocator's dummy block
// DC refinement, blockRef |= bit << al (JpegHuffmanProgressiveScanDecoder.cs ReadBlockProgressiveDC, the Ah != 0 arm), on the
// 16-bit DC ALONE.  Round 1 spelled this as a 32-bit atomic OR on the block's first word; that read-modify-write also writes
// coefficient 1 back, and coefficient 1 belongs to the AC scans of the same frame, which store it with plain 16-bit writes at
// the same time (same launch level; any time in the pipelined launch): whenever the atomic's read and write straddled such a
// store, coefficient 1 reverted to its old value and the next refinement of that band lost its place in the bit stream
// ("invalid Huffman code" in a handful of frames per thousand, only under tight following -- the long hunt is in DESIGN.md).
// The DC has one writer at a time (DC scans of a component follow each other), so no atomic is needed.
__device__ __forceinline__ void dc_refine_or(int16_t *dc, uint32_t al) {
    *dc = (int16_t)((uint16_t)*dc | (uint16_t)(1u << al));
}
__device__ __forceinline__ bool prog_block_index(const ProgFrame &f, const ProgComp &p, uint32_t bx, uint32_t by, uint64_t &index) {
    if (bx >= p.hblocks || by >= p.vblocks) return false;
    const uint32_t mx = bx / p.h, my = by / p.v;
    index = f.coef_off + ((uint64_t)my * f.mcus_per_line + mx) * f.bpm + p.base + (by - my * p.v) * p.h + (bx - mx * p.h);
    return true;
}

// Raster walk over the blocks of a non-interleaved scan without divisions: (bx, by) plus their split into MCU
// coordinates and position inside the MCU.
struct ProgWalk {
    uint32_t bx, by, mx, rx, my, ry;
};
__device__ __forceinline__ void prog_walk_init(ProgWalk &w, const ProgComp &p, uint32_t unit, uint32_t units_per_line) {
    w.by = unit / units_per_line;
    w.bx = unit - w.by * units_per_line;
    w.mx = w.bx / p.h;
    w.rx = w.bx - w.mx * p.h;
    w.my = w.by / p.v;
    w.ry = w.by - w.my * p.v;
}
__device__ __forceinline__ void prog_walk_next(ProgWalk &w, const ProgComp &p, uint32_t units_per_line) {
    w.bx++;
    if (++w.rx == p.h) {
        w.rx = 0;
        w.mx++;
    }
    if (w.bx == units_per_line) {
        w.bx = w.mx = w.rx = 0;
        w.by++;
        if (++w.ry == p.v) {
            w.ry = 0;
            w.my++;
        }
    }
}
__device__ __forceinline__ bool prog_walk_index(const ProgFrame &f, const ProgComp &p, const ProgWalk &w, uint64_t &index) {
    if (w.bx >= p.hblocks || w.by >= p.vblocks) return false;
    index = f.coef_off + ((uint64_t)w.my * f.mcus_per_line + w.mx) * f.bpm + p.base + w.ry * p.h + w.rx;
    return true;
}

__global__ __launch_bounds__(kProgThreads) void progressive_scan_kernel(const uint8_t *__restrict__ udata, const DevScan *__restrict__ scans,
                                                                        const HuffWork *__restrict__ work,
                                                                        const uint32_t *__restrict__ ends_u,
                                                                        DevScanStatus *__restrict__ status,
                                                                        const DevHuffTable *__restrict__ huff_pool,
                                                                        int16_t *__restrict__ coefs, int n_slots) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t *tabs = smem;                                                    // n_slots * sizeof(DevHuffTable)
    uint8_t *stage_all = smem + (size_t)n_slots * sizeof(DevHuffTable);     // kProgThreads * kProgBlockStride

    const HuffWork wk = work[blockIdx.x];
    const DevScan &s = scans[wk.scan];
    const uint32_t tid = threadIdx.x;
    for (int slot = 0; slot < kMaxHuffSlots && slot < n_slots; slot++) {
        const uint32_t pi = s.huff_pool[slot];
        if (pi == 0xFFFF) continue;
        const uint4 *src = reinterpret_cast<const uint4 *>(&huff_pool[pi]);
        uint4 *dst = reinterpret_cast<uint4 *>(tabs + slot * sizeof(DevHuffTable));
        for (uint32_t i = tid; i < sizeof(DevHuffTable) / 16; i += kProgThreads) dst[i] = src[i];
    }
    __syncthreads();

    const DevScanStatus st = status[wk.scan];
    const uint32_t n_ends = st.n_ends;
    const uint32_t n_intervals = s.n_intervals;
    const uint32_t total_units = s.total_mcus;
    const uint32_t dri_eff = s.dri ? s.dri : total_units;
    const uint32_t interval = wk.first_interval + tid;
    if (interval >= n_ends || interval > s.last_interval) return;  // nothing after the last barrier: lanes may leave
    const uint32_t *eu = ends_u + s.ends_off;
    const uint32_t ustart = interval == 0 ? 0u : eu[interval - 1] + 2u;
    UBits r;
    ub_init(r, udata + s.data_off, ustart, eu[interval]);
    const bool closed_by_marker = !(interval == n_ends - 1 && st.terminator == 0);
    const uint32_t my_units = (interval == n_intervals - 1) ? total_units - interval * dri_eff : dri_eff;
    const uint32_t first_unit = interval * dri_eff;

    // the scan descriptor, once
    const uint32_t al = s.al, ah = s.ah, ss = s.ss, se = s.se, ncomp = s.scan_components, units_per_line = s.units_per_line;
    ProgFrame fr;
    fr.coef_off = s.coef_off;
    fr.mcus_per_line = s.mcus_per_line;
    fr.bpm = s.frame_bpm;
    uint32_t err = 0;

    if (ncomp != 1) {
        // ---- interleaved scans are DC scans (:92-138), ReadBlockProgressiveDC (:227-253)
        ProgComp pc[kMaxScanComponents];
        uint32_t dc_slot[kMaxScanComponents];
#pragma unroll
        for (uint32_t c = 0; c < kMaxScanComponents; c++) {
            pc[c] = prog_comp(s, c < ncomp ? c : 0);
            dc_slot[c] = s.comp[c < ncomp ? c : 0].dc_slot;
        }
        int32_t pred[kMaxScanComponents] = {0, 0, 0, 0};
        uint32_t uy = first_unit / units_per_line, ux = first_unit - uy * units_per_line;
        for (uint32_t u = 0; u < my_units && err == 0; u++) {
#pragma unroll
            for (uint32_t c = 0; c < kMaxScanComponents; c++) {
                if (c >= ncomp || err != 0) continue;
                const ProgComp p = pc[c];
                const LdsHuff hdc = lds_huff(tabs, dc_slot[c]);
                for (uint32_t y = 0; y < p.v && err == 0; y++)
                    for (uint32_t x = 0; x < p.h; x++) {
                        uint64_t index = 0;
                        const bool real = prog_block_index(fr, p, ux * p.h + x, uy * p.v + y, index);
                        if (ah == 0) {
                            uint32_t sym;
                            int32_t value;
                            err = ub_symbol(r, hdc, true, closed_by_marker, sym, value);
                            if (err != 0) break;
                            const int32_t t = pred[c] + value;
                            pred[c] = t;
                            if (real) coefs[index * 64] = (int16_t)((uint32_t)t << al);
                        } else {
                            uint32_t bit;
                            if (!ub_try_read_bits(r, 1, bit)) {
                                err = kDetailUnexpectedEnd;
                                break;
                            }
                            if (real && bit) dc_refine_or(coefs + index * 64, al);
                        }
                    }
            }
            if (++ux == units_per_line) {
                ux = 0;
                uy++;
            }
        }
    } else {
        const ProgComp p = prog_comp(s, 0);
        ProgWalk w;
        prog_walk_init(w, p, first_unit, units_per_line);
        if (ss == 0) {
            // ---- DC scan of one component (:148-168)
            const LdsHuff hdc = lds_huff(tabs, s.comp[0].dc_slot);
            int32_t pred = 0;
            for (uint32_t u = 0; u < my_units; u++, prog_walk_next(w, p, units_per_line)) {
                uint64_t index = 0;
                const bool real = prog_walk_index(fr, p, w, index);
                if (ah == 0) {
                    uint32_t sym;
                    int32_t value;
                    err = ub_symbol(r, hdc, true, closed_by_marker, sym, value);
                    if (err != 0) break;
                    pred += value;
                    if (real) coefs[index * 64] = (int16_t)((uint32_t)pred << al);
                } else {
                    uint32_t bit;
                    if (!ub_try_read_bits(r, 1, bit)) {
                        err = kDetailUnexpectedEnd;
                        break;
                    }
                    if (real && bit) dc_refine_or(coefs + index * 64, al);
                }
            }
        } else if (ah == 0) {
            // ---- AC first pass (:255-311)
            const LdsHuff hac = lds_huff(tabs, s.comp[0].ac_slot);
            uint32_t eobrun = 0;
            for (uint32_t u = 0; u < my_units && err == 0; u++, prog_walk_next(w, p, units_per_line)) {
                if (eobrun != 0) {
                    eobrun--;
                    continue;
                }
                uint64_t index = 0;
                const bool real = prog_walk_index(fr, p, w, index);
                int16_t *blk = coefs + index * 64;
                for (uint32_t i = ss; i <= se; i++) {
                    uint32_t sym;
                    int32_t value;
                    err = ub_symbol(r, hac, false, closed_by_marker, sym, value);
                    if (err != 0) break;
                    const uint32_t rr = sym >> 4;
                    i += rr;
                    if ((sym & 15u) != 0) {
                        if (real) blk[i < 63u ? i : 63u] = (int16_t)((uint32_t)value << al);
                    } else if (rr != 15u) {
                        eobrun = 1u << rr;
                        if (rr != 0) {
                            uint32_t bits;
                            if (!ub_try_read_bits(r, rr, bits)) {
                                err = kDetailUnexpectedEnd;
                                break;
                            }
                            eobrun += bits;
                        }
                        eobrun--;
                        break;
                    }
                }
            }
        } else {
            // ---- AC refinement (:313-419)
            // The reference walks the band one coefficient at a time: a correction bit for every coefficient that is
            // already non-zero, counting down the run over the ones that are still zero.  Which coefficients are non-zero
            // is fixed when the block is staged, so the walk is done on a 64-bit mask: the stop position is the
            // (r + 1)-th zero bit, the correction bits of the non-zero positions passed are read as one field.
            const LdsHuff hac = lds_huff(tabs, s.comp[0].ac_slot);
            const int16_t p1 = (int16_t)(1u << al), m1 = (int16_t)(0xFFFFFFFFu << al);
            int16_t *lb = reinterpret_cast<int16_t *>(stage_all + tid * kProgBlockStride);
            aliasing_u32 *lw = reinterpret_cast<aliasing_u32 *>(lb);
            const uint64_t band = (se >= 63u ? ~0ull : ((1ull << (se + 1u)) - 1ull)) & ~((1ull << ss) - 1ull);
            uint32_t eobrun = 0;
            // the next block is loaded into registers while the current one is decoded in LDS
            uint4 n0, n1, n2, n3, n4, n5, n6, n7;
            bool nb_real = false;
            uint64_t nb_index = 0;
#define JPGPU_PREFETCH_BLOCK(have_)                                                         \
    {                                                                                       \
        nb_real = (have_) && prog_walk_index(fr, p, w, nb_index);                           \
        const uint4 *src_ = reinterpret_cast<const uint4 *>(coefs + (nb_real ? nb_index : fr.coef_off) * 64); \
        n0 = src_[0]; n1 = src_[1]; n2 = src_[2]; n3 = src_[3];                             \
        n4 = src_[4]; n5 = src_[5]; n6 = src_[6]; n7 = src_[7];                             \
    }
            // correction bits of the non-zero coefficients in `m_` (ascending zig-zag order == stream order)
#define JPGPU_CORRECT(m_, strict_)                                                                      \
    {                                                                                                   \
        uint64_t mm_ = (m_);                                                                            \
        uint32_t left_ = (uint32_t)__builtin_popcountll(mm_);                                           \
        while (left_ != 0 && err == 0) {                                                                \
            uint32_t n_ = left_ < 16u ? left_ : 16u;                                                    \
            uint32_t field_ = 0;                                                                        \
            /* the reference reads these one at a time: when the data ends inside the field, the corrections in   */ \
            /* front of the end are applied before it throws (the partial flush of a failing file shows them)      */ \
            const bool short_ = (int32_t)n_ > r.rem;                                                    \
            if (short_) n_ = r.rem > 0 ? (uint32_t)r.rem : 0u;                                          \
            if (n_ != 0) (void)ub_try_read_bits(r, n_, field_);                                         \
            if (short_) err = kDetailUnexpectedEnd;                                                     \
            for (uint32_t i_ = 0; i_ < n_; i_++) {                                                      \
                const uint32_t pos_ = (uint32_t)__builtin_ctzll(mm_);                                   \
                mm_ &= mm_ - 1;                                                                         \
                if ((field_ >> (n_ - 1u - i_)) & 1u) {                                                  \
                    const int16_t c_ = lb[pos_];                                                        \
                    if ((c_ & p1) == 0) {                                                               \
                        const int16_t nc_ = (int16_t)(c_ + ((strict_ ? c_ > 0 : c_ >= 0) ? p1 : m1));   \
                        lb[pos_] = nc_;                                                                 \
                        if (real) gblk[pos_] = nc_;                                                     \
                    }                                                                                   \
                }                                                                                       \
            }                                                                                           \
            left_ -= n_;                                                                                \
        }                                                                                               \
    }
            JPGPU_PREFETCH_BLOCK(my_units > 0)
            for (uint32_t u = 0; u < my_units && err == 0; u++) {
                const bool real = nb_real;
                const uint64_t index = nb_index;
                uint64_t nz = 0;  // bit k: coefficient k of the block is non-zero before this scan touches it
#define JPGPU_STAGE(i_, v_)                                                                                           \
    lw[(i_) * 4 + 0] = (v_).x; lw[(i_) * 4 + 1] = (v_).y; lw[(i_) * 4 + 2] = (v_).z; lw[(i_) * 4 + 3] = (v_).w;       \
    {                                                                                                                 \
        const uint32_t q_[4] = {(v_).x, (v_).y, (v_).z, (v_).w};                                                      \
        uint32_t b_ = 0;                                                                                              \
        for (int j_ = 0; j_ < 4; j_++) b_ |= (((q_[j_] & 0xFFFFu) != 0 ? 1u : 0u) | ((q_[j_] >> 16) != 0 ? 2u : 0u)) << (2 * j_); \
        nz |= (uint64_t)b_ << (8 * (i_));                                                                             \
    }
                JPGPU_STAGE(0, n0) JPGPU_STAGE(1, n1) JPGPU_STAGE(2, n2) JPGPU_STAGE(3, n3)
                JPGPU_STAGE(4, n4) JPGPU_STAGE(5, n5) JPGPU_STAGE(6, n6) JPGPU_STAGE(7, n7)
#undef JPGPU_STAGE
                prog_walk_next(w, p, units_per_line);
                JPGPU_PREFETCH_BLOCK(u + 1 < my_units)

                // every change is written through as a 2-byte store: scans of other bands / the DC refinement of the same
                // blocks may run concurrently (host: ProgressiveFrame::add_scan levels)
                int16_t *gblk = coefs + index * 64;
                uint32_t k = ss;
                if (eobrun == 0) {
                    for (; k <= se; k++) {
                        uint32_t sym;
                        err = ub_huff(r, hac, sym);
                        if (err != 0) break;
                        const uint32_t rr = sym >> 4;
                        int16_t sval = 0;
                        const bool nonzero = (sym & 15u) != 0;
                        if (nonzero) {
                            uint32_t bit;
                            if (!ub_try_read_bits(r, 1, bit)) {
                                err = kDetailUnexpectedEnd;
                                break;
                            }
                            sval = bit ? p1 : m1;
                        } else if (rr != 15u) {
                            eobrun = 1u << rr;
                            if (rr != 0) {
                                uint32_t bits;
                                if (!ub_try_read_bits(r, rr, bits)) {
                                    err = kDetailUnexpectedEnd;
                                    break;
                                }
                                eobrun += bits;
                            }
                            break;
                        }
                        // the do/while of :340-372: stop at the (r + 1)-th still-zero coefficient at or after k
                        const uint64_t from_k = band & ~((1ull << k) - 1ull);
                        uint64_t z = ~nz & from_k;
                        for (uint32_t j = 0; j < rr && z != 0; j++) z &= z - 1;
                        const uint32_t stop = z != 0 ? (uint32_t)__builtin_ctzll(z) : se + 1u;
                        const uint64_t passed = nz & from_k & (stop >= 64u ? ~0ull : ((1ull << stop) - 1ull));
                        JPGPU_CORRECT(passed, false)
                        if (err != 0) break;
                        k = stop;
                        if (nonzero && k < 64u) {
                            lb[k] = sval;
                            if (real) gblk[k] = sval;
                        }
                    }
                }
                if (err == 0 && eobrun > 0) {
                    if (k <= se) {
                        const uint64_t rest = nz & band & ~((1ull << k) - 1ull);
                        JPGPU_CORRECT(rest, true)
                    }
                    eobrun--;
                }
            }
#undef JPGPU_PREFETCH_BLOCK
#undef JPGPU_CORRECT
        }
    }

    // HandleRestart (:196-224) after the interval's last unit: same rules as the sequential decoder's restart check
    const uint32_t code = restart_check(s, st, &status[wk.scan], interval, n_ends, n_intervals, dri_eff, r.rem, err);
    if (code != kNoError) atomicMin(&status[wk.scan].first_error, code);
}

// ------------------------------------------------------------------------------------------------
// K2P, wave-per-stream form: the same scan semantics for scans with FEW, LONG restart intervals (DRI = 0: one stream per
// scan).  A lone lane of the lane-per-interval kernel above pays a memory round trip per block and runs the
// coefficient-by-coefficient walks as scalar loops on one SIMD lane.  Here ONE WAVE owns the stream and decodes it as a
// wave-UNIFORM program: the decoder state (bit position, EOB run, zig-zag position, the 64-bit masks) lives in scalar
// registers, the 64 lanes are the 64 coefficients of the current block:
//   * the unstuffed stream is staged MSB-first in a 4 KB LDS ring, topped up 1 KB at a time by all lanes (small enough that
//     all ten scans of 256 frames are resident at once);
//   * a WINDOW holds, per lane l, the 32 stream bits at offset l past the window base and the Huffman lookup of those bits
//     (one LDS gather for 64 bit offsets at once); decoding a symbol is then two v_readlane at the current offset plus
//     scalar arithmetic, the window is rebuilt when the offset runs past 63 (every ~7 symbols);
//   * lane j holds coefficient j: the "non-zero before this scan" mask is one ballot, a new coefficient is a predicated
//     move, a correction field is spread over the lanes by rank (mbcnt) instead of a loop over its bits, and the block's
//     changes leave as ONE masked 2-byte store instruction (other bands / the DC bit of the same blocks may be written
//     concurrently by other scans);
//   * refinement blocks are staged in LDS in rounds of kPsChunk blocks by all lanes (addresses computed by the lanes).
// ------------------------------------------------------------------------------------------------
// LDS per stream (one wave): the stream ring (a power of two >= 2 KB: top-ups come in 1 KB pieces) and the staging of the
// refinement blocks.  Both are launch parameters: the smaller they are, the more streams a CU holds (the kernel is bound
// by instruction issue latency, co-resident waves are what hides it) -- 2 KB + 16 blocks = 23 streams per CU with two
// Huffman tables, 4 KB + 32 blocks = 14 (JPGPU_PS_RING / JPGPU_PS_CHUNK, A/B in profiles/r02_progressive_lds.txt).
constexpr int32_t kPsUnitBytes = 384;     // stream bytes staged before a block / MCU is started (unless the stream ends)
constexpr uint32_t kPsNoBlock = 0xFFFFFFFFu;
constexpr uint32_t kPsBadCode = 17u << 8;  // window entry: no code of 16 bits or less matches


// the wave's LDS accesses so far are complete before those behind this line start (one wave per workgroup: no s_barrier)
#define PS_WAVE_SYNC()                                        \
    do {                                                      \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    \
        __builtin_amdgcn_wave_barrier();                      \
    } while (0)

// Scope of the release / acquire pair of the pipelined progressive launch: "agent" (buffer_wbl2 sc1 / buffer_inv sc1), what the
// memory model asks for between workgroups of one device.  A build switch because the system-scope forms were tried during the
// hunt for the wrong parses that turned out to be dc_refine_or's story (they only moved the timing); no measurable cost either way.
#ifndef JPGPU_PS_SCOPE
#define JPGPU_PS_SCOPE "agent"
#endif
#ifdef JPGPU_PS_PROFILE
// cycle accounting of the refinement path of progressive_stream_kernel (diagnostic build only: -DJPGPU_PS_PROFILE)
__device__ unsigned long long ps_prof[16];
#define PS_TICK() __builtin_readcyclecounter()
#define PS_ADD(i, v) do { if (lane == 0) atomicAdd(&ps_prof[i], (unsigned long long)(v)); } while (0)
#if JPGPU_PS_PROFILE > 1
#define PS_COUNT(i) PS_ADD(i, 1)  /* event counts (perturbs the timing: use for counts only) */
#else
#define PS_COUNT(i) do { } while (0)
#endif
extern "C" int jpgpu_debug_ps_profile(unsigned long long *out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(ps_prof), sizeof(ps_prof)) != hipSuccess) return 1;
    if (reset) { unsigned long long z[16] = {}; if (hipMemcpyToSymbol(HIP_SYMBOL(ps_prof), z, sizeof z) != hipSuccess) return 1; }
    return 0;
}
#else
#define PS_TICK() 0ull
#define PS_ADD(i, v) do { (void)(v); } while (0)
#define PS_COUNT(i) do { } while (0)
#endif

#ifdef JPGPU_PS_TRACE
// What every luma AC refinement block was given and what it made of it (diagnostic build only, tools/trace/refine_trace.py): per
// (image >= first, scan kind, block) four words -- which of the 64 coefficients the scan found non-zero, its bit position in front
// of the block, the bits it consumed and the end-of-band run it left.  Copies of one source image must agree word for word; where
// a failing copy first differs says whether it READ something else or PARSED differently.
__device__ uint32_t *ps_trace_buf;
__device__ uint32_t ps_trace_first_image, ps_trace_images, ps_trace_units;
extern "C" int jpgpu_debug_ps_trace(uint32_t *device_buffer, uint32_t first_image, uint32_t images, uint32_t units) {
    return hipMemcpyToSymbol(HIP_SYMBOL(ps_trace_buf), &device_buffer, sizeof device_buffer) != hipSuccess ||
           hipMemcpyToSymbol(HIP_SYMBOL(ps_trace_first_image), &first_image, 4) != hipSuccess ||
           hipMemcpyToSymbol(HIP_SYMBOL(ps_trace_images), &images, 4) != hipSuccess ||
           hipMemcpyToSymbol(HIP_SYMBOL(ps_trace_units), &units, 4) != hipSuccess;
}
#endif

struct WBits {
    const uint32_t *ring;  // MSB-first words of the stream; ring byte 0 = the 16-byte aligned address at or below its first byte
    uint32_t wmask;        // uniform: ring size in words - 1
    uint32_t pos;          // uniform: bit position of the next unread bit
    uint32_t cur;          // uniform: pos - window base; > 63 = the window has to be rebuilt
    int32_t rem;           // uniform: bits left before the interval's end (the reference's "bits available")
    uint32_t peek;         // per lane: the 32 bits at window base + lane
    uint32_t ent;          // per lane: (code size << 8) | symbol for those bits, 0 = longer than the lookup
    uint32_t ent2;         // per lane, refinement scans only: `ent` pre-digested for the scalar symbol loop (r2_digest)
#ifdef JPGPU_PS_PROFILE
    unsigned long long t_pro = 0, t_loop = 0, t_epi = 0, t_refresh = 0, n_exits = 0;  // cycles inside w_ac_refine_parse
    uint32_t n_trips = 0;  // trips of the symbol loop (JPGPU_PS_PROFILE > 1)
#endif
};

// JpegHuffmanDecodingTable.Lookup for one 16-bit code, all lanes the same (ref: JpegHuffmanDecodingTable.cs:73-113)
__device__ __forceinline__ uint32_t w_huff_scalar(const LdsHuff &h, uint32_t code16) {
    const uint32_t e = uni(h.lut[code16 >> (16 - kHuffLutBits)]);
    if ((e >> 8) != 0) return e;
    uint32_t size = kHuffLutBits + 1;
    while (code16 > uni(h.maxcode[size])) size++;  // maxcode[17] = 0xFFFF terminates
    if (size > 16) return kPsBadCode;
    return (size << 8) | uni(h.values[(uni(h.valoffset[size]) + (code16 >> (16 - size))) & 0xFF]);
}

template <bool LUT>
__device__ __forceinline__ void w_refresh(WBits &d, uint32_t lane, const LdsHuff &h) {
    const uint32_t bit = d.pos + lane;
    const uint32_t w = bit >> 5, sh = bit & 31u;
    const uint32_t w0 = d.ring[w & d.wmask], w1 = d.ring[(w + 1u) & d.wmask];
    d.peek = (uint32_t)(((((uint64_t)w0) << 32) | w1) >> (32u - sh));
    if (LUT) d.ent = h.lut[d.peek >> (32 - kHuffLutBits)];
    d.cur = 0;
    PS_COUNT(7);
}

// DecodeHuffmanCode (ref: ScanDecoder/JpegHuffmanScanDecoder.cs:81-88); returns the entry, the peeked bits in pk.
// FAST = the caller has checked that the interval holds more bits than a whole block can consume: the reference's
// "bits available" rules (JpegBitReader.cs:157-204) cannot trigger and `rem` is settled once per block instead.
template <bool LUT, bool FAST>
__device__ __forceinline__ uint32_t w_code(WBits &d, uint32_t lane, const LdsHuff &h, uint32_t &pk) {
    if (d.cur > 63u) w_refresh<LUT>(d, lane, h);
    pk = lane_get(d.peek, d.cur);
    uint32_t e;
    if (FAST || d.rem > 0) {
        e = LUT ? lane_get(d.ent, d.cur) : 0u;
        if ((e >> 8) == 0) e = w_huff_scalar(h, pk >> 16);
    } else {
        e = w_huff_scalar(h, 0xFFFFu);  // PeekBits with nothing left: all ones
    }
    return e;
}
template <bool LUT, bool FAST>
__device__ __forceinline__ uint32_t w_huff(WBits &d, uint32_t lane, const LdsHuff &h, uint32_t &sym_out) {
    uint32_t pk;
    const uint32_t e = w_code<LUT, FAST>(d, lane, h, pk);
    const uint32_t size = e >> 8;
    if (size > 16u) return kDetailInvalidHuffmanCode;
    sym_out = e & 0xFFu;
    if (!FAST) d.rem = d.rem > (int32_t)size ? d.rem - (int32_t)size : 0;  // advance Math.Min(entry.CodeSize, bitsRead)
    d.pos += size;
    d.cur += size;
    return 0;
}
// symbol + ReceiveAndExtend (ref: ScanDecoder/JpegHuffmanScanDecoder.cs:100-115), as ub_symbol
template <bool LUT, bool FAST>
__device__ __forceinline__ uint32_t w_symbol(WBits &d, uint32_t lane, const LdsHuff &h, bool is_dc, bool closed_by_marker,
                                             uint32_t &sym_out, int32_t &value) {
    uint32_t pk;
    const uint32_t e = w_code<LUT, FAST>(d, lane, h, pk);
    const uint32_t size = e >> 8, sym = e & 0xFFu;
    if (size > 16u) return kDetailInvalidHuffmanCode;
    sym_out = sym;
    const uint32_t s = is_dc ? sym : (sym & 15u);
    if (!FAST) d.rem = d.rem > (int32_t)size ? d.rem - (int32_t)size : 0;
    value = 0;
    if (s != 0) {
        if (s > 16u) return kDetailInvalidHuffmanCode;
        if (!FAST && (int32_t)s > d.rem) return (d.rem == 0 && closed_by_marker) ? kDetailMarkerInData : kDetailStreamEnded;
        const int32_t v = (int32_t)((pk << size) >> (32u - s));
        value = v - ((((v + v) >> s) - 1) & ((1 << s) - 1));  // Extend(v, nbits)
        if (!FAST) d.rem -= (int32_t)s;
    }
    d.pos += size + s;
    d.cur += size + s;
    return 0;
}
// TryReadBits(n), 1 <= n <= 32
template <bool LUT, bool FAST>
__device__ __forceinline__ bool w_read_bits(WBits &d, uint32_t lane, const LdsHuff &h, uint32_t n, uint32_t &bits) {
    if (!FAST && (int32_t)n > d.rem) return false;
    if (d.cur > 63u) w_refresh<LUT>(d, lane, h);
    bits = lane_get(d.peek, d.cur) >> (32u - n);
    if (!FAST) d.rem -= (int32_t)n;
    d.pos += n;
    d.cur += n;
    return true;
}

// One AC first-pass block (:255-311): lane j = coefficient j, `changed` = positions written.
template <bool FAST>
__device__ __forceinline__ uint32_t w_ac_first_block(WBits &d, uint32_t lane, const LdsHuff &hac, bool closed_by_marker, uint32_t ss,
                                                     uint32_t se, uint32_t al, uint32_t &eobrun, int32_t &c, uint64_t &changed) {
    for (uint32_t i = ss; i <= se; i++) {
        uint32_t sym;
        int32_t value;
        const uint32_t err = w_symbol<true, FAST>(d, lane, hac, false, closed_by_marker, sym, value);
        if (err != 0) return err;
        const uint32_t rr = sym >> 4;
        i += rr;
        if ((sym & 15u) != 0) {
            const uint32_t at = i < 63u ? i : 63u;
            if (lane == at) c = (int32_t)((uint32_t)value << al);
            changed |= 1ull << at;
        } else if (rr != 15u) {
            eobrun = 1u << rr;
            if (rr != 0) {
                uint32_t bits;
                if (!w_read_bits<true, FAST>(d, lane, hac, rr, bits)) return kDetailUnexpectedEnd;
                eobrun += bits;
            }
            eobrun--;
            break;
        }
    }
    return 0;
}

// Correction bits of the non-zero coefficients in `mask` (ascending zig-zag order == stream order): one field, spread
// over the lanes by rank.  (:349-361, :386-401)
template <bool FAST>
__device__ __forceinline__ uint32_t w_correct(WBits &d, uint32_t lane, const LdsHuff &hac, uint64_t mask, uint32_t count, bool strict,
                                              int32_t p1, int32_t m1, int32_t &c, bool &mine) {
    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
    const bool in_mask = ((mask >> lane) & 1ull) != 0;
    uint32_t taken = 0;
    while (count != 0) {
        const uint32_t n = count < 32u ? count : 32u;
        uint32_t field;
        if (!w_read_bits<true, FAST>(d, lane, hac, n, field)) return kDetailUnexpectedEnd;
        const uint32_t r = rank - taken;
        if (in_mask && r < n && ((field >> (n - 1u - r)) & 1u) != 0 && (c & p1) == 0) {
            c += (strict ? c > 0 : c >= 0) ? p1 : m1;
            mine = true;
        }
        taken += n;
        count -= n;
    }
    return 0;
}

// One AC refinement block (:313-419).  The reference walks the band one coefficient at a time: a correction bit for every
// coefficient that is already non-zero (`nz`, fixed when the scan reaches the block), counting down the run over the
// ones that are still zero.  On the masks: the stop position is the (r + 1)-th zero at or after k -- the lane whose
// zero-rank matches -- and the positions passed hold exactly (stop - k - r) non-zero coefficients.
template <bool FAST>
__device__ __forceinline__ uint32_t w_ac_refine_block(WBits &d, uint32_t lane, const LdsHuff &hac, uint32_t ss, uint32_t se, int32_t p1,
                                                      int32_t m1, uint64_t band, uint64_t nz, uint32_t &eobrun, int32_t &c, bool &mine) {
    uint32_t k = ss;
    const uint64_t zeros = ~nz & band;
    // zero-rank of every lane: zeros of the band strictly below it
    const uint32_t zrank = __builtin_amdgcn_mbcnt_hi((uint32_t)(zeros >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)zeros, 0u));
    const bool is_zero = ((zeros >> lane) & 1ull) != 0;
    if (eobrun == 0) {
        for (; k <= se; k++) {
            uint32_t sym;
            const uint32_t err = w_huff<true, FAST>(d, lane, hac, sym);
            if (err != 0) return err;
            const uint32_t rr = sym >> 4;
            int32_t sval = 0;
            const bool nonzero = (sym & 15u) != 0;
            if (nonzero) {
                uint32_t bit;
                if (!w_read_bits<true, FAST>(d, lane, hac, 1, bit)) return kDetailUnexpectedEnd;
                sval = bit ? p1 : m1;
            } else if (rr != 15u) {
                eobrun = 1u << rr;
                if (rr != 0) {
                    uint32_t bits;
                    if (!w_read_bits<true, FAST>(d, lane, hac, rr, bits)) return kDetailUnexpectedEnd;
                    eobrun += bits;
                }
                break;
            }
            // zeros of the band below k, then the lane that is the (rr + 1)-th zero from k on
            const uint32_t below = (uint32_t)__builtin_popcountll(zeros & ((1ull << k) - 1ull));
            const uint64_t hit = __ballot(is_zero && zrank == below + rr);
            uint32_t stop, count;
            if (hit != 0) {
                stop = (uint32_t)__builtin_ctzll(hit);
                count = stop - k - rr;
            } else {
                stop = se + 1u;  // the run outlasts the band
                count = (uint32_t)__builtin_popcountll(nz & band & ~((1ull << k) - 1ull));
            }
            if (count != 0) {
                const uint64_t passed = nz & band & ~((1ull << k) - 1ull) & (stop >= 64u ? ~0ull : ((1ull << stop) - 1ull));
                const uint32_t cerr = w_correct<FAST>(d, lane, hac, passed, count, false, p1, m1, c, mine);
                if (cerr != 0) return cerr;
            }
            k = stop;
            if (nonzero && k < 64u && lane == k) {
                c = sval;
                mine = true;
            }
        }
    }
    if (eobrun > 0) {
        if (k <= se) {
            const uint64_t rest = nz & band & ~((1ull << k) - 1ull);
            if (rest != 0) {
                const uint32_t cerr = w_correct<FAST>(d, lane, hac, rest, (uint32_t)__builtin_popcountll(rest), true, p1, m1, c, mine);
                if (cerr != 0) return cerr;
            }
        }
        eobrun--;
    }
    return 0;
}
// ---- AC first pass, bulk path, parse-only form (same idea as w_ac_refine_parse): the serial loop reads the pre-digested window
// entry, advances the zig-zag index and the position, and writes down in lane n where symbol n's coefficient goes, how many
// magnitude bits it has and where the symbol ends; afterwards every symbol lane cuts its own magnitude out of the LDS ring, extends
// it (ReceiveAndExtend, :100-115) and stores it.  (ReadBlockProgressiveAC, :255-311.)
constexpr uint32_t kF2Special = 1u << 20;
__device__ __forceinline__ uint32_t f2_digest(uint32_t e /* (code size << 8) | symbol; code size 0 = not in the lookup */) {
    const uint32_t size = e >> 8, rr = (e >> 4) & 15u, sz = e & 15u;
    const uint32_t special = (size == 0 || (sz == 0 && rr != 15u)) ? kF2Special : 0u;
    return special | (rr << 12) | (sz << 6) | (size + sz);  // bits 0-5: code + magnitude bits, 6-10: magnitude bits, 12-15: run
}
// (The loop below writes M0 itself -- v_writelane with two scalar operands needs the lane select there on gfx9 -- and says so in
// its clobber list; clang warns that M0 is reserved.  It is safe here: this kernel issues no LDS-DMA and no other instruction that
// reads M0 implicitly, and hipcc re-materialises M0 in front of every use of its own (it never keeps a value there across
// statements).  K3, the one kernel whose global_load_lds reads M0, contains no inline asm that touches it.)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ uint32_t w_ac_first_parse(WBits &d, uint32_t lane, const LdsHuff &hac, uint32_t ss, uint32_t se, uint32_t al,
                                                     uint32_t &eobrun, int16_t *blk /* nullptr = the dummy block */) {
    se = uni(se);
    const uint32_t ringbits = uni((d.wmask + 1u) * 32u - 1u);
    uint32_t rec = 0;  // lane n: position (<= 63) | magnitude bits << 6 | ring bit position behind the symbol << 11
    uint32_t nsym = 0, i = uni(ss);
    uint32_t cur = uni(d.cur);
    uint32_t winpos = uni(d.pos) - cur;
    for (;;) {
        uint32_t reason, e, t, at;
        cur = uni(cur);
        i = uni(i);
        nsym = uni(nsym);
        asm volatile(
            "1:\n\t"
            "s_cmp_gt_u32 %[cur], 63\n\t"
            "s_cbranch_scc1 2f\n\t"
            "v_readlane_b32 %[e], %[ent2], %[cur]\n\t"
            "s_cmp_ge_u32 %[e], 0x100000\n\t"
            "s_cbranch_scc1 3f\n\t"
            "s_bfe_u32 %[t], %[e], 0x4000c\n\t"
            "s_add_u32 %[i], %[i], %[t]\n\t"
            "s_min_u32 %[at], %[i], 63\n\t"
            "s_and_b32 %[t], %[e], 63\n\t"
            "s_add_u32 %[cur], %[cur], %[t]\n\t"
            "s_add_u32 %[t], %[winpos], %[cur]\n\t"
            "s_and_b32 %[t], %[t], %[ringbits]\n\t"
            "s_lshl_b32 %[t], %[t], 11\n\t"
            "s_and_b32 %[e], %[e], 0x7c0\n\t"
            "s_or_b32 %[t], %[t], %[e]\n\t"
            "s_or_b32 %[t], %[t], %[at]\n\t"
            "s_mov_b32 m0, %[nsym]\n\t"
            "v_writelane_b32 %[rec], %[t], m0\n\t"
            "s_add_u32 %[nsym], %[nsym], 1\n\t"
            "s_add_u32 %[i], %[i], 1\n\t"
            "s_cmp_le_u32 %[i], %[se]\n\t"
            "s_cbranch_scc1 1b\n\t"
            "s_mov_b32 %[reason], 0\n\t"
            "s_branch 5f\n"
            "2:\n\t"
            "s_mov_b32 %[reason], 1\n\t"
            "s_branch 5f\n"
            "3:\n\t"
            "s_mov_b32 %[reason], 2\n"
            "5:\n\t"
            : [cur] "+s"(cur), [i] "+s"(i), [nsym] "+s"(nsym), [rec] "+v"(rec), [reason] "=&s"(reason), [e] "=&s"(e), [t] "=&s"(t), [at] "=&s"(at)
            : [winpos] "s"(winpos), [ringbits] "s"(ringbits), [se] "s"(se), [ent2] "v"(d.ent2)
            : "scc", "m0", "memory");
        cur = uni(cur);
        i = uni(i);
        nsym = uni(nsym);
        reason = uni(reason);
        if (reason == 0) break;
        if (reason == 1) {
            d.pos = winpos + cur;
            w_refresh<true>(d, lane, hac);
            d.ent2 = f2_digest(d.ent);
            winpos = uni(d.pos);
            cur = 0;
            continue;
        }
        // long code or end-of-band symbol
        const uint32_t pk = lane_get(d.peek, cur);
        uint32_t raw = lane_get(d.ent, cur);
        if ((raw >> 8) == 0) {
            raw = w_huff_scalar(hac, pk >> 16);
            if (raw >= kPsBadCode) return kDetailInvalidHuffmanCode;
        }
        const uint32_t size = raw >> 8, rr = (raw >> 4) & 15u, sz = raw & 15u;
        if (sz == 0 && rr != 15u) {
            eobrun = (1u << rr) - 1u + (uint32_t)(((uint64_t)(pk << size) << rr) >> 32);  // rr = 0 reads nothing
            cur += size + rr;
            break;
        }
        i += rr;
        cur += size + sz;
        rec = lane_put(rec, (((winpos + cur) & ringbits) << 11) | (sz << 6) | (i < 63u ? i : 63u), nsym);
        nsym++;
        if (++i > se) break;
    }
    d.cur = cur;
    d.pos = winpos + cur;
    nsym = uni(nsym);
    const uint32_t sz = (rec >> 6) & 31u, at = rec & 63u;
    bool store = lane < nsym && sz != 0 && blk != nullptr;
    // a corrupted stream may run the index past 63: the reference then overwrites coefficient 63 again and again (:283), the last wins
    const uint64_t at63 = __ballot(store && at == 63u);
    if (at63 != 0 && lane != 63u - (uint32_t)__builtin_clzll(at63) && at == 63u) store = false;
    if (store) {
        const uint32_t mp = ((rec >> 11) - sz) & ringbits;  // the first magnitude bit
        const uint32_t w0 = d.ring[(mp >> 5) & d.wmask], w1 = d.ring[((mp >> 5) + 1u) & d.wmask];
        const uint32_t top = (uint32_t)(((((uint64_t)w0) << 32) | w1) >> (32u - (mp & 31u)));
        const int32_t v = (int32_t)(top >> (32u - sz));
        const int32_t value = v - ((((v + v) >> sz) - 1) & ((1 << sz) - 1));  // Extend(v, nbits)
        blk[at] = (int16_t)((uint32_t)value << al);
    }
    return 0;
}

#pragma clang diagnostic pop
// ---- AC refinement, bulk path, fourth form: the symbol loop in the VECTOR unit's instruction stream.
// tools/microbench/issue_latency.hip (one wave, cycles per instruction): any simple instruction ~4.2; a SALU instruction that
// reads an SGPR the VALU has just written (v_readlane -> s_cmp, v_cmp -> s_and) stalls ~20 more; a conditional branch costs ~15
// even when it is NOT taken, ~16-20 when it is; a dependent LDS read 53, ds_bpermute 61.  The third form above (scalar loop,
// 33 instructions, two such crossings and four branches per symbol) therefore ran at ~465 cycles per symbol.  Here every value
// of the chain lives in a VGPR (the same in all lanes), the only SGPRs are the lane selects of the two v_readlane (written by
// v_readfirstlane, read by the VALU: no stall), nothing is decided by a branch but the loop itself (commits are selects on one
// "this symbol is a plain one inside the window and its run ends inside the band" condition), and the stop position is not
// searched for but looked up:
//   * per block, lane r of `ntab` holds Ss + (the number of non-zero coefficients of the band below its r-th zero), scattered
//     there by one ds_permute; the stop of a symbol that consumes zeros up to rank t is then t + ntab[t], and the correction
//     bits its run passes are ntab[t] - Ss minus those passed before: position = block's first offset + code bits + ntab[t];
//   * the new coefficient (sign bit pre-digested into the window entry) goes into lane `stop` of the block's register, the
//     lanes from the symbol's first position on note the code bits consumed so far (their correction bit comes after them);
//   * afterwards every lane that was non-zero fetches ITS correction bit: block start + noted code bits + its rank.
constexpr uint32_t kR4Special = 0x80000000u, kR4Eob = 0x40000000u, kR4Zrl = 0x2000u;
// window entry as the loop wants it: code bits (+ 1 sign bit) in bits 0-5, run + 1 in bits 6-12 (64 + for a symbol the loop
// cannot apply: its target then lies behind every zero), ZRL in bit 13, the new coefficient itself in bits 14-29; bit 31: not
// for the loop, bit 30: ... because it is EOBn, n = run
__device__ __forceinline__ uint32_t r4_digest(uint32_t e /* (code size << 8) | symbol; code size 0 = not in the lookup */, uint32_t peek,
                                              int32_t p1, int32_t m1) {
    const uint32_t size = e >> 8, rr = (e >> 4) & 15u, nonzero = (e & 15u) != 0 ? 1u : 0u;
    const bool special = size == 0 || (nonzero == 0 && rr != 15u);
    const uint32_t sign = (peek << (size & 31u)) >> 31;  // the bit behind the code
    const uint32_t value = (uint32_t)(sign != 0 ? p1 : m1) & 0xFFFFu;
    // (an end-of-band symbol found in the lookup says so: what ends almost every block is then applied without a second lookup)
    return (special ? kR4Special | (size != 0 ? kR4Eob : 0u) | (64u << 6) : (value << 14)) | (nonzero != 0 ? 0u : kR4Zrl) | ((rr + 1u) << 6) |
           (size + nonzero);
}

constexpr uint32_t kR4NoZero = 0xFFu;  // zero-table entry: no such zero, the run outlasts the band
// (the first, third and fourth forms of the refinement decoder: tools/microbench/refine_forms.inc, ablation builds only)
#ifdef JPGPU_PS_EARLIER_FORMS
#include "../../tools/microbench/refine_forms.inc"
#endif
// ---- AC refinement, bulk path, fifth form: the WHOLE block in one instruction sequence.
// rocprofv3's counters on the fourth form (tools/trace/progressive_pmc.sh, per block of the last luma refinement): 480 wave
// instructions of which the symbol loop is ~200, 30 branches, and as many wave cycles waiting as issuing -- one wave issues one
// instruction every four cycles whatever it is (tools/microbench/fetch_rate.hip), a branch costs ~16 more, a scalar instruction
// that reads what a vector instruction has just written ~20.  The compiler's code around the loop (masks and ranks of the
// band, the zero table, the hand-over of uniform values between the scalar and the vector unit, the correction bits, a
// state machine of a dozen branches for the loop's exits) is therefore written out here as well, straight-line and in the
// vector unit's instruction stream: prologue (~40 instructions), the loop and its end-of-band tail as in the fourth form, the
// correction bits (~25).  What it does not do -- rebuild the window, decode a code longer than the lookup, a run that
// outlasts the band -- it hands back (status 1) with all of its state in registers, and is re-entered at the loop (`resume`
// 1) or at the correction bits (2) once the C++ below has dealt with it.
// the per-lane constants of the sequence, made once per stream and held in registers (hipcc would otherwise re-materialise
// them in front of every block)
struct R5Consts {
    uint32_t lane, inband /* all ones in the lanes Ss..Se */, p1, m1, none, zrl, noz;
};
__device__ __forceinline__ R5Consts r5_consts(uint32_t lane, uint64_t band, int32_t p1, int32_t m1) {
    R5Consts k;
    k.lane = lane;
    k.inband = ((band >> lane) & 1ull) != 0 ? 0xFFFFFFFFu : 0u;
    k.p1 = (uint32_t)p1;
    k.m1 = (uint32_t)m1;
    k.none = 0xFFFFu;
    k.zrl = kR4Zrl;
    k.noz = kR4NoZero;
    asm volatile("" : "+v"(k.lane), "+v"(k.inband), "+v"(k.p1), "+v"(k.m1), "+v"(k.none), "+v"(k.zrl), "+v"(k.noz));
    return k;
}
#define JPGPU_R5_PROLOGUE \
    /* ---- prologue: the band's non-zero coefficients (ranks, count) and its zeros (ranks, count); the zero table -- */ \
    /* lane r: Ss + the number of non-zero coefficients below the r-th zero -- by one ds_permute */ \
    "v_and_b32_e32 %[x0], %[c], %[inb]\n\t" \
    "v_not_b32_e32 %[x1], %[inb]\n\t" \
    "v_or_b32_e32 %[x1], %[x1], %[c]\n\t" \
    "v_cmp_ne_u32_e32 vcc, 0, %[x0]\n\t" \
    "v_mov_b32_e32 %[cv], %[c]\n\t" \
    "v_mov_b32_e32 %[symbits], 0\n\t" \
    "v_mbcnt_lo_u32_b32 %[nrank], vcc_lo, 0\n\t" \
    "v_mbcnt_hi_u32_b32 %[nrank], vcc_hi, %[nrank]\n\t" \
    "v_bcnt_u32_b32 %[nnz], vcc_lo, 0\n\t" \
    "v_bcnt_u32_b32 %[nnz], vcc_hi, %[nnz]\n\t" \
    "v_cmp_eq_u32_e32 vcc, 0, %[x1]\n\t" \
    "v_mov_b32_e32 %[bits], 0\n\t" \
    "v_add_u32_e32 %[x3], %[ss], %[nrank]\n\t" \
    "v_mbcnt_lo_u32_b32 %[x0], vcc_lo, 0\n\t" \
    "v_mbcnt_hi_u32_b32 %[x0], vcc_hi, %[x0]\n\t"  /* rank among the zeros */ \
    "v_bcnt_u32_b32 %[x1], vcc_lo, 0\n\t" \
    "v_bcnt_u32_b32 %[x1], vcc_hi, %[x1]\n\t"  /* zeros in the band */ \
    "v_sub_u32_e32 %[x2], %[lane], %[x0]\n\t" \
    "v_add_u32_e32 %[x2], %[x2], %[x1]\n\t" \
    "v_cndmask_b32_e32 %[x2], %[x2], %[x0], vcc\n\t"  /* r-th zero -> lane r, the others behind (a permutation) */ \
    "v_lshlrev_b32_e32 %[x2], 2, %[x2]\n\t" \
    "ds_permute_b32 %[ntab], %[x2], %[x3]\n\t" \
    "v_cmp_gt_u32_e32 vcc, %[x1], %[lane]\n\t"  /* (two wait states before the select below reads VCC, however soon the permute is back) */ \
    "v_mov_b32_e32 %[cur], %[cur0]\n\t" \
    "v_subrev_u32_e32 %[base], %[ss], %[cur]\n\t"  /* a symbol's window offset = base + code bits before it + ntab[zeros before it] */ \
    "v_mov_b32_e32 %[zq], -1\n\t" \
    "v_add_u32_e64 %[kprev], %[ss], -1\n\t" \
    "s_waitcnt lgkmcnt(0)\n\t" \
    "v_cndmask_b32_e32 %[ntab], %[noz], %[ntab], vcc\n\t"  /* no such zero: the run outlasts the band */ \
    "v_cmp_ne_u32_e32 vcc, 0, %[eobv]\n\t"  /* inside an end-of-band run: correction bits only */ \
    "s_cbranch_vccnz 5f\n\t"
#define JPGPU_R5_LOOP_TAIL_END \
    /* ---- the symbol loop (fourth form): one symbol per trip; the previous symbol's commits fill the hazard gaps */ \
    "2:\n\t" \
    "v_mov_b32_e32 %[stop], %[none]\n\t" \
    "s_mov_b32 %[se_], 0\n\t" \
    "s_mov_b64 %[sok], 0\n\t" \
    "s_mov_b64 %[spb], 0\n\t" \
    "1:\n\t" \
    "v_readfirstlane_b32 %[scur], %[cur]\n\t" \
    "v_and_or_b32 %[x5], %[se_], %[zrl], %[stop]\n\t" \
    "v_cndmask_b32_e64 %[symbits], %[symbits], %[x3], %[sok]\n\t" \
    "v_cmp_eq_u32_e64 %[spp], %[lane], %[x5]\n\t" \
    "v_cndmask_b32_e64 %[bits], %[bits], %[symbits], %[spb]\n\t" \
    "v_readlane_b32 %[se_], %[ent2], %[scur]\n\t" \
    "v_cndmask_b32_e64 %[kprev], %[kprev], %[x9], %[sok]\n\t" \
    "v_cndmask_b32_e64 %[zq], %[zq], %[x2], %[sok]\n\t" \
    "v_bfe_u32 %[x0], %[se_], 6, 7\n\t" \
    "v_cndmask_b32_e64 %[cv], %[cv], %[x8], %[spp]\n\t" \
    "v_add_u32_e32 %[x2], %[zq], %[x0]\n\t"  /* x2 = t: rank of the zero the symbol's run ends on */ \
    "v_min_u32_e32 %[x1], 63, %[x2]\n\t" \
    "v_and_b32_e64 %[x4], 63, %[se_]\n\t" \
    "v_readfirstlane_b32 %[st], %[x1]\n\t" \
    "v_add_u32_e32 %[x3], %[symbits], %[x4]\n\t"  /* x3 = symn */ \
    "v_bfe_u32 %[x8], %[se_], 14, 16\n\t"  /* x8 = the new coefficient */ \
    "v_cmp_gt_u32_e64 %[spb], %[lane], %[kprev]\n\t" \
    "s_nop 0\n\t" \
    "v_readlane_b32 %[sn], %[ntab], %[st]\n\t" \
    "s_nop 1\n\t" \
    "v_add3_u32 %[x6], %[base], %[x3], %[sn]\n\t"  /* x6 = curn */ \
    "v_or3_b32 %[x7], %[x2], %[cur], %[sn]\n\t" \
    "v_cmp_gt_u32_e64 %[sok], 64, %[x7]\n\t" \
    "v_add_u32_e32 %[x9], %[sn], %[x2]\n\t"  /* x9 = stopr */ \
    "s_nop 0\n\t" \
    "v_cndmask_b32_e64 %[cur], %[cur], %[x6], %[sok]\n\t" \
    "v_cndmask_b32_e64 %[stop], %[none], %[x9], %[sok]\n\t" \
    "v_cmp_gt_u32_e32 vcc, %[se], %[stop]\n\t" \
    "s_cbranch_vccnz 1b\n\t" \
    "v_and_or_b32 %[x5], %[se_], %[zrl], %[stop]\n\t" \
    "v_cndmask_b32_e64 %[symbits], %[symbits], %[x3], %[sok]\n\t" \
    "v_cmp_eq_u32_e64 %[spp], %[lane], %[x5]\n\t" \
    "v_cndmask_b32_e64 %[bits], %[bits], %[symbits], %[spb]\n\t" \
    "v_cndmask_b32_e64 %[kprev], %[kprev], %[x9], %[sok]\n\t" \
    "v_cndmask_b32_e64 %[zq], %[zq], %[x2], %[sok]\n\t" \
    "v_cndmask_b32_e64 %[cv], %[cv], %[x8], %[spp]\n\t" \
    /* what stopped the loop?  EOBn found in the lookup is applied here (fourth form's tail) */ \
    "v_readlane_b32 %[sn], %[peek], %[scur]\n\t" \
    "v_bfe_u32 %[x0], %[se_], 6, 5\n\t" \
    "v_and_b32_e64 %[x4], 63, %[se_]\n\t" \
    "v_add_u32_e32 %[x0], -1, %[x0]\n\t" \
    "v_lshlrev_b32_e64 %[x7], %[x4], %[sn]\n\t" \
    "v_sub_u32_e32 %[x1], 32, %[x0]\n\t" \
    "v_bfe_u32 %[x7], %[x7], %[x1], %[x0]\n\t" \
    "v_lshlrev_b32_e64 %[x6], %[x0], 1\n\t" \
    "v_add_u32_e32 %[x6], %[x6], %[x7]\n\t"  /* the end-of-band run */ \
    "v_add3_u32 %[x3], %[symbits], %[x4], %[x0]\n\t" \
    "v_lshrrev_b32_e32 %[x5], 6, %[cur]\n\t" \
    "v_bfe_u32 %[x7], %[se_], 30, 1\n\t" \
    "v_xor_b32_e32 %[x7], 1, %[x7]\n\t" \
    "v_or_b32_e32 %[x5], %[x5], %[x7]\n\t" \
    "v_cmp_eq_u32_e64 %[spp], %[stop], %[none]\n\t" \
    "v_cmp_gt_u32_e64 %[spb], %[lane], %[kprev]\n\t" \
    "s_nop 0\n\t" \
    "v_cndmask_b32_e64 %[slow], 0, %[x5], %[spp]\n\t" \
    "v_cndmask_b32_e64 %[x5], 1, %[x5], %[spp]\n\t" \
    "v_cmp_eq_u32_e64 %[sok], 0, %[x5]\n\t" \
    "v_cmp_ne_u32_e32 vcc, 0, %[slow]\n\t" \
    "s_nop 0\n\t" \
    "v_cndmask_b32_e64 %[symbits], %[symbits], %[x3], %[sok]\n\t" \
    "v_cndmask_b32_e64 %[eobv], 0, %[x6], %[sok]\n\t" \
    "v_cndmask_b32_e64 %[bits], %[bits], %[symbits], %[spb]\n\t" \
    "v_mov_b32_e32 %[status], 1\n\t" \
    "s_cbranch_vccnz 9f\n\t"  /* something for the C++ below */ \
    /* ---- the block's end: this block counts against the run; every coefficient of the band that was non-zero takes */ \
    /* exactly one correction bit, behind the code bits noted for its lane, in rank order */ \
    "5:\n\t" \
    "v_sub_u32_e64 %[eobv], %[eobv], 1 clamp\n\t" \
    "v_add3_u32 %[x0], %[pos0], %[bits], %[nrank]\n\t" \
    "v_lshrrev_b32_e32 %[x1], 5, %[x0]\n\t" \
    "v_and_b32_e32 %[x1], %[wmask], %[x1]\n\t" \
    "v_lshl_add_u32 %[x1], %[x1], 2, %[ring]\n\t" \
    "ds_read_b32 %[x1], %[x1]\n\t" \
    "v_not_b32_e32 %[x0], %[x0]\n\t" \
    "v_and_b32_e32 %[x2], %[c], %[inb]\n\t" \
    "v_and_b32_e32 %[x3], %[p1v], %[c]\n\t" \
    "v_cmp_ne_u32_e64 %[spp], 0, %[x2]\n\t"  /* was non-zero, in the band */ \
    "v_cmp_eq_u32_e64 %[sok], 0, %[x3]\n\t"  /* this bit not set yet */ \
    "v_cmp_gt_i32_e64 %[spb], 0, %[c]\n\t" \
    "v_add_u32_e32 %[consumed], %[symbits], %[nnz]\n\t" \
    "v_mov_b32_e32 %[status], 0\n\t" \
    "v_cndmask_b32_e64 %[x2], %[p1v], %[m1v], %[spb]\n\t" \
    "v_cndmask_b32_e64 %[x2], 0, %[x2], %[sok]\n\t" \
    "v_cndmask_b32_e64 %[x2], 0, %[x2], %[spp]\n\t" \
    "s_waitcnt lgkmcnt(0)\n\t" \
    "v_lshrrev_b32_e32 %[x1], %[x0], %[x1]\n\t" \
    "v_bfe_i32 %[x1], %[x1], 0, 1\n\t" \
    "v_and_b32_e32 %[x2], %[x2], %[x1]\n\t" \
    "v_add_u32_e32 %[cn], %[cv], %[x2]\n\t" \
    "v_bfe_i32 %[cn], %[cn], 0, 16\n\t" \
    "9:\n\t"
#define JPGPU_R5_OPERANDS \
    : [cv] "+v"(cv), [eobv] "+v"(eobv), [cur] "+v"(cur), [zq] "+v"(zq), [symbits] "+v"(symbits), [kprev] "+v"(kprev), \
    [bits] "+v"(bits), [ntab] "+v"(ntab), [nrank] "+v"(nrank), [nnz] "+v"(nnz), [base] "+v"(base), [stop] "=&v"(stop), \
    [slow] "=&v"(slow), [status] "=&v"(status), [consumed] "=&v"(consumed), [cn] "=&v"(cn), [scur] "=&s"(scur), [st] "=&s"(st), \
    [sn] "=&s"(sn), [se_] "=&s"(se_), [sok] "=&s"(sok), [spb] "=&s"(spb), [spp] "=&s"(spp), [x0] "=&v"(x0), [x1] "=&v"(x1), \
    [x2] "=&v"(x2), [x3] "=&v"(x3), [x4] "=&v"(x4), [x5] "=&v"(x5), [x6] "=&v"(x6), [x7] "=&v"(x7), [x8] "=&v"(x8), [x9] "=&v"(x9) \
    : [c] "v"(c), [inb] "v"(k.inband), [ent2] "v"(d.ent2), [peek] "v"(d.peek), [lane] "v"(k.lane), [p1v] "v"(k.p1), [m1v] "v"(k.m1), \
    [none] "v"(k.none), [zrl] "v"(k.zrl), [noz] "v"(k.noz), [ss] "s"(ss), [se] "s"(se), [pos0] "s"(blockpos), \
    [cur0] "s"(cur0), [wmask] "s"(wmask), [ring] "s"(ring_lds), [resume] "s"(resume_s)

// `eobv`: the end-of-band run, kept in a vector register from block to block (the same in all lanes).
__device__ __forceinline__ uint32_t w_ac_refine_v5(WBits &d, const R5Consts &k, const LdsHuff &hac, uint32_t ss, uint32_t se, int32_t p1,
                                                   int32_t m1, uint32_t &eobv, int32_t &c, bool &mine) {
    const uint32_t lane = k.lane;
    ss = uni(ss);
    se = uni(se);
    const uint32_t blockpos = uni(d.pos), cur0 = uni(d.cur), wmask = uni(d.wmask);
    const uint32_t ring_lds = uni((uint32_t)(uintptr_t)(__attribute__((address_space(3))) const uint32_t *)d.ring);
    uint32_t winpos = blockpos - cur0;  // stream position of the window's first bit
    auto undefined = [] {
        uint32_t x;
        asm volatile("" : "=v"(x));  // (a register, no instruction: the sequence below sets it)
        return x;
    };
    uint32_t cv = undefined(), cur = undefined(), zq = undefined(), symbits = undefined(), kprev = undefined(), bits = undefined(),
             ntab = undefined(), nrank = undefined(), nnz = undefined(), base = undefined();
    uint32_t status, consumed, cn, fail = 0;
    {
        uint32_t stop, slow, scur, st, sn, se_, x0, x1, x2, x3, x4, x5, x6, x7, x8, x9;
        uint64_t sok, spb, spp;
        const uint32_t resume_s = 0;  // (not read on this path)
        asm volatile(JPGPU_R5_PROLOGUE JPGPU_R5_LOOP_TAIL_END JPGPU_R5_OPERANDS : "vcc", "scc", "memory");
    }
    if (uni(status) != 0) {
        // ---- handed back (about one block in five): the window is used up, or one symbol by hand; then in again at the loop
        // (1) or at the correction bits (2).  One way round this loop and one way out of it: with a `return` or a `break` in the
        // middle hipcc turns the exits into a state variable and a dozen scalar branches.
        uint32_t resume = 1;
        do {
            cur = uni(cur);
            zq = uni(zq);
            symbits = uni(symbits);
            kprev = uni(kprev);
            resume = 1;
            if (cur > 63u) {
                d.pos = winpos + cur;
                w_refresh<true>(d, lane, hac);
                d.ent2 = r4_digest(d.ent, d.peek, p1, m1);
                winpos = uni(d.pos);
                base -= cur;
                cur = 0;
            } else {
                // a code longer than the lookup (EOBn among them), or a run that outlasts the band
                const uint32_t pk = lane_get(d.peek, cur);
                uint32_t raw = lane_get(d.ent, cur);
                if ((raw >> 8) == 0) raw = w_huff_scalar(hac, pk >> 16);
                const uint32_t e2 = r4_digest(raw, pk, p1, m1);
                const uint32_t size = raw >> 8, r2 = (raw >> 4) & 15u;
                if (raw >= kPsBadCode) {
                    fail = kDetailInvalidHuffmanCode;
                    resume = 2;  // (out through the block's end; the result is not used)
                } else if (e2 >= kR4Special) {  // EOBn (:337-350): the run's low bits follow the code; the tail's lanes come behind all of it
                    eobv = (1u << r2) + (uint32_t)(((uint64_t)(pk << size) << r2) >> 32);
                    symbits += size + r2;
                    if (lane > kprev) bits = symbits;
                    resume = 2;
                } else {
                    const uint32_t adv2 = e2 & 63u, tgt2 = zq + 1u + r2;
                    const uint32_t n2 = lane_get(ntab, tgt2 < 63u ? tgt2 : 63u);
                    symbits += adv2;
                    if (lane > kprev) bits = symbits;
                    const uint32_t stop2 = n2 >= 64u ? se + 1u : tgt2 + n2;  // no such zero: the new coefficient lands behind Se (:363-367)
                    if ((e2 & kR4Zrl) == 0 && lane == stop2) cv = (e2 >> 14) & 0xFFFFu;
                    const bool on = n2 < 64u && stop2 < se;  // (else every non-zero coefficient left has been passed)
                    cur = on ? base + symbits + n2 : cur;
                    zq = on ? tgt2 : zq;
                    kprev = on ? stop2 : kprev;
                    resume = on ? 1u : 2u;
                }
            }
            uint32_t stop, slow, scur, st, sn, se_, x0, x1, x2, x3, x4, x5, x6, x7, x8, x9;
            uint64_t sok, spb, spp;
            const uint32_t resume_s = uni(resume);
            asm volatile(
                "s_cmp_eq_u32 %[resume], 2\n\t"
                "s_cbranch_scc1 5f\n\t"
                JPGPU_R5_LOOP_TAIL_END JPGPU_R5_OPERANDS
                : "vcc", "scc", "memory");
        } while (uni(status) != 0);
    }
    if (fail != 0) return fail;
    d.pos = blockpos + uni(consumed);
    d.cur = d.pos - winpos;
    mine = (int32_t)cn != c;  // a new coefficient is never 0, a correction never leaves the value alone
    c = (int32_t)cn;
    return 0;
}
constexpr int32_t kPsFastBits = 2560;  // more than any block can consume: 63 x (16 + 16) + 14 (first), 63 x 17 + 63 + 14 (refinement)


// One scan (one restart interval of it) on the calling wave; the kernel below runs it for its work item and, in the pipelined
// launch, for the scans chained behind that one (DevScan::wave_next).
__device__ __forceinline__ void ps_run_scan(const uint8_t *__restrict__ udata, const DevScan *__restrict__ scans, HuffWork wk,
                                            const uint32_t *__restrict__ ends_u, DevScanStatus *__restrict__ status,
                                            const DevHuffTable *__restrict__ huff_pool, int16_t *__restrict__ coefs, int n_slots,
                                            int pipelined, uint32_t spin_budget, uint32_t ring_bytes, uint32_t chunk_blocks,
                                            uint32_t *__restrict__ started, bool first_in_wave) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t *tabs = smem;  // n_slots * sizeof(DevHuffTable)
    uint8_t *base = smem + (size_t)n_slots * sizeof(DevHuffTable);
    uint32_t *ring = reinterpret_cast<uint32_t *>(base);
    const uint32_t kPsRingBytes = ring_bytes, kPsChunk = chunk_blocks;  // uniform launch parameters (see kPsRingMax)
    int16_t *stage = reinterpret_cast<int16_t *>(base + kPsRingBytes);
    uint32_t *idx = reinterpret_cast<uint32_t *>(base + kPsRingBytes + kPsChunk * 128);

    const DevScan &s = scans[wk.scan];
    const uint32_t lane = threadIdx.x;
    PS_WAVE_SYNC();  // (a chained scan: the previous one's LDS reads are done)
    for (int slot = 0; slot < kMaxHuffSlots && slot < n_slots; slot++) {
        const uint32_t pi = s.huff_pool[slot];
        if (pi == 0xFFFF) continue;
        const uint4 *src = reinterpret_cast<const uint4 *>(&huff_pool[pi]);
        uint4 *dst = reinterpret_cast<uint4 *>(tabs + slot * sizeof(DevHuffTable));
        for (uint32_t i = lane; i < sizeof(DevHuffTable) / 16; i += 64) dst[i] = src[i];
    }

    const DevScanStatus st = status[wk.scan];
    const uint32_t n_ends = st.n_ends;
    const uint32_t n_intervals = s.n_intervals;
    const uint32_t total_units = s.total_mcus;
    const uint32_t dri_eff = s.dri ? s.dri : total_units;
    const uint32_t interval = wk.first_interval;
    if (interval >= n_ends) {  // no data for this interval; followers must not wait for it
        if (pipelined != 0 && s.publishes != 0 && lane == 0)
            __hip_atomic_store(&status[wk.scan].pad[1], 0xFFFFFFFFu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    const uint32_t *eu = ends_u + s.ends_off;
    const uint32_t ustart = interval == 0 ? 0u : eu[interval - 1] + 2u;
    const uint32_t uend = eu[interval];
    const bool closed_by_marker = !(interval == n_ends - 1 && st.terminator == 0);
    const uint32_t my_units = (interval == n_intervals - 1) ? total_units - interval * dri_eff : dri_eff;
    const uint32_t first_unit = interval * dri_eff;

    // ---- the stream ring
    const uint8_t *p0 = udata + s.data_off + ustart;
    const uint32_t skip = (uint32_t)(reinterpret_cast<uintptr_t>(p0) & 15u);
    const uint8_t *origin = p0 - skip;
    const uint32_t fill_end = ((skip + (uend - ustart) + 15u) & ~15u) + 16u;  // staged up to here (<= 32 bytes past the stream)
    uint32_t fill_hi = 0;
    uint4 *ring16 = reinterpret_cast<uint4 *>(ring);
    WBits d;
    d.ring = ring;
    d.wmask = kPsRingBytes / 4u - 1u;
    d.pos = skip * 8u;
    d.cur = 64;
    d.rem = (int32_t)((uend - ustart) * 8u);
    d.peek = d.ent = 0;
    // everything before the window base (16-byte granules) may be overwritten
#define JPGPU_TOP_UP()                                                                                       \
    {                                                                                                        \
        const uint32_t rp_ = ((d.pos - (d.cur > 63u ? 0u : d.cur)) >> 3) & ~15u;                             \
        while (fill_hi < fill_end && fill_hi + 1024u <= rp_ + kPsRingBytes) {                                \
            const uint32_t off_ = fill_hi + lane * 16u;                                                      \
            if (off_ < fill_end) {                                                                           \
                uint4 q_ = *reinterpret_cast<const uint4 *>(origin + off_);                                  \
                q_.x = __builtin_bswap32(q_.x); q_.y = __builtin_bswap32(q_.y);                              \
                q_.z = __builtin_bswap32(q_.z); q_.w = __builtin_bswap32(q_.w);                              \
                ring16[(off_ & (kPsRingBytes - 1u)) >> 4] = q_;                                              \
            }                                                                                                \
            fill_hi += 1024u;                                                                                \
        }                                                                                                    \
        __syncthreads();                                                                                     \
    }
    // a unit's worst case is staged, or everything there is
#define JPGPU_ENSURE_STAGED()                                                                                \
    if (fill_hi < fill_end && (int32_t)(fill_hi - (d.pos >> 3)) < kPsUnitBytes) JPGPU_TOP_UP()
    JPGPU_TOP_UP()
    // the exact block decoders are compiled as divergent code (lane-predicated updates next to early returns); the
    // decoder state they leave is uniform all the same: saying so keeps the bulk path on the scalar unit
#define JPGPU_SETTLE()                      \
    {                                       \
        d.pos = uni(d.pos);                 \
        d.cur = uni(d.cur);                 \
        d.rem = (int32_t)uni((uint32_t)d.rem); \
        eobrun = uni(eobrun);               \
        err = uni(err);                     \
    }

    const uint32_t al = s.al, ah = s.ah, ss = s.ss, se = s.se, ncomp = s.scan_components, units_per_line = s.units_per_line;
    ProgFrame fr;
    fr.coef_off = s.coef_off;
    fr.mcus_per_line = s.mcus_per_line;
    fr.bpm = s.frame_bpm;
    uint32_t err = 0;

    // ---- one launch for all scans of all frames (pipelined != 0): a scan that refines what earlier scans stored follows
    // them MCU row by MCU row.  Every scan that has followers publishes the number of restart units it has completed
    // (after a device-scope release fence) every kPsPublishEvery units; a follower converts that to whole MCU rows of
    // the frame -- the one currency scans of different interleaving share -- and waits (acquire) before it touches a row.
    // Workgroups start in work-list order and the list is sorted by dependency level, so whatever a scan waits for
    // is running or finished.  That order is what the dispatcher is OBSERVED to do, not a contract (HIP promises no dispatch
    // order): every poll draws on `spin_budget`; a scan that exhausts it gives up with kDetailSpinTimeout, publishes
    // "finished" so that its own followers drain too, and the host re-issues the frame's scans level by level in fresh
    // launches (DeviceBatch::fetch_status).
    const bool publishes = pipelined != 0 && s.publishes != 0;
    uint32_t *my_progress = &status[wk.scan].pad[1];
    const uint32_t my_units_per_row = units_per_line * (ncomp == 1 ? (uint32_t)s.comp[0].v : 1u);
    uint32_t dep_scan[3], dep_units_per_row[3];
    uint32_t rows_ready = pipelined != 0 ? 0u : 0xFFFFFFFFu;  // MCU rows every producer has completed
#pragma unroll
    for (int k = 0; k < 3; k++) {
        dep_scan[k] = pipelined != 0 ? s.dep[k] : kNoDep;
        dep_units_per_row[k] = 1;
        if (dep_scan[k] != kNoDep) {
            const DevScan &ds = scans[dep_scan[k]];
            dep_units_per_row[k] = ds.units_per_line * (ds.scan_components == 1 ? (uint32_t)ds.comp[0].v : 1u);
        }
    }
    if (dep_scan[0] == kNoDep) rows_ready = 0xFFFFFFFFu;
    // RESIDENCY RULE.  The follow-your-producers protocol below is run with every workgroup of the launch co-resident: HIP
    // promises no dispatch order, so a follower that holds a slot while its producer still waits for one can starve the
    // machine.  Every workgroup counts itself in at its start; a follower goes on only once all have (a fully resident
    // grid starts within a microsecond), and gives up with kDetailSpinTimeout otherwise -- the host then re-issues the
    // scans level by level.  The host only chooses the pipelined launch for grids that fit (DeviceBatch::run_progressive),
    // so this is the safety net for co-tenants on the device and for an occupancy estimate that was too generous.
    // (Larger grids do work on MI355X as dispatched today -- workgroups start in work-list order, producers first -- and
    // JPGPU_PROG_FORCE_PIPELINE=1 runs them that way, skipping the count-in; measured no faster than level by level.)
    if (pipelined != 0) {
        if (lane == 0 && first_in_wave) __hip_atomic_fetch_add(started, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (dep_scan[0] != kNoDep && pipelined != 2) {  // pipelined == 2: a grid that is not resident (JPGPU_PROG_FORCE_PIPELINE)
            uint32_t polls = spin_budget < 4096u ? spin_budget : 4096u;
            for (;;) {
                const uint32_t n_ = uni(__hip_atomic_load(started, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                if (n_ >= gridDim.x) break;
                if (polls == 0) {
                    err = kDetailSpinTimeout;
                    break;
                }
                polls--;
                __builtin_amdgcn_s_sleep(8);
            }
        }
    }
    // (tests: JPGPU_DEBUG_DELAY_SCAN makes a scan slow -- it idles this long at its start and after every progress word)
    const uint32_t debug_delay = uni((uint32_t)s.debug_delay_ms * 256u);
#define JPGPU_DEBUG_DELAY() \
    if (debug_delay != 0)   \
        for (uint32_t i_ = 0; i_ < debug_delay; i_++) __builtin_amdgcn_s_sleep(127);  // ~4 us each
    JPGPU_DEBUG_DELAY()
    // wait until the producers have finished MCU row `row_`
#define JPGPU_FOLLOW(row_)                                                                                      \
    if ((row_) >= rows_ready) {                                                                                 \
        for (;;) {                                                                                              \
            uint32_t r_ = 0xFFFFFFFFu;                                                                          \
            _Pragma("unroll") for (int k_ = 0; k_ < 3; k_++) {                                                  \
                if (dep_scan[k_] == kNoDep) continue;                                                           \
                const uint32_t p_ = __hip_atomic_load(&status[dep_scan[k_]].pad[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); \
                const uint32_t q_ = p_ == 0xFFFFFFFFu ? p_ : p_ / dep_units_per_row[k_];                       \
                r_ = q_ < r_ ? q_ : r_;                                                                         \
            }                                                                                                   \
            rows_ready = uni(r_);                                                                               \
            if ((row_) < rows_ready) break;                                                                     \
            if (spin_budget == 0) {                                                                             \
                err = kDetailSpinTimeout;                                                                       \
                break;                                                                                          \
            }                                                                                                   \
            spin_budget--;                                                                                      \
            __builtin_amdgcn_s_sleep(32);                                                                       \
        }                                                                                                       \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, JPGPU_PS_SCOPE);                                               \
        /* the invalidate completes asynchronously: nothing may be loaded before it has (MI355X guide, G16) */  \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                        \
    }
    // Release: the wave's own stores drained, the XCD's L2 written back, and -- spelled out in asm because hipcc (ROCm 7.2)
    // drops the wait behind buffer_wbl2 when it believes the counter is empty, which would let the flag overtake the
    // write-back -- only then the progress word.
#define JPGPU_PUBLISH(units_)                                                                                   \
    if (publishes) {                                                                                            \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                        \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, JPGPU_PS_SCOPE);                                               \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                        \
        if (lane == 0) __hip_atomic_store(my_progress, (units_), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   \
        JPGPU_DEBUG_DELAY()                                                                                     \
    }
    // How often?  A release is a write-back of the XCD's L2 -- of every wave's dirty lines, not only the publisher's -- so its
    // cost grows with the batch: with the first scans and the DC scans publishing every 64 units a 256-frame launch took 214 ms
    // against 165 ms for 32 frames; at 512 it takes 178 (tools/trace/progressive_ablation.sh, PIPELINED=1).  A follower only
    // needs whole MCU rows (480 luma blocks in a 4K frame), so it loses nothing but a row of lag.
    // The REFINEMENT scans keep their 32 blocks: they are one publisher in ten and cost nothing measurable (178.7 vs 178.4 ms).
    // (A cost setting only.  For a while 32 looked load-bearing -- with 256 the forced oversubscribed launch decoded a few frames
    // per thousand wrongly -- but that was an AC first scan announcing blocks of an end-of-band run without following its own
    // producer, see the loop below and DESIGN.md "A scan that skipped its producer"; tools/trace/race_probe.sh is clean at any
    // cadence since.)
#ifndef JPGPU_PS_PUBLISH_EVERY
#define JPGPU_PS_PUBLISH_EVERY 512
#endif
#ifndef JPGPU_PS_PUBLISH_REFINE
#define JPGPU_PS_PUBLISH_REFINE 32
#endif
    constexpr uint32_t kPsPublishEvery = JPGPU_PS_PUBLISH_EVERY;  // (a power of two)
    // the scans at the end of the dependency chains are the long poles (the last refinement carries most of the bits):
    // they win the issue arbitration against the scans sharing their SIMD
    if (pipelined != 0) {
        if (dep_scan[0] != kNoDep && !publishes) __builtin_amdgcn_s_setprio(3);
        else if (dep_scan[0] != kNoDep) __builtin_amdgcn_s_setprio(2);
    }

    if (ncomp != 1 || ss == 0) {
        // ---- DC scans (:92-168, ReadBlockProgressiveDC :227-253): interleaved, or one component.  A different table per
        // component: the window holds the stream bits only, codes are looked up one at a time.
        ProgComp pc[kMaxScanComponents];
        uint32_t dc_slot[kMaxScanComponents];
#pragma unroll
        for (uint32_t c = 0; c < kMaxScanComponents; c++) {
            pc[c] = prog_comp(s, c < ncomp ? c : 0);
            dc_slot[c] = s.comp[c < ncomp ? c : 0].dc_slot;
        }
        int32_t pred[kMaxScanComponents] = {0, 0, 0, 0};
        uint32_t u0 = 0;
        // A scan header may name one frame component twice (InitDecodeComponents keeps both entries, each with its own
        // predictor): the two blocks of a unit are then the SAME block of the store, written one after the other.  The passes
        // below take 64 blocks at once -- two lanes with one address in one store, or one read-modify-write -- so such a scan
        // goes block by block (tests/golden/stress/progressive_duplicate_component_*.jpg: corrupted headers do this)
        bool aliased = false;
#pragma unroll
        for (uint32_t c = 0; c < kMaxScanComponents; c++)
#pragma unroll
            for (uint32_t c2 = c + 1; c2 < kMaxScanComponents; c2++)
                if (c2 < ncomp && s.comp[c].component_index == s.comp[c2].component_index) aliased = true;
        if (ah != 0) {
            // DC refinement (ReadBlockProgressiveDC, the Ah != 0 arm, :240-252): ONE bit per block and nothing else in the
            // stream, so block n of the interval owns stream bit n -- no serial parse at all.  64 blocks per pass, one per lane:
            // the lane's bit is the top bit of its window word.  Whole units per pass (the serial loop below finishes the
            // interval's tail, where the "bits available" rules may matter).
            uint32_t bpu = 0, cbase[kMaxScanComponents + 1];
#pragma unroll
            for (uint32_t c = 0; c < kMaxScanComponents; c++) {
                cbase[c] = bpu;
                if (c < ncomp) bpu += ncomp == 1 ? 1u : pc[c].h * pc[c].v;
            }
            cbase[kMaxScanComponents] = bpu;
            const uint32_t group = !aliased && bpu != 0 && bpu <= 64u ? 64u / bpu : 0u;
            uint32_t passes = 0;
            while (group != 0 && u0 + group <= my_units && err == 0 && d.rem >= 128) {
                JPGPU_ENSURE_STAGED()
                if (rows_ready != 0xFFFFFFFFu) JPGPU_FOLLOW((first_unit + u0 + group - 1u) / my_units_per_row)
                if (err != 0) break;
                if ((passes++ & (kPsPublishEvery / 8u - 1u)) == kPsPublishEvery / 8u - 1u) JPGPU_PUBLISH(u0)
                d.cur = 64;
                w_refresh<false>(d, lane, lds_huff(tabs, dc_slot[0]));
                const uint32_t nb = group * bpu;
                if (lane < nb && (d.peek >> 31) != 0) {
                    const uint32_t unit = first_unit + u0 + lane / bpu, within = lane - (lane / bpu) * bpu;
                    uint32_t c = 0;
#pragma unroll
                    for (uint32_t q = 1; q < kMaxScanComponents; q++)
                        if (q < ncomp && within >= cbase[q]) c = q;
                    ProgComp p = pc[0];
#pragma unroll
                    for (uint32_t q = 1; q < kMaxScanComponents; q++)
                        if (c == q) p = pc[q];
                    const uint32_t r = within - (c == 0 ? 0u : (c == 1 ? cbase[1] : (c == 2 ? cbase[2] : cbase[3])));
                    const uint32_t vy = unit / units_per_line, vx = unit - vy * units_per_line;
                    uint64_t index = 0;
                    const bool real = ncomp == 1 ? prog_block_index(fr, p, vx, vy, index)
                                                 : prog_block_index(fr, p, vx * p.h + r % p.h, vy * p.v + r / p.h, index);
                    if (real) dc_refine_or(coefs + index * 64, al);
                }
                d.pos += nb;
                d.cur = 64;
                d.rem -= (int32_t)nb;
                u0 += group;
            }
        }
        if (ah == 0) {
            // DC first pass (ReadBlockProgressiveDC, :232-243): ONE symbol per block -- the category, then that many magnitude
            // bits -- so the only serial thing is the position.  64 blocks (whole MCUs) per pass: a hand-written loop walks the
            // window entries (~25 instructions per block: pre-digested entry of the block's DC table, advance, note in lane n
            // where block n's magnitude ends), then all lanes at once cut their magnitudes out of the ring, extend them, run
            // one prefix sum per component for the predictors and store.  Up to two distinct DC tables per scan; the tail of
            // the interval (and anything unusual: a code longer than the lookup, a window rebuild) goes block by block below.
            uint32_t bpu = 0, cbase[kMaxScanComponents + 1];
#pragma unroll
            for (uint32_t c = 0; c < kMaxScanComponents; c++) {
                cbase[c] = bpu;
                if (c < ncomp) bpu += ncomp == 1 ? 1u : pc[c].h * pc[c].v;
            }
            cbase[kMaxScanComponents] = bpu;
            const uint32_t slot_a = dc_slot[0];
            uint32_t slot_b = slot_a;
            bool two_tables = true;
#pragma unroll
            for (uint32_t c = 1; c < kMaxScanComponents; c++)
                if (c < ncomp && dc_slot[c] != slot_a) {
                    if (slot_b == slot_a) slot_b = dc_slot[c];
                    else if (dc_slot[c] != slot_b) two_tables = false;
                }
            const uint32_t group = !aliased && two_tables && bpu != 0 && bpu <= 64u ? 64u / bpu : 0u;
            const uint32_t nb = group * bpu;
            // component of the block this lane stands for in a pass, and its place inside the MCU
            const uint32_t within = bpu != 0 ? lane - (lane / bpu) * bpu : 0u;
            uint32_t my_c = 0;
#pragma unroll
            for (uint32_t q = 1; q < kMaxScanComponents; q++)
                if (q < ncomp && within >= cbase[q]) my_c = q;
            ProgComp my_p = pc[0];
            uint32_t my_slot = dc_slot[0];
#pragma unroll
            for (uint32_t q = 1; q < kMaxScanComponents; q++)
                if (my_c == q) {
                    my_p = pc[q];
                    my_slot = dc_slot[q];
                }
            const uint32_t my_r = within - (my_c == 0 ? 0u : (my_c == 1 ? cbase[1] : (my_c == 2 ? cbase[2] : cbase[3])));
            const uint64_t tabmask = __ballot(lane < nb && my_slot != slot_a);  // blocks of a pass that decode with the second table
            const LdsHuff ha = lds_huff(tabs, slot_a), hb = lds_huff(tabs, slot_b);
            const uint32_t ringbits = uni((d.wmask + 1u) * 32u - 1u);
            uint32_t ent_a = 0, ent_b = 0;  // the window entries, pre-digested: special << 31 | category << 6 | code + magnitude bits
            auto digest = [](uint32_t e) {
                const uint32_t size = e >> 8, cat = e & 0xFFu;
                return ((size == 0 || cat > 16u) ? 0x80000000u : 0u) | ((cat & 31u) << 6) | ((size + cat) & 63u);
            };
            uint32_t passes = 0;
            while (group != 0 && u0 + group <= my_units && err == 0 && d.rem >= kPsFastBits) {
                JPGPU_ENSURE_STAGED()
                if (rows_ready != 0xFFFFFFFFu) JPGPU_FOLLOW((first_unit + u0 + group - 1u) / my_units_per_row)
                if (err != 0) break;
                if ((passes++ & (kPsPublishEvery / 8u - 1u)) == kPsPublishEvery / 8u - 1u) JPGPU_PUBLISH(u0)
                const uint32_t pos0 = uni(d.pos);
                uint32_t cur = uni(d.cur), winpos = pos0 - cur;
                uint32_t rec = 0;   // lane n: (category << 6 | code + magnitude bits) << 16 | ring position behind block n's magnitude
                uint32_t nblk = 0;  // blocks of the pass parsed so far
                for (;;) {
                    uint32_t sblk = uni(nblk), stop_, scur, sa, sb, ve, vb, adv, curn, u, pp, val, nbn;
                    uint64_t sel, sp;
                    cur = uni(cur);
                    asm volatile(
                        "1:\n\t"
                        "v_readfirstlane_b32 %[scur], %[cur]\n\t"
                        "s_bitcmp1_b64 %[tabmask], %[sblk]\n\t"
                        "s_cselect_b64 %[sel], -1, 0\n\t"
                        "s_add_u32 %[sblk], %[sblk], 1\n\t"
                        "s_nop 0\n\t"
                        "v_readlane_b32 %[sa], %[enta], %[scur]\n\t"
                        "v_readlane_b32 %[sb], %[entb], %[scur]\n\t"
                        "v_add_u32_e32 %[nbn], 1, %[nblk]\n\t"
                        "v_mov_b32_e32 %[ve], %[sa]\n\t"
                        "v_cmp_eq_u32_e64 %[sp], %[lane], %[nblk]\n\t"
                        "v_mov_b32_e32 %[vb], %[sb]\n\t"
                        "v_cndmask_b32_e64 %[ve], %[ve], %[vb], %[sel]\n\t"
                        "v_and_b32_e32 %[adv], 63, %[ve]\n\t"
                        "v_add_u32_e32 %[curn], %[cur], %[adv]\n\t"
                        "v_and_b32_e32 %[u], 0x80000000, %[ve]\n\t"
                        "v_or_b32_e32 %[u], %[u], %[cur]\n\t"
                        "v_cmp_gt_u32_e32 vcc, 64, %[u]\n\t"
                        "v_add_u32_e32 %[pp], %[winposv], %[curn]\n\t"
                        "v_and_b32_e32 %[pp], %[ringv], %[pp]\n\t"
                        "v_lshl_or_b32 %[val], %[ve], 16, %[pp]\n\t"
                        "v_cndmask_b32_e32 %[cur], %[cur], %[curn], vcc\n\t"
                        "v_cndmask_b32_e64 %[rec], %[rec], %[val], %[sp]\n\t"
                        "v_cndmask_b32_e32 %[nblk], %[nblk], %[nbn], vcc\n\t"
                        "v_cndmask_b32_e32 %[stop], %[none], %[nbn], vcc\n\t"
                        "s_nop 0\n\t"
                        "v_cmp_gt_u32_e32 vcc, %[nb], %[stop]\n\t"
                        "s_nop 1\n\t"
                        "s_cbranch_vccnz 1b\n\t"
                        : [cur] "+v"(cur), [nblk] "+v"(nblk), [rec] "+v"(rec), [sblk] "+s"(sblk), [stop] "=&v"(stop_), [scur] "=&s"(scur),
                          [sa] "=&s"(sa), [sb] "=&s"(sb), [sel] "=&s"(sel), [sp] "=&s"(sp), [ve] "=&v"(ve), [vb] "=&v"(vb), [adv] "=&v"(adv),
                          [curn] "=&v"(curn), [u] "=&v"(u), [pp] "=&v"(pp), [val] "=&v"(val), [nbn] "=&v"(nbn)
                        : [enta] "v"(ent_a), [entb] "v"(ent_b), [lane] "v"(lane), [winposv] "v"(winpos), [ringv] "v"(ringbits),
                          [none] "v"(0xFFFFu), [tabmask] "s"(tabmask), [nb] "s"(nb)
                        : "vcc", "scc", "memory");
                    cur = uni(cur);
                    nblk = uni(nblk);
                    if (uni(stop_) != 0xFFFFu) break;  // the pass is complete
                    if (cur > 63u) {
                        d.pos = winpos + cur;
                        d.cur = 64;
                        w_refresh<false>(d, lane, ha);
                        ent_a = digest(ha.lut[d.peek >> (32 - kHuffLutBits)]);
                        ent_b = digest(hb.lut[d.peek >> (32 - kHuffLutBits)]);
                        winpos = uni(d.pos);
                        cur = 0;
                        continue;
                    }
                    // one block by hand: a code longer than the lookup (or no code at all)
                    const bool use_b = ((tabmask >> nblk) & 1ull) != 0;
                    const uint32_t pk = lane_get(d.peek, cur);
                    const uint32_t raw = w_huff_scalar(use_b ? hb : ha, pk >> 16);
                    const uint32_t size = raw >> 8, cat = raw & 0xFFu;
                    if (size > 16u || cat > 16u) {
                        err = kDetailInvalidHuffmanCode;
                        break;
                    }
                    cur += size + cat;
                    if (lane == nblk) rec = ((((cat & 31u) << 6) | ((size + cat) & 63u)) << 16) | ((winpos + cur) & ringbits);
                    nblk++;
                    if (nblk >= nb) break;
                }
                if (err != 0) {
                    // the failing block is somewhere in this pass: let the block-by-block loop find it from the pass's first unit
                    d.pos = pos0;
                    d.cur = 64;
                    err = 0;
                    break;
                }
                // all lanes: magnitude, ReceiveAndExtend (:100-115), predictor prefix per component, store
                const uint32_t cat = (rec >> 22) & 31u;
                const uint32_t mp = ((rec & 0xFFFFu) - cat) & ringbits;
                const uint32_t w0 = d.ring[(mp >> 5) & d.wmask], w1 = d.ring[((mp >> 5) + 1u) & d.wmask];
                const uint32_t top = (uint32_t)(((((uint64_t)w0) << 32) | w1) >> (32u - (mp & 31u)));
                const int32_t v = cat != 0 ? (int32_t)(top >> (32u - cat)) : 0;
                const int32_t diff = lane < nb && cat != 0 ? v - ((((v + v) >> cat) - 1) & ((1 << cat) - 1)) : 0;
                int32_t value = 0;
#pragma unroll
                for (uint32_t q = 0; q < kMaxScanComponents; q++) {
                    if (q >= ncomp) continue;
                    const bool in_q = lane < nb && my_c == q;
                    const uint32_t incl = wave_inclusive_scan(in_q ? (uint32_t)diff : 0u);
                    if (in_q) value = pred[q] + (int32_t)incl;
                    pred[q] += (int32_t)lane_get(incl, 63);
                }
                if (lane < nb) {
                    const uint32_t unit = first_unit + u0 + lane / bpu;
                    const uint32_t vy = unit / units_per_line, vx = unit - vy * units_per_line;
                    uint64_t index = 0;
                    const bool real = ncomp == 1 ? prog_block_index(fr, my_p, vx, vy, index)
                                                 : prog_block_index(fr, my_p, vx * my_p.h + my_r % my_p.h, vy * my_p.v + my_r / my_p.h, index);
                    if (real) coefs[index * 64] = (int16_t)((uint32_t)value << al);
                }
                d.pos = winpos + cur;
                d.cur = cur;  // the window and its pre-digested entries stay valid for the next pass
                d.rem -= (int32_t)(d.pos - pos0);
                u0 += group;
            }
        }
        uint32_t uy = (first_unit + u0) / units_per_line, ux = (first_unit + u0) - uy * units_per_line;  // interleaved: MCU; else block
        for (uint32_t u = u0; u < my_units && err == 0; u++) {
            JPGPU_ENSURE_STAGED()
            if (rows_ready != 0xFFFFFFFFu) JPGPU_FOLLOW((first_unit + u) / my_units_per_row)
            if ((u & (kPsPublishEvery - 1u)) == 0 && u != 0) JPGPU_PUBLISH(u)
#pragma unroll
            for (uint32_t c = 0; c < kMaxScanComponents; c++) {
                if (c >= ncomp || err != 0) continue;
                const ProgComp p = pc[c];
                const LdsHuff hdc = lds_huff(tabs, dc_slot[c]);
                const uint32_t nh = ncomp == 1 ? 1u : p.h, nv = ncomp == 1 ? 1u : p.v;
                for (uint32_t y = 0; y < nv && err == 0; y++)
                    for (uint32_t x = 0; x < nh; x++) {
                        uint64_t index = 0;
                        const bool real = ncomp == 1 ? prog_block_index(fr, p, ux, uy, index)
                                                     : prog_block_index(fr, p, ux * p.h + x, uy * p.v + y, index);
                        if (ah == 0) {
                            uint32_t sym;
                            int32_t value;
                            err = w_symbol<false, false>(d, lane, hdc, true, closed_by_marker, sym, value);
                            if (err != 0) break;
                            const int32_t t = pred[c] + value;
                            pred[c] = t;
                            if (real && lane == 0) coefs[index * 64] = (int16_t)((uint32_t)t << al);
                        } else {
                            uint32_t bit;
                            if (!w_read_bits<false, false>(d, lane, hdc, 1, bit)) {
                                err = kDetailUnexpectedEnd;
                                break;
                            }
                            if (real && bit != 0 && lane == 0) dc_refine_or(coefs + index * 64, al);
                        }
                    }
            }
            if (++ux == units_per_line) {
                ux = 0;
                uy++;
            }
        }
    } else {
        // ---- AC scans of one component: first pass (:255-311) and refinement (:313-419)
        const ProgComp p = prog_comp(s, 0);
        const LdsHuff hac = lds_huff(tabs, s.comp[0].ac_slot);
        const int32_t p1 = (int32_t)(int16_t)(1u << al), m1 = (int32_t)(int16_t)(0xFFFFFFFFu << al);
        const uint64_t band = (se >= 63u ? ~0ull : ((1ull << (se + 1u)) - 1ull)) & ~((1ull << ss) - 1ull);
        const uint64_t lane_bit = 1ull << lane;
        uint32_t eobrun = 0;
        if (ah == 0) {
            ProgWalk w;
            prog_walk_init(w, p, first_unit, units_per_line);
            for (uint32_t u = 0; u < my_units && err == 0;) {
                // (with the write-back also where nothing was stored since the last word, inside an end-of-band run: leaving it out
                // there is legal -- the producers' stores were written back before THEIR words -- and measured slower, 186 vs 181 ms
                // per 256 frames)
                if ((u & (kPsPublishEvery - 1u)) == 0 && u != 0) JPGPU_PUBLISH(u)
                // BEFORE the end-of-band skip: a scan must not announce units its own producers have not reached.  The host
                // drops a dependency that another one implies (the Y refinement follows Y AC 6-63 only, which follows Y AC 1-5);
                // a first scan that is one long end-of-band run -- 15 bytes for a whole 4K frame -- used to skip its blocks
                // without looking at its producer, announced them, and the refinement behind it went ahead of Y AC 1-5
                // whenever that scan was the slower one: the last hundred frames of the forced, oversubscribed launch, where
                // the scans of a frame start in any order ("invalid Huffman code", round 3's long hunt in DESIGN.md).
                JPGPU_FOLLOW(w.my)
                if (err != 0) break;  // gave up waiting (kDetailSpinTimeout)
                if (eobrun != 0) {
                    // blocks inside an end-of-band run are not touched: to the end of the block row in one step (the next row is
                    // followed and announced like any other), never across a progress word.  (Round 3 walked them one by one:
                    // the Y AC 6-63 first scan of a smooth 4K frame is ONE run of 129 600 blocks, 17.8 ms per 256 frames.)
                    uint32_t skip = eobrun;
                    const uint32_t row_left = units_per_line - w.bx, pub_left = kPsPublishEvery - (u & (kPsPublishEvery - 1u)), left = my_units - u;
                    skip = skip < row_left ? skip : row_left;
                    skip = skip < pub_left ? skip : pub_left;
                    skip = skip < left ? skip : left;
                    eobrun -= skip;
                    u += skip;
                    if (skip == 1) prog_walk_next(w, p, units_per_line);
                    else prog_walk_init(w, p, first_unit + u, units_per_line);
                    continue;
                }
                JPGPU_ENSURE_STAGED()
                uint64_t index = 0;
                const bool real = prog_walk_index(fr, p, w, index);
                int32_t c = 0;
                uint64_t changed = 0;
                if (d.rem >= kPsFastBits) {
                    const uint32_t pos0 = d.pos;
#ifdef JPGPU_PS_OLD_REFINE
                    err = w_ac_first_fast(d, lane, hac, ss, se, al, eobrun, c, changed);
#else
                    err = w_ac_first_parse(d, lane, hac, ss, se, al, eobrun, real ? coefs + index * 64 : nullptr);
#endif
                    d.rem -= (int32_t)(d.pos - pos0);
                } else {
                    err = w_ac_first_block<false>(d, lane, hac, closed_by_marker, ss, se, al, eobrun, c, changed);
                    JPGPU_SETTLE()
                }
                if (real && (changed & lane_bit) != 0) coefs[index * 64 + lane] = (int16_t)c;
                u++;
                prog_walk_next(w, p, units_per_line);
            }
        } else {
            unsigned long long ps_stage = 0, ps_wait = 0, ps_blocks = 0, ps_t0 = PS_TICK();
            const R5Consts k5 = r5_consts(lane, band, p1, m1);
            uint32_t eobv = 0;  // the end-of-band run, in a vector register (the same in all lanes)
            // The next chunk's blocks are fetched while this one is parsed (two 16-byte pieces per lane in registers; their block
            // indices in the other half of idx[]) -- when the rows they lie in are already known to be complete.
            uint4 pre0 = make_uint4(0, 0, 0, 0), pre1 = pre0;
            bool have_pre = false;
            uint32_t ihalf = 0;  // which half of idx[] holds the current chunk's indices
            auto chunk_indices = [&](uint32_t first, uint32_t n_, uint32_t *dst) {
                if (lane < n_) {
                    ProgWalk w;
                    prog_walk_init(w, p, first_unit + first + lane, units_per_line);
                    uint64_t index = 0;
                    const bool real = prog_walk_index(fr, p, w, index);
                    dst[lane] = real ? (uint32_t)index : kPsNoBlock;
                }
            };
            auto piece = [&](uint32_t q, const uint32_t *ix_) {
                const uint32_t b = q >> 3, part = q & 7u, ix = ix_[b];
                return *reinterpret_cast<const uint4 *>(coefs + (ix != kPsNoBlock ? (uint64_t)ix : fr.coef_off) * 64 + part * 8u);
            };
            for (uint32_t done = 0; done < my_units && err == 0;) {
                const uint32_t n = my_units - done < (uint32_t)kPsChunk ? my_units - done : (uint32_t)kPsChunk;
                const unsigned long long ps_a = PS_TICK();
                if (done != 0 && (done & (JPGPU_PS_PUBLISH_REFINE - 1u)) == 0) JPGPU_PUBLISH(done)  // (a release fence costs microseconds)
                JPGPU_FOLLOW((first_unit + done + n - 1u) / my_units_per_row)
                if (err != 0) break;  // gave up waiting (kDetailSpinTimeout)
                const unsigned long long ps_b = PS_TICK();
                ps_wait += ps_b - ps_a;
                uint32_t *idx_cur = idx + ihalf * kPsChunk;
                if (have_pre) {
                    if (lane < n * 8u) reinterpret_cast<uint4 *>(stage)[lane] = pre0;
                    if (lane + 64u < n * 8u) reinterpret_cast<uint4 *>(stage)[lane + 64u] = pre1;
                } else {
                    chunk_indices(done, n, idx_cur);
                    __syncthreads();
                    for (uint32_t q = lane; q < n * 8u; q += 64u) reinterpret_cast<uint4 *>(stage)[q] = piece(q, idx_cur);
                }
                __syncthreads();
                have_pre = false;
                if (kPsChunk <= 16u && done + n < my_units) {
                    const uint32_t first2 = done + n, n2 = my_units - first2 < (uint32_t)kPsChunk ? my_units - first2 : (uint32_t)kPsChunk;
                    if ((first_unit + first2 + n2 - 1u) / my_units_per_row < rows_ready) {
                        uint32_t *idx_next = idx + (ihalf ^ 1u) * kPsChunk;
                        chunk_indices(first2, n2, idx_next);
                        __syncthreads();
                        if (lane < n2 * 8u) pre0 = piece(lane, idx_next);
                        if (lane + 64u < n2 * 8u) pre1 = piece(lane + 64u, idx_next);
                        have_pre = true;
                    }
                }
                ihalf ^= have_pre ? 1u : 0u;  // (the next chunk reads the half just filled; else it refills this one)
                const unsigned long long ps_c = PS_TICK();
                ps_stage += ps_c - ps_b;
                int32_t c_next = stage[lane];
                uint32_t ix_next = idx_cur[0];
                for (uint32_t b = 0; b < n && err == 0; b++) {
                    JPGPU_ENSURE_STAGED()
                    int32_t c = c_next;
                    const uint32_t ix = uni(ix_next);
                    {  // the next block's coefficients are on their way while this one is parsed
                        const uint32_t bn = b + 1u < n ? b + 1u : b;
                        c_next = stage[bn * 64u + lane];
                        ix_next = idx_cur[bn];
                    }
                    bool mine = false;  // this lane's coefficient changed
#ifdef JPGPU_PS_TRACE
                    // (as little as possible: the failures are shy -- a hash of the 64 values in front of every block made them go away)
                    const uint64_t trace_nz = __ballot(c != 0);
                    const uint32_t trace_pos = d.pos;
#endif
                    if (d.rem >= kPsFastBits) {
                        const uint32_t pos0 = d.pos;
#if defined(JPGPU_PS_OLD_REFINE) || defined(JPGPU_PS_REFINE3) || defined(JPGPU_PS_REFINE4)
                        const uint64_t nz = __ballot(c != 0);
                        uint32_t eobrun = uni(eobv);
#ifdef JPGPU_PS_OLD_REFINE
                        err = w_ac_refine_fast(d, lane, hac, ss, se, p1, m1, band, nz, eobrun, c, mine);
#elif defined(JPGPU_PS_REFINE3)
                        err = w_ac_refine_parse(d, lane, hac, ss, se, p1, m1, band, nz, eobrun, c, mine,
                                                ix != kPsNoBlock ? coefs + (uint64_t)ix * 64 : nullptr);
#else
                        err = w_ac_refine_v4(d, lane, hac, ss, se, p1, m1, band, nz, eobrun, c, mine);
#endif
                        eobv = eobrun;
#else
#ifdef JPGPU_PS_CHECK
                        {
                            const uint64_t nz = __ballot(c != 0);
                            WBits d3 = d;
                            d3.ent2 = r2_digest(d3.ent);  // (the window entries as the third form wants them)
                            uint32_t eob3 = uni(eobv);
                            int32_t c3 = c;
                            bool mine3 = false;
                            const uint32_t e3 = w_ac_refine_parse(d3, lane, hac, ss, se, p1, m1, band, nz, eob3, c3, mine3, nullptr);
                            WBits d4 = d;
                            uint32_t eob4 = eobv;
                            int32_t c4 = c;
                            bool mine4 = false;
                            const uint32_t e4 = w_ac_refine_v5(d4, k5, hac, ss, se, p1, m1, eob4, c4, mine4);
                            const bool bad = e3 != e4 || d3.pos != d4.pos || eob3 != uni(eob4);
                            const uint64_t cbad = __ballot(((nz >> lane) & 1ull) != 0 && c3 != c4);
                            if ((bad || cbad != 0) && lane == 0)
                                printf("refine mismatch: block %u pos0 %u cur0 %u eobrun0 %u nz %llx band %llx | v3 err %u pos %u eob %u | v5 err %u pos %u eob %u | corr lanes %llx\n",
                                       done, d.pos, d.cur, uni(eobv), (unsigned long long)nz, (unsigned long long)band, e3, d3.pos, eob3, e4, d4.pos,
                                       uni(eob4), (unsigned long long)cbad);
                        }
#endif
                        err = w_ac_refine_v5(d, k5, hac, ss, se, p1, m1, eobv, c, mine);
#endif
                        d.rem -= (int32_t)(d.pos - pos0);
                    } else {
                        const uint64_t nz = __ballot(c != 0);
                        uint32_t eobrun = uni(eobv);
                        err = w_ac_refine_block<false>(d, lane, hac, ss, se, p1, m1, band, nz, eobrun, c, mine);
                        JPGPU_SETTLE()
                        eobv = eobrun;
                    }
#ifdef JPGPU_PS_TRACE
                    if (ps_trace_buf != nullptr && s.comp[0].component_index == 0 && s.image_index >= ps_trace_first_image &&
                        s.image_index - ps_trace_first_image < ps_trace_images && done < ps_trace_units) {
                        uint32_t *t = ps_trace_buf + (((uint64_t)(s.image_index - ps_trace_first_image) * 2u + (al == 0 ? 1u : 0u)) * ps_trace_units + done) * 4u;
                        if (lane == 0) {
                            t[0] = (uint32_t)trace_nz;
                            t[1] = (uint32_t)(trace_nz >> 32);
                            t[2] = trace_pos;
                            t[3] = (uni(eobv) << 16) | (d.pos - trace_pos);
                        }
                    }
#endif
                    if (ix != kPsNoBlock && mine) coefs[(uint64_t)ix * 64 + lane] = (int16_t)c;
                    if (err == 0) done++;
                }
                ps_blocks += PS_TICK() - ps_c;
            }
            if (ah == 1 && al == 0) {  // the last luma / chroma refinements
                PS_ADD(0, 1);
                PS_ADD(1, my_units);
                PS_ADD(2, ps_wait);
                PS_ADD(3, ps_stage);
                PS_ADD(4, ps_blocks);
                PS_ADD(5, PS_TICK() - ps_t0);
#ifdef JPGPU_PS_PROFILE
                PS_ADD(8, d.t_pro);
                PS_ADD(9, d.t_loop);
                PS_ADD(10, d.t_epi);
                PS_ADD(11, d.t_refresh);
                PS_ADD(12, d.n_exits);
                PS_ADD(13, d.n_trips);
#endif
            }
        }
    }
    JPGPU_PUBLISH(0xFFFFFFFFu)  // finished (or failed: followers must not wait for units that will never come)
#undef JPGPU_TOP_UP
#undef JPGPU_ENSURE_STAGED
#undef JPGPU_SETTLE
#undef JPGPU_FOLLOW
#undef JPGPU_DEBUG_DELAY
#undef JPGPU_PUBLISH

    if (lane == 0) {
        const uint32_t code = restart_check(s, st, &status[wk.scan], interval, n_ends, n_intervals, dri_eff, d.rem, err);
        if (code != kNoError) atomicMin(&status[wk.scan].first_error, code);
    }
}

__global__ __launch_bounds__(64) void progressive_stream_kernel(const uint8_t *__restrict__ udata, const DevScan *__restrict__ scans,
                                                                const HuffWork *__restrict__ work, const uint32_t *__restrict__ ends_u,
                                                                DevScanStatus *__restrict__ status,
                                                                const DevHuffTable *__restrict__ huff_pool, int16_t *__restrict__ coefs,
                                                                int n_slots, int pipelined, uint32_t spin_budget, uint32_t ring_bytes,
                                                                uint32_t chunk_blocks, uint32_t *__restrict__ started) {
    HuffWork wk = work[blockIdx.x];
    bool first_in_wave = true;
    for (;;) {
        ps_run_scan(udata, scans, wk, ends_u, status, huff_pool, coefs, n_slots, pipelined, spin_budget, ring_bytes, chunk_blocks, started,
                    first_in_wave);
        const uint32_t next = pipelined != 0 ? uni(scans[wk.scan].wave_next) : 0u;
        if (next == 0) break;
        wk.scan += next;  // (the launch's list holds one-interval scans only: first_interval stays 0)
        first_in_wave = false;
    }
}



// One ordinal of progressive scans (all frames of the batch advance together).
hipError_t launch_progressive(hipStream_t stream, const uint8_t *udata, const DevScan *scans, const HuffWork *work, int n_work,
                              const uint32_t *ends_u, DevScanStatus *status, const DevHuffTable *huff_pool, int16_t *coefs, int n_slots) {
    if (n_work <= 0) return hipSuccess;
    const size_t lds = (size_t)n_slots * sizeof(DevHuffTable) + (size_t)kProgThreads * kProgBlockStride;
    hipLaunchKernelGGL(progressive_scan_kernel, dim3(n_work), dim3(kProgThreads), lds, stream, udata, scans, work, ends_u, status,
                       huff_pool, coefs, n_slots);
    return hipGetLastError();
}

// LDS one stream workgroup (one wave) takes: the host sizes pipelined launches by it (every workgroup must be resident).
static void ps_lds_shape(uint32_t &ring, uint32_t &chunk) {
    static const uint32_t ring_ = [] {
        const char *ev = getenv("JPGPU_PS_RING");
        const uint32_t v = ev ? (uint32_t)atoi(ev) : 2048u;
        return v >= 4096u ? 4096u : 2048u;
    }();
    static const uint32_t chunk_ = [] {
        const char *ev = getenv("JPGPU_PS_CHUNK");
        const uint32_t v = ev ? (uint32_t)atoi(ev) : 16u;
        return v >= 32u ? 32u : (v >= 16u ? 16u : 8u);
    }();
    ring = ring_;
    chunk = chunk_;
}
size_t progressive_stream_lds_bytes(int n_slots) {
    uint32_t ring, chunk;
    ps_lds_shape(ring, chunk);
    return (size_t)n_slots * sizeof(DevHuffTable) + ring + (size_t)chunk * 128 + chunk * 8;  // (idx[]: two halves)
}

// Stream workgroups (one wave each) a CU really holds at once: LDS AND registers (the residency gate of the
// pipelined launch counted LDS alone; a few more VGPRs would have made "resident" grids non-resident).  0 = unknown.
int progressive_stream_blocks_per_cu(int n_slots) {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, progressive_stream_kernel, 64, progressive_stream_lds_bytes(n_slots)) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

// The same, one wave per (scan, restart interval): for scans with few, long intervals.
hipError_t launch_progressive_streams(hipStream_t stream, const uint8_t *udata, const DevScan *scans, const HuffWork *work, int n_work,
                                      const uint32_t *ends_u, DevScanStatus *status, const DevHuffTable *huff_pool, int16_t *coefs,
                                      int n_slots, int pipelined, uint32_t spin_budget, uint32_t *started) {
    if (n_work <= 0) return hipSuccess;
    uint32_t ring, chunk;
    ps_lds_shape(ring, chunk);
    const size_t lds = progressive_stream_lds_bytes(n_slots);
    hipLaunchKernelGGL(progressive_stream_kernel, dim3(n_work), dim3(64), lds, stream, udata, scans, work, ends_u, status, huff_pool,
                       coefs, n_slots, pipelined, spin_budget, ring, chunk, started);
    return hipGetLastError();
}

}  // namespace jpgpu
