// jpeglibrary_amd/csrc/kt_transcode.hip -- KT / KTS: symbol-level Huffman transcode of baseline scans (JpegOptimizer)
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
// KT: symbol-level Huffman transcode of baseline scans (ref: JpegOptimizer.cs:360-516 ProcessScanBaseline /
// ProcessBlockBaseline, :719-880 CopyScanBaseline / CopyBlockBaseline; JpegWriter.cs:93-227).
//
// The optimizer never reconstructs coefficients: it re-reads the scan symbol by symbol and re-writes every symbol with a
// table built from the scan's own statistics, copying the magnitude bits.  One lane per restart interval, lanes run
// free; the same walk runs three times:
//   MODE 0  count    -- IncrementCodeCount per (table, symbol): LDS histograms per workgroup, merged into the scan's
//                       8 x 256 counters in HBM (the host then runs JpegHuffmanEncodingTableBuilder.Build per table);
//   MODE 1  measure  -- the exact number of bytes the interval will occupy in the output: the writer is simulated in
//                       registers (byte stuffing FF -> FF 00 and the all-ones padding of ExitBitMode, :141-166), + 2
//                       for the RSTn the reference re-emits after the interval (:805-807);
//   MODE 2  emit     -- the same walk writing the bytes at the interval's offset (exclusive scan of MODE 1's sizes).
// Errors are the reference's: invalid code, Receive() hitting a marker / the end of the data (:505-517), a restart check
// that does not find RSTn / EOI (:796-803).
// ------------------------------------------------------------------------------------------------
constexpr int kTcThreads = 256;

struct TcWriter {
    uint64_t acc;    // pending bits, right-aligned
    uint32_t nbits;  // < 32 between puts
    uint32_t bytes;  // bytes produced so far (stuffing included)
};

template <int MODE>
__device__ __forceinline__ void tc_flush_bytes(TcWriter &w, uint32_t n, uint8_t *out) {
    // the n oldest whole bytes leave the accumulator
    for (uint32_t k = 0; k < n; k++) {
        const uint32_t b = (uint32_t)(w.acc >> (w.nbits - 8u)) & 0xFFu;
        w.nbits -= 8u;
        if (MODE == 2) {
            out[w.bytes] = (uint8_t)b;
            if (b == 0xFFu) out[w.bytes + 1] = 0;
        }
        w.bytes += b == 0xFFu ? 2u : 1u;
    }
}
template <int MODE>
__device__ __forceinline__ void tc_put(TcWriter &w, uint32_t bits, uint32_t len, uint8_t *out) {  // WriteBits, len <= 16
    w.acc = (w.acc << len) | bits;
    w.nbits += len;
    if (w.nbits >= 32u) tc_flush_bytes<MODE>(w, w.nbits >> 3, out);
}

template <int MODE>
__global__ __launch_bounds__(kTcThreads) void transcode_kernel(const uint8_t *__restrict__ udata, const uint8_t *__restrict__ input,
                                                               const DevScan *__restrict__ scans, const HuffWork *__restrict__ work,
                                                               const uint32_t *__restrict__ ends_u, const uint32_t *__restrict__ ends_raw,
                                                               DevScanStatus *__restrict__ status,
                                                               const DevHuffTable *__restrict__ huff_pool, uint32_t *__restrict__ hist,
                                                               const EncHuffTable *__restrict__ enc, uint32_t *__restrict__ sizes,
                                                               const uint64_t *__restrict__ offsets, uint8_t *__restrict__ out, int n_slots) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t *tabs = smem;  // n_slots * sizeof(DevHuffTable)
    uint32_t *blk_info = reinterpret_cast<uint32_t *>(smem + (size_t)n_slots * sizeof(DevHuffTable));  // [kMaxBlocksPerMcu]
    uint8_t *extra = reinterpret_cast<uint8_t *>(blk_info + kMaxBlocksPerMcu);  // MODE 0: hist[8][256] u32; else EncHuffTable[8]
    uint32_t *lhist = reinterpret_cast<uint32_t *>(extra);
    const EncHuffTable *lenc = reinterpret_cast<const EncHuffTable *>(extra);

    const HuffWork wk = work[blockIdx.x];
    const DevScan &s = scans[wk.scan];
    const uint32_t tid = threadIdx.x;
    for (int slot = 0; slot < kMaxHuffSlots && slot < n_slots; slot++) {
        const uint32_t pi = s.huff_pool[slot];
        if (pi == 0xFFFF) continue;
        const uint4 *src = reinterpret_cast<const uint4 *>(&huff_pool[pi]);
        uint4 *dst = reinterpret_cast<uint4 *>(tabs + slot * sizeof(DevHuffTable));
        for (uint32_t i = tid; i < sizeof(DevHuffTable) / 16; i += kTcThreads) dst[i] = src[i];
    }
    if (MODE == 0) {
        for (uint32_t i = tid; i < kMaxHuffSlots * 256u; i += kTcThreads) lhist[i] = 0;
    } else {
        const uint4 *src = reinterpret_cast<const uint4 *>(enc + (size_t)wk.scan * kMaxHuffSlots);
        uint4 *dst = reinterpret_cast<uint4 *>(extra);
        for (uint32_t i = tid; i < kMaxHuffSlots * sizeof(EncHuffTable) / 16; i += kTcThreads) dst[i] = src[i];
    }
    const uint32_t bpm = s.blocks_per_mcu;
    if (tid < bpm) {
        const DevScanComponent &c = s.comp[s.blk_comp[tid]];
        blk_info[tid] = (uint32_t)c.dc_slot | ((uint32_t)c.ac_slot << 8);
    }
    __syncthreads();

    const DevScanStatus st = status[wk.scan];
    const uint32_t n_ends = st.n_ends;
    const uint32_t n_intervals = s.n_intervals;
    const uint32_t total_mcus = s.total_mcus;
    const uint32_t dri_eff = s.dri ? s.dri : total_mcus;
    const uint32_t interval = wk.first_interval + tid;
    bool active = interval < n_ends && interval < n_intervals;
    // emit: an interval that failed in the measure pass owns no bytes of the output (its size is 0): it must not write
    if (MODE == 2 && active && sizes[s.ends_off + interval] == 0) active = false;
    uint32_t err = 0;
    if (active) {
        const uint32_t *eu = ends_u + s.ends_off;
        const uint32_t ustart = interval == 0 ? 0u : eu[interval - 1] + 2u;
        UBits r;
        ub_init(r, udata + s.data_off, ustart, eu[interval]);
        const bool closed_by_marker = !(interval == n_ends - 1 && st.terminator == 0);
        const uint32_t my_mcus = (interval == n_intervals - 1) ? total_mcus - interval * dri_eff : dri_eff;
        const uint32_t my_blocks = my_mcus * bpm;
        TcWriter w;
        w.acc = 0;
        w.nbits = 0;
        w.bytes = 0;
        uint8_t *dst = MODE == 2 ? out + offsets[s.ends_off + interval] : nullptr;

        uint32_t b_in_mcu = 0;
        for (uint32_t blk = 0; blk < my_blocks && err == 0; blk++) {
            const uint32_t info = blk_info[b_in_mcu];
            b_in_mcu = (b_in_mcu + 1 == bpm) ? 0u : b_in_mcu + 1;
            const uint32_t dc_slot = info & 0xFFu, ac_slot = info >> 8;
            if (dc_slot == 0xFFu) {  // the reference dereferences a table that was never defined
                err = kDetailNullTable;
                break;
            }
            const LdsHuff hdc = lds_huff(tabs, dc_slot), hac = lds_huff(tabs, ac_slot);
            // DC (:470-476 / :839-847)
            uint32_t sym;
            err = ub_huff(r, hdc, sym);
            if (err != 0) break;
            if (MODE == 0) atomicAdd(&lhist[dc_slot * 256u + sym], 1u);
            else tc_put<MODE>(w, lenc[dc_slot].code[sym], lenc[dc_slot].len[sym], dst);
            if (sym != 0) {
                uint32_t bits;
                if (sym > 16u) {  // Receive() of more than 16 bits: outside the verified envelope
                    err = kDetailInvalidHuffmanCode;
                    break;
                }
                if (!ub_try_read_bits(r, sym, bits)) {
                    err = (r.rem == 0 && closed_by_marker) ? kDetailMarkerInData : kDetailStreamEnded;
                    break;
                }
                if (MODE != 0) tc_put<MODE>(w, bits, sym, dst);
            }
            // AC (:478-493 / :849-876)
            if (ac_slot == 0xFFu) {
                err = kDetailNullTable;
                break;
            }
            for (uint32_t i = 1; i < 64u;) {
                err = ub_huff(r, hac, sym);
                if (err != 0) break;
                if (MODE == 0) atomicAdd(&lhist[ac_slot * 256u + sym], 1u);
                else tc_put<MODE>(w, lenc[ac_slot].code[sym], lenc[ac_slot].len[sym], dst);
                const uint32_t rr = sym >> 4, sz = sym & 15u;
                if (sz != 0) {
                    i += rr + 1u;
                    uint32_t bits;
                    if (!ub_try_read_bits(r, sz, bits)) {
                        err = (r.rem == 0 && closed_by_marker) ? kDetailMarkerInData : kDetailStreamEnded;
                        break;
                    }
                    if (MODE != 0) tc_put<MODE>(w, bits, sz, dst);
                } else {
                    if (rr == 0) break;
                    i += 16u;
                }
            }
        }
        if (MODE != 0 && err == 0) {
            // ExitBitMode (:141-166): whole bytes out, then the partial byte padded with ones
            tc_flush_bytes<MODE>(w, w.nbits >> 3, dst);
            if (w.nbits != 0) {
                const uint32_t pad = 8u - w.nbits;
                w.acc = (w.acc << pad) | ((1u << pad) - 1u);
                w.nbits = 8;
                tc_flush_bytes<MODE>(w, 1, dst);
            }
            // the RSTn the reference copies from the input after every interval it continues from (:805-807)
            // ... and an RSTn behind the LAST one (more restart markers than the frame needs): the restart check behind a
            // complete interval copies it like the others (:796-811); behind a partial one the bit reader has pulled it in
            // while refilling unless four or more bytes are still unread, and hands the reader back
            // RemainingBits / 8 bytes in front of the marker's END (:818-831): with two or more unread bytes the outer
            // walk finds the marker again and copies it (:603-612), with none or one it is lost.  RSTn markers further
            // behind are the host walk's (OptimizeBatch::plan_file).
            const bool rst_closes_last = interval == n_intervals - 1 && interval == n_ends - 1 && (st.terminator & 0xF8u) == 0xD0u;
            const bool rst_behind_last = rst_closes_last && (s.restart_check_at_end ? r.rem < 8 : r.rem >= 16);
            const bool marker_follows = (interval + 1 < n_ends && interval + 1 < n_intervals) || rst_behind_last;
            if (marker_follows) {
                if (MODE == 2) {
                    dst[w.bytes] = 0xFF;
                    dst[w.bytes + 1] = input[s.data_off + (ends_raw + s.ends_off)[interval] + 1];
                }
                w.bytes += 2;
            }
            if (MODE == 1) sizes[s.ends_off + interval] = w.bytes;
        }
        const uint32_t code = restart_check(s, st, &status[wk.scan], interval, n_ends, n_intervals, dri_eff, r.rem, err);
        if (code != kNoError) {
            atomicMin(&status[wk.scan].first_error, code);
            if (MODE == 1) sizes[s.ends_off + interval] = 0;
        }
    } else if (MODE == 1 && interval < n_intervals) {
        sizes[s.ends_off + interval] = 0;  // intervals the marker index never found (EOI came early / data ran out)
    }
    if (MODE == 0) {
        __syncthreads();
        uint32_t *gh = hist + (size_t)wk.scan * kMaxHuffSlots * 256u;
        for (uint32_t i = tid; i < kMaxHuffSlots * 256u; i += kTcThreads) {
            const uint32_t v = lhist[i];
            if (v != 0) atomicAdd(&gh[i], v);
        }
    }
}

// Exclusive scan of the interval sizes of every scan: offsets[i] = base[scan] + sum of sizes before i; totals[scan] = sum.
// One workgroup per scan (a 4K DRI = 4 scan has 8 100 intervals).
__global__ __launch_bounds__(1024) void transcode_offsets_kernel(const DevScan *__restrict__ scans, const uint32_t *__restrict__ scan_ids,
                                                                 const uint32_t *__restrict__ sizes, const uint64_t *__restrict__ base,
                                                                 uint64_t *__restrict__ offsets, uint64_t *__restrict__ totals) {
    __shared__ uint64_t part[1024];
    const uint32_t j = scan_ids[blockIdx.x];
    const DevScan &s = scans[j];
    const uint32_t n = s.n_intervals, tid = threadIdx.x;
    const uint32_t per = (n + 1023u) / 1024u;
    const uint32_t lo = tid * per < n ? tid * per : n, hi = lo + per < n ? lo + per : n;
    const uint32_t *sz = sizes + s.ends_off;
    uint64_t sum = 0;
    for (uint32_t i = lo; i < hi; i++) sum += sz[i];
    part[tid] = sum;
    __syncthreads();
    for (uint32_t d = 1; d < 1024u; d <<= 1) {
        const uint64_t v = tid >= d ? part[tid - d] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    if (totals != nullptr && tid == 1023) totals[blockIdx.x] = part[1023];
    if (offsets != nullptr) {
        uint64_t at = base[blockIdx.x] + part[tid] - sum;
        uint64_t *of = offsets + s.ends_off;
        for (uint32_t i = lo; i < hi; i++) {
            of[i] = at;
            at += sz[i];
        }
    }
}

hipError_t launch_transcode(hipStream_t stream, int mode, const uint8_t *udata, const uint8_t *input, const DevScan *scans,
                            const HuffWork *work, int n_work, const uint32_t *ends_u, const uint32_t *ends_raw, DevScanStatus *status,
                            const DevHuffTable *huff_pool, uint32_t *hist, const EncHuffTable *enc, uint32_t *sizes,
                            const uint64_t *offsets, uint8_t *out, int n_slots) {
    if (n_work <= 0) return hipSuccess;
    const size_t lds = (size_t)n_slots * sizeof(DevHuffTable) + kMaxBlocksPerMcu * 4 + (size_t)kMaxHuffSlots * 1024;
    if (mode == 0)
        hipLaunchKernelGGL(transcode_kernel<0>, dim3(n_work), dim3(kTcThreads), lds, stream, udata, input, scans, work, ends_u, ends_raw,
                           status, huff_pool, hist, enc, sizes, offsets, out, n_slots);
    else if (mode == 1)
        hipLaunchKernelGGL(transcode_kernel<1>, dim3(n_work), dim3(kTcThreads), lds, stream, udata, input, scans, work, ends_u, ends_raw,
                           status, huff_pool, hist, enc, sizes, offsets, out, n_slots);
    else
        hipLaunchKernelGGL(transcode_kernel<2>, dim3(n_work), dim3(kTcThreads), lds, stream, udata, input, scans, work, ends_u, ends_raw,
                           status, huff_pool, hist, enc, sizes, offsets, out, n_slots);
    return hipGetLastError();
}

hipError_t launch_transcode_offsets(hipStream_t stream, const DevScan *scans, const uint32_t *scan_ids, int n_scans, const uint32_t *sizes,
                                    const uint64_t *base, uint64_t *offsets, uint64_t *totals) {
    if (n_scans <= 0) return hipSuccess;
    hipLaunchKernelGGL(transcode_offsets_kernel, dim3(n_scans), dim3(1024), 0, stream, scans, scan_ids, sizes, base, offsets, totals);
    return hipGetLastError();
}


// ------------------------------------------------------------------------------------------------
// KTS: the transcode of scans WITHOUT restart intervals.  One lane per restart interval leaves a DRI = 0 scan to a single
// lane; instead the scan is cut into the decoder's self-synchronising subsequences (K2S: launch_subseq_sync), and lane i
// transcodes the WHOLE blocks that start inside subsequence i (block-aligned ownership, as subseq_final_kernel).  The
// output of a lane is no longer byte aligned: MODE 1 measures bits, an exclusive scan gives bit offsets, MODE 2 ORs the
// code words into a zeroed raw buffer (as the encoder's emit_kernel) and the encoder's stuffing kernels turn that into
// the final bytes.
// ------------------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(256) void subseq_transcode_kernel(const uint8_t *__restrict__ udata, const DevScan *__restrict__ scans,
                                                                const HuffWork *__restrict__ work, const uint32_t *__restrict__ ends_u,
                                                                DevScanStatus *__restrict__ status,
                                                                const DevHuffTable *__restrict__ huff_pool,
                                                                const uint32_t *__restrict__ exit_state,
                                                                const uint32_t *__restrict__ first_block, uint32_t *__restrict__ hist,
                                                                const EncHuffTable *__restrict__ enc, uint32_t *__restrict__ sub_bits,
                                                                const uint64_t *__restrict__ sub_bitoff,
                                                                const uint64_t *__restrict__ scan_raw_off, uint8_t *__restrict__ raw,
                                                                int n_slots) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t *tabs = smem;
    uint32_t *blk_info = reinterpret_cast<uint32_t *>(smem + (size_t)n_slots * sizeof(DevHuffTable));
    uint8_t *extra = reinterpret_cast<uint8_t *>(blk_info + kMaxBlocksPerMcu);  // MODE 0: hist[8][256] u32; else EncHuffTable[8]
    uint32_t *lhist = reinterpret_cast<uint32_t *>(extra);
    const EncHuffTable *lenc = reinterpret_cast<const EncHuffTable *>(extra);
    const HuffWork wk = work[blockIdx.x];
    const DevScan &s = scans[wk.scan];
    const uint32_t tid = threadIdx.x;
    if (MODE == 0) {
        for (uint32_t i = tid; i < kMaxHuffSlots * 256u; i += 256u) lhist[i] = 0;
    } else {
        const uint4 *src = reinterpret_cast<const uint4 *>(enc + (size_t)wk.scan * kMaxHuffSlots);
        uint4 *dst = reinterpret_cast<uint4 *>(extra);
        for (uint32_t i = tid; i < kMaxHuffSlots * sizeof(EncHuffTable) / 16; i += 256u) dst[i] = src[i];
    }
    sub_stage_tables(s, huff_pool, tabs, blk_info, n_slots, 256);  // ends with a barrier
    const DevScanStatus st = status[wk.scan];
    const uint32_t ulen = ends_u[s.ends_off];
    const uint32_t total_bits = ulen * 8;
    const uint32_t sub = wk.first_interval + tid;
    const uint32_t slot = s.sub_off + (sub < s.n_subs ? sub : 0);
    const uint32_t total_blocks = s.total_mcus * s.blocks_per_mcu;
    const uint32_t bpm = s.blocks_per_mcu;
    const bool closed_by_marker = st.terminator != 0;
    constexpr uint32_t kSlot16 = (uint32_t)(sizeof(DevHuffTable) / 16);

    bool live = sub < s.n_subs && st.n_ends != 0;
    uint32_t entry = 0;
    if (live && sub > 0) {
        const uint32_t prev = exit_state[slot - 1];
        if (prev & kSubBad) live = false;  // the stream ended or failed in an earlier subsequence: reported by that lane
        else entry = prev;
    }
    uint32_t b_in_mcu = (entry >> 6) & 31u, k = (entry >> 11) & 127u;
    uint32_t my_first = total_blocks, my_end = total_blocks;
    if (live) {
        my_first = first_block[slot] + (k != 0 ? 1u : 0u);
        if (sub + 1 < s.n_subs) {
            const uint32_t ex = exit_state[slot];
            if (!(ex & kSubBad)) my_end = first_block[slot + 1] + ((((ex >> 11) & 127u) != 0) ? 1u : 0u);
        }
        if (my_end > total_blocks) my_end = total_blocks;  // the reference stops after the last MCU
        if (my_first > my_end) my_first = my_end;
    }
    uint32_t count = my_end - my_first;
    // emit: a lane that failed in the measure pass owns no bits of the output (its size is 0): it must not write
    if (MODE == 2 && live && sub_bits[slot] == 0) count = 0;
    uint32_t err = 0;
    uint32_t nbits = 0;  // bits this lane produces
    uint32_t leftover_bits = 0;  // bits of the stream behind this lane's last block
    if (live && count != 0) {
        UBits r;
        const uint32_t start_bit = (sub << s.sub_shift) + (entry & 63u);
        ub_init(r, udata + s.data_off, start_bit >> 3, (total_bits + 7) >> 3);
        if (start_bit & 7u) ub_consume(r, start_bit & 7u);
        r.rem = (int32_t)total_bits - (int32_t)start_bit;
        uint32_t info = blk_info[b_in_mcu];
        // the tail of the block the previous lane owns: parsed, not transcoded
        while (k != 0 && err == 0) {
            const LdsHuff h = lds_huff16(tabs, (info >> 12) & 0xFFF);
            uint32_t sym;
            int32_t v;
            err = ub_symbol(r, h, false, closed_by_marker, sym, v);
            const uint32_t rr = sym >> 4;
            k = (sym & 15u) != 0 ? k + rr + 1u : (rr == 0 ? 64u : k + 16u);
            if (k >= 64u) {
                k = 0;
                b_in_mcu = (b_in_mcu + 1 == bpm) ? 0u : b_in_mcu + 1;
                info = blk_info[b_in_mcu];
            }
        }
        // MODE 2: the bit writer (emit_kernel's): words of the scan's raw buffer, MSB first, byte-swapped into stream order
        uint32_t *words = nullptr;
        uint64_t wi = 0;
        uint32_t fill = 0, acc = 0;
        if (MODE == 2) {
            words = reinterpret_cast<uint32_t *>(raw + scan_raw_off[wk.scan]);
            const uint64_t start = sub_bitoff[slot];
            wi = start >> 5;
            fill = (uint32_t)(start & 31);
        }
        auto put = [&](uint32_t code, uint32_t len) {
            nbits += len;
            if (MODE != 2) return;
            while (len) {
                const uint32_t room = 32u - fill;
                const uint32_t take = len < room ? len : room;
                const uint32_t part = (take == 32u) ? code : ((code >> (len - take)) & ((1u << take) - 1u));
                acc |= (take == 32u) ? part : (part << (room - take));
                fill += take;
                len -= take;
                if (fill == 32u) {
                    atomicOr(&words[wi], __builtin_bswap32(acc));
                    wi++;
                    fill = 0;
                    acc = 0;
                }
            }
        };
        for (uint32_t j = 0; j < count && err == 0; j++) {
            const uint32_t dc_slot = (info & 0xFFFu) / kSlot16, ac_slot = ((info >> 12) & 0xFFFu) / kSlot16;
            const LdsHuff hdc = lds_huff16(tabs, info & 0xFFF), hac = lds_huff16(tabs, (info >> 12) & 0xFFF);
            uint32_t sym;
            err = ub_huff(r, hdc, sym);
            if (err != 0) break;
            if (MODE == 0) atomicAdd(&lhist[dc_slot * 256u + sym], 1u);
            else put(lenc[dc_slot].code[sym], lenc[dc_slot].len[sym]);
            if (sym != 0) {
                uint32_t bits;
                if (sym > 16u) {
                    err = kDetailInvalidHuffmanCode;
                    break;
                }
                if (!ub_try_read_bits(r, sym, bits)) {
                    err = (r.rem == 0 && closed_by_marker) ? kDetailMarkerInData : kDetailStreamEnded;
                    break;
                }
                if (MODE != 0) put(bits, sym);
            }
            for (uint32_t i = 1; i < 64u;) {
                err = ub_huff(r, hac, sym);
                if (err != 0) break;
                if (MODE == 0) atomicAdd(&lhist[ac_slot * 256u + sym], 1u);
                else put(lenc[ac_slot].code[sym], lenc[ac_slot].len[sym]);
                const uint32_t rr = sym >> 4, sz = sym & 15u;
                if (sz != 0) {
                    i += rr + 1u;
                    uint32_t bits;
                    if (!ub_try_read_bits(r, sz, bits)) {
                        err = (r.rem == 0 && closed_by_marker) ? kDetailMarkerInData : kDetailStreamEnded;
                        break;
                    }
                    if (MODE != 0) put(bits, sz);
                } else {
                    if (rr == 0) break;
                    i += 16u;
                }
            }
            b_in_mcu = (b_in_mcu + 1 == bpm) ? 0u : b_in_mcu + 1;
            info = blk_info[b_in_mcu];
        }
        leftover_bits = (uint32_t)(r.rem > 0 ? r.rem : 0);
        if (MODE == 2 && err == 0) {
            if (my_end == total_blocks) {
                // ExitBitMode (JpegWriter.cs:141-166): the lane that writes the scan's last block pads the last byte with ones
                const uint64_t total = sub_bitoff[slot] + nbits;
                const uint32_t pad = (uint32_t)((8u - (total & 7u)) & 7u);
                if (pad) put((1u << pad) - 1u, pad);
            }
            if (fill) atomicOr(&words[wi], __builtin_bswap32(acc));
        }
    }
    if (MODE == 1 && sub < s.n_subs) sub_bits[slot] = err == 0 ? nbits : 0u;
    if (live && err != 0) atomicMin(&status[wk.scan].first_error, (sub << 8) | err);
    if (MODE == 0 && live && err == 0 && count != 0 && my_end == total_blocks) status[wk.scan].pad[2] = leftover_bits;
    if (MODE == 0) {
        __syncthreads();
        uint32_t *gh = hist + (size_t)wk.scan * kMaxHuffSlots * 256u;
        for (uint32_t i = tid; i < kMaxHuffSlots * 256u; i += 256u) {
            const uint32_t v = lhist[i];
            if (v != 0) atomicAdd(&gh[i], v);
        }
    }
}

// Exclusive scan of the subsequence bit counts of every scan: bitoff[i] = bits before subsequence i; totals[scan] = all bits.
__global__ __launch_bounds__(1024) void subseq_bit_offsets_kernel(const DevScan *__restrict__ scans, const uint32_t *__restrict__ scan_ids,
                                                                  const uint32_t *__restrict__ sub_bits, uint64_t *__restrict__ bitoff,
                                                                  uint64_t *__restrict__ totals) {
    __shared__ uint64_t part[1024];
    const DevScan &s = scans[scan_ids[blockIdx.x]];
    const uint32_t n = s.n_subs, tid = threadIdx.x;
    const uint32_t per = (n + 1023u) / 1024u;
    const uint32_t lo = tid * per < n ? tid * per : n, hi = lo + per < n ? lo + per : n;
    const uint32_t *sz = sub_bits + s.sub_off;
    uint64_t sum = 0;
    for (uint32_t i = lo; i < hi; i++) sum += sz[i];
    part[tid] = sum;
    __syncthreads();
    for (uint32_t d = 1; d < 1024u; d <<= 1) {
        const uint64_t v = tid >= d ? part[tid - d] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    if (tid == 1023) totals[blockIdx.x] = part[1023];
    uint64_t at = part[tid] - sum;
    uint64_t *of = bitoff + s.sub_off;
    for (uint32_t i = lo; i < hi; i++) {
        of[i] = at;
        at += sz[i];
    }
}

hipError_t launch_subseq_transcode(hipStream_t stream, int mode, const uint8_t *udata, const DevScan *scans, const HuffWork *work, int n_work,
                                   const uint32_t *ends_u, DevScanStatus *status, const DevHuffTable *huff_pool, const uint32_t *exit_state,
                                   const uint32_t *first_block, uint32_t *hist, const EncHuffTable *enc, uint32_t *sub_bits,
                                   const uint64_t *sub_bitoff, const uint64_t *scan_raw_off, uint8_t *raw, int n_slots) {
    if (n_work <= 0) return hipSuccess;
    const size_t lds = (size_t)n_slots * sizeof(DevHuffTable) + kMaxBlocksPerMcu * 4 + (size_t)kMaxHuffSlots * 1024;
    if (mode == 0)
        hipLaunchKernelGGL(subseq_transcode_kernel<0>, dim3(n_work), dim3(256), lds, stream, udata, scans, work, ends_u, status, huff_pool,
                           exit_state, first_block, hist, enc, sub_bits, sub_bitoff, scan_raw_off, raw, n_slots);
    else if (mode == 1)
        hipLaunchKernelGGL(subseq_transcode_kernel<1>, dim3(n_work), dim3(256), lds, stream, udata, scans, work, ends_u, status, huff_pool,
                           exit_state, first_block, hist, enc, sub_bits, sub_bitoff, scan_raw_off, raw, n_slots);
    else
        hipLaunchKernelGGL(subseq_transcode_kernel<2>, dim3(n_work), dim3(256), lds, stream, udata, scans, work, ends_u, status, huff_pool,
                           exit_state, first_block, hist, enc, sub_bits, sub_bitoff, scan_raw_off, raw, n_slots);
    return hipGetLastError();
}

hipError_t launch_subseq_bit_offsets(hipStream_t stream, const DevScan *scans, const uint32_t *scan_ids, int n_scans, const uint32_t *sub_bits,
                                     uint64_t *bitoff, uint64_t *totals) {
    if (n_scans <= 0) return hipSuccess;
    hipLaunchKernelGGL(subseq_bit_offsets_kernel, dim3(n_scans), dim3(1024), 0, stream, scans, scan_ids, sub_bits, bitoff, totals);
    return hipGetLastError();
}

}  // namespace jpgpu
