// jpeglibrary_amd/csrc/k2_huffman.hip -- K2: Huffman MCU decode, one lane per restart interval; the pooled lookups (lut_pool_kernel)
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
// K2: Huffman MCU decode.  One lane per restart interval.
// ------------------------------------------------------------------------------------------------

__global__ __launch_bounds__(256) void lut_pool_kernel(const DevHuffTable *__restrict__ pool, uint8_t *__restrict__ lut_pool) {
    const DevHuffTable &h = pool[blockIdx.x >> 1];
    const bool is_dc = (blockIdx.x & 1) != 0;
    uint8_t *tab = lut_pool + (size_t)(blockIdx.x >> 1) * kLutPoolBytesPerTable;
    uint8_t *img = tab + (is_dc ? kK2TabBytes : 0u);  // the u16 image (the K2S round kernel's source)
    uint16_t *l1 = reinterpret_cast<uint16_t *>(img);
    uint16_t *l2 = reinterpret_cast<uint16_t *>(img + kK2L1Bytes);
    // the u32 image (K2, the K2S final pass): first level with values and pairs | the same second level, header and arrays
    const uint32_t lb = is_dc ? kK2DcBits : kK2AcBits;
    uint8_t *img32 = tab + (is_dc ? kPoolNewDc : kPoolNewAc);
    uint32_t *f1 = reinterpret_cast<uint32_t *>(img32);
    uint16_t *f2 = reinterpret_cast<uint16_t *>(img32 + (4u << lb));
    __shared__ uint32_t first_miss, first_miss32;
    if (threadIdx.x == 0) first_miss = 1u << kK2LutBits, first_miss32 = 1u << lb;
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < (1u << kK2LutBits); i += 256) {
        // a code of at most 11 bits is decided by the prefix alone (maxcode[l] has its low 16 - l bits set): evaluate with ones behind it
        const uint32_t e = k2_entry_of(h, (i << (16 - kK2LutBits)) | ((1u << (16 - kK2LutBits)) - 1u), is_dc, kK2LutBits);
        l1[i] = (uint16_t)e;
        if (e == 0) atomicMin(&first_miss, i);  // (a bad category is an answer, not a miss)
    }
    for (uint32_t i = threadIdx.x; i < (1u << lb); i += 256) {
        const uint32_t e = k2_fast_entry(h, i, is_dc, lb);
        f1[i] = e;
        if (e == 0) atomicMin(&first_miss32, i);
    }
    __syncthreads();
    {
        const uint32_t lo = first_miss << (16 - kK2LutBits);
        const uint32_t t16 = lo > 65536u - kK2L2Entries ? lo : 65536u - kK2L2Entries;
        for (uint32_t j = threadIdx.x; j < kK2L2Entries; j += 256) l2[j] = t16 + j < 65536u ? (uint16_t)k2_entry_of(h, t16 + j, is_dc, 16) : (uint16_t)0;
        if (threadIdx.x < 4) reinterpret_cast<uint32_t *>(img + kK2L1Bytes + 2u * kK2L2Entries)[threadIdx.x] = threadIdx.x == 0 ? t16 : 0u;
        if (threadIdx.x < kK2SmallBytes / 16)
            reinterpret_cast<uint4 *>(img + kK2L1Bytes + 2u * kK2L2Entries + 16u)[threadIdx.x] =
                reinterpret_cast<const uint4 *>(reinterpret_cast<const uint8_t *>(&h) + offsetof(DevHuffTable, maxcode))[threadIdx.x];
    }
    {
        const uint32_t lo = first_miss32 << (16 - lb);
        const uint32_t t16 = lo > 65536u - kK2L2Entries ? lo : 65536u - kK2L2Entries;
        for (uint32_t j = threadIdx.x; j < kK2L2Entries; j += 256) f2[j] = t16 + j < 65536u ? (uint16_t)k2_entry_of(h, t16 + j, is_dc, 16) : (uint16_t)0;
        if (threadIdx.x < 4) reinterpret_cast<uint32_t *>(img32 + (4u << lb) + 2u * kK2L2Entries)[threadIdx.x] = threadIdx.x == 0 ? t16 : 0u;
        if (threadIdx.x < kK2SmallBytes / 16)
            reinterpret_cast<uint4 *>(img32 + (4u << lb) + 2u * kK2L2Entries + 16u)[threadIdx.x] =
                reinterpret_cast<const uint4 *>(reinterpret_cast<const uint8_t *>(&h) + offsetof(DevHuffTable, maxcode))[threadIdx.x];
    }
}

#ifdef JPGPU_K2_PROFILE
__device__ unsigned long long k2_prof[8];
#define K2_TICK() __builtin_readcyclecounter()
#define K2_PROF_ADD(i, v) do { if (lane == 0) atomicAdd(&k2_prof[i], (unsigned long long)(v)); } while (0)
extern "C" int jpgpu_debug_k2_profile(unsigned long long *out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(k2_prof), sizeof(k2_prof)) != hipSuccess) return 1;
    if (reset) { unsigned long long z[8] = {}; if (hipMemcpyToSymbol(HIP_SYMBOL(k2_prof), z, sizeof z) != hipSuccess) return 1; }
    return 0;
}
#else
#define K2_TICK() 0ull
#define K2_PROF_ADD(i, v) do { (void)(v); } while (0)
#endif

// One wave's 64 restart intervals wave_first .. wave_first + 63 of scan `scan_index` (tables staged, the wave's coefficient staging zero on
// entry and on exit).
__device__ __forceinline__ void k2_wave(const uint8_t *__restrict__ udata, const DevScan &s, uint32_t scan_index, uint32_t wave_first,
                                        const uint32_t *__restrict__ ends_u, DevScanStatus *__restrict__ status, int16_t *__restrict__ coefs,
                                        const uint8_t *tabs, const uint32_t *blk_info, uint8_t *stage, uint8_t *ring, uint32_t lane,
                                        unsigned long long k2_t0) {
    struct { uint32_t scan; } wk = {scan_index};
    const unsigned long long k2_t1 = K2_TICK();
    const DevScanStatus st = status[wk.scan];
    const uint32_t n_ends = st.n_ends;
    const uint32_t n_intervals = s.n_intervals;
    const uint32_t total_mcus = s.total_mcus;
    const uint32_t dri_eff = s.dri ? s.dri : total_mcus;
    const uint32_t bpm = s.blocks_per_mcu;
    const uint32_t interval = wave_first + lane;
    const bool active = interval < n_ends;
    const uint32_t *eu = ends_u + s.ends_off;
    const uint8_t *ubase = udata + s.data_off;

    // the lane's stream == a fresh JpegBitReader on its restart interval (ref: JpegBitReader.cs)
    uint32_t my_mcus = 0;
    bool closed_by_marker = false;
    uint32_t u0 = 0, u1 = 0;
    if (active) {
        u0 = interval == 0 ? 0u : eu[interval - 1] + 2u;
        u1 = eu[interval];
        my_mcus = (interval == n_intervals - 1) ? total_mcus - interval * dri_eff : dri_eff;
        closed_by_marker = !(interval == n_ends - 1 && st.terminator == 0);
    }
    const int32_t pm1_0 = (int32_t)((u0 & 3u) * 8u) - 1;             // bit position - 1, relative to the aligned origin
    const int32_t endpos = pm1_0 + 1 + (int32_t)((u1 - u0) * 8u);    // first bit after the interval's data
    K2Feed feed;
    K2Pos pos;
    {
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
    }
    // the wave iterates to the largest MCU count among its lanes (only the image's last interval is shorter)
    uint32_t wave_mcus = 0;
    if (wave_first < n_ends) {
        wave_mcus = dri_eff;
        if (wave_first == n_intervals - 1) wave_mcus = total_mcus - wave_first * dri_eff;
    }

    int32_t pred0 = 0, pred1 = 0, pred2 = 0, pred3 = 0;  // DcPredictor per scan component
    uint32_t err = 0;
    const unsigned long long k2_t2 = K2_TICK();
    unsigned long long k2_dec = 0, k2_top = 0, k2_fl = 0;
    uint8_t *my_stage = stage + lane * 128;
    const uint32_t swz16 = ((lane >> 1) & 7u) << 4;  // XOR swizzle of the 16-byte chunks of the lane's staged block

    // flush addressing: lane (blk, chunk) of pass `it` stores 16 bytes of the block staged by lane blk
    const uint64_t coef_off = s.coef_off;

    for (uint32_t mcu = 0; mcu < wave_mcus; mcu++) {
        for (uint32_t b = 0; b < bpm; b++) {
            const uint32_t bi = __builtin_amdgcn_readfirstlane(blk_info[b]);  // wave-uniform
            const uint32_t ci = bi & 0xFFu;
            const K2Tab hdc = k2_tab_dc(tabs, bi);
            const K2Tab hac = k2_tab_ac(tabs, bi);
            const unsigned long long k2_a = K2_TICK();
            int32_t lim = k2_limit(endpos, feed.wr);
            if (active && err == 0 && mcu < my_mcus) {
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
                uint32_t i2 = err == 0 ? 2u : 128u;  // 2 x zig-zag index of the next coefficient
                while (i2 < 128u) {
                    // one step = one symbol or, where the lookup held two, both (round 6): Math.Min(i++, 63) for a coefficient; EOB /
                    // ZRL store a zero at a position nothing was written to yet; a step of one symbol stores it twice
                    k2_symbol<false>(ring, feed, pos, endpos, lim, hac, closed_by_marker, i2, v, vb, ia, adv_b, err);
                    const uint32_t at = ia - 2u < 126u ? ia - 2u : 126u;
                    *reinterpret_cast<int16_t *>(my_stage + (at ^ swz16)) = (int16_t)v;
                    i2 = ia + 2u * adv_b;
                    const uint32_t at_b = i2 - 2u < 126u ? i2 - 2u : 126u;
                    *reinterpret_cast<int16_t *>(my_stage + (at_b ^ swz16)) = (int16_t)vb;
                }
                // the block the reference throws in: blocks in front of it have reached the writer, this one and the rest have not
                if (err != 0)
                    atomicMax(&status[wk.scan].pad[1], fail_block_word(((uint64_t)interval * dri_eff + mcu) * bpm + b));
            }
            // top up the ring HERE: the wait for the prefetched chunk then only covers memory operations issued before this
            // block was decoded (the chunk itself and the previous block's coefficient stores), never fresh ones
            const unsigned long long k2_b = K2_TICK();
            k2_topup(ring, feed, pos.pm1);
            const unsigned long long k2_c = K2_TICK();
            // flush 64 blocks of this wave to the coefficient buffer as whole 128-byte lines, re-zero the staging
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
                const uint32_t owner = wave_first + blk;
                if (owner < n_ends) {
                    const uint32_t owner_mcus = (owner == n_intervals - 1) ? total_mcus - owner * dri_eff : dri_eff;
                    if (mcu < owner_mcus) {
                        const uint64_t block_index = coef_off + ((uint64_t)owner * dri_eff + mcu) * bpm + b;
                        *reinterpret_cast<uint4 *>(coefs + block_index * 64 + chunk * 8) = v;
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const unsigned long long k2_d = K2_TICK();
            k2_dec += k2_b - k2_a;
            k2_top += k2_c - k2_b;
            k2_fl += k2_d - k2_c;
        }
    }
    K2_PROF_ADD(0, 1);
    K2_PROF_ADD(1, k2_t1 - k2_t0);
    K2_PROF_ADD(2, k2_t2 - k2_t1);
    K2_PROF_ADD(3, k2_dec);
    K2_PROF_ADD(4, k2_top);
    K2_PROF_ADD(5, k2_fl);
    K2_PROF_ADD(6, K2_TICK() - k2_t0);

    if (active) {
        int32_t rem = endpos - (pos.pm1 + 1);
        if (rem < 0) rem = 0;
        const uint32_t code = restart_check(s, st, &status[wk.scan], interval, n_ends, n_intervals, dri_eff, rem, err);
        if (code != kNoError) atomicMin(&status[wk.scan].first_error, code);
    }
}

// K2 proper: a workgroup per WAVES * 64 consecutive restart intervals of one scan.
template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void huffman_decode_kernel(const uint8_t *__restrict__ udata,
                                                                    const DevScan *__restrict__ scans,
                                                                    const HuffWork *__restrict__ work,
                                                                    const uint32_t *__restrict__ ends_u,
                                                                    DevScanStatus *__restrict__ status,
                                                                    const DevHuffTable *__restrict__ huff_pool,
                                                                    int16_t *__restrict__ coefs, int n_slots,
                                                                    const uint8_t *__restrict__ lut_pool, uint32_t tab_bytes) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t *tabs = smem;                  // tab_bytes: the batch's largest set of staged tables
    uint8_t *wave_all = smem + tab_bytes;  // WAVES * kK2WaveBytes
    uint32_t *blk_info = reinterpret_cast<uint32_t *>(wave_all + WAVES * kK2WaveBytes);  // [kMaxBlocksPerMcu]
    const HuffWork wk = work[blockIdx.x];
    const DevScan &s = scans[wk.scan];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long k2_t0 = K2_TICK();
    // stage this scan's Huffman tables (ref: InitDecodeComponents resolves them per scan, JpegHuffmanScanDecoder.cs:63-64)
    k2_stage_scan_tables(s, lut_pool, tabs, blk_info, n_slots, 64 * WAVES);
    uint8_t *stage = wave_all + wave * kK2WaveBytes;
    uint8_t *ring = stage + 8192 + lane * kK2RingStride;
    {
        const uint4 z = {0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < 8; i++) reinterpret_cast<uint4 *>(stage)[i * 64 + lane] = z;
    }
    __syncthreads();
    k2_wave(udata, s, wk.scan, wk.first_interval + wave * 64, ends_u, status, coefs, tabs, blk_info, stage, ring, lane, k2_t0);
}

// POOL (round 6; the K2S final pass has had it since round 5).  Runs of consecutive scans that stage the same tables in the same slots
// -- every batch of files from one encoder -- need no workgroup per WAVES * 64 intervals of ONE scan: one workgroup per CU stages the
// tables once and every WAVE takes the next 64 intervals of the run from a counter until there are none.  No wave waits for a slower one
// of its workgroup before the CU gets new work, nothing is restaged (a workgroup of the plain form lives for 24 block steps), and the
// last workgroup of a scan is no longer half empty (8 100 intervals = 11.5 workgroups of 704).  The counter is never cleared: a launch
// draws n_chunks + (waves launched) tickets -- every wave exactly one beyond the end -- and the host passes the sum of the launches before.
template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void huffman_pool_kernel(const uint8_t *__restrict__ udata, const DevScan *__restrict__ scans,
                                                                  const HuffWork *__restrict__ work, uint32_t n_chunks,
                                                                  uint32_t *__restrict__ counter, uint32_t ticket_base,
                                                                  const uint32_t *__restrict__ ends_u, DevScanStatus *__restrict__ status,
                                                                  int16_t *__restrict__ coefs, int n_slots, const uint8_t *__restrict__ lut_pool,
                                                                  uint32_t tab_bytes) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t *tabs = smem;
    uint8_t *wave_all = smem + tab_bytes;
    uint32_t *blk_info = reinterpret_cast<uint32_t *>(wave_all + WAVES * kK2WaveBytes);
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long k2_t0 = K2_TICK();
    k2_stage_scan_tables(scans[work[0].scan], lut_pool, tabs, blk_info, n_slots, 64 * WAVES);  // (every entry of `work` stages the same)
    uint8_t *stage = wave_all + wave * kK2WaveBytes;
    uint8_t *ring = stage + 8192 + lane * kK2RingStride;
    {
        const uint4 z = {0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < 8; i++) reinterpret_cast<uint4 *>(stage)[i * 64 + lane] = z;
    }
    __syncthreads();
#if !defined(JPGPU_K2_TICKET_AHEAD)
    for (;;) {
        uint32_t c = 0;
        if (lane == 0) c = atomicAdd(counter, 1u) - ticket_base;
        c = (uint32_t)__builtin_amdgcn_readfirstlane((int)c);
        if (c >= n_chunks) break;
        const HuffWork wk = work[c];
        k2_wave(udata, scans[wk.scan], wk.scan, wk.first_interval, ends_u, status, coefs, tabs, blk_info, stage, ring, lane, c == 0 ? k2_t0 : K2_TICK());
    }
#else
    // (measured and NOT adopted, round 6: a wave drawing the ticket of its NEXT chunk before it starts on the one it holds, so that the
    // counter's round trip lies under the chunk's first loads.  K2 4.93 -> 5.10 ms per 1024 x 4K, the benchmark canvases' Huffman stage
    // 2.26 -> 2.48, DRI = 0 unchanged (profiles/r06_ticket_ahead_ab.txt): the loads of the chunk then queue behind the atomic in the
    // wave's in-order counter, and a chunk waits for ITS wave.)
    uint32_t c = 0;
    if (lane == 0) c = atomicAdd(counter, 1u) - ticket_base;
    c = (uint32_t)__builtin_amdgcn_readfirstlane((int)c);
    while (c < n_chunks) {
        uint32_t ahead = 0;
        if (lane == 0) ahead = atomicAdd(counter, 1u) - ticket_base;
        const HuffWork wk = work[c];
        k2_wave(udata, scans[wk.scan], wk.scan, wk.first_interval, ends_u, status, coefs, tabs, blk_info, stage, ring, lane, c == 0 ? k2_t0 : K2_TICK());
        c = (uint32_t)__builtin_amdgcn_readfirstlane((int)ahead);
    }
#endif
}

static size_t k2_lds_bytes(uint32_t tab_bytes, int waves) { return (size_t)tab_bytes + (size_t)waves * kK2WaveBytes + kMaxBlocksPerMcu * sizeof(uint32_t); }

template <int WAVES>
static hipError_t launch_huffman_w(hipStream_t stream, const uint8_t *data, const DevScan *scans, const HuffWork *work, int n_work,
                                   const uint32_t *ends, DevScanStatus *status, const DevHuffTable *huff_pool, int16_t *coefs,
                                   int n_slots, const uint8_t *lut_pool, uint32_t tab_bytes) {
    const size_t lds = k2_lds_bytes(tab_bytes, WAVES);
    static std::atomic<uint64_t> configured{0};
    const hipError_t ea = allow_dynamic_lds(reinterpret_cast<const void *>(&huffman_decode_kernel<WAVES>), 160 * 1024, configured);
    if (ea != hipSuccess) return ea;
    hipLaunchKernelGGL((huffman_decode_kernel<WAVES>), dim3(n_work), dim3(64 * WAVES), lds, stream, data, scans, work, ends, status, huff_pool, coefs,
                       n_slots, lut_pool, tab_bytes);
    return hipGetLastError();
}

// `work` holds one entry per huffman_waves(tab_bytes) * 64 restart intervals (DeviceBatch builds it with the same function);
// tab_bytes = the largest k2_scan_tab_bytes() among the batch's scans
hipError_t launch_huffman(hipStream_t stream, const uint8_t *data, const DevScan *scans, const HuffWork *work, int n_work,
                          const uint32_t *ends, DevScanStatus *status, const DevHuffTable *huff_pool, int16_t *coefs,
                          int n_slots, const uint8_t *lut_pool, uint32_t tab_bytes) {
    if (n_work <= 0) return hipSuccess;
#define JPGPU_K2_CASE(W) case W: return launch_huffman_w<W>(stream, data, scans, work, n_work, ends, status, huff_pool, coefs, n_slots, lut_pool, tab_bytes)
    switch (huffman_waves(tab_bytes)) {
        JPGPU_K2_CASE(11);
        JPGPU_K2_CASE(10);
        JPGPU_K2_CASE(9);
        JPGPU_K2_CASE(8);
    default:
        JPGPU_K2_CASE(7);  // (eight AC tables)
    }
#undef JPGPU_K2_CASE
}

template <int WAVES>
static hipError_t launch_huffman_pool_w(hipStream_t stream, const uint8_t *data, const DevScan *scans, const HuffWork *work, int n_chunks, uint32_t *counter,
                                        uint32_t ticket_base, int groups, const uint32_t *ends, DevScanStatus *status, int16_t *coefs, int n_slots,
                                        const uint8_t *lut_pool, uint32_t tab_bytes) {
    const size_t lds = k2_lds_bytes(tab_bytes, WAVES);
    static std::atomic<uint64_t> configured{0};
    const hipError_t ea = allow_dynamic_lds(reinterpret_cast<const void *>(&huffman_pool_kernel<WAVES>), 160 * 1024, configured);
    if (ea != hipSuccess) return ea;
    hipLaunchKernelGGL((huffman_pool_kernel<WAVES>), dim3(groups), dim3(64 * WAVES), lds, stream, data, scans, work, (uint32_t)n_chunks, counter, ticket_base,
                       ends, status, coefs, n_slots, lut_pool, tab_bytes);
    return hipGetLastError();
}
// One pooled run: `work` = its n_chunks entries (scan, first interval) of 64 intervals each, all staging the same tables; `groups`
// workgroups of huffman_waves(tab_bytes) waves; draws n_chunks + groups * waves tickets from *counter, the first of them ticket_base.
hipError_t launch_huffman_pool(hipStream_t stream, const uint8_t *data, const DevScan *scans, const HuffWork *work, int n_chunks, uint32_t *counter,
                               uint32_t ticket_base, int groups, const uint32_t *ends, DevScanStatus *status, int16_t *coefs, int n_slots,
                               const uint8_t *lut_pool, uint32_t tab_bytes) {
    if (n_chunks <= 0 || groups <= 0) return hipSuccess;
#define JPGPU_K2_CASE(W) case W: return launch_huffman_pool_w<W>(stream, data, scans, work, n_chunks, counter, ticket_base, groups, ends, status, coefs, n_slots, lut_pool, tab_bytes)
    switch (huffman_waves(tab_bytes)) {
        JPGPU_K2_CASE(11);
        JPGPU_K2_CASE(10);
        JPGPU_K2_CASE(9);
        JPGPU_K2_CASE(8);
    default:
        JPGPU_K2_CASE(7);
    }
#undef JPGPU_K2_CASE
}

// Lookups of every table of the pool (once per upload: the tables of a batch do not change between decodes).
hipError_t launch_lut_pool(hipStream_t stream, const DevHuffTable *huff_pool, int n_tables, uint8_t *lut_pool) {
    if (n_tables <= 0) return hipSuccess;
    hipLaunchKernelGGL(lut_pool_kernel, dim3(2 * n_tables), dim3(256), 0, stream, huff_pool, lut_pool);
    return hipGetLastError();
}

}  // namespace jpgpu
