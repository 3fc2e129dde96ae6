// jpeglibrary_amd/csrc/k1_markers.hip -- K1: marker index + unstuffing, ingest verification (first_marker_kernel), gather of pinned pieces
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
// K1: marker index + unstuffing.  One workgroup per scan job.
//
// In entropy-coded data 0xFF is always "fresh" (the second byte of FF00 / FFxx is never FF), so
// "data[p]==FF && data[p+1] not in {00,FF}" identifies a marker without context -- the same rule
// JpegBitReader.FillBuffer applies byte-serially (ref: JpegBitReader.cs:95-138).
// ends[k] = offset of the FF that closes restart interval k.  Indexing stops at the first non-RST marker
// (or at ANY marker when DRI == 0: the reference's bit reader stops feeding bits at every marker).
//
// The same pass writes `udata`, the entropy segment as the reference's bit reader sees it: stuffed zeros (FF00 -> FF)
// and fill bytes (FFFF -> skip) removed, every marker replaced by FF FF (16 one-bits: exactly the ones-padding
// PeekBits(16) applies when an interval runs dry, JpegBitReader.cs:163-167).  ends_u[k] = position in udata of the
// FF FF pair closing interval k.  The Huffman kernels read udata with plain word loads: no per-byte FF handling in
// their hot loops.  udata occupies the same byte range as the raw segment (it is never longer).
// ------------------------------------------------------------------------------------------------

constexpr int kScanThreads = 256;
constexpr uint32_t kInf = 0xFFFFFFFFu;
constexpr uint32_t kChunkBytes = kScanThreads * 16;  // one chunk = one 4 KiB tile, 16 bytes per lane
constexpr int kCountChunksPerWg = 4;                  // marker_count_kernel: tiles in flight per workgroup

// Per-lane classification of 16 consecutive bytes of an entropy segment (bit j = byte off + j).
struct ByteClass {
    uint32_t w[5];      // the 16 bytes + the byte after them
    uint32_t rst, term; // FF of an RSTn marker / of any other marker (every marker when DRI == 0)
    uint32_t keep;      // bytes copied to udata as they are (markers' FF included, their code byte excluded)
};

// The lane's 16 bytes and, for the two lanes at the ends of a wave, the byte before / after them.  Nothing here waits for the
// data, so a caller can issue several of these before it classifies the first (lanes out of range read the segment's first bytes
// and drop them: no branch around the 16-byte load).
struct Raw16 {
    uint4 v;
    uint32_t edge;
};
__device__ __forceinline__ Raw16 load16(const uint8_t *p, int64_t off, uint32_t len) {
    const bool in_range = off < (int64_t)len && off + 16 > 0;
    const uint32_t l = lane_id();
    Raw16 r;
    r.v = *reinterpret_cast<const uint4 *>(p + (in_range ? off : 0));
    r.edge = 0;
    if (in_range && (l == 63 || (l == 0 && off >= 1))) r.edge = *(p + (l == 0 ? off - 1 : off + 16));  // one two-lane load
    if (!in_range) r.v = uint4{0, 0, 0, 0};
    return r;
}

__device__ __forceinline__ ByteClass classify16(const Raw16 &raw, int64_t off, uint32_t len, bool any_marker_terminates) {
    ByteClass c;
    c.w[0] = raw.v.x;
    c.w[1] = raw.v.y;
    c.w[2] = raw.v.z;
    c.w[3] = raw.v.w;
    c.rst = c.term = c.keep = 0;
    uint32_t prev = 0;
    const bool in_range = off < (int64_t)len && off + 16 > 0;
    // the byte after / before the lane's 16: the neighbouring lane holds it (consecutive lanes take consecutive 16 bytes; a lane
    // out of range holds zeros, and its byte is only ever asked for by positions whose own range checks fail); the two lanes
    // at the ends of the wave use the byte they loaded.  All 64 lanes get here (DPP reads the neighbours' registers).
    {
        const uint32_t l = lane_id();
        const uint32_t nxt = dpp0<0x130>(c.w[0]) & 0xFFu, prv = dpp0<0x138>(c.w[3]) >> 24;
        c.w[4] = l == 63 ? raw.edge : nxt;
        if (off >= 1) prev = l == 0 ? raw.edge : prv;
        if (!in_range) c.w[4] = 0, prev = 0;
    }
    if (off >= 1 && off + 17 <= (int64_t)len) {
        // interior lane (all but the first and last few lanes of a segment): SWAR over the four dwords, flags in bit 7 of
        // each byte, then packed to one bit per byte.
        const uint32_t dw[6] = {prev << 24, c.w[0], c.w[1], c.w[2], c.w[3], c.w[4]};
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint32_t x = dw[i + 1];
            const uint32_t nbw = __builtin_amdgcn_alignbit(dw[i + 2], x, 8);   // byte j+1 of every byte j
            const uint32_t pvw = __builtin_amdgcn_alignbit(x, dw[i], 24);      // byte j-1
#define JPGPU_Z80(v_) (~((((v_)&0x7F7F7F7Fu) + 0x7F7F7F7Fu) | (v_) | 0x7F7F7F7Fu))  /* 0x80 where the byte is 0 */
            const uint32_t ff = JPGPU_Z80(~x), nff = JPGPU_Z80(~nbw), n00 = JPGPU_Z80(nbw), pff = JPGPU_Z80(~pvw);
            const uint32_t nrst = JPGPU_Z80((nbw & 0xF8F8F8F8u) ^ 0xD0D0D0D0u);
#undef JPGPU_Z80
            const uint32_t marker = ff & ~n00 & ~nff;
            const uint32_t rst = any_marker_terminates ? 0u : (marker & nrst);
            const uint32_t dropped = (pff & ~ff) | (ff & nff);
#define JPGPU_PACK4(m_) (((((m_) >> 7) * 0x00204081u) >> 21) & 0xFu)
            c.rst |= JPGPU_PACK4(rst) << (4 * i);
            c.term |= JPGPU_PACK4(marker & ~rst) << (4 * i);
            c.keep |= JPGPU_PACK4(~dropped & 0x80808080u) << (4 * i);
#undef JPGPU_PACK4
        }
        return c;
    }
#pragma unroll
    for (int j = 0; j < 16; j++) {
        const uint32_t b = (c.w[j >> 2] >> ((j & 3) * 8)) & 0xFF;
        const uint32_t nb = (c.w[(j + 1) >> 2] >> (((j + 1) & 3) * 8)) & 0xFF;
        const uint32_t pv = j == 0 ? prev : ((c.w[(j - 1) >> 2] >> (((j - 1) & 3) * 8)) & 0xFF);
        const int64_t pos = off + j;
        const bool in_seg = pos >= 0 && pos < (int64_t)len;
        const bool has_next = pos + 1 < (int64_t)len;
        const bool prev_ff = pos >= 1 && pv == 0xFF;
        const bool is_marker = in_seg && has_next && b == 0xFF && nb != 0x00 && nb != 0xFF;
        const bool is_rst = is_marker && ((nb & 0xF8) == 0xD0) && !any_marker_terminates;
        // not copied: the stuffed 00 of FF00, the first FF of FFFF, an FF that is the very last byte, and the code byte of
        // a marker (the lane that owns the marker's FF writes FF FF for the pair)
        const bool dropped = (prev_ff && b != 0xFF) || (b == 0xFF && (!has_next || nb == 0xFF));
        c.rst |= (uint32_t)is_rst << j;
        c.term |= (uint32_t)(is_marker && !is_rst) << j;
        c.keep |= (uint32_t)(in_seg && !dropped) << j;
    }
    return c;
}

// workgroup-wide exclusive prefix + total of one value per lane
__device__ __forceinline__ ByteClass classify16(const uint8_t *p, int64_t off, uint32_t len, bool any_marker_terminates) {
    return classify16(load16(p, off, len), off, len, any_marker_terminates);
}

__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t *sh_wave /*[kScanThreads/64]*/, uint32_t &total) {
    const uint32_t incl = wave_inclusive_scan(v);
    const uint32_t wave = threadIdx.x >> 6;
    __syncthreads();  // sh_wave may still be read from a previous call
    if (lane_id() == 63) sh_wave[wave] = incl;
    __syncthreads();
    uint32_t base = 0;
    total = 0;
#pragma unroll
    for (int i = 0; i < kScanThreads / 64; i++) {
        const uint32_t t = sh_wave[i];
        if ((uint32_t)i < wave) base += t;
        total += t;
    }
    return base + incl - v;
}

// K1a: per-chunk summary (RST markers, bytes udata will receive, first terminating marker).
__global__ __launch_bounds__(kScanThreads) void marker_count_kernel(const uint8_t *__restrict__ data, const DevScan *__restrict__ scans,
                                                                     const ChunkWork *__restrict__ work, uint32_t n_work,
                                                                     ChunkSum *__restrict__ sums) {
    // kCountChunksPerWg entries of the work list per workgroup (they may belong to different scans), all their loads issued
    // before the first is classified: a workgroup that waits for one 4 KiB tile at a time spends its life in load latency
    // (work entry -> scan descriptor -> data: 2.1 TB/s with 262 144 such workgroups per 1024 x 4K).
    __shared__ uint32_t sh_cnt[kCountChunksPerWg][kScanThreads / 64], sh_term[kCountChunksPerWg][kScanThreads / 64];
    const uint32_t first = blockIdx.x * kCountChunksPerWg;
    Raw16 raw[kCountChunksPerWg];
    int64_t off[kCountChunksPerWg];
    uint32_t len[kCountChunksPerWg], sum_at[kCountChunksPerWg];
    bool any_marker[kCountChunksPerWg];
    // three separate loops: the work entries, then the descriptors, then the data -- each level's loads in flight together
    ChunkWork wk[kCountChunksPerWg];
#pragma unroll
    for (int i = 0; i < kCountChunksPerWg; i++) wk[i] = work[first + i < n_work ? first + i : n_work - 1];  // a tail entry repeats the last one (nothing is written for it)
    uint64_t data_off[kCountChunksPerWg];
#pragma unroll
    for (int i = 0; i < kCountChunksPerWg; i++) {
        const DevScan &s = scans[wk[i].scan];
        data_off[i] = s.data_off;
        len[i] = s.data_len;
        sum_at[i] = s.chunk_off + wk[i].chunk;
        any_marker[i] = s.dri == 0;
    }
#pragma unroll
    for (int i = 0; i < kCountChunksPerWg; i++) {
        off[i] = -(int64_t)(data_off[i] & 15u) + (int64_t)wk[i].chunk * kChunkBytes + (int64_t)threadIdx.x * 16;
        raw[i] = load16(data + data_off[i], off[i], len[i]);
    }
#pragma unroll
    for (int i = 0; i < kCountChunksPerWg; i++) {
        const ByteClass c = classify16(raw[i], off[i], len[i], any_marker[i]);
        // totals only: both counts in one word (a wave holds at most 512 markers and writes at most 1536 bytes), one sum per wave
        const uint32_t cnt = wave_sum((uint32_t)__builtin_popcount(c.rst) |
                                      ((uint32_t)(__builtin_popcount(c.keep) + __builtin_popcount(c.rst | c.term)) << 16));
        // first terminating marker of the wave: offsets grow with the lane, so it is in the lowest lane that has one
        const uint64_t has_term = __ballot(c.term != 0);
        uint32_t tpos = kInf;
        if (has_term != 0) tpos = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(off[i] + __builtin_ctz(c.term | 0x10000u)), (int)__builtin_ctzll(has_term));
        if (lane_id() == 0) {
            sh_cnt[i][threadIdx.x >> 6] = cnt;
            sh_term[i][threadIdx.x >> 6] = tpos;
        }
    }
    __syncthreads();
    if (threadIdx.x < kCountChunksPerWg && first + threadIdx.x < n_work) {
        uint32_t total = 0, term = kInf;
#pragma unroll
        for (int w = 0; w < kScanThreads / 64; w++) {
            total += sh_cnt[threadIdx.x][w];
            term = sh_term[threadIdx.x][w] < term ? sh_term[threadIdx.x][w] : term;
        }
        ChunkSum cs;
        cs.rst_cnt = total & 0xFFFFu;
        cs.keep_cnt = total >> 16;
        cs.first_term = term;
        cs.pad = 0;
        uint32_t at = sum_at[0];
#pragma unroll
        for (int i = 1; i < kCountChunksPerWg; i++) at = threadIdx.x == (uint32_t)i ? sum_at[i] : at;
        sums[at] = cs;
    }
}

// K0 (ingest): where does the entropy data that follows a file's first SOS end?  The host parses headers only and plans
// every file as "one scan whose data runs up to an EOI"; this kernel looks at the bytes the host never touches and reports,
// per file, the offset of the first marker in them that is not RSTn -- what JpegDecoder.Identify's walk over the same bytes
// would stop at next (ref: JpegDecoder.cs:75-162, JpegReader.cs:120-158: FF00 and FFFF are skipped, RSTn is a no-op).  The
// host keeps its plan only where that marker is EOI; every other file takes the full host walk.  One workgroup per 4 KiB,
// grid (chunks of the longest segment, segments); segs[i] = {byte offset in `data`, length}.
// Zero-copy ingest of segments that lie scattered in page-locked host memory (JPGPU_UPLOAD_PINNED without the arena layout):
// the device pulls them over the host link itself -- one workgroup per 32 KiB piece, 16 bytes per lane and step -- instead of one
// hipMemcpyAsync per segment (~33 us of host time each: 1 MiB files arrive at 30 GB/s from one thread and at 13 GB/s when
// two or three contexts issue them side by side; profiles/r03a_multi_slots.jsonl).  Page-locked memory is mapped into the
// device's address space under the host's own addresses (hipHostMalloc; hipHostRegister with the Mapped flag).
__global__ __launch_bounds__(256) void gather_pinned_kernel(const GatherPiece *__restrict__ pieces, uint8_t *__restrict__ dst) {
    const GatherPiece pc = pieces[blockIdx.x];
    const uint8_t *src = reinterpret_cast<const uint8_t *>(pc.src);
    uint8_t *d = dst + pc.dst_off;
    const uint32_t n16 = pc.len & ~15u;
    for (uint32_t i = threadIdx.x * 16u; i < n16; i += 256u * 16u) {
        uint4 v;
        __builtin_memcpy(&v, src + i, 16);  // (the source may sit at any byte address: unaligned global loads are fine on gfx9+)
        __builtin_memcpy(d + i, &v, 16);
    }
    if (threadIdx.x < (pc.len & 15u)) d[n16 + threadIdx.x] = src[n16 + threadIdx.x];
}

__global__ __launch_bounds__(kScanThreads) void first_marker_kernel(const uint8_t *__restrict__ data, const uint2 *__restrict__ segs,
                                                                     const uint32_t *__restrict__ seg_hi, uint32_t *__restrict__ first) {
    const uint32_t seg = blockIdx.y;
    const uint64_t seg_off = (uint64_t)segs[seg].x | ((uint64_t)seg_hi[seg] << 32);
    const uint32_t len = segs[seg].y;
    const int32_t misalign = (int32_t)(seg_off & 15u);
    const int64_t off = -(int64_t)misalign + (int64_t)blockIdx.x * kChunkBytes + (int64_t)threadIdx.x * 16;
    if ((int64_t)blockIdx.x * kChunkBytes - misalign >= (int64_t)len) return;
    const ByteClass c = classify16(data + seg_off, off, len, false);
    const uint64_t has_term = __ballot(c.term != 0);  // offsets grow with the lane: the lowest lane with a marker has the first
    if (has_term == 0) return;
    const uint32_t tpos = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(off + __builtin_ctz(c.term | 0x10000u)), (int)__builtin_ctzll(has_term));
    if (lane_id() == 0) atomicMin(&first[seg], tpos);
}

// K1p: the summaries of one scan turned into what each of its chunks needs: RSTs / udata bytes of the chunks BEFORE it (in
// place of its own counts) and the earliest terminator of the whole scan (pad).  One workgroup per scan job.
__global__ __launch_bounds__(kScanThreads) void marker_prefix_kernel(const DevScan *__restrict__ scans, ChunkSum *__restrict__ sums) {
    const DevScan &s = scans[blockIdx.x];
    const uint32_t n = s.n_chunks, tid = threadIdx.x;
    if (n == 0) return;
    ChunkSum *cs = sums + s.chunk_off;
    __shared__ uint32_t sh_a[kScanThreads / 64], sh_b[kScanThreads / 64];
    __shared__ uint32_t sh_term;
    if (tid == 0) sh_term = kInf;
    __syncthreads();
    uint32_t rst_run = 0, keep_run = 0;
    for (uint32_t base = 0; base < n; base += kScanThreads) {
        const uint32_t i = base + tid;
        ChunkSum c = {0, 0, kInf, 0};
        if (i < n) c = cs[i];
        uint32_t rst_total, keep_total;
        const uint32_t r = block_exclusive_scan(c.rst_cnt, sh_a, rst_total);
        const uint32_t k = block_exclusive_scan(c.keep_cnt, sh_b, keep_total);
        const uint32_t t = wave_reduce_min(c.first_term);
        if (lane_id() == 0 && t != kInf) atomicMin(&sh_term, t);
        if (i < n) {
            cs[i].rst_cnt = rst_run + r;
            cs[i].keep_cnt = keep_run + k;
        }
        rst_run += rst_total;
        keep_run += keep_total;
        __syncthreads();  // sh_a / sh_b are reused by the next round
    }
    const uint32_t term = sh_term;
    for (uint32_t i = tid; i < n; i += kScanThreads) cs[i].pad = term;
}

// K1b: every chunk takes its position in the scan from K1p and writes its part of ends[] / ends_u[] / udata; the chunk
// that holds the closing entry also writes the scan status.
struct ChunkRef {
    uint32_t scan, chunk;
};
// `mine`: RSTs / udata bytes of the chunks BEFORE this one, and (pad) the earliest terminator among this chunk and the ones before it
// (the three-kernel form hands over the earliest of the whole scan: a terminator in a LATER chunk changes nothing here).
// `c`: the chunk's bytes, 16 per lane, classified (classify16 of what load16 returned for them).
__device__ __forceinline__ void marker_write_chunk(const uint8_t *__restrict__ data, const DevScan &s, const ChunkRef wk, const ChunkSum mine,
                                                   ByteClass c, uint32_t *__restrict__ ends, DevScanStatus *__restrict__ status,
                                                   uint8_t *__restrict__ udata, uint32_t *__restrict__ ends_u) {
    const uint8_t *p = data + s.data_off;
    uint8_t *up = udata + s.data_off;
    const uint32_t len = s.data_len;
    const uint32_t cap = s.n_intervals;
    uint32_t *out = ends + s.ends_off;
    uint32_t *out_u = ends_u + s.ends_off;
    const uint32_t tid = threadIdx.x;

    __shared__ uint32_t sh_a[kScanThreads / 64], sh_b[kScanThreads / 64];
    __shared__ __attribute__((aligned(16))) uint8_t sh_tile[kChunkBytes + 2 * kScanThreads + 16];

    const int32_t misalign = (int32_t)(s.data_off & 15u);
    const int64_t chunk_first = -(int64_t)misalign + (int64_t)wk.chunk * kChunkBytes;
    const int64_t off = chunk_first + (int64_t)tid * 16;
    const uint32_t rst_base = mine.rst_cnt, ubase = mine.keep_cnt, term = mine.pad;
    if (term != kInf && (int64_t)term < chunk_first) return;  // the scan's data ended in an earlier chunk
    if (rst_base >= cap && cap > 0) return;                    // every interval was closed in an earlier chunk
    if (cap == 0) return;

    const bool term_here = term != kInf && (int64_t)term < chunk_first + (int64_t)kChunkBytes;
    if (term_here) {  // RST markers behind the terminator do not count
#pragma unroll
        for (int j = 0; j < 16; j++)
            if ((int64_t)(off + j) > (int64_t)term) c.rst &= ~(1u << j);
    }
    // One packed scan serves both prefixes in every chunk but the one that closes the scan: RSTs in the low half, udata
    // bytes (kept bytes + one more per entry) in the high half -- a chunk holds at most 2048 of the first and 6144 of the second.
    uint32_t packed_total;
    const uint32_t packed_excl = block_exclusive_scan((uint32_t)__builtin_popcount(c.rst) |
                                                      ((uint32_t)(__builtin_popcount(c.keep) + __builtin_popcount(c.rst)) << 16), sh_a, packed_total);
    const uint32_t rst_excl = packed_excl & 0xFFFFu, rst_total = packed_total & 0xFFFFu;
    // closing entry of the scan, if it lies in this chunk: the cap-th RST, else the terminator
    const bool cap_here = rst_base + rst_total >= cap;
    uint32_t last_pos = kInf;  // raw position of the closing entry when it is in this chunk
    uint32_t markers = c.rst;  // entries this lane owns: every one becomes FF FF in udata
    uint32_t keep_excl = packed_excl >> 16, keep_total = packed_total >> 16;
    if (cap_here || term_here) {  // workgroup-uniform: one chunk per scan
        __shared__ uint32_t sh_last;
        if (tid == 0) sh_last = kInf;
        __syncthreads();
        {
            uint32_t idx = rst_base + rst_excl;
            uint32_t m = c.rst;
            while (m) {
                const int j = __builtin_ctz(m);
                m &= m - 1;
                if (idx == cap - 1) sh_last = (uint32_t)(off + j);
                idx++;
            }
        }
        __syncthreads();
        if (cap_here) last_pos = sh_last;
        else last_pos = term;
        // bytes behind the closing entry are not copied; markers behind it are not entries
#pragma unroll
        for (int j = 0; j < 16; j++)
            if ((int64_t)(off + j) > (int64_t)last_pos) {
                c.keep &= ~(1u << j);
                c.rst &= ~(1u << j);
                c.term &= ~(1u << j);
            }
        markers = c.rst;
        if (term_here && !cap_here && (int64_t)term >= off && (int64_t)term < off + 16) markers |= 1u << (uint32_t)((int64_t)term - off);
        keep_excl = block_exclusive_scan(__builtin_popcount(c.keep) + __builtin_popcount(markers), sh_b, keep_total);
    }
#if defined(JPGPU_K1_PRICE)
    // (pricing build only: the compaction loop left out -- wrong output, the time of everything else)
    *reinterpret_cast<uint4 *>(sh_tile + tid * 16) = uint4{c.w[0], c.w[1], c.w[2], c.w[3] ^ keep_excl ^ rst_excl};
    if (false)
#endif
    {
        uint32_t dst = keep_excl;  // chunk-relative udata position
        uint32_t idx = rst_base + rst_excl;
#pragma unroll
        for (int j = 0; j < 16; j++) {
            if (c.keep & (1u << j)) {
                const uint32_t b = (c.w[j >> 2] >> ((j & 3) * 8)) & 0xFF;
                sh_tile[dst] = (uint8_t)b;
                if (markers & (1u << j)) {
                    // interval end: FF FF in udata
                    if (c.rst & (1u << j)) {
                        if (idx < cap) {
                            out[idx] = (uint32_t)(off + j);
                            out_u[idx] = ubase + dst;
                        }
                        idx++;
                    } else {
                        out[rst_base + rst_total] = (uint32_t)(off + j);  // the terminator closes interval number (RSTs before it)
                        out_u[rst_base + rst_total] = ubase + dst;
                    }
                    sh_tile[dst + 1] = 0xFF;
                    dst += 2;
                } else {
                    dst += 1;
                }
            }
        }
    }
    __syncthreads();
    {
        const uint32_t c0 = tid * 16;
        for (uint32_t cbeg = c0; cbeg < keep_total; cbeg += kScanThreads * 16) {
            if (cbeg + 16 <= keep_total) {
                const uint4 v = *reinterpret_cast<const uint4 *>(sh_tile + cbeg);
#if defined(JPGPU_K1_PRICE) && JPGPU_K1_PRICE == 3
                *reinterpret_cast<uint4 *>(up + ((ubase + cbeg) & ~15u)) = v;  // (pricing build only: aligned)
#else
                __builtin_memcpy(up + ubase + cbeg, &v, 16);  // unaligned 16-byte global store
#endif
            } else {
                for (uint32_t i = cbeg; i < keep_total; i++) up[ubase + i] = sh_tile[i];
            }
        }
    }
    // scan status: written by the chunk that holds the closing entry, or by the last chunk when the data ran out
    const bool ran_out = (term == kInf) && !cap_here && (wk.chunk == s.n_chunks - 1);
    if (tid == 0 && (last_pos != kInf || ran_out)) {
        DevScanStatus st;
        uint32_t found;
        uint32_t lp;
        if (cap_here) {
            found = cap;
            lp = last_pos;
        } else if (term_here) {
            found = rst_base + rst_total + 1;
            lp = term;
        } else {
            // data ran out without a terminating marker: pseudo entry at the end of the data; pad udata with ones
            found = rst_base + rst_total + 1;
            lp = len;
            out[rst_base + rst_total] = len;
            out_u[rst_base + rst_total] = ubase + keep_total;
            up[ubase + keep_total] = 0xFF;
            up[ubase + keep_total + 1] = 0xFF;
        }
        st.n_ends = found;
        st.terminator = (lp + 1 < len) ? p[lp + 1] : 0;
        st.first_error = kNoError;
        const uint64_t covered = (uint64_t)found * (s.dri ? s.dri : s.total_mcus);
        st.decoded_mcus = covered < s.total_mcus ? (uint32_t)covered : s.total_mcus;
        st.end_pos = lp;
        st.pad[0] = ubase + keep_total;
        st.pad[1] = st.pad[2] = 0;
        status[wk.scan] = st;
    }
}

__global__ __launch_bounds__(kScanThreads) void marker_write_kernel(const uint8_t *__restrict__ data, const DevScan *__restrict__ scans,
                                                                     const ChunkWork *__restrict__ work, const ChunkSum *__restrict__ sums,
                                                                     uint32_t *__restrict__ ends, DevScanStatus *__restrict__ status,
                                                                     uint8_t *__restrict__ udata, uint32_t *__restrict__ ends_u) {
    const ChunkWork wk0 = work[blockIdx.x];
    const DevScan &s = scans[wk0.scan];
    // kMarkerChunksPerWg consecutive chunks per workgroup (see marker_count_kernel)
    for (uint32_t chunk = wk0.chunk; chunk < wk0.chunk + kMarkerChunksPerWg && chunk < s.n_chunks; chunk++) {
        __syncthreads();  // the shared tile / scan scratch of the previous chunk is free
        // the tile's bytes are asked for BEFORE the chunk's summary is waited for: two load latencies side by side instead of in a row
        const int64_t off = -(int64_t)(s.data_off & 15u) + (int64_t)chunk * kChunkBytes + (int64_t)threadIdx.x * 16;
        const Raw16 raw = load16(data + s.data_off, off, s.data_len);
        const ChunkSum mine = sums[s.chunk_off + chunk];
        marker_write_chunk(data, s, ChunkRef{wk0.scan, chunk}, mine, classify16(raw, off, s.data_len, s.dri == 0), ends, status, udata, ends_u);
    }
}

// ------------------------------------------------------------------------------------------------
// K1 in ONE pass (round 5).  The three-kernel form reads the entropy segments twice: marker_count_kernel for the chunk summaries,
// marker_prefix_kernel for their running sums, marker_write_kernel -- which classifies the same bytes again -- for the writing.
// Here a workgroup takes a GROUP of consecutive chunks of a scan, classifies their tiles once, publishes the group's summary, finds
// the running sums of the groups in front of it by looking BACK (decoupled look-back: Merrill & Garland) and writes.  Per group two
// descriptors of three 8-byte granules {value, tag}: AGGREGATE (this group alone: RSTs, udata bytes, first terminator) and
// INCLUSIVE (everything up to and including it).  A granule is one agent-scope 8-byte store and is read by one agent-scope 8-byte
// load: it arrives whole or not at all, no fence on either side (the per-XCD L2s are not coherent: sc1 traffic goes past them, and a
// release / acquire pair would write back / invalidate the whole L2 at every publish and poll); tag = a number no earlier decode of
// the batch object has used, so nothing is cleared between decodes.  Wave 0 of the workgroup looks at 64 predecessors at a time: the
// nearest INCLUSIVE one ends the walk, AGGREGATEs in front of it are summed, a predecessor that has published neither is waited for.
// No workgroup waits for one that has not started: the group a workgroup takes is a TICKET into the order list (one atomic counter),
// in which a scan's groups stand in the scan's order, so every group in front of it has been taken by a workgroup that is running
// or done.  The wait is bounded all the same: a workgroup that runs out of patience counts the chunks in front of its group itself.
// ------------------------------------------------------------------------------------------------
constexpr uint32_t kK1DescWords = 8;  // uint64 per chunk: [0..2] aggregate {rst, keep, term}, [3..5] inclusive, [6..7] unused (64-byte stride)
__device__ __forceinline__ void k1_publish(unsigned long long *d, uint32_t rst, uint32_t keep, uint32_t term, uint32_t tag) {
    __hip_atomic_store(d + 0, ((unsigned long long)tag << 32) | rst, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(d + 1, ((unsigned long long)tag << 32) | keep, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(d + 2, ((unsigned long long)tag << 32) | term, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool k1_read(const unsigned long long *d, uint32_t tag, uint32_t &rst, uint32_t &keep, uint32_t &term) {
    const unsigned long long a = __hip_atomic_load(d + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long b = __hip_atomic_load(d + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long c = __hip_atomic_load(d + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    rst = (uint32_t)a;
    keep = (uint32_t)b;
    term = (uint32_t)c;
    return (uint32_t)(a >> 32) == tag && (uint32_t)(b >> 32) == tag && (uint32_t)(c >> 32) == tag;
}

// Second form (round 5, behind the encoder's bits_emit_kernel): the first one gave every 4 KiB chunk a workgroup of its own that
// lived ~2 us, polled for about as long, and classified its tile twice (for the summary, and again in the writer): 2.39 ms against
// the three kernels' 1.03.  Here a workgroup takes kK1Group consecutive chunks of a scan (all their loads in flight together),
// keeps their classification in registers for the writer -- so the bytes are classified ONCE -- publishes ONE summary for the group,
// and the groups are handed out (round 5: a global ticket into `order`; round 6: the workgroup's own index) by
// (place inside the scan, scan): the workgroups that run side by side belong to as many scans as the batch has, a scan's chain
// is a record or two long at any time, and a scan's groups still start in the scan's order.
// Round 6 priced the kernel (-DJPGPU_K1_PRICE=1..4, profiles/r06_k1_price.txt): neither the two byte loops nor the look-back's
// wait bound it, the life of its 65 536 workgroups per 1024 x 4K does -- hence the group tile below, five waves per SIMD, no ticket.
constexpr uint32_t kK1Group = kMarkerGroupChunks;
#ifndef JPGPU_K1_WAVES
#define JPGPU_K1_WAVES 5
#endif
__global__ __launch_bounds__(kScanThreads, JPGPU_K1_WAVES) void marker_onepass_kernel(const uint8_t *__restrict__ data, const DevScan *__restrict__ scans,
                                                                       const ChunkWork *__restrict__ order, uint32_t n_groups,
                                                                       unsigned long long *__restrict__ desc, uint32_t *__restrict__ tickets, uint32_t epoch,
                                                                       uint32_t tag, uint32_t spin_budget,
                                                                       uint32_t *__restrict__ host_giveup, uint32_t *__restrict__ ends,
                                                                       DevScanStatus *__restrict__ status, uint8_t *__restrict__ udata,
                                                                       uint32_t *__restrict__ ends_u) {
    __shared__ uint32_t sh_cnt[kK1Group][kScanThreads / 64], sh_term[kK1Group][kScanThreads / 64], sh_mine[3], sh_ok;
#if defined(JPGPU_K1_TICKETS)
    __shared__ uint32_t sh_ticket;
#endif
    const uint32_t tid = threadIdx.x;
#if !defined(JPGPU_K1_NO_GROUP_TILE)
    // the tile a group in the middle of its scan assembles its udata bytes in (below): 16 bytes in front of them, room behind them for
    // the last lane's fifth word.  Its lanes OR their bytes into it, so it starts as zeros -- written here, in front of everything the
    // kernel waits for; the barriers behind the summary and the look-back are in between.
    constexpr uint32_t kTileFront = 16, kTileBytes = kTileFront + kK1Group * kChunkBytes + 48;
    __shared__ __attribute__((aligned(16))) uint32_t sh_group_w[kTileBytes / 4];
    uint8_t *const sh_group = reinterpret_cast<uint8_t *>(sh_group_w);
#if !defined(JPGPU_K1_TILE_BYTE_LOOP)
#pragma unroll
    for (uint32_t i = 0; i < kK1Group; i++) *reinterpret_cast<uint4 *>(sh_group + i * kChunkBytes + tid * 16) = uint4{0, 0, 0, 0};
    if (tid < (kTileBytes - kK1Group * kChunkBytes) / 16) *reinterpret_cast<uint4 *>(sh_group + kK1Group * kChunkBytes + tid * 16) = uint4{0, 0, 0, 0};
#endif
#endif
    // (-DJPGPU_K1_TICKETS: a ticket into the order list; tickets[0] has counted n_groups per earlier decode of this upload)
#if !defined(JPGPU_K1_TICKETS)
    // The group: the workgroup's own index into the order list.  A group only ever waits for groups in FRONT of it in that list, and the
    // dispatcher hands workgroups out in index order (round-robin over the XCDs, in order on each): the lowest unfinished group is
    // always resident, so somebody always makes progress.  That is an observation about this part, not a promise of the programming
    // model -- which is why the wait below is bounded and a group that runs out of patience counts for itself: were the order ever
    // different, the index would be slow, never wrong or stuck (jpgpu_batch_marker_fallbacks says whether it happened; the tests
    // assert 0).  The promise costs a device-scope atomic on ONE address per workgroup and a barrier in front of the first load:
    // 0.84 -> 0.76 ms per 1024 x 4K, 0.50 -> 0.42 at 1080p Q90, 0.82 -> 0.68 DRI = 0 (-DJPGPU_K1_TICKETS, profiles/r06_k1_group_tile_ab.txt).
    const ChunkWork wk = order[blockIdx.x];
#else
    if (tid == 0) sh_ticket = atomicAdd(&tickets[0], 1u) - epoch * n_groups;
    __syncthreads();
    if (sh_ticket >= n_groups) return;  // (cannot happen: one workgroup per listed group)
    const ChunkWork wk = order[sh_ticket];
#endif
    const uint32_t scan = wk.scan, first_chunk = wk.chunk;
    const DevScan &s = scans[scan];
    const uint32_t len = s.data_len;
    const bool any_marker = s.dri == 0;
    // ---- the group's tiles: every load issued before the first is classified (a chunk behind the scan's last holds nothing)
    Raw16 raw[kK1Group];
    int64_t off[kK1Group];
#pragma unroll
    for (uint32_t i = 0; i < kK1Group; i++) {
        off[i] = -(int64_t)(s.data_off & 15u) + (int64_t)(first_chunk + i) * kChunkBytes + (int64_t)tid * 16;
        raw[i] = load16(data + s.data_off, off[i], len);
    }
    ByteClass c[kK1Group];
#pragma unroll
    for (uint32_t i = 0; i < kK1Group; i++) {
        c[i] = classify16(raw[i], off[i], len, any_marker);
#if defined(JPGPU_K1_PRICE) && JPGPU_K1_PRICE >= 2
        // (pricing build only: no classification either)
        c[i].keep = off[i] < (int64_t)len && off[i] + 16 > 0 ? 0xFFFFu : 0u;
        c[i].rst = (c[i].w[1] & c[i].w[2] & 0x10101u) == 0x10101u ? 1u : 0u;
        c[i].term = 0;
#endif
        // the chunk's summary (what marker_count_kernel computes)
        const uint32_t own = (uint32_t)__builtin_popcount(c[i].rst) | ((uint32_t)(__builtin_popcount(c[i].keep) + __builtin_popcount(c[i].rst | c[i].term)) << 16);
        const uint32_t cnt = (uint32_t)__builtin_amdgcn_readlane((int)wave_inclusive_scan(own), 63);
        const uint64_t has_term = __ballot(c[i].term != 0);
        uint32_t tpos = kInf;
        if (has_term != 0) tpos = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(off[i] + __builtin_ctz(c[i].term | 0x10000u)), (int)__builtin_ctzll(has_term));
        if (lane_id() == 0) {
            sh_cnt[i][tid >> 6] = cnt;
            sh_term[i][tid >> 6] = tpos;
        }
    }
    __syncthreads();
    // per chunk: RSTs / udata bytes of the group's chunks in front of it, earliest terminator up to and including it
    uint32_t pre_rst[kK1Group], pre_keep[kK1Group], upto_term[kK1Group];
    uint32_t a_rst = 0, a_keep = 0, a_term = kInf;
#pragma unroll
    for (uint32_t i = 0; i < kK1Group; i++) {
        uint32_t total = 0, t = kInf;
#pragma unroll
        for (int w = 0; w < kScanThreads / 64; w++) {
            total += sh_cnt[i][w];
            t = sh_term[i][w] < t ? sh_term[i][w] : t;
        }
        pre_rst[i] = a_rst;
        pre_keep[i] = a_keep;
        a_rst += total & 0xFFFFu;
        a_keep += total >> 16;
        a_term = t < a_term ? t : a_term;
        upto_term[i] = a_term;
    }
    const uint32_t group = first_chunk / kK1Group;
    unsigned long long *my = desc + (size_t)(s.chunk_off + first_chunk) * kK1DescWords;  // (the record of the group's first chunk)
    // ---- publish, look back, publish (wave 0; the others wait at the barrier below)
    if (tid < 64) {
        uint32_t x_rst = 0, x_keep = 0, x_term = kInf;  // exclusive: the groups in front of this one
        bool ok = true;
        if (group == 0) {
            if (tid == 0) k1_publish(my + 3, a_rst, a_keep, a_term, tag);
        } else {
            if (tid == 0) k1_publish(my, a_rst, a_keep, a_term, tag);
            int64_t window_first = (int64_t)group - 1;  // lane l looks at group window_first - l
            uint32_t polls = 0;
            for (;;) {
                const int64_t pg = window_first - (int64_t)tid;
                uint32_t r = 0, k = 0, t = kInf;
                uint32_t state = 3;  // 0 nothing yet, 1 aggregate, 2 inclusive, 3 in front of the scan's first group
                if (pg >= 0) {
                    const unsigned long long *pd = desc + (size_t)(s.chunk_off + (uint32_t)pg * kK1Group) * kK1DescWords;
                    if (k1_read(pd + 3, tag, r, k, t)) state = 2;
                    else if (k1_read(pd, tag, r, k, t)) state = 1;
                    else state = 0;
                }
                const uint64_t inc = __ballot(state == 2), none = __ballot(state == 0);
                // the lanes that count: the ones in front of (and including) the nearest inclusive predecessor
                const uint32_t stop = inc != 0 ? (uint32_t)__builtin_ctzll(inc) : 63u;
                const uint64_t upto = stop >= 63u ? ~0ull : ((2ull << stop) - 1ull);
#if defined(JPGPU_K1_PRICE) && JPGPU_K1_PRICE == 4
                if (false) {  // (pricing build only: nobody waits)
#else
                if ((none & upto) != 0) {  // somebody there has not published yet
#endif
                    if (++polls > spin_budget) {
                        ok = false;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(4);
                    continue;
                }
                const bool counts = tid <= stop && state != 3;
                x_rst += wave_sum(counts ? r : 0u);
                x_keep += wave_sum(counts ? k : 0u);
                x_term = min(x_term, wave_reduce_min(counts ? t : kInf));
                if (inc != 0 || window_first - 63 <= 0) break;  // an inclusive prefix, or the scan's first group, was inside the window
                window_first -= 64;
            }
            if (ok && tid == 0) k1_publish(my + 3, x_rst + a_rst, x_keep + a_keep, min(x_term, a_term), tag);
        }
        if (tid == 0) {
            sh_mine[0] = x_rst;
            sh_mine[1] = x_keep;
            sh_mine[2] = x_term;
            sh_ok = ok ? 1u : 0u;
        }
    }
    __syncthreads();
    uint32_t x_rst = sh_mine[0], x_keep = sh_mine[1], x_term = sh_mine[2];
    if (sh_ok == 0) {
        // Out of patience (it cannot happen while the groups are handed out by ticket; the bound is there so that nothing can hang,
        // and the tests set it to zero): the group counts the chunks in front of it ITSELF -- slow, and dependent on nobody -- and
        // goes on as if it had been told.  (The first form left here and had the host issue the three kernels: the kernels
        // enqueued behind this one ran on a half-written index in the meantime.)
        x_rst = 0, x_keep = 0, x_term = kInf;
        for (uint32_t ch = 0; ch < first_chunk; ch++) {
            const int64_t o = -(int64_t)(s.data_off & 15u) + (int64_t)ch * kChunkBytes + (int64_t)tid * 16;
            const ByteClass cc = classify16(load16(data + s.data_off, o, len), o, len, any_marker);
            const uint32_t cnt = wave_sum((uint32_t)__builtin_popcount(cc.rst) | ((uint32_t)(__builtin_popcount(cc.keep) + __builtin_popcount(cc.rst | cc.term)) << 16));
            const uint64_t has_term = __ballot(cc.term != 0);
            uint32_t tpos = kInf;
            if (has_term != 0) tpos = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(o + __builtin_ctz(cc.term | 0x10000u)), (int)__builtin_ctzll(has_term));
            __syncthreads();
            if (lane_id() == 0) {
                sh_cnt[0][tid >> 6] = cnt;
                sh_term[0][tid >> 6] = tpos;
            }
            __syncthreads();
#pragma unroll
            for (int w = 0; w < kScanThreads / 64; w++) {
                x_rst += sh_cnt[0][w] & 0xFFFFu;
                x_keep += sh_cnt[0][w] >> 16;
                x_term = sh_term[0][w] < x_term ? sh_term[0][w] : x_term;
            }
        }
        if (tid == 0) {
            k1_publish(my + 3, x_rst + a_rst, x_keep + a_keep, min(x_term, a_term), tag);
            __hip_atomic_store(host_giveup, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);  // (page-locked host memory: a count for the tests)
        }
    }
#if !defined(JPGPU_K1_NO_GROUP_TILE)
    // ---- a group in the middle of its scan (nearly all of them): every chunk interior, no terminator up to and including it, the scan's
    // last interval not closed inside it.  Nothing of the per-chunk writer's special cases can apply, the places of its bytes follow from
    // what the summary left behind (the waves' counts in sh_cnt, the lane's place inside its wave): ONE tile of the group's udata bytes,
    // one barrier, one copy -- the per-chunk writer below pays two block-wide scans and four barriers per chunk.
    {
        const int64_t g_first = -(int64_t)(s.data_off & 15u) + (int64_t)first_chunk * kChunkBytes;
        const bool plain = g_first >= 1 && g_first + (int64_t)(kK1Group * kChunkBytes) + 1 <= (int64_t)len && x_term == kInf && a_term == kInf &&
                           x_rst + a_rst < s.n_intervals && sh_ok != 0;  // (a group that gave up waiting has used sh_cnt[0] as its scratch)
        if (plain) {  // (uniform)
            uint32_t *out = ends + s.ends_off, *out_u = ends_u + s.ends_off;
            uint8_t *up = udata + s.data_off;
            const uint32_t wave = tid >> 6;
#pragma unroll
            for (uint32_t i = 0; i < kK1Group; i++) {
                uint32_t before = 0;  // the chunk's waves in front of this one
#pragma unroll
                for (uint32_t w = 0; w < kScanThreads / 64; w++)
                    if (w < wave) before += sh_cnt[i][w];
                // (the lane's place inside its wave: the summary's scan once more -- four registers less across the look-back)
                const uint32_t own = (uint32_t)__builtin_popcount(c[i].rst) | ((uint32_t)(__builtin_popcount(c[i].keep) + __builtin_popcount(c[i].rst)) << 16);
                const uint32_t packed_excl = before + wave_inclusive_scan(own) - own;
                uint32_t dst = pre_keep[i] + (packed_excl >> 16);  // place in the group's udata bytes
                uint32_t idx = x_rst + pre_rst[i] + (packed_excl & 0xFFFFu);
#if defined(JPGPU_K1_PRICE)
                // (pricing build only: the compaction left out -- wrong output, the time of everything else)
                *reinterpret_cast<uint4 *>(sh_group + kTileFront + i * kChunkBytes + tid * 16) = uint4{c[i].w[0], c[i].w[1], c[i].w[2], c[i].w[3] ^ dst ^ idx};
#elif defined(JPGPU_K1_TILE_BYTE_LOOP)
                // (the first form of the tile's writer: byte by byte, 16 single-byte LDS stores and their branches per lane)
#pragma unroll
                for (int j = 0; j < 16; j++) {
                    if (c[i].keep & (1u << j)) {
                        sh_group[kTileFront + dst] = (uint8_t)((c[i].w[j >> 2] >> ((j & 3) * 8)) & 0xFF);
                        if (c[i].rst & (1u << j)) {  // interval end: FF FF in udata
                            out[idx] = (uint32_t)(off[i] + j);
                            out_u[idx] = x_keep + dst;
                            idx++;
                            sh_group[kTileFront + dst + 1] = 0xFF;
                            dst += 2;
                        } else {
                            dst += 1;
                        }
                    }
                }
#else
                // The lane's 16 bytes as four words.  What udata does not take -- a stuffed zero, a fill FF, the code byte of a marker whose
                // FF is the lane in front's last byte -- is rare per lane (one lane in sixteen) and present in nearly every wave: taken out
                // of the words one byte per trip, lowest first, the bytes above it moving down (the wave makes as many trips as its
                // worst lane has such bytes: one or two).  A marker's code byte inside the lane keeps its place: the second FF of the
                // pair is ORed over it below.  Then the words go into the tile at the lane's byte offset: five aligned words, the outer
                // two shared with the neighbours, all by OR into zeros -- no single-byte stores, no branch per byte.
                const uint32_t rst = c[i].rst;
                const uint32_t kept = c[i].keep | ((rst << 1) & 0xFFFFu);
                uint32_t drop = ~kept & 0xFFFFu;
                uint32_t w0 = c[i].w[0], w1 = c[i].w[1], w2 = c[i].w[2], w3 = c[i].w[3];
                while (__ballot(drop != 0) != 0) {  // (uniform)
                    if (drop != 0) {
                        const int32_t k8 = 8 * (int32_t)__builtin_ctz(drop);  // bit position of the byte that goes
                        const uint32_t s0 = __builtin_amdgcn_alignbit(w1, w0, 8), s1 = __builtin_amdgcn_alignbit(w2, w1, 8),
                                       s2 = __builtin_amdgcn_alignbit(w3, w2, 8), s3 = w3 >> 8;
                        // bits of word d at and above the byte: they take the word moved down one byte
                        const uint32_t h0 = (uint32_t)(~0ull << (uint32_t)min(max(k8, 0), 32)), h1 = (uint32_t)(~0ull << (uint32_t)min(max(k8 - 32, 0), 32)),
                                       h2 = (uint32_t)(~0ull << (uint32_t)min(max(k8 - 64, 0), 32)), h3 = (uint32_t)(~0ull << (uint32_t)min(max(k8 - 96, 0), 32));
                        w0 = (w0 & ~h0) | (s0 & h0);
                        w1 = (w1 & ~h1) | (s1 & h1);
                        w2 = (w2 & ~h2) | (s2 & h2);
                        w3 = (w3 & ~h3) | (s3 & h3);
                        drop = (drop & (drop - 1u)) >> 1;  // (the bytes above it have moved down one place)
                    }
                }
                {
                    // byte offset dst of the tile's data = byte t + 1 counted from the tile's first word: the words move up by
                    // ((t & 3) + 1) bytes from the word t lies in (1 .. 4: a funnel shift by 24, 16, 8 or 0 bits takes the right halves)
                    const uint32_t t = dst + (kTileFront - 1), r = 24u - 8u * (t & 3u);
                    uint32_t *q = sh_group_w + (t >> 2);
                    __hip_atomic_fetch_or(q + 0, __builtin_amdgcn_alignbit(w0, 0u, r), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_fetch_or(q + 1, __builtin_amdgcn_alignbit(w1, w0, r), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_fetch_or(q + 2, __builtin_amdgcn_alignbit(w2, w1, r), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_fetch_or(q + 3, __builtin_amdgcn_alignbit(w3, w2, r), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_fetch_or(q + 4, __builtin_amdgcn_alignbit(0u, w3, r), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                // interval ends: the index entries, and FF over the place behind the marker's FF (its code byte, or -- the marker in the
                // lane's last byte -- the place the lane's count has reserved behind its bytes)
                uint32_t m = rst;
                while (__ballot(m != 0) != 0) {  // (uniform)
                    if (m != 0) {
                        const uint32_t j = (uint32_t)__builtin_ctz(m);
                        m &= m - 1u;
                        const uint32_t pos = dst + (uint32_t)__builtin_popcount(kept & ((1u << j) - 1u));
                        out[idx] = (uint32_t)(off[i] + j);
                        out_u[idx] = x_keep + pos;
                        idx++;
                        const uint32_t a = pos + 1u + kTileFront;
                        __hip_atomic_fetch_or(sh_group_w + (a >> 2), 0xFFu << (8u * (a & 3u)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                }
#endif
            }
            __syncthreads();
            for (uint32_t cbeg = tid * 16; cbeg < a_keep; cbeg += kScanThreads * 16) {
                if (cbeg + 16 <= a_keep) {
                    const uint4 v = *reinterpret_cast<const uint4 *>(sh_group + kTileFront + cbeg);
                    __builtin_memcpy(up + x_keep + cbeg, &v, 16);  // unaligned 16-byte global store
                } else {
                    for (uint32_t k = cbeg; k < a_keep; k++) up[x_keep + k] = sh_group[kTileFront + k];
                }
            }
            return;
        }
    }
#endif
#pragma unroll
    for (uint32_t i = 0; i < kK1Group; i++) {
        if (first_chunk + i >= s.n_chunks) break;  // (uniform)
        __syncthreads();  // the shared tile / scan scratch of the previous chunk is free
        ChunkSum mine;
        mine.rst_cnt = x_rst + pre_rst[i];
        mine.keep_cnt = x_keep + pre_keep[i];
        mine.first_term = 0;
        mine.pad = x_term < upto_term[i] ? x_term : upto_term[i];
        marker_write_chunk(data, s, ChunkRef{scan, first_chunk + i}, mine, c[i], ends, status, udata, ends_u);
    }
}


hipError_t launch_gather_pinned(hipStream_t stream, const GatherPiece *pieces, int n_pieces, uint8_t *dst) {
    if (n_pieces <= 0) return hipSuccess;
    hipLaunchKernelGGL(gather_pinned_kernel, dim3((uint32_t)n_pieces), dim3(256), 0, stream, pieces, dst);
    return hipGetLastError();
}

hipError_t launch_first_marker(hipStream_t stream, const uint8_t *data, const void *segs, const uint32_t *seg_hi, int n_segs,
                               uint32_t max_len, uint32_t *first) {
    if (n_segs <= 0) return hipSuccess;
    const uint32_t chunks = (max_len + 15u + kChunkBytes - 1) / kChunkBytes + 1;
    for (int base = 0; base < n_segs; base += 65535) {  // grid.y limit
        const int n = n_segs - base < 65535 ? n_segs - base : 65535;
        hipLaunchKernelGGL(first_marker_kernel, dim3(chunks, n), dim3(kScanThreads), 0, stream, data, (const uint2 *)segs + base, seg_hi + base,
                           first + base);
    }
    return hipGetLastError();
}

hipError_t launch_marker_onepass(hipStream_t stream, const uint8_t *data, const DevScan *scans, const ChunkWork *order, int n_groups,
                                 void *desc, uint32_t *tickets, uint32_t epoch, uint32_t tag, uint32_t spin_budget,
                                 uint32_t *host_giveup, uint32_t *ends, DevScanStatus *status, uint8_t *udata, uint32_t *ends_u) {
    if (n_groups <= 0) return hipSuccess;
    hipLaunchKernelGGL(marker_onepass_kernel, dim3(n_groups), dim3(kScanThreads), 0, stream, data, scans, order, (uint32_t)n_groups, (unsigned long long *)desc,
                       tickets, epoch, tag, spin_budget, host_giveup, ends, status, udata, ends_u);
    return hipGetLastError();
}

hipError_t launch_marker_index(hipStream_t stream, const uint8_t *data, const DevScan *scans, int n_scans, const ChunkWork *work,
                               int n_chunks, ChunkSum *sums, uint32_t *ends, DevScanStatus *status, uint8_t *udata, uint32_t *ends_u) {
    if (n_chunks <= 0) return hipSuccess;
    hipLaunchKernelGGL(marker_count_kernel, dim3((n_chunks + kCountChunksPerWg - 1) / kCountChunksPerWg), dim3(kScanThreads), 0, stream, data,
                       scans, work, (uint32_t)n_chunks, sums);
    hipLaunchKernelGGL(marker_prefix_kernel, dim3(n_scans), dim3(kScanThreads), 0, stream, scans, sums);
    hipLaunchKernelGGL(marker_write_kernel, dim3(n_chunks), dim3(kScanThreads), 0, stream, data, scans, work, sums, ends, status, udata,
                       ends_u);
    return hipGetLastError();
}

}  // namespace jpgpu
