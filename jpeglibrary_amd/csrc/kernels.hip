// jpeglibrary_amd/csrc/kernels.hip -- hand-written HIP kernels for gfx950 (MI355X, CDNA4, wave64).
//
// The reference's hot loop (ScanDecoder/JpegHuffmanBaselineScanDecoder.cs:51-177) is serial:
//   per MCU, per block: ReadBlockBaseline -> DequantizeBlockAndUnZigZag -> TransformIDCT -> ShiftDataLevel -> WriteBlock.
// Here it becomes three device stages over a whole batch of scan jobs:
//   K1 marker_index_kernel    byte-scan of the entropy segment for FF Dn / terminating marker (restart index)
//   K2 huffman_decode_kernel  one LANE per restart interval, 64 intervals per wavefront; block staged in LDS,
//                             flushed as whole 128-byte lines to the coefficient buffer (zig-zag int16, MCU order)
//   K3 idct_output_kernel     one lane per 8x8 block: dequantise + float32 IDCT (strict op order, no FMA) +
//                             round-half-even + level shift, then block output in the requested layout
//
// MUST be compiled with -ffp-contract=off: the reference's Vector4 arithmetic never fuses a*b+c
// (FastFloatingPointDCT.cs:79-185).  No fast-math.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

#include "common.h"
#include "kernels.h"
#include "encode_kernels.h"

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
__device__ __forceinline__ void marker_write_chunk(const uint8_t *__restrict__ data, const DevScan &s, const ChunkRef wk,
                                                   const ChunkSum *__restrict__ sums, uint32_t *__restrict__ ends,
                                                   DevScanStatus *__restrict__ status, uint8_t *__restrict__ udata,
                                                   uint32_t *__restrict__ ends_u) {
    const uint8_t *p = data + s.data_off;
    uint8_t *up = udata + s.data_off;
    const uint32_t len = s.data_len;
    const uint32_t cap = s.n_intervals;
    uint32_t *out = ends + s.ends_off;
    uint32_t *out_u = ends_u + s.ends_off;
    const ChunkSum *cs = sums + s.chunk_off;
    const uint32_t tid = threadIdx.x;

    __shared__ uint32_t sh_a[kScanThreads / 64], sh_b[kScanThreads / 64];
    __shared__ __attribute__((aligned(16))) uint8_t sh_tile[kChunkBytes + 2 * kScanThreads + 16];

    // the tile's bytes are asked for BEFORE the chunk's summary is waited for: two load latencies side by side instead of in a row
    const int32_t misalign = (int32_t)(s.data_off & 15u);
    const int64_t chunk_first = -(int64_t)misalign + (int64_t)wk.chunk * kChunkBytes;
    const int64_t off = chunk_first + (int64_t)tid * 16;
    const Raw16 raw = load16(p, off, len);
    // RSTs / udata bytes of the chunks before this one, earliest terminator of the whole scan (marker_prefix_kernel)
    const ChunkSum mine = cs[wk.chunk];
    const uint32_t rst_base = mine.rst_cnt, ubase = mine.keep_cnt, term = mine.pad;
    if (term != kInf && (int64_t)term < chunk_first) return;  // the scan's data ended in an earlier chunk
    if (rst_base >= cap && cap > 0) return;                    // every interval was closed in an earlier chunk
    if (cap == 0) return;

    ByteClass c = classify16(raw, off, len, s.dri == 0);
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
                __builtin_memcpy(up + ubase + cbeg, &v, 16);  // unaligned 16-byte global store
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
        marker_write_chunk(data, s, ChunkRef{wk0.scan, chunk}, sums, ends, status, udata, ends_u);
    }
}

// ------------------------------------------------------------------------------------------------
// K2: Huffman MCU decode.  One lane per restart interval.
// ------------------------------------------------------------------------------------------------

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
    if (rem >= 8) return (interval << 8) | kDetailExpectRestart;
    if (closing == 0xD9) {
        if (interval < n_intervals - 1) atomicMin(&status_out->decoded_mcus, (interval + 1) * dri_eff);
        return kNoError;
    }
    if ((closing & 0xF8) != 0xD0) return (interval << 8) | kDetailExpectRestart;
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
constexpr uint32_t kK2TabBytes = kK2L1Bytes + 2u * kK2L2Entries + 16u + kK2SmallBytes;  // == kLutPoolBytesPerTable / 2 (kernels.h)

struct K2Tab {
    const uint16_t *lut;
    const uint16_t *l2;
    const uint32_t *hdr;  // {t16, 0, 0, 0}
    const uint8_t *small;  // maxcode[18] | valoffset[20] | values[256]
};

__device__ __forceinline__ K2Tab k2_tab(const uint8_t *tabs, uint32_t slot) {
    const uint8_t *t = tabs + slot * kK2TabBytes;
    K2Tab h;
    h.lut = reinterpret_cast<const uint16_t *>(t);
    h.l2 = reinterpret_cast<const uint16_t *>(t + kK2L1Bytes);
    h.hdr = reinterpret_cast<const uint32_t *>(t + kK2L1Bytes + 2u * kK2L2Entries);
    h.small = t + kK2L1Bytes + 2u * kK2L2Entries + 16u;
    return h;
}

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
// as ub_symbol).  Returns 0 or the failure detail; n = bits consumed, value, adv = zig-zag advance (AC, in coefficients).
__device__ __forceinline__ uint32_t k2_slow_symbol(uint8_t *ring, K2Feed &f, int32_t pm1, int32_t endpos, const K2Tab &h, bool is_dc,
                                                bool closed_by_marker, uint32_t &n, int32_t &value, uint32_t &adv) {
    const int32_t pos = pm1 + 1;
    while ((int32_t)(f.wr * 128u) < pos + 160) k2_topup(ring, f, pm1);  // always has room here
    const uint32_t hi = k2_window(ring, pm1);
    int32_t rem = endpos - pos;
    if (rem < 0) rem = 0;
    const uint32_t code16 = rem > 0 ? (hi >> 16) : 0xFFFFu;
    uint32_t e = h.lut[code16 >> (16 - kK2LutBits)];
    if (is_dc && (e & kK2BadCat) != 0) return kDetailInvalidHuffmanCode;  // categories above 16 are outside the verified envelope
    const uint32_t t16 = h.hdr[0];
    if (e == 0 && code16 >= t16) {  // a long code: the second level (t16 >= 65536 - 256)
        e = h.l2[code16 - t16];
        if (is_dc && (e & kK2BadCat) != 0) return kDetailInvalidHuffmanCode;
    }
    uint32_t size, s;
    if (e == 0) {
        // longer than the first level decides and not in the second: the reference's walk (an entry of the first level is empty
        // exactly when the code has more than 11 bits, so the walk may start there)
        const uint16_t *maxcode = reinterpret_cast<const uint16_t *>(h.small);
        size = kK2LutBits + 1;
        while (code16 > maxcode[size]) size++;  // maxcode[17] = 0xFFFF terminates
        if (size > 16) return kDetailInvalidHuffmanCode;
        const uint32_t sym = h.small[56 + ((h.small[36 + size] + (code16 >> (16 - size))) & 0xFF)];
        s = is_dc ? sym : (sym & 15u);
        adv = (sym & 15u) ? (sym >> 4) + 1u : ((sym >> 4) ? 16u : 63u);
        if (s > 16u) return kDetailInvalidHuffmanCode;
    } else {
        s = is_dc ? ((e >> 6) & 31u) : (e >> 12);
        size = (e & 63u) - s;
        adv = (e >> 6) & 63u;
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

// One symbol, fast path for every lane, then ONE branch the wave skips unless some lane needs more: a long code (second
// level), the last bits of the interval, a ring that ran dry (exact path); advances the position.
// adv2 = zig-zag advance in int16 BYTES (2 x coefficients; 126 = EOB: past the end from any AC position).
// On failure the lane gets adv2 = 254 (leaves the AC loop), n = 0, value = 0 and the detail code is returned.
template <bool IS_DC>
__device__ __forceinline__ uint32_t k2_symbol(uint8_t *ring, K2Feed &f, K2Pos &p, int32_t endpos, int32_t &lim, const K2Tab &h, bool closed_by_marker,
                                              int32_t &value, uint32_t &adv2) {
    const uint32_t nxt = *reinterpret_cast<const uint32_t *>(ring + __builtin_amdgcn_ubfe((uint32_t)(p.pm1 + 96), 5, 4) * 4);
    const uint32_t hi = __builtin_amdgcn_alignbit(p.w0, p.w1, ~(uint32_t)p.pm1);
    const uint32_t e = h.lut[hi >> (32 - kK2LutBits)];
    uint32_t n = e & 63u;
    const uint32_t cat = IS_DC ? ((e >> 6) & 31u) : (e >> 12);
    const int32_t raw = (int32_t)__builtin_amdgcn_ubfe(hi, 32u - n, cat);
    value = raw - ((((raw + raw) >> cat) - 1) & ((1 << cat) - 1));  // Extend(v, nbits)
    adv2 = (e >> 5) & 0x7Eu;
    uint32_t err = 0;
    // slow: the entry is empty (e - 1 is negative), a DC category above 16 (bit 15), or the symbol does not fit below the
    // limit -- one signed test
    const bool slow = (int32_t)((e - 1u) | (IS_DC ? e << 16 : 0u) | (uint32_t)(lim - (p.pm1 + 1) - (int32_t)n)) < 0;
    if (slow) {  // exec-masked; the wave skips it when no lane is flagged
        uint32_t adv = 0;
        err = k2_slow_symbol(ring, f, p.pm1, endpos, h, IS_DC, closed_by_marker, n, value, adv);
        adv2 = adv * 2u;
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
// (the same with the table kind known only per lane: the K2S final pass on its way to its first MCU)
__device__ __forceinline__ uint32_t k2_symbol_any(uint8_t *ring, K2Feed &f, K2Pos &p, int32_t endpos, int32_t &lim, const K2Tab &h, bool is_dc,
                                                  bool closed_by_marker, int32_t &value, uint32_t &adv2) {
    const uint32_t nxt = *reinterpret_cast<const uint32_t *>(ring + __builtin_amdgcn_ubfe((uint32_t)(p.pm1 + 96), 5, 4) * 4);
    const uint32_t hi = __builtin_amdgcn_alignbit(p.w0, p.w1, ~(uint32_t)p.pm1);
    const uint32_t e = h.lut[hi >> (32 - kK2LutBits)];
    uint32_t n = e & 63u;
    const uint32_t cat = is_dc ? ((e >> 6) & 31u) : (e >> 12);
    const int32_t raw = (int32_t)__builtin_amdgcn_ubfe(hi, 32u - n, cat);
    value = raw - ((((raw + raw) >> cat) - 1) & ((1 << cat) - 1));
    adv2 = is_dc ? 0u : ((e >> 5) & 0x7Eu);
    uint32_t err = 0;
    const bool slow = (int32_t)((e - 1u) | (is_dc ? e << 16 : 0u) | (uint32_t)(lim - (p.pm1 + 1) - (int32_t)n)) < 0;
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
#ifndef JPGPU_SR_LB
#define JPGPU_SR_LB 10
#endif
constexpr int kSrLutBits = JPGPU_SR_LB;  // lookup width of the round kernel (a 10-bit prefix decides codes of up to 10 bits)
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
__global__ __launch_bounds__(256) void lut_pool_kernel(const DevHuffTable *__restrict__ pool, uint8_t *__restrict__ lut_pool) {
    const DevHuffTable &h = pool[blockIdx.x >> 1];
    const bool is_dc = (blockIdx.x & 1) != 0;
    uint8_t *img = lut_pool + (size_t)blockIdx.x * kK2TabBytes;
    uint16_t *l1 = reinterpret_cast<uint16_t *>(img);
    uint16_t *l2 = reinterpret_cast<uint16_t *>(img + kK2L1Bytes);
    __shared__ uint32_t first_miss;
    if (threadIdx.x == 0) first_miss = 1u << kK2LutBits;
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < (1u << kK2LutBits); i += 256) {
        // a code of at most 11 bits is decided by the prefix alone (maxcode[l] has its low 16 - l bits set): evaluate with ones behind it
        const uint32_t e = k2_entry_of(h, (i << (16 - kK2LutBits)) | ((1u << (16 - kK2LutBits)) - 1u), is_dc, kK2LutBits);
        l1[i] = (uint16_t)e;
        if (e == 0) atomicMin(&first_miss, i);  // (a bad category is an answer, not a miss)
    }
    __syncthreads();
    const uint32_t lo = first_miss << (16 - kK2LutBits);
    const uint32_t t16 = lo > 65536u - kK2L2Entries ? lo : 65536u - kK2L2Entries;
    for (uint32_t j = threadIdx.x; j < kK2L2Entries; j += 256) l2[j] = t16 + j < 65536u ? (uint16_t)k2_entry_of(h, t16 + j, is_dc, 16) : (uint16_t)0;
    if (threadIdx.x < 4) reinterpret_cast<uint32_t *>(img + kK2L1Bytes + 2u * kK2L2Entries)[threadIdx.x] = threadIdx.x == 0 ? t16 : 0u;
    if (threadIdx.x < kK2SmallBytes / 16)
        reinterpret_cast<uint4 *>(img + kK2L1Bytes + 2u * kK2L2Entries + 16u)[threadIdx.x] =
            reinterpret_cast<const uint4 *>(reinterpret_cast<const uint8_t *>(&h) + offsetof(DevHuffTable, maxcode))[threadIdx.x];
}

#ifdef JPGPU_K2_PROFILE
__device__ unsigned long long k2_prof[8];
#define K2_TICK() __builtin_readcyclecounter()
#define K2_PROF_ADD(i, v) do { if (lane == 0) atomicAdd(&k2_prof[i], (unsigned long long)(v)); } while (0)
extern "C" int jpgpu_debug_k2_profile(unsigned long long *out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(k2_prof), sizeof(k2_prof)) != hipSuccess) return 1;
    if (reset) { unsigned long long z[16] = {}; if (hipMemcpyToSymbol(HIP_SYMBOL(k2_prof), z, sizeof z) != hipSuccess) return 1; }
    return 0;
}
#else
#define K2_TICK() 0ull
#define K2_PROF_ADD(i, v) do { (void)(v); } while (0)
#endif

// the scan's tables as the K2 family keeps them in LDS: the pooled images (lut_pool_kernel) copied as they are
__device__ __forceinline__ void k2_stage_scan_tables(const DevScan &s, const uint8_t *lut_pool, uint8_t *tabs, uint32_t *blk_info, int n_slots,
                                                     uint32_t nthreads) {
    const uint32_t tid = threadIdx.x;
    for (int sl = 0; sl < kMaxHuffSlots && sl < n_slots; sl++) {
        const uint32_t pi = s.huff_pool[sl];
        if (pi == 0xFFFF) continue;
        bool is_dc = false;
        for (int c = 0; c < s.scan_components; c++) is_dc |= s.comp[c].dc_slot == sl;
        const uint4 *src = reinterpret_cast<const uint4 *>(lut_pool + ((size_t)pi * 2 + (is_dc ? 1 : 0)) * kK2TabBytes);
        uint4 *dst = reinterpret_cast<uint4 *>(tabs + sl * kK2TabBytes);
        for (uint32_t i = tid; i < kK2TabBytes / 16; i += nthreads) dst[i] = src[i];
    }
    // per block-in-MCU: scan component | DC slot << 8 | AC slot << 16 (kept in LDS: the block loop must not touch global
    // memory for it, a vector load there would wait for the coefficient stores of the previous block)
    if (tid < kMaxBlocksPerMcu) {
        const uint32_t ci = s.blk_comp[tid];
        blk_info[tid] = ci | ((uint32_t)s.comp[ci].dc_slot << 8) | ((uint32_t)s.comp[ci].ac_slot << 16);
    }
}

template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void huffman_decode_kernel(const uint8_t *__restrict__ udata,
                                                                    const DevScan *__restrict__ scans,
                                                                    const HuffWork *__restrict__ work,
                                                                    const uint32_t *__restrict__ ends_u,
                                                                    DevScanStatus *__restrict__ status,
                                                                    const DevHuffTable *__restrict__ huff_pool,
                                                                    int16_t *__restrict__ coefs, int n_slots,
                                                                    const uint8_t *__restrict__ lut_pool) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t *tabs = smem;                                      // n_slots * kK2TabBytes
    uint8_t *wave_all = smem + (size_t)n_slots * kK2TabBytes;  // WAVES * kK2WaveBytes
    uint32_t *blk_info = reinterpret_cast<uint32_t *>(wave_all + WAVES * kK2WaveBytes);  // [kMaxBlocksPerMcu]

    const HuffWork wk = work[blockIdx.x];
    const DevScan &s = scans[wk.scan];
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63;
    const uint32_t wave = tid >> 6;
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

    const unsigned long long k2_t1 = K2_TICK();
    const DevScanStatus st = status[wk.scan];
    const uint32_t n_ends = st.n_ends;
    const uint32_t n_intervals = s.n_intervals;
    const uint32_t total_mcus = s.total_mcus;
    const uint32_t dri_eff = s.dri ? s.dri : total_mcus;
    const uint32_t bpm = s.blocks_per_mcu;
    const uint32_t wave_first = wk.first_interval + wave * 64;
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
            const K2Tab hdc = k2_tab(tabs, (bi >> 8) & 0xFFu);
            const K2Tab hac = k2_tab(tabs, bi >> 16);
            const unsigned long long k2_a = K2_TICK();
            int32_t lim = k2_limit(endpos, feed.wr);
            if (active && err == 0 && mcu < my_mcus) {
                // ReadBlockBaseline (ref: ScanDecoder/JpegHuffmanBaselineScanDecoder.cs:179-222)
                int32_t v;
                uint32_t adv = 0;
                err = k2_symbol<true>(ring, feed, pos, endpos, lim, hdc, closed_by_marker, v, adv);
                const int32_t pred = ci == 0 ? pred0 : (ci == 1 ? pred1 : (ci == 2 ? pred2 : pred3));
                v += pred;
                if (ci == 0) pred0 = v;
                else if (ci == 1) pred1 = v;
                else if (ci == 2) pred2 = v;
                else pred3 = v;
                *reinterpret_cast<int16_t *>(my_stage + swz16) = (int16_t)v;  // zig-zag index 0
                uint32_t i2 = err == 0 ? 2u : 128u;  // 2 x zig-zag index of the next coefficient
                while (i2 < 128u) {
                    const uint32_t e2 = k2_symbol<false>(ring, feed, pos, endpos, lim, hac, closed_by_marker, v, adv);
                    err |= e2;
                    i2 += adv;
                    // Math.Min(i++, 63) for a coefficient; EOB / ZRL store a zero at a position nothing was written to yet
                    const uint32_t at = i2 - 2u < 126u ? i2 - 2u : 126u;
                    *reinterpret_cast<int16_t *>(my_stage + (at ^ swz16)) = (int16_t)v;
                }
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

__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
// the wave's LDS accesses so far are complete before those behind this line start (one wave per workgroup: no s_barrier)
#define PS_WAVE_SYNC()                                        \
    do {                                                      \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    \
        __builtin_amdgcn_wave_barrier();                      \
    } while (0)
__device__ __forceinline__ uint32_t lane_get(uint32_t v, uint32_t l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)l); }
__device__ __forceinline__ uint32_t mbcnt64(uint64_t m) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}
__device__ __forceinline__ uint32_t lane_put(uint32_t old, uint32_t v, uint32_t l) {
    // (no clang builtin for v_writelane in ROCm 7.2, and two scalar operands need M0 on gfx9; this runs once per long code:
    // a compare and a select instead of hand-written M0 traffic)
    return __lane_id() == l ? v : old;
}

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
    const uint32_t e = sr_ld32(tab + (hi >> (32 - LB)) * 4u);
    const int32_t rem = endpos - pos;
    if ((e & 63u) != 0 && L.pm1 < L.slim) return;  // it was the ring: the next fast step takes it
    uint32_t n, adv;
    int32_t v;
    if ((e & 63u) != 0) {  // decided by the lookup, but inside the last 32 bits of the data
        n = e & 63u;
        adv = __builtin_amdgcn_ubfe(e, 6, 7);
        v = (int32_t)e >> 16;
    } else {
        const uint32_t sl = (tab - lut0) >> (LB + 2);
        const uint32_t code16 = hi >> 16;
        uint32_t size, cat;
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
template <int LB>
__device__ __forceinline__ uint32_t sr_entry(const uint16_t *l1, uint32_t i, bool is_dc) {
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
        return n | (1u << 6) | ((uint32_t)v << 16);
    }
    const uint32_t adv = (e >> 6) & 63u;
    return n | ((adv == 63u ? 64u : adv) << 6);
}

__global__ __launch_bounds__(256) void subseq_round_kernel(const uint8_t *__restrict__ udata, const DevScan *__restrict__ scans,
                                                            const HuffWork *__restrict__ work, const uint32_t *__restrict__ ends_u,
                                                            const DevScanStatus *__restrict__ status, const DevHuffTable *__restrict__ huff_pool,
                                                            const uint8_t *__restrict__ lut_pool,
                                                            const uint32_t *__restrict__ exit_in, uint32_t *__restrict__ exit_out,
                                                            uint32_t *__restrict__ nblk_out, uint32_t *__restrict__ entry_used,
                                                            int4 *__restrict__ dcsum_out, uint32_t *__restrict__ changed, int round,
                                                            int n_slots, uint32_t warm_bits) {
    constexpr int LB = kSrLutBits;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
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
    const uint32_t sub = wk.first_interval + tid;
    const bool in_range = sub < s.n_subs;
    const uint32_t slot = s.sub_off + (in_range ? sub : 0);
    uint32_t entry = 0;  // start of a block of the first component, no overshoot
    if (in_range && sub > 0 && round > 0) {
        const uint32_t prev = exit_in[slot - 1];
        if (!(prev & kSubBad)) entry = prev;
    }
    // a lane whose entry state did not change since it last decoded keeps its exit state (and block count);
    // a workgroup with no lane left to decode leaves before staging anything (most workgroups after round 1)
    const bool need = in_range && !(round > 0 && (sub == 0 || entry_used[slot] == entry));
    if (in_range && !need) exit_out[slot] = exit_in[slot];
    if (!__syncthreads_or(need ? 1 : 0)) return;

    // ---- stage: lookups, the reference's small arrays (long codes), block info
    for (int sl = 0; sl < kMaxHuffSlots && sl < n_slots; sl++) {
        const uint32_t pi = s.huff_pool[sl];
        if (pi == 0xFFFF) continue;
        bool is_dc = false;
        for (int c = 0; c < s.scan_components; c++) is_dc |= s.comp[c].dc_slot == sl;
        const uint16_t *src = reinterpret_cast<const uint16_t *>(lut_pool + ((size_t)pi * 2 + (is_dc ? 1 : 0)) * kK2TabBytes);
        if (tid == 0) pool_off[sl] = (uint32_t)(((size_t)pi * 2 + (is_dc ? 1 : 0)) * kK2TabBytes);
        uint32_t *dst = reinterpret_cast<uint32_t *>(smem) + ((size_t)sl << LB);
        for (uint32_t i = tid; i < (1u << LB); i += 256) dst[i] = sr_entry<LB>(src, i, is_dc);
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
    if (!need) return;
    // ROUND 0 IS A WARM-UP.  Its entry state is a guess for every lane but the first, so all it can deliver is a plausible
    // exit state (right when the decode re-synchronises before the subsequence ends) -- and every such lane is decoded again
    // in round 1 anyway, from its predecessor's exit.  It therefore only decodes the LAST warm_bits bits of the subsequence:
    // less work in round 0, more lanes to redo in rounds 2-3 (the block phase is what converges slowly: the total number of
    // re-decodes is set by how far the nearest upstream synchronisation point is, not by round 0) -- a small net gain,
    // 19.6 -> 19.2-19.4 ms per 1024 x 4K.  entry_used is poisoned so that round 1 decodes the lane whatever its entry turns out to be.
    const bool warm = round == 0 && sub > 0 && warm_bits != 0 && warm_bits < (1u << s.sub_shift);
    entry_used[slot] = warm ? 0xFFFFFFFFu : entry;

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
        // how many exits this round changed (the host's convergence test; one atomic per wave that changed anything)
        const bool ch = round == 0 || ex != exit_in[slot];
        const uint64_t m = __ballot(ch);
        if (ch && (uint32_t)__builtin_ctzll(m) == (threadIdx.x & 63u)) atomicAdd(changed, (uint32_t)__builtin_popcountll(m));
    }
    exit_out[slot] = ex;
    nblk_out[slot] = nblk;
    dcsum_out[slot] = dcs;
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
__global__ __launch_bounds__(256) void subseq_same_kernel(const uint8_t *__restrict__ udata, const DevScan *__restrict__ scans,
                                                          const HuffWork *__restrict__ work, const uint32_t *__restrict__ ends_u,
                                                          const DevScanStatus *__restrict__ status, uint32_t *__restrict__ same_dist) {
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
                                                              int4 *__restrict__ dcsum, uint32_t *__restrict__ n_copied) {
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
            for (uint32_t l = (base == 0 ? 1u : 0u); l < n_here; l++) {
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
constexpr int kSubFinalMaxWaves = JPGPU_SF_WAVES > 4 ? JPGPU_SF_WAVES : 4;  // the launch picks subseq_final_waves(n_slots)
constexpr int kSfWaveBytes = kK2WaveBytes + 64 * 2 * (int)sizeof(uint32_t);  // K2's staging + rings, then (first MCU, count) per lane
__global__ __launch_bounds__(64 * kSubFinalMaxWaves) void subseq_final_kernel(const uint8_t *__restrict__ udata, const DevScan *__restrict__ scans,
                                                                           const HuffWork *__restrict__ work, const uint32_t *__restrict__ ends_u,
                                                                           DevScanStatus *__restrict__ status,
                                                                           const DevHuffTable *__restrict__ huff_pool,
                                                                           const uint8_t *__restrict__ lut_pool,
                                                                           const uint32_t *__restrict__ exit_state,
                                                                           const uint32_t *__restrict__ first_block,
                                                                           const int4 *__restrict__ dc_entry, int16_t *__restrict__ coefs,
                                                                           int n_slots) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t n_waves = blockDim.x >> 6;
    uint8_t *tabs = smem;                                                    // n_slots * kK2TabBytes
    uint8_t *wave_all = smem + (size_t)n_slots * kK2TabBytes;                // n_waves * kSfWaveBytes
    uint32_t *blk_info = reinterpret_cast<uint32_t *>(wave_all + n_waves * kSfWaveBytes);  // [kMaxBlocksPerMcu]
    const HuffWork wk = work[blockIdx.x];
    const DevScan &s = scans[wk.scan];
    const DevScanStatus st = status[wk.scan];
    if (st.n_ends == 0) return;
    k2_stage_scan_tables(s, lut_pool, tabs, blk_info, n_slots, blockDim.x);
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint8_t *stage = wave_all + wave * kSfWaveBytes;
    uint8_t *ring = stage + 8192 + lane * kK2RingStride;
    uint32_t *meta = reinterpret_cast<uint32_t *>(stage + kK2WaveBytes);
    {
        const uint4 z = {0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < 8; i++) reinterpret_cast<uint4 *>(stage)[i * 64 + lane] = z;
    }
    __syncthreads();
    const uint32_t ulen = ends_u[s.ends_off];
    const uint32_t total_bits = ulen * 8;
    // a lane takes kSubFinalSubsPerLane consecutive subsequences: half the lanes, but half the parsed-not-stored blocks and a
    // narrower spread of MCU counts inside a wave (the wave iterates to its largest)
    const uint32_t sub = wk.first_interval + tid * (uint32_t)kSubFinalSubsPerLane;
    const uint32_t slot = s.sub_off + (sub < s.n_subs ? sub : 0);
    const uint32_t total_mcus = s.total_mcus;
    const uint32_t bpm = s.blocks_per_mcu;
    const uint64_t coef_off = s.coef_off;
    const bool closed_by_marker = st.terminator != 0;

    bool live = sub < s.n_subs;
    uint32_t entry = 0;
    if (live && sub > 0) {
        const uint32_t prev = exit_state[slot - 1];
        if (prev & kSubBad) live = false;  // the stream ended or failed in an earlier subsequence: reported by that lane
        else entry = prev;
    }
    uint32_t b_in_mcu = (entry >> 6) & 31u;
    uint32_t i2 = ((entry >> 11) & 127u) * 2u;  // 2 x zig-zag position inside the block in progress at the entry
    // first_block = blocks completed before the entry = index of the block in progress (i2 != 0: the previous lane's) or
    // about to start there.  This lane owns the MCUs from the first one that starts at or behind its entry ...
    uint32_t my_first = total_mcus, my_end = total_mcus, skip = 0;
    if (live) {
        const uint32_t at = first_block[slot];
        my_first = (at + (i2 != 0 ? 1u : 0u) + bpm - 1) / bpm;
        skip = my_first * bpm - at;  // block ends between the entry and that MCU (the first of them may be half a block away)
        // ... up to the first one that starts at or behind the next lane's entry (a stream that failed or ran out inside this
        // lane's subsequences leaves it everything that remains)
        const uint32_t n_mine = s.n_subs - sub < (uint32_t)kSubFinalSubsPerLane ? s.n_subs - sub : (uint32_t)kSubFinalSubsPerLane;
        bool open_end = sub + n_mine >= s.n_subs;
        uint32_t ex = 0;
        for (uint32_t q = 0; q < n_mine; q++) {
            ex = exit_state[slot + q];
            open_end |= (ex & kSubBad) != 0;
        }
        if (!open_end) my_end = (first_block[slot + n_mine] + ((((ex >> 11) & 127u) != 0) ? 1u : 0u) + bpm - 1) / bpm;
        if (my_end > total_mcus) my_end = total_mcus;  // the reference stops after the last MCU
        if (my_first > my_end) my_first = my_end;
    }
    const uint32_t count = my_end - my_first;
    meta[lane * 2] = my_first;
    meta[lane * 2 + 1] = count;
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
                err = k2_symbol_any(ring, feed, pos, endpos, lim, k2_tab(tabs, is_dc ? ((info >> 8) & 0xFFu) : (info >> 16)), is_dc, closed_by_marker,
                                    v, adv);
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

    for (uint32_t j = 0; j < wave_count; j++) {
        for (uint32_t b = 0; b < bpm; b++) {
            const uint32_t bi = __builtin_amdgcn_readfirstlane(blk_info[b]);  // wave-uniform
            const uint32_t ci = bi & 0xFFu;
            const K2Tab hdc = k2_tab(tabs, (bi >> 8) & 0xFFu);
            const K2Tab hac = k2_tab(tabs, bi >> 16);
            lim = k2_limit(endpos, feed.wr);
            if (j < count && err == 0) {
                // ReadBlockBaseline (ref: ScanDecoder/JpegHuffmanBaselineScanDecoder.cs:179-222)
                int32_t v;
                uint32_t adv = 0;
                err = k2_symbol<true>(ring, feed, pos, endpos, lim, hdc, closed_by_marker, v, adv);
                const int32_t pred = ci == 0 ? pred0 : (ci == 1 ? pred1 : (ci == 2 ? pred2 : pred3));
                v += pred;
                if (ci == 0) pred0 = v;
                else if (ci == 1) pred1 = v;
                else if (ci == 2) pred2 = v;
                else pred3 = v;
                *reinterpret_cast<int16_t *>(my_stage + swz16) = (int16_t)v;  // zig-zag index 0
                uint32_t k2i = err == 0 ? 2u : 128u;
                while (k2i < 128u) {
                    const uint32_t e2 = k2_symbol<false>(ring, feed, pos, endpos, lim, hac, closed_by_marker, v, adv);
                    err |= e2;
                    k2i += adv;
                    // Math.Min(i++, 63) for a coefficient; EOB / ZRL store a zero at a position nothing was written to yet
                    const uint32_t at = k2i - 2u < 126u ? k2i - 2u : 126u;
                    *reinterpret_cast<int16_t *>(my_stage + (at ^ swz16)) = (int16_t)v;
                }
            }
            k2_topup(ring, feed, pos.pm1);
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
                const uint32_t owner_first = meta[blk * 2], owner_count = meta[blk * 2 + 1];
                if (j < owner_count) *reinterpret_cast<uint4 *>(coefs + (coef_off + ((uint64_t)owner_first + j) * bpm + b) * 64 + chunk * 8) = v;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    }
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

// ------------------------------------------------------------------------------------------------
// K3: dequantise + IDCT + level shift + block output.  One lane per block.
// ------------------------------------------------------------------------------------------------

// ref: JpegZigZag.cs:27-38
__device__ constexpr uint8_t kNat[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                         41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                         30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

// The 8-point butterfly of IDCT8x4_LeftPart/RightPart (ref: FastFloatingPointDCT.cs:79-127), one column.
// Operation order and parenthesisation are normative.
#define JPGPU_IDCT8(y0, y1, y2, y3, y4, y5, y6, y7)                     \
    {                                                                   \
        float mz0 = y1 + y7;                                            \
        float mz2 = y3 + y7;                                            \
        float mz1 = y3 + y5;                                            \
        float mz3 = y1 + y5;                                            \
        float mz4 = (mz0 + mz1) * 1.175875602f;                         \
        mz2 = (mz2 * -1.961570560f) + mz4;                              \
        mz3 = (mz3 * -0.390180644f) + mz4;                              \
        mz0 = mz0 * -0.899976223f;                                      \
        mz1 = mz1 * -2.562915447f;                                      \
        const float mb3 = ((y7 * 0.298631336f) + mz0) + mz2;            \
        const float mb2 = ((y5 * 2.053119869f) + mz1) + mz3;            \
        const float mb1 = ((y3 * 3.072711026f) + mz1) + mz2;            \
        const float mb0 = ((y1 * 1.501321110f) + mz0) + mz3;            \
        mz4 = (y2 + y6) * 0.541196100f;                                 \
        mz0 = y0 + y4;                                                  \
        mz1 = y0 - y4;                                                  \
        mz2 = mz4 + (y6 * -1.847759065f);                               \
        mz3 = mz4 + (y2 * 0.765366865f);                                \
        const float a0 = mz0 + mz3;                                     \
        const float a3 = mz0 - mz3;                                     \
        const float a1 = mz1 + mz2;                                     \
        const float a2 = mz1 - mz2;                                     \
        y0 = a0 + mb0;                                                  \
        y7 = a0 - mb0;                                                  \
        y1 = a1 + mb1;                                                  \
        y6 = a1 - mb1;                                                  \
        y2 = a2 + mb2;                                                  \
        y5 = a2 - mb2;                                                  \
        y3 = a3 + mb3;                                                  \
        y4 = a3 - mb3;                                                  \
    }

// DequantizeBlockAndUnZigZag for one block (ref: ScanDecoder/JpegScanDecoder.cs:50-62).
// c_lds: this lane's 64 int16 coefficients (zig-zag) in the swizzled LDS staging (8 chunks of 16 B, chunk p at
// c_lds + ((p ^ swz) * 16)); q_lds: 64 uint16 quantisers (zig-zag) of the block's component.
__device__ __forceinline__ void block_dequant(const uint8_t *c_lds, uint32_t swz, const uint16_t *q_lds, float (&f)[64]) {
#pragma unroll
    for (int piece = 0; piece < 8; piece++) {
        const uint4 cv = *reinterpret_cast<const uint4 *>(c_lds + ((piece ^ swz) * 16));
        const uint4 qv = reinterpret_cast<const uint4 *>(q_lds)[piece];
        const uint32_t cw[4] = {cv.x, cv.y, cv.z, cv.w};
        const uint32_t qw[4] = {qv.x, qv.y, qv.z, qv.w};
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int k = piece * 8 + j;
            const uint32_t w = cw[j >> 1], q = qw[j >> 1];
            const int32_t c = (j & 1) ? ((int32_t)w >> 16) : (int32_t)(int16_t)(w & 0xFFFF);
            const int32_t qq = (j & 1) ? (int32_t)(q >> 16) : (int32_t)(q & 0xFFFF);
            f[kNat[k]] = (float)(qq * c);  // ushort * short -> int -> float
        }
    }
}

typedef float float2v __attribute__((ext_vector_type(2)));
typedef short short2v __attribute__((ext_vector_type(2)));

// The 8-point butterfly of IDCT8x4_LeftPart/RightPart (ref: FastFloatingPointDCT.cs:79-127) on T = float or a pair of
// floats (two independent columns at once: v_pk_add_f32 / v_pk_mul_f32, each component one IEEE operation, never fused).
// Operation order and parenthesisation are normative.
template <typename T>
__device__ __forceinline__ void idct8(T &y0, T &y1, T &y2, T &y3, T &y4, T &y5, T &y6, T &y7) {
    T mz0 = y1 + y7;
    T mz2 = y3 + y7;
    T mz1 = y3 + y5;
    T mz3 = y1 + y5;
    T mz4 = (mz0 + mz1) * 1.175875602f;
    mz2 = (mz2 * -1.961570560f) + mz4;
    mz3 = (mz3 * -0.390180644f) + mz4;
    mz0 = mz0 * -0.899976223f;
    mz1 = mz1 * -2.562915447f;
    const T mb3 = ((y7 * 0.298631336f) + mz0) + mz2;
    const T mb2 = ((y5 * 2.053119869f) + mz1) + mz3;
    const T mb1 = ((y3 * 3.072711026f) + mz1) + mz2;
    const T mb0 = ((y1 * 1.501321110f) + mz0) + mz3;
    mz4 = (y2 + y6) * 0.541196100f;
    mz0 = y0 + y4;
    mz1 = y0 - y4;
    mz2 = mz4 + (y6 * -1.847759065f);
    mz3 = mz4 + (y2 * 0.765366865f);
    const T a0 = mz0 + mz3;
    const T a3 = mz0 - mz3;
    const T a1 = mz1 + mz2;
    const T a2 = mz1 - mz2;
    y0 = a0 + mb0;
    y7 = a0 - mb0;
    y1 = a1 + mb1;
    y6 = a1 - mb1;
    y2 = a2 + mb2;
    y5 = a2 - mb2;
    y3 = a3 + mb3;
    y4 = a3 - mb3;
}

// TransformIDCT + ShiftDataLevel (ref: FastFloatingPointDCT.cs:54-70, ScanDecoder/JpegScanDecoder.cs:64-73) on a
// dequantised block held in registers, two lanes of the butterfly per instruction.
// out[r * 4 + c2] = samples (r, 2*c2) | (r, 2*c2 + 1) << 16 as int16: (short)(Round(v) + levelShift), unclamped.
__device__ __forceinline__ void block_idct(const float (&f)[64], int32_t level_shift, uint32_t (&out)[32]) {
    // pass 1: 1-D IDCT along each ROW (the reference transposes, runs the column butterfly, transposes back);
    // a[r2][c] holds rows 2*r2 and 2*r2+1 of column c
    float2v a[4][8];
#pragma unroll
    for (int r2 = 0; r2 < 4; r2++)
#pragma unroll
        for (int c = 0; c < 8; c++) a[r2][c] = float2v{f[(2 * r2) * 8 + c], f[(2 * r2 + 1) * 8 + c]};
#pragma unroll
    for (int r2 = 0; r2 < 4; r2++) idct8(a[r2][0], a[r2][1], a[r2][2], a[r2][3], a[r2][4], a[r2][5], a[r2][6], a[r2][7]);
    // pass 2: along each COLUMN; b[r][c2] holds columns 2*c2 and 2*c2+1 of row r
    float2v b[8][4];
#pragma unroll
    for (int r2 = 0; r2 < 4; r2++)
#pragma unroll
        for (int c2 = 0; c2 < 4; c2++) {
            b[2 * r2][c2] = float2v{a[r2][2 * c2].x, a[r2][2 * c2 + 1].x};
            b[2 * r2 + 1][c2] = float2v{a[r2][2 * c2].y, a[r2][2 * c2 + 1].y};
        }
#pragma unroll
    for (int c2 = 0; c2 < 4; c2++) idct8(b[0][c2], b[1][c2], b[2][c2], b[3][c2], b[4][c2], b[5][c2], b[6][c2], b[7][c2]);
    const uint32_t shift2 = ((uint32_t)level_shift & 0xFFFFu) * 0x00010001u;
#pragma unroll
    for (int r = 0; r < 8; r++)
#pragma unroll
        for (int c2 = 0; c2 < 4; c2++) {
            const float2v v = b[r][c2] * 0.1250f;                  // MultiplyInplace(C_0_125)
            const int32_t x = (int32_t)__builtin_rintf(v.x);       // MathF.Round: half to even (v_rndne_f32)
            const int32_t y = (int32_t)__builtin_rintf(v.y);
            const uint32_t pk = ((uint32_t)x & 0xFFFFu) | ((uint32_t)y << 16);
            // (short)(Round + levelShift): 16-bit wrap-around add on both halves (v_pk_add_u16)
            const short2v sum = __builtin_bit_cast(short2v, pk) + __builtin_bit_cast(short2v, shift2);
            out[r * 4 + c2] = __builtin_bit_cast(uint32_t, sum);
        }
}

// signed clamp of two int16 samples to [0, 255] (JpegBufferOutputWriter8Bit.ClampTo8Bit): v_pk_max_i16 + v_pk_min_i16
__device__ __forceinline__ uint32_t clamp2_u8(uint32_t pk) {
    short2v v = __builtin_bit_cast(short2v, pk);
    v = __builtin_elementwise_max(v, short2v{0, 0});
    v = __builtin_elementwise_min(v, short2v{255, 255});
    return __builtin_bit_cast(uint32_t, v);
}
// four clamped samples (two packed pairs) -> four bytes
__device__ __forceinline__ uint32_t pack4_u8(uint32_t pk01, uint32_t pk23) {
    return __builtin_amdgcn_perm(clamp2_u8(pk23), clamp2_u8(pk01), 0x06040200u);  // bytes 0,2 of pk01 then 0,2 of pk23
}
// byte gather from the 8 bytes {lo (indices 0-3), hi (indices 4-7)}: one v_perm_b32
__device__ __forceinline__ uint32_t pick4(uint32_t lo, uint32_t hi, uint32_t sel) { return __builtin_amdgcn_perm(hi, lo, sel); }
#define JPGPU_SEL(a, b, c, d) ((uint32_t)(a) | ((uint32_t)(b) << 8) | ((uint32_t)(c) << 16) | ((uint32_t)(d) << 24))

constexpr int kIdctThreads = 256;
constexpr uint32_t kPxRowStride = kIdctThreads * 8;  // bytes between sample rows in the LDS pixel tile

// Output layout classes of the INTERLEAVED_U8 format (chosen per scan on the host, see idct_layout_class()).
enum IdctLayout : int { kLayGeneric = 0, kLayYccH1V1 = 1, kLayYccH2V1 = 2, kLayYccH2V2 = 3, kLayGray = 4, kNumIdctLayouts = 5 };

// ---- YCbCr -> RGB(A) (ref: apps/JpegDecode/JpegYCbCrToRgbConverter.cs:134-206).  The reference looks the terms up in
// tables built by Init (:66-118); with ReferenceBlackWhite = {0,255,128,255,128,255} the tables are exactly
//   yTable[i] = i, crRTable[i] = (cr_r * (i-128) + half) >> 16, cbBTable[i] = (cb_b * (i-128) + half) >> 16,
//   crGTable[i] = cr_g * (i-128), cbGTable[i] = cb_g * (i-128) + half,  and the clamp table is a clamp to [0, 255],
// so the terms are computed instead of fetched (the factors come from the host, derived like Init derives them).
struct ChromaTerms {
    int32_t r, g, b;
};
__device__ __forceinline__ ChromaTerms chroma_terms(uint32_t cb_sample, uint32_t cr_sample, const YccRgbFactors &k) {
    const int32_t cb = (int32_t)cb_sample - 128, cr = (int32_t)cr_sample - 128;
    ChromaTerms t;
    t.r = (k.cr_r * cr + 32768) >> 16;
    t.b = (k.cb_b * cb + 32768) >> 16;
    t.g = (k.cb_g * cb + 32768 + k.cr_g * cr) >> 16;
    return t;
}
__device__ __forceinline__ uint32_t clamp_u8_i32(int32_t v) { return (uint32_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }
// one pixel: R | G << 8 | B << 16
__device__ __forceinline__ uint32_t rgb_pixel(uint32_t y, const ChromaTerms &t) {
    return clamp_u8_i32((int32_t)y + t.r) | (clamp_u8_i32((int32_t)y + t.g) << 8) | (clamp_u8_i32((int32_t)y + t.b) << 16);
}
__device__ __forceinline__ uint32_t byte_of(uint32_t lo, uint32_t hi, int i) { return ((i < 4 ? lo : hi) >> (8 * (i & 3))) & 0xFFu; }
// N pixels (R | G << 8 | B << 16 each) -> interleaved bytes at dst (16-byte aligned for N = 16, 8-byte aligned for N = 8)
template <int N, int BPP>
__device__ __forceinline__ void store_rgb_pixels(uint8_t *dst, const uint32_t (&p)[N]) {
    if (BPP == 4) {
#pragma unroll
        for (int i = 0; i < N; i += 4) {
            const uint4 v = {p[i] | 0xFF000000u, p[i + 1] | 0xFF000000u, p[i + 2] | 0xFF000000u, p[i + 3] | 0xFF000000u};
            *reinterpret_cast<uint4 *>(dst + i * 4) = v;
        }
    } else {
        uint32_t w[N * 3 / 4];
#pragma unroll
        for (int i = 0; i < N; i += 4) {  // four pixels -> three dwords
            w[i * 3 / 4 + 0] = p[i] | (p[i + 1] << 24);
            w[i * 3 / 4 + 1] = (p[i + 1] >> 8) | (p[i + 2] << 16);
            w[i * 3 / 4 + 2] = (p[i + 2] >> 16) | (p[i + 3] << 8);
        }
        if (N == 16) {
#pragma unroll
            for (int i = 0; i < 3; i++) {
                const uint4 v = {w[i * 4], w[i * 4 + 1], w[i * 4 + 2], w[i * 4 + 3]};
                *reinterpret_cast<uint4 *>(dst + i * 16) = v;
            }
        } else {
#pragma unroll
            for (int i = 0; i < N * 3 / 8; i++) {
                const uint2 v = {w[i * 2], w[i * 2 + 1]};
                *reinterpret_cast<uint2 *>(dst + i * 8) = v;
            }
        }
    }
}

// Stand-alone conversion of an interleaved u8 image (C = 3: Y,Cb,Cr; C = 1: Y with Cb = Cr = 128 like
// apps/JpegDecode/DecodeAction.cs:57-65) for the layouts the writer kernel has no fused path for.
__global__ __launch_bounds__(256) void ycc_to_rgb_kernel(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, uint64_t n_pixels, int comps,
                                                         int bpp, YccRgbFactors k) {
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n_pixels; i += (uint64_t)gridDim.x * 256) {
        const uint32_t y = src[i * comps];
        const uint32_t cb = comps == 3 ? src[i * 3 + 1] : 128u, cr = comps == 3 ? src[i * 3 + 2] : 128u;
        const uint32_t px = rgb_pixel(y, chroma_terms(cb, cr, k));
        uint8_t *d = dst + i * bpp;
        d[0] = (uint8_t)px;
        d[1] = (uint8_t)(px >> 8);
        d[2] = (uint8_t)(px >> 16);
        if (bpp == 4) d[3] = 255;
    }
}

// "O3", the xunit tests' sink (ref: tests/JpegLibrary.Tests/Utils/JpegExtendingOutputWriter.cs:30-112) as a device format:
// out[(y * W + x) * 4 + c] (componentCount = 4, the way every test constructs it), uint16.  Input: the PLANAR_I16 planes K3
// wrote = WriteBlock's arguments before chroma expansion.  WriteBlockSlow replicates with shifts (:238-268), so pixel (x, y)
// of component c is plane_c[y >> vshift][x >> hshift]; the writer then takes (ushort)sample -- a negative sample becomes a
// large value -- clamps to 2^P - 1 and spreads the P bits over 16 (FastExpandBits for P >= 8, ExpandBits below).
// One launch for the whole batch: blockIdx.y = image (its descriptor in HBM), blockIdx.x strides over the image's pixels.
__global__ __launch_bounds__(256) void extend_u16_kernel(const uint8_t *__restrict__ planes, uint8_t *__restrict__ out_base,
                                                         const ExtendPlanes *__restrict__ images) {
    const ExtendPlanes g = images[blockIdx.y];
    uint16_t *out = reinterpret_cast<uint16_t *>(out_base + g.out_off);
    for (uint64_t px = (uint64_t)blockIdx.x * 256 + threadIdx.x; px < (uint64_t)g.width * g.height; px += (uint64_t)gridDim.x * 256) {
    const uint32_t y = (uint32_t)(px / g.width), x = (uint32_t)(px - (uint64_t)y * g.width);
    const uint32_t p = g.precision, mx = (1u << p) - 1u;
    uint16_t v4[4] = {0, 0, 0, 0};
#pragma unroll  // (compile-time component index: the descriptor's arrays stay in registers, no scratch)
    for (uint32_t c = 0; c < 4u; c++) {
        if (c >= g.ncomp) continue;
        const int16_t *pl = reinterpret_cast<const int16_t *>(planes + g.plane_off[c]);
        const uint32_t s = (uint32_t)(uint16_t)pl[(uint64_t)(y >> g.vshift[c]) * g.pitch[c] + (x >> g.hshift[c])];
        uint32_t bits = s < mx ? s : mx;  // Clamp((ushort)sample, max)
        if (p >= 8u) {
            const uint32_t rem = 16u - p;
            bits = (bits << rem) | (bits & ((1u << rem) - 1u));  // FastExpandBits, as written
        } else {
            uint32_t cur = p;
            while (cur < 16u) {
                bits = (bits << p) | bits;
                cur += p;
            }
            if (cur > 16u) {
                bits >>= p;
                cur -= p;
                const uint32_t rem = 16u - cur;
                bits = (bits << rem) | (bits & ((1u << rem) - 1u));
            }
        }
        v4[c] = (uint16_t)bits;
    }
    // channels the frame does not have keep what the caller's (fresh, zeroed) buffer held: the batch owns the buffer, so zero
    *reinterpret_cast<uint2 *>(out + px * 4) = uint2{(uint32_t)v4[0] | ((uint32_t)v4[1] << 16), (uint32_t)v4[2] | ((uint32_t)v4[3] << 16)};
    }
}

// Output assembly of the INTERLEAVED_U8 format from the LDS sample tile [8 rows][256 blocks][8 B] (phase C).
// CONV: 0 = the samples as they are (Y,Cb,Cr), 3 / 4 = converted to R,G,B / R,G,B,A bytes (fast layouts only).
template <int LAY, int CONV>
__device__ __forceinline__ void interleaved_output_from_tile(const uint8_t *sh_px, const DevScan &s, uint32_t tile_first, uint32_t n_mcu,
                                                             uint32_t tid, bool have_block, const DevScanComponent &comp, uint32_t mcu_x,
                                                             uint32_t mcu_y, uint32_t b, uint8_t *out, const YccRgbFactors &kf) {
    const uint32_t W = s.width, H = s.height, C = s.frame_components;
    uint8_t *img = out + s.out_off;

    if (LAY == kLayGeneric) {
        // any component count / sampling: bytewise stores with WriteBlockSlow's replication
        // (ref: ScanDecoder/JpegHuffmanBaselineScanDecoder.cs:238-268) and the sink's clipping (x < W, y < H)
        if (have_block) {
            const uint32_t hs = comp.hs, vs = comp.vs;
            const uint32_t x0 = (mcu_x * s.max_h + s.blk_x[b]) * 8, y0 = (mcu_y * s.max_v + s.blk_y[b]) * 8;
            const uint32_t hshift = 31 - __builtin_clz(hs | 1), vshift = 31 - __builtin_clz(vs | 1);
            for (uint32_t v = 0; v < vs; v++)
                for (uint32_t i = 0; i < 8; i++) {
                    const uint32_t y = y0 + 8 * v + i;
                    if (y >= H) continue;
                    const uint8_t *srow = sh_px + ((8 * v + i) >> vshift) * kPxRowStride + tid * 8;
                    for (uint32_t h = 0; h < hs; h++)
                        for (uint32_t j = 0; j < 8; j++) {
                            const uint32_t x = x0 + 8 * h + j;
                            if (x < W) img[((size_t)y * W + x) * C + comp.component_index] = srow[(8 * h + j) >> hshift];
                        }
                }
        }
    } else {
    // YCbCr fast paths: one task = one pixel row of one MCU (8*max_h pixels); consecutive lanes take consecutive MCUs of
    // the same row, so a wave writes one contiguous run of the output row per store instruction group.
    constexpr uint32_t max_h = (LAY == kLayYccH1V1) ? 1 : 2;
    constexpr uint32_t max_v = (LAY == kLayYccH2V2) ? 2 : 1;
    constexpr uint32_t rows_per_mcu = 8 * max_v;
    constexpr uint32_t vshift = max_v >> 1;
    constexpr uint32_t kbpm = max_h * max_v + 2;
    const uint32_t n_tasks = rows_per_mcu * n_mcu;
    for (uint32_t t = tid; t < n_tasks; t += kIdctThreads) {
        const uint32_t row = t / n_mcu, m = t - row * n_mcu;
        const uint32_t gm = tile_first + m;
        const uint32_t gx = gm % s.mcus_per_line, gy = gm / s.mcus_per_line;
        const uint32_t y = gy * rows_per_mcu + row;
        if (y >= H) continue;
        const uint8_t *yrow = sh_px + (row & 7) * kPxRowStride + (m * kbpm + (row >> 3) * max_h) * 8;
        const uint8_t *crow = sh_px + (row >> vshift) * kPxRowStride + (m * kbpm + max_h * max_v) * 8;
        if (max_h == 2 && CONV != 0) {
            const uint4 yv = *reinterpret_cast<const uint4 *>(yrow);  // 16 luma samples (two adjacent blocks)
            const uint4 cv = *reinterpret_cast<const uint4 *>(crow);  // 8 Cb (x,y) + 8 Cr (z,w)
            uint32_t px[16];
#pragma unroll
            for (int j = 0; j < 8; j++) {  // one chroma sample pair covers two pixels
                const ChromaTerms t = chroma_terms(byte_of(cv.x, cv.y, j), byte_of(cv.z, cv.w, j), kf);
                px[2 * j] = rgb_pixel(byte_of(j < 4 ? yv.x : yv.z, j < 4 ? yv.y : yv.w, (2 * j) & 7), t);
                px[2 * j + 1] = rgb_pixel(byte_of(j < 4 ? yv.x : yv.z, j < 4 ? yv.y : yv.w, (2 * j + 1) & 7), t);
            }
            store_rgb_pixels<16, (CONV == 4 ? 4 : 3)>(img + ((size_t)y * W + gx * 16) * (CONV == 4 ? 4 : 3), px);
        } else if (CONV != 0) {
            const uint2 yv = *reinterpret_cast<const uint2 *>(yrow);
            const uint2 bv = *reinterpret_cast<const uint2 *>(crow);
            const uint2 rv = *reinterpret_cast<const uint2 *>(crow + 8);
            uint32_t px[8];
#pragma unroll
            for (int j = 0; j < 8; j++) px[j] = rgb_pixel(byte_of(yv.x, yv.y, j), chroma_terms(byte_of(bv.x, bv.y, j), byte_of(rv.x, rv.y, j), kf));
            store_rgb_pixels<8, (CONV == 4 ? 4 : 3)>(img + ((size_t)y * W + gx * 8) * (CONV == 4 ? 4 : 3), px);
        } else if (max_h == 2) {
            const uint4 yv = *reinterpret_cast<const uint4 *>(yrow);  // 16 luma samples (two adjacent blocks)
            const uint4 cv = *reinterpret_cast<const uint4 *>(crow);  // 8 Cb (x,y) + 8 Cr (z,w)
            const uint32_t cc0 = pick4(cv.x, cv.z, JPGPU_SEL(0, 4, 1, 5)), cc1 = pick4(cv.x, cv.z, JPGPU_SEL(2, 6, 3, 7));
            const uint32_t cc2 = pick4(cv.y, cv.w, JPGPU_SEL(0, 4, 1, 5)), cc3 = pick4(cv.y, cv.w, JPGPU_SEL(2, 6, 3, 7));
            uint4 o0, o1, o2;
            o0.x = pick4(yv.x, cc0, JPGPU_SEL(0, 4, 5, 1));
            o0.y = pick4(yv.x, cc0, JPGPU_SEL(4, 5, 2, 6));
            o0.z = pick4(yv.x, cc0, JPGPU_SEL(7, 3, 6, 7));
            o0.w = pick4(yv.y, cc1, JPGPU_SEL(0, 4, 5, 1));
            o1.x = pick4(yv.y, cc1, JPGPU_SEL(4, 5, 2, 6));
            o1.y = pick4(yv.y, cc1, JPGPU_SEL(7, 3, 6, 7));
            o1.z = pick4(yv.z, cc2, JPGPU_SEL(0, 4, 5, 1));
            o1.w = pick4(yv.z, cc2, JPGPU_SEL(4, 5, 2, 6));
            o2.x = pick4(yv.z, cc2, JPGPU_SEL(7, 3, 6, 7));
            o2.y = pick4(yv.w, cc3, JPGPU_SEL(0, 4, 5, 1));
            o2.z = pick4(yv.w, cc3, JPGPU_SEL(4, 5, 2, 6));
            o2.w = pick4(yv.w, cc3, JPGPU_SEL(7, 3, 6, 7));
            uint4 *dst = reinterpret_cast<uint4 *>(img + ((size_t)y * W + gx * 16) * 3);
            dst[0] = o0;
            dst[1] = o1;
            dst[2] = o2;
        } else {
            const uint2 yv = *reinterpret_cast<const uint2 *>(yrow);
            const uint2 bv = *reinterpret_cast<const uint2 *>(crow);
            const uint2 rv = *reinterpret_cast<const uint2 *>(crow + 8);
            uint2 o0, o1, o2;
            {
                const uint32_t lo = pick4(bv.x, rv.x, JPGPU_SEL(0, 4, 1, 5)), hi = pick4(bv.x, rv.x, JPGPU_SEL(2, 6, 3, 7));
                const uint32_t mid = pick4(lo, hi, JPGPU_SEL(2, 3, 4, 5));
                o0.x = pick4(yv.x, lo, JPGPU_SEL(0, 4, 5, 1));
                o0.y = pick4(yv.x, mid, JPGPU_SEL(4, 5, 2, 6));
                o1.x = pick4(yv.x, hi, JPGPU_SEL(5, 3, 6, 7));
            }
            {
                const uint32_t lo = pick4(bv.y, rv.y, JPGPU_SEL(0, 4, 1, 5)), hi = pick4(bv.y, rv.y, JPGPU_SEL(2, 6, 3, 7));
                const uint32_t mid = pick4(lo, hi, JPGPU_SEL(2, 3, 4, 5));
                o1.y = pick4(yv.y, lo, JPGPU_SEL(0, 4, 5, 1));
                o2.x = pick4(yv.y, mid, JPGPU_SEL(4, 5, 2, 6));
                o2.y = pick4(yv.y, hi, JPGPU_SEL(5, 3, 6, 7));
            }
            uint2 *dst = reinterpret_cast<uint2 *>(img + ((size_t)y * W + gx * 8) * 3);
            dst[0] = o0;
            dst[1] = o1;
            dst[2] = o2;
        }
    }
    }  // YCbCr fast paths
}

// Each workgroup walks a run of consecutive tiles (kIdctThreads / blocks_per_mcu MCUs each) of one scan.
// Pipeline per tile:  lanes dequantise their block out of the LDS staging into registers -> barrier -> the staging is
// refilled for tile i+1 by LDS-DMA (global_load_lds_dwordx4: no VGPRs, asynchronous; the XOR swizzle is applied to the
// per-lane SOURCE address because the LDS side of the DMA is lane-linear) -> IDCT in registers while the DMA is in
// flight -> clamped samples to the LDS tile -> wait for the DMA -> barrier -> output assembly + global stores.
// All LDS lives in one array: staging 32 KiB | u8 sample tile [8 rows][256 blocks][8 B] 16 KiB | quant tables 512 B.
typedef __attribute__((address_space(3))) void jpgpu_lds_void;
typedef const __attribute__((address_space(1))) void jpgpu_gbl_void;

// PRE: the store already holds SAMPLES (the generic Dispose() pass of a progressive frame whose component slots do not map one
// to one onto its components, dispose_pass_kernel below): no dequantisation, no transform -- the block goes to the writer as it lies
template <int FMT, int LAY, bool PRE>
__device__ __forceinline__ void idct_output_body(
    const int16_t *__restrict__ coefs, const DevScan *__restrict__ scans, const IdctWork *__restrict__ work,
    const DevScanStatus *__restrict__ status, const DevQuantTable *__restrict__ quant_pool, uint8_t *__restrict__ out, YccRgbFactors kf) {
    constexpr int CONV = FMT == kFmtRgbU8 ? 3 : (FMT == kFmtRgbaU8 ? 4 : 0);  // fused YCbCr -> RGB(A), fast layouts and gray only
    __shared__ __attribute__((aligned(16))) uint8_t sh_all[kIdctThreads * 128 + kIdctThreads * 64 + kMaxScanComponents * 128];
    uint8_t *sh = sh_all;
    uint8_t *sh_px = sh_all + kIdctThreads * 128;
    uint16_t(*sh_q)[64] = reinterpret_cast<uint16_t(*)[64]>(sh_all + kIdctThreads * 128 + kIdctThreads * 64);

    const IdctWork wk = work[blockIdx.x];
    const DevScan &s = scans[wk.scan];
    const uint32_t tid = threadIdx.x;
    const uint32_t wave = tid >> 6;
    const uint32_t bpm = s.blocks_per_mcu;
    const uint32_t mcus_per_tile = wk.mcus_per_tile;
    // MCUs the scan never reached (EOI met in a restart check, :144-150): the reference leaves their samples as the caller's
    // buffer held them -- zero in the buffer the batch owns -- so they go through the same output code with zero samples
    uint32_t decoded = status ? status[wk.scan].decoded_mcus : s.total_mcus;
    if (decoded > s.total_mcus) decoded = s.total_mcus;
    uint32_t range_end = wk.first_mcu + wk.n_mcus;
    if ((s.shadow_mask & kKeepUnreachedMcus) != 0) {  // the caller's canvas (jpgpu_decode_scan): unreached MCUs are not touched
        if (range_end > decoded) range_end = decoded;
        if (wk.first_mcu >= range_end) return;
    }

    // quantisation tables of the scan components
    if (tid < (uint32_t)s.scan_components * 32) {
        const uint32_t c = tid >> 5, i = tid & 31;
        reinterpret_cast<uint32_t *>(sh_q[c])[i] =
            reinterpret_cast<const uint32_t *>(quant_pool[s.quant_pool[s.comp[c].quant_slot]].q)[i];
    }

    const uint32_t mcu_local = tid / bpm;
    const uint32_t b = tid - mcu_local * bpm;
    const uint32_t ci_early = s.blk_comp[b < kMaxBlocksPerMcu ? b : 0];
    const uint8_t *coef_bytes = reinterpret_cast<const uint8_t *>(coefs + s.coef_off * 64);

    auto tile_mcus = [&](uint32_t first) { return (range_end - first) < mcus_per_tile ? (range_end - first) : mcus_per_tile; };
    // LDS-DMA of one tile: linear 16-byte slot c = k * 256 + tid (block c >> 3, slot c & 7) receives piece
    // (slot ^ swizzle(block)); the swizzle term ((block >> 1) & 7) does not depend on k, so every lane's source is
    // one fixed offset plus k * 4096.  Always a full tile: the coefficient buffer has a tile of slack behind it.
    const uint32_t tile_blocks = mcus_per_tile * bpm;
    auto dma_tile = [&](uint32_t tile_first) {
        // (recomputed per tile, three instructions, rather than kept in a register across the transform)
        uint32_t t_ = tid;
        asm volatile("" : "+v"(t_));
        const uint32_t dma_lane_off = (t_ >> 3) * 128 + (((t_ & 7) ^ ((t_ >> 4) & 7)) * 16);
        const uint8_t *src = coef_bytes + (uint64_t)tile_first * bpm * 128;  // wave-uniform
#pragma unroll
        for (int k = 0; k < 8; k++)
            if (k < 6 || (uint32_t)k * 32 + (tid >> 3) < tile_blocks)  // blocks behind the tile's last MCU are not fetched
                __builtin_amdgcn_global_load_lds((jpgpu_gbl_void *)(src + (uint32_t)(k * 4096) + dma_lane_off),
                                                 (jpgpu_lds_void *)(sh + ((uint32_t)k * kIdctThreads + wave * 64) * 16), 16, 0, 0);
    };

    dma_tile(wk.first_mcu);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

  for (uint32_t tile_first = wk.first_mcu; tile_first < range_end; tile_first += mcus_per_tile) {
    const uint32_t n_mcu = tile_mcus(tile_first);
    const uint32_t n_blk = n_mcu * bpm;
    const uint32_t next_first = tile_first + mcus_per_tile;
    const bool have_next = next_first < range_end;

    const uint32_t mcu = tile_first + mcu_local;
    const bool have_block = tid < n_blk;
    // a scan component whose frame component a LATER scan component also resolves to: the reference writes its blocks first
    // and the later component's over them (WriteBlock by ComponentIndex, :118-134), so only the later ones reach the output
    const bool writes = have_block && ((s.shadow_mask >> ci_early) & 1u) == 0;

    // phase B1: dequantise this lane's block out of the staging into registers
    float f[64];
    uint32_t px[32];  // int16 sample pairs
    if (PRE) {
        if (have_block) {
#pragma unroll
            for (int piece = 0; piece < 8; piece++) {
                const uint4 cv = *reinterpret_cast<const uint4 *>(sh + tid * 128 + ((piece ^ ((tid >> 1) & 7)) * 16));
                px[piece * 4] = cv.x;
                px[piece * 4 + 1] = cv.y;
                px[piece * 4 + 2] = cv.z;
                px[piece * 4 + 3] = cv.w;
            }
        }
    } else {
        // The lane's eight swizzled staging addresses do not change from tile to tile; hipcc computes them in front of the tile
        // loop -- and in the three variants with the most state in their output assembly spills five of them to scratch for
        // the length of the transform.  There they are derived again in every tile (sixteen instructions) from a copy of the
        // lane id the compiler cannot see through.
        constexpr bool kPerTile = (FMT == kFmtRgbU8 && (LAY == kLayYccH2V1 || LAY == kLayYccH2V2)) || (FMT == kFmtInterleavedU8 && LAY == kLayGeneric);
        uint32_t t_ = tid;
        if (kPerTile) asm volatile("" : "+v"(t_));
        if (have_block) block_dequant(sh + t_ * 128, (t_ >> 1) & 7, sh_q[ci_early], f);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // every lane holds its coefficients: the staging can be refilled
    if (have_next) dma_tile(next_first);  // in flight during the whole transform below

    // phase B2: IDCT entirely in registers
    if (!PRE && have_block) block_idct(f, (int32_t)s.level_shift, px);
    if (mcu >= decoded) {
#pragma unroll
        for (int i = 0; i < 32; i++) px[i] = 0;
    }
    bool synced = false;
    // the MCU's place in the image is only needed from here on: computed behind the transform (an empty asm the compiler may not
    // move across keeps it from being hoisted in front of it), two registers fewer are alive while the 64 + 32 of the
    // transform are -- what four of the sixteen variants spilled (profiles/r03_kernel_resources.txt)
    uint32_t mcu_late = mcu, b_late = b;
    asm volatile("" : "+v"(mcu_late), "+v"(b_late));
    const uint32_t mcu_y = mcu_late / s.mcus_per_line, mcu_x = mcu_late - mcu_y * s.mcus_per_line;
    const uint32_t ci = s.blk_comp[b_late < kMaxBlocksPerMcu ? b_late : 0];  // (again: one byte from the L1-resident descriptor)
    const DevScanComponent comp = s.comp[ci];

    if (FMT == kFmtPlanarI16) {
        // "O1": unclamped int16 at component-native resolution, planes padded to whole MCUs
        if (writes) {
            int16_t *plane = reinterpret_cast<int16_t *>(out + s.out_off + s.plane_off[ci]);
            const uint32_t pitch = s.plane_pitch[ci];
            const uint32_t x0 = (mcu_x * comp.h + s.blk_x[b]) * 8, y0 = (mcu_y * comp.v + s.blk_y[b]) * 8;
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const uint4 v = {px[r * 4 + 0], px[r * 4 + 1], px[r * 4 + 2], px[r * 4 + 3]};
                *reinterpret_cast<uint4 *>(plane + (size_t)(y0 + r) * pitch + x0) = v;
            }
        }
    } else {
    // u8 formats: clamp (signed, like JpegBufferOutputWriter8Bit.ClampTo8Bit) and pack 8 samples per row
    uint2 rows[8];
#pragma unroll
    for (int r = 0; r < 8; r++) {
        rows[r].x = pack4_u8(px[r * 4 + 0], px[r * 4 + 1]);
        rows[r].y = pack4_u8(px[r * 4 + 2], px[r * 4 + 3]);
    }

    if (CONV != 0 && LAY == kLayGray) {
        // a single-component image as R = G = B = Y (Cb = Cr = 128 contribute nothing, DecodeAction.cs:57-65)
        if (writes) {
            const uint32_t x0 = (mcu_x * comp.h + s.blk_x[b]) * 8, y0 = (mcu_y * comp.v + s.blk_y[b]) * 8;
#pragma unroll
            for (int r = 0; r < 8; r++) {
                if (y0 + r >= s.height) continue;
                uint32_t px[8];
#pragma unroll
                for (int j = 0; j < 8; j++) px[j] = byte_of(rows[r].x, rows[r].y, j) * 0x010101u;
                store_rgb_pixels<8, (CONV == 4 ? 4 : 3)>(out + s.out_off + ((size_t)(y0 + r) * s.width + x0) * (CONV == 4 ? 4 : 3), px);
            }
        }
    } else if (FMT == kFmtPlanarU8 || (FMT == kFmtInterleavedU8 && LAY == kLayGray)) {
        // planar u8 (planes padded to whole MCUs), or a single-component interleaved image (same addressing,
        // pitch = W, clipped at the bottom; the host only picks kLayGray when W is a multiple of 8)
        if (writes) {
            const bool gray = (FMT == kFmtInterleavedU8);
            uint8_t *plane = out + s.out_off + (gray ? 0 : s.plane_off[ci]);
            const uint32_t pitch = gray ? s.width : s.plane_pitch[ci];
            const uint32_t x0 = (mcu_x * comp.h + s.blk_x[b]) * 8, y0 = (mcu_y * comp.v + s.blk_y[b]) * 8;
#pragma unroll
            for (int r = 0; r < 8; r++)
                if (!gray || y0 + r < s.height) *reinterpret_cast<uint2 *>(plane + (size_t)(y0 + r) * pitch + x0) = rows[r];
        }
    } else {
    // ---- interleaved u8 ("O2", JpegBufferOutputWriter8Bit semantics): stage the clamped samples in LDS, tile[r][block]
    if (have_block) {
#pragma unroll
        for (int r = 0; r < 8; r++) *reinterpret_cast<uint2 *>(sh_px + r * kPxRowStride + tid * 8) = rows[r];
    }
    // The DMA of the next tile has had the whole transform to land.  Wait for it BEFORE this tile's global stores are
    // issued (vmcnt retires in order: waiting later would also wait for those stores to drain), then one barrier
    // publishes both the sample tile and the refilled staging.
    asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    synced = true;

    interleaved_output_from_tile<LAY, CONV>(sh_px, s, tile_first, n_mcu, tid, writes, comp, mcu_x, mcu_y, b, out, kf);
    }  // interleaved
    }  // u8 formats

    if (!synced) {  // planar / gray paths: publish the refilled staging
        asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
  }  // tile loop
}

// ------------------------------------------------------------------------------------------------
// launch wrappers
// ------------------------------------------------------------------------------------------------

hipError_t launch_extend_u16(hipStream_t stream, const uint8_t *planes, uint8_t *out_base, const ExtendPlanes *images, int n_images,
                             uint32_t max_pixels) {
    if (n_images <= 0 || max_pixels == 0) return hipSuccess;
    const uint32_t bx = (uint32_t)std::min<uint64_t>(((uint64_t)max_pixels + 255) / 256, 4096);
    for (int base = 0; base < n_images; base += 65535) {  // grid.y limit
        const int n = n_images - base < 65535 ? n_images - base : 65535;
        hipLaunchKernelGGL(extend_u16_kernel, dim3(bx, (uint32_t)n), dim3(256), 0, stream, planes, out_base, images + base);
    }
    return hipGetLastError();
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

// K2 lookup width: 11 bits when the scan stages at most 4 tables, 10 bits for up to 8 (LDS budget: 160 KB per CU)
static size_t k2_lds_bytes(int n_slots, int waves) {
    return (size_t)n_slots * kK2TabBytes + (size_t)waves * kK2WaveBytes + kMaxBlocksPerMcu * sizeof(uint32_t);
}
size_t huffman_lds_bytes(int n_slots) { return k2_lds_bytes(n_slots, huffman_waves(n_slots)); }

// More than 64 KB of dynamic LDS has to be allowed per kernel -- and per DEVICE: a process that drives several devices (one
// jpgpu_ctx each) must do it on each of them.  done: one bit per device ordinal.
static hipError_t allow_dynamic_lds(const void *kernel, int bytes, std::atomic<uint64_t> &done) {
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

template <int WAVES>
static hipError_t launch_huffman_w(hipStream_t stream, const uint8_t *data, const DevScan *scans, const HuffWork *work, int n_work,
                                   const uint32_t *ends, DevScanStatus *status, const DevHuffTable *huff_pool, int16_t *coefs,
                                   int n_slots, const uint8_t *lut_pool) {
    const size_t lds = k2_lds_bytes(n_slots, WAVES);
    static std::atomic<uint64_t> configured{0};
    const hipError_t ea = allow_dynamic_lds(reinterpret_cast<const void *>(&huffman_decode_kernel<WAVES>), 160 * 1024, configured);
    if (ea != hipSuccess) return ea;
    hipLaunchKernelGGL((huffman_decode_kernel<WAVES>), dim3(n_work), dim3(64 * WAVES), lds, stream, data, scans, work, ends, status, huff_pool, coefs,
                       n_slots, lut_pool);
    return hipGetLastError();
}

// `work` holds one entry per huffman_waves(n_slots) * 64 restart intervals (DeviceBatch builds it with the same function)
hipError_t launch_huffman(hipStream_t stream, const uint8_t *data, const DevScan *scans, const HuffWork *work, int n_work,
                          const uint32_t *ends, DevScanStatus *status, const DevHuffTable *huff_pool, int16_t *coefs,
                          int n_slots, const uint8_t *lut_pool) {
    if (n_work <= 0) return hipSuccess;
    if (huffman_waves(n_slots) == kHuffWaves)
        return launch_huffman_w<kHuffWaves>(stream, data, scans, work, n_work, ends, status, huff_pool, coefs, n_slots, lut_pool);
    return launch_huffman_w<kHuffWavesManyTables>(stream, data, scans, work, n_work, ends, status, huff_pool, coefs, n_slots, lut_pool);
}

// Fused lookups of every table of the pool (once per upload: the tables of a batch do not change between decodes).
hipError_t launch_lut_pool(hipStream_t stream, const DevHuffTable *huff_pool, int n_tables, uint8_t *lut_pool) {
    if (n_tables <= 0) return hipSuccess;
    hipLaunchKernelGGL(lut_pool_kernel, dim3(2 * n_tables), dim3(256), 0, stream, huff_pool, lut_pool);
    return hipGetLastError();
}

template <int FMT, int LAY>
__global__ __launch_bounds__(kIdctThreads, (FMT == kFmtPlanarI16 ? 2 : 3)) void idct_output_kernel(
    const int16_t *__restrict__ coefs, const DevScan *__restrict__ scans, const IdctWork *__restrict__ work,
    const DevScanStatus *__restrict__ status, const DevQuantTable *__restrict__ quant_pool, uint8_t *__restrict__ out, YccRgbFactors kf) {
    idct_output_body<FMT, LAY, false>(coefs, scans, work, status, quant_pool, out, kf);
}
template <int FMT>
__global__ __launch_bounds__(kIdctThreads, 2) void flush_output_kernel(
    const int16_t *__restrict__ coefs, const DevScan *__restrict__ scans, const IdctWork *__restrict__ work,
    const DevScanStatus *__restrict__ status, const DevQuantTable *__restrict__ quant_pool, uint8_t *__restrict__ out, YccRgbFactors kf) {
    idct_output_body<FMT, kLayGeneric, true>(coefs, scans, work, status, quant_pool, out, kf);
}

// The reference's Dispose() as it is written (ScanDecoder/JpegHuffmanProgressiveScanDecoder.cs:421-470): every component SLOT of
// the scan decoder, as the last scans left it, dequantises + transforms + level-shifts the blocks of its component IN PLACE.  For
// files in the usual scan order that is one transform per component and K3 does it on the way to the writer.  When the slots
// do not map one to one onto the components (a file whose last scan of slot 0 is not the first component: e.g. slots
// {Cb, Cb, Cr}) a component is transformed twice -- the second time reading its own samples as zig-zag coefficients -- and
// another never (its quantised coefficients reach the writer as samples); a file without any scan flushes zeros.  This kernel
// does literally that to the frame's store, one lane per block, `n` transforms with the slots' tables in slot order;
// flush_output_kernel then writes the store out.  (Both also serve the partial flush of a progressive file that failed.)
__global__ __launch_bounds__(64) void dispose_pass_kernel(int16_t *__restrict__ coefs, const DisposeJob *__restrict__ jobs,
                                                          const DevQuantTable *__restrict__ quant_pool) {
    const DisposeJob &j = jobs[blockIdx.y];
    const uint32_t g = blockIdx.x * 64u + threadIdx.x;
    if (g >= j.n_blocks) return;
    const uint32_t c = j.blk_comp[g % j.bpm];
    const uint32_t n = j.n[c];
    if (n == 0) return;
    int16_t *blk = coefs + (j.coef_off + g) * 64;
    uint32_t w[32];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const uint4 v = reinterpret_cast<const uint4 *>(blk)[i];
        w[i * 4] = v.x;
        w[i * 4 + 1] = v.y;
        w[i * 4 + 2] = v.z;
        w[i * 4 + 3] = v.w;
    }
    for (uint32_t t = 0; t < n; t++) {
        const uint16_t *q = quant_pool[j.quant[c][t]].q;
        float f[64];
#pragma unroll
        for (int k = 0; k < 64; k++) {
            const int32_t cv = (k & 1) ? ((int32_t)w[k >> 1] >> 16) : (int32_t)(int16_t)(w[k >> 1] & 0xFFFFu);
            f[kNat[k]] = (float)((int32_t)q[k] * cv);  // ushort * short -> int -> float (DequantizeBlockAndUnZigZag)
        }
        block_idct(f, (int32_t)j.level_shift, w);
    }
#pragma unroll
    for (int i = 0; i < 8; i++) reinterpret_cast<uint4 *>(blk)[i] = make_uint4(w[i * 4], w[i * 4 + 1], w[i * 4 + 2], w[i * 4 + 3]);
}

template <int FMT, int LAY>
static void launch_idct_one(hipStream_t stream, const int16_t *coefs, const DevScan *scans, const IdctWork *work, int n_work,
                            const DevScanStatus *status, const DevQuantTable *quant_pool, uint8_t *out,
                            YccRgbFactors kf = YccRgbFactors{0, 0, 0, 0}) {
    hipLaunchKernelGGL((idct_output_kernel<FMT, LAY>), dim3(n_work), dim3(kIdctThreads), 0, stream, coefs, scans, work, status,
                       quant_pool, out, kf);
}

// work is sorted by layout class; class_begin[c]..class_begin[c+1] are the workgroups of class c.
hipError_t launch_idct(hipStream_t stream, const int16_t *coefs, const DevScan *scans, const IdctWork *work,
                       const int class_begin[kNumIdctLayoutClasses + 1], const DevScanStatus *status,
                       const DevQuantTable *quant_pool, uint8_t *out, int format, const YccRgbFactors &kf, uint8_t *generic_out) {
    for (int c = 0; c < kNumIdctLayoutClasses; c++) {
        const int n = class_begin[c + 1] - class_begin[c];
        if (n <= 0) continue;
        const IdctWork *w = work + class_begin[c];
        if (c == kIdctClassStoreHoldsSamples) {  // the generic Dispose() pass has run: the store goes to the writer as it is
            if (format == kFmtPlanarI16) hipLaunchKernelGGL((flush_output_kernel<kFmtPlanarI16>), dim3(n), dim3(kIdctThreads), 0, stream, coefs, scans, w, status, quant_pool, out, kf);
            else if (format == kFmtPlanarU8) hipLaunchKernelGGL((flush_output_kernel<kFmtPlanarU8>), dim3(n), dim3(kIdctThreads), 0, stream, coefs, scans, w, status, quant_pool, out, kf);
            else hipLaunchKernelGGL((flush_output_kernel<kFmtInterleavedU8>), dim3(n), dim3(kIdctThreads), 0, stream, coefs, scans, w, status, quant_pool,
                                    (format == kFmtRgbU8 || format == kFmtRgbaU8) ? generic_out : out, kf);
        } else if (format == kFmtPlanarI16) {
            launch_idct_one<kFmtPlanarI16, kLayGeneric>(stream, coefs, scans, w, n, status, quant_pool, out);
        } else if (format == kFmtPlanarU8) {
            launch_idct_one<kFmtPlanarU8, kLayGeneric>(stream, coefs, scans, w, n, status, quant_pool, out);
        } else if (format == kFmtRgbU8) {
            switch (c) {
            case kLayYccH1V1: launch_idct_one<kFmtRgbU8, kLayYccH1V1>(stream, coefs, scans, w, n, status, quant_pool, out, kf); break;
            case kLayYccH2V1: launch_idct_one<kFmtRgbU8, kLayYccH2V1>(stream, coefs, scans, w, n, status, quant_pool, out, kf); break;
            case kLayYccH2V2: launch_idct_one<kFmtRgbU8, kLayYccH2V2>(stream, coefs, scans, w, n, status, quant_pool, out, kf); break;
            case kLayGray: launch_idct_one<kFmtRgbU8, kLayGray>(stream, coefs, scans, w, n, status, quant_pool, out, kf); break;
            // no fused path: the samples go to `generic_out` as INTERLEAVED_U8 and are converted by launch_ycc_to_rgb
            default: launch_idct_one<kFmtInterleavedU8, kLayGeneric>(stream, coefs, scans, w, n, status, quant_pool, generic_out); break;
            }
        } else if (format == kFmtRgbaU8) {
            switch (c) {
            case kLayYccH1V1: launch_idct_one<kFmtRgbaU8, kLayYccH1V1>(stream, coefs, scans, w, n, status, quant_pool, out, kf); break;
            case kLayYccH2V1: launch_idct_one<kFmtRgbaU8, kLayYccH2V1>(stream, coefs, scans, w, n, status, quant_pool, out, kf); break;
            case kLayYccH2V2: launch_idct_one<kFmtRgbaU8, kLayYccH2V2>(stream, coefs, scans, w, n, status, quant_pool, out, kf); break;
            case kLayGray: launch_idct_one<kFmtRgbaU8, kLayGray>(stream, coefs, scans, w, n, status, quant_pool, out, kf); break;
            default: launch_idct_one<kFmtInterleavedU8, kLayGeneric>(stream, coefs, scans, w, n, status, quant_pool, generic_out); break;
            }
        } else {
            switch (c) {
            case kLayYccH1V1: launch_idct_one<kFmtInterleavedU8, kLayYccH1V1>(stream, coefs, scans, w, n, status, quant_pool, out); break;
            case kLayYccH2V1: launch_idct_one<kFmtInterleavedU8, kLayYccH2V1>(stream, coefs, scans, w, n, status, quant_pool, out); break;
            case kLayYccH2V2: launch_idct_one<kFmtInterleavedU8, kLayYccH2V2>(stream, coefs, scans, w, n, status, quant_pool, out); break;
            case kLayGray: launch_idct_one<kFmtInterleavedU8, kLayGray>(stream, coefs, scans, w, n, status, quant_pool, out); break;
            default: launch_idct_one<kFmtInterleavedU8, kLayGeneric>(stream, coefs, scans, w, n, status, quant_pool, out); break;
            }
        }
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

// INTERLEAVED_U8 image (comps = 1 or 3) -> RGB / RGBA, for the layouts without a fused path
hipError_t launch_ycc_to_rgb(hipStream_t stream, const uint8_t *src, uint8_t *dst, uint64_t n_pixels, int comps, int bpp, const YccRgbFactors &kf) {
    if (n_pixels == 0) return hipSuccess;
    const uint64_t want = (n_pixels + 255) / 256;
    const int grid = (int)(want < 65536 ? want : 65536);
    hipLaunchKernelGGL(ycc_to_rgb_kernel, dim3(grid), dim3(256), 0, stream, src, dst, n_pixels, comps, bpp, kf);
    return hipGetLastError();
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

// DRI = 0 scans: self-synchronising subsequence decode.  `work` lists (scan, first subsequence) per workgroup of 256 lanes;
// `scan_ids` the scans concerned.  Runs synchronisation rounds until no exit state changes (host-checked flag).
// The synchronisation part alone (rounds until the exit states stop changing, then the block / DC prefix sums); leaves the
// converged exit states in *final_state.  Shared by the decoder (subseq_final_kernel) and the optimizer (subseq_transcode_kernel).
hipError_t launch_subseq_sync(hipStream_t stream, const uint8_t *udata, const DevScan *scans, const HuffWork *work, int n_work,
                              const uint32_t *scan_ids, int n_scans, const uint32_t *ends_u, DevScanStatus *status,
                              const DevHuffTable *huff_pool, uint32_t *exit_a, uint32_t *exit_b, uint32_t *nblk, uint32_t *first_block,
                              uint32_t *entry_used, void *dcsum, void *dc_entry, uint32_t *changed_dev, int n_slots, int max_rounds,
                              int *rounds_used, const uint8_t *lut_pool, const uint32_t **final_state_out, uint32_t *same_dist,
                              bool *same_valid) {
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
    hipError_t e = hipMemsetAsync(changed_dev, 0, 64 * sizeof(uint32_t), stream);
    if (e != hipSuccess) return e;
    int round = 0;
    bool converged = false;
    uint32_t last_count = 0xFFFFFFFFu, count_before = 0xFFFFFFFFu;  // exits the last two rounds changed
    while (!converged && round < max_rounds) {
        const int batch_first = round;
        const int batch = last_count <= few_changes ? 1 : kCheckEvery;
        for (int i = 0; i < batch && round < max_rounds; i++, round++) {
            const uint32_t *in = bufs[(round + 1) & 1];
            uint32_t *out = bufs[round & 1];
            hipLaunchKernelGGL(subseq_round_kernel, dim3(n_work), dim3(256), lds_round, stream, udata, scans, work, ends_u, status, huff_pool,
                               lut_pool, in, out, nblk, entry_used, (int4 *)dcsum, changed_dev + (round % 62), round, n_slots, warm_bits);
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
                    hipLaunchKernelGGL(subseq_same_kernel, dim3(n_work), dim3(256), 0, stream, udata, scans, work, ends_u, status, same_dist);
                    *same_valid = true;
                }
                hipLaunchKernelGGL(subseq_propagate_kernel, dim3(n_scans), dim3(64), 0, stream, scans, scan_ids, same_dist, bufs[(round + 1) & 1], entry_used,
                                   nblk, (int4 *)dcsum, changed_dev + 63);
            }
        }
    }
    if (rounds_used) *rounds_used = round;
    *final_state_out = bufs[(round + 1) & 1];  // buffer written by the last round executed
    hipLaunchKernelGGL(subseq_scan_kernel, dim3(n_scans), dim3(1024), 0, stream, scans, scan_ids, nblk, first_block, (const int4 *)dcsum,
                       (int4 *)dc_entry);
    return hipGetLastError();
}

hipError_t launch_subseq_decode(hipStream_t stream, const uint8_t *udata, const DevScan *scans, const HuffWork *work, int n_work,
                                const uint32_t *scan_ids, int n_scans, const uint32_t *ends_u, DevScanStatus *status,
                                const DevHuffTable *huff_pool, uint32_t *exit_a, uint32_t *exit_b, uint32_t *nblk, uint32_t *first_block,
                                uint32_t *entry_used, void *dcsum, void *dc_entry, uint32_t *changed_dev, int16_t *coefs, int n_slots,
                                int max_rounds, int *rounds_used, const uint8_t *lut_pool, const HuffWork *final_work, int n_final_work,
                                uint32_t *same_dist, bool *same_valid) {
    if (n_work <= 0 || n_scans <= 0) return hipSuccess;
    const uint32_t *final_state = nullptr;
    hipError_t e = launch_subseq_sync(stream, udata, scans, work, n_work, scan_ids, n_scans, ends_u, status, huff_pool, exit_a, exit_b, nblk,
                                      first_block, entry_used, dcsum, dc_entry, changed_dev, n_slots, max_rounds, rounds_used, lut_pool,
                                      &final_state, same_dist, same_valid);
    if (e != hipSuccess) return e;
    const int waves = subseq_final_waves(n_slots);
    const size_t lds_final = (size_t)n_slots * kK2TabBytes + (size_t)waves * kSfWaveBytes + kMaxBlocksPerMcu * sizeof(uint32_t);
    static std::atomic<uint64_t> configured{0};
    const hipError_t ea = allow_dynamic_lds(reinterpret_cast<const void *>(&subseq_final_kernel), 160 * 1024, configured);
    if (ea != hipSuccess) return ea;
    // final_work: (scan, first subsequence) per workgroup of waves * 64 lanes (the rounds' work list is per 256)
    hipLaunchKernelGGL(subseq_final_kernel, dim3(n_final_work), dim3(64 * waves), lds_final, stream, udata, scans, final_work, ends_u, status,
                       huff_pool, lut_pool, final_state, first_block, (const int4 *)dc_entry, coefs, n_slots);
    return hipGetLastError();
}

hipError_t launch_dispose_pass(hipStream_t stream, int16_t *coefs, const DisposeJob *jobs, int n_jobs, uint32_t max_blocks, const DevQuantTable *quant_pool) {
    if (n_jobs <= 0 || max_blocks == 0) return hipSuccess;
    hipLaunchKernelGGL(dispose_pass_kernel, dim3((max_blocks + 63u) / 64u, n_jobs), dim3(64), 0, stream, coefs, jobs, quant_pool);
    return hipGetLastError();
}

// Layout class of a scan for the INTERLEAVED_U8 format (0 = generic bytewise path).
int idct_layout_class(const DevScan &s) {
    const uint32_t W = s.width;
    if (s.frame_components == 1 && s.scan_components == 1 && s.comp[0].h == 1 && s.comp[0].v == 1 && (W % 8) == 0 && (s.out_off % 8) == 0)
        return kLayGray;
    const bool ycc = s.frame_components == 3 && s.scan_components == 3 && s.comp[0].component_index == 0 &&
                     s.comp[1].component_index == 1 && s.comp[2].component_index == 2 && s.comp[0].hs == 1 && s.comp[0].vs == 1 &&
                     s.comp[1].h == 1 && s.comp[1].v == 1 && s.comp[2].h == 1 && s.comp[2].v == 1;
    if (!ycc) return kLayGeneric;
    if (s.max_h == 1 && s.max_v == 1 && (W % 8) == 0 && (s.out_off % 8) == 0) return kLayYccH1V1;
    if (s.max_h == 2 && (W % 16) == 0 && (s.out_off % 16) == 0) {
        if (s.max_v == 1) return kLayYccH2V1;
        if (s.max_v == 2) return kLayYccH2V2;
    }
    return kLayGeneric;
}


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
