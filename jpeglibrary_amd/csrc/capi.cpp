// jpeglibrary_amd/csrc/capi.cpp -- extern "C" entry points declared in include/jpgpu.h.
#include <hip/hip_runtime.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <exception>
#include <memory>
#include <new>
#include <thread>
#include <vector>

#include "../../include/jpgpu.h"
#include "device_batch.h"
#include "device_encode.h"
#include "device_optimize.h"
#include "host.h"

using namespace jpgpu;

struct jpgpu_batch {
    jpgpu_ctx *ctx;
    DeviceBatch impl;
    explicit jpgpu_batch(jpgpu_ctx *c) : ctx(c), impl(c) {}
};

static thread_local std::string g_create_error;

extern "C" {

int jpgpu_version(void) { return JPGPU_VERSION; }
size_t jpgpu_sizeof_image_result(void) { return sizeof(jpgpu_image_result); }

int jpgpu_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int jpgpu_create(int device, jpgpu_ctx **out) {
    if (!out) return JPGPU_ERR_ARGUMENT;
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        g_create_error = "No HIP device is visible (libjpgpu has no CPU fallback).";
        return JPGPU_ERR_NO_DEVICE;
    }
    if (device < 0 || device >= n) {
        g_create_error = "Device index out of range.";
        return JPGPU_ERR_ARGUMENT;
    }
    if ((e = hipSetDevice(device)) != hipSuccess) {
        g_create_error = std::string("hipSetDevice: ") + hipGetErrorString(e);
        return JPGPU_ERR_DEVICE;
    }
    std::unique_ptr<jpgpu_ctx> ctx(new (std::nothrow) jpgpu_ctx());
    if (!ctx) return JPGPU_ERR_OUT_OF_MEMORY;
    ctx->device = device;
    if ((e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)) != hipSuccess) {
        g_create_error = std::string("hipStreamCreate: ") + hipGetErrorString(e);
        return JPGPU_ERR_DEVICE;
    }
    if ((e = hipStreamCreateWithFlags(&ctx->upload_stream, hipStreamNonBlocking)) != hipSuccess ||
        (e = hipStreamCreateWithFlags(&ctx->stream2, hipStreamNonBlocking)) != hipSuccess) {
        g_create_error = std::string("hipStreamCreate: ") + hipGetErrorString(e);
        (void)hipStreamDestroy(ctx->stream);
        if (ctx->upload_stream) (void)hipStreamDestroy(ctx->upload_stream);
        return JPGPU_ERR_DEVICE;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) {
        ctx->num_cus = prop.multiProcessorCount;
        ctx->device_bytes = (uint64_t)prop.totalGlobalMem;
    }
    if (const char *ev = getenv("JPGPU_STAGING_SLOTS")) ctx->staging.n_slots = std::min((int)StagingRing::kMaxSlots, std::max(2, atoi(ev)));
    if (const char *ev = getenv("JPGPU_STAGING_SLOT_MB")) ctx->staging.slot_bytes = (size_t)std::min(256, std::max(1, atoi(ev))) << 20;
    *out = ctx.release();
    return JPGPU_OK;
}

void jpgpu_destroy(jpgpu_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    if (ctx->upload_stream) (void)hipStreamDestroy(ctx->upload_stream);
    if (ctx->stream2) (void)hipStreamDestroy(ctx->stream2);
    for (hipStream_t st : ctx->prog_stream)
        if (st) (void)hipStreamDestroy(st);
    for (hipEvent_t ev : ctx->prog_ev)
        if (ev) (void)hipEventDestroy(ev);
    for (int i = 0; i < StagingRing::kMaxSlots; i++) {
        if (ctx->staging.slot[i]) (void)hipHostFree(ctx->staging.slot[i]);
        if (ctx->staging.drained[i]) (void)hipEventDestroy(ctx->staging.drained[i]);
    }
    if (ctx->staging.verdict) (void)hipHostFree(ctx->staging.verdict);
    delete ctx;
}

int jpgpu_set_host_threads(jpgpu_ctx *ctx, int threads) {
    if (!ctx || threads < 0) return JPGPU_ERR_ARGUMENT;
    ctx->host_threads = threads;
    return JPGPU_OK;
}

void jpgpu_shard(int n_items, int rank, int world, int *first, int *stride, int *count) {
    // image i -> GPU i mod G: rank r takes items r, r + world, r + 2 world, ...
    if (world < 1) world = 1;
    if (rank < 0) rank = 0;
    if (first) *first = rank;
    if (stride) *stride = world;
    if (count) *count = n_items > rank ? (n_items - rank + world - 1) / world : 0;
}

const char *jpgpu_last_error(const jpgpu_ctx *ctx) { return ctx ? ctx->last_error.c_str() : g_create_error.c_str(); }

// ---------------------------------------------------------------------------------------------- page-locked host memory
static int host_mem_call(jpgpu_ctx *ctx, hipError_t e, const char *what) {
    if (e == hipSuccess) return JPGPU_OK;
    if (ctx) ctx->last_error = std::string(what) + ": " + hipGetErrorString(e);
    (void)hipGetLastError();
    return e == hipErrorOutOfMemory ? JPGPU_ERR_OUT_OF_MEMORY : JPGPU_ERR_DEVICE;
}
int jpgpu_host_alloc(jpgpu_ctx *ctx, size_t bytes, void **out) {
    if (!ctx || !out) return JPGPU_ERR_ARGUMENT;
    *out = nullptr;
    (void)hipSetDevice(ctx->device);
    return host_mem_call(ctx, hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocPortable | hipHostMallocMapped), "hipHostMalloc");
}
int jpgpu_host_free(jpgpu_ctx *ctx, void *p) {
    if (!ctx) return JPGPU_ERR_ARGUMENT;
    if (!p) return JPGPU_OK;
    return host_mem_call(ctx, hipHostFree(p), "hipHostFree");
}
int jpgpu_host_register(jpgpu_ctx *ctx, void *p, size_t bytes) {
    if (!ctx || !p || !bytes) return JPGPU_ERR_ARGUMENT;
    (void)hipSetDevice(ctx->device);
    return host_mem_call(ctx, hipHostRegister(p, bytes, hipHostRegisterPortable | hipHostRegisterMapped), "hipHostRegister");
}
int jpgpu_host_unregister(jpgpu_ctx *ctx, void *p) {
    if (!ctx || !p) return JPGPU_ERR_ARGUMENT;
    return host_mem_call(ctx, hipHostUnregister(p), "hipHostUnregister");
}

// ---------------------------------------------------------------------------------------------- several devices
// one context + two batches + one host thread per device slot, image i on slot i mod G, nothing shared.
// Two batches per slot: call k + 1 is parsed and sent to HBM (upload stream) while call k decodes (decode stream).
struct jpgpu_multi {
    struct Slot {
        jpgpu_ctx *ctx = nullptr;
        jpgpu_batch *batch[2] = {nullptr, nullptr};
        std::vector<jpgpu_segment> segs;
        std::vector<int> per_file;
        int status[2] = {JPGPU_OK, JPGPU_OK};
        double upload_ms[2] = {0, 0}, decode_ms[2] = {0, 0};
        std::chrono::steady_clock::time_point launched[2];
    };
    std::vector<Slot> slots;
    int n_items[2] = {0, 0};
    int next_ticket = 0;      // tickets count up; ticket t lives in batch t & 1
    int last_ticket = -1;
    bool pending[2] = {false, false};
    int ticket_in[2] = {-1, -1};  // the ticket whose call lives in batch k
    std::string last_error;
};

int jpgpu_multi_create(const int *devices, int n_devices, jpgpu_multi **out) {
    if (!out) return JPGPU_ERR_ARGUMENT;
    *out = nullptr;
    if (!devices || n_devices <= 0) {
        g_create_error = "jpgpu_multi_create: no devices";
        return JPGPU_ERR_ARGUMENT;
    }
    std::unique_ptr<jpgpu_multi> m(new (std::nothrow) jpgpu_multi);
    if (!m) return JPGPU_ERR_OUT_OF_MEMORY;
    m->slots.resize((size_t)n_devices);
    // G slots share the CPUs the process is granted: each slot's crew gets its share (an explicit JPGPU_HOST_THREADS is per slot)
    const int crew = getenv("JPGPU_HOST_THREADS") ? 0 : std::max(1, std::min(16, jpgpu::granted_host_cpus() / n_devices));
    int rc = JPGPU_OK;
    for (int s = 0; s < n_devices && rc == JPGPU_OK; s++) {
        jpgpu_multi::Slot &sl = m->slots[(size_t)s];
        rc = jpgpu_create(devices[s], &sl.ctx);
        if (rc == JPGPU_OK) sl.ctx->host_threads = crew;
        for (int k = 0; k < 2 && rc == JPGPU_OK; k++) rc = jpgpu_batch_create(sl.ctx, &sl.batch[k]);
    }
    if (rc != JPGPU_OK) {
        jpgpu_multi_destroy(m.release());
        return rc;
    }
    *out = m.release();
    return JPGPU_OK;
}

void jpgpu_multi_destroy(jpgpu_multi *m) {
    if (!m) return;
    for (jpgpu_multi::Slot &s : m->slots) {
        for (jpgpu_batch *b : s.batch)
            if (b) jpgpu_batch_destroy(b);
        if (s.ctx) jpgpu_destroy(s.ctx);
    }
    delete m;
}

int jpgpu_multi_devices(const jpgpu_multi *m) { return m ? (int)m->slots.size() : 0; }
const char *jpgpu_multi_last_error(const jpgpu_multi *m) { return m ? m->last_error.c_str() : g_create_error.c_str(); }

int jpgpu_multi_submit(jpgpu_multi *m, const uint8_t *const *jpeg, const size_t *len, int n, int format, unsigned flags, int *ticket) {
    if (!m || n < 0 || (n > 0 && (!jpeg || !len))) return JPGPU_ERR_ARGUMENT;
    const int world = (int)m->slots.size();
    const int t = m->next_ticket, k = t & 1;
    m->last_error.clear();
    if (m->pending[k]) {
        m->last_error = "jpgpu_multi_submit: two calls are in flight already (wait for the older one first)";
        return JPGPU_ERR_INVALID_OPERATION;
    }
    for (int s = 0; s < world; s++) {
        jpgpu_multi::Slot &sl = m->slots[(size_t)s];
        int first, stride, count;
        jpgpu_shard(n, s, world, &first, &stride, &count);
        sl.segs.clear();
        for (int j = 0, i = first; j < count; j++, i += stride) sl.segs.push_back({jpeg[i], len[i]});
        sl.per_file.assign((size_t)count, 1);
        sl.status[k] = JPGPU_OK;
        sl.upload_ms[k] = sl.decode_ms[k] = 0;
    }
    // one host thread per device slot: upload (header parse, staging or pinned DMA, H2D) and the decode's launches; the shards
    // never meet.  The threads are joined before this returns: what stays in flight is device work only.
    auto run = [format, flags, k](jpgpu_multi::Slot *sl) {
        using clk = std::chrono::steady_clock;
        const auto t0 = clk::now();
        int rc = jpgpu_batch_upload_segments(sl->batch[k], sl->segs.data(), sl->per_file.data(), (int)sl->per_file.size(), format, flags);
        const auto t1 = clk::now();
        if (rc == JPGPU_OK) rc = jpgpu_batch_decode(sl->batch[k]);
        sl->status[k] = rc;
        sl->upload_ms[k] = std::chrono::duration<double, std::milli>(t1 - t0).count();
        sl->launched[k] = t1;
    };
    {
        std::vector<std::thread> threads;
        struct Joiner {  // a thread that failed to start (EAGAIN) must not leave started ones joinable behind it
            std::vector<std::thread> &t;
            ~Joiner() {
                for (std::thread &x : t)
                    if (x.joinable()) x.join();
            }
        } joiner{threads};
        threads.reserve((size_t)world);
        int started = 1;
        try {
            for (int s = 1; s < world; s++, started++) threads.emplace_back(run, &m->slots[(size_t)s]);
        } catch (const std::exception &) {
            // the slots without a thread of their own run on this one, after slot 0
        }
        run(&m->slots[0]);
        for (int s = started; s < world; s++) run(&m->slots[(size_t)s]);
    }
    int rc = JPGPU_OK;
    for (int s = 0; s < world && rc == JPGPU_OK; s++)
        if (m->slots[(size_t)s].status[k] != JPGPU_OK) {
            rc = m->slots[(size_t)s].status[k];
            m->last_error = std::string("device slot ") + std::to_string(s) + ": " + jpgpu_last_error(m->slots[(size_t)s].ctx);
        }
    if (rc != JPGPU_OK) {
        // a failed call takes no ticket: the slots that did launch are waited for here, so that nothing the caller cannot
        // address stays in flight and batch k is free for the next submit
        for (int s = 0; s < world; s++)
            if (m->slots[(size_t)s].status[k] == JPGPU_OK) (void)jpgpu_batch_sync(m->slots[(size_t)s].batch[k]);
        if (ticket) *ticket = -1;
        return rc;
    }
    m->n_items[k] = n;
    m->pending[k] = true;
    m->ticket_in[k] = t;
    m->last_ticket = t;
    m->next_ticket++;
    if (ticket) *ticket = t;
    return JPGPU_OK;
}

int jpgpu_multi_wait(jpgpu_multi *m, int ticket, double *upload_ms, double *decode_ms) {
    if (!m || ticket < 0 || ticket >= m->next_ticket || ticket + 2 < m->next_ticket) return JPGPU_ERR_ARGUMENT;
    const int k = ticket & 1;
    const int world = (int)m->slots.size();
    int rc = JPGPU_OK;
    double up = 0, dec = 0;
    for (int s = 0; s < world; s++) {
        jpgpu_multi::Slot &sl = m->slots[(size_t)s];
        if (m->pending[k] && sl.status[k] == JPGPU_OK) {
            sl.status[k] = jpgpu_batch_sync(sl.batch[k]);
            sl.decode_ms[k] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - sl.launched[k]).count();
        }
        if (sl.status[k] != JPGPU_OK && rc == JPGPU_OK) {
            rc = sl.status[k];
            m->last_error = std::string("device slot ") + std::to_string(s) + ": " + jpgpu_last_error(sl.ctx);
        }
        up = std::max(up, sl.upload_ms[k]);
        dec = std::max(dec, sl.decode_ms[k]);
    }
    m->pending[k] = false;
    if (upload_ms) *upload_ms = up;
    if (decode_ms) *decode_ms = dec;
    return rc;
}

int jpgpu_multi_decode(jpgpu_multi *m, const uint8_t *const *jpeg, const size_t *len, int n, int format, double *upload_ms,
                       double *decode_ms) {
    if (!m) return JPGPU_ERR_ARGUMENT;
    // the synchronous form: nothing else may be in flight behind it
    for (int k = 0; k < 2; k++)
        if (m->pending[k]) (void)jpgpu_multi_wait(m, m->ticket_in[k], nullptr, nullptr);
    int ticket = -1;
    const int rc = jpgpu_multi_submit(m, jpeg, len, n, format, 0, &ticket);
    if (ticket < 0) return rc;
    const int rc2 = jpgpu_multi_wait(m, ticket, upload_ms, decode_ms);
    return rc != JPGPU_OK ? rc : rc2;
}

int jpgpu_multi_locate(const jpgpu_multi *m, int i, int *slot, int *local_index) {
    if (!m || m->last_ticket < 0 || i < 0 || i >= m->n_items[m->last_ticket & 1]) return JPGPU_ERR_ARGUMENT;
    const int world = (int)m->slots.size();
    if (slot) *slot = i % world;
    if (local_index) *local_index = i / world;
    return JPGPU_OK;
}
jpgpu_batch *jpgpu_multi_batch_of(jpgpu_multi *m, int ticket, int slot) {
    if (!m || slot < 0 || slot >= (int)m->slots.size() || ticket < 0 || ticket >= m->next_ticket || ticket + 2 < m->next_ticket) return nullptr;
    return m->slots[(size_t)slot].batch[ticket & 1];
}
jpgpu_batch *jpgpu_multi_batch(jpgpu_multi *m, int slot) { return m ? jpgpu_multi_batch_of(m, m->last_ticket, slot) : nullptr; }
jpgpu_ctx *jpgpu_multi_context(jpgpu_multi *m, int slot) { return m && slot >= 0 && slot < (int)m->slots.size() ? m->slots[(size_t)slot].ctx : nullptr; }

const char *jpgpu_status_string(int status) {
    switch (status) {
    case JPGPU_OK: return "OK";
    case JPGPU_ERR_INVALID_DATA: return "InvalidDataException";
    case JPGPU_ERR_INVALID_OPERATION: return "InvalidOperationException";
    case JPGPU_ERR_NOT_SUPPORTED: return "NotSupportedException";
    case JPGPU_ERR_ARGUMENT: return "ArgumentException";
    case JPGPU_ERR_DEVICE: return "HIP device error";
    case JPGPU_ERR_NO_DEVICE: return "no HIP device";
    case JPGPU_ERR_OUT_OF_MEMORY: return "out of memory";
    default: return "unknown status";
    }
}

const char *jpgpu_detail_string(int detail) {
    switch (detail) {
    case JPGPU_DETAIL_NONE: return "";
    case JPGPU_DETAIL_INVALID_HUFFMAN_CODE: return "Invalid Huffman code encountered.";
    case JPGPU_DETAIL_MARKER_IN_DATA: return "Failed to decode JPEG data. Expect raw data from bit stream. Yet a marker is encountered.";
    case JPGPU_DETAIL_STREAM_ENDED: return "Failed to decode JPEG data. The bit stream ended prematurely.";
    case JPGPU_DETAIL_EXPECT_RESTART: return "Expect restart marker.";
    case JPGPU_DETAIL_MISSING_TABLE: return "Failed to decode JPEG data. A Huffman or quantization table is not defined.";
    case JPGPU_DETAIL_UNSUPPORTED_FRAME: return "This type of JPEG stream is not supported on the GPU path.";
    case JPGPU_DETAIL_BAD_HEADER: return "Failed to decode JPEG data. Malformed marker segment.";
    case JPGPU_DETAIL_EARLY_EOI: return "EndOfImage met at a restart boundary; image partially decoded.";
    case JPGPU_DETAIL_UNEXPECTED_END: return "Failed to decode JPEG data. Unexpected end of JPEG data stream.";
    case JPGPU_DETAIL_NULL_TABLE: return "Object reference not set to an instance of an object.";
    default: return "unknown detail";
    }
}

// ------------------------------------------------------------------------------------------------ batch

int jpgpu_batch_create(jpgpu_ctx *ctx, jpgpu_batch **out) {
    if (!ctx || !out) return JPGPU_ERR_ARGUMENT;
    *out = new (std::nothrow) jpgpu_batch(ctx);
    return *out ? JPGPU_OK : JPGPU_ERR_OUT_OF_MEMORY;
}
void jpgpu_batch_destroy(jpgpu_batch *b) { delete b; }

#define JPGPU_GUARD(b, expr)                                                   \
    if (!(b)) return JPGPU_ERR_ARGUMENT;                                       \
    try {                                                                      \
        return (expr);                                                         \
    } catch (const DecodeError &e) {                                           \
        (b)->ctx->last_error = e.what();                                       \
        return e.status;                                                       \
    } catch (const std::bad_alloc &) {                                         \
        (b)->ctx->last_error = "host allocation failed";                       \
        return JPGPU_ERR_OUT_OF_MEMORY;                                        \
    } catch (const std::exception &e) {                                        \
        (b)->ctx->last_error = e.what();                                       \
        return JPGPU_ERR_DEVICE;                                               \
    }

int jpgpu_batch_upload(jpgpu_batch *b, const uint8_t *const *jpeg, const size_t *len, int n, int format) {
    JPGPU_GUARD(b, b->impl.upload_files(jpeg, len, n, format));
}
int jpgpu_batch_upload_segments(jpgpu_batch *b, const jpgpu_segment *segments, const int *segments_per_file, int n, int format, unsigned flags) {
    JPGPU_GUARD(b, b->impl.upload_segments(segments, segments_per_file, n, format, flags));
}
int jpgpu_batch_upload_frames(jpgpu_batch *b, const jpgpu_frame *frames, const uint16_t *qt, int n, int format) {
    JPGPU_GUARD(b, b->impl.upload_frames(frames, qt, n, format));
}
int jpgpu_batch_decode(jpgpu_batch *b) { JPGPU_GUARD(b, b->impl.decode()); }
static int run_entropy_stages(DeviceBatch &impl) {
    impl.note_entropy_only_request();
    const int rc = impl.run_marker_index();
    return rc != JPGPU_OK ? rc : impl.run_huffman();
}
int jpgpu_batch_run_entropy(jpgpu_batch *b) { JPGPU_GUARD(b, run_entropy_stages(b->impl)); }
int jpgpu_batch_run_idct(jpgpu_batch *b) { JPGPU_GUARD(b, b->impl.run_idct()); }
int jpgpu_batch_sync(jpgpu_batch *b) { JPGPU_GUARD(b, b->impl.sync()); }
int jpgpu_batch_size(const jpgpu_batch *b) { return b ? b->impl.size() : 0; }

int jpgpu_batch_image_info(const jpgpu_batch *b, int i, jpgpu_image_info *info) {
    if (!b || !info) return JPGPU_ERR_ARGUMENT;
    const ImagePlan *img = b->impl.image(i);
    if (!img) return JPGPU_ERR_ARGUMENT;
    memset(info, 0, sizeof *info);
    info->status = img->status;
    info->detail = img->detail;
    info->width = img->width;
    info->height = img->height;
    info->precision = img->precision;
    info->num_components = img->num_components;
    info->sof = img->sof;
    info->restart_interval = img->restart_interval;
    info->mcus_per_line = img->mcus_per_line;
    info->mcus_per_column = img->mcus_per_column;
    info->blocks_per_mcu = img->blocks_per_mcu;
    info->total_blocks = img->total_blocks;
    info->out_offset = img->out_offset;
    info->out_bytes = img->out_bytes;
    info->coef_offset = img->coef_offset;
    memcpy(info->plane, img->plane, sizeof info->plane);
    if (img->status != JPGPU_OK) b->ctx->last_error = img->error;
    return JPGPU_OK;
}

int jpgpu_batch_result(jpgpu_batch *b, int i, jpgpu_image_result *res) { JPGPU_GUARD(b, b->impl.result(i, res)); }
void *jpgpu_batch_output_device(const jpgpu_batch *b, uint64_t *total_bytes) { return b ? b->impl.output_device(total_bytes) : nullptr; }
void *jpgpu_batch_coefficients_device(const jpgpu_batch *b, uint64_t *total_blocks) { return b ? b->impl.coefs_device(total_blocks) : nullptr; }
int jpgpu_batch_download_output(jpgpu_batch *b, int i, void *dst, size_t cap) { JPGPU_GUARD(b, b->impl.download_output(i, dst, cap)); }
int jpgpu_batch_download_coefficients(jpgpu_batch *b, int i, int16_t *dst, size_t cap_blocks) {
    JPGPU_GUARD(b, b->impl.download_coefficients(i, dst, cap_blocks));
}
int jpgpu_batch_upload_coefficients(jpgpu_batch *b, int i, const int16_t *src, size_t nblocks) {
    JPGPU_GUARD(b, b->impl.upload_coefficients(i, src, nblocks));
}
int jpgpu_batch_stage_ms(jpgpu_batch *b, float ms[4]) { JPGPU_GUARD(b, b->impl.stage_ms(ms)); }
int jpgpu_batch_subseq_rounds(const jpgpu_batch *b) { return b ? b->impl.last_subseq_rounds() : 0; }
int jpgpu_batch_set_partial_flush(jpgpu_batch *b, int on) {
    if (!b) return JPGPU_ERR_ARGUMENT;
    b->impl.set_partial_flush(on != 0);
    return JPGPU_OK;
}
int jpgpu_batch_marker_fallbacks(const jpgpu_batch *b) { return b ? b->impl.marker_fallbacks() : 0; }
int jpgpu_batch_progressive_replays(const jpgpu_batch *b) { return b ? b->impl.progressive_replays() : 0; }
int jpgpu_batch_subseq_fallbacks(const jpgpu_batch *b) { return b ? b->impl.subseq_fallbacks() : 0; }
int jpgpu_batch_progressive_fallbacks(const jpgpu_batch *b) { return b ? b->impl.progressive_fallbacks() : 0; }
int jpgpu_batch_ingest_stats(const jpgpu_batch *b, jpgpu_ingest_stats *stats) {
    if (!b || !stats) return JPGPU_ERR_ARGUMENT;
    const IngestStats &s = b->impl.ingest_stats();
    stats->threads = s.threads;
    stats->n_header_only = s.n_header_only;
    stats->n_full_walk = s.n_full_walk;
    stats->parse_ms = s.parse_ms;
    stats->copy_ms = s.copy_ms;
    stats->full_walk_ms = s.full_walk_ms;
    stats->layout_ms = s.layout_ms;
    stats->total_ms = s.total_ms;
    stats->n_pinned_dma = s.n_pinned_dma;
    stats->n_linearised = s.n_linearised;
    return JPGPU_OK;
}
int jpgpu_batch_totals(const jpgpu_batch *b, uint64_t *compressed_bytes, uint64_t *blocks, uint64_t *pixels, uint64_t *output_bytes) {
    if (!b) return JPGPU_ERR_ARGUMENT;
    b->impl.totals(compressed_bytes, blocks, pixels, output_bytes);
    return JPGPU_OK;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------ shared scan runner

namespace {

// Exception the reference throws for a device-reported failure.
[[noreturn]] void throw_for_result(const jpgpu_image_result &res) {
    throw DecodeError(res.status, jpgpu_detail_string(res.detail), res.detail);
}

// WriteBlock + WriteBlockSlow (ref: ScanDecoder/JpegHuffmanBaselineScanDecoder.cs:225-268) over one decoded block.
void write_block_expanded(jpgpu_write_block_fn fn, void *user, const int16_t *block, int component_index, int x, int y, int hs, int vs) {
    if (hs == 1 && vs == 1) {
        fn(user, block, component_index, x, y);
        return;
    }
    int16_t temp[64];
    int hshift = 0, vshift = 0;
    while ((1 << (hshift + 1)) <= hs) hshift++;
    while ((1 << (vshift + 1)) <= vs) vshift++;
    for (int v = 0; v < vs; v++)
        for (int h = 0; h < hs; h++) {
            for (int i = 0; i < 8; i++) {
                const int16_t *row = block + ((8 * v + i) >> vshift) * 8;
                for (int j = 0; j < 8; j++) temp[8 * i + j] = row[(8 * h + j) >> hshift];
            }
            fn(user, temp, component_index, x + 8 * h, y + 8 * v);
        }
}

// Replays the WriteBlock call sequence of ProcessScan (:99-134) from PLANAR_I16 planes, for the first `mcus` MCUs.
// (max_blocks: a failing scan has written the blocks in front of the one it threw in, inside the failing MCU too)
void replay_blocks(const ScanJob &job, const ImagePlan &img, const uint8_t *planes, uint32_t mcus, uint64_t max_blocks, jpgpu_write_block_fn fn, void *user) {
    const BaselineGeometry &g = job.geo;
    uint64_t n_block = 0;
    for (uint32_t m = 0; m < mcus; m++) {
        const int row_mcu = (int)(m / (uint32_t)g.mcus_per_line), col_mcu = (int)(m % (uint32_t)g.mcus_per_line);
        const int offset_x = col_mcu * g.max_h, offset_y = row_mcu * g.max_v;
        for (int c = 0; c < job.scan_components; c++) {
            const ResolvedScanComponent &rc = job.comp[c];
            if (rc.component_index >= 4) continue;
            const jpgpu_plane_info &pl = img.plane[rc.component_index];
            const int16_t *plane = reinterpret_cast<const int16_t *>(planes + pl.offset);
            for (int y = 0; y < rc.v; y++)
                for (int x = 0; x < rc.h; x++) {
                    if (n_block++ >= max_blocks) return;
                    int16_t blk[64];
                    const size_t px = (size_t)(col_mcu * rc.h + x) * 8, py = (size_t)(row_mcu * rc.v + y) * 8;
                    for (int i = 0; i < 8; i++) memcpy(blk + 8 * i, plane + (py + i) * pl.pitch + px, 16);
                    write_block_expanded(fn, user, blk, rc.component_index, (offset_x + x) * 8, (offset_y + y) * 8, rc.hs, rc.vs);
                }
        }
    }
}

// ref: apps/JpegDecode/JpegBufferOutputWriter8Bit.cs:28-60 -- host restatement used when the caller's 8-bit sink
// geometry differs from the frame (then the GPU's INTERLEAVED_U8 layout cannot be used directly).
struct HostSink8 {
    int width, height, component_count;
    uint8_t *out;
    static void write(void *user, const int16_t *block, int component_index, int x, int y) {
        HostSink8 *s = static_cast<HostSink8 *>(user);
        if (x > s->width || y > s->height) return;
        const int ww = std::min(s->width - x, 8), wh = std::min(s->height - y, 8);
        uint8_t *dst = s->out + ((size_t)y * s->width + x) * s->component_count + component_index;
        for (int dy = 0; dy < wh; dy++) {
            uint8_t *row = dst + (size_t)dy * s->width * s->component_count;
            for (int dx = 0; dx < ww; dx++) {
                const int16_t v = block[dx];
                row[(size_t)dx * s->component_count] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
            }
            block += 8;
        }
    }
};

// JpegBlockAllocator.Flush (JpegBlockAllocator.cs:120-190) from PLANAR_I16 planes: component, block row, block column, only the
// component's own block grid, sub-sampled components expanded like WriteBlockSlow.
void flush_planes_to_writer(const BaselineGeometry &g, const ImagePlan &img, const uint8_t *planes, jpgpu_write_block_fn fn, void *user) {
    const FrameHeader &fh = g.frame;
    const int hb = (fh.samples_per_line + 7) / 8, vb = (fh.lines + 7) / 8;
    for (int i = 0; i < fh.num_components && i < 4; i++) {
        const int hs = g.max_h / fh.components[i].h, vs = g.max_v / fh.components[i].v;
        const int hblocks = (hb + hs - 1) / hs, vblocks = (vb + vs - 1) / vs;
        const jpgpu_plane_info &pl = img.plane[i];
        const int16_t *plane = reinterpret_cast<const int16_t *>(planes + pl.offset);
        for (int row = 0; row < vblocks; row++)
            for (int col = 0; col < hblocks; col++) {
                int16_t blk[64];
                for (int r = 0; r < 8; r++) memcpy(blk + 8 * r, plane + ((size_t)row * 8 + r) * pl.pitch + (size_t)col * 8, 16);
                write_block_expanded(fn, user, blk, i, col * hs * 8, row * vs * 8, hs, vs);
            }
    }
}

struct ScanOutcome {
    jpgpu_image_result result;
    size_t reader_advance;  // what ProcessScan advances the outer reader by
};

// Runs one scan job on the GPU and delivers its output.  `direct8` non-null: INTERLEAVED_U8 straight into that
// buffer (geometry must equal the frame's); else PLANAR_I16 + replay through (fn, user).
ScanOutcome run_scan_on_gpu(DeviceBatch &batch, const ScanJob &job, uint8_t *direct8, size_t direct8_bytes, jpgpu_write_block_fn fn,
                            void *user, std::string *err) {
    const int format = direct8 ? JPGPU_FMT_INTERLEAVED_U8 : JPGPU_FMT_PLANAR_I16;
    int rc = batch.upload_single_job(job, format, direct8, direct8_bytes);
    if (rc == JPGPU_OK) rc = batch.decode();
    if (rc == JPGPU_OK) rc = batch.sync();
    ScanOutcome oc;
    memset(&oc, 0, sizeof oc);
    if (rc == JPGPU_OK) rc = batch.result(0, &oc.result);
    if (rc != JPGPU_OK) {
        if (err) *err = "device failure";
        throw DecodeError(rc, "GPU scan decode failed (see jpgpu_last_error)");
    }
    const ImagePlan &img = *batch.image(0);
    const uint32_t total = (uint32_t)(job.geo.mcus_per_line * job.geo.mcus_per_column);
    const uint32_t dri_eff = job.geo.restart_interval ? job.geo.restart_interval : total;
    uint32_t good_mcus = std::min(oc.result.decoded_mcus, total);
    uint64_t good_blocks = ~0ull;
    if (oc.result.status != JPGPU_OK) {
        good_mcus = std::min<uint64_t>((uint64_t)oc.result.error_interval * dri_eff, total);
        if (oc.result.error_block != 0xFFFFFFFFu && job.blocks_per_mcu > 0) {  // block by block up to the throw, inside the failing interval too
            good_blocks = oc.result.error_block;
            good_mcus = (uint32_t)std::min<uint64_t>((good_blocks + (uint64_t)job.blocks_per_mcu - 1) / (uint64_t)job.blocks_per_mcu, total);
        }
    }
    if (direct8) {
        if (batch.download_output(0, direct8, direct8_bytes) != JPGPU_OK) throw DecodeError(JPGPU_ERR_DEVICE, "output download failed");
    } else if (fn) {
        std::vector<uint8_t> planes(img.out_bytes);
        if (batch.download_output(0, planes.data(), planes.size()) != JPGPU_OK) throw DecodeError(JPGPU_ERR_DEVICE, "output download failed");
        replay_blocks(job, img, planes.data(), good_mcus, good_blocks, fn, user);
    }
    // where ProcessScan leaves the outer reader (:145-149, :167-176)
    const uint32_t term = oc.result.terminator;
    size_t adv = oc.result.bytes_consumed;
    if (term >= 0xD0 && term <= 0xD7) adv += 2;  // a trailing RSTn is consumed, any other marker is left unread
    oc.reader_advance = std::min(adv, job.entropy_len);
    return oc;
}

}  // namespace

// ------------------------------------------------------------------------------------------------ (2) per scan

extern "C" int jpgpu_decode_scan(jpgpu_ctx *ctx, const jpgpu_frame *frame, const jpgpu_scan *scan, const uint16_t qt[4][64],
                                 const uint8_t qt_present[4], const jpgpu_dht dht[2][4], uint16_t restart_interval,
                                 const uint8_t *entropy, size_t len, int format, void *out, size_t cap, jpgpu_image_result *result,
                                 size_t *bytes_consumed) {
    if (!ctx) return JPGPU_ERR_ARGUMENT;
    if (!frame || !scan || !qt || !qt_present || !dht || !entropy || !out) {
        ctx->last_error = "jpgpu_decode_scan: null argument";
        return JPGPU_ERR_ARGUMENT;
    }
    try {
        if (frame->sof != kSOF0 && frame->sof != kSOF1)
            throw DecodeError(JPGPU_ERR_NOT_SUPPORTED, "This image type is not supported.", kDetailUnsupportedFrame);
        if (frame->num_components > 4 || scan->num_components > 4)
            throw DecodeError(JPGPU_ERR_NOT_SUPPORTED, "More than 4 components are not supported.", kDetailUnsupportedFrame);
        HostDecoder dec;
        FrameHeader fh;
        fh.precision = frame->precision;
        fh.lines = frame->height;
        fh.samples_per_line = frame->width;
        fh.num_components = frame->num_components;
        for (int i = 0; i < frame->num_components; i++) fh.components.push_back({frame->comp[i].identifier, frame->comp[i].h, frame->comp[i].v, frame->comp[i].tq});
        dec.set_frame_header(fh);
        dec.set_start_of_frame(frame->sof);
        dec.set_restart_interval(restart_interval);
        for (int id = 0; id < 4; id++) {
            if (qt_present[id]) {
                QuantTable q;
                q.identifier = (uint8_t)id;
                memcpy(q.elements, qt[id], sizeof q.elements);
                dec.set_quantization_table(q);
            }
            for (int cls = 0; cls < 2; cls++) {
                const jpgpu_dht &d = dht[cls][id];
                if (!d.present) continue;
                HuffTable t;
                if (!HuffTable::from_bits_values((uint8_t)cls, (uint8_t)id, d.bits, d.values, d.num_values, &t))
                    throw_invalid_data("Failed to parse Huffman table.", kDetailBadHeader);
                dec.set_huffman_table(t);
            }
        }
        ScanHeader sh;
        sh.num_components = scan->num_components;
        sh.ss = scan->ss;
        sh.se = scan->se;
        sh.ah = scan->ah;
        sh.al = scan->al;
        for (int i = 0; i < scan->num_components; i++) sh.components.push_back({scan->comp[i].selector, scan->comp[i].td, scan->comp[i].ta});
        const BaselineGeometry geo = BaselineGeometry::latch(dec, fh);
        const ScanJob job = make_scan_job(dec, geo, sh, entropy, len);
        DeviceBatch batch(ctx);
        int rc = batch.upload_single_job(job, format, nullptr, 0);
        if (rc != JPGPU_OK) return rc;
        if ((rc = batch.decode()) != JPGPU_OK) return rc;
        if ((rc = batch.sync()) != JPGPU_OK) return rc;
        jpgpu_image_result res;
        if ((rc = batch.result(0, &res)) != JPGPU_OK) return rc;
        if (result) *result = res;
        if (bytes_consumed) {
            size_t adv = res.bytes_consumed;
            if (res.terminator >= 0xD0 && res.terminator <= 0xD7) adv += 2;
            *bytes_consumed = std::min(adv, len);
        }
        if ((rc = batch.download_output(0, out, cap)) != JPGPU_OK) return rc;
        if (res.status != JPGPU_OK) ctx->last_error = jpgpu_detail_string(res.detail);
        return res.status;
    } catch (const DecodeError &e) {
        ctx->last_error = e.what();
        if (result) {
            memset(result, 0, sizeof *result);
            result->status = e.status;
            result->detail = e.detail;
        }
        return e.status;
    } catch (const std::exception &e) {
        ctx->last_error = e.what();
        return JPGPU_ERR_DEVICE;
    }
}

// ------------------------------------------------------------------------------------------------ (3) decoder mirror

struct jpgpu_decoder {
    jpgpu_ctx *ctx = nullptr;
    HostDecoder host;
    std::string last_error;
    enum WriterKind { kNone, kCallback, kBuffer8 } writer = kNone;
    jpgpu_write_block_fn fn = nullptr;
    void *user = nullptr;
    HostSink8 sink8 = {0, 0, 0, nullptr};
    size_t sink8_cap = 0;
    std::unique_ptr<DeviceBatch> batch;
};

namespace {

// What JpegScanDecoder.Create / ProcessScan / Dispose do in the mirror: run every baseline scan on the GPU.
class GpuScanHandler final : public ScanHandler {
  public:
    explicit GpuScanHandler(jpgpu_decoder *d) : d_(d) {}
    void on_frame(HostDecoder &dec, int sof) override {
        dispose_progressive(dec);  // a new SOF replaces the scan decoder; the old one is disposed first
        sof_ = sof;
        baseline_ = (sof == kSOF0 || sof == kSOF1);
        if (baseline_) geo_ = BaselineGeometry::latch(dec, dec.frame_header());
        if (sof == kSOF2) prog_.begin(dec, dec.frame_header());
    }
    void on_scan(HostDecoder &dec, MarkerReader &reader, const ScanHeader &scan) override {
        if (!baseline_ && !prog_.active())
            throw DecodeError(JPGPU_ERR_NOT_SUPPORTED, "Only Huffman DCT frames (SOF0, SOF1, SOF2) run on the GPU path.", kDetailUnsupportedFrame);
        if (d_->writer == jpgpu_decoder::kNone) throw_invalid_operation("Output writer is not specified.");
        if (!d_->ctx) throw DecodeError(JPGPU_ERR_NO_DEVICE, "No HIP device: scans are decoded on the GPU only (no CPU fallback).");
        const uint8_t *entropy = reader.remaining_bytes();
        const size_t len = (size_t)reader.remaining_byte_count();
        if (prog_.active()) {
            // ref: JpegHuffmanProgressiveScanDecoder.ProcessScan: coefficients accumulate, nothing is written yet and
            // the outer reader is not advanced; the GPU work happens in Dispose()
            prog_.add_scan(dec, scan, entropy, len);
            return;
        }
        const ScanJob job = make_scan_job(dec, geo_, scan, entropy, len);
        if (!d_->batch) d_->batch.reset(new DeviceBatch(d_->ctx));
        const FrameHeader &fh = geo_.frame;
        ScanOutcome oc;
        const bool direct = d_->writer == jpgpu_decoder::kBuffer8 && d_->sink8.width == fh.samples_per_line && d_->sink8.height == fh.lines &&
                            d_->sink8.component_count == fh.num_components;
        if (direct) {
            oc = run_scan_on_gpu(*d_->batch, job, d_->sink8.out, (size_t)fh.samples_per_line * fh.lines * fh.num_components, nullptr, nullptr, &d_->last_error);
        } else if (d_->writer == jpgpu_decoder::kBuffer8) {
            oc = run_scan_on_gpu(*d_->batch, job, nullptr, 0, HostSink8::write, &d_->sink8, &d_->last_error);
        } else {
            oc = run_scan_on_gpu(*d_->batch, job, nullptr, 0, d_->fn, d_->user, &d_->last_error);
        }
        if (oc.result.status != JPGPU_OK) throw_for_result(oc.result);
        reader.try_advance((int)oc.reader_advance);
    }
    void on_dispose(HostDecoder &dec) override {
        // (Decode() failing in its marker walk: the reference's `finally` still disposes the scan decoder, which transforms and
        // flushes what the scans in front of the failure left in the store -- JpegDecoder.cs:545-549; so does this.  The
        // failure that ended the walk is the one the caller sees: HostDecoder::decode swallows what this throws then.)
        dispose_progressive(dec);
    }

  private:
    // JpegHuffmanProgressiveScanDecoder.Dispose (:421-470): every scan of the frame on the GPU, in file order, then
    // the dequantise + IDCT pass, then JpegBlockAllocator.Flush (JpegBlockAllocator.cs:120-149) into the writer.
    void dispose_progressive(HostDecoder &dec) {
        deferred_scan_failed_ = false;
        if (!prog_.active()) return;
        ProgressiveFrame frame = std::move(prog_);
        prog_.reset();
        if (frame.scans().empty()) return;
        if (!d_->batch) d_->batch.reset(new DeviceBatch(d_->ctx));
        DeviceBatch &batch = *d_->batch;
        const FrameHeader &fh = frame.geo().frame;
        const bool direct = d_->writer == jpgpu_decoder::kBuffer8 && d_->sink8.width == fh.samples_per_line && d_->sink8.height == fh.lines &&
                            d_->sink8.component_count == fh.num_components;
        int rc = batch.upload_progressive_frame(frame, dec.input(), dec.input_len(), sof_, direct ? JPGPU_FMT_INTERLEAVED_U8 : JPGPU_FMT_PLANAR_I16);
        if (rc == JPGPU_OK && direct) {
            // the sink's buffer keeps whatever the caller had outside the decoded area
        }
        if (rc == JPGPU_OK) rc = batch.decode();
        if (rc == JPGPU_OK) rc = batch.sync();
        jpgpu_image_result res;
        memset(&res, 0, sizeof res);
        if (rc == JPGPU_OK) rc = batch.result(0, &res);
        if (rc != JPGPU_OK) throw DecodeError(rc, "GPU progressive decode failed (see jpgpu_last_error)");
        // A scan that failed: the writer still receives the flush of the store as that scan left it (the batch has issued the failed
        // frame again in file order up to the throw: DeviceBatch::replay_failed_progressive), THEN the exception leaves.
        const bool unsupported = res.status == JPGPU_ERR_NOT_SUPPORTED;
        const ImagePlan &img = *batch.image(0);
        if (!unsupported && img.status == JPGPU_OK) {
            if (direct) {
                if (batch.download_output(0, d_->sink8.out, (size_t)fh.samples_per_line * fh.lines * fh.num_components) != JPGPU_OK)
                    throw DecodeError(JPGPU_ERR_DEVICE, "output download failed");
            } else {
                std::vector<uint8_t> planes(img.out_bytes);
                if (batch.download_output(0, planes.data(), planes.size()) != JPGPU_OK) throw DecodeError(JPGPU_ERR_DEVICE, "output download failed");
                jpgpu_write_block_fn fn = d_->writer == jpgpu_decoder::kBuffer8 ? HostSink8::write : d_->fn;
                void *user = d_->writer == jpgpu_decoder::kBuffer8 ? (void *)&d_->sink8 : d_->user;
                flush_planes_to_writer(frame.geo(), img, planes.data(), fn, user);
            }
        }
        if (res.status != JPGPU_OK) {
            deferred_scan_failed_ = !unsupported;
            throw_for_result(res);
        }
    }
    bool dispose_failure_is_a_deferred_scan() const override { return deferred_scan_failed_; }

    bool deferred_scan_failed_ = false;
    jpgpu_decoder *d_;
    BaselineGeometry geo_;
    ProgressiveFrame prog_;
    bool baseline_ = false;
    int sof_ = 0;
};

template <typename F>
int guarded(jpgpu_decoder *d, F &&f) {
    if (!d) return JPGPU_ERR_ARGUMENT;
    try {
        d->last_error.clear();
        return f();
    } catch (const DecodeError &e) {
        d->last_error = e.what();
        return e.status;
    } catch (const std::bad_alloc &) {
        d->last_error = "host allocation failed";
        return JPGPU_ERR_OUT_OF_MEMORY;
    } catch (const std::exception &e) {
        d->last_error = e.what();
        return JPGPU_ERR_DEVICE;
    }
}

}  // namespace

extern "C" {

int jpgpu_decoder_create(jpgpu_ctx *ctx, jpgpu_decoder **out) {
    // ctx may be NULL: the host-only calls (SetInput, Identify, metadata, tables) work without a GPU;
    // Decode then fails with JPGPU_ERR_NO_DEVICE -- there is no CPU decode path.
    if (!out) return JPGPU_ERR_ARGUMENT;
    *out = new (std::nothrow) jpgpu_decoder();
    if (!*out) return JPGPU_ERR_OUT_OF_MEMORY;
    (*out)->ctx = ctx;
    return JPGPU_OK;
}
void jpgpu_decoder_destroy(jpgpu_decoder *d) { delete d; }
const char *jpgpu_decoder_last_error(const jpgpu_decoder *d) { return d ? d->last_error.c_str() : ""; }

int jpgpu_decoder_set_input(jpgpu_decoder *d, const uint8_t *data, size_t len) {
    return guarded(d, [&] {
        d->host.set_input(data, len);
        return JPGPU_OK;
    });
}
int jpgpu_decoder_identify(jpgpu_decoder *d, int load_quantization_tables, int *stream_length) {
    return guarded(d, [&] {
        const int n = d->host.identify(load_quantization_tables != 0);
        if (stream_length) *stream_length = n;
        return JPGPU_OK;
    });
}
int jpgpu_decoder_try_estimate_quality(jpgpu_decoder *d, float *quality) {
    if (!d || !quality) return 0;
    return d->host.try_estimate_quality(quality) ? 1 : 0;
}
int jpgpu_decoder_width(const jpgpu_decoder *d) { return d && d->host.has_frame_header() ? d->host.frame_header().samples_per_line : -1; }
int jpgpu_decoder_height(const jpgpu_decoder *d) { return d && d->host.has_frame_header() ? d->host.frame_header().lines : -1; }
int jpgpu_decoder_precision(const jpgpu_decoder *d) { return d && d->host.has_frame_header() ? d->host.frame_header().precision : -1; }
int jpgpu_decoder_number_of_components(const jpgpu_decoder *d) { return d && d->host.has_frame_header() ? d->host.frame_header().num_components : -1; }
int jpgpu_decoder_start_of_frame(const jpgpu_decoder *d) { return d ? d->host.start_of_frame() : 0; }
int jpgpu_decoder_get_maximum_horizontal_sampling(jpgpu_decoder *d) {
    int v = -1;
    guarded(d, [&] {
        v = d->host.maximum_horizontal_sampling();
        return JPGPU_OK;
    });
    return v;
}
int jpgpu_decoder_get_maximum_vertical_sampling(jpgpu_decoder *d) {
    int v = -1;
    guarded(d, [&] {
        v = d->host.maximum_vertical_sampling();
        return JPGPU_OK;
    });
    return v;
}
int jpgpu_decoder_get_horizontal_sampling(jpgpu_decoder *d, int component_index) {
    int v = -1;
    guarded(d, [&] {
        const FrameHeader &fh = d->host.frame_header();
        if ((unsigned)component_index >= fh.components.size()) throw DecodeError(JPGPU_ERR_ARGUMENT, "Specified argument was out of the range of valid values. (Parameter 'componentIndex')");
        v = fh.components[(size_t)component_index].h;
        return JPGPU_OK;
    });
    return v;
}
int jpgpu_decoder_get_vertical_sampling(jpgpu_decoder *d, int component_index) {
    int v = -1;
    guarded(d, [&] {
        const FrameHeader &fh = d->host.frame_header();
        if ((unsigned)component_index >= fh.components.size()) throw DecodeError(JPGPU_ERR_ARGUMENT, "Specified argument was out of the range of valid values. (Parameter 'componentIndex')");
        v = fh.components[(size_t)component_index].v;
        return JPGPU_OK;
    });
    return v;
}
int jpgpu_decoder_get_restart_interval(const jpgpu_decoder *d) { return d ? d->host.restart_interval() : 0; }
int jpgpu_decoder_set_restart_interval(jpgpu_decoder *d, int restart_interval) {
    return guarded(d, [&] {
        d->host.set_restart_interval(restart_interval);
        return JPGPU_OK;
    });
}
int jpgpu_decoder_load_tables(jpgpu_decoder *d, const uint8_t *data, size_t len) {
    return guarded(d, [&] {
        d->host.load_tables(data, len);
        return JPGPU_OK;
    });
}
int jpgpu_decoder_set_output_writer(jpgpu_decoder *d, jpgpu_write_block_fn fn, void *user) {
    return guarded(d, [&] {
        if (!fn) throw DecodeError(JPGPU_ERR_ARGUMENT, "Value cannot be null. (Parameter 'outputWriter')");
        d->writer = jpgpu_decoder::kCallback;
        d->fn = fn;
        d->user = user;
        return JPGPU_OK;
    });
}
int jpgpu_decoder_set_output_buffer8(jpgpu_decoder *d, int width, int height, int component_count, uint8_t *out, size_t cap) {
    return guarded(d, [&] {
        if (!out) throw DecodeError(JPGPU_ERR_ARGUMENT, "Value cannot be null. (Parameter 'output')");
        if (width < 0 || height < 0 || component_count <= 0 || cap < (size_t)width * height * component_count)
            throw DecodeError(JPGPU_ERR_ARGUMENT, "Destination buffer is too small.");  // ref: JpegBufferOutputWriter8Bit.cs:17-20
        d->writer = jpgpu_decoder::kBuffer8;
        d->sink8 = {width, height, component_count, out};
        d->sink8_cap = cap;
        return JPGPU_OK;
    });
}
int jpgpu_decoder_decode(jpgpu_decoder *d) {
    return guarded(d, [&] {
        GpuScanHandler handler(d);
        d->host.decode(handler, d->writer != jpgpu_decoder::kNone);
        return JPGPU_OK;
    });
}
void jpgpu_decoder_reset_input(jpgpu_decoder *d) {
    if (d) d->host.reset_input();
}
void jpgpu_decoder_reset_header(jpgpu_decoder *d) {
    if (d) d->host.reset_header();
}
void jpgpu_decoder_reset_tables(jpgpu_decoder *d) {
    if (d) d->host.reset_tables();
}
void jpgpu_decoder_reset_output_writer(jpgpu_decoder *d) {
    if (d) {
        d->writer = jpgpu_decoder::kNone;
        d->fn = nullptr;
        d->user = nullptr;
    }
}
void jpgpu_decoder_reset(jpgpu_decoder *d) {
    jpgpu_decoder_reset_input(d);
    jpgpu_decoder_reset_header(d);
    jpgpu_decoder_reset_tables(d);
    jpgpu_decoder_reset_output_writer(d);
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------ (2b) progressive, per scan

struct jpgpu_progressive {
    jpgpu_ctx *ctx;
    HostDecoder dec;         // the frame header, and the tables / restart interval in force at the scan being processed
    ProgressiveFrame frame;  // the scan decoder's host state: component slots, block grids (ref: ...ProgressiveScanDecoder.cs:23-55)
    DeviceBatch batch;       // owns the coefficient store in HBM between the calls
    int n_scans = 0;
    explicit jpgpu_progressive(jpgpu_ctx *c) : ctx(c), batch(c) {}
};

namespace {
void load_tables_from_arguments(HostDecoder &dec, const uint16_t qt[4][64], const uint8_t qt_present[4], const jpgpu_dht dht[2][4]) {
    dec.reset_tables();
    for (int id = 0; id < 4; id++) {
        if (qt && qt_present && qt_present[id]) {
            QuantTable q;
            q.identifier = (uint8_t)id;
            memcpy(q.elements, qt[id], sizeof q.elements);
            dec.set_quantization_table(q);
        }
        for (int cls = 0; cls < 2 && dht; cls++) {
            const jpgpu_dht &d = dht[cls][id];
            if (!d.present) continue;
            HuffTable t;
            if (!HuffTable::from_bits_values((uint8_t)cls, (uint8_t)id, d.bits, d.values, d.num_values, &t))
                throw_invalid_data("Failed to parse Huffman table.", kDetailBadHeader);
            dec.set_huffman_table(t);
        }
    }
}
FrameHeader frame_from_argument(const jpgpu_frame *frame) {
    FrameHeader fh;
    fh.precision = frame->precision;
    fh.lines = frame->height;
    fh.samples_per_line = frame->width;
    fh.num_components = frame->num_components;
    for (int i = 0; i < frame->num_components && i < 4; i++) fh.components.push_back({frame->comp[i].identifier, frame->comp[i].h, frame->comp[i].v, frame->comp[i].tq});
    return fh;
}
ScanHeader scan_from_argument(const jpgpu_scan *scan) {
    ScanHeader sh;
    sh.num_components = scan->num_components;
    sh.ss = scan->ss;
    sh.se = scan->se;
    sh.ah = scan->ah;
    sh.al = scan->al;
    for (int i = 0; i < scan->num_components && i < 4; i++) sh.components.push_back({scan->comp[i].selector, scan->comp[i].td, scan->comp[i].ta});
    return sh;
}
template <typename F>
int guarded_ctx(jpgpu_ctx *ctx, jpgpu_image_result *result, F &&f) {
    try {
        return f();
    } catch (const DecodeError &e) {
        ctx->last_error = e.what();
        if (result) {
            memset(result, 0, sizeof *result);
            result->status = e.status;
            result->detail = e.detail;
        }
        return e.status;
    } catch (const std::bad_alloc &) {
        ctx->last_error = "host allocation failed";
        return JPGPU_ERR_OUT_OF_MEMORY;
    } catch (const std::exception &e) {
        ctx->last_error = e.what();
        return JPGPU_ERR_DEVICE;
    }
}
}  // namespace

extern "C" {

int jpgpu_progressive_begin(jpgpu_ctx *ctx, const jpgpu_frame *frame, jpgpu_progressive **out) {
    if (!ctx || !out) return JPGPU_ERR_ARGUMENT;
    *out = nullptr;
    if (!frame) {
        ctx->last_error = "jpgpu_progressive_begin: null frame header";
        return JPGPU_ERR_ARGUMENT;
    }
    return guarded_ctx(ctx, nullptr, [&] {
        if (frame->sof != kSOF2) throw DecodeError(JPGPU_ERR_NOT_SUPPORTED, "This image type is not supported.", kDetailUnsupportedFrame);
        if (frame->num_components > 4) throw DecodeError(JPGPU_ERR_NOT_SUPPORTED, "More than 4 components are not supported.", kDetailUnsupportedFrame);
        std::unique_ptr<jpgpu_progressive> p(new jpgpu_progressive(ctx));
        const FrameHeader fh = frame_from_argument(frame);
        p->dec.set_frame_header(fh);
        p->dec.set_start_of_frame(kSOF2);
        p->frame.begin(p->dec, fh);  // the scan decoder's constructor + JpegBlockAllocator.Allocate
        *out = p.release();
        return (int)JPGPU_OK;
    });
}

int jpgpu_progressive_scan(jpgpu_progressive *p, const jpgpu_scan *scan, const uint16_t qt[4][64], const uint8_t qt_present[4],
                           const jpgpu_dht dht[2][4], uint16_t restart_interval, const uint8_t *entropy, size_t len,
                           jpgpu_image_result *result, size_t *bytes_consumed) {
    if (!p) return JPGPU_ERR_ARGUMENT;
    if (!scan || !qt || !qt_present || !dht || (!entropy && len)) {
        p->ctx->last_error = "jpgpu_progressive_scan: null argument";
        return JPGPU_ERR_ARGUMENT;
    }
    if (bytes_consumed) *bytes_consumed = 0;  // ProcessScan leaves the outer reader where it is
    return guarded_ctx(p->ctx, result, [&] {
        if (scan->num_components > 4) throw DecodeError(JPGPU_ERR_NOT_SUPPORTED, "More than 4 components are not supported.", kDetailUnsupportedFrame);
        load_tables_from_arguments(p->dec, qt, qt_present, dht);
        p->dec.set_restart_interval(restart_interval);
        const size_t before = p->frame.scans().size();
        p->frame.add_scan(p->dec, scan_from_argument(scan), entropy, len);  // ProcessScan's checks (:57-69) and the component slots
        int rc = p->batch.upload_progressive_scan(p->frame, (int)before, p->n_scans == 0);
        // the caller's entropy bytes are in HBM now; the recorded job must not keep pointing at them
        p->frame.scans()[before].entropy = nullptr;
        p->frame.scans()[before].entropy_len = 0;
        if (rc != JPGPU_OK) return rc;
        const bool first_scan = p->n_scans == 0;
        p->n_scans++;
        if (!first_scan && (rc = p->batch.snapshot_progressive_store()) != JPGPU_OK) return rc;  // (a device copy: what a failing scan is re-issued from)
        if ((rc = p->batch.run_marker_index()) != JPGPU_OK) return rc;
        if ((rc = p->batch.run_huffman()) != JPGPU_OK) return rc;
        if ((rc = p->batch.sync()) != JPGPU_OK) return rc;
        jpgpu_image_result res;
        if ((rc = p->batch.result(0, &res)) != JPGPU_OK) return rc;
        // a failing scan leaves the store where the reference's ProcessScan left it: every coefficient in front of the throw, none
        // behind it (what jpgpu_progressive_dispose then flushes is what Decode()'s `finally` flushes, JpegDecoder.cs:545-549)
        if (res.status != JPGPU_OK && res.detail != kDetailUnsupportedFrame && (rc = p->batch.rerun_failed_progressive_scan(first_scan)) != JPGPU_OK) return rc;
        if (result) *result = res;
        if (res.status != JPGPU_OK) p->ctx->last_error = jpgpu_detail_string(res.detail);
        return (int)res.status;
    });
}

namespace {
int progressive_idct_pass(jpgpu_progressive *p, int format) {
    int rc = p->batch.upload_progressive_dispose(p->frame, format);
    // Dispose() without a ProcessScan before it: the allocator's zeroed blocks go to the writer untransformed (the decoder's
    // component slots still have sampling factors 0: JpegHuffmanProgressiveScanDecoder.cs:52-56, 425-447)
    if (rc == JPGPU_OK && p->n_scans == 0) rc = p->batch.clear_progressive_stores();
    if (rc == JPGPU_OK) rc = p->batch.run_idct();
    if (rc == JPGPU_OK) rc = p->batch.sync();
    if (rc != JPGPU_OK) return rc;
    jpgpu_image_result res;
    if ((rc = p->batch.result(0, &res)) != JPGPU_OK) return rc;
    if (res.status != JPGPU_OK) return res.status;  // a scan order whose Dispose() pass the reference mangles
    return JPGPU_OK;
}
}  // namespace

int jpgpu_progressive_dispose(jpgpu_progressive *p, int format, void *out, size_t cap) {
    if (!p) return JPGPU_ERR_ARGUMENT;
    if (!out) {
        p->ctx->last_error = "jpgpu_progressive_dispose: null output";
        return JPGPU_ERR_ARGUMENT;
    }
    return guarded_ctx(p->ctx, nullptr, [&] {
        int rc = progressive_idct_pass(p, format);
        if (rc != JPGPU_OK) return rc;
        return p->batch.download_output(0, out, cap);
    });
}

int jpgpu_progressive_dispose_to_writer(jpgpu_progressive *p, jpgpu_write_block_fn fn, void *user) {
    if (!p) return JPGPU_ERR_ARGUMENT;
    if (!fn) {
        p->ctx->last_error = "Value cannot be null. (Parameter 'outputWriter')";
        return JPGPU_ERR_ARGUMENT;
    }
    return guarded_ctx(p->ctx, nullptr, [&] {
        int rc = progressive_idct_pass(p, JPGPU_FMT_PLANAR_I16);
        if (rc != JPGPU_OK) return rc;
        const ImagePlan &img = *p->batch.image(0);
        std::vector<uint8_t> planes(img.out_bytes);
        if ((rc = p->batch.download_output(0, planes.data(), planes.size())) != JPGPU_OK) return rc;
        flush_planes_to_writer(p->frame.geo(), img, planes.data(), fn, user);
        return (int)JPGPU_OK;
    });
}

int jpgpu_progressive_output_size(jpgpu_progressive *p, int format, size_t *bytes) {
    if (!p || !bytes) return JPGPU_ERR_ARGUMENT;
    return guarded_ctx(p->ctx, nullptr, [&] {
        const FrameHeader &fh = p->frame.geo().frame;
        const BaselineGeometry &g = p->frame.geo();
        size_t n = 0;
        if (format == JPGPU_FMT_INTERLEAVED_U8) n = (size_t)fh.samples_per_line * fh.lines * fh.num_components;
        else if (format == JPGPU_FMT_RGB_U8) n = (size_t)fh.samples_per_line * fh.lines * 3;
        else if (format == JPGPU_FMT_RGBA_U8) n = (size_t)fh.samples_per_line * fh.lines * 4;
        else if (format == JPGPU_FMT_EXTENDED_U16) n = (size_t)fh.samples_per_line * fh.lines * 8;
        else if (format == JPGPU_FMT_PLANAR_U8 || format == JPGPU_FMT_PLANAR_I16) {
            const size_t sb = format == JPGPU_FMT_PLANAR_I16 ? 2 : 1;
            for (int c = 0; c < fh.num_components && c < 4; c++) {
                const size_t plane = (size_t)g.mcus_per_line * fh.components[c].h * 8 * (size_t)g.mcus_per_column * fh.components[c].v * 8 * sb;
                n = (n + plane + 255) / 256 * 256;
            }
        } else {
            throw DecodeError(JPGPU_ERR_ARGUMENT, "unknown format");
        }
        *bytes = n;
        return (int)JPGPU_OK;
    });
}

void jpgpu_progressive_destroy(jpgpu_progressive *p) { delete p; }

// ------------------------------------------------------------------------------------------------ (3b) TIFF-style decoder surface

int jpgpu_decoder_set_start_of_frame(jpgpu_decoder *d, int marker) {
    return guarded(d, [&] {
        d->host.set_start_of_frame(marker & 0xFF);
        return JPGPU_OK;
    });
}
int jpgpu_decoder_set_frame_header(jpgpu_decoder *d, const jpgpu_frame *frame) {
    return guarded(d, [&] {
        if (!frame) throw DecodeError(JPGPU_ERR_ARGUMENT, "Value cannot be null. (Parameter 'frameHeader')");
        if (frame->num_components > 4) throw DecodeError(JPGPU_ERR_NOT_SUPPORTED, "More than 4 components are not supported.", kDetailUnsupportedFrame);
        d->host.set_frame_header(frame_from_argument(frame));
        return JPGPU_OK;
    });
}
int jpgpu_decoder_set_huffman_table(jpgpu_decoder *d, int table_class, int identifier, const uint8_t bits[16], const uint8_t *values, int num_values) {
    return guarded(d, [&] {
        if (!bits || (!values && num_values)) throw DecodeError(JPGPU_ERR_ARGUMENT, "Value cannot be null. (Parameter 'table')");
        HuffTable t;
        if (table_class < 0 || table_class > 1 || identifier < 0 || identifier > 3 || num_values < 0 || num_values > 256 ||
            !HuffTable::from_bits_values((uint8_t)table_class, (uint8_t)identifier, bits, values, num_values, &t))
            throw_invalid_data("Failed to parse Huffman table.", kDetailBadHeader);
        d->host.set_huffman_table(t);
        return JPGPU_OK;
    });
}
int jpgpu_decoder_set_quantization_table(jpgpu_decoder *d, int element_precision, int identifier, const uint16_t *zigzag64) {
    return guarded(d, [&] {
        if (!zigzag64) throw DecodeError(JPGPU_ERR_ARGUMENT, "No actual quantization table is provided. (Parameter 'table')");  // :842-845
        if (identifier < 0 || identifier > 3 || element_precision < 0 || element_precision > 1)
            throw DecodeError(JPGPU_ERR_ARGUMENT, "Specified argument was out of the range of valid values. (Parameter 'identifier')");
        QuantTable q;
        q.precision = (uint8_t)element_precision;
        q.identifier = (uint8_t)identifier;
        memcpy(q.elements, zigzag64, sizeof q.elements);
        d->host.set_quantization_table(q);
        return JPGPU_OK;
    });
}
int jpgpu_decoder_clear_huffman_table(jpgpu_decoder *d) {
    return guarded(d, [&] {
        d->host.clear_huffman_tables();
        return JPGPU_OK;
    });
}
int jpgpu_decoder_clear_quantization_table(jpgpu_decoder *d) {
    return guarded(d, [&] {
        d->host.clear_quantization_tables();
        return JPGPU_OK;
    });
}
// JpegDecoder.ProcessScan(ref JpegReader, JpegScanHeader) (:624-632): a scan decoder for StartOfFrame over the frame header
// set before, ONE scan, Dispose (for a progressive frame that is the IDCT pass + Flush over this one scan's coefficients).
int jpgpu_decoder_process_scan(jpgpu_decoder *d, const jpgpu_scan *scan, const uint8_t *entropy, size_t len, size_t *bytes_consumed) {
    if (bytes_consumed) *bytes_consumed = 0;
    return guarded(d, [&] {
        if (!scan || (!entropy && len)) throw DecodeError(JPGPU_ERR_ARGUMENT, "Value cannot be null. (Parameter 'scanHeader')");
        const int sof = d->host.start_of_frame();
        (void)d->host.frame_header();  // "Call Identify() before this operation." when SetFrameHeader was not called (:378)
        if (sof != kSOF0 && sof != kSOF1 && sof != kSOF2)
            throw DecodeError(JPGPU_ERR_NOT_SUPPORTED, "This image type is not supported.", kDetailUnsupportedFrame);  // :629 (scan decoders outside this path)
        if (scan->num_components > 4) throw DecodeError(JPGPU_ERR_NOT_SUPPORTED, "More than 4 components are not supported.", kDetailUnsupportedFrame);
        GpuScanHandler handler(d);
        handler.on_frame(d->host, sof);  // JpegScanDecoder.Create: the restart interval is latched here
        MarkerReader reader(entropy, len);
        handler.on_scan(d->host, reader, scan_from_argument(scan));
        handler.on_dispose(d->host);
        if (bytes_consumed) *bytes_consumed = (size_t)reader.consumed_byte_count();
        return JPGPU_OK;
    });
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------ (4) encoder

struct jpgpu_encoder {
    jpgpu_ctx *ctx;
    EncodeBatch impl;
    explicit jpgpu_encoder(jpgpu_ctx *c) : ctx(c), impl(c) {}
};

extern "C" {

int jpgpu_encoder_create(jpgpu_ctx *ctx, jpgpu_encoder **out) {
    if (!ctx || !out) return JPGPU_ERR_ARGUMENT;
    *out = new (std::nothrow) jpgpu_encoder(ctx);
    return *out ? JPGPU_OK : JPGPU_ERR_OUT_OF_MEMORY;
}
void jpgpu_encoder_destroy(jpgpu_encoder *enc) { delete enc; }
int jpgpu_encoder_upload(jpgpu_encoder *enc, const uint8_t *const *pixels, const jpgpu_encode_params *params, int n) {
    JPGPU_GUARD(enc, enc->impl.upload(pixels, params, n));
}
int jpgpu_encoder_set_quantization_table(jpgpu_encoder *enc, int i, int identifier, const uint16_t *zigzag64) {
    JPGPU_GUARD(enc, enc->impl.set_quantization_table(i, identifier, zigzag64));
}
int jpgpu_encoder_encode(jpgpu_encoder *enc) { JPGPU_GUARD(enc, enc->impl.encode()); }
int jpgpu_encoder_stage_ms(jpgpu_encoder *enc, float ms[5]) { JPGPU_GUARD(enc, enc->impl.stage_ms(ms)); }
int jpgpu_encoder_emit_passes(const jpgpu_encoder *enc, int *one_pass, int *fell_back) {
    if (!enc) return JPGPU_ERR_ARGUMENT;
    enc->impl.emit_counters(one_pass, fell_back);
    return JPGPU_OK;
}
int jpgpu_encoder_encoded_size(const jpgpu_encoder *enc, int i, size_t *bytes) { return enc ? enc->impl.encoded_size(i, bytes) : JPGPU_ERR_ARGUMENT; }
int jpgpu_encoder_download(jpgpu_encoder *enc, int i, void *dst, size_t cap) { JPGPU_GUARD(enc, enc->impl.download(i, dst, cap)); }
void *jpgpu_encoder_output_device(const jpgpu_encoder *enc, int i, size_t *bytes) { return enc ? enc->impl.output_device(i, bytes) : nullptr; }
int jpgpu_encoder_download_coefficients(jpgpu_encoder *enc, int i, int16_t *dst, size_t cap_blocks) {
    JPGPU_GUARD(enc, enc->impl.download_coefficients(i, dst, cap_blocks));
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------ (5) optimizer

struct jpgpu_optimizer {
    jpgpu_ctx *ctx;
    OptimizeBatch impl;
    explicit jpgpu_optimizer(jpgpu_ctx *c) : ctx(c), impl(c) {}
};

extern "C" {

int jpgpu_optimizer_create(jpgpu_ctx *ctx, jpgpu_optimizer **out) {
    if (!ctx || !out) return JPGPU_ERR_ARGUMENT;
    *out = new (std::nothrow) jpgpu_optimizer(ctx);
    return *out ? JPGPU_OK : JPGPU_ERR_OUT_OF_MEMORY;
}
void jpgpu_optimizer_destroy(jpgpu_optimizer *opt) { delete opt; }
int jpgpu_optimizer_upload(jpgpu_optimizer *opt, const uint8_t *const *jpeg, const size_t *len, int n, int strip) {
    JPGPU_GUARD(opt, opt->impl.upload(jpeg, len, n, strip));
}
int jpgpu_optimizer_set_most_optimal_coding(jpgpu_optimizer *opt, int on) {
    if (!opt) return JPGPU_ERR_ARGUMENT;
    opt->impl.set_most_optimal_coding(on != 0);
    return JPGPU_OK;
}
int jpgpu_optimizer_run(jpgpu_optimizer *opt) { JPGPU_GUARD(opt, opt->impl.run()); }
int jpgpu_optimizer_result(jpgpu_optimizer *opt, int i, jpgpu_image_result *res, size_t *out_len) {
    JPGPU_GUARD(opt, opt->impl.result(i, res, out_len));
}
int jpgpu_optimizer_download(jpgpu_optimizer *opt, int i, void *dst, size_t cap) { JPGPU_GUARD(opt, opt->impl.download(i, dst, cap)); }
int jpgpu_optimizer_statistics(const jpgpu_optimizer *opt, int i, int table, uint8_t *table_class, uint8_t *identifier, uint32_t *counts) {
    if (!opt || !table_class || !identifier || !counts) return JPGPU_ERR_ARGUMENT;
    return opt->impl.statistics(i, table, table_class, identifier, counts) ? JPGPU_OK : JPGPU_ERR_ARGUMENT;
}
int jpgpu_optimizer_last_ms(const jpgpu_optimizer *opt, float *ms) {
    if (!opt || !ms) return JPGPU_ERR_ARGUMENT;
    *ms = opt->impl.last_ms();
    return JPGPU_OK;
}
int jpgpu_net_sort_permutation(const int32_t *keys, int n, int32_t *perm) {
    if (n < 0 || (n > 0 && (!keys || !perm))) return JPGPU_ERR_ARGUMENT;
    net_sort_permutation(keys, n, perm);
    return JPGPU_OK;
}
int jpgpu_build_optimal_huffman_table(const uint32_t *counts, int most_optimal, uint8_t *bits, uint8_t *values, int *num_values, uint16_t *code,
                                      uint8_t *length) {
    if (!counts || !bits || !values || !num_values) return JPGPU_ERR_ARGUMENT;
    std::vector<OptimalCode> codes;
    if (!build_optimal_table(counts, &codes, most_optimal != 0)) return JPGPU_ERR_INVALID_OPERATION;
    memset(bits, 0, 16);
    for (size_t i = 0; i < codes.size(); i++) {
        if (codes[i].length >= 1 && codes[i].length <= 16) bits[codes[i].length - 1]++;
        values[i] = codes[i].symbol;
    }
    *num_values = (int)codes.size();
    if (code && length) {
        for (int s = 0; s < 256; s++) {
            code[s] = codes[0].code;
            length[s] = codes[0].length;
        }
        for (const OptimalCode &c : codes) {
            code[c.symbol] = c.code;
            length[c.symbol] = c.length;
        }
    }
    return JPGPU_OK;
}

}  // extern "C"
