// capi.hip — the extern "C" boundary of libmi355rag.so (include/mi355rag.h): contexts, the
// flat index (FAISS.from_documents / as_retriever, advanced_rag.py:295-296) and the search
// orchestration: MFMA scan + certificate where available, canonical exact scan otherwise.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <mutex>
#include <new>
#include <vector>
#include "kernels.h"

// ---------------------------------------------------------------- errors
static thread_local char g_err[512] = "";
void mi_set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* mi_last_error(void) { return g_err; }
extern "C" const char* mi_version(void) { return "mi355rag 0.1 (gfx950)"; }

// ---------------------------------------------------------------- objects
struct mi_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::mutex mu;
    // ping-pong scratch of mi_topk_merge_dev for more than 4096 entries per query (grow-only, under mu)
    void* merge_buf[2] = {nullptr, nullptr};
    size_t merge_cap[2] = {0, 0};
};

int mi_ctx_device(mi_ctx* ctx) { return ctx->device; }
hipStream_t mi_ctx_stream(mi_ctx* ctx) { return ctx->stream; }

#define MI_ROW_SLACK 128
// host-buffer searches whose queries and results both fit in this go through a pinned block of the index
#define MI_PIN_BYTES (64u << 10)

// grow-only device buffer
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes)
    {
        if (bytes <= cap) return MI_OK;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        const size_t want = (bytes + 255) / 256 * 256;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) { p = nullptr; mi_set_error("hipMalloc(%zu) failed: %s", want, hipGetErrorString(e)); return MI_ERR_OOM; }
        cap = want;
        return MI_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    template <typename T> T* as() const { return reinterpret_cast<T*>(p); }
};

struct mi_index {
    mi_ctx* ctx = nullptr;
    int d = 0, d_pad = 0, metric = 0, dtype = 0, mode = MI_MODE_AUTO;
    int d_pad_scan = 0;         // row pitch the MFMA scan streams (128/384/768), 0 = no MFMA path
    int64_t n = 0, cap_rows = 0, id_base = 0;
    void* rows = nullptr;       // [cap_rows][d_pad] fp32 or bf16
    // F32 index only: bf16 (RNE) shadow of the rows, [shadow_cap][d_pad_scan], built lazily by the
    // first fast search and kept up to row shadow_n.  The scan filters on it; scores never come from it.
    DevBuf shadow;
    int64_t shadow_n = 0, shadow_cap = 0;
    float* xx = nullptr;        // [cap_rows] canonical squared norms
    float* max_xx = nullptr;    // device scalar: max over rows of xx
    std::mutex mu;
    // pinned host staging for small host-buffer searches (MI_PIN_BYTES, allocated by the first one)
    void* pin = nullptr;
    // recorded behind every add on the stream that carried it; a search on another stream waits on it
    hipEvent_t ev_data = nullptr;
    // search workspace (grow-only)
    DevBuf stage;               // host-buffer adds and get_rows go through it (grow-only, <= one 256-MB slab)
    DevBuf q_in, q_canon, q_bf16, qq, dense, pair_s, pair_i, lists, cnts, thrs, gthr, fail, part_s, part_i, sub_q, sub_qq, sub_s, sub_i, out_s, out_i;
    // stats
    int64_t st_queries = 0, st_fast = 0, st_exact = 0, st_rows = 0;
    // scan timing of the last search
    std::vector<hipEvent_t> ev_pool;
    int ev_used = 0;
    int last_kind = 0;
    float last_ms = 0.0f;
    int last_launches = 0;
    size_t elem() const { return dtype == MI_DTYPE_F32 ? 4 : 2; }
};

// the index's pinned host block: MI_PIN_BYTES of staging, then one 256-byte line for the certificate's fail count
static int ensure_pin(mi_index* x)
{
    if (x->pin) return MI_OK;
    hipError_t e = hipHostMalloc(&x->pin, MI_PIN_BYTES + 256, hipHostMallocDefault);
    if (e != hipSuccess) { x->pin = nullptr; mi_set_error("hipHostMalloc failed: %s", hipGetErrorString(e)); return MI_ERR_OOM; }
    return MI_OK;
}

// ---------------------------------------------------------------- context
extern "C" int mi_init(int device_id, mi_ctx** out)
{
    if (!out) { mi_set_error("mi_init: out is NULL"); return MI_ERR_INVALID_ARG; }
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        mi_set_error("mi_init: no HIP device visible (this library has no CPU fallback)");
        return MI_ERR_NO_DEVICE;
    }
    if (device_id < 0 || device_id >= ndev) { mi_set_error("mi_init: device %d out of range [0,%d)", device_id, ndev); return MI_ERR_INVALID_ARG; }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) != hipSuccess) { mi_set_error("mi_init: hipGetDeviceProperties failed"); return MI_ERR_NO_DEVICE; }
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        mi_set_error("mi_init: device %d is %s; this library is built for gfx950 (MI355X) only", device_id, prop.gcnArchName);
        return MI_ERR_NO_DEVICE;
    }
    DeviceGuard g(device_id);
    mi_ctx* c = new (std::nothrow) mi_ctx();
    if (!c) { mi_set_error("mi_init: out of host memory"); return MI_ERR_OOM; }
    c->device = device_id;
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete c; mi_set_error("mi_init: hipStreamCreate failed: %s", hipGetErrorString(e)); return MI_ERR_HIP; }
    *out = c;
    return MI_OK;
}

extern "C" void mi_ctx_destroy(mi_ctx* ctx)
{
    if (!ctx) return;
    DeviceGuard g(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipStreamDestroy(ctx->stream);
    for (void* b : ctx->merge_buf) if (b) (void)hipFree(b);
    delete ctx;
}

extern "C" int mi_ctx_synchronize(mi_ctx* ctx)
{
    MI_REQUIRE(ctx, "mi_ctx_synchronize: ctx is NULL");
    DeviceGuard g(ctx->device);
    MI_HIP(hipStreamSynchronize(ctx->stream));
    return MI_OK;
}

// ---------------------------------------------------------------- index: storage
extern "C" int mi_index_create(mi_ctx* ctx, int dim, int metric, int dtype, mi_index** out)
{
    MI_REQUIRE(ctx && out, "mi_index_create: NULL argument");
    *out = nullptr;
    MI_REQUIRE(dim > 0 && dim <= 65536, "mi_index_create: dim %d out of range", dim);
    MI_REQUIRE(metric == MI_METRIC_IP || metric == MI_METRIC_L2, "mi_index_create: unknown metric %d", metric);
    MI_REQUIRE(dtype == MI_DTYPE_F32 || dtype == MI_DTYPE_BF16, "mi_index_create: unknown dtype %d", dtype);
    DeviceGuard g(ctx->device);
    mi_index* x = new (std::nothrow) mi_index();
    if (!x) { mi_set_error("mi_index_create: out of host memory"); return MI_ERR_OOM; }
    x->ctx = ctx; x->d = dim; x->metric = metric; x->dtype = dtype;
    x->d_pad = mi_pad_dim(dim);
    // the MFMA scan is built for 128 / 384 / 768-element rows: pad a BF16 index up to the next one
    x->d_pad_scan = mi_mfma_pitch(dim);
    if (dtype == MI_DTYPE_BF16 && x->d_pad_scan) x->d_pad = x->d_pad_scan;
    hipError_t e = hipMalloc((void**)&x->max_xx, 256);
    if (e != hipSuccess) { delete x; mi_set_error("mi_index_create: hipMalloc failed: %s", hipGetErrorString(e)); return MI_ERR_OOM; }
    e = hipMemsetAsync(x->max_xx, 0, 256, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) { (void)hipFree(x->max_xx); delete x; mi_set_error("mi_index_create: memset failed"); return MI_ERR_HIP; }
    e = hipEventCreateWithFlags(&x->ev_data, hipEventDisableTiming);
    if (e != hipSuccess) { (void)hipFree(x->max_xx); delete x; mi_set_error("mi_index_create: hipEventCreate failed"); return MI_ERR_HIP; }
    *out = x;
    return MI_OK;
}

extern "C" void mi_index_destroy(mi_index* idx)
{
    if (!idx) return;
    DeviceGuard g(idx->ctx->device);
    (void)hipStreamSynchronize(idx->ctx->stream);
    if (idx->rows) (void)hipFree(idx->rows);
    if (idx->xx) (void)hipFree(idx->xx);
    if (idx->max_xx) (void)hipFree(idx->max_xx);
    DevBuf* bufs[] = {&idx->q_in, &idx->q_canon, &idx->q_bf16, &idx->qq, &idx->dense, &idx->pair_s, &idx->pair_i,
                      &idx->lists, &idx->cnts, &idx->thrs, &idx->gthr, &idx->fail, &idx->part_s, &idx->part_i, &idx->sub_q, &idx->sub_qq,
                      &idx->sub_s, &idx->sub_i, &idx->out_s, &idx->out_i, &idx->shadow, &idx->stage};
    for (DevBuf* b : bufs) b->release();
    for (hipEvent_t ev : idx->ev_pool) (void)hipEventDestroy(ev);
    if (idx->ev_data) (void)hipEventDestroy(idx->ev_data);
    if (idx->pin) (void)hipHostFree(idx->pin);
    delete idx;
}

static int index_reserve_locked(mi_index* x, int64_t rows_total)
{
    if (rows_total <= x->cap_rows) return MI_OK;
    if (rows_total >= ((int64_t)1 << 32)) { mi_set_error("index: more than 2^32-1 rows per shard is not supported"); return MI_ERR_UNSUPPORTED; }
    hipStream_t st = x->ctx->stream;
    void* nrows = nullptr;
    float* nxx = nullptr;
    const size_t row_bytes = (size_t)x->d_pad * x->elem();
    // + MI_ROW_SLACK rows: the MFMA scan streams whole 128-row tiles and may read (never use) up
    // to 127 rows past the last one
    hipError_t e = hipMalloc(&nrows, (size_t)(rows_total + MI_ROW_SLACK) * row_bytes);
    if (e != hipSuccess) { mi_set_error("index: hipMalloc(%zu) for rows failed: %s", (size_t)rows_total * row_bytes, hipGetErrorString(e)); return MI_ERR_OOM; }
    e = hipMalloc((void**)&nxx, (size_t)(rows_total + MI_ROW_SLACK) * sizeof(float));
    if (e != hipSuccess) { (void)hipFree(nrows); mi_set_error("index: hipMalloc for norms failed: %s", hipGetErrorString(e)); return MI_ERR_OOM; }
    // An earlier mi_index_add_dev on a caller's stream may still be writing the old rows: the copy waits for
    // the event recorded behind it (every add chains onto that event, so it covers all of them).  The
    // synchronise below then makes the new buffers complete before any stream can be handed them.
    hipError_t ce = hipSuccess;
    if (x->n > 0) {
        ce = hipStreamWaitEvent(st, x->ev_data, 0);
        if (ce == hipSuccess) ce = hipMemcpyAsync(nrows, x->rows, (size_t)x->n * row_bytes, hipMemcpyDeviceToDevice, st);
        if (ce == hipSuccess) ce = hipMemcpyAsync(nxx, x->xx, (size_t)x->n * sizeof(float), hipMemcpyDeviceToDevice, st);
    }
    if (ce == hipSuccess) ce = hipStreamSynchronize(st);
    if (ce != hipSuccess) {
        (void)hipFree(nrows); (void)hipFree(nxx);
        mi_set_error("index: regrow copy failed: %s", hipGetErrorString(ce));
        return MI_ERR_HIP;
    }
    if (x->rows) (void)hipFree(x->rows);
    if (x->xx) (void)hipFree(x->xx);
    x->rows = nrows; x->xx = nxx; x->cap_rows = rows_total;
    return MI_OK;
}

extern "C" int mi_index_reserve(mi_index* idx, int64_t n_rows)
{
    MI_REQUIRE(idx && n_rows >= 0, "mi_index_reserve: bad argument");
    std::lock_guard<std::mutex> lk(idx->mu);
    DeviceGuard g(idx->ctx->device);
    return index_reserve_locked(idx, n_rows);
}

static int grow_for(mi_index* x, int64_t add)
{
    const int64_t need = x->n + add;
    if (need <= x->cap_rows) return MI_OK;
    int64_t cap = x->cap_rows > 0 ? x->cap_rows + x->cap_rows / 2 : 1024;
    if (cap < need) cap = need;
    return index_reserve_locked(x, cap);
}

static inline void* row_ptr(mi_index* x, int64_t r) { return (char*)x->rows + (size_t)r * x->d_pad * x->elem(); }

// Caller holds idx->mu and has set the device.
static int add_dev_locked(mi_index* idx, int64_t n, const void* x_dev, int src_dtype, hipStream_t st)
{
    MI_TRY(grow_for(idx, n));
    // adds may arrive on different streams: this one runs behind every earlier one, so the event recorded at
    // its end covers them all (a search waits on that one event)
    if (idx->n > 0) MI_HIP(hipStreamWaitEvent(st, idx->ev_data, 0));
    MI_TRY(mi_launch_ingest(st, x_dev, src_dtype, idx->d, n, idx->d, idx->d_pad, idx->dtype, row_ptr(idx, idx->n)));
    MI_TRY(mi_launch_row_norms(st, row_ptr(idx, idx->n), idx->dtype, n, idx->d_pad, idx->xx + idx->n, idx->max_xx));
    MI_HIP(hipEventRecord(idx->ev_data, st));
    idx->n += n;
    return MI_OK;
}

extern "C" int mi_index_add_dev(mi_index* idx, int64_t n, const void* x_dev, int src_dtype, void* stream)
{
    MI_REQUIRE(idx && n >= 0 && (x_dev || n == 0), "mi_index_add_dev: bad argument");
    MI_REQUIRE(src_dtype == MI_DTYPE_F32 || src_dtype == MI_DTYPE_BF16, "mi_index_add_dev: unknown src dtype");
    if (n == 0) return MI_OK;
    std::lock_guard<std::mutex> lk(idx->mu);
    DeviceGuard g(idx->ctx->device);
    return add_dev_locked(idx, n, x_dev, src_dtype, mi_pick_stream(stream, idx->ctx->stream));
}

// Host rows of either dtype, staged through a device buffer in slabs (bounded extra memory for huge adds).
static int add_host_locked(mi_index* idx, int64_t n, const void* x_host, int src_dtype)
{
    hipStream_t st = idx->ctx->stream;
    const size_t esz = src_dtype == MI_DTYPE_F32 ? 4 : 2;
    const size_t row_bytes = (size_t)idx->d * esz;
    const int64_t slab = (int64_t)((256ull << 20) / row_bytes) > 0 ? (int64_t)((256ull << 20) / row_bytes) : 1;
    const int64_t first = n < slab ? n : slab;
    MI_TRY(idx->stage.ensure((size_t)first * row_bytes));
    for (int64_t off = 0; off < n; off += slab) {
        const int64_t m = (n - off < slab) ? n - off : slab;
        MI_HIP(hipMemcpyAsync(idx->stage.p, static_cast<const char*>(x_host) + (size_t)off * row_bytes, (size_t)m * row_bytes, hipMemcpyHostToDevice, st));
        MI_TRY(add_dev_locked(idx, m, idx->stage.p, src_dtype, st));
        MI_HIP(hipStreamSynchronize(st));          // the slab is reused; the caller's buffer is free on return
    }
    return MI_OK;
}

extern "C" int mi_index_add(mi_index* idx, int64_t n, const float* x_host)
{
    MI_REQUIRE(idx && n >= 0 && (x_host || n == 0), "mi_index_add: bad argument");
    if (n == 0) return MI_OK;
    // one lock for the whole call: the rows of two concurrent adds never interleave
    std::lock_guard<std::mutex> lk(idx->mu);
    DeviceGuard g(idx->ctx->device);
    return add_host_locked(idx, n, x_host, MI_DTYPE_F32);
}

extern "C" int mi_index_add_raw(mi_index* idx, int64_t n, const void* x_host, int src_dtype)
{
    MI_REQUIRE(idx && n >= 0 && (x_host || n == 0), "mi_index_add_raw: bad argument");
    MI_REQUIRE(src_dtype == MI_DTYPE_F32 || src_dtype == MI_DTYPE_BF16, "mi_index_add_raw: unknown src dtype");
    if (n == 0) return MI_OK;
    std::lock_guard<std::mutex> lk(idx->mu);
    DeviceGuard g(idx->ctx->device);
    return add_host_locked(idx, n, x_host, src_dtype);
}

// Appends n synthetic rows that are rows [global_row0, global_row0 + n) of the generator's sequence (global_row0 < 0:
// the index's own numbering, id_base + ntotal).  mi_multi_index_add_synthetic places blocks of one corpus on several
// shards through this.
int mi_index_add_synthetic_at(mi_index* idx, int64_t n, uint64_t seed, int normalize, int64_t global_row0)
{
    MI_REQUIRE(idx && n >= 0, "mi_index_add_synthetic: bad argument");
    if (n == 0) return MI_OK;
    std::lock_guard<std::mutex> lk(idx->mu);
    DeviceGuard g(idx->ctx->device);
    hipStream_t st = idx->ctx->stream;
    MI_TRY(grow_for(idx, n));
    const int64_t g0 = global_row0 >= 0 ? global_row0 : idx->id_base + idx->n;
    // chunked so the launch grid stays small
    const int64_t chunk = (int64_t)1 << 24;
    for (int64_t off = 0; off < n; off += chunk) {
        const int64_t m = (n - off < chunk) ? n - off : chunk;
        MI_TRY(mi_launch_synth(st, row_ptr(idx, idx->n + off), idx->dtype, m, idx->d, idx->d_pad, seed,
                               g0 + off, normalize));
        MI_TRY(mi_launch_row_norms(st, row_ptr(idx, idx->n + off), idx->dtype, m, idx->d_pad, idx->xx + idx->n + off, idx->max_xx));
    }
    MI_HIP(hipStreamSynchronize(st));      // later searches may run on a caller's stream
    idx->n += n;
    return MI_OK;
}

extern "C" int mi_index_add_synthetic(mi_index* idx, int64_t n, uint64_t seed, int normalize)
{
    return mi_index_add_synthetic_at(idx, n, seed, normalize, -1);
}

extern "C" int64_t mi_index_ntotal(const mi_index* idx) { return idx ? idx->n : -1; }
extern "C" int mi_index_dim(const mi_index* idx) { return idx ? idx->d : -1; }
extern "C" int mi_index_set_id_base(mi_index* idx, int64_t id_base)
{
    MI_REQUIRE(idx && id_base >= 0, "mi_index_set_id_base: bad argument");
    std::lock_guard<std::mutex> lk(idx->mu);
    idx->id_base = id_base;
    return MI_OK;
}
extern "C" int mi_index_set_mode(mi_index* idx, int mode)
{
    MI_REQUIRE(idx && mode >= MI_MODE_AUTO && mode <= MI_MODE_FAST, "mi_index_set_mode: bad argument");
    std::lock_guard<std::mutex> lk(idx->mu);
    idx->mode = mode;
    return MI_OK;
}
extern "C" int mi_index_reset(mi_index* idx)
{
    MI_REQUIRE(idx, "mi_index_reset: idx is NULL");
    std::lock_guard<std::mutex> lk(idx->mu);
    DeviceGuard g(idx->ctx->device);
    idx->n = 0;
    idx->shadow_n = 0;
    MI_HIP(hipMemsetAsync(idx->max_xx, 0, 256, idx->ctx->stream));
    MI_HIP(hipStreamSynchronize(idx->ctx->stream));
    return MI_OK;
}

extern "C" int mi_index_get_rows(mi_index* idx, int64_t row0, int64_t n, float* out_host)
{
    MI_REQUIRE(idx && out_host && row0 >= 0 && n >= 0, "mi_index_get_rows: bad argument");
    std::lock_guard<std::mutex> lk(idx->mu);
    MI_REQUIRE(row0 + n <= idx->n, "mi_index_get_rows: range [%lld,%lld) outside [0,%lld)",
               (long long)row0, (long long)(row0 + n), (long long)idx->n);
    if (n == 0) return MI_OK;
    DeviceGuard g(idx->ctx->device);
    hipStream_t st = idx->ctx->stream;
    MI_HIP(hipStreamWaitEvent(st, idx->ev_data, 0));           // rows added on another stream
    const int64_t slab = (int64_t)((256ull << 20) / ((size_t)idx->d * 4)) > 0 ? (int64_t)((256ull << 20) / ((size_t)idx->d * 4)) : 1;
    MI_TRY(idx->stage.ensure((size_t)(n < slab ? n : slab) * idx->d * 4));
    for (int64_t off = 0; off < n; off += slab) {
        const int64_t m = (n - off < slab) ? n - off : slab;
        MI_TRY(mi_launch_export_rows(st, row_ptr(idx, row0 + off), idx->dtype, m, idx->d, idx->d_pad, idx->stage.as<float>()));
        MI_HIP(hipMemcpyAsync(out_host + off * idx->d, idx->stage.p, (size_t)m * idx->d * 4, hipMemcpyDeviceToHost, st));
        MI_HIP(hipStreamSynchronize(st));
    }
    return MI_OK;
}

// Rows [row0, row0 + n) in the index's own storage type (fp32, or bf16 bit patterns), row pitch = dim elements: what a
// shard file holds (no widening to fp32 on the way out).
extern "C" int mi_index_get_rows_raw(mi_index* idx, int64_t row0, int64_t n, void* out_host)
{
    MI_REQUIRE(idx && out_host && row0 >= 0 && n >= 0, "mi_index_get_rows_raw: bad argument");
    std::lock_guard<std::mutex> lk(idx->mu);
    MI_REQUIRE(row0 + n <= idx->n, "mi_index_get_rows_raw: range [%lld,%lld) outside [0,%lld)",
               (long long)row0, (long long)(row0 + n), (long long)idx->n);
    if (n == 0) return MI_OK;
    DeviceGuard g(idx->ctx->device);
    hipStream_t st = idx->ctx->stream;
    MI_HIP(hipStreamWaitEvent(st, idx->ev_data, 0));
    const size_t esz = idx->elem();
    MI_HIP(hipMemcpy2DAsync(out_host, (size_t)idx->d * esz, row_ptr(idx, row0), (size_t)idx->d_pad * esz, (size_t)idx->d * esz, (size_t)n,
                            hipMemcpyDeviceToHost, st));
    MI_HIP(hipStreamSynchronize(st));
    return MI_OK;
}

extern "C" int mi_index_dtype(const mi_index* idx) { return idx ? idx->dtype : -1; }

extern "C" int mi_index_stats(const mi_index* idx, int64_t out[4])
{
    MI_REQUIRE(idx && out, "mi_index_stats: NULL argument");
    std::lock_guard<std::mutex> lk(const_cast<mi_index*>(idx)->mu);
    out[0] = idx->st_queries; out[1] = idx->st_fast; out[2] = idx->st_exact; out[3] = idx->st_rows;
    return MI_OK;
}

// ---------------------------------------------------------------- scan timing (HIP events on the scan's own stream)
static int ev_begin(mi_index* x, hipStream_t st)
{
    if ((size_t)x->ev_used + 2 > x->ev_pool.size()) {
        for (int i = 0; i < 2; ++i) {
            hipEvent_t ev;
            MI_HIP(hipEventCreate(&ev));
            x->ev_pool.push_back(ev);
        }
    }
    MI_HIP(hipEventRecord(x->ev_pool[x->ev_used], st));
    return MI_OK;
}
static int ev_end(mi_index* x, hipStream_t st)
{
    MI_HIP(hipEventRecord(x->ev_pool[x->ev_used + 1], st));
    x->ev_used += 2;
    return MI_OK;
}

extern "C" int mi_index_last_scan_time(const mi_index* idx, float* out_ms, int* out_launches, int* out_kind)
{
    MI_REQUIRE(idx, "mi_index_last_scan_time: idx is NULL");
    mi_index* x = const_cast<mi_index*>(idx);
    std::lock_guard<std::mutex> lk(x->mu);
    if (x->ev_used > 0) {
        DeviceGuard g(x->ctx->device);
        float total = 0.0f;
        for (int i = 0; i + 1 < x->ev_used; i += 2) {
            MI_HIP(hipEventSynchronize(x->ev_pool[i + 1]));
            float ms = 0.0f;
            MI_HIP(hipEventElapsedTime(&ms, x->ev_pool[i], x->ev_pool[i + 1]));
            total += ms;
        }
        x->last_ms = total;
        x->last_launches = x->ev_used / 2;
        x->ev_used = 0;
    }
    if (out_ms) *out_ms = x->last_ms;
    if (out_launches) *out_launches = x->last_launches;
    if (out_kind) *out_kind = x->last_kind;
    return MI_OK;
}

// ---------------------------------------------------------------- search: exact path
// Scores nq prepared queries (q_canon/qq, device) against all rows in dense segments and
// leaves the best k per query at out_s/out_i (device, row stride k).
static int exact_search(mi_index* x, hipStream_t st, const float* q_canon, const float* qq, int nq, int k,
                        float* out_s, int64_t* out_i, bool timed)
{
    if (nq <= 0) return MI_OK;
    const int64_t n = x->n;
    if (n == 0) return mi_launch_fill_pad(st, x->metric, nq, k, out_s, out_i, k);
    // dense workspace: up to 256M scores (1 GiB); queries in chunks of <= 4096
    const int64_t ws_elems = (int64_t)256 << 20;
    const int qc_max = nq < 4096 ? nq : 4096;
    int64_t seg_rows = ws_elems / qc_max / 256 * 256;
    if (seg_rows > ((int64_t)65535 * 256)) seg_rows = (int64_t)65535 * 256;
    // a short index is ONE segment: round its length up, not down (2,000 rows used to become 1,792 + 208,
    // two scans, two selects and two merges)
    if (seg_rows > (n + 255) / 256 * 256) seg_rows = (n + 255) / 256 * 256;
    if (seg_rows < 256) seg_rows = 256;
    MI_TRY(x->dense.ensure((size_t)qc_max * seg_rows * sizeof(float)));
    const bool multi = n > seg_rows;
    if (multi) {
        MI_TRY(x->pair_s.ensure((size_t)qc_max * 2 * k * sizeof(float)));
        MI_TRY(x->pair_i.ensure((size_t)qc_max * 2 * k * sizeof(int64_t)));
    }
    float* S = x->dense.as<float>();
    for (int q0 = 0; q0 < nq; q0 += qc_max) {
        const int qc = (nq - q0 < qc_max) ? nq - q0 : qc_max;
        float* os = out_s + (int64_t)q0 * k;
        int64_t* oi = out_i + (int64_t)q0 * k;
        if (multi) MI_TRY(mi_launch_fill_pad(st, x->metric, qc, k, x->pair_s.as<float>(), x->pair_i.as<int64_t>(), 2 * k));
        for (int64_t r0 = 0; r0 < n; r0 += seg_rows) {
            const int nr = (int)((n - r0 < seg_rows) ? n - r0 : seg_rows);
            if (timed) MI_TRY(ev_begin(x, st));
            MI_TRY(mi_launch_scan_dense(st, x->rows, x->dtype, r0, nr, x->d_pad, q_canon + (int64_t)q0 * x->d_pad, qc, S, seg_rows));
            if (timed) MI_TRY(ev_end(x, st));
            // where the segment's k best go: the answer itself, or the right half of the running pair
            float* ds = multi ? x->pair_s.as<float>() + k : os;
            int64_t* di = multi ? x->pair_i.as<int64_t>() + k : oi;
            const int64_t dstride = multi ? 2 * k : k;
            int seg_len = nr;
            const int pieces = mi_select_dense_pieces(nr, qc, k, &seg_len);
            if (pieces == 1) {
                MI_TRY(mi_launch_select_dense(st, S, seg_rows, nr, nr, x->xx + r0, qq + q0, x->metric, qc, k, x->id_base + r0, ds, di, dstride));
            } else {
                // few queries, long segment: one workgroup per (query, piece), then one merge of the pieces' lists
                const int64_t m = (int64_t)pieces * k;
                MI_TRY(x->part_s.ensure((size_t)qc * m * sizeof(float)));
                MI_TRY(x->part_i.ensure((size_t)qc * m * sizeof(int64_t)));
                MI_TRY(mi_launch_select_dense(st, S, seg_rows, nr, seg_len, x->xx + r0, qq + q0, x->metric, qc, k, x->id_base + r0,
                                              x->part_s.as<float>(), x->part_i.as<int64_t>(), m));
                MI_TRY(mi_launch_merge(st, x->metric, qc, (int)m, x->part_s.as<float>(), x->part_i.as<int64_t>(), m, k, ds, di, dstride));
            }
            if (multi) {
                const bool last = r0 + seg_rows >= n;
                if (!last) MI_TRY(mi_launch_merge(st, x->metric, qc, 2 * k, x->pair_s.as<float>(), x->pair_i.as<int64_t>(), 2 * k, k,
                                                  x->pair_s.as<float>(), x->pair_i.as<int64_t>(), 2 * k));
                else MI_TRY(mi_launch_merge(st, x->metric, qc, 2 * k, x->pair_s.as<float>(), x->pair_i.as<int64_t>(), 2 * k, k, os, oi, k));
            }
        }
    }
    return MI_OK;
}

// ---------------------------------------------------------------- search: orchestration
// F32 index: bring the bf16 shadow up to date with rows [0, n).
static int ensure_shadow(mi_index* x, hipStream_t st)
{
    if (x->shadow_cap < x->cap_rows) {
        MI_HIP(hipStreamSynchronize(st));
        MI_TRY(x->shadow.ensure((size_t)(x->cap_rows + MI_ROW_SLACK) * x->d_pad_scan * sizeof(uint16_t)));
        x->shadow_cap = x->cap_rows;
        x->shadow_n = 0;
    }
    if (x->shadow_n < x->n) {
        const int64_t m = x->n - x->shadow_n;
        MI_TRY(mi_launch_ingest(st, (const float*)x->rows + x->shadow_n * x->d_pad, MI_DTYPE_F32, x->d_pad, m, x->d, x->d_pad_scan,
                                MI_DTYPE_BF16, x->shadow.as<uint16_t>() + x->shadow_n * x->d_pad_scan));
        x->shadow_n = x->n;
    }
    return MI_OK;
}

static int search_dev_locked(mi_index* x, int64_t nq64, const float* q_dev, int k, float* out_s, int64_t* out_i, hipStream_t st)
{
    MI_REQUIRE(nq64 >= 0 && nq64 < ((int64_t)1 << 24), "search: nq %lld out of range", (long long)nq64);
    MI_REQUIRE(k >= 1, "search: k must be >= 1");
    if (k > MI_MAX_K) { mi_set_error("search: k=%d exceeds MI_MAX_K=%d", k, MI_MAX_K); return MI_ERR_UNSUPPORTED; }
    const int nq = (int)nq64;
    if (nq == 0) return MI_OK;
    const int64_t n = x->n;
    x->ev_used = 0;
    if (n > 0) MI_HIP(hipStreamWaitEvent(st, x->ev_data, 0));      // rows added on another stream

    bool fast = false;
    mi_mfma_plan plan{};
    if (x->mode != MI_MODE_EXACT && x->d_pad_scan != 0 && mi_mfma_supported(x->d_pad_scan, k) &&
        mi_mfma_make_plan(n, x->d_pad_scan, nq, k, &plan) == MI_OK)
        fast = true;
    // A handful of queries over up to a million rows is a latency problem: the MFMA route is scan + finish +
    // certify + a D2H of the fail count (~130 us beyond the scan), the exact kernels are scan + select + merge.
    // Measured per call, 1 query, L2 (f32 / bf16 index): 200,000 rows 115 / 94 us exact vs 183 / 181 us MFMA;
    // 10^6 rows 372 / 245 vs 344 / 343 (AUTO only: FAST still forces the MFMA scan).
    if (fast && x->mode == MI_MODE_AUTO && nq <= 4 && n <= ((int64_t)1 << 20)) fast = false;
    if (x->mode == MI_MODE_FAST && !fast) {
        mi_set_error("search: MI_MODE_FAST requested but the MFMA path does not cover this index (dtype=%d metric=%d d=%d n=%lld k=%d)",
                     x->dtype, x->metric, x->d, (long long)n, k);
        return MI_ERR_UNSUPPORTED;
    }
    const bool bf16_index = x->dtype == MI_DTYPE_BF16;

    const int nq_alloc = fast ? plan.nq_pad : nq;
    MI_TRY(x->q_canon.ensure((size_t)nq_alloc * x->d_pad * sizeof(float)));
    MI_TRY(x->qq.ensure((size_t)nq_alloc * sizeof(float)));
    uint16_t* qb = nullptr;
    if (fast) {
        MI_TRY(x->q_bf16.ensure((size_t)plan.nq_pad * x->d_pad_scan * sizeof(uint16_t)));
        qb = x->q_bf16.as<uint16_t>();
        if (plan.nq_pad > nq)
            MI_HIP(hipMemsetAsync(qb + (size_t)nq * x->d_pad_scan, 0, (size_t)(plan.nq_pad - nq) * x->d_pad_scan * sizeof(uint16_t), st));
    }
    MI_TRY(mi_launch_prep_queries(st, q_dev, nq, x->d, x->d_pad, bf16_index ? 1 : 0, x->q_canon.as<float>(),
                                  x->d_pad_scan, qb, x->qq.as<float>()));

    x->st_queries += nq;
    x->st_rows = n;
    if (!fast) {
        x->last_kind = 0;
        MI_TRY(exact_search(x, st, x->q_canon.as<float>(), x->qq.as<float>(), nq, k, out_s, out_i, true));
        x->st_exact += nq;
        MI_HIP(hipStreamSynchronize(st));
        return MI_OK;
    }

    // ---- MFMA scan + finish + certificate
    x->last_kind = 1;
    const uint16_t* scan_rows = (const uint16_t*)x->rows;
    if (!bf16_index) {
        MI_TRY(ensure_shadow(x, st));
        scan_rows = x->shadow.as<uint16_t>();
    }
    float eps_acc = 0.0f, eps_round = 0.0f;
    mi_mfma_eps(x->d_pad_scan, !bf16_index, &eps_acc, &eps_round);
    MI_TRY(x->lists.ensure(plan.list_bytes));
    MI_TRY(x->cnts.ensure(plan.cnt_bytes));
    MI_TRY(x->thrs.ensure(plan.thr_bytes));
    // shared thresholds [nq_pad], then the scan's per-slice progress words [nslices][64]
    // (and the seed pass's own block behind them: kernels.h)
    MI_TRY(x->gthr.ensure(plan.gthr_bytes));
    MI_HIP(hipMemsetAsync(x->gthr.p, 0, plan.gthr_bytes, st));
    // fail: [0] = count, [64 ..) = list of failed queries, then the per-query overflow flags
    MI_TRY(x->fail.ensure((size_t)(2 * (size_t)nq + 64) * sizeof(int32_t)));
    int32_t* fail_count = x->fail.as<int32_t>();
    int32_t* fail_list = fail_count + 64;
    int32_t* over_flags = fail_list + nq;
    MI_HIP(hipMemsetAsync(fail_count, 0, 256, st));
    MI_HIP(hipMemsetAsync(over_flags, 0, (size_t)nq * sizeof(int32_t), st));
    MI_TRY(ev_begin(x, st));
    MI_TRY(mi_launch_mfma_seed(st, &plan, x->metric, scan_rows, x->xx, x->d_pad_scan, qb, nq, x->gthr.as<uint32_t>()));
    MI_TRY(mi_launch_mfma_scan(st, &plan, x->metric, scan_rows, x->xx, n, x->d_pad_scan, qb, nq,
                               x->lists.as<uint64_t>(), x->cnts.as<int32_t>(), x->thrs.as<float>(), x->gthr.as<uint32_t>()));
    MI_TRY(ev_end(x, st));
    if (plan.ngroups == 1) {
        MI_TRY(mi_launch_mfma_finish(st, &plan, x->metric, x->rows, x->dtype, x->xx, x->max_xx, x->d_pad, eps_acc, eps_round,
                                     x->q_canon.as<float>(), x->qq.as<float>(), nq, k, x->id_base,
                                     x->lists.as<uint64_t>(), x->cnts.as<int32_t>(), out_s, out_i, k, over_flags));
    } else {
        const int64_t m = (int64_t)plan.ngroups * k;
        MI_TRY(x->part_s.ensure((size_t)nq * m * sizeof(float)));
        MI_TRY(x->part_i.ensure((size_t)nq * m * sizeof(int64_t)));
        MI_TRY(mi_launch_mfma_finish(st, &plan, x->metric, x->rows, x->dtype, x->xx, x->max_xx, x->d_pad, eps_acc, eps_round,
                                     x->q_canon.as<float>(), x->qq.as<float>(), nq, k, x->id_base,
                                     x->lists.as<uint64_t>(), x->cnts.as<int32_t>(),
                                     x->part_s.as<float>(), x->part_i.as<int64_t>(), m, over_flags));
        MI_TRY(mi_launch_merge(st, x->metric, nq, (int)m, x->part_s.as<float>(), x->part_i.as<int64_t>(), m, k, out_s, out_i, k));
    }
    MI_TRY(mi_launch_mfma_certify(st, &plan, x->metric, nq, k, eps_acc, eps_round, x->qq.as<float>(), x->max_xx,
                                  x->cnts.as<int32_t>(), x->thrs.as<float>(), out_s, out_i, over_flags, fail_list, fail_count));
    MI_TRY(ensure_pin(x));
    volatile int32_t* nfail_pin = reinterpret_cast<int32_t*>(static_cast<char*>(x->pin) + MI_PIN_BYTES);
    MI_HIP(hipMemcpyAsync(const_cast<int32_t*>(nfail_pin), fail_count, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    MI_HIP(hipStreamSynchronize(st));
    const int32_t nfail = *nfail_pin;
    x->st_fast += nq - nfail;
    if (nfail > 0) {
        // certificate failed for nfail queries: re-run exactly those on the exact path
        x->st_exact += nfail;
        MI_TRY(x->sub_q.ensure((size_t)nfail * x->d_pad * sizeof(float)));
        MI_TRY(x->sub_qq.ensure((size_t)nfail * sizeof(float)));
        MI_TRY(x->sub_s.ensure((size_t)nfail * k * sizeof(float)));
        MI_TRY(x->sub_i.ensure((size_t)nfail * k * sizeof(int64_t)));
        MI_TRY(mi_launch_gather_queries(st, fail_list, nfail, x->d_pad, x->q_canon.as<float>(), x->qq.as<float>(),
                                        x->sub_q.as<float>(), x->sub_qq.as<float>()));
        MI_TRY(exact_search(x, st, x->sub_q.as<float>(), x->sub_qq.as<float>(), nfail, k, x->sub_s.as<float>(), x->sub_i.as<int64_t>(), false));
        MI_TRY(mi_launch_copy_lists(st, nfail, k, x->sub_s.as<float>(), x->sub_i.as<int64_t>(), k, fail_list, out_s, out_i, k));
        MI_HIP(hipStreamSynchronize(st));
    }
    return MI_OK;
}

extern "C" int mi_index_search_dev(mi_index* idx, int64_t nq, const float* q_dev, int k,
                                   float* out_scores_dev, int64_t* out_ids_dev, void* stream)
{
    MI_REQUIRE(idx && (nq == 0 || (q_dev && out_scores_dev && out_ids_dev)), "mi_index_search_dev: NULL argument");
    std::lock_guard<std::mutex> lk(idx->mu);
    DeviceGuard g(idx->ctx->device);
    hipStream_t st = mi_pick_stream(stream, idx->ctx->stream);
    return search_dev_locked(idx, nq, q_dev, k, out_scores_dev, out_ids_dev, st);
}

extern "C" int mi_index_search(mi_index* idx, int64_t nq, const float* q_host, int k, float* out_scores, int64_t* out_ids)
{
    MI_REQUIRE(idx && (nq == 0 || (q_host && out_scores && out_ids)), "mi_index_search: NULL argument");
    MI_REQUIRE(nq >= 0 && k >= 1, "mi_index_search: nq/k out of range");
    if (nq == 0) return MI_OK;
    std::lock_guard<std::mutex> lk(idx->mu);
    DeviceGuard g(idx->ctx->device);
    hipStream_t st = idx->ctx->stream;
    const size_t q_bytes = (size_t)nq * idx->d * sizeof(float);
    const size_t s_bytes = (size_t)nq * k * sizeof(float), i_bytes = (size_t)nq * k * sizeof(int64_t);
    const size_t i_off = (s_bytes + 255) & ~(size_t)255;      // ids follow the scores in one buffer: one copy back
    MI_TRY(idx->q_in.ensure(q_bytes));
    MI_TRY(idx->out_s.ensure(i_off + i_bytes));
    float* ds = idx->out_s.as<float>();
    int64_t* di = reinterpret_cast<int64_t*>(idx->out_s.as<char>() + i_off);
    if (q_bytes <= MI_PIN_BYTES && i_off + i_bytes <= MI_PIN_BYTES) {
        // the interactive case (a query or a handful): pageable copies cost a staging hop and a sync each inside the
        // runtime, so go through our own pinned block, one copy each way
        MI_TRY(ensure_pin(idx));
        memcpy(idx->pin, q_host, q_bytes);
        MI_HIP(hipMemcpyAsync(idx->q_in.p, idx->pin, q_bytes, hipMemcpyHostToDevice, st));
        MI_TRY(search_dev_locked(idx, nq, idx->q_in.as<float>(), k, ds, di, st));
        MI_HIP(hipMemcpyAsync(idx->pin, idx->out_s.p, i_off + i_bytes, hipMemcpyDeviceToHost, st));
        MI_HIP(hipStreamSynchronize(st));
        memcpy(out_scores, idx->pin, s_bytes);
        memcpy(out_ids, static_cast<char*>(idx->pin) + i_off, i_bytes);
        return MI_OK;
    }
    MI_HIP(hipMemcpyAsync(idx->q_in.p, q_host, q_bytes, hipMemcpyHostToDevice, st));
    MI_TRY(search_dev_locked(idx, nq, idx->q_in.as<float>(), k, ds, di, st));
    MI_HIP(hipMemcpyAsync(out_scores, ds, s_bytes, hipMemcpyDeviceToHost, st));
    MI_HIP(hipMemcpyAsync(out_ids, di, i_bytes, hipMemcpyDeviceToHost, st));
    MI_HIP(hipStreamSynchronize(st));
    return MI_OK;
}

// ---------------------------------------------------------------- merge of per-shard lists
// Up to MI_MERGE_CHUNK entries per query fit one merge_kernel workgroup.  Longer inputs (8 ranks x k = 2048) are
// merged in passes: every chunk of MI_MERGE_CHUNK entries keeps its k best (k <= MI_MAX_K = half a chunk, so a pass
// at least halves the list), until one chunk is left.  The intermediate lists live in two grow-only buffers of the
// ctx; queries are taken in groups so those stay below ~128 MB each.
#define MI_MERGE_CHUNK 4096
static int merge_scratch(mi_ctx* ctx, int which, size_t bytes, hipStream_t st)
{
    if (bytes <= ctx->merge_cap[which]) return MI_OK;
    if (ctx->merge_buf[which]) { MI_HIP(hipStreamSynchronize(st)); (void)hipFree(ctx->merge_buf[which]); ctx->merge_buf[which] = nullptr; ctx->merge_cap[which] = 0; }
    hipError_t e = hipMalloc(&ctx->merge_buf[which], bytes);
    if (e != hipSuccess) { ctx->merge_buf[which] = nullptr; mi_set_error("merge: hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e)); return MI_ERR_OOM; }
    ctx->merge_cap[which] = bytes;
    return MI_OK;
}

// Layout of the first pass's input (see merge_kernel).
struct MergeIn { const float* s; const int64_t* i; int k_in; int64_t lstride_s, lstride_i, qstride; int64_t q_step_s, q_step_i; };

static int merge_any(mi_ctx* ctx, hipStream_t st, int metric, int64_t nq, int64_t m, const MergeIn& in,
                     int k, float* out_s, int64_t* out_i)
{
    if (m <= MI_MERGE_CHUNK)
        return mi_launch_merge_ex(st, metric, nq, (int)m, (int)m, in.s, in.i, in.k_in, in.lstride_s, in.lstride_i, in.qstride, k, out_s, out_i, k);
    std::lock_guard<std::mutex> lk(ctx->mu);
    const int64_t m1 = (m + MI_MERGE_CHUNK - 1) / MI_MERGE_CHUNK * k;           // entries per query after the first pass
    int64_t qg = ((int64_t)128 << 20) / (m1 * 12);
    if (qg < 1) qg = 1;
    if (qg > nq) qg = nq;
    const size_t s_bytes = ((size_t)qg * m1 * sizeof(float) + 255) & ~(size_t)255;
    MI_TRY(merge_scratch(ctx, 0, s_bytes + (size_t)qg * m1 * sizeof(int64_t), st));
    MI_TRY(merge_scratch(ctx, 1, s_bytes + (size_t)qg * m1 * sizeof(int64_t), st));   // pass 2 writes at most half of it
    for (int64_t q0 = 0; q0 < nq; q0 += qg) {
        const int64_t qc = (nq - q0 < qg) ? nq - q0 : qg;
        const float* cs = in.s + q0 * in.q_step_s;
        const int64_t* ci = in.i + q0 * in.q_step_i;
        int ck = in.k_in;
        int64_t cm = m, cls = in.lstride_s, cli = in.lstride_i, cq = in.qstride;
        int which = 0;
        while (cm > MI_MERGE_CHUNK) {
            const int64_t nch = (cm + MI_MERGE_CHUNK - 1) / MI_MERGE_CHUNK;
            float* os = reinterpret_cast<float*>(ctx->merge_buf[which]);
            int64_t* oi = reinterpret_cast<int64_t*>(static_cast<char*>(ctx->merge_buf[which]) + s_bytes);
            MI_TRY(mi_launch_merge_ex(st, metric, qc, (int)cm, MI_MERGE_CHUNK, cs, ci, ck, cls, cli, cq, k, os, oi, nch * k));
            cs = os; ci = oi; cm = nch * k; ck = (int)cm; cls = 0; cli = 0; cq = cm;
            which ^= 1;
        }
        MI_TRY(mi_launch_merge_ex(st, metric, qc, (int)cm, (int)cm, cs, ci, ck, cls, cli, cq, k, out_s + q0 * k, out_i + q0 * k, k));
    }
    // the ctx's two scratch buffers serve every caller and stream: the next multi-pass merge (maybe on another stream)
    // must not start while this one still reads them.  Rare path (more than 4096 pairs per query): a host sync is cheap.
    MI_HIP(hipStreamSynchronize(st));
    return MI_OK;
}

extern "C" int mi_topk_merge_dev(mi_ctx* ctx, int metric, int64_t nq, int nlists, int k_in,
                                 const float* in_scores_dev, const int64_t* in_ids_dev,
                                 int k, float* out_scores_dev, int64_t* out_ids_dev, void* stream)
{
    MI_REQUIRE(ctx && nq >= 0 && nlists >= 1 && k_in >= 1 && k >= 1, "mi_topk_merge_dev: bad argument");
    MI_REQUIRE(metric == MI_METRIC_IP || metric == MI_METRIC_L2, "mi_topk_merge_dev: unknown metric");
    if (k > MI_MAX_K) { mi_set_error("mi_topk_merge_dev: k=%d exceeds MI_MAX_K=%d", k, MI_MAX_K); return MI_ERR_UNSUPPORTED; }
    if (nq == 0) return MI_OK;
    MI_REQUIRE(in_scores_dev && in_ids_dev && out_scores_dev && out_ids_dev, "mi_topk_merge_dev: NULL buffer");
    const int64_t m = (int64_t)nlists * k_in;
    MI_REQUIRE(m < ((int64_t)1 << 24), "mi_topk_merge_dev: nlists*k_in = %lld is out of range", (long long)m);
    DeviceGuard g(ctx->device);
    hipStream_t st = mi_pick_stream(stream, ctx->stream);
    const MergeIn in{in_scores_dev, in_ids_dev, (int)m, 0, 0, m, m, m};
    return merge_any(ctx, st, metric, nq, m, in, k, out_scores_dev, out_ids_dev);
}

// ---------------------------------------------------------------- packed per-rank lists (the all-gather's message)
// One block per rank: [nq][k] fp32 scores, then (8-byte aligned) [nq][k] int64 ids.  A rank's local search writes
// its block in place (mi_index_search_dev with pointers into it), ONE all-gather of block_bytes per rank leaves the
// blocks rank-major, and the merge reads that layout directly.
extern "C" int mi_topk_packed_layout(int64_t nq, int k, size_t* ids_offset, size_t* block_bytes)
{
    MI_REQUIRE(nq >= 0 && k >= 1 && ids_offset && block_bytes, "mi_topk_packed_layout: bad argument");
    const size_t off = ((size_t)nq * k * sizeof(float) + 255) & ~(size_t)255;
    *ids_offset = off;
    *block_bytes = (off + (size_t)nq * k * sizeof(int64_t) + 255) & ~(size_t)255;
    return MI_OK;
}

extern "C" int mi_topk_merge_packed_dev(mi_ctx* ctx, int metric, int64_t nq, int nlists, int k_in, const void* packed_dev,
                                        int k, float* out_scores_dev, int64_t* out_ids_dev, void* stream)
{
    MI_REQUIRE(ctx && nq >= 0 && nlists >= 1 && k_in >= 1 && k >= 1, "mi_topk_merge_packed_dev: bad argument");
    MI_REQUIRE(metric == MI_METRIC_IP || metric == MI_METRIC_L2, "mi_topk_merge_packed_dev: unknown metric");
    if (k > MI_MAX_K) { mi_set_error("mi_topk_merge_packed_dev: k=%d exceeds MI_MAX_K=%d", k, MI_MAX_K); return MI_ERR_UNSUPPORTED; }
    if (nq == 0) return MI_OK;
    MI_REQUIRE(packed_dev && out_scores_dev && out_ids_dev, "mi_topk_merge_packed_dev: NULL buffer");
    MI_REQUIRE(((uintptr_t)packed_dev & 7) == 0, "mi_topk_merge_packed_dev: the packed buffer must be 8-byte aligned");
    const int64_t m = (int64_t)nlists * k_in;
    MI_REQUIRE(m < ((int64_t)1 << 24), "mi_topk_merge_packed_dev: nlists*k_in = %lld is out of range", (long long)m);
    size_t ids_off = 0, block = 0;
    MI_TRY(mi_topk_packed_layout(nq, k_in, &ids_off, &block));
    DeviceGuard g(ctx->device);
    hipStream_t st = mi_pick_stream(stream, ctx->stream);
    const char* base = static_cast<const char*>(packed_dev);
    const MergeIn in{reinterpret_cast<const float*>(base), reinterpret_cast<const int64_t*>(base + ids_off), k_in,
                     (int64_t)(block / sizeof(float)), (int64_t)(block / sizeof(int64_t)), k_in, k_in, k_in};
    return merge_any(ctx, st, metric, nq, m, in, k, out_scores_dev, out_ids_dev);
}

// ---------------------------------------------------------------- communicator (SURVEY.md section 8e for a host without torch.distributed)
// RCCL is loaded with dlopen at the first mi_comm_* call, so the library has no link-time dependency on it and a
// single-GPU host never touches it; an absent librccl is a loud MI_ERR_UNSUPPORTED.  One communicator = one rank = one
// process with one GPU; the caller moves the 128-byte unique id between its processes however it likes (file, socket, MPI).
#include <dlfcn.h>
namespace {
struct RcclId { char internal[128]; };
typedef int (*fn_get_unique_id)(RcclId*);
typedef int (*fn_comm_init_rank)(void**, int, RcclId, int);
typedef int (*fn_all_gather)(const void*, void*, size_t, int, void*, hipStream_t);
typedef int (*fn_comm_destroy)(void*);
typedef const char* (*fn_error_string)(int);
struct Rccl {
    void* h = nullptr;
    fn_get_unique_id get_unique_id = nullptr;
    fn_comm_init_rank comm_init_rank = nullptr;
    fn_all_gather all_gather = nullptr;
    fn_comm_destroy comm_destroy = nullptr;
    fn_error_string error_string = nullptr;
};
Rccl g_rccl;
std::mutex g_rccl_mu;
int rccl_load()
{
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (g_rccl.h) return MI_OK;
    const char* names[] = {getenv("MI355RAG_RCCL"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* h = nullptr;
    for (const char* n : names) if (n && *n && (h = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
    if (!h) { mi_set_error("mi_comm: librccl.so not found (set MI355RAG_RCCL to its path): %s", dlerror()); return MI_ERR_UNSUPPORTED; }
    Rccl r;
    r.h = h;
    r.get_unique_id = (fn_get_unique_id)dlsym(h, "ncclGetUniqueId");
    r.comm_init_rank = (fn_comm_init_rank)dlsym(h, "ncclCommInitRank");
    r.all_gather = (fn_all_gather)dlsym(h, "ncclAllGather");
    r.comm_destroy = (fn_comm_destroy)dlsym(h, "ncclCommDestroy");
    r.error_string = (fn_error_string)dlsym(h, "ncclGetErrorString");
    if (!r.get_unique_id || !r.comm_init_rank || !r.all_gather || !r.comm_destroy) {
        mi_set_error("mi_comm: the RCCL library lacks ncclGetUniqueId / ncclCommInitRank / ncclAllGather / ncclCommDestroy");
        dlclose(h);
        return MI_ERR_UNSUPPORTED;
    }
    g_rccl = r;
    return MI_OK;
}
int rccl_fail(const char* what, int rc)
{
    mi_set_error("%s failed: %s", what, g_rccl.error_string ? g_rccl.error_string(rc) : "RCCL error");
    return MI_ERR_HIP;
}
}  // namespace

struct mi_comm {
    mi_ctx* ctx = nullptr;
    void* comm = nullptr;
    int nranks = 1, rank = 0;
    std::mutex mu;
    DevBuf mine, gathered;        // this rank's packed block, the rank-major blocks of all ranks (grow-only)
};

extern "C" int mi_comm_unique_id(void* out_id128)
{
    MI_REQUIRE(out_id128, "mi_comm_unique_id: NULL argument");
    MI_TRY(rccl_load());
    RcclId id;
    const int rc = g_rccl.get_unique_id(&id);
    if (rc != 0) return rccl_fail("ncclGetUniqueId", rc);
    memcpy(out_id128, &id, sizeof(id));
    return MI_OK;
}

extern "C" int mi_comm_create(mi_ctx* ctx, const void* id128, int nranks, int rank, mi_comm** out)
{
    MI_REQUIRE(ctx && id128 && out && nranks >= 1 && rank >= 0 && rank < nranks, "mi_comm_create: bad argument");
    MI_TRY(rccl_load());
    DeviceGuard guard(ctx->device);
    mi_comm* c = new (std::nothrow) mi_comm();
    if (!c) { mi_set_error("mi_comm_create: out of host memory"); return MI_ERR_OOM; }
    c->ctx = ctx; c->nranks = nranks; c->rank = rank;
    RcclId id;
    memcpy(&id, id128, sizeof(id));
    const int rc = g_rccl.comm_init_rank(&c->comm, nranks, id, rank);
    if (rc != 0) { delete c; return rccl_fail("ncclCommInitRank", rc); }
    *out = c;
    return MI_OK;
}

extern "C" void mi_comm_destroy(mi_comm* c)
{
    if (!c) return;
    DeviceGuard guard(c->ctx->device);
    (void)hipDeviceSynchronize();          // collectives may have been enqueued on a caller's stream, not only the ctx's
    if (c->comm) (void)g_rccl.comm_destroy(c->comm);
    c->mine.release();
    c->gathered.release();
    delete c;
}

extern "C" int mi_comm_size(const mi_comm* c) { return c ? c->nranks : -1; }
extern "C" int mi_comm_rank(const mi_comm* c) { return c ? c->rank : -1; }

// caller holds c->mu: an RCCL communicator takes one collective at a time
static int all_gather_locked(mi_comm* c, const void* send_dev, void* recv_dev, size_t bytes_per_rank, void* stream)
{
    DeviceGuard guard(c->ctx->device);
    hipStream_t st = mi_pick_stream(stream, c->ctx->stream);
    const int rc = g_rccl.all_gather(send_dev, recv_dev, bytes_per_rank, /* ncclUint8 */ 1, c->comm, st);
    if (rc != 0) return rccl_fail("ncclAllGather", rc);
    return MI_OK;
}

extern "C" int mi_comm_all_gather_dev(mi_comm* c, const void* send_dev, void* recv_dev, size_t bytes_per_rank, void* stream)
{
    MI_REQUIRE(c && send_dev && recv_dev, "mi_comm_all_gather_dev: NULL argument");
    std::lock_guard<std::mutex> lk(c->mu);
    return all_gather_locked(c, send_dev, recv_dev, bytes_per_rank, stream);
}

// The whole sharded search of one rank: local exact top-k into the rank's packed block -> ONE all-gather -> the same
// deterministic merge on every rank (rag_amd/distributed.py does the same three steps over torch.distributed).
extern "C" int mi_index_search_sharded_dev(mi_index* idx, mi_comm* c, int64_t nq, const float* q_dev, int k,
                                           float* out_scores_dev, int64_t* out_ids_dev, void* stream)
{
    MI_REQUIRE(idx && c && q_dev && out_scores_dev && out_ids_dev && nq > 0 && k > 0, "mi_index_search_sharded_dev: bad argument");
    MI_REQUIRE(idx->ctx == c->ctx, "mi_index_search_sharded_dev: the index and the communicator belong to different contexts");
    // everything that can fail for a reason the other ranks do not share is checked BEFORE the collective: a rank that
    // returns early leaves its peers blocked in ncclAllGather (then the caller must tear the communicator down)
    MI_REQUIRE(nq < ((int64_t)1 << 24), "mi_index_search_sharded_dev: nq %lld out of range", (long long)nq);
    if (k > MI_MAX_K) { mi_set_error("mi_index_search_sharded_dev: k=%d exceeds MI_MAX_K=%d", k, MI_MAX_K); return MI_ERR_UNSUPPORTED; }
    MI_REQUIRE((int64_t)c->nranks * k < ((int64_t)1 << 24), "mi_index_search_sharded_dev: nranks*k out of range");
    DeviceGuard guard(c->ctx->device);
    std::lock_guard<std::mutex> lk(c->mu);
    size_t ids_off = 0, block = 0;
    MI_TRY(mi_topk_packed_layout(nq, k, &ids_off, &block));
    MI_TRY(c->mine.ensure(block));
    MI_TRY(c->gathered.ensure(block * (size_t)c->nranks));
    char* mine = (char*)c->mine.p;
    hipStream_t st = mi_pick_stream(stream, c->ctx->stream);
    int rc = mi_index_search_dev(idx, nq, q_dev, k, (float*)mine, (int64_t*)(mine + ids_off), stream);
    if (rc != MI_OK) {
        // keep the collective matched on every rank: this rank contributes an all-padding block and still reports its error
        char msg[512];
        strncpy(msg, mi_last_error(), sizeof(msg) - 1); msg[sizeof(msg) - 1] = 0;
        (void)mi_launch_fill_pad(st, idx->metric, nq, k, (float*)mine, (int64_t*)(mine + ids_off), k);
        (void)all_gather_locked(c, mine, c->gathered.p, block, stream);
        (void)hipStreamSynchronize(st);
        mi_set_error("%s", msg);
        return rc;
    }
    MI_TRY(all_gather_locked(c, mine, c->gathered.p, block, stream));
    MI_TRY(mi_topk_merge_packed_dev(c->ctx, idx->metric, nq, c->nranks, k, c->gathered.p, k, out_scores_dev, out_ids_dev, stream));
    // `mine` / `gathered` belong to the communicator and the next call may come on another stream: done before return
    MI_HIP(hipStreamSynchronize(st));
    return MI_OK;
}
