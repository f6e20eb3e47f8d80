// multi.hip — mi_multi_index_*: ONE flat index over several GPUs of a node, driven from ONE host process.
//
// advanced_rag.py is a single Python process with a module-global chain (:336-337); FAISS.from_documents (:295)
// hands it one store and as_retriever(...).invoke (:296, :409) searches it from whatever thread LangChain picks.
// A drop-in that wants the whole node from that process cannot be one-process-per-rank (that is mi_comm_* /
// rag_amd.distributed, for hosts that already run a rank per GPU).  Here the shards live behind one handle:
//
//   rows      global row g belongs to block b = g / MI_MULTI_BLOCK; block b lives on shard b % G at local block b / G
//             ("shards rows round-robin-by-block"): appends keep working, every shard stays within
//             one block of N / G rows, and local row order == global row order on every shard, so the per-shard
//             tie-break (lower id first) is the global one.
//   search    queries reach every shard (peer copy from the caller's device, or straight from the host buffer), one
//             host thread per shard runs mi_index_search_dev into the shard's packed block, a kernel rewrites local
//             row numbers as global ids, the block is copied peer-to-peer into slot s of the gathered buffer on the
//             first device, and mi_topk_merge_packed_dev merges there: the packed layout and the merge are the ones
//             the all-gather path uses, so the answer is the single-index answer bit for bit.  No RCCL: inside one
//             process the exchange step is G small peer copies over xGMI (Q*k*12 bytes each).
//
// The same device may be listed more than once (a peer copy to oneself is an ordinary device copy): that is how the
// path is tested on a one-GPU box.
#include <string.h>
#include <mutex>
#include <new>
#include <thread>
#include <vector>
#include "kernels.h"

int mi_index_add_synthetic_at(mi_index* idx, int64_t n, uint64_t seed, int normalize, int64_t global_row0);   // capi.hip

namespace {

struct DBuf {
    void* p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes)
    {
        if (bytes <= cap) return MI_OK;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        const size_t want = (bytes + 255) / 256 * 256;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) { p = nullptr; mi_set_error("mi_multi_index: hipMalloc(%zu) failed: %s", want, hipGetErrorString(e)); return MI_ERR_OOM; }
        cap = want;
        return MI_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

struct Shard {
    int device = 0;
    mi_ctx* ctx = nullptr;
    mi_index* idx = nullptr;
    DBuf q, block;               // this shard's copy of the queries, its packed result block
    hipEvent_t ev_q = nullptr;   // recorded on the caller's stream behind the queries' producer
    int rc = MI_OK;
    char err[512] = "";
};

// local row number on shard s -> global id; padding (-1) stays
__global__ void multi_remap_kernel(int64_t* ids, int64_t n, int shard, int nshards, int64_t block_rows)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t l = ids[i];
    if (l < 0) return;
    const int64_t lb = l / block_rows;
    ids[i] = (lb * nshards + shard) * block_rows + (l - lb * block_rows);
}

}  // namespace

struct mi_multi_index {
    std::vector<Shard> sh;
    int d = 0, metric = 0, dtype = 0;
    int64_t n = 0;               // global rows
    int64_t block_rows = MI_MULTI_BLOCK;
    std::mutex mu;
    DBuf gathered, q0, out;      // on shard 0's device: the G blocks, host-path queries, host-path result
    int64_t st_queries = 0;
};

static int shard_of(const mi_multi_index* m, int64_t g) { return (int)((g / m->block_rows) % (int64_t)m->sh.size()); }

extern "C" int mi_multi_index_create(const int* device_ids, int n_devices, int dim, int metric, int dtype, mi_multi_index** out)
{
    MI_REQUIRE(device_ids && out && n_devices >= 1 && n_devices <= 64, "mi_multi_index_create: bad argument (1..64 devices)");
    *out = nullptr;
    mi_multi_index* m = new (std::nothrow) mi_multi_index();
    if (!m) { mi_set_error("mi_multi_index_create: out of host memory"); return MI_ERR_OOM; }
    m->d = dim; m->metric = metric; m->dtype = dtype;
    m->sh.resize(n_devices);
    int rc = MI_OK;
    for (int s = 0; s < n_devices && rc == MI_OK; ++s) {
        Shard& S = m->sh[s];
        S.device = device_ids[s];
        rc = mi_init(S.device, &S.ctx);
        if (rc == MI_OK) rc = mi_index_create(S.ctx, dim, metric, dtype, &S.idx);
        if (rc == MI_OK) {
            DeviceGuard g(S.device);
            if (hipEventCreateWithFlags(&S.ev_q, hipEventDisableTiming) != hipSuccess) { mi_set_error("mi_multi_index_create: hipEventCreate failed"); rc = MI_ERR_HIP; }
        }
    }
    // direct xGMI peer copies where the topology allows; hipMemcpyPeerAsync works (staged) without it
    for (int a = 0; a < n_devices && rc == MI_OK; ++a)
        for (int b = 0; b < n_devices; ++b) {
            const int da = m->sh[a].device, db = m->sh[b].device;
            if (da == db) continue;
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, da, db) == hipSuccess && can) {
                DeviceGuard g(da);
                hipError_t e = hipDeviceEnablePeerAccess(db, 0);
                if (e == hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();
            }
        }
    if (rc != MI_OK) { mi_multi_index_destroy(m); return rc; }
    *out = m;
    return MI_OK;
}

extern "C" void mi_multi_index_destroy(mi_multi_index* m)
{
    if (!m) return;
    for (Shard& S : m->sh) {
        if (S.ctx) {
            DeviceGuard g(S.device);
            (void)hipDeviceSynchronize();
            S.q.release(); S.block.release();
            if (S.ev_q) (void)hipEventDestroy(S.ev_q);
        }
        if (S.idx) mi_index_destroy(S.idx);
    }
    if (!m->sh.empty() && m->sh[0].ctx) {
        DeviceGuard g(m->sh[0].device);
        m->gathered.release(); m->q0.release(); m->out.release();
    }
    for (Shard& S : m->sh) if (S.ctx) mi_ctx_destroy(S.ctx);
    delete m;
}

extern "C" int mi_multi_index_devices(const mi_multi_index* m) { return m ? (int)m->sh.size() : -1; }
extern "C" int64_t mi_multi_index_ntotal(const mi_multi_index* m) { return m ? m->n : -1; }
extern "C" int64_t mi_multi_index_shard_ntotal(const mi_multi_index* m, int shard)
{
    if (!m || shard < 0 || shard >= (int)m->sh.size()) return -1;
    return mi_index_ntotal(m->sh[shard].idx);
}

extern "C" int mi_multi_index_set_block_rows(mi_multi_index* m, int64_t rows)
{
    MI_REQUIRE(m && rows >= 1 && rows <= ((int64_t)1 << 30), "mi_multi_index_set_block_rows: bad argument");
    std::lock_guard<std::mutex> lk(m->mu);
    MI_REQUIRE(m->n == 0, "mi_multi_index_set_block_rows: the index already holds rows");
    m->block_rows = rows;
    return MI_OK;
}

extern "C" int mi_multi_index_set_mode(mi_multi_index* m, int mode)
{
    MI_REQUIRE(m, "mi_multi_index_set_mode: NULL argument");
    std::lock_guard<std::mutex> lk(m->mu);
    for (Shard& S : m->sh) MI_TRY(mi_index_set_mode(S.idx, mode));
    return MI_OK;
}

extern "C" int mi_multi_index_reserve(mi_multi_index* m, int64_t n_rows)
{
    MI_REQUIRE(m && n_rows >= 0, "mi_multi_index_reserve: bad argument");
    std::lock_guard<std::mutex> lk(m->mu);
    const int64_t G = (int64_t)m->sh.size();
    const int64_t nb = (n_rows + m->block_rows - 1) / m->block_rows;
    int64_t per = (nb + G - 1) / G * m->block_rows;          // whole blocks per shard, never more than everything
    if (per > n_rows) per = n_rows;
    for (Shard& S : m->sh) MI_TRY(mi_index_reserve(S.idx, per));
    return MI_OK;
}

// Rows [m->n, m->n + n) in global order, cut at block boundaries; piece by piece to the shard that owns the block.
template <typename F> static int for_pieces(const mi_multi_index* m, int64_t g0, int64_t n, F&& f)
{
    for (int64_t g = g0; g < g0 + n;) {
        const int64_t end_of_block = (g / m->block_rows + 1) * m->block_rows;
        const int64_t len = (g0 + n < end_of_block ? g0 + n : end_of_block) - g;
        MI_TRY(f(shard_of(m, g), g, len));
        g += len;
    }
    return MI_OK;
}
// local row of global row g on its shard
static int64_t local_row(const mi_multi_index* m, int64_t g)
{
    const int64_t b = g / m->block_rows;
    return (b / (int64_t)m->sh.size()) * m->block_rows + (g - b * m->block_rows);
}

extern "C" int mi_multi_index_add_raw(mi_multi_index* m, int64_t n, const void* x_host, int src_dtype)
{
    MI_REQUIRE(m && n >= 0 && (x_host || n == 0), "mi_multi_index_add: bad argument");
    MI_REQUIRE(src_dtype == MI_DTYPE_F32 || src_dtype == MI_DTYPE_BF16, "mi_multi_index_add: unknown src dtype");
    if (n == 0) return MI_OK;
    std::lock_guard<std::mutex> lk(m->mu);
    const size_t row_bytes = (size_t)m->d * (src_dtype == MI_DTYPE_F32 ? 4 : 2);
    const int64_t g0 = m->n;
    // Shards receive their pieces concurrently (one host thread per shard: each has its own PCIe link); within a shard the
    // pieces arrive in global order, which is its local order.
    const int G = (int)m->sh.size();
    std::vector<std::thread> th;
    for (int s = 0; s < G; ++s) {
        m->sh[s].rc = MI_OK;
        th.emplace_back([m, s, g0, n, x_host, row_bytes, src_dtype]() {
            Shard& S = m->sh[s];
            S.rc = for_pieces(m, g0, n, [&](int owner, int64_t g, int64_t len) -> int {
                if (owner != s) return MI_OK;
                return mi_index_add_raw(S.idx, len, static_cast<const char*>(x_host) + (size_t)(g - g0) * row_bytes, src_dtype);
            });
            if (S.rc != MI_OK) { strncpy(S.err, mi_last_error(), sizeof(S.err) - 1); S.err[sizeof(S.err) - 1] = 0; }
        });
    }
    for (auto& t : th) t.join();
    for (Shard& S : m->sh) if (S.rc != MI_OK) { mi_set_error("%s", S.err); return S.rc; }     // rows of a failed add are not counted
    m->n += n;
    return MI_OK;
}

extern "C" int mi_multi_index_add(mi_multi_index* m, int64_t n, const float* x_host)
{
    return mi_multi_index_add_raw(m, n, x_host, MI_DTYPE_F32);
}

extern "C" int mi_multi_index_add_synthetic(mi_multi_index* m, int64_t n, uint64_t seed, int normalize)
{
    MI_REQUIRE(m && n >= 0, "mi_multi_index_add_synthetic: bad argument");
    if (n == 0) return MI_OK;
    std::lock_guard<std::mutex> lk(m->mu);
    const int64_t g0 = m->n;
    MI_TRY(for_pieces(m, g0, n, [&](int owner, int64_t g, int64_t len) -> int {
        return mi_index_add_synthetic_at(m->sh[owner].idx, len, seed, normalize, g);
    }));
    m->n += n;
    return MI_OK;
}

extern "C" int mi_multi_index_reset(mi_multi_index* m)
{
    MI_REQUIRE(m, "mi_multi_index_reset: NULL argument");
    std::lock_guard<std::mutex> lk(m->mu);
    for (Shard& S : m->sh) MI_TRY(mi_index_reset(S.idx));
    m->n = 0;
    return MI_OK;
}

// raw != 0: rows in the storage type (fp32 / bf16 bit patterns); else widened to fp32
extern "C" int mi_multi_index_get_rows(mi_multi_index* m, int64_t row0, int64_t n, void* out_host, int raw)
{
    MI_REQUIRE(m && out_host && row0 >= 0 && n >= 0, "mi_multi_index_get_rows: bad argument");
    std::lock_guard<std::mutex> lk(m->mu);
    MI_REQUIRE(row0 + n <= m->n, "mi_multi_index_get_rows: range [%lld,%lld) outside [0,%lld)", (long long)row0, (long long)(row0 + n), (long long)m->n);
    const size_t row_bytes = (size_t)m->d * ((raw && m->dtype == MI_DTYPE_BF16) ? 2 : 4);
    return for_pieces(m, row0, n, [&](int owner, int64_t g, int64_t len) -> int {
        char* dst = static_cast<char*>(out_host) + (size_t)(g - row0) * row_bytes;
        return raw ? mi_index_get_rows_raw(m->sh[owner].idx, local_row(m, g), len, dst)
                   : mi_index_get_rows(m->sh[owner].idx, local_row(m, g), len, reinterpret_cast<float*>(dst));
    });
}

// out[0] queries searched, out[1] (query, shard) searches certified by the MFMA path, out[2] re-run on the exact path,
// out[3] global rows
extern "C" int mi_multi_index_stats(const mi_multi_index* m, int64_t out[4])
{
    MI_REQUIRE(m && out, "mi_multi_index_stats: NULL argument");
    out[0] = m->st_queries; out[1] = 0; out[2] = 0; out[3] = m->n;
    for (const Shard& S : m->sh) {
        int64_t t[4];
        MI_TRY(mi_index_stats(S.idx, t));
        out[1] += t[1]; out[2] += t[2];
    }
    return MI_OK;
}

// q: device pointer on shard 0's device (q_on_host == 0) or a host pointer; results on shard 0's device.
static int multi_search_locked(mi_multi_index* m, int64_t nq, const float* q, bool q_on_host, int k,
                               float* out_s_dev, int64_t* out_i_dev, hipStream_t st0)
{
    const int G = (int)m->sh.size();
    const int d0 = m->sh[0].device;
    size_t ids_off = 0, block = 0;
    MI_TRY(mi_topk_packed_layout(nq, k, &ids_off, &block));
    const size_t q_bytes = (size_t)nq * m->d * sizeof(float);
    {
        DeviceGuard g(d0);
        MI_TRY(m->gathered.ensure(block * (size_t)G));
        // the queries' producer sits on the caller's stream: every shard's stream waits for this event
        if (!q_on_host) for (Shard& S : m->sh) MI_HIP(hipEventRecord(S.ev_q, st0));
    }
    char* gathered = static_cast<char*>(m->gathered.p);
    std::vector<std::thread> th;
    for (int s = 0; s < G; ++s) {
        m->sh[s].rc = MI_OK;
        th.emplace_back([=]() {
            Shard& S = m->sh[s];
            auto body = [&]() -> int {
                DeviceGuard g(S.device);
                hipStream_t st = mi_ctx_stream(S.ctx);
                MI_TRY(S.block.ensure(block));
                const float* qs;
                if (q_on_host) {
                    MI_TRY(S.q.ensure(q_bytes));
                    MI_HIP(hipMemcpyAsync(S.q.p, q, q_bytes, hipMemcpyHostToDevice, st));
                    qs = static_cast<const float*>(S.q.p);
                } else {
                    MI_HIP(hipStreamWaitEvent(st, S.ev_q, 0));
                    if (S.device == d0) {
                        qs = q;                                               // same device: read in place
                    } else {
                        MI_TRY(S.q.ensure(q_bytes));
                        MI_HIP(hipMemcpyPeerAsync(S.q.p, S.device, q, d0, q_bytes, st));
                        qs = static_cast<const float*>(S.q.p);
                    }
                }
                char* mine = static_cast<char*>(S.block.p);
                MI_TRY(mi_index_search_dev(S.idx, nq, qs, k, reinterpret_cast<float*>(mine), reinterpret_cast<int64_t*>(mine + ids_off), st));
                const int64_t cnt = nq * k;
                multi_remap_kernel<<<dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, st>>>(reinterpret_cast<int64_t*>(mine + ids_off), cnt, s, G, m->block_rows);
                MI_HIP(hipGetLastError());
                if (S.device == d0) MI_HIP(hipMemcpyAsync(gathered + (size_t)s * block, mine, block, hipMemcpyDeviceToDevice, st));
                else MI_HIP(hipMemcpyPeerAsync(gathered + (size_t)s * block, d0, mine, S.device, block, st));
                MI_HIP(hipStreamSynchronize(st));
                return MI_OK;
            };
            S.rc = body();
            if (S.rc != MI_OK) { strncpy(S.err, mi_last_error(), sizeof(S.err) - 1); S.err[sizeof(S.err) - 1] = 0; }
        });
    }
    for (auto& t : th) t.join();
    for (Shard& S : m->sh) if (S.rc != MI_OK) { mi_set_error("%s", S.err); return S.rc; }
    m->st_queries += nq;
    // every block is complete in `gathered` (each shard synchronised its stream): merge on the caller's stream
    return mi_topk_merge_packed_dev(m->sh[0].ctx, m->metric, nq, G, k, gathered, k, out_s_dev, out_i_dev,
                                    st0 == nullptr ? MI_STREAM_LEGACY : (void*)st0);
}

extern "C" int mi_multi_index_search_dev(mi_multi_index* m, int64_t nq, const float* q_dev, int k,
                                         float* out_scores_dev, int64_t* out_ids_dev, void* stream)
{
    MI_REQUIRE(m && nq >= 0 && k >= 1 && (nq == 0 || (q_dev && out_scores_dev && out_ids_dev)), "mi_multi_index_search_dev: bad argument");
    if (k > MI_MAX_K) { mi_set_error("search: k=%d exceeds MI_MAX_K=%d", k, MI_MAX_K); return MI_ERR_UNSUPPORTED; }
    if (nq == 0) return MI_OK;
    std::lock_guard<std::mutex> lk(m->mu);
    DeviceGuard g(m->sh[0].device);
    hipStream_t st0 = mi_pick_stream(stream, mi_ctx_stream(m->sh[0].ctx));
    MI_TRY(multi_search_locked(m, nq, q_dev, false, k, out_scores_dev, out_ids_dev, st0));
    MI_HIP(hipStreamSynchronize(st0));       // as mi_index_search_dev: the answer is complete on return
    return MI_OK;
}

extern "C" int mi_multi_index_search(mi_multi_index* m, int64_t nq, const float* q_host, int k, float* out_scores, int64_t* out_ids)
{
    MI_REQUIRE(m && nq >= 0 && k >= 1 && (nq == 0 || (q_host && out_scores && out_ids)), "mi_multi_index_search: bad argument");
    if (k > MI_MAX_K) { mi_set_error("search: k=%d exceeds MI_MAX_K=%d", k, MI_MAX_K); return MI_ERR_UNSUPPORTED; }
    if (nq == 0) return MI_OK;
    std::lock_guard<std::mutex> lk(m->mu);
    DeviceGuard g(m->sh[0].device);
    hipStream_t st0 = mi_ctx_stream(m->sh[0].ctx);
    const size_t s_bytes = (size_t)nq * k * sizeof(float), i_bytes = (size_t)nq * k * sizeof(int64_t);
    const size_t i_off = (s_bytes + 255) & ~(size_t)255;
    MI_TRY(m->out.ensure(i_off + i_bytes));
    float* ds = reinterpret_cast<float*>(m->out.p);
    int64_t* di = reinterpret_cast<int64_t*>(static_cast<char*>(m->out.p) + i_off);
    // every shard copies the queries from the host buffer itself: G PCIe links in parallel, no hop through device 0
    MI_TRY(multi_search_locked(m, nq, q_host, true, k, ds, di, st0));
    MI_HIP(hipMemcpyAsync(out_scores, ds, s_bytes, hipMemcpyDeviceToHost, st0));
    MI_HIP(hipMemcpyAsync(out_ids, di, i_bytes, hipMemcpyDeviceToHost, st0));
    MI_HIP(hipStreamSynchronize(st0));
    return MI_OK;
}
