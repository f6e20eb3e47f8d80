// kernels.h — host-side launchers of the HIP kernels (internal to libmi355rag).
#pragma once
#include "common.h"

// ---- context accessors (capi.hip owns struct mi_ctx)
int mi_ctx_device(mi_ctx* ctx);
hipStream_t mi_ctx_stream(mi_ctx* ctx);

// ---- ingest / synthetic data (ingest.hip)
// dst[r][0..d_pad) <- src[r][0..d) zero padded; src rows are src_pitch elements apart.
int mi_launch_ingest(hipStream_t st, const void* src, int src_dtype, int64_t src_pitch, int64_t n, int d, int d_pad,
                     int dst_dtype, void* dst);
int mi_launch_row_norms(hipStream_t st, const void* rows, int dtype, int64_t n, int d_pad,
                        float* xx, float* max_xx);
int mi_launch_synth(hipStream_t st, void* dst, int dtype, int64_t n, int d, int d_pad,
                    uint64_t seed, int64_t row0, int normalize);
// q_canon [nq][d_pad_c] fp32 (values rounded to bf16 first when round_canon), qq = canonical
// dot(q,q) of those values; optional q_bf16 [nq][d_pad_s] = the bf16-rounded queries.
int mi_launch_prep_queries(hipStream_t st, const float* q, int64_t nq, int d, int d_pad_c, int round_canon,
                           float* q_canon, int d_pad_s, uint16_t* q_bf16, float* qq);
int mi_launch_export_rows(hipStream_t st, const void* rows, int dtype, int64_t n, int d, int d_pad,
                          float* out);

// ---- exact path (search_exact.hip)
// S[qi][r] = canonical dot(q_canon[qi], row[row0 + r]), qi < nq, r < nrows.
int mi_launch_scan_dense(hipStream_t st, const void* rows, int dtype, int64_t row0, int nrows,
                         int d_pad, const float* q_canon, int nq, float* S, int64_t s_stride);
// Per query: exact top-k of one dense segment, best-first, written at out[q*out_stride + j].
// xx points at the segment's first row; id0 = id of the segment's first row.
// seg_len < nrows: the segment is selected in pieces of seg_len rows, grid (nq, pieces); piece p's k best
// land at out[q * out_stride + p * k ..] and the caller merges them (mi_select_dense_pieces picks seg_len).
int mi_launch_select_dense(hipStream_t st, const float* S, int64_t s_stride, int nrows, int seg_len,
                           const float* xx, const float* qq, int metric, int nq, int k,
                           int64_t id0, float* out_s, int64_t* out_i, int64_t out_stride);
int mi_select_dense_pieces(int nrows, int nq, int k, int* seg_len);
// Per query: merge m (<= 4096) (score,id) entries (id < 0 = padding) into the k best.
int mi_launch_merge(hipStream_t st, int metric, int64_t nq, int m, const float* in_s,
                    const int64_t* in_i, int64_t in_stride, int k, float* out_s, int64_t* out_i,
                    int64_t out_stride);
// The same per chunk of `chunk` (<= 4096) consecutive entries: grid (nq, ceil(m / chunk)); chunk c's k best land
// at out[q * out_stride + c * k ..] (padded with id -1 where the chunk holds fewer).
int mi_launch_merge_chunks(hipStream_t st, int metric, int64_t nq, int m, int chunk, const float* in_s,
                           const int64_t* in_i, int64_t in_stride, int k, float* out_s, int64_t* out_i,
                           int64_t out_stride);
// General form: entry i of query q is element i % k_in of list i / k_in, stored at l * lstride + q * qstride + j
// (see merge_kernel), chunked as above.
int mi_launch_merge_ex(hipStream_t st, int metric, int64_t nq, int m, int chunk, const float* in_s, const int64_t* in_i,
                       int k_in, int64_t lstride_s, int64_t lstride_i, int64_t qstride,
                       int k, float* out_s, int64_t* out_i, int64_t out_stride);
int mi_launch_fill_pad(hipStream_t st, int metric, int64_t nq, int k, float* s, int64_t* i,
                       int64_t stride);
// out[dst_q[j]] rows <- in[j] rows (scatter of re-run queries); dst_q == nullptr: identity.
int mi_launch_copy_lists(hipStream_t st, int64_t nq, int k, const float* in_s, const int64_t* in_i,
                         int64_t in_stride, const int32_t* dst_q, float* out_s, int64_t* out_i,
                         int64_t out_stride);
// Gather rows of a [n][d_pad] fp32 matrix / scalars by index list.
int mi_launch_gather_queries(hipStream_t st, const int32_t* sel, int nsel, int d_pad,
                             const float* q_canon, const float* qq, float* q_out, float* qq_out);

// ---- MFMA path (search_mfma.hip)
struct mi_mfma_plan {
    int kprime;        // candidates kept per (slice, query)
    int cap;           // list capacity per (slice, query)
    int nslices;       // row slices
    int ngroups;       // slice groups of the finish kernel (each <= spg slices)
    int spg;           // slices per group
    int m16;           // 1: score with 16x16x32 MFMAs (MFMA-bound variant only)
    int pd;            // LDS-DMA prefetch distance of the scan: 3 (HBM-bound) or 2 (MFMA-bound)
    int64_t slice_rows;// rows per slice (multiple of the tile height)
    int nq_pad;        // queries padded to the query tile
    size_t list_bytes, cnt_bytes, thr_bytes;
    // threshold seed pass (0 rows: none): the scan kernel in SEED mode over the first seed_rows rows, seed_slices slices
    int64_t seed_rows;
    int seed_slices;
    size_t gthr_bytes; // the shared-threshold block: [nq_pad + 64 nslices] of the scan, then [nq_pad + 64 seed_slices] + [nq_pad][64] of the seed pass
};
bool mi_mfma_supported(int d_pad, int k);
// Row pitch (elements) the scan streams for dimension d: 128 / 384 / 768, or 0 (unsupported).
int  mi_mfma_pitch(int d);
int  mi_mfma_make_plan(int64_t n, int d_pad, int64_t nq, int k, mi_mfma_plan* plan);
// Relative error terms of the phase-1 score (see mf_err_bound in search_mfma.hip).
void mi_mfma_eps(int d_pad_scan, bool rounded_shadow, float* eps_acc, float* eps_round);
// Phase 1: bf16 MFMA scan + threshold filter -> per-(slice,query) candidate lists.
// rows: bf16 [n][d_pad]; xx: canonical squared norms (L2 only).
int  mi_launch_mfma_scan(hipStream_t st, const mi_mfma_plan* p, int metric, const uint16_t* rows,
                         const float* xx, int64_t n, int d_pad, const uint16_t* q_bf16, int64_t nq,
                         uint64_t* lists, int32_t* cnts, float* thrs, uint32_t* gthr /* plan->gthr_bytes, zeroed */);
// Threshold seed (plan->seed_rows > 0; enqueue BEFORE mi_launch_mfma_scan on the same stream): scores the first seed_rows
// rows against every query with the scan kernel's own arithmetic, keeps the maximum of each of 64 disjoint row classes per
// query, and writes the 16th largest of the 64 (less one ulp) into gthr[q]: sixteen different rows score above it, so it is
// a lower bound of the query's 16th best score and every workgroup of the scan starts with it instead of -inf.
int  mi_launch_mfma_seed(hipStream_t st, const mi_mfma_plan* p, int metric, const uint16_t* rows, const float* xx,
                         int d_pad, const uint16_t* q_bf16, int64_t nq, uint32_t* gthr);
// Phase 2: per (query, slice group): prefilter by MFMA score, canonical re-score from the
// canonical rows (canon_dtype, pitch d_pad), exact top-k at out[q * out_stride + g * k + j].
// over_flags[q] = 1 where a group had more near-ties than it can sort.
int  mi_launch_mfma_finish(hipStream_t st, const mi_mfma_plan* p, int metric, const void* canon_rows, int canon_dtype,
                           const float* xx, const float* max_xx, int d_pad, float eps_acc, float eps_round,
                           const float* q_canon, const float* qq, int64_t nq, int k, int64_t id_base,
                           const uint64_t* lists, const int32_t* cnts,
                           float* out_s, int64_t* out_i, int64_t out_stride, int32_t* over_flags);
// Phase 3: certificate on the final list (stride k); failing queries are appended to fail_list.
int  mi_launch_mfma_certify(hipStream_t st, const mi_mfma_plan* p, int metric, int64_t nq, int k, float eps_acc, float eps_round,
                            const float* qq, const float* max_xx, const int32_t* cnts, const float* thrs,
                            const float* out_s, const int64_t* out_i, const int32_t* over_flags,
                            int32_t* fail_list, int32_t* fail_count);
