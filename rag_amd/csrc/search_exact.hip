// search_exact.hip — the canonical-order exact scan (IndexFlat::search, advanced_rag.py:296):
//   scan_dense  : S[q][r] = fmaf chain over k of q[k]*x_r[k]   (fp32 VALU, bit-exact definition)
//   select_dense: per query, exact top-k of a dense segment by MSB radix select on the
//                 64-bit key (rank word << 32 | ~row), then an LDS bitonic sort of the k winners
//   merge       : per query, best k of up to 4096 (score,id) entries (segments, shards)
// This is the ground-truth path: the MFMA scan (search_mfma.hip) falls back to it whenever
// its certificate does not hold, and an F32 index uses it directly.
#include <stdlib.h>
#include "kernels.h"

// ---------------------------------------------------------------- scan_dense
// Workgroup = 256 threads = 256 rows x 16 queries.  Each k-chunk of 64 elements of the row
// tile is fetched as whole 256-B (fp32) / 128-B (bf16) row segments, converted to fp32 and
// parked in LDS with a 272-B row stride (conflict-free ds_read_b128 per row); the 16 query
// chunks are broadcast reads.  Thread t owns row t: 16 independent fmaf chains, k ascending.
// blockIdx.x walks the query groups so workgroups that share a row tile are dispatched
// together and the tile is served from L2 after the first fetch.
#define SD_ROWS 256
#define SD_Q 16
#define SD_KC 64

template <typename T>
__global__ __launch_bounds__(256) void scan_dense_kernel(const T* __restrict__ rows, int64_t row0, int nrows,
                                                         int d_pad, const float* __restrict__ q_canon, int nq,
                                                         float* __restrict__ S, int64_t s_stride)
{
    __shared__ float xs[SD_ROWS][SD_KC + 4];
    __shared__ float qs[SD_Q][SD_KC];
    const int t = threadIdx.x;
    const int q0 = blockIdx.x * SD_Q;
    const int r0 = blockIdx.y * SD_ROWS;
    const T* base = rows + (row0 + r0) * (int64_t)d_pad;

    float acc[SD_Q];
#pragma unroll
    for (int j = 0; j < SD_Q; ++j) acc[j] = 0.0f;

    for (int k0 = 0; k0 < d_pad; k0 += SD_KC) {
        if constexpr (sizeof(T) == 4) {
#pragma unroll 4
            for (int p = 0; p < 16; ++p) {
                const int r = p * 16 + (t >> 4), c = (t & 15) * 4;
                float4 v = make_float4(0, 0, 0, 0);
                if (r0 + r < nrows) v = *reinterpret_cast<const float4*>(base + (int64_t)r * d_pad + k0 + c);
                *reinterpret_cast<float4*>(&xs[r][c]) = v;
            }
        } else {
#pragma unroll 4
            for (int p = 0; p < 8; ++p) {
                const int r = p * 32 + (t >> 3), c = (t & 7) * 8;
                uint4 v = make_uint4(0, 0, 0, 0);
                if (r0 + r < nrows) v = *reinterpret_cast<const uint4*>(base + (int64_t)r * d_pad + k0 + c);
                float4 lo, hi;
                lo.x = __builtin_bit_cast(float, v.x << 16); lo.y = __builtin_bit_cast(float, v.x & 0xFFFF0000u);
                lo.z = __builtin_bit_cast(float, v.y << 16); lo.w = __builtin_bit_cast(float, v.y & 0xFFFF0000u);
                hi.x = __builtin_bit_cast(float, v.z << 16); hi.y = __builtin_bit_cast(float, v.z & 0xFFFF0000u);
                hi.z = __builtin_bit_cast(float, v.w << 16); hi.w = __builtin_bit_cast(float, v.w & 0xFFFF0000u);
                *reinterpret_cast<float4*>(&xs[r][c]) = lo;
                *reinterpret_cast<float4*>(&xs[r][c + 4]) = hi;
            }
        }
        {
            const int j = t >> 4, c = (t & 15) * 4;   // 16 queries x 64 floats
            float4 v = make_float4(0, 0, 0, 0);
            if (q0 + j < nq) v = *reinterpret_cast<const float4*>(q_canon + (int64_t)(q0 + j) * d_pad + k0 + c);
            *reinterpret_cast<float4*>(&qs[j][c]) = v;
        }
        __syncthreads();
#pragma unroll 4
        for (int kk = 0; kk < SD_KC; kk += 4) {
            const float4 xv = *reinterpret_cast<const float4*>(&xs[t][kk]);
#pragma unroll
            for (int j = 0; j < SD_Q; ++j) {
                const float4 qv = *reinterpret_cast<const float4*>(&qs[j][kk]);
                float a = acc[j];
                a = __builtin_fmaf(qv.x, xv.x, a);
                a = __builtin_fmaf(qv.y, xv.y, a);
                a = __builtin_fmaf(qv.z, xv.z, a);
                a = __builtin_fmaf(qv.w, xv.w, a);
                acc[j] = a;
            }
        }
        __syncthreads();
    }
    if (r0 + t < nrows) {
#pragma unroll
        for (int j = 0; j < SD_Q; ++j)
            if (q0 + j < nq) S[(int64_t)(q0 + j) * s_stride + r0 + t] = acc[j];
    }
}

// ---------------------------------------------------------------- scan_dense, a handful of queries
// The same chains (thread = row, k ascending, one fmaf per element) for nq <= 4: one wave and 64
// rows per workgroup, so a few thousand rows already spread over the chip, no idle query slots,
// and the next k-chunk's row segments are in flight (registers) while this one is consumed.
// 2,000 x 384 fp32 rows, one query: 54 us with the 256 x 16 kernel, a few us with this one.
template <typename T, int Q>
__global__ __launch_bounds__(64) void scan_dense_small_kernel(const T* __restrict__ rows, int64_t row0, int nrows,
                                                              int d_pad, const float* __restrict__ q_canon, int nq,
                                                              float* __restrict__ S, int64_t s_stride)
{
    constexpr int NL = (sizeof(T) == 4) ? 16 : 8;      // 16-byte loads per lane per chunk
    constexpr int RPL = 64 / NL;                        // rows one load instruction covers
    constexpr int LPR = 64 / RPL;                       // lanes per row segment
    __shared__ __attribute__((aligned(16))) float xs[64][SD_KC + 4];
    __shared__ __attribute__((aligned(16))) float qs[Q][SD_KC];
    const int t = threadIdx.x;
    const int q0 = blockIdx.x * Q;
    const int r0 = blockIdx.y * 64;
    const T* base = rows + (row0 + r0) * (int64_t)d_pad;
    const int lr = t / LPR, lc = t % LPR;
    const int qj = t >> 4, qc = (t & 15) * 4;

    uint4 pre[NL];
    float4 qpre;
    auto fetch = [&](int k0) {
#pragma unroll
        for (int p = 0; p < NL; ++p) {
            const int r = p * RPL + lr;
            pre[p] = make_uint4(0, 0, 0, 0);
            if (r0 + r < nrows) pre[p] = *reinterpret_cast<const uint4*>(base + (int64_t)r * d_pad + k0 + lc * (16 / (int)sizeof(T)));
        }
        qpre = make_float4(0, 0, 0, 0);
        if (qj < Q && q0 + qj < nq) qpre = *reinterpret_cast<const float4*>(q_canon + (int64_t)(q0 + qj) * d_pad + k0 + qc);
    };

    float acc[Q];
#pragma unroll
    for (int j = 0; j < Q; ++j) acc[j] = 0.0f;

    fetch(0);
    for (int k0 = 0; k0 < d_pad; k0 += SD_KC) {
#pragma unroll
        for (int p = 0; p < NL; ++p) {
            const int r = p * RPL + lr;
            const uint4 v = pre[p];
            if constexpr (sizeof(T) == 4) {
                *reinterpret_cast<uint4*>(&xs[r][lc * 4]) = v;
            } else {
                float4 lo, hi;
                lo.x = __builtin_bit_cast(float, v.x << 16); lo.y = __builtin_bit_cast(float, v.x & 0xFFFF0000u);
                lo.z = __builtin_bit_cast(float, v.y << 16); lo.w = __builtin_bit_cast(float, v.y & 0xFFFF0000u);
                hi.x = __builtin_bit_cast(float, v.z << 16); hi.y = __builtin_bit_cast(float, v.z & 0xFFFF0000u);
                hi.z = __builtin_bit_cast(float, v.w << 16); hi.w = __builtin_bit_cast(float, v.w & 0xFFFF0000u);
                *reinterpret_cast<float4*>(&xs[r][lc * 8]) = lo;
                *reinterpret_cast<float4*>(&xs[r][lc * 8 + 4]) = hi;
            }
        }
        if (qj < Q) *reinterpret_cast<float4*>(&qs[qj][qc]) = qpre;
        __syncthreads();
        if (k0 + SD_KC < d_pad) fetch(k0 + SD_KC);
#pragma unroll 4
        for (int kk = 0; kk < SD_KC; kk += 4) {
            const float4 xv = *reinterpret_cast<const float4*>(&xs[t][kk]);
#pragma unroll
            for (int j = 0; j < Q; ++j) {
                const float4 qv = *reinterpret_cast<const float4*>(&qs[j][kk]);
                float a = acc[j];
                a = __builtin_fmaf(qv.x, xv.x, a);
                a = __builtin_fmaf(qv.y, xv.y, a);
                a = __builtin_fmaf(qv.z, xv.z, a);
                a = __builtin_fmaf(qv.w, xv.w, a);
                acc[j] = a;
            }
        }
        __syncthreads();
    }
    if (r0 + t < nrows) {
#pragma unroll
        for (int j = 0; j < Q; ++j)
            if (q0 + j < nq) S[(int64_t)(q0 + j) * s_stride + r0 + t] = acc[j];
    }
}

template <typename T>
static void launch_scan_dense_small(hipStream_t st, const T* rows, int64_t row0, int nrows, int d_pad, const float* q_canon,
                                    int nq, float* S, int64_t s_stride)
{
    const dim3 grid(1, (nrows + 63) / 64);
    if (nq == 1) scan_dense_small_kernel<T, 1><<<grid, 64, 0, st>>>(rows, row0, nrows, d_pad, q_canon, nq, S, s_stride);
    else scan_dense_small_kernel<T, 4><<<grid, 64, 0, st>>>(rows, row0, nrows, d_pad, q_canon, nq, S, s_stride);
}

// ---------------------------------------------------------------- scan_dense on the fp32 matrix pipe
// The same numbers from v_mfma_f32_32x32x2_f32: its result is bit for bit the k-ordered fp32 fmaf chain
// D = fma(a_k1, b_k1, fma(a_k0, b_k0, C)) -- one rounding per product, no wider accumulation, denormals kept
// (cdna_hip_programming.md, "FP32-input MFMA") -- so chaining the instruction over k = 0, 2, 4, ... from C = +0
// IS the canonical score, at the matrix pipe's 64 FLOP/clk/SIMD with one operand register per 2,048 FMAs instead
// of the VALU kernel's LDS read per 4.  Held to bit equality with the oracle by tests/test_search_gpu.py.
//
// Workgroup = 4 waves = 256 rows x (32 NQB) queries; wave w owns rows 64 w .. 64 w + 63 as two 32-row blocks
// (the B operand: lane l supplies row (l & 31), k = 2 t + (l >> 5)) against NQB 32-query blocks (the A operand),
// so D[query][row]: a lane's column is a ROW and S[q][r .. r + 31] leaves as 128-byte segments.  Operands pass
// through LDS in k-chunks of 32 (row stride 33 floats: the 32 lanes of a half-wave hit 32 banks), the next
// chunk waits in registers while this one is multiplied.
#define SM_ROWS 256
#define SM_KC 32
#define SM_LD 33
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <typename T, int NQB>
__global__ __launch_bounds__(256) void scan_dense_mfma_kernel(const T* __restrict__ rows, int64_t row0, int nrows,
                                                              int d_pad, const float* __restrict__ q_canon, int nq,
                                                              float* __restrict__ S, int64_t s_stride)
{
    constexpr int TQ = 32 * NQB;
    constexpr int XV = (sizeof(T) == 4) ? 8 : 4;          // 16-byte loads per thread per chunk of the row tile
    constexpr int QV = TQ * SM_KC / 4 / 256;              // float4 loads per thread per chunk of the query tile
    static_assert(QV >= 1, "query tile too small for 256 threads");
    __shared__ float xs[SM_ROWS * SM_LD];
    __shared__ float qs[TQ * SM_LD];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int q0 = blockIdx.x * TQ;
    const int r0 = blockIdx.y * SM_ROWS;
    const T* base = rows + (row0 + r0) * (int64_t)d_pad;

    uint4 xpre[XV];
    float4 qpre[QV];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int p = 0; p < XV; ++p) {
            int r, c;                                      // row of the tile, first element of the 16-byte piece
            if constexpr (sizeof(T) == 4) { r = (t >> 3) + 32 * p; c = (t & 7) * 4; }
            else { r = (t >> 2) + 64 * p; c = (t & 3) * 8; }
            xpre[p] = make_uint4(0, 0, 0, 0);
            if (r0 + r < nrows) xpre[p] = *reinterpret_cast<const uint4*>(base + (int64_t)r * d_pad + k0 + c);
        }
#pragma unroll
        for (int p = 0; p < QV; ++p) {
            const int j = (t >> 3) + 32 * p, c = (t & 7) * 4;
            qpre[p] = make_float4(0, 0, 0, 0);
            if (q0 + j < nq) qpre[p] = *reinterpret_cast<const float4*>(q_canon + (int64_t)(q0 + j) * d_pad + k0 + c);
        }
    };
    auto park = [&]() {
#pragma unroll
        for (int p = 0; p < XV; ++p) {
            const uint4 v = xpre[p];
            if constexpr (sizeof(T) == 4) {
                float* dst = xs + ((t >> 3) + 32 * p) * SM_LD + (t & 7) * 4;
                dst[0] = __builtin_bit_cast(float, v.x); dst[1] = __builtin_bit_cast(float, v.y);
                dst[2] = __builtin_bit_cast(float, v.z); dst[3] = __builtin_bit_cast(float, v.w);
            } else {
                float* dst = xs + ((t >> 2) + 64 * p) * SM_LD + (t & 3) * 8;
                dst[0] = __builtin_bit_cast(float, v.x << 16); dst[1] = __builtin_bit_cast(float, v.x & 0xFFFF0000u);
                dst[2] = __builtin_bit_cast(float, v.y << 16); dst[3] = __builtin_bit_cast(float, v.y & 0xFFFF0000u);
                dst[4] = __builtin_bit_cast(float, v.z << 16); dst[5] = __builtin_bit_cast(float, v.z & 0xFFFF0000u);
                dst[6] = __builtin_bit_cast(float, v.w << 16); dst[7] = __builtin_bit_cast(float, v.w & 0xFFFF0000u);
            }
        }
#pragma unroll
        for (int p = 0; p < QV; ++p) {
            float* dst = qs + ((t >> 3) + 32 * p) * SM_LD + (t & 7) * 4;
            dst[0] = qpre[p].x; dst[1] = qpre[p].y; dst[2] = qpre[p].z; dst[3] = qpre[p].w;
        }
    };

    f32x16 acc[NQB][2];
#pragma unroll
    for (int a = 0; a < NQB; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.0f;      // the chain starts from +0.0f

    const int l31 = lane & 31, h = lane >> 5;
    const float* xa = xs + (64 * wave + l31) * SM_LD + h;          // + 32 rb rows, + 2 t
    const float* qa = qs + l31 * SM_LD + h;                        // + 32 qb rows, + 2 t
    fetch(0);
    for (int k0 = 0; k0 < d_pad; k0 += SM_KC) {
        park();
        __syncthreads();
        if (k0 + SM_KC < d_pad) fetch(k0 + SM_KC);
#pragma unroll 4
        for (int tp = 0; tp < SM_KC / 2; ++tp) {
            float bx[2], aq[NQB];
#pragma unroll
            for (int b = 0; b < 2; ++b) bx[b] = xa[32 * b * SM_LD + 2 * tp];
#pragma unroll
            for (int a = 0; a < NQB; ++a) aq[a] = qa[32 * a * SM_LD + 2 * tp];
#pragma unroll
            for (int a = 0; a < NQB; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[a], bx[b], acc[a][b], 0, 0, 0);
        }
        __syncthreads();
    }
    // D[i = query][j = row]: lane column = row (l & 31), register g -> query (g & 3) + 8 (g >> 2) + 4 h
#pragma unroll
    for (int a = 0; a < NQB; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int r = r0 + 64 * wave + 32 * b + l31;
            if (r < nrows) {
#pragma unroll
                for (int g = 0; g < 16; ++g) {
                    const int q = q0 + 32 * a + (g & 3) + 8 * (g >> 2) + 4 * h;
                    if (q < nq) S[(int64_t)q * s_stride + r] = acc[a][b][g];
                }
            }
        }
}

template <typename T>
static void launch_scan_dense_mfma(hipStream_t st, const T* rows, int64_t row0, int nrows, int d_pad, const float* q_canon,
                                   int nq, float* S, int64_t s_stride)
{
    const unsigned gy = (unsigned)((nrows + SM_ROWS - 1) / SM_ROWS);
    if (nq <= 32) scan_dense_mfma_kernel<T, 1><<<dim3(1, gy), 256, 0, st>>>(rows, row0, nrows, d_pad, q_canon, nq, S, s_stride);
    else scan_dense_mfma_kernel<T, 2><<<dim3((unsigned)((nq + 63) / 64), gy), 256, 0, st>>>(rows, row0, nrows, d_pad, q_canon, nq, S, s_stride);
}

int mi_launch_scan_dense(hipStream_t st, const void* rows, int dtype, int64_t row0, int nrows, int d_pad,
                         const float* q_canon, int nq, float* S, int64_t s_stride)
{
    if (nrows <= 0 || nq <= 0) return MI_OK;
    if (nq <= 4 && nrows <= 65535 * 64) {
        if (dtype == MI_DTYPE_F32) launch_scan_dense_small<float>(st, (const float*)rows, row0, nrows, d_pad, q_canon, nq, S, s_stride);
        else launch_scan_dense_small<uint16_t>(st, (const uint16_t*)rows, row0, nrows, d_pad, q_canon, nq, S, s_stride);
        MI_HIP(hipGetLastError());
        return MI_OK;
    }
    dim3 grid((nq + SD_Q - 1) / SD_Q, (nrows + SD_ROWS - 1) / SD_ROWS);
    if (grid.y > 65535) { mi_set_error("scan_dense: segment too long"); return MI_ERR_INTERNAL; }
    // more than one 16-query group: the fp32 MFMA kernel (MI_EXACT_MFMA=0 keeps the VALU kernel, for A/B runs and
    // so that the tests can hold both to the oracle)
    static const bool mfma_off = getenv("MI_EXACT_MFMA") && atoi(getenv("MI_EXACT_MFMA")) == 0;
    if (nq > SD_Q && !mfma_off) {
        if (dtype == MI_DTYPE_F32) launch_scan_dense_mfma<float>(st, (const float*)rows, row0, nrows, d_pad, q_canon, nq, S, s_stride);
        else launch_scan_dense_mfma<uint16_t>(st, (const uint16_t*)rows, row0, nrows, d_pad, q_canon, nq, S, s_stride);
        MI_HIP(hipGetLastError());
        return MI_OK;
    }
    if (dtype == MI_DTYPE_F32)
        scan_dense_kernel<float><<<grid, 256, 0, st>>>((const float*)rows, row0, nrows, d_pad, q_canon, nq, S, s_stride);
    else
        scan_dense_kernel<uint16_t><<<grid, 256, 0, st>>>((const uint16_t*)rows, row0, nrows, d_pad, q_canon, nq, S, s_stride);
    MI_HIP(hipGetLastError());
    return MI_OK;
}

// ---------------------------------------------------------------- LDS bitonic sort helpers
// Sorts P (power of two) 64-bit keys descending.
__device__ inline void bitonic_desc_u64(uint64_t* keys, int P, int t, int nt)
{
    for (int size = 2; size <= P; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            __syncthreads();
            for (int i = t; i < (P >> 1); i += nt) {
                const int lo = 2 * i - (i & (stride - 1));
                const int hi = lo + stride;
                const bool desc = (lo & size) == 0;
                const uint64_t a = keys[lo], b = keys[hi];
                if ((a < b) == desc) { keys[lo] = b; keys[hi] = a; }
            }
        }
    }
    __syncthreads();
}

// ---------------------------------------------------------------- select_dense
#define SEL_THREADS 512

__device__ inline uint32_t dense_rank(const float* __restrict__ Srow, int i, const float* __restrict__ xx,
                                      float qqv, int metric)
{
    const float dot = Srow[i];
    return (metric == MI_METRIC_L2) ? mi_rank_l2(dot, qqv, xx[i]) : mi_rank_ip(dot);
}

// One workgroup per query.  The 64-bit key of row i is (rank word << 32) | ~i: unique, so the k-th
// largest is well defined and ties on the score resolve to the smaller id.  KPT > 0: the segment has
// at most SEL_THREADS * KPT rows and each thread keeps its rows' rank words in registers (one pass over
// S and xx); KPT == 0: any length, rank words recomputed from S in every pass.
// Per pass: histogram of one key byte over the keys that match the prefix so far, suffix sums by
// wave 0 (shuffles, no block-wide scan), pick the byte where the k-th key lies.  The pass loop ends
// as soon as every key in the chosen bucket is wanted - with float scores that is after 3-4 of the 8.
// The top byte of a rank word is the sign and most of the exponent - nearly every key lands in one
// or two bins - so that pass counts per wave with ballots instead of 20,000 LDS atomics on one word.
template <int KPT>
__global__ __launch_bounds__(SEL_THREADS) void select_dense_kernel(const float* __restrict__ S, int64_t s_stride,
        int nrows_all, int seg_len, const float* __restrict__ xx_all, const float* __restrict__ qq, int metric, int k,
        int64_t id0_all, float* __restrict__ out_s, int64_t* __restrict__ out_i, int64_t out_stride)
{
    __shared__ uint32_t hist[256];
    __shared__ uint64_t s_prefix;
    __shared__ int s_kk;
    __shared__ int s_done;
    __shared__ int s_cnt;
    __shared__ uint64_t keys[MI_MAX_K];

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int64_t q = blockIdx.x;
    // blockIdx.y: which seg_len-row piece of the segment this workgroup selects from; its k best go to
    // out[q * out_stride + blockIdx.y * k ..] (one piece: the final list; several: merged afterwards)
    const int64_t s0 = (int64_t)blockIdx.y * seg_len;
    const int nrows = (nrows_all - s0 < seg_len) ? (int)(nrows_all - s0) : seg_len;
    const float* Srow = S + q * s_stride + s0;
    const float* xx = xx_all + s0;
    const int64_t id0 = id0_all + s0;
    const float qqv = qq[q];
    float* os = out_s + q * out_stride + (int64_t)blockIdx.y * k;
    int64_t* oi = out_i + q * out_stride + (int64_t)blockIdx.y * k;

    constexpr int NREG = KPT > 0 ? KPT : 1;
    uint32_t rkc[NREG];
    if constexpr (KPT > 0) {
#pragma unroll
        for (int e = 0; e < KPT; ++e) {
            const int i = t + e * SEL_THREADS;
            rkc[e] = (i < nrows) ? dense_rank(Srow, i, xx, qqv, metric) : 0u;
        }
    }
    auto key_at = [&](int e, int i) -> uint64_t {
        uint32_t rk;
        if constexpr (KPT > 0) rk = rkc[e]; else rk = dense_rank(Srow, i, xx, qqv, metric);
        return ((uint64_t)rk << 32) | (uint32_t)(~(uint32_t)i);
    };

    int kk = (k < nrows) ? k : nrows;
    const int nsel = kk;
    uint64_t prefix = 0;
    if (nsel > 0 && nsel < nrows) {
        for (int pass = 7; pass >= 0; --pass) {
            if (t < 256) hist[t] = 0;
            if (t == 0) s_done = 0;
            __syncthreads();
            const int shift = pass * 8;
            auto count = [&](int e, int i) {
                const bool in = i < nrows;
                const uint64_t key = in ? key_at(e, i) : 0;
                const bool match = in && ((pass == 7) || ((key >> (shift + 8)) == (prefix >> (shift + 8))));
                const uint32_t digit = (uint32_t)(key >> shift) & 255u;
                if (pass == 7) {
                    // whole-wave loop (every lane runs it): one atomic per distinct digit in the wave
                    uint64_t todo = __ballot(match);
                    while (todo) {
                        const int first = __builtin_ctzll(todo);
                        const uint32_t dsel = (uint32_t)__shfl((int)digit, first, 64);
                        const uint64_t same = __ballot(match && digit == dsel) & todo;
                        if (lane == first) atomicAdd(&hist[dsel], (uint32_t)__builtin_popcountll(same));
                        todo &= ~same;
                    }
                } else if (match) {
                    atomicAdd(&hist[digit], 1u);
                }
            };
            if constexpr (KPT > 0) {
#pragma unroll
                for (int e = 0; e < KPT; ++e) count(e, t + e * SEL_THREADS);
            } else {
                const int nround = (nrows + SEL_THREADS - 1) / SEL_THREADS;      // uniform trip count: ballots inside
                for (int e = 0; e < nround; ++e) count(0, t + e * SEL_THREADS);
            }
            __syncthreads();
            if (t < 64) {
                // lane l owns bins 4l .. 4l+3; above(b) = keys whose digit is > b
                const uint32_t h0 = hist[4 * lane], h1 = hist[4 * lane + 1], h2 = hist[4 * lane + 2], h3 = hist[4 * lane + 3];
                const uint32_t mine = h0 + h1 + h2 + h3;
                uint32_t suf = mine;                                            // inclusive suffix sum over lanes
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const uint32_t o = (uint32_t)__shfl_down((int)suf, off, 64);
                    if (lane + off < 64) suf += o;
                }
                const uint32_t ab3 = suf - mine, ab2 = ab3 + h3, ab1 = ab2 + h2, ab0 = ab1 + h1;
                const uint32_t hh[4] = {h0, h1, h2, h3};
                const uint32_t ab[4] = {ab0, ab1, ab2, ab3};
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    if (ab[b] < (uint32_t)kk && (uint32_t)kk <= ab[b] + hh[b]) {  // exactly one bin qualifies
                        s_kk = kk - (int)ab[b];
                        s_prefix = prefix | ((uint64_t)(4 * lane + b) << shift);
                        s_done = ((uint32_t)kk == ab[b] + hh[b]) ? 1 : 0;       // the whole bucket is wanted: stop here
                    }
                }
            }
            __syncthreads();
            kk = s_kk;
            prefix = s_prefix;
            const int done = s_done;
            __syncthreads();
            if (done) break;
        }
    }
    // collect every key >= prefix (exactly nsel of them; all rows when nsel == nrows)
    if (t == 0) s_cnt = 0;
    __syncthreads();
    if (nsel > 0) {
        auto take = [&](int e, int i) {
            if (i < nrows) {
                const uint64_t key = key_at(e, i);
                if (key >= prefix) {
                    const int pos = atomicAdd(&s_cnt, 1);
                    if (pos < MI_MAX_K) keys[pos] = key;
                }
            }
        };
        if constexpr (KPT > 0) {
#pragma unroll
            for (int e = 0; e < KPT; ++e) take(e, t + e * SEL_THREADS);
        } else {
            for (int i = t; i < nrows; i += SEL_THREADS) take(0, i);
        }
    }
    __syncthreads();
    const int P = (nsel > 1) ? (1 << (32 - __builtin_clz(nsel - 1))) : 1;
    for (int i = nsel + t; i < P; i += SEL_THREADS) keys[i] = 0;
    bitonic_desc_u64(keys, P, t, SEL_THREADS);
    for (int j = t; j < k; j += SEL_THREADS) {
        if (j < nsel) {
            const uint64_t key = keys[j];
            os[j] = mi_score_from_rank(metric, (uint32_t)(key >> 32));
            oi[j] = id0 + (int64_t)(uint32_t)(~(uint32_t)key);
        } else {
            os[j] = mi_pad_score(metric);
            oi[j] = -1;
        }
    }
}

// Pieces of a segment for the select: one when there are enough queries to fill the chip (or the segment
// is short), else pieces of >= 4096 rows, at most 4096 / k of them (what one merge_kernel launch takes).
int mi_select_dense_pieces(int nrows, int nq, int k, int* seg_len)
{
    *seg_len = nrows;
    if (nq > 64 || nrows <= SEL_THREADS * 8) return 1;
    const int max_pieces = 4096 / k;
    if (max_pieces < 2) return 1;
    int64_t len = SEL_THREADS * 8;
    while ((nrows + len - 1) / len > max_pieces) len *= 2;
    if (len >= nrows) return 1;
    *seg_len = (int)len;
    return (int)((nrows + len - 1) / len);
}

int mi_launch_select_dense(hipStream_t st, const float* S, int64_t s_stride, int nrows, int seg_len, const float* xx,
                           const float* qq, int metric, int nq, int k, int64_t id0, float* out_s,
                           int64_t* out_i, int64_t out_stride)
{
    if (nq <= 0) return MI_OK;
    if (seg_len <= 0 || seg_len > nrows) seg_len = nrows;
    const dim3 grid((unsigned)nq, (unsigned)((nrows + seg_len - 1) / seg_len));
    if (seg_len <= SEL_THREADS * 8)
        select_dense_kernel<8><<<grid, SEL_THREADS, 0, st>>>(S, s_stride, nrows, seg_len, xx, qq, metric, k, id0, out_s, out_i, out_stride);
    else if (seg_len <= SEL_THREADS * 64)
        select_dense_kernel<64><<<grid, SEL_THREADS, 0, st>>>(S, s_stride, nrows, seg_len, xx, qq, metric, k, id0, out_s, out_i, out_stride);
    else
        select_dense_kernel<0><<<grid, SEL_THREADS, 0, st>>>(S, s_stride, nrows, seg_len, xx, qq, metric, k, id0, out_s, out_i, out_stride);
    MI_HIP(hipGetLastError());
    return MI_OK;
}

// ---------------------------------------------------------------- merge
// Order: rank word descending, then id ascending.  rank word 0 never occurs for a valid
// entry (NaN is canonicalised), so it marks padding.
#define MERGE_MAX 4096
#define MERGE_THREADS 256

// blockIdx.y = chunk: entries [y * chunk, min(m_all, (y + 1) * chunk)) of the query, k best to out[.. + y * k].
// Entry i of query q is element j = i % k_in of list l = i / k_in, stored at l * lstride + q * qstride + j:
//   one row per query, lists side by side  ([nq][nlists * k_in]):  lstride = k_in, qstride = row stride;
//   one block per list, as an all-gather leaves them ([nlists][nq][k_in]):  lstride = block stride, qstride = k_in.
__global__ __launch_bounds__(MERGE_THREADS) void merge_kernel(int metric, int m_all, int chunk, int P,
        const float* __restrict__ in_s, const int64_t* __restrict__ in_i, int k_in, int64_t lstride_s, int64_t lstride_i,
        int64_t qstride, int k, float* __restrict__ out_s, int64_t* __restrict__ out_i, int64_t out_stride)
{
    __shared__ uint32_t rk[MERGE_MAX];
    __shared__ int64_t id[MERGE_MAX];
    const int t = threadIdx.x;
    const int64_t q = blockIdx.x;
    const int e0 = (int)blockIdx.y * chunk;
    const int m = (m_all - e0 < chunk) ? m_all - e0 : chunk;
    out_s += (int64_t)blockIdx.y * k; out_i += (int64_t)blockIdx.y * k;
    for (int i = t; i < P; i += MERGE_THREADS) {
        uint32_t r = 0;
        int64_t v = INT64_MAX;
        if (i < m) {
            const int l = (e0 + i) / k_in, j = (e0 + i) - l * k_in;
            const int64_t x = in_i[l * lstride_i + q * qstride + j];
            if (x >= 0) {
                float s = in_s[l * lstride_s + q * qstride + j];
                if (metric == MI_METRIC_L2) { if (s != s) s = __builtin_inff(); s += 0.0f; r = ~mi_ord_f32(s); }
                else { if (s != s) s = -__builtin_inff(); s += 0.0f; r = mi_ord_f32(s); }
                v = x;
            }
        }
        rk[i] = r;
        id[i] = v;
    }
    for (int size = 2; size <= P; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            __syncthreads();
            for (int i = t; i < (P >> 1); i += MERGE_THREADS) {
                const int lo = 2 * i - (i & (stride - 1));
                const int hi = lo + stride;
                const bool desc = (lo & size) == 0;
                const uint32_t ra = rk[lo], rb = rk[hi];
                const int64_t ia = id[lo], ib = id[hi];
                const bool a_worse = (ra < rb) || (ra == rb && ia > ib);
                if (a_worse == desc && !(ra == rb && ia == ib)) { rk[lo] = rb; rk[hi] = ra; id[lo] = ib; id[hi] = ia; }
            }
        }
    }
    __syncthreads();
    for (int j = t; j < k; j += MERGE_THREADS) {
        const bool valid = (j < P) && id[j] != INT64_MAX;
        out_s[q * out_stride + j] = valid ? mi_score_from_rank(metric, rk[j]) : mi_pad_score(metric);
        out_i[q * out_stride + j] = valid ? id[j] : -1;
    }
}

int mi_launch_merge_ex(hipStream_t st, int metric, int64_t nq, int m, int chunk, const float* in_s, const int64_t* in_i,
                       int k_in, int64_t lstride_s, int64_t lstride_i, int64_t qstride,
                       int k, float* out_s, int64_t* out_i, int64_t out_stride)
{
    if (nq <= 0 || m <= 0) return MI_OK;
    if (chunk < 1 || chunk > MERGE_MAX || k_in < 1) { mi_set_error("merge: chunk %d outside [1, %d]", chunk, MERGE_MAX); return MI_ERR_INTERNAL; }
    const int nch = (m + chunk - 1) / chunk;
    if (nq > 0x7FFFFFFF || nch > 65535) { mi_set_error("merge: grid too large"); return MI_ERR_INTERNAL; }
    const int P = mi_next_pow2(chunk < m ? chunk : m);
    merge_kernel<<<dim3((unsigned)nq, (unsigned)nch), MERGE_THREADS, 0, st>>>(metric, m, chunk, P, in_s, in_i, k_in, lstride_s, lstride_i,
                                                                            qstride, k, out_s, out_i, out_stride);
    MI_HIP(hipGetLastError());
    return MI_OK;
}

int mi_launch_merge(hipStream_t st, int metric, int64_t nq, int m, const float* in_s, const int64_t* in_i,
                    int64_t in_stride, int k, float* out_s, int64_t* out_i, int64_t out_stride)
{
    if (nq <= 0) return MI_OK;
    if (m > MERGE_MAX) { mi_set_error("merge: %d entries per query exceeds %d", m, MERGE_MAX); return MI_ERR_INTERNAL; }
    if (m <= 0) return mi_launch_fill_pad(st, metric, nq, k, out_s, out_i, out_stride);
    return mi_launch_merge_ex(st, metric, nq, m, m, in_s, in_i, m, 0, 0, in_stride, k, out_s, out_i, out_stride);
}

int mi_launch_merge_chunks(hipStream_t st, int metric, int64_t nq, int m, int chunk, const float* in_s, const int64_t* in_i,
                           int64_t in_stride, int k, float* out_s, int64_t* out_i, int64_t out_stride)
{
    return mi_launch_merge_ex(st, metric, nq, m, chunk, in_s, in_i, m, 0, 0, in_stride, k, out_s, out_i, out_stride);
}

// ---------------------------------------------------------------- small list utilities
__global__ __launch_bounds__(256) void fill_pad_kernel(int metric, int64_t nq, int k, float* s, int64_t* i, int64_t stride)
{
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= nq * k) return;
    const int64_t q = g / k;
    const int j = (int)(g - q * k);
    s[q * stride + j] = mi_pad_score(metric);
    i[q * stride + j] = -1;
}
int mi_launch_fill_pad(hipStream_t st, int metric, int64_t nq, int k, float* s, int64_t* i, int64_t stride)
{
    if (nq <= 0) return MI_OK;
    fill_pad_kernel<<<(unsigned)((nq * k + 255) / 256), 256, 0, st>>>(metric, nq, k, s, i, stride);
    MI_HIP(hipGetLastError());
    return MI_OK;
}

__global__ __launch_bounds__(256) void copy_lists_kernel(int64_t nq, int k, const float* __restrict__ in_s,
        const int64_t* __restrict__ in_i, int64_t in_stride, const int32_t* __restrict__ dst_q,
        float* __restrict__ out_s, int64_t* __restrict__ out_i, int64_t out_stride)
{
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= nq * k) return;
    const int64_t q = g / k;
    const int j = (int)(g - q * k);
    const int64_t dq = dst_q ? (int64_t)dst_q[q] : q;
    out_s[dq * out_stride + j] = in_s[q * in_stride + j];
    out_i[dq * out_stride + j] = in_i[q * in_stride + j];
}
int mi_launch_copy_lists(hipStream_t st, int64_t nq, int k, const float* in_s, const int64_t* in_i,
                         int64_t in_stride, const int32_t* dst_q, float* out_s, int64_t* out_i, int64_t out_stride)
{
    if (nq <= 0) return MI_OK;
    copy_lists_kernel<<<(unsigned)((nq * k + 255) / 256), 256, 0, st>>>(nq, k, in_s, in_i, in_stride, dst_q, out_s, out_i, out_stride);
    MI_HIP(hipGetLastError());
    return MI_OK;
}

__global__ __launch_bounds__(256) void gather_queries_kernel(const int32_t* __restrict__ sel, int nsel, int d_pad,
        const float* __restrict__ q_canon, const float* __restrict__ qq, float* __restrict__ q_out, float* __restrict__ qq_out)
{
    const int j = blockIdx.x;
    const int64_t src = sel[j];
    for (int c = threadIdx.x; c < d_pad; c += 256) q_out[(int64_t)j * d_pad + c] = q_canon[src * d_pad + c];
    if (threadIdx.x == 0) qq_out[j] = qq[src];
}
int mi_launch_gather_queries(hipStream_t st, const int32_t* sel, int nsel, int d_pad, const float* q_canon,
                             const float* qq, float* q_out, float* qq_out)
{
    if (nsel <= 0) return MI_OK;
    gather_queries_kernel<<<nsel, 256, 0, st>>>(sel, nsel, d_pad, q_canon, qq, q_out, qq_out);
    MI_HIP(hipGetLastError());
    return MI_OK;
}
