// ingest.hip — index build side of the hot path (FAISS.from_documents, advanced_rag.py:295):
// copy/convert rows into the padded row-major device matrix, per-row squared norms, query
// preparation, and the counter-based synthetic corpus generator.
#include "kernels.h"

// ---------------------------------------------------------------- ingest: convert + zero-pad
// One thread per output element pair is plenty: mi_index_add is bounded by PCIe, and the
// bench corpus is produced by synth_kernel below.
template <typename SRC, typename DST>
__global__ __launch_bounds__(256) void ingest_kernel(const SRC* __restrict__ src, int64_t src_pitch, DST* __restrict__ dst,
                                                     int64_t n, int d, int d_pad)
{
    const int64_t total = n * (int64_t)d_pad;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / d_pad;
        const int c = (int)(i - r * d_pad);
        float v = 0.0f;
        if (c < d) {
            if constexpr (sizeof(SRC) == 4) v = (float)src[r * src_pitch + c];
            else v = mi_bf16_to_f32((uint16_t)src[r * src_pitch + c]);
        }
        if constexpr (sizeof(DST) == 4) dst[i] = v;
        else dst[i] = mi_f32_to_bf16(v);
    }
}

int mi_launch_ingest(hipStream_t st, const void* src, int src_dtype, int64_t src_pitch, int64_t n, int d, int d_pad,
                     int dst_dtype, void* dst)
{
    if (n == 0) return MI_OK;
    const int64_t total = n * (int64_t)d_pad;
    const int grid = (int)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
    if (src_dtype == MI_DTYPE_F32 && dst_dtype == MI_DTYPE_F32)
        ingest_kernel<float, float><<<grid, 256, 0, st>>>((const float*)src, src_pitch, (float*)dst, n, d, d_pad);
    else if (src_dtype == MI_DTYPE_F32 && dst_dtype == MI_DTYPE_BF16)
        ingest_kernel<float, uint16_t><<<grid, 256, 0, st>>>((const float*)src, src_pitch, (uint16_t*)dst, n, d, d_pad);
    else if (src_dtype == MI_DTYPE_BF16 && dst_dtype == MI_DTYPE_BF16)
        ingest_kernel<uint16_t, uint16_t><<<grid, 256, 0, st>>>((const uint16_t*)src, src_pitch, (uint16_t*)dst, n, d, d_pad);
    else
        ingest_kernel<uint16_t, float><<<grid, 256, 0, st>>>((const uint16_t*)src, src_pitch, (float*)dst, n, d, d_pad);
    MI_HIP(hipGetLastError());
    return MI_OK;
}

template <typename T>
__global__ __launch_bounds__(256) void export_kernel(const T* __restrict__ rows, float* __restrict__ out,
                                                     int64_t n, int d, int d_pad)
{
    const int64_t total = n * (int64_t)d;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / d;
        const int c = (int)(i - r * d);
        if constexpr (sizeof(T) == 4) out[i] = rows[r * d_pad + c];
        else out[i] = mi_bf16_to_f32(rows[r * d_pad + c]);
    }
}
int mi_launch_export_rows(hipStream_t st, const void* rows, int dtype, int64_t n, int d, int d_pad, float* out)
{
    if (n == 0) return MI_OK;
    const int64_t total = n * (int64_t)d;
    const int grid = (int)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
    if (dtype == MI_DTYPE_F32) export_kernel<float><<<grid, 256, 0, st>>>((const float*)rows, out, n, d, d_pad);
    else export_kernel<uint16_t><<<grid, 256, 0, st>>>((const uint16_t*)rows, out, n, d, d_pad);
    MI_HIP(hipGetLastError());
    return MI_OK;
}

// ---------------------------------------------------------------- row norms: xx[r] = canonical dot(x_r, x_r)
// A workgroup takes 64 rows; each k-chunk of 64 elements is read coalesced (whole 128/256-B
// row segments), transposed through LDS, and thread r<64 continues row r's fmaf chain in k
// order.  The other three waves only help with the loads.
template <typename T>
__global__ __launch_bounds__(256) void row_norms_kernel(const T* __restrict__ rows, int64_t n, int d_pad,
                                                        float* __restrict__ xx, float* __restrict__ max_xx)
{
    __shared__ float xs[64][68];
    const int t = threadIdx.x;
    const int64_t r0 = (int64_t)blockIdx.x * 64;
    float acc = 0.0f;
    for (int k0 = 0; k0 < d_pad; k0 += 64) {
        if constexpr (sizeof(T) == 4) {
            // 64 rows x 256 B: 16 lanes x 16 B per row, 16 rows per pass
            for (int p = 0; p < 4; ++p) {
                const int r = p * 16 + (t >> 4), c = (t & 15) * 4;
                float4 v = make_float4(0, 0, 0, 0);
                if (r0 + r < n) v = *reinterpret_cast<const float4*>(rows + (r0 + r) * d_pad + k0 + c);
                *reinterpret_cast<float4*>(&xs[r][c]) = v;
            }
        } else {
            // 64 rows x 128 B: 8 lanes x 16 B per row, 32 rows per pass
            for (int p = 0; p < 2; ++p) {
                const int r = p * 32 + (t >> 3), c = (t & 7) * 8;
                uint4 v = make_uint4(0, 0, 0, 0);
                if (r0 + r < n) v = *reinterpret_cast<const uint4*>(rows + (r0 + r) * d_pad + k0 + c);
                float4 lo, hi;
                lo.x = __builtin_bit_cast(float, v.x << 16); lo.y = __builtin_bit_cast(float, v.x & 0xFFFF0000u);
                lo.z = __builtin_bit_cast(float, v.y << 16); lo.w = __builtin_bit_cast(float, v.y & 0xFFFF0000u);
                hi.x = __builtin_bit_cast(float, v.z << 16); hi.y = __builtin_bit_cast(float, v.z & 0xFFFF0000u);
                hi.z = __builtin_bit_cast(float, v.w << 16); hi.w = __builtin_bit_cast(float, v.w & 0xFFFF0000u);
                *reinterpret_cast<float4*>(&xs[r][c]) = lo;
                *reinterpret_cast<float4*>(&xs[r][c + 4]) = hi;
            }
        }
        __syncthreads();
        if (t < 64) {
#pragma unroll
            for (int kk = 0; kk < 64; kk += 4) {
                const float4 v = *reinterpret_cast<const float4*>(&xs[t][kk]);
                acc = __builtin_fmaf(v.x, v.x, acc);
                acc = __builtin_fmaf(v.y, v.y, acc);
                acc = __builtin_fmaf(v.z, v.z, acc);
                acc = __builtin_fmaf(v.w, v.w, acc);
            }
        }
        __syncthreads();
    }
    if (t < 64 && r0 + t < n) {
        xx[r0 + t] = acc;
        // non-negative floats order like their bit patterns
        atomicMax(reinterpret_cast<unsigned int*>(max_xx), __builtin_bit_cast(unsigned int, acc));
    }
}

int mi_launch_row_norms(hipStream_t st, const void* rows, int dtype, int64_t n, int d_pad, float* xx, float* max_xx)
{
    if (n == 0) return MI_OK;
    const int64_t grid = (n + 63) / 64;
    if (dtype == MI_DTYPE_F32) row_norms_kernel<float><<<(unsigned)grid, 256, 0, st>>>((const float*)rows, n, d_pad, xx, max_xx);
    else row_norms_kernel<uint16_t><<<(unsigned)grid, 256, 0, st>>>((const uint16_t*)rows, n, d_pad, xx, max_xx);
    MI_HIP(hipGetLastError());
    return MI_OK;
}

// ---------------------------------------------------------------- synthetic rows
// Bit-identical to oracle_synth_rows (oracle/oracle_search.c): integer Irwin-Hall values,
// exact int64 sum of squares, one correctly rounded 1/sqrtf, one multiply per element.
__host__ __device__ inline uint64_t splitmix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__host__ __device__ inline int32_t synth_m(uint64_t seed, uint64_t row, uint32_t k) {
    const uint64_t h = splitmix64(seed ^ splitmix64(row * 0x100000000ull + k));
    return (int32_t)((h & 0xFFFF) + ((h >> 16) & 0xFFFF) + ((h >> 32) & 0xFFFF) + (h >> 48)) - 131070;
}

template <typename T>
__global__ __launch_bounds__(256) void synth_kernel(T* __restrict__ dst, int64_t n, int d, int d_pad,
                                                    uint64_t seed, int64_t row0, int normalize)
{
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n) return;                       // whole wave leaves together
    const uint64_t grow = (uint64_t)(row0 + r);
    long long ss = 0;
    for (int k = lane; k < d; k += 64) { const long long m = synth_m(seed, grow, (uint32_t)k); ss += m * m; }
    for (int off = 32; off > 0; off >>= 1) {
        const int lo = __shfl_xor((int)(ss & 0xFFFFFFFFll), off, 64);
        const int hi = __shfl_xor((int)(ss >> 32), off, 64);
        ss += (long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
    }
    float scale = 1.0f / 65536.0f;
    if (normalize) scale = (ss > 0) ? 1.0f / sqrtf((float)ss) : 0.0f;
    T* out = dst + r * (int64_t)d_pad;
    for (int k = lane; k < d_pad; k += 64) {
        const float v = (k < d) ? (float)synth_m(seed, grow, (uint32_t)k) * scale : 0.0f;
        if constexpr (sizeof(T) == 4) out[k] = v;
        else out[k] = mi_f32_to_bf16(v);
    }
}

int mi_launch_synth(hipStream_t st, void* dst, int dtype, int64_t n, int d, int d_pad, uint64_t seed,
                    int64_t row0, int normalize)
{
    if (n == 0) return MI_OK;
    const int64_t grid = (n + 3) / 4;
    if (grid > 0x7FFFFFFF) { mi_set_error("synth: too many rows in one call"); return MI_ERR_INVALID_ARG; }
    if (dtype == MI_DTYPE_F32) synth_kernel<float><<<(unsigned)grid, 256, 0, st>>>((float*)dst, n, d, d_pad, seed, row0, normalize);
    else synth_kernel<uint16_t><<<(unsigned)grid, 256, 0, st>>>((uint16_t*)dst, n, d, d_pad, seed, row0, normalize);
    MI_HIP(hipGetLastError());
    return MI_OK;
}

// ---------------------------------------------------------------- query preparation
// q_canon [nq][d_pad_c] fp32 (bf16-rounded values when round_canon, zero padded), optional
// q_bf16 [nq][d_pad_s] bf16 bits (always rounded), qq[q] = canonical dot(q,q) of the q_canon
// values.  One wave per query.  The chain for qq is one lane's work by definition (k ascending,
// one fmaf each); the wave stages 256 values at a time in LDS so that lane reads them as
// ds_read_b128 instead of 384 dependent global loads (25 us -> a few us for a single query).
// Padding a block with zeros leaves the chain's value unchanged (fmaf(0,0,acc) == acc, acc >= +0).
__global__ __launch_bounds__(256) void prep_queries_kernel(const float* __restrict__ q, int64_t nq, int d, int d_pad_c,
                                                           int round_canon, float* __restrict__ q_canon, int d_pad_s,
                                                           uint16_t* __restrict__ q_bf16, float* __restrict__ qq)
{
    __shared__ __attribute__((aligned(16))) float stage[4][256];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t qi = (int64_t)blockIdx.x * 4 + w;
    const bool live = qi < nq;
    const int kmax = d_pad_c > d_pad_s ? d_pad_c : d_pad_s;
    float acc = 0.0f;
    for (int k0 = 0; k0 < kmax; k0 += 256) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = k0 + lane + 64 * i;
            const float v = (live && k < d) ? q[qi * d + k] : 0.0f;
            const uint16_t hb = mi_f32_to_bf16(v);
            const float c = round_canon ? mi_bf16_to_f32(hb) : v;
            if (live && k < kmax) {
                if (q_bf16 && k < d_pad_s) q_bf16[qi * d_pad_s + k] = hb;
                if (k < d_pad_c) q_canon[qi * d_pad_c + k] = c;
            }
            stage[w][lane + 64 * i] = c;
        }
        __syncthreads();
        if (live && lane == 0 && k0 < d) {
            const int m = (d - k0 < 256) ? d - k0 : 256;
            for (int j = 0; j < m; j += 4) {
                const float4 v = *reinterpret_cast<const float4*>(&stage[w][j]);
                acc = __builtin_fmaf(v.x, v.x, acc);
                acc = __builtin_fmaf(v.y, v.y, acc);
                acc = __builtin_fmaf(v.z, v.z, acc);
                acc = __builtin_fmaf(v.w, v.w, acc);
            }
        }
        __syncthreads();
    }
    if (live && lane == 0) qq[qi] = acc;
}

int mi_launch_prep_queries(hipStream_t st, const float* q, int64_t nq, int d, int d_pad_c, int round_canon,
                           float* q_canon, int d_pad_s, uint16_t* q_bf16, float* qq)
{
    if (nq == 0) return MI_OK;
    prep_queries_kernel<<<(unsigned)((nq + 3) / 4), 256, 0, st>>>(q, nq, d, d_pad_c, round_canon, q_canon, d_pad_s, q_bf16, qq);
    MI_HIP(hipGetLastError());
    return MI_OK;
}
