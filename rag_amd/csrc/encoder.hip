// encoder.hip — MiniLM/BERT-class bi-encoder forward on gfx950 (MI355X), from token ids to
// L2-normalised sentence embeddings.  Replaces HuggingFaceEmbeddings(...).embed_documents /
// embed_query (advanced_rag.py:104-107, called inside FAISS.from_documents :295 and the
// retriever :296), i.e. sentence-transformers -> transformers BertModel (modeling_bert.py):
//
//   embed_ln_kernel   word + token_type(0) + position, LayerNorm            (:98-107)
//   gemm_ws_kernel    QKV projection (bias), weight-stationary: W fragments in registers, the
//                     activations streamed once through LDS; also FFN1 (bias + erf-GELU) in the
//                     unfused form                                         (:175-177, :334-337)
//   attn_kernel       softmax(Q K^T dh^-1/2 + mask) V per (sequence, head): swapped-QK^T MFMA
//                     so each lane owns one query's scores, online softmax in registers, the
//                     P tile re-used as the next MFMA's B operand without leaving registers
//                                                                          (:111-136)
//   ffn3_kernel       (encoder_ffn3.hip; hidden 384 / intermediate 1536, the default layer tail) and
//   ffn2_kernel       per 128-row tile: [attention output projection + residual + LayerNorm]
//                     -> FFN1 + erf-GELU -> FFN2 + residual + LayerNorm; neither the
//                     post-attention hidden state nor the 1536-wide intermediate leaves the CU
//                                                                          (:289-293, :334-351)
//   gemm_ln_kernel    one row-complete GEMM + residual + LayerNorm (the pieces of the above, unfused)
//   gemm_bf16_kernel  x W^T + b on 128x128 tiles with fused epilogues: the general form
//                     (other hidden sizes, small batches); add_ln_kernel: its LayerNorm
//   pool_kernel       masked mean over tokens, x / max(|x|, 1e-12)          (sentence-transformers)
//
// Hidden state bf16 [tokens][hidden] row-major; QKV / context blocked [tokens/128][cols/32][128][32]
// when the large-batch kernels produce them.  Every reduction (LN statistics, softmax max/sum,
// MFMA accumulation, pooling) is fp32.  Tolerance vs the fp32 CPU oracle: 1 - cos <= 1e-3
// (BASELINE.json north_star), measured ~5e-6 in tests/test_encoder_gpu.py.
#include <float.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <type_traits>
#include <utility>
#include <mutex>
#include <new>
#include <vector>
#include "kernels.h"
#include "encoder_ffn3.h"
#include "encoder_qkv_attn.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef short tr4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) tr4_t tr4_lds_t;

template <int V> using IC = std::integral_constant<int, V>;
template <int... I, class F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, I...>, F&& f) { (f(IC<I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(std::make_integer_sequence<int, N>{}, static_cast<F&&>(f)); }
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

__device__ __forceinline__ float bf2f(uint16_t h) { return __builtin_bit_cast(float, (uint32_t)h << 16); }
// fp32 -> bf16, round-to-nearest-even: a plain cast is one v_cvt_pk_bf16_f32 per pair on gfx950
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint16_t f2bf(float f) { return __builtin_bit_cast(uint16_t, (__bf16)f); }
__device__ __forceinline__ uint32_t pack2(float lo, float hi)
{
    bf16x2 v;
    v[0] = (__bf16)lo;
    v[1] = (__bf16)hi;
    return __builtin_bit_cast(uint32_t, v);
}

// erf-GELU, x * 0.5 * (1 + erf(x / sqrt 2))  (modeling_bert.py:334-337, activations.py GELUActivation), as
//   gelu(x) = max(x, 0) - |x| / 2 * erfc(|x| / sqrt 2),   erfc(z) ~= 2^(-z q(z)),
// q a degree-4 polynomial fitted (minimax, z in [0, 4.5], increasing beyond) to -log2(erfc(z)) / z:
// |erfc error| <= 6.8e-7 evaluated in fp32, far below the bf16 rounding of the result.  One v_exp,
// no reciprocal, no sign fix-up: 9 full-rate operations per value against 14 + v_rcp for
// Abramowitz-Stegun 7.1.26, and the Horner chain runs two values per instruction (v_pk_fma_f32).
// FFN1 spends as long in this function as in its MFMAs, so its cost is the kernel's.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 gelu_erf2(f32x2 x)
{
    f32x2 z;
    z.x = fabsf(x.x) * 0.70710678118654752440f;
    z.y = fabsf(x.y) * 0.70710678118654752440f;
    f32x2 q = __builtin_elementwise_fma(z, f32x2{0.0029441735549769255f, 0.0029441735549769255f},
                                        f32x2{-0.0295900954803597f, -0.0295900954803597f});
    q = __builtin_elementwise_fma(q, z, f32x2{0.14866568630789534f, 0.14866568630789534f});
    q = __builtin_elementwise_fma(q, z, f32x2{0.9185093434591132f, 0.9185093434591132f});
    q = __builtin_elementwise_fma(q, z, f32x2{1.6278890079463915f, 1.6278890079463915f});
    const f32x2 p = q * z;
    f32x2 e;
    e.x = __builtin_amdgcn_exp2f(-p.x);
    e.y = __builtin_amdgcn_exp2f(-p.y);
    const f32x2 t = z * -0.70710678118654752440f;                    // -|x| / 2
    f32x2 r;                                                          // max(x, 0) in one instruction
    r.x = __builtin_amdgcn_fmed3f(x.x, 0.0f, __builtin_inff());
    r.y = __builtin_amdgcn_fmed3f(x.y, 0.0f, __builtin_inff());
    return __builtin_elementwise_fma(t, e, r);
}
__device__ __forceinline__ void gelu_erf4(float& v0, float& v1, float& v2, float& v3)
{
    const f32x2 a = gelu_erf2(f32x2{v0, v1}), b = gelu_erf2(f32x2{v2, v3});
    v0 = a.x; v1 = a.y; v2 = b.x; v3 = b.y;
}

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// ---------------------------------------------------------------- embeddings + LayerNorm
// One wave per token; lane holds columns lane, lane+64, ... (H <= 1024).
template <int NCH>
__global__ __launch_bounds__(256) void embed_ln_kernel(const int32_t* __restrict__ ids, const int32_t* __restrict__ type_ids,
        const int32_t* __restrict__ pos_ids, int ntypes, int64_t T, int L, int H, int vocab,
        const float* __restrict__ word, const float* __restrict__ pos, const float* __restrict__ type_tab,
        const float* __restrict__ gamma, const float* __restrict__ beta, float eps, uint16_t* __restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const int64_t tok = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tok >= T) return;
    int id = ids[tok];
    if (id < 0 || id >= vocab) id = 0;              // ids are validated on the host; never index out of the table
    // token type: 0 for a bi-encoder (type_ids == nullptr); the segment id of a cross-encoder pair otherwise
    int tt = type_ids ? type_ids[tok] : 0;
    if (tt < 0 || tt >= ntypes) tt = 0;
    const float* const type0 = type_tab + (int64_t)tt * H;
    const int p = pos_ids ? pos_ids[tok] : (int)(tok % L);          // packed (unpadded) sequences bring their positions
    float x[NCH][8];
    float s = 0.0f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int col = 8 * lane + 512 * c;
#pragma unroll
        for (int e = 0; e < 8; ++e) x[c][e] = 0.0f;
        if (col < H) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const float4 w = *reinterpret_cast<const float4*>(word + (int64_t)id * H + col + 4 * q);
                const float4 t0 = *reinterpret_cast<const float4*>(type0 + col + 4 * q);
                const float4 ps = *reinterpret_cast<const float4*>(pos + (int64_t)p * H + col + 4 * q);
                x[c][4 * q + 0] = (w.x + t0.x) + ps.x; x[c][4 * q + 1] = (w.y + t0.y) + ps.y;
                x[c][4 * q + 2] = (w.z + t0.z) + ps.z; x[c][4 * q + 3] = (w.w + t0.w) + ps.w;
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) s += x[c][e];
        }
    }
    const float mean = wave_sum(s) / (float)H;
    float v = 0.0f;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
        if (8 * lane + 512 * c < H) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float dlt = x[c][e] - mean; v += dlt * dlt; }
        }
    const float inv = 1.0f / sqrtf(wave_sum(v) / (float)H + eps);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int col = 8 * lane + 512 * c;
        if (col < H) {
            const float4 g0 = *reinterpret_cast<const float4*>(gamma + col), g1 = *reinterpret_cast<const float4*>(gamma + col + 4);
            const float4 b0 = *reinterpret_cast<const float4*>(beta + col), b1 = *reinterpret_cast<const float4*>(beta + col + 4);
            uint4 o;
            o.x = pack2((x[c][0] - mean) * inv * g0.x + b0.x, (x[c][1] - mean) * inv * g0.y + b0.y);
            o.y = pack2((x[c][2] - mean) * inv * g0.z + b0.z, (x[c][3] - mean) * inv * g0.w + b0.w);
            o.z = pack2((x[c][4] - mean) * inv * g1.x + b1.x, (x[c][5] - mean) * inv * g1.y + b1.y);
            o.w = pack2((x[c][6] - mean) * inv * g1.z + b1.z, (x[c][7] - mean) * inv * g1.w + b1.w);
            *reinterpret_cast<uint4*>(out + tok * H + col) = o;
        }
    }
}

// LayerNorm of a bf16 row (the GEMM epilogue already added bias and residual).  One wave per
// token; lane l holds the 8 columns [8 l + 512 c, +8) of chunk c (16-byte loads and stores).
template <int NCH>
__global__ __launch_bounds__(256) void add_ln_kernel(const uint16_t* __restrict__ in, int64_t T, int H,
        const float* __restrict__ gamma, const float* __restrict__ beta, float eps, uint16_t* __restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const int64_t tok = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tok >= T) return;
    float x[NCH][8];
    float s = 0.0f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int col = 8 * lane + 512 * c;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (col < H) v = *reinterpret_cast<const uint4*>(in + tok * H + col);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            x[c][2 * e] = __builtin_bit_cast(float, w[e] << 16);
            x[c][2 * e + 1] = __builtin_bit_cast(float, w[e] & 0xFFFF0000u);
            s += x[c][2 * e] + x[c][2 * e + 1];
        }
    }
    const float mean = wave_sum(s) / (float)H;
    float v2 = 0.0f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int col = 8 * lane + 512 * c;
        if (col < H) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float dlt = x[c][e] - mean; v2 += dlt * dlt; }
        }
    }
    const float inv = 1.0f / sqrtf(wave_sum(v2) / (float)H + eps);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int col = 8 * lane + 512 * c;
        if (col < H) {
            const float4 g0 = *reinterpret_cast<const float4*>(gamma + col), g1 = *reinterpret_cast<const float4*>(gamma + col + 4);
            const float4 b0 = *reinterpret_cast<const float4*>(beta + col), b1 = *reinterpret_cast<const float4*>(beta + col + 4);
            uint4 o;
            o.x = pack2((x[c][0] - mean) * inv * g0.x + b0.x, (x[c][1] - mean) * inv * g0.y + b0.y);
            o.y = pack2((x[c][2] - mean) * inv * g0.z + b0.z, (x[c][3] - mean) * inv * g0.w + b0.w);
            o.z = pack2((x[c][4] - mean) * inv * g1.x + b1.x, (x[c][5] - mean) * inv * g1.y + b1.y);
            o.w = pack2((x[c][6] - mean) * inv * g1.z + b1.z, (x[c][7] - mean) * inv * g1.w + b1.w);
            *reinterpret_cast<uint4*>(out + tok * H + col) = o;
        }
    }
}

// ---------------------------------------------------------------- bf16 MFMA GEMM
// C[M][N] = A[M][K] * W[N][K]^T (+ epilogue).  Tile 128 x 128 x 64, four waves (2 x 2), each
// wave a 64 x 64 sub-tile = 2 x 2 MFMA 32x32x16 blocks.  Both operands are K-contiguous, so
// both stream global -> LDS by LDS-DMA as whole 128-B row segments (8 rows per 1-KiB piece);
// the 16-B chunk order inside a row is XOR-swizzled with ((row >> 1) & 7) on the SOURCE side,
// which makes every ds_read_b128 fragment read conflict-free on 128-B rows.  Two stages of
// 32 KB -> two workgroups per CU, i.e. two waves per SIMD hiding each other's waits.
//
// The product is formed transposed (W fragment as the MFMA A operand, activation fragment as
// B): a lane then owns one output ROW (token) and its 16 accumulator registers walk the columns
// in groups of four consecutive ones, so bias / GELU are per-register, the tile is parked in LDS
// with 8- or 16-byte writes, and leaves as whole 256-B row segments (16-byte coalesced stores;
// the residual is read the same way).  One rounding to bf16, after bias + residual.
//
// Workgroups that share an activation row tile (the N / 128 column tiles of one m-tile) carry
// the same blockIdx % 8, i.e. run on one XCD and fetch that tile from HBM once.
#define GM_BM 128
#define GM_BN 128
#define GM_BK 64
#define GM_STAGE (2 * 128 * 128)   // A tile + W tile, bytes
#define GM_TS16 272                // row stride of the parked bf16 tile (bytes)
#define GM_TS32 528                // row stride of the parked fp32 tile (bytes)
enum { EPI_BIAS = 0, EPI_GELU = 1, EPI_RESID = 2 };

__device__ __forceinline__ void lds_read2(bf16x8& a, bf16x8& b, unsigned addr)
{
    asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:4096" : "=&v"(a), "=&v"(b) : "v"(addr));
}

// NST = 1: the form above.  NST = 4 (small grids, one workgroup per CU at most): a ring of four stages with the
// loads of k-steps kt + 1 .. kt + 3 in flight while kt is multiplied.  With hundreds of tiles the four co-resident
// workgroups of NST = 1 hide each other's load latency; with a few tiles (a query, a few chunks) nothing does, and
// every k-step of NST = 1 costs a full trip to memory: 16 x 384 x 1152 took 9.8 us, K = 1536 three times that.
template <int EPI, int NST>
__global__ __launch_bounds__(256, NST == 1 ? 4 : 1) void gemm_bf16_kernel(const uint16_t* __restrict__ A, const uint16_t* __restrict__ W,
        const float* __restrict__ bias, const uint16_t* __restrict__ resid, uint16_t* __restrict__ C,
        int M, int N, int K, int n_tiles, int m_tiles)
{
    // ONE 32-KB stage (A tile + W tile) and a two-barrier k-loop: the workgroup itself does not
    // overlap its loads with its MFMAs; four workgroups per CU (36 KB LDS, <= 128 registers) do.
    // Measured against the double-buffered two-per-CU form of this kernel: see DESIGN.md.
    constexpr int PARK = (EPI == EPI_RESID) ? 64 * GM_TS32 : 64 * GM_TS16;     // half a tile (64 rows) parked at a time
    constexpr int LDS_BYTES = PARK > NST * GM_STAGE ? PARK : NST * GM_STAGE;
    __shared__ __attribute__((aligned(1024))) char lds[LDS_BYTES];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5, col = lane & 31;
    const int wm = wave >> 1, wn = wave & 1;
    // XCD-aware tile order: blocks b, b + 8, ... (one XCD) walk the column tiles of m-tile
    // 8 (j / n_tiles) + (b & 7); the host pads the grid to a multiple of 8 m-tiles.
    const int bx = blockIdx.x;
    const int mt = 8 * ((bx >> 3) / n_tiles) + (bx & 7);
    if (mt >= m_tiles) return;                           // whole workgroup, before any barrier
    const int n0 = ((bx >> 3) % n_tiles) * GM_BN;
    const int m0 = mt * GM_BM;

    // LDS-DMA: 16 A pieces + 16 W pieces per stage; wave w issues A pieces 4w..4w+3 and W pieces 4w..4w+3.
    const int lr = lane >> 3, lt = lane & 7;
    const char* Ab = reinterpret_cast<const char*>(A);
    const char* Wb = reinterpret_cast<const char*>(W);
    const int64_t rowbytes = (int64_t)K * 2;
    const char *a_src[4], *w_src[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int p = wave * 4 + i;
        const int r = 8 * p + lr;
        const int sw = (lt ^ ((r >> 1) & 7)) << 4;
        int ra = m0 + r; if (ra > M - 1) ra = M - 1;
        a_src[i] = Ab + (int64_t)ra * rowbytes + sw;
        w_src[i] = Wb + (int64_t)(n0 + r) * rowbytes + sw;     // N is a multiple of 128
    }
    auto issue = [&](int kt) {
        const int koff = kt * (GM_BK * 2);
        char* const stg = lds + (kt & (NST - 1)) * GM_STAGE;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int p = wave * 4 + i;
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)(a_src[i] + koff), (lds_ptr_t)(stg + p * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)(w_src[i] + koff), (lds_ptr_t)(stg + 16384 + p * 1024), 16, 0, 0);
        }
    };

    // acc[ni][mi]: rows (registers) = columns n of C, lane column = row m of C
    f32x16 acc[2][2];
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[ni][mi][i] = 0.0f;

    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
    const int sw = (col >> 1) & 7;
    const unsigned a_off0 = lds_base + (unsigned)((wm * 64 + col) * 128);
    const unsigned w_off0 = lds_base + (unsigned)(16384 + (wn * 64 + col) * 128);

    const int nkt = K / GM_BK;
    if constexpr (NST > 1) {
        for (int s = 0; s < NST - 1 && s < nkt; ++s) issue(s);
    }
    for (int kt = 0; kt < nkt; ++kt) {
        if constexpr (NST == 1) {
            issue(kt);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            // stage (kt + NST - 1) % NST held k-step kt - 1: everyone left it at the barrier that closed kt - 1
            if (kt + NST - 1 < nkt) issue(kt + NST - 1);
            // k-step kt has landed once at most the younger steps' loads (8 per step and wave) are outstanding
            const int younger = (nkt - 1 - kt < NST - 1) ? nkt - 1 - kt : NST - 1;
            if (younger >= 3) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
            else if (younger == 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            else if (younger == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        const unsigned a_off = a_off0 + (unsigned)((kt & (NST - 1)) * GM_STAGE);
        const unsigned w_off = w_off0 + (unsigned)((kt & (NST - 1)) * GM_STAGE);
        bf16x8 af[2][2], wf[2][2];       // [parity][block]: rows +0 and +32 (offset 4096 B)
        lds_read2(af[0][0], af[0][1], a_off + (((0 + h) ^ sw) << 4));
        lds_read2(wf[0][0], wf[0][1], w_off + (((0 + h) ^ sw) << 4));
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int c = s & 1;
            if (s < 3) {
                lds_read2(af[c ^ 1][0], af[c ^ 1][1], a_off + (((2 * (s + 1) + h) ^ sw) << 4));
                lds_read2(wf[c ^ 1][0], wf[c ^ 1][1], w_off + (((2 * (s + 1) + h) ^ sw) << 4));
                asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(af[c][0]), "+v"(af[c][1]), "+v"(wf[c][0]), "+v"(wf[c][1]));
            } else {
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af[c][0]), "+v"(af[c][1]), "+v"(wf[c][0]), "+v"(wf[c][1]));
            }
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
                    acc[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[c][ni], af[c][mi], acc[ni][mi], 0, 0, 0);
        }
        __builtin_amdgcn_s_barrier();                    // every wave's reads are retired (lgkmcnt(0) above)
    }

    // ---- epilogue, 64 rows (block mi of every wave) at a time: bias (+ GELU), park as T[r][n] with
    // r = 32 wm + col, then whole 256-B row segments out (thread t: row t / 16 + 16 p, chunk t % 16)
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        const int rl = wm * 32 + col;
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int nl = wn * 64 + ni * 32 + 8 * g + 4 * h;            // local column of registers 4g..4g+3
                const float4 bv = *reinterpret_cast<const float4*>(bias + n0 + nl);
                float v0 = acc[ni][mi][4 * g + 0] + bv.x, v1 = acc[ni][mi][4 * g + 1] + bv.y;
                float v2 = acc[ni][mi][4 * g + 2] + bv.z, v3 = acc[ni][mi][4 * g + 3] + bv.w;
                if constexpr (EPI == EPI_GELU) gelu_erf4(v0, v1, v2, v3);
                if constexpr (EPI == EPI_RESID) {
                    *reinterpret_cast<float4*>(lds + rl * GM_TS32 + nl * 4) = make_float4(v0, v1, v2, v3);
                } else {
                    *reinterpret_cast<uint2*>(lds + rl * GM_TS16 + nl * 2) = make_uint2(pack2(v0, v1), pack2(v2, v3));
                }
            }
        }
        __syncthreads();
        const int r0 = threadIdx.x >> 4, ch = threadIdx.x & 15;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int r = r0 + 16 * p;                                       // parked row: wm' = r / 32, col' = r % 32
            const int m = m0 + (r >> 5) * 64 + mi * 32 + (r & 31);
            if (m < M) {
                uint16_t* dst = C + (int64_t)m * N + n0 + 8 * ch;
                if constexpr (EPI == EPI_RESID) {
                    const float4 x0 = *reinterpret_cast<const float4*>(lds + r * GM_TS32 + ch * 32);
                    const float4 x1 = *reinterpret_cast<const float4*>(lds + r * GM_TS32 + ch * 32 + 16);
                    const uint4 rv = *reinterpret_cast<const uint4*>(resid + (int64_t)m * N + n0 + 8 * ch);
                    uint4 o;
                    o.x = pack2(x0.x + __builtin_bit_cast(float, rv.x << 16), x0.y + __builtin_bit_cast(float, rv.x & 0xFFFF0000u));
                    o.y = pack2(x0.z + __builtin_bit_cast(float, rv.y << 16), x0.w + __builtin_bit_cast(float, rv.y & 0xFFFF0000u));
                    o.z = pack2(x1.x + __builtin_bit_cast(float, rv.z << 16), x1.y + __builtin_bit_cast(float, rv.z & 0xFFFF0000u));
                    o.w = pack2(x1.z + __builtin_bit_cast(float, rv.w << 16), x1.w + __builtin_bit_cast(float, rv.w & 0xFFFF0000u));
                    *reinterpret_cast<uint4*>(dst) = o;
                } else {
                    *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(lds + r * GM_TS16 + ch * 16);
                }
            }
        }
        if (mi == 0) __syncthreads();
    }
}

// ---------------------------------------------------------------- weight-stationary GEMM, K = 384
// C[M][N] = A[M][384] * W[N][384]^T + bias (+ GELU) for the projections whose reduction is the hidden
// size and whose N is large (QKV, FFN1).  The 128x128 tile kernel above re-reads both operands
// from L2 for every tile (64 flop per LDS-staged byte).  This one is the flat-search scan turned
// into a GEMM:
//   * a workgroup of 8 waves owns 256 output columns; wave w keeps the W fragments of its 32
//     columns x 384 in registers (96 VGPRs) for its whole life;
//   * activation rows stream through LDS once per 256 columns, in units of 128 rows x 256 B by
//     LDS-DMA (SGPR base + 32-bit lane offset), ring of three, source-side swizzle;
//   * v_mfma_f32_16x16x32_bf16 with the W fragment as the A operand: a lane owns one output row
//     (lane & 15) and four consecutive columns per accumulator block, so bias / GELU are per-register;
//   * the workgroup barrier sits in the MIDDLE of a unit's chain and certifies the NEXT unit (as
//     in the scan); unit u + 2 is requested right behind it.  Unit and tile boundaries are not
//     synchronisation points, so the SIMD's two waves drift apart and one wave's epilogue (GELU:
//     as many VALU cycles as the tile has MFMA cycles) runs behind the other's MFMAs;
//   * the epilogue is wave-local: the wave parks its 128 x 32 slab (8 KB, swizzled) in its own LDS
//     region and writes it out itself right behind the NEXT barrier, so the stores have a whole
//     unit to retire before the following vmcnt(0).  BLK: C is [M / 128][N / 32][128][32] (what
//     gemm_ln_kernel stages as whole 8-KB blocks) and a wave's slab is one contiguous block;
//   * one workgroup per CU, placed by hand: the 32 workgroups of an XCD (blockIdx % 8) form
//     floor(32 / G) row chunks x G column groups; the G groups of a chunk walk the same row
//     tiles in step, so each activation tile comes from HBM once per XCD.
// Waves whose columns lie beyond N only help with the loads.
#define WS_TILE_R 128
#define WS_STAGE (WS_TILE_R * 256)
#define WS_PARK_OFF (3 * WS_STAGE)
#define WS_LDS_BYTES (WS_PARK_OFF + 8 * 8192)
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int EPI, bool BLK>
__global__ __launch_bounds__(512) void gemm_ws_kernel(const uint16_t* __restrict__ A, const uint16_t* __restrict__ W,
        const float* __restrict__ bias, uint16_t* __restrict__ C, int M, int N, int n_full, int chunks_full, int chunks_tail)
{
    constexpr int K = 384, NKB = 3, PPW = 4;
    static_assert(EPI == EPI_BIAS || EPI == EPI_GELU, "the residual epilogue lives in gemm_ln_kernel / gemm_bf16_kernel");
    __shared__ __attribute__((aligned(1024))) char lds[WS_LDS_BYTES];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l15 = lane & 15, g4 = lane >> 4;

    // placement: XCD = blockIdx % 8 (round-robin dispatch); k-th workgroup of the XCD
    const int xcd = blockIdx.x & 7, kx = blockIdx.x >> 3;
    // n_full column groups of 256 get chunks_full row chunks each; a last, narrower group (N % 256
    // columns, some waves idle) gets chunks_tail: fewer, longer chunks, so that its workgroups finish
    // with the others instead of leaving their CUs half used
    if (kx >= n_full * chunks_full + chunks_tail) return;          // spare CUs of this XCD
    const bool tail = kx >= n_full * chunks_full;
    const int chunk = tail ? kx - n_full * chunks_full : kx / n_full;
    const int ng = tail ? n_full : kx % n_full;
    const int n_chunks = tail ? chunks_tail : chunks_full;
    const int m_tiles = (M + WS_TILE_R - 1) / WS_TILE_R;
    const int tiles_x = (m_tiles - xcd + 7) / 8;                   // row tiles xcd, xcd + 8, ... of this XCD
    const int t_lo = (int)((int64_t)tiles_x * chunk / n_chunks), t_hi = (int)((int64_t)tiles_x * (chunk + 1) / n_chunks);
    const int ntiles = t_hi - t_lo;
    const int nunits = ntiles * NKB;
    const int n0 = ng * 256;
    const bool active = n0 + wave * 32 < N;                        // wave-uniform (N is a multiple of 32)

    // ---- stationary W fragments: lane (l15, g4) holds rows n0 + 32 w + 16 nb + l15, k = 32 s + 8 g4 + j
    bf16x8 wq[12][2];
    if (active) {
#pragma unroll
        for (int s = 0; s < 12; ++s)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
                wq[s][nb] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(
                    W + (int64_t)(n0 + wave * 32 + 16 * nb + l15) * K + 32 * s + 8 * g4));
#pragma unroll
        for (int s = 0; s < 12; ++s) asm volatile("" :: "v"(wq[s][0]), "v"(wq[s][1]));   // retire before the loop (see search_mfma.hip)
    }
    float4 bv[2] = {make_float4(0, 0, 0, 0), make_float4(0, 0, 0, 0)};
    if (active) {
        bv[0] = *reinterpret_cast<const float4*>(bias + n0 + wave * 32 + 4 * g4);
        bv[1] = *reinterpret_cast<const float4*>(bias + n0 + wave * 32 + 16 + 4 * g4);
    }
    asm volatile("s_waitcnt vmcnt(0)" :: "v"(bv[0].x), "v"(bv[1].x) : "memory");

    // ---- LDS-DMA: wave w issues pieces 4w..4w+3 of a unit (4 rows x 256 B each); unit (tile, kb) -> ring buffer kb
    const int lr = lane >> 4, lt = lane & 15;
    auto tile_row0 = [&](int tl) { return (int64_t)(8 * (t_lo + tl) + xcd) * WS_TILE_R; };
    const char* Ab = reinterpret_cast<const char*>(A) + (int64_t)(4 * wave * PPW) * (K * 2);
    const unsigned voff0 = (unsigned)lr * (K * 2) + ((unsigned)(lt ^ lr) << 4);
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
    auto issue_unit = [&](int tl, int kb) __attribute__((always_inline)) {
        const char* ub = Ab + (tile_row0(tl) * (K * 2) + kb * 256);              // rows past M: workspace slack, never stored
#pragma unroll
        for (int i = 0; i < PPW; ++i)
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                         :: "v"(voff0 ^ ((unsigned)(i & 3) << 6)), "s"(ub + (int64_t)i * 4 * (K * 2)),
                            "s"(lds_base + kb * WS_STAGE + (wave * PPW + i) * 1024) : "memory", "m0");
    };

    f32x4 acc[8][2];                                               // [16-row block of the tile][16-column block of the wave]
    auto chain = [&](auto KBc, auto&& mid) __attribute__((always_inline)) {
        constexpr int KB = decltype(KBc)::value;
        const unsigned cur = (lds_base + KB * WS_STAGE + l15 * 256) | ((unsigned)(g4 ^ l15) << 4);
        bf16x8 a[2][4];
        auto rd = [&](bf16x8 (&f)[4], int grp) __attribute__((always_inline)) {
            const unsigned ad = cur ^ ((unsigned)(grp >> 1) << 6);
            if (grp & 1)
                asm volatile("ds_read_b128 %0, %4 offset:16384\n\tds_read_b128 %1, %4 offset:20480\n\t"
                             "ds_read_b128 %2, %4 offset:24576\n\tds_read_b128 %3, %4 offset:28672"
                             : "=&v"(f[0]), "=&v"(f[1]), "=&v"(f[2]), "=&v"(f[3]) : "v"(ad));
            else
                asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:4096\n\t"
                             "ds_read_b128 %2, %4 offset:8192\n\tds_read_b128 %3, %4 offset:12288"
                             : "=&v"(f[0]), "=&v"(f[1]), "=&v"(f[2]), "=&v"(f[3]) : "v"(ad));
        };
        rd(a[0], 0);
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            if (t < 7) {
                rd(a[(t + 1) & 1], t + 1);
                asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(a[t & 1][0]), "+v"(a[t & 1][1]), "+v"(a[t & 1][2]), "+v"(a[t & 1][3]));
            } else {
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[t & 1][0]), "+v"(a[t & 1][1]), "+v"(a[t & 1][2]), "+v"(a[t & 1][3]));
            }
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)             // the tile's first MFMA per accumulator takes C = 0: no clearing pass
                    acc[4 * (t & 1) + r][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                        wq[KB * 4 + (t >> 1)][nb], a[t & 1][r], (KB == 0 && t < 2) ? f32x4{} : acc[4 * (t & 1) + r][nb], 0, 0, 0);
            if (t == 3) mid();
        }
    };

    // ---- epilogue, phase 1 (end of a tile): bias (+ GELU), bf16, into the wave's own slab: row r at r * 64,
    // 8-byte slot (4 nb + g4) ^ 2 ((r >> 2) & 3)  (16 rows x 8 B per ds_write group: all 32 bank pairs once)
    // (r >> 2) & 3 does not depend on the 16-row block, so every address is one lane constant plus an
    // immediate: as separate loop invariants the 24 addresses are hoisted and spilled.
    const unsigned park = lds_base + WS_PARK_OFF + wave * 8192;
    unsigned pk_w0 = park + l15 * 64 + (((unsigned)g4 ^ ((((unsigned)l15 >> 2) & 3) << 1)) << 3);       // nb = 0; nb = 1: ^ 32
    auto park_tile = [&]() __attribute__((always_inline)) {
        static_for<8>([&](auto MBc) __attribute__((always_inline)) {
            constexpr int mb = decltype(MBc)::value;
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                const f32x4 v = acc[mb][nb];
                float v0 = v[0] + bv[nb].x, v1 = v[1] + bv[nb].y, v2 = v[2] + bv[nb].z, v3 = v[3] + bv[nb].w;
                if constexpr (EPI == EPI_GELU) gelu_erf4(v0, v1, v2, v3);
                const uint2 o = make_uint2(pack2(v0, v1), pack2(v2, v3));
                const unsigned adw = pk_w0 ^ (nb ? 32u : 0u);
                asm volatile("ds_write_b64 %0, %1 offset:%2" :: "v"(adw), "v"(o), "n"(mb * 1024) : "memory");
            }
        });
    };
    // ---- phase 2 (behind the next tile's first unit): the slab leaves as 16-byte chunks, lane -> row lane / 4 + 16 p, chunk lane % 4
    auto store_tile = [&](int tl) __attribute__((always_inline)) {
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        const int64_t m_base = tile_row0(tl);
        const unsigned rr = (unsigned)lane >> 2, c = (unsigned)lane & 3;
        unsigned pk_r = park + rr * 64 + ((c ^ ((rr >> 2) & 3)) << 4);
        // one contiguous 8-KB block when BLK; rows past M lie in the padded last tile of the blocked buffer
        // wave-uniform base + a lane offset formed here (as loop invariants the 64-bit lane pointers get spilled)
        unsigned l4;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0\n\tv_lshlrev_b32 %0, 2, %0" : "=v"(l4));
        const uint16_t* const ubase = BLK ? C + ((m_base >> 7) * (N >> 5) + (n0 >> 5) + wave) * 4096
                                          : C + m_base * N + n0 + wave * 32;
        uint16_t* const blk = const_cast<uint16_t*>(ubase) + (BLK ? (size_t)(l4 * 2)                       // lane * 8 elements
                                                                  : (size_t)((l4 >> 4) * (unsigned)N + ((l4 >> 2) & 3u) * 8u));
        static_for<4>([&](auto Pc) __attribute__((always_inline)) {
            constexpr int p = 2 * decltype(Pc)::value;             // two chunks at a time
            u32x4 o0, o1;
            const unsigned adr = pk_r;
            asm volatile("ds_read_b128 %0, %2 offset:%3\n\tds_read_b128 %1, %2 offset:%4\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(o0), "=&v"(o1) : "v"(adr), "n"(p * 1024), "n"((p + 1) * 1024) : "memory");
            if constexpr (BLK) {
                *reinterpret_cast<u32x4*>(blk + p * 512) = o0;
                *reinterpret_cast<u32x4*>(blk + (p + 1) * 512) = o1;
            } else {
                if (m_base + rr + 16 * p < M) *reinterpret_cast<u32x4*>(blk + (int64_t)16 * p * N) = o0;
                if (m_base + rr + 16 * (p + 1) < M) *reinterpret_cast<u32x4*>(blk + (int64_t)16 * (p + 1) * N) = o1;
            }
        });
    };

    // ---- main loop.  At the barrier of unit u: unit u + 1 has landed for everyone (it was requested
    // behind the barrier of unit u - 1), everyone has left unit u - 1, whose buffer takes unit u + 2.
    // vmcnt(0) also retires the previous tile's stores; they were issued one whole unit earlier.
    if (nunits > 0) {
        issue_unit(0, 0);
        if (nunits > 1) issue_unit(1 / NKB, 1 % NKB);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    int u = 0;
    for (int tl = 0; tl < ntiles; ++tl) {
        static_for<NKB>([&](auto KBc) __attribute__((always_inline)) {
            constexpr int kb = decltype(KBc)::value;
            auto mid = [&]() __attribute__((always_inline)) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                if (u + 2 < nunits) issue_unit(tl + (kb + 2) / NKB, (kb + 2) % NKB);
            };
            if (active) chain(IC<kb>{}, mid); else mid();
            // the previous tile's slab leaves behind the first unit of this one (inside the chain there is no register to spare)
            if constexpr (kb == 0) { if (active && tl > 0) store_tile(tl - 1); }
            if constexpr (kb == NKB - 1) { if (active) park_tile(); }
            ++u;
        });
    }
    if (active && ntiles > 0) store_tile(ntiles - 1);
}

#ifdef FF_TRACE
// Debug build only (tools/build_trace.sh): cycle stamps of the phases of eight workgroups of ffn2_kernel.
__device__ uint32_t ff_trace_buf[8][8][48];
#define FF_STAMP(k) do { if (ff_slot >= 0 && lane == 0) ff_trace_buf[ff_slot][wave][k] = (uint32_t)__builtin_amdgcn_s_memtime(); } while (0)
// stamps INSIDE one stage (FF_TRACE_KT): s_memtime into SGPRs without a wait (a wait would drain the fragment reads in
// flight and change what is measured); written out behind the next stage's own lgkmcnt(0)
#ifndef FF_TRACE_KT
#define FF_TRACE_KT 20
#endif
// (fixed SGPR pairs s60.. above what the allocator uses: the value lands long after the instruction, so it must not be a
// register the compiler may copy or reuse)
#define FF_TICK(k) asm volatile("s_memtime s[%0:%1]" :: "n"(74 + 2 * (k)), "n"(75 + 2 * (k)) : "memory", "s101")
#define FF_TICK_RD(k, dst) asm volatile("v_mov_b32 %0, s[%1]" : "=v"(dst) : "n"(74 + 2 * (k)))
#define FF_TICKS_OUT_AT(base, cond) do { if ((cond) && ff_slot >= 0) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
    static_for<14>([&](auto Kc) __attribute__((always_inline)) { uint32_t _v; FF_TICK_RD(decltype(Kc)::value, _v); \
        if (lane == 0) ff_trace_buf[ff_slot][wave][(base) + decltype(Kc)::value] = _v; }); } } while (0)
#define FF_TICKS_OUT(stage) FF_TICKS_OUT_AT(16, (stage) == FF_TRACE_KT)
// the stages of the front GEMM (rowgemm384_mainloop, K = 384: twelve): tick kt at the top of stage kt
#define FF_TICK_STAGE(kt) do { switch (kt) { case 0: FF_TICK(0); break; case 1: FF_TICK(1); break; case 2: FF_TICK(2); break; \
    case 3: FF_TICK(3); break; case 4: FF_TICK(4); break; case 5: FF_TICK(5); break; case 6: FF_TICK(6); break; case 7: FF_TICK(7); break; \
    case 8: FF_TICK(8); break; case 9: FF_TICK(9); break; case 10: FF_TICK(10); break; case 11: FF_TICK(11); break; default: break; } } while (0)
#else
#define FF_STAMP(k) do { } while (0)
#define FF_TICK(k) do { } while (0)
#define FF_TICKS_OUT(stage) do { } while (0)
#define FF_TICKS_OUT_AT(base, cond) do { } while (0)
#define FF_TICK_STAGE(kt) do { } while (0)
#endif

#define LN_N 384
#define LN_TS 784                            // parked bf16 row: 384 columns + 8 pad
// ---- shared epilogue of the row-complete kernels: bias + residual, LayerNorm over the 384 columns, bf16 rows out.
// The lane holds, for rows m = m0 + 32 wm + 16 mb + l15, columns 192 wn + 16 nb + 4 g4 + (0..3).
// lds: >= 4096 + 128 * LN_TS bytes of scratch nobody else touches any more; prm: bias | gamma | beta (fp32, LDS).
template <bool STORE = true>
__device__ __forceinline__ void ln_rows_epilogue(char* lds, const float* prm, f32x4 (&acc)[2][12], const uint2 (&rres)[2][12],
        uint16_t* __restrict__ out, int m0, int M, float eps, int wm, int wn, int l15, int g4)
{
    float* rs = reinterpret_cast<float*>(lds);                 // [2 passes][2 column halves][128 rows] row sums
    char* park = lds + 4096;
    float mean[2], inv[2];
    // bias + residual (8-byte loads: 4 consecutive columns of the lane's row), row sum
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
        const int ml = 32 * wm + 16 * mb + l15;
        float sum = 0.0f;
#pragma unroll
        for (int nb = 0; nb < 12; ++nb) {
            const int n = 192 * wn + 16 * nb + 4 * g4;
            const float4 bv = *reinterpret_cast<const float4*>(prm + n);
            const uint2 rv = rres[mb][nb];
            acc[mb][nb][0] += bv.x + __builtin_bit_cast(float, rv.x << 16);
            acc[mb][nb][1] += bv.y + __builtin_bit_cast(float, rv.x & 0xFFFF0000u);
            acc[mb][nb][2] += bv.z + __builtin_bit_cast(float, rv.y << 16);
            acc[mb][nb][3] += bv.w + __builtin_bit_cast(float, rv.y & 0xFFFF0000u);
            sum += (acc[mb][nb][0] + acc[mb][nb][1]) + (acc[mb][nb][2] + acc[mb][nb][3]);
        }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        if (g4 == 0) rs[wn * 128 + ml] = sum;
    }
    __syncthreads();
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
        const int ml = 32 * wm + 16 * mb + l15;
        mean[mb] = (rs[ml] + rs[128 + ml]) * (1.0f / LN_N);
        float sq = 0.0f;
#pragma unroll
        for (int nb = 0; nb < 12; ++nb)
#pragma unroll
            for (int i = 0; i < 4; ++i) { const float dlt = acc[mb][nb][i] - mean[mb]; sq += dlt * dlt; }
        sq += __shfl_xor(sq, 16, 64);
        sq += __shfl_xor(sq, 32, 64);
        if (g4 == 0) rs[256 + wn * 128 + ml] = sq;
    }
    __syncthreads();
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
        const int ml = 32 * wm + 16 * mb + l15;
        inv[mb] = 1.0f / sqrtf((rs[256 + ml] + rs[384 + ml]) * (1.0f / LN_N) + eps);
    }
#pragma unroll
    for (int nb = 0; nb < 12; ++nb) {
        const int n = 192 * wn + 16 * nb + 4 * g4;
        const float4 gv = *reinterpret_cast<const float4*>(prm + LN_N + n);
        const float4 be = *reinterpret_cast<const float4*>(prm + 2 * LN_N + n);
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
            const int ml = 32 * wm + 16 * mb + l15;
            const float y0 = (acc[mb][nb][0] - mean[mb]) * inv[mb] * gv.x + be.x, y1 = (acc[mb][nb][1] - mean[mb]) * inv[mb] * gv.y + be.y;
            const float y2 = (acc[mb][nb][2] - mean[mb]) * inv[mb] * gv.z + be.z, y3 = (acc[mb][nb][3] - mean[mb]) * inv[mb] * gv.w + be.w;
            *reinterpret_cast<uint2*>(park + ml * LN_TS + n * 2) = make_uint2(pack2(y0, y1), pack2(y2, y3));
        }
    }
    __syncthreads();
    if constexpr (!STORE) return;                              // the caller goes on with the parked tile (lds + 4096, rows of LN_TS bytes)
    // whole 768-B rows out: thread t takes 16-B chunk t % 48 of rows t / 48 + 10 p (480 of the 512 threads)
    if (threadIdx.x < 480) {
        const int r0 = threadIdx.x / 48, ch = threadIdx.x % 48;
        for (int r = r0; r < 128; r += 10) {
            const int m = m0 + r;
            if (m < M) *reinterpret_cast<uint4*>(out + (int64_t)m * LN_N + 8 * ch) = *reinterpret_cast<const uint4*>(park + r * LN_TS + ch * 16);
        }
    }
}

// ---------------------------------------------------------------- row-complete GEMM + bias + residual + LayerNorm, N = 384
// out[M][384] = LayerNorm(A[M][K] * W[384][K]^T + bias + resid) * gamma + beta   (modeling_bert.py:289-293, :347-351)
// for the attention output (K = 384) and FFN output (K = 1536) projections.  One workgroup of
// 8 waves owns 128 COMPLETE rows (4 row groups x 2 column halves: a wave holds 32 rows x 192
// columns = 24 accumulator blocks of 16x16), so the LayerNorm statistics never leave the
// workgroup, the separate LN kernel and the [tokens][384] round trip through HBM disappear, and
// the activation tile is staged once per 384 columns instead of once per 128 (96 flop per staged
// byte instead of 64).  v_mfma_f32_16x16x32_bf16 with the W fragment as the A operand: a lane owns
// row (lane & 15) of a block and four consecutive columns, so bias / gamma / beta are float4 and
// the finished tile is parked in LDS and leaves as whole 768-B rows.  Stages of 32 k (A 8 KB +
// W 24 KB) by LDS-DMA in a ring of four, requested three stages ahead.
// `out` may alias `resid`: a workgroup reads and writes only its own rows.
#define LN_STAGE (512 * 64)                  // 128 A rows + 384 W rows, 64 B (32 k) each

// The main loop of the row-complete GEMM: acc = A[m0 .. m0 + 127][K] * W^T for all 384 columns, the residual rows
// requested on the way (rres), bias / gamma / beta copied to LDS behind the ring.  Ends with a workgroup
// barrier: the ring is free.  Shared by gemm_ln_kernel and the front phase of ffn2_kernel.
__device__ __forceinline__ void rowgemm384_mainloop(char* lds, const uint16_t* __restrict__ A, const uint16_t* __restrict__ W,
        const float* __restrict__ bias, const uint16_t* __restrict__ resid, const float* __restrict__ gamma,
        const float* __restrict__ beta, int M, int K, int a_blocked, int m0, int lane, int wave,
        f32x4 (&acc)[2][12], uint2 (&rres)[2][12])
{
    const int l15 = lane & 15, g4 = lane >> 4;
    const int wm = wave >> 1, wn = wave & 1;
    // LDS-DMA.  A stage is 32 k: 512 rows (128 A + 384 W) x 64 B = 32 KB, ring of four.  (The first form
    // had two 64-KB stages: the next stage could only be requested one stage ahead, and a stage took
    // 2.6 us against 0.75 us of MFMA -- the L2 -> LDS latency, not the arithmetic, set the pace.)
    // A piece is one DMA instruction: 16 rows x 64 B, lane l -> row l / 4, 16-B slot l % 4.
    // Slot = chunk ^ swz(row), swz(r) = (0 - (r >> 2)) & 3: with that the four 16-lane groups a
    // ds_read_b128 is served in (rows {0-3, 12-15} of one k-chunk and rows 4-11 of the next) each
    // touch all 64 banks once.  Wave w issues pieces 4w..4w+3 (waves 0, 1: the A rows).
    const int nkt = K / 32;                                        // even (K is a multiple of 64), >= 12
    // bias / gamma / beta live in LDS for the epilogue (36 dependent float4 global loads there cost microseconds per tile): one
    // 16-B load per thread < 288, the OLDEST request of the wave, stored behind the first wait below.
    float* const prm = reinterpret_cast<float*>(lds + 4 * LN_STAGE);
    const bool has_p = threadIdx.x < 3 * LN_N / 4;
    f32x4 pv = {0.0f, 0.0f, 0.0f, 0.0f};
    if (has_p) {
        const int which = threadIdx.x / (LN_N / 4), o = (threadIdx.x % (LN_N / 4)) * 4;
        const float* srcp = (which == 0 ? bias : which == 1 ? gamma : beta) + o;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(pv) : "v"(srcp));
    }
    // The residual tile is 128 whole rows = 96 KB contiguous: 12 coalesced 16-B loads per thread (chunk c = t + 512 i, row c / 48),
    // requested behind the first three stages, in registers through the main loop, and turned into the epilogue's layout
    // through LDS at the end.  (Round 1 fetched it in the epilogue's layout, 24 loads of 8 B per lane from 16 rows each,
    // behind stage 0: stamps showed the waves ~7,000 cycles in stage 1 issuing them -- front GEMM 29,000 -> 24,800 cycles.
    // Requested behind stage 2 instead, the 12 loads stall stage 3 for 5,000 cycles: 26,500.)
    u32x4 rt[12];
    auto prefetch_resid = [&]() __attribute__((always_inline)) {
        int row = threadIdx.x / 48, cc = threadIdx.x % 48;
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            int m = m0 + row; if (m > M - 1) m = M - 1;
            const uint16_t* rp = resid + (int64_t)m * LN_N + cc * 8;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(rt[i]) : "v"(rp));
            row += 10; cc += 32;                                   // + 512 chunks = 10 rows + 32 chunks
            if (cc >= 48) { cc -= 48; row += 1; }
        }
    };
    const int64_t rowbytes = (int64_t)K * 2;
    const char* src[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = 16 * (4 * wave + i) + (lane >> 2);                          // stage row: 0..127 A, 128..511 W
        const int ch = (lane & 3) ^ ((0 - (lane >> 4)) & 3);
        if (wave < 2) {
            if (a_blocked) {                                       // A is [M / 128][K / 32][128][32]: a stage is one 8-KB block
                src[i] = reinterpret_cast<const char*>(A) + ((int64_t)(m0 >> 7) * (K >> 5) * 128 + r) * 64 + ch * 16;
            } else {
                int ra = m0 + r; if (ra > M - 1) ra = M - 1;
                src[i] = reinterpret_cast<const char*>(A) + ra * rowbytes + ch * 16;
            }
        }
        else src[i] = reinterpret_cast<const char*>(W) + (int64_t)(r - 128) * 64 + ch * 16;          // W is packed [K / 32][384][32]
    }
    const int kstride = (wave < 2) ? (a_blocked ? 128 * 64 : 64) : LN_N * 64;               // A: 32 k further along the row; W: the next packed block
    auto issue = [&](int kt) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)(src[i] + (int64_t)kt * kstride),
                (lds_ptr_t)(lds + (kt & 3) * LN_STAGE + (4 * wave + i) * 1024), 16, 0, 0);
    };

#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int nb = 0; nb < 12; ++nb)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[mb][nb][i] = 0.0f;

    // fragment addresses: stage row r = (multiple of 16) + l15, 16-B chunk g4 ^ swz(l15): constant per lane
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
    const unsigned fsw = (unsigned)(g4 ^ ((0 - (l15 >> 2)) & 3)) << 4;
    const unsigned a_adr = lds_base + (unsigned)((32 * wm + l15) * 64) + fsw;             // + mb * 1024
    const unsigned w_adr = lds_base + (unsigned)((128 + 192 * wn + l15) * 64) + fsw;      // + nb * 1024

    // A stage is three groups of 4 W fragments x 2 A fragments = 8 MFMAs; fragments are read one group
    // ahead, ACROSS stage boundaries.  The workgroup barrier sits behind the second group and
    // certifies the NEXT stage (landed for everyone; everyone has left the previous one), so a
    // stage boundary is no synchronisation point and stage kt + 3 is requested right behind it.
    bf16x8 af[2][2], wf[2][4];
    auto rd_w = [&](bf16x8 (&f)[4], int kt, int j) __attribute__((always_inline)) {
        const unsigned ad = w_adr + (kt & 3) * LN_STAGE + j * 4096;
        asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:1024\n\t"
                     "ds_read_b128 %2, %4 offset:2048\n\tds_read_b128 %3, %4 offset:3072"
                     : "=&v"(f[0]), "=&v"(f[1]), "=&v"(f[2]), "=&v"(f[3]) : "v"(ad));
    };
    auto rd_a = [&](bf16x8 (&f)[2], int kt) __attribute__((always_inline)) {
        const unsigned ad = a_adr + (kt & 3) * LN_STAGE;
        asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:1024" : "=&v"(f[0]), "=&v"(f[1]) : "v"(ad));
    };
    auto mma = [&](bf16x8 (&w4)[4], bf16x8 (&a2)[2], int j) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
                acc[mb][4 * j + q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w4[q], a2[mb], acc[mb][4 * j + q], 0, 0, 0);
    };
    // PAR = parity of the stage's first group (3 groups per stage: it alternates)
    auto stage = [&](int kt, auto PARc) __attribute__((always_inline)) {
        constexpr int P = decltype(PARc)::value;
        FF_TICK_STAGE(kt);
        bf16x8 (&a2)[2] = af[P];                                   // the stage's A fragments arrived with its group 0
        rd_w(wf[P ^ 1], kt, 1);
        asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(wf[P][0]), "+v"(wf[P][1]), "+v"(wf[P][2]), "+v"(wf[P][3]), "+v"(a2[0]), "+v"(a2[1]));
        mma(wf[P], a2, 0);
        rd_w(wf[P], kt, 2);
        asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(wf[P ^ 1][0]), "+v"(wf[P ^ 1][1]), "+v"(wf[P ^ 1][2]), "+v"(wf[P ^ 1][3]));
        mma(wf[P ^ 1], a2, 1);
        // in flight behind stage kt + 1: stage kt + 2 (4 loads) and, in stages 0 and 1, the 12 residual loads (requested behind
        // stage 2, older than stage 3: stage 2's wait retires them)
        if (kt < 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else if (kt + 2 < nkt) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (kt + 3 < nkt) issue(kt + 3);
        if (kt + 1 < nkt) {
            rd_a(af[P ^ 1], kt + 1);
            rd_w(wf[P ^ 1], kt + 1, 0);
            asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(wf[P][0]), "+v"(wf[P][1]), "+v"(wf[P][2]), "+v"(wf[P][3]));
        } else {
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(wf[P][0]), "+v"(wf[P][1]), "+v"(wf[P][2]), "+v"(wf[P][3]));
        }
        mma(wf[P], a2, 2);
    };

    issue(0); issue(1); issue(2);                                  // (nkt >= 12: K is a multiple of 64 and >= 384)
    prefetch_resid();
    asm volatile("s_waitcnt vmcnt(20)" : "+v"(pv) :: "memory");     // stage 0 (and the parameter load before it); in flight: stages 1, 2, residual
    if (has_p) *reinterpret_cast<f32x4*>(prm + (threadIdx.x / (LN_N / 4)) * LN_N + (threadIdx.x % (LN_N / 4)) * 4) = pv;
    __builtin_amdgcn_s_barrier();
    rd_a(af[0], 0);
    rd_w(wf[0], 0, 0);
    for (int kt = 0; kt < nkt; kt += 2) {
        stage(kt, IC<0>{});
        stage(kt + 1, IC<1>{});
    }
    __syncthreads();                                               // the ring becomes the epilogue's scratch
    // the residual tile: registers (contiguous chunks) -> LDS rows of LN_TS bytes -> the epilogue's layout
    {
        char* const park = lds + 4096;
        int row = threadIdx.x / 48, cc = threadIdx.x % 48;
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            asm volatile("" : "+v"(rt[i]));
            *reinterpret_cast<u32x4*>(park + row * LN_TS + cc * 16) = rt[i];
            row += 10; cc += 32;
            if (cc >= 48) { cc -= 48; row += 1; }
        }
        __syncthreads();
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int nb = 0; nb < 12; ++nb)
                rres[mb][nb] = *reinterpret_cast<const uint2*>(park + (32 * wm + 16 * mb + l15) * LN_TS + (192 * wn + 16 * nb + 4 * g4) * 2);
        __syncthreads();                                           // (the epilogue parks its output in the same rows)
    }
}


__global__ __launch_bounds__(512) void gemm_ln_kernel(const uint16_t* __restrict__ A, const uint16_t* __restrict__ W,
        const float* __restrict__ bias, const uint16_t* __restrict__ resid, const float* __restrict__ gamma,
        const float* __restrict__ beta, float eps, uint16_t* __restrict__ out, int M, int K, int a_blocked)
{
    __shared__ __attribute__((aligned(1024))) char lds[4 * LN_STAGE + 3 * LN_N * 4];   // ring | bias, gamma, beta (fp32)
    static_assert(4 * LN_STAGE >= 4096 + 128 * LN_TS, "the epilogue parks the tile in the ring");
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l15 = lane & 15, g4 = lane >> 4;
    const int wm = wave >> 1, wn = wave & 1;
    // last rows first: the producer kernel wrote them last, so they are the part of A still in the Infinity Cache
    const int m0 = (int)(gridDim.x - 1 - blockIdx.x) * 128;

    f32x4 acc[2][12];                                  // [16-row block][16-column block]
    uint2 rres[2][12];
    rowgemm384_mainloop(lds, A, W, bias, resid, gamma, beta, M, K, a_blocked, m0, lane, wave, acc, rres);
    const float* const prm = reinterpret_cast<const float*>(lds + 4 * LN_STAGE);
    ln_rows_epilogue(lds, prm, acc, rres, out, m0, M, eps, wm, wn, l15, g4);
}

// ---------------------------------------------------------------- fused FFN: LayerNorm(h + W2 gelu(W1 h + b1) + b2), hidden 384
// out[M][384] = LayerNorm(h + gelu(h W1^T + b1) W2^T + b2) * gamma + beta   (modeling_bert.py:334-351)
// The [tokens][1536] intermediate never exists: a workgroup owns 128 rows and walks the
// intermediate dimension 32 columns at a time.  Per step j ("stage"):
//   step 1 (wave w, rows 16 w .. 16 w + 15): mid_j = gelu(h W1_j^T + b1_j), 16 x 32 per wave, h fragments
//           register-resident (48 VGPRs), W1_j fragments from LDS; bf16 into the A buffer (LDS);
//   step 2 (waves as 4 row groups x 2 column halves, as in gemm_ln_kernel): acc += mid_j W2_j^T.
// Step 1 runs two stages ahead of step 2, its GELU one.  Both weight blocks are 24 KB, contiguous in HBM
// (packed at load time; W1 already in its swizzled LDS image), LDS-DMA into rings of three, requested two
// stages before use.  LDS: 3 x 24 (W2) + 3 x 24 (W1) + 8 (A) + 6 (b1) KB.
// Against gemm_ws<GELU> + gemm_ln this saves the 2 x 200 MB round trip of the intermediate per 64K
// tokens and one kernel's prologue/epilogue per row tile.  `out` may alias `h`.
#define FF_WS (384 * 64)                     // one weight block: 24 KB
#define FF_W1_OFF (3 * FF_WS)
#define FF_A_OFF (6 * FF_WS)
#define FF_B1_OFF (FF_A_OFF + 8192)

// fp32 [I][384] -> bf16 [I / 32][12 k-steps][32 rows][4 chunks of 8], chunk g of row r stored at g ^ ((0 - (r >> 2)) & 3)
__global__ __launch_bounds__(256) void cvt_pack_w1_kernel(const float* __restrict__ in, uint16_t* __restrict__ out, int I)
{
    const int64_t total = (int64_t)I * 384;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int e = (int)(i & 7), gs = (int)((i >> 3) & 3), r = (int)((i >> 5) & 31);
        const int64_t t = i >> 10;
        const int sk = (int)(t % 12), j = (int)(t / 12);
        const int g = gs ^ ((0 - (r >> 2)) & 3);
        out[i] = f2bf(in[(int64_t)(32 * j + r) * 384 + 32 * sk + 8 * g + e]);
    }
}


// ---------------------------------------------------------------- fused FFN, second form (round 2): step 1 split over K
// Round 1's fused FFN (ffn_ln_kernel: step 1 by 16-row strips, rings of three per weight; retired in round 3, git history
// and DESIGN.md 4.5 keep it) with another division of labour.  Cycle stamps of that first form (profiles/r02_ffn_trace.md):
// a 3,640-cycle stage against 1,536 cycles of matrix pipe per SIMD -- 470 cycles without any MFMA between its two
// barriers (fragment reads that could not be requested earlier, a vmcnt(0) hipcc put in front of a plain LDS load of
// b1), six 4-MFMA groups whose fragments (24 ds_read_b128 per wave: every wave read the whole W1 block for its 16 rows)
// arrive no faster than ~170 cycles per group, the GELU of a block as one VALU run with the wave's MFMAs standing
// still, and the younger wave of each SIMD 700 cycles behind the older at every barrier.  Here:
//   * wave (rg, kh) = (wave >> 1, wave & 1) multiplies rows 32 rg .. + 31 (two row blocks) by K half kh of W1_j: 12
//     fragment reads instead of 24, every fragment feeds two MFMAs, three groups of 8 MFMAs like step 2.  The
//     partial sums of the partner's row block go through LDS (X, 2 KB per wave), the wave adds the partner's
//     partial to its own row block (rows 16 wave .. + 15), bias, GELU, bf16 -> A as before;
//   * one ring of FIVE 24-KB slots holds W2 and W1 blocks in the order they are consumed
//     (W1_0 W1_1 | W2_0 W1_2 | W2_1 W1_3 | ...), a block is requested when the slot five back is free;
//   * two barriers per stage, behind groups 0 and 3, each certifying what is needed one group LATER (fragments are
//     read one group ahead across them), so neither is followed by a dependent LDS read;
//   * every LDS access of the loop is inline asm (hipcc waits for every LDS-DMA in flight before an LDS load it can see);
//   * the steady-state stage has no branches; the last four stages are separate instantiations.
// LDS: 5 x 24 (ring) + 8 (A) + 16 (X) + 6 (b1) KB.
#define F2_SLOTS 5
#define F2_RING (F2_SLOTS * FF_WS)
#define F2_A_OFF F2_RING
#define F2_X_OFF (F2_A_OFF + 8192)
#define F2_B1_OFF (F2_X_OFF + 8 * 2048)
#define F2_LDS_BYTES (F2_B1_OFF + 1536 * 4)
#define F2_NKT 48                            // intermediate size 1536

// FRONT: the attention output projection + residual + LayerNorm (what gemm_ln_kernel does for K = 384) runs first, on the
// same 128 rows: h is then the layer's input (the residual of that LayerNorm), its result never goes to HBM -- it is
// parked in LDS, becomes the h fragments of step 1 and the residual of the final LayerNorm.
template <bool FRONT>
__global__ __launch_bounds__(512) void ffn2_kernel(const uint16_t* __restrict__ h, const uint16_t* __restrict__ W1p,
        const float* __restrict__ b1, const uint16_t* __restrict__ W2p, const float* __restrict__ b2,
        const float* __restrict__ gamma, const float* __restrict__ beta, float eps, uint16_t* __restrict__ out, int M,
        const uint16_t* __restrict__ ctx, const uint16_t* __restrict__ Wo_p, const float* __restrict__ bo,
        const float* __restrict__ gamma1, const float* __restrict__ beta1)
{
    __shared__ __attribute__((aligned(1024))) char lds[F2_LDS_BYTES];
    static_assert(F2_LDS_BYTES <= 160 * 1024, "LDS");
    static_assert(F2_RING >= 4096 + 128 * LN_TS + 3 * LN_N * 4, "the epilogue's scratch fits in the ring");
    static_assert(F2_B1_OFF >= 4 * LN_STAGE + 3 * LN_N * 4, "b1 sits behind the front phase's ring and parameters");
    constexpr int nkt = F2_NKT;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l15 = lane & 15, g4 = lane >> 4;
    const int wm = wave >> 1, wn = wave & 1;                       // step 2: row group, column half
    const int kh = wave & 1;                                       // step 1: K half (row group wm)
    const int m0 = (int)(gridDim.x - 1 - blockIdx.x) * 128;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
#ifdef FF_TRACE
    const int ff_slot = (blockIdx.x % 64 == 0 && blockIdx.x / 64 < 8) ? (int)(blockIdx.x / 64) : -1;
    if (ff_slot >= 0 && lane == 0) ff_trace_buf[ff_slot][wave][15] = (uint32_t)__builtin_amdgcn_s_memrealtime();
#endif
    FF_STAMP(0);

    f32x4 acc[2][12];
    // h fragments (B operand of step 1), k-steps 6 kh .. 6 kh + 5: hqo of the wave's own row block (rows 16 wave + l15),
    // hqp of the partner's (rows 16 (wave ^ 1) + l15); lane (l15, g4): k = 32 s + 8 g4 ..
    bf16x8 hqo[6], hqp[6];
    if constexpr (FRONT) {
        uint2 rres[2][12];
        rowgemm384_mainloop(lds, ctx, Wo_p, bo, h, gamma1, beta1, M, LN_N, 1, m0, lane, wave, acc, rres);
        FF_STAMP(1);
        FF_TICKS_OUT_AT(32, true);
        ln_rows_epilogue<false>(lds, reinterpret_cast<const float*>(lds + 4 * LN_STAGE), acc, rres, nullptr, m0, M, eps, wm, wn, l15, g4);
        FF_STAMP(2);
#pragma unroll
        for (int s = 0; s < 6; ++s) {
            hqo[s] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(lds + 4096 + (16 * wave + l15) * LN_TS + (32 * (6 * kh + s) + 8 * g4) * 2));
            hqp[s] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(lds + 4096 + (16 * (wave ^ 1) + l15) * LN_TS + (32 * (6 * kh + s) + 8 * g4) * 2));
        }
#pragma unroll
        for (int s = 0; s < 6; ++s) asm volatile("" :: "v"(hqo[s]), "v"(hqp[s]));
        __syncthreads();                                           // the parked tile gives way to the weight ring
    } else {
        int mo = m0 + 16 * wave + l15; if (mo > M - 1) mo = M - 1;
        int mp = m0 + 16 * (wave ^ 1) + l15; if (mp > M - 1) mp = M - 1;
#pragma unroll
        for (int s = 0; s < 6; ++s) {
            hqo[s] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(h + (int64_t)mo * LN_N + 32 * (6 * kh + s) + 8 * g4));
            hqp[s] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(h + (int64_t)mp * LN_N + 32 * (6 * kh + s) + 8 * g4));
        }
#pragma unroll
        for (int s = 0; s < 6; ++s) asm volatile("" :: "v"(hqo[s]), "v"(hqp[s]));
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();                                               // b1 is in LDS for everyone
    FF_STAMP(3);

    // ---- LDS-DMA: a block is 24 pieces of 1 KB, wave w issues pieces 3w .. 3w + 2
    // piece p of a block: 16 rows x 64 B of W2 (lane -> row 16 p + lane / 4, swizzled 16-B slot) or the p-th KB of a W1 block (stored as its LDS image)
    const char* const w2src = reinterpret_cast<const char*>(W2p) + (lane >> 2) * 64 + (((lane & 3) ^ ((0 - (lane >> 4)) & 3)) * 16);
    const char* const w1src = reinterpret_cast<const char*>(W1p) + lane * 16;
    auto dma = [&](const char* src, int j, int slot_off, int piece) __attribute__((always_inline)) {
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)(src + ((int64_t)j * FF_WS + piece * 1024)),
            (lds_ptr_t)(lds + slot_off + piece * 1024), 16, 0, 0);
    };
    auto wrap = [](int o) __attribute__((always_inline)) { return o >= F2_RING ? o - F2_RING : o; };
    const int rot = wave >> 2;                                     // waves 4-7: one barrier late (see `stage`)
    // in the loop waves 0-3 request W2 blocks, waves 4-7 W1 blocks, six pieces per wave, with the base address in SGPRs and one
    // constant 32-bit lane offset (the builtin takes a 64-bit vector address: a 64-bit vector add per piece)
    const char* const baseD = reinterpret_cast<const char*>(rot ? W1p : W2p);
    const unsigned voffD = rot ? (unsigned)(lane * 16) : (unsigned)((lane >> 2) * 64 + (((lane & 3) ^ ((0 - (lane >> 4)) & 3)) * 16));
    auto dma_s = [&](int j, int slot_off, int piece) __attribute__((always_inline)) {
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                     :: "v"(voffD), "s"(baseD + ((int64_t)j * FF_WS + piece * 1024)), "s"(lds_base + (unsigned)(slot_off + piece * 1024)) : "memory", "m0");
    };
    const int pc0 = 6 * (wave & 3);
    const int pp0 = 3 * wave;                                      // prologue: every wave three pieces of every block

#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int nb = 0; nb < 12; ++nb)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[mb][nb][i] = 0.0f;

    const unsigned swz = (unsigned)((0 - (l15 >> 2)) & 3);
    const unsigned fsw = ((unsigned)g4 ^ swz) << 4;
    const unsigned a_rd = lds_base + F2_A_OFF + (unsigned)((32 * wm + l15) * 64) + fsw;       // + mb * 1024
    const unsigned w2_rd = lds_base + (unsigned)((192 * wn + l15) * 64) + fsw;                 // + slot + g * 4096 + q * 1024
    const unsigned w1_rd = lds_base + (unsigned)(l15 * 64) + fsw + (unsigned)kh * 12288;       // + slot + sp * 4096 + (2 q + cb) * 1024
    const unsigned a_wr = lds_base + F2_A_OFF + (unsigned)((16 * wave + l15) * 64) + ((((unsigned)g4 >> 1) ^ swz) << 4) + (g4 & 1) * 8;   // cb = 1: ^ 32
    const unsigned x_wr = lds_base + F2_X_OFF + (unsigned)(wave * 2048 + lane * 16);           // + cb * 1024
    const unsigned x_rd = lds_base + F2_X_OFF + (unsigned)((wave ^ 1) * 2048 + lane * 16);
    const unsigned b1_rd = lds_base + F2_B1_OFF + (unsigned)(4 * g4 * 4);                      // + (32 j + 16 cb) * 4

    bf16x8 fr[2][4], af[2];
    f32x4 Po[2], Pp[2];                                            // step-1 partial sums: own row block, partner's row block
    f32x4 xr[2], bb[2];
    auto rd_group = [&](bf16x8 (&d)[4], unsigned base) __attribute__((always_inline)) {
        asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:1024\n\t"
                     "ds_read_b128 %2, %4 offset:2048\n\tds_read_b128 %3, %4 offset:3072"
                     : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3]) : "v"(base));
    };
    auto rd_af = [&]() __attribute__((always_inline)) {
        asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:1024" : "=&v"(af[0]), "=&v"(af[1]) : "v"(a_rd));
    };
    auto rd_xb = [&](int j) __attribute__((always_inline)) {                   // partner's partial + bias of block j
        asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:1024" : "=&v"(xr[0]), "=&v"(xr[1]) : "v"(x_rd));
        asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:64" : "=&v"(bb[0]), "=&v"(bb[1]) : "v"(b1_rd + (unsigned)j * 128u));
    };
    auto wr_x = [&]() __attribute__((always_inline)) {
        // (Pp comes straight out of the MFMAs and hipcc does not see an inline-asm reader: the wait states are written out)
        asm volatile("s_nop 15\n\ts_nop 3\n\tds_write_b128 %0, %1\n\tds_write_b128 %0, %2 offset:1024" :: "v"(x_wr), "v"(Pp[0]), "v"(Pp[1]) : "memory");
    };
#define F2_WAIT_FR(N, d) asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]))
    auto wait_fr = [&](auto Nc, bf16x8 (&d)[4]) __attribute__((always_inline)) {
        constexpr int N = decltype(Nc)::value;
        static_assert(N == 0 || N == 4 || N == 6 || N == 8, "lgkmcnt");
        if constexpr (N == 0) F2_WAIT_FR(0, d); else if constexpr (N == 4) F2_WAIT_FR(4, d);
        else if constexpr (N == 6) F2_WAIT_FR(6, d); else F2_WAIT_FR(8, d);
    };
    auto wait_lgkm = [&](auto Nc) __attribute__((always_inline)) {
        constexpr int N = decltype(Nc)::value;
        if constexpr (N == 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        else if constexpr (N == 4) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
        else if constexpr (N == 6) asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
        else static_assert(N == 0, "lgkmcnt");
    };
    auto wait_vm = [&](auto Nc) __attribute__((always_inline)) {
        constexpr int N = decltype(Nc)::value;
        if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else if constexpr (N == 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
        else static_assert(N < 0, "vmcnt");                        // negative: nothing to wait for
    };
    // step 2, group g: 4 W2 fragments (column blocks 4 g .. 4 g + 3) x the two A fragments
    auto mma2q = [&](bf16x8 (&d)[4], auto Gc, int q0, int q1) __attribute__((always_inline)) {
        constexpr int g = decltype(Gc)::value;
#pragma unroll
        for (int q = q0; q < q1; ++q)
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
                acc[mb][4 * g + q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(d[q], af[mb], acc[mb][4 * g + q], 0, 0, 0);
    };
    auto mma2 = [&](bf16x8 (&d)[4], auto Gc) __attribute__((always_inline)) { mma2q(d, Gc, 0, 4); };
    // step 1, group sp: W1 fragments of k-steps 6 kh + 2 sp + q, column blocks cb (d[2 q + cb]) x both row blocks
    auto mma1 = [&](bf16x8 (&d)[4], auto SPc) __attribute__((always_inline)) {
        constexpr int sp = decltype(SPc)::value;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
                Pp[cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(d[2 * q + cb], hqp[2 * sp + q], (sp == 0 && q == 0) ? f32x4{} : Pp[cb], 0, 0, 0);
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
                Po[cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(d[2 * q + cb], hqo[2 * sp + q], (sp == 0 && q == 0) ? f32x4{} : Po[cb], 0, 0, 0);
        }
    };
    // own partial + partner's partial + bias -> GELU -> bf16 -> A (rows 16 wave + l15, columns 16 cb + 4 g4 ..)
    f32x4 tv[2];
    auto fin_sum = [&]() __attribute__((always_inline)) {
        asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(xr[0]), "+v"(xr[1]), "+v"(bb[0]), "+v"(bb[1]));
        tv[0] = (Po[0] + xr[0]) + bb[0];
        tv[1] = (Po[1] + xr[1]) + bb[1];
    };
    auto fin_half = [&](int cb) __attribute__((always_inline)) {
        float v0 = tv[cb][0], v1 = tv[cb][1], v2 = tv[cb][2], v3 = tv[cb][3];
#ifndef F2_ABL_GELU                                                // (F2_ABL_*: timing ablations of the debug build, wrong results)
        gelu_erf4(v0, v1, v2, v3);
#endif
        const uint2 o = make_uint2(pack2(v0, v1), pack2(v2, v3));
        asm volatile("ds_write_b64 %0, %1" :: "v"(a_wr ^ (cb ? 32u : 0u)), "v"(o) : "memory");
    };
#define F2_INTERLEAVE(NV) do { _Pragma("unroll") for (int _i = 0; _i < 8; ++_i) { \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, NV, 0); } } while (0)

    // ---- prologue: blocks 0..4 requested; step 1 of blocks 0 and 1 without overlap; A(0) written
    {
        // b1 -> LDS rides along: loaded ahead of the block requests, stored behind them (first read behind the second barrier below)
        float4 b1v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (threadIdx.x < 1536 / 4) b1v = *reinterpret_cast<const float4*>(b1 + 4 * threadIdx.x);
#pragma unroll
        for (int i = 0; i < 3; ++i) dma(w1src, 0, 0 * FF_WS, pp0 + i);
#pragma unroll
        for (int i = 0; i < 3; ++i) dma(w1src, 1, 1 * FF_WS, pp0 + i);
#pragma unroll
        for (int i = 0; i < 3; ++i) dma(w2src, 0, 2 * FF_WS, pp0 + i);
#pragma unroll
        for (int i = 0; i < 3; ++i) dma(w1src, 2, 3 * FF_WS, pp0 + i);
#pragma unroll
        for (int i = 0; i < 3; ++i) dma(w2src, 1, 4 * FF_WS, pp0 + i);
        if (threadIdx.x < 1536 / 4) *reinterpret_cast<float4*>(lds + F2_B1_OFF + threadIdx.x * 16) = b1v;
        wait_vm(IC<9>{});
        __builtin_amdgcn_s_barrier();                              // W1(0), W1(1) have landed for everyone
        FF_STAMP(4);
        auto step1_all = [&](unsigned wb) __attribute__((always_inline)) {
            rd_group(fr[0], wb);
            rd_group(fr[1], wb + 4096);
            wait_fr(IC<4>{}, fr[0]); mma1(fr[0], IC<0>{});
            rd_group(fr[0], wb + 8192);
            wait_fr(IC<4>{}, fr[1]); mma1(fr[1], IC<1>{});
            wait_fr(IC<0>{}, fr[0]); mma1(fr[0], IC<2>{});
        };
        step1_all(w1_rd + 0 * FF_WS);
        wr_x();
        wait_lgkm(IC<0>{});
        __builtin_amdgcn_s_barrier();                              // X(0) is written
        rd_xb(0);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xr[0]), "+v"(xr[1]), "+v"(bb[0]), "+v"(bb[1]));
        tv[0] = (Po[0] + xr[0]) + bb[0];
        tv[1] = (Po[1] + xr[1]) + bb[1];
        fin_half(0); fin_half(1);
        step1_all(w1_rd + 1 * FF_WS);
        wait_vm(IC<0>{});                                          // W2(0), W1(2), W2(1): the loop's own waits cover only what it requests
        wait_lgkm(IC<0>{});
        __builtin_amdgcn_s_barrier();                              // A(0) complete, X(0) consumed, W2(0) landed, slots 0 and 1 free
        rd_af();
        rd_group(fr[0], w2_rd + 2 * FF_WS);
        wr_x();                                                    // X(1)
    }
    FF_STAMP(5);

    // One stage of a wave.  Groups 0-2: step 2 of block kt (W2(kt) in slot off2); groups 3-5: step 1 of block kt + 2
    // (W1(kt + 2) in slot off1).  Barrier "alpha" behind group 0: X(kt + 1) is written.  Barrier "beta" behind group 2:
    // A(kt + 1) is written, X(kt + 1) consumed.  Between alpha and beta a wave has 16 MFMAs and the GELU of a block
    // (~100 VALU instructions, which its own MFMAs do not hide: an in-order wave pays for both); between beta and the next
    // alpha it has 32 MFMAs and requests a quarter of a weight block (six LDS-DMA pieces at 60-180 cycles of issue each).
    // Waves 4-7 (rot = 1) run the same program ONE BARRIER LATE, so on every SIMD one wave is in the first kind of
    // interval while the other is in the second, and the two intervals cost a wave about the same time.
    // The ring is kept by the count of hardware barriers b_0, b_1, ... (alpha(kt) = b_{2 kt + rot}, beta(kt) = b_{2 kt + 1 + rot}):
    // every wave has left block i behind b_i, so block i + 5 is requested right behind b_i -- by the four waves for which b_i
    // is a beta: waves 0-3 request the W2 blocks, waves 4-7 the W1 blocks -- and its requesters wait for it before they arrive
    // at b_{i+2}, their next beta (vmcnt(0): nothing else of theirs is in flight); the first read of block i + 5 comes behind b_{i+2}.
    // Flags: S1 step 1 of block kt + 2 exists, FIN block kt + 1 exists.
    auto stage = [&](int kt, int off2, int off1, auto S1c, auto FINc) __attribute__((always_inline)) {
        constexpr bool S1 = decltype(S1c)::value != 0, FIN = decltype(FINc)::value != 0;
        const int off2n = wrap(off2 + 2 * FF_WS);                  // slot of W2(kt + 1)
        const int jD = rot ? kt + 4 : kt + 2;                      // behind beta: W1(kt + 4) into the slot of W2(kt) / W2(kt + 2) into that of W1(kt + 1)
        const int slotD = rot ? off2 : wrap(off2 + 4 * FF_WS);
#ifdef F2_ABL_DMA
        const bool doD = false;
#else
        const bool doD = jD < nkt;
#endif
        FF_TICK(0);
        // ---- group 0
        rd_group(fr[1], w2_rd + off2 + 4096);
        wait_fr(IC<(FIN ? 6 : 4)>{}, fr[0]);                        // younger: X write (2), group 1 (4)
        asm volatile("" : "+v"(af[0]), "+v"(af[1]));
        mma2(fr[0], IC<0>{});
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (FIN) wait_lgkm(IC<4>{});                      // the X write is done
        FF_TICK(1);
        __builtin_amdgcn_s_barrier();                              // alpha
        FF_TICK(2);
        // ---- group 1
        if constexpr (FIN) rd_xb(kt + 1);
        rd_group(fr[0], w2_rd + off2 + 8192);
        wait_fr(IC<(FIN ? 8 : 4)>{}, fr[1]);
        mma2q(fr[1], IC<1>{}, 0, 2);                               // four MFMAs cover the partner's partial on its way from LDS
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (FIN) fin_sum();
        mma2q(fr[1], IC<1>{}, 2, 4);
        if constexpr (FIN) { fin_half(0); F2_INTERLEAVE(12); }
        __builtin_amdgcn_sched_barrier(0);
        FF_TICK(3);
        // ---- group 2
        if constexpr (S1) rd_group(fr[1], w1_rd + off1);
        wait_fr(IC<(S1 ? 4 : 0)>{}, fr[0]);
        mma2(fr[0], IC<2>{});
        if constexpr (FIN) { fin_half(1); F2_INTERLEAVE(6); }
        __builtin_amdgcn_sched_barrier(0);
        FF_TICK(4);
        wait_lgkm(IC<0>{});                                        // the A write is done
        wait_vm(IC<0>{});                                          // and the pieces requested behind the previous beta have landed
        FF_TICK(5);
        __builtin_amdgcn_s_barrier();                              // beta
        FF_TICK(6);
        // ---- group 3
        if constexpr (S1) {
            rd_group(fr[0], w1_rd + off1 + 4096);
            wait_fr(IC<4>{}, fr[1]);
            mma1(fr[1], IC<0>{});
        }
        FF_TICK(9);
        if (doD) { dma_s(jD, slotD, pc0 + 0); dma_s(jD, slotD, pc0 + 1); }
        FF_TICK(10);
        __builtin_amdgcn_sched_barrier(0);
        // ---- group 4
        if constexpr (S1) {
            rd_group(fr[1], w1_rd + off1 + 8192);
            wait_fr(IC<4>{}, fr[0]);
            mma1(fr[0], IC<1>{});
        }
        FF_TICK(11);
        if (doD) { dma_s(jD, slotD, pc0 + 2); dma_s(jD, slotD, pc0 + 3); }
        FF_TICK(12);
        __builtin_amdgcn_sched_barrier(0);
        FF_TICK(7);
        // ---- group 5
        if constexpr (FIN) { rd_af(); rd_group(fr[0], w2_rd + off2n); }        // block kt + 1 exists: its step 2 follows
        if constexpr (S1) {
            wait_fr(IC<(FIN ? 6 : 0)>{}, fr[1]);
            mma1(fr[1], IC<2>{});
            wr_x();                                                // X(kt + 2)
        }
        if (doD) { dma_s(jD, slotD, pc0 + 4); dma_s(jD, slotD, pc0 + 5); }
        __builtin_amdgcn_sched_barrier(0);
        FF_TICK(8);
    };

    {
        int off2 = 2 * FF_WS, off1 = 3 * FF_WS;
        int kt = 0;
        if (rot) {                                                 // b_0: the older half's first alpha
            __builtin_amdgcn_s_barrier();
#pragma unroll
            for (int i = 0; i < 6; ++i) dma(w1src, 3, 0 * FF_WS, pc0 + i);     // block 5 = W1(3), behind b_0
        }
        for (; kt < nkt - 2; ++kt) {
            FF_TICKS_OUT(kt - 1);
            stage(kt, off2, off1, IC<1>{}, IC<1>{});
            off2 = wrap(off2 + 2 * FF_WS);
            off1 = wrap(off1 + 2 * FF_WS);
        }
        stage(kt, off2, off1, IC<0>{}, IC<1>{});                   // nkt - 2
        off2 = wrap(off2 + 2 * FF_WS); off1 = wrap(off1 + 2 * FF_WS); ++kt;
        stage(kt, off2, off1, IC<0>{}, IC<0>{});                   // nkt - 1
        if (!rot) __builtin_amdgcn_s_barrier();                    // the younger half's last beta
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();                                               // the ring becomes the epilogue's scratch
    FF_STAMP(6);

    float* const prm = reinterpret_cast<float*>(lds + 4096 + 128 * LN_TS);
    if (threadIdx.x < 3 * LN_N / 4) {
        const int which = threadIdx.x / (LN_N / 4), o = (threadIdx.x % (LN_N / 4)) * 4;
        const float* srcp = which == 0 ? b2 : which == 1 ? gamma : beta;
        *reinterpret_cast<float4*>(prm + which * LN_N + o) = *reinterpret_cast<const float4*>(srcp + o);
    }
    // the residual is the tile of h the workgroup holds (step-1 layout): through LDS into the epilogue's layout
    char* const park = lds + 4096;
#pragma unroll
    for (int s = 0; s < 6; ++s) {
        *reinterpret_cast<uint4*>(park + (16 * wave + l15) * LN_TS + (32 * (6 * kh + s) + 8 * g4) * 2) = __builtin_bit_cast(uint4, hqo[s]);
        *reinterpret_cast<uint4*>(park + (16 * (wave ^ 1) + l15) * LN_TS + (32 * (6 * kh + s) + 8 * g4) * 2) = __builtin_bit_cast(uint4, hqp[s]);
    }
    __syncthreads();
    uint2 rres[2][12];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int nb = 0; nb < 12; ++nb)
            rres[mb][nb] = *reinterpret_cast<const uint2*>(park + (32 * wm + 16 * mb + l15) * LN_TS + (192 * wn + 16 * nb + 4 * g4) * 2);
    FF_STAMP(7);
    ln_rows_epilogue(lds, prm, acc, rres, out, m0, M, eps, wm, wn, l15, g4);
    FF_STAMP(8);
#ifdef FF_TRACE
    if (ff_slot >= 0 && lane == 0) ff_trace_buf[ff_slot][wave][14] = (uint32_t)__builtin_amdgcn_s_memrealtime();
#endif
}

// ---------------------------------------------------------------- attention (head size 32)
// One workgroup per (head, sequence, 256-query block); wave w owns 64 queries = 2 lane blocks.
// S^T = K Q^T (keys on accumulator rows, the query on the lane), so max/sum over keys are
// in-register plus one cross-half exchange; P (bf16) feeds O^T = V^T P^T directly as the B
// operand (the accumulator-as-operand k-permutation is matched by the V^T fragment reads).
__global__ __launch_bounds__(256, 4) void attn_kernel(const uint16_t* __restrict__ qkv, const int32_t* __restrict__ mask,
        const int32_t* __restrict__ seq_start, uint16_t* __restrict__ ctx, int L_in, int Lp_in, int H, int qkv_blocked, int ctx_blocked)
{
    // padded [B][L] tokens with a key mask, or packed sequences: tokens seq_start[b] .. seq_start[b + 1], no padding anywhere
    // (L_in / Lp_in then only size the launch: the longest sequence of the pass)
    int L = L_in, Lp = Lp_in;
    int64_t tok0 = (int64_t)blockIdx.y * L_in;
    if (seq_start) {
        tok0 = seq_start[blockIdx.y];
        L = seq_start[blockIdx.y + 1] - seq_start[blockIdx.y];
        Lp = (L + 31) / 32 * 32;
        if ((int)blockIdx.z * 256 >= L) return;              // whole workgroup: this sequence has no such query block
    }
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // Ks [Lp][64 B] swizzled | Vs [Lp][64 B] | maskf [Lp]
    char* Ks = smem;
    char* Vs = smem + (size_t)Lp * 64;
    float* maskf = reinterpret_cast<float*>(Vs + (size_t)Lp * 64);

    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int hp = lane >> 5, col = lane & 31;
    const int head = blockIdx.x;
    const int H3 = 3 * H;
    const uint16_t* base = qkv + tok0 * H3 + head * 32;
    // blocked layouts ([tokens / 128][columns / 32][128][32], written by gemm_ws_kernel / read by
    // gemm_ln_kernel): the 32 columns of a head are one column block, 128 tokens of it one 8-KB block
    const int nblk_q = H3 >> 5, nblk_c = H >> 5;
    auto qkv_at = [&](int key, int which, int c) -> const uint16_t* {     // which: 0 Q, 1 K, 2 V; c: 8-column chunk
        if (qkv_blocked) {
            const int64_t tg = tok0 + key;
            return qkv + (((tg >> 7) * nblk_q + which * nblk_c + head) * 128 + (tg & 127)) * 32 + c * 8;
        }
        return base + (int64_t)key * H3 + which * H + c * 8;
    };

    for (int idx = t; idx < Lp * 4; idx += 256) {
        const int key = idx >> 2, c = idx & 3;
        uint4 kv = make_uint4(0, 0, 0, 0), vv = make_uint4(0, 0, 0, 0);
        if (key < L) {
            kv = *reinterpret_cast<const uint4*>(qkv_at(key, 1, c));
            vv = *reinterpret_cast<const uint4*>(qkv_at(key, 2, c));
        }
        *reinterpret_cast<uint4*>(Ks + key * 64 + ((c ^ ((key >> 2) & 3)) << 4)) = kv;
        *reinterpret_cast<uint4*>(Vs + key * 64 + c * 16) = vv;      // row-major: the transposed fragments come from ds_read_b64_tr_b16
    }
    for (int key = t; key < Lp; key += 256)
        maskf[key] = (key < L && (seq_start || mask[tok0 + key] != 0)) ? 0.0f : -FLT_MAX;
    __syncthreads();

    const int q0 = blockIdx.z * 256 + wave * 64;
    if (q0 >= L) return;                                  // whole wave: nothing to do (no barriers follow)

    bf16x8 qf[2][2];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
        int q = q0 + 32 * nb + col;
        if (q > L - 1) q = L - 1;
#pragma unroll
        for (int s = 0; s < 2; ++s)
            qf[nb][s] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(qkv_at(q, 0, 2 * s + hp)));
    }
    // Scores arrive in log2 units: dh^-1/2 log2(e) is folded into Wq and bq when the weights are loaded.  A lane holds one
    // query (column) and 16 keys of the block.  m_run is the exponent reference of the running softmax; it only has to keep
    // exp2 in range, so it moves (and O, Ls are rescaled) only when a block's maximum exceeds it by more than 2^8 (p <= 2^8 is
    // as exact in fp32 and bf16 as p <= 1; the final 1 / l takes the same reference).
    // The loop is VALU-bound (16 v_exp_f32 at quarter rate per 32 x 32 block are most of it): the additive mask is the C
    // operand of the first QK^T MFMA, everything else runs two values per instruction (v_pk_add_f32) or three (v_max3_f32).
    // (Row sums by a third MFMA against an all-ones operand were measured too: 16 VALU instructions less per block, but 32
    // more registers -- three workgroups per CU instead of four, 53.5 against 50.9 us.)
    float m_run[2] = {-__builtin_inff(), -__builtin_inff()};
    f32x16 O[2];
    f32x2 ps2[2] = {{0.0f, 0.0f}, {0.0f, 0.0f}};              // running row sums (this lane's keys, even / odd registers)
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int i = 0; i < 16; ++i) O[nb][i] = 0.0f;
    auto max3 = [](float a, float b2, float c) __attribute__((always_inline)) {
        float r;
        asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b2), "v"(c));
        return r;
    };

    const int ksw = (col >> 2) & 3;
    for (int kb = 0; kb < Lp / 32; ++kb) {
        // K fragments of this key block (A operand): row = key 32 kb + col
        bf16x8 kf[2];
#pragma unroll
        for (int s = 0; s < 2; ++s)
            kf[s] = *reinterpret_cast<const bf16x8*>(Ks + (32 * kb + col) * 64 + (((2 * s + hp) ^ ksw) << 4));
        // V^T fragments (A operand of the PV product): row = dh (col), element j <-> key 32 kb + 16 s + 8 (j >> 2) + 4 hp + (j & 3),
        // by the hardware transpose read: a 16-lane group reads 4 keys x 16 dh, lane 4 q + p supplies the address of key q,
        // dh 4 p .., and lane i receives dh i of the 4 keys -- conflict-free on plain 64-B rows, and V is staged with
        // four 16-B stores per thread instead of 32 two-byte ones
        bf16x8 vf[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const char* vp = Vs + (32 * kb + 16 * s + 4 * hp + ((lane & 15) >> 2)) * 64 + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
            const tr4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tr4_lds_t*)(vp));
            const tr4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tr4_lds_t*)(vp + 8 * 64));
            vf[s] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
        }
        // additive mask of the 16 keys this lane's registers hold: key = 32 kb + 8 g + 4 hp + e; 0 or -FLT_MAX (finite, as the
        // reference's finfo.min: a masked score stays at -FLT_MAX, so an all-masked row softmaxes to uniform)
        f32x16 MK;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 mk = *reinterpret_cast<const float4*>(maskf + 32 * kb + 8 * g + 4 * hp);
            MK[4 * g + 0] = mk.x; MK[4 * g + 1] = mk.y; MK[4 * g + 2] = mk.z; MK[4 * g + 3] = mk.w;
        }
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            f32x16 S = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[0], qf[nb][0], MK, 0, 0, 0);        // score + mask
            S = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[1], qf[nb][1], S, 0, 0, 0);
            // d = S - m_run, two per instruction; the block maximum is taken of d (8 v_max3 written out: they read the
            // results of the subtraction, never an MFMA's directly -- hipcc pads MFMA results only for readers it can see)
            f32x2 d2[8];
            {
                const f32x2 mn2 = {m_run[nb], m_run[nb]};
#pragma unroll
                for (int i = 0; i < 8; ++i) d2[i] = f32x2{S[2 * i], S[2 * i + 1]} - mn2;
            }
            float bm = max3(max3(d2[0].x, d2[0].y, d2[1].x), max3(d2[1].y, d2[2].x, d2[2].y), max3(d2[3].x, d2[3].y, d2[4].x));
            bm = max3(bm, max3(d2[4].y, d2[5].x, d2[5].y), max3(d2[6].x, d2[6].y, d2[7].x));
            bm = fmaxf(bm, d2[7].y);
            bm = fmaxf(bm, __shfl_xor(bm, 32, 64));
            if (__builtin_amdgcn_ballot_w64(!(bm <= 8.0f)) != 0) {          // rare (the first block: m_run = -inf, d = +inf): move the reference exactly
                float tm = fmaxf(fmaxf(S[0], S[1]), S[2]);
#pragma unroll
                for (int i = 3; i < 16; ++i) tm = fmaxf(tm, S[i]);
                tm = fmaxf(tm, __shfl_xor(tm, 32, 64));
                const float m_new = fmaxf(m_run[nb], tm);
                const float alpha = __builtin_amdgcn_exp2f(m_run[nb] - m_new);
                O[nb] = O[nb] * alpha;
                ps2[nb] = ps2[nb] * alpha;
                m_run[nb] = m_new;
                const f32x2 mn2 = {m_new, m_new};
#pragma unroll
                for (int i = 0; i < 8; ++i) d2[i] = f32x2{S[2 * i], S[2 * i + 1]} - mn2;
            }
            // P^T as B operand: k-step s takes registers 8 s .. 8 s + 7
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                f32x2 e2[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    e2[i].x = __builtin_amdgcn_exp2f(d2[4 * s + i].x);
                    e2[i].y = __builtin_amdgcn_exp2f(d2[4 * s + i].y);
                    ps2[nb] = ps2[nb] + e2[i];
                }
                const uint4 pk = make_uint4(pack2(e2[0].x, e2[0].y), pack2(e2[1].x, e2[1].y), pack2(e2[2].x, e2[2].y), pack2(e2[3].x, e2[3].y));
                O[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[s], __builtin_bit_cast(bf16x8, pk), O[nb], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
        const int q = q0 + 32 * nb + col;
        const float lh = ps2[nb].x + ps2[nb].y;
        const float inv = 1.0f / (lh + __shfl_xor(lh, 32, 64));
        if (q < L) {
            const int64_t tg = tok0 + q;
            uint16_t* dst = ctx_blocked ? ctx + (((tg >> 7) * nblk_c + head) * 128 + (tg & 127)) * 32
                                        : ctx + tg * H + head * 32;
#pragma unroll
            for (int g = 0; g < 4; ++g) {       // dh = 8 g + 4 hp + (0..3)
                uint2 o;
                o.x = pack2(O[nb][4 * g + 0] * inv, O[nb][4 * g + 1] * inv);
                o.y = pack2(O[nb][4 * g + 2] * inv, O[nb][4 * g + 3] * inv);
                *reinterpret_cast<uint2*>(dst + 8 * g + 4 * hp) = o;
            }
        }
    }
}

// ---------------------------------------------------------------- masked mean pool + L2 normalise
// One workgroup of 16 waves per sequence; wave w sums tokens i = w (mod 16), lane l the 8 columns
// [8 l + 512 c, +8); the partial sums meet in LDS.  fp32 throughout.
#define POOL_WAVES 16
template <int NCH>
__global__ __launch_bounds__(64 * POOL_WAVES) void pool_kernel(const uint16_t* __restrict__ hid, const int32_t* __restrict__ mask,
        const int32_t* __restrict__ seq_start, int L_in, int H, int normalize, float* __restrict__ out)
{
    __shared__ float part[POOL_WAVES][NCH * 512];
    __shared__ float red[POOL_WAVES];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int64_t b = blockIdx.x;
    float acc[NCH][8];
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[c][e] = 0.0f;
    int cnt = 0;
    // padded [B][L] rows with a mask, or packed sequences (rows seq_start[b] .. seq_start[b + 1], every token valid)
    const int64_t tok0 = seq_start ? (int64_t)seq_start[b] : b * L_in;
    const int L = seq_start ? seq_start[b + 1] - seq_start[b] : L_in;
    for (int i = wave; i < L; i += POOL_WAVES) {
        if (!seq_start && mask[tok0 + i] == 0) continue;      // wave-uniform
        ++cnt;
        const uint16_t* row = hid + (tok0 + i) * H;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int col = 8 * lane + 512 * c;
            if (col < H) {
                const uint4 v = *reinterpret_cast<const uint4*>(row + col);
                const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc[c][2 * e] += __builtin_bit_cast(float, w[e] << 16);
                    acc[c][2 * e + 1] += __builtin_bit_cast(float, w[e] & 0xFFFF0000u);
                }
            }
        }
    }
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int e = 0; e < 8; ++e) part[wave][512 * c + 8 * lane + e] = acc[c][e];
    if (lane == 0) red[wave] = (float)cnt;
    __syncthreads();
    float cnt_all = 0.0f;
#pragma unroll
    for (int w = 0; w < POOL_WAVES; ++w) cnt_all += red[w];
    const float den = fmaxf(cnt_all, 1e-9f);
    __syncthreads();
    // threads 0 .. 255 finish: column t + 256 e
    float v[2 * NCH];
    float ss = 0.0f;
    if (t < 256) {
#pragma unroll
        for (int e = 0; e < 2 * NCH; ++e) {
            const int col = t + 256 * e;
            float a = 0.0f;
            if (col < H) {
#pragma unroll
                for (int w = 0; w < POOL_WAVES; ++w) a += part[w][col];
            }
            v[e] = a / den;
            ss += v[e] * v[e];
        }
        ss = wave_sum(ss);
        if (lane == 0) red[wave] = ss;
    }
    __syncthreads();
    if (t < 256) {
        const float nrm = fmaxf(sqrtf(red[0] + red[1] + red[2] + red[3]), 1e-12f);
#pragma unroll
        for (int e = 0; e < 2 * NCH; ++e) {
            const int col = t + 256 * e;
            if (col < H) out[b * H + col] = normalize ? v[e] / nrm : v[e];
        }
    }
}

// ---------------------------------------------------------------- weights: fp32 [n][k] -> bf16
// fp32 [N][K] -> bf16 [K / 32][N][32]
__global__ __launch_bounds__(256) void cvt_pack32_kernel(const float* __restrict__ in, uint16_t* __restrict__ out, int N, int K)
{
    const int64_t total = (int64_t)N * K;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int kk = (int)(i & 31);
        const int64_t t = i >> 5;
        const int n = (int)(t % N), kb = (int)(t / N);
        out[i] = f2bf(in[(int64_t)n * K + kb * 32 + kk]);
    }
}

__global__ __launch_bounds__(256) void scale_f32_kernel(const float* __restrict__ in, float* __restrict__ out, int n, float scale)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = in[i] * scale;
}

__global__ __launch_bounds__(256) void cvt_bf16_kernel(const float* __restrict__ in, uint16_t* __restrict__ out, int64_t n, float scale)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) out[i] = mi_f32_to_bf16(in[i] * scale);
}


// ---------------------------------------------------------------- cross-encoder head
// BertPooler + classifier (modeling_bert.py BertPooler: tanh(W_p h[CLS] + b_p); BertForSequenceClassification:
// logits = W_c pooled + b_c; dropout is the identity at inference).  One workgroup per sequence, fp32 throughout;
// an output is one wave's dot product (lanes over the hidden columns, coalesced rows of W).
#define CLS_PB 4                                 // pairs per workgroup: the pooler matrix is read once per four pairs
// One thread per pooler output; WpT is the pooler matrix transposed at load ([in][out]) so a wave reads 64 consecutive
// outputs of one input column, and the four [CLS] rows sit in LDS as [in][CLS_PB] for one broadcast b128 read per column.
__global__ __launch_bounds__(1024) void cls_head_kernel(const uint16_t* __restrict__ hid, const int32_t* __restrict__ seq_start, int L, int H,
        const float* __restrict__ WpT, const float* __restrict__ bp, const float* __restrict__ Wc, const float* __restrict__ bc,
        int nlabels, int64_t B, float* __restrict__ out)
{
    extern __shared__ float cls_smem[];              // x [H][CLS_PB] | pooled [CLS_PB][H]
    float* x = cls_smem;
    float* pooled = cls_smem + CLS_PB * H;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, nw = blockDim.x >> 6;
    const int64_t b0 = (int64_t)blockIdx.x * CLS_PB;
    const int np = (int)((B - b0 < CLS_PB) ? B - b0 : CLS_PB);
    for (int i = t; i < CLS_PB * H; i += blockDim.x) {
        const int p = i / H, c = i - p * H;
        float v = 0.0f;
        if (p < np) {
            const int64_t b = b0 + p;
            v = bf2f(hid[(seq_start ? (int64_t)seq_start[b] : b * L) * (int64_t)H + c]);      // the [CLS] token is position 0
        }
        x[c * CLS_PB + p] = v;
    }
    __syncthreads();
    if (t < H) {
        float s[CLS_PB];
#pragma unroll
        for (int p = 0; p < CLS_PB; ++p) s[p] = 0.0f;
#pragma unroll 8
        for (int c = 0; c < H; ++c) {
            const float w = WpT[(int64_t)c * H + t];
            const float4 xv = *reinterpret_cast<const float4*>(x + c * CLS_PB);
            s[0] += w * xv.x; s[1] += w * xv.y; s[2] += w * xv.z; s[3] += w * xv.w;
        }
        const float bo = bp[t];
#pragma unroll
        for (int p = 0; p < CLS_PB; ++p) pooled[p * H + t] = tanhf(s[p] + bo);
    }
    __syncthreads();
    for (int i = wave; i < np * nlabels; i += nw) {
        const int p = i / nlabels, j = i - p * nlabels;
        float s = 0.0f;
        for (int c = lane; c < H; c += 64) s += Wc[(int64_t)j * H + c] * pooled[p * H + c];
        s = wave_sum(s);
        if (lane == 0) out[(b0 + p) * nlabels + j] = s + bc[j];
    }
}
static_assert(CLS_PB == 4, "cls_head_kernel reads the four [CLS] values of a column as one float4");

// ================================================================ host side
// Activation workspace of the allocation-free entry point (mi_encoder_encode_dev_noalloc): callers that cannot
// call hipMalloc -- code under HIP graph capture, tight serving loops that embed one query at a time -- get the
// encoder's per-pass workspace from this array instead of the grow-only hipMalloc'd one.  One copy per GPU the
// library runs on, held until the process ends.  `scratch` is a host-side stand-in here, not a device address:
// host code only ever uses the pointer hipGetSymbolAddress returned for the current device (g_scratch_ptr).
#define MI_ENC_NOALLOC_BYTES ((size_t)16 << 20)
static __device__ __align__(256) unsigned char scratch[MI_ENC_NOALLOC_BYTES];
#define MI_MAX_DEVICES 64
static void* g_scratch_ptr[MI_MAX_DEVICES];          // per device: each GPU holds its own copy at its own address
static std::mutex g_scratch_mu;

// Once per device, with that device current (mi_encoder_create, the encoder's init call).
static int scratch_for_device(int device, void** out)
{
    if (device < 0 || device >= MI_MAX_DEVICES) { mi_set_error("encoder: device %d beyond the %d this build keeps scratch pointers for", device, MI_MAX_DEVICES); return MI_ERR_UNSUPPORTED; }
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    if (!g_scratch_ptr[device]) {
        void* ptr = nullptr;
        hipError_t er = hipGetSymbolAddress((void**)&ptr, HIP_SYMBOL(scratch));
        if (er != hipSuccess || !ptr) { mi_set_error("encoder: hipGetSymbolAddress(scratch) failed on device %d: %s", device, hipGetErrorString(er)); return MI_ERR_HIP; }
        g_scratch_ptr[device] = ptr;
    }
    *out = g_scratch_ptr[device];
    return MI_OK;
}

struct LayerW {
    uint16_t *Wqkv, *Wo, *Wi, *Wd;                 // bf16 [out][in]
    uint16_t* Wqkv_p = nullptr;                    // Wqkv per head in MFMA operand order (encoder_qkv_attn.hip)
    uint16_t *Wi_p = nullptr;                      // [I / 32][12][32][32] swizzled LDS image of Wi blocks (ffn2_kernel)
    uint16_t *Wo_p = nullptr, *Wd_p = nullptr;     // the same, [in / 32][out][32]: a 32-k stage of gemm_ln_kernel is one contiguous block
    float *bqkv, *bo, *bi, *bd, *g1, *b1, *g2, *b2; // fp32
    // ffn3_kernel (encoder_ffn3.hip): Wo, Wi, Wd as ONE stream of 108 fragment blocks in consumption order, and the layer's
    // biases / LayerNorm parameters as one block (hidden 384, intermediate 1536 only; nullptr otherwise)
    void* w3 = nullptr;
    float* p3 = nullptr;
};

// grow-only pinned host block
struct PinBuf {
    void* p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes)
    {
        if (bytes <= cap) return MI_OK;
        if (p) { (void)hipHostFree(p); p = nullptr; cap = 0; }
        hipError_t e = hipHostMalloc(&p, bytes, hipHostMallocDefault);
        if (e != hipSuccess) { p = nullptr; mi_set_error("hipHostMalloc(%zu) failed: %s", bytes, hipGetErrorString(e)); return MI_ERR_OOM; }
        cap = bytes;
        return MI_OK;
    }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }
};

struct mi_encoder {
    int device = 0;
    hipStream_t stream = nullptr;
    mi_bert_config cfg{};
    float* blob = nullptr;          // the fp32 blob on device (embedding tables, biases, LN params live here)
    uint16_t* wbf = nullptr;        // all bf16 weight matrices
    float *word = nullptr, *pos = nullptr, *type0 = nullptr, *eg = nullptr, *eb = nullptr;
    std::vector<LayerW> layers;
    std::mutex mu;
    // Two LANES: a call of several passes runs them alternately on two streams, each lane with its own workspace, staging
    // and events, so one pass's memory-bound kernels (QKV projection, attention) and launch gaps run under the other's
    // matrix-bound layer tail.  Lane 0 uses the caller's stream, lane 1 the encoder's own `st2`; a single pass is lane 0.
    struct Lane {
        void* ws = nullptr;             // activation workspace (grow-only)
        size_t ws_bytes = 0;
        int32_t* d_io = nullptr;        // device staging of the host-pointer entry points: ids | types | mask ...
        float* d_out = nullptr;         // ... and the outputs
        size_t io_cap = 0, out_cap = 0;
        hipEvent_t ev0 = nullptr, ev1 = nullptr;
        // ev1 also orders the lane's workspace: it is recorded behind every pass, and a pass enqueued on ANOTHER stream than
        // the lane's last one (a PyTorch stream through *_dev, then the ctx stream through a host entry point) waits on it first
        hipStream_t last_st = nullptr;
        bool have_last = false;
    } ln[2];
    hipStream_t st2 = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    int last_lane = 0;                  // lane of the last pass (mi_encoder_last_time)
    // pinned host staging: two sets, so the host fills one while the GPU reads the other
    PinBuf pin_io[2], pin_out[2];
    hipEvent_t ev_pin[2] = {nullptr, nullptr};
    bool timed = false;
    // cross-encoder head (mi_encoder_set_head): pooler W[H][H] b[H], classifier W[nl][H] b[nl], fp32 on device
    float* head = nullptr;
    int nlabels = 0;
    // counters since creation: sequences, padded tokens, passes on the bulk / on the latency kernel family
    int64_t st_seqs = 0, st_tokens = 0, st_bulk = 0, st_latency = 0;
    // this device's address of the file-scope `scratch` array (mi_encoder_encode_dev_noalloc)
    void* scratch_dev = nullptr;
};

extern "C" int64_t mi_encoder_blob_elems(const mi_bert_config* c)
{
    if (!c) return -1;
    const int64_t H = c->hidden_size, I = c->intermediate_size;
    int64_t n = (int64_t)c->vocab_size * H + (int64_t)c->max_position_embeddings * H + (int64_t)c->type_vocab_size * H + 2 * H;
    n += (int64_t)c->num_hidden_layers * (4 * (H * H + H) + 2 * H + (I * H + I) + (H * I + H) + 2 * H);
    return n;
}

static void encoder_free(mi_encoder* e)
{
    if (!e) return;
    for (LayerW& lw : e->layers) { if (lw.bqkv) (void)hipFree(lw.bqkv); if (lw.w3) (void)hipFree(lw.w3); if (lw.p3) (void)hipFree(lw.p3); }
    if (e->blob) (void)hipFree(e->blob);
    if (e->wbf) (void)hipFree(e->wbf);
    for (auto& n : e->ln) {
        if (n.ws) (void)hipFree(n.ws);
        if (n.d_io) (void)hipFree(n.d_io);
        if (n.d_out) (void)hipFree(n.d_out);
        if (n.ev0) (void)hipEventDestroy(n.ev0);
        if (n.ev1) (void)hipEventDestroy(n.ev1);
    }
    if (e->ev_fork) (void)hipEventDestroy(e->ev_fork);
    if (e->ev_join) (void)hipEventDestroy(e->ev_join);
    if (e->st2) (void)hipStreamDestroy(e->st2);
    if (e->head) (void)hipFree(e->head);
    for (int s = 0; s < 2; ++s) { e->pin_io[s].release(); e->pin_out[s].release(); if (e->ev_pin[s]) (void)hipEventDestroy(e->ev_pin[s]); }
    delete e;
}

extern "C" int mi_encoder_create(mi_ctx* ctx, const mi_bert_config* cfg, const float* blob_host, size_t n_elems, mi_encoder** out)
{
    MI_REQUIRE(ctx && cfg && blob_host && out, "mi_encoder_create: NULL argument");
    *out = nullptr;
    const int H = cfg->hidden_size, I = cfg->intermediate_size, NH = cfg->num_attention_heads;
    MI_REQUIRE(H > 0 && H % 128 == 0 && H <= 1024, "mi_encoder_create: hidden_size %d must be a multiple of 128 and <= 1024", H);
    MI_REQUIRE(I > 0 && I % 128 == 0, "mi_encoder_create: intermediate_size %d must be a multiple of 128", I);
    MI_REQUIRE(NH > 0 && H % NH == 0, "mi_encoder_create: hidden_size %% heads != 0");
    if (H / NH != 32) { mi_set_error("mi_encoder_create: head size %d not supported (32 only in this build)", H / NH); return MI_ERR_UNSUPPORTED; }
    MI_REQUIRE(cfg->num_hidden_layers > 0 && cfg->vocab_size > 0 && cfg->max_position_embeddings > 0 && cfg->type_vocab_size > 0,
               "mi_encoder_create: bad config");
    const int64_t want = mi_encoder_blob_elems(cfg);
    MI_REQUIRE((int64_t)n_elems == want, "mi_encoder_create: blob has %zu elements, config needs %lld", n_elems, (long long)want);

    const int device = mi_ctx_device(ctx);
    DeviceGuard guard(device);
    mi_encoder* e = new (std::nothrow) mi_encoder();
    if (!e) { mi_set_error("mi_encoder_create: out of host memory"); return MI_ERR_OOM; }
    e->device = device;
    e->stream = mi_ctx_stream(ctx);
    e->cfg = *cfg;
    auto fail = [&](int code) { encoder_free(e); return code; };
#define ENC_HIP(call) do { hipError_t _e = (call); if (_e != hipSuccess) { mi_set_error("%s failed: %s", #call, hipGetErrorString(_e)); return fail(_e == hipErrorOutOfMemory ? MI_ERR_OOM : MI_ERR_HIP); } } while (0)
    { const int src = scratch_for_device(device, &e->scratch_dev); if (src != MI_OK) return fail(src); }
    ENC_HIP(hipMalloc((void**)&e->blob, (size_t)want * sizeof(float)));
    ENC_HIP(hipMemcpyAsync(e->blob, blob_host, (size_t)want * sizeof(float), hipMemcpyHostToDevice, e->stream));
    const int64_t per_layer_w = (int64_t)4 * H * H + 2 * (int64_t)I * H + ((int64_t)H * H + 2 * (int64_t)H * I) + (int64_t)3 * H * H;   // + packed Wo, Wd, Wi, Wqkv
    ENC_HIP(hipMalloc((void**)&e->wbf, (size_t)(per_layer_w * cfg->num_hidden_layers) * sizeof(uint16_t)));
    for (auto& n : e->ln) { ENC_HIP(hipEventCreate(&n.ev0)); ENC_HIP(hipEventCreate(&n.ev1)); }
    ENC_HIP(hipStreamCreateWithFlags(&e->st2, hipStreamNonBlocking));
    ENC_HIP(hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming));
    ENC_HIP(hipEventCreateWithFlags(&e->ev_join, hipEventDisableTiming));
    for (int s = 0; s < 2; ++s) ENC_HIP(hipEventCreateWithFlags(&e->ev_pin[s], hipEventDisableTiming));

    float* p = e->blob;
    e->word = p; p += (int64_t)cfg->vocab_size * H;
    e->pos = p; p += (int64_t)cfg->max_position_embeddings * H;
    e->type0 = p; p += (int64_t)cfg->type_vocab_size * H;
    e->eg = p; p += H;
    e->eb = p; p += H;
    uint16_t* w = e->wbf;
    auto cvt = [&](const float* src, uint16_t* dst, int64_t n, float scale = 1.0f) {
        cvt_bf16_kernel<<<(unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096), 256, 0, e->stream>>>(src, dst, n, scale);
    };
    // attention scores in log2 units straight out of the QK^T MFMA: dh^-1/2 * log2(e) is folded into Wq and bq (modeling_bert.py:
    // scores / sqrt(head size), then softmax; attn_kernel exponentiates with exp2).  One fp32 multiply before the bf16 rounding.
    const float qscale = (1.0f / sqrtf((float)(H / cfg->num_attention_heads))) * 1.4426950408889634f;
    for (int l = 0; l < cfg->num_hidden_layers; ++l) {
        LayerW lw{};
        // blob: Wq bq Wk bk Wv bv Wo bo g1 b1 Wi bi Wd bd g2 b2 ; fused QKV needs Wq|Wk|Wv rows back to back
        float* Wq = p; float* bq = Wq + (int64_t)H * H;
        float* Wk = bq + H; float* bk = Wk + (int64_t)H * H;
        float* Wv = bk + H; float* bv = Wv + (int64_t)H * H;
        float* Wo = bv + H; float* bo = Wo + (int64_t)H * H;
        float* g1 = bo + H; float* b1 = g1 + H;
        float* Wi = b1 + H; float* bi = Wi + (int64_t)I * H;
        float* Wd = bi + I; float* bd = Wd + (int64_t)H * I;
        float* g2 = bd + H; float* b2 = g2 + H;
        p = b2 + H;
        lw.Wqkv = w; cvt(Wq, w, (int64_t)H * H, qscale); cvt(Wk, w + (int64_t)H * H, (int64_t)H * H); cvt(Wv, w + 2 * (int64_t)H * H, (int64_t)H * H);
        w += 3 * (int64_t)H * H;
        lw.Wo = w; cvt(Wo, w, (int64_t)H * H); w += (int64_t)H * H;
        lw.Wi = w; cvt(Wi, w, (int64_t)I * H); w += (int64_t)I * H;
        lw.Wd = w; cvt(Wd, w, (int64_t)H * I); w += (int64_t)H * I;
        lw.Wo_p = w; cvt_pack32_kernel<<<1024, 256, 0, e->stream>>>(Wo, w, H, H); w += (int64_t)H * H;
        lw.Wd_p = w; cvt_pack32_kernel<<<1024, 256, 0, e->stream>>>(Wd, w, H, I); w += (int64_t)H * I;
        lw.Wi_p = w; if (H == 384) cvt_pack_w1_kernel<<<1024, 256, 0, e->stream>>>(Wi, w, I); w += (int64_t)H * I;
        lw.Wqkv_p = w; { const int prc = mi_qkv_attn_pack(e->stream, lw.Wqkv, w, H, NH); if (prc != MI_OK) return fail(prc); } w += (int64_t)3 * H * H;
        // fused qkv bias: bq|bk|bv are not contiguous in the blob -> a small packed copy
        lw.bqkv = nullptr;
        hipError_t me = hipMalloc((void**)&lw.bqkv, (size_t)3 * H * sizeof(float));
        lw.bo = bo; lw.bi = bi; lw.bd = bd; lw.g1 = g1; lw.b1 = b1; lw.g2 = g2; lw.b2 = b2;
        e->layers.push_back(lw);                       // owned by e from here on (encoder_free releases bqkv)
        ENC_HIP(me);
        if (H == 384 && I == 1536) {
            LayerW& L = e->layers.back();
            ENC_HIP(hipMalloc(&L.w3, mi_ffn3_stream_bytes()));
            ENC_HIP(hipMalloc((void**)&L.p3, mi_ffn3_param_floats() * sizeof(float)));
            const int prc = mi_ffn3_pack(e->stream, Wo, Wi, Wd, L.w3, bo, g1, b1, bi, bd, g2, b2, L.p3);
            if (prc != MI_OK) return fail(prc);
        }
        scale_f32_kernel<<<(unsigned)((H + 255) / 256), 256, 0, e->stream>>>(bq, lw.bqkv, H, qscale);
        ENC_HIP(hipMemcpyAsync(lw.bqkv + H, bk, (size_t)H * 4, hipMemcpyDeviceToDevice, e->stream));
        ENC_HIP(hipMemcpyAsync(lw.bqkv + 2 * H, bv, (size_t)H * 4, hipMemcpyDeviceToDevice, e->stream));
    }
    ENC_HIP(hipGetLastError());
    ENC_HIP(hipStreamSynchronize(e->stream));
    ENC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
#undef ENC_HIP
    *out = e;
    return MI_OK;
}

extern "C" void mi_encoder_destroy(mi_encoder* e)
{
    if (!e) return;
    DeviceGuard guard(e->device);
    (void)hipStreamSynchronize(e->stream);
    encoder_free(e);
}

// Cross-encoder head: pooler_w [H][H], pooler_b [H], cls_w [num_labels][H], cls_b [num_labels], host fp32.
extern "C" int mi_encoder_set_head(mi_encoder* e, const float* pooler_w, const float* pooler_b,
                                   const float* cls_w, const float* cls_b, int num_labels)
{
    MI_REQUIRE(e && pooler_w && pooler_b && cls_w && cls_b, "mi_encoder_set_head: NULL argument");
    MI_REQUIRE(num_labels >= 1 && num_labels <= 64, "mi_encoder_set_head: num_labels %d outside [1, 64]", num_labels);
    std::lock_guard<std::mutex> lk(e->mu);
    DeviceGuard guard(e->device);
    const size_t H = (size_t)e->cfg.hidden_size;
    const size_t n = H * H + H + (size_t)num_labels * H + (size_t)num_labels;
    MI_HIP(hipStreamSynchronize(e->stream));
    if (e->head) { (void)hipFree(e->head); e->head = nullptr; e->nlabels = 0; }
    MI_HIP(hipMalloc((void**)&e->head, n * sizeof(float)));
    float* d = e->head;
    std::vector<float> wT(H * H);                    // [in][out]: cls_head_kernel's threads own outputs
    for (size_t o = 0; o < H; ++o) for (size_t c = 0; c < H; ++c) wT[c * H + o] = pooler_w[o * H + c];
    MI_HIP(hipMemcpyAsync(d, wT.data(), H * H * 4, hipMemcpyHostToDevice, e->stream)); d += H * H;
    MI_HIP(hipMemcpyAsync(d, pooler_b, H * 4, hipMemcpyHostToDevice, e->stream)); d += H;
    MI_HIP(hipMemcpyAsync(d, cls_w, (size_t)num_labels * H * 4, hipMemcpyHostToDevice, e->stream)); d += (size_t)num_labels * H;
    MI_HIP(hipMemcpyAsync(d, cls_b, (size_t)num_labels * 4, hipMemcpyHostToDevice, e->stream));
    MI_HIP(hipStreamSynchronize(e->stream));
    e->nlabels = num_labels;
    return MI_OK;
}

// K = 384 projections on enough rows to fill the chip: the weight-stationary kernel
template <int EPI>
static bool launch_gemm_ws(hipStream_t st, const uint16_t* A, const uint16_t* W, const float* bias, const uint16_t* resid,
                           uint16_t* C, int M, int N, int K, bool blocked_out)
{
    static const bool off = getenv("MI_GEMM_WS") && atoi(getenv("MI_GEMM_WS")) == 0;
    // N = 384 (attention output) fills only 1.5 column groups: measured slower than the tile kernel
    if (off || EPI == EPI_RESID || K != 384 || N % 32 != 0 || N < 512 || M < 16384) return false;
    const int n_full = N / 256, has_tail = (N % 256) ? 1 : 0;
    const int n_groups = n_full + has_tail;
    if (n_groups > 32 || n_full < 1) return false;
    const int m_tiles = (M + WS_TILE_R - 1) / WS_TILE_R;
    // 32 CUs per XCD: x chunks for each full group, y for the tail group, n_full x + y <= 32, minimising the
    // longer of the two (a tail workgroup takes about 0.6 of the time per row tile: same stream, half the MFMAs)
    int cf = 32 / n_groups, ct = has_tail ? cf : 0;
    if (has_tail) {
        double best = 1.0 / cf;
        for (int x = cf; x <= 32; ++x) {
            const int y = 32 - n_full * x;
            if (y < 1) break;
            const double cost = (1.0 / x > 0.6 / y) ? 1.0 / x : 0.6 / y;
            if (cost < best - 1e-9) { best = cost; cf = x; ct = y; }
        }
    }
    if (cf > m_tiles / 8) { cf = m_tiles / 8; if (has_tail && ct > cf) ct = cf; }
    if (cf < 1 || (has_tail && ct < 1)) return false;
    if constexpr (EPI != EPI_RESID) {
        if (blocked_out) gemm_ws_kernel<EPI, true><<<256, 512, 0, st>>>(A, W, bias, C, M, N, n_full, cf, ct);
        else gemm_ws_kernel<EPI, false><<<256, 512, 0, st>>>(A, W, bias, C, M, N, n_full, cf, ct);
    }
    return true;
}

// ---------------------------------------------------------------- skinny GEMM: M <= 128 rows (a query, a few chunks)
// C[M][N] = epilogue(A[M][K] W[N][K]^T + bias (+ resid)) for the call sizes of embed_query (advanced_rag.py:296 embeds the
// query before every search): the general kernel's 128 x 128 tiles leave a 16-token query with 3 to 12 workgroups that
// each walk K in dependent stages (6-11 us per GEMM, 214 of the 280 us of a query).  Here a wave owns 16 output columns
// and a K range of 384: its W fragments (12 k-steps) and the x fragments of 32 rows are requested in one go, straight into
// registers -- one memory round trip, then 12-24 MFMAs.  K = 384: the four waves of a workgroup take four column blocks;
// K = 1536 (KSPLIT = 4): they share one column block, each a quarter of K, and wave 0 adds the partial sums through LDS.
// More than 32 rows: blockIdx.y walks the groups of 32 rows, side by side (W comes from L2 again; still one round trip).
template <int EPI, int KSPLIT>
__global__ __launch_bounds__(256) void gemm_skinny_kernel(const uint16_t* __restrict__ A, const uint16_t* __restrict__ W,
        const float* __restrict__ bias, const uint16_t* __restrict__ resid, uint16_t* __restrict__ C, int M, int N, int K)
{
    __shared__ float red[3][2][64][4];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l15 = lane & 15, g4 = lane >> 4;
    const int n0 = KSPLIT == 1 ? ((int)blockIdx.x * 4 + wave) * 16 : (int)blockIdx.x * 16;
    const int k0 = KSPLIT == 1 ? 0 : wave * 384;
    bf16x8 wf[12];
    {
        const uint16_t* wp = W + (int64_t)(n0 + l15) * K + k0 + 8 * g4;
#pragma unroll
        for (int s = 0; s < 12; ++s) wf[s] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(wp + 32 * s));
    }
    {
        const int m0 = (int)blockIdx.y * 32;                        // 32 rows per workgroup row: all row groups run side by side
        const bool two = M - m0 > 16;                               // (uniform) a second block of 16 rows
        bf16x8 xf[2][12];
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
            if (mb == 0 || two) {
                int m = m0 + 16 * mb + l15; if (m > M - 1) m = M - 1;
                const uint16_t* xp = A + (int64_t)m * K + k0 + 8 * g4;
#pragma unroll
                for (int s = 0; s < 12; ++s) xf[mb][s] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(xp + 32 * s));
            }
        }
        f32x4 acc[2] = {{0.0f, 0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f, 0.0f}};
#pragma unroll
        for (int s = 0; s < 12; ++s) {
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[s], xf[0][s], acc[0], 0, 0, 0);
            if (two) acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[s], xf[1][s], acc[1], 0, 0, 0);
        }
        if constexpr (KSPLIT > 1) {
            if (wave > 0) {
                *reinterpret_cast<f32x4*>(&red[wave - 1][0][lane][0]) = acc[0];
                *reinterpret_cast<f32x4*>(&red[wave - 1][1][lane][0]) = acc[1];
            }
            __syncthreads();
            if (wave == 0) {
#pragma unroll
                for (int w = 0; w < 3; ++w) {
                    acc[0] += *reinterpret_cast<const f32x4*>(&red[w][0][lane][0]);
                    acc[1] += *reinterpret_cast<const f32x4*>(&red[w][1][lane][0]);
                }
            }
        }
        if (KSPLIT == 1 || wave == 0) {
            // lane (l15, g4): row m0 + 16 mb + l15, columns n0 + 4 g4 .. + 3
            const float4 bv = *reinterpret_cast<const float4*>(bias + n0 + 4 * g4);
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
                const int m = m0 + 16 * mb + l15;
                if ((mb == 0 || two) && m < M) {
                    float v0 = acc[mb][0] + bv.x, v1 = acc[mb][1] + bv.y, v2 = acc[mb][2] + bv.z, v3 = acc[mb][3] + bv.w;
                    if constexpr (EPI == EPI_GELU) gelu_erf4(v0, v1, v2, v3);
                    if constexpr (EPI == EPI_RESID) {
                        const uint2 rv = *reinterpret_cast<const uint2*>(resid + (int64_t)m * N + n0 + 4 * g4);
                        v0 += __builtin_bit_cast(float, rv.x << 16); v1 += __builtin_bit_cast(float, rv.x & 0xFFFF0000u);
                        v2 += __builtin_bit_cast(float, rv.y << 16); v3 += __builtin_bit_cast(float, rv.y & 0xFFFF0000u);
                    }
                    *reinterpret_cast<uint2*>(C + (int64_t)m * N + n0 + 4 * g4) = make_uint2(pack2(v0, v1), pack2(v2, v3));
                }
            }
        }
    }
}

template <int EPI>
static bool launch_gemm(hipStream_t st, const uint16_t* A, const uint16_t* W, const float* bias, const uint16_t* resid,
                        uint16_t* C, int M, int N, int K, bool want_blocked = false)
{
    // returns true iff C was written in the blocked layout (only the weight-stationary kernel can)
    if (launch_gemm_ws<EPI>(st, A, W, bias, resid, C, M, N, K, want_blocked)) return want_blocked;
    // a query or a few short chunks: one memory round trip per GEMM instead of a walk over K (MI_GEMM_SKINNY=0: the general kernel)
    static const bool skinny_off = getenv("MI_GEMM_SKINNY") && atoi(getenv("MI_GEMM_SKINNY")) == 0;
    static const int skinny_rows = getenv("MI_GEMM_SKINNY_ROWS") ? atoi(getenv("MI_GEMM_SKINNY_ROWS")) : 384;
    if (!skinny_off && M <= skinny_rows && N % 64 == 0) {
        const unsigned gy = (unsigned)((M + 31) / 32);
        if (K == 384) { gemm_skinny_kernel<EPI, 1><<<dim3((unsigned)(N / 64), gy), 256, 0, st>>>(A, W, bias, resid, C, M, N, K); return false; }
        if (K == 1536) { gemm_skinny_kernel<EPI, 4><<<dim3((unsigned)(N / 16), gy), 256, 0, st>>>(A, W, bias, resid, C, M, N, K); return false; }
    }
    const int n_tiles = N / GM_BN, m_tiles = (M + GM_BM - 1) / GM_BM;
    const unsigned grid = (unsigned)(((m_tiles + 7) / 8) * 8 * n_tiles);
    // a grid that cannot put four workgroups on every CU takes the ring-of-four form (MI_GEMM_RING=0: never)
    static const bool ring_off = getenv("MI_GEMM_RING") && atoi(getenv("MI_GEMM_RING")) == 0;
    if (!ring_off && m_tiles * n_tiles <= 512)
        gemm_bf16_kernel<EPI, 4><<<grid, 256, 0, st>>>(A, W, bias, resid, C, M, N, K, n_tiles, m_tiles);
    else
        gemm_bf16_kernel<EPI, 1><<<grid, 256, 0, st>>>(A, W, bias, resid, C, M, N, K, n_tiles, m_tiles);
    return false;
}

// Bytes of activation workspace one pass over T tokens needs: h, tmp [T][H]; ctx [Tp][H]; qkv [Tp][3H]; mid [Tp][I] (bf16)
// + 128 rows of slack: the weight-stationary GEMM streams whole 128-row tiles of its A operand.
// ctx, qkv and mid may be written in the blocked [rows / 128][cols / 32][128][32] layout, which covers whole
// 128-row tiles: each of them gets roundup(T, 128) rows, so a tile of one buffer never reaches into the next
// (the attention kernel writes ctx while other workgroups still read qkv).
static size_t encoder_ws_need(const mi_bert_config& c, int64_t T)
{
    const size_t H = (size_t)c.hidden_size, I = (size_t)c.intermediate_size;
    const size_t Tp = ((size_t)T + 127) / 128 * 128;
    return ((size_t)T * 2 * H + Tp * (H + 3 * H + I) + (size_t)128 * I) * sizeof(uint16_t);
}

// The encoder's own grow-only workspace, sized for a pass over T tokens.
static int ensure_ws(mi_encoder* e, int64_t T, hipStream_t st, int lane = 0)
{
    mi_encoder::Lane& n = e->ln[lane];
    const size_t need = encoder_ws_need(e->cfg, T);
    if (need <= n.ws_bytes) return MI_OK;
    if (n.ws) {
        // the lane's last pass may still be running on another stream than this one
        if (n.have_last) MI_HIP(hipEventSynchronize(n.ev1));
        MI_HIP(hipStreamSynchronize(st));
        (void)hipFree(n.ws); n.ws = nullptr; n.ws_bytes = 0;
    }
    hipError_t er = hipMalloc(&n.ws, need);
    if (er != hipSuccess) { n.ws = nullptr; mi_set_error("encode: hipMalloc(%zu) for the workspace failed: %s", need, hipGetErrorString(er)); return MI_ERR_OOM; }
    n.ws_bytes = need;
    return MI_OK;
}

// One pass: all device pointers, B*L tokens.  Caller holds the encoder mutex and has set the device.
// types: token type ids or nullptr (all 0).  out_emb: pooled sentence embeddings [B][H], or nullptr;
// out_logits: cross-encoder logits [B][nlabels], or nullptr.  ws_ext: caller-supplied workspace of ws_ext_bytes
// (the allocation-free entry point), or nullptr: the encoder's own grow-only workspace.
// One pass over B sequences.  Padded: ids / types / mask are [B][L].  Packed (seq_start != nullptr): ids and pos_ids hold the
// T_packed tokens of the sequences back to back, seq_start[b] .. seq_start[b + 1] are sequence b's, L is the longest, mask
// and types are unused -- no padding token is computed anywhere (mi_encoder_encode_ragged).
static int encode_pass_locked(mi_encoder* e, int64_t B, int L, const int32_t* ids, const int32_t* types, const int32_t* mask,
                              float* out_emb, float* out_logits, hipStream_t st, void* ws_ext, size_t ws_ext_bytes,
                              const int32_t* pos_ids = nullptr, const int32_t* seq_start = nullptr, int64_t T_packed = 0, int lane = 0)
{
    mi_encoder::Lane& ln = e->ln[lane];
    const mi_bert_config& c = e->cfg;
    const int H = c.hidden_size, I = c.intermediate_size, NH = c.num_attention_heads;
    const int64_t T = seq_start ? T_packed : B * L;
    MI_REQUIRE(T < ((int64_t)1 << 31) / 16, "encode: too many tokens in one call (%lld)", (long long)T);
    const size_t need = encoder_ws_need(c, T);
    // Under HIP graph capture (mi_encoder_encode_dev_noalloc's callers) nothing may be recorded on or waited for from outside
    // the capture: the timing events would become graph nodes and a later mi_encoder_last_time would fail on them.
    hipStreamCaptureStatus cap_status = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(st, &cap_status);
    const bool capturing = cap_status != hipStreamCaptureStatusNone;
    // One workspace per lane (and one `scratch` array) serves every pass of this encoder; passes on one stream serialise by
    // themselves, a pass on another stream than the lane's previous one waits for that one's end.
    if (!capturing && ln.have_last && ln.last_st != st) MI_HIP(hipStreamWaitEvent(st, ln.ev1, 0));
    void* wsp = nullptr;
    if (ws_ext) {
        if (need > ws_ext_bytes) {
            mi_set_error("encode: %lld tokens need %zu bytes of workspace, the allocation-free entry point holds %zu", (long long)T, need, ws_ext_bytes);
            return MI_ERR_UNSUPPORTED;
        }
        wsp = ws_ext;
    } else {
        MI_TRY(ensure_ws(e, T, st, lane));
        wsp = ln.ws;
    }
    const size_t Tp = ((size_t)T + 127) / 128 * 128;
    uint16_t* h = (uint16_t*)wsp;
    uint16_t* tmp = h + (size_t)T * H;
    uint16_t* ctx = tmp + (size_t)T * H;
    uint16_t* qkv = ctx + Tp * H;
    uint16_t* mid = qkv + Tp * 3 * H;

    const unsigned tok_grid = (unsigned)((T + 3) / 4);
    const int Lp = (L + 31) / 32 * 32;
    const size_t attn_smem = (size_t)Lp * 128 + (size_t)Lp * 4;      // K rows, V rows, additive mask
    if (attn_smem > 96 * 1024) { mi_set_error("encode: sequence length %d too long for the attention kernel", L); return MI_ERR_UNSUPPORTED; }

    // hidden size 384: the two N = 384 projections run fused with their LayerNorm (MI_GEMM_LN=0: separate kernels)
    static const bool ln_off = getenv("MI_GEMM_LN") && atoi(getenv("MI_GEMM_LN")) == 0;
    // The fused kernels own a 128-row tile each and walk all of K (and I) inside one workgroup: built for bulk
    // embedding, where there are hundreds of tiles.  A query or a few chunks fill one to a few dozen of the 256
    // CUs and wait 97 us per layer on one workgroup's pipeline; the general GEMMs spread the same tile over the
    // output columns instead.  Measured (MiniLM-L6, device time per call): 16 tokens 645 -> 357 us, 4,096 tokens
    // 737 -> 517, 16,384 tokens 885 vs 880, 65,536 tokens 2,065 vs 2,598.  MI_ENC_LATENCY_TOKENS moves the
    // hand-over (0: always the bulk kernels); read per call so a test can flip it.
    int64_t latency_tokens = 12288;
    if (const char* s = getenv("MI_ENC_LATENCY_TOKENS")) latency_tokens = atoll(s);
    const bool bulk = T >= latency_tokens;
    const bool fused_ln = !ln_off && bulk && H == LN_N && I % 64 == 0;
    static const bool ffn_off = getenv("MI_FFN_FUSED") && atoi(getenv("MI_FFN_FUSED")) == 0;
    const bool fused_ffn = fused_ln && !ffn_off && I == 1536;
    static const bool front_off = getenv("MI_FFN_FRONT") && atoi(getenv("MI_FFN_FRONT")) == 0;
    const bool fused_front = fused_ffn && !front_off;
    // MI_FFN3=0: the round-2 kernel (ffn2_kernel) instead of the token-on-the-lane one (encoder_ffn3.hip); read per call (A/B)
    const bool ffn3 = fused_front && !(getenv("MI_FFN3") && atoi(getenv("MI_FFN3")) == 0);
    if (!capturing) MI_HIP(hipEventRecord(ln.ev0, st));
    if (H <= 512) embed_ln_kernel<1><<<tok_grid, 256, 0, st>>>(ids, types, pos_ids, c.type_vocab_size, T, L, H, c.vocab_size, e->word, e->pos, e->type0, e->eg, e->eb, c.layer_norm_eps, h);
    else embed_ln_kernel<2><<<tok_grid, 256, 0, st>>>(ids, types, pos_ids, c.type_vocab_size, T, L, H, c.vocab_size, e->word, e->pos, e->type0, e->eg, e->eb, c.layer_norm_eps, h);
    // bulk passes of sequences up to 256 tokens: QKV projection and attention in one kernel (encoder_qkv_attn.hip), the QKV
    // tensor never goes to memory.  MI_QKV_FUSED=0: the two kernels; read per call (A/B, tests)
    // The fused kernel streams a head's weight slice once per sequence whatever its length: passes of short sequences stay
    // with the two kernels (mean length below MI_QKV_FUSED_MIN_MEAN tokens).
    int64_t fused_min_mean = 128;
    if (const char* s = getenv("MI_QKV_FUSED_MIN_MEAN")) fused_min_mean = atoll(s);
    const bool qkv_fused = bulk && L <= MI_QKV_ATTN_MAX_L && B <= 65535 && T >= B * fused_min_mean &&
                           !(getenv("MI_QKV_FUSED") && atoi(getenv("MI_QKV_FUSED")) == 0);
    for (const LayerW& lw : e->layers) {
        if (qkv_fused) {
            MI_TRY(mi_launch_qkv_attn(st, h, lw.Wqkv_p, lw.bqkv, mask, seq_start, ctx, B, L, H, NH, fused_ln ? 1 : 0));
        } else {
            const bool qkv_blocked = launch_gemm<EPI_BIAS>(st, h, lw.Wqkv, lw.bqkv, nullptr, qkv, (int)T, 3 * H, H, true);
            attn_kernel<<<dim3(NH, (unsigned)B, (L + 255) / 256), 256, attn_smem, st>>>(qkv, mask, seq_start, ctx, L, Lp, H, qkv_blocked ? 1 : 0, fused_ln ? 1 : 0);
        }
        if (ffn3 && lw.w3) {
            MI_TRY(mi_launch_ffn3(st, ctx, h, lw.w3, lw.p3, c.layer_norm_eps, h, (int)T));
            continue;
        }
        if (fused_front) {
            ffn2_kernel<true><<<(unsigned)((T + 127) / 128), 512, 0, st>>>(h, lw.Wi_p, lw.bi, lw.Wd_p, lw.bd, lw.g2, lw.b2, c.layer_norm_eps, h, (int)T,
                                                                           ctx, lw.Wo_p, lw.bo, lw.g1, lw.b1);
            continue;
        }
        if (fused_ln) {
            gemm_ln_kernel<<<(unsigned)((T + 127) / 128), 512, 0, st>>>(ctx, lw.Wo_p, lw.bo, h, lw.g1, lw.b1, c.layer_norm_eps, h, (int)T, H, 1);
        } else {
            launch_gemm<EPI_RESID>(st, ctx, lw.Wo, lw.bo, h, tmp, (int)T, H, H);
            if (H <= 512) add_ln_kernel<1><<<tok_grid, 256, 0, st>>>(tmp, T, H, lw.g1, lw.b1, c.layer_norm_eps, h);
            else add_ln_kernel<2><<<tok_grid, 256, 0, st>>>(tmp, T, H, lw.g1, lw.b1, c.layer_norm_eps, h);
        }
        if (fused_ffn) {
            ffn2_kernel<false><<<(unsigned)((T + 127) / 128), 512, 0, st>>>(h, lw.Wi_p, lw.bi, lw.Wd_p, lw.bd, lw.g2, lw.b2, c.layer_norm_eps, h, (int)T,
                                                                            nullptr, nullptr, nullptr, nullptr, nullptr);
            continue;
        }
        const bool mid_blocked = launch_gemm<EPI_GELU>(st, h, lw.Wi, lw.bi, nullptr, mid, (int)T, I, H, fused_ln);
        if (fused_ln) {
            gemm_ln_kernel<<<(unsigned)((T + 127) / 128), 512, 0, st>>>(mid, lw.Wd_p, lw.bd, h, lw.g2, lw.b2, c.layer_norm_eps, h, (int)T, I, mid_blocked ? 1 : 0);
        } else {
            launch_gemm<EPI_RESID>(st, mid, lw.Wd, lw.bd, h, tmp, (int)T, H, I);
            if (H <= 512) add_ln_kernel<1><<<tok_grid, 256, 0, st>>>(tmp, T, H, lw.g2, lw.b2, c.layer_norm_eps, h);
            else add_ln_kernel<2><<<tok_grid, 256, 0, st>>>(tmp, T, H, lw.g2, lw.b2, c.layer_norm_eps, h);
        }
    }
    if (out_emb) {
        if (H <= 512) pool_kernel<1><<<(unsigned)B, 64 * POOL_WAVES, 0, st>>>(h, mask, seq_start, L, H, c.normalize, out_emb);
        else pool_kernel<2><<<(unsigned)B, 64 * POOL_WAVES, 0, st>>>(h, mask, seq_start, L, H, c.normalize, out_emb);
    }
    if (out_logits) {
        const float* hw = e->head;
        const size_t HH = (size_t)H;
        cls_head_kernel<<<(unsigned)((B + CLS_PB - 1) / CLS_PB), (unsigned)((H + 63) / 64 * 64), 2 * CLS_PB * HH * sizeof(float), st>>>(
            h, seq_start, L, H, hw, hw + HH * HH, hw + HH * HH + HH, hw + HH * HH + HH + (size_t)e->nlabels * HH, e->nlabels, B, out_logits);
    }
    MI_HIP(hipGetLastError());
    if (!capturing) {
        MI_HIP(hipEventRecord(ln.ev1, st));
        e->timed = true;
        e->last_lane = lane;
        ln.last_st = st;
        ln.have_last = true;
    }
    e->st_seqs += B;
    e->st_tokens += T;
    if (bulk) e->st_bulk += 1; else e->st_latency += 1;
    return MI_OK;
}

static int64_t encoder_pass_tokens()
{
    // bounded workspace: at most 64K tokens per pass
    static const int64_t pass_tokens = getenv("MI_ENC_PASS_TOKENS") ? atoll(getenv("MI_ENC_PASS_TOKENS")) : 65536;
    return pass_tokens > 0 ? pass_tokens : 65536;
}

// MI_ENC_LANES=1: every pass on the caller's stream, one after the other (read per call; A/B and debugging)
static bool encoder_two_lanes()
{
    const char* s = getenv("MI_ENC_LANES");
    return !(s && atoi(s) == 1);
}

// Device buffers [B][L] in passes of at most MI_ENC_PASS_TOKENS tokens.  Caller holds the mutex.
// Two passes and more alternate between the encoder's two lanes: pass 2i on the caller's stream, pass 2i + 1 on st2, which
// starts behind everything the caller's stream holds at this point (fork event) and which the caller's stream waits for at
// the end (join event) -- to the caller the call is still one piece of work on its stream.
static int encode_dev_locked(mi_encoder* e, int64_t B, int L, const int32_t* ids, const int32_t* types, const int32_t* mask,
                             float* out_emb, float* out_logits, hipStream_t st)
{
    const int64_t bmax = (encoder_pass_tokens() / L) > 0 ? encoder_pass_tokens() / L : 1;
    hipStreamCaptureStatus cap_status = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(st, &cap_status);
    const bool lanes = B > bmax && cap_status == hipStreamCaptureStatusNone && encoder_two_lanes();
    if (lanes) {
        MI_HIP(hipEventRecord(e->ev_fork, st));
        MI_HIP(hipStreamWaitEvent(e->st2, e->ev_fork, 0));
    }
    int64_t pass = 0;
    for (int64_t b0 = 0; b0 < B; b0 += bmax, ++pass) {
        const int64_t nb = (B - b0 < bmax) ? B - b0 : bmax;
        const int lane = lanes ? (int)(pass & 1) : 0;
        MI_TRY(encode_pass_locked(e, nb, L, ids + b0 * L, types ? types + b0 * L : nullptr, mask + b0 * L,
                                  out_emb ? out_emb + b0 * e->cfg.hidden_size : nullptr,
                                  out_logits ? out_logits + b0 * e->nlabels : nullptr, lane ? e->st2 : st, nullptr, 0, nullptr, nullptr, 0, lane));
    }
    if (lanes) {
        MI_HIP(hipEventRecord(e->ev_join, e->st2));
        MI_HIP(hipStreamWaitEvent(st, e->ev_join, 0));
    }
    return MI_OK;
}

extern "C" int mi_encoder_encode_dev(mi_encoder* e, int64_t B, int L, const int32_t* ids_dev, const int32_t* mask_dev,
                                     float* out_emb_dev, void* stream)
{
    MI_REQUIRE(e && B >= 0 && L > 0, "mi_encoder_encode_dev: bad argument");
    MI_REQUIRE(L <= e->cfg.max_position_embeddings, "mi_encoder_encode_dev: L=%d exceeds max_position_embeddings=%d", L, e->cfg.max_position_embeddings);
    if (B == 0) return MI_OK;
    MI_REQUIRE(ids_dev && mask_dev && out_emb_dev, "mi_encoder_encode_dev: NULL buffer");
    std::lock_guard<std::mutex> lk(e->mu);
    DeviceGuard guard(e->device);
    return encode_dev_locked(e, B, L, ids_dev, nullptr, mask_dev, out_emb_dev, nullptr, mi_pick_stream(stream, e->stream));
}

// The allocation-free variant of mi_encoder_encode_dev: the same kernels, the activation workspace taken from the
// file-scope `scratch` array of this device instead of hipMalloc.  One pass; inputs whose workspace exceeds
// MI_ENC_NOALLOC_BYTES are rejected.  Nothing in the workspace has to start at zero (every kernel writes what a
// later one reads; rows past the last token are read as padding of whole tiles and never stored), so nothing is
// cleared.  Every call on a GPU uses that GPU's one array: calls that overlap on one GPU -- two encoders, or one
// encoder on two streams -- race; serialise them.
extern "C" int mi_encoder_encode_dev_noalloc(mi_encoder* e, int64_t B, int L, const int32_t* ids_dev, const int32_t* mask_dev,
                                             float* out_emb_dev, void* stream)
{
    MI_REQUIRE(e && B >= 0 && L > 0, "mi_encoder_encode_dev_noalloc: bad argument");
    MI_REQUIRE(L <= e->cfg.max_position_embeddings, "mi_encoder_encode_dev_noalloc: L=%d exceeds max_position_embeddings=%d", L, e->cfg.max_position_embeddings);
    if (B == 0) return MI_OK;
    MI_REQUIRE(ids_dev && mask_dev && out_emb_dev, "mi_encoder_encode_dev_noalloc: NULL buffer");
    MI_REQUIRE(B <= ((int64_t)1 << 20), "mi_encoder_encode_dev_noalloc: B=%lld out of range", (long long)B);
    const size_t need = encoder_ws_need(e->cfg, B * L);
    if (need > MI_ENC_NOALLOC_BYTES) {
        mi_set_error("mi_encoder_encode_dev_noalloc: %lld tokens need %zu bytes of workspace; this entry point holds %zu "
                     "(use mi_encoder_encode_dev for larger batches)", (long long)(B * L), need, (size_t)MI_ENC_NOALLOC_BYTES);
        return MI_ERR_UNSUPPORTED;
    }
    std::lock_guard<std::mutex> lk(e->mu);
    DeviceGuard guard(e->device);
    return encode_pass_locked(e, B, L, ids_dev, nullptr, mask_dev, out_emb_dev, nullptr, mi_pick_stream(stream, e->stream),
                              e->scratch_dev, MI_ENC_NOALLOC_BYTES);
}

extern "C" int mi_encoder_classify_dev(mi_encoder* e, int64_t B, int L, const int32_t* ids_dev, const int32_t* type_ids_dev,
                                       const int32_t* mask_dev, float* out_logits_dev, void* stream)
{
    MI_REQUIRE(e && B >= 0 && L > 0, "mi_encoder_classify_dev: bad argument");
    MI_REQUIRE(L <= e->cfg.max_position_embeddings, "mi_encoder_classify_dev: L=%d exceeds max_position_embeddings=%d", L, e->cfg.max_position_embeddings);
    if (B == 0) return MI_OK;
    MI_REQUIRE(ids_dev && mask_dev && out_logits_dev, "mi_encoder_classify_dev: NULL buffer");
    std::lock_guard<std::mutex> lk(e->mu);
    if (!e->head) { mi_set_error("mi_encoder_classify_dev: no classification head loaded (mi_encoder_set_head)"); return MI_ERR_INVALID_ARG; }
    DeviceGuard guard(e->device);
    return encode_dev_locked(e, B, L, ids_dev, type_ids_dev, mask_dev, nullptr, out_logits_dev, mi_pick_stream(stream, e->stream));
}

// Device staging of the host-pointer entry points (caller holds the mutex): room for three [tokens] int32 planes
// and `out_bytes` of results.
static int ensure_io(mi_encoder* e, size_t tokens, size_t out_bytes, hipStream_t st, int lane = 0)
{
    mi_encoder::Lane& n = e->ln[lane];
    const size_t nio = 3 * tokens * sizeof(int32_t);
    if (nio > n.io_cap) {
        MI_HIP(hipStreamSynchronize(st));
        if (n.d_io) (void)hipFree(n.d_io);
        n.d_io = nullptr; n.io_cap = 0;
        hipError_t er = hipMalloc((void**)&n.d_io, nio);
        if (er != hipSuccess) { n.d_io = nullptr; mi_set_error("encode: hipMalloc(%zu) failed: %s", nio, hipGetErrorString(er)); return MI_ERR_OOM; }
        n.io_cap = nio;
    }
    if (out_bytes > n.out_cap) {
        MI_HIP(hipStreamSynchronize(st));
        if (n.d_out) (void)hipFree(n.d_out);
        n.d_out = nullptr; n.out_cap = 0;
        hipError_t er = hipMalloc((void**)&n.d_out, out_bytes);
        if (er != hipSuccess) { n.d_out = nullptr; mi_set_error("encode: hipMalloc(%zu) failed: %s", out_bytes, hipGetErrorString(er)); return MI_ERR_OOM; }
        n.out_cap = out_bytes;
    }
    return MI_OK;
}

static int check_ids(const char* who, const mi_encoder* e, const int32_t* ids, int64_t n)
{
    for (int64_t i = 0; i < n; ++i)
        if (ids[i] < 0 || ids[i] >= e->cfg.vocab_size) {
            mi_set_error("%s: token id %d at position %lld outside [0,%d)", who, ids[i], (long long)i, e->cfg.vocab_size);
            return MI_ERR_INVALID_ARG;
        }
    return MI_OK;
}

// Host [B][L] arrays in, host results out.  The mutex is held from the first staging copy to the last read-back:
// the staging buffers and the stream are the encoder's, so two threads must not interleave on them.
static int encode_host(mi_encoder* e, const char* who, int64_t B, int L, const int32_t* ids, const int32_t* types,
                       const int32_t* mask, float* out, bool logits)
{
    MI_TRY(check_ids(who, e, ids, B * L));
    std::lock_guard<std::mutex> lk(e->mu);
    if (logits && !e->head) { mi_set_error("%s: no classification head loaded (mi_encoder_set_head)", who); return MI_ERR_INVALID_ARG; }
    DeviceGuard guard(e->device);
    hipStream_t st = e->stream;
    const size_t T = (size_t)B * L, nio = T * sizeof(int32_t);
    const size_t width = logits ? (size_t)e->nlabels : (size_t)e->cfg.hidden_size;
    const size_t nout = (size_t)B * width * sizeof(float);
    MI_TRY(ensure_io(e, T, nout, st));
    int32_t* const d_io = e->ln[0].d_io;
    float* const d_out = e->ln[0].d_out;
    int32_t* d_ids = d_io;
    int32_t* d_types = types ? d_io + T : nullptr;
    int32_t* d_mask = d_io + 2 * T;
    // small calls (a query, a few chunks) go through the pinned block: a pageable copy costs a staging hop and a
    // sync inside the runtime each way
    const bool pinned = 3 * nio <= ((size_t)1 << 20) && nout <= ((size_t)1 << 20);
    if (pinned) {
        MI_TRY(e->pin_io[0].ensure((size_t)1 << 20));
        MI_TRY(e->pin_out[0].ensure((size_t)1 << 20));
        char* pb = static_cast<char*>(e->pin_io[0].p);
        memcpy(pb, ids, nio);
        memcpy(pb + nio, mask, nio);
        if (types) memcpy(pb + 2 * nio, types, nio);
        MI_HIP(hipMemcpyAsync(d_ids, pb, nio, hipMemcpyHostToDevice, st));
        MI_HIP(hipMemcpyAsync(d_mask, pb + nio, nio, hipMemcpyHostToDevice, st));
        if (types) MI_HIP(hipMemcpyAsync(d_types, pb + 2 * nio, nio, hipMemcpyHostToDevice, st));
    } else {
        MI_HIP(hipMemcpyAsync(d_ids, ids, nio, hipMemcpyHostToDevice, st));
        MI_HIP(hipMemcpyAsync(d_mask, mask, nio, hipMemcpyHostToDevice, st));
        if (types) MI_HIP(hipMemcpyAsync(d_types, types, nio, hipMemcpyHostToDevice, st));
    }
    MI_TRY(encode_dev_locked(e, B, L, d_ids, d_types, d_mask, logits ? nullptr : d_out, logits ? d_out : nullptr, st));
    if (pinned) {
        MI_HIP(hipMemcpyAsync(e->pin_out[0].p, d_out, nout, hipMemcpyDeviceToHost, st));
        MI_HIP(hipStreamSynchronize(st));
        memcpy(out, e->pin_out[0].p, nout);
    } else {
        MI_HIP(hipMemcpyAsync(out, d_out, nout, hipMemcpyDeviceToHost, st));
        MI_HIP(hipStreamSynchronize(st));
    }
    return MI_OK;
}

extern "C" int mi_encoder_encode(mi_encoder* e, int64_t B, int L, const int32_t* ids, const int32_t* mask, float* out_emb)
{
    MI_REQUIRE(e && B >= 0 && L > 0, "mi_encoder_encode: bad argument");
    if (B == 0) return MI_OK;
    MI_REQUIRE(ids && mask && out_emb, "mi_encoder_encode: NULL buffer");
    MI_REQUIRE(L <= e->cfg.max_position_embeddings, "mi_encoder_encode: L=%d exceeds max_position_embeddings=%d", L, e->cfg.max_position_embeddings);
    return encode_host(e, "mi_encoder_encode", B, L, ids, nullptr, mask, out_emb, false);
}

extern "C" int mi_encoder_classify(mi_encoder* e, int64_t B, int L, const int32_t* ids, const int32_t* type_ids,
                                   const int32_t* mask, float* out_logits)
{
    MI_REQUIRE(e && B >= 0 && L > 0, "mi_encoder_classify: bad argument");
    if (B == 0) return MI_OK;
    MI_REQUIRE(ids && mask && out_logits, "mi_encoder_classify: NULL buffer");
    MI_REQUIRE(L <= e->cfg.max_position_embeddings, "mi_encoder_classify: L=%d exceeds max_position_embeddings=%d", L, e->cfg.max_position_embeddings);
    if (type_ids)
        for (int64_t i = 0; i < B * L; ++i)
            if (type_ids[i] < 0 || type_ids[i] >= e->cfg.type_vocab_size) {
                mi_set_error("mi_encoder_classify: token type %d at position %lld outside [0,%d)", type_ids[i], (long long)i, e->cfg.type_vocab_size);
                return MI_ERR_INVALID_ARG;
            }
    return encode_host(e, "mi_encoder_classify", B, L, ids, type_ids, mask, out_logits, true);
}

// Ragged bulk embedding: n sequences as one flat token array + offsets (sequence i = flat_ids[offsets[i] ..
// offsets[i + 1])), results in input order.  This is sentence-transformers' encode loop moved below the boundary:
// sort by length (longest first), cut into passes by a TOKEN budget (MI_ENC_PASS_TOKENS, 64K) instead of a fixed 32 rows, so
// bulk calls reach the fused per-tile kernels whatever the length mix -- and, where sentence-transformers pads a batch to its
// longest sequence, the tokens of a pass are PACKED back to back: the GEMM / LayerNorm / FFN kernels work on token rows and
// never see sequence boundaries, attention and pooling take the sequence starts, the embedding lookup takes the positions.
// No padding token is computed (the padded form of a uniform 32..256 mix carried 8.9 % of them).  Passes are staged through two pinned blocks: the host packs pass i + 1 (and unpacks
// pass i - 1) while the GPU runs pass i.
// `flat_types` != nullptr: cross-encoder pairs -- token types ride along, the result is the classification head's logits
// ([n][num_labels]); sequences are never truncated or empty there (the tokenizer above the boundary owns truncation: it
// must keep the closing [SEP]).
static int ragged_run(mi_encoder* e, const char* who, int64_t n, const int32_t* flat_ids, const int32_t* flat_types,
                      const int64_t* offsets, int max_len, int max_batch, float* out_emb)
{
    const bool logits = flat_types != nullptr;
    const int Lcap = (max_len > 0 && max_len < e->cfg.max_position_embeddings) ? max_len : e->cfg.max_position_embeddings;
    std::vector<int32_t> lens((size_t)n);
    for (int64_t i = 0; i < n; ++i) {
        const int64_t len = offsets[i + 1] - offsets[i];
        MI_REQUIRE(offsets[i] >= 0 && len >= 0, "%s: offsets must be non-decreasing (sequence %lld)", who, (long long)i);
        if (logits) {
            MI_REQUIRE(len >= 1 && len <= e->cfg.max_position_embeddings, "%s: pair %lld has %lld tokens (1 .. %d allowed)", who,
                       (long long)i, (long long)len, e->cfg.max_position_embeddings);
            for (int64_t c = 0; c < len; ++c)
                MI_REQUIRE(flat_types[offsets[i] + c] >= 0 && flat_types[offsets[i] + c] < e->cfg.type_vocab_size,
                           "%s: token type %d outside [0,%d)", who, flat_types[offsets[i] + c], e->cfg.type_vocab_size);
        }
        lens[(size_t)i] = (int32_t)(len < Lcap ? len : Lcap);               // truncation, as the tokenizer's max_length would
        MI_TRY(check_ids(who, e, flat_ids + offsets[i], lens[(size_t)i]));
    }
    std::vector<int64_t> order((size_t)n);
    for (int64_t i = 0; i < n; ++i) order[(size_t)i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int64_t a, int64_t b) { return lens[(size_t)a] > lens[(size_t)b]; });

    std::lock_guard<std::mutex> lk(e->mu);
    if (logits && !e->head) { mi_set_error("%s: no classification head loaded (mi_encoder_set_head)", who); return MI_ERR_INVALID_ARG; }
    const int H = logits ? e->nlabels : e->cfg.hidden_size;       // floats per result row
    DeviceGuard guard(e->device);
    hipStream_t st = e->stream;
    const int64_t budget = encoder_pass_tokens();
    const int64_t bcap = (max_batch > 0 && max_batch < 4096) ? max_batch : 4096;          // rows per pass (bounds the pinned blocks)
    // MI_ENC_VARLEN=0: pad every pass to its longest sequence (the round-2 start form) instead of packing the tokens
    static const bool varlen_env = !(getenv("MI_ENC_VARLEN") && atoi(getenv("MI_ENC_VARLEN")) == 0);
    const bool varlen = varlen_env || logits;                       // (pairs are always packed)
    struct Pass { int64_t first, B; int L; int64_t T; };
    // cut the passes first: every buffer is then sized once, for the largest of them.  Packed: a pass takes sequences while
    // their TRUE token count fits the budget; padded: B x longest.  An empty sequence is given one token (padded: masked;
    // packed: its embedding is overwritten with zeros on the way out).
    std::vector<Pass> passes;
    size_t Tmax = 0, Bmax = 0;
    // packed passes are balanced: ceil(total / budget) passes of about equal token counts, so the last one is no stub that
    // would drop to the small-call kernels (a pass may exceed the budget by less than one sequence)
    int64_t total = 0;
    for (int64_t i = 0; i < n; ++i) total += lens[(size_t)i] > 0 ? lens[(size_t)i] : 1;
    const int64_t npass = (total + budget - 1) / budget;
    const int64_t budget_eff = (total + npass - 1) / npass;
    for (int64_t next = 0; next < n;) {
        int L = lens[(size_t)order[(size_t)next]];
        if (L < 1) L = 1;
        int64_t B = 0, T = 0;
        if (varlen) {
            while (next + B < n && B < bcap) {
                const int len = lens[(size_t)order[(size_t)(next + B)]] > 0 ? lens[(size_t)order[(size_t)(next + B)]] : 1;
                if (B > 0 && T >= budget_eff) break;
                T += len; ++B;
            }
        } else {
            B = budget / L;
            if (B < 1) B = 1;
            if (B > bcap) B = bcap;
            if (B > n - next) B = n - next;
            T = B * L;
        }
        passes.push_back({next, B, L, T});
        if ((size_t)T > Tmax) Tmax = (size_t)T;
        if ((size_t)B > Bmax) Bmax = (size_t)B;
        next += B;
    }
    // staging planes: ids [T] | mask or positions [T] | token types [T] (pairs only) | sequence starts [B + 1]
    const size_t io_elems = (logits ? 3 : 2) * Tmax + Bmax + 1;
    const size_t io_bytes = std::max(io_elems * sizeof(int32_t), (size_t)1 << 20), out_bytes = std::max(Bmax * H * sizeof(float), (size_t)1 << 20);
    // staging set s and lane s go together: set 0 on the encoder's stream, set 1 on st2 (which starts behind whatever the
    // stream holds now); the host waits for each set's event before it reads its results, so nothing else joins them
    const bool lanes = passes.size() > 1 && encoder_two_lanes();
    hipStream_t lane_st[2] = {st, lanes ? e->st2 : st};
    MI_TRY(ensure_io(e, (io_elems + 2) / 3, Bmax * H * sizeof(float), st, 0));
    MI_TRY(ensure_ws(e, (int64_t)Tmax, st, 0));
    if (lanes) {
        MI_TRY(ensure_io(e, (io_elems + 2) / 3, Bmax * H * sizeof(float), e->st2, 1));
        MI_TRY(ensure_ws(e, (int64_t)Tmax, e->st2, 1));
        MI_HIP(hipEventRecord(e->ev_fork, st));
        MI_HIP(hipStreamWaitEvent(e->st2, e->ev_fork, 0));
    }
    for (int s = 0; s < 2 && s < (int)passes.size(); ++s) { MI_TRY(e->pin_io[s].ensure(io_bytes)); MI_TRY(e->pin_out[s].ensure(out_bytes)); }
    Pass inflight[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    auto unpack = [&](int s) -> int {                    // results of the pass staged in set s -> caller's rows
        if (inflight[s].B == 0) return MI_OK;
        MI_HIP(hipEventSynchronize(e->ev_pin[s]));
        const float* src = static_cast<const float*>(e->pin_out[s].p);
        for (int64_t r = 0; r < inflight[s].B; ++r) {
            const int64_t i = order[(size_t)(inflight[s].first + r)];
            if (lens[(size_t)i] == 0) memset(out_emb + i * H, 0, (size_t)H * sizeof(float));      // an empty sequence embeds to the zero vector
            else memcpy(out_emb + i * H, src + r * H, (size_t)H * sizeof(float));
        }
        inflight[s].B = 0;
        return MI_OK;
    };
    for (size_t pi = 0; pi < passes.size(); ++pi) {
        const int s = (int)(pi & 1);
        const Pass ps = passes[pi];
        const int L = ps.L;
        const int64_t B = ps.B;
        MI_TRY(unpack(s));                               // set s is free once its previous pass has been read back
        const size_t T = (size_t)ps.T;
        int32_t* pid = static_cast<int32_t*>(e->pin_io[s].p);
        int32_t* pmask = pid + T;                        // mask (padded) or positions (packed)
        int32_t* ptype = pmask + T;                      // pairs only
        int32_t* pstart = logits ? ptype + T : ptype;
        const int lane = lanes ? s : 0;
        hipStream_t pst = lane_st[lane];
        float* const d_out = e->ln[lane].d_out;
        int32_t* d_ids = e->ln[lane].d_io;
        int32_t* d_mask = d_ids + T;
        int32_t* d_type = d_mask + T;
        int32_t* d_start = logits ? d_type + T : d_type;
        if (varlen) {
            size_t at = 0;
            for (int64_t r = 0; r < B; ++r) {
                const int64_t i = order[(size_t)(ps.first + r)];
                const int len = lens[(size_t)i];
                pstart[r] = (int32_t)at;
                if (len == 0) { pid[at] = 0; pmask[at] = 0; at += 1; continue; }
                memcpy(pid + at, flat_ids + offsets[i], (size_t)len * sizeof(int32_t));
                if (logits) memcpy(ptype + at, flat_types + offsets[i], (size_t)len * sizeof(int32_t));
                for (int c = 0; c < len; ++c) pmask[at + c] = c;
                at += (size_t)len;
            }
            pstart[B] = (int32_t)at;
            MI_HIP(hipMemcpyAsync(d_ids, pid, ((logits ? 3 : 2) * T + (size_t)B + 1) * sizeof(int32_t), hipMemcpyHostToDevice, pst));
            MI_TRY(encode_pass_locked(e, B, L, d_ids, logits ? d_type : nullptr, nullptr, logits ? nullptr : d_out, logits ? d_out : nullptr,
                                      pst, nullptr, 0, d_mask, d_start, (int64_t)T, lane));
        } else {
            for (int64_t r = 0; r < B; ++r) {
                const int64_t i = order[(size_t)(ps.first + r)];
                const int len = lens[(size_t)i];
                memcpy(pid + r * L, flat_ids + offsets[i], (size_t)len * sizeof(int32_t));
                for (int c = 0; c < len; ++c) pmask[r * L + c] = 1;
                if (len < L) {
                    memset(pid + r * L + len, 0, (size_t)(L - len) * sizeof(int32_t));
                    memset(pmask + r * L + len, 0, (size_t)(L - len) * sizeof(int32_t));
                }
            }
            MI_HIP(hipMemcpyAsync(d_ids, pid, 2 * T * sizeof(int32_t), hipMemcpyHostToDevice, pst));
            MI_TRY(encode_pass_locked(e, B, L, d_ids, nullptr, d_mask, d_out, nullptr, pst, nullptr, 0, nullptr, nullptr, 0, lane));
        }
        MI_HIP(hipMemcpyAsync(e->pin_out[s].p, d_out, (size_t)B * H * sizeof(float), hipMemcpyDeviceToHost, pst));
        MI_HIP(hipEventRecord(e->ev_pin[s], pst));
        inflight[s] = ps;
    }
    // oldest first
    const int s_old = (inflight[0].B && inflight[1].B && inflight[1].first < inflight[0].first) ? 1 : 0;
    MI_TRY(unpack(s_old));
    MI_TRY(unpack(s_old ^ 1));
    return MI_OK;
}

extern "C" int mi_encoder_encode_ragged(mi_encoder* e, int64_t n, const int32_t* flat_ids, const int64_t* offsets,
                                        int max_len, int max_batch, float* out_emb)
{
    MI_REQUIRE(e && n >= 0, "mi_encoder_encode_ragged: bad argument");
    if (n == 0) return MI_OK;
    MI_REQUIRE(flat_ids && offsets && out_emb, "mi_encoder_encode_ragged: NULL buffer");
    MI_REQUIRE(max_len >= 0 && max_batch >= 0, "mi_encoder_encode_ragged: max_len / max_batch must be >= 0 (0 = no limit)");
    return ragged_run(e, "mi_encoder_encode_ragged", n, flat_ids, nullptr, offsets, max_len, max_batch, out_emb);
}

// Cross-encoder logits for n (query, passage) pairs of different lengths, packed like mi_encoder_encode_ragged's passes.
extern "C" int mi_encoder_classify_ragged(mi_encoder* e, int64_t n, const int32_t* flat_ids, const int32_t* flat_type_ids,
                                          const int64_t* offsets, int max_batch, float* out_logits)
{
    MI_REQUIRE(e && n >= 0, "mi_encoder_classify_ragged: bad argument");
    if (n == 0) return MI_OK;
    MI_REQUIRE(flat_ids && flat_type_ids && offsets && out_logits, "mi_encoder_classify_ragged: NULL buffer");
    MI_REQUIRE(max_batch >= 0, "mi_encoder_classify_ragged: max_batch must be >= 0 (0 = no limit)");
    return ragged_run(e, "mi_encoder_classify_ragged", n, flat_ids, flat_type_ids, offsets, 0, max_batch, out_logits);
}

extern "C" int mi_encoder_stats(const mi_encoder* enc, int64_t out[4])
{
    MI_REQUIRE(enc && out, "mi_encoder_stats: NULL argument");
    mi_encoder* e = const_cast<mi_encoder*>(enc);
    std::lock_guard<std::mutex> lk(e->mu);
    out[0] = e->st_seqs; out[1] = e->st_tokens; out[2] = e->st_bulk; out[3] = e->st_latency;
    return MI_OK;
}

extern "C" int mi_encoder_last_time(const mi_encoder* enc, float* out_ms)
{
    MI_REQUIRE(enc && out_ms, "mi_encoder_last_time: NULL argument");
    mi_encoder* e = const_cast<mi_encoder*>(enc);
    std::lock_guard<std::mutex> lk(e->mu);
    DeviceGuard guard(e->device);
    *out_ms = 0.0f;
    if (!e->timed) return MI_OK;
    MI_HIP(hipEventSynchronize(e->ln[e->last_lane].ev1));
    MI_HIP(hipEventElapsedTime(out_ms, e->ln[e->last_lane].ev0, e->ln[e->last_lane].ev1));
    return MI_OK;
}

#ifdef FF_TRACE
extern "C" int mi_debug_read_ff_trace(void* out)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(ff_trace_buf), sizeof(ff_trace_buf));
}
#endif
