// encoder_qkv_attn.hip — the head of a BERT layer in one kernel: Q, K, V = h Wqkv^T + b, then softmax(Q K^T + mask) V, per
// (head, sequence), for bulk embedding (sequences of at most 256 tokens, heads of 32).
// Replaces gemm_ws_kernel<EPI_BIAS> + attn_kernel of encoder.hip on that path (modeling_bert.py BertSelfAttention.forward:
// query / key / value Linear, scores / sqrt(dh), + mask, softmax, context): the [T][3 H] QKV tensor -- 151 MB written and
// read back per layer of a 65,536-token pass -- never exists.  The attention loop is VALU-bound (softmax) and left the matrix
// pipe two thirds idle; the projection's MFMAs of one workgroup now run under the softmax of the others on the same SIMD.
//
// One workgroup (4 waves) per (head, sequence); wave w owns tokens 64 w .. 64 w + 63 as two blocks of 32.
//   1 projection  C[out][token] = Wslice[out][k] h[token][k]: the token sits on the MFMA lane (B operand, straight from
//                 global memory: a lane reads 64 contiguous bytes of its row per 64-k chunk), the weight slice of the head
//                 (96 rows: Q, K, V) is the A operand, staged chunk by chunk through LDS in operand order (packed once at
//                 load, mi_qkv_attn_pack), double-buffered in the space K and V take afterwards.  One token block at a
//                 time (48 accumulator registers), the slice is streamed twice.
//   2 hand-over   an accumulator column is one token's 16 values of the half hp: exactly an MFMA operand row.  Q stays in
//                 registers as the B operand of S^T = K Q^T; K goes to LDS in the same element order (the k-permutation is
//                 the same on both sides of the dot product), V row-major for the transposing LDS read.
//   3 attention   the loop of attn_kernel (encoder.hip): running softmax in log2 units, P as the B operand of O^T = V^T P^T.
#include <hip/hip_runtime.h>
#include <cfloat>
#include <cstdint>
#include <type_traits>
#include "encoder_qkv_attn.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef short tr4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) tr4_t tr4_lds_t;

#ifndef QA_ABL
#define QA_ABL 0               // timing ablations (wrong results): 1 no h loads, 2 no weight staging, 4 no attention, 8 no projection
#endif
#define QA_WB 12288             // bytes of one packed chunk
#define QA_WCHUNK 6144          // bf16 elements of one packed chunk: 3 matrices x 4 k-steps x 64 lanes x 8

static __device__ __forceinline__ uint32_t qa_pack2(float lo, float hi)
{
    bf16x2 v;
    v[0] = (__bf16)lo;
    v[1] = (__bf16)hi;
    return __builtin_bit_cast(uint32_t, v);
}

// unit u (16 bytes) of the packed stream: chunk (head, c) | matrix w | k-step s | lane  <-  W[w H + 32 head + (lane & 31)][64 c + 32 (lane >> 5) + 8 s ..]
__global__ __launch_bounds__(256) void qkv_pack_kernel(const uint16_t* __restrict__ W, uint16_t* __restrict__ out, int H, int NH)
{
    const int64_t total = (int64_t)3 * H * H / 8;
    const int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (u >= total) return;
    const int lane = (int)(u & 63), s = (int)((u >> 6) & 3), w = (int)((u >> 8) % 3);
    const int64_t ch = u / 768;
    const int nkc = H >> 6;
    const int c = (int)(ch % nkc), head = (int)(ch / nkc);
    const int64_t row = (int64_t)w * H + head * 32 + (lane & 31);
    const int k = 64 * c + 32 * (lane >> 5) + 8 * s;
    *reinterpret_cast<uint4*>(out + u * 8) = *reinterpret_cast<const uint4*>(W + row * H + k);
}

__global__ __launch_bounds__(512, 4) void qkv_attn_kernel(const uint16_t* __restrict__ h, const uint16_t* __restrict__ wpk,
        const float* __restrict__ bqkv, const int32_t* __restrict__ mask, const int32_t* __restrict__ seq_start,
        uint16_t* __restrict__ ctx, int L_in, int H, int NH, int B, int ctx_blocked)
{
    // projection: Wb [12 KB] one weight chunk | Hb [256][128 B] one 64-k chunk of the sequence's rows of h, swizzled;
    // attention: Ks [256][64 B] swizzled | Vs [256][64 B] over the same bytes; maskf [256] behind
    __shared__ __attribute__((aligned(16))) char smem[QA_WB + 256 * 128 + 256 * 4];
    // Workgroups go to the 8 XCDs round robin by their linear id, and each XCD has its own L2: the NH heads of a sequence all
    // read the sequence's rows of h, so they sit on ONE XCD, next to each other in time (one HBM fetch, NH - 1 L2 hits).
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int seq = (slot / NH) * 8 + xcd, head = slot % NH;
    if (seq >= B) return;                                   // (the grid is rounded up to 8 sequences)
    int L = L_in;
    int64_t tok0 = (int64_t)seq * L_in;
    if (seq_start) {
        tok0 = seq_start[seq];
        L = seq_start[seq + 1] - seq_start[seq];
    }
    if (L <= 0 || L > MI_QKV_ATTN_MAX_L) return;            // whole workgroup, before any barrier (the launcher checks L)
    const int Lp = (L + 31) / 32 * 32;
    char* const Ks = smem;
    char* const Vs = smem + 256 * 64;
    char* const Wb = smem;
    char* const Hb = smem + QA_WB;
    float* const maskf = reinterpret_cast<float*>(smem + QA_WB + 256 * 128);

    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int hp = lane >> 5, col = lane & 31;
    const int nkc = H >> 6;
    const int nblk_c = H >> 5;

    for (int key = t; key < Lp; key += 512)
        maskf[key] = (key < L && (seq_start || mask[tok0 + key] != 0)) ? 0.0f : -FLT_MAX;

    const int q0 = wave * 32;
    const bool active = q0 < L;                             // a wave without tokens stages weights and meets the barriers only
    // staging of h: thread t moves piece t & 7 (16 bytes) of rows (t >> 3) + 64 j -- a wave instruction covers 8 whole
    // 128-byte lines (a lane reading its own token's row straight into the B operand touches 32 lines per instruction, and
    // the address path of the CU, not the matrix pipe, then sets the pace: measured).  Rows past the sequence re-read row
    // L - 1.  In LDS the 16-byte pieces of row r are XOR-ed with (r >> 1) & 7: 16 lanes reading one piece column of 16
    // consecutive rows hit all 64 banks.
    const int hpiece = t & 7, hr0 = t >> 3;
    const uint16_t* hsrc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int r = hr0 + 64 * j;
        if (r > L - 1) r = L - 1;
        hsrc[j] = h + (tok0 + r) * H + hpiece * 8;
    }
    const int hdst = hr0 * 128 + ((hpiece ^ ((hr0 >> 1) & 7)) << 4);     // + 8192 j
    const int brow = q0 + col;                                            // this lane's token (B operand column)
    const char* const bsrc = Hb + brow * 128;
    const int bsw = (brow >> 1) & 7;
    // a 12-KB weight chunk is 768 units of 16 bytes: every thread moves unit t and unit 512 + (t & 255) (the upper half of
    // the workgroup repeats the lower half's second unit: same bytes to the same place, and no branch in the loop -- hipcc
    // waits for ALL outstanding loads wherever loads sit under a branch, which costs the whole prefetch)
    const uint16_t* const wsrc0 = wpk + (size_t)head * nkc * QA_WCHUNK + t * 8;
    const uint16_t* const wsrc1 = wpk + (size_t)head * nkc * QA_WCHUNK + (512 + (t & 255)) * 8;
    const int wdst0 = t * 16, wdst1 = (512 + (t & 255)) * 16;

    // ---- 1 projection
    u32x4 wr0 = *reinterpret_cast<const u32x4*>(wsrc0);
    u32x4 wr1 = *reinterpret_cast<const u32x4*>(wsrc1);
    u32x4 hr[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) hr[j] = *reinterpret_cast<const u32x4*>(hsrc[j]);
    auto put = [&]() __attribute__((always_inline)) {
        *reinterpret_cast<u32x4*>(Wb + wdst0) = wr0;
        *reinterpret_cast<u32x4*>(Wb + wdst1) = wr1;
#pragma unroll
        for (int j = 0; j < 4; ++j) *reinterpret_cast<u32x4*>(Hb + hdst + 8192 * j) = hr[j];
    };
    auto request = [&](int c) __attribute__((always_inline)) {
        if (!(QA_ABL & 2)) {
            wr0 = *reinterpret_cast<const u32x4*>(wsrc0 + (size_t)c * QA_WCHUNK);
            wr1 = *reinterpret_cast<const u32x4*>(wsrc1 + (size_t)c * QA_WCHUNK);
        }
        if (!(QA_ABL & 1)) {
#pragma unroll
            for (int j = 0; j < 4; ++j) hr[j] = *reinterpret_cast<const u32x4*>(hsrc[j] + 64 * c);
        }
    };
    put();
    __builtin_amdgcn_sched_barrier(0);
    request(1);
    __syncthreads();

    f32x16 acc[3];
#pragma unroll
    for (int w = 0; w < 3; ++w)
#pragma unroll
        for (int g = 0; g < 4; ++g) {                       // register 4 g + e <-> output row 8 g + 4 hp + e
            const float4 b4 = *reinterpret_cast<const float4*>(bqkv + (size_t)w * H + head * 32 + 8 * g + 4 * hp);
            acc[w][4 * g + 0] = b4.x; acc[w][4 * g + 1] = b4.y; acc[w][4 * g + 2] = b4.z; acc[w][4 * g + 3] = b4.w;
        }
    // chunk c: the MFMAs on what LDS holds; when every wave is done with it, chunk c + 1 (requested a chunk ago) goes from
    // registers to LDS and chunk c + 2 is requested.  Straight-line code: the last chunks request the last one again.
    auto chunk = [&](int c, auto ACT) __attribute__((always_inline)) {
        if (decltype(ACT)::value) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const bf16x8 bfr = *reinterpret_cast<const bf16x8*>(bsrc + (((4 * hp + s) ^ bsw) << 4));
#pragma unroll
                for (int w = 0; w < 3; ++w) {
                    const bf16x8 a = *reinterpret_cast<const bf16x8*>(Wb + ((w * 4 + s) * 64 + lane) * 16);
                    acc[w] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bfr, acc[w], 0, 0, 0);
                }
            }
        }
        __syncthreads();
        put();
        __builtin_amdgcn_sched_barrier(0);
        request(c + 2 < nkc ? c + 2 : nkc - 1);
        __syncthreads();
    };
    const int nkc_run = (QA_ABL & 8) ? 0 : nkc;
    if (active) {
#pragma unroll 1
        for (int c = 0; c < nkc_run; ++c) chunk(c, std::true_type{});
    } else {                                                // a wave without tokens: staging and barriers only
#pragma unroll 1
        for (int c = 0; c < nkc_run; ++c) chunk(c, std::false_type{});
    }
    uint32_t R[3][8];                                       // bf16 pairs: [Q K V][dh pair]
#pragma unroll
    for (int w = 0; w < 3; ++w)
#pragma unroll
        for (int r = 0; r < 8; ++r) R[w][r] = qa_pack2(acc[w][2 * r], acc[w][2 * r + 1]);

    // ---- 2 hand-over (the last barrier above: nobody reads a weight buffer any more)
    if (active) {
        const int tk = q0 + col;
        if (tk < Lp) {
            const bool real = tk < L;                       // rows L .. Lp - 1 are zero, as attn_kernel stages them
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const uint4 kk = real ? make_uint4(R[1][4 * s], R[1][4 * s + 1], R[1][4 * s + 2], R[1][4 * s + 3]) : make_uint4(0, 0, 0, 0);
                *reinterpret_cast<uint4*>(Ks + tk * 64 + (((2 * s + hp) ^ ((tk >> 2) & 3)) << 4)) = kk;
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const uint2 vv = real ? make_uint2(R[2][2 * g], R[2][2 * g + 1]) : make_uint2(0, 0);
                *reinterpret_cast<uint2*>(Vs + tk * 64 + (8 * g + 4 * hp) * 2) = vv;
            }
        }
    }
    __syncthreads();
    if (!active || (QA_ABL & 4)) return;                    // whole wave; no barrier follows

    // ---- 3 attention (see attn_kernel in encoder.hip for the reasoning behind each step)
    bf16x8 qf[2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
        qf[s] = __builtin_bit_cast(bf16x8, make_uint4(R[0][4 * s], R[0][4 * s + 1], R[0][4 * s + 2], R[0][4 * s + 3]));

    float m_run = -__builtin_inff();
    f32x16 O;
    f32x2 ps2 = {0.0f, 0.0f};
#pragma unroll
    for (int i = 0; i < 16; ++i) O[i] = 0.0f;
    auto max3 = [](float a, float b2, float c) __attribute__((always_inline)) {
        float r;
        asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b2), "v"(c));
        return r;
    };

    const int ksw = (col >> 2) & 3;
    for (int kb = 0; kb < Lp / 32; ++kb) {
        bf16x8 kf[2];
#pragma unroll
        for (int s = 0; s < 2; ++s)
            kf[s] = *reinterpret_cast<const bf16x8*>(Ks + (32 * kb + col) * 64 + (((2 * s + hp) ^ ksw) << 4));
        bf16x8 vf[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const char* vp = Vs + (32 * kb + 16 * s + 4 * hp + ((lane & 15) >> 2)) * 64 + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
            const tr4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tr4_lds_t*)(vp));
            const tr4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tr4_lds_t*)(vp + 8 * 64));
            vf[s] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
        }
        f32x16 MK;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 mk = *reinterpret_cast<const float4*>(maskf + 32 * kb + 8 * g + 4 * hp);
            MK[4 * g + 0] = mk.x; MK[4 * g + 1] = mk.y; MK[4 * g + 2] = mk.z; MK[4 * g + 3] = mk.w;
        }
        f32x16 S = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[0], qf[0], MK, 0, 0, 0);
        S = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[1], qf[1], S, 0, 0, 0);
        f32x2 d2[8];
        {
            const f32x2 mn2 = {m_run, m_run};
#pragma unroll
            for (int i = 0; i < 8; ++i) d2[i] = f32x2{S[2 * i], S[2 * i + 1]} - mn2;
        }
        float bm = max3(max3(d2[0].x, d2[0].y, d2[1].x), max3(d2[1].y, d2[2].x, d2[2].y), max3(d2[3].x, d2[3].y, d2[4].x));
        bm = max3(bm, max3(d2[4].y, d2[5].x, d2[5].y), max3(d2[6].x, d2[6].y, d2[7].x));
        bm = fmaxf(bm, d2[7].y);
        bm = fmaxf(bm, __shfl_xor(bm, 32, 64));
        if (__builtin_amdgcn_ballot_w64(!(bm <= 8.0f)) != 0) {
            float tm = fmaxf(fmaxf(S[0], S[1]), S[2]);
#pragma unroll
            for (int i = 3; i < 16; ++i) tm = fmaxf(tm, S[i]);
            tm = fmaxf(tm, __shfl_xor(tm, 32, 64));
            const float m_new = fmaxf(m_run, tm);
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            O = O * alpha;
            ps2 = ps2 * alpha;
            m_run = m_new;
            const f32x2 mn2 = {m_new, m_new};
#pragma unroll
            for (int i = 0; i < 8; ++i) d2[i] = f32x2{S[2 * i], S[2 * i + 1]} - mn2;
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            f32x2 e2[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                e2[i].x = __builtin_amdgcn_exp2f(d2[4 * s + i].x);
                e2[i].y = __builtin_amdgcn_exp2f(d2[4 * s + i].y);
                ps2 = ps2 + e2[i];
            }
            const uint4 pk = make_uint4(qa_pack2(e2[0].x, e2[0].y), qa_pack2(e2[1].x, e2[1].y), qa_pack2(e2[2].x, e2[2].y), qa_pack2(e2[3].x, e2[3].y));
            O = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[s], __builtin_bit_cast(bf16x8, pk), O, 0, 0, 0);
        }
    }
    {
        const int q = q0 + col;
        const float lh = ps2.x + ps2.y;
        const float inv = 1.0f / (lh + __shfl_xor(lh, 32, 64));
        if (q < L) {
            const int64_t tg = tok0 + q;
            uint16_t* dst = ctx_blocked ? ctx + (((tg >> 7) * nblk_c + head) * 128 + (tg & 127)) * 32
                                        : ctx + tg * H + head * 32;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                uint2 o;
                o.x = qa_pack2(O[4 * g + 0] * inv, O[4 * g + 1] * inv);
                o.y = qa_pack2(O[4 * g + 2] * inv, O[4 * g + 3] * inv);
                *reinterpret_cast<uint2*>(dst + 8 * g + 4 * hp) = o;
            }
        }
    }
}

int mi_qkv_attn_pack(hipStream_t st, const uint16_t* Wqkv, uint16_t* dst, int H, int NH)
{
    MI_REQUIRE(Wqkv && dst && H > 0 && H % 128 == 0 && NH > 0 && H == NH * 32, "mi_qkv_attn_pack: hidden size %d / %d heads not supported", H, NH);
    const int64_t total = (int64_t)3 * H * H / 8;
    qkv_pack_kernel<<<(unsigned)((total + 255) / 256), 256, 0, st>>>(Wqkv, dst, H, NH);
    MI_HIP(hipGetLastError());
    return MI_OK;
}

int mi_launch_qkv_attn(hipStream_t st, const uint16_t* h, const uint16_t* wpk, const float* bqkv, const int32_t* mask,
                       const int32_t* seq_start, uint16_t* ctx, int64_t B, int L, int H, int NH, int ctx_blocked)
{
    MI_REQUIRE(h && wpk && bqkv && ctx && (mask || seq_start), "mi_launch_qkv_attn: NULL argument");
    MI_REQUIRE(B > 0 && B <= 0x7fffffff && L > 0 && L <= MI_QKV_ATTN_MAX_L, "mi_launch_qkv_attn: B %lld / L %d out of range", (long long)B, L);
    MI_REQUIRE(H % 128 == 0 && NH > 0 && H == NH * 32 && NH <= 4096 && B <= 65535, "mi_launch_qkv_attn: hidden size %d / %d heads not supported", H, NH);
    qkv_attn_kernel<<<(unsigned)((B + 7) / 8 * 8 * NH), 512, 0, st>>>(h, wpk, bqkv, mask, seq_start, ctx, L, H, NH, (int)B, ctx_blocked);
    MI_HIP(hipGetLastError());
    return MI_OK;
}
