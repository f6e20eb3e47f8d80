// encoder_qkv_attn.hip — the head of a BERT layer in one kernel: Q, K, V = h Wqkv^T + b, then softmax(Q K^T + mask) V, per
// (head, sequence), for bulk embedding (sequences of at most 256 tokens, heads of 32).
// Replaces gemm_ws_kernel<EPI_BIAS> + attn_kernel of encoder.hip on that path (modeling_bert.py BertSelfAttention.forward:
// query / key / value Linear, scores / sqrt(dh), + mask, softmax, context): the [T][3 H] QKV tensor -- 151 MB written and
// read back per layer of a 65,536-token pass -- never exists, and attn_kernel's own start (K, V, Q of the sequence from HBM)
// goes with it.  Measured: 286 against 299 us per layer pass on one lane, 153-157k against 142-144k chunks/s on two
// (profiles/r03_qkv_attn.md, with the three variants that came before this one and what bounded each).
//
// One workgroup (4 waves, 3 per CU) per (head, sequence); wave w owns tokens 64 w .. 64 w + 63 as two blocks of 32.
//   1 projection  C[out][token] = Wslice[out][k] h[token][k], the token on the MFMA lane (B operand), the head's 96 weight
//                 rows (Q, K, V) the A operand: a weight fragment feeds two MFMAs, 96 accumulator registers.  Both operands
//                 come through LDS in half chunks of 32 k, double-buffered, brought by LDS-DMA (no staging registers):
//                 the weights packed once at load in operand order (mi_qkv_attn_pack), the rows of h by whole 64-byte
//                 pieces with the bank swizzle applied on the source side.  (A lane reading its own token's row straight
//                 from global memory touches 32 cache lines per wave instruction and the CU's address path sets the pace;
//                 32 tokens per wave read one weight fragment per MFMA and LDS bandwidth does: both measured.)
//   2 hand-over   an accumulator column is one token's 16 values of the half hp: exactly an MFMA operand row.  Q stays in
//                 registers as the B operand of S^T = K Q^T; K goes to LDS in the same element order (the k-permutation is
//                 the same on both sides of the dot product), V row-major for the transposing LDS read.
//   3 attention   the loop of attn_kernel (encoder.hip): running softmax in log2 units, P as the B operand of O^T = V^T P^T.
// The heads of a sequence sit on one XCD (one HBM fetch of its rows of h, then L2 hits).
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
#define QA_ABL 0               // timing ablations (wrong results): 1 no h staging, 2 no weight staging, 4 no attention, 8 no projection
#endif
#define QA_WHALF 3072           // bf16 elements of one packed half chunk (32 k): 3 matrices x 2 k-steps x 64 lanes x 8
#define QA_HB 16384             // bytes of one half chunk of the sequence's rows of h in LDS: 256 rows x 64 B
#define QA_WBY 6144             // bytes of one packed half chunk

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

static __device__ __forceinline__ uint32_t qa_pack2(float lo, float hi)
{
    bf16x2 v;
    v[0] = (__bf16)lo;
    v[1] = (__bf16)hi;
    return __builtin_bit_cast(uint32_t, v);
}

// unit u (16 bytes) of the packed stream: half chunk (head, c) | matrix w | k-step s | lane
//   <-  W[w H + 32 head + (lane & 31)][32 c + 16 (lane >> 5) + 8 s ..]
__global__ __launch_bounds__(256) void qkv_pack_kernel(const uint16_t* __restrict__ W, uint16_t* __restrict__ out, int H, int NH)
{
    const int64_t total = (int64_t)3 * H * H / 8;
    const int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (u >= total) return;
    const int lane = (int)(u & 63), s = (int)((u >> 6) & 1), w = (int)((u >> 7) % 3);
    const int64_t ch = u / 384;
    const int nhc = H >> 5;
    const int c = (int)(ch % nhc), head = (int)(ch / nhc);
    const int64_t row = (int64_t)w * H + head * 32 + (lane & 31);
    const int k = 32 * c + 16 * (lane >> 5) + 8 * s;
    *reinterpret_cast<uint4*>(out + u * 8) = *reinterpret_cast<const uint4*>(W + row * H + k);
}

__global__ __launch_bounds__(256, 3) void qkv_attn_kernel(const uint16_t* __restrict__ h, const uint16_t* __restrict__ wpk,
        const float* __restrict__ bqkv, const int32_t* __restrict__ mask, const int32_t* __restrict__ seq_start,
        uint16_t* __restrict__ ctx, int L_in, int H, int NH, int B, int ctx_blocked)
{
    // projection: Hb[2] [256][64 B] two half chunks (32 k) of the sequence's rows of h, swizzled | Wb[2] [6 KB] their weights;
    // attention: Ks [256][64 B] swizzled | Vs [256][64 B] over Hb; maskf [256] behind
    __shared__ __attribute__((aligned(16))) char smem[2 * QA_HB + 2 * QA_WBY + 256 * 4];
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
    char* const Hb = smem;
    char* const Wb = smem + 2 * QA_HB;
    float* const maskf = reinterpret_cast<float*>(smem + 2 * QA_HB + 2 * QA_WBY);

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int hp = lane >> 5, col = lane & 31;
    const int nhc = H >> 5;                                 // half chunks
    const int nblk_c = H >> 5;

    for (int key = t; key < Lp; key += 256)
        maskf[key] = (key < L && (seq_start || mask[tok0 + key] != 0)) ? 0.0f : -FLT_MAX;

    const int q0 = wave * 64;
    const bool active = q0 < L;                             // a wave without tokens stages weights and meets the barriers only

    // Staging by LDS-DMA (no registers: 96 accumulators leave none).  A DMA instruction writes 64 x 16 bytes to consecutive
    // LDS addresses, so the swizzle sits on the SOURCE side: lane i of instruction q (rows 64 wave + 16 q ..) fetches, for LDS
    // slot (row = .. + i / 4, slot i % 4), the piece slot ^ ((row >> 2) & 3) of that row -- the four lanes of a row still
    // cover its 64 bytes.  A wave stages its OWN 64 rows (it is their only reader); rows past the sequence re-read row L - 1.
    // Reader: lane (col, hp), k-step s wants piece 2 hp + s of row q0 + 32 nb + col: 16 lanes reading one piece of 16
    // consecutive rows hit all 64 banks.
    const uint16_t* hsrc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int row = q0 + 16 * q + (lane >> 2);
        int r = row;
        if (r > L - 1) r = L - 1;
        hsrc[q] = h + (tok0 + r) * H + (((lane & 3) ^ ((row >> 2) & 3)) << 3);
    }
    // weights: a half chunk is 6 instructions of 1 KB; wave w issues instructions w and 4 + (w & 1) (waves 2, 3 repeat what
    // waves 0, 1 bring: same bytes, same place)
    const uint16_t* const wbase = wpk + (size_t)head * nhc * QA_WHALF + lane * 8;
    const int wi0 = wave, wi1 = 4 + (wave & 1);
    auto stage = [&](int c, int buf) __attribute__((always_inline)) {
        if (!(QA_ABL & 2)) {
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)(wbase + (size_t)c * QA_WHALF + wi0 * 512), (lds_ptr_t)(Wb + buf * QA_WBY + wi0 * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)(wbase + (size_t)c * QA_WHALF + wi1 * 512), (lds_ptr_t)(Wb + buf * QA_WBY + wi1 * 1024), 16, 0, 0);
        }
        if (active && !(QA_ABL & 1)) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                __builtin_amdgcn_global_load_lds((gbl_ptr_t)(hsrc[q] + 32 * c), (lds_ptr_t)(Hb + buf * QA_HB + (q0 + 16 * q) * 64), 16, 0, 0);
        }
    };
    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // ---- 1 projection
    f32x16 acc[2][3];
#pragma unroll
    for (int w = 0; w < 3; ++w)
#pragma unroll
        for (int g = 0; g < 4; ++g) {                       // register 4 g + e <-> output row 8 g + 4 hp + e
            const float4 b4 = *reinterpret_cast<const float4*>(bqkv + (size_t)w * H + head * 32 + 8 * g + 4 * hp);
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                acc[nb][w][4 * g + 0] = b4.x; acc[nb][w][4 * g + 1] = b4.y; acc[nb][w][4 * g + 2] = b4.z; acc[nb][w][4 * g + 3] = b4.w;
            }
        }
    const int brow0 = q0 + col, brow1 = q0 + 32 + col;
    const char* const bsrc0 = Hb + brow0 * 64;
    const char* const bsrc1 = Hb + brow1 * 64;
    const int bsw0 = (brow0 >> 2) & 3, bsw1 = (brow1 >> 2) & 3;
    const int nhc_run = (QA_ABL & 8) ? 0 : nhc;
#pragma unroll 1
    for (int c = 0; c < nhc_run; ++c) {
        const int buf = c & 1;
        stage(c + 1 < nhc ? c + 1 : c, buf ^ 1);            // (the last half chunk requests itself again: nobody reads that copy)
        if (active) {
            const char* const wb = Wb + buf * QA_WBY + lane * 16;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const bf16x8 b0 = *reinterpret_cast<const bf16x8*>(bsrc0 + buf * QA_HB + (((2 * hp + s) ^ bsw0) << 4));
                const bf16x8 b1 = *reinterpret_cast<const bf16x8*>(bsrc1 + buf * QA_HB + (((2 * hp + s) ^ bsw1) << 4));
#pragma unroll
                for (int w = 0; w < 3; ++w) {
                    const bf16x8 a = *reinterpret_cast<const bf16x8*>(wb + (w * 2 + s) * 1024);
                    acc[0][w] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b0, acc[0][w], 0, 0, 0);
                    acc[1][w] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b1, acc[1][w], 0, 0, 0);
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // this wave's DMA of the next half chunk has landed ...
        __syncthreads();                                    // ... and everybody's; everybody is done with this one
    }
    uint32_t R[2][3][8];                                    // bf16 pairs: [token block][Q K V][dh pair]
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int w = 0; w < 3; ++w)
#pragma unroll
            for (int r = 0; r < 8; ++r) R[nb][w][r] = qa_pack2(acc[nb][w][2 * r], acc[nb][w][2 * r + 1]);

    // ---- 2 hand-over (the last barrier above: nobody reads a staging buffer any more, no DMA is in flight)
    if (active) {
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            const int tk = q0 + 32 * nb + col;
            if (tk < Lp) {
                const bool real = tk < L;                   // rows L .. Lp - 1 are zero, as attn_kernel stages them
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const uint4 kk = real ? make_uint4(R[nb][1][4 * s], R[nb][1][4 * s + 1], R[nb][1][4 * s + 2], R[nb][1][4 * s + 3]) : make_uint4(0, 0, 0, 0);
                    *reinterpret_cast<uint4*>(Ks + tk * 64 + (((2 * s + hp) ^ ((tk >> 2) & 3)) << 4)) = kk;
                }
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const uint2 vv = real ? make_uint2(R[nb][2][2 * g], R[nb][2][2 * g + 1]) : make_uint2(0, 0);
                    *reinterpret_cast<uint2*>(Vs + tk * 64 + (8 * g + 4 * hp) * 2) = vv;
                }
            }
        }
    }
    __syncthreads();
    if (!active || (QA_ABL & 4)) return;                    // whole wave; no barrier follows

    // ---- 3 attention (see attn_kernel in encoder.hip for the reasoning behind each step)
    bf16x8 qf[2][2];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int s = 0; s < 2; ++s)
            qf[nb][s] = __builtin_bit_cast(bf16x8, make_uint4(R[nb][0][4 * s], R[nb][0][4 * s + 1], R[nb][0][4 * s + 2], R[nb][0][4 * s + 3]));

    float m_run[2] = {-__builtin_inff(), -__builtin_inff()};
    f32x16 O[2];
    f32x2 ps2[2] = {{0.0f, 0.0f}, {0.0f, 0.0f}};
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
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            f32x16 S = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[0], qf[nb][0], MK, 0, 0, 0);
            S = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[1], qf[nb][1], S, 0, 0, 0);
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
            if (__builtin_amdgcn_ballot_w64(!(bm <= 8.0f)) != 0) {
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
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                f32x2 e2[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    e2[i].x = __builtin_amdgcn_exp2f(d2[4 * s + i].x);
                    e2[i].y = __builtin_amdgcn_exp2f(d2[4 * s + i].y);
                    ps2[nb] = ps2[nb] + e2[i];
                }
                const uint4 pk = make_uint4(qa_pack2(e2[0].x, e2[0].y), qa_pack2(e2[1].x, e2[1].y), qa_pack2(e2[2].x, e2[2].y), qa_pack2(e2[3].x, e2[3].y));
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
            for (int g = 0; g < 4; ++g) {
                uint2 o;
                o.x = qa_pack2(O[nb][4 * g + 0] * inv, O[nb][4 * g + 1] * inv);
                o.y = qa_pack2(O[nb][4 * g + 2] * inv, O[nb][4 * g + 3] * inv);
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
    qkv_attn_kernel<<<(unsigned)((B + 7) / 8 * 8 * NH), 256, 0, st>>>(h, wpk, bqkv, mask, seq_start, ctx, L, H, NH, (int)B, ctx_blocked);
    MI_HIP(hipGetLastError());
    return MI_OK;
}
