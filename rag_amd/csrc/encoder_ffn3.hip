// encoder_ffn3.hip — the fused tail of a BERT layer with the TOKEN ON THE MFMA LANE (round 3).
//
//   h1  = LayerNorm(ctx Wo^T + bo + h)                       (modeling_bert.py:289-293, BertSelfOutput)
//   out = LayerNorm(gelu(h1 W1^T + b1) W2^T + b2 + h1)       (:334-337 BertIntermediate, :347-351 BertOutput)
//
// for 128 tokens per workgroup, hidden 384, intermediate 1536 — the same arithmetic as ffn2_kernel<true> in
// encoder.hip (the call site: HuggingFaceEmbeddings.embed_documents inside FAISS.from_documents, advanced_rag.py:104-107,
// :295), organised the other way round:
//
//   * every product is computed TRANSPOSED, Y^T = W X^T: the weight tile is the MFMA's A operand, the activations its B
//     operand, so a lane owns ONE token (column = lane & 15) and the accumulator registers run over output channels.
//     Four waves x 32 tokens, ONE wave per SIMD with the whole 512-entry register file: the [384 x 32] fp32 output tile
//     of a wave is 192 registers, the bf16 activations it multiplies (ctx^T, then h1^T) 96 more.
//   * a 16 x 16 accumulator tile is, converted to bf16 as it stands, half of the next product's B fragment (k = the
//     accumulator's row index; cdna_hip_programming.md section 3, "An accumulator tile as the next MFMA's operand").  The
//     k permutation that implies is folded into the weight packing, once, at load time.  So neither the LayerNorm'ed
//     hidden state nor the GELU'ed intermediate ever touches LDS: no partial-sum exchange, no A buffer, no barrier
//     between the two products.
//   * LayerNorm statistics are sums over a lane's own registers plus two cross-lane adds; bias and residual are the
//     accumulators' initial values.
//   * all LDS is ONE ring of five 24-KB slots that the weights stream through in consumption order — per 128-token tile
//     108 blocks of 24 fragments (Wo: 12 k-steps; then W1_j / W2_j interleaved) stored contiguously by the packing
//     kernel — with one barrier per block and a counted vmcnt; the workgroup is persistent and the stream simply
//     continues into the next tile.
//
// Every LDS fragment read is a lane-linear ds_read_b128 (conflict-free by construction); each feeds two
// v_mfma_f32_16x16x32_bf16 (the wave's two token groups).
#include <stdlib.h>
#include <type_traits>
#include <utility>
#include "kernels.h"
#include "encoder_ffn3.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int V> using IC = std::integral_constant<int, V>;
template <int... I, class F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, I...>, F&& f) { (f(IC<I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(std::make_integer_sequence<int, N>{}, static_cast<F&&>(f)); }

#define F3_H 384
#define F3_I 1536
#define F3_BLK 24576                          // one weight block: 24 fragments x 1 KB
#define F3_SLOTS 5                            // the ring stays below 128 KB: no LDS-DMA destination with bit 17 set (M0)
#define F3_RING (F3_SLOTS * F3_BLK)
#define F3_NFRONT 12
#define F3_NBLK (F3_NFRONT + 2 * (F3_I / 32)) // 108 blocks per tile
#define F3_PRM_FLOATS (6 * F3_H + F3_I)       // bo g1 be1 b2 g2 be2 | b1
#define F3_LDS_BYTES (F3_RING + F3_PRM_FLOATS * 4)
#define F3_AHEAD 3                            // block b + 3 is requested while block b is consumed (into the slot of block b - 2)

static_assert(F3_LDS_BYTES <= 160 * 1024, "LDS");
static_assert(F3_RING <= 128 * 1024, "LDS-DMA destinations stay below 128 KB");
static_assert(F3_AHEAD + 2 == F3_SLOTS, "block b + AHEAD reuses the slot of block b - 2");

size_t mi_ffn3_stream_bytes() { return (size_t)F3_NBLK * F3_BLK; }
size_t mi_ffn3_param_floats() { return F3_PRM_FLOATS; }

// ---------------------------------------------------------------- packing (once per layer, at mi_encoder_create)
// Block q of the main stream (q = 0 .. 95), in the order the loop consumes them: W1_0, W1_1, then W2_j, W1_{j+2} for
// j = 0 .. 45, then W2_46, W2_47.  Returns j; *is_w2 says which matrix.
__host__ __device__ static inline int f3_main_block(int q, bool* is_w2)
{
    if (q < 2) { *is_w2 = false; return q; }
    if (q >= 94) { *is_w2 = true; return q - 48; }
    const int m = q - 2;
    *is_w2 = (m & 1) == 0;
    return *is_w2 ? m / 2 : m / 2 + 2;
}

// One thread per (block, fragment, lane): its 8 bf16 elements.  Lane (r = lane & 15, g = lane >> 4), element e:
//   Wo block s (k-step s), fragment T:             Wo[16 T + r][32 s + 8 g + e]                    natural k (B = ctx^T from memory)
//   W1_j, fragment 12 u + s (rows 16 u .., k-step s): W1[32 j + 16 u + r][32 s + 16 (e >> 2) + 4 g + (e & 3)]   k = channel, permuted as
//   W2_j, fragment T:                              W2[16 T + r][32 j + 16 (e >> 2) + 4 g + (e & 3)]         an accumulator pair holds it
__global__ void ffn3_pack_kernel(const float* __restrict__ Wo, const float* __restrict__ W1, const float* __restrict__ W2,
                                 uint16_t* __restrict__ dst)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= F3_NBLK * 24 * 64) return;
    const int lane = idx & 63, f = (idx >> 6) % 24, b = idx / (24 * 64);
    const int r = lane & 15, g = lane >> 4;
    uint16_t v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        float x;
        const int kp = 16 * (e >> 2) + 4 * g + (e & 3);
        if (b < F3_NFRONT) {
            x = Wo[(size_t)(16 * f + r) * F3_H + 32 * b + 8 * g + e];
        } else {
            bool w2;
            const int j = f3_main_block(b - F3_NFRONT, &w2);
            if (w2) x = W2[(size_t)(16 * f + r) * F3_I + 32 * j + kp];
            else x = W1[(size_t)(32 * j + 16 * (f / 12) + r) * F3_H + 32 * (f % 12) + kp];
        }
        v[e] = mi_f32_to_bf16(x);
    }
    uint4 o;
    o.x = v[0] | ((uint32_t)v[1] << 16); o.y = v[2] | ((uint32_t)v[3] << 16);
    o.z = v[4] | ((uint32_t)v[5] << 16); o.w = v[6] | ((uint32_t)v[7] << 16);
    *reinterpret_cast<uint4*>(dst + (size_t)idx * 8) = o;
}

int mi_ffn3_pack(hipStream_t st, const float* Wo, const float* W1, const float* W2, void* stream_dst,
                 const float* bo, const float* g1, const float* be1, const float* b1, const float* b2, const float* g2, const float* be2,
                 float* prm_dst)
{
    const int n = F3_NBLK * 24 * 64;
    ffn3_pack_kernel<<<(n + 255) / 256, 256, 0, st>>>(Wo, W1, W2, reinterpret_cast<uint16_t*>(stream_dst));
    MI_HIP(hipGetLastError());
    const float* src[6] = {bo, g1, be1, b2, g2, be2};
    for (int i = 0; i < 6; ++i) MI_HIP(hipMemcpyAsync(prm_dst + i * F3_H, src[i], F3_H * sizeof(float), hipMemcpyDeviceToDevice, st));
    MI_HIP(hipMemcpyAsync(prm_dst + 6 * F3_H, b1, F3_I * sizeof(float), hipMemcpyDeviceToDevice, st));
    return MI_OK;
}

// ---------------------------------------------------------------- device helpers
__device__ __forceinline__ uint32_t f3_pack2(float lo, float hi)
{
    bf16x2 v;
    v[0] = (__bf16)lo;
    v[1] = (__bf16)hi;
    return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ float f3_lo(uint32_t w) { return __builtin_bit_cast(float, w << 16); }
__device__ __forceinline__ float f3_hi(uint32_t w) { return __builtin_bit_cast(float, w & 0xFFFF0000u); }

// erf-GELU as in encoder.hip (gelu_erf2): max(x, 0) - |x| / 2 * erfc(|x| / sqrt 2), erfc(z) ~= 2^(-z q(z)), one v_exp per value.
// Written one value per instruction: beside MFMAs the packed-fp32 forms cost more than two plain ones
// (MI355X_MICROARCH.md, "price of one filler beside MFMAs").
__device__ __forceinline__ float f3_gelu(float x)
{
    const float z = fabsf(x) * 0.70710678118654752440f;
    float q = __builtin_fmaf(z, 0.0029441735549769255f, -0.0295900954803597f);
    q = __builtin_fmaf(q, z, 0.14866568630789534f);
    q = __builtin_fmaf(q, z, 0.9185093434591132f);
    q = __builtin_fmaf(q, z, 1.6278890079463915f);
    const float e = __builtin_amdgcn_exp2f(-(q * z));
    const float r = __builtin_amdgcn_fmed3f(x, 0.0f, __builtin_inff());
    return __builtin_fmaf(z * -0.70710678118654752440f, e, r);
}

typedef const __attribute__((address_space(3))) f32x4 f3_lds_cf32x4;
__device__ __forceinline__ f32x4 f3_ld4(const __attribute__((address_space(3))) float* p) { return *reinterpret_cast<f3_lds_cf32x4*>(p); }

__device__ __forceinline__ float f3_group_sum(float v)
{
    // a token's channels live in lanes l, l + 16, l + 32, l + 48 (and their registers)
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}

// ---------------------------------------------------------------- the kernel
// The wave's [384 x 32] fp32 output tile is OWNED BY THE ASM below: tile (T, tt) -- output channels 16 T .., token group tt --
// is a[8 T + 4 tt : 8 T + 4 tt + 3], named literally in every MFMA, and hipcc never sees it (left to the register allocator,
// a 192-register accumulator tile beside 96 registers of B fragments ended in ~180 accvgpr copies per stage and MFMAs
// whose destination was not their source).  F3_ACC_CLOBBERS makes the kernel descriptor allocate a0 .. a191; hipcc's own
// values must fit the 256 architectural VGPRs -- the build is audited for spills and for v_accvgpr_* outside these
// statements (tools/audit_ffn3.sh).
#define F3_ACC_CLOBBERS "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127", "a128", "a129", "a130", "a131", "a132", "a133", "a134", "a135", "a136", "a137", "a138", "a139", "a140", "a141", "a142", "a143", "a144", "a145", "a146", "a147", "a148", "a149", "a150", "a151", "a152", "a153", "a154", "a155", "a156", "a157", "a158", "a159", "a160", "a161", "a162", "a163", "a164", "a165", "a166", "a167", "a168", "a169", "a170", "a171", "a172", "a173", "a174", "a175", "a176", "a177", "a178", "a179", "a180", "a181", "a182", "a183", "a184", "a185", "a186", "a187", "a188", "a189", "a190", "a191"

// acc(T, 0) += A x B0, acc(T, 1) += A x B1
template <int T>
__device__ __forceinline__ void f3_mfma_acc(const bf16x8& a, const bf16x8& b0, const bf16x8& b1)
{
    asm volatile("v_mfma_f32_16x16x32_bf16 a[%c3:%c4], %0, %1, a[%c3:%c4]\n\t"
                 "v_mfma_f32_16x16x32_bf16 a[%c5:%c6], %0, %2, a[%c5:%c6]"
                 :: "v"(a), "v"(b0), "v"(b1), "n"(8 * T), "n"(8 * T + 3), "n"(8 * T + 4), "n"(8 * T + 7));
}
// x0 += A x B0, x1 += A x B1 on accumulators hipcc owns (vector registers)
__device__ __forceinline__ void f3_mfma_x(f32x4& x0, f32x4& x1, const bf16x8& a, const bf16x8& b0, const bf16x8& b1)
{
    // (x0 / x1 may come straight out of a v_mov hipcc placed in front -- the bias -- and it pads nothing it cannot see)
    asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_bf16 %0, %2, %3, %0\n\t"
                 "v_mfma_f32_16x16x32_bf16 %1, %2, %4, %1"
                 : "+v"(x0), "+v"(x1) : "v"(a), "v"(b0), "v"(b1));
}
template <int T, int TT>
__device__ __forceinline__ f32x4 f3_acc_read()
{
    f32x4 v;
    asm volatile("v_accvgpr_read_b32 %0, a[%c4]\n\tv_accvgpr_read_b32 %1, a[%c5]\n\t"
                 "v_accvgpr_read_b32 %2, a[%c6]\n\tv_accvgpr_read_b32 %3, a[%c7]"
                 : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3])
                 : "n"(8 * T + 4 * TT), "n"(8 * T + 4 * TT + 1), "n"(8 * T + 4 * TT + 2), "n"(8 * T + 4 * TT + 3));
    return v;
}
template <int T, int TT>
__device__ __forceinline__ void f3_acc_write(const f32x4& v)
{
    // (the values come out of VALU instructions hipcc scheduled: two wait states in front, as for any v_accvgpr_write)
    asm volatile("s_nop 1\n\tv_accvgpr_write_b32 a[%c4], %0\n\tv_accvgpr_write_b32 a[%c5], %1\n\t"
                 "v_accvgpr_write_b32 a[%c6], %2\n\tv_accvgpr_write_b32 a[%c7], %3"
                 :: "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]),
                    "n"(8 * T + 4 * TT), "n"(8 * T + 4 * TT + 1), "n"(8 * T + 4 * TT + 2), "n"(8 * T + 4 * TT + 3));
}
// between the last MFMA that wrote an accumulator and the first v_accvgpr_read of it (and between a v_accvgpr_write and
// the first MFMA that takes it as C): more wait states than any of the pairs needs
#define F3_ACC_SETTLE() asm volatile("s_nop 15\n\ts_nop 15" ::: "memory")

// amdgpu_num_vgpr(256): hipcc gets the 256 architectural VGPRs and NO accumulator registers of its own -- with a larger budget
// it parks values (LDS addresses) in a0, a1, ... across the statements above, inside the tile they own.
__global__ __launch_bounds__(256, 1) __attribute__((amdgpu_num_vgpr(256))) void ffn3_kernel(const uint16_t* __restrict__ ctx, const uint16_t* hin,
        const char* __restrict__ wstream, const float* __restrict__ prm_g, float eps, uint16_t* hout, int M, int ntiles)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    asm volatile("" ::: F3_ACC_CLOBBERS);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 15, g = lane >> 4;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
    float* const prm = reinterpret_cast<float*>(lds + F3_RING);

    // ---- the weight stream: block b of this workgroup (b = 108 * (tiles done) + block of the tile) lives in slot b % 5.
    // While block b is consumed, block b + 3 is requested into the slot block b - 2 left (every wave has passed the barrier at
    // the top of block b, so it has consumed block b - 1's fragments in MFMAs, and LDS reads return in order: block b - 2's
    // are all back).  Six pieces of 1 KB per wave and block: pieces 6 wave .. 6 wave + 5, SGPR base + one constant lane offset.
    const unsigned voff = (unsigned)lane * 16u;
    int cslot = 0;                                                 // slot of the block being consumed
    int rslot = 0, rtb = 0;                                        // next block to request: its slot, its number within the tile's stream
    auto request_piece = [&](int p) __attribute__((always_inline)) {
        const char* src = wstream + (size_t)rtb * F3_BLK + (size_t)wave * 6144 + p * 1024;
        const unsigned dst = lds_base + (unsigned)rslot * F3_BLK + (unsigned)wave * 6144u + (unsigned)p * 1024u;
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(src), "s"(dst) : "m0");
    };
    auto request_done = [&]() __attribute__((always_inline)) {
        rslot = rslot == F3_SLOTS - 1 ? 0 : rslot + 1;
        rtb = rtb == F3_NBLK - 1 ? 0 : rtb + 1;
    };
    // Top of a block: this wave's pieces of it have landed (only the 12 pieces of the two younger blocks may still be in
    // flight), then the barrier: everyone's pieces have landed.  The requests never stop -- past the workgroup's last block
    // they fetch the stream's next blocks into slots nobody reads again (72 KB of L2 reads per workgroup) -- so the count in
    // flight is the same at every block and the loop has no branch; the kernel's last instruction drains them.
    auto block_top = [&]() __attribute__((always_inline)) {
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(6 * (F3_AHEAD - 1)) : "memory");
        asm volatile("s_barrier" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };
    auto block_end = [&]() __attribute__((always_inline)) {
        cslot = cslot == F3_SLOTS - 1 ? 0 : cslot + 1;
        request_done();
    };

    // parameters -> LDS, once per workgroup
    for (int i = threadIdx.x; i < F3_PRM_FLOATS / 4; i += 256)
        reinterpret_cast<float4*>(prm)[i] = reinterpret_cast<const float4*>(prm_g)[i];
    // prologue of the stream: blocks 0 .. 2
    for (int b = 0; b < F3_AHEAD; ++b) {
#pragma unroll
        for (int p = 0; p < 6; ++p) request_piece(p);
        request_done();
    }
    __syncthreads();
    // one lane base for every parameter read (channels 4 g .. of a 16-channel block), the rest in the instructions' offsets;
    // opaque, or hipcc keeps one address register per (array, block)
    unsigned pl_addr = lds_base + F3_RING + 16u * (unsigned)g;
    asm volatile("" : "+v"(pl_addr));
    typedef const __attribute__((address_space(3))) float lds_cfloat;
    lds_cfloat* const pl = reinterpret_cast<lds_cfloat*>((size_t)pl_addr);
    lds_cfloat* const p_bo = pl, * const p_g1 = pl + F3_H, * const p_be1 = pl + 2 * F3_H;
    lds_cfloat* const p_b2 = pl + 3 * F3_H, * const p_g2 = pl + 4 * F3_H, * const p_be2 = pl + 5 * F3_H;
    lds_cfloat* const p_b1 = pl + 6 * F3_H;

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int m0 = tile * 128 + 32 * wave;
        bf16x8 cf[12][2];                                          // ctx^T, later h1^T: B fragments, k-step s, token group tt
        // ---- token loads: ctx (blocked [tile][12][128][32]: a k-step is one 64-B row of a block) and the residual rows;
        // the accumulators start at bo + h
        static_for<2>([&](auto TTc) {
            constexpr int tt = decltype(TTc)::value;
            const int tl = 32 * wave + 16 * tt + r;                // token within the tile (ctx is padded to whole tiles)
#pragma unroll
            for (int s = 0; s < 12; ++s)
                cf[s][tt] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(ctx + (((size_t)tile * 12 + s) * 128 + tl) * 32 + 8 * g));
            int tok = m0 + 16 * tt + r; if (tok > M - 1) tok = M - 1;
            static_for<24>([&](auto Tc) {
                constexpr int T = decltype(Tc)::value;
                const uint2 hv = *reinterpret_cast<const uint2*>(hin + (size_t)tok * F3_H + 16 * T + 4 * g);
                const f32x4 bv = f3_ld4(p_bo + 16 * T);
                f3_acc_write<T, tt>(f32x4{bv.x + f3_lo(hv.x), bv.y + f3_hi(hv.x), bv.z + f3_lo(hv.y), bv.w + f3_hi(hv.y)});
            });
        });
        F3_ACC_SETTLE();

        // ---- attention output projection: 12 blocks, one per k-step.  A block is 24 steps: fragment T + 3 is requested,
        // fragment T feeds two MFMAs (the wave's two token groups); every fourth step requests a piece of the block three ahead.
        static_for<12>([&](auto Sc) {
            constexpr int s = decltype(Sc)::value;
            block_top();
            const char* slot = lds + cslot * F3_BLK + lane * 16;
            bf16x8 a[24];
#pragma unroll
            for (int T = 0; T < 3; ++T) a[T] = *reinterpret_cast<const bf16x8*>(slot + T * 1024);
            static_for<24>([&](auto Tc) {
                constexpr int T = decltype(Tc)::value;
                if constexpr (T + 3 < 24) a[T + 3] = *reinterpret_cast<const bf16x8*>(slot + (T + 3) * 1024);
                f3_mfma_acc<T>(a[T], cf[s][0], cf[s][1]);
                if constexpr ((T & 3) == 1) request_piece(T >> 2);
                __builtin_amdgcn_sched_barrier(0);
            });
            block_end();
        });
        F3_ACC_SETTLE();

        // ---- LayerNorm 1 in registers; h1 (bf16) becomes the B fragments of FFN1 and the residual of LayerNorm 2
        static_for<2>([&](auto TTc) {
            constexpr int tt = decltype(TTc)::value;
            float s1 = 0.0f;
            static_for<24>([&](auto Tc) {
                const f32x4 v = f3_acc_read<decltype(Tc)::value, tt>();
                s1 += (v[0] + v[1]) + (v[2] + v[3]);
            });
            const float mean = f3_group_sum(s1) * (1.0f / F3_H);
            float s2 = 0.0f;
            static_for<24>([&](auto Tc) {
                const f32x4 v = f3_acc_read<decltype(Tc)::value, tt>();
#pragma unroll
                for (int i = 0; i < 4; ++i) { const float d = v[i] - mean; s2 = __builtin_fmaf(d, d, s2); }
            });
            const float inv = 1.0f / sqrtf(f3_group_sum(s2) * (1.0f / F3_H) + eps);
            static_for<12>([&](auto Sc) {
                constexpr int s = decltype(Sc)::value;
                uint32_t w[4];
                static_for<2>([&](auto Hc) {
                    constexpr int hh = decltype(Hc)::value, T = 2 * s + hh;
                    const f32x4 v = f3_acc_read<T, tt>();
                    const f32x4 gv = f3_ld4(p_g1 + 16 * T);
                    const f32x4 bv = f3_ld4(p_be1 + 16 * T);
                    const float y0 = (v[0] - mean) * inv * gv.x + bv.x, y1 = (v[1] - mean) * inv * gv.y + bv.y;
                    const float y2 = (v[2] - mean) * inv * gv.z + bv.z, y3 = (v[3] - mean) * inv * gv.w + bv.w;
                    w[2 * hh] = f3_pack2(y0, y1);
                    w[2 * hh + 1] = f3_pack2(y2, y3);
                    // the second product accumulates onto b2 + h1 (the bf16 values the next layer would read)
                    const f32x4 b2v = f3_ld4(p_b2 + 16 * T);
                    f3_acc_write<T, tt>(f32x4{b2v.x + f3_lo(w[2 * hh]), b2v.y + f3_hi(w[2 * hh]), b2v.z + f3_lo(w[2 * hh + 1]), b2v.w + f3_hi(w[2 * hh + 1])});
                });
                cf[s][tt] = __builtin_bit_cast(bf16x8, uint4{w[0], w[1], w[2], w[3]});
            });
        });
        F3_ACC_SETTLE();

        // ---- FFN: stream order W1_0, W1_1, (W2_j, W1_{j+2}) for j = 0 .. 45, W2_46, W2_47.
        // A single in-order wave per SIMD pays for every instruction it issues, so the GELU of a block (16 values per lane,
        // ~12 VALU instructions each) is cut into three parts per value and dealt one part per step over the 48 steps of the
        // two blocks that lie between the FFN1 product that made X and the FFN2 product that needs it: four VALU
        // instructions beside every pair of MFMAs.  X is double-buffered (Xa, Xb) and the loop body is two stages.
        // (An MFMA result needs wait states before a VALU instruction may read it, and hipcc pads nothing it cannot see:
        // every GELU part reads an X whose last MFMA lies a whole block boundary -- a barrier and more -- back.)
        f32x4 Xa[2][2], Xb[2][2];                                  // [rows 16 u ..][token group]: FFN1 results in flight
        bf16x8 G[2];                                               // gelu(FFN1) of the block whose FFN2 is next: B fragments
        float gz[16], gq[16], ge[16];                              // GELU in progress: value k = 8 u + 4 tt + i  <->  X[u][tt][i]
        uint32_t gw[8];                                            // packed pairs: word 4 tt + 2 u + (i >> 1)
        auto gelu_part = [&](const f32x4 (&X)[2][2], auto Kc, auto Pc) __attribute__((always_inline)) {
            constexpr int k = decltype(Kc)::value, part = decltype(Pc)::value;
            constexpr int u = k >> 3, tt = (k >> 2) & 1, i = k & 3;
            const float x = X[u][tt][i];
            if constexpr (part == 0) {
                gz[k] = fabsf(x) * 0.70710678118654752440f;
                gq[k] = __builtin_fmaf(__builtin_fmaf(gz[k], 0.0029441735549769255f, -0.0295900954803597f), gz[k], 0.14866568630789534f);
                asm volatile("" : "+v"(gz[k]), "+v"(gq[k]));        // the part stays in ITS step (hipcc would sink it to its use)
            } else if constexpr (part == 1) {
                const float q = __builtin_fmaf(__builtin_fmaf(gq[k], gz[k], 0.9185093434591132f), gz[k], 1.6278890079463915f);
                ge[k] = __builtin_amdgcn_exp2f(-(q * gz[k]));
                asm volatile("" : "+v"(ge[k]));
            } else {
                float rl;
                asm("v_max_f32 %0, 0, %1" : "=v"(rl) : "v"(x));    // (fmaxf on an opaque value costs a canonicalising v_max in front)
                ge[k] = __builtin_fmaf(gz[k] * -0.70710678118654752440f, ge[k], rl);
                if constexpr ((i & 1) == 1) {
                    gw[4 * tt + 2 * u + (i >> 1)] = f3_pack2(ge[k - 1], ge[k]);
                    asm volatile("" : "+v"(gw[4 * tt + 2 * u + (i >> 1)]));
                } else {
                    asm volatile("" : "+v"(ge[k]));
                }
            }
        };
        auto gelu_finish = [&]() __attribute__((always_inline)) {
            G[0] = __builtin_bit_cast(bf16x8, uint4{gw[0], gw[1], gw[2], gw[3]});
            G[1] = __builtin_bit_cast(bf16x8, uint4{gw[4], gw[5], gw[6], gw[7]});
        };
        // step t of a block whose GELU share is values K0 .. K0 + NV - 1 of Xg, three steps per value (NV = 8: all 24 steps)
        auto gelu_step = [&](const f32x4 (&Xg)[2][2], auto Tc, auto K0c, auto NVc) __attribute__((always_inline)) {
            constexpr int t = decltype(Tc)::value, K0 = decltype(K0c)::value, NV = decltype(NVc)::value;
            if constexpr (NV == 8) {
                gelu_part(Xg, IC<K0 + t / 3>{}, IC<t % 3>{});
            } else if constexpr (NV == 16) {                       // the first and the last block: two parts per step
                constexpr int p0 = 2 * t, p1 = 2 * t + 1;          // parts 0 .. 47 in value-major order
                gelu_part(Xg, IC<p0 / 3>{}, IC<p0 % 3>{});
                gelu_part(Xg, IC<p1 / 3>{}, IC<p1 % 3>{});
            }
        };
        // FFN1 of block j into Xd (= b1 + W1_j h1^T); GELU share of Xg alongside
        auto ffn1 = [&](int j, f32x4 (&Xd)[2][2], const f32x4 (&Xg)[2][2], auto K0c, auto NVc) __attribute__((always_inline)) {
            block_top();
            const char* slot = lds + cslot * F3_BLK + lane * 16;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const f32x4 bv = f3_ld4(p_b1 + 32 * j + 16 * u);
                Xd[u][0] = f32x4{bv.x, bv.y, bv.z, bv.w};
                Xd[u][1] = Xd[u][0];
            }
            bf16x8 a[24];                                          // step f = 2 s + u: rows 16 u .., k-step s = fragment 12 u + s
#pragma unroll
            for (int f = 0; f < 3; ++f) a[f] = *reinterpret_cast<const bf16x8*>(slot + (12 * (f & 1) + (f >> 1)) * 1024);
            static_for<24>([&](auto Fc) {
                constexpr int f = decltype(Fc)::value, s = f >> 1, u = f & 1;
                if constexpr (f + 3 < 24) a[f + 3] = *reinterpret_cast<const bf16x8*>(slot + (12 * ((f + 3) & 1) + ((f + 3) >> 1)) * 1024);
                f3_mfma_x(Xd[u][0], Xd[u][1], a[f], cf[s][0], cf[s][1]);
                gelu_step(Xg, Fc, K0c, NVc);
                if constexpr ((f & 3) == 1) request_piece(f >> 2);
                __builtin_amdgcn_sched_barrier(0);
            });
            block_end();
        };
        // FFN2 of the block whose gelu is in G; GELU share of Xg alongside
        auto ffn2 = [&](const f32x4 (&Xg)[2][2], auto K0c, auto NVc) __attribute__((always_inline)) {
            block_top();
            const char* slot = lds + cslot * F3_BLK + lane * 16;
            bf16x8 a[24];
#pragma unroll
            for (int T = 0; T < 3; ++T) a[T] = *reinterpret_cast<const bf16x8*>(slot + T * 1024);
            static_for<24>([&](auto Tc) {
                constexpr int T = decltype(Tc)::value;
                if constexpr (T + 3 < 24) a[T + 3] = *reinterpret_cast<const bf16x8*>(slot + (T + 3) * 1024);
                f3_mfma_acc<T>(a[T], G[0], G[1]);
                gelu_step(Xg, Tc, K0c, NVc);
                if constexpr ((T & 3) == 1) request_piece(T >> 2);
                __builtin_amdgcn_sched_barrier(0);
            });
            block_end();
        };
        ffn1(0, Xa, Xa, IC<0>{}, IC<0>{});                         // X_0
        ffn1(1, Xb, Xa, IC<0>{}, IC<16>{});                        // X_1 beside the whole gelu of X_0
        gelu_finish();                                             // G = gelu(X_0)
        for (int j = 0; j < F3_I / 32 - 2; j += 2) {
            ffn2(Xb, IC<0>{}, IC<8>{});                            // block j (G); first half of gelu(X_{j+1})
            ffn1(j + 2, Xa, Xb, IC<8>{}, IC<8>{});                 // X_{j+2}; second half
            gelu_finish();                                         // G = gelu(X_{j+1})
            ffn2(Xa, IC<0>{}, IC<8>{});                            // block j + 1
            ffn1(j + 3, Xb, Xa, IC<8>{}, IC<8>{});                 // X_{j+3}
            gelu_finish();                                         // G = gelu(X_{j+2})
        }
        ffn2(Xb, IC<0>{}, IC<16>{});                               // block 46 beside the whole gelu of X_47
        gelu_finish();
        ffn2(Xb, IC<0>{}, IC<0>{});                                // block 47
        F3_ACC_SETTLE();

        // ---- LayerNorm 2 and the rows out (8 bytes per lane and channel block)
        static_for<2>([&](auto TTc) {
            constexpr int tt = decltype(TTc)::value;
            float s1 = 0.0f;
            static_for<24>([&](auto Tc) {
                const f32x4 v = f3_acc_read<decltype(Tc)::value, tt>();
                s1 += (v[0] + v[1]) + (v[2] + v[3]);
            });
            const float mean = f3_group_sum(s1) * (1.0f / F3_H);
            float s2 = 0.0f;
            static_for<24>([&](auto Tc) {
                const f32x4 v = f3_acc_read<decltype(Tc)::value, tt>();
#pragma unroll
                for (int i = 0; i < 4; ++i) { const float d = v[i] - mean; s2 = __builtin_fmaf(d, d, s2); }
            });
            const float inv = 1.0f / sqrtf(f3_group_sum(s2) * (1.0f / F3_H) + eps);
            // rows past the last token are computed like the others (a column of the tile never mixes with another); their
            // lanes are switched off for the stores, once, here
            const int tok = m0 + 16 * tt + r;
            if (tok < M) {
                uint16_t* const orow = hout + (size_t)tok * F3_H + 4 * g;
                static_for<24>([&](auto Tc) {
                    constexpr int T = decltype(Tc)::value;
                    const f32x4 v = f3_acc_read<T, tt>();
                    const f32x4 gv = f3_ld4(p_g2 + 16 * T);
                    const f32x4 bv = f3_ld4(p_be2 + 16 * T);
                    const float y0 = (v[0] - mean) * inv * gv.x + bv.x, y1 = (v[1] - mean) * inv * gv.y + bv.y;
                    const float y2 = (v[2] - mean) * inv * gv.z + bv.z, y3 = (v[3] - mean) * inv * gv.w + bv.w;
                    *reinterpret_cast<uint2*>(orow + 16 * T) = make_uint2(f3_pack2(y0, y1), f3_pack2(y2, y3));
                });
            }
        });
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// M tokens: ctx blocked [ceil(M / 128)][12][128][32] bf16, hin / hout [M][384] bf16 (may alias), the layer's packed stream.
int mi_launch_ffn3(hipStream_t st, const uint16_t* ctx, const uint16_t* hin, const void* wstream, const float* prm, float eps,
                   uint16_t* hout, int M)
{
    static bool attr_set = false;
    if (!attr_set) {
        MI_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ffn3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, F3_LDS_BYTES));
        attr_set = true;
    }
    const int ntiles = (M + 127) / 128;
    static const int max_grid = getenv("MI_FFN3_GRID") ? atoi(getenv("MI_FFN3_GRID")) : 256;
    const int grid = ntiles < max_grid ? ntiles : max_grid;
    ffn3_kernel<<<grid, 256, F3_LDS_BYTES, st>>>(ctx, hin, reinterpret_cast<const char*>(wstream), prm, eps, hout, M, ntiles);
    MI_HIP(hipGetLastError());
    return MI_OK;
}
