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
//     operand, so a lane owns ONE token (column = lane & 31) and the accumulator registers run over output channels.
//     Four waves x 32 tokens, ONE wave per SIMD with the whole 512-entry register file: the [384 x 32] fp32 output tile
//     of a wave is 192 registers, the bf16 activations it multiplies (ctx^T, then h1^T) 96 more.
//   * a 32 x 32 accumulator tile is, converted to bf16 as it stands, two of the next product's B fragments (k = the
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
// Every LDS fragment read is a lane-linear ds_read_b128 (conflict-free by construction) and feeds one
// v_mfma_f32_32x32x16_bf16 (the wave's 32 tokens).  The MFMA shape is the single wave's issue budget: an in-order wave
// pays for every instruction it issues, a 16x16x32 MFMA blocks vector issue for 8 of its 16 cycles, a 32x32x16 one for 8
// of its 32 -- with the 16 x 16 shape a step of two MFMAs, one fragment read and four GELU instructions measured 58
// cycles against 32 of matrix-pipe time (cycle stamps, profiles/r03_ffn3_trace.md).
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
#ifndef F3_BAR2
#define F3_BAR2 1                             // 1: one barrier per TWO blocks (six slots, four blocks ahead); 0: one per block (5, 3)
#endif
#if F3_BAR2
#define F3_SLOTS 6
#define F3_AHEAD 4                            // block b + 4 is requested while block b is consumed (into the slot of block b - 2)
#else
#define F3_SLOTS 5
#define F3_AHEAD 3
#endif
#define F3_RING (F3_SLOTS * F3_BLK)
#define F3_NFRONT 12
#define F3_NBLK (F3_NFRONT + 2 * (F3_I / 32)) // 108 blocks per tile
#define F3_PRM_FLOATS (6 * F3_H + F3_I)       // bo g1 be1 b2 g2 be2 | b1
#define F3_LDS_BYTES (F3_RING + F3_PRM_FLOATS * 4)
// a block opens with its first fragments already on their way from LDS (nx[], requested behind the barrier inside the
// block before)
#define BLOCK_OPEN() \
    const char* const slot = lds + cslot * F3_BLK + lane * 16; \
    const char* const nslot = lds + (cslot == F3_SLOTS - 1 ? 0 : cslot + 1) * F3_BLK + lane * 16; \
    bf16x8 a[24]; \
    _Pragma("unroll") for (int f_ = 0; f_ < F3_RA; ++f_) a[f_] = nx[f_]
#ifndef F3_RA
#define F3_RA 3                               // fragments requested from LDS ahead of the one being multiplied
#endif
#define F3_PIECES_EARLY ((24 - F3_RA + 2) / 4) // pieces a block requests (steps 1, 5, ...) before the barrier at step 24 - F3_RA

static_assert(F3_LDS_BYTES <= 160 * 1024, "LDS");
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

// One thread per (block, fragment, lane): its 8 bf16 elements.  Lane (r = lane & 31, h = lane >> 5), element e; a fragment is
// the A operand of one v_mfma_f32_32x32x16_bf16: 32 rows x 16 k.  kp(s) = 16 s + 8 (e >> 2) + 4 h + (e & 3) is the row of a
// 32 x 32 accumulator tile that element e of lane half h holds once registers 8 s .. 8 s + 7 are converted to bf16.
//   Wo block b, fragment 12 sl + T (k-step 2 b + sl, rows 32 T ..):  Wo[32 T + r][16 (2 b + sl) + 8 h + e]      natural k (B = ctx^T from memory)
//   W1_j, fragment 2 T + s (k = channels 32 T + 16 s ..):            W1[32 j + r][32 T + kp(s)]
//   W2_j, fragment 2 T + s (rows 32 T .., k = the block's columns):  W2[32 T + r][32 j + kp(s)]
__global__ void ffn3_pack_kernel(const float* __restrict__ Wo, const float* __restrict__ W1, const float* __restrict__ W2,
                                 uint16_t* __restrict__ dst)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= F3_NBLK * 24 * 64) return;
    const int lane = idx & 63, f = (idx >> 6) % 24, b = idx / (24 * 64);
    const int r = lane & 31, h = lane >> 5;
    uint16_t v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        float x;
        if (b < F3_NFRONT) {
            x = Wo[(size_t)(32 * (f % 12) + r) * F3_H + 16 * (2 * b + f / 12) + 8 * h + e];
        } else {
            bool w2;
            const int j = f3_main_block(b - F3_NFRONT, &w2);
            const int kp = 16 * (f & 1) + 8 * (e >> 2) + 4 * h + (e & 3);
            if (w2) x = W2[(size_t)(32 * (f >> 1) + r) * F3_I + 32 * j + kp];
            else x = W1[(size_t)(32 * j + r) * F3_H + 32 * (f >> 1) + kp];
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

// lanes l < 32 give y and take the upper lane's x; lanes l >= 32 give x and take the lower lane's y  (v_permlane32_swap_b32:
// x[32 + r] <-> y[r]).  The two lanes of a token hold channels 4 h .. 4 h + 3 of every 8: two swaps turn a lane's two
// 8-byte quads into one 16-byte run of 8 channels (and back), so rows move as dwordx4.
__device__ __forceinline__ void f3_swap(uint32_t& x, uint32_t& y)
{
    const auto r = __builtin_amdgcn_permlane32_swap(x, y, false, false);
    x = r[0];
    y = r[1];
}

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

__device__ __forceinline__ float f3_pair_sum(float v)
{
    // a token's channels live in lanes l and l + 32 (and their registers)
    return v + __shfl_xor(v, 32, 64);
}

// ---------------------------------------------------------------- cycle stamps (debug build only: -DF3_TRACE, tools/trace_ffn3.py)
#ifdef F3_TRACE
__device__ unsigned long long f3_trace_buf[8][4][8];               // [workgroup 0..7][wave][stamp]
#define F3_STAMP(k) do { if (blockIdx.x < 8 && tile == (int)blockIdx.x && lane == 0) { unsigned long long _t; \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t) :: "memory"); f3_trace_buf[blockIdx.x][wave][k] = _t; } } while (0)
extern "C" int mi_ffn3_trace_read(unsigned long long* out)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(f3_trace_buf), sizeof(f3_trace_buf)) == hipSuccess ? 0 : -1;
}
#else
#define F3_STAMP(k) do { } while (0)
#endif

// ---------------------------------------------------------------- the kernel
// The wave's [384 x 32] fp32 output tile is OWNED BY THE ASM below: tile T -- output channels 32 T .. 32 T + 31 of the wave's 32
// tokens -- is a[16 T : 16 T + 15], named literally in every MFMA, and hipcc never sees it (left to the register allocator,
// a 192-register accumulator tile beside 96 registers of B fragments ended in ~180 accvgpr copies per stage and MFMAs
// whose destination was not their source).  Register i of a tile, lane (r = lane & 31, h = lane >> 5): channel
// 32 T + 8 (i >> 2) + 4 h + (i & 3) of token r.  F3_ACC_CLOBBERS makes the kernel descriptor allocate a0 .. a191; hipcc's own
// values must fit the architectural VGPRs: the build is audited for spills and for accumulator registers outside these
// statements (tests/test_ffn3_audit.py).
#define F3_ACC_CLOBBERS "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127", "a128", "a129", "a130", "a131", "a132", "a133", "a134", "a135", "a136", "a137", "a138", "a139", "a140", "a141", "a142", "a143", "a144", "a145", "a146", "a147", "a148", "a149", "a150", "a151", "a152", "a153", "a154", "a155", "a156", "a157", "a158", "a159", "a160", "a161", "a162", "a163", "a164", "a165", "a166", "a167", "a168", "a169", "a170", "a171", "a172", "a173", "a174", "a175", "a176", "a177", "a178", "a179", "a180", "a181", "a182", "a183", "a184", "a185", "a186", "a187", "a188", "a189", "a190", "a191"

typedef float f32x16 __attribute__((ext_vector_type(16)));

// acc(T) += A x B
template <int T>
__device__ __forceinline__ void f3_mfma_acc(const bf16x8& a, const bf16x8& b)
{
    asm volatile("v_mfma_f32_32x32x16_bf16 a[%c2:%c3], %0, %1, a[%c2:%c3]" :: "v"(a), "v"(b), "n"(16 * T), "n"(16 * T + 15));
}
// x += A x B on an accumulator hipcc owns (vector registers)
template <bool PAD>
__device__ __forceinline__ void f3_mfma_x(f32x16& x, const bf16x8& a, const bf16x8& b)
{
    // (PAD: x may come straight out of a v_mov hipcc placed in front -- the bias -- and it pads nothing it cannot see)
    if constexpr (PAD) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(x) : "v"(a), "v"(b));
    else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(x) : "v"(a), "v"(b));
}
// registers 4 Q .. 4 Q + 3 of tile T: four consecutive channels
template <int T, int Q>
__device__ __forceinline__ f32x4 f3_acc_read()
{
    f32x4 v;
    asm volatile("v_accvgpr_read_b32 %0, a[%c4]\n\tv_accvgpr_read_b32 %1, a[%c5]\n\t"
                 "v_accvgpr_read_b32 %2, a[%c6]\n\tv_accvgpr_read_b32 %3, a[%c7]"
                 : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3])
                 : "n"(16 * T + 4 * Q), "n"(16 * T + 4 * Q + 1), "n"(16 * T + 4 * Q + 2), "n"(16 * T + 4 * Q + 3));
    return v;
}
template <int T, int Q>
__device__ __forceinline__ void f3_acc_write(const f32x4& v)
{
    // (the values come out of VALU instructions hipcc scheduled: two wait states in front, as for any v_accvgpr_write)
    asm volatile("s_nop 1\n\tv_accvgpr_write_b32 a[%c4], %0\n\tv_accvgpr_write_b32 a[%c5], %1\n\t"
                 "v_accvgpr_write_b32 a[%c6], %2\n\tv_accvgpr_write_b32 a[%c7], %3"
                 :: "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]),
                    "n"(16 * T + 4 * Q), "n"(16 * T + 4 * Q + 1), "n"(16 * T + 4 * Q + 2), "n"(16 * T + 4 * Q + 3));
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
    const int r = lane & 31, h = lane >> 5;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
    float* const prm = reinterpret_cast<float*>(lds + F3_RING);

    // ---- the weight stream: block b of this workgroup (b = 108 * (tiles done) + block of the tile) lives in slot b % 5.
    // While block b is consumed, block b + 3 is requested into the slot block b - 2 left (every wave has passed the barrier at
    // the top of block b, so it has consumed block b - 1's fragments in MFMAs, and LDS reads return in order: block b - 2's
    // are all back).  Six pieces of 1 KB per wave and block: pieces 6 wave .. 6 wave + 5, SGPR base + one constant lane offset.
    // A block's six pieces share ONE source base (SGPR pair) and two LDS bases (M0): the instruction's immediate offset
    // moves the global address and the LDS address alike, pieces 0 .. 3 are offsets 0 .. 3072, pieces 4 and 5 take the lane
    // offset + 4096 and M0 + 4096 (the immediate ends at 4095).  The bases advance by one block per block, by additions: a
    // lone wave pays ~4 cycles for every scalar instruction too, and the address arithmetic of six independent pieces was
    // ~30 of them at every block boundary.
    unsigned voff = (unsigned)lane * 16u, voff2 = (unsigned)lane * 16u + 4096u;
    asm volatile("" : "+v"(voff), "+v"(voff2));                    // (two registers; hipcc would re-add the constant per piece)
    int cslot = 0;                                                 // slot of the block being consumed
    int rslot = 0, rtb = 0;                                        // next block to request: its slot, its number within the tile's stream
    const char* const wsrc0 = wstream + (size_t)wave * 6144;
    const unsigned rdst0 = lds_base + (unsigned)wave * 6144u;
    const char* rsrc = wsrc0;
    unsigned rdst = rdst0;
    auto request_piece = [&](auto Pc) __attribute__((always_inline)) {
        constexpr int p = decltype(Pc)::value;
        const unsigned vo = p < 4 ? voff : voff2, dst = rdst;      // (locals: asm operands alone do not capture)
        const char* const src = rsrc;
#ifndef F3_ABL_DMA                                                 // (F3_ABL_*: timing ablations of the debug build, wrong results)
        if constexpr (p < 4)
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:%3" :: "v"(vo), "s"(src), "s"(dst), "n"(1024 * p) : "m0");
        else
            asm volatile("s_add_u32 m0, %2, 0x1000\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:%3" :: "v"(vo), "s"(src), "s"(dst), "n"(1024 * (p - 4)) : "m0", "scc");
#else
        asm volatile("" :: "v"(vo), "s"(src), "s"(dst));
#endif
    };
    auto request_done = [&]() __attribute__((always_inline)) {
        const bool wrap_s = rslot == F3_SLOTS - 1, wrap_b = rtb == F3_NBLK - 1;
        rslot = wrap_s ? 0 : rslot + 1;
        rtb = wrap_b ? 0 : rtb + 1;
        rdst = wrap_s ? rdst0 : rdst + F3_BLK;
        rsrc = wrap_b ? wsrc0 : rsrc + F3_BLK;
    };
    // The blocks are software-pipelined across their boundary: at step 24 - F3_RA of block b this wave's pieces of block
    // b + 1 have landed (only the pieces of block b + 2 and the five of block b + 3 requested so far may still be in flight),
    // the barrier says everyone's have, and the last F3_RA steps request block b + 1's first fragments from LDS (nx[]), so
    // the first MFMA of a block never waits for a barrier plus an LDS round trip (stamps: ~200 of ~1000 cycles per block
    // before).  The requests never stop -- past the workgroup's last block they fetch the stream's next blocks into slots
    // nobody reads again (72 KB of L2 reads per workgroup) -- so the count in flight is the same at every block and the
    // loop has no branch; the kernel's last instruction drains them.
    // With F3_BAR2 only the odd blocks carry the barrier, and it covers the next TWO blocks (four waves in lockstep on one LDS
    // cost ~100 cycles per barrier: stamps with the barrier ablated).
    // Slot reuse: a piece requested in block b overwrites the slot of block b - 2.  Every wave has passed the barrier inside
    // block b - 1 (or b - 2, F3_BAR2 and b odd -- then it has at least ISSUED its last LDS read of block b - 2, at step
    // 23 - F3_RA, before the barrier, a block and more before the request, and the LDS serves its queue in order).
    bf16x8 nx[F3_RA];                                              // fragments 0 .. F3_RA - 1 of the next block
    auto next_block_barrier = [&](auto Startc) __attribute__((always_inline)) {
        __builtin_amdgcn_sched_barrier(0);
#ifndef F3_ABL_DMA
        // in flight behind the pieces waited for: at the start, the prologue's later blocks; in a block, the blocks requested
        // since plus this block's early pieces
        constexpr int later = decltype(Startc)::value ? 6 * (F3_AHEAD - 1 - F3_BAR2) : 6 * (F3_AHEAD - 2 - F3_BAR2) + F3_PIECES_EARLY;
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(later) : "memory");
#endif
#ifndef F3_ABL_BARRIER
        asm volatile("s_barrier" ::: "memory");
#else
        asm volatile("" ::: "memory");
#endif
        __builtin_amdgcn_sched_barrier(0);
    };
    auto block_end = [&]() __attribute__((always_inline)) {
        cslot = cslot == F3_SLOTS - 1 ? 0 : cslot + 1;
        request_done();
    };
    // one step of a block: LDS read-ahead (this block's fragment f + F3_RA, or the next block's first ones behind the barrier)
    // into a[] / nx[]; the caller's MFMA and fillers follow
    auto step_reads = [&](bf16x8 (&a)[24], const char* slot, const char* nslot, auto Fc, auto Oddc) __attribute__((always_inline)) {
        constexpr int f = decltype(Fc)::value;
        if constexpr (f == 24 - F3_RA && (!F3_BAR2 || decltype(Oddc)::value)) next_block_barrier(IC<0>{});
        if constexpr (f + F3_RA < 24) a[f + F3_RA] = *reinterpret_cast<const bf16x8*>(slot + (f + F3_RA) * 1024);
        else nx[f + F3_RA - 24] = *reinterpret_cast<const bf16x8*>(nslot + (f + F3_RA - 24) * 1024);
    };
    auto step_request = [&](auto Fc) __attribute__((always_inline)) {
        constexpr int f = decltype(Fc)::value;
        if constexpr ((f & 3) == 1) request_piece(IC<(f >> 2)>{});
    };

    // parameters -> LDS, once per workgroup
    for (int i = threadIdx.x; i < F3_PRM_FLOATS / 4; i += 256)
        reinterpret_cast<float4*>(prm)[i] = reinterpret_cast<const float4*>(prm_g)[i];
    // prologue of the stream: blocks 0 .. 2
    for (int b = 0; b < F3_AHEAD; ++b) {
        static_for<6>([&](auto Pc) { request_piece(Pc); });
        request_done();
    }
    __syncthreads();
    next_block_barrier(IC<1>{});                                   // block 0 (and block 1) landed
#pragma unroll
    for (int f = 0; f < F3_RA; ++f) nx[f] = *reinterpret_cast<const bf16x8*>(lds + lane * 16 + f * 1024);
    // one lane base for every parameter read (channels 4 h .. of an 8-channel group), the rest in the instructions' offsets;
    // opaque, or hipcc keeps one address register per (array, block)
    unsigned pl_addr = lds_base + F3_RING + 16u * (unsigned)h;
    asm volatile("" : "+v"(pl_addr));
    typedef const __attribute__((address_space(3))) float lds_cfloat;
    lds_cfloat* const pl = reinterpret_cast<lds_cfloat*>((size_t)pl_addr);
    lds_cfloat* const p_bo = pl, * const p_g1 = pl + F3_H, * const p_be1 = pl + 2 * F3_H;
    lds_cfloat* const p_b2 = pl + 3 * F3_H, * const p_g2 = pl + 4 * F3_H, * const p_be2 = pl + 5 * F3_H;
    lds_cfloat* const p_b1 = pl + 6 * F3_H;

    // ctx^T of a tile (blocked [tile][12][128][32]: two k-steps are one 64-B row of a block): requested for the first tile
    // here, for every later one behind the FFN loop of the tile before -- its 96 registers are free from the last FFN1 block
    // on, and the requests run under LayerNorm 2 and the row stores instead of adding to the next tile's load burst.
    bf16x8 cf[24];                                                 // ctx^T, later h1^T: B fragments, k-step s (16 channels)
    auto load_ctx = [&](int tile) __attribute__((always_inline)) {
        const int tl = 32 * wave + r;                              // token within the tile (ctx is padded to whole tiles)
#pragma unroll
        for (int s = 0; s < 24; ++s)
            cf[s] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(ctx + (((size_t)tile * 12 + (s >> 1)) * 128 + tl) * 32 + 16 * (s & 1) + 8 * h));
    };
#ifndef F3_NO_CTX_AHEAD
    if ((int)blockIdx.x < ntiles) load_ctx(blockIdx.x);
#endif
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int m0 = tile * 128 + 32 * wave;
        F3_STAMP(0);
        // ---- token loads: the residual row (ctx^T is on its way already); the accumulators start at bo + h.  A lane loads 8 consecutive channels of its token (channels 16 s + 8 h ..); the
        // swap hands each lane of the pair its 4 + 4.  The parameter reads run one step ahead of their use (pr[]): a lone
        // wave that reads LDS and waits, 48 times, pays the LDS latency 48 times.
        {
            int tok = m0 + r; if (tok > M - 1) tok = M - 1;
            const uint16_t* const hrow = hin + (size_t)tok * F3_H + 8 * h;
            f32x4 pr[2][2];
            auto ld = [&](auto Sc) __attribute__((always_inline)) {
                constexpr int s = decltype(Sc)::value;
                pr[s & 1][0] = f3_ld4(p_bo + 16 * s);
                pr[s & 1][1] = f3_ld4(p_bo + 16 * s + 8);
            };
            static_for<2>([&](auto Gc) {                           // 12 requests at a time (48 registers beside the 96 of ctx^T)
                constexpr int S0 = 12 * decltype(Gc)::value;
                uint4 hv[12];
#pragma unroll
                for (int i = 0; i < 12; ++i) hv[i] = *reinterpret_cast<const uint4*>(hrow + 16 * (S0 + i));
                if constexpr (S0 == 0) {
#ifdef F3_NO_CTX_AHEAD
                    load_ctx(tile);
#endif
                    ld(IC<0>{});
                }
                static_for<12>([&](auto Ic) {
                    constexpr int i = decltype(Ic)::value, s = S0 + i, T = s >> 1, Q = 2 * (s & 1);
                    if constexpr (s + 1 < 24) ld(IC<s + 1>{});
                    uint4 v = hv[i];
                    f3_swap(v.x, v.z);
                    f3_swap(v.y, v.w);
                    const f32x4 b0 = pr[s & 1][0], b1 = pr[s & 1][1];
                    f3_acc_write<T, Q>(f32x4{b0.x + f3_lo(v.x), b0.y + f3_hi(v.x), b0.z + f3_lo(v.y), b0.w + f3_hi(v.y)});
                    f3_acc_write<T, Q + 1>(f32x4{b1.x + f3_lo(v.z), b1.y + f3_hi(v.z), b1.z + f3_lo(v.w), b1.w + f3_hi(v.w)});
                    __builtin_amdgcn_sched_barrier(0);
                });
            });
        }
        F3_ACC_SETTLE();
        F3_STAMP(1);

        // ---- attention output projection: 12 blocks of two k-steps.  A block is 24 steps: fragment f + F3_RA is requested,
        // fragment f feeds one MFMA; every fourth step requests a piece of the block three ahead.
        static_for<12>([&](auto Bc) {
            constexpr int b = decltype(Bc)::value;
            BLOCK_OPEN();
            static_for<24>([&](auto Fc) {
                constexpr int f = decltype(Fc)::value;
                step_reads(a, slot, nslot, Fc, IC<(b & 1)>{});
                f3_mfma_acc<f % 12>(a[f], cf[2 * b + f / 12]);
                step_request(Fc);
                __builtin_amdgcn_sched_barrier(0);
            });
            block_end();
        });
        F3_ACC_SETTLE();
        F3_STAMP(2);

        // ---- LayerNorm 1 in registers; h1 (bf16) becomes the B fragments of FFN1 and the residual of LayerNorm 2.
        // One pass over the accumulator registers for the statistics, shifted by the token's first channel (the two lanes of a
        // token share it) so that E[d^2] - E[d]^2 does not cancel; a second pass normalises, two quads (one k-step of FFN1)
        // per step, with the step's twelve parameter quads requested a step ahead.
        auto row_stats = [&](float& scale, float& shift) __attribute__((always_inline)) {
            const f32x4 first = f3_acc_read<0, 0>();
            const float K = __shfl(first[0], r, 64);
            const f32x4 nK = {-K, -K, -K, -K};
            f32x4 sv = {0.0f, 0.0f, 0.0f, 0.0f}, qv = {0.0f, 0.0f, 0.0f, 0.0f};
            static_for<48>([&](auto Ic) {
                const f32x4 d = f3_acc_read<decltype(Ic)::value / 4, decltype(Ic)::value % 4>() + nK;
                sv += d;
                qv = __builtin_elementwise_fma(d, d, qv);
            });
            const float m1 = f3_pair_sum((sv[0] + sv[1]) + (sv[2] + sv[3])) * (1.0f / F3_H);
            const float var = fmaxf(__builtin_fmaf(-m1, m1, f3_pair_sum((qv[0] + qv[1]) + (qv[2] + qv[3])) * (1.0f / F3_H)), 0.0f);
            scale = 1.0f / sqrtf(var + eps);
            shift = -(K + m1) * scale;                             // (x - mean) / sd as one multiply-add
        };
        {
            float inv, nm;
            row_stats(inv, nm);
            const f32x4 inv4 = {inv, inv, inv, inv}, nm4 = {nm, nm, nm, nm};
            f32x4 pr[2][6];
            auto ld = [&](auto Sc) __attribute__((always_inline)) {
                constexpr int s = decltype(Sc)::value;
                pr[s & 1][0] = f3_ld4(p_g1 + 16 * s);  pr[s & 1][1] = f3_ld4(p_g1 + 16 * s + 8);
                pr[s & 1][2] = f3_ld4(p_be1 + 16 * s); pr[s & 1][3] = f3_ld4(p_be1 + 16 * s + 8);
                pr[s & 1][4] = f3_ld4(p_b2 + 16 * s);  pr[s & 1][5] = f3_ld4(p_b2 + 16 * s + 8);
            };
            ld(IC<0>{});
            static_for<24>([&](auto Sc) {                          // k-step s = registers 8 (s & 1) .. of tile s >> 1
                constexpr int s = decltype(Sc)::value, T = s >> 1, Q = 2 * (s & 1);
                if constexpr (s + 1 < 24) ld(IC<s + 1>{});
                const f32x4 y0 = __builtin_elementwise_fma(__builtin_elementwise_fma(f3_acc_read<T, Q>(), inv4, nm4), pr[s & 1][0], pr[s & 1][2]);
                const f32x4 y1 = __builtin_elementwise_fma(__builtin_elementwise_fma(f3_acc_read<T, Q + 1>(), inv4, nm4), pr[s & 1][1], pr[s & 1][3]);
                uint32_t w0 = f3_pack2(y0[0], y0[1]), w1 = f3_pack2(y0[2], y0[3]), w2 = f3_pack2(y1[0], y1[1]), w3 = f3_pack2(y1[2], y1[3]);
                asm volatile("" : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3));     // (or hipcc converts every low half a second time)
                // the second product accumulates onto b2 + h1 (the bf16 values the next layer would read)
                const f32x4 c0 = pr[s & 1][4], c1 = pr[s & 1][5];
                f3_acc_write<T, Q>(f32x4{c0.x + f3_lo(w0), c0.y + f3_hi(w0), c0.z + f3_lo(w1), c0.w + f3_hi(w1)});
                f3_acc_write<T, Q + 1>(f32x4{c1.x + f3_lo(w2), c1.y + f3_hi(w2), c1.z + f3_lo(w3), c1.w + f3_hi(w3)});
                cf[s] = __builtin_bit_cast(bf16x8, uint4{w0, w1, w2, w3});
                __builtin_amdgcn_sched_barrier(0);
            });
        }
        F3_ACC_SETTLE();
        F3_STAMP(3);

        // ---- FFN: stream order W1_0, W1_1, (W2_j, W1_{j+2}) for j = 0 .. 45, W2_46, W2_47.
        // A single in-order wave per SIMD pays for every instruction it issues, so the GELU of a block (16 values per lane,
        // ~10 VALU instructions each) is dealt over the 48 steps of the two blocks that lie between the FFN1 product that
        // made X and the FFN2 product that needs it: four VALU instructions beside every MFMA (gelu_op below).  X is
        // double-buffered (Xa, Xb) and the loop body is two stages.
        // (An MFMA result needs wait states before a VALU instruction may read it, and hipcc pads nothing it cannot see:
        // every GELU part reads an X whose last MFMA lies a whole block boundary -- a barrier and more -- back.)
        f32x16 Xa, Xb;                                             // FFN1 results in flight: [32 columns of the block][32 tokens]
        bf16x8 G[2];                                               // gelu(FFN1) of the block whose FFN2 is next: B fragments, k-steps 0, 1
        float gz[16], gq[16], gr[16];                              // GELU in progress, value k = register k of X
        uint32_t gw[8];                                            // packed pairs (k - 1, k)
        // The GELU of a block (16 values per lane) rides in the MFMA gaps of the two blocks between the FFN1 product that made
        // X and the FFN2 product that needs it.  Two rules from the stamps and the issue-cost table (MI355X_MICROARCH.md):
        // a lone wave issues a DEPENDENT vector instruction only every ~8 cycles, an independent one every 4 -- so a step
        // carries one operation of each of FOUR values (a "quad", values 4 c .. 4 c + 3), never two links of one chain; and a
        // gap hides ~24 cycles of issue beside its ds_read and MFMA -- so at most four plain instructions, or one v_exp (8) and
        // three plain ones.  Twelve steps per quad, two quads per block:
        //   0  z = |x| / sqrt 2   1-4  Horner steps of q(z)   5  p = -(q z)
        //   6  e0, t0 t1 t2       7  e1, t3, m0 m1            8  e2, m2 m3, y0       (e = 2^p, t = -z / sqrt 2 = -|x| / 2,
        //   9  e3, y1 y2         10  y3, w01                  11  w23                  m = max(x, 0), y = t e + m, w = bf16 pair)
        auto gelu_op = [&](const f32x16& X, auto Cc, auto Oc) __attribute__((always_inline)) {
            constexpr int c = decltype(Cc)::value, op = decltype(Oc)::value, k0 = 4 * c;
            auto e_ = [&](int i) __attribute__((always_inline)) { gq[k0 + i] = __builtin_amdgcn_exp2f(gq[k0 + i]); };
            auto t_ = [&](int i) __attribute__((always_inline)) { gz[k0 + i] *= -0.70710678118654752440f; };
            auto m_ = [&](int i) __attribute__((always_inline)) { asm("v_max_f32 %0, 0, %1" : "=v"(gr[k0 + i]) : "v"(X[k0 + i])); };
            auto y_ = [&](int i) __attribute__((always_inline)) { gq[k0 + i] = __builtin_fmaf(gz[k0 + i], gq[k0 + i], gr[k0 + i]); };
            if constexpr (op == 0) {
#pragma unroll
                for (int i = 0; i < 4; ++i) gz[k0 + i] = fabsf(X[k0 + i]) * 0.70710678118654752440f;
                asm volatile("" : "+v"(gz[k0]), "+v"(gz[k0 + 1]), "+v"(gz[k0 + 2]), "+v"(gz[k0 + 3]));
            } else if constexpr (op >= 1 && op <= 4) {
                constexpr float cs[4] = {-0.0295900954803597f, 0.14866568630789534f, 0.9185093434591132f, 1.6278890079463915f};
#pragma unroll
                for (int i = 0; i < 4; ++i) gq[k0 + i] = __builtin_fmaf(op == 1 ? 0.0029441735549769255f : gq[k0 + i], gz[k0 + i], cs[op - 1]);
                asm volatile("" : "+v"(gq[k0]), "+v"(gq[k0 + 1]), "+v"(gq[k0 + 2]), "+v"(gq[k0 + 3]));
            } else if constexpr (op == 5) {
#pragma unroll
                for (int i = 0; i < 4; ++i) gq[k0 + i] = -(gq[k0 + i] * gz[k0 + i]);
                asm volatile("" : "+v"(gq[k0]), "+v"(gq[k0 + 1]), "+v"(gq[k0 + 2]), "+v"(gq[k0 + 3]));
            } else if constexpr (op == 6) {
                e_(0); t_(0); t_(1); t_(2);
                asm volatile("" : "+v"(gq[k0]), "+v"(gz[k0]), "+v"(gz[k0 + 1]), "+v"(gz[k0 + 2]));
            } else if constexpr (op == 7) {
                e_(1); t_(3); m_(0); m_(1);
                asm volatile("" : "+v"(gq[k0 + 1]), "+v"(gz[k0 + 3]), "+v"(gr[k0]), "+v"(gr[k0 + 1]));
            } else if constexpr (op == 8) {
                e_(2); m_(2); m_(3); y_(0);
                asm volatile("" : "+v"(gq[k0 + 2]), "+v"(gr[k0 + 2]), "+v"(gr[k0 + 3]), "+v"(gq[k0]));
            } else if constexpr (op == 9) {
                e_(3); y_(1); y_(2);
                asm volatile("" : "+v"(gq[k0 + 3]), "+v"(gq[k0 + 1]), "+v"(gq[k0 + 2]));
            } else if constexpr (op == 10) {
                y_(3);
                gw[2 * c] = f3_pack2(gq[k0], gq[k0 + 1]);
                asm volatile("" : "+v"(gq[k0 + 3]), "+v"(gw[2 * c]));
            } else if constexpr (op == 11) {
                gw[2 * c + 1] = f3_pack2(gq[k0 + 2], gq[k0 + 3]);
                asm volatile("" : "+v"(gw[2 * c + 1]));
            }
        };
        auto gelu_finish = [&]() __attribute__((always_inline)) {
            G[0] = __builtin_bit_cast(bf16x8, uint4{gw[0], gw[1], gw[2], gw[3]});
            G[1] = __builtin_bit_cast(bf16x8, uint4{gw[4], gw[5], gw[6], gw[7]});
        };
        // step t of a block whose GELU share is quads C0 .. C0 + NQ - 1 of Xg (NQ = 2: twelve steps per quad; NQ = 4, the
        // first and the last block of a tile: two quads per step)
        auto gelu_step = [&](const f32x16& Xg, auto Tc, auto C0c, auto NQc) __attribute__((always_inline)) {
            constexpr int t = decltype(Tc)::value, C0 = decltype(C0c)::value, NQ = decltype(NQc)::value;
            if constexpr (NQ == 2) {
                gelu_op(Xg, IC<C0 + t / 12>{}, IC<t % 12>{});
            } else if constexpr (NQ == 4) {
                gelu_op(Xg, IC<t / 12>{}, IC<t % 12>{});
                gelu_op(Xg, IC<2 + t / 12>{}, IC<t % 12>{});
            }
        };
        // X = b1 of block j, requested a block before the FFN1 product that accumulates onto it (X is dead by then: its last
        // GELU operation lies in the block before)
        auto load_bias = [&](f32x16& X, int j) __attribute__((always_inline)) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 bv = f3_ld4(p_b1 + 32 * j + 8 * q);
                X[4 * q] = bv.x; X[4 * q + 1] = bv.y; X[4 * q + 2] = bv.z; X[4 * q + 3] = bv.w;
            }
        };
        // FFN1 of block j onto Xd (= b1 + W1_j h1^T); GELU share of Xg alongside
        auto ffn1 = [&](int j, f32x16& Xd, const f32x16& Xg, auto K0c, auto NVc, auto Oddc) __attribute__((always_inline)) {
            BLOCK_OPEN();
            static_for<24>([&](auto Fc) {
                constexpr int f = decltype(Fc)::value;
                step_reads(a, slot, nslot, Fc, Oddc);
                f3_mfma_x<f == 0>(Xd, a[f], cf[f]);
                gelu_step(Xg, Fc, K0c, NVc);
                step_request(Fc);
                __builtin_amdgcn_sched_barrier(0);
            });
            block_end();
        };
        // FFN2 of the block whose gelu is in G; GELU share of Xg alongside
        auto ffn2 = [&](const f32x16& Xg, auto K0c, auto NVc, f32x16& Xn, int jn, auto Oddc) __attribute__((always_inline)) {
            BLOCK_OPEN();
            static_for<24>([&](auto Fc) {
                constexpr int f = decltype(Fc)::value;
                if constexpr (f == 4) { if (jn >= 0) load_bias(Xn, jn); }
                step_reads(a, slot, nslot, Fc, Oddc);
                f3_mfma_acc<f / 2>(a[f], G[f & 1]);
                gelu_step(Xg, Fc, K0c, NVc);
                step_request(Fc);
                __builtin_amdgcn_sched_barrier(0);
            });
            block_end();
        };
        load_bias(Xa, 0);
        load_bias(Xb, 1);
        // (the last argument: the block's number within the tile is odd -- the blocks that carry the barrier when F3_BAR2)
        ffn1(0, Xa, Xa, IC<0>{}, IC<0>{}, IC<0>{});                // X_0
        ffn1(1, Xb, Xa, IC<0>{}, IC<4>{}, IC<1>{});                // X_1 beside the whole gelu of X_0
        gelu_finish();                                             // G = gelu(X_0)
        for (int j = 0; j < F3_I / 32 - 2; j += 2) {
            ffn2(Xb, IC<0>{}, IC<2>{}, Xa, j + 2, IC<0>{});        // block j (G); first half of gelu(X_{j+1}); Xa = b1 of block j + 2
            ffn1(j + 2, Xa, Xb, IC<2>{}, IC<2>{}, IC<1>{});        // X_{j+2}; second half
            gelu_finish();                                         // G = gelu(X_{j+1})
            ffn2(Xa, IC<0>{}, IC<2>{}, Xb, j + 3, IC<0>{});        // block j + 1
            ffn1(j + 3, Xb, Xa, IC<2>{}, IC<2>{}, IC<1>{});        // X_{j+3}
            gelu_finish();                                         // G = gelu(X_{j+2})
        }
        ffn2(Xb, IC<0>{}, IC<4>{}, Xa, -1, IC<0>{});               // block 46 beside the whole gelu of X_47
        gelu_finish();
        ffn2(Xb, IC<0>{}, IC<0>{}, Xa, -1, IC<1>{});               // block 47
        F3_ACC_SETTLE();
        F3_STAMP(4);
#ifndef F3_NO_CTX_AHEAD
        if (tile + (int)gridDim.x < ntiles) load_ctx(tile + (int)gridDim.x);
#endif

        // ---- LayerNorm 2 and the row out: 16 bytes per lane and k-step (the swap of the loads, backwards), 32 contiguous
        // bytes per row and instruction
        {
            float inv, nm;
            row_stats(inv, nm);
            const f32x4 inv4 = {inv, inv, inv, inv}, nm4 = {nm, nm, nm, nm};
            // rows past the last token are computed like the others (a column of the tile never mixes with another); their
            // lanes are switched off, once, here -- both lanes of a token or neither, so the swaps stay among active pairs
            const int tok = m0 + r;
            if (tok < M) {
                uint16_t* const orow = hout + (size_t)tok * F3_H + 8 * h;
                f32x4 pr[2][4];
                auto ld = [&](auto Sc) __attribute__((always_inline)) {
                    constexpr int s = decltype(Sc)::value;
                    pr[s & 1][0] = f3_ld4(p_g2 + 16 * s);  pr[s & 1][1] = f3_ld4(p_g2 + 16 * s + 8);
                    pr[s & 1][2] = f3_ld4(p_be2 + 16 * s); pr[s & 1][3] = f3_ld4(p_be2 + 16 * s + 8);
                };
                ld(IC<0>{});
                static_for<24>([&](auto Sc) {
                    constexpr int s = decltype(Sc)::value, T = s >> 1, Q = 2 * (s & 1);
                    if constexpr (s + 1 < 24) ld(IC<s + 1>{});
                    const f32x4 y0 = __builtin_elementwise_fma(__builtin_elementwise_fma(f3_acc_read<T, Q>(), inv4, nm4), pr[s & 1][0], pr[s & 1][2]);
                    const f32x4 y1 = __builtin_elementwise_fma(__builtin_elementwise_fma(f3_acc_read<T, Q + 1>(), inv4, nm4), pr[s & 1][1], pr[s & 1][3]);
                    uint32_t w0 = f3_pack2(y0[0], y0[1]), w1 = f3_pack2(y0[2], y0[3]), w2 = f3_pack2(y1[0], y1[1]), w3 = f3_pack2(y1[2], y1[3]);
                    f3_swap(w0, w2);
                    f3_swap(w1, w3);
                    *reinterpret_cast<uint4*>(orow + 16 * s) = uint4{w0, w1, w2, w3};
                    __builtin_amdgcn_sched_barrier(0);
                });
            }
        }
        F3_STAMP(5);
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
