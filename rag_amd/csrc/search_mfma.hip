// search_mfma.hip — the fast flat-search path on gfx950 (MI355X), for BF16 indexes and (through
// a bf16 shadow copy of the rows) F32 indexes, IP and L2 metrics:
//
//   phase 1  mfma_scan_kernel    bf16 MFMA scoring of a query tile against a streamed row
//                                slice, with a per-query running threshold: only scores above
//                                the slice's current K'-th best leave the accumulators.
//   phase 2  mfma_finish_kernel  per (query, slice group): pick the candidates that can still be
//                                in the top-k from their MFMA scores, re-score those in the
//                                canonical fp32 fmaf order, sort by (score, id).
//            (merge_kernel of search_exact.hip joins the slice groups)
//   phase 3  mfma_certify_kernel per query: CERTIFY that no row that was filtered out can belong
//                                to the top-k; otherwise the query is re-run on the exact path.
//
// The MFMA accumulation order is not the canonical order (and for an F32 index the shadow rows
// are rounded), so phase-1 scores v~ are only used to discard.  With E a bound on
// |v~ - canonical| (mf_err_bound: 4 d 2^-24 relative for the accumulation order, plus 2^-8 |q||x|
// for a rounded shadow):
//   * every row dropped in phase 1 has v~ <= thr(slice), hence canonical <= max_slices(thr) + E;
//   * every candidate dropped in phase 2 has v~ < T_k - 2E where T_k is the k-th largest v~ of
//     its group, hence k candidates of the group are strictly better in canonical order.
// If the k-th best canonical score of the final list is strictly above max(thr) + E the list is
// provably the exact top-k.  L2 uses v = dot - xx/2 (the accumulators start at -xx/2), which
// orders rows like the negated distance.  Replaces faiss IndexFlat::search (advanced_rag.py:296).
//
// Data movement (d = 384: NKB = 3 k-blocks of 128 elements):
//   * queries are register-stationary: each wave keeps the B fragments of its 32 queries x d
//     (96 VGPRs at d = 384) for the whole kernel.  d <= 384: 8 waves per workgroup (two per
//     SIMD, 256 registers each, 256 queries per workgroup) so one wave's LDS reads hide under
//     its partner's MFMAs; d = 768: 4 waves, one per SIMD, 512-register budget.  Waves whose 32
//     queries are all padding only help with the loads.
//   * corpus rows stream HBM -> LDS by LDS-DMA (global_load_lds_dwordx4) in units of
//     128 rows x 256 B through a ring of PD + 1 buffers: PD units are in flight while one is
//     consumed, retired by counted s_waitcnt vmcnt(N) + a raw s_barrier (PD = 3 where the scan is
//     HBM-bound, 1 where it is MFMA-bound).
//     Each 256-B row segment is XOR-swizzled by (row & 15) on the SOURCE side so the
//     ds_read_b128 A-fragment reads are bank-conflict free.
//   * workgroups that scan the same row slice carry the same (blockIdx % 8) so they share an
//     XCD's L2: a slice is fetched from HBM once per XCD pass, not once per query tile.
#include <stdlib.h>
#include <type_traits>
#include <utility>
#include "kernels.h"

template <int V> using IC = std::integral_constant<int, V>;
template <int... I, class F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, I...>, F&& f) { (f(IC<I>{}), ...); }
// f(IC<0>), f(IC<1>), ... f(IC<N-1>): a loop whose index is a compile-time constant in the body
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(std::make_integer_sequence<int, N>{}, static_cast<F&&>(f)); }

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

#define MF_TILE_R 128
#define MF_STAGE_BYTES (MF_TILE_R * 256)
#define MF_XXB 4                                   // ring depth of the per-tile xx / threshold buffers
#define MF_PH2_MAX 8192                            // candidate entries per (query, group) in LDS
#define MF_SORT_MAX 1024                           // survivors re-scored and sorted per (query, group)
#define MF_MIN_ROWS 32768
#define MF_SEED_CLASSES 64                         // disjoint row classes per query in the threshold seed pass

#ifndef MF_CAP_EXTRA
#define MF_CAP_EXTRA 32
#endif
__host__ __device__ constexpr int mf_cap(int kp) { return 2 * kp + MF_CAP_EXTRA; }
// Waves per workgroup and 16-query blocks per wave.  d <= 384: 8 waves x 32 queries (96 VGPRs of query fragments).
// d = 768: the 32x32x16 (HBM-bound) variant keeps 4 waves x 32 queries, one per SIMD with the 512-register budget;
// the 16x16x32 (MFMA-bound) variant runs 8 waves x 16 queries, two per SIMD: a wave alone on its SIMD has nothing
// to hide its barriers, DMA issue and LDS latency behind (0.84 PFLOP/s at 5M x 768 x 10k in round 1).
// At 16 queries per wave every 1-KB fragment read feeds ONE 16-cycle MFMA: four SIMDs then ask the LDS for 256 B/clk,
// all it has -- the variant is LDS-bound as much as MFMA-bound.  MF_768_QB = 2 / 3 build the 16x16x32 variant with
// four waves x 32 / 48 queries instead (half / a third of the reads; 288 fragment registers spill at 48): measured
// at 20M x 768 x 10k on one device, k = 10 / 100: 1,053 / 1,028 (8 x 16), 1,077 / 991 (4 x 32), 968 / 892 TFLOP/s
// (4 x 48) -- what the reads give back, a wave alone on its SIMD loses again.
#ifndef MF_768_QB
#define MF_768_QB 1
#endif
__host__ __device__ constexpr int mf_waves(int nkb, bool m16) { return (nkb <= 3 || (m16 && MF_768_QB == 1)) ? 8 : 4; }
__host__ __device__ constexpr int mf_qb(int nkb, bool m16) { return (nkb > 3 && m16) ? MF_768_QB : 2; }
__host__ __device__ constexpr int mf_qtile(int nkb, bool m16) { return 16 * mf_qb(nkb, m16) * mf_waves(nkb, m16); }

// Four A fragments (rows 32 m + col, m = 0..3) of one k-step; completion is the caller's
// s_waitcnt lgkmcnt (hipcc does not count inline-asm LDS reads).
__device__ __forceinline__ void lds_read4(bf16x8 (&a)[4], unsigned addr)
{
    asm volatile("ds_read_b128 %0, %4\n\t"
                 "ds_read_b128 %1, %4 offset:8192\n\t"
                 "ds_read_b128 %2, %4 offset:16384\n\t"
                 "ds_read_b128 %3, %4 offset:24576"
                 : "=&v"(a[0]), "=&v"(a[1]), "=&v"(a[2]), "=&v"(a[3]) : "v"(addr));
}

// Waits until at most `units` whole units (PPW LDS-DMA pieces each) of this wave are in flight.
// Younger stores / loads of the filter only make the wait stricter (vmcnt counts in issue order).
template <int PPW>
__device__ __forceinline__ void wait_units(int units)
{
    static_assert(PPW * 2 <= 63, "vmcnt immediate");
    if (units >= 2) { if constexpr (PPW == 4) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)" ::: "memory"); }
    else if (units == 1) { if constexpr (PPW == 4) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory"); }
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
}

// 16-byte LDS-DMA with the base address in SGPRs and a 32-bit per-lane offset (the builtin always
// takes a 64-bit VGPR address: two 64-bit vector adds per piece, and the hoisted pairs spilled).
__device__ __forceinline__ void dma16_saddr(const char* ubase, unsigned voff, unsigned lds_addr)
{
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                 :: "v"(voff), "s"(ubase), "s"(lds_addr) : "memory", "m0");
}

// 4-byte LDS-DMA past this CU's L1 (sc1), SGPR base + 32-bit lane offset
__device__ __forceinline__ void dma4_saddr_sc1(const void* ubase, unsigned voff, unsigned lds_addr)
{
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1 sc1"
                 :: "v"(voff), "s"(ubase), "s"(lds_addr) : "memory", "m0");
}
__device__ __forceinline__ void dma4_saddr(const void* ubase, unsigned voff, unsigned lds_addr)
{
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1"
                 :: "v"(voff), "s"(ubase), "s"(lds_addr) : "memory", "m0");
}
// lane * 4, formed where it is used: as a loop invariant hipcc hoists it, runs out of registers,
// and reloads it from scratch in the loop (a vmem load + vmcnt(0): the prefetch drains)
__device__ __forceinline__ unsigned lane_x4()
{
    unsigned l4;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0\n\tv_lshlrev_b32 %0, 2, %0" : "=v"(l4));
    return l4;
}

template <int N>
__device__ __forceinline__ void wait_vm()
{
    static_assert(N >= 0 && N <= 63, "vmcnt immediate");
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory");
}

// ---------------------------------------------------------------- phase 1
// PD = LDS-DMA prefetch distance in units, ring of 4 buffers.  Measured on one device at
// 100M x 384: a deep prefetch (PD = 3, issued at the top of an iteration) is 5-15 % faster where
// the scan is HBM-bound (one or two query tiles per slice); a shallow one is 1-5 % faster where it
// is MFMA-bound (the workgroups of a slice stay closer together in the XCD's L2): PD = 2, issued
// at the bottom of the iteration.  The plan picks.
// M16: score with v_mfma_f32_16x16x32_bf16 instead of 32x32x16 (same LDS image, same bytes read,
// same MFMA cycles per k; the guide measured a higher sustained clock on that shape).
#ifdef MF_TRACE
// Debug build only (tools/trace_scan.sh): cycle stamps of 64 consecutive units of four workgroups.
#ifndef MF_TRACE_U0
#define MF_TRACE_U0 3000
#endif
__device__ uint32_t mf_trace_buf[4][8][64][8];
__device__ __forceinline__ int mf_trace_slot(int b) { return b == 3 ? 0 : b == 300 ? 1 : b == 700 ? 2 : b == 1200 ? 3 : -1; }
#define MF_TRACE_LDS 16384
#else
#define MF_TRACE_LDS 0
#endif

// SEED (16x16x32 variant only): no lists, no thresholds -- the kernel keeps, per query, the maximum score of each of 64
// disjoint row classes (slice mod 4 x lane group g4 x row block mod 4) over the slice and folds them into
// gmax[q][MF_SEED_CLASSES] with atomic max (mi_launch_mfma_seed).
template <int NKB, int KP, bool L2, int PD, bool M16, bool SEED = false>
__global__ __launch_bounds__(64 * mf_waves(NKB, M16)) void mfma_scan_kernel(
    const uint16_t* __restrict__ rows, const float* __restrict__ xx, int64_t n, int pitch_bytes,
    const uint16_t* __restrict__ q_bf16, int nq, int nq_pad, int nqt, int64_t slice_rows,
    uint64_t* __restrict__ lists, int32_t* __restrict__ cnts, float* __restrict__ thrs, uint32_t* __restrict__ gthr,
    uint32_t* __restrict__ gmax)
{
    static_assert(!SEED || M16, "the seed pass runs the 16x16x32 variant");
    constexpr int WAVES = mf_waves(NKB, M16);
    constexpr int QB = mf_qb(NKB, M16);               // 16-query blocks per wave
    constexpr int QW = 16 * QB;                       // queries per wave
    static_assert(M16 || QB == 2, "the 32x32x16 variant holds 32 queries per wave");
    constexpr int PPW = 32 / WAVES;                   // 1-KiB LDS-DMA pieces per wave per unit
    constexpr int CAP = mf_cap(KP);
    constexpr int EPL = (CAP + 63) / 64;
    constexpr int KS = NKB * 8;                       // 16-wide k-steps per row
    constexpr int MF_PD = PD;
    constexpr int MF_NST = 4;                          // ring depth (units)
    constexpr int MF_XX_OFF = MF_NST * MF_STAGE_BYTES; // per-tile xx: [MF_XXB][128] f32
    constexpr int MF_GT_OFF = MF_XX_OFF + MF_XXB * MF_TILE_R * 4;   // shared-threshold words: [MF_XXB][waves][64] u32
    constexpr int MF_PG_OFF = MF_GT_OFF + MF_XXB * 8 * 256;        // progress words of the slice's workgroups: [MF_XXB][64] u32
    constexpr int MF_LDS_BYTES = MF_PG_OFF + MF_XXB * 256;
    static_assert(PD == 2 || PD == 3, "ring of 4 buffers: PD + 1 (+ 1 when the issue sits before the barrier)");
    __shared__ __attribute__((aligned(1024))) char lds[MF_LDS_BYTES + MF_TRACE_LDS];

    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
#ifdef MF_TRACE
    const uint32_t t_entry = (uint32_t)__builtin_amdgcn_s_memtime();
#endif
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5;
    const int col = lane & 31;

    // blocks with equal (blockIdx % 8) share an XCD (round-robin dispatch): give them one slice
    const int b = blockIdx.x;
    const int xcd = b & 7, j = b >> 3;
    const int slice = xcd + 8 * (j / nqt);
    const int qt = j % nqt;

    const int d_pad = pitch_bytes >> 1;
    const int64_t qglob = (int64_t)qt * (QW * WAVES) + wave * QW + col;   // this lane's query (32x32x16 variant)
    const bool active = (int64_t)qt * (QW * WAVES) + wave * QW < nq;      // wave-uniform

    const int64_t qbase = (int64_t)qt * (QW * WAVES) + wave * QW;         // first query of this wave
    const int l15 = lane & 15, g4 = lane >> 4;                            // 16x16x32 lane coordinates

    // ---- B fragments.  32x32x16: lane (col, h) holds query qglob, dims [16 s + 8 h, +8).
    // 16x16x32: lane (l15, g4) holds queries qbase + 16 cb + l15 (cb = 0, 1), dims [32 s + 8 g4, +8).
    bf16x8 bq[M16 ? 1 : KS];
    bf16x8 bq16[M16 ? KS / 2 : 1][QB];
    if (active) {
        if constexpr (M16) {
#pragma unroll
            for (int s = 0; s < KS / 2; ++s)
#pragma unroll
                for (int cb = 0; cb < QB; ++cb) {
                    const uint4 v = *reinterpret_cast<const uint4*>(q_bf16 + (qbase + 16 * cb + l15) * d_pad + 32 * s + 8 * g4);
                    bq16[s][cb] = __builtin_bit_cast(bf16x8, v);
                }
#pragma unroll
            for (int s = 0; s < KS / 2; ++s)
#pragma unroll
                for (int cb = 0; cb < QB; ++cb) asm volatile("" :: "v"(bq16[s][cb]));
        } else {
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const uint4 v = *reinterpret_cast<const uint4*>(q_bf16 + qglob * d_pad + 16 * s + 8 * h);
                bq[s] = __builtin_bit_cast(bf16x8, v);
            }
            // make hipcc wait for these loads HERE: with a register load still pending at the loop
            // header its waitcnt pass puts a vmcnt(0) in front of the first MFMA of every unit, which
            // drains the LDS-DMA prefetch that was just issued
#pragma unroll
            for (int s = 0; s < KS; ++s) asm volatile("" :: "v"(bq[s]));
        }
    }

    const int64_t row_begin = (int64_t)slice * slice_rows;
    int64_t row_end = row_begin + slice_rows;
    if (row_end > n) row_end = n;
    const int ntiles = (row_begin < row_end) ? (int)((row_end - row_begin + MF_TILE_R - 1) / MF_TILE_R) : 0;
    const int nunits = ntiles * NKB;

    // ---- LDS-DMA geometry: wave w issues pieces p = PPW w + i, 4 rows x 256 B each.
    // LDS is lane-linear (row 4p + lane/16, slot lane%16); the lane fetches the 16-B chunk that
    // belongs in that slot under the swizzle  slot = chunk ^ (row & 15).
    const int lr = lane >> 4;            // row within the piece
    const int lt = lane & 15;            // LDS slot
    // Source of piece i = pb0 + (wave-uniform unit offset + i * 4 rows) + swizzled chunk.  The chunk
    // that belongs in LDS slot lt of row 4p + lr is lt ^ ((4p + lr) & 15) = (lt ^ lr) ^ 4 (i & 3)
    // (4p = 16 (..) + 4 (i & 3), lr < 4), so one register pair and one xor constant per piece do.
    // Tiles may run past row_end / n by up to MF_TILE_R - 1 rows: the index allocates that much
    // slack behind its rows (capi.hip), and the filter masks those rows.
    // The address is formed as (wave-uniform 64-bit base) + (per-lane 32-bit offset) so the load takes
    // its base from SGPRs: no 64-bit vector adds per piece.
    const char* const ub0 = reinterpret_cast<const char*>(rows) + (row_begin + 4 * wave * PPW) * (int64_t)pitch_bytes;
    // Kept in a register where there is one (2 % faster than forming it per unit); the K' = 128
    // variants have none to spare (it would be spilled and reloaded with a vmcnt(0) per unit).
    constexpr bool VOFF_INLINE = (KP >= 128) && M16;
    const unsigned voff_reg = (unsigned)lr * (unsigned)pitch_bytes + ((unsigned)(lt ^ lr) << 4);   // pitch is a multiple of 256

    // gthr[q]: the best threshold any workgroup has published for query q (orderable word, 0 =
    // none yet).  Any slice's K'-th best is a lower bound of the global K'-th best, so every
    // workgroup may filter with it; a stale value is only a weaker, still valid, threshold.  It
    // arrives by LDS-DMA (sc1: past this CU's L1) with the tile's first unit.
    const uint32_t* const gsrc_u = gthr + qbase;         // wave-uniform; lanes 32..63 re-read words 0..31
    // prog[w]: tile reached by workgroup (slice, query tile w), + 1; 0 = not started, ~0 = done.
    // The workgroups that stream one slice share one XCD's L2; nothing else keeps them together,
    // and with a prefetch that never stalls they drift until most of their reads miss L2.  Every
    // fourth tile a workgroup publishes its position and, if it leads a workgroup of its cohort by
    // more than MF_MAX_LEAD tiles, naps until that one has caught up (bounded; the slowest never waits).
    uint32_t* const prog = gthr + nq_pad + (int64_t)slice * 64;
    // Measured at 100M x 384 x 10k on one device, lead limit 4 / 8 / 12 / 24 / none: 528 / 530 / 532 /
    // 545 / 547 ms, fabric reads 194 / 200 / 257 / 489 / 1267 GB, clock under load 2.0 vs 1.93 GHz.
    constexpr uint32_t MF_MAX_LEAD = 8, MF_COHORT = 512;
    // unit u = (tile, kb) lives in ring buffer u & (MF_NST - 1); its tile's xx in xx buffer tile & 3
    auto issue_extras = [&](int tile, int kb) __attribute__((always_inline)) {
        if (kb == 0) {
            const unsigned l4 = lane_x4();
            dma4_saddr_sc1(gsrc_u, l4 & (QW > 32 ? 252u : 124u), lds_base + MF_GT_OFF + ((tile & (MF_XXB - 1)) * 8 + wave) * 256);
            if constexpr (M16) {
                if (wave == 0) dma4_saddr_sc1(prog, l4, lds_base + MF_PG_OFF + (tile & (MF_XXB - 1)) * 256);
            }
        }
        if constexpr (L2) {
            if (kb == 0 && wave == 0) {
                const unsigned l4 = lane_x4();
#pragma unroll
                for (int half = 0; half < 2; ++half)
                    dma4_saddr(xx + row_begin + (int64_t)tile * MF_TILE_R + 64 * half, l4,
                               lds_base + MF_XX_OFF + (tile & (MF_XXB - 1)) * (MF_TILE_R * 4) + half * 256);
            }
        }
    };
    // pieces [I0, I0 + NI) of this wave's share of a unit
    auto issue_pieces = [&](int tile, int kb, int buf, auto I0c, auto NIc) __attribute__((always_inline)) {
        constexpr int I0 = decltype(I0c)::value, NI = decltype(NIc)::value;
        const char* const ub = ub0 + ((int64_t)tile * MF_TILE_R * pitch_bytes + kb * 256);      // wave-uniform
        unsigned voff0 = voff_reg;
        if constexpr (VOFF_INLINE) {
            const unsigned l4v = lane_x4(), lrv = l4v >> 6, ltv = (l4v >> 2) & 15u;
            voff0 = lrv * (unsigned)pitch_bytes + ((ltv ^ lrv) << 4);
        }
#pragma unroll
        for (int i = I0; i < I0 + NI; ++i)
            dma16_saddr(ub + (int64_t)i * 4 * pitch_bytes, voff0 ^ ((unsigned)(i & 3) << 6),
                        lds_base + buf * MF_STAGE_BYTES + (wave * PPW + i) * 1024);
    };
    auto issue_unit = [&](int tile, int kb, int buf) __attribute__((always_inline)) {
        issue_extras(tile, kb);
        issue_pieces(tile, kb, buf, IC<0>{}, IC<PPW>{});
    };
    auto issue_linear = [&](int u) __attribute__((always_inline)) {                      // u < nunits
        issue_unit(u / NKB, u % NKB, u & (MF_NST - 1));
    };

    // ---- per-lane filter state, replicated in lanes (col) and (col + 32)
    // Lanes of padding queries (q >= nq: zero vectors) start at +inf and never take a row.  Left at -inf they
    // cost more than the real ones: under L2 every row scores -xx/2, equal up to rounding, so their lists
    // kept compacting for the first three tiles of every slice (75 us each, traced at 200,000 rows x 1 query).
    float thr = (qglob < nq) ? -__builtin_inff() : __builtin_inff();
    int cnt = 0;
    uint64_t* const list = lists + ((int64_t)slice * nq_pad + qglob) * CAP;
    float thr2[QB];                                                                   // 16x16x32: the lane's QB queries
    int cnt2[QB];
#pragma unroll
    for (int cb = 0; cb < QB; ++cb) { thr2[cb] = (qbase + 16 * cb + l15 < nq) ? -__builtin_inff() : __builtin_inff(); cnt2[cb] = 0; }

    f32x16 acc[M16 ? 1 : 4];
    f32x4 acc16[M16 ? 8 : 1][QB];                                 // [16-row block][16-query block]
    auto init_acc = [&](int tile) __attribute__((always_inline)) {
        if constexpr (M16) {
            if constexpr (L2) {
                const float* xl = reinterpret_cast<const float*>(lds + MF_XX_OFF + (tile & (MF_XXB - 1)) * (MF_TILE_R * 4));
#pragma unroll
                for (int rb = 0; rb < 8; ++rb) {
                    const float4 v = *reinterpret_cast<const float4*>(xl + 16 * rb + 4 * g4);
#pragma unroll
                    for (int cb = 0; cb < QB; ++cb) {
                        acc16[rb][cb][0] = -0.5f * v.x; acc16[rb][cb][1] = -0.5f * v.y;
                        acc16[rb][cb][2] = -0.5f * v.z; acc16[rb][cb][3] = -0.5f * v.w;
                    }
                }
            }
            // IP: nothing to do, the tile's first MFMA per accumulator takes a zero C operand
        } else if constexpr (L2) {
            const float* xl = reinterpret_cast<const float*>(lds + MF_XX_OFF + (tile & (MF_XXB - 1)) * (MF_TILE_R * 4));
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 v = *reinterpret_cast<const float4*>(xl + 32 * m + 8 * g + 4 * h);
                    acc[m][4 * g + 0] = -0.5f * v.x; acc[m][4 * g + 1] = -0.5f * v.y;
                    acc[m][4 * g + 2] = -0.5f * v.z; acc[m][4 * g + 3] = -0.5f * v.w;
                }
        }
    };

    // A-fragment read address: row (32 m + col), chunk (2 s + h) ^ (row & 15)
    const int a_row_off = col * 256;
    const unsigned a_swz = (unsigned)(h ^ (lane & 15)) << 4;

    // ---- prologue: put MF_PD units in flight, retire unit 0
    // M16 variant: DMA lead in tenths of a unit.  25 = unit u + 3 issued at the bottom of unit u;
    // 10 = unit u + 2 issued right behind the barrier of unit u.  Measured on one device (20M and
    // 100M rows, 10k queries): 25 is 2-3 % faster at d = 384, 10 is 2 % faster at d = 768 and moves
    // half the bytes over the fabric (the workgroups of a slice stay closer in L2); 15 and 20 lose.
#ifndef MF_LEAD_768
#define MF_LEAD_768 10
#endif
    constexpr int MF_LEAD = (NKB >= 6) ? MF_LEAD_768 : 25;
#ifndef MF_ASYM_DMA
#define MF_ASYM_DMA 1
#endif
    constexpr bool MF_ASYM = (MF_ASYM_DMA != 0) && WAVES == 8;
    constexpr int MF_PRO = M16 ? (MF_LEAD >= 20 ? 3 : 2) : MF_PD;
#pragma unroll
    for (int u0 = 0; u0 < MF_PRO; ++u0)
        if (u0 < nunits) issue_linear(u0);
    if (nunits > 0) {
        if constexpr (M16) {
            wait_vm<0>();                                // once per workgroup: simply drain
        } else {
            const int younger = (nunits - 1 < MF_PD - 1) ? nunits - 1 : MF_PD - 1;
            wait_units<PPW>(younger);
        }
        __builtin_amdgcn_s_barrier();
        if (active) init_acc(0);
    }

    // ---- k-steps [S0, S0 + NS) of the unit in ring buffer `buf`, k-block KB of the queries.
    // A fragments are read one k-step ahead of the MFMAs that consume them.  The reads are inline
    // asm so hipcc cannot sink each one next to its use (it serialises ds_read -> wait -> MFMA
    // otherwise); LDS returns in order, so with the 4 reads of step s+1 just issued, lgkmcnt(4)
    // means the 4 reads of step s have landed.
    auto chain = [&](int buf, auto KBc, auto S0c, auto NSc) __attribute__((always_inline)) {
        constexpr int KB = decltype(KBc)::value, S0 = decltype(S0c)::value, NS = decltype(NSc)::value;
        // row (32 m + col), chunk (2 s + h) ^ (row & 15): (2 s + h) ^ x = (h ^ x) ^ 2 s, and the row base
        // is a multiple of 256, so step s is one xor with a constant on a single address register
        const unsigned cur = (lds_base + buf * MF_STAGE_BYTES + a_row_off) | a_swz;
        bf16x8 a[2][4];
        lds_read4(a[0], cur ^ (S0 << 5));
#pragma unroll
        for (int t = 0; t < NS; ++t) {
            const int s = S0 + t;
            if (t < NS - 1) {
                lds_read4(a[(t + 1) & 1], cur ^ ((s + 1) << 5));
                asm volatile("s_waitcnt lgkmcnt(4)"
                             : "+v"(a[t & 1][0]), "+v"(a[t & 1][1]), "+v"(a[t & 1][2]), "+v"(a[t & 1][3]));
            } else {
                asm volatile("s_waitcnt lgkmcnt(0)"
                             : "+v"(a[t & 1][0]), "+v"(a[t & 1][1]), "+v"(a[t & 1][2]), "+v"(a[t & 1][3]));
            }
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                if (!L2 && KB == 0 && s == 0)          // first touch in the tile: C = 0 (no zeroing pass)
                    acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[t & 1][m], bq[KB * 8 + s], f32x16{}, 0, 0, 0);
                else
                    acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[t & 1][m], bq[KB * 8 + s], acc[m], 0, 0, 0);
            }
        }
    };

    // 16x16x32: per 32-wide k-step s (4 per unit) eight A fragments, rows 16 rb + l15, chunk
    // (4 s + g4) ^ l15 = (g4 ^ l15) ^ 4 s; read and consumed four row blocks at a time, one
    // group ahead, 8 MFMAs (4 row blocks x 2 query blocks, 16 cycles each) per group.
    auto chain16 = [&](int buf, auto KBc, auto&& mid) __attribute__((always_inline)) {
        constexpr int KB = decltype(KBc)::value;
        const unsigned cur = (lds_base + buf * MF_STAGE_BYTES + l15 * 256) | ((unsigned)(g4 ^ l15) << 4);
        bf16x8 a[2][4];
        auto rd = [&](bf16x8 (&f)[4], int grp) __attribute__((always_inline)) {     // grp = 2 s + half
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
        if constexpr (QB == 1) {
            // 16 queries per wave: a fragment feeds ONE MFMA, so four reads cover only 64 cycles of MFMA -- less than the
            // LDS round trip.  Read a whole k-step (eight row blocks) ahead instead: 128 cycles of cover.
            bf16x8 b[2][8];
            auto rd8 = [&](bf16x8 (&f)[8], int ks) __attribute__((always_inline)) {
                rd(reinterpret_cast<bf16x8 (&)[4]>(f[0]), 2 * ks);
                rd(reinterpret_cast<bf16x8 (&)[4]>(f[4]), 2 * ks + 1);
            };
            rd8(b[0], 0);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (t < 3) {
                    rd8(b[(t + 1) & 1], t + 1);
                    asm volatile("s_waitcnt lgkmcnt(8)"
                                 : "+v"(b[t & 1][0]), "+v"(b[t & 1][1]), "+v"(b[t & 1][2]), "+v"(b[t & 1][3]),
                                   "+v"(b[t & 1][4]), "+v"(b[t & 1][5]), "+v"(b[t & 1][6]), "+v"(b[t & 1][7]));
                } else {
                    asm volatile("s_waitcnt lgkmcnt(0)"
                                 : "+v"(b[t & 1][0]), "+v"(b[t & 1][1]), "+v"(b[t & 1][2]), "+v"(b[t & 1][3]),
                                   "+v"(b[t & 1][4]), "+v"(b[t & 1][5]), "+v"(b[t & 1][6]), "+v"(b[t & 1][7]));
                }
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    if (!L2 && KB == 0 && t == 0)      // first touch in the tile: C = 0 (no zeroing pass)
                        acc16[r][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[t & 1][r], bq16[KB * 4 + t][0], f32x4{}, 0, 0, 0);
                    else
                        acc16[r][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[t & 1][r], bq16[KB * 4 + t][0], acc16[r][0], 0, 0, 0);
                }
                if (t == 1) mid();
            }
        } else {
        rd(a[0], 0);
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            if (t < 7) {
                rd(a[(t + 1) & 1], t + 1);
                asm volatile("s_waitcnt lgkmcnt(4)"
                             : "+v"(a[t & 1][0]), "+v"(a[t & 1][1]), "+v"(a[t & 1][2]), "+v"(a[t & 1][3]));
            } else {
                asm volatile("s_waitcnt lgkmcnt(0)"
                             : "+v"(a[t & 1][0]), "+v"(a[t & 1][1]), "+v"(a[t & 1][2]), "+v"(a[t & 1][3]));
            }
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int cb = 0; cb < QB; ++cb) {
                    if (!L2 && KB == 0 && t < 2)       // first touch in the tile: C = 0 (no zeroing pass)
                        acc16[4 * (t & 1) + r][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                            a[t & 1][r], bq16[KB * 4 + (t >> 1)][cb], f32x4{}, 0, 0, 0);
                    else
                        acc16[4 * (t & 1) + r][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                            a[t & 1][r], bq16[KB * 4 + (t >> 1)][cb], acc16[4 * (t & 1) + r][cb], 0, 0, 0);
                }
            if (t == 3) mid();
        }
        }
    };

    // Compacts the list of the wave's query ql (0..31) from cq entries to its KP best; returns the
    // KP-th largest score word.  The whole wave takes part.
    auto compact_list = [&](int ql, int cq) __attribute__((always_inline)) -> uint32_t {
        uint64_t* lq = lists + ((int64_t)slice * nq_pad + (qbase + ql)) * CAP;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        uint64_t e[EPL];
        uint32_t uo[EPL];
#pragma unroll
        for (int t = 0; t < EPL; ++t) {
            const int idx = t * 64 + lane;
            e[t] = (idx < cq) ? lq[idx] : 0ull;
            uo[t] = (uint32_t)(e[t] >> 32);     // 0 = below every valid score word
        }
        // KP-th largest score word by bitwise binary search
        uint32_t T = 0;
        for (int bit = 31; bit >= 0; --bit) {
            const uint32_t cand = T | (1u << bit);
            int c2 = 0;
#pragma unroll
            for (int t = 0; t < EPL; ++t)
                c2 += __builtin_popcountll(__builtin_amdgcn_ballot_w64(uo[t] >= cand));
            if (c2 >= KP) T = cand;
        }
        int n_gt = 0;
#pragma unroll
        for (int t = 0; t < EPL; ++t)
            n_gt += __builtin_popcountll(__builtin_amdgcn_ballot_w64(uo[t] > T));
        const int need_eq = KP - n_gt;
        const uint64_t lt_mask = (1ull << lane) - 1ull;
        int base_gt = 0, base_eq = 0;
#pragma unroll
        for (int t = 0; t < EPL; ++t) {
            const bool gt = uo[t] > T;
            const bool eq = (uo[t] == T);
            const uint64_t m_gt = __builtin_amdgcn_ballot_w64(gt);
            const uint64_t m_eq = __builtin_amdgcn_ballot_w64(eq);
            if (gt) lq[base_gt + __builtin_popcountll(m_gt & lt_mask)] = e[t];
            if (eq) {
                const int r = base_eq + __builtin_popcountll(m_eq & lt_mask);
                if (r < need_eq) lq[n_gt + r] = e[t];
            }
            base_gt += __builtin_popcountll(m_gt);
            base_eq += __builtin_popcountll(m_eq);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        return T;
    };

    // ---- filter: anything above the query's running threshold leaves the accumulators.
    // Rows are < 2^32 (plan); the row base changes per tile, so the per-element row numbers are
    // formed where they are used (hoisted out of the loop they would be spilled).
    // the shared thresholds of this tile arrived with its first unit; read ahead of the last chain
    auto refresh_thr = [&](int tile) __attribute__((always_inline)) {
        const char* gslot = lds + MF_GT_OFF + ((tile & (MF_XXB - 1)) * 8 + wave) * 256;
        if constexpr (M16) {
#pragma unroll
            for (int cb = 0; cb < QB; ++cb) {
                const uint32_t gw = *reinterpret_cast<const uint32_t*>(gslot + (16 * cb + l15) * 4);
                if (gw != 0) thr2[cb] = __builtin_fmaxf(thr2[cb], mi_unord_f32(gw));
            }
        } else {
            const uint32_t gw = *reinterpret_cast<const uint32_t*>(gslot + col * 4);
            if (gw != 0) thr = __builtin_fmaxf(thr, mi_unord_f32(gw));
        }
    };
    auto filter_tile = [&](int tile) __attribute__((always_inline)) {
        const uint32_t rend = (uint32_t)row_end;
        if constexpr (M16) {
            // the lane holds rows 16 rb + 4 g4 + reg of queries 16 cb + l15; a query's results sit in
            // the four lanes l15 + 16 g
            // common case first, as one wide max tree and ONE branch per tile (eight dependent
            // max -> compare -> branch rounds cost ~90 cycles each)
            {
                float m[QB];
#pragma unroll
                for (int cb = 0; cb < QB; ++cb) {
                    float m4[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) m4[i] = __builtin_fmaxf(acc16[0][cb][i], acc16[1][cb][i]);
#pragma unroll
                    for (int rb = 2; rb < 8; ++rb)
#pragma unroll
                        for (int i = 0; i < 4; ++i) m4[i] = __builtin_fmaxf(m4[i], acc16[rb][cb][i]);
                    m[cb] = __builtin_fmaxf(__builtin_fmaxf(m4[0], m4[1]), __builtin_fmaxf(m4[2], m4[3]));
                }
                bool any = m[0] > thr2[0];
#pragma unroll
                for (int cb = 1; cb < QB; ++cb) any = any || (m[cb] > thr2[cb]);
                if (__builtin_amdgcn_ballot_w64(any) == 0) return;
            }
            const uint32_t rowbase = (uint32_t)(row_begin + (int64_t)tile * MF_TILE_R) + 4u * (uint32_t)g4;
#pragma unroll
            for (int cb = 0; cb < QB; ++cb) {
                uint64_t* const list16 = lists + ((int64_t)slice * nq_pad + qbase + 16 * cb + l15) * CAP;
#pragma unroll
                for (int rp = 0; rp < 4; ++rp) {                  // 32 rows at a time, as the list capacity assumes
                    float bm = __builtin_fmaxf(acc16[2 * rp][cb][0], acc16[2 * rp][cb][1]);
#pragma unroll
                    for (int i = 2; i < 8; ++i) bm = __builtin_fmaxf(bm, acc16[2 * rp + (i >> 2)][cb][i & 3]);
                    if (__builtin_amdgcn_ballot_w64(bm > thr2[cb]) == 0) continue;
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const float v = acc16[2 * rp + (i >> 2)][cb][i & 3];
                        const uint32_t row = rowbase + (uint32_t)(16 * (2 * rp + (i >> 2)) + (i & 3));
                        const bool pass = (v > thr2[cb]) && (row < rend);
                        const uint64_t mask = __builtin_amdgcn_ballot_w64(pass);
                        if (mask == 0) continue;
                        const int b0 = (int)((mask >> l15) & 1ull), b1 = (int)((mask >> (l15 + 16)) & 1ull);
                        const int b2 = (int)((mask >> (l15 + 32)) & 1ull), b3 = (int)((mask >> (l15 + 48)) & 1ull);
                        const int lower = (g4 > 0 ? b0 : 0) + (g4 > 1 ? b1 : 0) + (g4 > 2 ? b2 : 0);
                        if (pass) list16[cnt2[cb] + lower] = ((uint64_t)mi_ord_f32(v) << 32) | row;
                        cnt2[cb] += b0 + b1 + b2 + b3;
                    }
                    uint64_t need = __builtin_amdgcn_ballot_w64(cnt2[cb] > CAP - 32) & 0xFFFFull;
                    while (need) {
                        const int ql = __builtin_ctzll(need);
                        need &= need - 1;
                        const uint32_t T = compact_list(16 * cb + ql, __builtin_amdgcn_readlane(cnt2[cb], ql));
                        if (l15 == ql) {
                            cnt2[cb] = KP;
                            const float tnew = mi_unord_f32(T);
                            if (tnew > thr2[cb]) {
                                thr2[cb] = tnew;
                                if (g4 == 0) __hip_atomic_fetch_max(gthr + qbase + 16 * cb + l15, T, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            }
                        }
                    }
                }
            }
        } else {
            const uint32_t rowbase = (uint32_t)(row_begin + (int64_t)tile * MF_TILE_R) + 4u * (uint32_t)h;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const f32x16 c = acc[m];
                float bm = __builtin_fmaxf(c[0], c[1]);
#pragma unroll
                for (int i = 2; i < 16; ++i) bm = __builtin_fmaxf(bm, c[i]);
                if (__builtin_amdgcn_ballot_w64(bm > thr) == 0) continue;
                // ---------------- rare path: append the passing elements of this 32x32 block
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float v = c[i];
                    const uint32_t row = rowbase + (uint32_t)(32 * m + (i & 3) + 8 * (i >> 2));
                    const bool pass = (v > thr) && (row < rend);
                    const uint64_t mask = __builtin_amdgcn_ballot_w64(pass);
                    if (mask == 0) continue;
                    const int partner = (int)((mask >> (lane ^ 32)) & 1ull);
                    const int pos = cnt + ((h == 1) ? partner : 0);
                    if (pass) list[pos] = ((uint64_t)mi_ord_f32(v) << 32) | row;
                    cnt += (pass ? 1 : 0) + partner;
                }
                // ---------------- compact the lists that can no longer take a whole block
                uint64_t need = __builtin_amdgcn_ballot_w64(cnt > CAP - 32) & 0xFFFFFFFFull;
                while (need) {
                    const int ql = __builtin_ctzll(need);
                    need &= need - 1;
                    const uint32_t T = compact_list(ql, __builtin_amdgcn_readlane(cnt, ql));
                    if (col == ql) {
                        cnt = KP;
                        const float tnew = mi_unord_f32(T);
                        if (tnew > thr) {
                            thr = tnew;
                            if (h == 0) __hip_atomic_fetch_max(gthr + qglob, T, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                    }
                }
            }
        }
    };

    // ---- SEED: running maxima of the lane's scores in 4 classes (row block mod 4); with the lane group g4 and the slice
    // number mod 4 that makes MF_SEED_CLASSES = 64 disjoint row classes per query.  (More classes per lane spill in the
    // d = 384 instance, and a scratch reload drains the LDS-DMA queue: 8 per lane 109 us, 16 per lane 245 us for the pass
    // against 52.)
    float gm[SEED ? QB : 1][4];
    if constexpr (SEED) {
#pragma unroll
        for (int cb = 0; cb < QB; ++cb)
#pragma unroll
            for (int j = 0; j < 4; ++j) gm[cb][j] = -__builtin_inff();
    }
    auto seed_tile = [&]() __attribute__((always_inline)) {
        if constexpr (M16 && SEED) {
#pragma unroll
            for (int cb = 0; cb < QB; ++cb)
#pragma unroll
                for (int rb = 0; rb < 8; ++rb) {                                        // (fmax drops a NaN score)
                    const float m01 = __builtin_fmaxf(acc16[rb][cb][0], acc16[rb][cb][1]), m23 = __builtin_fmaxf(acc16[rb][cb][2], acc16[rb][cb][3]);
                    gm[cb][rb & 3] = __builtin_fmaxf(gm[cb][rb & 3], __builtin_fmaxf(m01, m23));
                }
        }
    };

    // my LDS reads are retired; my pieces of unit u + 1 have landed; everyone meets
    auto end_of_unit = [&](int u) __attribute__((always_inline)) {
        const int rem = nunits - 2 - u;               // units younger than u + 1 that exist
        wait_units<PPW>(rem < 0 ? 0 : (rem < MF_PD - 1 ? rem : MF_PD - 1));
        __builtin_amdgcn_s_barrier();
    };

    // (A half-unit stagger of waves 4..7 against their SIMD partners was built and dropped in round
    // 1: the second code path costs ~15 registers and hipcc spilled query fragments into the loop.)
    int u = 0;
#ifdef MF_TRACE
    const int tslot = mf_trace_slot(blockIdx.x);
    uint32_t* const tr = reinterpret_cast<uint32_t*>(lds + MF_LDS_BYTES) + wave * 512;
    auto stamp = [&](int k) __attribute__((always_inline)) {
        if (tslot >= 0 && u >= MF_TRACE_U0 && u < MF_TRACE_U0 + 64) {
            const uint32_t t = (uint32_t)__builtin_amdgcn_s_memtime();
            if (lane == 0) tr[(u - MF_TRACE_U0) * 8 + k] = t;
        }
    };
    const uint32_t t_loop = (uint32_t)__builtin_amdgcn_s_memtime();
#else
    auto stamp = [&](int) __attribute__((always_inline)) {};
#endif
    if constexpr (M16) {
        // MFMA-bound variant.  The barrier sits in the MIDDLE of a unit's chain and certifies the NEXT
        // unit: past it, unit u + 1 has landed for everyone and everyone has left unit u - 1.  A unit
        // boundary is then no synchronisation point: a wave runs from the last MFMA of unit u straight
        // into the first reads of unit u + 1, the two waves of a SIMD drift apart (the older one
        // wins the pipe) and each one's filter / DMA issue / first-read latency hides behind the
        // other's MFMAs.  Unit u + 3 is issued at the bottom of unit u into the buffer of unit u - 1
        // (ring of 4); at the barrier of unit u the only younger DMA in flight is unit u + 2's.
        for (int tile = 0; tile < ntiles; ++tile) {
            static_for<NKB>([&](auto KBc) __attribute__((always_inline)) {
                constexpr int kb = decltype(KBc)::value;
                constexpr bool g2 = ((kb + 2) % NKB) == 0;          // unit u + 2 carries the threshold (and xx) loads
                stamp(0);
                if constexpr (kb == 0) {
                    if (wave == 0 && (tile & 3) == 0 && nqt <= 64) {
                        const uint32_t mine = (uint32_t)tile + 1u;
                        const unsigned l4 = lane_x4();
                        if (l4 == 0) __hip_atomic_store(prog + qt, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        uint32_t v = *reinterpret_cast<const uint32_t*>(lds + MF_PG_OFF + (tile & (MF_XXB - 1)) * 256 + l4);
                        auto behind = [&](uint32_t w) { return w != 0u && w < mine && mine - w > MF_MAX_LEAD && mine - w <= MF_COHORT; };
                        if (__builtin_amdgcn_ballot_w64(behind(v)) != 0) {
                            for (int spin = 0; spin < 32; ++spin) {
                                __builtin_amdgcn_s_sleep(32);
                                v = __hip_atomic_load(reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(prog) + (size_t)l4),
                                                      __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                if (__builtin_amdgcn_ballot_w64(behind(v)) == 0) break;
                            }
                        }
                    }
                }
                if constexpr (kb == NKB - 1 && !SEED) { if (active) refresh_thr(tile); }
                auto mid = [&]() __attribute__((always_inline)) {
                    stamp(1);
                    if constexpr (MF_LEAD == 10 && MF_ASYM) {
                        // The two waves of a SIMD (w and w + 4) issue their pieces of unit u + 2 on opposite sides of the
                        // barrier: the younger one BEFORE it (behind its MFMAs), the older one right behind it.  Issued by both
                        // at the same point the ~500 cycles of LDS-DMA issue leave the matrix pipe idle; this way each wave's
                        // issue runs under the other's MFMAs.  The buffer of unit u + 2 held unit u - 2, which everyone left
                        // before the barrier of unit u - 1.
                        if (wave >= 4) {
                            if (u + 2 < nunits) {
                                issue_unit(tile + (kb + 2) / NKB, (kb + 2) % NKB, (u + 2) & (MF_NST - 1));
                                wait_vm<PPW + (g2 ? 1 : 0)>();
                            } else {
                                wait_vm<0>();
                            }
                        } else {
                            wait_vm<0>();
                        }
                        stamp(3);
                        __builtin_amdgcn_s_barrier();
                        if (wave < 4 && u + 2 < nunits) issue_unit(tile + (kb + 2) / NKB, (kb + 2) % NKB, (u + 2) & (MF_NST - 1));
                        stamp(4);
                        return;
                    }
                    if (MF_LEAD != 10 && u + 2 < nunits) {
                        if (wave == 0) wait_vm<PPW + (g2 ? (L2 ? 4 : 2) : 0)>(); else wait_vm<PPW + (g2 ? 1 : 0)>();
                    } else {
                        wait_vm<0>();
                    }
                    stamp(3);
                    __builtin_amdgcn_s_barrier();
                    // (Alternating the SIMD's two waves with s_setprio after this barrier does make them
                    // take turns, but a wave alone on the pipe is LDS-latency-bound with one group of
                    // read-ahead: measured 1 % slower.)
                    if constexpr (MF_LEAD == 10) { if (u + 2 < nunits) issue_unit(tile + (kb + 2) / NKB, (kb + 2) % NKB, (u + 2) & (MF_NST - 1)); }
                    stamp(4);
                };
                if (active) chain16(u & (MF_NST - 1), IC<kb>{}, mid); else mid();
                stamp(5);
                if constexpr (MF_LEAD == 25) { if (u + 3 < nunits) issue_unit(tile + (kb + 3) / NKB, (kb + 3) % NKB, (u + 3) & (MF_NST - 1)); }
                stamp(6);
                if constexpr (kb == NKB - 1) {
                    if (active) {
                        if constexpr (SEED) seed_tile(); else filter_tile(tile);
                        if constexpr (L2) { if (tile + 1 < ntiles) init_acc(tile + 1); }
                    }
                }
                stamp(2);
                ++u;
            });
        }
    } else {
        for (int tile = 0; tile < ntiles; ++tile) {
            static_for<NKB>([&](auto KBc) __attribute__((always_inline)) {
                constexpr int kb = decltype(KBc)::value;
                // PD = 3 (HBM-bound): unit u + 3 is issued here, into the buffer everyone left at the
                // last barrier (unit u - 1's).  PD = 2 (MFMA-bound): unit u + 2 is issued at the BOTTOM
                // of the iteration instead, behind the unit's MFMAs; its buffer held unit u - 2.
                const bool more = u + PD < nunits;
                stamp(0);
                if constexpr (PD == 3) { if (more) issue_unit(tile + (kb + PD) / NKB, (kb + PD) % NKB, (u + PD) & (MF_NST - 1)); }
                if constexpr (kb == NKB - 1) { if (active) refresh_thr(tile); }
                if (active) { if constexpr (M16) chain16(u & (MF_NST - 1), IC<kb>{}, [] {}); else chain(u & (MF_NST - 1), IC<kb>{}, IC<0>{}, IC<8>{}); }
                stamp(1);
                if constexpr (PD != 3) { if (more) issue_unit(tile + (kb + PD) / NKB, (kb + PD) % NKB, (u + PD) & (MF_NST - 1)); }
                // The filter needs nothing from the other waves, so it runs BEFORE the barrier: the waves
                // that won the matrix pipe (the older four, in practice) filter while their SIMD
                // partners still have MFMAs in flight, instead of everyone filtering after the barrier
                // with the pipe idle.
                stamp(6);
                if constexpr (kb == NKB - 1) { if (active) filter_tile(tile); }
                stamp(2);
    #ifdef MF_TRACE
                { const int rem = nunits - 2 - u; wait_units<PPW>(rem < 0 ? 0 : (rem < MF_PD - 1 ? rem : MF_PD - 1)); stamp(3); __builtin_amdgcn_s_barrier(); stamp(4); }
    #else
                end_of_unit(u);
    #endif
                ++u;
            });
            if constexpr (L2) { if (active && tile + 1 < ntiles) init_acc(tile + 1); }
    #ifdef MF_TRACE
            --u; stamp(5); ++u;
    #endif
        }
    }
#ifdef MF_TRACE
    if (tslot >= 0) {
        // start-up trace (MF_TRACE_U0 = 0, < 63 units per slice): the last unit slot carries kernel entry,
        // loop entry and loop exit
        if (lane == 0) { tr[63 * 8 + 0] = t_entry; tr[63 * 8 + 1] = t_loop; tr[63 * 8 + 2] = (uint32_t)__builtin_amdgcn_s_memtime(); }
        __builtin_amdgcn_s_waitcnt(0);
        for (int i = lane; i < 512; i += 64) (&mf_trace_buf[tslot][wave][0][0])[i] = tr[i];
    }
#endif

    if constexpr (SEED) {
        if (threadIdx.x == 0 && nqt <= 64) __hip_atomic_store(prog + qt, ~0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (active && ntiles > 0) {
#pragma unroll
            for (int cb = 0; cb < QB; ++cb)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    __hip_atomic_fetch_max(gmax + (qbase + 16 * cb + l15) * MF_SEED_CLASSES + 16 * (slice & 3) + 4 * g4 + j, mi_ord_f32(gm[cb][j]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    } else if constexpr (M16) {
        if (threadIdx.x == 0 && nqt <= 64) __hip_atomic_store(prog + qt, ~0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (g4 == 0) {
#pragma unroll
            for (int cb = 0; cb < QB; ++cb) {
                cnts[(int64_t)slice * nq_pad + qbase + 16 * cb + l15] = cnt2[cb];
                thrs[(int64_t)slice * nq_pad + qbase + 16 * cb + l15] = thr2[cb];
            }
        }
    } else if (h == 0) {
        cnts[(int64_t)slice * nq_pad + qglob] = cnt;
        thrs[(int64_t)slice * nq_pad + qglob] = thr;
    }
}

// E: bound on |v~ - c| where v~ is the phase-1 score (MFMA accumulation, possibly of rounded
// shadow rows) and c the canonical one in the same units (IP: dot; L2: (qq - dist) / 2).
//   accumulation: both are fp32 sums of the same terms in different orders, |diff| <= eps_acc * sum|terms|,
//                 sum|terms| <= |q||x| (+ xx/2 + qq/2 for L2, which also covers the rounding of the distance formula)
//   rounding    : |q.x - q~.x~| <= eps_round * |q||x|  (bf16 shadow of an F32 index; 0 for a BF16 index)
//   underflow   : d * 2^-125 absolute (flushed denormal products / partial sums)
__device__ __forceinline__ float mf_err_bound(int metric, float qqv, float mx, float eps_acc, float eps_round)
{
    const float qx = sqrtf(qqv) * sqrtf(mx);              // not sqrtf(qqv * mx): the product leaves the fp32 range first
    float E = (eps_acc + eps_round) * qx;
    if (metric == MI_METRIC_L2) E += eps_acc * 0.5f * (qqv + mx);
    // absolute floor, d * 2^-125 (eps_acc = 4 d 2^-24): products and partial sums below the normal range may be
    // flushed by the MFMA while the canonical chain keeps them as denormals -- a relative bound alone goes to zero
    // with the norms and would certify a list the scan never ordered
    E += eps_acc * 9.8607613e-32f;
    return E;
}

// ---------------------------------------------------------------- phase 2
__device__ __forceinline__ int block_sum_256(int v, int* red)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    __syncthreads();                       // red may still be read from the previous call
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

template <typename RT>
__device__ __forceinline__ float canon_dot(const RT* __restrict__ xr, const float* __restrict__ qrow, int d_pad)
{
    float acc = 0.0f;
    if constexpr (sizeof(RT) == 2) {
        for (int c = 0; c < d_pad; c += 8) {
            const uint4 v = *reinterpret_cast<const uint4*>(xr + c);
            const float4 q0 = *reinterpret_cast<const float4*>(qrow + c);
            const float4 q1 = *reinterpret_cast<const float4*>(qrow + c + 4);
            acc = __builtin_fmaf(q0.x, __builtin_bit_cast(float, v.x << 16), acc);
            acc = __builtin_fmaf(q0.y, __builtin_bit_cast(float, v.x & 0xFFFF0000u), acc);
            acc = __builtin_fmaf(q0.z, __builtin_bit_cast(float, v.y << 16), acc);
            acc = __builtin_fmaf(q0.w, __builtin_bit_cast(float, v.y & 0xFFFF0000u), acc);
            acc = __builtin_fmaf(q1.x, __builtin_bit_cast(float, v.z << 16), acc);
            acc = __builtin_fmaf(q1.y, __builtin_bit_cast(float, v.z & 0xFFFF0000u), acc);
            acc = __builtin_fmaf(q1.z, __builtin_bit_cast(float, v.w << 16), acc);
            acc = __builtin_fmaf(q1.w, __builtin_bit_cast(float, v.w & 0xFFFF0000u), acc);
        }
    } else {
        for (int c = 0; c < d_pad; c += 4) {
            const float4 v = *reinterpret_cast<const float4*>(xr + c);
            const float4 q0 = *reinterpret_cast<const float4*>(qrow + c);
            acc = __builtin_fmaf(q0.x, v.x, acc);
            acc = __builtin_fmaf(q0.y, v.y, acc);
            acc = __builtin_fmaf(q0.z, v.z, acc);
            acc = __builtin_fmaf(q0.w, v.w, acc);
        }
    }
    return acc;
}

// grid (nq, ngroups).  rows: the CANONICAL rows (bf16 index: the bf16 rows; F32 index: the fp32
// rows), pitch d_pad elements.  Output: the group's best k at out[q * out_stride + g * k + j].
template <typename RT>
__global__ __launch_bounds__(256) void mfma_finish_kernel(
    int metric, const RT* __restrict__ rows, const float* __restrict__ xx, const float* __restrict__ max_xx,
    int d_pad, const float* __restrict__ q_canon, const float* __restrict__ qq, int nq_pad, int k,
    int64_t id_base, int ns, int spg, int cap, float eps_acc, float eps_round,
    const uint64_t* __restrict__ lists, const int32_t* __restrict__ cnts,
    float* __restrict__ out_s, int64_t* __restrict__ out_i, int64_t out_stride, int32_t* __restrict__ over_flags)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint64_t* ent = reinterpret_cast<uint64_t*>(smem);                              // [MF_PH2_MAX]
    uint64_t* keys = ent + MF_PH2_MAX;                                              // [MF_SORT_MAX]
    float* qrow = reinterpret_cast<float*>(keys + MF_SORT_MAX);                     // [d_pad]
    int* offs = reinterpret_cast<int*>(qrow + d_pad);                               // [spg + 1]
    __shared__ int red[4];
    __shared__ int s_nsurv;

    const int t = threadIdx.x;
    const int64_t q = blockIdx.x;
    const int g = blockIdx.y;
    const int s0 = g * spg;
    const int s1 = (s0 + spg < ns) ? s0 + spg : ns;
    const int nsl = s1 - s0;

    // list lengths of the group's slices: loaded by as many threads (one strided word each -- read one after the other by
    // a single thread they were half of this kernel's time at 120 slices), then summed from LDS
    for (int s = t; s < nsl; s += 256) offs[s + 1] = cnts[(int64_t)(s0 + s) * nq_pad + q];
    for (int c = t; c < d_pad; c += 256) qrow[c] = q_canon[q * d_pad + c];
    __syncthreads();
    if (t == 0) {
        int o = 0;
        offs[0] = 0;
        for (int s = 1; s <= nsl; ++s) { o += offs[s]; offs[s] = o; }
        s_nsurv = 0;
    }
    __syncthreads();
    const int total = offs[nsl];                       // <= spg * cap <= MF_PH2_MAX
    const float qqv = qq[q];

    // ---- gather the group's candidate entries (MFMA score word << 32 | row)
    for (int e = t; e < total; e += 256) {
        int lo = 0, hi = nsl;                          // slice with offs[s] <= e < offs[s+1]
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (offs[mid] <= e) lo = mid; else hi = mid; }
        ent[e] = lists[((int64_t)(s0 + lo) * nq_pad + q) * cap + (e - offs[lo])];
    }
    __syncthreads();

    // ---- T_k = k-th largest MFMA score word (0 when total <= k: keep everything)
    uint32_t T = 0;
    if (total > k) {
        for (int bit = 31; bit >= 0; --bit) {
            const uint32_t cand = T | (1u << bit);
            int c = 0;
            for (int e = t; e < total; e += 256) c += ((uint32_t)(ent[e] >> 32) >= cand) ? 1 : 0;
            if (block_sum_256(c, red) >= k) T = cand;
        }
    }
    // survivors: v~ >= T_k - 2E  (every other candidate has k strictly better ones in this group)
    uint32_t Tlow = 0;
    if (T != 0) {
        const float E = mf_err_bound(metric, qqv, *max_xx, eps_acc, eps_round);
        const float lowf = mi_unord_f32(T) - 2.0f * E;
        Tlow = (lowf == lowf) ? mi_ord_f32(lowf) : 0u;
    }
    for (int e = t; e < total; e += 256) {
        const uint64_t en = ent[e];
        if ((uint32_t)(en >> 32) >= Tlow) {
            const int pos = atomicAdd(&s_nsurv, 1);
            if (pos < MF_SORT_MAX) keys[pos] = en;
        }
    }
    __syncthreads();
    int nsurv = s_nsurv;
    if (nsurv > MF_SORT_MAX) {                           // too many near-ties: hand the query to the exact path
        if (t == 0) over_flags[q] = 1;
        nsurv = MF_SORT_MAX;
    }

    // ---- canonical re-score of the survivors
    for (int e = t; e < nsurv; e += 256) {
        const uint32_t row = (uint32_t)keys[e];
        const float dot = canon_dot<RT>(rows + (int64_t)row * d_pad, qrow, d_pad);
        const uint32_t rk = (metric == MI_METRIC_L2) ? mi_rank_l2(dot, qqv, xx[row]) : mi_rank_ip(dot);
        keys[e] = ((uint64_t)rk << 32) | (uint32_t)(~row);
    }
    const int P = (nsurv > 1) ? (1 << (32 - __builtin_clz(nsurv - 1))) : 1;
    for (int e = nsurv + t; e < P; e += 256) keys[e] = 0;
    // sort descending (bitonic, LDS)
    for (int size = 2; size <= P; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            __syncthreads();
            for (int i = t; i < (P >> 1); i += 256) {
                const int lo = 2 * i - (i & (stride - 1));
                const int hi = lo + stride;
                const bool desc = (lo & size) == 0;
                const uint64_t a = keys[lo], b2 = keys[hi];
                if ((a < b2) == desc) { keys[lo] = b2; keys[hi] = a; }
            }
        }
    }
    __syncthreads();

    const int nsel = (nsurv < k) ? nsurv : k;
    float* os = out_s + q * out_stride + (int64_t)g * k;
    int64_t* oi = out_i + q * out_stride + (int64_t)g * k;
    for (int jj = t; jj < k; jj += 256) {
        if (jj < nsel) {
            const uint64_t key = keys[jj];
            os[jj] = mi_score_from_rank(metric, (uint32_t)(key >> 32));
            oi[jj] = id_base + (int64_t)(uint32_t)(~(uint32_t)key);
        } else {
            os[jj] = mi_pad_score(metric);
            oi[jj] = -1;
        }
    }
}

// ---------------------------------------------------------------- phase 3: certificate
// One wave per query.  out_s/out_i: the final best-first list (stride k).
__global__ __launch_bounds__(256) void mfma_certify_kernel(
    int metric, int64_t nq, int nq_pad, int k, int ns, float eps_acc, float eps_round,
    const float* __restrict__ qq, const float* __restrict__ max_xx,
    const int32_t* __restrict__ cnts, const float* __restrict__ thrs,
    const float* __restrict__ out_s, const int64_t* __restrict__ out_i,
    const int32_t* __restrict__ over_flags, int32_t* __restrict__ fail_list, int32_t* __restrict__ fail_count)
{
    const int lane = threadIdx.x & 63;
    const int64_t q = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= nq) return;
    float tmax = -__builtin_inff();
    int total = 0;
    for (int s = lane; s < ns; s += 64) {
        tmax = __builtin_fmaxf(tmax, thrs[(int64_t)s * nq_pad + q]);
        total += cnts[(int64_t)s * nq_pad + q];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        tmax = __builtin_fmaxf(tmax, __shfl_xor(tmax, off, 64));
        total += __shfl_xor(total, off, 64);
    }
    if (lane != 0) return;
    bool ok;
    if (over_flags[q] != 0 || total < k || out_i[q * k + k - 1] < 0) {
        ok = false;                                          // n >= MF_MIN_ROWS > k: rows went missing (NaN scores, overflow)
    } else if (tmax == -__builtin_inff()) {
        ok = true;                                           // nothing was ever filtered out in phase 1
    } else {
        const float qqv = qq[q];
        const float ck = out_s[q * k + k - 1];
        // x1.25: head-room for the fp32 evaluation of the bound itself
        const float E = 1.25f * mf_err_bound(metric, qqv, *max_xx, eps_acc, eps_round);
        if (metric == MI_METRIC_L2) {
            // filtered rows have (dot~ - xx/2) <= tmax  =>  canonical dist >= qq - 2 (tmax + E)
            ok = ck < qqv - 2.0f * (tmax + E);
        } else {
            ok = ck > tmax + E;
        }
        if (!(ck == ck)) ok = false;
    }
    if (!ok) {
        const int pos = atomicAdd(fail_count, 1);
        fail_list[pos] = (int32_t)q;
    }
}

// ---------------------------------------------------------------- host side
static int pick_kp(int k)
{
    const int want = k + (k / 8 > 4 ? k / 8 : 4);
    if (want <= 16) return 16;
    if (want <= 32) return 32;
    if (want <= 128) return 128;
    return 0;
}
static int pick_nkb(int d_pad)
{
    if (d_pad == 128) return 1;
    if (d_pad == 384) return 3;
    if (d_pad == 768) return 6;
    return 0;
}

bool mi_mfma_supported(int d_pad, int k) { return pick_nkb(d_pad) != 0 && pick_kp(k) != 0; }

// Row pitch (elements) of a bf16 matrix the scan can stream: 128 / 384 / 768, or 0.
int mi_mfma_pitch(int d)
{
    const int p = mi_pad_dim(d);
    if (p <= 128) return 128;
    if (p <= 384) return 384;
    if (p <= 768) return 768;
    return 0;
}

int mi_mfma_make_plan(int64_t n, int d_pad, int64_t nq, int k, mi_mfma_plan* plan)
{
    const int nkb = pick_nkb(d_pad), kp = pick_kp(k);
    if (!nkb || !kp || n < MF_MIN_ROWS || n >= ((int64_t)1 << 32) || k > MF_SORT_MAX / 2) return MI_ERR_UNSUPPORTED;
    // Up to 128 queries (four active waves, one per SIMD) the scan is HBM-bound: deep prefetch, 32x32x16.  Beyond,
    // both waves of every SIMD carry MFMAs and the variant with the mid-unit barrier wins, also where the slice is
    // still read by one or two query tiles only (100M x 384, one device: Q = 256 20.7 -> 18.9 ms, Q = 512
    // 36.5 -> 33.3 ms, Q = 128 13.9 vs 13.85 ms; profiles/r02_midq_variants.jsonl)
    plan->pd = (nq <= 128) ? 3 : 2;
    if (const char* e = getenv("MI_MFMA_PD")) { const int v = atoi(e); if (v == 2 || v == 3) plan->pd = v; }
    // MFMA shape: 16x16x32 in the MFMA-bound variant (+8.6 % at 100M x 384 on one device: same
    // cycles, higher sustained clock); MI_MFMA_M16=0 selects 32x32x16 for A/B runs
    plan->m16 = (plan->pd == 2) ? 1 : 0;
    if (const char* e = getenv("MI_MFMA_M16")) plan->m16 = (atoi(e) != 0 && plan->pd == 2) ? 1 : 0;
    const int qtile = mf_qtile(nkb, plan->m16 != 0);              // d = 768: the two variants tile the queries differently
    static_assert(mf_qtile(3, false) == mf_qtile(3, true) && mf_qtile(1, false) == mf_qtile(1, true), "query tile");
    const int cap = mf_cap(kp);
    const int64_t nqt = (nq + qtile - 1) / qtile;
    const int spg_max = (MF_PH2_MAX / cap) / 8 * 8;               // slices whose lists fit one finish block
    int gmax = 4096 / k;                                          // merge_kernel takes <= 4096 entries per query
    if (gmax > 16) gmax = 16;
    // >= ~1024 workgroups in the grid (four per CU over the launch), one slice per XCD at least
    int ns = (int)mi_round_up((1024 + nqt - 1) / nqt, 8);
    if (ns > spg_max * gmax) ns = spg_max * gmax;
    // at least ~32 tiles per slice so the running thresholds have time to tighten; with one or two
    // query tiles the scan is a latency problem instead (200,000 rows x 1 query: 48 slices took
    // 293 us on 48 CUs), so there 8 tiles per slice are enough - the lists are rescored exactly anyway
    // (1M x 384 x 1,000 queries with the threshold seed below, warm clocks, 64 / 128 / 192 / 256 slices: scan 0.64 / 0.65 /
    // 0.65 / 0.66 ms, wall 0.78 / 0.78 / 0.85 / 0.85 -- the scan does not care, the finish pays per slice and per group of
    // 128 slices; 60 tiles stop the search at 128 slices there, whole rounds of 256 workgroups with four query tiles:
    // tools/config2_sweep.py)
    const int min_tiles = (nqt <= 2) ? 8 : 60;
    // (never below one workgroup per CU for the sake of longer slices, unless the slices would be shorter than 8 tiles)
    while (ns > 8 && n / ns < (int64_t)min_tiles * MF_TILE_R && ((int64_t)(ns - 8) * nqt >= 256 || n / ns < 8 * MF_TILE_R)) ns -= 8;
    if (ns < 8) ns = 8;
    if (const char* e = getenv("MI_MFMA_NS")) {
        const int v = atoi(e) / 8 * 8;
        if (v >= 8 && v <= spg_max * gmax && n / v >= MF_TILE_R) ns = v;
    }
    const int ngroups = (ns + spg_max - 1) / spg_max;
    plan->kprime = kp;
    plan->cap = cap;
    plan->nslices = ns;
    plan->ngroups = ngroups;
    plan->spg = (ns + ngroups - 1) / ngroups;
    plan->slice_rows = mi_round_up((n + ns - 1) / ns, MF_TILE_R);
    plan->nq_pad = (int)(nqt * qtile);
    // Threshold seed: K' = 16 lists (k <= 15: the seed guarantees 16 rows above it, no more) scored by the 16x16x32
    // variant.  A sixteenth of the corpus up to 65,536 rows, in slices of whole tiles on >= ~256 workgroups; below
    // 8,192 rows it would not tighten anything.  What it buys: a slice no longer spends its first tiles with thresholds
    // that pass every row (1M x 384 x 1,000 queries: scan 1.23 -> see DESIGN.md 4.1), and costs seed_rows / n of the scan.
    plan->seed_rows = 0;
    plan->seed_slices = 0;
    if (kp == 16 && plan->m16 && !(getenv("MI_MFMA_SEED") && atoi(getenv("MI_MFMA_SEED")) == 0)) {
        int ss = (int)mi_round_up((256 + nqt - 1) / nqt, 8);
        if (ss > 64) ss = 64;
        int64_t want = n / 16;
        if (want > 65536) want = 65536;
        if (const char* e = getenv("MI_MFMA_SEED_ROWS")) { const int64_t v = atoll(e); if (v > 0 && v <= n) want = v; }
        while (ss > 8 && want / ss < 2 * MF_TILE_R) ss -= 8;
        const int64_t per = want / ss / MF_TILE_R * MF_TILE_R;         // whole tiles per slice: no row past the sample
        if (per * ss >= 8192) { plan->seed_rows = per * ss; plan->seed_slices = ss; }
    }
    plan->gthr_bytes = ((size_t)plan->nq_pad + (size_t)ns * 64 + (size_t)plan->nq_pad + (size_t)plan->seed_slices * 64 + (size_t)plan->nq_pad * MF_SEED_CLASSES) * sizeof(uint32_t);
    plan->list_bytes = (size_t)ns * plan->nq_pad * cap * sizeof(uint64_t);
    plan->cnt_bytes = (size_t)ns * plan->nq_pad * sizeof(int32_t);
    plan->thr_bytes = (size_t)ns * plan->nq_pad * sizeof(float);
    return MI_OK;
}

template <int NKB, int KP, bool L2, int PD, bool M16>
static void launch_scan(hipStream_t st, const mi_mfma_plan* p, const uint16_t* rows, const float* xx, int64_t n,
                        int d_pad, const uint16_t* q_bf16, int nq, uint64_t* lists, int32_t* cnts, float* thrs, uint32_t* gthr)
{
    const int qtile = mf_qtile(NKB, M16);
    const int nqt = p->nq_pad / qtile;
    const unsigned grid = (unsigned)(nqt * p->nslices);
    mfma_scan_kernel<NKB, KP, L2, PD, M16><<<grid, 64 * mf_waves(NKB, M16), 0, st>>>(rows, xx, n, d_pad * 2, q_bf16, nq, p->nq_pad, nqt,
                                                                       p->slice_rows, lists, cnts, thrs, gthr, nullptr);
}

// One wave per query: gthr[q] = (the 16th largest of the query's MF_SEED_CLASSES class maxima) - 1 ulp.  Sixteen classes -- sixteen
// different rows -- score at or above that word, so it is a lower bound of the query's 16th best score.
__global__ __launch_bounds__(256) void mfma_seed_min_kernel(const uint32_t* __restrict__ gmax, int nq, uint32_t* __restrict__ gthr)
{
    static_assert(MF_SEED_CLASSES <= 64, "one class per lane");
    const int lane = threadIdx.x & 63;
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= nq) return;
    const uint32_t w = lane < MF_SEED_CLASSES ? gmax[(size_t)q * MF_SEED_CLASSES + lane] : 0u;
    uint32_t T = 0;
    for (int bit = 31; bit >= 0; --bit) {
        const uint32_t cand = T | (1u << bit);
        if (__builtin_popcountll(__builtin_amdgcn_ballot_w64(w >= cand)) >= 16) T = cand;
    }
    // (at or below ord(-inf) = 0x007FFFFF: fewer than 16 classes saw a finite score; leave the query unseeded)
    if (lane == 0 && T > 0x00800000u) atomicMax(gthr + q, T - 1u);
}

template <int NKB, bool L2>
static void launch_seed(hipStream_t st, const mi_mfma_plan* p, const uint16_t* rows, const float* xx, int d_pad,
                        const uint16_t* q_bf16, int nq, uint32_t* gthr)
{
    static_assert(mf_qtile(NKB, true) > 0, "");
    const int nqt = p->nq_pad / mf_qtile(NKB, true);
    uint32_t* const seed_gthr = gthr + p->nq_pad + (size_t)p->nslices * 64;      // zero thresholds + the seed launch's progress words
    uint32_t* const gmax = seed_gthr + p->nq_pad + (size_t)p->seed_slices * 64;
    mfma_scan_kernel<NKB, 16, L2, 2, true, true><<<(unsigned)(nqt * p->seed_slices), 64 * mf_waves(NKB, true), 0, st>>>(
        rows, xx, p->seed_rows, d_pad * 2, q_bf16, nq, p->nq_pad, nqt, p->seed_rows / p->seed_slices, nullptr, nullptr, nullptr, seed_gthr, gmax);
    mfma_seed_min_kernel<<<(unsigned)((nq + 3) / 4), 256, 0, st>>>(gmax, nq, gthr);
}

int mi_launch_mfma_seed(hipStream_t st, const mi_mfma_plan* p, int metric, const uint16_t* rows, const float* xx,
                        int d_pad, const uint16_t* q_bf16, int64_t nq, uint32_t* gthr)
{
    if (p->seed_rows <= 0) return MI_OK;
    const int nkb = pick_nkb(d_pad);
    const bool l2 = metric == MI_METRIC_L2;
    if (nkb == 1) { if (l2) launch_seed<1, true>(st, p, rows, xx, d_pad, q_bf16, (int)nq, gthr); else launch_seed<1, false>(st, p, rows, xx, d_pad, q_bf16, (int)nq, gthr); }
    else if (nkb == 3) { if (l2) launch_seed<3, true>(st, p, rows, xx, d_pad, q_bf16, (int)nq, gthr); else launch_seed<3, false>(st, p, rows, xx, d_pad, q_bf16, (int)nq, gthr); }
    else if (nkb == 6) { if (l2) launch_seed<6, true>(st, p, rows, xx, d_pad, q_bf16, (int)nq, gthr); else launch_seed<6, false>(st, p, rows, xx, d_pad, q_bf16, (int)nq, gthr); }
    else { mi_set_error("mfma seed: no kernel for d_pad=%d", d_pad); return MI_ERR_UNSUPPORTED; }
    MI_HIP(hipGetLastError());
    return MI_OK;
}

int mi_launch_mfma_scan(hipStream_t st, const mi_mfma_plan* p, int metric, const uint16_t* rows,
                        const float* xx, int64_t n, int d_pad, const uint16_t* q_bf16, int64_t nq,
                        uint64_t* lists, int32_t* cnts, float* thrs, uint32_t* gthr)
{
    const int nkb = pick_nkb(d_pad);
    const bool l2 = metric == MI_METRIC_L2;
#define MF_CASE(NKB_, KP_) \
    if (nkb == NKB_ && p->kprime == KP_) { \
        if (p->m16) { \
            if (l2) launch_scan<NKB_, KP_, true, 2, true>(st, p, rows, xx, n, d_pad, q_bf16, (int)nq, lists, cnts, thrs, gthr); \
            else launch_scan<NKB_, KP_, false, 2, true>(st, p, rows, xx, n, d_pad, q_bf16, (int)nq, lists, cnts, thrs, gthr); \
        } \
        else if (l2 && p->pd == 3) launch_scan<NKB_, KP_, true, 3, false>(st, p, rows, xx, n, d_pad, q_bf16, (int)nq, lists, cnts, thrs, gthr); \
        else if (l2) launch_scan<NKB_, KP_, true, 2, false>(st, p, rows, xx, n, d_pad, q_bf16, (int)nq, lists, cnts, thrs, gthr); \
        else if (p->pd == 3) launch_scan<NKB_, KP_, false, 3, false>(st, p, rows, xx, n, d_pad, q_bf16, (int)nq, lists, cnts, thrs, gthr); \
        else launch_scan<NKB_, KP_, false, 2, false>(st, p, rows, xx, n, d_pad, q_bf16, (int)nq, lists, cnts, thrs, gthr); \
        MI_HIP(hipGetLastError()); return MI_OK; }
    MF_CASE(1, 16) MF_CASE(1, 32) MF_CASE(1, 128)
    MF_CASE(3, 16) MF_CASE(3, 32) MF_CASE(3, 128)
    MF_CASE(6, 16) MF_CASE(6, 32) MF_CASE(6, 128)
#undef MF_CASE
    mi_set_error("mfma scan: no kernel for d_pad=%d k'=%d", d_pad, p->kprime);
    return MI_ERR_UNSUPPORTED;
}

void mi_mfma_eps(int d_pad_scan, bool rounded_shadow, float* eps_acc, float* eps_round)
{
    *eps_acc = 4.0f * (float)d_pad_scan * 5.9604645e-8f;
    // shadow rows and queries rounded to bf16: |q.x - q~.x~| <= (2^-9 + 2^-9 + 2^-18) sum|q_k x_k|
    *eps_round = rounded_shadow ? 0.00390625f * 1.01f : 0.0f;
}

int mi_launch_mfma_finish(hipStream_t st, const mi_mfma_plan* p, int metric, const void* canon_rows, int canon_dtype,
                          const float* xx, const float* max_xx, int d_pad, float eps_acc, float eps_round,
                          const float* q_canon, const float* qq, int64_t nq, int k, int64_t id_base,
                          const uint64_t* lists, const int32_t* cnts,
                          float* out_s, int64_t* out_i, int64_t out_stride, int32_t* over_flags)
{
    if (nq <= 0) return MI_OK;
    const size_t smem = (size_t)MF_PH2_MAX * 8 + (size_t)MF_SORT_MAX * 8 + (size_t)d_pad * 4 + (size_t)(p->spg + 1) * 4;
    if (smem > 150 * 1024) { mi_set_error("mfma finish: d_pad %d too large", d_pad); return MI_ERR_UNSUPPORTED; }
    // per call: the attribute is per device, and callers arrive from several threads
    MI_HIP(hipFuncSetAttribute(canon_dtype == MI_DTYPE_BF16 ? reinterpret_cast<const void*>(mfma_finish_kernel<uint16_t>)
                                                              : reinterpret_cast<const void*>(mfma_finish_kernel<float>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    const dim3 grid((unsigned)nq, (unsigned)p->ngroups);
    if (canon_dtype == MI_DTYPE_BF16)
        mfma_finish_kernel<uint16_t><<<grid, 256, smem, st>>>(metric, (const uint16_t*)canon_rows, xx, max_xx, d_pad, q_canon, qq,
            p->nq_pad, k, id_base, p->nslices, p->spg, p->cap, eps_acc, eps_round, lists, cnts, out_s, out_i, out_stride, over_flags);
    else
        mfma_finish_kernel<float><<<grid, 256, smem, st>>>(metric, (const float*)canon_rows, xx, max_xx, d_pad, q_canon, qq,
            p->nq_pad, k, id_base, p->nslices, p->spg, p->cap, eps_acc, eps_round, lists, cnts, out_s, out_i, out_stride, over_flags);
    MI_HIP(hipGetLastError());
    return MI_OK;
}

int mi_launch_mfma_certify(hipStream_t st, const mi_mfma_plan* p, int metric, int64_t nq, int k, float eps_acc, float eps_round,
                           const float* qq, const float* max_xx, const int32_t* cnts, const float* thrs,
                           const float* out_s, const int64_t* out_i, const int32_t* over_flags,
                           int32_t* fail_list, int32_t* fail_count)
{
    if (nq <= 0) return MI_OK;
    mfma_certify_kernel<<<(unsigned)((nq + 3) / 4), 256, 0, st>>>(metric, nq, p->nq_pad, k, p->nslices, eps_acc, eps_round, qq, max_xx,
                                                                  cnts, thrs, out_s, out_i, over_flags, fail_list, fail_count);
    MI_HIP(hipGetLastError());
    return MI_OK;
}

#ifdef MF_TRACE
extern "C" int mi_debug_read_trace(void* out)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(mf_trace_buf), sizeof(mf_trace_buf));
}
#endif
