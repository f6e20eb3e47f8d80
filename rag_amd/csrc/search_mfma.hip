// search_mfma.hip — the fast flat-search path for a BF16 index on gfx950 (MI355X):
//
//   phase 1  mfma_scan_kernel    bf16 MFMA scoring of a query tile against a streamed row
//                                slice, with a per-query running threshold: only scores above
//                                the slice's current K'-th best leave the accumulators.
//   phase 2  mfma_finish_kernel  per query: re-score every surviving candidate in the canonical
//                                fp32 fmaf order, sort by (score, id), and CERTIFY that no
//                                row that was filtered out can belong to the top-k.
//
// The MFMA accumulation order is not the canonical order, so phase-1 scores v~ are only used
// to discard: every discarded row satisfies v~ <= thr (per slice), hence its canonical score
// is <= max_slices(thr) + E, where E bounds |v~ - canonical| (both are fp32 accumulations of
// the same exact products; E = eps_rel * |q| * max|x|, eps_rel = 4 * d * 2^-24).  If the k-th
// best canonical score among the candidates is strictly above that bound the answer is
// provably the exact top-k; otherwise the query is re-run on the exact path
// (search_exact.hip).  Replaces faiss IndexFlat::search (advanced_rag.py:296).
//
// Data movement (d = 384: NKB = 3 k-blocks of 128 elements):
//   * queries are register-stationary: each wave keeps the B fragments of its 32 queries x d
//     (96 VGPRs at d = 384) for the whole kernel.  d <= 384: 8 waves per workgroup (two per
//     SIMD, 256 registers each, 256 queries per workgroup) so one wave's LDS reads hide under
//     its partner's MFMAs; d = 768: 4 waves, one per SIMD, 512-register budget.
//   * corpus rows stream HBM -> LDS by LDS-DMA (global_load_lds_dwordx4) in units of
//     128 rows x 256 B, double buffered, each 256-B row segment XOR-swizzled by (row & 15) on
//     the SOURCE side so the ds_read_b128 A-fragment reads are bank-conflict free.
//   * workgroups that scan the same row slice carry the same (blockIdx % 8) so they share an
//     XCD's L2: a slice is fetched from HBM once per XCD pass, not once per query tile.
#include "kernels.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

#define MF_TILE_R 128
#define MF_STAGE_BYTES (MF_TILE_R * 256)
#define MF_PH2_MAX 8192
#define MF_MIN_ROWS 32768

__host__ __device__ constexpr int mf_cap(int kp) { return 2 * kp + 32; }
__host__ __device__ constexpr int mf_waves(int nkb) { return nkb <= 3 ? 8 : 4; }   // 32 queries per wave

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

// ---------------------------------------------------------------- phase 1
template <int NKB, int KP>
__global__ __launch_bounds__(64 * mf_waves(NKB)) void mfma_scan_kernel(
    const uint16_t* __restrict__ rows, int64_t n, int pitch_bytes,
    const uint16_t* __restrict__ q_bf16, int nq_pad, int nqt, int64_t slice_rows,
    uint64_t* __restrict__ lists, int32_t* __restrict__ cnts, float* __restrict__ thrs)
{
    constexpr int WAVES = mf_waves(NKB);
    constexpr int PPW = 32 / WAVES;                   // 1-KiB LDS-DMA pieces per wave per unit
    constexpr int CAP = mf_cap(KP);
    constexpr int EPL = (CAP + 63) / 64;
    constexpr int KS = NKB * 8;                       // 16-wide k-steps per row
    __shared__ __attribute__((aligned(1024))) char lds[2 * MF_STAGE_BYTES];

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
    const int64_t qglob = (int64_t)qt * (32 * WAVES) + wave * 32 + col;   // this lane's query

    // ---- B fragments: lane (col, h) holds query qglob, dims [16 s + 8 h, +8)
    bf16x8 bq[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const uint4 v = *reinterpret_cast<const uint4*>(q_bf16 + qglob * d_pad + 16 * s + 8 * h);
        bq[s] = __builtin_bit_cast(bf16x8, v);
    }

    const int64_t row_begin = (int64_t)slice * slice_rows;
    int64_t row_end = row_begin + slice_rows;
    if (row_end > n) row_end = n;
    const int ntiles = (row_begin < row_end) ? (int)((row_end - row_begin + MF_TILE_R - 1) / MF_TILE_R) : 0;

    // ---- LDS-DMA geometry: wave w issues pieces p = PPW w + i, 4 rows x 256 B each.
    // LDS is lane-linear (row 4p + lane/16, slot lane%16); the lane fetches the 16-B chunk that
    // belongs in that slot under the swizzle  slot = chunk ^ (row & 15).
    const int lr = lane >> 4;            // row within the piece
    const int lt = lane & 15;            // LDS slot
    int src_sw[4];                       // byte offset of the chunk; (4p + lr) & 15 depends on p & 3
#pragma unroll
    for (int i = 0; i < 4; ++i) src_sw[i] = (lt ^ ((4 * i + lr) & 15)) << 4;
    const char* rows_b = reinterpret_cast<const char*>(rows);
    const int64_t last_row = n - 1;

    auto issue_unit = [&](int tile, int kb, int buf) {
        const int64_t trow0 = row_begin + (int64_t)tile * MF_TILE_R;
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int p = wave * PPW + i;                 // PPW is 4 or 8: p & 3 == i & 3
            int64_t r = trow0 + 4 * p + lr;
            if (r > last_row) r = last_row;               // never read past the matrix; masked later
            const char* src = rows_b + r * pitch_bytes + kb * 256 + src_sw[i & 3];
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)src,
                (lds_ptr_t)(lds + buf * MF_STAGE_BYTES + p * 1024), 16, 0, 0);
        }
    };

    // ---- per-lane filter state, replicated in lanes (col) and (col + 32)
    float thr = -__builtin_inff();
    int cnt = 0;
    uint64_t* const list = lists + ((int64_t)slice * nq_pad + qglob) * CAP;

    f32x16 acc[4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[m][i] = 0.0f;

    // A-fragment read address: row (32 m + col), chunk (2 s + h) ^ (row & 15)
    const int a_row_off = col * 256;
    const int a_lo4 = lane & 15;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;

    if (ntiles > 0) {
        issue_unit(0, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }

    int u = 0;
    for (int tile = 0; tile < ntiles; ++tile) {
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb, ++u) {
            const int buf = u & 1;
            {   // prefetch the next unit into the other buffer (everyone left it at the last barrier)
                const int ntile = (kb == NKB - 1) ? tile + 1 : tile;
                const int nkb = (kb == NKB - 1) ? 0 : kb + 1;
                if (ntile < ntiles) issue_unit(ntile, nkb, buf ^ 1);
            }
            // A fragments are read one k-step ahead of the MFMAs that consume them.  The reads
            // are inline asm so hipcc cannot sink each one next to its use (it serialises
            // ds_read -> wait -> MFMA otherwise); LDS returns in order, so with the 4 reads of
            // step s+1 just issued, lgkmcnt(4) means the 4 reads of step s have landed.
            const unsigned cur = lds_base + buf * MF_STAGE_BYTES + a_row_off;
            bf16x8 a[2][4];
            lds_read4(a[0], cur + (((0 + h) ^ a_lo4) << 4));
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                if (s < 7) {
                    lds_read4(a[(s + 1) & 1], cur + (((2 * (s + 1) + h) ^ a_lo4) << 4));
                    asm volatile("s_waitcnt lgkmcnt(4)"
                                 : "+v"(a[s & 1][0]), "+v"(a[s & 1][1]), "+v"(a[s & 1][2]), "+v"(a[s & 1][3]));
                } else {
                    asm volatile("s_waitcnt lgkmcnt(0)"
                                 : "+v"(a[s & 1][0]), "+v"(a[s & 1][1]), "+v"(a[s & 1][2]), "+v"(a[s & 1][3]));
                }
#pragma unroll
                for (int m = 0; m < 4; ++m)
                    acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s & 1][m], bq[kb * 8 + s], acc[m], 0, 0, 0);
            }
            // my LDS reads of this buffer are retired and my DMA into the other one has landed
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }

        // ---- filter: anything above the query's running threshold leaves the accumulators
        const int64_t trow0 = row_begin + (int64_t)tile * MF_TILE_R;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const f32x16 c = acc[m];
            float bm = __builtin_fmaxf(c[0], c[1]);
#pragma unroll
            for (int i = 2; i < 16; ++i) bm = __builtin_fmaxf(bm, c[i]);
            if (__builtin_amdgcn_ballot_w64(bm > thr) != 0) {
                // ---------------- rare path: append the passing elements of this 32x32 block
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float v = c[i];
                    const int64_t row = trow0 + 32 * m + (i & 3) + 8 * (i >> 2) + 4 * h;
                    const bool pass = (v > thr) && (row < row_end);
                    const uint64_t mask = __builtin_amdgcn_ballot_w64(pass);
                    if (mask == 0) continue;
                    const int partner = (int)((mask >> (lane ^ 32)) & 1ull);
                    const int pos = cnt + ((h == 1) ? partner : 0);
                    if (pass) list[pos] = ((uint64_t)mi_ord_f32(v) << 32) | (uint32_t)row;
                    cnt += (pass ? 1 : 0) + partner;
                }
                // ---------------- compact the lists that can no longer take a whole block
                uint64_t need = __builtin_amdgcn_ballot_w64(cnt > CAP - 32) & 0xFFFFFFFFull;
                while (need) {
                    const int ql = __builtin_ctzll(need);
                    need &= need - 1;
                    const int cq = __builtin_amdgcn_readlane(cnt, ql);
                    uint64_t* lq = lists + ((int64_t)slice * nq_pad + (qglob - col + ql)) * CAP;
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
                    if (col == ql) { cnt = KP; thr = mi_unord_f32(T); }
                }
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[m][i] = 0.0f;
        }
    }

    if (h == 0) {
        cnts[(int64_t)slice * nq_pad + qglob] = cnt;
        thrs[(int64_t)slice * nq_pad + qglob] = thr;
    }
}

// ---------------------------------------------------------------- phase 2
__global__ __launch_bounds__(256) void mfma_finish_kernel(
    int metric, const uint16_t* __restrict__ rows, const float* __restrict__ xx,
    const float* __restrict__ max_xx, int64_t n, int d_pad, const float* __restrict__ q_canon,
    const float* __restrict__ qq, int nq_pad, int k, int64_t id_base, int ns, int cap, float eps_rel,
    const uint64_t* __restrict__ lists, const int32_t* __restrict__ cnts, const float* __restrict__ thrs,
    float* __restrict__ out_s, int64_t* __restrict__ out_i, int32_t* __restrict__ fail_list,
    int32_t* __restrict__ fail_count)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint64_t* keys = reinterpret_cast<uint64_t*>(smem);                 // [MF_PH2_MAX]
    float* qrow = reinterpret_cast<float*>(smem + MF_PH2_MAX * 8);      // [d_pad]
    int* offs = reinterpret_cast<int*>(qrow + d_pad);                   // [ns + 1]
    __shared__ float s_tmax;

    const int t = threadIdx.x;
    const int64_t q = blockIdx.x;

    if (t == 0) {
        int o = 0;
        float tm = -__builtin_inff();
        for (int s = 0; s < ns; ++s) {
            offs[s] = o;
            o += cnts[(int64_t)s * nq_pad + q];
            tm = __builtin_fmaxf(tm, thrs[(int64_t)s * nq_pad + q]);
        }
        offs[ns] = o;
        s_tmax = tm;
    }
    for (int c = t; c < d_pad; c += 256) qrow[c] = q_canon[q * d_pad + c];
    __syncthreads();
    const int total = offs[ns];
    const float qqv = qq[q];

    // canonical re-score of every candidate
    for (int e = t; e < total; e += 256) {
        int lo = 0, hi = ns;                 // slice with offs[s] <= e < offs[s+1]
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (offs[mid] <= e) lo = mid; else hi = mid; }
        const uint64_t ent = lists[((int64_t)lo * nq_pad + q) * cap + (e - offs[lo])];
        const uint32_t row = (uint32_t)ent;
        const uint16_t* xr = rows + (int64_t)row * d_pad;
        float acc = 0.0f;
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
        const uint32_t rk = (metric == MI_METRIC_L2) ? mi_rank_l2(acc, qqv, xx[row]) : mi_rank_ip(acc);
        keys[e] = ((uint64_t)rk << 32) | (uint32_t)(~row);
    }
    const int P = (total > 1) ? (1 << (32 - __builtin_clz(total - 1))) : 1;
    for (int e = total + t; e < P; e += 256) keys[e] = 0;
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

    const int nsel = (total < k) ? total : k;
    for (int jj = t; jj < k; jj += 256) {
        if (jj < nsel) {
            const uint64_t key = keys[jj];
            out_s[q * k + jj] = mi_score_from_rank(metric, (uint32_t)(key >> 32));
            out_i[q * k + jj] = id_base + (int64_t)(uint32_t)(~(uint32_t)key);
        } else {
            out_s[q * k + jj] = mi_pad_score(metric);
            out_i[q * k + jj] = -1;
        }
    }

    // certificate
    if (t == 0) {
        const float tmax = s_tmax;
        bool ok;
        if (tmax == -__builtin_inff()) {
            ok = true;                                       // nothing was ever filtered out
        } else if (total < k) {
            ok = false;
        } else {
            const float mx = *max_xx;
            const float ck = mi_score_from_rank(metric, (uint32_t)(keys[k - 1] >> 32));
            if (metric == MI_METRIC_L2) {
                // filtered rows have (dot~ - xx/2) <= tmax  =>  true dist >= qq - 2 tmax - slack
                const float mag = qqv + mx + 2.0f * sqrtf(qqv * mx);
                const float bound = qqv - 2.0f * tmax - 4.0f * eps_rel * mag;
                ok = ck < bound;
            } else {
                const float bound = tmax + eps_rel * sqrtf(qqv * mx);
                ok = ck > bound;
            }
            if (!(ck == ck)) ok = false;
        }
        if (!ok) {
            const int pos = atomicAdd(fail_count, 1);
            fail_list[pos] = (int32_t)q;
        }
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

int mi_mfma_make_plan(int64_t n, int d_pad, int64_t nq, int k, mi_mfma_plan* plan)
{
    const int nkb = pick_nkb(d_pad), kp = pick_kp(k);
    if (!nkb || !kp || n < MF_MIN_ROWS || n >= ((int64_t)1 << 32)) return MI_ERR_UNSUPPORTED;
    const int qtile = 32 * mf_waves(nkb);
    const int cap = mf_cap(kp);
    const int64_t nqt = (nq + qtile - 1) / qtile;
    int ns_max = (MF_PH2_MAX / cap) / 8 * 8;
    int ns = (int)mi_round_up((1024 + nqt - 1) / nqt, 8);
    if (ns > ns_max) ns = ns_max;
    // at least ~32 tiles per slice so the running thresholds have time to tighten
    while (ns > 8 && n / ns < 32 * MF_TILE_R) ns -= 8;
    if (ns < 8) ns = 8;
    plan->kprime = kp;
    plan->cap = cap;
    plan->nslices = ns;
    plan->slice_rows = mi_round_up((n + ns - 1) / ns, MF_TILE_R);
    plan->nq_pad = (int)(nqt * qtile);
    plan->list_bytes = (size_t)ns * plan->nq_pad * cap * sizeof(uint64_t);
    plan->cnt_bytes = (size_t)ns * plan->nq_pad * sizeof(int32_t);
    plan->thr_bytes = (size_t)ns * plan->nq_pad * sizeof(float);
    return MI_OK;
}

template <int NKB, int KP>
static void launch_scan(hipStream_t st, const mi_mfma_plan* p, const uint16_t* rows, int64_t n, int d_pad,
                        const uint16_t* q_bf16, uint64_t* lists, int32_t* cnts, float* thrs)
{
    const int qtile = 32 * mf_waves(NKB);
    const int nqt = p->nq_pad / qtile;
    const unsigned grid = (unsigned)(nqt * p->nslices);
    mfma_scan_kernel<NKB, KP><<<grid, 64 * mf_waves(NKB), 0, st>>>(rows, n, d_pad * 2, q_bf16, p->nq_pad, nqt,
                                                     p->slice_rows, lists, cnts, thrs);
}

int mi_launch_mfma_scan(hipStream_t st, const mi_mfma_plan* p, int metric, const uint16_t* rows,
                        const float* xx, int64_t n, int d_pad, const uint16_t* q_bf16, int64_t nq,
                        uint64_t* lists, int32_t* cnts, float* thrs)
{
    (void)xx; (void)nq;
    if (metric != MI_METRIC_IP) { mi_set_error("mfma scan: L2 metric not built yet"); return MI_ERR_UNSUPPORTED; }
    const int nkb = pick_nkb(d_pad);
#define MF_CASE(NKB_, KP_) \
    if (nkb == NKB_ && p->kprime == KP_) { launch_scan<NKB_, KP_>(st, p, rows, n, d_pad, q_bf16, lists, cnts, thrs); MI_HIP(hipGetLastError()); return MI_OK; }
    MF_CASE(1, 16) MF_CASE(1, 32) MF_CASE(1, 128)
    MF_CASE(3, 16) MF_CASE(3, 32) MF_CASE(3, 128)
    MF_CASE(6, 16) MF_CASE(6, 32) MF_CASE(6, 128)
#undef MF_CASE
    mi_set_error("mfma scan: no kernel for d_pad=%d k'=%d", d_pad, p->kprime);
    return MI_ERR_UNSUPPORTED;
}

int mi_launch_mfma_finish(hipStream_t st, const mi_mfma_plan* p, int metric, const uint16_t* rows,
                          const float* xx, const float* max_xx, int64_t n, int d_pad,
                          const float* q_canon, const float* qq, int64_t nq, int k, int64_t id_base,
                          const uint64_t* lists, const int32_t* cnts, const float* thrs,
                          float* out_s, int64_t* out_i, int32_t* fail_list, int32_t* fail_count)
{
    if (nq <= 0) return MI_OK;
    const size_t smem = (size_t)MF_PH2_MAX * 8 + (size_t)d_pad * 4 + (size_t)(p->nslices + 1) * 4;
    static bool attr_set = false;
    if (!attr_set) {
        MI_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(mfma_finish_kernel),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
        attr_set = true;
    }
    const float eps_rel = 4.0f * (float)d_pad * 5.9604645e-8f;
    mfma_finish_kernel<<<(unsigned)nq, 256, smem, st>>>(metric, rows, xx, max_xx, n, d_pad, q_canon, qq,
        p->nq_pad, k, id_base, p->nslices, p->cap, eps_rel, lists, cnts, thrs, out_s, out_i, fail_list, fail_count);
    MI_HIP(hipGetLastError());
    return MI_OK;
}
