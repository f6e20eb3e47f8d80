// common.h — shared host/device helpers of libmi355rag (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include "../../include/mi355rag.h"

// ---------------------------------------------------------------- errors (host)
void mi_set_error(const char* fmt, ...);

#define MI_HIP(call)                                                                  \
    do {                                                                              \
        hipError_t _e = (call);                                                       \
        if (_e != hipSuccess) {                                                       \
            mi_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(_e),       \
                         __FILE__, __LINE__);                                         \
            return (_e == hipErrorOutOfMemory) ? MI_ERR_OOM : MI_ERR_HIP;             \
        }                                                                             \
    } while (0)

#define MI_TRY(expr)                                                                  \
    do {                                                                              \
        int _s = (expr);                                                              \
        if (_s != MI_OK) return _s;                                                   \
    } while (0)

#define MI_REQUIRE(cond, ...)                                                         \
    do {                                                                              \
        if (!(cond)) {                                                                \
            mi_set_error(__VA_ARGS__);                                                \
            return MI_ERR_INVALID_ARG;                                                \
        }                                                                             \
    } while (0)

// ---------------------------------------------------------------- ordering (device + host)
// Monotone map float -> uint32 (larger float = larger word); NaN handled by the callers.
__host__ __device__ inline uint32_t mi_ord_f32(float s) {
    uint32_t u = __builtin_bit_cast(uint32_t, s);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__host__ __device__ inline float mi_unord_f32(uint32_t o) {
    uint32_t u = (o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o;
    return __builtin_bit_cast(float, u);
}
// Rank word (larger = better) from the canonical dot; identical to oracle/oracle_search.c
// rank_ip / rank_l2.  The library is compiled with -ffp-contract=off: only the explicit
// fmaf below is fused.
__host__ __device__ inline uint32_t mi_rank_ip(float dot) {
    float s = dot;
    if (s != s) s = -__builtin_inff();
    s = s + 0.0f;
    return mi_ord_f32(s);
}
__host__ __device__ inline uint32_t mi_rank_l2(float dot, float qq, float xx) {
    float dist = __builtin_fmaf(-2.0f, dot, qq + xx);
    if (dist != dist) dist = __builtin_inff();
    if (dist < 0.0f) dist = 0.0f;
    dist = dist + 0.0f;
    return ~mi_ord_f32(dist);
}
__host__ __device__ inline float mi_score_from_rank(int metric, uint32_t r) {
    return mi_unord_f32(metric == MI_METRIC_L2 ? ~r : r);
}
__host__ __device__ inline float mi_pad_score(int metric) {
    return metric == MI_METRIC_L2 ? __builtin_inff() : -__builtin_inff();
}

// ---------------------------------------------------------------- bf16 (device)
// fp32 -> bf16 bits, round-to-nearest-even, NaN stays NaN (same as oracle_f32_to_bf16).
__host__ __device__ inline uint16_t mi_f32_to_bf16(float f) {
    uint32_t u = __builtin_bit_cast(uint32_t, f);
    if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (uint16_t)((u >> 16) | 0x0040u);
    return (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}
__host__ __device__ inline float mi_bf16_to_f32(uint16_t h) {
    return __builtin_bit_cast(float, (uint32_t)h << 16);
}

// Selects a device for the current host thread for the lifetime of the object (entry points are
// called from arbitrary threads and set their ctx's device themselves).
struct DeviceGuard {
    explicit DeviceGuard(int dev) { (void)hipGetDevice(&prev); if (prev != dev) (void)hipSetDevice(dev); want = dev; }
    ~DeviceGuard() { if (prev != want && prev >= 0) (void)hipSetDevice(prev); }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
    int prev = -1, want = -1;
};

// The `stream` argument of the *_dev entry points (include/mi355rag.h): NULL = the ctx's own stream,
// MI_STREAM_LEGACY ((void*)1, the value of hipStreamLegacy) = the legacy default ("null") stream, anything
// else a hipStream_t.  A caller whose work sits on the null stream (PyTorch's default stream has handle 0)
// must not pass 0: that selects the ctx stream, which is non-blocking and would not wait for it.
static inline hipStream_t mi_pick_stream(void* s, hipStream_t own)
{
    if (s == nullptr) return own;
    if (s == MI_STREAM_LEGACY) return (hipStream_t) nullptr;
    return (hipStream_t)s;
}

static inline int64_t mi_round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }
static inline int mi_next_pow2(int x) { int p = 1; while (p < x) p <<= 1; return p; }

// Row pitch (elements) of corpus / query matrices: dim padded with zeros to a multiple of
// 128 elements, so every row is a whole number of 256-byte (bf16) k-blocks for the MFMA scan.
static inline int mi_pad_dim(int d) { return (int)mi_round_up(d, 128); }
