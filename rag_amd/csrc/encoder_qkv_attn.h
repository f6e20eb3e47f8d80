// encoder_qkv_attn.h — launchers of encoder_qkv_attn.hip (internal to libmi355rag): the head of a BERT layer, QKV projection
// and attention in one kernel, for sequences of at most 256 tokens and heads of 32.
#pragma once
#include "common.h"

#define MI_QKV_ATTN_MAX_L 256

// Packs one layer's fused QKV weight (bf16 [3 H][H], torch.nn.Linear's [out][in], Wq already scaled) into per-head chunks
// in MFMA operand order; dst holds 3 H H elements.  H a multiple of 128, H / NH == 32.
int mi_qkv_attn_pack(hipStream_t st, const uint16_t* Wqkv, uint16_t* dst, int H, int NH);
// h [T][H] bf16 row-major; bqkv fp32 [3 H]; padded batches ([B][L] tokens, key mask) or packed sequences (seq_start [B + 1]);
// ctx [T][H] row-major or blocked [ceil(T / 128)][H / 32][128][32].  L <= MI_QKV_ATTN_MAX_L, B <= 65,535.
int mi_launch_qkv_attn(hipStream_t st, const uint16_t* h, const uint16_t* wpk, const float* bqkv, const int32_t* mask,
                       const int32_t* seq_start, uint16_t* ctx, int64_t B, int L, int H, int NH, int ctx_blocked);
