// encoder_ffn3.h — launchers of encoder_ffn3.hip (internal to libmi355rag): the fused tail of a BERT layer
// (attention output projection + LayerNorm + FFN + LayerNorm) with the token on the MFMA lane.
#pragma once
#include "common.h"

// Bytes of one layer's packed weight stream, floats of its parameter block (bo g1 be1 b2 g2 be2 | b1).
size_t mi_ffn3_stream_bytes();
size_t mi_ffn3_param_floats();
// Packs one layer (device fp32 matrices in torch.nn.Linear's [out][in] layout) for ffn3_kernel.  hidden 384, intermediate 1536.
int mi_ffn3_pack(hipStream_t st, const float* Wo, const float* W1, const float* W2, void* stream_dst,
                 const float* bo, const float* g1, const float* be1, const float* b1, const float* b2, const float* g2, const float* be2,
                 float* prm_dst);
// ctx blocked [ceil(M / 128)][12][128][32] bf16; hin / hout [M][384] bf16 row-major (may be the same buffer).
int mi_launch_ffn3(hipStream_t st, const uint16_t* ctx, const uint16_t* hin, const void* wstream, const float* prm, float eps,
                   uint16_t* hout, int M);
