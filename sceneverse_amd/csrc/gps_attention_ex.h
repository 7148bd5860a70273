// Internal (not installed): functions shared by gps_attention.hip (bf16 kernels, argument checks) and
// gps_attention_ex.hip (fp32-operand kernels, fp8-product forward kernel).
#pragma once
#include <hip/hip_runtime.h>

#include "gps_hip.h"

namespace gps_attn {
int run_ex(const gps_attn_args *a, bool backward, hipStream_t s);       // gps_attention.hip
int run_f32(const gps_attn_args *a, bool backward, hipStream_t s);      // gps_attention_ex.hip
int run_fp8_forward(const gps_attn_args *a, hipStream_t s);             // gps_attention_ex.hip
int run_spatial_planes(const gps_attn_args *a, bool backward, hipStream_t s);   // gps_attention_sp.hip
int run_plain_blocks(const gps_attn_args *a, bool backward, hipStream_t s);     // gps_attention_fa.hip
int run_plain_resident(const gps_attn_args *a, bool backward, hipStream_t s);   // gps_attention_sp.hip
}  // namespace gps_attn
