// gps_losses.hip -- row-sparse cross-entropy for the masked-LM head on MI355X (gfx950).
//
// Reference: optim/loss/loss.py:56-61  lm_cls_loss = F.cross_entropy(logits (B,V,L), labels (B,L),
// ignore_index=-1) over txt_lm_cls_logits (B, L, 30522) from modules/heads/pretrain_head.py:22-56.
// Only ~15 % of the real tokens carry a label (dataset_wrapper masks 15 %), i.e. < 10 % of the B*L
// rows; torch's log_softmax nevertheless reads and writes all B*L*V logits in fp32 twice (5.8 ms of
// the 48 ms step in profiles/r1/bench_f_kernel_stats.csv).  Here one workgroup owns one row:
// ignored rows cost one label load (forward) / one zero-fill of the row (backward); labelled rows
// do a single-pass online log-sum-exp straight from the bf16/fp32 logits.  HBM-bound; algorithmic
// bytes: forward = labelled rows x V x sizeof(logit); backward = N x V x sizeof(logit) written +
// labelled rows read.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gps_hip.h"

namespace gps_loss {

constexpr int kBlock = 256;

__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(uint16_t v) { return __uint_as_float((unsigned int)v << 16); }
__device__ __forceinline__ void from_f32(float &d, float v) { d = v; }
__device__ __forceinline__ void from_f32(uint16_t &d, float v) {
  unsigned int u = __float_as_uint(v);
  u += 0x7FFFu + ((u >> 16) & 1u);
  d = (uint16_t)(u >> 16);
}

// (max, sum of exp(x - max)) merge
__device__ __forceinline__ void lse_merge(float &m, float &s, float m2, float s2) {
  const float mx = fmaxf(m, m2);
  if (mx == -INFINITY) { m = mx; s = 0.f; return; }
  s = s * __expf(m - mx) + s2 * __expf(m2 - mx);
  m = mx;
}

template <typename T>
__global__ __launch_bounds__(kBlock) void masked_ce_fwd_kernel(int n_rows, int V, const T *__restrict__ logits,
                                                                long long ld, const int64_t *__restrict__ labels,
                                                                long long ignore_index, float *__restrict__ loss,
                                                                float *__restrict__ lse_out) {
  __shared__ float s_m[kBlock / 64], s_s[kBlock / 64];
  const int row = blockIdx.x;
  const long long label = labels[row];
  if (label == ignore_index || label < 0 || label >= V) {   // whole workgroup exits together
    if (threadIdx.x == 0) { loss[row] = 0.f; lse_out[row] = 0.f; }
    return;
  }
  const T *x = logits + (size_t)row * ld;
  float m = -INFINITY, s = 0.f;
  for (int v = threadIdx.x; v < V; v += kBlock) {
    const float xv = to_f32(x[v]);
    if (xv > m) { s = s * __expf(m - xv) + 1.f; m = xv; }
    else s += __expf(xv - m);
  }
  for (int off = 32; off >= 1; off >>= 1) {
    const float m2 = __shfl_xor(m, off, 64), s2 = __shfl_xor(s, off, 64);
    lse_merge(m, s, m2, s2);
  }
  if ((threadIdx.x & 63) == 0) { s_m[threadIdx.x >> 6] = m; s_s[threadIdx.x >> 6] = s; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < kBlock / 64; ++w) lse_merge(m, s, s_m[w], s_s[w]);
    const float lse = m + __logf(s);
    lse_out[row] = lse;
    loss[row] = lse - to_f32(x[label]);
  }
}

template <typename T>
__global__ __launch_bounds__(kBlock) void masked_ce_bwd_kernel(int n_rows, int V, const T *__restrict__ logits,
                                                                long long ld, const int64_t *__restrict__ labels,
                                                                long long ignore_index, const float *__restrict__ lse,
                                                                const float *__restrict__ grad_rows,
                                                                T *__restrict__ dlogits, long long ldd) {
  const int row = blockIdx.x;
  const long long label = labels[row];
  T *d = dlogits + (size_t)row * ldd;
  const bool valid = !(label == ignore_index || label < 0 || label >= V);
  const float g = valid ? grad_rows[row] : 0.f;
  if (!valid || g == 0.f) {
    T z;
    from_f32(z, 0.f);
    for (int v = threadIdx.x; v < V; v += kBlock) d[v] = z;
    return;
  }
  const T *x = logits + (size_t)row * ld;
  const float l = lse[row];
  for (int v = threadIdx.x; v < V; v += kBlock) {
    float p = __expf(to_f32(x[v]) - l);
    if (v == (int)label) p -= 1.f;
    from_f32(d[v], p * g);
  }
}

}  // namespace gps_loss

extern "C" {

int gps_masked_ce_forward(int n_rows, int vocab, int logits_bf16, const void *logits, long long ld,
                          const long long *labels, long long ignore_index, float *loss_rows, float *lse,
                          gps_stream_t stream) {
  if (n_rows < 0 || vocab < 1 || ld < vocab) return GPS_ERR_INVALID_ARGUMENT;
  if (n_rows == 0) return GPS_OK;
  if (!logits || !labels || !loss_rows || !lse) return GPS_ERR_INVALID_ARGUMENT;
  hipStream_t s = (hipStream_t)stream;
  if (logits_bf16)
    hipLaunchKernelGGL(gps_loss::masked_ce_fwd_kernel<uint16_t>, dim3(n_rows), dim3(gps_loss::kBlock), 0, s, n_rows,
                       vocab, (const uint16_t *)logits, ld, (const int64_t *)labels, ignore_index, loss_rows, lse);
  else
    hipLaunchKernelGGL(gps_loss::masked_ce_fwd_kernel<float>, dim3(n_rows), dim3(gps_loss::kBlock), 0, s, n_rows,
                       vocab, (const float *)logits, ld, (const int64_t *)labels, ignore_index, loss_rows, lse);
  return hipGetLastError() == hipSuccess ? GPS_OK : GPS_ERR_LAUNCH;
}

int gps_masked_ce_backward(int n_rows, int vocab, int logits_bf16, const void *logits, long long ld,
                           const long long *labels, long long ignore_index, const float *lse,
                           const float *grad_rows, void *dlogits, long long ldd, gps_stream_t stream) {
  if (n_rows < 0 || vocab < 1 || ld < vocab || ldd < vocab) return GPS_ERR_INVALID_ARGUMENT;
  if (n_rows == 0) return GPS_OK;
  if (!logits || !labels || !lse || !grad_rows || !dlogits) return GPS_ERR_INVALID_ARGUMENT;
  hipStream_t s = (hipStream_t)stream;
  if (logits_bf16)
    hipLaunchKernelGGL(gps_loss::masked_ce_bwd_kernel<uint16_t>, dim3(n_rows), dim3(gps_loss::kBlock), 0, s, n_rows,
                       vocab, (const uint16_t *)logits, ld, (const int64_t *)labels, ignore_index, lse, grad_rows,
                       (uint16_t *)dlogits, ldd);
  else
    hipLaunchKernelGGL(gps_loss::masked_ce_bwd_kernel<float>, dim3(n_rows), dim3(gps_loss::kBlock), 0, s, n_rows,
                       vocab, (const float *)logits, ld, (const int64_t *)labels, ignore_index, lse, grad_rows,
                       (float *)dlogits, ldd);
  return hipGetLastError() == hipSuccess ? GPS_OK : GPS_ERR_LAUNCH;
}

}  // extern "C"
