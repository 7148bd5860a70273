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


// ---- bf16 rows, 16-byte accesses (row pitch a multiple of 8 elements, 16-byte aligned base) -------------------------
// The scalar forms above walk a 30 522-column row in 120 dependent trips of 2-byte loads (44 / 66 us per launch at the
// bench workload, one or two rounds of ~480 labelled rows); here a lane takes 8 columns per trip (15 trips), one running
// maximum update per trip.  `rows_dev` (optional, device int): rows at or past it are DEAD -- not read, not written
// (the masked-LM head hands the labelled rows first and its extent-aware GEMMs never read the rest).
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

__device__ __forceinline__ void unpack8(const u32x4 q, float (&x)[8]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    x[2 * i] = __uint_as_float(q[i] << 16);
    x[2 * i + 1] = __uint_as_float(q[i] & 0xFFFF0000u);
  }
}

__global__ __launch_bounds__(kBlock) void masked_ce_fwd_vec_kernel(int n_rows, int V, const uint16_t *__restrict__ logits,
                                                                    long long ld, const int64_t *__restrict__ labels,
                                                                    long long ignore_index, const int *__restrict__ rows_dev,
                                                                    float *__restrict__ loss, float *__restrict__ lse_out,
                                                                    float *__restrict__ mean_out,
                                                                    unsigned int *__restrict__ ticket) {
  __shared__ float s_m[kBlock / 64], s_s[kBlock / 64];
  __shared__ int s_last;
  const int row = blockIdx.x;
  const int live_rows = rows_dev ? min(n_rows, *rows_dev) : n_rows;
  const bool dead = row >= live_rows;                           // the per-row outputs of dead rows are still defined (0)
  const long long label = dead ? ignore_index : labels[row];
  // mean over the labelled rows, taken by the LAST LIVE workgroup to arrive (row order, no float atomics): every live
  // workgroup publishes its row loss and takes a ticket; the last one adds the live values (64 lanes, fixed order) and
  // counts labels.  Dead rows take no ticket: 3 200 workgroups incrementing one word cost 30 us of serialised atomics.
  auto finish = [&](float mine) {
    if (!mean_out) { if (threadIdx.x == 0) loss[row] = mine; return; }
    if (threadIdx.x == 0) {
      __hip_atomic_store(loss + row, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const unsigned int t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s_last = (t == (unsigned int)live_rows - 1u);
      if (s_last) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (!s_last || threadIdx.x >= 64) return;
    float tot = 0.f;
    int cnt = 0;
    for (int r = threadIdx.x; r < live_rows; r += 64) {
      tot += __hip_atomic_load(loss + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const long long lb = labels[r];
      cnt += (lb != ignore_index && lb >= 0 && lb < V) ? 1 : 0;
    }
    for (int off = 32; off >= 1; off >>= 1) { tot += __shfl_xor(tot, off, 64); cnt += __shfl_xor(cnt, off, 64); }
    if (threadIdx.x == 0) { mean_out[0] = tot / (float)cnt; mean_out[1] = (float)cnt; }
  };
  if (dead) {
    if (threadIdx.x == 0) {
      loss[row] = 0.f;
      lse_out[row] = 0.f;
      if (mean_out && live_rows == 0 && row == 0) { mean_out[0] = __int_as_float(0x7FC00000); mean_out[1] = 0.f; }   // mean of nothing
    }
    return;
  }
  if (label == ignore_index || label < 0 || label >= V) {
    if (threadIdx.x == 0) lse_out[row] = 0.f;
    finish(0.f);
    return;
  }
  const uint16_t *x = logits + (size_t)row * ld;
  const u32x4 *x8 = reinterpret_cast<const u32x4 *>(x);
  const int groups = (V + 7) >> 3;
  float m = -INFINITY, s = 0.f;
  for (int gidx = threadIdx.x; gidx < groups; gidx += kBlock) {
    float v[8];
    unpack8(x8[gidx], v);
    const int left = V - 8 * gidx;                               // columns of this group inside the row (>= 1)
    float gm = v[0];
#pragma unroll
    for (int i = 1; i < 8; ++i) { if (i < left) gm = fmaxf(gm, v[i]); }
    const float mx = fmaxf(m, gm);
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) { if (i < left) acc += __expf(v[i] - mx); }
    s = (m == -INFINITY ? 0.f : s * __expf(m - mx)) + acc;
    m = mx;
  }
  for (int off = 32; off >= 1; off >>= 1) {
    const float m2 = __shfl_xor(m, off, 64), s2 = __shfl_xor(s, off, 64);
    lse_merge(m, s, m2, s2);
  }
  if ((threadIdx.x & 63) == 0) { s_m[threadIdx.x >> 6] = m; s_s[threadIdx.x >> 6] = s; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < kBlock / 64; ++w) lse_merge(m, s, s_m[w], s_s[w]);
    s_m[0] = m + __logf(s);
  }
  __syncthreads();
  const float lse = s_m[0];
  if (threadIdx.x == 0) lse_out[row] = lse;
  finish(lse - to_f32(x[label]));
}

__global__ __launch_bounds__(kBlock) void masked_ce_bwd_vec_kernel(int n_rows, int V, const uint16_t *__restrict__ logits,
                                                                    long long ld, const int64_t *__restrict__ labels,
                                                                    long long ignore_index, const int *__restrict__ rows_dev,
                                                                    const float *__restrict__ lse,
                                                                    const float *__restrict__ grad_rows,
                                                                    const float *__restrict__ grad_out,
                                                                    const float *__restrict__ count,
                                                                    uint16_t *__restrict__ dlogits, long long ldd) {
  const int row = blockIdx.x;
  if (rows_dev && row >= *rows_dev) return;
  const long long label = labels[row];
  u32x4 *d8 = reinterpret_cast<u32x4 *>(dlogits + (size_t)row * ldd);
  const int groups = (V + 7) >> 3;                               // the pad columns [V, 8 groups) are written too (zeros)
  const bool valid = !(label == ignore_index || label < 0 || label >= V);
  // without per-row factors: upstream gradient of the MEAN / number of labelled rows (the forward pass's mean_out[1])
  const float g = !valid ? 0.f : (grad_rows ? grad_rows[row] : grad_out[0] / count[0]);
  if (!valid || g == 0.f) {
    const u32x4 z = {0u, 0u, 0u, 0u};
    for (int gidx = threadIdx.x; gidx < groups; gidx += kBlock) d8[gidx] = z;
    return;
  }
  const u32x4 *x8 = reinterpret_cast<const u32x4 *>(logits + (size_t)row * ld);
  const float l = lse[row];
  const int lab = (int)label;
  for (int gidx = threadIdx.x; gidx < groups; gidx += kBlock) {
    float v[8];
    unpack8(x8[gidx], v);
    const int c0 = 8 * gidx;
    uint16_t o[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float p = __expf(v[i] - l);
      if (c0 + i == lab) p -= 1.f;
      p = (c0 + i < V) ? p * g : 0.f;
      from_f32(o[i], p);
    }
    u32x4 q;
#pragma unroll
    for (int i = 0; i < 4; ++i) q[i] = (unsigned int)o[2 * i] | ((unsigned int)o[2 * i + 1] << 16);
    d8[gidx] = q;
  }
}

// ---- row plan of the masked-LM head: labelled rows first ------------------------------------------------------------
// perm = the stable permutation of [0, n) that puts the rows whose label is a valid class id first (both classes in
// their original order); labels_out[i] = label of row perm[i] (ignore_index for the others); n_valid = their count.
// ONE workgroup (n is a few thousand token rows): a ballot / prefix count per 1024-row trip.  Replaces
// valid-mask -> sum -> stable argsort (radix sort) -> where -> index_select, 16 launches / 80 us per step.
__global__ __launch_bounds__(1024) void lm_row_plan_kernel(int n, int V, const int64_t *__restrict__ labels,
                                                            long long ignore_index, int64_t *__restrict__ perm,
                                                            int64_t *__restrict__ labels_out, int *__restrict__ n_valid) {
  __shared__ int s_cnt[16], s_total;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  // pass 1: number of valid rows (where the second class starts)
  int mine = 0;
  for (int i = threadIdx.x; i < n; i += 1024) {
    const long long lb = labels[i];
    mine += (lb != ignore_index && lb >= 0 && lb < V) ? 1 : 0;
  }
  for (int off = 32; off >= 1; off >>= 1) mine += __shfl_xor(mine, off, 64);
  if (lane == 0) s_cnt[w] = mine;
  __syncthreads();
  if (threadIdx.x == 0) {
    int t = 0;
    for (int i = 0; i < 16; ++i) t += s_cnt[i];
    s_total = t;
    *n_valid = t;
  }
  __syncthreads();
  const int total = s_total;
  // pass 2: slots, 1024 rows per trip in row order
  int base_valid = 0;                                            // valid rows before this trip
  for (int i0 = 0; i0 < n; i0 += 1024) {
    const int i = i0 + threadIdx.x;
    const long long lb = i < n ? labels[i] : ignore_index;
    const bool ok = i < n && lb != ignore_index && lb >= 0 && lb < V;
    const unsigned long long mask = __ballot(ok);
    const int before = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
    __syncthreads();                                             // s_cnt of the previous trip is consumed
    if (lane == 0) s_cnt[w] = __popcll(mask);
    __syncthreads();
    int wave_base = 0, trip = 0;
    for (int k = 0; k < 16; ++k) { const int c = s_cnt[k]; wave_base += (k < w) ? c : 0; trip += c; }
    if (i < n) {
      const int v_rank = base_valid + wave_base + before;        // valid rows before row i
      const int slot = ok ? v_rank : total + (i - v_rank);
      perm[slot] = i;
      labels_out[slot] = ok ? lb : ignore_index;
    }
    base_valid += trip;
  }
}

}  // namespace gps_loss

static bool vec_ok(const void *p, long long ld) { return (ld % 8) == 0 && (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

extern "C" {

int gps_masked_ce_forward_rows(int n_rows, int vocab, int logits_bf16, const void *logits, long long ld,
                               const long long *labels, long long ignore_index, const int *rows_dev, float *loss_rows,
                               float *lse, float *mean_out, unsigned int *ticket, gps_stream_t stream) {
  if ((mean_out == nullptr) != (ticket == nullptr)) return GPS_ERR_INVALID_ARGUMENT;
  if (n_rows < 0 || vocab < 1 || ld < vocab) return GPS_ERR_INVALID_ARGUMENT;
  if (n_rows == 0) return GPS_OK;
  if (!logits || !labels || !loss_rows || !lse) return GPS_ERR_INVALID_ARGUMENT;
  hipStream_t s = (hipStream_t)stream;
  if (logits_bf16 && vec_ok(logits, ld))
    hipLaunchKernelGGL(gps_loss::masked_ce_fwd_vec_kernel, dim3(n_rows), dim3(gps_loss::kBlock), 0, s, n_rows, vocab,
                       (const uint16_t *)logits, ld, (const int64_t *)labels, ignore_index, rows_dev, loss_rows, lse, mean_out,
                       ticket);
  else if (rows_dev || mean_out)
    return GPS_ERR_UNSUPPORTED;
  else if (logits_bf16)
    hipLaunchKernelGGL(gps_loss::masked_ce_fwd_kernel<uint16_t>, dim3(n_rows), dim3(gps_loss::kBlock), 0, s, n_rows,
                       vocab, (const uint16_t *)logits, ld, (const int64_t *)labels, ignore_index, loss_rows, lse);
  else
    hipLaunchKernelGGL(gps_loss::masked_ce_fwd_kernel<float>, dim3(n_rows), dim3(gps_loss::kBlock), 0, s, n_rows,
                       vocab, (const float *)logits, ld, (const int64_t *)labels, ignore_index, loss_rows, lse);
  return hipGetLastError() == hipSuccess ? GPS_OK : GPS_ERR_LAUNCH;
}

int gps_masked_ce_forward(int n_rows, int vocab, int logits_bf16, const void *logits, long long ld,
                          const long long *labels, long long ignore_index, float *loss_rows, float *lse,
                          gps_stream_t stream) {
  return gps_masked_ce_forward_rows(n_rows, vocab, logits_bf16, logits, ld, labels, ignore_index, nullptr, loss_rows, lse,
                                    nullptr, nullptr, stream);
}

int gps_masked_ce_backward_rows(int n_rows, int vocab, int logits_bf16, const void *logits, long long ld,
                                const long long *labels, long long ignore_index, const int *rows_dev, const float *lse,
                                const float *grad_rows, const float *grad_out, const float *count, void *dlogits,
                                long long ldd, gps_stream_t stream) {
  if (n_rows < 0 || vocab < 1 || ld < vocab || ldd < vocab) return GPS_ERR_INVALID_ARGUMENT;
  if (n_rows == 0) return GPS_OK;
  if (!logits || !labels || !lse || (!grad_rows && (!grad_out || !count)) || !dlogits) return GPS_ERR_INVALID_ARGUMENT;
  hipStream_t s = (hipStream_t)stream;
  // the 16-byte form also writes the pad columns [vocab, 8 ceil(vocab / 8)) of each row: they must exist (ldd covers them)
  if (logits_bf16 && vec_ok(logits, ld) && vec_ok(dlogits, ldd) && ldd >= (vocab + 7) / 8 * 8)
    hipLaunchKernelGGL(gps_loss::masked_ce_bwd_vec_kernel, dim3(n_rows), dim3(gps_loss::kBlock), 0, s, n_rows, vocab,
                       (const uint16_t *)logits, ld, (const int64_t *)labels, ignore_index, rows_dev, lse, grad_rows, grad_out,
                       count, (uint16_t *)dlogits, ldd);
  else if (rows_dev || !grad_rows)
    return GPS_ERR_UNSUPPORTED;
  else if (logits_bf16)
    hipLaunchKernelGGL(gps_loss::masked_ce_bwd_kernel<uint16_t>, dim3(n_rows), dim3(gps_loss::kBlock), 0, s, n_rows,
                       vocab, (const uint16_t *)logits, ld, (const int64_t *)labels, ignore_index, lse, grad_rows,
                       (uint16_t *)dlogits, ldd);
  else
    hipLaunchKernelGGL(gps_loss::masked_ce_bwd_kernel<float>, dim3(n_rows), dim3(gps_loss::kBlock), 0, s, n_rows,
                       vocab, (const float *)logits, ld, (const int64_t *)labels, ignore_index, lse, grad_rows,
                       (float *)dlogits, ldd);
  return hipGetLastError() == hipSuccess ? GPS_OK : GPS_ERR_LAUNCH;
}

int gps_masked_ce_backward(int n_rows, int vocab, int logits_bf16, const void *logits, long long ld,
                           const long long *labels, long long ignore_index, const float *lse,
                           const float *grad_rows, void *dlogits, long long ldd, gps_stream_t stream) {
  return gps_masked_ce_backward_rows(n_rows, vocab, logits_bf16, logits, ld, labels, ignore_index, nullptr, lse, grad_rows,
                                     nullptr, nullptr, dlogits, ldd, stream);
}

int gps_lm_row_plan(int n_rows, int vocab, const long long *labels, long long ignore_index, long long *perm,
                    long long *labels_out, int *n_valid, gps_stream_t stream) {
  if (n_rows < 0 || vocab < 1) return GPS_ERR_INVALID_ARGUMENT;
  if (!n_valid || (n_rows > 0 && (!labels || !perm || !labels_out))) return GPS_ERR_INVALID_ARGUMENT;
  hipLaunchKernelGGL(gps_loss::lm_row_plan_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, n_rows, vocab,
                     (const int64_t *)labels, ignore_index, (int64_t *)perm, (int64_t *)labels_out, n_valid);
  return hipGetLastError() == hipSuccess ? GPS_OK : GPS_ERR_LAUNCH;
}

}  // extern "C"
