// gps_optim.hip -- the optimizer step of the GPS trainer on MI355X (gfx950): gradient-norm clipping + AdamW over
// EVERY parameter tensor in three launches, also refreshing the bf16 (and packed fp32) copies the MFMA GEMMs read.
//
// Reference behaviour restated (trainer/default_trainer.py:18-24 `backward`):
//     accelerator.clip_grad_norm_(model.parameters(), grad_norm)       torch.nn.utils.clip_grad_norm_, L2
//     optimizer.step()                                                 torch.optim.AdamW (optim/optimizer/optim.py:9-14)
// torch runs this as ~50 foreach launches (norms, stack, clamp, mul, the multi-tensor Adam kernels) plus, under
// autocast, one fp32->bf16 cast launch per weight and forward pass.  Here:
//   1. sumsq_kernel        one pass over all gradients: per-workgroup partial sums of squares (fixed order);
//   2. finish_norm_kernel  total norm, clip coefficient min(1, max_norm / (norm + 1e-6)), step counter += 1;
//   3. adamw_kernel        one pass: g * coef -> decoupled weight decay -> moments -> bias-corrected update of the
//                          fp32 master, then the bf16 shadow / fp32 mirror of the parameter if it has one.
// HBM-bound: 16 B read + 12 B written per parameter (+ 2 B shadow); all tensors and chunks come from device tables.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "gps_hip.h"

namespace gps_optim {

constexpr int kThreads = 256;
constexpr int kChunk = 8192;            // elements per workgroup: 32 per thread, eight 16-byte accesses per array

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

__device__ __forceinline__ float block_sum(float v, float *red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float s = 0.f;
#pragma unroll
  for (int w = 0; w < kThreads / 64; ++w) s += red[w];
  __syncthreads();
  return s;
}

__global__ __launch_bounds__(kThreads) void sumsq_kernel(const gps_adamw_tensor *__restrict__ tensors,
                                                          const int2 *__restrict__ chunks, float *__restrict__ partial, int reverse) {
  __shared__ float red[kThreads / 64];
  const int2 c = chunks[reverse ? (int)gridDim.x - 1 - (int)blockIdx.x : (int)blockIdx.x];
  const gps_adamw_tensor t = tensors[c.x];
  const long long begin = (long long)c.y * kChunk;
  const long long end = min(t.numel, begin + kChunk);
  const float *g = reinterpret_cast<const float *>(t.grad);
  float s = 0.f;
  // 16-byte loads only for 16-byte aligned views: gradients that are slices of one flat buffer (DDP bucket views, the
  // split-graph flat gradient) start wherever the preceding tensors end
  if ((t.numel & 3) == 0 && (reinterpret_cast<uintptr_t>(g) & 15) == 0) {
    // (plain loads on purpose: what this pass leaves in the MALL is what adamw_kernel reads first -- with nontemporal loads
    // here adamw_kernel takes 632 instead of 601 us)
    for (long long e = begin + threadIdx.x * 4; e < end; e += kThreads * 4) {
      const f32x4 v = *reinterpret_cast<const f32x4 *>(g + e);
      s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    }
  } else {
    for (long long e = begin + threadIdx.x; e < end; e += kThreads) s += g[e] * g[e];
  }
  s = block_sum(s, red);
  if (threadIdx.x == 0) partial[reverse ? (int)gridDim.x - 1 - (int)blockIdx.x : (int)blockIdx.x] = s;
}

// scal[0] = clip coefficient, scal[1] = total gradient norm; steps[slot] += 1 for every tensor updated by this call
// (torch.optim.AdamW counts steps PER PARAMETER: one that had no gradient in some iterations lags behind)
// (1 024 threads, four loads in flight per thread: the ~15 k partials of the bench model took 58 dependent trips of one
// 256-thread workgroup, 18 us; the sum is fp64, its order fixed)
constexpr int kFinishThreads = 1024;
__global__ __launch_bounds__(kFinishThreads) void finish_norm_kernel(int n_partial, const float *__restrict__ partial, float max_norm,
                                                                      float *__restrict__ scal, int n_tensors,
                                                                      const gps_adamw_tensor *__restrict__ tensors,
                                                                      float *__restrict__ steps) {
  __shared__ double red[kFinishThreads];
  double s4[4] = {0.0, 0.0, 0.0, 0.0};
  int i = threadIdx.x;
  for (; i + 3 * kFinishThreads < n_partial; i += 4 * kFinishThreads) {
    float v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = partial[i + u * kFinishThreads];
#pragma unroll
    for (int u = 0; u < 4; ++u) s4[u] += (double)v[u];
  }
  for (; i < n_partial; i += kFinishThreads) s4[0] += (double)partial[i];
  red[threadIdx.x] = (s4[0] + s4[1]) + (s4[2] + s4[3]);
  __syncthreads();
  for (int o = kFinishThreads / 2; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float norm = (float)sqrt(red[0]);
    float coef = 1.f;
    if (max_norm > 0.f) coef = fminf(max_norm / (norm + 1e-6f), 1.f);
    scal[0] = coef;
    scal[1] = norm;
  }
  for (int t = threadIdx.x; t < n_tensors; t += kFinishThreads) steps[tensors[t].step_slot] += 1.f;
}

__device__ __forceinline__ void adam_one(float &p, float g, float &m, float &v, float coef, float lr, float b1, float b2, float eps,
                                         float wd, float step_size, float inv_sqrt_bc2) {
  g *= coef;
  p *= 1.f - lr * wd;
  m = b1 * m + (1.f - b1) * g;
  v = b2 * v + (1.f - b2) * g * g;
  const float denom = sqrtf(v) * inv_sqrt_bc2 + eps;
  p -= step_size * (m / denom);
}

template <bool NT>
__global__ __launch_bounds__(kThreads) void adamw_kernel(const gps_adamw_tensor *__restrict__ tensors,
                                                          const gps_adamw_group *__restrict__ groups,
                                                          const int2 *__restrict__ chunks, const float *__restrict__ scal,
                                                          const float *__restrict__ steps, int reverse) {
  // reverse: the chunks are walked from the last one down.  sumsq_kernel has just read every gradient front to back, so
  // the MALL holds the TAIL of that stream; a front-to-back walk finds none of it (LRU streaming), a back-to-front walk
  // starts inside it.
  const int2 c = chunks[reverse ? (int)gridDim.x - 1 - (int)blockIdx.x : (int)blockIdx.x];
  const gps_adamw_tensor t = tensors[c.x];
  const gps_adamw_group G = groups[t.group];
  const float coef = scal[0], step = steps[t.step_slot];
  const float lr = G.lr_dev ? *reinterpret_cast<const float *>(G.lr_dev) : G.lr;
  const float bc1 = 1.f - powf(G.beta1, step), bc2 = 1.f - powf(G.beta2, step);
  const float step_size = lr / bc1, inv_sqrt_bc2 = 1.f / sqrtf(bc2);
  const long long begin = (long long)c.y * kChunk;
  const long long end = min(t.numel, begin + kChunk);
  float *p = reinterpret_cast<float *>(t.param), *m = reinterpret_cast<float *>(t.exp_avg), *v = reinterpret_cast<float *>(t.exp_avg_sq);
  const float *g = reinterpret_cast<const float *>(t.grad);
  uint16_t *sh = reinterpret_cast<uint16_t *>(t.shadow_bf16);
  float *mir = reinterpret_cast<float *>(t.mirror_f32);
  const uintptr_t align = reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(m) | reinterpret_cast<uintptr_t>(v) |
                          reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(mir) | (reinterpret_cast<uintptr_t>(sh) << 1);
  if ((t.numel & 3) == 0 && (align & 15) == 0) {
    for (long long e = begin + threadIdx.x * 4; e < end; e += kThreads * 4) {
      f32x4 P, M, V, Gr;
      if (NT) {          // every array is streamed once per step: no reuse to keep in L2 / MALL
        P = __builtin_nontemporal_load(reinterpret_cast<f32x4 *>(p + e));
        M = __builtin_nontemporal_load(reinterpret_cast<f32x4 *>(m + e));
        V = __builtin_nontemporal_load(reinterpret_cast<f32x4 *>(v + e));
        Gr = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(g + e));
      } else {
        P = *reinterpret_cast<f32x4 *>(p + e); M = *reinterpret_cast<f32x4 *>(m + e); V = *reinterpret_cast<f32x4 *>(v + e);
        Gr = *reinterpret_cast<const f32x4 *>(g + e);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float pi = P[i], mi = M[i], vi = V[i];
        adam_one(pi, Gr[i], mi, vi, coef, lr, G.beta1, G.beta2, G.eps, G.weight_decay, step_size, inv_sqrt_bc2);
        P[i] = pi; M[i] = mi; V[i] = vi;
      }
      if (NT) {
        __builtin_nontemporal_store(P, reinterpret_cast<f32x4 *>(p + e));
        __builtin_nontemporal_store(M, reinterpret_cast<f32x4 *>(m + e));
        __builtin_nontemporal_store(V, reinterpret_cast<f32x4 *>(v + e));
      } else {
        *reinterpret_cast<f32x4 *>(p + e) = P;
        *reinterpret_cast<f32x4 *>(m + e) = M;
        *reinterpret_cast<f32x4 *>(v + e) = V;
      }
      if (sh) {
        const bf16x4 h = {(__bf16)P[0], (__bf16)P[1], (__bf16)P[2], (__bf16)P[3]};
        *reinterpret_cast<u32x2 *>(sh + e) = __builtin_bit_cast(u32x2, h);
      }
      if (mir) *reinterpret_cast<f32x4 *>(mir + e) = P;
    }
  } else {
    for (long long e = begin + threadIdx.x; e < end; e += kThreads) {
      float P = p[e], M = m[e], V = v[e];
      adam_one(P, g[e], M, V, coef, lr, G.beta1, G.beta2, G.eps, G.weight_decay, step_size, inv_sqrt_bc2);
      p[e] = P; m[e] = M; v[e] = V;
      if (sh) { const __bf16 h = (__bf16)P; sh[e] = __builtin_bit_cast(uint16_t, h); }
      if (mir) mir[e] = P;
    }
  }
}

}  // namespace gps_optim

extern "C" {

int gps_adamw_chunk_elems(void) { return gps_optim::kChunk; }

int gps_adamw_step(int n_tensors, int n_chunks, const gps_adamw_tensor *tensors, const gps_adamw_group *groups,
                   const int32_t *chunks, float max_grad_norm, float *partial, float *scalars, float *steps,
                   gps_stream_t stream) {
  using namespace gps_optim;
  if (n_chunks < 0 || n_tensors < 0) return GPS_ERR_INVALID_ARGUMENT;
  if (n_chunks == 0 || n_tensors == 0) return GPS_OK;
  if (!tensors || !groups || !chunks || !partial || !scalars || !steps) return GPS_ERR_INVALID_ARGUMENT;
  hipStream_t s = (hipStream_t)stream;
  const int2 *ch = reinterpret_cast<const int2 *>(chunks);
  // GPS_OPT_WALK: 0 = both passes front to back (rounds 4 - 5), 1 = norm pass forward, update pass backward (default:
  // 603 -> 579 us in two same-box A/Bs), 2 = norm pass backward, update pass forward
  static const int walk = [] {
    const char *e = getenv("GPS_OPT_WALK");
    const int v = e ? atoi(e) : 1;
    return v < 0 || v > 2 ? 1 : v;
  }();
  if (max_grad_norm > 0.f)
    hipLaunchKernelGGL(sumsq_kernel, dim3(n_chunks), dim3(kThreads), 0, s, tensors, ch, partial, walk == 2 ? 1 : 0);
  hipLaunchKernelGGL(finish_norm_kernel, dim3(1), dim3(kFinishThreads), 0, s, max_grad_norm > 0.f ? n_chunks : 0, partial, max_grad_norm,
                     scalars, n_tensors, tensors, steps);
  // [r6] nontemporal loads / stores of p, m, v, g (each streamed once per step, 3.6 GB against 256 MB of MALL): 590 -> 555 us
  // in two same-box A/Bs, and the step gains more than that (the bf16 shadows written last stay cached for the forward pass).
  // GPS_ADAMW_NT=0 restores the plain accesses.
  static const bool nontemporal = [] {
    const char *e = getenv("GPS_ADAMW_NT");
    return e ? atoi(e) != 0 : true;
  }();
  const int rev = (max_grad_norm > 0.f && walk == 1) ? 1 : 0;     // (without the norm pass nothing was read before)
  if (nontemporal) hipLaunchKernelGGL(adamw_kernel<true>, dim3(n_chunks), dim3(kThreads), 0, s, tensors, groups, ch, scalars, steps, rev);
  else hipLaunchKernelGGL(adamw_kernel<false>, dim3(n_chunks), dim3(kThreads), 0, s, tensors, groups, ch, scalars, steps, rev);
  return hipGetLastError() == hipSuccess ? GPS_OK : GPS_ERR_LAUNCH;
}

}  // extern "C"
