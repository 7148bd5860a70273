// gps_loc_embed.hip -- the box-location embedding  y = LayerNorm(x W^T + b)  of the object streams, one launch per
// direction on MI355X (gfx950).
//
// Reference: `loc_layers = nn.Sequential(nn.Linear(dim_loc = 6, hidden), nn.LayerNorm(hidden))` of the object encoder
// (modules/vision/pcd_openvocab_encoder.py:64-66, applied at :177) and of the unified encoder
// (modules/grounding/unified_encoder.py:28-30, :158): a (B * O, 6) x (6, 768) product whose weight gradient is a
// (768 x 5120) x (5120 x 6) GEMM -- the library picks a 37 us kernel for it -- followed by torch's LayerNorm forward /
// backward pair and a bias column sum: ~115 us per site and step in eight launches (profiles/r3/step_attrib_o.txt).
// Here the reduction length is the TINY dimension: every lane keeps its 12 output columns' weights (12 x 6 floats) in
// registers, a row costs 72 FMAs per lane, LayerNorm happens in the same registers, and the backward pass accumulates
// dW, db, dgamma, dbeta per lane across the rows of its wave (no input gradient: the boxes are data).
//   forward   one wave per row (grid-stride), y fp32 + mean / rstd
//   backward  per-workgroup partial sums [k_in + 3][d] (dW columns k, db, dgamma, dbeta) + a second kernel that adds
//             the workgroups' partials in order (deterministic)
// fp32 throughout (the reference's autocast runs the 6-deep product in bf16; this is the tighter of the two).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gps_hip.h"

namespace gps_loc {

constexpr int kBlock = 256;
constexpr int kWaves = kBlock / 64;
constexpr int kMaxK = 8;
constexpr int kIters = 3;          // d = 768 (hidden size of every GPS stream)
constexpr int kGridMax = 256;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// lane-resident slice of W (d, k_in) and b: columns (i * 64 + lane) * 4 + j
struct Slice {
  float w[kIters][4][kMaxK];
  float b[kIters][4];
};
template <int K>
__device__ __forceinline__ void load_slice(Slice &s, const float *__restrict__ W, const float *__restrict__ bias, int lane) {
#pragma unroll
  for (int i = 0; i < kIters; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = (i * 64 + lane) * 4 + j;
#pragma unroll
      for (int k = 0; k < K; ++k) s.w[i][j][k] = W[(size_t)c * K + k];
      s.b[i][j] = bias ? bias[c] : 0.f;
    }
}
// z = x W^T + b for the lane's 12 columns; returns the row sum of its part
template <int K>
__device__ __forceinline__ float project(const Slice &s, const float (&x)[kMaxK], float (&z)[kIters][4]) {
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < kIters; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float a = s.b[i][j];
#pragma unroll
      for (int k = 0; k < K; ++k) a = fmaf(x[k], s.w[i][j][k], a);
      z[i][j] = a;
      sum += a;
    }
  return sum;
}

template <int K>
__global__ __launch_bounds__(kBlock) void fwd_kernel(int n, int d, const float *__restrict__ x, const float *__restrict__ W,
                                                     const float *__restrict__ bias, const float *__restrict__ gamma,
                                                     const float *__restrict__ beta, float eps, float *__restrict__ y,
                                                     float *__restrict__ mean_out, float *__restrict__ rstd_out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  Slice s;
  load_slice<K>(s, W, bias, lane);
  const float inv_d = 1.f / (float)d;
  for (int row = blockIdx.x * kWaves + wave; row < n; row += gridDim.x * kWaves) {
    float xr[kMaxK];
#pragma unroll
    for (int k = 0; k < K; ++k) xr[k] = x[(size_t)row * K + k];
    float z[kIters][4];
    const float mean = wave_sum(project<K>(s, xr, z)) * inv_d;
    float v = 0.f;
#pragma unroll
    for (int i = 0; i < kIters; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) v += (z[i][j] - mean) * (z[i][j] - mean);
    const float rstd = rsqrtf(wave_sum(v) * inv_d + eps);
    if (lane == 0) { mean_out[row] = mean; rstd_out[row] = rstd; }
#pragma unroll
    for (int i = 0; i < kIters; ++i) {
      const int c0 = (i * 64 + lane) * 4;
      const float4 g = *reinterpret_cast<const float4 *>(gamma + c0);
      const float4 bt = *reinterpret_cast<const float4 *>(beta + c0);
      float4 o;
      o.x = (z[i][0] - mean) * rstd * g.x + bt.x;
      o.y = (z[i][1] - mean) * rstd * g.y + bt.y;
      o.z = (z[i][2] - mean) * rstd * g.z + bt.z;
      o.w = (z[i][3] - mean) * rstd * g.w + bt.w;
      *reinterpret_cast<float4 *>(y + (size_t)row * d + c0) = o;
    }
  }
}

// partial sums per workgroup: part[blockIdx.x][set][d], sets 0 .. K - 1 = dW[:, k], K = db, K + 1 = dgamma, K + 2 = dbeta
template <int K>
__global__ __launch_bounds__(kBlock, 1) void bwd_kernel(int n, int d, const float *__restrict__ dy, const float *__restrict__ x,
                                                        const float *__restrict__ W, const float *__restrict__ bias,
                                                        const float *__restrict__ gamma, const float *__restrict__ mean_in,
                                                        const float *__restrict__ rstd_in, float *__restrict__ part) {
  extern __shared__ float red[];      // [kWaves][d]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  Slice s;
  load_slice<K>(s, W, bias, lane);
  float gm[kIters][4];
  float acc[K + 3][kIters][4];
#pragma unroll
  for (int i = 0; i < kIters; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      gm[i][j] = gamma[(i * 64 + lane) * 4 + j];
#pragma unroll
      for (int q = 0; q < K + 3; ++q) acc[q][i][j] = 0.f;
    }
  const float inv_d = 1.f / (float)d;
  for (int row = blockIdx.x * kWaves + wave; row < n; row += gridDim.x * kWaves) {
    float xr[kMaxK];
#pragma unroll
    for (int k = 0; k < K; ++k) xr[k] = x[(size_t)row * K + k];
    float z[kIters][4];
    project<K>(s, xr, z);
    const float mean = mean_in[row], rstd = rstd_in[row];
    float a[kIters][4], zh[kIters][4];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < kIters; ++i) {
      const float4 g = *reinterpret_cast<const float4 *>(dy + (size_t)row * d + (i * 64 + lane) * 4);
      const float gv[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        zh[i][j] = (z[i][j] - mean) * rstd;
        a[i][j] = gv[j] * gm[i][j];
        s1 += a[i][j];
        s2 += a[i][j] * zh[i][j];
        acc[K + 1][i][j] += gv[j] * zh[i][j];
        acc[K + 2][i][j] += gv[j];
      }
    }
    s1 = wave_sum(s1) * inv_d;
    s2 = wave_sum(s2) * inv_d;
#pragma unroll
    for (int i = 0; i < kIters; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float dz = rstd * (a[i][j] - s1 - zh[i][j] * s2);
        acc[K][i][j] += dz;
#pragma unroll
        for (int k = 0; k < K; ++k) acc[k][i][j] = fmaf(dz, xr[k], acc[k][i][j]);
      }
  }
  float *dst = part + (size_t)blockIdx.x * (K + 3) * d;
#pragma unroll
  for (int q = 0; q < K + 3; ++q) {      // cross-wave reduction, one set at a time through the same LDS rows (unrolled: acc stays in registers)
    __syncthreads();
#pragma unroll
    for (int i = 0; i < kIters; ++i)
      *reinterpret_cast<float4 *>(red + wave * d + (i * 64 + lane) * 4) = make_float4(acc[q][i][0], acc[q][i][1], acc[q][i][2], acc[q][i][3]);
    __syncthreads();
    for (int c = threadIdx.x; c < d; c += kBlock) {
      float t = 0.f;
#pragma unroll
      for (int w2 = 0; w2 < kWaves; ++w2) t += red[w2 * d + c];
      dst[(size_t)q * d + c] = t;
    }
  }
}

// out[set][c] = sum over the workgroups of part[g][set][c], in workgroup order
__global__ __launch_bounds__(kBlock) void reduce_kernel(int parts, int sets, int d, const float *__restrict__ part,
                                                        float *__restrict__ out) {
  __shared__ float red[kWaves][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane, set = blockIdx.y;
  float acc = 0.f;
  if (c < d) {
    int g = wave;
    for (; g + 7 * kWaves < parts; g += 8 * kWaves) {       // 8 independent loads in flight, added in workgroup order
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = part[((size_t)(g + u * kWaves) * sets + set) * d + c];
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += v[u];
    }
    for (; g < parts; g += kWaves) acc += part[((size_t)g * sets + set) * d + c];
  }
  red[wave][lane] = acc;
  __syncthreads();
  if (wave == 0 && c < d) out[(size_t)set * d + c] = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
}

inline int grid_rows(int n) {
  int g = (n + kWaves - 1) / kWaves;
  return g > kGridMax ? kGridMax : (g < 1 ? 1 : g);
}

}  // namespace gps_loc

extern "C" {

int gps_loc_embed_partial_rows(int n_rows) { return gps_loc::grid_rows(n_rows); }

int gps_loc_embed_forward(int n_rows, int k_in, int d, const float *x, const float *w, const float *bias, const float *gamma,
                          const float *beta, float eps, float *y, float *mean, float *rstd, gps_stream_t stream) {
  using namespace gps_loc;
  if (n_rows < 0 || k_in < 1 || d < 1) return GPS_ERR_INVALID_ARGUMENT;
  if (d != 256 * kIters || (k_in != 6 && k_in != 3 && k_in != 8)) return GPS_ERR_UNSUPPORTED;
  if (n_rows == 0) return GPS_OK;
  if (!x || !w || !gamma || !beta || !y || !mean || !rstd) return GPS_ERR_INVALID_ARGUMENT;
  if (((uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)y) & 15) return GPS_ERR_UNSUPPORTED;
  const dim3 grid(grid_rows(n_rows) * 2 < (n_rows + kWaves - 1) / kWaves ? grid_rows(n_rows) * 2 : (n_rows + kWaves - 1) / kWaves), block(kBlock);   // <= 512: a wave keeps its weight slice for several rows
  hipStream_t s = (hipStream_t)stream;
  switch (k_in) {
    case 3: hipLaunchKernelGGL((fwd_kernel<3>), grid, block, 0, s, n_rows, d, x, w, bias, gamma, beta, eps, y, mean, rstd); break;
    case 6: hipLaunchKernelGGL((fwd_kernel<6>), grid, block, 0, s, n_rows, d, x, w, bias, gamma, beta, eps, y, mean, rstd); break;
    default: hipLaunchKernelGGL((fwd_kernel<8>), grid, block, 0, s, n_rows, d, x, w, bias, gamma, beta, eps, y, mean, rstd); break;
  }
  return hipGetLastError() == hipSuccess ? GPS_OK : GPS_ERR_LAUNCH;
}

int gps_loc_embed_backward(int n_rows, int k_in, int d, const float *dy, const float *x, const float *w, const float *bias,
                           const float *gamma, const float *mean, const float *rstd, float *partials, float *sums,
                           gps_stream_t stream) {
  using namespace gps_loc;
  if (n_rows < 0 || k_in < 1 || d < 1) return GPS_ERR_INVALID_ARGUMENT;
  if (d != 256 * kIters || (k_in != 6 && k_in != 3 && k_in != 8)) return GPS_ERR_UNSUPPORTED;
  if (!sums) return GPS_ERR_INVALID_ARGUMENT;
  hipStream_t s = (hipStream_t)stream;
  const int sets = k_in + 3;
  if (n_rows == 0) return hipMemsetAsync(sums, 0, (size_t)sets * d * sizeof(float), s) == hipSuccess ? GPS_OK : GPS_ERR_LAUNCH;
  if (!dy || !x || !w || !gamma || !mean || !rstd || !partials) return GPS_ERR_INVALID_ARGUMENT;
  if ((uintptr_t)dy & 15) return GPS_ERR_UNSUPPORTED;
  const int parts = grid_rows(n_rows);
  const size_t lds = sizeof(float) * kWaves * d;
  switch (k_in) {
    case 3: hipLaunchKernelGGL((bwd_kernel<3>), dim3(parts), dim3(kBlock), lds, s, n_rows, d, dy, x, w, bias, gamma, mean, rstd, partials); break;
    case 6: hipLaunchKernelGGL((bwd_kernel<6>), dim3(parts), dim3(kBlock), lds, s, n_rows, d, dy, x, w, bias, gamma, mean, rstd, partials); break;
    default: hipLaunchKernelGGL((bwd_kernel<8>), dim3(parts), dim3(kBlock), lds, s, n_rows, d, dy, x, w, bias, gamma, mean, rstd, partials); break;
  }
  hipLaunchKernelGGL(reduce_kernel, dim3((d + 63) / 64, sets), dim3(kBlock), 0, s, parts, sets, d, partials, sums);
  return hipGetLastError() == hipSuccess ? GPS_OK : GPS_ERR_LAUNCH;
}

}  // extern "C"
