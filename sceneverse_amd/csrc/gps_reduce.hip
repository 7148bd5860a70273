// gps_reduce.hip -- column sums of a (rows x cols) bf16 matrix in fp32: the bias gradients of the
// transformer Linears (db = sum over tokens of dY) on MI355X (gfx950).
//
// Reference: the bias gradient autograd derives for every nn.Linear of the GPS transformer stacks
// (modules/layers/transformers.py:115-154, 285-316; HF BertLayer behind modules/language/bert.py:21-26):
// 67 reductions per step over dY matrices of up to 19 200 x 3 072 bf16 (2.1 GB per step in total).  torch's
// generic reduce_kernel takes 1.4 ms/step for them (profiles/r1/bench_t_kernel_stats.csv); the HBM floor
// is 0.3-0.4 ms.  HBM-bound, algorithmic bytes = rows * cols * 2 read once.
//
// Stage 1: grid (cols / 256, P row chunks), 256 threads = 8 row lanes x 32 column groups of 8 bf16
//          (one 16-byte load per thread and row, 512 contiguous bytes per row and wave half; eight loads
//          in flight), fp32 accumulation, 8 -> 1 over the row lanes through LDS, partial row written.
// Stage 2: the P partial rows summed in a fixed order (deterministic; no atomics).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gps_hip.h"

namespace gps_red {

constexpr int kBlock = 256;
constexpr int kColsPerBlock = 256;
constexpr int kRowLanes = 8;

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

__device__ __forceinline__ void add8(float (&acc)[8], const u32x4 v) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    acc[2 * i] += __uint_as_float(v[i] << 16);
    acc[2 * i + 1] += __uint_as_float(v[i] & 0xFFFF0000u);
  }
}

__global__ __launch_bounds__(kBlock) void colsum_stage1_kernel(int rows, int cols, const uint16_t *__restrict__ x,
                                                               long long ld, int rows_per_part,
                                                               float *__restrict__ partials) {
  __shared__ float s_acc[kRowLanes][kColsPerBlock + 8];
  const int cg = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int col = blockIdx.x * kColsPerBlock + cg * 8;
  const int r0 = blockIdx.y * rows_per_part;
  const int r1 = min(rows, r0 + rows_per_part);
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (col < cols) {
    const uint16_t *p = x + col;
    int r = r0 + rl;
    for (; r + 7 * kRowLanes < r1; r += 8 * kRowLanes) {          // eight 16-byte loads in flight
      u32x4 v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = *reinterpret_cast<const u32x4 *>(p + (size_t)(r + k * kRowLanes) * ld);
#pragma unroll
      for (int k = 0; k < 8; ++k) add8(acc, v[k]);
    }
    for (; r < r1; r += kRowLanes) add8(acc, *reinterpret_cast<const u32x4 *>(p + (size_t)r * ld));
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) s_acc[rl][cg * 8 + i] = acc[i];
  __syncthreads();
  const int c = blockIdx.x * kColsPerBlock + threadIdx.x;
  if (c < cols) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < kRowLanes; ++k) s += s_acc[k][threadIdx.x];
    partials[(size_t)blockIdx.y * cols + c] = s;
  }
}

// out[c] = sum_p partials[p][c]: 64 columns per workgroup, 16 waves split the partial rows (four loads in
// flight each); the 16 wave sums are added in a fixed order
constexpr int kStage2Waves = 16;
__global__ __launch_bounds__(kStage2Waves * 64) void colsum_stage2_kernel(int parts, int cols,
                                                                          const float *__restrict__ partials,
                                                                          float *__restrict__ out) {
  __shared__ float s[kStage2Waves][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (c < cols) {
    int p = w;
    for (; p + 3 * kStage2Waves < parts; p += 4 * kStage2Waves) {
      a0 += partials[(size_t)p * cols + c];
      a1 += partials[(size_t)(p + kStage2Waves) * cols + c];
      a2 += partials[(size_t)(p + 2 * kStage2Waves) * cols + c];
      a3 += partials[(size_t)(p + 3 * kStage2Waves) * cols + c];
    }
    for (; p < parts; p += kStage2Waves) a0 += partials[(size_t)p * cols + c];
  }
  s[w][lane] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (w == 0 && c < cols) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < kStage2Waves; ++k) t += s[k][lane];
    out[c] = t;
  }
}

__host__ inline int parts_for(int rows, int cols) {
  const int col_blocks = (cols + kColsPerBlock - 1) / kColsPerBlock;
  int parts = 768 / (col_blocks > 0 ? col_blocks : 1);       // ~3 workgroups per CU
  if (parts > 256) parts = 256;
  const int max_parts = (rows + 63) / 64;                     // at least 64 rows per part
  if (parts > max_parts) parts = max_parts;
  return parts < 1 ? 1 : parts;
}

}  // namespace gps_red

extern "C" int gps_colsum_parts(int rows, int cols) {
  if (rows <= 0 || cols <= 0) return 0;
  const int parts = gps_red::parts_for(rows, cols);
  const int rpp = ((rows + parts - 1) / parts + gps_red::kRowLanes - 1) / gps_red::kRowLanes * gps_red::kRowLanes;
  return (rows + rpp - 1) / rpp;
}

extern "C" int gps_colsum_bf16(int rows, int cols, const void *x, long long ld, float *partials, float *out,
                               gps_stream_t stream) {
  if (rows < 0 || cols < 0 || ld < cols) return GPS_ERR_INVALID_ARGUMENT;
  if (cols == 0) return GPS_OK;
  if (!out) return GPS_ERR_INVALID_ARGUMENT;
  hipStream_t s = (hipStream_t)stream;
  if (rows == 0) return hipMemsetAsync(out, 0, (size_t)cols * sizeof(float), s) == hipSuccess ? GPS_OK : GPS_ERR_LAUNCH;
  if (!x || !partials) return GPS_ERR_INVALID_ARGUMENT;
  if ((cols & 7) || (ld & 7) || ((uintptr_t)x & 15u)) return GPS_ERR_UNSUPPORTED;   // 16-byte row pieces
  const int parts0 = gps_red::parts_for(rows, cols);
  const int rpp = ((rows + parts0 - 1) / parts0 + gps_red::kRowLanes - 1) / gps_red::kRowLanes * gps_red::kRowLanes;
  const int parts = (rows + rpp - 1) / rpp;
  const dim3 grid1((cols + gps_red::kColsPerBlock - 1) / gps_red::kColsPerBlock, parts);
  hipLaunchKernelGGL(gps_red::colsum_stage1_kernel, grid1, dim3(gps_red::kBlock), 0, s, rows, cols,
                     (const uint16_t *)x, ld, rpp, partials);
  hipLaunchKernelGGL(gps_red::colsum_stage2_kernel, dim3((cols + 63) / 64), dim3(gps_red::kStage2Waves * 64), 0, s,
                     parts, cols, partials, out);
  return hipGetLastError() == hipSuccess ? GPS_OK : GPS_ERR_LAUNCH;
}
