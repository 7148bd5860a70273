// PROBE (not part of libgps_hip.so): split-K weight-gradient GEMM  C (M x N, fp32) = A^T B  with
// A = dY (R x M) and B = X (R x N), both bf16 row-major with the REDUCTION index R (tokens) outermost --
// the shape class hipBLASLt serves worst in the GPS step.
//
// v0 structure (kept simple; every idiom is one already validated in gps_attention.hip):
//   grid (M/128, N/128, S); 4 waves, each a 64 x 64 sub-tile = 4 x 4 tiles of v_mfma_f32_16x16x32_bf16;
//   per stage of 64 tokens: 16-byte global loads (register prefetch of the next stage), TRANSPOSING 2-byte
//   LDS stores into At[m][r] / Bt[n][r] (8-token groups XOR-swizzled by (row >> 3) & 7 so that the lanes of
//   one store instruction spread over banks), 16-byte fragment reads, 2 x 16 MFMAs per wave;
//   partial tiles to a (S, M, N) fp32 workspace, summed in split order by a second kernel (deterministic).
// Known ceiling of v0: the transposing stores are LDS-store-issue bound; the follow-up stores
// the tiles as loaded and reads them with ds_read_b64_tr_b16 (tools/probes/tr_b16_probe.hip).
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

constexpr int BM = 128, BN = 128, BR = 64, PITCH = BR + 8;      // LDS row pitch in bf16 elements (144 B)
constexpr int kThreads = 256;

__device__ __forceinline__ bf16x8 as_frag(u32x4 v) { return __builtin_bit_cast(bf16x8, v); }

// element (row, r) of a transposed tile lives at row * PITCH + ((r >> 3) ^ ((row >> 3) & 7)) * 8 + (r & 7)
__device__ __forceinline__ int slot(int row, int r) { return row * PITCH + ((((r >> 3) ^ (row >> 3)) & 7) << 3) + (r & 7); }

__device__ __forceinline__ void store_transposed(uint16_t *tile, const u32x4 (&v)[4], int tid) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = tid + kThreads * i;            // 16-byte chunk id: row r = c / 16, column octet q = c % 16
    const int r = c >> 4, q = c & 15;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      tile[slot(q * 8 + 2 * e, r)] = (uint16_t)(v[i][e] & 0xFFFFu);
      tile[slot(q * 8 + 2 * e + 1, r)] = (uint16_t)(v[i][e] >> 16);
    }
  }
}

__device__ __forceinline__ void load_stage(u32x4 (&v)[4], const uint16_t *src, long long ld, int r0, int r_end, int col0,
                                           int cols, int tid) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = tid + kThreads * i;
    const int r = r0 + (c >> 4), col = col0 + (c & 15) * 8;
    u32x4 z = {0u, 0u, 0u, 0u};
    v[i] = (r < r_end && col < cols) ? *reinterpret_cast<const u32x4 *>(src + (size_t)r * ld + col) : z;
  }
}

extern "C" __global__ __launch_bounds__(kThreads) void wgrad_splitk_kernel(int R, int M, int N, const uint16_t *__restrict__ A,
                                                                          const uint16_t *__restrict__ B,
                                                                          float *__restrict__ partial, int stages_per_split) {
  __shared__ __attribute__((aligned(16))) uint16_t At[BM * PITCH];
  __shared__ __attribute__((aligned(16))) uint16_t Bt[BN * PITCH];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;                      // 2 x 2 waves, 64 x 64 each
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN, split = blockIdx.z;
  const int n_stages = (R + BR - 1) / BR;
  const int s_begin = split * stages_per_split;
  const int s_end = min(n_stages, s_begin + stages_per_split);
  f32x4 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

  u32x4 va[4], vb[4];
  if (s_begin < s_end) {
    load_stage(va, A, M, s_begin * BR, R, m0, M, tid);
    load_stage(vb, B, N, s_begin * BR, R, n0, N, tid);
  }
  const int i16 = lane & 15, g = lane >> 4;
  for (int s = s_begin; s < s_end; ++s) {
    __syncthreads();                                            // previous stage's fragment reads are done
    store_transposed(At, va, tid);
    store_transposed(Bt, vb, tid);
    __syncthreads();
    if (s + 1 < s_end) {                                        // next stage's global loads fly during the MFMAs
      load_stage(va, A, M, (s + 1) * BR, R, m0, M, tid);
      load_stage(vb, B, N, (s + 1) * BR, R, n0, N, tid);
    }
#pragma unroll
    for (int c = 0; c < BR / 32; ++c) {
      bf16x8 fa[4], fb[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int row_a = wm * 64 + t * 16 + i16, row_b = wn * 64 + t * 16 + i16;
        fa[t] = as_frag(*reinterpret_cast<const u32x4 *>(At + slot(row_a, 32 * c + 8 * g)));
        fb[t] = as_frag(*reinterpret_cast<const u32x4 *>(Bt + slot(row_b, 32 * c + 8 * g)));
      }
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[a], fb[b], acc[a][b], 0, 0, 0);
    }
  }
  // D fragment: row (A operand) = 4 * (lane >> 4) + reg, column (B operand) = lane & 15
  float *P = partial + (size_t)split * M * N;
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + wm * 64 + a * 16 + 4 * g + r, n = n0 + wn * 64 + b * 16 + i16;
        if (m < M && n < N) P[(size_t)m * N + n] = acc[a][b][r];
      }
}

extern "C" __global__ __launch_bounds__(256) void wgrad_reduce_kernel(int splits, long long elems, const float *__restrict__ partial,
                                                                     float *__restrict__ out) {
  const long long e = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (e >= elems) return;
  f32x4 s = *reinterpret_cast<const f32x4 *>(partial + e);
  for (int k = 1; k < splits; ++k) s += *reinterpret_cast<const f32x4 *>(partial + (size_t)k * elems + e);
  *reinterpret_cast<f32x4 *>(out + e) = s;
}

// C launcher (ctypes): A (R x M), B (R x N) bf16 row-major; workspace (splits, M, N) fp32; out (M, N) fp32.
extern "C" int wgrad_probe_launch(int R, int M, int N, int splits, const void *A, const void *B, float *workspace, float *out,
                                  void *stream) {
  if ((M & 7) || (N & 7) || ((long long)M * N & 3)) return -2;
  const int n_stages = (R + BR - 1) / BR;
  const int sps = (n_stages + splits - 1) / splits;
  dim3 grid((M + BM - 1) / BM, (N + BN - 1) / BN, splits);
  hipLaunchKernelGGL(wgrad_splitk_kernel, grid, dim3(kThreads), 0, (hipStream_t)stream, R, M, N, (const uint16_t *)A,
                     (const uint16_t *)B, workspace, sps);
  const long long elems = (long long)M * N;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((elems / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, splits,
                     elems, workspace, out);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}
