// gps_embedding.hip -- dense gradient of an embedding table on MI355X (gfx950), deterministic.
//
// Reference: the word-embedding lookup of the language encoder (HF BertEmbeddings behind
// modules/language/bert.py:21-26; 30 522 x 768 table, 3 200 + 19 200 token ids per step) whose backward
// torch runs as sort -> segment bookkeeping -> sum_and_scatter (~60 launches, 1.27 ms/step for the two
// BERT passes, profiles/r1/bench_z_kernel_stats.csv).  Here:
//   memset   out = 0, first[] = +big, count[] = 0
//   mark     one thread per token: atomicMin(first[id], t), atomicAdd(count[id], 1)  (order-independent)
//   sum      one wave per token; only the FIRST occurrence of an id works: it adds the rows of all
//            tokens with that id in ASCENDING token order (ids scanned 64 at a time with a ballot, stop
//            after count[id] matches) and writes the table row once.  No floating-point atomics: the
//            result does not depend on scheduling.
// HBM-bound: n rows of d floats read once, the touched table rows written once, plus the table memset.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gps_hip.h"

namespace gps_emb {

constexpr int kBlock = 256;
constexpr int kMaxChunks = 8;                   // d <= 8 * 256 floats

__global__ __launch_bounds__(kBlock) void mark_kernel(int n, int num_rows, const int64_t *__restrict__ ids,
                                                      long long padding_idx, int32_t *__restrict__ first,
                                                      int32_t *__restrict__ count) {
  const int t = blockIdx.x * kBlock + threadIdx.x;
  if (t >= n) return;
  const long long id = ids[t];
  if (id < 0 || id >= num_rows || id == padding_idx) return;
  atomicMin(first + id, t);
  atomicAdd(count + id, 1);
}

__global__ __launch_bounds__(kBlock) void sum_kernel(int n, int d, int num_rows, const int64_t *__restrict__ ids,
                                                     const float *__restrict__ dy, long long ld,
                                                     long long padding_idx, const int32_t *__restrict__ first,
                                                     const int32_t *__restrict__ count, float *__restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int t = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);          // one wave per token
  if (t >= n) return;
  const long long id = ids[t];
  if (id < 0 || id >= num_rows || id == padding_idx) return;
  if (first[id] != t) return;                                             // a later duplicate: its first occurrence sums it
  const int want = count[id];
  float4 acc[kMaxChunks];
  const int chunks = (d + 255) / 256;
#pragma unroll
  for (int c = 0; c < kMaxChunks; ++c) {
    const int col = c * 256 + lane * 4;
    acc[c] = (c < chunks && col < d) ? *reinterpret_cast<const float4 *>(dy + (size_t)t * ld + col)
                                     : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  int found = 1;
  // ids are scanned kAhead x 64 at a time: the loads of one trip are independent, so an id that occurs
  // across the whole batch ([CLS], [SEP]) costs n / (64 * kAhead) memory round trips, not n / 64
  constexpr int kAhead = 16;
  for (int base = t + 1; base < n && found < want; base += 64 * kAhead) {
    unsigned long long masks[kAhead];
#pragma unroll
    for (int u = 0; u < kAhead; ++u) {
      const int tt = base + u * 64 + lane;
      const bool match = tt < n && ids[tt] == id;
      masks[u] = __ballot(match);
    }
#pragma unroll
    for (int u = 0; u < kAhead; ++u) {
      unsigned long long mask = masks[u];
      while (mask) {                                                      // ascending token order, four rows per trip:
        int rows[4];                                                      // their loads are issued together, the adds
        int nr = 0;                                                       // stay in token order
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          rows[q] = -1;
          if (mask) {
            rows[q] = base + u * 64 + (__ffsll((long long)mask) - 1);
            mask &= mask - 1ull;
            ++nr;
          }
        }
        float4 v[4][kMaxChunks];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int c = 0; c < kMaxChunks; ++c) {
            const int col = c * 256 + lane * 4;
            v[q][c] = (rows[q] >= 0 && c < chunks && col < d) ? *reinterpret_cast<const float4 *>(dy + (size_t)rows[q] * ld + col)
                                                              : make_float4(0.f, 0.f, 0.f, 0.f);
          }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if (rows[q] < 0) continue;
#pragma unroll
          for (int c = 0; c < kMaxChunks; ++c) {
            acc[c].x += v[q][c].x; acc[c].y += v[q][c].y; acc[c].z += v[q][c].z; acc[c].w += v[q][c].w;
          }
        }
        found += nr;
      }
    }
  }
#pragma unroll
  for (int c = 0; c < kMaxChunks; ++c) {
    const int col = c * 256 + lane * 4;
    if (c < chunks && col < d) *reinterpret_cast<float4 *>(out + (size_t)id * d + col) = acc[c];
  }
}

}  // namespace gps_emb

extern "C" int gps_embedding_grad(int n, int d, int num_rows, const int64_t *ids, const float *dy, long long ld,
                                  long long padding_idx, int32_t *scratch, float *out, gps_stream_t stream) {
  if (n < 0 || d < 0 || num_rows < 0 || ld < d) return GPS_ERR_INVALID_ARGUMENT;
  if ((long long)num_rows * d == 0) return GPS_OK;
  if (!out || !scratch || (n > 0 && (!ids || !dy))) return GPS_ERR_INVALID_ARGUMENT;
  if ((d & 3) || (ld & 3) || d > gps_emb::kMaxChunks * 256 || ((uintptr_t)dy & 15u) || ((uintptr_t)out & 15u))
    return GPS_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  int32_t *first = scratch, *count = scratch + num_rows;
  if (hipMemsetAsync(out, 0, (size_t)num_rows * d * sizeof(float), s) != hipSuccess) return GPS_ERR_LAUNCH;
  if (n == 0) return GPS_OK;
  if (hipMemsetAsync(first, 0x7F, (size_t)num_rows * sizeof(int32_t), s) != hipSuccess) return GPS_ERR_LAUNCH;
  if (hipMemsetAsync(count, 0, (size_t)num_rows * sizeof(int32_t), s) != hipSuccess) return GPS_ERR_LAUNCH;
  hipLaunchKernelGGL(gps_emb::mark_kernel, dim3((n + gps_emb::kBlock - 1) / gps_emb::kBlock), dim3(gps_emb::kBlock), 0, s,
                     n, num_rows, ids, padding_idx, first, count);
  const int waves_per_block = gps_emb::kBlock / 64;
  hipLaunchKernelGGL(gps_emb::sum_kernel, dim3((n + waves_per_block - 1) / waves_per_block), dim3(gps_emb::kBlock), 0, s,
                     n, d, num_rows, ids, dy, ld, padding_idx, first, count, out);
  return hipGetLastError() == hipSuccess ? GPS_OK : GPS_ERR_LAUNCH;
}
