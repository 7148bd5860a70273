// gps_embedding.hip -- dense gradient of an embedding table on MI355X (gfx950), deterministic.
//
// Reference: the word-embedding lookup of the language encoder (HF BertEmbeddings behind
// modules/language/bert.py:21-26; 30 522 x 768 table, 3 200 + 19 200 token ids per step) whose backward
// torch runs as sort -> segment bookkeeping -> sum_and_scatter (~60 launches, 1.27 ms/step for the two
// BERT passes, profiles/r1/bench_z_kernel_stats.csv).  Here:
//   memset   out = 0, first[] = +big, count[] = 0
//   mark     one thread per token: atomicMin(first[id], t), atomicAdd(count[id], 1)  (order-independent)
//   sum      one wave per (token, 256-column chunk); only the FIRST occurrence of an id works: it adds the rows of
//            all tokens with that id in ASCENDING token order (ids scanned 64 at a time with a ballot, stop
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

// [r4] one wave per (token, 256-column chunk): a wave keeps ONE float4 accumulator per lane and fetches the rows of up to
// 16 duplicates per trip (16 independent loads in flight).  The first version gave a token's whole row (3 chunks at
// d = 768) to one wave with 4 rows in flight: the wave of [MASK] (~270 duplicates at the bench workload) and of [CLS] /
// [SEP] (128 each) walked 30 - 70 dependent memory round trips while every other wave had long finished -- 157 - 208 us
// for a 19 us memory job.  The order of the additions is unchanged (first occurrence, then ascending token order).
__global__ __launch_bounds__(kBlock) void sum_kernel(int n, int d, int num_rows, const int64_t *__restrict__ ids,
                                                     const float *__restrict__ dy, long long ld,
                                                     long long padding_idx, const int32_t *__restrict__ first,
                                                     const int32_t *__restrict__ count, float *__restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int t = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);          // token
  const int col = blockIdx.y * 256 + lane * 4;                            // this wave's 256-column chunk
  if (t >= n) return;
  const long long id = ids[t];
  if (id < 0 || id >= num_rows || id == padding_idx) return;
  if (first[id] != t) return;                                             // a later duplicate: its first occurrence sums it
  const int want = count[id];
  const bool live = col < d;
  const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 acc = live ? *reinterpret_cast<const float4 *>(dy + (size_t)t * ld + col) : zero;
  int found = 1;
  // ids are scanned kAhead x 64 at a time: the loads of one trip are independent, so an id that occurs
  // across the whole batch ([CLS], [SEP]) costs n / (64 * kAhead) memory round trips, not n / 64
  constexpr int kAhead = 16, kRows = 16;
  for (int base = t + 1; base < n && found < want; base += 64 * kAhead) {
    unsigned long long masks[kAhead];
#pragma unroll
    for (int u = 0; u < kAhead; ++u) {
      const int tt = base + u * 64 + lane;
      const bool match = tt < n && ids[tt] == id;
      masks[u] = __ballot(match);
    }
    // the matching rows of this trip in ascending token order, kRows at a time
    int u = 0;
    unsigned long long mask = masks[0];
    for (;;) {
      int rows[kRows];
      int nr = 0;
#pragma unroll
      for (int q = 0; q < kRows; ++q) {
        rows[q] = -1;
        while (!mask && u + 1 < kAhead) {            // (wave-uniform: masks are ballots)
          ++u;
          mask = u == 1 ? masks[1] : u == 2 ? masks[2] : u == 3 ? masks[3] : u == 4 ? masks[4] : u == 5 ? masks[5] : u == 6 ? masks[6]
               : u == 7 ? masks[7] : u == 8 ? masks[8] : u == 9 ? masks[9] : u == 10 ? masks[10] : u == 11 ? masks[11]
               : u == 12 ? masks[12] : u == 13 ? masks[13] : u == 14 ? masks[14] : masks[15];
        }
        if (mask) {
          rows[q] = base + u * 64 + (__ffsll((long long)mask) - 1);
          mask &= mask - 1ull;
          ++nr;
        }
      }
      if (nr == 0) break;
      float4 v[kRows];
#pragma unroll
      for (int q = 0; q < kRows; ++q)
        v[q] = (rows[q] >= 0 && live) ? *reinterpret_cast<const float4 *>(dy + (size_t)rows[q] * ld + col) : zero;
#pragma unroll
      for (int q = 0; q < kRows; ++q) {
        if (rows[q] < 0) continue;
        acc.x += v[q].x; acc.y += v[q].y; acc.z += v[q].z; acc.w += v[q].w;
      }
      found += nr;
    }
  }
  if (live) *reinterpret_cast<float4 *>(out + (size_t)id * d + col) = acc;
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
  hipLaunchKernelGGL(gps_emb::sum_kernel, dim3((n + waves_per_block - 1) / waves_per_block, (d + 255) / 256), dim3(gps_emb::kBlock),
                     0, s, n, d, num_rows, ids, dy, ld, padding_idx, first, count, out);
  return hipGetLastError() == hipSuccess ? GPS_OK : GPS_ERR_LAUNCH;
}
