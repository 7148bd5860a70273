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

// [r4] one wave per (token, 256-column chunk): a wave keeps ONE float4 accumulator per lane.  The first version gave a
// token's whole row (3 chunks at d = 768) to one wave with 4 rows in flight: the wave of [MASK] (~270 duplicates at the
// bench workload) and of [CLS] / [SEP] (128 each) walked 30 - 70 dependent memory round trips while every other wave had
// long finished -- 157 - 208 us for a 19 us memory job.  Now the id scan and the row fetches are decoupled: a scan trip
// looks at kScan x 64 ids (independent loads, L2 resident) and appends the matching token numbers, in ascending order,
// to the wave's list in LDS; rows are fetched kRows at a time (independent loads) as soon as kRows are pending, so a
// token with c duplicates costs n / (64 kScan) + c / kRows round trips instead of one or two per 1024 ids.
// The order of the additions is unchanged (first occurrence, then ascending token order).
constexpr int kScan = 32, kRows = 16;
constexpr int kListCap = kScan * 64 + kRows;

__global__ __launch_bounds__(kBlock) void sum_kernel(int n, int d, int num_rows, const int64_t *__restrict__ ids,
                                                     const float *__restrict__ dy, long long ld,
                                                     long long padding_idx, const int32_t *__restrict__ first,
                                                     const int32_t *__restrict__ count, float *__restrict__ out) {
  __shared__ int list_s[kBlock / 64][kListCap];
  const int lane = threadIdx.x & 63;
  int *list = list_s[threadIdx.x >> 6];
  const int t = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);          // token
  const int col = blockIdx.y * 256 + lane * 4;                            // this wave's 256-column chunk
  if (t >= n) return;
  const long long id = ids[t];
  if (id < 0 || id >= num_rows || id == padding_idx) return;
  if (first[id] != t) return;                                             // a later duplicate: its first occurrence sums it
  int missing = count[id] - 1;                                            // duplicates still to be found (wave-uniform)
  const bool live = col < d;
  const int colc = live ? col : 0;                                        // dead lanes read (and drop) the first columns
  float4 acc = *reinterpret_cast<const float4 *>(dy + (size_t)t * ld + colc);
  const unsigned long long below = (1ull << lane) - 1ull;
  int pending = 0;                                                        // list[0 .. pending) wait for their rows
  for (int base = t + 1; base < n && missing > 0; base += 64 * kScan) {
    unsigned long long masks[kScan];
#pragma unroll
    for (int u = 0; u < kScan; ++u) {
      const int tt = base + u * 64 + lane;
      const long long other = ids[min(tt, n - 1)];                        // unconditional: kScan independent loads
      masks[u] = __ballot(tt < n && other == id);
    }
#pragma unroll
    for (int u = 0; u < kScan; ++u) {
      const unsigned long long m = masks[u];
      if (m) {                                                            // (wave-uniform)
        if ((m >> lane) & 1ull) list[pending + __popcll(m & below)] = base + u * 64 + lane;
        const int c = __popcll(m);
        pending += c;
        missing -= c;
      }
    }
    const bool last = missing <= 0 || base + 64 * kScan >= n;
    int head = 0;
    while (pending - head >= kRows || (last && head < pending)) {
      const int cnt = min(kRows, pending - head);
      float4 v[kRows];
#pragma unroll
      for (int q = 0; q < kRows; ++q) {
        const int r = list[head + min(q, cnt - 1)];                       // same address in every lane: a broadcast read
        v[q] = *reinterpret_cast<const float4 *>(dy + (size_t)r * ld + colc);   // unconditional (q >= cnt re-reads the last row)
      }
#pragma unroll
      for (int q = 0; q < kRows; ++q)
        if (q < cnt) { acc.x += v[q].x; acc.y += v[q].y; acc.z += v[q].z; acc.w += v[q].w; }   // (wave-uniform)
      head += cnt;
    }
    // fewer than kRows left over: to the front of the list, the next trip appends behind them
    const int left = pending - head;
    if (left > 0 && head > 0) {
      const int keep = lane < left ? list[head + lane] : 0;
      if (lane < left) list[lane] = keep;
    }
    pending = left;
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
