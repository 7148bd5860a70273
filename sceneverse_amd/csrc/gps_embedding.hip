// gps_embedding.hip -- dense gradient of an embedding table on MI355X (gfx950), deterministic.
//
// Reference: the word-embedding lookup of the language encoder (HF BertEmbeddings behind
// modules/language/bert.py:21-26; 30 522 x 768 table, 3 200 + 19 200 token ids per step) whose backward
// torch runs as sort -> segment bookkeeping -> sum_and_scatter (~60 launches, 1.27 ms/step for the two
// BERT passes, profiles/r1/bench_z_kernel_stats.csv).  Here:
//   init     first[] = +big, count[] = 0
//   mark     one thread per token: atomicMin(first[id], t), atomicAdd(count[id], 1)  (order-independent)
//   heavy    [r5] ids with >= 64 tokens ([MASK]: ~1 500 at the bench workload, [CLS] / [SEP]: 128): found in the count
//            table (one workgroup), their tokens listed in ascending order (one workgroup per id), summed 64 rows per
//            wave into partial rows, the partial rows added in order.  Left on the chain below, [MASK] alone is a
//            94-trip dependent chain of row fetches: 98 us, whatever the rest of the launch does.
//   dups     [r5] one workgroup: the other tokens whose id occurs more than once, compacted in ascending token order
//            (token number + id as int32) -- the only ones a duplicate search has to look at
//   fill     [r5] table rows without a token = 0, rows with ONE token = that token's row (one wave per table row and
//            256-column chunk): the table is written exactly once
//   sum      waves walk the duplicate list; an entry that is the FIRST occurrence of its id adds the rows of all tokens
//            with that id in ASCENDING token order (the list behind it scanned 64 entries at a time with a ballot,
//            stop after count[id] matches) and writes the table row once.  No floating-point atomics: the result does
//            not depend on scheduling.
// HBM-bound: n rows of d floats read once, the touched table rows written once, plus the table memset.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gps_hip.h"

namespace gps_emb {

constexpr int kBlock = 256;
constexpr int kMaxChunks = 8;                   // d <= 8 * 256 floats

// first[] = +big, count[] = 0 (a kernel, not two hipMemsetAsync calls: one launch, and no memset nodes in a captured graph)
__global__ __launch_bounds__(kBlock) void init_kernel(int num_rows, int32_t *__restrict__ first, int32_t *__restrict__ count) {
  const int r = blockIdx.x * kBlock + threadIdx.x;
  if (r < num_rows) { first[r] = 0x7F7F7F7F; count[r] = 0; }
}

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

constexpr int kHeavyMin = 64;    // ids with at least this many tokens take the parallel path ...
constexpr int kHeavyCap = 64;    // ... the first kHeavyCap of them in ascending id order (the others stay on the chain)
constexpr int kPartRows = 64;    // token rows per partial sum

// tables of the heavy path inside the scratch buffer (int32 units; the partial rows are floats)
struct Heavy {
  int32_t *n_heavy, *n_parts;      // [1] each
  int32_t *id;                     // [kHeavyCap]       the heavy ids
  int32_t *off;                    // [kHeavyCap + 1]   token-list offsets
  int32_t *part0;                  // [kHeavyCap + 1]   first partial row of each heavy id
  int32_t *part_slot;              // [max_parts]       heavy slot of each partial row
  int32_t *list;                   // [n]               tokens of the heavy ids, slot by slot, ascending
  float *partial;                  // [max_parts][d]
};
__host__ __device__ inline int heavy_max_parts(int n) { return n / kPartRows + kHeavyCap + 1; }

// Stable compaction by ONE workgroup of 16 waves: wave w owns the contiguous range [w per, (w + 1) per) of the n
// items, counts its hits, one barrier, then writes them behind the hits of the waves before it -- two passes over
// L2-resident words and ONE barrier (a trip-by-trip walk of the whole workgroup pays two barriers and a memory round
// trip per 1 024 items: 21 - 70 us at n = 22 400).  pred(i) must be a pure function of i; emit(i, slot) stores.
struct NoFetch {};
template <typename Key, typename Fetch, typename Decide, typename Emit>
__device__ __forceinline__ int compact_ranges(int n, int *s_cnt, Key key, Fetch fetch, Decide decide, Emit emit) {
  // key(i) -> the word read from item i, UNPROCESSED (first memory level); fetch(key) -> the words of a second memory
  // level that the key names, unprocessed and read unconditionally (NoFetch{} when there is none); decide(key, fetched)
  // -> payload (>= 0: hit carrying this int; < 0: no hit), pure arithmetic; emit(i, slot, payload).  Loads and their uses
  // are kept in separate loops: a use next to its load puts a full wait between any two loads of the range.
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int per = ((n + 15) / 16 + 63) & ~63;
  const int lo = min(n, w * per), hi = min(n, lo + per);
  constexpr int kKeep = 32, kBatch = 16;             // ranges of up to 2 048 items: payloads stay in registers
  int val[kKeep];
  const bool keep = per <= 64 * kKeep;               // (workgroup-uniform)
  int mine = 0;
  if (n <= 0) {
    // nothing to look at (key / pred are never called)
  } else if (keep) {
    // CLAMPED indices, no branches, kBatch items per lane at a time: their first-level loads are issued back to back,
    // then their second-level loads -- two memory round trips per 1 024 items of the range instead of two per 64
    // (a test per 64 items: 33 us for n = 22 400)
#pragma unroll
    for (int k0 = 0; k0 < kKeep; k0 += kBatch) {
      decltype(key(0)) kv[kBatch];                   // RAW loaded words: nothing consumes them before all are requested
#pragma unroll
      for (int k = 0; k < kBatch; ++k) kv[k] = key(min(lo + (k0 + k) * 64 + lane, n - 1));
      asm volatile("" ::: "memory");                 // the loads above are issued before anything below
      decltype(fetch(kv[0])) fv[kBatch];
#pragma unroll
      for (int k = 0; k < kBatch; ++k) fv[k] = fetch(kv[k]);
      asm volatile("" ::: "memory");
#pragma unroll
      for (int k = 0; k < kBatch; ++k) val[k0 + k] = decide(kv[k], fv[k]);
    }
#pragma unroll
    for (int k = 0; k < kKeep; ++k) {
      if (lo + k * 64 + lane >= hi) val[k] = -1;
      mine += __popcll(__ballot(val[k] >= 0));
    }
  } else {
    for (int i0 = lo; i0 < hi; i0 += 64) {
      const auto kq = key(min(i0 + lane, n - 1));
      mine += __popcll(__ballot(i0 + lane < hi && decide(kq, fetch(kq)) >= 0));
    }
  }
  if (lane == 0) s_cnt[w] = mine;
  __syncthreads();
  int base = 0, total = 0;
  for (int k = 0; k < 16; ++k) { const int c = s_cnt[k]; base += (k < w) ? c : 0; total += c; }
  if (n <= 0) return 0;
  if (keep) {
#pragma unroll
    for (int k = 0; k < kKeep; ++k) {
      const unsigned long long mask = __ballot(val[k] >= 0);
      if (val[k] >= 0)
        emit(lo + k * 64 + lane, base + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u)), val[k]);
      base += __popcll(mask);
    }
  } else {
    for (int i0 = lo; i0 < hi; i0 += 64) {
      const auto kq = key(min(i0 + lane, n - 1));
      const int dv = decide(kq, fetch(kq));
      const int v = i0 + lane < hi ? dv : -1;
      const unsigned long long mask = __ballot(v >= 0);
      if (v >= 0) emit(i0 + lane, base + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u)), v);
      base += __popcll(mask);
    }
  }
  return total;
}

// heavy ids out of the count table: slot = rank in ascending id order; first[id] = -1 - slot marks them (no token's
// wave of sum_kernel is then a first occurrence: they all leave), offsets of their token lists and partial rows.
__global__ __launch_bounds__(1024) void heavy_find_kernel(int n, int num_rows, int32_t *__restrict__ first,
                                                           const int32_t *__restrict__ count, Heavy H) {
  __shared__ int s_cnt[16];
  __shared__ int s_id[kHeavyCap];
  const int total = compact_ranges(num_rows, s_cnt, [&](int r) { return count[r]; }, [&](int) { return NoFetch{}; },
                                   [&](int c, NoFetch) { return c >= kHeavyMin ? 0 : -1; },
                                   [&](int r, int slot, int) { if (slot < kHeavyCap) s_id[slot] = r; });
  const int nh = min(total, kHeavyCap);
  __syncthreads();
  if (threadIdx.x == 0) {
    int off = 0, parts = 0;
    for (int k = 0; k < nh; ++k) {
      const int c = count[s_id[k]];
      H.id[k] = s_id[k];
      H.off[k] = off;
      H.part0[k] = parts;
      first[s_id[k]] = -1 - k;
      off += c;
      parts += (c + kPartRows - 1) / kPartRows;
    }
    H.off[nh] = off;
    H.part0[nh] = parts;
    *H.n_heavy = nh;
    *H.n_parts = parts;
    s_cnt[0] = parts;
  }
  __syncthreads();
  // partial row -> heavy slot (a heavy id owns a contiguous run of partial rows)
  if ((int)threadIdx.x < nh) {
    const int k = threadIdx.x;
    const int c = count[s_id[k]];
    int p0 = 0;
    for (int q = 0; q < k; ++q) p0 += (count[s_id[q]] + kPartRows - 1) / kPartRows;
    for (int q = 0; q < (c + kPartRows - 1) / kPartRows; ++q) H.part_slot[p0 + q] = k;
  }
}

// one workgroup per heavy id: its tokens in ascending order
// ... and, first, every workgroup's share of the duplicate FLAGS: key32[t] = id of token t if that id has more than one
// token and is not on the heavy path, else -1.  These are two gathers per token into the count / first tables: spread
// over the 64 workgroups of this launch they are nothing; issued by the ONE workgroup that compacts the flags
// (dup_list_kernel) they were 45 K cache-line requests through a single CU's texture path -- 20 of its 26 us.
__global__ __launch_bounds__(1024) void heavy_list_kernel(int n, int num_rows, const int64_t *__restrict__ ids,
                                                           long long padding_idx, const int32_t *__restrict__ first,
                                                           const int32_t *__restrict__ count, int32_t *__restrict__ key32,
                                                           Heavy H) {
  __shared__ int s_cnt[16];
  for (int t = blockIdx.x * 1024 + threadIdx.x; t < n; t += gridDim.x * 1024) {
    const long long id = ids[t];
    const bool in = (id >= 0) & (id < num_rows) & (id != padding_idx);
    const int r = in ? (int)id : 0;
    const int c = count[r], f = first[r];
    key32[t] = (in & (c > 1) & (f >= 0)) ? r : -1;
  }
  const int slot = blockIdx.x;
  if (slot >= *H.n_heavy) return;
  const long long id = H.id[slot];
  int32_t *dst = H.list + H.off[slot];
  compact_ranges(n, s_cnt, [&](int t) { return ids[t]; }, [&](long long) { return NoFetch{}; },
                 [&](long long v, NoFetch) { return v == id ? 0 : -1; },
                 [&](int t, int r, int) { dst[r] = t; });
}

// one wave per (partial row, 256-column chunk): up to kPartRows token rows added in list order
__global__ __launch_bounds__(kBlock) void heavy_partial_kernel(int d, const float *__restrict__ dy, long long ld, Heavy H) {
  const int lane = threadIdx.x & 63;
  const int q = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  if (q >= *H.n_parts) return;
  const int col = blockIdx.y * 256 + lane * 4;
  const bool live = col < d;
  const int colc = live ? col : 0;
  const int slot = H.part_slot[q];
  const int begin = H.off[slot] + (q - H.part0[slot]) * kPartRows, end = min(H.off[slot + 1], begin + kPartRows);
  float4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int r0 = begin; r0 < end; r0 += 16) {
    const int cnt = min(16, end - r0);
    float4 v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = *reinterpret_cast<const float4 *>(dy + (size_t)H.list[r0 + min(u, cnt - 1)] * ld + colc);
#pragma unroll
    for (int u = 0; u < 16; ++u)
      if (u < cnt) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
  }
  if (live) *reinterpret_cast<float4 *>(H.partial + (size_t)q * d + col) = acc;
}

// one wave per (heavy id, 256-column chunk): its partial rows added in order -> the table row
__global__ __launch_bounds__(64) void heavy_combine_kernel(int d, Heavy H, float *__restrict__ out) {
  const int lane = threadIdx.x, slot = blockIdx.x;
  if (slot >= *H.n_heavy) return;
  const int col = blockIdx.y * 256 + lane * 4;
  if (col >= d) return;
  const int begin = H.part0[slot], end = H.part0[slot + 1];
  float4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int r0 = begin; r0 < end; r0 += 16) {
    const int cnt = min(16, end - r0);
    float4 v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = *reinterpret_cast<const float4 *>(H.partial + (size_t)(r0 + min(u, cnt - 1)) * d + col);
#pragma unroll
    for (int u = 0; u < 16; ++u)
      if (u < cnt) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
  }
  *reinterpret_cast<float4 *>(out + (size_t)H.id[slot] * d + col) = acc;
}

// [r5] The duplicate search of sum_kernel used to walk ALL ids behind a token (int64, n = 22 400: up to 179 KB per
// searching wave, ~0.6 GB of L2 reads per launch at the bench workload).  Only tokens whose id occurs more than once
// (and is not on the heavy path) can be anybody's duplicate: they are listed here once, in ascending token order, as
// (token, id) int32 pairs -- 8 bytes per entry of a list a fraction of n long instead of 8 bytes per TOKEN.
__global__ __launch_bounds__(1024) void dup_list_kernel(int n, const int32_t *__restrict__ key32,
                                                         int32_t *__restrict__ dup_t, int32_t *__restrict__ dup_id,
                                                         int32_t *__restrict__ n_dup) {
  __shared__ int s_cnt[16];
  const int total = compact_ranges(n, s_cnt, [&](int t) { return key32[t]; }, [&](int) { return NoFetch{}; },
                                   [&](int k, NoFetch) { return k; },
                                   [&](int t, int slot, int id) { dup_t[slot] = t; dup_id[slot] = id; });
  if (threadIdx.x == 0) *n_dup = total;
}

// [r5] Every table row nobody sums: rows without a token are zero, rows with exactly ONE token (three quarters of the
// live tokens for random ids) are a copy of that token's row.  One wave per (table row, 256-column chunk), no LDS, a
// handful of registers: bandwidth-bound, and the table is written once (it used to be zero-filled by a memset and then
// written again; the single-token rows also sat in sum_kernel's workgroups, which reserve LDS for duplicate lists --
// four workgroups per CU, each alive for two dependent memory round trips: 90 us for what is a 77 MB copy).
__global__ __launch_bounds__(kBlock) void fill_rows_kernel(int d, int num_rows, const float *__restrict__ dy, long long ld,
                                                           const int32_t *__restrict__ first, const int32_t *__restrict__ count,
                                                           float *__restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  const int col = blockIdx.y * 256 + lane * 4;
  if (r >= num_rows || col >= d) return;
  const int c = count[r];
  if (c > 1) return;                                                      // sum_kernel / the heavy path write the row
  float4 v = {0.f, 0.f, 0.f, 0.f};
  if (c == 1) v = *reinterpret_cast<const float4 *>(dy + (size_t)first[r] * ld + col);
  *reinterpret_cast<float4 *>(out + (size_t)r * d + col) = v;
}

// [r4] the duplicate search and the row fetches of a summing wave are decoupled: a scan trip looks at kScan x 64 list
// entries (independent loads, L2 resident) and appends the matching token numbers, in ascending order, to the wave's list
// in LDS; rows are fetched kRows at a time (independent loads) as soon as kRows are pending, so an id with c tokens
// costs (entries behind it) / (64 kScan) + c / kRows round trips.
constexpr int kScan = 8, kRows = 16;
constexpr int kListCap = kScan * 64 + kRows;

// ids with 2 .. 63 tokens (and the heavy ones beyond kHeavyCap): the waves of a fixed-size grid walk the DUPLICATE
// LIST; an entry that is the first occurrence of its id adds the rows of its duplicates, in list (= token) order,
// behind its own and writes the table row once.
__global__ __launch_bounds__(kBlock) void sum_kernel(int d, const float *__restrict__ dy, long long ld,
                                                     const int32_t *__restrict__ first, const int32_t *__restrict__ count,
                                                     const int32_t *__restrict__ dup_t, const int32_t *__restrict__ dup_id,
                                                     const int32_t *__restrict__ n_dup, float *__restrict__ out) {
  __shared__ int list_s[kBlock / 64][kListCap];
  const int lane = threadIdx.x & 63;
  int *list = list_s[threadIdx.x >> 6];
  const int nd = *n_dup;                                                  // entries of the duplicate list (uniform)
  const int col = blockIdx.y * 256 + lane * 4;                            // this wave's 256-column chunk
  const bool live = col < d;
  const int colc = live ? col : 0;                                        // dead lanes read (and drop) the first columns
  const unsigned long long below = (1ull << lane) - 1ull;
  for (int e = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6); e < nd; e += gridDim.x * (kBlock / 64)) {
    const int t = dup_t[e], id32 = dup_id[e];
    if (first[id32] != t) continue;                                       // a later duplicate: its first occurrence sums it
    int missing = count[id32] - 1;                                        // duplicates still to be found (wave-uniform)
    float4 acc = *reinterpret_cast<const float4 *>(dy + (size_t)t * ld + colc);
    int pending = 0;                                                      // list[0 .. pending) wait for their rows
    // the duplicates lie behind entry e (the list is in token order)
    for (int base = e + 1; base < nd && missing > 0; base += 64 * kScan) {
      unsigned long long masks[kScan];
      int toks[kScan], idv[kScan];
#pragma unroll
      for (int u = 0; u < kScan; ++u) {
        const int q = min(base + u * 64 + lane, nd - 1);                  // unconditional: 2 kScan independent loads ...
        toks[u] = dup_t[q];
        idv[u] = dup_id[q];
      }
      asm volatile("" ::: "memory");                                      // ... all requested before the first is used
#pragma unroll
      for (int u = 0; u < kScan; ++u) masks[u] = __ballot((base + u * 64 + lane < nd) & (idv[u] == id32));
#pragma unroll
      for (int u = 0; u < kScan; ++u) {
        const unsigned long long m = masks[u];
        if (m) {                                                          // (wave-uniform)
          if ((m >> lane) & 1ull) list[pending + __popcll(m & below)] = toks[u];
          const int c = __popcll(m);
          pending += c;
          missing -= c;
        }
      }
      const bool last = missing <= 0 || base + 64 * kScan >= nd;
      int head = 0;
      while (pending - head >= kRows || (last && head < pending)) {
        const int cnt = min(kRows, pending - head);
        float4 v[kRows];
#pragma unroll
        for (int q = 0; q < kRows; ++q) {
          const int r = list[head + min(q, cnt - 1)];                     // same address in every lane: a broadcast read
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
    if (live) *reinterpret_cast<float4 *>(out + (size_t)id32 * d + col) = acc;
  }
}

}  // namespace gps_emb

// scratch layout (int32 units)
static long long heavy_ints(int n, int d) {
  using namespace gps_emb;
  return 4 + kHeavyCap + 2 * (kHeavyCap + 1) + heavy_max_parts(n) + n + (long long)heavy_max_parts(n) * d + 8;
}

extern "C" long long gps_embedding_grad_scratch_ints(int n, int num_rows, int d) {
  n = n > 0 ? n : 0;
  return 2ll * (num_rows > 0 ? num_rows : 0) + 3ll * n + 4 + heavy_ints(n, d > 0 ? d : 0);
}

extern "C" int gps_embedding_grad(int n, int d, int num_rows, const int64_t *ids, const float *dy, long long ld,
                                  long long padding_idx, int32_t *scratch, float *out, gps_stream_t stream) {
  using namespace gps_emb;
  if (n < 0 || d < 0 || num_rows < 0 || ld < d) return GPS_ERR_INVALID_ARGUMENT;
  if ((long long)num_rows * d == 0) return GPS_OK;
  if (!out || !scratch || (n > 0 && (!ids || !dy))) return GPS_ERR_INVALID_ARGUMENT;
  if ((d & 3) || (ld & 3) || d > kMaxChunks * 256 || ((uintptr_t)dy & 15u) || ((uintptr_t)out & 15u) || ((uintptr_t)scratch & 15u))
    return GPS_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  int32_t *first = scratch, *count = scratch + num_rows, *dup_t = scratch + 2 * (size_t)num_rows, *dup_id = dup_t + n,
          *key32 = dup_id + n, *n_dup = key32 + n;
  Heavy H;
  int32_t *h = n_dup + 4;
  H.n_heavy = h; H.n_parts = h + 1; h += 4;
  H.id = h; h += kHeavyCap;
  H.off = h; h += kHeavyCap + 1;
  H.part0 = h; h += kHeavyCap + 1;
  H.part_slot = h; h += heavy_max_parts(n);
  H.list = h; h += n;
  h = reinterpret_cast<int32_t *>((reinterpret_cast<uintptr_t>(h) + 15u) & ~(uintptr_t)15u);      // float4 rows
  H.partial = reinterpret_cast<float *>(h);
  if (n == 0) return hipMemsetAsync(out, 0, (size_t)num_rows * d * sizeof(float), s) == hipSuccess ? GPS_OK : GPS_ERR_LAUNCH;
  hipLaunchKernelGGL(init_kernel, dim3((num_rows + kBlock - 1) / kBlock), dim3(kBlock), 0, s, num_rows, first, count);
  const int chunks = (d + 255) / 256;
  hipLaunchKernelGGL(mark_kernel, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, s, n, num_rows, ids, padding_idx, first,
                     count);
  hipLaunchKernelGGL(heavy_find_kernel, dim3(1), dim3(1024), 0, s, n, num_rows, first, count, H);
  hipLaunchKernelGGL(heavy_list_kernel, dim3(kHeavyCap), dim3(1024), 0, s, n, num_rows, ids, padding_idx, first, count, key32, H);
  hipLaunchKernelGGL(dup_list_kernel, dim3(1), dim3(1024), 0, s, n, key32, dup_t, dup_id, n_dup);
  const int waves_per_block = kBlock / 64;
  hipLaunchKernelGGL(fill_rows_kernel, dim3((num_rows + waves_per_block - 1) / waves_per_block, chunks), dim3(kBlock), 0, s, d,
                     num_rows, dy, ld, first, count, out);
  // the duplicate list is at most n entries long; 1024 workgroups (4 K waves) walk it with a grid stride
  hipLaunchKernelGGL(sum_kernel, dim3(min(1024, (n + waves_per_block - 1) / waves_per_block), chunks), dim3(kBlock), 0, s, d, dy,
                     ld, first, count, dup_t, dup_id, n_dup, out);
  hipLaunchKernelGGL(heavy_partial_kernel, dim3((heavy_max_parts(n) + waves_per_block - 1) / waves_per_block, chunks),
                     dim3(kBlock), 0, s, d, dy, ld, H);
  hipLaunchKernelGGL(heavy_combine_kernel, dim3(kHeavyCap, chunks), dim3(64), 0, s, d, H, out);
  return hipGetLastError() == hipSuccess ? GPS_OK : GPS_ERR_LAUNCH;
}
