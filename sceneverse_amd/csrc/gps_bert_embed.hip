// gps_bert_embed.hip -- the embedding block of the BERT text encoder as one launch per direction on MI355X (gfx950).
//
// Reference: HF BertEmbeddings.forward behind modules/language/bert.py:21-26 (token_type_ids = None, position_ids =
// None):   e = word[id] + type[0] + pos[p];   y = dropout(LayerNorm(e))
// which torch runs as gather, gather, add, add, layer_norm, dropout (+ the fp32 -> bf16 copy the first GEMM wants) and,
// backward, dropout_backward, layer_norm_backward (2 kernels), a 22 400-row column sum for the type row, index_add for
// the position table and the word-table scatter: ~0.5 ms of the 15.6 ms pre-train step in a dozen launches
// (profiles/r3/step_attrib_o.txt).  Here:
//   forward   one wave per token row: the three table rows are summed in registers, mean / variance two-pass from
//             registers, y (fp32) and its bf16 copy written once; mean and rstd (2 floats per row) are all that is
//             saved -- the pre-LayerNorm sum is NOT written: the backward pass gathers it again (the position and type
//             rows hit L2, the word rows are the same 3 KB per token a saved copy would cost to read, without the write)
//   backward  dz = LayerNorm'(dy * keep / (1 - p)) per row -> dz (fp32, the operand of the two table-gradient scatters,
//             gps_embedding_grad) + per-workgroup partial column sums for dgamma / dbeta (summed by
//             gps_ln_reduce_partials); rows past the device-side row count get dz = 0.
// The type-row gradient is the column sum of the position-table gradient (every row has exactly one position).
// Dropout: the counter-based stream of gps_layernorm.hip (element index = row * d + column), recomputed in backward.
// HBM-bound: forward reads 3 KB (word row) and writes 4.5 KB per live row at d = 768; backward reads 6 KB, writes 3 KB.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gps_hip.h"

namespace gps_bert_embed {

constexpr int kBlock = 256;
constexpr int kWaves = kBlock / 64;

__device__ __forceinline__ unsigned int mix32(unsigned int x) {      // as in gps_layernorm.hip
  x ^= x >> 16;
  x *= 0x21F0AAADu;
  x ^= x >> 15;
  x *= 0x735A2D97u;
  x ^= x >> 15;
  return x;
}
__device__ __forceinline__ unsigned int rng_u32(unsigned long long seed, unsigned long long idx) {
  const unsigned int s = mix32((unsigned int)seed ^ mix32((unsigned int)(seed >> 32) + 0x9E3779B9u));
  return mix32(((unsigned int)idx + (unsigned int)(idx >> 32) * 0x85EBCA6Bu) ^ s);
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ unsigned int pack2(float lo, float hi) {   // round to nearest even (v_cvt_pk_bf16_f32)
  return (unsigned int)__builtin_bit_cast(uint16_t, (__bf16)lo) | ((unsigned int)__builtin_bit_cast(uint16_t, (__bf16)hi) << 16);
}
// keep * scale of the 4 elements starting at element index e0 (1 everywhere without dropout)
__device__ __forceinline__ float4 keep4(unsigned int thr, float scale, unsigned long long seed, unsigned long long e0) {
  if (thr == 0u) return make_float4(1.f, 1.f, 1.f, 1.f);
  return make_float4(rng_u32(seed, e0 + 0) >= thr ? scale : 0.f, rng_u32(seed, e0 + 1) >= thr ? scale : 0.f,
                     rng_u32(seed, e0 + 2) >= thr ? scale : 0.f, rng_u32(seed, e0 + 3) >= thr ? scale : 0.f);
}
// (word[id] + type) + pos[p] for the 4 columns at c0: HF's order of the two additions
__device__ __forceinline__ float4 gather4(const float *__restrict__ w, const float *__restrict__ p, const float *__restrict__ t,
                                          int c0) {
  const float4 a = *reinterpret_cast<const float4 *>(w + c0);
  const float4 b = *reinterpret_cast<const float4 *>(t + c0);
  const float4 c = *reinterpret_cast<const float4 *>(p + c0);
  return make_float4((a.x + b.x) + c.x, (a.y + b.y) + c.y, (a.z + b.z) + c.z, (a.w + b.w) + c.w);
}

template <int ITERS>
__global__ __launch_bounds__(kBlock) void fwd_kernel(int n_rows, int d, const int64_t *__restrict__ ids,
                                                     const int64_t *__restrict__ pos, const float *__restrict__ word,
                                                     const float *__restrict__ pos_tab, const float *__restrict__ type_row,
                                                     const float *__restrict__ gamma, const float *__restrict__ beta, float eps,
                                                     float p_drop, unsigned int thr, unsigned long long seed,
                                                     const unsigned long long *__restrict__ seed_dev, float *__restrict__ y,
                                                     uint16_t *__restrict__ y16, float *__restrict__ mean_out,
                                                     float *__restrict__ rstd_out, const int *__restrict__ rows_dev,
                                                     const int *__restrict__ poison_dev) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (rows_dev) n_rows = min(n_rows, *rows_dev);        // device-side count of leading rows that carry work
  // a plan built from masks that break its precondition (gps_varlen_plan's violation word): every row leaves as NaN, so
  // that the loss of the step is NaN instead of silently wrong
  const float poison = (poison_dev && *poison_dev != 0) ? __builtin_nanf("") : 1.f;
  const float scale = thr ? 1.f / (1.f - p_drop) : 1.f;
  const unsigned long long sd = seed + ((thr && seed_dev) ? *seed_dev : 0ull);
  const float inv_d = 1.f / (float)d;
  for (int row = blockIdx.x * kWaves + wave; row < n_rows; row += gridDim.x * kWaves) {
    const float *w = word + (size_t)ids[row] * d, *p = pos_tab + (size_t)pos[row] * d;
    float4 z[ITERS];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < ITERS; ++i) {
      z[i] = gather4(w, p, type_row, (i * 64 + lane) * 4);
      s += (z[i].x + z[i].y) + (z[i].z + z[i].w);
    }
    const float mean = wave_sum(s) * inv_d;
    float v = 0.f;
#pragma unroll
    for (int i = 0; i < ITERS; ++i) {
      const float a = z[i].x - mean, b = z[i].y - mean, c = z[i].z - mean, e = z[i].w - mean;
      v += (a * a + b * b) + (c * c + e * e);
    }
    const float rstd = rsqrtf(wave_sum(v) * inv_d + eps) * poison;       // poisoned plan: NaN into every element
    if (lane == 0) { mean_out[row] = mean; rstd_out[row] = rstd; }
    const size_t base = (size_t)row * d;
#pragma unroll
    for (int i = 0; i < ITERS; ++i) {
      const int c0 = (i * 64 + lane) * 4;
      const float4 g = *reinterpret_cast<const float4 *>(gamma + c0);
      const float4 bt = *reinterpret_cast<const float4 *>(beta + c0);
      const float4 k = keep4(thr, scale, sd, base + c0);
      float4 o;
      o.x = ((z[i].x - mean) * rstd * g.x + bt.x) * k.x;
      o.y = ((z[i].y - mean) * rstd * g.y + bt.y) * k.y;
      o.z = ((z[i].z - mean) * rstd * g.z + bt.z) * k.z;
      o.w = ((z[i].w - mean) * rstd * g.w + bt.w) * k.w;
      *reinterpret_cast<float4 *>(y + base + c0) = o;
      if (y16) *reinterpret_cast<uint2 *>(y16 + base + c0) = make_uint2(pack2(o.x, o.y), pack2(o.z, o.w));
    }
  }
}

template <int ITERS>
__global__ __launch_bounds__(kBlock) void bwd_kernel(int n_rows, int d, const float *__restrict__ dy,
                                                     const uint16_t *__restrict__ dy16, const int64_t *__restrict__ ids,
                                                     const int64_t *__restrict__ pos, const float *__restrict__ word,
                                                     const float *__restrict__ pos_tab, const float *__restrict__ type_row,
                                                     const float *__restrict__ gamma, const float *__restrict__ mean_in,
                                                     const float *__restrict__ rstd_in, float p_drop, unsigned int thr,
                                                     unsigned long long seed, const unsigned long long *__restrict__ seed_dev,
                                                     float *__restrict__ dz_out, float *__restrict__ dgamma_part,
                                                     float *__restrict__ dbeta_part, const int *__restrict__ rows_dev) {
  extern __shared__ float red[];      // [kWaves][d], used for dgamma then dbeta
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n_live = rows_dev ? min(n_rows, *rows_dev) : n_rows;
  const float scale = thr ? 1.f / (1.f - p_drop) : 1.f;
  const unsigned long long sd = seed + ((thr && seed_dev) ? *seed_dev : 0ull);
  const float inv_d = 1.f / (float)d;
  float4 gacc[ITERS], bacc[ITERS], gm[ITERS];
#pragma unroll
  for (int i = 0; i < ITERS; ++i) {
    gacc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    bacc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    gm[i] = *reinterpret_cast<const float4 *>(gamma + (i * 64 + lane) * 4);
  }
  for (int row = blockIdx.x * kWaves + wave; row < n_rows; row += gridDim.x * kWaves) {
    const size_t base = (size_t)row * d;
    if (row >= n_live) {               // rows nobody computed: their gradient is zero for the table scatters
#pragma unroll
      for (int i = 0; i < ITERS; ++i) *reinterpret_cast<float4 *>(dz_out + base + (i * 64 + lane) * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
      continue;
    }
    const float *w = word + (size_t)ids[row] * d, *p = pos_tab + (size_t)pos[row] * d;
    const float mean = mean_in[row], rstd = rstd_in[row];
    float4 zh[ITERS], a[ITERS];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < ITERS; ++i) {
      const int c0 = (i * 64 + lane) * 4;
      float4 g = *reinterpret_cast<const float4 *>(dy + base + c0);
      if (dy16) {                      // gradient that arrived through the bf16 copy of y
        const uint2 u = *reinterpret_cast<const uint2 *>(dy16 + base + c0);
        g.x += __uint_as_float(u.x << 16); g.y += __uint_as_float(u.x & 0xFFFF0000u);
        g.z += __uint_as_float(u.y << 16); g.w += __uint_as_float(u.y & 0xFFFF0000u);
      }
      const float4 k = keep4(thr, scale, sd, base + c0);
      g = make_float4(g.x * k.x, g.y * k.y, g.z * k.z, g.w * k.w);         // gradient of the LayerNorm output
      const float4 e = gather4(w, p, type_row, c0);
      zh[i] = make_float4((e.x - mean) * rstd, (e.y - mean) * rstd, (e.z - mean) * rstd, (e.w - mean) * rstd);
      a[i] = make_float4(g.x * gm[i].x, g.y * gm[i].y, g.z * gm[i].z, g.w * gm[i].w);
      s1 += (a[i].x + a[i].y) + (a[i].z + a[i].w);
      s2 += (a[i].x * zh[i].x + a[i].y * zh[i].y) + (a[i].z * zh[i].z + a[i].w * zh[i].w);
      gacc[i].x += g.x * zh[i].x; gacc[i].y += g.y * zh[i].y; gacc[i].z += g.z * zh[i].z; gacc[i].w += g.w * zh[i].w;
      bacc[i].x += g.x; bacc[i].y += g.y; bacc[i].z += g.z; bacc[i].w += g.w;
    }
    s1 = wave_sum(s1) * inv_d;
    s2 = wave_sum(s2) * inv_d;
#pragma unroll
    for (int i = 0; i < ITERS; ++i) {
      float4 dz;
      dz.x = rstd * (a[i].x - s1 - zh[i].x * s2);
      dz.y = rstd * (a[i].y - s1 - zh[i].y * s2);
      dz.z = rstd * (a[i].z - s1 - zh[i].z * s2);
      dz.w = rstd * (a[i].w - s1 - zh[i].w * s2);
      *reinterpret_cast<float4 *>(dz_out + base + (i * 64 + lane) * 4) = dz;
    }
  }
  for (int pass = 0; pass < 2; ++pass) {      // cross-wave reduction of the column sums, one partial row per workgroup
    __syncthreads();
#pragma unroll
    for (int i = 0; i < ITERS; ++i) *reinterpret_cast<float4 *>(red + wave * d + (i * 64 + lane) * 4) = pass == 0 ? gacc[i] : bacc[i];
    __syncthreads();
    float *dst = (pass == 0 ? dgamma_part : dbeta_part) + (size_t)blockIdx.x * d;
    for (int c = threadIdx.x; c < d; c += kBlock) {
      float t = 0.f;
#pragma unroll
      for (int w2 = 0; w2 < kWaves; ++w2) t += red[w2 * d + c];
      dst[c] = t;
    }
  }
}

// Position-table gradient when the rows are whole sequences laid end to end with position = offset inside the sequence
// (the variable-length text batch): out[p] = sum over the sequences s longer than p of dz[cu[s] + p], in sequence order.
// One workgroup per position; its 4 waves take every 4th sequence (8 independent row loads in flight each) and combine
// through LDS in wave order: deterministic, every dz row is read once.  (gps_embedding_grad on the position ids does the
// same sums with one wave per DISTINCT id walking its ~100 duplicates one after the other: 193 us against ~15.)
template <int ITERS>
__global__ __launch_bounds__(kBlock) void pos_grad_kernel(int n_seq, int d, const int *__restrict__ cu,
                                                          const float *__restrict__ dz, float *__restrict__ out) {
  extern __shared__ float red[];      // [kWaves][d]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int p = blockIdx.x;
  float4 acc[ITERS];
#pragma unroll
  for (int i = 0; i < ITERS; ++i) acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  constexpr int kAhead = 8;
  for (int s0 = wave; s0 < n_seq; s0 += kWaves * kAhead) {
    float4 v[kAhead][ITERS];
#pragma unroll
    for (int u = 0; u < kAhead; ++u) {
      const int s = s0 + u * kWaves;
      int row = -1;
      if (s < n_seq) {
        const int b = cu[s], e = cu[s + 1];
        if (p < e - b) row = b + p;
      }
#pragma unroll
      for (int i = 0; i < ITERS; ++i)
        v[u][i] = row >= 0 ? *reinterpret_cast<const float4 *>(dz + (size_t)row * d + (i * 64 + lane) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < kAhead; ++u)
#pragma unroll
      for (int i = 0; i < ITERS; ++i) {
        acc[i].x += v[u][i].x; acc[i].y += v[u][i].y; acc[i].z += v[u][i].z; acc[i].w += v[u][i].w;
      }
  }
#pragma unroll
  for (int i = 0; i < ITERS; ++i) *reinterpret_cast<float4 *>(red + wave * d + (i * 64 + lane) * 4) = acc[i];
  __syncthreads();
  for (int c = threadIdx.x; c < d; c += kBlock) {
    float t = 0.f;
#pragma unroll
    for (int w2 = 0; w2 < kWaves; ++w2) t += red[w2 * d + c];
    out[(size_t)p * d + c] = t;
  }
}

inline int grid_rows(int n_rows) {
  int g = (n_rows + kWaves - 1) / kWaves;
  return g > 1024 ? 1024 : (g < 1 ? 1 : g);
}


// ---- index plan of the variable-length text path: everything the host derived from the attention masks with ~30 tiny
// torch launches (cat / sum / stable argsort / cumsum / argsort / index_select / scatter / arange ...), in two launches.
// Masks are non-empty PREFIXES of their rows (the caller checked, see modules/language/bert.py::_masks_are_prefixes), so
// the stable "valid rows first" permutation has a closed form: token j < len_s of sequence s is compact row cu[s] + j,
// a padded position with flat index e is compact row n_valid + e - cu[s + 1] (the padded positions in flat order).
constexpr int kPlanMaxSeq = 8192;

struct PlanText {
  const int64_t *ids;
  const void *mask;
  int elem_bytes, is_float, n_seq, len;
  int seq0;           // sequences before this text
  long long tok0;     // token positions before this text
};
struct PlanArgs {
  PlanText t[GPS_VARLEN_MAX_TEXTS];
  int n_texts, n_seq, n_seq_full;
  long long n_tok, n_tok_full;
};

__device__ __forceinline__ bool mask_set(const void *m, size_t i, int eb, int is_float) {
  switch (eb) {
    case 1: return reinterpret_cast<const uint8_t *>(m)[i] != 0;
    case 2: { const uint16_t v = reinterpret_cast<const uint16_t *>(m)[i]; return (is_float ? (v & 0x7FFFu) : v) != 0; }
    case 4: { const uint32_t v = reinterpret_cast<const uint32_t *>(m)[i]; return (is_float ? (v & 0x7FFFFFFFu) : v) != 0; }
    default: { const uint64_t v = reinterpret_cast<const uint64_t *>(m)[i];
               return (is_float ? (v & 0x7FFFFFFFFFFFFFFFull) : v) != 0; }
  }
}

// [r6] Two launches instead of one 1 024-thread workgroup (45 us of serial memory round trips: 16 waves x 8 sequences x
// two passes): (1) one WAVE per sequence counts its mask, (2) one WORKGROUP per sequence scans the S lengths again (S ints
// from L2), ranks its sequence for the dispatch order and writes its rows of the plan.
constexpr int kLensWaves = 4;
__device__ __forceinline__ int plan_text_of(const PlanArgs &A, int s) {      // text holding global sequence s
  int ti = 0;
  while (ti + 1 < A.n_texts && s >= A.t[ti + 1].seq0) ++ti;
  return ti;
}

// (a) sequence lengths: lens[s] = set elements; flag[s] = 1 when the mask is not a non-empty prefix of its row
__global__ __launch_bounds__(64 * kLensWaves) void varlen_lens_kernel(const PlanArgs A, int32_t *__restrict__ o_lens,
                                                                      int32_t *__restrict__ o_flag) {
  const int lane = threadIdx.x & 63;
  const int s = blockIdx.x * kLensWaves + (threadIdx.x >> 6);
  if (s >= A.n_seq) return;
  const PlanText &T = A.t[plan_text_of(A, s)];
  const int b = s - T.seq0;
  int cnt = 0, end = 0;                                    // set elements | one past the last set element
  for (int j0 = 0; j0 < T.len; j0 += 256) {               // four independent mask loads per trip
    bool set[4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
      set[u] = mask_set(T.mask, (size_t)b * T.len + min(j0 + 64 * u + lane, T.len - 1), T.elem_bytes, T.is_float);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const unsigned long long bits = __ballot(j0 + 64 * u + lane < T.len && set[u]);
      cnt += __popcll(bits);
      if (bits) end = j0 + 64 * u + 64 - __clzll(bits);
    }
  }
  if (lane == 0) {
    o_lens[s] = cnt;
    o_flag[s] = (cnt == 0 || cnt != end) ? 1 : 0;          // empty row, a hole, or left padding
  }
}

constexpr int kRowsThreads = 256;
__global__ __launch_bounds__(kRowsThreads) void varlen_plan_kernel(const PlanArgs A, int32_t *__restrict__ i32_out,
                                                                   int64_t *__restrict__ i64_out,
                                                                   uint8_t *__restrict__ valid_out) {
  extern __shared__ int plan_lds[];            // lens[S] | cu[S + 1]
  int *lens = plan_lds, *cu = plan_lds + A.n_seq;
  __shared__ int red[kRowsThreads / 64];
  const int S = A.n_seq, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int S_full = A.n_seq_full;
  int32_t *o_lens = i32_out, *o_cu = i32_out + S, *o_order = i32_out + 2 * S + 1, *o_qlim = i32_out + 3 * S + 1,
          *o_scal = i32_out + 4 * S + 1;
  for (int s = tid; s < S; s += kRowsThreads) lens[s] = o_lens[s];
  // workgroup 0 alone reads the flags varlen_lens_kernel left in the q_limit slots, and alone overwrites those slots
  int viol = 0;
  if (blockIdx.x == 0)
    for (int s = tid; s < S; s += kRowsThreads) viol |= o_qlim[s];
  __syncthreads();
  // (b) row offsets of the compacted sequences (wave 0, 64 sequences per step)
  if (wave == 0) {
    int carry = 0;
    for (int s0 = 0; s0 < S; s0 += 64) {
      const int s = s0 + lane;
      const int v = s < S ? lens[s] : 0;
      int incl = v;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_up(incl, off, 64);
        if (lane >= off) incl += o;
      }
      if (s < S) cu[s] = carry + incl - v;
      carry += __shfl(incl, 63, 64);
    }
    if (lane == 0) cu[S] = carry;
  }
  __syncthreads();
  const int n_valid = cu[S];
  // (c) dispatch order: longest sequence first (ties by index), by counting; every workgroup ranks its own sequence
  const int s = blockIdx.x, l = lens[s];
  {
    int rank = 0;
    for (int q = tid; q < S; q += kRowsThreads) {
      const int lq = lens[q];
      rank += (lq > l || (lq == l && q < s)) ? 1 : 0;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) rank += __shfl_xor(rank, off, 64);
    if (lane == 0) red[wave] = rank;
    __syncthreads();
    if (tid == 0) {
      int r = 0;
#pragma unroll
      for (int w = 0; w < kRowsThreads / 64; ++w) r += red[w];
      o_order[r] = s;
      o_cu[s] = cu[s];
    }
  }
  if (blockIdx.x == 0) {
    __syncthreads();                               // red[] is reused
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) viol |= __shfl_xor(viol, off, 64);
    if (lane == 0) red[wave] = viol;
    __syncthreads();
    for (int q = tid; q < S; q += kRowsThreads) o_qlim[q] = q < S_full ? lens[q] : 1;
    if (tid == 0) {
      int v = 0;
#pragma unroll
      for (int w = 0; w < kRowsThreads / 64; ++w) v |= red[w];
      o_cu[S] = n_valid;
      o_scal[0] = n_valid;
      o_scal[1] = cu[S_full];                        // live rows of the fully-read texts
      o_scal[2] = cu[S_full] + (S - S_full);         // rows of the last layer's tail batch
      o_scal[3] = v;                                 // != 0: the plan is NOT what the torch formulation would give
    }
  }
  // (d) the compaction itself: this workgroup's sequence, 256 consecutive positions per step (coalesced, no divisions)
  int64_t *o_ids = i64_out, *o_pos = i64_out + A.n_tok, *o_inv = i64_out + 2 * A.n_tok, *o_sel = i64_out + 3 * A.n_tok;
  {
    const PlanText &T = A.t[plan_text_of(A, s)];
    const int b = s - T.seq0;
    const long long c_valid = cu[s];                                   // compact row of the sequence's first token
    const long long row0 = (long long)b * T.len, e0 = T.tok0 + row0;   // first position: inside the text / flat
    const long long c_pad = (long long)n_valid + e0 - cu[s + 1];       // compact row of flat position e0 if it were padding
    // two steps per trip, their id loads first: the stores below may alias the ids as far as the compiler knows
    for (int j0 = tid; j0 < T.len + tid; j0 += 2 * kRowsThreads) {
      int64_t idv[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) idv[u] = T.ids[row0 + min(j0 + kRowsThreads * u, T.len - 1)];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int j = j0 + kRowsThreads * u;
        if (j < T.len) {
          const bool ok = j < l;
          const long long c = (ok ? c_valid : c_pad) + j;
          o_ids[c] = idv[u];
          o_pos[c] = j;
          o_inv[e0 + j] = c;
          valid_out[e0 + j] = ok ? 1 : 0;
        }
      }
    }
  }
  // (e) row selection of the last layer's tail batch: first row of every [CLS]-only sequence, then the fully-read rows
  if (S_full > 0 && S_full < S) {
    const int n_cls = S - S_full;
    for (long long i = (long long)blockIdx.x * kRowsThreads + tid; i < n_cls + A.n_tok_full; i += (long long)gridDim.x * kRowsThreads)
      o_sel[i] = i < n_cls ? (long long)cu[S_full + (int)i] : i - n_cls;
  }
}

}  // namespace gps_bert_embed

// ---------------------------------------------------------------------------------------------------------
// Row-compaction plan of a batch of fixed-length sequences with an arbitrary validity mask (the joint text + object
// sequences of the unified encoder: [valid text tokens | text padding | valid objects | padded object slots], reference
// modules/grounding/unified_encoder.py:147-177 runs every padded row through its four layers; only the valid rows are ever
// read: as attention keys the padded ones are masked, as outputs they are ignored by every head and loss).
//   perm (n) int64   compact row r <- flat row perm[r]: the valid rows in their order, then the invalid ones
//   inv  (n) int64   flat row e -> its compact row
//   cu (n_seq + 1) int32   first compact row of every sequence (its valid rows are contiguous), cu[n_seq] = n_live
//   n_live (1) int32
// One workgroup; two passes over the mask around one block-wide exclusive scan.
// ---------------------------------------------------------------------------------------------------------
namespace gps_rowplan {
constexpr int kThreads = 1024;
__global__ __launch_bounds__(kThreads) void plan_kernel(int n_seq, int seq_len, const unsigned char *__restrict__ valid,
                                                        long long *__restrict__ perm, long long *__restrict__ inv,
                                                        int *__restrict__ cu, int *__restrict__ n_live) {
  __shared__ int part[kThreads];
  __shared__ int wave_tot[kThreads / 64];
  const int n = n_seq * seq_len;
  const int per = (n + kThreads - 1) / kThreads;
  const int lo = min(n, (int)threadIdx.x * per), hi = min(n, lo + per);
  int cnt = 0;
  for (int e = lo; e < hi; ++e) cnt += valid[e] ? 1 : 0;
  // exclusive scan of the per-thread counts: inside each wave by shuffles, across the 16 waves through LDS
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int incl = cnt;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int v = __shfl_up(incl, off, 64);
    if (lane >= off) incl += v;
  }
  if (lane == 63) wave_tot[wave] = incl;
  __syncthreads();
  int base = 0, total = 0;
  for (int w = 0; w < kThreads / 64; ++w) {
    const int t = wave_tot[w];
    if (w < wave) base += t;
    total += t;
  }
  int before = base + incl - cnt;                 // valid rows in front of this thread's chunk
  (void)part;
  for (int e = lo; e < hi; ++e) {
    if (e % seq_len == 0) cu[e / seq_len] = before;
    const bool v = valid[e] != 0;
    const int r = v ? before : total + (e - before);      // invalid rows follow the valid ones, in their order
    perm[r] = e;
    inv[e] = r;
    before += v ? 1 : 0;
  }
  if (threadIdx.x == 0) {
    cu[n_seq] = total;
    n_live[0] = total;
  }
}
// rows of `q` 16-byte words moved between two row arrays of any element type: row r of the launch reads source row
// (src_idx ? src_idx[r] : r) and writes destination row (dst_idx ? dst_idx[r] : r) when r < *n_live (n_live optional) and both
// rows exist; with zero_dead the destination rows of the launch rows at or past *n_live are zeroed instead.  One wave per row.
__global__ __launch_bounds__(256) void rows_move_kernel(int n, long long n_src, long long n_dst, int q, const uint4 *__restrict__ src,
                                                        const long long *__restrict__ src_idx, uint4 *__restrict__ dst,
                                                        const long long *__restrict__ dst_idx, const int *__restrict__ n_live,
                                                        int zero_dead) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int live = n_live ? min(n, max(*n_live, 0)) : n;
  for (int r = blockIdx.x * 4 + wave; r < n; r += gridDim.x * 4) {
    if (r >= live) {
      const long long z = !zero_dead ? -1 : dst_idx ? dst_idx[r] : (long long)r;
      if (z >= 0 && z < n_dst)
        for (int c = lane; c < q; c += 64) dst[(size_t)z * q + c] = make_uint4(0u, 0u, 0u, 0u);
      continue;
    }
    const long long s = src_idx ? src_idx[r] : (long long)r;
    const long long d = dst_idx ? dst_idx[r] : (long long)r;
    if (d < 0 || d >= n_dst) continue;
    const bool ok = s >= 0 && s < n_src;
    for (int c = lane; c < q; c += 64) dst[(size_t)d * q + c] = ok ? src[(size_t)s * q + c] : make_uint4(0u, 0u, 0u, 0u);
  }
}

// The padded side of the joint sequences is TWO arrays -- text rows (B, La, d) and object rows (B, Lb, d) -- that the reference
// concatenates along the sequence axis (modules/grounding/unified_encoder.py:147-177); flat row e = (b, t), t in [0, La + Lb),
// lives in the first array for t < La, in the second otherwise.  pack2: out[r] = flat[perm[r]] for r < *n_live, zeros past it
// (+ the bf16 copy); unpack2: flat[e] = valid[e] ? packed[inv[e]] : 0, written straight into the two arrays.  Each is the other's
// gradient.  No concatenated (B, La + Lb, d) tensor exists on either side, and the halves come out contiguous.
__device__ __forceinline__ size_t flat_row_offset(long long e, int T, int La, int d4, bool &second) {
  const long long b = e / T;
  const int t = (int)(e - b * T);
  second = t >= La;
  return second ? (size_t)(b * (T - La) + (t - La)) * d4 : (size_t)(b * La + t) * d4;
}
__global__ __launch_bounds__(256) void rows_pack2_kernel(int n, int d4, int T, int La, const float4 *__restrict__ a,
                                                         const float4 *__restrict__ b, const long long *__restrict__ perm,
                                                         const int *__restrict__ n_live, float4 *__restrict__ out,
                                                         uint2 *__restrict__ out16) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int live = n_live ? min(n, max(*n_live, 0)) : n;
  for (int r = blockIdx.x * 4 + wave; r < n; r += gridDim.x * 4) {
    bool on = r < live;
    const long long e = on ? perm[r] : 0;
    on = on && e >= 0 && e < (long long)n;
    bool second = false;
    const size_t off = on ? flat_row_offset(e, T, La, d4, second) : 0;
    const float4 *src = second ? b : a;
    for (int c = lane; c < d4; c += 64) {
      const float4 v = on ? src[off + c] : make_float4(0.f, 0.f, 0.f, 0.f);
      out[(size_t)r * d4 + c] = v;
      if (out16) {
        typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
        const bf16x4_t h = {(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
        out16[(size_t)r * d4 + c] = __builtin_bit_cast(uint2, h);
      }
    }
  }
}
__global__ __launch_bounds__(256) void rows_unpack2_kernel(int n, int d4, int T, int La, const float4 *__restrict__ packed,
                                                           const long long *__restrict__ inv, const unsigned char *__restrict__ valid,
                                                           float4 *__restrict__ out_a, float4 *__restrict__ out_b) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int e = blockIdx.x * 4 + wave; e < n; e += gridDim.x * 4) {
    bool on = valid[e] != 0;
    const long long r = on ? inv[e] : 0;
    on = on && r >= 0 && r < (long long)n;
    bool second = false;
    const size_t off = flat_row_offset(e, T, La, d4, second);
    float4 *dst = second ? out_b : out_a;
    for (int c = lane; c < d4; c += 64) dst[off + c] = on ? packed[(size_t)r * d4 + c] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
}

// The input of the unified encoder's first layer in one launch: packed rows x = joint + extra, e = extra (the addend the later
// layers re-add), x16 = bf16(x), where joint = (a | b) and extra = (ea | eb) in the flat (sequence, position) order and row r of
// the outputs is flat row perm[r] (zeros for r >= *n_live) -- replaces cat + cat + add + two gathers + the bf16 cast in front of
// the first projection.  Its gradient launch takes dx, the bf16 gradient that came back through x16 and de, and writes the four
// flat-side gradients (zeros at invalid positions): d(a | b) = g, d(ea | eb) = g + de with g = dx + dx16.
__global__ __launch_bounds__(256) void joint_embed_fwd_kernel(int n, int d4, int T, int La, const float4 *__restrict__ a,
                                                              const float4 *__restrict__ b, const float4 *__restrict__ ea,
                                                              const float4 *__restrict__ eb, const long long *__restrict__ perm,
                                                              const int *__restrict__ n_live, float4 *__restrict__ x,
                                                              float4 *__restrict__ e_out, uint2 *__restrict__ x16) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int live = n_live ? min(n, max(*n_live, 0)) : n;
  for (int r = blockIdx.x * 4 + wave; r < n; r += gridDim.x * 4) {
    bool on = r < live;
    const long long fe = on ? perm[r] : 0;
    on = on && fe >= 0 && fe < (long long)n;
    bool second = false;
    const size_t off = on ? flat_row_offset(fe, T, La, d4, second) : 0;
    const float4 *sj = second ? b : a, *se = second ? eb : ea;
    for (int c = lane; c < d4; c += 64) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f), w = v;
      if (on) {
        const float4 j = sj[off + c];
        w = se[off + c];
        v = make_float4(j.x + w.x, j.y + w.y, j.z + w.z, j.w + w.w);
      }
      x[(size_t)r * d4 + c] = v;
      e_out[(size_t)r * d4 + c] = w;
      typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
      const bf16x4_t h = {(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
      x16[(size_t)r * d4 + c] = __builtin_bit_cast(uint2, h);
    }
  }
}
__global__ __launch_bounds__(256) void joint_embed_bwd_kernel(int n, int d4, int T, int La, const float4 *__restrict__ dx,
                                                              const uint2 *__restrict__ dx16, const float4 *__restrict__ de,
                                                              const long long *__restrict__ inv, const unsigned char *__restrict__ valid,
                                                              float4 *__restrict__ da, float4 *__restrict__ db,
                                                              float4 *__restrict__ dea, float4 *__restrict__ deb) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int fe = blockIdx.x * 4 + wave; fe < n; fe += gridDim.x * 4) {
    bool on = valid[fe] != 0;
    const long long r = on ? inv[fe] : 0;
    on = on && r >= 0 && r < (long long)n;
    bool second = false;
    const size_t off = flat_row_offset(fe, T, La, d4, second);
    float4 *oj = second ? db : da, *oe = second ? deb : dea;
    for (int c = lane; c < d4; c += 64) {
      float4 g = make_float4(0.f, 0.f, 0.f, 0.f), ge = g;
      if (on) {
        const size_t p = (size_t)r * d4 + c;
        if (dx) g = dx[p];
        if (dx16) {
          const uint2 h = dx16[p];
          g.x += __uint_as_float(h.x << 16); g.y += __uint_as_float(h.x & 0xFFFF0000u);
          g.z += __uint_as_float(h.y << 16); g.w += __uint_as_float(h.y & 0xFFFF0000u);
        }
        ge = g;
        if (de) { const float4 t = de[p]; ge = make_float4(g.x + t.x, g.y + t.y, g.z + t.z, g.w + t.w); }
      }
      oj[off + c] = g;
      oe[off + c] = ge;
    }
  }
}

}  // namespace gps_rowplan


extern "C" {

int gps_bert_embed_partial_rows(int n_rows) { return gps_bert_embed::grid_rows(n_rows); }

int gps_bert_embed_forward(int n_rows, int d, const long long *ids, const long long *pos, const float *word,
                           const float *pos_table, const float *type_row, const float *gamma, const float *beta, float eps,
                           float p_drop, unsigned long long seed, const void *seed_dev, float *y, void *y_bf16, float *mean,
                           float *rstd, const int *rows_dev, const int *poison_dev, gps_stream_t stream) {
  using namespace gps_bert_embed;
  if (n_rows < 0 || d < 1 || p_drop < 0.f || p_drop >= 1.f) return GPS_ERR_INVALID_ARGUMENT;
  if ((d & 255) || d > 1024) return GPS_ERR_UNSUPPORTED;
  if (n_rows == 0) return GPS_OK;
  if (!ids || !pos || !word || !pos_table || !type_row || !gamma || !beta || !y || !mean || !rstd) return GPS_ERR_INVALID_ARGUMENT;
  const uintptr_t align = (uintptr_t)word | (uintptr_t)pos_table | (uintptr_t)type_row | (uintptr_t)gamma | (uintptr_t)beta |
                          (uintptr_t)y | (uintptr_t)y_bf16;
  if (align & 15) return GPS_ERR_UNSUPPORTED;
  const unsigned int thr = p_drop > 0.f ? (unsigned int)((double)p_drop * 4294967296.0) : 0u;
  const dim3 grid(grid_rows(n_rows)), block(kBlock);
  hipStream_t s = (hipStream_t)stream;
#define GPS_BE_FWD(IT)                                                                                                     \
  hipLaunchKernelGGL((fwd_kernel<IT>), grid, block, 0, s, n_rows, d, (const int64_t *)ids, (const int64_t *)pos, word,   \
                     pos_table, type_row, gamma, beta, eps, p_drop, thr, seed, (const unsigned long long *)seed_dev, y, \
                     (uint16_t *)y_bf16, mean, rstd, rows_dev, poison_dev)
  switch (d >> 8) {
    case 1: GPS_BE_FWD(1); break;
    case 2: GPS_BE_FWD(2); break;
    case 3: GPS_BE_FWD(3); break;
    case 4: GPS_BE_FWD(4); break;
    default: return GPS_ERR_UNSUPPORTED;
  }
#undef GPS_BE_FWD
  return hipGetLastError() == hipSuccess ? GPS_OK : GPS_ERR_LAUNCH;
}

int gps_bert_position_grad(int n_seq, int n_pos, int d, const int *cu_rows, const float *dz, float *out, gps_stream_t stream) {
  using namespace gps_bert_embed;
  if (n_seq < 0 || n_pos < 0 || d < 1) return GPS_ERR_INVALID_ARGUMENT;
  if ((d & 255) || d > 1024) return GPS_ERR_UNSUPPORTED;
  if (n_pos == 0) return GPS_OK;
  if (!cu_rows || !dz || !out || (((uintptr_t)dz | (uintptr_t)out) & 15)) return GPS_ERR_INVALID_ARGUMENT;
  const size_t lds = sizeof(float) * kWaves * d;
  hipStream_t s = (hipStream_t)stream;
  switch (d >> 8) {
    case 1: hipLaunchKernelGGL((pos_grad_kernel<1>), dim3(n_pos), dim3(kBlock), lds, s, n_seq, d, cu_rows, dz, out); break;
    case 2: hipLaunchKernelGGL((pos_grad_kernel<2>), dim3(n_pos), dim3(kBlock), lds, s, n_seq, d, cu_rows, dz, out); break;
    case 3: hipLaunchKernelGGL((pos_grad_kernel<3>), dim3(n_pos), dim3(kBlock), lds, s, n_seq, d, cu_rows, dz, out); break;
    case 4: hipLaunchKernelGGL((pos_grad_kernel<4>), dim3(n_pos), dim3(kBlock), lds, s, n_seq, d, cu_rows, dz, out); break;
    default: return GPS_ERR_UNSUPPORTED;
  }
  return hipGetLastError() == hipSuccess ? GPS_OK : GPS_ERR_LAUNCH;
}

int gps_bert_embed_backward(int n_rows, int d, const float *dy, const void *dy_bf16, const long long *ids, const long long *pos,
                            const float *word, const float *pos_table, const float *type_row, const float *gamma,
                            const float *mean, const float *rstd, float p_drop, unsigned long long seed, const void *seed_dev,
                            float *dz, float *dgamma_part, float *dbeta_part, const int *rows_dev, gps_stream_t stream) {
  using namespace gps_bert_embed;
  if (n_rows < 0 || d < 1 || p_drop < 0.f || p_drop >= 1.f) return GPS_ERR_INVALID_ARGUMENT;
  if ((d & 255) || d > 1024) return GPS_ERR_UNSUPPORTED;
  if (n_rows == 0) return GPS_OK;
  if (!dy || !ids || !pos || !word || !pos_table || !type_row || !gamma || !mean || !rstd || !dz || !dgamma_part || !dbeta_part)
    return GPS_ERR_INVALID_ARGUMENT;
  const uintptr_t align = (uintptr_t)word | (uintptr_t)pos_table | (uintptr_t)type_row | (uintptr_t)gamma | (uintptr_t)dy |
                          (uintptr_t)dy_bf16 | (uintptr_t)dz;
  if (align & 15) return GPS_ERR_UNSUPPORTED;
  const unsigned int thr = p_drop > 0.f ? (unsigned int)((double)p_drop * 4294967296.0) : 0u;
  const dim3 grid(grid_rows(n_rows)), block(kBlock);
  const size_t lds = sizeof(float) * kWaves * d;
  hipStream_t s = (hipStream_t)stream;
#define GPS_BE_BWD(IT)                                                                                                       \
  hipLaunchKernelGGL((bwd_kernel<IT>), grid, block, lds, s, n_rows, d, dy, (const uint16_t *)dy_bf16, (const int64_t *)ids, \
                     (const int64_t *)pos, word, pos_table, type_row, gamma, mean, rstd, p_drop, thr, seed,                \
                     (const unsigned long long *)seed_dev, dz, dgamma_part, dbeta_part, rows_dev)
  switch (d >> 8) {
    case 1: GPS_BE_BWD(1); break;
    case 2: GPS_BE_BWD(2); break;
    case 3: GPS_BE_BWD(3); break;
    case 4: GPS_BE_BWD(4); break;
    default: return GPS_ERR_UNSUPPORTED;
  }
#undef GPS_BE_BWD
  return hipGetLastError() == hipSuccess ? GPS_OK : GPS_ERR_LAUNCH;
}

int gps_varlen_plan(const gps_varlen_text *texts, int n_texts, int n_seq_full, int *i32_out, long long *i64_out,
                    unsigned char *valid_out, gps_stream_t stream) {
  using namespace gps_bert_embed;
  if (n_texts < 1 || n_texts > GPS_VARLEN_MAX_TEXTS || !texts || !i32_out || !i64_out || !valid_out || n_seq_full < 0)
    return GPS_ERR_INVALID_ARGUMENT;
  PlanArgs A = {};
  int seq = 0;
  long long tok = 0;
  A.n_tok_full = 0;
  for (int i = 0; i < n_texts; ++i) {
    const gps_varlen_text &q = texts[i];
    if (q.n_seq < 1 || q.len < 1 || !q.ids || !q.mask) return GPS_ERR_INVALID_ARGUMENT;
    if (q.mask_elem_bytes != 1 && q.mask_elem_bytes != 2 && q.mask_elem_bytes != 4 && q.mask_elem_bytes != 8)
      return GPS_ERR_INVALID_ARGUMENT;
    A.t[i].ids = (const int64_t *)q.ids;
    A.t[i].mask = q.mask;
    A.t[i].elem_bytes = q.mask_elem_bytes;
    A.t[i].is_float = q.mask_is_float ? 1 : 0;
    A.t[i].n_seq = q.n_seq;
    A.t[i].len = q.len;
    A.t[i].seq0 = seq;
    A.t[i].tok0 = tok;
    if (seq == n_seq_full) A.n_tok_full = tok;
    seq += q.n_seq;
    tok += (long long)q.n_seq * q.len;
  }
  if (seq == n_seq_full) A.n_tok_full = tok;
  if (n_seq_full > seq) return GPS_ERR_INVALID_ARGUMENT;
  if (n_seq_full > 0 && n_seq_full < seq && A.n_tok_full == 0) return GPS_ERR_INVALID_ARGUMENT;   // not a text boundary
  if (seq > kPlanMaxSeq || tok > 0x7FFFFFFFll) return GPS_ERR_UNSUPPORTED;
  A.n_texts = n_texts;
  A.n_seq = seq;
  A.n_seq_full = n_seq_full;
  A.n_tok = tok;
  const size_t lds = sizeof(int) * (2 * (size_t)seq + 1);
  // lengths (+ the violation flags, parked in the q_limit slots), then one workgroup per sequence
  hipLaunchKernelGGL(varlen_lens_kernel, dim3((seq + kLensWaves - 1) / kLensWaves), dim3(64 * kLensWaves), 0, (hipStream_t)stream, A,
                     (int32_t *)i32_out, (int32_t *)i32_out + 3 * seq + 1);
  hipLaunchKernelGGL(varlen_plan_kernel, dim3(seq), dim3(kRowsThreads), lds, (hipStream_t)stream, A, (int32_t *)i32_out,
                     (int64_t *)i64_out, (uint8_t *)valid_out);
  return hipGetLastError() == hipSuccess ? GPS_OK : GPS_ERR_LAUNCH;
}


int gps_rows_move(int n, long long n_src_rows, long long n_dst_rows, int row_bytes, const void *src, const long long *src_idx,
                  void *dst, const long long *dst_idx, const int *n_live, int zero_dead, gps_stream_t stream) {
  if (n < 0 || n_src_rows < 1 || n_dst_rows < 1 || row_bytes < 16 || !src || !dst) return GPS_ERR_INVALID_ARGUMENT;
  if ((row_bytes & 15) || (((uintptr_t)src | (uintptr_t)dst) & 15)) return GPS_ERR_UNSUPPORTED;
  if (n == 0) return GPS_OK;
  const int blocks = (n + 3) / 4 < 4096 ? (n + 3) / 4 : 4096;
  hipLaunchKernelGGL(gps_rowplan::rows_move_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, n, n_src_rows, n_dst_rows,
                     row_bytes / 16, reinterpret_cast<const uint4 *>(src), src_idx, reinterpret_cast<uint4 *>(dst), dst_idx, n_live,
                     zero_dead);
  return hipGetLastError() == hipSuccess ? GPS_OK : GPS_ERR_LAUNCH;
}

int gps_rows_pack2(int n_seq, int len_a, int len_b, int d, const float *a, const float *b, const long long *perm, const int *n_live,
                   float *out, unsigned short *out16, gps_stream_t stream) {
  if (n_seq < 1 || len_a < 1 || len_b < 1 || d < 4 || !a || !b || !perm || !out) return GPS_ERR_INVALID_ARGUMENT;
  if (d % 4 || (long long)n_seq * (len_a + len_b) > (1 << 22)) return GPS_ERR_UNSUPPORTED;
  const int n = n_seq * (len_a + len_b);
  const int blocks = (n + 3) / 4 < 4096 ? (n + 3) / 4 : 4096;
  hipLaunchKernelGGL(gps_rowplan::rows_pack2_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, n, d / 4, len_a + len_b, len_a,
                     reinterpret_cast<const float4 *>(a), reinterpret_cast<const float4 *>(b), perm, n_live,
                     reinterpret_cast<float4 *>(out), reinterpret_cast<uint2 *>(out16));
  return hipGetLastError() == hipSuccess ? GPS_OK : GPS_ERR_LAUNCH;
}

int gps_rows_unpack2(int n_seq, int len_a, int len_b, int d, const float *packed, const long long *inv, const unsigned char *valid,
                     float *out_a, float *out_b, gps_stream_t stream) {
  if (n_seq < 1 || len_a < 1 || len_b < 1 || d < 4 || !packed || !inv || !valid || !out_a || !out_b) return GPS_ERR_INVALID_ARGUMENT;
  if (d % 4 || (long long)n_seq * (len_a + len_b) > (1 << 22)) return GPS_ERR_UNSUPPORTED;
  const int n = n_seq * (len_a + len_b);
  const int blocks = (n + 3) / 4 < 4096 ? (n + 3) / 4 : 4096;
  hipLaunchKernelGGL(gps_rowplan::rows_unpack2_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, n, d / 4, len_a + len_b, len_a,
                     reinterpret_cast<const float4 *>(packed), inv, valid, reinterpret_cast<float4 *>(out_a),
                     reinterpret_cast<float4 *>(out_b));
  return hipGetLastError() == hipSuccess ? GPS_OK : GPS_ERR_LAUNCH;
}

int gps_joint_embed_forward(int n_seq, int len_a, int len_b, int d, const float *a, const float *b, const float *ea, const float *eb,
                            const long long *perm, const int *n_live, float *x, float *e, unsigned short *x16, gps_stream_t stream) {
  if (n_seq < 1 || len_a < 1 || len_b < 1 || d < 4 || !a || !b || !ea || !eb || !perm || !x || !e || !x16) return GPS_ERR_INVALID_ARGUMENT;
  if (d % 4 || (long long)n_seq * (len_a + len_b) > (1 << 22)) return GPS_ERR_UNSUPPORTED;
  const int n = n_seq * (len_a + len_b);
  const int blocks = (n + 3) / 4 < 4096 ? (n + 3) / 4 : 4096;
  hipLaunchKernelGGL(gps_rowplan::joint_embed_fwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, n, d / 4, len_a + len_b,
                     len_a, reinterpret_cast<const float4 *>(a), reinterpret_cast<const float4 *>(b),
                     reinterpret_cast<const float4 *>(ea), reinterpret_cast<const float4 *>(eb), perm, n_live,
                     reinterpret_cast<float4 *>(x), reinterpret_cast<float4 *>(e), reinterpret_cast<uint2 *>(x16));
  return hipGetLastError() == hipSuccess ? GPS_OK : GPS_ERR_LAUNCH;
}

int gps_joint_embed_backward(int n_seq, int len_a, int len_b, int d, const float *dx, const unsigned short *dx16, const float *de,
                             const long long *inv, const unsigned char *valid, float *da, float *db, float *dea, float *deb,
                             gps_stream_t stream) {
  if (n_seq < 1 || len_a < 1 || len_b < 1 || d < 4 || !inv || !valid || !da || !db || !dea || !deb) return GPS_ERR_INVALID_ARGUMENT;
  if (d % 4 || (long long)n_seq * (len_a + len_b) > (1 << 22)) return GPS_ERR_UNSUPPORTED;
  const int n = n_seq * (len_a + len_b);
  const int blocks = (n + 3) / 4 < 4096 ? (n + 3) / 4 : 4096;
  hipLaunchKernelGGL(gps_rowplan::joint_embed_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, n, d / 4, len_a + len_b,
                     len_a, reinterpret_cast<const float4 *>(dx), reinterpret_cast<const uint2 *>(dx16),
                     reinterpret_cast<const float4 *>(de), inv, valid, reinterpret_cast<float4 *>(da), reinterpret_cast<float4 *>(db),
                     reinterpret_cast<float4 *>(dea), reinterpret_cast<float4 *>(deb));
  return hipGetLastError() == hipSuccess ? GPS_OK : GPS_ERR_LAUNCH;
}

int gps_rows_plan(int n_seq, int seq_len, const unsigned char *valid, long long *perm, long long *inv, int *cu, int *n_live,
                  gps_stream_t stream) {
  if (n_seq < 1 || seq_len < 1 || !valid || !perm || !inv || !cu || !n_live) return GPS_ERR_INVALID_ARGUMENT;
  if ((long long)n_seq * seq_len > (1 << 22)) return GPS_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(gps_rowplan::plan_kernel, dim3(1), dim3(gps_rowplan::kThreads), 0, (hipStream_t)stream, n_seq, seq_len,
                     valid, perm, inv, cu, n_live);
  return hipGetLastError() == hipSuccess ? GPS_OK : GPS_ERR_LAUNCH;
}


}  // extern "C"
