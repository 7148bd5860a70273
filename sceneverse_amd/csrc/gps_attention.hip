// gps_attention.hip -- fused multi-head self-attention core for the GPS object/joint transformers on
// MI355X (gfx950): QK^T, language-conditioned pairwise-spatial bias, key-padding mask, softmax,
// attention dropout and PV in one launch; backward in one launch.  bf16 in/out, fp32 accumulate on
// v_mfma_f32_16x16x32_bf16.
//
// Reference behaviour restated:
//   modules/layers/transformers.py:193-239  MultiHeadAttentionSpatial.forward, fusion 'cond':
//       attn = q k^T / sqrt(d_h);  w = lang_cond_fc(x) -> per (token, head): bias b, weights w_1..5
//       loc = sigmoid(w . pairwise[b,l,t,:] + b);  masked keys: attn = -inf, loc = 0
//       probs = softmax(log(clamp(loc, 1e-6)) + attn);  out = probs v
//   modules/layers/transformers.py:141 (nn.MultiheadAttention with key_padding_mask, dropout on the
//       probabilities) -- the same core without the spatial term.
// The reference materialises ~12 (B,H,L,L) fp32 tensors per layer for this;
// here nothing of size L x L ever reaches HBM.
//
// Tiny-sequence design (L = 80 objects, 130 joint tokens; d_h = 64): one workgroup per (batch, head),
// one wave per 16-row strip.  K (row-major) and V^T live in LDS; scores are computed TRANSPOSED
// (S^T = K Q^T) so that a lane owns one query column: the softmax row reduction is in-lane plus two
// cross-row shuffles, and the D fragments of S^T are directly the A fragments of the P V product
// (the K index of that product is permuted to the D row order, and V^T is read in the same order).
// The pairwise tensor (B,L,L,5) is shared by the 12 heads of a scene: the block -> (b,h) map sends
// all heads of a scene to the same XCD so that it is served from that XCD's L2.
//
// Backward, two passes in one launch: pass 1 (query strips, transposed scores) produces dQ, the
// gradient of the conditioning vector (lang_cond_fc output) and delta = rowsum(P dP); pass 2 (key
// strips, plain orientation so that a lane owns one key) recomputes the probabilities and produces
// dK and dV.  Probabilities are recomputed from the saved log-sum-exp, never stored.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "gps_hip.h"
#include "gps_device_flags.h"
#include "gps_attention_ex.h"

namespace gps_attn {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

constexpr int DH = 64;        // head width (the only one the GPS configs use)
constexpr int KS = DH + 8;    // LDS row pitch of row-major tiles, bf16 elements (144 B: conflict-free b128)
constexpr int SD = 6;         // conditioning vector per (token, head): bias + 5 weights

struct Params {
  int B, H, L, nt;            // L = number of KEYS (= queries in self-attention), nt = ceil(L / 16)
  int Lq, ntq;                // number of QUERIES (cross-attention: q from `tgt`, k / v from `memory`, Lq != L allowed)
  int ld_qkv, ld_o;           // row pitches in elements: k, v (and q, dq, dk, dv in self-attention) | out, dout
  int ld_q, ld_dq, ld_dkv;    // pitches of q | dq | dk, dv (streaming kernels; = ld_qkv in the packed self-attention call)
  const uint16_t *q, *k, *v;  // (B, Lq, ld_q) and (B, L, ld_qkv) views, head h at column h * 64
  const float *sw;            // (B, L, H * 6) or null
  const float *pl;            // (B, L, L, 5)  or null
  const uint8_t *mask;        // (B, L), 1 = padded key, or null
  uint16_t *out;              // (B, L, ld_o)
  float *lse;                 // (B, H, L)
  // backward only
  const uint16_t *dout;       // (B, L, ld_o)
  uint16_t *dq, *dk, *dv;     // (B, L, ld_qkv) views
  float *dsw;                 // (B, L, H * 6)
  float p_drop;
  unsigned int drop_thr;      // keep iff rng >= drop_thr
  unsigned long long seed;
  const unsigned long long *seed_dev;   // optional device word added to `seed` (HIP-graph replays
                                        // get fresh dropout masks by advancing it on the device)
  const int *seq_order;                 // optional (B) permutation: workgroups visit the sequences in this order
                                        // (longest first: the dispatcher hands out workgroups in block order, so the
                                        // long ones start early and the short ones fill the tail)
  const int *q_limit;                   // optional (B), with cu_rows: number of leading QUERY rows of a sequence that are computed
  const int *cu_rows;                   // optional (B + 1) row offsets of VARIABLE-LENGTH sequences packed back to back
                                        // (self-attention, streaming kernels): sequence b = rows [cu[b], cu[b + 1]),
                                        // L / Lq are then the CAPACITY (longest sequence; LDS sizing, lse pitch)
};

__device__ __forceinline__ unsigned long long effective_seed(const Params &P) {
  return P.seed + (P.seed_dev ? *P.seed_dev : 0ull);
}

__device__ __forceinline__ uint16_t f2bf(float f) {   // round to nearest even (hardware conversion)
  return __builtin_bit_cast(uint16_t, (__bf16)f);
}
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
__device__ __forceinline__ unsigned int pack2(float lo, float hi) {     // v_cvt_pk_bf16_f32: round to nearest even
  const bf16x2_t h = {(__bf16)lo, (__bf16)hi};
  return __builtin_bit_cast(unsigned int, h);
}
__device__ __forceinline__ bf16x8 as_frag(u32x4 v) { return __builtin_bit_cast(bf16x8, v); }
__device__ __forceinline__ u32x4 zero4() { u32x4 z = {0u, 0u, 0u, 0u}; return z; }

__device__ __forceinline__ f32x4 mfma(bf16x8 a, bf16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// counter-based RNG for attention dropout (splitmix64 finaliser): the same (seed, element) gives the
// same keep decision in forward and backward
// Counter-based dropout stream of the attention kernels: a 32-bit avalanche hash (two multiplies, three
// xor-shifts; "lowbias32" constants) of the element (or key-pair) index folded with the seed.  All that matters is
// that forward and backward draw the same bits for the same (seed, index) and that the keep rate is 1 - p; a
// 64-bit splitmix per element cost ~30 vector instructions of the ~45 these kernels spent per score.
__device__ __forceinline__ unsigned int mix32(unsigned int x) {
  x ^= x >> 16;
  x *= 0x21F0AAADu;
  x ^= x >> 15;
  x *= 0x735A2D97u;
  x ^= x >> 15;
  return x;
}
__device__ __forceinline__ unsigned int seed_fold(unsigned long long seed) {      // wave-uniform
  return mix32((unsigned int)seed ^ mix32((unsigned int)(seed >> 32) + 0x9E3779B9u));
}
__device__ __forceinline__ unsigned int rng_u32(unsigned long long seed, unsigned long long idx) {
  return mix32(((unsigned int)idx + (unsigned int)(idx >> 32) * 0x85EBCA6Bu) ^ seed_fold(seed));
}

__device__ __forceinline__ void block_to_bh(const Params &P, int &b, int &h) {
  const int id = blockIdx.x;
  if ((P.B & 7) == 0) {       // heads of one scene on one XCD (block id mod 8), scenes spread over XCDs
    const int xcd = id & 7, slot = id >> 3;
    b = (slot / P.H) * 8 + xcd;
    h = slot % P.H;
  } else {
    b = id / P.H;
    h = id % P.H;
  }
}

// rows [0, rows_total) of a (.., ld) bf16 matrix (64 columns of head h) -> LDS [rows_total][KS];
// rows >= rows_valid are zero
__device__ __forceinline__ void stage_rows(uint16_t *dst, const uint16_t *src, int ld, int rows_valid,
                                           int rows_total) {
  for (int e = threadIdx.x; e < rows_total * 8; e += blockDim.x) {
    const int r = e >> 3, ch = e & 7;
    u32x4 v = zero4();
    if (r < rows_valid) v = *reinterpret_cast<const u32x4 *>(src + (size_t)r * ld + ch * 8);
    *reinterpret_cast<u32x4 *>(dst + r * KS + ch * 8) = v;
  }
}
// the same rows transposed -> LDS [64][ts] (column t of row d), zero for t >= rows_valid
__device__ __forceinline__ void stage_transposed(uint16_t *dst, const uint16_t *src, int ld,
                                                 int rows_valid, int rows_total, int ts) {
  for (int e = threadIdx.x; e < rows_total * 8; e += blockDim.x) {
    const int r = e >> 3, ch = e & 7;
    u32x4 v = zero4();
    if (r < rows_valid) v = *reinterpret_cast<const u32x4 *>(src + (size_t)r * ld + ch * 8);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      dst[(ch * 8 + 2 * i) * ts + r] = (uint16_t)(v[i] & 0xFFFFu);
      dst[(ch * 8 + 2 * i + 1) * ts + r] = (uint16_t)(v[i] >> 16);
    }
  }
}
// B fragment of a product whose K index runs over rows of the transposed LDS tile `t` (pitch ts):
// lane (n = lane & 15, g = lane >> 4) gets rows {32c + 4g + 0..3, 32c + 16 + 4g + 0..3} of column
// 16 * ntile + n -- the K order in which the D fragments of two adjacent 16-row tiles are packed.
__device__ __forceinline__ bf16x8 frag_from_transposed(const uint16_t *t, int ts, int ntile, int c,
                                                       int lane) {
  const uint16_t *p = t + (16 * ntile + (lane & 15)) * ts + 32 * c + 4 * (lane >> 4);
  const u32x2 lo = *reinterpret_cast<const u32x2 *>(p);
  const u32x2 hi = *reinterpret_cast<const u32x2 *>(p + 16);
  u32x4 v = {lo[0], lo[1], hi[0], hi[1]};
  return as_frag(v);
}
// pack the D fragments of two adjacent 16-row tiles into one A fragment (same K order as above)
__device__ __forceinline__ bf16x8 pack_tiles(const f32x4 &a, const f32x4 &b) {
  u32x4 v = {pack2(a[0], a[1]), pack2(a[2], a[3]), pack2(b[0], b[1]), pack2(b[2], b[3])};
  return as_frag(v);
}

__device__ __forceinline__ float xor_reduce_max_rows(float v) {   // across the 4 lane groups
  v = fmaxf(v, __shfl_xor(v, 16, 64));
  return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float xor_reduce_sum_rows(float v) {
  v += __shfl_xor(v, 16, 64);
  return v + __shfl_xor(v, 32, 64);
}
// spatial term of one (query, key) pair: z = w0 + sum_d w_d pl_d; sig = sigmoid(z);
// returns log(clamp(sig, 1e-6)) (masked keys: log(1e-6)), and sig through `sig`.
__device__ __forceinline__ float spatial_bias(const float *__restrict__ plp, const float (&w)[SD],
                                              bool key_masked, float &sig) {
  float z = w[0];
#pragma unroll
  for (int d = 0; d < 5; ++d) z = fmaf(w[1 + d], plp[d], z);
  sig = key_masked ? 0.f : __builtin_amdgcn_rcpf(1.f + __expf(-z));   // v_rcp_f32 (1 ulp): `1.f / x` is a 10-instruction IEEE division
  return __logf(fmaxf(sig, 1e-6f));
}

// ==========================================================================================
// forward
// ==========================================================================================
template <int NT, bool SPATIAL>
__global__ __launch_bounds__(512) void attn_fwd_kernel(const Params P) {
  constexpr int NC = (NT + 1) / 2;          // 32-key chunks of the P V product
  constexpr int TS = NC * 32 + 8;           // pitch of the transposed V tile
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint16_t *Ks = reinterpret_cast<uint16_t *>(smem);   // [NT*16][KS]
  uint16_t *Vt = Ks + NT * 16 * KS;                     // [64][TS]
  uint8_t *s_mask = reinterpret_cast<uint8_t *>(Vt + 64 * TS);   // [NT*16] key-padding flags

  int b, h;
  block_to_bh(P, b, h);
  const int L = P.L, nt = P.nt;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  const int m = lane & 15, g = lane >> 4;
  const size_t row0 = (size_t)b * L;
  const uint16_t *qb = P.q + row0 * P.ld_qkv + h * DH;
  const uint16_t *kb = P.k + row0 * P.ld_qkv + h * DH;
  const uint16_t *vb = P.v + row0 * P.ld_qkv + h * DH;
  stage_rows(Ks, kb, P.ld_qkv, L, NT * 16);
  stage_transposed(Vt, vb, P.ld_qkv, L, NC * 32, TS);
  for (int t = threadIdx.x; t < NT * 16; t += blockDim.x)
    s_mask[t] = (t < L && P.mask) ? P.mask[row0 + t] : 0;
  __syncthreads();

  const bool dropout = P.drop_thr != 0u;
  const float keep_scale = dropout ? 1.f / (1.f - P.p_drop) : 1.f;
  const unsigned long long seed = dropout ? effective_seed(P) : 0ull;

  for (int s = wave; s < nt; s += nwaves) {
    const int qi = 16 * s + m;              // this lane's query
    const bool q_ok = qi < L;
    bf16x8 bq[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      u32x4 v = zero4();
      if (q_ok) v = *reinterpret_cast<const u32x4 *>(qb + (size_t)qi * P.ld_qkv + 32 * c + 8 * g);
      bq[c] = as_frag(v);
    }
    // S^T tiles: acc[j][r] = <q_qi, k_t>, t = 16 j + 4 g + r
    f32x4 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (j < nt) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const u32x4 a = *reinterpret_cast<const u32x4 *>(Ks + (16 * j + m) * KS + 32 * c + 8 * g);
          acc[j] = mfma(as_frag(a), bq[c], acc[j]);
        }
      }
    }
    float w[SD];
    if (SPATIAL) {
#pragma unroll
      for (int d = 0; d < SD; ++d)
        w[d] = q_ok ? P.sw[((row0 + qi) * P.H + h) * SD + d] : 0.f;
    }
    // logits, row max
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int t = 16 * j + 4 * g + r;
        const bool t_ok = (j < nt) && t < L;
        const bool km = t_ok && s_mask[t];
        float x = acc[j][r] * 0.125f;
        if (SPATIAL && t_ok && q_ok) {
          float sig;
          x += spatial_bias(P.pl + ((row0 + qi) * L + t) * 5, w, km, sig);
        }
        if (!t_ok || km) x = -INFINITY;
        acc[j][r] = x;
        mx = fmaxf(mx, x);
      }
    }
    mx = xor_reduce_max_rows(mx);
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = __expf(acc[j][r] - mx);   // exp(-inf - mx) = 0; all keys masked -> NaN like torch
        acc[j][r] = p;
        sum += p;
      }
    sum = xor_reduce_sum_rows(sum);
    const float inv = 1.f / sum;
    if (g == 0 && q_ok) P.lse[((size_t)b * P.H + h) * L + qi] = mx + __logf(sum);
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float p = acc[j][r] * inv;
        if (dropout) {
          const int t = 16 * j + 4 * g + r;
          const unsigned long long idx = (((unsigned long long)b * P.H + h) * L + qi) * L + t;
          p = rng_u32(seed, idx) >= P.drop_thr ? p * keep_scale : 0.f;
        }
        acc[j][r] = p;
      }
    // O strip = P V
    f32x4 o[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) o[n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      if (2 * c < nt) {
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        const bf16x8 pa = pack_tiles(acc[2 * c], (2 * c + 1 < NT) ? acc[(2 * c + 1 < NT) ? 2 * c + 1 : 0] : z);
#pragma unroll
        for (int n = 0; n < 4; ++n) o[n] = mfma(pa, frag_from_transposed(Vt, TS, n, c, lane), o[n]);
      }
    }
    // o[n][r] = O[query 16 s + 4 g + r][d = 16 n + m]
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int qr = 16 * s + 4 * g + r;
      if (qr < L) {
        uint16_t *op = P.out + (row0 + qr) * P.ld_o + h * DH + m;
#pragma unroll
        for (int n = 0; n < 4; ++n) op[16 * n] = f2bf(o[n][r]);
      }
    }
  }
}

// ==========================================================================================
// backward
// ==========================================================================================
template <int NT, bool SPATIAL>
__global__ __launch_bounds__(512) void attn_bwd_recompute_kernel(const Params P) {
  constexpr int NC = (NT + 1) / 2;
  constexpr int TS = NC * 32 + 8;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // pass 1: Ks [NT*16][KS] | Vs [NT*16][KS] | Kt [64][TS];  pass 2 (aliased): Qt [64][TS] | dOt [64][TS]
  uint16_t *Ks = reinterpret_cast<uint16_t *>(smem);
  uint16_t *Vs = Ks + NT * 16 * KS;
  uint16_t *Kt = Vs + NT * 16 * KS;
  uint16_t *Qt = reinterpret_cast<uint16_t *>(smem);
  uint16_t *dOt = Qt + 64 * TS;
  constexpr int kBig = (2 * NT * 16 * KS + 64 * TS) * 2;   // bytes of the aliased region
  float *delta_s = reinterpret_cast<float *>(smem + kBig);   // [NT*16]

  int b, h;
  block_to_bh(P, b, h);
  const int L = P.L, nt = P.nt;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  const int m = lane & 15, g = lane >> 4;
  const size_t row0 = (size_t)b * L;
  const uint16_t *qb = P.q + row0 * P.ld_qkv + h * DH;
  const uint16_t *kb = P.k + row0 * P.ld_qkv + h * DH;
  const uint16_t *vb = P.v + row0 * P.ld_qkv + h * DH;
  const uint16_t *dob = P.dout + row0 * P.ld_o + h * DH;
  const float *lse = P.lse + ((size_t)b * P.H + h) * L;
  const bool dropout = P.drop_thr != 0u;
  const float keep_scale = dropout ? 1.f / (1.f - P.p_drop) : 1.f;
  const unsigned long long seed = dropout ? effective_seed(P) : 0ull;

  stage_rows(Ks, kb, P.ld_qkv, L, NT * 16);
  stage_rows(Vs, vb, P.ld_qkv, L, NT * 16);
  stage_transposed(Kt, kb, P.ld_qkv, L, NC * 32, TS);
  __syncthreads();

  // ---------------- pass 1: query strips, transposed scores -> dQ, dsw, delta ----------------
  for (int s = wave; s < nt; s += nwaves) {
    const int qi = 16 * s + m;
    const bool q_ok = qi < L;
    bf16x8 bq[2], bdo[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      u32x4 v = zero4(), u = zero4();
      if (q_ok) {
        v = *reinterpret_cast<const u32x4 *>(qb + (size_t)qi * P.ld_qkv + 32 * c + 8 * g);
        u = *reinterpret_cast<const u32x4 *>(dob + (size_t)qi * P.ld_o + 32 * c + 8 * g);
      }
      bq[c] = as_frag(v);
      bdo[c] = as_frag(u);
    }
    f32x4 acc[NT], dacc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      dacc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (j < nt) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const u32x4 a = *reinterpret_cast<const u32x4 *>(Ks + (16 * j + m) * KS + 32 * c + 8 * g);
          const u32x4 av = *reinterpret_cast<const u32x4 *>(Vs + (16 * j + m) * KS + 32 * c + 8 * g);
          acc[j] = mfma(as_frag(a), bq[c], acc[j]);       // S^T
          dacc[j] = mfma(as_frag(av), bdo[c], dacc[j]);   // (dO V^T)^T
        }
      }
    }
    float w[SD];
    if (SPATIAL) {
#pragma unroll
      for (int d = 0; d < SD; ++d) w[d] = q_ok ? P.sw[((row0 + qi) * P.H + h) * SD + d] : 0.f;
    }
    const float lse_q = q_ok ? lse[qi] : 0.f;
    // p, gate (1 - sig where the spatial term has a gradient), dP
    f32x4 gate[SPATIAL ? NT : 1];
    float delta = 0.f;
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int t = 16 * j + 4 * g + r;
        const bool t_ok = (j < nt) && t < L && q_ok;
        const bool km = t_ok && P.mask && P.mask[row0 + t];
        float x = acc[j][r] * 0.125f;
        float gt = 0.f;
        if (SPATIAL && t_ok) {
          float sig;
          x += spatial_bias(P.pl + ((row0 + qi) * L + t) * 5, w, km, sig);
          gt = sig > 1e-6f ? 1.f - sig : 0.f;     // d/dz log(clamp(sigmoid(z), 1e-6))
        }
        if (SPATIAL) gate[SPATIAL ? j : 0][r] = gt;
        const float p = (t_ok && !km) ? __expf(x - lse_q) : 0.f;
        float dp = dacc[j][r];
        if (dropout) {
          const unsigned long long idx = (((unsigned long long)b * P.H + h) * L + qi) * L + t;
          dp = rng_u32(seed, idx) >= P.drop_thr ? dp * keep_scale : 0.f;
        }
        acc[j][r] = p;
        dacc[j][r] = dp;
        delta += p * dp;
      }
    delta = xor_reduce_sum_rows(delta);
    if (g == 0) delta_s[16 * s + m] = delta;
    float dw[SD];
#pragma unroll
    for (int d = 0; d < SD; ++d) dw[d] = 0.f;
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float dlogit = acc[j][r] * (dacc[j][r] - delta);
        if (SPATIAL) {
          const int t = 16 * j + 4 * g + r;
          const float dz = dlogit * gate[SPATIAL ? j : 0][r];
          if (dz != 0.f) {
            const float *plp = P.pl + ((row0 + qi) * L + t) * 5;
            dw[0] += dz;
#pragma unroll
            for (int d = 0; d < 5; ++d) dw[1 + d] = fmaf(dz, plp[d], dw[1 + d]);
          }
        }
        acc[j][r] = dlogit * 0.125f;            // dS
      }
    if (SPATIAL) {
#pragma unroll
      for (int d = 0; d < SD; ++d) dw[d] = xor_reduce_sum_rows(dw[d]);
      if (g == 0 && q_ok) {
#pragma unroll
        for (int d = 0; d < SD; ++d) P.dsw[((row0 + qi) * P.H + h) * SD + d] = dw[d];
      }
    }
    // dQ strip = dS K
    f32x4 o[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) o[n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      if (2 * c < nt) {
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        const bf16x8 da = pack_tiles(acc[2 * c], (2 * c + 1 < NT) ? acc[(2 * c + 1 < NT) ? 2 * c + 1 : 0] : z);
#pragma unroll
        for (int n = 0; n < 4; ++n) o[n] = mfma(da, frag_from_transposed(Kt, TS, n, c, lane), o[n]);
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int qr = 16 * s + 4 * g + r;
      if (qr < L) {
        uint16_t *op = P.dq + (row0 + qr) * P.ld_qkv + h * DH + m;
#pragma unroll
        for (int n = 0; n < 4; ++n) op[16 * n] = f2bf(o[n][r]);
      }
    }
  }
  __syncthreads();   // every strip's delta is in LDS; K/V tiles no longer needed
  stage_transposed(Qt, qb, P.ld_qkv, L, NC * 32, TS);
  stage_transposed(dOt, dob, P.ld_o, L, NC * 32, TS);
  __syncthreads();

  // ---------------- pass 2: key strips, plain orientation -> dK, dV ----------------
  for (int js = wave; js < nt; js += nwaves) {
    const int t = 16 * js + m;            // this lane's key
    const bool t_ok = t < L;
    const bool km = t_ok && P.mask && P.mask[row0 + t];
    bf16x8 bk[2], bv[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      u32x4 v = zero4(), u = zero4();
      if (t_ok) {
        v = *reinterpret_cast<const u32x4 *>(kb + (size_t)t * P.ld_qkv + 32 * c + 8 * g);
        u = *reinterpret_cast<const u32x4 *>(vb + (size_t)t * P.ld_qkv + 32 * c + 8 * g);
      }
      bk[c] = as_frag(v);
      bv[c] = as_frag(u);
    }
    f32x4 dk[4], dv[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      dk[n] = f32x4{0.f, 0.f, 0.f, 0.f};
      dv[n] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    for (int c = 0; c < NC; ++c) {        // 32-query chunks
      f32x4 pt[2], ds[2];
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        const int i = 2 * c + hh;          // query tile
        pt[hh] = f32x4{0.f, 0.f, 0.f, 0.f};
        ds[hh] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (i < nt) {
          const int qa = 16 * i + m;       // A-fragment row of this lane
          f32x4 sacc = {0.f, 0.f, 0.f, 0.f}, dacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int cc = 0; cc < 2; ++cc) {
            u32x4 v = zero4(), u = zero4();
            if (qa < L) {
              v = *reinterpret_cast<const u32x4 *>(qb + (size_t)qa * P.ld_qkv + 32 * cc + 8 * g);
              u = *reinterpret_cast<const u32x4 *>(dob + (size_t)qa * P.ld_o + 32 * cc + 8 * g);
            }
            sacc = mfma(as_frag(v), bk[cc], sacc);    // S[query 16 i + 4 g + r][key t]
            dacc = mfma(as_frag(u), bv[cc], dacc);    // dO V^T
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int qi = 16 * i + 4 * g + r;
            const bool ok = t_ok && qi < L;
            float x = sacc[r] * 0.125f;
            if (SPATIAL && ok) {
              float w[SD];
#pragma unroll
              for (int d = 0; d < SD; ++d) w[d] = P.sw[((row0 + qi) * P.H + h) * SD + d];
              float sig;
              x += spatial_bias(P.pl + ((row0 + qi) * L + t) * 5, w, km, sig);
            }
            const float p = (ok && !km) ? __expf(x - lse[qi]) : 0.f;
            float dp = dacc[r], pd = p;
            if (dropout) {
              const unsigned long long idx = (((unsigned long long)b * P.H + h) * L + qi) * L + t;
              const bool keep = rng_u32(seed, idx) >= P.drop_thr;
              dp = keep ? dp * keep_scale : 0.f;
              pd = keep ? p * keep_scale : 0.f;
            }
            const float dl = ok ? p * (dp - delta_s[qi]) : 0.f;
            pt[hh][r] = pd;
            ds[hh][r] = dl * 0.125f;
          }
        }
      }
      const bf16x8 pa = pack_tiles(pt[0], pt[1]);
      const bf16x8 da = pack_tiles(ds[0], ds[1]);
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        dv[n] = mfma(pa, frag_from_transposed(dOt, TS, n, c, lane), dv[n]);
        dk[n] = mfma(da, frag_from_transposed(Qt, TS, n, c, lane), dk[n]);
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int tr = 16 * js + 4 * g + r;
      if (tr < L) {
        uint16_t *pk = P.dk + (row0 + tr) * P.ld_qkv + h * DH + m;
        uint16_t *pv = P.dv + (row0 + tr) * P.ld_qkv + h * DH + m;
#pragma unroll
        for (int n = 0; n < 4; ++n) {
          pk[16 * n] = f2bf(dk[n][r]);
          pv[16 * n] = f2bf(dv[n][r]);
        }
      }
    }
  }
}


// ==========================================================================================
// backward, LDS-resident probabilities (L <= 144): pass 1 as above, but it also leaves the dropped
// probabilities P^T and dS^T (bf16, [key][query]) in LDS; pass 2 is then pure MFMA --
//   dV[t][d] = sum_m P^T[t][m] dO[m][d],   dK[t][d] = sum_m dS^T[t][m] Q[m][d]
// with A fragments read row-wise from those tiles and B fragments from the transposed dO / Q tiles:
// no recomputation, no global loads, standard K order.
// LDS: region A = Ks | Vs | Kt (pass 1), aliased by Qt | dOt (pass 2); region B = PT | dST.
// ==========================================================================================
template <int NT, bool SPATIAL>
__global__ __launch_bounds__(512) void attn_bwd_kernel(const Params P) {
  constexpr int NC = (NT + 1) / 2;
  constexpr int TS = NC * 32 + 8;             // pitch of transposed tiles and of PT / dST
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint16_t *Ks = reinterpret_cast<uint16_t *>(smem);
  uint16_t *Vs = Ks + NT * 16 * KS;
  uint16_t *Kt = Vs + NT * 16 * KS;
  uint16_t *Qt = reinterpret_cast<uint16_t *>(smem);
  uint16_t *dOt = Qt + 64 * TS;
  constexpr int kRegionA = 2 * NT * 16 * KS + 64 * TS;       // elements
  uint16_t *PT = reinterpret_cast<uint16_t *>(smem) + kRegionA;   // [NT*16][TS]
  uint16_t *dST = PT + NT * 16 * TS;
  uint8_t *s_mask = reinterpret_cast<uint8_t *>(dST + NT * 16 * TS);   // [NT*16] key-padding flags

  int b, h;
  block_to_bh(P, b, h);
  const int L = P.L, nt = P.nt;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  const int m = lane & 15, g = lane >> 4;
  const size_t row0 = (size_t)b * L;
  const uint16_t *qb = P.q + row0 * P.ld_qkv + h * DH;
  const uint16_t *kb = P.k + row0 * P.ld_qkv + h * DH;
  const uint16_t *vb = P.v + row0 * P.ld_qkv + h * DH;
  const uint16_t *dob = P.dout + row0 * P.ld_o + h * DH;
  const float *lse = P.lse + ((size_t)b * P.H + h) * L;
  const bool dropout = P.drop_thr != 0u;
  const float keep_scale = dropout ? 1.f / (1.f - P.p_drop) : 1.f;
  const unsigned long long seed = dropout ? effective_seed(P) : 0ull;

  stage_rows(Ks, kb, P.ld_qkv, L, NT * 16);
  stage_rows(Vs, vb, P.ld_qkv, L, NT * 16);
  stage_transposed(Kt, kb, P.ld_qkv, L, NC * 32, TS);
  {  // pad columns [nt*16, NC*32) of PT / dST are read by the last chunk but written by no strip
    u32x4 *z = reinterpret_cast<u32x4 *>(PT);
    for (int e = threadIdx.x; e < (2 * NT * 16 * TS) / 8; e += blockDim.x) z[e] = zero4();
    for (int t = threadIdx.x; t < NT * 16; t += blockDim.x)
      s_mask[t] = (t < L && P.mask) ? P.mask[row0 + t] : 0;
  }
  __syncthreads();

  // ---------------- pass 1: query strips -> dQ, d cond-vector, P^T and dS^T tiles ----------------
  for (int s = wave; s < nt; s += nwaves) {
    const int qi = 16 * s + m;
    const bool q_ok = qi < L;
    bf16x8 bq[2], bdo[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      u32x4 v = zero4(), u = zero4();
      if (q_ok) {
        v = *reinterpret_cast<const u32x4 *>(qb + (size_t)qi * P.ld_qkv + 32 * c + 8 * g);
        u = *reinterpret_cast<const u32x4 *>(dob + (size_t)qi * P.ld_o + 32 * c + 8 * g);
      }
      bq[c] = as_frag(v);
      bdo[c] = as_frag(u);
    }
    f32x4 acc[NT], dacc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      dacc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (j < nt) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const u32x4 a = *reinterpret_cast<const u32x4 *>(Ks + (16 * j + m) * KS + 32 * c + 8 * g);
          const u32x4 av = *reinterpret_cast<const u32x4 *>(Vs + (16 * j + m) * KS + 32 * c + 8 * g);
          acc[j] = mfma(as_frag(a), bq[c], acc[j]);
          dacc[j] = mfma(as_frag(av), bdo[c], dacc[j]);
        }
      }
    }
    float w[SD];
    if (SPATIAL) {
#pragma unroll
      for (int d = 0; d < SD; ++d) w[d] = q_ok ? P.sw[((row0 + qi) * P.H + h) * SD + d] : 0.f;
    }
    const float lse_q = q_ok ? lse[qi] : 0.f;
    float delta = 0.f;
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int t = 16 * j + 4 * g + r;
        const bool t_ok = (j < nt) && t < L && q_ok;
        const bool km = t_ok && s_mask[t];
        float x = acc[j][r] * 0.125f;
        if (SPATIAL && t_ok) {
          float sig;
          x += spatial_bias(P.pl + ((row0 + qi) * L + t) * 5, w, km, sig);
        }
        const float p = (t_ok && !km) ? __expf(x - lse_q) : 0.f;
        float dp = dacc[j][r], pd = p;
        if (dropout) {
          const unsigned long long idx = (((unsigned long long)b * P.H + h) * L + qi) * L + t;
          const bool keep = rng_u32(seed, idx) >= P.drop_thr;
          dp = keep ? dp * keep_scale : 0.f;
          pd = keep ? p * keep_scale : 0.f;
        }
        if (j < nt) PT[t * TS + 16 * s + m] = f2bf(pd);
        acc[j][r] = p;
        dacc[j][r] = dp;
        delta += p * dp;
      }
    delta = xor_reduce_sum_rows(delta);
    float dw[SD];
#pragma unroll
    for (int d = 0; d < SD; ++d) dw[d] = 0.f;
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int t = 16 * j + 4 * g + r;
        const float dlogit = acc[j][r] * (dacc[j][r] - delta);
        if (SPATIAL) {
          if (dlogit != 0.f) {          // p > 0: key is real and not masked
            const float *plp = P.pl + ((row0 + qi) * L + t) * 5;
            float z = w[0];
#pragma unroll
            for (int d = 0; d < 5; ++d) z = fmaf(w[1 + d], plp[d], z);
            const float sig = __builtin_amdgcn_rcpf(1.f + __expf(-z));
            const float dz = sig > 1e-6f ? dlogit * (1.f - sig) : 0.f;   // d/dz log(clamp(sigmoid z, 1e-6))
            dw[0] += dz;
#pragma unroll
            for (int d = 0; d < 5; ++d) dw[1 + d] = fmaf(dz, plp[d], dw[1 + d]);
          }
        }
        const float dsv = dlogit * 0.125f;
        if (j < nt) dST[t * TS + 16 * s + m] = f2bf(dsv);
        acc[j][r] = dsv;
      }
    if (SPATIAL) {
#pragma unroll
      for (int d = 0; d < SD; ++d) dw[d] = xor_reduce_sum_rows(dw[d]);
      if (g == 0 && q_ok) {
#pragma unroll
        for (int d = 0; d < SD; ++d) P.dsw[((row0 + qi) * P.H + h) * SD + d] = dw[d];
      }
    }
    f32x4 o[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) o[n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      if (2 * c < nt) {
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        const bf16x8 da = pack_tiles(acc[2 * c], (2 * c + 1 < NT) ? acc[(2 * c + 1 < NT) ? 2 * c + 1 : 0] : z);
#pragma unroll
        for (int n = 0; n < 4; ++n) o[n] = mfma(da, frag_from_transposed(Kt, TS, n, c, lane), o[n]);
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int qr = 16 * s + 4 * g + r;
      if (qr < L) {
        uint16_t *op = P.dq + (row0 + qr) * P.ld_qkv + h * DH + m;
#pragma unroll
        for (int n = 0; n < 4; ++n) op[16 * n] = f2bf(o[n][r]);
      }
    }
  }
  __syncthreads();   // PT / dST complete; K / V tiles dead
  stage_transposed(Qt, qb, P.ld_qkv, L, NC * 32, TS);
  stage_transposed(dOt, dob, P.ld_o, L, NC * 32, TS);
  __syncthreads();

  // ---------------- pass 2: key strips, MFMA only ----------------
  for (int js = wave; js < nt; js += nwaves) {
    f32x4 dk[4], dv[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      dk[n] = f32x4{0.f, 0.f, 0.f, 0.f};
      dv[n] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const uint16_t *prow = PT + (16 * js + m) * TS + 8 * g;
    const uint16_t *drow = dST + (16 * js + m) * TS + 8 * g;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      if (2 * c < nt) {
        const bf16x8 pa = as_frag(*reinterpret_cast<const u32x4 *>(prow + 32 * c));
        const bf16x8 da = as_frag(*reinterpret_cast<const u32x4 *>(drow + 32 * c));
#pragma unroll
        for (int n = 0; n < 4; ++n) {
          const bf16x8 bo = as_frag(*reinterpret_cast<const u32x4 *>(dOt + (16 * n + m) * TS + 32 * c + 8 * g));
          const bf16x8 bqf = as_frag(*reinterpret_cast<const u32x4 *>(Qt + (16 * n + m) * TS + 32 * c + 8 * g));
          dv[n] = mfma(pa, bo, dv[n]);
          dk[n] = mfma(da, bqf, dk[n]);
        }
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int tr = 16 * js + 4 * g + r;
      if (tr < L) {
        uint16_t *pk = P.dk + (row0 + tr) * P.ld_qkv + h * DH + m;
        uint16_t *pv = P.dv + (row0 + tr) * P.ld_qkv + h * DH + m;
#pragma unroll
        for (int n = 0; n < 4; ++n) {
          pk[16 * n] = f2bf(dk[n][r]);
          pv[16 * n] = f2bf(dv[n][r]);
        }
      }
    }
  }
}

// ==========================================================================================
// streaming kernels for long sequences (144 < L <= 512: the 300-token scene captions, all_pretrain.yaml:35-36,46,
// the T = 512 joint sequence of BASELINE configs[4], 256-object scenes).  Same mathematics, same dropout stream and
// same operand layouts as the kernels above, but no query strip ever holds a whole score row in registers:
//   forward   pass A streams the key tiles once for the row maximum and the normaliser (online, per lane, merged
//             across the four lane groups at the end), pass B streams them again, recomputes the logits, and feeds
//             P V chunk by chunk;
//   backward  delta = rowsum(dO * O) is computed first (needs the forward output), so the dQ pass can treat every
//             32-key chunk independently; the dK / dV pass is the chunked loop of the recompute kernel.
// K and V stay ROW-major in LDS and serve both operand kinds: A fragments by 16-byte reads, B fragments by the
// hardware-transposed ds_read_b64_tr_b16 (no transposing stores, no V^T / K^T tiles).
// ==========================================================================================
typedef __attribute__((ext_vector_type(4))) short s16x4;

// dropout of the streaming kernels: ONE hash per pair of adjacent keys (t even, t odd) of a query, its low / high
// 16 bits decide the two elements (threshold p * 2^16: the drop probability is quantised to 1 / 65536).  The three
// places that need the mask (forward pass B, both backward passes) evaluate the same function of (query, key pair).
// one hash decides the two keys of a pair (16-bit thresholds): index of the pair = (row base + key) / 2 with an even
// row pitch, so that the lane holding keys (4 g + 0..3) of a query needs two hashes
__device__ __forceinline__ unsigned int pair_rng(unsigned int seedmix, unsigned int row_pair_base, int t) {
  return mix32((row_pair_base + (unsigned int)(t >> 1)) ^ seedmix);
}
__device__ __forceinline__ bool pair_keep(unsigned int r, int t, unsigned int thr16) {
  return ((t & 1) ? (r >> 16) : (r & 0xFFFFu)) >= thr16;
}
constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;

// B fragment (32 keys x 16 columns, K order of pack_tiles) from a row-major [key][KS] tile
__device__ __forceinline__ bf16x8 frag_from_rows_tr(const uint16_t *rows, int ntile, int c, int lane) {
  const int i = lane & 15, g = lane >> 4;
  const uint16_t *p = rows + (32 * c + 4 * g + (i >> 2)) * KS + 16 * ntile + 4 * (i & 3);
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)p);
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(p + 16 * KS));
  const u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
  const u32x4 v = {l2[0], l2[1], h2[0], h2[1]};
  return as_frag(v);
}

template <bool SPATIAL>
__global__ __launch_bounds__(1024) void attn_fwd_stream_kernel(const Params P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  int b, h;
  block_to_bh(P, b, h);
  if (P.seq_order) b = P.seq_order[b];
  // fixed-length batch: sequence b = rows [b L, b L + L); packed variable-length batch: rows [cu[b], cu[b + 1])
  int L = P.L, Lq = P.Lq;
  size_t row0 = (size_t)b * P.L, row0q = (size_t)b * P.Lq;
  if (P.cu_rows) {
    row0 = row0q = (size_t)P.cu_rows[b];
    L = Lq = P.cu_rows[b + 1] - P.cu_rows[b];
    if (L <= 0) return;                                     // workgroup-uniform: an empty sequence has no rows at all
    if (P.q_limit) Lq = min(Lq, max(P.q_limit[b], 0));      // only the leading queries are wanted
  }
  const int nt = (L + 15) / 16, nc = (nt + 1) / 2, rows = nc * 32;       // keys
  const int ntq = (Lq + 15) / 16;                                         // queries
  const int Lq_cap = P.Lq;                                                // pitch of lse and of the dropout counter
  uint16_t *Ks = reinterpret_cast<uint16_t *>(smem);       // [rows][KS]
  uint16_t *Vs = Ks + rows * KS;                            // [rows][KS]
  float *mb = reinterpret_cast<float *>(Vs + rows * KS);    // [rows] additive key term: 0, or -inf (padded / past L)

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  const int m = lane & 15, g = lane >> 4;
  const uint16_t *qb = P.q + row0q * P.ld_q + h * DH;
  const uint16_t *kb = P.k + row0 * P.ld_qkv + h * DH;
  const uint16_t *vb = P.v + row0 * P.ld_qkv + h * DH;
  stage_rows(Ks, kb, P.ld_qkv, L, rows);
  stage_rows(Vs, vb, P.ld_qkv, L, rows);
  for (int t = threadIdx.x; t < rows; t += blockDim.x) mb[t] = (t < L && !(P.mask && P.mask[row0 + t])) ? 0.f : -INFINITY;
  __syncthreads();

  const bool dropout = P.drop_thr != 0u;
  const float keep_scale = dropout ? 1.f / (1.f - P.p_drop) : 1.f;
  const unsigned int seedmix = dropout ? seed_fold(effective_seed(P)) : 0u;
  const unsigned int thr16 = P.drop_thr >> 16;
  const unsigned int pitch2 = (unsigned int)((P.L + 1) >> 1);     // key pairs per query row (capacity: the same in backward)

  for (int s = wave; s < ntq; s += nwaves) {
    const int qi = 16 * s + m;
    const bool q_ok = qi < Lq;
    bf16x8 bq[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      u32x4 v = zero4();
      if (q_ok) v = *reinterpret_cast<const u32x4 *>(qb + (size_t)qi * P.ld_q + 32 * c + 8 * g);
      bq[c] = as_frag(v);
    }
    float w[SD];
    if (SPATIAL) {
#pragma unroll
      for (int d = 0; d < SD; ++d) w[d] = q_ok ? P.sw[((row0q + qi) * P.H + h) * SD + d] : 0.f;
    }
    // base-2 logits of (query qi, keys 16 j + 4 g + 0..3): log2(e) * (q . k / 8 [+ spatial term]) + key term;
    // one fused multiply-add per element in the plain form
    auto logits2 = [&](int j, f32x4 &x) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const u32x4 a = *reinterpret_cast<const u32x4 *>(Ks + (16 * j + m) * KS + 32 * c + 8 * g);
        acc = mfma(as_frag(a), bq[c], acc);
      }
      const f32x4 kt = *reinterpret_cast<const f32x4 *>(mb + 16 * j + 4 * g);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (SPATIAL) {
          const int t = 16 * j + 4 * g + r;
          float v = acc[r] * 0.125f;
          if (t < L && q_ok) {
            float sig;
            v += spatial_bias(P.pl + ((row0q + qi) * L + t) * 5, w, kt[r] < 0.f, sig);
          }
          x[r] = fmaf(v, kLog2e, kt[r]);
        } else {
          x[r] = fmaf(acc[r], 0.125f * kLog2e, kt[r]);
        }
      }
    };
    // pass A: the row maximum only (no exponentials)
    float mx = -INFINITY;
    for (int j = 0; j < nt; ++j) {
      f32x4 x;
      logits2(j, x);
      mx = fmaxf(mx, fmaxf(fmaxf(x[0], x[1]), fmaxf(x[2], x[3])));
    }
    const float gmx = xor_reduce_max_rows(mx);
    // pass B: p = 2^(x - max) chunk by chunk; the normaliser accumulates beside the P V product and is applied,
    // with the dropout scale, to the 16 x 64 output strip at the end
    const unsigned int rp = (((unsigned int)b * P.H + h) * Lq_cap + qi) * pitch2;
    float lsum = 0.f;
    f32x4 o[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) o[n] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < nc; ++c) {
      f32x4 pt[2];
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        const int j = 2 * c + hh;
        pt[hh] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (j < nt) {
          f32x4 x;
          logits2(j, x);
#pragma unroll
          for (int r = 0; r < 4; ++r) pt[hh][r] = __builtin_amdgcn_exp2f(x[r] - gmx);
          lsum += (pt[hh][0] + pt[hh][1]) + (pt[hh][2] + pt[hh][3]);
          if (dropout) {        // keys 16 j + 4 g + {0,1} and {2,3}: two hashes for the four elements
            const int t0 = 16 * j + 4 * g;
            const unsigned int r01 = pair_rng(seedmix, rp, t0), r23 = pair_rng(seedmix, rp, t0 + 2);
            pt[hh][0] = (r01 & 0xFFFFu) >= thr16 ? pt[hh][0] : 0.f;
            pt[hh][1] = (r01 >> 16) >= thr16 ? pt[hh][1] : 0.f;
            pt[hh][2] = (r23 & 0xFFFFu) >= thr16 ? pt[hh][2] : 0.f;
            pt[hh][3] = (r23 >> 16) >= thr16 ? pt[hh][3] : 0.f;
          }
        }
      }
      const bf16x8 pa = pack_tiles(pt[0], pt[1]);
#pragma unroll
      for (int n = 0; n < 4; ++n) o[n] = mfma(pa, frag_from_rows_tr(Vs, n, c, lane), o[n]);
    }
    lsum = xor_reduce_sum_rows(lsum);             // all keys masked -> NaN row, like torch
    if (g == 0 && q_ok) P.lse[((size_t)b * P.H + h) * Lq_cap + qi] = (gmx + __builtin_amdgcn_logf(lsum)) * kLn2;
    const float scale_q = keep_scale / lsum;      // of query 16 s + m; the output rows of this lane are 4 g + r
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int qr = 16 * s + 4 * g + r;
      const float sc = __shfl(scale_q, 4 * g + r, 64);
      if (qr < Lq) {
        uint16_t *op = P.out + (row0q + qr) * P.ld_o + h * DH + m;
#pragma unroll
        for (int n = 0; n < 4; ++n) op[16 * n] = f2bf(o[n][r] * sc);
      }
    }
  }
}

template <bool SPATIAL>
__global__ __launch_bounds__(1024) void attn_bwd_stream_kernel(const Params P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  int b, h;
  block_to_bh(P, b, h);
  if (P.seq_order) b = P.seq_order[b];
  int L = P.L, Lq = P.Lq;
  size_t row0 = (size_t)b * P.L, row0q = (size_t)b * P.Lq;
  if (P.cu_rows) {                                           // packed variable-length sequences (see the forward kernel)
    row0 = row0q = (size_t)P.cu_rows[b];
    L = Lq = P.cu_rows[b + 1] - P.cu_rows[b];
    if (L <= 0) return;
    if (P.q_limit) Lq = min(Lq, max(P.q_limit[b], 0));       // the other queries' dout counts as zero
  }
  const int nt = (L + 15) / 16, nc = (nt + 1) / 2, rows = nc * 32;               // keys
  const int ntq = (Lq + 15) / 16, ncq = (ntq + 1) / 2, rows_q = ncq * 32;         // queries
  const int rows_max = rows > rows_q ? rows : rows_q;
  const int Lq_cap = P.Lq;
  // pass 1: Ks [rows][KS] | Vs [rows][KS];  pass 2 (same storage): Qs [rows_q][KS] | dOs [rows_q][KS];  then fp32 rows:
  // delta, lse2 = log2(e) * lse (+inf past Lq: such queries get p = 0) per query and the additive key term (0 / -inf).
  // Every tile is ROW-major: A fragments are 16-byte reads, B fragments hardware-transposed reads of the same rows.
  uint16_t *Ks = reinterpret_cast<uint16_t *>(smem);
  uint16_t *Vs = Ks + rows * KS;
  uint16_t *Qs = Ks, *dOs = Qs + rows_q * KS;
  float *delta_s = reinterpret_cast<float *>(smem + (size_t)2 * rows_max * KS * 2);
  float *lse_s = delta_s + rows_q;
  float *mb = lse_s + rows_q;

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  const int m = lane & 15, g = lane >> 4;
  const uint16_t *qb = P.q + row0q * P.ld_q + h * DH;
  const uint16_t *kb = P.k + row0 * P.ld_qkv + h * DH;
  const uint16_t *vb = P.v + row0 * P.ld_qkv + h * DH;
  const uint16_t *dob = P.dout + row0q * P.ld_o + h * DH;
  const uint16_t *ob = P.out + row0q * P.ld_o + h * DH;
  const float *lse = P.lse + ((size_t)b * P.H + h) * Lq_cap;
  const bool dropout = P.drop_thr != 0u;
  const float keep_scale = dropout ? 1.f / (1.f - P.p_drop) : 1.f;
  const unsigned int seedmix = dropout ? seed_fold(effective_seed(P)) : 0u;
  const unsigned int thr16 = P.drop_thr >> 16;
  const unsigned int pitch2 = (unsigned int)((P.L + 1) >> 1);
  const unsigned int bh_base = ((unsigned int)b * P.H + h) * Lq_cap;

  stage_rows(Ks, kb, P.ld_qkv, L, rows);
  stage_rows(Vs, vb, P.ld_qkv, L, rows);
  // delta[q] = sum_d dO[q][d] O[q][d] = rowsum(P' dP') for the dropped, rescaled probabilities of the forward pass
  for (int t = threadIdx.x; t < rows; t += blockDim.x) mb[t] = (t < L && !(P.mask && P.mask[row0 + t])) ? 0.f : -INFINITY;
  for (int t = threadIdx.x; t < rows_q; t += blockDim.x) {
    float d = 0.f;
    if (t < Lq) {
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const u32x4 a = *reinterpret_cast<const u32x4 *>(dob + (size_t)t * P.ld_o + 8 * c);
        const u32x4 o = *reinterpret_cast<const u32x4 *>(ob + (size_t)t * P.ld_o + 8 * c);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          d = fmaf(__uint_as_float(a[e] << 16), __uint_as_float(o[e] << 16), d);
          d = fmaf(__uint_as_float(a[e] & 0xFFFF0000u), __uint_as_float(o[e] & 0xFFFF0000u), d);
        }
      }
    }
    delta_s[t] = d;
    lse_s[t] = t < Lq ? lse[t] * kLog2e : INFINITY;
  }
  __syncthreads();

  // ---------------- pass 1: query strips, key chunks streamed -> dQ, dsw ----------------
  for (int s = wave; s < ntq; s += nwaves) {
    const int qi = 16 * s + m;
    const bool q_ok = qi < Lq;
    bf16x8 bq[2], bdo[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      u32x4 v = zero4(), u = zero4();
      if (q_ok) {
        v = *reinterpret_cast<const u32x4 *>(qb + (size_t)qi * P.ld_q + 32 * c + 8 * g);
        u = *reinterpret_cast<const u32x4 *>(dob + (size_t)qi * P.ld_o + 32 * c + 8 * g);
      }
      bq[c] = as_frag(v);
      bdo[c] = as_frag(u);
    }
    float w[SD], dw[SD];
#pragma unroll
    for (int d = 0; d < SD; ++d) {
      w[d] = (SPATIAL && q_ok) ? P.sw[((row0q + qi) * P.H + h) * SD + d] : 0.f;
      dw[d] = 0.f;
    }
    const float lse_q = lse_s[qi < rows_q ? qi : 0];
    const float delta = delta_s[qi < rows_q ? qi : 0];
    const unsigned int rp = (bh_base + qi) * pitch2;
    f32x4 o[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) o[n] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < nc; ++c) {
      f32x4 ds[2];
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        const int j = 2 * c + hh;
        ds[hh] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (j < nt) {
          f32x4 acc = {0.f, 0.f, 0.f, 0.f}, dacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int cc = 0; cc < 2; ++cc) {
            const u32x4 a = *reinterpret_cast<const u32x4 *>(Ks + (16 * j + m) * KS + 32 * cc + 8 * g);
            const u32x4 av = *reinterpret_cast<const u32x4 *>(Vs + (16 * j + m) * KS + 32 * cc + 8 * g);
            acc = mfma(as_frag(a), bq[cc], acc);        // S^T
            dacc = mfma(as_frag(av), bdo[cc], dacc);    // (dO V^T)^T
          }
          const int t0 = 16 * j + 4 * g;
          const f32x4 kt = *reinterpret_cast<const f32x4 *>(mb + t0);
          bool keep[4] = {true, true, true, true};
          if (dropout) {
            const unsigned int r01 = pair_rng(seedmix, rp, t0), r23 = pair_rng(seedmix, rp, t0 + 2);
            keep[0] = (r01 & 0xFFFFu) >= thr16;
            keep[1] = (r01 >> 16) >= thr16;
            keep[2] = (r23 & 0xFFFFu) >= thr16;
            keep[3] = (r23 >> 16) >= thr16;
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float p, gt = 0.f;
            if (SPATIAL) {
              const int t = t0 + r;
              float x = acc[r] * 0.125f;
              if (t < L && q_ok) {
                float sig;
                x += spatial_bias(P.pl + ((row0q + qi) * L + t) * 5, w, kt[r] < 0.f, sig);
                gt = sig > 1e-6f ? 1.f - sig : 0.f;
              }
              p = __builtin_amdgcn_exp2f(fmaf(x, kLog2e, kt[r]) - lse_q);
            } else {
              p = __builtin_amdgcn_exp2f(fmaf(acc[r], 0.125f * kLog2e, kt[r]) - lse_q);
            }
            const float dp = keep[r] ? dacc[r] : 0.f;                       // dropped gradient, before its 1/(1-p) scale
            const float dlogit = p * fmaf(dp, keep_scale, -delta);
            if (SPATIAL) {
              const float dz = dlogit * gt;
              if (dz != 0.f) {
                const float *plp = P.pl + ((row0q + qi) * L + t0 + r) * 5;
                dw[0] += dz;
#pragma unroll
                for (int d = 0; d < 5; ++d) dw[1 + d] = fmaf(dz, plp[d], dw[1 + d]);
              }
            }
            ds[hh][r] = dlogit;                                             // the 1/8 of the logits goes onto dQ below
          }
        }
      }
      const bf16x8 da = pack_tiles(ds[0], ds[1]);
#pragma unroll
      for (int n = 0; n < 4; ++n) o[n] = mfma(da, frag_from_rows_tr(Ks, n, c, lane), o[n]);
    }
    if (SPATIAL) {
#pragma unroll
      for (int d = 0; d < SD; ++d) dw[d] = xor_reduce_sum_rows(dw[d]);
      if (g == 0 && q_ok) {
#pragma unroll
        for (int d = 0; d < SD; ++d) P.dsw[((row0q + qi) * P.H + h) * SD + d] = dw[d];
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int qr = 16 * s + 4 * g + r;
      if (qr < Lq) {
        uint16_t *op = P.dq + (row0q + qr) * P.ld_dq + h * DH + m;
#pragma unroll
        for (int n = 0; n < 4; ++n) op[16 * n] = f2bf(o[n][r] * 0.125f);
      }
    }
  }
  if (P.q_limit && Lq < L) {               // queries that were not computed: their dQ rows are zero
    for (int e = threadIdx.x; e < (L - Lq) * 8; e += blockDim.x) {
      const int qr = Lq + (e >> 3), c8 = e & 7;
      *reinterpret_cast<u32x4 *>(P.dq + (row0q + qr) * P.ld_dq + h * DH + 8 * c8) = zero4();
    }
  }
  __syncthreads();   // K / V tiles no longer needed: the same storage now takes Q and dO (row-major as well)
  stage_rows(Qs, qb, P.ld_q, Lq, rows_q);
  stage_rows(dOs, dob, P.ld_o, Lq, rows_q);
  __syncthreads();

  // ---------------- pass 2: key strips, query chunks streamed -> dK, dV ----------------
  for (int js = wave; js < nt; js += nwaves) {
    const int t = 16 * js + m;            // this lane's key
    const bool t_ok = t < L;
    const float kt = mb[t < rows ? t : 0];
    bf16x8 bk[2], bv[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      u32x4 v = zero4(), u = zero4();
      if (t_ok) {
        v = *reinterpret_cast<const u32x4 *>(kb + (size_t)t * P.ld_qkv + 32 * c + 8 * g);
        u = *reinterpret_cast<const u32x4 *>(vb + (size_t)t * P.ld_qkv + 32 * c + 8 * g);
      }
      bk[c] = as_frag(v);
      bv[c] = as_frag(u);
    }
    f32x4 dk[4], dv[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      dk[n] = f32x4{0.f, 0.f, 0.f, 0.f};
      dv[n] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    for (int c = 0; c < ncq; ++c) {       // 32-query chunks
      f32x4 pt[2], ds[2];
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        const int i = 2 * c + hh;          // query tile
        pt[hh] = f32x4{0.f, 0.f, 0.f, 0.f};
        ds[hh] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (i < ntq) {
          f32x4 sacc = {0.f, 0.f, 0.f, 0.f}, dacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int cc = 0; cc < 2; ++cc) {
            const u32x4 v = *reinterpret_cast<const u32x4 *>(Qs + (16 * i + m) * KS + 32 * cc + 8 * g);
            const u32x4 u = *reinterpret_cast<const u32x4 *>(dOs + (16 * i + m) * KS + 32 * cc + 8 * g);
            sacc = mfma(as_frag(v), bk[cc], sacc);    // S[query 16 i + 4 g + r][key t]
            dacc = mfma(as_frag(u), bv[cc], dacc);    // dO V^T
          }
          const int q0 = 16 * i + 4 * g;
          const f32x4 lq = *reinterpret_cast<const f32x4 *>(lse_s + q0);
          const f32x4 dq4 = *reinterpret_cast<const f32x4 *>(delta_s + q0);
          // dropout hashes of (query q0 + r, key pair t >> 1): this lane computes two of the four, the lane of
          // the other key of the pair (m ^ 1) the other two
          bool keep[4] = {true, true, true, true};
          if (dropout) {
            const int par = m & 1;
            const unsigned int qa = (unsigned int)(q0 + 2 * par);
            const unsigned int ha = pair_rng(seedmix, (bh_base + qa) * pitch2, t);
            const unsigned int hb = pair_rng(seedmix, (bh_base + qa + 1u) * pitch2, t);
            const unsigned int oa = (unsigned int)__shfl_xor((int)ha, 1, 64), ob2 = (unsigned int)__shfl_xor((int)hb, 1, 64);
            const unsigned int hr0 = par ? oa : ha, hr1 = par ? ob2 : hb, hr2 = par ? ha : oa, hr3 = par ? hb : ob2;
            const unsigned int sh = par ? 16u : 0u;
            keep[0] = ((hr0 >> sh) & 0xFFFFu) >= thr16;
            keep[1] = ((hr1 >> sh) & 0xFFFFu) >= thr16;
            keep[2] = ((hr2 >> sh) & 0xFFFFu) >= thr16;
            keep[3] = ((hr3 >> sh) & 0xFFFFu) >= thr16;
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float p;
            if (SPATIAL) {
              const int qi = q0 + r;
              float x = sacc[r] * 0.125f;
              if (t_ok && qi < Lq) {
                float w[SD];
#pragma unroll
                for (int d = 0; d < SD; ++d) w[d] = P.sw[((row0q + qi) * P.H + h) * SD + d];
                float sig;
                x += spatial_bias(P.pl + ((row0q + qi) * L + t) * 5, w, kt < 0.f, sig);
              }
              p = __builtin_amdgcn_exp2f(fmaf(x, kLog2e, kt) - lq[r]);
            } else {
              p = __builtin_amdgcn_exp2f(fmaf(sacc[r], 0.125f * kLog2e, kt) - lq[r]);
            }
            const float dp = keep[r] ? dacc[r] : 0.f;
            pt[hh][r] = keep[r] ? p : 0.f;                                  // 1/(1-p) goes onto dV below
            ds[hh][r] = p * fmaf(dp, keep_scale, -dq4[r]);                  // 1/8 goes onto dK below
          }
        }
      }
      const bf16x8 pa = pack_tiles(pt[0], pt[1]);
      const bf16x8 da = pack_tiles(ds[0], ds[1]);
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        dv[n] = mfma(pa, frag_from_rows_tr(dOs, n, c, lane), dv[n]);
        dk[n] = mfma(da, frag_from_rows_tr(Qs, n, c, lane), dk[n]);
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int tr = 16 * js + 4 * g + r;
      if (tr < L) {
        uint16_t *pk = P.dk + (row0 + tr) * P.ld_dkv + h * DH + m;
        uint16_t *pv = P.dv + (row0 + tr) * P.ld_dkv + h * DH + m;
#pragma unroll
        for (int n = 0; n < 4; ++n) {
          pk[16 * n] = f2bf(dk[n][r] * 0.125f);
          pv[16 * n] = f2bf(dv[n][r] * keep_scale);
        }
      }
    }
  }
}

inline size_t stream_fwd_lds(int nt) {
  const size_t rows = (size_t)((nt + 1) / 2) * 32;
  return 2 * (2 * rows * KS) + 4 * rows;
}
inline size_t stream_bwd_lds(int nt, int ntq) {
  const size_t rows = (size_t)((nt + 1) / 2) * 32, rows_q = (size_t)((ntq + 1) / 2) * 32;
  const size_t rows_max = rows > rows_q ? rows : rows_q;
  return 2 * (2 * rows_max * KS) + 8 * rows_q + 4 * rows;
}

inline int pick_waves_stream(int nt) {        // up to 16 waves (one 1024-thread workgroup per (scene, head))
  const int rounds = (nt + 15) / 16;
  return (nt + rounds - 1) / rounds;
}

int launch_stream(const Params &P, bool backward, hipStream_t s) {
  // forward and backward pass 1 walk query strips, backward pass 2 key strips
  const int nw = pick_waves_stream(backward ? (P.nt > P.ntq ? P.nt : P.ntq) : P.ntq);
  const dim3 grid(P.B * P.H), block(64 * nw);
  const bool spatial = P.sw != nullptr;
  const size_t lds = backward ? stream_bwd_lds(P.nt, P.ntq) : stream_fwd_lds(P.nt);
  if (lds > 160 * 1024) return GPS_ERR_UNSUPPORTED;
  const void *fn = backward ? (spatial ? (const void *)&attn_bwd_stream_kernel<true> : (const void *)&attn_bwd_stream_kernel<false>)
                            : (spatial ? (const void *)&attn_fwd_stream_kernel<true> : (const void *)&attn_fwd_stream_kernel<false>);
  static gps_dev::PerDevice<size_t, 4> granted_dev;
  size_t *granted = granted_dev.row();
  const int slot = (backward ? 2 : 0) + (spatial ? 1 : 0);
  if (lds > 64 * 1024 && lds > granted[slot]) {
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return GPS_ERR_LAUNCH;
    granted[slot] = 160 * 1024;
  }
  if (backward) {
    if (spatial) hipLaunchKernelGGL((attn_bwd_stream_kernel<true>), grid, block, lds, s, P);
    else hipLaunchKernelGGL((attn_bwd_stream_kernel<false>), grid, block, lds, s, P);
  } else {
    if (spatial) hipLaunchKernelGGL((attn_fwd_stream_kernel<true>), grid, block, lds, s, P);
    else hipLaunchKernelGGL((attn_fwd_stream_kernel<false>), grid, block, lds, s, P);
  }
  return hipGetLastError() == hipSuccess ? GPS_OK : GPS_ERR_LAUNCH;
}

template <int NT>
size_t fwd_lds() {
  constexpr int NC = (NT + 1) / 2;
  return (size_t)2 * (NT * 16 * KS + 64 * (NC * 32 + 8)) + (size_t)NT * 16;
}
template <int NT>
size_t bwd_recompute_lds() {
  constexpr int NC = (NT + 1) / 2;
  return (size_t)2 * (2 * NT * 16 * KS + 64 * (NC * 32 + 8)) + (size_t)4 * NT * 16;
}
template <int NT>
size_t bwd_lds() {                      // region A + PT + dST
  constexpr int NC = (NT + 1) / 2;
  constexpr int TS = NC * 32 + 8;
  return (size_t)2 * (2 * NT * 16 * KS + 64 * TS + 2 * NT * 16 * TS) + (size_t)NT * 16;
}
constexpr size_t kLdsMax = 160 * 1024;
inline int pick_waves(int nt) {
  const int rounds = (nt + 7) / 8;
  return (nt + rounds - 1) / rounds;
}

template <int NT>
int launch(const Params &P, bool backward, hipStream_t s) {
  const int nw = pick_waves(P.nt);
  const dim3 grid(P.B * P.H), block(64 * nw);
  const bool spatial = P.sw != nullptr;
  const bool resident = bwd_lds<NT>() <= kLdsMax;     // P^T / dS^T fit LDS
  const size_t lds = backward ? (resident ? bwd_lds<NT>() : bwd_recompute_lds<NT>()) : fwd_lds<NT>();
  if (lds > 64 * 1024) {
    static gps_dev::PerDevice<bool, 4> done_dev;
    bool *done = done_dev.row();
    const int slot = (backward ? 2 : 0) + (spatial ? 1 : 0);
    if (!done[slot]) {
      const void *fn =
          backward ? (resident ? (spatial ? (const void *)&attn_bwd_kernel<NT, true> : (const void *)&attn_bwd_kernel<NT, false>)
                               : (spatial ? (const void *)&attn_bwd_recompute_kernel<NT, true>
                                          : (const void *)&attn_bwd_recompute_kernel<NT, false>))
                   : (spatial ? (const void *)&attn_fwd_kernel<NT, true> : (const void *)&attn_fwd_kernel<NT, false>);
      if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return GPS_ERR_LAUNCH;
      done[slot] = true;
    }
  }
  if (backward && resident) {
    if (spatial) hipLaunchKernelGGL((attn_bwd_kernel<NT, true>), grid, block, lds, s, P);
    else hipLaunchKernelGGL((attn_bwd_kernel<NT, false>), grid, block, lds, s, P);
  } else if (backward) {
    if (spatial) hipLaunchKernelGGL((attn_bwd_recompute_kernel<NT, true>), grid, block, lds, s, P);
    else hipLaunchKernelGGL((attn_bwd_recompute_kernel<NT, false>), grid, block, lds, s, P);
  } else {
    if (spatial) hipLaunchKernelGGL((attn_fwd_kernel<NT, true>), grid, block, lds, s, P);
    else hipLaunchKernelGGL((attn_fwd_kernel<NT, false>), grid, block, lds, s, P);
  }
  return hipGetLastError() == hipSuccess ? GPS_OK : GPS_ERR_LAUNCH;
}

static int g_stream_min_nt[2] = {-1, -1};      // [plain, spatial]; -1 = not read yet
// plain form (no pairwise term), bits: 1 = forward on the block-streaming kernels of gps_attention_fa.hip, 2 = backward
// on them too, 4 = fixed-length self-attention up to 144 tokens on the K / V-resident kernels of gps_attention_sp.hip
static int g_plain_mode = [] {
  const char *e = getenv("GPS_ATTN_PLAIN_MODE");            // A/B runs of whole bench commands
  return e ? (atoi(e) & 7) : (1 | 4);
}();

int dispatch(Params &P, bool backward, hipStream_t s) {
  P.nt = (P.L + 15) / 16;
  if (P.Lq <= 0) P.Lq = P.L;
  P.ntq = (P.Lq + 15) / 16;
  if (P.ld_q <= 0) P.ld_q = P.ld_qkv;
  if (P.ld_dq <= 0) P.ld_dq = P.ld_qkv;
  if (P.ld_dkv <= 0) P.ld_dkv = P.ld_qkv;
  // cross-attention (Lq != Lk) and operands with separate pitches are served by the streaming kernels only
  const bool packed_self = P.Lq == P.L && P.ld_q == P.ld_qkv && P.ld_dq == P.ld_qkv && P.ld_dkv == P.ld_qkv && !P.cu_rows;
  if (!packed_self) {
    if (P.nt > 32 || P.ntq > 32 || (backward && P.out == nullptr)) return GPS_ERR_UNSUPPORTED;
    return launch_stream(P, backward, s);
  }
  // The streaming kernels (key / query chunks streamed, no whole score row in registers) serve every length of the
  // plain form -- after their rewrite they beat the register-resident kernels at 50 and 130 tokens too
  // (profiles/r2: 0.33 + 0.13 ms vs 0.48 + 0.29 ms per step) -- and the rows above 144 tokens of the spatial form,
  // whose resident kernels (whole score row per query strip) stay ahead at 80 tokens.  The thresholds (in 16-token
  // tiles) can be moved with gps_attn_set_stream_min_tiles (tests run both families against each other) or the
  // environment (GPS_ATTN_STREAM_MIN_NT: spatial, GPS_ATTN_STREAM_MIN_NT_PLAIN).
  if (g_stream_min_nt[0] < 0) {
    const char *e0 = getenv("GPS_ATTN_STREAM_MIN_NT_PLAIN"), *e1 = getenv("GPS_ATTN_STREAM_MIN_NT");
    g_stream_min_nt[0] = e0 ? (atoi(e0) < 1 ? 1 : atoi(e0)) : 1;
    g_stream_min_nt[1] = e1 ? (atoi(e1) < 1 ? 1 : atoi(e1)) : 10;
  }
  const int stream_min_nt = g_stream_min_nt[P.pl != nullptr ? 1 : 0];
  if (P.nt >= stream_min_nt && P.nt <= 32 && (!backward || P.out != nullptr)) return launch_stream(P, backward, s);
  if (P.nt <= 5) return launch<5>(P, backward, s);
  if (P.nt <= 9) return launch<9>(P, backward, s);
  // longer rows stream their key / query chunks (no whole score row in registers); the backward form needs the
  // forward output for delta = rowsum(dO * O) -- without it (legacy callers) rows up to 256 tokens take the
  // register-resident recompute kernel
  if (P.nt <= 32 && (!backward || P.out != nullptr)) return launch_stream(P, backward, s);
  if (P.nt <= 16) return launch<16>(P, backward, s);
  return GPS_ERR_UNSUPPORTED;
}

// the general entry: argument checks, then bf16 -> the kernels of this file, fp32 / fp8 -> gps_attention_ex.hip
int run_ex(const gps_attn_args *a, bool backward, hipStream_t s) {
  if (!a) return GPS_ERR_INVALID_ARGUMENT;
  if (a->B < 0 || a->H < 1 || a->Lq < 0 || a->Lk < 0 || a->ld_q < a->H * 64 || a->ld_kv < a->H * 64 || a->ld_o < a->H * 64 ||
      a->p_drop < 0.f || a->p_drop >= 1.f)
    return GPS_ERR_INVALID_ARGUMENT;
  if (a->head_dim != DH) return GPS_ERR_UNSUPPORTED;
  if (a->B == 0 || a->Lq == 0) return GPS_OK;
  if (a->Lk == 0) return GPS_ERR_INVALID_ARGUMENT;              // a softmax over no keys
  if (a->pl_planes) {                                            // plane form of the spatial term: gps_attention_sp.hip
    if (!a->q || !a->k || !a->v || !a->out || !a->lse || a->pl || a->sw || a->dsw) return GPS_ERR_INVALID_ARGUMENT;
    if (backward && (!a->dout || !a->dq || !a->dk || !a->dv || a->ld_dq < a->H * 64 || a->ld_dkv < a->H * 64)) return GPS_ERR_INVALID_ARGUMENT;
    if ((a->ld_q & 7) || (a->ld_kv & 7) || (a->ld_o & 3) || (backward && ((a->ld_dq & 3) || (a->ld_dkv & 3)))) return GPS_ERR_UNSUPPORTED;
    return run_spatial_planes(a, backward, s);
  }
  if (!a->q || !a->k || !a->v || !a->out || !a->lse || ((a->sw == nullptr) != (a->pl == nullptr))) return GPS_ERR_INVALID_ARGUMENT;
  if (a->sw && a->Lq != a->Lk) return GPS_ERR_INVALID_ARGUMENT;  // the pairwise term is a self-attention term
  if (a->cu_rows && (a->Lq != a->Lk || a->sw || a->mask || a->dtype != GPS_ATTN_BF16 || a->compute != GPS_ATTN_COMPUTE_NATIVE))
    return GPS_ERR_UNSUPPORTED;                                  // packed variable-length form: plain bf16 self-attention
  if (backward && (!a->dout || !a->dq || !a->dk || !a->dv || a->ld_dq < a->H * 64 || a->ld_dkv < a->H * 64 || (a->sw && !a->dsw)))
    return GPS_ERR_INVALID_ARGUMENT;
  if (a->dtype == GPS_ATTN_F32) {
    if (a->compute != GPS_ATTN_COMPUTE_NATIVE) return GPS_ERR_UNSUPPORTED;
    return run_f32(a, backward, s);
  }
  if (a->dtype != GPS_ATTN_BF16) return GPS_ERR_UNSUPPORTED;
  if ((a->ld_q & 7) || (a->ld_kv & 7) || (a->ld_o & 7) || (backward && ((a->ld_dq & 7) || (a->ld_dkv & 7)))) return GPS_ERR_UNSUPPORTED;
  if (a->compute == GPS_ATTN_COMPUTE_FP8 && !backward) return run_fp8_forward(a, s);
  if (a->compute != GPS_ATTN_COMPUTE_NATIVE && a->compute != GPS_ATTN_COMPUTE_FP8) return GPS_ERR_UNSUPPORTED;
  // plain form: K / V-resident kernels for short fixed-length rows, block-streaming kernels otherwise (their backward
  // call needs the forward output and the delta scratch), else the whole-sequence kernels below
  if (!a->sw) {
    if ((g_plain_mode & 4) && a->Lq == a->Lk && a->Lk <= 144 && !a->cu_rows && a->ld_q == a->ld_kv && (!backward || a->ld_dq == a->ld_dkv) &&
        !(a->ld_o & 3) && (!backward || !(a->ld_dq & 3)))
      return run_plain_resident(a, backward, s);
    if (!backward ? (g_plain_mode & 1) : ((g_plain_mode & 2) && a->out && a->delta_ws)) return run_plain_blocks(a, backward, s);
  }
  Params P = {};
  P.B = a->B; P.H = a->H; P.L = a->Lk; P.Lq = a->Lq; P.ld_qkv = a->ld_kv; P.ld_q = a->ld_q; P.ld_o = a->ld_o;
  P.q = (const uint16_t *)a->q; P.k = (const uint16_t *)a->k; P.v = (const uint16_t *)a->v;
  P.sw = a->sw; P.pl = a->pl; P.mask = a->mask; P.out = (uint16_t *)a->out; P.lse = a->lse;
  P.p_drop = a->p_drop; P.seed = a->seed; P.seed_dev = (const unsigned long long *)a->seed_dev;
  P.drop_thr = a->p_drop > 0.f ? (unsigned int)((double)a->p_drop * 4294967296.0) : 0u;
  P.cu_rows = a->cu_rows;
  P.seq_order = a->cu_rows ? a->seq_order : nullptr;
  P.q_limit = a->cu_rows ? a->q_limit : nullptr;
  if (backward) {
    P.dout = (const uint16_t *)a->dout; P.dq = (uint16_t *)a->dq; P.dk = (uint16_t *)a->dk; P.dv = (uint16_t *)a->dv;
    P.ld_dq = a->ld_dq; P.ld_dkv = a->ld_dkv; P.dsw = a->dsw;
  } else {
    P.ld_dq = a->ld_q; P.ld_dkv = a->ld_kv;
  }
  return dispatch(P, backward, s);
}

}  // namespace gps_attn

extern "C" {

int gps_attn_forward_ex(const gps_attn_args *a, gps_stream_t stream) {
  return gps_attn::run_ex(a, false, (hipStream_t)stream);
}
int gps_attn_backward_ex(const gps_attn_args *a, gps_stream_t stream) {
  return gps_attn::run_ex(a, true, (hipStream_t)stream);
}

int gps_attn_set_plain_blocks(int mode) {
  const int was = gps_attn::g_plain_mode;
  if (mode >= 0) gps_attn::g_plain_mode = mode & 7;
  return was;
}

void gps_attn_set_stream_min_tiles(int plain, int spatial) {
  gps_attn::g_stream_min_nt[0] = plain < 1 ? 1 : plain;
  gps_attn::g_stream_min_nt[1] = spatial < 1 ? 1 : spatial;
}


int gps_attn_forward(int B, int H, int L, int head_dim, const void *q, const void *k, const void *v,
                     int ld_qkv, const float *sw, const float *pl, const unsigned char *mask,
                     float p_drop, unsigned long long seed, const void *seed_dev, void *out, int ld_o,
                     float *lse, gps_stream_t stream) {
  if (B < 0 || H < 1 || L < 0 || ld_qkv < H * 64 || ld_o < H * 64 || p_drop < 0.f || p_drop >= 1.f)
    return GPS_ERR_INVALID_ARGUMENT;
  if (head_dim != gps_attn::DH || (ld_qkv & 7) || (ld_o & 7)) return GPS_ERR_UNSUPPORTED;
  if (B == 0 || L == 0) return GPS_OK;
  if (!q || !k || !v || !out || !lse || ((sw == nullptr) != (pl == nullptr))) return GPS_ERR_INVALID_ARGUMENT;
  gps_attn::Params P = {};
  P.B = B; P.H = H; P.L = L; P.ld_qkv = ld_qkv; P.ld_o = ld_o;
  P.q = (const uint16_t *)q; P.k = (const uint16_t *)k; P.v = (const uint16_t *)v;
  P.sw = sw; P.pl = pl; P.mask = mask; P.out = (uint16_t *)out; P.lse = lse;
  P.p_drop = p_drop; P.seed = seed; P.seed_dev = (const unsigned long long *)seed_dev;
  P.drop_thr = p_drop > 0.f ? (unsigned int)((double)p_drop * 4294967296.0) : 0u;
  return gps_attn::dispatch(P, false, (hipStream_t)stream);
}

int gps_attn_backward(int B, int H, int L, int head_dim, const void *q, const void *k, const void *v,
                      int ld_qkv, const float *sw, const float *pl, const unsigned char *mask,
                      float p_drop, unsigned long long seed, const void *seed_dev, const void *dout,
                      int ld_o, const float *lse, const void *out, void *dq, void *dk, void *dv, float *dsw,
                      gps_stream_t stream) {
  if (B < 0 || H < 1 || L < 0 || ld_qkv < H * 64 || ld_o < H * 64 || p_drop < 0.f || p_drop >= 1.f)
    return GPS_ERR_INVALID_ARGUMENT;
  if (head_dim != gps_attn::DH || (ld_qkv & 7) || (ld_o & 7)) return GPS_ERR_UNSUPPORTED;
  if (B == 0 || L == 0) return GPS_OK;
  if (!q || !k || !v || !dout || !lse || !dq || !dk || !dv || ((sw == nullptr) != (pl == nullptr)) ||
      (sw && !dsw))
    return GPS_ERR_INVALID_ARGUMENT;
  gps_attn::Params P = {};
  P.B = B; P.H = H; P.L = L; P.ld_qkv = ld_qkv; P.ld_o = ld_o;
  P.q = (const uint16_t *)q; P.k = (const uint16_t *)k; P.v = (const uint16_t *)v;
  P.sw = sw; P.pl = pl; P.mask = mask; P.lse = const_cast<float *>(lse);
  P.dout = (const uint16_t *)dout; P.dq = (uint16_t *)dq; P.dk = (uint16_t *)dk; P.dv = (uint16_t *)dv;
  P.out = (uint16_t *)const_cast<void *>(out);
  P.dsw = dsw; P.p_drop = p_drop; P.seed = seed; P.seed_dev = (const unsigned long long *)seed_dev;
  P.drop_thr = p_drop > 0.f ? (unsigned int)((double)p_drop * 4294967296.0) : 0u;
  return gps_attn::dispatch(P, true, (hipStream_t)stream);
}

}  // extern "C"
