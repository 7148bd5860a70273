// gps_attention_sp.hip -- the language-conditioned pairwise-spatial self-attention core of the object encoder
// (reference modules/layers/transformers.py:193-239, fusion 'cond') for short rows (L <= 144: the 80 objects of the GPS
// configs), round-5 form.  Same mathematics as gps_attention.hip's attn_fwd_kernel / attn_bwd_kernel; what changed is
// where every operand comes from and when:
//
//   * the pairwise tensor arrives as five fp16 PLANES (B, 5, L, ld_pl) written by gps_pairwise_locs_planes: the four
//     keys 16 j + 4 g + 0..3 of a lane's query are one 8-byte load per plane and tile -- 5 NT loads per lane and strip,
//     all issued at kernel entry, before the K / V staging and its barrier -- instead of 100 exec-masked 4-byte loads
//     with 64-bit address arithmetic each, chained behind the score MFMAs (the 20 dependent L2 round trips of the old
//     kernels);
//   * the conditioning vector (bias, w_1..w_5) is read as bf16 straight from the packed projection row and its gradient
//     is written as bf16 into the packed gradient row (no fp32 side copies, no copy-back launch);
//   * the spatial term is evaluated in base 2 on whole tiles:  with u = -log2(e) (w_0 + sum_d w_d pl_d),  e = 2^u,
//         log2(clamp(sigmoid(z), 1e-6)) = max(-log2(1 + e), log2(1e-6)),      1 - sigmoid(z) = e / (1 + e)
//     (five mixed-precision fmas, v_exp, v_add, v_log, v_max per score); in the forward kernel it is computed while the
//     K / V tiles are still in flight and becomes the INITIAL VALUE of the score accumulators, so the MFMA adds q.k onto
//     it and the softmax reads finished logits;
//   * outputs leave in the TRANSPOSED orientation (O^T = V^T P^T, dQ^T = K^T dS^T, dV^T = dO^T P, dK^T = Q^T dS): a lane
//     then owns one row and four adjacent columns of the result -- one 8-byte store per 16-column tile instead of four
//     scattered 2-byte stores -- and the per-query softmax normaliser is lane-local;
//   * no transposed LDS copies: every "column" operand is a hardware-transposed read (ds_read_b64_tr_b16) of a row-major
//     tile; the probabilities and dS are parked ROW-major per query (one 8-byte LDS write per tile) and read back the same
//     way; an odd number of 16-token tiles ends in a 16-deep MFMA (v_mfma_f32_16x16x16_bf16) instead of zero padding, which
//     brings the backward kernel's LDS to 51.5 KB at L = 80: three workgroups per CU, the whole grid resident at once;
//   * the backward evaluates every score ONCE (P and dS parked in LDS for the dK / dV pass); delta = rowsum(P dP) stays in
//     fp32 (rowsum(dO * O) from the bf16 forward output costs 1e-2 on the conditioning-vector gradient, see bwd_strip).
// No dropout on this path (the reference's MultiHeadAttentionSpatial takes a `dropout` argument and never applies it,
// transformers.py:188-239); callers that want it use the general kernels of gps_attention.hip.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "gps_hip.h"
#include "gps_device_flags.h"
#include "gps_attention_ex.h"

namespace gps_attn_sp {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;

// register budget of the forward kernel (waves per SIMD it is compiled for); the backward kernel takes it as a template argument
#ifndef GPS_SP_FWD_OCC
#define GPS_SP_FWD_OCC __attribute__((amdgpu_waves_per_eu(NT <= 5 ? 4 : 2, NT <= 5 ? 4 : 2)))
#endif

constexpr int DH = 64;
constexpr int KS = DH + 8;                       // pitch of the row-major K / V / Q / dO tiles (144 B)
constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;
constexpr float kC = 0.125f * kLog2e;            // base-2 logit per unit of q . k
constexpr float kInvC = 8.f * kLn2;              // 1 / kC
constexpr float kClamp2 = -19.931568569324174f;  // log2(1e-6)

struct Params {
  int B, H, L, nt;
  int ld_qkv, ld_o, ld_pl, ld_sw, ld_dqkv, ld_dsw;
  const uint16_t *q, *k, *v;       // (B, L, ld_qkv), head h at column 64 h
  const _Float16 *pl;              // (B, 5, L, ld_pl)
  const uint16_t *sw;              // bf16: row (b, l) at sw + (b L + l) ld_sw, head h at + 6 h
  const uint8_t *mask;             // (B, L), 1 = padded key, or null
  uint16_t *out;                   // (B, L, ld_o)
  float *lse;                      // (B, H, L), natural log
  const uint16_t *dout;            // (B, L, ld_o)
  uint16_t *dq, *dk, *dv;          // (B, L, ld_dqkv)
  uint16_t *dsw;                   // bf16, addressed like sw with ld_dsw
  // plain form only: dropout on the probabilities
  float p_drop;
  unsigned int drop_thr;
  unsigned long long seed;
  const unsigned long long *seed_dev;
};

__device__ __forceinline__ unsigned int pack2(float lo, float hi) {     // v_cvt_pk_bf16_f32: round to nearest even
  const bf16x2_t h = {(__bf16)lo, (__bf16)hi};
  return __builtin_bit_cast(unsigned int, h);
}
__device__ __forceinline__ bf16x8 as_frag(u32x4 v) { return __builtin_bit_cast(bf16x8, v); }
__device__ __forceinline__ u32x4 zero4() { u32x4 z = {0u, 0u, 0u, 0u}; return z; }
__device__ __forceinline__ f32x4 zero_acc() { f32x4 z = {0.f, 0.f, 0.f, 0.f}; return z; }
__device__ __forceinline__ float bf2f(unsigned int bits16) { return __uint_as_float(bits16 << 16); }

__device__ __forceinline__ f32x4 mfma32(bf16x8 a, bf16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma16(u32x2 a, u32x2 b, f32x4 c) {   // 16-deep reduction: k = 4 g + 0..3 per lane
  return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, a), __builtin_bit_cast(s16x4, b), c, 0, 0, 0);
}

__device__ __forceinline__ void block_to_bh(int B, int H, int &b, int &h) {
  const int id = blockIdx.x;
  if ((B & 7) == 0) {       // the 12 heads of a scene on one XCD (block id mod 8): they share its pairwise planes in that L2
    const int xcd = id & 7, slot = id >> 3;
    b = (slot / H) * 8 + xcd;
    h = slot % H;
  } else {
    b = id / H;
    h = id % H;
  }
}

constexpr int kThreads = 320;                    // five waves: one query strip per wave up to 80 tokens, two up to 144

// Two row-major tiles (head h's 64 columns of rows [0, R) of two bf16 matrices; rows >= rows_valid zero) -> LDS [R][KS], in
// two steps: issue() requests every 16-byte piece of BOTH tiles, commit() writes them to LDS.  The kernels put the
// strip's own operand requests between the two, so that everything a wave needs from global memory is one round trip.
template <int R, int THREADS = kThreads>
struct StagePair {
  static constexpr int N = (R * 8 + THREADS - 1) / THREADS;
  u32x4 va[N], vb[N];
  __device__ __forceinline__ void issue(const uint16_t *src_a, int ld_a, const uint16_t *src_b, int ld_b, int rows_valid) {
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const int e = threadIdx.x + i * THREADS, r = e >> 3, ch = e & 7;
      va[i] = zero4();
      vb[i] = zero4();
      if (r < rows_valid) {
        va[i] = *reinterpret_cast<const u32x4 *>(reinterpret_cast<const char *>(src_a) + (unsigned int)(r * ld_a + ch * 8) * 2u);
        vb[i] = *reinterpret_cast<const u32x4 *>(reinterpret_cast<const char *>(src_b) + (unsigned int)(r * ld_b + ch * 8) * 2u);
      }
    }
  }
  __device__ __forceinline__ void commit(uint16_t *dst_a, uint16_t *dst_b) const {
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const int e = threadIdx.x + i * THREADS, r = e >> 3, ch = e & 7;
      if ((R * 8) % THREADS == 0 || r < R) {
        *reinterpret_cast<u32x4 *>(dst_a + r * KS + ch * 8) = va[i];
        *reinterpret_cast<u32x4 *>(dst_b + r * KS + ch * 8) = vb[i];
      }
    }
  }
};

// hardware-transposed read: the 16-lane group of `lane` gets rows row0 .. row0 + 3 of columns col0 .. col0 + 15 of a
// row-major bf16 tile, lane i (= lane & 15) receiving column col0 + i (4 values = 2 dwords)
__device__ __forceinline__ u32x2 tr4(const uint16_t *tile, int pitch, int row0, int col0, int lane) {
  const int i = lane & 15;
  const uint16_t *p = tile + (row0 + (i >> 2)) * pitch + col0 + 4 * (i & 3);
  const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)p);
  return __builtin_bit_cast(u32x2, v);
}
// MFMA operand (A: row = lane & 15 of M^T; B: column = lane & 15 of M) holding M[rows][col0 + (lane & 15)] for the eight rows
//   permuted:  32 c + 4 g + 0..3, 32 c + 16 + 4 g + 0..3   (the K order of pack_tiles: D fragments of two adjacent tiles)
//   natural:   32 c + 8 g + 0..7
__device__ __forceinline__ bf16x8 tr_frag_perm(const uint16_t *tile, int pitch, int c, int col0, int lane) {
  const int g = lane >> 4;
  const u32x2 lo = tr4(tile, pitch, 32 * c + 4 * g, col0, lane), hi = tr4(tile, pitch, 32 * c + 16 + 4 * g, col0, lane);
  const u32x4 v = {lo[0], lo[1], hi[0], hi[1]};
  return as_frag(v);
}
__device__ __forceinline__ bf16x8 tr_frag_nat(const uint16_t *tile, int pitch, int c, int col0, int lane) {
  const int g = lane >> 4;
  const u32x2 lo = tr4(tile, pitch, 32 * c + 8 * g, col0, lane), hi = tr4(tile, pitch, 32 * c + 8 * g + 4, col0, lane);
  const u32x4 v = {lo[0], lo[1], hi[0], hi[1]};
  return as_frag(v);
}
__device__ __forceinline__ bf16x8 pack_tiles(const f32x4 &a, const f32x4 &b) {
  const u32x4 v = {pack2(a[0], a[1]), pack2(a[2], a[3]), pack2(b[0], b[1]), pack2(b[2], b[3])};
  return as_frag(v);
}
__device__ __forceinline__ u32x2 pack_tile(const f32x4 &a) {
  const u32x2 v = {pack2(a[0], a[1]), pack2(a[2], a[3])};
  return v;
}

__device__ __forceinline__ float xor_max_g(float v) {   // across the 4 lane groups (same lane & 15)
  v = fmaxf(v, __shfl_xor(v, 16, 64));
  return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float xor_sum_g(float v) {
  v += __shfl_xor(v, 16, 64);
  return v + __shfl_xor(v, 32, 64);
}

// what a query strip reads from global memory for the spatial term: the (pre-scaled) conditioning vector of the lane's
// query and the fp16 planes of (query, keys 16 j + 4 g + 0..3) for every tile
template <int NT>
struct Spatial {
  unsigned int wraw[3];   // (bias, w_1..w_5) as loaded: six bf16
  u32x2 pl[5][NT];        // 4 halves each
};
// -log2(e) * (bias, w_1..w_5); unpacked where it is used, so that nothing waits on the load at request time
template <int NT>
__device__ __forceinline__ void cond_vector(const Spatial<NT> &S, float (&w)[6]) {
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    w[2 * i] = -kLog2e * bf2f(S.wraw[i] & 0xFFFFu);
    w[2 * i + 1] = -kLog2e * __uint_as_float(S.wraw[i] & 0xFFFF0000u);
  }
}
// planes of tiles [J0, J1) of the lane's query row
template <int NT, int J0, int J1>
__device__ __forceinline__ void load_planes(const Params &P, int b, int qc, int g, Spatial<NT> &S) {
  // uniform (scalar) bases + 32-bit lane offsets: one address register per load instead of a 64-bit pair
  const unsigned int plane = (unsigned int)(P.L * P.ld_pl) * 2u;              // bytes
  const char *base = reinterpret_cast<const char *>(P.pl) + (size_t)b * 5 * plane;
  const unsigned int rowoff = (unsigned int)(qc * P.ld_pl) * 2u;
#pragma unroll
  for (int j = J0; j < J1; ++j) {
    const unsigned int off = rowoff + 2u * (unsigned int)min(16 * j + 4 * g, P.ld_pl - 4);   // past the row: in-bounds columns (masked keys)
#pragma unroll
    for (int d = 0; d < 5; ++d) S.pl[d][j] = *reinterpret_cast<const u32x2 *>(base + d * plane + off);
  }
}
template <int NT>
__device__ __forceinline__ void load_cond(const Params &P, int b, int h, int qc, Spatial<NT> &S) {
  const char *wbase = reinterpret_cast<const char *>(P.sw + (size_t)b * P.L * P.ld_sw + h * 6);
  const unsigned int woff = (unsigned int)(qc * P.ld_sw) * 2u;
  S.wraw[0] = *reinterpret_cast<const unsigned int *>(wbase + woff);
  S.wraw[1] = *reinterpret_cast<const unsigned int *>(wbase + woff + 4u);
  S.wraw[2] = *reinterpret_cast<const unsigned int *>(wbase + woff + 8u);
}
template <int NT>
__device__ __forceinline__ void load_spatial(const Params &P, int b, int h, int qc, int g, Spatial<NT> &S) {
  load_cond<NT>(P, b, h, qc, S);
  load_planes<NT, 0, NT>(P, b, qc, g, S);
}
// one 16-byte fragment load: row `row` (pitch ld elements) of a bf16 matrix whose head block starts at `base` (uniform)
__device__ __forceinline__ u32x4 load_frag(const uint16_t *base, int row, int ld, int col) {
  return *reinterpret_cast<const u32x4 *>(reinterpret_cast<const char *>(base) + (unsigned int)(row * ld + col) * 2u);
}
// u = -log2(e) z of (query, key 16 j + 4 g + r)
template <int NT>
__device__ __forceinline__ float spatial_u(const Spatial<NT> &S, const float (&w)[6], int j, int r) {
  float u = w[0];
#pragma unroll
  for (int d = 0; d < 5; ++d) u = fmaf((float)__builtin_bit_cast(f16x4, S.pl[d][j])[r], w[1 + d], u);
  return u;
}

// ==========================================================================================
// forward
// ==========================================================================================
template <int NT>
struct FwdStrip {
  bf16x8 bq[2];
  Spatial<NT> S;
  f32x4 acc[NT];
};
template <int NT>
__device__ __forceinline__ void fwd_request(const Params &P, const uint16_t *qb, int b, int h, int s, int m, int g, FwdStrip<NT> &F) {
  const int qc = min(16 * s + m, P.L - 1);        // rows past L: any valid row (their results are never stored)
  load_spatial<NT>(P, b, h, qc, g, F.S);
#pragma unroll
  for (int c = 0; c < 2; ++c) F.bq[c] = as_frag(load_frag(qb, qc, P.ld_qkv, 32 * c + 8 * g));
}
// spatial term -> initial value of the score accumulators, in units of q . k (needs no MFMA result)
template <int NT>
__device__ __forceinline__ void fwd_bias(FwdStrip<NT> &F) {
  float w[6];
  cond_vector<NT>(F.S, w);
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float e = __builtin_amdgcn_exp2f(spatial_u<NT>(F.S, w, j, r));
      const float l2 = __builtin_amdgcn_logf(1.f + e);            // v_log_f32 = log2
      F.acc[j][r] = fmaxf(-l2, kClamp2) * kInvC;
    }
}
template <int NT>
__device__ __forceinline__ void fwd_finish(const Params &P, const uint16_t *Ks, const uint16_t *Vs, const float *mbs, int b, int h,
                                           int s, int lane, FwdStrip<NT> &F) {
  const int m = lane & 15, g = lane >> 4, L = P.L;
  const int qi = 16 * s + m;
  f32x4 (&acc)[NT] = F.acc;
  // S^T tiles on top of the spatial term: acc[j][r] = (spatial + key term) / c + <q_qi, k_t>,  t = 16 j + 4 g + r
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const f32x4 kt = *reinterpret_cast<const f32x4 *>(mbs + 16 * j + 4 * g);
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[j][r] += kt[r];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const u32x4 a = *reinterpret_cast<const u32x4 *>(Ks + (16 * j + m) * KS + 32 * c + 8 * g);
      acc[j] = mfma32(as_frag(a), F.bq[c], acc[j]);
    }
  }
  float mx = -INFINITY;
#pragma unroll
  for (int j = 0; j < NT; ++j) mx = fmaxf(fmaxf(mx, fmaxf(acc[j][0], acc[j][1])), fmaxf(acc[j][2], acc[j][3]));
  mx = xor_max_g(mx);
  const float mxc = mx * kC;                      // all keys masked: -inf -> NaN row below, like torch
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float p = __builtin_amdgcn_exp2f(fmaf(acc[j][r], kC, -mxc));
      acc[j][r] = p;
      sum += p;
    }
  sum = xor_sum_g(sum);
  if (g == 0 && qi < L) P.lse[((size_t)b * P.H + h) * L + qi] = (mxc + __builtin_amdgcn_logf(sum)) * kLn2;
  const float inv = __builtin_amdgcn_rcpf(sum);
  // O^T strip = V^T P^T: o[n][r] = O[query qi][d = 16 n + 4 g + r] -- the lane's own query, four adjacent columns
  f32x4 o[4];
#pragma unroll
  for (int n = 0; n < 4; ++n) o[n] = zero_acc();
#pragma unroll
  for (int c = 0; c < NT / 2; ++c) {
    const bf16x8 pb = pack_tiles(acc[2 * c], acc[2 * c + 1]);
#pragma unroll
    for (int n = 0; n < 4; ++n) o[n] = mfma32(tr_frag_perm(Vs, KS, c, 16 * n, lane), pb, o[n]);
  }
  if (NT & 1) {
    const u32x2 pb = pack_tile(acc[NT - 1]);
#pragma unroll
    for (int n = 0; n < 4; ++n) o[n] = mfma16(tr4(Vs, KS, 16 * (NT - 1) + 4 * g, 16 * n, lane), pb, o[n]);
  }
  if (qi < L) {
    uint16_t *op = P.out + ((size_t)b * L + qi) * P.ld_o + h * DH + 4 * g;
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      const u32x2 v = {pack2(o[n][0] * inv, o[n][1] * inv), pack2(o[n][2] * inv, o[n][3] * inv)};
      *reinterpret_cast<u32x2 *>(op + 16 * n) = v;
    }
  }
}

template <int NT>
__global__ __launch_bounds__(kThreads) GPS_SP_FWD_OCC void fwd_kernel(const Params P) {
  constexpr int R = NT * 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint16_t *Ks = reinterpret_cast<uint16_t *>(smem);       // [R][KS]
  uint16_t *Vs = Ks + R * KS;                               // [R][KS]
  float *mbs = reinterpret_cast<float *>(Vs + R * KS);      // [R] additive key term: 0, or -inf (padded / past L)

  int b, h;
  block_to_bh(P.B, P.H, b, h);
  const int L = P.L, nt = P.nt;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int m = lane & 15, g = lane >> 4;
  const size_t row0 = (size_t)b * L;
  const uint16_t *qb = P.q + row0 * P.ld_qkv + h * DH;
  const uint16_t *kb = P.k + row0 * P.ld_qkv + h * DH;
  const uint16_t *vb = P.v + row0 * P.ld_qkv + h * DH;

  // first strip of the wave: every global operand is requested before the K / V staging, the spatial term is evaluated
  // while the tiles are still in flight
  FwdStrip<NT> F;
  StagePair<R> st;
  const bool live = wave < nt;
  st.issue(kb, P.ld_qkv, vb, P.ld_qkv, L);
  if (live) fwd_request<NT>(P, qb, b, h, wave, m, g, F);
  st.commit(Ks, Vs);
  for (int t = threadIdx.x; t < R; t += kThreads) mbs[t] = (t < L && !(P.mask && P.mask[row0 + t])) ? 0.f : -INFINITY;
  if (live) fwd_bias<NT>(F);
  __syncthreads();
  if (live) fwd_finish<NT>(P, Ks, Vs, mbs, b, h, wave, lane, F);
  if (NT > 5) {                                   // second strip (rows above 80 tokens)
    const int s = wave + 5;
    if (s < nt) {
      fwd_request<NT>(P, qb, b, h, s, m, g, F);
      fwd_bias<NT>(F);
      fwd_finish<NT>(P, Ks, Vs, mbs, b, h, s, lane, F);
    }
  }
}

// ==========================================================================================
// backward
// ==========================================================================================
template <int NT>
constexpr int kBwdEarlyTiles = NT <= 5 ? 3 : NT;       // planes of that many tiles are requested before the staging barrier
template <int NT>
struct BwdStrip {
  bf16x8 bq[2], bdo[2];
  Spatial<NT> S;
  float lse2;             // natural log; +inf for queries past L
};
template <int NT>
__device__ __forceinline__ void bwd_request(const Params &P, const uint16_t *qb, const uint16_t *dob, const float *lse, int b, int h, int s, int m, int g, BwdStrip<NT> &F) {
  const int qi = 16 * s + m, qc = min(qi, P.L - 1);
  load_cond<NT>(P, b, h, qc, F.S);
  load_planes<NT, 0, kBwdEarlyTiles<NT>>(P, b, qc, g, F.S);
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    F.bq[c] = as_frag(load_frag(qb, qc, P.ld_qkv, 32 * c + 8 * g));
    F.bdo[c] = as_frag(load_frag(dob, qc, P.ld_o, 32 * c + 8 * g));
  }
  F.lse2 = qi < P.L ? lse[qc] : INFINITY;                   // natural log here; queries past L: p = 2^(x - inf) = 0
}
// query strip s: all score and dP tiles (MFMA), then two elementwise sweeps -- (A) spatial term, probabilities,
// delta = rowsum(P dP) in fp32, the gate of the spatial gradient; (B) dS, P and dS parked in LDS, d cond-vector -- then
// dQ^T = K^T dS^T from the parked rows of the strip.  delta is NOT taken from rowsum(dO * O): the bf16 roundings of P and O
// in the forward pass put 1e-3 relative on it, which the cancellation in p (dp - delta) turns into 1e-2 on the gradient
// of the conditioning vector (measured against the fp32 formulation).
template <int NT>
__device__ __forceinline__ void bwd_strip(const Params &P, const uint16_t *Ks, const uint16_t *Vs, uint16_t *PS, uint16_t *dSS,
                                          const float *mbs, int b, int h, int s, int lane, BwdStrip<NT> &F) {
  constexpr int TP = NT * 16 + 8;
  const int m = lane & 15, g = lane >> 4, L = P.L;
  const int qi = 16 * s + m;
  // the planes of the later tiles: requested now, consumed after the MFMA section (register budget)
  load_planes<NT, kBwdEarlyTiles<NT>, NT>(P, b, min(qi, L - 1), g, F.S);
  f32x4 pr[NT], dp[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    pr[j] = zero_acc();
    dp[j] = zero_acc();
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const u32x4 a = *reinterpret_cast<const u32x4 *>(Ks + (16 * j + m) * KS + 32 * c + 8 * g);
      const u32x4 av = *reinterpret_cast<const u32x4 *>(Vs + (16 * j + m) * KS + 32 * c + 8 * g);
      pr[j] = mfma32(as_frag(a), F.bq[c], pr[j]);       // S^T
      dp[j] = mfma32(as_frag(av), F.bdo[c], dp[j]);     // (dO V^T)^T
    }
    __builtin_amdgcn_sched_barrier(0);            // one tile's fragments in flight, not all 4 NT of them (registers)
  }
  float w[6], dw[6];
  cond_vector<NT>(F.S, w);
  const float lse2 = F.lse2 * kLog2e;
  unsigned int gate2[NT][2];                       // the gate e / (1 + e) of the spatial gradient, bf16 pairs
  float delta = 0.f;
#pragma unroll
  for (int j = 0; j < NT; ++j) {                   // sweep A
    const f32x4 kt = *reinterpret_cast<const f32x4 *>(mbs + 16 * j + 4 * g);
    float gt[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float e = __builtin_amdgcn_exp2f(spatial_u<NT>(F.S, w, j, r));
      const float ope = 1.f + e;
      const float bias2 = fmaxf(-__builtin_amdgcn_logf(ope), kClamp2);
      const float p = __builtin_amdgcn_exp2f(fmaf(pr[j][r], kC, bias2) + (kt[r] - lse2));
      // d/dz log(clamp(sigmoid z, 1e-6)) = 1 - sigmoid z = e / (1 + e) where sigmoid z > 1e-6, else 0
      gt[r] = ope < 1e6f ? e * __builtin_amdgcn_rcpf(ope) : 0.f;
      pr[j][r] = p;
      delta = fmaf(p, dp[j][r], delta);
    }
    gate2[j][0] = pack2(gt[0], gt[1]);
    gate2[j][1] = pack2(gt[2], gt[3]);
    __builtin_amdgcn_sched_barrier(0);
  }
  delta = xor_sum_g(delta);
#pragma unroll
  for (int d = 0; d < 6; ++d) dw[d] = 0.f;
  uint16_t *prow = PS + (16 * s + m) * TP + 4 * g, *drow = dSS + (16 * s + m) * TP + 4 * g;
#pragma unroll
  for (int j = 0; j < NT; ++j) {                   // sweep B
    // the fp16 planes are used by both sweeps: without this the compiler converts all 20 NT values to fp32 once and
    // keeps them (20 NT more registers) instead of issuing mixed-precision fmas on the packed halves again
#pragma unroll
    for (int d = 0; d < 5; ++d) asm volatile("" : "+v"(F.S.pl[d][j]));
    f32x4 dsj;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float dl = pr[j][r] * (dp[j][r] - delta);                          // d loss / d logit
      const unsigned int gw = gate2[j][r >> 1];
      const float gate = (r & 1) ? __uint_as_float(gw & 0xFFFF0000u) : bf2f(gw & 0xFFFFu);
      const float dz = dl * gate;
      dw[0] += dz;
#pragma unroll
      for (int d = 0; d < 5; ++d) dw[1 + d] = fmaf((float)__builtin_bit_cast(f16x4, F.S.pl[d][j])[r], dz, dw[1 + d]);
      dsj[r] = dl;
    }
    *reinterpret_cast<u32x2 *>(prow + 16 * j) = pack_tile(pr[j]);
    *reinterpret_cast<u32x2 *>(drow + 16 * j) = pack_tile(dsj);
    __builtin_amdgcn_sched_barrier(0);
  }
  // gradient of the conditioning vector: sum over the lane groups, one 12-byte bf16 store per (query, head)
#pragma unroll
  for (int d = 0; d < 6; ++d) dw[d] = xor_sum_g(dw[d]);
  if (g == 0 && qi < L) {
    unsigned int *wp = reinterpret_cast<unsigned int *>(P.dsw + ((size_t)b * L + qi) * P.ld_dsw + h * 6);
    wp[0] = pack2(dw[0], dw[1]);
    wp[1] = pack2(dw[2], dw[3]);
    wp[2] = pack2(dw[4], dw[5]);
  }
  // dQ^T strip = K^T dS^T: dS of the lane's query, keys 32 c + 8 g + 0..7, straight from the row this wave just parked
  // (same wave, LDS operations of a wave complete in order), K^T by transposed reads in the same key order
  f32x4 o[4];
#pragma unroll
  for (int n = 0; n < 4; ++n) o[n] = zero_acc();
  const uint16_t *dq_row = dSS + (16 * s + m) * TP + 8 * g;
#pragma unroll
  for (int c = 0; c < NT / 2; ++c) {
    const bf16x8 db = as_frag(*reinterpret_cast<const u32x4 *>(dq_row + 32 * c));
#pragma unroll
    for (int n = 0; n < 4; ++n) o[n] = mfma32(tr_frag_nat(Ks, KS, c, 16 * n, lane), db, o[n]);
  }
  if (NT & 1) {
    const u32x2 db = *reinterpret_cast<const u32x2 *>(dSS + (16 * s + m) * TP + 16 * (NT - 1) + 4 * g);
#pragma unroll
    for (int n = 0; n < 4; ++n) o[n] = mfma16(tr4(Ks, KS, 16 * (NT - 1) + 4 * g, 16 * n, lane), db, o[n]);
  }
  if (qi < L) {
    uint16_t *op = P.dq + ((size_t)b * L + qi) * P.ld_dqkv + h * DH + 4 * g;
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      const u32x2 v = {pack2(o[n][0] * 0.125f, o[n][1] * 0.125f), pack2(o[n][2] * 0.125f, o[n][3] * 0.125f)};
      *reinterpret_cast<u32x2 *>(op + 16 * n) = v;
    }
  }
}

template <int NT, int OCC>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(OCC, OCC))) void bwd_kernel(const Params P) {
  constexpr int R = NT * 16;
  constexpr int TP = R + 8;                                 // pitch of the parked probability / dS tiles
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint16_t *Ks = reinterpret_cast<uint16_t *>(smem);       // [R][KS]   pass 2: Q rows
  uint16_t *Vs = Ks + R * KS;                               // [R][KS]   pass 2: dO rows
  uint16_t *PS = Vs + R * KS;                               // [R][TP]   P[query][key]
  uint16_t *dSS = PS + R * TP;                              // [R][TP]   dS[query][key] (before the 1/8 of the logits)
  float *mbs = reinterpret_cast<float *>(dSS + R * TP);     // [R]

  int b, h;
  block_to_bh(P.B, P.H, b, h);
  const int L = P.L, nt = P.nt;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int m = lane & 15, g = lane >> 4;
  const size_t row0 = (size_t)b * L;
  const uint16_t *qb = P.q + row0 * P.ld_qkv + h * DH;
  const uint16_t *kb = P.k + row0 * P.ld_qkv + h * DH;
  const uint16_t *vb = P.v + row0 * P.ld_qkv + h * DH;
  const uint16_t *dob = P.dout + row0 * P.ld_o + h * DH;
  const float *lse = P.lse + ((size_t)b * P.H + h) * L;

  // ---------------- pass 1: query strips -> dQ, d cond-vector, P and dS tiles ----------------
  BwdStrip<NT> F;
  StagePair<R> st;
  const bool live = wave < nt;
  st.issue(kb, P.ld_qkv, vb, P.ld_qkv, L);
  if (live) bwd_request<NT>(P, qb, dob, lse, b, h, wave, m, g, F);
  st.commit(Ks, Vs);
  for (int t = threadIdx.x; t < R; t += kThreads) mbs[t] = (t < L && !(P.mask && P.mask[row0 + t])) ? 0.f : -INFINITY;
  if (nt < NT) {            // query tiles no strip writes are still read by pass 2: zeros
    u32x4 *z = reinterpret_cast<u32x4 *>(PS + nt * 16 * TP);
    for (int e = threadIdx.x; e < (NT - nt) * 16 * TP / 8; e += kThreads) z[e] = zero4();
    z = reinterpret_cast<u32x4 *>(dSS + nt * 16 * TP);
    for (int e = threadIdx.x; e < (NT - nt) * 16 * TP / 8; e += kThreads) z[e] = zero4();
  }
  __syncthreads();
  if (live) bwd_strip<NT>(P, Ks, Vs, PS, dSS, mbs, b, h, wave, lane, F);
  if (NT > 5) {
    const int s = wave + 5;
    if (s < nt) {
      bwd_request<NT>(P, qb, dob, lse, b, h, s, m, g, F);
      bwd_strip<NT>(P, Ks, Vs, PS, dSS, mbs, b, h, s, lane, F);
    }
  }
  st.issue(qb, P.ld_qkv, dob, P.ld_o, L);       // pass 2's tiles: requested before the barrier, written behind it
  __syncthreads();   // P / dS complete; K / V tiles dead: the same storage takes Q and dO (row-major as well)
  st.commit(Ks, Vs);
  __syncthreads();

  // ---------------- pass 2: key strips, MFMA only -> dK^T = Q^T dS, dV^T = dO^T P ----------------
  for (int js = wave; js < nt; js += 5) {
    f32x4 dk[4], dv[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      dk[n] = zero_acc();
      dv[n] = zero_acc();
    }
#pragma unroll
    for (int c = 0; c < NT / 2; ++c) {
      const bf16x8 bp = tr_frag_nat(PS, TP, c, 16 * js, lane);       // P[queries 32 c + 8 g + 0..7][key 16 js + m]
      const bf16x8 bs = tr_frag_nat(dSS, TP, c, 16 * js, lane);
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        dv[n] = mfma32(tr_frag_nat(Vs, KS, c, 16 * n, lane), bp, dv[n]);    // Vs holds dO here
        dk[n] = mfma32(tr_frag_nat(Ks, KS, c, 16 * n, lane), bs, dk[n]);    // Ks holds Q
      }
    }
    if (NT & 1) {
      const int q0 = 16 * (NT - 1) + 4 * g;
      const u32x2 bp = tr4(PS, TP, q0, 16 * js, lane), bs = tr4(dSS, TP, q0, 16 * js, lane);
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        dv[n] = mfma16(tr4(Vs, KS, q0, 16 * n, lane), bp, dv[n]);
        dk[n] = mfma16(tr4(Ks, KS, q0, 16 * n, lane), bs, dk[n]);
      }
    }
    const int t = 16 * js + m;                   // dk[n][r] = dK[key t][d = 16 n + 4 g + r]
    if (t < L) {
      uint16_t *pk = P.dk + (row0 + t) * P.ld_dqkv + h * DH + 4 * g;
      uint16_t *pv = P.dv + (row0 + t) * P.ld_dqkv + h * DH + 4 * g;
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        const u32x2 vk = {pack2(dk[n][0] * 0.125f, dk[n][1] * 0.125f), pack2(dk[n][2] * 0.125f, dk[n][3] * 0.125f)};
        const u32x2 vv = {pack2(dv[n][0], dv[n][1]), pack2(dv[n][2], dv[n][3])};
        *reinterpret_cast<u32x2 *>(pk + 16 * n) = vk;
        *reinterpret_cast<u32x2 *>(pv + 16 * n) = vv;
      }
    }
  }
}

template <int NT>
constexpr size_t fwd_lds() { return (size_t)2 * (2 * NT * 16 * KS) + (size_t)4 * NT * 16; }
template <int NT>
constexpr size_t bwd_lds() { return (size_t)2 * (2 * NT * 16 * KS + 2 * NT * 16 * (NT * 16 + 8)) + (size_t)4 * NT * 16; }

// The backward kernel at L <= 80 is built for two register budgets: 4 waves per SIMD (128 registers, a few long-lived
// addresses spilled: three workgroups per CU, the 768 workgroups of B = 64 resident at once) and 3 (168 registers, no
// spill, two workgroups per CU).  GPS_ATTN_SP_BWD_OCC=3|4 picks one (default 4; tools/attn_bench.py measures both).
static int bwd_occ() {
  static int occ = 0;
  if (!occ) {
    const char *e = getenv("GPS_ATTN_SP_BWD_OCC");
    occ = (e && atoi(e) == 3) ? 3 : 4;
  }
  return occ;
}

template <int NT>
int launch(const Params &P, bool backward, hipStream_t s) {
  const dim3 grid(P.B * P.H), block(kThreads);
  const size_t lds = backward ? bwd_lds<NT>() : fwd_lds<NT>();
  constexpr int kOccBig = NT <= 5 ? 4 : 2;
  const bool alt = backward && NT <= 5 && bwd_occ() == 3;
  // one flag per INSTANTIATION that can be launched below (forward | backward | backward at the alternative occupancy)
  static gps_dev::PerDevice<bool, 3> granted_dev;
  bool *granted = granted_dev.row();
  const int which = !backward ? 0 : (alt ? 2 : 1);
  if (lds > 64 * 1024 && !granted[which]) {
    const void *fn = !backward ? (const void *)&fwd_kernel<NT>
                     : alt     ? (const void *)&bwd_kernel<NT, (NT <= 5 ? 3 : kOccBig)>
                               : (const void *)&bwd_kernel<NT, kOccBig>;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return GPS_ERR_LAUNCH;
    granted[which] = true;
  }
  if (!backward) hipLaunchKernelGGL((fwd_kernel<NT>), grid, block, lds, s, P);
  else if (alt) hipLaunchKernelGGL((bwd_kernel<NT, (NT <= 5 ? 3 : kOccBig)>), grid, block, lds, s, P);
  else hipLaunchKernelGGL((bwd_kernel<NT, kOccBig>), grid, block, lds, s, P);
  return hipGetLastError() == hipSuccess ? GPS_OK : GPS_ERR_LAUNCH;
}


// ==========================================================================================
// PLAIN form (no pairwise term) for short fixed-length rows -- the 130-token joint sequences of the unified encoder
// (nn.MultiheadAttention with key_padding_mask and dropout on the probabilities, modules/layers/transformers.py:141):
// the same single-sweep structure (K / V resident, one query strip per wave, P and dS parked in LDS, dK / dV as pure
// MFMA), dropout from the streaming kernels' stream (one hash per (query, key pair): gps_attention.hip pair_rng), so a
// forward of this family pairs with a backward of any other.
// ==========================================================================================
__device__ __forceinline__ unsigned int mix32(unsigned int x) {
  x ^= x >> 16;
  x *= 0x21F0AAADu;
  x ^= x >> 15;
  x *= 0x735A2D97u;
  x ^= x >> 15;
  return x;
}
__device__ __forceinline__ unsigned int seed_fold(unsigned long long seed) {
  return mix32((unsigned int)seed ^ mix32((unsigned int)(seed >> 32) + 0x9E3779B9u));
}
__device__ __forceinline__ unsigned int pair_rng(unsigned int seedmix, unsigned int row_pair_base, int t) {
  return mix32((row_pair_base + (unsigned int)(t >> 1)) ^ seedmix);
}
struct Drop {                 // dropout state of a query row
  bool on;
  float keep_scale;
  unsigned int seedmix, thr16, rp;
  __device__ __forceinline__ void init(const Params &P, int b, int h, int qi) {
    on = P.drop_thr != 0u;
    keep_scale = on ? 1.f / (1.f - P.p_drop) : 1.f;
    seedmix = on ? seed_fold(P.seed + (P.seed_dev ? *P.seed_dev : 0ull)) : 0u;
    thr16 = P.drop_thr >> 16;
    rp = (((unsigned int)b * P.H + h) * P.L + qi) * (unsigned int)((P.L + 1) >> 1);
  }
  // keep flags of keys t0 .. t0 + 3 (t0 a multiple of 4): two hashes
  __device__ __forceinline__ void keep4(int t0, bool (&k)[4]) const {
    const unsigned int r01 = pair_rng(seedmix, rp, t0), r23 = pair_rng(seedmix, rp, t0 + 2);
    k[0] = (r01 & 0xFFFFu) >= thr16;
    k[1] = (r01 >> 16) >= thr16;
    k[2] = (r23 & 0xFFFFu) >= thr16;
    k[3] = (r23 >> 16) >= thr16;
  }
};

template <int NT>
__global__ __launch_bounds__(64 * NT) void pfwd_kernel(const Params P) {
  constexpr int R = NT * 16, THREADS = 64 * NT;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint16_t *Ks = reinterpret_cast<uint16_t *>(smem);
  uint16_t *Vs = Ks + R * KS;
  float *mbs = reinterpret_cast<float *>(Vs + R * KS);
  int b, h;
  block_to_bh(P.B, P.H, b, h);
  const int L = P.L, nt = P.nt;
  const int lane = threadIdx.x & 63, s = threadIdx.x >> 6;       // one query strip per wave
  const int m = lane & 15, g = lane >> 4;
  const size_t row0 = (size_t)b * L;
  const uint16_t *qb = P.q + row0 * P.ld_qkv + h * DH;
  const uint16_t *kb = P.k + row0 * P.ld_qkv + h * DH;
  const uint16_t *vb = P.v + row0 * P.ld_qkv + h * DH;
  const int qi = 16 * s + m, qc = min(qi, L - 1);

  StagePair<R, THREADS> st;
  st.issue(kb, P.ld_qkv, vb, P.ld_qkv, L);
  bf16x8 bq[2];
#pragma unroll
  for (int c = 0; c < 2; ++c) bq[c] = as_frag(load_frag(qb, qc, P.ld_qkv, 32 * c + 8 * g));
  st.commit(Ks, Vs);
  for (int t = threadIdx.x; t < R; t += THREADS) mbs[t] = (t < L && !(P.mask && P.mask[row0 + t])) ? 0.f : -INFINITY;
  __syncthreads();
  if (s >= nt) return;
  Drop dr;
  dr.init(P, b, h, qi);
  f32x4 acc[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    acc[j] = *reinterpret_cast<const f32x4 *>(mbs + 16 * j + 4 * g);       // key term (0 / -inf) under the scores
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const u32x4 a = *reinterpret_cast<const u32x4 *>(Ks + (16 * j + m) * KS + 32 * c + 8 * g);
      acc[j] = mfma32(as_frag(a), bq[c], acc[j]);
    }
  }
  float mx = -INFINITY;
#pragma unroll
  for (int j = 0; j < NT; ++j) mx = fmaxf(fmaxf(mx, fmaxf(acc[j][0], acc[j][1])), fmaxf(acc[j][2], acc[j][3]));
  mx = xor_max_g(mx);
  const float mxc = mx * kC;
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < NT; ++j) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float p = __builtin_amdgcn_exp2f(fmaf(acc[j][r], kC, -mxc));
      acc[j][r] = p;
      sum += p;
    }
    if (dr.on) {
      bool k[4];
      dr.keep4(16 * j + 4 * g, k);
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[j][r] = k[r] ? acc[j][r] : 0.f;
    }
  }
  sum = xor_sum_g(sum);
  if (g == 0 && qi < L) P.lse[((size_t)b * P.H + h) * L + qi] = (mxc + __builtin_amdgcn_logf(sum)) * kLn2;
  const float sc = dr.keep_scale * __builtin_amdgcn_rcpf(sum);
  f32x4 o[4];
#pragma unroll
  for (int n = 0; n < 4; ++n) o[n] = zero_acc();
#pragma unroll
  for (int c = 0; c < NT / 2; ++c) {
    const bf16x8 pb = pack_tiles(acc[2 * c], acc[2 * c + 1]);
#pragma unroll
    for (int n = 0; n < 4; ++n) o[n] = mfma32(tr_frag_perm(Vs, KS, c, 16 * n, lane), pb, o[n]);
  }
  if (NT & 1) {
    const u32x2 pb = pack_tile(acc[NT - 1]);
#pragma unroll
    for (int n = 0; n < 4; ++n) o[n] = mfma16(tr4(Vs, KS, 16 * (NT - 1) + 4 * g, 16 * n, lane), pb, o[n]);
  }
  if (qi < L) {
    uint16_t *op = P.out + (row0 + qi) * P.ld_o + h * DH + 4 * g;
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      const u32x2 v = {pack2(o[n][0] * sc, o[n][1] * sc), pack2(o[n][2] * sc, o[n][3] * sc)};
      *reinterpret_cast<u32x2 *>(op + 16 * n) = v;
    }
  }
}

template <int NT>
__global__ __launch_bounds__(64 * NT) void pbwd_kernel(const Params P) {
  constexpr int R = NT * 16, TP = R + 8, THREADS = 64 * NT;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint16_t *Ks = reinterpret_cast<uint16_t *>(smem);
  uint16_t *Vs = Ks + R * KS;
  uint16_t *PS = Vs + R * KS;
  uint16_t *dSS = PS + R * TP;
  float *mbs = reinterpret_cast<float *>(dSS + R * TP);
  int b, h;
  block_to_bh(P.B, P.H, b, h);
  const int L = P.L, nt = P.nt;
  const int lane = threadIdx.x & 63, s = threadIdx.x >> 6;
  const int m = lane & 15, g = lane >> 4;
  const size_t row0 = (size_t)b * L;
  const uint16_t *qb = P.q + row0 * P.ld_qkv + h * DH;
  const uint16_t *kb = P.k + row0 * P.ld_qkv + h * DH;
  const uint16_t *vb = P.v + row0 * P.ld_qkv + h * DH;
  const uint16_t *dob = P.dout + row0 * P.ld_o + h * DH;
  const int qi = 16 * s + m, qc = min(qi, L - 1);
  const bool live = s < nt;

  StagePair<R, THREADS> st;
  st.issue(kb, P.ld_qkv, vb, P.ld_qkv, L);
  bf16x8 bq[2], bdo[2];
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    bq[c] = as_frag(load_frag(qb, qc, P.ld_qkv, 32 * c + 8 * g));
    bdo[c] = as_frag(load_frag(dob, qc, P.ld_o, 32 * c + 8 * g));
  }
  float lse2 = qi < L ? P.lse[((size_t)b * P.H + h) * L + qc] : INFINITY;     // queries past L: p = 0
  st.commit(Ks, Vs);
  for (int t = threadIdx.x; t < R; t += THREADS) mbs[t] = (t < L && !(P.mask && P.mask[row0 + t])) ? 0.f : -INFINITY;
  if (nt < NT) {            // query tiles no strip writes are still read by pass 2: zeros
    u32x4 *z = reinterpret_cast<u32x4 *>(PS + nt * 16 * TP);
    for (int e = threadIdx.x; e < (NT - nt) * 16 * TP / 8; e += THREADS) z[e] = zero4();
    z = reinterpret_cast<u32x4 *>(dSS + nt * 16 * TP);
    for (int e = threadIdx.x; e < (NT - nt) * 16 * TP / 8; e += THREADS) z[e] = zero4();
  }
  __syncthreads();
  Drop dr;
  dr.init(P, b, h, qi);
  if (live) {
    lse2 *= kLog2e;
    f32x4 pr[NT], dp[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      pr[j] = zero_acc();
      dp[j] = zero_acc();
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const u32x4 a = *reinterpret_cast<const u32x4 *>(Ks + (16 * j + m) * KS + 32 * c + 8 * g);
        const u32x4 av = *reinterpret_cast<const u32x4 *>(Vs + (16 * j + m) * KS + 32 * c + 8 * g);
        pr[j] = mfma32(as_frag(a), bq[c], pr[j]);         // S^T
        dp[j] = mfma32(as_frag(av), bdo[c], dp[j]);       // (dO V^T)^T
      }
    }
    // sweep A: probabilities; dropped keys keep their probability with the sign flipped (P itself is wanted for dS,
    // the dropped P for dV) and lose their dP; delta = rowsum(P' dP') of the dropped, rescaled pair
    float delta = 0.f;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const f32x4 kt = *reinterpret_cast<const f32x4 *>(mbs + 16 * j + 4 * g);
      bool k[4] = {true, true, true, true};
      if (dr.on) dr.keep4(16 * j + 4 * g, k);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = __builtin_amdgcn_exp2f(fmaf(pr[j][r], kC, kt[r] - lse2));
        const float dpk = k[r] ? dp[j][r] : 0.f;
        delta = fmaf(p, dpk, delta);
        pr[j][r] = k[r] ? p : -p;
        dp[j][r] = dpk;
      }
    }
    delta = xor_sum_g(delta) * dr.keep_scale;
    uint16_t *prow = PS + (16 * s + m) * TP + 4 * g, *drow = dSS + (16 * s + m) * TP + 4 * g;
#pragma unroll
    for (int j = 0; j < NT; ++j) {                 // sweep B
      f32x4 pk, dsj;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        pk[r] = fmaxf(pr[j][r], 0.f);                                          // 1 / (1 - p) goes onto dV below
        dsj[r] = fabsf(pr[j][r]) * fmaf(dp[j][r], dr.keep_scale, -delta);      // 1/8 goes onto dQ / dK below
      }
      *reinterpret_cast<u32x2 *>(prow + 16 * j) = pack_tile(pk);
      *reinterpret_cast<u32x2 *>(drow + 16 * j) = pack_tile(dsj);
    }
    // dQ^T strip = K^T dS^T from the row this wave just parked
    f32x4 o[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) o[n] = zero_acc();
    const uint16_t *dq_row = dSS + (16 * s + m) * TP + 8 * g;
#pragma unroll
    for (int c = 0; c < NT / 2; ++c) {
      const bf16x8 db = as_frag(*reinterpret_cast<const u32x4 *>(dq_row + 32 * c));
#pragma unroll
      for (int n = 0; n < 4; ++n) o[n] = mfma32(tr_frag_nat(Ks, KS, c, 16 * n, lane), db, o[n]);
    }
    if (NT & 1) {
      const u32x2 db = *reinterpret_cast<const u32x2 *>(dSS + (16 * s + m) * TP + 16 * (NT - 1) + 4 * g);
#pragma unroll
      for (int n = 0; n < 4; ++n) o[n] = mfma16(tr4(Ks, KS, 16 * (NT - 1) + 4 * g, 16 * n, lane), db, o[n]);
    }
    if (qi < L) {
      uint16_t *op = P.dq + (row0 + qi) * P.ld_dqkv + h * DH + 4 * g;
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        const u32x2 v = {pack2(o[n][0] * 0.125f, o[n][1] * 0.125f), pack2(o[n][2] * 0.125f, o[n][3] * 0.125f)};
        *reinterpret_cast<u32x2 *>(op + 16 * n) = v;
      }
    }
  }
  st.issue(qb, P.ld_qkv, dob, P.ld_o, L);
  __syncthreads();
  st.commit(Ks, Vs);
  __syncthreads();
  if (!live) return;
  // pass 2: key strip s, MFMA only
  f32x4 dk[4], dv[4];
#pragma unroll
  for (int n = 0; n < 4; ++n) {
    dk[n] = zero_acc();
    dv[n] = zero_acc();
  }
#pragma unroll
  for (int c = 0; c < NT / 2; ++c) {
    const bf16x8 bp = tr_frag_nat(PS, TP, c, 16 * s, lane);
    const bf16x8 bs = tr_frag_nat(dSS, TP, c, 16 * s, lane);
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      dv[n] = mfma32(tr_frag_nat(Vs, KS, c, 16 * n, lane), bp, dv[n]);
      dk[n] = mfma32(tr_frag_nat(Ks, KS, c, 16 * n, lane), bs, dk[n]);
    }
  }
  if (NT & 1) {
    const int q0 = 16 * (NT - 1) + 4 * g;
    const u32x2 bp = tr4(PS, TP, q0, 16 * s, lane), bs = tr4(dSS, TP, q0, 16 * s, lane);
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      dv[n] = mfma16(tr4(Vs, KS, q0, 16 * n, lane), bp, dv[n]);
      dk[n] = mfma16(tr4(Ks, KS, q0, 16 * n, lane), bs, dk[n]);
    }
  }
  const int t = 16 * s + m;
  if (t < L) {
    const float ks = dr.keep_scale;
    uint16_t *pk = P.dk + (row0 + t) * P.ld_dqkv + h * DH + 4 * g;
    uint16_t *pv = P.dv + (row0 + t) * P.ld_dqkv + h * DH + 4 * g;
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      const u32x2 vk = {pack2(dk[n][0] * 0.125f, dk[n][1] * 0.125f), pack2(dk[n][2] * 0.125f, dk[n][3] * 0.125f)};
      const u32x2 vv = {pack2(dv[n][0] * ks, dv[n][1] * ks), pack2(dv[n][2] * ks, dv[n][3] * ks)};
      *reinterpret_cast<u32x2 *>(pk + 16 * n) = vk;
      *reinterpret_cast<u32x2 *>(pv + 16 * n) = vv;
    }
  }
}

template <int NT>
int launch_plain(const Params &P, bool backward, hipStream_t s) {
  const dim3 grid(P.B * P.H), block(64 * NT);
  const size_t lds = backward ? bwd_lds<NT>() : fwd_lds<NT>();
  static gps_dev::PerDevice<bool, 2> granted_dev;
  bool *granted = granted_dev.row();
  if (lds > 64 * 1024 && !granted[backward ? 1 : 0]) {
    const void *fn = backward ? (const void *)&pbwd_kernel<NT> : (const void *)&pfwd_kernel<NT>;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return GPS_ERR_LAUNCH;
    granted[backward ? 1 : 0] = true;
  }
  if (backward) hipLaunchKernelGGL((pbwd_kernel<NT>), grid, block, lds, s, P);
  else hipLaunchKernelGGL((pfwd_kernel<NT>), grid, block, lds, s, P);
  return hipGetLastError() == hipSuccess ? GPS_OK : GPS_ERR_LAUNCH;
}

}  // namespace gps_attn_sp

namespace gps_attn {

// the plane form of the spatial self-attention call (gps_attn_args.pl_planes): argument checks done by run_ex
int run_spatial_planes(const gps_attn_args *a, bool backward, hipStream_t s) {
  if (a->Lq != a->Lk || a->Lk > 144 || a->p_drop != 0.f || a->dtype != GPS_ATTN_BF16 || a->cu_rows) return GPS_ERR_UNSUPPORTED;
  if (!a->sw16 || a->sw || a->ld_q != a->ld_kv || (a->ld_pl & 3) || a->ld_pl < a->Lk || (a->ld_sw & 1) || a->ld_sw < a->H * 6)
    return GPS_ERR_INVALID_ARGUMENT;
  if (((uintptr_t)a->pl_planes & 7) || ((uintptr_t)a->sw16 & 3)) return GPS_ERR_UNSUPPORTED;
  if (backward && (!a->dsw16 || (a->ld_dsw & 1) || a->ld_dsw < a->H * 6 || ((uintptr_t)a->dsw16 & 3) || a->ld_dq != a->ld_dkv))
    return GPS_ERR_INVALID_ARGUMENT;
  gps_attn_sp::Params P = {};
  P.B = a->B; P.H = a->H; P.L = a->Lk; P.nt = (a->Lk + 15) / 16;
  P.ld_qkv = a->ld_kv; P.ld_o = a->ld_o; P.ld_pl = a->ld_pl; P.ld_sw = a->ld_sw;
  P.q = (const uint16_t *)a->q; P.k = (const uint16_t *)a->k; P.v = (const uint16_t *)a->v;
  P.pl = (const _Float16 *)a->pl_planes; P.sw = (const uint16_t *)a->sw16; P.mask = a->mask;
  P.out = (uint16_t *)a->out; P.lse = a->lse;
  if (backward) {
    P.dout = (const uint16_t *)a->dout; P.dq = (uint16_t *)a->dq; P.dk = (uint16_t *)a->dk; P.dv = (uint16_t *)a->dv;
    P.ld_dqkv = a->ld_dq; P.dsw = (uint16_t *)a->dsw16; P.ld_dsw = a->ld_dsw;
  }
  return P.nt <= 5 ? gps_attn_sp::launch<5>(P, backward, s) : gps_attn_sp::launch<9>(P, backward, s);
}


// plain form (no pairwise term), fixed-length self-attention up to 144 tokens, K / V resident (argument checks by run_ex)
int run_plain_resident(const gps_attn_args *a, bool backward, hipStream_t s) {
  if (a->Lq != a->Lk || a->Lk > 144 || a->dtype != GPS_ATTN_BF16 || a->cu_rows || a->sw || a->pl || a->pl_planes ||
      a->ld_q != a->ld_kv || (backward && a->ld_dq != a->ld_dkv))
    return GPS_ERR_UNSUPPORTED;
  gps_attn_sp::Params P = {};
  P.B = a->B; P.H = a->H; P.L = a->Lk; P.nt = (a->Lk + 15) / 16;
  P.ld_qkv = a->ld_kv; P.ld_o = a->ld_o;
  P.q = (const uint16_t *)a->q; P.k = (const uint16_t *)a->k; P.v = (const uint16_t *)a->v;
  P.mask = a->mask; P.out = (uint16_t *)a->out; P.lse = a->lse;
  P.p_drop = a->p_drop; P.seed = a->seed; P.seed_dev = (const unsigned long long *)a->seed_dev;
  P.drop_thr = a->p_drop > 0.f ? (unsigned int)((double)a->p_drop * 4294967296.0) : 0u;
  if (backward) {
    P.dout = (const uint16_t *)a->dout; P.dq = (uint16_t *)a->dq; P.dk = (uint16_t *)a->dk; P.dv = (uint16_t *)a->dv;
    P.ld_dqkv = a->ld_dq;
  }
  return P.nt <= 5 ? gps_attn_sp::launch_plain<5>(P, backward, s) : gps_attn_sp::launch_plain<9>(P, backward, s);
}

}  // namespace gps_attn
