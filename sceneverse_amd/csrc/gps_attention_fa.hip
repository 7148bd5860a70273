// gps_attention_fa.hip -- the PLAIN attention core (no pairwise term) in block-streaming form, round 5: every text layer
// of BERT (variable-length sentences and captions packed back to back), the joint text + object layers of the unified
// encoder (130 tokens, key-padding mask), cross-attention.  Reference behaviour: nn.MultiheadAttention with
// key_padding_mask and dropout on the probabilities (modules/layers/transformers.py:141; HF BertSelfAttention for the
// text encoder, modules/language/bert.py:21-26).
//
// Why a second streaming family (gps_attention.hip has one): that one gives a workgroup a whole (sequence, head) and
// keeps the sequence's K and V resident in LDS -- 92 KB at 300 tokens, sized for the LONGEST sequence of the launch, one
// workgroup per CU.  A 30-token sentence then occupies a CU like a 300-token caption does, nothing overlaps a workgroup's
// staging round trip, and the two-pass softmax evaluates every score twice.  Here:
//   * a workgroup (4 waves) owns 64 queries (forward, dQ) or 64 keys (dK / dV) of one (sequence, head) and streams the
//     other side through LDS in 64-row blocks, double-buffered: 37 KB whatever the sequence length -> 4 workgroups per CU,
//     work proportional to the sequence's own length (blocks past its end exit at once);
//   * online softmax (running maximum and normaliser per query, lane-local: scores are computed transposed, S^T = K Q^T,
//     so a lane owns one query): every score is evaluated once in the forward pass;
//   * the backward pass is two launches -- dQ per query block (also writes delta = rowsum(dO * O) per query), dK / dV per
//     key block (reads it) -- each with the 64-row blocks of the other side streamed the same way;
//   * results leave in the transposed orientation (O^T = V^T P^T, dQ^T = K^T dS^T, dV^T = dO^T P, dK^T = Q^T dS): 8-byte
//     stores, lane-local normalisation; "column" operands are hardware-transposed LDS reads (ds_read_b64_tr_b16).
// Same dropout stream (one hash per (query, key pair), 16-bit thresholds), same lse, same argument conventions as the
// streaming kernels of gps_attention.hip: forward and backward of the two families can be mixed (tests do).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "gps_hip.h"
#include "gps_attention_ex.h"

namespace gps_attn_fa {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;

constexpr int DH = 64;
constexpr int KS = DH + 8;            // LDS row pitch (144 B)
constexpr int BLK = 64;               // rows per streamed block, and queries / keys per workgroup
constexpr int kThreads = 256;
constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;
constexpr float kC = 0.125f * kLog2e;

struct Params {
  int B, H, L, Lq;                    // L / Lq: capacities of keys / queries (pitch of lse, delta and of the dropout counter)
  int ld_q, ld_kv, ld_o, ld_dq, ld_dkv;
  int nblk;                           // blocks of the OWNED side per sequence: ceil(capacity / 64)
  const uint16_t *q, *k, *v;
  const uint8_t *mask;
  uint16_t *out;
  float *lse;
  const uint16_t *dout;
  uint16_t *dq, *dk, *dv;
  float *delta;                       // (B, H, Lq) workspace: written by the dQ launch, read by the dK / dV launch
  float p_drop;
  unsigned int drop_thr;
  unsigned long long seed;
  const unsigned long long *seed_dev;
  const int *seq_order, *q_limit, *cu_rows;
};

__device__ __forceinline__ unsigned int pack2(float lo, float hi) {
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
__device__ __forceinline__ bf16x8 pack_tiles(const f32x4 &a, const f32x4 &b) {
  const u32x4 v = {pack2(a[0], a[1]), pack2(a[2], a[3]), pack2(b[0], b[1]), pack2(b[2], b[3])};
  return as_frag(v);
}
__device__ __forceinline__ float xor_max_g(float v) {
  v = fmaxf(v, __shfl_xor(v, 16, 64));
  return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float xor_sum_g(float v) {
  v += __shfl_xor(v, 16, 64);
  return v + __shfl_xor(v, 32, 64);
}
// the dropout stream of gps_attention.hip's streaming kernels, bit for bit
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

__device__ __forceinline__ u32x2 tr4(const uint16_t *tile, int row0, int col0, int lane) {
  const int i = lane & 15;
  const uint16_t *p = tile + (row0 + (i >> 2)) * KS + col0 + 4 * (i & 3);
  const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)p);
  return __builtin_bit_cast(u32x2, v);
}
// MFMA operand holding M[rows][col0 + (lane & 15)] for the rows 32 c + 4 g + 0..3, 32 c + 16 + 4 g + 0..3 of a row-major
// [64][KS] block: the K order in which pack_tiles lays out the D fragments of two adjacent 16-row tiles
__device__ __forceinline__ bf16x8 tr_frag_perm(const uint16_t *tile, int c, int col0, int lane) {
  const int g = lane >> 4;
  const u32x2 lo = tr4(tile, 32 * c + 4 * g, col0, lane), hi = tr4(tile, 32 * c + 16 + 4 * g, col0, lane);
  const u32x4 v = {lo[0], lo[1], hi[0], hi[1]};
  return as_frag(v);
}

// which (sequence, head, block) a workgroup owns: the blocks of one (sequence, head) -- which re-read the same K / V (or
// Q / dO) rows -- sit on ONE XCD (block id mod 8) and share its L2
struct Work {
  int b, h, blk;
  int L, Lq;                  // this sequence's keys / computed queries
  size_t row0, row0q;
};
__device__ __forceinline__ bool locate(const Params &P, Work &W) {
  const int id = blockIdx.x, units = P.B * P.H;
  int u, blk;
  if ((units & 7) == 0) {
    const int xcd = id & 7, slot = id >> 3;
    u = (slot / P.nblk) * 8 + xcd;
    blk = slot % P.nblk;
  } else {
    u = id / P.nblk;
    blk = id % P.nblk;
  }
  int b = u / P.H;
  W.h = u % P.H;
  if (P.seq_order) b = P.seq_order[b];
  W.b = b;
  W.blk = blk;
  W.L = P.L;
  W.Lq = P.Lq;
  W.row0 = (size_t)b * P.L;
  W.row0q = (size_t)b * P.Lq;
  if (P.cu_rows) {                                            // packed variable-length sequences
    W.row0 = W.row0q = (size_t)P.cu_rows[b];
    W.L = W.Lq = P.cu_rows[b + 1] - P.cu_rows[b];
    if (W.L <= 0) return false;
    if (P.q_limit) W.Lq = min(W.Lq, max(P.q_limit[b], 0));    // only the leading queries are wanted
  }
  return true;
}

// one 64-row block of two bf16 matrices (head h's 64 columns; rows >= rows_valid zero) -> two LDS tiles [64][KS]:
// issue() requests the four 16-byte pieces of a thread, commit() writes them
struct BlockPair {
  u32x4 va[2], vb[2];
  __device__ __forceinline__ void issue(const uint16_t *src_a, int ld_a, const uint16_t *src_b, int ld_b, int row_first, int rows_valid) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int e = threadIdx.x + i * kThreads, r = row_first + (e >> 3), ch = e & 7;
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
    for (int i = 0; i < 2; ++i) {
      const int e = threadIdx.x + i * kThreads, r = e >> 3, ch = e & 7;
      *reinterpret_cast<u32x4 *>(dst_a + r * KS + ch * 8) = va[i];
      *reinterpret_cast<u32x4 *>(dst_b + r * KS + ch * 8) = vb[i];
    }
  }
};

__device__ __forceinline__ u32x4 load_frag(const uint16_t *base, int row, int ld, int col) {
  return *reinterpret_cast<const u32x4 *>(reinterpret_cast<const char *>(base) + (unsigned int)(row * ld + col) * 2u);
}

constexpr int kTile = BLK * KS;           // elements of one LDS tile
constexpr size_t kLdsFwd = (size_t)4 * kTile * 2 + 2 * BLK * 4;              // K, V double-buffered + key terms
constexpr size_t kLdsDkv = (size_t)4 * kTile * 2 + 2 * 2 * BLK * 4;          // Q, dO double-buffered + lse2, delta

// ==========================================================================================
// forward: workgroup = 64 queries of one (sequence, head); key blocks streamed
// ==========================================================================================
__global__ __launch_bounds__(kThreads) void fwd_kernel(const Params P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint16_t *Kb = reinterpret_cast<uint16_t *>(smem);            // [2][64][KS]
  uint16_t *Vb = Kb + 2 * kTile;                                 // [2][64][KS]
  float *mbs = reinterpret_cast<float *>(Vb + 2 * kTile);        // [2][64] additive key term (base 2): 0 or -inf
  Work W;
  if (!locate(P, W)) return;
  if (W.blk * BLK >= W.Lq) return;                               // workgroup-uniform: no query of this block exists
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, m = lane & 15, g = lane >> 4;
  const int L = W.L, Lq = W.Lq, h = W.h;
  const uint16_t *qb = P.q + W.row0q * P.ld_q + h * DH;
  const uint16_t *kb = P.k + W.row0 * P.ld_kv + h * DH;
  const uint16_t *vb = P.v + W.row0 * P.ld_kv + h * DH;
  const int nkb = (L + BLK - 1) / BLK;
  const int qi = W.blk * BLK + 16 * wave + m;                    // this lane's query
  const bool active = W.blk * BLK + 16 * wave < Lq;              // wave-uniform: the strip has at least one query
  const int qc = min(qi, Lq - 1);

  BlockPair st;
  st.issue(kb, P.ld_kv, vb, P.ld_kv, 0, L);
  bf16x8 bq[2];
#pragma unroll
  for (int c = 0; c < 2; ++c) bq[c] = as_frag(load_frag(qb, qc, P.ld_q, 32 * c + 8 * g));
  st.commit(Kb, Vb);
  if (threadIdx.x < BLK) mbs[threadIdx.x] = (threadIdx.x < L && !(P.mask && P.mask[W.row0 + threadIdx.x])) ? 0.f : -INFINITY;
  __syncthreads();

  const bool dropout = P.drop_thr != 0u;
  const float keep_scale = dropout ? 1.f / (1.f - P.p_drop) : 1.f;
  const unsigned int seedmix = dropout ? seed_fold(P.seed + (P.seed_dev ? *P.seed_dev : 0ull)) : 0u;
  const unsigned int thr16 = P.drop_thr >> 16;
  const unsigned int rp = (((unsigned int)W.b * P.H + h) * P.Lq + qi) * (unsigned int)((P.L + 1) >> 1);

  float m_run = -INFINITY, l_run = 0.f;                          // running maximum (base-2 logits) and this lane's part of the normaliser
  f32x4 o[4];
#pragma unroll
  for (int n = 0; n < 4; ++n) o[n] = zero_acc();

  for (int kbi = 0; kbi < nkb; ++kbi) {
    const int cur = kbi & 1, nxt = cur ^ 1;
    const bool more = kbi + 1 < nkb;
    if (more) st.issue(kb, P.ld_kv, vb, P.ld_kv, (kbi + 1) * BLK, L);
    float mnext = 0.f;
    if (more && threadIdx.x < BLK) {
      const int t = (kbi + 1) * BLK + threadIdx.x;
      mnext = (t < L && !(P.mask && P.mask[W.row0 + t])) ? 0.f : -INFINITY;
    }
    if (active) {
      const uint16_t *Kc = Kb + cur * kTile, *Vc = Vb + cur * kTile;
      const float *mc = mbs + cur * BLK;
      f32x4 x[4];
      float bmax = -INFINITY;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f32x4 acc = zero_acc();
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const u32x4 a = *reinterpret_cast<const u32x4 *>(Kc + (16 * j + m) * KS + 32 * c + 8 * g);
          acc = mfma32(as_frag(a), bq[c], acc);
        }
        const f32x4 kt = *reinterpret_cast<const f32x4 *>(mc + 16 * j + 4 * g);
#pragma unroll
        for (int r = 0; r < 4; ++r) x[j][r] = fmaf(acc[r], kC, kt[r]);
        bmax = fmaxf(fmaxf(bmax, fmaxf(x[j][0], x[j][1])), fmaxf(x[j][2], x[j][3]));
      }
      bmax = xor_max_g(bmax);
      const float m_new = fmaxf(m_run, bmax);
      const float m_use = m_new == -INFINITY ? 0.f : m_new;      // every key so far masked: keep the arithmetic finite
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_use);
      m_run = m_new;
      float psum = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          x[j][r] = __builtin_amdgcn_exp2f(x[j][r] - m_use);
          psum += x[j][r];
        }
        if (dropout) {            // keys 16 j + 4 g + {0,1} and {2,3} of the block: two hashes for the four elements
          const int t0 = kbi * BLK + 16 * j + 4 * g;
          const unsigned int r01 = pair_rng(seedmix, rp, t0), r23 = pair_rng(seedmix, rp, t0 + 2);
          x[j][0] = (r01 & 0xFFFFu) >= thr16 ? x[j][0] : 0.f;
          x[j][1] = (r01 >> 16) >= thr16 ? x[j][1] : 0.f;
          x[j][2] = (r23 & 0xFFFFu) >= thr16 ? x[j][2] : 0.f;
          x[j][3] = (r23 >> 16) >= thr16 ? x[j][3] : 0.f;
        }
      }
      l_run = fmaf(l_run, alpha, psum);
#pragma unroll
      for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) o[n][r] *= alpha;
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const bf16x8 pb = pack_tiles(x[2 * c], x[2 * c + 1]);
#pragma unroll
        for (int n = 0; n < 4; ++n) o[n] = mfma32(tr_frag_perm(Vc, c, 16 * n, lane), pb, o[n]);
      }
    }
    if (more) {
      st.commit(Kb + nxt * kTile, Vb + nxt * kTile);
      if (threadIdx.x < BLK) mbs[nxt * BLK + threadIdx.x] = mnext;
    }
    __syncthreads();
  }
  if (!active) return;
  const float lsum = xor_sum_g(l_run);                           // all keys masked -> 0 -> NaN row, like torch
  if (g == 0 && qi < Lq) P.lse[((size_t)W.b * P.H + h) * P.Lq + qi] = (m_run + __builtin_amdgcn_logf(lsum)) * kLn2;
  const float sc = keep_scale / lsum;
  if (qi < Lq) {
    uint16_t *op = P.out + (W.row0q + qi) * P.ld_o + h * DH + 4 * g;
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      const u32x2 v = {pack2(o[n][0] * sc, o[n][1] * sc), pack2(o[n][2] * sc, o[n][3] * sc)};
      *reinterpret_cast<u32x2 *>(op + 16 * n) = v;
    }
  }
}

// ==========================================================================================
// backward, launch 1: workgroup = 64 queries -> dQ (and delta per query); key blocks streamed
// ==========================================================================================
__global__ __launch_bounds__(kThreads) void bwd_dq_kernel(const Params P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint16_t *Kb = reinterpret_cast<uint16_t *>(smem);
  uint16_t *Vb = Kb + 2 * kTile;
  float *mbs = reinterpret_cast<float *>(Vb + 2 * kTile);
  Work W;
  if (!locate(P, W)) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, m = lane & 15, g = lane >> 4;
  const int L = W.L, Lq = W.Lq, h = W.h;
  if (W.blk * BLK >= Lq) {
    // queries of the sequence that were not computed (q_limit): their dQ rows are zero
    if (P.q_limit && P.cu_rows && W.blk * BLK < L) {
      for (int e = threadIdx.x; e < BLK * 8; e += kThreads) {
        const int qr = W.blk * BLK + (e >> 3);
        if (qr < L) *reinterpret_cast<u32x4 *>(P.dq + (W.row0q + qr) * P.ld_dq + h * DH + 8 * (e & 7)) = zero4();
      }
    }
    return;
  }
  const uint16_t *qb = P.q + W.row0q * P.ld_q + h * DH;
  const uint16_t *kb = P.k + W.row0 * P.ld_kv + h * DH;
  const uint16_t *vb = P.v + W.row0 * P.ld_kv + h * DH;
  const uint16_t *dob = P.dout + W.row0q * P.ld_o + h * DH;
  const uint16_t *ob = P.out + W.row0q * P.ld_o + h * DH;
  const int nkb = (L + BLK - 1) / BLK;
  const int qi = W.blk * BLK + 16 * wave + m;
  const bool active = W.blk * BLK + 16 * wave < Lq;
  const int qc = min(qi, Lq - 1);

  BlockPair st;
  st.issue(kb, P.ld_kv, vb, P.ld_kv, 0, L);
  bf16x8 bq[2], bdo[2];
  u32x4 ov[2];
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    bq[c] = as_frag(load_frag(qb, qc, P.ld_q, 32 * c + 8 * g));
    bdo[c] = as_frag(qi < Lq ? load_frag(dob, qc, P.ld_o, 32 * c + 8 * g) : zero4());
    ov[c] = load_frag(ob, qc, P.ld_o, 32 * c + 8 * g);
  }
  const size_t stat = ((size_t)W.b * P.H + h) * P.Lq;            // row of lse / delta
  float lse2 = qi < Lq ? P.lse[stat + qc] : INFINITY;            // queries past Lq: p = 2^(x - inf) = 0
  st.commit(Kb, Vb);
  if (threadIdx.x < BLK) mbs[threadIdx.x] = (threadIdx.x < L && !(P.mask && P.mask[W.row0 + threadIdx.x])) ? 0.f : -INFINITY;
  // delta = sum_d dO[q][d] O[q][d] = rowsum(P' dP') for the dropped, rescaled probabilities of the forward pass
  float delta = 0.f;
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const u32x4 dv = __builtin_bit_cast(u32x4, bdo[c]);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      delta = fmaf(bf2f(dv[e] & 0xFFFFu), bf2f(ov[c][e] & 0xFFFFu), delta);
      delta = fmaf(__uint_as_float(dv[e] & 0xFFFF0000u), __uint_as_float(ov[c][e] & 0xFFFF0000u), delta);
    }
  }
  delta = xor_sum_g(delta);
  lse2 *= kLog2e;
  if (g == 0 && qi < Lq && P.delta) P.delta[stat + qi] = delta;
  __syncthreads();

  const bool dropout = P.drop_thr != 0u;
  const float keep_scale = dropout ? 1.f / (1.f - P.p_drop) : 1.f;
  const unsigned int seedmix = dropout ? seed_fold(P.seed + (P.seed_dev ? *P.seed_dev : 0ull)) : 0u;
  const unsigned int thr16 = P.drop_thr >> 16;
  const unsigned int rp = (((unsigned int)W.b * P.H + h) * P.Lq + qi) * (unsigned int)((P.L + 1) >> 1);

  f32x4 o[4];                                                    // dQ^T strip = K^T dS^T
#pragma unroll
  for (int n = 0; n < 4; ++n) o[n] = zero_acc();
  for (int kbi = 0; kbi < nkb; ++kbi) {
    const int cur = kbi & 1, nxt = cur ^ 1;
    const bool more = kbi + 1 < nkb;
    if (more) st.issue(kb, P.ld_kv, vb, P.ld_kv, (kbi + 1) * BLK, L);
    float mnext = 0.f;
    if (more && threadIdx.x < BLK) {
      const int t = (kbi + 1) * BLK + threadIdx.x;
      mnext = (t < L && !(P.mask && P.mask[W.row0 + t])) ? 0.f : -INFINITY;
    }
    if (active) {
      const uint16_t *Kc = Kb + cur * kTile, *Vc = Vb + cur * kTile;
      const float *mc = mbs + cur * BLK;
      f32x4 ds[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f32x4 acc = zero_acc(), dacc = zero_acc();
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const u32x4 a = *reinterpret_cast<const u32x4 *>(Kc + (16 * j + m) * KS + 32 * c + 8 * g);
          const u32x4 av = *reinterpret_cast<const u32x4 *>(Vc + (16 * j + m) * KS + 32 * c + 8 * g);
          acc = mfma32(as_frag(a), bq[c], acc);            // S^T
          dacc = mfma32(as_frag(av), bdo[c], dacc);        // (dO V^T)^T
        }
        const f32x4 kt = *reinterpret_cast<const f32x4 *>(mc + 16 * j + 4 * g);
        if (dropout) {
          const int t0 = kbi * BLK + 16 * j + 4 * g;
          const unsigned int r01 = pair_rng(seedmix, rp, t0), r23 = pair_rng(seedmix, rp, t0 + 2);
          dacc[0] = (r01 & 0xFFFFu) >= thr16 ? dacc[0] : 0.f;    // dropped gradient, before its 1 / (1 - p) scale
          dacc[1] = (r01 >> 16) >= thr16 ? dacc[1] : 0.f;
          dacc[2] = (r23 & 0xFFFFu) >= thr16 ? dacc[2] : 0.f;
          dacc[3] = (r23 >> 16) >= thr16 ? dacc[3] : 0.f;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float p = __builtin_amdgcn_exp2f(fmaf(acc[r], kC, kt[r]) - lse2);
          ds[j][r] = p * fmaf(dacc[r], keep_scale, -delta);      // the 1/8 of the logits goes onto dQ below
        }
      }
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const bf16x8 db = pack_tiles(ds[2 * c], ds[2 * c + 1]);
#pragma unroll
        for (int n = 0; n < 4; ++n) o[n] = mfma32(tr_frag_perm(Kc, c, 16 * n, lane), db, o[n]);
      }
    }
    if (more) {
      st.commit(Kb + nxt * kTile, Vb + nxt * kTile);
      if (threadIdx.x < BLK) mbs[nxt * BLK + threadIdx.x] = mnext;
    }
    __syncthreads();
  }
  if (active && qi < Lq) {
    uint16_t *op = P.dq + (W.row0q + qi) * P.ld_dq + h * DH + 4 * g;
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      const u32x2 v = {pack2(o[n][0] * 0.125f, o[n][1] * 0.125f), pack2(o[n][2] * 0.125f, o[n][3] * 0.125f)};
      *reinterpret_cast<u32x2 *>(op + 16 * n) = v;
    }
  }
  if (P.q_limit && P.cu_rows && Lq < L) {     // the uncomputed queries that share this block with computed ones
    for (int e = threadIdx.x; e < BLK * 8; e += kThreads) {
      const int qr = W.blk * BLK + (e >> 3);
      if (qr >= Lq && qr < L) *reinterpret_cast<u32x4 *>(P.dq + (W.row0q + qr) * P.ld_dq + h * DH + 8 * (e & 7)) = zero4();
    }
  }
}

// ==========================================================================================
// backward, launch 2: workgroup = 64 keys -> dK, dV; query blocks (Q, dO, lse, delta) streamed
// ==========================================================================================
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(3, 3))) void bwd_dkv_kernel(const Params P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint16_t *Qb = reinterpret_cast<uint16_t *>(smem);            // [2][64][KS]
  uint16_t *Ob = Qb + 2 * kTile;                                 // [2][64][KS]  dO rows
  float *ls = reinterpret_cast<float *>(Ob + 2 * kTile);         // [2][64] log2(e) * lse (+inf past Lq)
  float *dl = ls + 2 * BLK;                                      // [2][64] delta
  Work W;
  if (!locate(P, W)) return;
  const int L = W.L, Lq = W.Lq, h = W.h;
  if (W.blk * BLK >= L) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, m = lane & 15, g = lane >> 4;
  const uint16_t *qb = P.q + W.row0q * P.ld_q + h * DH;
  const uint16_t *kb = P.k + W.row0 * P.ld_kv + h * DH;
  const uint16_t *vb = P.v + W.row0 * P.ld_kv + h * DH;
  const uint16_t *dob = P.dout + W.row0q * P.ld_o + h * DH;
  const size_t stat = ((size_t)W.b * P.H + h) * P.Lq;
  const int nqb = (Lq + BLK - 1) / BLK;
  const int t = W.blk * BLK + 16 * wave + m;                     // this lane's key
  const bool active = W.blk * BLK + 16 * wave < L;
  const int tc = min(t, L - 1);

  BlockPair st;
  st.issue(qb, P.ld_q, dob, P.ld_o, 0, Lq);
  bf16x8 bk[2], bv[2];
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    bk[c] = as_frag(load_frag(kb, tc, P.ld_kv, 32 * c + 8 * g));
    bv[c] = as_frag(load_frag(vb, tc, P.ld_kv, 32 * c + 8 * g));
  }
  const float kt = (t < L && !(P.mask && P.mask[W.row0 + tc])) ? 0.f : -INFINITY;
  float l_next = INFINITY, d_next = 0.f;
  if (threadIdx.x < BLK && threadIdx.x < Lq) {
    l_next = P.lse[stat + threadIdx.x] * kLog2e;
    d_next = P.delta[stat + threadIdx.x];
  }
  st.commit(Qb, Ob);
  if (threadIdx.x < BLK) {
    ls[threadIdx.x] = l_next;
    dl[threadIdx.x] = d_next;
  }
  __syncthreads();

  const bool dropout = P.drop_thr != 0u;
  const float keep_scale = dropout ? 1.f / (1.f - P.p_drop) : 1.f;
  const unsigned int seedmix = dropout ? seed_fold(P.seed + (P.seed_dev ? *P.seed_dev : 0ull)) : 0u;
  const unsigned int thr16 = P.drop_thr >> 16;
  const unsigned int pitch2 = (unsigned int)((P.L + 1) >> 1);
  const unsigned int bh_base = ((unsigned int)W.b * P.H + h) * P.Lq;

  f32x4 dk[4], dv[4];
#pragma unroll
  for (int n = 0; n < 4; ++n) {
    dk[n] = zero_acc();
    dv[n] = zero_acc();
  }
  for (int qbi = 0; qbi < nqb; ++qbi) {
    const int cur = qbi & 1, nxt = cur ^ 1;
    const bool more = qbi + 1 < nqb;
    if (more) {
      st.issue(qb, P.ld_q, dob, P.ld_o, (qbi + 1) * BLK, Lq);
      const int qn = (qbi + 1) * BLK + threadIdx.x;
      l_next = INFINITY;
      d_next = 0.f;
      if (threadIdx.x < BLK && qn < Lq) {
        l_next = P.lse[stat + qn] * kLog2e;
        d_next = P.delta[stat + qn];
      }
    }
    if (active) {
      const uint16_t *Qc = Qb + cur * kTile, *Oc = Ob + cur * kTile;
      const float *lc = ls + cur * BLK, *dc = dl + cur * BLK;
      // 32 queries at a time: scores, dP, probabilities and dS of two query tiles, then their share of dV^T += dO^T P and
      // dK^T += Q^T dS (reduction over those 32 queries, in the order pack_tiles lays them out) -- two tiles of P / dS
      // live at once, not four (registers: three waves per SIMD)
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        f32x4 pt[2], ds[2];
#pragma unroll
        for (int ii = 0; ii < 2; ++ii) {
          const int i = 2 * c + ii;            // query tile of the block
          f32x4 sacc = zero_acc(), dacc = zero_acc();
#pragma unroll
          for (int cc = 0; cc < 2; ++cc) {
            const u32x4 vq = *reinterpret_cast<const u32x4 *>(Qc + (16 * i + m) * KS + 32 * cc + 8 * g);
            const u32x4 vo = *reinterpret_cast<const u32x4 *>(Oc + (16 * i + m) * KS + 32 * cc + 8 * g);
            sacc = mfma32(as_frag(vq), bk[cc], sacc);       // S[query 16 i + 4 g + r][key t]
            dacc = mfma32(as_frag(vo), bv[cc], dacc);       // dO V^T
          }
          const f32x4 lq = *reinterpret_cast<const f32x4 *>(lc + 16 * i + 4 * g);
          const f32x4 dq4 = *reinterpret_cast<const f32x4 *>(dc + 16 * i + 4 * g);
          bool keep[4] = {true, true, true, true};
          if (dropout) {
            // hashes of (query q0 + r, key pair t >> 1): this lane computes two of the four, the lane of the other key of
            // the pair (m ^ 1) the other two
            const int q0 = qbi * BLK + 16 * i + 4 * g, par = m & 1;
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
            const float p = __builtin_amdgcn_exp2f(fmaf(sacc[r], kC, kt) - lq[r]);
            const float dp = keep[r] ? dacc[r] : 0.f;
            pt[ii][r] = keep[r] ? p : 0.f;                                  // 1 / (1 - p) goes onto dV below
            ds[ii][r] = p * fmaf(dp, keep_scale, -dq4[r]);                  // 1/8 goes onto dK below
          }
        }
        const bf16x8 pb = pack_tiles(pt[0], pt[1]);
        const bf16x8 db = pack_tiles(ds[0], ds[1]);
#pragma unroll
        for (int n = 0; n < 4; ++n) {
          dv[n] = mfma32(tr_frag_perm(Oc, c, 16 * n, lane), pb, dv[n]);
          dk[n] = mfma32(tr_frag_perm(Qc, c, 16 * n, lane), db, dk[n]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (more) {
      st.commit(Qb + nxt * kTile, Ob + nxt * kTile);
      if (threadIdx.x < BLK) {
        ls[nxt * BLK + threadIdx.x] = l_next;
        dl[nxt * BLK + threadIdx.x] = d_next;
      }
    }
    __syncthreads();
  }
  if (active && t < L) {
    uint16_t *pk = P.dk + (W.row0 + t) * P.ld_dkv + h * DH + 4 * g;
    uint16_t *pv = P.dv + (W.row0 + t) * P.ld_dkv + h * DH + 4 * g;
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      const u32x2 vk = {pack2(dk[n][0] * 0.125f, dk[n][1] * 0.125f), pack2(dk[n][2] * 0.125f, dk[n][3] * 0.125f)};
      const u32x2 vv = {pack2(dv[n][0] * keep_scale, dv[n][1] * keep_scale), pack2(dv[n][2] * keep_scale, dv[n][3] * keep_scale)};
      *reinterpret_cast<u32x2 *>(pk + 16 * n) = vk;
      *reinterpret_cast<u32x2 *>(pv + 16 * n) = vv;
    }
  }
}

}  // namespace gps_attn_fa

namespace gps_attn {

// block-streaming plain form (no pairwise term); argument checks done by run_ex.  Backward needs `out` and the
// (B, H, Lq) fp32 workspace `delta_ws`.
int run_plain_blocks(const gps_attn_args *a, bool backward, hipStream_t s) {
  using namespace gps_attn_fa;
  if (a->sw || a->pl || a->pl_planes || a->dtype != GPS_ATTN_BF16) return GPS_ERR_UNSUPPORTED;
  if (backward && (!a->out || !a->delta_ws)) return GPS_ERR_UNSUPPORTED;
  Params P = {};
  P.B = a->B; P.H = a->H; P.L = a->Lk; P.Lq = a->Lq;
  P.ld_q = a->ld_q; P.ld_kv = a->ld_kv; P.ld_o = a->ld_o;
  P.q = (const uint16_t *)a->q; P.k = (const uint16_t *)a->k; P.v = (const uint16_t *)a->v;
  P.mask = a->mask; P.out = (uint16_t *)a->out; P.lse = a->lse;
  P.p_drop = a->p_drop; P.seed = a->seed; P.seed_dev = (const unsigned long long *)a->seed_dev;
  P.drop_thr = a->p_drop > 0.f ? (unsigned int)((double)a->p_drop * 4294967296.0) : 0u;
  P.cu_rows = a->cu_rows;
  P.seq_order = a->cu_rows ? a->seq_order : nullptr;
  P.q_limit = a->cu_rows ? a->q_limit : nullptr;
  const int nqb = (a->Lq + BLK - 1) / BLK, nkb = (a->Lk + BLK - 1) / BLK;
  if (!backward) {
    P.nblk = nqb;
    hipLaunchKernelGGL(fwd_kernel, dim3(P.B * P.H * nqb), dim3(kThreads), kLdsFwd, s, P);
    return hipGetLastError() == hipSuccess ? GPS_OK : GPS_ERR_LAUNCH;
  }
  P.dout = (const uint16_t *)a->dout; P.dq = (uint16_t *)a->dq; P.dk = (uint16_t *)a->dk; P.dv = (uint16_t *)a->dv;
  P.ld_dq = a->ld_dq; P.ld_dkv = a->ld_dkv; P.delta = a->delta_ws;
  P.nblk = nqb;
  hipLaunchKernelGGL(bwd_dq_kernel, dim3(P.B * P.H * nqb), dim3(kThreads), kLdsFwd, s, P);
  if (hipGetLastError() != hipSuccess) return GPS_ERR_LAUNCH;
  P.nblk = nkb;
  hipLaunchKernelGGL(bwd_dkv_kernel, dim3(P.B * P.H * nkb), dim3(kThreads), kLdsDkv, s, P);
  return hipGetLastError() == hipSuccess ? GPS_OK : GPS_ERR_LAUNCH;
}

}  // namespace gps_attn
