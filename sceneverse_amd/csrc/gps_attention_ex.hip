// gps_attention_ex.hip -- two more forms of the GPS attention core (gps_attention.hip has the bf16 kernels):
//
//   * fp32 operands on the fp32 MFMA (v_mfma_f32_16x16x4_f32): the "fp32 master path" of the reference's mathematics
//       modules/layers/transformers.py:193-239 (MultiHeadAttentionSpatial, fusion 'cond'), :141 (nn.MultiheadAttention)
//     for parity runs at fp32 tolerances; self- and cross-attention (Lq != Lk), plain and spatial form, forward and
//     backward, attention dropout with the same counter-based stream as the bf16 kernels.
//   * bf16 operands with Q K^T and P V on the OCP e4m3 MFMA (v_mfma_f32_16x16x32_fp8_fp8), forward: the attention
//     path of BASELINE configs[4] (256 objects + 256 tokens).
//
// Both keep the decomposition of the streaming bf16 kernels: one workgroup per (scene, head), one wave per 16-query
// strip (forward, backward pass 1) or 16-key strip (backward pass 2), K / V (then Q / dO) of the head resident in LDS,
// scores computed TRANSPOSED (S^T = K Q^T) so that a lane owns a query column and the D fragment of S^T is the A
// operand of the P V product without any data movement.
//
// fp32 MFMA operand map (v_mfma_f32_16x16x4_f32, D = A(16x4) B(4x16) + C):
//   A: lane l holds A[i = l % 16][k = l / 16];  B: lane l holds B[k = l / 16][j = l % 16];
//   D: lane l, register r holds D[i = 4 (l / 16) + r][j = l % 16]          (the D layout of the 16x16x32 bf16 form)
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gps_attention_ex.h"
#include "gps_hip.h"
#include "gps_device_flags.h"

namespace gps_attn {
namespace x {

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

constexpr int DH = 64;
constexpr int SD = 6;
constexpr int KSF = DH + 4;        // fp32 LDS row pitch (272 B, 16-byte aligned)
constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;

struct PX {
  int B, H, Lq, Lk, ntq, ntk;
  int ld_q, ld_kv, ld_o, ld_dq, ld_dkv;
  const void *q, *k, *v;
  const float *sw, *pl;
  const uint8_t *mask;
  void *out;
  float *lse;
  const void *dout;
  void *dq, *dk, *dv;
  float *dsw;
  float p_drop;
  unsigned int drop_thr;
  unsigned long long seed;
  const unsigned long long *seed_dev;
};

// ---- the dropout stream of gps_attention.hip's streaming kernels (same function of (seed, query, key pair)) ----
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
__device__ __forceinline__ bool pair_keep(unsigned int r, int t, unsigned int thr16) {
  return ((t & 1) ? (r >> 16) : (r & 0xFFFFu)) >= thr16;
}
__device__ __forceinline__ unsigned long long effective_seed(const PX &P) { return P.seed + (P.seed_dev ? *P.seed_dev : 0ull); }

__device__ __forceinline__ void block_to_bh(const PX &P, int &b, int &h) {
  const int id = blockIdx.x;
  if ((P.B & 7) == 0) {       // heads of one scene on one XCD: they share the pairwise tensor in its L2
    const int xcd = id & 7, slot = id >> 3;
    b = (slot / P.H) * 8 + xcd;
    h = slot % P.H;
  } else {
    b = id / P.H;
    h = id % P.H;
  }
}
__device__ __forceinline__ float xor_reduce_max_rows(float v) {
  v = fmaxf(v, __shfl_xor(v, 16, 64));
  return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float xor_reduce_sum_rows(float v) {
  v += __shfl_xor(v, 16, 64);
  return v + __shfl_xor(v, 32, 64);
}
// z = w0 + sum_d w_d pl_d; returns log(clamp(sigmoid(z), 1e-6)) (masked keys: log(1e-6)); sig through `sig`.
// Full-precision expf / logf: this is the fp32 parity path.
__device__ __forceinline__ float spatial_bias(const float *__restrict__ plp, const float (&w)[SD], bool key_masked, float &sig) {
  float z = w[0];
#pragma unroll
  for (int d = 0; d < 5; ++d) z = fmaf(w[1 + d], plp[d], z);
  sig = key_masked ? 0.f : 1.f / (1.f + expf(-z));
  return logf(fmaxf(sig, 1e-6f));
}
__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// rows [0, rows_total) x 64 fp32 columns of head h -> LDS [rows_total][KSF]; rows >= rows_valid are zero
__device__ __forceinline__ void stage_rows_f32(float *dst, const float *src, int ld, int rows_valid, int rows_total) {
  for (int e = threadIdx.x; e < rows_total * 16; e += blockDim.x) {
    const int r = e >> 4, ch = e & 15;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (r < rows_valid) v = *reinterpret_cast<const f32x4 *>(src + (size_t)r * ld + ch * 4);
    *reinterpret_cast<f32x4 *>(dst + r * KSF + ch * 4) = v;
  }
}

// =====================================================================================================
// fp32 forward
// =====================================================================================================
template <bool SPATIAL>
__global__ __launch_bounds__(1024) void attn_f32_fwd_kernel(const PX P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int Lk = P.Lk, ntk = P.ntk, rows = ntk * 16, Lq = P.Lq, ntq = P.ntq;
  float *Ks = reinterpret_cast<float *>(smem);       // [rows][KSF]
  float *Vs = Ks + rows * KSF;                        // [rows][KSF]
  float *mb = Vs + rows * KSF;                        // [rows] additive key term: 0, or -inf (padded / past Lk)
  int b, h;
  block_to_bh(P, b, h);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  const int m = lane & 15, g = lane >> 4;
  const size_t row0k = (size_t)b * Lk, row0q = (size_t)b * Lq;
  const float *qb = reinterpret_cast<const float *>(P.q) + row0q * P.ld_q + h * DH;
  const float *kb = reinterpret_cast<const float *>(P.k) + row0k * P.ld_kv + h * DH;
  const float *vb = reinterpret_cast<const float *>(P.v) + row0k * P.ld_kv + h * DH;
  float *ob = reinterpret_cast<float *>(P.out) + row0q * P.ld_o + h * DH;
  stage_rows_f32(Ks, kb, P.ld_kv, Lk, rows);
  stage_rows_f32(Vs, vb, P.ld_kv, Lk, rows);
  for (int t = threadIdx.x; t < rows; t += blockDim.x) mb[t] = (t < Lk && !(P.mask && P.mask[row0k + t])) ? 0.f : -INFINITY;
  __syncthreads();

  const bool dropout = P.drop_thr != 0u;
  const float keep_scale = dropout ? 1.f / (1.f - P.p_drop) : 1.f;
  const unsigned int seedmix = dropout ? seed_fold(effective_seed(P)) : 0u;
  const unsigned int thr16 = P.drop_thr >> 16;
  const unsigned int pitch2 = (unsigned int)((Lk + 1) >> 1);

  for (int s = wave; s < ntq; s += nwaves) {
    const int qi = 16 * s + m;
    const bool q_ok = qi < Lq;
    float bq[16];                           // B operand of S^T = K Q^T: q[qi][4 kk + g]
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) bq[kk] = q_ok ? qb[(size_t)qi * P.ld_q + 4 * kk + g] : 0.f;
    float w[SD];
#pragma unroll
    for (int d = 0; d < SD; ++d) w[d] = (SPATIAL && q_ok) ? P.sw[((row0q + qi) * P.H + h) * SD + d] : 0.f;
    // base-2 logits of (query qi, keys 16 j + 4 g + 0..3)
    auto logits2 = [&](int j, f32x4 &x) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      const float *kr = Ks + (16 * j + m) * KSF + g;
#pragma unroll
      for (int kk = 0; kk < 16; ++kk) acc = mfma4(kr[4 * kk], bq[kk], acc);
      const f32x4 kt = *reinterpret_cast<const f32x4 *>(mb + 16 * j + 4 * g);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = acc[r] * 0.125f;
        if (SPATIAL) {
          const int t = 16 * j + 4 * g + r;
          if (t < Lk && q_ok) {
            float sig;
            v += spatial_bias(P.pl + ((row0q + qi) * Lk + t) * 5, w, kt[r] < 0.f, sig);
          }
        }
        x[r] = fmaf(v, kLog2e, kt[r]);
      }
    };
    float mx = -INFINITY;
    for (int j = 0; j < ntk; ++j) {
      f32x4 x;
      logits2(j, x);
      mx = fmaxf(mx, fmaxf(fmaxf(x[0], x[1]), fmaxf(x[2], x[3])));
    }
    const float gmx = xor_reduce_max_rows(mx);
    const unsigned int rp = (((unsigned int)b * P.H + h) * Lq + qi) * pitch2;
    float lsum = 0.f;
    f32x4 o[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) o[n] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int j = 0; j < ntk; ++j) {
      f32x4 x, pt;
      logits2(j, x);
#pragma unroll
      for (int r = 0; r < 4; ++r) pt[r] = exp2f(x[r] - gmx);
      lsum += (pt[0] + pt[1]) + (pt[2] + pt[3]);
      if (dropout) {
        const int t0 = 16 * j + 4 * g;
        const unsigned int r01 = pair_rng(seedmix, rp, t0), r23 = pair_rng(seedmix, rp, t0 + 2);
        pt[0] = (r01 & 0xFFFFu) >= thr16 ? pt[0] : 0.f;
        pt[1] = (r01 >> 16) >= thr16 ? pt[1] : 0.f;
        pt[2] = (r23 & 0xFFFFu) >= thr16 ? pt[2] : 0.f;
        pt[3] = (r23 >> 16) >= thr16 ? pt[3] : 0.f;
      }
      // O[query][d] += P[query][key] V[key][d]: for every r the k-slot g of the A operand is key 16 j + 4 g + r
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float *vr = Vs + (16 * j + 4 * g + r) * KSF + m;
#pragma unroll
        for (int n = 0; n < 4; ++n) o[n] = mfma4(pt[r], vr[16 * n], o[n]);
      }
    }
    lsum = xor_reduce_sum_rows(lsum);
    if (g == 0 && q_ok) P.lse[((size_t)b * P.H + h) * Lq + qi] = (gmx + log2f(lsum)) * kLn2;
    const float scale_q = keep_scale / lsum;      // of query 16 s + m; the output rows of this lane are 4 g + r
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int qr = 16 * s + 4 * g + r;
      const float sc = __shfl(scale_q, 4 * g + r, 64);
      if (qr < Lq) {
        float *op = ob + (size_t)qr * P.ld_o + m;
#pragma unroll
        for (int n = 0; n < 4; ++n) op[16 * n] = o[n][r] * sc;
      }
    }
  }
}

// =====================================================================================================
// fp32 backward: pass 1 (query strips) -> dQ, d cond-vector; pass 2 (key strips) -> dK, dV.  delta = rowsum(dO * O).
// =====================================================================================================
template <bool SPATIAL>
__global__ __launch_bounds__(1024) void attn_f32_bwd_kernel(const PX P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int Lk = P.Lk, ntk = P.ntk, rows = ntk * 16, Lq = P.Lq, ntq = P.ntq, rows_q = ntq * 16;
  const int rows_max = rows > rows_q ? rows : rows_q;
  float *Ks = reinterpret_cast<float *>(smem);        // pass 1: Ks [rows][KSF] | Vs [rows][KSF]
  float *Vs = Ks + rows * KSF;
  float *Qs = Ks, *dOs = Qs + rows_q * KSF;            // pass 2 (same storage): Qs [rows_q][KSF] | dOs [rows_q][KSF]
  float *delta_s = Ks + 2 * rows_max * KSF;            // [rows_q]
  float *lse_s = delta_s + rows_q;                     // [rows_q] log2(e) * lse, +inf past Lq
  float *mb = lse_s + rows_q;                          // [rows]
  int b, h;
  block_to_bh(P, b, h);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  const int m = lane & 15, g = lane >> 4;
  const size_t row0k = (size_t)b * Lk, row0q = (size_t)b * Lq;
  const float *qb = reinterpret_cast<const float *>(P.q) + row0q * P.ld_q + h * DH;
  const float *kb = reinterpret_cast<const float *>(P.k) + row0k * P.ld_kv + h * DH;
  const float *vb = reinterpret_cast<const float *>(P.v) + row0k * P.ld_kv + h * DH;
  const float *dob = reinterpret_cast<const float *>(P.dout) + row0q * P.ld_o + h * DH;
  const float *ob = reinterpret_cast<const float *>(P.out) + row0q * P.ld_o + h * DH;
  float *dqb = reinterpret_cast<float *>(P.dq) + row0q * P.ld_dq + h * DH;
  float *dkb = reinterpret_cast<float *>(P.dk) + row0k * P.ld_dkv + h * DH;
  float *dvb = reinterpret_cast<float *>(P.dv) + row0k * P.ld_dkv + h * DH;
  const float *lse = P.lse + ((size_t)b * P.H + h) * Lq;
  const bool dropout = P.drop_thr != 0u;
  const float keep_scale = dropout ? 1.f / (1.f - P.p_drop) : 1.f;
  const unsigned int seedmix = dropout ? seed_fold(effective_seed(P)) : 0u;
  const unsigned int thr16 = P.drop_thr >> 16;
  const unsigned int pitch2 = (unsigned int)((Lk + 1) >> 1);
  const unsigned int bh_base = ((unsigned int)b * P.H + h) * Lq;

  stage_rows_f32(Ks, kb, P.ld_kv, Lk, rows);
  stage_rows_f32(Vs, vb, P.ld_kv, Lk, rows);
  for (int t = threadIdx.x; t < rows; t += blockDim.x) mb[t] = (t < Lk && !(P.mask && P.mask[row0k + t])) ? 0.f : -INFINITY;
  for (int t = threadIdx.x; t < rows_q; t += blockDim.x) {
    float d = 0.f;
    if (t < Lq) {
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        const f32x4 a = *reinterpret_cast<const f32x4 *>(dob + (size_t)t * P.ld_o + 4 * c);
        const f32x4 o = *reinterpret_cast<const f32x4 *>(ob + (size_t)t * P.ld_o + 4 * c);
        d = fmaf(a[0], o[0], d); d = fmaf(a[1], o[1], d); d = fmaf(a[2], o[2], d); d = fmaf(a[3], o[3], d);
      }
    }
    delta_s[t] = d;
    lse_s[t] = t < Lq ? lse[t] * kLog2e : INFINITY;
  }
  __syncthreads();

  // ---------------- pass 1: query strips ----------------
  for (int s = wave; s < ntq; s += nwaves) {
    const int qi = 16 * s + m;
    const bool q_ok = qi < Lq;
    float bq[16], bdo[16];
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      bq[kk] = q_ok ? qb[(size_t)qi * P.ld_q + 4 * kk + g] : 0.f;
      bdo[kk] = q_ok ? dob[(size_t)qi * P.ld_o + 4 * kk + g] : 0.f;
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
    for (int j = 0; j < ntk; ++j) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f}, dacc = {0.f, 0.f, 0.f, 0.f};
      const float *kr = Ks + (16 * j + m) * KSF + g, *vr = Vs + (16 * j + m) * KSF + g;
#pragma unroll
      for (int kk = 0; kk < 16; ++kk) {
        acc = mfma4(kr[4 * kk], bq[kk], acc);         // S^T
        dacc = mfma4(vr[4 * kk], bdo[kk], dacc);      // (dO V^T)^T
      }
      const int t0 = 16 * j + 4 * g;
      const f32x4 kt = *reinterpret_cast<const f32x4 *>(mb + t0);
      f32x4 ds;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int t = t0 + r;
        float x = acc[r] * 0.125f, gt = 0.f;
        if (SPATIAL && t < Lk && q_ok) {
          float sig;
          x += spatial_bias(P.pl + ((row0q + qi) * Lk + t) * 5, w, kt[r] < 0.f, sig);
          gt = sig > 1e-6f ? 1.f - sig : 0.f;
        }
        const float p = exp2f(fmaf(x, kLog2e, kt[r]) - lse_q);       // lse_q = +inf past Lq -> 0
        const bool keep = !dropout || pair_keep(pair_rng(seedmix, rp, t), t, thr16);
        const float dp = keep ? dacc[r] : 0.f;
        const float dlogit = p * fmaf(dp, keep_scale, -delta);
        if (SPATIAL) {
          const float dz = dlogit * gt;
          if (dz != 0.f) {
            const float *plp = P.pl + ((row0q + qi) * Lk + t) * 5;
            dw[0] += dz;
#pragma unroll
            for (int d = 0; d < 5; ++d) dw[1 + d] = fmaf(dz, plp[d], dw[1 + d]);
          }
        }
        ds[r] = dlogit;
      }
      // dQ[query][d] += dS[query][key] K[key][d]
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float *kc = Ks + (t0 + r) * KSF + m;
#pragma unroll
        for (int n = 0; n < 4; ++n) o[n] = mfma4(ds[r], kc[16 * n], o[n]);
      }
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
        float *op = dqb + (size_t)qr * P.ld_dq + m;
#pragma unroll
        for (int n = 0; n < 4; ++n) op[16 * n] = o[n][r] * 0.125f;
      }
    }
  }
  __syncthreads();
  stage_rows_f32(Qs, qb, P.ld_q, Lq, rows_q);
  stage_rows_f32(dOs, dob, P.ld_o, Lq, rows_q);
  __syncthreads();

  // ---------------- pass 2: key strips (a lane owns key t) ----------------
  for (int js = wave; js < ntk; js += nwaves) {
    const int t = 16 * js + m;
    const bool t_ok = t < Lk;
    const float kt = mb[t < rows ? t : 0];
    float bk[16], bv[16];
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      bk[kk] = t_ok ? kb[(size_t)t * P.ld_kv + 4 * kk + g] : 0.f;
      bv[kk] = t_ok ? vb[(size_t)t * P.ld_kv + 4 * kk + g] : 0.f;
    }
    f32x4 dk[4], dv[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      dk[n] = f32x4{0.f, 0.f, 0.f, 0.f};
      dv[n] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    for (int i = 0; i < ntq; ++i) {         // query tiles
      f32x4 sacc = {0.f, 0.f, 0.f, 0.f}, dacc = {0.f, 0.f, 0.f, 0.f};
      const float *qr_ = Qs + (16 * i + m) * KSF + g, *dr_ = dOs + (16 * i + m) * KSF + g;
#pragma unroll
      for (int kk = 0; kk < 16; ++kk) {
        sacc = mfma4(qr_[4 * kk], bk[kk], sacc);      // S[query 16 i + 4 g + r][key t]
        dacc = mfma4(dr_[4 * kk], bv[kk], dacc);      // dO V^T
      }
      const int q0 = 16 * i + 4 * g;
      const f32x4 lq = *reinterpret_cast<const f32x4 *>(lse_s + q0);
      const f32x4 dq4 = *reinterpret_cast<const f32x4 *>(delta_s + q0);
      f32x4 pt, ds;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int qi = q0 + r;
        float x = sacc[r] * 0.125f;
        if (SPATIAL && t_ok && qi < Lq) {
          float w[SD];
#pragma unroll
          for (int d = 0; d < SD; ++d) w[d] = P.sw[((row0q + qi) * P.H + h) * SD + d];
          float sig;
          x += spatial_bias(P.pl + ((row0q + qi) * Lk + t) * 5, w, kt < 0.f, sig);
        }
        const float p = exp2f(fmaf(x, kLog2e, kt) - lq[r]);
        const bool keep = !dropout || pair_keep(pair_rng(seedmix, (bh_base + (unsigned int)qi) * pitch2, t), t, thr16);
        const float dp = keep ? dacc[r] : 0.f;
        pt[r] = keep ? p : 0.f;
        ds[r] = p * fmaf(dp, keep_scale, -dq4[r]);
      }
      // dV[key][d] += P^T[key][query] dO[query][d];  dK[key][d] += dS^T[key][query] Q[query][d]
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float *dc = dOs + (q0 + r) * KSF + m, *qc = Qs + (q0 + r) * KSF + m;
#pragma unroll
        for (int n = 0; n < 4; ++n) {
          dv[n] = mfma4(pt[r], dc[16 * n], dv[n]);
          dk[n] = mfma4(ds[r], qc[16 * n], dk[n]);
        }
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int tr = 16 * js + 4 * g + r;
      if (tr < Lk) {
        float *pk = dkb + (size_t)tr * P.ld_dkv + m;
        float *pv = dvb + (size_t)tr * P.ld_dkv + m;
#pragma unroll
        for (int n = 0; n < 4; ++n) {
          pk[16 * n] = dk[n][r] * 0.125f;
          pv[16 * n] = dv[n][r] * keep_scale;
        }
      }
    }
  }
}

inline int pick_waves(int nt) {               // up to 16 waves per workgroup
  const int rounds = (nt + 15) / 16;
  return (nt + rounds - 1) / rounds;
}

// =====================================================================================================
// fp8 forward: bf16 operands, Q K^T and P V on v_mfma_f32_16x16x32_fp8_fp8 (OCP e4m3 on gfx950)
//   K8 = e4m3(K / s_K), V8 = e4m3(V / s_V)  with ONE scale per (scene, head) tile (amax / 224);
//   Q8 = e4m3(Q / s_q[query]) per query row;  logits = (Q8 . K8) s_q s_K / 8;  P8 = e4m3(128 p), p = 2^(x - max) <= 1;
//   O = (P8 . V8) s_V / 128 / sum(p).  Softmax statistics, the spatial term and all accumulation stay fp32.
// LDS: K8 row-major [rows][KS8] (A fragments: 8-byte reads), V8 TRANSPOSED [64][TS8] (B fragments of P V: the 8
// k-slots of lane group g are keys {32 c + 4 g + 0..3, 32 c + 16 + 4 g + 0..3}, two 4-byte reads), key term fp32.
// =====================================================================================================
constexpr int KS8 = DH + 8;                   // bytes per K8 row (8-byte aligned)
// amax maps to 224: representable in OCP e4m3 (max 448) with headroom for the rounding of the scale itself (and also
// inside the range of the e4m3 FNUZ encoding, max 240, should a part interpret the conversion that way)
constexpr float kFp8Max = 224.f;
constexpr float kPScale = 128.f;              // probabilities p <= 1 are quantised as e4m3(128 p)

__device__ __forceinline__ unsigned int cvt4_fp8(float a, float b, float c, float d) {   // 4 x e4m3 in one dword
  int v = 0;
  v = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, v, false);
  v = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, v, true);
  return (unsigned int)v;
}
__device__ __forceinline__ float bf_lo(unsigned int u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf_hi(unsigned int u) { return __uint_as_float(u & 0xFFFF0000u); }
__device__ __forceinline__ f32x4 mfma_fp8(unsigned long long a, unsigned long long b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8((long)a, (long)b, c, 0, 0, 0);
}
__device__ __forceinline__ float amax8(const u32x4 v) {
  float a = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) a = fmaxf(a, fmaxf(fabsf(bf_lo(v[i])), fabsf(bf_hi(v[i]))));
  return a;
}

template <bool SPATIAL>
__global__ __launch_bounds__(1024) void attn_fp8_fwd_kernel(const PX P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int Lk = P.Lk, ntk = P.ntk, nc = (ntk + 1) / 2, rows = nc * 32, Lq = P.Lq, ntq = P.ntq;
  const int TS8 = rows + 8;                                  // bytes per V8^T row
  unsigned char *K8 = smem;                                  // [rows][KS8]
  unsigned char *V8t = K8 + rows * KS8;                      // [64][TS8]
  float *mb = reinterpret_cast<float *>(V8t + 64 * TS8);     // [rows]
  float *red = mb + rows;                                    // [2][16] per-wave amax of K, V; then [0], [16] hold the tile scales
  int b, h;
  block_to_bh(P, b, h);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  const int m = lane & 15, g = lane >> 4;
  const size_t row0k = (size_t)b * Lk, row0q = (size_t)b * Lq;
  const uint16_t *qb = reinterpret_cast<const uint16_t *>(P.q) + row0q * P.ld_q + h * DH;
  const uint16_t *kb = reinterpret_cast<const uint16_t *>(P.k) + row0k * P.ld_kv + h * DH;
  const uint16_t *vb = reinterpret_cast<const uint16_t *>(P.v) + row0k * P.ld_kv + h * DH;
  uint16_t *ob = reinterpret_cast<uint16_t *>(P.out) + row0q * P.ld_o + h * DH;

  // ---- tile amax of K and V (first read of the rows; the second read below hits L2) ----
  float ak = 0.f, av = 0.f;
  for (int e = threadIdx.x; e < Lk * 8; e += blockDim.x) {
    const int r = e >> 3, ch = e & 7;
    ak = fmaxf(ak, amax8(*reinterpret_cast<const u32x4 *>(kb + (size_t)r * P.ld_kv + ch * 8)));
    av = fmaxf(av, amax8(*reinterpret_cast<const u32x4 *>(vb + (size_t)r * P.ld_kv + ch * 8)));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    ak = fmaxf(ak, __shfl_xor(ak, o, 64));
    av = fmaxf(av, __shfl_xor(av, o, 64));
  }
  if (lane == 0) {
    red[wave] = ak;
    red[16 + wave] = av;
  }
  __syncthreads();
  ak = 0.f;
  av = 0.f;
  for (int w2 = 0; w2 < nwaves; ++w2) {
    ak = fmaxf(ak, red[w2]);
    av = fmaxf(av, red[16 + w2]);
  }
  const float sK = ak > 0.f ? ak / kFp8Max : 1.f, sV = av > 0.f ? av / kFp8Max : 1.f;
  const float rK = 1.f / sK, rV = 1.f / sV;
  // ---- quantise into LDS ----
  for (int e = threadIdx.x; e < rows * 8; e += blockDim.x) {
    const int r = e >> 3, ch = e & 7;
    u32x4 kv = {0u, 0u, 0u, 0u}, vv = {0u, 0u, 0u, 0u};
    if (r < Lk) {
      kv = *reinterpret_cast<const u32x4 *>(kb + (size_t)r * P.ld_kv + ch * 8);
      vv = *reinterpret_cast<const u32x4 *>(vb + (size_t)r * P.ld_kv + ch * 8);
    }
    u32x2 k8;
    k8[0] = cvt4_fp8(bf_lo(kv[0]) * rK, bf_hi(kv[0]) * rK, bf_lo(kv[1]) * rK, bf_hi(kv[1]) * rK);
    k8[1] = cvt4_fp8(bf_lo(kv[2]) * rK, bf_hi(kv[2]) * rK, bf_lo(kv[3]) * rK, bf_hi(kv[3]) * rK);
    *reinterpret_cast<u32x2 *>(K8 + r * KS8 + ch * 8) = k8;
    const unsigned int v0 = cvt4_fp8(bf_lo(vv[0]) * rV, bf_hi(vv[0]) * rV, bf_lo(vv[1]) * rV, bf_hi(vv[1]) * rV);
    const unsigned int v1 = cvt4_fp8(bf_lo(vv[2]) * rV, bf_hi(vv[2]) * rV, bf_lo(vv[3]) * rV, bf_hi(vv[3]) * rV);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      V8t[(ch * 8 + i) * TS8 + r] = (unsigned char)(v0 >> (8 * i));
      V8t[(ch * 8 + 4 + i) * TS8 + r] = (unsigned char)(v1 >> (8 * i));
    }
  }
  for (int t = threadIdx.x; t < rows; t += blockDim.x) mb[t] = (t < Lk && !(P.mask && P.mask[row0k + t])) ? 0.f : -INFINITY;
  __syncthreads();

  const bool dropout = P.drop_thr != 0u;
  const float keep_scale = dropout ? 1.f / (1.f - P.p_drop) : 1.f;
  const unsigned int seedmix = dropout ? seed_fold(effective_seed(P)) : 0u;
  const unsigned int thr16 = P.drop_thr >> 16;
  const unsigned int pitch2 = (unsigned int)((Lk + 1) >> 1);

  for (int s = wave; s < ntq; s += nwaves) {
    const int qi = 16 * s + m;
    const bool q_ok = qi < Lq;
    // B operand of S^T = K Q^T: q[qi][32 c + 8 g .. + 7], quantised with the row's own scale
    u32x4 qv[2];
    float aq = 0.f;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      qv[c] = u32x4{0u, 0u, 0u, 0u};
      if (q_ok) qv[c] = *reinterpret_cast<const u32x4 *>(qb + (size_t)qi * P.ld_q + 32 * c + 8 * g);
      aq = fmaxf(aq, amax8(qv[c]));
    }
    aq = xor_reduce_max_rows(aq);                   // the 64 features of a query row sit in lanes m + 16 g'
    const float sq = aq > 0.f ? aq / kFp8Max : 1.f, rq = 1.f / sq;
    unsigned long long bq[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const unsigned int lo = cvt4_fp8(bf_lo(qv[c][0]) * rq, bf_hi(qv[c][0]) * rq, bf_lo(qv[c][1]) * rq, bf_hi(qv[c][1]) * rq);
      const unsigned int hi = cvt4_fp8(bf_lo(qv[c][2]) * rq, bf_hi(qv[c][2]) * rq, bf_lo(qv[c][3]) * rq, bf_hi(qv[c][3]) * rq);
      bq[c] = ((unsigned long long)hi << 32) | lo;
    }
    const float qk_scale = sq * sK * 0.125f;
    float w[SD];
#pragma unroll
    for (int d = 0; d < SD; ++d) w[d] = (SPATIAL && q_ok) ? P.sw[((row0q + qi) * P.H + h) * SD + d] : 0.f;
    auto logits2 = [&](int j, f32x4 &x) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const unsigned long long a = *reinterpret_cast<const unsigned long long *>(K8 + (16 * j + m) * KS8 + 32 * c + 8 * g);
        acc = mfma_fp8(a, bq[c], acc);
      }
      const f32x4 kt = *reinterpret_cast<const f32x4 *>(mb + 16 * j + 4 * g);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = acc[r] * qk_scale;
        if (SPATIAL) {
          const int t = 16 * j + 4 * g + r;
          if (t < Lk && q_ok) {
            float z = w[0];
            const float *plp = P.pl + ((row0q + qi) * Lk + t) * 5;
#pragma unroll
            for (int d = 0; d < 5; ++d) z = fmaf(w[1 + d], plp[d], z);
            const float sig = kt[r] < 0.f ? 0.f : __builtin_amdgcn_rcpf(1.f + __expf(-z));
            v += __logf(fmaxf(sig, 1e-6f));
          }
        }
        x[r] = fmaf(v, kLog2e, kt[r]);
      }
    };
    float mx = -INFINITY;
    for (int j = 0; j < ntk; ++j) {
      f32x4 x;
      logits2(j, x);
      mx = fmaxf(mx, fmaxf(fmaxf(x[0], x[1]), fmaxf(x[2], x[3])));
    }
    const float gmx = xor_reduce_max_rows(mx);
    const unsigned int rp = (((unsigned int)b * P.H + h) * Lq + qi) * pitch2;
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
        if (j < ntk) {
          f32x4 x;
          logits2(j, x);
#pragma unroll
          for (int r = 0; r < 4; ++r) pt[hh][r] = __builtin_amdgcn_exp2f(x[r] - gmx);
          lsum += (pt[hh][0] + pt[hh][1]) + (pt[hh][2] + pt[hh][3]);
          if (dropout) {
            const int t0 = 16 * j + 4 * g;
            const unsigned int r01 = pair_rng(seedmix, rp, t0), r23 = pair_rng(seedmix, rp, t0 + 2);
            pt[hh][0] = (r01 & 0xFFFFu) >= thr16 ? pt[hh][0] : 0.f;
            pt[hh][1] = (r01 >> 16) >= thr16 ? pt[hh][1] : 0.f;
            pt[hh][2] = (r23 & 0xFFFFu) >= thr16 ? pt[hh][2] : 0.f;
            pt[hh][3] = (r23 >> 16) >= thr16 ? pt[hh][3] : 0.f;
          }
        }
      }
      // A operand of P V: k-slots 8 g + 0..3 = keys 32 c + 4 g + r (tile 2 c), 8 g + 4..7 = keys 32 c + 16 + 4 g + r
      const unsigned int plo = cvt4_fp8(pt[0][0] * kPScale, pt[0][1] * kPScale, pt[0][2] * kPScale, pt[0][3] * kPScale);
      const unsigned int phi = cvt4_fp8(pt[1][0] * kPScale, pt[1][1] * kPScale, pt[1][2] * kPScale, pt[1][3] * kPScale);
      const unsigned long long pa = ((unsigned long long)phi << 32) | plo;
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        const unsigned char *vp = V8t + (16 * n + m) * TS8 + 32 * c + 4 * g;
        const unsigned int vlo = *reinterpret_cast<const unsigned int *>(vp);
        const unsigned int vhi = *reinterpret_cast<const unsigned int *>(vp + 16);
        o[n] = mfma_fp8(pa, ((unsigned long long)vhi << 32) | vlo, o[n]);
      }
    }
    lsum = xor_reduce_sum_rows(lsum);
    if (g == 0 && q_ok) P.lse[((size_t)b * P.H + h) * Lq + qi] = (gmx + __builtin_amdgcn_logf(lsum)) * kLn2;
    const float scale_q = keep_scale * sV * (1.f / kPScale) / lsum;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int qr = 16 * s + 4 * g + r;
      const float sc = __shfl(scale_q, 4 * g + r, 64);
      if (qr < Lq) {
        uint16_t *op = ob + (size_t)qr * P.ld_o + m;
#pragma unroll
        for (int n = 0; n < 4; ++n) op[16 * n] = __builtin_bit_cast(uint16_t, (__bf16)(o[n][r] * sc));
      }
    }
  }
}

PX make_px(const gps_attn_args *a) {
  PX P = {};
  P.B = a->B; P.H = a->H; P.Lq = a->Lq; P.Lk = a->Lk;
  P.ntq = (a->Lq + 15) / 16; P.ntk = (a->Lk + 15) / 16;
  P.ld_q = a->ld_q; P.ld_kv = a->ld_kv; P.ld_o = a->ld_o; P.ld_dq = a->ld_dq; P.ld_dkv = a->ld_dkv;
  P.q = a->q; P.k = a->k; P.v = a->v; P.sw = a->sw; P.pl = a->pl; P.mask = a->mask;
  P.out = a->out; P.lse = a->lse; P.dout = a->dout; P.dq = a->dq; P.dk = a->dk; P.dv = a->dv; P.dsw = a->dsw;
  P.p_drop = a->p_drop; P.seed = a->seed; P.seed_dev = (const unsigned long long *)a->seed_dev;
  P.drop_thr = a->p_drop > 0.f ? (unsigned int)((double)a->p_drop * 4294967296.0) : 0u;
  return P;
}

template <typename K>
int set_lds(K kernel, size_t lds, size_t &granted) {
  if (lds > 64 * 1024 && lds > granted) {
    if (hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return GPS_ERR_LAUNCH;
    granted = 160 * 1024;
  }
  return GPS_OK;
}

}  // namespace x

int run_f32(const gps_attn_args *a, bool backward, hipStream_t s) {
  using namespace x;
  if ((a->ld_q & 3) || (a->ld_kv & 3) || (a->ld_o & 3) || (backward && ((a->ld_dq & 3) || (a->ld_dkv & 3)))) return GPS_ERR_UNSUPPORTED;
  const PX P = make_px(a);
  if (P.ntq > 16 || P.ntk > 16) return GPS_ERR_UNSUPPORTED;       // Lq, Lk <= 256 (K and V of a head in LDS as fp32)
  const size_t rows = (size_t)P.ntk * 16, rows_q = (size_t)P.ntq * 16, rows_max = rows > rows_q ? rows : rows_q;
  const size_t lds = backward ? 4 * (2 * rows_max * KSF + 2 * rows_q + rows) : 4 * (2 * rows * KSF + rows);
  if (lds > 160 * 1024) return GPS_ERR_UNSUPPORTED;
  const int nw = pick_waves(backward ? (P.ntk > P.ntq ? P.ntk : P.ntq) : P.ntq);
  const dim3 grid(P.B * P.H), block(64 * nw);
  const bool spatial = P.sw != nullptr;
  static gps_dev::PerDevice<size_t, 4> granted_dev;
  size_t *granted = granted_dev.row();
  int st;
  if (backward) {
    if (spatial) {
      if ((st = set_lds(&attn_f32_bwd_kernel<true>, lds, granted[3])) != GPS_OK) return st;
      hipLaunchKernelGGL((attn_f32_bwd_kernel<true>), grid, block, lds, s, P);
    } else {
      if ((st = set_lds(&attn_f32_bwd_kernel<false>, lds, granted[2])) != GPS_OK) return st;
      hipLaunchKernelGGL((attn_f32_bwd_kernel<false>), grid, block, lds, s, P);
    }
  } else {
    if (spatial) {
      if ((st = set_lds(&attn_f32_fwd_kernel<true>, lds, granted[1])) != GPS_OK) return st;
      hipLaunchKernelGGL((attn_f32_fwd_kernel<true>), grid, block, lds, s, P);
    } else {
      if ((st = set_lds(&attn_f32_fwd_kernel<false>, lds, granted[0])) != GPS_OK) return st;
      hipLaunchKernelGGL((attn_f32_fwd_kernel<false>), grid, block, lds, s, P);
    }
  }
  return hipGetLastError() == hipSuccess ? GPS_OK : GPS_ERR_LAUNCH;
}

int run_fp8_forward(const gps_attn_args *a, hipStream_t s) {
  using namespace x;
  const PX P = make_px(a);
  if (P.ntq > 32 || P.ntk > 32) return GPS_ERR_UNSUPPORTED;       // Lq, Lk <= 512
  const size_t rows = (size_t)((P.ntk + 1) / 2) * 32;
  const size_t lds = rows * KS8 + 64 * (rows + 8) + 4 * rows + 4 * 32;
  const int nw = pick_waves(P.ntq);
  const dim3 grid(P.B * P.H), block(64 * nw);
  static gps_dev::PerDevice<size_t, 2> granted_dev;
  size_t *granted = granted_dev.row();
  int st;
  if (P.sw != nullptr) {
    if ((st = set_lds(&attn_fp8_fwd_kernel<true>, lds, granted[1])) != GPS_OK) return st;
    hipLaunchKernelGGL((attn_fp8_fwd_kernel<true>), grid, block, lds, s, P);
  } else {
    if ((st = set_lds(&attn_fp8_fwd_kernel<false>, lds, granted[0])) != GPS_OK) return st;
    hipLaunchKernelGGL((attn_fp8_fwd_kernel<false>), grid, block, lds, s, P);
  }
  return hipGetLastError() == hipSuccess ? GPS_OK : GPS_ERR_LAUNCH;
}

}  // namespace gps_attn
