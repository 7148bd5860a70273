// gps_layernorm.hip -- fused  y = LayerNorm(x + dropout(h)) * gamma + beta  (forward and backward)
// for the post-norm transformer layers of the GPS path on MI355X (gfx950).
//
// Reference pattern (every encoder layer, twice): modules/layers/transformers.py:143-153, :311-315
//     tgt = self.norm1(tgt + self.dropout1(tgt2));  tgt = self.norm2(tgt + self.dropout2(ffn(tgt)))
// which under bf16 autocast runs as dropout (bf16) -> add (promotes to fp32) -> layer_norm (fp32)
// -> a separate fp32->bf16 copy for the next GEMM, and three kernels plus casts in backward.
// Here: one launch forward (optionally also emitting the bf16 copy the next GEMM wants), one launch
// backward (+ one tiny column reduction for dgamma/dbeta).  HBM-bound: forward moves
// sizeof(x) + sizeof(h) + sizeof(y) [+2] bytes per element, backward sizeof(dy) + sizeof(x) +
// sizeof(h) + sizeof(dx) + sizeof(dh).
//
// One wave per row, the row (<= 2048 elements, 4 consecutive elements per lane and step, 16-byte
// loads) stays in VGPRs: mean and variance are computed two-pass from registers, row reductions are
// wave shuffles, no LDS in forward.  Dropout uses the same counter-based RNG as the attention core
// (seed + optional device seed word), recomputed in backward.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gps_hip.h"

namespace gps_ln {

constexpr int kBlock = 256;
constexpr int kWaves = kBlock / 64;
constexpr int kMaxIter = 8;          // D <= 8 * 256 (ITERS = D / 256 is a template parameter)

__device__ __forceinline__ unsigned int mix32(unsigned int x) {      // 32-bit avalanche hash ("lowbias32" constants)
  x ^= x >> 16;
  x *= 0x21F0AAADu;
  x ^= x >> 15;
  x *= 0x735A2D97u;
  x ^= x >> 15;
  return x;
}
// counter-based dropout stream: forward and backward draw the same bits for the same (seed, element index); the seed
// part is wave-uniform (scalar unit), the element part costs 2 multiplies and 3 xor-shifts (the 64-bit splitmix of
// the first version: ~30 vector instructions per element)
__device__ __forceinline__ unsigned int rng_u32(unsigned long long seed, unsigned long long idx) {
  const unsigned int s = mix32((unsigned int)seed ^ mix32((unsigned int)(seed >> 32) + 0x9E3779B9u));
  return mix32(((unsigned int)idx + (unsigned int)(idx >> 32) * 0x85EBCA6Bu) ^ s);
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ uint16_t f2bf(float f) {      // round to nearest even (v_cvt_pk_bf16_f32)
  return __builtin_bit_cast(uint16_t, (__bf16)f);
}
// 4 consecutive elements at p (element index e0), as fp32
__device__ __forceinline__ float4 load4(const float *p, size_t e0) {
  return *reinterpret_cast<const float4 *>(p + e0);
}
__device__ __forceinline__ float4 load4(const uint16_t *p, size_t e0) {
  const uint2 v = *reinterpret_cast<const uint2 *>(p + e0);
  return make_float4(__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xFFFF0000u),
                     __uint_as_float(v.y << 16), __uint_as_float(v.y & 0xFFFF0000u));
}
// the same 4 elements as they lie in memory (conversion deferred: every load of a row is issued before the first use)
__device__ __forceinline__ float4 load_raw(const float *p, size_t e0) { return *reinterpret_cast<const float4 *>(p + e0); }
__device__ __forceinline__ uint2 load_raw(const uint16_t *p, size_t e0) { return *reinterpret_cast<const uint2 *>(p + e0); }
__device__ __forceinline__ float4 to_f4(float4 v) { return v; }
__device__ __forceinline__ float4 to_f4(uint2 v) {
  return make_float4(__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xFFFF0000u),
                     __uint_as_float(v.y << 16), __uint_as_float(v.y & 0xFFFF0000u));
}
template <typename T> struct Raw4 { using type = float4; };
template <> struct Raw4<uint16_t> { using type = uint2; };
// loads above this line are issued before anything below it (the compiler otherwise sinks them next to their first use,
// i.e. behind the wave-uniform branches of the optional operands: one memory round trip per branch instead of one per row)
#define GPS_LN_LOADS_ISSUED() asm volatile("" ::: "memory")
__device__ __forceinline__ void store4(float *p, size_t e0, float4 v) {
  *reinterpret_cast<float4 *>(p + e0) = v;
}
__device__ __forceinline__ void store4(uint16_t *p, size_t e0, float4 v) {
  uint2 o;
  o.x = (unsigned int)f2bf(v.x) | ((unsigned int)f2bf(v.y) << 16);
  o.y = (unsigned int)f2bf(v.z) | ((unsigned int)f2bf(v.w) << 16);
  *reinterpret_cast<uint2 *>(p + e0) = o;
}

struct Drop {
  unsigned int thr;          // keep iff rng >= thr; 0 = no dropout
  float scale;               // 1 / (1 - p)
  unsigned long long seed;
};
__device__ __forceinline__ float4 drop4(float4 h, const Drop &d, unsigned long long e0) {
  if (d.thr == 0u) return h;
  h.x = rng_u32(d.seed, e0 + 0) >= d.thr ? h.x * d.scale : 0.f;
  h.y = rng_u32(d.seed, e0 + 1) >= d.thr ? h.y * d.scale : 0.f;
  h.z = rng_u32(d.seed, e0 + 2) >= d.thr ? h.z * d.scale : 0.f;
  h.w = rng_u32(d.seed, e0 + 3) >= d.thr ? h.w * d.scale : 0.f;
  return h;
}

// bit j set = element e0 + j is kept (all four when there is no dropout)
__device__ __forceinline__ unsigned int keep4(const Drop &d, unsigned long long e0) {
  if (d.thr == 0u) return 15u;
  return (rng_u32(d.seed, e0 + 0) >= d.thr ? 1u : 0u) | (rng_u32(d.seed, e0 + 1) >= d.thr ? 2u : 0u) |
         (rng_u32(d.seed, e0 + 2) >= d.thr ? 4u : 0u) | (rng_u32(d.seed, e0 + 3) >= d.thr ? 8u : 0u);
}

template <typename TX, typename TH, int ITERS>
__global__ __launch_bounds__(kBlock) void add_dropout_ln_fwd_kernel(
    int n_rows, int d, const TX *__restrict__ x, const TH *__restrict__ h, const float *__restrict__ gamma,
    const float *__restrict__ beta, float eps, float p_drop, unsigned int thr, unsigned long long seed,
    const unsigned long long *__restrict__ seed_dev, TX *__restrict__ y, uint16_t *__restrict__ y16,
    float *__restrict__ mean_out, float *__restrict__ rstd_out, const int *__restrict__ rows_dev,
    const float *__restrict__ post) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (rows_dev) n_rows = min(n_rows, *rows_dev);        // device-side count of leading rows that carry work
  Drop dr{thr, thr ? 1.f / (1.f - p_drop) : 1.f, seed + ((thr && seed_dev) ? *seed_dev : 0ull)};
  const float inv_d = 1.f / (float)d;
  // gamma / beta stay in registers for d <= 1024 (24 VGPRs at d = 768); wider rows fetch them with the row's own loads
  constexpr bool kHoist = ITERS <= 4;
  float4 gmh[kHoist ? ITERS : 1], bth[kHoist ? ITERS : 1];
  if constexpr (kHoist) {
#pragma unroll
    for (int i = 0; i < ITERS; ++i) {
      gmh[i] = *reinterpret_cast<const float4 *>(gamma + (i * 64 + lane) * 4);
      bth[i] = *reinterpret_cast<const float4 *>(beta + (i * 64 + lane) * 4);
    }
  }
  for (int row = blockIdx.x * kWaves + wave; row < n_rows; row += gridDim.x * kWaves) {
    const size_t base = (size_t)row * d;
    // phase 1: every load of the row in flight
    typename Raw4<TX>::type xr[ITERS];
    typename Raw4<TH>::type hr[ITERS];
    float4 pv[ITERS], gml[kHoist ? 1 : ITERS], btl[kHoist ? 1 : ITERS];
#pragma unroll
    for (int i = 0; i < ITERS; ++i) {
      const size_t e0 = base + (size_t)(i * 64 + lane) * 4;
      xr[i] = load_raw(x, e0);
      hr[i] = load_raw(h, e0);
      if constexpr (!kHoist) {
        gml[i] = *reinterpret_cast<const float4 *>(gamma + (i * 64 + lane) * 4);
        btl[i] = *reinterpret_cast<const float4 *>(beta + (i * 64 + lane) * 4);
      }
    }
    if (post) {      // y = LayerNorm(...) + post: the addend the NEXT layer would add to its input (same shape as y)
#pragma unroll
      for (int i = 0; i < ITERS; ++i) pv[i] = *reinterpret_cast<const float4 *>(post + base + (size_t)(i * 64 + lane) * 4);
    }
    GPS_LN_LOADS_ISSUED();
    // phase 2: the row lives in registers
    float4 z[ITERS];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < ITERS; ++i) {
      const size_t e0 = base + (size_t)(i * 64 + lane) * 4;
      const float4 xv = to_f4(xr[i]);
      const float4 hv = drop4(to_f4(hr[i]), dr, e0);
      z[i] = make_float4(xv.x + hv.x, xv.y + hv.y, xv.z + hv.z, xv.w + hv.w);
      s += (z[i].x + z[i].y) + (z[i].z + z[i].w);
    }
    const float mean = wave_sum(s) * inv_d;
    float v = 0.f;
#pragma unroll
    for (int i = 0; i < ITERS; ++i) {
      const float a = z[i].x - mean, b = z[i].y - mean, c = z[i].z - mean, e = z[i].w - mean;
      v += (a * a + b * b) + (c * c + e * e);
    }
    const float rstd = rsqrtf(wave_sum(v) * inv_d + eps);
    if (lane == 0) { mean_out[row] = mean; rstd_out[row] = rstd; }
#pragma unroll
    for (int i = 0; i < ITERS; ++i) {
      const int c0 = (i * 64 + lane) * 4;
      const float4 g = kHoist ? gmh[kHoist ? i : 0] : gml[kHoist ? 0 : i];
      const float4 bt = kHoist ? bth[kHoist ? i : 0] : btl[kHoist ? 0 : i];
      float4 o;
      o.x = (z[i].x - mean) * rstd * g.x + bt.x;
      o.y = (z[i].y - mean) * rstd * g.y + bt.y;
      o.z = (z[i].z - mean) * rstd * g.z + bt.z;
      o.w = (z[i].w - mean) * rstd * g.w + bt.w;
      if (post) { o.x += pv[i].x; o.y += pv[i].y; o.z += pv[i].z; o.w += pv[i].w; }
      store4(y, base + c0, o);
      if (y16) store4(y16, base + c0, o);
    }
  }
}

// dx = dz, dh = dz * keep * scale, partial dgamma/dbeta per workgroup ([gridDim.x][d] each)
template <typename TX, typename TH, int ITERS>
__global__ __launch_bounds__(kBlock, ITERS <= 3 ? 4 : 1) void add_dropout_ln_bwd_kernel(   // d <= 768: 4 waves per SIMD
   
    int n_rows, int d, const TX *__restrict__ dy, const uint16_t *__restrict__ dy16, const TX *__restrict__ x,
    const TH *__restrict__ h, const float *__restrict__ gamma, const float *__restrict__ mean_in,
    const float *__restrict__ rstd_in, float p_drop, unsigned int thr, unsigned long long seed,
    const unsigned long long *__restrict__ seed_dev, TX *__restrict__ dx, TH *__restrict__ dh,
    float *__restrict__ dgamma_part, float *__restrict__ dbeta_part, const int *__restrict__ rows_dev,
    float *__restrict__ g_out, int g_acc) {
  extern __shared__ float red[];      // [kWaves][d] reused for dgamma then dbeta
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (rows_dev) n_rows = min(n_rows, *rows_dev);        // rows past it contribute nothing to dgamma / dbeta either
  Drop dr{thr, thr ? 1.f / (1.f - p_drop) : 1.f, seed + ((thr && seed_dev) ? *seed_dev : 0ull)};
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
    // phase 1: every load of the row in flight
    typename Raw4<TX>::type xr[ITERS], gr[ITERS];
    typename Raw4<TH>::type hr[ITERS];
    uint2 g2r[ITERS];
    const float mean = mean_in[row], rstd = rstd_in[row];
#pragma unroll
    for (int i = 0; i < ITERS; ++i) {
      const size_t e0 = base + (size_t)(i * 64 + lane) * 4;
      gr[i] = load_raw(dy, e0);
      xr[i] = load_raw(x, e0);
      hr[i] = load_raw(h, e0);
    }
    if (dy16) {             // gradient that arrived through the bf16 copy of y
#pragma unroll
      for (int i = 0; i < ITERS; ++i) g2r[i] = load_raw(dy16, base + (size_t)(i * 64 + lane) * 4);
    }
    GPS_LN_LOADS_ISSUED();
    // phase 2
    float4 zh[ITERS], a[ITERS];
    unsigned int keep = 0u;          // 4 bits per step: the dropout decisions, drawn once for dz -> dh below
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < ITERS; ++i) {
      const size_t e0 = base + (size_t)(i * 64 + lane) * 4;
      float4 g = to_f4(gr[i]);
      if (dy16) {
        const float4 g2 = to_f4(g2r[i]);
        g = make_float4(g.x + g2.x, g.y + g2.y, g.z + g2.z, g.w + g2.w);
      }
      if (g_out) {                                                 // gradient of the output = gradient of a post-addend
        float4 go = g;
        if (g_acc) {    // the post-addend's gradient so far (the later layers' launches stored / added theirs); read here, not with
                        // the row's other loads: 12 more live registers there spill at 4 waves per SIMD (tests/test_asm_audit.py)
          const float4 o = *reinterpret_cast<const float4 *>(g_out + e0);
          go = make_float4(g.x + o.x, g.y + o.y, g.z + o.z, g.w + o.w);
        }
        *reinterpret_cast<float4 *>(g_out + e0) = go;
      }
      const float4 xv = to_f4(xr[i]);
      float4 hv = to_f4(hr[i]);
      const unsigned int m = keep4(dr, e0);
      keep |= m << (4 * i);
      if (dr.thr) {
        hv.x = (m & 1u) ? hv.x * dr.scale : 0.f;
        hv.y = (m & 2u) ? hv.y * dr.scale : 0.f;
        hv.z = (m & 4u) ? hv.z * dr.scale : 0.f;
        hv.w = (m & 8u) ? hv.w * dr.scale : 0.f;
      }
      zh[i] = make_float4((xv.x + hv.x - mean) * rstd, (xv.y + hv.y - mean) * rstd,
                          (xv.z + hv.z - mean) * rstd, (xv.w + hv.w - mean) * rstd);
      a[i] = make_float4(g.x * gm[i].x, g.y * gm[i].y, g.z * gm[i].z, g.w * gm[i].w);
      s1 += (a[i].x + a[i].y) + (a[i].z + a[i].w);
      s2 += (a[i].x * zh[i].x + a[i].y * zh[i].y) + (a[i].z * zh[i].z + a[i].w * zh[i].w);
      gacc[i].x += g.x * zh[i].x; gacc[i].y += g.y * zh[i].y;
      gacc[i].z += g.z * zh[i].z; gacc[i].w += g.w * zh[i].w;
      bacc[i].x += g.x; bacc[i].y += g.y; bacc[i].z += g.z; bacc[i].w += g.w;
    }
    s1 = wave_sum(s1) * inv_d;
    s2 = wave_sum(s2) * inv_d;
#pragma unroll
    for (int i = 0; i < ITERS; ++i) {
      const size_t e0 = base + (size_t)(i * 64 + lane) * 4;
      float4 dz;
      dz.x = rstd * (a[i].x - s1 - zh[i].x * s2);
      dz.y = rstd * (a[i].y - s1 - zh[i].y * s2);
      dz.z = rstd * (a[i].z - s1 - zh[i].z * s2);
      dz.w = rstd * (a[i].w - s1 - zh[i].w * s2);
      store4(dx, e0, dz);
      float4 dhv = dz;
      if (dr.thr) {
        const unsigned int m = keep >> (4 * i);
        dhv = make_float4(dz.x * ((m & 1u) ? dr.scale : 0.f), dz.y * ((m & 2u) ? dr.scale : 0.f),
                          dz.z * ((m & 4u) ? dr.scale : 0.f), dz.w * ((m & 8u) ? dr.scale : 0.f));
      }
      store4(dh, e0, dhv);
    }
  }
  // cross-wave reduction of the column sums, then one partial row per workgroup
  for (int pass = 0; pass < 2; ++pass) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < ITERS; ++i)
      *reinterpret_cast<float4 *>(red + wave * d + (i * 64 + lane) * 4) = pass == 0 ? gacc[i] : bacc[i];
    __syncthreads();
    float *dst = (pass == 0 ? dgamma_part : dbeta_part) + (size_t)blockIdx.x * d;
    for (int c = threadIdx.x; c < d; c += kBlock) {
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < kWaves; ++w) t += red[w * d + c];
      dst[c] = t;
    }
  }
}

// out[c] = sum_r part[r][c] for the 2*d columns of the [dgamma | dbeta] partial rows
// (part is [2][parts][d]; out is [2][d]).  Workgroup (cg, rg) owns 64 columns x one slice of the rows: its 4 waves
// split the slice (256 contiguous bytes per row and wave, 8 rows in flight per thread), combine in LDS and write
// one row of second-level partials; the LAST workgroup of a column group to arrive (ticket counter) adds those
// rows in slice order, so the result does not depend on the arrival order.  scratch: [kRedSlices][2 d] floats +
// one counter per column group, zero before the first launch and left zero by every launch.
constexpr int kRedSlices = 16;
__global__ __launch_bounds__(kBlock) void reduce_partials_kernel(int parts, int d, const float *__restrict__ part,
                                                                  float *__restrict__ out, float *__restrict__ part2,
                                                                  unsigned int *__restrict__ tickets) {
  __shared__ float red[kWaves][64];
  __shared__ int last;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;           // column in [0, 2 d)
  const int slices = gridDim.y;
  const int per = (parts + slices - 1) / slices;
  const int r_begin = blockIdx.y * per, r_end = min(parts, r_begin + per);
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (c < 2 * d) {
    const int which = c / d, col = c - which * d;
    const float *p = part + (size_t)which * parts * d + col;
    int r = r_begin + wave;
    for (; r + 7 * kWaves < r_end; r += 8 * kWaves) {
#pragma unroll
      for (int u = 0; u < 8; ++u) acc[u] += p[(size_t)(r + u * kWaves) * d];
    }
    for (; r < r_end; r += kWaves) acc[0] += p[(size_t)r * d];
  }
  red[wave][lane] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
  __syncthreads();
  if (slices == 1) {
    if (wave == 0 && c < 2 * d) out[c] = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
    return;
  }
  // Hand-off without L2 write-back / invalidate episodes (the previous kernel leaves > 100 MB dirty): the second-level
  // rows are written through (agent-scope relaxed atomic stores = sc1), waited for, then the ticket is taken; the
  // last arriver reads them with agent-scope loads.
  if (wave == 0) {
    if (c < 2 * d)
      __hip_atomic_store(part2 + (size_t)blockIdx.y * 2 * d + c, (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]),
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) {
      const unsigned int t = __hip_atomic_fetch_add(&tickets[blockIdx.x], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      last = (t == (unsigned int)slices - 1u);
      if (last) __hip_atomic_store(&tickets[blockIdx.x], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  __syncthreads();
  if (!last || wave != 0 || c >= 2 * d) return;
  // all slice rows are requested before the first add (16 independent loads in flight, not 16 round trips)
  float v[kRedSlices];
#pragma unroll
  for (int s = 0; s < kRedSlices; ++s)
    v[s] = (s < slices) ? __hip_atomic_load(part2 + (size_t)s * 2 * d + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.f;
  float sum = v[0];
#pragma unroll
  for (int s = 1; s < kRedSlices; ++s) sum += v[s];
  out[c] = sum;
}

// [r4] The same reduction for MANY LayerNorms in one launch (blockIdx.z = problem): the 24 + fused residual-LayerNorm
// backward passes of a step each ended in their own 7 us reduce launch (0.19 ms per step); their parameter gradients are
// only needed by the optimizer, so the host defers them (modules/layers/fused_norm.py) and sums them together.  Results
// go straight into gamma.grad / beta.grad (plain store, or added to what is there).  Same two-level scheme and the same
// fixed summation order per problem as reduce_partials_kernel.
struct LnRedProblem {          // 40 bytes
  const float *part;           // [2][parts][d]
  float *out_g, *out_b;        // (d) each
  int parts, accumulate;
};
constexpr int kLnRedMax = 64;
struct LnRedArgs { LnRedProblem p[kLnRedMax]; };

__global__ __launch_bounds__(kBlock) void reduce_partials_grouped_kernel(const LnRedArgs args, int d, float *__restrict__ part2_all,
                                                                          unsigned int *__restrict__ tickets_all) {
  __shared__ float red[kWaves][64];
  __shared__ int last;
  const LnRedProblem P = args.p[blockIdx.z];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;           // column in [0, 2 d)
  const int groups = gridDim.x;
  const int parts = P.parts;
  int slices = (parts + 63) / 64;
  slices = slices > (int)gridDim.y ? (int)gridDim.y : slices;       // this problem's slices (<= kRedSlices)
  if ((int)blockIdx.y >= slices) return;
  float *part2 = part2_all + (size_t)blockIdx.z * kRedSlices * 2 * d;
  unsigned int *tickets = tickets_all + (size_t)blockIdx.z * groups;
  const int per = (parts + slices - 1) / slices;
  const int r_begin = blockIdx.y * per, r_end = min(parts, r_begin + per);
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (c < 2 * d) {
    const int which = c / d, col = c - which * d;
    const float *p = P.part + (size_t)which * parts * d + col;
    int r = r_begin + wave;
    for (; r + 7 * kWaves < r_end; r += 8 * kWaves) {
#pragma unroll
      for (int u = 0; u < 8; ++u) acc[u] += p[(size_t)(r + u * kWaves) * d];
    }
    for (; r < r_end; r += kWaves) acc[0] += p[(size_t)r * d];
  }
  red[wave][lane] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
  __syncthreads();
  auto emit = [&](float v) {
    float *dst = c < d ? P.out_g + c : P.out_b + (c - d);
    *dst = P.accumulate ? *dst + v : v;
  };
  if (slices == 1) {
    if (wave == 0 && c < 2 * d) emit((red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]));
    return;
  }
  if (wave == 0) {
    if (c < 2 * d)
      __hip_atomic_store(part2 + (size_t)blockIdx.y * 2 * d + c, (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]),
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) {
      const unsigned int t = __hip_atomic_fetch_add(&tickets[blockIdx.x], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      last = (t == (unsigned int)slices - 1u);
      if (last) __hip_atomic_store(&tickets[blockIdx.x], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  __syncthreads();
  if (!last || wave != 0 || c >= 2 * d) return;
  float v[kRedSlices];
#pragma unroll
  for (int s2 = 0; s2 < kRedSlices; ++s2)
    v[s2] = (s2 < slices) ? __hip_atomic_load(part2 + (size_t)s2 * 2 * d + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.f;
  float sum = v[0];
#pragma unroll
  for (int s2 = 1; s2 < kRedSlices; ++s2) sum += v[s2];
  emit(sum);
}

// ---- y = x / max(||x||_2, eps) per row (F.normalize(x, p=2, dim=-1)) and its backward ---------------------------------
// One wave per row; the row is read once (d <= 2048: kept in registers), forward also stores 1 / max(norm, eps).
// backward: dx = inv * (dy - y (y . dy)), or inv * dy where the norm was clamped (torch treats the clamp as a constant).
__global__ __launch_bounds__(kBlock) void l2norm_fwd_kernel(int n_rows, int d, const float *__restrict__ x, float eps,
                                                            float *__restrict__ y, float *__restrict__ inv_out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int row = blockIdx.x * kWaves + wave; row < n_rows; row += gridDim.x * kWaves) {
    const size_t base = (size_t)row * d;
    float4 v[kMaxIter];
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < kMaxIter; ++i) {
      const int c0 = (i * 64 + lane) * 4;
      v[i] = c0 < d ? load4(x, base + c0) : make_float4(0.f, 0.f, 0.f, 0.f);
      ss += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
    }
    const float inv = 1.f / fmaxf(sqrtf(wave_sum(ss)), eps);
    if (lane == 0) inv_out[row] = inv;
#pragma unroll
    for (int i = 0; i < kMaxIter; ++i) {
      const int c0 = (i * 64 + lane) * 4;
      if (c0 < d) store4(y, base + c0, make_float4(v[i].x * inv, v[i].y * inv, v[i].z * inv, v[i].w * inv));
    }
  }
}
__global__ __launch_bounds__(kBlock) void l2norm_bwd_kernel(int n_rows, int d, const float *__restrict__ dy,
                                                            const float *__restrict__ y, const float *__restrict__ inv_in,
                                                            float eps, float *__restrict__ dx) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float inv_cap = 1.f / eps;
  for (int row = blockIdx.x * kWaves + wave; row < n_rows; row += gridDim.x * kWaves) {
    const size_t base = (size_t)row * d;
    float4 g[kMaxIter], yv[kMaxIter];
    float dot = 0.f;
#pragma unroll
    for (int i = 0; i < kMaxIter; ++i) {
      const int c0 = (i * 64 + lane) * 4;
      g[i] = c0 < d ? load4(dy, base + c0) : make_float4(0.f, 0.f, 0.f, 0.f);
      yv[i] = c0 < d ? load4(y, base + c0) : make_float4(0.f, 0.f, 0.f, 0.f);
      dot += (g[i].x * yv[i].x + g[i].y * yv[i].y) + (g[i].z * yv[i].z + g[i].w * yv[i].w);
    }
    const float inv = inv_in[row];
    const float tot = wave_sum(dot);
    dot = inv >= inv_cap ? 0.f : tot;                    // clamped norm: a constant divisor
#pragma unroll
    for (int i = 0; i < kMaxIter; ++i) {
      const int c0 = (i * 64 + lane) * 4;
      if (c0 < d)
        store4(dx, base + c0, make_float4(inv * (g[i].x - yv[i].x * dot), inv * (g[i].y - yv[i].y * dot),
                                          inv * (g[i].z - yv[i].z * dot), inv * (g[i].w - yv[i].w * dot)));
    }
  }
}

inline int grid_rows(int n_rows) {
  int g = (n_rows + kWaves - 1) / kWaves;
  return g > 1024 ? 1024 : (g < 1 ? 1 : g);
}

}  // namespace gps_ln

extern "C" {

int gps_ln_partial_rows(int n_rows) { return gps_ln::grid_rows(n_rows); }

long long gps_ln_reduce_scratch_bytes(int d) {
  return (long long)gps_ln::kRedSlices * 2 * d * 4 + (long long)((2 * d + 63) / 64) * 4;
}

int gps_ln_reduce_partials(int parts, int d, const float *part, float *out, void *scratch, gps_stream_t stream) {
  if (parts < 1 || d < 1 || !part || !out) return GPS_ERR_INVALID_ARGUMENT;
  const int groups = (2 * d + 63) / 64;
  // few partial rows (or no scratch): one workgroup per column group does the whole sum
  int slices = scratch ? (parts + 63) / 64 : 1;
  if (slices > gps_ln::kRedSlices) slices = gps_ln::kRedSlices;
  float *part2 = reinterpret_cast<float *>(scratch);
  unsigned int *tickets = scratch ? reinterpret_cast<unsigned int *>(part2 + (size_t)gps_ln::kRedSlices * 2 * d) : nullptr;
  hipLaunchKernelGGL(gps_ln::reduce_partials_kernel, dim3(groups, slices), dim3(gps_ln::kBlock), 0,
                     (hipStream_t)stream, parts, d, part, out, part2, tickets);
  return hipGetLastError() == hipSuccess ? GPS_OK : GPS_ERR_LAUNCH;
}

int gps_ln_reduce_partials_grouped(const gps_ln_reduce_problem *problems, int n_problems, int d, void *scratch,
                                   gps_stream_t stream) {
  using namespace gps_ln;
  if (n_problems < 0 || d < 1 || (n_problems > 0 && (!problems || !scratch))) return GPS_ERR_INVALID_ARGUMENT;
  const int groups = (2 * d + 63) / 64;
  for (int base = 0; base < n_problems; base += kLnRedMax) {
    const int n = n_problems - base < kLnRedMax ? n_problems - base : kLnRedMax;
    LnRedArgs args = {};
    int max_slices = 1;
    for (int i = 0; i < n; ++i) {
      const gps_ln_reduce_problem &q = problems[base + i];
      if (q.parts < 1 || !q.part || !q.out_gamma || !q.out_beta) return GPS_ERR_INVALID_ARGUMENT;
      args.p[i].part = q.part;
      args.p[i].out_g = q.out_gamma;
      args.p[i].out_b = q.out_beta;
      args.p[i].parts = q.parts;
      args.p[i].accumulate = q.accumulate ? 1 : 0;
      int sl = (q.parts + 63) / 64;
      sl = sl > kRedSlices ? kRedSlices : sl;
      max_slices = sl > max_slices ? sl : max_slices;
    }
    float *part2 = reinterpret_cast<float *>(scratch);
    unsigned int *tickets = reinterpret_cast<unsigned int *>(part2 + (size_t)kLnRedMax * kRedSlices * 2 * d);
    hipLaunchKernelGGL(reduce_partials_grouped_kernel, dim3(groups, max_slices, n), dim3(kBlock), 0, (hipStream_t)stream, args, d,
                       part2, tickets);
    if (hipGetLastError() != hipSuccess) return GPS_ERR_LAUNCH;
  }
  return GPS_OK;
}

long long gps_ln_reduce_grouped_scratch_bytes(int d) {
  return (long long)gps_ln::kLnRedMax * (gps_ln::kRedSlices * 2 * d * 4 + ((2 * d + 63) / 64) * 4);
}

int gps_l2_normalize_forward(int n_rows, int d, const float *x, float eps, float *y, float *inv_norm, gps_stream_t stream) {
  if (n_rows < 0 || d < 1 || !(eps > 0.f)) return GPS_ERR_INVALID_ARGUMENT;
  if ((d & 3) || d > 256 * gps_ln::kMaxIter) return GPS_ERR_UNSUPPORTED;
  if (n_rows == 0) return GPS_OK;
  if (!x || !y || !inv_norm || (((uintptr_t)x | (uintptr_t)y) & 15)) return GPS_ERR_INVALID_ARGUMENT;
  hipLaunchKernelGGL(gps_ln::l2norm_fwd_kernel, dim3(gps_ln::grid_rows(n_rows)), dim3(gps_ln::kBlock), 0, (hipStream_t)stream,
                     n_rows, d, x, eps, y, inv_norm);
  return hipGetLastError() == hipSuccess ? GPS_OK : GPS_ERR_LAUNCH;
}

int gps_l2_normalize_backward(int n_rows, int d, const float *dy, const float *y, const float *inv_norm, float eps, float *dx,
                              gps_stream_t stream) {
  if (n_rows < 0 || d < 1 || !(eps > 0.f)) return GPS_ERR_INVALID_ARGUMENT;
  if ((d & 3) || d > 256 * gps_ln::kMaxIter) return GPS_ERR_UNSUPPORTED;
  if (n_rows == 0) return GPS_OK;
  if (!dy || !y || !inv_norm || !dx || (((uintptr_t)dy | (uintptr_t)y | (uintptr_t)dx) & 15)) return GPS_ERR_INVALID_ARGUMENT;
  hipLaunchKernelGGL(gps_ln::l2norm_bwd_kernel, dim3(gps_ln::grid_rows(n_rows)), dim3(gps_ln::kBlock), 0, (hipStream_t)stream,
                     n_rows, d, dy, y, inv_norm, eps, dx);
  return hipGetLastError() == hipSuccess ? GPS_OK : GPS_ERR_LAUNCH;
}

int gps_add_dropout_layernorm_forward(int n_rows, int d, int x_bf16, int h_bf16, const void *x, const void *h,
                                      const float *gamma, const float *beta, float eps, float p_drop,
                                      unsigned long long seed, const void *seed_dev, void *y, void *y_bf16,
                                      float *mean, float *rstd, gps_stream_t stream) {
  return gps_add_dropout_layernorm_forward_rows(n_rows, d, x_bf16, h_bf16, x, h, gamma, beta, eps, p_drop, seed, seed_dev, y,
                                                y_bf16, mean, rstd, nullptr, stream);
}

int gps_add_dropout_layernorm_forward_rows(int n_rows, int d, int x_bf16, int h_bf16, const void *x, const void *h,
                                           const float *gamma, const float *beta, float eps, float p_drop,
                                           unsigned long long seed, const void *seed_dev, void *y, void *y_bf16,
                                           float *mean, float *rstd, const int *rows_dev, gps_stream_t stream) {
  return gps_add_dropout_layernorm_forward_post(n_rows, d, x_bf16, h_bf16, x, h, gamma, beta, eps, p_drop, seed, seed_dev, y,
                                                y_bf16, mean, rstd, rows_dev, nullptr, stream);
}

int gps_add_dropout_layernorm_forward_post(int n_rows, int d, int x_bf16, int h_bf16, const void *x, const void *h,
                                           const float *gamma, const float *beta, float eps, float p_drop,
                                           unsigned long long seed, const void *seed_dev, void *y, void *y_bf16,
                                           float *mean, float *rstd, const int *rows_dev, const float *post,
                                           gps_stream_t stream) {
  if (post && (x_bf16 || ((uintptr_t)post & 15))) return GPS_ERR_UNSUPPORTED;      // fp32 rows only
  if (n_rows < 0 || d < 1 || p_drop < 0.f || p_drop >= 1.f) return GPS_ERR_INVALID_ARGUMENT;
  if ((d & 255) || d > 256 * gps_ln::kMaxIter || ((d >> 8) > 4 && (d >> 8) != 8)) return GPS_ERR_UNSUPPORTED;
  if (n_rows == 0) return GPS_OK;
  if (!x || !h || !gamma || !beta || !y || !mean || !rstd) return GPS_ERR_INVALID_ARGUMENT;
  const unsigned int thr = p_drop > 0.f ? (unsigned int)((double)p_drop * 4294967296.0) : 0u;
  const dim3 grid(gps_ln::grid_rows(n_rows)), block(gps_ln::kBlock);
  hipStream_t s = (hipStream_t)stream;
  const unsigned long long *sd = (const unsigned long long *)seed_dev;
#define GPS_LN_FWD_I(TX, TH, IT)                                                                                  \
  hipLaunchKernelGGL((gps_ln::add_dropout_ln_fwd_kernel<TX, TH, IT>), grid, block, 0, s, n_rows, d, (const TX *)x, \
                     (const TH *)h, gamma, beta, eps, p_drop, thr, seed, sd, (TX *)y, (uint16_t *)y_bf16, mean, rstd, rows_dev, post)
#define GPS_LN_FWD(TX, TH)                          \
  do { switch (d >> 8) {                            \
    case 1: GPS_LN_FWD_I(TX, TH, 1); break;         \
    case 2: GPS_LN_FWD_I(TX, TH, 2); break;         \
    case 3: GPS_LN_FWD_I(TX, TH, 3); break;         \
    case 4: GPS_LN_FWD_I(TX, TH, 4); break;         \
    case 8: GPS_LN_FWD_I(TX, TH, 8); break;         \
    default: return GPS_ERR_UNSUPPORTED;            \
  } } while (0)
  if (x_bf16 && h_bf16) GPS_LN_FWD(uint16_t, uint16_t);
  else if (x_bf16) GPS_LN_FWD(uint16_t, float);
  else if (h_bf16) GPS_LN_FWD(float, uint16_t);
  else GPS_LN_FWD(float, float);
#undef GPS_LN_FWD
#undef GPS_LN_FWD_I
  return hipGetLastError() == hipSuccess ? GPS_OK : GPS_ERR_LAUNCH;
}

int gps_add_dropout_layernorm_backward(int n_rows, int d, int x_bf16, int h_bf16, const void *dy,
                                       const void *dy_bf16, const void *x, const void *h, const float *gamma,
                                       const float *mean, const float *rstd, float p_drop,
                                       unsigned long long seed, const void *seed_dev, void *dx, void *dh,
                                       float *dgamma_part, float *dbeta_part, gps_stream_t stream) {
  return gps_add_dropout_layernorm_backward_rows(n_rows, d, x_bf16, h_bf16, dy, dy_bf16, x, h, gamma, mean, rstd, p_drop, seed,
                                                 seed_dev, dx, dh, dgamma_part, dbeta_part, nullptr, stream);
}

int gps_add_dropout_layernorm_backward_rows(int n_rows, int d, int x_bf16, int h_bf16, const void *dy,
                                            const void *dy_bf16, const void *x, const void *h, const float *gamma,
                                            const float *mean, const float *rstd, float p_drop,
                                            unsigned long long seed, const void *seed_dev, void *dx, void *dh,
                                            float *dgamma_part, float *dbeta_part, const int *rows_dev,
                                            gps_stream_t stream) {
  return gps_add_dropout_layernorm_backward_post(n_rows, d, x_bf16, h_bf16, dy, dy_bf16, x, h, gamma, mean, rstd, p_drop, seed,
                                                 seed_dev, dx, dh, dgamma_part, dbeta_part, rows_dev, nullptr, stream);
}

int gps_add_dropout_layernorm_backward_post(int n_rows, int d, int x_bf16, int h_bf16, const void *dy,
                                            const void *dy_bf16, const void *x, const void *h, const float *gamma,
                                            const float *mean, const float *rstd, float p_drop,
                                            unsigned long long seed, const void *seed_dev, void *dx, void *dh,
                                            float *dgamma_part, float *dbeta_part, const int *rows_dev, float *dpost,
                                            gps_stream_t stream) {
  return gps_add_dropout_layernorm_backward_post_acc(n_rows, d, x_bf16, h_bf16, dy, dy_bf16, x, h, gamma, mean, rstd, p_drop, seed,
                                                     seed_dev, dx, dh, dgamma_part, dbeta_part, rows_dev, dpost, 0, stream);
}

int gps_add_dropout_layernorm_backward_post_acc(int n_rows, int d, int x_bf16, int h_bf16, const void *dy,
                                                const void *dy_bf16, const void *x, const void *h, const float *gamma,
                                                const float *mean, const float *rstd, float p_drop,
                                                unsigned long long seed, const void *seed_dev, void *dx, void *dh,
                                                float *dgamma_part, float *dbeta_part, const int *rows_dev, float *dpost,
                                                int dpost_accumulate, gps_stream_t stream) {
  if (dpost && (x_bf16 || ((uintptr_t)dpost & 15))) return GPS_ERR_UNSUPPORTED;
  if (dpost_accumulate && !dpost) return GPS_ERR_INVALID_ARGUMENT;
  if (n_rows < 0 || d < 1 || p_drop < 0.f || p_drop >= 1.f) return GPS_ERR_INVALID_ARGUMENT;
  if ((d & 255) || d > 256 * gps_ln::kMaxIter || ((d >> 8) > 4 && (d >> 8) != 8)) return GPS_ERR_UNSUPPORTED;
  if (n_rows == 0) return GPS_OK;
  if (!dy || !x || !h || !gamma || !mean || !rstd || !dx || !dh || !dgamma_part || !dbeta_part)
    return GPS_ERR_INVALID_ARGUMENT;
  const unsigned int thr = p_drop > 0.f ? (unsigned int)((double)p_drop * 4294967296.0) : 0u;
  const dim3 grid(gps_ln::grid_rows(n_rows)), block(gps_ln::kBlock);
  const size_t lds = sizeof(float) * gps_ln::kWaves * d;
  hipStream_t s = (hipStream_t)stream;
  const unsigned long long *sd = (const unsigned long long *)seed_dev;
#define GPS_LN_BWD_I(TX, TH, IT)                                                                                    \
  hipLaunchKernelGGL((gps_ln::add_dropout_ln_bwd_kernel<TX, TH, IT>), grid, block, lds, s, n_rows, d, (const TX *)dy, \
                     (const uint16_t *)dy_bf16, (const TX *)x, (const TH *)h, gamma, mean, rstd, p_drop, thr,       \
                     seed, sd, (TX *)dx, (TH *)dh, dgamma_part, dbeta_part, rows_dev, dpost, dpost_accumulate ? 1 : 0)
#define GPS_LN_BWD(TX, TH)                          \
  do { switch (d >> 8) {                            \
    case 1: GPS_LN_BWD_I(TX, TH, 1); break;         \
    case 2: GPS_LN_BWD_I(TX, TH, 2); break;         \
    case 3: GPS_LN_BWD_I(TX, TH, 3); break;         \
    case 4: GPS_LN_BWD_I(TX, TH, 4); break;         \
    case 8: GPS_LN_BWD_I(TX, TH, 8); break;         \
    default: return GPS_ERR_UNSUPPORTED;            \
  } } while (0)
  if (x_bf16 && h_bf16) GPS_LN_BWD(uint16_t, uint16_t);
  else if (x_bf16) GPS_LN_BWD(uint16_t, float);
  else if (h_bf16) GPS_LN_BWD(float, uint16_t);
  else GPS_LN_BWD(float, float);
#undef GPS_LN_BWD
#undef GPS_LN_BWD_I
  return hipGetLastError() == hipSuccess ? GPS_OK : GPS_ERR_LAUNCH;
}

}  // extern "C"
