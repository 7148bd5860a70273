// gps_contrastive.hip -- the two contrastive losses of the GPS pre-training step as two launches each (gfx950).
//
// Reference: optim/loss/contra_loss.py
//   :22-43  TextObjWithinBatch    F.normalize both sides, einsum('bod,bd->bo'), masked_fill(-inf), F.cross_entropy
//   :11-17  _symmetric_clip_loss  (CE(s a b^T, arange) + CE(s b a^T, arange)) / 2, a / b normalised rows (:60, :82-83),
//           s = clamp(logit_scale, max = 100) (:57, :79)
// The tensors are tiny (64 scenes x 80 objects x 768, 64 x 64 logits), the torch form is ~55 launches of 3 - 10 us per
// step (normalisations, casts, two bmm / four mm on hipBLASLt, log-softmax, nll, their backward twins).  Here each loss
// is ONE forward launch (normalisation factors, logits, log-sum-exp, mean -- the mean taken by the last workgroup to
// arrive, in row order) and ONE backward launch that produces the gradients of the RAW (un-normalised) inputs directly:
// with n = x inv (inv = 1 / max(|x|, eps)) and upstream dn:  dx = inv (dn - n (n . dn))  (no projection term when the
// norm was clamped, torch's clamp_min semantics), and n . dn is a sum over the logits' own gradients, so no second pass
// over the feature rows is needed.  fp32 throughout (the reference under bf16 autocast rounds the operands of the
// einsum / matmul to bf16; the fp32 oracle port does not).  Latency-bound: 64 - 512 workgroups of 256 threads.
// Arrival tickets: one unsigned int per loss, zero before the first launch and left zero by every launch.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gps_hip.h"

namespace gps_contra {

constexpr int kBlock = 256, kWaves = 4;

__device__ __forceinline__ float wave_sum(float v) {
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
  for (int off = 32; off >= 1; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}
// sum over the workgroup, the same value in every thread; `red` = kWaves floats of LDS
__device__ __forceinline__ float block_sum(float v, float *red) {
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  const float r = (red[0] + red[1]) + (red[2] + red[3]);
  __syncthreads();
  return r;
}
// the same for a workgroup of NW waves (`red` = NW floats), partial sums added in wave order
template <int NW>
__device__ __forceinline__ float block_sum_n(float v, float *red) {
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float r = 0.f;
#pragma unroll
  for (int w = 0; w < NW; ++w) r += red[w];
  __syncthreads();
  return r;
}
// [r6] the two forward kernels run 16 waves per workgroup: a wave's trips over the other rows are dependent memory round
// trips (4 waves: 5 / 8 of them in a row, 21.5 / 24.5 us per launch for a few MB)
constexpr int kFwdWaves = 16;
__device__ __forceinline__ float dot4(const float4 a, const float4 b) { return (a.x * b.x + a.y * b.y) + (a.z * b.z + a.w * b.w); }

// publish `mine` as row `idx` of `rows`, take a ticket; the last of `count` arrivals returns true (thread 0 only) after
// which rows[0 .. count) may be read with agent-scope loads.  The ticket word is left at zero.
__device__ __forceinline__ bool publish_and_last(float *rows, int idx, float mine, unsigned int *ticket, int count) {
  __hip_atomic_store(rows + idx, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned int t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const bool last = t == (unsigned int)count - 1u;
  if (last) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return last;
}
__device__ __forceinline__ float peek(const float *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// ---------------------------------------------------------------------------------------------------------------------
// TextObjWithinBatch.  One workgroup per scene b.
//   cosv[b][o] = <obj_n[b][o], text_n[b]>     prob[b][o] = softmax over the unmasked objects    inv_o, inv_t: 1 / max(norm, eps)
//   loss = mean over the scenes whose label != ignore_index of (lse_b - cosv[b][label_b])       scal[0] = loss, scal[1] = count
// ---------------------------------------------------------------------------------------------------------------------
template <int NW>
__global__ __launch_bounds__(64 * NW) void text_obj_fwd_kernel(int B, int O, int D, const float *__restrict__ obj,
                                                               const float *__restrict__ text,
                                                               const int64_t *__restrict__ labels,
                                                               const uint8_t *__restrict__ masks, float eps,
                                                               long long ignore_index, float *__restrict__ cosv,
                                                               float *__restrict__ prob, float *__restrict__ inv_o,
                                                               float *__restrict__ inv_t, float *__restrict__ loss_rows,
                                                               float *__restrict__ scal, unsigned int *__restrict__ ticket) {
  extern __shared__ __attribute__((aligned(16))) float sm[];          // D floats: text_n | O floats: logits | NW
  float *tn = sm, *lg = sm + D, *red = sm + D + O;
  const int b = blockIdx.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int D4 = D >> 2;
  const float4 *t4 = reinterpret_cast<const float4 *>(text + (size_t)b * D);
  float4 *tn4 = reinterpret_cast<float4 *>(tn);
  float ss = 0.f;
  for (int i = threadIdx.x; i < D4; i += 64 * NW) {
    const float4 v = t4[i];
    tn4[i] = v;
    ss += dot4(v, v);
  }
  ss = block_sum_n<NW>(ss, red);
  const float it = 1.f / fmaxf(sqrtf(ss), eps);
  for (int i = threadIdx.x; i < D4; i += 64 * NW) {
    float4 v = tn4[i];
    v.x *= it; v.y *= it; v.z *= it; v.w *= it;
    tn4[i] = v;
  }
  __syncthreads();
  // four objects per wave and trip: their rows are requested together (one object at a time was one dependent memory
  // round trip + two wave reductions per object, 20 in a row at O = 80: 37 us for a 16 MB read)
  for (int o0 = w; o0 < O; o0 += 4 * NW) {
    const float4 *x4[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) x4[u] = reinterpret_cast<const float4 *>(obj + ((size_t)b * O + min(o0 + u * NW, O - 1)) * D);
    float so[4] = {0.f, 0.f, 0.f, 0.f}, dt[4] = {0.f, 0.f, 0.f, 0.f};
    for (int i = lane; i < D4; i += 64) {
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = x4[u][i];
      const float4 t = tn4[i];
#pragma unroll
      for (int u = 0; u < 4; ++u) { so[u] += dot4(v[u], v[u]); dt[u] += dot4(v[u], t); }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int o = o0 + u * NW;
      const float s2 = wave_sum(so[u]), d2 = wave_sum(dt[u]);
      const float io = 1.f / fmaxf(sqrtf(s2), eps);
      if (lane == 0 && o < O) {
        inv_o[(size_t)b * O + o] = io;
        lg[o] = d2 * io;
        cosv[(size_t)b * O + o] = d2 * io;
      }
    }
  }
  __syncthreads();
  if (w != 0) return;
  const uint8_t *mk = masks + (size_t)b * O;
  float m = -INFINITY;
  for (int o = lane; o < O; o += 64) m = fmaxf(m, mk[o] ? lg[o] : -INFINITY);
  m = wave_max(m);
  float s = 0.f;
  for (int o = lane; o < O; o += 64) s += mk[o] ? __expf(lg[o] - m) : 0.f;
  s = wave_sum(s);
  const float lse = m + __logf(s);
  for (int o = lane; o < O; o += 64) prob[(size_t)b * O + o] = mk[o] ? __expf(lg[o] - lse) : 0.f;
  const long long lab = labels[b];
  const bool counted = lab != ignore_index && lab >= 0 && lab < O;
  int last = 0;
  if (lane == 0) {
    inv_t[b] = it;
    // a label on a masked object has logit -inf: loss +inf, as F.cross_entropy gives
    const float mine = counted ? lse - (mk[lab] ? lg[lab] : -INFINITY) : 0.f;
    last = publish_and_last(loss_rows, b, mine, ticket, B) ? 1 : 0;
  }
  if (!__shfl(last, 0, 64)) return;
  // the last scene to arrive: the mean, by the whole wave (lane-strided partial sums in a fixed order; one lane walking
  // the B values paid a memory round trip per value: 30 us)
  float total = 0.f;
  int cnt = 0;
  for (int i = lane; i < B; i += 64) {
    total += peek(loss_rows + i);
    const long long li = labels[i];
    cnt += (li != ignore_index && li >= 0 && li < O) ? 1 : 0;
  }
  total = wave_sum(total);
  for (int off = 32; off >= 1; off >>= 1) cnt += __shfl_xor(cnt, off, 64);
  if (lane == 0) {
    scal[0] = total / (float)cnt;
    scal[1] = (float)cnt;
  }
}

// grid (G + 1, B), G = ceil(O / 16) when the objects need a gradient, else 0: blockIdx.x < G -> gradient of 16 objects
// (4 per wave); blockIdx.x == G -> gradient of the scene's text row (sequential over the objects: fixed order).
//   c_o = (prob_o - [o == label]) g / count;  dobj_o = inv_o c_o (text_n - obj_n_o cos_o);
//   dtext = inv_t (sum_o c_o obj_n_o - text_n sum_o c_o cos_o)
__global__ __launch_bounds__(kBlock) void text_obj_bwd_kernel(int B, int O, int D, int G, const float *__restrict__ obj,
                                                               const float *__restrict__ text,
                                                               const int64_t *__restrict__ labels, float eps,
                                                               long long ignore_index, const float *__restrict__ cosv,
                                                               const float *__restrict__ prob, const float *__restrict__ inv_o,
                                                               const float *__restrict__ inv_t, const float *__restrict__ scal,
                                                               const float *__restrict__ gout, float *__restrict__ dobj,
                                                               float *__restrict__ dtext) {
  extern __shared__ __attribute__((aligned(16))) float sm[];          // D floats: text_n | O floats: c_o inv_o | kWaves
  float *tn = sm, *co = sm + D, *red = sm + D + O;
  const int b = blockIdx.y, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int D4 = D >> 2;
  const float it = inv_t[b];
  const float4 *t4 = reinterpret_cast<const float4 *>(text + (size_t)b * D);
  float4 *tn4 = reinterpret_cast<float4 *>(tn);
  for (int i = threadIdx.x; i < D4; i += kBlock) {
    float4 v = t4[i];
    v.x *= it; v.y *= it; v.z *= it; v.w *= it;
    tn4[i] = v;
  }
  const long long lab = labels[b];
  const bool counted = lab != ignore_index && lab >= 0 && lab < O;
  const float gs = counted ? gout[0] / scal[1] : 0.f;
  const float *pb = prob + (size_t)b * O, *cb = cosv + (size_t)b * O, *ib = inv_o + (size_t)b * O;
  if ((int)blockIdx.x < G) {
    __syncthreads();
    for (int k = 0; k < 4; ++k) {
      const int o = blockIdx.x * 16 + w * 4 + k;
      if (o >= O) break;
      const float c = (pb[o] - (o == (int)lab ? 1.f : 0.f)) * gs, io = ib[o];
      const float cs = cb[o];
      const bool clamped = io >= 1.f / eps;                       // norm below eps: y = x / eps, no projection term
      const float proj = clamped ? 0.f : cs * io;
      const float4 *x4 = reinterpret_cast<const float4 *>(obj + ((size_t)b * O + o) * D);
      float4 *d4 = reinterpret_cast<float4 *>(dobj + ((size_t)b * O + o) * D);
      const float k0 = io * c;
      for (int i = lane; i < D4; i += 64) {
        const float4 x = x4[i], t = tn4[i];
        float4 r;
        r.x = k0 * (t.x - x.x * proj); r.y = k0 * (t.y - x.y * proj);
        r.z = k0 * (t.z - x.z * proj); r.w = k0 * (t.w - x.w * proj);
        d4[i] = r;
      }
    }
    return;
  }
  if (!dtext) return;
  float part = 0.f;
  for (int o = threadIdx.x; o < O; o += kBlock) {
    const float c = (pb[o] - (o == (int)lab ? 1.f : 0.f)) * gs;
    co[o] = c * ib[o];
    part += c * cb[o];
  }
  const float sum_cl = block_sum(part, red);                      // also orders the co[] writes before the reads below
  const bool clamped = it >= 1.f / eps;
  const float proj = clamped ? 0.f : sum_cl;
  float4 *d4 = reinterpret_cast<float4 *>(dtext + (size_t)b * D);
  for (int i = threadIdx.x; i < D4; i += kBlock) {
    float4 acc = {0.f, 0.f, 0.f, 0.f};
    const float4 *x4 = reinterpret_cast<const float4 *>(obj + (size_t)b * O * D) + i;
    for (int o0 = 0; o0 < O; o0 += 8) {                            // 8 rows requested together, added in object order
      float4 x[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) x[u] = x4[(size_t)min(o0 + u, O - 1) * D4];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const float c = (o0 + u < O) ? co[min(o0 + u, O - 1)] : 0.f;
        acc.x += c * x[u].x; acc.y += c * x[u].y; acc.z += c * x[u].z; acc.w += c * x[u].w;
      }
    }
    const float4 t = tn4[i];
    float4 r;
    r.x = it * (acc.x - t.x * proj); r.y = it * (acc.y - t.y * proj);
    r.z = it * (acc.z - t.z * proj); r.w = it * (acc.w - t.w * proj);
    d4[i] = r;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// symmetric CLIP loss over n rows.  M[i][j] = a_n[i] . b_n[j]; a2b = s M, b2a = s M^T, targets = the diagonal.
//   loss = (sum_i (lse_row_i - s M_ii) + sum_j (lse_col_j - s M_jj)) / (2 n),   s = min(scale, max_scale)
// Workgroup i computes row i of M (saved: the backward pass reads it) and column i (its own dots: a_n[j] . b_n[i]).
// normalize = 0: the rows are taken as they are (inv = 1; data-parallel runs hand over gathered, already normalised rows).
// ---------------------------------------------------------------------------------------------------------------------
template <int NW>
__global__ __launch_bounds__(64 * NW) void clip_fwd_kernel(int n, int D, int normalize, const float *__restrict__ a,
                                                           const float *__restrict__ bm, const float *__restrict__ scale,
                                                           float max_scale, float eps, float *__restrict__ M,
                                                           float *__restrict__ lse_row, float *__restrict__ lse_col,
                                                           float *__restrict__ inv_a, float *__restrict__ inv_b,
                                                           float *__restrict__ loss_rows, float *__restrict__ loss,
                                                           unsigned int *__restrict__ ticket) {
  extern __shared__ __attribute__((aligned(16))) float sm[];          // D: a_n[i] | D: b_n[i] | n: row | n: col | NW
  float *an = sm, *bn = sm + D, *row = sm + 2 * D, *col = row + n, *red = col + n;
  const int i = blockIdx.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int D4 = D >> 2;
  float4 *an4 = reinterpret_cast<float4 *>(an), *bn4 = reinterpret_cast<float4 *>(bn);
  const float4 *ai4 = reinterpret_cast<const float4 *>(a + (size_t)i * D), *bi4 = reinterpret_cast<const float4 *>(bm + (size_t)i * D);
  float sa = 0.f, sb = 0.f;
  for (int k = threadIdx.x; k < D4; k += 64 * NW) {
    const float4 x = ai4[k], y = bi4[k];
    an4[k] = x; bn4[k] = y;
    sa += dot4(x, x); sb += dot4(y, y);
  }
  sa = block_sum_n<NW>(sa, red);
  sb = block_sum_n<NW>(sb, red);
  const float ia = normalize ? 1.f / fmaxf(sqrtf(sa), eps) : 1.f, ib = normalize ? 1.f / fmaxf(sqrtf(sb), eps) : 1.f;
  for (int k = threadIdx.x; k < D4; k += 64 * NW) {
    float4 x = an4[k], y = bn4[k];
    x.x *= ia; x.y *= ia; x.z *= ia; x.w *= ia;
    y.x *= ib; y.y *= ib; y.z *= ib; y.w *= ib;
    an4[k] = x; bn4[k] = y;
  }
  __syncthreads();
  for (int j0 = w; j0 < n; j0 += 2 * NW) {                     // two rows of each operand per wave and trip
    const float4 *aj4[2], *bj4[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int j = min(j0 + u * NW, n - 1);
      aj4[u] = reinterpret_cast<const float4 *>(a + (size_t)j * D);
      bj4[u] = reinterpret_cast<const float4 *>(bm + (size_t)j * D);
    }
    float d_row[2] = {0.f, 0.f}, s_b[2] = {0.f, 0.f}, d_col[2] = {0.f, 0.f}, s_a[2] = {0.f, 0.f};
    for (int k = lane; k < D4; k += 64) {
      float4 x[2], y[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) { y[u] = bj4[u][k]; x[u] = aj4[u][k]; }
      const float4 an_k = an4[k], bn_k = bn4[k];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        d_row[u] += dot4(an_k, y[u]); s_b[u] += dot4(y[u], y[u]);
        d_col[u] += dot4(bn_k, x[u]); s_a[u] += dot4(x[u], x[u]);
      }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int j = j0 + u * NW;
      const float dr = wave_sum(d_row[u]), sb2 = wave_sum(s_b[u]), dc = wave_sum(d_col[u]), sa2 = wave_sum(s_a[u]);
      if (lane == 0 && j < n) {
        const float jb = normalize ? 1.f / fmaxf(sqrtf(sb2), eps) : 1.f, ja = normalize ? 1.f / fmaxf(sqrtf(sa2), eps) : 1.f;
        row[j] = dr * jb;
        col[j] = dc * ja;
        M[(size_t)i * n + j] = dr * jb;
      }
    }
  }
  __syncthreads();
  if (w != 0) return;
  const float s = fminf(scale[0], max_scale);
  float mr = -INFINITY, mc = -INFINITY;
  for (int j = lane; j < n; j += 64) { mr = fmaxf(mr, s * row[j]); mc = fmaxf(mc, s * col[j]); }
  mr = wave_max(mr); mc = wave_max(mc);
  float er = 0.f, ec = 0.f;
  for (int j = lane; j < n; j += 64) { er += __expf(s * row[j] - mr); ec += __expf(s * col[j] - mc); }
  er = wave_sum(er); ec = wave_sum(ec);
  int last = 0;
  if (lane == 0) {
    const float lr = mr + __logf(er), lc = mc + __logf(ec);
    lse_row[i] = lr;
    lse_col[i] = lc;
    inv_a[i] = ia;
    inv_b[i] = ib;
    const float mine = (lr - s * row[i]) + (lc - s * col[i]);
    last = publish_and_last(loss_rows, i, mine, ticket, n) ? 1 : 0;
  }
  if (!__shfl(last, 0, 64)) return;
  float total = 0.f;                                              // the mean, by the whole wave of the last arrival
  for (int k = lane; k < n; k += 64) total += peek(loss_rows + k);
  total = wave_sum(total);
  if (lane == 0) loss[0] = total / (2.f * (float)n);
}

// Workgroup i: dM[i][j] (row) and dM[j][i] (column) with
//   dM[i][j] = (exp(s M_ij - lse_row_i) + exp(s M_ij - lse_col_j) - 2 [i == j]) g / (2 n)
//   dscale   = sum_ij dM_ij M_ij  (workgroup partials summed in row order by the last arrival; 0 when the clamp is active)
//   da_n[i]  = s sum_j dM_ij b_n[j],  db_n[i] = s sum_j dM_ji a_n[j],  then through the normalisation (see the file header)
__global__ __launch_bounds__(kBlock) void clip_bwd_kernel(int n, int D, int normalize, int need_feats,
                                                           const float *__restrict__ a, const float *__restrict__ bm,
                                                           const float *__restrict__ scale, float max_scale, float eps,
                                                           const float *__restrict__ M, const float *__restrict__ lse_row,
                                                           const float *__restrict__ lse_col, const float *__restrict__ inv_a,
                                                           const float *__restrict__ inv_b, const float *__restrict__ gout,
                                                           float *__restrict__ da, float *__restrict__ db,
                                                           float *__restrict__ dscale, float *__restrict__ ds_rows,
                                                           unsigned int *__restrict__ ticket) {
  extern __shared__ __attribute__((aligned(16))) float sm[];          // n: dM row x inv_b | n: dM col x inv_a | kWaves
  float *cr = sm, *cc = sm + n, *red = sm + 2 * n;
  const int i = blockIdx.x;
  const int D4 = D >> 2;
  const float raw = scale[0];
  const float s = fminf(raw, max_scale);
  const float gs = gout[0] / (2.f * (float)n);
  const float lri = lse_row[i], lci = lse_col[i];
  float p_row = 0.f, p_col = 0.f;
  for (int j = threadIdx.x; j < n; j += kBlock) {
    const float mij = M[(size_t)i * n + j], mji = M[(size_t)j * n + i];
    const float diag = (i == j) ? 2.f : 0.f;
    const float dr = (__expf(s * mij - lri) + __expf(s * mij - lse_col[j]) - diag) * gs;
    const float dc = (__expf(s * mji - lse_row[j]) + __expf(s * mji - lci) - diag) * gs;
    cr[j] = dr * inv_b[j];
    cc[j] = dc * inv_a[j];
    p_row += dr * mij;
    p_col += dc * mji;
  }
  p_row = block_sum(p_row, red);                                  // = a_n[i] . da_n[i] / s; orders cr / cc too
  p_col = block_sum(p_col, red);
  if (threadIdx.x < 64) {                                          // wave 0: the last arrival sums the row partials
    int last = 0;
    if (threadIdx.x == 0) last = publish_and_last(ds_rows, i, p_row, ticket, n) ? 1 : 0;
    if (__shfl(last, 0, 64)) {
      float total = 0.f;
      for (int k = threadIdx.x; k < n; k += 64) total += peek(ds_rows + k);
      total = wave_sum(total);
      if (threadIdx.x == 0) dscale[0] = raw <= max_scale ? total : 0.f;   // torch.clamp(max=): gradient where raw <= max
    }
  }
  if (!need_feats) return;
  const float ia = inv_a[i], ib = inv_b[i];
  const bool clamp_a = normalize && ia >= 1.f / eps, clamp_b = normalize && ib >= 1.f / eps;
  const float proj_a = (normalize && !clamp_a) ? p_row : 0.f, proj_b = (normalize && !clamp_b) ? p_col : 0.f;
  for (int k = threadIdx.x; k < D4; k += kBlock) {
    float4 ua = {0.f, 0.f, 0.f, 0.f}, ub = {0.f, 0.f, 0.f, 0.f};
    const float4 *b4 = reinterpret_cast<const float4 *>(bm) + k, *a4 = reinterpret_cast<const float4 *>(a) + k;
    for (int j0 = 0; j0 < n; j0 += 8) {                            // 8 rows of each operand requested together, added in row order
      float4 y[8], x[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const size_t j = (size_t)min(j0 + u, n - 1);
        y[u] = b4[j * D4];
        x[u] = a4[j * D4];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const bool in = j0 + u < n;
        const float r = in ? cr[min(j0 + u, n - 1)] : 0.f, c = in ? cc[min(j0 + u, n - 1)] : 0.f;
        ua.x += r * y[u].x; ua.y += r * y[u].y; ua.z += r * y[u].z; ua.w += r * y[u].w;
        ub.x += c * x[u].x; ub.y += c * x[u].y; ub.z += c * x[u].z; ub.w += c * x[u].w;
      }
    }
    const float4 xi = a4[(size_t)i * D4], yi = b4[(size_t)i * D4];
    float4 ra, rb;
    // dx = inv (dn - n (n . dn)) with n = x inv, dn = s u, n . dn = s p:  dx = s inv (u - x inv p)
    ra.x = s * ia * (ua.x - xi.x * ia * proj_a); ra.y = s * ia * (ua.y - xi.y * ia * proj_a);
    ra.z = s * ia * (ua.z - xi.z * ia * proj_a); ra.w = s * ia * (ua.w - xi.w * ia * proj_a);
    rb.x = s * ib * (ub.x - yi.x * ib * proj_b); rb.y = s * ib * (ub.y - yi.y * ib * proj_b);
    rb.z = s * ib * (ub.z - yi.z * ib * proj_b); rb.w = s * ib * (ub.w - yi.w * ib * proj_b);
    reinterpret_cast<float4 *>(da + (size_t)i * D)[k] = ra;
    reinterpret_cast<float4 *>(db + (size_t)i * D)[k] = rb;
  }
}

}  // namespace gps_contra

static inline int launch_status() { return hipGetLastError() == hipSuccess ? GPS_OK : GPS_ERR_LAUNCH; }
static inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

extern "C" {

int gps_text_obj_ce_forward(int B, int O, int D, const float *obj, const float *text, const long long *labels,
                            const unsigned char *masks, float eps, long long ignore_index, float *cosv, float *prob,
                            float *inv_o, float *inv_t, float *loss_rows, float *scal, unsigned int *ticket,
                            gps_stream_t stream) {
  if (B < 0 || O < 1 || D < 4) return GPS_ERR_INVALID_ARGUMENT;
  if (B == 0) return GPS_OK;
  if (!obj || !text || !labels || !masks || !cosv || !prob || !inv_o || !inv_t || !loss_rows || !scal || !ticket)
    return GPS_ERR_INVALID_ARGUMENT;
  if ((D & 3) || D > 8192 || O > 4096 || !aligned16(obj) || !aligned16(text)) return GPS_ERR_UNSUPPORTED;
  const size_t lds = (size_t)(D + O + gps_contra::kFwdWaves) * sizeof(float);
  if (lds > 64 * 1024) return GPS_ERR_UNSUPPORTED;      // (no MaxDynamicSharedMemorySize grant: the default limit; callers fall back)
  hipLaunchKernelGGL(gps_contra::text_obj_fwd_kernel<gps_contra::kFwdWaves>, dim3(B), dim3(64 * gps_contra::kFwdWaves), lds, (hipStream_t)stream, B, O, D,
                     obj, text, (const int64_t *)labels, masks, eps, ignore_index, cosv, prob, inv_o, inv_t, loss_rows, scal,
                     ticket);
  return launch_status();
}

int gps_text_obj_ce_backward(int B, int O, int D, const float *obj, const float *text, const long long *labels, float eps,
                             long long ignore_index, const float *cosv, const float *prob, const float *inv_o,
                             const float *inv_t, const float *scal, const float *grad_out, float *dobj, float *dtext,
                             gps_stream_t stream) {
  if (B < 0 || O < 1 || D < 4) return GPS_ERR_INVALID_ARGUMENT;
  if (B == 0 || (!dobj && !dtext)) return GPS_OK;
  if (!obj || !text || !labels || !cosv || !prob || !inv_o || !inv_t || !scal || !grad_out) return GPS_ERR_INVALID_ARGUMENT;
  if ((D & 3) || D > 8192 || O > 4096 || !aligned16(obj) || !aligned16(text) || (dobj && !aligned16(dobj)) ||
      (dtext && !aligned16(dtext)))
    return GPS_ERR_UNSUPPORTED;
  const int G = dobj ? (O + 15) / 16 : 0;
  const size_t lds = (size_t)(D + O + gps_contra::kWaves) * sizeof(float);
  if (lds > 64 * 1024) return GPS_ERR_UNSUPPORTED;      // (no MaxDynamicSharedMemorySize grant: the default limit; callers fall back)
  hipLaunchKernelGGL(gps_contra::text_obj_bwd_kernel, dim3(G + 1, B), dim3(gps_contra::kBlock), lds, (hipStream_t)stream, B,
                     O, D, G, obj, text, (const int64_t *)labels, eps, ignore_index, cosv, prob, inv_o, inv_t, scal, grad_out,
                     dobj, dtext);
  return launch_status();
}

int gps_clip_loss_forward(int n, int D, int normalize, const float *a, const float *b, const float *scale, float max_scale,
                          float eps, float *M, float *lse_row, float *lse_col, float *inv_a, float *inv_b, float *loss_rows,
                          float *loss, unsigned int *ticket, gps_stream_t stream) {
  if (n < 1 || D < 4) return GPS_ERR_INVALID_ARGUMENT;
  if (!a || !b || !scale || !M || !lse_row || !lse_col || !inv_a || !inv_b || !loss_rows || !loss || !ticket)
    return GPS_ERR_INVALID_ARGUMENT;
  if ((D & 3) || D > 8192 || n > 8192 || !aligned16(a) || !aligned16(b)) return GPS_ERR_UNSUPPORTED;
  const size_t lds = (size_t)(2 * D + 2 * n + gps_contra::kFwdWaves) * sizeof(float);
  if (lds > 64 * 1024) return GPS_ERR_UNSUPPORTED;      // (no MaxDynamicSharedMemorySize grant: the default limit; callers fall back)
  hipLaunchKernelGGL(gps_contra::clip_fwd_kernel<gps_contra::kFwdWaves>, dim3(n), dim3(64 * gps_contra::kFwdWaves), lds, (hipStream_t)stream, n, D, normalize,
                     a, b, scale, max_scale, eps, M, lse_row, lse_col, inv_a, inv_b, loss_rows, loss, ticket);
  return launch_status();
}

int gps_clip_loss_backward(int n, int D, int normalize, const float *a, const float *b, const float *scale, float max_scale,
                           float eps, const float *M, const float *lse_row, const float *lse_col, const float *inv_a,
                           const float *inv_b, const float *grad_out, float *da, float *db, float *dscale, float *ds_rows,
                           unsigned int *ticket, gps_stream_t stream) {
  if (n < 1 || D < 4) return GPS_ERR_INVALID_ARGUMENT;
  if (!a || !b || !scale || !M || !lse_row || !lse_col || !inv_a || !inv_b || !grad_out || !dscale || !ds_rows || !ticket)
    return GPS_ERR_INVALID_ARGUMENT;
  if ((da == nullptr) != (db == nullptr)) return GPS_ERR_INVALID_ARGUMENT;
  if ((D & 3) || D > 8192 || n > 8192 || !aligned16(a) || !aligned16(b) || (da && (!aligned16(da) || !aligned16(db))))
    return GPS_ERR_UNSUPPORTED;
  const size_t lds = (size_t)(2 * n + gps_contra::kWaves) * sizeof(float);
  if (lds > 64 * 1024) return GPS_ERR_UNSUPPORTED;      // (no MaxDynamicSharedMemorySize grant: the default limit; callers fall back)
  hipLaunchKernelGGL(gps_contra::clip_bwd_kernel, dim3(n), dim3(gps_contra::kBlock), lds, (hipStream_t)stream, n, D, normalize,
                     da ? 1 : 0, a, b, scale, max_scale, eps, M, lse_row, lse_col, inv_a, inv_b, grad_out, da, db, dscale,
                     ds_rows, ticket);
  return launch_status();
}

}  // extern "C"
