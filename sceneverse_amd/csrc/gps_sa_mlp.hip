// gps_sa_mlp.hip -- fused set-abstraction level for a FROZEN PointNet++ encoder on MI355X (gfx950).
//
// One launch replaces, for one SA level of the reference
//   modules/third_party/pointnet2/pointnet2_utils.py:345-356   group xyz, subtract centre, group feats, cat
//   modules/third_party/pointnet2/pytorch_utils.py:11-36       SharedMLP = 3 x (conv1x1 no-bias, BN, ReLU)
//   modules/third_party/pointnet2/pointnet2_modules.py:65-71   max_pool2d over nsample
// i.e. 2 group_points launches, `-=`, `cat`, 3 x (GEMM, batch-norm, ReLU) and a max-pool -- about
// half of the GPS training step in the first rocprof trace (profiles/r1/bench_b_kernel_stats.csv).
// In 35 of the reference's 37 configs the encoder is frozen: BN runs on its running statistics and
// nothing needs a gradient, so conv+BN folds into W' = diag(gamma/sqrt(var+eps)) W and a shift.
//
// Design (CDNA4):
//   * one 256-thread workgroup per object; the object's points/features/indices are staged once in
//     LDS (coalesced 16-byte loads), so the grouped (3+C, npoint, nsample) tensor -- 268 KB per
//     object at SA2, the largest HBM stream of the point path -- never exists in HBM;
//   * a wave owns one group = one 32-column tile (nsample == 32) through ALL three layers.  Layers
//     run on v_mfma_f32_32x32x2_f32 (fp32 in, fp32 accumulate: bitwise an fmaf chain, so results
//     stay within fp32 summation-order noise of the reference's fp32 conv).  The D fragment of one
//     layer (lane = column, registers = rows) IS the B fragment of the next layer's MFMAs once the
//     K index is permuted to the D row order -- the permutation is applied to the packed weights
//     instead, so activations never leave the VGPRs between layers;
//   * weights stream through LDS one 32-row output tile at a time (double buffered, one barrier per
//     tile); folded BN shift initialises the accumulator, ReLU is one v_max per register;
//   * max over the 32 samples = DPP row reduction + row_bcast15; pooled rows are collected in LDS
//     and written with coalesced stores.
//
// K-slot order.  For v_mfma_f32_32x32x2_f32 lane l holds A[i = l&31][k = l>>5], B[k = l>>5][j = l&31]
// and D[row = (r&3) + 8*(r>>2) + 4*(l>>5)][col = l&31] in register r.  Input channels are therefore
// consumed in "slots" (it, r): the step for slot (it, r) multiplies channel it*32 + (r&3) + 8*(r>>2)
// (lanes 0-31) and that + 4 (lanes 32-63).  gps_sa_mlp_pack_layer() writes the weights in that
// order, zero where the channel is >= c_in.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gps_hip.h"
#include "gps_device_flags.h"

namespace gps { const int *object_extent(); }   // gps_point_ops.hip: device int or null (gps_point_set_object_extent)

#ifndef GPS_SA1_WAVES
#define GPS_SA1_WAVES 8   // weights resident in LDS: waves only share the object
#endif
#ifndef GPS_SA2_WAVES
#define GPS_SA2_WAVES 8   // streamed weights: 8 groups per weight tile
#endif

namespace gps_sa {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kBlock = 256;
constexpr int kWaves = 4;
constexpr int kNS = 32;  // samples per group == MFMA N

__host__ __device__ constexpr int slot_channel(int it, int r, int h) {
  return it * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
}
// number of K steps of a layer with c_in input channels (slots whose lower channel exists)
__host__ __device__ constexpr int layer_steps(int c_in) {
  int s = 0;
  for (int it = 0; it < (c_in + 31) / 32; ++it)
    for (int r = 0; r < 16; ++r)
      if (slot_channel(it, r, 0) < c_in) ++s;
  return s;
}
// floats of one packed 32-row output tile: steps x 64 weights + 32 shifts, padded to whole
// 1 KiB pieces (one global_load_lds_dwordx4 wave-instruction each)
__host__ __device__ constexpr int tile_floats(int c_in) {
  return (layer_steps(c_in) * 64 + 32 + 255) / 256 * 256;
}

// ------------------------------------------------------------------------------------------
// weight packing: w (c_out, c_in) row-major (already BN-folded), shift (c_out) ->
//   dst[mt][step][h*32 + i] = w[mt*32 + i][channel(step, h)]   (0 beyond c_in)
//   dst[mt][steps*64 + i]   = shift[mt*32 + i]
// ------------------------------------------------------------------------------------------
__global__ void pack_layer_kernel(int c_in, int c_out, const float *__restrict__ w,
                                  const float *__restrict__ shift, float *__restrict__ dst) {
  const int steps = layer_steps(c_in);
  const int tf = tile_floats(c_in);
  const int total = (c_out / 32) * tf;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
    const int mt = e / tf, o = e - mt * tf;
    if (o >= steps * 64) {
      dst[e] = o < steps * 64 + 32 ? shift[mt * 32 + (o - steps * 64)] : 0.f;
      continue;
    }
    const int s = o >> 6, h = (o >> 5) & 1, i = o & 31;
    // s-th valid slot
    int it = 0, r = 0, cnt = 0;
    for (int a = 0; a < (c_in + 31) / 32; ++a)
      for (int q = 0; q < 16; ++q)
        if (slot_channel(a, q, 0) < c_in) {
          if (cnt == s) { it = a; r = q; }
          ++cnt;
        }
    const int k = slot_channel(it, r, h);
    dst[e] = k < c_in ? w[(size_t)(mt * 32 + i) * c_in + k] : 0.f;
  }
}

// ------------------------------------------------------------------------------------------
// helpers
// ------------------------------------------------------------------------------------------
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_max_f32(float v) {
  const int o = __builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, ROW_MASK,
                                            0xf, false);
  return fmaxf(v, __int_as_float(o));
}
// max over lanes 0-31 -> lane 31, over lanes 32-63 -> lane 63
__device__ __forceinline__ float half_wave_max(float v) {
  v = dpp_max_f32<0x111, 0xf>(v);  // row_shr:1
  v = dpp_max_f32<0x112, 0xf>(v);  // row_shr:2
  v = dpp_max_f32<0x114, 0xf>(v);  // row_shr:4
  v = dpp_max_f32<0x118, 0xf>(v);  // row_shr:8  -> lane 15 of each 16-lane row = row max
  v = dpp_max_f32<0x142, 0xa>(v);  // row_bcast:15 into rows 1 and 3 -> lanes 31 / 63
  return v;
}

// One 32-row output tile: acc = shift; acc += sum_s A[s] (LDS) x B[s] (registers).
template <int STEPS>
__device__ __forceinline__ f32x16 mfma_tile(const float *__restrict__ tile, const float (&B)[STEPS],
                                            int lane) {
  f32x16 acc;
  const float *sh = tile + STEPS * 64 + 4 * (lane >> 5);
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = sh[(r & 3) + 8 * (r >> 2)];
#pragma unroll
  for (int s = 0; s < STEPS; ++s)
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(tile[s * 64 + lane], B[s], acc, 0, 0, 0);
  return acc;
}

// Cooperative asynchronous copy of one packed tile (len floats, a multiple of 256) from global
// memory straight into LDS: each wave-instruction moves 1 KiB (64 lanes x 16 B) to
// wave-uniform base + lane * 16, no VGPRs held while the MFMAs run.  Completion = the issuing
// wave's vmcnt(0), then the workgroup barrier.
template <int NWAVES = kWaves>
__device__ __forceinline__ void tile_copy_async(const float *__restrict__ src, float *dst, int len,
                                                int wave, int lane) {
  for (int c = wave; c * 256 < len; c += NWAVES)
    __builtin_amdgcn_global_load_lds(
        (const __attribute__((address_space(1))) void *)(src + c * 256 + lane * 4),
        (__attribute__((address_space(3))) void *)(dst + c * 256), 16, 0, 0);
}
__device__ __forceinline__ void tile_copy_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// ------------------------------------------------------------------------------------------
// fused SA level: ball-query indices in, pooled features out.
//   xyz (b,n,3), new_xyz (b,npoint,3), feats (b,CF,n), idx (b,npoint,32) -> out (b,C3,npoint)
//   wpack = [layer 1 tiles | layer 2 tiles | layer 3 tiles]
// ------------------------------------------------------------------------------------------
template <int CF, int C1, int C2, int C3>
__global__ __launch_bounds__(kBlock, 2) void sa_mlp_kernel(
    int b, int n, int npoint, const float *__restrict__ xyz, const float *__restrict__ new_xyz,
    const float *__restrict__ feats, const int32_t *__restrict__ idx,
    const float *__restrict__ wpack, float *__restrict__ out, const int *__restrict__ n_obj_dev) {
  if (n_obj_dev && (int)blockIdx.x >= *n_obj_dev) return;      // object extent: nothing read or written
  constexpr int CIN = 3 + CF;
  constexpr int S1 = layer_steps(CIN), S2 = C1 / 2, S3 = C2 / 2;
  constexpr int T1 = tile_floats(CIN), T2 = tile_floats(C1), T3 = tile_floats(C2);
  constexpr int M1 = C1 / 32, M2 = C2 / 32, M3 = C3 / 32;
  constexpr int TMAX = T1 > T2 ? (T1 > T3 ? T1 : T3) : (T2 > T3 ? T2 : T3);
  constexpr int G = M1 + M2 + M3;            // weight tiles (stages) per round

  extern __shared__ __attribute__((aligned(16))) float lds[];
  float *s_w0 = lds;                         // TMAX
  float *s_w1 = s_w0 + TMAX;                 // TMAX
  float *s_out = s_w1 + TMAX;                // C3 * npoint
  float *s_feat = s_out + C3 * npoint;       // CF * n   (channel-major, as in HBM)
  float *s_xyz = s_feat + CF * n;            // n * 3
  float *s_ctr = s_xyz + n * 3;              // npoint * 3
  int32_t *s_idx = reinterpret_cast<int32_t *>(s_ctr + npoint * 3);  // npoint * 32

  const int obj = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int col = lane & 31, h = lane >> 5;

  // ---- stage the object ------------------------------------------------------------------
  {
    const float *gf = feats + (size_t)obj * CF * n;
    const float *gx = xyz + (size_t)obj * n * 3;
    const float *gc = new_xyz + (size_t)obj * npoint * 3;
    const int32_t *gi = idx + (size_t)obj * npoint * kNS;
    if (((CF * n) & 3) == 0) {
      const float4 *g4 = reinterpret_cast<const float4 *>(gf);
      float4 *l4 = reinterpret_cast<float4 *>(s_feat);
      for (int e = tid; e < (CF * n) >> 2; e += kBlock) l4[e] = g4[e];
    } else {
      for (int e = tid; e < CF * n; e += kBlock) s_feat[e] = gf[e];
    }
    for (int e = tid; e < n * 3; e += kBlock) s_xyz[e] = gx[e];
    for (int e = tid; e < npoint * 3; e += kBlock) s_ctr[e] = gc[e];
    for (int e = tid; e < npoint * kNS; e += kBlock) s_idx[e] = gi[e];
  }
  tile_copy_async(wpack, s_w0, T1, wave, lane);
  tile_copy_wait();
  __syncthreads();

  auto tile_src = [&](int g, int &len) -> const float * {  // packed tile of stage g (mod G)
    if (g >= G) g -= G;
    if (g < M1) { len = T1; return wpack + (size_t)g * T1; }
    if (g < M1 + M2) { len = T2; return wpack + (size_t)M1 * T1 + (size_t)(g - M1) * T2; }
    len = T3;
    return wpack + (size_t)M1 * T1 + (size_t)M2 * T2 + (size_t)(g - M1 - M2) * T3;
  };

  const int rounds = (npoint + kWaves - 1) / kWaves;
  for (int rd = 0; rd < rounds; ++rd) {
    const int tile = rd * kWaves + wave;
    const bool live = tile < npoint;
    const int j = live ? tile : npoint - 1;

    // ---- layer-1 B operand: X0[channel][sample] gathered from LDS ------------------------
    float a0[S1];
    {
      const int p = s_idx[j * kNS + col];
      int s = 0;
#pragma unroll
      for (int it = 0; it < (CIN + 31) / 32; ++it)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (slot_channel(it, r, 0) < CIN) {
            const int k = slot_channel(it, r, h);          // this lane's channel of the slot
            float v = 0.f;
            if (slot_channel(it, r, 1) < 3) {              // both halves are xyz rows
              v = s_xyz[p * 3 + k] - s_ctr[j * 3 + k];
            } else if (slot_channel(it, r, 0) >= 3 && slot_channel(it, r, 1) < CIN) {
              v = s_feat[(k - 3) * n + p];                 // both halves are feature rows
            } else {                                       // mixed slot: decide per lane
              if (k < 3) v = s_xyz[p * 3 + k] - s_ctr[j * 3 + k];
              else if (k < CIN) v = s_feat[(k - 3) * n + p];
            }
            a0[s++] = v;
          }
    }

    float a1[M1 * 16], a2[M2 * 16];
    // g = stage within the round; gg = running stage count.  Tile gg sits in buffer gg & 1 and the
    // next tile is prefetched into the other one (G may be odd, so the parity carries over rounds).
    int g = 0;
    // ---- layer 1 ----------------------------------------------------------------------------
#pragma unroll
    for (int mt = 0; mt < M1; ++mt, ++g) {
      const int gg = rd * G + g;
      int len;
      const float *src = tile_src(g + 1, len);
      tile_copy_async(src, (gg & 1) ? s_w0 : s_w1, len, wave, lane);
      const f32x16 acc = mfma_tile<S1>((gg & 1) ? s_w1 : s_w0, a0, lane);
#pragma unroll
      for (int r = 0; r < 16; ++r) a1[mt * 16 + r] = fmaxf(acc[r], 0.f);
      tile_copy_wait();
      __syncthreads();
    }
    // ---- layer 2 ----------------------------------------------------------------------------
#pragma unroll
    for (int mt = 0; mt < M2; ++mt, ++g) {
      const int gg = rd * G + g;
      int len;
      const float *src = tile_src(g + 1, len);
      tile_copy_async(src, (gg & 1) ? s_w0 : s_w1, len, wave, lane);
      const f32x16 acc = mfma_tile<S2>((gg & 1) ? s_w1 : s_w0, a1, lane);
#pragma unroll
      for (int r = 0; r < 16; ++r) a2[mt * 16 + r] = fmaxf(acc[r], 0.f);
      tile_copy_wait();
      __syncthreads();
    }
    // ---- layer 3 + max over the 32 samples ---------------------------------------------------
    for (int mt = 0; mt < M3; ++mt, ++g) {
      const int gg = rd * G + g;
      int len;
      const float *src = tile_src(g + 1, len);
      tile_copy_async(src, (gg & 1) ? s_w0 : s_w1, len, wave, lane);
      const f32x16 acc = mfma_tile<S3>((gg & 1) ? s_w1 : s_w0, a2, lane);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float m = half_wave_max(acc[r]);
        if (col == 31 && live)
          s_out[(mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * npoint + tile] = fmaxf(m, 0.f);
      }
      tile_copy_wait();
      __syncthreads();
    }
  }
  // ---- pooled features out, coalesced -----------------------------------------------------------
  float *go = out + (size_t)obj * C3 * npoint;
  const int total = C3 * npoint;
  if ((total & 3) == 0) {
    const float4 *l4 = reinterpret_cast<const float4 *>(s_out);
    float4 *g4 = reinterpret_cast<float4 *>(go);
    for (int e = tid; e < total >> 2; e += kBlock) g4[e] = l4[e];
  } else {
    for (int e = tid; e < total; e += kBlock) go[e] = s_out[e];
  }
}

template <int CF, int C1, int C2, int C3>
size_t sa_lds_bytes(int n, int npoint) {
  constexpr int CIN = 3 + CF;
  constexpr int T1 = tile_floats(CIN), T2 = tile_floats(C1), T3 = tile_floats(C2);
  constexpr int TMAX = T1 > T2 ? (T1 > T3 ? T1 : T3) : (T2 > T3 ? T2 : T3);
  return sizeof(float) * ((size_t)2 * TMAX + (size_t)C3 * npoint + (size_t)CF * n + (size_t)n * 3 +
                          (size_t)npoint * 3 + (size_t)npoint * kNS);
}

template <int CF, int C1, int C2, int C3>
int launch_sa(int b, int n, int npoint, const float *xyz, const float *new_xyz, const float *feats,
              const int32_t *idx, const float *wpack, float *out, hipStream_t s) {
  const size_t lds = sa_lds_bytes<CF, C1, C2, C3>(n, npoint);
  if (lds > 80 * 1024) return GPS_ERR_UNSUPPORTED;   // keep two workgroups per CU
  static gps_dev::PerDevice<bool, 1> attr_dev;          // dynamic LDS above 64 KiB needs opting in, per device
  bool &attr_set = attr_dev.row()[0];
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(&sa_mlp_kernel<CF, C1, C2, C3>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024) != hipSuccess)
      return GPS_ERR_LAUNCH;
    attr_set = true;
  }
  hipLaunchKernelGGL((sa_mlp_kernel<CF, C1, C2, C3>), dim3(b), dim3(kBlock), lds, s, b, n, npoint, xyz,
                     new_xyz, feats, idx, wpack, out, gps::object_extent());
  return GPS_OK;
}


// ==========================================================================================
// split-bf16 variant: every fp32 operand x is carried as (hi, lo) = (bf16(x), bf16(x - hi)) and each
// product as  W_hi X_hi + W_hi X_lo + W_lo X_hi  on v_mfma_f32_32x32x16_bf16 with fp32 accumulation:
// 3 MFMAs at the bf16 rate (16x the fp32 MFMA rate) instead of 8, i.e. ~5.3x fewer matrix-pipe
// cycles, for a relative error ~2^-16 per product (the dropped W_lo X_lo term and the 2^-17
// representation residuals) -- inside the 1e-4 tolerance of the parity tests.
//
// Same chaining trick: D regs [8u, 8u+8) of a 32-row tile are the B fragment (K = 16) of step
// (tile, u) of the next layer; K-slot (half h, element e) <-> channel 32 it + 16 u + 8 (e>>2) + 4 h + (e&3).
// Packed tile (bytes): step s: [hi: 64 lanes x 16 B][lo: 64 lanes x 16 B]; then 32 fp32 shifts;
// padded to 1 KiB.
// ==========================================================================================
namespace x3 {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x8 __attribute__((ext_vector_type(8)));

__host__ __device__ constexpr int steps16(int c_in) { return (c_in + 15) / 16; }
__host__ __device__ constexpr int slot_channel16(int it_u, int e, int h) {   // it_u = 2 * it + u
  return 16 * it_u + 8 * (e >> 2) + 4 * h + (e & 3);
}
// floats (4-byte units) of one packed 32-row tile
__host__ __device__ constexpr int tile_floats16(int c_in) {
  return (steps16(c_in) * 512 + 32 + 255) / 256 * 256;
}

__device__ __forceinline__ uint16_t f2bf(float f) {
  unsigned int u = __float_as_uint(f);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}

__global__ void pack_layer16_kernel(int c_in, int c_out, const float *__restrict__ w,
                                    const float *__restrict__ shift, float *__restrict__ dst) {
  const int steps = steps16(c_in);
  const int tf = tile_floats16(c_in);
  const int total = (c_out / 32) * tf;
  for (int e4 = blockIdx.x * blockDim.x + threadIdx.x; e4 < total; e4 += gridDim.x * blockDim.x) {
    const int mt = e4 / tf, o = e4 - mt * tf;   // o: 4-byte unit inside the tile
    if (o >= steps * 512) {
      dst[e4] = o < steps * 512 + 32 ? shift[mt * 32 + (o - steps * 512)] : 0.f;
      continue;
    }
    const int s = o >> 9, part = (o >> 8) & 1, lane = (o >> 2) & 63, pair = o & 3;   // 2 bf16 per unit
    const int i = lane & 31, h = lane >> 5;
    unsigned int packed = 0;
    for (int q = 0; q < 2; ++q) {
      const int e = 2 * pair + q;
      const int k = slot_channel16(s, e, h);
      const float x = k < c_in ? w[(size_t)(mt * 32 + i) * c_in + k] : 0.f;
      const uint16_t hi = f2bf(x);
      const float back = __uint_as_float((unsigned int)hi << 16);
      const uint16_t v = part == 0 ? hi : f2bf(x - back);
      packed |= (unsigned int)v << (16 * q);
    }
    dst[e4] = __uint_as_float(packed);
  }
}

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
// two fp32 -> one dword of two bf16 (round to nearest even); hipcc's vector convert emits one
// single-element v_cvt_pk per value plus shifts/ors, so the instruction is named explicitly
__device__ __forceinline__ unsigned int cvt_pk_bf16(float a, float b) {
  unsigned int r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// max(x, 0) as ONE compiler-visible instruction: on the raw bits, signed-integer max with 0 maps
// every negative float (and -0) to +0 and keeps every non-negative one (v_max_i32).  fmaxf() would
// cost a canonicalising v_max first; an inline-asm v_max_f32 must NOT be used here: hipcc's hazard
// recogniser does not see inline-asm readers of MFMA results and would not pad the XDL-write ->
// VALU-read wait states (observed: stale accumulators).
__device__ __forceinline__ float relu1(float x) {
  const int i = __float_as_int(x);
  return __int_as_float(i > 0 ? i : 0);
}
// v -> (hi, lo) bf16 fragments with v ~= hi + lo:  3 VALU per element
__device__ __forceinline__ void split8(const f32x8 &v, bf16x8 &hi, bf16x8 &lo) {
  u32x4 h, l;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    h[i] = cvt_pk_bf16(v[2 * i], v[2 * i + 1]);
    const float b0 = __uint_as_float(h[i] << 16), b1 = __uint_as_float(h[i] & 0xFFFF0000u);
    l[i] = cvt_pk_bf16(v[2 * i] - b0, v[2 * i + 1] - b1);
  }
  hi = __builtin_bit_cast(bf16x8, h);
  lo = __builtin_bit_cast(bf16x8, l);
}
// NPROD = 3: the three products above (fp32-accurate); NPROD = 1: W_hi X_hi only -- plain bf16 operands with fp32
// accumulation, what torch's bf16 autocast computes for the reference's Conv2d stacks (opt-in: gps_sa_mlp_set_products)
template <int STEPS, bool TRANSPOSED = false, int NPROD = 3>
__device__ __forceinline__ f32x16 mfma_tile16(const float *__restrict__ tile, const bf16x8 (&Bhi)[STEPS],
                                              const bf16x8 (&Blo)[STEPS], int lane) {
  f32x16 acc;
  if (TRANSPOSED) {
    const float shc = tile[STEPS * 512 + (lane & 31)];        // shift of this lane's output channel
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = shc;
  } else {
    const float *sh = tile + STEPS * 512 + 4 * (lane >> 5);
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = sh[(r & 3) + 8 * (r >> 2)];
  }
  const bf16x8 *frag = reinterpret_cast<const bf16x8 *>(tile) + lane;
#pragma unroll
  for (int s = 0; s < STEPS; ++s) {
    const bf16x8 ahi = frag[s * 128];
    if (NPROD == 1) {
      acc = TRANSPOSED ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(Bhi[s], ahi, acc, 0, 0, 0)
                       : __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi, Bhi[s], acc, 0, 0, 0);
      continue;
    }
    const bf16x8 alo = frag[s * 128 + 64];
    if (TRANSPOSED) {
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Bhi[s], alo, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Blo[s], ahi, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Bhi[s], ahi, acc, 0, 0, 0);
    } else {
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(alo, Bhi[s], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi, Blo[s], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi, Bhi[s], acc, 0, 0, 0);
    }
  }
  return acc;
}

// WAVES waves per workgroup (each owns one 32-sample group per round).  RESIDENT: the whole packed
// MLP fits in LDS next to the object -> loaded once per workgroup, no per-tile barriers; otherwise
// the 32-row weight tiles stream through a double buffer (one barrier per tile).  More waves per
// workgroup = more columns per streamed weight byte (the LDS-DMA weight stream, not the matrix pipe,
// bounds the streaming form: profiles/r1).
// PM: the features arrive point-major -- feats[(obj * n + p) * ld_feat + c], e.g. the rgb columns of an interleaved
// (B, N, 3 + C) cloud (pointer at column 3, ld_feat = 6) -- and are transposed into the channel-major LDS image while
// they are staged, instead of by a separate full-cloud transpose copy in HBM (63 us per step at SA1).
template <int CF, int C1, int C2, int C3, int WAVES, bool RESIDENT, bool PM = false, int NPROD = 3>
__global__ __launch_bounds__(WAVES * 64) void sa_mlp_x3_kernel(
    int b, int n, int npoint, const float *__restrict__ xyz, const float *__restrict__ new_xyz,
    const float *__restrict__ feats, const int32_t *__restrict__ idx,
    const float *__restrict__ wpack, float *__restrict__ out, int ld_feat, const int *__restrict__ n_obj_dev) {
  if (n_obj_dev && (int)blockIdx.x >= *n_obj_dev) return;      // object extent: nothing read or written
  constexpr int BLOCK = WAVES * 64;
  constexpr int CIN = 3 + CF;
  constexpr int S1 = steps16(CIN), S2 = C1 / 16, S3 = C2 / 16;
  constexpr int T1 = tile_floats16(CIN), T2 = tile_floats16(C1), T3 = tile_floats16(C2);
  constexpr int M1 = C1 / 32, M2 = C2 / 32, M3 = C3 / 32;
  constexpr int TMAX = T1 > T2 ? (T1 > T3 ? T1 : T3) : (T2 > T3 ? T2 : T3);
  constexpr int TOTAL = M1 * T1 + M2 * T2 + M3 * T3;
  constexpr int G = M1 + M2 + M3;
  constexpr int WBUF = RESIDENT ? TOTAL : 2 * TMAX;

  extern __shared__ __attribute__((aligned(16))) float lds[];
  float *s_w0 = lds;
  float *s_w1 = s_w0 + TMAX;                 // streaming form only
  float *s_out = lds + WBUF;                 // C3 * npoint
  float *s_feat = s_out + C3 * npoint;       // CF * n
  float *s_xyz = s_feat + CF * n;            // n * 3
  float *s_ctr = s_xyz + n * 3;              // npoint * 3
  int32_t *s_idx = reinterpret_cast<int32_t *>(s_ctr + npoint * 3);

  const int obj = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int col = lane & 31, h = lane >> 5;
  {
    const float *gx = xyz + (size_t)obj * n * 3;
    const float *gc = new_xyz + (size_t)obj * npoint * 3;
    const int32_t *gi = idx + (size_t)obj * npoint * kNS;
    if (PM) {
      const float *gf = feats + (size_t)obj * n * ld_feat;
      for (int e = tid; e < CF * n; e += BLOCK) {
        const int p = e / CF, c = e - p * CF;
        s_feat[c * n + p] = gf[(size_t)p * ld_feat + c];
      }
    } else {
      const float *gf = feats + (size_t)obj * CF * n;
      if (((CF * n) & 3) == 0) {
        const float4 *g4 = reinterpret_cast<const float4 *>(gf);
        float4 *l4 = reinterpret_cast<float4 *>(s_feat);
        for (int e = tid; e < (CF * n) >> 2; e += BLOCK) l4[e] = g4[e];
      } else {
        for (int e = tid; e < CF * n; e += BLOCK) s_feat[e] = gf[e];
      }
    }
    for (int e = tid; e < n * 3; e += BLOCK) s_xyz[e] = gx[e];
    for (int e = tid; e < npoint * 3; e += BLOCK) s_ctr[e] = gc[e];
    for (int e = tid; e < npoint * kNS; e += BLOCK) s_idx[e] = gi[e];
  }
  tile_copy_async<WAVES>(wpack, s_w0, RESIDENT ? TOTAL : T1, wave, lane);
  tile_copy_wait();
  __syncthreads();

  auto tile_off = [&](int g, int &len) -> int {      // offset (floats) of weight tile g (mod G)
    if (g >= G) g -= G;
    if (g < M1) { len = T1; return g * T1; }
    if (g < M1 + M2) { len = T2; return M1 * T1 + (g - M1) * T2; }
    len = T3;
    return M1 * T1 + M2 * T2 + (g - M1 - M2) * T3;
  };
  // begin stage g of round rd: returns the LDS tile to compute on; streaming form also starts the
  // copy of the next tile into the other buffer
  auto stage_begin = [&](int rd, int g) -> const float * {
    int len;
    if (RESIDENT) return s_w0 + tile_off(g, len);
    const int gg = rd * G + g;
    const int off = tile_off(g + 1, len);
    tile_copy_async<WAVES>(wpack + off, (gg & 1) ? s_w0 : s_w1, len, wave, lane);
    return (gg & 1) ? s_w1 : s_w0;
  };
  auto stage_end = [&]() {
    if (!RESIDENT) {
      tile_copy_wait();
      __syncthreads();
    }
  };

  const int rounds = (npoint + WAVES - 1) / WAVES;
  for (int rd = 0; rd < rounds; ++rd) {
    const int tile = rd * WAVES + wave;
    const bool live = tile < npoint;
    const int j = live ? tile : npoint - 1;

    bf16x8 a0h[S1], a0l[S1];
    {
      const int p = s_idx[j * kNS + col];
#pragma unroll
      for (int s = 0; s < S1; ++s) {
        f32x8 v;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int k = slot_channel16(s, e, h);
          float x;
          if (slot_channel16(s, e, 0) >= 3 && slot_channel16(s, e, 1) < CIN) {
            x = s_feat[(k - 3) * n + p];
          } else {
            x = 0.f;
            if (k < 3) x = s_xyz[p * 3 + k] - s_ctr[j * 3 + k];
            else if (k < CIN) x = s_feat[(k - 3) * n + p];
          }
          v[e] = x;
        }
        split8(v, a0h[s], a0l[s]);
      }
    }
    bf16x8 a1h[S2], a1l[S2], a2h[S3], a2l[S3];
    // Software pipeline: the post-processing (ReLU + hi/lo split, or ReLU + max-pool) of output tile
    // mt-1 is issued after the MFMAs of tile mt, in the same scheduling region, so that its VALU work
    // runs in the shadow of the matrix pipe (independent registers).  Only the last tile of a layer
    // is post-processed in the open (the next layer needs all of it).
    auto split_tile = [&](const f32x16 &acc, bf16x8 &h0, bf16x8 &l0, bf16x8 &h1, bf16x8 &l1) {
      f32x8 v0, v1;
#pragma unroll
      for (int e = 0; e < 8; ++e) { v0[e] = relu1(acc[e]); v1[e] = relu1(acc[8 + e]); }
      split8(v0, h0, l0);
      split8(v1, h1, l1);
    };
    // last layer in the transposed form: lane = output channel, registers (+ the other half of the
    // wave) = the 32 samples of the group -> max-pool = 15 in-lane max + one cross-half exchange
    auto pool_tile = [&](const f32x16 &acc, int mt) {
      float m0 = fmaxf(fmaxf(acc[0], acc[1]), fmaxf(acc[2], acc[3]));
      float m1 = fmaxf(fmaxf(acc[4], acc[5]), fmaxf(acc[6], acc[7]));
      float m2 = fmaxf(fmaxf(acc[8], acc[9]), fmaxf(acc[10], acc[11]));
      float m3 = fmaxf(fmaxf(acc[12], acc[13]), fmaxf(acc[14], acc[15]));
      float m = fmaxf(fmaxf(m0, m1), fmaxf(m2, m3));
      m = fmaxf(m, __shfl_xor(m, 32, 64));
      if (lane < 32 && live) s_out[(mt * 32 + lane) * npoint + tile] = fmaxf(m, 0.f);   // ReLU after the max
    };
    int g = 0;
    f32x16 prev;
#pragma unroll
    for (int mt = 0; mt < M1; ++mt, ++g) {
      const float *wt = stage_begin(rd, g);
      const f32x16 acc = mfma_tile16<S1, false, NPROD>(wt, a0h, a0l, lane);
      if (mt > 0) split_tile(prev, a1h[2 * mt - 2], a1l[2 * mt - 2], a1h[2 * mt - 1], a1l[2 * mt - 1]);
      prev = acc;
      stage_end();
    }
    split_tile(prev, a1h[2 * M1 - 2], a1l[2 * M1 - 2], a1h[2 * M1 - 1], a1l[2 * M1 - 1]);
#pragma unroll
    for (int mt = 0; mt < M2; ++mt, ++g) {
      const float *wt = stage_begin(rd, g);
      const f32x16 acc = mfma_tile16<S2, false, NPROD>(wt, a1h, a1l, lane);
      if (mt > 0) split_tile(prev, a2h[2 * mt - 2], a2l[2 * mt - 2], a2h[2 * mt - 1], a2l[2 * mt - 1]);
      prev = acc;
      stage_end();
    }
    split_tile(prev, a2h[2 * M2 - 2], a2l[2 * M2 - 2], a2h[2 * M2 - 1], a2l[2 * M2 - 1]);
    for (int mt = 0; mt < M3; ++mt, ++g) {
      const float *wt = stage_begin(rd, g);
      const f32x16 acc = mfma_tile16<S3, true, NPROD>(wt, a2h, a2l, lane);
      if (mt > 0) pool_tile(prev, mt - 1);
      prev = acc;
      stage_end();
    }
    pool_tile(prev, M3 - 1);
  }
  if (RESIDENT) __syncthreads();
  float *go = out + (size_t)obj * C3 * npoint;
  const int total = C3 * npoint;
  if ((total & 3) == 0) {
    const float4 *l4 = reinterpret_cast<const float4 *>(s_out);
    float4 *g4 = reinterpret_cast<float4 *>(go);
    for (int e = tid; e < total >> 2; e += BLOCK) g4[e] = l4[e];
  } else {
    for (int e = tid; e < total; e += BLOCK) go[e] = s_out[e];
  }
}

static int g_products = 3;      // 3 = split-bf16 triple product (fp32-accurate, default); 1 = single bf16 product (opt-in)

template <int CF, int C1, int C2, int C3, int WAVES, bool RESIDENT, bool PM, int NPROD>
int launch_sa_x3_n(int b, int n, int npoint, const float *xyz, const float *new_xyz, const float *feats,
                   const int32_t *idx, const float *wpack, float *out, hipStream_t s, int ld_feat);

template <int CF, int C1, int C2, int C3, int WAVES, bool RESIDENT, bool PM = false>
int launch_sa_x3(int b, int n, int npoint, const float *xyz, const float *new_xyz, const float *feats,
                 const int32_t *idx, const float *wpack, float *out, hipStream_t s, int ld_feat = 0) {
  return g_products == 1
             ? launch_sa_x3_n<CF, C1, C2, C3, WAVES, RESIDENT, PM, 1>(b, n, npoint, xyz, new_xyz, feats, idx, wpack, out, s, ld_feat)
             : launch_sa_x3_n<CF, C1, C2, C3, WAVES, RESIDENT, PM, 3>(b, n, npoint, xyz, new_xyz, feats, idx, wpack, out, s, ld_feat);
}

template <int CF, int C1, int C2, int C3, int WAVES, bool RESIDENT, bool PM, int NPROD>
int launch_sa_x3_n(int b, int n, int npoint, const float *xyz, const float *new_xyz, const float *feats,
                   const int32_t *idx, const float *wpack, float *out, hipStream_t s, int ld_feat) {
  constexpr int CIN = 3 + CF;
  constexpr int T1 = tile_floats16(CIN), T2 = tile_floats16(C1), T3 = tile_floats16(C2);
  constexpr int TMAX = T1 > T2 ? (T1 > T3 ? T1 : T3) : (T2 > T3 ? T2 : T3);
  constexpr int TOTAL = (C1 / 32) * T1 + (C2 / 32) * T2 + (C3 / 32) * T3;
  constexpr int WBUF = RESIDENT ? TOTAL : 2 * TMAX;
  const size_t lds = sizeof(float) * ((size_t)WBUF + (size_t)C3 * npoint + (size_t)CF * n + (size_t)n * 3 +
                                      (size_t)npoint * 3 + (size_t)npoint * kNS);
  if (lds > 160 * 1024) return GPS_ERR_UNSUPPORTED;
  static size_t attr_lds = 0;
  if (lds > attr_lds) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(&sa_mlp_x3_kernel<CF, C1, C2, C3, WAVES, RESIDENT, PM, NPROD>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return GPS_ERR_LAUNCH;
    attr_lds = lds;
  }
  hipLaunchKernelGGL((sa_mlp_x3_kernel<CF, C1, C2, C3, WAVES, RESIDENT, PM, NPROD>), dim3(b), dim3(WAVES * 64), lds, s, b, n,
                     npoint, xyz, new_xyz, feats, idx, wpack, out, ld_feat, gps::object_extent());
  return GPS_OK;
}

}  // namespace x3

}  // namespace gps_sa

extern "C" {

long long gps_sa_mlp_wpack_floats(int c_in, int c1, int c2, int c3) {
  if (c_in < 1 || c1 < 32 || c2 < 32 || c3 < 32 || (c1 & 31) || (c2 & 31) || (c3 & 31)) return -1;
  return (long long)(c1 / 32) * gps_sa::tile_floats(c_in) + (long long)(c2 / 32) * gps_sa::tile_floats(c1) +
         (long long)(c3 / 32) * gps_sa::tile_floats(c2);
}

long long gps_sa_mlp_layer_floats(int c_in, int c_out) {
  if (c_in < 1 || c_out < 32 || (c_out & 31)) return -1;
  return (long long)(c_out / 32) * gps_sa::tile_floats(c_in);
}

int gps_sa_mlp_pack_layer(int c_in, int c_out, const float *w, const float *shift, float *dst,
                          gps_stream_t stream) {
  if (c_in < 1 || c_out < 32 || (c_out & 31) || !w || !shift || !dst) return GPS_ERR_INVALID_ARGUMENT;
  const int total = (c_out / 32) * gps_sa::tile_floats(c_in);
  hipLaunchKernelGGL(gps_sa::pack_layer_kernel, dim3((total + 255) / 256), dim3(256), 0,
                     (hipStream_t)stream, c_in, c_out, w, shift, dst);
  return hipGetLastError() == hipSuccess ? GPS_OK : GPS_ERR_LAUNCH;
}

int gps_sa_mlp_set_products(int n) {
  const int was = gps_sa::x3::g_products;
  if (n == 1 || n == 3) gps_sa::x3::g_products = n;
  return was;
}

long long gps_sa_mlp_layer_floats_bf16x3(int c_in, int c_out) {
  if (c_in < 1 || c_out < 32 || (c_out & 31)) return -1;
  return (long long)(c_out / 32) * gps_sa::x3::tile_floats16(c_in);
}

int gps_sa_mlp_pack_layer_bf16x3(int c_in, int c_out, const float *w, const float *shift, float *dst,
                                 gps_stream_t stream) {
  if (c_in < 1 || c_out < 32 || (c_out & 31) || !w || !shift || !dst) return GPS_ERR_INVALID_ARGUMENT;
  const int total = (c_out / 32) * gps_sa::x3::tile_floats16(c_in);
  hipLaunchKernelGGL(gps_sa::x3::pack_layer16_kernel, dim3((total + 255) / 256), dim3(256), 0,
                     (hipStream_t)stream, c_in, c_out, w, shift, dst);
  return hipGetLastError() == hipSuccess ? GPS_OK : GPS_ERR_LAUNCH;
}

int gps_sa_mlp_forward_bf16x3(int b, int n, int npoint, int nsample, int c_feat, int c1, int c2, int c3,
                              const float *xyz, const float *new_xyz, const float *features,
                              const int32_t *idx, const float *wpack, float *out, gps_stream_t stream) {
  if (b < 0 || n < 1 || npoint < 1 || nsample < 1 || c_feat < 0) return GPS_ERR_INVALID_ARGUMENT;
  if (b == 0) return GPS_OK;
  if (!xyz || !new_xyz || !idx || !wpack || !out || (c_feat > 0 && !features))
    return GPS_ERR_INVALID_ARGUMENT;
  if (nsample != gps_sa::kNS) return GPS_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  int st = GPS_ERR_UNSUPPORTED;
  if (c_feat == 3 && c1 == 64 && c2 == 64 && c3 == 128)
    st = gps_sa::x3::launch_sa_x3<3, 64, 64, 128, GPS_SA1_WAVES, true>(b, n, npoint, xyz, new_xyz, features, idx,
                                                                      wpack, out, s);
  else if (c_feat == 128 && c1 == 128 && c2 == 128 && c3 == 256)
    st = gps_sa::x3::launch_sa_x3<128, 128, 128, 256, GPS_SA2_WAVES, false>(b, n, npoint, xyz, new_xyz, features,
                                                                            idx, wpack, out, s);
  if (st != GPS_OK) return st;
  return hipGetLastError() == hipSuccess ? GPS_OK : GPS_ERR_LAUNCH;
}

int gps_sa_mlp_forward_bf16x3_pm(int b, int n, int npoint, int nsample, int c_feat, int c1, int c2, int c3,
                                 const float *xyz, const float *new_xyz, const float *features_pm, long long ld_feat,
                                 const int32_t *idx, const float *wpack, float *out, gps_stream_t stream) {
  if (b < 0 || n < 1 || npoint < 1 || nsample < 1 || c_feat < 1 || ld_feat < c_feat || ld_feat > 0x7FFFFFFF)
    return GPS_ERR_INVALID_ARGUMENT;
  if (b == 0) return GPS_OK;
  if (!xyz || !new_xyz || !idx || !wpack || !out || !features_pm) return GPS_ERR_INVALID_ARGUMENT;
  if (nsample != gps_sa::kNS) return GPS_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  int st = GPS_ERR_UNSUPPORTED;
  if (c_feat == 3 && c1 == 64 && c2 == 64 && c3 == 128)
    st = gps_sa::x3::launch_sa_x3<3, 64, 64, 128, GPS_SA1_WAVES, true, true>(b, n, npoint, xyz, new_xyz, features_pm, idx,
                                                                            wpack, out, s, (int)ld_feat);
  if (st != GPS_OK) return st;
  return hipGetLastError() == hipSuccess ? GPS_OK : GPS_ERR_LAUNCH;
}

int gps_sa_mlp_forward(int b, int n, int npoint, int nsample, int c_feat, int c1, int c2, int c3,
                       const float *xyz, const float *new_xyz, const float *features,
                       const int32_t *idx, const float *wpack, float *out, gps_stream_t stream) {
  if (b < 0 || n < 1 || npoint < 1 || nsample < 1 || c_feat < 0) return GPS_ERR_INVALID_ARGUMENT;
  if (b == 0) return GPS_OK;
  if (!xyz || !new_xyz || !idx || !wpack || !out || (c_feat > 0 && !features))
    return GPS_ERR_INVALID_ARGUMENT;
  if (nsample != gps_sa::kNS) return GPS_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  int st = GPS_ERR_UNSUPPORTED;
  if (c_feat == 3 && c1 == 64 && c2 == 64 && c3 == 128)
    st = gps_sa::launch_sa<3, 64, 64, 128>(b, n, npoint, xyz, new_xyz, features, idx, wpack, out, s);
  else if (c_feat == 128 && c1 == 128 && c2 == 128 && c3 == 256)
    st = gps_sa::launch_sa<128, 128, 128, 256>(b, n, npoint, xyz, new_xyz, features, idx, wpack, out, s);
  if (st != GPS_OK) return st;
  return hipGetLastError() == hipSuccess ? GPS_OK : GPS_ERR_LAUNCH;
}

}  // extern "C"
