// gps_gemm.hip -- hand-written bf16 MFMA GEMMs for the projection / FFN layers of the GPS transformers on
// MI355X (gfx950): every contraction of
//     MultiHeadAttentionSpatial  w_qs / w_ks / w_vs / lang_cond_fc / fc   modules/layers/transformers.py:173-186,193-197
//     nn.MultiheadAttention      in_proj / out_proj                        modules/layers/transformers.py:120-121,141
//     FFN linear1 / linear2 (+ GELU / ReLU, dropout)                       modules/layers/transformers.py:123-125,148-152,301-316
// in the three operand orders a training step needs, as ONE kernel template:
//
//     form NT   Y  (M x N) = X (M x K) . W (N x K)^T   forward            both operands K-major
//     form NN   dX (M x N) = dY (M x K) . W (K x N)    input gradient     B reduction-major  (ds_read_b64_tr_b16)
//     form TN   dW (M x N) = dY (K x M)^T . X (K x N)  weight gradient    both reduction-major, split over K
//                                                                          (K = tokens), fp32 out, bias gradient
//                                                                          (column sums of dY) from the same tiles
//
// bf16 operands, fp32 accumulation on v_mfma_f32_16x16x32_bf16.  Structure of a workgroup (4 or 8 waves):
//   * BM x BN output tile, BK = 64 per stage, every operand stage copied global -> LDS by global_load_lds_dwordx4
//     (no VGPR round trip), two or three stage buffers, ONE workgroup barrier per stage;
//   * the LDS images are swizzled through the per-lane SOURCE address (gps_gemm_layout.h): K-major fragments are
//     conflict-free ds_read_b128, reduction-major ones conflict-free hardware-transposed reads, so
//     no operand is ever transposed by stores and the weight never needs a transposed copy in HBM;
//   * the MFMA is issued with the operands swapped (D = B-frag x A-frag), so a lane ends up with FOUR CONSECUTIVE
//     output columns of one row: 8-byte bf16 / 16-byte fp32 stores, bias / activation / dropout applied in registers;
//   * out-of-range rows, columns and the K tail read a 64-byte block of zeros instead of being predicated;
//   * block id -> tile map keeps the tiles of one XCD (block id mod 8) contiguous in (split, tile_m, tile_n).
// Epilogues: bias | bias + GELU (+ dropout, also stores the pre-activation) | bias + ReLU (+ dropout) |
//            x GELU'(pre) x dropout-mask | x ReLU'(h) | fp32 (partial) sums + column sums.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "gps_gemm_layout.h"
#include "gps_hip.h"

namespace gps { const int *object_extent(); }   // gps_point_ops.hip

#ifndef GPS_GEMM_8P_PHASES
#define GPS_GEMM_8P_PHASES 2      // phases per K tile of the two-group 256 x 256 kernel (4 = the walk of rounds 3 - 6a; A/B builds)
#endif

namespace gps_gemm {

using namespace gps_gemm_layout;

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;

__device__ __attribute__((aligned(64))) const unsigned int g_zero_block[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

// Host-side state that belongs to a DEVICE (attributes granted to a kernel, CU counts, occupancy answers) is kept per
// device ordinal: one process may drive several GPUs (the usual deployment is one process per GPU, where index 0 is all
// that is ever used).
constexpr int kMaxDevices = 32;
inline int current_device() {
  int d = 0;
  return (hipGetDevice(&d) == hipSuccess && d >= 0 && d < kMaxDevices) ? d : -1;
}
struct PerDeviceFlag { bool done[kMaxDevices] = {}; };
// dynamic LDS above 64 KB must be granted to a kernel once per device
template <typename K>
inline bool grant_dynamic_lds(K kern, int bytes, PerDeviceFlag &f) {
  const int d = current_device();
  if (d < 0) return false;
  if (f.done[d]) return true;
  if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return false;
  f.done[d] = true;
  return true;
}
// CUs of the current device (0 on failure)
inline int device_cu_count() {
  static int n_cu[kMaxDevices] = {};
  const int d = current_device();
  if (d < 0) return 0;
  if (n_cu[d] == 0) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, d) != hipSuccess) return 0;
    n_cu[d] = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  return n_cu[d];
}

enum Epi : int {
  EPI_BIAS = 0,        // C bf16 = acc + bias
  EPI_BIAS_GELU = 1,   // aux_out bf16 = pre = acc + bias; C bf16 = dropout(gelu(bf16(pre)))
  EPI_BIAS_RELU = 2,   // C bf16 = dropout(relu(acc + bias))
  EPI_DGELU = 3,       // C bf16 = acc * gelu'(aux) * dropout-mask(idx)          (aux = saved pre-activation)
  EPI_DRELU = 4,       // C bf16 = acc * (aux != 0 ? keep_scale : 0)              (aux = saved FFN hidden)
  EPI_F32 = 5,         // C fp32 = acc (splits == 1) or partial[split] = acc; optional column sums of A
  EPI_RELU_SPLIT = 6,  // v = relu(acc + bias) as a bf16 pair (hi, lo = v - hi): C[m][n | N + n | 2N + n] = hi | lo | hi
  EPI_RELU_MAX16 = 7,  // C fp32 [m / 16][n] = max over the 16 rows of the block of relu(acc + bias)
  EPI_BIAS_GELU_FACTOR = 8,  // as EPI_BIAS_GELU, but aux_out bf16 = gelu'(bf16(pre)) * dropout-mask / (1 - p): the factor the
                             // backward pass multiplies by, computed here from the erf terms the activation needs anyway
  EPI_MUL_AUX = 9,     // C bf16 = acc * aux                                        (aux = that saved factor)
};

struct Params {
  int M, N, K;
  const uint16_t *A;
  long long lda;
  const uint16_t *B;
  long long ldb;
  void *C;
  long long ldc;
  const float *bias;
  const uint16_t *aux;
  long long ldaux;
  uint16_t *aux_out;
  long long ldaux_out;
  float *partial;          // EPI_F32, splits > 1: (splits, M, N)
  float *colsum;           // EPI_F32: (splits, M) partial column sums of A over k (or (M) when splits == 1), or null
  int splits, kt_per_split, nkt;
  int ntm, ntn, gm;        // gm: tile rows walked together (L2 working set of the workgroups resident on an XCD)
  float keep_scale;
  unsigned int drop_thr;
  unsigned long long seed;
  const unsigned long long *seed_dev;
  const int *extent_dev;   // optional device count of leading token rows that carry work (rows of M for NT / NN, of K for TN)
  int row0;                // NT / NN: this launch covers rows row0 .. row0 + M - 1 of a larger product (dropout stream index only)
  int accumulate;          // EPI_F32, splits == 1: C += acc, colsum += column sums (grouped weight gradients into live .grad buffers)
  unsigned int *sk_flags;  // stream-K form: one arrival word per workgroup position (zero between launches) + an error word;
                           // `partial` then holds one 256 x 256 fp32 slab per workgroup position
#ifdef GPS_GEMM_TRACE
  unsigned long long *trace;   // tools/probes/gemm_probe.hip: (workgroups, kTraceSlots) shader-clock stamps of wave 0
#endif
};

// Per-workgroup time stamps for tools/probes/gemm_probe.hip (compiled out of the library): slot 0 = kernel entry,
// 1 = first stage landed, 2 = main loop done, 3 = epilogue issued, 4 = s_memrealtime at entry, 5 = at exit, 8.. = K tiles.
#ifdef GPS_GEMM_TRACE
constexpr int kTraceSlots = 96;     // 0..31 workgroup stamps; 32..63 / 64..95: the phases of ONE K tile (the 9th) as seen by wave 0 / wave 4
#define GPS_TRACE(P, slot)                                                                                    \
  do {                                                                                                        \
    if ((P).trace && threadIdx.x == 0 && (slot) < kTraceSlots)                                                \
      (P).trace[(size_t)blockIdx.x * kTraceSlots + (slot)] = ((slot) == 4 || (slot) == 5) ? __builtin_amdgcn_s_memrealtime() : __builtin_amdgcn_s_memtime(); \
  } while (0)
#define GPS_TRACE_DRAIN() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")     // slot 6: the epilogue's stores acknowledged
#define GPS_PTRACE(P, t, idx)                                                                                 \
  do {                                                                                                        \
    if ((P).trace && (t) == 8 && (threadIdx.x == 0 || threadIdx.x == 256))                                    \
      (P).trace[(size_t)blockIdx.x * kTraceSlots + (threadIdx.x ? 64 : 32) + (idx)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#define GPS_TRACE_VAL(P, slot, v)                                                                             \
  do {                                                                                                        \
    if ((P).trace && threadIdx.x == 0 && (slot) < kTraceSlots) (P).trace[(size_t)blockIdx.x * kTraceSlots + (slot)] = (v); \
  } while (0)
#else
#define GPS_PTRACE(P, t, idx) do { } while (0)
#define GPS_TRACE_VAL(P, slot, v) do { } while (0)
#define GPS_TRACE(P, slot) do { } while (0)
#define GPS_TRACE_DRAIN() do { } while (0)
#endif

__device__ __forceinline__ uint16_t f2bf(float f) {   // round to nearest even (finite inputs)
  unsigned int u = __float_as_uint(f);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
__device__ __forceinline__ float bf2f(uint16_t h) { return __uint_as_float((unsigned int)h << 16); }
__device__ __forceinline__ unsigned int pack2(float lo, float hi) {
  return (unsigned int)f2bf(lo) | ((unsigned int)f2bf(hi) << 16);
}
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
__device__ __forceinline__ u32x2 pack4(const f32x4 &v) {     // v_cvt_pk_bf16_f32: round to nearest even
  const bf16x4 h = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
  return __builtin_bit_cast(u32x2, h);
}
// counter-based dropout RNG shared with gps_layernorm.hip / gps_attention.hip (splitmix64 finaliser)
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
// erf GELU (the reference's F.gelu / HF "gelu") and its derivative.  erf through Abramowitz-Stegun 7.1.26
// (|error| <= 1.5e-7, far below the bf16 rounding of the result) so that the epilogue costs ~15 VALU per element
// instead of libm erff's ~40; the one exponential exp(-x^2 / 2) serves both erf(x / sqrt 2) and the Gaussian density.
__device__ __forceinline__ void erf_terms(float x, float &erf_abs, float &e) {      // erf(|x| / sqrt 2), exp(-x^2 / 2)
  const float z = fabsf(x) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.f));   // v_rcp_f32, 1 ulp (__frcp_rn: a 10-instruction IEEE division per element)
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  e = __expf(-0.5f * x * x);
  erf_abs = 1.f - p * t * e;
}
__device__ __forceinline__ float gelu_f(float x) {
  float er, e;
  erf_terms(x, er, e);
  return 0.5f * x * (1.f + copysignf(er, x));
}
__device__ __forceinline__ float dgelu_f(float x) {
  float er, e;
  erf_terms(x, er, e);
  return 0.5f * (1.f + copysignf(er, x)) + x * 0.3989422804014327f * e;
}
__device__ __forceinline__ void gelu_and_dgelu(float x, float &g, float &dg) {    // both from one set of erf terms
  float er, e;
  erf_terms(x, er, e);
  const float cdf = 0.5f * (1.f + copysignf(er, x));
  g = x * cdf;
  dg = fmaf(x * 0.3989422804014327f, e, cdf);
}

__device__ __forceinline__ void glds16(const void *src, void *lds_dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                   (__attribute__((address_space(3))) void *)lds_dst, 16, 0, 0);
}

// ---------------------------------------------------------------------------------------------------------
// per-operand staging state.  A stage is copied with buffer_load_dwordx4 ... lds: the per-lane byte offset of its
// 16 bytes (voff, one VGPR per piece, fixed for the whole kernel) + a wave-uniform SGPR offset that advances by
// one stage per iteration -- no per-stage vector arithmetic at all.  Rows / columns outside the matrix are
// CLAMPED to the last valid one (finite duplicates that only feed output rows / columns which are never stored);
// only a ragged K tail needs zeros, and only its stage takes the slower path that points invalid lanes at a
// zero block.
// ---------------------------------------------------------------------------------------------------------
template <int ROWS, bool RM, int NW>
struct Stager {
  static constexpr int NPIECE = ROWS / 8 / NW;      // pieces per wave and stage (tile = ROWS x 64 bf16 = ROWS / 8 KiB)
  static_assert(ROWS % (8 * NW) == 0, "tile rows must split evenly over the waves");
  unsigned int voff[NPIECE];
  const unsigned char *base;
  unsigned int soff, step;

  // mat: K-major  -> element (r, k) at mat[r * ld + k], r in [0, rows): tile rows r0 .. r0 + ROWS - 1
  //      red-major -> element (k, c) at mat[k * ld + c], c in [0, rows): tile columns r0 .. r0 + ROWS - 1
  __device__ __forceinline__ void init(const uint16_t *mat, long long ld, int rows, int r0, int k_begin, int wave,
                                       int lane) {
    base = reinterpret_cast<const unsigned char *>(mat);
#pragma unroll
    for (int j = 0; j < NPIECE; ++j) {
      const int q = j * NW + wave;
      if (!RM) {
        int row, chunk;
        km_stage_src(q, lane, row, chunk);
        const int r = min(r0 + row, rows - 1);
        voff[j] = (unsigned int)(((long long)r * ld + 8 * chunk) * 2);
      } else {
        int k, chunk;
        rm_stage_src<ROWS>(q, lane, k, chunk);
        const int c = min(r0 + 8 * chunk, rows - 8);
        voff[j] = (unsigned int)(((long long)k * ld + c) * 2);
      }
    }
    step = RM ? (unsigned int)(BK * ld * 2) : (unsigned int)(BK * 2);
    soff = (unsigned int)k_begin / BK * step;
  }
  // copy one full stage into `tile` (LDS, ROWS * 128 bytes)
  __device__ __forceinline__ void issue_full(unsigned char *tile, int wave) {
#if defined(__HIP_DEVICE_COMPILE__)   // the buffer-resource builtins do not exist in hipcc's HOST pass over this file
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char *>(base), 0, 0x7FFFFFFF, 0x00020000);
#pragma unroll
    for (int j = 0; j < NPIECE; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void *)(tile + (j * NW + wave) * PIECE),
                                               16, voff[j], soff, 0, 0);
#endif
    soff += step;
  }
  // the same stage into REGISTERS (one 16-byte chunk per piece and lane) ...
  __device__ __forceinline__ void load_regs(u32x4 (&r)[NPIECE]) {
#if defined(__HIP_DEVICE_COMPILE__)
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char *>(base), 0, 0x7FFFFFFF, 0x00020000);
#pragma unroll
    for (int j = 0; j < NPIECE; ++j) r[j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff[j], soff, 0));
#endif
    soff += step;
  }
  // ... and from there into the stage buffer: the image the LDS-DMA form writes (piece base + lane * 16)
  __device__ __forceinline__ void store_regs(unsigned char *tile, const u32x4 (&r)[NPIECE], int wave, int lane) const {
#pragma unroll
    for (int j = 0; j < NPIECE; ++j) *reinterpret_cast<u32x4 *>(tile + (j * NW + wave) * PIECE + lane * 16) = r[j];
  }
  // the ragged last stage: only k_left (< 64) reduction indices are inside the matrix, the rest reads zeros
  __device__ __forceinline__ void issue_tail(unsigned char *tile, int k_left, int wave, int lane) {
    const unsigned long long zero = (unsigned long long)(uintptr_t)g_zero_block;
#pragma unroll
    for (int j = 0; j < NPIECE; ++j) {
      // reduction index of this lane's 16 bytes inside the stage (K-major: first k of the chunk), recomputed here
      // rather than carried in registers through the main loop
      int r_or_k, chunk;
      if (!RM) km_stage_src(j * NW + wave, lane, r_or_k, chunk);
      else rm_stage_src<ROWS>(j * NW + wave, lane, r_or_k, chunk);
      const int kk = RM ? r_or_k : 8 * chunk;
      unsigned int vo = voff[j];
      asm volatile("" : "+v"(vo));      // opaque: keeps the 64-bit forms of this rare path out of the main loop's registers
      const unsigned long long p = (unsigned long long)(uintptr_t)base + vo + soff;
      const unsigned long long full = (kk < k_left) ? ~0ull : 0ull;
      glds16(reinterpret_cast<const void *>((uintptr_t)(zero + ((p - zero) & full))), tile + (j * NW + wave) * PIECE);
    }
    soff += step;
  }
};

// ---------------------------------------------------------------------------------------------------------
// fragment reads (lane (i, g): element order of gps_gemm_layout.h)
// ---------------------------------------------------------------------------------------------------------
template <int ROWS, bool RM>
__device__ __forceinline__ bf16x8 read_frag(const unsigned char *tile, int r0, int ks, int lane) {
  if (RM) {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) s16x4 *)(tile + rm_frag<ROWS>(r0, ks, lane, 0)));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) s16x4 *)(tile + rm_frag<ROWS>(r0, ks, lane, 1)));
    const u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
    const u32x4 v = {l2[0], l2[1], h2[0], h2[1]};
    return __builtin_bit_cast(bf16x8, v);
  } else {
    const u32x4 v = *reinterpret_cast<const u32x4 *>(tile + km_frag(r0 + (lane & 15), ks, lane >> 4));
    return __builtin_bit_cast(bf16x8, v);
  }
}

// ---------------------------------------------------------------------------------------------------------
// the kernel
// ---------------------------------------------------------------------------------------------------------
// one stage of both operands into the stage buffer at `base` (A tile, then B tile)
template <int BM, int BN, bool ATR, bool BTR, int NW>
__device__ __forceinline__ void issue_stage(Stager<BM, ATR, NW> &sa, Stager<BN, BTR, NW> &sb, unsigned char *base,
                                            int &k_left, int wave, int lane) {
  if (k_left >= BK) {
    sa.issue_full(base, wave);
    sb.issue_full(base + BM * BK * 2, wave);
  } else {
    sa.issue_tail(base, k_left, wave, lane);
    sb.issue_tail(base + BM * BK * 2, k_left, wave, lane);
  }
  k_left -= BK;
}

// the MFMAs of one stage: acc[a][b] += B-fragment b x A-fragment a (operands swapped: a lane ends up with four
// consecutive output columns of one row); optional column sums of A through an all-ones operand
template <int BM, int BN, int TM, int TN, bool ATR, bool BTR, bool COLSUM, bool PF>
__device__ __forceinline__ void compute_stage(const unsigned char *As, int wm0, int wn0, int lane, f32x4 (&acc)[TM][TN],
                                              f32x4 (&csum)[TM], bool do_colsum) {
  const unsigned char *Bs = As + BM * BK * 2;
  const u32x4 ones_u = {0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};   // 8 x bf16 1.0
  const bf16x8 ones = __builtin_bit_cast(bf16x8, ones_u);
  if (PF) {
    // all fragment reads of the stage are issued before its first MFMA: one exposed LDS latency per stage, the
    // MFMAs of the first K step then cover the reads of the second
    bf16x8 af[BK / 32][TM], bf[BK / 32][TN];
#pragma unroll
    for (int ks = 0; ks < BK / 32; ++ks) {
#pragma unroll
      for (int a = 0; a < TM; ++a) af[ks][a] = read_frag<BM, ATR>(As, wm0 + 16 * a, ks, lane);
#pragma unroll
      for (int b = 0; b < TN; ++b) bf[ks][b] = read_frag<BN, BTR>(Bs, wn0 + 16 * b, ks, lane);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < BK / 32; ++ks) {
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[ks][b], af[ks][a], acc[a][b], 0, 0, 0);
      if (COLSUM) {
        if (do_colsum) {
#pragma unroll
          for (int a = 0; a < TM; ++a) csum[a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, af[ks][a], csum[a], 0, 0, 0);
        }
      }
    }
    return;
  }
  if constexpr (TM * TN >= 32) {
    // big wave tiles (128 accumulator registers): walk the tile in groups of 4 A fragments against all B fragments
    // of the K step, so that at most TN + 8 fragments are live (the next group's reads overlap this group's MFMAs);
    // the scheduling barriers keep the compiler from hoisting every read of the stage to its top (which spills)
    static_assert(TM % 4 == 0 && !COLSUM, "quadrant walk: 4 A fragments per group, no column sums");
#pragma unroll
    for (int ks = 0; ks < BK / 32; ++ks) {
      bf16x8 bf[TN], af[2][4];
#pragma unroll
      for (int b = 0; b < TN; ++b) bf[b] = read_frag<BN, BTR>(Bs, wn0 + 16 * b, ks, lane);
#pragma unroll
      for (int a = 0; a < 4; ++a) af[0][a] = read_frag<BM, ATR>(As, wm0 + 16 * a, ks, lane);
#pragma unroll
      for (int ag = 0; ag < TM / 4; ++ag) {
        if (ag + 1 < TM / 4) {
#pragma unroll
          for (int a = 0; a < 4; ++a) af[(ag + 1) & 1][a] = read_frag<BM, ATR>(As, wm0 + 16 * (4 * (ag + 1) + a), ks, lane);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int b = 0; b < TN; ++b)
            acc[4 * ag + a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[b], af[ag & 1][a], acc[4 * ag + a][b], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    return;
  }
#pragma unroll
  for (int ks = 0; ks < BK / 32; ++ks) {
    bf16x8 af[TM], bf[TN];
#pragma unroll
    for (int a = 0; a < TM; ++a) af[a] = read_frag<BM, ATR>(As, wm0 + 16 * a, ks, lane);
#pragma unroll
    for (int b = 0; b < TN; ++b) bf[b] = read_frag<BN, BTR>(Bs, wn0 + 16 * b, ks, lane);
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int b = 0; b < TN; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[b], af[a], acc[a][b], 0, 0, 0);
    if (COLSUM) {
      if (do_colsum) {
#pragma unroll
        for (int a = 0; a < TM; ++a) csum[a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, af[a], csum[a], 0, 0, 0);
      }
    }
  }
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  static_assert(N >= 0 && N <= 63, "vmcnt is a 6-bit field");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// wait until at most `stages` * NP of this wave's copies are outstanding (stages <= MAXS, wave-uniform)
template <int NP, int MAXS>
__device__ __forceinline__ void wait_stages(int stages) {
  if constexpr (MAXS == 0) {
    wait_vmcnt<0>();
  } else {
    if (stages >= MAXS) wait_vmcnt<NP * MAXS>();
    else wait_stages<NP, MAXS - 1>(stages);
  }
}

constexpr int lds_bytes(int BM, int BN, int NBUF) { return NBUF * (BM + BN) * BK * 2; }
constexpr int waves_per_simd(int BM, int BN, int NW, int NBUF) {
  const int blocks = (160 * 1024) / lds_bytes(BM, BN, NBUF);
  const int w = (blocks < 1 ? 1 : blocks) * NW / 4;
  if (NW >= 16) return 4;                 // one 1024-thread workgroup: 4 waves per SIMD, 128 VGPRs each
  return w < 1 ? 1 : (w > 2 ? 2 : w);     // the accumulators never leave room for more than 2
}

// Lane (i, g) holds the packed 4-column groups o0 (fragment b) and o1 (fragment b + 1) of output row i: columns 4 g .. 4 g + 3
// of each.  v_permlane16_swap exchanges the ODD 16-lane rows of its first operand with the EVEN rows of its second, i.e.
// lane group g = 2 j + 1 hands its o0 to group 2 j and gets that group's o1: afterwards an even group holds 8 consecutive
// columns of fragment b (8 j .. 8 j + 7) and an odd group 8 consecutive columns of fragment b + 1 -- one 16-byte store per
// lane, no LDS round trip (the first version: two ds_bpermute, a wait and four selects per pair).
__device__ __forceinline__ u32x4 pair_exchange(const u32x2 &o0, const u32x2 &o1) {
  const auto x0 = __builtin_amdgcn_permlane16_swap(o0[0], o1[0], false, false);
  const auto x1 = __builtin_amdgcn_permlane16_swap(o0[1], o1[1], false, false);
  return u32x4{x0[0], x1[0], x0[1], x1[1]};
}

// bias / activation / dropout of one 4-column group of row m -> the packed bf16 result (and the packed
// pre-activation / low halves through `pre` for the GELU and split forms); aux4 = this group's 4 saved values
template <int EPI>
__device__ __forceinline__ u32x2 epi_finish(const Params &P, bool dropout, unsigned long long seed, f32x4 v, int m, int n,
                                           const f32x4 &bias, const u32x2 &aux4, u32x2 &pre) {
  v = v + bias;
  const unsigned long long idx = (unsigned long long)(m + P.row0) * (unsigned long long)P.N + (unsigned long long)n;
  if (EPI == EPI_BIAS_GELU) {
    pre = pack4(v);
    v[0] = gelu_f(bf2f((uint16_t)(pre[0] & 0xFFFFu)));
    v[1] = gelu_f(bf2f((uint16_t)(pre[0] >> 16)));
    v[2] = gelu_f(bf2f((uint16_t)(pre[1] & 0xFFFFu)));
    v[3] = gelu_f(bf2f((uint16_t)(pre[1] >> 16)));
  } else if (EPI == EPI_BIAS_GELU_FACTOR) {
    // the activation from the bf16-rounded pre-activation (as EPI_BIAS_GELU), and the backward factor beside it
    const u32x2 pr = pack4(v);
    f32x4 fac;
    const float xin[4] = {bf2f((uint16_t)(pr[0] & 0xFFFFu)), bf2f((uint16_t)(pr[0] >> 16)), bf2f((uint16_t)(pr[1] & 0xFFFFu)),
                          bf2f((uint16_t)(pr[1] >> 16))};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float gv, dgv;
      gelu_and_dgelu(xin[r], gv, dgv);
      v[r] = gv;
      fac[r] = dgv;
    }
    if (dropout) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float k = rng_u32(seed, idx + r) >= P.drop_thr ? P.keep_scale : 0.f;
        v[r] *= k;
        fac[r] *= k;
      }
    }
    pre = pack4(fac);
    return pack4(v);
  } else if (EPI == EPI_MUL_AUX) {
    v[0] *= bf2f((uint16_t)(aux4[0] & 0xFFFFu));
    v[1] *= __uint_as_float(aux4[0] & 0xFFFF0000u);
    v[2] *= bf2f((uint16_t)(aux4[1] & 0xFFFFu));
    v[3] *= __uint_as_float(aux4[1] & 0xFFFF0000u);
  } else if (EPI == EPI_BIAS_RELU) {
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
  } else if (EPI == EPI_RELU_SPLIT) {
    // fp32 value carried as two bf16: hi = rne(v), lo = rne(v - hi) (exact difference); `pre` takes the lo words
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
    const u32x2 hi = pack4(v);
    f32x4 rest;
    rest[0] = v[0] - bf2f((uint16_t)(hi[0] & 0xFFFFu));
    rest[1] = v[1] - bf2f((uint16_t)(hi[0] >> 16));
    rest[2] = v[2] - bf2f((uint16_t)(hi[1] & 0xFFFFu));
    rest[3] = v[3] - bf2f((uint16_t)(hi[1] >> 16));
    pre = pack4(rest);
    return hi;
  } else if (EPI == EPI_DGELU) {
    v[0] *= dgelu_f(bf2f((uint16_t)(aux4[0] & 0xFFFFu)));
    v[1] *= dgelu_f(bf2f((uint16_t)(aux4[0] >> 16)));
    v[2] *= dgelu_f(bf2f((uint16_t)(aux4[1] & 0xFFFFu)));
    v[3] *= dgelu_f(bf2f((uint16_t)(aux4[1] >> 16)));
  } else if (EPI == EPI_DRELU) {
    v[0] = (aux4[0] & 0x7FFFu) ? v[0] * P.keep_scale : 0.f;
    v[1] = (aux4[0] & 0x7FFF0000u) ? v[1] * P.keep_scale : 0.f;
    v[2] = (aux4[1] & 0x7FFFu) ? v[2] * P.keep_scale : 0.f;
    v[3] = (aux4[1] & 0x7FFF0000u) ? v[3] * P.keep_scale : 0.f;
  }
  if (dropout && (EPI == EPI_BIAS_GELU || EPI == EPI_BIAS_RELU || EPI == EPI_DGELU)) {
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = rng_u32(seed, idx + r) >= P.drop_thr ? v[r] * P.keep_scale : 0.f;
  }
  return pack4(v);
}

// ---- epilogue of one tile: lane (i, g) holds C[m0 + wm0 + 16 a + i][n0 + wn0 + 16 b + 4 g + 0..3] -----------------
template <int TM, int TN, int EPI>
__device__ __forceinline__ void store_tile(const Params &P, f32x4 (&acc)[TM][TN], f32x4 (&csum)[TM], bool do_colsum,
                                           int split, int m0, int n0, int wm0, int wn0, int lane) {
  const int i = lane & 15, g = lane >> 4;
  if (EPI == EPI_F32) {
    float *out = (P.splits > 1) ? P.partial + (size_t)split * P.M * P.N : reinterpret_cast<float *>(P.C);
    const long long ldo = (P.splits > 1) ? (long long)P.N : P.ldc;
    const bool accum = P.accumulate && P.splits == 1;           // partial tiles are always plain stores
#pragma unroll
    for (int a = 0; a < TM; ++a) {
      const int m = m0 + wm0 + 16 * a + i;
#pragma unroll
      for (int b = 0; b < TN; ++b) {
        const int n = n0 + wn0 + 16 * b + 4 * g;
        if (m < P.M && n < P.N) {
          f32x4 *dst = reinterpret_cast<f32x4 *>(out + (size_t)m * ldo + n);
          *dst = accum ? *dst + acc[a][b] : acc[a][b];
        }
      }
      if (do_colsum && g == 0 && m < P.M) {
        float *cs = P.colsum + (size_t)split * P.M + m;
        *cs = accum ? *cs + csum[a][0] : csum[a][0];
      }
    }
    return;
  }
  if (EPI == EPI_RELU_MAX16) {
    // one 16-row fragment block = one group of 16 rows: its maximum lives in the 16 lanes i of a lane group
    float *out = reinterpret_cast<float *>(P.C);
#pragma unroll
    for (int b = 0; b < TN; ++b) {
      const int n = n0 + wn0 + 16 * b + 4 * g;
      f32x4 bias = {0.f, 0.f, 0.f, 0.f};
      if (P.bias && n < P.N) bias = *reinterpret_cast<const f32x4 *>(P.bias + n);
#pragma unroll
      for (int a = 0; a < TM; ++a) {
        const int mb = m0 + wm0 + 16 * a;
        f32x4 v = acc[a][b] + bias;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float x = (mb + i < P.M) ? fmaxf(v[r], 0.f) : 0.f;      // rows past M: the neutral element of max(relu)
          x = fmaxf(x, __shfl_xor(x, 1, 64));
          x = fmaxf(x, __shfl_xor(x, 2, 64));
          x = fmaxf(x, __shfl_xor(x, 4, 64));
          x = fmaxf(x, __shfl_xor(x, 8, 64));
          v[r] = x;
        }
        if (i == 0 && mb < P.M && n < P.N) *reinterpret_cast<f32x4 *>(out + (size_t)(mb >> 4) * P.ldc + n) = v;
      }
    }
    return;
  }
  const bool dropout = P.drop_thr != 0u;
  const unsigned long long seed = dropout ? P.seed + (P.seed_dev ? *P.seed_dev : 0ull) : 0ull;
  auto finish = [&](f32x4 v, int m, int n, const f32x4 &bias, const u32x2 &aux4, u32x2 &pre) -> u32x2 {
    return epi_finish<EPI>(P, dropout, seed, v, m, n, bias, aux4, pre);
  };
  constexpr bool HAS_BIAS = EPI == EPI_BIAS || EPI == EPI_BIAS_GELU || EPI == EPI_BIAS_RELU || EPI == EPI_RELU_SPLIT || EPI == EPI_BIAS_GELU_FACTOR;
  constexpr bool HAS_AUX = EPI == EPI_DGELU || EPI == EPI_DRELU || EPI == EPI_MUL_AUX;
  constexpr bool HAS_PRE = EPI == EPI_BIAS_GELU || EPI == EPI_BIAS_GELU_FACTOR;      // second bf16 output through aux_out
#if defined(__HIP_DEVICE_COMPILE__)   // buffer-resource builtins: device pass only
  // Everything goes through buffer descriptors with exact sizes: rows past M fall outside the descriptor and are
  // dropped (stores) or read as zero (loads) by the hardware, columns past N are sent there on purpose (kOOB) --
  // so the epilogue is straight-line code, and ALL its loads (bias, saved activations) are issued before its first
  // store.  That matters: vmcnt retires loads and stores in one order, so a load placed behind a store makes the
  // wave wait for the store's acknowledgement -- the first version did that once per 16-row fragment.
  constexpr unsigned int kOOB = 0x80000000u;
  const auto bytes_of = [&](long long ld) { return (unsigned int)((((long long)P.M - 1) * ld + P.N) * 2); };
  const __amdgpu_buffer_rsrc_t rC = __builtin_amdgcn_make_buffer_rsrc(P.C, 0, bytes_of(P.ldc) + (EPI == EPI_RELU_SPLIT ? 4u * P.N : 0u), 0x00020000);
  const __amdgpu_buffer_rsrc_t rAux = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t *>(P.aux), 0, HAS_AUX ? bytes_of(P.ldaux) : 0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rPre = __builtin_amdgcn_make_buffer_rsrc(P.aux_out, 0, (HAS_PRE && P.aux_out) ? bytes_of(P.ldaux_out) : 0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rBias = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(P.bias), 0, (HAS_BIAS && P.bias) ? 4u * P.N : 0u, 0x00020000);
  // Pairs of adjacent 16-column fragments leave as 16-byte stores: inside a pair, the even lane groups (g = 0, 2)
  // send their 4 columns of fragment b + 1 to the odd group next to them and receive that group's 4 columns of
  // fragment b, so every lane ends up with 8 consecutive columns of one row -- half the store instructions of the
  // natural 8-byte form (guide T21); the exchange is pair_exchange() = two v_permlane16_swap.  Needs whole 8-column groups
  // (N % 8 == 0) and 16-byte aligned rows; otherwise the 8-byte form below.
  const bool wide = (TN % 2 == 0) && (P.N % 8 == 0) && (P.ldc % 8 == 0) &&
                    (!HAS_PRE || P.aux_out == nullptr || P.ldaux_out % 8 == 0);
  if (EPI == EPI_RELU_SPLIT && !wide) return;        // the C entry point only admits N % 8 == 0 == ldc % 8 for this form
  // ---- all loads first ----
  f32x4 bias[TN];
  u32x2 aux[TM][TN];
#pragma unroll
  for (int b = 0; b < TN; ++b) {
    const int n = n0 + wn0 + 16 * b + 4 * g;
    bias[b] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rBias, n < P.N ? 4u * n : kOOB, 0, 0));
#pragma unroll
    for (int a = 0; a < TM; ++a) {
      aux[a][b] = u32x2{0u, 0u};
      if (HAS_AUX) {
        const int m = m0 + wm0 + 16 * a + i;
        aux[a][b] = __builtin_amdgcn_raw_buffer_load_b64(rAux, n < P.N ? (unsigned int)(((long long)m * P.ldaux + n) * 2) : kOOB, 0, 0);
      }
    }
  }
  if (wide) {
    const bool odd = g & 1;
#pragma unroll
    for (int b = 0; b < TN; b += 2) {
      const int nb0 = n0 + wn0 + 16 * b + 4 * g, nb1 = nb0 + 16;       // this lane's 4 columns in fragments b, b + 1
      const int n_out = n0 + wn0 + 16 * (b + (odd ? 1 : 0)) + 8 * (g >> 1);     // first of the 8 columns stored
      const unsigned int col_ok = n_out < P.N ? 0u : kOOB;
#pragma unroll
      for (int a = 0; a < TM; ++a) {
        const int m = m0 + wm0 + 16 * a + i;
        u32x2 pre0 = {0u, 0u}, pre1 = {0u, 0u};
        const u32x2 o0 = finish(acc[a][b], m, nb0, bias[b], aux[a][b], pre0);
        const u32x2 o1 = finish(acc[a][b + 1], m, nb1, bias[b + 1], aux[a][b + 1], pre1);
        const u32x4 out = pair_exchange(o0, o1);
        const unsigned int off = (unsigned int)(((long long)m * P.ldc + n_out) * 2) | col_ok;
        __builtin_amdgcn_raw_buffer_store_b128(out, rC, off, 0, 0);
        if (EPI == EPI_RELU_SPLIT || HAS_PRE) {
          const u32x4 po = pair_exchange(pre0, pre1);
          if (EPI == EPI_RELU_SPLIT) {
            __builtin_amdgcn_raw_buffer_store_b128(po, rC, off + 2u * P.N, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128(out, rC, off + 4u * P.N, 0, 0);
          } else {
            __builtin_amdgcn_raw_buffer_store_b128(po, rPre, (unsigned int)(((long long)m * P.ldaux_out + n_out) * 2) | col_ok, 0, 0);
          }
        }
      }
    }
    return;
  }
#pragma unroll
  for (int b = 0; b < TN; ++b) {
    const int n = n0 + wn0 + 16 * b + 4 * g;
    const unsigned int col_ok = n < P.N ? 0u : kOOB;
#pragma unroll
    for (int a = 0; a < TM; ++a) {
      const int m = m0 + wm0 + 16 * a + i;
      u32x2 pre = {0u, 0u};
      const u32x2 o = finish(acc[a][b], m, n, bias[b], aux[a][b], pre);
      if (HAS_PRE)
        __builtin_amdgcn_raw_buffer_store_b64(pre, rPre, (unsigned int)(((long long)m * P.ldaux_out + n) * 2) | col_ok, 0, 0);
      __builtin_amdgcn_raw_buffer_store_b64(o, rC, (unsigned int)(((long long)m * P.ldc + n) * 2) | col_ok, 0, 0);
    }
  }
#endif
}

// ---- epilogue of the two-group 256 x 256 kernels: the wave's four 64 x 32 quadrants in one pass ---------------------------
// Four store_tile calls would each start with their own loads (bias, saved activations) and wait for them: four exposed
// memory round trips per tile and wave (profiles/r6/gemm_probe_trace_a.jsonl: 4.4 us of epilogue per tile at EPI_BIAS
// although the store path takes 128 KB in 1.8 us, gemm_probe storebw).  Here the bias vectors arrive as arguments (fetched by
// load_bias_8p before the main loop: 16 registers) and the saved activations of quadrant q + 1 are requested before
// quadrant q is finished and stored.
template <int EPI>
__device__ __forceinline__ void load_bias_8p(const Params &P, int n0, int wc, int lane, f32x4 (&bias4)[2][2]) {
  constexpr bool HAS_BIAS = EPI == EPI_BIAS || EPI == EPI_BIAS_GELU || EPI == EPI_BIAS_RELU || EPI == EPI_RELU_SPLIT || EPI == EPI_BIAS_GELU_FACTOR;
  if constexpr (EPI == EPI_F32 || EPI == EPI_RELU_SPLIT || EPI == EPI_RELU_MAX16) return;      // (store_quads takes the store_tile path)
#if defined(__HIP_DEVICE_COMPILE__)
  const __amdgpu_buffer_rsrc_t rBias = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(P.bias), 0, (HAS_BIAS && P.bias) ? 4u * P.N : 0u, 0x00020000);
  const int g = lane >> 4;
#pragma unroll
  for (int qb = 0; qb < 2; ++qb)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int n = n0 + 128 * qb + 32 * wc + 16 * b + 4 * g;
      bias4[qb][b] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rBias, n < P.N ? 4u * n : 0x80000000u, 0, 0));
    }
#endif
}

template <int EPI>
__device__ __forceinline__ void store_quads(const Params &P, f32x4 (&acc)[2][2][4][2], int split, int m0, int n0, int wr, int wc,
                                            int lane, const f32x4 (&bias4)[2][2]) {
  constexpr bool HAS_AUX = EPI == EPI_DGELU || EPI == EPI_DRELU || EPI == EPI_MUL_AUX;
  constexpr bool HAS_PRE = EPI == EPI_BIAS_GELU || EPI == EPI_BIAS_GELU_FACTOR;
  const bool wide = (P.N % 8 == 0) && (P.ldc % 8 == 0) && (!HAS_PRE || P.aux_out == nullptr || P.ldaux_out % 8 == 0);
  if (EPI == EPI_F32 || EPI == EPI_RELU_SPLIT || EPI == EPI_RELU_MAX16 || !wide) {
    f32x4 unused[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
      store_tile<4, 2, EPI>(P, acc[q >> 1][q & 1], unused, false, split, m0, n0, 128 * (q >> 1) + 64 * wr, 128 * (q & 1) + 32 * wc, lane);
    return;
  }
#if defined(__HIP_DEVICE_COMPILE__)
  const int i = lane & 15, g = lane >> 4;
  const bool odd = g & 1;
  const bool dropout = P.drop_thr != 0u;
  const unsigned long long seed = dropout ? P.seed + (P.seed_dev ? *P.seed_dev : 0ull) : 0ull;
  constexpr unsigned int kOOB = 0x80000000u;
  const auto bytes_of = [&](long long ld) { return (unsigned int)((((long long)P.M - 1) * ld + P.N) * 2); };
  const __amdgpu_buffer_rsrc_t rC = __builtin_amdgcn_make_buffer_rsrc(P.C, 0, bytes_of(P.ldc), 0x00020000);
  const __amdgpu_buffer_rsrc_t rAux = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t *>(P.aux), 0, HAS_AUX ? bytes_of(P.ldaux) : 0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rPre = __builtin_amdgcn_make_buffer_rsrc(P.aux_out, 0, (HAS_PRE && P.aux_out) ? bytes_of(P.ldaux_out) : 0u, 0x00020000);
  auto row_of = [&](int q, int a) { return m0 + 128 * (q >> 1) + 64 * wr + 16 * a + i; };
  auto col_of = [&](int q, int b) { return n0 + 128 * (q & 1) + 32 * wc + 16 * b + 4 * g; };
  auto load_aux = [&](int q, u32x2 (&aux)[4][2]) {
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        aux[a][b] = u32x2{0u, 0u};
        if (HAS_AUX) {
          const int n = col_of(q, b);
          aux[a][b] = __builtin_amdgcn_raw_buffer_load_b64(rAux, n < P.N ? (unsigned int)(((long long)row_of(q, a) * P.ldaux + n) * 2) : kOOB, 0, 0);
        }
      }
  };
  // quadrant order C00, C01, C11, C10: the order their last MFMAs were issued in
  constexpr int kOrder[4] = {0, 1, 3, 2};
  u32x2 aux[2][4][2];
  load_aux(kOrder[0], aux[0]);
#pragma unroll
  for (int qi = 0; qi < 4; ++qi) {
    const int q = kOrder[qi];
    if (HAS_AUX && qi + 1 < 4) load_aux(kOrder[qi + 1], aux[(qi + 1) & 1]);
    const int n_out = n0 + 128 * (q & 1) + 32 * wc + 16 * (odd ? 1 : 0) + 8 * (g >> 1);     // first of the 8 columns this lane stores
    const unsigned int col_ok = n_out < P.N ? 0u : kOOB;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int m = row_of(q, a);
      u32x2 pre0 = {0u, 0u}, pre1 = {0u, 0u};
      const u32x2 o0 = epi_finish<EPI>(P, dropout, seed, acc[q >> 1][q & 1][a][0], m, col_of(q, 0), bias4[q & 1][0], aux[qi & 1][a][0], pre0);
      const u32x2 o1 = epi_finish<EPI>(P, dropout, seed, acc[q >> 1][q & 1][a][1], m, col_of(q, 1), bias4[q & 1][1], aux[qi & 1][a][1], pre1);
      __builtin_amdgcn_raw_buffer_store_b128(pair_exchange(o0, o1), rC, (unsigned int)(((long long)m * P.ldc + n_out) * 2) | col_ok, 0, 0);
      if (HAS_PRE)
        __builtin_amdgcn_raw_buffer_store_b128(pair_exchange(pre0, pre1), rPre, (unsigned int)(((long long)m * P.ldaux_out + n_out) * 2) | col_ok, 0, 0);
    }
  }
#endif
}

// One workgroup = one output tile (PERSIST = false: grid = tiles x splits), or a resident workgroup that walks
// every (workgroups per XCD)-th tile of its XCD's contiguous range (PERSIST = true: grid = what the chip holds at
// once).  The persistent form keeps the stage ring running ACROSS tiles: the first stage(s) of the next tile are
// issued before the last MFMAs of the current one, so its epilogue (conversions, activation, stores) overlaps the
// next tile's first copies, and the per-workgroup launch cost is paid once.
template <int BM, int BN, int WGM, int WGN, bool ATR, bool BTR, int EPI, int NBUF, bool PF, bool PERSIST>
__global__ __launch_bounds__(WGM *WGN * 64, waves_per_simd(BM, BN, WGM *WGN, NBUF)) void gemm_kernel(const Params P) {
  constexpr int NW = WGM * WGN;
  constexpr int WM = BM / WGM, WN = BN / WGN;
  constexpr int TM = WM / 16, TN = WN / 16;
  constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, STAGE = A_BYTES + B_BYTES;
  static_assert(!PERSIST || EPI != EPI_F32, "the split-K form is one workgroup per (tile, split)");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // NBUF * STAGE

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm0 = (wave / WGN) * WM, wn0 = (wave % WGN) * WN;

  // block -> (split, tile_m, tile_n).  The blocks of one XCD (block id mod 8) take a contiguous range of virtual
  // ids; inside a split the tiles are walked in groups of GM tile rows, tile_m fastest, so that the workgroups
  // resident on an XCD at any time share a few A row panels and a few B panels (both stay in its 4 MiB L2).
  const int GM = P.gm;
  // NT / NN with a device-side row extent: only the tile rows below it exist.  The LIVE tiles are re-enumerated (and
  // spread over the XCDs) as if the matrix ended there: with the static enumeration the dead tiles -- the last tile
  // rows -- are exactly the virtual-id range of the last XCDs, which then idle while the first ones do all the work.
  int ntm = P.ntm;
  if (!ATR && !PERSIST && P.extent_dev) ntm = min(P.ntm, max(0, (*P.extent_dev + BM - 1) / BM));
  const int ntiles = ntm * P.ntn;
  const int total = ntiles * P.splits;
  int vid, vid_end, vid_step;
  if (!PERSIST && (int)blockIdx.x >= total) return;
  if (PERSIST) {
    const int xcd = blockIdx.x & 7, per = (total + 7) >> 3;
    vid = xcd * per + (int)(blockIdx.x >> 3);
    vid_end = min(total, (xcd + 1) * per);
    vid_step = (int)(gridDim.x >> 3);
    if (vid >= vid_end) return;
  } else {
    vid = xcd_virtual_id(blockIdx.x, total);
    vid_end = vid + 1;
    vid_step = 1;
  }
  int split, m0, n0, tile_n;
  auto decode = [&](int v) {
    split = v / ntiles;
    const int tile = v - split * ntiles;
    const int group = tile / (GM * P.ntn), in_group = tile - group * (GM * P.ntn);
    const int gm = min(GM, ntm - group * GM);
    tile_n = in_group / gm;
    m0 = (group * GM + (in_group - tile_n * gm)) * BM;
    n0 = tile_n * BN;
  };
  decode(vid);
  int kt0 = split * P.kt_per_split;
  int nst = min(P.nkt, kt0 + P.kt_per_split) - kt0;             // the same for every tile of a persistent walk
  int k_eff = P.K;                                               // reduction indices that carry work
  if (P.extent_dev) {
    const int extent = *P.extent_dev;
    if (ATR) {
      // TN: token rows are the reduction.  Only the first `extent` of them exist: the live stages are spread evenly
      // over the splits (a static partition would leave the later splits idle), and the stage that contains row
      // `extent` takes the ragged-tail path, which reads zeros past it -- rows beyond the extent may hold anything
      // (they were never written by the extent-aware producers) and must not reach the sums.
      k_eff = min(P.K, max(extent, 0));
      const int nkt_eff = (k_eff + BK - 1) / BK;
      const int per = (nkt_eff + P.splits - 1) / P.splits;
      kt0 = split * per;
      nst = max(0, min(nkt_eff, kt0 + per) - kt0);
    } else if (!PERSIST && m0 >= extent) {
      return;                                                    // NT / NN: a tile of rows nobody reads
    }
  }

  GPS_TRACE(P, 0);
  GPS_TRACE(P, 4);
  Stager<BM, ATR, NW> sa;
  Stager<BN, BTR, NW> sb;
  sa.init(P.A, P.lda, P.M, m0, kt0 * BK, wave, lane);
  sb.init(P.B, P.ldb, P.N, n0, kt0 * BK, wave, lane);
  constexpr int NP = Stager<BM, ATR, NW>::NPIECE + Stager<BN, BTR, NW>::NPIECE;
  int k_left = k_eff - kt0 * BK;                     // reduction indices from the next stage to issue onwards

  // ring of NBUF stage buffers: NBUF - 1 stages are in flight ahead of the one being computed; each wave waits for
  // ITS OWN copies of the stage with a counted vmcnt, the (raw) barrier then makes every wave's copies visible and
  // guarantees that the buffer about to be refilled (the one computed in the previous iteration) is no longer read
#pragma unroll
  for (int s = 0; s < NBUF - 1; ++s)
    if (s < nst) issue_stage<BM, BN, ATR, BTR, NW>(sa, sb, smem + s * STAGE, k_left, wave, lane);
  int cur = 0, fill = NBUF - 1;

  for (;;) {
    f32x4 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int b = 0; b < TN; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    // bias gradient of the TN form: column sums of A over k via an all-ones operand (wave column 0 of tile_n 0)
    const bool do_colsum = (EPI == EPI_F32) && P.colsum != nullptr && tile_n == 0 && wn0 == 0;
    f32x4 csum[TM];
#pragma unroll
    for (int a = 0; a < TM; ++a) csum[a] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int split_c = split, m0_c = m0, n0_c = n0;
    const int vnext = vid + vid_step;
    const bool has_next = PERSIST && vnext < vid_end;

    for (int it = 0; it < nst; ++it) {
      // stages that may stay in flight behind the one about to be read; with a next tile the ring stays full (the
      // epilogue's stores, issued after the prefetched copies, only make the count conservative)
      wait_stages<NP, NBUF - 2>(has_next ? NBUF - 2 : min(nst - 1 - it, NBUF - 2));
      __builtin_amdgcn_s_barrier();
      if (it == 0) GPS_TRACE(P, 1);
      GPS_TRACE(P, 8 + it);
      {
        if (it + NBUF - 1 < nst) {
          issue_stage<BM, BN, ATR, BTR, NW>(sa, sb, smem + fill * STAGE, k_left, wave, lane);
        } else if (PERSIST && has_next) {
          if (it + NBUF - 1 == nst) {                 // first stage of the next tile: re-aim the stagers
            decode(vnext);
            kt0 = split * P.kt_per_split;
            sa.init(P.A, P.lda, P.M, m0, kt0 * BK, wave, lane);
            sb.init(P.B, P.ldb, P.N, n0, kt0 * BK, wave, lane);
            k_left = P.K - kt0 * BK;
          }
          if (it + NBUF - 1 - nst < nst) issue_stage<BM, BN, ATR, BTR, NW>(sa, sb, smem + fill * STAGE, k_left, wave, lane);
        }
      }
      compute_stage<BM, BN, TM, TN, ATR, BTR, EPI == EPI_F32, PF>(smem + cur * STAGE, wm0, wn0, lane, acc, csum, do_colsum);
      cur = (cur == NBUF - 1) ? 0 : cur + 1;
      fill = (fill == NBUF - 1) ? 0 : fill + 1;
    }
    GPS_TRACE(P, 2);
    store_tile<TM, TN, EPI>(P, acc, csum, do_colsum, split_c, m0_c, n0_c, wm0, wn0, lane);
    GPS_TRACE(P, 3);
    GPS_TRACE_DRAIN();
    GPS_TRACE(P, 6);
    GPS_TRACE(P, 5);
    if (!has_next) break;
    vid = vnext;
  }
}

// ---------------------------------------------------------------------------------------------------------
// 256 x 256 tile, FOUR waves (one per SIMD, 512 registers each: the 256 accumulators of a 128 x 128 wave tile live in
// AGPRs), operands staged through REGISTERS two stages ahead.  Why: with LDS-resident staging a CU
// never has more than one stage in flight (64 KB at this tile) and a stage's L2 -> LDS round trip under full-chip
// load (1.1 - 2 us) exceeds its MFMA time (0.86 us); the 128 x 128 tiles have two workgroups per CU but need twice
// the L2 bytes per flop (the eight L2s cannot deliver them).  Here stage s is requested into registers at the middle
// of iteration s - 3, written to its LDS buffer at the middle of iteration s - 1 (the buffer was freed by the barrier
// that opened that iteration) and read by the MFMAs of iteration s: two full iterations of latency cover, 128 KB of
// operands in flight per CU besides the 64 KB being computed on.  Forms NT / NN (B_TR), K % 64 == 0; the LDS images,
// fragment reads, tile walk and epilogues are those of gemm_kernel.
// The MFMAs are inline asm with the accumulator tied in place ("+a"): left to the register allocator, the two
// copies of the loop body (one per register set) get DIFFERENT accumulator assignments and every iteration pays
// ~480 v_accvgpr moves and 46 scratch accesses (whose vmcnt waits also serialise the prefetch); pinned, the loop
// is 256 MFMAs, 64 fragment reads, 32 buffer loads, 32 ds_write_b128 and nothing else.
// ---------------------------------------------------------------------------------------------------------
template <bool BTR, int EPI>
__global__ __launch_bounds__(256, 1) void gemm256_kernel(const Params P) {
  constexpr int BM = 256, BN = 256, NW = 4, WGN = 2, WM = 128, WN = 128, TM = 8, TN = 8;
  constexpr int A_BYTES = BM * BK * 2, STAGE = 2 * A_BYTES;
  constexpr int NPA = Stager<BM, false, NW>::NPIECE, NPB = Stager<BN, BTR, NW>::NPIECE;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // 2 * STAGE

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm0 = (wave / WGN) * WM, wn0 = (wave % WGN) * WN;
  const int GM = P.gm;
  int ntm = P.ntm;
  if (P.extent_dev) ntm = min(P.ntm, max(0, (*P.extent_dev + BM - 1) / BM));      // live tile rows only (see gemm_kernel)
  const int ntiles = ntm * P.ntn;
  if ((int)blockIdx.x >= ntiles) return;
  const int vid = xcd_virtual_id(blockIdx.x, ntiles);
  const int group = vid / (GM * P.ntn), in_group = vid - group * (GM * P.ntn);
  const int gmr = min(GM, ntm - group * GM);
  const int tile_n = in_group / gmr;
  const int m0 = (group * GM + (in_group - tile_n * gmr)) * BM, n0 = tile_n * BN;
  if (P.extent_dev && m0 >= *P.extent_dev) return;                  // a tile of rows nobody reads
  const int nst = P.nkt;

  Stager<BM, false, NW> sa;
  Stager<BN, BTR, NW> sb;
  sa.init(P.A, P.lda, P.M, m0, 0, wave, lane);
  sb.init(P.B, P.ldb, P.N, n0, 0, wave, lane);
  u32x4 ra0[NPA], rb0[NPB], ra1[NPA], rb1[NPB];

  f32x4 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  // one K step (32 reduction indices) of the stage at `As`: quadrant walk, 4 A fragments x 8 B fragments per group
  auto compute_ks = [&](const unsigned char *As, int ks) {
    const unsigned char *Bs = As + A_BYTES;
    bf16x8 bf[TN], af[2];
#pragma unroll
    for (int b = 0; b < TN; ++b) bf[b] = read_frag<BN, BTR>(Bs, wn0 + 16 * b, ks, lane);
    af[0] = read_frag<BM, false>(As, wm0, ks, lane);
#pragma unroll
    for (int a = 0; a < TM; ++a) {
      if (a + 1 < TM) af[(a + 1) & 1] = read_frag<BM, false>(As, wm0 + 16 * (a + 1), ks, lane);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int b = 0; b < TN; ++b)
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[a][b]) : "v"(bf[b]), "v"(af[a & 1]));
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // iteration `it`: MFMAs of stage it; in its middle the registers holding stage it + 1 go to the other LDS buffer and
  // are re-used for the request of stage it + 3 (the other register set holds stage it + 2, still in flight).
  // Branch-free: requests past the last stage re-read the last one (soff is clamped) and are never consumed.
  const unsigned int soff_last_a = (unsigned int)(nst - 1) * sa.step, soff_last_b = (unsigned int)(nst - 1) * sb.step;
  auto request = [&](u32x4 (&ra)[NPA], u32x4 (&rb)[NPB]) {
    sa.soff = min(sa.soff, soff_last_a);
    sb.soff = min(sb.soff, soff_last_b);
    sa.load_regs(ra);
    sb.load_regs(rb);
  };
  auto body = [&](int it, u32x4 (&ra)[NPA], u32x4 (&rb)[NPB]) {
    unsigned char *cur = smem + (it & 1) * STAGE, *nxt = smem + ((it + 1) & 1) * STAGE;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // this wave's ds_writes of stage `it` are in LDS
    __builtin_amdgcn_s_barrier();                            // ... everybody's; and nobody still reads `nxt`
    compute_ks(cur, 0);
    sa.store_regs(nxt, ra, wave, lane);                      // the compiler's vmcnt wait for (ra, rb) sits here
    sb.store_regs(nxt + A_BYTES, rb, wave, lane);
    request(ra, rb);
    compute_ks(cur, 1);
  };
  // prologue: stage 0 -> registers -> LDS buffer 0; stages 1 and 2 stay in flight in the two register sets
  request(ra1, rb1);
  request(ra0, rb0);
  sa.store_regs(smem, ra1, wave, lane);
  sb.store_regs(smem + A_BYTES, rb1, wave, lane);
  request(ra1, rb1);
  // even iterations hand over register set 0 (stages 1, 3, ...), odd ones set 1 (stages 2, 4, ...)
  int it = 0;
#pragma clang loop unroll(disable)
  for (; it + 1 < nst; it += 2) {
    body(it, ra0, rb0);
    body(it + 1, ra1, rb1);
  }
  if (it < nst) body(it, ra0, rb0);
  // the MFMAs above are inline asm: the compiler does not know their latency, so the wait states between the last
  // of them and the first read of an accumulator are spelled out (8 passes of 4 cycles, guide: MFMA -> VALU read)
  asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
  f32x4 csum[TM];
  store_tile<TM, TN, EPI>(P, acc, csum, false, 0, m0, n0, wm0, wn0, lane);
}

template <bool BTR, int EPI>
int launch_256(Params &P, hipStream_t s) {
  constexpr int LDS = 2 * 2 * 256 * BK * 2;              // two stages of (A | B) = 128 KB
  P.ntm = (P.M + 255) / 256;
  P.ntn = (P.N + 255) / 256;
  P.gm = 4;
  auto kern = &gemm256_kernel<BTR, EPI>;
  static PerDeviceFlag granted;
  if (!grant_dynamic_lds(kern, LDS, granted)) return GPS_ERR_LAUNCH;
  const long long blocks = (long long)P.ntm * P.ntn;
  if (blocks <= 0 || blocks > 0x7FFFFFFFLL) return GPS_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), LDS, s, P);
  return hipGetLastError() == hipSuccess ? GPS_OK : GPS_ERR_LAUNCH;
}

// ---------------------------------------------------------------------------------------------------------
// 256 x 256 tile, EIGHT waves in two groups of four that alternate on the matrix pipe (the guide's "8-phase" schedule,
// here with 4 phases per 64-deep K tile and the buffer parity at run time).  One workgroup per CU, two waves per SIMD
// (one of each group).  A phase of a wave = [fragment reads of one 64 x 32 output quadrant | LDS-DMA of one half-tile
// of a later K tile] barrier [16 MFMAs] barrier; group 1 runs ONE BARRIER behind group 0, so while one group's waves
// issue their 16 MFMAs (256 cycles of its SIMD) the other group's waves read fragments and issue copies -- the matrix
// pipe of every SIMD always has a wave in its MFMA segment.  Per wave and K tile: 24 ds_read_b128 (or 48 tr reads) for
// 64 MFMAs, a quarter of the LDS bytes per flop of the 32 x 64 wave tiles of gemm_kernel.
//
// LDS: two buffers of four half-tile images (A rows 0..127 | A rows 128..255 | B cols 0..127 | B cols 128..255, 16 KB
// each, the images of gemm_kernel's 128-row tiles).  Wave (wr, wc) owns output rows {64 wr + [0, 64)} of BOTH A halves
// and columns {32 wc + [0, 32)} of BOTH B halves, i.e. four 64 x 32 quadrants C[qa][qb]; the phases walk
// C00 (reads Bq0, Aq0) -> C01 (reads Bq1) -> C11 (reads Aq1) -> C10 (Bq0 kept in registers), so that half B0 is read in
// phase 1 only, B1 in phase 2, A0 in phase 1, A1 in phase 3, and each half can be refilled early:
//   phase 1 of tile t : copy A1(t + 1) -> other buffer     (last read: phase 3 of tile t - 1)
//   phase 2           : copy B0(t + 2) -> this buffer       (read in phase 1, retired by lgkmcnt(8) BEFORE that phase's
//                                                            first barrier: safe one phase later)
//   phase 3           : copy A0(t + 2) -> this buffer       (read in phase 1: two phases later)
//   phase 4           : copy B1(t + 2) -> this buffer       (read in phase 2); vmcnt(6): everything up to A1(t + 1) has
//                                                            landed, the three halves of tile t + 2 stay in flight
// RAW: a wave's counted vmcnt precedes the first barrier of its phase 4, the reads of tile t + 1 start after the
// second one (for group 0: after group 1's wait as well, because group 1's first barrier of a phase IS group 0's second).
// Forms NT / NN; K a multiple of 64; no split-K.
// ---------------------------------------------------------------------------------------------------------
// one 256 x 256 output tile at (m0, n0): K tiles kt0 .. kt0 + nst - 1 of the reduction (k_eff = reduction indices that
// carry work), partial index `split`; tile_n == 0 computes the column sums of the TN form
template <bool ATR, bool BTR, int EPI, bool RAGGED = false>
__device__ __forceinline__ void gemm8p_tile(const Params &P, unsigned char *smem, int lane, int wave, int m0, int n0,
                                            int tile_n, int split, int kt0, int nst, int k_eff);

template <bool ATR, bool BTR, int EPI>
__global__ __launch_bounds__(512, 2) void gemm8p_kernel(const Params P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // 2 * BUF = 128 KB

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int GM = P.gm;
  int ntm = P.ntm;
  if (!ATR && P.extent_dev) ntm = min(P.ntm, max(0, (*P.extent_dev + 255) / 256));   // live tile rows only (see gemm_kernel)
  const int ntiles = ntm * P.ntn;
  const int total = ntiles * P.splits;
  if ((int)blockIdx.x >= total) return;
  const int vid = xcd_virtual_id(blockIdx.x, total);
  const int split = vid / ntiles, tile = vid - split * ntiles;
  const int group = tile / (GM * P.ntn), in_group = tile - group * (GM * P.ntn);
  const int gmr = min(GM, ntm - group * GM);
  const int tile_n = in_group / gmr;
  const int m0 = (group * GM + (in_group - tile_n * gmr)) * 256, n0 = tile_n * 256;
  // reduction range of this workgroup: K tiles kt0 .. kt0 + nst - 1; TN (ATR) with a device-side extent: only the
  // first `extent` token rows exist, their K tiles are spread evenly over the splits and the tile that contains
  // row `extent` reads zeros past it (gemm_kernel's rule)
  int kt0 = split * P.kt_per_split;
  int nst = min(P.nkt, kt0 + P.kt_per_split) - kt0;
  int k_eff = P.K;
  if (ATR && P.extent_dev) {
    k_eff = min(P.K, max(*P.extent_dev, 0));
    const int nkt_eff = (k_eff + BK - 1) / BK;
    const int per = (nkt_eff + P.splits - 1) / P.splits;
    kt0 = split * per;
    nst = max(0, min(nkt_eff, kt0 + per) - kt0);
  }
  gemm8p_tile<ATR, BTR, EPI>(P, smem, lane, wave, m0, n0, tile_n, split, kt0, nst, k_eff);
}

// RAGGED (forms NT / NN): K % 64 != 0, the last K tile reads zeros past K (as the token-row reduction of the TN form always may)
template <bool ATR, bool BTR, int EPI, bool RAGGED>
__device__ __forceinline__ void gemm8p_tile(const Params &P, unsigned char *smem, int lane, int wave, int m0, int n0,
                                            int tile_n, int split, int kt0, int nst, int k_eff) {
  constexpr int HALF = 128 * BK * 2, BUF = 4 * HALF;
  constexpr int OFF_A0 = 0, OFF_A1 = HALF, OFF_B0 = 2 * HALF, OFF_B1 = 3 * HALF;
  constexpr bool COLSUM = EPI == EPI_F32 && ATR;
  const int wr = wave >> 2, wc = wave & 3;
  GPS_TRACE(P, 0);
  GPS_TRACE(P, 4);
  const int k_span = k_eff - kt0 * BK;                      // reduction indices from this workgroup's first K tile on

  Stager<128, ATR, 8> sa0, sa1;
  Stager<128, BTR, 8> sb0, sb1;
  sa0.init(P.A, P.lda, P.M, m0, kt0 * BK, wave, lane);
  sa1.init(P.A, P.lda, P.M, m0 + 128, kt0 * BK, wave, lane);
  sb0.init(P.B, P.ldb, P.N, n0, kt0 * BK, wave, lane);
  sb1.init(P.B, P.ldb, P.N, n0 + 128, kt0 * BK, wave, lane);
  // half-tile of K tile `tj` (relative to kt0) into `dst`; only a reduction-major (token-row) K can be ragged
  auto issue = [&](auto &st, unsigned char *dst, int tj) {
    if constexpr (ATR || RAGGED) {
      const int kl = k_span - tj * BK;
      if (kl < BK) {
        st.issue_tail(dst, kl, wave, lane);
        return;
      }
    }
    st.issue_full(dst, wave);
  };

  f32x4 acc[2][2][4][2];
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) acc[q >> 1][q & 1][a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 bias4[2][2];                                        // the epilogue's bias vectors: their latency hides behind the main loop
  load_bias_8p<EPI>(P, n0, wc, lane, bias4);
  // bias gradient of the TN form: column sums of A over k, on the vector ALU beside the MFMAs (tile column 0 only).
  // The four waves of a wave row hold the SAME A fragments; wave column wc sums fragment row block a = wc (rows
  // 64 wr + 16 wc + [0, 16) of each half): lane (i, g) adds the 8 k values it holds of row i with four v_dot2c against
  // bf16 ones (exact products, fp32 sums) -- two registers per wave and 8 instead of 32 dot instructions per phase and
  // wave (all of them on wave column 0 stretched phases 1 and 3 of every tile of column 0 by half).
  const bool do_colsum = COLSUM && P.colsum != nullptr && tile_n == 0;
  float csum[2] = {0.f, 0.f};
  bf16x8 aq[2][4], bq0[2][2], bq1[2][2];

  auto read_a = [&](const unsigned char *half) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int a = 0; a < 4; ++a) aq[ks][a] = read_frag<128, ATR>(half, 64 * wr + 16 * a, ks, lane);
  };
  auto read_b = [&](const unsigned char *half, bf16x8 (&bq)[2][2]) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int b = 0; b < 2; ++b) bq[ks][b] = read_frag<128, BTR>(half, 32 * wc + 16 * b, ks, lane);
  };
  auto mfma16 = [&](f32x4 (&c)[4][2], const bf16x8 (&bq)[2][2]) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) c[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bq[ks][b], aq[ks][a], c[a][b], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
  };
  auto colsum8 = [&](float &cs) {
    if constexpr (COLSUM) {
      if (do_colsum) {
        typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
        const bf16x2 ones = __builtin_bit_cast(bf16x2, 0x3F803F80u);
        auto add8 = [&](const bf16x8 f) {
          cs = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(f, f, 0, 1), ones, cs, false);
          cs = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(f, f, 2, 3), ones, cs, false);
          cs = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(f, f, 4, 5), ones, cs, false);
          cs = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(f, f, 6, 7), ones, cs, false);
        };
        // (wc is wave-uniform: scalar branches, no selects)
        if (wc == 0) { add8(aq[0][0]); add8(aq[1][0]); }
        else if (wc == 1) { add8(aq[0][1]); add8(aq[1][1]); }
        else if (wc == 2) { add8(aq[0][2]); add8(aq[1][2]); }
        else { add8(aq[0][3]); add8(aq[1][3]); }
      }
    }
  };

  if (nst > 0) {
    unsigned char *cur = smem, *oth = smem + BUF;
    // prologue: tile 0 complete in buffer 0, the first three halves of tile 1 in flight into buffer 1
    issue(sb0, cur + OFF_B0, 0);
    issue(sa0, cur + OFF_A0, 0);
    issue(sb1, cur + OFF_B1, 0);
    issue(sa1, cur + OFF_A1, 0);
    if (nst > 1) {
      issue(sb0, oth + OFF_B0, 1);
      issue(sa0, oth + OFF_A0, 1);
      issue(sb1, oth + OFF_B1, 1);
      wait_vmcnt<6>();
    } else {
      wait_vmcnt<0>();
    }
    __builtin_amdgcn_s_barrier();
    GPS_TRACE(P, 1);
    if (wr == 1) __builtin_amdgcn_s_barrier();               // group 1 runs one barrier behind

#if GPS_GEMM_8P_PHASES == 2
    // TWO phases of 32 MFMAs per K tile: [B0 B1 A0 reads | copies] barrier [C00 C01] barrier, [A1 reads | copies | vmcnt]
    // barrier [C11 C10] barrier.  A read segment with transposing reads costs 450 - 600 cycles whatever it holds
    // (profiles/r6/gemm_sched_ab.txt): behind a 16-MFMA segment of the other group (290 cycles) it is exposed, behind a
    // 32-MFMA one (580) it is not -- and a K tile takes 4 barriers instead of 8.  Same half-tile protocol as the
    // four-phase walk with phases (1, 2) -> A and (3, 4) -> B: A1(t + 1) is requested in phase A, B0 A0 B1 of tile t + 2 in
    // phase B (all three halves were last read in phase A), and each phase waits only for the halves the NEXT phase
    // reads (a wait for "all of tile t + 1" in phase B gave A1(t + 1) one phase to arrive: the single-problem TN loop
    // stalled 700 cycles per K tile there); every wave retires its fragment reads BEFORE the first
    // barrier of a phase (group 1's first barrier is group 0's second: what group 0 overwrites after it must have been
    // read by group 1 before it).
#pragma clang loop unroll(disable)
    for (int t = 0; t < nst; ++t) {
      const bool n1 = t + 1 < nst, n2 = t + 2 < nst;
      if (t < 24) GPS_TRACE(P, 8 + t);            // slots 8..31; 32.. hold the phase stamps of K tile 8
      // ---- phase A: C00, C01 ----
      GPS_PTRACE(P, t, 0);
      read_b(cur + OFF_B0, bq0);
      read_b(cur + OFF_B1, bq1);
      read_a(cur + OFF_A0);
      // copies in flight, oldest first: A1(t) | B0 A0 B1 (t + 1) | A1(t + 1) | B0 A0 B1 (t + 2) | ...  Each wait lets the four
      // newest halves (8 instructions) stay in flight, so every half has a whole K tile to land: here A1(t), read in
      // phase B, must be in; in phase B the three halves of tile t + 1
      if (n1) {
        issue(sa1, oth + OFF_A1, t + 1);
        wait_vmcnt<8>();
      } else {
        wait_vmcnt<0>();
      }
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      GPS_PTRACE(P, t, 1);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      GPS_PTRACE(P, t, 2);
      mfma16(acc[0][0], bq0);
      mfma16(acc[0][1], bq1);
      colsum8(csum[0]);
      __builtin_amdgcn_sched_barrier(0);
      GPS_PTRACE(P, t, 3);
      __builtin_amdgcn_s_barrier();
      GPS_PTRACE(P, t, 4); GPS_PTRACE(P, t, 5); GPS_PTRACE(P, t, 6); GPS_PTRACE(P, t, 7);
      // ---- phase B: C11, C10 ----
      GPS_PTRACE(P, t, 8);
      read_a(cur + OFF_A1);
      if (n2) {
        issue(sb0, cur + OFF_B0, t + 2);
        issue(sa0, cur + OFF_A0, t + 2);
        issue(sb1, cur + OFF_B1, t + 2);
        wait_vmcnt<8>();
      } else if (n1) {
        wait_vmcnt<2>();                                      // only A1(t + 1) is newer than the halves phase A reads next
      } else {
        wait_vmcnt<0>();
      }
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      GPS_PTRACE(P, t, 9);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      GPS_PTRACE(P, t, 10);
      mfma16(acc[1][1], bq1);
      mfma16(acc[1][0], bq0);
      colsum8(csum[1]);
      __builtin_amdgcn_sched_barrier(0);
      GPS_PTRACE(P, t, 11);
      __builtin_amdgcn_s_barrier();
      GPS_PTRACE(P, t, 12); GPS_PTRACE(P, t, 13); GPS_PTRACE(P, t, 14); GPS_PTRACE(P, t, 15); GPS_PTRACE(P, t, 16);
      unsigned char *tmp = cur;
      cur = oth;
      oth = tmp;
    }
#else
#pragma clang loop unroll(disable)
    for (int t = 0; t < nst; ++t) {
      const bool n1 = t + 1 < nst, n2 = t + 2 < nst;
      if (t < 24) GPS_TRACE(P, 8 + t);            // slots 8..31; 32.. hold the phase stamps of K tile 8
      // ---- phase 1: C00 ----
      GPS_PTRACE(P, t, 0);
      read_b(cur + OFF_B0, bq0);
      __builtin_amdgcn_sched_barrier(0);
      // transposing B reads (NN, TN): a read segment that holds any of them costs ~450 - 600 cycles whether it holds 8 or
      // 32 (profiles/r6/gemm_sched_ab.txt), one without ~230: B1 (phase 2's operand; its registers are free since phase 3
      // of the previous tile, its half-tile landed with that tile's phase-4 wait) is read HERE, phase 2 reads nothing
      if constexpr (BTR) {
        read_b(cur + OFF_B1, bq1);
        __builtin_amdgcn_sched_barrier(0);
      }
      read_a(cur + OFF_A0);
      if (n1) issue(sa1, oth + OFF_A1, t + 1);
      __builtin_amdgcn_sched_barrier(0);
      // the B0 reads (issued first) are done: B0 may be refilled one phase from now
      if constexpr (ATR) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (tr reads: two per fragment; wait for all)
      else if constexpr (BTR) asm volatile("s_waitcnt lgkmcnt(15)" ::: "memory");  // 8 + 8 tr reads of B0, B1, then 8 of A0: >= 9 retired
      else asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
      GPS_PTRACE(P, t, 1);
      __builtin_amdgcn_s_barrier();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      GPS_PTRACE(P, t, 2);
      mfma16(acc[0][0], bq0);
      colsum8(csum[0]);
      __builtin_amdgcn_sched_barrier(0);
      GPS_PTRACE(P, t, 3);
      __builtin_amdgcn_s_barrier();
      // ---- phase 2: C01 ----
      GPS_PTRACE(P, t, 4);
      if constexpr (!BTR) read_b(cur + OFF_B1, bq1);
      if (n2) issue(sb0, cur + OFF_B0, t + 2);
      __builtin_amdgcn_sched_barrier(0);
      GPS_PTRACE(P, t, 5);
      __builtin_amdgcn_s_barrier();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      GPS_PTRACE(P, t, 6);
      mfma16(acc[0][1], bq1);
      __builtin_amdgcn_sched_barrier(0);
      GPS_PTRACE(P, t, 7);
      __builtin_amdgcn_s_barrier();
      // ---- phase 3: C11 ----
      GPS_PTRACE(P, t, 8);
      read_a(cur + OFF_A1);
      if (n2) issue(sa0, cur + OFF_A0, t + 2);
      __builtin_amdgcn_sched_barrier(0);
      GPS_PTRACE(P, t, 9);
      __builtin_amdgcn_s_barrier();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      GPS_PTRACE(P, t, 10);
      mfma16(acc[1][1], bq1);
      colsum8(csum[1]);
      __builtin_amdgcn_sched_barrier(0);
      GPS_PTRACE(P, t, 11);
      __builtin_amdgcn_s_barrier();
      // ---- phase 4: C10 ----
      GPS_PTRACE(P, t, 12);
      if (n2) {
        issue(sb1, cur + OFF_B1, t + 2);
        wait_vmcnt<6>();
      } else {
        wait_vmcnt<0>();
      }
      __builtin_amdgcn_sched_barrier(0);
      GPS_PTRACE(P, t, 13);
      __builtin_amdgcn_s_barrier();
      GPS_PTRACE(P, t, 14);
      mfma16(acc[1][0], bq0);
      __builtin_amdgcn_sched_barrier(0);
      GPS_PTRACE(P, t, 15);
      __builtin_amdgcn_s_barrier();
      GPS_PTRACE(P, t, 16);
      unsigned char *tmp = cur;
      cur = oth;
      oth = tmp;
    }
#endif
    if (wr == 0) __builtin_amdgcn_s_barrier();
  }

  if constexpr (COLSUM) {
    if (do_colsum) {
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        float v = csum[hf];
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        const int m = m0 + 128 * hf + 64 * wr + 16 * wc + (lane & 15);
        if (lane < 16 && m < P.M) {
          float *cs = P.colsum + (size_t)split * P.M + m;
          *cs = (P.accumulate && P.splits == 1) ? *cs + v : v;
        }
      }
    }
  }
  GPS_TRACE(P, 2);
  store_quads<EPI>(P, acc, split, m0, n0, wr, wc, lane, bias4);
  GPS_TRACE(P, 3);
  GPS_TRACE_DRAIN();
  GPS_TRACE(P, 6);
  GPS_TRACE(P, 5);
}

template <bool ATR, bool BTR, int EPI>
int launch_8p(Params &P, hipStream_t s) {
  constexpr int LDS = 2 * 4 * 128 * BK * 2;              // two buffers of four half-tile images = 128 KB
  P.ntm = (P.M + 255) / 256;
  P.ntn = (P.N + 255) / 256;
  P.gm = 4;
  auto kern = &gemm8p_kernel<ATR, BTR, EPI>;
  static PerDeviceFlag granted;
  if (!grant_dynamic_lds(kern, LDS, granted)) return GPS_ERR_LAUNCH;
  const long long blocks = (long long)P.ntm * P.ntn * P.splits;
  if (blocks <= 0 || blocks > 0x7FFFFFFFLL) return GPS_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(512), LDS, s, P);
  return hipGetLastError() == hipSuccess ? GPS_OK : GPS_ERR_LAUNCH;
}

// ---------------------------------------------------------------------------------------------------------
// Grouped forward / input-gradient launches: up to kGroupMax INDEPENDENT products of one form and epilogue (each with its own
// operands, shape, bias, saved activations and device-side row extent) as ONE launch of the two-group 256 x 256 kernel over
// the union of their output tiles.
//
// Why: the text stack (12 608 live rows) and the object stack (5 120 rows) of the GPS model are independent until the
// joint layers (reference model/openvocab.py:41-63) and their layers have the same op sequence; one at a time the 768-wide
// products are 150 and 60 tiles of 256 x 256 on 256 CUs -- two launches that leave 41 % and 77 % of the chip idle -- together
// 210 tiles, one pass.  (A second HIP stream or graph branch would also overlap them, but forked graphs are what the
// allocator's stream-ordered reuse cannot survive on this ROCm: HISTORY 9a.)
// Tiles are enumerated problem by problem in the order given (the host puts the longest reduction first); the blocks of an
// XCD take a contiguous range of that list, so an XCD mostly works on ONE problem's panels.
// ---------------------------------------------------------------------------------------------------------
constexpr int kGroupMax = 4;
struct GroupParams {
  Params p[kGroupMax];
  int n;
};

template <bool BTR, int EPI, bool RAGGED>
__global__ __launch_bounds__(512, 2) void gemm8p_grouped_kernel(const GroupParams G) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // 2 * BUF = 128 KB
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // live tiles of every problem (device-side row extents) and this block's place in their concatenation
  int ntm[kGroupMax], tiles[kGroupMax], total = 0;
#pragma unroll
  for (int j = 0; j < kGroupMax; ++j) {
    ntm[j] = 0;
    tiles[j] = 0;
    if (j < G.n) {
      ntm[j] = G.p[j].ntm;
      if (G.p[j].extent_dev) ntm[j] = min(ntm[j], max(0, (*G.p[j].extent_dev + 255) / 256));
      tiles[j] = ntm[j] * G.p[j].ntn;
      total += tiles[j];
    }
  }
  if ((int)blockIdx.x >= total) return;
  int tile = xcd_virtual_id(blockIdx.x, total), j = 0;
#pragma unroll
  for (int q = 0; q < kGroupMax - 1; ++q)
    if (j == q && tile >= tiles[q]) {
      tile -= tiles[q];
      j = q + 1;
    }
  // the problem's record, selected on the scalar unit (j is wave-uniform: pointers stay in SGPRs, no waterfall loops)
  Params P = G.p[0];
  int ntm_j = ntm[0];
#pragma unroll
  for (int q = 1; q < kGroupMax; ++q)
    if (j == q) {
      P = G.p[q];
      ntm_j = ntm[q];
    }
  const int GM = 4;
  const int group = tile / (GM * P.ntn), in_group = tile - group * (GM * P.ntn);
  const int gmr = min(GM, ntm_j - group * GM);
  const int tile_n = in_group / gmr;
  const int m0 = (group * GM + (in_group - tile_n * gmr)) * 256, n0 = tile_n * 256;
  gemm8p_tile<false, BTR, EPI, RAGGED>(P, smem, lane, wave, __builtin_amdgcn_readfirstlane(m0), __builtin_amdgcn_readfirstlane(n0),
                                       __builtin_amdgcn_readfirstlane(tile_n), 0, 0, P.nkt, P.K);
}

template <bool BTR, int EPI, bool RAGGED>
int launch_grouped(GroupParams &G, hipStream_t s) {
  constexpr int LDS = 2 * 4 * 128 * BK * 2;
  auto kern = &gemm8p_grouped_kernel<BTR, EPI, RAGGED>;
  static PerDeviceFlag granted;
  if (!grant_dynamic_lds(kern, LDS, granted)) return GPS_ERR_LAUNCH;
  long long blocks = 0;
  for (int j = 0; j < G.n; ++j) {
    G.p[j].ntm = (G.p[j].M + 255) / 256;
    G.p[j].ntn = (G.p[j].N + 255) / 256;
    G.p[j].gm = 4;
    blocks += (long long)G.p[j].ntm * G.p[j].ntn;
  }
  if (blocks <= 0 || blocks > 0x7FFFFFFFLL) return GPS_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(512), LDS, s, G);
  return hipGetLastError() == hipSuccess ? GPS_OK : GPS_ERR_LAUNCH;
}
template <bool BTR, int EPI>
int launch_grouped_any(GroupParams &G, bool ragged, hipStream_t s) {
  return ragged ? launch_grouped<BTR, EPI, true>(G, s) : launch_grouped<BTR, EPI, false>(G, s);
}

// ---------------------------------------------------------------------------------------------------------
// Stream-K form of the two-group 256 x 256 kernel (forms NT / NN, every bf16 epilogue): variant 13.
//
// Why: one workgroup per CU and whole tiles means the launch runs in ROUNDS -- 450 tiles (12 608 x 2 304) take two
// rounds of 256 although they are 1.76 rounds of work, 150 tiles (12 608 x 768, K = 2 304) leave 106 CUs idle for the
// whole launch, 264 tiles (8 320 x 2 048) run a second round for 8 tiles -- and every CU of a round reaches its
// epilogue at the same moment: 33 MB of stores hit HBM together while the matrix pipes wait
// (profiles/r6/gemm_probe_trace_a.jsonl: main loop 16.2 us, epilogue 4.4 us, prologue 1.8 us per tile and CU).
// Here the K tiles of ALL output tiles form one sequence (tile-major, tile_n fastest); the launch has one resident
// workgroup per CU and workgroup position w takes the w-th equal share of that sequence, whatever tile boundaries it
// crosses.  A share is a few SEGMENTS (K-tile runs inside one output tile):
//   * a segment that covers its tile's whole reduction ends in the ordinary epilogue;
//   * a segment that starts behind k = 0 (always the FIRST segment of a share) is a contribution: the fp32
//     accumulators go to this position's slab with write-through (sc1) stores and the position's arrival word is set;
//   * a segment that starts at k = 0 but ends early (always the LAST segment of a share) owns the tile: it adds the
//     slabs of the following positions whose shares begin inside this tile (in position order: deterministic),
//     then runs the epilogue.  Contributions are computed at the very start of a launch and consumed at the very
//     end of another workgroup's share, so owners practically never wait; contributors never wait at all (no deadlock
//     whatever the dispatch order; the spin is bounded all the same and reports through the error word).
// Slab hand-off as the guide prescribes for large payloads: sc1 stores -> every wave s_waitcnt vmcnt(0) -> barrier ->
// one relaxed agent-scope flag store; the owner polls relaxed from one lane, then every lane reads with sc1 loads.
// Tile boundaries fall at different times on different CUs, so epilogue stores no longer arrive in bursts.
// ---------------------------------------------------------------------------------------------------------
constexpr int kSkSlabFloats = 256 * 256;          // one workgroup's accumulators
constexpr int kSkMaxGrid = 256;
constexpr int kSkFlagBytes = 4096;                 // kSkMaxGrid arrival words + the error word, padded
constexpr unsigned int kSkSpinLimit = 1u << 21;

// The kernel is ONE loop over the K tiles of the share.  The copies of K tile g + 1 / g + 2 are issued while K tile g is
// computed (gemm8p_tile's schedule) -- also when those K tiles belong to the NEXT output tile: the stager of each half-tile
// is re-aimed when the K tile it is about to copy is the first of a tile.  A segment's ending (epilogue, or the slab
// stores of a contribution) therefore runs with the next segment's first operands already on their way: no per-tile
// prologue after the first one.  At a segment end the two wave groups align on one barrier (group 0 waits one barrier for
// group 1), run the ending side by side, and group 1 takes its one-barrier lag back.
// RAGGED: K % 64 != 0 -- the last K tile of every output tile takes the zero-filling copy path; kept out of the K % 64 == 0
// instantiations because its mere presence in the loop costs 10 % (gemm_probe trace, 36 K tiles: 61.6 vs 56.0 us).
template <bool BTR, int EPI, bool RAGGED>
__global__ __launch_bounds__(512, 2) void gemm8p_sk_kernel(const Params P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // 2 * BUF = 128 KB
  constexpr int HALF = 128 * BK * 2, BUF = 4 * HALF;
  constexpr int OFF_A0 = 0, OFF_A1 = HALF, OFF_B0 = 2 * HALF, OFF_B1 = 3 * HALF;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int G = (int)gridDim.x;
  const int w = xcd_virtual_id(blockIdx.x, G);                  // position in the K-tile sequence (contiguous per XCD)
  int ntm = P.ntm;
  if (P.extent_dev) ntm = min(P.ntm, max(0, (*P.extent_dev + 255) / 256));   // live tile rows only
  const int nkt = P.nkt, ntn = P.ntn;
  const int U = ntm * ntn * nkt;                                // K-tile units of the launch
  const int per = U / G, rem = U - per * G;
  auto share_begin = [&](int pos) { return pos * per + min(pos, rem); };
  const int u0 = share_begin(w);
  const int n_units = share_begin(w + 1) - u0;
  GPS_TRACE(P, 0);
  GPS_TRACE(P, 4);
  if (n_units <= 0) return;
#if defined(__HIP_DEVICE_COMPILE__)
  const __amdgpu_buffer_rsrc_t rS = __builtin_amdgcn_make_buffer_rsrc(P.partial, 0, (unsigned int)G * kSkSlabFloats * 4u, 0x00020000);

  // (tile row, tile column, K tile) of a unit; tile_n fastest, then K
  struct Unit { int tm, tn, kt; };
  auto next_unit = [&](const Unit &d) {
    Unit r = d;
    if (++r.kt == nkt) {
      r.kt = 0;
      if (++r.tn == ntn) { r.tn = 0; ++r.tm; }
    }
    return r;
  };
  Unit d0;                                                      // the unit being computed
  {
    const int T = __builtin_amdgcn_readfirstlane(u0 / nkt);
    d0.kt = u0 - T * nkt;
    d0.tm = __builtin_amdgcn_readfirstlane(T / ntn);
    d0.tn = T - d0.tm * ntn;
  }
  const int first_kt = d0.kt;                                   // > 0: the share starts with a contribution
  Unit d1 = next_unit(d0), d2 = next_unit(d1);                  // the units whose copies are issued during d0

  Stager<128, false, 8> sa0, sa1;
  Stager<128, BTR, 8> sb0, sb1;
  // half-tile `half` (0 / 1) of operand A or B of unit d into dst; the stager is re-aimed at the first K tile of a tile
  auto issue_a = [&](Stager<128, false, 8> &st, int half, unsigned char *dst, const Unit &d, bool aim) {
    if (aim || d.kt == 0) st.init(P.A, P.lda, P.M, d.tm * 256 + 128 * half, d.kt * BK, wave, lane);
    if constexpr (RAGGED) {
      const int kl = P.K - d.kt * BK;
      if (kl < BK) {
        st.issue_tail(dst, kl, wave, lane);
        return;
      }
    }
    st.issue_full(dst, wave);
  };
  auto issue_b = [&](Stager<128, BTR, 8> &st, int half, unsigned char *dst, const Unit &d, bool aim) {
    if (aim || d.kt == 0) st.init(P.B, P.ldb, P.N, d.tn * 256 + 128 * half, d.kt * BK, wave, lane);
    if constexpr (RAGGED) {
      const int kl = P.K - d.kt * BK;
      if (kl < BK) {
        st.issue_tail(dst, kl, wave, lane);
        return;
      }
    }
    st.issue_full(dst, wave);
  };

  f32x4 acc[2][2][4][2];
  f32x4 bias4[2][2];
  bf16x8 aq[2][4], bq0[2][2], bq1[2][2];
  auto read_a = [&](const unsigned char *half) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int a = 0; a < 4; ++a) aq[ks][a] = read_frag<128, false>(half, 64 * wr + 16 * a, ks, lane);
  };
  auto read_b = [&](const unsigned char *half, bf16x8 (&bq)[2][2]) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int b = 0; b < 2; ++b) bq[ks][b] = read_frag<128, BTR>(half, 32 * wc + 16 * b, ks, lane);
  };
  auto mfma16 = [&](f32x4 (&c)[4][2], const bf16x8 (&bq)[2][2]) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) c[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bq[ks][b], aq[ks][a], c[a][b], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
  };

  unsigned char *cur = smem, *oth = smem + BUF;
  // prologue: unit 0 complete in buffer 0, the first three halves of unit 1 in flight into buffer 1
  issue_b(sb0, 0, cur + OFF_B0, d0, true);
  issue_a(sa0, 0, cur + OFF_A0, d0, true);
  issue_b(sb1, 1, cur + OFF_B1, d0, true);
  issue_a(sa1, 1, cur + OFF_A1, d0, true);
  if (n_units > 1) {
    issue_b(sb0, 0, oth + OFF_B0, d1, false);
    issue_a(sa0, 0, oth + OFF_A0, d1, false);
    issue_b(sb1, 1, oth + OFF_B1, d1, false);
    wait_vmcnt<6>();
  } else {
    wait_vmcnt<0>();
  }
  __builtin_amdgcn_s_barrier();
  GPS_TRACE(P, 1);
  if (wr == 1) __builtin_amdgcn_s_barrier();                   // group 1 runs one barrier behind

  bool seg_first = true;            // d0 is the first K tile of a segment
  bool owner_tail = false;
  Unit d_last = d0;
  int pending = 0;                  // 1: slab stores issued, arrival word not yet set
  int seg_no = 0;
#pragma clang loop unroll(disable)
  for (int g = 0; g < n_units; ++g) {
    const bool n1 = g + 1 < n_units, n2 = g + 2 < n_units;
    if (seg_first) {
      GPS_TRACE(P, 8 + 6 * seg_no);
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b) acc[q >> 1][q & 1][a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
      seg_first = false;
    }
    // ---- phase 1: C00 ----
    read_b(cur + OFF_B0, bq0);
    __builtin_amdgcn_sched_barrier(0);
    read_a(cur + OFF_A0);
    if (n1) issue_a(sa1, 1, oth + OFF_A1, d1, false);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (BTR) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (tr reads: two per fragment; wait for all)
    else asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    mfma16(acc[0][0], bq0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    // ---- phase 2: C01 ----
    read_b(cur + OFF_B1, bq1);
    if (n2) issue_b(sb0, 0, cur + OFF_B0, d2, false);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    mfma16(acc[0][1], bq1);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    // ---- phase 3: C11 ----
    read_a(cur + OFF_A1);
    if (n2) issue_a(sa0, 0, cur + OFF_A0, d2, false);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    mfma16(acc[1][1], bq1);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    // ---- phase 4: C10 ----
    if (n2) issue_b(sb1, 1, cur + OFF_B1, d2, false);
    // a contribution's slab stores (issued at the end of the previous K tile) must be acknowledged before its arrival word
    // is set: this one wait is a full drain, and the word is set behind the barrier after next (by then group 1's waves,
    // one barrier behind, have drained theirs as well)
    if (pending == 1 || !n2) wait_vmcnt<0>();
    else wait_vmcnt<6>();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    mfma16(acc[1][0], bq0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    const bool seg_last = d0.kt == nkt - 1 || !n1;
    if (pending == 1 && !seg_last) {
      if (threadIdx.x == 0) __hip_atomic_store(&P.sk_flags[w], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      pending = 0;
    }
    if (seg_last) {
      GPS_TRACE(P, 8 + 6 * seg_no + 2);
      if (wr == 0) __builtin_amdgcn_s_barrier();               // both groups side by side through the ending
      const bool contribution = seg_no == 0 && first_kt > 0;
      const int m0 = d0.tm * 256, n0 = d0.tn * 256;
      // (offsets derived from the lane index are recomputed here, not carried through the main loop)
      int lane_e = lane;
      asm volatile("" : "+v"(lane_e));
      const unsigned int lane_off = (unsigned int)(wave * 32 * 64 + lane_e) * 16u;     // + r * 1024: this lane's register r in a slab
      GPS_TRACE_VAL(P, 8 + 6 * seg_no + 5, (unsigned long long)(contribution ? 1 : d0.kt != nkt - 1 ? 2 : 0));
      if (contribution) {
        if (pending == 1) {          // (cannot happen: a share holds one contribution; kept for the invariant's sake)
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __syncthreads();
          if (threadIdx.x == 0) __hip_atomic_store(&P.sk_flags[w], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        const unsigned int so = (unsigned int)w * (kSkSlabFloats * 4u);
#pragma unroll
        for (int r = 0; r < 32; ++r)
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[r >> 4][(r >> 3) & 1][(r >> 1) & 3][r & 1]), rS,
                                                 lane_off + (unsigned int)r * 1024u, so, 16);
        pending = 1;
      } else if (d0.kt == nkt - 1) {
        // (the bias is fetched here, not ahead of the main loop as in gemm8p_tile: beside the LDS-DMA copies in flight the
        // compiler drains vmcnt to 0 in front of any ordinary load, i.e. the whole prefetch ring once per tile: +0.6 us)
        load_bias_8p<EPI>(P, n0, wc, lane_e, bias4);
        store_quads<EPI>(P, acc, 0, m0, n0, wr, wc, lane_e, bias4);
      } else {
        owner_tail = true;          // the owner of a split tile: finished behind the loop (this is the share's last K tile)
      }
      GPS_TRACE(P, 8 + 6 * seg_no + 3);
      GPS_TRACE_VAL(P, 8 + 6 * seg_no + 4, (unsigned long long)g);
      ++seg_no;
      seg_first = true;
      if (wr == 1 && n1) __builtin_amdgcn_s_barrier();         // group 1 one barrier behind again
    }
    d_last = d0;
    d0 = d1;
    d1 = d2;
    d2 = next_unit(d2);
    unsigned char *tmp = cur;
    cur = oth;
    oth = tmp;
  }
  if (pending == 1) {                // the share ended on (or one K tile after) its contribution; set the word BEFORE
                                     // waiting for anybody else's (no chains of owners waiting on owners)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(&P.sk_flags[w], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (owner_tail) {
    // ---- owner of a split tile (always the last segment of a share; outside the loop so that its registers are not
    // live across the main loop): add the slabs of the positions whose shares begin inside this tile, in position order.
    // Eight registers at a time: the runtime loop over contributors then carries 32 values, not the 128 accumulators
    // (carried whole, entry and back-edge copies get different registers).
    const Unit dl = d_last;
    int lane_e = lane;
    asm volatile("" : "+v"(lane_e));
    const unsigned int lane_off = (unsigned int)(wave * 32 * 64 + lane_e) * 16u;
    const int tile_end = ((dl.tm * ntn + dl.tn) + 1) * nkt;
    int c_end = w + 1;
    while (c_end < G && share_begin(c_end) < tile_end && share_begin(c_end) < U) ++c_end;
    if (threadIdx.x == 0) {
      for (int c = w + 1; c < c_end; ++c) {
        unsigned int spins = 0;
        while (__hip_atomic_load(&P.sk_flags[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
          __builtin_amdgcn_s_sleep(4);
          if (++spins > kSkSpinLimit) {                        // never expected: report, do not hang
            __hip_atomic_store(&P.sk_flags[kSkMaxGrid], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
          }
        }
        __hip_atomic_store(&P.sk_flags[c], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
      }
    }
    __syncthreads();
    GPS_TRACE(P, 8 + 6 * (seg_no - 1) + 1);
#pragma unroll
    for (int r0 = 0; r0 < 32; r0 += 8) {
      f32x4 v[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) v[r] = f32x4{0.f, 0.f, 0.f, 0.f};
      for (int c = w + 1; c < c_end; ++c) {
        const unsigned int so = (unsigned int)c * (kSkSlabFloats * 4u);
        f32x4 part[8];
#pragma unroll
        for (int r = 0; r < 8; ++r)
          part[r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rS, lane_off + (unsigned int)(r0 + r) * 1024u, so, 16));
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] += part[r];
      }
      // (the adds are pinned here: sunk behind the other chunks' loops, all four chunks' sums would be live at once)
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        f32x4 &dst = acc[(r0 + r) >> 4][((r0 + r) >> 3) & 1][((r0 + r) >> 1) & 3][(r0 + r) & 1];
        dst += v[r];
        asm volatile("" : "+v"(dst));
      }
    }
    load_bias_8p<EPI>(P, dl.tn * 256, wc, lane_e, bias4);
    store_quads<EPI>(P, acc, 0, dl.tm * 256, dl.tn * 256, wr, wc, lane_e, bias4);
    GPS_TRACE(P, 8 + 6 * (seg_no - 1) + 3);
  }
#endif
  GPS_TRACE_DRAIN();
  GPS_TRACE(P, 6);
  GPS_TRACE(P, 5);
}

template <bool BTR, int EPI, bool RAGGED>
int launch_sk_r(Params &P, hipStream_t s);
template <bool BTR, int EPI>
int launch_sk(Params &P, hipStream_t s) {
  return (P.K % BK) ? launch_sk_r<BTR, EPI, true>(P, s) : launch_sk_r<BTR, EPI, false>(P, s);
}
template <bool BTR, int EPI, bool RAGGED>
int launch_sk_r(Params &P, hipStream_t s) {
  constexpr int LDS = 2 * 4 * 128 * BK * 2;
  P.ntm = (P.M + 255) / 256;
  P.ntn = (P.N + 255) / 256;
  auto kern = &gemm8p_sk_kernel<BTR, EPI, RAGGED>;
  static PerDeviceFlag granted;
  if (!grant_dynamic_lds(kern, LDS, granted)) return GPS_ERR_LAUNCH;
  int n_cu = device_cu_count();
  if (n_cu <= 0) return GPS_ERR_LAUNCH;
  if (n_cu > kSkMaxGrid) n_cu = kSkMaxGrid;
  const long long units = (long long)P.ntm * P.ntn * P.nkt;
  if (units <= 0 || units > 0x3FFFFFFFLL) return GPS_ERR_UNSUPPORTED;
  static const int grid_env = [] { const char *e = getenv("GPS_GEMM_SK_GRID"); return e ? atoi(e) : 0; }();
  static const int min_units = [] { const char *e = getenv("GPS_GEMM_SK_MIN_UNITS"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 4; }();
  long long grid = units / min_units;                     // every share keeps >= min_units K tiles
  if (grid > n_cu) grid = n_cu;
  if (grid_env > 0 && grid_env < grid) grid = grid_env;
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), LDS, s, P);
  return hipGetLastError() == hipSuccess ? GPS_OK : GPS_ERR_LAUNCH;
}

// ---------------------------------------------------------------------------------------------------------
// Grouped weight gradients: MANY dW_p (M_p x N_p) [+]= dY_p^T X_p in ONE launch, no split over K.
//
// A single weight gradient of this model has a few dozen output tiles and a reduction over 5 000 - 22 000 token rows:
// to fill 256 CUs it must be split over K, and every split dumps its fp32 accumulators (one 256 x 256 or two 128 x 128
// tiles per CU = 33 MB per GEMM, written and read again by splitk_reduce_kernel) -- for the 768 x 768 ... 768 x 2 376
// gradients of the object / joint layers that round trip costs as much as their MFMAs.  The gradients of a backward pass
// are independent of each other and of the input-gradient chain, so the host defers them (modules/layers/gemm.py
// grouped_wgrads) and issues them together: ~1 100 tiles of 256 x 256 at the bench workload, each walked over its WHOLE
// reduction by one workgroup with gemm8p_tile -- the long-K regime where that schedule runs at library speed -- and
// written once, straight into the parameter's .grad (plain store, or read-modify-write when the buffer already holds
// a gradient: the flat buffer of the data-parallel step).  Deterministic: every output element has exactly one writer
// and one summation order.
//
// Tiles are enumerated problem by problem in the order given (the host sorts by decreasing reduction length).  The
// min(tiles, CUs) persistent workgroups take their first tile statically (the workgroups of an XCD take consecutive
// tiles: tiles of one output row share their dY panel in that XCD's L2) and every later one from an atomic counter --
// longest first, whoever is free takes the next: the greedy schedule keeps the makespan within one short tile of the mean.  The problem table lives in device memory (written by
// wgrad_table_write_kernel from kernel arguments, 32 records per launch: capturable, no host buffer to keep alive).
// ---------------------------------------------------------------------------------------------------------
struct WgradRec {          // 64 bytes
  const uint16_t *A, *B;
  float *C, *colsum;
  const int *extent;
  int lda, ldb, ldc, M, N, K;
  int tile0;               // id of this problem's first tile
  int flags;               // bit 0: accumulate
};
constexpr int kWgradMaxProblems = 256, kWgradSlots = 4, kWgradChunk = 32, kWgradQueues = 8;
__device__ WgradRec g_wgrad_table[kWgradSlots][kWgradMaxProblems];
__device__ unsigned int g_wgrad_next[kWgradSlots];       // next tile id to hand out (dynamic part of the schedule)
// [r4] XCD-local queues: the problems are dealt to one queue per XCD (tiles of a problem stay together, every class of
// reduction length is spread evenly), a workgroup takes tiles from the queue of the XCD it runs on (blockIdx mod 8) and
// only when that one is empty from the others'.  Tiles of one problem then run side by side on ONE XCD and, having the
// same reduction length, keep walking K together: their operand panels are fetched once per XCD, not once per tile.
struct WgradQueueState {
  int begin[kWgradQueues], end[kWgradQueues], first_problem[kWgradQueues];   // tile ids / first record of every queue
  unsigned int next[kWgradQueues];
  int enabled;
};
__device__ WgradQueueState g_wgrad_queues[kWgradSlots];
struct WgradChunkArgs { WgradRec r[kWgradChunk]; };
struct WgradQueueArgs { int begin[kWgradQueues], end[kWgradQueues], first_problem[kWgradQueues], enabled; };

__global__ __launch_bounds__(64) void wgrad_table_write_kernel(const WgradChunkArgs c, int slot, int offset, int count,
                                                               int first_dynamic_tile, const WgradQueueArgs q, int grid) {
  if ((int)threadIdx.x < count) g_wgrad_table[slot][offset + threadIdx.x] = c.r[threadIdx.x];
  if (offset == 0 && threadIdx.x == 0) g_wgrad_next[slot] = (unsigned int)first_dynamic_tile;
  if (offset == 0 && threadIdx.x < kWgradQueues) {
    const int x = threadIdx.x;
    WgradQueueState &st = g_wgrad_queues[slot];
    st.begin[x] = q.begin[x];
    st.end[x] = q.end[x];
    st.first_problem[x] = q.first_problem[x];
    // the workgroups blockIdx = x, x + 8, ... take the queue's first tiles statically
    const int n_static = grid > x ? (grid - x + kWgradQueues - 1) / kWgradQueues : 0;
    st.next[x] = (unsigned int)(q.begin[x] + min(n_static, q.end[x] - q.begin[x]));
    if (x == 0) st.enabled = q.enabled;
  }
}

#ifdef GPS_GEMM_TRACE
#define GPS_WGRAD_TRACE_PARAM , unsigned long long *trace
#define GPS_WGRAD_TRACE_ARG , g_probe_trace
#else
#define GPS_WGRAD_TRACE_PARAM
#define GPS_WGRAD_TRACE_ARG
#endif
__global__ __launch_bounds__(512, 2) void wgrad_grouped_kernel(int slot, int n_problems, int total_tiles, int xcd_queues GPS_WGRAD_TRACE_PARAM) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // 128 KB (gemm8p_tile)
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  __shared__ int s_next, s_queue;
  const WgradRec *tab = g_wgrad_table[slot];
  WgradQueueState *Q = &g_wgrad_queues[slot];
  const int G = (int)gridDim.x;
  // One queue (xcd_queues = 0).  First tile: static, the workgroups of an XCD take consecutive tiles.  Every later tile:
  // the next one of the longest-first list, handed out by an atomic counter -- whoever finishes first takes it (greedy
  // longest-processing-time schedule; every tile is still computed by exactly one workgroup, so results do not depend on
  // who that is).  XCD-local queues: see WgradQueueState.
  const int my_queue = (int)blockIdx.x & (kWgradQueues - 1);
  unsigned int empty = 0u;                                   // queues this workgroup has seen run dry (thread 0)
  auto take = [&]() {                                        // thread 0: the next tile (or total_tiles) and its queue
    if (!xcd_queues) {
      s_next = (int)atomicAdd(&g_wgrad_next[slot], 1u);
      s_queue = 0;
      return;
    }
    for (int k = 0; k < kWgradQueues; ++k) {
      const int y = (my_queue + k) & (kWgradQueues - 1);
      if ((empty >> y) & 1u) continue;
      const int t = (int)atomicAdd(&Q->next[y], 1u);
      if (t < Q->end[y]) { s_next = t; s_queue = y; return; }
      empty |= 1u << y;
    }
    s_next = total_tiles;
    s_queue = 0;
  };
#ifdef GPS_GEMM_TRACE
  unsigned long long tr_kt = 0, tr_tiles = 0;
  if (trace && threadIdx.x == 0) { trace[(size_t)blockIdx.x * kTraceSlots + 0] = __builtin_amdgcn_s_memtime(); trace[(size_t)blockIdx.x * kTraceSlots + 4] = __builtin_amdgcn_s_memrealtime(); }
#endif
  int t, queue = 0, p = 0;
  if (!xcd_queues) {
    t = xcd_virtual_id(blockIdx.x, G);
  } else {
    const int i = (int)blockIdx.x / kWgradQueues;
    if (i < Q->end[my_queue] - Q->begin[my_queue]) {
      t = Q->begin[my_queue] + i;
      queue = my_queue;
    } else {
      if (threadIdx.x == 0) take();
      __syncthreads();
      t = __builtin_amdgcn_readfirstlane(s_next);
      queue = __builtin_amdgcn_readfirstlane(s_queue);
      __syncthreads();
    }
    p = Q->first_problem[queue];
  }
  for (;;) {
    if (t >= total_tiles) break;
    while (p + 1 < n_problems && tab[p + 1].tile0 <= t) ++p;
    // The record is the same for the whole workgroup, but it arrives through vector loads: every field goes back to
    // scalar registers here.  Left in VGPRs, the operand pointers make the buffer descriptors of the stage copies
    // "divergent" and the compiler wraps EVERY buffer_load ... lds of the main loop in a waterfall loop (4
    // v_readfirstlane + 2 v_cmp + saveexec + branch per copy: 83 instead of 30 vector and 93 instead of 45 scalar
    // instructions per K tile and wave, profiles/r4/pmc_wgrad_grouped_*.txt).
    const WgradRec rec_v = tab[p];
    auto uni = [](int v) { return __builtin_amdgcn_readfirstlane(v); };
    auto uni_ptr = [&](const void *q) {
      const unsigned long long u = reinterpret_cast<unsigned long long>(q);
      const unsigned int lo = (unsigned int)uni((int)(unsigned int)u), hi = (unsigned int)uni((int)(unsigned int)(u >> 32));
      return reinterpret_cast<void *>(((unsigned long long)hi << 32) | lo);
    };
    WgradRec rec;
    rec.A = (const uint16_t *)uni_ptr(rec_v.A); rec.B = (const uint16_t *)uni_ptr(rec_v.B);
    rec.C = (float *)uni_ptr(rec_v.C); rec.colsum = (float *)uni_ptr(rec_v.colsum);
    rec.extent = (const int *)uni_ptr(rec_v.extent);
    rec.lda = uni(rec_v.lda); rec.ldb = uni(rec_v.ldb); rec.ldc = uni(rec_v.ldc);
    rec.M = uni(rec_v.M); rec.N = uni(rec_v.N); rec.K = uni(rec_v.K);
    rec.tile0 = uni(rec_v.tile0); rec.flags = uni(rec_v.flags);
    Params P = {};
    P.M = rec.M; P.N = rec.N; P.K = rec.K;
    P.A = rec.A; P.lda = rec.lda;
    P.B = rec.B; P.ldb = rec.ldb;
    P.C = rec.C; P.ldc = rec.ldc;
    P.colsum = rec.colsum;
    P.splits = 1;
    P.accumulate = rec.flags & 1;
    int k_eff = rec.K;
    if (rec.extent) k_eff = min(rec.K, max(*rec.extent, 0));
    const int ntn = (rec.N + 255) >> 8;
    const int lt = t - rec.tile0;
    // (integer division runs on the vector ALU: back to scalar registers, the tile code branches on these)
    const int tile_m = __builtin_amdgcn_readfirstlane(lt / ntn);
    const int tile_n = lt - tile_m * ntn;
    k_eff = __builtin_amdgcn_readfirstlane(k_eff);
    const int nst = (k_eff + BK - 1) / BK;
    // every scalar load of the record has landed before the tile starts: the tile's counted lgkmcnt waits must only
    // ever see its own LDS traffic
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#ifdef GPS_GEMM_TRACE
    P.trace = trace ? trace + (size_t)512 * kTraceSlots : nullptr;   // the tile's own stamps (phases of its 9th K tile): second region of the probe's buffer
#endif
    gemm8p_tile<true, true, EPI_F32>(P, smem, lane, wave, tile_m * 256, tile_n * 256, tile_n, 0, 0, nst, k_eff);
#ifdef GPS_GEMM_TRACE
    tr_kt += (unsigned long long)nst;
    ++tr_tiles;
    if (trace && threadIdx.x == 0) {
      unsigned long long *tw = trace + (size_t)blockIdx.x * kTraceSlots;
      tw[6] = __builtin_amdgcn_s_memtime(); tw[5] = __builtin_amdgcn_s_memrealtime(); tw[7] = tr_kt; tw[8] = tr_tiles;
    }
#endif
    if (threadIdx.x == 0) take();
    __syncthreads();                                        // s_next visible; the next tile's first copies overwrite the stage buffers
    t = __builtin_amdgcn_readfirstlane(s_next);
    const int nq = __builtin_amdgcn_readfirstlane(s_queue);
    __syncthreads();                                        // (s_next is rewritten at the end of the next tile)
    if (nq != queue) {                                      // a tile of another queue: its records start elsewhere
      queue = nq;
      p = xcd_queues ? Q->first_problem[queue] : p;
    }
  }
}

// out[e] = sum over splits of partial[s][e] in split order (deterministic); the same for the column sums
__global__ __launch_bounds__(256) void splitk_reduce_kernel(int splits, long long elems, const float *__restrict__ partial,
                                                           float *__restrict__ out, long long ldo, int ncols,
                                                           int m_rows, const float *__restrict__ colsum_partial,
                                                           float *__restrict__ colsum_out, int main_blocks,
                                                           const int *__restrict__ row_extent) {
  if ((int)blockIdx.x < main_blocks) {
    const long long e = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (e >= elems) return;
    // NN form with a device-side row extent: rows past it were never computed (their partial tiles hold anything) and
    // nobody reads them -- the masked-LM decoder's input gradient has 3 200 rows of which ~ 480 are labelled
    if (row_extent && e >= (long long)max(*row_extent, 0) * ncols) return;
    // the partial tiles of up to 16 splits are requested together (independent loads in flight, not one memory round
    // trip per split) and then added in split order
    f32x4 v[16];
#pragma unroll
    for (int k = 0; k < 16; ++k)
      v[k] = k < splits ? *reinterpret_cast<const f32x4 *>(partial + (size_t)k * elems + e) : f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 s = v[0];
#pragma unroll
    for (int k = 1; k < 16; ++k) s += v[k];
    for (int k = 16; k < splits; ++k) s += *reinterpret_cast<const f32x4 *>(partial + (size_t)k * elems + e);
    const long long r = e / ncols, c = e - r * ncols;
    *reinterpret_cast<f32x4 *>(out + r * ldo + c) = s;
  } else {
    const int m = ((int)blockIdx.x - main_blocks) * 256 + threadIdx.x;
    if (m >= m_rows) return;
    float w[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) w[k] = k < splits ? colsum_partial[(size_t)k * m_rows + m] : 0.f;
    float s = w[0];
#pragma unroll
    for (int k = 1; k < 16; ++k) s += w[k];
    for (int k = 16; k < splits; ++k) s += colsum_partial[(size_t)k * m_rows + m];
    colsum_out[m] = s;
  }
}

// Operand of the first split-bf16 layer of a group-all point level: row (b, j) = [xyz[b][j][0..2] | feats[b][0..C-1][j]]
// (the reference's cat of grouped xyz and features, pointnet2_utils.py GroupAll), written as [hi | lo | hi] with
// hi = rne_bf16(v), lo = rne_bf16(v - hi) and zeros in the k_pad - (3 + C) padding columns of each third.
// One workgroup per object; the (C, n) feature block is staged through LDS so that both sides are coalesced.
__global__ __launch_bounds__(256) void split3_points_kernel(int n, int C, const float *__restrict__ xyz,
                                                            const float *__restrict__ feats, int k_pad,
                                                            uint16_t *__restrict__ out, const int *__restrict__ n_obj_dev) {
  if (n_obj_dev && (int)blockIdx.x >= *n_obj_dev) return;      // object extent (gps_point_set_object_extent)
  extern __shared__ float tile[];                    // [n][3 + C + 1]
  const int b = blockIdx.x, K = 3 + C, ldt = K + 1;
  for (int e = threadIdx.x; e < n * 3; e += 256) tile[(e / 3) * ldt + (e % 3)] = xyz[(size_t)b * n * 3 + e];
  for (int e = threadIdx.x; e < C * n; e += 256) {   // feats[b][c][j], j fastest
    const int c = e / n, j = e - c * n;
    tile[j * ldt + 3 + c] = feats[(size_t)b * C * n + e];
  }
  __syncthreads();
  uint16_t *o = out + (size_t)b * n * 3 * k_pad;
  const int kp2 = k_pad >> 1;                         // two columns per thread: 4-byte stores
  for (int e = threadIdx.x; e < n * kp2; e += 256) {
    const int j = e / kp2, k = 2 * (e - j * kp2);
    unsigned int hi = 0u, lo = 0u;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      if (k + t < K) {
        const float v = tile[j * ldt + k + t];
        const uint16_t h = f2bf(v);
        hi |= (unsigned int)h << (16 * t);
        lo |= (unsigned int)f2bf(v - bf2f(h)) << (16 * t);
      }
    }
    unsigned int *row = reinterpret_cast<unsigned int *>(o + (size_t)j * 3 * k_pad);
    row[k >> 1] = hi;
    row[(k_pad + k) >> 1] = lo;
    row[(2 * k_pad + k) >> 1] = hi;
  }
}

// ---------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------
template <int BM, int BN, int WGM, int WGN, bool ATR, bool BTR, int EPI, int NBUF, bool PF = false, bool PERSIST = false>
int launch_cfg(Params &P, hipStream_t s) {
  constexpr int LDS = lds_bytes(BM, BN, NBUF);
  static_assert(LDS <= 160 * 1024, "stage buffers exceed the LDS of a CU");
  P.ntm = (P.M + BM - 1) / BM;
  P.ntn = (P.N + BN - 1) / BN;
  P.gm = BM >= 256 ? 4 : 8;           // ~16 (128-row) panels of K = 768 bf16 stay under the 4 MiB L2 of an XCD
  auto kern = &gemm_kernel<BM, BN, WGM, WGN, ATR, BTR, EPI, NBUF, PF, PERSIST>;
  constexpr int THREADS = WGM * WGN * 64;
  static PerDeviceFlag granted;
  if (LDS > 64 * 1024 && !grant_dynamic_lds(kern, LDS, granted)) return GPS_ERR_LAUNCH;
  long long blocks = (long long)P.ntm * P.ntn * P.splits;
  if (blocks <= 0 || blocks > 0x7FFFFFFFLL) return GPS_ERR_UNSUPPORTED;
  if (PERSIST) {
    // as many workgroups as the chip holds at once (a multiple of 8: the same number on every XCD)
    static int slots_of[kMaxDevices] = {};
    const int dev = current_device();
    if (dev < 0) return GPS_ERR_LAUNCH;
    int &slots = slots_of[dev];
    if (slots == 0) {
      int per_cu = 0;
      const int n_cu = device_cu_count();
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void *>(kern), THREADS, LDS) != hipSuccess ||
          n_cu <= 0 || per_cu < 1)
        return GPS_ERR_LAUNCH;
      slots = per_cu * n_cu / 8 * 8;
      if (slots < 8) slots = 8;
    }
    const long long want = (blocks + 7) / 8 * 8;
    blocks = want < slots ? want : slots;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(THREADS), LDS, s, P);
  return hipGetLastError() == hipSuccess ? GPS_OK : GPS_ERR_LAUNCH;
}

// tile configurations ("variants"): tile (BM x BN), wave grid, stage buffers (LDS) -> resident workgroups per CU,
// PF = all fragment reads of a stage issued before its first MFMA
//   0  128x128  2x2  2 bufs ( 64 KB)  2/CU      1  256x128  8x2  3 bufs (144 KB) 16 waves  2  128x128  4x2  2 bufs  PF
//   3  128x64   2x2  3 bufs ( 72 KB)  2/CU      4  256x256  2x4  2 bufs (128 KB)  1/CU  5  256x128  4x2  2 bufs ( 96 KB) 1/CU
//   6  128x64   2x2  2 bufs ( 48 KB)  3/CU      7  128x128  4x2  2 bufs ( 64 KB)  2/CU
//   8  = 7, persistent workgroups (stage ring runs across tiles)                  9  = 6, persistent
//  10  = 4, persistent                                                        11  256x256  2x2 (4 waves, 512 registers), operands
//                                                                                register-staged two stages ahead (gemm256_kernel)
constexpr int kVariants = 14;
struct VariantShape { int bm, bn; };
constexpr VariantShape kShapes[kVariants] = {{128, 128}, {256, 128}, {128, 128}, {128, 64}, {256, 256}, {256, 128}, {128, 64}, {128, 128},
                                              {128, 128}, {128, 64}, {256, 256}, {256, 256}, {256, 256}, {256, 256}};
template <bool ATR, bool BTR, int EPI>
int launch_variant(Params &P, int variant, hipStream_t s) {
  switch (variant) {
    case 0: return launch_cfg<128, 128, 2, 2, ATR, BTR, EPI, 2>(P, s);
    case 1: return launch_cfg<256, 128, 8, 2, ATR, BTR, EPI, 3>(P, s);
    case 2: return launch_cfg<128, 128, 4, 2, ATR, BTR, EPI, 2, true>(P, s);
    case 3:
      if constexpr (ATR) return launch_cfg<128, 128, 2, 2, ATR, BTR, EPI, 2>(P, s);   // reduction-major A tiles are >= 128 wide
      else return launch_cfg<128, 64, 2, 2, ATR, BTR, EPI, 3>(P, s);
    case 4:
      if constexpr (ATR) return launch_cfg<128, 128, 2, 2, ATR, BTR, EPI, 2>(P, s);   // 256 x 256 with two transposed
      else return launch_cfg<256, 256, 2, 4, ATR, BTR, EPI, 2>(P, s);                 // operands exceeds 256 VGPRs
    case 5: return launch_cfg<256, 128, 4, 2, ATR, BTR, EPI, 2>(P, s);
    case 6:
      if constexpr (ATR) return launch_cfg<128, 128, 4, 2, ATR, BTR, EPI, 2>(P, s);
      else return launch_cfg<128, 64, 2, 2, ATR, BTR, EPI, 2>(P, s);
    case 7: return launch_cfg<128, 128, 4, 2, ATR, BTR, EPI, 2>(P, s);
    case 8:                                                                            // 7, persistent
      if constexpr (EPI == EPI_F32) return launch_cfg<128, 128, 4, 2, ATR, BTR, EPI, 2>(P, s);
      else return launch_cfg<128, 128, 4, 2, ATR, BTR, EPI, 2, false, true>(P, s);
    case 10:                                                                           // 4, persistent
      if constexpr (EPI == EPI_F32 || ATR) return launch_cfg<128, 128, 4, 2, ATR, BTR, EPI, 2>(P, s);
      else return launch_cfg<256, 256, 2, 4, ATR, BTR, EPI, 2, false, true>(P, s);
    case 9:                                                                            // 6, persistent
      if constexpr (EPI == EPI_F32 || ATR) return launch_cfg<128, 128, 4, 2, ATR, BTR, EPI, 2>(P, s);
      else return launch_cfg<128, 64, 2, 2, ATR, BTR, EPI, 2, false, true>(P, s);
    case 11:      // 256 x 256, four waves, register-staged operands two stages ahead (gemm256_kernel): NT / NN, whole K stages
      if constexpr (ATR || EPI == EPI_F32 || EPI == EPI_RELU_SPLIT || EPI == EPI_RELU_MAX16) return launch_cfg<128, 128, 4, 2, ATR, BTR, EPI, 2>(P, s);
      else return (P.K % BK == 0 && P.K >= BK && P.splits == 1) ? launch_256<BTR, EPI>(P, s) : launch_cfg<128, 128, 4, 2, ATR, BTR, EPI, 2>(P, s);
    case 12:      // 256 x 256, eight waves in two alternating groups (gemm8p_kernel): NT / NN with whole K stages, TN
      if constexpr (ATR && BTR && EPI == EPI_F32) return launch_8p<ATR, BTR, EPI>(P, s);
      else if constexpr (ATR || EPI == EPI_F32 || EPI == EPI_RELU_SPLIT || EPI == EPI_RELU_MAX16) return launch_cfg<128, 128, 4, 2, ATR, BTR, EPI, 2>(P, s);
      else return (P.K % BK == 0 && P.K >= BK && P.splits == 1) ? launch_8p<ATR, BTR, EPI>(P, s) : launch_cfg<128, 128, 4, 2, ATR, BTR, EPI, 2>(P, s);
    case 13:      // stream-K form of 12 (gemm8p_sk_kernel): NT / NN with bf16 epilogues and a workspace; else as 12 / the 128 x 128 tiles
      if constexpr (ATR || EPI == EPI_F32 || EPI == EPI_RELU_SPLIT || EPI == EPI_RELU_MAX16) return launch_cfg<128, 128, 4, 2, ATR, BTR, EPI, 2>(P, s);
      else if (P.partial && P.sk_flags && P.K >= BK && P.splits == 1) return launch_sk<BTR, EPI>(P, s);
      else return (P.K % BK == 0 && P.K >= BK && P.splits == 1) ? launch_8p<ATR, BTR, EPI>(P, s) : launch_cfg<128, 128, 4, 2, ATR, BTR, EPI, 2>(P, s);
    default: return GPS_ERR_INVALID_ARGUMENT;
  }
}

// default variant from the per-shape timings of profiles/r2/gemm_bench_*.json: the 8-wave 128 x 128 tile (two
// workgroups = 16 waves per CU) once there are enough tiles to fill the chip more than once, else 128 x 64 tiles at
// three workgroups per CU; weight gradients: 8 waves with the fragment reads of a stage issued up front
// weight gradients with >= 18 output tiles of 256 x 256 and a long token reduction: the two-group 256 x 256 kernel with one
// workgroup per CU, the reduction split so that (nearly) every CU gets one (profiles/r3/gemm_8p_*.log)
inline int tn_8p_splits(int M, int N, int K) {
  const long long tiles = (long long)((M + 255) / 256) * ((N + 255) / 256);
  const int nkt = (K + 63) / 64;
  if (tiles < 18 || tiles > 256 || nkt < 64) return 0;
  long long s = 256 / tiles;
  if (s > nkt / 24) s = nkt / 24;        // >= 24 K tiles per workgroup: below that the fp32 partial tiles and the prologue dominate
  return tiles * s >= 160 ? (int)s : 0;
}
inline int pick_variant(int form, int M, int N, int K, int splits, int epilogue = EPI_BIAS) {
  const long long tiles256 = (long long)((M + 255) / 256) * ((N + 255) / 256);
  if (form == GPS_GEMM_TN) return (tiles256 >= 18 && tiles256 * splits >= 160 && tiles256 * splits <= 256) ? 12 : 2;
  // long reductions over >= 140 tiles of 256 x 256 (more than half the CUs busy in the last round): the two-group kernel
  // (12 608 x 768 x 3072: 62.5 vs 71.7 us forward, 71.1 vs 86.0 us input gradient; at K = 768 its per-tile prologue and
  // 128 KB store tail cancel the gain -- profiles/r3/gemm_8p_shapes.log)
  static const int min_k_8p = [] {                      // GPS_GEMM_8P_MIN_K: A/B of the threshold (tools, bench runs)
    const char *e = getenv("GPS_GEMM_8P_MIN_K");
    const int v = e ? atoi(e) : 0;
    return v > 0 ? v : 1536;
  }();
  if (K % 64 == 0 && K >= min_k_8p && tiles256 >= 140) return 12;
  // [r6] with the four-quadrant epilogue (store_quads: 4.4 -> 2.2 us per tile at EPI_BIAS) the two-group kernel also wins at
  // K = 768 once the launch is several rounds of tiles (12 608 x 2 304 x 768: 52.4 vs 66.8 us, x 3 072 with the GELU epilogues
  // 111 vs 124 / 106 vs 122; profiles/r6/gemm_probe_table_h_streamk.txt); the 264 / 297-tile products of the joint layers
  // (1.03 / 1.16 rounds) stay on the 128 x 128 tiles
  if (K % 64 == 0 && K >= 768 && tiles256 >= 400 && epilogue != EPI_RELU_SPLIT && epilogue != EPI_RELU_MAX16) return 12;
  const long long tiles = (long long)((M + 127) / 128) * ((N + 127) / 128) * splits;
  return tiles >= 384 ? 7 : 6;
}

#ifdef GPS_GEMM_TRACE
static unsigned long long *g_probe_trace = nullptr;      // set by tools/probes/gemm_probe.hip
#endif

}  // namespace gps_gemm

#ifndef GPS_GEMM_NO_ENTRY      // (tools/probes/sk_regs.hip compiles single kernels of this file)
extern "C" {

int gps_gemm_pick_splits(int form, int M, int N, int K) {
  if (form != GPS_GEMM_TN) return 1;
  if (const int s8 = gps_gemm::tn_8p_splits(M, N, K)) return s8;
  // weight gradient: M x N is small (a few dozen 128 x 128 tiles), K = tokens is long.  The chip holds 512 of
  // these workgroups at once (2 per CU); the time is a staircase in ceil(tiles * splits / 512), so take the largest
  // split count that still fits ONE resident round, as long as every split keeps >= 8 stages of work
  // (profiles/r2/split_sweep.json: 36 tiles -> 14, 96 -> 5, 108 / 114 -> 4, 144 -> 3; 50-stage K -> at most 6).
  const long long tiles = (long long)((M + 127) / 128) * ((N + 127) / 128);
  const int nkt = (K + 63) / 64;
  long long s = 512 / (tiles < 1 ? 1 : tiles);
  if (s > 16) s = 16;
  if (s > nkt / 8) s = nkt / 8;
  return s < 1 ? 1 : (int)s;
}

int gps_gemm_pick_variant(int form, int M, int N, int K, int splits) {
  return gps_gemm_pick_variant_ex(form, M, N, K, splits, GPS_GEMM_EPI_BIAS);
}

int gps_gemm_pick_variant_ex(int form, int M, int N, int K, int splits, int epilogue) {
  if (form != GPS_GEMM_NT && form != GPS_GEMM_NN && form != GPS_GEMM_TN) return -1;
  return gps_gemm::pick_variant(form, M, N, K, splits < 1 ? gps_gemm_pick_splits(form, M, N, K) : splits, epilogue);
}

long long gps_gemm_sk_workspace_bytes(void) {
  return (long long)gps_gemm::kSkFlagBytes + (long long)gps_gemm::kSkMaxGrid * gps_gemm::kSkSlabFloats * 4;
}

long long gps_gemm_workspace_floats(int form, int M, int N, int splits) {
  if (form == GPS_GEMM_NT || splits <= 1) return 0;
  return (long long)splits * ((long long)M * N + M);
}

int gps_split3_points(int b, int n, int c, const float *xyz, const float *feats, int k_pad, void *out,
                      gps_stream_t stream) {
  if (b < 0 || n < 1 || c < 0 || k_pad < 3 + c || (k_pad & 7)) return GPS_ERR_INVALID_ARGUMENT;
  if (b == 0) return GPS_OK;
  if (!xyz || (c > 0 && !feats) || !out) return GPS_ERR_INVALID_ARGUMENT;
  const size_t lds = (size_t)n * (3 + c + 1) * sizeof(float);
  if (lds > 64 * 1024) return GPS_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(gps_gemm::split3_points_kernel, dim3(b), dim3(256), lds, (hipStream_t)stream, n, c, xyz, feats, k_pad,
                     (uint16_t *)out, gps::object_extent());
  return hipGetLastError() == hipSuccess ? GPS_OK : GPS_ERR_LAUNCH;
}

static int g_wgrad_xcd_queues = 1;
int gps_gemm_wgrad_grouped_set_xcd_queues(int on) {
  const int prev = g_wgrad_xcd_queues;
  if (on >= 0) g_wgrad_xcd_queues = on ? 1 : 0;
  return prev;
}

int gps_gemm_wgrad_grouped(const gps_wgrad_problem *problems, int n_problems, gps_stream_t stream) {
  using namespace gps_gemm;
  if (n_problems < 0 || (n_problems > 0 && !problems)) return GPS_ERR_INVALID_ARGUMENT;
  hipStream_t s = (hipStream_t)stream;
  constexpr int LDS = 2 * 4 * 128 * BK * 2;
  static PerDeviceFlag granted;
  if (!grant_dynamic_lds(&wgrad_grouped_kernel, LDS, granted)) return GPS_ERR_LAUNCH;
  const int n_cu = device_cu_count();
  if (n_cu <= 0) return GPS_ERR_LAUNCH;
  // validate everything before the first launch (requirements of the TN form of gps_gemm_bf16)
  int live = 0;
  for (int i = 0; i < n_problems; ++i) {
    const gps_wgrad_problem &q = problems[i];
    if (q.M < 0 || q.N < 0 || q.K < 0) return GPS_ERR_INVALID_ARGUMENT;
    if (q.M == 0 || q.N == 0) continue;
    if (!q.A || !q.B || !q.C) return GPS_ERR_INVALID_ARGUMENT;
    if ((q.lda & 7) || (q.ldb & 7) || (q.ldc & 3) || (q.M & 7) || (q.N & 7)) return GPS_ERR_UNSUPPORTED;
    if (((uintptr_t)q.A & 15) || ((uintptr_t)q.B & 15) || ((uintptr_t)q.C & 15)) return GPS_ERR_UNSUPPORTED;
    if ((long long)q.K * q.lda * 2 >= 0x7FFFFFFFLL || (long long)q.K * q.ldb * 2 >= 0x7FFFFFFFLL) return GPS_ERR_UNSUPPORTED;
    if (q.lda >= 0x7FFFFFFFLL || q.ldb >= 0x7FFFFFFFLL || q.ldc >= 0x7FFFFFFFLL) return GPS_ERR_UNSUPPORTED;
    ++live;
  }
  if (live == 0) return GPS_OK;
  // longest reductions first (stable): with the snake deal of wgrad_grouped_kernel every workgroup gets one tile of each cost class
  int order[kWgradMaxProblems];
  static unsigned int slot_counters[kMaxDevices] = {};       // (the tables are __device__ globals: one set per device)
  const int dev_now = current_device();
  if (dev_now < 0) return GPS_ERR_LAUNCH;
  unsigned int &slot_counter = slot_counters[dev_now];
  for (int base = 0; base < n_problems;) {
    int cnt = 0, end = base;
    while (end < n_problems && cnt < kWgradMaxProblems) {
      if (problems[end].M > 0 && problems[end].N > 0) order[cnt++] = end;
      ++end;
    }
    base = end;
    if (cnt == 0) break;
    for (int i = 1; i < cnt; ++i) {                           // insertion sort by K descending (cnt <= 256)
      const int v = order[i];
      int j = i - 1;
      while (j >= 0 && problems[order[j]].K < problems[v].K) { order[j + 1] = order[j]; --j; }
      order[j + 1] = v;
    }
    const int slot = (int)(slot_counter++ % kWgradSlots);
    long long all_tiles = 0;
    for (int i = 0; i < cnt; ++i) all_tiles += (long long)((problems[order[i]].M + 255) / 256) * ((problems[order[i]].N + 255) / 256);
    if (all_tiles > 0x3FFFFFFFLL) return GPS_ERR_UNSUPPORTED;
    const int grid = all_tiles < n_cu ? (int)all_tiles : n_cu;
    // XCD-local queues (see WgradQueueState): the problems of one class -- same reduction length, same device extent,
    // i.e. tiles that take the same time whatever the live row count turns out to be -- are dealt to the queue that
    // holds the fewest tiles of that class so far (ties: fewest tiles overall).  Inside a queue: longest first.
    WgradQueueArgs qa = {};
    const bool queues = g_wgrad_xcd_queues && grid >= 2 * kWgradQueues && grid % kWgradQueues == 0;
    if (queues) {
      int qof[kWgradMaxProblems];
      long long q_class[kWgradQueues], q_all[kWgradQueues] = {};
      for (int i = 0; i < cnt;) {
        int j = i;
        while (j < cnt && problems[order[j]].K == problems[order[i]].K && problems[order[j]].extent_dev == problems[order[i]].extent_dev) ++j;
        for (int x = 0; x < kWgradQueues; ++x) q_class[x] = 0;
        // within a class: most tiles first, so the big problems are placed before the small ones fill the gaps
        for (int a = i + 1; a < j; ++a) {
          const int v = order[a];
          const long long tv = (long long)((problems[v].M + 255) / 256) * ((problems[v].N + 255) / 256);
          int b = a - 1;
          while (b >= i && (long long)((problems[order[b]].M + 255) / 256) * ((problems[order[b]].N + 255) / 256) < tv) { order[b + 1] = order[b]; --b; }
          order[b + 1] = v;
        }
        for (int a = i; a < j; ++a) {
          const gps_wgrad_problem &q = problems[order[a]];
          const long long tl = (long long)((q.M + 255) / 256) * ((q.N + 255) / 256);
          int best = 0;
          for (int x = 1; x < kWgradQueues; ++x)
            if (q_class[x] < q_class[best] || (q_class[x] == q_class[best] && q_all[x] < q_all[best])) best = x;
          qof[a] = best;
          q_class[best] += tl;
          q_all[best] += tl;
        }
        i = j;
      }
      // records queue by queue (stable: the K-descending order survives inside a queue)
      int order2[kWgradMaxProblems], n2 = 0;
      long long tiles_before = 0;
      for (int x = 0; x < kWgradQueues; ++x) {
        qa.begin[x] = (int)tiles_before;
        qa.first_problem[x] = n2;
        for (int a = 0; a < cnt; ++a)
          if (qof[a] == x) {
            order2[n2++] = order[a];
            tiles_before += (long long)((problems[order[a]].M + 255) / 256) * ((problems[order[a]].N + 255) / 256);
          }
        qa.end[x] = (int)tiles_before;
        if (qa.first_problem[x] >= cnt) qa.first_problem[x] = cnt - 1;
      }
      for (int a = 0; a < cnt; ++a) order[a] = order2[a];
      qa.enabled = 1;
    }
    long long tile0 = 0;
    for (int c0 = 0; c0 < cnt; c0 += kWgradChunk) {
      WgradChunkArgs args = {};
      const int n = cnt - c0 < kWgradChunk ? cnt - c0 : kWgradChunk;
      for (int i = 0; i < n; ++i) {
        const gps_wgrad_problem &q = problems[order[c0 + i]];
        WgradRec &r = args.r[i];
        r.A = (const uint16_t *)q.A; r.B = (const uint16_t *)q.B; r.C = q.C; r.colsum = q.colsum; r.extent = q.extent_dev;
        r.lda = (int)q.lda; r.ldb = (int)q.ldb; r.ldc = (int)q.ldc; r.M = q.M; r.N = q.N; r.K = q.K;
        r.tile0 = (int)tile0;
        r.flags = q.accumulate ? 1 : 0;
        tile0 += (long long)((q.M + 255) / 256) * ((q.N + 255) / 256);
      }
      hipLaunchKernelGGL(wgrad_table_write_kernel, dim3(1), dim3(64), 0, s, args, slot, c0, n, grid, qa, grid);
      if (hipGetLastError() != hipSuccess) return GPS_ERR_LAUNCH;
    }
    const int total = (int)tile0;
    hipLaunchKernelGGL(wgrad_grouped_kernel, dim3((unsigned)grid), dim3(512), LDS, s, slot, cnt, total, queues ? 1 : 0 GPS_WGRAD_TRACE_ARG);
    if (hipGetLastError() != hipSuccess) return GPS_ERR_LAUNCH;
  }
  return GPS_OK;
}

// argument checks of gps_gemm_bf16 / gps_gemm_bf16_grouped and the kernel's record; *done: nothing to compute (M or N == 0)
static int gps_gemm_fill_params(const gps_gemm_args *a, gps_gemm::Params &P, bool *done) {
  using namespace gps_gemm;
  *done = false;
  if (!a || a->M < 0 || a->N < 0 || a->K < 0) return GPS_ERR_INVALID_ARGUMENT;
  if (a->form < 0 || a->form > 2 || a->epilogue < 0 || a->epilogue > 9) return GPS_ERR_INVALID_ARGUMENT;
  if (a->M == 0 || a->N == 0) {
    *done = true;
    return GPS_OK;
  }
  if (!a->A || !a->B || !a->C) return GPS_ERR_INVALID_ARGUMENT;
  // 16-byte global chunks and 8 / 16-byte stores: leading dimensions in multiples of 8 elements, N of 4, K of 8
  if ((a->lda & 7) || (a->ldb & 7) || (a->N & 3)) return GPS_ERR_UNSUPPORTED;
  if (a->form != GPS_GEMM_TN && (a->K & 7)) return GPS_ERR_UNSUPPORTED;     // K-major operands: whole 16-byte chunks
  if (((uintptr_t)a->A & 15) || ((uintptr_t)a->B & 15) || ((uintptr_t)a->C & 15)) return GPS_ERR_UNSUPPORTED;
  // 32-bit byte offsets into A and B (buffer addressing)
  {
    const long long ra = a->form == GPS_GEMM_TN ? a->K : a->M, rb = a->form == GPS_GEMM_NT ? a->N : a->K;
    if (ra * a->lda * 2 >= 0x7FFFFFFFLL || rb * a->ldb * 2 >= 0x7FFFFFFFLL) return GPS_ERR_UNSUPPORTED;
  }
  // 32-bit byte offsets into C and the saved-activation operands (buffer addressing in the epilogue; rows of the
  // last tile past M are addressed too, and must not wrap)
  if (a->epilogue != GPS_GEMM_EPI_F32 && a->epilogue != GPS_GEMM_EPI_RELU_MAX16) {
    const long long rows = (long long)a->M + 256;
    if (rows * a->ldc * 2 >= 0x7FFFFFFFLL) return GPS_ERR_UNSUPPORTED;
    if (a->aux && rows * a->ldaux * 2 >= 0x7FFFFFFFLL) return GPS_ERR_UNSUPPORTED;
    if (a->aux_out && rows * a->ldaux_out * 2 >= 0x7FFFFFFFLL) return GPS_ERR_UNSUPPORTED;
  }
  const bool f32out = a->epilogue == GPS_GEMM_EPI_F32;
  // fp32 sums (optionally split over K): the weight-gradient form, and the input-gradient form for long reductions
  // over few rows (the masked-LM head: K = vocabulary); every other epilogue belongs to NT / NN
  if (f32out ? a->form == GPS_GEMM_NT : a->form == GPS_GEMM_TN) return GPS_ERR_UNSUPPORTED;
  if (f32out && a->form == GPS_GEMM_NN && a->colsum) return GPS_ERR_UNSUPPORTED;
  if (a->ldc & 3) return GPS_ERR_UNSUPPORTED;
  if (a->form == GPS_GEMM_TN && (a->M & 7)) return GPS_ERR_UNSUPPORTED;     // A is M-contiguous there
  if (a->form != GPS_GEMM_NT && (a->N & 7)) return GPS_ERR_UNSUPPORTED;     // B is N-contiguous there
  if ((a->epilogue == GPS_GEMM_EPI_DGELU || a->epilogue == GPS_GEMM_EPI_DRELU || a->epilogue == GPS_GEMM_EPI_MUL_AUX) && (!a->aux || (a->ldaux & 3)))
    return GPS_ERR_INVALID_ARGUMENT;
  if (a->aux_out && (a->ldaux_out & 3)) return GPS_ERR_UNSUPPORTED;
  if (a->p_drop < 0.f || a->p_drop >= 1.f) return GPS_ERR_INVALID_ARGUMENT;
  if (a->bias && ((uintptr_t)a->bias & 15)) return GPS_ERR_UNSUPPORTED;
  if (a->epilogue == GPS_GEMM_EPI_RELU_SPLIT || a->epilogue == GPS_GEMM_EPI_RELU_MAX16) {
    if (a->form != GPS_GEMM_NT || a->p_drop != 0.f) return GPS_ERR_UNSUPPORTED;
    if (a->epilogue == GPS_GEMM_EPI_RELU_SPLIT && ((a->N & 7) || (a->ldc & 7) || a->ldc < 3LL * a->N)) return GPS_ERR_UNSUPPORTED;
    if (a->epilogue == GPS_GEMM_EPI_RELU_MAX16 && (a->M & 15)) return GPS_ERR_UNSUPPORTED;
  }

  P = Params{};
  P.M = a->M; P.N = a->N; P.K = a->K;
  P.A = (const uint16_t *)a->A; P.lda = a->lda;
  P.B = (const uint16_t *)a->B; P.ldb = a->ldb;
  P.C = a->C; P.ldc = a->ldc;
  P.bias = a->bias;
  P.aux = (const uint16_t *)a->aux; P.ldaux = a->ldaux;
  P.aux_out = (uint16_t *)a->aux_out; P.ldaux_out = a->ldaux_out;
  P.nkt = (a->K + BK - 1) / BK;
  P.splits = 1;
  if (f32out && a->splits > 1) P.splits = a->splits < P.nkt ? a->splits : (P.nkt > 0 ? P.nkt : 1);
  P.kt_per_split = (P.nkt + P.splits - 1) / P.splits;
  P.splits = P.kt_per_split > 0 ? (P.nkt + P.kt_per_split - 1) / P.kt_per_split : 1;   // no empty split
  if (P.splits < 1) P.splits = 1;
  P.keep_scale = a->p_drop > 0.f ? 1.f / (1.f - a->p_drop) : 1.f;
  P.drop_thr = a->p_drop > 0.f ? (unsigned int)((double)a->p_drop * 4294967296.0) : 0u;
  P.seed = a->seed; P.seed_dev = (const unsigned long long *)a->seed_dev;
  P.extent_dev = a->extent_dev;
  P.row0 = a->form != GPS_GEMM_TN ? a->reserved2 : 0;
  if (a->K == 0) P.nkt = 0;

  return GPS_OK;
}

int gps_gemm_bf16_grouped(const gps_gemm_args *args, int n, gps_stream_t stream) {
  using namespace gps_gemm;
  if (n < 0 || (n > 0 && !args)) return GPS_ERR_INVALID_ARGUMENT;
  if (n == 0) return GPS_OK;
  if (n == 1) return gps_gemm_bf16(&args[0], stream);
  if (n > kGroupMax) return GPS_ERR_UNSUPPORTED;
  const int form = args[0].form, epi = args[0].epilogue;
  if (form != GPS_GEMM_NT && form != GPS_GEMM_NN) return GPS_ERR_UNSUPPORTED;
  if (epi == GPS_GEMM_EPI_F32 || epi == GPS_GEMM_EPI_RELU_SPLIT || epi == GPS_GEMM_EPI_RELU_MAX16) return GPS_ERR_UNSUPPORTED;
  GroupParams G = {};
  bool ragged = false;
  for (int i = 0; i < n; ++i) {
    const gps_gemm_args &a = args[i];
    if (a.form != form || a.epilogue != epi) return GPS_ERR_INVALID_ARGUMENT;     // one kernel instantiation per launch
    Params P;
    bool done = false;
    const int st = gps_gemm_fill_params(&a, P, &done);
    if (st != GPS_OK) return st;
    if (done) continue;
    if (a.K < BK) return GPS_ERR_UNSUPPORTED;
    ragged = ragged || (a.K % BK) != 0;
    // longest reduction first (stable insertion): its tiles get the lowest block ids, i.e. start first
    int at = G.n;
    while (at > 0 && G.p[at - 1].K < P.K) {
      G.p[at] = G.p[at - 1];
      --at;
    }
    G.p[at] = P;
    ++G.n;
  }
  if (G.n == 0) return GPS_OK;
  if (ragged && epi != GPS_GEMM_EPI_BIAS) return GPS_ERR_UNSUPPORTED;           // ragged K: the plain-bias instantiations only
  hipStream_t s = (hipStream_t)stream;
  if (form == GPS_GEMM_NT) {
    switch (epi) {
      case GPS_GEMM_EPI_BIAS: return launch_grouped_any<false, EPI_BIAS>(G, ragged, s);
      case GPS_GEMM_EPI_BIAS_GELU: return launch_grouped<false, EPI_BIAS_GELU, false>(G, s);
      case GPS_GEMM_EPI_BIAS_RELU: return launch_grouped<false, EPI_BIAS_RELU, false>(G, s);
      case GPS_GEMM_EPI_BIAS_GELU_FACTOR: return launch_grouped<false, EPI_BIAS_GELU_FACTOR, false>(G, s);
      default: return GPS_ERR_UNSUPPORTED;
    }
  }
  switch (epi) {
    case GPS_GEMM_EPI_BIAS: return launch_grouped_any<true, EPI_BIAS>(G, ragged, s);
    case GPS_GEMM_EPI_DGELU: return launch_grouped<true, EPI_DGELU, false>(G, s);
    case GPS_GEMM_EPI_DRELU: return launch_grouped<true, EPI_DRELU, false>(G, s);
    case GPS_GEMM_EPI_MUL_AUX: return launch_grouped<true, EPI_MUL_AUX, false>(G, s);
    default: return GPS_ERR_UNSUPPORTED;
  }
}

int gps_gemm_bf16(const gps_gemm_args *a, gps_stream_t stream) {
  using namespace gps_gemm;
  Params P;
  bool done = false;
  {
    const int st0 = gps_gemm_fill_params(a, P, &done);
    if (st0 != GPS_OK || done) return st0;
  }
  const bool f32out = a->epilogue == GPS_GEMM_EPI_F32;
  hipStream_t s = (hipStream_t)stream;
#ifdef GPS_GEMM_TRACE
  P.trace = g_probe_trace;
#endif
  int variant = a->variant;
  if (variant < 0) variant = pick_variant(a->form, a->M, a->N, a->K, P.splits, a->epilogue);
  if (variant >= kVariants) return GPS_ERR_INVALID_ARGUMENT;

  if (variant == 13 && a->form != GPS_GEMM_TN && !f32out && a->workspace) {
    // stream-K scratch (gps_gemm_sk_workspace_bytes()): [arrival words | one fp32 slab per workgroup position]
    if ((uintptr_t)a->workspace & 15) return GPS_ERR_UNSUPPORTED;
    P.sk_flags = reinterpret_cast<unsigned int *>(a->workspace);
    P.partial = a->workspace + kSkFlagBytes / 4;
  }
  int st;
  // ([r4] negative result: cutting the rows of a product whose 128 x 128 tiles leave a nearly empty last round -- 8 320 x
  // 2 048: 1 040 tiles = 2.03 rounds of 512 -- into whole rounds + a tail launch of 128 x 64 tiles was built, verified
  // bit-equal and measured: 13.79 / 13.88 ms per step against 13.69 / 13.54 without it on one box.  A nearly empty
  // round is short -- its few tiles have their CUs to themselves -- so the tail launch only adds its own ramp.)
  if (a->form == GPS_GEMM_NT) {
    switch (a->epilogue) {
      case GPS_GEMM_EPI_BIAS: st = launch_variant<false, false, EPI_BIAS>(P, variant, s); break;
      case GPS_GEMM_EPI_BIAS_GELU: st = launch_variant<false, false, EPI_BIAS_GELU>(P, variant, s); break;
      case GPS_GEMM_EPI_BIAS_RELU: st = launch_variant<false, false, EPI_BIAS_RELU>(P, variant, s); break;
      case GPS_GEMM_EPI_BIAS_GELU_FACTOR: st = launch_variant<false, false, EPI_BIAS_GELU_FACTOR>(P, variant, s); break;
      // the split-bf16 MLP forms exist for the two default tile configurations and the two-group 256 x 256 kernel only
      case GPS_GEMM_EPI_RELU_SPLIT:
        st = (variant == 12 && P.K % BK == 0 && P.K >= BK) ? launch_8p<false, false, EPI_RELU_SPLIT>(P, s)
             : variant == 6 ? launch_cfg<128, 64, 2, 2, false, false, EPI_RELU_SPLIT, 2>(P, s)
                            : launch_cfg<128, 128, 4, 2, false, false, EPI_RELU_SPLIT, 2>(P, s);
        break;
      case GPS_GEMM_EPI_RELU_MAX16:
        st = (variant == 12 && P.K % BK == 0 && P.K >= BK) ? launch_8p<false, false, EPI_RELU_MAX16>(P, s)
             : variant == 6 ? launch_cfg<128, 64, 2, 2, false, false, EPI_RELU_MAX16, 2>(P, s)
                            : launch_cfg<128, 128, 4, 2, false, false, EPI_RELU_MAX16, 2>(P, s);
        break;
      default: return GPS_ERR_UNSUPPORTED;
    }
  } else if (a->form == GPS_GEMM_NN) {
    switch (a->epilogue) {
      case GPS_GEMM_EPI_BIAS: st = launch_variant<false, true, EPI_BIAS>(P, variant, s); break;
      case GPS_GEMM_EPI_DGELU: st = launch_variant<false, true, EPI_DGELU>(P, variant, s); break;
      case GPS_GEMM_EPI_DRELU: st = launch_variant<false, true, EPI_DRELU>(P, variant, s); break;
      case GPS_GEMM_EPI_MUL_AUX: st = launch_variant<false, true, EPI_MUL_AUX>(P, variant, s); break;
      case GPS_GEMM_EPI_F32: {
        if (P.splits > 1) {
          if (!a->workspace) return GPS_ERR_INVALID_ARGUMENT;
          P.partial = a->workspace;
        }
        st = launch_cfg<128, 128, 4, 2, false, true, EPI_F32, 2>(P, s);
        if (st == GPS_OK && P.splits > 1) {
          const long long elems = (long long)a->M * a->N;
          const int main_blocks = (int)((elems / 4 + 255) / 256);
          hipLaunchKernelGGL(splitk_reduce_kernel, dim3(main_blocks), dim3(256), 0, s, P.splits, elems, P.partial,
                             reinterpret_cast<float *>(a->C), a->ldc, a->N, a->M, nullptr, nullptr, main_blocks, a->extent_dev);
          st = hipGetLastError() == hipSuccess ? GPS_OK : GPS_ERR_LAUNCH;
        }
        break;
      }
      default: return GPS_ERR_UNSUPPORTED;
    }
  } else {
    if (P.splits > 1) {
      if (!a->workspace) return GPS_ERR_INVALID_ARGUMENT;
      P.partial = a->workspace;
      P.colsum = a->colsum ? a->workspace + (size_t)P.splits * a->M * a->N : nullptr;
    } else {
      P.colsum = a->colsum;
    }
    st = launch_variant<true, true, EPI_F32>(P, variant, s);
    if (st == GPS_OK && P.splits > 1) {
      const long long elems = (long long)a->M * a->N;
      const int main_blocks = (int)((elems / 4 + 255) / 256);
      const int cs_blocks = a->colsum ? (a->M + 255) / 256 : 0;
      hipLaunchKernelGGL(splitk_reduce_kernel, dim3(main_blocks + cs_blocks), dim3(256), 0, s, P.splits, elems, P.partial,
                         reinterpret_cast<float *>(a->C), a->ldc, a->N, a->M, P.colsum, a->colsum, main_blocks, nullptr);
      st = hipGetLastError() == hipSuccess ? GPS_OK : GPS_ERR_LAUNCH;
    }
  }
  return st;
}

}  // extern "C"
#endif  // GPS_GEMM_NO_ENTRY
