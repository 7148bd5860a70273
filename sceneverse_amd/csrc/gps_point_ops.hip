// gps_point_ops.hip -- PointNet++ set-abstraction ops for MI355X (gfx950 / CDNA4).
//
// Hand-written for 64-wide wavefronts; exported through the C ABI of include/gps_hip.h.
// Compile with -ffp-contract=off: every distance is evaluated as ((dx*dx + dy*dy) + dz*dz) with
// each fp32 op individually rounded, the pinned arithmetic of DESIGN.md / SURVEY.md App. B.0, so
// that indices agree bit for bit with oracle/pointnet2_oracle.c.
//
// Reference behaviour restated (not translated) from
//   /root/reference/modules/third_party/pointnet2/_ext_src/src/{sampling,ball_query,group_points,
//   interpolate}_gpu.cu   -- cited per kernel below.
//
// Design notes (why this is not the reference's launch shape):
//   * FPS: one WAVE per object, the whole cloud + running distances resident in VGPRs, arg-max by
//     DPP wave reduction + v_cmp ballots; zero LDS, zero barriers, `temp` never touches HBM.
//     The reference's 512-thread LDS tree (9 __syncthreads per round) only survives as the
//     tie-break ORDER, which is folded into the point->(lane,register) assignment.
//   * ball query: lanes = 64 consecutive points, radius test -> 64-bit ballot -> mbcnt compaction,
//     which emits hits in ascending index order exactly like the reference's serial scan.
//   * group: LDS-staged source rows, 16-byte coalesced stores along (npoint*nsample).
//   * group grad: deterministic CSR gather instead of atomicAdd.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#include <type_traits>

#include "gps_hip.h"

#pragma clang fp contract(off)

namespace gps {

// Object extent of the per-object launches (gps_point_set_object_extent): a device int, or null.  Defined below.
const int *object_extent();

constexpr int kWave = 64;
constexpr int kBlock = 256;            // 4 waves per workgroup, one per SIMD
constexpr int kWavesPerBlock = kBlock / kWave;

// ------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & (kWave - 1)); }
__device__ __forceinline__ int wave_id() { return (int)(threadIdx.x >> 6); }

// bit-reverse the low `bits` bits of x (bits == 0 -> 0)
__host__ __device__ __forceinline__ int bitrev(int x, int bits) {
#if defined(__HIP_DEVICE_COMPILE__)
  return bits ? (int)(__brev((unsigned)x) >> (32 - bits)) : 0;
#else
  int r = 0;
  for (int i = 0; i < bits; ++i) r |= ((x >> i) & 1) << (bits - 1 - i);
  return r;
#endif
}

template <int CTRL>
__device__ __forceinline__ int dpp_max_i32(int v) {
  // lanes whose DPP source is out of range keep `old` (= v): harmless for an idempotent max
  const int o = __builtin_amdgcn_update_dpp(v, v, CTRL, 0xf, 0xf, false);
  return o > v ? o : v;
}

// Max over the 64 lanes of a wave of a signed int; result is wave-uniform (SGPR).
__device__ __forceinline__ int wave_max_i32(int v) {
  v = dpp_max_i32<0x111>(v);  // row_shr:1
  v = dpp_max_i32<0x112>(v);  // row_shr:2
  v = dpp_max_i32<0x114>(v);  // row_shr:4
  v = dpp_max_i32<0x118>(v);  // row_shr:8   -> lane 15 of every row holds the row max
  v = dpp_max_i32<0x142>(v);  // row_bcast:15 -> rows 1,3 fold in rows 0,2
  v = dpp_max_i32<0x143>(v);  // row_bcast:31 -> rows 2,3 fold in rows 0,1; lane 63 = wave max
  return __builtin_amdgcn_readlane(v, 63);
}

// reference: src/sampling_gpu.cu:100-101 -- `if (mag <= 1e-3) continue;` compares the fp32
// magnitude with a DOUBLE literal.
__device__ __forceinline__ bool fps_skipped(float x, float y, float z) {
  const float mag = (x * x) + (y * y) + (z * z);
  return (double)mag <= 1e-3;
}

// ------------------------------------------------------------------------------------------
// Furthest point sampling, register-resident form (n <= 64 * 32).
//
// Reference: src/sampling_gpu.cu:69-173.  idx[0] = 0; every round updates temp[k] =
// min(d(k, old), temp[k]) for the non-skipped points and picks the arg-max.  Ties are resolved by
// the reference's launch shape (thread tid = k mod bs keeps its first maximum, then a pairwise
// tree over strides bs/2..1 where the lower slot wins): among equal maxima the winner minimises
//     key(k) = ( bitreverse_{log2 bs}(k mod bs), k div bs ),   bs = opt_n_threads(n).
// Here the point stored in register r of lane L is chosen so that key order == (L, r)
// lexicographic order:
//     bs >= 64 (p = log2 bs >= 6):  k = bitrev6(L) + 64 * bitrev_{p-6}(r / Q) + bs * (r % Q)
//     bs <  64:                     k = bitrev_p(L) + bs * r          (lanes >= bs idle)
// with Q = ceil(n / bs).  The tie-break is then "lowest lane, then lowest register".
//
// Running distances are kept as raw fp32 bit patterns in int registers: every value is -1.0f
// (skipped / padding slot, never selectable, never updated) or >= +0, and on that set signed
// integer order == float order, so v_min_i32 / v_max3_i32 replace fminf/fmaxf without the
// canonicalising v_max the IEEE mode otherwise forces.  Skipped slots get x = 1e30 so that their
// distance overflows to +inf and min(inf, -1) keeps -1.
// ------------------------------------------------------------------------------------------
__host__ __device__ __forceinline__ int fps_point_index(int L, int r, int p, int Q) {
  if (p >= 6) {
    const int w = r / Q, q = r - w * Q;
    if (w >= (1 << (p - 6))) return -1;
    return bitrev(L, 6) + 64 * bitrev(w, p - 6) + (q << p);
  }
  if (L >= (1 << p) || r >= Q) return -1;
  return bitrev(L, p) + (r << p);
}

template <int R>
__global__ __launch_bounds__(kBlock) void fps_resident_kernel(int b, int n, int m, int p, int Q,
                                                               const float *__restrict__ dataset,
                                                               int32_t *__restrict__ idxs,
                                                               float *__restrict__ centres,
                                                               const int *__restrict__ n_obj_dev) {
  const int obj = blockIdx.x * kWavesPerBlock + wave_id();
  if (obj >= b) return;  // whole wave exits together
  if (n_obj_dev && obj >= *n_obj_dev) return;   // object extent (gps_point_set_object_extent): nothing read or written
  const int L = lane_id();
  const float *ds = dataset + (size_t)obj * n * 3;
  int32_t *out = idxs + (size_t)obj * m;
  // [r5] optional (b, m, 3): the sampled points themselves, written with the indices (the host side otherwise ran
  // index cast -> expand -> gather, four launches per level, for what is 384 bytes per object)
  float *cen = centres ? centres + (size_t)obj * m * 3 : nullptr;

  // [r6] the slots are held as PAIRS (r, r + 1) so that the distance arithmetic is packed fp32 (v_pk_add_f32 /
  // v_pk_mul_f32: each half individually IEEE-rounded, i.e. the pinned ((dx*dx + dy*dy) + dz*dz) per point; this file is
  // compiled with -ffp-contract=off): 8 packed + 4 integer instructions per two points instead of 16 + 4.
  typedef float f2 __attribute__((ext_vector_type(2)));
  constexpr int RP = (R + 1) / 2;
  f2 x[RP], y[RP], z[RP];
  int t[2 * RP];  // fp32 bit patterns, see above
  constexpr int kNeg1 = (int)0xBF800000u;   // -1.0f
  constexpr int kInit = (int)0x501502F9u;   // 1e10f, src/sampling.cpp:74-76
#pragma unroll
  for (int r = 0; r < 2 * RP; ++r) {
    // branch-free: out-of-range slots read point 0 and are poisoned like skipped points
    const int k = r < R ? fps_point_index(L, r, p, Q) : -1;
    const bool in = k >= 0 && k < n;
    const int kk = in ? k : 0;
    const float px = ds[kk * 3 + 0], py = ds[kk * 3 + 1], pz = ds[kk * 3 + 2];
    const bool ok = in && !fps_skipped(px, py, pz);
    x[r >> 1][r & 1] = ok ? px : 1e30f;
    y[r >> 1][r & 1] = py;
    z[r >> 1][r & 1] = pz;
    t[r] = ok ? kInit : kNeg1;
  }

  // first centre: point 0 with its TRUE coordinates (it may itself be a skipped point)
  float x1 = ds[0], y1 = ds[1], z1 = ds[2];
  x1 = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(x1)));
  y1 = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(y1)));
  z1 = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(z1)));

  int mine = 0;  // lane (j & 63) buffers idx j; flushed every 64 rounds with one coalesced store
  for (int j = 1; j < m; ++j) {
    int lane_best = kNeg1;
    const f2 cx = {x1, x1}, cy = {y1, y1}, cz = {z1, z1};
#pragma unroll
    for (int i = 0; i < RP; ++i) {
      const f2 dx = x[i] - cx, dy = y[i] - cy, dz = z[i] - cz;
      const f2 d = (dx * dx + dy * dy) + dz * dz;
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int r = 2 * i + e;
        const int di = __float_as_int(d[e]);
        t[r] = di < t[r] ? di : t[r];
        lane_best = t[r] > lane_best ? t[r] : lane_best;
      }
    }
    const int M = wave_max_i32(lane_best);
    const unsigned long long cand = __ballot(lane_best == M);
    const int Lw = __builtin_amdgcn_readfirstlane(__ffsll((long long)cand) - 1);
    // lowest register of lane Lw that holds M; fetch the next centre from that lane's registers
    int rw = 0;
    bool found = false;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (!found) {
        const int tr = __builtin_amdgcn_readlane(t[r], Lw);
        if (tr == M) {
          found = true;
          rw = r;
          x1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x[r >> 1][r & 1]), Lw));
          y1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(y[r >> 1][r & 1]), Lw));
          z1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(z[r >> 1][r & 1]), Lw));
        }
      }
    }
    // M == -1 only when every point is skipped: then (Lw, rw) = (0, 0) -> index 0, as the
    // reference's (best = -1, besti = 0) initial state yields.
    const int old = fps_point_index(Lw, rw, p, Q);
    if (L == (j & 63)) mine = old;
    if ((j & 63) == 63) {
      out[j - 63 + L] = mine;  // j-63 .. j, all < m
      if (cen) {
        float *c = cen + (size_t)(j - 63 + L) * 3;
        c[0] = ds[mine * 3 + 0]; c[1] = ds[mine * 3 + 1]; c[2] = ds[mine * 3 + 2];
      }
      mine = 0;
    }
  }
  const int base = (m - 1) & ~63;  // first index of the unflushed tail (covers idx[0] = 0 too)
  if (((m - 1) & 63) != 63 && base + L < m) {
    out[base + L] = mine;
    if (cen) {
      float *c = cen + (size_t)(base + L) * 3;
      c[0] = ds[mine * 3 + 0]; c[1] = ds[mine * 3 + 1]; c[2] = ds[mine * 3 + 2];
    }
  }
}

// ------------------------------------------------------------------------------------------
// Furthest point sampling, streaming form for n > 2048: one 512-thread workgroup per object,
// running distances in HBM scratch `temp` (b,n).  Same semantics, same tie order: each thread
// keeps the first strict maximum of its strided sequence (= lowest k div bs), threads are ranked
// by bitreverse(tid).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void fps_streaming_kernel(int b, int n, int m,
                                                             const float *__restrict__ dataset,
                                                             float *__restrict__ temp,
                                                             int32_t *__restrict__ idxs) {
  constexpr int BS = 512, P = 9;
  __shared__ unsigned long long red[BS / kWave];
  __shared__ int winner_k[BS];
  __shared__ int s_old;
  const int obj = blockIdx.x;
  const int tid = threadIdx.x;
  const float *ds = dataset + (size_t)obj * n * 3;
  float *tp = temp + (size_t)obj * n;
  int32_t *out = idxs + (size_t)obj * m;
  for (int k = tid; k < n; k += BS) tp[k] = 1e10f;
  if (tid == 0) out[0] = 0;
  int old = 0;
  // rank of this thread in the reference's tree: lower bitrev wins ties
  const unsigned rank = (unsigned)bitrev(tid, P);
  for (int j = 1; j < m; ++j) {
    const float x1 = ds[old * 3 + 0], y1 = ds[old * 3 + 1], z1 = ds[old * 3 + 2];
    int besti = 0;
    float best = -1.f;
    for (int k = tid; k < n; k += BS) {
      const float x2 = ds[k * 3 + 0], y2 = ds[k * 3 + 1], z2 = ds[k * 3 + 2];
      if (fps_skipped(x2, y2, z2)) continue;
      const float dx = x2 - x1, dy = y2 - y1, dz = z2 - z1;
      const float d = (dx * dx + dy * dy) + dz * dz;
      const float d2 = d < tp[k] ? d : tp[k];
      tp[k] = d2;
      if (d2 > best) { best = d2; besti = k; }
    }
    winner_k[tid] = besti;
    // sortable key: value (0 for the -1 sentinel, bits+1 otherwise) high, inverted rank low
    const unsigned vb = best < 0.f ? 0u : (unsigned)__float_as_int(best) + 1u;
    unsigned long long key = ((unsigned long long)vb << 32) | (unsigned long long)(0xFFFFFFFFu - rank);
    for (int off = 32; off >= 1; off >>= 1) {
      const unsigned long long o = __shfl_xor(key, off, kWave);
      key = o > key ? o : key;
    }
    if (lane_id() == 0) red[wave_id()] = key;
    __syncthreads();
    if (tid == 0) {
      unsigned long long k2 = red[0];
      for (int w = 1; w < BS / kWave; ++w) k2 = red[w] > k2 ? red[w] : k2;
      const unsigned wr = 0xFFFFFFFFu - (unsigned)(k2 & 0xFFFFFFFFu);
      const int wt = bitrev((int)wr, P);
      s_old = winner_k[wt];
      out[j] = s_old;
    }
    __syncthreads();
    old = s_old;
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------
// gather_points: out[i,l,j] = points[i,l,idx[i,j]]     (src/sampling_gpu.cu:8-20)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void gather_points_kernel(int b, int c, int n, int m,
                                                                const float *__restrict__ points,
                                                                const int32_t *__restrict__ idx,
                                                                float *__restrict__ out) {
  const long long total = (long long)b * c * m;
  for (long long e = (long long)blockIdx.x * kBlock + threadIdx.x; e < total;
       e += (long long)gridDim.x * kBlock) {
    const int j = (int)(e % m);
    const long long il = e / m;  // i*c + l
    const int i = (int)(il / c);
    out[e] = points[il * n + idx[(long long)i * m + j]];
  }
}

// gather_points_grad: scatter-add (src/sampling_gpu.cu:34-47).  Unordered fp32 atomics like the
// reference; not on the GPS path (xyz carries no gradient).
__global__ __launch_bounds__(kBlock) void gather_points_grad_kernel(
    int b, int c, int n, int m, const float *__restrict__ grad_out,
    const int32_t *__restrict__ idx, float *__restrict__ grad_points) {
  const long long total = (long long)b * c * m;
  for (long long e = (long long)blockIdx.x * kBlock + threadIdx.x; e < total;
       e += (long long)gridDim.x * kBlock) {
    const int j = (int)(e % m);
    const long long il = e / m;
    const int i = (int)(il / c);
    atomicAdd(grad_points + il * n + idx[(long long)i * m + j], grad_out[e]);
  }
}

// ------------------------------------------------------------------------------------------
// ball query (src/ball_query_gpu.cu:9-44).
//
// One workgroup per object, its 4 waves split the centres; lane L of a wave owns points
// L, L+64, ... (R registers per coordinate when the cloud fits, else streamed from L1/L2).
// Per (centre, 64-point chunk): d2 -> v_cmp_lt ballot -> mbcnt prefix -> hits written at
// cnt + prefix into a per-wave LDS row; the row (+ first-hit padding, or zeros when no hit) is
// stored with one coalesced write.  Chunks are visited in ascending order and a centre stops as
// soon as nsample hits are found, so the output equals the reference's serial scan.
// ------------------------------------------------------------------------------------------
constexpr int kCentresPerLoad = 21;   // 63 lanes = 21 centres x 3 coordinates
// compile-time loop (the block schedule below indexes registers with its counter)
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F &&f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}
// Block schedule of the register-resident scan (R % 8 == 0): the chunks of a cloud are tested in blocks, the scan of a
// centre ends at the first block boundary where nsample hits exist.  A block costs ~10 vector instructions per chunk
// whether or not its hits are needed, so short first blocks pay when many scans end early (padding objects: every
// point is a hit; dense clouds: 32 hits inside the first 100 - 300 points) and long ones when the whole cloud is
// walked (one scalar decision per block).  S = (A, B, C, REST): blocks of A, B, C chunks (0 = absent), then blocks of
// REST chunks to the end of the cloud.  All even (registers hold PAIRS of chunks) and <= 8 (overshoot area of a row).
template <int A_, int B_, int C_, int REST_>
struct BqSched {
  static constexpr int A = A_, B = B_, C = C_, REST = REST_;
  static_assert(A % 2 == 0 && B % 2 == 0 && C % 2 == 0 && REST % 2 == 0 && REST >= 2, "blocks are whole chunk pairs");
  static_assert(A <= 8 && B <= 8 && C <= 8 && REST <= 8, "a block may overshoot nsample by < 512 slots");
};
template <int R, typename S = BqSched<8, 0, 0, 8>>  // R > 0: register-resident cloud of <= 64*R points; R == 0: streamed
__global__ __launch_bounds__(kBlock) void ball_query_kernel(int b, int n, int m, float radius,
                                                             int nsample,
                                                             const float *__restrict__ new_xyz,
                                                             const float *__restrict__ xyz,
                                                             int32_t *__restrict__ idx,
                                                             const int *__restrict__ n_obj_dev) {
  extern __shared__ int32_t bq_rows[];  // kWavesPerBlock rows of nsample ints
  const int obj = blockIdx.x;
  if (n_obj_dev && obj >= *n_obj_dev) return;   // object extent: nothing read or written
  const int L = lane_id(), w = __builtin_amdgcn_readfirstlane(wave_id());
  const float *p = xyz + (size_t)obj * n * 3;
  const float *q = new_xyz + (size_t)obj * m * 3;
  int32_t *o = idx + (size_t)obj * m * nsample;
  int32_t *row = bq_rows + w * nsample;
  const float radius2 = radius * radius;
  const int nchunks = (n + kWave - 1) / kWave;

  // One compaction step: `mask` = lanes of chunk `chunk` inside the ball.  Hits are appended to the
  // wave's LDS row in lane (= index) order; wave-uniform bookkeeping stays in SGPRs.
  auto take = [&](unsigned long long mask, int chunk, bool lane_hit, int &cnt, int &first) {
    const int pos = cnt + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32),
                                                         __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
    if (lane_hit && pos < nsample) row[pos] = chunk * kWave + L;
    if (cnt == 0) first = chunk * kWave + (__ffsll((long long)mask) - 1);
    cnt += __popcll(mask);
  };

  if (R > 1) {
    // Register-resident cloud held as PAIRS of consecutive chunks, so that the distance arithmetic of
    // two chunks against one centre is packed fp32 (v_pk_add_f32 / v_pk_mul_f32: each half individually
    // IEEE-rounded, i.e. the pinned ((dx*dx + dy*dy) + dz*dz) per point): 8 packed ops + 2 compares per
    // 128 points.
    //
    // Compaction without a per-chunk scalar chain (round 1 spent its time there: ballot -> branch -> mbcnt ->
    // LDS write -> popcount, ~12 scalar instructions per hit chunk on the CU's one scalar unit).  Now every
    // chunk of a pair does, unconditionally: compare -> mask in an SGPR pair; slot = hits so far + mbcnt(mask)
    // (2 VALU); one LDS store whose ADDRESS is selected by the mask (hit lanes -> row[slot], the others ->
    // a private dummy word: no exec masking, no branch); hits so far += s_bcnt1(mask) (2 scalar ops).  One
    // scalar compare per PAIR of chunks ends the scan once nsample hits exist.  Slots are the ascending point
    // index order, so the first nsample of them are exactly the reference's serial scan.
    typedef float f2 __attribute__((ext_vector_type(2)));
    constexpr int RP = R > 1 ? R / 2 : 1;
    f2 px[RP], py[RP], pz[RP];
#pragma unroll
    for (int i = 0; i < RP; ++i) {
      const int k0 = (2 * i) * kWave + L, k1 = k0 + kWave;
      const bool in0 = k0 < n, in1 = k1 < n;
      // lanes past the end of the cloud sit at x = +inf: d2 = +inf is never inside the ball, so the
      // ballots need no validity mask (finite centres; no inf - inf anywhere)
      px[i].x = in0 ? p[k0 * 3 + 0] : INFINITY;  px[i].y = in1 ? p[k1 * 3 + 0] : INFINITY;
      py[i].x = in0 ? p[k0 * 3 + 1] : 0.f;  py[i].y = in1 ? p[k1 * 3 + 1] : 0.f;
      pz[i].x = in0 ? p[k0 * 3 + 2] : 0.f;  pz[i].y = in1 ? p[k1 * 3 + 2] : 0.f;
    }
    // per-wave LDS: row of nsample + 512 slots (a block of 8 chunks may overshoot nsample by < 512) + 64 dummy words
    const int row_len = nsample + 8 * kWave;
    int32_t *wrow = bq_rows + w * (row_len + kWave);
    int32_t *dummy = wrow + row_len + L;
    const int dummy_at = row_len + L;
    // the wave's centres are fetched 21 at a time with ONE vector load (lane 3 jj + c = coordinate c of its jj-th
    // centre) and broadcast from there: a load per centre put a memory round trip in front of every scan
    for (int j0 = w; j0 < m; j0 += kCentresPerLoad * kWavesPerBlock) {
     const int jl = j0 + (L / 3) * kWavesPerBlock;
     const int cv = (L < 3 * kCentresPerLoad && jl < m) ? __float_as_int(q[jl * 3 + L % 3]) : 0;
     for (int jj = 0; jj < kCentresPerLoad; ++jj) {
      const int j = j0 + jj * kWavesPerBlock;
      if (j >= m) break;
      const float c0 = __int_as_float(__builtin_amdgcn_readlane(cv, 3 * jj + 0));
      const float c1 = __int_as_float(__builtin_amdgcn_readlane(cv, 3 * jj + 1));
      const float c2 = __int_as_float(__builtin_amdgcn_readlane(cv, 3 * jj + 2));
      const f2 cx = {c0, c0}, cy = {c1, c1}, cz = {c2, c2};
      int cnt = 0;                                                          // wave-uniform (SGPR)
      if constexpr (R >= 8 && R % 8 == 0) {
        // [r3] blocks of 8 chunks (512 points): ALL eight radius tests first (pure VALU throughput: 4 packed
        // distance evaluations + 8 compares, their 64-bit masks land in SGPR pairs), then ONE scalar prefix over the
        // eight popcounts, then the compaction of the chunks that have hits and still start below nsample.  The
        // scalar unit is consulted once per block instead of once per chunk pair: the compare -> popcount -> add ->
        // branch chain that bounded the previous form (57 us, VALU active 40 %) is off the critical path.
        // one block of NB chunks starting at chunk C0 (both compile-time: they index the register file)
        auto scan = [&](auto c0_, auto nb_) {
          constexpr int C0 = decltype(c0_)::value, NB = decltype(nb_)::value;
          if (cnt < nsample && C0 < nchunks) {
            unsigned long long mk[NB];
            bool hit[NB];                                                   // a lane's own bit of mk[i]: the same SGPR pair
#pragma unroll
            for (int i = 0; i < NB / 2; ++i) {
              const f2 dx = cx - px[C0 / 2 + i], dy = cy - py[C0 / 2 + i], dz = cz - pz[C0 / 2 + i];
              const f2 d2 = (dx * dx + dy * dy) + dz * dz;
              hit[2 * i] = d2.x < radius2;
              hit[2 * i + 1] = d2.y < radius2;
              mk[2 * i] = __ballot(hit[2 * i]);
              mk[2 * i + 1] = __ballot(hit[2 * i + 1]);
            }
            // The CU has ONE scalar unit for its four SIMDs: the counters of the previous form (profiles/r3:
            // 23.3 M scalar against 19.0 M vector instructions per launch, scalar pipe saturated) say the per-chunk
            // scalar tests (empty? still below nsample?) cost more than the vector work they skipped.  So the
            // compaction of a block is unconditional and purely vector: slot = base + rank in the mask (2 mbcnt), the
            // store address is SELECTED by the lane's hit bit (hit -> row[slot], miss -> a private dummy word), no
            // exec masking, no branch; the scalar unit only keeps the running count (popcount + add per chunk) and
            // takes one decision per block.  Slots past nsample land in the overshoot area of the row and are ignored.
            int base[NB + 1];
            base[0] = cnt;
#pragma unroll
            for (int i = 0; i < NB; ++i) base[i + 1] = base[i] + __popcll(mk[i]);
#pragma unroll
            for (int i = 0; i < NB; ++i) {
              // (a hand-written v_mbcnt / v_lshl_add / v_cndmask / ds_write sequence -- 4 vector + 3 scalar instructions
              // per chunk instead of the compiler's exec-masked 5 + 4 -- measured no faster: 56.9 vs 56.6 us, r4h)
              const int slot = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mk[i] >> 32),
                                                              __builtin_amdgcn_mbcnt_lo((unsigned)mk[i], (unsigned)base[i]));
              const int at = hit[i] ? slot : dummy_at;                      // v_cndmask on the compare's own SGPR pair
              wrow[at] = (C0 + i) * kWave + L;
            }
            cnt = base[NB];
          }
        };
        constexpr int HEAD = S::A + S::B + S::C;
        static_assert(HEAD <= R && (R - HEAD) % S::REST == 0, "the schedule must tile the R chunks");
        if constexpr (S::A > 0) scan(std::integral_constant<int, 0>{}, std::integral_constant<int, S::A>{});
        if constexpr (S::B > 0) scan(std::integral_constant<int, S::A>{}, std::integral_constant<int, S::B>{});
        if constexpr (S::C > 0) scan(std::integral_constant<int, S::A + S::B>{}, std::integral_constant<int, S::C>{});
        static_for<0, (R - HEAD) / S::REST>([&](auto k) {
          scan(std::integral_constant<int, HEAD + decltype(k)::value * S::REST>{}, std::integral_constant<int, S::REST>{});
        });
      } else {
#pragma unroll
        for (int i = 0; i < RP; ++i) {
          if (cnt < nsample && 2 * i < nchunks) {                            // one scalar test per pair
            const f2 dx = cx - px[i], dy = cy - py[i], dz = cz - pz[i];
            const f2 d2 = (dx * dx + dy * dy) + dz * dz;
            const bool h0 = d2.x < radius2, h1 = d2.y < radius2;
            const unsigned long long m0 = __ballot(h0), m1 = __ballot(h1);
            const int s0 = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m0 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m0, (unsigned)cnt));
            *(h0 ? wrow + s0 : dummy) = (2 * i) * kWave + L;
            cnt += __popcll(m0);
            const int s1 = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m1 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m1, (unsigned)cnt));
            *(h1 ? wrow + s1 : dummy) = (2 * i + 1) * kWave + L;
            cnt += __popcll(m1);
          }
        }
      }
      if (cnt > nsample) cnt = nsample;
      // same-wave LDS write -> read: the compiler's lgkmcnt wait orders them, no barrier needed
      const int first = cnt > 0 ? wrow[0] : 0;
      for (int t = L; t < nsample; t += kWave) o[j * nsample + t] = t < cnt ? wrow[t] : first;
     }
    }
    return;
  }

  // R == 1: a single (possibly partial) chunk in registers; R == 0: streamed from L1/L2
  float qx = 0.f, qy = 0.f, qz = 0.f;
  if (R == 1 && L < n) { qx = p[L * 3 + 0]; qy = p[L * 3 + 1]; qz = p[L * 3 + 2]; }
  for (int j0 = w; j0 < m; j0 += kCentresPerLoad * kWavesPerBlock) {
   const int jl = j0 + (L / 3) * kWavesPerBlock;
   const int cv = (L < 3 * kCentresPerLoad && jl < m) ? __float_as_int(q[jl * 3 + L % 3]) : 0;
   for (int jj = 0; jj < kCentresPerLoad; ++jj) {
    const int j = j0 + jj * kWavesPerBlock;
    if (j >= m) break;
    const float cx = __int_as_float(__builtin_amdgcn_readlane(cv, 3 * jj + 0));
    const float cy = __int_as_float(__builtin_amdgcn_readlane(cv, 3 * jj + 1));
    const float cz = __int_as_float(__builtin_amdgcn_readlane(cv, 3 * jj + 2));
    int cnt = 0, first = 0;
    for (int i = 0; i < nchunks && cnt < nsample; ++i) {
      const int k = i * kWave + L;
      const bool in = k < n;
      float x = qx, y = qy, z = qz;
      if (R == 0) {
        x = in ? p[k * 3 + 0] : 0.f;
        y = in ? p[k * 3 + 1] : 0.f;
        z = in ? p[k * 3 + 2] : 0.f;
      }
      const float dx = cx - x, dy = cy - y, dz = cz - z;
      const float d2 = (dx * dx + dy * dy) + dz * dz;
      const bool hit = in && (d2 < radius2);
      const unsigned long long mask = __ballot(hit);
      if (mask) take(mask, i, hit, cnt, first);
    }
    if (cnt > nsample) cnt = nsample;
    for (int t = L; t < nsample; t += kWave) o[j * nsample + t] = t < cnt ? row[t] : first;
   }
  }
}

// [r3] Clouds of at most 64 points (SA2: 32 points, 16 centres): ONE WAVE per object, four objects per workgroup.
// Lane L owns point L; for every centre the radius test gives one 64-bit mask, a hit lane's output slot is its
// rank in the mask (mbcnt) and it stores its index straight to idx[j][slot]; lanes cnt <= t < nsample store the
// first hit (or 0 when there is none).  No LDS, no barrier, a quarter of the workgroups of the block-per-object form
// (whose 5120 launches of 256 threads each did ~30 instructions of work per wave).
__global__ __launch_bounds__(kBlock) void ball_query_small_kernel(int b, int n, int m, float radius, int nsample,
                                                                  const float *__restrict__ new_xyz,
                                                                  const float *__restrict__ xyz, int32_t *__restrict__ idx,
                                                                  const int *__restrict__ n_obj_dev) {
  const int L = lane_id();
  const int obj = blockIdx.x * kWavesPerBlock + __builtin_amdgcn_readfirstlane(wave_id());
  if (obj >= b) return;
  if (n_obj_dev && obj >= *n_obj_dev) return;
  const float *p = xyz + (size_t)obj * n * 3;
  const float *q = new_xyz + (size_t)obj * m * 3;
  int32_t *o = idx + (size_t)obj * m * nsample;
  const float radius2 = radius * radius;
  const bool in = L < n;
  const float px = in ? p[L * 3 + 0] : 0.f, py = in ? p[L * 3 + 1] : 0.f, pz = in ? p[L * 3 + 2] : 0.f;
  for (int j0 = 0; j0 < m; j0 += kCentresPerLoad) {
    const int jl = j0 + L / 3;
    const int cv = (L < 3 * kCentresPerLoad && jl < m) ? __float_as_int(q[jl * 3 + L % 3]) : 0;
    for (int jj = 0; jj < kCentresPerLoad && j0 + jj < m; ++jj) {
      const int j = j0 + jj;
      const float cx = __int_as_float(__builtin_amdgcn_readlane(cv, 3 * jj + 0));
      const float cy = __int_as_float(__builtin_amdgcn_readlane(cv, 3 * jj + 1));
      const float cz = __int_as_float(__builtin_amdgcn_readlane(cv, 3 * jj + 2));
      const float dx = cx - px, dy = cy - py, dz = cz - pz;
      const float d2 = (dx * dx + dy * dy) + dz * dz;
      const bool hit = in && (d2 < radius2);
      const unsigned long long mask = __ballot(hit);
      const int cnt = __popcll(mask);
      const int first = cnt ? (__ffsll((long long)mask) - 1) : 0;
      const int slot = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
      if (hit && slot < nsample) o[j * nsample + slot] = L;
      for (int t = L; t < nsample; t += kWave)
        if (t >= cnt) o[j * nsample + t] = first;
    }
  }
}

// ------------------------------------------------------------------------------------------
// group_points: out[i,l,j,k] = points[i,l,idx[i,j,k]]     (src/group_points_gpu.cu:8-28)
//
// grid = (b, channel tiles).  A workgroup stages its [CT][n] slab of `points` in LDS with
// coalesced loads, then streams the output: thread (tx, ty) owns 4 consecutive (j,k) positions
// (one int4 of idx, loaded once) and walks the tile's channels, emitting one 16-byte store per
// channel -- a wave writes 1 KiB contiguous per instruction.  The gather itself is an LDS read.
// ------------------------------------------------------------------------------------------
template <bool VEC4>
__global__ __launch_bounds__(kBlock) void group_points_kernel(int b, int c, int n, int S, int CT,
                                                               int tx_count,
                                                               const float *__restrict__ points,
                                                               const int32_t *__restrict__ idx,
                                                               float *__restrict__ out) {
  extern __shared__ float gp_tile[];  // CT * n floats
  const int obj = blockIdx.x;
  const int c0 = blockIdx.y * CT;
  const int ct = (c - c0) < CT ? (c - c0) : CT;
  const float *src = points + ((size_t)obj * c + c0) * n;
  const int32_t *ix = idx + (size_t)obj * S;
  float *dst = out + ((size_t)obj * c + c0) * S;

  const int tile_elems = ct * n;
  if ((n & 3) == 0 && (reinterpret_cast<uintptr_t>(points) & 15) == 0) {
    const float4 *src4 = reinterpret_cast<const float4 *>(src);
    float4 *t4 = reinterpret_cast<float4 *>(gp_tile);
    for (int e = threadIdx.x; e < tile_elems / 4; e += kBlock) t4[e] = src4[e];
  } else {
    for (int e = threadIdx.x; e < tile_elems; e += kBlock) gp_tile[e] = src[e];
  }
  __syncthreads();

  const int tx = threadIdx.x % tx_count, ty = threadIdx.x / tx_count;
  const int ty_count = kBlock / tx_count;
  if (VEC4) {
    const int S4 = S >> 2;
    const int4 *ix4 = reinterpret_cast<const int4 *>(ix);
    for (int s4 = tx; s4 < S4; s4 += tx_count) {
      const int4 id = ix4[s4];
      for (int l = ty; l < ct; l += ty_count) {
        const float *rowp = gp_tile + l * n;
        float4 v;
        v.x = rowp[id.x]; v.y = rowp[id.y]; v.z = rowp[id.z]; v.w = rowp[id.w];
        typedef __attribute__((ext_vector_type(4))) float f32x4_nt;
        const f32x4_nt nv = {v.x, v.y, v.z, v.w};
        __builtin_nontemporal_store(nv, reinterpret_cast<f32x4_nt *>(dst + (size_t)l * S) + s4);   // streamed once, never re-read here
      }
    }
  } else {
    for (int s = tx; s < S; s += tx_count) {
      const int id = ix[s];
      for (int l = ty; l < ct; l += ty_count) dst[(size_t)l * S + s] = gp_tile[l * n + id];
    }
  }
}

// Fallback when one [1][n] row does not fit the LDS budget: plain cached gather.
__global__ __launch_bounds__(kBlock) void group_points_direct_kernel(
    int b, int c, int n, int S, const float *__restrict__ points, const int32_t *__restrict__ idx,
    float *__restrict__ out) {
  const long long total = (long long)b * c * S;
  for (long long e = (long long)blockIdx.x * kBlock + threadIdx.x; e < total;
       e += (long long)gridDim.x * kBlock) {
    const int s = (int)(e % S);
    const long long il = e / S;
    const int i = (int)(il / c);
    out[e] = points[il * n + idx[(long long)i * S + s]];
  }
}

// ------------------------------------------------------------------------------------------
// group_points_grad (src/group_points_gpu.cu:43-64), deterministic.
//
// The reference scatter-adds with atomicAdd in an undefined order.  Here each workgroup
// (object, channel tile) inverts idx into a CSR list per target point, STABLE in (j,k), then
// thread (l, target) sums its list sequentially: ascending (j,k), the order
// oracle_group_points_grad uses, hence bit-identical to it and run-to-run reproducible.
//   LDS: idx[S] | list[S] | offset[n+1] | cursor[n] | grad tile [CT][S+1]
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void group_points_grad_kernel(
    int b, int c, int n, int S, int CT, const float *__restrict__ grad_out,
    const int32_t *__restrict__ idx, float *__restrict__ grad_points) {
  extern __shared__ int32_t gg_lds[];
  int32_t *s_idx = gg_lds;              // S
  int32_t *s_list = s_idx + S;          // S
  int32_t *s_off = s_list + S;          // n + 1
  int32_t *s_cur = s_off + n + 1;       // n
  float *s_g = reinterpret_cast<float *>(s_cur + n);  // CT * (S + 1)
  const int obj = blockIdx.x;
  const int c0 = blockIdx.y * CT;
  const int ct = (c - c0) < CT ? (c - c0) : CT;
  const int tid = threadIdx.x;
  const int32_t *ix = idx + (size_t)obj * S;
  const float *g = grad_out + ((size_t)obj * c + c0) * S;
  float *gp = grad_points + ((size_t)obj * c + c0) * n;

  for (int e = tid; e < S; e += kBlock) s_idx[e] = ix[e];
  for (int e = tid; e <= n; e += kBlock) s_off[e] = 0;
  for (int e = tid; e < n; e += kBlock) s_cur[e] = 0;
  // grad tile, row pitch S+1 so that lanes reading one jk of different channels spread over banks
  for (int e = tid; e < ct * S; e += kBlock) {
    const int l = e / S, s = e - l * S;
    s_g[l * (S + 1) + s] = g[e];
  }
  __syncthreads();
  for (int e = tid; e < S; e += kBlock) atomicAdd(&s_off[s_idx[e] + 1], 1);  // integer: exact
  __syncthreads();
  if (tid < kWave) {  // exclusive scan of the counts by wave 0
    int carry = 0;
    for (int base = 0; base < n; base += kWave) {
      const int e = base + tid;
      int v = e < n ? s_off[e + 1] : 0;
      int incl = v;
      for (int d = 1; d < kWave; d <<= 1) {
        const int o = __shfl_up(incl, d, kWave);
        if (tid >= d) incl += o;
      }
      if (e < n) s_off[e + 1] = carry + incl;
      carry += __shfl(incl, kWave - 1, kWave);
    }
  }
  __syncthreads();
  if (tid < kWave) {  // stable fill by wave 0: 64 consecutive (j,k) per step, grouped by target
    for (int base = 0; base < S; base += kWave) {
      const int e = base + tid;
      const bool live = e < S;
      const int tgt = live ? s_idx[e] : -1;
      unsigned long long todo = __ballot(live);
      while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const int t0 = __shfl(tgt, leader, kWave);
        const unsigned long long same = __ballot(live && tgt == t0) & todo;
        if (live && tgt == t0) {
          const int rank = (int)__builtin_amdgcn_mbcnt_hi(
              (unsigned)(same >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)same, 0u));
          s_list[s_off[t0] + s_cur[t0] + rank] = e;
        }
        if (tid == leader) s_cur[t0] += __popcll(same);
        todo &= ~same;
      }
    }
  }
  __syncthreads();
  for (int e = tid; e < ct * n; e += kBlock) {
    const int l = e / n, tgt = e - l * n;
    const int beg = s_off[tgt], end = s_off[tgt + 1];
    const float *rowp = s_g + l * (S + 1);
    float acc = 0.f;
    for (int i = beg; i < end; ++i) acc += rowp[s_list[i]];
    gp[(size_t)l * n + tgt] = acc;
  }
}

// Fallback for shapes whose CSR does not fit LDS: unordered fp32 atomics like the reference.
__global__ __launch_bounds__(kBlock) void group_points_grad_atomic_kernel(
    int b, int c, int n, int S, const float *__restrict__ grad_out,
    const int32_t *__restrict__ idx, float *__restrict__ grad_points) {
  const long long total = (long long)b * c * S;
  for (long long e = (long long)blockIdx.x * kBlock + threadIdx.x; e < total;
       e += (long long)gridDim.x * kBlock) {
    const int s = (int)(e % S);
    const long long il = e / S;
    const int i = (int)(il / c);
    atomicAdd(grad_points + il * n + idx[(long long)i * S + s], grad_out[e]);
  }
}

// ------------------------------------------------------------------------------------------
// three_nn (src/interpolate_gpu.cu:9-59): thread per unknown point, known points broadcast from
// LDS.  The reference keeps its running bests in double initialised to 1e40 and compares the fp32
// distance against them; on fp32 inputs that is the same decision sequence as fp32 bests
// initialised to +inf (every finite d < 1e40; inf < 1e40 is false like inf < inf), and
// (float)1e40 == +inf is what gets stored when fewer than 3 known points exist.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void three_nn_kernel(int b, int n, int m,
                                                           const float *__restrict__ unknown,
                                                           const float *__restrict__ known,
                                                           float *__restrict__ dist2,
                                                           int32_t *__restrict__ idx) {
  constexpr int kTile = 1024;  // known points per LDS pass
  __shared__ float s_k[kTile * 3];
  const int obj = blockIdx.x;
  const float *u = unknown + (size_t)obj * n * 3;
  const float *kn = known + (size_t)obj * m * 3;
  const int j = blockIdx.y * kBlock + threadIdx.x;
  const bool live = j < n;
  const float ux = live ? u[j * 3 + 0] : 0.f, uy = live ? u[j * 3 + 1] : 0.f,
              uz = live ? u[j * 3 + 2] : 0.f;
  const float inf = __int_as_float(0x7F800000);
  float b1 = inf, b2 = inf, b3 = inf;
  int i1 = 0, i2 = 0, i3 = 0;
  for (int base = 0; base < m; base += kTile) {
    const int cnt = (m - base) < kTile ? (m - base) : kTile;
    __syncthreads();
    for (int e = threadIdx.x; e < cnt * 3; e += kBlock) s_k[e] = kn[base * 3 + e];
    __syncthreads();
    for (int k = 0; k < cnt; ++k) {
      const float x = s_k[k * 3 + 0], y = s_k[k * 3 + 1], z = s_k[k * 3 + 2];
      const float dx = ux - x, dy = uy - y, dz = uz - z;
      const float d = (dx * dx + dy * dy) + dz * dz;
      const int kk = base + k;
      if (d < b1) { b3 = b2; i3 = i2; b2 = b1; i2 = i1; b1 = d; i1 = kk; }
      else if (d < b2) { b3 = b2; i3 = i2; b2 = d; i2 = kk; }
      else if (d < b3) { b3 = d; i3 = kk; }
    }
  }
  if (live) {
    float *d2o = dist2 + ((size_t)obj * n + j) * 3;
    int32_t *io = idx + ((size_t)obj * n + j) * 3;
    d2o[0] = b1; d2o[1] = b2; d2o[2] = b3;
    io[0] = i1; io[1] = i2; io[2] = i3;
  }
}

// three_interpolate (src/interpolate_gpu.cu:72-101): ((p1*w1 + p2*w2) + p3*w3), unfused.
__global__ __launch_bounds__(kBlock) void three_interpolate_kernel(
    int b, int c, int m, int n, const float *__restrict__ points, const int32_t *__restrict__ idx,
    const float *__restrict__ weight, float *__restrict__ out) {
  const long long total = (long long)b * c * n;
  for (long long e = (long long)blockIdx.x * kBlock + threadIdx.x; e < total;
       e += (long long)gridDim.x * kBlock) {
    const int j = (int)(e % n);
    const long long il = e / n;
    const int i = (int)(il / c);
    const int32_t *ix = idx + ((long long)i * n + j) * 3;
    const float *w = weight + ((long long)i * n + j) * 3;
    const float *p = points + il * m;
    out[e] = (p[ix[0]] * w[0] + p[ix[1]] * w[1]) + p[ix[2]] * w[2];
  }
}

// three_interpolate_grad (src/interpolate_gpu.cu:116-143): unordered fp32 atomics like the
// reference (not on the GPS path).
__global__ __launch_bounds__(kBlock) void three_interpolate_grad_kernel(
    int b, int c, int n, int m, const float *__restrict__ grad_out,
    const int32_t *__restrict__ idx, const float *__restrict__ weight,
    float *__restrict__ grad_points) {
  const long long total = (long long)b * c * n;
  for (long long e = (long long)blockIdx.x * kBlock + threadIdx.x; e < total;
       e += (long long)gridDim.x * kBlock) {
    const int j = (int)(e % n);
    const long long il = e / n;
    const int i = (int)(il / c);
    const int32_t *ix = idx + ((long long)i * n + j) * 3;
    const float *w = weight + ((long long)i * n + j) * 3;
    float *gp = grad_points + il * m;
    const float go = grad_out[e];
    atomicAdd(gp + ix[0], go * w[0]);
    atomicAdd(gp + ix[1], go * w[1]);
    atomicAdd(gp + ix[2], go * w[2]);
  }
}


// ------------------------------------------------------------------------------------------
// pairwise object geometry (reference modules/utils.py:38-87, pairwise_rel_type 'center',
// spatial_dist_norm, spatial_dim 5):  for every ordered pair (l, t) of a scene
//     delta = c_l - c_t;  d = sqrt(dx^2 + dy^2 + dz^2 + eps);  d_xy = sqrt(dx^2 + dy^2 + eps)
//     out = [ d / d_max,  dz / d,  d_xy / d,  dy / d_xy,  dx / d_xy ],   d_max = max over all L*L pairs
// The reference runs ~15 elementwise/reduction launches over (B,L,L,*) tensors per step; here one
// workgroup per scene, two passes over the pairs from an LDS copy of the centres.  Operation order
// of the torch formulation ((x^2 + y^2) + z^2, then + eps; IEEE sqrt and divide; this file is
// compiled -ffp-contract=off): agrees with it to a few ulp (<= 1e-6 on features in [-1, 1]).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void pairwise_locs_kernel(int L, const float *__restrict__ centers, float eps,
                                                                float *__restrict__ out, _Float16 *__restrict__ planes,
                                                                int ld_pl) {
  extern __shared__ float pw_c[];                 // L * 3 centres | kWavesPerBlock partial maxima
  float *pw_red = pw_c + L * 3;
  const int scene = blockIdx.x;
  const float *c = centers + (size_t)scene * L * 3;
  for (int e = threadIdx.x; e < L * 3; e += kBlock) pw_c[e] = c[e];
  __syncthreads();
  const int pairs = L * L;
  // d_max = max sqrt(q) = sqrt(max q) exactly (IEEE sqrt is monotonic): the maximum runs over the squared distances, one
  // square root at the end; rows by wave, columns by lane (no integer division).  [r6] every slice workgroup of a scene
  // repeats this pass, so its cost is the kernel's: ~50 instructions per pair (division, IEEE sqrt) -> ~12
  float mq = 0.f;
  for (int l = wave_id(); l < L; l += kWavesPerBlock) {
    const float lx = pw_c[l * 3 + 0], ly = pw_c[l * 3 + 1], lz = pw_c[l * 3 + 2];
    for (int t = lane_id(); t < L; t += kWave) {
      const float dx = lx - pw_c[t * 3 + 0], dy = ly - pw_c[t * 3 + 1], dz = lz - pw_c[t * 3 + 2];
      const float q = ((dx * dx + dy * dy) + dz * dz) + eps;
      mq = q > mq ? q : mq;
    }
  }
  for (int off = 32; off >= 1; off >>= 1) {
    const float o = __shfl_xor(mq, off, kWave);
    mq = o > mq ? o : mq;
  }
  if (lane_id() == 0) pw_red[wave_id()] = mq;
  __syncthreads();
  float qmax = pw_red[0];
#pragma unroll
  for (int w = 1; w < kWavesPerBlock; ++w) qmax = pw_red[w] > qmax ? pw_red[w] : qmax;
  const float dmax = sqrtf(qmax);                  // d >= sqrt(eps) > 0
  float *o = out ? out + (size_t)scene * pairs * 5 : nullptr;
  // plane form (gps_attn_args.pl_planes): planes[scene][d][l][t], t contiguous (pitch ld_pl), fp16 round-to-nearest of the
  // same fp32 values; the pad columns L .. ld_pl - 1 are zero
  _Float16 *pp = planes ? planes + (size_t)scene * 5 * L * ld_pl : nullptr;
  const int cols = planes ? ld_pl : L;
  // [r6] gridDim.y workgroups share a scene's output (each found d_max over ALL pairs above: a maximum, so every slice
  // holds the same value): one workgroup per scene left 3/4 of the CUs idle at B = 64 (24 us -> 8)
  for (int e = blockIdx.y * kBlock + threadIdx.x; e < L * cols; e += gridDim.y * kBlock) {
    const int l = e / cols, t = e - l * cols;
    if (t >= L) {
#pragma unroll
      for (int d5 = 0; d5 < 5; ++d5) pp[((size_t)d5 * L + l) * ld_pl + t] = (_Float16)0.f;
      continue;
    }
    const float dx = pw_c[l * 3 + 0] - pw_c[t * 3 + 0], dy = pw_c[l * 3 + 1] - pw_c[t * 3 + 1],
                dz = pw_c[l * 3 + 2] - pw_c[t * 3 + 2];
    const float xy2 = dx * dx + dy * dy;
    const float d = sqrtf((xy2 + dz * dz) + eps);
    const float dxy = sqrtf(xy2 + eps);
    const float f[5] = {d / dmax, dz / d, dxy / d, dy / dxy, dx / dxy};
    if (o) {
      float *p = o + ((size_t)l * L + t) * 5;
#pragma unroll
      for (int d5 = 0; d5 < 5; ++d5) p[d5] = f[d5];
    }
    if (pp) {
#pragma unroll
      for (int d5 = 0; d5 < 5; ++d5) pp[((size_t)d5 * L + l) * ld_pl + t] = (_Float16)f[d5];
    }
  }
}

// slices of one scene's pairs: enough workgroups for ~4 per CU, each with >= 2 trips of output work
inline int pairwise_slices(int b, int L) {
  const long long trips = ((long long)L * L + kBlock - 1) / kBlock;
  long long s = (1024 + b - 1) / b;
  if (s > trips / 2) s = trips / 2;
  return s < 1 ? 1 : (s > 16 ? 16 : (int)s);
}

// planes from an existing (B, L, L, 5) fp32 tensor (callers that built the pairwise tensor themselves)
__global__ __launch_bounds__(kBlock) void pairwise_to_planes_kernel(int L, const float *__restrict__ pl, _Float16 *__restrict__ planes,
                                                                     int ld_pl) {
  const int scene = blockIdx.x;
  const float *src = pl + (size_t)scene * L * L * 5;
  _Float16 *pp = planes + (size_t)scene * 5 * L * ld_pl;
  for (int e = threadIdx.x; e < L * ld_pl; e += kBlock) {
    const int l = e / ld_pl, t = e - l * ld_pl;
#pragma unroll
    for (int d5 = 0; d5 < 5; ++d5)
      pp[((size_t)d5 * L + l) * ld_pl + t] = t < L ? (_Float16)src[((size_t)l * L + t) * 5 + d5] : (_Float16)0.f;
  }
}

// ------------------------------------------------------------------------------------------
// [r5] Distinct-cloud plan of a batch of object clouds.  The reference pads a scene to its maximum object count with
// CONSTANT clouds (data/datasets/dataset_wrapper.py:64-65: pad_tensors(obj_fts, lens=max_obj_len, pad=1.0)) and runs
// PointNet++ on every slot (modules/vision/pcd_openvocab_encoder.py:156-160): at the bench workload 37 % of the 5 120
// object slots are such pads, and every one of them gets the same features.  Per-object kernels are independent across
// objects, so the encoder can run on the objects that are NOT pads plus ONE pad representative, and every pad slot
// reads the representative's result: bit-identical outputs, a third less work in FPS / ball_query / the SA levels.
// "Pad" is decided here from the data, never from a mask: an object whose (n, ld) cloud is ONE 32-bit word repeated,
// with the same word as the first such object of the batch (other constant clouds are ordinary objects).
//   flag kernel     one workgroup per object: uniform? + the word
//   plan kernel     one workgroup: stable order [ordinary objects | first pad object], slot_of[] for the way back
//   copy kernel     compact xyz (slot, n, 3) and point-major features (slot, n, ld - 3) of the planned objects
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void cloud_uniform_kernel(int words, const uint32_t *__restrict__ cloud,
                                                                int32_t *__restrict__ uniform, uint32_t *__restrict__ word) {
  const int obj = blockIdx.x;
  const uint32_t *p = cloud + (size_t)obj * words;
  const uint32_t w0 = p[0];
  unsigned int diff = 0u;
  if ((words & 3) == 0 && (reinterpret_cast<uintptr_t>(cloud) & 15) == 0) {
    const uint4 *p4 = reinterpret_cast<const uint4 *>(p);
    const int n4 = words / 4;
    // [r6] first trip alone: an ordinary object differs from its first word within its first kBlock * 16 bytes and leaves
    // here having read 4 KB of its 24 KB; only pads (and clouds that start with 4 KB of one word) are read to the end, four
    // loads in flight per thread (indices clamped, not predicated: a predicated load waits for its own return)
    if ((int)threadIdx.x < n4) {
      const uint4 v = p4[threadIdx.x];
      diff = ((v.x ^ w0) | (v.y ^ w0)) | ((v.z ^ w0) | (v.w ^ w0));
    }
    if (__syncthreads_or(diff != 0u)) {
      if (threadIdx.x == 0) { uniform[obj] = 0; word[obj] = w0; }
      return;
    }
    for (int i = kBlock + (int)threadIdx.x; i < n4; i += 4 * kBlock) {
      uint4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = p4[min(i + u * kBlock, n4 - 1)];
#pragma unroll
      for (int u = 0; u < 4; ++u) diff |= ((v[u].x ^ w0) | (v[u].y ^ w0)) | ((v[u].z ^ w0) | (v[u].w ^ w0));
    }
  } else {
    for (int i = threadIdx.x; i < words; i += kBlock) diff |= p[i] ^ w0;
  }
  const int differs = __syncthreads_or(diff != 0u);
  if (threadIdx.x == 0) { uniform[obj] = differs ? 0 : 1; word[obj] = w0; }
}

__global__ __launch_bounds__(1024) void cloud_plan_kernel(int b, int rows_mult, const int32_t *__restrict__ uniform,
                                                           const uint32_t *__restrict__ word,
                                                           int32_t *__restrict__ obj_of, int64_t *__restrict__ slot_of,
                                                           int32_t *__restrict__ scal) {
  __shared__ int s_cnt[2][16];
  __shared__ int s_first_pad;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (threadIdx.x == 0) s_first_pad = 0x7FFFFFFF;
  __syncthreads();
  // one LDS atomic per wave and trip (a third of the bench batch's objects are pads: an atomic per pad lane was ~1 900
  // serialised updates of one word)
  for (int i0 = 0; i0 < b; i0 += 1024) {
    const int o = i0 + (int)threadIdx.x;
    const unsigned long long um = __ballot(o < b && uniform[o] != 0);
    if (um && lane == 0) atomicMin(&s_first_pad, i0 + w * 64 + (__ffsll((long long)um) - 1));
  }
  __syncthreads();
  const int fp = s_first_pad;
  const bool has_pad = fp != 0x7FFFFFFF;
  const uint32_t pad_word = has_pad ? word[fp] : 0u;
  auto is_pad = [&](int o) { return has_pad && uniform[o] != 0 && word[o] == pad_word; };
  int base = 0, par = 0;
  // [r6] the flags of eight trips are requested together: one trip at a time was a memory round trip in front of every
  // scan step (5 at b = 5 120: 16.5 us for 40 KB)
  for (int i00 = 0; i00 < b; i00 += 8 * 1024) {
    bool wk[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int o = i00 + u * 1024 + (int)threadIdx.x;
      wk[u] = o < b && !is_pad(o);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i0 = i00 + u * 1024;
      if (i0 >= b) break;                                // uniform
      const int o = i0 + threadIdx.x;
      const bool work = wk[u];
      const unsigned long long mask = __ballot(work);
      const int before = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
      if (lane == 0) s_cnt[par][w] = __popcll(mask);
      __syncthreads();
      int wave_base = 0, trip = 0;
      for (int k = 0; k < 16; ++k) { const int c = s_cnt[par][k]; wave_base += (k < w) ? c : 0; trip += c; }
      if (work) {
        const int slot = base + wave_base + before;
        obj_of[slot] = o;
        slot_of[o] = slot;
      }
      base += trip;
      par ^= 1;
    }
  }
  const int n_work = base + (has_pad ? 1 : 0);          // base = number of ordinary objects (the same in every thread)
  for (int o = threadIdx.x; o < b; o += 1024) {
    if (is_pad(o)) slot_of[o] = base;                   // every pad reads the representative's slot
    if (o >= n_work) obj_of[o] = has_pad ? fp : 0;      // slots past the extent name a valid object (never processed)
  }
  if (threadIdx.x == 0) {
    if (has_pad) obj_of[base] = fp;
    scal[0] = n_work;
    scal[1] = base;
    scal[2] = has_pad ? fp : -1;
    scal[3] = n_work * rows_mult;
  }
}

__global__ __launch_bounds__(kBlock) void cloud_copy_kernel(int n, int ld, const float *__restrict__ cloud,
                                                             const int32_t *__restrict__ obj_of,
                                                             const int32_t *__restrict__ scal, float *__restrict__ xyz_c,
                                                             float *__restrict__ feat_c) {
  const int slot = blockIdx.x;
  if (slot >= scal[0]) return;
  const int c_feat = ld - 3;
  const float *src = cloud + (size_t)obj_of[slot] * n * ld;
  float *dx = xyz_c + (size_t)slot * n * 3, *df = feat_c + (size_t)slot * n * c_feat;
  if (ld == 6 && (n & 1) == 0 && ((reinterpret_cast<uintptr_t>(cloud) & 15) | (reinterpret_cast<uintptr_t>(xyz_c) & 7) |
                                  (reinterpret_cast<uintptr_t>(feat_c) & 7)) == 0) {
    // [r6] xyz + rgb clouds, two points (48 bytes) per step: three 16-byte loads, six 8-byte stores, the loads of two steps
    // requested together (indices clamped, not predicated) -- the element loop below pays an integer division and a 4-byte
    // round trip per word (40 us for 155 MB)
    const float4 *s4 = reinterpret_cast<const float4 *>(src);
    float2 *x2 = reinterpret_cast<float2 *>(dx), *f2 = reinterpret_cast<float2 *>(df);
    const int pairs = n >> 1;
    for (int q0 = threadIdx.x; q0 < pairs; q0 += 2 * kBlock) {
      float4 a[2][3];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int q = min(q0 + u * kBlock, pairs - 1);
#pragma unroll
        for (int k = 0; k < 3; ++k) a[u][k] = s4[(size_t)q * 3 + k];
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int q = q0 + u * kBlock;
        if (q < pairs) {
          x2[(size_t)q * 3 + 0] = make_float2(a[u][0].x, a[u][0].y);
          x2[(size_t)q * 3 + 1] = make_float2(a[u][0].z, a[u][1].z);
          x2[(size_t)q * 3 + 2] = make_float2(a[u][1].w, a[u][2].x);
          f2[(size_t)q * 3 + 0] = make_float2(a[u][0].w, a[u][1].x);
          f2[(size_t)q * 3 + 1] = make_float2(a[u][1].y, a[u][2].y);
          f2[(size_t)q * 3 + 2] = make_float2(a[u][2].z, a[u][2].w);
        }
      }
    }
    return;
  }
  for (int e = threadIdx.x; e < n * ld; e += kBlock) {
    const int pnt = e / ld, c = e - pnt * ld;
    const float v = src[e];
    if (c < 3) dx[pnt * 3 + c] = v;
    else df[pnt * c_feat + (c - 3)] = v;
  }
}

static thread_local const int *g_object_extent = nullptr;      // per host thread: a launch issued from another thread (a loader
//                                                                  thread's point ops) must not inherit an extent it did not set
const int *object_extent() { return g_object_extent; }

}  // namespace gps

// ==========================================================================================
// C ABI
// ==========================================================================================
namespace {

thread_local char g_last_hip_error[256] = "";

int finish_launch() {
  const hipError_t e = hipGetLastError();
  if (e == hipSuccess) return GPS_OK;
  const char *s = hipGetErrorString(e);
  int i = 0;
  for (; s && s[i] && i < 255; ++i) g_last_hip_error[i] = s[i];
  g_last_hip_error[i] = 0;
  return GPS_ERR_LAUNCH;
}

// include/cuda_utils.h:15-19 of the reference: the block size that decides FPS tie order.
int ref_opt_n_threads(int work_size) {
  const int pow_2 = (int)(log((double)work_size) / log(2.0));
  int t = 1 << pow_2;
  if (t > 512) t = 512;
  if (t < 1) t = 1;
  return t;
}

int ilog2(int v) {
  int p = 0;
  while ((1 << (p + 1)) <= v) ++p;
  return p;
}

int grid_for(long long total, int per_block) {
  long long g = (total + per_block - 1) / per_block;
  const long long cap = 256LL * 8;  // 256 CUs x 8 resident workgroups; grid-stride beyond that
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

constexpr int kLdsBudget = 64 * 1024;  // per-workgroup LDS we allow ourselves (160 KiB per CU)

}  // namespace

extern "C" {

int gps_abi_version(void) { return GPS_HIP_ABI_VERSION; }

const char *gps_error_string(int status) {
  switch (status) {
    case GPS_OK: return "ok";
    case GPS_ERR_INVALID_ARGUMENT: return "invalid argument";
    case GPS_ERR_UNSUPPORTED: return "shape not supported by the gfx950 kernels";
    case GPS_ERR_LAUNCH: return "HIP launch failure (see gps_last_hip_error)";
    default: return "unknown status";
  }
}

const char *gps_last_hip_error(void) { return g_last_hip_error; }

static int fps_launch(int b, int n, int m, const float *dataset, float *temp, int32_t *idxs, float *centres,
                      gps_stream_t stream) {
  if (b < 0 || n < 0 || m < 0) return GPS_ERR_INVALID_ARGUMENT;
  if (b == 0 || m == 0) return GPS_OK;
  if (n < 1 || !dataset || !idxs) return GPS_ERR_INVALID_ARGUMENT;
  hipStream_t s = (hipStream_t)stream;
  const int bs = ref_opt_n_threads(n);
  const int p = ilog2(bs);
  const int Q = (n + bs - 1) / bs;
  const int need = p >= 6 ? (1 << (p - 6)) * Q : Q;
  const dim3 grid((b + gps::kWavesPerBlock - 1) / gps::kWavesPerBlock), block(gps::kBlock);
#define GPS_FPS_CASE(R_)                                                                       \
  hipLaunchKernelGGL(gps::fps_resident_kernel<R_>, grid, block, 0, s, b, n, m, p, Q, dataset, \
                     idxs, centres, gps::object_extent())
  if (need <= 1) GPS_FPS_CASE(1);
  else if (need <= 2) GPS_FPS_CASE(2);
  else if (need <= 4) GPS_FPS_CASE(4);
  else if (need <= 8) GPS_FPS_CASE(8);
  else if (need <= 16) GPS_FPS_CASE(16);
  else if (need <= 32) GPS_FPS_CASE(32);
  else {
    if (centres) return GPS_ERR_UNSUPPORTED;     // the streaming form (n > 2048) writes indices only
    if (!temp) return GPS_ERR_INVALID_ARGUMENT;  // streaming form needs the (b,n) scratch
    hipLaunchKernelGGL(gps::fps_streaming_kernel, dim3(b), dim3(512), 0, s, b, n, m, dataset, temp,
                       idxs);
  }
#undef GPS_FPS_CASE
  return finish_launch();
}

int gps_furthest_point_sampling(int b, int n, int m, const float *dataset, float *temp,
                                int32_t *idxs, gps_stream_t stream) {
  return fps_launch(b, n, m, dataset, temp, idxs, nullptr, stream);
}

int gps_furthest_point_sampling_xyz(int b, int n, int m, const float *dataset, int32_t *idxs, float *new_xyz,
                                    gps_stream_t stream) {
  if (b > 0 && m > 0 && !new_xyz) return GPS_ERR_INVALID_ARGUMENT;
  return fps_launch(b, n, m, dataset, nullptr, idxs, new_xyz, stream);
}

void gps_point_set_object_extent(const int *n_objects_dev) { gps::g_object_extent = n_objects_dev; }

int gps_cloud_compact(int b, int n, int ld, const float *cloud, int rows_mult, int32_t *flag_scratch, int32_t *obj_of,
                      long long *slot_of, int32_t *scal, float *xyz_c, float *feat_c, gps_stream_t stream) {
  if (b < 0 || n < 1 || ld < 3) return GPS_ERR_INVALID_ARGUMENT;
  if (!scal || (b > 0 && (!cloud || !flag_scratch || !obj_of || !slot_of || !xyz_c || (ld > 3 && !feat_c))))
    return GPS_ERR_INVALID_ARGUMENT;
  if ((long long)n * ld > 0x7FFFFFFFll) return GPS_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  uint32_t *word = reinterpret_cast<uint32_t *>(flag_scratch + b);
  if (b > 0)
    hipLaunchKernelGGL(gps::cloud_uniform_kernel, dim3(b), dim3(gps::kBlock), 0, s, n * ld, (const uint32_t *)cloud,
                       flag_scratch, word);
  hipLaunchKernelGGL(gps::cloud_plan_kernel, dim3(1), dim3(1024), 0, s, b, rows_mult, flag_scratch, word, obj_of,
                     (int64_t *)slot_of, scal);
  if (b > 0)
    hipLaunchKernelGGL(gps::cloud_copy_kernel, dim3(b), dim3(gps::kBlock), 0, s, n, ld, cloud, obj_of, scal, xyz_c, feat_c);
  return finish_launch();
}

int gps_gather_points(int b, int c, int n, int npoints, const float *points, const int32_t *idx,
                      float *out, gps_stream_t stream) {
  if (b < 0 || c < 0 || n < 0 || npoints < 0) return GPS_ERR_INVALID_ARGUMENT;
  const long long total = (long long)b * c * npoints;
  if (total == 0) return GPS_OK;
  if (!points || !idx || !out) return GPS_ERR_INVALID_ARGUMENT;
  hipLaunchKernelGGL(gps::gather_points_kernel, dim3(grid_for(total, gps::kBlock)),
                     dim3(gps::kBlock), 0, (hipStream_t)stream, b, c, n, npoints, points, idx, out);
  return finish_launch();
}

int gps_gather_points_grad(int b, int c, int n, int npoints, const float *grad_out,
                           const int32_t *idx, float *grad_points, gps_stream_t stream) {
  if (b < 0 || c < 0 || n < 0 || npoints < 0) return GPS_ERR_INVALID_ARGUMENT;
  hipStream_t s = (hipStream_t)stream;
  const size_t out_elems = (size_t)b * c * n;
  if (out_elems == 0) return GPS_OK;
  if (!grad_points) return GPS_ERR_INVALID_ARGUMENT;
  if (hipMemsetAsync(grad_points, 0, out_elems * sizeof(float), s) != hipSuccess)
    return finish_launch();
  const long long total = (long long)b * c * npoints;
  if (total == 0) return GPS_OK;
  if (!grad_out || !idx) return GPS_ERR_INVALID_ARGUMENT;
  hipLaunchKernelGGL(gps::gather_points_grad_kernel, dim3(grid_for(total, gps::kBlock)),
                     dim3(gps::kBlock), 0, s, b, c, n, npoints, grad_out, idx, grad_points);
  return finish_launch();
}

constexpr int kBqDefaultSched = 2;   // index into the GPS_BQ_SCHEDS table below: blocks of 2, 6, 8 chunks (53.1 vs 56.6 us, r4f)

int gps_ball_query(int b, int n, int m, float radius, int nsample, const float *new_xyz,
                   const float *xyz, int32_t *idx, gps_stream_t stream) {
  if (b < 0 || n < 0 || m < 0 || nsample < 0) return GPS_ERR_INVALID_ARGUMENT;
  if ((long long)b * m * nsample == 0) return GPS_OK;
  if (!new_xyz || !idx || (n > 0 && !xyz)) return GPS_ERR_INVALID_ARGUMENT;
  hipStream_t s = (hipStream_t)stream;
  // per wave: a row of nsample slots (+ 512 overshoot slots + 64 dummy words of the register-resident form)
  const size_t lds = (size_t)gps::kWavesPerBlock * (nsample + 9 * gps::kWave) * sizeof(int32_t);
  if (lds > (size_t)kLdsBudget) return GPS_ERR_UNSUPPORTED;
  const dim3 grid(b), block(gps::kBlock);
  const int need = (n + gps::kWave - 1) / gps::kWave;
#define GPS_BQ_CASE(R_)                                                                         \
  hipLaunchKernelGGL(gps::ball_query_kernel<R_>, grid, block, lds, s, b, n, m, radius, nsample, \
                     new_xyz, xyz, idx, gps::object_extent())
  // block schedule of the register-resident scan for 1024 / 2048-point clouds (gps::BqSched); GPS_BQ_SCHED selects one
  // of the compiled alternatives for tuning runs (tools/kernel_bench.py), the default is the measured best
#define GPS_BQ_SCHED_CASE(R_, A_, B_, C_, REST_)                                                                          \
  hipLaunchKernelGGL((gps::ball_query_kernel<R_, gps::BqSched<A_, B_, C_, REST_>>), grid, block, lds, s, b, n, m, radius, \
                     nsample, new_xyz, xyz, idx, gps::object_extent())
#define GPS_BQ_SCHEDS(R_)                                        \
  switch (sched) {                                               \
    case 1: GPS_BQ_SCHED_CASE(R_, 4, 4, 0, 8); break;            \
    case 2: GPS_BQ_SCHED_CASE(R_, 2, 6, 0, 8); break;            \
    case 3: GPS_BQ_SCHED_CASE(R_, 4, 4, 0, 4); break;            \
    case 4: GPS_BQ_SCHED_CASE(R_, 2, 2, 4, 8); break;            \
    case 5: GPS_BQ_SCHED_CASE(R_, 2, 2, 4, 4); break;            \
    case 6: GPS_BQ_SCHED_CASE(R_, 2, 6, 0, 4); break;            \
    default: GPS_BQ_SCHED_CASE(R_, 8, 0, 0, 8); break;           \
  }
  static const int sched = [] {
    const char *e = getenv("GPS_BQ_SCHED");
    return e ? atoi(e) : kBqDefaultSched;
  }();
  if (need <= 1) {
    hipLaunchKernelGGL(gps::ball_query_small_kernel, dim3((b + gps::kWavesPerBlock - 1) / gps::kWavesPerBlock), block, 0, s, b, n,
                       m, radius, nsample, new_xyz, xyz, idx, gps::object_extent());
  } else if (need <= 2) GPS_BQ_CASE(2);
  else if (need <= 4) GPS_BQ_CASE(4);
  else if (need <= 8) GPS_BQ_CASE(8);
  else if (need <= 16) { GPS_BQ_SCHEDS(16) }
  else if (need <= 32) { GPS_BQ_SCHEDS(32) }
  else GPS_BQ_CASE(0);
#undef GPS_BQ_CASE
#undef GPS_BQ_SCHED_CASE
#undef GPS_BQ_SCHEDS
  return finish_launch();
}

int gps_group_points(int b, int c, int n, int npoints, int nsample, const float *points,
                     const int32_t *idx, float *out, gps_stream_t stream) {
  if (b < 0 || c < 0 || n < 0 || npoints < 0 || nsample < 0) return GPS_ERR_INVALID_ARGUMENT;
  const long long S = (long long)npoints * nsample;
  if ((long long)b * c * S == 0) return GPS_OK;
  if (!points || !idx || !out || n == 0) return GPS_ERR_INVALID_ARGUMENT;
  if (S > 0x7fffffffLL) return GPS_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  const size_t row_bytes = (size_t)n * sizeof(float);
  if (row_bytes > (size_t)kLdsBudget) {
    const long long total = (long long)b * c * S;
    hipLaunchKernelGGL(gps::group_points_direct_kernel, dim3(grid_for(total, gps::kBlock)),
                       dim3(gps::kBlock), 0, s, b, c, n, (int)S, points, idx, out);
    return finish_launch();
  }
  // channel tile: as many rows as fit 32 KiB of LDS, but not so many that one workgroup writes
  // more than ~64 KiB (keeps >= ~8 workgroups per CU in flight at GPS shapes)
  int CT = (int)((32 * 1024) / row_bytes);
  const int by_out = (int)((64 * 1024) / (S * sizeof(float)));
  if (CT > by_out) CT = by_out;
  if (CT < 1) CT = 1;
  if (CT > c) CT = c;
  const bool vec4 = (S % 4) == 0 && ((reinterpret_cast<uintptr_t>(idx) & 15) == 0) &&
                    ((reinterpret_cast<uintptr_t>(out) & 15) == 0);
  const int work_x = vec4 ? (int)(S / 4) : (int)S;
  int tx = 1;
  while (tx < work_x && tx < gps::kBlock) tx <<= 1;  // power of two <= 256
  const dim3 grid(b, (c + CT - 1) / CT), block(gps::kBlock);
  const size_t lds = (size_t)CT * row_bytes;
  if (vec4)
    hipLaunchKernelGGL(gps::group_points_kernel<true>, grid, block, lds, s, b, c, n, (int)S, CT, tx,
                       points, idx, out);
  else
    hipLaunchKernelGGL(gps::group_points_kernel<false>, grid, block, lds, s, b, c, n, (int)S, CT,
                       tx, points, idx, out);
  return finish_launch();
}

int gps_group_points_grad(int b, int c, int n, int npoints, int nsample, const float *grad_out,
                          const int32_t *idx, float *grad_points, gps_stream_t stream) {
  if (b < 0 || c < 0 || n < 0 || npoints < 0 || nsample < 0) return GPS_ERR_INVALID_ARGUMENT;
  const size_t out_elems = (size_t)b * c * n;
  if (out_elems == 0) return GPS_OK;
  if (!grad_points) return GPS_ERR_INVALID_ARGUMENT;
  hipStream_t s = (hipStream_t)stream;
  const long long S = (long long)npoints * nsample;
  if (S == 0) {
    if (hipMemsetAsync(grad_points, 0, out_elems * sizeof(float), s) != hipSuccess)
      return finish_launch();
    return GPS_OK;
  }
  if (!grad_out || !idx) return GPS_ERR_INVALID_ARGUMENT;
  if (S > 0x7fffffffLL) return GPS_ERR_UNSUPPORTED;
  const size_t fixed = ((size_t)2 * S + 2 * (size_t)n + 1) * sizeof(int32_t);
  const size_t per_ch = ((size_t)S + 1) * sizeof(float);
  if (fixed + per_ch <= (size_t)kLdsBudget) {
    int CT = (int)(((size_t)kLdsBudget - fixed) / per_ch);
    if (CT > 16) CT = 16;
    if (CT > c) CT = c;
    const dim3 grid(b, (c + CT - 1) / CT), block(gps::kBlock);
    hipLaunchKernelGGL(gps::group_points_grad_kernel, grid, block, fixed + CT * per_ch, s, b, c, n,
                       (int)S, CT, grad_out, idx, grad_points);
    return finish_launch();
  }
  if (hipMemsetAsync(grad_points, 0, out_elems * sizeof(float), s) != hipSuccess)
    return finish_launch();
  const long long total = (long long)b * c * S;
  hipLaunchKernelGGL(gps::group_points_grad_atomic_kernel, dim3(grid_for(total, gps::kBlock)),
                     dim3(gps::kBlock), 0, s, b, c, n, (int)S, grad_out, idx, grad_points);
  return finish_launch();
}

int gps_three_nn(int b, int n, int m, const float *unknown, const float *known, float *dist2,
                 int32_t *idx, gps_stream_t stream) {
  if (b < 0 || n < 0 || m < 0) return GPS_ERR_INVALID_ARGUMENT;
  if ((long long)b * n == 0) return GPS_OK;
  if (!unknown || !dist2 || !idx || (m > 0 && !known)) return GPS_ERR_INVALID_ARGUMENT;
  const dim3 grid(b, (n + gps::kBlock - 1) / gps::kBlock), block(gps::kBlock);
  hipLaunchKernelGGL(gps::three_nn_kernel, grid, block, 0, (hipStream_t)stream, b, n, m, unknown,
                     known, dist2, idx);
  return finish_launch();
}

int gps_three_interpolate(int b, int c, int m, int n, const float *points, const int32_t *idx,
                          const float *weight, float *out, gps_stream_t stream) {
  if (b < 0 || c < 0 || m < 0 || n < 0) return GPS_ERR_INVALID_ARGUMENT;
  const long long total = (long long)b * c * n;
  if (total == 0) return GPS_OK;
  if (!points || !idx || !weight || !out) return GPS_ERR_INVALID_ARGUMENT;
  hipLaunchKernelGGL(gps::three_interpolate_kernel, dim3(grid_for(total, gps::kBlock)),
                     dim3(gps::kBlock), 0, (hipStream_t)stream, b, c, m, n, points, idx, weight,
                     out);
  return finish_launch();
}

int gps_pairwise_locs(int b, int l, const float *centers, float eps, float *out, gps_stream_t stream) {
  if (b < 0 || l < 0) return GPS_ERR_INVALID_ARGUMENT;
  if (b == 0 || l == 0) return GPS_OK;
  if (!centers || !out) return GPS_ERR_INVALID_ARGUMENT;
  if (l > 2048) return GPS_ERR_UNSUPPORTED;
  const size_t lds = ((size_t)l * 3 + gps::kWavesPerBlock) * sizeof(float);
  hipLaunchKernelGGL(gps::pairwise_locs_kernel, dim3(b, gps::pairwise_slices(b, l)), dim3(gps::kBlock), lds, (hipStream_t)stream, l,
                     centers, eps, out, (_Float16 *)nullptr, 0);
  return finish_launch();
}

int gps_pairwise_locs_planes(int b, int l, const float *centers, float eps, float *out, void *planes, int ld_pl,
                             gps_stream_t stream) {
  if (b < 0 || l < 0) return GPS_ERR_INVALID_ARGUMENT;
  if (b == 0 || l == 0) return GPS_OK;
  if (!centers || !planes || ld_pl < l || (ld_pl & 3)) return GPS_ERR_INVALID_ARGUMENT;
  if (l > 2048) return GPS_ERR_UNSUPPORTED;
  const size_t lds = ((size_t)l * 3 + gps::kWavesPerBlock) * sizeof(float);
  hipLaunchKernelGGL(gps::pairwise_locs_kernel, dim3(b, gps::pairwise_slices(b, l)), dim3(gps::kBlock), lds, (hipStream_t)stream, l,
                     centers, eps, out, (_Float16 *)planes, ld_pl);
  return finish_launch();
}

int gps_pairwise_to_planes(int b, int l, const float *pl, void *planes, int ld_pl, gps_stream_t stream) {
  if (b < 0 || l < 0) return GPS_ERR_INVALID_ARGUMENT;
  if (b == 0 || l == 0) return GPS_OK;
  if (!pl || !planes || ld_pl < l || (ld_pl & 3)) return GPS_ERR_INVALID_ARGUMENT;
  hipLaunchKernelGGL(gps::pairwise_to_planes_kernel, dim3(b), dim3(gps::kBlock), 0, (hipStream_t)stream, l, pl,
                     (_Float16 *)planes, ld_pl);
  return finish_launch();
}

int gps_three_interpolate_grad(int b, int c, int n, int m, const float *grad_out,
                               const int32_t *idx, const float *weight, float *grad_points,
                               gps_stream_t stream) {
  if (b < 0 || c < 0 || m < 0 || n < 0) return GPS_ERR_INVALID_ARGUMENT;
  hipStream_t s = (hipStream_t)stream;
  const size_t out_elems = (size_t)b * c * m;
  if (out_elems == 0) return GPS_OK;
  if (!grad_points) return GPS_ERR_INVALID_ARGUMENT;
  if (hipMemsetAsync(grad_points, 0, out_elems * sizeof(float), s) != hipSuccess)
    return finish_launch();
  const long long total = (long long)b * c * n;
  if (total == 0) return GPS_OK;
  if (!grad_out || !idx || !weight) return GPS_ERR_INVALID_ARGUMENT;
  hipLaunchKernelGGL(gps::three_interpolate_grad_kernel, dim3(grid_for(total, gps::kBlock)),
                     dim3(gps::kBlock), 0, s, b, c, n, m, grad_out, idx, weight, grad_points);
  return finish_launch();
}

}  // extern "C"
