// gps_objects.hip -- per-object input processing of the GPS data loader on MI355X (gfx950).
//
// Reference: ScanBase._obj_processing_post, data/datasets/base.py:697-740 (rotate, centre/size ->
// obj_locs, box, subsample num_points, centre, scale to the unit ball), the loader's colour scaling
// base.py:74-76 and the padding to max_obj_len + obj_masks of data/datasets/dataset_wrapper.py:62-70.
// The reference does this per object in numpy on the data-loader workers and ships the result
// (126 MB/step/GPU at B=64) over PCIe; here the scenes stay resident in HBM in their RAW form
// (xyz f32 + rgb u8, as 16-byte records or as two arrays; objects contiguous, CSR offsets) and ONE launch produces the
// batch-ready (rows, num_points, 6) f32 tensor + obj_locs + obj_boxes + obj_masks.
//
// One workgroup (256 threads) per output row (= object slot of a scene, or a padding slot).
//   pass 1  all k points of the object: rotate, sum / min / max  -> centre, size, box
//   pass 2  the num_points sampled points (indices given, or drawn on the device) kept in VGPRs:
//           mean -> max norm -> (x - mean) / max_dist, colours, written as f32
// HBM-bound byte work: k*16 B streamed once (the gathers of pass 2 then hit lines the stream left in
// L2) + num_points*24 B written.
// Arithmetic is float64 like the reference (numpy promotes [xyz f32 | rgb u8 / 127.5 - 1] to f64)
// and rounded to f32 at the end like the loader's `.float()`; only the summation ORDER of the two
// means differs from numpy's pairwise sum (<= 1 f32 ulp after rounding, tests/test_gpu_objects.py).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gps_hip.h"

namespace gps_obj {

constexpr int kBlock = 256;
constexpr int kWaves = kBlock / 64;
constexpr int kMaxPer = 8;                      // sampled points per thread: num_points <= 2048

__device__ __forceinline__ uint64_t mix64(uint64_t z) {          // splitmix64 finaliser
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

// Device sampler.  k < num_points: WITH replacement (np.random.choice(..., replace=True) semantics):
// an independent uniform index per draw.  k >= num_points: WITHOUT replacement: the first num_points
// images of a keyed pseudo-random PERMUTATION of [0, k) -- a 4-round Feistel network on the smallest
// even-width domain 2^(2h) >= k, cycle-walked back into [0, k) (a permutation of the domain restricted
// to a subset by cycle-walking is a permutation of the subset), so indices are distinct by construction.
__device__ __forceinline__ uint32_t sample_index(uint64_t key, uint32_t j, uint32_t k, bool replace) {
  if (replace) return (uint32_t)(((mix64(key ^ ((uint64_t)j << 1 | 1)) >> 32) * (uint64_t)k) >> 32);
  int bits = 32 - __clz((int)(k - 1) | 1);                       // bits needed for k-1 (>= 1)
  const int h = (bits + 1) >> 1;
  const uint32_t mask = (1u << h) - 1u;
  uint32_t x = j;
  do {
    uint32_t l = x >> h, r = x & mask;
#pragma unroll
    for (int round = 0; round < 4; ++round) {
      const uint32_t f = (uint32_t)mix64(key + 0xD1B54A32D192ED03ull * (uint64_t)(round + 1) + r) & mask;
      const uint32_t nl = r;
      r = l ^ f;
      l = nl;
    }
    x = (l << h) | r;
  } while (x >= k);
  return x;
}

__device__ __forceinline__ double wave_sum(double v) {
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ double wave_min(double v) {
  for (int off = 32; off >= 1; off >>= 1) v = fmin(v, __shfl_xor(v, off, 64));
  return v;
}
__device__ __forceinline__ double wave_max(double v) {
  for (int off = 32; off >= 1; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
  return v;
}

struct Rot {
  double m[9];
  bool on;
  __device__ __forceinline__ void apply(double &x, double &y, double &z) const {
    if (!on) return;
    // obj_pcd[:, :3] @ rot.T  (base.py:708): out_j = sum_c p_c * R[j][c]
    const double a = (x * m[0] + y * m[1]) + z * m[2];
    const double b = (x * m[3] + y * m[4]) + z * m[5];
    const double c = (x * m[6] + y * m[7]) + z * m[8];
    x = a; y = b; z = c;
  }
};

// REC16: `xyz` is an array of 16-byte records {f32 x, y, z; u8 r, g, b, pad} (one aligned 16-B load per
// point in the stream, and the sampled points -- colours included -- are gathered from lines the stream
// just pulled through L2); otherwise xyz (N,3) and the colours are separate arrays.
template <bool REC16>
__global__ __launch_bounds__(kBlock) void obj_processing_post_kernel(
    int n_points, const float *__restrict__ xyz, const uint8_t *__restrict__ rgb_u8,
    const float *__restrict__ rgb_f32, const int64_t *__restrict__ obj_offsets,
    const int32_t *__restrict__ row_obj, const int32_t *__restrict__ sample_idx, uint64_t seed,
    const float *__restrict__ rot, const int32_t *__restrict__ row_rot, float *__restrict__ obj_fts,
    float *__restrict__ obj_locs, float *__restrict__ obj_boxes, uint8_t *__restrict__ obj_masks) {
  __shared__ double s_red[kWaves][9];
  __shared__ double s_out[9];
  const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int obj = row_obj[row];
  float *fts = obj_fts + (size_t)row * n_points * 6;
  int64_t begin = 0, k64 = 0;
  if (obj >= 0) {
    begin = obj_offsets[obj];
    k64 = obj_offsets[obj + 1] - begin;
  }
  if (obj < 0 || k64 <= 0) {                    // padding slot: features 1.0, locations 0.0, mask off
    for (int i = tid; i < n_points * 6; i += kBlock) fts[i] = 1.0f;
    if (tid < 6) {
      obj_locs[(size_t)row * 6 + tid] = 0.0f;
      if (obj_boxes) obj_boxes[(size_t)row * 6 + tid] = 0.0f;
    }
    if (tid == 0 && obj_masks) obj_masks[row] = 0;
    return;
  }
  const uint32_t k = (uint32_t)k64;
  Rot R;
  R.on = rot != nullptr && row_rot != nullptr && row_rot[row] >= 0;
  if (R.on) {
    const float *r = rot + (size_t)row_rot[row] * 9;
#pragma unroll
    for (int i = 0; i < 9; ++i) R.m[i] = (double)r[i];
  }
  const float *p = xyz + (size_t)begin * (REC16 ? 4 : 3);
  const float4 *p4 = reinterpret_cast<const float4 *>(xyz) + begin;

  // ---- pass 1: centre / size / box over ALL points of the (rotated) object --------------------
  double sx = 0, sy = 0, sz = 0;
  double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  auto fold = [&](double x, double y, double z) {
    R.apply(x, y, z);
    sx += x; sy += y; sz += z;
    lo[0] = fmin(lo[0], x); lo[1] = fmin(lo[1], y); lo[2] = fmin(lo[2], z);
    hi[0] = fmax(hi[0], x); hi[1] = fmax(hi[1], y); hi[2] = fmax(hi[2], z);
  };
  uint32_t i = tid;
  if (REC16) {
    // four independent 16-byte loads in flight per thread: a 20 000-point object is 20 trips, not 79
    for (; i + 3u * kBlock < k; i += 4u * kBlock) {
      const float4 r0 = p4[i], r1 = p4[i + kBlock], r2 = p4[i + 2u * kBlock], r3 = p4[i + 3u * kBlock];
      fold(r0.x, r0.y, r0.z); fold(r1.x, r1.y, r1.z); fold(r2.x, r2.y, r2.z); fold(r3.x, r3.y, r3.z);
    }
    for (; i < k; i += kBlock) {
      const float4 r = p4[i];
      fold(r.x, r.y, r.z);
    }
  } else {
    for (; i < k; i += kBlock) fold(p[(size_t)i * 3], p[(size_t)i * 3 + 1], p[(size_t)i * 3 + 2]);
  }
  sx = wave_sum(sx); sy = wave_sum(sy); sz = wave_sum(sz);
#pragma unroll
  for (int c = 0; c < 3; ++c) { lo[c] = wave_min(lo[c]); hi[c] = wave_max(hi[c]); }
  if (lane == 0) {
    s_red[wave][0] = sx; s_red[wave][1] = sy; s_red[wave][2] = sz;
#pragma unroll
    for (int c = 0; c < 3; ++c) { s_red[wave][3 + c] = lo[c]; s_red[wave][6 + c] = hi[c]; }
  }
  __syncthreads();
  if (tid < 9) {
    double v = s_red[0][tid];
    for (int w = 1; w < kWaves; ++w)
      v = tid < 3 ? v + s_red[w][tid] : (tid < 6 ? fmin(v, s_red[w][tid]) : fmax(v, s_red[w][tid]));
    s_out[tid] = v;
  }
  __syncthreads();
  if (tid < 3) {
    const double mn = s_out[3 + tid], mx = s_out[6 + tid];
    obj_locs[(size_t)row * 6 + tid] = (float)(s_out[tid] / (double)k);
    obj_locs[(size_t)row * 6 + 3 + tid] = (float)(mx - mn);
    if (obj_boxes) {
      obj_boxes[(size_t)row * 6 + tid] = (float)((mx + mn) / 2.0);
      obj_boxes[(size_t)row * 6 + 3 + tid] = (float)(mx - mn);
    }
  }
  if (tid == 0 && obj_masks) obj_masks[row] = 1;
  __syncthreads();                               // s_red / s_out are reused below

  // ---- pass 2: the sampled points, held in registers -----------------------------------------
  const bool replace = k < (uint32_t)n_points;
  const uint64_t key = mix64(seed ^ mix64((uint64_t)row));   // per output row: a scene drawn twice differs
  double px[kMaxPer], py[kMaxPer], pz[kMaxPer];
  uint32_t src[kMaxPer];                         // sampled index, or (REC16) the record's colour word
  sx = sy = sz = 0;
#pragma unroll
  for (int i = 0; i < kMaxPer; ++i) {
    const int j = tid + i * kBlock;
    if (j < n_points) {
      uint32_t s = sample_idx ? (uint32_t)sample_idx[(size_t)row * n_points + j]
                              : sample_index(key, (uint32_t)j, k, replace);
      s = s < k ? s : k - 1;                     // a caller-supplied index is clamped, never trusted
      double x, y, z;
      if (REC16) {
        const float4 r = p4[s];
        x = r.x; y = r.y; z = r.z;
        src[i] = __float_as_uint(r.w);
      } else {
        x = p[(size_t)s * 3]; y = p[(size_t)s * 3 + 1]; z = p[(size_t)s * 3 + 2];
        src[i] = s;
      }
      R.apply(x, y, z);
      px[i] = x; py[i] = y; pz[i] = z;
      sx += x; sy += y; sz += z;
    }
  }
  sx = wave_sum(sx); sy = wave_sum(sy); sz = wave_sum(sz);
  if (lane == 0) { s_red[wave][0] = sx; s_red[wave][1] = sy; s_red[wave][2] = sz; }
  __syncthreads();
  if (tid < 3) {
    double v = s_red[0][tid];
    for (int w = 1; w < kWaves; ++w) v += s_red[w][tid];
    s_out[tid] = v / (double)n_points;
  }
  __syncthreads();
  const double mx_ = s_out[0], my_ = s_out[1], mz_ = s_out[2];
  double far = 0.0;
#pragma unroll
  for (int i = 0; i < kMaxPer; ++i) {
    const int j = tid + i * kBlock;
    if (j < n_points) {
      px[i] -= mx_; py[i] -= my_; pz[i] -= mz_;
      far = fmax(far, sqrt((px[i] * px[i] + py[i] * py[i]) + pz[i] * pz[i]));
    }
  }
  far = wave_max(far);
  if (lane == 0) s_red[wave][3] = far;
  __syncthreads();
  if (tid == 0) {
    double v = s_red[0][3];
    for (int w = 1; w < kWaves; ++w) v = fmax(v, s_red[w][3]);
    s_out[3] = v < 1e-6 ? 1.0 : v;               // tiny point clouds, i.e. padding (base.py:726-727)
  }
  __syncthreads();
  const double max_dist = s_out[3];
#pragma unroll
  for (int i = 0; i < kMaxPer; ++i) {
    const int j = tid + i * kBlock;
    if (j < n_points) {
      float *o = fts + (size_t)j * 6;
      o[0] = (float)(px[i] / max_dist);
      o[1] = (float)(py[i] / max_dist);
      o[2] = (float)(pz[i] / max_dist);
      const size_t c = ((size_t)begin + src[i]) * 3;
      if (REC16) {                               // colours came with the record (little-endian r, g, b)
        o[3] = (float)((double)(src[i] & 255u) / 127.5 - 1.0);
        o[4] = (float)((double)((src[i] >> 8) & 255u) / 127.5 - 1.0);
        o[5] = (float)((double)((src[i] >> 16) & 255u) / 127.5 - 1.0);
      } else if (rgb_u8) {                       // colors / 127.5 - 1 in f64 (uint8 promotes), then .float()
        o[3] = (float)((double)rgb_u8[c] / 127.5 - 1.0);
        o[4] = (float)((double)rgb_u8[c + 1] / 127.5 - 1.0);
        o[5] = (float)((double)rgb_u8[c + 2] / 127.5 - 1.0);
      } else {                                   // float32 colours stay float32 under numpy's rules
        o[3] = rgb_f32[c] / 127.5f - 1.0f;
        o[4] = rgb_f32[c + 1] / 127.5f - 1.0f;
        o[5] = rgb_f32[c + 2] / 127.5f - 1.0f;
      }
    }
  }
}

}  // namespace gps_obj

extern "C" int gps_obj_processing_post(int n_rows, int n_points, const float *xyz, const void *rgb, int rgb_is_u8,
                                       const int64_t *obj_offsets, const int32_t *row_obj,
                                       const int32_t *sample_idx, uint64_t seed, const float *rot,
                                       const int32_t *row_rot, float *obj_fts, float *obj_locs, float *obj_boxes,
                                       uint8_t *obj_masks, gps_stream_t stream) {
  if (n_rows < 0 || n_points <= 0) return GPS_ERR_INVALID_ARGUMENT;
  if (n_points > gps_obj::kBlock * gps_obj::kMaxPer) return GPS_ERR_UNSUPPORTED;
  if (n_rows == 0) return GPS_OK;
  if (!xyz || !obj_offsets || !row_obj || !obj_fts || !obj_locs) return GPS_ERR_INVALID_ARGUMENT;
  if ((rot == nullptr) != (row_rot == nullptr)) return GPS_ERR_INVALID_ARGUMENT;
  if (!rgb) {                                    // 16-byte records: colours travel in the 4th word
    if (((uintptr_t)xyz & 15u) != 0) return GPS_ERR_INVALID_ARGUMENT;
    hipLaunchKernelGGL(gps_obj::obj_processing_post_kernel<true>, dim3(n_rows), dim3(gps_obj::kBlock), 0,
                       (hipStream_t)stream, n_points, xyz, nullptr, nullptr, obj_offsets, row_obj, sample_idx, seed,
                       rot, row_rot, obj_fts, obj_locs, obj_boxes, obj_masks);
  } else {
    hipLaunchKernelGGL(gps_obj::obj_processing_post_kernel<false>, dim3(n_rows), dim3(gps_obj::kBlock), 0,
                       (hipStream_t)stream, n_points, xyz, rgb_is_u8 ? (const uint8_t *)rgb : nullptr,
                       rgb_is_u8 ? nullptr : (const float *)rgb, obj_offsets, row_obj, sample_idx, seed, rot,
                       row_rot, obj_fts, obj_locs, obj_boxes, obj_masks);
  }
  return hipGetLastError() == hipSuccess ? GPS_OK : GPS_ERR_LAUNCH;
}
