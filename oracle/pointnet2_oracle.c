/*
 * pointnet2_oracle.c -- CPU restatement of the reference's pointnet2 `_ext` ops.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under sceneverse_amd/ may link, import or
 * call this file; only tests/, __graft_entry__.smoke() and bench.py's
 * `cpu_baseline` leg use it, and only as the checker / reported baseline.
 *
 * Every function restates, on one CPU thread, what the corresponding reference
 * device kernel computes.  Citations are into
 *   /root/reference/modules/third_party/pointnet2/_ext_src/
 * Arithmetic is pinned: fp32, every multiply/add individually rounded, in the
 * left-to-right order of the reference source expressions (build with
 * -ffp-contract=off; see oracle/Makefile).  SURVEY.md App. B.0 explains why the
 * pin is needed (compilers contract the reference expression differently).
 *
 * Parity pin status: the reference ships no golden vectors for these ops
 * (SURVEY.md 8c).  The oracle is pinned against the reference itself run on the
 * MI355X (oracle/_ref, the reference sources compiled unmodified): committed fixtures
 * tests/golden/point_ops_ref_gpu.pt (made by tests/golden/make_golden_gpu.py), checked on CPU by
 * tests/test_oracle_vs_golden_gpu.py and live on the GPU by tests/test_gpu_vs_reference_ext.py.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORACLE_API __attribute__((visibility("default")))

/* include/cuda_utils.h:13-19 -- TOTAL_THREADS 512, opt_n_threads(work) =
 * clamp(2^floor(log(work)/log(2)), 1, 512); the double log quotient is kept
 * verbatim because its truncation decides the block size, and the block size
 * decides FPS tie-breaking. */
ORACLE_API int oracle_opt_n_threads(int work_size) {
  const int pow_2 = (int)(log((double)work_size) / log(2.0));
  int t = 1 << pow_2;
  if (t > 512) t = 512;
  if (t < 1) t = 1;
  return t;
}

/* include/cuda_utils.h:21-28 -- opt_block_config(x, y). */
ORACLE_API void oracle_opt_block_config(int x, int y, int *bx, int *by) {
  const int xt = oracle_opt_n_threads(x);
  int yt = oracle_opt_n_threads(y);
  if (yt > 512 / xt) yt = 512 / xt;
  if (yt < 1) yt = 1;
  *bx = xt;
  *by = yt;
}

/* src/sampling_gpu.cu:8-20 gather_points_kernel: out[i,l,j] = points[i,l,idx[i,j]] */
ORACLE_API void oracle_gather_points(int b, int c, int n, int m, const float *points,
                                     const int32_t *idx, float *out) {
  for (int i = 0; i < b; ++i)
    for (int l = 0; l < c; ++l)
      for (int j = 0; j < m; ++j) {
        const int a = idx[(size_t)i * m + j];
        out[((size_t)i * c + l) * m + j] = points[((size_t)i * c + l) * n + a];
      }
}

/* src/sampling_gpu.cu:34-47 gather_points_grad_kernel: atomicAdd scatter into a
 * zero-initialised (b,c,n) buffer (src/sampling.cpp:52-54).  The reference's
 * summation order is nondeterministic; the oracle fixes ascending j. */
ORACLE_API void oracle_gather_points_grad(int b, int c, int n, int m, const float *grad_out,
                                          const int32_t *idx, float *grad_points) {
  memset(grad_points, 0, sizeof(float) * (size_t)b * c * n);
  for (int i = 0; i < b; ++i)
    for (int l = 0; l < c; ++l)
      for (int j = 0; j < m; ++j) {
        const int a = idx[(size_t)i * m + j];
        grad_points[((size_t)i * c + l) * n + a] += grad_out[((size_t)i * c + l) * m + j];
      }
}

/* src/sampling_gpu.cu:59-65 __update: slot idx1 takes slot idx2 only when
 * strictly greater. */
static void fps_update(float *dists, int *dists_i, int idx1, int idx2) {
  const float v1 = dists[idx1], v2 = dists[idx2];
  const int i1 = dists_i[idx1], i2 = dists_i[idx2];
  dists[idx1] = v1 > v2 ? v1 : v2; /* max(v1, v2) */
  dists_i[idx1] = v2 > v1 ? i2 : i1;
}

/* src/sampling_gpu.cu:69-173 furthest_point_sampling_kernel<block_size>, host
 * src/sampling.cpp:66-87 (idx zero-init, temp = 1e10) and dispatch :175-229
 * (block_size = opt_n_threads(n)).  Literal re-enactment: the per-thread strided
 * scan, then the shared-memory pairwise tree, thread by thread.  `temp` is the
 * (b,n) scratch the host wrapper allocates; the oracle allocates it itself. */
ORACLE_API void oracle_furthest_point_sampling(int b, int n, int m, const float *dataset,
                                               int32_t *idxs) {
  if (m <= 0) return;
  memset(idxs, 0, sizeof(int32_t) * (size_t)b * m);
  const int bs = oracle_opt_n_threads(n);
  float *temp = (float *)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1));
  float dists[512];
  int dists_i[512];
  for (int bi = 0; bi < b; ++bi) {
    const float *ds = dataset + (size_t)bi * n * 3;
    int32_t *out = idxs + (size_t)bi * m;
    for (int k = 0; k < n; ++k) temp[k] = 1e10f;
    int old = 0;
    out[0] = old;
    for (int j = 1; j < m; ++j) {
      const float x1 = ds[old * 3 + 0], y1 = ds[old * 3 + 1], z1 = ds[old * 3 + 2];
      for (int tid = 0; tid < bs; ++tid) {
        int besti = 0;
        float best = -1.0f;
        for (int k = tid; k < n; k += bs) {
          const float x2 = ds[k * 3 + 0], y2 = ds[k * 3 + 1], z2 = ds[k * 3 + 2];
          const float mag = (x2 * x2) + (y2 * y2) + (z2 * z2);
          if ((double)mag <= 1e-3) continue; /* double literal in the source, :101 */
          const float d = (x2 - x1) * (x2 - x1) + (y2 - y1) * (y2 - y1) + (z2 - z1) * (z2 - z1);
          const float d2 = d < temp[k] ? d : temp[k]; /* min(d, temp[k]) */
          temp[k] = d2;
          besti = d2 > best ? k : besti;
          best = d2 > best ? d2 : best;
        }
        dists[tid] = best;
        dists_i[tid] = besti;
      }
      for (int s = bs / 2; s >= 1; s >>= 1) /* :116-169, strides bs/2 .. 1 */
        for (int tid = 0; tid < s; ++tid) fps_update(dists, dists_i, tid, tid + s);
      old = dists_i[0];
      out[j] = old;
    }
  }
  free(temp);
}

/* src/ball_query_gpu.cu:9-44 query_ball_point_kernel; idx is zero-initialised by
 * the host wrapper (src/ball_query.cpp:19-21) so rows without a hit stay 0. */
ORACLE_API void oracle_ball_query(int b, int n, int m, float radius, int nsample,
                                  const float *new_xyz, const float *xyz, int32_t *idx) {
  memset(idx, 0, sizeof(int32_t) * (size_t)b * m * nsample);
  const float radius2 = radius * radius;
  for (int bi = 0; bi < b; ++bi) {
    const float *p = xyz + (size_t)bi * n * 3;
    const float *q = new_xyz + (size_t)bi * m * 3;
    int32_t *o = idx + (size_t)bi * m * nsample;
    for (int j = 0; j < m; ++j) {
      const float new_x = q[j * 3 + 0], new_y = q[j * 3 + 1], new_z = q[j * 3 + 2];
      for (int k = 0, cnt = 0; k < n && cnt < nsample; ++k) {
        const float x = p[k * 3 + 0], y = p[k * 3 + 1], z = p[k * 3 + 2];
        const float d2 =
            (new_x - x) * (new_x - x) + (new_y - y) * (new_y - y) + (new_z - z) * (new_z - z);
        if (d2 < radius2) {
          if (cnt == 0)
            for (int l = 0; l < nsample; ++l) o[j * nsample + l] = k;
          o[j * nsample + cnt] = k;
          ++cnt;
        }
      }
    }
  }
}

/* src/group_points_gpu.cu:8-28 group_points_kernel:
 * out[b,l,j,k] = points[b,l,idx[b,j,k]] */
ORACLE_API void oracle_group_points(int b, int c, int n, int npoints, int nsample,
                                    const float *points, const int32_t *idx, float *out) {
  for (int bi = 0; bi < b; ++bi) {
    const float *p = points + (size_t)bi * n * c;
    const int32_t *ix = idx + (size_t)bi * npoints * nsample;
    float *o = out + (size_t)bi * npoints * nsample * c;
    for (int l = 0; l < c; ++l)
      for (int j = 0; j < npoints; ++j)
        for (int k = 0; k < nsample; ++k)
          o[((size_t)l * npoints + j) * nsample + k] = p[(size_t)l * n + ix[j * nsample + k]];
  }
}

/* src/group_points_gpu.cu:43-64 group_points_grad_kernel: atomicAdd scatter into a
 * zero-initialised (b,c,n) buffer (src/group_points.cpp:48-50).  Oracle order:
 * ascending (j,k) per (l, target) -- the order the HIP kernel also uses, so the
 * two agree bit for bit; the reference itself is run-to-run nondeterministic. */
ORACLE_API void oracle_group_points_grad(int b, int c, int n, int npoints, int nsample,
                                         const float *grad_out, const int32_t *idx,
                                         float *grad_points) {
  memset(grad_points, 0, sizeof(float) * (size_t)b * c * n);
  for (int bi = 0; bi < b; ++bi) {
    const float *g = grad_out + (size_t)bi * npoints * nsample * c;
    const int32_t *ix = idx + (size_t)bi * npoints * nsample;
    float *gp = grad_points + (size_t)bi * n * c;
    for (int l = 0; l < c; ++l)
      for (int j = 0; j < npoints; ++j)
        for (int k = 0; k < nsample; ++k)
          gp[(size_t)l * n + ix[j * nsample + k]] += g[((size_t)l * npoints + j) * nsample + k];
  }
}

/* Same scatter accumulated in double, for tolerance checks against the
 * order-free mathematical sum. */
ORACLE_API void oracle_group_points_grad_f64(int b, int c, int n, int npoints, int nsample,
                                             const float *grad_out, const int32_t *idx,
                                             double *grad_points) {
  memset(grad_points, 0, sizeof(double) * (size_t)b * c * n);
  for (int bi = 0; bi < b; ++bi) {
    const float *g = grad_out + (size_t)bi * npoints * nsample * c;
    const int32_t *ix = idx + (size_t)bi * npoints * nsample;
    double *gp = grad_points + (size_t)bi * n * c;
    for (int l = 0; l < c; ++l)
      for (int j = 0; j < npoints; ++j)
        for (int k = 0; k < nsample; ++k)
          gp[(size_t)l * n + ix[j * nsample + k]] +=
              (double)g[((size_t)l * npoints + j) * nsample + k];
  }
}

/* src/interpolate_gpu.cu:9-59 three_nn_kernel: running bests kept in double,
 * initial 1e40, strict '<' cascade, outputs cast to float (1e40 -> +inf). */
ORACLE_API void oracle_three_nn(int b, int n, int m, const float *unknown, const float *known,
                                float *dist2, int32_t *idx) {
  for (int bi = 0; bi < b; ++bi) {
    const float *u = unknown + (size_t)bi * n * 3;
    const float *kn = known + (size_t)bi * m * 3;
    float *d2o = dist2 + (size_t)bi * n * 3;
    int32_t *io = idx + (size_t)bi * n * 3;
    for (int j = 0; j < n; ++j) {
      const float ux = u[j * 3 + 0], uy = u[j * 3 + 1], uz = u[j * 3 + 2];
      double best1 = 1e40, best2 = 1e40, best3 = 1e40;
      int besti1 = 0, besti2 = 0, besti3 = 0;
      for (int k = 0; k < m; ++k) {
        const float x = kn[k * 3 + 0], y = kn[k * 3 + 1], z = kn[k * 3 + 2];
        const float d = (ux - x) * (ux - x) + (uy - y) * (uy - y) + (uz - z) * (uz - z);
        if (d < best1) {
          best3 = best2; besti3 = besti2;
          best2 = best1; besti2 = besti1;
          best1 = d; besti1 = k;
        } else if (d < best2) {
          best3 = best2; besti3 = besti2;
          best2 = d; besti2 = k;
        } else if (d < best3) {
          best3 = d; besti3 = k;
        }
      }
      d2o[j * 3 + 0] = (float)best1;
      d2o[j * 3 + 1] = (float)best2;
      d2o[j * 3 + 2] = (float)best3;
      io[j * 3 + 0] = besti1;
      io[j * 3 + 1] = besti2;
      io[j * 3 + 2] = besti3;
    }
  }
}

/* src/interpolate_gpu.cu:72-101 three_interpolate_kernel:
 * out[l,j] = p[l,i1]*w1 + p[l,i2]*w2 + p[l,i3]*w3, in that order. */
ORACLE_API void oracle_three_interpolate(int b, int c, int m, int n, const float *points,
                                         const int32_t *idx, const float *weight, float *out) {
  for (int bi = 0; bi < b; ++bi) {
    const float *p = points + (size_t)bi * m * c;
    const int32_t *ix = idx + (size_t)bi * n * 3;
    const float *w = weight + (size_t)bi * n * 3;
    float *o = out + (size_t)bi * n * c;
    for (int l = 0; l < c; ++l)
      for (int j = 0; j < n; ++j) {
        const float w1 = w[j * 3 + 0], w2 = w[j * 3 + 1], w3 = w[j * 3 + 2];
        const int i1 = ix[j * 3 + 0], i2 = ix[j * 3 + 1], i3 = ix[j * 3 + 2];
        o[(size_t)l * n + j] =
            p[(size_t)l * m + i1] * w1 + p[(size_t)l * m + i2] * w2 + p[(size_t)l * m + i3] * w3;
      }
  }
}

/* src/interpolate_gpu.cu:116-143 three_interpolate_grad_kernel: three atomicAdds
 * per (l,j) into zero-initialised (b,c,m) (src/interpolate.cpp:88-90).  Oracle
 * order: ascending j, then t = 1,2,3. */
ORACLE_API void oracle_three_interpolate_grad(int b, int c, int n, int m, const float *grad_out,
                                              const int32_t *idx, const float *weight,
                                              float *grad_points) {
  memset(grad_points, 0, sizeof(float) * (size_t)b * c * m);
  for (int bi = 0; bi < b; ++bi) {
    const float *g = grad_out + (size_t)bi * n * c;
    const int32_t *ix = idx + (size_t)bi * n * 3;
    const float *w = weight + (size_t)bi * n * 3;
    float *gp = grad_points + (size_t)bi * m * c;
    for (int l = 0; l < c; ++l)
      for (int j = 0; j < n; ++j) {
        const float go = g[(size_t)l * n + j];
        for (int t = 0; t < 3; ++t)
          gp[(size_t)l * m + ix[j * 3 + t]] += go * w[j * 3 + t];
      }
  }
}
