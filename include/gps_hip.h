/*
 * gps_hip.h -- C ABI of libgps_hip.so, the MI355X (gfx950) native ops of the GPS hot path.
 *
 * This is the drop-in boundary for the reference's native layer: the nine functions of the
 * pybind module `pointnet2._ext`
 *   /root/reference/modules/third_party/pointnet2/_ext_src/src/bindings.cpp:6-19
 * which are thin ATen wrappers around the `*_kernel_wrapper` C prototypes this header mirrors
 * (same argument order and meaning; one `stream` argument appended; `int` status returned
 * instead of the reference's print-and-exit(-1) of include/cuda_utils.h:30-39).
 *
 * Conventions (all entry points):
 *   - every pointer is a DEVICE pointer into HBM on the current HIP device; tensors are dense,
 *     row-major, fp32 / int32, exactly the layouts of the reference (include/utils.h:5-25);
 *   - the caller owns and allocates inputs, outputs and scratch; nothing is allocated here;
 *   - outputs are fully written by the call: no pre-zeroing is required (the reference's host
 *     wrappers zero-fill because their kernels accumulate or skip; these kernels do not);
 *   - `stream` is a hipStream_t (NULL = the default stream); calls are asynchronous;
 *   - no torch types, no exceptions, re-entrant, no global state;
 *   - return value: GPS_OK or a negative GPS_ERR_* (gps_error_string() explains it).
 */
#ifndef GPS_HIP_H_
#define GPS_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GPS_HIP_ABI_VERSION 11

#define GPS_OK 0
#define GPS_ERR_INVALID_ARGUMENT (-1) /* negative size, NULL pointer with non-empty tensor ...   */
#define GPS_ERR_UNSUPPORTED (-2)      /* shape outside what the kernels implement (documented)    */
#define GPS_ERR_LAUNCH (-3)           /* hipGetLastError() reported a launch failure             */

typedef void *gps_stream_t; /* hipStream_t */

#if defined(__GNUC__)
#define GPS_API __attribute__((visibility("default")))
#else
#define GPS_API
#endif

GPS_API int gps_abi_version(void);
GPS_API const char *gps_error_string(int status);
/* Text of the last HIP runtime error seen by a GPS_ERR_LAUNCH on this thread ("" if none). */
GPS_API const char *gps_last_hip_error(void);

/* Furthest point sampling.  Replaces furthest_point_sampling_kernel_wrapper
 * (src/sampling.cpp:11-13, kernel src/sampling_gpu.cu:69-173, dispatch :175-229).
 *   dataset (b,n,3) f32  ->  idxs (b,m) i32.
 * Semantics incl. the `mag <= 1e-3` skip and the block-size dependent tie-break of the
 * reference's shared-memory tree are reproduced exactly (bit-exact indices).
 * `temp` is the (b,n) f32 scratch of the reference prototype; it is only used when
 * n > GPS_FPS_MAX_RESIDENT_N (running distances then live in HBM) and may be NULL otherwise.
 * It needs no initialisation. */
#define GPS_FPS_MAX_RESIDENT_N 2048
GPS_API int gps_furthest_point_sampling(int b, int n, int m, const float *dataset, float *temp,
                                int32_t *idxs, gps_stream_t stream);
/* The same sampling, also writing the sampled points: new_xyz (b,m,3) f32 = dataset[i][idxs[i][j]] -- what
 * PointnetSAModule gets from transpose -> gather_operation -> transpose on the indices
 * (pointnet2_modules.py:47-54), out of the launch that picked them.  n <= GPS_FPS_MAX_RESIDENT_N
 * (GPS_ERR_UNSUPPORTED above it: use the two calls). */
GPS_API int gps_furthest_point_sampling_xyz(int b, int n, int m, const float *dataset, int32_t *idxs,
                                            float *new_xyz, gps_stream_t stream);

/* ---- distinct clouds only (frozen object encoder) ------------------------------------------------------------
 * The reference pads a scene to its maximum object count with CONSTANT clouds (data/datasets/dataset_wrapper.py:64-65:
 * pad_tensors(obj_fts, lens=max_obj_len, pad=1.0)) and runs PointNet++ on every slot
 * (modules/vision/pcd_openvocab_encoder.py:156-160); every pad gets the same features.  gps_cloud_compact decides from
 * the DATA which objects of cloud (b, n, ld) [xyz | ld - 3 feature columns] are pads -- one 32-bit word repeated over
 * the whole cloud, the same word as the first such object of the batch -- and lays out the work list: the other
 * objects in their order, then ONE pad representative (if any).
 *   obj_of (b) int32      object at each work slot (slots past the extent name a valid object)
 *   slot_of (b) int64     work slot whose result object o reads (every pad: the representative's)
 *   scal (4) int32        [work slots, ordinary objects, first pad object or -1, work slots * rows_mult]
 *   xyz_c (b, n, 3), feat_c (b, n, ld - 3) point-major     the clouds of the work slots (the rest is not written)
 *   flag_scratch (2 b) int32
 * gps_point_set_object_extent(p): while p is not NULL the per-object launches of gps_furthest_point_sampling[_xyz]
 * (register-resident form), gps_ball_query, gps_sa_mlp_forward*, gps_split3_points process objects [0, *p) only and
 * neither read nor write the others (p = scal of the plan; device memory; a setting of the CALLING HOST THREAD -- launches
 * issued by other threads never see it; gps_sa_mlp_set_products is process-wide -- set it around the encoder's launches,
 * reset it to NULL afterwards).  Per-object results do not depend on the other
 * objects of the batch, so result[slot_of[o]] is bit-identical to running every object. */
GPS_API int gps_cloud_compact(int b, int n, int ld, const float *cloud, int rows_mult, int32_t *flag_scratch,
                              int32_t *obj_of, long long *slot_of, int32_t *scal, float *xyz_c, float *feat_c,
                              gps_stream_t stream);
GPS_API void gps_point_set_object_extent(const int *n_objects_dev);

/* out[i,l,j] = points[i,l,idx[i,j]].  Replaces gather_points_kernel_wrapper
 * (src/sampling.cpp:4-6, kernel src/sampling_gpu.cu:8-20).
 *   points (b,c,n) f32, idx (b,npoints) i32 -> out (b,c,npoints) f32. */
GPS_API int gps_gather_points(int b, int c, int n, int npoints, const float *points, const int32_t *idx,
                      float *out, gps_stream_t stream);

/* Scatter-add adjoint of gps_gather_points.  Replaces gather_points_grad_kernel_wrapper
 * (src/sampling.cpp:7-9, kernel src/sampling_gpu.cu:34-47).
 *   grad_out (b,c,npoints), idx (b,npoints) -> grad_points (b,c,n), overwritten (zeroed here). */
GPS_API int gps_gather_points_grad(int b, int c, int n, int npoints, const float *grad_out,
                           const int32_t *idx, float *grad_points, gps_stream_t stream);

/* Ball query: per centre the first `nsample` points (ascending index) with d2 < radius^2,
 * remaining slots padded with the first hit, rows without a hit all 0.  Replaces
 * query_ball_point_kernel_wrapper (src/ball_query.cpp:4-6, kernel src/ball_query_gpu.cu:9-44).
 *   new_xyz (b,m,3), xyz (b,n,3) -> idx (b,m,nsample) i32.  Bit-exact indices. */
GPS_API int gps_ball_query(int b, int n, int m, float radius, int nsample, const float *new_xyz,
                   const float *xyz, int32_t *idx, gps_stream_t stream);

/* out[i,l,j,k] = points[i,l,idx[i,j,k]].  Replaces group_points_kernel_wrapper
 * (src/group_points.cpp:4-6, kernel src/group_points_gpu.cu:8-28).
 *   points (b,c,n), idx (b,npoints,nsample) -> out (b,c,npoints,nsample). */
GPS_API int gps_group_points(int b, int c, int n, int npoints, int nsample, const float *points,
                     const int32_t *idx, float *out, gps_stream_t stream);

/* Scatter-add adjoint of gps_group_points.  Replaces group_points_grad_kernel_wrapper
 * (src/group_points.cpp:8-10, kernel src/group_points_gpu.cu:43-64).
 *   grad_out (b,c,npoints,nsample), idx -> grad_points (b,c,n), overwritten.
 * Deterministic: each target sums its contributions in ascending (j,k) order (the reference's
 * atomicAdd order is undefined). */
GPS_API int gps_group_points_grad(int b, int c, int n, int npoints, int nsample, const float *grad_out,
                          const int32_t *idx, float *grad_points, gps_stream_t stream);

/* Three nearest neighbours (squared distances).  Replaces three_nn_kernel_wrapper
 * (src/interpolate.cpp:4-5, kernel src/interpolate_gpu.cu:9-59).
 *   unknown (b,n,3), known (b,m,3) -> dist2 (b,n,3) f32, idx (b,n,3) i32. */
GPS_API int gps_three_nn(int b, int n, int m, const float *unknown, const float *known, float *dist2,
                 int32_t *idx, gps_stream_t stream);

/* out[i,l,j] = sum_t points[i,l,idx[i,j,t]] * weight[i,j,t], t = 0,1,2 in that order.
 * Replaces three_interpolate_kernel_wrapper (src/interpolate.cpp:6-8, kernel
 * src/interpolate_gpu.cu:72-101).  points (b,c,m), idx/weight (b,n,3) -> out (b,c,n). */
GPS_API int gps_three_interpolate(int b, int c, int m, int n, const float *points, const int32_t *idx,
                          const float *weight, float *out, gps_stream_t stream);

/* Adjoint of gps_three_interpolate w.r.t. points.  Replaces three_interpolate_grad_kernel_wrapper
 * (src/interpolate.cpp:9-12, kernel src/interpolate_gpu.cu:116-143).
 *   grad_out (b,c,n), idx/weight (b,n,3) -> grad_points (b,c,m), overwritten (zeroed here). */
GPS_API int gps_three_interpolate_grad(int b, int c, int n, int m, const float *grad_out,
                               const int32_t *idx, const float *weight, float *grad_points,
                               gps_stream_t stream);

/* Pairwise object geometry of a scene.  Replaces calc_pairwise_locs (modules/utils.py:38-87) for
 * pairwise_rel_type 'center', spatial_dist_norm=True, spatial_dim=5:
 *   centers (b,l,3) f32 -> out (b,l,l,5) f32 = [d/d_max, dz/d, d_xy/d, dy/d_xy, dx/d_xy] of pair
 *   (l,t), d = sqrt(|c_l-c_t|^2 + eps), d_max = per-scene maximum over all l*l pairs.
 * Operation order of the reference's torch formulation; agrees with it to <= 1e-6 (features in [-1,1]). */
GPS_API int gps_pairwise_locs(int b, int l, const float *centers, float eps, float *out, gps_stream_t stream);
/* The same launch also writing the PLANE form the spatial attention kernels read (gps_attn_args.pl_planes):
 * planes (b, 5, l, ld_pl) fp16, planes[s][d][i][t] = out[s][i][t][d] rounded to nearest, columns l .. ld_pl - 1 zero;
 * ld_pl a multiple of 4 >= l.  out may be NULL (planes only). */
GPS_API int gps_pairwise_locs_planes(int b, int l, const float *centers, float eps, float *out, void *planes, int ld_pl,
                                     gps_stream_t stream);
/* planes from an existing (b, l, l, 5) fp32 pairwise tensor (callers that built it themselves). */
GPS_API int gps_pairwise_to_planes(int b, int l, const float *pl, void *planes, int ld_pl, gps_stream_t stream);

/* ---- fused set-abstraction level (frozen encoder) --------------------------------------------
 * Additions to the nine reference entry points: one launch for what the reference runs as
 * QueryAndGroup's gathers + `-=` + `cat` (modules/third_party/pointnet2/pointnet2_utils.py:345-356),
 * SharedMLP's 3 x (conv1x1 no-bias, BatchNorm2d in eval mode, ReLU) (pytorch_utils.py:11-36) and
 * max_pool2d over nsample (pointnet2_modules.py:65-71).  Valid when the encoder is frozen (BN on
 * running statistics, no gradient) -- pcd_openvocab_encoder.py:54-57,121-129, 35 of 37 configs.
 * The caller folds BN into the conv: w' = diag(gamma / sqrt(var + eps)) w,
 * shift = beta - mean * gamma / sqrt(var + eps), and packs each layer once with
 * gps_sa_mlp_pack_layer into one buffer [layer 1 | layer 2 | layer 3].
 * Arithmetic: fp32 MFMA (v_mfma_f32_32x32x2_f32), i.e. fp32 products and sums like the
 * reference's fp32 conv, in a different summation order (tolerance stated in the tests). */

/* floats of the packed weight buffer of an MLP c_in -> c1 -> c2 -> c3 (-1: unsupported widths;
 * c1, c2, c3 must be multiples of 32). */
GPS_API long long gps_sa_mlp_wpack_floats(int c_in, int c1, int c2, int c3);

/* floats of one packed layer c_in -> c_out (-1 unless c_out is a multiple of 32). */
GPS_API long long gps_sa_mlp_layer_floats(int c_in, int c_out);

/* Pack one layer.  w (c_out, c_in) row-major, BN-folded; shift (c_out); dst = the layer's slice of
 * the packed buffer ((c_out/32) * tile floats, see gps_sa_mlp_wpack_floats).  Device pointers. */
GPS_API int gps_sa_mlp_pack_layer(int c_in, int c_out, const float *w, const float *shift, float *dst,
                                  gps_stream_t stream);

/* xyz (b,n,3), new_xyz (b,npoint,3), features (b,c_feat,n), idx (b,npoint,nsample) from
 * gps_ball_query  ->  out (b,c3,npoint) = max_k relu(mlp([xyz[idx]-new_xyz ; features[idx]])).
 * Implemented shapes: nsample == 32 and (c_feat,c1,c2,c3) in {(3,64,64,128), (128,128,128,256)}
 * (the GPS encoder, modules/layers/pointnet.py:22-63); anything else returns GPS_ERR_UNSUPPORTED
 * and the host mirror runs the level unfused. */
GPS_API int gps_sa_mlp_forward(int b, int n, int npoint, int nsample, int c_feat, int c1, int c2, int c3,
                               const float *xyz, const float *new_xyz, const float *features,
                               const int32_t *idx, const float *wpack, float *out,
                               gps_stream_t stream);

/* Split-bf16 variant of the three calls above: each fp32 operand is carried as bf16 (hi, lo) and each
 * product as W_hi X_hi + W_hi X_lo + W_lo X_hi on v_mfma_f32_32x32x16_bf16 with fp32 accumulation
 * (~2^-16 relative error per product, ~5x fewer matrix-pipe cycles than the fp32 MFMA form).  The
 * packed buffer has its own format: pack with gps_sa_mlp_pack_layer_bf16x3. */
GPS_API long long gps_sa_mlp_layer_floats_bf16x3(int c_in, int c_out);
/* Products per fp32 multiply-accumulate of gps_sa_mlp_forward_bf16x3[_pm]: 3 (default) = W_hi X_hi + W_hi X_lo + W_lo X_hi,
 * 2^-16 relative per product: within 1e-4 of the fp32 op-by-op path; 1 = W_hi X_hi only: bf16 operands, fp32
 * accumulation -- what torch's bf16 autocast computes for the reference's Conv2d stacks (pytorch_utils.py:11-36), a
 * third of the MFMA work, features within 2e-2 of the fp32 path's scale (tests/test_gpu_sa_fused.py states the
 * measured figure).  Opt-in, process-wide; anything else = query.  Returns the previous setting. */
GPS_API int gps_sa_mlp_set_products(int n);
GPS_API int gps_sa_mlp_pack_layer_bf16x3(int c_in, int c_out, const float *w, const float *shift, float *dst,
                                         gps_stream_t stream);
GPS_API int gps_sa_mlp_forward_bf16x3(int b, int n, int npoint, int nsample, int c_feat, int c1, int c2,
                                      int c3, const float *xyz, const float *new_xyz, const float *features,
                                      const int32_t *idx, const float *wpack, float *out,
                                      gps_stream_t stream);
/* the same launch with the features given POINT-major: features_pm[(obj * n + p) * ld_feat + c], c < c_feat -- e.g. the
 * colour columns of the interleaved (B, N, 3 + C) cloud the reference hands to break_up_pc
 * (modules/layers/pointnet.py:10-17; pointer at column 3, ld_feat = 3 + C): no transposed copy of the cloud in HBM.
 * First level only (c_feat == 3, 64-64-128). */
GPS_API int gps_sa_mlp_forward_bf16x3_pm(int b, int n, int npoint, int nsample, int c_feat, int c1, int c2, int c3,
                                         const float *xyz, const float *new_xyz, const float *features_pm,
                                         long long ld_feat, const int32_t *idx, const float *wpack, float *out,
                                         gps_stream_t stream);

/* ---- fused self-attention core (object-level spatial transformer, joint text+object transformer) --
 * One launch for what the reference runs between the QKV projections and the output projection:
 *   modules/layers/transformers.py:193-239  MultiHeadAttentionSpatial.forward, fusion 'cond':
 *       probs = softmax(log(clamp(sigmoid(w . pairwise + b), 1e-6)) + q k^T / sqrt(64)), masked keys
 *       get probability 0; out = probs v     (w, b = lang_cond_fc(x) per token and head)
 *   modules/layers/transformers.py:141 (torch.nn.MultiheadAttention, key_padding_mask, dropout on
 *       the probabilities) -- the same core without the spatial term (sw == pl == NULL).
 * q, k, v: bf16 (B, L, ld_qkv) views, head h in columns [64 h, 64 h + 64) -- three column blocks of
 * one packed projection output are fine (ld_qkv = its row pitch, a multiple of 8).  sw (B,L,H*6)
 * fp32: per token and head [b, w_1..w_5].  pl (B,L,L,5) fp32 (calc_pairwise_locs,
 * modules/utils.py:38-87).  mask (B,L) bytes, 1 = padded key.  p_drop/seed: dropout on the
 * probabilities (counter-based, reproducible between forward and backward); seed_dev: optional
 * device pointer to one uint64 that is added to `seed` when the kernel runs (lets a captured HIP
 * graph draw a fresh mask on every replay), NULL to use `seed` alone.
 * out (B, L, ld_o) bf16; lse (B,H,L) fp32 log-sum-exp of the logits (saved for backward).
 * bf16 MFMA with fp32 accumulation; softmax in fp32.  head_dim must be 64, L <= 512 (rows of more than 144
 * tokens stream their key chunks through a two-pass softmax instead of holding the whole score row). */
GPS_API int gps_attn_forward(int B, int H, int L, int head_dim, const void *q, const void *k, const void *v,
                             int ld_qkv, const float *sw, const float *pl, const unsigned char *mask,
                             float p_drop, unsigned long long seed, const void *seed_dev, void *out,
                             int ld_o, float *lse, gps_stream_t stream);

/* Gradients of gps_attn_forward: dout (B,L,ld_o) bf16 -> dq, dk, dv (bf16, same layout/pitch as
 * q, k, v) and dsw (B,L,H*6) fp32 (when sw != NULL).  pl and mask carry no gradient (inputs of the
 * data pipeline).  Probabilities are recomputed from lse.  out = the forward output (B,L,ld_o) bf16: required for
 * L > 256 (the streaming kernels take delta = rowsum(dout * out) from it), optional (may be NULL) below that. */
/* Which kernel family serves a sequence length (in 16-token tiles: a row of L tokens has ceil(L / 16)): lengths of
 * at least `plain` (no pairwise term) / `spatial` (with it) tiles take the streaming kernels, shorter ones the
 * register-resident kernels.  Defaults 1 and 10.  Results agree to bf16 rounding either way; process-wide. */
GPS_API void gps_attn_set_stream_min_tiles(int plain, int spatial);
/* Which kernels serve the PLAIN form (no pairwise term), bf16 -- a bit mask:
 *   1  forward calls on the block-streaming kernels of gps_attention_fa.hip (64 queries per workgroup, the keys streamed
 *      through LDS in 64-row blocks, online softmax; any length, variable-length batches, cross-attention),
 *   2  backward calls on them too (two launches: dQ per query block, dK / dV per key block; needs `out` and `delta_ws`),
 *   4  fixed-length self-attention up to 144 tokens on the K / V-resident kernels of gps_attention_sp.hip (one query strip
 *      per wave, probabilities and dS parked in LDS for the dK / dV pass: every score evaluated once).
 * Calls no set bit covers take the whole-sequence kernels of gps_attention.hip.  Default 1 | 4 (measured, profiles/r5:
 * the block-streaming backward ties the whole-sequence one on the variable-length text batches).  All families share lse
 * and the dropout stream, so forward and backward may come from different ones.  mode < 0 = query.  Returns the previous
 * mode.  Process-wide. */
GPS_API int gps_attn_set_plain_blocks(int mode);
GPS_API int gps_attn_backward(int B, int H, int L, int head_dim, const void *q, const void *k, const void *v,
                              int ld_qkv, const float *sw, const float *pl, const unsigned char *mask,
                              float p_drop, unsigned long long seed, const void *seed_dev,
                              const void *dout, int ld_o, const float *lse, const void *out, void *dq, void *dk,
                              void *dv, float *dsw, gps_stream_t stream);

/* ---- general form of the attention core: cross-attention, fp32 operands, fp8 products ---------------
 * One argument block for everything gps_attn_forward / gps_attn_backward do, plus
 *   - CROSS-ATTENTION (queries from `tgt`, keys / values from `memory`, Lq != Lk): the core of
 *       nn.MultiheadAttention(tgt, memory, memory, key_padding_mask=memory_key_padding_mask) in CrossAttentionLayer,
 *       TransformerDecoderLayer and TransformerSpatialDecoderLayer (modules/layers/transformers.py:12-63, 66-112,
 *       242-282).  q (B, Lq, ld_q), k / v (B, Lk, ld_kv), mask (B, Lk) over the keys, out (B, Lq, ld_o),
 *       lse (B, H, Lq).  The spatial term (sw, pl) needs Lq == Lk.
 *   - dtype GPS_ATTN_F32: q, k, v, out, dout, dq, dk, dv are fp32 and every product runs on the fp32 MFMA
 *       (v_mfma_f32_16x16x4_f32, bitwise an fmaf chain): the fp32 "master" path for parity runs against the
 *       reference's fp32 mathematics (transformers.py:193-239).  Lq, Lk <= 256.
 *   - compute GPS_ATTN_COMPUTE_FP8 (forward only, dtype bf16): Q K^T and P V on the OCP e4m3 MFMA
 *       (v_mfma_f32_16x16x32_fp8_fp8) with per-query-row scales for Q, one scale per (scene, head) tile for K and
 *       for V, probabilities quantised as 128 p; softmax, spatial term and accumulation stay fp32 (BASELINE
 *       configs[4]: 256 objects + 256 tokens).  The backward call always runs its products in the operand dtype.
 * Dropout / seeds / lse as in gps_attn_forward.  head_dim must be 64; bf16: Lq, Lk <= 512. */
#define GPS_ATTN_BF16 0
#define GPS_ATTN_F32 1
#define GPS_ATTN_COMPUTE_NATIVE 0
#define GPS_ATTN_COMPUTE_FP8 1
typedef struct gps_attn_args {
  int B, H, Lq, Lk, head_dim;
  int dtype;                 /* GPS_ATTN_BF16 | GPS_ATTN_F32: element type of q, k, v, out, dout, dq, dk, dv */
  int compute;               /* GPS_ATTN_COMPUTE_NATIVE | GPS_ATTN_COMPUTE_FP8 */
  int reserved;              /* 0 */
  const void *q;  int ld_q;  /* (B, Lq, ld_q), head h in columns [64 h, 64 h + 64) */
  const void *k, *v; int ld_kv;   /* (B, Lk, ld_kv) */
  const float *sw;           /* (B, Lq, H * 6) or NULL */
  const float *pl;           /* (B, Lq, Lk, 5) or NULL (with sw) */
  const unsigned char *mask; /* (B, Lk), 1 = padded key, or NULL */
  float p_drop; unsigned long long seed; const void *seed_dev;
  void *out; int ld_o;       /* forward: written; backward: the forward output (read) */
  float *lse;                /* (B, H, Lq): written by forward, read by backward */
  /* backward only */
  const void *dout;          /* (B, Lq, ld_o) */
  void *dq; int ld_dq;       /* (B, Lq, ld_dq) */
  void *dk, *dv; int ld_dkv; /* (B, Lk, ld_dkv) */
  float *dsw;                /* (B, Lq, H * 6) when sw != NULL */
  /* packed VARIABLE-LENGTH self-attention (bf16, plain form, no mask needed): B sequences stored back to back, sequence
   * b = rows [cu_rows[b], cu_rows[b + 1]) of q / k / v / out / dout / dq / dk / dv (B + 1 int32 on the device, lengths
   * may be 0); Lq == Lk = the CAPACITY (an upper bound of every length; sizes LDS and is the pitch of lse (B, H, Lq)).
   * NULL = fixed-length batch.  Work and traffic scale with the real lengths (sum L_b^2), not with B * Lq^2. */
  const int *cu_rows;
  /* with cu_rows: optional (B) int32 permutation of the sequences (device); workgroups are dispatched in block order,
   * so listing the longest sequences first balances the tail of the launch.  NULL = natural order. */
  const int *seq_order;
  /* with cu_rows: optional (B) int32 (device): only the first min(length_b, q_limit[b]) QUERY rows of sequence b are
   * computed -- forward leaves the other out / lse rows unwritten, backward takes their dout as zero (dq rows zero-filled,
   * no contribution to dk / dv).  For consumers that read a sequence at a few leading rows only (a caption read at
   * [CLS] in its last layer).  NULL = every row. */
  const int *q_limit;
  /* PLANE FORM of the spatial term (bf16 self-attention, Lq == Lk <= 144, p_drop == 0; replaces sw / pl / dsw, which
   * must then be NULL): pl_planes (B, 5, Lq, ld_pl) fp16 = the pairwise tensor as five planes, pl_planes[b][d][l][t] =
   * pl[b][l][t][d] (gps_pairwise_locs_planes / gps_pairwise_to_planes write it; ld_pl a multiple of 4 >= Lk, base 8-byte
   * aligned, columns >= Lk finite); sw16 = the conditioning vector in bf16, row (b, l) at sw16 + (b Lq + l) ld_sw, head h
   * at + 6 h (i.e. read in place from the packed projection output; ld_sw even); dsw16 / ld_dsw (backward): its
   * gradient, bf16, addressed the same way.  Served by gps_attention_sp.hip: every
   * operand of a query strip is requested at kernel entry, results leave as 8-byte stores. */
  const void *pl_planes; int ld_pl;
  const void *sw16; int ld_sw;
  void *dsw16; int ld_dsw;
  /* backward of the PLAIN form (no pairwise term), block-streaming kernels (gps_attention_fa.hip): (B, H, Lq) fp32
   * scratch -- the dQ launch writes delta = rowsum(dout * out) per query, the dK / dV launch reads it.  NULL: the
   * backward call takes the whole-sequence kernels of gps_attention.hip instead (same results to bf16 rounding). */
  float *delta_ws;
} gps_attn_args;
GPS_API int gps_attn_forward_ex(const gps_attn_args *args, gps_stream_t stream);
GPS_API int gps_attn_backward_ex(const gps_attn_args *args, gps_stream_t stream);

/* ---- row-sparse cross-entropy (masked-LM head) ----------------------------------------------------
 * Replaces the F.cross_entropy(..., ignore_index=-1) of lm_cls_loss (optim/loss/loss.py:56-61) over
 * txt_lm_cls_logits (modules/heads/pretrain_head.py:22-56).  logits (n_rows, ld >= vocab) bf16
 * (logits_bf16 != 0) or fp32; labels int64; rows whose label == ignore_index (or is out of range)
 * contribute loss 0 and gradient 0 and are never read.  loss_rows[n] = lse[n] - logits[n][label];
 * the caller takes sum(loss_rows) / count(valid) (the reference's mean over non-ignored targets). */
GPS_API int gps_masked_ce_forward(int n_rows, int vocab, int logits_bf16, const void *logits, long long ld,
                                  const long long *labels, long long ignore_index, float *loss_rows,
                                  float *lse, gps_stream_t stream);
/* dlogits[n][v] = (softmax(logits[n])[v] - [v == label]) * grad_rows[n], 0 for ignored rows; same
 * dtype as logits, row pitch ldd. */
GPS_API int gps_masked_ce_backward(int n_rows, int vocab, int logits_bf16, const void *logits, long long ld,
                                   const long long *labels, long long ignore_index, const float *lse,
                                   const float *grad_rows, void *dlogits, long long ldd,
                                   gps_stream_t stream);
/* The same two with a DEVICE-side row extent (rows_dev, may be NULL = n_rows): rows at or past *rows_dev are neither read
 * nor written (the masked-LM head orders the labelled rows first -- gps_lm_row_plan -- and its extent-aware GEMMs never
 * read the rest).  bf16 rows with a pitch that is a multiple of 8 elements and 16-byte aligned bases take 16-byte
 * accesses (the backward form then also writes zeros to the pad columns [vocab, 8 ceil(vocab / 8)), which ldd must
 * cover); with rows_dev, mean_out or grad_out set, any other layout returns GPS_ERR_UNSUPPORTED.
 * mean_out (optional, 2 floats) + ticket (one unsigned int, ZERO before the first launch and left zero): mean_out[0] =
 * sum(loss_rows) / count(labelled rows) -- the reference's mean -- taken by the last workgroup to arrive, in row order;
 * mean_out[1] = that count.  backward: grad_rows (per-row factors) or, when NULL, grad_out (the upstream gradient of
 * the mean, one device float) and count (one device float, e.g. mean_out + 1): every row's factor = *grad_out / *count. */
GPS_API int gps_masked_ce_forward_rows(int n_rows, int vocab, int logits_bf16, const void *logits, long long ld,
                                       const long long *labels, long long ignore_index, const int *rows_dev,
                                       float *loss_rows, float *lse, float *mean_out, unsigned int *ticket,
                                       gps_stream_t stream);
GPS_API int gps_masked_ce_backward_rows(int n_rows, int vocab, int logits_bf16, const void *logits, long long ld,
                                        const long long *labels, long long ignore_index, const int *rows_dev,
                                        const float *lse, const float *grad_rows, const float *grad_out,
                                        const float *count, void *dlogits, long long ldd, gps_stream_t stream);
/* Row plan of the masked-LM head (replaces the valid-mask / argsort / index_select chain in front of lm_cls_loss,
 * optim/loss/loss.py:56-61): perm[n_rows] = the STABLE permutation of the token rows that puts those whose label is a
 * class id in [0, vocab) (and != ignore_index) first, labels_out[i] = label of row perm[i] (ignore_index for the rest),
 * *n_valid = their count.  One workgroup; all pointers device memory. */
GPS_API int gps_lm_row_plan(int n_rows, int vocab, const long long *labels, long long ignore_index, long long *perm,
                            long long *labels_out, int *n_valid, gps_stream_t stream);

/* ---- contrastive losses (optim/loss/contra_loss.py), one launch per direction ------------------------------------------
 * fp32, contiguous rows, D a multiple of 4 (<= 8192), 16-byte aligned feature pointers.  `ticket`: one unsigned int of
 * device memory per call site, ZERO before the first launch and left zero by every launch (the last workgroup to
 * arrive takes the mean in row order: no atomics on floats, run-to-run identical).  grad_out: the upstream gradient of
 * the loss, a device float.  Gradients are those of the RAW rows (the normalisations are folded in, torch clamp_min
 * semantics for norms below eps).
 *
 * TextObjWithinBatch (:22-43, the cross-entropy branch): obj (B, O, D), text (B, D), labels (B) int64 in [0, O) (or
 * ignore_index: the scene does not count), masks (B, O) bytes (0 = padded object: logit -inf).
 * forward: cosv / prob / inv_o (B, O), inv_t (B), loss_rows (B) scratch, scal[0] = the loss (mean over the counted
 * scenes), scal[1] = their number.  backward: dobj (B, O, D) and / or dtext (B, D) (either may be NULL). */
GPS_API int gps_text_obj_ce_forward(int B, int O, int D, const float *obj, const float *text, const long long *labels,
                                    const unsigned char *masks, float eps, long long ignore_index, float *cosv,
                                    float *prob, float *inv_o, float *inv_t, float *loss_rows, float *scal,
                                    unsigned int *ticket, gps_stream_t stream);
GPS_API int gps_text_obj_ce_backward(int B, int O, int D, const float *obj, const float *text, const long long *labels,
                                     float eps, long long ignore_index, const float *cosv, const float *prob,
                                     const float *inv_o, const float *inv_t, const float *scal, const float *grad_out,
                                     float *dobj, float *dtext, gps_stream_t stream);
/* _symmetric_clip_loss (:11-17) with the F.normalize of its two inputs (:60, :82-83; normalize = 0: rows taken as they
 * are) and s = min(*scale, max_scale) (the clamp of :57, :79).  a, b (n, D).  forward: M (n, n) = a_n b_n^T, lse_row /
 * lse_col / inv_a / inv_b / loss_rows (n), loss[0].  backward: da, db (n, D) (both or neither), dscale[0] (0 when the
 * clamp is active), ds_rows (n) scratch. */
GPS_API int gps_clip_loss_forward(int n, int D, int normalize, const float *a, const float *b, const float *scale,
                                  float max_scale, float eps, float *M, float *lse_row, float *lse_col, float *inv_a,
                                  float *inv_b, float *loss_rows, float *loss, unsigned int *ticket, gps_stream_t stream);
GPS_API int gps_clip_loss_backward(int n, int D, int normalize, const float *a, const float *b, const float *scale,
                                   float max_scale, float eps, const float *M, const float *lse_row, const float *lse_col,
                                   const float *inv_a, const float *inv_b, const float *grad_out, float *da, float *db,
                                   float *dscale, float *ds_rows, unsigned int *ticket, gps_stream_t stream);

/* ---- y = x / max(||x||_2, eps) per row and its backward ------------------------------------------------------------
 * Replaces F.normalize(x, dim=-1, p=2) in the contrastive losses (optim/loss/contra_loss.py:38-39, 60, 82-83): torch runs
 * norm, clamp, expand, div forward and seven elementwise / reduction kernels backward.  x, y, dy, dx (n_rows, d) fp32
 * contiguous, inv_norm (n_rows) fp32 = 1 / max(norm, eps) saved by the forward pass; d a multiple of 4, <= 2048;
 * backward: dx = inv (dy - y (y . dy)); rows whose norm was clamped get dx = inv dy (torch's clamp_min semantics). */
GPS_API int gps_l2_normalize_forward(int n_rows, int d, const float *x, float eps, float *y, float *inv_norm,
                                     gps_stream_t stream);
GPS_API int gps_l2_normalize_backward(int n_rows, int d, const float *dy, const float *y, const float *inv_norm, float eps,
                                      float *dx, gps_stream_t stream);

/* ---- fused residual + dropout + LayerNorm (post-norm transformer layers) ---------------------------
 * y = LayerNorm(x + dropout(h)) * gamma + beta, the pattern of modules/layers/transformers.py:143-153
 * and :311-315 (`self.norm1(tgt + self.dropout1(tgt2))`), one launch instead of dropout, add,
 * layer_norm and the fp32->bf16 copy autocast inserts for the next GEMM.
 * x (n_rows, d) residual stream, fp32 or bf16 (x_bf16); h (n_rows, d) branch output, fp32 or bf16
 * (h_bf16); y has x's dtype; y_bf16 (optional, may be NULL) receives a bf16 copy of y.
 * mean/rstd (n_rows) fp32 are saved for backward.  d must be a multiple of 256, d <= 2048.
 * Dropout: counter-based like gps_attn_forward (p_drop, seed, optional device seed word). */
GPS_API int gps_add_dropout_layernorm_forward(int n_rows, int d, int x_bf16, int h_bf16, const void *x,
                                              const void *h, const float *gamma, const float *beta, float eps,
                                              float p_drop, unsigned long long seed, const void *seed_dev,
                                              void *y, void *y_bf16, float *mean, float *rstd,
                                              gps_stream_t stream);
/* rows of the partial dgamma/dbeta buffers the backward call fills (one per workgroup). */
GPS_API int gps_ln_partial_rows(int n_rows);
/* out[2][d] = column sums of part[2][parts][d] ([dgamma | dbeta] partial rows -> their totals), in a fixed
 * summation order.  scratch (optional): gps_ln_reduce_scratch_bytes(d) bytes of device memory, ZERO before its
 * first use and left zero by every call (second-level partial rows + arrival counters of the row slices; one
 * stream at a time); NULL = one workgroup per 64 columns walks all rows (slow above a few dozen rows). */
GPS_API long long gps_ln_reduce_scratch_bytes(int d);
/* The same reduction for many LayerNorms in one launch: problem i sums part_i[2][parts_i][d] into out_gamma_i (d) and
 * out_beta_i (d) (accumulate != 0: added to what is there), same summation order as gps_ln_reduce_partials.  Meant for
 * the dgamma / dbeta of every fused residual-LayerNorm backward of a step, deferred by the host (they are only needed by
 * the optimizer).  scratch: gps_ln_reduce_grouped_scratch_bytes(d) bytes, ZERO before its first use, left zero by every
 * call; one stream at a time.  The problem array is host memory, consumed before the call returns. */
typedef struct gps_ln_reduce_problem {
  const float *part;
  float *out_gamma;
  float *out_beta;
  int parts;
  int accumulate;
} gps_ln_reduce_problem;
GPS_API long long gps_ln_reduce_grouped_scratch_bytes(int d);
GPS_API int gps_ln_reduce_partials_grouped(const gps_ln_reduce_problem *problems, int n_problems, int d, void *scratch,
                                           gps_stream_t stream);
GPS_API int gps_ln_reduce_partials(int parts, int d, const float *part, float *out, void *scratch,
                                   gps_stream_t stream);
/* dy (x's dtype) [+ dy_bf16: gradient that arrived through the bf16 copy, may be NULL] -> dx (x's
 * dtype), dh (h's dtype), dgamma_part / dbeta_part (gps_ln_partial_rows(n_rows), d) fp32: the caller
 * sums them over the first axis. */
GPS_API int gps_add_dropout_layernorm_backward(int n_rows, int d, int x_bf16, int h_bf16, const void *dy,
                                               const void *dy_bf16, const void *x, const void *h,
                                               const float *gamma, const float *mean, const float *rstd,
                                               float p_drop, unsigned long long seed, const void *seed_dev,
                                               void *dx, void *dh, float *dgamma_part, float *dbeta_part,
                                               gps_stream_t stream);

/* The same two calls with a DEVICE-side row count: only the first min(n_rows, *rows_dev) rows are read / written and
 * enter the dgamma / dbeta partial sums (rows_dev NULL = all n_rows).  For row batches whose number of live rows is
 * only known on the device (text tokens compacted valid-first: modules/language/bert.py), with static launch shapes
 * so that the step stays one replayable HIP graph. */
GPS_API int gps_add_dropout_layernorm_forward_rows(int n_rows, int d, int x_bf16, int h_bf16, const void *x,
                                                   const void *h, const float *gamma, const float *beta, float eps,
                                                   float p_drop, unsigned long long seed, const void *seed_dev,
                                                   void *y, void *y_bf16, float *mean, float *rstd,
                                                   const int *rows_dev, gps_stream_t stream);
GPS_API int gps_add_dropout_layernorm_backward_rows(int n_rows, int d, int x_bf16, int h_bf16, const void *dy,
                                                    const void *dy_bf16, const void *x, const void *h,
                                                    const float *gamma, const float *mean, const float *rstd,
                                                    float p_drop, unsigned long long seed, const void *seed_dev,
                                                    void *dx, void *dh, float *dgamma_part, float *dbeta_part,
                                                    const int *rows_dev, gps_stream_t stream);
/* the same pair with an addend behind the normalisation: y = LayerNorm(x + dropout(h)) * gamma + beta + post, post (n_rows, d)
 * fp32 (x fp32) -- the per-layer `obj_embeds + loc_embeds` / `joint + extra` of the object and unified encoders
 * (modules/vision/pcd_openvocab_encoder.py:176-178, modules/grounding/unified_encoder.py:154-164) folded into the previous
 * layer's last LayerNorm, so that the layer input and its bf16 copy leave one launch.  backward: dpost (n_rows, d) fp32, if not
 * NULL, receives the gradient of y (dy + dy_bf16) = the gradient of the addend; everything else as in the _rows forms. */
GPS_API int gps_add_dropout_layernorm_forward_post(int n_rows, int d, int x_bf16, int h_bf16, const void *x, const void *h,
                                                   const float *gamma, const float *beta, float eps, float p_drop,
                                                   unsigned long long seed, const void *seed_dev, void *y, void *y_bf16,
                                                   float *mean, float *rstd, const int *rows_dev, const float *post,
                                                   gps_stream_t stream);
GPS_API int gps_add_dropout_layernorm_backward_post(int n_rows, int d, int x_bf16, int h_bf16, const void *dy,
                                                    const void *dy_bf16, const void *x, const void *h, const float *gamma,
                                                    const float *mean, const float *rstd, float p_drop,
                                                    unsigned long long seed, const void *seed_dev, void *dx, void *dh,
                                                    float *dgamma_part, float *dbeta_part, const int *rows_dev,
                                                    float *dpost, gps_stream_t stream);
/* ... with dpost_accumulate != 0 the launch ADDS its gradient of the addend to dpost (rows below the device count): the
 * SAME addend enters every layer (the reference re-adds the location / type embeddings per layer), so the layers'
 * backward launches, which run last layer first, build its gradient in one buffer -- the first of them stores, the
 * others add -- instead of one buffer each + an elementwise add per layer. */
GPS_API int gps_add_dropout_layernorm_backward_post_acc(int n_rows, int d, int x_bf16, int h_bf16, const void *dy,
                                                        const void *dy_bf16, const void *x, const void *h, const float *gamma,
                                                        const float *mean, const float *rstd, float p_drop,
                                                        unsigned long long seed, const void *seed_dev, void *dx, void *dh,
                                                        float *dgamma_part, float *dbeta_part, const int *rows_dev,
                                                        float *dpost, int dpost_accumulate, gps_stream_t stream);

/* ---- per-object input processing of the data loader ------------------------------------------------
 * Replaces ScanBase._obj_processing_post (data/datasets/base.py:697-740: optional rotation, centre/size
 * -> obj_locs, box, subsample n_points, centre on the sample mean, scale to the unit ball), the loader's
 * colour scaling colors / 127.5 - 1 (base.py:74-76) and the padding to max_obj_len + obj_masks of
 * data/datasets/dataset_wrapper.py:62-70, for all object slots of a batch in one launch.
 *   xyz (N,3) f32, rgb (N,3) u8 (rgb_is_u8 != 0) or f32 in 0..255: the RAW scene points, every object's
 *   points contiguous; obj_offsets (n_obj+1) int64 CSR into them.  rgb == NULL selects the packed layout:
 *   xyz then points at N 16-byte records {f32 x, y, z; u8 r, g, b, pad} (16-byte aligned) -- one vector
 *   load per point, colours gathered together with the coordinates.
 *   row_obj (n_rows) int32: object id of output row r, or -1 = padding slot (features 1.0, locs 0, mask 0).
 *   sample_idx (n_rows, n_points) int32 object-local indices (the loader's np.random.choice draw), or NULL:
 *   drawn on the device from `seed` (with replacement iff the object has < n_points points, else a keyed
 *   permutation prefix -- distinct indices).  rot (n_rot,3,3) f32 + row_rot (n_rows) int32 (-1 = none),
 *   both NULL for no rotation.  Outputs: obj_fts (n_rows, n_points, 6) f32, obj_locs (n_rows, 6) f32,
 *   obj_boxes (n_rows, 6) f32 or NULL, obj_masks (n_rows) u8 or NULL.  n_points <= 2048.
 * Arithmetic in float64, rounded to f32 at the end, like the reference for uint8 colours. */
GPS_API int gps_obj_processing_post(int n_rows, int n_points, const float *xyz, const void *rgb, int rgb_is_u8,
                                    const int64_t *obj_offsets, const int32_t *row_obj,
                                    const int32_t *sample_idx, uint64_t seed, const float *rot,
                                    const int32_t *row_rot, float *obj_fts, float *obj_locs, float *obj_boxes,
                                    uint8_t *obj_masks, gps_stream_t stream);

/* ---- bias gradients: column sums of a bf16 matrix ---------------------------------------------------
 * Replaces the `dY.sum(0)` autograd derives for the bias of every nn.Linear in the transformer stacks
 * (modules/layers/transformers.py:115-154, 285-316; HF BertLayer behind modules/language/bert.py:21-26).
 * x (rows, ld >= cols) bf16, cols and ld multiples of 8, x 16-byte aligned (else GPS_ERR_UNSUPPORTED);
 * partials: scratch of gps_colsum_parts(rows, cols) * cols floats; out (cols) fp32 = column sums with
 * fp32 accumulation in a fixed (deterministic) order. */
GPS_API int gps_colsum_parts(int rows, int cols);
GPS_API int gps_colsum_bf16(int rows, int cols, const void *x, long long ld, float *partials, float *out,
                            gps_stream_t stream);

/* ---- dense gradient of an embedding table -----------------------------------------------------------
 * Replaces the backward of the word-embedding lookup of the language encoder (HF BertEmbeddings behind
 * modules/language/bert.py:21-26; torch: sort + segment reduction + scatter).  ids (n) int64, dy (n, ld >= d)
 * fp32, out (num_rows, d) fp32 = for every table row the sum of dy[t] over the tokens t with ids[t] == row
 * (zero for rows never referenced and for row padding_idx; pass -1 for "no padding row"; ids outside
 * [0, num_rows) are ignored).  Deterministic: no floating-point atomics, a fixed order of additions.
 * scratch: gps_embedding_grad_scratch_ints(n, num_rows, d) int32, 16-byte aligned (first occurrence + count per row, the
 * list of tokens whose id occurs more than once, token lists and partial rows of the ids with >= 64 tokens).  Ids with
 * fewer than 64 tokens are added in ascending token order; heavier ones 64 rows per partial sum, the partial sums in
 * order -- a fixed association either way.  d and ld multiples of 4, d <= 2048, dy / out 16-byte aligned. */
GPS_API long long gps_embedding_grad_scratch_ints(int n, int num_rows, int d);
GPS_API int gps_embedding_grad(int n, int d, int num_rows, const int64_t *ids, const float *dy, long long ld,
                               long long padding_idx, int32_t *scratch, float *out, gps_stream_t stream);

/* ---- embedding block of the BERT text encoder: lookups + LayerNorm + dropout in one launch per direction ----
 * Replaces HF BertEmbeddings.forward as the language encoder calls it (modules/language/bert.py:21-26: input_ids only,
 * i.e. token type 0 everywhere and caller-supplied positions):
 *     y[r] = dropout(LayerNorm(word[ids[r]] + type_row + pos_table[pos[r]]) * gamma + beta)        r < n_rows
 * ids / pos (n_rows) int64 (ids in [0, vocabulary), pos in [0, positions): NOT checked); tables fp32 with row pitch d;
 * y fp32 and, if y_bf16 is not null, the same values as bf16; mean / rstd (n_rows) fp32 are the only saved state: the
 * backward pass gathers the three rows again.  rows_dev (optional, device int32): only the first min(n_rows, *rows_dev)
 * rows are computed, the others are not written.  Dropout: the counter-based stream of
 * gps_add_dropout_layernorm_forward (element index r * d + c, seed + *seed_dev).
 * backward: dz (n_rows, d) fp32 = gradient of the pre-LayerNorm sum for dy (+ dy_bf16 if not null) -- the operand of
 * gps_embedding_grad for the word table (ids) and the position table (pos); rows past the device row count get zeros;
 * the type-row gradient is the column sum of the position-table gradient.  dgamma_part / dbeta_part:
 * (gps_bert_embed_partial_rows(n_rows), d) fp32 per-workgroup partial sums for gps_ln_reduce_partials.
 * d a multiple of 256, <= 1024; all fp32 / bf16 pointers 16-byte aligned. */
GPS_API int gps_bert_embed_partial_rows(int n_rows);
/* position-table gradient when the rows are n_seq whole sequences laid end to end (sequence s = rows cu_rows[s] ..
 * cu_rows[s + 1] - 1, cu_rows (n_seq + 1) device int32) and a row's position is its offset inside its sequence (HF's
 * default position_ids): out (n_pos, d) fp32, out[p] = sum over the sequences longer than p of dz[cu_rows[s] + p], in
 * sequence order (deterministic); positions no sequence reaches get zeros. */
GPS_API int gps_bert_position_grad(int n_seq, int n_pos, int d, const int *cu_rows, const float *dz, float *out,
                                   gps_stream_t stream);
GPS_API int gps_bert_embed_forward(int n_rows, int d, const long long *ids, const long long *pos, const float *word,
                                   const float *pos_table, const float *type_row, const float *gamma, const float *beta,
                                   float eps, float p_drop, unsigned long long seed, const void *seed_dev, float *y,
                                   void *y_bf16, float *mean, float *rstd, const int *rows_dev, const int *poison_dev,
                                   gps_stream_t stream);
/* poison_dev (optional device word, e.g. gps_varlen_plan's violation word): non-zero -> every output row is NaN. */
GPS_API int gps_bert_embed_backward(int n_rows, int d, const float *dy, const void *dy_bf16, const long long *ids,
                                    const long long *pos, const float *word, const float *pos_table, const float *type_row,
                                    const float *gamma, const float *mean, const float *rstd, float p_drop,
                                    unsigned long long seed, const void *seed_dev, float *dz, float *dgamma_part,
                                    float *dbeta_part, const int *rows_dev, gps_stream_t stream);

/* index plan of the variable-length text path (modules/language/bert.py::_fast_forward_varlen; the reference runs the
 * padded batch, modules/language/bert.py:26-30 -- this is what lets the encoder stack skip padded rows): from the
 * attention masks of n_texts texts (text i = n_seq x len ids + mask, masks NON-EMPTY PREFIXES of their rows -- the
 * caller's promise), S = sum n_seq sequences and T = sum n_seq * len token positions, two launches (lengths; rows) write
 *   i32_out [4 S + 5]: lens[S] | cu_rows[S + 1] (row offsets of the compacted sequences) | order[S] (sequence indices,
 *            longest first, ties by index) | q_limit[S] (lens for the first n_seq_full sequences, 1 behind them) |
 *            n_valid | live rows of the first n_seq_full sequences | that + (S - n_seq_full) | violation (1 when some
 *            mask is NOT a non-empty prefix of its row -- an empty row, a hole, left padding: the plan then differs from
 *            the torch formulation; consumers poison their output with it, see gps_bert_embed_forward's poison_dev)
 *   i64_out [3 T + (S - n_seq_full) + T_full]: ids of the compacted rows (valid tokens in flat order, then the padded
 *            positions in flat order) | their positions inside their row | inv (compact row of every flat position) |
 *            sel = cu_rows[n_seq_full .. S) followed by 0 .. T_full - 1 (T_full = token positions of the first
 *            n_seq_full sequences; written only when 0 < n_seq_full < S, which must be a text boundary)
 *   valid_out [T] bytes: 1 where the mask is set.
 * Equal, element for element, to the torch formulation (stable argsort of the valid flag, cumsum, ...) for prefix masks:
 * tests/test_gpu_bert_varlen.py.  S <= 8192, T < 2^31. */
#define GPS_VARLEN_MAX_TEXTS 8
typedef struct gps_varlen_text {
  const long long *ids;   /* (n_seq, len) int64, contiguous */
  const void *mask;       /* (n_seq, len), contiguous; element != 0 = valid token */
  int mask_elem_bytes;    /* 1, 2, 4 or 8 */
  int mask_is_float;      /* floating-point mask: -0.0 counts as 0 */
  int n_seq, len;
} gps_varlen_text;
GPS_API int gps_varlen_plan(const gps_varlen_text *texts, int n_texts, int n_seq_full, int *i32_out, long long *i64_out,
                            unsigned char *valid_out, gps_stream_t stream);
/* Row-compaction plan of n_seq sequences of seq_len rows each with an ARBITRARY validity mask (valid: (n_seq * seq_len)
 * bytes, non-zero = valid) -- the joint text + object sequences of the unified encoder (reference
 * modules/grounding/unified_encoder.py:147-177), whose padded rows are masked as attention keys and ignored as outputs:
 *   perm (n) int64: compact row r <- flat row perm[r] (the valid rows in their order, then the invalid ones);
 *   inv (n) int64: flat row -> compact row;  cu (n_seq + 1) int32: first compact row of every sequence, cu[n_seq] = n_live;
 *   n_live (1) int32.  n_seq * seq_len <= 2^22.  One launch. */
GPS_API int gps_rows_plan(int n_seq, int seq_len, const unsigned char *valid, long long *perm, long long *inv, int *cu, int *n_live,
                          gps_stream_t stream);

/* Pack / unpack of the JOINT rows without a concatenated tensor: the padded side is the text rows a (n_seq, len_a, d) and the
 * object rows b (n_seq, len_b, d), fp32, which the reference joins along the sequence axis (modules/grounding/unified_encoder.py:
 * 147-177); flat row e = (sequence, position in [0, len_a + len_b)); perm / inv / valid / n_live as gps_rows_plan writes them.
 *   pack2:   out[r] = flat[perm[r]] for r < *n_live, zeros past it; out16 (optional) = the same rows as bf16;
 *   unpack2: flat[e] = valid[e] ? packed[inv[e]] : 0, written into out_a / out_b (both contiguous).
 * Each is the gradient of the other.  d % 4 == 0, n_seq * (len_a + len_b) <= 2^22. */
GPS_API int gps_rows_pack2(int n_seq, int len_a, int len_b, int d, const float *a, const float *b, const long long *perm,
                           const int *n_live, float *out, unsigned short *out16, gps_stream_t stream);
GPS_API int gps_rows_unpack2(int n_seq, int len_a, int len_b, int d, const float *packed, const long long *inv,
                             const unsigned char *valid, float *out_a, float *out_b, gps_stream_t stream);

/* First-layer input of the unified encoder in one launch (reference modules/grounding/unified_encoder.py:147-164: the text and
 * object embeddings concatenated, the token-type / location embeddings added): with joint = (a | b), extra = (ea | eb) in the
 * flat order of gps_rows_pack2,  x[r] = joint[perm[r]] + extra[perm[r]],  e[r] = extra[perm[r]],  x16 = bf16(x)  for
 * r < *n_live, zeros past it.  backward: g = dx + dx16 (each optional: NULL = zero; dx16 bf16),  d(a | b)[f] = valid[f] ?
 * g[inv[f]] : 0,  d(ea | eb)[f] = valid[f] ? g[inv[f]] + de[inv[f]] : 0  (de optional), all four outputs contiguous fp32. */
GPS_API int gps_joint_embed_forward(int n_seq, int len_a, int len_b, int d, const float *a, const float *b, const float *ea,
                                    const float *eb, const long long *perm, const int *n_live, float *x, float *e,
                                    unsigned short *x16, gps_stream_t stream);
GPS_API int gps_joint_embed_backward(int n_seq, int len_a, int len_b, int d, const float *dx, const unsigned short *dx16,
                                     const float *de, const long long *inv, const unsigned char *valid, float *da, float *db,
                                     float *dea, float *deb, gps_stream_t stream);

/* Row mover for rows of ANY element type (row_bytes a multiple of 16, both arrays 16-byte aligned): launch row r < n moves
 * source row (src_idx ? src_idx[r] : r) to destination row (dst_idx ? dst_idx[r] : r) when r < *n_live (n_live optional);
 * rows whose source index is outside [0, n_src_rows) arrive as zeros, rows whose destination index is outside are dropped.
 * With zero_dead != 0 the destination rows of the launch rows at or past *n_live are zeroed (rows r in the gather form, rows
 * dst_idx[r] in the scatter form: with a bijective dst_idx every destination row is then written).  Scatter form (dst_idx)
 * otherwise: only addressed rows are written -- the caller zero-fills dst; the destination indices must be distinct.
 * Replaces index_select + the dead-row mask and, in backward, the atomic index_add_ of the [CLS]-tail selection of the
 * variable-length text path (no counterpart in the reference, which runs the padded batch: modules/language/bert.py:26-30). */
GPS_API int gps_rows_move(int n, long long n_src_rows, long long n_dst_rows, int row_bytes, const void *src, const long long *src_idx,
                          void *dst, const long long *dst_idx, const int *n_live, int zero_dead, gps_stream_t stream);

/* ---- box-location embedding  y = LayerNorm(x W^T + b)  (tiny reduction length) ---------------------------------
 * Replaces `loc_layers = nn.Sequential(nn.Linear(dim_loc, hidden), nn.LayerNorm(hidden))` of the object encoder and the
 * unified encoder (modules/vision/pcd_openvocab_encoder.py:64-66, :177; modules/grounding/unified_encoder.py:28-30, :158).
 * x (n_rows, k_in) fp32 contiguous (k_in in {3, 6, 8}), w (d, k_in) fp32 contiguous, bias (d) or NULL, gamma / beta (d),
 * y (n_rows, d) fp32, mean / rstd (n_rows) saved for the backward pass; d == 768.
 * backward (no input gradient: the boxes are data): sums (k_in + 3, d) fp32 = [dW^T rows k = 0 .. k_in - 1 | db | dgamma |
 * dbeta]; partials: (gps_loc_embed_partial_rows(n_rows), k_in + 3, d) fp32 scratch.  Deterministic. */
GPS_API int gps_loc_embed_partial_rows(int n_rows);
GPS_API int gps_loc_embed_forward(int n_rows, int k_in, int d, const float *x, const float *w, const float *bias,
                                  const float *gamma, const float *beta, float eps, float *y, float *mean, float *rstd,
                                  gps_stream_t stream);
GPS_API int gps_loc_embed_backward(int n_rows, int k_in, int d, const float *dy, const float *x, const float *w,
                                   const float *bias, const float *gamma, const float *mean, const float *rstd,
                                   float *partials, float *sums, gps_stream_t stream);

/* ---- bf16 MFMA GEMMs of the transformer projections / FFNs -------------------------------------------
 * Replaces the nn.Linear contractions of the GPS transformer layers -- w_qs / w_ks / w_vs / fc / lang_cond_fc
 * (modules/layers/transformers.py:173-186, 193-197), nn.MultiheadAttention's in/out projections (:120-121, 141)
 * and the FFNs (:123-125, 148-152, 301-316) -- forward, input gradient and weight gradient, with the bias,
 * activation, dropout and activation-derivative steps the reference runs as separate elementwise kernels
 * applied in the epilogue.  bf16 operands, fp32 accumulation (v_mfma_f32_16x16x32_bf16).
 *
 *   form GPS_GEMM_NT  C (M,N) = A (M,K) . B (N,K)^T      A, B K-major (lda, ldb = row pitches in elements)
 *   form GPS_GEMM_NN  C (M,N) = A (M,K) . B (K,N)        B reduction-major
 *   form GPS_GEMM_TN  C (M,N) = A (K,M)^T . B (K,N)      both reduction-major; C fp32; K may be split
 *
 * epilogue (C is bf16 unless stated):
 *   GPS_GEMM_EPI_BIAS       C = acc + bias                                   (bias (N) fp32, may be NULL)
 *   GPS_GEMM_EPI_BIAS_GELU  pre = bf16(acc + bias) -> aux_out (if not NULL); C = dropout(gelu(pre))   (erf GELU)
 *   GPS_GEMM_EPI_BIAS_RELU  C = dropout(relu(acc + bias))
 *   GPS_GEMM_EPI_DGELU      C = acc * gelu'(aux) * dropout-mask              (aux (M,N) bf16 = saved pre)
 *   GPS_GEMM_EPI_DRELU      C = acc * (aux != 0 ? 1/(1-p) : 0)               (aux (M,N) bf16 = saved dropout(relu()))
 *   GPS_GEMM_EPI_BIAS_GELU_FACTOR  C as GPS_GEMM_EPI_BIAS_GELU; aux_out = bf16(gelu'(pre) * dropout-mask / (1 - p)): the
 *                           factor the backward pass multiplies by (it costs the forward epilogue four more vector
 *                           instructions per element -- the erf terms are shared -- and saves the backward ~30)
 *   GPS_GEMM_EPI_MUL_AUX    C = acc * aux                                    (aux (M,N) bf16 = that saved factor)
 *   GPS_GEMM_EPI_RELU_SPLIT v = relu(acc + bias) written as a bf16 pair hi = rne(v), lo = rne(v - hi):
 *                           C[m][n] = hi, C[m][N + n] = lo, C[m][2N + n] = hi (ldc >= 3N): the [hi | lo | hi] operand
 *                           that, against weights laid out [W_hi | W_hi | W_lo] along K, gives the next layer's
 *                           x W^T to ~2^-16 relative (fp32-accurate MLP chains on the bf16 MFMA path).  Form NT.
 *   GPS_GEMM_EPI_RELU_MAX16 C fp32 (M / 16, N): max over each block of 16 consecutive rows of relu(acc + bias)
 *                           (shared MLP + max-pool over a 16-point group).  Form NT, M % 16 == 0.
 *   GPS_GEMM_EPI_F32        C fp32 = acc; form TN only.  With colsum != NULL also colsum (M) fp32 = column sums of
 *                           A over K (the bias gradient when A = dY).
 * dropout: keep an element iff rng(seed + *seed_dev, m * N + n) >= p * 2^32, scale kept ones by 1/(1-p); the mask
 *   of GPS_GEMM_EPI_DGELU is recomputed from the same (seed, index), nothing is stored.  p_drop = 0: none.
 * splits (TN only): K is cut into `splits` ranges whose fp32 partial tiles go to `workspace`
 *   (gps_gemm_workspace_floats() floats) and are summed in split order by a second launch (deterministic);
 *   gps_gemm_pick_splits() gives the default.  variant: tile configuration 0..13 (8, 9, 10: persistent workgroups;
 *   11: four-wave 256 x 256 with register-staged operands; 12: eight-wave two-group 256 x 256, the default for long
 *   reductions and wide weight gradients; 13: its stream-K form, see gps_gemm_sk_workspace_bytes), or -1 = chosen from
 *   the shape.
 * Requirements (else GPS_ERR_UNSUPPORTED): lda, ldb multiples of 8, K too unless form TN; N, ldc, ldaux multiples of 4; for
 *   reduction-major operands their column count (N, and M in form TN) a multiple of 8; A, B, C, bias 16-byte aligned. */
#define GPS_GEMM_NT 0
#define GPS_GEMM_NN 1
#define GPS_GEMM_TN 2
#define GPS_GEMM_EPI_BIAS_GELU_FACTOR 8
#define GPS_GEMM_EPI_MUL_AUX 9
#define GPS_GEMM_EPI_BIAS 0
#define GPS_GEMM_EPI_BIAS_GELU 1
#define GPS_GEMM_EPI_BIAS_RELU 2
#define GPS_GEMM_EPI_DGELU 3
#define GPS_GEMM_EPI_DRELU 4
#define GPS_GEMM_EPI_F32 5
#define GPS_GEMM_EPI_RELU_SPLIT 6
#define GPS_GEMM_EPI_RELU_MAX16 7
typedef struct gps_gemm_args {
  int form, epilogue, M, N, K, splits, variant, reserved; /* reserved: 0 */
  const void *A;
  long long lda;
  const void *B;
  long long ldb;
  void *C;
  long long ldc;
  const float *bias;
  const void *aux;
  long long ldaux;
  void *aux_out;
  long long ldaux_out;
  float *workspace;
  float *colsum;
  const void *seed_dev; /* optional device uint64 added to `seed` (HIP-graph replays advance it on the device) */
  unsigned long long seed;
  float p_drop;
  int reserved2; /* forms NT / NN: index of this call's first row in the (larger) product whose dropout stream it continues
                  * (the mask of row m, column n is drawn at index (reserved2 + m) * N + n); 0 for a whole product */
  /* optional device int32: the number of LEADING token rows that carry work.  Forms NT / NN: output rows at or
   * past it are not computed (their tiles exit at once; rows of the last started tile may be written).  Form TN:
   * the reduction stops at the end of the 64-row stage that contains row *extent_dev - 1 (operand rows between
   * the extent and that stage end must hold zeros).  Lets a HIP-graph-captured step skip rows whose count is only
   * known on the device (the unlabelled ~85 % of the masked-LM head's tokens).  NULL = all rows. */
  const int *extent_dev;
} gps_gemm_args;
GPS_API int gps_gemm_pick_splits(int form, int M, int N, int K);
/* the tile configuration gps_gemm_bf16 takes for variant = -1 (splits < 1: the default split count): 12 = the two-group
 * 256 x 256 kernel (gemm8p_kernel), 7 / 2 = 128 x 128 tiles, 6 = 128 x 64 tiles (gemm_kernel instantiations) -- what a
 * profile reader needs to match a launch with its rocprofv3 row */
GPS_API int gps_gemm_pick_variant(int form, int M, int N, int K, int splits);
/* the same for a given epilogue (the split-bf16 epilogues 6 / 7 keep the 128 x 128 tiles at short reductions) */
GPS_API int gps_gemm_pick_variant_ex(int form, int M, int N, int K, int splits, int epilogue);
GPS_API long long gps_gemm_workspace_floats(int form, int M, int N, int splits);
/* variant 13 (forms NT / NN, bf16 epilogues): the stream-K form of variant 12 -- one resident workgroup per CU, each
 * taking an equal share of the K tiles of ALL 256 x 256 output tiles, so that a launch is never a whole number of
 * "rounds" of tiles; a tile cut by a share boundary is finished by the workgroup that holds its k = 0 end, which adds
 * the other shares' fp32 accumulators (in a fixed order: deterministic for a given shape, device-side extent and CU
 * count).  Needs `workspace` = gps_gemm_sk_workspace_bytes() bytes, 16-byte aligned, whose first 4 KiB are ZERO before
 * the first launch (the kernel leaves them zero); without it variant 13 runs as variant 12.  One launch at a time per
 * workspace (launches on one stream are fine).  Word 256 of the workspace is set to 1 if a bounded wait ever expired. */
GPS_API long long gps_gemm_sk_workspace_bytes(void);
/* Grouped weight gradients: for every problem p,  C_p (M,N) fp32 [+]= A_p (K,M)^T . B_p (K,N)  and, when colsum_p is not
 * NULL, colsum_p (M) [+]= column sums of A_p over K -- the weight and bias gradient of one nn.Linear (A = dY, B = X, bf16,
 * both reduction-major as in form GPS_GEMM_TN) -- for ALL problems in one persistent launch, each 256 x 256 output tile
 * walked over its whole reduction by one workgroup: no split over K, no partial tiles, no reduce launch.  Meant for the
 * weight gradients of a whole backward pass, deferred by the host and issued together (their reductions over 5 000 -
 * 22 000 token rows cannot fill the chip one GEMM at a time without splitting).  accumulate != 0: read-modify-write
 * (the destination already holds a gradient); every output element has one writer, results are deterministic.
 * extent_dev: as in gps_gemm_args (the reduction stops after the first *extent_dev rows).  Requirements as form TN:
 * M, N, lda, ldb multiples of 8, ldc of 4, 16-byte aligned A, B, C, K * ld* * 2 < 2^31.  The problem array is host
 * memory and is consumed before the call returns (capturable: the table travels as kernel arguments).
 * Threading / streams: the launcher keeps process-wide state (the tile-queue counters and a small ring of device
 * tables): calls must be issued from ONE host thread at a time and on ONE stream (or on streams ordered after one
 * another) of ONE device; two launches in flight on unordered streams would share queue counters.  A captured launch
 * keeps its ring slot for the life of the graph. */
typedef struct gps_wgrad_problem {
  int M, N, K, accumulate;
  const void *A;
  long long lda;
  const void *B;
  long long ldb;
  float *C;
  long long ldc;
  float *colsum;
  const int *extent_dev;
} gps_wgrad_problem;
GPS_API int gps_gemm_wgrad_grouped(const gps_wgrad_problem *problems, int n_problems, gps_stream_t stream);
/* tile schedule of gps_gemm_wgrad_grouped: 1 (default) = one queue per XCD -- the tiles of a problem run side by side on
 * one XCD and share their operand panels in its L2; workgroups take tiles of other XCDs' queues only when their own is
 * empty -- 0 = one global longest-first queue; < 0 = query.  Returns the previous setting.  Results do not depend on it
 * (every tile is computed by exactly one workgroup, the same way). */
GPS_API int gps_gemm_wgrad_grouped_set_xcd_queues(int on);
/* First operand of a split-bf16 MLP chain over a group-all point level (reference: GroupAll in
 * modules/third_party/pointnet2/pointnet2_utils.py -- cat of the grouped xyz and features): row (b, j) =
 * [xyz (b, n, 3)[b][j] | feats (b, c, n)[b][:, j]] as bf16 [hi | lo | hi], each third k_pad >= 3 + c columns wide
 * (k_pad % 8 == 0, padding columns zero).  out: (b * n, 3 * k_pad) bf16.  n * (4 + c) * 4 bytes of LDS <= 64 KiB. */
GPS_API int gps_split3_points(int b, int n, int c, const float *xyz, const float *feats, int k_pad, void *out,
                              gps_stream_t stream);
GPS_API int gps_gemm_bf16(const gps_gemm_args *args, gps_stream_t stream);
/* n (<= 4) INDEPENDENT products of ONE form (NT or NN) and ONE epilogue -- each with its own operands, shape, bias, saved
 * activations, dropout stream and device-side row extent -- as one launch of the 256 x 256 kernel (variant 12) over the union
 * of their output tiles: e.g. the same Linear of the text stack and of the object stack of the GPS model, which are
 * independent until the joint layers (reference model/openvocab.py:41-63) and alone fill 23 - 59 % of the chip.  Results are
 * those of n gps_gemm_bf16 calls with variant 12.  `variant`, `splits`, `workspace` are ignored.  GPS_ERR_UNSUPPORTED (nothing
 * launched: the caller issues the products one by one) for fp32 / split-bf16 epilogues, K < 64, or K % 64 != 0 with an
 * epilogue other than GPS_GEMM_EPI_BIAS; GPS_ERR_INVALID_ARGUMENT when forms or epilogues differ.  n == 1 is gps_gemm_bf16. */
GPS_API int gps_gemm_bf16_grouped(const gps_gemm_args *args, int n, gps_stream_t stream);

/* ---- optimizer step: gradient clipping + AdamW over all parameter tensors ------------------------------
 * Replaces `accelerator.clip_grad_norm_` + `optimizer.step()` of the reference's training step
 * (trainer/default_trainer.py:18-24; torch.nn.utils.clip_grad_norm_ with the L2 norm, torch.optim.AdamW built by
 * optim/optimizer/optim.py:9-14) with three launches over device-resident tables:
 *   tensors  one record per parameter with a gradient: fp32 master / gradient / both moments (numel each), and
 *            optionally a bf16 shadow and / or an fp32 mirror of the parameter that are rewritten with the new
 *            values (what gps_gemm_bf16 and packed bias vectors read), and the index of its hyper-parameter group;
 *   groups   lr (read from *lr_dev when lr_dev is not NULL: HIP-graph replays see the scheduler's in-place
 *            updates), betas, eps, decoupled weight decay;
 *   chunks   (tensor index, chunk index) pairs, gps_adamw_chunk_elems() elements per chunk, one workgroup each.
 * max_grad_norm > 0: every gradient is scaled by min(1, max_grad_norm / (||g||_2 + 1e-6)) inside the update
 * (gradients themselves are not modified); <= 0: no clipping.  scalars (2 floats): [0] clip coefficient,
 * [1] total gradient norm of this step.  steps: persistent per-parameter step counts (float), indexed by each
 * record's step_slot; the call adds 1 to the slot of every tensor in the table and uses the result for the bias
 * corrections 1 - beta^step (torch counts steps per parameter); zero them before the first step.
 * partial: scratch of n_chunks floats.
 * Update rule per element, exactly torch.optim.AdamW (amsgrad = False, maximize = False):
 *   p *= 1 - lr * wd;  m = b1 m + (1 - b1) g;  v = b2 v + (1 - b2) g^2;
 *   p -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps). */
typedef struct gps_adamw_tensor {
  void *param;
  const void *grad;
  void *exp_avg;
  void *exp_avg_sq;
  void *shadow_bf16; /* may be NULL */
  void *mirror_f32;  /* may be NULL */
  long long numel;
  int group;
  int step_slot;
} gps_adamw_tensor;
typedef struct gps_adamw_group {
  const void *lr_dev; /* device float or NULL */
  float lr, beta1, beta2, eps, weight_decay;
  int reserved;
} gps_adamw_group;
GPS_API int gps_adamw_chunk_elems(void);
GPS_API int gps_adamw_step(int n_tensors, int n_chunks, const gps_adamw_tensor *tensors, const gps_adamw_group *groups,
                           const int32_t *chunks, float max_grad_norm, float *partial, float *scalars, float *steps,
                           gps_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* GPS_HIP_H_ */
