// gemm_probe.hip -- standalone instrument for the bf16 MFMA GEMMs of sceneverse_amd/csrc/gps_gemm.hip (no torch, no python:
// a gpurun call spends its minutes on kernels, not on imports).  It compiles the library source itself with
// GPS_GEMM_TRACE, which adds per-workgroup shader-clock stamps (entry, first stage landed, every K tile, main loop done,
// epilogue issued, stores acknowledged) that the product build does not carry.
//
//   gemm_probe bench  [--warm] [--rounds R] [--inner I] [--set in_step|small]     per-shape times of the listed variants
//   gemm_probe trace  FORM EPI M N K VARIANT                                      one launch, timeline summary of its workgroups
//   gemm_probe loop   FORM EPI M N K VARIANT ITERS                                ITERS launches (for rocprofv3 --pmc)
//
// Operands are uniform random bf16 in [-1, 1) (guide rule 25: zero-filled operands clock higher).  "cold" timing (the
// default) rotates over enough operand sets that no launch finds its inputs in the 256 MiB infinity cache -- what a
// launch inside the training step sees; --warm replays one set.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -o tools/probes/gemm_probe tools/probes/gemm_probe.hip
#define GPS_GEMM_TRACE 1
#include "../../sceneverse_amd/csrc/gps_gemm.hip"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace gps { const int *object_extent() { return nullptr; } }

#define CK(x)                                                                               \
  do {                                                                                      \
    hipError_t e_ = (x);                                                                    \
    if (e_ != hipSuccess) {                                                                 \
      fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_));   \
      exit(2);                                                                              \
    }                                                                                       \
  } while (0)

__global__ void fill_bf16(uint16_t *p, size_t n, unsigned int seed, float scale) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned int x = (unsigned int)i * 2654435761u ^ seed;
    x ^= x >> 16; x *= 0x21F0AAADu; x ^= x >> 15; x *= 0x735A2D97u; x ^= x >> 15;
    const float v = ((float)(x >> 8) * (1.f / 8388608.f) - 1.f) * scale;
    unsigned int u = __float_as_uint(v);
    u += 0x7FFFu + ((u >> 16) & 1u);
    p[i] = (uint16_t)(u >> 16);
  }
}
__global__ void fill_f32(float *p, size_t n, unsigned int seed) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned int x = (unsigned int)i * 2654435761u ^ seed;
    x ^= x >> 16; x *= 0x21F0AAADu; x ^= x >> 15;
    p[i] = (float)(x >> 8) * (1.f / 8388608.f) - 1.f;
  }
}
// number of bf16 words that differ / the largest difference between two outputs
__global__ void diff_bf16(const uint16_t *a, const uint16_t *b, size_t n, unsigned long long *cnt, float *maxd) {
  float m = 0.f;
  unsigned long long c = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    if (a[i] != b[i]) {
      ++c;
      const float d = fabsf(__uint_as_float((unsigned)a[i] << 16) - __uint_as_float((unsigned)b[i] << 16));
      m = fmaxf(m, d);
    }
  }
  if (c) {
    atomicAdd(cnt, c);
    atomicMax(reinterpret_cast<int *>(maxd), __float_as_int(m));
  }
}

// independent reference for EPI_BIAS: one thread per output element, fp32 sums in k order
__global__ void ref_gemm_bias(int form, int M, int N, int K, const uint16_t *A, const uint16_t *B, const float *bias, uint16_t *C) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (size_t)M * N) return;
  const int m = (int)(e / N), n = (int)(e % N);
  float s = 0.f;
  for (int k = 0; k < K; ++k) {
    const float a = __uint_as_float((unsigned)A[(size_t)m * K + k] << 16);
    const float b = __uint_as_float((unsigned)(form == 0 ? B[(size_t)n * K + k] : B[(size_t)k * N + n]) << 16);
    s = fmaf(a, b, s);
  }
  s += bias[n];
  unsigned int u = __float_as_uint(s);
  u += 0x7FFFu + ((u >> 16) & 1u);
  C[e] = (uint16_t)(u >> 16);
}
// elements whose difference exceeds `tol` (absolute)
__global__ void count_far(const uint16_t *a, const uint16_t *b, size_t n, float tol, unsigned long long *cnt) {
  unsigned long long c = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    if (!(fabsf(__uint_as_float((unsigned)a[i] << 16) - __uint_as_float((unsigned)b[i] << 16)) <= tol)) ++c;
  if (c) atomicAdd(cnt, c);
}

struct Shape { int form, epi, M, N, K; const char *what; };
// one GPS pre-train step at B = 64 (live text rows 12 608): the NT / NN launches by time (profiles/r5/bench_detail_final.json)
static const Shape kInStep[] = {
    {1, 9, 12608, 3072, 768, "text ffn1 dgrad"},   {1, 0, 12608, 768, 2304, "text qkv dgrad"},
    {0, 8, 12608, 3072, 768, "text ffn1 fwd"},     {1, 0, 12608, 768, 3072, "text ffn2 dgrad"},
    {1, 4, 8320, 2048, 768, "joint ffn1 dgrad"},   {0, 0, 12608, 2304, 768, "text qkv fwd"},
    {0, 0, 12608, 768, 3072, "text ffn2 fwd"},     {1, 0, 8320, 768, 2304, "joint qkv dgrad"},
    {0, 2, 8320, 2048, 768, "joint ffn1 fwd"},     {1, 0, 5120, 768, 2376, "obj qkv dgrad"},
    {1, 0, 8320, 768, 2048, "joint ffn2 dgrad"},   {0, 0, 8320, 2304, 768, "joint qkv fwd"},
    {1, 9, 5120, 2048, 768, "obj ffn1 dgrad"},     {0, 8, 5120, 2048, 768, "obj ffn1 fwd"},
    {1, 0, 5120, 768, 2048, "obj ffn2 dgrad"},     {0, 0, 8320, 768, 2048, "joint ffn2 fwd"},
    {0, 0, 5120, 2376, 768, "obj qkv fwd"},        {0, 0, 5120, 768, 2048, "obj ffn2 fwd"},
    {1, 0, 12608, 768, 768, "text out dgrad"},     {1, 0, 8320, 768, 768, "joint out dgrad"},
    {0, 0, 12608, 768, 768, "text out fwd"},       {0, 0, 8320, 768, 768, "joint out fwd"},
    {1, 0, 5120, 768, 768, "obj out dgrad"},       {0, 0, 5120, 768, 768, "obj out fwd"},
};
static const Shape kSmall[] = {
    {0, 0, 12608, 2304, 768, "text qkv fwd"}, {1, 0, 12608, 768, 2304, "text qkv dgrad"}, {0, 8, 5120, 2048, 768, "obj ffn1 fwd"},
    {1, 0, 5120, 768, 2376, "obj qkv dgrad"}, {0, 0, 8320, 768, 768, "joint out fwd"},
};

struct Buffers {
  int sets = 1;
  std::vector<uint16_t *> A, C, aux, aux_out;
  uint16_t *B = nullptr;
  float *bias = nullptr;
  size_t a_elems = 0, b_elems = 0, c_elems = 0;
};

static bool has_aux(int epi) { return epi == 3 || epi == 4 || epi == 9; }
static bool has_pre(int epi) { return epi == 1 || epi == 8; }

static Buffers make(const Shape &s, int sets) {
  Buffers b;
  b.sets = sets;
  b.a_elems = (size_t)s.M * s.K;
  b.b_elems = (size_t)s.N * s.K;
  b.c_elems = (size_t)s.M * s.N;
  CK(hipMalloc(&b.B, b.b_elems * 2));
  CK(hipMalloc(&b.bias, (size_t)s.N * 4));
  fill_bf16<<<1024, 256>>>(b.B, b.b_elems, 0x1234u, 0.05f);
  fill_f32<<<64, 256>>>(b.bias, (size_t)s.N, 0x77u);
  for (int i = 0; i < sets; ++i) {
    uint16_t *a, *c, *x = nullptr, *y = nullptr;
    CK(hipMalloc(&a, b.a_elems * 2));
    CK(hipMalloc(&c, b.c_elems * 4));      // (fp32 results of form TN fit as well)
    fill_bf16<<<2048, 256>>>(a, b.a_elems, 0x9000u + i, 1.f);
    if (has_aux(s.epi)) {
      CK(hipMalloc(&x, b.c_elems * 2));
      fill_bf16<<<2048, 256>>>(x, b.c_elems, 0x5000u + i, 1.f);
    }
    if (has_pre(s.epi)) CK(hipMalloc(&y, b.c_elems * 2));
    b.A.push_back(a); b.C.push_back(c); b.aux.push_back(x); b.aux_out.push_back(y);
  }
  CK(hipDeviceSynchronize());
  return b;
}
static void release(Buffers &b) {
  for (auto p : b.A) CK(hipFree(p));
  for (auto p : b.C) CK(hipFree(p));
  for (auto p : b.aux) if (p) CK(hipFree(p));
  for (auto p : b.aux_out) if (p) CK(hipFree(p));
  CK(hipFree(b.B));
  CK(hipFree(b.bias));
}

static float *g_sk_ws = nullptr;
static float *sk_workspace() {
  if (!g_sk_ws) {
    CK(hipMalloc(&g_sk_ws, (size_t)gps_gemm_sk_workspace_bytes()));
    CK(hipMemset(g_sk_ws, 0, 4096));
  }
  return g_sk_ws;
}
static void sk_check(const char *what) {
  unsigned int err = 0;
  CK(hipMemcpy(&err, reinterpret_cast<unsigned int *>(sk_workspace()) + 256, 4, hipMemcpyDeviceToHost));
  if (err) { fprintf(stderr, "stream-K wait expired (%s)\n", what); exit(3); }
}

static int launch(const Shape &s, const Buffers &b, int set, int variant, hipStream_t st) {
  gps_gemm_args a;
  memset(&a, 0, sizeof(a));
  if (variant == 13) a.workspace = sk_workspace();
  a.form = s.form; a.epilogue = s.epi; a.M = s.M; a.N = s.N; a.K = s.K; a.splits = 1; a.variant = variant;
  a.A = b.A[set]; a.lda = s.form == 2 ? s.M : s.K;  // TN: dY (K, M)
  a.B = b.B; a.ldb = s.form == 0 ? s.K : s.N;      // NT: W (N, K); NN: W (K, N); TN: X (K, N)
  if (s.form == 2) {
    a.splits = gps_gemm_pick_splits(2, s.M, s.N, s.K);
    static float *ws = nullptr;
    if (!ws) CK(hipMalloc(&ws, (size_t)512 << 20));
    a.workspace = ws;
  }
  a.C = b.C[set]; a.ldc = s.N;
  a.bias = (s.epi == 0 || s.epi == 1 || s.epi == 2 || s.epi == 8) ? b.bias : nullptr;
  a.aux = b.aux[set]; a.ldaux = s.N;
  a.aux_out = b.aux_out[set]; a.ldaux_out = s.N;
  a.p_drop = (s.epi == 8 || s.epi == 2 || s.epi == 4 || s.epi == 1 || s.epi == 3) ? 0.1f : 0.f;
  a.seed = 42;
  return gps_gemm_bf16(&a, (gps_stream_t)st);
}

static double time_variant(const Shape &s, const Buffers &b, int variant, int rounds, int inner, hipStream_t st) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int i = 0; i < 2; ++i)
    if (launch(s, b, i % b.sets, variant, st) != 0) return -1.0;
  CK(hipStreamSynchronize(st));
  std::vector<double> t;
  int set = 0;
  for (int r = 0; r < rounds; ++r) {
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < inner; ++i) { launch(s, b, set, variant, st); set = (set + 1) % b.sets; }
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    t.push_back(1e3 * ms / inner);
  }
  std::sort(t.begin(), t.end());
  CK(hipEventDestroy(e0));
  CK(hipEventDestroy(e1));
  return t[t.size() / 2];
}

static const char *form_name(int f) { return f == 0 ? "nt" : f == 1 ? "nn" : "tn"; }

static std::vector<int> parse_list(const char *s) {
  std::vector<int> v;
  for (const char *p = s; *p;) {
    v.push_back(atoi(p));
    while (*p && *p != ',') ++p;
    if (*p == ',') ++p;
  }
  return v;
}

static void cmd_bench(int argc, char **argv) {
  bool warm = false;
  int rounds = 7, inner = 8;
  std::string set = "in_step";
  std::vector<int> variants = {-1, 6, 7, 12};
  for (int i = 2; i < argc; ++i) {
    if (!strcmp(argv[i], "--warm")) warm = true;
    else if (!strcmp(argv[i], "--rounds")) rounds = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--inner")) inner = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--set")) set = argv[++i];
    else if (!strcmp(argv[i], "--variants")) variants = parse_list(argv[++i]);
  }
  const Shape *shapes = set == "small" ? kSmall : kInStep;
  const int n = set == "small" ? (int)(sizeof(kSmall) / sizeof(Shape)) : (int)(sizeof(kInStep) / sizeof(Shape));
  hipStream_t st;
  CK(hipStreamCreate(&st));
  unsigned long long *cnt;
  float *maxd;
  CK(hipMalloc(&cnt, 8));
  CK(hipMalloc(&maxd, 4));
  printf("{\"mode\": \"%s\", \"rounds\": %d, \"inner\": %d, \"rows\": [\n", warm ? "warm" : "cold", rounds, inner);
  for (int si = 0; si < n; ++si) {
    const Shape &s = shapes[si];
    const size_t per_set = ((size_t)s.M * s.K + (size_t)s.M * s.N * (1 + has_aux(s.epi) + has_pre(s.epi))) * 2;
    int sets = warm ? 1 : (int)std::min<size_t>(12, std::max<size_t>(2, (600ull << 20) / per_set + 1));
    Buffers b = make(s, sets);
    const double flops = 2.0 * s.M * s.N * s.K;
    printf(" {\"what\": \"%s\", \"form\": \"%s\", \"epi\": %d, \"M\": %d, \"N\": %d, \"K\": %d, \"sets\": %d, \"default_variant\": %d",
           s.what, form_name(s.form), s.epi, s.M, s.N, s.K, sets, gps_gemm_pick_variant(s.form, s.M, s.N, s.K, 1));
    // reference output: variant 7 on set 0
    launch(s, b, 0, 7, st);
    CK(hipStreamSynchronize(st));
    uint16_t *ref;
    CK(hipMalloc(&ref, b.c_elems * 2));
    CK(hipMemcpy(ref, b.C[0], b.c_elems * 2, hipMemcpyDeviceToDevice));
    if (s.epi == 0) {
      // variant 7 against the naive reference (layout / exchange errors would be far beyond one bf16 rounding of |C| ~ 1)
      uint16_t *naive;
      CK(hipMalloc(&naive, b.c_elems * 2));
      ref_gemm_bias<<<(unsigned)((b.c_elems + 255) / 256), 256, 0, st>>>(s.form, s.M, s.N, s.K, b.A[0], b.B, b.bias, naive);
      CK(hipMemsetAsync(cnt, 0, 8, st));
      count_far<<<1024, 256, 0, st>>>(ref, naive, b.c_elems, 0.0625f, cnt);
      unsigned long long far = 0;
      CK(hipMemcpyAsync(&far, cnt, 8, hipMemcpyDeviceToHost, st));
      CK(hipStreamSynchronize(st));
      printf(", \"v7_far_from_naive\": %llu", far);
      CK(hipFree(naive));
    }
    for (int v : variants) {
      const double us = time_variant(s, b, v, rounds, inner, st);
      // correctness vs variant 7 (same set, same seed)
      CK(hipMemset(b.C[0], 0xFF, b.c_elems * 2));
      launch(s, b, 0, v, st);
      CK(hipMemsetAsync(cnt, 0, 8, st));
      CK(hipMemsetAsync(maxd, 0, 4, st));
      diff_bf16<<<1024, 256, 0, st>>>(ref, b.C[0], b.c_elems, cnt, maxd);
      unsigned long long hc = 0;
      float hm = 0.f;
      CK(hipMemcpyAsync(&hc, cnt, 8, hipMemcpyDeviceToHost, st));
      CK(hipMemcpyAsync(&hm, maxd, 4, hipMemcpyDeviceToHost, st));
      CK(hipStreamSynchronize(st));
      if (v == 13) sk_check(s.what);
      printf(", \"v%d\": {\"us\": %.2f, \"TF\": %.1f, \"diff_words\": %llu, \"max_diff\": %.4g}", v, us, us > 0 ? flops / us * 1e-6 : 0.0, hc, hm);
      fflush(stdout);
    }
    printf("}%s\n", si + 1 < n ? "," : "");
    CK(hipFree(ref));
    release(b);
  }
  printf("]}\n");
}

static void summarize_trace(const std::vector<unsigned long long> &tr, int wgs, double wall_us) {
  using gps_gemm::kTraceSlots;
  // shader clock from the two stamp kinds of every workgroup (s_memrealtime ticks at 100 MHz)
  std::vector<double> mhz, pro, loop, epi_issue, epi_drain, total, start_rt, end_rt, ktile;
  unsigned long long rt0 = ~0ull;
  for (int w = 0; w < wgs; ++w) {
    const unsigned long long *t = &tr[(size_t)w * kTraceSlots];
    if (t[0] == 0 || t[5] == 0) continue;
    rt0 = std::min(rt0, t[4]);
  }
  for (int w = 0; w < wgs; ++w) {
    const unsigned long long *t = &tr[(size_t)w * kTraceSlots];
    if (t[0] == 0 || t[5] == 0) continue;
    const double rt_us = (double)(t[5] - t[4]) / 100.0;
    const double cyc = (double)(t[6] - t[0]);
    if (rt_us > 1.0) mhz.push_back(cyc / rt_us);
    pro.push_back((double)(t[1] - t[0]));
    loop.push_back((double)(t[2] - t[1]));
    epi_issue.push_back((double)(t[3] - t[2]));
    epi_drain.push_back((double)(t[6] - t[3]));
    total.push_back(cyc);
    start_rt.push_back((double)(t[4] - rt0) / 100.0);
    end_rt.push_back((double)(t[5] - rt0) / 100.0);
    for (int k = 9; k < 32; ++k)
      if (t[k] && t[k - 1]) ktile.push_back((double)(t[k] - t[k - 1]));
  }
  auto med = [](std::vector<double> v) { if (v.empty()) return 0.0; std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
  auto pct = [](std::vector<double> v, double p) { if (v.empty()) return 0.0; std::sort(v.begin(), v.end()); return v[(size_t)(p * (v.size() - 1))]; };
  const double f = med(mhz) > 0 ? med(mhz) : 2400.0;       // cycles per us
  printf("{\"workgroups_traced\": %zu, \"wall_us\": %.2f, \"shader_MHz\": %.0f,\n", total.size(), wall_us, f);
  printf(" \"us\": {\"prologue\": %.2f, \"main_loop\": %.2f, \"per_k_tile\": %.3f, \"per_k_tile_p90\": %.3f, \"epilogue_issue\": %.2f, \"epilogue_drain\": %.2f, \"workgroup\": %.2f, \"workgroup_p90\": %.2f},\n",
         med(pro) / f, med(loop) / f, med(ktile) / f, pct(ktile, 0.9) / f, med(epi_issue) / f, med(epi_drain) / f, med(total) / f, pct(total, 0.9) / f);
  // phase stamps of K tile 8 (waves 0 and 4): medians of the 16 intervals, in shader cycles
  for (int wv = 0; wv < 2; ++wv) {
    printf(" \"phase_cycles_wave%d\": [", wv ? 4 : 0);
    for (int i = 0; i < 16; ++i) {
      std::vector<double> d;
      for (int w = 0; w < wgs; ++w) {
        const unsigned long long *t = &tr[(size_t)w * kTraceSlots + 32 + 32 * wv];
        if (t[i] && t[i + 1] && t[i + 1] > t[i]) d.push_back((double)(t[i + 1] - t[i]));
      }
      printf("%s%.0f", i ? ", " : "", med(d));
    }
    printf("],\n");
  }
  printf(" \"start_us\": {\"p10\": %.2f, \"p50\": %.2f, \"p90\": %.2f, \"max\": %.2f}, \"end_us\": {\"p10\": %.2f, \"p50\": %.2f, \"p90\": %.2f, \"max\": %.2f}}\n",
         pct(start_rt, 0.1), pct(start_rt, 0.5), pct(start_rt, 0.9), pct(start_rt, 1.0), pct(end_rt, 0.1), pct(end_rt, 0.5), pct(end_rt, 0.9), pct(end_rt, 1.0));
}

static void cmd_trace(int argc, char **argv) {
  if (argc < 8) { fprintf(stderr, "trace FORM EPI M N K VARIANT\n"); exit(1); }
  Shape s = {atoi(argv[2]), atoi(argv[3]), atoi(argv[4]), atoi(argv[5]), atoi(argv[6]), "trace"};
  const int variant = atoi(argv[7]);
  Buffers b = make(s, 2);
  hipStream_t st;
  CK(hipStreamCreate(&st));
  const int max_wgs = 1 << 15;
  unsigned long long *tr;
  CK(hipMalloc(&tr, (size_t)max_wgs * gps_gemm::kTraceSlots * 8));
  launch(s, b, 0, variant, st);
  launch(s, b, 1, variant, st);
  CK(hipStreamSynchronize(st));
  CK(hipMemset(tr, 0, (size_t)max_wgs * gps_gemm::kTraceSlots * 8));
  gps_gemm::g_probe_trace = tr;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0, st));
  launch(s, b, 0, variant, st);
  CK(hipEventRecord(e1, st));
  CK(hipEventSynchronize(e1));
  gps_gemm::g_probe_trace = nullptr;
  float ms = 0.f;
  CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> h((size_t)max_wgs * gps_gemm::kTraceSlots);
  CK(hipMemcpy(h.data(), tr, h.size() * 8, hipMemcpyDeviceToHost));
  if (argc > 8) {            // raw stamps of the first 512 workgroups (offline analysis: tools/sk_trace_segments.py)
    FILE *f = fopen(argv[8], "wb");
    if (f) { fwrite(h.data(), 8, (size_t)512 * gps_gemm::kTraceSlots, f); fclose(f); }
  }
  printf("{\"form\": \"%s\", \"epi\": %d, \"M\": %d, \"N\": %d, \"K\": %d, \"variant\": %d, \"trace\":\n", form_name(s.form), s.epi, s.M, s.N, s.K, variant);
  summarize_trace(h, max_wgs, 1e3 * ms);
  printf("}\n");
  release(b);
}

static void cmd_loop(int argc, char **argv) {
  if (argc < 9) { fprintf(stderr, "loop FORM EPI M N K VARIANT ITERS\n"); exit(1); }
  Shape s = {atoi(argv[2]), atoi(argv[3]), atoi(argv[4]), atoi(argv[5]), atoi(argv[6]), "loop"};
  const int variant = atoi(argv[7]), iters = atoi(argv[8]);
  Buffers b = make(s, 4);
  hipStream_t st;
  CK(hipStreamCreate(&st));
  for (int i = 0; i < iters; ++i) launch(s, b, i % 4, variant, st);
  CK(hipStreamSynchronize(st));
  release(b);
}

// ---- grouped launches: the same Linear of the text stack and of the object stack as one launch vs two -------------------
static void fill_args(gps_gemm_args &a, const Shape &s, const Buffers &b, int set, int variant) {
  memset(&a, 0, sizeof(a));
  a.form = s.form; a.epilogue = s.epi; a.M = s.M; a.N = s.N; a.K = s.K; a.splits = 1; a.variant = variant;
  a.A = b.A[set]; a.lda = s.K;
  a.B = b.B; a.ldb = s.form == 0 ? s.K : s.N;
  a.C = b.C[set]; a.ldc = s.N;
  a.bias = (s.epi == 0 || s.epi == 1 || s.epi == 2 || s.epi == 8) ? b.bias : nullptr;
  a.aux = b.aux[set]; a.ldaux = s.N;
  a.aux_out = b.aux_out[set]; a.ldaux_out = s.N;
  a.p_drop = (s.epi == 8 || s.epi == 2 || s.epi == 4 || s.epi == 1 || s.epi == 3) ? 0.1f : 0.f;
  a.seed = 42;
}
static void cmd_group(int argc, char **argv) {
  struct Pair { Shape a, b; };
  static const Pair pairs[] = {
      {{0, 0, 12608, 2304, 768, "text qkv fwd"}, {0, 0, 5120, 2376, 768, "obj qkv fwd"}},
      {{0, 0, 12608, 768, 768, "text out fwd"}, {0, 0, 5120, 768, 768, "obj out fwd"}},
      {{0, 8, 12608, 3072, 768, "text ffn1 fwd"}, {0, 8, 5120, 2048, 768, "obj ffn1 fwd"}},
      {{0, 0, 12608, 768, 3072, "text ffn2 fwd"}, {0, 0, 5120, 768, 2048, "obj ffn2 fwd"}},
      {{1, 9, 12608, 3072, 768, "text ffn1 dgrad"}, {1, 9, 5120, 2048, 768, "obj ffn1 dgrad"}},
      {{1, 0, 12608, 768, 3072, "text ffn2 dgrad"}, {1, 0, 5120, 768, 2048, "obj ffn2 dgrad"}},
      {{1, 0, 12608, 768, 768, "text out dgrad"}, {1, 0, 5120, 768, 768, "obj out dgrad"}},
      {{1, 0, 12608, 768, 2304, "text qkv dgrad"}, {1, 0, 5120, 768, 2376, "obj qkv dgrad"}},
  };
  const int rounds = 7, inner = 6, sets = 3;
  hipStream_t st;
  CK(hipStreamCreate(&st));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  unsigned long long *cnt;
  float *maxd;
  CK(hipMalloc(&cnt, 8));
  CK(hipMalloc(&maxd, 4));
  double sum_sep = 0, sum_best = 0, sum_grp = 0;
  printf("%-34s %10s %10s %10s   %s\n", "pair", "separate", "best 7/12", "grouped", "grouped == variant 12 (words differing)");
  for (const Pair &pr : pairs) {
    Buffers ba = make(pr.a, sets), bb = make(pr.b, sets);
    auto time_it = [&](auto &&body) {
      std::vector<double> t;
      int set = 0;
      for (int r = 0; r < rounds + 1; ++r) {
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < inner; ++i) { body(set); set = (set + 1) % sets; }
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms = 0.f;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (r) t.push_back(1e3 * ms / inner);
      }
      std::sort(t.begin(), t.end());
      return t[t.size() / 2];
    };
    const double sep = time_it([&](int set) { launch(pr.a, ba, set, -1, st); launch(pr.b, bb, set, -1, st); });
    double best = 1e30;
    for (int va : {7, 12})
      for (int vb : {6, 7, 12})
        best = std::min(best, time_it([&](int set) { launch(pr.a, ba, set, va, st); launch(pr.b, bb, set, vb, st); }));
    int status = 0;
    const double grp = time_it([&](int set) {
      gps_gemm_args a[2];
      fill_args(a[0], pr.a, ba, set, -1);
      fill_args(a[1], pr.b, bb, set, -1);
      status |= gps_gemm_bf16_grouped(a, 2, (gps_stream_t)st);
    });
    // correctness: grouped outputs against variant 12 of each product (same kernel body: bit-equal)
    unsigned long long diff = 0;
    for (int which = 0; which < 2; ++which) {
      const Shape &s = which ? pr.b : pr.a;
      Buffers &b = which ? bb : ba;
      uint16_t *ref;
      CK(hipMalloc(&ref, b.c_elems * 2));
      launch(s, b, 0, 12, st);
      CK(hipMemcpyAsync(ref, b.C[0], b.c_elems * 2, hipMemcpyDeviceToDevice, st));
      CK(hipMemsetAsync(b.C[0], 0xFF, b.c_elems * 2, st));
      gps_gemm_args a[2];
      fill_args(a[0], pr.a, ba, 0, -1);
      fill_args(a[1], pr.b, bb, 0, -1);
      status |= gps_gemm_bf16_grouped(a, 2, (gps_stream_t)st);
      CK(hipMemsetAsync(cnt, 0, 8, st));
      CK(hipMemsetAsync(maxd, 0, 4, st));
      diff_bf16<<<1024, 256, 0, st>>>(ref, b.C[0], b.c_elems, cnt, maxd);
      unsigned long long hc = 0;
      CK(hipMemcpyAsync(&hc, cnt, 8, hipMemcpyDeviceToHost, st));
      CK(hipStreamSynchronize(st));
      diff += hc;
      CK(hipFree(ref));
    }
    printf("%-16s + %-15s %10.1f %10.1f %10.1f   status %d, %llu\n", pr.a.what, pr.b.what, sep, best, grp, status, diff);
    fflush(stdout);
    sum_sep += sep; sum_best += best; sum_grp += grp;
    release(ba);
    release(bb);
  }
  printf("%-34s %10.1f %10.1f %10.1f\n", "sum (one layer, fwd + dgrad)", sum_sep, sum_best, sum_grp);
}

// ---- the grouped weight-gradient launch of one step (48 of its 69 problems): per-workgroup K-tile rate and finish times -------
static void cmd_wgrad(int argc, char **argv) {
  struct W { int M, N, K; };
  std::vector<W> ws;
  for (int l = 0; l < 4; ++l) {
    for (W w : {W{2304, 768, 12608}, W{768, 768, 12608}, W{3072, 768, 12608}, W{768, 3072, 12608}}) ws.push_back(w);
    for (W w : {W{2376, 768, 5120}, W{768, 768, 5120}, W{2048, 768, 5120}, W{768, 2048, 5120}}) ws.push_back(w);
    for (W w : {W{2304, 768, 8320}, W{768, 768, 8320}, W{2048, 768, 8320}, W{768, 2048, 8320}}) ws.push_back(w);
  }
  // operands: dY (K, M), X (K, N) per problem, shared buffers per (K, width) class to bound memory
  auto alloc_bf16 = [&](size_t elems, unsigned seed) {
    uint16_t *p;
    CK(hipMalloc(&p, elems * 2));
    fill_bf16<<<2048, 256>>>(p, elems, seed, 1.f);
    return p;
  };
  std::vector<gps_wgrad_problem> probs;
  std::vector<void *> keep;
  double flops = 0;
  for (size_t i = 0; i < ws.size(); ++i) {
    const W &w = ws[i];
    uint16_t *A = alloc_bf16((size_t)w.K * w.M, 100 + (unsigned)i), *B = alloc_bf16((size_t)w.K * w.N, 300 + (unsigned)i);
    float *C, *cs;
    CK(hipMalloc(&C, (size_t)w.M * w.N * 4));
    CK(hipMalloc(&cs, (size_t)w.M * 4));
    gps_wgrad_problem q;
    memset(&q, 0, sizeof(q));
    q.M = w.M; q.N = w.N; q.K = w.K; q.accumulate = 0;
    q.A = A; q.lda = w.M; q.B = B; q.ldb = w.N; q.C = C; q.ldc = w.N; q.colsum = cs;
    probs.push_back(q);
    flops += 2.0 * w.M * w.N * w.K;
  }
  CK(hipDeviceSynchronize());
  hipStream_t st;
  CK(hipStreamCreate(&st));
  const int max_wgs = 512;
  unsigned long long *tr;
  CK(hipMalloc(&tr, (size_t)2 * max_wgs * gps_gemm::kTraceSlots * 8));      // [the launch's stamps | the tile function's own]
  for (int mode = 1; mode >= 0; --mode) {
    gps_gemm_wgrad_grouped_set_xcd_queues(mode);
    for (int i = 0; i < 2; ++i) gps_gemm_wgrad_grouped(probs.data(), (int)probs.size(), (gps_stream_t)st);
    CK(hipStreamSynchronize(st));
    CK(hipMemset(tr, 0, (size_t)2 * max_wgs * gps_gemm::kTraceSlots * 8));
    gps_gemm::g_probe_trace = tr;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, st));
    gps_gemm_wgrad_grouped(probs.data(), (int)probs.size(), (gps_stream_t)st);
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    gps_gemm::g_probe_trace = nullptr;
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h((size_t)2 * max_wgs * gps_gemm::kTraceSlots);
    CK(hipMemcpy(h.data(), tr, h.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> per_kt, busy, endt;
    unsigned long long rt0 = ~0ull, kts = 0, tiles = 0;
    for (int w = 0; w < max_wgs; ++w) if (h[(size_t)w * gps_gemm::kTraceSlots]) rt0 = std::min(rt0, h[(size_t)w * gps_gemm::kTraceSlots + 4]);
    for (int w = 0; w < max_wgs; ++w) {
      const unsigned long long *t = &h[(size_t)w * gps_gemm::kTraceSlots];
      if (!t[0] || !t[6]) continue;
      const double us = (double)(t[5] - t[4]) / 100.0;
      per_kt.push_back(us / (double)t[7]);
      busy.push_back(us);
      endt.push_back((double)(t[5] - rt0) / 100.0);
      kts += t[7]; tiles += t[8];
    }
    std::sort(per_kt.begin(), per_kt.end()); std::sort(busy.begin(), busy.end()); std::sort(endt.begin(), endt.end());
    const size_t n = per_kt.size();
    printf("xcd_queues %d: wall %.1f us (%.0f TFLOP/s), %zu workgroups, %llu tiles, %llu K tiles; per K tile us p10 %.3f p50 %.3f p90 %.3f; "
           "workgroup finish us p10 %.1f p50 %.1f p90 %.1f max %.1f; ideal at p50 rate %.1f us\n",
           mode, 1e3 * ms, flops / (1e3 * ms) * 1e-6, n, tiles, kts, per_kt[n / 10], per_kt[n / 2], per_kt[n * 9 / 10], endt[n / 10], endt[n / 2],
           endt[n * 9 / 10], endt[n - 1], (double)kts * per_kt[n / 2] / (double)n);
    // phases of the 9th K tile of each workgroup's LAST tile (waves 0 / 4), medians over the workgroups, shader cycles
    // incl. ~170 per stamp: [reads, barrier + lgkm, mfma, barrier] x 4
    for (int wv = 0; wv < 2; ++wv) {
      printf("  phase cycles wave %d:", wv ? 4 : 0);
      for (int i = 0; i < 16; ++i) {
        std::vector<double> d;
        for (int w = 0; w < max_wgs; ++w) {
          const unsigned long long *t = &h[(size_t)(max_wgs + w) * gps_gemm::kTraceSlots + 32 + 32 * wv];
          if (t[i] && t[i + 1] && t[i + 1] > t[i]) d.push_back((double)(t[i + 1] - t[i]));
        }
        std::sort(d.begin(), d.end());
        printf("%s%.0f", (i % 4) ? " " : " | ", d.empty() ? 0.0 : d[d.size() / 2]);
      }
      printf("\n");
    }
  }
}

// ---- store-path microbenchmark: what one CU's 8 waves can push per clock, by access pattern ---------------------------
// pattern 0: 1 KiB contiguous per wave-instruction; 1: 16 rows x 64 B (the GEMM epilogue's 16-byte stores today);
// 2: 64 rows x 16 B (row per lane); 3: 8 rows x 128 B; 4: 4 rows x 256 B; 5: 2 rows x 512 B.  Row pitch 512 B... `pitch`.
template <int PATTERN, bool SC1>
__global__ __launch_bounds__(512) void store_pattern_kernel(unsigned char *base, int pitch, int iters, unsigned long long *cycles, int cold) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned char *tile = base + (size_t)blockIdx.x * (256 * (size_t)pitch) * (cold ? iters : 1);       // 256 rows per workgroup (and pass, cold)
  constexpr int SEG = PATTERN == 0 ? 1024 : PATTERN == 1 ? 64 : PATTERN == 2 ? 16 : PATTERN == 3 ? 128 : PATTERN == 4 ? 256 : 512;
  constexpr int LPR = SEG / 16;                      // lanes per row segment
  constexpr int ROWS = 64 / LPR;                     // rows per wave-instruction
  const gps_gemm::u32x4 v = {(unsigned)lane, 1u, 2u, 3u};
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it, tile += cold ? 256 * (size_t)pitch : 0) {
    // every wave stores 16 KB per pass: 16 instructions
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      size_t off;
      if (PATTERN == 0) off = (size_t)((wave * 16 + j) * 1024 + lane * 16);
      else {
        // instruction j of wave w covers rows r0 .. r0 + ROWS - 1 at column byte c0 (the wave owns a 32-row x 512-byte band)
        const int per_band = 512 / SEG;              // instructions side by side in a band of ROWS rows
        const int r0 = wave * 32 + (j / per_band) * ROWS % 32, c0 = (j % per_band) * SEG;
        off = (size_t)(r0 + lane / LPR) * pitch + c0 + (lane % LPR) * 16;
      }
      if (SC1) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(tile + off), "v"(v) : "memory");
      else *reinterpret_cast<gps_gemm::u32x4 *>(tile + off) = v;
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t2 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) { cycles[blockIdx.x * 2] = t1 - t0; cycles[blockIdx.x * 2 + 1] = t2 - t0; }
}
template <bool SC1>
__global__ __launch_bounds__(512) void load_slab_kernel(const unsigned char *base, int iters, unsigned long long *cycles, float *sink) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned char *slab = base + (size_t)blockIdx.x * 262144;
  gps_gemm::f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r0 = 0; r0 < 32; r0 += 8) {
      gps_gemm::f32x4 p[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const unsigned char *a = slab + (size_t)((wave * 32 + r0 + r) * 64 + lane) * 16;
        if (SC1) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(p[r]) : "v"(a) : "memory");
        else p[r] = *reinterpret_cast<const gps_gemm::f32x4 *>(a);
      }
      if (SC1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int r = 0; r < 8; ++r) acc += p[r];
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) cycles[blockIdx.x * 2] = cycles[blockIdx.x * 2 + 1] = t1 - t0;
  if (acc[0] == 123.456f) sink[0] = acc[1];
}

static void cmd_storebw(int argc, char **argv) {
  const int iters = 8;
  unsigned char *buf;
  unsigned long long *cyc;
  float *sink;
  const size_t bytes = (size_t)256 * 262144 * 2;
  CK(hipMalloc(&buf, bytes));
  CK(hipMemset(buf, 0, bytes));
  CK(hipMalloc(&cyc, 256 * 2 * 8));
  CK(hipMalloc(&sink, 16));
  std::vector<unsigned long long> h(512);
  auto report = [&](const char *name, int wgs, double bytes_per_wg) {
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(h.data(), cyc, 512 * 8, hipMemcpyDeviceToHost));
    std::vector<double> a, b;
    for (int i = 0; i < wgs; ++i) { a.push_back((double)h[2 * i]); b.push_back((double)h[2 * i + 1]); }
    std::sort(a.begin(), a.end());
    std::sort(b.begin(), b.end());
    printf("  %-44s wgs %3d: issue %8.0f cyc (%.1f B/clk/CU)  drained %8.0f cyc (%.1f B/clk/CU)\n", name, wgs, a[wgs / 2],
           bytes_per_wg / a[wgs / 2], b[wgs / 2], bytes_per_wg / b[wgs / 2]);
  };
  for (int wgs : {32, 256}) {
    const double per = 131072.0 * iters;
#define RUN(PAT, SC, NAME)                                                                   \
  store_pattern_kernel<PAT, SC><<<wgs, 512>>>(buf, 512, iters, cyc, 0);                        \
  store_pattern_kernel<PAT, SC><<<wgs, 512>>>(buf, 512, iters, cyc, 0);                        \
  report(NAME, wgs, per);                                                                    \
  CK(hipMemset(buf, 1, bytes));                                                              \
  store_pattern_kernel<PAT, SC><<<wgs, 512>>>(buf, 512, 4, cyc, 1);                            \
  report(NAME " [cold lines, 4 passes]", wgs, per / 2);
    RUN(0, false, "store 1 KiB contiguous / instruction");
    RUN(5, false, "store 2 rows x 512 B");
    RUN(4, false, "store 4 rows x 256 B");
    RUN(3, false, "store 8 rows x 128 B");
    RUN(1, false, "store 16 rows x 64 B (epilogue today)");
    RUN(0, true, "store 1 KiB contiguous, sc1 (slab)");
    RUN(1, true, "store 16 rows x 64 B, sc1");
#undef RUN
    load_slab_kernel<false><<<wgs, 512>>>(buf, iters, cyc, sink);
    load_slab_kernel<false><<<wgs, 512>>>(buf, iters, cyc, sink);
    report("load slab 256 KiB, plain (L2 warm)", wgs, 262144.0 * iters);
    load_slab_kernel<true><<<wgs, 512>>>(buf, iters, cyc, sink);
    load_slab_kernel<true><<<wgs, 512>>>(buf, iters, cyc, sink);
    report("load slab 256 KiB, sc1", wgs, 262144.0 * iters);
  }
}

int main(int argc, char **argv) {
  if (argc < 2) { fprintf(stderr, "usage: gemm_probe bench|trace|loop ...\n"); return 1; }
  if (!strcmp(argv[1], "bench")) cmd_bench(argc, argv);
  else if (!strcmp(argv[1], "trace")) cmd_trace(argc, argv);
  else if (!strcmp(argv[1], "loop")) cmd_loop(argc, argv);
  else if (!strcmp(argv[1], "storebw")) cmd_storebw(argc, argv);
  else if (!strcmp(argv[1], "group")) cmd_group(argc, argv);
  else if (!strcmp(argv[1], "wgrad")) cmd_wgrad(argc, argv);
  else { fprintf(stderr, "unknown mode %s\n", argv[1]); return 1; }
  return 0;
}
