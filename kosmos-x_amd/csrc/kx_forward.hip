// Stage-level entry points of libkosmosx_hip.so: the launch sequences for the CLIP ViT-L/14 tower,
// the Perceiver resampler (+image_proj) and the Magneto sub-LN / XPos decoder.  Pure launch
// orchestration: no allocation, no synchronisation, everything asynchronous on the caller's stream
// (hipGraph-capturable).  Scratch comes from one caller-owned workspace carved deterministically.
#include <stdarg.h>
#include <atomic>
#include <mutex>
#include <vector>
#include "kx_common.h"

// ---------------------------------------------------------------------------------------------
// error plumbing
// ---------------------------------------------------------------------------------------------
static thread_local char g_err[512] = {0};
void kx_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" int kx_version(void) { return KX_ABI_VERSION; }
extern "C" int kx_last_error(char* buf, size_t n) {
  const size_t len = strlen(g_err);
  if (buf && n) {
    const size_t c = len < n - 1 ? len : n - 1;
    memcpy(buf, g_err, c);
    buf[c] = 0;
  }
  return (int)len;
}

// ---------------------------------------------------------------------------------------------
// tuning knobs (kernel-variant A/B from one process; defaults are the shipped configuration)
// ---------------------------------------------------------------------------------------------
// Process-wide A/B switches (not part of the forward's state: defaults are the shipped configuration).  Atomics, so that
// a tool flipping a knob from one thread while another thread launches is a data-race-free read of either value.
static std::atomic<int> g_tuning[KX_TUNE_COUNT];
int kx_tuning_get(int key) { return (key >= 0 && key < KX_TUNE_COUNT) ? g_tuning[key].load(std::memory_order_relaxed) : 0; }
extern "C" int kx_set_tuning(int key, int value) {
  if (key < 0 || key >= KX_TUNE_COUNT) {
    kx_set_error("kx_set_tuning: unknown key %d", key);
    return KX_ERR_INVALID_ARG;
  }
  g_tuning[key].store(value, std::memory_order_relaxed);
  return KX_OK;
}

// ---------------------------------------------------------------------------------------------
// launch timing
// ---------------------------------------------------------------------------------------------
namespace {
// Launch timing is a process-wide recorder (one list of records, in launch order per thread): every access takes
// g_prof_mu, so launches from several host threads / streams interleave safely; the enable flag is an atomic that
// the fast path (profiling off) reads without the lock.
struct ProfRec { int kind; int64_t a, b, c; hipEvent_t e0, e1; };
std::vector<ProfRec> g_prof;
std::vector<hipEvent_t> g_prof_pool;
std::mutex g_prof_mu;
std::atomic<bool> g_prof_enabled{false};
thread_local size_t g_prof_open = 0;      // index of this thread's record between kx_prof_begin and kx_prof_end
hipEvent_t prof_event() {
  if (!g_prof_pool.empty()) { hipEvent_t e = g_prof_pool.back(); g_prof_pool.pop_back(); return e; }
  hipEvent_t e;
  (void)hipEventCreate(&e);
  return e;
}
}  // namespace
bool kx_prof_on() { return g_prof_enabled.load(std::memory_order_relaxed); }
void kx_prof_begin(int kind, int64_t a, int64_t b, int64_t c, hipStream_t s) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  ProfRec r{kind, a, b, c, prof_event(), prof_event()};
  (void)hipEventRecord(r.e0, s);
  g_prof_open = g_prof.size();
  g_prof.push_back(r);
}
void kx_prof_end(hipStream_t s) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  if (g_prof_open < g_prof.size()) (void)hipEventRecord(g_prof[g_prof_open].e1, s);
}
extern "C" int kx_prof_enable(int on) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  for (auto& r : g_prof) { g_prof_pool.push_back(r.e0); g_prof_pool.push_back(r.e1); }
  g_prof.clear();
  g_prof_enabled.store(on != 0, std::memory_order_relaxed);
  return KX_OK;
}
extern "C" int kx_prof_collect(kx_prof_record* out, int max_records) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  int n = 0;
  for (auto& r : g_prof) {
    if (n >= max_records) break;
    float ms = 0.f;
    (void)hipEventSynchronize(r.e1);
    (void)hipEventElapsedTime(&ms, r.e0, r.e1);
    out[n].kind = r.kind; out[n].reserved = 0; out[n].a = r.a; out[n].b = r.b; out[n].c = r.c;
    out[n].ms = ms; out[n].reserved2 = 0.f;
    ++n;
  }
  return n;
}

// split-K scratch of the stage currently being launched (set by the stage entry points, read by gemm())
thread_local void* g_splitk_ws = nullptr;
thread_local size_t g_splitk_ws_bytes = 0;
constexpr size_t KX_SPLITK_WS = 32u << 20;   // 32 MB: enough for 16 slices of every batch-1 GEMM on the path
// The last 4 KB of the stage's split-K scratch are the arrival words of the in-launch reduction (kx_gemm_args.splitk_flags: one
// per workgroup; every launch writes its own epoch, so the launches of a stream share them): cleared once per stage call on
// the stage's stream.
thread_local unsigned* g_coop_counters = nullptr;
constexpr size_t KX_COOP_WORDS = 1024;
struct SplitkScope {
  bool ok = true;
  SplitkScope(void* p, size_t n, hipStream_t s) {
    g_splitk_ws = p; g_splitk_ws_bytes = n - KX_COOP_WORDS * 4;
    g_coop_counters = p ? (unsigned*)((char*)p + n - KX_COOP_WORDS * 4) : nullptr;
    // (only when the in-launch reduction is switched on, tuning key 17 = 1: the shipped path never reads these words)
    if (g_coop_counters && kx_tuning_get(KX_TUNE_SPLITK_COOP) == 1 && hipMemsetAsync(g_coop_counters, 0, KX_COOP_WORDS * 4, s) != hipSuccess) {
      kx_set_error("clearing the split-K arrival counters failed");
      ok = false;
    }
  }
  ~SplitkScope() { g_splitk_ws = nullptr; g_splitk_ws_bytes = 0; g_coop_counters = nullptr; }
};
#define KX_SPLITK_SCOPE(ptr, stream) SplitkScope sk((ptr), KX_SPLITK_WS, (stream)); if (!sk.ok) return KX_ERR_LAUNCH
// scratch of the 256x256 kernel's pair split (kx_gemm_args.pair_ws) of the stage being launched: 4 KB of hand-off words,
// cleared once per stage call, + one 128 KB slab per workgroup of a full round
thread_local void* g_pair_ws = nullptr;
thread_local size_t g_pair_ws_bytes = 0;
constexpr size_t KX_PAIR_WS = 4096 + (size_t)256 * 131072;
struct PairScope {
  PairScope(void* p, size_t n) { g_pair_ws = p; g_pair_ws_bytes = p ? n : 0; }
  ~PairScope() { g_pair_ws = nullptr; g_pair_ws_bytes = 0; }
};

namespace {

struct Carver {
  char* base; size_t off;
  void* take(size_t bytes) {
    void* p = base ? base + off : nullptr;
    off += (bytes + 255) & ~(size_t)255;
    return p;
  }
};
// GEMM-operand activations: bf16 (2 B), fp32 (4 B), bf16x3 (3 x 2 B per value: [hi | hi | lo] rows) or f16c (4 B per
// value: [fp16 | fp8 | fp8 residual] rows) — kx_precision doc
inline size_t esz(int prec) { return (prec == KX_PREC_BF16 || prec == KX_PREC_F16) ? 2 : prec == KX_PREC_BF16X3 ? 6 : 4; }
inline int cdt(int prec) {
  return prec == KX_PREC_BF16 ? KX_BF16 : prec == KX_PREC_F16 ? KX_F16 : prec == KX_PREC_BF16X3 ? KX_BF16X3
         : prec == KX_PREC_F16C ? KX_F16C : KX_F32;
}
// q/k/v (attention inputs) and the attention arithmetic: bf16x3 keeps them in fp32 on the exact-f32 matrix instruction,
// f16c keeps them in fp32 and multiplies split fp16 (hi, lo) pairs (attn_f16s_kernel)
// plain fp16 (KX_PREC_F16) is the bf16 pipeline on fp16 values: fp16 q/k/v, the v2 attention kernel on fp16 MFMAs
inline size_t qes(int prec) { return (prec == KX_PREC_BF16 || prec == KX_PREC_F16) ? 2 : 4; }
inline int qdt(int prec) { return prec == KX_PREC_BF16 ? KX_BF16 : prec == KX_PREC_F16 ? KX_F16 : KX_F32; }
inline int aprec(int prec) {
  return prec == KX_PREC_BF16 ? KX_PREC_BF16 : prec == KX_PREC_F16 ? KX_PREC_F16 : prec == KX_PREC_F16C ? KX_PREC_F16C : KX_PREC_F32;
}
// operand row length in 2-byte units per value (the unit lda / ldc / attention output strides count for these formats)
inline int64_t kmul(int prec) { return prec == KX_PREC_BF16X3 ? 3 : prec == KX_PREC_F16C ? 2 : 1; }

// what the row-owning split-K reduce can absorb (kx_gemm_args: stats_partials, ln_out)
// GEMM family of the NEXT gemm() call of this thread, for the per-family fp8-correction assignment of KX_PREC_F16C stages
// (kx_gemm_args.f16c_corr, tuning key 16: two bits per family — 0 decoder qkv, 1 out_proj, 2 fc1, 3 fc2, 4 output projection,
// 5 every Perceiver GEMM).  gemm() consumes it; a call without KX_FAM contracts both corrections.
thread_local int g_gemm_family = -1;
#define KX_FAM(f) g_gemm_family = (f)
// The shipped assignment (tuning key 16 = -1 selects it too): DESIGN.md §5 "one correction instead of two, per family".
constexpr int KX_F16C_CORR_DEFAULT = 0;
static int f16c_corr_of(int family) {
  if (family < 0) return KX_CORR_BOTH;
  int v = kx_tuning_get(KX_TUNE_F16C_CORR);
  if (v < 0) v = KX_F16C_CORR_DEFAULT;
  return (v >> (2 * family)) & 3;
}

struct RowFusion {
  const float* partials = nullptr; int64_t nseg = 0, seg = 0;            // folded-LN statistics straight from the producer
  float* scratch = nullptr;                                              // tile kernels: kx_gemm_args.row_stats_scratch (finalised (mean, rstd))
  void* ln_out = nullptr; int ln_dt = 0; const float* ln_g = nullptr; const float* ln_b = nullptr;   // the LayerNorm that follows
  float eps = 0.f;
};

// folded pre-LayerNorm, producer side of a residual GEMM (kx_gemm_args.ln_operand_out)
struct LnOp { void* out; int dt; float* stats; };
inline bool fold_prec(int prec) { return prec == KX_PREC_BF16 || prec == KX_PREC_F16 || prec == KX_PREC_F16C; }

int gemm(const void* A, int64_t lda, const void* W, int64_t K, void* C, int64_t ldc, int cdtype, int64_t M, int64_t N,
         const float* bias, const float* residual, int act, float qscale, int64_t qcols, int prec, hipStream_t s,
         const float* xq_cs = nullptr, const float* xq_ss = nullptr, const float* xk_cs = nullptr,
         const float* xk_ss = nullptr, int64_t xT = 0, int64_t xdim = 0, const float* row_stats = nullptr,
         const float* colsum = nullptr, float* stats_out = nullptr, const RowFusion* rf = nullptr,
         const LnOp* lo = nullptr) {
  kx_gemm_args g;
  memset(&g, 0, sizeof(g));
  // bf16x3: the same bf16 kernels over the 3K-wide split operands ([hi|hi|lo] activations x [hi|lo|hi] weights)
  // f16c: fp16 + fp8 correction segments, 4K bytes per row; a packed weight matrix is N rows followed by N scale bytes
  const int64_t km = kmul(prec);
  const bool f16c = prec == KX_PREC_F16C;
  g.A = A; g.lda = lda * km; g.W = W; g.ldw = K * km; g.C = C;
  g.ldc = cdtype == KX_BF16X3 ? 3 * ldc : cdtype == KX_F16C ? 2 * ldc : ldc; g.cdt = cdtype;
  g.bias = bias; g.residual = residual; g.ldr = ldc; g.M = M; g.N = N; g.K = f16c ? K : K * km;
  if (f16c) g.w_scale = (const uint8_t*)W + (size_t)N * (size_t)K * 4;
  g.f16c_corr = f16c ? f16c_corr_of(g_gemm_family) : KX_CORR_BOTH;
  g_gemm_family = -1;
  g.act = act; g.qscale = qscale; g.qcols = qcols;
  g.xq_cs = xq_cs; g.xq_ss = xq_ss; g.xk_cs = xk_cs; g.xk_ss = xk_ss; g.xpos_T = xT; g.xpos_dim = xdim;
  g.prec = km == 3 ? KX_PREC_BF16 : prec;   // bf16x3 runs the bf16 kernels over 3K
  g.tile = kx_tuning_get(KX_TUNE_GEMM_TILE);
  g.row_stats = row_stats; g.colsum = colsum; g.stats_out = stats_out;
  g.splitk_ws = g_splitk_ws; g.splitk_ws_bytes = g_splitk_ws_bytes; g.splitk = 0;
  // opt-in (tuning key 17 = 1): the in-launch reduction measured 0.4-4 us SLOWER per GEMM than the reduce launch it replaces on
  // every batch-1 shape (write-through partials + the arrival poll against a 1.5-1.9 us launch boundary), the batch-1 forward
  // 4.52 vs 3.85 ms (profiles/r06_d_coop_bench.log, r06_d_b1_ab.log)
  g.splitk_flags = kx_tuning_get(KX_TUNE_SPLITK_COOP) == 1 ? g_coop_counters : nullptr;
  g.pair_ws = g_pair_ws; g.pair_ws_bytes = g_pair_ws_bytes;
  if (rf) {
    g.stats_partials = rf->partials; g.stats_in_nseg = rf->nseg; g.stats_in_seg = rf->seg; g.stats_eps = rf->eps;
    g.row_stats_scratch = rf->scratch;
    g.ln_out = rf->ln_out; g.ln_out_dt = rf->ln_dt; g.ln_out_gamma = rf->ln_g; g.ln_out_beta = rf->ln_b; g.ln_out_eps = rf->eps;
  }
  if (lo) { g.ln_operand_out = lo->out; g.ln_operand_dt = lo->dt; g.ln_operand_stats = lo->stats; }
  return kx_gemm(&g, (void*)s);
}

// Will kx_gemm's automatic choice split this (M, N, K) problem, i.e. is the row-owning reduce (RowFusion) available?
// Same rule as kx_gemm itself; an explicit tile override (A/B runs) keeps the separate kernels.
bool row_reduce_available(int64_t M, int64_t N, int64_t K, int prec) {
  if (kx_tuning_get(KX_TUNE_GEMM_TILE) != 0 || N > 8192 || N % 4 != 0) return false;
  if (prec == KX_PREC_F16C) return kx_gemm_auto_splits(M, N, K, prec, g_splitk_ws_bytes) > 1;
  const int gp = prec == KX_PREC_BF16X3 ? KX_PREC_BF16 : prec;
  return kx_gemm_auto_splits(M, N, K * kmul(prec), gp, g_splitk_ws_bytes) > 1;
}

static int cu_count() {   // CUs of the current device (cached for device 0..63)
  static std::atomic<int> cus[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  int n = cus[dev].load(std::memory_order_relaxed);
  if (n == 0) {
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    cus[dev].store(n, std::memory_order_relaxed);
  }
  return n;
}

// tile 16 (weight streaming, bf16, M <= 16) with its prologues — see kx_gemm_args in the header
struct Gemv16 {
  const void* A; int64_t lda; const void* W; int64_t K; void* C; int64_t ldc; int cdt; int64_t M, N;
  const float* bias = nullptr; const float* residual = nullptr; int act = 0; float qscale = 1.f; int64_t qcols = 0;
  const float *xq_cs = nullptr, *xq_ss = nullptr, *xk_cs = nullptr, *xk_ss = nullptr; int64_t xT = 0, xdim = 0;
  const float *ln_g = nullptr, *ln_b = nullptr; float eps = 0.f;
  const float* partials_in = nullptr; int64_t nseg_in = 0, seg_in = 0; const float* colsum = nullptr;
  float* stats_out = nullptr;
  const void* W_tiled = nullptr;          // the same matrix in the streaming layout (kx_gemm_args.w_tiled), or null
  int tiled_fmt = 1;                      // kx_gemm_args.w_tiled of W_tiled: 1 = operand-dtype tiles, 2 = 24-bit planes
  int prec = KX_PREC_BF16;                // KX_PREC_BF16 or KX_PREC_F32 (operand dtype of A, unless ln_g, and of W)
  int ksplit = 0; void* C2 = nullptr; const float* residual2 = nullptr; const float* a_add = nullptr;   // kx_gemm_args: the pair form
};
int gemv16(const Gemv16& v, hipStream_t s) {
  kx_gemm_args g;
  memset(&g, 0, sizeof(g));
  g.A = v.A; g.lda = v.lda; g.W = v.W_tiled ? v.W_tiled : v.W; g.w_tiled = v.W_tiled ? v.tiled_fmt : 0; g.ldw = v.K;
  g.C = v.C; g.ldc = v.ldc; g.cdt = v.cdt;
  g.bias = v.bias; g.residual = v.residual; g.ldr = v.ldc; g.M = v.M; g.N = v.N; g.K = v.K;
  g.act = v.act; g.qscale = v.qscale; g.qcols = v.qcols;
  g.xq_cs = v.xq_cs; g.xq_ss = v.xq_ss; g.xk_cs = v.xk_cs; g.xk_ss = v.xk_ss; g.xpos_T = v.xT; g.xpos_dim = v.xdim;
  g.prec = v.prec; g.tile = 16;
  g.ln_gamma = v.ln_g; g.ln_beta = v.ln_b; g.ln_eps = v.eps;
  g.stats_partials = v.partials_in; g.stats_in_nseg = v.nseg_in; g.stats_in_seg = v.seg_in; g.stats_eps = v.eps;
  g.colsum = v.colsum;
  g.stats_out = v.stats_out; g.stats_out_seg = v.stats_out ? 16 : 0;
  g.ksplit = v.ksplit; g.C2 = v.C2; g.residual2 = v.residual2; g.a_add = v.a_add;
  return kx_gemm(&g, (void*)s);
}

int ln(const float* x, const float* pre, const float* g, const float* b, void* y, int ydt, int64_t rows, int64_t cols,
       float eps, hipStream_t s, int64_t rpg = 0, int64_t ogs = 0, int64_t oro = 0) {
  return kx_layernorm(x, pre, g, b, y, (kx_dtype)ydt, rows, cols, eps, rpg ? rpg : rows, ogs, oro, (void*)s);
}

// Binding guard (header, "Binding safety"): the caller's sizeof() of the weights struct and of its per-layer struct
#define KX_CHECK_BINDING(w, WT, LT, fn)                                                                                   \
  KX_REQUIRE((w)->struct_bytes == sizeof(WT) && (w)->layer_bytes == sizeof(LT),                                           \
             fn ": stale binding — caller declares " #WT " as %u bytes with %u-byte " #LT " elements, this library (ABI %d) " \
             "has %zu / %zu; regenerate the binding from include/kosmosx_hip.h",                                          \
             (unsigned)(w)->struct_bytes, (unsigned)(w)->layer_bytes, KX_ABI_VERSION, sizeof(WT), sizeof(LT))
template <class WT, class LT> static bool binding_ok(const WT* w) {
  return w && w->struct_bytes == sizeof(WT) && w->layer_bytes == sizeof(LT);
}

extern "C" size_t kx_struct_bytes(int32_t id) {
  switch (id) {
    case KX_STRUCT_GEMM_ARGS: return sizeof(kx_gemm_args);
    case KX_STRUCT_ATTN_ARGS: return sizeof(kx_attn_args);
    case KX_STRUCT_VIT_LAYER: return sizeof(kx_vit_layer);
    case KX_STRUCT_VIT_WEIGHTS: return sizeof(kx_vit_weights);
    case KX_STRUCT_PERCEIVER_LAYER: return sizeof(kx_perceiver_layer);
    case KX_STRUCT_PERCEIVER_WEIGHTS: return sizeof(kx_perceiver_weights);
    case KX_STRUCT_DECODER_LAYER: return sizeof(kx_decoder_layer);
    case KX_STRUCT_DECODER_WEIGHTS: return sizeof(kx_decoder_weights);
    case KX_STRUCT_RESAMPLE_PLAN: return sizeof(kx_resample_plan);
    case KX_STRUCT_PROF_RECORD: return sizeof(kx_prof_record);
    default: return 0;
  }
}

// ---------------- ViT ----------------
struct VitBufs { void *patches, *h, *qkv, *att, *ff, *splitk; float *patch_out, *xpre, *partials2, *stats2; size_t total; };
VitBufs vit_plan(const kx_vit_weights* w, int64_t B, int prec, char* base) {
  const int64_t G = w->image / w->patch, P = G * G, S = P + 1, M = B * S, MP = B * P;
  const size_t es = esz(prec);
  Carver c{base, 0};
  VitBufs v;
  v.patches = c.take((size_t)MP * w->kpad * es);
  v.patch_out = (float*)c.take((size_t)MP * w->dim * 4);
  v.xpre = (float*)c.take((size_t)M * w->dim * 4);
  v.h = c.take((size_t)M * w->dim * es);
  v.qkv = c.take((size_t)M * 3 * w->dim * qes(prec));
  v.att = c.take((size_t)M * w->dim * es);
  v.ff = c.take((size_t)M * w->ffn * es);
  v.partials2 = (float*)c.take((size_t)M * ((w->dim + 63) / 64) * 2 * 4);    // folded layer_norm1/2: per-64-column partials
  v.stats2 = (float*)c.take((size_t)M * 2 * 4);
  v.splitk = c.take(KX_SPLITK_WS);
  v.total = c.off;
  return v;
}

// ---------------- Perceiver ----------------
struct PerBufs { float* lat; void *kvin, *lnq, *qb, *kvb, *att, *ffh, *fin, *splitk; size_t total; };
PerBufs per_plan(const kx_perceiver_weights* w, int64_t B, int64_t m, int prec, char* base) {
  const int64_t n = w->latents, inner = (int64_t)w->heads * 64, MQ = B * n, MK = B * (m + n);
  const size_t es = esz(prec);
  Carver c{base, 0};
  PerBufs p;
  p.lat = (float*)c.take((size_t)MQ * w->dim * 4);
  p.kvin = c.take((size_t)MK * w->dim * es);
  p.lnq = c.take((size_t)MQ * w->dim * es);
  p.qb = c.take((size_t)MQ * inner * qes(prec));
  p.kvb = c.take((size_t)MK * 2 * inner * qes(prec));
  p.att = c.take((size_t)MQ * inner * es);
  p.ffh = c.take((size_t)MQ * w->dim * w->ff_mult * es);
  p.fin = c.take((size_t)MQ * w->dim * es);
  p.splitk = c.take(KX_SPLITK_WS);
  p.total = c.off;
  return p;
}

// ---------------- Decoder ----------------
struct DecBufs { void *h, *qkv, *att, *g, *splitk, *pair; float *partials, *stats, *partials2, *stats2, *xb, *ya, *yb; size_t total; };
DecBufs dec_plan(const kx_decoder_weights* w, int64_t B, int64_t T, int prec, char* base) {
  const int64_t M = B * T;
  const size_t es = esz(prec);
  Carver c{base, 0};
  DecBufs d;
  d.h = c.take((size_t)M * w->dim * es);
  d.qkv = c.take((size_t)M * 3 * w->dim * qes(prec));
  d.att = c.take((size_t)M * w->dim * es);          // attention output, un-normalised (inner_attn_ln is folded)
  d.g = c.take((size_t)M * w->ffn * es);            // gelu(fc1), un-normalised (ffn_layernorm is folded)
  // per-segment statistics: 64 columns per segment from the tile kernels and attention (heads), 16 from the
  // weight-streaming decode path
  const int64_t nseg = w->ffn / 16 > w->heads ? w->ffn / 16 : w->heads;
  d.partials = (float*)c.take((size_t)M * nseg * 2 * 4);
  d.stats = (float*)c.take((size_t)M * 2 * 4);
  d.partials2 = (float*)c.take((size_t)M * ((w->dim + 63) / 64) * 2 * 4);   // folded pre-LayerNorms (residual-stream rows)
  d.stats2 = (float*)c.take((size_t)M * 2 * 4);
  // decode step (T = 1): the residual stream as a pair — second addend of x, and the pair the attention block writes
  d.xb = d.ya = d.yb = nullptr;
  if (T == 1) { d.xb = (float*)c.take((size_t)M * w->dim * 4); d.ya = (float*)c.take((size_t)M * w->dim * 4); d.yb = (float*)c.take((size_t)M * w->dim * 4); }
  d.splitk = c.take(KX_SPLITK_WS);
  d.pair = T > 1 ? c.take(KX_PAIR_WS) : nullptr;    // the tile kernels' pair split (out_proj / fc2 at half a round of 256x256 tiles)
  d.total = c.off;
  return d;
}

}  // namespace

extern "C" size_t kx_vit_workspace_bytes(const kx_vit_weights* w, int64_t B, int32_t prec) {
  if (!binding_ok<kx_vit_weights, kx_vit_layer>(w)) { kx_set_error("kx_vit_workspace_bytes: null or stale binding (struct_bytes / layer_bytes)"); return 0; }
  return vit_plan(w, B, prec, nullptr).total;
}

extern "C" int kx_vit_forward(const kx_vit_weights* w, const float* pixels, int64_t B, float* out, void* workspace,
                              size_t workspace_bytes, int32_t prec, void* stream) {
  KX_REQUIRE(w && pixels && out && workspace, "kx_vit_forward: null pointer");
  KX_CHECK_BINDING(w, kx_vit_weights, kx_vit_layer, "kx_vit_forward");
  KX_REQUIRE(prec >= KX_PREC_BF16 && prec <= KX_PREC_F16, "kx_vit_forward: bad precision %d (KX_PREC_F32W24 / W16 are decode-step formats)", prec);
  KX_REQUIRE(B > 0, "kx_vit_forward: empty batch");
  KX_REQUIRE(w->dim == w->heads * 64, "kx_vit_forward: head_dim must be 64 (dim=%d heads=%d)", w->dim, w->heads);
  KX_REQUIRE(w->image % w->patch == 0 && w->kpad >= 3 * w->patch * w->patch && w->kpad % 64 == 0,
             "kx_vit_forward: bad patch geometry");
  KX_REQUIRE(prec != KX_PREC_F16C || w->kpad % 128 == 0, "kx_vit_forward: KX_PREC_F16C needs kpad %% 128 == 0");
  KX_REQUIRE(((uintptr_t)workspace & 255) == 0, "kx_vit_forward: workspace must be 256-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  const VitBufs v = vit_plan(w, B, prec, (char*)workspace);
  if (v.total > workspace_bytes) {
    kx_set_error("kx_vit_forward: workspace %zu < required %zu", workspace_bytes, v.total);
    return KX_ERR_WORKSPACE;
  }
  const int64_t G = w->image / w->patch, P = G * G, S = P + 1, M = B * S, MP = B * P, D = w->dim;
  KX_SPLITK_SCOPE(v.splitk, s);
  const int ct = cdt(prec);
  KX_TRY(kx_launch_patchify(pixels, v.patches, B, w->image, w->patch, w->kpad, prec, s));
  KX_TRY(gemm(v.patches, w->kpad, w->wpatch, w->kpad, v.patch_out, D, KX_F32, MP, D, nullptr, nullptr, 0, 1.f, 0,
              prec, s));
  KX_TRY(kx_launch_vit_assemble(v.patch_out, w->cls, w->pos, v.xpre, B, (int)S, (int)D, s));
  KX_TRY(ln(v.xpre, nullptr, w->pre_g, w->pre_b, out, KX_F32, M, D, w->eps, s));
  // At batch 1 (M = 257) the GEMMs are split-K and the layer is a chain of dependent ~12 us launches: the residual
  // GEMMs' row-owning reduce kernels also write the LayerNorm that follows (layer_norm2 / the next layer's layer_norm1).
  const bool fuse_o = row_reduce_available(M, D, D, prec), fuse_2 = row_reduce_available(M, D, w->ffn, prec);
  // Folded layer_norm1 / layer_norm2 (large batches, the tile kernels): the residual GEMMs write the operand rows and the
  // row statistics of what they finish, qkv / fc1 multiply by gamma-folded weights — no LayerNorm kernel between layers.
  const bool fold = w->layers > 0 && w->layer[0].wqkv_f && fold_prec(prec) && !fuse_o && !fuse_2 && D % 64 == 0 &&
                    kx_tuning_get(KX_TUNE_GEMM_TILE) == 0 && kx_tuning_get(KX_TUNE_GEMM_EPILOGUE) != 1;
  const LnOp lop{v.h, ct, v.partials2};
  bool h_ready = false;                                   // v.h already holds layer_norm1(out) of this layer
  for (int i = 0; i < w->layers; ++i) {
    const kx_vit_layer& L = w->layer[i];
    const bool st1 = fold && i > 0;                        // v.h = un-normalised rows + v.stats2 from the previous fc2
    if (fold && i == 0) KX_TRY(ln(out, nullptr, nullptr, nullptr, v.h, ct, M, D, w->eps, s));   // unit affine: the fold carries gamma / beta
    else if (!fold && !h_ready) KX_TRY(ln(out, nullptr, L.ln1_g, L.ln1_b, v.h, ct, M, D, w->eps, s));
    KX_TRY(gemm(v.h, D, fold ? L.wqkv_f : L.wqkv, D, v.qkv, 3 * D, qdt(prec), M, 3 * D, fold ? L.bqkv_f : L.bqkv, nullptr, 0,
                0.125f, D, prec, s, nullptr, nullptr, nullptr, nullptr, 0, 0, st1 ? v.stats2 : nullptr,
                st1 ? L.wqkv_colsum : nullptr));
    kx_attn_args a;
    memset(&a, 0, sizeof(a));
    a.q = v.qkv; a.q_batch_stride = S * 3 * D; a.q_row_stride = 3 * D;
    a.k = (char*)v.qkv + D * qes(prec); a.v = (char*)v.qkv + 2 * D * qes(prec);
    a.kv_batch_stride = S * 3 * D; a.kv_row_stride = 3 * D;
    a.out = v.att; a.out_batch_stride = S * D * kmul(prec); a.out_row_stride = D * kmul(prec); a.odt = ct;
    a.B = B; a.H = w->heads; a.Tq = S; a.Tk = S; a.mask = KX_ATTN_FULL; a.prec = aprec(prec);
    KX_TRY(kx_attention(&a, stream));
    RowFusion ro, r2;
    ro.ln_out = v.h; ro.ln_dt = ct; ro.ln_g = L.ln2_g; ro.ln_b = L.ln2_b; ro.eps = w->eps;
    KX_TRY(gemm(v.att, D, L.wo, D, out, D, KX_F32, M, D, L.bo, out, 0, 1.f, 0, prec, s, nullptr, nullptr, nullptr, nullptr, 0,
                0, nullptr, nullptr, nullptr, fuse_o ? &ro : nullptr, fold ? &lop : nullptr));
    if (fold) KX_TRY(kx_row_stats_finalize(v.partials2, M, D / 64, 64, w->eps, v.stats2, stream));
    else if (!fuse_o) KX_TRY(ln(out, nullptr, L.ln2_g, L.ln2_b, v.h, ct, M, D, w->eps, s));
    KX_TRY(gemm(v.h, D, fold ? L.w1_f : L.w1, D, v.ff, w->ffn, ct, M, w->ffn, fold ? L.b1_f : L.b1, nullptr, w->act, 1.f, 0,
                prec, s, nullptr, nullptr, nullptr, nullptr, 0, 0, fold ? v.stats2 : nullptr, fold ? L.w1_colsum : nullptr));
    const bool next = fuse_2 && i + 1 < w->layers;
    if (next) { r2.ln_out = v.h; r2.ln_dt = ct; r2.ln_g = w->layer[i + 1].ln1_g; r2.ln_b = w->layer[i + 1].ln1_b; r2.eps = w->eps; }
    const bool emit = fold && i + 1 < w->layers;           // the last layer's output is the tower's output: nothing follows
    KX_TRY(gemm(v.ff, w->ffn, L.w2, w->ffn, out, D, KX_F32, M, D, L.b2, out, 0, 1.f, 0, prec, s, nullptr, nullptr, nullptr,
                nullptr, 0, 0, nullptr, nullptr, nullptr, next ? &r2 : nullptr, emit ? &lop : nullptr));
    if (emit) KX_TRY(kx_row_stats_finalize(v.partials2, M, D / 64, 64, w->eps, v.stats2, stream));
    h_ready = next;
  }
  return KX_OK;
}

extern "C" size_t kx_perceiver_workspace_bytes(const kx_perceiver_weights* w, int64_t B, int64_t m, int32_t prec) {
  if (!binding_ok<kx_perceiver_weights, kx_perceiver_layer>(w)) { kx_set_error("kx_perceiver_workspace_bytes: null or stale binding (struct_bytes / layer_bytes)"); return 0; }
  return per_plan(w, B, m, prec, nullptr).total;
}

extern "C" int kx_perceiver_forward(const kx_perceiver_weights* w, const float* x, int64_t B, int64_t m, float* out,
                                    float* lat_out, void* workspace, size_t workspace_bytes, int32_t prec,
                                    void* stream) {
  KX_REQUIRE(w && x && workspace, "kx_perceiver_forward: null pointer");
  KX_CHECK_BINDING(w, kx_perceiver_weights, kx_perceiver_layer, "kx_perceiver_forward");
  KX_REQUIRE(prec >= KX_PREC_BF16 && prec <= KX_PREC_F16, "kx_perceiver_forward: bad precision %d (KX_PREC_F32W24 / W16 are decode-step formats)", prec);
  KX_REQUIRE((out && w->wproj && w->out_dim > 0) || lat_out, "kx_perceiver_forward: nothing to produce");
  KX_REQUIRE(B > 0 && m > 0, "kx_perceiver_forward: empty input");
  KX_REQUIRE(((uintptr_t)workspace & 255) == 0, "kx_perceiver_forward: workspace must be 256-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  const PerBufs p = per_plan(w, B, m, prec, (char*)workspace);
  if (p.total > workspace_bytes) {
    kx_set_error("kx_perceiver_forward: workspace %zu < required %zu", workspace_bytes, p.total);
    return KX_ERR_WORKSPACE;
  }
  const int64_t n = w->latents, D = w->dim, inner = (int64_t)w->heads * 64, MQ = B * n, MK = B * (m + n);
  KX_SPLITK_SCOPE(p.splitk, s);
  const int64_t F = D * w->ff_mult;
  const int ct = cdt(prec);
  KX_TRY(kx_launch_rows_bcast(w->latents_p, p.lat, B, n, D, s));
  for (int i = 0; i < w->depth; ++i) {
    const kx_perceiver_layer& L = w->layer[i];
    // kv_input = cat(norm_media(x + media_pos_emb[:1]), norm_latents(latents)) along the token axis
    KX_TRY(ln(x, w->media_pos, L.nm_g, L.nm_b, p.kvin, ct, B * m, D, w->eps, s, m, m + n, 0));
    KX_TRY(ln(p.lat, nullptr, L.nl_g, L.nl_b, p.kvin, ct, MQ, D, w->eps, s, n, m + n, m));
    KX_TRY(ln(p.lat, nullptr, L.nl_g, L.nl_b, p.lnq, ct, MQ, D, w->eps, s));
    KX_FAM(5);
    KX_TRY(gemm(p.lnq, D, L.wq, D, p.qb, inner, qdt(prec), MQ, inner, nullptr, nullptr, 0, 0.125f, inner, prec, s));
    KX_FAM(5);
    KX_TRY(gemm(p.kvin, D, L.wkv, D, p.kvb, 2 * inner, qdt(prec), MK, 2 * inner, nullptr, nullptr, 0, 1.f, 0, prec, s));
    kx_attn_args a;
    memset(&a, 0, sizeof(a));
    a.q = p.qb; a.q_batch_stride = n * inner; a.q_row_stride = inner;
    a.k = p.kvb; a.v = (char*)p.kvb + inner * qes(prec);
    a.kv_batch_stride = (m + n) * 2 * inner; a.kv_row_stride = 2 * inner;
    a.out = p.att; a.out_batch_stride = n * inner * kmul(prec); a.out_row_stride = inner * kmul(prec); a.odt = ct;
    a.B = B; a.H = w->heads; a.Tq = n; a.Tk = m + n; a.mask = KX_ATTN_FULL; a.prec = aprec(prec);
    KX_TRY(kx_attention(&a, stream));
    KX_FAM(5);
    KX_TRY(gemm(p.att, inner, L.wout, inner, p.lat, D, KX_F32, MQ, D, nullptr, p.lat, 0, 1.f, 0, prec, s));
    KX_TRY(ln(p.lat, nullptr, L.ff_g, L.ff_b, p.lnq, ct, MQ, D, w->eps, s));
    KX_FAM(5);
    KX_TRY(gemm(p.lnq, D, L.w1, D, p.ffh, F, ct, MQ, F, nullptr, nullptr, KX_ACT_GELU, 1.f, 0, prec, s));
    KX_FAM(5);
    KX_TRY(gemm(p.ffh, F, L.w2, F, p.lat, D, KX_F32, MQ, D, nullptr, p.lat, 0, 1.f, 0, prec, s));
  }
  if (lat_out) KX_TRY(ln(p.lat, nullptr, w->norm_g, w->norm_b, lat_out, KX_F32, MQ, D, w->eps, s));
  if (out && w->wproj) {
    KX_TRY(ln(p.lat, nullptr, w->norm_g, w->norm_b, p.fin, ct, MQ, D, w->eps, s));
    KX_FAM(5);
    KX_TRY(gemm(p.fin, D, w->wproj, D, out, w->out_dim, KX_F32, MQ, w->out_dim, nullptr, nullptr, 0, 1.f, 0, prec,
                s));
  }
  return KX_OK;
}

extern "C" size_t kx_decoder_workspace_bytes(const kx_decoder_weights* w, int64_t B, int64_t T, int32_t prec) {
  if (!binding_ok<kx_decoder_weights, kx_decoder_layer>(w)) { kx_set_error("kx_decoder_workspace_bytes: null or stale binding (struct_bytes / layer_bytes)"); return 0; }
  return dec_plan(w, B, T, (prec == KX_PREC_F32W24 || prec == KX_PREC_F32W16) ? KX_PREC_F32 : prec, nullptr).total;
}

static int decoder_forward_impl(const kx_decoder_weights* w, float* x, int64_t B, int64_t T, const float* xq_cs,
                                const float* xq_ss, const float* xk_cs, const float* xk_ss, void* logits, int32_t ldt,
                                void* workspace, size_t workspace_bytes, int32_t prec, void* stream, void* kcache,
                                void* vcache, int64_t Tmax) {
  KX_REQUIRE(w && x && logits && workspace, "kx_decoder_forward: null pointer");
  KX_CHECK_BINDING(w, kx_decoder_weights, kx_decoder_layer, "kx_decoder_forward");
  KX_REQUIRE(prec >= KX_PREC_BF16 && prec <= KX_PREC_F16, "kx_decoder_forward: bad precision %d (KX_PREC_F32W24 / W16 are decode-step formats)", prec);
  KX_REQUIRE(!kcache == !vcache, "kx_decoder_prefill: kcache and vcache must be given together");
  KX_REQUIRE(!kcache || (prec != KX_PREC_BF16X3 && prec != KX_PREC_F16),
             "kx_decoder_prefill: incremental decoding is offered in bf16, fp32 and f16c (fp32 cache)");
  KX_REQUIRE(!kcache || T <= Tmax, "kx_decoder_prefill: %lld tokens do not fit a %lld-row cache", (long long)T,
             (long long)Tmax);
  KX_REQUIRE(B > 0 && T > 0, "kx_decoder_forward: empty input");
  KX_REQUIRE(w->dim == w->heads * 64, "kx_decoder_forward: head_dim must be 64 (dim=%d heads=%d)", w->dim, w->heads);
  KX_REQUIRE(!w->xpos || (xq_cs && xq_ss && xk_cs && xk_ss), "kx_decoder_forward: XPos tables missing");
  KX_REQUIRE(!w->subln || w->ffn % 64 == 0, "kx_decoder_forward: ffn must be a multiple of 64 for the folded sub-LN");
  KX_REQUIRE(((uintptr_t)workspace & 255) == 0, "kx_decoder_forward: workspace must be 256-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  const DecBufs d = dec_plan(w, B, T, prec, (char*)workspace);
  if (d.total > workspace_bytes) {
    kx_set_error("kx_decoder_forward: workspace %zu < required %zu", workspace_bytes, d.total);
    return KX_ERR_WORKSPACE;
  }
  const int64_t M = B * T, D = w->dim, F = w->ffn;
  KX_SPLITK_SCOPE(d.splitk, s);
  PairScope pk(d.pair, KX_PAIR_WS);
  if (d.pair && hipMemsetAsync(d.pair, 0, 4096, s) != hipSuccess) {      // hand-off words: zero when a call is issued
    kx_set_error("kx_decoder_forward: clearing the pair-split hand-off words failed");
    return KX_ERR_LAUNCH;
  }
  const int ct = cdt(prec);
  const size_t es = esz(prec);
  // Batch-1-sized problems (M = T = 114: every GEMM is split-K): the residual GEMMs' row-owning reduce kernels take the
  // folded-LN statistics straight from the producer's partials and write the LayerNorm that follows (final_layer_norm,
  // the next layer's self_attn_layer_norm, the decoder's last LayerNorm) — 9 launches per layer instead of 13.
  const bool fuse_o = row_reduce_available(M, D, D, prec), fuse_2 = row_reduce_available(M, D, F, prec);
  // Folded self_attn_layer_norm / final_layer_norm / decoder.layer_norm (large M, the tile kernels): out_proj and fc2 —
  // the GEMMs that finish a row of the residual stream — also write it as the next GEMM's operand (d.h) with its partial
  // statistics; qkv / fc1 / the output projection multiply by gamma-folded weights and apply rstd*(acc - mean*colsum) +
  // (W beta + b).  One LayerNorm launch is left (layer 0, unit affine) instead of 2 L + 1.
  const bool fold = w->layers > 0 && w->layer[0].wqkv_f && w->wout_f && fold_prec(prec) && !fuse_o && !fuse_2 && D % 64 == 0 &&
                    kx_tuning_get(KX_TUNE_GEMM_TILE) == 0 && kx_tuning_get(KX_TUNE_GEMM_EPILOGUE) != 1;
  const LnOp lop{d.h, ct, d.partials2};
  // KX_PREC_F16C at tile-kernel sizes: the qkv GEMM writes the attention kernel's operand pieces itself (KX_F16HL rows, same
  // bytes per value as fp32 q / k / v) instead of every attention workgroup re-deriving them per key tile (tuning key 15 & 4: off).
  // Not with a KV cache (the cache and the decode steps read fp32 q / k / v) and not without XPos (the piece store lives in the
  // prefetching store loop / the lean XPos epilogue).
  // Only where keys are re-used: a key tile is loaded by every 128-query block after it, so at T = 2046 each piece was derived
  // ~8 times and at T = 114 once — there the split just moves from the attention kernel into the GEMM epilogue (measured:
  // qkv 192 -> 211 us against attention 42 -> ~35 at B = 32 x 114; tools/hl_probe.py).
  const bool qkv_hl = prec == KX_PREC_F16C && w->xpos && !kcache && M >= 1024 && T >= 512 && D % 64 == 0 &&
                      !(kx_tuning_get(KX_TUNE_GEMM_RULES) & 4) && kx_tuning_get(KX_TUNE_GEMM_EPILOGUE) != 1 &&
                      kx_tuning_get(KX_TUNE_GEMM_TILE) == 0 && kx_tuning_get(KX_TUNE_ATTN_VARIANT) == 0;
  bool h_ready = false;                                   // d.h already holds the LayerNorm this layer starts with
  for (int i = 0; i < w->layers; ++i) {
    const kx_decoder_layer& L = w->layer[i];
    // x = x + out_proj(inner_attn_ln(attn(xpos(q), xpos(k), v)))   on self_attn_layer_norm(x)
    const bool st1 = fold && i > 0;                        // d.h = un-normalised rows of x + d.stats2 from the previous fc2
    if (fold && i == 0) KX_TRY(ln(x, nullptr, nullptr, nullptr, d.h, ct, M, D, w->eps, s));     // unit affine
    else if (!fold && !h_ready) KX_TRY(ln(x, nullptr, L.sa_g, L.sa_b, d.h, ct, M, D, w->eps, s));
    KX_FAM(0);
    KX_TRY(gemm(d.h, D, fold ? L.wqkv_f : L.wqkv, D, d.qkv, 3 * D, (qkv_hl && !st1) ? KX_F16HL : qdt(prec), M, 3 * D,
                fold ? L.bqkv_f : L.bqkv, nullptr, 0,
                0.125f, D, prec, s, w->xpos ? xq_cs : nullptr, xq_ss, xk_cs, xk_ss, w->xpos ? T : 0, w->xpos ? D : 0,
                st1 ? d.stats2 : nullptr, st1 ? L.wqkv_colsum : nullptr));
    if (kcache) {   // incremental decoding: keep this layer's (XPos-rotated) keys and values
      const size_t layer_bytes = (size_t)B * Tmax * D * qes(prec);
      KX_TRY(kx_launch_kv_prefill(d.qkv, (char*)kcache + i * layer_bytes, (char*)vcache + i * layer_bytes, B, T, D, Tmax,
                                  prec, s));
    }
    kx_attn_args a;
    memset(&a, 0, sizeof(a));
    a.q = d.qkv; a.q_batch_stride = T * 3 * D; a.q_row_stride = 3 * D;
    a.k = (char*)d.qkv + D * qes(prec); a.v = (char*)d.qkv + 2 * D * qes(prec);
    a.kv_batch_stride = T * 3 * D; a.kv_row_stride = 3 * D;
    a.B = B; a.H = w->heads; a.Tq = T; a.Tk = T; a.mask = KX_ATTN_CAUSAL; a.prec = (qkv_hl && !st1) ? KX_PREC_F16CHL : aprec(prec);
    a.out = d.att; a.out_batch_stride = T * D * kmul(prec); a.out_row_stride = D * kmul(prec); a.odt = ct;
    RowFusion ro, r2;
    ro.ln_out = d.h; ro.ln_dt = ct; ro.ln_g = L.fl_g; ro.ln_b = L.fl_b; ro.eps = w->eps;
    if (w->subln) {
      // inner_attn_ln folded into out_proj: the attention kernel emits per-(row, head) partial statistics, the GEMM
      // multiplies the un-normalised output by γ⊙Wo and applies rstd·(acc − mean·colsum) + (β·Woᵀ + bo) in its epilogue
      a.stats_out = d.partials;
      KX_TRY(kx_attention(&a, stream));
      if (fuse_o) {
        ro.partials = d.partials; ro.nseg = w->heads; ro.seg = 64;
        KX_FAM(1);
        KX_TRY(gemm(d.att, D, L.wo, D, x, D, KX_F32, M, D, L.bo, x, 0, 1.f, 0, prec, s, nullptr, nullptr, nullptr, nullptr,
                    0, 0, nullptr, L.wo_colsum, nullptr, &ro));
      } else {
        // (row_stats_scratch + stats_partials: kx_gemm finalises the partials inside the pair-split launch where it takes that, else
        //  runs kx_row_stats_finalize into d.stats itself)
        RowFusion fo; fo.partials = d.partials; fo.nseg = w->heads; fo.seg = 64; fo.eps = w->eps; fo.scratch = d.stats;
        KX_FAM(1);
        KX_TRY(gemm(d.att, D, L.wo, D, x, D, KX_F32, M, D, L.bo, x, 0, 1.f, 0, prec, s, nullptr, nullptr, nullptr, nullptr,
                    0, 0, nullptr, L.wo_colsum, nullptr, &fo, fold ? &lop : nullptr));
      }
    } else {
      KX_TRY(kx_attention(&a, stream));
      KX_FAM(1);
      KX_TRY(gemm(d.att, D, L.wo, D, x, D, KX_F32, M, D, L.bo, x, 0, 1.f, 0, prec, s, nullptr, nullptr, nullptr, nullptr, 0,
                  0, nullptr, nullptr, nullptr, fuse_o ? &ro : nullptr, fold ? &lop : nullptr));
    }
    // x = x + fc2(ffn_layernorm(gelu(fc1(final_layer_norm(x)))))
    if (fold) KX_TRY(kx_row_stats_finalize(d.partials2, M, D / 64, 64, w->eps, d.stats2, stream));
    else if (!fuse_o) KX_TRY(ln(x, nullptr, L.fl_g, L.fl_b, d.h, ct, M, D, w->eps, s));
    const bool last = i + 1 == w->layers;
    r2.ln_out = d.h; r2.ln_dt = ct; r2.eps = w->eps;
    r2.ln_g = last ? w->ln_g : w->layer[i + 1].sa_g; r2.ln_b = last ? w->ln_b : w->layer[i + 1].sa_b;
    const void* w1 = fold ? L.w1_f : L.w1;
    const float* b1 = fold ? L.b1_f : L.b1;
    const float* rs1 = fold ? d.stats2 : nullptr;
    const float* cs1 = fold ? L.w1_colsum : nullptr;
    if (w->subln) {
      // ffn_layernorm folded into fc2 the same way; fc1's epilogue emits the row statistics of gelu(fc1)
      KX_FAM(2);
      KX_TRY(gemm(d.h, D, w1, D, d.g, F, ct, M, F, b1, nullptr, w->act, 1.f, 0, prec, s, nullptr, nullptr, nullptr,
                  nullptr, 0, 0, rs1, cs1, d.partials));
      if (fuse_2) {
        r2.partials = d.partials; r2.nseg = F / 64; r2.seg = 64;
        KX_FAM(3);
        KX_TRY(gemm(d.g, F, L.w2, F, x, D, KX_F32, M, D, L.b2, x, 0, 1.f, 0, prec, s, nullptr, nullptr, nullptr, nullptr,
                    0, 0, nullptr, L.w2_colsum, nullptr, &r2));
      } else {
        RowFusion f2; f2.partials = d.partials; f2.nseg = F / 64; f2.seg = 64; f2.eps = w->eps; f2.scratch = d.stats;
        KX_FAM(3);
        KX_TRY(gemm(d.g, F, L.w2, F, x, D, KX_F32, M, D, L.b2, x, 0, 1.f, 0, prec, s, nullptr, nullptr, nullptr, nullptr,
                    0, 0, nullptr, L.w2_colsum, nullptr, &f2, fold ? &lop : nullptr));
      }
    } else {
      KX_FAM(2);
      KX_TRY(gemm(d.h, D, w1, D, d.g, F, ct, M, F, b1, nullptr, w->act, 1.f, 0, prec, s, nullptr, nullptr, nullptr, nullptr,
                  0, 0, rs1, cs1));
      KX_FAM(3);
      KX_TRY(gemm(d.g, F, L.w2, F, x, D, KX_F32, M, D, L.b2, x, 0, 1.f, 0, prec, s, nullptr, nullptr, nullptr, nullptr, 0, 0,
                  nullptr, nullptr, nullptr, fuse_2 ? &r2 : nullptr, fold ? &lop : nullptr));
    }
    if (fold) KX_TRY(kx_row_stats_finalize(d.partials2, M, D / 64, 64, w->eps, d.stats2, stream));
    h_ready = fuse_2;                                     // d.h = the next layer's (or the final) LayerNorm of x
  }
  if (fold) {                                             // decoder.layer_norm folded into the output projection
    KX_FAM(4);
    KX_TRY(gemm(d.h, D, w->wout_f, D, logits, w->vocab, ldt, M, w->vocab, w->bout_f, nullptr, 0, 1.f, 0, prec, s, nullptr,
                nullptr, nullptr, nullptr, 0, 0, d.stats2, w->wout_colsum));
    return KX_OK;
  }
  if (!h_ready) KX_TRY(ln(x, nullptr, w->ln_g, w->ln_b, d.h, ct, M, D, w->eps, s));
  KX_FAM(4);
  KX_TRY(gemm(d.h, D, w->wout, D, logits, w->vocab, ldt, M, w->vocab, nullptr, nullptr, 0, 1.f, 0, prec, s));
  return KX_OK;
}

extern "C" int kx_decoder_forward(const kx_decoder_weights* w, float* x, int64_t B, int64_t T, const float* xq_cs,
                                  const float* xq_ss, const float* xk_cs, const float* xk_ss, void* logits,
                                  int32_t ldt, void* workspace, size_t workspace_bytes, int32_t prec, void* stream) {
  return decoder_forward_impl(w, x, B, T, xq_cs, xq_ss, xk_cs, xk_ss, logits, ldt, workspace, workspace_bytes, prec,
                              stream, nullptr, nullptr, 0);
}

extern "C" int kx_decoder_prefill(const kx_decoder_weights* w, float* x, int64_t B, int64_t T, const float* xq_cs,
                                  const float* xq_ss, const float* xk_cs, const float* xk_ss, void* logits,
                                  int32_t ldt, void* kcache, void* vcache, int64_t Tmax, void* workspace,
                                  size_t workspace_bytes, int32_t prec, void* stream) {
  KX_REQUIRE(kcache && vcache && Tmax > 0, "kx_decoder_prefill: cache missing");
  return decoder_forward_impl(w, x, B, T, xq_cs, xq_ss, xk_cs, xk_ss, logits, ldt, workspace, workspace_bytes, prec,
                              stream, kcache, vcache, Tmax);
}

extern "C" int kx_decoder_decode_step(const kx_decoder_weights* w, float* x, int64_t B, int64_t t, const float* xq_cs,
                                      const float* xq_ss, const float* xk_cs, const float* xk_ss, void* kcache,
                                      void* vcache, int64_t Tmax, void* logits, int32_t ldt, void* workspace,
                                      size_t workspace_bytes, int32_t prec, void* stream) {
  const int tfmt = prec == KX_PREC_F32W24 ? 2 : prec == KX_PREC_F32W16 ? 3 : 1;   // what the streaming copies (w*_t) hold
  if (prec == KX_PREC_F32W24 || prec == KX_PREC_F32W16) prec = KX_PREC_F32;
  KX_REQUIRE(w && x && logits && workspace && kcache && vcache, "kx_decoder_decode_step: null pointer");
  KX_CHECK_BINDING(w, kx_decoder_weights, kx_decoder_layer, "kx_decoder_decode_step");
  KX_REQUIRE(prec != KX_PREC_BF16X3 && prec != KX_PREC_F16,
             "kx_decoder_decode_step: incremental decoding is offered in bf16, fp32 and f16c (fp32 cache)");
  KX_REQUIRE(B > 0 && t >= 0 && t < Tmax, "kx_decoder_decode_step: position %lld outside the cache of %lld rows",
             (long long)t, (long long)Tmax);
  KX_REQUIRE(w->dim == w->heads * 64, "kx_decoder_decode_step: head_dim must be 64");
  KX_REQUIRE(!w->xpos || (xq_cs && xq_ss && xk_cs && xk_ss), "kx_decoder_decode_step: XPos rows for position t missing");
  KX_REQUIRE(((uintptr_t)workspace & 255) == 0, "kx_decoder_decode_step: workspace must be 256-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  const DecBufs d = dec_plan(w, B, 1, prec, (char*)workspace);
  if (d.total > workspace_bytes) {
    kx_set_error("kx_decoder_decode_step: workspace %zu < required %zu", workspace_bytes, d.total);
    return KX_ERR_WORKSPACE;
  }
  const int64_t M = B, D = w->dim, F = w->ffn;
  KX_SPLITK_SCOPE(d.splitk, s);
  const int ct = cdt(prec);
  const size_t es = esz(prec);
  const size_t layer_bytes = (size_t)B * Tmax * D * qes(prec);   // the cache holds q/k/v-typed values (fp32 for f16c)
  // One token per sequence, up to 16 sequences: the step is 120 dependent launches of weight-streaming work, and launches
  // are what it costs (~6 us each) — tile 16 does each GEMM in one launch and takes the LayerNorm and
  // statistics-finalize kernels in as prologues: 5 launches per layer instead of 13.  bf16 operands, or fp32 operands on the
  // exact-f32 MFMA (KX_PREC_F32: what the Python side asks for in every precision that holds the north star's tolerance).
  if ((prec == KX_PREC_BF16 || prec == KX_PREC_F32) && M <= 16 && kx_tuning_get(KX_TUNE_GEMM_TILE) == 0 && D % 32 == 0 &&
      F % 32 == 0 && (size_t)M * (es * D + 16) <= 144 * 1024 && !(prec == KX_PREC_F32 && kx_tuning_get(KX_TUNE_DECODE_STREAM_F32) == 1)) {
    // The residual GEMMs have N = D columns = D / 16 workgroups (128 at full size: half the CUs stream).  Where that leaves
    // CUs idle they run as TWO workgroups per column block, each over half of K, and the residual stream becomes a pair
    // x = xa + xb (kx_gemm_args.ksplit): part 0 writes (xa' + xb') + bias + its product to the next pair's first member,
    // part 1 its product to the second; the LayerNorm prologues and the next residual read sum the pair in that order.
    // No cross-workgroup reduction, deterministic.  Tuning key 11 = 1 keeps one workgroup per block (in place: x += ...).
    const bool pair = kx_tuning_get(KX_TUNE_DECODE_KSPLIT) != 1 && kx_tuning_get(KX_TUNE_GEMV_VARIANT) != 1 &&
                      D / 16 <= (cu_count() * 3) / 4 && D % 128 == 0 && F % 128 == 0 &&
                      M <= 4;   // (five rows and more take the wave-per-row LayerNorm prologue, which walks each row three times: reading a
                                //  pair there cost more than the idle CUs — B = 8: 1.43 -> 1.55 ms per step in bf16)
    // Block-scaled 16-bit planes, three rows and more: the residual GEMMs multiply fp16 pieces (kx_gemm_args.w_tiled = 3), and what
    // they read — the attention output, gelu(fc1) — is written AS pieces by its producer (KX_F16P, w_tiled = 4): the same bits the
    // consumer would make of the fp32 rows, made once instead of in every lane of every workgroup.
    // the compressed plane formats ARE the weights of such a step: a layer without its streaming copies would make gemv16 fall
    // back to the row-major operand while its producers already wrote KX_F16P rows (garbage logits, no error)
    if (tfmt >= 2) {
      KX_REQUIRE(w->wout_t, "kx_decoder_decode_step: KX_PREC_F32W24 / KX_PREC_F32W16 need the streaming copy wout_t");
      for (int i = 0; i < w->layers; ++i)
        KX_REQUIRE(w->layer[i].wqkv_t && w->layer[i].wo_t && w->layer[i].w1_t && w->layer[i].w2_t,
                   "kx_decoder_decode_step: KX_PREC_F32W24 / KX_PREC_F32W16 need wqkv_t / wo_t / w1_t / w2_t of layer %d", i);
    }
    const int gv = kx_tuning_get(KX_TUNE_GEMV_VARIANT);
    const bool pieces = tfmt == 3 && M >= 3 && gv != 1 && gv != 4 && gv != 5 && gv < 10 && kx_tuning_get(KX_TUNE_DECODE_PIECES) != 1;
    float *pa = x, *pb = nullptr;                                 // the stream as the next reader finds it
    for (int i = 0; i < w->layers; ++i) {
      const kx_decoder_layer& L = w->layer[i];
      Gemv16 q{pa, D, L.wqkv, D, d.qkv, 3 * D, ct, M, 3 * D};
      q.bias = L.bqkv; q.qscale = 0.125f; q.qcols = D; q.ln_g = L.sa_g; q.ln_b = L.sa_b; q.eps = w->eps; q.a_add = pb;
      if (w->xpos) { q.xq_cs = xq_cs; q.xq_ss = xq_ss; q.xk_cs = xk_cs; q.xk_ss = xk_ss; q.xT = 1; q.xdim = D; }
      q.W_tiled = L.wqkv_t; q.tiled_fmt = tfmt; q.prec = prec;
      KX_TRY(gemv16(q, s));
      KX_TRY(kx_attention_decode(d.qkv, (char*)kcache + i * layer_bytes, (char*)vcache + i * layer_bytes, d.att, pieces ? KX_F16P : ct,
                                 w->subln ? d.partials : nullptr, B, w->heads, t, Tmax, prec, stream));
      float *qa = pair ? d.ya : pa, *qb = pair ? d.yb : nullptr;   // what out_proj writes
      Gemv16 o{d.att, D, L.wo, D, qa, D, KX_F32, M, D};
      o.bias = L.bo; o.residual = pa; o.residual2 = pb; o.eps = w->eps;
      if (pair) { o.ksplit = 2; o.C2 = qb; }
      if (w->subln) { o.partials_in = d.partials; o.nseg_in = w->heads; o.seg_in = 64; o.colsum = L.wo_colsum; }
      o.W_tiled = L.wo_t; o.tiled_fmt = pieces ? 4 : tfmt; o.prec = prec;
      KX_TRY(gemv16(o, s));
      Gemv16 f1{qa, D, L.w1, D, d.g, F, pieces ? KX_F16P : ct, M, F};
      f1.bias = L.b1; f1.act = w->act; f1.ln_g = L.fl_g; f1.ln_b = L.fl_b; f1.eps = w->eps; f1.a_add = qb;
      if (w->subln) f1.stats_out = d.partials;
      f1.W_tiled = L.w1_t; f1.tiled_fmt = tfmt; f1.prec = prec;
      KX_TRY(gemv16(f1, s));
      pa = x; pb = pair ? d.xb : nullptr;                         // what fc2 writes
      Gemv16 f2{d.g, F, L.w2, F, pa, D, KX_F32, M, D};
      f2.bias = L.b2; f2.residual = qa; f2.residual2 = qb; f2.eps = w->eps;
      if (pair) { f2.ksplit = 2; f2.C2 = pb; }
      if (w->subln) { f2.partials_in = d.partials; f2.nseg_in = F / 16; f2.seg_in = 16; f2.colsum = L.w2_colsum; }
      f2.W_tiled = L.w2_t; f2.tiled_fmt = pieces ? 4 : tfmt; f2.prec = prec;
      KX_TRY(gemv16(f2, s));
    }
    Gemv16 lo{pa, D, w->wout, D, logits, w->vocab, ldt, M, w->vocab};
    lo.ln_g = w->ln_g; lo.ln_b = w->ln_b; lo.eps = w->eps; lo.a_add = pb;
    lo.W_tiled = w->wout_t; lo.tiled_fmt = tfmt; lo.prec = prec;
    return gemv16(lo, s);
  }
  for (int i = 0; i < w->layers; ++i) {
    const kx_decoder_layer& L = w->layer[i];
    KX_TRY(ln(x, nullptr, L.sa_g, L.sa_b, d.h, ct, M, D, w->eps, s));
    // XPos rows of absolute position t (xpos_T = 1: every batch row is the same position)
    KX_FAM(0);
    KX_TRY(gemm(d.h, D, L.wqkv, D, d.qkv, 3 * D, qdt(prec), M, 3 * D, L.bqkv, nullptr, 0, 0.125f, D, prec, s,
                w->xpos ? xq_cs : nullptr, xq_ss, xk_cs, xk_ss, w->xpos ? 1 : 0, w->xpos ? D : 0));
    KX_TRY(kx_attention_decode(d.qkv, (char*)kcache + i * layer_bytes, (char*)vcache + i * layer_bytes, d.att, ct,
                               w->subln ? d.partials : nullptr, B, w->heads, t, Tmax, prec, stream));
    if (w->subln) {
      KX_TRY(kx_row_stats_finalize(d.partials, M, w->heads, 64, w->eps, d.stats, stream));
      KX_FAM(1);
      KX_TRY(gemm(d.att, D, L.wo, D, x, D, KX_F32, M, D, L.bo, x, 0, 1.f, 0, prec, s, nullptr, nullptr, nullptr, nullptr,
                  0, 0, d.stats, L.wo_colsum, nullptr));
    } else {
      KX_FAM(1);
      KX_TRY(gemm(d.att, D, L.wo, D, x, D, KX_F32, M, D, L.bo, x, 0, 1.f, 0, prec, s));
    }
    KX_TRY(ln(x, nullptr, L.fl_g, L.fl_b, d.h, ct, M, D, w->eps, s));
    if (w->subln) {
      KX_FAM(2);
      KX_TRY(gemm(d.h, D, L.w1, D, d.g, F, ct, M, F, L.b1, nullptr, w->act, 1.f, 0, prec, s, nullptr, nullptr, nullptr,
                  nullptr, 0, 0, nullptr, nullptr, d.partials));
      KX_TRY(kx_row_stats_finalize(d.partials, M, F / 64, 64, w->eps, d.stats, stream));
      KX_FAM(3);
      KX_TRY(gemm(d.g, F, L.w2, F, x, D, KX_F32, M, D, L.b2, x, 0, 1.f, 0, prec, s, nullptr, nullptr, nullptr, nullptr,
                  0, 0, d.stats, L.w2_colsum, nullptr));
    } else {
      KX_FAM(2);
      KX_TRY(gemm(d.h, D, L.w1, D, d.g, F, ct, M, F, L.b1, nullptr, w->act, 1.f, 0, prec, s));
      KX_FAM(3);
      KX_TRY(gemm(d.g, F, L.w2, F, x, D, KX_F32, M, D, L.b2, x, 0, 1.f, 0, prec, s));
    }
  }
  KX_TRY(ln(x, nullptr, w->ln_g, w->ln_b, d.h, ct, M, D, w->eps, s));
  KX_FAM(4);
  KX_TRY(gemm(d.h, D, w->wout, D, logits, w->vocab, ldt, M, w->vocab, nullptr, nullptr, 0, 1.f, 0, prec, s));
  return KX_OK;
}
