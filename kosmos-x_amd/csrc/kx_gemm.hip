// C = epilogue(A · Wᵀ) for gfx950 — the GEMM behind every nn.Linear on the Kosmos-X forward path.
//
// Layout decisions (MI355X-first):
//  * Both operands are K-contiguous (activations [M,K], PyTorch weights [N,K]), so A and B MFMA
//    fragments are plain 16-byte reads.  The WEIGHT tile is the MFMA "A" operand and the
//    ACTIVATION tile the "B" operand: the accumulator of v_mfma_f32_16x16x{32 bf16,4 f32} then
//    holds 4 consecutive output COLUMNS (n) per lane for one row (m) — 16-byte epilogue stores,
//    float4 bias/residual reads, and the XPos (2j,2j+1) pairs are lane-local.
//  * Tiles are staged HBM→LDS with global_load_lds_dwordx4 (no VGPR round trip).  The LDS image is
//    lane-linear, so the bank-conflict swizzle (16-B chunk ^= row&7 inside each 128-B tile row) is
//    applied to the per-lane SOURCE address and again on the ds_read_b128 (guide rule 21).
//  * A tile row is always 128 bytes (64 bf16 or 32 f32), so one kernel template serves the bf16
//    MFMA path and the exact-f32 MFMA path (fp32 parity mode).
//  * blockIdx → tile map is XCD-aware (bijective remap so each XCD's L2 sees a contiguous run of
//    tiles) and grouped 8 tile-rows deep so neighbouring blocks share operand panels.
#include "kx_gemm_impl.h"


// K slices for a 64x64-tile launch: ~512 workgroups streaming the weights, at most 16 slices, at least two K-tiles per
// slice, partials [slices, M, N] fp32 within the scratch.  `forced` > 0 overrides the count (tests), < 0 is a minimum.
static long long splitk_slices(int64_t M, int64_t N, int64_t K, int bk, size_t ws_bytes, int forced, bool f16c_rows = false) {
  const long long tiles = ((M + 63) / 64) * ((N + 63) / 64);
  const long long nk_all = K / bk;
  // the ring kernel keeps 64 KB of LDS per workgroup = two workgroups per CU = 512 resident: never more slices than fit in
  // ONE round (a 640-workgroup grid ran 1.25 rounds = twice the time); A/B: tuning key 4 = 6 restores the r1 rule
  // (KX_F16C rows keep the r1 rule and the two-stage kernel: twice the K-tiles per slice, measured 7 % slower on the ring)
  const bool ring = kx_tuning_get(KX_TUNE_GEMM_EPILOGUE) != 6 && !f16c_rows;
  long long sp = forced > 0 ? forced : ring ? (tiles >= 256 ? 1 : 512 / tiles) : (tiles >= 384 ? 1 : (512 + tiles - 1) / tiles);
  if (forced < 0 && sp < -forced) sp = -forced;   // callers whose epilogue lives in the reduce kernel (statistics producer)
  if (sp > 16) sp = 16;
  if (sp > nk_all / 2) sp = nk_all / 2;
  while (sp > 1 && (size_t)sp * M * N * 4 > ws_bytes) --sp;
  if (sp > 1) {
    const long long kchunk = (nk_all + sp - 1) / sp;
    sp = (nk_all + kchunk - 1) / kchunk;          // no empty slices
  }
  return sp < 1 ? 1 : sp;
}

int kx_gemm_auto_splits(int64_t M, int64_t N, int64_t K, int prec, size_t ws_bytes) {
  auto cdiv = [](long long x, long long y) { return (x + y - 1) / y; };
  if (!ws_bytes || cdiv(M, 128) * cdiv(N, 128) >= 192) return 1;       // the automatic tile choice is not 64x64
  if (prec == KX_PREC_F16C) return (int)splitk_slices(M, N, 2 * K, 64, ws_bytes, 0, true);   // 4K-byte rows = 2K 2-byte units
  return (int)splitk_slices(M, N, K, (prec == KX_PREC_BF16 || prec == KX_PREC_F16) ? 64 : 32, ws_bytes, 0);
}

// Sticky device word of the pair split's bounded hand-off (GemmParams.pk_err): 0, or 1 + the index of the last workgroup that
// gave up waiting for its partner's flag.  One word per process and device context; kx_pair_split_errors reads and clears it.
__device__ unsigned g_kx_pair_err;
static unsigned* pair_err_word() {          // the CURRENT device's copy of the word (cached per device ordinal, like kx_cu_count)
  static std::atomic<unsigned*> ptr[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  unsigned* w = ptr[dev].load(std::memory_order_relaxed);
  if (!w) {
    void* q = nullptr;
    if (hipGetSymbolAddress(&q, HIP_SYMBOL(g_kx_pair_err)) != hipSuccess) return nullptr;
    w = (unsigned*)q;
    ptr[dev].store(w, std::memory_order_relaxed);
  }
  return w;
}
extern "C" int kx_pair_split_errors(unsigned* word_out) {
  KX_REQUIRE(word_out != nullptr, "kx_pair_split_errors: null output");
  unsigned* w = pair_err_word();
  KX_REQUIRE(w != nullptr, "kx_pair_split_errors: device word unavailable");
  unsigned v = 0, zero = 0;
  if (hipMemcpy(&v, w, sizeof v, hipMemcpyDeviceToHost) != hipSuccess ||        // synchronises the device: diagnostics, not the hot path
      (v && hipMemcpy(w, &zero, sizeof zero, hipMemcpyHostToDevice) != hipSuccess)) {
    kx_set_error("kx_pair_split_errors: reading the device word failed");
    return KX_ERR_LAUNCH;
  }
  *word_out = v;
  return KX_OK;
}

extern "C" int kx_row_stats_finalize(const float* partials, int64_t rows, int64_t nseg, int64_t seg_size, float eps, float* out,
                                    void* stream);
extern "C" int kx_gemm(const kx_gemm_args* a_in, void* stream) {
  KX_REQUIRE(a_in != nullptr, "kx_gemm: null args");
  // row_stats_scratch AND stats_partials (tile kernels; ABI 7): the partials are what the producer wrote, the scratch is where
  // their finalised (mean, rstd) go when this launch does not finalise them itself: kx_gemm runs kx_row_stats_finalize first.  The pair split
  // with the lean residual epilogue CAN finalise them in the launch (each workgroup the 128 rows it finishes, while it waits for
  // its partner's flag: kx_row_stats_finalize's arithmetic, wave per row; row_stats untouched) — opt-in, tuning key 15 & 32:
  // measured SLOWER on the headline (25.14 vs 24.90 ms same-box alternating: with two steps in flight the 5 us finalize launches
  // hide under the other stream, the in-launch walk sits on the hand-off).  Below this point the call looks like a row_stats
  // call; `fin` remembers the partials.
  kx_gemm_args a_copy;
  const kx_gemm_args* a = a_in;
  struct { const float* partials; int64_t nseg, seg; float eps; } fin = {nullptr, 0, 0, 0.f};
  KX_REQUIRE(!(a_in->row_stats && a_in->stats_partials && a_in->tile != 16),
             "kx_gemm: row_stats together with stats_partials was the ABI 6 form; ABI 7 takes the output scratch as row_stats_scratch");
  KX_REQUIRE(!a_in->row_stats_scratch || (a_in->stats_partials && !a_in->row_stats && a_in->tile != 16),
             "kx_gemm: row_stats_scratch goes with stats_partials on the tile kernels (row_stats NULL)");
  float* fin_out = nullptr;
  if (a_in->row_stats_scratch) {
    KX_REQUIRE(a_in->stats_in_nseg > 0 && a_in->stats_in_seg > 0 && !a_in->ln_out,
               "kx_gemm: row_stats_scratch + stats_partials needs nseg, seg size (and no ln_out: that is the row-owning reduce's form)");
    fin = {a_in->stats_partials, a_in->stats_in_nseg, a_in->stats_in_seg, a_in->stats_eps};
    fin_out = a_in->row_stats_scratch;
    a_copy = *a_in; a_copy.stats_partials = nullptr; a_copy.stats_in_nseg = 0; a_copy.stats_in_seg = 0;
    a_copy.row_stats = fin_out; a_copy.row_stats_scratch = nullptr;     // below this point: a row_stats call
    a = &a_copy;
  }
  KX_REQUIRE(a->A && a->W && a->C, "kx_gemm: null operand pointer");
  KX_REQUIRE(a->M > 0 && a->N > 0 && a->K > 0, "kx_gemm: empty problem M=%lld N=%lld K=%lld", (long long)a->M,
             (long long)a->N, (long long)a->K);
  KX_REQUIRE(a->M < (1ll << 31) && a->N < (1ll << 31) && a->K < (1ll << 31), "kx_gemm: dimension overflow");
  const bool f16c = a->prec == KX_PREC_F16C;
  const bool f16 = a->prec == KX_PREC_F16;                    // plain fp16 rows on the KX_F16C kernels, no correction tiles
  const int es = (a->prec == KX_PREC_BF16 || f16c || f16) ? 2 : 4;   // KX_F16C rows: lda/ldw/ldc count 2-byte units
  const int bk = f16c ? 128 : 128 / es;
  KX_REQUIRE(a->prec == KX_PREC_BF16 || a->prec == KX_PREC_F32 || f16c || f16, "kx_gemm: bad precision %d", a->prec);
  KX_REQUIRE(!f16 || (a->tile != 16 && a->tile != 257), "kx_gemm: KX_PREC_F16 runs the tile kernels 64/128/160/256/384/512");
  KX_REQUIRE(a->cdt != KX_F16 || f16 || f16c, "kx_gemm: KX_F16 outputs come from the fp16 kernels (KX_PREC_F16 / KX_PREC_F16C)");
  KX_REQUIRE(a->cdt != KX_BF16 || !f16, "kx_gemm: KX_PREC_F16 writes KX_F16 or fp32");
  KX_REQUIRE(a->K % bk == 0, "kx_gemm: K=%lld must be a multiple of %d", (long long)a->K, bk);
  KX_REQUIRE(!f16c || (a->w_scale && a->lda >= 2 * a->K && a->ldw >= 2 * a->K && a->tile != 16 && a->tile != 257),
             "kx_gemm: KX_PREC_F16C needs w_scale, lda/ldw >= 2K (2-byte units) and a tile kernel (not 16 / 257)");
  KX_REQUIRE((a->lda * es) % 16 == 0 && (a->ldw * es) % 16 == 0, "kx_gemm: lda/ldw must give 16-byte row pitch");
  KX_REQUIRE(((uintptr_t)a->A & 15) == 0 && ((uintptr_t)a->W & 15) == 0, "kx_gemm: A/W must be 16-byte aligned");
  KX_REQUIRE(a->lda >= a->K && a->ldw >= a->K && a->ldc >= a->N, "kx_gemm: leading dimension too small");
  KX_REQUIRE(a->qcols % 4 == 0, "kx_gemm: qcols must be a multiple of 4");
  if (a->xpos_dim) {
    KX_REQUIRE(a->xq_cs && a->xq_ss && a->xk_cs && a->xk_ss && a->xpos_T > 0, "kx_gemm: xpos tables missing");
    KX_REQUIRE(a->xpos_dim % 64 == 0 && 2 * a->xpos_dim <= a->N, "kx_gemm: xpos_dim must be heads*64 and <= N/2");
  }
  GemmParams p;
  p.A = (const char*)a->A; p.W = (const char*)a->W;
  p.lda_b = a->lda * es; p.ldw_b = a->ldw * es;
  p.C = a->C; p.ldc = a->ldc; p.c_bf16 = a->cdt == KX_BF16 || a->cdt == KX_BF16X3 || a->cdt == KX_F16C || a->cdt == KX_F16;
  p.c_x3 = a->cdt == KX_BF16X3; p.c_f16c = a->cdt == KX_F16C; p.c_f16 = a->cdt == KX_F16;
  p.c_pieces = a->cdt == KX_F16P;
  p.c_hilo = a->cdt == KX_F16HL;
  KX_REQUIRE(!p.c_hilo || (f16c && a->xpos_dim > 0 && a->N % 64 == 0 && a->ldc % 4 == 0 && ((uintptr_t)a->C & 15) == 0 &&
                           !a->residual && !a->stats_out && !a->row_stats && a->act == KX_ACT_NONE && a->tile != 16 &&
                           !a->ln_operand_out && kx_tuning_get(KX_TUNE_GEMM_EPILOGUE) != 1),
             "kx_gemm: a KX_F16HL output comes from KX_PREC_F16C operands with the XPos epilogue (the prefetching store loop), "
             "N %% 64 == 0, 16-byte aligned fp32-pitched rows, no residual / statistics / activation");
  KX_REQUIRE(a->cdt != KX_F16P || (a->tile == 16 && a->prec == KX_PREC_F32 && a->N % 32 == 0 && a->ldc % 32 == 0 &&
                                   ((uintptr_t)a->C & 15) == 0 && !a->residual && a->ksplit <= 1),
             "kx_gemm: KX_F16P rows come from tile 16 on fp32 operands, N %% 32 == 0, ldc %% 32 == 0, no residual / ksplit");
  p.nk_main = f16c ? (int)(a->K / 64) : 0x7fffffff; p.wscale = a->w_scale; p.kskip = 0;
  KX_REQUIRE(a->f16c_corr >= KX_CORR_BOTH && a->f16c_corr <= KX_CORR_NONE && (f16c || a->f16c_corr == KX_CORR_BOTH),
             "kx_gemm: f16c_corr is a kx_f16c_corr value and belongs to KX_PREC_F16C operands");
  KX_REQUIRE(a->cdt != KX_F16C || (a->N % 8 == 0 && a->ldc >= 2 * a->N && a->ldc % 8 == 0 && a->tile != 16 &&
                                    ((uintptr_t)a->C & 15) == 0),
             "kx_gemm: a KX_F16C output needs N %% 8 == 0, ldc >= 2N (2-byte units), ldc %% 8 == 0 (and is not offered by tile 16)");
  KX_REQUIRE(a->cdt != KX_BF16X3 || (a->N % 8 == 0 && a->ldc >= 3 * a->N && a->ldc % 8 == 0 && a->tile != 16 &&
                                      ((uintptr_t)a->C & 15) == 0),
             "kx_gemm: a KX_BF16X3 output needs N %% 8 == 0, ldc >= 3N, ldc %% 8 == 0 (and is not offered by tile 16)");
  p.bias = a->bias; p.residual = a->residual; p.ldr = a->ldr;
  p.M = (int)a->M; p.N = (int)a->N; p.K = (int)(f16c ? 2 * a->K : a->K);   // f16c: 2-byte units of the 4K-byte row
  if (f16c && a->f16c_corr != KX_CORR_BOTH) {
    // one correction product (K / 128 fp8 tiles after the K / 64 fp16 tiles) or none: the loop is shorter, the rows are not
    p.K = (int)(a->f16c_corr == KX_CORR_NONE ? a->K : a->K + a->K / 2);
    if (a->f16c_corr == KX_CORR_ACT) p.kskip = (int)(a->K / 128);            // ... and its fp8 tiles are the rows' second region
  }
  p.act = (a->act == KX_ACT_GELU && (a->prec == KX_PREC_BF16 || f16c || f16)) ? KX_ACT_GELU_FAST : a->act;
  p.qscale = a->qscale; p.qcols = (int)a->qcols;
  p.xq_cs = a->xq_cs; p.xq_ss = a->xq_ss; p.xk_cs = a->xk_cs; p.xk_ss = a->xk_ss;
  p.xpos_T = (int)a->xpos_T; p.xpos_dim = (int)a->xpos_dim;
  p.row_stats = a->row_stats; p.colsum = a->colsum; p.stats_out = a->stats_out; p.stats_nseg = (int)(a->N / 64);
  KX_REQUIRE((!a->row_stats && !a->stats_partials) == !a->colsum, "kx_gemm: row_stats and colsum must be given together");
  KX_REQUIRE(!a->colsum || ((uintptr_t)a->colsum & 15) == 0, "kx_gemm: colsum must be 16-byte aligned");
  KX_REQUIRE(!a->stats_out || a->stats_out_seg == 16 || a->N % 64 == 0, "kx_gemm: stats_out needs N %% 64 == 0 (N=%lld)",
             (long long)a->N);
  KX_REQUIRE(!a->stats_out || (a->qcols == 0 && a->xpos_dim == 0),
             "kx_gemm: stats_out combines with the folded-LN consume, bias and activation only");
  KX_REQUIRE(!a->stats_out || !a->residual, "kx_gemm: stats_out is taken before the residual add; pass one of them");
  p.vec_ok = (a->ldc % 4 == 0) && (!a->residual || a->ldr % 4 == 0) &&
             (((uintptr_t)a->C & 15) == 0) && (!a->residual || ((uintptr_t)a->residual & 15) == 0);
  p.vec8_ok = p.vec_ok && (a->ldc % 8 == 0) && (a->N % 8 == 0);
  p.vec2_ok = a->cdt == KX_F32 && (a->ldc % 2 == 0) && (((uintptr_t)a->C & 7) == 0);
  KX_REQUIRE(!a->bias || ((uintptr_t)a->bias & 15) == 0, "kx_gemm: bias must be 16-byte aligned");
  p.splitk = 1; p.partial = nullptr;
  p.lnop_out = a->ln_operand_out; p.lnop_dt = a->ln_operand_dt; p.lnop_stats = a->ln_operand_stats;
  KX_REQUIRE(!a->ln_operand_out == !a->ln_operand_stats, "kx_gemm: ln_operand_out and ln_operand_stats go together");
  KX_REQUIRE(!a->ln_operand_out ||
                 (a->residual && a->cdt == KX_F32 && a->N % 64 == 0 && p.vec_ok && !a->stats_out && !a->ln_out &&
                  (a->ln_operand_dt == KX_BF16 || a->ln_operand_dt == KX_F16 || a->ln_operand_dt == KX_F16C) &&
                  ((uintptr_t)a->ln_operand_out & 15) == 0 && kx_tuning_get(KX_TUNE_GEMM_EPILOGUE) != 1),
             "kx_gemm: ln_operand_out needs an fp32 output with residual, N %% 64 == 0, aligned rows, a 2-byte / KX_F16C "
             "operand dtype and the prefetching store loop");
  p.stagger_ticks = 0; p.w_tiled = 0;
  p.pairk = 0; p.pk_slab = nullptr; p.pk_flag = nullptr; p.pk_epoch = 0; p.pk_err = nullptr; p.pk_spin_ticks = 100000000u; p.pk_fault = 0;
  p.coop = 0; p.coop_flags = nullptr;
  p.gsplit = 1; p.kfull = p.K; p.C2 = nullptr; p.residual2 = nullptr; p.a_add = nullptr;
  p.no_rowreg = kx_tuning_get(KX_TUNE_GEMV_VARIANT) == 3; p.a_pieces = 0; p.hp = 0; p.valu = 0; p.gb_staged = 0;
  p.ln_g = p.ln_b = nullptr; p.ln_eps = 0.f;
  p.stats_partials = nullptr; p.stats_in_nseg = 0; p.stats_in_seg = p.stats_eps = 0.f;
  p.ln_out = nullptr; p.ln_out_dt = 0; p.ln_out_g = p.ln_out_b = nullptr; p.ln_out_eps = 0.f;
  {
    // The prefetching store loop pays where the epilogue has per-row global operands to wait for (residual, folded-LN
    // statistics, XPos tables); bias-only bf16 epilogues measured ~5 % faster on the plain rolled loop.
    const int mode = kx_tuning_get(KX_TUNE_GEMM_EPILOGUE);   // 0 auto, 1 never, 2 always (A/B)
    p.skip_idle_waves = kx_tuning_get(KX_TUNE_GEMM_IDLE_SKIP) != 1;   // A/B: 1 = off
    // K loop of the 256-column kernel (tuning key 14: 0 = balanced, 1 = the first form, A/B).  tools/kloop_bench.py,
    // profiles/r05_c_kloop_bench.jsonl: +5...13 % on every shape of the forward at K >= 2048 (f16c, bf16, fp16), +3...7 % at
    // K = 1024, bit-identical
    p.bal = kx_tuning_get(KX_TUNE_GEMM_KLOOP) != 1;
    {   // one persistent workgroup per CU unless told otherwise (key 7: -1 = one workgroup per tile, n > 0 = n workgroups)
      const int pv = kx_tuning_get(KX_TUNE_GEMM_PERSISTENT);
      p.persistent = pv < 0 ? 0 : pv > 0 ? pv : kx_cu_count();
    }
    // the lean epilogue covers bf16 outputs (not bf16x3) without residual / folded-LN consume / XPos on
    // 16-byte-aligned rows; launch_p5 also asks for N % 256 == 0.  Everything else keeps the generic loops.
    p.lean_epilogue = kx_tuning_get(KX_TUNE_GEMM_EPILOGUE) != 1 && p.c_bf16 && !p.c_x3 && !p.c_f16c &&
                      ((f16 || f16c) ? p.c_f16 : !p.c_f16) && p.vec8_ok && !a->residual &&
                      !a->row_stats && !a->xpos_dim && a->qcols % 64 == 0 && !(a->stats_out && a->qcols);
    // the decoder's qkv GEMM in bf16: bias + q-scale + XPos, no folded-LN consume / residual / statistics, whole heads per wave
    // A/B only (tuning key 4 = 3): measured SLOWER than the row-major store loop it was meant to replace — C3 qkv 36.5 ms
    // (1084 TFLOP/s) on the generic loop, 49.3 ms with the tables read from global in accumulator layout, 44.5 ms with
    // them staged through LDS (the 256-row variant pushes half of its rotated accumulators through scratch).
    // KX_F16C output (the decoder's fc1 in f16c / mixed) without residual / XPos / q-scale split inside a tile: the 256-column
    // kernel's three-plane lean store (tuning key 4 = 8: generic loops, A/B)
    p.lean_f16c = kx_tuning_get(KX_TUNE_GEMM_EPILOGUE) != 1 && kx_tuning_get(KX_TUNE_GEMM_EPILOGUE) != 8 && p.c_f16c && f16c &&
                  !a->residual && !a->xpos_dim && a->qcols % 64 == 0 && a->N % 16 == 0 && (a->ldc * 2) % 16 == 0 &&
                  !a->ln_operand_out;
    // bias + q-scale + XPos at accumulator level (tables through LDS) and a whole-row tile store: bf16 operands -> bf16
    // rows, f16c operands -> the fp32 q / k / v the split attention reads.  Eligible = 1 (launch_p5 takes it on the 192-row
    // tiles), 2 = asked for on any tile (tuning key 4 = 3, A/B); tuning key 4 = 1 / 8 keep the generic loops.
    {
      const int k4 = kx_tuning_get(KX_TUNE_GEMM_EPILOGUE);
      const bool shape_ok = a->xpos_dim > 0 && !a->residual && !a->row_stats && !a->stats_out && a->act == KX_ACT_NONE &&
                            a->qcols % 256 == 0 && a->xpos_dim % 256 == 0 && !a->ln_operand_out;
      const bool bf = a->prec == KX_PREC_BF16 && a->cdt == KX_BF16 && p.vec8_ok;
      const bool fc = f16c && (a->cdt == KX_F32 || a->cdt == KX_F16HL) && p.vec_ok && a->N % 4 == 0;
      p.lean_xpos = (shape_ok && (bf || fc) && k4 != 1 && k4 != 8 && k4 != 9) ? ((k4 == 3 && bf) ? 2 : 1) : 0;   // (9: only this one off, A/B)
    }
    // fp32 output with residual on whole 256-column tiles (the decoder's out_proj / fc2; pair split or not): bias + folded-LN
    // consume at accumulator level, residual rows requested before the tile is parked (tuning key 15 & 16: generic loops, A/B)
    p.lean_res = mode != 1 && !(kx_tuning_get(KX_TUNE_GEMM_RULES) & 16) && a->cdt == KX_F32 && a->residual && p.vec_ok &&
                 a->act == KX_ACT_NONE && !a->xpos_dim && !a->stats_out && !a->ln_operand_out && a->qcols % 64 == 0 &&
                 a->N % 256 == 0 && !a->ln_out;
    p.ring = mode != 6 && !f16c;
    // A/B: tuning key 4 = 4 keeps the A&S erf in the lean epilogues
    // For plain fp16 rows out of plain fp16 operands (the tower in mixed mode) it is OPT-IN (tuning key 15 & 8): measured +0.6 %
    // on the headline step (1268 -> 1275 samples/s alternating, all model-level parity tests green), but the polynomial's 5.5e-5
    // is a ninth of fp16's rounding step at 1.0 and flips 23 % of the roundings, which breaks the kernel-level contract
    // "fp16 output == rounded fp32 output" (tests/test_f16c_gpu.py::test_gemm_f16_plain) — not worth it
    p.gelu_poly = mode != 4 && a->act == KX_ACT_GELU &&
                  ((a->prec == KX_PREC_BF16 && a->cdt == KX_BF16) ||
                   (f16 && a->cdt == KX_F16 && (kx_tuning_get(KX_TUNE_GEMM_RULES) & 8)));
    p.fast_epilogue = mode == 2 || (mode != 1 && (a->residual || a->row_stats || a->xpos_dim > 0));
  }
  hipStream_t s = (hipStream_t)stream;
  // Kernel-variant choice (measured on MI355X with tools/gemm_bench.py and in situ with bench.py):
  //   64x64    when 128x128 tiles would leave most of the 256 CUs idle (batch-1 shapes);
  //   256x256 phased (one workgroup per CU, 128x64 per wave) when its grid fills the chip — C3, decoder fc1;
  //   256x128 phased/pipelined (one workgroup per CU) when K is long enough to amortise its prologue and its grid
  //            fills the chip evenly — the decoder's M = B*114 GEMMs;
  //   160x128  when it removes a nearly empty trailing wave of 128x128 tiles — the ViT's M = B*257 = 64.25 x 128
  //            GEMMs (520 tiles on 512 slots) — estimated with the cost model below;
  //   128x128  otherwise.
  int tile = a->tile;
  const bool rare_act = a->act == KX_ACT_RELU || a->act == KX_ACT_SWISH;    // see kx_act: the generic 128 x 128 kernel, unsplit
  KX_REQUIRE(!rare_act || tile != 16, "kx_gemm: relu / swish are not offered by the weight-streaming kernel (tile 16)");
  KX_REQUIRE(a->act == KX_ACT_NONE || a->act == KX_ACT_GELU || a->act == KX_ACT_QUICK_GELU || rare_act, "kx_gemm: unknown activation %d", a->act);
  if (rare_act) tile = 128;
  // Pair split of the 256x256 kernel (kx_gemm_args.pair_ws): half a round of 256x256 tiles becomes a full round of
  // (tile, K half) workgroups.  Measured on the decoder's N = 2048 GEMMs at M = 3648 (tools/gemm_bench.py, HISTORY §4.1).
  auto pair_ok = [&]() {
    if (!(a->prec == KX_PREC_BF16 || f16c || f16) || !a->pair_ws || ((uintptr_t)a->pair_ws & 255)) return false;
    const long long t = ((a->M + 255) / 256) * ((a->N + 255) / 256);
    const long long nkt = p.K / (128 / es);
    const int cus = kx_cu_count();
    // (automatic choice: >= 64 K-tiles — half a K extent of 16 tiles does not amortise the exchange: bf16 / fp16 K = 2048
    //  measured 0.93-0.95x, K = 8192 1.16x; f16c rows hold twice the K-tiles: K = 2048 1.11x, K = 8192 1.25x)
    return t % 8 == 0 && 2 * t <= cus && 2 * t >= (85 * cus) / 100 && nkt % 2 == 0 && nkt >= (tile == 1024 ? 8 : 64) && a->act == KX_ACT_NONE &&
           !a->stats_out && !a->ln_out && !a->stats_partials && a->pair_ws_bytes >= 4096 + (size_t)(2 * t) * 131072;
  };
  if (tile == 1024) {
    KX_REQUIRE(pair_ok(), "kx_gemm: tile 1024 (pair split of the 256x256 kernel) needs pair_ws, 16-bit operands, tiles %% 8 == 0 with "
               "0.85 CUs <= 2 tiles <= CUs, an even number of K-tiles, no activation / statistics (M=%lld N=%lld K=%lld)",
               (long long)a->M, (long long)a->N, (long long)a->K);
    p.pairk = 1; tile = 512;
  } else if (tile == 0 && kx_tuning_get(KX_TUNE_GEMM_PAIRK) != 1 && !(kx_tuning_get(KX_TUNE_GEMM_RULES) & 128) && pair_ok()) {
    p.pairk = 1; tile = 512;
  } else if (tile == 0 && (kx_tuning_get(KX_TUNE_GEMM_RULES) & 128) && pair_ok()) {
    // A/B (tuning key 15 & 128, "CU-time" experiment of round 6): what the pair split would take runs as HALF a round of whole
    // 256 x 256 tiles instead — longer per launch, fewer CU-microseconds, the other CUs left to the second stream in flight
    tile = 512;
  }
  if (p.pairk) {
    static std::atomic<unsigned> epoch{0};
    p.pk_flag = (unsigned*)a->pair_ws;
    p.pk_slab = (float*)((char*)a->pair_ws + 4096);
    p.pk_epoch = epoch.fetch_add(1, std::memory_order_relaxed) % 0xfffffff0u + 1u;   // never 0 (= consumed / not yet published)
    p.pk_err = pair_err_word();
    KX_REQUIRE(p.pk_err != nullptr, "kx_gemm: the pair split's error word is unavailable");
    if (kx_tuning_get(KX_TUNE_GEMM_PAIRK) == 2) { p.pk_fault = 1; p.pk_spin_ticks = 200000u; }   // fault injection (tests): 2 ms
  }
  if (tile == 0) {
    auto cdiv = [](long long x, long long y) { return (x + y - 1) / y; };
    const long long t128 = cdiv(a->M, 128) * cdiv(a->N, 128);
    if (t128 < 192) {
      tile = 64;
    } else if (a->prec == KX_PREC_F32) {
      tile = 128;
    } else {
      const long long t256 = cdiv(a->M, 256) * cdiv(a->N, 128);
      const double eff256 = (double)t256 / (double)(cdiv(t256, 256) * 256);
      auto cost = [&](int bm) {  // tile-time units: full waves of 512 resident tiles, cheaper trailing wave if <= 1 tile/CU
        const long long t = cdiv(a->M, bm) * cdiv(a->N, 128);
        const long long full = t / 512, rem = t % 512;
        return bm * ((double)full + (rem == 0 ? 0.0 : (rem <= 256 ? 0.62 : 1.0)));
      };
      const long long t512 = cdiv(a->M, 256) * cdiv(a->N, 256);
      const double eff512 = (double)t512 / (double)(cdiv(t512, 256) * 256);
      // 256x256 (128x64 per wave) is the fastest main loop (1.37 vs 1.09 PFLOP/s at 8192^3: 512 vs 768 B of LDS
      // traffic per MFMA) but needs >= ~0.85 of a 256-CU wave of tiles to pay: C3-sized problems, decoder fc1
      if (a->K >= 1024 && eff512 >= 0.85) tile = 512;
      // ... and, for outputs that take its lean 16-bit tile store, already when the padded rounds cover <= 1.5x the problem:
      // the ViT's qkv / fc1 at M = 32 * 257 (396 / 528 tiles = 2 / 3 rounds, the third one 16 tiles of 32 live rows) measured
      // 63.7 / 95.4 us against 69.7 / 101.5 on the 160-row kernel in bf16 and 79 / 111 there in fp16 (tools/epi_probe.py)
      else if (!(kx_tuning_get(KX_TUNE_GEMM_RULES) & 1) && a->K >= 1024 && p.lean_epilogue && !a->stats_out && a->N % 256 == 0 &&
               (double)(cdiv(t512, 256) * 256) * 65536.0 <= 1.5 * (double)a->M * (double)a->N) tile = 512;
      else if (a->K >= 2048 && eff256 >= 0.85 && a->N <= 16384) tile = 256;
      else tile = cost(160) <= cost(128) ? 160 : 128;
      // 192x256 (96x64 per wave, same kernel): M = 32*114 = 19 x 192 exactly, and the decoder qkv GEMM (N = 6144)
      // then needs 456 tiles = 2 rounds instead of 360 256x256 tiles (also 2 rounds, each 1/0.83 longer) or 720
      // 256x128 tiles (3 rounds).  Relative round times measured with tools/gemm_bench.py: 1.0 / 0.83 / 0.6.
      if ((tile == 512 || tile == 256) && a->K >= 1024) {
        const long long t384 = cdiv(a->M, 192) * cdiv(a->N, 256);
        const double c384 = 0.83 * (double)cdiv(t384, 256);
        const double cur = tile == 512 ? (double)cdiv(t512, 256) : 0.6 * (double)cdiv(t256, 256);
        if (c384 < 0.97 * cur) tile = 384;
      }
      // The ViT's residual GEMMs (fc2, out_proj: N = 1024, fp32 residual epilogue) at M = 32 * 257: 43 x 4 tiles of 192 x 256 on
      // the 256-column kernel measured 91.6 / 38.3 us against 101.4 / 40.2 on the 160-row kernel (fp16, tools/epi_probe.py tower)
      if (!(kx_tuning_get(KX_TUNE_GEMM_RULES) & 2) && tile == 160 && (f16 || f16c || a->prec == KX_PREC_BF16) && a->residual &&
          a->N % 256 == 0 && a->K >= 1024 && cdiv(a->M, 192) * cdiv(a->N, 256) <= kx_cu_count() &&
          cdiv(a->M, 192) * cdiv(a->N, 256) >= kx_cu_count() / 2)
        tile = 384;
      // ... and any N % 256 == 0 problem whose 192 x 256 tiles make one round that fills >= 0.8 of the CUs (the Perceiver's to_kv at
      // B = 32: 10272 x 1024 x 1024 = 54 x 4 tiles: 55.8 us against 75.3 on 128 x 128, tools/tile_probe.py, profiles/r05_k_*)
      // (measured with f16c rows -> fp32 rows + bias, the to_kv launch itself; the rule excludes the epilogue classes
      //  nobody measured on these tiles — produced statistics, ln_operand_out, KX_F16C / KX_BF16X3 outputs, residual + 16-bit
      //  output — and has its own off bit, tuning key 15 & 64)
      if (!(kx_tuning_get(KX_TUNE_GEMM_RULES) & (2 | 64)) && (tile == 160 || tile == 128) && (f16 || f16c || a->prec == KX_PREC_BF16) &&
          !a->stats_out && !a->ln_operand_out && !p.c_f16c && !p.c_x3 && !(a->residual && p.c_bf16) &&
          a->N % 256 == 0 && a->K >= 1024 && cdiv(a->M, 192) * cdiv(a->N, 256) <= kx_cu_count() &&
          cdiv(a->M, 192) * cdiv(a->N, 256) >= (4 * kx_cu_count()) / 5)
        tile = 384;
      // A/B: tuning key 4 = 6 keeps the 128 / 160-row kernels for fp16 rows where bf16 takes the 256x128 ring
      if ((f16c || f16) && tile == 256 && kx_tuning_get(KX_TUNE_GEMM_EPILOGUE) == 6) tile = cost(160) <= cost(128) ? 160 : 128;
    }
  }
  // THROUGHPUT objective (tuning key 18 = 1; A/B alone: key 15 & 256): 256-row tiles wherever the rules above chose 192-row ones.
  // The rules above minimise the time of ONE launch alone on the chip (rounds x tile time).  A caller that keeps two steps in
  // flight on two streams fills a launch's idle CUs with the other step's kernels, and what counts then is the launch's
  // CU-microseconds = tiles x tile time: since the balanced K loop a K-tile costs the same at both heights (2436 vs 2450 cycles),
  // so 19 x 24 tiles of 192 rows (decoder qkv at M = 3648) are 27 % more CU time than 15 x 24 of 256 — for the same two rounds.
  // Measured, same box, alternating (tools/cutime_ab.sh, profiles/r06_h_*): two steps in flight 1280 -> 1314 samples/s (+2.7 %,
  // two boxes); ONE step at a time 27.7 -> 28.5 ms (-3 %: there the rounds are what counts) — hence an objective, not a default.
  // (The other CU-time candidate — whole 256 x 256 tiles on half the chip instead of the pair split, key 15 & 128 — measured
  // +1.3 % alone, +0.2 % on top of this rule, and -10 % with one step at a time: A/B only.)
  if (tile == 384 && a->tile == 0 && ((kx_tuning_get(KX_TUNE_GEMM_RULES) & 256) || kx_tuning_get(KX_TUNE_OBJECTIVE) == 1) &&
      !((kx_tuning_get(KX_TUNE_GEMM_RULES) & 1024) && a->xpos_dim) && !((kx_tuning_get(KX_TUNE_GEMM_RULES) & 2048) && !a->xpos_dim))
    tile = 512;                          // (A/B bits 15 & 1024 / 2048: the XPos launches / all the others keep their 192-row tiles)
  // A 64x64 wave owns 32 columns only: the statistics producer needs the split-K reduce kernel (whose threads walk whole
  // 64-column segments) — when the call will not actually be split, take 128x128 instead (its waves own 64 columns).
  if (tile == 64 && a->ln_operand_out) tile = 128;        // the producer lives in the 64-column store loops
  if (tile == 64 && a->stats_out &&
      !(a->splitk_ws && splitk_slices(a->M, a->N, p.K, 128 / es, a->splitk_ws_bytes, a->splitk ? a->splitk : (a->stats_out && !f16c ? -2 : 0), f16c) > 1))
    tile = 128;
  if (tile == 16) {
    KX_REQUIRE((a->prec == KX_PREC_BF16 || a->prec == KX_PREC_F32) && a->M <= 16,
               "kx_gemm: tile 16 (weight streaming) takes bf16 or fp32 operands, M <= 16 only");
    KX_REQUIRE(!a->w_tiled || a->K % 32 == 0, "kx_gemm: the streaming weight layout needs K %% 32 == 0");
    KX_REQUIRE(a->w_tiled >= 0 && a->w_tiled <= 4 && (a->w_tiled < 2 || (a->prec == KX_PREC_F32 && a->K % 32 == 0)),
               "kx_gemm: w_tiled is 0, 1 or (fp32 operands, K %% 32 == 0) 2 = 24-bit planes / 3 = block-scaled 16-bit weights / 4 = 3 with KX_F16P rows in A");
    KX_REQUIRE(a->w_tiled != 4 || (!a->ln_gamma && a->lda % 32 == 0), "kx_gemm: w_tiled = 4 takes KX_F16P rows (lda %% 32 == 0), no ln_gamma");
    p.w_tiled = a->w_tiled == 4 ? 3 : a->w_tiled;
    p.a_pieces = a->w_tiled == 4;
    KX_REQUIRE(!a->ln_gamma || (a->ln_beta && (size_t)a->M * (a->K * es + 16) <= 144 * 1024 && a->K % 4 == 0),     // (+ 16 KB of accumulators: 160)
               "kx_gemm: LayerNorm prologue needs beta and M*(K*%d+16) <= 144 KB", es);
    KX_REQUIRE(!a->ln_operand_out, "kx_gemm: tile 16 does not produce ln_operand_out");
    KX_REQUIRE(!a->residual2 || a->residual, "kx_gemm: residual2 is the second addend of `residual`");
    KX_REQUIRE(!a->a_add || a->ln_gamma, "kx_gemm: a_add is the second addend of the LayerNorm-prologue rows (ln_gamma)");
    KX_REQUIRE(!(a->residual2 || a->ksplit > 1) || (p.vec_ok && a->N % 16 == 0 && !a->xpos_dim && !(a->row_stats && !a->stats_partials)),
               "kx_gemm: residual2 / ksplit need N %% 16 == 0, 16-byte aligned rows, no XPos, statistics as partials");
    KX_REQUIRE(kx_tuning_get(KX_TUNE_GEMV_VARIANT) != 1 || !(a->residual2 || a->a_add || a->ksplit > 1),
               "kx_gemm: the pair form of the residual stream needs the second form of the streaming kernel (tuning key 8 != 1)");
    p.residual2 = a->residual2; p.a_add = a->a_add;
    if (a->ksplit > 1) {
      KX_REQUIRE(a->ksplit == 2 && a->C2 && a->cdt == KX_F32 && a->act == KX_ACT_NONE && !a->stats_out && !a->ln_gamma &&
                     a->K % (64 * a->ksplit) == 0 && ((uintptr_t)a->C2 & 15) == 0,
                 "kx_gemm: ksplit = 2 needs C2, an fp32 C, no activation / statistics producer / ln_gamma, K %% 128 == 0");
      p.gsplit = a->ksplit; p.C2 = a->C2; p.kfull = p.K; p.K = p.K / a->ksplit;
    }
    KX_REQUIRE(!a->stats_partials || (a->colsum && !a->row_stats && a->stats_in_nseg > 0 && a->stats_in_seg > 0),
               "kx_gemm: stats_partials needs colsum, nseg, seg size and excludes row_stats");
    KX_REQUIRE(!a->stats_out || (a->stats_out_seg == 16 && a->N % 16 == 0 && !a->residual),
               "kx_gemm: tile 16 emits statistics per 16-column segment (stats_out_seg = 16, N %% 16 == 0, no residual)");
    p.ln_g = a->ln_gamma; p.ln_b = a->ln_beta; p.ln_eps = a->ln_eps;
    if (a->ln_gamma) p.lda_b = a->lda * 4;              // A holds fp32 rows in this mode
    p.stats_partials = a->stats_partials; p.stats_in_nseg = (int)a->stats_in_nseg;
    p.stats_in_seg = (float)a->stats_in_seg; p.stats_eps = a->stats_eps;
    if (a->stats_out) p.stats_nseg = (int)(a->N / 16);
  } else {
    KX_REQUIRE(!(a->ln_gamma || (a->stats_out_seg != 0 && a->stats_out_seg != 64)),
               "kx_gemm: ln_gamma / stats_out_seg = 16 belong to tile 16 (weight streaming)");
    KX_REQUIRE(a->ksplit <= 1 && !a->C2 && !a->residual2 && !a->a_add, "kx_gemm: ksplit / C2 / residual2 / a_add belong to tile 16");
    if (a->stats_partials) {                            // consumed by the row-owning split-K reduce (checked below)
      KX_REQUIRE(a->colsum && !a->row_stats && a->stats_in_nseg > 0 && a->stats_in_seg > 0,
                 "kx_gemm: stats_partials needs colsum, nseg, seg size and excludes row_stats");
      p.stats_partials = a->stats_partials; p.stats_in_nseg = (int)a->stats_in_nseg;
      p.stats_in_seg = (float)a->stats_in_seg; p.stats_eps = a->stats_eps;
    }
  }
  if (tile == 64 && a->splitk_ws) {
    // Skinny problems (batch-1 shapes: M = 114 / 257 / 64) are weight-streaming bound and a 64x64 grid of N/64 x 2
    // workgroups leaves most CUs idle while each one walks all of K serially.  Slice K so that ~512 workgroups
    // stream the weights concurrently; partials are small ([splits][M][N] fp32, L2/MALL resident).
    long long sp = splitk_slices(a->M, a->N, p.K, 128 / es, a->splitk_ws_bytes, a->splitk ? a->splitk : (a->stats_out && !f16c ? -2 : 0), f16c);
    // ... unless the tiles alone nearly fill the chip (the batch-1 qkv GEMMs: 192 / 240 tiles) and nothing needs the reduce
    // kernel: then ONE workgroup per tile walks all of K on the 8-stage ring (six K-tiles in flight instead of three) and
    // the reduce launch with its dirty-L2 boundary disappears.  Tuning key 4 = 7 keeps the split.
    const long long tiles64 = ((a->M + 63) / 64) * ((a->N + 63) / 64);
    if (sp == 2 && !a->splitk && p.ring && es == 2 && !f16 && tiles64 >= 160 && tiles64 <= kx_cu_count() && !a->stats_out &&
        !a->ln_out && !a->stats_partials && kx_tuning_get(KX_TUNE_GEMM_EPILOGUE) != 7 && p.K / 64 >= 8) {
      sp = 1;
      p.ring = 8;
    }
    if (sp > 1 && p.c_hilo) sp = 1;                      // the reduce kernel's epilogue does not write KX_F16HL rows
    if (sp > 1) {
      p.splitk = (int)sp;
      p.partial = (float*)a->splitk_ws;
      // in-launch reduction (GemmParams.coop; the caller asks for it by passing splitk_flags — the stage entry points only with
      // tuning key 17 = 1: MEASURED no faster than the reduce launch it replaces, profiles/r06_d_*): every workgroup resident
      // (64 KB of LDS / 256 threads: two per CU), the row-owning reduce's preconditions, and a kernel that exists in this form
      // (the ring for bf16 / fp16 / fp32 rows, two stages for KX_F16C)
      p.coop = a->splitk_flags && kx_tuning_get(KX_TUNE_SPLITK_COOP) != 2 && a->N <= 8192 && a->N % 4 == 0 && p.vec_ok &&
               tiles64 * sp <= 2ll * kx_cu_count() && a->M <= tiles64 * sp && (p.ring == 1 || f16c) && !p.c_hilo && !p.c_pieces;
      if (p.coop) {
        static std::atomic<unsigned> coop_epoch{0};
        p.coop_flags = a->splitk_flags;
        p.pk_epoch = coop_epoch.fetch_add(1, std::memory_order_relaxed) % 0xfffffff0u + 1u;     // never 0: a cleared word is "not arrived"
        p.pk_err = pair_err_word();
        KX_REQUIRE(p.pk_err != nullptr, "kx_gemm: the hand-off error word is unavailable");
      }
    }
  }
  // row-owning reduce (LayerNorm of the output row as a second output; folded-LN statistics from partials)
  p.ln_out = a->ln_out; p.ln_out_dt = a->ln_out_dt; p.ln_out_g = a->ln_out_gamma; p.ln_out_b = a->ln_out_beta;
  p.ln_out_eps = a->ln_out_eps;
  if (a->ln_out || (a->stats_partials && tile != 16)) {
    KX_REQUIRE(p.splitk > 1 && a->N <= 8192 && a->N % 4 == 0 && p.vec_ok,
               "kx_gemm: ln_out / stats_partials need the split-K row reduce (skinny problem with scratch, N <= 8192, "
               "N %% 4 == 0, aligned C): M=%lld N=%lld K=%lld splits=%d", (long long)a->M, (long long)a->N, (long long)a->K,
               p.splitk);
    KX_REQUIRE(!a->ln_out || (a->ln_out_gamma && a->ln_out_beta && (((uintptr_t)a->ln_out | (uintptr_t)a->ln_out_gamma |
                                                                   (uintptr_t)a->ln_out_beta) & 15) == 0),
               "kx_gemm: ln_out needs 16-byte aligned gamma / beta / output");
    KX_REQUIRE(!a->stats_out, "kx_gemm: the row reduce does not produce statistics");
  }
  if (fin.partials) {
    if (p.pairk && p.lean_res && fin.nseg <= 128 && (kx_tuning_get(KX_TUNE_GEMM_RULES) & 32)) {   // opt-in: measured slower (see the header)
      p.stats_partials = fin.partials; p.stats_in_nseg = (int)fin.nseg; p.stats_in_seg = (float)fin.seg; p.stats_eps = fin.eps;
    } else {
      const int rc = kx_row_stats_finalize(fin.partials, a->M, fin.nseg, fin.seg, fin.eps, fin_out, stream);
      if (rc != KX_OK) return rc;
    }
  }
  // 16-bit tile kernels: [bf16 | f16c | f16] x [128, 64, 160, 256x128, 256x256]
  static const int kinds16[3][5] = {
      {KX_K_GEMM_BF16_128, KX_K_GEMM_BF16_64, KX_K_GEMM_BF16_160, KX_K_GEMM_BF16_256X128, KX_K_GEMM_BF16_256X256},
      {KX_K_GEMM_F16C_128, KX_K_GEMM_F16C_64, KX_K_GEMM_F16C_160, KX_K_GEMM_F16C_256X128, KX_K_GEMM_F16C_256X256},
      {KX_K_GEMM_F16_128, KX_K_GEMM_F16_64, KX_K_GEMM_F16_160, KX_K_GEMM_F16_256X128, KX_K_GEMM_F16_256X256}};
  const int tsel = (tile == 64 || tile == 16) ? 1 : tile == 160 ? 2 : (tile == 256 || tile == 257) ? 3 : (tile == 512 || tile == 384) ? 4 : 0;
  const int kind = a->prec == KX_PREC_F32 ? (tile == 64 || tile == 16 ? KX_K_GEMM_F32_64 : KX_K_GEMM_F32_128)
                                          : kinds16[f16c ? 1 : f16 ? 2 : 0][tsel];
  KxProfScope prof(kind, a->M, a->N, a->K, s);
  if (tile == 512 || tile == 384) {
    const int st = kx_tuning_get(KX_TUNE_GEMM_STAGGER);
    if (st >= 200000) { if (a->residual) p.stagger_ticks = st - 200000; }   // A/B: only the in-place residual epilogues
    else if (st > 0) p.stagger_ticks = st;
  }
  if (a->prec == KX_PREC_BF16) {
    if (tile == 256 || tile == 257 || tile == 512 || tile == 384) return kx_gemm_launch_phased_bf16(p, tile, s);
    return kx_gemm_launch_tiles_bf16(p, tile, s);
  }
  if (f16c || f16) return kx_gemm_launch_f16c(p, tile, s);
  return kx_gemm_launch_f32(p, tile, s);
}
