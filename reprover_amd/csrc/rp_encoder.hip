// ByT5 / T5 encoder forward + masked mean-pool + L2 normalise on gfx950.
//
// Replaces (reference lean-dojo/ReProver) retrieval/model.py:92-114 `_encode` and the HuggingFace
// T5Stack.forward it calls (transformers models/t5/modeling_t5.py:663-750, blocks :435-509,
// attention :281-369, RMSNorm :50-72, gated-GELU FFN :97-123).  Exact math: SURVEY.md App. A.
//
// Data layout in HBM for a pass over T packed tokens (Tp = T rounded up to 256):
//   xb   bf16 [Tp, D]        hi plane of the residual stream, bf16(x): also the A operand of the QKV / FFN-in GEMMs
//   xlo  bf16 [Tp, D]        lo plane, bf16(x - xb): x = xb + xlo to 2^-18 relative (HF-bf16 keeps 8 mantissa bits)
//   ssp  f32  [D/64, Tp]     slot-major partial sums of squares of x (RMSNorm statistic; the norm weight is folded
//   rs   f32  [Tp]           into the consuming weights, rs = rsqrt(mean x^2 + eps) multiplies in their epilogues)
//   qkv  bf16 [Tp, 3*H*64]   fused projection output, [q | k | v], head-major inside each
//   att  bf16 [Tp, H*64]     attention output
//   ff   bf16 [Tp, F]        gelu_new(wi_0 h) * (wi_1 h)
// Sequences are packed back to back (varlen): no padded token is ever computed except the
// < 256 rows that round the last GEMM tile.  Device code: rp_encoder_kernels.h (shared with rp_train.hip).
#include "rp_encoder_kernels.h"

namespace rp {

thread_local std::string g_last_error;

// ------------------------------------------------------------------------------------------
// options
// ------------------------------------------------------------------------------------------
extern int g_scan_waves, g_scan_small_tiles;
extern int g_scan_cfg, g_scan_impl, g_scan_filter_cfg, g_scan_sample_cfg, g_scan_stride, g_scan_no_epilogue, g_scan_impl_force_new,
    g_scan_cap, g_train_dbg, g_train_wgrad_form;
int g_gemm_group_m = 8;
// Tile configuration per encoder GEMM (see launch_gemm()), measured at 65536 tokens (tools/gemm_bench.py,
// profiles/).  20 / 26 = the software-pipelined 256 x 256 x 64 tile with 4 / 8 waves: the same main
// loop; 8 waves finish the epilogues sooner.  The gated-GELU GEMM ran the 4-wave form until round 3 (equal at 65 k tokens
// in round 2's isolated runs); inside the step the 8-wave form is 1.1 % faster at 70 k tokens (15.15 -> 14.99 ms over the 12
// launches, three A/B pairs) and 2-5 % at 9-44 k tokens (tools/gemm_bench.py), so every big GEMM uses it now.  The K = H*64
// attention-output projection is bound by the read-modify-write of x whatever the tiling: two small
// blocks per CU overlap one block's epilogue with the other's main loop.
int g_gemm_variant = 26;      // FFN-in (wi_0|wi_1 + gated GELU)
int g_gemm_variant_qkv = 26;  // QKV
int g_gemm_variant_wo = 26;   // FFN-out (+ residual)
int g_gemm_variant_o = -1;    // attention output (+ residual); -1 = by pass size: two 128 x 128 blocks per CU, or - from 57 k
                              // tokens - the 8-wave 256 x 256 tile (A/B inside the 70 k-token step: 2.553 -> 2.470 ms per 12
                              // launches, three pairs; in isolation the two alternate below that size, tools/gemm_bench.py)
int g_gemm_rs_lds = 0;      // 1: big tiles reduce the RMSNorm statistic in the consuming GEMM from slot rows DMA'd into LDS (RowScaleLds:
                            // 24 launches per pass fewer, the same bits).  Round 5 A/B inside the 70 k-token step, two boxes: the
                            // rowscale launches + a 4-byte global read per token in the epilogue are FASTER than the LDS form, by
                            // 0.48 ms per step with every lane summing its tokens' slots (round 4's form: 92 LDS reads + adds per
                            // lane and tile) and still by 0.12 ms with one thread per token summing once per tile (reduce())
int g_gemm_small_pipe = 1;  // few-token passes: the 64 x 128 x 64 tile on the software-pipelined loop (variant 17) instead of the plain one (16)
int g_gemm_helpers = 64;      // few-token launches: up to this many surplus workgroups prefetch the weight rows (0 = off)
int g_gemm_persist = 9;   // persistent workgroups (gemm_tiles_persist) per projection: 1 QKV, 4 attention-out, 8 FFN-in, 16 FFN-out
int g_pool_chunk = 64;      // tokens per workgroup of the pooling pass (32 / 64 / 128; round 5 A/B at 70 k tokens: 98 / 100 / 104 us)
int g_gemm_edge_layout = 1;  // big tiles: the last feature tile of 1152 / 1472 features on a wave grid over its valid features only
int g_gemm_mixed = 20;  // full and half tiles in ONE launch (gemm_kernel_mixed) per projection: 1 QKV, 4 attention-out, 16 FFN-out
int g_gemm_mixed_bwd = 1;  // the same for the training step's dgrad GEMMs whose tile count leaves a short last round (plan_mixed_loose)
int g_gemm_tail_variant = 30;  // tile configuration of that tail round: 30 = 256 x 128 x 64 half tiles, 0 = 128 x 128 x 32 quarter tiles
int g_gemm_tail_split = 1;  // big passes without a mixed launch: the last partial round of 256 x 256 tiles as one round of smaller tiles
int g_debug_skip_ffn = 0;  // parity debugging: stop each block after the attention sub-layer
int g_gemm_skinny = 1;
int g_gemm_skinny_variant = 0;  // 15: few-token passes forced onto variant 15 (a test); 0: the measured table of pick_gemm_variant
// Experiment knob: spread the first round of workgroups of the big GEMMs over this many microseconds (0 = off).
// All 256 CUs otherwise reach their epilogues at the same moment, round after round (tiles take equal time), and
// the epilogue traffic arrives in bursts.  Indexed by kernel class (RP_K_GEMM_*).
int g_gemm_stagger_us[RP_K_COUNT] = {};

// ------------------------------------------------------------------------------------------
// per-kernel event timing
// ------------------------------------------------------------------------------------------
struct ProfRec {
  int cls;
  hipEvent_t a, b;
};
static bool g_prof_on = false;
static std::vector<ProfRec> g_prof_recs;
static std::vector<hipEvent_t> g_prof_pool;

static hipEvent_t prof_event() {
  if (!g_prof_pool.empty()) {
    hipEvent_t e = g_prof_pool.back();
    g_prof_pool.pop_back();
    return e;
  }
  hipEvent_t e;
  (void)hipEventCreate(&e);
  return e;
}
void prof_begin(hipStream_t stream, int kernel_class) {
  if (!g_prof_on) return;
  ProfRec r{kernel_class, prof_event(), prof_event()};
  (void)hipEventRecord(r.a, stream);
  g_prof_recs.push_back(r);
}
void prof_end(hipStream_t stream) {
  if (!g_prof_on) return;
  (void)hipEventRecord(g_prof_recs.back().b, stream);
}

}  // namespace rp


using namespace rp;

extern "C" int32_t rp_abi_version(void) { return 6; }
extern "C" const char* rp_last_error(void) { return g_last_error.c_str(); }

extern "C" RpStatus rp_set_option(const char* name, int32_t value) {
  if (!strcmp(name, "gemm_group_m")) {
    RP_REQUIRE(value >= 1 && value <= 64, "gemm_group_m out of range");
    g_gemm_group_m = value;
    return RP_OK;
  }
  if (!strcmp(name, "gemm_variant")) {
    RP_REQUIRE(value >= 0 && value <= 30, "gemm_variant out of range");
    g_gemm_variant = value;
    return RP_OK;
  }
  if (!strcmp(name, "gemm_variant_all")) {  // benches/tests: one configuration for every GEMM; -1 = defaults
    RP_REQUIRE(value >= -1 && value <= 30, "gemm_variant_all out of range");
    if (value < 0) {
      g_gemm_variant = g_gemm_variant_qkv = g_gemm_variant_wo = 26;
      g_gemm_variant_o = -1;
    } else {
      g_gemm_variant = g_gemm_variant_qkv = g_gemm_variant_wo = g_gemm_variant_o = value;
    }
    return RP_OK;
  }
  if (!strcmp(name, "gemm_variant_qkv")) {
    RP_REQUIRE(value >= 0 && value <= 30, "gemm_variant_qkv out of range");
    g_gemm_variant_qkv = value;
    return RP_OK;
  }
  if (!strcmp(name, "gemm_variant_wo")) {
    RP_REQUIRE(value >= 0 && value <= 30, "gemm_variant_wo out of range");
    g_gemm_variant_wo = value;
    return RP_OK;
  }
  if (!strcmp(name, "gemm_variant_o")) {
    RP_REQUIRE(value >= 0 && value <= 30, "gemm_variant_o out of range");
    g_gemm_variant_o = value;
    return RP_OK;
  }
  if (!strcmp(name, "debug_skip_ffn")) {
    g_debug_skip_ffn = value != 0;
    return RP_OK;
  }
  if (!strcmp(name, "gemm_skinny_variant")) {
    RP_REQUIRE(value == 0 || value == 15, "gemm_skinny_variant must be 0 or 15");
    g_gemm_skinny_variant = value;
    return RP_OK;
  }
  if (!strcmp(name, "gemm_helpers")) {
    RP_REQUIRE(value >= 0 && value <= 248, "gemm_helpers out of range");
    g_gemm_helpers = value & ~7;
    return RP_OK;
  }
  if (!strcmp(name, "gemm_small_pipe")) {
    g_gemm_small_pipe = value != 0;
    return RP_OK;
  }
  if (!strcmp(name, "gemm_rs_lds")) {
    g_gemm_rs_lds = value != 0;
    return RP_OK;
  }
  if (!strcmp(name, "pool_chunk")) {
    RP_REQUIRE(value == 32 || value == 64 || value == 128, "pool_chunk must be 32, 64 or 128");
    g_pool_chunk = value;
    return RP_OK;
  }
  if (!strcmp(name, "gemm_persist")) {
    g_gemm_persist = value;
    return RP_OK;
  }
  if (!strcmp(name, "gemm_edge_layout")) {
    g_gemm_edge_layout = value != 0;
    return RP_OK;
  }
  if (!strcmp(name, "gemm_mixed")) {
    RP_REQUIRE(value >= 0 && value <= 31, "gemm_mixed: bit mask 0..31");
    g_gemm_mixed = value;
    return RP_OK;
  }
  if (!strcmp(name, "gemm_mixed_bwd")) {
    g_gemm_mixed_bwd = value != 0;
    return RP_OK;
  }
  if (!strcmp(name, "gemm_tail_variant")) {
    RP_REQUIRE(value == 0 || value == 30, "gemm_tail_variant: 0 or 30");
    g_gemm_tail_variant = value;
    return RP_OK;
  }
  if (!strcmp(name, "gemm_tail_split")) {
    g_gemm_tail_split = value != 0;
    return RP_OK;
  }
  if (!strcmp(name, "gemm_skinny")) {
    g_gemm_skinny = value != 0;
    return RP_OK;
  }
  if (!strcmp(name, "scan_cfg")) {
    RP_REQUIRE(value >= 0 && value <= 1, "scan_cfg out of range");
    g_scan_cfg = value;
    return RP_OK;
  }
#ifdef RP_EXPERIMENTS  // knobs of measured-and-rejected alternatives and timing probes: probe builds only
  if (!strncmp(name, "gemm_stagger_us_", 16)) {  // gemm_stagger_us_{qkv,o,wi,wo}
    const char* which = name + 16;
    const int cls = !strcmp(which, "qkv") ? RP_K_GEMM_QKV : !strcmp(which, "o") ? RP_K_GEMM_O
                    : !strcmp(which, "wi") ? RP_K_GEMM_WI : !strcmp(which, "wo") ? RP_K_GEMM_WO : -1;
    RP_REQUIRE(cls >= 0 && value >= 0 && value <= 1000, "gemm_stagger_us: unknown GEMM or value out of range");
    g_gemm_stagger_us[cls] = value;
    return RP_OK;
  }
  if (!strcmp(name, "scan_filter_cfg")) { g_scan_filter_cfg = value; return RP_OK; }
  if (!strcmp(name, "scan_sample_cfg")) { g_scan_sample_cfg = value; return RP_OK; }
  if (!strcmp(name, "scan_waves")) { g_scan_waves = value; return RP_OK; }
  if (!strcmp(name, "scan_stride")) { g_scan_stride = value; return RP_OK; }
  if (!strcmp(name, "train_dbg")) { g_train_dbg = value; return RP_OK; }
  if (!strcmp(name, "scan_no_epilogue")) { g_scan_no_epilogue = value; return RP_OK; }
#endif
  if (!strcmp(name, "train_wgrad_form")) {  // tests: the weight-gradient launch forms of the training step (rp_train.hip)
    RP_REQUIRE(value >= 0 && value <= 3, "train_wgrad_form: bit mask 0..3");
    g_train_wgrad_form = value;
    return RP_OK;
  }
  if (!strcmp(name, "scan_cap")) { g_scan_cap = value; return RP_OK; }  // tests: forces the overflow -> dense contract
  if (!strcmp(name, "scan_force_new")) { g_scan_impl_force_new = value; return RP_OK; }
  if (!strcmp(name, "scan_small_tiles")) {  // tests: 0 = at most 32 queries on the 128-query tiles (must be the same bits)
    RP_REQUIRE(value >= 0 && value <= 1, "scan_small_tiles out of range");
    g_scan_small_tiles = value;
    return RP_OK;
  }
  if (!strcmp(name, "scan_impl")) {
    RP_REQUIRE(value >= 0 && value <= 1, "scan_impl out of range");
    g_scan_impl = value;
    return RP_OK;
  }
  return fail(RP_E_INVALID, "unknown option %s", name);
}

extern "C" RpStatus rp_profile_enable(int32_t on) {
  for (auto& r : g_prof_recs) {
    g_prof_pool.push_back(r.a);
    g_prof_pool.push_back(r.b);
  }
  g_prof_recs.clear();
  g_prof_on = on != 0;
  return RP_OK;
}

extern "C" RpStatus rp_profile_read(int32_t kernel_class, double* total_ms, int64_t* launches) {
  RP_REQUIRE(total_ms && launches && kernel_class >= 0 && kernel_class < RP_K_COUNT, "bad argument");
  double tot = 0;
  int64_t n = 0;
  for (auto& r : g_prof_recs) {
    if (r.cls != kernel_class) continue;
    RP_HIP(hipEventSynchronize(r.b));
    float ms = 0;
    RP_HIP(hipEventElapsedTime(&ms, r.a, r.b));
    tot += ms;
    ++n;
  }
  *total_ms = tot;
  *launches = n;
  return RP_OK;
}

// modeling_t5.py:216-262, bidirectional branch; float32 arithmetic as torch evaluates it.
extern "C" int32_t rp_relative_position_bucket(int32_t rel, int32_t num_buckets, int32_t max_distance) {
  int nb = num_buckets / 2;
  int bucket = (rel > 0) ? nb : 0;
  int n = rel < 0 ? -rel : rel;
  int max_exact = nb / 2;
  if (n < max_exact) return bucket + n;
  float ratio = (float)n / (float)max_exact;
  float t = logf(ratio) / (float)log((double)max_distance / (double)max_exact) * (float)(nb - max_exact);
  int large = max_exact + (int)t;  // truncation toward zero, as .to(torch.long)
  if (large > nb - 1) large = nb - 1;
  return bucket + large;
}

template <typename T>
static RpStatus pack_all(RpEncoder* e, const RpT5Weights* w) {
  const RpT5Config& c = e->cfg;
  const int D = c.d_model, F = c.d_ff, inner = e->inner;
  auto alloc = [&](size_t bytes, void** p) -> RpStatus {
    RP_HIP(hipMalloc(p, bytes));
    e->allocs.push_back(*p);
    return RP_OK;
  };
  RpStatus st;
  if ((st = alloc((size_t)c.vocab_size * D * 4, (void**)&e->embed))) return st;
  hipLaunchKernelGGL((to_f32_kernel<T>), dim3((c.vocab_size * D + 255) / 256), dim3(256), 0, 0, e->embed,
                     w->embed, c.vocab_size * D);
  if ((st = alloc((size_t)c.vocab_size * D * 2, (void**)&e->embed_hi))) return st;
  if ((st = alloc((size_t)c.vocab_size * D, (void**)&e->embed_lo))) return st;
  if ((st = alloc((size_t)c.vocab_size * 4, (void**)&e->embed_ss))) return st;
  embed_table_x24(e, 0);
  if ((st = alloc((size_t)D * 4, (void**)&e->final_ln))) return st;
  hipLaunchKernelGGL((to_f32_kernel<T>), dim3((D + 255) / 256), dim3(256), 0, 0, e->final_ln, w->final_ln, D);
  e->layers.resize(c.num_layers);
  for (int i = 0; i < c.num_layers; ++i) {
    const RpT5LayerWeights& s = w->layers[i];
    LayerPacked& L = e->layers[i];
    if ((st = alloc((size_t)D * 4, (void**)&L.ln_attn))) return st;
    if ((st = alloc((size_t)D * 4, (void**)&L.ln_ff))) return st;
    if ((st = alloc((size_t)3 * inner * D * 2, (void**)&L.wqkv))) return st;
    if ((st = alloc((size_t)D * inner * 2, (void**)&L.wo))) return st;
    if ((st = alloc((size_t)2 * F * D * 2, (void**)&L.wi))) return st;
    if ((st = alloc((size_t)D * F * 2, (void**)&L.wo2))) return st;
    hipLaunchKernelGGL((to_f32_kernel<T>), dim3((D + 255) / 256), dim3(256), 0, 0, L.ln_attn, s.ln_attn, D);
    hipLaunchKernelGGL((to_f32_kernel<T>), dim3((D + 255) / 256), dim3(256), 0, 0, L.ln_ff, s.ln_ff, D);
    hipLaunchKernelGGL((pack_rows_kernel<T>), dim3(3 * inner), dim3(256), 0, 0, L.wqkv, s.q, s.k, s.v, 3 * inner,
                       D, inner, (int)PACK_CONCAT3, (const float*)L.ln_attn);
    hipLaunchKernelGGL((pack_rows_kernel<T>), dim3(D), dim3(256), 0, 0, L.wo, s.o, nullptr, nullptr, D, inner, 0,
                       (int)PACK_COPY, (const float*)nullptr);
    hipLaunchKernelGGL((pack_rows_kernel<T>), dim3(2 * F), dim3(256), 0, 0, L.wi, s.wi_0, s.wi_1, nullptr, 2 * F,
                       D, 0, (int)PACK_GEGLU, (const float*)L.ln_ff);
    hipLaunchKernelGGL((pack_rows_kernel<T>), dim3(D), dim3(256), 0, 0, L.wo2, s.wo, nullptr, nullptr, D, F, 0,
                       (int)PACK_COPY, (const float*)nullptr);
    RP_CHECK_LAUNCH();
  }
  // relative-position bias -> [H, 2*maxd+1] table (host; the raw table is tiny)
  const int nbk = c.rel_num_buckets, H = c.num_heads, maxd = e->maxd, ntab = 2 * maxd + 1;
  std::vector<float> raw((size_t)nbk * H);
  {
    float* tmp;
    RP_HIP(hipMalloc((void**)&tmp, raw.size() * 4));
    hipLaunchKernelGGL((to_f32_kernel<T>), dim3((nbk * H + 255) / 256), dim3(256), 0, 0, tmp, w->rel_bias, nbk * H);
    RP_HIP(hipMemcpy(raw.data(), tmp, raw.size() * 4, hipMemcpyDeviceToHost));
    RP_HIP(hipFree(tmp));
  }
  std::vector<float> tab((size_t)H * ntab);
  for (int d = -maxd; d <= maxd; ++d) {
    const int bk = rp_relative_position_bucket(d, nbk, c.rel_max_distance);
    for (int h = 0; h < H; ++h) tab[(size_t)h * ntab + d + maxd] = raw[(size_t)bk * H + h];
  }
  if ((st = alloc(tab.size() * 4, (void**)&e->bias_tab))) return st;
  RP_HIP(hipMemcpy(e->bias_tab, tab.data(), tab.size() * 4, hipMemcpyHostToDevice));
  RP_HIP(hipDeviceSynchronize());
  return RP_OK;
}

extern "C" RpStatus rp_encoder_create(const RpT5Config* cfg, const RpT5Weights* weights, int32_t weight_dtype,
                                      RpEncoder** out) {
  RP_REQUIRE(cfg && weights && out, "null argument");
  if (cfg->d_kv != 64) return fail(RP_E_UNSUPPORTED, "d_kv=%d: kernels implement d_kv=64", cfg->d_kv);
  if (cfg->d_model % 32 || cfg->d_model > RMS_MAX_V4 * 256 || cfg->d_ff % 32)
    return fail(RP_E_UNSUPPORTED, "d_model=%d (multiple of 32, <= %d) / d_ff=%d (multiple of 32) unsupported",
                cfg->d_model, RMS_MAX_V4 * 256, cfg->d_ff);
  if (2 * cfg->rel_max_distance + 1 + 128 > ATT_TAB_MAX)  // (+ the attention kernel's 2 x 64 padding entries)
    return fail(RP_E_UNSUPPORTED, "relative_attention_max_distance=%d too large", cfg->rel_max_distance);
  RP_REQUIRE(weight_dtype == RP_DT_F32 || weight_dtype == RP_DT_BF16, "weight_dtype");
  RpEncoder* e = new RpEncoder();
  e->cfg = *cfg;
  e->inner = cfg->num_heads * cfg->d_kv;
  e->maxd = cfg->rel_max_distance;
  RpStatus st = (weight_dtype == RP_DT_F32) ? pack_all<float>(e, weights) : pack_all<bf16_t>(e, weights);
  if (st != RP_OK) {
    rp_encoder_destroy(e);
    return st;
  }
  *out = e;
  return RP_OK;
}

extern "C" void rp_encoder_destroy(RpEncoder* e) {
  if (!e) return;
  for (void* p : e->allocs) (void)hipFree(p);
  delete e;
}

namespace {
struct Workspace {
  bf16_t *xb, *xlo;  // the residual stream: xb = its bf16 plane (A operand of the QKV / FFN-in GEMMs), xlo = the int8 extension
                     // plane of the 24-bit form (one byte per element; x24_update2)
  bf16_t *qkv, *att, *ff;
  float* ssp;   // [Tp, ceil(D/64)] per-row partial sums of squares of x
  float* rs;    // [Tp] rsqrt(mean(x^2) + eps)
  float* pool;  // [Tp / 128 + batch, D] partial column sums of the pooling pass
  int4* work;   // [Tp / 128 + batch] attention work list of the pass (worklist_kernel)
  int4* pwork;  // [Tp / 128 + batch] pooling work list
  size_t bytes;
};

Workspace carve(const RpEncoder* e, int T, int batch, char* base) {
  const size_t Tp = align_up((size_t)T, GEMM_M_ALIGN);
  const size_t D = e->cfg.d_model, F = e->cfg.d_ff, inner = e->inner;
  Workspace w;
  size_t off = 0;
  auto take = [&](size_t bytes) {
    char* p = base ? base + off : nullptr;
    off += align_up(bytes, 256);
    return p;
  };
  w.xb = (bf16_t*)take(Tp * D * 2);
  w.xlo = (bf16_t*)take(Tp * D);  // int8 extension plane
  w.ssp = (float*)take(Tp * ((D + 63) / 64) * 4);
  w.rs = (float*)take(Tp * 4);
  w.qkv = (bf16_t*)take(Tp * 3 * inner * 2);
  w.att = (bf16_t*)take(Tp * inner * 2);
  w.ff = (bf16_t*)take(Tp * F * 2);
  w.pool = (float*)take((Tp / POOL_CHUNK_MIN + (size_t)batch + 1) * D * 4);
  w.work = (int4*)take((Tp / ATT_Q + (size_t)batch + 1) * sizeof(int4));
  w.pwork = (int4*)take((Tp / POOL_CHUNK_MIN + (size_t)batch + 1) * sizeof(int4));
  w.bytes = off;
  return w;
}
}  // namespace

extern "C" size_t rp_encoder_workspace_bytes(const RpEncoder* enc, int32_t total_tokens, int32_t batch) {
  if (!enc || total_tokens <= 0 || batch <= 0) return 0;
  return carve(enc, total_tokens, batch, nullptr).bytes;
}

// The launch sequence of one encoder pass.  T / batch size the grids; when t_dev is given (rp_encode_padded) the
// real token count is known on the device only: T is then an upper bound, kernels skip the rows beyond *t_dev.
static RpStatus encode_pass(RpEncoder* e, const int32_t* ids, const int32_t* cu_seqlens, int32_t batch, int32_t T,
                            const int32_t* t_dev, void* out, int32_t out_dtype, const Workspace& w, hipStream_t stream) {
  const RpT5Config& c = e->cfg;
  const int D = c.d_model, F = c.d_ff, inner = e->inner, H = c.num_heads;
  const int Tp = (int)align_up((size_t)T, GEMM_M_ALIGN);
  const int tv = T;  // host-side bound of the token count: picks the small-token tile configurations, sizes grids
  RpStatus st;

  const int np = (D + 63) / 64;
  // When both row-scaled projections run small tiles (passes of up to ~1000 tokens) their epilogues reduce the
  // statistic themselves (RowScaleFromSlots): two launches fewer per layer.
  const bool fused_rs = small_variant(pick_gemm_variant(RP_K_GEMM_QKV, Tp, 3 * inner, D, tv)) &&
                        small_variant(pick_gemm_variant(RP_K_GEMM_WI, Tp, 2 * F, D, tv));
  const RowScale rs{w.rs};
  const RowScaleFromSlots rs_slots{w.ssp, np, Tp, 1.f / (float)D, c.layer_norm_eps};
  const RowScaleLds rs_lds{w.ssp, np, Tp, 1.f / (float)D, c.layer_norm_eps};
  // big tiles: each of the two row-scaled projections reduces the statistic itself from LDS (no rowscale launch)
  const bool lds_qkv = !fused_rs && g_gemm_rs_lds && np <= 32 && big_variant(pick_gemm_variant(RP_K_GEMM_QKV, Tp, 3 * inner, D, tv));
  const bool lds_wi = !fused_rs && g_gemm_rs_lds && np <= 32 && big_variant(pick_gemm_variant(RP_K_GEMM_WI, Tp, 2 * F, D, tv));
  // Tail of the FFN-out launch.  1644 tiles of 256 x 256 on 256 CUs are 6.42 rounds: the seventh runs 108 tiles
  // while 148 CUs idle (70 k tokens).  Since round 5 the projection runs as a MIXED launch (gemm_kernel_mixed, option
  // gemm_mixed: the token rows beyond the whole rounds are half tiles inside the same launch) and main_rows() returns Tp.
  // With that option off this is rounds 3 - 4's form: the token rows that make whole rounds go to the big tiles in
  // one launch, the rest (18 token tiles here) runs as ONE more round - of 256 x 128 half tiles (gemm_tail_variant 30,
  // one per CU) or 128 x 128 quarter tiles (0, two per CU).  Same K-ascending chains per output element: not a bit
  // changes.  Measured 9.28 -> 9.04 ms per step in round 3; the same split of the QKV projection (5.35 rounds) gained
  // nothing (3.11 -> 3.14 ms: its short K loop leaves the tail round cheap already).
  const int n_cus = device_cu_count();
  auto main_rows = [&](int prof_class, int n_features, int K) -> int {
    if (!g_gemm_tail_split || t_dev) return Tp;  // (token count known on the device only: one launch)
    const int v = pick_gemm_variant(prof_class, Tp, n_features, K, tv);
    if (v != 20 && v != 26) return Tp;
    const int tiles_f = (n_features + 255) / 256, tiles_t = Tp / 256;
    // The mixed launch carries its own half tiles - when launch_gemm_cfg actually takes it: the 8-wave 256 x 256
    // configuration (26; the 4-wave form 20 has no edge layouts), at least two k-tiles, and a plan (plan_mixed, the SAME
    // function and CU count launch_gemm_cfg uses).  Declined there, the tail split below still applies.
    if (((g_gemm_mixed >> (prof_class - RP_K_GEMM_QKV)) & 1) && v == 26 && K >= 128 && plan_mixed(tiles_f, tiles_t, n_cus).full_rows)
      return Tp;
    int g = tiles_f, b = n_cus;  // gcd
    while (b) {
      const int t = g % b;
      g = b;
      b = t;
    }
    const int unit = n_cus / g;  // token tiles per whole number of rounds
    const int t1 = tiles_t / unit * unit, rest = tiles_t - t1;
    if (t1 == 0 || rest == 0) return Tp;
    if (rest * tiles_f > (7 * n_cus) / 10) return Tp;                      // the last round is nearly full anyway
    if (g_gemm_tail_variant == 30 ? rest * 2 * tiles_f > n_cus             // the half tiles (one per CU) ...
                                  : rest * 2 * ((n_features + 127) / 128) > 2 * n_cus)  // ... the quarter tiles (two per CU)
      return Tp;                                                           // would not fit one round
    return t1 * 256;
  };
  const int wo_main = main_rows(RP_K_GEMM_WO, D, F);
  auto launch_rowscale = [&](bool needed_anyway = false) {
    if (fused_rs && !needed_anyway) return;
    ProfScope ps(stream, RP_K_RMSNORM);
    hipLaunchKernelGGL(rowscale_kernel, dim3((Tp + 63) / 64), dim3(64), 0, stream, w.ssp, w.rs, Tp, np,
                       1.f / (float)D, c.layer_norm_eps);
  };
  // the first QKV projection's RMSNorm factor straight from the embedding kernel (the table carries every row's sum of
  // squares) unless a consumer reads the statistic slots itself (few-token passes, the LDS form)
  const bool embed_rs = !fused_rs && !lds_qkv;
  {
    ProfScope ps(stream, RP_K_EMBED);
    // four token rows per wave, all their loads in flight before the first store (round 6, exp10: 66 -> 57 us at 70 k tokens;
    // one row per wave 66, two 61 - 63, eight 61).  Few tokens (a single proof state: 8 workgroups of four rows per wave
    // took 11.4 us, 32 of one row 5.2): one row per wave, as many workgroups as there are rows to spread.
    auto launch_embed = [&](auto kern, int rows_per_wave) {
      hipLaunchKernelGGL(kern, dim3((Tp + 4 * rows_per_wave - 1) / (4 * rows_per_wave)), dim3(256), 0, stream, ids,
                         (const bf16_t*)e->embed_hi, (const uint8_t*)e->embed_lo, (const float*)e->embed_ss, w.xb,
                         reinterpret_cast<uint8_t*>(w.xlo), w.ssp, np, embed_rs ? w.rs : (float*)nullptr, 1.f / (float)D,
                         c.layer_norm_eps, T, Tp, D, c.vocab_size, t_dev);
    };
    const bool few = Tp < 8192;
    if (D <= 1536) {
      if (few) launch_embed(embed_copy_kernel<3, 1>, 1);
      else launch_embed(embed_copy_kernel<3, 4>, 4);
    } else {
      if (few) launch_embed(embed_copy_kernel<4, 1>, 1);
      else launch_embed(embed_copy_kernel<4, 4>, 4);
    }
  }
  RP_CHECK_LAUNCH();
  const dim3 att_grid(H, T / ATT_Q + batch);  // upper bound of the number of 128-query blocks
  const int pc = g_pool_chunk;
  hipLaunchKernelGGL(worklist_kernel, dim3(1), dim3(1024), 0, stream, cu_seqlens, batch, w.work, (int)att_grid.y, w.pwork,
                     T / pc + batch, pc);
  for (int i = 0; i < c.num_layers; ++i) {
    const LayerPacked& L = e->layers[i];
    // attention sub-layer: qkv = rs * (xb Wqkv'^T)  ->  attention  ->  x += att Wo^T  (+ xb, ssp refreshed)
    if (!lds_qkv && !(i == 0 && embed_rs)) launch_rowscale();
    st = fused_rs ? launch_gemm<true>(w.xb, D, Tp, L.wqkv, D, 3 * inner, D,
                                      EpiStoreBf16Slots{w.qkv, 3 * inner, 3 * inner, rs_slots}, stream, RP_K_GEMM_QKV, tv, t_dev)
         : lds_qkv ? launch_gemm_big(w.xb, D, Tp, L.wqkv, D, 3 * inner, D,
                                     EpiStoreBf16Lds{{w.qkv, 3 * inner, 3 * inner, rs_lds}}, stream, RP_K_GEMM_QKV, tv, t_dev)
                   : launch_gemm(w.xb, D, Tp, L.wqkv, D, 3 * inner, D, EpiStoreBf16{w.qkv, 3 * inner, 3 * inner, rs}, stream,
                                 RP_K_GEMM_QKV, tv, t_dev);
    if (st) return st;
    {
      ProfScope ps(stream, RP_K_ATTENTION);
      hipLaunchKernelGGL(attention_kernel<false>, att_grid, dim3(256), 0, stream, w.qkv, (const int4*)w.work, e->bias_tab, w.att,
                         H, e->maxd, (float*)nullptr, 0, Drop{0u, 0u, 1.f}, 0u);
    }
    if ((st = launch_gemm(w.att, inner, Tp, L.wo, inner, D, inner, EpiResid8{w.xb, w.xlo, D, D, w.ssp, np, Tp}, stream,
                          RP_K_GEMM_O, tv, t_dev)))
      return st;
    if (g_debug_skip_ffn) continue;
    if (!lds_wi) launch_rowscale();
    // feed-forward sub-layer: ff = gelu(rs * g) * (rs * u)  ->  x += ff Wo2^T  (+ xb, ssp refreshed)
    st = fused_rs ? launch_gemm<true>(w.xb, D, Tp, L.wi, D, 2 * F, D, EpiGegluBf16Slots{w.ff, F, 2 * F, rs_slots}, stream,
                                      RP_K_GEMM_WI, tv, t_dev)
         : lds_wi ? launch_gemm_big(w.xb, D, Tp, L.wi, D, 2 * F, D, EpiGegluBf16Lds{{w.ff, F, 2 * F, rs_lds}}, stream,
                                    RP_K_GEMM_WI, tv, t_dev)
                  : launch_gemm(w.xb, D, Tp, L.wi, D, 2 * F, D, EpiGegluBf16{w.ff, F, 2 * F, rs}, stream, RP_K_GEMM_WI, tv,
                                t_dev);
    if (st) return st;
    if (wo_main < Tp) {
      const int r1 = wo_main;
      st = launch_gemm(w.ff, F, r1, L.wo2, F, D, F, EpiResid8{w.xb, w.xlo, D, D, w.ssp, np, Tp}, stream, RP_K_GEMM_WO);
      if (st) return st;
      st = launch_gemm(w.ff + (size_t)r1 * F, F, Tp - r1, L.wo2, F, D, F,
                       EpiResid8{w.xb + (size_t)r1 * D, (bf16_t*)((uint8_t*)w.xlo + (size_t)r1 * D), D, D, w.ssp + r1, np, Tp}, stream,
                       RP_K_GEMM_WO, std::max(1, tv - r1), nullptr, g_gemm_tail_variant);
    } else {
      st = launch_gemm(w.ff, F, Tp, L.wo2, F, D, F, EpiResid8{w.xb, w.xlo, D, D, w.ssp, np, Tp}, stream, RP_K_GEMM_WO, tv,
                       t_dev);
    }
    if (st) return st;
  }
  launch_rowscale(true);  // final RMSNorm statistic (the pooling pass reads rs per token row)
  {
    ProfScope ps(stream, RP_K_POOL);
    launch_pool_partial(dim3(T / pc + batch), stream, w.xb, w.xlo, w.rs, (const int4*)w.pwork, w.pool, D, pc, e->final_ln, out,
                        out_dtype == RP_DT_BF16 ? 1 : 0, 1, true);
    hipLaunchKernelGGL(pool_finish_kernel, dim3(batch), dim3(256), 0, stream, w.pool, e->final_ln, cu_seqlens, out,
                       out_dtype == RP_DT_BF16 ? 1 : 0, D, pc, 1);
  }
  RP_CHECK_LAUNCH();
  return RP_OK;
}

extern "C" RpStatus rp_encode_varlen(RpEncoder* e, const int32_t* ids, const int32_t* cu_seqlens, int32_t batch,
                                     int32_t T, int32_t max_len, void* out, int32_t out_dtype, void* workspace,
                                     size_t workspace_bytes, void* stream_) {
  RP_REQUIRE(e && ids && cu_seqlens && out, "null argument");
  RP_REQUIRE(batch > 0 && T > 0 && max_len > 0 && max_len <= T, "batch=%d total_tokens=%d max_len=%d", batch, T,
             max_len);
  RP_REQUIRE(out_dtype == RP_DT_F32 || out_dtype == RP_DT_BF16, "out_dtype");
  Workspace w = carve(e, T, batch, (char*)workspace);
  if (!workspace || workspace_bytes < w.bytes)
    return fail(RP_E_WORKSPACE, "workspace %zu < required %zu bytes", workspace_bytes, w.bytes);
  return encode_pass(e, ids, cu_seqlens, batch, T, nullptr, out, out_dtype, w, (hipStream_t)stream_);
}

// ------------------------------------------------------------------------------------------
// rp_encode_padded: the reference's call form, _encode(input_ids [B, L], attention_mask [B, L]) (model.py:92-114;
// the batches come from datamodule.py:130-144, right-padded by the tokenizer).  Three small kernels turn the
// padded batch into the packed form WITHOUT a host round trip: per-row lengths + right-padding check, an
// exclusive scan to cu_seqlens, id compaction (int64 -> int32).  The encoder pass is then launched for the upper
// bound B * L of the token count and skips, on the device, what lies beyond the real count.
// ------------------------------------------------------------------------------------------
namespace rp {
// lens[b] = number of non-zero mask entries; the row is right-padded iff that equals (last non-zero index + 1).
// A row that is not right-padded, or empty, is reported as -(count + 1): padded_scan_kernel turns that into meta[2].
// (No memset node: meta is written by plain stores only.  hipMemsetAsync on a 4-byte-aligned address inside a captured
// graph left 0x54545454 in one of the four words on replay - the low byte of the destination address.)
__global__ __launch_bounds__(256) void padded_lens_kernel(const int64_t* __restrict__ mask, int L,
                                                          int32_t* __restrict__ lens) {
  __shared__ int s_cnt[4], s_last[4];
  const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t* row = mask + (size_t)b * L;
  int cnt = 0, last = 0;
  for (int i = threadIdx.x; i < L; i += 256)
    if (row[i] != 0) {
      ++cnt;
      last = i + 1;  // i increases along the loop: the thread's largest
    }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    cnt += __shfl_xor(cnt, o, 64);
    last = max(last, __shfl_xor(last, o, 64));
  }
  if (lane == 0) {
    s_cnt[wave] = cnt;
    s_last[wave] = last;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    cnt = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
    last = max(max(s_last[0], s_last[1]), max(s_last[2], s_last[3]));
    lens[b] = (cnt != last || cnt == 0) ? -(cnt + 1) : cnt;  // not right-padded, or an empty sequence
  }
}

// cu[0] = 0, cu[b + 1] = sum of lens[0..b]; meta = {total, longest, any row rejected, 0}.  One workgroup of 1024 threads.
__global__ __launch_bounds__(1024) void padded_scan_kernel(const int32_t* __restrict__ lens_in, int B,
                                                           int32_t* __restrict__ cu, int32_t* __restrict__ meta) {
  __shared__ int s_part[1024];
  const int tid = threadIdx.x;
  const int per = (B + 1023) / 1024;
  const int lo = min(tid * per, B), hi = min(lo + per, B);
  auto len_of = [&](int i) {  // a rejected row keeps its count (the pass still runs; the caller raises afterwards)
    const int v = lens_in[i];
    return v < 0 ? -v - 1 : v;
  };
  int sum = 0, mx = 0, bad = 0;
  for (int i = lo; i < hi; ++i) {
    bad |= lens_in[i] < 0;
    sum += len_of(i);
    mx = max(mx, len_of(i));
  }
  const int any_bad = __syncthreads_or(bad);
  s_part[tid] = sum;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {  // Hillis-Steele inclusive scan of the per-thread sums
    const int v = (tid >= o) ? s_part[tid - o] : 0;
    __syncthreads();
    s_part[tid] += v;
    __syncthreads();
  }
  int run = s_part[tid] - sum;  // exclusive prefix of this thread's chunk
  const int total = s_part[1023];
  if (tid == 0) cu[0] = 0;
  for (int i = lo; i < hi; ++i) {
    run += len_of(i);
    cu[i + 1] = run;
  }
  __syncthreads();  // everyone has read its prefix and the total: s_part is reused for the maximum
  s_part[tid] = mx;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if (tid < o) s_part[tid] = max(s_part[tid], s_part[tid + o]);
    __syncthreads();
  }
  if (tid == 0) {
    meta[0] = total;
    meta[1] = s_part[0];
    meta[2] = any_bad ? 1 : 0;
    meta[3] = 0;
  }
}

__global__ __launch_bounds__(256) void padded_pack_kernel(const int64_t* __restrict__ ids, const int32_t* __restrict__ cu,
                                                          int L, int32_t* __restrict__ packed) {
  const int b = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
  const int s0 = cu[b], len = cu[b + 1] - s0;
  if (i < len) packed[s0 + i] = (int32_t)ids[(size_t)b * L + i];
}

// The three kernels above as ONE launch for a single row (the prover's retrieve(): every dependent launch of that path
// costs 4-5 us as a graph node): the same lens / cu / meta / packed values.
__global__ __launch_bounds__(1024) void padded_single_kernel(const int64_t* __restrict__ mask, const int64_t* __restrict__ ids,
                                                             int L, int32_t* __restrict__ lens, int32_t* __restrict__ cu,
                                                             int32_t* __restrict__ packed, int32_t* __restrict__ meta) {
  __shared__ int s_cnt[16], s_last[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int cnt = 0, last = 0;
  for (int i = tid; i < L; i += 1024)
    if (mask[i] != 0) {
      ++cnt;
      last = i + 1;
    }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    cnt += __shfl_xor(cnt, o, 64);
    last = max(last, __shfl_xor(last, o, 64));
  }
  if (lane == 0) {
    s_cnt[wave] = cnt;
    s_last[wave] = last;
  }
  __syncthreads();
  cnt = last = 0;
#pragma unroll
  for (int w = 0; w < 16; ++w) {
    cnt += s_cnt[w];
    last = max(last, s_last[w]);
  }
  const bool bad = cnt != last || cnt == 0;
  if (tid == 0) {
    lens[0] = bad ? -(cnt + 1) : cnt;
    cu[0] = 0;
    cu[1] = cnt;
    meta[0] = cnt;
    meta[1] = cnt;
    meta[2] = bad ? 1 : 0;
    meta[3] = 0;
  }
  for (int i = tid; i < cnt; i += 1024) packed[i] = (int32_t)ids[i];
}

struct PaddedPrep {
  int32_t *lens, *cu, *packed;
  size_t bytes;
};
static PaddedPrep carve_padded(int B, int L, char* base) {
  PaddedPrep p;
  size_t off = 0;
  auto take = [&](size_t bytes) {
    char* q = base ? base + off : nullptr;
    off += align_up(bytes, 256);
    return q;
  };
  p.lens = (int32_t*)take((size_t)B * 4);
  p.cu = (int32_t*)take((size_t)(B + 1) * 4);
  p.packed = (int32_t*)take((size_t)B * L * 4);
  p.bytes = off;
  return p;
}
}  // namespace rp

extern "C" size_t rp_encode_padded_workspace_bytes(const RpEncoder* enc, int32_t batch, int32_t padded_len) {
  if (!enc || batch <= 0 || padded_len <= 0) return 0;
  return carve_padded(batch, padded_len, nullptr).bytes + carve(enc, batch * padded_len, batch, nullptr).bytes;
}

extern "C" RpStatus rp_encode_padded(RpEncoder* e, const int64_t* input_ids, const int64_t* attention_mask,
                                     int32_t batch, int32_t padded_len, void* out, int32_t out_dtype, int32_t* meta,
                                     void* workspace, size_t workspace_bytes, void* stream_) {
  RP_REQUIRE(e && input_ids && attention_mask && out && meta, "null argument");
  RP_REQUIRE(batch > 0 && padded_len > 0 && (int64_t)batch * padded_len < (1ll << 30), "batch=%d padded_len=%d", batch,
             padded_len);
  RP_REQUIRE(out_dtype == RP_DT_F32 || out_dtype == RP_DT_BF16, "out_dtype");
  hipStream_t stream = (hipStream_t)stream_;
  const int T_max = batch * padded_len;
  PaddedPrep pp = carve_padded(batch, padded_len, (char*)workspace);
  Workspace w = carve(e, T_max, batch, workspace ? (char*)workspace + pp.bytes : nullptr);
  if (!workspace || workspace_bytes < pp.bytes + w.bytes)
    return fail(RP_E_WORKSPACE, "workspace %zu < required %zu bytes", workspace_bytes, pp.bytes + w.bytes);
  if (batch == 1) {
    hipLaunchKernelGGL(padded_single_kernel, dim3(1), dim3(1024), 0, stream, attention_mask, input_ids, padded_len, pp.lens,
                       pp.cu, pp.packed, meta);
  } else {
    hipLaunchKernelGGL(padded_lens_kernel, dim3(batch), dim3(256), 0, stream, attention_mask, padded_len, pp.lens);
    hipLaunchKernelGGL(padded_scan_kernel, dim3(1), dim3(1024), 0, stream, pp.lens, batch, pp.cu, meta);
    hipLaunchKernelGGL(padded_pack_kernel, dim3((padded_len + 255) / 256, batch), dim3(256), 0, stream, input_ids, pp.cu,
                       padded_len, pp.packed);
  }
  RP_CHECK_LAUNCH();
  return encode_pass(e, pp.packed, pp.cu, batch, T_max, meta /* meta[0] = token count */, out, out_dtype, w, stream);
}

// ------------------------------------------------------------------------------------------
// kernel-level test entry points
// ------------------------------------------------------------------------------------------
extern "C" RpStatus rp_dbg_gemm(const void* A, const void* W, void* out, int32_t M, int32_t N, int32_t K,
                                int32_t n_valid, int32_t epilogue, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  const bf16_t* a = (const bf16_t*)A;
  const bf16_t* w = (const bf16_t*)W;
  switch (epilogue) {
    case RP_EPI_STORE_BF16:
      return launch_gemm(a, K, M, w, K, N, K, EpiStoreBf16{(bf16_t*)out, n_valid, n_valid, RowScale{nullptr}}, stream,
                         RP_K_GEMM_QKV);
    case RP_EPI_RESID:  // out = the two planes of the residual stream, [2, M, n_valid] bf16 (hi, then lo)
      RP_REQUIRE(n_valid % 8 == 0, "n_valid=%d", n_valid);
      return launch_gemm(a, K, M, w, K, N, K,
                         EpiResid{(bf16_t*)out, (bf16_t*)out + (size_t)M * n_valid, n_valid, n_valid, nullptr, 0, 0}, stream,
                         RP_K_GEMM_WO);
    case RP_EPI_RESID8:  // out = the bf16 plane [M, n_valid], then the int8 extension plane [M, n_valid] (the 24-bit form)
      RP_REQUIRE(n_valid % 8 == 0, "n_valid=%d", n_valid);
      return launch_gemm(a, K, M, w, K, N, K,
                         EpiResid8{(bf16_t*)out, (bf16_t*)out + (size_t)M * n_valid, n_valid, n_valid, nullptr, 0, 0}, stream,
                         RP_K_GEMM_WO);
    case RP_EPI_GEGLU_BF16:
      return launch_gemm(a, K, M, w, K, N, K, EpiGegluBf16{(bf16_t*)out, n_valid / 2, n_valid, RowScale{nullptr}}, stream,
                         RP_K_GEMM_WI);
  }
  return fail(RP_E_INVALID, "unknown epilogue %d", epilogue);
}

extern "C" RpStatus rp_dbg_gemm_fused(const void* A, const void* W, void* out, int32_t M, int32_t N, int32_t K,
                                      int32_t n_valid, int32_t epilogue, const float* ssp_in, int32_t np_in,
                                      float inv_d, float eps, void* xb_out, float* ssp_out, int32_t np_out,
                                      void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  const bf16_t* a = (const bf16_t*)A;
  const bf16_t* w = (const bf16_t*)W;
  float* rs_buf = nullptr;
  if (ssp_in) {  // test entry only: a scratch allocation is fine here
    RP_HIP(hipMalloc((void**)&rs_buf, (size_t)M * 4));
    hipLaunchKernelGGL(rowscale_kernel, dim3((M + 63) / 64), dim3(64), 0, stream, ssp_in, rs_buf, M, np_in, inv_d, eps);
  }
  const RowScale rs{rs_buf};
  struct Free {
    float* p;
    hipStream_t s;
    ~Free() {
      if (p) {
        (void)hipStreamSynchronize(s);
        (void)hipFree(p);
      }
    }
  } free_rs{rs_buf, stream};
  switch (epilogue) {
    case RP_EPI_STORE_BF16:
      return launch_gemm(a, K, M, w, K, N, K, EpiStoreBf16{(bf16_t*)out, n_valid, n_valid, rs}, stream, RP_K_GEMM_QKV);
    case RP_EPI_RESID:  // out = [2, M, n_valid] bf16 planes (hi, lo); xb_out unused (hi is the operand copy)
      RP_REQUIRE(n_valid % 8 == 0, "n_valid=%d", n_valid);
      return launch_gemm(a, K, M, w, K, N, K,
                         EpiResid{(bf16_t*)out, (bf16_t*)out + (size_t)M * n_valid, n_valid, n_valid, ssp_out, np_out, M},
                         stream, RP_K_GEMM_WO);
    case RP_EPI_RESID8:
      RP_REQUIRE(n_valid % 8 == 0, "n_valid=%d", n_valid);
      return launch_gemm(a, K, M, w, K, N, K,
                         EpiResid8{(bf16_t*)out, (bf16_t*)out + (size_t)M * n_valid, n_valid, n_valid, ssp_out, np_out, M},
                         stream, RP_K_GEMM_WO);
    case RP_EPI_GEGLU_BF16:
      return launch_gemm(a, K, M, w, K, N, K, EpiGegluBf16{(bf16_t*)out, n_valid / 2, n_valid, rs}, stream,
                         RP_K_GEMM_WI);
  }
  return fail(RP_E_INVALID, "unknown epilogue %d", epilogue);
}

// Matrix-pipe rate of this box under load: MFMAs from registers only (include/reprover_hip.h).
namespace rp {
__global__ __launch_bounds__(256) void mfma_probe_kernel(int mfmas, float* __restrict__ sink) {
  const uint32_t seed = (uint32_t)(blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
  bf16x8 a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int e = 0; e < 8; ++e) {  // bf16 values in [-2, 2) with random mantissas and signs
      const uint32_t h = (seed + 40503u * (uint32_t)(i * 8 + e)) * 2246822519u;
      a[i][e] = (short)(0x3f00 | (h & 0x80ff) | ((h >> 9) & 0x0080));
      b[i][e] = (short)(0x3f00 | ((h >> 16) & 0x80ff) | ((h >> 3) & 0x0080));
    }
  f32x16 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  for (int it = 0; it < mfmas; it += 32) {  // (operand registers indexed by compile-time constants only)
#pragma unroll
    for (int rot = 0; rot < 4; ++rot)
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i & 3], b[(i + rot) & 3], acc[i], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][15];
  if (s == 12345.678f) sink[0] = s;  // (never true: keeps the accumulators live)
}
}  // namespace rp
extern "C" RpStatus rp_dbg_mfma_probe(int32_t waves_per_cu, int32_t mfmas, float* sink, void* stream_) {
  RP_REQUIRE(sink && waves_per_cu >= 4 && waves_per_cu % 4 == 0 && mfmas >= 32 && mfmas % 32 == 0, "waves_per_cu=%d mfmas=%d",
             waves_per_cu, mfmas);
  hipLaunchKernelGGL(mfma_probe_kernel, dim3(256 * (waves_per_cu / 4)), dim3(256), 0, (hipStream_t)stream_, mfmas, sink);
  RP_CHECK_LAUNCH();
  return RP_OK;
}

extern "C" RpStatus rp_dbg_rowscale(const float* ssp, float* rs, int32_t rows, int32_t np, float inv_d, float eps,
                                    void* stream_) {
  RP_REQUIRE(ssp && rs && rows > 0 && np > 0 && np <= 32, "rows=%d np=%d", rows, np);
  hipLaunchKernelGGL(rowscale_kernel, dim3((rows + 63) / 64), dim3(64), 0, (hipStream_t)stream_, ssp, rs, rows, np, inv_d,
                     eps);
  RP_CHECK_LAUNCH();
  return RP_OK;
}

extern "C" RpStatus rp_dbg_attention(const void* qkv, const int32_t* cu, const float* bias_tab, void* out,
                                     int32_t batch, int32_t max_len, int32_t H, int32_t rows_total, void* stream_) {
  const int maxd = 128;
  (void)max_len;
  hipStream_t stream = (hipStream_t)stream_;
  const dim3 grid(H, rows_total / ATT_Q + batch);  // rows_total >= the packed token count
  // test entry only: a stream-ordered scratch allocation is fine here (attention list | pooling list)
  const int n_p = rows_total / POOL_CHUNK + batch;
  int4* work = nullptr;
  RP_HIP(hipMallocAsync((void**)&work, ((size_t)grid.y + n_p) * sizeof(int4), stream));
  hipLaunchKernelGGL(worklist_kernel, dim3(1), dim3(1024), 0, stream, cu, batch, work, (int)grid.y, work + grid.y, n_p, POOL_CHUNK);
  hipLaunchKernelGGL(attention_kernel<false>, grid, dim3(256), 0, stream, (const bf16_t*)qkv, (const int4*)work, bias_tab,
                     (bf16_t*)out, H, maxd, (float*)nullptr, 0, Drop{0u, 0u, 1.f}, 0u);
  const hipError_t le = hipGetLastError();
  (void)hipFreeAsync(work, stream);
  if (le != hipSuccess) return rp::fail(RP_E_HIP, "attention launch failed: %s", hipGetErrorString(le));
  return RP_OK;
}

#ifdef RP_PHASE_PROBE  // probe builds only (tools/probes/gemm_phase.py): the host-side readers of the phase timestamps
#include "probes/rp_probe_exports.h"
#endif
