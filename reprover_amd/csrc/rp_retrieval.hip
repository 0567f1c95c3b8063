// Similarity scan + accessibility mask + exact top-k on gfx950.
//
// Replaces (reference lean-dojo/ReProver) common.py:299-326 Corpus.get_nearest_premises:
//   similarities = Q @ E.T (:307); argsort(descending).tolist() (:308); per-query Python walk over
//   the sorted ids keeping premises in get_accessible_premises(path, pos) (:312-322, :280-289).
//
// Design (HBM-bound scan: E is N x D bf16, read once):
//   * scan kernel = the bf16 MFMA GEMM core (rp_gemm.h) with A = Q (queries, L2-resident) and
//     W = E (premises, streamed); the epilogue turns each fp32 score into a 64-bit sortable key
//     (ordered(score) << 32 | ~id) after applying the accessibility predicate, so "masked top-k"
//     is exact: the mask is applied BEFORE selection, ties break toward the lower id.
//   * the B x N score matrix is never materialised.  Pass 1 scans every `stride`-th premise tile
//     and writes its keys densely; a radix-select finds each query's k best of that sample,
//     whose k-th key is a valid lower bound for the global k-th key.  Pass 2 scans the remaining
//     tiles and appends only keys above the bound to a per-query candidate list (expected size
//     ~ k * stride).  A final radix-select + bitonic sort over <= cap candidates yields the
//     sorted top-k.  Small shards (N <= 16384) and RP_TOPK_DENSE take the single dense pass.
#include <cmath>
#include "rp_gemm.h"

namespace rp {

constexpr int SIM_DENSE_MAX_N = 16384;
constexpr int64_t SIM_DENSE_MAX_KEYS = 16 << 20;  // B * N above which even a small shard is scanned in two passes
constexpr int SIM_STRIDE_MAX = 64;
constexpr int SIM_MAX_K = 1024;
constexpr int SIM_COUNT_STRIDE = 32;  // one candidate counter per 128-B line: contended atomics of different
                                      // queries must not serialise in the same L2 line
constexpr int SIM_PB = 256;           // two-pass plan: premises are sampled / skipped in blocks of 256 rows
// Dense single-pass configurations (small shards, RP_TOPK_DENSE): BM queries x 128 premises, queries are the
// MFMA row operand, premises the column operand (lane = premise: dense key rows are written coalesced).
typedef GemmCfg<256, 128, 32, 4, 2, 3> SimCfgQ256;  // B > 128: 8 waves, one workgroup sees up to 256 queries
typedef GemmCfg<128, 128, 64, 2, 2, 2> SimCfgQ128;  // B <= 128, D % 64 == 0
typedef GemmCfg<128, 128, 32, 2, 2, 3> SimCfgQ128K32;  // B <= 128, D % 32 == 0
// e4m3 index (rp_sim_topk_fp8): 64 fp8 values per K-step = the 64-B rows of the "BK = 32" geometry
typedef GemmCfg<256, 128, 32, 4, 2, 3, 0, 1> SimCfg8Q256;
typedef GemmCfg<128, 128, 32, 2, 2, 3, 0, 1> SimCfg8Q128;
// Sample pass of the two-pass plan: 1/stride of the rows, as many small workgroups as possible (128 queries x
// 64 premises, 6-deep ring: every workgroup is a latency-bound K loop, so depth and count are what matter).
typedef GemmCfg<128, 64, 32, 2, 2, 6> SimCfgSample;
typedef GemmCfg<128, 64, 32, 2, 2, 6, 0, 1> SimCfg8Sample;
// Filter pass: the encoder's software-pipelined 256 x 256 x 64 tile (one wave per SIMD, 128 x 128 per wave) with
// PREMISES as the MFMA row operand and QUERIES as the column operand: a lane then owns ONE query per column
// fragment, so the per-query bound sits in 4 registers and the pre-test is one v_cmp per score (EpiSimFilter).
typedef GemmCfg<256, 256, 64, 2, 2, 2, 1> SimCfgFilter;
typedef GemmCfg<256, 256, 64, 2, 2, 2, 1, 1> SimCfg8Filter;
// the same for rows that end half a k-tile early (D = 1472 e4m3 bytes = 11.5 tiles of 128): GemmCfg::KTAIL
typedef GemmCfg<256, 256, 64, 2, 2, 2, 1, 1, 0, 1> SimCfg8FilterTail;
// the same tile with 32-wide K slices in a 4-deep ring: three slices (48 KB of premises) in flight per CU instead of
// one (32 KB) - the premise operand streams from HBM, not from L2 like an encoder GEMM's weights
typedef GemmCfg<256, 256, 32, 2, 2, 4, 1> SimCfgFilterK32;
typedef GemmCfg<128, 64, 64, 2, 2, 3> SimCfgSampleK64;
// Round 6: the plain-loop tiles of this kernel run on EIGHT waves where their DMA split allows it (4 x 2: 32 queries per wave).
// One wave per SIMD waited out every LDS round trip of its k-steps by itself (~900 cycles per 64-wide k-tile for 256 of MFMA);
// with two per SIMD one multiplies while the other waits: sample pass at C2 23.3 -> 20.6 us, the first-generation filter pass
// 90.7 -> 83.9 us for a single query and 129 -> 109 us at 64 queries, the same bits (profiles/r06_raw/exp28*).  The four-wave
// forms stay selectable in probe builds (scan_waves = 4).
typedef GemmCfg<128, 128, 64, 4, 2, 2> SimCfgQ128W8;
typedef GemmCfg<128, 128, 32, 4, 2, 3> SimCfgQ128K32W8;
typedef GemmCfg<128, 128, 32, 4, 2, 3, 0, 1> SimCfg8Q128W8;
typedef GemmCfg<128, 64, 64, 4, 2, 3> SimCfgSampleK64W8;
typedef GemmCfg<128, 64, 64, 4, 2, 3, 0, 1> SimCfg8SampleK64W8;            // e4m3 rows of whole 128-byte k-tiles
typedef GemmCfg<128, 64, 64, 4, 2, 3, 0, 1, 0, 1> SimCfg8SampleK64TailW8;  // ... ending half a k-tile early (1472 bytes)
// At most 32 queries (a single proof state: retrieve()): 32 queries x 128 premises.  The 128-query tile spent half of every stage's
// LDS-DMA on a query panel of which one row is a query, and kept one 16-KB premise slice per workgroup in flight - 32 KB per CU
// where HBM at its latency wants ~64 (8 TB/s x 2 us / 256 CUs).  Here a stage is 4 KB of queries + 16 KB of premises, the ring is
// four deep and two workgroups share a CU: ~96 KB of the premise stream in flight per CU.  Same MFMA, same K order: the same bits.
typedef GemmCfg<32, 128, 64, 1, 4, 4> SimCfgQ32;
typedef GemmCfg<32, 128, 64, 1, 4, 4, 0, 1> SimCfg8Q32;            // e4m3 rows of whole 128-byte k-tiles
typedef GemmCfg<32, 128, 64, 1, 4, 4, 0, 1, 0, 1> SimCfg8Q32Tail;  // ... ending half a k-tile early (1472 bytes)
// the sample pass of such a call: 32 queries x 64 premises on two waves, six stages of 12 KB
typedef GemmCfg<32, 64, 64, 1, 2, 6> SimCfgSampleQ32;
typedef GemmCfg<32, 64, 64, 1, 2, 6, 0, 1> SimCfg8SampleQ32;
typedef GemmCfg<32, 64, 64, 1, 2, 6, 0, 1, 0, 1> SimCfg8SampleQ32Tail;
// 33 .. 64 queries (the reference's evaluation batch, datamodule.py's eval_batch_size 64): 64 queries x 128 premises on eight waves,
// 24-KB stages three deep, two workgroups per CU - 64 KB of premises in flight per CU where the 128-query tile keeps 32
typedef GemmCfg<64, 128, 64, 2, 4, 3> SimCfgQ64;
typedef GemmCfg<64, 128, 64, 2, 4, 3, 0, 1> SimCfg8Q64;
typedef GemmCfg<64, 128, 64, 2, 4, 3, 0, 1, 0, 1> SimCfg8Q64Tail;
typedef GemmCfg<32, 128, 64, 1, 4, 3> SimCfgQ32S3;   // (probe builds: three stages)
typedef GemmCfg<32, 256, 64, 1, 4, 4> SimCfgQ32P256;  // (probe builds: 256 premises per tile, one workgroup per CU)
// the sample pass when MANY queries share few rows (its 256 x 128 tiles fill the chip): MFMA-bound there, on the pipelined loop
typedef GemmCfg<256, 128, 64, 4, 2, 2, 1> SimCfgSampleBig;
// default: the premise stream (read once, by one CU) carries the nt cache policy: -7 % filter time
typedef GemmCfg<256, 256, 64, 2, 2, 2, 1, 0, 2> SimCfgFilterNt;
// the same tile on EIGHT waves, 2 x 4 (128 premises x 64 queries per wave: the epilogue's runs - 64 scores per lane and query,
// part = 2 x premise half + lane half - do not change): two waves per SIMD, one multiplies while the other waits
typedef GemmCfg<256, 256, 64, 2, 4, 2, 1, 0, 2> SimCfgFilterNt8;
// The e4m3 tile on eight waves runs the PLAIN loop: the pipelined loop's double-buffered 32-byte fragments do not fit 256
// registers beside 128 accumulators (192 bytes of scratch), one fragment set does (188 registers) - and with two waves per SIMD
// the compiler-scheduled loop beats the hand-pipelined four-wave one: filter pass at 256 queries x 130 k rows of 1536 e4m3
// bytes 76.7 -> 64.1 us, x 1 M rows 545 -> 487 us (configs[4]), 2048 queries x 16 k rows 65 -> 57 us, the same bits
// (profiles/r06_raw/exp29*); rows that end half a k-tile early (1472 bytes): SimCfg8FilterTailW8, the same half-tile handling
// as the pipelined loop's (GemmCfg::KTAIL).
typedef GemmCfg<256, 256, 64, 2, 4, 2, 0, 1> SimCfg8FilterW8;
typedef GemmCfg<256, 256, 64, 2, 4, 2, 0, 1, 0, 1> SimCfg8FilterTailW8;  // rows that end half a k-tile early (1472 e4m3 bytes)
constexpr int SIM_FILTER_META_BYTES = 5120;  // per-tile metadata behind the ring (EpiSimFilter::prologue)
int g_scan_cfg = 0;   // 0: auto; 1: force 128-query tiles in the dense path
int g_scan_impl = 0;  // 0: auto (pipelined filter kernel when the shape allows); 1: first-generation filter kernel
int g_scan_impl_force_new = 0;  // experiments: second-generation filter for every batch size
int g_scan_filter_cfg = 0;   // experiments: 0 = 8 waves, nt premise stream (default), 1 = 256x256x32 4-stage, 2 = 4 waves, default cache policy,
                             // 4 = 4 waves (the default until round 6)
int g_scan_small_tiles = 1;  // tests / A-B runs: 0 = calls of at most 64 queries on the 128-query sample and filter tiles, as before round 6
int g_scan_waves = 8;        // experiments: 4 = the four-wave forms of the plain-loop tiles
int g_scan_sample_cfg = 0;   // experiments: 0 = 128x64x64 3-stage when D % 64 == 0, 32x64x64 at <= 32 queries (default), 1 = 128x64x32 6-stage
                             // (a 6-stage 64-wide ring - five slices in flight - measured the same 23.5 us: not the depth)
int g_scan_stride = 0;       // experiments: > 0 overrides the sampling stride (power of two)
int g_scan_cap = 0;          // tests: > 0 overrides the candidate-list capacity (forces the overflow -> dense contract)
int g_scan_no_epilogue = 0;  // timing only: the filter pass drops every score (main loop in isolation)

__device__ __forceinline__ uint64_t make_key(float score, int32_t id) {
  return ((uint64_t)f2ord(score) << 32) | (uint32_t)(~(uint32_t)id);
}

struct EpiSim {
  // premise side (NULL file_of => no accessibility mask)
  const int32_t* file_of;
  const int64_t* end_key;
  int N;
  // query side
  const uint32_t* bits_t;  // [F, bits_words]
  int bits_words;
  const int32_t* own_file;
  const int64_t* q_key;
  int B;
  int id_offset;
  // e4m3 operands: score = acc * q_scale[query] * e_scale[premise] (NULL: bf16 operands, score = acc)
  const float* q_scale;
  const float* e_scale;
  // paging (rp_sim_topk_after): only keys strictly BELOW key(after_score[q], after_id[q]) qualify (NULL: no bound)
  const float* after_score;
  const int32_t* after_id;
  // output
  int filter;          // 0: dense write, 1: append keys > thr
  uint64_t* dense;     // [B, dense_ld]
  size_t dense_ld;
  int slot_shift;      // slot = premise_row + slot_shift (set per workgroup)
  const uint64_t* thr; // [B]
  uint64_t* cand;      // [B, cap]
  int cap;
  int32_t* count;      // [B]
  char* smem;          // GEMM LDS, free once the main loop is done
  int tile_q0;         // first query of this workgroup's tile (set per workgroup)
  int bm;              // queries per workgroup tile
  int debug_skip;      // timing only: 64 = the dense (sample) pass writes nothing

  // Accessibility predicate + key of one score.  imported/own exactly as common.py:280-289.
  __device__ __forceinline__ bool accessible(uint32_t word, int bit, int32_t f, int64_t ek, int32_t own,
                                             int64_t qk) const {
    return ((word >> bit) & 1u) || (f == own && ek <= qk);
  }

  template <int FM, int FN>
  __device__ __forceinline__ void run(f32x16 (&acc)[FM][FN], int m_base, int n_base, int lane, char* /*stage*/) {
    const int hi = lane >> 5, cl = lane & 31;
    if (!filter && (debug_skip & 64)) return;
    // stage the tile's queries: own_file / q_key / threshold key / float lower bound of the threshold
    int32_t* s_own = reinterpret_cast<int32_t*>(smem);                  // [bm]
    float* s_tau = reinterpret_cast<float*>(smem + 1024);               // [bm]
    int64_t* s_qk = reinterpret_cast<int64_t*>(smem + 2048);            // [bm]
    uint64_t* s_thr = reinterpret_cast<uint64_t*>(smem + 2048 + 2048);  // [bm]
    float* s_qs = reinterpret_cast<float*>(smem + 6144);                // [bm]
    uint64_t* s_upper = reinterpret_cast<uint64_t*>(smem + 7168);       // [bm] keys >= this are not candidates
    for (int t = threadIdx.x; t < bm; t += blockDim.x) {
      const int q = tile_q0 + t;
      const bool ok = q < B;
      s_upper[t] = (ok && after_id && after_id[q] >= 0) ? make_key(after_score[q], after_id[q]) : ~0ull;
      s_qs[t] = (ok && q_scale) ? q_scale[q] : 1.f;
      s_own[t] = (ok && file_of) ? own_file[q] : -1;
      s_qk[t] = (ok && file_of) ? q_key[q] : 0;
      const uint64_t th = (ok && filter) ? thr[q] : 0ull;
      s_thr[t] = th;
      // keys > th have ordered(score) >= th.hi, i.e. score >= ord2f(th.hi): a cheap float pre-test
      s_tau[t] = (ok && filter && th) ? ord2f((uint32_t)(th >> 32)) : -INFINITY;
    }
    __syncthreads();
    const int ql0 = m_base - tile_q0;  // this wave's first query inside the tile
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      const int p = n_base + j * 32 + cl;
      const bool pvalid = p < N;
      int32_t f = -2;
      int64_t ek = 0;
      uint32_t w[FM];
#pragma unroll
      for (int i = 0; i < FM; ++i) w[i] = 0xffffffffu;
      if (file_of) {
#pragma unroll
        for (int i = 0; i < FM; ++i) w[i] = 0;
        if (pvalid) {
          f = file_of[p];
          ek = end_key[p];
          const int wi = m_base >> 5;
#pragma unroll
          for (int i = 0; i < FM; ++i)
            if (wi + i < bits_words) w[i] = bits_t[(size_t)f * bits_words + wi + i];
        }
      }
      const int32_t id = p + id_offset;
      const float es = (e_scale && pvalid) ? e_scale[p] : 1.f;
      if (!filter) {
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int rr = mfma32_row(r, hi);
            const int ql = ql0 + i * 32 + rr;
            const int q = tile_q0 + ql;
            bool ok = pvalid && (q < B);
            if (file_of) ok = ok && accessible(w[i], rr, f, ek, s_own[ql], s_qk[ql]);
            const float sc = q_scale ? (acc[i][j][r] * s_qs[ql]) * es : acc[i][j][r];
            uint64_t key = ok ? make_key(sc, id) : 0ull;
            if (key >= s_upper[ql]) key = 0ull;
            if (q < B && (p + slot_shift) < (int)dense_ld) dense[(size_t)q * dense_ld + p + slot_shift] = key;
          }
      } else {
        // Almost every score is below its query's bound: test that first (one LDS read + one compare),
        // and only survivors pay for the accessibility predicate, the 64-bit key and the atomic.
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int rr = mfma32_row(r, hi);
            const int ql = ql0 + i * 32 + rr;
            const float sc = q_scale ? (acc[i][j][r] * s_qs[ql]) * es : acc[i][j][r];
            if (sc >= s_tau[ql]) {
              const int q = tile_q0 + ql;
              bool ok = pvalid && (q < B);
              if (file_of) ok = ok && accessible(w[i], rr, f, ek, s_own[ql], s_qk[ql]);
              const uint64_t key = ok ? make_key(sc, id) : 0ull;
              if (key > s_thr[ql] && key < s_upper[ql]) {
                const int pos = atomicAdd(&count[(size_t)q * SIM_COUNT_STRIDE], 1);
                if (pos < cap) cand[(size_t)q * cap + pos] = key;
              }
            }
          }
      }
    }
  }
};

// Tile -> premise rows.  The plan samples / skips premises in blocks of SIM_PB = 256 rows; a block holds
// `sub` = 256 / C::BN tiles of this kernel.  (Dense-only plans pass stride = 1, sub = 1: tile = ord.)
//   pass 0 (dense keys):  sub-tile s of sampled block j * stride  ->  key slots [ord * BN, +BN)
//   pass 1 (filter):      sub-tile s of the fb-th block that is NOT a multiple of stride
template <class C>
__global__ __launch_bounds__(C::THREADS) void sim_scan_kernel(GemmOperand Qop, GemmOperand Eop, int K, int tiles_q,
                                                              int stride, int sub, EpiSim epi) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int logical = xcd_remap(blockIdx.x, gridDim.x);
  const int qt = logical % tiles_q;
  const int ord = logical / tiles_q;
  int pt;
  if (!epi.filter) {
    pt = (ord / sub) * stride * sub + ord % sub;
    epi.slot_shift = ord * C::BN - pt * C::BN;
  } else {
    const int fb = ord / sub;
    pt = (fb + fb / (stride - 1) + 1) * sub + ord % sub;
    epi.slot_shift = 0;
  }
  epi.smem = smem;
  epi.tile_q0 = qt * C::BM;
  epi.bm = C::BM;
  if constexpr (C::PIPE != 0)
    gemm_tile_pipe<C>(Qop, Eop, K, qt, pt, epi, smem);
  else
    gemm_tile<C>(Qop, Eop, K, qt, pt, epi, smem);
}

// ------------------------------------------------------------------------------------------
// Filter pass, second generation: premises are the MFMA row operand, queries the column operand.
//   acc[i][j][r]  <->  premise m_base + 32 i + (r & 3) + 8 (r >> 2) + 4 hi,   query n_base + 32 j + (lane & 31)
// so a lane holds, per column fragment j, ONE query: its lower bound tau (the float of the sampled k-th key)
// lives in a register and the pre-test is a single v_cmp per score.
//
// No atomics, no cross-lane traffic: the 64 scores a lane holds for one query (4 row fragments x 16 registers)
// have a PRIVATE run of 64 slots in global memory,
//   slots[workgroup tile][query in tile (256)][entry (64)][part = 2 wave_row + hi (4)]   x  {score bits, 16 i + r}
// (entry-major inside a query's 2 KB so that the FIRST entries of its four runs - all that is live in most runs -
// share one 128-byte line for the reader),
// and the lane appends its survivors (score >= tau: ~k * stride / (N * accessible fraction), 1-3 % of the scores)
// with a private counter, written at the end to  scnt[query][filter block][part].  A run can hold every score of
// its lane, so nothing can overflow whatever the data (no bound at all, near-duplicate blocks ...).  First
// versions appended to one list per query with an atomic counter: 256 counters x ~1600 returning atomics each,
// issued by all CUs in the same few microseconds of every round, cost more than the main loop's epilogue budget.
// The accessibility predicate and the exact 64-bit key test move to gather_slots_kernel, which reads the runs of
// ONE query per workgroup (the counts tell it how much of each run is live).
// The bounds (and the e4m3 scales) of the tile ride into LDS behind the ring by LDS-DMA issued before the first
// operand DMA (prologue hook of gemm_tile_pipe).
// ------------------------------------------------------------------------------------------
constexpr int SLOT_RUN = 64;                         // slots per (query, tile, part) = scores per lane and column fragment
constexpr int SLOTS_PER_TILE = 256 * 4 * SLOT_RUN;   // uint2 entries per workgroup tile (512 KB)

// PAGED (rp_sim_topk_after): the page's upper bound joins the float pre-test - score <= after_score[q] (ties go on to the exact
// key test in gather_select_kernel) - so the rows already returned on earlier pages never enter the runs (without it every
// page re-admitted them: ~1024 more entries per query and page for the one-at-a-time path behind the raw list), and an
// exhausted query (bound (-inf, INT_MAX)) admits nothing instead of every row of the index.  A separate instantiation:
// the first page's kernel carries no second compare.
template <int FP8, bool PAGED = false>
struct EpiSimFilter {
  int N, B;
  const float* q_scale;  // FP8: score = (acc * q_scale[query]) * e_scale[premise]
  const float* e_scale;
  const float* tau;      // [B] score of the sampled k-th key (-inf when the sample held fewer than k)
  const float* after_score = nullptr;  // PAGED: [B] the paging bound, as EpiSim
  const int32_t* after_id = nullptr;
  uint2* slots;          // [workgroup tiles][256][4][64]
  int32_t* scnt;         // [tiles_q * 256][filter_blocks][4]
  int filter_blocks;
  int stride, tiles_q;   // the plan's (a tile's filter block and slot area follow from its coordinates)
  // per workgroup
  char* smem;
  int meta_off;  // byte offset of the metadata behind the ring
  int p0;        // first premise of this workgroup's tile (prologue of the e4m3 form: one tile per workgroup)
  int debug_drop_all;

  static constexpr int M_TAU = 0, M_QS = 1024, M_ES = 2048, M_AS = 3072, M_AI = 4096;  // metadata layout (bytes from meta_off)

  __device__ __forceinline__ void prologue(char* meta, int wave, int lane, int q0) {  // (q0 = first query of the tile)
    auto dma4 = [&](const void* g, char* dst) {
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)g, (lds_ptr_t)dst, 4, 0, 0);
    };
    if (wave >= 4) return;           // (an 8-wave tile: the first four waves bring the 256 entries)
    const int e = wave * 64 + lane;  // wave w brings entries [64 w, 64 w + 64)
    const int q = min(q0 + e, B - 1), p = min(p0 + e, N - 1);
    dma4(tau + q, meta + M_TAU + wave * 256);
    if constexpr (FP8 != 0) {
      dma4(q_scale + q, meta + M_QS + wave * 256);
      dma4(e_scale + p, meta + M_ES + wave * 256);
    }
    if constexpr (PAGED) {
      dma4(after_score + q, meta + M_AS + wave * 256);
      dma4(after_id + q, meta + M_AI + wave * 256);
    }
  }

  template <int FM, int FN>
  __device__ __forceinline__ void run(f32x16 (&acc)[FM][FN], int m_base, int n_base, int lane, char* /*stage*/) {
    static_assert(FM * 16 == SLOT_RUN, "a lane's scores for one query fill exactly one run");
    const int hi = lane >> 5, cl = lane & 31;
    const char* meta = smem + meta_off;
    const float* s_tau = reinterpret_cast<const float*>(meta + M_TAU);
    const float* s_qs = reinterpret_cast<const float*>(meta + M_QS);
    const float* s_es = reinterpret_cast<const float*>(meta + M_ES);
    // the tile from the wave's coordinates: premise block pb is the fb-th block that is not a multiple of stride
    const int pb = m_base >> 8, qt = n_base >> 8;
    const int fb = pb - pb / stride - 1, wg_tile = fb * tiles_q + qt, q0 = qt << 8;
    const int pl0 = m_base - (pb << 8), ql0 = n_base - q0;  // this wave's first premise / query inside the tile
    const int part = (pl0 >> 7) * 2 + hi;

    // Append without a branch: ~5 % of the scores pass (k * stride accessible rows above the bound, ~3 x that before the
    // predicate), so for nearly every score SOME lane of the wave appends, and a compiler-built `if` costs a saveexec, two taken
    // branches to an out-of-line block and back and a 64-bit address computation per score (6.3 us of epilogue per tile for
    // 128 compares per lane).  Here the compare narrows exec itself and the store + the cursor bump run under it:
    // compare, exec <- vcc, store, add, exec <- all - straight-line code.  The run's cursor is a 32-bit byte offset from the
    // tile's slot area (a uniform base in SGPRs); the count is what the cursor moved.  (exec is all ones on entry: the
    // epilogue runs in uniform control flow of full waves.)
    float tauv[FN], qsv[FN], upv[FN];
    uint32_t off[FN], off0[FN];
    const uint2* const tile_slots = slots + (size_t)wg_tile * 256 * (4 * SLOT_RUN);
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      const int ql = ql0 + j * 32 + cl;
      tauv[j] = (debug_drop_all & 1) ? INFINITY : s_tau[ql];
      upv[j] = INFINITY;
      if constexpr (PAGED) {  // (after_id < 0: no bound for this query)
        const float as = reinterpret_cast<const float*>(meta + M_AS)[ql];
        if (reinterpret_cast<const int*>(meta + M_AI)[ql] >= 0) upv[j] = as;
      }
      qsv[j] = FP8 ? s_qs[ql] : 1.f;
      off0[j] = off[j] = (uint32_t)((ql * (4 * SLOT_RUN) + part) * (int)sizeof(uint2));
    }
    auto append = [&](float sc, uint32_t code, int j) {
      const uint64_t entry = ((uint64_t)code << 32) | __float_as_uint(sc);  // {score bits, 16 i + r} as the uint2 the reader takes
      if constexpr (PAGED)
        asm volatile(
            "v_cmp_ge_f32_e32 vcc, %2, %3\n\t"
            "s_mov_b64 exec, vcc\n\t"
            "v_cmp_le_f32_e32 vcc, %2, %5\n\t"
            "s_mov_b64 exec, vcc\n\t"
            "global_store_dwordx2 %0, %1, %4\n\t"
            "v_add_u32_e32 %0, 32, %0\n\t"  // entry stride: the four parts of a query interleave
            "s_mov_b64 exec, -1"
            : "+v"(off[j])
            : "v"(entry), "v"(sc), "v"(tauv[j]), "s"(tile_slots), "v"(upv[j])
            : "vcc", "memory");
      else
        asm volatile(
            "v_cmp_ge_f32_e32 vcc, %2, %3\n\t"
            "s_mov_b64 exec, vcc\n\t"
            "global_store_dwordx2 %0, %1, %4\n\t"
            "v_add_u32_e32 %0, 32, %0\n\t"
            "s_mov_b64 exec, -1"
            : "+v"(off[j])
            : "v"(entry), "v"(sc), "v"(tauv[j]), "s"(tile_slots)
            : "vcc", "memory");
    };
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      float esv[16];
      if constexpr (FP8 != 0) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(s_es + pl0 + i * 32 + 8 * g + 4 * hi);
#pragma unroll
          for (int e = 0; e < 4; ++e) esv[4 * g + e] = v[e];
        }
      }
#pragma unroll
      for (int j = 0; j < FN; ++j) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float sc = FP8 ? (acc[i][j][r] * qsv[j]) * esv[r] : acc[i][j][r];
          append(sc, (uint32_t)(i * 16 + r), j);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < FN; ++j)
      scnt[((size_t)(q0 + ql0 + j * 32 + cl) * filter_blocks + fb) * 4 + part] = (int)((off[j] - off0[j]) >> 5);
  }
};

template <class C, bool PAGED>
__global__ __launch_bounds__(C::THREADS) void sim_filter_kernel(GemmOperand Eop, GemmOperand Qop, int K, int tiles_q,
                                                                int stride, EpiSimFilter<C::FP8, PAGED> epi) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  static_assert(C::BM == SIM_PB && C::BN == 256 && C::WM == 2, "filter tile geometry");
  const int logical = xcd_remap(blockIdx.x, gridDim.x);
  const int qt = logical % tiles_q;
  const int fb = logical / tiles_q;
  const int pb = fb + fb / (stride - 1) + 1;  // the fb-th block that is not a multiple of stride
  epi.smem = smem;
  epi.meta_off = C::RING_BYTES;
  epi.p0 = pb * C::BM;
  if constexpr (C::PIPE != 0)
    gemm_tile_pipe<C>(Eop, Qop, K, pb, qt, epi, smem);
  else
    gemm_tile<C>(Eop, Qop, K, pb, qt, epi, smem);
}

// ------------------------------------------------------------------------------------------
// One workgroup per query: walk the query's runs (scnt says how many entries of each are live), apply the
// accessibility predicate (common.py:280-289 in array form) and the exact key test against the sampled bound, and
// append the passing keys behind the sample's own top-k in cand[q] (count[q] is updated).  The raw survivors are
// first collected in LDS so that the dependent gathers of the predicate (file_of -> mask word) run entry-parallel,
// four entries per thread in flight; the rare excess beyond the LDS list is handled entry by entry.
// ------------------------------------------------------------------------------------------
struct GatherArgs {
  const uint2* slots;
  const int32_t* scnt;
  int filter_blocks, tiles_q, stride;
  int N, B, id_offset;
  const int32_t* file_of;  // NULL: no mask
  const int64_t* end_key;
  const uint32_t* bits_t;
  int bits_words;
  const int32_t* own_file;
  const int64_t* q_key;
  const uint64_t* thr;
  const float* after_score;  // paging bound (NULL: none), as EpiSim
  const int32_t* after_id;
  uint64_t* cand;  // [B, cap]: entries [0, count) hold the sample's top keys on entry
  size_t cap;
  int32_t* count;  // [B * SIM_COUNT_STRIDE]
  int debug;       // timing only: 8 = return at once, 16 = skip the entry loops of phase A, 32 = skip phase B
};

// ------------------------------------------------------------------------------------------
// exact top-k of one query's key list: MSB radix select (8-bit digits) + bitonic sort.
//   keys are distinct (ids differ) except for 0 = "not a candidate".
// ------------------------------------------------------------------------------------------
struct SelectArgs {
  const uint64_t* keys;   // [B, ld]
  size_t ld;
  const int32_t* counts;  // per-query list length at counts[q * count_stride] (NULL: n_fixed); > cap => overflow
  int count_stride;
  int n_fixed;
  int cap;
  int k;
  // mode A outputs (sample stage): top keys into out_keys[q, 0..kk), out_cnt[q] = kk, thr[q]
  uint64_t* out_keys;
  size_t out_ld;
  int32_t* out_cnt;
  uint64_t* out_thr;
  float* out_tau;  // score of out_thr (-inf when fewer than k keys were found): the filter pass's float pre-test
  // mode B outputs (final)
  float* out_scores;   // [B, k]
  int32_t* out_ids;    // [B, k]
  int32_t* out_count;  // [B]
  int debug;           // timing only (scan_no_epilogue): 256 = return once the list is loaded, 512 = after the first barrier,
                       // 1024 = after the radix passes, 2048 = after the sort
};

// Per-query stages: what bounds them is LATENCY (a chain of ~10 dependent global / LDS round trips per query), so what they
// need is workgroups per CU, i.e. little LDS and few registers each:
//  * select_kernel keeps the list in REGISTERS (KPT keys per thread, every pass walks them with static indices); LDS holds
//    only the histograms and the output list;
//  * gather_select_kernel's raw-entry list and key list share one LDS buffer (the key list grows from the front into what
//    the raw list has already given up);
//  * two shapes each: <1024 threads> for a few queries with long lists (one per CU is all there is), <256 threads, k <= 128>
//    for MANY queries with short lists (the rows of one GPU's shard under all the queries of a multi-GPU step): 8 workgroups
//    per CU - 2048 queries are one round of the chip.  Same code, same results; lists that do not fit are read from global
//    memory in either shape.
constexpr int SELECT_THREADS = 1024, SELECT_KPT = 8;          // 8192 keys in registers
constexpr int SELECT_THREADS_SMALL = 256, SELECT_KPT_SMALL = 16;  // 4096
constexpr int SELECT_SMALL_MAX_K = 128;

template <int THREADS, int SEL_CAP>
struct SelectSharedT {
  uint64_t sel[SEL_CAP];
  uint32_t s_or[THREADS / 64], s_and[THREADS / 64];
  int s_cntw[THREADS / 64];
  int hist[3][256];  // radix pass p counts into hist[p % 3] (see select_body)
  int s_cnt;
};

// the key list of one query: in LDS when it fits, else in global memory
struct KeySrc {
  const uint64_t* g;
  const uint64_t* l;
  bool in_lds;
  __device__ __forceinline__ uint64_t operator[](int i) const { return in_lds ? l[i] : g[i]; }
};

// Everything after the list is in place: exact top-k of src[0, n) -> the mode A / mode B outputs of query q.
// `each(f)` calls f(key) for every key of the list this thread is responsible for (0 = "not a candidate" may be among them).
template <int THREADS, int SEL_CAP, class Each>
__device__ __forceinline__ void select_body(const SelectArgs& a, int q, Each each, bool overflow,
                                            SelectSharedT<THREADS, SEL_CAP>& sh) {
  const int tid = threadIdx.x, lane = tid & 63;
  // number of real candidates, and the bits in which they differ at all: the radix passes start at the highest
  // differing bit.  (Scores of one query share sign, exponent and often a few mantissa bits: byte-aligned passes
  // from bit 63 spent their first rounds sending every key - and every 0 = "not a candidate" - to ONE histogram
  // bin, thousands of LDS atomics on one address.)
  // One workgroup barrier per phase: what a later phase needs cleared (the first histogram, the output list, the append
  // counter) is cleared here, before the first barrier; every wave then derives the same decisions from the same LDS
  // words by itself (no "wave 0 decides, barrier, everyone reads").
  for (int i = tid; i < SEL_CAP; i += THREADS) sh.sel[i] = 0ull;
  if (tid < 256) sh.hist[0][tid] = 0;
  if (tid == 0) sh.s_cnt = 0;
  {
    int c = 0;
    uint32_t vo = 0u, va = ~0u;  // over the score words (the keys' upper halves)
    each([&](uint64_t key) {
      if (key != 0ull) {
        ++c;
        vo |= (uint32_t)(key >> 32);
        va &= (uint32_t)(key >> 32);
      }
    });
    c = (int)wave_sum((float)c);  // exact: c <= 2^24 per wave
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      vo |= (uint32_t)__shfl_xor((int)vo, o, 64);
      va &= (uint32_t)__shfl_xor((int)va, o, 64);
    }
    if (lane == 0) {
      sh.s_cntw[tid >> 6] = c;
      sh.s_or[tid >> 6] = vo;
      sh.s_and[tid >> 6] = va;
    }
  }
  __syncthreads();
  int nvalid = 0;
  uint32_t all_or = 0u, all_and = ~0u;
#pragma unroll
  for (int w = 0; w < THREADS / 64; ++w) {
    nvalid += sh.s_cntw[w];
    all_or |= sh.s_or[w];
    all_and &= sh.s_and[w];
  }
  const int kk = min(a.k, nvalid);
  if (a.debug & 512) return;

  // The passes are what this stage costs (every key of the list is looked at in each, and a CU's instruction issue is
  // shared by every query resident on it), so they work on 32-bit words: first the score word alone; the id word only if
  // the kk-th score is tied.  One pass refines the bucket (word >> shift) == prefix by up to 8 more bits: counts into
  // hist[pass % 3] and clears hist[(pass + 1) % 3] for the next pass while it counts - that buffer's last readers were
  // the scans of pass - 2, which every wave finished before it arrived at the barrier of pass - 1.
  int pass = 0;
  auto refine = [&](auto&& word_of, int& shift, uint32_t& prefix, int& need, int& bucket) {
    const int new_shift = max(shift - 8, 0);
    const int width = shift - new_shift;
    const uint32_t dmask = (1u << width) - 1u;
    int* const h = sh.hist[pass % 3];
    if (tid < 256) sh.hist[(pass + 1) % 3][tid] = 0;
    each([&](uint64_t key) {
      uint32_t w;
      const bool member = word_of(key, w);
      if (member && (shift >= 32 || (w >> shift) == prefix)) atomicAdd(&h[(int)((w >> new_shift) & dmask)], 1);
    });
    __syncthreads();
    ++pass;
    // every wave: lane l owns digits 255-4l .. 252-4l (descending)
    const int d0 = 255 - 4 * lane;
    const int h0 = h[d0], h1 = h[d0 - 1], h2 = h[d0 - 2], h3 = h[d0 - 3];
    const int mine = h0 + h1 + h2 + h3;
    int incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      int t = __shfl_up(incl, o, 64);
      if (lane >= o) incl += t;
    }
    const int excl = incl - mine;
    const bool crossing = excl < need && incl >= need;  // the need-th word from the top lies in this lane's 4 digits
    int d = d0, take = need - excl, hsel = h0;
    {
      const int hs[4] = {h0, h1, h2, h3};
      int before = excl;
#pragma unroll
      for (int t = 0; t < 4; ++t) {  // the first digit whose cumulative count reaches `need`
        if (before < need && before + hs[t] >= need) {
          d = d0 - t;
          hsel = hs[t];
          take = need - before;
        }
        before += hs[t];
      }
    }
    const unsigned long long cm = __ballot(crossing);  // exactly one lane: the bucket holds at least `need` words
    const int src_lane = (int)__builtin_ctzll(cm);
    prefix = (shift >= 32 ? 0u : (prefix << width)) | (uint32_t)__shfl(d, src_lane, 64);
    shift = new_shift;
    need = __shfl(take, src_lane, 64);      // how many to take inside the chosen bucket
    bucket = __shfl(hsel, src_lane, 64);    // ... of how many: equal = the whole bucket is taken, no need to refine further
  };
  auto top_shift = [](uint32_t diff) { return diff ? 32 - (int)__builtin_clz(diff) : 0; };

  uint64_t T = ~0ull;  // keys >= T are selected
  if (kk > 0) {
    // score words agree above bit shift - 1; prefix = word >> shift of the bucket being refined
    int shift = top_shift(all_or ^ all_and), need = kk, bucket = nvalid;
    uint32_t prefix = shift >= 32 ? 0u : (all_and >> shift);
    auto score_word = [](uint64_t key, uint32_t& w) {
      w = (uint32_t)(key >> 32);
      return key != 0ull;
    };
    while (bucket != need && shift > 0) refine(score_word, shift, prefix, need, bucket);
    if (bucket == need) {
      T = (uint64_t)(shift >= 32 ? 0u : (prefix << shift)) << 32;
      if (T == 0ull) T = 1ull;
    } else {
      // `bucket` keys share the kk-th score word exactly: the `need` largest id words among them (= smallest ids)
      const uint32_t score = prefix;
      int shift2 = 32, need2 = need, bucket2 = bucket;
      uint32_t prefix2 = 0u;
      auto id_word = [score](uint64_t key, uint32_t& w) {
        w = (uint32_t)key;
        return (uint32_t)(key >> 32) == score && key != 0ull;
      };
      while (bucket2 != need2 && shift2 > 0) refine(id_word, shift2, prefix2, need2, bucket2);
      T = ((uint64_t)score << 32) | (shift2 >= 32 ? 0u : (prefix2 << shift2));
      if (T == 0ull) T = 1ull;
    }
  }

  if (a.debug & 1024) return;
  // collect the kk selected keys (the list was zeroed above: padded to a power of two for the sort), sort descending
  int P = 1;
  while (P < kk) P <<= 1;
  if (kk > 0) {
    each([&](uint64_t key) {
      if (key >= T && key != 0ull) {
        const int pos = atomicAdd(&sh.s_cnt, 1);
        if (pos < SEL_CAP) sh.sel[pos] = key;
      }
    });
  }
  __syncthreads();
  if (a.out_keys && !a.out_scores) {
    // sample stage: what follows needs the SET of the kk best keys (they join the candidates, which are selected again) and
    // the smallest of them as the bound - no order: one wave's minimum instead of a sort
    if (a.debug & 2048) return;
    for (int i = tid; i < kk; i += THREADS) a.out_keys[(size_t)q * a.out_ld + i] = sh.sel[i];
    if (tid < 64) {
      uint64_t m = ~0ull;
      for (int i = lane; i < kk; i += 64) m = sh.sel[i] < m ? sh.sel[i] : m;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const uint32_t olo = (uint32_t)__shfl_xor((int)(uint32_t)m, o, 64);
        const uint32_t ohi = (uint32_t)__shfl_xor((int)(uint32_t)(m >> 32), o, 64);
        const uint64_t x = ((uint64_t)ohi << 32) | olo;
        m = x < m ? x : m;
      }
      if (lane == 0) {
        a.out_cnt[(size_t)q * SIM_COUNT_STRIDE] = kk;
        const uint64_t th = (kk == a.k) ? m : 0ull;
        a.out_thr[q] = th;
        // keys > th have ordered(score) >= th.hi, i.e. score >= ord2f(th.hi)
        if (a.out_tau) a.out_tau[q] = th ? ord2f((uint32_t)(th >> 32)) : -INFINITY;
      }
    }
    return;
  }
  if (P <= 128) {
    // the common case (k <= 128): ONE wave sorts the list in registers, two keys per lane (element e = lane + 64 r), the
    // same bitonic network through lane exchanges - 28 stages without a workgroup barrier each (with 1024 threads a
    // barrier stage cost more than the compare it guarded)
    if (tid < 64) {
      uint64_t v0 = lane < P ? sh.sel[lane] : 0ull, v1 = lane + 64 < P ? sh.sel[lane + 64] : 0ull;
      auto xchg = [&](uint64_t& v, int e, int size, int strd) {
        const uint32_t olo = (uint32_t)__shfl_xor((int)(uint32_t)v, strd, 64);
        const uint32_t ohi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), strd, 64);
        const uint64_t o = ((uint64_t)ohi << 32) | olo;
        const bool lower = (e & strd) == 0, desc = (e & size) == 0;
        v = (lower == desc) ? (v > o ? v : o) : (v < o ? v : o);
      };
      for (int size = 2; size <= 128; size <<= 1)
        for (int strd = size >> 1; strd > 0; strd >>= 1) {
          if (strd == 64) {  // partner = the lane's other register (only in the last merge: descending)
            const uint64_t hi_ = v0 > v1 ? v0 : v1, lo_ = v0 > v1 ? v1 : v0;
            v0 = hi_;
            v1 = lo_;
          } else {
            xchg(v0, lane, size, strd);
            xchg(v1, lane + 64, size, strd);
          }
        }
      // (a list padded to P < 128 sorts as one of 128: the zero padding ends up behind every key)
      if (lane < P) sh.sel[lane] = v0;
      if (lane + 64 < P) sh.sel[lane + 64] = v1;
    }
    __syncthreads();
  } else
  for (int size = 2; size <= P; size <<= 1) {
    for (int strd = size >> 1; strd > 0; strd >>= 1) {
      for (int i = tid; i < (P >> 1); i += THREADS) {
        const int lo = ((i / strd) * strd * 2) + (i % strd);
        const int hi2 = lo + strd;
        const bool desc = ((lo & size) == 0);
        const uint64_t x = sh.sel[lo], y = sh.sel[hi2];
        if ((x < y) == desc) {
          sh.sel[lo] = y;
          sh.sel[hi2] = x;
        }
      }
      __syncthreads();
    }
  }

  if (a.debug & 2048) return;
  if (a.out_keys) {
    for (int i = tid; i < kk; i += THREADS) a.out_keys[(size_t)q * a.out_ld + i] = sh.sel[i];
    if (tid == 0) {
      a.out_cnt[(size_t)q * SIM_COUNT_STRIDE] = kk;
      const uint64_t th = (kk == a.k) ? sh.sel[kk - 1] : 0ull;
      a.out_thr[q] = th;
      // keys > th have ordered(score) >= th.hi, i.e. score >= ord2f(th.hi)
      if (a.out_tau) a.out_tau[q] = th ? ord2f((uint32_t)(th >> 32)) : -INFINITY;
    }
  }
  if (a.out_scores) {
    for (int i = tid; i < a.k; i += THREADS) {
      const bool v = i < kk;
      const uint64_t key = v ? sh.sel[i] : 0ull;
      a.out_scores[(size_t)q * a.k + i] = v ? ord2f((uint32_t)(key >> 32)) : -INFINITY;
      a.out_ids[(size_t)q * a.k + i] = v ? (int32_t)(~(uint32_t)key) : -1;
    }
    if (tid == 0) a.out_count[q] = overflow ? -1 : kk;
  }
}

template <int THREADS, int KPT>
__global__ __launch_bounds__(THREADS, THREADS <= 512 ? 8 : 4) void select_kernel(SelectArgs a) {
  __shared__ SelectSharedT<THREADS, SIM_MAX_K> sh;
  const int q = blockIdx.x, tid = threadIdx.x;
  int n = a.counts ? a.counts[(size_t)q * a.count_stride] : a.n_fixed;
  const bool overflow = a.counts && n > a.cap;
  if (n > a.cap && a.counts) n = a.cap;
  const uint64_t* gsrc = a.keys + (size_t)q * a.ld;
  // The passes read the list about six times: from registers when it fits (the sample's keys), from global memory
  // otherwise (dense plans, adversarial candidate counts).
  if (n <= KPT * THREADS) {
    uint64_t keys[KPT];
#pragma unroll
    for (int j = 0; j < KPT; ++j) {
      const int i = j * THREADS + tid;
      keys[j] = i < n ? gsrc[i] : 0ull;
    }
    if (a.debug & 256) {  // (keep the loads alive)
      uint64_t x = 0ull;
#pragma unroll
      for (int j = 0; j < KPT; ++j) x |= keys[j];
      if (x == 0x123456789abcdefull) a.out_thr[q] = x;
      return;
    }
    select_body(a, q, [&](auto&& f) {
#pragma unroll
      for (int j = 0; j < KPT; ++j) f(keys[j]);
    }, overflow, sh);
  } else {
    select_body(a, q, [&](auto&& f) {
      for (int i = tid; i < n; i += THREADS) f(gsrc[i]);
    }, overflow, sh);
  }
}

// ------------------------------------------------------------------------------------------
// Final stage of the second-generation plan, one workgroup per query: walk the query's runs (scnt says how many
// entries of each are live), apply the accessibility predicate (common.py:280-289 in array form) and the exact key
// test against the sampled bound, put the passing keys behind the sample's own top-k, and select (select_body).
// Latency is what this stage is made of (a few thousand scattered 8-byte reads per query), so every phase issues
// its independent loads together: a block's four counts AND the first four entries of its four runs (one 128-byte
// line) are requested at once - the line's address does not depend on the counts, and a run holds ~2 live entries
// on average; positions in the raw list come from ONE block-wide prefix sum of the per-thread totals (a first
// version took a wave-aggregated LDS atomic per entry round: 20 of its 37 us); the predicate's dependent gathers
// run for all raw survivors of a thread level by level.
// ------------------------------------------------------------------------------------------
// raw-entry capacity of the LDS list (+ SEL_CAP keys of the sample in front of it, see the kernel): entries beyond it take
// the slow path, one at a time - the capacity must cover the raw count's spread, not its mean (at B = 256 x 130 k rows,
// ~6400 expected: 7168 entries cost 50 us where 8192+ cost 36).  big: 91 KB; mid 36 KB (four workgroups per CU, 512
// threads); small 18.5 KB (eight, 256 threads)
constexpr int GATHER_ENTRIES = 9216, GATHER_ENTRIES_MID = 3968, GATHER_ENTRIES_SMALL = 1664;
constexpr int GATHER_THREADS_MID = 512;

__device__ __forceinline__ int wave_max_i32(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
  return v;
}

// wave-aggregated append: one LDS atomic per wave, positions by mbcnt
__device__ __forceinline__ int wave_append_pos(int* counter, bool want) {
  const unsigned long long ball = __ballot(want);
  int base = 0;
  if (ball) {
    const int lane = threadIdx.x & 63;
    if (lane == (int)__builtin_ctzll(ball)) base = atomicAdd(counter, __popcll(ball));
    base = __shfl(base, (int)__builtin_ctzll(ball), 64);
  }
  return base + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(ball >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)ball, 0u));
}

template <int THREADS, int RAW_ENTRIES, int SEL_CAP>
__global__ __launch_bounds__(THREADS, THREADS <= 512 ? 8 : 4) void gather_select_kernel(GatherArgs a, SelectArgs sa) {
  __shared__ SelectSharedT<THREADS, SEL_CAP> sh;
  // One buffer for two lists: keys [0, LDS_KEYS) from the front (the sample's top keys, then what passes), raw entries
  // {score bits, premise row} from slot SEL_CAP on.  The key list never passes the raw entries still to be read: it holds
  // at most n_sample <= SEL_CAP keys + one per entry processed, and phase B loads a whole round of entries into registers
  // (barrier) before it appends that round's keys.
  constexpr int LDS_KEYS = SEL_CAP + RAW_ENTRIES;
  __shared__ uint64_t ubuf[LDS_KEYS];
  uint64_t* const staged = ubuf;
  uint2* const raw = reinterpret_cast<uint2*>(ubuf + SEL_CAP);
  __shared__ int s_wave_tot[THREADS / 64];
  __shared__ int s_out, s_spill;
  const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int qt = q >> 8, qloc = q & 255;
  const uint64_t thr = a.thr[q];
  const uint64_t upper = (a.after_id && a.after_id[q] >= 0) ? make_key(a.after_score[q], a.after_id[q]) : ~0ull;
  const int32_t own = a.file_of ? a.own_file[q] : -1;
  const int64_t qk = a.file_of ? a.q_key[q] : 0;
  uint64_t* out = a.cand + (size_t)q * a.cap;  // global copy of the key list (read back only when LDS is too small)
  if (a.debug & 8) return;
  const int n_sample = a.count[(size_t)q * SIM_COUNT_STRIDE];
  for (int i = tid; i < n_sample; i += THREADS) staged[i] = out[i];  // n_sample <= k <= SEL_CAP
  if (tid == 0) {
    s_out = n_sample;
    s_spill = 0;
  }
  auto accessible = [&](int32_t f, uint32_t word, int p) {
    return ((word >> (q & 31)) & 1u) || (f == own && a.end_key[p] <= qk);
  };
  auto accessible_ek = [&](int32_t f, uint32_t word, int64_t ek) {  // (the row's end key already loaded)
    return ((word >> (q & 31)) & 1u) || (f == own && ek <= qk);
  };
  bool spill = false;  // set after phase A: keys went straight to the global list, whose positions the LDS bound does not cover
  auto put_key = [&](uint64_t key, int pos) {
    if (pos < (int)a.cap) out[pos] = key;  // beyond the capacity: counted, not stored (reported as overflow below)
    if (!spill && pos < LDS_KEYS) staged[pos] = key;
  };
  // ---- phase A: runs -> raw list (one thread per filter block)
  const i32x4* cnt4 = reinterpret_cast<const i32x4*>(a.scnt + (size_t)q * a.filter_blocks * 4);
  int raw_base = 0;  // raw entries of earlier rounds of this loop (a query has more than 1024 filter blocks at 1M rows)
  for (int fb0 = 0; fb0 < a.filter_blocks; fb0 += THREADS) {
    const int fb = fb0 + tid;
    const int fbc = min(fb, a.filter_blocks - 1);
    const uint2* base = a.slots + ((size_t)(fbc * a.tiles_q + qt) * 256 + qloc) * (4 * SLOT_RUN);
    i32x4 n4 = cnt4[fbc];
    if (fb >= a.filter_blocks || (a.debug & 16)) n4 = i32x4{0, 0, 0, 0};
    uint4 line[8];  // entries e = 0..3 of parts 0..3: line[2 e + (part >> 1)] holds parts {0,1} / {2,3}
#pragma unroll
    for (int w = 0; w < 8; ++w) line[w] = reinterpret_cast<const uint4*>(base)[w];
    const int row_pb = (fbc + fbc / (a.stride - 1) + 1) * SIM_PB;
    // block-wide exclusive prefix sum of the per-thread entry totals
    const int mine = n4[0] + n4[1] + n4[2] + n4[3];
    int incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int t = __shfl_up(incl, o, 64);
      if (lane >= o) incl += t;
    }
    __syncthreads();  // s_wave_tot free (previous round consumed)
    if (lane == 63) s_wave_tot[wave] = incl;
    __syncthreads();
    int pos = raw_base + incl - mine;
    int round_total = 0;
#pragma unroll
    for (int w = 0; w < THREADS / 64; ++w) {
      const int t = s_wave_tot[w];
      if (w < wave) pos += t;
      round_total += t;
    }
    raw_base += round_total;
    auto emit = [&](uint2 en, int part) {
      const int i = (int)(en.y >> 4), r = (int)(en.y & 15u);
      const int p = row_pb + (part >> 1) * 128 + 4 * (part & 1) + i * 32 + (r & 3) + 8 * (r >> 2);
      if (pos < RAW_ENTRIES) {
        raw[pos] = make_uint2(en.x, (uint32_t)p);
      } else if (p < a.N) {  // beyond the LDS list (rare): straight through, one entry at a time
        bool ok = true;
        if (a.file_of) {
          const int32_t f = a.file_of[p];
          ok = accessible(f, a.bits_t[(size_t)f * a.bits_words + (q >> 5)], p);
        }
        const uint64_t key = make_key(__uint_as_float(en.x), p + a.id_offset);
        if (ok && key > thr && key < upper) {  // global copy only (the LDS slot may still hold a raw entry): the select reads it
          const int kp = atomicAdd(&s_out, 1);
          if (kp < (int)a.cap) out[kp] = key;
          s_spill = 1;
        }
      }
      ++pos;
    };
#pragma unroll
    for (int e = 0; e < 4; ++e)  // from the registers (static indices)
#pragma unroll
      for (int part = 0; part < 4; ++part)
        if (e < n4[part]) {
          const uint4 v = line[2 * e + (part >> 1)];
          emit((part & 1) ? make_uint2(v.z, v.w) : make_uint2(v.x, v.y), part);
        }
    // longer runs (a loose bound: few rows per shard, many queries): one 128-byte line = entries e0 .. e0 + 3 of the four
    // parts per round trip, not one entry
    const int nmax = max(max(n4[0], n4[1]), max(n4[2], n4[3]));
    for (int e0 = 4; e0 < nmax; e0 += 4) {
#pragma unroll
      for (int w = 0; w < 8; ++w) line[w] = reinterpret_cast<const uint4*>(base + e0 * 4)[w];
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int part = 0; part < 4; ++part)
          if (e0 + e < n4[part]) {
            const uint4 v = line[2 * e + (part >> 1)];
            emit((part & 1) ? make_uint2(v.z, v.w) : make_uint2(v.x, v.y), part);
          }
    }
  }
  __syncthreads();
  // ---- phase B: entry-parallel predicate; every gather level of a thread's entries is issued before the next
  spill = s_spill != 0;
  const int nraw = (a.debug & 32) ? 0 : min(raw_base, RAW_ENTRIES);
  constexpr int U = 4;
  for (int e0 = 0; e0 < nraw; e0 += U * THREADS) {
    uint2 en[U];
    int32_t f[U];
    int64_t ek[U];  // requested with the file (its address does not depend on it): two gather levels, not three
    uint32_t word[U];
    bool live[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int e = e0 + u * THREADS + tid;
      en[u] = raw[min(e, nraw - 1)];
      live[u] = e < nraw && (int)en[u].y < a.N;  // padding rows of the last block never qualify
      if (!live[u]) en[u].y = 0;
      f[u] = a.file_of ? a.file_of[en[u].y] : 0;
      ek[u] = a.file_of ? a.end_key[en[u].y] : 0;
    }
    __syncthreads();  // this round's entries are in registers: their slots may be overwritten by keys
#pragma unroll
    for (int u = 0; u < U; ++u) word[u] = a.file_of ? a.bits_t[(size_t)f[u] * a.bits_words + (q >> 5)] : 0xffffffffu;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      bool ok = live[u];
      if (a.file_of && ok) ok = accessible_ek(f[u], word[u], ek[u]);
      const uint64_t key = make_key(__uint_as_float(en[u].x), (int32_t)en[u].y + a.id_offset);
      ok = ok && key > thr && key < upper;
      const int pos = wave_append_pos(&s_out, ok);
      if (ok) put_key(key, pos);
    }
  }
  __syncthreads();
  const bool overflow = s_out > (int)a.cap;  // out_count = -1: the caller repeats the search with the dense plan
  const int n = min(s_out, (int)a.cap);
  if (a.debug & 64) return;
  const KeySrc src{out, staged, n <= LDS_KEYS && !spill};
  select_body(sa, q, [&](auto&& f) {
    for (int i = tid; i < n; i += THREADS) f(src[i]);
  }, overflow, sh);
}

// (scores, ids, counts)[R, B, k] -> keys[B, R*k]
__global__ void merge_keys_kernel(const float* scores, const int32_t* ids, const int32_t* counts, int R, int B,
                                  int k, uint64_t* keys) {
  const int q = blockIdx.x;
  for (int t = threadIdx.x; t < R * k; t += blockDim.x) {
    const int r = t / k, i = t % k;
    const size_t src = ((size_t)r * B + q) * k + i;
    const int c = counts[(size_t)r * B + q];
    keys[(size_t)q * R * k + t] = (i < c) ? make_key(scores[src], ids[src]) : 0ull;
  }
}

// The same from one packed buffer per rank (an all-gather's receive buffer as it lies): element [r, q, i] of scores / ids
// at base + r * rank_stride + q * k + i, counts at base + r * rank_stride + q (4-byte units).
__global__ void merge_keys_strided_kernel(const float* scores, const int32_t* ids, const int32_t* counts,
                                          size_t rank_stride, int R, int k, uint64_t* keys) {
  const int q = blockIdx.x;
  for (int t = threadIdx.x; t < R * k; t += blockDim.x) {
    const int r = t / k, i = t % k;
    const size_t src = (size_t)r * rank_stride + (size_t)q * k + i;
    const int c = counts[(size_t)r * rank_stride + q];
    keys[(size_t)q * R * k + t] = (i < c) ? make_key(scores[src], ids[src]) : 0ull;
  }
}

// file_bits_t[f, w] bit j = query 32 w + j may use file f = bit f of row own_file[32 w + j] of the closure matrix
// `reach` (bit g of row f: f imports g, transitively; F x ceil(F / 64) words, resident on the device): the per-batch
// accessibility operand of rp_sim_topk built where it is used, from 4 bytes per query.
__global__ __launch_bounds__(256) void build_file_bits_kernel(const uint64_t* __restrict__ reach, int F, int W64,
                                                              const int32_t* __restrict__ own_file, int B, int Wq,
                                                              uint32_t* __restrict__ bits_t) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= F * Wq) return;
  const int f = i / Wq, w = i - f * Wq;
  uint32_t word = 0;
#pragma unroll 8
  for (int j = 0; j < 32; ++j) {
    const int q = 32 * w + j;
    if (q < B) {
      const int own = own_file[q];
      word |= (uint32_t)((reach[(size_t)own * W64 + (f >> 6)] >> (f & 63)) & 1ull) << j;
    }
  }
  bits_t[i] = word;
}

// many queries with short lists: the small shape (four workgroups per CU)
static bool select_small(int B, int64_t list_bound) {
  return B >= 512 && list_bound <= SELECT_KPT_SMALL * SELECT_THREADS_SMALL;
}

static void launch_select(const SelectArgs& a, int B, hipStream_t stream) {
  ProfScope ps(stream, RP_K_SELECT);
  if (select_small(B, a.counts ? (int64_t)a.cap : (int64_t)a.n_fixed))
    hipLaunchKernelGGL((select_kernel<SELECT_THREADS_SMALL, SELECT_KPT_SMALL>), dim3(B), dim3(SELECT_THREADS_SMALL), 0,
                       stream, a);
  else
    hipLaunchKernelGGL((select_kernel<SELECT_THREADS, SELECT_KPT>), dim3(B), dim3(SELECT_THREADS), 0, stream, a);
}

struct SimPlan {
  bool dense_only;
  bool new_filter;  // two-pass: second-generation filter kernel (needs whole 128-B operand rows per K-tile)
  int stride;       // two-pass: every stride-th 256-row block is sampled
  int blocks, sample_blocks, filter_blocks;
  int bm, tiles_q, tiles_p;  // dense-only: first-generation kernel, bm queries x 128 premises per tile
  size_t dense_ld;           // keys per query in the dense buffer
  size_t cap;                // candidate-list capacity per query: every row could pass, so it cannot overflow
  size_t off_dense, off_cand, off_count, off_thr, off_tau, off_slots, off_scnt, bytes;
};

// D2 = operand row length in 2-byte units
static SimPlan plan_sim(int B, int N, int D2, int k, int flags, bool fp8 = false) {
  SimPlan p;
  p.bm = (B > 128 && g_scan_cfg != 1) ? 256 : 128;
  p.tiles_q = (B + p.bm - 1) / p.bm;
  p.tiles_p = (N + 127) / 128;
  p.blocks = (N + SIM_PB - 1) / SIM_PB;
  // Sampling stride: the k-th best of a 1/stride sample leaves ~k*stride candidates above it in the
  // full set (spread ~stride*sqrt(k)), and the sample select reads N/stride keys per query.  The two
  // select costs balance near stride ~ sqrt(N/k) / 2 (measured: 16 at N = 130k, 32 at N = 1M for k = 100: the
  // filter pass pays an atomic append per candidate).
  int stride = 2;
  while (stride * 2 <= SIM_STRIDE_MAX && (int64_t)(stride * 2) * (stride * 2) * k * 4 <= N) stride *= 2;
  if (g_scan_stride > 1) stride = g_scan_stride;
  p.stride = stride;
  // small problems take the single dense pass; a shard of <= 16k rows still goes two-pass when many
  // queries share it (the 8-GPU shape: 2048 queries x 16k rows would write and re-read 266 MB of keys)
  p.dense_only = (flags & RP_TOPK_DENSE) != 0 || p.blocks < 2 * stride ||
                 (N <= SIM_DENSE_MAX_N && (int64_t)B * N <= SIM_DENSE_MAX_KEYS);
  // second-generation filter: needs whole 128-B operand rows per K-tile; its 256-query tile does the MFMA work of
  // 256 queries whatever B is, so small batches (single-state queries) stay on the first-generation kernel, whose
  // 128-query tiles are HBM-bound there (measured at B = 1: 142 vs 171 us per call)
  // (e4m3 rows may end half a k-tile early - 1472 bytes = 11.5 x 128: SimCfg8FilterTail)
  p.new_filter = !p.dense_only && g_scan_impl == 0 && (D2 % 64 == 0 || (fp8 && D2 % 64 == 32 && D2 > 64)) &&
                 (B > 128 || g_scan_impl_force_new);
  p.sample_blocks = p.dense_only ? 0 : (p.blocks + stride - 1) / stride;
  p.filter_blocks = p.dense_only ? 0 : p.blocks - p.sample_blocks;
  p.dense_ld = p.dense_only ? (size_t)p.tiles_p * 128 : (size_t)p.sample_blocks * SIM_PB;
  // Candidate-list capacity per query.  About k * stride keys lie above the sampled bound (spread ~ stride * sqrt(k)):
  // eight times that, never less than 8192 keys, never more than every row.  A query whose list would not fit (adversarial
  // scores, or a sample that gave no bound because fewer than k sampled rows were accessible) reports out_count = -1 and
  // the caller re-runs with RP_TOPK_DENSE - the contract of include/reprover_hip.h.  (Sized N + k, the list was
  // B * N * 8 bytes: 2 GB at 256 x 1M, as large as the dense plan the two-pass one exists to avoid.)
  p.cap = std::min((size_t)N + k, std::max((size_t)8192, (size_t)8 * k * stride) + k);
  if (g_scan_cap > 0) p.cap = std::min((size_t)N + k, (size_t)g_scan_cap + k);  // tests: force the overflow path
  size_t off = 0;
  p.off_dense = off;
  off += align_up((size_t)B * p.dense_ld * 8, 256);
  p.off_cand = off;
  off += p.dense_only ? 0 : align_up((size_t)B * p.cap * 8, 256);
  p.off_count = off;
  off += align_up((size_t)B * 4 * SIM_COUNT_STRIDE, 256);
  p.off_thr = off;
  off += align_up((size_t)B * 8, 256);
  p.off_tau = off;
  off += align_up((size_t)B * 4, 256);
  p.off_slots = off;  // second-generation filter kernel: private runs + their counts
  p.off_scnt = off;
  if (p.new_filter) {
    const size_t wg_tiles = (size_t)((B + 255) / 256) * p.filter_blocks;
    off += align_up(wg_tiles * SLOTS_PER_TILE * sizeof(uint2), 256);
    p.off_scnt = off;
    off += align_up(wg_tiles * 256 * 4 * sizeof(int32_t), 256);
  }
  p.bytes = off;
  return p;
}

// first-generation kernel (queries = MFMA rows): dense pass, sample pass, fallback filter pass
template <class C>
static RpStatus launch_scan_cfg(GemmOperand q, GemmOperand e, int D2, int n_tiles, int stride, const EpiSim& epi,
                                hipStream_t stream) {
  if (n_tiles <= 0) return RP_OK;
  static LdsAttrOnce attr;
  RP_HIP(attr.ensure((const void*)sim_scan_kernel<C>, C::LDS_BYTES));
  const int tiles_q = (epi.B + C::BM - 1) / C::BM;
  const int sub = (stride > 1 || epi.filter) ? SIM_PB / C::BN : 1;
  ProfScope ps(stream, (stride > 1 && !epi.filter) ? RP_K_SCAN_SAMPLE : RP_K_SCAN);
  hipLaunchKernelGGL((sim_scan_kernel<C>), dim3(tiles_q * n_tiles), dim3(C::THREADS), C::LDS_BYTES, stream, q, e, D2,
                     tiles_q, stride, sub, epi);
  RP_CHECK_LAUNCH();
  return RP_OK;
}

// the eight-wave form of a plain-loop tile (the four-wave one only in probe builds)
template <class C4, class C8>
static RpStatus launch_scan_w(GemmOperand q, GemmOperand e, int D2, int n_tiles, int stride, const EpiSim& epi, hipStream_t stream) {
  static_assert(C4::BM == C8::BM && C4::BN == C8::BN && C4::FP8 == C8::FP8, "the same tile");
#ifdef RP_EXPERIMENTS
  if (g_scan_waves == 4) return launch_scan_cfg<C4>(q, e, D2, n_tiles, stride, epi, stream);
#endif
  return launch_scan_cfg<C8>(q, e, D2, n_tiles, stride, epi, stream);
}

template <class C, bool PAGED>
static RpStatus launch_filter_cfg(GemmOperand e, GemmOperand q, int D2, int n_blocks, int stride,
                                  const EpiSimFilter<C::FP8, PAGED>& epi, hipStream_t stream) {
  if (n_blocks <= 0) return RP_OK;
  constexpr int LDS = C::RING_BYTES + SIM_FILTER_META_BYTES;
  static LdsAttrOnce attr;
  RP_HIP(attr.ensure((const void*)sim_filter_kernel<C, PAGED>, LDS));
  const int tiles_q = (epi.B + C::BN - 1) / C::BN;
  ProfScope ps(stream, RP_K_SCAN);
  hipLaunchKernelGGL((sim_filter_kernel<C, PAGED>), dim3(tiles_q * n_blocks), dim3(C::THREADS), LDS, stream, e, q, D2, tiles_q,
                     stride, epi);
  RP_CHECK_LAUNCH();
  return RP_OK;
}

}  // namespace rp

using namespace rp;

extern "C" size_t rp_sim_topk_workspace_bytes(int32_t B, int32_t N, int32_t D, int32_t k, int32_t flags) {
  if (B <= 0 || N <= 0 || k <= 0) return 0;
  // the e4m3 entry point has half the 2-byte units per row; its plan can only differ by taking the
  // first-generation filter kernel, which needs the same workspace
  return plan_sim(B, N, D, k, flags).bytes;
}

// shared by the bf16 and the e4m3 entry points (q_scale/e_scale NULL = bf16 operands)
static RpStatus sim_topk_impl(const void* Q, const void* E, const float* q_scale, const float* e_scale,
                              const float* after_score, const int32_t* after_id, int32_t B,
                              int32_t N, int32_t D, const int32_t* file_of, const int64_t* end_key,
                              const uint32_t* file_bits_t, int32_t F, const int32_t* own_file, const int64_t* q_key,
                              int32_t id_offset, int32_t k, int32_t flags, float* out_scores, int32_t* out_ids,
                              int32_t* out_count, void* workspace, size_t workspace_bytes, void* stream_) {
  RP_REQUIRE(Q && E && out_scores && out_ids && out_count, "null argument");
  const bool fp8 = q_scale != nullptr;
  if (fp8)
    RP_REQUIRE(B > 0 && N > 0 && D > 0 && D % 64 == 0, "B=%d N=%d D=%d (D must be a multiple of 64 for e4m3)", B, N, D);
  else
    RP_REQUIRE(B > 0 && N > 0 && D > 0 && D % 32 == 0, "B=%d N=%d D=%d (D must be a multiple of 32)", B, N, D);
  RP_REQUIRE(k > 0 && k <= SIM_MAX_K, "k=%d out of range (1..%d)", k, SIM_MAX_K);
  if (file_of) RP_REQUIRE(end_key && file_bits_t && own_file && q_key && F > 0, "mask arrays incomplete");
  hipStream_t stream = (hipStream_t)stream_;
  const int D2 = fp8 ? D / 2 : D;  // row length in 2-byte units
  const SimPlan p = plan_sim(B, N, D2, k, flags, fp8);
  if (!workspace || workspace_bytes < p.bytes)
    return fail(RP_E_WORKSPACE, "workspace %zu < required %zu bytes", workspace_bytes, p.bytes);
  char* ws = (char*)workspace;
  uint64_t* dense = (uint64_t*)(ws + p.off_dense);
  uint64_t* cand = (uint64_t*)(ws + p.off_cand);
  int32_t* count = (int32_t*)(ws + p.off_count);
  uint64_t* thr = (uint64_t*)(ws + p.off_thr);
  float* tau = (float*)(ws + p.off_tau);

  GemmOperand qop{(const bf16_t*)Q, D2, B}, eop{(const bf16_t*)E, D2, N};
  EpiSim epi;
  RP_REQUIRE((after_score == nullptr) == (after_id == nullptr), "after_score and after_id go together");
  epi.q_scale = q_scale;
  epi.e_scale = e_scale;
  epi.after_score = after_score;
  epi.after_id = after_id;
  epi.file_of = file_of;
  epi.end_key = end_key;
  epi.N = N;
  epi.bits_t = file_bits_t;
  epi.bits_words = (B + 31) / 32;
  epi.own_file = own_file;
  epi.q_key = q_key;
  epi.B = B;
  epi.id_offset = id_offset;
  epi.filter = 0;
  epi.dense = dense;
  epi.dense_ld = p.dense_ld;
  epi.slot_shift = 0;
  epi.thr = thr;
  epi.cand = cand;
  epi.cap = (int)p.cap;
  epi.count = count;
  epi.smem = nullptr;
  epi.tile_q0 = 0;
  epi.bm = p.bm;
  epi.debug_skip = g_scan_no_epilogue;

  RpStatus st;
  SelectArgs sa;
  sa.debug = 0;
  sa.keys = dense;
  sa.ld = p.dense_ld;
  sa.counts = nullptr;
  sa.count_stride = 1;
  sa.n_fixed = (int)p.dense_ld;
  sa.cap = (int)p.dense_ld;
  sa.k = k;
  sa.out_tau = nullptr;
  if (p.dense_only) {
    // single pass: dense keys of every premise tile, then one select
    if (fp8)
      st = (p.bm == 256) ? launch_scan_cfg<SimCfg8Q256>(qop, eop, D2, p.tiles_p, 1, epi, stream)
                         : launch_scan_w<SimCfg8Q128, SimCfg8Q128W8>(qop, eop, D2, p.tiles_p, 1, epi, stream);
    else if (p.bm == 256)
      st = launch_scan_cfg<SimCfgQ256>(qop, eop, D2, p.tiles_p, 1, epi, stream);
    else
      st = (D2 % 64 == 0) ? launch_scan_w<SimCfgQ128, SimCfgQ128W8>(qop, eop, D2, p.tiles_p, 1, epi, stream)
                          : launch_scan_w<SimCfgQ128K32, SimCfgQ128K32W8>(qop, eop, D2, p.tiles_p, 1, epi, stream);
    if (st) return st;
    sa.out_keys = nullptr;
    sa.out_ld = 0;
    sa.out_cnt = nullptr;
    sa.out_thr = nullptr;
    sa.out_scores = out_scores;
    sa.out_ids = out_ids;
    sa.out_count = out_count;
    launch_select(sa, B, stream);
    RP_CHECK_LAUNCH();
    return RP_OK;
  }
  // pass 0: dense keys of the sampled blocks (4 sub-tiles of 64 rows each), k best of the sample -> bound
  // (many queries - a shard under the queries of an 8-GPU step: the sample is a quarter of the rows there, MFMA-bound, and
  // the dense pass's 256-query x 128-row tile runs it faster than the 128 x 64 one - 45 vs 52 us, e4m3 34 vs 44 - once its
  // tiles fill the chip; with fewer it is slower: 41 vs 29 us at 1024 queries x 32.5 k rows)
  const int n_sub = p.sample_blocks * (SIM_PB / SimCfgSample::BN);
  const bool big_sample_tile = ((B + 255) / 256) * p.sample_blocks * (SIM_PB / SimCfgQ256::BN) >= 256;  // fills the chip
  st = big_sample_tile ? (fp8 ? launch_scan_cfg<SimCfg8Q256>(qop, eop, D2, p.sample_blocks * (SIM_PB / SimCfg8Q256::BN), p.stride, epi, stream)
                         : (D2 % 64 == 0)
                             ? launch_scan_cfg<SimCfgSampleBig>(qop, eop, D2, p.sample_blocks * (SIM_PB / SimCfgSampleBig::BN), p.stride, epi, stream)
                             : launch_scan_cfg<SimCfgQ256>(qop, eop, D2, p.sample_blocks * (SIM_PB / SimCfgQ256::BN), p.stride, epi, stream))
       : (fp8 && B <= 32 && D2 % 64 == 0 && g_scan_small_tiles) ? launch_scan_cfg<SimCfg8SampleQ32>(qop, eop, D2, n_sub, p.stride, epi, stream)
       : (fp8 && B <= 32 && D2 % 64 == 32 && D2 > 64 && g_scan_small_tiles)
           ? launch_scan_cfg<SimCfg8SampleQ32Tail>(qop, eop, D2, n_sub, p.stride, epi, stream)
       : (fp8 && D2 % 64 == 0) ? launch_scan_w<SimCfg8Sample, SimCfg8SampleK64W8>(qop, eop, D2, n_sub, p.stride, epi, stream)
       : (fp8 && D2 % 64 == 32 && D2 > 64) ? launch_scan_w<SimCfg8Sample, SimCfg8SampleK64TailW8>(qop, eop, D2, n_sub, p.stride, epi, stream)
       : fp8 ? launch_scan_cfg<SimCfg8Sample>(qop, eop, D2, n_sub, p.stride, epi, stream)
       : (g_scan_sample_cfg == 0 && g_scan_small_tiles && B <= 32 && D2 % 64 == 0) ? launch_scan_cfg<SimCfgSampleQ32>(qop, eop, D2, n_sub, p.stride, epi, stream)
       : (g_scan_sample_cfg == 0 && D2 % 64 == 0)
           ? launch_scan_w<SimCfgSampleK64, SimCfgSampleK64W8>(qop, eop, D2, n_sub, p.stride, epi, stream)
           : launch_scan_cfg<SimCfgSample>(qop, eop, D2, n_sub, p.stride, epi, stream);  // (D % 64 != 0: 32-wide K slices)
  if (st) return st;
  sa.out_keys = cand;
  sa.out_ld = p.cap;
  sa.out_cnt = count;
  sa.out_thr = thr;
  sa.out_tau = tau;
  sa.out_scores = nullptr;
  sa.out_ids = nullptr;
  sa.out_count = nullptr;
  sa.debug = g_scan_no_epilogue;
  launch_select(sa, B, stream);
  RP_CHECK_LAUNCH();
  // pass 1: remaining blocks, keep only keys above each query's bound
  if (p.new_filter) {
    auto fill = [&](auto& ef) {
      ef.N = N;
      ef.B = B;
      ef.q_scale = q_scale;
      ef.e_scale = e_scale;
      ef.tau = tau;
      ef.slots = (uint2*)(ws + p.off_slots);
      ef.scnt = (int32_t*)(ws + p.off_scnt);
      ef.filter_blocks = p.filter_blocks;
      ef.stride = p.stride;
      ef.tiles_q = (B + 255) / 256;
      ef.smem = nullptr;
      ef.meta_off = 0;
      ef.p0 = 0;
      ef.debug_drop_all = g_scan_no_epilogue;
    };
    // (bf16 index: the 8-wave tile; the 4-wave forms it replaced stay selectable in probe builds, scan_filter_cfg)
    if (fp8 && after_id) {  // a later page: the bound joins the pre-test (EpiSimFilter<., PAGED>)
      EpiSimFilter<1, true> ef;
      fill(ef);
      ef.after_score = after_score;
      ef.after_id = after_id;
      st = (D2 % 64 == 0) ? launch_filter_cfg<SimCfg8FilterW8, true>(eop, qop, D2, p.filter_blocks, p.stride, ef, stream)
                          : launch_filter_cfg<SimCfg8FilterTailW8, true>(eop, qop, D2, p.filter_blocks, p.stride, ef, stream);
    } else if (fp8) {
      EpiSimFilter<1> ef;
      fill(ef);
#ifdef RP_EXPERIMENTS
      if (g_scan_filter_cfg == 4 && D2 % 64 == 0)
        st = launch_filter_cfg<SimCfg8Filter, false>(eop, qop, D2, p.filter_blocks, p.stride, ef, stream);
      else if (g_scan_filter_cfg == 4 && D2 % 64 != 0)
        st = launch_filter_cfg<SimCfg8FilterTail, false>(eop, qop, D2, p.filter_blocks, p.stride, ef, stream);
      else
#endif
      st = (D2 % 64 == 0) ? launch_filter_cfg<SimCfg8FilterW8, false>(eop, qop, D2, p.filter_blocks, p.stride, ef, stream)
                          : launch_filter_cfg<SimCfg8FilterTailW8, false>(eop, qop, D2, p.filter_blocks, p.stride, ef, stream);
    } else if (after_id) {
      EpiSimFilter<0, true> ef;
      fill(ef);
      ef.after_score = after_score;
      ef.after_id = after_id;
      st = launch_filter_cfg<SimCfgFilterNt8, true>(eop, qop, D2, p.filter_blocks, p.stride, ef, stream);
    } else {
      EpiSimFilter<0> ef;
      fill(ef);
#ifdef RP_EXPERIMENTS  // alternatives measured and rejected: only a probe build (RP_EXPERIMENTS=1) carries them
      if (g_scan_filter_cfg == 1)
        st = launch_filter_cfg<SimCfgFilterK32, false>(eop, qop, D2, p.filter_blocks, p.stride, ef, stream);
      else if (g_scan_filter_cfg == 2)
        st = launch_filter_cfg<SimCfgFilter, false>(eop, qop, D2, p.filter_blocks, p.stride, ef, stream);
      else if (g_scan_filter_cfg == 4)
        st = launch_filter_cfg<SimCfgFilterNt, false>(eop, qop, D2, p.filter_blocks, p.stride, ef, stream);
      else
#endif
        st = launch_filter_cfg<SimCfgFilterNt8, false>(eop, qop, D2, p.filter_blocks, p.stride, ef, stream);
    }
  } else {
    epi.filter = 1;
    const int n_t = p.filter_blocks * (SIM_PB / 128);
    if (fp8 && B <= 32 && D2 % 64 == 0 && g_scan_small_tiles)
      st = launch_scan_cfg<SimCfg8Q32>(qop, eop, D2, n_t, p.stride, epi, stream);
    else if (fp8 && B <= 32 && D2 % 64 == 32 && D2 > 64 && g_scan_small_tiles)
      st = launch_scan_cfg<SimCfg8Q32Tail>(qop, eop, D2, n_t, p.stride, epi, stream);
    else if (fp8 && B <= 64 && D2 % 64 == 0 && g_scan_small_tiles)
      st = launch_scan_cfg<SimCfg8Q64>(qop, eop, D2, n_t, p.stride, epi, stream);
    else if (fp8 && B <= 64 && D2 % 64 == 32 && D2 > 64 && g_scan_small_tiles)
      st = launch_scan_cfg<SimCfg8Q64Tail>(qop, eop, D2, n_t, p.stride, epi, stream);
    else if (fp8)
      st = (p.bm == 256) ? launch_scan_cfg<SimCfg8Q256>(qop, eop, D2, n_t, p.stride, epi, stream)
                         : launch_scan_w<SimCfg8Q128, SimCfg8Q128W8>(qop, eop, D2, n_t, p.stride, epi, stream);
    else if (p.bm == 256)
      st = launch_scan_cfg<SimCfgQ256>(qop, eop, D2, n_t, p.stride, epi, stream);
#ifdef RP_EXPERIMENTS
    else if (B <= 32 && D2 % 64 == 0 && g_scan_filter_cfg == 9)
      st = launch_scan_cfg<SimCfgQ32S3>(qop, eop, D2, n_t, p.stride, epi, stream);
    else if (B <= 32 && D2 % 64 == 0 && g_scan_filter_cfg == 10)
      st = launch_scan_cfg<SimCfgQ32P256>(qop, eop, D2, p.filter_blocks, p.stride, epi, stream);
#endif
    else if (B <= 32 && D2 % 64 == 0 && g_scan_small_tiles)
      st = launch_scan_cfg<SimCfgQ32>(qop, eop, D2, n_t, p.stride, epi, stream);
    else if (B <= 64 && D2 % 64 == 0 && g_scan_small_tiles)
      st = launch_scan_cfg<SimCfgQ64>(qop, eop, D2, n_t, p.stride, epi, stream);
    else
      st = (D2 % 64 == 0) ? launch_scan_w<SimCfgQ128, SimCfgQ128W8>(qop, eop, D2, n_t, p.stride, epi, stream)
                          : launch_scan_w<SimCfgQ128K32, SimCfgQ128K32W8>(qop, eop, D2, n_t, p.stride, epi, stream);
  }
  if (st) return st;
  GatherArgs ga;
  if (p.new_filter) {
    ga.slots = (const uint2*)(ws + p.off_slots);
    ga.scnt = (const int32_t*)(ws + p.off_scnt);
    ga.filter_blocks = p.filter_blocks;
    ga.tiles_q = (B + 255) / 256;
    ga.stride = p.stride;
    ga.N = N;
    ga.B = B;
    ga.id_offset = id_offset;
    ga.file_of = file_of;
    ga.end_key = end_key;
    ga.bits_t = file_bits_t;
    ga.bits_words = (B + 31) / 32;
    ga.own_file = own_file;
    ga.q_key = q_key;
    ga.thr = thr;
    ga.after_score = after_score;
    ga.after_id = after_id;
    ga.cand = cand;
    ga.cap = p.cap;
    ga.count = count;
    ga.debug = g_scan_no_epilogue;
  }
  SelectArgs sb;
  sb.keys = cand;
  sb.ld = p.cap;
  sb.counts = count;
  sb.count_stride = SIM_COUNT_STRIDE;
  sb.n_fixed = 0;
  sb.cap = (int)p.cap;
  sb.k = k;
  sb.out_keys = nullptr;
  sb.out_ld = 0;
  sb.out_cnt = nullptr;
  sb.out_thr = nullptr;
  sb.out_tau = nullptr;
  sb.out_scores = out_scores;
  sb.out_ids = out_ids;
  sb.out_count = out_count;
  sb.debug = g_scan_no_epilogue;
  if (p.new_filter) {  // runs -> predicate -> key list -> select, one kernel
    ProfScope ps(stream, RP_K_SELECT);
    // ~k * stride keys lie above the sampled bound, ~3 x that before the accessibility predicate: the smallest shape whose
    // raw list holds that when many queries share the chip (longer lists stay correct - they spill to the slow path)
    const int64_t raw_bound = (int64_t)k * p.stride * 4;
    if (B >= 512 && k <= SELECT_SMALL_MAX_K && raw_bound <= GATHER_ENTRIES_SMALL)
      hipLaunchKernelGGL((gather_select_kernel<SELECT_THREADS_SMALL, GATHER_ENTRIES_SMALL, SELECT_SMALL_MAX_K>), dim3(B),
                         dim3(SELECT_THREADS_SMALL), 0, stream, ga, sb);
    else if (B >= 512 && k <= SELECT_SMALL_MAX_K && raw_bound <= GATHER_ENTRIES_MID)
      hipLaunchKernelGGL((gather_select_kernel<GATHER_THREADS_MID, GATHER_ENTRIES_MID, SELECT_SMALL_MAX_K>), dim3(B),
                         dim3(GATHER_THREADS_MID), 0, stream, ga, sb);
    else
      hipLaunchKernelGGL((gather_select_kernel<SELECT_THREADS, GATHER_ENTRIES, SIM_MAX_K>), dim3(B), dim3(SELECT_THREADS), 0,
                         stream, ga, sb);
  } else {
    launch_select(sb, B, stream);
  }
  RP_CHECK_LAUNCH();
  return RP_OK;
}

extern "C" RpStatus rp_sim_topk(const void* Q, const void* E, int32_t B, int32_t N, int32_t D,
                                const int32_t* file_of, const int64_t* end_key, const uint32_t* file_bits_t,
                                int32_t F, const int32_t* own_file, const int64_t* q_key, int32_t id_offset,
                                int32_t k, int32_t flags, float* out_scores, int32_t* out_ids, int32_t* out_count,
                                void* workspace, size_t workspace_bytes, void* stream_) {
  return sim_topk_impl(Q, E, nullptr, nullptr, nullptr, nullptr, B, N, D, file_of, end_key, file_bits_t, F, own_file, q_key,
                       id_offset, k, flags, out_scores, out_ids, out_count, workspace, workspace_bytes, stream_);
}

// The next page of the same ranking: only premises that come strictly AFTER (after_score[q], after_id[q]) in the (score
// descending, id ascending) order qualify - what a caller asks for when k exceeds the 1024 keys one call can sort.
extern "C" RpStatus rp_sim_topk_after(const void* Q, const void* E, int32_t B, int32_t N, int32_t D,
                                      const int32_t* file_of, const int64_t* end_key, const uint32_t* file_bits_t,
                                      int32_t F, const int32_t* own_file, const int64_t* q_key, int32_t id_offset,
                                      const float* after_score, const int32_t* after_id, int32_t k, int32_t flags,
                                      float* out_scores, int32_t* out_ids, int32_t* out_count, void* workspace,
                                      size_t workspace_bytes, void* stream_) {
  RP_REQUIRE(after_score && after_id, "null paging arrays (use rp_sim_topk for the first page)");
  return sim_topk_impl(Q, E, nullptr, nullptr, after_score, after_id, B, N, D, file_of, end_key, file_bits_t, F, own_file,
                       q_key, id_offset, k, flags, out_scores, out_ids, out_count, workspace, workspace_bytes, stream_);
}

extern "C" RpStatus rp_sim_topk_fp8(const void* Q8, const float* q_scale, const void* E8, const float* e_scale,
                                    int32_t B, int32_t N, int32_t D, const int32_t* file_of, const int64_t* end_key,
                                    const uint32_t* file_bits_t, int32_t F, const int32_t* own_file,
                                    const int64_t* q_key, int32_t id_offset, int32_t k, int32_t flags,
                                    float* out_scores, int32_t* out_ids, int32_t* out_count, void* workspace,
                                    size_t workspace_bytes, void* stream_) {
  RP_REQUIRE(q_scale && e_scale, "null scale array");
  return sim_topk_impl(Q8, E8, q_scale, e_scale, nullptr, nullptr, B, N, D, file_of, end_key, file_bits_t, F, own_file,
                       q_key, id_offset, k, flags, out_scores, out_ids, out_count, workspace, workspace_bytes, stream_);
}

extern "C" RpStatus rp_sim_topk_fp8_after(const void* Q8, const float* q_scale, const void* E8, const float* e_scale,
                                          int32_t B, int32_t N, int32_t D, const int32_t* file_of,
                                          const int64_t* end_key, const uint32_t* file_bits_t, int32_t F,
                                          const int32_t* own_file, const int64_t* q_key, int32_t id_offset,
                                          const float* after_score, const int32_t* after_id, int32_t k, int32_t flags,
                                          float* out_scores, int32_t* out_ids, int32_t* out_count, void* workspace,
                                          size_t workspace_bytes, void* stream_) {
  RP_REQUIRE(q_scale && e_scale, "null scale array");
  RP_REQUIRE(after_score && after_id, "null paging arrays (use rp_sim_topk_fp8 for the first page)");
  return sim_topk_impl(Q8, E8, q_scale, e_scale, after_score, after_id, B, N, D, file_of, end_key, file_bits_t, F, own_file,
                       q_key, id_offset, k, flags, out_scores, out_ids, out_count, workspace, workspace_bytes, stream_);
}

// ------------------------------------------------------------------------------------------
// Row-wise e4m3 quantisation of an embedding matrix: scale[r] = max|x[r,:]| / 448 (1 if the row is
// zero), q = RNE_e4m3(x * (448 / max|x|)).  OCP e4m3fn: 4 exponent bits (bias 7), 3 mantissa bits,
// max 448, subnormals k * 2^-9.  The rounding is done in integer arithmetic so the oracle
// (oracle/fp8_ref.py) can restate it bit for bit.
// ------------------------------------------------------------------------------------------
namespace rp {
__device__ __forceinline__ uint32_t f32_to_e4m3(float y) {
  const uint32_t sign = (__float_as_uint(y) >> 24) & 0x80u;
  const float a = fminf(fabsf(y), 448.f);
  uint32_t code;
  if (a >= 0.015625f) {  // normal range [2^-6, 448]: round the fp32 mantissa to 3 bits, nearest-even
    uint32_t u = __float_as_uint(a);
    u += 0x7FFFFu + ((u >> 20) & 1u);
    code = (((u >> 23) - 120u) << 3) | ((u >> 20) & 7u);  // exponent re-biased 127 -> 7
    code = min(code, 0x7Eu);
  } else {
    code = (uint32_t)rintf(a * 512.f);  // subnormals: multiples of 2^-9; 8 = the smallest normal
  }
  return sign | code;
}

template <typename T>
__global__ __launch_bounds__(256) void quantize_rows_kernel(const T* __restrict__ X, int64_t rows, int D,
                                                            uint8_t* __restrict__ out, float* __restrict__ scale) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const T* x = X + r * D;
  float amax = 0.f;
  for (int c = lane * 4; c < D; c += 256) {
#pragma unroll
    for (int e = 0; e < 4; ++e) amax = fmaxf(amax, fabsf(to_f32(x[c + e])));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
  const float inv = amax > 0.f ? 448.f / amax : 0.f;
  if (lane == 0) scale[r] = amax > 0.f ? amax / 448.f : 1.f;
  for (int c = lane * 4; c < D; c += 256) {
    uint32_t w = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) w |= f32_to_e4m3(to_f32(x[c + e]) * inv) << (8 * e);
    *reinterpret_cast<uint32_t*>(out + r * D + c) = w;
  }
}
}  // namespace rp

extern "C" RpStatus rp_quantize_rows_e4m3(const void* X, int32_t x_dtype, int64_t rows, int32_t D, void* out_fp8,
                                          float* out_scale, void* stream_) {
  RP_REQUIRE(X && out_fp8 && out_scale, "null argument");
  RP_REQUIRE(rows > 0 && D > 0 && D % 4 == 0, "rows=%lld D=%d (D must be a multiple of 4)", (long long)rows, D);
  RP_REQUIRE(x_dtype == RP_DT_F32 || x_dtype == RP_DT_BF16, "x_dtype %d", x_dtype);
  hipStream_t stream = (hipStream_t)stream_;
  const dim3 grid((unsigned)((rows + 3) / 4));
  if (x_dtype == RP_DT_F32)
    hipLaunchKernelGGL(quantize_rows_kernel<float>, grid, dim3(256), 0, stream, (const float*)X, (int64_t)rows, D,
                       (uint8_t*)out_fp8, out_scale);
  else
    hipLaunchKernelGGL(quantize_rows_kernel<bf16_t>, grid, dim3(256), 0, stream, (const bf16_t*)X, (int64_t)rows, D,
                       (uint8_t*)out_fp8, out_scale);
  RP_CHECK_LAUNCH();
  return RP_OK;
}

// ------------------------------------------------------------------------------------------
// Training forward, loss part (retrieval/model.py:116-140): similarity = context_emb @ all_premise_embs.T,
// loss = F.mse_loss(similarity, label) = mean over all B x P entries of (similarity - label)^2.
// B x P is small (batch x batch * (1 + negatives)): one wave per pair for the fp32 dot product, then ONE workgroup
// sums the squared errors in index order - deterministic, no atomics.
// ------------------------------------------------------------------------------------------
namespace rp {
__global__ __launch_bounds__(256) void pair_dots_kernel(const float* __restrict__ ctx, const float* __restrict__ prem,
                                                        const float* __restrict__ label, int B, int P, int D,
                                                        float* __restrict__ sim, float* __restrict__ err2) {
  const int lane = threadIdx.x & 63;
  const int pair = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (pair >= B * P) return;
  const int j = pair / P, k = pair % P;
  const float4* a = reinterpret_cast<const float4*>(ctx + (size_t)j * D);
  const float4* b = reinterpret_cast<const float4*>(prem + (size_t)k * D);
  float acc = 0.f;
  for (int c = lane; c < (D >> 2); c += 64) {
    const float4 x = a[c], y = b[c];
    acc = fmaf(x.x, y.x, fmaf(x.y, y.y, fmaf(x.z, y.z, fmaf(x.w, y.w, acc))));
  }
  acc = wave_sum(acc);
  if (lane == 0) {
    if (sim) sim[pair] = acc;
    const float d = acc - label[pair];
    err2[pair] = d * d;
  }
}

__global__ __launch_bounds__(256) void mean_kernel(const float* __restrict__ v, int n, float* __restrict__ out) {
  __shared__ float part[256];
  const int per = (n + 255) / 256, lo = threadIdx.x * per, hi = min(lo + per, n);
  float s = 0.f;
  for (int i = lo; i < hi; ++i) s += v[i];  // contiguous chunk per thread, index order
  part[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) part[threadIdx.x] += part[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = part[0] / (float)n;
}
}  // namespace rp

extern "C" size_t rp_contrastive_mse_workspace_bytes(int32_t B, int32_t P) {
  if (B <= 0 || P <= 0) return 0;
  return align_up((size_t)B * P * 4, 256);
}

extern "C" RpStatus rp_contrastive_mse(const float* context_emb, const float* premise_embs, const float* label, int32_t B,
                                       int32_t P, int32_t D, float* out_loss, float* out_similarity, void* workspace,
                                       size_t workspace_bytes, void* stream_) {
  RP_REQUIRE(context_emb && premise_embs && label && out_loss, "null argument");
  RP_REQUIRE(B > 0 && P > 0 && D > 0 && D % 4 == 0, "B=%d P=%d D=%d (D must be a multiple of 4)", B, P, D);
  const size_t need = rp_contrastive_mse_workspace_bytes(B, P);
  if (!workspace || workspace_bytes < need)
    return fail(RP_E_WORKSPACE, "workspace %zu < required %zu bytes", workspace_bytes, need);
  hipStream_t stream = (hipStream_t)stream_;
  float* err2 = (float*)workspace;
  hipLaunchKernelGGL(pair_dots_kernel, dim3((B * P + 3) / 4), dim3(256), 0, stream, context_emb, premise_embs, label, B,
                     P, D, out_similarity, err2);
  hipLaunchKernelGGL(mean_kernel, dim3(1), dim3(256), 0, stream, (const float*)err2, B * P, out_loss);
  RP_CHECK_LAUNCH();
  return RP_OK;
}

// ------------------------------------------------------------------------------------------
// First pieces of the training step behind the forward (oracle: oracle/train_ref.py, fixture G11).
//   * rp_contrastive_mse_backward: loss = mean((C P^T - label)^2)  =>  dS = 2 (S - label) / (B P),
//     dC = dS P, dP = dS^T C.  One workgroup per output row, the other index walked in order (deterministic).
//   * rp_adamw_step: torch.optim.AdamW's update (common.py:395 `torch.optim.AdamW(parameters, lr=lr)`), in place on
//     fp32 parameters and moments; 28 bytes per parameter, HBM-bound.
// The encoder's own backward is not built.
// ------------------------------------------------------------------------------------------
namespace rp {
__global__ __launch_bounds__(256) void mse_backward_kernel(const float* __restrict__ ctx, const float* __restrict__ prem,
                                                           const float* __restrict__ sim, const float* __restrict__ label,
                                                           int B, int P, int D, float* __restrict__ d_ctx,
                                                           float* __restrict__ d_prem) {
  const int row = blockIdx.x;  // [0, B): a context row; [B, B + P): a premise row
  const float scale = 2.f / ((float)B * (float)P);
  const bool is_ctx = row < B;
  const int r = is_ctx ? row : row - B;
  const int n_other = is_ctx ? P : B;
  const float* other = is_ctx ? prem : ctx;
  float* dst = (is_ctx ? d_ctx : d_prem) + (size_t)r * D;
  for (int c = threadIdx.x; c < D; c += 256) {
    float acc = 0.f;
    for (int o0 = 0; o0 < n_other; o0 += 8) {  // eight rows of the other side requested, then accumulated in index order
      float ds[8], ov[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int o = min(o0 + u, n_other - 1);
        const int j = is_ctx ? r : o, k = is_ctx ? o : r;
        ds[u] = (sim[(size_t)j * P + k] - label[(size_t)j * P + k]) * scale;
        ov[u] = other[(size_t)o * D + c];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (o0 + u < n_other) acc = fmaf(ds[u], ov[u], acc);
    }
    dst[c] = acc;
  }
}

__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                    float* __restrict__ m, float* __restrict__ v, size_t n4, size_t n,
                                                    float decay, float beta1, float beta2, float step_size,
                                                    float inv_sqrt_bc2, float eps, const float* __restrict__ total_norm,
                                                    float max_norm) {
  // torch.nn.utils.clip_grad_norm_: grads *= clamp(max_norm / (total_norm + 1e-6), max = 1)
  const float clip = total_norm ? fminf(max_norm / (total_norm[0] + 1e-6f), 1.f) : 1.f;
  auto upd = [&](float& pp, float gg, float& mm, float& vv) {
    if (total_norm) gg *= clip;
    pp *= decay;                              // param.mul_(1 - lr * weight_decay)
    mm = fmaf(gg - mm, 1.f - beta1, mm);      // exp_avg.lerp_(grad, 1 - beta1)
    vv = fmaf(gg * gg, 1.f - beta2, vv * beta2);
    const float denom = fmaf(sqrtf(vv), inv_sqrt_bc2, eps);
    pp -= step_size * (mm / denom);
  };
  const size_t stride = (size_t)gridDim.x * 256;
  constexpr int U = 2;  // two 16-byte pieces of each of the four arrays in flight per thread
  for (size_t i0 = (size_t)blockIdx.x * 256 + threadIdx.x; i0 < n4; i0 += U * stride) {
    float4 pp[U], mm[U], vv[U], gg[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t i = min(i0 + u * stride, n4 - 1);
      pp[u] = reinterpret_cast<float4*>(p)[i];
      mm[u] = reinterpret_cast<float4*>(m)[i];
      vv[u] = reinterpret_cast<float4*>(v)[i];
      gg[u] = reinterpret_cast<const float4*>(g)[i];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t i = i0 + u * stride;
      if (i >= n4) break;
      upd(pp[u].x, gg[u].x, mm[u].x, vv[u].x);
      upd(pp[u].y, gg[u].y, mm[u].y, vv[u].y);
      upd(pp[u].z, gg[u].z, mm[u].z, vv[u].z);
      upd(pp[u].w, gg[u].w, mm[u].w, vv[u].w);
      reinterpret_cast<float4*>(p)[i] = pp[u];
      reinterpret_cast<float4*>(m)[i] = mm[u];
      reinterpret_cast<float4*>(v)[i] = vv[u];
    }
  }
  if (blockIdx.x == 0 && threadIdx.x < (unsigned)(n - n4 * 4)) {  // tail of fewer than four elements
    const size_t i = n4 * 4 + threadIdx.x;
    upd(p[i], g[i], m[i], v[i]);
  }
}
}  // namespace rp

extern "C" RpStatus rp_contrastive_mse_backward(const float* context_emb, const float* premise_embs,
                                                const float* similarity, const float* label, int32_t B, int32_t P,
                                                int32_t D, float* d_context_emb, float* d_premise_embs, void* stream_) {
  RP_REQUIRE(context_emb && premise_embs && similarity && label && d_context_emb && d_premise_embs, "null argument");
  RP_REQUIRE(B > 0 && P > 0 && D > 0, "B=%d P=%d D=%d", B, P, D);
  hipLaunchKernelGGL(mse_backward_kernel, dim3(B + P), dim3(256), 0, (hipStream_t)stream_, context_emb, premise_embs,
                     similarity, label, B, P, D, d_context_emb, d_premise_embs);
  RP_CHECK_LAUNCH();
  return RP_OK;
}

extern "C" RpStatus rp_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                                  int32_t step, float lr, float beta1, float beta2, float eps, float weight_decay,
                                  void* stream_) {
  return rp_adamw_step_clipped(param, grad, exp_avg, exp_avg_sq, n, step, lr, beta1, beta2, eps, weight_decay, nullptr, 0.f,
                               stream_);
}

extern "C" RpStatus rp_adamw_step_clipped(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                                          int32_t step, float lr, float beta1, float beta2, float eps, float weight_decay,
                                          const float* total_norm, float max_norm, void* stream_) {
  RP_REQUIRE(param && grad && exp_avg && exp_avg_sq, "null argument");
  ProfScope ps((hipStream_t)stream_, RP_K_OPTIMIZER);
  RP_REQUIRE(n > 0 && step >= 1 && beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f, "n=%lld step=%d",
             (long long)n, step);
  RP_REQUIRE((((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) == 0,
             "16-byte aligned arrays");
  const double bc1 = 1.0 - std::pow((double)beta1, (double)step), bc2 = 1.0 - std::pow((double)beta2, (double)step);
  const size_t n4 = (size_t)n / 4;
  const unsigned grid = (unsigned)std::min<size_t>(std::max<size_t>((n4 + 511) / 512, 1), 256 * 32);
  hipLaunchKernelGGL(adamw_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream_, param, grad, exp_avg, exp_avg_sq, n4,
                     (size_t)n, 1.f - lr * weight_decay, beta1, beta2, (float)((double)lr / bc1),
                     (float)(1.0 / std::sqrt(bc2)), eps, total_norm, max_norm);
  RP_CHECK_LAUNCH();
  return RP_OK;
}

extern "C" size_t rp_topk_merge_workspace_bytes(int32_t R, int32_t B, int32_t k) {
  if (R <= 0 || B <= 0 || k <= 0) return 0;
  return align_up((size_t)R * B * k * 8, 256);
}

extern "C" RpStatus rp_build_file_bits(const uint64_t* reach, int32_t F, const int32_t* own_file, int32_t B,
                                       uint32_t* file_bits_t, void* stream_) {
  RP_REQUIRE(reach && own_file && file_bits_t && F > 0 && B > 0, "bad argument");
  const int Wq = (B + 31) / 32, W64 = (F + 63) / 64;
  hipLaunchKernelGGL(build_file_bits_kernel, dim3((F * Wq + 255) / 256), dim3(256), 0, (hipStream_t)stream_, reach, F, W64,
                     own_file, B, Wq, file_bits_t);
  RP_CHECK_LAUNCH();
  return RP_OK;
}

static RpStatus topk_merge_impl(const float* scores, const int32_t* ids, const int32_t* counts, int64_t rank_stride,
                                int32_t R, int32_t B, int32_t k, float* out_scores, int32_t* out_ids, int32_t* out_count,
                                void* workspace, size_t workspace_bytes, void* stream_);

extern "C" RpStatus rp_topk_merge(const float* scores, const int32_t* ids, const int32_t* counts, int32_t R,
                                  int32_t B, int32_t k, float* out_scores, int32_t* out_ids, int32_t* out_count,
                                  void* workspace, size_t workspace_bytes, void* stream_) {
  return topk_merge_impl(scores, ids, counts, -1, R, B, k, out_scores, out_ids, out_count, workspace, workspace_bytes, stream_);
}

extern "C" RpStatus rp_topk_merge_strided(const float* scores, const int32_t* ids, const int32_t* counts,
                                          int64_t rank_stride, int32_t R, int32_t B, int32_t k, float* out_scores,
                                          int32_t* out_ids, int32_t* out_count, void* workspace, size_t workspace_bytes,
                                          void* stream_) {
  RP_REQUIRE(rank_stride > 0, "rank_stride=%lld", (long long)rank_stride);
  return topk_merge_impl(scores, ids, counts, rank_stride, R, B, k, out_scores, out_ids, out_count, workspace, workspace_bytes,
                         stream_);
}

static RpStatus topk_merge_impl(const float* scores, const int32_t* ids, const int32_t* counts, int64_t rank_stride,
                                int32_t R, int32_t B, int32_t k, float* out_scores, int32_t* out_ids, int32_t* out_count,
                                void* workspace, size_t workspace_bytes, void* stream_) {
  RP_REQUIRE(scores && ids && counts && out_scores && out_ids && out_count, "null argument");
  RP_REQUIRE(R > 0 && B > 0 && k > 0 && k <= SIM_MAX_K, "R=%d B=%d k=%d", R, B, k);
  const size_t need = rp_topk_merge_workspace_bytes(R, B, k);
  if (!workspace || workspace_bytes < need)
    return fail(RP_E_WORKSPACE, "workspace %zu < required %zu bytes", workspace_bytes, need);
  hipStream_t stream = (hipStream_t)stream_;
  uint64_t* keys = (uint64_t*)workspace;
  if (rank_stride < 0)
    hipLaunchKernelGGL(merge_keys_kernel, dim3(B), dim3(256), 0, stream, scores, ids, counts, R, B, k, keys);
  else
    hipLaunchKernelGGL(merge_keys_strided_kernel, dim3(B), dim3(256), 0, stream, scores, ids, counts, (size_t)rank_stride, R,
                       k, keys);
  RP_CHECK_LAUNCH();
  SelectArgs sa;
  sa.debug = 0;
  sa.keys = keys;
  sa.ld = (size_t)R * k;
  sa.counts = nullptr;
  sa.count_stride = 1;
  sa.n_fixed = R * k;
  sa.cap = R * k;
  sa.k = k;
  sa.out_keys = nullptr;
  sa.out_ld = 0;
  sa.out_cnt = nullptr;
  sa.out_thr = nullptr;
  sa.out_scores = out_scores;
  sa.out_ids = out_ids;
  sa.out_count = out_count;
  launch_select(sa, B, stream);
  RP_CHECK_LAUNCH();
  return RP_OK;
}

#ifdef RP_PHASE_PROBE  // probe builds only (tools/probes/scan_phase.py): the host-side reader of the filter pass's phase timestamps
#define RP_PROBE_EXPORT_SCAN
#include "probes/rp_probe_exports.h"
#endif
