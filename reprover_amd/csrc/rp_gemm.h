// bf16 MFMA GEMM core for gfx950:  C[M,N] = A[M,K] · W[N,K]^T   (both operands K-contiguous,
// i.e. activations row-major and nn.Linear weights as stored), fp32 accumulate.
//
// Workgroup = WM x WN waves over a BM x BN x BK block tile; each wave owns an (FM*32) x (FN*32)
// sub-tile as FM x FN v_mfma_f32_32x32x16_bf16 accumulators (16 fp32 registers each).
//
// Staging: A and W tiles go HBM/L2 -> LDS by LDS-DMA (global_load_lds_dwordx4: no VGPR round
// trip, 1 KiB per wave-instruction) into an NSTAGE-deep ring, with a COUNTED s_waitcnt vmcnt(...)
// and a raw s_barrier per K-tile so the DMAs stay in flight across barriers
// (cdna_hip_programming.md §5 "Pipelining across barriers").  The LDS image is lane-linear per DMA
// instruction, so the bank-conflict swizzle is applied to the per-lane SOURCE address and again on
// the ds_read_b128 fragment reads: 16-B slot index ^= f(row), with f chosen per BK so that the
// 16-lane groups ds_read_b128 is serviced in hit 16 distinct slots of the 256-B bank row.
//
// Two main loops:
//   gemm_tile       the plain loop (compiler-scheduled): small tiles with several blocks per CU, the
//                   few-token configurations, the similarity scan (bf16 or e4m3 operands);
//   gemm_tile_pipe  the hand-software-pipelined loop of the encoder's big GEMMs (GemmCfg<..., PIPE=1>);
//   gemm_tiles_persist  the same loop for ONE workgroup per CU walking a list of tiles (next tile's first k-tile
//                   requested under the epilogue).
// Both pipelined forms take a WaveLayout: how the waves split the block tile.  The last tile of an M extent that ends
// inside it runs on a wave grid over its valid rows only (the LDS image, the DMA split and the k loop do not change).
//
// The epilogue is a functor so the same core serves the encoder GEMMs (bf16 store, fp32
// residual add, gated-GELU) and the similarity scan (accessibility mask + top-k filter).
#pragma once
#include <type_traits>
#include "rp_util.h"

namespace rp {

// Timing probes (tools/probes/gemm_phase.py builds a second copy of the library with -DRP_PHASE_PROBE, optionally with
// -DRP_PROBE_NO_DMA / _HALF_DMA / _SAME_TILE / _NO_READS and -DRP_ABL_*): their bodies live in probes/rp_probe_hooks.h, which
// the product build never includes - here every hook is an empty statement.
#ifdef RP_PHASE_PROBE
#include "probes/rp_probe_hooks.h"
#else
#define RP_TS(slot) ((void)0)
#define RP_TS_PLAIN(slot) ((void)0)
#define RP_HTS(kt, which) ((void)0)
#define RP_PROBE_STAGE_FILTER(kt, half) ((void)0)
#define RP_PROBE_READS_DECL ((void)0)
#define RP_PROBE_READS_GATE ((void)0)
#define RP_PROBE_READS_PRIME(read_frags, smem) ((void)0)
#define RP_PTS_DECL ((void)0)
#define RP_PTS(v) ((void)0)
#define RP_PROBE_PERSIST_TILE_END(more) ((void)0)
#endif

// C/D layout of v_mfma_f32_32x32x16_bf16: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
__device__ __forceinline__ int mfma32_row(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

// Tile configuration: block tile BM x BN x BK, WM x WN waves, NSTAGE-deep LDS ring.
template <int BM_, int BN_, int BK_, int WM_, int WN_, int NSTAGE_, int PIPE_ = 0, int FP8_ = 0, int AAUX_ = 0, int KTAIL_ = 0,
          int OCC_ = 0>
struct GemmCfg {
  // OCC = 2: TWO workgroups of this configuration are meant to share a CU (4 waves each, one per SIMD and workgroup, at most
  // 256 registers per lane, at most 80 KiB of LDS): the hardware scheduler then runs one workgroup's main loop under the
  // other's prologue / hand-over stalls / epilogue.  0: whatever the register allocation allows.
  static constexpr int OCC = OCC_;
  // KTAIL = 1 (gemm_tile_pipe): K may end half a k-tile early (K % BK == BK / 2).  The last tile then carries only its
  // first half: the lanes whose 16-byte chunk lies in the missing half fetch the chunk BK/2 earlier instead (a duplicate,
  // never multiplied, never out of bounds) and the tile runs half its k-steps.
  static constexpr int KTAIL = KTAIL_;
  static constexpr int AAUX = AAUX_;  // cache-policy bits of the A operand's LDS-DMA (2 = nt: streamed once)
  // FP8 = 1: the operands are e4m3 bytes, addressed as if they were bf16 rows of half the length (BK,
  // K and the operands' ld all count 2-byte units); only the fragment reads and the MFMA differ.
  static constexpr int FP8 = FP8_;
  static constexpr int BM = BM_, BN = BN_, BK = BK_, WM = WM_, WN = WN_, NSTAGE = NSTAGE_;
  static constexpr int PIPE = PIPE_;  // 1: one wave per SIMD, fragment reads software-pipelined (gemm_tile_pipe)
  static constexpr int NWAVES_ = WM_ * WN_;
  static constexpr int NWAVES = WM * WN, THREADS = NWAVES * 64;
  static constexpr int FM = BM / WM / 32, FN = BN / WN / 32;  // 32x32 accumulator fragments per wave
  static constexpr int ROW_BYTES = BK * 2;
  static constexpr int SLOTS = ROW_BYTES / 16;           // 16-B chunks per row: 4 (BK=32) / 8 (BK=64)
  static constexpr int A_BYTES = BM * ROW_BYTES, W_BYTES = BN * ROW_BYTES;
  static constexpr int STAGE_BYTES = A_BYTES + W_BYTES;
  static constexpr int RING_BYTES = NSTAGE * STAGE_BYTES;
  static constexpr int LDS_BYTES = RING_BYTES > NWAVES_ * 9216 ? RING_BYTES : NWAVES_ * 9216;
  static constexpr int ROWS_PER_DMA = 1024 / ROW_BYTES;  // rows covered by one wave-instruction
  static constexpr int A_DMA = BM / ROWS_PER_DMA / NWAVES, W_DMA = BN / ROWS_PER_DMA / NWAVES;  // per wave
  static_assert(BM % (ROWS_PER_DMA * NWAVES) == 0 && BN % (ROWS_PER_DMA * NWAVES) == 0, "DMA split");
  static_assert(BK == 32 || BK == 64, "BK");
  // physical 16-B slot of logical chunk kc in row `row`: conflict-free for the 16-lane groups of
  // ds_read_b128 when 32 consecutive rows read the same logical chunk
  __device__ static __forceinline__ int swz(int row) { return (BK == 32) ? ((row >> 2) & 3) : ((row >> 1) & 7); }
  __device__ static __forceinline__ int off(int row, int kc) { return row * ROW_BYTES + ((kc ^ swz(row)) << 4); }
};

// A wave grid WM x WN with FM x FN accumulator fragments per wave that covers only the first WM * FM * 32 rows of the
// block tile's M side: for the last tile of an M extent that ends inside it (1152 = 4.5 tiles, 1472 = 5.75 tiles of 256)
// the rows beyond are never multiplied instead of multiplied and thrown away.
template <int WM_, int WN_, int FM_, int FN_>
struct WaveLayout {
  static constexpr int WM = WM_, WN = WN_, FM = FM_, FN = FN_;
};

// Per-wave LDS staging area the epilogues may use after the main loop (the ring is dead by then) to
// turn the accumulator layout (lane = column) into row-contiguous 16-B-per-lane global accesses:
// up to 64 rows of 128 B + 16 B pad.
constexpr int EPI_ROW_BYTES = 144;
constexpr int EPI_STAGE_BYTES = 64 * EPI_ROW_BYTES;  // 9216 B per wave

// Operand: [rows, ld] row-major, K-contiguous; tile rows beyond `rows` are clamped to rows-1.
// (A panel-blocked form of the scan's premise operand - 256 rows x 128 B pieces, one K-slice of a row block = one
// contiguous 32 KB run - was tried in round 2: identical timings to the bit, the DRAM access pattern is not what
// bounds the scan; removed.)
struct GemmOperand {
  const bf16_t* ptr;
  int ld;    // elements
  int rows;  // rows that may be read
  __device__ __forceinline__ size_t row_off(int r, int kc) const { return (size_t)min(r, rows - 1) * ld + kc * 8; }
  __device__ __forceinline__ size_t k_off(int kt, int BK) const { return (size_t)kt * BK; }
};

// An epilogue may declare `void prologue(char* extra_lds, int wave, int lane, int n0)` (n0 = first W row of the
// tile): gemm_tile_pipe calls it before the first operand DMA so that per-tile metadata can ride into LDS (beyond
// the ring) by LDS-DMA and be complete, by the in-order vmcnt accounting, long before the epilogue reads it.
template <class E, class = void>
struct has_prologue : std::false_type {};
template <class E>
struct has_prologue<E, std::void_t<decltype(&E::prologue)>> : std::true_type {};

// ... and `void reduce(int tid)`: called once the main loop is over (the metadata has landed and every wave is past the
// barrier), followed by a barrier - turns the metadata into what the epilogue reads.
// An epilogue may ask for `template <int FN> void preload(int n_base, int lane)` (static constexpr bool preload_hook = true):
// the persistent tile loop calls it in front of a tile's LAST k-tile - nothing of the wave's is in flight there - so that
// per-token operands of the epilogue (the RMSNorm factors) are in registers when the epilogue starts instead of being
// requested by its first instructions (round 6: a round trip to another XCD's write, ~1 us per tile, 30 tiles per workgroup);
// the epilogue then runs as run<FM, FN, true>.
template <class E, class = void>
struct has_preload : std::false_type {};
template <class E>
struct has_preload<E, std::enable_if_t<E::preload_hook>> : std::true_type {};
template <class E, class = void>
struct has_reduce : std::false_type {};
template <class E>
struct has_reduce<E, std::void_t<decltype(&E::reduce)>> : std::true_type {};

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// acc[mfrag][nfrag]; wave covers rows m_base + mfrag*32 + mfma32_row(r, hi), cols n_base + nfrag*32 + (lane&31)
template <class C, class Epilogue>
__device__ __forceinline__ void gemm_tile(const GemmOperand A, const GemmOperand W, int K, int tile_m,
                                          int tile_n, Epilogue& epi, char* smem) {
  constexpr int BK = C::BK, NSTAGE = C::NSTAGE, FM = C::FM, FN = C::FN;
  RP_TS_PLAIN(0);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wave_row = wave / C::WN, wave_col = wave % C::WN;
  const int hi = lane >> 5;

  f32x16 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // ---- LDS-DMA assignment: instruction d of this wave fills rows [(wave*DPW + d) * RPD, +RPD) of an
  // operand image; lane l lands at byte (l * 16) of that 1-KiB piece = (row l / SLOTS, slot l % SLOTS),
  // and therefore fetches logical chunk (slot ^ swz(row)) of that row.
  const bf16_t* a_src[C::A_DMA];
  const bf16_t* w_src[C::W_DMA];
#pragma unroll
  for (int d = 0; d < C::A_DMA; ++d) {
    const int row = (wave * C::A_DMA + d) * C::ROWS_PER_DMA + lane / C::SLOTS;
    const int kc = (lane % C::SLOTS) ^ C::swz(row);
    a_src[d] = A.ptr + A.row_off(tile_m * C::BM + row, kc);
  }
#pragma unroll
  for (int d = 0; d < C::W_DMA; ++d) {
    const int row = (wave * C::W_DMA + d) * C::ROWS_PER_DMA + lane / C::SLOTS;
    const int kc = (lane % C::SLOTS) ^ C::swz(row);
    w_src[d] = W.ptr + W.row_off(tile_n * C::BN + row, kc);
  }
  // KTAIL (as in gemm_tile_pipe): K % BK is 0 or BK / 2; in the half tile the lanes whose 16-byte chunk lies beyond K fetch the
  // chunk 64 bytes earlier (a duplicate, never out of bounds) and the upper k-steps' A fragments are zeroed by selects
  static_assert(C::KTAIL == 0 || (BK == 64 && C::SLOTS == 8 && C::ROWS_PER_DMA == 8 && C::FP8 != 0), "KTAIL: 128-byte e4m3 rows");
  const int nk = C::KTAIL != 0 ? (K + BK / 2) / BK : K / BK;
  const bool has_tail = C::KTAIL != 0 && nk * BK != K;
  const int fix_lane = (lane >> 2) & 1;
  auto stage = [&](int kt, int buf) {
    char* base = smem + buf * C::STAGE_BYTES;
    const int ke = kt;
    const int fix_mask = (C::KTAIL != 0 && has_tail && kt == nk - 1) ? -1 : 0;  // (scalar)
#pragma unroll
    for (int d = 0; d < C::A_DMA; ++d) {
      const bf16_t* src = a_src[d] + A.k_off(ke, BK);
      if constexpr (C::KTAIL != 0) src += ((((fix_lane ^ (wave * C::A_DMA + d)) & 1) ? -(BK / 2) : 0) & fix_mask);
      if constexpr (C::AAUX == 2)  // (nt policy on a streamed A operand)
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)(base + (wave * C::A_DMA + d) * 1024), 16, 0, 2);
      else
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)(base + (wave * C::A_DMA + d) * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int d = 0; d < C::W_DMA; ++d) {
      const bf16_t* src = w_src[d] + W.k_off(ke, BK);
      if constexpr (C::KTAIL != 0) src += ((((fix_lane ^ (wave * C::W_DMA + d)) & 1) ? -(BK / 2) : 0) & fix_mask);
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)(base + C::A_BYTES + (wave * C::W_DMA + d) * 1024), 16, 0, 0);
    }
  };
  constexpr int DMA_PER_STAGE = C::A_DMA + C::W_DMA;  // per wave

  // fragment read offsets (within an operand image)
  int a_off[FM][BK / 16], b_off[FN][BK / 16];  // [frag][k16 step]
#pragma unroll
  for (int ks = 0; ks < BK / 16; ++ks) {
#pragma unroll
    for (int f = 0; f < FM; ++f) a_off[f][ks] = C::off(wave_row * (FM * 32) + f * 32 + (lane & 31), ks * 2 + hi);
#pragma unroll
    for (int f = 0; f < FN; ++f) b_off[f][ks] = C::off(wave_col * (FN * 32) + f * 32 + (lane & 31), ks * 2 + hi);
  }

  if constexpr (has_prologue<Epilogue>::value) epi.prologue(smem + C::RING_BYTES, wave, lane, tile_n * C::BN);
  // prologue: NSTAGE-1 tiles in flight
#pragma unroll
  for (int s = 0; s < NSTAGE - 1; ++s)
    if (s < nk) stage(s, s);

  int buf = 0;
  for (int kt = 0; kt < nk; ++kt) {
    // tile kt has landed once at most (NSTAGE-2) younger tiles' DMAs of this wave remain in flight
    if (kt + NSTAGE - 2 < nk)
      wait_vmcnt<(NSTAGE - 2) * DMA_PER_STAGE>();
    else
      wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();  // every wave's share of tile kt is in LDS; tile kt-1 fully consumed
    if (kt == 0) RP_TS_PLAIN(1);
    if (kt + NSTAGE - 1 < nk) {
      int nb = buf + NSTAGE - 1;
      if (nb >= NSTAGE) nb -= NSTAGE;
      stage(kt + NSTAGE - 1, nb);
    }
    const char* sa = smem + buf * C::STAGE_BYTES;
    const char* sb = sa + C::A_BYTES;
    if constexpr (C::FP8 != 0) {
      // 64 e4m3 values per MFMA step = 4 x 16-B slots of a row: lanes 0-31 take slots 0,1, lanes 32-63
      // slots 2,3 (any split works as long as both operands use the same one).  The two halves are
      // read in LOGICAL slot order: the swizzle may swap their physical order differently per row.
#pragma unroll
      for (int ks = 0; ks < C::ROW_BYTES / 64; ++ks) {
        i32x8 af8[FM], bf8[FN];
#pragma unroll
        for (int f = 0; f < FM; ++f) {
          const int row = wave_row * (FM * 32) + f * 32 + (lane & 31);
          const i32x4 lo = *reinterpret_cast<const i32x4*>(sa + C::off(row, ks * 4 + hi * 2));
          const i32x4 up = *reinterpret_cast<const i32x4*>(sa + C::off(row, ks * 4 + hi * 2 + 1));
          af8[f] = i32x8{lo[0], lo[1], lo[2], lo[3], up[0], up[1], up[2], up[3]};
          if constexpr (C::KTAIL != 0) {  // the half tile multiplies its lower k-steps only: acc + (+0 x b) is acc bit for bit
            const bool z = has_tail && kt == nk - 1 && ks >= C::ROW_BYTES / 128;
#pragma unroll
            for (int e = 0; e < 8; ++e) af8[f][e] = z ? 0 : af8[f][e];
          }
        }
#pragma unroll
        for (int f = 0; f < FN; ++f) {
          const int row = wave_col * (FN * 32) + f * 32 + (lane & 31);
          const i32x4 lo = *reinterpret_cast<const i32x4*>(sb + C::off(row, ks * 4 + hi * 2));
          const i32x4 up = *reinterpret_cast<const i32x4*>(sb + C::off(row, ks * 4 + hi * 2 + 1));
          bf8[f] = i32x8{lo[0], lo[1], lo[2], lo[3], up[0], up[1], up[2], up[3]};
        }
        // block-scaled MFMA with every E8M0 scale = 127 (2^0): plain e4m3 x e4m3 at the MX rate
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
          for (int j = 0; j < FN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(af8[i], bf8[j], acc[i][j], 0, 0, 0, 0x7f7f7f7f,
                                                                        0, 0x7f7f7f7f);
      }
    } else
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      bf16x8 af[FM], bfr[FN];
#pragma unroll
      for (int f = 0; f < FM; ++f) af[f] = *reinterpret_cast<const bf16x8*>(sa + a_off[f][ks]);
#pragma unroll
      for (int f = 0; f < FN; ++f) bfr[f] = *reinterpret_cast<const bf16x8*>(sb + b_off[f][ks]);
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
    if (++buf == NSTAGE) buf = 0;
  }
  __syncthreads();  // all waves done with LDS before an epilogue reuses it
  RP_TS_PLAIN(2);

  epi.template run<FM, FN>(acc, tile_m * C::BM + wave_row * (FM * 32), tile_n * C::BN + wave_col * (FN * 32), lane,
                           smem + wave * EPI_STAGE_BYTES);
  RP_TS_PLAIN(3);
}

// One-wave-per-SIMD variant (WM*WN = 4 waves, 128x128 per wave at BM = BN = 256): each fragment
// read from LDS feeds 4 MFMAs instead of 2.67, so the LDS pipe (fragment reads + DMA writes) needs
// 1536 clk per 256x256x64 tile against 2048 clk of MFMA issue, where the 8-wave tiling needs
// 2048 / 2048.  With a single wave per SIMD nothing else hides latency, so the k-steps are
// software-pipelined by hand: the fragments of k-step s+1 are read while the MFMAs of k-step s
// issue, and the tile hand-over (counted wait, barrier, next DMA issue, first fragment read of the
// next tile) sits in front of the last k-step's MFMAs.
// L: how the waves split the block tile (WaveLayout below; default: the configuration's own WM x WN grid).  The LDS
// image, the DMA split and the k loop do not depend on it, and every output element is the same K-ascending chain of
// MFMA steps under every layout.
template <class C, class Epilogue, class L = C>
__device__ __forceinline__ void gemm_tile_pipe(const GemmOperand A, const GemmOperand W, int K, int tile_m,
                                               int tile_n, Epilogue& epi, char* smem) {
  static_assert(L::WM * L::WN == C::NWAVES && L::WM * L::FM * 32 <= C::BM && L::WN * L::FN * 32 <= C::BN, "wave layout");
  // bf16: one MFMA k-step = 16 values = two 16-B slots of a row (one per lane half); e4m3: one k-step =
  // 64 values = four slots (two per lane half), so a 128-B row holds 4 bf16 or 2 e4m3 k-steps.
  constexpr int BK = C::BK, FM = L::FM, FN = L::FN, KS = C::FP8 ? C::ROW_BYTES / 64 : BK / 16;
  constexpr int RPF = C::FP8 ? 2 : 1;  // ds_read_b128 per fragment
  using frag_t = std::conditional_t<C::FP8 != 0, i32x8, bf16x8>;
  RP_TS(0);
  static_assert(KS % 2 == 0 && KS >= 2, "even number of k-steps (fragment double-buffer parity)");
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wave_row = wave / L::WN, wave_col = wave % L::WN;
  const int hi = lane >> 5;

  f32x16 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const bf16_t* a_src[C::A_DMA];
  const bf16_t* w_src[C::W_DMA];
#pragma unroll
  for (int d = 0; d < C::A_DMA; ++d) {
    const int row = (wave * C::A_DMA + d) * C::ROWS_PER_DMA + lane / C::SLOTS;
    const int kc = (lane % C::SLOTS) ^ C::swz(row);
    a_src[d] = A.ptr + A.row_off(tile_m * C::BM + row, kc);
  }
#pragma unroll
  for (int d = 0; d < C::W_DMA; ++d) {
    const int row = (wave * C::W_DMA + d) * C::ROWS_PER_DMA + lane / C::SLOTS;
    const int kc = (lane % C::SLOTS) ^ C::swz(row);
    w_src[d] = W.ptr + W.row_off(tile_n * C::BN + row, kc);
  }
  const int nk = C::KTAIL != 0 ? (K + BK / 2) / BK : K / BK;  // (KTAIL: K % BK is 0 or BK / 2)
  const bool has_tail = C::KTAIL != 0 && nk * BK != K;
  // tail tile: element offset added to the DMA source of lanes whose chunk lies beyond K (0 for the others).  With 128-byte
  // rows (8 rows, 8 chunks per DMA piece) the swizzle's top bit is the parity of the piece index, so the chunk lies in the
  // upper half iff ((lane >> 2) ^ piece) & 1: two values per lane, one for even and one for odd pieces.
  static_assert(C::KTAIL == 0 || (BK == 64 && C::SLOTS == 8 && C::ROWS_PER_DMA == 8), "KTAIL: 128-byte rows");
  const int fix_lane = (lane >> 2) & 1;
  if constexpr (has_prologue<Epilogue>::value) epi.prologue(smem + C::RING_BYTES, wave, lane, tile_n * C::BN);
  // half 0: the A image of a stage, half 1: the W image (issued one k-step apart, see tile_body)
  auto stage_half = [&](int kt, int buf, int half) {
    char* base = smem + buf * C::STAGE_BYTES;
    RP_PROBE_STAGE_FILTER(kt, half);
    const int fix_mask = (C::KTAIL != 0 && has_tail && kt == nk - 1) ? -1 : 0;  // (scalar)
    if (half == 0) {
#pragma unroll
      for (int d = 0; d < C::A_DMA; ++d) {
        const bf16_t* src = a_src[d] + A.k_off(kt, BK);
        if constexpr (C::KTAIL != 0) src += ((((fix_lane ^ (wave * C::A_DMA + d)) & 1) ? -(BK / 2) : 0) & fix_mask);
        if constexpr (C::AAUX == 2)
          __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)(base + (wave * C::A_DMA + d) * 1024), 16, 0, 2);
        else
          __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)(base + (wave * C::A_DMA + d) * 1024), 16, 0, 0);
      }
    } else {
#pragma unroll
      for (int d = 0; d < C::W_DMA; ++d) {
        const bf16_t* src = w_src[d] + W.k_off(kt, BK);
        if constexpr (C::KTAIL != 0) src += ((((fix_lane ^ (wave * C::W_DMA + d)) & 1) ? -(BK / 2) : 0) & fix_mask);
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)(base + C::A_BYTES + (wave * C::W_DMA + d) * 1024), 16, 0, 0);
      }
    }
  };
  auto stage = [&](int kt, int buf) {
    stage_half(kt, buf, 0);
    stage_half(kt, buf, 1);
  };

  int a_off[FM][KS * RPF], b_off[FN][KS * RPF];
#pragma unroll
  for (int ks = 0; ks < KS * RPF; ++ks) {
    // bf16: slot 2 ks + hi.  e4m3: k-step s = ks / 2 takes slots 4 s + 2 hi + {0, 1} (both operands split alike)
    const int slot = C::FP8 ? ((ks >> 1) * 4 + hi * 2 + (ks & 1)) : (ks * 2 + hi);
#pragma unroll
    for (int f = 0; f < FM; ++f) a_off[f][ks] = C::off(wave_row * (FM * 32) + f * 32 + (lane & 31), slot);
#pragma unroll
    for (int f = 0; f < FN; ++f) b_off[f][ks] = C::A_BYTES + C::off(wave_col * (FN * 32) + f * 32 + (lane & 31), slot);
  }

  frag_t af[2][FM], bfr[2][FN];  // double-buffered fragments, parity = k-step & 1
  RP_PROBE_READS_DECL;
  auto read_frags = [&](const char* st, int ks, int p) {
    RP_PROBE_READS_GATE;
    if constexpr (C::FP8 != 0) {
#pragma unroll
      for (int f = 0; f < FM; ++f) {
        const i32x4 lo = *reinterpret_cast<const i32x4*>(st + a_off[f][2 * ks]);
        const i32x4 up = *reinterpret_cast<const i32x4*>(st + a_off[f][2 * ks + 1]);
        af[p][f] = i32x8{lo[0], lo[1], lo[2], lo[3], up[0], up[1], up[2], up[3]};
      }
#pragma unroll
      for (int f = 0; f < FN; ++f) {
        const i32x4 lo = *reinterpret_cast<const i32x4*>(st + b_off[f][2 * ks]);
        const i32x4 up = *reinterpret_cast<const i32x4*>(st + b_off[f][2 * ks + 1]);
        bfr[p][f] = i32x8{lo[0], lo[1], lo[2], lo[3], up[0], up[1], up[2], up[3]};
      }
    } else {
#pragma unroll
      for (int f = 0; f < FM; ++f) af[p][f] = *reinterpret_cast<const bf16x8*>(st + a_off[f][ks]);
#pragma unroll
      for (int f = 0; f < FN; ++f) bfr[p][f] = *reinterpret_cast<const bf16x8*>(st + b_off[f][ks]);
    }
  };
  auto zero_a_if = [&](int p, bool z) {
#pragma unroll
    for (int f = 0; f < FM; ++f)
#pragma unroll
      for (int e = 0; e < (int)(sizeof(frag_t) / 4); ++e) {
        int v = reinterpret_cast<int*>(&af[p][f])[e];
        reinterpret_cast<int*>(&af[p][f])[e] = z ? 0 : v;
      }
  };
  auto mma = [&](int p) {
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        if constexpr (C::FP8 != 0)  // block-scaled MFMA with every E8M0 scale = 127 (2^0): e4m3 x e4m3 at the MX rate
          acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(af[p][i], bfr[p][j], acc[i][j], 0, 0, 0, 0x7f7f7f7f,
                                                                      0, 0x7f7f7f7f);
        else
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[p][i], bfr[p][j], acc[i][j], 0, 0, 0);
      }
  };

  constexpr int NSTAGE = C::NSTAGE, DPS = C::A_DMA + C::W_DMA;
#pragma unroll
  for (int t = 0; t < NSTAGE; ++t)
    if (t < nk) stage(t, t);
  if (nk >= NSTAGE)
    wait_vmcnt<(NSTAGE - 1) * DPS>();
  else
    wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();
  RP_TS(1);
  read_frags(smem, 0, 0);
  RP_PROBE_READS_PRIME(read_frags, smem);

  // Tile kt lives in ring slot kt % NSTAGE.  MODE 2: hand over to tile kt+1 and refill this slot with
  // tile kt+NSTAGE; 1: hand over only (tail of the K loop); 0: last tile.  A refill is issued in two
  // halves so that no more than one LDS-DMA instruction sits between two MFMAs (a DMA issue costs the
  // wave ~20 clk, an MFMA slot is 32 clk): the A image right after the hand-over, between the last
  // k-step's MFMAs, and the W image (PEND) between the MFMAs of the next tile's first k-step.
  // issue-order hint for one k-step: after every MFMA at most ceil(ops / MFMAs) of the pending
  // non-MFMA operations, fragment reads first (the next k-step waits on them), then DMA issues
  auto hint_order = [](auto nread_tag, auto ndma_tag) {
    constexpr int NREAD = decltype(nread_tag)::value, NDMA = decltype(ndma_tag)::value, NM = FM * FN;
    constexpr int PER = (NREAD + NDMA + NM - 1) / NM;
    int rd = NREAD, dm = NDMA;
#pragma unroll
    for (int n = 0; n < NM; ++n) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // MFMA
#pragma unroll
      for (int q = 0; q < PER; ++q) {
        if (rd > 0) {
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // DS read
          --rd;
        } else if (dm > 0) {
          __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);  // VMEM read (LDS-DMA)
          --dm;
        }
      }
    }
  };
  using NoDma = std::integral_constant<int, 0>;
  using Reads = std::integral_constant<int, (FM + FN) * RPF>;
  int buf = 0;
  auto tile_body = [&](int kt, auto mode_tag, auto pend_tag) {
    constexpr int MODE = decltype(mode_tag)::value;
    constexpr bool PEND = decltype(pend_tag)::value != 0;
    const char* st = smem + buf * C::STAGE_BYTES;
    // KTAIL: the last tile of a K that ends half a tile early multiplies its lower k-steps only: the A fragments of the
    // upper ones are replaced by zeros (selects, no branch: a branch around MFMAs made the register allocator copy
    // accumulators), and acc + (+0 x b) is acc bit for bit - an accumulator that starts at +0 never holds -0, and the
    // duplicate chunks the B side reads are finite values of the operand itself
    const bool half_tile = C::KTAIL != 0 && MODE == 0 && has_tail;
#pragma unroll
    for (int ks = 0; ks < KS - 1; ++ks) {
      read_frags(st, ks + 1, (ks + 1) & 1);
      if (PEND && ks == 0) stage_half(kt - 1 + NSTAGE, buf == 0 ? NSTAGE - 1 : buf - 1, 1);
      if (C::KTAIL != 0 && MODE == 0 && ks >= KS / 2) zero_a_if(ks & 1, half_tile);
      mma(ks & 1);
      if (PEND && ks == 0)
        hint_order(Reads(), std::integral_constant<int, C::W_DMA>());
      else
        hint_order(Reads(), NoDma());
    }
    if (MODE >= 1) {
      // every fragment of tile kt is in registers once this wave's LDS reads have returned
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      RP_HTS(kt, 0);
      if (MODE == 2)
        wait_vmcnt<(NSTAGE - 2) * DPS>();  // this wave's share of tile kt+1 has landed
      else
        wait_vmcnt<0>();
      RP_HTS(kt, 1);
      __builtin_amdgcn_s_barrier();  // ... and everyone's; the slot of tile kt is free
      RP_HTS(kt, 2);
        const int freed = buf;
      if (++buf == NSTAGE) buf = 0;
      read_frags(smem + buf * C::STAGE_BYTES, 0, 0);
      if (MODE == 2) stage_half(kt + NSTAGE, freed, 0);
    }
    if (C::KTAIL != 0 && MODE == 0) zero_a_if((KS - 1) & 1, half_tile);
    mma((KS - 1) & 1);
    if (MODE == 2)
      hint_order(Reads(), std::integral_constant<int, C::A_DMA>());
    else if (MODE == 1)
      hint_order(Reads(), NoDma());
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  int kt = 0;
  if (nk > NSTAGE) {
    tile_body(0, I2(), I0());
    for (kt = 1; kt + NSTAGE < nk; ++kt) tile_body(kt, I2(), I1());
    tile_body(kt, I1(), I1());  // kt == nk - NSTAGE: completes the last refill
    ++kt;
  }
  for (; kt + 1 < nk; ++kt) tile_body(kt, I1(), I0());
  tile_body(kt, I0(), I0());  // (a half tile skips its upper k-steps inside the body: no second set of MFMA code)
  __syncthreads();
  RP_TS(2);
  if constexpr (has_reduce<Epilogue>::value) {  // per-tile metadata -> what the epilogue reads (one pass, then visible to all)
    epi.reduce(tid);
    __syncthreads();
  }

  epi.template run<FM, FN>(acc, tile_m * C::BM + wave_row * (FM * 32), tile_n * C::BN + wave_col * (FN * 32), lane,
                           smem + wave * EPI_STAGE_BYTES);
  RP_TS(3);
}

// PERSISTENT form of gemm_tile_pipe (bf16 operands, 2-stage ring, no K tail): ONE workgroup per CU walks a list of tiles
// (`next_tile(tm, tn)` hands out the workgroup's next tile, false when it has none left).  Two things a launch with one
// workgroup per tile pays per tile are gone (tools/probes/gemm_phase.py, round 5: of 45.6 us per FFN-in tile 1.8 us are the
// prologue - the wait for the first k-tile with nothing else to do - and ~3 us the turn-over of the CU from one workgroup
// to the next):
//   * no workgroup turn-over between tiles;
//   * the first k-tile of tile t+1 is requested BEFORE the epilogue of tile t and lands under it.  LDS (all 160 KiB):
//       [0, 64 K)      ring slot 0 - free once the main loop is over: tile t+1's first k-tile goes there
//       [64 K, 136 K)  the epilogue's per-wave staging areas (8 x 9 KiB; over ring slot 1, which is dead by then)
//       [136 K, 160 K) per-tile metadata of an epilogue with a prologue() hook (<= 24 slot rows of 1 KiB)
//     After the epilogue one raw barrier frees the staging area; the metadata and the second k-tile of tile t+1 are
//     requested and the main loop starts on a k-tile that has already landed.
// The queue of a wave at that point, oldest first: [slot 0 of t+1] [epilogue stores of t] [metadata] [slot 1 of t+1];
// the first counted wait leaves only the youngest DPS operations outstanding, as in the one-tile form.
// Every output element is the same K-ascending chain of MFMA steps: not a bit differs from gemm_tile_pipe.
//
// ORDERING ASSUMPTION.  `wait_vmcnt<DPS>` is read as "everything older than the youngest DPS vector-memory
// operations of this wave has completed" across THREE kinds of operation in one queue: LDS-DMA loads (slot 0 of tile t+1),
// the epilogue's global STORES, and the metadata DMA.  That holds on the gfx9 family this file is written for - gfx942 /
// gfx950 keep loads and stores on ONE vmcnt counter that retires in issue order (the ISA has no separate vscnt; gfx10+
// split stores off into vscnt, where a store would no longer hold younger loads' count back and this wait would have to
// become a load-only count).  The guard below refuses to compile the persistent form for anything else.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__)
#error "gemm_tiles_persist relies on gfx942/gfx950 vmcnt semantics (loads and stores retire in order on one counter)"
#endif
constexpr int PERSIST_EPI_OFF = 64 * 1024, PERSIST_META_OFF = 136 * 1024, PERSIST_LDS_BYTES = 160 * 1024;
template <class C, bool EDGE = false, class L0 = C, class Epilogue, class NextTile>
__device__ __forceinline__ void gemm_tiles_persist(const GemmOperand A, const GemmOperand W, int K, NextTile next_tile,
                                                   Epilogue& epi, char* smem, bool edge_on = false) {
  static_assert(C::PIPE != 0 && C::FP8 == 0 && C::KTAIL == 0 && C::NSTAGE == 2 && C::STAGE_BYTES == 64 * 1024 &&
                    C::NWAVES * EPI_STAGE_BYTES <= PERSIST_META_OFF - PERSIST_EPI_OFF,
                "persistent form: the pipelined 256 x 256 x 64 bf16 tile");
  constexpr int BK = C::BK, KS = BK / 16, NSTAGE = 2, DPS = C::A_DMA + C::W_DMA;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int tile_m, tile_n;
  if (!next_tile(tile_m, tile_n)) return;

  const bf16_t* a_src[C::A_DMA];
  const bf16_t* w_src[C::W_DMA];
  auto set_src = [&](int tm, int tn) {
#pragma unroll
    for (int d = 0; d < C::A_DMA; ++d) {
      const int row = (wave * C::A_DMA + d) * C::ROWS_PER_DMA + lane / C::SLOTS;
      const int kc = (lane % C::SLOTS) ^ C::swz(row);
      a_src[d] = A.ptr + A.row_off(tm * C::BM + row, kc);
    }
#pragma unroll
    for (int d = 0; d < C::W_DMA; ++d) {
      const int row = (wave * C::W_DMA + d) * C::ROWS_PER_DMA + lane / C::SLOTS;
      const int kc = (lane % C::SLOTS) ^ C::swz(row);
      w_src[d] = W.ptr + W.row_off(tn * C::BN + row, kc);
    }
  };
  const int nk = K / BK;
  auto stage_half = [&](int kt, int buf, int half) {
    char* base = smem + buf * C::STAGE_BYTES;
    if (half == 0) {
#pragma unroll
      for (int d = 0; d < C::A_DMA; ++d)
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)(a_src[d] + A.k_off(kt, BK)), (lds_ptr_t)(base + (wave * C::A_DMA + d) * 1024),
                                         16, 0, 0);
    } else {
#pragma unroll
      for (int d = 0; d < C::W_DMA; ++d)
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)(w_src[d] + W.k_off(kt, BK)),
                                         (lds_ptr_t)(base + C::A_BYTES + (wave * C::W_DMA + d) * 1024), 16, 0, 0);
    }
  };
  auto stage = [&](int kt, int buf) {
    stage_half(kt, buf, 0);
    stage_half(kt, buf, 1);
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  RP_PTS_DECL;  // (probe builds: per-workgroup phase sums over its tiles)

  // One tile under wave layout L (WaveLayout; the configuration's own grid by default) up to and including its epilogue;
  // returns whether the workgroup has another tile (whose first k-tile is then on its way into ring slot 0).
  auto do_tile = [&](auto layout_tag, const auto& a_off, const auto& b_off) -> bool {
    using L = decltype(layout_tag);
    static_assert(L::WM * L::WN == C::NWAVES && L::WM * L::FM * 32 <= C::BM && L::WN * L::FN * 32 <= C::BN, "wave layout");
    constexpr int FM = L::FM, FN = L::FN;
    const int wave_row = wave / L::WN, wave_col = wave % L::WN;
    f32x16 acc[FM][FN];
    bf16x8 af[2][FM], bfr[2][FN];
    auto read_frags = [&](const char* st, int ks, int p) {
#pragma unroll
      for (int f = 0; f < FM; ++f) af[p][f] = *reinterpret_cast<const bf16x8*>(st + a_off[f][ks]);
#pragma unroll
      for (int f = 0; f < FN; ++f) bfr[p][f] = *reinterpret_cast<const bf16x8*>(st + b_off[f][ks]);
    };
    auto mma = [&](int p) {
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[p][i], bfr[p][j], acc[i][j], 0, 0, 0);
    };
    auto hint_order = [](auto nread_tag, auto ndma_tag) {
      constexpr int NREAD = decltype(nread_tag)::value, NDMA = decltype(ndma_tag)::value, NM = FM * FN;
      constexpr int PER = (NREAD + NDMA + NM - 1) / NM;
      int rd = NREAD, dm = NDMA;
#pragma unroll
      for (int n = 0; n < NM; ++n) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // MFMA
#pragma unroll
        for (int q = 0; q < PER; ++q) {
          if (rd > 0) {
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // DS read
            --rd;
          } else if (dm > 0) {
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);  // VMEM read (LDS-DMA)
            --dm;
          }
        }
      }
    };
    using NoDma = std::integral_constant<int, 0>;
    using Reads = std::integral_constant<int, FM + FN>;
    int buf = 0;
    auto tile_body = [&](int kt, auto mode_tag, auto pend_tag) {  // as in gemm_tile_pipe
      constexpr int MODE = decltype(mode_tag)::value;
      constexpr bool PEND = decltype(pend_tag)::value != 0;
      const char* st = smem + buf * C::STAGE_BYTES;
#pragma unroll
      for (int ks = 0; ks < KS - 1; ++ks) {
        read_frags(st, ks + 1, (ks + 1) & 1);
        if (PEND && ks == 0) stage_half(kt - 1 + NSTAGE, buf == 0 ? NSTAGE - 1 : buf - 1, 1);
        mma(ks & 1);
        if (PEND && ks == 0)
          hint_order(Reads(), std::integral_constant<int, C::W_DMA>());
        else
          hint_order(Reads(), NoDma());
      }
      if (MODE >= 1) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        wait_vmcnt<0>();  // (NSTAGE - 2) * DPS = 0: this wave's share of tile kt+1 has landed
        __builtin_amdgcn_s_barrier();
        const int freed = buf;
        if (++buf == NSTAGE) buf = 0;
        read_frags(smem + buf * C::STAGE_BYTES, 0, 0);
        if (MODE == 2) stage_half(kt + NSTAGE, freed, 0);
      }
      mma((KS - 1) & 1);
      if (MODE == 2)
        hint_order(Reads(), std::integral_constant<int, C::A_DMA>());
      else if (MODE == 1)
        hint_order(Reads(), NoDma());
    };

    RP_PTS(p_t0);
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    if constexpr (has_prologue<Epilogue>::value) epi.prologue(smem + PERSIST_META_OFF, wave, lane, tile_n * C::BN);
    stage(1, 1);
    wait_vmcnt<DPS>();  // everything older than slot 1's requests: slot 0 (and the last epilogue's stores, the metadata)
    __builtin_amdgcn_s_barrier();
    RP_PTS(p_t1);
    read_frags(smem, 0, 0);
    int kt = 0;
    if (nk > NSTAGE) {
      tile_body(0, I2(), I0());
      for (kt = 1; kt + NSTAGE < nk; ++kt) tile_body(kt, I2(), I1());
      tile_body(kt, I1(), I1());
      ++kt;
    }
    for (; kt + 1 < nk; ++kt) tile_body(kt, I1(), I0());
    if constexpr (has_preload<Epilogue>::value) epi.template preload<FN>(tile_n * C::BN + wave_col * (FN * 32), lane);
    tile_body(kt, I0(), I0());
    if constexpr (has_preload<Epilogue>::value) wait_vmcnt<0>();  // (landed under the k-tile; the next tile's requests come below)
    __syncthreads();  // every wave is done with the ring (nothing of this wave's is in flight here)
    if constexpr (has_reduce<Epilogue>::value) epi.reduce(tid);  // (the barrier behind it: below, after the next tile's requests)
    RP_PTS(p_t2);
    int nm, nn;
    const bool more = next_tile(nm, nn);
    const int em = tile_m * C::BM + wave_row * (FM * 32), en = tile_n * C::BN + wave_col * (FN * 32);
    if (more) {
      set_src(nm, nn);
      stage(0, 0);  // lands under the epilogue
      tile_m = nm;
      tile_n = nn;
    }
    if constexpr (has_reduce<Epilogue>::value) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();  // the reduced metadata is visible to every wave (raw: the requests above stay in flight)
    }
    if constexpr (has_preload<Epilogue>::value)
      epi.template run<FM, FN, true>(acc, em, en, lane, smem + PERSIST_EPI_OFF + wave * EPI_STAGE_BYTES);
    else
      epi.template run<FM, FN>(acc, em, en, lane, smem + PERSIST_EPI_OFF + wave * EPI_STAGE_BYTES);
    return more;
  };

  // LDS offsets of a wave's fragments under layout L
  auto fill_offs = [&](auto layout_tag, auto& a_off, auto& b_off, int ln) {
    using L = decltype(layout_tag);
    const int wave_row = wave / L::WN, wave_col = wave % L::WN;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int slot = ks * 2 + (ln >> 5);
#pragma unroll
      for (int f = 0; f < L::FM; ++f) a_off[f][ks] = C::off(wave_row * (L::FM * 32) + f * 32 + (ln & 31), slot);
#pragma unroll
      for (int f = 0; f < L::FN; ++f) b_off[f][ks] = C::A_BYTES + C::off(wave_col * (L::FN * 32) + f * 32 + (ln & 31), slot);
    }
  };
  // an edge tile computes its table when it comes up (a few dozen VALU operations; the empty asm keeps the compiler from
  // hoisting all three tables out of the tile loop, where they would be live across it and spill)
  auto edge_tile = [&](auto layout_tag) -> bool {
    using L = decltype(layout_tag);
    int ln = lane;
    asm volatile("" : "+v"(ln));
    int a_off[L::FM][KS], b_off[L::FN][KS];
    fill_offs(layout_tag, a_off, b_off, ln);
    return do_tile(layout_tag, a_off, b_off);
  };
  int a_off0[L0::FM][KS], b_off0[L0::FN][KS];  // L0: the wave layout of a full tile (the configuration's own grid by default)
  fill_offs(L0(), a_off0, b_off0, lane);

  set_src(tile_m, tile_n);
  stage(0, 0);  // the workgroup's first tile: both slots requested here; later tiles find slot 0 requested already
  for (;;) {
    bool more;
    if constexpr (EDGE) {
      // the last feature tile of an extent that ends inside it: a wave grid over the valid features only (WaveLayout)
      const int vf = __builtin_amdgcn_readfirstlane(A.rows - tile_m * C::BM);
      if (edge_on && vf <= 128)
        more = edge_tile(WaveLayout<2, 4, 2, 2>());
      else if (edge_on && vf <= 192)
        more = edge_tile(WaveLayout<1, 8, 6, 1>());
      else
        more = do_tile(L0(), a_off0, b_off0);
    } else {
      more = do_tile(L0(), a_off0, b_off0);
    }
    RP_PROBE_PERSIST_TILE_END(more);
    if (!more) break;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // the staging areas (over ring slot 1) and the metadata rows are free
  }
}

// XCD-aware tile order.  Workgroup b runs on XCD b % 8 (observed, speed only), so consecutive
// "logical" ids are handed out per XCD: logical = (b % 8) * ceil-chunk + b / 8 (bijective form),
// then logical ids walk the tile grid in column-groups of GROUP_M row-tiles so that the
// workgroups resident on one XCD share a few A row-panels and W column-panels in its L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

__device__ __forceinline__ void tile_coords(int logical, int tiles_m, int tiles_n, int group_m,
                                            int& tm, int& tn) {
  const int per_group = group_m * tiles_n;
  const int g = logical / per_group;
  const int first_m = g * group_m;
  const int gm = min(group_m, tiles_m - first_m);
  const int in_g = logical - g * per_group;
  tm = first_m + in_g % gm;
  tn = in_g / gm;
}

}  // namespace rp
