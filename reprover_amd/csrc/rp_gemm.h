// bf16 MFMA GEMM core for gfx950:  C[M,N] = A[M,K] · W[N,K]^T   (both operands K-contiguous,
// i.e. activations row-major and nn.Linear weights as stored), fp32 accumulate.
//
// Workgroup = WM x WN waves over a BM x BN x BK block tile; each wave owns an (FM*32) x (FN*32)
// sub-tile as FM x FN v_mfma_f32_32x32x16_bf16 accumulators (16 fp32 registers each).
//
// Staging: A and W tiles go HBM/L2 -> LDS by LDS-DMA (global_load_lds_dwordx4: no VGPR round
// trip, 1 KiB per wave-instruction) into an NSTAGE-deep ring; tile t+NSTAGE-1 is issued while
// tile t is consumed, with a COUNTED s_waitcnt vmcnt(...) and a raw s_barrier per K-step so the
// DMAs stay in flight across barriers (cdna_hip_programming.md §5 "Pipelining across barriers").
// The LDS image is lane-linear per DMA instruction, so the bank-conflict swizzle is applied to the
// per-lane SOURCE address and again on the ds_read_b128 fragment reads (rule 21): 16-B slot index
// ^= f(row), with f chosen per BK so that the 16-lane groups ds_read_b128 is serviced in hit 16
// distinct slots of the 256-B bank row (conflict-free).
//
// The epilogue is a functor so the same core serves the encoder GEMMs (bf16 store, fp32
// residual add, gated-GELU) and the similarity scan (accessibility mask + top-k filter).
#pragma once
#include "rp_util.h"

namespace rp {

// C/D layout of v_mfma_f32_32x32x16_bf16: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
__device__ __forceinline__ int mfma32_row(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

// Tile configuration: block tile BM x BN x BK, WM x WN waves, NSTAGE-deep LDS ring.
template <int BM_, int BN_, int BK_, int WM_, int WN_, int NSTAGE_>
struct GemmCfg {
  static constexpr int BM = BM_, BN = BN_, BK = BK_, WM = WM_, WN = WN_, NSTAGE = NSTAGE_;
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

// Per-wave LDS staging area the epilogues may use after the main loop (the ring is dead by then) to
// turn the accumulator layout (lane = column) into row-contiguous 16-B-per-lane global accesses:
// up to 64 rows of 128 B + 16 B pad.
constexpr int EPI_ROW_BYTES = 144;
constexpr int EPI_STAGE_BYTES = 64 * EPI_ROW_BYTES;  // 9216 B per wave

struct GemmOperand {
  const bf16_t* ptr;  // [rows, ld] row-major, K-contiguous
  int ld;             // elements
  int rows;           // rows that may be read; tile rows beyond are clamped to rows-1
};

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
    a_src[d] = A.ptr + (size_t)min(tile_m * C::BM + row, A.rows - 1) * A.ld + kc * 8;
  }
#pragma unroll
  for (int d = 0; d < C::W_DMA; ++d) {
    const int row = (wave * C::W_DMA + d) * C::ROWS_PER_DMA + lane / C::SLOTS;
    const int kc = (lane % C::SLOTS) ^ C::swz(row);
    w_src[d] = W.ptr + (size_t)min(tile_n * C::BN + row, W.rows - 1) * W.ld + kc * 8;
  }
  const int nk = K / BK;
  auto stage = [&](int kt, int buf) {
    char* base = smem + buf * C::STAGE_BYTES;
    const int ke = kt;
#pragma unroll
    for (int d = 0; d < C::A_DMA; ++d)
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(a_src[d] + (size_t)ke * BK),
                                       (lds_ptr_t)(base + (wave * C::A_DMA + d) * 1024), 16, 0, 0);
#pragma unroll
    for (int d = 0; d < C::W_DMA; ++d)
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(w_src[d] + (size_t)ke * BK),
                                       (lds_ptr_t)(base + C::A_BYTES + (wave * C::W_DMA + d) * 1024), 16, 0, 0);
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
    if (kt + NSTAGE - 1 < nk) {
      int nb = buf + NSTAGE - 1;
      if (nb >= NSTAGE) nb -= NSTAGE;
      stage(kt + NSTAGE - 1, nb);
    }
    const char* sa = smem + buf * C::STAGE_BYTES;
    const char* sb = sa + C::A_BYTES;
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

  epi.template run<FM, FN>(acc, tile_m * C::BM + wave_row * (FM * 32), tile_n * C::BN + wave_col * (FN * 32), lane,
                           smem + wave * EPI_STAGE_BYTES);
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
