// bf16 MFMA GEMM core for gfx950:  C[M,N] = A[M,K] · W[N,K]^T   (both operands K-contiguous,
// i.e. activations row-major and nn.Linear weights as stored), fp32 accumulate.
//
// Workgroup = 256 threads = 4 waves (2 x 2), tile 128 x 128 x 32; each wave owns a 64 x 64
// sub-tile as 2 x 2 v_mfma_f32_32x32x16_bf16 accumulators (64 fp32 VGPRs).  A and W tiles are
// staged through LDS (2 stages x (8 KB + 8 KB)); fragments are read with ds_read_b128 from an
// XOR-swizzled image (16-B chunk index ^= (row >> 2) & 3) that is conflict-free for the
// 16-lane groups ds_read_b128 is serviced in.  Staging variant 0 goes global -> VGPR -> LDS
// (prefetch of tile t+1 issued before the MFMAs of tile t); variant 1 uses the gfx950 LDS-DMA
// (global_load_lds_dwordx4) with the swizzle applied to the per-lane SOURCE address, LDS image
// linear per wave.
//
// The epilogue is a functor so the same core serves the encoder GEMMs (bf16 store, fp32
// residual add, gated-GELU) and the similarity scan (accessibility mask + top-k filter).
#pragma once
#include "rp_util.h"

namespace rp {

constexpr int GEMM_BM = 128, GEMM_BN = 128, GEMM_BK = 32;
constexpr int GEMM_STAGE_BYTES = (GEMM_BM + GEMM_BN) * GEMM_BK * 2;  // 16 KB
constexpr int GEMM_LDS_BYTES = 2 * GEMM_STAGE_BYTES;                  // 32 KB

// C/D layout of v_mfma_f32_32x32x16_bf16: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
__device__ __forceinline__ int mfma32_row(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

// byte offset of 16-B chunk `kc` (0..3) of row `row` inside a [rows][32] bf16 tile image
__device__ __forceinline__ int tile_off(int row, int kc) {
  return row * 64 + ((kc ^ ((row >> 2) & 3)) << 4);
}

struct GemmOperand {
  const bf16_t* ptr;  // [rows, ld] row-major, K-contiguous
  int ld;             // elements
  int rows;           // rows that may be read; tile rows beyond are clamped to rows-1
};

// acc[mfrag][nfrag]; wave covers rows m_base + mfrag*32 + mfma32_row(r, hi), cols n_base + nfrag*32 + (lane&31)
template <class Epilogue>
__device__ __forceinline__ void gemm_tile(const GemmOperand A, const GemmOperand W, int K,
                                          int tile_m, int tile_n, Epilogue& epi, char* smem) {
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wave_row = wave >> 1, wave_col = wave & 1;
  const int hi = lane >> 5;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // staging assignment: chunk c = tid + 256*i -> row c>>2, k-chunk c&3
  const bf16_t* a_src[2];
  const bf16_t* w_src[2];
  int st_off[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int c = tid + 256 * i;
    int row = c >> 2, kc = c & 3;
    int ar = min(tile_m * GEMM_BM + row, A.rows - 1);
    int wr = min(tile_n * GEMM_BN + row, W.rows - 1);
    a_src[i] = A.ptr + (size_t)ar * A.ld + kc * 8;
    w_src[i] = W.ptr + (size_t)wr * W.ld + kc * 8;
    st_off[i] = tile_off(row, kc);
  }
  // fragment read offsets (within a tile image)
  int a_off[2][2], b_off[2][2];  // [frag][ksub]
#pragma unroll
  for (int f = 0; f < 2; ++f)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      a_off[f][ks] = tile_off(wave_row * 64 + f * 32 + (lane & 31), ks * 2 + hi);
      b_off[f][ks] = tile_off(wave_col * 64 + f * 32 + (lane & 31), ks * 2 + hi);
    }

  const int nk = K / GEMM_BK;
  uint4 ra[2], rb[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    ra[i] = *reinterpret_cast<const uint4*>(a_src[i]);
    rb[i] = *reinterpret_cast<const uint4*>(w_src[i]);
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    *reinterpret_cast<uint4*>(smem + st_off[i]) = ra[i];
    *reinterpret_cast<uint4*>(smem + GEMM_BM * 64 + st_off[i]) = rb[i];
  }
  __syncthreads();

  int cur = 0;
  for (int kt = 0; kt < nk; ++kt) {
    const bool more = (kt + 1 < nk);
    if (more) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        ra[i] = *reinterpret_cast<const uint4*>(a_src[i] + (size_t)(kt + 1) * GEMM_BK);
        rb[i] = *reinterpret_cast<const uint4*>(w_src[i] + (size_t)(kt + 1) * GEMM_BK);
      }
    }
    const char* sa = smem + cur * GEMM_STAGE_BYTES;
    const char* sb = sa + GEMM_BM * 64;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 af[2], bfr[2];
#pragma unroll
      for (int f = 0; f < 2; ++f) {
        af[f] = *reinterpret_cast<const bf16x8*>(sa + a_off[f][ks]);
        bfr[f] = *reinterpret_cast<const bf16x8*>(sb + b_off[f][ks]);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
    if (more) {
      char* da = smem + (cur ^ 1) * GEMM_STAGE_BYTES;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        *reinterpret_cast<uint4*>(da + st_off[i]) = ra[i];
        *reinterpret_cast<uint4*>(da + GEMM_BM * 64 + st_off[i]) = rb[i];
      }
    }
    __syncthreads();
    cur ^= 1;
  }

  epi(acc, tile_m * GEMM_BM + wave_row * 64, tile_n * GEMM_BN + wave_col * 64, lane);
}

// XCD-aware tile order.  Workgroup b runs on XCD b % 8 (observed, speed only), so consecutive
// "logical" ids are handed out per XCD: logical = (b % 8) * ceil-chunk + b / 8 (bijective form),
// then logical ids walk the tile grid in column-groups of GROUP_M row-tiles so that the
// 32 workgroups resident on one XCD share a few A row-panels and W column-panels in its L2.
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
