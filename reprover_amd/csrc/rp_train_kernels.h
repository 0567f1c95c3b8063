// Device code of the retriever's training step on gfx950: the backward of the T5 encoder (rp_train.hip drives it).
//
// Reference: retrieval/model.py:155-181 (`training_step`, `configure_optimizers`) differentiates `forward`
// (:116-140) through `_encode` (:92-114) and HuggingFace's T5Stack with autograd; here every backward op is a kernel.
// Math (SURVEY.md App. A for the forward), per sub-layer with y = rs * (x W'^T), W' = W diag(ln), rs = rsqrt(mean x^2 + eps):
//   dzs  = rs * dy                                  (bf16 [T, O]: the one operand of both backward GEMMs)
//   dW'  = dzs^T x            (wgrad: K = tokens)   dW = dW' diag(ln),  d ln = colsum(dW' .* W)
//   v    = dzs W'             (dgrad: K = O)        dx += v - x * rs^2 * (sum_o dy_o y_o) / D
// GEMM shapes:
//   dgrad  - the forward's MFMA core (rp_gemm.h) on transposed bf16 weight copies made once per optimizer step;
//   wgrad  - both operands are token-major ([T, O] and [T, C]: the reduction index is the SLOW one), so the tiles are
//            staged as they lie (LDS-DMA of 128-byte row pieces) and the MFMA fragments are produced by the gfx950
//            transposing LDS read ds_read_b64_tr_b16: no transposed copy of any activation is ever written.
#pragma once
#include "rp_encoder_kernels.h"

namespace rp {

// ------------------------------------------------------------------------------------------
// wgrad:  dW[o, c] = sum_t Y[t, o] * X[t, c]      Y = dzs [T, O], X = saved activations [T, C], fp32 out
// ------------------------------------------------------------------------------------------
// LDS image of one operand tile (64 tokens x B features): [B/64 chunks][64 token rows][128 B = 64 features].
// One LDS-DMA wave-instruction fills 8 token rows of a chunk (8 lanes x 16 B = one 128-byte row piece: whole cache
// lines).  A transposing read covers 4 token rows x 32 B; with a 128-byte row pitch rows r and r+2 share their banks,
// so the 32-byte pieces of a row are XOR-swizzled with bit 1 of the row (applied to the DMA source address).
template <int BM_, int BN_, int WM_, int WN_, int NSTAGE_>
struct WgradCfg {
  static constexpr int BM = BM_, BN = BN_, BK = 64, WM = WM_, WN = WN_, NSTAGE = NSTAGE_;
  static constexpr int NWAVES = WM * WN, THREADS = NWAVES * 64;
  static constexpr int FM = BM / WM / 32, FN = BN / WN / 32;
  static constexpr int A_BYTES = BM * 128, W_BYTES = BN * 128, STAGE_BYTES = A_BYTES + W_BYTES;
  static constexpr int LDS_BYTES = NSTAGE * STAGE_BYTES;
  static constexpr int A_DMA = BM / 8 / NWAVES, W_DMA = BN / 8 / NWAVES;  // 1-KiB pieces per wave and stage
  static_assert(BM % (8 * NWAVES) == 0 && BN % (8 * NWAVES) == 0, "DMA split");
  static_assert(BM % (WM * 32) == 0 && BN % (WN * 32) == 0, "wave tiling");
};

// A product that runs WITHOUT split-K can finish its gradient in the epilogue (round 6: the feed-forward pair's dWi'): what
// unfold_kernel's GEGLU mode does on the partial matrices - de-interleave the packed gate | up rows, x ln, the 32-row-block
// partials of d ln = colsum(dW' .* W) - on the accumulators as they stand, so the 42-MB matrix is neither written as a
// partial nor read back.
struct WgradFinish {
  int geglu = 0;                    // 1: rows are the packed [32 gate | 32 up] order of wi_0 / wi_1
  float* g1 = nullptr;              // wi_1's gradient (wi_0's is the product's `out`)
  const float* w0 = nullptr;        // master weights [d_ff, C]
  const float* w1 = nullptr;
  const float* ln = nullptr;        // [C]
  float* dln_part = nullptr;        // [rows / 32][C]
};
struct EpiStoreF32 {  // out[o, c] = acc (rows o < n_o, columns c < n_c); 32 lanes cover 128 contiguous bytes
  float* out;
  int ldc, n_o, n_c;
  WgradFinish fin;
  template <int FM, int FN>
  __device__ __forceinline__ void run(f32x16 (&acc)[FM][FN], int m_base, int n_base, int lane) {
    const int hi = lane >> 5, cl = lane & 31;
    if (fin.geglu) {
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        const int ob = m_base + i * 32;  // a 32-row fragment = 32 gate rows or 32 up rows of one 64-row group
        if (ob >= n_o) continue;
        const bool up = (ob >> 5) & 1;
        const int sr0 = (ob >> 6) * 32;
        float* g = up ? fin.g1 : out;
        const float* w = up ? fin.w1 : fin.w0;
#pragma unroll
        for (int j = 0; j < FN; ++j) {
          const int c = n_base + j * 32 + cl;
          const bool live = c < n_c;
          const float lnv = live ? fin.ln[c] : 0.f;
          float wv[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) wv[r] = live ? w[(size_t)(sr0 + mfma32_row(r, hi)) * ldc + c] : 0.f;
          float dl = 0.f;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float v = acc[i][j][r];
            if (live) g[(size_t)(sr0 + mfma32_row(r, hi)) * ldc + c] = v * lnv;
            dl = __builtin_fmaf(v, wv[r], dl);
          }
          dl += __shfl_xor(dl, 32, 64);  // the fragment's other sixteen rows
          if (hi == 0 && live) fin.dln_part[(size_t)(ob >> 5) * ldc + c] = dl;
        }
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const int c = n_base + j * 32 + cl;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int o = m_base + i * 32 + mfma32_row(r, hi);
          if (o < n_o && c < n_c) out[(size_t)o * ldc + c] = acc[i][j][r];
        }
      }
  }
};

typedef __attribute__((ext_vector_type(4))) short tr4_t;
__device__ __forceinline__ bf16x8 tr_read_pair(const char* lo_p, const char* up_p) {
  const tr4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) tr4_t*)(lo_p));
  const tr4_t up = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) tr4_t*)(up_p));
  return __builtin_shufflevector(lo, up, 0, 1, 2, 3, 4, 5, 6, 7);  // the two halves of one 128-bit operand register tuple
}

// Token tiles [kt0, kt0 + nk) of 64 rows; Y / X rows are always readable (K ranges lie inside the padded token count),
// feature columns are clamped at the matrix edge (the clamped duplicates land in accumulators that are not stored).
template <class C, class Epilogue>
__device__ __forceinline__ void wgrad_tile(const bf16_t* __restrict__ Y, int ldy, int ny, const bf16_t* __restrict__ X,
                                           int ldx, int nx, int kt0, int nk, int tile_m, int tile_n, Epilogue& epi,
                                           char* smem) {
  constexpr int NSTAGE = C::NSTAGE, FM = C::FM, FN = C::FN;
  const int tid = threadIdx.x, lane = tid & 63;
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

  // DMA piece p of an image = chunk p / 8, token rows 8 (p % 8) .. +7; lane l lands at row l / 8, 16-byte slot l % 8
  // and fetches logical slot (l % 8) ^ (2 * bit 1 of the row)
  const bf16_t* a_src[C::A_DMA];
  const bf16_t* w_src[C::W_DMA];
#pragma unroll
  for (int d = 0; d < C::A_DMA; ++d) {
    const int p = wave * C::A_DMA + d;
    const int row = (p & 7) * 8 + (lane >> 3);
    const int slot = (lane & 7) ^ (((row >> 1) & 1) << 1);
    const int feat = min(tile_m * C::BM + (p >> 3) * 64 + slot * 8, ny - 8);
    a_src[d] = Y + (size_t)(kt0 * 64 + row) * ldy + feat;
  }
#pragma unroll
  for (int d = 0; d < C::W_DMA; ++d) {
    const int p = wave * C::W_DMA + d;
    const int row = (p & 7) * 8 + (lane >> 3);
    const int slot = (lane & 7) ^ (((row >> 1) & 1) << 1);
    const int feat = min(tile_n * C::BN + (p >> 3) * 64 + slot * 8, nx - 8);
    w_src[d] = X + (size_t)(kt0 * 64 + row) * ldx + feat;
  }
  auto stage = [&](int kt, int buf) {
    char* base = smem + buf * C::STAGE_BYTES;
#pragma unroll
    for (int d = 0; d < C::A_DMA; ++d)
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(a_src[d] + (size_t)kt * 64 * ldy),
                                       (lds_ptr_t)(base + (wave * C::A_DMA + d) * 1024), 16, 0, 0);
#pragma unroll
    for (int d = 0; d < C::W_DMA; ++d)
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(w_src[d] + (size_t)kt * 64 * ldx),
                                       (lds_ptr_t)(base + C::A_BYTES + (wave * C::W_DMA + d) * 1024), 16, 0, 0);
  };
  constexpr int DPS = C::A_DMA + C::W_DMA;

  // transposing fragment reads: lane = (hi, g1, q, l3); it addresses token row 16 ks + 4 hi + q (and + 8), the 8-byte
  // piece l3 of the 32-byte span of features 16 g1 .. + 15 of its 32-feature fragment, and receives feature
  // (lane & 31) of token rows 4 hi .. + 3 (slots 0-3) and 4 hi + 8 .. + 11 (slots 4-7) - the same token <-> k-slot map
  // for both operands.
  const int g1 = (lane >> 4) & 1, q = (lane & 15) >> 2, l3 = lane & 3;
  const int row0 = 4 * hi + q;
  const int sw = (q >> 1) & 1;  // bit 1 of the row (16 ks, 4 hi and + 8 do not touch it)
  int a_off[FM], b_off[FN];
#pragma unroll
  for (int f = 0; f < FM; ++f) {
    const int f32i = wave_row * FM + f;  // 32-feature fragment index inside the image
    a_off[f] = (f32i >> 1) * 8192 + row0 * 128 + (((2 * (f32i & 1) + g1) ^ sw) << 5) + l3 * 8;
  }
#pragma unroll
  for (int f = 0; f < FN; ++f) {
    const int f32i = wave_col * FN + f;
    b_off[f] = C::A_BYTES + (f32i >> 1) * 8192 + row0 * 128 + (((2 * (f32i & 1) + g1) ^ sw) << 5) + l3 * 8;
  }

#pragma unroll
  for (int s = 0; s < NSTAGE - 1; ++s)
    if (s < nk) stage(s, s);
  int buf = 0;
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + NSTAGE - 2 < nk)
      wait_vmcnt<(NSTAGE - 2) * DPS>();
    else
      wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    if (kt + NSTAGE - 1 < nk) {
      int nb = buf + NSTAGE - 1;
      if (nb >= NSTAGE) nb -= NSTAGE;
      stage(kt + NSTAGE - 1, nb);
    }
    const char* st = smem + buf * C::STAGE_BYTES;
    // k-steps software-pipelined by hand (as rp_gemm.h's gemm_tile_pipe): the fragments of k-step s + 1 are read while
    // the MFMAs of k-step s issue; sched_group_barrier pins ceil(reads / MFMAs) transposing reads behind each MFMA
    // (left to itself the compiler drained lgkmcnt to 0 four times per tile, the first time behind 20 reads).
    bf16x8 af[2][FM], bfr[2][FN];
    auto read_frags = [&](int ks, int pb) {
#pragma unroll
      for (int f = 0; f < FM; ++f) af[pb][f] = tr_read_pair(st + a_off[f] + ks * 2048, st + a_off[f] + ks * 2048 + 1024);
#pragma unroll
      for (int f = 0; f < FN; ++f) bfr[pb][f] = tr_read_pair(st + b_off[f] + ks * 2048, st + b_off[f] + ks * 2048 + 1024);
    };
    read_frags(0, 0);
    __builtin_amdgcn_sched_barrier(0);  // the first k-step's reads are not part of the interleaving below
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      if (ks < 3) read_frags(ks + 1, (ks + 1) & 1);
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks & 1][i], bfr[ks & 1][j], acc[i][j], 0, 0, 0);
      if (ks < 3) {
        constexpr int NREAD = 2 * (FM + FN), NM = FM * FN, PER = (NREAD + NM - 1) / NM;
        int rd = NREAD;
#pragma unroll
        for (int n = 0; n < NM; ++n) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // one MFMA
#pragma unroll
          for (int qq = 0; qq < PER; ++qq)
            if (rd > 0) {
              __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // one DS read
              --rd;
            }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (++buf == NSTAGE) buf = 0;
  }
  epi.template run<FM, FN>(acc, tile_m * C::BM + wave_row * (FM * 32), tile_n * C::BN + wave_col * (FN * 32), lane);
}

// One launch carries up to TWO products over the same token range (round 6: the two weight gradients of a feed-forward
// sub-layer - 168 + 84 tiles of 256 x 256 at d_model 1472 - fill one round of 256 CUs WITHOUT split-K, where each alone
// ran three splits and a finishing pass over its partial matrices).  grid = splits * (tiles of p0 + tiles of p1): the
// first blocks0 logical ids belong to p0.  Split s reduces token tiles [s nk / S, (s + 1) nk / S) into its own fp32 matrix
// out + s * split_stride (summed in split order by the finishing kernel: no atomics).
struct WgradProblem {
  const bf16_t* Y;  // [T, ny]
  int ldy, ny;
  const bf16_t* X;  // [T, nx]
  int ldx, nx;
  int tiles_m, tiles_n;
  float* out;  // [splits][ny, ldc]
  int ldc;
  size_t split_stride;
  WgradFinish fin;  // splits == 1 only
};
template <class C>
__global__ __launch_bounds__(C::THREADS) void wgrad_kernel(WgradProblem p0, WgradProblem p1, int blocks0, int nk_total,
                                                           int splits) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int logical = xcd_remap(blockIdx.x, gridDim.x);
  const bool second = logical >= blocks0;  // uniform across the workgroup
  if (second) logical -= blocks0;
  const bf16_t* Y = second ? p1.Y : p0.Y;
  const bf16_t* X = second ? p1.X : p0.X;
  const int ldy = second ? p1.ldy : p0.ldy, ny = second ? p1.ny : p0.ny;
  const int ldx = second ? p1.ldx : p0.ldx, nx = second ? p1.nx : p0.nx;
  const int tiles_m = second ? p1.tiles_m : p0.tiles_m, tiles_n = second ? p1.tiles_n : p0.tiles_n;
  float* out = second ? p1.out : p0.out;
  const int ldc = second ? p1.ldc : p0.ldc;
  const size_t split_stride = second ? p1.split_stride : p0.split_stride;
  const int per_split = tiles_m * tiles_n;
  const int s = logical / per_split, t = logical - s * per_split;
  const int tm = t / tiles_n, tn = t - tm * tiles_n;
  const int kt0 = (int)((long long)s * nk_total / splits), kt1 = (int)((long long)(s + 1) * nk_total / splits);
  EpiStoreF32 epi{out + (size_t)s * split_stride, ldc, ny, nx, WgradFinish{}};
  if (!second) epi.fin = p0.fin;  // (a finishing epilogue rides on the first product only)
  wgrad_tile<C>(Y, ldy, ny, X, ldx, nx, kt0, kt1 - kt0, tm, tn, epi, smem);
}

// ------------------------------------------------------------------------------------------
// dgrad epilogues (the forward's GEMM core, weights transposed once per optimizer step)
// ------------------------------------------------------------------------------------------
// Backward of ff = gelu_new(g) * u behind  dff = dx Wo2  (tile rows = d_ff features f, columns = tokens):
//   dy_g = dff * u * gelu_new'(g),  dy_u = dff * gelu_new(g);   dzs = rs * dy (packed order: 32 gate | 32 up per 64);
//   rdp[slot][token] = sum over the slot's 64 features f of (dy_g g + dy_u u)      (RMSNorm backward's row dot)
// g, u are the saved row-scaled pre-activations (bf16, packed order as the forward's weight interleave produced them).
struct EpiGegluBwd {
  static constexpr bool loose_mixed = true;
  const bf16_t* __restrict__ gu;  // [tokens, ld2]
  bf16_t* __restrict__ dzs;       // [tokens, ld2]
  int ld2, n_valid;               // ld2 = 2 d_ff, n_valid = d_ff (multiple of 64)
  const float* __restrict__ rs;   // [tokens]
  float* __restrict__ rdp;        // [np, ld_t]
  int np, ld_t;
  Drop drop;                      // dropout between the gated product and wo (HF:110): d(product) = mask * dff
  uint32_t drop_site;
  template <int FM, int FN>
  __device__ __forceinline__ void run(f32x16 (&acc)[FM][FN], int m_base, int n_base, int lane, char* stage) {
    static_assert(FM % 2 == 0, "blocks of 64 features (two row fragments)");
    const int hi = lane >> 5, cl = lane & 31;
    const int sub = lane & 7, rr = lane >> 3;
    constexpr int RB = 272;
    constexpr int NB = (FM / 2) * FN;
    // lane's 8 features: fragment sub / 4 of the pair, offset 8 (sub % 4); packed column of its gate values
    const int pcol_in = (sub >> 2) * 64 + (sub & 3) * 8;
    uint4 gg[2][4], uu[2][4];
    auto fetch = [&](int b, int p) {
      const int qb = b / FN, j = b % FN;
      const int f0 = min(m_base + qb * 64, n_valid - 64);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const size_t off = (size_t)(n_base + j * 32 + c * 8 + rr) * ld2 + 2 * f0 + pcol_in;
        gg[p][c] = *reinterpret_cast<const uint4*>(gu + off);
        uu[p][c] = *reinterpret_cast<const uint4*>(gu + off + 32);
      }
    };
    fetch(0, 0);
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const int qb = b / FN, j = b % FN;
      const int f0 = m_base + qb * 64;
      const int slot = f0 >> 6;
      if (b + 1 < NB) fetch(b + 1, (b + 1) & 1);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *reinterpret_cast<float4*>(stage + cl * RB + (i * 32 + 8 * g + 4 * hi) * 4) =
              make_float4(acc[2 * qb + i][j][4 * g], acc[2 * qb + i][j][4 * g + 1], acc[2 * qb + i][j][4 * g + 2],
                          acc[2 * qb + i][j][4 * g + 3]);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int t = c * 8 + rr;
        const int token = n_base + j * 32 + t;
        const float4 d0 = *reinterpret_cast<const float4*>(stage + t * RB + sub * 32);
        const float4 d1 = *reinterpret_cast<const float4*>(stage + t * RB + sub * 32 + 16);
        float dff[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
        if (drop.thresh) {
#pragma unroll
          for (int e = 0; e < 8; e += 2) {
            float m0, m1;
            drop_mul2(drop, drop_site, (uint32_t)token, (uint32_t)(f0 + sub * 8 + e), m0, m1);
            dff[e] *= m0;
            dff[e + 1] *= m1;
          }
        }
        const uint4 gq = gg[b & 1][c], uq = uu[b & 1][c];
        const uint32_t gw[4] = {gq.x, gq.y, gq.z, gq.w}, uw[4] = {uq.x, uq.y, uq.z, uq.w};
        const float rsv = rs[token];
        // gelu_new(g) = g * sg, sg = sigmoid(2 z), 2 z = 2 sqrt(2/pi) (g + 0.044715 g^3); element pairs as 2-vectors so that
        // everything but v_exp_f32 / v_rcp_f32 is a packed fp32 instruction (round 6; rp_util.h::geglu2 is the forward's)
        constexpr float k1 = -2.0f * 0.7978845608028654f * 1.4426950408889634f;
        constexpr float k2 = k1 * 0.044715f;
        constexpr float d2 = 2.0f * 0.7978845608028654f, d2c3 = d2 * 3.0f * 0.044715f;
        const f32x2 K1 = {k1, k1}, K2 = {k2, k2}, D2 = {d2, d2}, D2C3 = {d2c3, d2c3}, ONE = {1.f, 1.f}, RS = {rsv, rsv};
        f32x2 dot2 = {0.f, 0.f};
        uint32_t ogw[4], ouw[4];
#pragma unroll
        for (int e2 = 0; e2 < 4; ++e2) {
          const f32x2 gv = {__uint_as_float(gw[e2] << 16), __uint_as_float(gw[e2] & 0xffff0000u)};
          const f32x2 uv = {__uint_as_float(uw[e2] << 16), __uint_as_float(uw[e2] & 0xffff0000u)};
          const f32x2 df = {dff[2 * e2], dff[2 * e2 + 1]};
          const f32x2 g2 = gv * gv;
          const f32x2 a = gv * __builtin_elementwise_fma(K2, g2, K1);
          const f32x2 den = f32x2{__builtin_amdgcn_exp2f(a.x), __builtin_amdgcn_exp2f(a.y)} + ONE;
          const f32x2 sg = {__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y)};
          const f32x2 d2z = __builtin_elementwise_fma(D2C3, g2, D2);
          const f32x2 gs = gv * sg;                                   // gelu_new(g)
          const f32x2 t = __builtin_elementwise_fma(-sg, sg, sg);     // sg (1 - sg)
          const f32x2 dgelu = __builtin_elementwise_fma(gv * t, d2z, sg);
          const f32x2 dyg = (df * uv) * dgelu, dyu = df * gs;
          dot2 = __builtin_elementwise_fma(dyg, gv, __builtin_elementwise_fma(dyu, uv, dot2));
          const f32x2 og = dyg * RS, ou = dyu * RS;
          ogw[e2] = pack_bf2(og.x, og.y);
          ouw[e2] = pack_bf2(ou.x, ou.y);
        }
        float dot = dot2.x + dot2.y;
        if (f0 < n_valid) {
          const size_t off = (size_t)token * ld2 + 2 * f0 + pcol_in;
          *reinterpret_cast<uint4*>(dzs + off) = make_uint4(ogw[0], ogw[1], ogw[2], ogw[3]);
          *reinterpret_cast<uint4*>(dzs + off + 32) = make_uint4(ouw[0], ouw[1], ouw[2], ouw[3]);
        } else {
          dot = 0.f;
        }
        // the token's 64 features sit on 8 neighbouring lanes: three DPP adds (quad_perm(1,0,3,2), quad_perm(2,3,0,1),
        // row_half_mirror) leave the sum on every one of them
        dot += dpp_f32<0xB1>(dot);
        dot += dpp_f32<0x4E>(dot);
        dot += dpp_f32<0x141>(dot);
        if (sub == 0 && slot < np) rdp[(size_t)slot * ld_t + token] = dot;
      }
    }
  }
};

// dx += v - x * rcoef[token]  on the two planes of the residual-stream gradient (v = the dgrad accumulators,
// x = the sub-layer's saved input, bf16; rcoef = rs^2 * rowdot / D): RMSNorm backward fused into the dgrad GEMM.
// MASK (round 6): the updated gradient is the input of the NEXT branch's backward, which under dropout wants bf16(mask dx)
// as the operand of its two GEMMs - written here from the words just stored (mask_dx_kernel's arithmetic on the same
// inputs: the same bits) instead of a separate pass over both planes per branch (24 launches of 14 us per step).
template <bool MASK>
struct EpiRmsBwdResidT {
  static constexpr bool any_layout = true;  // blocks of 64 features per wave, whatever the wave grid (the edge tile of d_model 1472)
  bf16_t* __restrict__ dxhi;
  bf16_t* __restrict__ dxlo;
  int ldx, n_valid;  // n_valid % 8 == 0
  const bf16_t* __restrict__ xs;
  const float* __restrict__ rcoef;
  bf16_t* __restrict__ dxm = nullptr;  // MASK: [tokens, ldx] = bf16(mask(next_site) * (hi + lo))
  Drop drop = Drop{0u, 0u, 1.f};
  uint32_t next_site = 0;
  template <int FM, int FN>
  __device__ __forceinline__ void run(f32x16 (&acc)[FM][FN], int m_base, int n_base, int lane, char* stage) {
    static_assert(FM % 2 == 0, "blocks of 64 features");
    const int hi = lane >> 5, cl = lane & 31;
    const int sub = lane & 7, rr = lane >> 3;
    constexpr int RB = 272;
    constexpr int NB = (FM / 2) * FN;
    uint4 xh[2][4], xl[2][4], xv[2][4];
    auto fetch = [&](int b, int p) {
      const int qb = b / FN, j = b % FN;
      const int f = min(m_base + qb * 64 + sub * 8, n_valid - 8);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const size_t off = (size_t)(n_base + j * 32 + c * 8 + rr) * ldx + f;
        xh[p][c] = *reinterpret_cast<const uint4*>(dxhi + off);
        xl[p][c] = *reinterpret_cast<const uint4*>(dxlo + off);
        xv[p][c] = *reinterpret_cast<const uint4*>(xs + off);
      }
    };
    fetch(0, 0);
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const int qb = b / FN, j = b % FN;
      const int f = m_base + qb * 64 + sub * 8;
      if (b + 1 < NB) fetch(b + 1, (b + 1) & 1);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *reinterpret_cast<float4*>(stage + cl * RB + (i * 32 + 8 * g + 4 * hi) * 4) =
              make_float4(acc[2 * qb + i][j][4 * g], acc[2 * qb + i][j][4 * g + 1], acc[2 * qb + i][j][4 * g + 2],
                          acc[2 * qb + i][j][4 * g + 3]);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int t = c * 8 + rr;
        const int token = n_base + j * 32 + t;
        const float4 d0 = *reinterpret_cast<const float4*>(stage + t * RB + sub * 32);
        const float4 d1 = *reinterpret_cast<const float4*>(stage + t * RB + sub * 32 + 16);
        if (f < n_valid) {
          const float rc = rcoef[token];
          const size_t off = (size_t)token * ldx + f;
          const uint4 h = xh[b & 1][c], l = xl[b & 1][c], x = xv[b & 1][c];
          auto lo16 = [](uint32_t w) { return __uint_as_float(w << 16); };
          auto hi16 = [](uint32_t w) { return __uint_as_float(w & 0xffff0000u); };
          uint4 oh, ol;
          float ss = 0.f;
          hilo_update2(h.x, l.x, __builtin_fmaf(-rc, lo16(x.x), d0.x), __builtin_fmaf(-rc, hi16(x.x), d0.y), oh.x, ol.x, ss);
          hilo_update2(h.y, l.y, __builtin_fmaf(-rc, lo16(x.y), d0.z), __builtin_fmaf(-rc, hi16(x.y), d0.w), oh.y, ol.y, ss);
          hilo_update2(h.z, l.z, __builtin_fmaf(-rc, lo16(x.z), d1.x), __builtin_fmaf(-rc, hi16(x.z), d1.y), oh.z, ol.z, ss);
          hilo_update2(h.w, l.w, __builtin_fmaf(-rc, lo16(x.w), d1.z), __builtin_fmaf(-rc, hi16(x.w), d1.w), oh.w, ol.w, ss);
          *reinterpret_cast<uint4*>(dxhi + off) = oh;
          *reinterpret_cast<uint4*>(dxlo + off) = ol;
          if constexpr (MASK) {
            const uint32_t hw[4] = {oh.x, oh.y, oh.z, oh.w}, lw[4] = {ol.x, ol.y, ol.z, ol.w};
            uint32_t ow[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float m0, m1;
              drop_mul2(drop, next_site, (uint32_t)token, (uint32_t)(f + 2 * e), m0, m1);
              ow[e] = pack_bf2((lo16(hw[e]) + lo16(lw[e])) * m0, (hi16(hw[e]) + hi16(lw[e])) * m1);
            }
            *reinterpret_cast<uint4*>(dxm + off) = make_uint4(ow[0], ow[1], ow[2], ow[3]);
          }
        }
      }
    }
  }
};
using EpiRmsBwdResid = EpiRmsBwdResidT<false>;

// training forward, FFN-in: the row-scaled pre-activations (packed order) the backward needs AND the gated-GELU output with
// the dropout that sits between it and wo (HF:110).  The product part is EpiGegluBf16T's, plus the mask.
template <bool DROP>
struct EpiGegluTrainT {
  // (not `loose_mixed`: 28 x 41 = 1148 tiles plan as 4.5 instead of 5 periods, but the persistent form this launch takes
  // otherwise measured faster - 2.73 vs 2.80 ms per step)
  EpiStoreBf16 st;  // gu [tokens, 2 d_ff]
  bf16_t* out;      // ff [tokens, d_ff] = mask * gelu_new(g) * u
  int ldo, n_valid;
  RowScale rs;
  Drop drop;
  uint32_t drop_site;
  template <int FM, int FN>
  __device__ __forceinline__ void run(f32x16 (&acc)[FM][FN], int m_base, int n_base, int lane, char* stage) {
    st.template run<FM, FN>(acc, m_base, n_base, lane, stage);  // same wave-private staging area, LDS ops of a wave are in order
    static_assert(FM % 2 == 0 && FM * 32 <= 128, "gate/up fragment pairs; staging row");
    const int hi = lane >> 5, cl = lane & 31;
    constexpr int LPR = FM * 2, RPI = 64 / LPR;
    const int sub = lane % LPR, rr = lane / LPR;
    const int f = (m_base >> 1) + sub * 8;
    float scv[FN];
#pragma unroll
    for (int j = 0; j < FN; ++j) scv[j] = rs.get(n_base + j * 32 + cl);
#pragma unroll
    for (int jb = 0; jb < FN; jb += 2) {
#pragma unroll
      for (int jj = 0; jj < 2 && jb + jj < FN; ++jj) {
        const GegluConsts gc(scv[jb + jj]);  // the token's RMSNorm factor folded into the activation's constants
        const uint32_t row = (uint32_t)(n_base + (jb + jj) * 32 + cl);
#pragma unroll
        for (int i = 0; i < FM; i += 2)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            float y[4];
#pragma unroll
            for (int e = 0; e < 4; e += 2) {  // the inference epilogue's packed form (rp_util.h::geglu2): the same bits as it
              const f32x2 yy = geglu2(f32x2{acc[i][jb + jj][4 * g + e], acc[i][jb + jj][4 * g + e + 1]},
                                      f32x2{acc[i + 1][jb + jj][4 * g + e], acc[i + 1][jb + jj][4 * g + e + 1]}, gc);
              y[e] = yy.x;
              y[e + 1] = yy.y;
            }
            if constexpr (DROP) {
#pragma unroll
              for (int e = 0; e < 4; e += 2) {
                float m0, m1;
                drop_mul2(drop, drop_site, row, (uint32_t)((m_base >> 1) + (i >> 1) * 32 + 8 * g + 4 * hi + e), m0, m1);
                y[e] *= m0;
                y[e + 1] *= m1;
              }
            }
            uint2 v;
            v.x = pack_bf2(y[0], y[1]);
            v.y = pack_bf2(y[2], y[3]);
            *reinterpret_cast<uint2*>(stage + (jj * 32 + cl) * EPI_ROW_BYTES + ((i >> 1) * 32 + 8 * g + 4 * hi) * 2) = v;
          }
      }
      const int nrows = (FN - jb >= 2) ? 64 : 32;
#pragma unroll
      for (int t0 = 0; t0 < 64; t0 += RPI) {
        const int tt = t0 + rr;
        if (tt < nrows) {
          const uint4 v = *reinterpret_cast<const uint4*>(stage + tt * EPI_ROW_BYTES + sub * 16);
          if (2 * f < n_valid) *reinterpret_cast<uint4*>(out + (size_t)(n_base + jb * 32 + tt) * ldo + f) = v;
        }
      }
    }
  }
};

// ------------------------------------------------------------------------------------------
// attention backward (flash-style, varlen): P is recomputed from q, k, the bias table and the saved log-sum-exp.
//   dP = dO V^T,  delta_i = sum_d dO_i O_i,  dS = P (dP - delta),  dQ = dS K,  dK = dS^T Q,  dV = P^T dO  (no 1/sqrt(d))
//   d bias_tab[h][clamp(j - i) + maxd] += dS_ij
// Two launches of one kernel template, no atomics on HBM:
//   MODE 0 (dQ):    workgroup = 128 queries x one head, keys/values streamed;   also emits delta and the table gradient
//   MODE 1 (dK/dV): workgroup = 128 keys x one head, queries / dO streamed (stats lse, delta ride along)
// Both are the forward kernel's shape: S^T[m, n] = X1 R1^T with the resident index n (query / key) on the lanes, so the
// accumulators of S^T / dS^T are the B operands of the second MFMAs as they stand, and the streamed tile's transpose
// (A operand of dQ^T += K^T dS^T etc.) comes from the transposing LDS read.  Streamed tiles are staged ONCE, in a layout
// that serves both the plain and the transposing fragment reads: [d half][row][32 d = 64 B] with the 16-byte slots of a
// row XOR-swizzled by (row / 4) % 4.
// ------------------------------------------------------------------------------------------
constexpr int AB_TILE = 64 * 128;                  // 64 rows x 64 d, bf16
constexpr int AB_STAGE = 2 * AB_TILE + 2 * 256;    // X1, X2, two vectors of 64 floats (MODE 1: lse2, delta)

template <int MODE, bool DROP>
__global__ __launch_bounds__(256, 2) void attn_bwd_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ att,
                                                          const bf16_t* __restrict__ datt, const float* __restrict__ lse2,
                                                          float* __restrict__ delta, const int4* __restrict__ work,
                                                          const float* __restrict__ bias_tab, bf16_t* __restrict__ dqkv,
                                                          float* __restrict__ dtab_part, int H, int maxd, int ld_stat,
                                                          int dbg_flags, Drop drop, uint32_t drop_site) {
  constexpr int WTAB = (MODE == 0) ? 4 * ATT_TAB_MAX * 4 : 0;
  __shared__ __attribute__((aligned(16))) char smem[2 * AB_STAGE + ATT_TAB_MAX * 4 + WTAB];
  float* tab = reinterpret_cast<float*>(smem + 2 * AB_STAGE);
  float* wtab = reinterpret_cast<float*>(smem + 2 * AB_STAGE + ATT_TAB_MAX * 4);  // MODE 0: one table per wave

  const int tid = threadIdx.x, lane = tid & 63, hi = lane >> 5, cl = lane & 31;
  const int h = blockIdx.x;
  const int4 wk = work[blockIdx.y];
  const int s0 = wk.x, len = wk.y, n0 = wk.z;
  if (len == 0) return;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int inner = H * 64, ld = 3 * inner;
  const int ntab = 2 * maxd + 1;
  for (int i = tid; i < ntab; i += 256) tab[i] = bias_tab[h * ntab + i];
  if (MODE == 0)
    for (int i = tid; i < 4 * ATT_TAB_MAX; i += 256) wtab[i] = 0.f;

  const int wn0 = n0 + wave * 32;
  const bool active = wn0 < len;  // wave-uniform
  const int ni = wn0 + cl;
  const bool n_real = ni < len;
  const size_t nrow = (size_t)(s0 + min(ni, len - 1));
  // resident fragments (B operands): lane (cl, hi) holds d = 16 c + 8 hi .. + 7 of its row
  bf16x8 r1f[4], r2f[4];
  float lse_n = 0.f, delta_n = 0.f;
  if (MODE == 0) {
    const bf16_t* qp = qkv + nrow * ld + h * 64 + hi * 8;
    const bf16_t* dop = datt + nrow * inner + h * 64 + hi * 8;
    const bf16_t* op = att + nrow * inner + h * 64 + hi * 8;
    float dl = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      r1f[c] = *reinterpret_cast<const bf16x8*>(qp + c * 16);
      r2f[c] = *reinterpret_cast<const bf16x8*>(dop + c * 16);
      const bf16x8 of = *reinterpret_cast<const bf16x8*>(op + c * 16);
#pragma unroll
      for (int e = 0; e < 8; ++e) dl = __builtin_fmaf(bf2f((bf16_t)r2f[c][e]), bf2f((bf16_t)of[e]), dl);
    }
    delta_n = dl + __shfl_xor(dl, 32, 64);
    lse_n = lse2[(size_t)h * ld_stat + nrow];
    if (n_real && hi == 0) delta[(size_t)h * ld_stat + nrow] = delta_n;
  } else {
    const bf16_t* kp = qkv + nrow * ld + inner + h * 64 + hi * 8;
    const bf16_t* vp = qkv + nrow * ld + 2 * inner + h * 64 + hi * 8;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      r1f[c] = *reinterpret_cast<const bf16x8*>(kp + c * 16);
      r2f[c] = *reinterpret_cast<const bf16x8*>(vp + c * 16);
    }
  }

  // streamed operands
  const bf16_t* x1 = (MODE == 0) ? qkv + (size_t)s0 * ld + inner + h * 64 : qkv + (size_t)s0 * ld + h * 64;
  const bf16_t* x2 = (MODE == 0) ? qkv + (size_t)s0 * ld + 2 * inner + h * 64 : datt + (size_t)s0 * inner + h * 64;
  const int ld1 = ld, ld2 = (MODE == 0) ? ld : inner;
  const float* st_lse = lse2 + (size_t)h * ld_stat + s0;
  const float* st_del = delta + (size_t)h * ld_stat + s0;
  auto stage = [&](int kt, int buf) {
    char* base = smem + buf * AB_STAGE;
    const int m0 = kt * 64;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int p = wave * 2 + e;  // piece: d half p / 4, rows 16 (p % 4) .. + 15
      const int row = 16 * (p & 3) + (lane >> 2);
      const int slot = (lane & 3) ^ ((row >> 2) & 3);
      const size_t r = (size_t)min(m0 + row, len - 1);
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(x1 + r * ld1 + (p >> 2) * 32 + slot * 8), (lds_ptr_t)(base + p * 1024), 16,
                                       0, 0);
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(x2 + r * ld2 + (p >> 2) * 32 + slot * 8),
                                       (lds_ptr_t)(base + AB_TILE + p * 1024), 16, 0, 0);
    }
    if (MODE == 1 && wave < 2) {
      const float* src = (wave == 0 ? st_lse : st_del) + min(m0 + lane, len - 1);
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)(base + 2 * AB_TILE + wave * 256), 4, 0, 0);
    }
  };

  // plain fragment reads (A operand: row = streamed index, 16 d per k-step)
  int n_off[2][4];
#pragma unroll
  for (int mb = 0; mb < 2; ++mb)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int row = mb * 32 + cl;
      n_off[mb][c] = (c >> 1) * 4096 + row * 64 + (((2 * (c & 1) + hi) ^ ((row >> 2) & 3)) << 4);
    }
  // transposing reads: rows 4 hi + q (slots 0-3) and + 8 (slots 4-7) of a 16-row slab, d = 16 g1 + 4 l3 .. of a d half
  const int g1 = (lane >> 4) & 1, q4 = (lane & 15) >> 2, l3 = lane & 3;
  const int cl16 = 2 * g1 + (l3 >> 1);
  const int t_lo = (4 * hi + q4) * 64 + ((cl16 ^ hi) << 4) + 8 * (l3 & 1);
  const int t_up = (4 * hi + q4 + 8) * 64 + ((cl16 ^ (hi ^ 2)) << 4) + 8 * (l3 & 1);

  f32x16 acc1[2], acc2[2];  // MODE 0: acc1 = dQ^T;  MODE 1: acc1 = dK^T, acc2 = dV^T   ([d half][d, n])
#pragma unroll
  for (int d = 0; d < 2; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc1[d][r] = acc2[d][r] = 0.f;
  float g_lo = 0.f, g_hi = 0.f;  // MODE 0: table gradient of the saturated offsets

  const float LOG2E = 1.4426950408889634f;
  const int n_tiles = (len + 63) / 64;
  stage(0, 0);
  for (int kt = 0; kt < n_tiles; ++kt) {
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    if (kt + 1 < n_tiles) stage(kt + 1, (kt + 1) & 1);
    if (!active) continue;
    const char* sb = smem + (kt & 1) * AB_STAGE;
    const float* sstat = reinterpret_cast<const float*>(sb + 2 * AB_TILE);
    const int m0 = kt * 64;
    f32x16 s[2], dp[2];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[mb][r] = dp[mb][r] = 0.f;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const bf16x8 f1 = *reinterpret_cast<const bf16x8*>(sb + n_off[mb][c]);
        const bf16x8 f2 = *reinterpret_cast<const bf16x8*>(sb + AB_TILE + n_off[mb][c]);
        s[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f1, r1f[c], s[mb], 0, 0, 0);
        dp[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f2, r2f[c], dp[mb], 0, 0, 0);
      }
    }
    // ---- P = 2^((s + bias) log2 e - lse2), dS = P (dP - delta); rel = key - query
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
      const int c0 = m0 + mb * 32;
      const bool all_real = (c0 + 32 <= len);
      // offsets of the whole 32 x 32 block: streamed rows c0 .. c0 + 31 against the wave's wn0 .. wn0 + 31
      const int rel_min = (MODE == 0) ? c0 - (wn0 + 31) : wn0 - (c0 + 31);
      const int rel_max = (MODE == 0) ? c0 + 31 - wn0 : wn0 + 31 - c0;
      const int sat = (all_real && rel_min >= maxd) ? 2 : (all_real && rel_max <= -maxd) ? 1 : 0;  // wave-uniform
      const float bsat = sat == 2 ? tab[2 * maxd] : tab[0];
      float gsum = 0.f;
      // Table gradient of a non-saturated block: every accumulator row is rotated by its own row number (one
      // ds_bpermute), which brings the entries of a diagonal (key - query constant) into the same lane; a lane then
      // holds two diagonals of the 32 x 32 fragment - offset -l and 32 - l - summed in registers, and the block costs two
      // LDS atomics per lane instead of sixteen (ds_add_f32 runs at a fraction of the plain LDS rate: 0.9 of the
      // backward attention's 2.2 ms per step went into them).
      float diag_a = 0.f, diag_b = 0.f;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float lse_m[4], del_m[4];
        if (MODE == 1) {
          const float4 lv = *reinterpret_cast<const float4*>(sstat + mb * 32 + 8 * g + 4 * hi);
          const float4 dv = *reinterpret_cast<const float4*>(sstat + 64 + mb * 32 + 8 * g + 4 * hi);
          lse_m[0] = lv.x; lse_m[1] = lv.y; lse_m[2] = lv.z; lse_m[3] = lv.w;
          del_m[0] = dv.x; del_m[1] = dv.y; del_m[2] = dv.z; del_m[3] = dv.w;
        }
        float dmul[4];
        if constexpr (DROP) {
          const int jg = c0 + 8 * g + 4 * hi;  // even: the lane's four streamed indices are jg .. jg + 3
          if (MODE == 0) {  // streamed keys: two column pairs of the query's row
            drop_mul2(drop, drop_site, (uint32_t)(s0 + ni), ((uint32_t)h << 20) | (uint32_t)jg, dmul[0], dmul[1]);
            drop_mul2(drop, drop_site, (uint32_t)(s0 + ni), ((uint32_t)h << 20) | (uint32_t)(jg + 2), dmul[2], dmul[3]);
          } else {  // streamed queries: the lane's key picks the field, the row advances
#pragma unroll
            for (int e = 0; e < 4; ++e) dmul[e] = drop_mul(drop, drop_site, (uint32_t)(s0 + jg + e), ((uint32_t)h << 20) | (uint32_t)ni);
          }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = 4 * g + e;
          const int j = c0 + 8 * g + 4 * hi + e;  // streamed index of this accumulator row
          const int rel = (MODE == 0) ? j - ni : ni - j;
          const int idx = min(max(rel, -maxd), maxd) + maxd;
          const float b = sat ? bsat : tab[idx];
          const float lse = (MODE == 0) ? lse_n : lse_m[e];
          const float del = (MODE == 0) ? delta_n : del_m[e];
          float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[mb][r] + b, LOG2E, -lse));
          if (j >= len || !n_real) p = 0.f;
          // dropout on the probabilities (HF:168): O = (mask P) V, so dV takes mask P, dP = mask (dO V^T), and
          // delta = dO . O already carries the mask
          float pm = p, dpv = dp[mb][r];
          if constexpr (DROP) {
            pm *= dmul[e];
            dpv *= dmul[e];
          }
          const float ds = p * (dpv - del);
          s[mb][r] = pm;
          dp[mb][r] = ds;
          if (MODE == 0) {
            if (sat) {
              gsum += ds;
            } else {
              const int kr = 8 * g + 4 * hi + e;  // row of this accumulator inside the fragment
              const float w = __shfl(ds, (hi << 5) | ((cl + kr) & 31), 64);  // masked entries carry ds = 0
              if (cl + kr < 32)
                diag_a += w;
              else
                diag_b += w;
            }
          }
        }
      }
      if (MODE == 0) {
        if (sat == 2) g_hi += gsum;
        if (sat == 1) g_lo += gsum;
        if (!sat && !(dbg_flags & 1)) {
          // MODE 0: rel = key - query = (c0 + kr) - (wn0 + column); the lane's diagonals: kr - column = -cl and 32 - cl
          const int rel_a = c0 - wn0 - cl, rel_b = rel_a + 32;
          atomicAdd(&wtab[wave * ATT_TAB_MAX + min(max(rel_a, -maxd), maxd) + maxd], diag_a);  // LDS, wave-private table
          atomicAdd(&wtab[wave * ATT_TAB_MAX + min(max(rel_b, -maxd), maxd) + maxd], diag_b);
        }
      }
    }
    // ---- second MFMAs over four 16-row slabs of the streamed tile
#pragma unroll
    for (int sl = 0; sl < 4; ++sl) {
      const int mb = sl >> 1, sub = sl & 1;
      bf16x8 pf, dsf;
      {
        uint32_t pw[4], dw[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          pw[e] = pack_bf2(s[mb][8 * sub + 2 * e], s[mb][8 * sub + 2 * e + 1]);
          dw[e] = pack_bf2(dp[mb][8 * sub + 2 * e], dp[mb][8 * sub + 2 * e + 1]);
        }
        uint4 t = make_uint4(pw[0], pw[1], pw[2], pw[3]);
        pf = *reinterpret_cast<bf16x8*>(&t);
        uint4 u = make_uint4(dw[0], dw[1], dw[2], dw[3]);
        dsf = *reinterpret_cast<bf16x8*>(&u);
      }
#pragma unroll
      for (int d = 0; d < 2; ++d) {
        const char* b1 = sb + d * 4096 + sl * 1024;
        const bf16x8 x1t = tr_read_pair(b1 + t_lo, b1 + t_up);
        acc1[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x1t, dsf, acc1[d], 0, 0, 0);
        if (MODE == 1) {
          const bf16x8 x2t = tr_read_pair(b1 + AB_TILE + t_lo, b1 + AB_TILE + t_up);
          acc2[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x2t, pf, acc2[d], 0, 0, 0);
        }
      }
    }
  }

  if (active && n_real) {
    auto store = [&](const f32x16 (&o)[2], int col0) {
      bf16_t* op = dqkv + nrow * ld + col0 + h * 64;
#pragma unroll
      for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          uint2 v;
          v.x = pack_bf2(o[d][4 * g], o[d][4 * g + 1]);
          v.y = pack_bf2(o[d][4 * g + 2], o[d][4 * g + 3]);
          *reinterpret_cast<uint2*>(op + d * 32 + 8 * g + 4 * hi) = v;
        }
    };
    if (MODE == 0) {
      store(acc1, 0);
    } else {
      store(acc1, inner);
      store(acc2, 2 * inner);
    }
  }
  if (MODE == 0) {
    // table gradient of this workgroup: saturated sums by a fixed shuffle tree, wave tables in wave order, then
    // accumulated (across the layers: the bias is shared) into the workgroup's own row of dtab_part
    g_lo = wave_sum(g_lo);
    g_hi = wave_sum(g_hi);
    if (lane == 0) {
      wtab[wave * ATT_TAB_MAX] += g_lo;
      wtab[wave * ATT_TAB_MAX + 2 * maxd] += g_hi;
    }
    __syncthreads();
    float* dst = dtab_part + ((size_t)blockIdx.y * H + h) * ntab;
    for (int i = tid; i < ntab; i += 256)
      dst[i] += (wtab[i] + wtab[ATT_TAB_MAX + i]) + (wtab[2 * ATT_TAB_MAX + i] + wtab[3 * ATT_TAB_MAX + i]);
  }
}

// ------------------------------------------------------------------------------------------
// dropout variants of the forward's memory-bound kernels (training only)
// ------------------------------------------------------------------------------------------
// embed_kernel + dropout on the embeddings (HF:725)
__global__ __launch_bounds__(256) void embed_train_kernel(const int32_t* __restrict__ ids, const float* __restrict__ table,
                                                          bf16_t* __restrict__ xhi, bf16_t* __restrict__ xlo,
                                                          float* __restrict__ ssp, int np, int T, int Tp, int D, int vocab,
                                                          Drop drop) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= Tp) return;
  int id = (row < T) ? ids[row] : 0;
  id = min(max(id, 0), vocab - 1);
  const float4* src = reinterpret_cast<const float4*>(table + (size_t)id * D);
  uint4* dh = reinterpret_cast<uint4*>(xhi + (size_t)row * D);
  uint4* dl = reinterpret_cast<uint4*>(xlo + (size_t)row * D);
  float ss = 0.f;
  for (int c = lane; c < (D >> 3); c += 64) {
    const float4 a = src[2 * c], b = src[2 * c + 1];
    float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] *= drop_mul(drop, DROP_SITE_EMBED, (uint32_t)row, (uint32_t)(c * 8 + e));
    uint4 oh, ol;
    hilo_update2(0u, 0u, v[0], v[1], oh.x, ol.x, ss);
    hilo_update2(0u, 0u, v[2], v[3], oh.y, ol.y, ss);
    hilo_update2(0u, 0u, v[4], v[5], oh.z, ol.z, ss);
    hilo_update2(0u, 0u, v[6], v[7], oh.w, ol.w, ss);
    dh[c] = oh;
    dl[c] = ol;
  }
  ss = wave_sum(ss);
  for (int p = lane; p < np; p += 64) ssp[(size_t)p * Tp + row] = (p == 0) ? ss : 0.f;
}

// pool_partial_kernel + dropout on the final RMSNorm's output (HF:745): sum over the chunk's tokens of mask * x * rs
template <int NV>
__global__ __launch_bounds__(256) void pool_partial_train_kernel(const bf16_t* __restrict__ xhi, const bf16_t* __restrict__ xlo,
                                                                 const float* __restrict__ rs,
                                                                 const int4* __restrict__ pwork, float* __restrict__ partial,
                                                                 int D, Drop drop) {
  __shared__ float red[4][NV * 64 * 8];
  const int4 wk = pwork[blockIdx.x];
  const int s0 = wk.x, len = wk.y, c = wk.z, b = wk.w;
  if (len == 0) return;
  const int t0 = c * POOL_CHUNK, t1 = min(len, t0 + POOL_CHUNK);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nv = D >> 3;
  float acc[NV][8];
#pragma unroll
  for (int i = 0; i < NV; ++i)
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[i][e] = 0.f;
  for (int t = t0 + wave; t < t1; t += 4) {  // tokens in index order per wave, as the inference kernel
    const size_t row = (size_t)(s0 + t) * D;
    const float r = rs[s0 + t];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int col = min(lane + 64 * i, nv - 1);
      const uint4 vh = reinterpret_cast<const uint4*>(xhi + row)[col], vl = reinterpret_cast<const uint4*>(xlo + row)[col];
      const uint32_t hw[4] = {vh.x, vh.y, vh.z, vh.w}, lw[4] = {vl.x, vl.y, vl.z, vl.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float x0 = __uint_as_float(hw[e] << 16) + __uint_as_float(lw[e] << 16);
        const float x1 = __uint_as_float(hw[e] & 0xffff0000u) + __uint_as_float(lw[e] & 0xffff0000u);
        const float m0 = drop_mul(drop, DROP_SITE_FINAL, (uint32_t)(s0 + t), (uint32_t)(col * 8 + 2 * e));
        const float m1 = drop_mul(drop, DROP_SITE_FINAL, (uint32_t)(s0 + t), (uint32_t)(col * 8 + 2 * e + 1));
        acc[i][2 * e] = fmaf(x0 * m0, r, acc[i][2 * e]);
        acc[i][2 * e + 1] = fmaf(x1 * m1, r, acc[i][2 * e + 1]);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int col = lane + 64 * i;
    if (col < nv) {
      *reinterpret_cast<float4*>(&red[wave][col * 8]) = make_float4(acc[i][0], acc[i][1], acc[i][2], acc[i][3]);
      *reinterpret_cast<float4*>(&red[wave][col * 8 + 4]) = make_float4(acc[i][4], acc[i][5], acc[i][6], acc[i][7]);
    }
  }
  __syncthreads();
  float* dst = partial + (size_t)(s0 / POOL_CHUNK + b + c) * D;
  for (int col = threadIdx.x; col < D; col += 256) dst[col] = (red[0][col] + red[1][col]) + (red[2][col] + red[3][col]);
}

// dxm = bf16(mask * dx): the operand of the dgrad / wgrad GEMMs behind a residual-branch dropout (HF:140, 400) - the branch's
// output gradient is the residual gradient times the branch's mask
__global__ __launch_bounds__(256) void mask_dx_kernel(const bf16_t* __restrict__ dxhi, const bf16_t* __restrict__ dxlo,
                                                      bf16_t* __restrict__ dxm, int rows, int D, Drop drop, uint32_t site) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const uint4* sh = reinterpret_cast<const uint4*>(dxhi + (size_t)row * D);
  const uint4* sl = reinterpret_cast<const uint4*>(dxlo + (size_t)row * D);
  uint4* dm = reinterpret_cast<uint4*>(dxm + (size_t)row * D);
  for (int c = lane; c < (D >> 3); c += 64) {
    const uint4 vh = sh[c], vl = sl[c];
    const uint32_t hw[4] = {vh.x, vh.y, vh.z, vh.w}, lw[4] = {vl.x, vl.y, vl.z, vl.w};
    uint32_t ow[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float x0 = __uint_as_float(hw[e] << 16) + __uint_as_float(lw[e] << 16);
      const float x1 = __uint_as_float(hw[e] & 0xffff0000u) + __uint_as_float(lw[e] & 0xffff0000u);
      float m0, m1;
      drop_mul2(drop, site, (uint32_t)row, (uint32_t)(c * 8 + 2 * e), m0, m1);
      ow[e] = pack_bf2(x0 * m0, x1 * m1);
    }
    dm[c] = make_uint4(ow[0], ow[1], ow[2], ow[3]);
  }
}

// test entry: out[r, c] = 1 if element (site, row0 + r, col0 + c) is kept
__global__ void dropout_mask_kernel(Drop drop, uint32_t site, uint32_t row0, uint32_t col0, int rows, int cols,
                                    uint8_t* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * cols) return;
  out[i] = drop_mul(drop, site, row0 + (uint32_t)(i / cols), col0 + (uint32_t)(i % cols)) != 0.f;
}

// ------------------------------------------------------------------------------------------
// row-wise pieces
// ------------------------------------------------------------------------------------------
// rcoef[t] = rs[t]^2 * (sum_p rdp[p][t]) / D, slots summed in index order
__global__ __launch_bounds__(64) void rowdot_finish_kernel(const float* __restrict__ rdp, int np, int ld,
                                                           const float* __restrict__ rs, float inv_d,
                                                           float* __restrict__ rcoef, int rows) {
  const int t = blockIdx.x * 64 + threadIdx.x;
  if (t >= rows) return;
  // 32 slots requested at a time, then added in index order (round 6: the plain loop waited for each of the 56 slots of the
  // FFN projection in turn - 19 us for 10 k tokens)
  float s = 0.f;
  for (int p0 = 0; p0 < np; p0 += 32) {
    float v[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) v[i] = (p0 + i < np) ? rdp[(size_t)(p0 + i) * ld + t] : 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) s += v[i];
  }
  const float r = rs[t];
  rcoef[t] = r * r * s * inv_d;
}

// qkv projection, behind the attention backward: dzs = rs * dqkv (in place), rcoef = rs^2 * sum_o dqkv_o qkv_o / D.
// One wave per token row; rows >= T are zeroed (they are K rows of the wgrad that follows).
__global__ __launch_bounds__(256) void qkv_scale_dot_kernel(bf16_t* __restrict__ dqkv, const bf16_t* __restrict__ qkv,
                                                            const float* __restrict__ rs, float* __restrict__ rcoef,
                                                            int T, int rows, int width, float inv_d) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  uint4* dp = reinterpret_cast<uint4*>(dqkv + (size_t)row * width);
  const int nv = width >> 3;
  if (row >= T) {
    for (int c = lane; c < nv; c += 64) dp[c] = make_uint4(0, 0, 0, 0);
    if (lane == 0) rcoef[row] = 0.f;
    return;
  }
  const uint4* yp = reinterpret_cast<const uint4*>(qkv + (size_t)row * width);
  const float r = rs[row];
  float dot = 0.f;
  for (int c = lane; c < nv; c += 64) {
    const uint4 d = dp[c], y = yp[c];
    const uint32_t dw[4] = {d.x, d.y, d.z, d.w}, yw[4] = {y.x, y.y, y.z, y.w};
    uint32_t ow[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float d0 = __uint_as_float(dw[e] << 16), d1 = __uint_as_float(dw[e] & 0xffff0000u);
      const float y0 = __uint_as_float(yw[e] << 16), y1 = __uint_as_float(yw[e] & 0xffff0000u);
      dot = __builtin_fmaf(d0, y0, __builtin_fmaf(d1, y1, dot));
      ow[e] = pack_bf2(d0 * r, d1 * r);
    }
    dp[c] = make_uint4(ow[0], ow[1], ow[2], ow[3]);
  }
  dot = wave_sum(dot);
  if (lane == 0) rcoef[row] = r * r * dot * inv_d;
}

// ------------------------------------------------------------------------------------------
// pooling backward:  e = m / max(|m|, 1e-12),  m = w * s / len,  s = sum_t x_t rs_t   (model.py:108-114 + final RMSNorm)
//   pool_bwd_seq: per sequence, from the forward's chunk sums: dm = (de - e (e . de)) / |m|;  ds = dm w / len;
//                 dwf[b] = dm s / len (final_layer_norm.weight's gradient, summed over b afterwards)
//   pool_bwd_tok: per token, dx_t = rs_t ds - x_t rs_t^3 (ds . x_t) / D   -> the two planes of the residual gradient
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pool_bwd_seq_kernel(const float* __restrict__ partial, const float* __restrict__ w,
                                                           const int32_t* __restrict__ cu, const float* __restrict__ de,
                                                           float* __restrict__ ds, float* __restrict__ dwf, int D) {
  __shared__ float red[2][4];
  const int b = blockIdx.x;
  const int s0 = cu[b], len = cu[b + 1] - s0;
  const int nchunk = (len + POOL_CHUNK - 1) / POOL_CHUNK;
  const float* src = partial + (size_t)(s0 / POOL_CHUNK + b) * D;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float inv_len = 1.f / (float)len;
  float sv[8], mv[8], dv[8];
  float p_nn = 0.f, p_md = 0.f;
  int cnt = 0;
  for (int col = threadIdx.x; col < D; col += 256) {
    float sum = 0.f;
    for (int c = 0; c < nchunk; ++c) sum += src[(size_t)c * D + col];
    const float m = sum * inv_len * w[col];
    const float g = de[(size_t)b * D + col];
    sv[cnt] = sum;
    mv[cnt] = m;
    dv[cnt] = g;
    ++cnt;
    p_nn += m * m;
    p_md += m * g;
  }
  p_nn = wave_sum(p_nn);
  p_md = wave_sum(p_md);
  if (lane == 0) {
    red[0][wave] = p_nn;
    red[1][wave] = p_md;
  }
  __syncthreads();
  const float nn = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
  const float md = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
  const float norm = sqrtf(nn);
  const bool clamped = !(norm > 1e-12f);  // F.normalize's eps branch: e = m / 1e-12, no norm term
  const float inv = 1.f / fmaxf(norm, 1e-12f);
  const float proj = clamped ? 0.f : md * inv * inv;  // (e . de) / |m| = (m . de) / |m|^2
  cnt = 0;
  for (int col = threadIdx.x; col < D; col += 256) {
    const float dm = (dv[cnt] - mv[cnt] * proj) * inv;
    ds[(size_t)b * D + col] = dm * w[col] * inv_len;
    dwf[(size_t)b * D + col] = dm * sv[cnt] * inv_len;
    ++cnt;
  }
}

template <int NV>
__global__ __launch_bounds__(256) void pool_bwd_tok_kernel(const bf16_t* __restrict__ xhi, const bf16_t* __restrict__ xlo,
                                                           const float* __restrict__ rs, const int4* __restrict__ pwork,
                                                           const float* __restrict__ ds, bf16_t* __restrict__ dxhi,
                                                           bf16_t* __restrict__ dxlo, int D, float inv_d, Drop drop) {
  const int4 wk = pwork[blockIdx.x];
  const int s0 = wk.x, len = wk.y, c = wk.z, b = wk.w;
  if (len == 0) return;
  const int t0 = c * POOL_CHUNK, t1 = min(len, t0 + POOL_CHUNK);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nv = D >> 3;
  float dsv[NV][8];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int col = min(lane + 64 * i, nv - 1);
    const float4 a = *reinterpret_cast<const float4*>(ds + (size_t)b * D + col * 8);
    const float4 bb = *reinterpret_cast<const float4*>(ds + (size_t)b * D + col * 8 + 4);
    dsv[i][0] = a.x; dsv[i][1] = a.y; dsv[i][2] = a.z; dsv[i][3] = a.w;
    dsv[i][4] = bb.x; dsv[i][5] = bb.y; dsv[i][6] = bb.z; dsv[i][7] = bb.w;
  }
  // two token rows per wave in flight: both rows' planes are requested before either is reduced (round 6: one row at a time
  // ran 84 us for 10 k tokens - a load round trip, a wave reduction and a store per row, in series)
  for (int tb = t0 + 2 * wave; tb < t1; tb += 8) {
    uint4 vh[2][NV], vl[2][NV];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int t = min(tb + q, t1 - 1);
      const uint4* sh = reinterpret_cast<const uint4*>(xhi + (size_t)(s0 + t) * D);
      const uint4* sl = reinterpret_cast<const uint4*>(xlo + (size_t)(s0 + t) * D);
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        vh[q][i] = sh[min(lane + 64 * i, nv - 1)];
        vl[q][i] = sl[min(lane + 64 * i, nv - 1)];
      }
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int t = tb + q;
      if (t >= t1) break;
      const size_t row = (size_t)(s0 + t) * D;
      float xv[NV][8], dsm[NV][8];  // dsm = the sequence's ds with this token's final-dropout mask applied (HF:745)
      float dot = 0.f;
#pragma unroll
      for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e)
          dsm[i][e] = drop.thresh ? dsv[i][e] * drop_mul(drop, DROP_SITE_FINAL, (uint32_t)(s0 + t), (uint32_t)(min(lane + 64 * i, nv - 1) * 8 + e))
                                  : dsv[i][e];
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const bool live = lane + 64 * i < nv;
        const uint32_t hw[4] = {vh[q][i].x, vh[q][i].y, vh[q][i].z, vh[q][i].w}, lw[4] = {vl[q][i].x, vl[q][i].y, vl[q][i].z, vl[q][i].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          xv[i][2 * e] = __uint_as_float(hw[e] << 16) + __uint_as_float(lw[e] << 16);
          xv[i][2 * e + 1] = __uint_as_float(hw[e] & 0xffff0000u) + __uint_as_float(lw[e] & 0xffff0000u);
        }
        if (live)
#pragma unroll
          for (int e = 0; e < 8; ++e) dot = __builtin_fmaf(dsm[i][e], xv[i][e], dot);
      }
      dot = wave_sum(dot);
      const float r = rs[s0 + t];
      const float k = r * r * r * dot * inv_d;
      uint4* dh = reinterpret_cast<uint4*>(dxhi + row);
      uint4* dl = reinterpret_cast<uint4*>(dxlo + row);
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        if (lane + 64 * i >= nv) continue;
        uint4 oh, ol;
        float ss = 0.f;
        auto v = [&](int e) { return __builtin_fmaf(r, dsm[i][e], -k * xv[i][e]); };
        hilo_update2(0u, 0u, v(0), v(1), oh.x, ol.x, ss);
        hilo_update2(0u, 0u, v(2), v(3), oh.y, ol.y, ss);
        hilo_update2(0u, 0u, v(4), v(5), oh.z, ol.z, ss);
        hilo_update2(0u, 0u, v(6), v(7), oh.w, ol.w, ss);
        dh[lane + 64 * i] = oh;
        dl[lane + 64 * i] = ol;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// embedding backward:  d table[v] = sum over the tokens t with ids[t] == v, in token order, of dx[t]  (no float atomics).
// grid = (vocab, ceil(D / 256)); the workgroup scans the pass's ids 256 at a time, ballots the matches and walks the set
// bits in order (uniform control flow).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void embed_bwd_kernel(const int32_t* __restrict__ ids, int T, int vocab,
                                                        const bf16_t* __restrict__ dxhi, const bf16_t* __restrict__ dxlo,
                                                        int D, float* __restrict__ dtable, Drop drop) {
  __shared__ unsigned long long masks[4];
  const int v = blockIdx.x, col = blockIdx.y * 256 + threadIdx.x;
  const int wave = threadIdx.x >> 6;
  float acc = 0.f;
  for (int base = 0; base < T; base += 256) {
    const int t = base + threadIdx.x;
    int id = (t < T) ? ids[t] : -1;
    if (id >= 0) id = min(max(id, 0), vocab - 1);  // as the forward's gather clamps
    const unsigned long long m = __ballot(id == v);
    if ((threadIdx.x & 63) == 0) masks[wave] = m;
    __syncthreads();
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      unsigned long long mm = masks[w];
      while (mm) {
        const int bit = __builtin_ctzll(mm);
        mm &= mm - 1;
        const int tok = base + w * 64 + bit;
        const size_t off = (size_t)tok * D + col;
        if (col < D) {
          const float g = bf2f(dxhi[off]) + bf2f(dxlo[off]);
          acc += drop.thresh ? g * drop_mul(drop, DROP_SITE_EMBED, (uint32_t)tok, (uint32_t)col) : g;
        }
      }
    }
    __syncthreads();
  }
  if (col < D) dtable[(size_t)v * D + col] = acc;
}

// ------------------------------------------------------------------------------------------
// weight-gradient finishing: sum the split-K partials in split order, undo the packing / RMSNorm folding
// ------------------------------------------------------------------------------------------
enum UnfoldMode { UNFOLD_PLAIN = 0, UNFOLD_QKV = 1, UNFOLD_GEGLU = 2 };
struct UnfoldArgs {
  const float* part;     // [splits][rows][C]
  size_t split_stride;
  int splits, rows, C, mode, n;  // n = rows per source matrix (QKV: H * 64)
  float* g0;             // PLAIN: [rows, C];  QKV: q;  GEGLU: wi_0
  float* g1;             // QKV: k;  GEGLU: wi_1
  float* g2;             // QKV: v
  const float* w0;       // master weights of g0 / g1 / g2 (QKV, GEGLU: for the norm-weight gradient)
  const float* w1;
  const float* w2;
  const float* ln;       // [C] RMSNorm weight folded into the packed matrix (QKV, GEGLU)
  float* dln_part;       // [rows / 32][C] per-row-block partial column sums of dW' .* W
};
// grid = (ceil(C / 256), ceil(rows / 32)), 256 threads: a 32-row x 256-column tile, wave w takes rows 8 w .. 8 w + 7,
// a lane 4 consecutive columns (16-byte accesses; every load of a row - the S partials, the master weight - is
// independent of the others).  The norm-weight gradient's row-block partial is combined over the waves in wave order.
__global__ __launch_bounds__(256) void unfold_kernel(UnfoldArgs a) {
  __shared__ float4 red[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = blockIdx.x * 256 + lane * 4;
  const bool live = c < a.C;  // C % 4 == 0
  const int r0 = blockIdx.y * 32 + wave * 8;
  const float4 lnv = (a.mode == UNFOLD_PLAIN || !live) ? make_float4(1.f, 1.f, 1.f, 1.f)
                                                       : *reinterpret_cast<const float4*>(a.ln + c);
  float4 dl = make_float4(0.f, 0.f, 0.f, 0.f);
  if (live) {
#pragma unroll 2
    for (int i = 0; i < 8; ++i) {
      const int r = r0 + i;
      if (r >= a.rows) break;
      float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int k = 0; k < a.splits; ++k) {
        const float4 v = *reinterpret_cast<const float4*>(a.part + (size_t)k * a.split_stride + (size_t)r * a.C + c);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
      }
      if (a.mode == UNFOLD_PLAIN) {
        *reinterpret_cast<float4*>(a.g0 + (size_t)r * a.C + c) = s;
      } else {
        int which, sr;
        if (a.mode == UNFOLD_QKV) {
          which = r / a.n;
          sr = r - which * a.n;
        } else {
          which = (r >> 5) & 1;
          sr = (r >> 6) * 32 + (r & 31);
        }
        float* g = which == 0 ? a.g0 : which == 1 ? a.g1 : a.g2;
        const float* w = which == 0 ? a.w0 : which == 1 ? a.w1 : a.w2;
        const size_t o = (size_t)sr * a.C + c;
        const float4 wv = *reinterpret_cast<const float4*>(w + o);
        *reinterpret_cast<float4*>(g + o) = make_float4(s.x * lnv.x, s.y * lnv.y, s.z * lnv.z, s.w * lnv.w);
        dl.x = __builtin_fmaf(s.x, wv.x, dl.x);
        dl.y = __builtin_fmaf(s.y, wv.y, dl.y);
        dl.z = __builtin_fmaf(s.z, wv.z, dl.z);
        dl.w = __builtin_fmaf(s.w, wv.w, dl.w);
      }
    }
  }
  if (a.mode == UNFOLD_PLAIN) return;
  red[wave][lane] = dl;
  __syncthreads();
  if (wave == 0 && live) {
    const float4 p0 = red[0][lane], p1 = red[1][lane], p2 = red[2][lane], p3 = red[3][lane];
    *reinterpret_cast<float4*>(a.dln_part + (size_t)blockIdx.y * a.C + c) =
        make_float4((p0.x + p1.x) + (p2.x + p3.x), (p0.y + p1.y) + (p2.y + p3.y), (p0.z + p1.z) + (p2.z + p3.z),
                    (p0.w + p1.w) + (p2.w + p3.w));
  }
}

// out[c] = sum_r part[r][c]: 64 columns per workgroup of 16 waves, wave w sums rows w, w + 16, ... in order (eight loads in
// flight), then the sixteen wave sums pairwise in a fixed tree (deterministic).  Round 6: four waves per workgroup walked 56
// rows each one load at a time - 9.5 us per launch, 25 launches per step.
constexpr int COLSUM_WAVES = 16;
__global__ __launch_bounds__(64 * COLSUM_WAVES) void colsum_kernel(const float* __restrict__ part, int rows, int C,
                                                                   float* __restrict__ out) {
  __shared__ float red[COLSUM_WAVES][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  float s = 0.f;
  if (c < C) {
    for (int r0 = wave; r0 < rows; r0 += 8 * COLSUM_WAVES) {
      float v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int r = r0 + i * COLSUM_WAVES;
        v[i] = (r < rows) ? part[(size_t)r * C + c] : 0.f;
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) s += v[i];
    }
  }
  red[wave][lane] = s;
  __syncthreads();
  if (wave == 0 && c < C) {
    float t[COLSUM_WAVES];
#pragma unroll
    for (int i = 0; i < COLSUM_WAVES; ++i) t[i] = red[i][lane];
#pragma unroll
    for (int w = 1; w < COLSUM_WAVES; w *= 2)
#pragma unroll
      for (int i = 0; i < COLSUM_WAVES; i += 2 * w) t[i] += t[i + w];
    out[c] = t[0];
  }
}

// relative_attention_bias.weight's gradient: column sums of the workgroups' table rows, then offsets -> buckets
// (HF:216-262 through `bucket_of`, computed on the host once).  One workgroup per head.
__global__ __launch_bounds__(256) void bias_grad_kernel(const float* __restrict__ dtab_part, int nrows, int H, int ntab,
                                                        const int32_t* __restrict__ bucket_of, int nbuckets,
                                                        float* __restrict__ d_rel_bias /* [buckets, H] */) {
  __shared__ float tabsum[ATT_TAB_MAX];
  const int h = blockIdx.x;
  for (int i = threadIdx.x; i < ntab; i += 256) {
    float s = 0.f;
    for (int r0 = 0; r0 < nrows; r0 += 16) {  // sixteen rows requested, then added in row order (one at a time: 70 us per step)
      float v[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) v[k] = (r0 + k < nrows) ? dtab_part[((size_t)(r0 + k) * H + h) * ntab + i] : 0.f;
#pragma unroll
      for (int k = 0; k < 16; ++k) s += v[k];
    }
    tabsum[i] = s;
  }
  __syncthreads();
  for (int bk = threadIdx.x; bk < nbuckets; bk += 256) {
    float s = 0.f;
    for (int i = 0; i < ntab; ++i)
      if (bucket_of[i] == bk) s += tabsum[i];
    d_rel_bias[(size_t)bk * H + h] = s;
  }
}

// bias table [H, ntab] from relative_attention_bias.weight [buckets, H] (fp32 master), on the device
__global__ void bias_table_kernel(const float* __restrict__ rel_bias, const int32_t* __restrict__ bucket_of, int H,
                                  int ntab, float* __restrict__ tab) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= H * ntab) return;
  const int h = i / ntab, o = i - h * ntab;
  tab[i] = rel_bias[(size_t)bucket_of[o] * H + h];
}

// bf16 [R, C] -> [C, R] (weight copies for the dgrad GEMMs, once per optimizer step)
__global__ __launch_bounds__(256) void transpose_bf16_kernel(const bf16_t* __restrict__ in, int R, int C,
                                                             bf16_t* __restrict__ out) {
  __shared__ bf16_t tile[64][66];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    const int r = i >> 6, c = i & 63;
    tile[r][c] = (r0 + r < R && c0 + c < C) ? in[(size_t)(r0 + r) * C + c0 + c] : (bf16_t)0;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    const int c = i >> 6, r = i & 63;
    if (c0 + c < C && r0 + r < R) out[(size_t)(c0 + c) * R + r0 + r] = tile[r][c];
  }
}

// Refresh of one layer's bf16 compute copies from the fp32 masters, ONE launch: for each of the four matrices the packed
// form the forward reads (fused q|k|v rows / 32-row gate|up interleave / plain; RMSNorm weight folded in column-wise) AND its
// transpose for the dgrad GEMMs, from a single read of the masters (64 x 64 tiles through LDS).  Every dimension is a
// multiple of 64 (d_model, d_ff, H * 64).
struct RepackMat {
  const float* s0;
  const float* s1;
  const float* s2;
  const float* colscale;  // [cols] or NULL
  bf16_t* dst;            // [rows, cols]
  bf16_t* dst_t;          // [cols, rows]
  int rows, cols, n, mode, tile0;
};
struct RepackArgs {
  RepackMat m[4];
};
__global__ __launch_bounds__(256) void repack_layer_kernel(RepackArgs a) {
  __shared__ bf16_t tile[64][68];
  int mi = 3;
  while (mi > 0 && (int)blockIdx.x < a.m[mi].tile0) --mi;
  const RepackMat& M = a.m[mi];
  const int t = blockIdx.x - M.tile0;
  const int tiles_c = M.cols >> 6;
  const int r0 = (t / tiles_c) * 64, c0 = (t % tiles_c) * 64;
  const int q = threadIdx.x & 15, rr = threadIdx.x >> 4;  // 16 threads x float4 = one 64-column row piece
  float4 cs = make_float4(1.f, 1.f, 1.f, 1.f);
  if (M.colscale) cs = *reinterpret_cast<const float4*>(M.colscale + c0 + 4 * q);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int r = r0 + rr + 16 * k;
    const float* src;
    int sr;
    if (M.mode == PACK_CONCAT3) {
      src = (r < M.n) ? M.s0 : (r < 2 * M.n ? M.s1 : M.s2);
      sr = r % M.n;
    } else if (M.mode == PACK_GEGLU) {
      src = ((r >> 5) & 1) ? M.s1 : M.s0;
      sr = (r >> 6) * 32 + (r & 31);
    } else {
      src = M.s0;
      sr = r;
    }
    const float4 v = *reinterpret_cast<const float4*>(src + (size_t)sr * M.cols + c0 + 4 * q);
    const uint2 p = make_uint2(pack_bf2(v.x * cs.x, v.y * cs.y), pack_bf2(v.z * cs.z, v.w * cs.w));
    *reinterpret_cast<uint2*>(M.dst + (size_t)r * M.cols + c0 + 4 * q) = p;
    *reinterpret_cast<uint2*>(&tile[rr + 16 * k][4 * q]) = p;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int c = rr + 16 * k;  // a column of the tile = a row of the transpose; this thread's four consecutive rows 4 q ..
    const uint32_t lo = (uint32_t)tile[4 * q][c] | ((uint32_t)tile[4 * q + 1][c] << 16);
    const uint32_t hi = (uint32_t)tile[4 * q + 2][c] | ((uint32_t)tile[4 * q + 3][c] << 16);
    *reinterpret_cast<uint2*>(M.dst_t + (size_t)(c0 + c) * M.rows + r0 + 4 * q) = make_uint2(lo, hi);
  }
}

// sum of squares of n floats, deterministic: fixed grid of partials, then one workgroup
__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* __restrict__ g, size_t n, float* __restrict__ part) {
  __shared__ float red[4];
  float s = 0.f;
  const size_t n4 = n >> 2;  // 16-byte pieces, four in flight per thread
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += 4 * stride) {
    float4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = reinterpret_cast<const float4*>(g)[min(i + u * stride, n4 - 1)];
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (i + u * stride < n4)
        s = __builtin_fmaf(v[u].x, v[u].x, __builtin_fmaf(v[u].y, v[u].y, __builtin_fmaf(v[u].z, v[u].z, __builtin_fmaf(v[u].w, v[u].w, s))));
  }
  if (blockIdx.x == 0 && threadIdx.x < (unsigned)(n - n4 * 4)) s = __builtin_fmaf(g[n4 * 4 + threadIdx.x], g[n4 * 4 + threadIdx.x], s);
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ __launch_bounds__(256) void sumsq_finish_kernel(const float* __restrict__ part, int n, float* __restrict__ out_norm) {
  __shared__ float red[256];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s += part[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out_norm[0] = sqrtf(red[0]);
}

}  // namespace rp
