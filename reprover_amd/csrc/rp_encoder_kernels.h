// Device code of the T5 encoder shared by the inference pass (rp_encoder.hip) and the training step (rp_train.hip):
// weight packing, the two-plane residual stream, embedding, GEMM epilogues + launchers, attention, pooling.
#pragma once
#include <math.h>
#include <string.h>

#include <algorithm>
#include <functional>
#include <vector>

#include "rp_gemm.h"

namespace rp {

// tuning knobs (defined in rp_encoder.hip, set through rp_set_option)
extern int g_gemm_group_m, g_gemm_variant, g_gemm_variant_qkv, g_gemm_variant_wo, g_gemm_variant_o, g_gemm_tail_split,
    g_debug_skip_ffn, g_gemm_skinny, g_gemm_skinny_variant, g_gemm_rs_lds, g_gemm_small_pipe, g_gemm_helpers, g_gemm_persist, g_pool_chunk, g_gemm_edge_layout, g_gemm_tail_variant, g_gemm_mixed, g_gemm_mixed_bwd;
extern int g_gemm_stagger_us[RP_K_COUNT];

// ------------------------------------------------------------------------------------------
// weight packing (create time only)
// ------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ float load_as_f32(const void* p, size_t i);
template <>
__device__ __forceinline__ float load_as_f32<float>(const void* p, size_t i) {
  return reinterpret_cast<const float*>(p)[i];
}
template <>
__device__ __forceinline__ float load_as_f32<bf16_t>(const void* p, size_t i) {
  return bf2f(reinterpret_cast<const bf16_t*>(p)[i]);
}

enum PackMode { PACK_CONCAT3 = 0, PACK_GEGLU = 1, PACK_COPY = 2 };

// dst bf16 [rows_dst, cols]; source row mapping by mode:
//   CONCAT3: rows [0,n) from s0, [n,2n) from s1, [2n,3n) from s2      (fused q|k|v)
//   GEGLU  : 64-row blocks: 32 rows of s0 (gate, wi_0) then 32 rows of s1 (up, wi_1)
//   COPY   : row r from s0
//   colscale (optional, fp32 [cols]): every row is multiplied column-wise before rounding — used to
//   fold the T5 RMSNorm weight into the projection that consumes the normalised activations.
template <typename T>
__global__ void pack_rows_kernel(bf16_t* dst, const void* s0, const void* s1, const void* s2,
                                 int rows_dst, int cols, int n, int mode, const float* colscale) {
  const int r = blockIdx.x;
  const void* src;
  int sr;
  if (mode == PACK_CONCAT3) {
    src = (r < n) ? s0 : (r < 2 * n ? s1 : s2);
    sr = r % n;
  } else if (mode == PACK_GEGLU) {
    src = ((r >> 5) & 1) ? s1 : s0;
    sr = (r >> 6) * 32 + (r & 31);
  } else {
    src = s0;
    sr = r;
  }
  for (int c = threadIdx.x; c < cols; c += blockDim.x) {
    float v = load_as_f32<T>(src, (size_t)sr * cols + c);
    if (colscale) v *= colscale[c];
    dst[(size_t)r * cols + c] = f2bf(v);
  }
}

template <typename T>
__global__ void to_f32_kernel(float* dst, const void* src, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = load_as_f32<T>(src, i);
}

// Dropout of the TRAINING step (rp_train.hip; inference never drops): counter-based, so the backward regenerates the mask
// the forward used from (seed, site, row, column) - nothing is stored.  One 32-bit hash serves the column pair (2 c, 2 c + 1):
// its low / high 16 bits against a 16-bit threshold (round 6: a hash per element was 12 integer operations, three of them
// quarter-rate multiplies, in front of every dropped value - 1.1 ms of a 21.7-ms step; a pair costs 17).
// keep iff field >= thresh (thresh = round(p * 2^16): p = 0.1 drops 6554 / 65536 = 0.100006; 0 = off), kept values are
// scaled by 1 / (1 - p)  (torch.nn.Dropout).
struct Drop {
  uint32_t seed, thresh;
  float scale;
};
// murmur3's 32-bit finaliser over a counter that is LINEAR in (row, column pair): walking a lane's elements along either
// index is one add in front of it (the attention backward walks queries in one launch and keys in the other).
__device__ __forceinline__ uint32_t drop_hash(uint32_t seed, uint32_t site, uint32_t row, uint32_t col2) {
  uint32_t h = (seed ^ (site * 0x9E3779B1u)) + row * 0x85EBCA77u + col2 * 0x27D4EB2Fu;
  h ^= h >> 16;
  h *= 0x85EBCA6Bu;
  h ^= h >> 13;
  h *= 0xC2B2AE35u;
  h ^= h >> 16;
  return h;
}
// multipliers of columns col_even and col_even + 1 (col_even must be even)
__device__ __forceinline__ void drop_mul2(const Drop& d, uint32_t site, uint32_t row, uint32_t col_even, float& m0, float& m1) {
  const uint32_t h = drop_hash(d.seed, site, row, col_even >> 1);
  m0 = (h & 0xffffu) >= d.thresh ? d.scale : 0.f;
  m1 = (h >> 16) >= d.thresh ? d.scale : 0.f;
}
__device__ __forceinline__ float drop_mul(const Drop& d, uint32_t site, uint32_t row, uint32_t col) {
  const uint32_t h = drop_hash(d.seed, site, row, col >> 1);
  return ((col & 1u) ? (h >> 16) : (h & 0xffffu)) >= d.thresh ? d.scale : 0.f;
}
// element ids: (site, row = packed token index, col = feature) - attention probabilities: row = the query's packed token
// index, col = (head << 20) | key offset inside the sequence (sequences of up to 2^20 tokens: far beyond any max_seq_len)
enum { DROP_SITE_EMBED = 0, DROP_SITE_FINAL = 1, DROP_SITE_LAYER0 = 16 };  // layer i: 16 + 8 i + {0 probs, 1 attn residual, 2 FFN inner, 3 FFN residual}

// ------------------------------------------------------------------------------------------
// The residual stream x lives in HBM as TWO bf16 planes: hi = bf16(x) and lo = bf16(x - hi), x = hi + lo to 2^-18
// relative (16 mantissa bits).  hi IS the A operand of the next projection, so a residual update reads 4 B and writes
// 4 B per element where an fp32 stream with a separate bf16 copy read 4 and wrote 6: the two residual epilogues and the
// embedding move 20 % fewer bytes, and the chip read-modify-writes the two planes 29 % faster than the fp32 + bf16 form
// (tools/probes/rmw_probe.hip: 0.168 vs 0.238 ms for [70144, 1472] in 256 x 256 tiles).  The GEMM operands are
// bf16-rounded either way; the 2^-18 bound per update is far below that rounding (2^-9).
// ------------------------------------------------------------------------------------------
// one updated element pair: (hi + lo) + d, re-split; ss accumulates the squares of the values as stored
__device__ __forceinline__ void hilo_update2(uint32_t h2, uint32_t l2, float d0, float d1, uint32_t& oh, uint32_t& ol,
                                             float& ss) {
  const float v0 = (__uint_as_float(h2 << 16) + __uint_as_float(l2 << 16)) + d0;
  const float v1 = (__uint_as_float(h2 & 0xffff0000u) + __uint_as_float(l2 & 0xffff0000u)) + d1;
  oh = pack_bf2(v0, v1);
  const float h0 = __uint_as_float(oh << 16), h1 = __uint_as_float(oh & 0xffff0000u);
  ol = pack_bf2(v0 - h0, v1 - h1);
  const float x0 = h0 + __uint_as_float(ol << 16), x1 = h1 + __uint_as_float(ol & 0xffff0000u);
  // explicit fma chain: the same rounding sequence for every token, wherever it sits in the batch
  ss = __fmaf_rn(x1, x1, __fmaf_rn(x0, x0, ss));
}

// Round 5 - the INFERENCE pass keeps x as a 24-bit word per element instead: hi = the bf16 plane (still the next
// projection's A operand) and an int8 "extension" plane e, x = float((hi << 16) + (e << 8)) - the fp32 word of x rounded to
// its top 24 bits (16 significant bits, as the two bf16 planes gave), split so that hi is that word rounded to its top 16
// bits (half away from zero: differs from nearest-even on exact ties only, 1 value in 256) and e the signed remainder in
// units of 2^-8 ulp(hi): e is simply bits 8..15 of the rounded word read as int8.  A residual update reads 3 B and writes 3 B per element where the two bf16 planes move 4 + 4: the
// read-modify-write of x bounds the attention-out projection and the FFN-out epilogue (tools/probes/rmw_probe.hip:
// 0.144 -> 0.122 ms for [70144, 1472] in 256 x 256 tiles; in the step -0.5 ms).  The training step keeps the bf16 pair.
// Round 6 - the extension byte is stored BIASED: u = e + 128 = bits 8..15 of (the rounded word + 0x8000).  The same values, hi
// plane and rounding as before, and fewer instructions in the two residual epilogues, which are bound by their own
// instruction stream (profiles/r06_raw/exp4): decoding is ONE byte permute + ONE subtract per element -
// x = float(((hi << 16) | (u << 8)) - 0x8000) - where the signed byte took a sign-extending extract, a shift and an add; and
// encoding shares one add between the two planes: q = bits(v) + 0x8080 holds hi in its top half and u in byte 1.
// A zero is (hi 0, u 0x80): X24_ZERO_EXT.
constexpr uint32_t X24_ZERO_EXT = 0x80808080u;
// One updated element pair (bf16 pair h2; extension bytes K and K + 1 of word lw): decode, add, round, re-split; q0 / q1 carry
// the new extension bytes in their byte 1 (x24_pack_ext); ss accumulates the squares of the values as stored.
template <int K>
__device__ __forceinline__ void x24_update2(uint32_t h2, uint32_t lw, float d0, float d1, uint32_t& oh, uint32_t& q0,
                                            uint32_t& q1, float& ss) {
  // v_perm_b32: bytes 7..4 = h2, 3..0 = lw; selector 0x0c = the constant 0
  const uint32_t w0 = __builtin_amdgcn_perm(h2, lw, 0x05040000u | ((uint32_t)K << 8) | 0x0cu) - 0x8000u;
  const uint32_t w1 = __builtin_amdgcn_perm(h2, lw, 0x07060000u | ((uint32_t)(K + 1) << 8) | 0x0cu) - 0x8000u;
  const float v0 = __uint_as_float(w0) + d0, v1 = __uint_as_float(w1) + d1;
  const uint32_t r0 = __float_as_uint(v0) + 0x80u, r1 = __float_as_uint(v1) + 0x80u;  // to 24 bits, half away from zero
  q0 = r0 + 0x8000u;  // top half: the 24-bit word to 16 bits, half away from zero; byte 1: the biased extension byte
  q1 = r1 + 0x8000u;
  oh = __builtin_amdgcn_perm(q1, q0, 0x07060302u);
  const float x0 = __uint_as_float(r0 & 0xffffff00u), x1 = __uint_as_float(r1 & 0xffffff00u);
  ss = __fmaf_rn(x1, x1, __fmaf_rn(x0, x0, ss));  // explicit fma chain: the same rounding sequence for every token
}
// byte 1 of four words -> one word (element i = byte i)
__device__ __forceinline__ uint32_t x24_pack_ext(uint32_t q0, uint32_t q1, uint32_t q2, uint32_t q3) {
  const uint32_t lo = __builtin_amdgcn_perm(q1, q0, 0x0c0c0501u), hi = __builtin_amdgcn_perm(q3, q2, 0x0c0c0501u);
  return __builtin_amdgcn_perm(hi, lo, 0x05040100u);
}
// (hi by the hardware's nearest-even conversion + an explicit remainder was measured too: the same margins, and 6 % more
// time in the two residual epilogues - they are VALU-sensitive: two waves per SIMD, 128 elements per lane and tile.)
// eight elements: h = 8 bf16, l = 8 biased extension bytes (element i = byte i), d[0..7] added
__device__ __forceinline__ void x24_update8(const uint4 h, const uint2 l, const float4 d0, const float4 d1, uint4& oh, uint2& ol,
                                            float& ss) {
  uint32_t q[8];
  x24_update2<0>(h.x, l.x, d0.x, d0.y, oh.x, q[0], q[1], ss);
  x24_update2<2>(h.y, l.x, d0.z, d0.w, oh.y, q[2], q[3], ss);
  x24_update2<0>(h.z, l.y, d1.x, d1.y, oh.z, q[4], q[5], ss);
  x24_update2<2>(h.w, l.y, d1.z, d1.w, oh.w, q[6], q[7], ss);
  ol.x = x24_pack_ext(q[0], q[1], q[2], q[3]);
  ol.y = x24_pack_ext(q[4], q[5], q[6], q[7]);
}
// decode only (the pooling pass): elements 2 j, 2 j + 1 of an 8-element group (lw = the group's word holding their bytes)
__device__ __forceinline__ void x24_decode2(uint32_t h2, uint32_t lw, int j, float& x0, float& x1) {
  const uint32_t s0 = j ? 0x0504020cu : 0x0504000cu, s1 = j ? 0x0706030cu : 0x0706010cu;
  x0 = __uint_as_float(__builtin_amdgcn_perm(h2, lw, s0) - 0x8000u);
  x1 = __uint_as_float(__builtin_amdgcn_perm(h2, lw, s1) - 0x8000u);
}

// ------------------------------------------------------------------------------------------
// K1: byte-token embedding gather  x[t] = embed[ids[t]]   (HF:678); the table (vocab x D fp32,
//   2.3 MB for ByT5-small) is L2-resident.  Emits the two planes of x (hi = the A operand of the first
//   projection) and the row's sum of squares (RMSNorm statistic, applied in that GEMM's epilogue).
//   rows >= T (tile padding) get token 0 so every workspace row stays finite.
// ------------------------------------------------------------------------------------------
template <bool LO8>  // LO8: the extension plane of the 24-bit form (x24_update2) instead of the bf16 lo plane
__global__ __launch_bounds__(256) void embed_kernel(const int32_t* __restrict__ ids,
                                                    const float* __restrict__ table,
                                                    bf16_t* __restrict__ xhi, bf16_t* __restrict__ xlo,
                                                    float* __restrict__ ssp, int np, int T, int Tp, int D,
                                                    int vocab, const int32_t* __restrict__ t_dev) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  int rows = Tp;  // Tp stays the leading dimension of the slot-major statistics
  if (t_dev) {    // token count known on the device only (rp_encode_padded): T / Tp are upper bounds
    T = *t_dev;
    rows = min(Tp, (T + 255) & ~255);
  }
  if (row >= rows) return;
  int id = (row < T) ? (ids ? ids[row] : row) : 0;  // ids NULL: row r is token r (embed_table_x24: the table re-encoded)
  id = min(max(id, 0), vocab - 1);
  const float4* src = reinterpret_cast<const float4*>(table + (size_t)id * D);
  uint4* dh = reinterpret_cast<uint4*>(xhi + (size_t)row * D);
  uint4* dl = reinterpret_cast<uint4*>(xlo + (size_t)row * D);
  uint2* dl8 = reinterpret_cast<uint2*>(reinterpret_cast<uint8_t*>(xlo) + (size_t)row * D);
  float ss = 0.f;
  for (int c = lane; c < (D >> 3); c += 64) {  // 8 features per lane and step
    const float4 a = src[2 * c], b = src[2 * c + 1];
    if constexpr (LO8) {
      uint4 oh;
      uint2 ol;
      x24_update8(make_uint4(0u, 0u, 0u, 0u), make_uint2(X24_ZERO_EXT, X24_ZERO_EXT), a, b, oh, ol, ss);
      dh[c] = oh;
      dl8[c] = ol;
    } else {
      uint4 oh, ol;
      hilo_update2(0u, 0u, a.x, a.y, oh.x, ol.x, ss);
      hilo_update2(0u, 0u, a.z, a.w, oh.y, ol.y, ss);
      hilo_update2(0u, 0u, b.x, b.y, oh.z, ol.z, ss);
      hilo_update2(0u, 0u, b.z, b.w, oh.w, ol.w, ss);
      dh[c] = oh;  // (non-temporal stores measured in round 5: 83 us either way)
      dl[c] = ol;
    }
  }
  ss = wave_sum(ss);
  // sum-of-squares partials of the row (see EpiResid): slot 0 carries the whole row here
  for (int p = lane; p < np; p += 64) ssp[(size_t)p * Tp + row] = (p == 0) ? ss : 0.f;
}

constexpr int RMS_MAX_V4 = 8;

// The inference pass's embedding: the table is kept PRE-ENCODED in the 24-bit form (embed_table_x24: embed_kernel<true> run
// once over the vocabulary - at create time and whenever the trainer refreshes the fp32 table - so a token's planes and its
// sum of squares are the bits embed_kernel<true> computes per token), and a pass copies rows: one wave per token row,
// 16 B of the bf16 plane + 8 B of the int8 plane per lane and step, no arithmetic.  Round 5: the access pattern alone
// stores the 70 k x 1472 stream in 45 us (tools/probes/stream_probe.hip); embed_kernel<true> took 63-67 (it re-reads the
// fp32 table row - 413 MB through L2 per pass - encodes, and scatters 23 four-byte statistic slots per row).
// rs_out given: the row's RMSNorm factor is written directly (rowscale_kernel's arithmetic on slot 0 + zeros: the same
// bits) and the pass skips the first rowscale launch; else the statistic goes to the slots as embed_kernel writes it.
// NV = 16-byte pieces per lane covering a row of the bf16 plane (ceil(D / 512): 3 for d_model 1472 / 1536, 4 up to 2048);
// R = token rows per wave, all their loads requested before the first store (round 6: the 4-step form issued a fourth,
// clamped load of every plane for d_model 1472 - a quarter of the load instructions for nothing - and kept one row per wave
// in flight).
template <int NV, int R>
static __global__ __launch_bounds__(256) void embed_copy_kernel(const int32_t* __restrict__ ids, const bf16_t* __restrict__ thi,
                                                         const uint8_t* __restrict__ tlo, const float* __restrict__ tss,
                                                         bf16_t* __restrict__ xhi, uint8_t* __restrict__ xlo,
                                                         float* __restrict__ ssp, int np, float* __restrict__ rs_out,
                                                         float inv_d, float eps, int T, int Tp, int D, int vocab,
                                                         const int32_t* __restrict__ t_dev) {
  const int lane = threadIdx.x & 63;
  const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * R;
  int rows = Tp;
  if (t_dev) {  // token count known on the device only (rp_encode_padded): T / Tp are upper bounds
    T = *t_dev;
    rows = min(Tp, (T + 255) & ~255);
  }
  if (row0 >= rows) return;
  const int nv = D >> 3;
  uint4 vh[R][NV];
  uint2 vl[R][NV];
  int idv[R];
#pragma unroll
  for (int u = 0; u < R; ++u) {
    const int row = min(row0 + u, rows - 1);
    int id = (row < T) ? ids[row] : 0;
    idv[u] = id = min(max(id, 0), vocab - 1);
    const uint4* sh = reinterpret_cast<const uint4*>(thi + (size_t)id * D);
    const uint2* sl = reinterpret_cast<const uint2*>(tlo + (size_t)id * D);
#pragma unroll
    for (int i = 0; i < NV; ++i) {  // every load of the wave's rows requested before the first store
      const int c = min(lane + 64 * i, nv - 1);
      vh[u][i] = sh[c];
      vl[u][i] = sl[c];
    }
  }
#pragma unroll
  for (int u = 0; u < R; ++u) {
    const int row = row0 + u;
    if (row >= rows) break;
    uint4* dh = reinterpret_cast<uint4*>(xhi + (size_t)row * D);
    uint2* dl = reinterpret_cast<uint2*>(xlo + (size_t)row * D);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = lane + 64 * i;
      if (c < nv) {
        dh[c] = vh[u][i];
        dl[c] = vl[u][i];
      }
    }
    const float ss = tss[idv[u]];
    if (rs_out) {
      if (lane == 0) rs_out[row] = rsqrtf(ss * inv_d + eps);
    } else {
      for (int p = lane; p < np; p += 64) ssp[(size_t)p * Tp + row] = (p == 0) ? ss : 0.f;
    }
  }
}

  // a row is at most 8 float4 per lane of a wave: d_model <= 2048

// ------------------------------------------------------------------------------------------
// K3/K6/K7/K8: GEMM epilogues
// ------------------------------------------------------------------------------------------
// The encoder GEMMs are issued "transposed": the weight matrix is the MFMA row operand (rows of the
// accumulator tile = output features) and the activations the column operand (cols = tokens).  With
// the 32x32 C/D layout a lane then owns ONE token (col = lane & 31) and, per 4-register group, FOUR
// CONSECUTIVE output features (row = 8g + 4hi + 0..3) — so every epilogue moves 8 B (bf16x4) or 16 B
// (fp32x4) per lane per access instead of 2-4 B, a quarter of the store instructions.
//   acc[i][j][4g + e]  <->  feature m_base + 32 i + 8 g + 4 hi + e,  token n_base + 32 j + (lane & 31)
// Each epilogue goes through the wave's private LDS staging area (EPI_STAGE_BYTES, rows of
// EPI_ROW_BYTES): the accumulators are written token-row-major with ds_write_b128 / b64 (4
// consecutive features per lane), then read back so that 8 or 16 consecutive lanes cover one token's
// contiguous 128-B (fp32 x 32) or 64/128-B (bf16) span: global accesses become full cache lines,
// 16 B per lane, instead of 8-16 B scattered over 32 rows.
// T5 RMSNorm folded into the GEMMs (HF:59-72).  h = w * x * rsqrt(mean(x^2) + eps) feeds only the QKV
// and FFN-in projections, so:  (a) w is folded into those weights when they are packed;  (b) the
// producer of x (embedding kernel / residual-add epilogue) also stores xb = bf16(x), the GEMM A operand,
// and per-row partial sums of squares ssp[token][p], one slot per 64-feature wave tile (deterministic:
// no atomics);  (c) the consuming epilogue multiplies each accumulator by rs[token] =
// rsqrt(sum_p ssp[token][p] / D + eps), reduced once per sub-layer by the tiny rowscale_kernel.  No separate
// normalisation pass over x remains.
struct RowScale {
  const float* rs;  // [tokens] or NULL (no scaling); produced by rowscale_kernel from the ssp partials
  __device__ __forceinline__ float get(int token) const { return rs ? rs[token] : 1.f; }
};
// Passes of at most ~1000 tokens (a single proof state) skip the rowscale launches - there a launch costs more than
// its work - and the consuming epilogue sums the slots itself, in the same index order (the same bits).  A separate
// type, instantiated for the small tile configurations only: the big tiles' register allocation stays as it was.
struct RowScaleFromSlots {
  const float* ssp;  // [np, ld] slot-major partial sums of squares
  int np, ld;
  float inv_d, eps;
  __device__ __forceinline__ float get(int token) const {
    float v[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) v[i] = (i < np) ? ssp[(size_t)i * ld + token] : 0.f;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) s += v[i];
    return rsqrtf(s * inv_d + eps);
  }
};

// The big tiles (256 tokens per workgroup, gemm_tile_pipe) reduce the statistic in the consuming GEMM as well, without a
// register cost in the main loop: the tile's slot rows - np x 256 floats, one 1-KiB LDS-DMA piece per slot - ride into
// LDS behind the operand ring before the first operand DMA (the epilogue's prologue hook: in-order vmcnt has them
// landed long before the epilogue), and the epilogue sums a token's np slots from LDS in index order: the bits
// rowscale_kernel produces, 24 launches per pass fewer (6.7 us each + their boundaries at 70 k tokens).
struct RowScaleLds {
  static constexpr int EXTRA_LDS = 32 * 1024;  // np <= 32 slot rows of 256 floats
  const float* ssp;  // [np, ld] slot-major partial sums of squares
  int np, ld;
  float inv_d, eps;
  float* lds = nullptr;  // set by prologue()
  int n0 = 0;
  __device__ __forceinline__ void prologue(char* extra, int wave, int lane, int tok0) {
    lds = reinterpret_cast<float*>(extra);
    n0 = tok0;
    const int nw = (int)blockDim.x >> 6;
    for (int p = wave; p < np; p += nw)
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(ssp + (size_t)p * ld + tok0 + lane * 4), (lds_ptr_t)(extra + p * 1024), 16,
                                       0, 0);
  }
  // After the main loop (the rows have landed, every wave is past the barrier): ONE thread per token sums its np slots in
  // index order - rowscale_kernel's chain, whose zero padding beyond np adds exactly nothing - and leaves rs in row 0.
  // (Round 5: until then every lane summed the slots of its own four tokens in the epilogue, 8 x redundantly across the
  // workgroup: 92 LDS reads + adds per lane, 0.5 ms per step.)  The caller puts a barrier behind it.
  __device__ __forceinline__ void reduce(int tid) {
    if (tid < 256) {
      float* p = lds + tid;
      float s = 0.f;
#pragma unroll 4
      for (int i = 0; i < np; ++i) s += p[i * 256];
      p[0] = rsqrtf(s * inv_d + eps);
    }
  }
  __device__ __forceinline__ float get(int token) const { return lds[token - n0]; }
};
// an epilogue whose row scale has a prologue of its own exposes it to gemm_tile_pipe
template <class Base>
struct WithRsPrologue : Base {
  static constexpr int EXTRA_LDS = decltype(Base::rs)::EXTRA_LDS;
  __device__ __forceinline__ void prologue(char* extra, int wave, int lane, int n0) { this->rs.prologue(extra, wave, lane, n0); }
  __device__ __forceinline__ void reduce(int tid) { this->rs.reduce(tid); }
};
template <class E, class = void>
struct epi_extra_lds : std::integral_constant<int, 0> {};
template <class E>
struct epi_extra_lds<E, std::void_t<decltype(E::EXTRA_LDS)>> : std::integral_constant<int, E::EXTRA_LDS> {};

// rs[token] = rsqrt(sum_p ssp[p][token] / D + eps), slots summed in index order
static __global__ __launch_bounds__(64) void rowscale_kernel(const float* __restrict__ ssp, float* __restrict__ rs, int rows,
                                                      int np, float inv_d, float eps) {
  const int t = blockIdx.x * 64 + threadIdx.x;
  if (t >= rows) return;
  float v[32];  // np <= 32 (D <= 2048): all slots requested before the first add
#pragma unroll
  for (int i = 0; i < 32; ++i) v[i] = (i < np) ? ssp[(size_t)i * rows + t] : 0.f;  // slot-major: coalesced over tokens
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 32; ++i) s += v[i];  // index order; the zero padding adds exactly nothing
  rs[t] = rsqrtf(s * inv_d + eps);
}

template <class RS>
struct EpiStoreBf16T {  // out[token, feature] = bf16(acc * rs[token])
  static constexpr bool any_layout = true;
  bf16_t* out;
  int ldo, n_valid;  // n_valid = number of real output features (multiple of 8)
  RS rs;
  // (the preload hook of the gated-GELU epilogue below was measured here too - the persistent QKV launch: 2.874 vs 2.880 ms
  // per step, nothing - and is not carried)
  template <int FM, int FN>
  __device__ __forceinline__ void run(f32x16 (&acc)[FM][FN], int m_base, int n_base, int lane, char* stage) {
    const int hi = lane >> 5, cl = lane & 31;
    static_assert(FM % 2 == 0, "staging rows hold 64 features (two row fragments)");
    // 64 token rows x 64 features at a time: staging rows of 128 B
    constexpr int LPR = 8;              // lanes per token row (16 B each)
    constexpr int RPI = 64 / LPR;       // token rows per pass
    const int sub = lane % LPR, rr = lane / LPR;
    float scv[FN];
#pragma unroll
    for (int j = 0; j < FN; ++j) scv[j] = rs.get(n_base + j * 32 + cl);
#pragma unroll
    for (int ih = 0; ih < FM; ih += 2) {
      const int f = m_base + ih * 32 + sub * 8;
#pragma unroll
      for (int jb = 0; jb < FN; jb += 2) {
#pragma unroll
        for (int jj = 0; jj < 2 && jb + jj < FN; ++jj) {
          const float sc = scv[jb + jj];
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              uint2 v;
              v.x = pack_bf2(acc[ih + i][jb + jj][4 * g] * sc, acc[ih + i][jb + jj][4 * g + 1] * sc);
              v.y = pack_bf2(acc[ih + i][jb + jj][4 * g + 2] * sc, acc[ih + i][jb + jj][4 * g + 3] * sc);
              *reinterpret_cast<uint2*>(stage + (jj * 32 + cl) * EPI_ROW_BYTES + (i * 32 + 8 * g + 4 * hi) * 2) = v;
            }
        }
        const int nrows = (FN - jb >= 2) ? 64 : 32;
#pragma unroll
        for (int t0 = 0; t0 < 64; t0 += RPI) {
          const int t = t0 + rr;
          if (t < nrows) {
            const uint4 v = *reinterpret_cast<const uint4*>(stage + t * EPI_ROW_BYTES + sub * 16);
            if (f < n_valid) *reinterpret_cast<uint4*>(out + (size_t)(n_base + jb * 32 + t) * ldo + f) = v;
          }
        }
      }
    }
  }
};

// SPLIT_IN (the training forward, rp_train.hip): the old hi plane is read from xhi_in and the new one written to xhi, so
// the sub-layer's input bf16(x) - an operand of the backward - survives the update at no extra traffic.
template <bool SPLIT_IN, bool LO8 = false>
struct EpiResidT {  // x[token, feature] += acc on the two planes of the residual stream (+ ssp partials)
  static constexpr bool any_layout = true;
  // (no main_layout: the 2 x 4 wave grid that pays for the gated-GELU store was measured here too - QKV's store epilogue: no
  // difference, it writes whole lines already; this one: attention-out +3 %, FFN-out +2 % - round 5, exp30)
  bf16_t* __restrict__ xhi;  // bf16(x): also the next projection's A operand
  bf16_t* __restrict__ xlo;  // bf16(x - hi); LO8: the int8 extension plane of the 24-bit form (one byte per element)
  int ldx, n_valid;          // n_valid % 8 == 0
  float* __restrict__ ssp;   // optional: [np, ssp_ld] partial sums of squares, slot = feature / 64; slot-major so
                             // that a workgroup's statistics land in whole cache lines (token-major they were
                             // 16-B fragments of lines shared with workgroups on other XCDs)
  int np, ssp_ld;
  const bf16_t* __restrict__ xhi_in = nullptr;  // SPLIT_IN only
  Drop drop = {0u, 0u, 1.f};                    // SPLIT_IN only: dropout on the sub-layer's output before the add (HF:140, 400)
  uint32_t drop_site = 0;
  template <int FM, int FN>
  __device__ __forceinline__ void run(f32x16 (&acc)[FM][FN], int m_base, int n_base, int lane, char* stage) {
    static_assert(FM % 2 == 0, "one statistic slot per 64 features (two row fragments)");
    const int hi = lane >> 5, cl = lane & 31;
    // Blocks of 64 features x 32 tokens.  8 lanes x 16 B cover one token's 64 features = 128 contiguous bytes of
    // its row on EACH plane (8 tokens per wave-instruction).
    const int sub = lane & 7, rr = lane >> 3;
    constexpr int RB = 272;  // staging row: 64 floats + 16 B pad (32 rows = 8704 B <= EPI_STAGE_BYTES)
    constexpr int NB = (FM / 2) * FN;
    // The read-modify-write of x is latency-bound unless many loads are in flight: the old values of
    // the next block(s) are requested before the current one is staged, added and stored.  Loads are
    // unconditional, from a clamped address: a predicated load would sit in its own basic block and
    // make hipcc drain vmcnt to 0 around it, which serialises the whole epilogue.
    // (wave tiles of 128 x 128 keep their accumulators in AGPRs and have the VGPRs for 3 blocks ahead)
    // (round 5, 24-bit form on the 8-wave tile: 3 / 4 blocks in flight measured 2 / 5 % SLOWER than 2 on the attention-out projection)
    constexpr int DEPTH = (FM * FN >= 16) ? 4 : 2;
    uint4 xh[DEPTH][4];
    typename std::conditional<LO8, uint2, uint4>::type xl[DEPTH][4];
    const uint8_t* xlo8 = reinterpret_cast<const uint8_t*>(xlo);
    auto fetch = [&](int b, int p) {
      const int q = b / FN, j = b % FN;
      const int f = min(m_base + q * 64 + sub * 8, n_valid - 8);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const size_t off = (size_t)(n_base + j * 32 + c * 8 + rr) * ldx + f;
#ifdef RP_ABL_NOLOAD  // (probe builds only; body in probes/rp_probe_hooks.h)
        RP_ABL_NOLOAD_BODY(xh[p][c], xl[p][c], off);
#else
        if constexpr (SPLIT_IN)
          xh[p][c] = *reinterpret_cast<const uint4*>(xhi_in + off);
        else
          xh[p][c] = *reinterpret_cast<const uint4*>(xhi + off);
        if constexpr (LO8)
          xl[p][c] = *reinterpret_cast<const uint2*>(xlo8 + off);
        else
          xl[p][c] = *reinterpret_cast<const uint4*>(xlo + off);
#endif
      }
    };
#pragma unroll
    for (int b = 0; b < DEPTH - 1 && b < NB; ++b) fetch(b, b);
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const int q = b / FN, j = b % FN;
      const int f = m_base + q * 64 + sub * 8;
      const int slot = (m_base >> 6) + q;
      if (b + DEPTH - 1 < NB) fetch(b + DEPTH - 1, (b + DEPTH - 1) % DEPTH);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *reinterpret_cast<float4*>(stage + cl * RB + (i * 32 + 8 * g + 4 * hi) * 4) =
              make_float4(acc[2 * q + i][j][4 * g], acc[2 * q + i][j][4 * g + 1], acc[2 * q + i][j][4 * g + 2],
                          acc[2 * q + i][j][4 * g + 3]);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int t = c * 8 + rr;
        float4 d0 = *reinterpret_cast<const float4*>(stage + t * RB + sub * 32);
        float4 d1 = *reinterpret_cast<const float4*>(stage + t * RB + sub * 32 + 16);
        if constexpr (SPLIT_IN) {
          if (drop.thresh) {
            const uint32_t row = (uint32_t)(n_base + j * 32 + t), c0 = (uint32_t)f;
            float m[8];
#pragma unroll
            for (int e = 0; e < 8; e += 2) drop_mul2(drop, drop_site, row, c0 + e, m[e], m[e + 1]);
            d0.x *= m[0]; d0.y *= m[1]; d0.z *= m[2]; d0.w *= m[3];
            d1.x *= m[4]; d1.y *= m[5]; d1.z *= m[6]; d1.w *= m[7];
          }
        }
        float ss = 0.f;
        if (f < n_valid) {
          const size_t off = (size_t)(n_base + j * 32 + t) * ldx + f;
          const uint4 h = xh[b % DEPTH][c];
          const auto l = xl[b % DEPTH][c];
          uint4 oh;
          typename std::conditional<LO8, uint2, uint4>::type ol;
          if constexpr (LO8) {
            x24_update8(h, l, d0, d1, oh, ol, ss);
          } else {
            hilo_update2(h.x, l.x, d0.x, d0.y, oh.x, ol.x, ss);
            hilo_update2(h.y, l.y, d0.z, d0.w, oh.y, ol.y, ss);
            hilo_update2(h.z, l.z, d1.x, d1.y, oh.z, ol.z, ss);
            hilo_update2(h.w, l.w, d1.z, d1.w, oh.w, ol.w, ss);
          }
#ifdef RP_ABL_NOSTORE  // (probe builds only)
          RP_ABL_KEEP6(oh.x, oh.y, oh.z, oh.w, ol.x, ol.y);
#else
          *reinterpret_cast<uint4*>(xhi + off) = oh;
          if constexpr (LO8)
            *reinterpret_cast<uint2*>(reinterpret_cast<uint8_t*>(xlo) + off) = ol;
          else
            *reinterpret_cast<uint4*>(xlo + off) = ol;
#endif
        }
        if (ssp) {  // fixed shuffle tree over the 8 lanes of the token's 64 features
          // (round 6: DPP instead of __shfl_xor - 3 instructions instead of ~15 per token slice.  The third step takes the
          // value of lane 7 - i instead of lane i ^ 4: after two steps the four lanes of a quad hold the same bits, and both
          // lanes lie in the OTHER quad of the 8, so the sum and its rounding sequence are the ones of the xor tree)
          ss += dpp_f32<0xB1>(ss);
          ss += dpp_f32<0x4E>(ss);
          ss += dpp_f32<0x141>(ss);
          if (sub == 0 && slot < np) ssp[(size_t)slot * ssp_ld + n_base + j * 32 + t] = ss;
        }
      }
    }
  }
};

typedef EpiResidT<false> EpiResid;          // the two bf16 planes (kernel tests; the training step's SPLIT_IN form)
typedef EpiResidT<false, true> EpiResid8;   // bf16 plane + int8 extension plane: the inference pass

#ifndef RP_GEGLU_PRELOAD  // (0: the factors are requested by the epilogue's first instructions, as until round 6)
#define RP_GEGLU_PRELOAD 1
#endif
template <class RS>
struct EpiGegluBf16T {  // W rows interleaved 32 gate / 32 up: even row-fragments gate, odd up
  // full tiles of the 8-wave 256 x 256 configuration run as 2 x 4 waves of 128 features x 64 tokens: a wave's 64 outputs per
  // token are ONE 128-byte line of ff.  On the configuration's own 4 x 2 grid (64 x 128 per wave) two waves wrote the two
  // halves of every line: FFN-in 14.99 - 15.10 -> 14.81 ms per step (round 5, exp30).  Same MFMA chain per element.
  using main_layout = WaveLayout<2, 4, 4, 2>;
  bf16_t* out;         // [tokens, n_valid/2]
  int ldo, n_valid;    // n_valid counts interleaved rows (= 2 * d_ff)
  RS rs;
  // persistent launches: the tokens' RMSNorm factors requested in front of the tile's last k-tile (rp_gemm.h has_preload)
  static constexpr bool preload_hook = std::is_same<RS, RowScale>::value && RP_GEGLU_PRELOAD;
  float pre[4] = {0.f, 0.f, 0.f, 0.f};
  template <int FN>
  __device__ __forceinline__ void preload(int n_base, int lane) {
    static_assert(FN <= 4, "pre[]");
#pragma unroll
    for (int j = 0; j < FN; ++j) pre[j] = rs.get(n_base + j * 32 + (lane & 31));
  }
  template <int FM, int FN, bool PRE = false>
  __device__ __forceinline__ void run(f32x16 (&acc)[FM][FN], int m_base, int n_base, int lane, char* stage) {
    static_assert(FM % 2 == 0, "gate/up fragment pairs");
    const int hi = lane >> 5, cl = lane & 31;
    // 64 token rows at a time: staging rows of FM/2*32 outputs bf16 (FM*32 B <= 128 B)
    constexpr int LPR = FM * 2;         // lanes per token row (16 B each)
    constexpr int RPI = 64 / LPR;
    static_assert(FM * 32 <= 128, "staging row");
    const int sub = lane % LPR, rr = lane / LPR;
    const int f = (m_base >> 1) + sub * 8;
    float scv[FN];
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      if constexpr (PRE)
        scv[j] = pre[j];
      else
        scv[j] = rs.get(n_base + j * 32 + cl);
    }
#pragma unroll
    for (int jb = 0; jb < FN; jb += 2) {
#pragma unroll
      for (int jj = 0; jj < 2 && jb + jj < FN; ++jj) {
        const GegluConsts gc(scv[jb + jj]);  // the token's scale folded into the activation's constants (rp_util.h: geglu2)
#pragma unroll
        for (int i = 0; i < FM; i += 2)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            float y[4];
#ifdef RP_ABL_NOGELU  // (probe builds only)
#pragma unroll
            for (int e = 0; e < 4; ++e) y[e] = RP_ABL_NOGELU_BODY(acc[i][jb + jj][4 * g + e], acc[i + 1][jb + jj][4 * g + e]);
#else
#pragma unroll
            for (int e = 0; e < 4; e += 2) {
              const f32x2 yy = geglu2(f32x2{acc[i][jb + jj][4 * g + e], acc[i][jb + jj][4 * g + e + 1]},
                                      f32x2{acc[i + 1][jb + jj][4 * g + e], acc[i + 1][jb + jj][4 * g + e + 1]}, gc);
              y[e] = yy.x;
              y[e + 1] = yy.y;
            }
#endif
            uint2 v;
            v.x = pack_bf2(y[0], y[1]);
            v.y = pack_bf2(y[2], y[3]);
            *reinterpret_cast<uint2*>(stage + (jj * 32 + cl) * EPI_ROW_BYTES + ((i >> 1) * 32 + 8 * g + 4 * hi) * 2) = v;
          }
      }
      const int nrows = (FN - jb >= 2) ? 64 : 32;
#pragma unroll
      for (int t0 = 0; t0 < 64; t0 += RPI) {
        const int t = t0 + rr;
        if (t < nrows) {
          const uint4 v = *reinterpret_cast<const uint4*>(stage + t * EPI_ROW_BYTES + sub * 16);
#ifdef RP_ABL_NOSTORE  // (probe builds only)
          RP_ABL_KEEP4(v.x, v.y, v.z, v.w);
#else
          if (2 * f < n_valid) *reinterpret_cast<uint4*>(out + (size_t)(n_base + jb * 32 + t) * ldo + f) = v;
#endif
        }
      }
    }
  }
};
typedef EpiStoreBf16T<RowScale> EpiStoreBf16;
typedef EpiGegluBf16T<RowScale> EpiGegluBf16;
typedef EpiStoreBf16T<RowScaleFromSlots> EpiStoreBf16Slots;
typedef EpiGegluBf16T<RowScaleFromSlots> EpiGegluBf16Slots;
typedef WithRsPrologue<EpiStoreBf16T<RowScaleLds>> EpiStoreBf16Lds;
typedef WithRsPrologue<EpiGegluBf16T<RowScaleLds>> EpiGegluBf16Lds;

// Weight prefetch by the CUs a few-token launch leaves idle.  A pass of one proof state runs 18 - 112 workgroups, each
// streaming its 64 weight rows from HBM (a single-state retrieve() walks all 434 MB of weights: nothing is cached from
// call to call), and ONE CU pulls about 17 GB/s out of HBM however deep its ring is (measured: the FFN-out projection,
// 23 workgroups x 459 KB, takes 27 us on weights from HBM and 20 us on L2-resident ones, tools/gemm_bench.py COLD=48).
// The surplus workgroups of the launch touch those rows ahead of the compute workgroups - one dword per 128-byte
// line, K-major so that the first k-tiles arrive first - each on the XCD whose L2 its rows' consumer reads from
// (workgroup b runs on XCD b % 8 - observed, speed only - and xcd_remap gives every XCD a contiguous range of tiles).
// Helpers change no result: they only move cache lines.
__device__ __forceinline__ void gemm_prefetch_helper(const GemmOperand& A, int K, int BM, int nwg, int tiles_n, int hb,
                                                     int n_helpers) {
  const int x = hb & 7, j = hb >> 3, H = n_helpers >> 3;
  const int q = nwg >> 3, r = nwg & 7;
  const int base = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q, cnt = q + (x < r ? 1 : 0);
  if (cnt <= 0 || j >= H) return;
  const int f0 = base / tiles_n, f1 = (base + cnt - 1) / tiles_n;  // feature tiles this XCD's workgroups read
  const int r0 = f0 * BM, nrows = min((f1 + 1) * BM, A.rows) - r0;
  if (nrows <= 0) return;
  const int total = nrows * (K >> 6);  // 128-byte lines
  uint32_t tmp = 0;
  for (int l = j * (int)blockDim.x + (int)threadIdx.x; l < total; l += H * (int)blockDim.x) {
    const bf16_t* p = A.ptr + (size_t)(r0 + l % nrows) * A.ld + (size_t)(l / nrows) * 64;
    asm volatile("global_load_dword %0, %1, off" : "+v"(tmp) : "v"(p) : "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(tmp)::"memory");
}

// epilogues that take any wave layout with an even number of 32-feature fragments per wave declare `any_layout`
template <class E, class = void>
struct epi_any_layout : std::false_type {};
template <class E>
struct epi_any_layout<E, std::void_t<decltype(E::any_layout)>> : std::true_type {};
// an epilogue may name the wave grid it wants for FULL tiles of the 8-wave 256 x 256 configuration (`main_layout`)
template <class E, class C, class = void>
struct epi_main_layout {
  using type = C;
};
template <class E, class C>
struct epi_main_layout<E, C, std::void_t<typename E::main_layout>> {
  using type = std::conditional_t<C::BM == 256 && C::BN == 256 && C::NWAVES == 8, typename E::main_layout, C>;
};
template <class C, class Epi>
__host__ __device__ constexpr bool edge_layouts() {
  return C::PIPE != 0 && C::FP8 == 0 && C::BM == 256 && C::BN == 256 && C::NWAVES == 8 && C::OCC == 0 && epi_any_layout<Epi>::value;
}

// One pipelined tile; the last feature tile of an extent that ends inside it runs on a wave grid over the valid features
// only (rp_gemm.h WaveLayout).  128 of 256 (QKV, 1152 features): half the MFMA steps; 192 (d_model = 1472): three quarters.
template <class C, class Epi>
__device__ __forceinline__ void pipe_tile_edge(const GemmOperand& A, const GemmOperand& W, int K, int tm, int tn, Epi& epi,
                                               char* smem, bool edge_on) {
  if constexpr (edge_layouts<C, Epi>()) {
    const int vf = __builtin_amdgcn_readfirstlane(A.rows - tm * C::BM);
    if (edge_on && vf <= 192) {
      if (vf <= 128)
        gemm_tile_pipe<C, Epi, WaveLayout<2, 4, 2, 2>>(A, W, K, tm, tn, epi, smem);
      else
        gemm_tile_pipe<C, Epi, WaveLayout<1, 8, 6, 1>>(A, W, K, tm, tn, epi, smem);
      return;
    }
  }
  gemm_tile_pipe<C, Epi, typename epi_main_layout<Epi, C>::type>(A, W, K, tm, tn, epi, smem);
}

template <class C, class Epi>
__device__ __forceinline__ void gemm_kernel_body(GemmOperand A, GemmOperand W, int K, int tiles_m,
                                                  int tiles_n, int group_m, int stagger_ticks,
                                                  const int32_t* __restrict__ t_dev, Epi& epi, int n_helpers, char* smem) {
#ifdef RP_EXPERIMENTS  // first-round stagger of the big GEMMs (measured neutral): probe builds only
  if (stagger_ticks > 0 && blockIdx.x < 256) {
    // first round only (later workgroups inherit their CU's phase): phase = position among the 256 CUs, uniform
    // inside every XCD (workgroup b runs on XCD b % 8)
    const unsigned phase = ((blockIdx.x >> 3) & 31u) * 8u + (blockIdx.x & 7u);
    const unsigned long long until = wall_clock64() + (unsigned long long)stagger_ticks * phase / 256u;  // 100 MHz
    while (wall_clock64() < until) __builtin_amdgcn_s_sleep(16);
  }
#endif
  const int n_grid = (int)gridDim.x - n_helpers;  // compute workgroups (an upper bound with t_dev), padded to a multiple of 8
  int nwg = n_helpers ? tiles_m * tiles_n : n_grid;
  if (t_dev) {
    // rp_encode_padded: the grid covers an upper bound of the token count.  The live tiles are re-numbered over
    // the first nwg workgroups so that they still spread over all 8 XCDs (skipping by tile index left the live
    // token tiles - the first quarter of the logical range - on two XCDs: 3x slower).
    const int t_live = *t_dev;
    tiles_n = (t_live + C::BN - 1) / C::BN;
    nwg = tiles_m * tiles_n;
    W.rows = max(1, min(W.rows, t_live));  // (the token operand: rows beyond the live count read the last live row)
  }
  if ((int)blockIdx.x >= n_grid) {
    gemm_prefetch_helper(A, K, C::BM, nwg, tiles_n, (int)blockIdx.x - n_grid, n_helpers);
    return;
  }
  if ((int)blockIdx.x >= nwg) return;
  const int logical = xcd_remap(blockIdx.x, nwg);
  int tm, tn;
  tile_coords(logical, tiles_n, tiles_m, group_m, tn, tm);  // token tiles grouped, feature tiles inside
  if constexpr (C::PIPE != 0)
    pipe_tile_edge<C>(A, W, K, tm, tn, epi, smem, stagger_ticks >= 0);  // (stagger_ticks < 0: the option gemm_edge_layout = 0)
  else
    gemm_tile<C>(A, W, K, tm, tn, epi, smem);
}

template <class C, class Epi>
__global__ __launch_bounds__(C::THREADS) void gemm_kernel(GemmOperand A, GemmOperand W, int K, int tiles_m,
                                                          int tiles_n, int group_m, int stagger_ticks,
                                                          const int32_t* __restrict__ t_dev, Epi epi, int n_helpers) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  gemm_kernel_body<C>(A, W, K, tiles_m, tiles_n, group_m, stagger_ticks, t_dev, epi, n_helpers, smem);
}
// the same kernel held to 256 registers per lane so that two 4-wave workgroups share a CU (GemmCfg::OCC = 2)
template <class C, class Epi>
__global__ __launch_bounds__(C::THREADS, 2) void gemm_kernel_occ2(GemmOperand A, GemmOperand W, int K, int tiles_m,
                                                                  int tiles_n, int group_m, int stagger_ticks,
                                                                  const int32_t* __restrict__ t_dev, Epi epi, int n_helpers) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  gemm_kernel_body<C>(A, W, K, tiles_m, tiles_n, group_m, stagger_ticks, t_dev, epi, n_helpers, smem);
}

// FULL and HALF tiles in one launch (MixedPlan below).  The token rows that make whole rounds of 256 x 256 tiles are the
// full tiles; the rest are half tiles (CH: the same features x 128 tokens), some handed out FIRST and the others LAST:
//   blocks [0, nh1)  half tiles | [nh1, nh1_pad) exit | [nh1_pad, nh1_pad + n_full)  full tiles | then the last half tiles.
// About half the CUs start on a half tile, so the chip runs as two groups half a tile period apart from then on: the
// read-modify-write epilogues (a burst of 100 MB when all 256 CUs reach them together: HBM-bound for ~20 us while the
// MFMA pipes idle) of one group fall under the main loops of the other.  The group that started on full tiles finishes
// them half a period early and takes the last half tiles: both end together, as the separate tail round had it.
// Same K-ascending chains per output element: not a bit changes.
template <class C, class CH, class Epi>
__global__ __launch_bounds__(C::THREADS) void gemm_kernel_mixed(GemmOperand A, GemmOperand W, int K, int tiles_m,
                                                                int full_rows, int nh1, int nh1_pad, int group_m,
                                                                int edge_on, Epi epi) {
  static_assert(CH::BM == C::BM && 2 * CH::BN == C::BN && CH::THREADS == C::THREADS && CH::PIPE != 0 && C::PIPE != 0, "half tile");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int b = blockIdx.x, n_full = full_rows * tiles_m;
  if (b >= nh1_pad && b < nh1_pad + n_full) {
    const int logical = xcd_remap(b - nh1_pad, n_full);  // (nh1_pad % 8 == 0: block b - nh1_pad runs on XCD (b - nh1_pad) % 8)
    int tm, tn;
    tile_coords(logical, full_rows, tiles_m, group_m, tn, tm);
    pipe_tile_edge<C>(A, W, K, tm, tn, epi, smem, edge_on != 0);
  } else if (b < nh1 || b >= nh1_pad + n_full) {
    const int h = b < nh1 ? b : b - (nh1_pad + n_full) + nh1;  // feature tiles fastest: neighbours share the token rows
    gemm_tile_pipe<CH>(A, W, K, h % tiles_m, 2 * full_rows + h / tiles_m, epi, smem);
  }
}
// (device_cu_count(): rp_util.h)
struct MixedPlan {
  int full_rows = 0, half_first = 0, half_last = 0;  // token tiles of 256 | rows of 128 tokens handed out first | last
};
// tiles_f feature tiles x tiles_t token tiles of 256 on n_cus CUs; nothing (full_rows = 0) when the full tiles make whole
// rounds anyway, when the last round is nearly full, or when the half tiles would not fit one round
inline MixedPlan plan_mixed(int tiles_f, int tiles_t, int n_cus) {
  MixedPlan p;
  int g = tiles_f, b = n_cus;
  while (b) {
    const int t = g % b;
    g = b;
    b = t;
  }
  const int unit = n_cus / g;  // token tiles per whole number of rounds
  const int t1 = tiles_t / unit * unit, rest = tiles_t - t1;
  if (t1 == 0 || rest == 0 || rest * tiles_f > (7 * n_cus) / 10 || 2 * rest * tiles_f > n_cus) return p;
  p.full_rows = t1;
  p.half_first = std::min(2 * rest, (n_cus / 2) / tiles_f);
  p.half_last = 2 * rest - p.half_first;
  return p;
}

// The same hand-out for a tile count that has no whole rounds of whole token rows (round 6, the training step: the
// gated-GELU backward GEMM is 14 x 41 = 574 tiles - 2.24 rounds of 256 CUs, three tile periods; the FFN-in forward 28 x 41 =
// 1148 - 4.48 rounds, five periods).  Candidate plans - the token rows that fit floor(tiles / CUs) rounds (or one / two rows
// fewer) as full tiles, the rest as half tiles - are list-scheduled in the launch's block order (a half tile = 0.55
// periods) and the best is taken if it saves more than a quarter period: 2.5 and 4.5 periods here.  Cached per shape.
inline MixedPlan plan_mixed_loose(int tiles_f, int tiles_t, int n_cus) {
  struct Entry {
    int tf, tt, cus;
    MixedPlan p;
  };
  thread_local Entry cache[8] = {};
  thread_local int next = 0;
  for (const Entry& e : cache)
    if (e.tf == tiles_f && e.tt == tiles_t && e.cus == n_cus) return e.p;
  MixedPlan best;
  const int total = tiles_f * tiles_t, rounds = total / n_cus;
  if (rounds > 0 && total % n_cus != 0 && n_cus <= 1024) {
    double best_t = (double)(rounds + 1) - 0.26;
    std::vector<double> cu(n_cus);
    auto makespan = [&](const MixedPlan& p) {
      std::fill(cu.begin(), cu.end(), 0.0);
      std::make_heap(cu.begin(), cu.end(), std::greater<double>());
      double end = 0.0;
      auto run = [&](int n, double d) {
        for (int i = 0; i < n; ++i) {
          std::pop_heap(cu.begin(), cu.end(), std::greater<double>());
          cu.back() += d;
          end = std::max(end, cu.back());
          std::push_heap(cu.begin(), cu.end(), std::greater<double>());
        }
      };
      run(p.half_first * tiles_f, 0.55);
      run(p.full_rows * tiles_f, 1.0);
      run(p.half_last * tiles_f, 0.55);
      return end;
    };
    const int top = rounds * n_cus / tiles_f;
    for (int full_rows = top; full_rows >= std::max(1, top - 2); --full_rows) {
      const int rest = tiles_t - full_rows;
      if (rest <= 0) continue;
      MixedPlan p;
      p.full_rows = full_rows;
      p.half_first = std::min(2 * rest, (n_cus / 2) / tiles_f);
      p.half_last = 2 * rest - p.half_first;
      const double t = makespan(p);
      if (t < best_t) {
        best_t = t;
        best = p;
      }
    }
  }
  cache[next] = Entry{tiles_f, tiles_t, n_cus, best};
  next = (next + 1) % 8;
  return best;
}
// epilogues of the training step declare `loose_mixed`: their GEMMs take plan_mixed_loose whatever the projection class
template <class E, class = void>
struct epi_loose_mixed : std::false_type {};
template <class E>
struct epi_loose_mixed<E, std::void_t<decltype(E::loose_mixed)>> : std::true_type {};
template <class C, class Epi>
constexpr bool mixed_capable() {
  return C::PIPE != 0 && C::FP8 == 0 && C::BM == 256 && C::BN == 256 && C::NWAVES == 8 && C::OCC == 0 && epi_extra_lds<Epi>::value == 0;
}

// One workgroup per CU walking its share of the tiles (gemm_tiles_persist).  Workgroup b runs on XCD b % 8 (observed, speed
// only) and takes the tiles j, j + 32, j + 64, ... of that XCD's contiguous range of logical tile ids (j = b / 8): at any
// moment the 32 workgroups of an XCD sit on 32 consecutive logical ids, exactly as the one-tile-per-workgroup launch has them.
template <class C, class Epi>
__global__ __launch_bounds__(C::THREADS) void gemm_kernel_persist(GemmOperand A, GemmOperand W, int K, int tiles_m,
                                                                  int tiles_n, int group_m, int edge_on,
                                                                  const int32_t* __restrict__ t_dev, Epi epi) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int nwg = tiles_m * tiles_n;
  if (t_dev) {  // rp_encode_padded: the token count is known on the device only (see gemm_kernel_body)
    const int t_live = *t_dev;
    tiles_n = (t_live + C::BN - 1) / C::BN;
    nwg = tiles_m * tiles_n;
    W.rows = max(1, min(W.rows, t_live));
  }
  const int xcd = blockIdx.x & 7, per = (int)gridDim.x >> 3;  // (the grid is a multiple of 8)
  const int q = nwg >> 3, r = nwg & 7;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q, cnt = q + (xcd < r ? 1 : 0);
  int i = blockIdx.x >> 3;
  auto next_tile = [&](int& tm, int& tn) {
    if (i >= cnt) return false;
    tile_coords(base + i, tiles_n, tiles_m, group_m, tn, tm);  // token tiles grouped, feature tiles inside
    i += per;
    return true;
  };
  // full tiles: the epilogue's preferred wave grid when it names one (main_layout), else the configuration's own
  gemm_tiles_persist<C, edge_layouts<C, Epi>(), typename epi_main_layout<Epi, C>::type>(A, W, K, next_tile, epi, smem, edge_on != 0);
}
// an epilogue with metadata behind the ring (RowScaleLds) has 24 KiB for it in the persistent layout
template <class Epi>
static bool epi_fits_persist(const Epi& epi) {
  if constexpr (epi_extra_lds<Epi>::value != 0)
    return epi.rs.np * 1024 <= PERSIST_LDS_BYTES - PERSIST_META_OFF;
  else
    return true;
}
template <class C>
constexpr bool persist_capable() {
  return C::PIPE != 0 && C::FP8 == 0 && C::KTAIL == 0 && C::NSTAGE == 2 && C::STAGE_BYTES == 64 * 1024 && C::NWAVES == 8 && C::OCC == 0;
}

template <class C, class Epi>
static auto pick_gemm_kernel() {
  if constexpr (C::OCC == 2)
    return gemm_kernel_occ2<C, Epi>;
  else
    return gemm_kernel<C, Epi>;
}

// `w` = weight matrix [n_rows_w, K] (row operand: tile rows = output features, clamped at the edge),
// `a` = activations [M, K] (column operand: tile cols = tokens, M a multiple of the token tile).
template <class C, class Epi>
static RpStatus launch_gemm_cfg(GemmOperand w, GemmOperand a, int K, Epi epi, hipStream_t stream,
                                int prof_class, int tokens_valid, const int32_t* t_dev) {
  auto kern = pick_gemm_kernel<C, Epi>();
  static_assert(C::OCC == 0 || (C::OCC == 2 && C::THREADS == 256), "OCC: two 4-wave workgroups per CU");
  constexpr int LDS = (epi_extra_lds<Epi>::value ? C::RING_BYTES : C::LDS_BYTES) + epi_extra_lds<Epi>::value;
  static_assert(epi_extra_lds<Epi>::value == 0 || (C::PIPE != 0 && C::RING_BYTES >= C::NWAVES * EPI_STAGE_BYTES),
                "metadata behind the ring: pipelined tiles only");
  static LdsAttrOnce attr;
  RP_HIP(attr.ensure((const void*)kern, LDS));
  RP_REQUIRE(K % C::BK == 0 && a.rows % C::BN == 0, "gemm: K=%d must be a multiple of %d, M=%d of %d", K, C::BK,
             a.rows, C::BN);
  const int rows_needed = (tokens_valid > 0 && tokens_valid < a.rows) ? tokens_valid : a.rows;
  const int tiles_f = (w.rows + C::BM - 1) / C::BM, tiles_t = (rows_needed + C::BN - 1) / C::BN;
  // tile order: feature tiles fastest inside groups of `group` token tiles (shared activation panels)
  const int group = max(1, g_gemm_group_m * 128 / C::BN);
  ProfScope ps(stream, prof_class);
  const int stagger_ticks = !g_gemm_edge_layout ? -1 : (C::PIPE != 0 && tiles_f * tiles_t > 512) ? g_gemm_stagger_us[prof_class] * 100 : 0;
  // few-token launches (one group of token tiles, well under one round of the chip) on weights worth prefetching: the idle
  // CUs become prefetch helpers (above)
  int n_grid = tiles_f * tiles_t, n_helpers = 0;
  const int n_cus = device_cu_count();
  if (g_gemm_helpers && n_grid <= n_cus / 2 && tiles_t <= group && (size_t)w.rows * K * 2 >= ((size_t)2 << 20)) {
    n_grid = (n_grid + 7) & ~7;
    n_helpers = std::min(g_gemm_helpers, (n_cus - n_grid) & ~7);
  }
  // token rows beyond the valid count are read as copies of the last valid row (the operand clamps at `rows`): the 27
  // padding rows of a 101-token state cost one cache line per DMA piece instead of eight
  a.rows = rows_needed;
  if constexpr (mixed_capable<C, Epi>()) {
    const bool fwd_mixed = !epi_loose_mixed<Epi>::value && edge_layouts<C, Epi>() && prof_class >= RP_K_GEMM_QKV && prof_class <= RP_K_GEMM_WO &&
                           ((g_gemm_mixed >> (prof_class - RP_K_GEMM_QKV)) & 1);
    const bool bwd_mixed = epi_loose_mixed<Epi>::value && g_gemm_mixed_bwd;  // the training step's GEMMs
    if ((fwd_mixed || bwd_mixed) && n_helpers == 0 && !t_dev && K >= 2 * C::BK) {
      const MixedPlan mp = fwd_mixed ? plan_mixed(tiles_f, tiles_t, n_cus) : plan_mixed_loose(tiles_f, tiles_t, n_cus);
      if (mp.full_rows) {
        using CH = GemmCfg<C::BM, C::BN / 2, C::BK, C::WM, C::WN, C::NSTAGE, C::PIPE>;
        auto mk = gemm_kernel_mixed<C, CH, Epi>;
        static LdsAttrOnce mattr;
        RP_HIP(mattr.ensure((const void*)mk, C::LDS_BYTES));
        const int nh1 = mp.half_first * tiles_f, nh1_pad = (nh1 + 7) & ~7;
        hipLaunchKernelGGL(mk, dim3(nh1_pad + mp.full_rows * tiles_f + mp.half_last * tiles_f), dim3(C::THREADS), C::LDS_BYTES,
                           stream, w, a, K, tiles_f, mp.full_rows, nh1, nh1_pad, group, g_gemm_edge_layout, epi);
        RP_CHECK_LAUNCH();
        return RP_OK;
      }
    }
  }
  if constexpr (persist_capable<C>()) {
    // more tiles than CUs: one persistent workgroup per CU instead of one workgroup per tile
    const int slots = n_cus & ~7;
    // (K >= two k-tiles: the persistent loop requests ring slot 1 unconditionally)
    if (prof_class >= RP_K_GEMM_QKV && prof_class <= RP_K_GEMM_WO && ((g_gemm_persist >> (prof_class - RP_K_GEMM_QKV)) & 1) &&
        n_helpers == 0 && n_grid > slots && K >= 2 * C::BK && epi_fits_persist(epi)) {
      auto pk = gemm_kernel_persist<C, Epi>;
      static LdsAttrOnce pattr;
      RP_HIP(pattr.ensure((const void*)pk, PERSIST_LDS_BYTES));
      hipLaunchKernelGGL(pk, dim3(slots), dim3(C::THREADS), PERSIST_LDS_BYTES, stream, w, a, K, tiles_f, tiles_t, group,
                         g_gemm_edge_layout, t_dev, epi);
      RP_CHECK_LAUNCH();
      return RP_OK;
    }
  }
  hipLaunchKernelGGL(kern, dim3(n_grid + n_helpers), dim3(C::THREADS), LDS, stream, w, a, K, tiles_f, tiles_t, group,
                     stagger_ticks, t_dev, epi, n_helpers);
  RP_CHECK_LAUNCH();
  return RP_OK;
}

// rows of the activation workspace are padded to this so every variant tiles the tokens exactly
constexpr int GEMM_M_ALIGN = 256;

// Which tile configuration a projection runs with.  tokens_valid = real token count of the pass (0: all M rows).
static int pick_gemm_variant(int prof_class, int M, int n_rows_w, int K, int tokens_valid) {
  int v = prof_class == RP_K_GEMM_O     ? (g_gemm_variant_o >= 0 ? g_gemm_variant_o : (M >= 57344 ? 26 : 0))
          : prof_class == RP_K_GEMM_WO  ? g_gemm_variant_wo
          : prof_class == RP_K_GEMM_QKV ? g_gemm_variant_qkv
                                        : g_gemm_variant;
  const bool k64 = (K % 64 == 0), m256 = (M % 256 == 0);
  // Few tokens (the prover's single-state query): the 256 x 256 tiling would leave
  // most CUs idle and each workgroup latency-bound on its K loop.  Switch to 64-feature tiles with a
  // deep LDS ring so every workgroup streams its weight slab with several K-steps of DMA in flight.
  // Measured per launch (tools/gemm_bench.py, SKINNY=0 FUSED=1; us) at 256 / 512 / 1024 / 2048 tokens:
  //            64x128x64 (16)    64x128x32 (15)    64x256x32 (12)   128x128x32 (0)    256x256x64 (26)
  //   FFN-out  27/28/30/ -       29/30/32/60        - / - /47/52    39/41/42/46        - / - /76/78
  //   QKV      12/13/13/ -       14/14/15/26        - / - /20/21    17/17/18/19        - / - /29/32
  //   attn-out  8/ 8/ 9/ -        8/ 9/ 9/15        - / - /12/14    11/11/12/13        - / - /18/20
  //   FFN-in   14/25/47/ -       15/29/53/92        - / - /44/85    19/20/34/64        - / - /35/41
  // (FFN-out stays ~28 us from 256 to 1024 tokens: 23 feature tiles, each workgroup walking 459 KB of weights.)
  if (g_gemm_skinny && m256) {
    const int tv = (tokens_valid > 0 && tokens_valid < M) ? tokens_valid : M;
    const bool few_tiles = ((n_rows_w + 255) / 256) * (M / 256) < 96;
    if (few_tiles && g_gemm_skinny_variant != 0) v = g_gemm_skinny_variant;  // forced by a test (15)
    else if (prof_class == RP_K_GEMM_WI) v = (tv <= 256) ? 16 : (few_tiles || tv <= 1024) ? 0 : v;
    else if (few_tiles) v = (tv <= 1024) ? 16 : 0;
  }
  if (v == 26 && !k64) v = 9;  // 64-wide K tiles need K % 64 == 0
  if (v == 16 && !k64) v = 15;
  if (v == 16 && g_gemm_small_pipe) v = 17;  // the same tile on the software-pipelined loop
  if (v >= 5 && !m256) v = 0;
  return v;
}
inline bool small_variant(int v) { return v == 0 || (v >= 15 && v <= 17); }

// GemmCfg<feature tile, token tile, BK, waves over features, waves over tokens, stages[, pipelined]>:
//   26       pipelined 256 x 256 x 64, 8 waves          (the encoder's big GEMMs; the 4-wave form 20 - 128 x 128 per wave,
//            accumulators in AGPRs - lost the in-step A/B of round 3 and left the sources in round 6, as did 12 = 64 x 256 x 32)
//   27 / 28  (RP_EXPERIMENTS builds) pipelined 256 x 128 x 32 / 128 x 256 x 32 (features x tokens), 3 stages, 4 waves, TWO workgroups per CU
//   30       pipelined 256 x 128 x 64, 8 waves          (the FFN-out projection's tail round: half tiles, one per CU)
//   9        plain 256 x 256 x 32, 3 stages             (K % 64 != 0)
//   0        plain 128 x 128 x 32, 3 stages, 2 blocks/CU (attention-out; token counts not a multiple of 256)
//   17       64 x 128 x 64, 4 stages, pipelined loop    (up to ~1024 tokens: single-state queries)
//   16 / 15  the same on the plain loop / x 32, 7 stages (16: kept selectable; 15: K % 64 != 0)
// SMALL_ONLY: the epilogue type exists for the small configurations only (pick_gemm_variant said so).
template <bool SMALL_ONLY = false, class Epi>
static RpStatus launch_gemm(const bf16_t* A, int lda, int M, const bf16_t* W, int ldw, int n_rows_w,
                            int K, Epi epi, hipStream_t stream, int prof_class, int tokens_valid = 0,
                            const int32_t* t_dev = nullptr, int force_variant = -1) {
  GemmOperand a{A, lda, M}, w{W, ldw, n_rows_w};
  const int v = force_variant >= 0 ? force_variant : pick_gemm_variant(prof_class, M, n_rows_w, K, tokens_valid);
  if constexpr (!SMALL_ONLY) {
    switch (v) {
      case 26: return launch_gemm_cfg<GemmCfg<256, 256, 64, 4, 2, 2, 1>>(w, a, K, epi, stream, prof_class, tokens_valid, t_dev);
      case 9: return launch_gemm_cfg<GemmCfg<256, 256, 32, 4, 2, 3>>(w, a, K, epi, stream, prof_class, tokens_valid, t_dev);
      case 30: return launch_gemm_cfg<GemmCfg<256, 128, 64, 4, 2, 2, 1>>(w, a, K, epi, stream, prof_class, tokens_valid, t_dev);
#ifdef RP_EXPERIMENTS  // round 5, measured and rejected (profiles/r05_epilogue_overlap.md): FFN-in / FFN-out +17 % time; 29 = two
      // pipelined 128 x 128 x 64 workgroups per CU for the attention-out projection: 2.46 vs 2.29 ms per step
      case 29: return launch_gemm_cfg<GemmCfg<128, 128, 64, 2, 2, 2, 1, 0, 0, 0, 2>>(w, a, K, epi, stream, prof_class, tokens_valid, t_dev);
      case 27: return launch_gemm_cfg<GemmCfg<256, 128, 32, 2, 2, 3, 1, 0, 0, 0, 2>>(w, a, K, epi, stream, prof_class, tokens_valid, t_dev);
      case 28: return launch_gemm_cfg<GemmCfg<128, 256, 32, 2, 2, 3, 1, 0, 0, 0, 2>>(w, a, K, epi, stream, prof_class, tokens_valid, t_dev);
#endif
      default: break;
    }
  } else {
    RP_REQUIRE(small_variant(v), "tile configuration %d has no fused row-scale form", v);
  }
  switch (v) {
    case 15: return launch_gemm_cfg<GemmCfg<64, 128, 32, 1, 4, 7>>(w, a, K, epi, stream, prof_class, tokens_valid, t_dev);
    case 16: return launch_gemm_cfg<GemmCfg<64, 128, 64, 1, 4, 4>>(w, a, K, epi, stream, prof_class, tokens_valid, t_dev);
    case 17: return launch_gemm_cfg<GemmCfg<64, 128, 64, 1, 4, 4, 1>>(w, a, K, epi, stream, prof_class, tokens_valid, t_dev);
    default: return launch_gemm_cfg<GemmCfg<128, 128, 32, 2, 2, 3>>(w, a, K, epi, stream, prof_class, tokens_valid, t_dev);
  }
}

inline bool big_variant(int v) { return v == 26; }
// epilogues that exist for the pipelined 256 x 256 tiles only (metadata behind the ring)
template <class Epi>
static RpStatus launch_gemm_big(const bf16_t* A, int lda, int M, const bf16_t* W, int ldw, int n_rows_w, int K, Epi epi,
                                hipStream_t stream, int prof_class, int tokens_valid, const int32_t* t_dev) {
  GemmOperand a{A, lda, M}, w{W, ldw, n_rows_w};
  const int v = pick_gemm_variant(prof_class, M, n_rows_w, K, tokens_valid);
  RP_REQUIRE(big_variant(v), "tile configuration %d has no LDS row-scale form", v);
  return launch_gemm_cfg<GemmCfg<256, 256, 64, 4, 2, 2, 1>>(w, a, K, epi, stream, prof_class, tokens_valid, t_dev);
}

// ------------------------------------------------------------------------------------------
// K4+K5: T5 self-attention, flash-style, varlen.
//   scores = q·k (NO 1/sqrt(d) scaling, HF:197) + bias[h, clamp(j-i, -128, 128)]; keys >= len are
//   excluded (HF uses finfo.min via where(mask, bias, min): identical result whenever a row has
//   at least one real key, which every real query row does); fp32 online softmax; o = p·v.
//   The additive bias depends only on j-i and saturates at |j-i| >= max_distance, so it is a
//   [H, 2*max_distance+1] table (built on the host from relative_attention_bias.weight with HF's
//   bucket function) instead of HF's dense [1,H,L,L] tensor.
//
//   Workgroup = (128 queries of one sequence, one head); wave w owns queries 32w..32w+31.
//   S^T = K·Q^T is computed so that each lane holds, for ONE query (lane & 31), 16 keys per 32-key
//   block: the softmax row reductions are in-lane plus one exchange with lane^32.  O^T = V^T·P^T
//   reuses those registers directly as the MFMA B operand (the k-slot -> key permutation implied
//   by the accumulator layout is applied to V^T when its A fragment is read from LDS).
// ------------------------------------------------------------------------------------------
constexpr int ATT_Q = 128, ATT_KV = 64;
constexpr int ATT_TAB_MAX = 1024;  // max table entries (2*max_distance+1) + the kernel's 2 x 64 entries of padding
//   * K and V tiles go HBM -> LDS by LDS-DMA into a 2-stage ring (tile t+1 in flight under the
//     MFMAs/softmax of tile t, one barrier per tile, no VGPR staging, no ds_write at all);
//   * V stays row-major in LDS ([d-half][key][32 d], 64-B rows) and its MFMA A fragments
//     (V^T: 4 consecutive keys for one d per lane) are produced by the gfx950 transpose read
//     ds_read_b64_tr_b16: within each 16-lane group, lane i supplies the address of
//     V[k0 + i/4][d0 + 4 (i%4) .. +3] and lane l receives V[k0 .. k0+3][d0 + l]  (mapping measured
//     with tools/probes/tr_probe.hip); 4 rows x 64 B = one 256-B bank row: conflict-free;
//   * K is staged with the GEMM's XOR swizzle (slot ^= (row >> 1) & 7 on the DMA source address);
//   * waves whose 32 queries lie past the sequence end only help with the DMA;
//   * one workgroup per entry of the pass's work list (worklist_kernel below: 128-query blocks, longest
//     sequence first; at most T/128 + B entries).  A (max_len/128) x B grid would launch ~6 empty workgroups per
//     useful one on the benchmark's length mix.
// ------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(4))) short v4s16;
constexpr int AT2_K_BYTES = 64 * 128, AT2_V_BYTES = 64 * 128, AT2_STAGE = AT2_K_BYTES + AT2_V_BYTES;

// Work list of one encoder pass, built once and used by all layers: entry = {first token of the sequence, its length,
// first query of the block, 0}, one per 128-query block, ordered by DESCENDING key count (buckets of 64 keys, longest
// first; inside a bucket by sequence, then block).  A block's cost grows with its sequence's length (32 key tiles at
// 2048 tokens against 1-2 for a short state), and the grid is a few rounds deep: dispatched in corpus order, a long
// sequence met late kept a handful of CUs busy long after everything else had finished; longest-first closes that
// tail.  It also replaces the two block-wide counting rounds every workgroup of every layer spent finding its sequence.
// Entries beyond the live count have length 0.  One workgroup; a counting sort over 64 length buckets in LDS.
constexpr int ATT_BUCKETS = 64;
constexpr int POOL_CHUNK = 128;  // tokens per workgroup of the pooling pass (the training step's; the inference pass: g_pool_chunk)
constexpr int POOL_CHUNK_MIN = 32;  // smallest selectable chunk: sizes the inference workspace
// The same launch lays out the pooling pass's list: chunk c of sequence b is entry cu[b] / 128 + b + c = {first token,
// length, c, b} (strictly increasing in b, at most T/128 + B entries; a gap entry has length 0).
static __global__ __launch_bounds__(1024) void worklist_kernel(const int32_t* __restrict__ cu, int batch,
                                                        int4* __restrict__ work, int n_slots,
                                                        int4* __restrict__ pwork, int n_pslots, int pool_chunk) {
  __shared__ int s_cnt[ATT_BUCKETS], s_pos[ATT_BUCKETS], s_live;
  const int tid = threadIdx.x;
  // bucket k holds sequences of (k, k+1] * 64 keys; the last one everything longer
  auto bucket_of = [](int len) { return min((len - 1) >> 6, ATT_BUCKETS - 1); };
  if (tid < ATT_BUCKETS) s_cnt[tid] = 0;
  __syncthreads();
  for (int b = tid; b < batch; b += 1024) {
    const int len = cu[b + 1] - cu[b];
    if (len > 0) atomicAdd(&s_cnt[bucket_of(len)], (len + ATT_Q - 1) / ATT_Q);
  }
  __syncthreads();
  if (tid < ATT_BUCKETS) {  // blocks of all longer buckets come first
    int pos = 0;
    for (int j = ATT_BUCKETS - 1; j > tid; --j) pos += s_cnt[j];
    s_pos[tid] = pos;
    if (tid == 0) s_live = pos + s_cnt[0];
  }
  __syncthreads();
  // the order INSIDE a bucket is whatever the atomics give: every block's result is independent of its position
  for (int b = tid; b < batch; b += 1024) {
    const int s0 = cu[b], s1 = cu[b + 1], len = s1 - s0;
    const int pbase = s0 / pool_chunk + b;
    const int pnext = (b + 1 < batch) ? s1 / pool_chunk + b + 1 : n_pslots;
    int c = 0;
    for (; c * pool_chunk < len; ++c) pwork[pbase + c] = make_int4(s0, len, c, b);
    for (int i = pbase + c; i < pnext; ++i) pwork[i] = make_int4(0, 0, 0, 0);
    if (b == 0)
      for (int i = 0; i < pbase; ++i) pwork[i] = make_int4(0, 0, 0, 0);  // cu[0] is 0 in every caller; kept general
    if (len <= 0) continue;
    int pos = atomicAdd(&s_pos[bucket_of(len)], (len + ATT_Q - 1) / ATT_Q);
    for (int q0 = 0; q0 < len; q0 += ATT_Q) work[pos++] = make_int4(s0, len, q0, 0);
  }
  for (int i = s_live + tid; i < n_slots; i += 1024) work[i] = make_int4(0, 0, 0, 0);
  if (batch == 0)
    for (int i = tid; i < n_pslots; i += 1024) pwork[i] = make_int4(0, 0, 0, 0);
}

// TRAIN: dropout on the attention probabilities (HF:168, 360: after the softmax; the row sum uses the undropped values)
// (the dropout instantiation needs more registers than four workgroups per CU leave a wave)
template <bool TRAIN, bool DROP = false>
static __global__ __launch_bounds__(256, DROP ? 2 : 4) void attention_kernel(const bf16_t* __restrict__ qkv,
                                                         const int4* __restrict__ work,
                                                         const float* __restrict__ bias_tab,
                                                        bf16_t* __restrict__ out, int H, int maxd,
                                                         float* __restrict__ lse2, int lse_ld, Drop drop,
                                                         uint32_t drop_site) {
  __shared__ __attribute__((aligned(16))) char smem[2 * AT2_STAGE + ATT_TAB_MAX * 4];
  float* tab = reinterpret_cast<float*>(smem + 2 * AT2_STAGE);

  const int tid = threadIdx.x, lane = tid & 63, hi = lane >> 5, cl = lane & 31;
  // heads vary fastest in dispatch order: the six workgroups that read the six 128-byte pieces of the same qkv rows
  // (a 2304-byte row holds every head's q, k and v) run at the same time, so DRAM sees whole rows, not pieces
  const int h = blockIdx.x;
  const int4 wk = work[blockIdx.y];
  const int s0 = wk.x, len = wk.y, q0 = wk.z;
  if (len == 0) return;

  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int inner = H * 64, ld = 3 * inner;
  const int ntab = 2 * maxd + 1;
  // The LDS copy of the head's table carries TAB_PAD saturated entries on either side: a 32-key x 32-query block whose
  // offsets straddle +-maxd (they span 63 values) then indexes it without clamping, like an interior block.  Without the
  // padding those blocks took the per-score path below (clamp, compare, select: ~3 x the instructions), and the kernel is
  // bound by VALU issue (profiles/r04_attention_ablation.md).
  constexpr int TAB_PAD = 64;
  for (int i = tid; i < ntab + 2 * TAB_PAD; i += 256) tab[i] = bias_tab[h * ntab + min(max(i - TAB_PAD, 0), ntab - 1)];

  const int wq0 = q0 + wave * 32;
  const bool active = wq0 < len;  // wave-uniform
  const int qi = wq0 + cl;
  bf16x8 qf[4];
  {
    const bf16_t* qp = qkv + (size_t)(s0 + min(qi, len - 1)) * ld + h * 64 + hi * 8;
#pragma unroll
    for (int c = 0; c < 4; ++c) qf[c] = *reinterpret_cast<const bf16x8*>(qp + c * 16);
  }

  // DMA pieces of this wave: K pieces {2w, 2w+1} (8 keys x 128 B each), V pieces {2w, 2w+1}
  // (d-half p >> 2, 16 keys x 64 B each); LDS destinations are lane-linear.
  const bf16_t* kv_base = qkv + (size_t)s0 * ld + inner + h * 64;
  auto stage = [&](int kt, int buf) {
    char* base = smem + buf * AT2_STAGE;
    const int k0 = kt * ATT_KV;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int p = wave * 2 + e;
      {
        const int key = 8 * p + (lane >> 3);
        const int kc = (lane & 7) ^ ((key >> 1) & 7);
        const bf16_t* src = kv_base + (size_t)min(k0 + key, len - 1) * ld + kc * 8;
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)(base + p * 1024), 16, 0, 0);
      }
      {
        const int key = 16 * (p & 3) + (lane >> 2);
        const bf16_t* src = kv_base + (size_t)min(k0 + key, len - 1) * ld + inner + (p >> 2) * 32 + (lane & 3) * 8;
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)(base + AT2_K_BYTES + p * 1024), 16, 0, 0);
      }
    }
  };

  // fragment offsets
  int k_off[2][4];
#pragma unroll
  for (int kb = 0; kb < 2; ++kb)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int key = kb * 32 + cl;
      k_off[kb][c] = key * 128 + (((c * 2 + hi) ^ ((key >> 1) & 7)) << 4);
    }
  // V^T fragment, first transpose-read of slab 0 / d-half 0: row 4 hi + (lane&15)/4, 4 d's at 4 (lane&3) + 16 ((lane>>4)&1)
  const int v_off0 = AT2_K_BYTES + (4 * hi + ((lane & 15) >> 2)) * 64 + (4 * (lane & 3) + 16 * ((lane >> 4) & 1)) * 2;

  f32x16 o[2];
#pragma unroll
  for (int d = 0; d < 2; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  const int n_tiles = (len + ATT_KV - 1) / ATT_KV;
  stage(0, 0);
  for (int kt = 0; kt < n_tiles; ++kt) {
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();  // tile kt complete in LDS; tile kt-1's buffer free (and tab written)
    if (kt + 1 < n_tiles) stage(kt + 1, (kt + 1) & 1);
    if (!active) continue;
    const char* sb = smem + (kt & 1) * AT2_STAGE;
    const int k0 = kt * ATT_KV;
    // the tile's second 32-key block lies wholly beyond the sequence (its last tile, len % 64 in 1..32): every score of it
    // would be masked to -inf, p = 0 - the block is not multiplied, not exponentiated, not summed (+0 everywhere: same bits)
    const bool two = k0 + 32 < len;  // (wave-uniform)
    // ---- S^T = K Q^T
    f32x16 s[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      if (kb == 1 && !two) break;
#pragma unroll
      for (int r = 0; r < 16; ++r) s[kb][r] = 0.f;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        bf16x8 kf = *reinterpret_cast<const bf16x8*>(sb + k_off[kb][c]);
        s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[c], s[kb], 0, 0, 0);
      }
    }
    // ---- relative-position bias + key-padding mask, per 32-key block.  Four cases, wave-uniform:
    //   saturated right / left (one constant), interior (|j-i| <= maxd everywhere and all keys real:
    //   the table index is base + compile-time offset, no clamp, no mask), general.
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      if (kb == 1 && !two) break;
      const int c0 = k0 + kb * 32;
      const bool all_real = (c0 + 32 <= len);
      if (c0 - (wq0 + 31) >= maxd && all_real) {
        const float bb = tab[2 * maxd + TAB_PAD];
#pragma unroll
        for (int r = 0; r < 16; ++r) s[kb][r] += bb;
      } else if (c0 + 31 - wq0 <= -maxd && all_real) {
        const float bb = tab[TAB_PAD];
#pragma unroll
        for (int r = 0; r < 16; ++r) s[kb][r] += bb;
      } else if (all_real) {  // every offset of the block lies within maxd + 62 of zero: inside the padded table
        const float* tp = tab + (c0 - qi + maxd + TAB_PAD + 4 * hi);
#pragma unroll
        for (int r = 0; r < 16; ++r) s[kb][r] += tp[(r & 3) + 8 * (r >> 2)];
      } else if (c0 - (wq0 + 31) >= maxd) {  // the sequence's last block, saturated: one constant + the key mask
        const float bb = tab[2 * maxd + TAB_PAD];
        const int jn = len - c0 - 4 * hi;  // keys (r & 3) + 8 (r >> 2) < jn are real
#pragma unroll
        for (int r = 0; r < 16; ++r) s[kb][r] = ((r & 3) + 8 * (r >> 2) < jn) ? s[kb][r] + bb : -INFINITY;
      } else if (c0 + 31 - wq0 >= -maxd - 2) {  // ... every offset >= -maxd - 64, inside the padded table (always so here: a block with
                                                // masked keys ends beyond len > wq0): unclamped index + the key mask
        const float* tp = tab + (c0 - qi + maxd + TAB_PAD + 4 * hi);
        const int jn = len - c0 - 4 * hi;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[kb][r] = ((r & 3) + 8 * (r >> 2) < jn) ? s[kb][r] + tp[(r & 3) + 8 * (r >> 2)] : -INFINITY;
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int j = c0 + mfma32_row(r, hi);
          const int rel = min(max(j - qi, -maxd), maxd) + maxd + TAB_PAD;
          s[kb][r] = (j < len) ? s[kb][r] + tab[rel] : -INFINITY;
        }
      }
    }
    // ---- online softmax in the exp2 domain: p = 2^((s - m) * log2 e)
    float mx = s[0][0];
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[0][r]);
    if (two) {
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[1][r]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    const float LOG2E = 1.4426950408889634f;
    const float mneg = -m_new * LOG2E;
    float psum = 0.f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      if (kb == 1 && !two) break;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = __builtin_amdgcn_exp2f(fmaf(s[kb][r], LOG2E, mneg));
        s[kb][r] = p;
        psum += p;
      }
    }
    if (__any(m_new != m_run)) {  // rescale only when some row's running max moved (exact: alpha = 1 otherwise)
      const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * LOG2E);  // m_run = -inf first -> 0
      l_run *= alpha;
#pragma unroll
      for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[d][r] *= alpha;
    }
    l_run += psum;
    m_run = m_new;
    if constexpr (TRAIN && DROP) {
      {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int r = 0; r < 16 && (kb == 0 || two); r += 2) {  // keys r, r + 1 of a lane are neighbours (even, odd: k0 % 64 == 0)
            float m0, m1;
            drop_mul2(drop, drop_site, (uint32_t)(s0 + qi), ((uint32_t)h << 20) | (uint32_t)(k0 + kb * 32 + mfma32_row(r, hi)), m0, m1);
            s[kb][r] *= m0;
            s[kb][r + 1] *= m1;
          }
      }
    }
    // ---- O^T += V^T P^T over four 16-key slabs
#pragma unroll
    for (int sl = 0; sl < 4; ++sl) {
      if (sl == 2 && !two) break;
      const int kb = sl >> 1, sub = sl & 1;
      bf16x8 pf;
      {
        uint32_t pw[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) pw[e] = pack_bf2(s[kb][8 * sub + 2 * e], s[kb][8 * sub + 2 * e + 1]);
        uint4 t = make_uint4(pw[0], pw[1], pw[2], pw[3]);
        pf = *reinterpret_cast<bf16x8*>(&t);
      }
#pragma unroll
      for (int d = 0; d < 2; ++d) {
        const char* vp = sb + v_off0 + d * 4096 + sl * 16 * 64;
        v4s16 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s16*)(vp));
        v4s16 up = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s16*)(vp + 8 * 64));
        bf16x8 vf;
        vf[0] = lo[0]; vf[1] = lo[1]; vf[2] = lo[2]; vf[3] = lo[3];
        vf[4] = up[0]; vf[5] = up[1]; vf[6] = up[2]; vf[7] = up[3];
        o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, o[d], 0, 0, 0);
      }
    }
  }

  if (active && qi < len) {
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.f / l_tot;
    // training forward: the row's log-sum-exp in the exp2 domain, p = 2^(s log2 e - lse2) (rp_train.hip recomputes P)
    if (lse2 && hi == 0) lse2[(size_t)h * lse_ld + s0 + qi] = fmaf(m_run, 1.4426950408889634f, __log2f(l_tot));
    bf16_t* op = out + (size_t)(s0 + qi) * inner + h * 64;
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        uint2 v;
        v.x = pack_bf2(o[d][4 * g] * inv, o[d][4 * g + 1] * inv);
        v.y = pack_bf2(o[d][4 * g + 2] * inv, o[d][4 * g + 3] * inv);
        *reinterpret_cast<uint2*>(op + d * 32 + 8 * g + 4 * hi) = v;
      }
  }
}

// ------------------------------------------------------------------------------------------
// K2(final)+K9+K10: final RMSNorm, masked mean over the sequence's tokens, L2 normalise.
//   mean_t(w * x_t * rs_t) = w * mean_t(x_t * rs_t); e / max(||e||, 1e-12)  (model.py:108-114)
//   Two deterministic passes (no atomics, so results are bit-reproducible whatever the batch):
//   pool_partial_kernel: one workgroup per 128-token chunk of a sequence (the pass's list, worklist_kernel); wave w
//     takes tokens w, w+4, ...; rs_t from the last residual epilogue's statistics (as for every other RMSNorm);
//     per-lane partial column sums of x_t * rs_t in registers, combined through LDS, written to
//     partial[chunk_base(b) + c][D]   (chunk_base(b) = cu[b] / 128 + b);
//   pool_finish_kernel: one workgroup per sequence sums its chunks in order, applies w / len and the L2
//     normalisation.
//   Tried, both without gain: 64-token chunks (twice the workgroups, to balance the unequal chunks of the length mix
//   over the CUs: 0.129 -> 0.137 ms per pass); the finish done by whichever workgroup arrives last at a per-sequence
//   counter (one launch instead of two: the device-scope release fence that makes the other chunks' sums visible
//   across XCDs costs ~2 us per workgroup, serialised per XCD: 0.13 -> 0.45 ms per pass).
// ------------------------------------------------------------------------------------------

// NV = 16-byte pieces (8 features) per lane and plane covering a row (ceil(D / 512)): 3 for d_model 1472 / 1536,
// 4 up to 2048.  R token rows of a wave are in flight before the first is consumed (the rows are independent
// streams: rs comes from rowscale).  `chunk` = tokens per workgroup.  A sequence of ONE chunk is finished here (weight,
// 1 / len, L2 normalisation, output row): its column sums never travel through `partial`, and pool_finish_kernel skips it.
template <int NV, int R, bool LO8>
__global__ __launch_bounds__(256) void pool_partial_kernel(const bf16_t* __restrict__ xhi, const bf16_t* __restrict__ xlo,
                                                           const float* __restrict__ rs,
                                                           const int4* __restrict__ pwork,
                                                           float* __restrict__ partial, int D, int chunk,
                                                           const float* __restrict__ w, void* __restrict__ out, int out_bf16,
                                                           int fuse) {
  __shared__ float red[4][NV * 64 * 8];
  __shared__ float nrm[4];
  const int4 wk = pwork[blockIdx.x];
  const int s0 = wk.x, len = wk.y, c = wk.z, b = wk.w;
  if (len == 0) return;
  const int t0 = c * chunk;
  const int t1 = min(len, t0 + chunk);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nv = D >> 3;
  float acc[NV][8];
#pragma unroll
  for (int i = 0; i < NV; ++i)
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[i][e] = 0.f;
  // Tokens are accumulated in index order per wave (w, w+4, w+8, ...), whatever the unrolling.
  for (int t = t0 + wave; t < t1; t += 4 * R) {
    uint4 vh[R][NV];
    typename std::conditional<LO8, uint2, uint4>::type vl[R][NV];
    float r[R];
#pragma unroll
    for (int u = 0; u < R; ++u) {
      const int tu = t + 4 * u;
      const bool live = tu < t1;
      const size_t row = (size_t)(s0 + (live ? tu : t)) * D;
      const uint4* sh = reinterpret_cast<const uint4*>(xhi + row);
      r[u] = live ? rs[s0 + tu] : 0.f;
#pragma unroll
      for (int i = 0; i < NV; ++i) {  // clamped, unpredicated
        vh[u][i] = sh[min(lane + 64 * i, nv - 1)];
        if constexpr (LO8)
          vl[u][i] = reinterpret_cast<const uint2*>(reinterpret_cast<const uint8_t*>(xlo) + row)[min(lane + 64 * i, nv - 1)];
        else
          vl[u][i] = reinterpret_cast<const uint4*>(xlo + row)[min(lane + 64 * i, nv - 1)];
      }
    }
#pragma unroll
    for (int u = 0; u < R; ++u)
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const uint32_t h[4] = {vh[u][i].x, vh[u][i].y, vh[u][i].z, vh[u][i].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float x0, x1;
          if constexpr (LO8) {
            x24_decode2(h[e], (e < 2) ? vl[u][i].x : vl[u][i].y, e & 1, x0, x1);
          } else {
            const uint32_t l[4] = {vl[u][i].x, vl[u][i].y, vl[u][i].z, vl[u][i].w};
            x0 = __uint_as_float(h[e] << 16) + __uint_as_float(l[e] << 16);
            x1 = __uint_as_float(h[e] & 0xffff0000u) + __uint_as_float(l[e] & 0xffff0000u);
          }
          acc[i][2 * e] = fmaf(x0, r[u], acc[i][2 * e]);
          acc[i][2 * e + 1] = fmaf(x1, r[u], acc[i][2 * e + 1]);
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
  if (!fuse || len > chunk) {  // one chunk of several (or the training step, whose backward reads every row): to `partial`
    float* dst = partial + (size_t)(s0 / chunk + b + c) * D;
    for (int col = threadIdx.x; col < D; col += 256) dst[col] = (red[0][col] + red[1][col]) + (red[2][col] + red[3][col]);
    return;
  }
  // the whole sequence: pool_finish_kernel's arithmetic on the sums (0 + sum is sum: the same bits as through `partial`)
  const float inv_len = 1.f / (float)len;
  float vals[NV * 2];
  float part = 0.f;
  int cnt = 0;
  for (int col = threadIdx.x; col < D; col += 256) {
    const float sum = (red[0][col] + red[1][col]) + (red[2][col] + red[3][col]);
    const float e = sum * inv_len * w[col];
    vals[cnt++] = e;
    part += e * e;
  }
  part = wave_sum(part);
  if (lane == 0) nrm[wave] = part;
  __syncthreads();
  const float norm = sqrtf((nrm[0] + nrm[1]) + (nrm[2] + nrm[3]));
  const float sc = 1.f / fmaxf(norm, 1e-12f);
  cnt = 0;
  for (int col = threadIdx.x; col < D; col += 256) {
    const float e = vals[cnt++] * sc;
    if (out_bf16)
      reinterpret_cast<bf16_t*>(out)[(size_t)b * D + col] = f2bf(e);
    else
      reinterpret_cast<float*>(out)[(size_t)b * D + col] = e;
  }
}

static void launch_pool_partial(dim3 grid, hipStream_t stream, const bf16_t* xhi, const bf16_t* xlo, const float* rs,
                                const int4* pwork, float* partial, int D, int chunk, const float* w, void* out, int out_bf16,
                                int fuse, bool lo8) {
  if (lo8) {
    if (D <= 3 * 512)  // (eight rows in flight per wave instead of four: 71 vs 73 us, round 6 exp10 - not kept)
      hipLaunchKernelGGL((pool_partial_kernel<3, 4, true>), grid, dim3(256), 0, stream, xhi, xlo, rs, pwork, partial, D, chunk, w, out,
                         out_bf16, fuse);
    else
      hipLaunchKernelGGL((pool_partial_kernel<4, 4, true>), grid, dim3(256), 0, stream, xhi, xlo, rs, pwork, partial, D, chunk, w, out,
                         out_bf16, fuse);
  } else if (D <= 3 * 512) {
    hipLaunchKernelGGL((pool_partial_kernel<3, 4, false>), grid, dim3(256), 0, stream, xhi, xlo, rs, pwork, partial, D, chunk, w, out,
                       out_bf16, fuse);
  } else {
    hipLaunchKernelGGL((pool_partial_kernel<4, 4, false>), grid, dim3(256), 0, stream, xhi, xlo, rs, pwork, partial, D, chunk, w, out,
                       out_bf16, fuse);
  }
}

// Sequences of more than one chunk: one workgroup per sequence sums its chunks' column sums in chunk order - four chunks'
// rows requested before the first add (round 4's loop waited for one row at a time: 26 us for the 16 chunks of a
// 2048-token sequence) - then weight, 1 / len and the L2 normalisation.
static __global__ __launch_bounds__(256) void pool_finish_kernel(const float* __restrict__ partial,
                                                          const float* __restrict__ w,
                                                          const int32_t* __restrict__ cu,
                                                          void* __restrict__ out, int out_bf16, int D, int chunk,
                                                          int fuse) {
  __shared__ float nrm[4];
  const int b = blockIdx.x;
  const int s0 = cu[b], len = cu[b + 1] - s0;
  if (fuse && len <= chunk) return;  // finished by pool_partial_kernel
  const int nchunk = (len + chunk - 1) / chunk;
  const float* src = partial + (size_t)(s0 / chunk + b) * D;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float inv_len = 1.f / (float)len;
  constexpr int NC = RMS_MAX_V4;  // columns per thread: D <= 2048
  float sum[NC];
#pragma unroll
  for (int i = 0; i < NC; ++i) sum[i] = 0.f;
  for (int c0 = 0; c0 < nchunk; c0 += 4) {
    float v[4][NC];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float* row = src + (size_t)min(c0 + u, nchunk - 1) * D;
#pragma unroll
      for (int i = 0; i < NC; ++i) v[u][i] = row[min((int)threadIdx.x + 256 * i, D - 1)];  // clamped, unpredicated
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (c0 + u < nchunk) {
#pragma unroll
        for (int i = 0; i < NC; ++i) sum[i] += v[u][i];  // chunk order per column
      }
  }
  float vals[NC];
  float part = 0.f;
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    const int col = threadIdx.x + 256 * i;
    const float e = (col < D) ? sum[i] * inv_len * w[min(col, D - 1)] : 0.f;
    vals[i] = e;
    part += e * e;
  }
  part = wave_sum(part);
  if (lane == 0) nrm[wave] = part;
  __syncthreads();
  const float norm = sqrtf((nrm[0] + nrm[1]) + (nrm[2] + nrm[3]));
  const float sc = 1.f / fmaxf(norm, 1e-12f);
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    const int col = threadIdx.x + 256 * i;
    if (col < D) {
      const float e = vals[i] * sc;
      if (out_bf16)
        reinterpret_cast<bf16_t*>(out)[(size_t)b * D + col] = f2bf(e);
      else
        reinterpret_cast<float*>(out)[(size_t)b * D + col] = e;
    }
  }
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
struct LayerPacked {
  float* ln_attn;
  float* ln_ff;
  bf16_t* wqkv;  // [3*inner, D]
  bf16_t* wo;    // [D, inner]
  bf16_t* wi;    // [2F, D] gate/up interleaved by 32
  bf16_t* wo2;   // [D, F]
};

}  // namespace rp

struct RpEncoder {
  RpT5Config cfg;
  int inner;
  int maxd;
  float* embed;        // [V, D] f32
  rp::bf16_t* embed_hi = nullptr;   // the table in the inference pass's 24-bit form: bf16 plane [V, D],
  uint8_t* embed_lo = nullptr;  //   int8 extension plane [V, D],
  float* embed_ss = nullptr;    //   sum of squares of every row as stored [V]   (embed_table_x24)
  float* final_ln;     // [D]
  float* bias_tab;     // [H, 2*maxd+1]
  std::vector<rp::LayerPacked> layers;
  std::vector<void*> allocs;
};

namespace rp {
// (Re-)encode the embedding table into the 24-bit form: embed_kernel<true> over the vocabulary (token r = row r), the row
// statistic into embed_ss (slot 0 of a one-slot layout).  Launch-only; e->embed_* are allocated by rp_encoder_create.
inline void embed_table_x24(RpEncoder* e, hipStream_t stream) {
  const int V = e->cfg.vocab_size, D = e->cfg.d_model;
  hipLaunchKernelGGL(embed_kernel<true>, dim3((V + 3) / 4), dim3(256), 0, stream, (const int32_t*)nullptr, (const float*)e->embed,
                     e->embed_hi, reinterpret_cast<bf16_t*>(e->embed_lo), e->embed_ss, 1, V, V, D, V, (const int32_t*)nullptr);
}
}  // namespace rp
