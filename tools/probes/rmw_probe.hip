// How fast can the chip read-modify-write an fp32 [T, 1472] matrix in tile-shaped pieces?
// Each workgroup owns a (TOK tokens x FEAT features) tile, as a GEMM epilogue would, and updates it
// CHUNK bytes of a row at a time (x += 1; also writes a bf16 copy).  Prints TB/s per variant.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/rmw_probe.hip -o gpurun_out/rmw_probe && gpurun_out/rmw_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

template <int CHUNK_LANES>  // lanes (16 B each) covering one row segment: 8 -> 128 B, 16 -> 256 B, 32 -> 512 B
__global__ __launch_bounds__(256) void rmw(float* __restrict__ x, uint16_t* __restrict__ xb, int T, int D, int tile_tok,
                                           int tile_feat, int tiles_f) {
  const int tf = blockIdx.x % tiles_f, tt = blockIdx.x / tiles_f;
  const int f0 = tf * tile_feat, t0 = tt * tile_tok;
  const int lane_in = threadIdx.x % CHUNK_LANES, row_in = threadIdx.x / CHUNK_LANES;
  constexpr int ROWS = 256 / CHUNK_LANES;
  for (int fc = 0; fc < tile_feat; fc += CHUNK_LANES * 4) {
    const int f = f0 + fc + lane_in * 4;
    float4 v[8];
    for (int tb = 0; tb < tile_tok; tb += ROWS * 8) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int t = t0 + tb + u * ROWS + row_in;
        v[u] = (t < T && f < D) ? *reinterpret_cast<const float4*>(x + (size_t)t * D + f) : make_float4(0, 0, 0, 0);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int t = t0 + tb + u * ROWS + row_in;
        if (t < T && f < D) {
          float4 w = v[u];
          w.x += 1.f; w.y += 1.f; w.z += 1.f; w.w += 1.f;
          *reinterpret_cast<float4*>(x + (size_t)t * D + f) = w;
          uint2 o;
          o.x = (__float_as_uint(w.x) >> 16) | (__float_as_uint(w.y) & 0xffff0000u);
          o.y = (__float_as_uint(w.z) >> 16) | (__float_as_uint(w.w) & 0xffff0000u);
          *reinterpret_cast<uint2*>(xb + (size_t)t * D + f) = o;
        }
      }
    }
  }
}

// The same update on a residual stream kept as two bf16 planes (hi = bf16(x), lo = bf16(x - hi)): 8 bytes per element
// instead of 10, hi is the next GEMM's operand.  SEG_LANES lanes x 16 B (8 bf16) cover one row segment of a plane.
template <int SEG_LANES>
__global__ __launch_bounds__(256) void rmw_hilo(uint16_t* __restrict__ hi, uint16_t* __restrict__ lo, int T, int D,
                                                int tile_tok, int tile_feat, int tiles_f) {
  const int tf = blockIdx.x % tiles_f, tt = blockIdx.x / tiles_f;
  const int f0 = tf * tile_feat, t0 = tt * tile_tok;
  const int lane_in = threadIdx.x % SEG_LANES, row_in = threadIdx.x / SEG_LANES;
  constexpr int ROWS = 256 / SEG_LANES;
  for (int fc = 0; fc < tile_feat; fc += SEG_LANES * 8) {
    const int f = f0 + fc + lane_in * 8;
    uint4 vh[8], vl[8];
    for (int tb = 0; tb < tile_tok; tb += ROWS * 8) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int t = t0 + tb + u * ROWS + row_in;
        const bool ok = t < T && f < D;
        vh[u] = ok ? *reinterpret_cast<const uint4*>(hi + (size_t)t * D + f) : make_uint4(0, 0, 0, 0);
        vl[u] = ok ? *reinterpret_cast<const uint4*>(lo + (size_t)t * D + f) : make_uint4(0, 0, 0, 0);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int t = t0 + tb + u * ROWS + row_in;
        if (t < T && f < D) {
          const uint32_t h[4] = {vh[u].x, vh[u].y, vh[u].z, vh[u].w}, l[4] = {vl[u].x, vl[u].y, vl[u].z, vl[u].w};
          uint32_t oh[4], ol[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float a = __uint_as_float(h[e] << 16) + __uint_as_float(l[e] << 16) + 1.f;
            float b = __uint_as_float(h[e] & 0xffff0000u) + __uint_as_float(l[e] & 0xffff0000u) + 1.f;
            const uint32_t ah = __float_as_uint(a) & 0xffff0000u, bh = __float_as_uint(b) & 0xffff0000u;
            oh[e] = (ah >> 16) | bh;
            ol[e] = (__float_as_uint(a - __uint_as_float(ah)) >> 16) | (__float_as_uint(b - __uint_as_float(bh)) & 0xffff0000u);
          }
          *reinterpret_cast<uint4*>(hi + (size_t)t * D + f) = make_uint4(oh[0], oh[1], oh[2], oh[3]);
          *reinterpret_cast<uint4*>(lo + (size_t)t * D + f) = make_uint4(ol[0], ol[1], ol[2], ol[3]);
        }
      }
    }
  }
}

// hi = bf16(x), lo = the next 8 mantissa-extension bits as int8 (x = hi + q * ulp(hi) / 128): 6 bytes per element moved.
// SEG_LANES lanes cover one row segment: 16 B of hi + 8 B of lo per lane (8 features).
template <int SEG_LANES>
__global__ __launch_bounds__(256) void rmw_hilo8(uint16_t* __restrict__ hi, uint16_t* __restrict__ lo16, int T, int D,
                                                 int tile_tok, int tile_feat, int tiles_f) {
  int8_t* lo = reinterpret_cast<int8_t*>(lo16);
  const int tf = blockIdx.x % tiles_f, tt = blockIdx.x / tiles_f;
  const int f0 = tf * tile_feat, t0 = tt * tile_tok;
  const int lane_in = threadIdx.x % SEG_LANES, row_in = threadIdx.x / SEG_LANES;
  constexpr int ROWS = 256 / SEG_LANES;
  for (int fc = 0; fc < tile_feat; fc += SEG_LANES * 8) {
    const int f = f0 + fc + lane_in * 8;
    uint4 vh[8];
    uint2 vl[8];
    for (int tb = 0; tb < tile_tok; tb += ROWS * 8) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int t = t0 + tb + u * ROWS + row_in;
        const bool ok = t < T && f < D;
        vh[u] = ok ? *reinterpret_cast<const uint4*>(hi + (size_t)t * D + f) : make_uint4(0, 0, 0, 0);
        vl[u] = ok ? *reinterpret_cast<const uint2*>(lo + (size_t)t * D + f) : make_uint2(0, 0);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int t = t0 + tb + u * ROWS + row_in;
        if (t < T && f < D) {
          const uint32_t h[4] = {vh[u].x, vh[u].y, vh[u].z, vh[u].w};
          const uint32_t l[2] = {vl[u].x, vl[u].y};
          uint32_t oh[4], ol[2] = {0u, 0u};
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const uint32_t hb = (e & 1) ? (h[e >> 1] & 0xffff0000u) : (h[e >> 1] << 16);
            const int q = (int)(int8_t)(l[e >> 2] >> (8 * (e & 3)));
            uint32_t ex = (hb >> 23) & 0xffu;
            ex = ex < 20u ? 20u : ex;
            const float sc = __uint_as_float((ex - 14u) << 23);
            const float a = __uint_as_float(hb) + (float)q * sc + 1.f;
            const uint32_t ah = __float_as_uint(a) & 0xffff0000u;  // (truncation: a probe of the traffic, not of the rounding)
            uint32_t ex2 = (ah >> 23) & 0xffu;
            ex2 = ex2 < 20u ? 20u : ex2;
            const float inv = __uint_as_float((254u - (ex2 - 14u)) << 23);
            int q2 = (int)rintf((a - __uint_as_float(ah)) * inv);
            q2 = q2 > 127 ? 127 : (q2 < -127 ? -127 : q2);
            if (e & 1) oh[e >> 1] |= ah; else oh[e >> 1] = ah >> 16;
            ol[e >> 2] |= ((uint32_t)(q2 & 0xff)) << (8 * (e & 3));
          }
          *reinterpret_cast<uint4*>(hi + (size_t)t * D + f) = make_uint4(oh[0], oh[1], oh[2], oh[3]);
          *reinterpret_cast<uint2*>(lo + (size_t)t * D + f) = make_uint2(ol[0], ol[1]);
        }
      }
    }
  }
}

int main() {
  const int T = 70144, D = 1472;
  float* x; uint16_t* xb;
  hipMalloc(&x, (size_t)T * D * 4); hipMalloc(&xb, (size_t)T * D * 2);
  hipMemset(x, 0, (size_t)T * D * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const double bytes = (double)T * D * 10.0;
  auto run = [&](const char* name, auto kern, int tile_tok, int tile_feat) {
    const int tiles_f = (D + tile_feat - 1) / tile_feat, tiles_t = (T + tile_tok - 1) / tile_tok;
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(tiles_f * tiles_t), dim3(256), 0, 0, x, xb, T, D, tile_tok, tile_feat, tiles_f);
    hipEventRecord(e0);
    const int it = 10;
    for (int i = 0; i < it; ++i) hipLaunchKernelGGL(kern, dim3(tiles_f * tiles_t), dim3(256), 0, 0, x, xb, T, D, tile_tok, tile_feat, tiles_f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= it;
    printf("%-34s tile %3d tok x %4d feat: %.3f ms  %.2f TB/s\n", name, tile_tok, tile_feat, ms, bytes / ms / 1e9);
  };
  run("128-B row segments", rmw<8>, 128, 128);
  run("256-B row segments", rmw<16>, 128, 128);
  run("512-B row segments", rmw<32>, 128, 128);
  run("128-B row segments", rmw<8>, 256, 256);
  run("256-B row segments", rmw<16>, 256, 256);
  run("512-B row segments", rmw<32>, 256, 256);
  run("512-B segments, full rows", rmw<32>, 64, 1472);
  {
    uint16_t* lo; hipMalloc(&lo, (size_t)T * D * 2); hipMemset(lo, 0, (size_t)T * D * 2);
    const double b8 = (double)T * D * 8.0;
    auto run2 = [&](const char* name, auto kern, int tile_tok, int tile_feat) {
      const int tiles_f = (D + tile_feat - 1) / tile_feat, tiles_t = (T + tile_tok - 1) / tile_tok;
      for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(tiles_f * tiles_t), dim3(256), 0, 0, xb, lo, T, D, tile_tok, tile_feat, tiles_f);
      hipEventRecord(e0);
      const int it = 10;
      for (int i = 0; i < it; ++i) hipLaunchKernelGGL(kern, dim3(tiles_f * tiles_t), dim3(256), 0, 0, xb, lo, T, D, tile_tok, tile_feat, tiles_f);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); ms /= it;
      printf("%-34s tile %3d tok x %4d feat: %.3f ms  %.2f TB/s (8 B per element; fp32 form at this time: %.2f TB/s-equivalent)\n", name, tile_tok,
             tile_feat, ms, b8 / ms / 1e9, bytes / ms / 1e9);
    };
    run2("hi/lo planes, 128-B segments", rmw_hilo<8>, 256, 256);
    run2("hi/lo planes, 256-B segments", rmw_hilo<16>, 256, 256);
    run2("hi/lo planes, 256-B segments", rmw_hilo<16>, 128, 128);
    run2("hi/lo planes, full rows", rmw_hilo<32>, 64, 1472);
    run2("hi bf16 + lo int8, 128-B hi segments", rmw_hilo8<8>, 256, 256);
    run2("hi bf16 + lo int8, 256-B hi segments", rmw_hilo8<16>, 256, 256);
    run2("hi bf16 + lo int8, 256-B hi segments", rmw_hilo8<16>, 128, 128);
    run2("hi/lo planes, 128-B segments (again)", rmw_hilo<8>, 256, 256);
  }
  return 0;
}
