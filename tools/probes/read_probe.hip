// How fast can the chip READ a [T, 1472] two-plane bf16 residual stream (2 x 206 MB) when the result is a handful of
// column sums (the pooling pass's access pattern)?  Variants: flat grid-stride (the upper bound for a coalesced read),
// and row-structured forms with different numbers of rows in flight per wave.  Prints TB/s per variant.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/read_probe.hip -o gpurun_out/read_probe && gpurun_out/read_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__device__ __forceinline__ float eat(uint4 v) {
  return __uint_as_float(v.x << 16) + __uint_as_float(v.y << 16) + __uint_as_float(v.z << 16) + __uint_as_float(v.w << 16);
}

template <int UNROLL>
__global__ __launch_bounds__(256) void flat_read(const uint4* __restrict__ a, size_t n, float* out) {
  float acc = 0.f;
  const size_t stride = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + (UNROLL - 1) * stride < n; i += UNROLL * stride) {
    uint4 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) v[u] = a[i + u * stride];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) acc += eat(v[u]);
  }
  for (; i < n; i += stride) acc += eat(a[i]);
  if (acc == 12345.678f) out[0] = acc;
}

// each workgroup owns CHUNK consecutive rows of both planes; wave w takes rows w, w+4, ...; R rows in flight per wave
template <int R, int CHUNK>
__global__ __launch_bounds__(256) void row_read(const uint16_t* __restrict__ hi, const uint16_t* __restrict__ lo, int T,
                                                int D, float* out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int t0 = blockIdx.x * CHUNK, t1 = min(T, t0 + CHUNK);
  const int nv = D >> 3;
  float acc[3] = {0.f, 0.f, 0.f};
  for (int t = t0 + wave; t < t1; t += 4 * R) {
    uint4 vh[R][3], vl[R][3];
#pragma unroll
    for (int u = 0; u < R; ++u) {
      const int tu = min(t + 4 * u, t1 - 1);
      const uint4* sh = reinterpret_cast<const uint4*>(hi + (size_t)tu * D);
      const uint4* sl = reinterpret_cast<const uint4*>(lo + (size_t)tu * D);
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        vh[u][i] = sh[min(lane + 64 * i, nv - 1)];
        vl[u][i] = sl[min(lane + 64 * i, nv - 1)];
      }
    }
#pragma unroll
    for (int u = 0; u < R; ++u)
#pragma unroll
      for (int i = 0; i < 3; ++i) acc[i] += eat(vh[u][i]) + eat(vl[u][i]);
  }
  if (acc[0] + acc[1] + acc[2] == 12345.678f) out[0] = acc[0];
}

int main() {
  const int T = 70144, D = 1472;
  uint16_t *hi, *lo; float* out;
  hipMalloc(&hi, (size_t)T * D * 2 * 2); lo = hi + (size_t)T * D; hipMalloc(&out, 64);
  hipMemset(hi, 0, (size_t)T * D * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const double bytes = (double)T * D * 4.0;
  auto time = [&](const char* name, auto launch) {
    for (int i = 0; i < 3; ++i) launch();
    hipEventRecord(e0);
    const int it = 20;
    for (int i = 0; i < it; ++i) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= it;
    printf("%-58s %.3f ms  %.2f TB/s\n", name, ms, bytes / ms / 1e9);
  };
  const size_t n16 = (size_t)T * D * 4 / 16;
  time("flat grid-stride, 2048 workgroups, 4 loads in flight", [&] { hipLaunchKernelGGL(flat_read<4>, dim3(2048), dim3(256), 0, 0, (const uint4*)hi, n16, out); });
  time("flat grid-stride, 2048 workgroups, 8 loads in flight", [&] { hipLaunchKernelGGL(flat_read<8>, dim3(2048), dim3(256), 0, 0, (const uint4*)hi, n16, out); });
  time("flat grid-stride, 4096 workgroups, 8 loads in flight", [&] { hipLaunchKernelGGL(flat_read<8>, dim3(4096), dim3(256), 0, 0, (const uint4*)hi, n16, out); });
  time("flat grid-stride, 1024 workgroups, 16 loads in flight", [&] { hipLaunchKernelGGL(flat_read<16>, dim3(1024), dim3(256), 0, 0, (const uint4*)hi, n16, out); });
  time("rows: 128-row chunks, 4 rows in flight per wave (the pass)", [&] { hipLaunchKernelGGL((row_read<4, 128>), dim3((T + 127) / 128), dim3(256), 0, 0, hi, lo, T, D, out); });
  time("rows: 128-row chunks, 2 rows in flight per wave", [&] { hipLaunchKernelGGL((row_read<2, 128>), dim3((T + 127) / 128), dim3(256), 0, 0, hi, lo, T, D, out); });
  time("rows: 64-row chunks, 4 rows in flight per wave", [&] { hipLaunchKernelGGL((row_read<4, 64>), dim3((T + 63) / 64), dim3(256), 0, 0, hi, lo, T, D, out); });
  time("rows: 32-row chunks, 4 rows in flight per wave", [&] { hipLaunchKernelGGL((row_read<4, 32>), dim3((T + 31) / 32), dim3(256), 0, 0, hi, lo, T, D, out); });
  time("rows: 32-row chunks, 8 rows in flight per wave", [&] { hipLaunchKernelGGL((row_read<8, 32>), dim3((T + 31) / 32), dim3(256), 0, 0, hi, lo, T, D, out); });
  time("rows: 16-row chunks, 4 rows in flight per wave", [&] { hipLaunchKernelGGL((row_read<4, 16>), dim3((T + 15) / 16), dim3(256), 0, 0, hi, lo, T, D, out); });
  return 0;
}
