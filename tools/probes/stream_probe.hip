// What do the access patterns of the encode pass's memory-bound kernels reach with NOTHING else in the kernel?
// [70144, 1472] elements as a bf16 plane (2944-byte rows) + an int8 plane (1472-byte rows), one wave per row (the embedding
// kernel's and the pooling pass's shape), against flat grid-stride streams of the same bytes.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/stream_probe.hip -o tools/probes/stream_probe.bin && tools/probes/stream_probe.bin
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

__global__ __launch_bounds__(256) void store_rows(uint16_t* hi, uint8_t* lo, int T, int D) {  // embed's stores
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= T) return;
  uint4* dh = reinterpret_cast<uint4*>(hi + (size_t)row * D);
  uint2* dl = reinterpret_cast<uint2*>(lo + (size_t)row * D);
  for (int c = lane; c < (D >> 3); c += 64) {
    dh[c] = make_uint4(row, c, 3, 4);
    dl[c] = make_uint2(row, c);
  }
}
__global__ __launch_bounds__(256) void store_flat(uint4* p, size_t n16) {  // the same bytes as one flat 16-B stream
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) p[i] = make_uint4((uint32_t)i, 2, 3, 4);
}
__global__ __launch_bounds__(256) void read_rows(const uint16_t* hi, const uint8_t* lo, int T, int D, uint32_t* sink) {  // pooling's loads
  const int lane = threadIdx.x & 63, row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 16;  // 16 rows per wave, 4 in flight
  uint32_t acc = 0;
  for (int r = row0; r < min(row0 + 16, T); r += 4) {
    uint4 vh[4][3];
    uint2 vl[4][3];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int rr = min(r + u, T - 1);
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const int c = min(lane + 64 * i, (D >> 3) - 1);
        vh[u][i] = reinterpret_cast<const uint4*>(hi + (size_t)rr * D)[c];
        vl[u][i] = reinterpret_cast<const uint2*>(lo + (size_t)rr * D)[c];
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < 3; ++i) acc += vh[u][i].x ^ vh[u][i].y ^ vh[u][i].z ^ vh[u][i].w ^ vl[u][i].x ^ vl[u][i].y;
  }
  if (acc == 0x12345678u) sink[0] = acc;
}
__global__ __launch_bounds__(256) void read_flat(const uint4* p, size_t n16, uint32_t* sink) {
  uint32_t acc = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256 * 4) {
    uint4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = p[min(i + (size_t)u * gridDim.x * 256, n16 - 1)];
#pragma unroll
    for (int u = 0; u < 4; ++u) acc += v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

int main() {
  const int T = 70144, D = 1472;
  const size_t bytes = (size_t)T * D * 3;
  uint16_t* hi; uint8_t* lo; uint32_t* sink;
  hipMalloc(&hi, (size_t)T * D * 2 + (size_t)T * D); lo = reinterpret_cast<uint8_t*>(hi) + (size_t)T * D * 2;
  hipMalloc(&sink, 64);
  hipMemset(hi, 1, bytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto time = [&](const char* name, auto launch) {
    for (int i = 0; i < 3; ++i) launch();
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 20;
    printf("%-64s %7.1f us  %5.2f TB/s  (%.2f of 8 TB/s)\n", name, ms * 1e3, bytes / ms / 1e9, bytes / ms / 1e9 / 8.0);
  };
  time("stores, one wave per row: 16 B (bf16 plane) + 8 B (int8 plane)", [&] { hipLaunchKernelGGL(store_rows, dim3((T + 3) / 4), dim3(256), 0, 0, hi, lo, T, D); });
  for (int g : {1024, 4096, 16384})
    time(g == 1024 ? "stores, flat 16-B stream, 1024 workgroups" : g == 4096 ? "stores, flat 16-B stream, 4096 workgroups" : "stores, flat 16-B stream, 16384 workgroups",
         [&] { hipLaunchKernelGGL(store_flat, dim3(g), dim3(256), 0, 0, reinterpret_cast<uint4*>(hi), bytes / 16); });
  time("loads, one wave per 16 rows, 4 rows in flight: 16 B + 8 B", [&] { hipLaunchKernelGGL(read_rows, dim3((T / 16 + 3) / 4), dim3(256), 0, 0, hi, lo, T, D, sink); });
  for (int g : {1024, 4096})
    time(g == 1024 ? "loads, flat 16-B stream, 4 in flight, 1024 workgroups" : "loads, flat 16-B stream, 4 in flight, 4096 workgroups",
         [&] { hipLaunchKernelGGL(read_flat, dim3(g), dim3(256), 0, 0, reinterpret_cast<const uint4*>(hi), bytes / 16, sink); });
  return 0;
}
