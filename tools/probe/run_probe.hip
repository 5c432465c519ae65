// DEV TOOL: where should a partition pass pay for its 64-byte runs — on the WRITE side (today's pair pipeline: tile t stores run
// (t, b) into region b's range, so the next pass streams) or on the READ side (tile-major: every tile stores its sorted rows as one
// coalesced piece, the next pass collects run (t, b) from all tiles t)?  2^28 pairs of 8 bytes = 16384 tiles x 2048 runs x 8 pairs.
//   hipcc -O3 --offload-arch=gfx950 tools/probe/run_probe.hip -o /tmp/run_probe && /tmp/run_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr uint32_t T = 16384, B = 2048, RUN = 8;  // tiles, regions, pairs per run
constexpr uint64_t N = (uint64_t)T * B * RUN;

__device__ __forceinline__ uint32_t xcd_contig(uint32_t nblocks) {  // block j -> item (j % 8) * per + j / 8
  const uint32_t per = nblocks / 8;
  return (blockIdx.x % 8) * per + blockIdx.x / 8;
}

// streaming read / write of the whole array (16-byte accesses, 16 Ki pairs per block)
__global__ __launch_bounds__(1024) void stream_read(const u32x4* p, uint32_t* sink) {
  const uint64_t base = (uint64_t)blockIdx.x * 8192;
  uint32_t acc = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const u32x4 v = __builtin_nontemporal_load(p + base + k * 1024 + threadIdx.x);
    acc ^= v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x12345678u) sink[blockIdx.x] = acc;
}
__global__ __launch_bounds__(1024) void stream_write(u32x4* p) {
  const uint64_t base = (uint64_t)blockIdx.x * 8192;
#pragma unroll
  for (int k = 0; k < 8; k++) __builtin_nontemporal_store(u32x4{1u, 2u, 3u, threadIdx.x}, p + base + k * 1024 + threadIdx.x);
}

// READ side: block (region b, chunk c of 2048 tiles) collects run b of its tiles from the tile-major array; 8 lanes per run.
// ORDER 0: region-major blocks, XCD-contiguous (each XCD walks its eighth of the regions, chunk by chunk: few regions in flight);
// ORDER 1: chunk-major inside an XCD's eighth of the regions (the neighbour run (t, b+1) is read by the NEXT block).
// SHIFT: runs start (hash(t) % 8) pairs late — unaligned to their 64 bytes like real runs.
template <int ORDER, bool SHIFT>
__global__ __launch_bounds__(1024) void run_read(const u32x2* p, uint32_t* sink) {
  const uint32_t item = xcd_contig(gridDim.x);  // 0 .. B * 8
  uint32_t b, c;
  if (ORDER == 0) {
    b = item / 8;
    c = item % 8;
  } else {
    const uint32_t per_xcd = B / 8, x = item / (per_xcd * 8), r = item % (per_xcd * 8);
    b = x * per_xcd + r % per_xcd;
    c = r / per_xcd;
  }
  const uint32_t g = threadIdx.x / 8, l = threadIdx.x % 8;
  uint32_t acc = 0;
#pragma unroll
  for (int k = 0; k < 16; k++) {
    const uint32_t t = c * 2048 + k * 128 + g;
    const uint32_t sh = SHIFT ? ((t * 2654435761u) >> 29) : 0u;
    uint64_t at = (uint64_t)t * (B * RUN) + (uint64_t)b * RUN + sh + l;
    if (at >= N) at -= RUN;
    const u32x2 v = __builtin_nontemporal_load(p + at);
    acc ^= v.x ^ v.y;
  }
  if (acc == 0x12345678u) sink[blockIdx.x] = acc;
}

// WRITE side (today): block = tile t stores its 2048 runs into the region-major array: run (t, b) at (b * T + t) * RUN
template <bool SHIFT>
__global__ __launch_bounds__(1024) void run_write(u32x2* p) {
  const uint32_t t = xcd_contig(gridDim.x);
  const uint32_t g = threadIdx.x / 8, l = threadIdx.x % 8;
#pragma unroll
  for (int k = 0; k < 16; k++) {
    const uint32_t b = k * 128 + g;
    const uint32_t sh = SHIFT ? ((b * 2654435761u) >> 29) : 0u;
    uint64_t at = ((uint64_t)b * T + t) * RUN + sh + l;
    if (at >= N) at -= RUN;
    p[at] = u32x2{t, b};
  }
}

int main() {
  u32x2* p = nullptr;
  uint32_t* sink = nullptr;
  CK(hipMalloc(&p, N * 8));
  CK(hipMalloc(&sink, 1 << 20));
  CK(hipMemset(p, 1, N * 8));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  auto timeit = [&](const char* name, auto launch) -> int {
    std::vector<float> ts;
    for (int i = 0; i < 7; i++) {
      CK(hipEventRecord(e0));
      launch();
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms = 0;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (i >= 2) ts.push_back(ms);
    }
    std::sort(ts.begin(), ts.end());
    printf("%-64s %.4f ms  %.2f TB/s on 8 B/pair\n", name, ts[ts.size() / 2], (double)N * 8 / ts[ts.size() / 2] / 1e9);
    return 0;
  };
  timeit("stream read (16 B / lane)", [&] { hipLaunchKernelGGL(stream_read, dim3(N / 16384), dim3(1024), 0, 0, (const u32x4*)p, sink); });
  timeit("stream write", [&] { hipLaunchKernelGGL(stream_write, dim3(N / 16384), dim3(1024), 0, 0, (u32x4*)p); });
  timeit("READ side: 64-B runs, region-major blocks, aligned", [&] { hipLaunchKernelGGL((run_read<0, false>), dim3(B * 8), dim3(1024), 0, 0, p, sink); });
  timeit("READ side: 64-B runs, region-major blocks, unaligned", [&] { hipLaunchKernelGGL((run_read<0, true>), dim3(B * 8), dim3(1024), 0, 0, p, sink); });
  timeit("READ side: 64-B runs, chunk-major blocks, aligned", [&] { hipLaunchKernelGGL((run_read<1, false>), dim3(B * 8), dim3(1024), 0, 0, p, sink); });
  timeit("READ side: 64-B runs, chunk-major blocks, unaligned", [&] { hipLaunchKernelGGL((run_read<1, true>), dim3(B * 8), dim3(1024), 0, 0, p, sink); });
  timeit("WRITE side: 64-B runs, tile blocks XCD-contiguous, aligned", [&] { hipLaunchKernelGGL((run_write<false>), dim3(T), dim3(1024), 0, 0, p); });
  timeit("WRITE side: 64-B runs, tile blocks XCD-contiguous, unaligned", [&] { hipLaunchKernelGGL((run_write<true>), dim3(T), dim3(1024), 0, 0, p); });
  return 0;
}
