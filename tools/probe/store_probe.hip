// DEV TOOL: what does a store-dominated stream (broadcast, bool → f32, the widening casts) lose against a plain fill?
// 4 GB of f32 written by shapes that differ in ONE thing each: block size, stores per lane, nontemporal or not, and a
// small dependent load (one bitmap dword per lane) in front of the stores.  Standalone:
//   hipcc -O3 --offload-arch=gfx950 tools/probe/store_probe.hip -o /tmp/store_probe && /tmp/store_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

// BLOCK threads, U stores of 16 B per lane (rows of BLOCK lanes), NT = nontemporal, LOAD = 0 none / 1 one dword of
// `bits` per store (8 lanes share it: the bool → f32 pattern) / 2 one nontemporal dword of a 1-byte-per-row column per
// store (the u8 → f32 pattern, 5 B/row) / 3 the same through a cached load
template <int BLOCK, int U, bool NT, int LOAD>
__global__ __launch_bounds__(BLOCK) void store_kernel(const uint32_t* bits, f32x4* out, uint64_t npacks) {
  const uint64_t p0 = (uint64_t)blockIdx.x * (BLOCK * U) + threadIdx.x;
#pragma unroll
  for (int u = 0; u < U; u++) {
    const uint64_t pk = p0 + (uint64_t)u * BLOCK;
    if (pk >= npacks) break;
    f32x4 r = {1.0f, 2.0f, 3.0f, 4.0f};
    if constexpr (LOAD == 1) {
      const uint32_t w = bits[pk >> 3] >> ((pk & 7) * 4);  // cached: 8 lanes and 4 waves share the line
      r = f32x4{(w & 1) ? 1.0f : 0.0f, (w & 2) ? 1.0f : 0.0f, (w & 4) ? 1.0f : 0.0f, (w & 8) ? 1.0f : 0.0f};
    }
    if constexpr (LOAD == 2) {
      const uint32_t w = __builtin_nontemporal_load(bits + pk);  // 4 bytes = 4 rows per pack
      r = f32x4{(float)(w & 255u), (float)((w >> 8) & 255u), (float)((w >> 16) & 255u), (float)(w >> 24)};
    }
    if constexpr (LOAD == 3) {
      const uint32_t w = bits[pk];  // the same 4 rows per pack through a cached load
      r = f32x4{(float)(w & 255u), (float)((w >> 8) & 255u), (float)((w >> 16) & 255u), (float)(w >> 24)};
    }
    if constexpr (LOAD == 4) {  // 16 lanes of the wave load 16 bytes each (the wave's 256 rows), 4 ds_bpermute spread the dwords
      typedef uint32_t u32x4_ __attribute__((ext_vector_type(4)));
      const uint32_t lane = threadIdx.x & 63u;
      u32x4_ v = {0, 0, 0, 0};
      if (lane < 16) v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_*>(bits) + (pk >> 6) * 16 + lane);
      const int src = (int)((lane >> 2) * 4);
      const uint32_t w0 = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)v.x), w1 = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)v.y);
      const uint32_t w2 = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)v.z), w3 = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)v.w);
      const uint32_t sel = lane & 3u;
      const uint32_t w = sel == 0 ? w0 : sel == 1 ? w1 : sel == 2 ? w2 : w3;
      r = f32x4{(float)(w & 255u), (float)((w >> 8) & 255u), (float)((w >> 16) & 255u), (float)(w >> 24)};
    }
    if constexpr (NT) __builtin_nontemporal_store(r, out + pk);
    else out[pk] = r;
  }
}

struct Row { const char* name; double ms; };

template <int BLOCK, int U, bool NT, int LOAD>
static int run(const char* name, const uint32_t* bits, f32x4* out, uint64_t npacks, std::vector<Row>& rows) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const uint64_t per = (uint64_t)BLOCK * U;
  const unsigned grid = (unsigned)((npacks + per - 1) / per);
  std::vector<float> ts;
  for (int rep = 0; rep < 12; rep++) {
    CK(hipEventRecord(e0, nullptr));
    hipLaunchKernelGGL((store_kernel<BLOCK, U, NT, LOAD>), dim3(grid), dim3(BLOCK), 0, nullptr, bits, out, npacks);
    CK(hipEventRecord(e1, nullptr));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (rep >= 3) ts.push_back(ms);
  }
  std::sort(ts.begin(), ts.end());
  rows.push_back({name, ts[ts.size() / 2]});
  return 0;
}

int main() {
  const uint64_t n = 1000000000ull / 4096 * 4096, npacks = n / 4;
  f32x4* out;
  uint32_t* bits;
  CK(hipMalloc(&out, n * 4));
  CK(hipMalloc(&bits, n + 4096));  // up to 1 byte per row
  CK(hipMemset(bits, 0x5a, n + 4096));
  std::vector<Row> rows;
#define RUN(B, U, NT, LD) if (run<B, U, NT, LD>("block " #B " x " #U " stores, nt=" #NT ", load=" #LD, bits, out, npacks, rows)) return 1;
  for (int round = 0; round < 1; round++) {
    RUN(256, 1, false, 0) RUN(256, 1, true, 0) RUN(64, 1, true, 0) RUN(64, 4, true, 0) RUN(64, 4, false, 0) RUN(256, 4, true, 0)
    RUN(64, 4, true, 1) RUN(64, 4, false, 1) RUN(256, 1, true, 1) RUN(256, 1, false, 1) RUN(64, 1, true, 1) RUN(64, 2, true, 1) RUN(128, 2, true, 1)
    RUN(64, 4, true, 2) RUN(256, 1, true, 2) RUN(256, 1, false, 2) RUN(64, 1, true, 2) RUN(64, 2, true, 2)
    RUN(256, 1, true, 4) RUN(128, 1, true, 4) RUN(64, 1, true, 4) RUN(256, 2, true, 4) RUN(64, 4, true, 4)
    RUN(256, 1, true, 3) RUN(128, 1, true, 3) RUN(128, 2, true, 3) RUN(64, 4, true, 3) RUN(256, 2, true, 3)
  }
  for (const Row& r : rows) printf("%-46s %.4f ms  %.3f of 8 TB/s (4 B/row written)\n", r.name, r.ms, 4.0 * n / r.ms / 8e9);
  return 0;
}
