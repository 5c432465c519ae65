// DEV TOOL: does the compare kernel's allocation lottery (0.84 ↔ 0.89 of the roof) come from how the
// driver maps a hipMalloc block, and do virtual-memory-management allocations (hipMemCreate in big physically contiguous
// chunks, mapped at aligned addresses) take the luck out of it?  An eq-like kernel (two 4 GB i32 columns → bitmap words,
// nontemporal) over buffers allocated (a) with hipMalloc, several times, keeping the earlier ones; (b) through
// hipMemAddressReserve / hipMemCreate / hipMemMap with chunk sizes from the granularity up to 1 GiB.
//   hipcc -O3 --offload-arch=gfx950 tools/probe/vmm_probe.hip -o /tmp/vmm_probe && /tmp/vmm_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ __launch_bounds__(256) void eq_kernel(const int* a, const int* b, uint64_t* out, uint64_t ntiles) {
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (uint64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const uint64_t e0 = t * 1024 + wave * 256;
    int xa[4], xb[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
      xa[r] = __builtin_nontemporal_load(a + e0 + r * 64 + lane);
      xb[r] = __builtin_nontemporal_load(b + e0 + r * 64 + lane);
    }
    uint64_t word = 0;
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const uint64_t m = __ballot(xa[r] == xb[r]);
      if (lane == (uint32_t)r) word = m;
    }
    if (lane < 4) __builtin_nontemporal_store(word, out + e0 / 64 + lane);
  }
}
__global__ void fill_kernel(int* p, uint64_t n, uint32_t seed) {
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) p[i] = (int)((i * 2654435761u + seed) >> 22);
}

static const uint64_t N = 1000000000ull / 4096 * 4096;  // 4e9-byte columns: they fit the 4 GiB slots with room for the colour

static int time_eq(const int* a, const int* b, uint64_t* out, double* frac) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  std::vector<float> ts;
  for (int rep = 0; rep < 10; rep++) {
    CK(hipEventRecord(e0, nullptr));
    hipLaunchKernelGGL(eq_kernel, dim3((unsigned)(N / 1024)), dim3(256), 0, nullptr, a, b, out, N / 1024);
    CK(hipEventRecord(e1, nullptr));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (rep >= 3) ts.push_back(ms);
  }
  std::sort(ts.begin(), ts.end());
  *frac = 8.125 * N / ts[ts.size() / 2] / 8e9 * 1e-3 * 1e3;
  CK(hipEventDestroy(e0));
  CK(hipEventDestroy(e1));
  return 0;
}

// one VMM "allocation": `bytes` of address space at `align`, backed by physical chunks of `chunk` bytes
static int vmm_alloc(size_t bytes, size_t chunk, size_t align, void** out, std::vector<hipMemGenericAllocationHandle_t>& handles) {
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = 0;
  const size_t total = (bytes + chunk - 1) / chunk * chunk;
  void* va = nullptr;
  CK(hipMemAddressReserve(&va, total, align, nullptr, 0));
  for (size_t off = 0; off < total; off += chunk) {
    hipMemGenericAllocationHandle_t h;
    CK(hipMemCreate(&h, chunk, &prop, 0));
    CK(hipMemMap(static_cast<char*>(va) + off, chunk, 0, h, 0));
    handles.push_back(h);
  }
  hipMemAccessDesc acc = {};
  acc.location.type = hipMemLocationTypeDevice;
  acc.location.id = 0;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  CK(hipMemSetAccess(va, total, &acc, 1));
  *out = va;
  return 0;
}

int main() {
  const size_t col = (size_t)4 << 30, bm = (size_t)128 << 20;
  // (a) hipMalloc, the lottery as the product sees it
  std::vector<void*> keep;
  for (int trial = 0; trial < 6; trial++) {
    char* blk;
    CK(hipMalloc(&blk, 2 * col + 2 * bm + (1 << 20)));
    keep.push_back(blk);
    int* a = (int*)blk;
    int* b = (int*)(blk + col + 8192);
    uint64_t* o = (uint64_t*)(blk + 2 * col + (2 << 20));
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, nullptr, a, N, 1u);
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, nullptr, b, N, 77u);
    double f;
    if (time_eq(a, b, o, &f)) return 1;
    printf("hipMalloc block %d at %p: eq %.3f of 8 TB/s\n", trial, (void*)blk, f);
  }
  for (void* p : keep) CK(hipFree(p));
  // (b) VMM chunks
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = 0;
  size_t gran_min = 0, gran_rec = 0;
  CK(hipMemGetAllocationGranularity(&gran_min, &prop, hipMemAllocationGranularityMinimum));
  CK(hipMemGetAllocationGranularity(&gran_rec, &prop, hipMemAllocationGranularityRecommended));
  printf("VMM granularity: minimum %zu, recommended %zu\n", gran_min, gran_rec);
  const size_t chunks[] = {(size_t)2 << 20, (size_t)32 << 20, (size_t)1 << 30, (size_t)9 << 30};
  for (size_t chunk : chunks) {
    if (chunk < gran_min) continue;
    for (int trial = 0; trial < 4; trial++) {
      std::vector<hipMemGenericAllocationHandle_t> handles;
      void* blk = nullptr;  // ONE reservation laid out like the hipMalloc block above: the distance a → b is the same
      if (vmm_alloc(2 * col + 2 * bm + (1 << 20), chunk, (size_t)1 << 30, &blk, handles)) return 1;
      int* a = (int*)blk;
      int* b = (int*)((char*)blk + col + 8192);
      uint64_t* o = (uint64_t*)((char*)blk + 2 * col + (2 << 20));
      hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, nullptr, a, N, 1u);
      hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, nullptr, b, N, 77u);
      double f;
      if (time_eq(a, b, o, &f)) return 1;
      printf("VMM chunk %4zu MiB trial %d: block %p: eq %.3f of 8 TB/s\n", chunk >> 20, trial, blk, f);
      // keep the mappings (like the lottery keeps its blocks): later trials land on other physical memory
    }
  }
  return 0;
}
