// DEV TOOL (round 5): does the MEMORY TYPE of an allocation, or the cache-policy bits of the access, change what a random
// 4-byte gather / scatter costs?  A direct take sits on a "line-fetch roof" (131 B of HBM reads per 4-byte row through a
// default hipMalloc block, docs/experiments.md §4): gfx950's L2 can issue 32- / 64- / 128-byte fabric reads
// (TCC_EA0_RDREQ_32B / _64B / _128B), so the question is which allocation type / policy makes it ask for less.
//   allocation: hipMalloc | hipExtMallocWithFlags(Uncached | Finegrained | Contiguous)
//   access    : global_load_dword / global_store_dword with every {sc0, sc1, nt} combination (inline asm)
// Also: a plain 16-byte-per-lane copy through each allocation type (what an uncached column would cost the streaming kernels).
//   hipcc -O3 --offload-arch=gfx950 tools/probe/mtype_probe.hip -o tools/probe/mtype_probe
//   tools/probe/mtype_probe [log2_rows=28] [log2_source=28] [what=gsc] [reps=5]
// Kernel names carry the variant (template arguments), so a rocprofv3 --pmc pass over this binary attributes counters per variant.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#define POLICIES(X) X(0, "") X(1, "nt") X(2, "sc0") X(3, "sc1") X(4, "sc0 sc1") X(5, "sc0 nt") X(6, "sc1 nt") X(7, "sc0 sc1 nt")
static const char* kPolicy[8] = {"plain", "nt", "sc0", "sc1", "sc0 sc1", "sc0 nt", "sc1 nt", "sc0 sc1 nt"};

__device__ __forceinline__ uint32_t mix(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return (uint32_t)((x ^ (x >> 31)) >> 16);
}

__global__ void fill_idx(uint32_t* idx, uint64_t n, uint32_t mask, uint64_t seed) {
  for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
    idx[i] = mix(i ^ seed) & mask;
}
__global__ void fill_iota(uint32_t* v, uint64_t n) {
  for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
    v[i] = (uint32_t)i * 2654435761u;
}

template <int LD>
__device__ __forceinline__ uint32_t ld32(const uint32_t* p) {
  uint32_t r;
#define LD_CASE(K, MOD) if constexpr (LD == K) asm volatile("global_load_dword %0, %1, off " MOD : "=v"(r) : "v"(p) : "memory");
  POLICIES(LD_CASE)
  return r;
}
template <int ST>
__device__ __forceinline__ void st32(uint32_t* p, uint32_t v) {
#define ST_CASE(K, MOD) if constexpr (ST == K) asm volatile("global_store_dword %0, %1, off " MOD : : "v"(p), "v"(v) : "memory");
  POLICIES(ST_CASE)
}

// out[i] = values[idx[i]]: 8 gathers in flight per lane, 256-thread blocks, index / output streams as nt 16-byte accesses.
template <int LD, int MT>
__global__ __launch_bounds__(256) void gather_k(const uint32_t* values, const uint32_t* idx, uint32_t* out) {
  const uint64_t p0 = (uint64_t)blockIdx.x * 512 + threadIdx.x;
  const u32x4 a = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(idx) + p0);
  const u32x4 b = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(idx) + p0 + 256);
  u32x4 r, s;
  r.x = ld32<LD>(values + a.x); r.y = ld32<LD>(values + a.y); r.z = ld32<LD>(values + a.z); r.w = ld32<LD>(values + a.w);
  s.x = ld32<LD>(values + b.x); s.y = ld32<LD>(values + b.y); s.z = ld32<LD>(values + b.z); s.w = ld32<LD>(values + b.w);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_nontemporal_store(r, reinterpret_cast<u32x4*>(out) + p0);
  __builtin_nontemporal_store(s, reinterpret_cast<u32x4*>(out) + p0 + 256);
}

// dst[idx[i]] = i (a scatter of 4-byte values; duplicates race, nobody reads the result)
template <int ST, int MT>
__global__ __launch_bounds__(256) void scatter_k(uint32_t* dst, const uint32_t* idx) {
  const uint64_t p0 = (uint64_t)blockIdx.x * 512 + threadIdx.x;
  const u32x4 a = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(idx) + p0);
  const u32x4 b = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(idx) + p0 + 256);
  const uint32_t v = (uint32_t)p0;
  st32<ST>(dst + a.x, v); st32<ST>(dst + a.y, v); st32<ST>(dst + a.z, v); st32<ST>(dst + a.w, v);
  st32<ST>(dst + b.x, v); st32<ST>(dst + b.y, v); st32<ST>(dst + b.z, v); st32<ST>(dst + b.w, v);
}

// plain copy, one wave per block, one 16-byte pack per lane (the product's streaming shape), nt both ways
template <int MTS, int MTD>
__global__ __launch_bounds__(64) void copy_k(const u32x4* a, u32x4* o) {
  const uint64_t i = (uint64_t)blockIdx.x * 64 + threadIdx.x;
  __builtin_nontemporal_store(__builtin_nontemporal_load(a + i), o + i);
}

static const char* kMem[4] = {"hipMalloc", "uncached", "finegrained", "contiguous"};
static void* alloc_mt(int mt, size_t bytes) {
  void* p = nullptr;
  hipError_t e;
  if (mt == 0) e = hipMalloc(&p, bytes);
  else e = hipExtMallocWithFlags(&p, bytes, mt == 1 ? hipDeviceMallocUncached : mt == 2 ? hipDeviceMallocFinegrained : hipDeviceMallocContiguous);
  if (e != hipSuccess) { printf("  allocation of %zu bytes as %s failed: %s\n", bytes, kMem[mt], hipGetErrorString(e)); (void)hipGetLastError(); return nullptr; }
  return p;
}

template <class F>
static double time_ms(F&& launch, int reps) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<float> ts;
  for (int r = 0; r < reps + 2; r++) {
    CK(hipEventRecord(e0, nullptr));
    launch();
    CK(hipEventRecord(e1, nullptr));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (r >= 2) ts.push_back(ms);
  }
  CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
  std::sort(ts.begin(), ts.end());
  return ts[ts.size() / 2];
}

template <int MT>
static void run_gathers(const uint32_t* values, const uint32_t* idx, uint32_t* out, uint64_t n, int reps) {
  const unsigned grid = (unsigned)(n / 2048);
#define G_CASE(K, MOD) { const double ms = time_ms([&] { hipLaunchKernelGGL((gather_k<K, MT>), dim3(grid), dim3(256), 0, nullptr, values, idx, out); }, reps); \
    printf("{\"what\": \"gather\", \"mem\": \"%s\", \"policy\": \"%s\", \"ms\": %.4f, \"G_rows_s\": %.2f}\n", kMem[MT], kPolicy[K], ms, n / ms / 1e6); fflush(stdout); }
  POLICIES(G_CASE)
}
template <int MT>
static void run_scatters(uint32_t* dst, const uint32_t* idx, uint64_t n, int reps) {
  const unsigned grid = (unsigned)(n / 2048);
#define S_CASE(K, MOD) { const double ms = time_ms([&] { hipLaunchKernelGGL((scatter_k<K, MT>), dim3(grid), dim3(256), 0, nullptr, dst, idx); }, reps); \
    printf("{\"what\": \"scatter\", \"mem\": \"%s\", \"policy\": \"%s\", \"ms\": %.4f, \"G_rows_s\": %.2f}\n", kMem[MT], kPolicy[K], ms, n / ms / 1e6); fflush(stdout); }
  POLICIES(S_CASE)
}

int main(int argc, char** argv) {
  const int lr = argc > 1 ? atoi(argv[1]) : 28, ls = argc > 2 ? atoi(argv[2]) : 28;
  const char* what = argc > 3 ? argv[3] : "gsc";
  const int reps = argc > 4 ? atoi(argv[4]) : 5;
  const uint64_t n = 1ull << lr, m = 1ull << ls;
  uint32_t *idx, *out;
  CK(hipMalloc((void**)&idx, n * 4));
  CK(hipMalloc((void**)&out, n * 4));
  hipLaunchKernelGGL(fill_idx, dim3(4096), dim3(256), 0, nullptr, idx, n, (uint32_t)(m - 1), 20250418ull);
  CK(hipDeviceSynchronize());
  printf("{\"what\": \"setup\", \"rows\": %llu, \"source_elements\": %llu}\n", (unsigned long long)n, (unsigned long long)m);
  for (int mt = 0; mt < 4; mt++) {
    uint32_t* v = (uint32_t*)alloc_mt(mt, m * 4);
    if (!v) continue;
    hipLaunchKernelGGL(fill_iota, dim3(4096), dim3(256), 0, nullptr, v, m);
    CK(hipDeviceSynchronize());
    if (strchr(what, 'g')) {
      if (mt == 0) run_gathers<0>(v, idx, out, n, reps);
      if (mt == 1) run_gathers<1>(v, idx, out, n, reps);
      if (mt == 2) run_gathers<2>(v, idx, out, n, reps);
      if (mt == 3) run_gathers<3>(v, idx, out, n, reps);
      // the gather is only worth anything if it returns the values
      std::vector<uint32_t> hi(4096), ho(4096);
      CK(hipMemcpy(hi.data(), idx + (n - 4096), 4096 * 4, hipMemcpyDeviceToHost));
      CK(hipMemcpy(ho.data(), out + (n - 4096), 4096 * 4, hipMemcpyDeviceToHost));
      int bad = 0;
      for (int i = 0; i < 4096; i++) bad += ho[i] != hi[i] * 2654435761u;
      printf("{\"what\": \"gather_check\", \"mem\": \"%s\", \"mismatches_in_last_4096\": %d}\n", kMem[mt], bad);
    }
    if (strchr(what, 's')) {
      if (mt == 0) run_scatters<0>(v, idx, n, reps);
      if (mt == 1) run_scatters<1>(v, idx, n, reps);
      if (mt == 2) run_scatters<2>(v, idx, n, reps);
      if (mt == 3) run_scatters<3>(v, idx, n, reps);
    }
    if (strchr(what, 'c') && m == n) {
      const unsigned grid = (unsigned)(n / 4 / 64);
      double a = 0, b = 0;
#define C_CASE(MT) if (mt == MT) { \
        a = time_ms([&] { hipLaunchKernelGGL((copy_k<MT, 0>), dim3(grid), dim3(64), 0, nullptr, (const u32x4*)v, (u32x4*)out); }, reps); \
        b = time_ms([&] { hipLaunchKernelGGL((copy_k<0, MT>), dim3(grid), dim3(64), 0, nullptr, (const u32x4*)idx, (u32x4*)v); }, reps); }
      C_CASE(0) C_CASE(1) C_CASE(2) C_CASE(3)
      printf("{\"what\": \"copy\", \"mem\": \"%s\", \"read_from_it_ms\": %.4f, \"read_GBps\": %.0f, \"write_to_it_ms\": %.4f, \"write_GBps\": %.0f}\n", kMem[mt], a,
             8.0 * n / a / 1e6, b, 8.0 * n / b / 1e6);
    }
    fflush(stdout);
    CK(hipFree(v));
  }
  return 0;
}
