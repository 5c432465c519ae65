// DEV TOOL: f32 add (one-wave blocks, one 16-byte pack per lane — the product's shape) with every cache-policy
// combination of the gfx950 global load / store instructions (sc0, sc1, nt), which the compiler only exposes as
// "plain" and "nontemporal".  1e9 rows, three 4 GB columns in one block, 512 MiB-multiple + colour spacing.
//   hipcc -O3 --offload-arch=gfx950 tools/probe/cache_policy.hip -o /tmp/cache_policy && /tmp/cache_policy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define LOADS(X) X(0, "") X(1, "nt") X(2, "sc0") X(3, "sc1") X(4, "sc0 sc1") X(5, "sc0 nt") X(6, "sc1 nt") X(7, "sc0 sc1 nt")

template <int LD, int ST>
__global__ __launch_bounds__(64) void add_kernel(const f32x4* a, const f32x4* b, f32x4* o) {
  const uint64_t i = (uint64_t)blockIdx.x * 64 + threadIdx.x;
  f32x4 x, y;
  const f32x4* pa = a + i;
  const f32x4* pb = b + i;
  f32x4* po = o + i;
#define LD_CASE(K, MOD)                                                                                    \
  if constexpr (LD == K) {                                                                                 \
    asm volatile("global_load_dwordx4 %0, %1, off " MOD : "=v"(x) : "v"(pa) : "memory");                  \
    asm volatile("global_load_dwordx4 %0, %1, off " MOD : "=v"(y) : "v"(pb) : "memory");                  \
  }
  LOADS(LD_CASE)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const f32x4 r = x + y;
#define ST_CASE(K, MOD) \
  if constexpr (ST == K) asm volatile("global_store_dwordx4 %0, %1, off " MOD : : "v"(po), "v"(r) : "memory");
  LOADS(ST_CASE)
}

template <int LD, int ST>
static int run(const f32x4* a, const f32x4* b, f32x4* o, uint64_t npacks, const char* ld, const char* st) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  std::vector<float> ts;
  for (int rep = 0; rep < 12; rep++) {
    CK(hipEventRecord(e0, nullptr));
    hipLaunchKernelGGL((add_kernel<LD, ST>), dim3((unsigned)(npacks / 64)), dim3(64), 0, nullptr, a, b, o);
    CK(hipEventRecord(e1, nullptr));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (rep >= 3) ts.push_back(ms);
  }
  std::sort(ts.begin(), ts.end());
  const double ms = ts[ts.size() / 2];
  printf("load [%-10s] store [%-10s]  %.4f ms  %.3f of 8 TB/s\n", ld, st, ms, 12.0 * npacks * 4 / ms / 8e9);
  return 0;
}

int main() {
  const uint64_t n = 1000000000ull / 4096 * 4096, npacks = n / 4;
  const size_t stride = (size_t)4 << 30;  // 4e9 B rounded up to a 512 MiB multiple
  char* base;
  CK(hipMalloc(&base, 3 * stride + (1 << 20)));
  CK(hipMemset(base, 0, 3 * stride + (1 << 20)));
  const f32x4* a = (const f32x4*)base;
  const f32x4* b = (const f32x4*)(base + stride + 8192);
  f32x4* o = (f32x4*)(base + 2 * stride + 4096);
#define RUN(L, S, LN, SN) if (run<L, S>(a, b, o, npacks, LN, SN)) return 1;
  for (int round = 0; round < 2; round++) {
    RUN(1, 1, "nt", "nt") RUN(1, 0, "nt", "") RUN(1, 2, "nt", "sc0") RUN(1, 3, "nt", "sc1") RUN(1, 4, "nt", "sc0 sc1") RUN(1, 5, "nt", "sc0 nt")
    RUN(1, 6, "nt", "sc1 nt") RUN(1, 7, "nt", "sc0 sc1 nt") RUN(0, 1, "", "nt") RUN(2, 1, "sc0", "nt") RUN(3, 1, "sc1", "nt") RUN(4, 1, "sc0 sc1", "nt")
    RUN(5, 1, "sc0 nt", "nt") RUN(6, 1, "sc1 nt", "nt") RUN(7, 1, "sc0 sc1 nt", "nt") RUN(7, 7, "sc0 sc1 nt", "sc0 sc1 nt") RUN(6, 6, "sc1 nt", "sc1 nt")
    RUN(5, 5, "sc0 nt", "sc0 nt")
  }
  return 0;
}
