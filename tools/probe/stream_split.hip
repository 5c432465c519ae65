// DEV TOOL (round 6): does reading ONE column as S lock-step streams a distance D apart beat the sequential order?
// Two columns read together run at 7.0–7.3 TB/s when element i of the two falls into different classes of the memory-channel hash
// (DESIGN.md §3), a single column read front to back at 6.5–6.6 (the reductions).  Here the block → chunk order of a one-column
// kernel is permuted so that S chunks D bytes apart are in flight together:
//   virtual block v → stream v % S, index i = v / S → byte offset (i / per)·S·D + (v % S)·D + (i % per)·CH,  per = D / CH
// (S = 1: the sequential order).  Kernels: K1 read-only max over 64 KiB chunks per one-wave block (the reductions' shape),
// K2 out = −in with 1 KiB per one-wave block (the unary kernels' shape), K3 out = a + b in the same shape.
// Standalone: hipcc -O3 --offload-arch=gfx950 -o stream_split stream_split.hip && ./stream_split
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x)                                                                              \
  do {                                                                                     \
    hipError_t e_ = (x);                                                                   \
    if (e_ != hipSuccess) {                                                                \
      fprintf(stderr, "%s failed: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); \
      exit(2);                                                                             \
    }                                                                                      \
  } while (0)

__device__ __forceinline__ uint64_t chunk_offset(uint64_t v, uint32_t S, uint64_t D, uint64_t per, uint64_t CH) {
  if (S == 1) return v * CH;
  const uint64_t s = v % S, i = v / S;
  return (i / per) * (S * D) + s * D + (i % per) * CH;
}

__global__ __launch_bounds__(64) void k_read(const char* in, float* partials, uint32_t S, uint64_t D, uint64_t per) {
  const uint64_t off = chunk_offset(blockIdx.x, S, D, per, 65536);
  const f32x4* base = reinterpret_cast<const f32x4*>(in + off) + threadIdx.x;
  float m = -3.0e38f;
  for (int j0 = 0; j0 < 64; j0 += 8) {
    f32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; u++) v[u] = __builtin_nontemporal_load(base + (j0 + u) * 64);
#pragma unroll
    for (int u = 0; u < 8; u++) m = fmaxf(fmaxf(fmaxf(m, v[u].x), fmaxf(v[u].y, v[u].z)), v[u].w);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_down(m, o));
  if (threadIdx.x == 0) partials[blockIdx.x] = m;
}

__global__ __launch_bounds__(64) void k_neg(const char* in, char* out, uint32_t S, uint64_t D, uint64_t per) {
  const uint64_t off = chunk_offset(blockIdx.x, S, D, per, 1024) + threadIdx.x * 16;
  f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(in + off));
  v = -v;
  __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(out + off));
}

__global__ __launch_bounds__(64) void k_add(const char* a, const char* b, char* out, uint32_t S, uint64_t D, uint64_t per) {
  const uint64_t off = chunk_offset(blockIdx.x, S, D, per, 1024) + threadIdx.x * 16;
  const f32x4 x = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(a + off));
  const f32x4 y = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(b + off));
  __builtin_nontemporal_store(x + y, reinterpret_cast<f32x4*>(out + off));
}

__global__ void k_fill(float* p, uint64_t n) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
    p[i] = (float)(i % 1000) * 0.25f - 100.0f;
}

struct Cfg {
  uint32_t S;
  uint64_t D;
  const char* what;
};

int main(int argc, char** argv) {
  const uint64_t CAP = (1ull << 32) + (1ull << 28);  // bytes per buffer
  const uint64_t TARGET = 4000000000ull;             // ≈ the 1e9-row f32 column
  char *a, *b, *o;
  float* part;
  CK(hipMalloc(&a, CAP));
  CK(hipMalloc(&b, CAP));
  CK(hipMalloc(&o, CAP));
  CK(hipMalloc(&part, 4 * (CAP / 65536 + 64)));
  k_fill<<<4096, 256>>>(reinterpret_cast<float*>(a), CAP / 4);
  k_fill<<<4096, 256>>>(reinterpret_cast<float*>(b), CAP / 4);
  CK(hipDeviceSynchronize());
  printf("{\"buffers\": {\"a\": \"%p\", \"b\": \"%p\", \"out\": \"%p\"}}\n", (void*)a, (void*)b, (void*)o);
  const uint64_t K = 1ull << 10, M = 1ull << 20, G = 1ull << 30;
  std::vector<Cfg> cfgs = {
      {1, 0, "sequential"},
      {2, 256 * M, "bit 28"},
      {2, 256 * M + 2 * M, "bits 28 + 21 (cancel)"},
      {2, 512 * M, "bit 29 (none)"},
      {2, 512 * M + 2 * M, "bit 29 + 21"},
      {2, 512 * M + 8 * K, "bit 29 + 13"},
      {2, 128 * M, "bit 27 (weaker)"},
      {2, 64 * M, "bit 26"},
      {2, 2 * M, "bit 21 alone (4 MiB superblocks)"},
      {2, 2 * G - 512 * M + 2 * M, "1.5 GiB + bit 21"},
      {2, 2 * G - 512 * M, "1.5 GiB"},
      {4, 256 * M, "4 streams, 256 MiB"},
      {4, 512 * M + 2 * M, "4 streams, 512 MiB + 2 MiB"},
      {4, 512 * M + 8 * K, "4 streams, 512 MiB + 8 KiB"},
      {8, 512 * M + 2 * M, "8 streams, 512 MiB + 2 MiB"},
      {8, 256 * M + 8 * K, "8 streams, 256 MiB + 8 KiB"},
      {3, 512 * M + 2 * M, "3 streams, 512 MiB + 2 MiB"},
      {1, 0, "sequential (again)"},
  };
  const int reps = argc > 1 ? atoi(argv[1]) : 9;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int kern = 0; kern < 3; kern++) {
    const uint64_t CH = kern == 0 ? 65536 : 1024;
    for (const Cfg& c : cfgs) {
      // total bytes: whole superblocks of S·D bytes, as close to TARGET as the buffer allows
      uint64_t per = 0, total;
      if (c.S == 1) {
        total = TARGET / 65536 * 65536;
      } else {
        if (c.D % CH) {  // D must hold whole chunks of this kernel: round the chunk count down, the streams stay D apart
          per = c.D / CH;
        } else {
          per = c.D / CH;
        }
        const uint64_t sb = (uint64_t)c.S * c.D;
        uint64_t nsb = (TARGET + sb / 2) / sb;
        if (nsb == 0) nsb = 1;
        while (nsb * sb > CAP) nsb--;
        if (nsb == 0) continue;
        total = nsb * c.S * per * CH;
      }
      const uint64_t nblocks = total / CH;
      if (nblocks > 0x7fffffffull) continue;
      std::vector<float> ms;
      for (int r = 0; r < reps + 3; r++) {
        CK(hipEventRecord(e0, 0));
        if (kern == 0) k_read<<<(unsigned)nblocks, 64>>>(a, part, c.S, c.D, per);
        else if (kern == 1) k_neg<<<(unsigned)nblocks, 64>>>(a, o, c.S, c.D, per);
        else k_add<<<(unsigned)nblocks, 64>>>(a, b, o, c.S, c.D, per);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float t;
        CK(hipEventElapsedTime(&t, e0, e1));
        if (r >= 3) ms.push_back(t);
      }
      std::sort(ms.begin(), ms.end());
      const double med = ms[ms.size() / 2];
      const double bytes = (double)total * (kern == 0 ? 1 : kern == 1 ? 2 : 3);
      printf("{\"kernel\": \"%s\", \"S\": %u, \"D\": %llu, \"what\": \"%s\", \"bytes\": %llu, \"ms\": %.4f, \"TBps\": %.3f, \"frac\": %.4f}\n",
             kern == 0 ? "read" : kern == 1 ? "neg" : "add", c.S, (unsigned long long)c.D, c.what, (unsigned long long)total, med,
             bytes / med / 1e9, bytes / med / 1e9 / 8.0);
      fflush(stdout);
    }
  }
  return 0;
}
