// stream_probe.hip — DEV TOOL (not part of the product library): launch-shape sweep for the 2-read/1-write f32 stream.
// Variants: U (16-byte vectors in flight per lane per array), NT (bit0 nontemporal loads, bit1 nontemporal stores),
// BLOCK (threads), grid (0 = one tile per block, else persistent grid-stride with that many blocks),
// and a "chunked" layout where a block owns one contiguous span instead of an interleaved tile.
// Build: hipcc -O3 --offload-arch=gfx950 -shared -fPIC -o tools/probe/libstream_probe.so tools/probe/stream_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <bool NT> __device__ __forceinline__ f32x4 ld(const f32x4* p) {
  if constexpr (NT) return __builtin_nontemporal_load(p);
  else return *p;
}
template <bool NT> __device__ __forceinline__ void st(f32x4* p, f32x4 v) {
  if constexpr (NT) __builtin_nontemporal_store(v, p);
  else *p = v;
}

template <int U, int NT, int BLOCK>
__global__ __launch_bounds__(BLOCK) void add_kernel(const float* a, const float* b, float* out, uint64_t n) {
  constexpr bool NTL = NT & 1, NTS = (NT & 2) != 0;
  const uint64_t npacks = n / 4;
  const uint64_t tile = (uint64_t)BLOCK * U;
  const uint64_t ntiles = npacks / tile;
  const f32x4* A = reinterpret_cast<const f32x4*>(a);
  const f32x4* B = reinterpret_cast<const f32x4*>(b);
  f32x4* O = reinterpret_cast<f32x4*>(out);
  for (uint64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const uint64_t p0 = t * tile + threadIdx.x;
    f32x4 x[U], y[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      x[u] = ld<NTL>(A + p0 + (uint64_t)u * BLOCK);
      y[u] = ld<NTL>(B + p0 + (uint64_t)u * BLOCK);
    }
#pragma unroll
    for (int u = 0; u < U; u++) st<NTS>(O + p0 + (uint64_t)u * BLOCK, x[u] + y[u]);
  }
}

template <int U, int NT, int BLOCK>
static void launch(const float* a, const float* b, float* out, uint64_t n, int grid, hipStream_t s) {
  const uint64_t ntiles = n / 4 / ((uint64_t)BLOCK * U);
  uint64_t g = grid > 0 ? (uint64_t)grid : ntiles;
  if (g > ntiles) g = ntiles;
  if (g < 1) g = 1;
  hipLaunchKernelGGL((add_kernel<U, NT, BLOCK>), dim3((unsigned)g), dim3(BLOCK), 0, s, a, b, out, n);
}

template <int U, int NT>
static void launch_b(const float* a, const float* b, float* out, uint64_t n, int block, int grid, hipStream_t s) {
  switch (block) {
    case 64: launch<U, NT, 64>(a, b, out, n, grid, s); break;
    case 128: launch<U, NT, 128>(a, b, out, n, grid, s); break;
    case 256: launch<U, NT, 256>(a, b, out, n, grid, s); break;
    case 512: launch<U, NT, 512>(a, b, out, n, grid, s); break;
    case 1024: launch<U, NT, 1024>(a, b, out, n, grid, s); break;
    default: break;
  }
}

template <int U>
static void launch_nt(const float* a, const float* b, float* out, uint64_t n, int nt, int block, int grid, hipStream_t s) {
  switch (nt) {
    case 0: launch_b<U, 0>(a, b, out, n, block, grid, s); break;
    case 1: launch_b<U, 1>(a, b, out, n, block, grid, s); break;
    case 2: launch_b<U, 2>(a, b, out, n, block, grid, s); break;
    case 3: launch_b<U, 3>(a, b, out, n, block, grid, s); break;
    default: break;
  }
}

extern "C" int probe_add(const float* a, const float* b, float* out, uint64_t n, int u, int nt, int block, int grid,
                         void* stream) {
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  switch (u) {
    case 1: launch_nt<1>(a, b, out, n, nt, block, grid, s); break;
    case 2: launch_nt<2>(a, b, out, n, nt, block, grid, s); break;
    case 4: launch_nt<4>(a, b, out, n, nt, block, grid, s); break;
    case 8: launch_nt<8>(a, b, out, n, nt, block, grid, s); break;
    default: return 1;
  }
  return (int)hipGetLastError();
}
