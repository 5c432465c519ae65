// stream_probe.hip — DEV TOOL (not part of the product library): launch-shape sweeps for the two headline streams.
//   probe_add : f32 a+b → out.  U (16-byte vectors per lane per array), NT (bit0 loads, bit1 stores), BLOCK (threads),
//               grid (0 = one tile per block, else persistent), xcd (1 = each XCD streams one contiguous eighth)
//   probe_eq  : i32 a==b → bitmap + validity AND.  variant 0 = ballot with R rounds per wave (dword loads),
//               variant 1 = 16-byte vector loads + lane-group OR with U vectors per lane.
// Build: hipcc -O3 --offload-arch=gfx950 -shared -fPIC -o tools/probe/libstream_probe.so tools/probe/stream_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

template <bool NT, typename V> __device__ __forceinline__ V ld(const V* p) {
  if constexpr (NT) return __builtin_nontemporal_load(p);
  else return *p;
}
template <bool NT, typename V> __device__ __forceinline__ void st(V* p, V v) {
  if constexpr (NT) __builtin_nontemporal_store(v, p);
  else *p = v;
}

// ------------------------------------------------------------------ add
template <int U, int NT, int BLOCK, bool XCD>
__global__ __launch_bounds__(BLOCK) void add_kernel(const float* a, const float* b, float* out, uint64_t n) {
  constexpr bool NTL = NT & 1, NTS = (NT & 2) != 0;
  const uint64_t npacks = n / 4;
  const uint64_t tile = (uint64_t)BLOCK * U;
  const uint64_t ntiles = npacks / tile;
  const f32x4* A = reinterpret_cast<const f32x4*>(a);
  const f32x4* B = reinterpret_cast<const f32x4*>(b);
  f32x4* O = reinterpret_cast<f32x4*>(out);
  for (uint64_t t0 = blockIdx.x; t0 < ntiles; t0 += gridDim.x) {
    uint64_t t = t0;
    if constexpr (XCD) {  // blocks are dealt round-robin to the 8 XCDs: give XCD x the x-th contiguous eighth
      const uint64_t per = ntiles / 8;
      if (t0 < per * 8) t = (t0 & 7) * per + (t0 >> 3);
    }
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
static void launch(const float* a, const float* b, float* out, uint64_t n, int grid, int xcd, hipStream_t s) {
  const uint64_t ntiles = n / 4 / ((uint64_t)BLOCK * U);
  uint64_t g = grid > 0 ? (uint64_t)grid : ntiles;
  if (g > ntiles) g = ntiles;
  if (g < 1) g = 1;
  if (xcd)
    hipLaunchKernelGGL((add_kernel<U, NT, BLOCK, true>), dim3((unsigned)g), dim3(BLOCK), 0, s, a, b, out, n);
  else
    hipLaunchKernelGGL((add_kernel<U, NT, BLOCK, false>), dim3((unsigned)g), dim3(BLOCK), 0, s, a, b, out, n);
}

template <int U, int NT>
static void launch_b(const float* a, const float* b, float* out, uint64_t n, int block, int grid, int xcd, hipStream_t s) {
  switch (block) {
    case 64: launch<U, NT, 64>(a, b, out, n, grid, xcd, s); break;
    case 128: launch<U, NT, 128>(a, b, out, n, grid, xcd, s); break;
    case 256: launch<U, NT, 256>(a, b, out, n, grid, xcd, s); break;
    case 512: launch<U, NT, 512>(a, b, out, n, grid, xcd, s); break;
    case 1024: launch<U, NT, 1024>(a, b, out, n, grid, xcd, s); break;
    default: break;
  }
}

template <int U>
static void launch_nt(const float* a, const float* b, float* out, uint64_t n, int nt, int block, int grid, int xcd,
                      hipStream_t s) {
  switch (nt) {
    case 0: launch_b<U, 0>(a, b, out, n, block, grid, xcd, s); break;
    case 1: launch_b<U, 1>(a, b, out, n, block, grid, xcd, s); break;
    case 2: launch_b<U, 2>(a, b, out, n, block, grid, xcd, s); break;
    case 3: launch_b<U, 3>(a, b, out, n, block, grid, xcd, s); break;
    default: break;
  }
}

extern "C" int probe_add(const float* a, const float* b, float* out, uint64_t n, int u, int nt, int block, int grid,
                         int xcd, void* stream) {
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  switch (u) {
    case 1: launch_nt<1>(a, b, out, n, nt, block, grid, xcd, s); break;
    case 2: launch_nt<2>(a, b, out, n, nt, block, grid, xcd, s); break;
    case 4: launch_nt<4>(a, b, out, n, nt, block, grid, xcd, s); break;
    case 8: launch_nt<8>(a, b, out, n, nt, block, grid, xcd, s); break;
    default: return 1;
  }
  return (int)hipGetLastError();
}

// ------------------------------------------------------------------ eq → bitmap (+ validity AND)
template <int R, int BLOCK, bool NT>
__global__ __launch_bounds__(BLOCK) void eq_ballot_kernel(const int* a, const int* b, const uint64_t* va,
                                                         const uint64_t* vb, uint64_t* out, uint64_t* outv, uint64_t n) {
  constexpr uint64_t WAVE_TILE = 64ull * R;
  constexpr uint64_t TILE = WAVE_TILE * (BLOCK / 64);
  const uint64_t ntiles = n / TILE;
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (uint64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const uint64_t e0 = t * TILE + wave * WAVE_TILE;
    const uint64_t w0 = e0 / 64;
    int xa[R], xb[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
      xa[r] = ld<NT>(a + e0 + (uint64_t)r * 64 + lane);
      xb[r] = ld<NT>(b + e0 + (uint64_t)r * 64 + lane);
    }
    uint64_t vword = 0;
    if (lane < R) vword = va[w0 + lane] & vb[w0 + lane];
    uint64_t word = 0;
#pragma unroll
    for (int r = 0; r < R; r++) {
      const uint64_t m = __ballot(xa[r] == xb[r]);
      if (lane == (uint32_t)r) word = m;
    }
    if (lane < R) {
      out[w0 + lane] = word;
      outv[w0 + lane] = vword;
    }
  }
}

template <int U, int BLOCK, bool NT>
__global__ __launch_bounds__(BLOCK) void eq_vec_kernel(const int* a, const int* b, const uint32_t* va, const uint32_t* vb,
                                                      uint32_t* out, uint32_t* outv, uint64_t n) {
  constexpr uint64_t TILE_PACKS = (uint64_t)BLOCK * U;
  const uint64_t ntiles = n / 4 / TILE_PACKS;
  const uint32_t lane = threadIdx.x & 63;
  const i32x4* A = reinterpret_cast<const i32x4*>(a);
  const i32x4* B = reinterpret_cast<const i32x4*>(b);
  for (uint64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const uint64_t p0 = t * TILE_PACKS + threadIdx.x;
    i32x4 x[U], y[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      x[u] = ld<NT>(A + p0 + (uint64_t)u * BLOCK);
      y[u] = ld<NT>(B + p0 + (uint64_t)u * BLOCK);
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      const uint64_t pk = p0 + (uint64_t)u * BLOCK;
      uint32_t m = (uint32_t)(x[u].x == y[u].x) | ((uint32_t)(x[u].y == y[u].y) << 1) |
                   ((uint32_t)(x[u].z == y[u].z) << 2) | ((uint32_t)(x[u].w == y[u].w) << 3);
      uint32_t v = m << (4 * (lane & 7));
      v |= (uint32_t)__shfl_xor((int)v, 1);
      v |= (uint32_t)__shfl_xor((int)v, 2);
      v |= (uint32_t)__shfl_xor((int)v, 4);
      if ((lane & 7) == 0) {
        const uint64_t w = pk >> 3;
        out[w] = v;
        outv[w] = va[w] & vb[w];
      }
    }
  }
}

template <int R, bool NT>
static void launch_ballot(const int* a, const int* b, const void* va, const void* vb, void* out, void* outv, uint64_t n,
                          int block, int grid, hipStream_t s) {
#define GO(BLOCK)                                                                                                   \
  {                                                                                                                 \
    const uint64_t ntiles = n / (64ull * R * (BLOCK / 64));                                                         \
    uint64_t g = grid > 0 ? (uint64_t)grid : ntiles;                                                                \
    if (g > ntiles) g = ntiles;                                                                                     \
    if (g < 1) g = 1;                                                                                               \
    hipLaunchKernelGGL((eq_ballot_kernel<R, BLOCK, NT>), dim3((unsigned)g), dim3(BLOCK), 0, s, a, b,               \
                       (const uint64_t*)va, (const uint64_t*)vb, (uint64_t*)out, (uint64_t*)outv, n);               \
  }
  if (block == 64) GO(64) else if (block == 128) GO(128) else if (block == 512) GO(512) else GO(256)
#undef GO
}

template <int U, bool NT>
static void launch_vec(const int* a, const int* b, const void* va, const void* vb, void* out, void* outv, uint64_t n,
                       int block, int grid, hipStream_t s) {
#define GO(BLOCK)                                                                                                   \
  {                                                                                                                 \
    const uint64_t ntiles = n / 4 / ((uint64_t)BLOCK * U);                                                          \
    uint64_t g = grid > 0 ? (uint64_t)grid : ntiles;                                                                \
    if (g > ntiles) g = ntiles;                                                                                     \
    if (g < 1) g = 1;                                                                                               \
    hipLaunchKernelGGL((eq_vec_kernel<U, BLOCK, NT>), dim3((unsigned)g), dim3(BLOCK), 0, s, a, b,                  \
                       (const uint32_t*)va, (const uint32_t*)vb, (uint32_t*)out, (uint32_t*)outv, n);               \
  }
  if (block == 64) GO(64) else if (block == 128) GO(128) else if (block == 512) GO(512) else GO(256)
#undef GO
}

extern "C" int probe_eq(const int* a, const int* b, const void* va, const void* vb, void* out, void* outv, uint64_t n,
                        int variant, int ru, int nt, int block, int grid, void* stream) {
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (variant == 0) {
    if (nt) {
      switch (ru) {
        case 1: launch_ballot<1, true>(a, b, va, vb, out, outv, n, block, grid, s); break;
        case 2: launch_ballot<2, true>(a, b, va, vb, out, outv, n, block, grid, s); break;
        case 4: launch_ballot<4, true>(a, b, va, vb, out, outv, n, block, grid, s); break;
        case 8: launch_ballot<8, true>(a, b, va, vb, out, outv, n, block, grid, s); break;
        case 16: launch_ballot<16, true>(a, b, va, vb, out, outv, n, block, grid, s); break;
        default: return 1;
      }
    } else {
      switch (ru) {
        case 4: launch_ballot<4, false>(a, b, va, vb, out, outv, n, block, grid, s); break;
        case 8: launch_ballot<8, false>(a, b, va, vb, out, outv, n, block, grid, s); break;
        case 16: launch_ballot<16, false>(a, b, va, vb, out, outv, n, block, grid, s); break;
        default: return 1;
      }
    }
  } else {
    if (nt) {
      switch (ru) {
        case 1: launch_vec<1, true>(a, b, va, vb, out, outv, n, block, grid, s); break;
        case 2: launch_vec<2, true>(a, b, va, vb, out, outv, n, block, grid, s); break;
        case 4: launch_vec<4, true>(a, b, va, vb, out, outv, n, block, grid, s); break;
        default: return 1;
      }
    } else {
      switch (ru) {
        case 1: launch_vec<1, false>(a, b, va, vb, out, outv, n, block, grid, s); break;
        case 2: launch_vec<2, false>(a, b, va, vb, out, outv, n, block, grid, s); break;
        case 4: launch_vec<4, false>(a, b, va, vb, out, outv, n, block, grid, s); break;
        default: return 1;
      }
    }
  }
  return (int)hipGetLastError();
}
