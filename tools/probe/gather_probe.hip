// DEV TOOL: random 4-byte gather (take) shapes.  out[i] = values[idx[i]].
//   G  = gathers in flight per lane (4, 8, 16), NT = nontemporal gather loads, BLOCK = 64 / 256.
// Index and output streams are coalesced 16-byte accesses in every variant.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int G, bool NT, int BLOCK>
__global__ __launch_bounds__(BLOCK) void take_probe(const uint32_t* values, const uint32_t* idx, uint32_t* out, uint64_t ntiles) {
  constexpr int Q = G / 4;
  for (uint64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const uint64_t p0 = t * (uint64_t)BLOCK * Q + threadIdx.x;
    u32x4 ix[Q];
#pragma unroll
    for (int q = 0; q < Q; q++) ix[q] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(idx) + p0 + (uint64_t)q * BLOCK);
    u32x4 r[Q];
#pragma unroll
    for (int q = 0; q < Q; q++) {
      if (NT) {
        r[q].x = __builtin_nontemporal_load(values + ix[q].x);
        r[q].y = __builtin_nontemporal_load(values + ix[q].y);
        r[q].z = __builtin_nontemporal_load(values + ix[q].z);
        r[q].w = __builtin_nontemporal_load(values + ix[q].w);
      } else {
        r[q].x = values[ix[q].x];
        r[q].y = values[ix[q].y];
        r[q].z = values[ix[q].z];
        r[q].w = values[ix[q].w];
      }
    }
#pragma unroll
    for (int q = 0; q < Q; q++) __builtin_nontemporal_store(r[q], reinterpret_cast<u32x4*>(out) + p0 + (uint64_t)q * BLOCK);
  }
}

template <int G, bool NT>
static void launch_b(const uint32_t* v, const uint32_t* idx, uint32_t* out, uint64_t n, int block, int grid, hipStream_t s) {
  const uint64_t tile = (uint64_t)block * G;
  const uint64_t ntiles = n / tile;
  const unsigned g = grid > 0 ? (unsigned)grid : (unsigned)ntiles;
  if (block == 64) hipLaunchKernelGGL((take_probe<G, NT, 64>), dim3(g), dim3(64), 0, s, v, idx, out, ntiles);
  else hipLaunchKernelGGL((take_probe<G, NT, 256>), dim3(g), dim3(256), 0, s, v, idx, out, ntiles);
}

extern "C" int probe_take(const uint32_t* v, const uint32_t* idx, uint32_t* out, uint64_t n, int g, int nt, int block, int grid,
                          void* stream) {
  hipStream_t s = (hipStream_t)stream;
  switch (g * 2 + (nt ? 1 : 0)) {
    case 8: launch_b<4, false>(v, idx, out, n, block, grid, s); break;
    case 9: launch_b<4, true>(v, idx, out, n, block, grid, s); break;
    case 16: launch_b<8, false>(v, idx, out, n, block, grid, s); break;
    case 17: launch_b<8, true>(v, idx, out, n, block, grid, s); break;
    case 32: launch_b<16, false>(v, idx, out, n, block, grid, s); break;
    case 33: launch_b<16, true>(v, idx, out, n, block, grid, s); break;
    default: return -1;
  }
  return (int)hipGetLastError();
}
