// DEV TOOL: shapes for the 4-byte → 1-byte narrowing stream (cast f32→u8: 4 B read + 1 B written per row).
//   variant 0: lane loads U × 16 B (4 rows each), stores U × 4 B          (what cvt_kernel does; BLOCK 64 / 256)
//   variant 1: one wave per 1024 rows: 4 coalesced 16-byte loads per lane, each converted to one dword; the 256 dwords are
//              transposed inside the wave (16 ds_bpermute) so that every lane stores 16 contiguous bytes (ONE 1 KiB store)
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t cvt1(float x) {
  uint32_t u;
  if (!(x > 0.0f)) u = 0;
  else if (x >= 4294967296.0f) u = 0xFFFFFFFFu;
  else u = (uint32_t)x;
  return u & 255u;
}
__device__ __forceinline__ uint32_t cvt4(f32x4 v) { return cvt1(v.x) | (cvt1(v.y) << 8) | (cvt1(v.z) << 16) | (cvt1(v.w) << 24); }

template <int BLOCK, int U>
__global__ __launch_bounds__(BLOCK) void narrow_v0(const f32x4* in, uint32_t* out, uint64_t ntiles) {
  for (uint64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const uint64_t p0 = t * (uint64_t)(BLOCK * U) + threadIdx.x;
    f32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; u++) v[u] = __builtin_nontemporal_load(in + p0 + (uint64_t)u * BLOCK);
#pragma unroll
    for (int u = 0; u < U; u++) __builtin_nontemporal_store(cvt4(v[u]), out + p0 + (uint64_t)u * BLOCK);
  }
}

__global__ __launch_bounds__(64) void narrow_v1(const f32x4* in, u32x4* out, uint64_t nchunks) {
  const uint32_t lane = threadIdx.x;
  for (uint64_t c = blockIdx.x; c < nchunks; c += gridDim.x) {
    uint32_t w[4];
#pragma unroll
    for (int j = 0; j < 4; j++) w[j] = cvt4(__builtin_nontemporal_load(in + c * 256 + j * 64 + lane));
    // lane l' stores dwords q = 4 l' + k, k = 0..3; dword q was produced by load j = q / 64 = l' / 16 in lane q % 64
    const int src0 = (int)((4 * (lane & 15)) * 4);
    const uint32_t jsel = lane >> 4;
    u32x4 r;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int src = src0 + 4 * k;
      const uint32_t a0 = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)w[0]);
      const uint32_t a1 = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)w[1]);
      const uint32_t a2 = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)w[2]);
      const uint32_t a3 = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)w[3]);
      r[k] = jsel == 0 ? a0 : jsel == 1 ? a1 : jsel == 2 ? a2 : a3;
    }
    __builtin_nontemporal_store(r, out + c * 64 + lane);
  }
}

extern "C" int probe_narrow(const void* in, void* out, uint64_t n, int variant, int block, int u, void* stream) {
  hipStream_t s = (hipStream_t)stream;
#define GO0(B, U_)                                                                                                   \
  {                                                                                                                  \
    uint64_t nt = n / 4 / ((uint64_t)B * U_);                                                                        \
    hipLaunchKernelGGL((narrow_v0<B, U_>), dim3((unsigned)nt), dim3(B), 0, s, (const f32x4*)in, (uint32_t*)out, nt); \
  }
  if (variant == 0) {
    if (block == 64 && u == 1) GO0(64, 1) else if (block == 64 && u == 4) GO0(64, 4) else if (block == 256 && u == 1) GO0(256, 1)
    else if (block == 256 && u == 4) GO0(256, 4) else if (block == 256 && u == 2) GO0(256, 2) else if (block == 64 && u == 2) GO0(64, 2)
    else return 1;
  } else {
    const uint64_t nchunks = n / 1024;
    hipLaunchKernelGGL(narrow_v1, dim3((unsigned)nchunks), dim3(64), 0, s, (const f32x4*)in, (u32x4*)out, nchunks);
  }
  return (int)hipGetLastError();
}
