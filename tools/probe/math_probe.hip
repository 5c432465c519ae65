// DEV TOOL: which f32 device-library functions are already within 1 ULP on gfx950 (so the cheaper f32 version can be
// used instead of the f64-evaluated one)?  out[i] = f(in[i]) for fn index k.
#include <hip/hip_runtime.h>
#include <stdint.h>
__global__ void math_kernel(const float* in, const float* in2, float* out, uint64_t n, int fn) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float x = in[i], y = in2[i], r;
  switch (fn) {
    case 0: r = sinhf(x); break;
    case 1: r = acosf(x); break;
    case 2: r = powf(x, y); break;
    case 3: r = cbrtf(x); break;
    case 4: r = exp2f(x); break;
    case 5: r = log2f(x); break;
    case 6: r = logf(x); break;
    case 7: r = expf(x); break;
    default: r = x;
  }
  out[i] = r;
}
extern "C" int probe_math(const float* in, const float* in2, float* out, uint64_t n, int fn, void* stream) {
  hipLaunchKernelGGL(math_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, in, in2, out, n, fn);
  return (int)hipGetLastError();
}
