// small_ops.cpp — host-side cost of the reference-shaped default API (`a.add(b)`: new pipeline + new output buffer +
// finish per call [ref: impl_arithmetic_op! crates/arithmetic/src/lib.rs:11-50]) through the C++ host, at the sizes the
// reference's own tests use (100 rows) and at 1 Mi rows, with the resource pools on and off.
//   hipcc -std=c++17 -O2 -x c++ tools/probe/small_ops.cpp -o small_ops -Larrow_gpu_amd/lib -larrow_gpu_hip && ./small_ops
#include <chrono>
#include <cstdio>
#include <numeric>

#include "../../host/arrow_gpu.hpp"

using namespace arrow_gpu;

static double us_per_call(const Int32ArrayGPU& a, const Int32ArrayGPU& b, int reps) {
  for (int i = 0; i < 20; i++) (void)a.add(b);
  check(agpu_device_sync(a.gpu_device->raw), "sync");
  const auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < reps; i++) (void)a.add(b);  // result dropped at once, like a temporary in an expression
  const auto t1 = std::chrono::steady_clock::now();
  check(agpu_device_sync(a.gpu_device->raw), "sync");
  return std::chrono::duration<double, std::micro>(t1 - t0).count() / reps;
}

int main() {
  int32_t nd = 0;
  agpu_device_count(&nd);
  if (nd <= 0) {
    printf("no device\n");
    return 2;
  }
  auto dev = GpuDevice::create(0);
  printf("{\"what\": \"C++ host a.add(b), host-side microseconds per call (issue cost; the GPU runs behind)\"");
  for (size_t n : {(size_t)100, (size_t)1 << 20}) {
    std::vector<int32_t> h(n);
    std::iota(h.begin(), h.end(), 0);
    auto a = Int32ArrayGPU::from_slice(h, dev), b = Int32ArrayGPU::from_slice(h, dev);
    for (int pool : {1, 0}) {
      check(agpu_set_tuning("mem_pool", pool), "tuning");
      const double us = us_per_call(a, b, pool ? 2000 : 200);
      printf(", \"n%zu_pool%d_us\": %.2f", n, pool, us);
    }
    check(agpu_set_tuning("mem_pool", 1), "tuning");
    auto c = a.add(b).raw_values();
    if (c[n - 1] != 2 * (int32_t)(n - 1)) {
      printf(", \"error\": \"wrong result\"}\n");
      return 1;
    }
  }
  printf("}\n");
  return 0;
}
