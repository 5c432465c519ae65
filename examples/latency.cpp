// The reference's own criterion shapes from a compiled host, with the VALUE on the host at the end of every iteration
// [ref: crates/benchmarks/benches/compare_sum.rs:17-40 — UInt32ArrayGPU::broadcast(2, n).sum() at 1 Mi and 10 Mi rows;
//  compare_gpu_arrow.rs:18-43 — add_dyn(column, 1-element column) at 10 Mi rows].  criterion's loop returns after queue.submit; here an
// iteration ends when the host holds the sum (raw_values of the 1-element result) or, for the add, when the device is idle.
// Prints one JSON line: best and median µs of 300 iterations per shape.  No device → exit 2 (no CPU fallback).
// build: hipcc -std=c++17 -O2 -x c++ examples/latency.cpp -Larrow_gpu_amd/lib -larrow_gpu_hip -Wl,-rpath,$PWD/arrow_gpu_amd/lib
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>

#include "../host/arrow_gpu.hpp"

using namespace arrow_gpu;

template <typename F>
static void time_us(F&& f, int iters, double* best, double* median) {
  std::vector<double> t((size_t)iters);
  for (int i = 0; i < 20; i++) f();
  for (int i = 0; i < iters; i++) {
    const auto t0 = std::chrono::steady_clock::now();
    f();
    t[(size_t)i] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
  }
  std::sort(t.begin(), t.end());
  *best = t[0];
  *median = t[t.size() / 2];
}

int main() {
  DevicePtr dev;
  try {
    dev = GpuDevice::create(0);
  } catch (const std::exception& e) {
    fprintf(stderr, "no device: this program has no CPU fallback (%s)\n", e.what());
    return 2;
  }
  try {
    const int iters = 300;
    printf("{\"what\": \"C++ host, the reference's criterion shapes, an iteration ends with the value on the host\"");
    for (size_t n : {(size_t)1 << 20, (size_t)10 << 20}) {
      auto u = UInt32ArrayGPU::broadcast(2u, n, dev);
      uint32_t got = 0;
      double b0, m0, b1, m1;
      time_us([&] { got = u.sum().raw_values()[0]; }, iters, &b0, &m0);
      if (got != (uint32_t)(2 * n)) {
        fprintf(stderr, "wrong sum: %u\n", got);
        return 1;
      }
      time_us([&] { auto s = u.sum(); check(agpu_device_sync(dev->raw), "agpu_device_sync"); }, iters, &b1, &m1);
      printf(", \"u32_sum_%zuMi_value_on_host_us\": {\"best\": %.2f, \"median\": %.2f}, \"u32_sum_%zuMi_device_sync_us\": {\"best\": %.2f, \"median\": %.2f}",
             n >> 20, b0, m0, n >> 20, b1, m1);
    }
    {
      const size_t n = (size_t)10 << 20;
      std::vector<float> host(n);
      for (size_t i = 0; i < n; i++) host[i] = (float)i;
      auto col = Float32ArrayGPU::from_slice(host, dev);
      auto val = Float32ArrayGPU::from_slice({100.0f}, dev);
      double b, m;
      time_us([&] { auto r = add_dyn(col, val); check(agpu_device_sync(dev->raw), "agpu_device_sync"); }, iters, &b, &m);
      printf(", \"f32_add_scalar_10Mi_device_sync_us\": {\"best\": %.2f, \"median\": %.2f}", b, m);
    }
    printf("}\n");
    fflush(stdout);
    return 0;
  } catch (const std::exception& e) {
    fprintf(stderr, "failed: %s\n", e.what());
    return 1;
  }
}
