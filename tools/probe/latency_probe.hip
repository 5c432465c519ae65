// DEV TOOL (round 5): the host-visible latency floor of "one small kernel, one scalar back on the host".
//   a) kernel → hipStreamSynchronize                      (result stays on the device)
//   b) kernel → hipMemcpyAsync D2H 8 B (pinned) → sync     (what agpu_download does for a scalar)
//   c) kernel writes result + sequence number into pinned host memory, host spins on the sequence number (no sync call)
//   d) as (c) with __threadfence_system before the flag
// build: hipcc -O2 --offload-arch=gfx950 tools/probe/latency_probe.hip -o /tmp/latency_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <algorithm>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void k_dev(uint64_t* out, uint64_t v) { if (threadIdx.x == 0) out[0] = v; }
__global__ void k_host(volatile uint64_t* res, volatile uint64_t* flag, uint64_t v) {
  if (threadIdx.x == 0) {
    res[0] = v * 3;
    __threadfence_system();
    flag[0] = v;
  }
}
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
  if (argc > 1) {  // 1 = hipDeviceScheduleSpin, 2 = Yield, 4 = BlockingSync
    const unsigned f = (unsigned)atoi(argv[1]);
    printf("hipSetDeviceFlags(%u): %s\n", f, hipGetErrorString(hipSetDeviceFlags(f)));
  }
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  uint64_t* d;
  CK(hipMalloc(&d, 64));
  uint64_t* h;
  CK(hipHostMalloc(&h, 128, hipHostMallocDefault));
  h[0] = h[8] = 0;
  const int N = 2000;
  auto stat = [&](const char* name, std::vector<double>& t) {
    std::sort(t.begin(), t.end());
    printf("%-60s min %.2f  median %.2f  p90 %.2f us\n", name, t[0], t[t.size() / 2], t[t.size() * 9 / 10]);
  };
  std::vector<double> t(N);
  for (int w = 0; w < 2; w++) {
    for (int i = 0; i < N; i++) { const double t0 = now(); hipLaunchKernelGGL(k_dev, dim3(1), dim3(64), 0, s, d, (uint64_t)i); CK(hipStreamSynchronize(s)); t[i] = now() - t0; }
  }
  stat("a) kernel + hipStreamSynchronize", t);
  for (int i = 0; i < N; i++) { const double t0 = now(); hipLaunchKernelGGL(k_dev, dim3(1), dim3(64), 0, s, d, (uint64_t)i); CK(hipMemcpyAsync(h, d, 8, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s)); t[i] = now() - t0; }
  stat("b) kernel + hipMemcpyAsync D2H 8 B + sync", t);
  for (int i = 0; i < N; i++) { const double t0 = now(); hipLaunchKernelGGL(k_dev, dim3(1), dim3(64), 0, s, d, (uint64_t)i); hipLaunchKernelGGL(k_dev, dim3(1), dim3(64), 0, s, d + 1, (uint64_t)i); CK(hipStreamSynchronize(s)); t[i] = now() - t0; }
  stat("a2) two kernels + sync", t);
  volatile uint64_t* hv = h;
  for (int i = 0; i < N; i++) {
    const double t0 = now();
    hipLaunchKernelGGL(k_host, dim3(1), dim3(64), 0, s, h, h + 8, (uint64_t)(i + 1));
    while (hv[8] != (uint64_t)(i + 1)) {}
    t[i] = now() - t0;
    if (hv[0] != (uint64_t)(i + 1) * 3) { printf("result not visible with the flag!\n"); return 1; }
  }
  stat("c) kernel writes pinned host memory, host spins on a flag", t);
  CK(hipStreamSynchronize(s));
  for (int i = 0; i < N; i++) {
    const double t0 = now();
    hipLaunchKernelGGL(k_dev, dim3(1), dim3(64), 0, s, d, (uint64_t)i);
    hipLaunchKernelGGL(k_host, dim3(1), dim3(64), 0, s, h, h + 8, (uint64_t)(N + i + 1));
    while (hv[8] != (uint64_t)(N + i + 1)) {}
    t[i] = now() - t0;
  }
  stat("c2) kernel, then a kernel that posts to pinned host memory; host spins", t);
  CK(hipStreamSynchronize(s));
  hipEvent_t ev;
  CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  for (int i = 0; i < N; i++) { const double t0 = now(); hipLaunchKernelGGL(k_dev, dim3(1), dim3(64), 0, s, d, (uint64_t)i); CK(hipEventRecord(ev, s)); while (hipEventQuery(ev) == hipErrorNotReady) {} t[i] = now() - t0; }
  stat("e) kernel + event record + spin on hipEventQuery", t);
  for (int i = 0; i < N; i++) { const double t0 = now(); hipLaunchKernelGGL(k_dev, dim3(1), dim3(64), 0, s, d, (uint64_t)i); while (hipStreamQuery(s) == hipErrorNotReady) {} t[i] = now() - t0; }
  stat("f) kernel + spin on hipStreamQuery", t);
  CK(hipStreamSynchronize(s));
  for (int i = 0; i < N; i++) { const double t0 = now(); hipLaunchKernelGGL(k_dev, dim3(1), dim3(64), 0, s, d, (uint64_t)i); CK(hipDeviceSynchronize()); t[i] = now() - t0; }
  stat("g) kernel + hipDeviceSynchronize", t);
  for (int i = 0; i < N; i++) { const double t0 = now(); CK(hipDeviceSynchronize()); t[i] = now() - t0; }
  stat("h) hipDeviceSynchronize, idle", t);
  for (int i = 0; i < N; i++) { const double t0 = now(); CK(hipStreamSynchronize(s)); t[i] = now() - t0; }
  stat("i) hipStreamSynchronize, idle", t);
  for (int i = 0; i < N; i++) {
    const double t0 = now();
    hipLaunchKernelGGL(k_host, dim3(1), dim3(64), 0, s, h, h + 8, (uint64_t)(2 * N + i + 1));
    while (hv[8] != (uint64_t)(2 * N + i + 1)) {}
    CK(hipStreamSynchronize(s));
    t[i] = now() - t0;
  }
  stat("k) as (c), then hipStreamSynchronize once the flag is there", t);
  for (int i = 0; i < N; i++) {
    const double t0 = now();
    hipLaunchKernelGGL(k_host, dim3(1), dim3(64), 0, s, h, h + 8, (uint64_t)(5 * N + i + 1));
    while (hv[8] != (uint64_t)(5 * N + i + 1)) {}
    CK(hipDeviceSynchronize());
    t[i] = now() - t0;
  }
  stat("l) as (c), then hipDeviceSynchronize once the flag is there", t);
  hipStream_t s2;
  CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  for (int i = 0; i < N; i++) { const double t0 = now(); hipLaunchKernelGGL(k_dev, dim3(1), dim3(64), 0, s, d, (uint64_t)i); CK(hipEventRecord(ev, s)); CK(hipStreamWaitEvent(s2, ev, 0));
    hipLaunchKernelGGL(k_host, dim3(1), dim3(64), 0, s2, h, h + 8, (uint64_t)(3 * N + i + 1)); while (hv[8] != (uint64_t)(3 * N + i + 1)) {} t[i] = now() - t0; }
  stat("j) kernel on s, event, s2 waits for it, posting kernel on s2; host spins", t);
  return 0;
}
