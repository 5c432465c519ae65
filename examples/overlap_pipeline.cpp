// overlap_pipeline.cpp — host → HBM → host with transfers and compute overlapped, from a compiled host over the C ABI
// (the native counterpart of arrow_gpu_amd/interop.py map_chunks; SURVEY §8f-1 "async H2D overlap").
// Two f32 columns in pageable host memory are cut into chunks; an uploader thread stages chunk k+1 into one of two device
// buffer sets on ITS pipeline while the main thread, on a second pipeline tied to the first by
// agpu_pipeline_wait_pipeline, adds chunk k and streams the result back.  H2D and D2H use opposite directions of the
// link.  The reference uploads whole Vecs before the first dispatch and reads back after the last
// [ref: crates/array/src/array/primitive_array_gpu.rs:22-74].
//
//   hipcc -std=c++17 -O2 -x c++ examples/overlap_pipeline.cpp -o overlap_pipeline -Larrow_gpu_amd/lib -larrow_gpu_hip -lpthread
//   ./overlap_pipeline [rows] [chunk_rows]        → one JSON line
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include <sys/mman.h>

#include "../include/arrow_gpu.h"

// Pageable memory crosses the link at the link's rate only when it sits on transparent huge pages: the runtime pins the
// user pages on the fly, and pinning 4 KiB pages caps the copy at ≈ 20 GB/s (measured: this program with std::vector
// buffers) against 56 GB/s on 2 MiB pages (numpy asks for them with madvise, which is why the Python host saw the full
// rate).  A compiled host should allocate its big column buffers like this:
static float* huge_alloc_f32(uint64_t n) {
  const size_t bytes = ((n * 4 + (2u << 20) - 1) / (2u << 20)) * (2u << 20);
  void* p = nullptr;
  if (posix_memalign(&p, 2u << 20, bytes) != 0) return nullptr;
  (void)madvise(p, bytes, MADV_HUGEPAGE);
  return static_cast<float*>(p);
}

#define CHECK(call)                                                                        \
  do {                                                                                     \
    agpu_status s_ = (call);                                                               \
    if (s_ != AGPU_OK) {                                                                   \
      fprintf(stderr, "%s failed (%d): %s\n", #call, (int)s_, agpu_last_error());          \
      exit(s_ == AGPU_ERR_NO_DEVICE ? 2 : 1);                                              \
    }                                                                                      \
  } while (0)

struct Semaphore {
  std::mutex m;
  std::condition_variable cv;
  int count;
  explicit Semaphore(int c) : count(c) {}
  void acquire() {
    std::unique_lock<std::mutex> l(m);
    cv.wait(l, [&] { return count > 0; });
    count--;
  }
  void release() {
    {
      std::lock_guard<std::mutex> l(m);
      count++;
    }
    cv.notify_one();
  }
};

int main(int argc, char** argv) {
  const uint64_t n = argc > 1 ? strtoull(argv[1], nullptr, 10) : (1ull << 27);
  const uint64_t chunk = argc > 2 ? strtoull(argv[2], nullptr, 10) : (1ull << 24);
  agpu_device* dev = nullptr;
  agpu_status s = agpu_device_create(0, &dev);
  if (s == AGPU_ERR_NO_DEVICE) {
    printf("no device: %s\n", agpu_last_error());
    return 2;
  }
  CHECK(s);
  agpu_pipeline *up = nullptr, *comp = nullptr;
  CHECK(agpu_pipeline_create(dev, &up));
  CHECK(agpu_pipeline_create(dev, &comp));
  float *a = huge_alloc_f32(n), *b = huge_alloc_f32(n), *out = huge_alloc_f32(n);
  if (!a || !b || !out) {
    fprintf(stderr, "host allocation failed\n");
    return 1;
  }
  for (uint64_t i = 0; i < n; i++) {
    a[i] = (float)(i % 1000) * 0.5f;
    b[i] = (float)(i % 777) - 300.0f;
  }
  memset(out, 0, n * 4);  // touch the destination: first-touch page faults are the OS's cost, not the link's
  void* in[2][2];
  void* res[2];
  for (int k = 0; k < 2; k++) {
    CHECK(agpu_malloc(dev, chunk * 4, 0, &in[k][0]));
    CHECK(agpu_malloc(dev, chunk * 4, 0, &in[k][1]));
    CHECK(agpu_malloc(dev, chunk * 4, 0, &res[k]));
  }
  const uint64_t nchunks = (n + chunk - 1) / chunk;
  Semaphore free_[2] = {Semaphore(1), Semaphore(1)}, ready[2] = {Semaphore(0), Semaphore(0)};
  const auto t0 = std::chrono::steady_clock::now();
  std::thread uploader([&] {
    for (uint64_t k = 0; k < nchunks; k++) {
      const int st = (int)(k % 2);
      free_[st].acquire();
      const uint64_t r0 = k * chunk, rows = r0 + chunk <= n ? chunk : n - r0;
      CHECK(agpu_staged_copy(up, in[st][0], a + r0, rows * 4, 1));
      CHECK(agpu_staged_copy(up, in[st][1], b + r0, rows * 4, 1));
      ready[st].release();
    }
  });
  for (uint64_t k = 0; k < nchunks; k++) {
    const int st = (int)(k % 2);
    ready[st].acquire();
    const uint64_t r0 = k * chunk, rows = r0 + chunk <= n ? chunk : n - r0;
    CHECK(agpu_pipeline_wait_pipeline(comp, up));
    CHECK(agpu_binary(comp, AGPU_OP_ADD, AGPU_F32, in[st][0], in[st][1], res[st], rows));
    CHECK(agpu_staged_copy(comp, res[st], out + r0, rows * 4, 0));
    free_[st].release();
  }
  uploader.join();
  CHECK(agpu_pipeline_sync(comp));
  const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  uint64_t bad = 0;
  for (uint64_t i = 0; i < n; i += 997) bad += out[i] != a[i] + b[i];
  bad += out[n - 1] != a[n - 1] + b[n - 1];
  for (int k = 0; k < 2; k++) {
    CHECK(agpu_free(dev, in[k][0]));
    CHECK(agpu_free(dev, in[k][1]));
    CHECK(agpu_free(dev, res[k]));
  }
  CHECK(agpu_pipeline_destroy(up));
  CHECK(agpu_pipeline_destroy(comp));
  CHECK(agpu_device_destroy(dev));
  free(a);
  free(b);
  printf("{\"what\": \"f32 add from and to pageable host memory, chunked, upload | compute + download overlapped (C ABI, 2 host threads)\", "
         "\"rows\": %llu, \"chunk_rows\": %llu, \"seconds\": %.4f, \"GBps_host_bytes\": %.1f, \"ok\": %s}\n",
         (unsigned long long)n, (unsigned long long)chunk, sec, 12.0 * (double)n / sec / 1e9, bad ? "false" : "true");
  free(out);
  return bad ? 1 : 0;
}
