/*
 * cpu_baseline.c — the timed CPU baseline of bench.py's `cpu_baseline` leg.  TEST/BENCH INFRASTRUCTURE ONLY.
 *
 * north_star asks for arrow-rs's CPU compute kernels timed beside the GPU numbers; arrow-rs 54.2.1 is Rust
 * (the reference's benches call `arrow::compute::kernels::numeric::add` and `aggregate::sum`:
 * crates/benchmarks/benches/compare_gpu_arrow.rs:11, compare_sum.rs:10) and cannot be built here (no rustc).
 * These loops are a "port": the same single pass arrow-rs makes — one auto-vectorisable loop over the value
 * slices into a pre-allocated output plus a word-wise AND of the validity bitmaps (arrow-arith binary_op),
 * compare results packed 64 rows per u64 (arrow-ord cmp), sum with independent lane accumulators (arrow-arith
 * aggregate).  `threads` = 1 reproduces arrow-rs (its kernels are single-threaded); >1 splits rows statically with
 * OpenMP.  Results are checked against oracle/agpu_oracle.c in tests/test_cpu_baseline.py.
 */
#include <stdint.h>
#include <stddef.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static void range_of(int t, int nt, uint64_t n, uint64_t gran, uint64_t* lo, uint64_t* hi) {
  uint64_t chunks = (n + gran - 1) / gran;
  uint64_t per = (chunks + (uint64_t)nt - 1) / (uint64_t)nt;
  uint64_t a = (uint64_t)t * per * gran, b = a + per * gran;
  if (a > n) a = n;
  if (b > n) b = n;
  *lo = a; *hi = b;
}

/* out[i] = a[i] + b[i]; out_validity = va & vb (either may be NULL) */
void base_add_f32(const float* a, const float* b, float* out, const uint64_t* va, const uint64_t* vb,
                  uint64_t* out_validity, uint64_t n, int threads) {
  if (threads < 1) threads = 1;
#pragma omp parallel num_threads(threads)
  {
#ifdef _OPENMP
    int t = omp_get_thread_num(), nt = omp_get_num_threads();
#else
    int t = 0, nt = 1;
#endif
    uint64_t lo, hi;
    range_of(t, nt, n, 64, &lo, &hi);
    const float* __restrict__ pa = a; const float* __restrict__ pb = b; float* __restrict__ po = out;
    for (uint64_t i = lo; i < hi; i++) po[i] = pa[i] + pb[i];
    if (out_validity && va && vb)
      for (uint64_t w = lo / 64; w < (hi + 63) / 64; w++) out_validity[w] = va[w] & vb[w];
  }
}

/* out_bits bit i = a[i] == b[i] (64 rows per word); out_validity = va & vb */
void base_eq_i32(const int32_t* a, const int32_t* b, uint64_t* out_bits, const uint64_t* va, const uint64_t* vb,
                 uint64_t* out_validity, uint64_t n, int threads) {
  if (threads < 1) threads = 1;
#pragma omp parallel num_threads(threads)
  {
#ifdef _OPENMP
    int t = omp_get_thread_num(), nt = omp_get_num_threads();
#else
    int t = 0, nt = 1;
#endif
    uint64_t lo, hi;
    range_of(t, nt, n, 64, &lo, &hi);
    for (uint64_t w = lo / 64; w < (hi + 63) / 64; w++) {
      uint64_t base = w * 64, word = 0;
      uint64_t lim = n - base < 64 ? n - base : 64;
      if (lim == 64) {
        for (int k = 0; k < 64; k++) word |= (uint64_t)(a[base + (uint64_t)k] == b[base + (uint64_t)k]) << k;
      } else {
        for (uint64_t k = 0; k < lim; k++) word |= (uint64_t)(a[base + k] == b[base + k]) << k;
      }
      out_bits[w] = word;
      if (out_validity && va && vb) out_validity[w] = va[w] & vb[w];
    }
  }
}

/* lane-wise f32 sum (16 independent accumulators like arrow-rs's SIMD aggregate), returned in f32 */
float base_sum_f32(const float* a, uint64_t n, int threads) {
  if (threads < 1) threads = 1;
  double total = 0.0;
#pragma omp parallel num_threads(threads) reduction(+ : total)
  {
#ifdef _OPENMP
    int t = omp_get_thread_num(), nt = omp_get_num_threads();
#else
    int t = 0, nt = 1;
#endif
    uint64_t lo, hi;
    range_of(t, nt, n, 64, &lo, &hi);
    float acc[16];
    for (int k = 0; k < 16; k++) acc[k] = 0.0f;
    uint64_t i = lo;
    for (; i + 16 <= hi; i += 16)
      for (int k = 0; k < 16; k++) acc[k] += a[i + (uint64_t)k];
    float s = 0.0f;
    for (int k = 0; k < 16; k++) s += acc[k];
    for (; i < hi; i++) s += a[i];
    total += (double)s;
  }
  return (float)total;
}

/* The reference's own criterion workloads (crates/benchmarks/benches/compare_gpu_arrow.rs:18-43: f32 column + scalar,
 * 10 Mi rows, arrow::compute::kernels::numeric::add; compare_sum.rs:17-40: u32 sum, 1 Mi and 10 Mi rows,
 * arrow::compute::kernels::aggregate::sum — wrapping, single pass). */
void base_add_scalar_f32(const float* a, float s, float* out, uint64_t n) {
  for (uint64_t i = 0; i < n; i++) out[i] = a[i] + s;
}
uint32_t base_sum_u32(const uint32_t* a, uint64_t n) {
  uint32_t acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  uint64_t i = 0;
  for (; i + 8 <= n; i += 8)
    for (int k = 0; k < 8; k++) acc[k] += a[i + k];
  uint32_t t = 0;
  for (int k = 0; k < 8; k++) t += acc[k];
  for (; i < n; i++) t += a[i];
  return t;
}

/* First-touch a buffer with the SAME static row partition the kernels above use, so that on a multi-socket host every
 * thread's slice of every column lives on its own NUMA node (bench.py's all-core line; pages are placed where they are
 * first written).  gran = bytes per row-granule of the partition (64 rows of the column's element size). */
void base_first_touch(void* ptr, uint64_t bytes, uint64_t bytes_per_64_rows, int threads) {
  if (threads < 1) threads = 1;
  const uint64_t n64 = (bytes + bytes_per_64_rows - 1) / bytes_per_64_rows;  /* 64-row granules */
#pragma omp parallel num_threads(threads)
  {
#ifdef _OPENMP
    int t = omp_get_thread_num(), nt = omp_get_num_threads();
#else
    int t = 0, nt = 1;
#endif
    uint64_t lo, hi;
    range_of(t, nt, n64 * 64, 64, &lo, &hi);
    uint64_t b0 = lo / 64 * bytes_per_64_rows, b1 = hi / 64 * bytes_per_64_rows;
    if (b1 > bytes) b1 = bytes;
    for (uint64_t off = b0; off < b1; off += 4096) ((volatile char*)ptr)[off] = 0;
  }
}
