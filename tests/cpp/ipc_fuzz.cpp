// Sanitizer harness for the Arrow IPC reader: arrow_gpu_amd/csrc/arrow_ipc_reader.inc (plain C++, no HIP) is compiled INTO
// this program by g++ with -fsanitize=address,undefined (tests/test_sanitizers.py) — a CPU build, like the oracle's.  Reads an IPC stream / file from argv[1], then opens, walks and
// touches every readable column of (a) the pristine bytes, (b) every truncation on a coarse grid, (c) N random byte flips —
// each mutation on a heap copy of EXACTLY the mutated length, so that any read past the end is an ASan report.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include <cstdarg>
#include <memory>
#include <string>

#include "../../include/arrow_gpu.h"

// what csrc/common.hpp gives the library build
static char g_fuzz_err[512];
static void agpu_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_fuzz_err, sizeof(g_fuzz_err), fmt, ap);
  va_end(ap);
}
#define AGPU_REQUIRE(cond, code, msg)          \
  do {                                         \
    if (!(cond)) {                             \
      agpu_set_error("%s: %s", __func__, msg); \
      return code;                             \
    }                                          \
  } while (0)
#include "../../arrow_gpu_amd/csrc/arrow_ipc_reader.inc"

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t rnd() {
  rng_state ^= rng_state << 13;
  rng_state ^= rng_state >> 7;
  rng_state ^= rng_state << 17;
  return rng_state;
}

static uint64_t walk(const uint8_t* data, size_t n, int* opened) {
  agpu_ipc_reader* r = nullptr;
  uint64_t sum = 0;
  if (agpu_ipc_open(data, n, &r) != AGPU_OK) return 0;
  (*opened)++;
  int32_t nf = 0;
  int64_t nbat = 0;
  agpu_ipc_num_fields(r, &nf);
  agpu_ipc_num_batches(r, &nbat);
  for (int32_t i = 0; i < nf; i++) {
    agpu_ipc_field f;
    if (agpu_ipc_field_info(r, i, &f) == AGPU_OK) sum += strlen(f.name) + strlen(f.format);
  }
  for (int64_t b = 0; b < nbat; b++) {
    int64_t rows = 0;
    agpu_ipc_batch_rows(r, b, &rows);
    for (int32_t c = 0; c < nf; c++) {
      struct ArrowArray a;
      struct ArrowSchema s;
      if (agpu_ipc_column_view(r, b, c, &a, &s) != AGPU_OK) continue;
      agpu_ipc_field f;
      agpu_ipc_field_info(r, c, &f);
      const size_t w = f.dtype == AGPU_BOOL ? 0 : agpu_dtype_size((agpu_dtype)f.dtype);
      const size_t vbytes = w ? (size_t)a.length * w : (size_t)((a.length + 7) / 8);
      const uint8_t* values = static_cast<const uint8_t*>(a.buffers[1]);
      const uint8_t* validity = static_cast<const uint8_t*>(a.buffers[0]);
      for (size_t k = 0; k < vbytes; k++) sum += values[k];  // every byte the view claims must be inside `data`
      if (validity)
        for (size_t k = 0; k < (size_t)((a.length + 7) / 8); k++) sum += validity[k];
      a.release(&a);
      s.release(&s);
    }
  }
  agpu_ipc_close(r);
  return sum;
}

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  FILE* f = fopen(argv[1], "rb");
  if (!f) return 2;
  std::vector<uint8_t> src;
  uint8_t buf[65536];
  size_t k;
  while ((k = fread(buf, 1, sizeof(buf), f)) > 0) src.insert(src.end(), buf, buf + k);
  fclose(f);
  const int flips = argc > 2 ? atoi(argv[2]) : 2000;
  int opened = 0, cases = 0;
  uint64_t sum = 0;
  auto run = [&](const std::vector<uint8_t>& bytes) {  // exact-size heap copy: ASan guards both ends
    uint8_t* p = static_cast<uint8_t*>(malloc(bytes.size() ? bytes.size() : 1));
    if (!bytes.empty()) memcpy(p, bytes.data(), bytes.size());
    sum += walk(p, bytes.size(), &opened);
    free(p);
    cases++;
  };
  run(src);
  if (!opened) {
    fprintf(stderr, "the pristine input did not open\n");
    return 1;
  }
  for (size_t cut = 0; cut < src.size(); cut += (cut < 4096 ? 1 : 509)) run(std::vector<uint8_t>(src.begin(), src.begin() + (long)cut));
  for (int i = 0; i < flips; i++) {
    std::vector<uint8_t> m = src;
    const int nflip = 1 + (int)(rnd() % 4);
    for (int j = 0; j < nflip; j++) {
      // metadata lives at the head (schema, first batches) and — for files — at the tail (footer): bias towards both
      const uint64_t span = m.size() < 6000 ? m.size() : 6000;
      const size_t pos = (rnd() & 1) ? (size_t)(rnd() % span) : m.size() - 1 - (size_t)(rnd() % span);
      m[pos] = (uint8_t)rnd();
    }
    run(m);
  }
  // the LZ4 frame codec on its own: encode → decode must give the bytes back (runs, noise, tiny and multi-block inputs),
  // and decoding a truncated / flipped frame must fail or stay inside its buffers
  int codec_cases = 0;
  for (int t = 0; t < 120; t++) {
    size_t n = t < 30 ? (size_t)t : (size_t)(rnd() % 200000);
    if (t == 119) n = ((size_t)4 << 20) + 12345;
    std::vector<uint8_t> a(n), c, d;
    for (size_t i = 0; i < n; i++) a[i] = (t & 3) == 0 ? (uint8_t)rnd() : (t & 3) == 1 ? (uint8_t)(i / 100) : (t & 3) == 2 ? (uint8_t)((i % 7) * 3) : (uint8_t)(rnd() % 3);
    lz4_frame_encode(a.data(), n, &c);
    if (!lz4_frame_decode(c.data(), c.size(), n, &d) || d != a) {
      fprintf(stderr, "LZ4 round trip failed at case %d (n = %zu)\n", t, n);
      return 1;
    }
    for (int k = 0; k < 8 && !c.empty(); k++) {
      std::vector<uint8_t> m(c.begin(), c.begin() + (long)(rnd() % (c.size() + 1)));
      if (!m.empty() && (k & 1)) m[rnd() % m.size()] = (uint8_t)rnd();
      uint8_t* q = static_cast<uint8_t*>(malloc(m.size() ? m.size() : 1));
      if (!m.empty()) memcpy(q, m.data(), m.size());
      (void)lz4_frame_decode(q, m.size(), n, &d);
      free(q);
    }
    codec_cases++;
  }
  printf("ipc_fuzz OK: %d cases, %d opened, %d codec round trips, checksum %llu\n", cases, opened, codec_cases, (unsigned long long)sum);
  return 0;
}
