/* examples/simple.c — the reference's examples/simple.rs through the C ABI alone (what a cgo / JNI / Rust FFI shim
 * would call).  Plain C11: `gcc examples/simple.c -Iinclude -Larrow_gpu_amd/lib -larrow_gpu_hip -o simple`.
 * ((0..100) + 20) * 20 as two ordered kernels on one pipeline, then as ONE fused kernel.  Exits 2 without a gfx950
 * device (the library has no CPU fallback). */
#include <stdio.h>
#include <stdlib.h>

#include "arrow_gpu.h"

#define CHECK(call)                                                            \
  do {                                                                         \
    agpu_status s_ = (call);                                                   \
    if (s_ != AGPU_OK) {                                                       \
      fprintf(stderr, "%s failed (%d): %s\n", #call, (int)s_, agpu_last_error()); \
      return s_ == AGPU_ERR_NO_DEVICE ? 2 : 1;                                 \
    }                                                                          \
  } while (0)

int main(void) {
  enum { N = 100 };
  float host[N], back[N], twenty = 20.0f;
  agpu_device* dev = NULL;
  agpu_pipeline* p = NULL;
  void *x = NULL, *s = NULL, *t = NULL, *out = NULL, *fused = NULL;
  int i;
  for (i = 0; i < N; i++) host[i] = (float)i;

  CHECK(agpu_device_create(0, &dev));
  CHECK(agpu_pipeline_create(dev, &p));
  CHECK(agpu_malloc(dev, sizeof host, 0, &x));
  CHECK(agpu_malloc(dev, sizeof twenty, 0, &s));
  CHECK(agpu_malloc(dev, sizeof host, 0, &t));
  CHECK(agpu_malloc(dev, sizeof host, 0, &out));
  CHECK(agpu_malloc(dev, sizeof host, 0, &fused));
  CHECK(agpu_upload(p, x, host, sizeof host));
  CHECK(agpu_upload(p, s, &twenty, sizeof twenty));

  /* two ops, one pipeline, one finish() [simple.rs:45-72] */
  CHECK(agpu_scalar(p, AGPU_OP_ADD, AGPU_F32, x, s, t, N));
  CHECK(agpu_scalar(p, AGPU_OP_MUL, AGPU_F32, t, s, out, N));
  CHECK(agpu_pipeline_finish(p));
  CHECK(agpu_download(p, back, out, sizeof back));
  for (i = 0; i < N; i++)
    if (back[i] != (host[i] + 20.0f) * 20.0f) {
      fprintf(stderr, "mismatch at %d: %g\n", i, back[i]);
      return 1;
    }

  /* the same chain as ONE kernel */
  {
    agpu_chain_step steps[2];
    steps[0].op = AGPU_OP_ADD, steps[0].kind = AGPU_CHAIN_SCALAR, steps[0].operand = s;
    steps[1].op = AGPU_OP_MUL, steps[1].kind = AGPU_CHAIN_SCALAR, steps[1].operand = s;
    CHECK(agpu_fused_chain(p, AGPU_F32, x, steps, 2, fused, N));
    CHECK(agpu_download(p, back, fused, sizeof back));
    for (i = 0; i < N; i++)
      if (back[i] != (host[i] + 20.0f) * 20.0f) return 1;
  }

  CHECK(agpu_free(dev, x));
  CHECK(agpu_free(dev, s));
  CHECK(agpu_free(dev, t));
  CHECK(agpu_free(dev, out));
  CHECK(agpu_free(dev, fused));
  CHECK(agpu_pipeline_destroy(p));
  CHECK(agpu_device_destroy(dev));
  printf("simple.c OK: ((0..100) + 20) * 20, two kernels and one fused kernel agree\n");
  return 0;
}
