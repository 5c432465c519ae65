/* arrow_cdata.c — the Arrow C Data Interface at the C ABI, from plain C11 (what an arrow-rs `FFI_ArrowArray`, a cgo or a
 * JNI producer hands over): a hand-built, SLICED int32 ArrowArray with a validity bitmap goes to HBM with
 * agpu_import_arrow, `x + 20` runs on the GPU, agpu_export_arrow hands the result back as an ArrowArray whose release
 * callback frees it.  The reference can only build arrays from host Vecs and read them back as Vecs
 * [ref: crates/array/src/array/primitive_array_gpu.rs:22-104].
 *
 *   cc -std=c11 -Wall -Wextra -pedantic examples/arrow_cdata.c -Iinclude -Larrow_gpu_amd/lib -larrow_gpu_hip -o arrow_cdata */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/arrow_gpu.h"

#define CHECK(call)                                                                   \
  do {                                                                                \
    agpu_status s_ = (call);                                                          \
    if (s_ != AGPU_OK) {                                                              \
      printf("%s failed (%d): %s\n", #call, (int)s_, agpu_last_error());              \
      return s_ == AGPU_ERR_NO_DEVICE ? 2 : 1;                                        \
    }                                                                                 \
  } while (0)

static int released = 0;
static void release_array(struct ArrowArray* a) {
  released++;
  a->release = NULL;
}
static void release_schema(struct ArrowSchema* s) {
  released++;
  s->release = NULL;
}

int main(void) {
  agpu_device* dev = NULL;
  agpu_pipeline* p = NULL;
  agpu_status s = agpu_device_create(0, &dev);
  if (s == AGPU_ERR_NO_DEVICE) {
    printf("no device: %s\n", agpu_last_error());
    return 2;
  }
  CHECK(s);
  CHECK(agpu_pipeline_create(dev, &p));

  /* parent buffers: 20 values, element i valid unless i % 5 == 3; the array is the slice [3, 3 + 13) */
  enum { PARENT = 20, OFFSET = 3, LEN = 13 };
  int32_t values[PARENT];
  uint8_t validity[(PARENT + 7) / 8 + 8];
  memset(validity, 0, sizeof(validity));
  int64_t nulls = 0;
  for (int i = 0; i < PARENT; i++) {
    values[i] = 100 * i - 7;
    if (i % 5 != 3) validity[i / 8] |= (uint8_t)(1u << (i % 8));
    else if (i >= OFFSET && i < OFFSET + LEN) nulls++;
  }
  const void* buffers[2] = {validity, values};
  struct ArrowArray in;
  memset(&in, 0, sizeof(in));
  in.length = LEN;
  in.null_count = nulls;
  in.offset = OFFSET;
  in.n_buffers = 2;
  in.buffers = buffers;
  in.release = release_array;
  struct ArrowSchema schema;
  memset(&schema, 0, sizeof(schema));
  schema.format = "i";
  schema.name = "x";
  schema.flags = ARROW_FLAG_NULLABLE;
  schema.release = release_schema;

  agpu_arrow_column col;
  CHECK(agpu_import_arrow(p, &in, &schema, &col));
  if (col.dtype != AGPU_I32 || col.length != LEN || col.null_count != nulls || !col.validity) {
    printf("import produced the wrong column\n");
    return 1;
  }
  in.release(&in); /* the source may go as soon as the import returns; ownership stayed with the producer */
  schema.release(&schema);

  int32_t twenty = 20;
  void *scalar = NULL, *sum = NULL;
  CHECK(agpu_malloc(dev, 16, 0, &scalar));
  CHECK(agpu_malloc(dev, LEN * 4, 0, &sum));
  CHECK(agpu_upload(p, scalar, &twenty, 4));
  CHECK(agpu_scalar(p, AGPU_OP_ADD, AGPU_I32, col.values, scalar, sum, LEN));
  agpu_arrow_column out_col = col; /* same validity (scalar ops clone it [ref: crates/arithmetic/src/lib.rs:36-39]) */
  out_col.values = sum;
  out_col.values_bytes = LEN * 4;

  struct ArrowArray out;
  struct ArrowSchema out_schema;
  CHECK(agpu_export_arrow(p, &out_col, &out, &out_schema));
  int bad = 0;
  if (strcmp(out_schema.format, "i") != 0 || out.length != LEN || out.offset != 0 || out.n_buffers != 2 || out.null_count != nulls) bad++;
  const int32_t* ov = (const int32_t*)out.buffers[1];
  const uint8_t* ob = (const uint8_t*)out.buffers[0];
  for (int i = 0; i < LEN; i++) {
    const int src = i + OFFSET;
    const int valid = (ob[i / 8] >> (i % 8)) & 1;
    if (valid != (src % 5 != 3)) bad++;
    if (ov[i] != values[src] + 20) bad++;
  }
  for (int i = LEN; i < 16; i++)
    if ((ob[i / 8] >> (i % 8)) & 1) bad++; /* padding bits are zero */
  out.release(&out);
  out_schema.release(&out_schema);
  if (out.release != NULL || out_schema.release != NULL || released != 2) bad++;

  /* unsupported layouts are rejected, not misread */
  struct ArrowSchema utf8 = schema;
  utf8.format = "u";
  utf8.release = release_schema;
  in.release = release_array;
  agpu_arrow_column none;
  if (agpu_import_arrow(p, &in, &utf8, &none) != AGPU_ERR_UNSUPPORTED) bad++;

  CHECK(agpu_free(dev, scalar));
  CHECK(agpu_free(dev, sum));
  CHECK(agpu_arrow_column_free(dev, &col));
  CHECK(agpu_pipeline_destroy(p));
  CHECK(agpu_device_destroy(dev));
  printf(bad ? "FAILED: %d check(s)\n" : "arrow_cdata OK\n", bad);
  return bad ? 1 : 0;
}
