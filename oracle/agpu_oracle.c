/*
 * agpu_oracle.c — CPU ORACLE for the hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C, single-threaded restatement of what psvri/arrow-gpu's WGSL kernels and host macros compute
 * (the reference is Rust + WGSL and cannot be built in this image: no rustc/cargo, no wgpu — see DESIGN.md).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this; the product library
 * (libarrow_gpu_hip.so) never links, loads or calls it.
 *
 * Pinning: the reference's own unit-test vectors (≈200 macro invocations + the hand-written tests) are committed
 * under tests/golden/ (extracted by tools/extract_golden.py) and tests/test_oracle_golden.py checks every one of
 * them against this file.  Transcendental f32 functions are pinned by the reference only to 1e-2 absolute
 * (crates/test_macros/src/lib.rs:88-109); here they are the f64 libm result rounded once to f32.
 *
 * Each function cites the reference file:line it follows.  Build: see oracle/Makefile (gcc -O2 -ffp-contract=off).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_OK 0
#define ORC_UNSUPPORTED 1
#define ORC_ARG 4

/* enum values mirror include/arrow_gpu.h */
enum { T_BOOL = 0, T_F32, T_U32, T_U16, T_U8, T_I32, T_I16, T_I8, T_DATE32 };
enum { OP_ADD = 0, OP_SUB, OP_MUL, OP_DIV, OP_REM, OP_MIN, OP_MAX, OP_AND, OP_OR, OP_XOR, OP_SHL, OP_SHR, OP_POW };
enum { UN_NEG = 0, UN_ABS, UN_NOT, UN_SQRT, UN_CBRT, UN_EXP, UN_EXP2, UN_LOG, UN_LOG2, UN_SIN, UN_COS, UN_ACOS, UN_SINH, UN_POPCOUNT };
enum { CMP_GT = 0, CMP_GTEQ, CMP_LT, CMP_LTEQ, CMP_EQ };
enum { RED_SUM = 0, RED_MIN, RED_MAX };

static inline int bit_get(const uint8_t* b, uint64_t i) { return (b[i >> 3] >> (i & 7)) & 1; }
static inline void bit_put(uint8_t* b, uint64_t i, int v) {
  if (v) b[i >> 3] |= (uint8_t)(1u << (i & 7));
  else b[i >> 3] &= (uint8_t)~(1u << (i & 7));
}
size_t orc_bitmap_bytes(uint64_t n_bits) { return (size_t)((n_bits + 63) / 64 * 8); }
size_t orc_dtype_size(int t) {
  switch (t) {
    case T_F32: case T_U32: case T_I32: case T_DATE32: return 4;
    case T_U16: case T_I16: return 2;
    case T_U8: case T_I8: return 1;
    default: return 0;
  }
}

/* ------------------------------------------------------------------ scalar semantics */

/* WGSL integer division/remainder never trap: x/0 = x, x%0 = 0, MIN/-1 = MIN, MIN%-1 = 0 (WGSL spec §"Arithmetic
 * expressions"); the reference's own div-by-zero tests are commented out (crates/arithmetic/src/i32.rs:200-209). */
static inline int32_t i32_div(int32_t x, int32_t y) {
  if (y == 0) return x;
  if (x == INT32_MIN && y == -1) return x;
  return x / y;
}
static inline int32_t i32_rem(int32_t x, int32_t y) {
  if (y == 0) return 0;
  if (x == INT32_MIN && y == -1) return 0;
  return x % y;
}
static inline uint32_t u32_div(uint32_t x, uint32_t y) { return y == 0 ? x : x / y; }
static inline uint32_t u32_rem(uint32_t x, uint32_t y) { return y == 0 ? 0 : x % y; }

/* f32 `%` = x - y*trunc(x/y), each step rounded to f32 (crates/arithmetic/compute_shaders/f32/scalar.wgsl:15-17) */
static inline float f32_rem(float x, float y) {
  volatile float q = x / y;
  volatile float t = truncf(q);
  volatile float m = y * t;
  return x - m;
}
/* min/max: the non-NaN operand wins; NaN only if both NaN (crates/compare/src/f32.rs:260-352); -0 < +0 */
static inline float f32_max(float a, float b) {
  if (isnan(a)) return b;
  if (isnan(b)) return a;
  if (a == b) return signbit(a) ? b : a;
  return a > b ? a : b;
}
static inline float f32_min(float a, float b) {
  if (isnan(a)) return b;
  if (isnan(b)) return a;
  if (a == b) return signbit(a) ? a : b;
  return a < b ? a : b;
}
/* pow(x,y) is exp2(y*log2(x)) on the reference's GPUs: negative (and NaN) bases give NaN — pinned by
 * crates/math/src/f32.rs:209-271 ("gpu -1.0 ** 0.0 gives NAN"); non-negative bases follow libm. */
static inline float f32_pow(float x, float y) {
  if (isnan(x) || isnan(y) || x < 0.0f || (x == 0.0f && signbit(x))) return NAN;
  return (float)pow((double)x, (double)y);
}
/* i32 power_: repeated wrapping multiply for p>=0, repeated WGSL-division of 1 for p<0
 * (crates/math/compute_shaders/i32/binary.wgsl:13-29), in closed form. */
static inline int32_t i32_pow(int32_t x, int32_t p) {
  if (p >= 0) {
    uint32_t r = 1, b = (uint32_t)x;
    uint32_t e = (uint32_t)p;
    while (e) { if (e & 1) r *= b; b *= b; e >>= 1; }
    return (int32_t)r;
  }
  if (p == INT32_MIN) return 1; /* WGSL abs(MIN) = MIN → loop body never runs */
  uint32_t k = (uint32_t)(-p);
  if (x == 0) return 1;                 /* 1/0 = 1 (x/0 = x) */
  if (x == 1) return 1;
  if (x == -1) return (k & 1) ? -1 : 1;
  return 0;                             /* 1/x truncates to 0 and stays there */
}
static inline float f32_cbrt(float x) { /* sign-split pow(|x|,1/3): crates/math/compute_shaders/f32/floatunary.wgsl:46-54 */
  return (float)cbrt((double)x);
}
/* f32 → u8: u32(x) truncates toward 0 and clamps to [0, 2^32-1] (NaN → 0), then % 256
 * (crates/cast/compute_shaders/f32/cast_u8.wgsl:13-20; test crates/cast/src/f32_cast.rs:40-48) */
static inline uint8_t f32_to_u8(float x) {
  uint32_t u;
  if (!(x > 0.0f)) u = 0;
  else if (x >= 4294967296.0f) u = 0xFFFFFFFFu;
  else u = (uint32_t)x;
  return (uint8_t)(u % 256u);
}
/* f32 → i8 / i16 / u16 / i32 / u32: REFERENCE-ABSENT (the reference's table has f32 → u8 only,
 * crates/cast/src/lib.rs:135-161; north_star asks for "i8/i16/u8/u16 <-> f32").  Defined by analogy with
 * cast_u8.wgsl: the WGSL conversion to the 32-bit integer of the target's signedness — u32(x) / i32(x): truncate
 * toward 0, clamp to that type's range, NaN -> 0 — then keep the low bits of the target width (for u8 that is the
 * reference's `% 256`). */
static inline uint32_t f32_to_u32_wgsl(float x) {
  if (!(x > 0.0f)) return 0u;
  if (x >= 4294967296.0f) return 0xFFFFFFFFu;
  return (uint32_t)x;
}
static inline int32_t f32_to_i32_wgsl(float x) {
  if (x != x) return 0;
  if (x >= 2147483648.0f) return INT32_MAX;
  if (x <= -2147483648.0f) return INT32_MIN;
  return (int32_t)x;
}

/* ------------------------------------------------------------------ binary / scalar element-wise
 * crates/arithmetic/compute_shaders/{f32,i32,u32}/array.wgsl + scalar.wgsl; compare/ * /min_max.wgsl;
 * logical/ * /{logical,shift}.wgsl; math/{f32/floatbinary,i32/binary}.wgsl. `stride_b` = 0 gives the scalar form. */
#define INT_BIN(NAME, T, UT, WT, SIGNED)                                                                   \
  static int NAME(int op, const T* a, const void* bv, T* out, uint64_t n, int stride_b) {                  \
    const T* b = (const T*)bv;                                                                             \
    const uint32_t* sh = (const uint32_t*)bv;                                                              \
    for (uint64_t i = 0; i < n; i++) {                                                                     \
      T x = a[i];                                                                                          \
      T y = (op == OP_SHL || op == OP_SHR) ? 0 : b[stride_b ? i : 0];                                      \
      T r;                                                                                                 \
      switch (op) {                                                                                        \
        case OP_ADD: r = (T)((UT)x + (UT)y); break;                                                        \
        case OP_SUB: r = (T)((UT)x - (UT)y); break;                                                        \
        case OP_MUL: r = (T)((UT)((WT)(UT)x * (WT)(UT)y)); break;                                          \
        case OP_DIV: r = SIGNED ? (T)i32_div((int32_t)x, (int32_t)y) : (T)u32_div((uint32_t)x, (uint32_t)y); break; \
        case OP_REM: r = SIGNED ? (T)i32_rem((int32_t)x, (int32_t)y) : (T)u32_rem((uint32_t)x, (uint32_t)y); break; \
        case OP_MIN: r = x < y ? x : y; break;                                                             \
        case OP_MAX: r = x > y ? x : y; break;                                                             \
        case OP_AND: r = (T)(x & y); break;                                                                \
        case OP_OR: r = (T)(x | y); break;                                                                 \
        case OP_XOR: r = (T)(x ^ y); break;                                                                \
        case OP_SHL: { /* shift on the 32-bit extension, amount mod 32, truncate to width */               \
          uint32_t s = sh[stride_b ? i : 0] & 31u;                                                         \
          r = (T)((uint32_t)(int32_t)x << s); break; }                                                     \
        case OP_SHR: {                                                                                     \
          uint32_t s = sh[stride_b ? i : 0] & 31u;                                                         \
          r = SIGNED ? (T)((int32_t)x >> s) : (T)((uint32_t)x >> s); break; }                              \
        case OP_POW: if (sizeof(T) == 4 && SIGNED) { r = (T)i32_pow((int32_t)x, (int32_t)y); break; }      \
                     return ORC_UNSUPPORTED;                                                               \
        default: return ORC_UNSUPPORTED;                                                                   \
      }                                                                                                    \
      out[i] = r;                                                                                          \
    }                                                                                                      \
    return ORC_OK;                                                                                         \
  }
INT_BIN(bin_i32, int32_t, uint32_t, uint64_t, 1)
INT_BIN(bin_u32, uint32_t, uint32_t, uint64_t, 0)
INT_BIN(bin_i16, int16_t, uint16_t, uint32_t, 1)
INT_BIN(bin_u16, uint16_t, uint16_t, uint32_t, 0)
INT_BIN(bin_i8, int8_t, uint8_t, uint32_t, 1)
INT_BIN(bin_u8, uint8_t, uint8_t, uint32_t, 0)

static int bin_f32(int op, const float* a, const float* b, float* out, uint64_t n, int stride_b) {
  for (uint64_t i = 0; i < n; i++) {
    float x = a[i], y = b[stride_b ? i : 0], r;
    switch (op) {
      case OP_ADD: r = x + y; break;
      case OP_SUB: r = x - y; break;
      case OP_MUL: r = x * y; break;
      case OP_DIV: r = x / y; break;
      case OP_REM: r = f32_rem(x, y); break;
      case OP_MIN: r = f32_min(x, y); break;
      case OP_MAX: r = f32_max(x, y); break;
      case OP_POW: r = f32_pow(x, y); break;
      default: return ORC_UNSUPPORTED;
    }
    out[i] = r;
  }
  return ORC_OK;
}

static int binary_impl(int op, int dtype, const void* a, const void* b, void* out, uint64_t n, int stride_b) {
  switch (dtype) {
    case T_F32: return bin_f32(op, (const float*)a, (const float*)b, (float*)out, n, stride_b);
    case T_I32: case T_DATE32: return bin_i32(op, (const int32_t*)a, b, (int32_t*)out, n, stride_b);
    case T_U32: return bin_u32(op, (const uint32_t*)a, b, (uint32_t*)out, n, stride_b);
    case T_I16: if (op == OP_DIV || op == OP_REM) return ORC_UNSUPPORTED; return bin_i16(op, (const int16_t*)a, b, (int16_t*)out, n, stride_b);
    case T_U16: if (op == OP_DIV || op == OP_REM) return ORC_UNSUPPORTED; return bin_u16(op, (const uint16_t*)a, b, (uint16_t*)out, n, stride_b);
    case T_I8: if (op == OP_DIV || op == OP_REM) return ORC_UNSUPPORTED; return bin_i8(op, (const int8_t*)a, b, (int8_t*)out, n, stride_b);
    case T_U8: if (op == OP_DIV || op == OP_REM) return ORC_UNSUPPORTED; return bin_u8(op, (const uint8_t*)a, b, (uint8_t*)out, n, stride_b);
    default: return ORC_UNSUPPORTED;
  }
}
int orc_binary(int op, int dtype, const void* a, const void* b, void* out, uint64_t n) {
  return binary_impl(op, dtype, a, b, out, n, 1);
}
/* scalar = 1-element array (crates/arithmetic/src/lib.rs:11-50: apply_scalar_function(&self.data, &value.data, ..)) */
int orc_scalar(int op, int dtype, const void* a, const void* scalar, void* out, uint64_t n) {
  return binary_impl(op, dtype, a, scalar, out, n, 0);
}

/* ------------------------------------------------------------------ unary */
static float f32_unary(int op, float x, int* ok) {
  switch (op) {
    case UN_NEG: return -x;                                  /* arithmetic/compute_shaders/f32/neg.wgsl:10-14 */
    case UN_ABS: return fabsf(x);                            /* math/compute_shaders/f32/floatunary.wgsl:40-44 */
    case UN_SQRT: return sqrtf(x);
    case UN_CBRT: return f32_cbrt(x);
    case UN_EXP: return (float)exp((double)x);
    case UN_EXP2: return (float)exp2((double)x);
    case UN_LOG: return (float)log((double)x);
    case UN_LOG2: return (float)log2((double)x);
    case UN_SIN: return (float)sin((double)x);               /* trigonometry/compute_shaders/f32/trigonometry.wgsl */
    case UN_COS: return (float)cos((double)x);
    case UN_ACOS: return (float)acos((double)x);
    case UN_SINH: return (float)sinh((double)x);             /* trigonometry/compute_shaders/f32/hyperbolic.wgsl */
    default: *ok = 0; return 0.0f;
  }
}
int orc_unary(int op, int dtype, const void* in, void* out, uint64_t n) {
  int ok = 1;
  if (dtype == T_F32) {
    const float* a = (const float*)in; float* o = (float*)out;
    if (op == UN_NOT) return ORC_UNSUPPORTED;
    for (uint64_t i = 0; i < n; i++) o[i] = f32_unary(op, a[i], &ok);
    return ok ? ORC_OK : ORC_UNSUPPORTED;
  }
  /* fused cast+trig on packed small ints → f32 (trigonometry/compute_shaders/{u8,i8,u16,i16}/ *.wgsl) */
  if (op == UN_SIN || op == UN_COS || op == UN_SINH) {
    float* o = (float*)out;
    for (uint64_t i = 0; i < n; i++) {
      float x;
      switch (dtype) {
        case T_U8: x = (float)((const uint8_t*)in)[i]; break;
        case T_I8: x = (float)((const int8_t*)in)[i]; break;
        case T_U16: x = (float)((const uint16_t*)in)[i]; break;
        case T_I16: x = (float)((const int16_t*)in)[i]; break;
        default: return ORC_UNSUPPORTED;
      }
      o[i] = f32_unary(op, x, &ok);
    }
    return ORC_OK;
  }
#define INT_UN(T, UT)                                                              \
  { const T* a = (const T*)in; T* o = (T*)out;                                     \
    for (uint64_t i = 0; i < n; i++) {                                             \
      T x = a[i];                                                                  \
      switch (op) {                                                                \
        case UN_NOT: o[i] = (T)~x; break; /* logical/ * /not.wgsl */               \
        case UN_NEG: o[i] = (T)(0 - (UT)x); break;                                 \
        case UN_ABS: o[i] = (T)(x < 0 ? (T)(0 - (UT)x) : x); break; /* math/i32/unary.wgsl: abs(MIN)=MIN */ \
        case UN_POPCOUNT: { UT u = (UT)x; int c = 0; while (u) { c += (int)(u & 1); u = (UT)(u >> 1); } /* logical/u32/countbitones.wgsl:9-15 countOneBits */ \
                            o[i] = (T)c; break; }                                  \
        default: return ORC_UNSUPPORTED;                                           \
      }                                                                            \
    } return ORC_OK; }
  switch (dtype) {
    case T_I32: case T_DATE32: INT_UN(int32_t, uint32_t)
    case T_U32: INT_UN(uint32_t, uint32_t)
    case T_I16: INT_UN(int16_t, uint16_t)
    case T_U16: INT_UN(uint16_t, uint16_t)
    case T_I8: INT_UN(int8_t, uint8_t)
    case T_U8: INT_UN(uint8_t, uint8_t)
    default: return ORC_UNSUPPORTED;
  }
}

/* ------------------------------------------------------------------ casts (crates/cast/src/lib.rs:135-161 table) */
static int64_t load_int(int t, const void* p, uint64_t i) {
  switch (t) {
    case T_U8: return ((const uint8_t*)p)[i];
    case T_I8: return ((const int8_t*)p)[i];
    case T_U16: return ((const uint16_t*)p)[i];
    case T_I16: return ((const int16_t*)p)[i];
    case T_U32: return ((const uint32_t*)p)[i];
    case T_I32: case T_DATE32: return ((const int32_t*)p)[i];
    default: return 0;
  }
}
static void store_int(int t, void* p, uint64_t i, int64_t v) {
  switch (t) {
    case T_U8: ((uint8_t*)p)[i] = (uint8_t)v; break;
    case T_I8: ((int8_t*)p)[i] = (int8_t)v; break;
    case T_U16: ((uint16_t*)p)[i] = (uint16_t)v; break;
    case T_I16: ((int16_t*)p)[i] = (int16_t)v; break;
    case T_U32: ((uint32_t*)p)[i] = (uint32_t)v; break;
    case T_I32: case T_DATE32: ((int32_t*)p)[i] = (int32_t)v; break;
    default: break;
  }
}
int orc_cast(int from, int to, const void* in, void* out, uint64_t n) {
  if (from == T_BOOL && to == T_F32) { /* cast/compute_shaders/boolean/cast_f32.wgsl:9-20 */
    for (uint64_t i = 0; i < n; i++) ((float*)out)[i] = bit_get((const uint8_t*)in, i) ? 1.0f : 0.0f;
    return ORC_OK;
  }
  if (from == T_F32 && to == T_U8) {
    for (uint64_t i = 0; i < n; i++) ((uint8_t*)out)[i] = f32_to_u8(((const float*)in)[i]);
    return ORC_OK;
  }
  if (from == T_F32 && to != T_F32 && to != T_BOOL && orc_dtype_size(to)) { /* reference-absent, see f32_to_*_wgsl */
    const int is_signed = (to == T_I8 || to == T_I16 || to == T_I32 || to == T_DATE32);
    for (uint64_t i = 0; i < n; i++) {
      const float x = ((const float*)in)[i];
      const uint32_t bits = is_signed ? (uint32_t)f32_to_i32_wgsl(x) : f32_to_u32_wgsl(x);
      switch (orc_dtype_size(to)) {
        case 1: ((uint8_t*)out)[i] = (uint8_t)bits; break;
        case 2: ((uint16_t*)out)[i] = (uint16_t)bits; break;
        default: ((uint32_t*)out)[i] = bits; break;
      }
    }
    return ORC_OK;
  }
  if (from == T_F32 || from == T_BOOL || to == T_BOOL) return ORC_UNSUPPORTED;
  size_t fs = orc_dtype_size(from), ts = orc_dtype_size(to);
  if (!fs || !ts || ts < fs) return ORC_UNSUPPORTED; /* the reference has no narrowing int casts */
  if (to == T_F32) { /* int → f32, exact for ≤16-bit sources; u32/i32 → f32 is not in the reference's table */
    if (fs == 4) return ORC_UNSUPPORTED;
    for (uint64_t i = 0; i < n; i++) ((float*)out)[i] = (float)load_int(from, in, i);
    return ORC_OK;
  }
  /* widen by SOURCE signedness, then reinterpret (i8 → u32 gives 0xFFFFFFFF for -1: crates/cast/src/i8_cast.rs:72-89) */
  for (uint64_t i = 0; i < n; i++) store_int(to, out, i, load_int(from, in, i));
  return ORC_OK;
}
/* BitCast u32 → f32 = byte copy (crates/cast/src/lib.rs:90-107) */
int orc_bitcast(int from, int to, const void* in, void* out, uint64_t n) {
  if (orc_dtype_size(from) != orc_dtype_size(to) || !orc_dtype_size(from)) return ORC_UNSUPPORTED;
  memcpy(out, in, n * orc_dtype_size(from));
  return ORC_OK;
}

/* broadcast (crates/array/compute_shaders/{f32,i32,u32}/broadcast.wgsl; Boolean: boolean_gpu.rs:173-194) */
int orc_broadcast(int dtype, uint32_t value_bits, void* out, uint64_t n) {
  if (dtype == T_BOOL) {
    size_t nb = orc_bitmap_bytes(n);
    memset(out, 0, nb);
    if (value_bits) for (uint64_t i = 0; i < n; i++) bit_put((uint8_t*)out, i, 1);
    return ORC_OK;
  }
  size_t s = orc_dtype_size(dtype);
  if (!s) return ORC_UNSUPPORTED;
  for (uint64_t i = 0; i < n; i++) memcpy((uint8_t*)out + i * s, &value_bits, s);
  return ORC_OK;
}

/* ------------------------------------------------------------------ compare → LSB-first bitmap
 * crates/compare/compute_shaders/ * /cmp.wgsl:22-75 (bit `lid%32` of word `gid/32`); padding bits written 0. */
#define CMP_LOOP(T)                                                                 \
  { const T* x = (const T*)a; const T* y = (const T*)b;                             \
    for (uint64_t i = 0; i < n; i++) {                                              \
      int r;                                                                        \
      switch (op) {                                                                 \
        case CMP_GT: r = x[i] > y[i]; break;                                        \
        case CMP_GTEQ: r = x[i] >= y[i]; break;                                     \
        case CMP_LT: r = x[i] < y[i]; break;                                        \
        case CMP_LTEQ: r = x[i] <= y[i]; break;                                     \
        case CMP_EQ: r = x[i] == y[i]; break;                                       \
        default: return ORC_ARG;                                                    \
      }                                                                             \
      if (r) o[i >> 3] |= (uint8_t)(1u << (i & 7));                                 \
    } return ORC_OK; }
int orc_compare(int op, int dtype, const void* a, const void* b, void* out_bits, uint64_t n) {
  uint8_t* o = (uint8_t*)out_bits;
  memset(o, 0, orc_bitmap_bytes(n));
  switch (dtype) {
    case T_F32: CMP_LOOP(float)
    case T_I32: case T_DATE32: CMP_LOOP(int32_t)
    case T_U32: CMP_LOOP(uint32_t)
    case T_I16: CMP_LOOP(int16_t)
    case T_U16: CMP_LOOP(uint16_t)
    case T_I8: CMP_LOOP(int8_t)
    case T_U8: CMP_LOOP(uint8_t)
    default: return ORC_UNSUPPORTED;
  }
}

/* ------------------------------------------------------------------ bitmaps
 * validity AND: crates/array/src/array/null_bit_buffer.rs:168-204 → crates/logical/compute_shaders/u32/logical.wgsl */
int orc_bitmap_binary(int op, const void* a, const void* b, void* out, uint64_t n_bits) {
  size_t nb = orc_bitmap_bytes(n_bits);
  const uint8_t *x = (const uint8_t*)a, *y = (const uint8_t*)b; uint8_t* o = (uint8_t*)out;
  for (size_t i = 0; i < nb; i++) {
    switch (op) {
      case OP_AND: o[i] = x[i] & y[i]; break;
      case OP_OR: o[i] = x[i] | y[i]; break;
      case OP_XOR: o[i] = x[i] ^ y[i]; break;
      default: return ORC_UNSUPPORTED;
    }
  }
  return ORC_OK;
}
/* out bits [0,n) = src bits [off, off+n), padding zero (re-aligning the bitmap of a sliced Arrow array; not in the
 * reference, whose arrays always start at bit 0) */
int orc_bitmap_copy_bits(const void* src, uint64_t off, void* out, uint64_t n_bits) {
  memset(out, 0, orc_bitmap_bytes(n_bits));
  for (uint64_t i = 0; i < n_bits; i++)
    if (bit_get((const uint8_t*)src, off + i)) bit_put((uint8_t*)out, i, 1);
  return ORC_OK;
}
int orc_bitmap_not(const void* in, void* out, uint64_t n_bits) { /* u32/not.wgsl:9-13 — flips padding too */
  size_t nb = orc_bitmap_bytes(n_bits);
  for (size_t i = 0; i < nb; i++) ((uint8_t*)out)[i] = (uint8_t)~((const uint8_t*)in)[i];
  return ORC_OK;
}
int orc_bitmap_popcount(const void* bits, uint64_t n_bits, uint64_t* out) { /* boolean.rs:120-146 (first n_bits only) */
  uint64_t c = 0;
  for (uint64_t i = 0; i < n_bits; i++) c += (uint64_t)bit_get((const uint8_t*)bits, i);
  *out = c;
  return ORC_OK;
}
int orc_bitmap_any(const void* bits, uint64_t n_bits, uint32_t* out) { /* boolean.rs:106-118 */
  uint64_t c; orc_bitmap_popcount(bits, n_bits, &c); *out = c ? 1u : 0u; return ORC_OK;
}
/* ((va & m) | (vb & ~m)) & vm, absent = all ones (crates/routines/src/merge.rs:17-86; fixture routines/src/f32.rs:14-65) */
int orc_bitmap_merge_validity(const void* va, const void* vb, const void* mask, const void* vmask, void* out, uint64_t n_bits) {
  if (!mask) return ORC_ARG;
  size_t nb = orc_bitmap_bytes(n_bits);
  for (size_t i = 0; i < nb; i++) {
    uint8_t m = ((const uint8_t*)mask)[i];
    uint8_t a = va ? ((const uint8_t*)va)[i] : 0xFF, b = vb ? ((const uint8_t*)vb)[i] : 0xFF;
    uint8_t vm = vmask ? ((const uint8_t*)vmask)[i] : 0xFF;
    ((uint8_t*)out)[i] = (uint8_t)(((a & m) | (b & (uint8_t)~m)) & vm);
  }
  return ORC_OK;
}

/* ------------------------------------------------------------------ reductions
 * f32 sum in the REFERENCE'S ORDER: per 256-element workgroup an adjacent-pair tree (s = 1,2,4..128:
 * shared[2*s*lid] += shared[2*s*lid + s]), out-of-range lanes contribute +0.0, one output per workgroup; repeated on
 * the workgroup sums until one value remains (crates/arithmetic/compute_shaders/f32/aggregate.wgsl:21-41,
 * crates/arithmetic/src/aggregate_kernels.rs:24-51). */
static void tree_level_f32(const float* in, uint64_t n, float* out) {
  uint64_t groups = (n + 255) / 256;
  float sh[256];
  for (uint64_t g = 0; g < groups; g++) {
    for (int l = 0; l < 256; l++) { uint64_t i = g * 256 + (uint64_t)l; sh[l] = i < n ? in[i] : 0.0f; }
    for (int s = 1; s < 256; s *= 2)
      for (int idx = 0; idx + s < 256; idx += 2 * s) sh[idx] = sh[idx] + sh[idx + s];
    out[g] = sh[0];
  }
}
static int validity_ok(const void* v, uint64_t i) { return !v || bit_get((const uint8_t*)v, i); }

int orc_reduce(int op, int dtype, const void* in, const void* validity, uint64_t n, void* out) {
  if (op == RED_SUM) {
    if (dtype == T_F32) {
      uint64_t len = n ? n : 1;
      float* cur = (float*)malloc(sizeof(float) * len);
      if (!cur) return ORC_ARG;
      for (uint64_t i = 0; i < n; i++) cur[i] = validity_ok(validity, i) ? ((const float*)in)[i] : 0.0f;
      if (n == 0) { cur[0] = 0.0f; }
      uint64_t m = n ? n : 1;
      do { /* the reference always runs at least one level */
        uint64_t g = (m + 255) / 256;
        float* nxt = (float*)malloc(sizeof(float) * g);
        tree_level_f32(cur, m, nxt);
        free(cur); cur = nxt; m = g;
      } while (m != 1);
      *(float*)out = cur[0];
      free(cur);
      return ORC_OK;
    }
    if (dtype == T_I32 || dtype == T_U32 || dtype == T_DATE32) { /* wrapping; order-free */
      uint32_t s = 0;
      for (uint64_t i = 0; i < n; i++) if (validity_ok(validity, i)) s += ((const uint32_t*)in)[i];
      *(uint32_t*)out = s;
      return ORC_OK;
    }
    return ORC_UNSUPPORTED;
  }
  if (op != RED_MIN && op != RED_MAX) return ORC_ARG;
  int is_max = op == RED_MAX;
  switch (dtype) { /* Arrow min_max: NaN skipped unless nothing else; empty → identity */
    case T_F32: {
      float r = is_max ? -INFINITY : INFINITY; int seen = 0, seen_nan = 0;
      for (uint64_t i = 0; i < n; i++) {
        if (!validity_ok(validity, i)) continue;
        float x = ((const float*)in)[i];
        if (isnan(x)) { seen_nan = 1; continue; }
        seen = 1; r = is_max ? f32_max(r, x) : f32_min(r, x);
      }
      if (!seen && seen_nan) r = NAN;
      *(float*)out = r; return ORC_OK; }
    case T_I32: case T_DATE32: {
      int32_t r = is_max ? INT32_MIN : INT32_MAX;
      for (uint64_t i = 0; i < n; i++) if (validity_ok(validity, i)) { int32_t x = ((const int32_t*)in)[i]; r = is_max ? (x > r ? x : r) : (x < r ? x : r); }
      *(int32_t*)out = r; return ORC_OK; }
    case T_U32: {
      uint32_t r = is_max ? 0u : UINT32_MAX;
      for (uint64_t i = 0; i < n; i++) if (validity_ok(validity, i)) { uint32_t x = ((const uint32_t*)in)[i]; r = is_max ? (x > r ? x : r) : (x < r ? x : r); }
      *(uint32_t*)out = r; return ORC_OK; }
    default: return ORC_UNSUPPORTED;
  }
}
/* f64-accumulated sum of f32 (the multi-GPU partial): plain left-to-right in double — compare with a tolerance */
int orc_reduce_sum_f64(const float* in, const void* validity, uint64_t n, double* out) {
  double s = 0.0;
  for (uint64_t i = 0; i < n; i++) if (validity_ok(validity, i)) s += (double)in[i];
  *out = s;
  return ORC_OK;
}

/* ------------------------------------------------------------------ swizzle (crates/routines) */
int orc_take(int width, const void* values, uint64_t n_values, const uint32_t* idx, void* out, uint64_t n_idx) {
  if (width != 1 && width != 2 && width != 4) return ORC_UNSUPPORTED;
  for (uint64_t i = 0; i < n_idx; i++) { /* 32bit/take.wgsl:13-17 */
    if (idx[i] >= n_values) return ORC_ARG;
    memcpy((uint8_t*)out + i * (size_t)width, (const uint8_t*)values + (size_t)idx[i] * (size_t)width, (size_t)width);
  }
  return ORC_OK;
}
int orc_take_bits(const void* bits, uint64_t n_bits, const uint32_t* idx, void* out_bits, uint64_t n_idx) {
  memset(out_bits, 0, orc_bitmap_bytes(n_idx)); /* bool/take.wgsl:13-33 */
  for (uint64_t i = 0; i < n_idx; i++) {
    if (idx[i] >= n_bits) return ORC_ARG;
    if (bit_get((const uint8_t*)bits, idx[i])) bit_put((uint8_t*)out_bits, i, 1);
  }
  return ORC_OK;
}
int orc_put(int width, const void* src, const uint32_t* src_idx, void* dst, const uint32_t* dst_idx, uint64_t n) {
  if (width != 1 && width != 2 && width != 4) return ORC_UNSUPPORTED;
  for (uint64_t i = 0; i < n; i++) /* 32bit/put.wgsl:17-23 */
    memcpy((uint8_t*)dst + (size_t)dst_idx[i] * (size_t)width, (const uint8_t*)src + (size_t)src_idx[i] * (size_t)width, (size_t)width);
  return ORC_OK;
}
int orc_put_bits(const void* src_bits, const uint32_t* src_idx, void* dst_bits, const uint32_t* dst_idx, uint64_t n) {
  for (uint64_t i = 0; i < n; i++) /* bool/put.wgsl:17-34 */
    bit_put((uint8_t*)dst_bits, dst_idx[i], bit_get((const uint8_t*)src_bits, src_idx[i]));
  return ORC_OK;
}
int orc_merge(int width, const void* a, const void* b, const void* mask_bits, void* out, uint64_t n) {
  if (width != 1 && width != 2 && width != 4) return ORC_UNSUPPORTED;
  for (uint64_t i = 0; i < n; i++) { /* {32,16,8}bit/merge.wgsl: mask bit ? left : right */
    const void* s = bit_get((const uint8_t*)mask_bits, i) ? a : b;
    memcpy((uint8_t*)out + i * (size_t)width, (const uint8_t*)s + i * (size_t)width, (size_t)width);
  }
  return ORC_OK;
}
int orc_merge_bits(const void* a, const void* b, const void* mask_bits, void* out, uint64_t n_bits) {
  size_t nb = orc_bitmap_bytes(n_bits); /* bool/merge.wgsl:17-21 */
  for (size_t i = 0; i < nb; i++) {
    uint8_t m = ((const uint8_t*)mask_bits)[i];
    ((uint8_t*)out)[i] = (uint8_t)((((const uint8_t*)a)[i] & m) | (((const uint8_t*)b)[i] & (uint8_t)~m));
  }
  return ORC_OK;
}
int orc_index_max(const uint32_t* idx, uint64_t n, uint32_t* out) {
  uint32_t m = 0; for (uint64_t i = 0; i < n; i++) if (idx[i] > m) m = idx[i];
  *out = m; return ORC_OK;
}

/* ------------------------------------------------------------------ synthetic columns + checksum
 * (not reference code: the bench/parity input generator shared with the HIP library, include/arrow_gpu.h) */
static inline uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
static inline uint64_t row_hash(uint64_t seed, uint64_t row) { return splitmix64(seed ^ (row * 0x9E3779B97F4A7C15ull)); }
int orc_synth_f32(float* out, uint64_t n, uint64_t seed, uint64_t row0, float lo, float hi) {
  for (uint64_t i = 0; i < n; i++) {
    volatile float u = (float)(row_hash(seed, row0 + i) >> 40) * 0x1p-24f;
    volatile float w = (hi - lo) * u;
    out[i] = lo + w;
  }
  return ORC_OK;
}
int orc_synth_i32(int32_t* out, uint64_t n, uint64_t seed, uint64_t row0, uint32_t modulus) {
  for (uint64_t i = 0; i < n; i++) {
    uint32_t v = (uint32_t)(row_hash(seed, row0 + i) >> 32);
    out[i] = (int32_t)(modulus ? v % modulus : v);
  }
  return ORC_OK;
}
int orc_synth_u8(uint8_t* out, uint64_t n, uint64_t seed, uint64_t row0) {
  for (uint64_t i = 0; i < n; i++) out[i] = (uint8_t)(row_hash(seed, row0 + i) >> 56);
  return ORC_OK;
}
int orc_synth_bits(void* out_bits, uint64_t n_bits, uint64_t seed, uint64_t row0, double p_set) {
  memset(out_bits, 0, orc_bitmap_bytes(n_bits));
  for (uint64_t i = 0; i < n_bits; i++) {
    double u = (double)(row_hash(seed, row0 + i) >> 11) * 0x1p-53;
    if (u < p_set) bit_put((uint8_t*)out_bits, i, 1);
  }
  return ORC_OK;
}
/* sum over 8-byte words w_k of splitmix64(w_k ^ k); a trailing partial word is zero-extended */
int orc_checksum(const void* data, uint64_t bytes, uint64_t* out) {
  uint64_t s = 0, nw = bytes / 8;
  const uint8_t* p = (const uint8_t*)data;
  for (uint64_t k = 0; k < nw; k++) { uint64_t w; memcpy(&w, p + k * 8, 8); s += splitmix64(w ^ k); }
  if (bytes & 7) { uint64_t w = 0; memcpy(&w, p + nw * 8, bytes & 7); s += splitmix64(w ^ nw); }
  *out = s;
  return ORC_OK;
}
