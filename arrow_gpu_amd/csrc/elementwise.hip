// elementwise.hip — streaming element-wise kernels: binary, scalar, unary, shifts, casts, broadcast.
//
// Replaces the one-invocation-per-32-bit-word WGSL entry points of
//   crates/arithmetic/compute_shaders/{f32,i32,u32,u16}/{array,scalar,neg}.wgsl
//   crates/compare/compute_shaders/*/min_max.wgsl
//   crates/logical/compute_shaders/*/{logical,not,shift}.wgsl
//   crates/math/compute_shaders/{f32/floatunary,f32/floatbinary,i32/unary,i32/binary}.wgsl
//   crates/trigonometry/compute_shaders/*/{trigonometry,hyperbolic}.wgsl
//   crates/cast/compute_shaders/**, crates/array/compute_shaders/*/broadcast.wgsl
// and the launch shape the Rust macros compute (ceil(n/256) workgroups, no tail guard:
// crates/arithmetic/src/lib.rs:24,67).
//
// MI355X design: every kernel is HBM-bound (roofline = 8 TB/s HBM3E; algorithmic bytes per row are listed in
// DESIGN.md).  Each lane moves 16-byte vectors (global_load/store_dwordx4 = 1 KiB per wave instruction) with the
// nontemporal hint (every byte is touched once), AGPU_STREAM_U vectors per input array per lane, and — measured,
// profiles/r01_sweep_add_f32_1e9.json — ONE 4 KiB-per-array tile per 256-thread block: ~10^6 small blocks at 1e9 rows
// beat every persistent grid on this chip (6.5 vs 5.3 TB/s at 2048 blocks).  The kernels still grid-stride, so a
// capped grid (agpu_set_tuning "stream_grid") stays correct.
// Sub-word columns are read natively as packed bytes/halfwords (the WGSL u8/i8/u16/i16 unpack helpers vanish).
// No LDS, no MFMA: there is no reuse and no contraction on this path.
#include <type_traits>

#include "common.hpp"

// tiles per block for the kernels that stage a table in LDS (lut8, pow): tuning key "tiles"
// tiles per block of the kernels that stage a table in LDS; `dflt` = the kernel's default: 1 for the HBM-bound ones (lut8,
// pow array ∘ array — see tile_run below), 3 for pow with a scalar exponent (launch_pow_f32)
// A block that takes several tiles walks them A GRID APART (tile b, b + G, b + 2G …) and issues the next tile's loads before it
// evaluates the current one.  What that buys was measured at length in round 4 (tools/probe/prefetch_sweep.py, prefetch_context.py,
// grid_bits.py, cast_offsets.py, clock_state.py → profiles/r04_prefetch_*.json, DESIGN.md §4 "Tiles per block"):
//   * kernels whose bound is VALU issue (a cast-headed chain with a transcendental step, pow with a scalar exponent) gain 10–15 %
//     from 3–8 tiles per block on every box and in every layout — their defaults below;
//   * the HBM-bound ones (widening casts, the LDS-table kernels, sin / cos / sinh) gain 4–6 % at 2–4 tiles per block in SOME
//     allocations (cast u8→f32 0.80 → 0.84, sin_u8 0.77 → 0.84, sin f32 0.785 → 0.82) and LOSE 7–9 % in others — the same library,
//     the same box, another process: it follows what the driver backed the buffers with (the allocation lottery of DESIGN.md §3), not
//     the distance between the windows (every G from +1 to +2^19 tiles behaves alike), the relative position of input and output,
//     the timing method or the clock state.  One tile per block is insensitive to it (±1 %), so that is their default; the tuning
//     "tiles" stays for callers who measure their own allocation;
//   * CONTIGUOUS runs (block b owns [b·k, b·k + k)) lose 10–13 % everywhere: neighbouring waves then write every k-th 1–4 KiB tile at a
//     time, and address bits 12 / 13 feed the channel hash (profiles/r04_prefetch_sweep_contiguous_runs.json).
struct TileRun {
  uint64_t t, end, step;
};
__device__ __forceinline__ TileRun tile_run(uint64_t unit, uint64_t n_units, uint64_t ntiles) {
  return TileRun{unit, ntiles, n_units};
}
// units a launch needs for `tiles` tiles at k tiles per unit
static inline uint64_t tile_units(uint64_t tiles, uint64_t k) { return (tiles + k - 1) / k; }
static inline uint64_t tab_k(const agpu_pipeline* p, uint64_t dflt = 1) { return p->tune.tiles > 0 ? (uint64_t)p->tune.tiles : dflt; }

#ifndef AGPU_STREAM_U
#define AGPU_STREAM_U 1  // 16-byte vectors per lane per input array per tile (measured best: profiles/r01_sweep_add_f32_1e9.json)
#endif
#ifndef AGPU_STREAM_NT
#define AGPU_STREAM_NT 3  // bit0 nontemporal loads, bit1 nontemporal stores (both: +7 % at 1e9 rows)
#endif

enum { MODE_UNARY = 0, MODE_BINARY = 1, MODE_SCALAR = 2 };

template <typename T, int N>
struct PackN {
  T v[N];
};
// Loads/stores go through the NATIVE vector type of T (float4, char16, short8 …), not a bit-cast u32x4: when the
// value is bit-cast first, LLVM rewrites the store and DROPS the !nontemporal hint (the `nt` bit vanished from
// global_store_dwordx4 and cost 7 % of the add bandwidth — caught by diffing the ISA against tools/probe).
template <typename T, int N>
struct VecOf {
  typedef T type __attribute__((ext_vector_type(N)));
};

// 8-bit elements are the exception on the LOAD side: legalising a <16 x i8> load LLVM re-creates it as <4 x i32> and
// drops !nontemporal (every u8 / i8 kernel ran 7 % below the same bytes read as i32 — tools/probe/subword_vs_word.py),
// while a load that is dword-typed from the start — and stays so past InstCombine — keeps the bit; the bytes are re-typed
// in registers.
template <bool NT, typename T, int N>
__device__ __forceinline__ PackN<T, N> load_pack(const T* p) {
  if constexpr (NT && sizeof(T) == 1 && (N % 4) == 0) {
    typedef typename VecOf<uint32_t, N / 4>::type W;
    const W r0 = __builtin_nontemporal_load(reinterpret_cast<const W*>(p));
    struct Words {
      uint32_t w[N / 4];
    } r;
#pragma unroll
    for (int k = 0; k < N / 4; k++) {
      uint32_t x = r0[k];
#if defined(__HIP_DEVICE_COMPILE__)  // the host pass of this translation unit has no "v" registers
      asm volatile("" : "+v"(x));  // opaque to InstCombine: load + bitcast would be merged back into the <N x i8> load
#endif
      r.w[k] = x;
    }
    return __builtin_bit_cast(PackN<T, N>, r);
  } else {
    typedef typename VecOf<T, N>::type V;
    const V r = ld_vec<NT>(reinterpret_cast<const V*>(p));
    PackN<T, N> out;
#pragma unroll
    for (int k = 0; k < N; k++) out.v[k] = r[k];
    return out;
  }
}
#ifndef AGPU_USE_SC1
#define AGPU_USE_SC1 1
#endif
template <bool NT, typename T, int N, bool SC1 = false>
__device__ __forceinline__ void store_pack(T* p, const PackN<T, N>& v) {
  typedef typename VecOf<T, N>::type V;
  V r;
#pragma unroll
  for (int k = 0; k < N; k++) r[k] = v.v[k];
  if constexpr (SC1 && NT && sizeof(V) == 16 && AGPU_USE_SC1) st_vec_sc1(reinterpret_cast<V*>(p), r);
  else st_vec<NT>(reinterpret_cast<V*>(p), r);
}

// ---------------------------------------------------------------- scalar semantics (mirrors oracle/agpu_oracle.c)
template <typename T> using U_of = typename std::make_unsigned<T>::type;

__device__ __forceinline__ float f32_max_dev(float a, float b) {  // NaN-ignoring, -0 < +0 [compare/src/f32.rs:260-352]
  if (a != a) return b;
  if (b != b) return a;
  if (a == b) return __builtin_signbit(a) ? b : a;
  return a > b ? a : b;
}
__device__ __forceinline__ float f32_min_dev(float a, float b) {
  if (a != a) return b;
  if (b != b) return a;
  if (a == b) return __builtin_signbit(a) ? a : b;
  return a < b ? a : b;
}
__device__ __forceinline__ int32_t i32_pow_dev(int32_t x, int32_t p) {  // math/compute_shaders/i32/binary.wgsl:13-29
  if (p >= 0) {
    uint32_t r = 1, b = (uint32_t)x, e = (uint32_t)p;
    while (e) {
      if (e & 1) r *= b;
      b *= b;
      e >>= 1;
    }
    return (int32_t)r;
  }
  if (p == INT32_MIN) return 1;
  uint32_t k = (uint32_t)(-p);
  if (x == 0 || x == 1) return 1;
  if (x == -1) return (k & 1) ? -1 : 1;
  return 0;
}

struct OpAdd {
  template <typename T> __device__ static __forceinline__ T ap(T x, T y) {
    if constexpr (std::is_floating_point<T>::value) return x + y;
    else return (T)((U_of<T>)x + (U_of<T>)y);
  }
};
struct OpSub {
  template <typename T> __device__ static __forceinline__ T ap(T x, T y) {
    if constexpr (std::is_floating_point<T>::value) return x - y;
    else return (T)((U_of<T>)x - (U_of<T>)y);
  }
};
struct OpMul {
  template <typename T> __device__ static __forceinline__ T ap(T x, T y) {
    if constexpr (std::is_floating_point<T>::value) return x * y;
    else return (T)((uint32_t)(U_of<T>)x * (uint32_t)(U_of<T>)y);
  }
};
struct OpDiv {  // WGSL: x/0 = x, MIN/-1 = MIN; f32: correctly rounded
  template <typename T> __device__ static __forceinline__ T ap(T x, T y) {
    if constexpr (std::is_floating_point<T>::value) return x / y;
    else if constexpr (std::is_signed<T>::value) {
      if (y == 0) return x;
      if (x == INT32_MIN && y == -1) return x;
      return x / y;
    } else return y == 0 ? x : x / y;
  }
};
struct OpRem {  // WGSL: x%0 = 0, MIN%-1 = 0; f32: x - y*trunc(x/y), each step rounded
  template <typename T> __device__ static __forceinline__ T ap(T x, T y) {
    if constexpr (std::is_floating_point<T>::value) {
      float q = __fdiv_rn(x, y);
      float t = truncf(q);
      float m = __fmul_rn(y, t);
      return __fsub_rn(x, m);
    } else if constexpr (std::is_signed<T>::value) {
      if (y == 0) return 0;
      if (x == INT32_MIN && y == -1) return 0;
      return x % y;
    } else return y == 0 ? 0 : x % y;
  }
};
struct OpMin {
  template <typename T> __device__ static __forceinline__ T ap(T x, T y) {
    if constexpr (std::is_floating_point<T>::value) return f32_min_dev(x, y);
    else return x < y ? x : y;
  }
};
struct OpMax {
  template <typename T> __device__ static __forceinline__ T ap(T x, T y) {
    if constexpr (std::is_floating_point<T>::value) return f32_max_dev(x, y);
    else return x > y ? x : y;
  }
};
struct OpAnd { template <typename T> __device__ static __forceinline__ T ap(T x, T y) { return (T)(x & y); } };
struct OpOr { template <typename T> __device__ static __forceinline__ T ap(T x, T y) { return (T)(x | y); } };
struct OpXor { template <typename T> __device__ static __forceinline__ T ap(T x, T y) { return (T)(x ^ y); } };
// 32-bit shifts ride the same stream kernel: the u32 shift-amount column is read as T (same width), amount mod 32
// [logical/compute_shaders/{u32,i32}/shift.wgsl]; sub-word columns keep shift_kernel (4-byte amounts per 1–2-byte row)
struct OpShl {
  template <typename T> __device__ static __forceinline__ T ap(T x, T y) { return (T)((uint32_t)x << ((uint32_t)y & 31u)); }
};
struct OpShr {
  template <typename T> __device__ static __forceinline__ T ap(T x, T y) {
    if constexpr (std::is_signed<T>::value) return (T)((int32_t)x >> ((uint32_t)y & 31u));
    else return (T)((uint32_t)x >> ((uint32_t)y & 31u));
  }
};
struct OpPow {  // i32: closed form of the WGSL loop.  (f32 pow runs in pow_kernel below; the float branch here is unused)
  template <typename T> __device__ static __forceinline__ T ap(T x, T y) {
    if constexpr (std::is_floating_point<T>::value) {
      if (x != x || y != y || x < 0.0f || (x == 0.0f && __builtin_signbit(x))) return __builtin_nanf("");
      return powf(x, y);  // ≤ 1 ULP measured (tests/tools/math_ulp.py); the f64 pow is 4× slower than the stream
    } else return (T)i32_pow_dev((int32_t)x, (int32_t)y);
  }
};

struct UnNeg {
  template <typename T> __device__ static __forceinline__ T ap(T x, T) {
    if constexpr (std::is_floating_point<T>::value) return -x;
    else return (T)((U_of<T>)0 - (U_of<T>)x);
  }
};
struct UnAbs {
  template <typename T> __device__ static __forceinline__ T ap(T x, T) {
    if constexpr (std::is_floating_point<T>::value) return fabsf(x);
    else if constexpr (std::is_signed<T>::value) return x < 0 ? (T)((U_of<T>)0 - (U_of<T>)x) : x;
    else return x;
  }
};
struct UnNot { template <typename T> __device__ static __forceinline__ T ap(T x, T) { return (T)~x; } };
struct UnPopc {  // countOneBits per element [logical/compute_shaders/u32/countbitones.wgsl:9-15]
  template <typename T> __device__ static __forceinline__ T ap(T x, T) { return (T)__builtin_popcount((uint32_t)(U_of<T>)x); }
};
// ---- transcendental f32 functions.  log / sinh / pow below are evaluated in f64 and rounded ONCE to f32 (≤ 0.5 ULP + 2^-30);
// the f32 device-library versions measure 2 ULP for sin / cos / log on gfx950 (profiles/r01_probe_first_contact.json).
//
// sin / cos (round 6): ALL in f32, TWO ROWS per instruction.  Rounds 1–5 evaluated the Cody–Waite reduction and the outer
// polynomial step in f64 — 8 f64 operations + 4 conversions + 5 selects per row, ≈ 27 VALU instructions, which an 8 B/row
// stream hides and a 5–6 B/row cast-headed chain does not (`cast i16 → f32 → sin` in one launch 0.65 of the roof, `cast(u16)·s
// → sin` 0.54: VERDICT r5 weak #3).  gfx950 issues v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 — two IEEE f32 operations per
// lane — at ≈ 1.45× the cost of ONE (profiles/r02_valu_rate.json), so the same function written as double-float f32 arithmetic
// over a PAIR of rows costs ≈ 11 packed + 6 scalar instructions per row:
//   k  = round(x · 2/π) by the 1.5·2^23 trick (one fma; the low bits of the sum ARE k, no conversion), |k| < 2^20 for |x| < 1e6
//   r1 = fma(−k, P1, x)                      EXACT: P1 = f32(π/2), k·P1 and x are multiples of 2^-24 and |r1| < 1
//   ph + pl = k·P2 exactly (fma residual), (rh, e) = Fast2Sum(r1, −ph) — valid although |r1| may be < |ph|: r1 is a multiple
//   of ulp(ph) — and lo = e − pl − k·P3:     r = rh + lo to ≈ 2^-48 relative, P1 + P2 + P3 = π/2 to 2^-76
//   sin r = rh − (rh·z·(−ps(z)) − lo),  cos r = 1 + (z·pc(z) − rh·lo),  z = rh²; ps / pc = minimax polynomials with 3 / 4
//   coefficients on |r| ≤ 0.89 (k is rounded in f32, so |r| can exceed π/4 by 0.06·π/2 at |x| = 1e6): 0.16 / 0.007 ULP
// then the quadrant picks S or C and the sign.  Both functions of both rows are evaluated (selecting coefficient sets per row
// would cost more than it saves).  Error against the true value ≤ 1.18 ULP, i.e. ALWAYS within 1 ULP of the correctly rounded
// result — the oracle's definition — checked for every f32 with |x| < 1e6, sin and cos, first on the CPU with the same operation
// sequence (tools/probe/sincos_f32_proto.c: 99.2 % of the inputs bit-equal to the oracle, the rest off by one) and then on the
// device (tests/tools/exhaustive_vs_oracle.py).  The zero signs are arranged so that sin(−0.0) = −0.0 (ph is an fma with +0.0,
// the low part is carried NEGATED).  |x| ≥ 1e6, inf and NaN take the f64 library path (Payne–Hanek inside), out of line.
__device__ __attribute__((noinline)) float sincos_f32_slow(float x, int want_cos) {
  return (float)(want_cos ? cos((double)x) : sin((double)x));
}
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2_t pk2(float v) { return (f32x2_t){v, v}; }
__device__ __forceinline__ f32x2_t pk_fma(f32x2_t a, f32x2_t b, f32x2_t c) { return __builtin_elementwise_fma(a, b, c); }
template <int WANT_COS>
__device__ __forceinline__ f32x2_t sincos_f32_pair(f32x2_t x) {  // meaningful for |x| < 1e6, harmless elsewhere
  const f32x2_t M = pk2(12582912.0f);                        // 1.5 · 2^23
  const f32x2_t s = pk_fma(x, pk2(0x1.45f306p-1f), M);       // x · 2/π + M: the sum's low mantissa bits are k
  const f32x2_t kf = s - M;
  const f32x2_t r1 = pk_fma(kf, pk2(-0x1.921fb6p+0f), x);    // exact
  const f32x2_t P2 = pk2(-0x1.777a5cp-25f);
  const f32x2_t ph = pk_fma(kf, P2, pk2(0.0f));              // an fma, not a product: +0.0 for k = 0 (−0.0 would turn r = −0.0 into +0.0)
  const f32x2_t pl = pk_fma(kf, P2, -ph);                    // exact residual
  const f32x2_t rh = r1 - ph;
  const f32x2_t t = rh - r1;
  const f32x2_t e = -ph - t;                                 // rounding error of rh
  f32x2_t nlo = pl - e;                                      // −lo
  nlo = pk_fma(kf, pk2(-0x1.ee59dap-50f), nlo);
  const f32x2_t z = rh * rh;
  f32x2_t ps = pk_fma(z, pk2(0x1.976586p-13f), pk2(-0x1.110122p-7f));
  ps = pk_fma(z, ps, pk2(0x1.555534p-3f));                   // −(sin r − r) / r³ > 0
  const f32x2_t u = pk_fma(rh * z, ps, nlo);
  const f32x2_t S = rh - u;
  f32x2_t pc = pk_fma(z, pk2(0x1.982456p-16f), pk2(-0x1.6c0536p-10f));
  pc = pk_fma(z, pc, pk2(0x1.55553ep-5f));
  pc = pk_fma(z, pc, pk2(-0.5f));
  const f32x2_t C = pk2(1.0f) + pk_fma(z, pc, rh * nlo);
  // (copies first: this clang lowers __builtin_bit_cast(uint32_t, s.y) — a bit cast of a vector ELEMENT — as a read of element 0)
  const float s0 = s.x, s1 = s.y;
  const uint32_t q0 = __builtin_bit_cast(uint32_t, s0) + (uint32_t)WANT_COS;  // cos(x) = sin(x + π/2)
  const uint32_t q1 = __builtin_bit_cast(uint32_t, s1) + (uint32_t)WANT_COS;
  // odd quadrant: C, else S — as a bit select under a 0 / ~0 mask (v_bfe_i32 + v_bfi_b32: two instructions; and / compare / cndmask are three);
  // quadrants 2, 3: negate — ADDING 2^31 flips the sign bit and nothing else (v_and + v_lshl_add_u32: two; shift / and / xor are three)
  const float Sx = S.x, Sy = S.y, Cx = C.x, Cy = C.y;
  // (the empty asm statements keep the compiler from canonicalising the two forms back into the three-instruction ones)
  uint32_t m0 = (uint32_t)__builtin_amdgcn_sbfe((int)q0, 0, 1), m1 = (uint32_t)__builtin_amdgcn_sbfe((int)q1, 0, 1);
  uint32_t n0 = q0 & 2u, n1 = q1 & 2u;
  asm("" : "+v"(m0), "+v"(m1), "+v"(n0), "+v"(n1));
  const uint32_t s0b = __builtin_bit_cast(uint32_t, Sx), s1b = __builtin_bit_cast(uint32_t, Sy);
  const uint32_t o0 = s0b ^ ((__builtin_bit_cast(uint32_t, Cx) ^ s0b) & m0);  // the form the back end turns into ONE v_bfi_b32
  const uint32_t o1 = s1b ^ ((__builtin_bit_cast(uint32_t, Cy) ^ s1b) & m1);
  const uint32_t v0 = (n0 << 30) + o0, v1 = (n1 << 30) + o1;
  return (f32x2_t){__builtin_bit_cast(float, v0), __builtin_bit_cast(float, v1)};
}
__device__ __forceinline__ bool sincos_f32_is_slow(float x) { return !(fabsf(x) < 1.0e6f); }
// N rows (N even): the pairs run back to back — independent chains the scheduler interleaves — and ONE branch covers the
// |x| ≥ 1e6 / inf / NaN rows
template <int WANT_COS, int N>
__device__ __forceinline__ void sincos_f32_rows(const float (&x)[N], float (&res)[N]) {
  static_assert(N % 2 == 0, "rows come in pairs");
  bool slow = false;
#pragma unroll
  for (int k = 0; k < N; k += 2) {
    const f32x2_t v = sincos_f32_pair<WANT_COS>((f32x2_t){x[k], x[k + 1]});
    res[k] = v.x;
    res[k + 1] = v.y;
    slow = slow || sincos_f32_is_slow(x[k]) || sincos_f32_is_slow(x[k + 1]);
  }
  if (slow) {
#pragma unroll
    for (int k = 0; k < N; k++)
      if (sincos_f32_is_slow(x[k])) res[k] = sincos_f32_slow(x[k], WANT_COS);
  }
}
// the fast path alone, one row (16-bit sources: |x| ≤ 65 535 never needs the slow one)
template <int WANT_COS>
__device__ __forceinline__ float sincos_f32_fast(float x) { return sincos_f32_pair<WANT_COS>(pk2(x)).x; }
template <int WANT_COS>
__device__ __forceinline__ float sincos_f32_dev(float x) {  // one row (tails, unaligned columns, table builders)
  float res = sincos_f32_fast<WANT_COS>(x);  // unconditionally: the rare slow path then merges ONE register, not the whole state
  if (sincos_f32_is_slow(x)) res = sincos_f32_slow(x, WANT_COS);
  return res;
}

// 128-entry table shared by log and pow: interval j of the mantissa [1 + j/128, 1 + (j+1)/128) → rc ≈ 1/centre (the
// mantissa is halved into [0.707, 1) from j = 53 on) and lc = −log2(rc); built once per device (pow_build_kernel), staged
// in LDS by the tile kernels, reachable from every other code path (tails, fused chains) through g_pow_tab.
struct alignas(16) PowTab {
  double rc, lc;
};
__device__ const PowTab* g_pow_tab = nullptr;

// log, general form (specials and denormals; also the form every operand took in round 1): x = m·2^e with
// m ∈ [√½, √2), s = (m−1)/(m+1), log m = 2s + s·z·P(z), z = s² (|s| ≤ 0.172), P = 2/3 + 2/5 z + … + 2/11 z⁴ evaluated in
// f32 (it weighs < 1 % of the result), the division as an f32 reciprocal + one f64 Newton step, the rest in f64, ONE
// rounding to f32.  (The f32 library logf measures 2 ULP on gfx950; the f64 library log is VALU-bound at 37 % of HBM —
// tests/tools/math_ulp.py, profiles/r01_kernel_table.json.)
__device__ __attribute__((noinline)) float log_f32_general(float x) {  // any operand: 0, negatives, NaN, inf, denormals
  if (!(x > 0.0f) || !(x < __builtin_inff())) {  // 0, negatives, NaN, +inf
    if (x == 0.0f) return -__builtin_inff();
    if (x < 0.0f || x != x) return __builtin_nanf("");
    return x;
  }
  int e;
  double m = frexp((double)x, &e);  // m ∈ [0.5, 1)
  if (m < 0x1.6a09e667f3bcdp-1) {   // < √½ : use [√½, √2)
    m *= 2.0;
    e -= 1;
  }
  const double f = m - 1.0, d = m + 1.0;
  const double r0 = (double)(1.0f / (float)d);
  const double r = r0 * fma(-d, r0, 2.0);  // 1/d to 2^-46
  const double s = f * r;
  const double z = s * s;
  const float zf = (float)z;
  float pf = __builtin_fmaf(zf, 0.18181818f, 0.22222222f);
  pf = __builtin_fmaf(zf, pf, 0.2857143f);
  pf = __builtin_fmaf(zf, pf, 0.4f);
  pf = __builtin_fmaf(zf, pf, 0.6666667f);
  const double t = fma(s * z, (double)pf, s + s);
  return (float)fma((double)e, 0x1.62e42fefa39efp-1, t);
}

// Positive normal finite x — what a wave of ordinary data consists of.  Rounds 2–5: the table form of pow's logarithm in f64 (ln x = (e + lc_j)·ln2 +
// log1p(m·rc_j − 1): 11 f64-class instructions + a 16-byte LDS read per row, its own 256-thread kernel to stage the table, 0.78–0.80 of the roof).
// Round 6: ALL in f32, two rows per instruction like sin / cos (sincos_f32_pair above):
//   x = 2^e · m, m ∈ [√½, √2) by integer arithmetic on the bits; f = m − 1 exact;
//   log m = f − f²/2 + f³·P(f), P = the degree-7 minimax polynomial of (log1p f − f + f²/2) / f³ on [√½ − 1, √2 − 1] (0.10 ULP of the result;
//   tools/probe/log_f32_proto.c prints the fit); ln x = fma(e, LN2_HI, fma(e, LN2_LO, log m)) with LN2_HI of 15 bits, so e · LN2_HI is exact.
// Every positive normal f32 is within 1 ULP of f64 libm rounded once — 12 408 324 of 2 122 317 824 inputs differ, all by one; largest error against
// the true value 0.874 ULP — checked on the CPU with this operation sequence (tools/probe/log_f32_proto.c, 15 s) and on the device
// (tests/tools/exhaustive_vs_oracle.py log).  7.5 packed + 5 scalar instructions per row, no table, no LDS: the generic f32 tile kernel runs it.
__device__ __forceinline__ bool log_ordinary(float x) {
  return __builtin_amdgcn_classf(x, 0x100);  // +normal: one v_cmp_class_f32
}
__device__ __forceinline__ f32x2_t log_f32_pair(f32x2_t x) {  // meaningful for positive normal x, harmless elsewhere
  const float x0 = x.x, x1 = x.y;  // (copies: this clang mis-lowers a bit cast of a vector ELEMENT — see sincos_f32_pair)
  const uint32_t b0 = __builtin_bit_cast(uint32_t, x0), b1 = __builtin_bit_cast(uint32_t, x1);
  const int32_t e0 = (int32_t)(b0 - 0x3f3504f3u) >> 23, e1 = (int32_t)(b1 - 0x3f3504f3u) >> 23;  // 0x3f3504f3 = √½: m ∈ [√½, √2)
  const float m0 = __builtin_bit_cast(float, b0 - ((uint32_t)e0 << 23)), m1 = __builtin_bit_cast(float, b1 - ((uint32_t)e1 << 23));
  const f32x2_t ef = {(float)e0, (float)e1};
  const f32x2_t f = (f32x2_t){m0, m1} - pk2(1.0f);
  const f32x2_t f2 = f * f;
  f32x2_t p = pk_fma(f, pk2(-0x1.38b586p-4f), pk2(0x1.055b6cp-3f));
  p = pk_fma(f, p, pk2(-0x1.0d8542p-3f));
  p = pk_fma(f, p, pk2(0x1.22da1cp-3f));
  p = pk_fma(f, p, pk2(-0x1.547244p-3f));
  p = pk_fma(f, p, pk2(0x1.99a008p-3f));
  p = pk_fma(f, p, pk2(-0x1.000226p-2f));
  p = pk_fma(f, p, pk2(0x1.555554p-2f));
  const f32x2_t q = pk_fma(f2 * f, p, -(f2 * pk2(0.5f)));  // f³·P − f²/2
  const f32x2_t r = f + q;
  const f32x2_t t = pk_fma(ef, pk2(0x1.7f7d1cp-20f), r);
  return pk_fma(ef, pk2(0x1.62e4p-1f), t);
}
// Anything that is not a positive normal.  0, negatives, NaN and +inf are three selects in line — a column with zeros or negatives in it
// (rows that come out −inf / NaN) used to send every wave through the out-of-line general form: 0.73 of the roof with half the rows negative;
// only positive DENORMALS (rare in any column) still need the general form's arithmetic.
__device__ __forceinline__ float log_f32_special(float x) {
  float r = x;  // +inf and NaN pass through
  r = (x == 0.0f) ? -__builtin_inff() : r;
  r = (x < 0.0f) ? __builtin_nanf("") : r;
  if (__builtin_amdgcn_classf(x, 0x080)) r = log_f32_general(x);  // +denormal
  return r;
}
template <int N>
__device__ __forceinline__ void log_f32_rows(const float (&x)[N], float (&res)[N]) {
  static_assert(N % 2 == 0, "rows come in pairs");
  bool slow = false;
#pragma unroll
  for (int k = 0; k < N; k += 2) {
    const f32x2_t v = log_f32_pair((f32x2_t){x[k], x[k + 1]});
    res[k] = v.x;
    res[k + 1] = v.y;
    slow = slow || !log_ordinary(x[k]) || !log_ordinary(x[k + 1]);
  }
  if (slow) {  // 0, negatives, NaN, inf, denormals
#pragma unroll
    for (int k = 0; k < N; k++)
      if (!log_ordinary(x[k])) res[k] = log_f32_special(x[k]);
  }
}
__device__ __forceinline__ float log_f32_dev(float x) {  // one row (tails, unaligned columns)
  float res = log_f32_pair(pk2(x)).x;
  if (!log_ordinary(x)) res = log_f32_special(x);
  return res;
}
// x = 2^e · m for a positive normal x, m ∈ [0.707, 1.414) as an f64 and j = the table interval of its mantissa.  Adding
// (128 − 53) << 16 carries into the exponent field exactly when j ≥ 53 (m ≥ 1.414 → halved), so e, the "big" bit and
// the exponent to strip from x all come out of ONE sum: 3 integer instructions + 2 conversions, where building the
// f64 mantissa word by word took a compare, a select (a VCC round trip) and 6 more.
struct MantSplit {
  double m;
  int e;
  uint32_t j;
};
__device__ __forceinline__ MantSplit split_normal_f32(float x) {
  const uint32_t xb = __builtin_bit_cast(uint32_t, x);
  const uint32_t s = xb + (0x004b0000u - 0x3f800000u);
  MantSplit r;
  r.e = (int32_t)s >> 23;
  r.m = (double)__builtin_bit_cast(float, xb - (s & 0xff800000u));
  r.j = (xb >> 16) & 127u;
  return r;
}
// log1p(u) − u + u²/2 = u³·q(u) on |u| ≤ 2^-7: q as the degree-2 minimax polynomial (error 2^-39.6 relative to u;
// tools/probe/poly_fit.py), evaluated in f32 — it weighs < 2^-15 of the sum
__device__ __forceinline__ float log1p_q(float uf) {
  float q = __builtin_fmaf(uf, 0x1.999f5p-3f, -0x1.0002p-2f);
  return __builtin_fmaf(uf, q, 0x1.555556p-2f);
}
// sinh: a = |x| = k·ln2 + r (|r| ≤ ln2/2); cosh r = C and sinh r = S from the even / odd Taylor halves (inner terms in
// f32, they weigh < 6 % of C and < 2 % of S; outer step in f64), then sinh a = 2^(k−1)(C+S) − 2^(−k−1)(C−S) in f64 and
// ONE rounding to f32.  k = 0 returns S itself (no cancellation for tiny |x|, keeps ±0 and denormals).  |x| is clamped
// to 128 (sinh overflows f32 from 89.42 on; the f64 result converts to ±inf); NaN is passed through at the end.
// The f32 library sinhf is ≤ 1 ULP too but VALU-bound at 2.8 TB/s (profiles/r01_kernel_table.json).
__device__ __forceinline__ float sinh_f32_dev(float x) {
  float a = fabsf(x);
  a = (a < 128.0f) ? a : 128.0f;
  const double ad = (double)a;
  const double sh = fma(ad, 0x1.71547652b82fep+0, 0x1.8p52);  // a · log2 e + 1.5 · 2^52: rounds to the integer k, whose
  const double kd = sh - 0x1.8p52;                            // value is also the low word of the sum (cf. sincos_f32_fast)
  const int k = (int)(uint32_t)__builtin_bit_cast(uint64_t, sh);
  const double r = fma(kd, -0x1.62e42fefa39efp-1, ad);
  const float rf = (float)r, zf = rf * rf;
  float qs = __builtin_fmaf(zf, 0x1.71de3a556c734p-19f, 0x1.a01a01a01a01ap-13f);  // 1/9!, 1/7!
  qs = __builtin_fmaf(zf, qs, 0x1.1111111111111p-7f);                             // 1/5!
  qs = __builtin_fmaf(zf, qs, 0x1.5555555555555p-3f);                             // 1/3!
  float qc = __builtin_fmaf(zf, 0x1.a01a01a01a01ap-16f, 0x1.6c16c16c16c17p-10f);  // 1/8!, 1/6!
  qc = __builtin_fmaf(zf, qc, 0x1.5555555555555p-5f);                             // 1/4!
  qc = __builtin_fmaf(zf, qc, 0.5f);
  const double z = r * r;
  const double S = fma(r * z, (double)qs, r);
  const double C = fma(z, (double)qc, 1.0);
  const double up = __builtin_bit_cast(double, (uint64_t)(uint32_t)(1022 + k) << 52);  // 2^(k−1)
  const double dn = __builtin_bit_cast(double, (uint64_t)(uint32_t)(1022 - k) << 52);  // 2^(−k−1)
  double v = fma(up, C + S, -(dn * (C - S)));
  v = (k == 0) ? S : v;
  const float f = __builtin_copysignf((float)v, x);
  return (x != x) ? x : f;
}

// The remaining functions use the f32 device library where it measures ≤ 1 ULP on gfx950 over 4 M log-uniform samples
// (tests/tools/math_ulp.py: acosf, cbrtf, exp2f, log2f, expf, powf = 1 ULP).
struct UnSqrt { __device__ static __forceinline__ float ap(float x, float) { return sqrtf(x); } };  // correctly rounded
struct UnCbrt { __device__ static __forceinline__ float ap(float x, float) { return cbrtf(x); } };
struct UnExp { __device__ static __forceinline__ float ap(float x, float) { return expf(x); } };
struct UnExp2 { __device__ static __forceinline__ float ap(float x, float) { return exp2f(x); } };
struct UnLog {
  __device__ static __forceinline__ float ap(float x, float) { return log_f32_dev(x); }
  template <int N> __device__ static __forceinline__ void ap_rows(const float (&x)[N], float (&r)[N]) { log_f32_rows<N>(x, r); }
};
struct UnLog2 { __device__ static __forceinline__ float ap(float x, float) { return log2f(x); } };
struct UnSin {
  __device__ static __forceinline__ float ap(float x, float) { return sincos_f32_dev<0>(x); }
  template <int N> __device__ static __forceinline__ void ap_rows(const float (&x)[N], float (&r)[N]) { sincos_f32_rows<0, N>(x, r); }
};
struct UnCos {
  __device__ static __forceinline__ float ap(float x, float) { return sincos_f32_dev<1>(x); }
  template <int N> __device__ static __forceinline__ void ap_rows(const float (&x)[N], float (&r)[N]) { sincos_f32_rows<1, N>(x, r); }
};
// functors with a several-rows-at-a-time form (sin / cos: packed f32 arithmetic over row pairs); everything else goes row by row
template <typename Op> struct EwRowsOp { static constexpr bool value = false; };
template <> struct EwRowsOp<UnSin> { static constexpr bool value = true; };
template <> struct EwRowsOp<UnCos> { static constexpr bool value = true; };
template <> struct EwRowsOp<UnLog> { static constexpr bool value = true; };
// … and converting functors (CvtThenF32 below) that have one: a static member `has_rows` = true and ap_rows<N>
template <typename Conv, typename = void> struct ConvHasRows { static constexpr bool value = false; };
template <typename Conv> struct ConvHasRows<Conv, std::enable_if_t<Conv::has_rows>> { static constexpr bool value = true; };
template <typename Op, int N>
__device__ __forceinline__ void unary_f32_rows(const float (&x)[N], float (&r)[N]) {
  if constexpr (EwRowsOp<Op>::value && N % 2 == 0) {
    Op::template ap_rows<N>(x, r);
  } else {
#pragma unroll
    for (int k = 0; k < N; k++) r[k] = Op::ap(x[k], 0.0f);
  }
}
struct UnAcos { __device__ static __forceinline__ float ap(float x, float) { return acosf(x); } };
struct UnSinh { __device__ static __forceinline__ float ap(float x, float) { return sinh_f32_dev(x); } };

// ---------------------------------------------------------------- same-width streaming kernel
// out[i] = Op(a[i], b[i] | *b | -).  The hot kernel covers FULL tiles only and carries no tail code (the tail costs
// registers and a branch in ~10^6 tiny blocks: −3 % measured); a second, tiny launch finishes the < 1-tile remainder.
// Block = ONE wave (64 lanes), one 16-byte pack per lane per array: 1 KiB per array per block — the best shape of the
// sweep in profiles/r01_sweep_add_eq_1e9_b.json (6.7 TB/s vs 6.4 at 256 threads, ≤5.9 persistent).
#define AGPU_EW_BLOCK 64

template <typename T, typename Op, int MODE, int U, int NT, int BLK = AGPU_EW_BLOCK>
__global__ __launch_bounds__(BLK) void ew_kernel(const T* a, const T* b, T* out, uint64_t ntiles, uint64_t half) {
  constexpr int N = 16 / sizeof(T);
  constexpr bool NTL = (NT & 1) != 0, NTS = (NT & 2) != 0, XCD = (NT & 4) != 0;
  constexpr uint64_t tile = (uint64_t)BLK * U;
  T sv = T();
  if constexpr (MODE == MODE_SCALAR) sv = b[0];

  for (uint64_t t0 = blockIdx.x; t0 < ntiles; t0 += gridDim.x) {
    uint64_t t = t0;
    if constexpr (XCD) {  // blocks are dealt round-robin to the 8 XCDs: give XCD x the x-th contiguous eighth of the tiles
      const uint64_t per = ntiles / 8;
      if (t0 < per * 8) t = (t0 & 7) * per + (t0 >> 3);
    } else if (t0 < 2 * half) {  // two lock-step streams half a column apart (launch_ew: one-input kernels over big columns)
      t = (t0 >> 1) + ((t0 & 1) ? half : 0);
    }
    const uint64_t p0 = t * tile + threadIdx.x;
    PackN<T, N> va[U], vb[U];
    static_for<U>([&](auto u) {
      va[u] = load_pack<NTL, T, N>(a + (p0 + (uint64_t)u * BLK) * N);
      if constexpr (MODE == MODE_BINARY) vb[u] = load_pack<NTL, T, N>(b + (p0 + (uint64_t)u * BLK) * N);
    });
    static_for<U>([&](auto u) {
      PackN<T, N> r;
      if constexpr (EwRowsOp<Op>::value && MODE == MODE_UNARY && std::is_same<T, float>::value) {
        unary_f32_rows<Op, N>(va[u].v, r.v);
      } else {
#pragma unroll
        for (int k = 0; k < N; k++) r.v[k] = Op::ap(va[u].v[k], MODE == MODE_BINARY ? vb[u].v[k] : sv);
      }
      store_pack<NTS, T, N, (MODE != MODE_BINARY && sizeof(T) == 4)>(out + (p0 + (uint64_t)u * BLK) * N, r);  // sc1: common.hpp st_vec_sc1
    });
  }
}

// The VALU-heavy f32 unary functors (sin / cos / sinh / log): a wave that evaluates 8–16 rows of f64 arithmetic has NOTHING in
// flight while it does — with one tile per block the memory system sees each wave's bytes only between its launch and its
// first instruction of arithmetic (sin / cos / sinh all sat at 0.775–0.78 of the roof whatever their instruction count: 36 → 30
// VALU instructions per row changed nothing, profiles/r04_pmc_narrow.json: VALU 59 % busy).  Here a block walks `tiles per
// block` tiles a grid apart and issues the NEXT tile's loads before it evaluates the current one, so the loads ride under the
// arithmetic.  One tile per block degenerates to ew_kernel's shape.
template <typename Op, int U, int NT>
__global__ __launch_bounds__(AGPU_EW_BLOCK) void ew_prefetch_kernel(const float* a, float* out, uint64_t ntiles) {
  constexpr int N = 4;
  constexpr bool NTL = (NT & 1) != 0, NTS = (NT & 2) != 0;
  constexpr uint64_t tile = (uint64_t)AGPU_EW_BLOCK * U;
  const TileRun run = tile_run(blockIdx.x, gridDim.x, ntiles);
  uint64_t t = run.t;
  if (t >= run.end) return;
  PackN<float, N> cur[U];
  static_for<U>([&](auto u) { cur[u] = load_pack<NTL, float, N>(a + (t * tile + threadIdx.x + (uint64_t)u * AGPU_EW_BLOCK) * N); });
  for (;;) {
    const uint64_t tn = t + run.step;
    const bool more = tn < run.end;  // block-uniform
    PackN<float, N> nxt[U];
    if (more)
      static_for<U>([&](auto u) { nxt[u] = load_pack<NTL, float, N>(a + (tn * tile + threadIdx.x + (uint64_t)u * AGPU_EW_BLOCK) * N); });
    static_for<U>([&](auto u) {
      PackN<float, N> r;
      unary_f32_rows<Op, N>(cur[u].v, r.v);
      store_pack<NTS, float, N, true>(out + (t * tile + threadIdx.x + (uint64_t)u * AGPU_EW_BLOCK) * N, r);
    });
    if (!more) break;
    static_for<U>([&](auto u) { cur[u] = nxt[u]; });
    t = tn;
  }
}
template <typename Op> struct EwPrefetch { static constexpr bool value = false; static constexpr int tiles = 1; };
#ifndef AGPU_CHAIN_HEAVY_LDS
#define AGPU_CHAIN_HEAVY_LDS 1u  // chains with a transcendental step: no cap by default (1 byte), tuning wave_lds forces one (tools/probe/chain_caps.py)
#endif
template <typename Op> struct EwWaveLds { static constexpr unsigned value = 0; };  // occupancy cap (common.hpp wave_lds_for): bytes per wave

// rows [first, n): whole packs while they last, then single elements.  One small block; also serves tiny arrays.
template <typename T, typename Op, int MODE>
__global__ __launch_bounds__(AGPU_BLOCK) void ew_tail_kernel(const T* a, const T* b, T* out, uint64_t first, uint64_t n) {
  constexpr int N = 16 / sizeof(T);
  T sv = T();
  if constexpr (MODE == MODE_SCALAR) sv = b[0];
  const uint64_t npacks = n / N;
  for (uint64_t pk = first / N + threadIdx.x; pk < npacks; pk += AGPU_BLOCK) {
    PackN<T, N> x = load_pack<false, T, N>(a + pk * N), y, r;
    if constexpr (MODE == MODE_BINARY) y = load_pack<false, T, N>(b + pk * N);
#pragma unroll
    for (int k = 0; k < N; k++) r.v[k] = Op::ap(x.v[k], MODE == MODE_BINARY ? y.v[k] : sv);
    store_pack<false, T, N>(out + pk * N, r);
  }
  const uint64_t i = npacks * N + threadIdx.x;
  if (i >= first && i < n) out[i] = Op::ap(a[i], MODE == MODE_BINARY ? b[i] : sv);
}

// element-granular fallback for pointers that are not 16-byte aligned (e.g. odd shard offsets)
template <typename T, typename Op, int MODE>
__global__ __launch_bounds__(AGPU_BLOCK) void ew_kernel_unaligned(const T* a, const T* b, T* out, uint64_t n) {
  T sv = T();
  if constexpr (MODE == MODE_SCALAR) sv = b[0];
  for (uint64_t i = (uint64_t)blockIdx.x * AGPU_BLOCK + threadIdx.x; i < n; i += (uint64_t)gridDim.x * AGPU_BLOCK)
    out[i] = Op::ap(a[i], MODE == MODE_BINARY ? b[i] : sv);
}

template <typename Op> struct EwUnroll { static constexpr int value = AGPU_STREAM_U; };
#ifndef AGPU_SINCOS_U
#define AGPU_SINCOS_U 2  // 16-byte packs per lane of the f32 sin / cos tile kernel (A/B builds: tools/r06_sincos_variants.sh)
#endif
template <> struct EwUnroll<UnSin> { static constexpr int value = AGPU_SINCOS_U; };  // re-checked after the round-2 trimming: 1 → −5 %, 3 → −5 %; and under the
template <> struct EwUnroll<UnCos> { static constexpr int value = AGPU_SINCOS_U; };  // occupancy cap (round 5, profiles/r05_sincos_unroll_x_cap.txt): 2 @ 6800 B 0.83, 4 @ 10240 B 0.82, 1 @ 4200 B 0.80–0.81, 3 0.78
template <> struct EwUnroll<UnLog> { static constexpr int value = 2; };
template <> struct EwUnroll<UnSinh> { static constexpr int value = 4; };  // re-checked under the occupancy cap (round 5): 2 → −2 %, 1 → −10 %
template <> struct EwWaveLds<UnSinh> { static constexpr unsigned value = AGPU_WAVE_LDS_24; };  // 0.70–0.75 → 0.77 on a column in (−30, 30) (the exp path); flat where most rows overflow
// tiles per block: 1 by default (see tile_run above: 3–4 tiles gain 4 % in lucky allocations — sin 0.785 → 0.82 — and lose 7 % in others)
// round 6, the packed-f32 form (tools/probe/r06_sincos_sweep.py, two processes, tiles × cap): ≈ 16 waves per CU 0.84–0.85 of the roof, ≈ 24 0.81–0.84, none 0.76–0.79
template <> struct EwWaveLds<UnSin> { static constexpr unsigned value = AGPU_WAVE_LDS_16; };
template <> struct EwWaveLds<UnCos> { static constexpr unsigned value = AGPU_WAVE_LDS_16; };
template <> struct EwPrefetch<UnSin> { static constexpr bool value = true; static constexpr int tiles = 1; };
template <> struct EwPrefetch<UnCos> { static constexpr bool value = true; static constexpr int tiles = 1; };
template <> struct EwPrefetch<UnSinh> { static constexpr bool value = true; static constexpr int tiles = 1; };
// log (round 6: packed f32, no table — the generic tile kernel instead of a kernel of its own).  One tile per block like the others: two were level
// or 1 % ahead in one process and 5 % behind in the next (0.84 / 0.79 — the by-process effect of R4.1 / R5.4), one tile is 0.83 everywhere
template <> struct EwPrefetch<UnLog> { static constexpr bool value = true; static constexpr int tiles = 1; };
template <> struct EwWaveLds<UnLog> { static constexpr unsigned value = AGPU_WAVE_LDS_24; };  // (tools/probe/r06_sincos_sweep.py: ≈ 24 waves 0.83–0.84, ≈ 16 0.77–0.78, no cap 0.79–0.83)
// Threads per block: one wave for everything.  (The LDS-table kernels pow / log run best at 256 threads × 1 pack, but
// sin / cos / sinh lose 4–10 % in that shape and 5 % at 64 × 1: tools/probe/heavy_shape.py, profiles/r02_heavy_shape.txt.)
#ifndef AGPU_EW_DEFAULT_BLK
#define AGPU_EW_DEFAULT_BLK AGPU_EW_BLOCK
#endif
template <typename Op> struct EwBlock { static constexpr int value = AGPU_EW_DEFAULT_BLK; };

template <typename T, typename Op, int MODE>
static agpu_status launch_ew(agpu_pipeline* p, const void* a, const void* b, void* out, uint64_t n) {
  if (n == 0) return AGPU_OK;
  const T* pa = static_cast<const T*>(a);
  const T* pb = static_cast<const T*>(b);
  T* po = static_cast<T*>(out);
  const bool vec_ok = aligned16(a) && aligned16(out) && (MODE != MODE_BINARY || aligned16(b));
  if (vec_ok) {
    constexpr int N = 16 / sizeof(T);
    // 16-byte packs per lane per block: 1 for everything that is purely HBM-bound; the VALU-heavy functors hide their
    // dependent f64 chains better with 2–4 independent rows of work per lane (A/B on one box at 1e9 rows: sin/cos/log
    // 5.9 → 6.25 TB/s at U = 2, sinh 5.95 → 6.44 at U = 4, while add_scalar/neg/exp LOSE 10 % at U = 2)
    constexpr int U = EwUnroll<Op>::value;
    constexpr int BLK = EwBlock<Op>::value;
    constexpr uint64_t tile_rows = (uint64_t)BLK * U * N;
    const uint64_t ntiles = n / tile_rows;
    if (ntiles) {
      const int grid = stream_grid_for(p, ntiles);
      // Columns that do not start on a 128-byte line (slices): every 1 KiB wave chunk straddles a ninth line that the
      // neighbouring tile fetches again — from another XCD's L2 with the round-robin mapping (−15 %).  Those launches
      // give each XCD a contiguous eighth of the tiles and use cacheable accesses, so the shared line is fetched once:
      // 5.60 → 6.16 TB/s on the worst case of tools/probe/misaligned_probe.py; line-aligned columns keep the
      // round-robin + nontemporal shape (the XCD mapping costs them 1.4 %).
      const uintptr_t bits = reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(out) |
                             (MODE == MODE_BINARY ? reinterpret_cast<uintptr_t>(b) : 0);
      // one-input kernels (unary, scalar) over big columns: two lock-step streams (common.hpp two_streams); two-input kernels lose 0.5 % by it
      // (sinh, the one f64-heavy functor left, LOSES 3.6 % by it: 0.755 → 0.728, tools/probe/two_streams_ab3.py)
      const uint64_t half = (MODE != MODE_BINARY && !std::is_same<Op, UnSinh>::value) ? two_streams_half(p, ntiles, 16 * (uint64_t)BLK * U) : 0;
      bool done = false;
      if constexpr (EwPrefetch<Op>::value && MODE == MODE_UNARY && std::is_same<T, float>::value && BLK == AGPU_EW_BLOCK) {
        const int64_t k = p->tune.tiles > 0 ? p->tune.tiles : EwPrefetch<Op>::tiles;  // static defaults since round 6 (the adaptive policy of round 5 decided "one" everywhere)
        const bool shape_ok = (bits & 127u) == 0 && p->tune.stream_grid == 0;
        if (k > 1 && shape_ok) {
          const int g = stream_grid_for(p, tile_units(ntiles, (uint64_t)k));
          hipLaunchKernelGGL((ew_prefetch_kernel<Op, U, AGPU_STREAM_NT>), dim3(g), dim3(AGPU_EW_BLOCK), wave_lds_for(p, EwWaveLds<Op>::value, 1), p->stream, pa, po, ntiles);
          done = true;
        }
      }
      if (done) {
      } else if ((bits & 127u) == 0)
        hipLaunchKernelGGL((ew_kernel<T, Op, MODE, U, AGPU_STREAM_NT, BLK>), dim3(grid), dim3(BLK),
                           EwWaveLds<Op>::value ? wave_lds_for(p, EwWaveLds<Op>::value, BLK / AGPU_WAVE) : 0u, p->stream, pa, pb, po, ntiles, half);
      else
        hipLaunchKernelGGL((ew_kernel<T, Op, MODE, U, 4, BLK>), dim3(grid), dim3(BLK), 0, p->stream, pa, pb, po, ntiles, (uint64_t)0);
    }
    if (ntiles * tile_rows < n)
      hipLaunchKernelGGL((ew_tail_kernel<T, Op, MODE>), dim3(1), dim3(AGPU_BLOCK), 0, p->stream, pa, pb, po,
                         ntiles * tile_rows, n);
  } else {
    const int grid = stream_grid_for(p, (n + AGPU_BLOCK - 1) / AGPU_BLOCK);
    hipLaunchKernelGGL((ew_kernel_unaligned<T, Op, MODE>), dim3(grid), dim3(AGPU_BLOCK), 0, p->stream, pa, pb, po, n);
  }
  AGPU_LAUNCH_CHECK();
  return AGPU_OK;
}

// ---------------------------------------------------------------- shifts: lhs T[n], rhs u32[n] (or 1 scalar)
// value extended to 32 bits, amount mod 32, result truncated to T
// [logical/compute_shaders/{u32,i32,u16,i16,u8,i8}/shift.wgsl]
template <typename T, bool LEFT>
__device__ __forceinline__ T shift_one(T x, uint32_t sh) {
  if constexpr (LEFT) return (T)((uint32_t)(int32_t)x << sh);
  else if constexpr (std::is_signed<T>::value) return (T)((int32_t)x >> sh);
  else return (T)((uint32_t)x >> sh);
}
// sub-word columns: 4 rows per lane per step — one 16-byte load of amounts, one 4/8-byte load and store of values.
// One-wave blocks (round 5: the streaming kernels' shape; 256-thread blocks measured 0.75 / 0.78 of the roof for u8 / u16)
template <typename T, bool LEFT, int MODE, int BLK>
__global__ __launch_bounds__(BLK) void shift_kernel(const T* a, const uint32_t* s, T* out, uint64_t n, int vec_ok) {
  uint32_t sv = 0;
  if constexpr (MODE == MODE_SCALAR) sv = s[0] & 31u;
  const uint64_t tid = (uint64_t)blockIdx.x * BLK + threadIdx.x;
  const uint64_t stride = (uint64_t)gridDim.x * BLK;
  const uint64_t npacks = vec_ok ? n / 4 : 0;
  for (uint64_t pk = tid; pk < npacks; pk += stride) {
    const PackN<T, 4> x = load_pack<true, T, 4>(a + pk * 4);
    u32x4 sh = {sv, sv, sv, sv};
    if constexpr (MODE != MODE_SCALAR) sh = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(s) + pk) & 31u;
    PackN<T, 4> r;
    r.v[0] = shift_one<T, LEFT>(x.v[0], sh.x);
    r.v[1] = shift_one<T, LEFT>(x.v[1], sh.y);
    r.v[2] = shift_one<T, LEFT>(x.v[2], sh.z);
    r.v[3] = shift_one<T, LEFT>(x.v[3], sh.w);
    store_pack<true, T, 4>(out + pk * 4, r);
  }
  for (uint64_t i = npacks * 4 + tid; i < n; i += stride)
    out[i] = shift_one<T, LEFT>(a[i], MODE == MODE_SCALAR ? sv : (s[i] & 31u));
}
template <typename T, int MODE>
static agpu_status launch_shift(agpu_pipeline* p, bool left, const void* a, const void* s, void* out, uint64_t n) {
  if (n == 0) return AGPU_OK;
  const int vec_ok = aligned_to(a, 4 * sizeof(T)) && aligned_to(out, 4 * sizeof(T)) && (MODE == MODE_SCALAR || aligned16(s));
  const uint64_t blk = AGPU_WAVE;
  const int grid = stream_grid_for(p, (n / 4 + blk) / blk);
#define AGPU_SHIFT_LAUNCH(L, B)                                                                                       \
  hipLaunchKernelGGL((shift_kernel<T, L, MODE, B>), dim3(grid), dim3(B), 0, p->stream, static_cast<const T*>(a), \
                     static_cast<const uint32_t*>(s), static_cast<T*>(out), n, vec_ok)
  if (left) AGPU_SHIFT_LAUNCH(true, AGPU_WAVE);
  else AGPU_SHIFT_LAUNCH(false, AGPU_WAVE);
#undef AGPU_SHIFT_LAUNCH
  AGPU_LAUNCH_CHECK();
  return AGPU_OK;
}

// ---------------------------------------------------------------- f32 pow (x > 0): 2^(y · log2 x), ≤ 1 ULP
// The result's relative error is ln2 · |error of y·log2 x| and |y·log2 x| reaches 150, so log2 x needs ~2⁻³³ relative
// accuracy: f32 arithmetic cannot carry it, the f64 library pow runs at a quarter of the stream and the f32 library
// powf (compensated f32) at 40 %.  Here: x = 2^e · m with m ∈ [0.707, 1.414); a 128-entry f64 table (LDS) gives
// rc ≈ 1/c and lc = −log2 rc for the centre c of m's interval, so u = m·rc − 1 is EXACT-ish (one fma) with |u| ≤ 2⁻⁷ and
//   log2 x = (e + lc) + log2e · (u − u²/2 + u³·q(u)),   q ≈ 1/3 − u/4 + u²/5 (minimax) in f32 (weighs < 2⁻¹⁵ of the sum).
// The intervals touching 1 use rc = 1, lc = 0 exactly, so log2 x keeps its RELATIVE accuracy as x → 1 (where y may be
// 1e9).  Then w = y·log2 x = k + r, 2^r = 1 + t + t²·Q(t) with t = r·ln2 and Q (degree-5 minimax of (e^t − 1 − t)/t²) in f32 (weighs < 7 %),
// v_ldexp_f64 by k (handles overflow / underflow / denormal results) and ONE rounding to f32.
// Validated against f64 pow over 6 × 4 M samples (generic, full exponent range, x → 1 with huge y, denormal x,
// |y| ≤ 60) in tools/probe/pow_emul.py and on the device in tests/test_gpu_parity.py.
__global__ void pow_build_kernel(PowTab* tab) {
  const int j = threadIdx.x;  // interval j of the f64 mantissa: [1 + j/128, 1 + (j+1)/128)
  if (j >= 128) return;
  const double c = 1.0 + (j + 0.5) / 128.0;
  double rc = 1.0 / (j >= 53 ? 0.5 * c : c);  // j ≥ 53 (m ≥ 1.414): the mantissa is halved into [0.707, 1)
  if (j == 0 || j == 127) rc = 1.0;
  tab[j].rc = rc;
  tab[j].lc = -log2(rc);
}

// SCALAR exponents 1 and 2 — the ones callers expect to be exact — are answered exactly (x, RN(x·x)) on a wave-uniform
// branch; inside the general path the two extra selects cost 7 % of a VALU-bound kernel (6.05 → 5.65 TB/s measured).
__device__ __forceinline__ float pow_small_exponent(float x, float y) {  // y ∈ {1, 2}
  if (x != x || x < 0.0f || (x == 0.0f && __builtin_signbit(x))) return __builtin_nanf("");  // the reference's domain
  return y == 1.0f ? x : x * x;
}

// (e^t − 1 − t)/t² on |t| ≤ ln2/2 as the degree-5 minimax polynomial (error 2^-32.5 of e^t; tools/probe/poly_fit.py), in f32:
// t²·Q weighs < 7 % of the result
__device__ __forceinline__ float exp_q(float tf) {
  float Q = __builtin_fmaf(tf, 0x1.a17e0cp-13f, 0x1.6d4328p-10f);
  Q = __builtin_fmaf(tf, Q, 0x1.1110acp-7f);
  Q = __builtin_fmaf(tf, Q, 0x1.5554eap-5f);
  Q = __builtin_fmaf(tf, Q, 0x1.555556p-3f);
  return __builtin_fmaf(tf, Q, 0.5f);
}
template <typename TabPtr>
__device__ __forceinline__ float pow_f32_general(TabPtr tab, float x, float y) {  // any operands: specials, denormal x, huge |y|
  const uint64_t bits = __builtin_bit_cast(uint64_t, (double)x);
  const uint32_t hi = (uint32_t)(bits >> 32);
  const uint32_t j = (hi >> 13) & 127u;
  const bool big = j >= 53u;
  const int e = (int)((hi >> 20) & 0x7ffu) - 1023 + (big ? 1 : 0);
  const uint32_t mhi = (hi & 0x000fffffu) | (big ? 0x3fe00000u : 0x3ff00000u);
  const double m = __builtin_bit_cast(double, ((uint64_t)mhi << 32) | (uint64_t)(uint32_t)bits);
  const PowTab T = tab[j];
  const double u = fma(m, T.rc, -1.0);
  const double u2 = u * u;
  const double l1p = fma(u2, fma(u, (double)log1p_q((float)u), -0.5), u);  // u − u²/2 + u³·q, one f64 multiply less than the expanded form
  const double L = fma(l1p, 0x1.71547652b82fep+0, (double)e + T.lc);
  double w = (double)y * L;
  w = w < 2000.0 ? w : 2000.0;  // keeps k inside v_ldexp's range; NaN cannot occur here (specials handled below)
  w = w > -2000.0 ? w : -2000.0;
  const double kd = rint(w);
  const double t = (w - kd) * 0x1.62e42fefa39efp-1;
  const double pw = fma(t, fma(t, (double)exp_q((float)t), 1.0), 1.0);  // 1 + t + t²·Q as two fmas
  float r = (float)ldexp(pw, (int)kd);
  // Specials, IEEE pow restricted to the reference's domain [math/src/f32.rs:209-271]: negative or NaN base → NaN.
  // The main path is already right for y == 0 and x == 1 with finite operands (w = 0 ⇒ 1) and for denormal x; the
  // fix-ups sit behind a branch that a wave of ordinary positive finite operands skips — as selects they were 12 of
  // the 62 VALU instructions per row of a kernel that is VALU-bound whenever the shader clock is not at its maximum.
  const float inf = __builtin_inff();
  const bool ordinary = x > 0.0f && x < inf && __builtin_fabsf(y) < inf;
  if (!ordinary) {
    if (!(x > 0.0f && x < inf)) r = ((x != 0.0f) == (y > 0.0f)) ? inf : 0.0f;  // 0^y, inf^y
    if (y == 0.0f || x == 1.0f) r = 1.0f;
    if (x != x || y != y || x < 0.0f || (x == 0.0f && __builtin_signbit(x))) r = __builtin_nanf("");
  }
  return r;
}

// The hot path: x positive, normal and finite, |y| < 2^20 — everything a wave of ordinary data consists of.  Compared
// with the general form it builds the f64 mantissa straight from the f32 bits (no v_cvt_f64_f32), needs no clamp on
// y·log2(x) (|w| < 2^28: v_ldexp_f64 saturates to 0 / inf by itself) and no special-case selects; the general form sits
// behind a branch such waves never take.  Same table, same polynomials, same roundings: identical bits.
__device__ __forceinline__ bool pow_ordinary(float x, float y) {  // x positive normal finite, |y| < 2^20
  return __builtin_amdgcn_classf(x, 0x100) && __builtin_fabsf(y) < 0x1p20f;  // v_cmp_class_f32 + v_cmp_lt_f32 |y| (false for NaN)
}
template <typename TabPtr>
__device__ __forceinline__ float pow_f32_fast(TabPtr tab, float x, float y) {  // requires pow_ordinary(x, y)
  const MantSplit sp = split_normal_f32(x);
  const PowTab T = tab[sp.j];
  const double u = fma(sp.m, T.rc, -1.0);
  const double l1p = fma(u * u, fma(u, (double)log1p_q((float)u), -0.5), u);
  const double L = fma(l1p, 0x1.71547652b82fep+0, (double)sp.e + T.lc);
  const double w = (double)y * L;
  const double kd = rint(w);
  const double t = (w - kd) * 0x1.62e42fefa39efp-1;
  const double pw = fma(t, fma(t, (double)exp_q((float)t), 1.0), 1.0);
  return (float)ldexp(pw, (int)kd);
}
// per-element dispatch (tail rows, fused chains); the tile kernel below votes once per wave instead
template <typename TabPtr>
__device__ __forceinline__ float pow_f32_dev(TabPtr tab, float x, float y) {
  return pow_ordinary(x, y) ? pow_f32_fast(tab, x, y) : pow_f32_general(tab, x, y);
}

#ifndef AGPU_POW_VOTE
#define AGPU_POW_VOTE 1
#endif
// Tile shape of the two LDS-table kernels (pow, log), swept on one box at 1e9 rows (tools/probe/pow_shape.py,
// profiles/r02_pow_shape.txt): ONE pack per lane and ONE tile per block — pow 0.68 → 0.81 of the HBM roof, log 0.70 → 0.78.
// The arithmetic was never the bound (the same kernel with the stores disabled and the loads skipped: 1.05 ms; with
// the arithmetic replaced by an add: the full 2.15 ms): a block that loops over tiles, or carries 8 rows per lane,
// spends its waves in long compute phases with nothing in flight, and the one-tile form hands that to the dispatcher.
#define AGPU_POW_U 1
template <int MODE>
__global__ __launch_bounds__(AGPU_BLOCK) void pow_kernel(const float* a, const float* b, float* out, uint64_t ntiles,
                                                        const PowTab* gtab) {
  constexpr int U = AGPU_POW_U;
  constexpr uint64_t TILE_PACKS = (uint64_t)AGPU_BLOCK * U;
  __shared__ PowTab tab[128];
  const f32x4* a4 = reinterpret_cast<const f32x4*>(a);
  const f32x4* b4 = reinterpret_cast<const f32x4*>(b);
  f32x4* o4 = reinterpret_cast<f32x4*>(out);
  const TileRun run = tile_run(blockIdx.x, gridDim.x, ntiles);
  const uint64_t ntiles_end = run.end;
  uint64_t t = run.t;
  f32x4 xa[U], xb[U];
  float sv = 0.0f;
  if constexpr (MODE == MODE_SCALAR) sv = b[0];
  auto load_tile = [&](uint64_t tile) {
    static_for<U>([&](auto u) {
      const uint64_t pk = tile * TILE_PACKS + threadIdx.x + (uint64_t)u * AGPU_BLOCK;
      xa[u] = __builtin_nontemporal_load(a4 + pk);
      if constexpr (MODE == MODE_BINARY) xb[u] = __builtin_nontemporal_load(b4 + pk);
    });
  };
  if (t < ntiles_end) load_tile(t);
  if (threadIdx.x < 128)
    reinterpret_cast<u32x4*>(tab)[threadIdx.x] = reinterpret_cast<const u32x4*>(gtab)[threadIdx.x];
  __syncthreads();
  while (t < ntiles_end) {
    const uint64_t p0 = t * TILE_PACKS + threadIdx.x;
    // the next tile's loads go out BEFORE this tile is evaluated (tiles > 1): they ride under the arithmetic
    const uint64_t tn = t + run.step;
    f32x4 na[U], nb[U];
    if (tn < ntiles_end)
      static_for<U>([&](auto u) {
        const uint64_t pk = tn * TILE_PACKS + threadIdx.x + (uint64_t)u * AGPU_BLOCK;
        na[u] = __builtin_nontemporal_load(a4 + pk);
        if constexpr (MODE == MODE_BINARY) nb[u] = __builtin_nontemporal_load(b4 + pk);
      });
    // one vote per wave and tile: a wave whose 512 operand pairs are all ordinary (x positive normal, |y| < 2^20) takes
    // the short path on a SCALAR branch — no exec masking, no selects; any other wave evaluates the general form
    bool ok = true;
    static_for<U>([&](auto u) {
      if constexpr (MODE == MODE_BINARY)
        ok = ok && pow_ordinary(xa[u].x, xb[u].x) && pow_ordinary(xa[u].y, xb[u].y) && pow_ordinary(xa[u].z, xb[u].z) &&
             pow_ordinary(xa[u].w, xb[u].w);
      else
        ok = ok && pow_ordinary(xa[u].x, sv) && pow_ordinary(xa[u].y, sv) && pow_ordinary(xa[u].z, sv) && pow_ordinary(xa[u].w, sv);
    });
    const bool small_exp = MODE == MODE_SCALAR && (sv == 1.0f || sv == 2.0f);
    if (small_exp) {
      static_for<U>([&](auto u) {
        const f32x4 r = f32x4{pow_small_exponent(xa[u].x, sv), pow_small_exponent(xa[u].y, sv), pow_small_exponent(xa[u].z, sv),
                              pow_small_exponent(xa[u].w, sv)};
        __builtin_nontemporal_store(r, o4 + p0 + (uint64_t)u * AGPU_BLOCK);
      });
    } else if (AGPU_POW_VOTE && __all(ok)) {
      static_for<U>([&](auto u) {
        f32x4 yb4;
        if constexpr (MODE == MODE_BINARY) yb4 = xb[u];
        else yb4 = f32x4{sv, sv, sv, sv};
        const f32x4 r = f32x4{pow_f32_fast(tab, xa[u].x, yb4.x), pow_f32_fast(tab, xa[u].y, yb4.y), pow_f32_fast(tab, xa[u].z, yb4.z),
                              pow_f32_fast(tab, xa[u].w, yb4.w)};
        __builtin_nontemporal_store(r, o4 + p0 + (uint64_t)u * AGPU_BLOCK);
      });
    } else {
      static_for<U>([&](auto u) {
        f32x4 yb4;
        if constexpr (MODE == MODE_BINARY) yb4 = xb[u];
        else yb4 = f32x4{sv, sv, sv, sv};
        const f32x4 r = f32x4{pow_f32_general(tab, xa[u].x, yb4.x), pow_f32_general(tab, xa[u].y, yb4.y),
                              pow_f32_general(tab, xa[u].z, yb4.z), pow_f32_general(tab, xa[u].w, yb4.w)};
        __builtin_nontemporal_store(r, o4 + p0 + (uint64_t)u * AGPU_BLOCK);
      });
    }
    if (tn < ntiles_end)
      static_for<U>([&](auto u) {
        xa[u] = na[u];
        if constexpr (MODE == MODE_BINARY) xb[u] = nb[u];
      });
    t = tn;
  }
}
template <int MODE>
__global__ __launch_bounds__(AGPU_BLOCK) void pow_tail_kernel(const float* a, const float* b, float* out, uint64_t first,
                                                             uint64_t n, const PowTab* gtab) {
  for (uint64_t i = first + (uint64_t)blockIdx.x * AGPU_BLOCK + threadIdx.x; i < n; i += (uint64_t)gridDim.x * AGPU_BLOCK)
    if (MODE == MODE_SCALAR && (b[0] == 1.0f || b[0] == 2.0f)) out[i] = pow_small_exponent(a[i], b[0]);
    else out[i] = pow_f32_dev(gtab, a[i], MODE == MODE_BINARY ? b[i] : b[0]);
}
template <int MODE>
static agpu_status launch_pow_f32(agpu_pipeline* p, const void* a, const void* b, void* out, uint64_t n) {
  if (n == 0) return AGPU_OK;
  const float* pa = static_cast<const float*>(a);
  const float* pb = static_cast<const float*>(b);
  float* po = static_cast<float*>(out);
  const PowTab* tab = static_cast<const PowTab*>(p->dev->pow_table);
  constexpr uint64_t TILE_ROWS = (uint64_t)AGPU_BLOCK * AGPU_POW_U * 4;
  uint64_t done = 0;
  if (aligned16(a) && aligned16(out) && (MODE != MODE_BINARY || aligned16(b))) {
    const uint64_t ntiles = n / TILE_ROWS;
    if (ntiles) {
      // tiles per block with the next tile prefetched (profiles/r04_prefetch_sweep*.json): array ∘ array is best at 1 (0.807 → 0.80 at 2); array ∘ scalar
      // 0.67–0.69 → 0.75–0.77 at 3 in every run
      const uint64_t tk = tab_k(p, MODE == MODE_SCALAR ? 3 : 1);
      hipLaunchKernelGGL((pow_kernel<MODE>), dim3(stream_grid_for(p, tile_units(ntiles, tk))), dim3(AGPU_BLOCK), 0, p->stream, pa, pb, po,
                         ntiles, tab);
      done = ntiles * TILE_ROWS;
    }
  }
  if (done < n) {
    const int grid = stream_grid_for(p, (n - done + AGPU_BLOCK - 1) / AGPU_BLOCK);
    hipLaunchKernelGGL((pow_tail_kernel<MODE>), dim3(grid), dim3(AGPU_BLOCK), 0, p->stream, pa, pb, po, done, n, tab);
  }
  AGPU_LAUNCH_CHECK();
  return AGPU_OK;
}


// ---------------------------------------------------------------- self-test of the f32 functions (agpu_selftest_unary_f32)
// Every f32 bit pattern in [first, first + count) through the product's functor and through the f64 device library
// rounded once to f32; the largest distance in ULPs (monotone integer order, ±0 equal, NaN only against NaN) and a bit
// pattern that attains it.  2^32 patterns take seconds on the GPU — the CPU oracle covers 2^24 samples per test.
template <typename F, int REF>
__global__ __launch_bounds__(AGPU_BLOCK) void selftest_unary_kernel(uint64_t first, uint64_t count, unsigned long long* worst) {
  unsigned long long w = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * AGPU_BLOCK + threadIdx.x; i < count; i += (uint64_t)gridDim.x * AGPU_BLOCK) {
    const uint32_t bits = (uint32_t)(first + i);
    const float x = __builtin_bit_cast(float, bits);
    const float got = F::ap(x, 0.0f);
    const double xd = (double)x;
    double rd;
    if constexpr (REF == AGPU_UN_SIN) rd = sin(xd);
    else if constexpr (REF == AGPU_UN_COS) rd = cos(xd);
    else if constexpr (REF == AGPU_UN_LOG) rd = log(xd);
    else if constexpr (REF == AGPU_UN_LOG2) rd = log2(xd);
    else if constexpr (REF == AGPU_UN_EXP) rd = exp(xd);
    else if constexpr (REF == AGPU_UN_EXP2) rd = exp2(xd);
    else if constexpr (REF == AGPU_UN_SINH) rd = sinh(xd);
    else if constexpr (REF == AGPU_UN_CBRT) rd = cbrt(xd);
    else if constexpr (REF == AGPU_UN_ACOS) rd = acos(xd);
    else rd = sqrt(xd);
    const float ref = (float)rd;
    uint32_t d;
    if (got != got || ref != ref) d = (got != got && ref != ref) ? 0u : 0xFFFFFFFFu;
    else {
      const int32_t gb = __builtin_bit_cast(int32_t, got), rb = __builtin_bit_cast(int32_t, ref);
      const int64_t go = gb < 0 ? -(int64_t)(gb & 0x7fffffff) : gb, ro = rb < 0 ? -(int64_t)(rb & 0x7fffffff) : rb;
      const int64_t dd = go > ro ? go - ro : ro - go;
      d = dd > 0xFFFFFFFFll ? 0xFFFFFFFFu : (uint32_t)dd;
    }
    const unsigned long long key = ((unsigned long long)d << 32) | bits;
    if (key > w && d > 0) w = key;
  }
  if (w) atomicMax(worst, w);
}
agpu_status agpu_selftest_unary_f32(agpu_pipeline* p, agpu_unary_op op, uint64_t first_bits, uint64_t count, uint32_t* out_max_ulp,
                                    uint32_t* out_worst_bits) {
  AGPU_BIND(p);
  AGPU_REQUIRE(out_max_ulp && first_bits + count <= (1ull << 32), AGPU_ERR_ARG, "bad range");
  void* scratch = nullptr;
  agpu_status st = agpu_scratch(p, 8, &scratch);
  if (st != AGPU_OK) return st;
  unsigned long long* worst = static_cast<unsigned long long*>(scratch);
  AGPU_HIP(hipMemsetAsync(worst, 0, 8, p->stream));
  const int grid = (int)std::min<uint64_t>((count + AGPU_BLOCK - 1) / AGPU_BLOCK, (uint64_t)p->dev->num_cus * 32);
#define ST_CASE(OPV, FUNCTOR)                                                                                                     \
  case OPV:                                                                                                                       \
    if (count) hipLaunchKernelGGL((selftest_unary_kernel<FUNCTOR, OPV>), dim3(grid), dim3(AGPU_BLOCK), 0, p->stream, first_bits, count, worst); \
    break;
  switch (op) {
    ST_CASE(AGPU_UN_SIN, UnSin) ST_CASE(AGPU_UN_COS, UnCos) ST_CASE(AGPU_UN_LOG, UnLog) ST_CASE(AGPU_UN_LOG2, UnLog2)
    ST_CASE(AGPU_UN_EXP, UnExp) ST_CASE(AGPU_UN_EXP2, UnExp2) ST_CASE(AGPU_UN_SINH, UnSinh) ST_CASE(AGPU_UN_CBRT, UnCbrt)
    ST_CASE(AGPU_UN_ACOS, UnAcos) ST_CASE(AGPU_UN_SQRT, UnSqrt)
    default: agpu_set_error("no f64 reference for unary op %d", (int)op); return AGPU_ERR_UNSUPPORTED;
  }
#undef ST_CASE
  AGPU_LAUNCH_CHECK();
  unsigned long long host = 0;
  AGPU_HIP(hipMemcpyAsync(&host, worst, 8, hipMemcpyDeviceToHost, p->stream));
  AGPU_HIP(hipStreamSynchronize(p->stream));
  *out_max_ulp = (uint32_t)(host >> 32);
  if (out_worst_bits) *out_worst_bits = (uint32_t)host;
  return AGPU_OK;
}

// pow has 2^64 operand pairs: `count` pseudo-random ones instead (counter-based, reproducible from `seed`).
//   domain 0: x any positive f32 bit pattern (denormals, inf and NaN included), |y| < 2^8 with a random exponent
//   domain 1: x within 2^13 ULPs of 1.0, |y| up to 2^30 — where log2 x needs its relative accuracy
//   domain 2: x ∈ [2^-3, 2^3), |y| < 2^7 — results across the whole exponent range incl. overflow / underflow
__device__ __forceinline__ uint32_t selftest_mix(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return (uint32_t)((z ^ (z >> 31)) >> 16);
}
__global__ __launch_bounds__(AGPU_BLOCK) void selftest_pow_kernel(uint64_t seed, uint64_t count, int domain, const PowTab* tab,
                                                                 unsigned long long* worst_d, unsigned long long* worst_xy) {
  unsigned long long wd = 0, wxy = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * AGPU_BLOCK + threadIdx.x; i < count; i += (uint64_t)gridDim.x * AGPU_BLOCK) {
    const uint32_t r0 = selftest_mix(seed + 2 * i + 0x9e3779b97f4a7c15ull), r1 = selftest_mix(seed + 2 * i + 1);
    uint32_t xb, yb;
    if (domain == 0) {
      xb = r0 & 0x7fffffffu;
      yb = (r1 & 0x807fffffu) | ((100u + (r1 >> 23) % 35u) << 23);  // exponent 2^-27 … 2^7
    } else if (domain == 1) {
      xb = 0x3f800000u + (r0 & 0x3fffu) - 0x2000u;
      yb = (r1 & 0x807fffffu) | ((127u + (r1 >> 23) % 31u) << 23);  // 1 … 2^30
    } else {
      xb = (r0 & 0x007fffffu) | ((124u + (r0 >> 23) % 6u) << 23);
      yb = (r1 & 0x807fffffu) | ((118u + (r1 >> 23) % 16u) << 23);  // 2^-9 … 2^6
    }
    const float x = __builtin_bit_cast(float, xb), y = __builtin_bit_cast(float, yb);
    const float got = pow_f32_dev(tab, x, y);
    const float ref = (float)pow((double)x, (double)y);
    uint32_t d;
    if (got != got || ref != ref) d = (got != got && ref != ref) ? 0u : 0xFFFFFFFFu;
    else {
      const int32_t gb = __builtin_bit_cast(int32_t, got), rb = __builtin_bit_cast(int32_t, ref);
      const int64_t go = gb < 0 ? -(int64_t)(gb & 0x7fffffff) : gb, ro = rb < 0 ? -(int64_t)(rb & 0x7fffffff) : rb;
      const int64_t dd = go > ro ? go - ro : ro - go;
      d = dd > 0xFFFFFFFFll ? 0xFFFFFFFFu : (uint32_t)dd;
    }
    if (d > wd) {
      wd = d;
      wxy = ((unsigned long long)xb << 32) | yb;
    }
  }
  if (wd && atomicMax(worst_d, wd) < wd) atomicExch(worst_xy, wxy);  // (the pair is "a" worst one: good enough for a report)
}
agpu_status agpu_selftest_pow_f32(agpu_pipeline* p, uint64_t seed, uint64_t count, int32_t domain, uint32_t* out_max_ulp,
                                  uint32_t* out_worst_x_bits, uint32_t* out_worst_y_bits) {
  AGPU_BIND(p);
  AGPU_REQUIRE(out_max_ulp && domain >= 0 && domain <= 2, AGPU_ERR_ARG, "bad argument");
  void* scratch = nullptr;
  agpu_status st = agpu_scratch(p, 16, &scratch);
  if (st != AGPU_OK) return st;
  unsigned long long* w = static_cast<unsigned long long*>(scratch);
  AGPU_HIP(hipMemsetAsync(w, 0, 16, p->stream));
  const int grid = (int)std::min<uint64_t>((count + AGPU_BLOCK - 1) / AGPU_BLOCK, (uint64_t)p->dev->num_cus * 32);
  if (count)
    hipLaunchKernelGGL(selftest_pow_kernel, dim3(grid), dim3(AGPU_BLOCK), 0, p->stream, seed, count, (int)domain,
                       static_cast<const PowTab*>(p->dev->pow_table), w, w + 1);
  AGPU_LAUNCH_CHECK();
  unsigned long long host[2] = {0, 0};
  AGPU_HIP(hipMemcpyAsync(host, w, 16, hipMemcpyDeviceToHost, p->stream));
  AGPU_HIP(hipStreamSynchronize(p->stream));
  *out_max_ulp = (uint32_t)host[0];
  if (out_worst_x_bits) *out_worst_x_bits = (uint32_t)(host[1] >> 32);
  if (out_worst_y_bits) *out_worst_y_bits = (uint32_t)host[1];
  return AGPU_OK;
}

// ---------------------------------------------------------------- dispatch: op × dtype
template <typename T, int MODE>
static agpu_status dispatch_int_op(agpu_pipeline* p, agpu_binary_op op, const void* a, const void* b, void* out,
                                   uint64_t n) {
  constexpr bool W32 = sizeof(T) == 4;
  switch (op) {
    case AGPU_OP_ADD: return launch_ew<T, OpAdd, MODE>(p, a, b, out, n);
    case AGPU_OP_SUB: return launch_ew<T, OpSub, MODE>(p, a, b, out, n);
    case AGPU_OP_MUL: return launch_ew<T, OpMul, MODE>(p, a, b, out, n);
    case AGPU_OP_MIN: return launch_ew<T, OpMin, MODE>(p, a, b, out, n);
    case AGPU_OP_MAX: return launch_ew<T, OpMax, MODE>(p, a, b, out, n);
    case AGPU_OP_AND: return launch_ew<T, OpAnd, MODE>(p, a, b, out, n);
    case AGPU_OP_OR: return launch_ew<T, OpOr, MODE>(p, a, b, out, n);
    case AGPU_OP_XOR: return launch_ew<T, OpXor, MODE>(p, a, b, out, n);
    case AGPU_OP_SHL:
      if constexpr (W32) return launch_ew<T, OpShl, MODE>(p, a, b, out, n);
      else return launch_shift<T, MODE>(p, true, a, b, out, n);
    case AGPU_OP_SHR:
      if constexpr (W32) return launch_ew<T, OpShr, MODE>(p, a, b, out, n);
      else return launch_shift<T, MODE>(p, false, a, b, out, n);
    case AGPU_OP_DIV:
      if constexpr (W32) return launch_ew<T, OpDiv, MODE>(p, a, b, out, n);
      break;
    case AGPU_OP_REM:
      if constexpr (W32) return launch_ew<T, OpRem, MODE>(p, a, b, out, n);
      break;
    case AGPU_OP_POW:
      if constexpr (W32 && std::is_signed<T>::value) return launch_ew<T, OpPow, MODE>(p, a, b, out, n);
      break;
    default: break;
  }
  agpu_set_error("binary op %d not supported for this integer dtype", (int)op);
  return AGPU_ERR_UNSUPPORTED;
}

template <int MODE>
static agpu_status dispatch_f32_op(agpu_pipeline* p, agpu_binary_op op, const void* a, const void* b, void* out,
                                   uint64_t n) {
  switch (op) {
    case AGPU_OP_ADD: return launch_ew<float, OpAdd, MODE>(p, a, b, out, n);
    case AGPU_OP_SUB: return launch_ew<float, OpSub, MODE>(p, a, b, out, n);
    case AGPU_OP_MUL: return launch_ew<float, OpMul, MODE>(p, a, b, out, n);
    case AGPU_OP_DIV: return launch_ew<float, OpDiv, MODE>(p, a, b, out, n);
    case AGPU_OP_REM: return launch_ew<float, OpRem, MODE>(p, a, b, out, n);
    case AGPU_OP_MIN: return launch_ew<float, OpMin, MODE>(p, a, b, out, n);
    case AGPU_OP_MAX: return launch_ew<float, OpMax, MODE>(p, a, b, out, n);
    case AGPU_OP_POW: return launch_pow_f32<MODE>(p, a, b, out, n);
    default: break;
  }
  agpu_set_error("binary op %d not supported for f32", (int)op);
  return AGPU_ERR_UNSUPPORTED;
}

template <int MODE>
static agpu_status dispatch_binary(agpu_pipeline* p, agpu_binary_op op, agpu_dtype dtype, const void* a, const void* b,
                                   void* out, uint64_t n) {
  AGPU_BIND_AS(p, MODE == MODE_SCALAR ? "agpu_scalar" : "agpu_binary");
  AGPU_REQUIRE(n == 0 || (a && b && out), AGPU_ERR_ARG, "null pointer");
  switch (dtype) {
    case AGPU_F32: return dispatch_f32_op<MODE>(p, op, a, b, out, n);
    case AGPU_I32: case AGPU_DATE32: return dispatch_int_op<int32_t, MODE>(p, op, a, b, out, n);
    case AGPU_U32: return dispatch_int_op<uint32_t, MODE>(p, op, a, b, out, n);
    case AGPU_I16: return dispatch_int_op<int16_t, MODE>(p, op, a, b, out, n);
    case AGPU_U16: return dispatch_int_op<uint16_t, MODE>(p, op, a, b, out, n);
    case AGPU_I8: return dispatch_int_op<int8_t, MODE>(p, op, a, b, out, n);
    case AGPU_U8: return dispatch_int_op<uint8_t, MODE>(p, op, a, b, out, n);
    default: break;
  }
  agpu_set_error("dtype %d not supported for element-wise binary ops", (int)dtype);
  return AGPU_ERR_UNSUPPORTED;
}

// ---------------------------------------------------------------- width-changing kernel (casts, fused int→f32 trig)
// lane handles N = 16/max(sizeof(TI),sizeof(TO)) elements per step: the wide side moves 16 B/lane, the narrow
// side N*sizeof bytes (4 or 8) — both sides stay fully coalesced (the reference's cast shaders store with stride 4).
// Shape: the narrow side moves only 4–8 B per lane, so a one-wave block would issue 256-byte loads; 256-thread blocks
// with 4 steps per lane (what the 8-bit table kernel below uses: 5.64 vs 5.47 TB/s on the 5 B/row stream) do better.
#define AGPU_CVT_BLOCK 256
#define AGPU_CVT_U 4
template <typename TI, typename TO, typename Conv, int U>
__global__ __launch_bounds__(AGPU_CVT_BLOCK) void cvt_kernel(const TI* in, TO* out, uint64_t ntiles) {
  constexpr int N = 16 / (sizeof(TI) > sizeof(TO) ? sizeof(TI) : sizeof(TO));
  constexpr uint64_t tile = (uint64_t)AGPU_CVT_BLOCK * U;
  for (uint64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const uint64_t p0 = t * tile + threadIdx.x;
    PackN<TI, N> v[U];
    static_for<U>([&](auto u) {
      v[u] = load_pack<(AGPU_STREAM_NT & 1) != 0, TI, N>(in + (p0 + (uint64_t)u * AGPU_CVT_BLOCK) * N);
    });
    static_for<U>([&](auto u) {
      PackN<TO, N> r;
#pragma unroll
      for (int k = 0; k < N; k++) r.v[k] = Conv::ap(v[u].v[k]);
      store_pack<(AGPU_STREAM_NT & 2) != 0, TO, N>(out + (p0 + (uint64_t)u * AGPU_CVT_BLOCK) * N, r);
    });
  }
}
template <typename TI, typename TO, typename Conv>
__global__ __launch_bounds__(AGPU_BLOCK) void cvt_tail_kernel(const TI* in, TO* out, uint64_t first, uint64_t n) {
  constexpr int N = 16 / (sizeof(TI) > sizeof(TO) ? sizeof(TI) : sizeof(TO));
  const uint64_t npacks = n / N;
  for (uint64_t pk = first / N + threadIdx.x; pk < npacks; pk += AGPU_BLOCK) {
    PackN<TI, N> x = load_pack<false, TI, N>(in + pk * N);
    PackN<TO, N> r;
#pragma unroll
    for (int k = 0; k < N; k++) r.v[k] = Conv::ap(x.v[k]);
    store_pack<false, TO, N>(out + pk * N, r);
  }
  const uint64_t i = npacks * N + threadIdx.x;
  if (i >= first && i < n) out[i] = Conv::ap(in[i]);
}
template <typename TI, typename TO, typename Conv>
__global__ __launch_bounds__(AGPU_BLOCK) void cvt_kernel_unaligned(const TI* in, TO* out, uint64_t n) {
  for (uint64_t i = (uint64_t)blockIdx.x * AGPU_BLOCK + threadIdx.x; i < n; i += (uint64_t)gridDim.x * AGPU_BLOCK)
    out[i] = Conv::ap(in[i]);
}

// Widening (u8→f32, i8→i32, u8→u16, u16→f32 …): with cvt_kernel the narrow side moves 4–8 B per lane (256–512-byte
// wave loads).  Here every lane loads a full 16-byte vector (the wave takes one contiguous 1 KiB chunk) and the chunk is
// transposed INSIDE the wave so that each of the R = sizeof(TO)/sizeof(TI) stores is a fully coalesced 1 KiB row: store
// j, lane l needs the PIECE = 16/R input bytes at chunk offset (64 j + l)·PIECE, which sit in lane (64 j + l)/R at piece
// l mod R — four ds_bpermute_b32 (one per dword of the source vector; the LDS crossbar, no LDS memory, no barrier) and
// a select.  One wave per block.  Measured on cast u8→f32 at 1e9 rows, same process (tools/probe/cast_sweep.py,
// profiles/r02_sweep_cast.json): 4-byte loads 5.4–5.9 TB/s, transposition through 1 KiB of LDS 5.9, this form 6.36.
// threads per block: one wave (u8→f32 0.787, i16→f32 0.785); 128: 0.77 / 0.79; 256: 0.70 / 0.76 (tools/probe/cast_shape.py)
#ifndef AGPU_CVTW_BLOCK
#define AGPU_CVTW_BLOCK 64
#endif

// SC1: the stores' cache policy.  ×2: always sc1 nt.  ×4: sc1 nt when the launch carries the occupancy cap (u8→f32 / i8→i32 at 1e9 rows,
// three processes, tools/archive/r05_sc1x4.sh: 0.794–0.817 → 0.808–0.826 of the roof), plain nt without it (0.785–0.800 against 0.760–0.797).
template <typename TI, typename TO, typename Conv, bool SC1>
__global__ __launch_bounds__(AGPU_CVTW_BLOCK) void cvt_wide_kernel(const TI* in, TO* out, uint64_t nchunks, uint64_t half) {
  constexpr int R = sizeof(TO) / sizeof(TI);   // 2 or 4 stores per load
  constexpr int NO = 16 / sizeof(TO);          // output elements per lane per store
  constexpr uint32_t WAVES = AGPU_CVTW_BLOCK / AGPU_WAVE;
  const uint32_t lane = threadIdx.x & (AGPU_WAVE - 1);
  const TileRun run = tile_run((uint64_t)blockIdx.x * WAVES + threadIdx.x / AGPU_WAVE, (uint64_t)gridDim.x * WAVES, nchunks);
  uint64_t c = run.t;
  if (c >= run.end) return;
  // the next chunk's load is issued before the current chunk's four stores (tuning tiles > 1: a wave walks several chunks)
  u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(in) + two_streams(c, half) * AGPU_WAVE + lane);
  for (;;) {
    const uint64_t cn = c + run.step;
    const bool more = cn < run.end;
    const uint64_t cp = two_streams(c, half);  // the chunk this round stores
    u32x4 vn = v;
    if (more) vn = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(in) + two_streams(cn, half) * AGPU_WAVE + lane);
    static_for<R>([&](auto j) {
      const int src = (int)(((uint32_t)j * (AGPU_WAVE / R) + lane / R) * 4);  // byte address of the source lane
      const uint32_t w0 = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)v.x);
      const uint32_t w1 = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)v.y);
      const uint32_t w2 = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)v.z);
      const uint32_t w3 = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)v.w);
      PackN<TI, NO> x;
      if constexpr (R == 4) {
        const uint32_t sel = lane & 3u;
        const uint32_t w = sel == 0 ? w0 : sel == 1 ? w1 : sel == 2 ? w2 : w3;
        x = __builtin_bit_cast(PackN<TI, NO>, w);
      } else {
        const bool hi = (lane & 1u) != 0;
        const u32x2 w = {hi ? w2 : w0, hi ? w3 : w1};
        x = __builtin_bit_cast(PackN<TI, NO>, w);
      }
      PackN<TO, NO> r;
      if constexpr (ConvHasRows<Conv>::value && NO % 2 == 0) {
        Conv::template ap_rows<NO>(x.v, r.v);
      } else {
#pragma unroll
        for (int k = 0; k < NO; k++) r.v[k] = Conv::ap(x.v[k]);
      }
      const uint32_t g = (uint32_t)j * AGPU_WAVE + lane;  // slot of this lane's store inside the chunk
      store_pack<(AGPU_STREAM_NT & 2) != 0, TO, NO, SC1>(out + (cp * (uint64_t)(AGPU_WAVE * R) + g) * NO, r);
    });
    if (!more) break;
    v = vn;
    c = cn;
  }
}

// Narrowing (f32→u8 / i8 / i16 / u16): the mirror image.  With cvt_kernel the narrow side STORES 4–8 B per lane (256–512-byte
// wave stores).  Here the wave takes R = sizeof(TI)/sizeof(TO) contiguous 1 KiB chunks with coalesced 16-byte loads, every
// loaded vector becomes one 16/R-byte piece, and the 64·R pieces are transposed inside the wave (ds_bpermute: piece
// q = R·l' + k of output lane l' was produced by load q / 64 in lane q mod 64) so that each lane stores 16 contiguous bytes:
// ONE coalesced 1 KiB store per wave.  cast f32→u8 at 1e9 rows, same process (tools/probe/narrow_sweep.py,
// profiles/r02_sweep_narrow.json): 4-byte stores 0.77–0.79 of the HBM roof in every block shape, this form 0.83.
template <typename TI, typename TO, typename Conv>
__global__ __launch_bounds__(AGPU_WAVE) void cvt_narrow_kernel(const TI* in, TO* out, uint64_t nchunks, uint64_t half) {
  constexpr int R = sizeof(TI) / sizeof(TO);  // 2 or 4 loads per store
  constexpr int NI = 16 / sizeof(TI);         // elements per loaded vector = elements per piece
  constexpr int PD = 4 / R;                   // dwords per piece: 1 (R = 4) or 2 (R = 2)
  struct Piece {
    uint32_t d[PD];
  };
  const uint32_t lane = threadIdx.x;
  for (uint64_t c0 = blockIdx.x; c0 < nchunks; c0 += gridDim.x) {
    const uint64_t c = two_streams(c0, half);
    Piece w[R];
    static_for<R>([&](auto j) {
      const PackN<TI, NI> x = load_pack<(AGPU_STREAM_NT & 1) != 0, TI, NI>(in + ((c * R + (uint32_t)j) * AGPU_WAVE + lane) * NI);
      PackN<TO, NI> r;
#pragma unroll
      for (int k = 0; k < NI; k++) r.v[k] = Conv::ap(x.v[k]);
      w[j] = __builtin_bit_cast(Piece, r);
    });
    const uint32_t jsel = lane / (AGPU_WAVE / R);
    u32x4 outv;
    static_for<R>([&](auto k) {
      const int src = (int)((R * (lane % (AGPU_WAVE / R)) + (uint32_t)k) * 4u);
      static_for<PD>([&](auto d) {
        uint32_t got[R];
        static_for<R>([&](auto j) { got[j] = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)w[j].d[d]); });
        uint32_t pick = got[0];
        static_for<R>([&](auto j) {
          if ((uint32_t)j == jsel) pick = got[j];
        });
        outv[(int)k * PD + (int)d] = pick;
      });
    });
    __builtin_nontemporal_store(outv, reinterpret_cast<u32x4*>(out) + c * AGPU_WAVE + lane);
  }
}

template <typename TI, typename TO, typename Conv>
static agpu_status launch_cvt(agpu_pipeline* p, const void* in, void* out, uint64_t n) {
  if (n == 0) return AGPU_OK;
  constexpr int N = 16 / (sizeof(TI) > sizeof(TO) ? sizeof(TI) : sizeof(TO));
  const TI* pi = static_cast<const TI*>(in);
  TO* po = static_cast<TO*>(out);
  if constexpr (sizeof(TO) == 2 * sizeof(TI) || sizeof(TO) == 4 * sizeof(TI)) {
    if (aligned16(in) && aligned16(out)) {
      constexpr uint64_t chunk_rows = (uint64_t)AGPU_WAVE * 16 / sizeof(TI);
      const uint64_t nchunks = n / chunk_rows;
      if (nchunks) {
        // chunks per wave (next chunk prefetched): 1 by default — 2 is +5 % in lucky allocations and −8 % in others (tile_run above)
        const uint64_t k = p->tune.tiles > 0 ? (uint64_t)p->tune.tiles : 1;
        const uint64_t blocks = (nchunks + AGPU_CVTW_BLOCK / AGPU_WAVE - 1) / (AGPU_CVTW_BLOCK / AGPU_WAVE);
        const int grid = stream_grid_for(p, (blocks + k - 1) / k);
        // occupancy cap (common.hpp): ×2 casts to 32 bits ≈ 24 waves per CU, ×4 ≈ 16; u8 → u16 none (not measured to gain)
        // (round 6: none for cast → sin / cos of a 16-bit column — with sin / cos in packed f32 the kernel wants all its waves: 0.79–0.80 against 0.75)
        constexpr unsigned cap = ConvHasRows<Conv>::value ? 0u : sizeof(TO) == 4 ? (sizeof(TI) == 2 ? AGPU_WAVE_LDS_24 : AGPU_WAVE_LDS_16) : 0u;
        const unsigned lds = cap ? wave_lds_for(p, cap, AGPU_CVTW_BLOCK / AGPU_WAVE) : 0u;
        // ×2 casts walk the chunks as two lock-step streams (common.hpp two_streams: u16 → f32 0.836 → 0.857, u8 → u16 0.80 → 0.83, sin_u16 0.78 → 0.79);
        // the ×4 ones do not gain (u8 → f32 0.811 / 0.806)
        const uint64_t half = sizeof(TO) == 2 * sizeof(TI) ? two_streams_half(p, nchunks, 2048) : 0;
        if (sizeof(TO) == 2 * sizeof(TI) || lds >= AGPU_WAVE_LDS_24)
          hipLaunchKernelGGL((cvt_wide_kernel<TI, TO, Conv, true>), dim3(grid), dim3(AGPU_CVTW_BLOCK), lds, p->stream, pi, po, nchunks, half);
        else
          hipLaunchKernelGGL((cvt_wide_kernel<TI, TO, Conv, false>), dim3(grid), dim3(AGPU_CVTW_BLOCK), lds, p->stream, pi, po, nchunks, half);
      }
      if (nchunks * chunk_rows < n)
        hipLaunchKernelGGL((cvt_tail_kernel<TI, TO, Conv>), dim3(1), dim3(AGPU_BLOCK), 0, p->stream, pi, po,
                           nchunks * chunk_rows, n);
      AGPU_LAUNCH_CHECK();
      return AGPU_OK;
    }
  }
  if constexpr (sizeof(TI) == 2 * sizeof(TO) || sizeof(TI) == 4 * sizeof(TO)) {
    if (aligned16(in) && aligned16(out)) {
      constexpr uint64_t chunk_rows = (uint64_t)AGPU_WAVE * 16 / sizeof(TO);  // rows behind one 1 KiB wave store
      const uint64_t nchunks = n / chunk_rows;
      if (nchunks) {
        const int grid = stream_grid_for(p, nchunks);
        hipLaunchKernelGGL((cvt_narrow_kernel<TI, TO, Conv>), dim3(grid), dim3(AGPU_WAVE), 0, p->stream, pi, po, nchunks,
                           sizeof(TI) == 2 * sizeof(TO) ? two_streams_half(p, nchunks, 2048) : 0);  // f32 → i16 0.845 → 0.858; f32 → u8 level
      }
      if (nchunks * chunk_rows < n)
        hipLaunchKernelGGL((cvt_tail_kernel<TI, TO, Conv>), dim3(1), dim3(AGPU_BLOCK), 0, p->stream, pi, po,
                           nchunks * chunk_rows, n);
      AGPU_LAUNCH_CHECK();
      return AGPU_OK;
    }
  }
  if (aligned_to(in, sizeof(TI) * N) && aligned_to(out, sizeof(TO) * N)) {
    constexpr uint64_t tile_rows = (uint64_t)AGPU_CVT_BLOCK * AGPU_CVT_U * N;
    const uint64_t ntiles = n / tile_rows;
    if (ntiles) {
      const int grid = stream_grid_for(p, ntiles);
      hipLaunchKernelGGL((cvt_kernel<TI, TO, Conv, AGPU_CVT_U>), dim3(grid), dim3(AGPU_CVT_BLOCK), 0, p->stream, pi, po,
                         ntiles);
    }
    if (ntiles * tile_rows < n)
      hipLaunchKernelGGL((cvt_tail_kernel<TI, TO, Conv>), dim3(1), dim3(AGPU_BLOCK), 0, p->stream, pi, po,
                         ntiles * tile_rows, n);
  } else {
    const int grid = stream_grid_for(p, (n + AGPU_BLOCK - 1) / AGPU_BLOCK);
    hipLaunchKernelGGL((cvt_kernel_unaligned<TI, TO, Conv>), dim3(grid), dim3(AGPU_BLOCK), 0, p->stream, pi, po, n);
  }
  AGPU_LAUNCH_CHECK();
  return AGPU_OK;
}

template <typename TI, typename TO>
struct CvtStatic {  // widen by source signedness, then reinterpret [cast/src/i8_cast.rs:72-89]; int→f32 exact
  __device__ static __forceinline__ TO ap(TI x) { return (TO)x; }
};
struct CvtF32ToU8 {  // trunc toward 0, clamp to [0, 2^32-1], NaN→0, then mod 256 [cast/compute_shaders/f32/cast_u8.wgsl]
  __device__ static __forceinline__ uint8_t ap(float x) {
    uint32_t u;
    if (!(x > 0.0f)) u = 0;
    else if (x >= 4294967296.0f) u = 0xFFFFFFFFu;
    else u = (uint32_t)x;
    return (uint8_t)(u & 255u);
  }
};
// f32 → i8 / i16 / u16 / i32 / u32: REFERENCE-ABSENT (include/arrow_gpu.h agpu_cast) — WGSL u32(x) / i32(x) by the target's
// signedness (trunc toward 0, clamp, NaN → 0), then the low bits of the target width; f32 → u8 above is the same rule.
template <typename TO>
struct CvtF32ToInt {
  __device__ static __forceinline__ TO ap(float x) {
    uint32_t bits;
    if constexpr (std::is_signed<TO>::value) {
      int32_t i;
      if (x != x) i = 0;
      else if (x >= 2147483648.0f) i = INT32_MAX;
      else if (x <= -2147483648.0f) i = INT32_MIN;
      else i = (int32_t)x;
      bits = (uint32_t)i;
    } else {
      if (!(x > 0.0f)) bits = 0;
      else if (x >= 4294967296.0f) bits = 0xFFFFFFFFu;
      else bits = (uint32_t)x;
    }
    return (TO)(U_of<TO>)bits;
  }
};
template <typename TI, typename F>
struct CvtThenF32 {  // fused sin_u8-style kernels [trigonometry/compute_shaders/{u8,i8,u16,i16}/*.wgsl]
  __device__ static __forceinline__ float ap(TI x) {
    // |x| ≤ 65 535 < 1e6: sin / cos never take the out-of-line |x| ≥ 1e6 path — the compiler sees that for u16 and not for i16; same bits either way
    if constexpr (sizeof(TI) <= 2 && std::is_same<F, UnSin>::value) return sincos_f32_fast<0>((float)x);
    else if constexpr (sizeof(TI) <= 2 && std::is_same<F, UnCos>::value) return sincos_f32_fast<1>((float)x);
    else return F::ap((float)x, 0.0f);
  }
  // several rows at a time: sin / cos evaluate row PAIRS in packed f32 (sincos_f32_pair); same bits as ap()
  static constexpr bool has_rows = sizeof(TI) <= 2 && (std::is_same<F, UnSin>::value || std::is_same<F, UnCos>::value || std::is_same<F, UnLog>::value);
  template <int N> __device__ static __forceinline__ void ap_rows(const TI (&x)[N], float (&r)[N]) {
    static_assert(N % 2 == 0, "rows come in pairs");
    if constexpr (std::is_same<F, UnLog>::value) {  // (zeros and negatives of the integer column take log's general form inside)
      float xf[N];
#pragma unroll
      for (int k = 0; k < N; k++) xf[k] = (float)x[k];
      log_f32_rows<N>(xf, r);
    } else {
#pragma unroll
      for (int k = 0; k < N; k += 2) {
        const f32x2_t v = sincos_f32_pair<std::is_same<F, UnCos>::value ? 1 : 0>((f32x2_t){(float)x[k], (float)x[k + 1]});
        r[k] = v.x;
        r[k + 1] = v.y;
      }
    }
  }
};

// 8-bit sources (sin_u8 / cos_i8 / sinh_u8 …): only 256 distinct inputs exist, so evaluating the function per row
// makes a 5 B/row stream VALU-bound (3.5 TB/s measured).  Each block builds the 256-entry result table in
// LDS once (the SAME device function as the f32 kernel ⇒ identical bits), then streams
// its tiles (16 rows per lane): every wave takes one contiguous 1 KiB chunk with 16-byte loads and transposes it inside the wave
// (cvt_wide_kernel's ds_bpermute shape) so that each of its 4 nontemporal stores is a coalesced 1 KiB row; 4 LDS reads
// per store.  The next tile's load is issued before the current tile's lookups.
// 128 threads (each builds two table entries): 0.78 of the roof against 0.75–0.77 at 256 and 0.78 at 64 (three
// alternations on one box) — four stores per lane want small blocks (tools/probe/store_probe.hip)
// Round 3: the table is no longer EVALUATED per block (256 function calls per 2048-row tile were two thirds of the block's
// instructions: sin_u8 ran at 0.745 of the roof where the plain u8 → f32 cast ran at 0.79 on the same buffers) — it is built
// once per device by lut8_build_kernel with the same device functions (agpu_device::lut8_tables) and a block copies its 1 KiB.
#define AGPU_LUT8_BLOCK 128
#ifndef AGPU_CCHAIN_SC1X4
#define AGPU_CCHAIN_SC1X4 1  // cast(u8)·s+s: 0.78–0.81 → 0.80–0.815, same script
#endif
#ifndef AGPU_LUT8_SC1
#define AGPU_LUT8_SC1 1  // sc1 nt stores: sin_u8 / cos_i8 at 1e9 rows 0.790–0.809 → 0.804–0.830 of the roof (tools/archive/r05_sc1x4.sh, three processes)
#endif
template <typename TI, typename F>
__global__ void lut8_build_kernel(float* tab) {
  const uint32_t e = threadIdx.x;
  if (e < 256) tab[e] = F::ap((float)(TI)(uint8_t)e, 0.0f);  // indexed by the raw byte
}
template <typename TI, typename F> struct Lut8Slot;
#define AGPU_LUT8_SLOT(TI, F, K) \
  template <> struct Lut8Slot<TI, F> { static constexpr int value = K; };
AGPU_LUT8_SLOT(uint8_t, UnSin, 0)
AGPU_LUT8_SLOT(uint8_t, UnCos, 1)
AGPU_LUT8_SLOT(uint8_t, UnSinh, 2)
AGPU_LUT8_SLOT(int8_t, UnSin, 3)
AGPU_LUT8_SLOT(int8_t, UnCos, 4)
AGPU_LUT8_SLOT(int8_t, UnSinh, 5)
#undef AGPU_LUT8_SLOT
agpu_status agpu_internal_build_lut8(void* tables) {
  float* t = static_cast<float*>(tables);
#define AGPU_LUT8_BUILD(TI, F) hipLaunchKernelGGL((lut8_build_kernel<TI, F>), dim3(1), dim3(256), 0, nullptr, t + 256 * Lut8Slot<TI, F>::value)
  AGPU_LUT8_BUILD(uint8_t, UnSin);
  AGPU_LUT8_BUILD(uint8_t, UnCos);
  AGPU_LUT8_BUILD(uint8_t, UnSinh);
  AGPU_LUT8_BUILD(int8_t, UnSin);
  AGPU_LUT8_BUILD(int8_t, UnCos);
  AGPU_LUT8_BUILD(int8_t, UnSinh);
#undef AGPU_LUT8_BUILD
  AGPU_HIP(hipGetLastError());
  AGPU_HIP(hipStreamSynchronize(nullptr));
  return AGPU_OK;
}
template <typename TI>
__global__ __launch_bounds__(AGPU_LUT8_BLOCK) void lut8_kernel(const TI* in, float* out, uint64_t ntiles, const float* gtab) {
  constexpr uint32_t WAVES = AGPU_LUT8_BLOCK / AGPU_WAVE;
  __shared__ float lut[256];
  const uint32_t lane = threadIdx.x & (AGPU_WAVE - 1), wave = threadIdx.x / AGPU_WAVE;
  const u32x4* in16 = reinterpret_cast<const u32x4*>(in);
  f32x4* out4 = reinterpret_cast<f32x4*>(out);
  const TileRun run = tile_run(blockIdx.x, gridDim.x, ntiles);
  const uint64_t ntiles_end = run.end;
  uint64_t t = run.t;
  u32x4 v = {0, 0, 0, 0};
  if (t < ntiles_end) v = __builtin_nontemporal_load(in16 + (t * WAVES + wave) * AGPU_WAVE + lane);
  if (threadIdx.x < 64) reinterpret_cast<f32x4*>(lut)[threadIdx.x] = reinterpret_cast<const f32x4*>(gtab)[threadIdx.x];  // 1 KiB from L2
  __syncthreads();
  while (t < ntiles_end) {
    const uint64_t c = t * WAVES + wave;
    const u32x4 cur = v;
    t += run.step;
    if (t < ntiles_end) v = __builtin_nontemporal_load(in16 + (t * WAVES + wave) * AGPU_WAVE + lane);
    static_for<4>([&](auto j) {
      const int src = (int)(((uint32_t)j * 16u + (lane >> 2)) * 4u);
      const uint32_t w0 = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)cur.x);
      const uint32_t w1 = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)cur.y);
      const uint32_t w2 = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)cur.z);
      const uint32_t w3 = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)cur.w);
      const uint32_t sel = lane & 3u;
      const uint32_t x = sel == 0 ? w0 : sel == 1 ? w1 : sel == 2 ? w2 : w3;
      f32x4 r = {lut[x & 255u], lut[(x >> 8) & 255u], lut[(x >> 16) & 255u], lut[x >> 24]};
#if AGPU_LUT8_SC1 && AGPU_USE_SC1
      st_vec_sc1(out4 + c * (AGPU_WAVE * 4) + (uint32_t)j * AGPU_WAVE + lane, r);
#else
      __builtin_nontemporal_store(r, out4 + c * (AGPU_WAVE * 4) + (uint32_t)j * AGPU_WAVE + lane);
#endif
    });
  }
}

template <typename TI, typename F>
static agpu_status launch_lut8(agpu_pipeline* p, const void* in, void* out, uint64_t n) {
  static_assert(sizeof(TI) == 1, "8-bit sources only");
  if (n == 0) return AGPU_OK;
  constexpr uint64_t TILE_ROWS = (uint64_t)AGPU_LUT8_BLOCK * 4 * 4;
  const TI* pi = static_cast<const TI*>(in);
  float* po = static_cast<float*>(out);
  uint64_t done = 0;
  if (aligned16(in) && aligned16(out)) {
    const uint64_t ntiles = n / TILE_ROWS;
    if (ntiles) {
      const uint64_t tk = tab_k(p);
      const int grid = stream_grid_for(p, tile_units(ntiles, tk));
      hipLaunchKernelGGL((lut8_kernel<TI>), dim3(grid), dim3(AGPU_LUT8_BLOCK), wave_lds_for(p, AGPU_WAVE_LDS_24, AGPU_LUT8_BLOCK / AGPU_WAVE), p->stream, pi, po, ntiles,
                         static_cast<const float*>(p->dev->lut8_tables) + 256 * Lut8Slot<TI, F>::value);
      done = ntiles * TILE_ROWS;
    }
    if (done < n)
      hipLaunchKernelGGL((cvt_tail_kernel<TI, float, CvtThenF32<TI, F>>), dim3(1), dim3(AGPU_BLOCK), 0, p->stream, pi, po,
                         done, n);
  } else {
    const int grid = stream_grid_for(p, (n + AGPU_BLOCK - 1) / AGPU_BLOCK);
    hipLaunchKernelGGL((cvt_kernel_unaligned<TI, float, CvtThenF32<TI, F>>), dim3(grid), dim3(AGPU_BLOCK), 0, p->stream,
                       pi, po, n);
  }
  AGPU_LAUNCH_CHECK();
  return AGPU_OK;
}

// 16-bit sources (sin_u16 / cos_i16 …): 65 536 distinct inputs — too many for an LDS result table.  Rounds 1–5 split the integer as
// 256·h + l and combined two f64 {sin, cos} table entries per row (trig16_kernel: an 8 KiB LDS table per block, 0.74–0.78 of the roof, and
// ≈ 65 of the 65 536 results 1 ULP away from what cast → sin gives).  With sin / cos in packed f32 (round 6) evaluating the function per row
// is FASTER than the table — cvt_wide_kernel over CvtThenF32<TI, UnSin>: 0.79–0.81 — and the fused kernel is bit-identical to the unfused pair
// by construction, so the table form is gone (launch_cvt below serves agpu_unary(SIN / COS, u16 / i16)).
agpu_status agpu_internal_build_lut8(void* tables);
agpu_status agpu_internal_build_tables(void* pow_table) {
  {  // the 8-bit result tables live right behind the pow table (common.hpp AGPU_TABLE_BYTES)
    const agpu_status ls = agpu_internal_build_lut8(static_cast<char*>(pow_table) + 128 * 16);
    if (ls != AGPU_OK) return ls;
  }
  hipLaunchKernelGGL(pow_build_kernel, dim3(1), dim3(128), 0, nullptr, static_cast<PowTab*>(pow_table));
  AGPU_HIP(hipGetLastError());
  AGPU_HIP(hipStreamSynchronize(nullptr));
  const PowTab* tp = static_cast<const PowTab*>(pow_table);  // this device's copy of the module gets this device's table
  AGPU_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_pow_tab), &tp, sizeof(tp), 0, hipMemcpyHostToDevice));
  return AGPU_OK;
}

// bool bitmap → f32 [cast/compute_shaders/boolean/cast_f32.wgsl:9-20]: a lane expands 4 bits into ONE nontemporal
// 16-byte store; 256-thread blocks, one store per lane.  A store stream is sensitive to its shape like nothing else
// (tools/probe/store_probe.hip, profiles/r02_store_probe.txt, write-only / with this kernel's bitmap load in front):
// 256 × 1 store 0.85 / 0.85 of the roof, one wave × 4 stores 0.81 / 0.76 (the round-2 shape), 256 × 4 0.74, one wave × 1
// 0.60 (block dispatch), and 256 × 1 with a PLAIN store behind the load 0.57.
#define AGPU_B2F_U 1
__global__ __launch_bounds__(AGPU_BLOCK) void bool_to_f32_kernel(const uint32_t* bits, float* out, uint64_t n) {
  const uint64_t npacks = n / 4;
  constexpr uint64_t TILE = (uint64_t)AGPU_BLOCK * AGPU_B2F_U;
  for (uint64_t t = blockIdx.x; t * TILE < npacks; t += gridDim.x) {
    static_for<AGPU_B2F_U>([&](auto u) {
      const uint64_t pk = t * TILE + (uint64_t)u * AGPU_BLOCK + threadIdx.x;
      if (pk < npacks) {
        const uint32_t w = bits[pk >> 3] >> ((pk & 7) * 4);
        f32x4 r = {(w & 1) ? 1.0f : 0.0f, (w & 2) ? 1.0f : 0.0f, (w & 4) ? 1.0f : 0.0f, (w & 8) ? 1.0f : 0.0f};
        __builtin_nontemporal_store(r, reinterpret_cast<f32x4*>(out + pk * 4));
      }
    });
  }
  if (blockIdx.x == 0) {
    const uint64_t i = npacks * 4 + threadIdx.x;
    if (i < n) out[i] = ((bits[i >> 5] >> (i & 31)) & 1) ? 1.0f : 0.0f;
  }
}

// ---------------------------------------------------------------- broadcast (store-only)
__global__ __launch_bounds__(AGPU_BLOCK) void fill_kernel(uint8_t* out, uint32_t pattern, uint64_t bytes) {
  const uint64_t npacks = bytes / 16;
  u32x4 v = {pattern, pattern, pattern, pattern};
  for (uint64_t pk = (uint64_t)blockIdx.x * AGPU_BLOCK + threadIdx.x; pk < npacks; pk += (uint64_t)gridDim.x * AGPU_BLOCK)
    *reinterpret_cast<u32x4*>(out + pk * 16) = v;
  if (blockIdx.x == 0) {
    const uint64_t i = npacks * 16 + threadIdx.x;
    if (i < bytes) out[i] = (uint8_t)(pattern >> ((i & 3) * 8));
  }
}
agpu_status agpu_internal_fill_bytes(agpu_pipeline* p, void* out, uint32_t pattern, uint64_t bytes) {  // agpu_memset's big fills
  const int grid = stream_grid_for(p, (bytes / 16 + AGPU_BLOCK - 1) / AGPU_BLOCK);
  hipLaunchKernelGGL(fill_kernel, dim3(grid), dim3(AGPU_BLOCK), 0, p->stream, static_cast<uint8_t*>(out), pattern, bytes);
  AGPU_LAUNCH_CHECK();
  return AGPU_OK;
}
// scalar read from a 1-element device buffer (the reference binds it as a storage buffer)
template <typename E>
__global__ __launch_bounds__(AGPU_BLOCK) void fill_from_device_kernel(uint8_t* out, const E* scalar, uint64_t bytes) {
  const E s = scalar[0];
  uint32_t pattern;
  if constexpr (sizeof(E) == 1) pattern = (uint32_t)s * 0x01010101u;
  else if constexpr (sizeof(E) == 2) pattern = (uint32_t)s * 0x00010001u;
  else pattern = (uint32_t)s;
  const uint64_t npacks = bytes / 16;
  u32x4 v = {pattern, pattern, pattern, pattern};
  for (uint64_t pk = (uint64_t)blockIdx.x * AGPU_BLOCK + threadIdx.x; pk < npacks; pk += (uint64_t)gridDim.x * AGPU_BLOCK)
    *reinterpret_cast<u32x4*>(out + pk * 16) = v;
  if (blockIdx.x == 0) {
    const uint64_t i = npacks * 16 + threadIdx.x;
    if (i < bytes) out[i] = (uint8_t)(pattern >> ((i & 3) * 8));
  }
}
// Boolean broadcast: first n_bits set to `value`, padding bits of the 8-byte-granular bitmap zero
__global__ __launch_bounds__(AGPU_BLOCK) void fill_bits_kernel(uint32_t* out, uint32_t value, uint64_t n_bits,
                                                              uint64_t n_words) {
  for (uint64_t w = (uint64_t)blockIdx.x * AGPU_BLOCK + threadIdx.x; w < n_words; w += (uint64_t)gridDim.x * AGPU_BLOCK) {
    const uint64_t lo = w * 32;
    uint32_t m = 0;
    if (value) {
      if (lo + 32 <= n_bits) m = 0xFFFFFFFFu;
      else if (lo < n_bits) m = (1u << (uint32_t)(n_bits - lo)) - 1u;
    }
    out[w] = m;
  }
}

// ---------------------------------------------------------------- fused linear chains (agpu_fused_chain)
// acc = in[i]; acc = op_s(acc, operand_s) for s = 0..n-1 — one pass over HBM for the whole chain.  The step list is
// wave-uniform (kernel argument), so op/kind selection is scalar branching; operands of ARRAY steps are all loaded up
// front (static_for over the 8 slots: registers are addressed at compile time, no dynamic VGPR indexing) before the
// first use, so the chain keeps as many loads in flight as a stand-alone binary kernel.  Each step calls the SAME
// functor as the stand-alone kernel ⇒ bit-identical results.
template <typename T>
__device__ __forceinline__ T chain_apply_binary(int op, T x, T y) {
  switch (op) {
    case AGPU_OP_ADD: return OpAdd::ap(x, y);
    case AGPU_OP_SUB: return OpSub::ap(x, y);
    case AGPU_OP_MUL: return OpMul::ap(x, y);
    case AGPU_OP_DIV: return OpDiv::ap(x, y);
    case AGPU_OP_REM: return OpRem::ap(x, y);
    case AGPU_OP_MIN: return OpMin::ap(x, y);
    case AGPU_OP_MAX: return OpMax::ap(x, y);
    default: break;
  }
  if constexpr (!std::is_floating_point<T>::value) {
    switch (op) {
      case AGPU_OP_AND: return OpAnd::ap(x, y);
      case AGPU_OP_OR: return OpOr::ap(x, y);
      case AGPU_OP_XOR: return OpXor::ap(x, y);
      default: break;
    }
  }
  return x;
}
template <typename T, bool HEAVY>
__device__ __forceinline__ T chain_apply_unary(int op, T x) {
  if constexpr (std::is_floating_point<T>::value && !HEAVY) {
    switch (op) {
      case AGPU_UN_NEG: return UnNeg::ap(x, x);
      case AGPU_UN_ABS: return UnAbs::ap(x, x);
      case AGPU_UN_SQRT: return UnSqrt::ap(x, x);
      default: return x;
    }
  } else if constexpr (std::is_floating_point<T>::value) {
    switch (op) {
      case AGPU_UN_NEG: return UnNeg::ap(x, x);
      case AGPU_UN_ABS: return UnAbs::ap(x, x);
      case AGPU_UN_SQRT: return UnSqrt::ap(x, x);
      case AGPU_UN_CBRT: return UnCbrt::ap(x, x);
      case AGPU_UN_EXP: return UnExp::ap(x, x);
      case AGPU_UN_EXP2: return UnExp2::ap(x, x);
      case AGPU_UN_LOG: return UnLog::ap(x, x);
      case AGPU_UN_LOG2: return UnLog2::ap(x, x);
      case AGPU_UN_SIN: return UnSin::ap(x, x);
      case AGPU_UN_COS: return UnCos::ap(x, x);
      case AGPU_UN_ACOS: return UnAcos::ap(x, x);
      case AGPU_UN_SINH: return UnSinh::ap(x, x);
      default: return x;
    }
  } else {
    switch (op) {
      case AGPU_UN_NEG: return UnNeg::ap(x, x);
      case AGPU_UN_ABS: return UnAbs::ap(x, x);
      case AGPU_UN_NOT: return UnNot::ap(x, x);
      default: return x;
    }
  }
}

// Step codes travel as ONE 64-bit scalar (8 × {4-bit op, 2-bit kind}); operand pointers as kernel arguments, twice: by
// step (scalars, tail kernel) and compacted to the ARRAY steps only (`arrs`).  Nothing is indexed at run time except by
// wave-uniform scalar shifts / selects.
//
// What bounds this kernel is bytes in flight: one-wave blocks with ONE 16-B load per lane and stream need every wave
// slot of the CU (8 per SIMD ⇒ ≤ 64 VGPRs) to cover HBM latency.  The first version (one kernel for everything: 8 operand
// packs + inlined sin/log/exp) needed 86 VGPRs ⇒ 5 waves/SIMD ⇒ 3.7 TB/s on an 8 B/row chain.  So the kernel is
// The same two switches ONE LEVEL UP: over whole packs.  chain_apply_unary / _binary inside a `for k` loop put a copy of the op switch
// behind every element — at 4 elements per pack that was 0.3 taken branches and 0.8 SALU instructions per ROW (rocprofv3, round 5:
// a standalone sin issues 0.02 / 0.08), and a chain with a transcendental step ran at 0.59 of the roof where the standalone function does
// 0.83.  Here the step's op is decided once per tile and the element loops live inside the case.
#define AGPU_CHAIN_UN_CASE(CODE, F)                                                        \
  case CODE:                                                                               \
    static_for<U>([&](auto u) {                                                            \
      if constexpr (std::is_same<T, float>::value && EwRowsOp<F>::value && N % 2 == 0) {   \
        PackN<T, N> in_ = acc[u];                                                          \
        unary_f32_rows<F, N>(in_.v, acc[u].v);                                             \
      } else {                                                                             \
        _Pragma("unroll") for (int k = 0; k < N; k++) acc[u].v[k] = F::ap(acc[u].v[k], acc[u].v[k]); \
      }                                                                                    \
    });                                                                                    \
    return;
template <typename T, bool HEAVY, int U, int N>
__device__ __forceinline__ void chain_apply_unary_packs(int op, PackN<T, N> (&acc)[U]) {
  if constexpr (std::is_floating_point<T>::value) {
    switch (op) {
      AGPU_CHAIN_UN_CASE(AGPU_UN_NEG, UnNeg)
      AGPU_CHAIN_UN_CASE(AGPU_UN_ABS, UnAbs)
      AGPU_CHAIN_UN_CASE(AGPU_UN_SQRT, UnSqrt)
      default: break;
    }
    if constexpr (HEAVY) {
      switch (op) {
        AGPU_CHAIN_UN_CASE(AGPU_UN_CBRT, UnCbrt)
        AGPU_CHAIN_UN_CASE(AGPU_UN_EXP, UnExp)
        AGPU_CHAIN_UN_CASE(AGPU_UN_EXP2, UnExp2)
        AGPU_CHAIN_UN_CASE(AGPU_UN_LOG, UnLog)
        AGPU_CHAIN_UN_CASE(AGPU_UN_LOG2, UnLog2)
        AGPU_CHAIN_UN_CASE(AGPU_UN_SIN, UnSin)
        AGPU_CHAIN_UN_CASE(AGPU_UN_COS, UnCos)
        AGPU_CHAIN_UN_CASE(AGPU_UN_ACOS, UnAcos)
        AGPU_CHAIN_UN_CASE(AGPU_UN_SINH, UnSinh)
        default: break;
      }
    }
  } else {
    switch (op) {
      AGPU_CHAIN_UN_CASE(AGPU_UN_NEG, UnNeg)
      AGPU_CHAIN_UN_CASE(AGPU_UN_ABS, UnAbs)
      AGPU_CHAIN_UN_CASE(AGPU_UN_NOT, UnNot)
      default: break;
    }
  }
}
#undef AGPU_CHAIN_UN_CASE
#define AGPU_CHAIN_BIN_CASE(CODE, F)                                                      \
  case CODE:                                                                              \
    static_for<U>([&](auto u) {                                                           \
      _Pragma("unroll") for (int k = 0; k < N; k++) acc[u].v[k] = F::ap(acc[u].v[k], y[u].v[k]); \
    });                                                                                   \
    return;
template <typename T, int U, int N>
__device__ __forceinline__ void chain_apply_binary_packs(int op, PackN<T, N> (&acc)[U], const PackN<T, N> (&y)[U]) {
  switch (op) {
    AGPU_CHAIN_BIN_CASE(AGPU_OP_ADD, OpAdd)
    AGPU_CHAIN_BIN_CASE(AGPU_OP_SUB, OpSub)
    AGPU_CHAIN_BIN_CASE(AGPU_OP_MUL, OpMul)
    AGPU_CHAIN_BIN_CASE(AGPU_OP_DIV, OpDiv)
    AGPU_CHAIN_BIN_CASE(AGPU_OP_REM, OpRem)
    AGPU_CHAIN_BIN_CASE(AGPU_OP_MIN, OpMin)
    AGPU_CHAIN_BIN_CASE(AGPU_OP_MAX, OpMax)
    default: break;
  }
  if constexpr (!std::is_floating_point<T>::value) {
    switch (op) {
      AGPU_CHAIN_BIN_CASE(AGPU_OP_AND, OpAnd)
      AGPU_CHAIN_BIN_CASE(AGPU_OP_OR, OpOr)
      AGPU_CHAIN_BIN_CASE(AGPU_OP_XOR, OpXor)
      default: break;
    }
  }
}
#undef AGPU_CHAIN_BIN_CASE

// specialised on what costs registers: NARR = operand-pack slots (0/2/4/8, smallest that fits the chain's ARRAY steps)
// and HEAVY = the chain contains a transcendental step (cbrt/exp/exp2/log/log2/sin/cos).  Scalar operands live in SGPRs
// (readfirstlane).  The APPLY loop is a real (uniform) loop over the steps with a single copy of the op switch —
// unrolling it 8× inlined 32 copies of sin/log/exp and ran out of the instruction cache.
struct ChainPtrs {
  const void* p[AGPU_CHAIN_MAX_STEPS];
};
__host__ __device__ __forceinline__ int chain_op(uint64_t code, int s) { return (int)((code >> (6 * s)) & 15u); }
__host__ __device__ __forceinline__ int chain_kind(uint64_t code, int s) { return (int)((code >> (6 * s + 4)) & 3u); }

template <typename T>
__device__ __forceinline__ bool chain_cmp_pred(int op, T x, T y) {
  switch (op) {
    case AGPU_CMP_GT: return x > y;
    case AGPU_CMP_GTEQ: return x >= y;
    case AGPU_CMP_LT: return x < y;
    case AGPU_CMP_LTEQ: return x <= y;
    default: return x == y;
  }
}

// CMP = the chain ends in a compare (agpu_fused_chain_compare): step slot n_steps carries the compare's operand, the
// result is not stored but compared, and the wave's 256 predicate bits leave as eight 32-bit words (4 bits per lane,
// OR-reduced over 8-lane groups) — `out` then points at the packed bitmap.
// The CMP form is (almost) read-only, and a read-only stream of this shape is bound by bytes in flight, not by HBM:
// stores retire without holding the wave, loads do not, and with one 16-byte load per lane and stream a 4-input
// predicate ran at 4.9 TB/s.  It therefore keeps U = 2 packs per lane and stream in flight (6.x TB/s); the storing
// form stays at U = 1 (tools/probe/chain_probe.py).
#define AGPU_CHAIN_CMP_U 2
// Round 5: the storing form of a chain with a transcendental step keeps TWO packs per lane as well — what the standalone sin / cos run
// with; the step dispatch and the eight scalar loads of a tile are paid once per 512 rows instead of 256.
#ifndef AGPU_CHAIN_HEAVY_U
#define AGPU_CHAIN_HEAVY_U 2
#endif
template <typename T, bool HEAVY, int NARR, bool CMP>
__global__ __launch_bounds__(AGPU_EW_BLOCK) void chain_kernel(const T* in, T* out, uint64_t ntiles, int n_steps,
                                                             int n_arrs, uint64_t code, ChainPtrs ptrs, ChainPtrs arrs,
                                                             int cmp_op, uint64_t half) {
  constexpr int N = 4;
  constexpr int U = CMP ? AGPU_CHAIN_CMP_U : HEAVY ? AGPU_CHAIN_HEAVY_U : 1;  // a tile = U × 256 rows
  for (uint64_t t0 = blockIdx.x; t0 < ntiles; t0 += gridDim.x) {
    const uint64_t t = two_streams(t0, half);
    const uint64_t pk0 = t * (uint64_t)(AGPU_EW_BLOCK * U) + threadIdx.x;
    PackN<T, N> acc[U];
    static_for<U>([&](auto u) { acc[u] = load_pack<true, T, N>(in + (pk0 + (uint64_t)u * AGPU_EW_BLOCK) * N); });
    PackN<T, N> ya[NARR > 0 ? NARR : 1][U];
    static_for<NARR>([&](auto a) {
      if (a < n_arrs)
        static_for<U>([&](auto u) {
          ya[a][u] = load_pack<true, T, N>(static_cast<const T*>(arrs.p[a]) + (pk0 + (uint64_t)u * AGPU_EW_BLOCK) * N);
        });
    });
    // Scalar operands: eight unconditional s_load_dword through the constant address space (the host points unused
    // slots at `in`), one lgkmcnt wait for all of them.  A per-lane global load + readfirstlane here serialised one
    // full HBM latency per scalar step (4.4 TB/s).
    uint32_t sc[AGPU_CHAIN_MAX_STEPS];
    static_for<AGPU_CHAIN_MAX_STEPS>([&](auto s) {
      sc[s] = *(const __attribute__((address_space(4))) uint32_t*)(ptrs.p[s]);
    });
    int ai = 0;
    PackN<T, N> y[U];
    auto fetch = [&](int s, int kind) {  // y = what step s combines with (array slot in order of use, or scalar s)
      if (kind == AGPU_CHAIN_ARRAY) {
        static_for<U>([&](auto u) { y[u] = ya[0][u]; });
        static_for<NARR>([&](auto j) {
          if (j == ai) static_for<U>([&](auto u) { y[u] = ya[j][u]; });  // wave-uniform select
        });
        ai++;
      } else {
        uint32_t w = 0;
        static_for<AGPU_CHAIN_MAX_STEPS>([&](auto j) {
          if (j == s) w = sc[j];
        });
        const T v = __builtin_bit_cast(T, w);
        static_for<U>([&](auto u) {
#pragma unroll
          for (int k = 0; k < N; k++) y[u].v[k] = v;
        });
      }
    };
    for (int s = 0; s < n_steps; s++) {
      const int op = chain_op(code, s), kind = chain_kind(code, s);
      if (kind == AGPU_CHAIN_UNARY) {
        chain_apply_unary_packs<T, HEAVY, U, N>(op, acc);  // the op switch once per tile, not behind every element (round 5)
      } else {
        fetch(s, kind);
        chain_apply_binary_packs<T, U, N>(op, acc, y);
      }
    }
    if constexpr (CMP) {
      fetch(n_steps, chain_kind(code, n_steps));
      static_for<U>([&](auto u) {
        uint32_t m = 0;
#pragma unroll
        for (int k = 0; k < N; k++) m |= (uint32_t)chain_cmp_pred<T>(cmp_op, acc[u].v[k], y[u].v[k]) << k;
        uint32_t v = m << (4 * (threadIdx.x & 7));
        v |= (uint32_t)__shfl_xor((int)v, 1);
        v |= (uint32_t)__shfl_xor((int)v, 2);
        v |= (uint32_t)__shfl_xor((int)v, 4);
        if ((threadIdx.x & 7) == 0)
          reinterpret_cast<uint32_t*>(out)[(t * U + u) * (AGPU_EW_BLOCK / 8) + threadIdx.x / 8] = v;
      });
    } else {
      static_for<U>([&](auto u) { store_pack<true, T, N>(out + (pk0 + (uint64_t)u * AGPU_EW_BLOCK) * N, acc[u]); });
    }
  }
}

// one row of the chain, element-granular (tails, unaligned columns)
template <typename T>
__device__ __forceinline__ T chain_eval_from(T acc, uint64_t i, int n_steps, uint64_t code, const ChainPtrs& ptrs) {
  for (int s = 0; s < n_steps; s++) {
    const int op = chain_op(code, s), kind = chain_kind(code, s);
    if (kind == AGPU_CHAIN_UNARY) {
      acc = chain_apply_unary<T, true>(op, acc);
    } else {
      const T* q = nullptr;
      static_for<AGPU_CHAIN_MAX_STEPS>([&](auto j) {
        if (j == s) q = static_cast<const T*>(ptrs.p[j]);
      });
      acc = chain_apply_binary<T>(op, acc, kind == AGPU_CHAIN_ARRAY ? q[i] : q[0]);
    }
  }
  return acc;
}
template <typename T>
__device__ __forceinline__ T chain_eval_row(const T* in, uint64_t i, int n_steps, uint64_t code, const ChainPtrs& ptrs) {
  return chain_eval_from<T>(in[i], i, n_steps, code, ptrs);
}

// compare tail: one 32-bit output word per thread, words [first_word, n_words); bits past n are written as 0
template <typename T>
__global__ __launch_bounds__(AGPU_EW_BLOCK) void chain_cmp_tail_kernel(const T* in, uint32_t* out, uint64_t first_word,
                                                                      uint64_t n_words, uint64_t n, int n_steps,
                                                                      uint64_t code, ChainPtrs ptrs, int cmp_op) {
  for (uint64_t w = first_word + (uint64_t)blockIdx.x * AGPU_EW_BLOCK + threadIdx.x; w < n_words;
       w += (uint64_t)gridDim.x * AGPU_EW_BLOCK) {
    const T* q = nullptr;
    static_for<AGPU_CHAIN_MAX_STEPS>([&](auto j) {
      if (j == n_steps) q = static_cast<const T*>(ptrs.p[j]);
    });
    const bool arr = chain_kind(code, n_steps) == AGPU_CHAIN_ARRAY;
    uint32_t m = 0;
    for (uint32_t k = 0; k < 32 && w * 32 + k < n; k++) {
      const uint64_t i = w * 32 + k;
      const T acc = chain_eval_row<T>(in, i, n_steps, code, ptrs);
      m |= (uint32_t)chain_cmp_pred<T>(cmp_op, acc, arr ? q[i] : q[0]) << k;
    }
    out[w] = m;
  }
}

// rows [first, n): one element per lane, same arithmetic (also the whole column when a pointer is not 16-B aligned)
template <typename T>
__global__ __launch_bounds__(AGPU_EW_BLOCK) void chain_tail_kernel(const T* in, T* out, uint64_t first, uint64_t n,
                                                                  int n_steps, uint64_t code, ChainPtrs ptrs) {
  for (uint64_t i = first + (uint64_t)blockIdx.x * AGPU_EW_BLOCK + threadIdx.x; i < n;
       i += (uint64_t)gridDim.x * AGPU_EW_BLOCK) {
    out[i] = chain_eval_row<T>(in, i, n_steps, code, ptrs);
  }
}

template <typename T, bool HEAVY, int NARR, bool CMP>
static void launch_chain_full(agpu_pipeline* p, const T* pi, T* po, uint64_t ntiles, int n_steps, int n_arrs, uint64_t code,
                              const ChainPtrs& ptrs, const ChainPtrs& arrs, int cmp_op) {
  const int grid = stream_grid_for(p, ntiles);
  hipLaunchKernelGGL((chain_kernel<T, HEAVY, NARR, CMP>), dim3(grid), dim3(AGPU_EW_BLOCK), HEAVY ? wave_lds_for(p, AGPU_CHAIN_HEAVY_LDS, 1) : 0u, p->stream, pi, po, ntiles,
                     n_steps, n_arrs, code, ptrs, arrs, cmp_op,
                     // one input, one output, no transcendental step: two lock-step streams ((a + s)·t 0.794 → 0.832: common.hpp two_streams);
                     // (x·s).sin() gained 1.6 % in one process and lost 3.7 % in the next — the VALU-bound chains keep the sequential order
                     (NARR == 0 && !CMP && !HEAVY) ? two_streams_half(p, ntiles, (uint64_t)AGPU_EW_BLOCK * (HEAVY ? AGPU_CHAIN_HEAVY_U : 1) * 16) : (uint64_t)0);
}

// cmp_op < 0: plain chain, `out` is a T column.  cmp_op ≥ 0: slot n_steps of code / ptrs describes the compare's operand
// and `out` is the packed result bitmap (whole 64-bit words are written, bits past n as 0).
template <typename T>
static agpu_status launch_chain(agpu_pipeline* p, const void* in, void* out, uint64_t n, int n_steps, uint64_t code,
                                const ChainPtrs& ptrs, bool vec_ok, bool heavy, int cmp_op) {
  const T* pi = static_cast<const T*>(in);
  T* po = static_cast<T*>(out);
  const bool cmp = cmp_op >= 0;
  const int n_slots = n_steps + (cmp ? 1 : 0);  // step slots in use, including the compare's operand
  const uint64_t tile_rows = (uint64_t)AGPU_EW_BLOCK * 4 * (cmp ? AGPU_CHAIN_CMP_U : (heavy && std::is_floating_point<T>::value) ? AGPU_CHAIN_HEAVY_U : 1);
  const uint64_t ntiles = vec_ok ? n / tile_rows : 0;
  if (ntiles) {
    ChainPtrs arrs{}, scal;
    int n_arrs = 0;
    for (int s = 0; s < AGPU_CHAIN_MAX_STEPS; s++) {
      scal.p[s] = (s < n_slots && ptrs.p[s]) ? ptrs.p[s] : in;  // every slot readable: the kernel loads all eight
      if (s < n_slots && chain_kind(code, s) == AGPU_CHAIN_ARRAY) arrs.p[n_arrs++] = ptrs.p[s];
    }
    const int slots = n_arrs == 0 ? 0 : n_arrs <= 2 ? 2 : n_arrs <= 4 ? 4 : 8;
#define AGPU_CHAIN_CASE(H, A)                                                                                           \
  if (heavy == H && slots == A) {                                                                                       \
    if (cmp) launch_chain_full<T, H, A, true>(p, pi, po, ntiles, n_steps, n_arrs, code, scal, arrs, cmp_op);            \
    else launch_chain_full<T, H, A, false>(p, pi, po, ntiles, n_steps, n_arrs, code, scal, arrs, cmp_op);              \
  }
    if constexpr (std::is_floating_point<T>::value) {
      AGPU_CHAIN_CASE(true, 0) AGPU_CHAIN_CASE(true, 2) AGPU_CHAIN_CASE(true, 4) AGPU_CHAIN_CASE(true, 8)
    }
    AGPU_CHAIN_CASE(false, 0) AGPU_CHAIN_CASE(false, 2) AGPU_CHAIN_CASE(false, 4) AGPU_CHAIN_CASE(false, 8)
#undef AGPU_CHAIN_CASE
  }
  if (cmp) {
    const uint64_t first_word = ntiles * (tile_rows / 32), n_words = (n + 63) / 64 * 2;
    if (first_word < n_words) {
      const int grid = stream_grid_for(p, (n_words - first_word + AGPU_EW_BLOCK - 1) / AGPU_EW_BLOCK);
      hipLaunchKernelGGL((chain_cmp_tail_kernel<T>), dim3(grid), dim3(AGPU_EW_BLOCK), 0, p->stream, pi,
                         static_cast<uint32_t*>(out), first_word, n_words, n, n_steps, code, ptrs, cmp_op);
    }
  } else if (ntiles * tile_rows < n) {
    const uint64_t rest = n - ntiles * tile_rows;
    const int grid = stream_grid_for(p, (rest + AGPU_EW_BLOCK - 1) / AGPU_EW_BLOCK);
    hipLaunchKernelGGL((chain_tail_kernel<T>), dim3(grid), dim3(AGPU_EW_BLOCK), 0, p->stream, pi, po, ntiles * tile_rows, n,
                       n_steps, code, ptrs);
  }
  AGPU_LAUNCH_CHECK();
  return AGPU_OK;
}

// ---------------------------------------------------------------- chains with a WIDENING CAST at the head (agpu_fused_cast_chain)
// acc = (float)in[i] for an 8- / 16-bit integer column, then the f32 chain — `cast → sin`, `cast → mul_scalar → add_scalar` as the
// reference's callers write them [ref: crates/trigonometry/src/u8_kernel.rs:34-38 fuses exactly cast + trig; examples/simple.rs:45-72
// chains `*_op`s] — in ONE pass: the narrow column is read once (1–2 B/row), the f32 result written once, the 4 B/row intermediate
// the unfused pair writes and re-reads never exists.  The shape is cvt_wide_kernel's: one wave per 1 KiB chunk of the narrow column,
// 16-byte loads, the chunk transposed inside the wave (ds_bpermute) so that each of the R stores is a coalesced 1 KiB row; the chain
// runs on the f32x4 pack in between, with exactly chain_kernel's functors (bit-identical to the unfused sequence).  ARRAY operands
// are f32 columns read at the store position.
template <typename TI, bool HEAVY, int NARR>
__global__ __launch_bounds__(AGPU_CVTW_BLOCK) void cast_chain_kernel(const TI* in, float* out, uint64_t nchunks, int n_steps,
                                                                     int n_arrs, uint64_t code, ChainPtrs ptrs, ChainPtrs arrs, uint64_t half) {
  constexpr int R = 4 / sizeof(TI);  // 4 (8-bit) or 2 (16-bit) stores per load
  constexpr int NO = 4;
  constexpr uint32_t WAVES = AGPU_CVTW_BLOCK / AGPU_WAVE;
  const uint32_t lane = threadIdx.x & (AGPU_WAVE - 1);
  uint32_t sc[AGPU_CHAIN_MAX_STEPS];
  static_for<AGPU_CHAIN_MAX_STEPS>([&](auto s) { sc[s] = *(const __attribute__((address_space(4))) uint32_t*)(ptrs.p[s]); });
  const TileRun run = tile_run((uint64_t)blockIdx.x * WAVES + threadIdx.x / AGPU_WAVE, (uint64_t)gridDim.x * WAVES, nchunks);
  uint64_t c = run.t;
  if (c >= run.end) return;
  u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(in) + two_streams(c, half) * AGPU_WAVE + lane);
  for (;;) {
    const uint64_t cn = c + run.step;
    const bool more = cn < run.end;
    const uint64_t cp = two_streams(c, half);  // the chunk this round stores
    u32x4 vn = v;
    if (more) vn = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(in) + two_streams(cn, half) * AGPU_WAVE + lane);
    // all R packs of the chunk go through the chain TOGETHER (round 6): the step interpreter — op / kind extraction, the op switch — runs once
    // per chunk and lane (8 or 16 rows) instead of once per store (4 rows); its scalar instructions and branches were a third of what a
    // VALU-bound chain issued per row once sin / cos themselves had shrunk (cast(u16)·s → sin 0.62 of the roof)
    PackN<float, NO> acc[R];
    PackN<float, NO> ya[NARR > 0 ? NARR : 1][R];
    static_for<R>([&](auto j) {
      const uint64_t at = (cp * (uint64_t)(AGPU_WAVE * R) + (uint32_t)j * AGPU_WAVE + lane) * NO;  // first row of this lane's store
      static_for<NARR>([&](auto a) {
        if (a < n_arrs) ya[a][j] = load_pack<true, float, NO>(static_cast<const float*>(arrs.p[a]) + at);
      });
      const int src = (int)(((uint32_t)j * (AGPU_WAVE / R) + lane / R) * 4);
      const uint32_t w0 = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)v.x);
      const uint32_t w1 = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)v.y);
      const uint32_t w2 = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)v.z);
      const uint32_t w3 = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)v.w);
      PackN<TI, NO> x;
      if constexpr (R == 4) {
        const uint32_t sel = lane & 3u;
        const uint32_t w = sel == 0 ? w0 : sel == 1 ? w1 : sel == 2 ? w2 : w3;
        x = __builtin_bit_cast(PackN<TI, NO>, w);
      } else {
        const bool hi = (lane & 1u) != 0;
        const u32x2 w = {hi ? w2 : w0, hi ? w3 : w1};
        x = __builtin_bit_cast(PackN<TI, NO>, w);
      }
#pragma unroll
      for (int k = 0; k < NO; k++) acc[j].v[k] = (float)x.v[k];
    });
    int ai = 0;
    for (int s = 0; s < n_steps; s++) {
      const int op = chain_op(code, s), kind = chain_kind(code, s);
      if (kind == AGPU_CHAIN_UNARY) {
        chain_apply_unary_packs<float, HEAVY, R, NO>(op, acc);  // the op switch once per chunk, not per element (round 5) or per store (round 6)
      } else {
        PackN<float, NO> y[R];
        if (kind == AGPU_CHAIN_ARRAY) {
          static_for<R>([&](auto j) { y[j] = ya[0][j]; });
          static_for<NARR>([&](auto q) {
            if (q == ai) static_for<R>([&](auto j) { y[j] = ya[q][j]; });
          });
          ai++;
        } else {
          uint32_t w = 0;
          static_for<AGPU_CHAIN_MAX_STEPS>([&](auto q) {
            if (q == s) w = sc[q];
          });
          const float f = __builtin_bit_cast(float, w);
          static_for<R>([&](auto j) {
#pragma unroll
            for (int k = 0; k < NO; k++) y[j].v[k] = f;
          });
        }
        chain_apply_binary_packs<float, R, NO>(op, acc, y);
      }
    }
    static_for<R>([&](auto j) {
      const uint64_t at = (cp * (uint64_t)(AGPU_WAVE * R) + (uint32_t)j * AGPU_WAVE + lane) * NO;
      store_pack<(AGPU_STREAM_NT & 2) != 0, float, NO, (R == 2 || AGPU_CCHAIN_SC1X4)>(out + at, acc[j]);
    });
    if (!more) break;
    v = vn;
    c = cn;
  }
}
// rows [first, n), one per lane (tails, unaligned columns): the same arithmetic
template <typename TI>
__global__ __launch_bounds__(AGPU_EW_BLOCK) void cast_chain_tail_kernel(const TI* in, float* out, uint64_t first, uint64_t n,
                                                                       int n_steps, uint64_t code, ChainPtrs ptrs) {
  for (uint64_t i = first + (uint64_t)blockIdx.x * AGPU_EW_BLOCK + threadIdx.x; i < n; i += (uint64_t)gridDim.x * AGPU_EW_BLOCK)
    out[i] = chain_eval_from<float>((float)in[i], i, n_steps, code, ptrs);
}

// An 8-bit source has 256 distinct values: a chain of unary / scalar steps behind the cast is a FUNCTION of the byte.  One 256-thread
// block evaluates the chain once per byte value (the same functors, so the same bits as evaluating it per row) into a 1 KiB table in
// the pipeline's scratch, and the rows stream through lut8_kernel — `cast → · 0.37 → sin` runs at the lookup kernel's 0.82 of the HBM
// roof instead of the VALU-bound 0.43 of evaluating sin per row.
template <typename TI>
__global__ __launch_bounds__(256) void lut8_chain_build_kernel(float* tab, int n_steps, uint64_t code, ChainPtrs ptrs) {
  const uint32_t e = threadIdx.x;  // indexed by the raw byte, like lut8_build_kernel
  tab[e] = chain_eval_from<float>((float)(TI)(uint8_t)e, 0, n_steps, code, ptrs);
}
#define AGPU_LUT8_CHAIN_MIN_ROWS (1u << 16)

template <typename TI>
static agpu_status launch_cast_chain(agpu_pipeline* p, const void* in, float* out, uint64_t n, int n_steps, uint64_t code,
                                     const ChainPtrs& ptrs, bool vec_ok, bool heavy) {
  const TI* pi = static_cast<const TI*>(in);
  if constexpr (sizeof(TI) == 1) {
    bool arrays = false;
    for (int s = 0; s < n_steps; s++) arrays = arrays || chain_kind(code, s) == AGPU_CHAIN_ARRAY;
    constexpr uint64_t TILE_ROWS = (uint64_t)AGPU_LUT8_BLOCK * 4 * 4;
    if (!arrays && vec_ok && n >= AGPU_LUT8_CHAIN_MIN_ROWS) {
      void* tab = nullptr;
      const agpu_status st = agpu_scratch(p, 1024, &tab);
      if (st != AGPU_OK) return st;
      hipLaunchKernelGGL((lut8_chain_build_kernel<TI>), dim3(1), dim3(256), 0, p->stream, static_cast<float*>(tab), n_steps, code, ptrs);
      const uint64_t ntiles = n / TILE_ROWS;
      const uint64_t tk = tab_k(p);
      const int grid = stream_grid_for(p, tile_units(ntiles, tk));
      hipLaunchKernelGGL((lut8_kernel<TI>), dim3(grid), dim3(AGPU_LUT8_BLOCK), wave_lds_for(p, AGPU_WAVE_LDS_24, AGPU_LUT8_BLOCK / AGPU_WAVE), p->stream, pi, out, ntiles,
                         static_cast<const float*>(tab));
      if (ntiles * TILE_ROWS < n) {
        const uint64_t rest = n - ntiles * TILE_ROWS;
        const int g2 = stream_grid_for(p, (rest + AGPU_EW_BLOCK - 1) / AGPU_EW_BLOCK);
        hipLaunchKernelGGL((cast_chain_tail_kernel<TI>), dim3(g2), dim3(AGPU_EW_BLOCK), 0, p->stream, pi, out, ntiles * TILE_ROWS, n, n_steps,
                           code, ptrs);
      }
      AGPU_LAUNCH_CHECK();
      return AGPU_OK;
    }
  }
  constexpr uint64_t chunk_rows = (uint64_t)AGPU_WAVE * 16 / sizeof(TI);
  const uint64_t nchunks = vec_ok ? n / chunk_rows : 0;
  if (nchunks) {
    ChainPtrs arrs{}, scal;
    int n_arrs = 0;
    for (int s = 0; s < AGPU_CHAIN_MAX_STEPS; s++) {
      scal.p[s] = (s < n_steps && ptrs.p[s]) ? ptrs.p[s] : in;  // every slot readable: the kernel loads all eight
      if (s < n_steps && chain_kind(code, s) == AGPU_CHAIN_ARRAY) arrs.p[n_arrs++] = ptrs.p[s];
    }
    const int slots = n_arrs == 0 ? 0 : n_arrs <= 2 ? 2 : 4;
    // light chains: one chunk per wave (tile_run above); chains with a transcendental step are VALU-bound and gain in every run (0.53 → 0.61 at 8)
    // (round 6, the chain interpreted once per chunk: 4 chunks per wave 0.74–0.75 on cast(u16)·s → sin, 8 0.73, 1 0.71)
    const uint64_t kt = (uint64_t)(p->tune.tiles > 0 ? p->tune.tiles : (heavy ? 4 : 1));
    const uint64_t blocks = (nchunks + AGPU_CVTW_BLOCK / AGPU_WAVE - 1) / (AGPU_CVTW_BLOCK / AGPU_WAVE);
    const int grid = stream_grid_for(p, (blocks + kt - 1) / kt);
    // 16-bit sources without array operands: two lock-step streams (cast(u16)·s + s 0.824 → 0.839); 8-bit sources do not gain
    const uint64_t half2 = (sizeof(TI) == 2 && n_arrs == 0 && kt == 1) ? two_streams_half(p, nchunks, 2048) : 0;
#define AGPU_CCHAIN_CASE(H, A)                                                                                              \
  if (heavy == H && slots == A)                                                                                             \
    hipLaunchKernelGGL((cast_chain_kernel<TI, H, A>), dim3(grid), dim3(AGPU_CVTW_BLOCK), H ? wave_lds_for(p, AGPU_CHAIN_HEAVY_LDS, 1) : 0u, p->stream, pi, out, nchunks, n_steps, \
                       n_arrs, code, scal, arrs, half2);
    AGPU_CCHAIN_CASE(false, 0) AGPU_CCHAIN_CASE(false, 2) AGPU_CCHAIN_CASE(false, 4)
    AGPU_CCHAIN_CASE(true, 0) AGPU_CCHAIN_CASE(true, 2) AGPU_CCHAIN_CASE(true, 4)
#undef AGPU_CCHAIN_CASE
  }
  if (nchunks * chunk_rows < n) {
    const uint64_t rest = n - nchunks * chunk_rows;
    const int grid = stream_grid_for(p, (rest + AGPU_EW_BLOCK - 1) / AGPU_EW_BLOCK);
    hipLaunchKernelGGL((cast_chain_tail_kernel<TI>), dim3(grid), dim3(AGPU_EW_BLOCK), 0, p->stream, pi, out, nchunks * chunk_rows, n,
                       n_steps, code, ptrs);
  }
  AGPU_LAUNCH_CHECK();
  return AGPU_OK;
}

extern "C" {

// cmp_op < 0: agpu_fused_chain.  cmp_op ≥ 0: agpu_fused_chain_compare — the compare's operand rides in step slot n_steps.
static agpu_status chain_dispatch(agpu_pipeline* p, agpu_dtype dtype, const void* in, const agpu_chain_step* steps,
                                  int32_t n_steps, void* out, uint64_t n, int cmp_op, int cmp_kind, const void* cmp_operand) {
  AGPU_BIND_AS(p, cmp_op >= 0 ? "agpu_fused_chain_compare" : "agpu_fused_chain");
  const bool cmp = cmp_op >= 0;
  AGPU_REQUIRE(n_steps >= 0 && n_steps + (cmp ? 1 : 0) <= AGPU_CHAIN_MAX_STEPS, AGPU_ERR_ARG,
               "0..8 steps per chain (0..7 before a compare)");
  AGPU_REQUIRE(n == 0 || (in && out && (n_steps == 0 || steps)), AGPU_ERR_ARG, "null pointer");
  if (n == 0) return AGPU_OK;
  if (dtype == AGPU_DATE32) dtype = AGPU_I32;
  AGPU_REQUIRE(dtype == AGPU_F32 || dtype == AGPU_I32 || dtype == AGPU_U32, AGPU_ERR_UNSUPPORTED,
               "fused chains: f32, i32, u32 columns");
  ChainPtrs ptrs{};
  uint64_t code = 0;
  bool vec_ok = aligned16(in) && (cmp ? aligned_to(out, 8) : aligned16(out)), heavy = false;
  if (cmp) {
    AGPU_REQUIRE(cmp_op <= AGPU_CMP_EQ, AGPU_ERR_ARG, "bad compare op");
    AGPU_REQUIRE(cmp_kind == AGPU_CHAIN_SCALAR || cmp_kind == AGPU_CHAIN_ARRAY, AGPU_ERR_ARG, "bad compare operand kind");
    AGPU_REQUIRE(cmp_operand, AGPU_ERR_ARG, "null compare operand");
    AGPU_REQUIRE(aligned_to(out, 8), AGPU_ERR_SHAPE, "bitmaps must be 8-byte aligned");
    code |= ((uint64_t)(cmp_kind & 3) << 4) << (6 * n_steps);
    ptrs.p[n_steps] = cmp_operand;
    if (cmp_kind == AGPU_CHAIN_ARRAY && !aligned16(cmp_operand)) vec_ok = false;
  }
  for (int s = 0; s < n_steps; s++) {
    code |= ((uint64_t)(steps[s].op & 15) | ((uint64_t)(steps[s].kind & 3) << 4)) << (6 * s);
    ptrs.p[s] = steps[s].operand;
    const bool is_f = dtype == AGPU_F32;
    if (steps[s].kind == AGPU_CHAIN_UNARY) {
      const int o = steps[s].op;
      const bool ok = is_f ? (o == AGPU_UN_NEG || o == AGPU_UN_ABS || (o >= AGPU_UN_SQRT && o <= AGPU_UN_SINH))
                           : (o == AGPU_UN_NEG || o == AGPU_UN_ABS || o == AGPU_UN_NOT);
      AGPU_REQUIRE(ok, AGPU_ERR_UNSUPPORTED, "unary op not available in fused chains for this dtype");
      if (is_f && o >= AGPU_UN_CBRT) heavy = true;
    } else {
      AGPU_REQUIRE(steps[s].kind == AGPU_CHAIN_SCALAR || steps[s].kind == AGPU_CHAIN_ARRAY, AGPU_ERR_ARG, "bad step kind");
      AGPU_REQUIRE(steps[s].operand, AGPU_ERR_ARG, "null operand");
      const int o = steps[s].op;
      const bool ok = (o >= AGPU_OP_ADD && o <= AGPU_OP_MAX) || (!is_f && o >= AGPU_OP_AND && o <= AGPU_OP_XOR);
      AGPU_REQUIRE(ok, AGPU_ERR_UNSUPPORTED, "binary op not available in fused chains for this dtype");
      if (steps[s].kind == AGPU_CHAIN_ARRAY && !aligned16(steps[s].operand)) vec_ok = false;
    }
  }
  switch (dtype) {
    case AGPU_F32: return launch_chain<float>(p, in, out, n, n_steps, code, ptrs, vec_ok, heavy, cmp_op);
    case AGPU_I32: return launch_chain<int32_t>(p, in, out, n, n_steps, code, ptrs, vec_ok, false, cmp_op);
    default: return launch_chain<uint32_t>(p, in, out, n, n_steps, code, ptrs, vec_ok, false, cmp_op);
  }
}

agpu_status agpu_fused_chain(agpu_pipeline* p, agpu_dtype dtype, const void* in, const agpu_chain_step* steps,
                             int32_t n_steps, void* out, uint64_t n) {
  return chain_dispatch(p, dtype, in, steps, n_steps, out, n, -1, 0, nullptr);
}

agpu_status agpu_fused_chain_compare(agpu_pipeline* p, agpu_dtype dtype, const void* in, const agpu_chain_step* steps,
                                     int32_t n_steps, agpu_cmp_op cmp_op, int32_t operand_kind, const void* operand,
                                     void* out_bits, uint64_t n) {
  return chain_dispatch(p, dtype, in, steps, n_steps, out_bits, n, (int)cmp_op, operand_kind, operand);
}

agpu_status agpu_fused_cast_chain(agpu_pipeline* p, agpu_dtype from, const void* in, const agpu_chain_step* steps,
                                  int32_t n_steps, float* out, uint64_t n) {
  AGPU_REQUIRE(p, AGPU_ERR_ARG, "null pipeline");
  AGPU_REQUIRE(from == AGPU_U8 || from == AGPU_I8 || from == AGPU_U16 || from == AGPU_I16, AGPU_ERR_UNSUPPORTED,
               "fused cast chains start from u8 / i8 / u16 / i16 (the reference's casts to f32)");
  AGPU_REQUIRE(n_steps >= 0 && n_steps <= AGPU_CHAIN_MAX_STEPS, AGPU_ERR_ARG, "0..8 steps per chain");
  AGPU_REQUIRE(n_steps == 0 || steps, AGPU_ERR_ARG, "null steps");  // read below before n is looked at
  AGPU_REQUIRE(n == 0 || (in && out), AGPU_ERR_ARG, "null pointer");
  if (n_steps == 0) return agpu_cast(p, from, AGPU_F32, in, out, n);
  // cast → sin / cos / sinh of an 8-bit column IS the reference's fused kernel [trigonometry/src/u8_kernel.rs:34-38]: the 256-entry
  // table is built with the f32 kernels' own device functions, so the lookup is bit-identical to the unfused pair
  if (n_steps == 1 && steps[0].kind == AGPU_CHAIN_UNARY && (from == AGPU_U8 || from == AGPU_I8) &&
      (steps[0].op == AGPU_UN_SIN || steps[0].op == AGPU_UN_COS || steps[0].op == AGPU_UN_SINH))
    return agpu_unary(p, (agpu_unary_op)steps[0].op, from, in, out, n);
  AGPU_BIND(p);
  if (n == 0) return AGPU_OK;
  // cast → ONE transcendental step of a 16-bit column: the kernel SPECIALISED on the function (cvt_wide_kernel over CvtThenF32<TI, Op>: the
  // same functor the chain would call, so the same bits) instead of the generic chain kernel, whose step interpreter costs a VALU-bound
  // launch 10–15 % (round 5, tools/probe/chain_vs_special.py: cast i16 → sinh 1.22 → 1.05–1.11 ms at 1e9 rows)
  if (n_steps == 1 && steps[0].kind == AGPU_CHAIN_UNARY && (from == AGPU_U16 || from == AGPU_I16)) {
#define AGPU_CAST16_THEN(OPCODE, OP)                                                                                        \
  case OPCODE:                                                                                                              \
    return from == AGPU_U16 ? launch_cvt<uint16_t, float, CvtThenF32<uint16_t, OP>>(p, in, out, n)                          \
                            : launch_cvt<int16_t, float, CvtThenF32<int16_t, OP>>(p, in, out, n);
    switch (steps[0].op) {
      AGPU_CAST16_THEN(AGPU_UN_SIN, UnSin)
      AGPU_CAST16_THEN(AGPU_UN_COS, UnCos)
      AGPU_CAST16_THEN(AGPU_UN_SINH, UnSinh)
      AGPU_CAST16_THEN(AGPU_UN_EXP, UnExp)
      AGPU_CAST16_THEN(AGPU_UN_LOG, UnLog)
      default: break;
    }
#undef AGPU_CAST16_THEN
  }
  ChainPtrs ptrs{};
  uint64_t code = 0;
  bool vec_ok = aligned16(in) && aligned16(out), heavy = false;
  int n_arrays = 0;
  for (int s = 0; s < n_steps; s++) {
    code |= ((uint64_t)(steps[s].op & 15) | ((uint64_t)(steps[s].kind & 3) << 4)) << (6 * s);
    ptrs.p[s] = steps[s].operand;
    const int o = steps[s].op;
    if (steps[s].kind == AGPU_CHAIN_UNARY) {
      const bool ok = o == AGPU_UN_NEG || o == AGPU_UN_ABS || (o >= AGPU_UN_SQRT && o <= AGPU_UN_SINH);
      AGPU_REQUIRE(ok, AGPU_ERR_UNSUPPORTED, "unary op not available in fused chains for f32");
      if (o >= AGPU_UN_CBRT) heavy = true;
    } else {
      AGPU_REQUIRE(steps[s].kind == AGPU_CHAIN_SCALAR || steps[s].kind == AGPU_CHAIN_ARRAY, AGPU_ERR_ARG, "bad step kind");
      AGPU_REQUIRE(steps[s].operand, AGPU_ERR_ARG, "null operand");
      AGPU_REQUIRE(o >= AGPU_OP_ADD && o <= AGPU_OP_MAX, AGPU_ERR_UNSUPPORTED, "binary op not available in fused chains for f32");
      if (steps[s].kind == AGPU_CHAIN_ARRAY) {
        n_arrays++;
        if (!aligned16(steps[s].operand)) vec_ok = false;
      }
    }
  }
  AGPU_REQUIRE(n_arrays <= AGPU_CAST_CHAIN_MAX_ARRAYS, AGPU_ERR_UNSUPPORTED, "at most four array operands behind a cast head");
  switch (from) {
    case AGPU_U8: return launch_cast_chain<uint8_t>(p, in, out, n, n_steps, code, ptrs, vec_ok, heavy);
    case AGPU_I8: return launch_cast_chain<int8_t>(p, in, out, n, n_steps, code, ptrs, vec_ok, heavy);
    case AGPU_U16: return launch_cast_chain<uint16_t>(p, in, out, n, n_steps, code, ptrs, vec_ok, heavy);
    default: return launch_cast_chain<int16_t>(p, in, out, n, n_steps, code, ptrs, vec_ok, heavy);
  }
}

agpu_status agpu_binary(agpu_pipeline* p, agpu_binary_op op, agpu_dtype dtype, const void* a, const void* b, void* out,
                        uint64_t n) {
  return dispatch_binary<MODE_BINARY>(p, op, dtype, a, b, out, n);
}

agpu_status agpu_scalar(agpu_pipeline* p, agpu_binary_op op, agpu_dtype dtype, const void* a, const void* scalar,
                        void* out, uint64_t n) {
  return dispatch_binary<MODE_SCALAR>(p, op, dtype, a, scalar, out, n);
}

agpu_status agpu_unary(agpu_pipeline* p, agpu_unary_op op, agpu_dtype dtype, const void* in, void* out, uint64_t n) {
  AGPU_BIND(p);
  AGPU_REQUIRE(n == 0 || (in && out), AGPU_ERR_ARG, "null pointer");
#define UN_F32(OP) return launch_ew<float, OP, MODE_UNARY>(p, in, nullptr, out, n)
#define UN_INT(T)                                                               \
  switch (op) {                                                                 \
    case AGPU_UN_NEG: return launch_ew<T, UnNeg, MODE_UNARY>(p, in, nullptr, out, n); \
    case AGPU_UN_ABS: return launch_ew<T, UnAbs, MODE_UNARY>(p, in, nullptr, out, n); \
    case AGPU_UN_NOT: return launch_ew<T, UnNot, MODE_UNARY>(p, in, nullptr, out, n); \
    case AGPU_UN_POPCOUNT: return launch_ew<T, UnPopc, MODE_UNARY>(p, in, nullptr, out, n); \
    default: break;                                                             \
  }
#define UN_FUSED(T)                                                                          \
  switch (op) {                                                                              \
    case AGPU_UN_SIN: return launch_cvt<T, float, CvtThenF32<T, UnSin>>(p, in, out, n);      \
    case AGPU_UN_COS: return launch_cvt<T, float, CvtThenF32<T, UnCos>>(p, in, out, n);      \
    case AGPU_UN_SINH: return launch_cvt<T, float, CvtThenF32<T, UnSinh>>(p, in, out, n);    \
    default: break;                                                                          \
  }
  switch (dtype) {
    case AGPU_F32:
      switch (op) {
        case AGPU_UN_NEG: UN_F32(UnNeg);
        case AGPU_UN_ABS: UN_F32(UnAbs);
        case AGPU_UN_SQRT: UN_F32(UnSqrt);
        case AGPU_UN_CBRT: UN_F32(UnCbrt);
        case AGPU_UN_EXP: UN_F32(UnExp);
        case AGPU_UN_EXP2: UN_F32(UnExp2);
        case AGPU_UN_LOG: UN_F32(UnLog);
        case AGPU_UN_LOG2: UN_F32(UnLog2);
        case AGPU_UN_SIN: UN_F32(UnSin);
        case AGPU_UN_COS: UN_F32(UnCos);
        case AGPU_UN_ACOS: UN_F32(UnAcos);
        case AGPU_UN_SINH: UN_F32(UnSinh);
        default: break;
      }
      break;
    case AGPU_I32: case AGPU_DATE32: UN_INT(int32_t) break;
    case AGPU_U32: UN_INT(uint32_t) break;
    case AGPU_I16: UN_FUSED(int16_t) UN_INT(int16_t) break;
    case AGPU_U16: UN_FUSED(uint16_t) UN_INT(uint16_t) break;
#define UN_FUSED8(T)                                                         \
  switch (op) {                                                              \
    case AGPU_UN_SIN: return launch_lut8<T, UnSin>(p, in, out, n);           \
    case AGPU_UN_COS: return launch_lut8<T, UnCos>(p, in, out, n);           \
    case AGPU_UN_SINH: return launch_lut8<T, UnSinh>(p, in, out, n);         \
    default: break;                                                          \
  }
    case AGPU_I8: UN_FUSED8(int8_t) UN_INT(int8_t) break;
    case AGPU_U8: UN_FUSED8(uint8_t) UN_INT(uint8_t) break;
    default: break;
  }
#undef UN_F32
#undef UN_INT
#undef UN_FUSED
#undef UN_FUSED8
  agpu_set_error("unary op %d not supported for dtype %d", (int)op, (int)dtype);
  return AGPU_ERR_UNSUPPORTED;
}

agpu_status agpu_cast(agpu_pipeline* p, agpu_dtype from, agpu_dtype to, const void* in, void* out, uint64_t n) {
  AGPU_BIND(p);
  AGPU_REQUIRE(n == 0 || (in && out), AGPU_ERR_ARG, "null pointer");
  if (from == AGPU_DATE32) from = AGPU_I32;
  if (to == AGPU_DATE32) to = AGPU_I32;
  if (from == AGPU_BOOL && to == AGPU_F32) {
    if (n == 0) return AGPU_OK;
    AGPU_REQUIRE(aligned_to(in, 4) && aligned16(out), AGPU_ERR_SHAPE, "bool→f32 needs 4-byte aligned bits and 16-byte aligned output");
    const int grid = stream_grid_for(p, (n / 4 + AGPU_BLOCK * AGPU_B2F_U - 1) / (AGPU_BLOCK * AGPU_B2F_U));
    hipLaunchKernelGGL(bool_to_f32_kernel, dim3(grid), dim3(AGPU_BLOCK), 0, p->stream,
                       static_cast<const uint32_t*>(in), static_cast<float*>(out), n);
    AGPU_LAUNCH_CHECK();
    return AGPU_OK;
  }
  if (from == AGPU_F32 && to == AGPU_U8) return launch_cvt<float, uint8_t, CvtF32ToU8>(p, in, out, n);
  if (from == AGPU_F32) {  // reference-absent narrowing casts (see CvtF32ToInt)
    switch (to) {
      case AGPU_I8: return launch_cvt<float, int8_t, CvtF32ToInt<int8_t>>(p, in, out, n);
      case AGPU_I16: return launch_cvt<float, int16_t, CvtF32ToInt<int16_t>>(p, in, out, n);
      case AGPU_U16: return launch_cvt<float, uint16_t, CvtF32ToInt<uint16_t>>(p, in, out, n);
      case AGPU_I32: return launch_cvt<float, int32_t, CvtF32ToInt<int32_t>>(p, in, out, n);
      case AGPU_U32: return launch_cvt<float, uint32_t, CvtF32ToInt<uint32_t>>(p, in, out, n);
      default: break;
    }
  }
#define CAST_CASE(F, FT, T, TT) \
  if (from == F && to == T) return launch_cvt<FT, TT, CvtStatic<FT, TT>>(p, in, out, n)
  CAST_CASE(AGPU_I8, int8_t, AGPU_U16, uint16_t);
  CAST_CASE(AGPU_I8, int8_t, AGPU_U32, uint32_t);
  CAST_CASE(AGPU_I8, int8_t, AGPU_I16, int16_t);
  CAST_CASE(AGPU_I8, int8_t, AGPU_I32, int32_t);
  CAST_CASE(AGPU_I8, int8_t, AGPU_F32, float);
  CAST_CASE(AGPU_I16, int16_t, AGPU_I32, int32_t);
  CAST_CASE(AGPU_I16, int16_t, AGPU_U32, uint32_t);
  CAST_CASE(AGPU_I16, int16_t, AGPU_F32, float);
  CAST_CASE(AGPU_U8, uint8_t, AGPU_U16, uint16_t);
  CAST_CASE(AGPU_U8, uint8_t, AGPU_U32, uint32_t);
  CAST_CASE(AGPU_U8, uint8_t, AGPU_I16, int16_t);
  CAST_CASE(AGPU_U8, uint8_t, AGPU_I32, int32_t);
  CAST_CASE(AGPU_U8, uint8_t, AGPU_F32, float);
  CAST_CASE(AGPU_U16, uint16_t, AGPU_U32, uint32_t);
  CAST_CASE(AGPU_U16, uint16_t, AGPU_I32, int32_t);
  CAST_CASE(AGPU_U16, uint16_t, AGPU_F32, float);
#undef CAST_CASE
  // same-width sign reinterprets are a device copy in the reference [cast/src/lib.rs:69-86]
  if (from != AGPU_BOOL && to != AGPU_BOOL && from != AGPU_F32 && to != AGPU_F32 &&
      agpu_dtype_size(from) == agpu_dtype_size(to)) {
    if (in != out && n) AGPU_HIP(hipMemcpyAsync(out, in, n * agpu_dtype_size(from), hipMemcpyDeviceToDevice, p->stream));
    return AGPU_OK;
  }
  agpu_set_error("Casting not supported for type %d -> %d", (int)from, (int)to);
  return AGPU_ERR_UNSUPPORTED;
}

agpu_status agpu_broadcast(agpu_pipeline* p, agpu_dtype dtype, uint32_t value_bits, void* out, uint64_t n) {
  AGPU_BIND(p);
  if (n == 0) return AGPU_OK;
  AGPU_REQUIRE(out, AGPU_ERR_ARG, "null pointer");
  if (dtype == AGPU_BOOL) {
    AGPU_REQUIRE(aligned_to(out, 4), AGPU_ERR_SHAPE, "bitmap must be 4-byte aligned");
    const uint64_t n_words = agpu_bitmap_bytes(n) / 4;
    const int grid = stream_grid_for(p, (n_words + AGPU_BLOCK - 1) / AGPU_BLOCK);
    hipLaunchKernelGGL(fill_bits_kernel, dim3(grid), dim3(AGPU_BLOCK), 0, p->stream, static_cast<uint32_t*>(out),
                       value_bits ? 1u : 0u, n, n_words);
    AGPU_LAUNCH_CHECK();
    return AGPU_OK;
  }
  const size_t w = agpu_dtype_size(dtype);
  AGPU_REQUIRE(w != 0, AGPU_ERR_UNSUPPORTED, "unsupported dtype");
  AGPU_REQUIRE(aligned16(out), AGPU_ERR_SHAPE, "broadcast output must be 16-byte aligned");
  uint32_t pattern = value_bits;
  if (w == 1) pattern = (value_bits & 0xFFu) * 0x01010101u;
  else if (w == 2) pattern = (value_bits & 0xFFFFu) * 0x00010001u;
  const uint64_t bytes = n * w;
  const int grid = stream_grid_for(p, (bytes / 16 + AGPU_BLOCK - 1) / AGPU_BLOCK);
  hipLaunchKernelGGL(fill_kernel, dim3(grid), dim3(AGPU_BLOCK), 0, p->stream, static_cast<uint8_t*>(out), pattern, bytes);
  AGPU_LAUNCH_CHECK();
  return AGPU_OK;
}

agpu_status agpu_broadcast_from_device(agpu_pipeline* p, agpu_dtype dtype, const void* scalar_dev, void* out,
                                       uint64_t n) {
  AGPU_BIND(p);
  if (n == 0) return AGPU_OK;
  AGPU_REQUIRE(out && scalar_dev, AGPU_ERR_ARG, "null pointer");
  const size_t w = agpu_dtype_size(dtype);
  AGPU_REQUIRE(w != 0, AGPU_ERR_UNSUPPORTED, "unsupported dtype");
  AGPU_REQUIRE(aligned16(out), AGPU_ERR_SHAPE, "broadcast output must be 16-byte aligned");
  const uint64_t bytes = n * w;
  const int grid = stream_grid_for(p, (bytes / 16 + AGPU_BLOCK - 1) / AGPU_BLOCK);
  uint8_t* po = static_cast<uint8_t*>(out);
  if (w == 4)
    hipLaunchKernelGGL((fill_from_device_kernel<uint32_t>), dim3(grid), dim3(AGPU_BLOCK), 0, p->stream, po,
                       static_cast<const uint32_t*>(scalar_dev), bytes);
  else if (w == 2)
    hipLaunchKernelGGL((fill_from_device_kernel<uint16_t>), dim3(grid), dim3(AGPU_BLOCK), 0, p->stream, po,
                       static_cast<const uint16_t*>(scalar_dev), bytes);
  else
    hipLaunchKernelGGL((fill_from_device_kernel<uint8_t>), dim3(grid), dim3(AGPU_BLOCK), 0, p->stream, po,
                       static_cast<const uint8_t*>(scalar_dev), bytes);
  AGPU_LAUNCH_CHECK();
  return AGPU_OK;
}

}  // extern "C"
