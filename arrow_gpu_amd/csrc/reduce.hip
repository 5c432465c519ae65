// reduce.hip — whole-column reductions: sum (reference order for f32), min, max, f64-accumulated sum.
//
// Replaces crates/arithmetic/compute_shaders/{f32,i32,u32}/aggregate.wgsl ("sum": 256-wide LDS tree with a barrier
// per step, one dispatch per level) driven by Sum::sum_op (crates/arithmetic/src/aggregate_kernels.rs:24-51).
// min/max reductions do not exist in the reference (its MinMax is element-wise); they are north_star config 5 and
// follow Arrow min_max semantics.
//
// f32 SUM keeps the reference's summation ORDER so the result is bit-identical to its tree: an adjacent-pair binary
// tree inside each aligned 256-row block (missing rows count as +0.0), then the same tree over the block sums, level
// by level.  That tree is associative-free but it IS a perfect binary tree over aligned power-of-two spans, so it
// maps onto the machine without LDS barriers:
//   rows 4l..4l+3 of a 256-row block : one 16-byte load per lane, (x0+x1)+(x2+x3) in registers      (levels 1-2)
//   64 lanes                         : shift-down shuffle adds, offsets 1,2,4..32                     (levels 3-8)
//   64 consecutive blocks per wave   : block sums parked in lanes 0..63, same 6 shuffle steps          (levels 9-14)
// One WAVE (a one-wave block: no LDS, no barrier) retires an aligned 16,384-row quarter span and writes one partial.  A
// second, tiny launch — one wave per 256 spans — adds the four quarters of every span, (q0+q1)+(q2+q3) (levels 15-16,
// which completes the reference's second 256-ary level), and runs the third 256-ary level over the span sums in
// registers + 6 shuffle steps; whatever is left (≤ 256 values up to 4.3e9 rows) goes through a single-workgroup copy of
// the reference's own LDS tree.  HBM-bound: 4 B/row, every byte read once.  (tools/probe/sum_probe.py, 1e9 rows, one
// process: a 256-thread block per span 0.838 of the HBM roof, one wave per quarter span 0.861 — the access pattern's
// ceiling, 0.858 with the cross-lane work removed; and the old finish — ONE workgroup walking 15 259 span sums through
// 60 serial LDS trees — cost another 41 µs of a 0.64 ms launch.)
#include <type_traits>

#include "common.hpp"

#define SPAN_ROWS 65536u   // rows per workgroup span (256 * 256)
#define WAVE_ROWS 16384u   // rows per wave inside a span (64 blocks of 256)
#define AGPU_FOLD_BLOCK_SUM 1024

__device__ __forceinline__ float wave_tree_sum(float v) {  // lane 0 gets the adjacent-pair tree sum of the 64 lanes
#pragma unroll
  for (int off = 1; off < AGPU_WAVE; off <<= 1) v = v + __shfl_down(v, off);
  return v;
}

__device__ __forceinline__ uint32_t validity_nibble(const uint8_t* validity, uint64_t row) {  // rows row..row+3, row%4==0
  return ((uint32_t)validity[row >> 3] >> (row & 4)) & 0xFu;
}

template <bool GUARD, bool HASV>
__device__ __forceinline__ float load4_tree(const float* in, const uint8_t* validity, uint64_t row, uint64_t n) {
  float x0, x1, x2, x3;
  if constexpr (GUARD) {
    x0 = row + 0 < n ? in[row + 0] : 0.0f;
    x1 = row + 1 < n ? in[row + 1] : 0.0f;
    x2 = row + 2 < n ? in[row + 2] : 0.0f;
    x3 = row + 3 < n ? in[row + 3] : 0.0f;
    if constexpr (HASV) {
      if (row + 0 < n && !((validity[(row + 0) >> 3] >> ((row + 0) & 7)) & 1)) x0 = 0.0f;
      if (row + 1 < n && !((validity[(row + 1) >> 3] >> ((row + 1) & 7)) & 1)) x1 = 0.0f;
      if (row + 2 < n && !((validity[(row + 2) >> 3] >> ((row + 2) & 7)) & 1)) x2 = 0.0f;
      if (row + 3 < n && !((validity[(row + 3) >> 3] >> ((row + 3) & 7)) & 1)) x3 = 0.0f;
    }
  } else {
    const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(in + row));  // read-once stream
    x0 = v.x; x1 = v.y; x2 = v.z; x3 = v.w;
    if constexpr (HASV) {
      const uint32_t nib = validity_nibble(validity, row);
      if (!(nib & 1)) x0 = 0.0f;
      if (!(nib & 2)) x1 = 0.0f;
      if (!(nib & 4)) x2 = 0.0f;
      if (!(nib & 8)) x3 = 0.0f;
    }
  }
  return (x0 + x1) + (x2 + x3);
}

// Reduce EIGHT 256-row blocks at once ("transpose-reduce"): at butterfly step k the two registers of a pair are
// merged so that lanes with bit k clear keep the even block and lanes with bit k set keep the odd one.  Every add
// still pairs lane l with lane l^2^k in the order 1, 2, 4, … — the reference's adjacent-pair tree (f32 addition is
// commutative, so which side is "left" does not change the bits) — but 8 blocks cost 10 shuffles instead of 48.
// In: s[u] = this lane's 4-row partial of block u.  Out (every lane): the sum of block (lane & 7).
__device__ __forceinline__ float transpose_reduce8(const float (&s)[8], uint32_t lane) {
  float t4[4], t2[2];
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const bool odd = lane & 1;
    const float keep = odd ? s[2 * j + 1] : s[2 * j];
    const float send = odd ? s[2 * j] : s[2 * j + 1];
    t4[j] = keep + __shfl_xor(send, 1);
  }
#pragma unroll
  for (int j = 0; j < 2; j++) {
    const bool odd = lane & 2;
    const float keep = odd ? t4[2 * j + 1] : t4[2 * j];
    const float send = odd ? t4[2 * j] : t4[2 * j + 1];
    t2[j] = keep + __shfl_xor(send, 2);
  }
  float v;
  {
    const bool odd = lane & 4;
    const float keep = odd ? t2[1] : t2[0];
    const float send = odd ? t2[0] : t2[1];
    v = keep + __shfl_xor(send, 4);
  }
  v = v + __shfl_xor(v, 8);
  v = v + __shfl_xor(v, 16);
  v = v + __shfl_xor(v, 32);
  return v;
}

// One wave, one aligned 16,384-row quarter span (64 blocks of 256 rows) → its tree sum in lane 0.
// GUARD: quarters that cross the end of the column and columns that are only 4-byte aligned — same tree, guarded scalar loads.
template <bool GUARD, bool HASV>
__device__ __forceinline__ float quarter_tree_sum(const float* in, const uint8_t* validity, uint64_t wave_base, uint64_t n) {
  constexpr int UNR = 8;
  const uint32_t lane = threadIdx.x & (AGPU_WAVE - 1);
  float acc = 0.0f;  // lane j ends up holding the sum of this wave's j-th 256-row block
  for (int j0 = 0; j0 < AGPU_WAVE; j0 += UNR) {
    float s[UNR];
    if constexpr (HASV && !GUARD) {
      // validity of the 8 blocks of this step = 2048 bits = one 4-byte load per lane (256 contiguous bytes per wave);
      // lane l's nibble for block u sits in the word lane 8u + l/8 holds — one cross-lane read instead of a byte load per
      // block and lane
      const uint32_t vword = reinterpret_cast<const uint32_t*>(validity)[(wave_base + (uint64_t)j0 * 256) / 32 + lane];
#pragma unroll
      for (int u = 0; u < UNR; u++) {
        const uint64_t row = wave_base + (uint64_t)(j0 + u) * 256 + lane * 4;
        const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(in + row));
        const uint32_t nib = ((uint32_t)__shfl((int)vword, u * 8 + (int)(lane >> 3)) >> ((lane & 7u) * 4)) & 0xFu;
        const float x0 = (nib & 1) ? v.x : 0.0f, x1 = (nib & 2) ? v.y : 0.0f, x2 = (nib & 4) ? v.z : 0.0f, x3 = (nib & 8) ? v.w : 0.0f;
        s[u] = (x0 + x1) + (x2 + x3);
      }
    } else {
#pragma unroll
      for (int u = 0; u < UNR; u++)
        s[u] = load4_tree<GUARD, HASV>(in, validity, wave_base + (uint64_t)(j0 + u) * 256 + lane * 4, n);
    }
    const float v = transpose_reduce8(s, lane);
    // every lane now holds the sum of block j0 + (lane & 7); park it in lane j0 + (lane & 7)
    if ((lane >> 3) == (uint32_t)(j0 >> 3)) acc = v;
  }
  return wave_tree_sum(acc);
}

// first launch: quarters[q] for q < 4 * nspans (quarters that lie entirely beyond the column come out +0.0: the
// reference's zero padding of the last workgroup)
template <bool HASV>
__global__ __launch_bounds__(AGPU_WAVE) void sum_tree_quarter_kernel(const float* in, const uint8_t* validity, uint64_t n,
                                                                    float* quarters, uint64_t nquarters, int vec_ok, uint64_t q0) {
  for (uint64_t q = q0 + blockIdx.x; q < nquarters; q += gridDim.x) {
    const uint64_t base = q * WAVE_ROWS;
    float r;
    if (vec_ok && base + WAVE_ROWS <= n) r = quarter_tree_sum<false, HASV>(in, validity, base, n);
    else r = quarter_tree_sum<true, HASV>(in, validity, base, n);
    if (threadIdx.x == 0) quarters[q] = r;
  }
}

// second launch: one wave per 256 spans.  Lane l owns spans 256 g + 4 l … + 3: sixteen consecutive quarter sums (four
// 16-byte loads) → four span sums (q0+q1)+(q2+q3) → (s0+s1)+(s2+s3) → 6 shuffle steps = the reference's third 256-ary
// level over zero-padded span sums.  A column of ONE span has no third level: its span sum is the result.
__global__ __launch_bounds__(AGPU_WAVE) void sum_tree_combine_kernel(const float* quarters, uint64_t nspans, float* out) {
  const uint32_t lane = threadIdx.x;
  const uint64_t sp0 = ((uint64_t)blockIdx.x * AGPU_WAVE + lane) * 4;
  float s[4];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    if (sp0 + (uint64_t)k < nspans) {
      const f32x4 w = *reinterpret_cast<const f32x4*>(quarters + (sp0 + (uint64_t)k) * 4);
      s[k] = (w.x + w.y) + (w.z + w.w);
    } else {
      s[k] = 0.0f;
    }
  }
  if (nspans == 1) {
    if (lane == 0) out[0] = s[0];
    return;
  }
  const float t = wave_tree_sum((s[0] + s[1]) + (s[2] + s[3]));
  if (lane == 0) out[blockIdx.x] = t;
}

// second AND last launch for 257 … 65 536 spans (16.8 M … 4.29e9 rows): one 1024-thread workgroup does what the combine launch's ≤ 256
// one-wave blocks did — wave w takes the groups w, w + 16, …, four groups' sixteen 16-byte loads per lane in flight together — parks the
// group sums in LDS (zero-padded to 256, the reference's padding) and its first wave runs the reference's next 256-ary level over them:
// (e0+e1)+(e2+e3) per lane + 6 shuffle steps, the same adjacent-pair tree sum_tree_finish_kernel walks through LDS.  One launch boundary
// and ≈ 5 µs less behind a 0.57 ms read.
__device__ __forceinline__ void sum_combine_finish_body(const float* quarters, uint64_t nspans, uint32_t ngroups, float* out, float* sh /* [256] */) {
  const uint32_t lane = threadIdx.x & (AGPU_WAVE - 1), wave = threadIdx.x / AGPU_WAVE;
  constexpr uint32_t NW = AGPU_FOLD_BLOCK_SUM / AGPU_WAVE;
  if (threadIdx.x < 256) sh[threadIdx.x] = 0.0f;
  __syncthreads();
  for (uint32_t g0 = wave; g0 < ngroups; g0 += 4 * NW) {
    f32x4 w[4][4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const uint64_t sp0 = ((uint64_t)(g0 + r * NW) * AGPU_WAVE + lane) * 4;
#pragma unroll
      for (int k = 0; k < 4; k++)
        if (g0 + r * NW < ngroups && sp0 + (uint64_t)k < nspans) w[r][k] = *reinterpret_cast<const f32x4*>(quarters + (sp0 + (uint64_t)k) * 4);
    }
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const uint32_t g = g0 + r * NW;
      if (g >= ngroups) break;  // wave-uniform
      const uint64_t sp0 = ((uint64_t)g * AGPU_WAVE + lane) * 4;
      float s[4];
#pragma unroll
      for (int k = 0; k < 4; k++) s[k] = sp0 + (uint64_t)k < nspans ? (w[r][k].x + w[r][k].y) + (w[r][k].z + w[r][k].w) : 0.0f;
      const float t = wave_tree_sum((s[0] + s[1]) + (s[2] + s[3]));
      if (lane == 0) sh[g] = t;
    }
  }
  __syncthreads();
  if (wave == 0) {
    const float t = wave_tree_sum((sh[4 * lane] + sh[4 * lane + 1]) + (sh[4 * lane + 2] + sh[4 * lane + 3]));
    if (lane == 0) out[0] = t;
  }
}
__global__ __launch_bounds__(AGPU_FOLD_BLOCK_SUM) void sum_tree_combine_finish_kernel(const float* quarters, uint64_t nspans, uint32_t ngroups,
                                                                                   float* out) {
  __shared__ float sh[256];
  sum_combine_finish_body(quarters, nspans, ngroups, out, sh);
}

// remaining 256-ary levels over m values, exactly the reference's workgroup tree (aggregate.wgsl:21-41); one workgroup
__global__ __launch_bounds__(AGPU_BLOCK) void sum_tree_finish_kernel(const float* first, const uint8_t* validity,
                                                                    float* buf0, float* buf1, uint64_t m, float* out,
                                                                    int at_least_one) {
  __shared__ float sh[AGPU_BLOCK];
  const uint32_t tid = threadIdx.x;
  const float* src = first;
  float* dst = buf0;
  int levels = 0;
  while (m > 1 || (at_least_one && levels == 0)) {
    const uint64_t groups = (m + 255) / 256;
    for (uint64_t g = 0; g < groups; g++) {
      const uint64_t i = g * 256 + tid;
      float x = i < m ? src[i] : 0.0f;
      if (levels == 0 && validity && i < m && !((validity[i >> 3] >> (i & 7)) & 1)) x = 0.0f;
      sh[tid] = x;
      __syncthreads();
      for (uint32_t s = 1; s < 256; s *= 2) {
        const uint32_t idx = 2 * s * tid;
        if (idx + s < 256) sh[idx] = sh[idx] + sh[idx + s];
        __syncthreads();
      }
      if (tid == 0) dst[g] = sh[0];
      __syncthreads();
    }
    src = dst;
    dst = (dst == buf0) ? buf1 : buf0;
    m = groups;
    levels++;
    __threadfence_block();
    __syncthreads();
  }
  if (tid == 0) out[0] = (m == 0) ? 0.0f : src[0];
}

// ---------------------------------------------------------------- generic two-stage reductions (order-free results)
struct MinMaxF32 {
  float r;
  uint32_t flags;  // bit0: saw a non-NaN value, bit1: saw a NaN
};

template <typename T> struct RedSumWrap {  // wrapping integer sum [aggregate.wgsl i32/u32]
  typedef uint32_t Acc;
  __device__ static Acc identity() { return 0u; }
  __device__ static Acc load(T x) { return (uint32_t)x; }
  __device__ static Acc combine(Acc a, Acc b) { return a + b; }
  __device__ static T finish(Acc a) { return (T)a; }
  typedef T Out;
  typedef Acc Part;
  __device__ static Part to_part(Acc a) { return a; }
  __device__ static Acc from_part(Part q) { return q; }
};
struct RedSumF64 {
  typedef double Acc;
  typedef double Out;
  __device__ static Acc identity() { return 0.0; }
  __device__ static Acc load(float x) { return (double)x; }
  __device__ static Acc combine(Acc a, Acc b) { return a + b; }
  __device__ static Out finish(Acc a) { return a; }
  typedef Acc Part;
  __device__ static Part to_part(Acc a) { return a; }
  __device__ static Acc from_part(Part q) { return q; }
};
template <typename T, bool MAX> struct RedMinMaxInt {
  typedef T Acc;
  typedef T Out;
  __device__ static Acc identity() {
    if constexpr (std::is_signed<T>::value) return MAX ? INT32_MIN : INT32_MAX;
    else return MAX ? 0u : 0xFFFFFFFFu;
  }
  __device__ static Acc load(T x) { return x; }
  __device__ static Acc combine(Acc a, Acc b) { return MAX ? (a > b ? a : b) : (a < b ? a : b); }
  __device__ static Out finish(Acc a) { return a; }
  typedef Acc Part;
  __device__ static Part to_part(Acc a) { return a; }
  __device__ static Acc from_part(Part q) { return q; }
};
template <bool MAX> struct RedMinMaxF32 {  // Arrow min_max: NaN skipped unless every value is NaN; -0 < +0
  typedef MinMaxF32 Acc;
  typedef float Out;
  __device__ static Acc identity() { return MinMaxF32{MAX ? -__builtin_inff() : __builtin_inff(), 0u}; }
  __device__ static Acc load(float x) {
    if (x != x) return MinMaxF32{MAX ? -__builtin_inff() : __builtin_inff(), 2u};
    return MinMaxF32{x, 1u};
  }
  // neither operand is ever NaN here (load() maps NaN to the identity), so the hardware's v_min_f32 / v_max_f32 — which
  // order −0 below +0 — give Arrow's result in one instruction instead of two compares, a sign test and two selects
  // (the reduction reads 4 B/row: at 3 rows per clock and CU those ten operations were half of the VALU budget)
  // (written as the instruction itself: through fminf / fmaxf LLVM adds a canonicalising v_max_f32 x, x per operand)
  __device__ static float pick(float a, float b) {
#if defined(__HIP_DEVICE_COMPILE__)
    float r;
    if constexpr (MAX) asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    else asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
#else
    return MAX ? __builtin_fmaxf(a, b) : __builtin_fminf(a, b);
#endif
  }
  __device__ static Acc combine(Acc a, Acc b) { return MinMaxF32{pick(a.r, b.r), a.flags | b.flags}; }
  __device__ static Out finish(Acc a) {
    if (!(a.flags & 1u) && (a.flags & 2u)) return __builtin_nanf("");
    return a.r;
  }
  // what a NON-EMPTY chunk leaves for the fold: its finished value — NaN exactly when every row of the chunk was NaN, which load() turns
  // back into {identity, saw a NaN}.  Four bytes per chunk instead of eight: the folding workgroup reads 244 KB at 1e9 rows in ONE round
  // of loads (11.7 → 4.6 µs)
  typedef float Part;
  __device__ static Part to_part(Acc a) { return finish(a); }
  __device__ static Acc from_part(Part q) { return load(q); }
};

// … with a validity bitmap a chunk may hold NO valid row at all, which its partial must keep apart from "every valid row was NaN": the
// null-aware launches fold the accumulator itself ({value, flags}, 8 bytes per chunk)
template <bool MAX> struct RedMinMaxF32V : RedMinMaxF32<MAX> {
  typedef MinMaxF32 Part;
  __device__ static Part to_part(MinMaxF32 a) { return a; }
  __device__ static MinMaxF32 from_part(Part q) { return q; }
};
// the reduction a column WITH validity takes through the wave kernel (same results; only RedMinMaxF32's partial differs)
template <typename Red> struct NullAwareRed { typedef Red type; };
template <bool MAX> struct NullAwareRed<RedMinMaxF32<MAX>> { typedef RedMinMaxF32V<MAX> type; };

template <typename A> __device__ __forceinline__ A shfl_down_acc(A v, int off) {
  if constexpr (std::is_same<A, MinMaxF32>::value) return MinMaxF32{__shfl_down(v.r, off), (uint32_t)__shfl_down((int)v.flags, off)};
  else if constexpr (std::is_same<A, uint32_t>::value) return (uint32_t)__shfl_down((int)v, off);
  else return __shfl_down(v, off);
}

template <typename Red, typename A>
__device__ __forceinline__ A block_reduce(A v, A* lds) {
#pragma unroll
  for (int off = AGPU_WAVE / 2; off > 0; off >>= 1) v = Red::combine(v, shfl_down_acc(v, off));
  const uint32_t lane = threadIdx.x & (AGPU_WAVE - 1), wave = threadIdx.x / AGPU_WAVE;
  __syncthreads();
  if (lane == 0) lds[wave] = v;
  __syncthreads();
  A r = lds[0];
  for (int k = 1; k < AGPU_BLOCK / AGPU_WAVE; k++) r = Red::combine(r, lds[k]);
  return r;
}

template <typename T>
__device__ __forceinline__ T word_as(uint32_t w) {
  return __builtin_bit_cast(T, w);
}

template <typename T, typename Red, int U>
__global__ __launch_bounds__(AGPU_BLOCK) void reduce_partial_kernel(const T* in, const uint8_t* validity, uint64_t n,
                                                                   typename Red::Acc* partials, int vec_ok) {
  typedef typename Red::Acc A;
  __shared__ A lds[AGPU_BLOCK / AGPU_WAVE];
  A acc = Red::identity();
  const uint64_t npacks = vec_ok ? n / 4 : 0;
  const uint64_t tile = (uint64_t)AGPU_BLOCK * U;
  const uint64_t ntiles = npacks / tile;
  for (uint64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const uint64_t p0 = t * tile + threadIdx.x;
    u32x4 v[U];
    uint32_t nib[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const uint64_t pk = p0 + (uint64_t)u * AGPU_BLOCK;
      v[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(in + pk * 4));  // read-once stream
      nib[u] = validity ? validity_nibble(validity, pk * 4) : 0xFu;
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      // NB: copy the lanes out first — __builtin_bit_cast applied directly to a vector-element expression
      // (v[u].y) reads the vector's first element on this compiler.
      const uint32_t e0 = v[u].x, e1 = v[u].y, e2 = v[u].z, e3 = v[u].w;
      if (nib[u] & 1) acc = Red::combine(acc, Red::load(word_as<T>(e0)));
      if (nib[u] & 2) acc = Red::combine(acc, Red::load(word_as<T>(e1)));
      if (nib[u] & 4) acc = Red::combine(acc, Red::load(word_as<T>(e2)));
      if (nib[u] & 8) acc = Red::combine(acc, Red::load(word_as<T>(e3)));
    }
  }
  // rows past the last full tile: element-granular, spread over the whole grid
  for (uint64_t i = ntiles * tile * 4 + (uint64_t)blockIdx.x * AGPU_BLOCK + threadIdx.x; i < n;
       i += (uint64_t)gridDim.x * AGPU_BLOCK) {
    if (!validity || ((validity[i >> 3] >> (i & 7)) & 1)) acc = Red::combine(acc, Red::load(in[i]));
  }
  const A r = block_reduce<Red, A>(acc, lds);
  if (threadIdx.x == 0) partials[blockIdx.x] = r;
}

template <typename Red>
__global__ __launch_bounds__(AGPU_BLOCK) void reduce_finish_kernel(const typename Red::Acc* partials, uint32_t m,
                                                                  typename Red::Out* out) {
  typedef typename Red::Acc A;
  __shared__ A lds[AGPU_BLOCK / AGPU_WAVE];
  A acc = Red::identity();
  for (uint32_t i = threadIdx.x; i < m; i += AGPU_BLOCK) acc = Red::combine(acc, partials[i]);
  const A r = block_reduce<Red, A>(acc, lds);
  if (threadIdx.x == 0) out[0] = Red::finish(r);
}

// blocks_per_cu: 16 for the partial reductions (read-only grid-stride loops with 4 packs per lane in flight: 6.47 / 6.65 /
// 6.70 TB/s for min / f64 sum / i32 sum vs 6.31 / 6.45 / 6.58 at 64 per CU — tools/probe/reduce_sweep.py, one process,
// same buffer).  The tree sum launches one one-wave block per quarter span (sum_probe.py); the order-free reductions
// of columns without validity take that shape too (reduce_wave_kernel: min / max 0.826 → 0.840, f64 sum 0.830 → 0.847 in
// bench.py's 1e9-row shard) and keep this grid for everything else.
static int reduce_grid_for(const agpu_pipeline* p, uint64_t work_blocks, int blocks_per_cu) {
  int64_t g = (int64_t)p->dev->num_cus * blocks_per_cu;
  if ((uint64_t)g > work_blocks) g = (int64_t)work_blocks;
  if (g < 1) g = 1;
  return (int)g;
}

// Order-free reductions of a column WITHOUT validity in the tree sum's launch shape: one one-wave block per 16 384-row
// chunk (64 KiB, eight 16-byte loads per lane in flight), its result to partials[chunk]; a small second level folds the
// partials.  The grid-stride form above stays for columns with validity, unaligned columns and the < 1-chunk tail.
#define AGPU_REDUCE_WAVE 1
constexpr uint64_t RED_CHUNK_ROWS = 16384;
// HASV (round 6b): null-aware — the validity bits of an iteration's 8 × 256 rows are ONE 4-byte load per lane (256 contiguous bytes per wave);
// lane l's nibble for vector u sits in the word lane 8 u + l / 8 holds (quarter_tree_sum's scheme); invalid rows are skipped.
template <typename T, typename Red, bool HASV = false>
__global__ __launch_bounds__(AGPU_WAVE) void reduce_wave_kernel(const T* in, const uint8_t* validity, typename Red::Part* partials, uint64_t nchunks) {
  typedef typename Red::Acc A;
  const uint32_t lane = threadIdx.x;
  for (uint64_t c = blockIdx.x; c < nchunks; c += gridDim.x) {
    const u32x4* base = reinterpret_cast<const u32x4*>(in + c * RED_CHUNK_ROWS) + lane;
    A a0 = Red::identity(), a1 = Red::identity(), a2 = Red::identity(), a3 = Red::identity();
    for (int j0 = 0; j0 < 64; j0 += 8) {
      u32x4 v[8];
      uint32_t vword = 0;
      if constexpr (HASV) vword = reinterpret_cast<const uint32_t*>(validity)[(c * RED_CHUNK_ROWS + (uint64_t)j0 * 256) / 32 + lane];
#pragma unroll
      for (int u = 0; u < 8; u++) v[u] = __builtin_nontemporal_load(base + (j0 + u) * AGPU_WAVE);
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const uint32_t e0 = v[u].x, e1 = v[u].y, e2 = v[u].z, e3 = v[u].w;
        if constexpr (HASV) {
          const uint32_t nib = ((uint32_t)__shfl((int)vword, u * 8 + (int)(lane >> 3)) >> ((lane & 7u) * 4)) & 0xFu;
          if (nib & 1) a0 = Red::combine(a0, Red::load(word_as<T>(e0)));
          if (nib & 2) a1 = Red::combine(a1, Red::load(word_as<T>(e1)));
          if (nib & 4) a2 = Red::combine(a2, Red::load(word_as<T>(e2)));
          if (nib & 8) a3 = Red::combine(a3, Red::load(word_as<T>(e3)));
        } else {
          a0 = Red::combine(a0, Red::load(word_as<T>(e0)));
          a1 = Red::combine(a1, Red::load(word_as<T>(e1)));
          a2 = Red::combine(a2, Red::load(word_as<T>(e2)));
          a3 = Red::combine(a3, Red::load(word_as<T>(e3)));
        }
      }
    }
    A acc = Red::combine(Red::combine(a0, a1), Red::combine(a2, a3));
#pragma unroll
    for (int off = AGPU_WAVE / 2; off > 0; off >>= 1) acc = Red::combine(acc, shfl_down_acc(acc, off));
    if (lane == 0) partials[c] = Red::to_part(acc);
  }
}
// The wave kernel's second (and last) launch: ONE 1024-thread workgroup folds the < 1-chunk tail of the column and all the partials
// and writes the result.  Rounds 3–6a ran three launches here (a 256-thread block over the tail, ≤ 256 blocks folding the partials, one
// block finishing: 4.6 + 5.5 + 4.2 µs of kernels and three boundaries behind a 0.57–0.59 ms read of 1e9 rows — the whole distance
// between the reductions at 0.82–0.85 of the roof and a bare read-only kernel of the same shape at 0.87, tools/probe/stream_split.hip).
// 61 035 partials are 244 KB (488 for the f64 sum): every thread issues all its 16-byte loads at once (≤ 16 in flight, two rounds for 8-byte partials)
// and the workgroup pays one memory latency, not sixty.  The order of the fold is a function of the number of partials alone (and with it
// the workgroup's size: 256 threads up to 16 384 partials, 1024 beyond), so the f64 sum's last bits depend on the column's length only.
#define AGPU_FOLD_BLOCK 1024
// the body: every thread of the block calls it; the first `nthr` of them (a multiple of 64) do the work — the order of the fold is nthr's
template <typename T, typename Red>
__device__ __forceinline__ void fold_finish_body(const typename Red::Part* partials, uint64_t m, const T* tail_in, uint32_t tail, typename Red::Out* out,
                                                 uint32_t nthr, typename Red::Acc* lds /* [AGPU_FOLD_BLOCK / AGPU_WAVE] */,
                                                 const uint8_t* validity = nullptr, uint64_t tail_row0 = 0) {
  typedef typename Red::Acc A;
  typedef typename Red::Part Q;
  constexpr int PER = 16 / (int)sizeof(Q);  // partials per 16-byte vector: 4 or 2
  constexpr int B = 16;                     // vectors in flight per thread
  struct Pack {
    Q a[PER];
  };
  static_assert(sizeof(Pack) == 16, "partials tile a 16-byte vector");
  A acc = Red::identity();
  if (threadIdx.x < nthr) {
    for (uint32_t i = threadIdx.x; i < tail; i += nthr) {
      const uint64_t row = tail_row0 + i;
      if (!validity || ((validity[row >> 3] >> (row & 7)) & 1)) acc = Red::combine(acc, Red::load(tail_in[i]));
    }
    const uint64_t nvec = m / PER;
    const u32x4* pv = reinterpret_cast<const u32x4*>(partials);
    for (uint64_t v0 = 0; v0 < nvec; v0 += (uint64_t)nthr * B) {
      u32x4 v[B];
#pragma unroll
      for (int u = 0; u < B; u++) {
        const uint64_t idx = v0 + (uint64_t)u * nthr + threadIdx.x;
        if (idx < nvec) v[u] = pv[idx];
      }
#pragma unroll
      for (int u = 0; u < B; u++) {
        const uint64_t idx = v0 + (uint64_t)u * nthr + threadIdx.x;
        if (idx < nvec) {
          const Pack pk = __builtin_bit_cast(Pack, v[u]);
#pragma unroll
          for (int k = 0; k < PER; k++) acc = Red::combine(acc, Red::from_part(pk.a[k]));
        }
      }
    }
    for (uint64_t i = nvec * PER + threadIdx.x; i < m; i += nthr) acc = Red::combine(acc, Red::from_part(partials[i]));
#pragma unroll
    for (int off = AGPU_WAVE / 2; off > 0; off >>= 1) acc = Red::combine(acc, shfl_down_acc(acc, off));
    if ((threadIdx.x & (AGPU_WAVE - 1)) == 0) lds[threadIdx.x / AGPU_WAVE] = acc;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    A r = lds[0];
    for (uint32_t k = 1; k < nthr / AGPU_WAVE; k++) r = Red::combine(r, lds[k]);
    out[0] = Red::finish(r);
  }
}
template <typename T, typename Red>
__global__ __launch_bounds__(AGPU_FOLD_BLOCK) void reduce_fold_finish_kernel(const typename Red::Part* partials, uint64_t m, const T* tail_in,
                                                                            uint32_t tail, typename Red::Out* out, const uint8_t* validity,
                                                                            uint64_t tail_row0) {
  __shared__ typename Red::Acc lds[AGPU_FOLD_BLOCK / AGPU_WAVE];
  fold_finish_body<T, Red>(partials, m, tail_in, tail, out, blockDim.x, lds, validity, tail_row0);
}

template <typename T, typename Red>
static agpu_status launch_reduce(agpu_pipeline* p, const void* in, const void* validity, uint64_t n, void* out) {
  typedef typename Red::Acc A;
  constexpr int U = 4;
  if (AGPU_REDUCE_WAVE && aligned16(in) && n >= 64 * RED_CHUNK_ROWS && (!validity || aligned_to(validity, 4))) {
    const uint64_t nchunks = n / RED_CHUNK_ROWS, tail = n - nchunks * RED_CHUNK_ROWS;
    const uint64_t g = nchunks < 0x3FFFFFFFull ? nchunks : 0x3FFFFFFFull;
    const unsigned fb = nchunks > 16384 ? AGPU_FOLD_BLOCK : 256;
    const T* tin = static_cast<const T*>(in);
    const uint8_t* vb = static_cast<const uint8_t*>(validity);
    void* scratch = nullptr;
    if (!validity) {
      typedef typename Red::Part Q;
      agpu_status st = agpu_scratch(p, sizeof(Q) * (size_t)nchunks + 16, &scratch);
      if (st != AGPU_OK) return st;
      Q* partials = static_cast<Q*>(scratch);
      hipLaunchKernelGGL((reduce_wave_kernel<T, Red, false>), dim3((unsigned)g), dim3(AGPU_WAVE), 0, p->stream, tin, vb, partials, nchunks);
      AGPU_LAUNCH_CHECK();
      hipLaunchKernelGGL((reduce_fold_finish_kernel<T, Red>), dim3(1), dim3(fb), 0, p->stream, (const Q*)partials, nchunks,
                         tin + nchunks * RED_CHUNK_ROWS, (uint32_t)tail, static_cast<typename Red::Out*>(out), vb, nchunks * RED_CHUNK_ROWS);
    } else {  // null-aware (round 6b): the same two launches, the validity bits read beside the values
      typedef typename NullAwareRed<Red>::type RedV;
      typedef typename RedV::Part Q;
      agpu_status st = agpu_scratch(p, sizeof(Q) * (size_t)nchunks + 16, &scratch);
      if (st != AGPU_OK) return st;
      Q* partials = static_cast<Q*>(scratch);
      hipLaunchKernelGGL((reduce_wave_kernel<T, RedV, true>), dim3((unsigned)g), dim3(AGPU_WAVE), 0, p->stream, tin, vb, partials, nchunks);
      AGPU_LAUNCH_CHECK();
      hipLaunchKernelGGL((reduce_fold_finish_kernel<T, RedV>), dim3(1), dim3(fb), 0, p->stream, (const Q*)partials, nchunks,
                         tin + nchunks * RED_CHUNK_ROWS, (uint32_t)tail, static_cast<typename RedV::Out*>(out), vb, nchunks * RED_CHUNK_ROWS);
    }
    AGPU_LAUNCH_CHECK();
    return AGPU_OK;
  }
  const int grid = reduce_grid_for(p, (n / 4 + (uint64_t)AGPU_BLOCK * U - 1) / ((uint64_t)AGPU_BLOCK * U), 16);
  void* scratch = nullptr;
  agpu_status st = agpu_scratch(p, sizeof(A) * (size_t)grid, &scratch);
  if (st != AGPU_OK) return st;
  hipLaunchKernelGGL((reduce_partial_kernel<T, Red, U>), dim3(grid), dim3(AGPU_BLOCK), 0, p->stream,
                     static_cast<const T*>(in), static_cast<const uint8_t*>(validity), n, static_cast<A*>(scratch),
                     aligned16(in) ? 1 : 0);
  AGPU_LAUNCH_CHECK();
  hipLaunchKernelGGL((reduce_finish_kernel<Red>), dim3(1), dim3(AGPU_BLOCK), 0, p->stream,
                     static_cast<const A*>(scratch), (uint32_t)grid, static_cast<typename Red::Out*>(out));
  AGPU_LAUNCH_CHECK();
  return AGPU_OK;
}

static agpu_status sum_tree_after_quarters(agpu_pipeline* p, const float* quarters, uint64_t nspans, float* groups, float* buf0, float* buf1,
                                           float* out);
static agpu_status launch_sum_tree_f32(agpu_pipeline* p, const float* in, const uint8_t* validity, uint64_t n,
                                       float* out) {
  const uint64_t nspans = (n + SPAN_ROWS - 1) / SPAN_ROWS, nquarters = nspans * 4;
  const uint64_t ngroups = (nspans + 255) / 256;  // values left after the combine launch
  // scratch: quarters[4 nspans] + groups[ngroups] + two ping-pong buffers for the finishing levels
  const size_t level_cap = (size_t)((ngroups + 255) / 256 + 1);
  const size_t floats = (size_t)nquarters + 4 + (size_t)ngroups + 4 + 2 * level_cap + 8;
  void* scratch = nullptr;
  agpu_status st = agpu_scratch(p, floats * sizeof(float), &scratch);
  if (st != AGPU_OK) return st;
  float* quarters = static_cast<float*>(scratch);
  float* groups = quarters + nquarters + 4;
  float* buf0 = groups + ngroups + 4;
  float* buf1 = buf0 + level_cap;
  if (n <= 256) {
    // one plain reference level straight from the input (a single 256-row workgroup)
    hipLaunchKernelGGL(sum_tree_finish_kernel, dim3(1), dim3(AGPU_BLOCK), 0, p->stream, in, validity, buf0, buf1, n, out, 1);
    AGPU_LAUNCH_CHECK();
    return AGPU_OK;
  }
  const int grid = reduce_grid_for(p, nquarters, 1 << 16);  // one-wave blocks: as many as there are quarters (sum_probe.py)
  const int vec_ok = (aligned16(in) && (!validity || aligned_to(validity, 4))) ? 1 : 0;
  if (validity)
    hipLaunchKernelGGL((sum_tree_quarter_kernel<true>), dim3(grid), dim3(AGPU_WAVE), 0, p->stream, in, validity, n, quarters, nquarters, vec_ok, (uint64_t)0);
  else
    hipLaunchKernelGGL((sum_tree_quarter_kernel<false>), dim3(grid), dim3(AGPU_WAVE), 0, p->stream, in, validity, n, quarters, nquarters, vec_ok, (uint64_t)0);
  AGPU_LAUNCH_CHECK();
  return sum_tree_after_quarters(p, quarters, nspans, groups, buf0, buf1, out);
}

// the levels behind the quarter sums (launch_sum_tree_f32 and the one-pass statistics share them)
static agpu_status sum_tree_after_quarters(agpu_pipeline* p, const float* quarters, uint64_t nspans, float* groups, float* buf0, float* buf1,
                                           float* out) {
  const uint64_t ngroups = (nspans + 255) / 256;
  if (ngroups > 1 && ngroups <= 256) {  // 16.8 M … 4.29e9 rows: combine + the last level in ONE workgroup
    hipLaunchKernelGGL(sum_tree_combine_finish_kernel, dim3(1), dim3(AGPU_FOLD_BLOCK_SUM), 0, p->stream, (const float*)quarters, nspans,
                       (uint32_t)ngroups, out);
    AGPU_LAUNCH_CHECK();
    return AGPU_OK;
  }
  // ≤ 256 spans (16.7 M rows): the combine launch produces the result itself
  hipLaunchKernelGGL(sum_tree_combine_kernel, dim3((unsigned)ngroups), dim3(AGPU_WAVE), 0, p->stream, (const float*)quarters, nspans,
                     ngroups == 1 ? out : groups);
  AGPU_LAUNCH_CHECK();
  if (ngroups > 1) {
    hipLaunchKernelGGL(sum_tree_finish_kernel, dim3(1), dim3(AGPU_BLOCK), 0, p->stream, (const float*)groups,
                       (const uint8_t*)nullptr, buf0, buf1, ngroups, out, 0);
    AGPU_LAUNCH_CHECK();
  }
  return AGPU_OK;
}

// ---------------------------------------------------------------- whole-column f32 statistics in ONE pass
// north_star config 5 wants sum / min / max of one f32 column (the bench adds the f64-accumulated sum): four agpu_reduce calls read the
// column four times — 4 × 0.58 ms at 1e9 rows.  Here one wave retires one FULL aligned 16 384-row quarter for all four statistics from the
// same registers: the tree sum exactly as quarter_tree_sum does it (→ quarters[q]), min and max with NaN rows mapped to the identities
// (partial = NaN exactly when the whole quarter was NaN: RedMinMaxF32's 4-byte partial), the f64 sum in reduce_wave_kernel's order (four
// interleaved accumulators per lane over the lane's 64 vectors, (a0+a1)+(a2+a3), shift-down shuffles) — so every partial, and with the
// SAME finishing launches behind them every result, is bit-identical to what the four separate reductions give.  4 B/row, HBM-bound: ≈ 8
// VALU instructions per row (two of them f64) ride under the loads.
// HASV (null-aware): the validity bits as reduce_wave_kernel<…, true> / quarter_tree_sum<false, true> read them; an invalid row counts as +0.0
// in the two sums and is skipped by min / max, whose partials are then RedMinMaxF32V's {value, flags} (a quarter may hold no valid row at all).
template <bool HASV>
__global__ __launch_bounds__(AGPU_WAVE) void stats_quarter_kernel(const float* in, const uint8_t* validity, float* quarters, void* mins_v, void* maxs_v,
                                                                 double* dsums, uint64_t nfull) {
  constexpr int UNR = 8;
  const uint32_t lane = threadIdx.x;
  for (uint64_t q = blockIdx.x; q < nfull; q += gridDim.x) {
    const uint64_t base = q * WAVE_ROWS;
    float acc = 0.0f;
    float mn = __builtin_inff(), mx = -__builtin_inff();  // the identities: mn > mx for as long as no (valid) non-NaN row has been seen
    bool saw_nan = false;                                 // HASV only: a VALID NaN row (without validity: "nothing seen" = "only NaN seen")
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    for (int j0 = 0; j0 < AGPU_WAVE; j0 += UNR) {
      f32x4 v[UNR];
      uint32_t vword = 0;
      if constexpr (HASV) vword = reinterpret_cast<const uint32_t*>(validity)[(base + (uint64_t)j0 * 256) / 32 + lane];
#pragma unroll
      for (int u = 0; u < UNR; u++) v[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(in + base + (uint64_t)(j0 + u) * 256 + lane * 4));
      float s[UNR];
#pragma unroll
      for (int u = 0; u < UNR; u++) {
        float x0 = v[u].x, x1 = v[u].y, x2 = v[u].z, x3 = v[u].w;
        float c0, c1, c2, c3;
        if constexpr (HASV) {
          const uint32_t nib = ((uint32_t)__shfl((int)vword, u * 8 + (int)(lane >> 3)) >> ((lane & 7u) * 4)) & 0xFu;
          const bool v0 = nib & 1, v1 = nib & 2, v2 = nib & 4, v3 = nib & 8;
          const float qnan = __builtin_nanf("");
          // min / max see an invalid row as a (quiet) NaN — skipped; a valid NaN row is remembered
          c0 = v0 ? __builtin_canonicalizef(x0) : qnan;
          c1 = v1 ? __builtin_canonicalizef(x1) : qnan;
          c2 = v2 ? __builtin_canonicalizef(x2) : qnan;
          c3 = v3 ? __builtin_canonicalizef(x3) : qnan;
          saw_nan = saw_nan || (v0 && x0 != x0) || (v1 && x1 != x1) || (v2 && x2 != x2) || (v3 && x3 != x3);
          x0 = v0 ? x0 : 0.0f;  // the sums: load4_tree<…, true>'s rule
          x1 = v1 ? x1 : 0.0f;
          x2 = v2 ? x2 : 0.0f;
          x3 = v3 ? x3 : 0.0f;
          // (reduce_wave_kernel skips an invalid row where this adds +0.0: the same bits — an accumulator that starts at +0.0 never holds −0.0)
        } else {
          // NaN rows: a QUIET NaN operand makes v_min_f32 / v_max_f32 return the other one (IEEE mode), so one canonicalising v_max_f32 x, x per
          // row (it quiets a signalling NaN) replaces the compare and the two selects that map NaN to the identities — same result: the
          // extremes of the non-NaN rows, −0.0 below +0.0
          c0 = __builtin_canonicalizef(x0), c1 = __builtin_canonicalizef(x1), c2 = __builtin_canonicalizef(x2), c3 = __builtin_canonicalizef(x3);
        }
        s[u] = (x0 + x1) + (x2 + x3);
        a0 += (double)x0;
        a1 += (double)x1;
        a2 += (double)x2;
        a3 += (double)x3;
        mn = RedMinMaxF32<false>::pick(RedMinMaxF32<false>::pick(mn, c0), RedMinMaxF32<false>::pick(c1, c2));
        mn = RedMinMaxF32<false>::pick(mn, c3);
        mx = RedMinMaxF32<true>::pick(RedMinMaxF32<true>::pick(mx, c0), RedMinMaxF32<true>::pick(c1, c2));
        mx = RedMinMaxF32<true>::pick(mx, c3);
      }
      const float t = transpose_reduce8(s, lane);
      if ((lane >> 3) == (uint32_t)(j0 >> 3)) acc = t;
    }
    const float r = wave_tree_sum(acc);
    double d = (a0 + a1) + (a2 + a3);
    uint32_t sn = saw_nan ? 1u : 0u;
#pragma unroll
    for (int off = AGPU_WAVE / 2; off > 0; off >>= 1) {
      d = d + __shfl_down(d, off);
      mn = RedMinMaxF32<false>::pick(mn, __shfl_down(mn, off));
      mx = RedMinMaxF32<true>::pick(mx, __shfl_down(mx, off));
      if constexpr (HASV) sn |= (uint32_t)__shfl_down((int)sn, off);
    }
    if (lane == 0) {
      const bool seen = mn <= mx;  // any non-NaN row v leaves mn ≤ v ≤ mx; the untouched identities are +inf > −inf
      quarters[q] = r;
      dsums[q] = d;
      if constexpr (HASV) {
        const uint32_t flags = (seen ? 1u : 0u) | (sn ? 2u : 0u);
        static_cast<MinMaxF32*>(mins_v)[q] = MinMaxF32{mn, flags};
        static_cast<MinMaxF32*>(maxs_v)[q] = MinMaxF32{mx, flags};
      } else {
        static_cast<float*>(mins_v)[q] = seen ? mn : __builtin_nanf("");
        static_cast<float*>(maxs_v)[q] = seen ? mx : __builtin_nanf("");
      }
    }
  }
}

// ONE finishing launch behind stats_quarter_kernel, four workgroups side by side: block 0 the tree sum (the quarters at and beyond the end of
// the column in the tree sum's guarded form, one wave each, then sum_tree_combine_finish_kernel's body), blocks 1 / 2 / 3 min / max / the f64
// sum through reduce_fold_finish_kernel's body with the SAME thread count that kernel would be launched with (the fold's order, hence the f64
// sum's last bits, is a function of it).  do_sum = 0 (≤ 256 or > 65 536 spans): the host runs the tree sum's own launches instead.
template <bool HASV>
__global__ __launch_bounds__(AGPU_FOLD_BLOCK) void stats_finish_kernel(const float* in, const uint8_t* validity, uint64_t n, float* quarters, uint64_t nfull,
                                                                      uint64_t nquarters, uint64_t nspans, uint32_t ngroups, int do_sum, const void* mins,
                                                                      const void* maxs, const double* dsums, const float* tail_in, uint32_t tail,
                                                                      uint32_t nthr, agpu_f32_stats* out) {
  typedef typename std::conditional<HASV, RedMinMaxF32V<false>, RedMinMaxF32<false>>::type RMin;
  typedef typename std::conditional<HASV, RedMinMaxF32V<true>, RedMinMaxF32<true>>::type RMax;
  const uint8_t* vb = HASV ? validity : nullptr;
  const uint64_t tail_row0 = nfull * WAVE_ROWS;
  __shared__ float sh[256];
  __shared__ MinMaxF32 ldsm[AGPU_FOLD_BLOCK / AGPU_WAVE];
  __shared__ double ldsd[AGPU_FOLD_BLOCK / AGPU_WAVE];
  if (blockIdx.x == 0) {
    if (threadIdx.x == 0) out->reserved = 0;
    if (do_sum) {
      const uint32_t wave = threadIdx.x / AGPU_WAVE;
      if (nfull + wave < nquarters) {  // ≤ 4 quarters: the one that crosses the end of the column and the span's zero padding
        const float r = quarter_tree_sum<true, HASV>(in, vb, (nfull + wave) * WAVE_ROWS, n);
        if ((threadIdx.x & (AGPU_WAVE - 1)) == 0) quarters[nfull + wave] = r;
      }
      __threadfence_block();
      __syncthreads();
      sum_combine_finish_body(quarters, nspans, ngroups, &out->sum, sh);
    }
  } else if (blockIdx.x == 1) {
    fold_finish_body<float, RMin>(static_cast<const typename RMin::Part*>(mins), nfull, tail_in, tail, &out->min, nthr, ldsm, vb, tail_row0);
  } else if (blockIdx.x == 2) {
    fold_finish_body<float, RMax>(static_cast<const typename RMax::Part*>(maxs), nfull, tail_in, tail, &out->max, nthr, ldsm, vb, tail_row0);
  } else {
    fold_finish_body<float, RedSumF64>(dsums, nfull, tail_in, tail, &out->sum_f64, nthr, ldsd, vb, tail_row0);
  }
}

static agpu_status launch_stats_f32(agpu_pipeline* p, const float* in, const uint8_t* validity, uint64_t n, agpu_f32_stats* out) {
  float* o_sum = &out->sum;
  float* o_min = &out->min;
  float* o_max = &out->max;
  double* o_f64 = &out->sum_f64;
  const uint64_t nfull = n / WAVE_ROWS;
  if (!aligned16(in) || nfull < 64 || (validity && !aligned_to(validity, 4))) {  // unaligned or small columns: the four reductions one after the other
    AGPU_HIP(hipMemsetAsync(&out->reserved, 0, sizeof(out->reserved), p->stream));
    agpu_status st = launch_sum_tree_f32(p, in, validity, n, o_sum);
    if (st == AGPU_OK) st = launch_reduce<float, RedMinMaxF32<false>>(p, in, validity, n, o_min);
    if (st == AGPU_OK) st = launch_reduce<float, RedMinMaxF32<true>>(p, in, validity, n, o_max);
    if (st == AGPU_OK) st = launch_reduce<float, RedSumF64>(p, in, validity, n, o_f64);
    return st;
  }
  const uint64_t nspans = (n + SPAN_ROWS - 1) / SPAN_ROWS, nquarters = nspans * 4;
  const uint64_t ngroups = (nspans + 255) / 256;
  const size_t level_cap = (size_t)((ngroups + 255) / 256 + 1);
  // scratch: the tree sum's layout (launch_sum_tree_f32) + mins[nfull] + maxs[nfull] + dsums[nfull], every piece 16-byte aligned
  // (null-aware: the min / max partials are {value, flags} pairs, two floats each)
  const size_t f_tree = ((size_t)nquarters + 4 + (size_t)ngroups + 4 + 2 * level_cap + 8 + 3) & ~(size_t)3;
  const size_t f_part = ((size_t)nfull * (validity ? 2 : 1) + 3) & ~(size_t)3;
  void* scratch = nullptr;
  agpu_status st = agpu_scratch(p, (f_tree + 2 * f_part) * sizeof(float) + (size_t)nfull * sizeof(double) + 16, &scratch);
  if (st != AGPU_OK) return st;
  float* quarters = static_cast<float*>(scratch);
  float* groups = quarters + nquarters + 4;
  float* buf0 = groups + ngroups + 4;
  float* buf1 = buf0 + level_cap;
  float* mins = quarters + f_tree;
  float* maxs = mins + f_part;
  double* dsums = reinterpret_cast<double*>(maxs + f_part);
  const uint64_t g = nfull < 0x3FFFFFFFull ? nfull : 0x3FFFFFFFull;
  if (validity) hipLaunchKernelGGL((stats_quarter_kernel<true>), dim3((unsigned)g), dim3(AGPU_WAVE), 0, p->stream, in, validity, quarters, (void*)mins, (void*)maxs, dsums, nfull);
  else hipLaunchKernelGGL((stats_quarter_kernel<false>), dim3((unsigned)g), dim3(AGPU_WAVE), 0, p->stream, in, validity, quarters, (void*)mins, (void*)maxs, dsums, nfull);
  AGPU_LAUNCH_CHECK();
  const bool fused_sum = ngroups > 1 && ngroups <= 256 && nquarters - nfull <= AGPU_FOLD_BLOCK / AGPU_WAVE;
  if (!fused_sum) {  // ≤ 16.7 M or > 4.29e9 rows: the tree sum's own launches behind the quarters
    if (nfull < nquarters) {
      if (validity)
        hipLaunchKernelGGL((sum_tree_quarter_kernel<true>), dim3((unsigned)(nquarters - nfull)), dim3(AGPU_WAVE), 0, p->stream, in, validity, n, quarters,
                           nquarters, 1, nfull);
      else
        hipLaunchKernelGGL((sum_tree_quarter_kernel<false>), dim3((unsigned)(nquarters - nfull)), dim3(AGPU_WAVE), 0, p->stream, in, validity, n, quarters,
                           nquarters, 1, nfull);
      AGPU_LAUNCH_CHECK();
    }
    st = sum_tree_after_quarters(p, quarters, nspans, groups, buf0, buf1, o_sum);
    if (st != AGPU_OK) return st;
  }
  // min / max / f64 sum: reduce_wave_kernel's finishing workgroup over the same partials (the < 1-quarter tail's rows included), with the thread
  // count launch_reduce would give it
  const float* tail_in = in + nfull * WAVE_ROWS;
  const uint32_t tail = (uint32_t)(n - nfull * WAVE_ROWS);
  const uint32_t nthr = nfull > 16384 ? AGPU_FOLD_BLOCK : 256;
  if (validity)
    hipLaunchKernelGGL((stats_finish_kernel<true>), dim3(4), dim3(AGPU_FOLD_BLOCK), 0, p->stream, in, validity, n, quarters, nfull, nquarters, nspans,
                       (uint32_t)ngroups, fused_sum ? 1 : 0, (const void*)mins, (const void*)maxs, (const double*)dsums, tail_in, tail, nthr, out);
  else
    hipLaunchKernelGGL((stats_finish_kernel<false>), dim3(4), dim3(AGPU_FOLD_BLOCK), 0, p->stream, in, validity, n, quarters, nfull, nquarters, nspans,
                       (uint32_t)ngroups, fused_sum ? 1 : 0, (const void*)mins, (const void*)maxs, (const double*)dsums, tail_in, tail, nthr, out);
  AGPU_LAUNCH_CHECK();
  return AGPU_OK;
}

// ---------------------------------------------------------------- one level of the reference's tree (by_name.hip)
// The reference's `sum` entry point reduces 256 rows per workgroup and is dispatched once per level by Sum::sum_op
// [ref: aggregate.wgsl:21-41, aggregate_kernels.rs:26-43].  agpu_launch_by_name_sized keeps that call shape: workgroup g
// writes out[g] = adjacent-pair tree over in[256g .. 256g+255], rows ≥ m count as 0.  (agpu_reduce runs the whole tree.)
template <typename T>
__global__ __launch_bounds__(AGPU_BLOCK) void sum_level_kernel(const T* in, uint64_t m, T* out, uint64_t groups) {
  __shared__ T sh[AGPU_BLOCK];
  const uint32_t tid = threadIdx.x;
  for (uint64_t g = blockIdx.x; g < groups; g += gridDim.x) {
    const uint64_t i = g * 256 + tid;
    sh[tid] = i < m ? in[i] : (T)0;
    __syncthreads();
    for (uint32_t s = 1; s < 256; s *= 2) {
      const uint32_t idx = 2 * s * tid;
      if (idx + s < 256) sh[idx] = (T)(sh[idx] + sh[idx + s]);  // u32/i32: wrapping; f32: rounds once per add
      __syncthreads();
    }
    if (tid == 0) out[g] = sh[0];
    __syncthreads();
  }
}

agpu_status agpu_internal_sum_level(agpu_pipeline* p, agpu_dtype dtype, const void* in, uint64_t m, void* out, uint64_t groups) {
  if (groups == 0) return AGPU_OK;
  AGPU_REQUIRE(in && out, AGPU_ERR_ARG, "null pointer");
  const int grid = (int)(groups < (1u << 20) ? groups : (1u << 20));
  if (dtype == AGPU_F32)
    hipLaunchKernelGGL((sum_level_kernel<float>), dim3(grid), dim3(AGPU_BLOCK), 0, p->stream, static_cast<const float*>(in), m, static_cast<float*>(out), groups);
  else if (dtype == AGPU_I32 || dtype == AGPU_U32 || dtype == AGPU_DATE32)
    hipLaunchKernelGGL((sum_level_kernel<uint32_t>), dim3(grid), dim3(AGPU_BLOCK), 0, p->stream, static_cast<const uint32_t*>(in), m, static_cast<uint32_t*>(out), groups);
  else {
    agpu_set_error("sum: 32-bit types only (like the reference's Sum32Bit)");
    return AGPU_ERR_UNSUPPORTED;
  }
  AGPU_LAUNCH_CHECK();
  return AGPU_OK;
}

// ---------------------------------------------------------------- multi-GPU final reduce (comm.hip)
// One 16-byte record per rank {statistic in the low 4 / 8 bytes, n_local}, gathered over RCCL, combined here IN RANK
// ORDER by one workgroup on every rank — deterministic and identical everywhere, independent of RCCL's ring order.
struct CommRecord {
  uint64_t bits;
  uint64_t n_local;
};

__global__ void comm_pack_kernel(CommRecord* rec, uint64_t n_local) {
  rec->bits = 0;
  rec->n_local = n_local;
}

// f32 SUM: the shard sums are one more level of the reference's tree — adjacent pairs, zero-padded to 256 entries
// [ref: aggregate.wgsl:21-41].  Shards of 256^k rows therefore reproduce the reference's whole-column tree bit for bit.
__global__ __launch_bounds__(AGPU_BLOCK) void comm_finish_sum_f32_kernel(const CommRecord* rec, int world, float* out) {
  __shared__ float sh[AGPU_BLOCK];
  const uint32_t tid = threadIdx.x;
  sh[tid] = (int)tid < world ? __builtin_bit_cast(float, (uint32_t)rec[tid].bits) : 0.0f;
  __syncthreads();
  for (uint32_t s = 1; s < 256; s *= 2) {
    const uint32_t idx = 2 * s * tid;
    if (idx + s < 256) sh[idx] = sh[idx] + sh[idx + s];
    __syncthreads();
  }
  if (tid == 0) out[0] = sh[0];
}

template <typename T, typename Red>
__global__ void comm_finish_kernel(const CommRecord* rec, int world, typename Red::Out* out) {
  if (threadIdx.x != 0) return;
  typename Red::Acc acc = Red::identity();
  for (int r = 0; r < world; r++) {
    if (rec[r].n_local == 0) continue;  // an empty shard contributes the identity
    T x;
    if constexpr (sizeof(T) == 8) x = __builtin_bit_cast(T, rec[r].bits);
    else x = __builtin_bit_cast(T, (uint32_t)rec[r].bits);
    acc = Red::combine(acc, Red::load(x));
  }
  out[0] = Red::finish(acc);
}

struct RedSumF64Plain {  // rank-ordered f64 sum of f64 partials
  typedef double Acc;
  typedef double Out;
  __device__ static Acc identity() { return 0.0; }
  __device__ static Acc load(double x) { return x; }
  __device__ static Acc combine(Acc a, Acc b) { return a + b; }
  __device__ static Out finish(Acc a) { return a; }
};

agpu_status agpu_internal_comm_pack(agpu_pipeline* p, void* record, uint64_t n_local) {
  hipLaunchKernelGGL(comm_pack_kernel, dim3(1), dim3(1), 0, p->stream, static_cast<CommRecord*>(record), n_local);
  AGPU_LAUNCH_CHECK();
  return AGPU_OK;
}

agpu_status agpu_internal_comm_finish(agpu_pipeline* p, int kind, agpu_dtype dtype, const void* records, int world,
                                      void* out_dev) {
  const CommRecord* rec = static_cast<const CommRecord*>(records);
  typedef RedMinMaxInt<int32_t, false> MinI32;
  typedef RedMinMaxInt<uint32_t, false> MinU32;
  typedef RedMinMaxInt<int32_t, true> MaxI32;
  typedef RedMinMaxInt<uint32_t, true> MaxU32;
#define FIN(T, RED) \
  hipLaunchKernelGGL((comm_finish_kernel<T, RED>), dim3(1), dim3(AGPU_WAVE), 0, p->stream, rec, world, static_cast<typename RED::Out*>(out_dev))
  if (kind == 3) {
    FIN(double, RedSumF64Plain);
  } else if (kind == AGPU_RED_SUM) {
    if (dtype == AGPU_F32)
      hipLaunchKernelGGL(comm_finish_sum_f32_kernel, dim3(1), dim3(AGPU_BLOCK), 0, p->stream, rec, world, static_cast<float*>(out_dev));
    else if (dtype == AGPU_I32) FIN(int32_t, RedSumWrap<int32_t>);
    else if (dtype == AGPU_U32) FIN(uint32_t, RedSumWrap<uint32_t>);
    else goto unsupported;
  } else if (kind == AGPU_RED_MIN) {
    if (dtype == AGPU_F32) FIN(float, RedMinMaxF32<false>);
    else if (dtype == AGPU_I32) FIN(int32_t, MinI32);
    else if (dtype == AGPU_U32) FIN(uint32_t, MinU32);
    else goto unsupported;
  } else if (kind == AGPU_RED_MAX) {
    if (dtype == AGPU_F32) FIN(float, RedMinMaxF32<true>);
    else if (dtype == AGPU_I32) FIN(int32_t, MaxI32);
    else if (dtype == AGPU_U32) FIN(uint32_t, MaxU32);
    else goto unsupported;
  } else {
    goto unsupported;
  }
#undef FIN
  AGPU_LAUNCH_CHECK();
  return AGPU_OK;
unsupported:
  agpu_set_error("final reduce: op %d not supported for dtype %d", kind, (int)dtype);
  return AGPU_ERR_UNSUPPORTED;
}

extern "C" {

agpu_status agpu_reduce(agpu_pipeline* p, agpu_reduce_op op, agpu_dtype dtype, const void* in, const void* validity,
                        uint64_t n, void* out_dev) {
  AGPU_BIND(p);
  AGPU_REQUIRE(out_dev && (n == 0 || in), AGPU_ERR_ARG, "null pointer");
  AGPU_REQUIRE(n == 0 || aligned_to(in, 4), AGPU_ERR_SHAPE, "input must be 4-byte aligned");
  if (dtype == AGPU_DATE32) dtype = AGPU_I32;
  if (op == AGPU_RED_SUM) {
    switch (dtype) {
      case AGPU_F32:
        return launch_sum_tree_f32(p, static_cast<const float*>(in), static_cast<const uint8_t*>(validity), n,
                                   static_cast<float*>(out_dev));
      case AGPU_I32: return launch_reduce<int32_t, RedSumWrap<int32_t>>(p, in, validity, n, out_dev);
      case AGPU_U32: return launch_reduce<uint32_t, RedSumWrap<uint32_t>>(p, in, validity, n, out_dev);
      default: break;
    }
  } else if (op == AGPU_RED_MIN) {
    switch (dtype) {
      case AGPU_F32: return launch_reduce<float, RedMinMaxF32<false>>(p, in, validity, n, out_dev);
      case AGPU_I32: return launch_reduce<int32_t, RedMinMaxInt<int32_t, false>>(p, in, validity, n, out_dev);
      case AGPU_U32: return launch_reduce<uint32_t, RedMinMaxInt<uint32_t, false>>(p, in, validity, n, out_dev);
      default: break;
    }
  } else if (op == AGPU_RED_MAX) {
    switch (dtype) {
      case AGPU_F32: return launch_reduce<float, RedMinMaxF32<true>>(p, in, validity, n, out_dev);
      case AGPU_I32: return launch_reduce<int32_t, RedMinMaxInt<int32_t, true>>(p, in, validity, n, out_dev);
      case AGPU_U32: return launch_reduce<uint32_t, RedMinMaxInt<uint32_t, true>>(p, in, validity, n, out_dev);
      default: break;
    }
  } else {
    agpu_set_error("bad reduce op %d", (int)op);
    return AGPU_ERR_ARG;
  }
  agpu_set_error("reduce op %d not supported for dtype %d (32-bit types only, like the reference's Sum32Bit)", (int)op,
                 (int)dtype);
  return AGPU_ERR_UNSUPPORTED;
}

agpu_status agpu_reduce_combine(agpu_pipeline* p, agpu_reduce_op op, agpu_dtype dtype, int32_t kind_f64,
                                const void* records_dev, int32_t world, void* out_dev) {
  AGPU_BIND(p);
  AGPU_REQUIRE(records_dev && out_dev, AGPU_ERR_ARG, "null pointer");
  AGPU_REQUIRE(world >= 1 && world <= 256, AGPU_ERR_ARG, "1..256 records");
  AGPU_REQUIRE((int)op >= 0 && (int)op <= 2, AGPU_ERR_ARG, "bad reduce op");
  AGPU_REQUIRE(!kind_f64 || op == AGPU_RED_SUM, AGPU_ERR_UNSUPPORTED, "f64 partials are sums only");
  if (dtype == AGPU_DATE32) dtype = AGPU_I32;
  return agpu_internal_comm_finish(p, kind_f64 ? 3 : (int)op, dtype, records_dev, world, out_dev);
}

agpu_status agpu_reduce_sum_f64(agpu_pipeline* p, const float* in, const void* validity, uint64_t n, double* out_dev) {
  AGPU_BIND(p);
  AGPU_REQUIRE(out_dev && (n == 0 || in), AGPU_ERR_ARG, "null pointer");
  return launch_reduce<float, RedSumF64>(p, in, validity, n, out_dev);
}

agpu_status agpu_reduce_stats_f32(agpu_pipeline* p, const float* in, const void* validity, uint64_t n, agpu_f32_stats* out_dev) {
  AGPU_BIND(p);
  AGPU_REQUIRE(out_dev && (n == 0 || in), AGPU_ERR_ARG, "null pointer");
  AGPU_REQUIRE(n == 0 || aligned_to(in, 4), AGPU_ERR_SHAPE, "input must be 4-byte aligned");
  AGPU_REQUIRE(aligned_to(out_dev, 8), AGPU_ERR_SHAPE, "the statistics record must be 8-byte aligned");
  return launch_stats_f32(p, in, static_cast<const uint8_t*>(validity), n, out_dev);
}

}  // extern "C"
