// compare.hip — compare → LSB-first packed bitmap, optionally fused with the validity AND.
//
// Replaces crates/compare/compute_shaders/{f32,i32,u32,u16,i16,u8,i8}/cmp.wgsl (entry points gt gteq lt lteq eq:
// one invocation per element, atomicOr into 8 workgroup words, barrier, lanes gid%32==0 store) and the host step
// that follows it, NullBitBufferGpu::merge_null_bit_buffer_op (crates/compare/src/lib.rs:85-111,
// crates/array/src/array/null_bit_buffer.rs:206-243).
//
// MI355X design (HBM-bound: 8 B in + 1 bit out per row for 32-bit types, +3 bits with validity):
//   variant 0 "ballot"  (32-bit types): lane r-th load covers element e0 + r*64 + lane, so `__ballot(pred)` IS the
//       output word for those 64 rows — no LDS, no atomics, no barrier.  R = 4 rounds are kept in flight per wave
//       (4 dword loads per input array per lane), the masks are steered to lanes 0..3 and stored as one 32-byte row.
//       The validity AND runs in dedicated blocks of the same launch (validity_block below).
//   variant 1 "vector"  (all types; default for sub-word types): lane loads 16-byte vectors (4/8/16 elements),
//       builds an N-bit mask, and G = 32/N neighbouring lanes OR their shifted masks with xor-shuffles.
// Bits past n in the last word(s) are written as 0 (the reference leaves them unspecified).
#include <type_traits>

#include "common.hpp"
// The result bitmaps are 1/64 of a compare's traffic, and where the driver backed THEIR allocation decides 5–7 % of the
// launch (0.81 ↔ 0.87 for eq + validity with identical inputs and layout; any offset inside one allocation behaves the
// same: tools/probe/placement_lottery.py, out_offset.py).  Nontemporal stores for them halve the penalty of an
// unlucky allocation (compare words: 0.813 → 0.834 / 0.870 → 0.875; with the validity words too: 0.848 / 0.886 —
// alternations of bench.py on one box).
#ifndef AGPU_CMP_OUT_NT
#define AGPU_CMP_OUT_NT 1
#endif
#ifndef AGPU_CMP_OUTV_NT
#define AGPU_CMP_OUTV_NT 1
#endif

template <int OP, typename T>
__device__ __forceinline__ bool cmp_pred(T x, T y) {
  if constexpr (OP == AGPU_CMP_GT) return x > y;
  else if constexpr (OP == AGPU_CMP_GTEQ) return x >= y;
  else if constexpr (OP == AGPU_CMP_LT) return x < y;
  else if constexpr (OP == AGPU_CMP_LTEQ) return x <= y;
  else return x == y;
}

// one 64-bit output word computed element by element with bounds (tails only)
template <int OP, typename T>
__device__ __forceinline__ uint64_t cmp_word_guarded(const T* a, const T* b, uint64_t w, uint64_t n) {
  uint64_t m = 0;
  const uint64_t lo = w * 64;
  for (uint32_t k = 0; k < 64 && lo + k < n; k++) m |= (uint64_t)cmp_pred<OP, T>(a[lo + k], b[lo + k]) << k;
  return m;
}

__device__ __forceinline__ uint64_t validity_word(const uint64_t* va, const uint64_t* vb, uint64_t w) {
  const uint64_t x = va ? va[w] : ~0ull;
  const uint64_t y = vb ? vb[w] : ~0ull;
  return x & y;
}

// Validity AND inside the compare launch.  The two jobs share no bytes, so "fusing" them per tile (every compare block
// also moving its 16 validity words) only adds 3 small memory instructions to 10^6 short blocks: 1.395 ms fused vs
// 1.252 + 0.067 ms as two launches at 1e9 rows.  Instead the grid gets DEDICATED validity blocks in front of the compare
// tiles (virtual block ids [0, nvb) of the same launch): each moves 4 KiB per bitmap with 16-byte accesses, the shape
// the stand-alone bitmap kernel uses, and overlaps with the first compare tiles.
typedef uint64_t cmp_u64x2 __attribute__((ext_vector_type(2)));
#define CMP_VBLOCK_WORDS (2 * AGPU_BLOCK)  // u64 words of each bitmap per validity block
// vcount != nullptr: the NULL COUNT as a by-product — the block adds v_bcnt of the words it stores (wave shuffles, four
// words through LDS) and leaves one u32 at vcount[vblk]; bitmap.hip's count_fold_kernel sums them (no atomics, no second
// pass over the bitmap)
__device__ __forceinline__ void validity_block(const uint64_t* va, const uint64_t* vb, uint64_t* outv, uint64_t vblk,
                                               uint64_t n_words, bool vec16, uint32_t* vcount) {
  const uint64_t w = vblk * CMP_VBLOCK_WORDS + 2 * threadIdx.x;
  uint32_t cnt = 0;
  if (vec16 && w + 2 <= n_words) {
    const cmp_u64x2 ones = {~0ull, ~0ull};
    const cmp_u64x2 x = va ? *reinterpret_cast<const cmp_u64x2*>(va + w) : ones;  // (nontemporal loads here: −0.4 %)
    const cmp_u64x2 y = vb ? *reinterpret_cast<const cmp_u64x2*>(vb + w) : ones;
    const cmp_u64x2 r = x & y;
#if AGPU_CMP_OUTV_NT
    __builtin_nontemporal_store(r, reinterpret_cast<cmp_u64x2*>(outv + w));
#else
    *reinterpret_cast<cmp_u64x2*>(outv + w) = r;
#endif
    cnt = (uint32_t)__popcll(r.x) + (uint32_t)__popcll(r.y);
  } else {
    if (w < n_words) {
      const uint64_t r = validity_word(va, vb, w);
      outv[w] = r;
      cnt = (uint32_t)__popcll(r);
    }
    if (w + 1 < n_words) {
      const uint64_t r = validity_word(va, vb, w + 1);
      outv[w + 1] = r;
      cnt += (uint32_t)__popcll(r);
    }
  }
  if (vcount) {  // uniform across the block: every lane takes part in the shuffles and the barrier
    __shared__ uint32_t wcnt[AGPU_BLOCK / AGPU_WAVE];
#pragma unroll
    for (int off = AGPU_WAVE / 2; off > 0; off >>= 1) cnt += (uint32_t)__shfl_down((int)cnt, off);
    __syncthreads();  // a grid-stride block may come through here twice: the previous round's reads of wcnt are over
    if ((threadIdx.x & (AGPU_WAVE - 1)) == 0) wcnt[threadIdx.x / AGPU_WAVE] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) {
      uint32_t s = 0;
#pragma unroll
      for (int k = 0; k < AGPU_BLOCK / AGPU_WAVE; k++) s += wcnt[k];
      vcount[vblk] = s;
    }
  }
}

// ---------------------------------------------------------------- variant 0: ballot
// R = rounds per wave tile (R output u64 words per wave).  R = 4 with 256-thread blocks (1024 rows per block) measured
// best at 1e9 rows (6.07 TB/s vs 5.89 at R = 16: profiles/r01_sweep_add_eq_1e9_b.json; re-checked in round 2 with the
// nontemporal result stores, lucky / unlucky allocation: R = 4 0.89 / 0.85, R = 8 0.875 / 0.80, R = 2 0.73 / 0.685).  Full
// tiles only; the remainder is a separate tiny launch of cmp_word_kernel.
#define CMP_R 4
template <typename T, int OP, bool NT>
__global__ __launch_bounds__(AGPU_BLOCK) void cmp_ballot_kernel(const T* a, const T* b, const uint64_t* va,
                                                               const uint64_t* vb, uint64_t* out, uint64_t* outv,
                                                               uint64_t ntiles, uint64_t nvb, int vec16, uint32_t* vcount) {
  constexpr int R = CMP_R;
  constexpr uint64_t WAVE_TILE = (uint64_t)AGPU_WAVE * R;
  constexpr uint64_t TILE = WAVE_TILE * (AGPU_BLOCK / AGPU_WAVE);
  const uint32_t lane = threadIdx.x & (AGPU_WAVE - 1);
  const uint32_t wave = threadIdx.x / AGPU_WAVE;

  for (uint64_t v = blockIdx.x; v < nvb + ntiles; v += gridDim.x) {
    if (v < nvb) {  // validity words of the tiled rows: [0, ntiles * TILE / 64); interleaving these blocks with the
      validity_block(va, vb, outv, v, ntiles * (TILE / 64), vec16 != 0, vcount);  // tiles (every 33rd) measured 3 % slower
      continue;
    }
    const uint64_t t = v - nvb;
    const uint64_t e0 = t * TILE + wave * WAVE_TILE;
    const uint64_t w0 = e0 / 64;
    T xa[R], xb[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
      xa[r] = ld_vec<NT, T>(a + e0 + (uint64_t)r * AGPU_WAVE + lane);
      xb[r] = ld_vec<NT, T>(b + e0 + (uint64_t)r * AGPU_WAVE + lane);
    }
    uint64_t word = 0;
#pragma unroll
    for (int r = 0; r < R; r++) {
      const uint64_t m = __ballot(cmp_pred<OP, T>(xa[r], xb[r]));
      if (lane == (uint32_t)r) word = m;
    }
#if AGPU_CMP_OUT_NT
    if (lane < R) __builtin_nontemporal_store(word, out + w0 + lane);
#else
    if (lane < R) out[w0 + lane] = word;
#endif
  }
}

// ---------------------------------------------------------------- variant 1: 16-byte vector loads + lane-group OR
template <typename T, int N>
struct CmpPack {
  T v[N];
};

template <typename T, int OP, int U, bool NT>
__global__ __launch_bounds__(AGPU_BLOCK) void cmp_vec_kernel(const T* a, const T* b, const uint32_t* va,
                                                            const uint32_t* vb, uint32_t* out, uint32_t* outv,
                                                            uint64_t ntiles, uint64_t nvb, int vec16, uint32_t* vcount) {
  constexpr int N = 16 / sizeof(T);  // rows per lane per vector: 4, 8 or 16
  constexpr int G = 32 / N;          // lanes per 32-bit output word: 8, 4 or 2
  constexpr uint64_t TILE_PACKS = (uint64_t)AGPU_BLOCK * U;
  const uint32_t lane = threadIdx.x & (AGPU_WAVE - 1);

  for (uint64_t vid = blockIdx.x; vid < nvb + ntiles; vid += gridDim.x) {
    if (vid < nvb) {  // validity words of the tiled rows: [0, ntiles * TILE_PACKS * N / 64)
      validity_block(reinterpret_cast<const uint64_t*>(va), reinterpret_cast<const uint64_t*>(vb),
                     reinterpret_cast<uint64_t*>(outv), vid, ntiles * (TILE_PACKS * N / 64), vec16 != 0, vcount);
      continue;
    }
    const uint64_t t = vid - nvb;
    const uint64_t p0 = t * TILE_PACKS + threadIdx.x;
    CmpPack<T, N> xa[U], xb[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const uint64_t pk = p0 + (uint64_t)u * AGPU_BLOCK;
      const u32x4 ra = ld_vec<NT, u32x4>(reinterpret_cast<const u32x4*>(a + pk * N));
      const u32x4 rb = ld_vec<NT, u32x4>(reinterpret_cast<const u32x4*>(b + pk * N));
      xa[u] = __builtin_bit_cast(CmpPack<T, N>, ra);
      xb[u] = __builtin_bit_cast(CmpPack<T, N>, rb);
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      const uint64_t pk = p0 + (uint64_t)u * AGPU_BLOCK;
      uint32_t m = 0;
#pragma unroll
      for (int k = 0; k < N; k++) m |= (uint32_t)cmp_pred<OP, T>(xa[u].v[k], xb[u].v[k]) << k;
      uint32_t v = m << (N * (lane % G));
#pragma unroll
      for (int s = 1; s < G; s <<= 1) v |= (uint32_t)__shfl_xor((int)v, s);
      if (lane % G == 0) {
        const uint64_t w = pk / G;
#if AGPU_CMP_OUT_NT
        __builtin_nontemporal_store(v, out + w);
#else
        out[w] = v;
#endif
      }
    }
  }
}

// one guarded 64-bit output word per thread, words [first_word, ceil(n/64)): the tail of the tiled kernels and the
// element-granular fallback for inputs that are not 16-byte aligned (whole 8-byte padded bitmap is written)
template <typename T, int OP>
__global__ __launch_bounds__(AGPU_BLOCK) void cmp_word_kernel(const T* a, const T* b, const uint64_t* va,
                                                             const uint64_t* vb, uint64_t* out, uint64_t* outv,
                                                             uint64_t first_word, uint64_t n) {
  const uint64_t nwords = (n + 63) / 64;
  for (uint64_t w = first_word + (uint64_t)blockIdx.x * AGPU_BLOCK + threadIdx.x; w < nwords;
       w += (uint64_t)gridDim.x * AGPU_BLOCK) {
    out[w] = cmp_word_guarded<OP, T>(a, b, w, n);
    if (outv) outv[w] = validity_word(va, vb, w);
  }
}

template <typename T, int OP, int U>
static uint64_t launch_cmp_vec(agpu_pipeline* p, const T* pa, const T* pb, const void* va, const void* vb, void* out,
                               void* outv, uint64_t n, bool nt, int vec16, uint32_t* vcount, uint64_t* n_vcount) {
  constexpr int N = 16 / sizeof(T);
  constexpr uint64_t TILE = (uint64_t)AGPU_BLOCK * U * N;
  const uint64_t ntiles = n / TILE;
  if (ntiles) {
    const uint64_t nvb = outv ? (ntiles * (TILE / 64) + CMP_VBLOCK_WORDS - 1) / CMP_VBLOCK_WORDS : 0;
    *n_vcount = nvb;
    const int grid = stream_grid_for(p, ntiles + nvb);
    auto k = nt ? cmp_vec_kernel<T, OP, U, true> : cmp_vec_kernel<T, OP, U, false>;
    hipLaunchKernelGGL(k, dim3(grid), dim3(AGPU_BLOCK), 0, p->stream, pa, pb, static_cast<const uint32_t*>(va),
                       static_cast<const uint32_t*>(vb), static_cast<uint32_t*>(out), static_cast<uint32_t*>(outv), ntiles,
                       nvb, vec16, vcount);
  }
  return ntiles * TILE;
}

agpu_status agpu_internal_count_fold(agpu_pipeline* p, const uint32_t* partials, uint64_t m, const void* bits,
                                     uint64_t tail_first_word, uint64_t n_bits, bool sub_padding, bool complement, uint64_t* out_dev);  // bitmap.hip

// vblocks_for: validity blocks a tiled launch over `rows` rows will run (the by-product count needs its scratch up front)
static uint64_t vcount_slots_for(uint64_t rows) {
  return (rows / 64 + CMP_VBLOCK_WORDS - 1) / CMP_VBLOCK_WORDS + 4;
}

template <typename T, int OP>
static agpu_status launch_cmp(agpu_pipeline* p, const void* a, const void* b, const void* va, const void* vb,
                              void* out, void* outv, uint64_t n, uint64_t* out_null_count) {
  const T* pa = static_cast<const T*>(a);
  const T* pb = static_cast<const T*>(b);
  const bool use_ballot = sizeof(T) == 4 && p->tune.cmp_variant == 0;
  const bool nt = true;  // inputs are read exactly once: nontemporal loads
  const uint64_t* va64 = static_cast<const uint64_t*>(va);
  const uint64_t* vb64 = static_cast<const uint64_t*>(vb);
  uint64_t* out64 = static_cast<uint64_t*>(out);
  uint64_t* outv64 = static_cast<uint64_t*>(outv);
  uint64_t done_rows = 0;  // rows covered by the tiled kernel (a multiple of 64)
  const int vec16 = (!va || aligned16(va)) && (!vb || aligned16(vb)) && (!outv || aligned16(outv));
  uint32_t* vcount = nullptr;  // per-wave set-bit counts of the validity words the tiled kernel stores
  uint64_t n_vcount = 0;
  if (out_null_count && outv) {
    void* scratch = nullptr;
    agpu_status st = agpu_scratch(p, vcount_slots_for(n) * sizeof(uint32_t), &scratch);
    if (st != AGPU_OK) return st;
    vcount = static_cast<uint32_t*>(scratch);
  }
  if (use_ballot) {
    constexpr uint64_t TILE = (uint64_t)AGPU_WAVE * CMP_R * (AGPU_BLOCK / AGPU_WAVE);
    const uint64_t ntiles = n / TILE;
    if (ntiles) {
      const uint64_t nvb = outv ? (ntiles * (TILE / 64) + CMP_VBLOCK_WORDS - 1) / CMP_VBLOCK_WORDS : 0;
      n_vcount = nvb;
      const int grid = stream_grid_for(p, ntiles + nvb);
      auto k = nt ? cmp_ballot_kernel<T, OP, true> : cmp_ballot_kernel<T, OP, false>;
      hipLaunchKernelGGL(k, dim3(grid), dim3(AGPU_BLOCK), 0, p->stream, pa, pb, va64, vb64, out64, outv64, ntiles, nvb,
                         vec16, vcount);
    }
    done_rows = ntiles * TILE;
  } else if (aligned16(a) && aligned16(b)) {
    // packs per lane and array in flight: TWO (round 3, tools/probe/narrow_tunings.py, one process: u8 eq 0.755–0.786
    // of the roof with one pack, 0.81–0.82 with two, 0.80–0.81 with four; u16 lt 0.81–0.84 / 0.84–0.85 / 0.84)
    done_rows = launch_cmp_vec<T, OP, 2>(p, pa, pb, va, vb, out, outv, n, nt, vec16, vcount, &n_vcount);
  }
  if (done_rows < n) {
    const uint64_t first_word = done_rows / 64, nwords = (n + 63) / 64;
    const int grid = stream_grid_for(p, (nwords - first_word + AGPU_BLOCK - 1) / AGPU_BLOCK);
    hipLaunchKernelGGL((cmp_word_kernel<T, OP>), dim3(grid), dim3(AGPU_BLOCK), 0, p->stream, pa, pb, va64, vb64, out64,
                       outv64, first_word, n);
  }
  AGPU_LAUNCH_CHECK();
  if (out_null_count) {
    if (!outv) {  // (None, None) → None: no nulls
      AGPU_HIP(hipMemsetAsync(out_null_count, 0, sizeof(uint64_t), p->stream));
      return AGPU_OK;
    }
    // the stream-grid may be smaller than the work (tuning): the kernels are grid-stride over VIRTUAL block ids, so the
    // slots [0, n_vcount) are all written either way; words ≥ done_rows / 64 came from the tail kernel and are the fold's
    return agpu_internal_count_fold(p, vcount, n_vcount, outv, done_rows / 64, n, false, true, out_null_count);
  }
  return AGPU_OK;
}

template <typename T>
static agpu_status dispatch_cmp_op(agpu_pipeline* p, agpu_cmp_op op, const void* a, const void* b, const void* va,
                                   const void* vb, void* out, void* outv, uint64_t n, uint64_t* nc) {
  switch (op) {
    case AGPU_CMP_GT: return launch_cmp<T, AGPU_CMP_GT>(p, a, b, va, vb, out, outv, n, nc);
    case AGPU_CMP_GTEQ: return launch_cmp<T, AGPU_CMP_GTEQ>(p, a, b, va, vb, out, outv, n, nc);
    case AGPU_CMP_LT: return launch_cmp<T, AGPU_CMP_LT>(p, a, b, va, vb, out, outv, n, nc);
    case AGPU_CMP_LTEQ: return launch_cmp<T, AGPU_CMP_LTEQ>(p, a, b, va, vb, out, outv, n, nc);
    case AGPU_CMP_EQ: return launch_cmp<T, AGPU_CMP_EQ>(p, a, b, va, vb, out, outv, n, nc);
    default: break;
  }
  agpu_set_error("bad compare op %d", (int)op);
  return AGPU_ERR_ARG;
}

static agpu_status compare_impl(agpu_pipeline* p, agpu_cmp_op op, agpu_dtype dtype, const void* a, const void* b,
                                const void* va, const void* vb, void* out, void* outv, uint64_t n, uint64_t* nc = nullptr) {
  AGPU_BIND_AS(p, outv ? "agpu_compare_validity" : "agpu_compare");
  if (n == 0) {
    if (nc) AGPU_HIP(hipMemsetAsync(nc, 0, sizeof(uint64_t), p->stream));
    return AGPU_OK;
  }
  AGPU_REQUIRE(a && b && out, AGPU_ERR_ARG, "null pointer");
  AGPU_REQUIRE(aligned_to(out, 8) && (!outv || aligned_to(outv, 8)) && (!va || aligned_to(va, 8)) &&
                   (!vb || aligned_to(vb, 8)),
               AGPU_ERR_SHAPE, "bitmaps must be 8-byte aligned");
  if (!va && !vb) outv = nullptr;  // (None, None) → None [null_bit_buffer.rs:211]
  AGPU_REQUIRE(!(va || vb) || outv, AGPU_ERR_ARG, "out_validity required when an input validity is given");
  switch (dtype) {
    case AGPU_F32: return dispatch_cmp_op<float>(p, op, a, b, va, vb, out, outv, n, nc);
    case AGPU_I32: case AGPU_DATE32: return dispatch_cmp_op<int32_t>(p, op, a, b, va, vb, out, outv, n, nc);
    case AGPU_U32: return dispatch_cmp_op<uint32_t>(p, op, a, b, va, vb, out, outv, n, nc);
    case AGPU_I16: return dispatch_cmp_op<int16_t>(p, op, a, b, va, vb, out, outv, n, nc);
    case AGPU_U16: return dispatch_cmp_op<uint16_t>(p, op, a, b, va, vb, out, outv, n, nc);
    case AGPU_I8: return dispatch_cmp_op<int8_t>(p, op, a, b, va, vb, out, outv, n, nc);
    case AGPU_U8: return dispatch_cmp_op<uint8_t>(p, op, a, b, va, vb, out, outv, n, nc);
    default: break;
  }
  agpu_set_error("dtype %d not supported for compare", (int)dtype);
  return AGPU_ERR_UNSUPPORTED;
}

extern "C" {

agpu_status agpu_compare(agpu_pipeline* p, agpu_cmp_op op, agpu_dtype dtype, const void* a, const void* b,
                         void* out_bits, uint64_t n) {
  return compare_impl(p, op, dtype, a, b, nullptr, nullptr, out_bits, nullptr, n);
}

agpu_status agpu_compare_validity(agpu_pipeline* p, agpu_cmp_op op, agpu_dtype dtype, const void* a, const void* b,
                                  const void* va, const void* vb, void* out_bits, void* out_validity, uint64_t n) {
  return compare_impl(p, op, dtype, a, b, va, vb, out_bits, out_validity, n);
}

agpu_status agpu_compare_validity_count(agpu_pipeline* p, agpu_cmp_op op, agpu_dtype dtype, const void* a, const void* b,
                                        const void* va, const void* vb, void* out_bits, void* out_validity, uint64_t n,
                                        uint64_t* out_null_count_dev) {
  return compare_impl(p, op, dtype, a, b, va, vb, out_bits, out_validity, n, out_null_count_dev);
}

}  // extern "C"
