// bitmap.hip — validity / Boolean bitmap kernels: and/or/xor/not, merge, popcount, any.
//
// Replaces crates/logical/compute_shaders/u32/{logical,not,any,countbitones}.wgsl as used by
// NullBitBufferGpu::merge_null_bit_buffer (crates/array/src/array/null_bit_buffer.rs:168-243 — the validity AND),
// BooleanArrayGPU's Logical/LogicalContains impls (crates/logical/src/boolean.rs:18-147) and
// merge_null_buffers_op (crates/routines/src/merge.rs:17-86, routines/compute_shaders/u32/merge_null_buffer.wgsl,
// bool/merge.wgsl).
//
// MI355X design: bitmaps are 1/32 of the column traffic, so these are short HBM-bound streams: 16-byte vector
// loads/stores per lane, grid-stride; popcount uses v_bcnt on 64-bit words + a wave shuffle reduce + ONE integer
// atomic per block (integer adds are order-independent ⇒ deterministic).  The reference's 4-dispatch validity merge
// for `merge` is one fused kernel here.
#include "common.hpp"

enum { BM_AND = 0, BM_OR = 1, BM_XOR = 2, BM_NOT = 3, BM_SELECT = 4, BM_MERGE_VALIDITY = 5, BM_ANDNOT = 6 };

template <int OP>
__device__ __forceinline__ uint64_t bm_apply(uint64_t a, uint64_t b, uint64_t c, uint64_t d) {
  if constexpr (OP == BM_AND) return a & b;
  else if constexpr (OP == BM_OR) return a | b;
  else if constexpr (OP == BM_XOR) return a ^ b;
  else if constexpr (OP == BM_NOT) return ~a;
  else if constexpr (OP == BM_ANDNOT) return a & ~b;
  else if constexpr (OP == BM_SELECT) return (a & c) | (b & ~c);          // c = mask
  else return ((a & c) | (b & ~c)) & d;                                    // d = mask validity
}

typedef uint64_t u64x2 __attribute__((ext_vector_type(2)));

// ---------------------------------------------------------------- set-bit counts as a BY-PRODUCT (north_star: "wavefront
// ballot/popc for null counts").  A kernel that produces a bitmap has every word of it in a register once: it adds
// v_bcnt of what it stores, reduces inside the wave (shuffles, no LDS) and leaves ONE u32 per wave in a scratch array —
// no same-address atomics (they serialise at ≈ 12 ns each) — and a single small block folds the array, adds whatever
// words a tail kernel wrote, masks the padding bits of the last word and writes the u64 result.  The reference needs a
// second pass over the bitmap for this (countob + Sum [ref: crates/logical/src/boolean.rs:120-146]); here the count costs
// one ~3 µs launch and no bitmap traffic.
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t c) {
#pragma unroll
  for (int off = AGPU_WAVE / 2; off > 0; off >>= 1) c += (uint32_t)__shfl_down((int)c, off);
  return c;  // valid in lane 0
}

// out[0] = Σ partials[0..m) + popcount(bits words [tail_first_word, n_words), last word masked to n_bits)
//          − (sub_padding ? popcount(padding bits of the last word of `bits`) : 0);   complement: out = n_bits − that
#define AGPU_FOLD_BLOCK 1024
__global__ __launch_bounds__(AGPU_FOLD_BLOCK) void count_fold_kernel(const uint32_t* partials, uint64_t m, const uint64_t* bits,
                                                                   uint64_t tail_first_word, uint64_t n_bits, int sub_padding,
                                                                   int complement, uint64_t* out) {
  const uint64_t n_words = (n_bits + 63) / 64;
  uint64_t c = 0;
  // ONE block walks the whole array, so the loads must not queue behind each other: 16-byte loads, eight in flight per
  // lane (a dependent one-word-per-iteration loop over 122 000 per-wave partials took 30 µs — longer than the kernel it served)
  const uint64_t m4 = m / 4;  // the scratch block is 256-byte aligned
  const u32x4* p4 = reinterpret_cast<const u32x4*>(partials);
  for (uint64_t i0 = threadIdx.x; i0 < m4; i0 += 8 * AGPU_FOLD_BLOCK) {
    u32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const uint64_t i = i0 + (uint64_t)u * AGPU_FOLD_BLOCK;
      v[u] = i < m4 ? p4[i] : u32x4{0, 0, 0, 0};
    }
#pragma unroll
    for (int u = 0; u < 8; u++) c += (uint64_t)v[u].x + v[u].y + v[u].z + v[u].w;
  }
  for (uint64_t i = m4 * 4 + threadIdx.x; i < m; i += AGPU_FOLD_BLOCK) c += partials[i];
  for (uint64_t w = tail_first_word + threadIdx.x; w < n_words; w += AGPU_FOLD_BLOCK) {
    uint64_t v = bits[w];
    if (w == n_words - 1 && (n_bits & 63)) v &= (1ull << (n_bits & 63)) - 1ull;
    c += (uint64_t)__popcll(v);
  }
#pragma unroll
  for (int off = AGPU_WAVE / 2; off > 0; off >>= 1) c += __shfl_down(c, off);
  __shared__ uint64_t wsum[AGPU_FOLD_BLOCK / AGPU_WAVE];
  if ((threadIdx.x & (AGPU_WAVE - 1)) == 0) wsum[threadIdx.x / AGPU_WAVE] = c;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint64_t s = 0;
    for (int k = 0; k < AGPU_FOLD_BLOCK / AGPU_WAVE; k++) s += wsum[k];
    if (sub_padding && (n_bits & 63) && tail_first_word >= n_words) s -= (uint64_t)__popcll(bits[n_words - 1] >> (n_bits & 63));
    out[0] = complement ? n_bits - s : s;
  }
}
// compare.hip / bitmap.hip: fold the per-wave partials (see above)
agpu_status agpu_internal_count_fold(agpu_pipeline* p, const uint32_t* partials, uint64_t m, const void* bits,
                                     uint64_t tail_first_word, uint64_t n_bits, bool sub_padding, bool complement, uint64_t* out_dev) {
  hipLaunchKernelGGL(count_fold_kernel, dim3(1), dim3(AGPU_FOLD_BLOCK), 0, p->stream, partials, m, static_cast<const uint64_t*>(bits),
                     tail_first_word, n_bits, sub_padding ? 1 : 0, complement ? 1 : 0, out_dev);
  AGPU_LAUNCH_CHECK();
  return AGPU_OK;
}

// inputs a,b,c,d may be null (treated as all-ones) when NULLABLE; COUNT: partials[blockIdx.x] = set bits this block stored
template <int OP, bool NULLABLE, bool COUNT = false, int BLK = AGPU_BLOCK>
__global__ __launch_bounds__(BLK) void bitmap_kernel(const uint64_t* a, const uint64_t* b, const uint64_t* c,
                                                    const uint64_t* d, uint64_t* out, uint64_t n_words,
                                                    int vec_ok, uint32_t* partials = nullptr) {
  const uint64_t tid = (uint64_t)blockIdx.x * BLK + threadIdx.x;
  const uint64_t stride = (uint64_t)gridDim.x * BLK;
  const uint64_t n_pairs = vec_ok ? n_words / 2 : 0;
  const u64x2 ones = {~0ull, ~0ull};
  uint32_t cnt = 0;
  for (uint64_t i = tid; i < n_pairs; i += stride) {
    u64x2 x = ones, y = ones, z = ones, w = ones;
    if (!NULLABLE || a) x = reinterpret_cast<const u64x2*>(a)[i];
    if (OP != BM_NOT && (!NULLABLE || b)) y = reinterpret_cast<const u64x2*>(b)[i];
    if ((OP == BM_SELECT || OP == BM_MERGE_VALIDITY) && (!NULLABLE || c)) z = reinterpret_cast<const u64x2*>(c)[i];
    if (OP == BM_MERGE_VALIDITY && (!NULLABLE || d)) w = reinterpret_cast<const u64x2*>(d)[i];
    u64x2 r = {bm_apply<OP>(x.x, y.x, z.x, w.x), bm_apply<OP>(x.y, y.y, z.y, w.y)};
    reinterpret_cast<u64x2*>(out)[i] = r;
    if constexpr (COUNT) cnt += (uint32_t)__popcll(r.x) + (uint32_t)__popcll(r.y);
  }
  for (uint64_t i = n_pairs * 2 + tid; i < n_words; i += stride) {
    const uint64_t x = (!NULLABLE || a) ? a[i] : ~0ull;
    const uint64_t y = (OP != BM_NOT && (!NULLABLE || b)) ? b[i] : ~0ull;
    const uint64_t z = ((OP == BM_SELECT || OP == BM_MERGE_VALIDITY) && (!NULLABLE || c)) ? c[i] : ~0ull;
    const uint64_t w = (OP == BM_MERGE_VALIDITY && (!NULLABLE || d)) ? d[i] : ~0ull;
    const uint64_t r = bm_apply<OP>(x, y, z, w);
    out[i] = r;
    if constexpr (COUNT) cnt += (uint32_t)__popcll(r);
  }
  if constexpr (COUNT) {  // every lane reaches this point: no divergence around the shuffles / the barrier
    __shared__ uint32_t wcnt[BLK / AGPU_WAVE];
    cnt = wave_sum_u32(cnt);
    if ((threadIdx.x & (AGPU_WAVE - 1)) == 0) wcnt[threadIdx.x / AGPU_WAVE] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) {
      uint32_t s = 0;
#pragma unroll
      for (int k = 0; k < BLK / AGPU_WAVE; k++) s += wcnt[k];
      partials[blockIdx.x] = s;
    }
  }
}

template <int OP, bool NULLABLE>
static agpu_status launch_bitmap(agpu_pipeline* p, const void* a, const void* b, const void* c, const void* d,
                                 void* out, uint64_t n_bits, uint64_t* out_count_dev = nullptr) {
  AGPU_BIND_AS(p, OP == BM_MERGE_VALIDITY ? "agpu_bitmap_merge_validity" : OP == BM_SELECT ? "agpu_merge_bits"
                  : OP == BM_NOT ? "agpu_bitmap_not" : "agpu_bitmap_binary");
  if (n_bits == 0) {
    if (out_count_dev) AGPU_HIP(hipMemsetAsync(out_count_dev, 0, sizeof(uint64_t), p->stream));
    return AGPU_OK;
  }
  AGPU_REQUIRE(out, AGPU_ERR_ARG, "null output");
  const void* ptrs[5] = {a, b, c, d, out};
  int vec_ok = 1;
  for (const void* q : ptrs) {
    if (!q) continue;
    AGPU_REQUIRE(aligned_to(q, 8), AGPU_ERR_SHAPE, "bitmaps must be 8-byte aligned");
    if (!aligned16(q)) vec_ok = 0;
  }
  const uint64_t n_words = (n_bits + 63) / 64;
  const int grid = stream_grid_for(p, (n_words / 2 + AGPU_BLOCK - 1) / AGPU_BLOCK);
  if (out_count_dev) {
    const uint64_t m = (uint64_t)grid;
    void* scratch = nullptr;
    agpu_status st = agpu_scratch(p, m * sizeof(uint32_t), &scratch);
    if (st != AGPU_OK) return st;
    hipLaunchKernelGGL((bitmap_kernel<OP, NULLABLE, true>), dim3(grid), dim3(AGPU_BLOCK), 0, p->stream,
                       static_cast<const uint64_t*>(a), static_cast<const uint64_t*>(b), static_cast<const uint64_t*>(c),
                       static_cast<const uint64_t*>(d), static_cast<uint64_t*>(out), n_words, vec_ok, static_cast<uint32_t*>(scratch));
    AGPU_LAUNCH_CHECK();
    // the kernel counted every bit of every word it stored: take the padding bits of the last word out again
    return agpu_internal_count_fold(p, static_cast<const uint32_t*>(scratch), m, out, n_words, n_bits, true, false, out_count_dev);
  }
  // two- to four-input ops in one-wave blocks (the element-wise kernels' shape), `not` in 256-thread blocks: at 125 MB per bitmap,
  // alternating runs of tools/kernel_table.py on one box — AND 0.725–0.740 → 0.749–0.755 of the roof, merge validity 0.885–0.905 →
  // 0.915–0.936, not 0.834–0.841 → 0.818–0.833 (AGPU_BITMAP_BLOCK = 64 / 256 forces one shape for the A/B)
  static const int forced = [] { const char* e = getenv("AGPU_BITMAP_BLOCK"); return e ? atoi(e) : 0; }();
  const int blk = forced == 64 || forced == 256 ? forced : (OP == BM_NOT ? 256 : 64);
  if (blk == 64) {
    const int g64 = stream_grid_for(p, (n_words / 2 + 63) / 64);
    hipLaunchKernelGGL((bitmap_kernel<OP, NULLABLE, false, 64>), dim3(g64), dim3(64), 0, p->stream,
                       static_cast<const uint64_t*>(a), static_cast<const uint64_t*>(b), static_cast<const uint64_t*>(c),
                       static_cast<const uint64_t*>(d), static_cast<uint64_t*>(out), n_words, vec_ok, (uint32_t*)nullptr);
    AGPU_LAUNCH_CHECK();
    return AGPU_OK;
  }
  hipLaunchKernelGGL((bitmap_kernel<OP, NULLABLE>), dim3(grid), dim3(AGPU_BLOCK), 0, p->stream,
                     static_cast<const uint64_t*>(a), static_cast<const uint64_t*>(b), static_cast<const uint64_t*>(c),
                     static_cast<const uint64_t*>(d), static_cast<uint64_t*>(out), n_words, vec_ok, (uint32_t*)nullptr);
  AGPU_LAUNCH_CHECK();
  return AGPU_OK;
}

// ---------------------------------------------------------------- stand-alone popcount: one-wave blocks + fold
// The atomic form below (≤ 4 blocks per CU, one same-address atomic per block, a memset in front) took 46.6 µs for a
// 125 MB bitmap — 0.34 of the HBM roof, twice as long as `bitmap not` needs to read AND write as much.  One-wave blocks
// over 16 KiB chunks (sixteen 16-byte loads per lane, eight in flight) + the fold launch: no atomics, no memset.
constexpr uint64_t POPC_CHUNK_WORDS = 2048;  // 16 KiB
__global__ __launch_bounds__(AGPU_WAVE) void popcount_wave_kernel(const uint64_t* bits, uint32_t* partials, uint64_t nchunks) {
  const uint32_t lane = threadIdx.x;
  for (uint64_t c = blockIdx.x; c < nchunks; c += gridDim.x) {
    const u64x2* base = reinterpret_cast<const u64x2*>(bits + c * POPC_CHUNK_WORDS) + lane;
    uint32_t cnt = 0;
#pragma unroll
    for (int j0 = 0; j0 < 16; j0 += 8) {
      u64x2 v[8];
#pragma unroll
      for (int u = 0; u < 8; u++) v[u] = __builtin_nontemporal_load(base + (j0 + u) * AGPU_WAVE);
#pragma unroll
      for (int u = 0; u < 8; u++) cnt += (uint32_t)__popcll(v[u].x) + (uint32_t)__popcll(v[u].y);
    }
    cnt = wave_sum_u32(cnt);
    if (lane == 0) partials[c] = cnt;
  }
}

// ---------------------------------------------------------------- popcount / any over the first n_bits
template <bool ANY>
__global__ __launch_bounds__(AGPU_BLOCK) void popcount_kernel(const uint64_t* bits, uint64_t n_bits,
                                                             unsigned long long* out_count, uint32_t* out_any) {
  const uint64_t n_full = n_bits / 64;
  const uint64_t tid = (uint64_t)blockIdx.x * AGPU_BLOCK + threadIdx.x;
  const uint64_t stride = (uint64_t)gridDim.x * AGPU_BLOCK;
  uint64_t c = 0;
  auto count = [](uint64_t w) { return ANY ? (uint64_t)(w != 0) : (uint64_t)__popcll(w); };
  // read-only and grid-capped (one atomic per block): keep 4 × 16 bytes per lane in flight or the kernel is bound by
  // latency, not HBM (0.037 → 0.02x ms on a 125 MB bitmap)
  const uint64_t n_pairs = ((reinterpret_cast<uintptr_t>(bits) & 15u) == 0) ? n_full / 2 : 0;
  const u64x2* b2 = reinterpret_cast<const u64x2*>(bits);
  uint64_t i = tid;
  for (; i + 3 * stride < n_pairs; i += 4 * stride) {
    const u64x2 w0 = __builtin_nontemporal_load(b2 + i), w1 = __builtin_nontemporal_load(b2 + i + stride),
                w2 = __builtin_nontemporal_load(b2 + i + 2 * stride), w3 = __builtin_nontemporal_load(b2 + i + 3 * stride);
    c += count(w0.x) + count(w0.y) + count(w1.x) + count(w1.y) + count(w2.x) + count(w2.y) + count(w3.x) + count(w3.y);
  }
  for (; i < n_pairs; i += stride) {
    const u64x2 w = b2[i];
    c += count(w.x) + count(w.y);
  }
  for (uint64_t j = n_pairs * 2 + tid; j < n_full; j += stride) c += count(bits[j]);
  if (tid == 0 && (n_bits & 63)) {
    const uint64_t w = bits[n_full] & ((1ull << (n_bits & 63)) - 1ull);
    c += ANY ? (uint64_t)(w != 0) : (uint64_t)__popcll(w);
  }
  // wave reduce (64 lanes), then one atomic per wave-leader via LDS block reduce
#pragma unroll
  for (int off = AGPU_WAVE / 2; off > 0; off >>= 1) c += __shfl_down(c, off);
  __shared__ uint64_t wsum[AGPU_BLOCK / AGPU_WAVE];
  const uint32_t lane = threadIdx.x & (AGPU_WAVE - 1), wave = threadIdx.x / AGPU_WAVE;
  if (lane == 0) wsum[wave] = c;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint64_t s = 0;
    for (int k = 0; k < AGPU_BLOCK / AGPU_WAVE; k++) s += wsum[k];
    if (ANY) {
      if (s) atomicOr(out_any, 1u);
    } else if (s) {
      atomicAdd(out_count, (unsigned long long)s);
    }
  }
}

// out word w = src bits [off + 64w, off + 64w + 64): funnel shift of two source words; padding bits zeroed
__global__ __launch_bounds__(AGPU_BLOCK) void bitmap_copy_bits_kernel(const uint64_t* src, uint64_t bit_off, uint64_t* out,
                                                                     uint64_t n_bits) {
  const uint64_t n_words = (n_bits + 63) / 64;
  const uint64_t w0 = bit_off / 64;
  const uint32_t sh = (uint32_t)(bit_off & 63);
  const uint64_t last_src_word = (bit_off + n_bits - 1) / 64;  // never read past the word holding the last bit
  for (uint64_t w = (uint64_t)blockIdx.x * AGPU_BLOCK + threadIdx.x; w < n_words; w += (uint64_t)gridDim.x * AGPU_BLOCK) {
    uint64_t v = src[w0 + w] >> sh;
    if (sh && w0 + w + 1 <= last_src_word) v |= src[w0 + w + 1] << (64 - sh);
    const uint64_t remaining = n_bits - w * 64;
    if (remaining < 64) v &= (1ull << remaining) - 1ull;
    out[w] = v;
  }
}

// merge_not_selected of routines/compute_shaders/u32/merge_null_buffer.wgsl (only reachable through by_name)
agpu_status agpu_bitmap_andnot_internal(agpu_pipeline* p, const void* a, const void* b, void* out, uint64_t n_bits) {
  AGPU_REQUIRE(n_bits == 0 || (a && b), AGPU_ERR_ARG, "null input");
  return launch_bitmap<BM_ANDNOT, false>(p, a, b, nullptr, nullptr, out, n_bits);
}

extern "C" {

agpu_status agpu_bitmap_binary(agpu_pipeline* p, agpu_binary_op op, const void* a, const void* b, void* out,
                               uint64_t n_bits) {
  AGPU_REQUIRE(n_bits == 0 || (a && b), AGPU_ERR_ARG, "null input");
  switch (op) {
    case AGPU_OP_AND: return launch_bitmap<BM_AND, false>(p, a, b, nullptr, nullptr, out, n_bits);
    case AGPU_OP_OR: return launch_bitmap<BM_OR, false>(p, a, b, nullptr, nullptr, out, n_bits);
    case AGPU_OP_XOR: return launch_bitmap<BM_XOR, false>(p, a, b, nullptr, nullptr, out, n_bits);
    default: break;
  }
  agpu_set_error("bitmap op %d not supported (and/or/xor only)", (int)op);
  return AGPU_ERR_UNSUPPORTED;
}

agpu_status agpu_bitmap_copy_bits(agpu_pipeline* p, const void* src, uint64_t src_bit_offset, void* out, uint64_t n_bits) {
  AGPU_BIND(p);
  if (n_bits == 0) return AGPU_OK;
  AGPU_REQUIRE(src && out, AGPU_ERR_ARG, "null pointer");
  AGPU_REQUIRE(aligned_to(src, 8) && aligned_to(out, 8), AGPU_ERR_SHAPE, "bitmaps must be 8-byte aligned");
  const uint64_t n_words = (n_bits + 63) / 64;
  const int grid = stream_grid_for(p, (n_words + AGPU_BLOCK - 1) / AGPU_BLOCK);
  hipLaunchKernelGGL(bitmap_copy_bits_kernel, dim3(grid), dim3(AGPU_BLOCK), 0, p->stream, static_cast<const uint64_t*>(src),
                     src_bit_offset, static_cast<uint64_t*>(out), n_bits);
  AGPU_LAUNCH_CHECK();
  return AGPU_OK;
}

agpu_status agpu_bitmap_not(agpu_pipeline* p, const void* in, void* out, uint64_t n_bits) {
  AGPU_REQUIRE(n_bits == 0 || in, AGPU_ERR_ARG, "null input");
  return launch_bitmap<BM_NOT, false>(p, in, nullptr, nullptr, nullptr, out, n_bits);
}

agpu_status agpu_merge_bits(agpu_pipeline* p, const void* a, const void* b, const void* mask_bits, void* out,
                            uint64_t n_bits) {
  AGPU_REQUIRE(n_bits == 0 || (a && b && mask_bits), AGPU_ERR_ARG, "null input");
  return launch_bitmap<BM_SELECT, false>(p, a, b, mask_bits, nullptr, out, n_bits);
}

agpu_status agpu_bitmap_merge_validity(agpu_pipeline* p, const void* va, const void* vb, const void* mask,
                                       const void* vmask, void* out, uint64_t n_bits) {
  AGPU_REQUIRE(mask, AGPU_ERR_ARG, "null mask");
  AGPU_REQUIRE(va || vb || vmask, AGPU_ERR_ARG, "all validity inputs are null: the result is None, nothing to compute");
  return launch_bitmap<BM_MERGE_VALIDITY, true>(p, va, vb, mask, vmask, out, n_bits);
}

agpu_status agpu_bitmap_popcount(agpu_pipeline* p, const void* bits, uint64_t n_bits, uint64_t* out_count_dev) {
  AGPU_BIND(p);
  AGPU_REQUIRE(out_count_dev, AGPU_ERR_ARG, "null output");
  if (n_bits == 0) {
    AGPU_HIP(hipMemsetAsync(out_count_dev, 0, sizeof(uint64_t), p->stream));
    return AGPU_OK;
  }
  AGPU_REQUIRE(bits && aligned_to(bits, 8), AGPU_ERR_SHAPE, "bitmap must be 8-byte aligned");
  const uint64_t n_words = (n_bits + 63) / 64;
  const uint64_t n_full = n_bits / 64;  // words counted whole; a partial last word is the fold's
  if (aligned16(bits) && n_full >= 8 * POPC_CHUNK_WORDS) {
    const uint64_t nchunks = n_full / POPC_CHUNK_WORDS;
    void* scratch = nullptr;
    agpu_status st = agpu_scratch(p, nchunks * sizeof(uint32_t), &scratch);
    if (st != AGPU_OK) return st;
    const uint64_t g = nchunks < 0x3FFFFFFFull ? nchunks : 0x3FFFFFFFull;
    hipLaunchKernelGGL(popcount_wave_kernel, dim3((unsigned)g), dim3(AGPU_WAVE), 0, p->stream, static_cast<const uint64_t*>(bits),
                       static_cast<uint32_t*>(scratch), nchunks);
    AGPU_LAUNCH_CHECK();
    return agpu_internal_count_fold(p, static_cast<const uint32_t*>(scratch), nchunks, bits, nchunks * POPC_CHUNK_WORDS, n_bits, false,
                                    false, out_count_dev);
  }
  if (n_words <= 16 * AGPU_FOLD_BLOCK)  // small bitmaps: the fold block alone reads them (one launch, no memset)
    return agpu_internal_count_fold(p, nullptr, 0, bits, 0, n_bits, false, false, out_count_dev);
  AGPU_HIP(hipMemsetAsync(out_count_dev, 0, sizeof(uint64_t), p->stream));
  // ≤ 1024 blocks: every block ends with ONE atomic on the same word, and same-address atomics serialise
  // (15 259 blocks → 0.19 ms for a 125 MB bitmap; 1024 → bandwidth-bound)
  const int grid = atomic_grid_for(p, (n_words + AGPU_BLOCK * 4 - 1) / (AGPU_BLOCK * 4));
  hipLaunchKernelGGL((popcount_kernel<false>), dim3(grid), dim3(AGPU_BLOCK), 0, p->stream,
                     static_cast<const uint64_t*>(bits), n_bits, reinterpret_cast<unsigned long long*>(out_count_dev),
                     (uint32_t*)nullptr);
  AGPU_LAUNCH_CHECK();
  return AGPU_OK;
}

agpu_status agpu_bitmap_binary_count(agpu_pipeline* p, agpu_binary_op op, const void* a, const void* b, void* out, uint64_t n_bits,
                                     uint64_t* out_set_count_dev) {
  AGPU_REQUIRE(n_bits == 0 || (a && b), AGPU_ERR_ARG, "null input");
  switch (op) {
    case AGPU_OP_AND: return launch_bitmap<BM_AND, false>(p, a, b, nullptr, nullptr, out, n_bits, out_set_count_dev);
    case AGPU_OP_OR: return launch_bitmap<BM_OR, false>(p, a, b, nullptr, nullptr, out, n_bits, out_set_count_dev);
    case AGPU_OP_XOR: return launch_bitmap<BM_XOR, false>(p, a, b, nullptr, nullptr, out, n_bits, out_set_count_dev);
    default: break;
  }
  agpu_set_error("bitmap op %d not supported (and/or/xor only)", (int)op);
  return AGPU_ERR_UNSUPPORTED;
}

agpu_status agpu_bitmap_merge_validity_count(agpu_pipeline* p, const void* va, const void* vb, const void* mask, const void* vmask,
                                             void* out, uint64_t n_bits, uint64_t* out_set_count_dev) {
  AGPU_REQUIRE(mask, AGPU_ERR_ARG, "null mask");
  AGPU_REQUIRE(va || vb || vmask, AGPU_ERR_ARG, "all validity inputs are null: the result is None, nothing to compute");
  return launch_bitmap<BM_MERGE_VALIDITY, true>(p, va, vb, mask, vmask, out, n_bits, out_set_count_dev);
}

agpu_status agpu_bitmap_any(agpu_pipeline* p, const void* bits, uint64_t n_bits, uint32_t* out_any_dev) {
  AGPU_BIND(p);
  AGPU_REQUIRE(out_any_dev, AGPU_ERR_ARG, "null output");
  AGPU_HIP(hipMemsetAsync(out_any_dev, 0, sizeof(uint32_t), p->stream));
  if (n_bits == 0) return AGPU_OK;
  AGPU_REQUIRE(bits && aligned_to(bits, 8), AGPU_ERR_SHAPE, "bitmap must be 8-byte aligned");
  const uint64_t n_words = (n_bits + 63) / 64;
  // ≤ 1024 blocks: every block ends with ONE atomic on the same word, and same-address atomics serialise
  // (15 259 blocks → 0.19 ms for a 125 MB bitmap; 1024 → bandwidth-bound)
  const int grid = atomic_grid_for(p, (n_words + AGPU_BLOCK * 4 - 1) / (AGPU_BLOCK * 4));
  hipLaunchKernelGGL((popcount_kernel<true>), dim3(grid), dim3(AGPU_BLOCK), 0, p->stream,
                     static_cast<const uint64_t*>(bits), n_bits, (unsigned long long*)nullptr, out_any_dev);
  AGPU_LAUNCH_CHECK();
  return AGPU_OK;
}

}  // extern "C"
