// swizzle.hip — take (gather), put (scatter), merge (select by mask) over values and bitmaps.
//
// Replaces crates/routines/compute_shaders/{32bit,16bit,8bit,bool}/{take,put,merge}.wgsl and the launch helpers
// apply_take_op (crates/routines/src/take.rs:9-55), apply_put_op (put.rs:9-56), Swizzle::merge_op
// (crates/routines/src/lib.rs:82-120), bool take/put (crates/routines/src/bool.rs:15-128).
// The reference implements take/put for 32-bit values and Boolean only (TAKE_SHADER = "todo!()" for 8/16-bit:
// crates/routines/src/u8.rs:3-7); here widths 1, 2 and 4 share one template.
//
// MI355X design: the index stream and the output stream are coalesced 16-byte accesses; the gather/scatter side is
// element-granular by nature (each 4-byte access pulls a whole 64/128-byte line), so these kernels are bound by
// line fetches, not by algorithmic bytes — DESIGN.md reports them honestly against 12/16 B per row.  Measured
// (profiles/r01_gather_sweep.json): a random 4-byte gather from a 1 GiB column runs at 52 G rows/s and exactly doubles
// when index pairs share a 128-byte line, whichever half or sector they hit ⇒ every miss moves a full 128-byte line and
// 52 G × 128 B = 6.7 TB/s is the HBM roof; 4 / 8 / 16 gathers in flight per lane, one-wave blocks and nontemporal
// gathers change nothing (nt: −7 %).  Sorted indices reach 245–720 G rows/s.  The index and output streams are
// nontemporal so they do not evict a cache-resident source (4 MiB source: 165 → 183 G rows/s).  Bit gathers use the wave ballot exactly like the compare kernel.
#include "common.hpp"

template <int W> struct ElemOf;
template <> struct ElemOf<1> { typedef uint8_t type; };
template <> struct ElemOf<2> { typedef uint16_t type; };
template <> struct ElemOf<4> { typedef uint32_t type; };

template <typename E, int N> struct OutPack { E v[N]; };

// ---------------------------------------------------------------- take
template <typename V>
__device__ __forceinline__ V swz_ld(const V* p, bool nt) { return nt ? __builtin_nontemporal_load(p) : *p; }

// Index ranges are checked IN the kernels (the reference leans on WGSL's robust buffer access: an out-of-range read
// yields 0, an out-of-range write is dropped).  Same result here, plus a sticky bit in the pipeline's pinned error word
// that agpu_pipeline_sync turns into AGPU_ERR_SHAPE — no separate max-reduction pass over the index column and no
// readback before the gather (that pre-check cost a 4 B/row pass and a device sync per take).
template <int W, bool NT>
__global__ __launch_bounds__(AGPU_BLOCK) void take_kernel(const typename ElemOf<W>::type* values, uint64_t n_values,
                                                         const uint32_t* idx, typename ElemOf<W>::type* out, uint64_t n,
                                                         int vec_ok, uint32_t* flags) {
  typedef typename ElemOf<W>::type E;
  constexpr int N = 16 / W;  // output elements per lane (one 16-byte store)
  const uint64_t tid = (uint64_t)blockIdx.x * AGPU_BLOCK + threadIdx.x;
  const uint64_t stride = (uint64_t)gridDim.x * AGPU_BLOCK;
  const uint64_t npacks = vec_ok ? n / N : 0;
  bool bad = false;
  for (uint64_t pk = tid; pk < npacks; pk += stride) {
    uint32_t ix[N];
#pragma unroll
    for (int q = 0; q < N / 4; q++) {
      const u32x4 t = swz_ld(reinterpret_cast<const u32x4*>(idx + pk * N + q * 4), NT);
      ix[q * 4 + 0] = t.x; ix[q * 4 + 1] = t.y; ix[q * 4 + 2] = t.z; ix[q * 4 + 3] = t.w;
    }
    OutPack<E, N> r;
#pragma unroll
    for (int k = 0; k < N; k++) {
      const bool ok = ix[k] < n_values;
      bad |= !ok;
      r.v[k] = values[ok ? ix[k] : 0];
      if (!ok) r.v[k] = E(0);
    }
    if (NT) __builtin_nontemporal_store(__builtin_bit_cast(u32x4, r), reinterpret_cast<u32x4*>(out + pk * N));
    else *reinterpret_cast<u32x4*>(out + pk * N) = __builtin_bit_cast(u32x4, r);
  }
  for (uint64_t i = npacks * N + tid; i < n; i += stride) {
    const uint32_t ix = idx[i];
    const bool ok = ix < n_values;
    bad |= !ok;
    out[i] = ok ? values[ix] : E(0);
  }
  if (bad) *reinterpret_cast<volatile uint32_t*>(flags) = AGPU_FLAG_INDEX_RANGE;
}

// out bit i = bits[idx[i]]: lane handles one index per round, ballot = 64 output bits
__global__ __launch_bounds__(AGPU_BLOCK) void take_bits_kernel(const uint32_t* bits, uint64_t n_bits, const uint32_t* idx,
                                                              uint64_t* out, uint64_t n, uint32_t* flags) {
  const uint32_t lane = threadIdx.x & (AGPU_WAVE - 1);
  const uint64_t wave_id = ((uint64_t)blockIdx.x * AGPU_BLOCK + threadIdx.x) / AGPU_WAVE;
  const uint64_t n_waves = (uint64_t)gridDim.x * (AGPU_BLOCK / AGPU_WAVE);
  const uint64_t nwords = (n + 63) / 64;
  bool bad = false;
  for (uint64_t w = wave_id; w < nwords; w += n_waves) {
    const uint64_t i = w * 64 + lane;
    bool bit = false;
    if (i < n) {
      const uint32_t ix = __builtin_nontemporal_load(idx + i);
      if (ix < n_bits) bit = (bits[ix >> 5] >> (ix & 31)) & 1u;
      else bad = true;
    }
    const uint64_t m = __ballot(bit);
    if (lane == 0) out[w] = m;
  }
  if (bad) *reinterpret_cast<volatile uint32_t*>(flags) = AGPU_FLAG_INDEX_RANGE;
}

// ---------------------------------------------------------------- put (in place on dst)
// n_src / n_dst = UINT64_MAX for the unchecked entry points (agpu_put / agpu_put_bits: lengths unknown to the ABI call)
template <int W, bool NT>
__global__ __launch_bounds__(AGPU_BLOCK) void put_kernel(const typename ElemOf<W>::type* src, uint64_t n_src,
                                                        const uint32_t* src_idx, typename ElemOf<W>::type* dst,
                                                        uint64_t n_dst, const uint32_t* dst_idx, uint64_t n, int vec_ok,
                                                        uint32_t* flags) {
  const uint64_t tid = (uint64_t)blockIdx.x * AGPU_BLOCK + threadIdx.x;
  const uint64_t stride = (uint64_t)gridDim.x * AGPU_BLOCK;
  const uint64_t npacks = vec_ok ? n / 4 : 0;
  bool bad = false;
  auto move = [&](uint32_t s, uint32_t d) {
    if (s < n_src && d < n_dst) dst[d] = src[s];
    else bad = true;
  };
  for (uint64_t pk = tid; pk < npacks; pk += stride) {
    const u32x4 si = swz_ld(reinterpret_cast<const u32x4*>(src_idx + pk * 4), NT);
    const u32x4 di = swz_ld(reinterpret_cast<const u32x4*>(dst_idx + pk * 4), NT);
    move(si.x, di.x); move(si.y, di.y); move(si.z, di.z); move(si.w, di.w);
  }
  for (uint64_t i = npacks * 4 + tid; i < n; i += stride) move(src_idx[i], dst_idx[i]);
  if (bad) *reinterpret_cast<volatile uint32_t*>(flags) = AGPU_FLAG_INDEX_RANGE;
}

// bit scatter: clear then set, word atomics like the reference (bool/put.wgsl:17-34)
__global__ __launch_bounds__(AGPU_BLOCK) void put_bits_kernel(const uint32_t* src, uint64_t n_src, const uint32_t* src_idx,
                                                             uint32_t* dst, uint64_t n_dst, const uint32_t* dst_idx,
                                                             uint64_t n, uint32_t* flags) {
  const uint64_t tid = (uint64_t)blockIdx.x * AGPU_BLOCK + threadIdx.x;
  const uint64_t stride = (uint64_t)gridDim.x * AGPU_BLOCK;
  bool bad = false;
  for (uint64_t i = tid; i < n; i += stride) {
    const uint32_t s = src_idx[i], d = dst_idx[i];
    if (!(s < n_src && d < n_dst)) {
      bad = true;
      continue;
    }
    const uint32_t bit = (src[s >> 5] >> (s & 31)) & 1u;
    if (bit) atomicOr(&dst[d >> 5], 1u << (d & 31));
    else atomicAnd(&dst[d >> 5], ~(1u << (d & 31)));
  }
  if (bad) *reinterpret_cast<volatile uint32_t*>(flags) = AGPU_FLAG_INDEX_RANGE;
}

// ---------------------------------------------------------------- merge: out[i] = mask bit i ? a[i] : b[i]
// Shape of the element-wise stream (elementwise.hip): one-wave blocks, one 16-byte pack per lane, nontemporal loads and
// stores; the wave's 8 mask words are one 32-byte row read through 8-lane broadcasts.
#define AGPU_MERGE_BLOCK 64
template <int W>
__global__ __launch_bounds__(AGPU_MERGE_BLOCK) void merge_kernel(const typename ElemOf<W>::type* a,
                                                          const typename ElemOf<W>::type* b, const uint32_t* mask,
                                                          typename ElemOf<W>::type* out, uint64_t n, int vec_ok) {
  typedef typename ElemOf<W>::type E;
  constexpr int N = 16 / W;  // 4, 8, 16 rows per lane; N divides 32 so a pack's mask bits sit in one word
  const uint64_t tid = (uint64_t)blockIdx.x * AGPU_MERGE_BLOCK + threadIdx.x;
  const uint64_t stride = (uint64_t)gridDim.x * AGPU_MERGE_BLOCK;
  const uint64_t npacks = vec_ok ? n / N : 0;
  for (uint64_t pk = tid; pk < npacks; pk += stride) {
    const OutPack<E, N> x = __builtin_bit_cast(OutPack<E, N>, __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(a + pk * N)));
    const OutPack<E, N> y = __builtin_bit_cast(OutPack<E, N>, __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(b + pk * N)));
    const uint64_t row = pk * N;
    const uint32_t m = mask[row >> 5] >> (row & 31);
    OutPack<E, N> r;
#pragma unroll
    for (int k = 0; k < N; k++) r.v[k] = ((m >> k) & 1u) ? x.v[k] : y.v[k];
    __builtin_nontemporal_store(__builtin_bit_cast(u32x4, r), reinterpret_cast<u32x4*>(out + pk * N));
  }
  for (uint64_t i = npacks * N + tid; i < n; i += stride)
    out[i] = ((mask[i >> 5] >> (i & 31)) & 1u) ? a[i] : b[i];
}

__global__ __launch_bounds__(AGPU_BLOCK) void index_max_kernel(const uint32_t* idx, uint64_t n, uint32_t* out) {
  const uint64_t tid = (uint64_t)blockIdx.x * AGPU_BLOCK + threadIdx.x;
  const uint64_t stride = (uint64_t)gridDim.x * AGPU_BLOCK;
  uint32_t m = 0;
  const uint64_t npacks = ((reinterpret_cast<uintptr_t>(idx) & 15u) == 0) ? n / 4 : 0;  // 16-byte loads when aligned
  for (uint64_t pk = tid; pk < npacks; pk += stride) {
    const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(idx) + pk);
    const uint32_t a = v.x > v.y ? v.x : v.y, b = v.z > v.w ? v.z : v.w;
    const uint32_t c = a > b ? a : b;
    m = c > m ? c : m;
  }
  for (uint64_t i = npacks * 4 + tid; i < n; i += stride) m = idx[i] > m ? idx[i] : m;
#pragma unroll
  for (int off = AGPU_WAVE / 2; off > 0; off >>= 1) {
    const uint32_t o = (uint32_t)__shfl_down((int)m, off);
    m = o > m ? o : m;
  }
  if ((threadIdx.x & (AGPU_WAVE - 1)) == 0 && m) atomicMax(out, m);
}

// nontemporal index / output streams: neutral for HBM-resident sources (A/B on one box: 638 vs 637 GB/s), +10 % when the
// source fits in L2 (they stop evicting it)
static constexpr bool swz_nt() { return true; }
static int gs_grid(const agpu_pipeline* p, uint64_t items) {
  return stream_grid_for(p, (items + AGPU_BLOCK - 1) / AGPU_BLOCK);
}

extern "C" {

agpu_status agpu_take(agpu_pipeline* p, int32_t width, const void* values, uint64_t n_values, const uint32_t* idx,
                      void* out, uint64_t n_idx) {
  AGPU_BIND(p);
  if (n_idx == 0) return AGPU_OK;
  AGPU_REQUIRE(values && idx && out, AGPU_ERR_ARG, "null pointer");
  AGPU_REQUIRE(n_values > 0, AGPU_ERR_SHAPE, "take from an empty array");
  const int vec_ok = aligned16(idx) && aligned16(out);
  const int grid = gs_grid(p, n_idx / (16 / (width > 0 ? width : 1)) + 1);
  switch (width) {
    case 4:
      hipLaunchKernelGGL((swz_nt() ? take_kernel<4, true> : take_kernel<4, false>), dim3(grid), dim3(AGPU_BLOCK), 0, p->stream, static_cast<const uint32_t*>(values),
                         n_values, idx, static_cast<uint32_t*>(out), n_idx, vec_ok, p->flags);
      break;
    case 2:
      hipLaunchKernelGGL((swz_nt() ? take_kernel<2, true> : take_kernel<2, false>), dim3(grid), dim3(AGPU_BLOCK), 0, p->stream, static_cast<const uint16_t*>(values),
                         n_values, idx, static_cast<uint16_t*>(out), n_idx, vec_ok, p->flags);
      break;
    case 1:
      hipLaunchKernelGGL((swz_nt() ? take_kernel<1, true> : take_kernel<1, false>), dim3(grid), dim3(AGPU_BLOCK), 0, p->stream, static_cast<const uint8_t*>(values),
                         n_values, idx, static_cast<uint8_t*>(out), n_idx, vec_ok, p->flags);
      break;
    default:
      agpu_set_error("take: width %d not supported (1, 2, 4)", width);
      return AGPU_ERR_UNSUPPORTED;
  }
  AGPU_LAUNCH_CHECK();
  return AGPU_OK;
}

agpu_status agpu_take_bits(agpu_pipeline* p, const void* bits, uint64_t n_bits, const uint32_t* idx, void* out_bits,
                           uint64_t n_idx) {
  AGPU_BIND(p);
  if (n_idx == 0) return AGPU_OK;
  AGPU_REQUIRE(bits && idx && out_bits, AGPU_ERR_ARG, "null pointer");
  AGPU_REQUIRE(n_bits > 0, AGPU_ERR_SHAPE, "take from an empty bitmap");
  AGPU_REQUIRE(aligned_to(bits, 4) && aligned_to(out_bits, 8), AGPU_ERR_SHAPE, "bitmap alignment");
  const uint64_t nwords = (n_idx + 63) / 64;
  const int grid = stream_grid_for(p, (nwords + 3) / 4);
  hipLaunchKernelGGL(take_bits_kernel, dim3(grid), dim3(AGPU_BLOCK), 0, p->stream, static_cast<const uint32_t*>(bits),
                     n_bits, idx, static_cast<uint64_t*>(out_bits), n_idx, p->flags);
  AGPU_LAUNCH_CHECK();
  return AGPU_OK;
}

agpu_status agpu_put_bounded(agpu_pipeline* p, int32_t width, const void* src, uint64_t n_src, const uint32_t* src_idx,
                             void* dst, uint64_t n_dst, const uint32_t* dst_idx, uint64_t n) {
  AGPU_BIND(p);
  if (n == 0) return AGPU_OK;
  AGPU_REQUIRE(src && src_idx && dst && dst_idx, AGPU_ERR_ARG, "null pointer");
  const int vec_ok = aligned16(src_idx) && aligned16(dst_idx);
  const int grid = gs_grid(p, n / 4 + 1);
#define AGPU_PUT_CASE(W, E)                                                                                              \
  case W:                                                                                                                \
    hipLaunchKernelGGL((swz_nt() ? put_kernel<W, true> : put_kernel<W, false>), dim3(grid), dim3(AGPU_BLOCK), 0, p->stream, \
                       static_cast<const E*>(src), n_src, src_idx, static_cast<E*>(dst), n_dst, dst_idx, n, vec_ok,     \
                       p->flags);                                                                                        \
    break;
  switch (width) {
    AGPU_PUT_CASE(4, uint32_t)
    AGPU_PUT_CASE(2, uint16_t)
    AGPU_PUT_CASE(1, uint8_t)
    default:
      agpu_set_error("put: width %d not supported (1, 2, 4)", width);
      return AGPU_ERR_UNSUPPORTED;
  }
#undef AGPU_PUT_CASE
  AGPU_LAUNCH_CHECK();
  return AGPU_OK;
}

agpu_status agpu_put(agpu_pipeline* p, int32_t width, const void* src, const uint32_t* src_idx, void* dst,
                     const uint32_t* dst_idx, uint64_t n) {
  return agpu_put_bounded(p, width, src, UINT64_MAX, src_idx, dst, UINT64_MAX, dst_idx, n);
}

agpu_status agpu_put_bits_bounded(agpu_pipeline* p, const void* src_bits, uint64_t n_src_bits, const uint32_t* src_idx,
                                  void* dst_bits, uint64_t n_dst_bits, const uint32_t* dst_idx, uint64_t n) {
  AGPU_BIND(p);
  if (n == 0) return AGPU_OK;
  AGPU_REQUIRE(src_bits && src_idx && dst_bits && dst_idx, AGPU_ERR_ARG, "null pointer");
  AGPU_REQUIRE(aligned_to(src_bits, 4) && aligned_to(dst_bits, 4), AGPU_ERR_SHAPE, "bitmap alignment");
  const int grid = gs_grid(p, n);
  hipLaunchKernelGGL(put_bits_kernel, dim3(grid), dim3(AGPU_BLOCK), 0, p->stream, static_cast<const uint32_t*>(src_bits),
                     n_src_bits, src_idx, static_cast<uint32_t*>(dst_bits), n_dst_bits, dst_idx, n, p->flags);
  AGPU_LAUNCH_CHECK();
  return AGPU_OK;
}

agpu_status agpu_put_bits(agpu_pipeline* p, const void* src_bits, const uint32_t* src_idx, void* dst_bits,
                          const uint32_t* dst_idx, uint64_t n) {
  return agpu_put_bits_bounded(p, src_bits, UINT64_MAX, src_idx, dst_bits, UINT64_MAX, dst_idx, n);
}

agpu_status agpu_merge(agpu_pipeline* p, int32_t width, const void* a, const void* b, const void* mask_bits, void* out,
                       uint64_t n) {
  AGPU_BIND(p);
  if (n == 0) return AGPU_OK;
  AGPU_REQUIRE(a && b && mask_bits && out, AGPU_ERR_ARG, "null pointer");
  AGPU_REQUIRE(aligned_to(mask_bits, 4), AGPU_ERR_SHAPE, "mask bitmap must be 4-byte aligned");
  const int vec_ok = aligned16(a) && aligned16(b) && aligned16(out);
  const int grid = stream_grid_for(p, (n / (16 / (width > 0 ? width : 1)) + AGPU_MERGE_BLOCK) / AGPU_MERGE_BLOCK);
  switch (width) {
    case 4:
      hipLaunchKernelGGL((merge_kernel<4>), dim3(grid), dim3(AGPU_MERGE_BLOCK), 0, p->stream, static_cast<const uint32_t*>(a),
                         static_cast<const uint32_t*>(b), static_cast<const uint32_t*>(mask_bits),
                         static_cast<uint32_t*>(out), n, vec_ok);
      break;
    case 2:
      hipLaunchKernelGGL((merge_kernel<2>), dim3(grid), dim3(AGPU_MERGE_BLOCK), 0, p->stream, static_cast<const uint16_t*>(a),
                         static_cast<const uint16_t*>(b), static_cast<const uint32_t*>(mask_bits),
                         static_cast<uint16_t*>(out), n, vec_ok);
      break;
    case 1:
      hipLaunchKernelGGL((merge_kernel<1>), dim3(grid), dim3(AGPU_MERGE_BLOCK), 0, p->stream, static_cast<const uint8_t*>(a),
                         static_cast<const uint8_t*>(b), static_cast<const uint32_t*>(mask_bits),
                         static_cast<uint8_t*>(out), n, vec_ok);
      break;
    default:
      agpu_set_error("merge: width %d not supported (1, 2, 4)", width);
      return AGPU_ERR_UNSUPPORTED;
  }
  AGPU_LAUNCH_CHECK();
  return AGPU_OK;
}

agpu_status agpu_index_max(agpu_pipeline* p, const uint32_t* idx, uint64_t n, uint32_t* out_max_dev) {
  AGPU_BIND(p);
  AGPU_REQUIRE(out_max_dev, AGPU_ERR_ARG, "null output");
  AGPU_HIP(hipMemsetAsync(out_max_dev, 0, sizeof(uint32_t), p->stream));
  if (n == 0) return AGPU_OK;
  AGPU_REQUIRE(idx, AGPU_ERR_ARG, "null pointer");
  const int grid = atomic_grid_for(p, (n + AGPU_BLOCK * 8 - 1) / (AGPU_BLOCK * 8));  // few blocks: same-address atomics
  hipLaunchKernelGGL(index_max_kernel, dim3(grid), dim3(AGPU_BLOCK), 0, p->stream, idx, n, out_max_dev);
  AGPU_LAUNCH_CHECK();
  return AGPU_OK;
}

}  // extern "C"
