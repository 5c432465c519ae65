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
#include <atomic>
#include <chrono>

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
                                                         int vec_ok, uint32_t* flags, const uint32_t* only_if = nullptr) {
  typedef typename ElemOf<W>::type E;
  if (only_if && !*only_if) return;  // launched behind a pipeline whose locality probe decides which of the two does the work
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
                                                              uint64_t* out, uint64_t n, uint32_t* flags, const uint32_t* only_if = nullptr) {
  if (only_if && !*only_if) return;
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
                                                        uint32_t* flags, const uint32_t* only_if) {
  if (only_if && !*only_if) return;  // behind the bucketed pipeline: see take_kernel
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

// ---------------------------------------------------------------- the bucketed pipelines (the locality lever): two include files of this
// translation unit — swizzle_put.inc (pair pipeline of the put + everything both share), swizzle_take.inc (merge-back pipeline of the take)
#include "swizzle_put.inc"
#include "swizzle_take.inc"

// tuning "gather_bucket": 0 = auto, 1 = always direct, 2 = bucketed whenever the shape qualifies (take: the merge-back
// pipeline), 4 = like 2 plus the device-side locality probe (tests).
// Auto is two decisions: THIS one, on the host, by size and sparsity — is a pipeline worth enqueuing at all? — and the probe's,
// on the device, by what the index columns look like (idx_locality_kernel) — which of the enqueued forms does the work.
// The size thresholds, from one-process A/B sweeps on MI355X (tools/probe/bucket_sweep.py --crossover3 → profiles/r03_gather_crossover.json,
// r03_gather_sweep.json; uniformly random 4-byte rows): TAKE goes bucketed from 2^25 rows when the source is at least
// 16 MiB and not sparser than 1 row in 8 elements — merge-back 1.2–1.5× at 2^25 rows, 1.3–1.7× at 2^26, 1.8–2.1× at 2^28
// (48 → 90–113 G rows/s); at 2^24 rows it wins only 1.1–1.2× and loses against a sparse source, below that the six launches
// cost more than the random transactions they save.  PUT goes bucketed from 2^24 rows when neither side is sparser than
// 1 row in 8 elements: 1.25–1.35× at 2^24, 1.5–1.75× at 2^25, 1.6–2.4× at 2^26, 2.4–3.5× at 2^28 (15 → 53 G rows/s with both
// sides random over 1 GiB).
static bool want_bucketed(const agpu_pipeline* p, int width, uint64_t n, uint64_t n_src, uint64_t n_dst, bool is_put) {
  const int64_t mode = p->tune.gather_bucket;
  if (mode == 1) return false;
  if (mode == 2 || mode == 4) return n >= BKT_TILE;  // 4: like 2, but with the device-side probe (tests: both outcomes at small sizes)
  if (n_src / 8 > n || n_dst / 8 > n) return false;
  if (is_put) return n >= ((uint64_t)1 << 24);
  return n >= ((uint64_t)1 << 25) && n_src * (uint64_t)width >= ((uint64_t)16 << 20);
}

// nontemporal index / output streams: neutral for HBM-resident sources (A/B on one box: 638 vs 637 GB/s), +10 % when the
// source fits in L2 (they stop evicting it)
static constexpr bool swz_nt() { return true; }
static int gs_grid(const agpu_pipeline* p, uint64_t items) {
  return stream_grid_for(p, (items + AGPU_BLOCK - 1) / AGPU_BLOCK);
}

// the direct gather; only_if != nullptr: launched behind a pipeline, does the work only when the locality probe said so
static agpu_status launch_take_direct(agpu_pipeline* p, int width, const void* values, uint64_t n_values, const uint32_t* idx, void* out,
                                      uint64_t n_idx, const uint32_t* only_if) {
  const int vec_ok = aligned16(idx) && aligned16(out);
  int grid = gs_grid(p, n_idx / (16 / (width > 0 ? width : 1)) + 1);
  // behind a pipeline: a grid-stride launch of 128 blocks per CU — when the probe chose the pipeline, the blocks that return at once
  // are few (131 072 empty blocks cost 50 µs at 2^27 rows), when it chose this kernel the grid still covers the chip many times
  // (32 per CU: 0.36 → 0.50 ms on sorted indices)
  if (only_if && grid > p->dev->num_cus * 128) grid = p->dev->num_cus * 128;
  switch (width) {
    case 4:
      hipLaunchKernelGGL((swz_nt() ? take_kernel<4, true> : take_kernel<4, false>), dim3(grid), dim3(AGPU_BLOCK), 0, p->stream, static_cast<const uint32_t*>(values),
                         n_values, idx, static_cast<uint32_t*>(out), n_idx, vec_ok, p->flags, only_if);
      break;
    case 2:
      hipLaunchKernelGGL((swz_nt() ? take_kernel<2, true> : take_kernel<2, false>), dim3(grid), dim3(AGPU_BLOCK), 0, p->stream, static_cast<const uint16_t*>(values),
                         n_values, idx, static_cast<uint16_t*>(out), n_idx, vec_ok, p->flags, only_if);
      break;
    case 1:
      hipLaunchKernelGGL((swz_nt() ? take_kernel<1, true> : take_kernel<1, false>), dim3(grid), dim3(AGPU_BLOCK), 0, p->stream, static_cast<const uint8_t*>(values),
                         n_values, idx, static_cast<uint8_t*>(out), n_idx, vec_ok, p->flags, only_if);
      break;
    default:
      agpu_set_error("take: width %d not supported (1, 2, 4)", width);
      return AGPU_ERR_UNSUPPORTED;
  }
  AGPU_LAUNCH_CHECK();
  return AGPU_OK;
}
static agpu_status launch_put_direct(agpu_pipeline* p, int width, const void* src, uint64_t n_src, const uint32_t* src_idx, void* dst,
                                     uint64_t n_dst, const uint32_t* dst_idx, uint64_t n, const uint32_t* only_if) {
  const int vec_ok = aligned16(src_idx) && aligned16(dst_idx);
  int grid = gs_grid(p, n / 4 + 1);
  if (only_if && grid > p->dev->num_cus * 128) grid = p->dev->num_cus * 128;  // see launch_take_direct
#define AGPU_PUT_CASE(W, E)                                                                                              \
  case W:                                                                                                                \
    hipLaunchKernelGGL((swz_nt() ? put_kernel<W, true> : put_kernel<W, false>), dim3(grid), dim3(AGPU_BLOCK), 0, p->stream, \
                       static_cast<const E*>(src), n_src, src_idx, static_cast<E*>(dst), n_dst, dst_idx, n, vec_ok,     \
                       p->flags, only_if);                                                                               \
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

static agpu_status launch_take_bits_direct(agpu_pipeline* p, const void* bits, uint64_t n_bits, const uint32_t* idx, void* out_bits, uint64_t n_idx,
                                           const uint32_t* only_if) {
  const uint64_t nwords = (n_idx + 63) / 64;
  int grid = stream_grid_for(p, (nwords + 3) / 4);
  if (only_if && grid > p->dev->num_cus * 32) grid = p->dev->num_cus * 32;  // see launch_take_direct
  hipLaunchKernelGGL(take_bits_kernel, dim3(grid), dim3(AGPU_BLOCK), 0, p->stream, static_cast<const uint32_t*>(bits), n_bits, idx,
                     static_cast<uint64_t*>(out_bits), n_idx, p->flags, only_if);
  AGPU_LAUNCH_CHECK();
  return AGPU_OK;
}



extern "C" {

agpu_status agpu_take(agpu_pipeline* p, int32_t width, const void* values, uint64_t n_values, const uint32_t* idx,
                      void* out, uint64_t n_idx) {
  AGPU_BIND(p);
  if (n_idx == 0) return AGPU_OK;
  AGPU_REQUIRE(values && idx && out, AGPU_ERR_ARG, "null pointer");
  AGPU_REQUIRE(n_values > 0, AGPU_ERR_SHAPE, "take from an empty array");
  if ((width == 1 || width == 2 || width == 4) && n_values != UINT64_MAX && want_bucketed(p, width, n_idx, n_values, n_idx, false)) {
    if (n_idx >= TK2_TILE) {
      bool adaptive = p->tune.gather_bucket == 0 || p->tune.gather_bucket == 4;
      if (p->tune.gather_bucket == 0) {  // the probe's answer on the host, if it comes in time: only the chosen form is enqueued
        const int d = probe_decide(p, idx, nullptr, n_idx, width == 4 ? 5 : width == 2 ? 6 : 7, 0);
        if (d >= 0 && (d & 1)) return launch_take_direct(p, width, values, n_values, idx, out, n_idx, nullptr);
        if (d >= 0) adaptive = false;
      }
      const agpu_status ms = launch_take_mergeback(p, width, values, n_values, idx, out, n_idx, nullptr, nullptr, adaptive);
      if (ms != AGPU_ERR_UNSUPPORTED) return ms;
    }
  }
  return launch_take_direct(p, width, values, n_values, idx, out, n_idx, nullptr);
}

// The columns of one table by ONE index column.  At pipeline sizes the index work of the merge-back take (a third of its time) is done once
// for all of them (launch_take_mergeback's `more`); otherwise — small takes, sorted / local indices (the probe), unsupported widths — column
// by column through agpu_take / agpu_take_validity.  The first column decides (width of the probe's threshold, want_bucketed): the columns
// share n_values.  validities / out_validities may be NULL (no column has nulls); entry c NULL: column c has none.
static agpu_status take_columns_impl(agpu_pipeline* p, int32_t n_cols, const int32_t* widths, const void* const* values, const void* const* validities,
                                     uint64_t n_values, const uint32_t* idx, void* const* outs, void* const* out_validities, uint64_t n_idx) {
  auto vb = [&](int32_t c) -> const void* { return validities ? validities[c] : nullptr; };
  {
    AGPU_BIND(p);
    AGPU_REQUIRE(n_cols >= 0 && n_cols <= 64, AGPU_ERR_ARG, "0..64 columns");
    if (n_cols == 0 || n_idx == 0) return AGPU_OK;
    AGPU_REQUIRE(widths && values && outs && idx, AGPU_ERR_ARG, "null pointer");
    for (int32_t c = 0; c < n_cols; c++) {
      AGPU_REQUIRE(values[c] && outs[c], AGPU_ERR_ARG, "null column pointer");
      AGPU_REQUIRE(!vb(c) || (out_validities && out_validities[c]), AGPU_ERR_ARG, "out_validity required for a column with a validity bitmap");
      AGPU_REQUIRE(!vb(c) || (aligned_to(vb(c), 4) && aligned_to(out_validities[c], 8)), AGPU_ERR_SHAPE, "bitmap alignment");
    }
    AGPU_REQUIRE(n_values > 0, AGPU_ERR_SHAPE, "take from an empty array");
    const int w0 = widths[0];
    if (n_cols > 1 && (w0 == 1 || w0 == 2 || w0 == 4) && n_values != UINT64_MAX && n_idx >= TK2_TILE &&
        want_bucketed(p, w0, n_idx, n_values, n_idx, false)) {
      bool adaptive = p->tune.gather_bucket == 0 || p->tune.gather_bucket == 4;
      bool go_direct = false;
      if (p->tune.gather_bucket == 0) {
        const int d = probe_decide(p, idx, nullptr, n_idx, w0 == 4 ? 5 : w0 == 2 ? 6 : 7, 0);
        if (d >= 0 && (d & 1)) go_direct = true;
        else if (d >= 0) adaptive = false;
      }
      if (!go_direct) {
        TakeCol more[64];
        for (int32_t c = 1; c < n_cols; c++)
          more[c - 1] = TakeCol{widths[c], values[c], outs[c], static_cast<const uint32_t*>(vb(c)), vb(c) ? static_cast<uint64_t*>(out_validities[c]) : nullptr};
        const agpu_status ms = launch_take_mergeback(p, w0, values[0], n_values, idx, outs[0], n_idx, static_cast<const uint32_t*>(vb(0)),
                                                     vb(0) ? static_cast<uint64_t*>(out_validities[0]) : nullptr, adaptive, nullptr, 0, nullptr, more,
                                                     n_cols - 1);
        if (ms != AGPU_ERR_UNSUPPORTED) return ms;
      } else {
        for (int32_t c = 0; c < n_cols; c++) {
          agpu_status st = launch_take_direct(p, widths[c], values[c], n_values, idx, outs[c], n_idx, nullptr);
          if (st == AGPU_OK && vb(c)) st = launch_take_bits_direct(p, vb(c), n_values, idx, out_validities[c], n_idx, nullptr);
          if (st != AGPU_OK) return st;
        }
        return AGPU_OK;
      }
    }
  }
  for (int32_t c = 0; c < n_cols; c++) {
    const agpu_status st = vb(c) ? agpu_take_validity(p, widths[c], values[c], n_values, vb(c), idx, outs[c], out_validities[c], n_idx)
                                 : agpu_take(p, widths[c], values[c], n_values, idx, outs[c], n_idx);
    if (st != AGPU_OK) return st;
  }
  return AGPU_OK;
}
agpu_status agpu_take_columns(agpu_pipeline* p, int32_t n_cols, const int32_t* widths, const void* const* values, uint64_t n_values,
                              const uint32_t* idx, void* const* outs, uint64_t n_idx) {
  return take_columns_impl(p, n_cols, widths, values, nullptr, n_values, idx, outs, nullptr, n_idx);
}
agpu_status agpu_take_columns_validity(agpu_pipeline* p, int32_t n_cols, const int32_t* widths, const void* const* values,
                                       const void* const* validities, uint64_t n_values, const uint32_t* idx, void* const* outs,
                                       void* const* out_validities, uint64_t n_idx) {
  return take_columns_impl(p, n_cols, widths, values, validities, n_values, idx, outs, out_validities, n_idx);
}

agpu_status agpu_take_validity(agpu_pipeline* p, int32_t width, const void* values, uint64_t n_values, const void* validity,
                               const uint32_t* idx, void* out, void* out_validity, uint64_t n_idx) {
  if (!validity) return agpu_take(p, width, values, n_values, idx, out, n_idx);
  AGPU_REQUIRE(out_validity, AGPU_ERR_ARG, "out_validity required when the source has a validity bitmap");
  {
    AGPU_BIND(p);
    if (n_idx == 0) return AGPU_OK;
    AGPU_REQUIRE(values && idx && out, AGPU_ERR_ARG, "null pointer");
    AGPU_REQUIRE(n_values > 0, AGPU_ERR_SHAPE, "take from an empty array");
    AGPU_REQUIRE(aligned_to(validity, 4) && aligned_to(out_validity, 8), AGPU_ERR_SHAPE, "bitmap alignment");
    if ((width == 4 || width == 2 || width == 1) && n_values != UINT64_MAX && n_idx >= TK2_TILE &&
        want_bucketed(p, width, n_idx, n_values, n_idx, false)) {
      bool adaptive = p->tune.gather_bucket == 0 || p->tune.gather_bucket == 4;
      bool go_direct = false;
      if (p->tune.gather_bucket == 0) {
        const int d = probe_decide(p, idx, nullptr, n_idx, width == 4 ? 5 : width == 2 ? 6 : 7, 0);
        if (d >= 0 && (d & 1)) go_direct = true;
        else if (d >= 0) adaptive = false;
      }
      if (go_direct) {
        const agpu_status st1 = launch_take_direct(p, width, values, n_values, idx, out, n_idx, nullptr);
        if (st1 != AGPU_OK) return st1;
        return launch_take_bits_direct(p, validity, n_values, idx, out_validity, n_idx, nullptr);
      }
      const agpu_status ms = launch_take_mergeback(p, width, values, n_values, idx, out, n_idx,
                                                   static_cast<const uint32_t*>(validity), static_cast<uint64_t*>(out_validity), adaptive);
      if (ms != AGPU_ERR_UNSUPPORTED) return ms;
    }
  }
  const agpu_status st = agpu_take(p, width, values, n_values, idx, out, n_idx);
  if (st != AGPU_OK) return st;
  return agpu_take_bits(p, validity, n_values, idx, out_validity, n_idx);
}

static agpu_status take_bits_impl(agpu_pipeline* p, const void* bits, uint64_t n_bits, const uint32_t* idx, void* out_bits, uint64_t n_idx) {
  if (n_idx == 0) return AGPU_OK;
  AGPU_REQUIRE(bits && idx && out_bits, AGPU_ERR_ARG, "null pointer");
  AGPU_REQUIRE(n_bits > 0, AGPU_ERR_SHAPE, "take from an empty bitmap");
  AGPU_REQUIRE(aligned_to(bits, 4) && aligned_to(out_bits, 8), AGPU_ERR_SHAPE, "bitmap alignment");
  if (n_bits != UINT64_MAX && n_idx >= TK2_TILE && p->tune.gather_bucket != 1 &&
      (p->tune.gather_bucket == 2 || p->tune.gather_bucket == 4 ||
       (n_idx >= ((uint64_t)1 << 25) && n_bits >= ((uint64_t)1 << 27) && n_bits / 8 <= n_idx))) {
    // round 3: the merge-back pipeline with the bitmap's words as the elements (auto: ≥ 2^25 rows from a bitmap of ≥ 16 MiB)
    bool adaptive = p->tune.gather_bucket == 0 || p->tune.gather_bucket == 4;
    if (p->tune.gather_bucket == 0) {
      const int d = probe_decide(p, idx, nullptr, n_idx, 10, 0);
      if (d >= 0 && (d & 1)) return launch_take_bits_direct(p, bits, n_bits, idx, out_bits, n_idx, nullptr);
      if (d >= 0) adaptive = false;
    }
    const agpu_status ms = launch_take_bits_mergeback(p, static_cast<const uint32_t*>(bits), n_bits, idx, static_cast<uint64_t*>(out_bits), n_idx,
                                                      nullptr, 0, nullptr, adaptive);
    if (ms != AGPU_ERR_UNSUPPORTED) return ms;
  }
  return launch_take_bits_direct(p, bits, n_bits, idx, out_bits, n_idx, nullptr);
}

agpu_status agpu_take_bits(agpu_pipeline* p, const void* bits, uint64_t n_bits, const uint32_t* idx, void* out_bits,
                           uint64_t n_idx) {
  AGPU_BIND(p);
  return take_bits_impl(p, bits, n_bits, idx, out_bits, n_idx);
}

// ---------------------------------------------------------------- Boolean put, round 3: bucketed by destination region
// dst bit dst_idx[i] = src bit src_idx[i] [ref: crates/routines/src/bool.rs put_op + bool/put.wgsl; the validity of a null-aware
// put].  The direct kernel above is one device-scope atomic per row on a random bitmap word: the chip retires ≈ 26 G of them per
// second (10.4 ms at 2^28 rows).  Here:
//   T   tbits[i] = src bit src_idx[i]: a Boolean take in natural order (the merge-back pipeline above at these sizes)
//   E   ent[i] = dst_idx[i] * 2 + tbits[i], or 0xFFFFFFFF for a row with either index out of range (dropped + sticky flag)   12 B/row
//   H2 / scans / P2 of the take pipeline over `ent` with the key ent >> (r + 1): runs of entries by DESTINATION region (2^r bits)
//   S   one workgroup per destination region: its 2^r bits (32 KiB for r = 18) live in LDS, the region's entries are applied
//       with LDS atomics, the words go back — no global atomic anywhere.
// Duplicate destinations: unspecified winner, like the direct kernel.
__global__ __launch_bounds__(BKT_T) void pb_apply_kernel(const uint32_t* ents, const BktCtl* ctl, int r, uint32_t* dst, uint64_t n_dst) {
  extern __shared__ uint32_t pb_words[];
  const uint32_t b = blockIdx.x;
  const uint64_t nwords_all = (n_dst + 31) / 32, w0 = (uint64_t)b << (r - 5);
  const uint32_t nw = (uint32_t)((nwords_all - w0) < ((uint64_t)1 << (r - 5)) ? (nwords_all - w0) : ((uint64_t)1 << (r - 5)));
  const uint32_t beg = ctl->base_s[b], end = ctl->base_s[b + 1];
  if (beg == end) return;  // nothing lands in this region: its words stay as they are
  for (uint32_t k = threadIdx.x; k < nw; k += BKT_T) pb_words[k] = dst[w0 + k];
  __syncthreads();
  auto apply = [&](uint32_t e) {
    const uint32_t d = e >> 1, w = (d >> 5) - (uint32_t)w0;
    if (e & 1u) atomicOr(&pb_words[w], 1u << (d & 31));
    else atomicAnd(&pb_words[w], ~(1u << (d & 31)));
  };
  uint32_t j = beg + threadIdx.x;
  for (; j + 3 * BKT_T < end; j += 4 * BKT_T) {  // four loads in flight per lane
    const uint32_t e0 = ents[j], e1 = ents[j + BKT_T], e2 = ents[j + 2 * BKT_T], e3 = ents[j + 3 * BKT_T];
    apply(e0); apply(e1); apply(e2); apply(e3);
  }
  for (; j < end; j += BKT_T) apply(ents[j]);
  __syncthreads();
  for (uint32_t k = threadIdx.x; k < nw; k += BKT_T) dst[w0 + k] = pb_words[k];
}

static agpu_status launch_put_bits_bucketed(agpu_pipeline* p, const uint32_t* src_bits, uint64_t n_src, const uint32_t* si, uint32_t* dst_bits,
                                            uint64_t n_dst, const uint32_t* di, uint64_t n) {
  if (n >= 0xFFFF0000ull || n_dst >= 0x7FFFFFFFull || n_src > 0xFFFFFFFFull || !aligned16(si) || !aligned16(di) || p->capturing)
    return AGPU_ERR_UNSUPPORTED;
  int r = 18;  // 2^18 destination bits = 32 KiB of LDS per region; larger bitmaps: larger regions, up to 128 KiB
  while (((n_dst + ((uint64_t)1 << r) - 1) >> r) > BKT_MAX - 1) r++;
  if (r > 20) return AGPU_ERR_UNSUPPORTED;
  const int rs = r + 1;  // the partition key of an entry (destination * 2 + bit)
  const uint64_t n_ent = 2 * n_dst;
  const uint32_t bs = (uint32_t)((n_dst + ((uint64_t)1 << r) - 1) >> r);
  agpu_device* dev = p->dev;
  const uint32_t ntiles = (uint32_t)((n + TK2_TILE - 1) / TK2_TILE);
  const uint32_t nbp = (bs + 1 + 3) & ~3u;
  const uint32_t nchunks = (ntiles + BKT_CHUNK - 1) / BKT_CHUNK;
  void *ctl_v = nullptr, *tb_v = nullptr, *ent_v = nullptr, *srt_v = nullptr, *cnt_v = nullptr, *off_v = nullptr, *csum_v = nullptr;
  agpu_status st = agpu_malloc(dev, sizeof(BktCtl), 0, &ctl_v);
  if (st == AGPU_OK) st = agpu_malloc(dev, 4 * n + 16, 0, &ent_v);
  if (st == AGPU_OK) st = agpu_malloc(dev, 4 * n + 16, 0, &srt_v);
  if (st == AGPU_OK) st = agpu_malloc(dev, (size_t)ntiles * nbp * 2, 0, &cnt_v);
  if (st == AGPU_OK) st = agpu_malloc(dev, (size_t)ntiles * nbp * 4, 0, &off_v);
  if (st == AGPU_OK) st = agpu_malloc(dev, (size_t)nchunks * nbp * 4, 0, &csum_v);
  if (st != AGPU_OK) st = AGPU_ERR_UNSUPPORTED;
  bool have_entries = false, src_local_known = false;
  if (st == AGPU_OK && n >= TK2_TILE &&
      (p->tune.gather_bucket == 2 || p->tune.gather_bucket == 4 || (n >= ((uint64_t)1 << 25) && n_src >= ((uint64_t)1 << 27) && n_src / 8 <= n))) {
    // T + E in one: the Boolean take's merge pass emits the entries itself (no natural-order bitmap in between)
    bool adaptive = p->tune.gather_bucket == 0 || p->tune.gather_bucket == 4;  // local source indices: the merge-back kernels return, the direct gather + E run
    if (p->tune.gather_bucket == 0) {
      const int d = probe_decide(p, si, nullptr, n, 10, 0);
      if (d >= 0) {
        adaptive = false;
        src_local_known = (d & 1) != 0;
      }
    }
    if (adaptive && agpu_malloc(dev, (n + 63) / 64 * 8 + 16, 0, &tb_v) != AGPU_OK) tb_v = nullptr;
    if (!src_local_known) {  // (known local: straight to the direct bit gather + E below)
      const agpu_status ms = launch_take_bits_mergeback(p, src_bits, n_src, si, nullptr, n, di, n_dst, static_cast<uint32_t*>(ent_v), adaptive && tb_v, tb_v);
      if (ms == AGPU_OK) have_entries = true;
      else if (ms != AGPU_ERR_UNSUPPORTED) st = ms;
    }
  }
  if (st == AGPU_OK && !have_entries) {
    // T: out-of-range source indices read 0 here and raise the flag; E drops those rows
    if (!tb_v) st = agpu_malloc(dev, (n + 63) / 64 * 8 + 16, 0, &tb_v);
    if (st != AGPU_OK) st = AGPU_ERR_UNSUPPORTED;
    else st = src_local_known ? launch_take_bits_direct(p, src_bits, n_src, si, tb_v, n, nullptr) : take_bits_impl(p, src_bits, n_src, si, tb_v, n);
  }
  if (st == AGPU_OK) {
    BktCtl* ctl = static_cast<BktCtl*>(ctl_v);
    uint16_t* counts = static_cast<uint16_t*>(cnt_v);
    uint32_t* offsets = static_cast<uint32_t*>(off_v);
    uint32_t* csum = static_cast<uint32_t*>(csum_v);
    uint32_t* ent = static_cast<uint32_t*>(ent_v);
    hipError_t e = hipMemsetAsync(ctl, 0, sizeof(BktCtl), p->stream);
    if (e != hipSuccess) {
      agpu_set_error("hipMemsetAsync failed: %s", hipGetErrorString(e));
      st = AGPU_ERR_HIP;
    } else {
      const dim3 cgrid((nbp + 255) / 256, nchunks);
      uint64_t hg = (uint64_t)dev->num_cus * 2;
      if (hg > ntiles) hg = ntiles;
      if (!have_entries)
        hipLaunchKernelGGL(pb_entries_kernel, dim3((uint32_t)((n + 1023) / 1024)), dim3(256), 0, p->stream, si, di, static_cast<const uint32_t*>(tb_v), n,
                           n_src, n_dst, ent, static_cast<const uint32_t*>(nullptr));
      hipLaunchKernelGGL(tk2_hist_kernel, dim3((uint32_t)hg), dim3(BKT_T), 0, p->stream, ent, n, n_ent, rs, bs, p->flags, counts, nbp, ntiles);
      hipLaunchKernelGGL(bkt_colsum_kernel, cgrid, dim3(256), 0, p->stream, counts, nbp, ntiles, csum);
      hipLaunchKernelGGL(bkt_colscan_kernel, dim3((nbp + 255) / 256), dim3(256), 0, p->stream, csum, nbp, nchunks, ctl->hist_s);
      hipLaunchKernelGGL(bkt_scan_kernel, dim3(1), dim3(BKT_T), 0, p->stream, ctl, bs, 0u, 1u, 1u);
      hipLaunchKernelGGL(bkt_offsets_kernel, cgrid, dim3(256), 0, p->stream, counts, csum, nbp, ntiles, ctl->base_s, offsets);
      hipLaunchKernelGGL(tk2_partition_kernel, dim3((ntiles + 7) / 8 * 8), dim3(BKT_T), 0, p->stream, ent, n, n_ent, rs, bs, offsets, nbp, ntiles,
                         static_cast<uint32_t*>(srt_v), static_cast<uint16_t*>(nullptr));
      if (r > 18) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(pb_apply_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 1 << (r - 3));
      hipLaunchKernelGGL(pb_apply_kernel, dim3(bs), dim3(BKT_T), (size_t)1 << (r - 3), p->stream, static_cast<const uint32_t*>(srt_v), ctl, r, dst_bits,
                         n_dst);
      if (hipGetLastError() != hipSuccess) {
        agpu_set_error("bucketed put_bits launch failed");
        st = AGPU_ERR_HIP;
      }
    }
  }
  for (void* q : {csum_v, off_v, cnt_v, srt_v, ent_v, tb_v, ctl_v})
    if (q) (void)agpu_free(dev, q);
  return st;
}

agpu_status agpu_put_bounded(agpu_pipeline* p, int32_t width, const void* src, uint64_t n_src, const uint32_t* src_idx,
                             void* dst, uint64_t n_dst, const uint32_t* dst_idx, uint64_t n) {
  AGPU_BIND(p);
  if (n == 0) return AGPU_OK;
  AGPU_REQUIRE(src && src_idx && dst && dst_idx, AGPU_ERR_ARG, "null pointer");
  if ((width == 1 || width == 2 || width == 4) && n_src != UINT64_MAX && n_dst != UINT64_MAX &&
      want_bucketed(p, width, n, n_src, n_dst, true)) {
    bool adaptive = p->tune.gather_bucket == 0 || p->tune.gather_bucket == 4;
    if (p->tune.gather_bucket == 0) {  // the probe's answer on the host, if it comes in time: ONE form is enqueued, nothing gated
      const int sh = width == 4 ? 5 : width == 2 ? 6 : 7;
      const int d = probe_decide(p, src_idx, dst_idx, n, sh, sh);
      if (d == 3) return launch_put_direct(p, width, src, n_src, src_idx, dst, n_dst, dst_idx, n, nullptr);
      if (d == 1) {  // source local, destination random
        const agpu_status ls = launch_put_dst_only(p, width, src, n_src, src_idx, dst, n_dst, dst_idx, n);
        if (ls != AGPU_ERR_UNSUPPORTED) return ls;
      }
      if (d == 2 && n >= TK2_TILE) {  // source random, destination local
        const agpu_status ls = launch_put_through_take(p, width, src, n_src, src_idx, dst, n, dst_idx, n_dst, nullptr);
        if (ls != AGPU_ERR_UNSUPPORTED) return ls;
      }
      if (d >= 0) adaptive = false;  // both random (or a form that could not run): the full pipeline, nothing gated
    }
    const agpu_status bs = launch_bucketed(p, width, src, n_src, src_idx, dst, n_dst, dst_idx, n, adaptive);
    if (bs != AGPU_ERR_UNSUPPORTED) return bs;
  }
  return launch_put_direct(p, width, src, n_src, src_idx, dst, n_dst, dst_idx, n, nullptr);
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
  if (n_src_bits != UINT64_MAX && n_dst_bits != UINT64_MAX && n_src_bits > 0 && n_dst_bits > 0 && p->tune.gather_bucket != 1 &&
      (p->tune.gather_bucket == 2 || p->tune.gather_bucket == 4 ||
       (n >= ((uint64_t)1 << 24) && n_dst_bits >= ((uint64_t)1 << 22)))) {
    // (a destination of fewer than 16 regions leaves the apply pass with too few workgroups: the direct kernel keeps those)
    // round 3: bucketed by destination region, no global atomics (auto from 2^24 rows: 0.67 → 0.46 ms there, 10.4 → 3.6 at 2^28)
    const agpu_status bs = launch_put_bits_bucketed(p, static_cast<const uint32_t*>(src_bits), n_src_bits, src_idx, static_cast<uint32_t*>(dst_bits),
                                                    n_dst_bits, dst_idx, n);
    if (bs != AGPU_ERR_UNSUPPORTED) return bs;
  }
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
