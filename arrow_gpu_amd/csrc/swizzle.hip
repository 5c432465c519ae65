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

// ---------------------------------------------------------------- bucketed take / put (the locality lever)
// A uniformly random 4-byte gather pulls one 128-byte line per row (131.5 B/row measured, profiles/r01_gather_sweep.json)
// and a random 4-byte scatter one 32-byte partial-sector read-modify-write per row: both are bound by the NUMBER of
// random HBM transactions (52 G rows/s take, 15 G rows/s put at 2^28 rows), not by bytes.  The bucketed form turns every
// random access into an L2 hit by moving the indices to the data instead of the data to the indices:
//   H  histogram of source buckets (bucket = index >> R, a 2^R-element region of at most 1 MiB) and, for put, of
//      destination buckets — one streaming pass over the index column(s);
//   P  partition: pairs (src index, destination) written to their source bucket's range (LDS ranks inside a 16 Ki-row
//      tile, one global atomic per non-empty (tile, bucket));
//   G  gather + re-partition: blocks walk the pair list IN BUCKET ORDER, each XCD a contiguous eighth of it, so at any
//      moment an XCD's L2 serves one or two 1 MiB source regions; the fetched value is written with its destination to
//      the DESTINATION bucket's range;
//   F  final store: the same walk over destination buckets — every 4-byte store lands in a region the XCD's L2 is busy
//      assembling, and leaves for HBM as full lines.
// Streaming traffic ≈ 52 B/row instead of 132–160 B/row of random transactions.  Two 8 B/row temporaries come from
// the pool.  Out-of-range indices keep the robust-access outcome: take reads 0, put drops the row, the sticky flag is set.
#ifndef BKT_T
#define BKT_T 1024     // threads per block
#endif
#define BKT_MAX (4 * BKT_T)  // keys per tile sort = buckets per side (every thread owns 4 counters of the scan)
#ifndef BKT_E
#define BKT_E 16       // rows per thread → 16 Ki-row tiles (128 KiB of LDS per workgroup)
#endif
#ifndef BKT_RD_EXTRA
#define BKT_RD_EXTRA 0  // tools/probe/put_variants.sh: destination regions 2^BKT_RD_EXTRA times the source regions
#endif
#define BKT_TILE (BKT_T * BKT_E)
#define BKT_INVALID 0xFFFFFFFFu

// A region's range cursor takes one global atomic per (tile, region); with all cursors in one 16 KiB array every tile's
// 2048 atomics land in a handful of L2 channels and the reservation phase was 57 % of the partition kernel's tile time
// (tools/probe/bkt_phases.py: 35 600 of 62 500 cycles).  One cursor per 128-byte line spreads them over the channels.
#define BKT_CUR_STRIDE 32
struct BktCtl {  // device-side control block
  uint32_t hist_s[BKT_MAX + 1];  // +1: take's out-of-range rows (value 0 at the end)
  uint32_t hist_d[BKT_MAX + 1];
  uint32_t cur_s[(BKT_MAX + 1) * BKT_CUR_STRIDE];
  uint32_t cur_d[(BKT_MAX + 1) * BKT_CUR_STRIDE];
  uint32_t base_s[BKT_MAX + 1];  // first pair of every source region (exclusive scan of hist_s)
  uint32_t base_d[BKT_MAX + 1];  // first pair of every destination region in G's output
  uint32_t total;                // rows that reach the gather (take: n; put: rows with both indices in range)
  uint32_t use_direct;           // set by idx_locality_kernel: the index columns are local — the pipeline's kernels return at once and
                                 // the direct kernel launched behind them does the work (no host round trip)
  uint32_t loc_distinct[2], loc_rows[2], loc_done;
  uint32_t run_direct;           // 1: the direct kernel behind the pipelines does the work (take: = use_direct; put: both columns local)
  uint32_t pad[1];
};
#define BKT_GATE(g)                         \
  do {                                      \
    if ((g) && (g)->use_direct) return;     \
  } while (0)

// Locality probe (round 3): the pipelines below win against RANDOM indices — against sorted, sequential, clustered or
// few-valued ones the direct kernels run at streaming speed (take of 2^27 sorted rows: 0.36 ms direct, 2.0 ms through the
// pipeline; tools/probe/take_distributions.py), and sorted indices are what a take after a filter gets.  LOC_BLOCKS windows of
// LOC_ROWS consecutive rows, spread over the column, each count the DISTINCT lines (2^line_shift elements) their rows touch (an
// LDS hash set); the last block to finish sums up: fewer than one distinct line per two rows ⇒ ctl->use_direct = 1.  The decision
// stays on the device — the pipeline's kernels are launched either way and return at once when it is set, the direct kernel
// behind them returns at once when it is not: ≈ 60 µs of empty launches in the worst case, no host round trip, `take_op` still
// never blocks.  A put probes both of its index columns and goes direct only when both are local.
#define LOC_BLOCKS 128
#define LOC_ROWS 2048
#define LOC_SLOTS 4096
// ctl_lr != nullptr (put): the control block of the destination-only pipeline, which runs when the SOURCE column is local and the
// destination column is not (a scatter of a contiguous or sorted selection: the source side needs no partition at all)
__global__ __launch_bounds__(256) void idx_locality_kernel(const uint32_t* idx0, const uint32_t* idx1, uint64_t n, int shift0, int shift1, BktCtl* ctl,
                                                          BktCtl* ctl_lr = nullptr, BktCtl* ctl_rl = nullptr, uint32_t* host_word = nullptr,
                                                          uint32_t tag = 0) {
  __shared__ uint32_t tab[LOC_SLOTS];
  __shared__ uint32_t cnt;
  const int which = blockIdx.x >= LOC_BLOCKS ? 1 : 0;
  const uint32_t* idx = which ? idx1 : idx0;
  const int shift = which ? shift1 : shift0;
  const uint32_t b = blockIdx.x % LOC_BLOCKS;
  for (uint32_t k = threadIdx.x; k < LOC_SLOTS; k += 256) tab[k] = 0xFFFFFFFFu;
  if (threadIdx.x == 0) cnt = 0;
  __syncthreads();
  const uint64_t start = n > LOC_ROWS ? (uint64_t)b * (n - LOC_ROWS) / (LOC_BLOCKS - 1) : 0;
  const uint32_t rows = (uint32_t)(n - start < LOC_ROWS ? n - start : LOC_ROWS);
  uint32_t mine = 0;
  for (uint32_t j = threadIdx.x; j < rows; j += 256) {
    const uint32_t key = idx[start + j] >> shift;  // < 2^27: never the empty marker
    uint32_t h = (key * 2654435761u) >> 20;
    for (;;) {
      const uint32_t old = atomicCAS(&tab[h], 0xFFFFFFFFu, key);
      if (old == 0xFFFFFFFFu) {
        mine++;
        break;
      }
      if (old == key) break;
      h = (h + 1) & (LOC_SLOTS - 1);
    }
  }
  if (mine) atomicAdd(&cnt, mine);
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicAdd(&ctl->loc_distinct[which], cnt);
    atomicAdd(&ctl->loc_rows[which], rows);
    __threadfence();
    if (atomicAdd(&ctl->loc_done, 1u) == gridDim.x - 1) {  // the last block: every total above is visible
      bool loc[2] = {true, true};
      for (int w = 0; w < (idx1 ? 2 : 1); w++) {
        const uint32_t d = atomicAdd(&ctl->loc_distinct[w], 0u), r = atomicAdd(&ctl->loc_rows[w], 0u);
        loc[w] = (uint64_t)d * 2 < r;
      }
      const bool direct = loc[0] && loc[1];
      ctl->run_direct = direct ? 1u : 0u;
      // the same answer for the HOST, should it be listening (probe_decide): {tag : 28, valid : 1, -, column 1 local, column 0 local}
      if (host_word)
        __hip_atomic_store(host_word, (tag << 4) | 8u | (loc[1] ? 2u : 0u) | (loc[0] ? 1u : 0u), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      // put, four ways: both local → direct; source local only → the destination-only pipeline (ctl_lr); destination local only → the
      // take's merge-back pipeline storing through the destination column (ctl_rl); neither → the full pair pipeline (ctl)
      const bool lr = ctl_lr && loc[0] && !loc[1], rl = ctl_rl && !loc[0] && loc[1];
      if (ctl_lr) ctl_lr->use_direct = lr ? 0u : 1u;
      if (ctl_rl) ctl_rl->use_direct = rl ? 0u : 1u;
      ctl->use_direct = (direct || lr || rl) ? 1u : 0u;
    }
  }
}

// H of a put.  Source side: the count of every (tile, region) pair
// goes to `counts` (u16, row stride nbp) — the partition pass gets its range starts from a column scan over these
// instead of reserving them with global atomics: one reservation per (tile, region) is n/8 device-scope atomics per
// pass, and the chip retires ≈ 26 G of them per second (2^28 rows: 33.5 M atomics = 1.3 ms, 57 % of the pass —
// tools/probe/bkt_phases.py).  Destination side (put): region totals only, accumulated in LDS across the block's tiles.
__global__ __launch_bounds__(BKT_T) void bkt_hist_kernel(const uint32_t* si, const uint32_t* di, uint64_t n, uint64_t n_src,
                                                        uint64_t n_dst, int rs, int rd, uint32_t bs, uint32_t bd,
                                                        BktCtl* ctl, uint32_t* flags, uint16_t* counts, uint32_t nbp,
                                                        uint32_t ntiles, int tile_quads = BKT_E / 4) {  // tile = tile_quads · 4 · BKT_T rows
  if (ctl->use_direct) return;  // the locality probe chose the direct kernel launched behind this pipeline
  __shared__ uint32_t ls[BKT_MAX + 1], ld[BKT_MAX + 1];
  for (uint32_t b = threadIdx.x; b <= BKT_MAX; b += BKT_T) ld[b] = 0;
  bool bad = false;
  auto count = [&](uint32_t s, uint32_t d) {
    if (s < n_src && d < n_dst) {
      atomicAdd(&ls[s >> rs], 1u);
      atomicAdd(&ld[d >> rd], 1u);
    } else {
      bad = true;
    }
  };
  for (uint32_t b = threadIdx.x; b <= BKT_MAX; b += BKT_T) ls[b] = 0;
  __syncthreads();
  for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    for (uint32_t b = threadIdx.x; b < nbp; b += BKT_T) ls[b] = 0;
    __syncthreads();
    const uint64_t base = (uint64_t)tile * ((uint64_t)tile_quads * 4 * BKT_T);
#pragma unroll 4
    for (int q = 0; q < tile_quads; q++) {
      const uint64_t i0 = base + ((uint64_t)q * BKT_T + threadIdx.x) * 4;
      if (i0 + 4 <= n) {
        const u32x4 sv = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(si + i0));
        const u32x4 dv = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(di + i0));
        count(sv.x, dv.x); count(sv.y, dv.y); count(sv.z, dv.z); count(sv.w, dv.w);
      } else {
        for (int k = 0; k < 4; k++)
          if (i0 + k < n) count(si[i0 + k], di[i0 + k]);
      }
    }
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < nbp; b += BKT_T) counts[(uint64_t)tile * nbp + b] = (uint16_t)ls[b];
    __syncthreads();
  }
  __syncthreads();
  for (uint32_t b = threadIdx.x; b < bd; b += BKT_T)
    if (ld[b]) atomicAdd(&ctl->hist_d[b], ld[b]);
  (void)bs;
  if (bad) *reinterpret_cast<volatile uint32_t*>(flags) = AGPU_FLAG_INDEX_RANGE;
}

// Column scan of the (tile × region) count matrix in three small kernels (the matrix is 2 B per 8 rows of input):
// chunk sums over BKT_CHUNK tiles → per-region exclusive scan over chunks (+ region totals) → [bkt_scan_kernel turns the
// totals into region bases] → per-tile range starts.
#define BKT_CHUNK 128
__global__ __launch_bounds__(256) void bkt_colsum_kernel(const uint16_t* counts, uint32_t nbp, uint32_t ntiles, uint32_t* csum, const BktCtl* gate = nullptr) {
  BKT_GATE(gate);
  const uint32_t b = blockIdx.x * 256 + threadIdx.x, c = blockIdx.y;
  if (b >= nbp) return;
  const uint32_t t0 = c * BKT_CHUNK, t1 = t0 + BKT_CHUNK < ntiles ? t0 + BKT_CHUNK : ntiles;
  uint32_t acc = 0;
  for (uint32_t t = t0; t < t1; t++) acc += counts[(uint64_t)t * nbp + b];
  csum[(uint64_t)c * nbp + b] = acc;
}
__global__ __launch_bounds__(256) void bkt_colscan_kernel(uint32_t* csum, uint32_t nbp, uint32_t nchunks, uint32_t* totals, const BktCtl* gate = nullptr) {
  BKT_GATE(gate);
  const uint32_t b = blockIdx.x * 256 + threadIdx.x;
  if (b >= nbp) return;
  uint32_t run = 0;
  for (uint32_t c = 0; c < nchunks; c++) {
    const uint32_t v = csum[(uint64_t)c * nbp + b];
    csum[(uint64_t)c * nbp + b] = run;
    run += v;
  }
  if (totals && b <= BKT_MAX) totals[b] = run;
}
__global__ __launch_bounds__(256) void bkt_offsets_kernel(const uint16_t* counts, const uint32_t* csum, uint32_t nbp,
                                                         uint32_t ntiles, const uint32_t* base, uint32_t* offsets, const BktCtl* gate = nullptr) {
  BKT_GATE(gate);
  const uint32_t b = blockIdx.x * 256 + threadIdx.x, c = blockIdx.y;
  if (b >= nbp) return;
  const uint32_t t0 = c * BKT_CHUNK, t1 = t0 + BKT_CHUNK < ntiles ? t0 + BKT_CHUNK : ntiles;
  uint32_t run = (b <= BKT_MAX ? base[b] : 0u) + csum[(uint64_t)c * nbp + b];
  for (uint32_t t = t0; t < t1; t++) {
    offsets[(uint64_t)t * nbp + b] = run;
    run += counts[(uint64_t)t * nbp + b];
  }
}

// Where cursor b of a pass lives.  Plain: word b · stride.  PAIRED (stride's top bit): cursors 2j and 2j + 1 share one aligned 8-byte
// word at (stride & 0x7fffffff) · j — one 64-bit fetch-add reserves both ranges (round 4: G's reservations are bound by how many
// atomics the chip retires, so two ranges per atomic; neither half can carry into the other: a cursor never exceeds the row count < 2^32)
#define BKT_CUR_PAIRED 0x80000000u
__host__ __device__ __forceinline__ uint32_t bkt_cur_index(uint32_t b, uint32_t stride) {
  return (stride & BKT_CUR_PAIRED) ? (b >> 1) * (stride & ~BKT_CUR_PAIRED) + (b & 1u) : b * stride;
}
// exclusive scans → range start of every bucket; one workgroup
__global__ __launch_bounds__(BKT_T) void bkt_scan_kernel(BktCtl* ctl, uint32_t bs, uint32_t bd, uint32_t stride_s, uint32_t stride_d) {
  if (ctl->use_direct) return;
  __shared__ uint32_t sh[BKT_MAX + 2];
  __shared__ uint32_t wtot[BKT_T / AGPU_WAVE];
  {  // exclusive scan of hist_s[0 .. bs]: thread t owns entries 4t .. 4t+3, the last entry (bs == BKT_MAX) is thread 0's extra
     // (a thread-0 loop over 2049 entries was 30 µs of a 3.5 ms take)
    uint32_t c[4], sum = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const uint32_t b = threadIdx.x * 4 + k;
      c[k] = b <= bs && b < BKT_MAX ? ctl->hist_s[b] : 0u;
      sum += c[k];
    }
    const uint32_t lane = threadIdx.x & (AGPU_WAVE - 1), wave = threadIdx.x / AGPU_WAVE;
    uint32_t incl = sum;
#pragma unroll
    for (int off = 1; off < AGPU_WAVE; off <<= 1) {
      const uint32_t o = (uint32_t)__shfl_up((int)incl, off);
      if (lane >= (uint32_t)off) incl += o;
    }
    if (lane == AGPU_WAVE - 1) wtot[wave] = incl;
    __syncthreads();
    uint32_t pre = 0;
    for (uint32_t w = 0; w < wave; w++) pre += wtot[w];
    uint32_t run = pre + incl - sum;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const uint32_t b = threadIdx.x * 4 + k;
      if (b <= bs && b < BKT_MAX) sh[b] = run;
      run += c[k];
    }
    if (threadIdx.x == BKT_T - 1) {
      uint32_t acc = run;  // Σ of entries 0 .. BKT_MAX−1
      if (bs == BKT_MAX) {
        sh[BKT_MAX] = acc;
        acc += ctl->hist_s[BKT_MAX];
      }
      ctl->total = acc;
    }
  }
  __syncthreads();
  for (uint32_t b = threadIdx.x; b <= bs; b += BKT_T) {
    ctl->base_s[b] = sh[b];
    ctl->cur_s[bkt_cur_index(b, stride_s)] = sh[b];
  }
  __syncthreads();
  if (bd) {  // the same scan over hist_d[0 .. bd-1] (bd ≤ BKT_MAX); bd == 0: the caller has no destination side (the take pipelines)
    uint32_t c[4], sum = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const uint32_t b = threadIdx.x * 4 + k;
      c[k] = b < bd ? ctl->hist_d[b] : 0u;
      sum += c[k];
    }
    const uint32_t lane = threadIdx.x & (AGPU_WAVE - 1), wave = threadIdx.x / AGPU_WAVE;
    uint32_t incl = sum;
#pragma unroll
    for (int off = 1; off < AGPU_WAVE; off <<= 1) {
      const uint32_t o = (uint32_t)__shfl_up((int)incl, off);
      if (lane >= (uint32_t)off) incl += o;
    }
    if (lane == AGPU_WAVE - 1) wtot[wave] = incl;
    __syncthreads();
    uint32_t pre = 0;
    for (uint32_t w = 0; w < wave; w++) pre += wtot[w];
    uint32_t run = pre + incl - sum;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const uint32_t b = threadIdx.x * 4 + k;
      if (b < bd) {
        ctl->cur_d[bkt_cur_index(b, stride_d)] = run;
        ctl->base_d[b] = run;
      }
      run += c[k];
    }
  }
}

// Tile-local counting sort shared by P, G and F: every thread holds BKT_E rows {payload a, payload b, key}; rows with
// key == BKT_INVALID are dropped.  The tile's pairs end up in LDS ordered by key (`sorted`), `lcnt[k]` holds the
// EXCLUSIVE start of key k inside the tile and `*tile_rows` the number of kept rows — so the caller's copy-out loop lets
// consecutive lanes write consecutive pairs: a bucket's rows leave as one contiguous run (a handful of memory requests
// per wave store instead of 64).  Scattered 4–8-byte accesses retire at ≈100 G requests/s chip-wide even when every
// one of them hits L2, which is what bounded the first version of these kernels (profiles/r02_gather_passes.json).
struct BktRow {
  uint32_t a, b, key;
};
static_assert(offsetof(BktCtl, cur_d) % 8 == 0 && offsetof(BktCtl, cur_s) % 8 == 0 && BKT_CUR_STRIDE % 2 == 0, "paired cursors are 8-byte words");
__device__ __forceinline__ void bkt_tile_sort(BktRow (&row)[BKT_E], uint32_t nkeys, uint32_t* lcnt, u32x2* sorted,
                                              uint32_t* wave_tot, uint32_t* tile_rows) {
  // nkeys ≤ BKT_MAX = 4 · BKT_T: thread t owns counters 4t .. 4t+3
  for (uint32_t k = threadIdx.x; k < BKT_MAX; k += BKT_T) lcnt[k] = 0;
  __syncthreads();
  uint32_t rank[BKT_E];
#pragma unroll
  for (int e = 0; e < BKT_E; e++) rank[e] = row[e].key != BKT_INVALID ? atomicAdd(&lcnt[row[e].key], 1u) : 0u;
  __syncthreads();
  // exclusive scan of the counters
  const uint32_t lane = threadIdx.x & (AGPU_WAVE - 1), wave = threadIdx.x / AGPU_WAVE;
  uint32_t c[4], sum = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    c[k] = lcnt[threadIdx.x * 4 + k];
    sum += c[k];
  }
  uint32_t incl = sum;
#pragma unroll
  for (int off = 1; off < AGPU_WAVE; off <<= 1) {
    const uint32_t o = (uint32_t)__shfl_up((int)incl, off);
    if (lane >= (uint32_t)off) incl += o;
  }
  if (lane == AGPU_WAVE - 1) wave_tot[wave] = incl;
  __syncthreads();
  uint32_t base = 0;
  for (uint32_t w = 0; w < wave; w++) base += wave_tot[w];
  uint32_t run = base + incl - sum;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    lcnt[threadIdx.x * 4 + k] = run;
    run += c[k];
  }
  if (threadIdx.x == BKT_T - 1) *tile_rows = run;
  __syncthreads();
#pragma unroll
  for (int e = 0; e < BKT_E; e++)
    if (row[e].key != BKT_INVALID) {
      u32x2 v = {row[e].a, row[e].b};
      sorted[lcnt[row[e].key] + rank[e]] = v;
    }
  __syncthreads();
  (void)nkeys;
}

// copy the sorted tile out: key k's rows go to out_pairs[tile_starts[k] ...] — the range starts come from the column scan of H's counts
// (deterministic; rounds 2–5 also carried a form that reserved them with global atomics: n/8 device-scope atomics per pass at ≈ 26 G/s, and
// never faster).  On entry lcnt = exclusive starts inside the tile; on exit lcnt[k] = tile_starts[k] − start[k] (wrapping).
template <typename KeyOf>
__device__ __forceinline__ void bkt_copy_out(uint32_t nkeys, uint32_t* lcnt, const u32x2* sorted, uint32_t tile_rows, u32x2* out_pairs,
                                             KeyOf key_of, const uint32_t* tile_starts) {
  uint32_t st[4], cnt[4];
#pragma unroll
  for (int k = 0; k < 4; k++) st[k] = lcnt[threadIdx.x * 4 + k];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const uint32_t kk = threadIdx.x * 4 + k;
    const uint32_t nxt = k < 3 ? st[k + 1] : (kk + 1 < BKT_MAX ? lcnt[kk + 1] : tile_rows);
    cnt[k] = nxt - st[k];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const uint32_t kk = threadIdx.x * 4 + k;
    if (kk < nkeys && cnt[k]) lcnt[kk] = tile_starts[kk] - st[k];
  }
  __syncthreads();
  for (uint32_t j = threadIdx.x; j < tile_rows; j += BKT_T) {
    const u32x2 v = sorted[j];
    out_pairs[(uint64_t)(uint32_t)(lcnt[key_of(v)] + j)] = v;
  }
}

#define BKT_LDS_DECL                                                   \
  __shared__ u32x2 sorted[BKT_TILE];                                   \
  __shared__ uint32_t lcnt[BKT_MAX];                                   \
  __shared__ uint32_t wave_tot[BKT_T / AGPU_WAVE];                     \
  __shared__ uint32_t tile_rows

// XCD-contiguous walk: workgroups are dealt round-robin to the 8 XCDs, so workgroup j takes tile (j % 8) · per + j / 8 —
// each XCD streams one contiguous eighth of the bucket-ordered list and its L2 holds the one or two regions in flight
__device__ __forceinline__ bool bkt_tile_of_block(uint32_t ntiles, uint64_t* tile) {
  const uint32_t per = (ntiles + 7) / 8;
  const uint64_t t = (uint64_t)(blockIdx.x % 8) * per + blockIdx.x / 8;
  *tile = t;
  return (blockIdx.x / 8) < per && t < ntiles;
}

// P over 16 Ki-row tiles: rows in natural order → pairs {source index, destination} in source-bucket order; rows with either index out of
// range are dropped.  What is left of it since round 4 is the DESTINATION-ONLY pipeline of a put (the full pipeline partitions with
// bkt_partition2_kernel): `si` is then the destination column, `di` the (local) source column, and the pair carries the VALUE (gvals).
__global__ __launch_bounds__(BKT_T) void bkt_partition_kernel(const uint32_t* si, const uint32_t* di, uint64_t n,
                                                             uint64_t n_src, uint64_t n_dst, int rs, uint32_t bs,
                                                             BktCtl* ctl, u32x2* pairs, const uint32_t* offsets, uint32_t nbp,
                                                             uint32_t ntiles, const void* gvals = nullptr, int gw = 0) {
  if (ctl->use_direct) return;  // the locality probe chose the direct kernel launched behind this pipeline
  BKT_LDS_DECL;
  // XCD-contiguous tiles (round 3): with range starts from the column scan, the runs of tiles t and t + 1 are neighbours in
  // every region's range — one XCD handles both a few dispatches apart and their 64-byte halves meet in its L2
  uint64_t tile64;
  if (!bkt_tile_of_block(ntiles, &tile64)) return;
  const uint64_t base = tile64 * BKT_TILE;
  BktRow row[BKT_E];
#pragma unroll
  for (int q = 0; q < BKT_E / 4; q++) {
    const uint64_t i0 = base + ((uint64_t)q * BKT_T + threadIdx.x) * 4;
    uint32_t s[4] = {0, 0, 0, 0}, d[4] = {0, 0, 0, 0};
    if (i0 + 4 <= n) {
      const u32x4 t = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(si + i0));
      s[0] = t.x; s[1] = t.y; s[2] = t.z; s[3] = t.w;
      const u32x4 u = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(di + i0));
      d[0] = u.x; d[1] = u.y; d[2] = u.z; d[3] = u.w;
    } else {
      for (int k = 0; k < 4; k++)
        if (i0 + k < n) {
          s[k] = si[i0 + k];
          d[k] = di[i0 + k];
        }
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
      BktRow& r = row[q * 4 + k];
      const uint64_t i = i0 + k;
      r.a = s[k];
      r.b = d[k];
      r.key = (i < n && s[k] < n_src && d[k] < n_dst) ? (s[k] >> rs) : BKT_INVALID;
      if (gvals && r.key != BKT_INVALID) {  // the destination-only pipeline of a put: `si` is the DESTINATION column here, `di` the (local)
                                            // source column — the pair carries the value itself, fetched with a near-streaming gather
        r.b = gw == 4 ? static_cast<const uint32_t*>(gvals)[d[k]] : gw == 2 ? (uint32_t)static_cast<const uint16_t*>(gvals)[d[k]]
                                                                          : (uint32_t)static_cast<const uint8_t*>(gvals)[d[k]];
      }
    }
  }
  bkt_tile_sort(row, bs + 1, lcnt, sorted, wave_tot, &tile_rows);
  const uint32_t n_src32 = (uint32_t)(n_src > 0xFFFFFFFFull ? 0xFFFFFFFFull : n_src);
  bkt_copy_out(bs + 1, lcnt, sorted, tile_rows, pairs, [=](const u32x2& v) { return v.x < n_src32 ? (v.x >> rs) : bs; }, offsets + tile64 * nbp);
}

__device__ __forceinline__ void bkt_load_tile(const u32x2* pairs_in, uint64_t base, uint64_t total, uint32_t (&pa)[BKT_E],
                                              uint32_t (&pb)[BKT_E], bool (&live)[BKT_E]) {
#pragma unroll
  for (int q = 0; q < BKT_E / 2; q++) {
    const uint64_t i0 = base + ((uint64_t)q * BKT_T + threadIdx.x) * 2;
    u32x4 t = {0, 0, 0, 0};
    if (i0 + 2 <= total) t = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(pairs_in + i0));
    else if (i0 < total) {
      const u32x2 one = pairs_in[i0];
      t.x = one.x; t.y = one.y;
    }
    pa[q * 2] = t.x; pb[q * 2] = t.y; live[q * 2] = i0 < total;
    pa[q * 2 + 1] = t.z; pb[q * 2 + 1] = t.w; live[q * 2 + 1] = i0 + 1 < total;
  }
}

// G: pairs in source-bucket order → {destination, value} in destination-bucket order, over 32 Ki-pair tiles (round 4; rounds 2–3 ran a
// 16 Ki-pair form through two tile sorts of 8-byte pairs, removed in round 6 with the pair-pipeline take it still served).  A 32 Ki-pair tile does not fit LDS as
// 8-byte pairs; it does as ONE 4-byte array (128 KiB + 16 KiB of counters), 1024 threads holding 32 rows each:
//   sort 1 (by source line)       the array receives the SOURCE INDEX only; every thread remembers where its rows went; the lane that
//                                 finds an index at position j gathers the value and puts it back at j; the owner picks it up from there;
//   sort 2 (by destination region) the values travel first (array → lane j keeps value j in a register), the destinations second;
//                                 lane j then holds pair j of the sorted tile and stores it as one 8-byte word.
// Ranks, positions and range deltas live in the one counter array, one after the other.  What the bigger tile buys: a (tile, region) run is
// 16 pairs = one full 128-byte line instead of half of one, half as many range reservations per row, eight rows per source line and
// gather instruction instead of four.  (The form was first built for 16 Ki-pair tiles at two workgroups per CU — 13 % SLOWER than
// the two-sort kernel: docs/experiments.md R4.4, tools/probe/patches/.)  117–119 VGPRs, no scratch — see BK2_PIN.
#define BK2_T 1024
#define BK2_TILE 32768  // pairs per tile: twice P's and F's
#define BK2_E (BK2_TILE / BK2_T)
#define BK2_K (BKT_MAX / BK2_T)  // counters per thread in the scans
#define BK2_PIN(x) asm volatile("" : "+v"(x))  // the value is materialised at this point of the program and nothing is known about it afterwards
// exclusive scan of C[0 .. BKT_MAX) in place; `scratch` = BK2_T / 64 words nobody else uses right now; returns nothing (callers know the total)
__device__ __forceinline__ void bk2_scan(uint32_t* C, uint32_t* scratch) {
  const uint32_t lane = threadIdx.x & (AGPU_WAVE - 1), wave = threadIdx.x / AGPU_WAVE;
  uint32_t c[BK2_K], sum = 0;
#pragma unroll
  for (int k = 0; k < BK2_K; k += 4) {
    const u32x4 v = *reinterpret_cast<const u32x4*>(&C[threadIdx.x * BK2_K + k]);
    c[k] = v.x; c[k + 1] = v.y; c[k + 2] = v.z; c[k + 3] = v.w;
    sum += v.x + v.y + v.z + v.w;
  }
  uint32_t incl = sum;
#pragma unroll
  for (int off = 1; off < AGPU_WAVE; off <<= 1) {
    const uint32_t o = (uint32_t)__shfl_up((int)incl, off);
    if (lane >= (uint32_t)off) incl += o;
  }
  if (lane == AGPU_WAVE - 1) scratch[wave] = incl;
  __syncthreads();
  uint32_t run = incl - sum;
  for (uint32_t w = 0; w < wave; w++) run += scratch[w];
#pragma unroll
  for (int k = 0; k < BK2_K; k += 4) {
    u32x4 v;
    v.x = run; run += c[k];
    v.y = run; run += c[k + 1];
    v.z = run; run += c[k + 2];
    v.w = run; run += c[k + 3];
    *reinterpret_cast<u32x4*>(&C[threadIdx.x * BK2_K + k]) = v;
  }
  __syncthreads();
}

// one tile; FULL: all BK2_TILE pairs are live (every tile but the list's last one) — no per-row predicates
template <int W, bool FULL>
__device__ __forceinline__ void bk2_tile(const typename ElemOf<W>::type* values, uint32_t n_src32, const u32x2* pairs_in, uint32_t rows, int rd,
                                         uint32_t bd, int src_line_shift, uint32_t* cursors, uint32_t cur_stride, u32x2* pairs_out, uint32_t* A, uint32_t* C) {
  uint32_t s[BK2_E], d[BK2_E], r[BK2_E / 2];  // r: two 14-bit ranks / positions per register (rows 2q and 2q + 1)
  auto live = [&](int e) { return FULL || (((uint32_t)(e / 2) * BK2_T + threadIdx.x) * 2 + (uint32_t)(e % 2)) < rows; };
  for (uint32_t k = threadIdx.x; k < BKT_MAX; k += BK2_T) C[k] = 0;
  // row e of this thread is pair (e / 2 · BK2_T + thread) · 2 + e % 2 of the tile
#pragma unroll
  for (int q = 0; q < BK2_E / 2; q++) {
    const uint32_t l0 = ((uint32_t)q * BK2_T + threadIdx.x) * 2;
    u32x4 t = {0, 0, 0, 0};
    if (FULL || l0 + 2 <= rows) t = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(pairs_in + l0));
    else if (l0 < rows) {
      const u32x2 one = pairs_in[l0];
      t.x = one.x; t.y = one.y;
    }
    s[q * 2] = t.x; d[q * 2] = t.y;
    s[q * 2 + 1] = t.z; d[q * 2 + 1] = t.w;
  }
  __syncthreads();
  // ---- sort 1: by source line
  auto key1 = [&](uint32_t v) { return (v >> src_line_shift) & (uint32_t)(BKT_MAX - 1); };
#pragma unroll
  for (int q = 0; q < BK2_E / 2; q++) {
    const uint32_t lo = live(2 * q) ? atomicAdd(&C[key1(s[2 * q])], 1u) : 0u;
    const uint32_t hi = live(2 * q + 1) ? atomicAdd(&C[key1(s[2 * q + 1])], 1u) : 0u;
    r[q] = lo | (hi << 16);
    BK2_PIN(r[q]);  // the packed word exists HERE (two ranks per register; else the compiler keeps 32 results and 32 counter addresses alive
                    // across the scan and spills the destinations)
  }
#pragma unroll
  for (int e = 0; e < BK2_E; e++) BK2_PIN(s[e]);
  __syncthreads();
  bk2_scan(C, A);
#pragma unroll
  for (int q = 0; q < BK2_E / 2; q++) {
    const uint32_t lo = (r[q] & 0xFFFFu) + C[key1(s[2 * q])], hi = (r[q] >> 16) + C[key1(s[2 * q + 1])];
    r[q] = lo | (hi << 16);
    BK2_PIN(r[q]);
  }
  __syncthreads();  // the scan's scratch words (A[0 .. 8)) have been read by everyone
#pragma unroll
  for (int e = 0; e < BK2_E; e++)
    if (live(e)) A[(r[e / 2] >> (16 * (e % 2))) & 0xFFFFu] = s[e];
  __syncthreads();
  // the L2-resident gather, in line order; the value replaces the index at its place.  Eight rows per step: thirty-two 64-bit addresses
  // at once would not fit beside the rows' destinations in 128 registers (all 32 in flight through a scalar base + 32-bit offsets was
  // built and measured: no faster — the phase is not waiting on the round trips)
  for (uint32_t k = threadIdx.x; k < BKT_MAX; k += BK2_T) C[k] = 0;  // (everyone is past the position lookups: two barriers ago)
#pragma unroll 1
  for (uint32_t c = 0; c < BK2_E; c += 8) {
    uint32_t v[8];
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const uint32_t j = (c + (uint32_t)u) * BK2_T + threadIdx.x;
      v[u] = (FULL || j < rows) ? A[j] : 0xFFFFFFFFu;
    }
#pragma unroll
    for (int u = 0; u < 8; u++) v[u] = v[u] < n_src32 ? (uint32_t)values[v[u]] : 0u;
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const uint32_t j = (c + (uint32_t)u) * BK2_T + threadIdx.x;
      if (FULL || j < rows) A[j] = v[u];
    }
  }
  __syncthreads();
#pragma unroll
  for (int e = 0; e < BK2_E; e++)
    if (live(e)) s[e] = A[(r[e / 2] >> (16 * (e % 2))) & 0xFFFFu];  // the value of MY row e
  // ---- sort 2: by destination region
#pragma unroll
  for (int q = 0; q < BK2_E / 2; q++) {
    const uint32_t lo = live(2 * q) ? atomicAdd(&C[d[2 * q] >> rd], 1u) : 0u;
    const uint32_t hi = live(2 * q + 1) ? atomicAdd(&C[d[2 * q + 1] >> rd], 1u) : 0u;
    r[q] = lo | (hi << 16);
    BK2_PIN(r[q]);
  }
#pragma unroll
  for (int e = 0; e < BK2_E; e++) BK2_PIN(d[e]);
  __syncthreads();  // … which also ends the reads of A above: the scan may use its first words
  bk2_scan(C, A);
  // ranges of the tile's runs in the output: thread t owns keys BK2_K·t …; one reservation per non-empty (tile, region), all of them in
  // flight while the tile is moved
  uint32_t g[BK2_K];
  {
    static_assert(BK2_K == 4, "two paired reservations per thread");
    uint32_t st[BK2_K], cnt[BK2_K];
#pragma unroll
    for (int k = 0; k < BK2_K; k++) st[k] = C[threadIdx.x * BK2_K + k];
#pragma unroll
    for (int k = 0; k < BK2_K; k++) {
      const uint32_t kk = threadIdx.x * BK2_K + k;
      const uint32_t nxt = k < BK2_K - 1 ? st[k + 1] : (kk + 1 < BKT_MAX ? C[kk + 1] : rows);
      cnt[k] = kk < bd ? nxt - st[k] : 0u;
    }
    {  // two ranges per 64-bit atomic (keys 4t, 4t+1 | 4t+2, 4t+3: cursors 2j and 2j + 1 share one aligned 8-byte word — bkt_cur_index)
      const uint32_t S = cur_stride & ~BKT_CUR_PAIRED;
      unsigned long long old[2] = {0ull, 0ull};
#pragma unroll
      for (int q = 0; q < 2; q++) {
        const uint32_t kk = threadIdx.x * 4 + 2 * q;
        if (cnt[2 * q] | cnt[2 * q + 1])
          old[q] = __hip_atomic_fetch_add(reinterpret_cast<unsigned long long*>(cursors + (size_t)(kk >> 1) * S),
                                          (unsigned long long)cnt[2 * q] | ((unsigned long long)cnt[2 * q + 1] << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      g[0] = (uint32_t)old[0] - st[0]; g[1] = (uint32_t)(old[0] >> 32) - st[1];
      g[2] = (uint32_t)old[1] - st[2]; g[3] = (uint32_t)(old[1] >> 32) - st[3];
    }
  }
#pragma unroll
  for (int q = 0; q < BK2_E / 2; q++) {
    const uint32_t lo = (r[q] & 0xFFFFu) + (live(2 * q) ? C[d[2 * q] >> rd] : 0u), hi = (r[q] >> 16) + (live(2 * q + 1) ? C[d[2 * q + 1] >> rd] : 0u);
    r[q] = lo | (hi << 16);
    BK2_PIN(r[q]);
  }
  __syncthreads();  // everyone has its positions and starts
#pragma unroll
  for (int e = 0; e < BK2_E; e++)
    if (live(e)) A[(r[e / 2] >> (16 * (e % 2))) & 0xFFFFu] = s[e];  // values first …
#pragma unroll
  for (int k = 0; k < BK2_K; k++) C[threadIdx.x * BK2_K + k] = g[k];  // the counters turn into deltas: range start − start inside the tile
  __syncthreads();
#pragma unroll
  for (int e = 0; e < BK2_E; e++) s[e] = A[(uint32_t)e * BK2_T + threadIdx.x];  // … value j of the sorted tile stays with lane j
  __syncthreads();
#pragma unroll
  for (int e = 0; e < BK2_E; e++)
    if (live(e)) A[(r[e / 2] >> (16 * (e % 2))) & 0xFFFFu] = d[e];  // … then the destinations
  __syncthreads();
#pragma unroll
  for (int e = 0; e < BK2_E; e++) {
    const uint32_t j = (uint32_t)e * BK2_T + threadIdx.x;
    if (FULL || j < rows) {
      const uint32_t dj = A[j];
      const u32x2 v = {dj, s[e]};
      pairs_out[(uint64_t)(uint32_t)(C[dj >> rd] + j)] = v;
    }
    if (e % 8 == 7) __builtin_amdgcn_sched_barrier(0);  // (eight addresses at a time)
  }
}

template <int W>
__global__ __launch_bounds__(BK2_T, 4) void bkt_gather2_kernel(const typename ElemOf<W>::type* values, uint64_t n_src,
                                                              const u32x2* pairs_in, int rd, uint32_t bd, uint32_t ntiles,
                                                              int src_line_shift, BktCtl* ctl, u32x2* pairs_out, uint32_t cur_stride) {
  if (ctl->use_direct) return;  // the locality probe chose the direct kernel launched behind this pipeline
  static_assert(BK2_T * BK2_E == BK2_TILE && BK2_K % 4 == 0 && BK2_E % 2 == 0 && BK2_TILE <= 65536, "tile shape (positions travel as 16-bit halves)");
  __shared__ __attribute__((aligned(16))) uint32_t A[BK2_TILE];
  __shared__ __attribute__((aligned(16))) uint32_t C[BKT_MAX];
  uint64_t tile;
  if (!bkt_tile_of_block(ntiles, &tile)) return;
  const uint64_t total = ctl->total, base = tile * BK2_TILE;
  if (base >= total) return;
  const uint32_t n_src32 = (uint32_t)(n_src > 0xFFFFFFFFull ? 0xFFFFFFFFull : n_src);
  if (total - base >= BK2_TILE)  // every pair of P's output is a live row: only the list's last tile is ragged
    bk2_tile<W, true>(values, n_src32, pairs_in + base, BK2_TILE, rd, bd, src_line_shift, ctl->cur_d, cur_stride, pairs_out, A, C);
  else
    bk2_tile<W, false>(values, n_src32, pairs_in + base, (uint32_t)(total - base), rd, bd, src_line_shift, ctl->cur_d, cur_stride, pairs_out, A, C);
}

// P over 32 Ki-row tiles (round 4; put only): rows in natural order → pairs {source index, destination} in source-region order, the tile
// sent through the ONE 4-byte array like G's second sort (sources first, destinations second).  A (tile, region) run is 16 pairs — one
// whole 128-byte line — where bkt_partition_kernel writes halves of lines and relies on the neighbouring tile's half meeting it in L2.
// Range starts from the column scan of H's counts (H counts the same 32 Ki-row tiles).  Rows with either index out of range are dropped.
template <bool FULL>
__device__ __forceinline__ void bk2_partition_tile(const uint32_t* si, const uint32_t* di, uint32_t rows, uint32_t n_src32, uint32_t n_dst32, int rs,
                                                   const uint32_t* tile_starts, uint32_t nbp, u32x2* pairs, uint32_t* A, uint32_t* C) {
  uint32_t s[BK2_E], d[BK2_E], r[BK2_E / 2];
  for (uint32_t k = threadIdx.x; k < BKT_MAX; k += BK2_T) C[k] = 0;
  // row e of this thread is row (e / 4 · BK2_T + thread) · 4 + e % 4 of the tile
#pragma unroll
  for (int q = 0; q < BK2_E / 4; q++) {
    const uint32_t l0 = ((uint32_t)q * BK2_T + threadIdx.x) * 4;
    u32x4 a = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, b = a;  // (an index of 2^32 − 1 is out of range for every array)
    if (FULL || l0 + 4 <= rows) {
      a = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(si + l0));
      b = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(di + l0));
    } else {
      if (l0 < rows) { a.x = si[l0]; b.x = di[l0]; }
      if (l0 + 1 < rows) { a.y = si[l0 + 1]; b.y = di[l0 + 1]; }
      if (l0 + 2 < rows) { a.z = si[l0 + 2]; b.z = di[l0 + 2]; }
    }
    s[q * 4] = a.x; s[q * 4 + 1] = a.y; s[q * 4 + 2] = a.z; s[q * 4 + 3] = a.w;
    d[q * 4] = b.x; d[q * 4 + 1] = b.y; d[q * 4 + 2] = b.z; d[q * 4 + 3] = b.w;
  }
  auto kept = [&](int e) { return s[e] < n_src32 && d[e] < n_dst32; };
  __syncthreads();
#pragma unroll
  for (int q = 0; q < BK2_E / 2; q++) {
    const uint32_t lo = kept(2 * q) ? atomicAdd(&C[s[2 * q] >> rs], 1u) : 0u;
    const uint32_t hi = kept(2 * q + 1) ? atomicAdd(&C[s[2 * q + 1] >> rs], 1u) : 0u;
    r[q] = lo | (hi << 16);
    BK2_PIN(r[q]);
  }
#pragma unroll
  for (int e = 0; e < BK2_E; e++) BK2_PIN(s[e]);
  __syncthreads();
  uint32_t own_total = 0;  // rows kept in the tile = the last counter's start + its count: thread BK2_T − 1 has it after the scan
  if (threadIdx.x == BK2_T - 1) own_total = C[BKT_MAX - 1];
  bk2_scan(C, A);
  if (threadIdx.x == BK2_T - 1) A[BK2_TILE - 1] = own_total + C[BKT_MAX - 1];  // (the scan's scratch is A[0 .. 16); nobody uses A's last word yet)
#pragma unroll
  for (int q = 0; q < BK2_E / 2; q++) {
    const uint32_t lo = (r[q] & 0xFFFFu) + (kept(2 * q) ? C[s[2 * q] >> rs] : 0u), hi = (r[q] >> 16) + (kept(2 * q + 1) ? C[s[2 * q + 1] >> rs] : 0u);
    r[q] = lo | (hi << 16);
    BK2_PIN(r[q]);
  }
  // the counters turn into deltas: range start of (tile, region) − start inside the tile
  uint32_t g[BK2_K];
#pragma unroll
  for (int k = 0; k < BK2_K; k++) g[k] = (threadIdx.x * BK2_K + k < nbp ? tile_starts[threadIdx.x * BK2_K + k] : 0u) - C[threadIdx.x * BK2_K + k];
  __syncthreads();  // everyone has its positions (and the kept-row count is in place)
  const uint32_t tile_rows = A[BK2_TILE - 1];
  __syncthreads();
#pragma unroll
  for (int e = 0; e < BK2_E; e++)
    if (kept(e)) A[(r[e / 2] >> (16 * (e % 2))) & 0xFFFFu] = s[e];  // sources first …
#pragma unroll
  for (int k = 0; k < BK2_K; k++) C[threadIdx.x * BK2_K + k] = g[k];
  __syncthreads();
  uint32_t sj[BK2_E];
#pragma unroll
  for (int e = 0; e < BK2_E; e++) sj[e] = A[(uint32_t)e * BK2_T + threadIdx.x];  // … source j of the sorted tile stays with lane j
  __syncthreads();
#pragma unroll
  for (int e = 0; e < BK2_E; e++)
    if (kept(e)) A[(r[e / 2] >> (16 * (e % 2))) & 0xFFFFu] = d[e];  // … then the destinations
  __syncthreads();
#pragma unroll
  for (int e = 0; e < BK2_E; e++) {
    const uint32_t j = (uint32_t)e * BK2_T + threadIdx.x;
    if (j < tile_rows) {
      const u32x2 v = {sj[e], A[j]};
      pairs[(uint64_t)(uint32_t)(C[sj[e] >> rs] + j)] = v;
    }
    if (e % 8 == 7) __builtin_amdgcn_sched_barrier(0);  // (eight addresses at a time)
  }
}
__global__ __launch_bounds__(BK2_T, 4) void bkt_partition2_kernel(const uint32_t* si, const uint32_t* di, uint64_t n, uint64_t n_src, uint64_t n_dst, int rs,
                                                                 const BktCtl* ctl, u32x2* pairs, const uint32_t* offsets, uint32_t nbp, uint32_t ntiles) {
  if (ctl->use_direct) return;  // the locality probe chose the direct kernel launched behind this pipeline
  __shared__ __attribute__((aligned(16))) uint32_t A[BK2_TILE];
  __shared__ __attribute__((aligned(16))) uint32_t C[BKT_MAX];
  uint64_t tile;
  if (!bkt_tile_of_block(ntiles, &tile)) return;
  const uint64_t base = tile * BK2_TILE;
  const uint32_t n_src32 = (uint32_t)(n_src > 0xFFFFFFFFull ? 0xFFFFFFFFull : n_src), n_dst32 = (uint32_t)(n_dst > 0xFFFFFFFFull ? 0xFFFFFFFFull : n_dst);
  if (n - base >= BK2_TILE) bk2_partition_tile<true>(si + base, di + base, BK2_TILE, n_src32, n_dst32, rs, offsets + tile * nbp, nbp, pairs, A, C);
  else bk2_partition_tile<false>(si + base, di + base, (uint32_t)(n - base), n_src32, n_dst32, rs, offsets + tile * nbp, nbp, pairs, A, C);
}

// F: {destination, value} in destination-bucket order → dst[destination] = value.  The tile is first ordered by
// destination LINE (128 bytes) in LDS, so lanes that share a line sit next to each other and the store instruction's
// coalescer merges them: the smaller the destination region, the more rows per line in one tile.
template <int W>
__global__ __launch_bounds__(BKT_T) void bkt_store_kernel(const u32x2* pairs, uint32_t ntiles, const BktCtl* ctl, int line_shift,
                                                         typename ElemOf<W>::type* dst) {
  if (ctl->use_direct) return;  // the locality probe chose the direct kernel launched behind this pipeline
  typedef typename ElemOf<W>::type E;
  BKT_LDS_DECL;
  uint64_t tile;
  if (!bkt_tile_of_block(ntiles, &tile)) return;
  const uint64_t total = ctl->total, base = tile * BKT_TILE;
  if (base >= total) return;
  uint32_t pa[BKT_E], pb[BKT_E];
  bool live[BKT_E];
  bkt_load_tile(pairs, base, total, pa, pb, live);
  BktRow row[BKT_E];
#pragma unroll
  for (int e = 0; e < BKT_E; e++) {
    row[e].a = pa[e];
    row[e].b = pb[e];
    row[e].key = live[e] ? ((pa[e] >> line_shift) & (BKT_MAX - 1)) : BKT_INVALID;
  }
  bkt_tile_sort(row, BKT_MAX, lcnt, sorted, wave_tot, &tile_rows);
  for (uint32_t j = threadIdx.x; j < tile_rows; j += BKT_T) {
    const u32x2 v = sorted[j];
    dst[v.x] = (E)v.y;
  }
}

// region = 2^r elements = 512 KiB on both sides: an XCD has 32 tiles (2^19 rows) in flight, i.e. two or three regions,
// beside the pair streams in its 4 MiB L2.  Measured at 2^28 rows (tools/probe/bucket_sweep.py --region-bits): 512 KiB
// regions 4.8 / 5.5 ms (take / put), 1 MiB 5.2 / 5.8, 2 MiB 5.5 / 5.8, 4 MiB 5.8 / 6.4.
static int bkt_region_bits(const agpu_pipeline* p, uint64_t n_elems, int width) {
  (void)p;
  int r = 17 + (width == 2 ? 1 : width == 1 ? 2 : 0);
  while (((n_elems + ((uint64_t)1 << r) - 1) >> r) > BKT_MAX - 1) r++;
  return r;
}

static agpu_status launch_put_direct(agpu_pipeline* p, int width, const void* src, uint64_t n_src, const uint32_t* src_idx, void* dst,
                                     uint64_t n_dst, const uint32_t* dst_idx, uint64_t n, const uint32_t* only_if);
// The probe's answer on the HOST, when it is there in time: the probe is launched by itself, the host polls a pinned word (the
// pipeline's error-word slot, second word) for at most 150 µs.  On an idle stream — every default-API call has a pipeline of its own —
// the answer arrives in ≈ 25 µs and ONLY the chosen form is enqueued: no empty launches at all.  On a busy stream the wait times out
// (−1) and the caller enqueues all forms gated by the device-side copy of the same decision, as before.  Returns bit 0: column 0 local,
// bit 1: column 1 local.
static std::atomic<uint32_t> g_probe_tag{1};
static int probe_decide(agpu_pipeline* p, const uint32_t* idx0, const uint32_t* idx1, uint64_t n, int shift0, int shift1) {
  if (!p->flags || p->capturing) return -1;
  void* ctl_v = nullptr;
  if (agpu_malloc(p->dev, sizeof(BktCtl), 0, &ctl_v) != AGPU_OK) return -1;
  int result = -1;
  if (hipMemsetAsync(ctl_v, 0, sizeof(BktCtl), p->stream) == hipSuccess) {
    const uint32_t tag = g_probe_tag.fetch_add(1, std::memory_order_relaxed) & 0x0FFFFFFFu;
    volatile uint32_t* w = p->flags + 1;
    hipLaunchKernelGGL(idx_locality_kernel, dim3(idx1 ? 2 * LOC_BLOCKS : LOC_BLOCKS), dim3(256), 0, p->stream, idx0, idx1, n, shift0, shift1,
                       static_cast<BktCtl*>(ctl_v), static_cast<BktCtl*>(nullptr), static_cast<BktCtl*>(nullptr), const_cast<uint32_t*>(w), tag);
    if (hipGetLastError() == hipSuccess) {
      const auto t0 = std::chrono::steady_clock::now();
      for (;;) {
        const uint32_t v = *w;
        if ((v >> 4) == tag && (v & 8u)) {
          result = (int)(v & 3u);
          break;
        }
        if (std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(150)) break;
      }
    }
  }
  (void)agpu_free(p->dev, ctl_v);
  return result;
}

// The pair pipeline of a PUT (both index columns are data).  Returns AGPU_ERR_UNSUPPORTED when the shape does not qualify.
// (Rounds 2–5 also ran takes through it — di == nullptr, tuning gather_bucket = 3 — and carried four ways of getting the range starts,
// tuning gather_offsets; the merge-back pipeline below is 1.5× faster for takes and the defaults were never beaten: removed in round 6.)
static agpu_status launch_put_through_take(agpu_pipeline* p, int width, const void* src, uint64_t n_src, const uint32_t* si, void* dst, uint64_t n,
                                           const uint32_t* di, uint64_t n_dst, const BktCtl* gate);
// the rare case in which the destination-local variant of a put could not be enqueued after the probe was told about it: whatever
// the probe gave to that variant goes to the direct kernel
__global__ void bkt_flag_or_kernel(const BktCtl* rl, BktCtl* main) {
  if (!rl->use_direct) main->run_direct = 1u;
}
static agpu_status launch_bucketed(agpu_pipeline* p, int width, const void* src, uint64_t n_src, const uint32_t* si,
                                   void* dst, uint64_t n_dst, const uint32_t* di, uint64_t n, bool adaptive = false) {
  if (n >= 0xFFFF0000ull || n_src > 0xFFFFFFFFull || n_dst > 0xFFFFFFFFull || !aligned16(si) || !di || !aligned16(di) || p->capturing)
    return AGPU_ERR_UNSUPPORTED;
  // A put's SOURCE regions (round 4, tools/probe/put_tile_sweep.py with the region size forced, 2^25 … 2^28 rows): about a thousand of them is
  // the optimum at every size — 32-pair runs out of P's 32 Ki-row tiles, a count matrix half the size — between 256 KiB and 1 MiB each
  // (2^28 rows 4.41 → 4.34 ms, 2^26 1.11 → 1.09, 2^25 0.61 → 0.59; 2^27 rows have it already).  The destination regions keep their 512 KiB.
  int rs;
  {
    const uint64_t bytes = n_src * (uint64_t)width;
    int rb = 18;  // log2 of the region's bytes
    while (rb < 20 && (bytes >> rb) > 1024) rb++;
    rs = rb - (width == 4 ? 2 : width == 2 ? 1 : 0);
    while (((n_src + ((uint64_t)1 << rs) - 1) >> rs) > BKT_MAX - 1) rs++;
  }
  const int rd = bkt_region_bits(p, n_dst, width) + BKT_RD_EXTRA;
  // F orders a tile by destination line: 128-byte lines, widened until a region's lines fit the BKT_MAX keys
  int line_shift = width == 4 ? 5 : width == 2 ? 6 : 7;
  while ((1 << (rd - line_shift)) > BKT_MAX) line_shift++;
  int src_line_shift = width == 4 ? 5 : width == 2 ? 6 : 7;
  while ((1 << (rs - src_line_shift)) > BKT_MAX) src_line_shift++;
  const uint32_t bs = (uint32_t)((n_src + ((uint64_t)1 << rs) - 1) >> rs), bd = (uint32_t)((n_dst + ((uint64_t)1 << rd) - 1) >> rd);
  agpu_device* dev = p->dev;
  void *ctl_v = nullptr, *p1 = nullptr, *p2 = nullptr, *cnt_v = nullptr, *off_v = nullptr, *csum_v = nullptr, *ctlb_v = nullptr, *ctlc_v = nullptr;
  const uint32_t ntiles = (uint32_t)((n + BKT_TILE - 1) / BKT_TILE);  // F's tiles, and P's in the destination-only pipeline (16 Ki rows)
  const uint32_t nbp = (bs + 1 + 3) & ~3u;   // padded row stride of the (tile × source region) matrix
  const uint32_t nbpB = (bd + 1 + 3) & ~3u;  // … of the destination-only pipeline's (tile × destination region) matrix
  const uint32_t nbpm = nbp > nbpB ? nbp : nbpB;
  const uint32_t nchunks = (ntiles + BKT_CHUNK - 1) / BKT_CHUNK;
  // Range starts of the partition pass: from a column scan of per-tile counts (round 3).  With XCD-CONTIGUOUS tiles the deterministic
  // layout puts the runs of tiles t and t + 1 side by side in every region's range, the two halves of a line meet in one L2, and the pass
  // gains what reserving the ranges with global atomics could never give: put 5.42 → 5.07 ms at 2^28 rows.  G's range starts ARE reserved
  // with atomics — two ranges per 64-bit fetch-add (round 4: 5.05 → 4.77 ms) — because a count pass over P's output costs more than it
  // saves (0.52 + 0.11 ms against G 2.10 → 1.77: docs/experiments.md §4).
  agpu_status st = agpu_malloc(dev, sizeof(BktCtl), 0, &ctl_v);
  if (st == AGPU_OK) st = agpu_malloc(dev, 8 * n + 16, 0, &p1);
  if (st == AGPU_OK) st = agpu_malloc(dev, 8 * n + 16, 0, &p2);
  if (st == AGPU_OK) st = agpu_malloc(dev, (size_t)ntiles * nbpm * 2, 0, &cnt_v);
  if (st == AGPU_OK) st = agpu_malloc(dev, (size_t)ntiles * nbpm * 4, 0, &off_v);
  if (st == AGPU_OK) st = agpu_malloc(dev, (size_t)nchunks * nbpm * 4, 0, &csum_v);
  // P, H and G over 32 Ki-row tiles through ONE 4-byte LDS array (bkt_partition2_kernel / bkt_gather2_kernel: put of 2^28 random rows
  // 4.78 → 4.33 ms, one process, alternating)
  const uint32_t ntiles2 = (uint32_t)((n + BK2_TILE - 1) / BK2_TILE), nblk2 = (ntiles2 + 7) / 8 * 8;
  // put under the auto policy, three ways (idx_locality_kernel): both columns local → the direct scatter; SOURCE local only (the scatter
  // of a contiguous or sorted selection) → the destination-only pipeline below: the values are fetched by a near-streaming gather inside
  // the partition pass, pairs {destination, value} are partitioned by destination region once and stored by F — no source-side
  // partition, no G; otherwise the full pipeline.  All three are enqueued over the same temporaries, two return at once.
  // (from 2^26 rows: every variant costs a handful of empty launches when it stands down — ≈ 0.1 ms for both, too much for a 0.4 ms put)
  const bool lr = adaptive && (n >= ((uint64_t)1 << 26) || p->tune.gather_bucket == 4);  // (4: tests, any size)
  if (lr && st == AGPU_OK) st = agpu_malloc(dev, sizeof(BktCtl), 0, &ctlb_v);
  // … and DESTINATION local only (a gather into a contiguous or sorted selection): the take's merge-back pipeline with its merge pass
  // storing through the destination column — no pairs at all
  const bool rl = lr && n >= (uint64_t)BKT_T * 32;  // one tile of the merge-back pipeline (TK2_TILE, defined below)
  if (rl && st == AGPU_OK) st = agpu_malloc(dev, sizeof(BktCtl), 0, &ctlc_v);
  if (st != AGPU_OK) st = AGPU_ERR_UNSUPPORTED;  // no room for the 16 B/row of temporaries: the direct kernel needs none
  if (st == AGPU_OK) {
    BktCtl* ctl = static_cast<BktCtl*>(ctl_v);
    const uint32_t nblk = (ntiles + 7) / 8 * 8;
    hipError_t e = hipMemsetAsync(ctl, 0, sizeof(BktCtl), p->stream);
    if (e == hipSuccess && lr) e = hipMemsetAsync(ctlb_v, 0, sizeof(BktCtl), p->stream);
    if (e == hipSuccess && rl) e = hipMemsetAsync(ctlc_v, 0, sizeof(BktCtl), p->stream);
    if (e != hipSuccess) {
      agpu_set_error("hipMemsetAsync failed: %s", hipGetErrorString(e));
      st = AGPU_ERR_HIP;
    } else {
      const uint32_t nchunks_p = (ntiles2 + BKT_CHUNK - 1) / BKT_CHUNK;
      uint64_t hg = (uint64_t)dev->num_cus * 2;
      if (hg > ntiles2) hg = ntiles2;
      const dim3 cgrid_p((nbp + 255) / 256, nchunks_p);
      uint16_t* counts = static_cast<uint16_t*>(cnt_v);
      uint32_t* offsets = static_cast<uint32_t*>(off_v);
      uint32_t* csum = static_cast<uint32_t*>(csum_v);
      // adaptive (put under the auto policy): BOTH index columns local ⇒ the direct scatter launched behind the pipeline does the work
      // (tools/probe/put_distributions.py: sorted → sorted 2.1 ms bucketed, 1.2 direct; sequential → sequential 1.8 vs 0.23; with either
      // side random the pipeline wins)
      const BktCtl* gate = adaptive ? ctl : nullptr;
      if (adaptive) {
        const int sh = width == 4 ? 5 : width == 2 ? 6 : 7;
        hipLaunchKernelGGL(idx_locality_kernel, dim3(2 * LOC_BLOCKS), dim3(256), 0, p->stream, si, di, n, sh, sh, ctl, static_cast<BktCtl*>(ctlb_v), static_cast<BktCtl*>(ctlc_v));
      }
      hipLaunchKernelGGL(bkt_hist_kernel, dim3((uint32_t)hg), dim3(BKT_T), 0, p->stream, si, di, n, n_src, n_dst, rs, rd, bs, bd, ctl, p->flags, counts, nbp, ntiles2,
                         BK2_TILE / (4 * BKT_T));
      hipLaunchKernelGGL(bkt_colsum_kernel, cgrid_p, dim3(256), 0, p->stream, counts, nbp, ntiles2, csum, gate);
      hipLaunchKernelGGL(bkt_colscan_kernel, dim3((nbp + 255) / 256), dim3(256), 0, p->stream, csum, nbp, nchunks_p, ctl->hist_s, gate);
      // G's cursors (the one pass that reserves with atomics): cursors of adjacent regions share an 8-byte word, the words one per 128-byte
      // line while the regions are few (the atomics spread over the L2 channels), 64 bytes apart beyond 1024 regions (round 4,
      // tools/probe/cur_stride_probe.py: reservations to one line queue up behind each other)
      const uint32_t stride_s = bs + 1 <= 1024 ? BKT_CUR_STRIDE : 1;
      const uint32_t stride_d = (bd <= 1024 ? BKT_CUR_STRIDE : 16) | BKT_CUR_PAIRED;
      hipLaunchKernelGGL(bkt_scan_kernel, dim3(1), dim3(BKT_T), 0, p->stream, ctl, bs, bd, stride_s, stride_d);
      hipLaunchKernelGGL(bkt_offsets_kernel, cgrid_p, dim3(256), 0, p->stream, counts, csum, nbp, ntiles2, ctl->base_s, offsets, gate);
      hipLaunchKernelGGL(bkt_partition2_kernel, dim3(nblk2), dim3(BK2_T), 0, p->stream, si, di, n, n_src, n_dst, rs, ctl, static_cast<u32x2*>(p1), offsets, nbp, ntiles2);
#define BKT_GF(W, E)                                                                                                         \
  case W:                                                                                                                    \
    hipLaunchKernelGGL((bkt_gather2_kernel<W>), dim3(nblk2), dim3(BK2_T), 0, p->stream, static_cast<const E*>(src), n_src,   \
                       static_cast<const u32x2*>(p1), rd, bd, ntiles2, src_line_shift, ctl, static_cast<u32x2*>(p2), stride_d); \
    hipLaunchKernelGGL((bkt_store_kernel<W>), dim3(nblk), dim3(BKT_T), 0, p->stream, static_cast<const u32x2*>(p2), ntiles,  \
                       ctl, line_shift, static_cast<E*>(dst));                                                               \
    break;
      switch (width) {
        BKT_GF(4, uint32_t)
        BKT_GF(2, uint16_t)
        BKT_GF(1, uint8_t)
        default: st = AGPU_ERR_UNSUPPORTED; break;
      }
#undef BKT_GF
      if (st == AGPU_OK && lr) {  // the destination-only pipeline: H, P and F with the two index columns in each other's roles
        BktCtl* cb = static_cast<BktCtl*>(ctlb_v);
        const dim3 cgridB((nbpB + 255) / 256, nchunks);
        uint64_t hgB = (uint64_t)dev->num_cus * 2;
        if (hgB > ntiles) hgB = ntiles;
        const uint32_t stride_b = bd + 1 <= 1024 ? BKT_CUR_STRIDE : 1, stride_b2 = bs <= 1024 ? BKT_CUR_STRIDE : 1;
        hipLaunchKernelGGL(bkt_hist_kernel, dim3((uint32_t)hgB), dim3(BKT_T), 0, p->stream, di, si, n, n_dst, n_src, rd, rs, bd, bs, cb, p->flags, counts, nbpB, ntiles, BKT_E / 4);
        hipLaunchKernelGGL(bkt_colsum_kernel, cgridB, dim3(256), 0, p->stream, counts, nbpB, ntiles, csum, cb);
        hipLaunchKernelGGL(bkt_colscan_kernel, dim3((nbpB + 255) / 256), dim3(256), 0, p->stream, csum, nbpB, nchunks, cb->hist_s, cb);
        hipLaunchKernelGGL(bkt_scan_kernel, dim3(1), dim3(BKT_T), 0, p->stream, cb, bd, bs, stride_b, stride_b2);
        hipLaunchKernelGGL(bkt_offsets_kernel, cgridB, dim3(256), 0, p->stream, counts, csum, nbpB, ntiles, cb->base_s, offsets, cb);
        hipLaunchKernelGGL(bkt_partition_kernel, dim3(nblk), dim3(BKT_T), 0, p->stream, di, si, n, n_dst, n_src, rd, bd, cb, static_cast<u32x2*>(p1), offsets, nbpB,
                           ntiles, src, width);
        switch (width) {
          case 4: hipLaunchKernelGGL((bkt_store_kernel<4>), dim3(nblk), dim3(BKT_T), 0, p->stream, static_cast<const u32x2*>(p1), ntiles, cb, line_shift, static_cast<uint32_t*>(dst)); break;
          case 2: hipLaunchKernelGGL((bkt_store_kernel<2>), dim3(nblk), dim3(BKT_T), 0, p->stream, static_cast<const u32x2*>(p1), ntiles, cb, line_shift, static_cast<uint16_t*>(dst)); break;
          default: hipLaunchKernelGGL((bkt_store_kernel<1>), dim3(nblk), dim3(BKT_T), 0, p->stream, static_cast<const u32x2*>(p1), ntiles, cb, line_shift, static_cast<uint8_t*>(dst)); break;
        }
      }
      if (st == AGPU_OK && rl) {
        const agpu_status rs_ = launch_put_through_take(p, width, src, n_src, si, dst, n, di, n_dst, static_cast<const BktCtl*>(ctlc_v));
        if (rs_ != AGPU_OK && rs_ != AGPU_ERR_UNSUPPORTED) st = rs_;
        else if (rs_ == AGPU_ERR_UNSUPPORTED) {  // this variant cannot run: hand its case back to the full pipeline… which has already been
                                                 // told to stand down by the probe — so let the direct kernel take it instead
          hipLaunchKernelGGL(bkt_flag_or_kernel, dim3(1), dim3(1), 0, p->stream, static_cast<const BktCtl*>(ctlc_v), ctl);
        }
      }
      if (st == AGPU_OK && adaptive) (void)launch_put_direct(p, width, src, n_src, si, dst, n_dst, di, n, &ctl->run_direct);
      if (st == AGPU_OK && hipGetLastError() != hipSuccess) {
        agpu_set_error("bucketed put launch failed");
        st = AGPU_ERR_HIP;
      }
    }
  }
  // the pool hands these blocks out again only after the stream has passed the kernels above (runtime.hip markers)
  for (void* q : {csum_v, off_v, cnt_v, p2, p1, ctl_v, ctlb_v, ctlc_v})
    if (q) (void)agpu_free(dev, q);
  return st;
}

// The destination-only pipeline of a put by itself (the host already knows the source column is local — probe_decide): the same
// kernels as inside launch_bucketed, nothing gated, only its own temporaries.
static agpu_status launch_put_dst_only(agpu_pipeline* p, int width, const void* src, uint64_t n_src, const uint32_t* si, void* dst, uint64_t n_dst,
                                       const uint32_t* di, uint64_t n) {
  if (n >= 0xFFFF0000ull || n_src > 0xFFFFFFFFull || n_dst > 0xFFFFFFFFull || !aligned16(si) || !aligned16(di) || p->capturing) return AGPU_ERR_UNSUPPORTED;
  const int rs = bkt_region_bits(p, n_src, width), rd = bkt_region_bits(p, n_dst, width);
  int line_shift = width == 4 ? 5 : width == 2 ? 6 : 7;
  while ((1 << (rd - line_shift)) > BKT_MAX) line_shift++;
  const uint32_t bs = (uint32_t)((n_src + ((uint64_t)1 << rs) - 1) >> rs), bd = (uint32_t)((n_dst + ((uint64_t)1 << rd) - 1) >> rd);
  agpu_device* dev = p->dev;
  const uint32_t ntiles = (uint32_t)((n + BKT_TILE - 1) / BKT_TILE), nbpB = (bd + 1 + 3) & ~3u, nchunks = (ntiles + BKT_CHUNK - 1) / BKT_CHUNK;
  void *ctl_v = nullptr, *p1 = nullptr, *cnt_v = nullptr, *off_v = nullptr, *csum_v = nullptr;
  agpu_status st = agpu_malloc(dev, sizeof(BktCtl), 0, &ctl_v);
  if (st == AGPU_OK) st = agpu_malloc(dev, 8 * n + 16, 0, &p1);
  if (st == AGPU_OK) st = agpu_malloc(dev, (size_t)ntiles * nbpB * 2, 0, &cnt_v);
  if (st == AGPU_OK) st = agpu_malloc(dev, (size_t)ntiles * nbpB * 4, 0, &off_v);
  if (st == AGPU_OK) st = agpu_malloc(dev, (size_t)nchunks * nbpB * 4, 0, &csum_v);
  if (st != AGPU_OK) st = AGPU_ERR_UNSUPPORTED;
  if (st == AGPU_OK) {
    BktCtl* cb = static_cast<BktCtl*>(ctl_v);
    uint16_t* counts = static_cast<uint16_t*>(cnt_v);
    uint32_t* offsets = static_cast<uint32_t*>(off_v);
    uint32_t* csum = static_cast<uint32_t*>(csum_v);
    if (hipMemsetAsync(cb, 0, sizeof(BktCtl), p->stream) != hipSuccess) {
      agpu_set_error("hipMemsetAsync failed");
      st = AGPU_ERR_HIP;
    } else {
      const uint32_t nblk = (ntiles + 7) / 8 * 8;
      uint64_t hg = (uint64_t)dev->num_cus * 2;
      if (hg > ntiles) hg = ntiles;
      const dim3 cgridB((nbpB + 255) / 256, nchunks);
      const uint32_t stride_b = bd + 1 <= 1024 ? BKT_CUR_STRIDE : 1, stride_b2 = bs <= 1024 ? BKT_CUR_STRIDE : 1;
      hipLaunchKernelGGL(bkt_hist_kernel, dim3((uint32_t)hg), dim3(BKT_T), 0, p->stream, di, si, n, n_dst, n_src, rd, rs, bd, bs, cb, p->flags, counts, nbpB, ntiles);
      hipLaunchKernelGGL(bkt_colsum_kernel, cgridB, dim3(256), 0, p->stream, counts, nbpB, ntiles, csum, static_cast<const BktCtl*>(nullptr));
      hipLaunchKernelGGL(bkt_colscan_kernel, dim3((nbpB + 255) / 256), dim3(256), 0, p->stream, csum, nbpB, nchunks, cb->hist_s, static_cast<const BktCtl*>(nullptr));
      hipLaunchKernelGGL(bkt_scan_kernel, dim3(1), dim3(BKT_T), 0, p->stream, cb, bd, bs, stride_b, stride_b2);
      hipLaunchKernelGGL(bkt_offsets_kernel, cgridB, dim3(256), 0, p->stream, counts, csum, nbpB, ntiles, cb->base_s, offsets, static_cast<const BktCtl*>(nullptr));
      hipLaunchKernelGGL(bkt_partition_kernel, dim3(nblk), dim3(BKT_T), 0, p->stream, di, si, n, n_dst, n_src, rd, bd, cb, static_cast<u32x2*>(p1), offsets, nbpB,
                         ntiles, src, width);
      switch (width) {
        case 4: hipLaunchKernelGGL((bkt_store_kernel<4>), dim3(nblk), dim3(BKT_T), 0, p->stream, static_cast<const u32x2*>(p1), ntiles, cb, line_shift, static_cast<uint32_t*>(dst)); break;
        case 2: hipLaunchKernelGGL((bkt_store_kernel<2>), dim3(nblk), dim3(BKT_T), 0, p->stream, static_cast<const u32x2*>(p1), ntiles, cb, line_shift, static_cast<uint16_t*>(dst)); break;
        case 1: hipLaunchKernelGGL((bkt_store_kernel<1>), dim3(nblk), dim3(BKT_T), 0, p->stream, static_cast<const u32x2*>(p1), ntiles, cb, line_shift, static_cast<uint8_t*>(dst)); break;
        default: st = AGPU_ERR_UNSUPPORTED; break;
      }
      if (st == AGPU_OK && hipGetLastError() != hipSuccess) {
        agpu_set_error("destination-only put launch failed");
        st = AGPU_ERR_HIP;
      }
    }
  }
  for (void* q : {csum_v, off_v, cnt_v, p1, ctl_v})
    if (q) (void)agpu_free(dev, q);
  return st;
}

// ---------------------------------------------------------------- take, round 3: the "merge-back" pipeline (4-byte values)
// The bucketed take above moves PAIRS {source index, destination} through two partitions (by source region, then by
// destination region): 54 B/row of HBM traffic, three tile sorts, 64-byte runs of 8-byte pairs.  But a take's destination is
// the row NUMBER: nothing has to travel with the index if the way back is remembered instead —
//   H2  counts[t][b] (u16) = rows of tile t (32 Ki rows) that fall into source region b (512 KiB)            4 B/row read
//       column scan of the matrix (bkt_colsum / colscan / offsets, shared with the deterministic partition) → slot of
//       every (tile, region) run: offs[t][b].  No reservation atomics anywhere.
//   P2  the tile's source indices, counting-sorted by region in LDS, leave as runs of 4-byte entries        4 r + 4 w (runs)
//       srcs[offs[t][b] + rank]; every row's rank inside its run goes to rank16[i] (natural order)         + 2 w
//   G2  16 Ki-slot tiles of srcs, in region order (each XCD a contiguous eighth: its L2 holds the regions in flight):
//       ordered by source line in LDS, gathered, put BACK into slot order in LDS, stored as vals[slot]     4 r + ~5 gather + 4 w
//   F2  tile t again: key from idx[i], slot = offs[t][key] + rank16[i]; the tile's runs are read from vals as contiguous
//       pieces into LDS and every row picks its value: out[i] in natural order, one coalesced store        4 + 2 + 4 (runs) r, 4 w
// ≈ 41 B/row, three LDS passes instead of five, entries half the size so a 32 Ki-row tile fits (the runs keep their
// 64 bytes), and nothing is nondeterministic but the ranks, which are recorded.  Out-of-range indices: their own bucket,
// value 0, sticky flag (as above).  1- / 2-byte values: the same kernels (template parameter W).  put (whose destinations are data)
// keeps the pair pipeline.
static agpu_status launch_take_direct(agpu_pipeline* p, int width, const void* values, uint64_t n_values, const uint32_t* idx, void* out,
                                      uint64_t n_idx, const uint32_t* only_if);
static agpu_status launch_take_bits_direct(agpu_pipeline* p, const void* bits, uint64_t n_bits, const uint32_t* idx, void* out_bits, uint64_t n_idx,
                                           const uint32_t* only_if);
// E of the Boolean put (see "Boolean put" below): entries from natural-order bits
__global__ __launch_bounds__(256) void pb_entries_kernel(const uint32_t* si, const uint32_t* di, const uint32_t* tbits, uint64_t n, uint64_t n_src,
                                                        uint64_t n_dst, uint32_t* ent, const uint32_t* only_if) {
  if (only_if && !*only_if) return;
  const uint64_t i0 = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i0 >= n) return;
  const uint32_t tb = tbits[i0 >> 5] >> (i0 & 31);  // the four rows' bits (i0 is a multiple of 4)
  if (i0 + 4 <= n) {
    const u32x4 s = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(si + i0));
    const u32x4 d = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(di + i0));
    u32x4 e;
    e.x = (s.x < n_src && d.x < n_dst) ? (d.x << 1) | (tb & 1u) : 0xFFFFFFFFu;
    e.y = (s.y < n_src && d.y < n_dst) ? (d.y << 1) | ((tb >> 1) & 1u) : 0xFFFFFFFFu;
    e.z = (s.z < n_src && d.z < n_dst) ? (d.z << 1) | ((tb >> 2) & 1u) : 0xFFFFFFFFu;
    e.w = (s.w < n_src && d.w < n_dst) ? (d.w << 1) | ((tb >> 3) & 1u) : 0xFFFFFFFFu;
    __builtin_nontemporal_store(e, reinterpret_cast<u32x4*>(ent + i0));
  } else {
    for (int k = 0; k < 4; k++)
      if (i0 + k < n) ent[i0 + k] = (si[i0 + k] < n_src && di[i0 + k] < n_dst) ? (di[i0 + k] << 1) | ((tb >> k) & 1u) : 0xFFFFFFFFu;
  }
}

#define TK2_E 32
#define TK2_TILE (BKT_T * TK2_E)  // 32 Ki rows: P2 / F2 tiles
#define TK2_GE 16
#define TK2_GTILE (BKT_T * TK2_GE)  // 16 Ki slots: G2 tiles
#define TK2_REL_BITS 18             // a G2 tile takes the sorted path when its sources span < 2^18 elements (two regions)
#define TK2_POS_BITS 14
#define TK2_GKEY_SHIFT 7                              // G2 orders a tile by groups of 128 source elements (four lines)
#define TK2_GKEYS (1 << (TK2_REL_BITS - TK2_GKEY_SHIFT))  // 2048 keys: two counters per thread

// exclusive scan of lcnt[0 .. 4·BKT_T) in place (thread t owns counters 4t .. 4t+3); returns the grand total via *total
__device__ __forceinline__ void tk2_scan4(uint32_t* lcnt, uint32_t* wave_tot, uint32_t* total) {
  const uint32_t lane = threadIdx.x & (AGPU_WAVE - 1), wave = threadIdx.x / AGPU_WAVE;
  uint32_t c[4], sum = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    c[k] = lcnt[threadIdx.x * 4 + k];
    sum += c[k];
  }
  uint32_t incl = sum;
#pragma unroll
  for (int off = 1; off < AGPU_WAVE; off <<= 1) {
    const uint32_t o = (uint32_t)__shfl_up((int)incl, off);
    if (lane >= (uint32_t)off) incl += o;
  }
  if (lane == AGPU_WAVE - 1) wave_tot[wave] = incl;
  __syncthreads();
  uint32_t base = 0;
  for (uint32_t w = 0; w < wave; w++) base += wave_tot[w];
  uint32_t run = base + incl - sum;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    lcnt[threadIdx.x * 4 + k] = run;
    run += c[k];
  }
  if (threadIdx.x == BKT_T - 1) *total = run;
  __syncthreads();
}

// H2: counts[t][b] for 32 Ki-row tiles; sets the sticky flag for out-of-range indices
__global__ __launch_bounds__(BKT_T) void tk2_hist_kernel(const uint32_t* si, uint64_t n, uint64_t n_src, int rs, uint32_t bs,
                                                        uint32_t* flags, uint16_t* counts, uint32_t nbp, uint32_t ntiles, const BktCtl* gate = nullptr) {
  BKT_GATE(gate);
  __shared__ uint32_t ls[BKT_MAX];
  bool bad = false;
  for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    for (uint32_t b = threadIdx.x; b < nbp; b += BKT_T) ls[b] = 0;
    __syncthreads();
    const uint64_t base = (uint64_t)tile * TK2_TILE;
#pragma unroll
    for (int q = 0; q < TK2_E / 4; q++) {
      const uint64_t i0 = base + ((uint64_t)q * BKT_T + threadIdx.x) * 4;
      uint32_t s[4];
      int live = 0;
      if (i0 + 4 <= n) {
        const u32x4 t = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(si + i0));
        s[0] = t.x; s[1] = t.y; s[2] = t.z; s[3] = t.w;
        live = 4;
      } else {
        for (int k = 0; k < 4; k++)
          if (i0 + k < n) {
            s[k] = si[i0 + k];
            live = k + 1;
          }
      }
#pragma unroll
      for (int k = 0; k < 4; k++)
        if (k < live) {
          const bool ok = s[k] < n_src;
          bad |= !ok;
          atomicAdd(&ls[ok ? (s[k] >> rs) : bs], 1u);
        }
    }
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < nbp; b += BKT_T) counts[(uint64_t)tile * nbp + b] = (uint16_t)ls[b];
    __syncthreads();
  }
  if (bad) *reinterpret_cast<volatile uint32_t*>(flags) = AGPU_FLAG_INDEX_RANGE;
}

// P2: one 32 Ki-row tile per workgroup
__global__ __launch_bounds__(BKT_T) void tk2_partition_kernel(const uint32_t* si, uint64_t n, uint64_t n_src, int rs, uint32_t bs,
                                                             const uint32_t* offsets, uint32_t nbp, uint32_t ntiles, uint32_t* srcs,
                                                             uint16_t* rank16, const BktCtl* gate = nullptr) {
  BKT_GATE(gate);
  __shared__ uint32_t sorted[TK2_TILE];
  __shared__ uint32_t lcnt[BKT_MAX];
  __shared__ uint32_t wave_tot[BKT_T / AGPU_WAVE];
  __shared__ uint32_t tile_rows;
  // XCD-contiguous walk: the runs of tiles t and t + 1 are NEIGHBOURS in every region's range (64 bytes each, the slots
  // come from a column scan) — handled by the same XCD a few dispatches apart, the two halves of a 128-byte line meet in
  // that XCD's L2 and leave as one full line (round-robin tiles put them into two different L2s)
  uint64_t tile64;
  if (!bkt_tile_of_block(ntiles, &tile64)) return;
  const uint32_t tile = (uint32_t)tile64;
  const uint64_t base = (uint64_t)tile * TK2_TILE;
  const uint32_t n_src32 = (uint32_t)(n_src > 0xFFFFFFFFull ? 0xFFFFFFFFull : n_src);
  uint32_t s[TK2_E];
#pragma unroll
  for (int q = 0; q < TK2_E / 4; q++) {
    const uint64_t i0 = base + ((uint64_t)q * BKT_T + threadIdx.x) * 4;
    u32x4 t = {0, 0, 0, 0};
    if (i0 + 4 <= n) t = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(si + i0));
    else {
      if (i0 < n) t.x = si[i0];
      if (i0 + 1 < n) t.y = si[i0 + 1];
      if (i0 + 2 < n) t.z = si[i0 + 2];
    }
    s[q * 4] = t.x; s[q * 4 + 1] = t.y; s[q * 4 + 2] = t.z; s[q * 4 + 3] = t.w;
  }
  for (uint32_t k = threadIdx.x; k < BKT_MAX; k += BKT_T) lcnt[k] = 0;
  __syncthreads();
  uint16_t rank[TK2_E];
#pragma unroll
  for (int e = 0; e < TK2_E; e++) {
    const uint64_t i = base + ((uint64_t)(e / 4) * BKT_T + threadIdx.x) * 4 + (e & 3);
    rank[e] = 0;
    if (i < n) rank[e] = (uint16_t)atomicAdd(&lcnt[s[e] < n_src32 ? (s[e] >> rs) : bs], 1u);
  }
  // the ranks leave at once (natural order, 8 bytes per lane and quad): F2 finds every row's slot with them
  // (rank16 == nullptr: the Boolean put below never merges back)
#pragma unroll
  for (int q = 0; q < TK2_E / 4; q++) {
    const uint64_t i0 = base + ((uint64_t)q * BKT_T + threadIdx.x) * 4;
    if (!rank16) break;
    if (i0 + 4 <= n) {
      const u32x2 pk = {(uint32_t)rank[q * 4] | ((uint32_t)rank[q * 4 + 1] << 16), (uint32_t)rank[q * 4 + 2] | ((uint32_t)rank[q * 4 + 3] << 16)};
      __builtin_nontemporal_store(pk, reinterpret_cast<u32x2*>(rank16 + i0));
    } else {
      for (int k = 0; k < 4; k++)
        if (i0 + k < n) rank16[i0 + k] = rank[q * 4 + k];
    }
  }
  __syncthreads();
  tk2_scan4(lcnt, wave_tot, &tile_rows);  // lcnt[k] = exclusive start of key k inside the tile
#pragma unroll
  for (int e = 0; e < TK2_E; e++) {
    const uint64_t i = base + ((uint64_t)(e / 4) * BKT_T + threadIdx.x) * 4 + (e & 3);
    if (i < n) sorted[lcnt[s[e] < n_src32 ? (s[e] >> rs) : bs] + rank[e]] = s[e];
  }
  __syncthreads();
  // lcnt[k] := global slot of the run's first entry − its start inside the tile, so that slot(j) = lcnt[key(j)] + j
  {
    uint32_t st[4];
#pragma unroll
    for (int k = 0; k < 4; k++) st[k] = lcnt[threadIdx.x * 4 + k];
    const uint32_t* orow = offsets + (uint64_t)tile * nbp;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const uint32_t kk = threadIdx.x * 4 + k;
      if (kk <= bs) lcnt[kk] = orow[kk] - st[k];
    }
  }
  __syncthreads();
  const uint32_t rows_here = tile_rows;
  for (uint32_t j = threadIdx.x; j < rows_here; j += BKT_T) {
    const uint32_t v = sorted[j];
    srcs[(uint32_t)(lcnt[v < n_src32 ? (v >> rs) : bs] + j)] = v;
  }
}

// G2: vals[slot] = values[srcs[slot]] (0 when out of range), 16 Ki slots per workgroup, XCD-contiguous walk
// BITS: the source column's VALIDITY bit travels with the value (take of an array with nulls [ref: crates/routines/src/take.rs
// :9-55 values + bool.rs:33-46 validity]): the bit of row idx sits in the 16 KiB of bitmap that belong to the tile's 512 KiB
// region — after the line-group sort neighbouring lanes read the same 16 bytes of it — and leaves as vbits_slot, one bit per
// slot in slot order (256 u64 words per tile).
// W = the value width in bytes (4, 2, 1): regions, entries and the LDS arrays are in ELEMENTS either way; only the gather, the
// slow path and the last store see the type.
template <int WPE, bool BITS, int W = 4, int T = BKT_T, int GE = TK2_GE>
__global__ __launch_bounds__(T, WPE) void tk2_gather_kernel(const typename ElemOf<W>::type* values, uint64_t n_src, const uint32_t* srcs,
                                                               uint64_t total, uint32_t ntiles, typename ElemOf<W>::type* vals,
                                                               const uint32_t* vbits_src, uint64_t* vbits_slot, const BktCtl* gate = nullptr) {
  BKT_GATE(gate);
  typedef typename ElemOf<W>::type E;
  // GE = slots per thread: 16 (16 Ki-slot tiles, 18-bit offsets: regions up to 2^17 elements) or 8 (8 Ki slots, 19-bit offsets: the
  // 2^18 … 2^19-element regions a source of more than 2^29 elements needs to stay within 4095 of them)
  constexpr int GT = GE * T, PB = GE == 16 ? 14 : 13, RB = 32 - PB, NK = 1 << (RB - TK2_GKEY_SHIFT);  // slots per tile (T = 1024: 16 Ki; the 512-thread variant: 8 Ki, four workgroups per CU)
  constexpr int KPT = NK / T;  // line-group counters per thread in the scan
  // 64 KiB + 8 KiB of LDS and ≤ 64 VGPRs: TWO workgroups per CU, so that one's loads and gathers run under the other's LDS
  // phases.  To stay inside 64 registers the tile's sources are loaded twice (the second time from L2) instead of being
  // kept across the ranking, ranks are packed two to a register, and only the 16 sorted entries live across the barrier
  // that turns the entry array into the value array (a first version with everything kept spilled 18–32 VGPRs: 4.5 B/row of
  // scratch traffic by PMC).
  __shared__ uint32_t sorted[GT];
  __shared__ uint32_t lcnt[NK];
  __shared__ uint32_t wave_tot[T / AGPU_WAVE];
  __shared__ uint32_t tile_rows;
  __shared__ uint32_t red[2 * (T / AGPU_WAVE)];
  __shared__ uint32_t bitl[BITS ? GT / 32 : 1];  // the tile's validity bits by slot
  uint64_t tile;
  if (!bkt_tile_of_block(ntiles, &tile)) return;
  const uint64_t base = tile * GT;
  if (base >= total) return;
  const uint32_t lane = threadIdx.x & (AGPU_WAVE - 1), wave = threadIdx.x / AGPU_WAVE;
  if constexpr (BITS)
    if (threadIdx.x < GT / 32) bitl[threadIdx.x] = 0;
  auto store_bits = [&]() {  // after a barrier: slot-ordered validity words of this tile
    if constexpr (BITS) {
      if (threadIdx.x < GT / 64 && base + (uint64_t)threadIdx.x * 64 < total)
        vbits_slot[base / 64 + threadIdx.x] = (uint64_t)bitl[threadIdx.x * 2] | ((uint64_t)bitl[threadIdx.x * 2 + 1] << 32);
    }
  };
  auto load4 = [&](int q) -> u32x4 {
    const uint64_t i0 = base + ((uint64_t)q * T + threadIdx.x) * 4;
    u32x4 t = {0, 0, 0, 0};
    if (i0 + 4 <= total) t = *reinterpret_cast<const u32x4*>(srcs + i0);
    else {
      if (i0 < total) t.x = srcs[i0];
      if (i0 + 1 < total) t.y = srcs[i0 + 1];
      if (i0 + 2 < total) t.z = srcs[i0 + 2];
    }
    return t;
  };
  auto live_at = [&](int q, int k) { return base + ((uint64_t)q * T + threadIdx.x) * 4 + (uint64_t)k < total; };
  // pass 1: the tile's source span decides the path (uniform over the block)
  uint32_t mn = 0xFFFFFFFFu, mx = 0;
  for (uint32_t k = threadIdx.x; k < NK; k += T) lcnt[k] = 0;
#pragma unroll
  for (int q = 0; q < GE / 4; q++) {
    const u32x4 t = load4(q);
    const uint32_t sv[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
    for (int k = 0; k < 4; k++)
      if (live_at(q, k)) {
        mn = sv[k] < mn ? sv[k] : mn;
        mx = sv[k] > mx ? sv[k] : mx;
      }
  }
#pragma unroll
  for (int off = AGPU_WAVE / 2; off > 0; off >>= 1) {
    const uint32_t a = (uint32_t)__shfl_down((int)mn, off), b2 = (uint32_t)__shfl_down((int)mx, off);
    mn = a < mn ? a : mn;
    mx = b2 > mx ? b2 : mx;
  }
  if (lane == 0) {
    red[wave] = mn;
    red[T / AGPU_WAVE + wave] = mx;
  }
  __syncthreads();
  mn = red[0];
  mx = red[T / AGPU_WAVE];
  for (int w = 1; w < T / AGPU_WAVE; w++) {
    mn = red[w] < mn ? red[w] : mn;
    mx = red[T / AGPU_WAVE + w] > mx ? red[T / AGPU_WAVE + w] : mx;
  }
  const uint32_t origin = mn & ~((1u << TK2_GKEY_SHIFT) - 1u);
  const bool fast = mx < n_src && (mx - origin) < (1u << RB);
  if (!fast) {  // a tile of out-of-range rows, or one that straddles many small regions: row by row, slots keep their place
#pragma unroll
    for (int q = 0; q < GE / 4; q++) {
      const u32x4 t = load4(q);
      const uint32_t sv[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
      for (int k = 0; k < 4; k++)
        if (live_at(q, k)) {
          const bool ok = sv[k] < n_src;
          vals[base + ((uint64_t)q * T + threadIdx.x) * 4 + (uint64_t)k] = ok ? values[sv[k]] : (E)0;
          if constexpr (BITS) {
            const uint32_t pos = ((uint32_t)q * T + threadIdx.x) * 4 + (uint32_t)k;
            if (ok && ((vbits_src[sv[k] >> 5] >> (sv[k] & 31)) & 1u)) atomicOr(&bitl[pos >> 5], 1u << (pos & 31));
          }
        }
    }
    if constexpr (BITS) {
      __syncthreads();
      store_bits();
    }
    return;
  }
  // pass 2 (sources from L2): rank every row inside its source LINE GROUP (128 elements): neighbouring lanes of the gather
  // will share a request.  Ranks < 2^14: two to a register.
  uint32_t rank2[GE / 2];
#pragma unroll
  for (int q = 0; q < GE / 4; q++) {
    const u32x4 t = load4(q);
    const uint32_t sv[4] = {t.x, t.y, t.z, t.w};
    uint32_t r[4];
#pragma unroll
    for (int k = 0; k < 4; k++) r[k] = live_at(q, k) ? atomicAdd(&lcnt[(sv[k] - origin) >> TK2_GKEY_SHIFT], 1u) : 0u;
    rank2[q * 2] = r[0] | (r[1] << 16);
    rank2[q * 2 + 1] = r[2] | (r[3] << 16);
  }
  __syncthreads();
  {  // exclusive scan of the 2048 counters: thread t owns KPT consecutive ones
    uint32_t c[KPT], sum = 0;
#pragma unroll
    for (int k = 0; k < KPT; k++) {
      c[k] = lcnt[threadIdx.x * KPT + k];
      sum += c[k];
    }
    uint32_t incl = sum;
#pragma unroll
    for (int off = 1; off < AGPU_WAVE; off <<= 1) {
      const uint32_t o = (uint32_t)__shfl_up((int)incl, off);
      if (lane >= (uint32_t)off) incl += o;
    }
    if (lane == AGPU_WAVE - 1) wave_tot[wave] = incl;
    __syncthreads();
    uint32_t pre = 0;
    for (uint32_t w = 0; w < wave; w++) pre += wave_tot[w];
    uint32_t run = pre + incl - sum;
#pragma unroll
    for (int k = 0; k < KPT; k++) {
      lcnt[threadIdx.x * KPT + k] = run;
      run += c[k];
    }
    if (threadIdx.x == T - 1) tile_rows = run;
    __syncthreads();
  }
  // pass 3 (sources from L2 again): entries {source − origin, slot inside the tile} into line-group order
#pragma unroll
  for (int q = 0; q < GE / 4; q++) {
    const u32x4 t = load4(q);
    const uint32_t sv[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
    for (int k = 0; k < 4; k++)
      if (live_at(q, k)) {
        const uint32_t rel = sv[k] - origin;
        const uint32_t pos = ((uint32_t)q * T + threadIdx.x) * 4 + (uint32_t)k;
        const uint32_t rk = (rank2[q * 2 + (k >> 1)] >> ((k & 1) * 16)) & 0xFFFFu;
        sorted[lcnt[rel >> TK2_GKEY_SHIFT] + rk] = (rel << PB) | pos;
      }
  }
  __syncthreads();
  const uint32_t rows_here = tile_rows;
  uint32_t ent[GE];
#pragma unroll
  for (int e = 0; e < GE; e++) {
    const uint32_t j = (uint32_t)e * T + threadIdx.x;
    ent[e] = j < rows_here ? sorted[j] : 0u;
  }
  __syncthreads();  // every entry has been read: the same array takes the values, back in slot order
#ifndef TK2_GGRP
#define TK2_GGRP 8  // gathers in flight per lane (16: 58 VGPRs, measured no faster — tools/probe/put_variants.sh)
#endif
#pragma unroll
  for (int h0 = 0; h0 < GE; h0 += TK2_GGRP) {
    uint32_t v[TK2_GGRP];
#pragma unroll
    for (int e = 0; e < TK2_GGRP; e++) {
      const uint32_t j = (uint32_t)(h0 + e) * T + threadIdx.x;
      v[e] = j < rows_here ? (uint32_t)values[origin + (ent[h0 + e] >> PB)] : 0u;  // the L2-resident gather
    }
#pragma unroll
    for (int e = 0; e < TK2_GGRP; e++) {
      const uint32_t j = (uint32_t)(h0 + e) * T + threadIdx.x;
      if (j < rows_here) sorted[ent[h0 + e] & ((1u << PB) - 1u)] = v[e];
    }
    if constexpr (BITS) {  // the validity bits of the same eight sources: 16 bytes of bitmap per line group, shared by neighbours
      uint32_t w[TK2_GGRP];
#pragma unroll
      for (int e = 0; e < TK2_GGRP; e++) {
        const uint32_t j = (uint32_t)(h0 + e) * T + threadIdx.x;
        w[e] = j < rows_here ? vbits_src[(origin + (ent[h0 + e] >> PB)) >> 5] : 0u;
      }
#pragma unroll
      for (int e = 0; e < TK2_GGRP; e++) {
        const uint32_t j = (uint32_t)(h0 + e) * T + threadIdx.x;
        const uint32_t src = origin + (ent[h0 + e] >> PB), pos = ent[h0 + e] & ((1u << PB) - 1u);
        if (j < rows_here && ((w[e] >> (src & 31)) & 1u)) atomicOr(&bitl[pos >> 5], 1u << (pos & 31));
      }
    }
  }
  __syncthreads();
  store_bits();
#pragma unroll
  for (int q = 0; q < GE / 4; q++) {
    const uint32_t l0 = ((uint32_t)q * T + threadIdx.x) * 4;
    const uint64_t i0 = base + l0;
    if (i0 + 4 <= total) {
      if constexpr (W == 4) {
        const u32x4 v = {sorted[l0], sorted[l0 + 1], sorted[l0 + 2], sorted[l0 + 3]};
        *reinterpret_cast<u32x4*>(vals + i0) = v;
      } else if constexpr (W == 2) {
        const u32x2 v = {sorted[l0] | (sorted[l0 + 1] << 16), sorted[l0 + 2] | (sorted[l0 + 3] << 16)};
        *reinterpret_cast<u32x2*>(vals + i0) = v;
      } else {
        *reinterpret_cast<uint32_t*>(vals + i0) = sorted[l0] | (sorted[l0 + 1] << 8) | (sorted[l0 + 2] << 16) | (sorted[l0 + 3] << 24);
      }
    } else {
      for (int k = 0; k < 4; k++)
        if (i0 + k < total) vals[i0 + k] = (E)sorted[l0 + k];
    }
  }
}

// F2: out[i] = vals[offs[t][key(i)] + rank16[i]] — the tile's runs come into LDS as contiguous pieces, rows pick from there
// MODE 0: values; 1: values + the source validity bits (agpu_take_validity); 2: bits only (Boolean take: no value array at all);
// 3: bits only, leaving as the Boolean put's entries ent[i] = dst_idx[i] * 2 + bit (0xFFFFFFFF for a row with either index out of range);
// 4: a PUT whose destination column is local: out = the destination array, row i's value goes to out[dst_idx[i]] (rows with either index out
//    of range are dropped and raise the sticky flag) — the source side is this pipeline's random gather, the destination side needs none
template <int MODE, int W = 4>
__global__ __launch_bounds__(BKT_T) void tk2_merge_kernel(const uint32_t* si, uint64_t n, uint64_t n_src, int rs, uint32_t bs,
                                                         const uint16_t* counts, const uint32_t* offsets, uint32_t nbp,
                                                         uint32_t ntiles, const uint16_t* rank16, const typename ElemOf<W>::type* vals,
                                                         typename ElemOf<W>::type* out, const uint32_t* vbits_slot, uint64_t* out_validity,
                                                         const uint32_t* di = nullptr, uint64_t n_dst = 0, const BktCtl* gate = nullptr,
                                                         uint32_t* flags = nullptr) {
  BKT_GATE(gate);
  typedef typename ElemOf<W>::type E;
  static_assert(MODE != 3 || W == 4, "the entry array is 4 bytes wide");
  __shared__ uint32_t A[TK2_TILE];
  constexpr bool PUT = MODE == 4, BITS = MODE >= 1 && MODE <= 3, VALUES = MODE <= 1 || PUT, ENT = MODE == 3;
  __shared__ uint32_t bl[BITS ? TK2_TILE / 32 : 1];  // the validity bits of the tile's slots (BITS)
  __shared__ uint32_t lcnt[BKT_MAX];
  __shared__ uint32_t wave_tot[BKT_T / AGPU_WAVE];
  __shared__ uint32_t tile_rows;
  uint64_t tile64;  // XCD-contiguous, like P2: the 64-byte runs of neighbouring tiles share their lines in one L2
  if (!bkt_tile_of_block(ntiles, &tile64)) return;
  const uint32_t tile = (uint32_t)tile64;
  // (bl needs no zeroing: every word of it is written whole by a ballot below)
  const uint64_t base = (uint64_t)tile * TK2_TILE;
  const uint32_t n_src32 = (uint32_t)(n_src > 0xFFFFFFFFull ? 0xFFFFFFFFull : n_src);
  // start[k] (exclusive scan of the tile's counts) and the runs' global slots
  const uint16_t* crow = counts + (uint64_t)tile * nbp;
  const uint32_t* orow = offsets + (uint64_t)tile * nbp;
  uint32_t goff[4];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const uint32_t kk = threadIdx.x * 4 + k;
    lcnt[kk] = kk <= bs ? (uint32_t)crow[kk] : 0u;
    goff[k] = kk <= bs ? orow[kk] : 0u;
  }
  uint32_t s[TK2_E];
  uint16_t rank[TK2_E];
#pragma unroll
  for (int q = 0; q < TK2_E / 4; q++) {
    const uint64_t i0 = base + ((uint64_t)q * BKT_T + threadIdx.x) * 4;
    u32x4 t = {0, 0, 0, 0};
    u32x2 r = {0, 0};
    if (i0 + 4 <= n) {
      t = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(si + i0));
      r = __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(rank16 + i0));
    } else {
      uint32_t tt[4] = {0, 0, 0, 0}, rr[4] = {0, 0, 0, 0};
      for (int k = 0; k < 4; k++)
        if (i0 + k < n) {
          tt[k] = si[i0 + k];
          rr[k] = rank16[i0 + k];
        }
      t = u32x4{tt[0], tt[1], tt[2], tt[3]};
      r = u32x2{rr[0] | (rr[1] << 16), rr[2] | (rr[3] << 16)};
    }
    s[q * 4] = t.x; s[q * 4 + 1] = t.y; s[q * 4 + 2] = t.z; s[q * 4 + 3] = t.w;
    rank[q * 4] = (uint16_t)r.x; rank[q * 4 + 1] = (uint16_t)(r.x >> 16); rank[q * 4 + 2] = (uint16_t)r.y; rank[q * 4 + 3] = (uint16_t)(r.y >> 16);
  }
  __syncthreads();
  tk2_scan4(lcnt, wave_tot, &tile_rows);  // lcnt[k] = start of run k inside the tile
  // every row names its global slot at its local slot; then the slots are filled with the values, coalesced per run
  // (LDS budget: A 128 KiB + 16 KiB + 8 KiB — the starts fit 16 bits, their 32-bit array is reused for the deltas)
  __shared__ uint16_t start16[BKT_MAX];
  uint32_t st[4];
#pragma unroll
  for (int k = 0; k < 4; k++) st[k] = lcnt[threadIdx.x * 4 + k];
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; k++) {
    start16[threadIdx.x * 4 + k] = (uint16_t)st[k];
    lcnt[threadIdx.x * 4 + k] = goff[k] - st[k];  // global slot of run k − its local start
  }
  __syncthreads();
  const uint32_t rows_here = tile_rows;
  uint32_t sl[TK2_E];
#pragma unroll
  for (int e = 0; e < TK2_E; e++) {
    const uint64_t i = base + ((uint64_t)(e / 4) * BKT_T + threadIdx.x) * 4 + (e & 3);
    sl[e] = 0;
    if (i < n) {
      const uint32_t key = s[e] < n_src32 ? (s[e] >> rs) : bs;
      sl[e] = (uint32_t)start16[key] + rank[e];
      A[sl[e]] = lcnt[key] + sl[e];
      if constexpr (ENT || PUT) sl[e] |= s[e] < n_src32 ? 0u : 0x80000000u;  // the row's source index is out of range: it is dropped
    }
  }
  __syncthreads();
#ifndef TK2_F2_CHUNK_ALL
#define TK2_F2_CHUNK_ALL 0  // 1: the plain take's merge in chunks of eight as well (tools/probe A/B)
#endif
  if constexpr (VALUES && (BITS || TK2_F2_CHUNK_ALL)) {
    // values AND bits: eight slots at a time — with all 32 slot numbers, bit words and values in flight at once the kernel
    // needed 137 registers (9 spilled: 1.38 ms against 0.92 for either half alone)
#pragma unroll
    for (int c = 0; c < TK2_E; c += 8) {
      uint32_t g8[8], w8[8], v8[8];
#pragma unroll
      for (int e = 0; e < 8; e++) {
        const uint32_t j = (uint32_t)(c + e) * BKT_T + threadIdx.x;
        g8[e] = j < rows_here ? A[j] : 0u;
      }
#pragma unroll
      for (int e = 0; e < 8; e++) {
        const uint32_t j = (uint32_t)(c + e) * BKT_T + threadIdx.x;
        if constexpr (BITS) w8[e] = j < rows_here ? vbits_slot[g8[e] >> 5] : 0u;
        v8[e] = j < rows_here ? (uint32_t)__builtin_nontemporal_load(vals + g8[e]) : 0u;
      }
#pragma unroll
      for (int e = 0; e < 8; e++) {
        const uint32_t j = (uint32_t)(c + e) * BKT_T + threadIdx.x;
        if constexpr (BITS) {
          const uint64_t m = __ballot((w8[e] >> (g8[e] & 31)) & 1u);
          const uint32_t lane = threadIdx.x & (AGPU_WAVE - 1);
          if (lane == 0) bl[j >> 5] = (uint32_t)m;
          if (lane == 32) bl[j >> 5] = (uint32_t)(m >> 32);
        }
        if (j < rows_here) A[j] = v8[e];
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  } else {  // all 32 slot numbers first, then 32 loads in flight, then the values back into the slots
    uint32_t g[TK2_E];
#pragma unroll
    for (int e = 0; e < TK2_E; e++) {
      const uint32_t j = (uint32_t)e * BKT_T + threadIdx.x;
      g[e] = j < rows_here ? A[j] : 0u;
    }
    if constexpr (BITS) {  // the slot's validity bit: one word per 32 slots, a run's 16 slots share it.  A wave's lanes hold 64
                           // CONSECUTIVE local slots: its ballot is two whole words of bl — no LDS atomics (32 lanes ORing
                           // into one word serialise)
#pragma unroll
      for (int e = 0; e < TK2_E; e++) {
        const uint32_t j = (uint32_t)e * BKT_T + threadIdx.x;
        const bool bit = j < rows_here && ((vbits_slot[g[e] >> 5] >> (g[e] & 31)) & 1u);
        const uint64_t m = __ballot(bit);
        const uint32_t lane = threadIdx.x & (AGPU_WAVE - 1);
        if (lane == 0) bl[j >> 5] = (uint32_t)m;
        if (lane == 32) bl[j >> 5] = (uint32_t)(m >> 32);
      }
    }
    if constexpr (VALUES) {
#pragma unroll
      for (int e = 0; e < TK2_E; e++) {
        const uint32_t j = (uint32_t)e * BKT_T + threadIdx.x;
        if (j < rows_here) g[e] = (uint32_t)__builtin_nontemporal_load(vals + g[e]);
      }
#pragma unroll
      for (int e = 0; e < TK2_E; e++) {
        const uint32_t j = (uint32_t)e * BKT_T + threadIdx.x;
        if (j < rows_here) A[j] = g[e];
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < TK2_E / 4; q++) {
    const uint64_t i0 = base + ((uint64_t)q * BKT_T + threadIdx.x) * 4;
    if constexpr (PUT) {
      u32x4 d = {0, 0, 0, 0};
      if (i0 + 4 <= n) d = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(di + i0));
      else
        for (int k = 0; k < 4; k++)
          if (i0 + k < n) d[k] = di[i0 + k];
      bool bad = false;
#pragma unroll
      for (int k = 0; k < 4; k++)
        if (i0 + k < n) {
          const uint32_t slk = sl[q * 4 + k];
          if (!(slk >> 31) && d[k] < n_dst) out[d[k]] = (E)A[slk];
          else bad = true;
        }
      if (bad) *reinterpret_cast<volatile uint32_t*>(flags) = AGPU_FLAG_INDEX_RANGE;
    } else if constexpr (VALUES) {
      if (i0 + 4 <= n) {
        const uint32_t a0 = A[sl[q * 4]], a1 = A[sl[q * 4 + 1]], a2 = A[sl[q * 4 + 2]], a3 = A[sl[q * 4 + 3]];
        if constexpr (W == 4) __builtin_nontemporal_store(u32x4{a0, a1, a2, a3}, reinterpret_cast<u32x4*>(out + i0));
        else if constexpr (W == 2) __builtin_nontemporal_store(u32x2{a0 | (a1 << 16), a2 | (a3 << 16)}, reinterpret_cast<u32x2*>(out + i0));
        else __builtin_nontemporal_store(a0 | (a1 << 8) | (a2 << 16) | (a3 << 24), reinterpret_cast<uint32_t*>(out + i0));
      } else {
        for (int k = 0; k < 4; k++)
          if (i0 + k < n) out[i0 + k] = (E)A[sl[q * 4 + k]];
      }
    }
    if constexpr (ENT) {  // entries of the Boolean put in natural order (out = the entry array)
      u32x4 d = {0, 0, 0, 0};
      if (i0 + 4 <= n) d = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(di + i0));
      else
        for (int k = 0; k < 4; k++)
          if (i0 + k < n) d[k] = di[i0 + k];
      u32x4 e;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const uint32_t slk = sl[q * 4 + k], l = slk & 0x7FFFFFFFu;
        e[k] = (!(slk >> 31) && d[k] < n_dst) ? (d[k] << 1) | ((bl[l >> 5] >> (l & 31)) & 1u) : 0xFFFFFFFFu;
      }
      if (i0 + 4 <= n) __builtin_nontemporal_store(e, reinterpret_cast<u32x4*>(out + i0));
      else
        for (int k = 0; k < 4; k++)
          if (i0 + k < n) out[i0 + k] = e[k];
    } else if constexpr (BITS) {  // 16 neighbouring lanes hold the 64 rows of one output validity word (rows past n: 0)
      uint64_t nib = 0;
#pragma unroll
      for (int k = 0; k < 4; k++)
        if (i0 + k < n && ((bl[sl[q * 4 + k] >> 5] >> (sl[q * 4 + k] & 31)) & 1u)) nib |= 1ull << k;
      uint64_t word = nib << (4 * (threadIdx.x & 15));
#pragma unroll
      for (int sft = 1; sft < 16; sft <<= 1) word |= __shfl_xor(word, sft);
      if ((threadIdx.x & 15) == 0 && i0 < n) out_validity[i0 / 64] = word;
    }
  }
}

// take of 4-byte values through the merge-back pipeline; AGPU_ERR_UNSUPPORTED when the shape does not qualify
// G2 for BITS as the data (Boolean take: out bit i = bits[idx[i]] [ref: crates/routines/src/bool.rs:15-46, bool/take.wgsl]): the
// "elements" are the bitmap's 32-bit words, regions are 2^rsw words (16 KiB of bitmap by default), the slot entries are
// {word − origin : 13 bits, bit position : 5, slot : 14}, ordered by source LINE (32 words); every gathered bit goes straight to
// the tile's slot-ordered bit array — there is no value array at all — and leaves as vbits_slot like above.
// WIDE (bitmaps over 2^29 bits — a 1e9-row column's — whose regions have to be larger than 2^12 words for ≤ 4095 of them): the entry
// is {word − origin : 18 bits, slot : 14} like G2's, ordered by groups of four lines (2048 keys), and the BIT POSITION, which no longer
// fits, is read back from the tile's (L2-resident) index entries.
template <bool WIDE>
__global__ __launch_bounds__(BKT_T, 8) void tk2_gather_bits_kernel(const uint32_t* bits, uint64_t n_bits, const uint32_t* srcs, uint64_t total,
                                                                  uint32_t ntiles, uint64_t* vbits_slot, const BktCtl* gate = nullptr) {
  BKT_GATE(gate);
  __shared__ uint32_t sorted[TK2_GTILE];
  constexpr int TK2B_REL_BITS = WIDE ? 18 : 13, KS = WIDE ? 7 : 5;  // key = (word − origin) >> KS
  __shared__ uint32_t lcnt[1 << (TK2B_REL_BITS - KS)];  // 256 line keys (2048 groups of four lines when WIDE)
  __shared__ uint32_t wave_tot[BKT_T / AGPU_WAVE];
  __shared__ uint32_t tile_rows;
  __shared__ uint32_t red[2 * (BKT_T / AGPU_WAVE)];
  __shared__ uint32_t bitl[TK2_GTILE / 32];
  constexpr uint32_t NKEYS = 1u << (TK2B_REL_BITS - KS);
  uint64_t tile;
  if (!bkt_tile_of_block(ntiles, &tile)) return;
  const uint64_t base = tile * TK2_GTILE;
  if (base >= total) return;
  const uint32_t lane = threadIdx.x & (AGPU_WAVE - 1), wave = threadIdx.x / AGPU_WAVE;
  if (threadIdx.x < TK2_GTILE / 32) bitl[threadIdx.x] = 0;
  for (uint32_t kk = threadIdx.x; kk < NKEYS; kk += BKT_T) lcnt[kk] = 0;
  auto load4 = [&](int q) -> u32x4 {
    const uint64_t i0 = base + ((uint64_t)q * BKT_T + threadIdx.x) * 4;
    u32x4 t = {0, 0, 0, 0};
    if (i0 + 4 <= total) t = *reinterpret_cast<const u32x4*>(srcs + i0);
    else {
      if (i0 < total) t.x = srcs[i0];
      if (i0 + 1 < total) t.y = srcs[i0 + 1];
      if (i0 + 2 < total) t.z = srcs[i0 + 2];
    }
    return t;
  };
  auto live_at = [&](int q, int k) { return base + ((uint64_t)q * BKT_T + threadIdx.x) * 4 + (uint64_t)k < total; };
  uint32_t mn = 0xFFFFFFFFu, mx = 0;
#pragma unroll
  for (int q = 0; q < TK2_GE / 4; q++) {
    const u32x4 t = load4(q);
    const uint32_t sv[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
    for (int k = 0; k < 4; k++)
      if (live_at(q, k)) {
        mn = sv[k] < mn ? sv[k] : mn;
        mx = sv[k] > mx ? sv[k] : mx;
      }
  }
#pragma unroll
  for (int off = AGPU_WAVE / 2; off > 0; off >>= 1) {
    const uint32_t a = (uint32_t)__shfl_down((int)mn, off), b2 = (uint32_t)__shfl_down((int)mx, off);
    mn = a < mn ? a : mn;
    mx = b2 > mx ? b2 : mx;
  }
  if (lane == 0) {
    red[wave] = mn;
    red[BKT_T / AGPU_WAVE + wave] = mx;
  }
  __syncthreads();
  mn = red[0];
  mx = red[BKT_T / AGPU_WAVE];
  for (int w = 1; w < BKT_T / AGPU_WAVE; w++) {
    mn = red[w] < mn ? red[w] : mn;
    mx = red[BKT_T / AGPU_WAVE + w] > mx ? red[BKT_T / AGPU_WAVE + w] : mx;
  }
  const uint32_t origin = (mn >> 5) & ~31u;  // in WORDS, line-aligned
  const bool fast = mx < n_bits && ((mx >> 5) - origin) < (1u << TK2B_REL_BITS);
  auto store_bits = [&]() {
    if (threadIdx.x < TK2_GTILE / 64 && base + (uint64_t)threadIdx.x * 64 < total)
      vbits_slot[base / 64 + threadIdx.x] = (uint64_t)bitl[threadIdx.x * 2] | ((uint64_t)bitl[threadIdx.x * 2 + 1] << 32);
  };
  if (!fast) {  // out-of-range rows (bit 0) or a tile that spans many regions: row by row
#pragma unroll
    for (int q = 0; q < TK2_GE / 4; q++) {
      const u32x4 t = load4(q);
      const uint32_t sv[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
      for (int k = 0; k < 4; k++)
        if (live_at(q, k) && sv[k] < n_bits && ((bits[sv[k] >> 5] >> (sv[k] & 31)) & 1u)) {
          const uint32_t pos = ((uint32_t)q * BKT_T + threadIdx.x) * 4 + (uint32_t)k;
          atomicOr(&bitl[pos >> 5], 1u << (pos & 31));
        }
    }
    __syncthreads();
    store_bits();
    return;
  }
  uint32_t rank2[TK2_GE / 2];
#pragma unroll
  for (int q = 0; q < TK2_GE / 4; q++) {
    const u32x4 t = load4(q);
    const uint32_t sv[4] = {t.x, t.y, t.z, t.w};
    uint32_t r[4];
#pragma unroll
    for (int k = 0; k < 4; k++) r[k] = live_at(q, k) ? atomicAdd(&lcnt[((sv[k] >> 5) - origin) >> KS], 1u) : 0u;
    rank2[q * 2] = r[0] | (r[1] << 16);
    rank2[q * 2 + 1] = r[2] | (r[3] << 16);
  }
  __syncthreads();
  {  // exclusive scan of the counters: 256 of them, one per thread of the first four waves — or 2048, two per thread
    constexpr uint32_t KPT = NKEYS >= BKT_T ? NKEYS / BKT_T : 1;
    uint32_t c[KPT], sum = 0;
#pragma unroll
    for (uint32_t kk = 0; kk < KPT; kk++) {
      c[kk] = threadIdx.x * KPT + kk < NKEYS ? lcnt[threadIdx.x * KPT + kk] : 0u;
      sum += c[kk];
    }
    uint32_t incl = sum;
#pragma unroll
    for (int off = 1; off < AGPU_WAVE; off <<= 1) {
      const uint32_t o = (uint32_t)__shfl_up((int)incl, off);
      if (lane >= (uint32_t)off) incl += o;
    }
    if (lane == AGPU_WAVE - 1) wave_tot[wave] = incl;
    __syncthreads();
    uint32_t pre = 0;
    for (uint32_t w = 0; w < wave; w++) pre += wave_tot[w];
    uint32_t run = pre + incl - sum;
#pragma unroll
    for (uint32_t kk = 0; kk < KPT; kk++) {
      if (threadIdx.x * KPT + kk < NKEYS) lcnt[threadIdx.x * KPT + kk] = run;
      run += c[kk];
    }
    if (threadIdx.x == BKT_T - 1) tile_rows = run;
    __syncthreads();
  }
#pragma unroll
  for (int q = 0; q < TK2_GE / 4; q++) {
    const u32x4 t = load4(q);
    const uint32_t sv[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
    for (int k = 0; k < 4; k++)
      if (live_at(q, k)) {
        const uint32_t rel = (sv[k] >> 5) - origin;
        const uint32_t pos = ((uint32_t)q * BKT_T + threadIdx.x) * 4 + (uint32_t)k;
        const uint32_t rk = (rank2[q * 2 + (k >> 1)] >> ((k & 1) * 16)) & 0xFFFFu;
        sorted[lcnt[rel >> KS] + rk] = WIDE ? (rel << TK2_POS_BITS) | pos : (rel << 19) | ((sv[k] & 31u) << TK2_POS_BITS) | pos;
      }
  }
  __syncthreads();
  const uint32_t rows_here = tile_rows;
#pragma unroll
  for (int h0 = 0; h0 < TK2_GE; h0 += 8) {
    uint32_t ent[8], w[8], bp[8];
#pragma unroll
    for (int e = 0; e < 8; e++) {
      const uint32_t j = (uint32_t)(h0 + e) * BKT_T + threadIdx.x;
      ent[e] = j < rows_here ? sorted[j] : 0u;
      w[e] = j < rows_here ? bits[origin + (ent[e] >> (WIDE ? TK2_POS_BITS : 19))] : 0u;  // the L2-resident gather: neighbours share the line
      if constexpr (WIDE) bp[e] = j < rows_here ? srcs[base + (ent[e] & ((1u << TK2_POS_BITS) - 1u))] & 31u : 0u;  // the bit position: from the index entry itself
      else bp[e] = (ent[e] >> TK2_POS_BITS) & 31u;
    }
#pragma unroll
    for (int e = 0; e < 8; e++) {
      const uint32_t j = (uint32_t)(h0 + e) * BKT_T + threadIdx.x;
      const uint32_t pos = ent[e] & ((1u << TK2_POS_BITS) - 1u);
      if (j < rows_here && ((w[e] >> bp[e]) & 1u)) atomicOr(&bitl[pos >> 5], 1u << (pos & 31));
    }
  }
  __syncthreads();
  store_bits();
}

// Boolean take through the merge-back pipeline; AGPU_ERR_UNSUPPORTED when the shape does not qualify
// ent_out != nullptr (the Boolean put): the gathered bits leave as entries dst_idx[i] * 2 + bit instead of a bitmap
static agpu_status launch_take_bits_mergeback(agpu_pipeline* p, const uint32_t* bits, uint64_t n_bits, const uint32_t* si, uint64_t* out_bits,
                                              uint64_t n, const uint32_t* di = nullptr, uint64_t n_dst = 0, uint32_t* ent_out = nullptr, bool adaptive = false,
                                              void* tbits_tmp = nullptr) {
  if (n >= 0xFFFF0000ull || n_bits > 0xFFFFFFFFull || !aligned16(si) || p->capturing) return AGPU_ERR_UNSUPPORTED;
  const uint64_t n_words = (n_bits + 31) / 32;
  int rsw = 12;  // 2^12 words = 16 KiB of bitmap per region: two regions span < 2^13 words (the narrow entry's 13 bits)
  while (((n_words + ((uint64_t)1 << rsw) - 1) >> rsw) > BKT_MAX - 1) rsw++;  // > 2^29 bits (a 1e9-row column's bitmap): larger regions and
  const bool wide = rsw > 12;                                                    // the WIDE gather (18-bit entries: two regions of ≤ 2^17 words)
  if (rsw > 17) return AGPU_ERR_UNSUPPORTED;
  const int rs = rsw + 5;  // H2 / P2 / F2 key an index by (idx >> rs): idx is a BIT number here
  const uint32_t bs = (uint32_t)((n_words + ((uint64_t)1 << rsw) - 1) >> rsw);
  agpu_device* dev = p->dev;
  const uint32_t ntiles = (uint32_t)((n + TK2_TILE - 1) / TK2_TILE);
  const uint32_t gtiles = (uint32_t)((n + TK2_GTILE - 1) / TK2_GTILE);
  const uint32_t nbp = (bs + 1 + 3) & ~3u;
  const uint32_t nchunks = (ntiles + BKT_CHUNK - 1) / BKT_CHUNK;
  void *ctl_v = nullptr, *srcs_v = nullptr, *rank_v = nullptr, *cnt_v = nullptr, *off_v = nullptr, *csum_v = nullptr, *vslot_v = nullptr;
  agpu_status st = agpu_malloc(dev, sizeof(BktCtl), 0, &ctl_v);
  if (st == AGPU_OK) st = agpu_malloc(dev, 4 * n + 16, 0, &srcs_v);
  if (st == AGPU_OK) st = agpu_malloc(dev, 2 * n + 16, 0, &rank_v);
  if (st == AGPU_OK) st = agpu_malloc(dev, ((size_t)gtiles * TK2_GTILE) / 8 + 16, 0, &vslot_v);
  if (st == AGPU_OK) st = agpu_malloc(dev, (size_t)ntiles * nbp * 2, 0, &cnt_v);
  if (st == AGPU_OK) st = agpu_malloc(dev, (size_t)ntiles * nbp * 4, 0, &off_v);
  if (st == AGPU_OK) st = agpu_malloc(dev, (size_t)nchunks * nbp * 4, 0, &csum_v);
  if (st != AGPU_OK) st = AGPU_ERR_UNSUPPORTED;
  if (st == AGPU_OK) {
    BktCtl* ctl = static_cast<BktCtl*>(ctl_v);
    uint16_t* counts = static_cast<uint16_t*>(cnt_v);
    uint32_t* offsets = static_cast<uint32_t*>(off_v);
    uint32_t* csum = static_cast<uint32_t*>(csum_v);
    hipError_t e = hipMemsetAsync(ctl, 0, sizeof(BktCtl), p->stream);
    if (e != hipSuccess) {
      agpu_set_error("hipMemsetAsync failed: %s", hipGetErrorString(e));
      st = AGPU_ERR_HIP;
    } else {
      const dim3 cgrid((nbp + 255) / 256, nchunks);
      const BktCtl* gate = adaptive ? ctl : nullptr;  // see launch_take_mergeback
      if (adaptive)
        hipLaunchKernelGGL(idx_locality_kernel, dim3(LOC_BLOCKS), dim3(256), 0, p->stream, si, static_cast<const uint32_t*>(nullptr), n, 10, 0, ctl);
      hipLaunchKernelGGL(tk2_hist_kernel, dim3(ntiles), dim3(BKT_T), 0, p->stream, si, n, n_bits, rs, bs, p->flags, counts, nbp, ntiles, gate);
      hipLaunchKernelGGL(bkt_colsum_kernel, cgrid, dim3(256), 0, p->stream, counts, nbp, ntiles, csum, gate);
      hipLaunchKernelGGL(bkt_colscan_kernel, dim3((nbp + 255) / 256), dim3(256), 0, p->stream, csum, nbp, nchunks, ctl->hist_s, gate);
      hipLaunchKernelGGL(bkt_scan_kernel, dim3(1), dim3(BKT_T), 0, p->stream, ctl, bs, 0u, 1u, 1u);
      hipLaunchKernelGGL(bkt_offsets_kernel, cgrid, dim3(256), 0, p->stream, counts, csum, nbp, ntiles, ctl->base_s, offsets, gate);
      hipLaunchKernelGGL(tk2_partition_kernel, dim3((ntiles + 7) / 8 * 8), dim3(BKT_T), 0, p->stream, si, n, n_bits, rs, bs, offsets, nbp,
                         ntiles, static_cast<uint32_t*>(srcs_v), static_cast<uint16_t*>(rank_v), gate);
      if (wide)
        hipLaunchKernelGGL((tk2_gather_bits_kernel<true>), dim3((gtiles + 7) / 8 * 8), dim3(BKT_T), 0, p->stream, bits, n_bits,
                           static_cast<const uint32_t*>(srcs_v), n, gtiles, static_cast<uint64_t*>(vslot_v), gate);
      else
        hipLaunchKernelGGL((tk2_gather_bits_kernel<false>), dim3((gtiles + 7) / 8 * 8), dim3(BKT_T), 0, p->stream, bits, n_bits,
                           static_cast<const uint32_t*>(srcs_v), n, gtiles, static_cast<uint64_t*>(vslot_v), gate);
      if (ent_out)
        hipLaunchKernelGGL((tk2_merge_kernel<3>), dim3((ntiles + 7) / 8 * 8), dim3(BKT_T), 0, p->stream, si, n, n_bits, rs, bs, counts, offsets,
                           nbp, ntiles, static_cast<const uint16_t*>(rank_v), static_cast<const uint32_t*>(nullptr), ent_out,
                           static_cast<const uint32_t*>(vslot_v), static_cast<uint64_t*>(nullptr), di, n_dst, gate);
      else
        hipLaunchKernelGGL((tk2_merge_kernel<2>), dim3((ntiles + 7) / 8 * 8), dim3(BKT_T), 0, p->stream, si, n, n_bits, rs, bs, counts, offsets,
                           nbp, ntiles, static_cast<const uint16_t*>(rank_v), static_cast<const uint32_t*>(nullptr),
                           static_cast<uint32_t*>(nullptr), static_cast<const uint32_t*>(vslot_v), out_bits, static_cast<const uint32_t*>(nullptr),
                           (uint64_t)0, gate);
      if (adaptive && !ent_out) (void)launch_take_bits_direct(p, bits, n_bits, si, out_bits, n, &ctl->use_direct);
      if (adaptive && ent_out) {  // the Boolean put's entries the direct way: bits in natural order, then one pass that builds the entries
        (void)launch_take_bits_direct(p, bits, n_bits, si, tbits_tmp, n, &ctl->use_direct);
        hipLaunchKernelGGL(pb_entries_kernel, dim3((uint32_t)((n + 1023) / 1024)), dim3(256), 0, p->stream, si, di, static_cast<const uint32_t*>(tbits_tmp), n,
                           n_bits, n_dst, ent_out, &ctl->use_direct);
      }
      if (hipGetLastError() != hipSuccess) {
        agpu_set_error("merge-back take_bits launch failed");
        st = AGPU_ERR_HIP;
      }
    }
  }
  for (void* q : {csum_v, off_v, cnt_v, vslot_v, rank_v, srcs_v, ctl_v})
    if (q) (void)agpu_free(dev, q);
  return st;
}

// vbits_src != nullptr: the source's validity bitmap is gathered with the values into out_validity (agpu_take_validity)
#ifndef TK2_GTHREADS
#define TK2_GTHREADS BKT_T  // threads of a G2 workgroup (tile = 16 slots per thread): 1024 → two workgroups per CU; 512 → four (A/B, tools/probe)
#endif
#define TK2_GG(BITS_, WW, E, grid_, ...)                                                                                   \
  do {                                                                                                                     \
    if (ge == 8) hipLaunchKernelGGL((tk2_gather_kernel<8, BITS_, WW, TK2_GTHREADS, 8>), grid_, __VA_ARGS__);               \
    else hipLaunchKernelGGL((tk2_gather_kernel<8, BITS_, WW, TK2_GTHREADS, TK2_GE>), grid_, __VA_ARGS__);                 \
  } while (0)
static agpu_status launch_take_mergeback(agpu_pipeline* p, int width, const void* values, uint64_t n_src, const uint32_t* si, void* out,
                                         uint64_t n, const uint32_t* vbits_src = nullptr, uint64_t* out_validity = nullptr, bool adaptive = false,
                                         const uint32_t* put_di = nullptr, uint64_t put_n_dst = 0, const BktCtl* ext_gate = nullptr) {
  // put_di != nullptr: a PUT whose destination column is local — `out` is the destination array, the merge pass stores row i's value at
  // out[put_di[i]]; ext_gate: the control block the put's locality probe wrote its decision for this pipeline to
  if (n >= 0xFFFF0000ull || n_src > 0xFFFFFFFFull || !aligned16(si) || (!put_di && !aligned16(out)) || (put_di && !aligned16(put_di)) || p->capturing)
    return AGPU_ERR_UNSUPPORTED;
  if (width != 4 && width != 2 && width != 1) return AGPU_ERR_UNSUPPORTED;
  const int rs = bkt_region_bits(p, n_src, 4);
  const uint32_t bs = (uint32_t)((n_src + ((uint64_t)1 << rs) - 1) >> rs);
  if (bs + 1 > BKT_MAX) return AGPU_ERR_UNSUPPORTED;
  agpu_device* dev = p->dev;
  const uint32_t ntiles = (uint32_t)((n + TK2_TILE - 1) / TK2_TILE);
  // regions of 2^19 elements (sources over 4095 · 2^18 ≈ 1.07e9 elements): the 8-slot gather, whose 19-bit offsets cover a region — with 16
  // slots every tile would take the row-by-row path (2^30 rows: 63.6 → 75.7 G rows/s).  At 2^18-element regions (1e9 rows) the 16-slot gather
  // stays: its straddling tiles (1 in 16) cost less than half the coalescing (80.6 against 75.6).
  const int ge = rs >= 19 ? 8 : TK2_GE;
  const uint32_t gtiles = (uint32_t)((n + (uint64_t)ge * TK2_GTHREADS - 1) / ((uint64_t)ge * TK2_GTHREADS));
  const uint32_t nbp = (bs + 1 + 3) & ~3u;
  const uint32_t nchunks = (ntiles + BKT_CHUNK - 1) / BKT_CHUNK;
  void *ctl_v = nullptr, *srcs_v = nullptr, *vals_v = nullptr, *rank_v = nullptr, *cnt_v = nullptr, *off_v = nullptr, *csum_v = nullptr;
  void* vslot_v = nullptr;
  agpu_status st = agpu_malloc(dev, sizeof(BktCtl), 0, &ctl_v);
  if (st == AGPU_OK && vbits_src) st = agpu_malloc(dev, ((size_t)gtiles * ge * TK2_GTHREADS) / 8 + 16, 0, &vslot_v);
  if (st == AGPU_OK) st = agpu_malloc(dev, 4 * n + 16, 0, &srcs_v);
  if (st == AGPU_OK) st = agpu_malloc(dev, (size_t)width * n + 16, 0, &vals_v);
  if (st == AGPU_OK) st = agpu_malloc(dev, 2 * n + 16, 0, &rank_v);
  if (st == AGPU_OK) st = agpu_malloc(dev, (size_t)ntiles * nbp * 2, 0, &cnt_v);
  if (st == AGPU_OK) st = agpu_malloc(dev, (size_t)ntiles * nbp * 4, 0, &off_v);
  if (st == AGPU_OK) st = agpu_malloc(dev, (size_t)nchunks * nbp * 4, 0, &csum_v);
  if (st != AGPU_OK) st = AGPU_ERR_UNSUPPORTED;  // no room for the temporaries: the direct kernel needs none
  if (st == AGPU_OK) {
    BktCtl* ctl = static_cast<BktCtl*>(ctl_v);
    uint16_t* counts = static_cast<uint16_t*>(cnt_v);
    uint32_t* offsets = static_cast<uint32_t*>(off_v);
    uint32_t* csum = static_cast<uint32_t*>(csum_v);
    hipError_t e = hipMemsetAsync(ctl, 0, sizeof(BktCtl), p->stream);
    if (e != hipSuccess) {
      agpu_set_error("hipMemsetAsync failed: %s", hipGetErrorString(e));
      st = AGPU_ERR_HIP;
    } else {
      const dim3 cgrid((nbp + 255) / 256, nchunks);
      // adaptive (the auto policy): the locality probe decides on the device whether these kernels or the direct one behind them run
      const BktCtl* gate = ext_gate ? ext_gate : adaptive ? ctl : nullptr;
      if (adaptive)
        hipLaunchKernelGGL(idx_locality_kernel, dim3(LOC_BLOCKS), dim3(256), 0, p->stream, si, static_cast<const uint32_t*>(nullptr), n,
                           width == 4 ? 5 : width == 2 ? 6 : 7, 0, ctl);
      hipLaunchKernelGGL(tk2_hist_kernel, dim3(ntiles), dim3(BKT_T), 0, p->stream, si, n, n_src, rs, bs, p->flags, counts, nbp, ntiles, gate);
      hipLaunchKernelGGL(bkt_colsum_kernel, cgrid, dim3(256), 0, p->stream, counts, nbp, ntiles, csum, gate);
      hipLaunchKernelGGL(bkt_colscan_kernel, dim3((nbp + 255) / 256), dim3(256), 0, p->stream, csum, nbp, nchunks, ctl->hist_s, gate);
      hipLaunchKernelGGL(bkt_scan_kernel, dim3(1), dim3(BKT_T), 0, p->stream, ctl, bs, 0u, 1u, 1u);
      hipLaunchKernelGGL(bkt_offsets_kernel, cgrid, dim3(256), 0, p->stream, counts, csum, nbp, ntiles, ctl->base_s, offsets, gate);
      hipLaunchKernelGGL(tk2_partition_kernel, dim3((ntiles + 7) / 8 * 8), dim3(BKT_T), 0, p->stream, si, n, n_src, rs, bs, offsets, nbp,
                         ntiles, static_cast<uint32_t*>(srcs_v), static_cast<uint16_t*>(rank_v), gate);
      const dim3 ggrid((gtiles + 7) / 8 * 8), fgrid((ntiles + 7) / 8 * 8);
      uint64_t* vslot = static_cast<uint64_t*>(vslot_v);
#define TK2_GF(WW, E) /* G2 with 16 or 8 slots per thread (TK2_GG) */                                                                                                                    \
  case WW:                                                                                                                               \
    if (vbits_src) {                                                                                                                     \
      TK2_GG(true, WW, E, ggrid, dim3(TK2_GTHREADS), 0, p->stream, static_cast<const E*>(values), n_src,        \
                         static_cast<const uint32_t*>(srcs_v), n, gtiles, static_cast<E*>(vals_v), vbits_src, vslot, gate);              \
      hipLaunchKernelGGL((tk2_merge_kernel<1, WW>), fgrid, dim3(BKT_T), 0, p->stream, si, n, n_src, rs, bs, counts, offsets, nbp, ntiles, \
                         static_cast<const uint16_t*>(rank_v), static_cast<const E*>(vals_v), static_cast<E*>(out),                      \
                         reinterpret_cast<const uint32_t*>(vslot), out_validity, static_cast<const uint32_t*>(nullptr), (uint64_t)0,     \
                         gate);                                                                                                          \
    } else if (put_di) {                                                                                                                 \
      TK2_GG(false, WW, E, ggrid, dim3(TK2_GTHREADS), 0, p->stream, static_cast<const E*>(values), n_src, \
                         static_cast<const uint32_t*>(srcs_v), n, gtiles, static_cast<E*>(vals_v), static_cast<const uint32_t*>(nullptr), \
                         static_cast<uint64_t*>(nullptr), gate);                                                                         \
      hipLaunchKernelGGL((tk2_merge_kernel<4, WW>), fgrid, dim3(BKT_T), 0, p->stream, si, n, n_src, rs, bs, counts, offsets, nbp, ntiles, \
                         static_cast<const uint16_t*>(rank_v), static_cast<const E*>(vals_v), static_cast<E*>(out),                      \
                         static_cast<const uint32_t*>(nullptr), static_cast<uint64_t*>(nullptr), put_di, put_n_dst, gate, p->flags);     \
    } else {                                                                                                                             \
      TK2_GG(false, WW, E, ggrid, dim3(TK2_GTHREADS), 0, p->stream, static_cast<const E*>(values), n_src,       \
                         static_cast<const uint32_t*>(srcs_v), n, gtiles, static_cast<E*>(vals_v), static_cast<const uint32_t*>(nullptr), \
                         static_cast<uint64_t*>(nullptr), gate);                                                                         \
      hipLaunchKernelGGL((tk2_merge_kernel<0, WW>), fgrid, dim3(BKT_T), 0, p->stream, si, n, n_src, rs, bs, counts, offsets, nbp, ntiles, \
                         static_cast<const uint16_t*>(rank_v), static_cast<const E*>(vals_v), static_cast<E*>(out),                      \
                         static_cast<const uint32_t*>(nullptr), static_cast<uint64_t*>(nullptr), static_cast<const uint32_t*>(nullptr),  \
                         (uint64_t)0, gate);                                                                                             \
    }                                                                                                                                    \
    break;
      switch (width) {
        TK2_GF(4, uint32_t)
        TK2_GF(2, uint16_t)
        TK2_GF(1, uint8_t)
      }
#undef TK2_GF
      if (adaptive) {  // … and the direct form, which returns at once unless the probe chose it
        (void)launch_take_direct(p, width, values, n_src, si, out, n, &ctl->use_direct);
        if (vbits_src) (void)launch_take_bits_direct(p, vbits_src, n_src, si, out_validity, n, &ctl->use_direct);
      }
      if (hipGetLastError() != hipSuccess) {
        agpu_set_error("merge-back take launch failed");
        st = AGPU_ERR_HIP;
      }
    }
  }
  for (void* q : {vslot_v, csum_v, off_v, cnt_v, rank_v, vals_v, srcs_v, ctl_v})
    if (q) (void)agpu_free(dev, q);
  return st;
}

static agpu_status launch_put_through_take(agpu_pipeline* p, int width, const void* src, uint64_t n_src, const uint32_t* si, void* dst, uint64_t n,
                                           const uint32_t* di, uint64_t n_dst, const BktCtl* gate) {
  return launch_take_mergeback(p, width, src, n_src, si, dst, n, nullptr, nullptr, false, di, n_dst, gate);
}

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
