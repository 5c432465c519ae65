// common.hpp — shared internals of libarrow_gpu_hip.so (gfx950 only; no CPU path, no CUDA shims).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <unordered_map>
#include <utility>
#include <vector>

#include "../../include/arrow_gpu.h"

#define AGPU_STR2(x) #x
#define AGPU_STR(x) AGPU_STR2(x)

// ---------------------------------------------------------------- tuning knobs (bench sweeps; see agpu_set_tuning)
// A pipeline carries its OWN copy (taken from the process defaults when it is created, changed by
// agpu_pipeline_set_tuning), so one thread's sweep never changes the kernels another pipeline launches.
struct agpu_tuning {
  int64_t stream_grid;    // blocks for streaming kernels (0 = auto: one tile per block); > 0 forces a persistent grid-stride launch (tests: the loop paths)
  int64_t cmp_variant;    // 0 = ballot (dword loads), 1 = vector loads + nibble shuffle
  int64_t gather_bucket;  // take/put: 0 = auto (size thresholds + the device-side locality probe), 1 = always direct, 2 = always bucketed, 4 = like 2 but with the probe (tests)
  int64_t h2d_mode;       // host↔device staging of agpu_import/export_arrow: 0 = auto, 1 = pageable hipMemcpy, 2 = threaded pinned staging, 3 = hipHostRegister in place
  int64_t tiles;          // tiles per block of the kernels that issue the NEXT tile's loads before they evaluate the current one — the VALU-heavy f32 unary
                          // kernels, the widening casts and cast-headed chains, the LDS-table kernels (lut8 / pow): 0 = each kernel's static default (one; pow
                          // with a scalar exponent three; cast-headed chains with a transcendental step four), > 0 = this many (round 6: one key instead of heavy_tiles /
                          // cast_tiles / table_tiles, and no adaptive policy behind "auto" any more: docs/experiments.md R6.9)
  int64_t wave_lds;       // unused dynamic LDS per wave that caps the waves per CU of sin / cos f32, the ×2 / ×4 widening casts and the 8-bit table kernels: 0 = each kernel's measured default (6800 B ≈ 24 waves per CU; sin / cos and the u8 → 32-bit casts 10240 B ≈ 16), < 0 = no cap, > 0 = this many bytes
  int64_t sync_spin;      // agpu_pipeline_sync and uploads / downloads of ≤ 3840 bytes wait for a kernel that posts into the pipeline's pinned mailbox instead of hipStreamSynchronize (12 → 7 µs for "one kernel, one scalar back"): 0 = yes, spinning for at most 200 µs before the blocking wait; > 0 = this many µs; < 0 = off
};
#define AGPU_TUNE_KEYS 7
agpu_tuning agpu_tuning_defaults();  // snapshot of the process-wide defaults (atomics, runtime.hip)
bool agpu_mem_pool_enabled();        // process-wide "mem_pool" switch (device-level behaviour, not per pipeline)

// ---------------------------------------------------------------- handles
// A pooled HIP event that several waiters share.  All fields are guarded by agpu_device::mu.
struct agpu_event_ref {
  hipEvent_t ev;
  int refs;
  bool done;  // someone observed hipEventQuery == hipSuccess
};

// Reduction scratch.  Ref-counted (under agpu_device::mu) because a captured hipGraph bakes the pointer into its
// kernel nodes: the block outlives the stream's next, larger scratch for as long as such a graph exists.
struct agpu_scratch_block {
  void* ptr;
  size_t bytes;
  int refs;
};

// One HIP stream the library may have queued work on (owned + wrapped, live or idle).
struct agpu_stream_slot {
  hipStream_t stream = nullptr;
  bool owned = false;
  agpu_scratch_block* scratch = nullptr;  // travels with the stream through the idle pool
  std::atomic<uint64_t> enq{0};           // +1 when an ABI call on this stream starts, +1 when it has enqueued its work (odd = in progress)
  std::atomic<uint64_t> clean_enq{0};     // value of `enq` up to which the stream is KNOWN to have drained (a completed host wait: pipeline sync, small download, device wait)
  std::atomic<bool> capturing{false};     // the stream is in graph capture: nothing but its own pipeline may launch on it
  std::atomic<bool> exposed{false};       // its raw handle was handed out (agpu_pipeline_stream): work this library never saw may be queued on it —
                                          // like a wrapped stream it is never taken for drained by the device-level waits
  // ---- guarded by agpu_device::mu
  uint64_t mark_seq = 0;                  // value of `enq` the newest marker covers
  agpu_event_ref* mark = nullptr;         // newest marker event recorded on the stream (free / finish / destroy)
  uint64_t finish_gen = 0;                // device generation at which `finish_ev` was published
  agpu_event_ref* finish_ev = nullptr;    // marker of the last agpu_pipeline_finish / destroy on this stream
};

struct agpu_device {
  int ordinal;
  hipDeviceProp_t props;
  int num_cus;
  // set when a deadline-aware collective wait gave up (comm.hip): a collective nobody will ever join is queued on some
  // stream, so nothing may wait for the device any more — calls fail fast, destroy calls leak, the process should exit
  std::atomic<bool> poisoned{false};
  // 2 KiB of f64 {1/c, −log2(1/c)} pairs for f32 pow (elementwise.hip: pow_f32_dev), built once at device creation
  void* pow_table = nullptr;
  // 6 KiB behind it: f32 results of sin / cos / sinh for every u8 and every i8 value, by the SAME device functions as the f32
  // kernels (identical bits), built once per device — a block copies its 1 KiB instead of evaluating 256 functions
  void* lut8_tables = nullptr;

  // ---- resource pools (runtime.hip).  Measured on MI355X / ROCm 7: hipStreamCreate 4.3 ms + hipStreamDestroy 2.6 ms,
  // hipFree 0.2 ms (implicit device sync), hipMalloc of a 4 GB block 0.2–60 ms — against 0.19 ms for the kernel of a
  // 1e8-row add.  The reference's default API creates a pipeline and an output buffer PER OP, so both are pooled.
  struct CachedBlock {  // a freed device block; `pending` = markers of the streams that were busy at free time
    void* ptr;
    std::vector<agpu_event_ref*> pending;
    bool sync_all;  // freed while a stream was capturing (no marker could be recorded): device-sync before reuse
  };
  struct Slab {  // 2 MiB of HBM carved into equal blocks of one small size class
    void* base;
    int cls;
    uint32_t nblocks, nfree;
  };
  std::mutex mu;
  std::vector<agpu_stream_slot*> slots;             // every stream work may be queued on
  std::vector<agpu_stream_slot*> idle;              // owned streams without a live pipeline
  std::atomic<uint64_t> finish_gen{0};              // bumped by every published finish (cross-pipeline ordering)
  std::unordered_map<void*, size_t> block_size;     // live pooled blocks handed out by agpu_malloc (large and small)
  struct TableGroup {  // agpu_malloc_table: several columns carved out of one block, freed column by column
    void* base;
    uint32_t live;
  };
  std::unordered_map<void*, TableGroup*> table_member;  // column pointer → its group
  // Placed arenas (runtime.hip "pool placement"): pool blocks of ≥ 1 GiB are carved out of big hipMalloc'ed arenas at
  // multiples of 512 MiB plus a rotating colour, so that the separate outputs / inputs of ordinary agpu_malloc callers
  // get the layout agpu_malloc_table gives the columns of one table.
  struct Arena {
    char* base;
    std::vector<uint8_t> used;  // one flag per 512 MiB unit
    uint32_t live;              // units in use
  };
  struct ArenaBlock {
    uint32_t arena, first, units;
  };
  std::vector<Arena> arenas;
  std::unordered_map<void*, ArenaBlock> arena_block;  // pointer handed out (base + unit offset + colour) → its units
  uint32_t arena_colour = 0;                           // rotates 0 / 8 / 4 / 12 KiB over successive carvings
  std::multimap<size_t, CachedBlock> cache;         // size → freed blocks ≥ 1 MiB
  size_t cached_bytes = 0, cache_cap = 0;
  static constexpr int kSmallClasses = 13;          // 256 B … 1 MiB, powers of two
  std::deque<CachedBlock> small_free[kSmallClasses];
  std::map<uintptr_t, Slab> slabs;                  // base address → slab
  size_t slab_bytes = 0;
  std::vector<hipEvent_t> event_pool;
  // sticky error words (pinned host memory, 64 B apart): one per live pipeline, recycled only once the stream has
  // passed every kernel of the previous owner
  struct RetiredFlag {
    uint32_t* word;
    agpu_event_ref* after;
  };
  std::vector<uint32_t*> flag_free;
  std::vector<RetiredFlag> flag_retired;
  std::vector<void*> flag_slabs;
  // the DEVICE's mailbox (agpu_device_sync / agpu_device_download, runtime.hip device_wait_all): one pinned slot of the same layout, one wait at a time
  std::mutex mbox_mu;
  uint32_t* mbox = nullptr;
  uint64_t mbox_seq = 0;
  // page-locked staging chunks for host↔HBM transfers of pageable memory (arrow_cdata.hip); one transfer at a time
  struct StageSlot {
    void* host;
    hipEvent_t ev;  // recorded after the DMA that last used the slot
    bool used;
  };
  std::mutex stage_mu;
  std::vector<StageSlot> stage;
  // bounce slots of the small-transfer path (agpu_internal_bounce_copy): one per direction so an upload and a download
  // from two host threads do not serialise
  std::mutex bounce_mu[2];
  StageSlot bounce[2] = {{nullptr, nullptr, false}, {nullptr, nullptr, false}};
};
struct agpu_pipeline;
#define AGPU_STAGE_CHUNK ((size_t)4 << 20)
void agpu_internal_free_staging(agpu_device* dev);  // arrow_cdata.hip
#define AGPU_BOUNCE_MAX_BYTES ((size_t)4 << 20)  // = one stage slot
struct agpu_pipeline;
agpu_status agpu_internal_bounce_copy(agpu_pipeline* p, void* dev_ptr, void* host_ptr, size_t bytes, bool to_device);  // arrow_cdata.hip
agpu_status agpu_internal_host_copy(agpu_pipeline* p, void* dev_ptr, void* host_ptr, size_t bytes, bool to_device);    // arrow_cdata.hip: complete on return
// + 6 result tables of 256 f32 for the fused 8-bit kernels (elementwise.hip lut8_kernel): {u8, i8} × {sin, cos, sinh}
#define AGPU_LUT8_TABLES 6
#define AGPU_TABLE_BYTES (128 * 16 + AGPU_LUT8_TABLES * 256 * 4)
agpu_status agpu_internal_build_tables(void* pow_table);  // elementwise.hip; synchronous
agpu_status agpu_internal_fill_bytes(struct agpu_pipeline* p, void* out, uint32_t pattern, uint64_t bytes);  // elementwise.hip: fill_kernel, out 16-byte aligned

struct agpu_pipeline {
  agpu_device* dev;
  agpu_stream_slot* slot;
  hipStream_t stream;  // == slot->stream
  bool owns_stream;
  bool capturing;
  uint64_t seen_gen;   // device finish generation this stream has already ordered itself behind
  agpu_tuning tune;
  // sticky error word in pinned host memory, written by kernels (bit 0: take/put index out of range), read and cleared
  // by agpu_pipeline_sync — no pre-pass over the index column, no readback, the pipeline stays asynchronous
  uint32_t* flags;
  // the same 4 KiB pinned slot carries the pipeline's MAILBOX (runtime.hip pipeline_wait_mailbox): a sequence word at +128 that a one-wave
  // kernel posts behind everything queued so far — the host spins on it — and 3840 payload bytes at +256 (small downloads AND uploads).  `dirty`: an ABI call has bound the
  // pipeline since the last completed wait (an idle stream keeps the plain, cheap hipStreamSynchronize).
  uint64_t mbox_seq;
  bool dirty;
  // the last thing queued is a runtime copy (agpu_upload_async / agpu_download_async): a kernel posted behind a copy-engine transfer waits for
  // the engine's signal through a barrier packet — 10–15 µs more than the runtime's own wait on that signal (R5.10) — so the next sync is the runtime's
  bool copy_tail;
  // profiling [ref: CmpQuery compute_query.rs:7-89, insert_debug_marker gpu_device.rs:132]
  uint32_t profile;    // AGPU_PROF_* bits
  int scope_depth;
  hipEvent_t t0, t1;   // event pair around the outermost ABI call (created on first use)
  bool t_valid;
  const char* last_name;
};
#define AGPU_FLAG_SLOT_BYTES 4096
#define AGPU_MBOX_SEQ 128      // byte offset of the mailbox sequence word inside a pipeline's pinned slot
#define AGPU_MBOX_PAYLOAD 256  // byte offset of the mailbox payload: AGPU_MAILBOX_MAX_BYTES (include/arrow_gpu.h) = 4096 − 256
#define AGPU_FLAG_INDEX_RANGE 1u
#define AGPU_PROF_ROCTX 1u   // roctx range named after the ABI call / reference entry point around every launch
#define AGPU_PROF_TIMING 2u  // HIP event pair around every launch (agpu_pipeline_last_kernel_ns)
#define AGPU_PROF_LOG 4u     // wait + log "Time taken for compute pass" to stderr after every launch

struct agpu_event {
  agpu_device* dev;
  hipEvent_t ev;
};

struct agpu_graph {
  agpu_device* dev;
  hipGraph_t graph;
  hipGraphExec_t exec;
  agpu_scratch_block* scratch;  // the capturing pipeline's scratch block, kept alive for the graph's lifetime
};

// ---------------------------------------------------------------- errors
void agpu_set_error(const char* fmt, ...);

#define AGPU_HIP(call)                                                                             \
  do {                                                                                             \
    hipError_t _e = (call);                                                                        \
    if (_e != hipSuccess) {                                                                        \
      agpu_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(_e), __FILE__, __LINE__);   \
      return AGPU_ERR_HIP;                                                                         \
    }                                                                                              \
  } while (0)

#define AGPU_REQUIRE(cond, code, msg)              \
  do {                                             \
    if (!(cond)) {                                 \
      agpu_set_error("%s: %s", __func__, msg);     \
      return code;                                 \
    }                                              \
  } while (0)

// post-launch check that does not synchronise
#define AGPU_LAUNCH_CHECK()                                                          \
  do {                                                                               \
    hipError_t _e = hipGetLastError();                                               \
    if (_e != hipSuccess) {                                                          \
      agpu_set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(_e), __FILE__, __LINE__); \
      return AGPU_ERR_HIP;                                                           \
    }                                                                                \
  } while (0)

// Every ABI call that takes a pipeline opens a scope: bind the device, order the stream behind whatever other
// pipelines have finished since it last looked (runtime.hip: cross-pipeline ordering), open the profiling range;
// leaving the scope counts the call on the stream and closes the range.
agpu_status agpu_scope_enter(agpu_pipeline* p, const char* name);
void agpu_scope_exit(agpu_pipeline* p);
struct agpu_call_scope {
  agpu_pipeline* p = nullptr;
  ~agpu_call_scope() {
    if (p) agpu_scope_exit(p);
  }
};
#define AGPU_BIND_AS(p, name)                              \
  agpu_call_scope _agpu_scope;                             \
  do {                                                     \
    agpu_status _s = agpu_scope_enter((p), (name));        \
    if (_s != AGPU_OK) return _s;                          \
    _agpu_scope.p = (p);                                   \
  } while (0)
#define AGPU_BIND(p) AGPU_BIND_AS(p, __func__)  // internal helpers that open the scope name the ABI call explicitly

agpu_status agpu_scratch(agpu_pipeline* p, size_t bytes, void** out);
void agpu_scope_label(agpu_pipeline* p, const char* label);


// Occupancy caps (round 5, docs/experiments.md R5.5).  The VALU-heavy sin / cos, the widening casts and the LDS-table kernels of the 8-bit
// sources run 3–9 % FASTER with fewer waves per CU than the 32 their registers allow (sin 0.80 → 0.83 of the roof, cos 0.79 → 0.835, u16 → f32
// 0.81 → 0.845, sin_u8 0.74 → 0.81 at ≈ 24 waves; u8 → f32 0.79 → 0.81 at ≈ 16; four boxes, every process), while everything light or already
// LDS-bound loses (add −1.5 %, compare −11 %, log −20 %).  The cap is unused dynamic LDS requested at launch — bytes per WAVE of the block.
static inline unsigned wave_lds_for(const agpu_pipeline* p, unsigned dflt_bytes_per_wave, unsigned waves_per_block) {
  const int64_t t = p->tune.wave_lds;
  const unsigned per_wave = t > 0 ? (unsigned)(t > 65536 ? 65536 : t) : t < 0 ? 0u : dflt_bytes_per_wave;
  const uint64_t total = (uint64_t)per_wave * waves_per_block;
  return (unsigned)(total > 65536 ? 65536 : total);  // the default dynamic-LDS limit of a launch
}
#define AGPU_WAVE_LDS_24 6800u
#define AGPU_WAVE_LDS_16 10240u

// Grid for a streaming kernel that owns `tiles` block-tiles.  Measured on MI355X at 1e9 rows (profiles/
// r01_sweep_add_f32_1e9.json): ONE tile per block beats every persistent grid (6.54 vs ≤6.52 TB/s at 32768 blocks,
// 5.3 TB/s at 2048), so the default is grid = tiles; stream_grid > 0 forces a persistent grid-stride launch (kept for
// sweeps and for the tests of the loop paths).  Kernels still grid-stride, so any grid is correct.
static inline int stream_grid_for(const agpu_pipeline* p, uint64_t tiles) {
  uint64_t g = tiles;
  if (p->tune.stream_grid > 0) g = (uint64_t)p->tune.stream_grid;
  if (g > tiles) g = tiles;
  if (g > 0x3FFFFFFFull) g = 0x3FFFFFFFull;  // hipDim3.x limit headroom
  if (g < 1) g = 1;
  return (int)g;
}

// TWO LOCK-STEP STREAMS (round 6b).  A kernel with ONE input and ONE output stream over a big column runs 1.2–4.7 % faster when its
// blocks walk the tiles as two streams half a column apart — even blocks the first half, odd blocks the second — than front to back,
// measured by alternating the two orders in one process on the same table-placed buffers at 1e9 rows (tools/probe/two_streams_ab.py,
// two_streams_ab2.py → profiles/r06b_two_streams_ab*.json): add_scalar 0.853 → 0.863 of the roof, neg 0.852 → 0.865, sqrt 0.849 → 0.866,
// exp 0.851 → 0.865, sin 0.850 → 0.871, u16 → f32 0.836 → 0.857, u8 → u16 0.80 → 0.83, f32 → i16 0.845 → 0.858, (a + s)·t in one launch
// 0.794 → 0.832; from 128 MiB columns on (2^25 f32 rows +0.9 %, 2^27 +1.5 %, 2^29 +2.0 %).  Any distance of ≥ 64 MiB between the streams
// does and no bit of the channel hash matters (tools/probe/stream_split.hip: 64 MiB … 1.5 GiB alike; 2 MiB −7 %).  What does NOT gain:
// two-input kernels (f32 add −0.5 %: three placed streams already interleave; the 32-bit compare −2 %: 0.88 → 0.865 in three allocations),
// read-only kernels (a column reads at 0.87 in any order), sinh (−3.6 %) and chains with a transcendental step (+1.6 % / −3.7 % by process),
// ×4 width changes (u8 → f32, the 8-bit table kernels, f32 → u8: level), in-place kernels (level), four streams (level) and eight (−2 %).
// logical tile t → the tile it touches; half = 0 keeps the sequential order (SALU only: the block index is uniform)
__device__ __forceinline__ uint64_t two_streams(uint64_t t, uint64_t half) {
  return t < 2 * half ? (t >> 1) + ((t & 1) ? half : 0) : t;
}
// host side: the `half` argument of a launch over `ntiles` tiles whose bigger stream moves `tile_bytes` per tile — one tile per block only
// (an explicit grid or several tiles per block keep their own order)
static inline uint64_t two_streams_half(const agpu_pipeline* p, uint64_t ntiles, uint64_t tile_bytes) {
  return (p->tune.stream_grid == 0 && p->tune.tiles <= 1 && ntiles * tile_bytes >= (128ull << 20)) ? ntiles / 2 : 0;
}

// Grid for kernels that finish with an atomic on ONE word per block/wave (popcount, any, index max, checksum):
// same-address atomics serialise at ≈12 ns each, so the block count is capped and the kernels grid-stride.
static inline int atomic_grid_for(const agpu_pipeline* p, uint64_t work_blocks) {
  uint64_t g = (uint64_t)p->dev->num_cus * 4;
  if (g > work_blocks) g = work_blocks;
  if (g < 1) g = 1;
  return (int)g;
}

// ---------------------------------------------------------------- device helpers
#define AGPU_BLOCK 256
#define AGPU_WAVE 64

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
static inline bool aligned_to(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) & (a - 1)) == 0; }

#ifdef __HIPCC__
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <bool NT, typename V>
__device__ __forceinline__ V ld_vec(const V* p) {
  if constexpr (NT) return __builtin_nontemporal_load(p);
  else return *p;
}
template <bool NT, typename V>
__device__ __forceinline__ void st_vec(V* p, V v) {
  if constexpr (NT) __builtin_nontemporal_store(v, p);
  else *p = v;
}
// 16-byte streaming store with the sc1 bit beside nt — a cache policy the compiler cannot be asked for (it knows plain and
// nontemporal).  tools/probe/cache_policy.hip tries all eight sc0 / sc1 / nt combinations on both sides of an f32 add (loads:
// nt is what matters, the scope bits change nothing; stores: "sc1 nt" +0.7 %); in the product it is worth +2 % to the
// 8 B/row kernels (unary / scalar: 0.831 → 0.848) and +3 % to the ×2 widening casts (i16 → f32 0.785 → 0.81), nothing to
// the binary kernels, and it COSTS the 8-bit kernels and the ×4 widening casts 1–2 % AT 32 WAVES PER CU — so it is opt-in per call site.
// Round 5: under the occupancy cap (wave_lds_for below) the ×4 answer flips — sc1 nt is +1–2 % for the u8 → 32-bit casts, the 8-bit and 16-bit table
// kernels and the ×4 cast chain (docs/experiments.md R5.8); nontemporal LOADS stay 4–10 % ahead of plain ones with or without a cap
// (tools/archive/r05_ldnt.sh: sin 0.82 / 0.77, cos 0.825 / 0.73, add 0.835 / 0.775).
template <typename V>
__device__ __forceinline__ void st_vec_sc1(V* p, V v) {
  static_assert(sizeof(V) == 16, "16-byte vectors only");
#if defined(__HIP_DEVICE_COMPILE__)
  typedef uint32_t st_u32x4 __attribute__((ext_vector_type(4)));
  const st_u32x4 w = __builtin_bit_cast(st_u32x4, v);
  // s_nop: a store of more than 8 bytes reads its data registers over several cycles and the next VALU instruction may
  // overwrite them — the compiler pads its OWN stores for this hazard but cannot see through inline asm (found the hard
  // way: sinh_i16 came out with two of four components clobbered in lanes 12–15 of every 16)
  asm volatile("global_store_dwordx4 %0, %1, off sc1 nt\n\ts_nop 1" : : "v"(p), "v"(w) : "memory");
#else
  *p = v;
#endif
}

// Compile-time unrolled `for (u = 0; u < U; u++) f(u)`.  NOT a `#pragma unroll` loop on purpose: with U == 1 LICM sees a
// single-trip loop whose store address is loop-invariant, promotes the store out of it and the re-created store loses
// its !nontemporal metadata (found by diffing the ISA: the `nt` bit vanished and the add stream lost 7 %).
template <typename F, int... I>
__device__ __forceinline__ void static_for_impl(F& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int U, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(f, std::make_integer_sequence<int, U>{});
}

__device__ __forceinline__ uint64_t splitmix64_dev(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
__device__ __forceinline__ uint64_t row_hash_dev(uint64_t seed, uint64_t row) {
  return splitmix64_dev(seed ^ (row * 0x9E3779B97F4A7C15ull));
}
#endif
