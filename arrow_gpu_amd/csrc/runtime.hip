// runtime.hip — device / buffer / pipeline / event / graph half of the C ABI (include/arrow_gpu.h).
// Replaces arrow_gpu_array::gpu_utils::{GpuDevice, ArrowComputePipeline, CmpQuery}
// [ref: crates/array/src/gpu_utils/gpu_device.rs:29-514, compute_pipeline.rs:8-300, compute_query.rs:7-90].
// A pipeline is a HIP stream: launches are eager and ordered, `finish` is the (already satisfied) submit point.
#include <rocprofiler-sdk-roctx/roctx.h>

#include <chrono>
#include <cstdlib>
#include <thread>
#include <memory>

#include <algorithm>

#include "common.hpp"

#define AGPU_SMALL_MAX ((size_t)1 << 20)        // largest size class of the slab pool (two blocks per slab)
#define AGPU_POOL_MIN_BYTES (AGPU_SMALL_MAX + 1) // larger blocks: size-keyed cache of whole hipMalloc'ed blocks (2 MiB granules)
#define AGPU_POOL_GRANULE ((size_t)2 << 20)
#define AGPU_SMALL_MIN ((size_t)256)            // smallest size class of the slab pool
#define AGPU_SLAB_BYTES ((size_t)2 << 20)
#define AGPU_SLAB_CAP ((size_t)1 << 30)         // never more than 1 GiB of slabs just to avoid waiting on a marker

static thread_local char g_err[512] = "";

void agpu_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// ---------------------------------------------------------------- process-wide tuning defaults
// Plain atomics: agpu_set_tuning may be called from any thread; a pipeline snapshots them when it is created.
static std::atomic<int64_t> g_tune_default[AGPU_TUNE_KEYS] = {/*stream_grid*/ {0}, /*cmp_variant*/ {0}, /*gather_bucket*/ {0}, /*h2d_mode*/ {0},
                                                                 /*tiles*/ {0}, /*wave_lds*/ {0}, /*sync_spin*/ {0}};
// DEV SWITCH (tools/probe, docs/experiments.md R5): AGPU_DEVICE_MALLOC_FLAGS=<hipExtMallocWithFlags flags> makes every block the
// pool, the arenas and the tables take from the driver a hipDeviceMallocContiguous (4) / Uncached (3) / Finegrained (1) one.
// Unset or 0 = plain hipMalloc, which is what the product ships with.
static hipError_t dev_malloc(void** out, size_t bytes) {
  static const unsigned flags = [] { const char* e = getenv("AGPU_DEVICE_MALLOC_FLAGS"); return e && *e ? (unsigned)strtoul(e, nullptr, 0) : 0u; }();
  return flags ? hipExtMallocWithFlags(out, bytes, flags) : hipMalloc(out, bytes);
}
// AGPU_ALLOC_TRACE=1: one stderr line per large allocation saying where the block came from (cache / arena carving / the driver)
static bool alloc_trace() {
  static const bool on = [] { const char* e = getenv("AGPU_ALLOC_TRACE"); return e && *e && *e != '0'; }();
  return on;
}
static std::atomic<int64_t> g_mem_pool{1};  // 1 = recycle device blocks and idle streams (default), 0 = hipMalloc/hipFree every time
static std::atomic<int64_t> g_pool_arena{1};  // 1 = pool blocks of ≥ 1 GiB come out of placed arenas (default), 0 = one hipMalloc each
static const char* const g_tune_keys[AGPU_TUNE_KEYS] = {"stream_grid", "cmp_variant", "gather_bucket", "h2d_mode", "tiles", "wave_lds", "sync_spin"};

agpu_tuning agpu_tuning_defaults() {
  agpu_tuning t;
  int64_t* f = reinterpret_cast<int64_t*>(&t);
  for (int i = 0; i < AGPU_TUNE_KEYS; i++) f[i] = g_tune_default[i].load(std::memory_order_relaxed);
  return t;
}
bool agpu_mem_pool_enabled() { return g_mem_pool.load(std::memory_order_relaxed) != 0; }
static int tune_index(const char* key) {
  if (!key) return -1;
  for (int i = 0; i < AGPU_TUNE_KEYS; i++)
    if (!strcmp(key, g_tune_keys[i])) return i;
  return -1;
}
static_assert(sizeof(agpu_tuning) == AGPU_TUNE_KEYS * sizeof(int64_t), "agpu_tuning is indexed as an int64 array");

// AGPU_PROFILE=<bits>: 1 = roctx ranges, 2 = per-launch HIP-event timing, 4 = also wait + log every launch (stderr)
static uint32_t env_profile() {
  static const uint32_t v = [] {
    const char* e = getenv("AGPU_PROFILE");
    if (!e || !*e) return 0u;
    uint32_t b = (uint32_t)strtoul(e, nullptr, 0);
    if (b & AGPU_PROF_LOG) b |= AGPU_PROF_TIMING;
    return b;
  }();
  return v;
}

// ---------------------------------------------------------------- shared events (all under dev->mu)
static hipEvent_t event_get_locked(agpu_device* dev) {
  hipEvent_t ev = nullptr;
  if (!dev->event_pool.empty()) {
    ev = dev->event_pool.back();
    dev->event_pool.pop_back();
  } else if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) {
    (void)hipGetLastError();
    ev = nullptr;
  }
  return ev;
}
static void ref_release_locked(agpu_device* dev, agpu_event_ref* r) {
  if (r && --r->refs == 0) {
    dev->event_pool.push_back(r->ev);
    delete r;
  }
}
static bool ref_done_locked(agpu_event_ref* r) {
  if (!r || r->done) return true;
  hipError_t e = hipEventQuery(r->ev);
  if (e == hipSuccess) r->done = true;
  else (void)hipGetLastError();
  return r->done;
}
// Newest marker of a stream, recording a fresh one if ABI calls completed on it since the last (wrapped streams are
// always treated as dirty: their owner enqueues work the library does not see).  *ok = false if it could not be
// recorded (the stream is being captured into a graph).
static agpu_event_ref* slot_mark_locked(agpu_device* dev, agpu_stream_slot* s, bool* ok) {
  *ok = true;
  // `enq` is odd while an ABI call is in progress on the stream (it may have enqueued part of its work already, e.g. a
  // call that frees its own temporary): record a fresh marker and leave the stream dirty
  const uint64_t e = s->enq.load(std::memory_order_acquire);
  if (s->owned && !(e & 1) && e == s->mark_seq) return s->mark;
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(s->stream, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) {
    (void)hipGetLastError();  // an event recorded now would become a graph node
    *ok = false;
    return s->mark;
  }
  hipEvent_t ev = event_get_locked(dev);
  if (!ev || hipEventRecord(ev, s->stream) != hipSuccess) {
    (void)hipGetLastError();
    if (ev) dev->event_pool.push_back(ev);
    *ok = false;
    return s->mark;
  }
  ref_release_locked(dev, s->mark);
  s->mark = new agpu_event_ref{ev, 1, false};
  s->mark_seq = e & ~(uint64_t)1;  // an in-progress call is never counted as covered
  return s->mark;
}

static void scratch_release_locked(agpu_scratch_block* b) {  // the caller made sure no queued work still uses it
  if (b && --b->refs == 0) {
    (void)hipFree(b->ptr);
    delete b;
  }
}

extern "C" {

int32_t agpu_abi_version(void) { return AGPU_ABI_VERSION; }
const char* agpu_last_error(void) { return g_err; }
const char* agpu_build_info(void) {
  return "arrow_gpu_hip gfx950 hip-" AGPU_STR(HIP_VERSION_MAJOR) "." AGPU_STR(HIP_VERSION_MINOR);
}

size_t agpu_dtype_size(agpu_dtype t) {
  switch (t) {
    case AGPU_F32: case AGPU_U32: case AGPU_I32: case AGPU_DATE32: return 4;
    case AGPU_U16: case AGPU_I16: return 2;
    case AGPU_U8: case AGPU_I8: return 1;
    default: return 0;
  }
}
size_t agpu_bitmap_bytes(uint64_t n_bits) { return (size_t)((n_bits + 63) / 64 * 8); }

// ---------------------------------------------------------------- device
agpu_status agpu_device_count(int32_t* out_count) {
  AGPU_REQUIRE(out_count, AGPU_ERR_ARG, "null out_count");
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    n = 0;
  }
  *out_count = n;
  return AGPU_OK;
}

agpu_status agpu_device_create(int32_t ordinal, agpu_device** out_device) {
  AGPU_REQUIRE(out_device, AGPU_ERR_ARG, "null out_device");
  *out_device = nullptr;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) {
    (void)hipGetLastError();
    agpu_set_error("no HIP device visible (%s); this library has no CPU fallback",
                   e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    return AGPU_ERR_NO_DEVICE;
  }
  AGPU_REQUIRE(ordinal >= 0 && ordinal < n, AGPU_ERR_ARG, "device ordinal out of range");
  {  // AGPU_SYNC_SPIN=<n>: the process-wide default of tuning "sync_spin" without a code change (−1 = every wait through the runtime, as before round 5)
    static const bool once = [] {
      const char* e = getenv("AGPU_SYNC_SPIN");
      if (e && *e) g_tune_default[tune_index("sync_spin")].store(strtoll(e, nullptr, 10), std::memory_order_relaxed);
      return true;
    }();
    (void)once;
  }
  std::unique_ptr<agpu_device> d(new agpu_device());  // released on every early-error return below
  d->ordinal = ordinal;
  AGPU_HIP(hipSetDevice(ordinal));
  AGPU_HIP(hipGetDeviceProperties(&d->props, ordinal));
  d->num_cus = d->props.multiProcessorCount;
  if (strncmp(d->props.gcnArchName, "gfx950", 6) != 0) {
    agpu_set_error("device %d is %s; this library is built for gfx950 only", ordinal, d->props.gcnArchName);
    return AGPU_ERR_NO_DEVICE;
  }
  // Function tables (10 KiB): ONE copy per physical device for the life of the process — every agpu_device created on
  // the ordinal shares it, and the device-side pointer elementwise.hip keeps (g_pow_tab, read by tails and fused chains)
  // never dangles when one of several handles on the same GPU is destroyed.
  {
    static std::mutex tab_mu;
    static void* tab_of[64] = {nullptr};
    std::lock_guard<std::mutex> lock(tab_mu);
    const int slot = ordinal < 64 ? ordinal : 63;
    if (!tab_of[slot]) {
      void* t = nullptr;
      hipError_t me = hipMalloc(&t, AGPU_TABLE_BYTES);
      if (me != hipSuccess) {
        agpu_set_error("hipMalloc of the function tables failed: %s", hipGetErrorString(me));
        return AGPU_ERR_HIP;
      }
      agpu_status ts = agpu_internal_build_tables(t);
      if (ts != AGPU_OK) {
        (void)hipFree(t);
        return ts;
      }
      tab_of[slot] = t;
    }
    d->pow_table = tab_of[slot];
    d->lut8_tables = static_cast<char*>(d->pow_table) + 128 * 16;
  }
  d->cache_cap = d->props.totalGlobalMem / 2;  // cached (idle) blocks never hold more than half of HBM
  *out_device = d.release();
  return AGPU_OK;
}

// ---------------------------------------------------------------- pool placement: arenas for blocks of ≥ 1 GiB
// The columns an ordinary caller allocates one by one (a, b, and the output every op of the reference's API allocates for
// itself) used to be separate hipMallocs, and where they lie relative to each other decides up to 10 % of a compare's and
// 2–4 % of an add's bandwidth (the HBM channel hash, DESIGN.md §3: bench step over nine pool blocks 6 400–6 500 GB/s against
// 6 850 for agpu_malloc_table's layout).  So big pool blocks are carved out of ARENAS — one hipMalloc of up to 32 GiB each,
// inside which physical placement follows the virtual one — at multiples of 512 MiB (no hash bit below 2^29 differs between
// row i of two blocks) plus a colour of 0 / 8 / 4 / 12 KiB that rotates over successive carvings: consecutive allocations
// differ in the strongest hash bit, any two of four in the first or second — the rule agpu_malloc_table applies to the
// columns of one table, now without the caller knowing.  A freed block goes to the size-keyed cache like any other (and
// keeps its colour); its units return to the arena when the cache lets go of it; an arena whose units are all free goes
// back to the driver at trim time.  Cost: sizes round up to 512 MiB (7 % for a 4e9-byte column), and the first big block
// reserves a whole arena.  Tuning "pool_arena" = 0 switches it off (A/B: tools/probe/bench_layout.py).
#define AGPU_ARENA_UNIT ((size_t)512 << 20)
#define AGPU_ARENA_MIN_BLOCK ((size_t)1 << 30)
#define AGPU_ARENA_UNITS 64  // 32 GiB
static const size_t kArenaColour[4] = {0, 8192, 4096, 12288};

static size_t arena_padded(size_t bytes) { return (bytes + 16384 + AGPU_ARENA_UNIT - 1) / AGPU_ARENA_UNIT * AGPU_ARENA_UNIT; }

// Which colour for a new block?  `neighbours` = buffers the caller will read or write TOGETHER with it (agpu_malloc_like:
// the inputs of the op whose output this is).  Colour c sits at bits 13..12 of the offset: bit 13 feeds the strongest
// bit of the channel hash, bit 12 the second.  Pick the colour whose worst relation to a neighbour inside `arena` is best
// (differs in bit 13: 2 points, in bit 12: 1 point); ties and the no-neighbour case fall to the rotating counter.
static uint32_t arena_pick_colour_locked(agpu_device* dev, uint32_t arena, const void* const* neighbours, int n_neighbours) {
  const uint32_t rot = dev->arena_colour++ & 3;
  if (n_neighbours <= 0 || !neighbours) return rot;
  const agpu_device::Arena& a = dev->arenas[arena];
  const uintptr_t lo = reinterpret_cast<uintptr_t>(a.base), hi = lo + a.used.size() * AGPU_ARENA_UNIT;
  int best = -1;
  uint32_t best_c = rot;
  for (uint32_t k = 0; k < 4; k++) {
    const uint32_t c = (rot + k) & 3;  // start at the rotating colour so that ties rotate
    const uintptr_t off = kArenaColour[c];
    int worst = 3;
    bool any = false;
    for (int i = 0; i < n_neighbours; i++) {
      const uintptr_t q = reinterpret_cast<uintptr_t>(neighbours[i]);
      if (!q || q < lo || q >= hi) continue;  // another arena / an ordinary block: its physical relation is unknown
      any = true;
      const uintptr_t d = ((q - lo) ^ off) & 0x3000;
      const int score = (d & 0x2000 ? 2 : 0) + (d & 0x1000 ? 1 : 0);
      if (score < worst) worst = score;
    }
    if (!any) return rot;
    if (worst > best) {
      best = worst;
      best_c = c;
    }
  }
  return best_c;
}

// a cached arena block is handed out again: its 16 KiB of colour room lets the colour be chosen afresh for the new use.
// Only a request that was arena-padded (`has_room`) reserved that room: any other one (a table's block, pool_arena = 0) may
// need every byte of the units, so it gets the block at colour 0 — a pointer moved up by 4–12 KiB would let the request's
// tail (and its zero fill) run into the next arena block.
static void* arena_recolour_locked(agpu_device* dev, void* ptr, bool has_room, const void* const* neighbours, int n_neighbours) {
  auto it = dev->arena_block.find(ptr);
  if (it == dev->arena_block.end()) return ptr;
  const agpu_device::ArenaBlock blk = it->second;
  char* q = dev->arenas[blk.arena].base + (size_t)blk.first * AGPU_ARENA_UNIT +
            (has_room ? kArenaColour[arena_pick_colour_locked(dev, blk.arena, neighbours, n_neighbours)] : 0);
  if (q != ptr) {
    dev->arena_block.erase(it);
    dev->arena_block[q] = blk;
  }
  return q;
}

// carve `units` units; returns nullptr when no arena has room and a new one cannot be had
static void* arena_carve_locked(agpu_device* dev, uint32_t units, const void* const* neighbours, int n_neighbours) {
  auto try_arena = [&](uint32_t ai) -> void* {
    agpu_device::Arena& a = dev->arenas[ai];
    const uint32_t total = (uint32_t)a.used.size();
    if (!a.base || total - a.live < units) return nullptr;
    for (uint32_t u = 0, run = 0; u < total; u++) {
      run = a.used[u] ? 0 : run + 1;
      if (run == units) {
        const uint32_t first = u + 1 - units;
        for (uint32_t k = first; k <= u; k++) a.used[k] = 1;
        a.live += units;
        char* ptr = a.base + (size_t)first * AGPU_ARENA_UNIT + kArenaColour[arena_pick_colour_locked(dev, ai, neighbours, n_neighbours)];
        dev->arena_block[ptr] = agpu_device::ArenaBlock{ai, first, units};
        return ptr;
      }
    }
    return nullptr;
  };
  // the arena a neighbour lives in first: only inside one arena is the relative placement known
  for (int i = 0; i < n_neighbours && neighbours; i++)
    for (uint32_t ai = 0; ai < dev->arenas.size(); ai++) {
      const agpu_device::Arena& a = dev->arenas[ai];
      const uintptr_t q = reinterpret_cast<uintptr_t>(neighbours[i]), lo = reinterpret_cast<uintptr_t>(a.base);
      if (a.base && q >= lo && q < lo + a.used.size() * AGPU_ARENA_UNIT)
        if (void* ptr = try_arena(ai)) return ptr;
    }
  for (uint32_t ai = 0; ai < dev->arenas.size(); ai++)
    if (void* ptr = try_arena(ai)) return ptr;
  // a new arena: 32 GiB, or what the block needs if that is more; halve towards the need while the driver refuses
  size_t free_b = 0, total_b = 0;
  (void)hipMemGetInfo(&free_b, &total_b);
  uint32_t want = units > AGPU_ARENA_UNITS ? units : AGPU_ARENA_UNITS;
  while (want > units && (size_t)want * AGPU_ARENA_UNIT > free_b / 2) want = want / 2 > units ? want / 2 : units;
  for (;;) {
    void* base = nullptr;
    if (dev_malloc(&base, (size_t)want * AGPU_ARENA_UNIT) == hipSuccess) {
      uint32_t slot = 0;
      for (; slot < dev->arenas.size(); slot++)
        if (!dev->arenas[slot].base) break;
      if (slot == dev->arenas.size()) dev->arenas.push_back(agpu_device::Arena{});
      dev->arenas[slot] = agpu_device::Arena{static_cast<char*>(base), std::vector<uint8_t>(want, 0), 0};
      return try_arena(slot);
    }
    (void)hipGetLastError();
    if (want == units) return nullptr;
    want = want / 2 > units ? want / 2 : units;
  }
}

// give a block's units back; true when `ptr` was an arena block.  A fully free arena stays reserved until trim.
static bool arena_release_locked(agpu_device* dev, void* ptr) {
  auto it = dev->arena_block.find(ptr);
  if (it == dev->arena_block.end()) return false;
  agpu_device::Arena& a = dev->arenas[it->second.arena];
  for (uint32_t k = it->second.first; k < it->second.first + it->second.units; k++) a.used[k] = 0;
  a.live -= it->second.units;
  dev->arena_block.erase(it);
  return true;
}

static void arena_trim_locked(agpu_device* dev) {
  for (agpu_device::Arena& a : dev->arenas)
    if (a.base && a.live == 0) {
      (void)hipFree(a.base);
      a.base = nullptr;
      a.used.clear();
    }
}

// release every cached block and the scratch of idle streams; blocks until the device is idle
static void device_trim_locked(agpu_device* dev) {
  (void)hipDeviceSynchronize();
  for (auto& kv : dev->cache) {
    for (agpu_event_ref* r : kv.second.pending) ref_release_locked(dev, r);
    if (!arena_release_locked(dev, kv.second.ptr)) (void)hipFree(kv.second.ptr);
  }
  arena_trim_locked(dev);
  dev->cache.clear();
  dev->cached_bytes = 0;
  // small blocks: a slab goes back to the driver when every block carved from it is free
  for (auto it = dev->slabs.begin(); it != dev->slabs.end();) {
    agpu_device::Slab& sl = it->second;
    if (sl.nfree != sl.nblocks) {
      ++it;
      continue;
    }
    const uintptr_t lo = it->first, hi = lo + AGPU_SLAB_BYTES;
    auto& fl = dev->small_free[sl.cls];
    for (auto b = fl.begin(); b != fl.end();) {
      const uintptr_t a = reinterpret_cast<uintptr_t>(b->ptr);
      if (a >= lo && a < hi) {
        for (agpu_event_ref* r : b->pending) ref_release_locked(dev, r);
        b = fl.erase(b);
      } else {
        ++b;
      }
    }
    (void)hipFree(sl.base);
    dev->slab_bytes -= AGPU_SLAB_BYTES;
    it = dev->slabs.erase(it);
  }
  for (agpu_stream_slot* s : dev->idle) {
    if (s->scratch) scratch_release_locked(s->scratch);
    s->scratch = nullptr;
  }
}

static const char* kPoisonedMsg = "the device is poisoned: a collective timed out and is still queued (comm.hip) — exit this process";
#define AGPU_NOT_POISONED(dev)                                          \
  do {                                                                  \
    if ((dev)->poisoned.load(std::memory_order_acquire)) {              \
      agpu_set_error("%s", kPoisonedMsg);                               \
      return AGPU_ERR_HIP;                                              \
    }                                                                   \
  } while (0)

agpu_status agpu_device_trim(agpu_device* dev) {
  AGPU_REQUIRE(dev, AGPU_ERR_ARG, "null device");
  AGPU_NOT_POISONED(dev);
  AGPU_HIP(hipSetDevice(dev->ordinal));
  std::lock_guard<std::mutex> lock(dev->mu);
  device_trim_locked(dev);
  return AGPU_OK;
}

agpu_status agpu_device_pool_info(agpu_device* dev, uint64_t* out_cached_bytes, uint64_t* out_cached_blocks,
                                  uint64_t* out_idle_streams) {
  AGPU_REQUIRE(dev, AGPU_ERR_ARG, "null device");
  std::lock_guard<std::mutex> lock(dev->mu);
  if (out_cached_bytes) *out_cached_bytes = dev->cached_bytes;
  if (out_cached_blocks) *out_cached_blocks = dev->cache.size();
  if (out_idle_streams) *out_idle_streams = dev->idle.size();
  return AGPU_OK;
}

agpu_status agpu_device_small_pool_info(agpu_device* dev, uint64_t* out_slab_bytes, uint64_t* out_free_blocks,
                                        uint64_t* out_live_blocks) {
  AGPU_REQUIRE(dev, AGPU_ERR_ARG, "null device");
  std::lock_guard<std::mutex> lock(dev->mu);
  uint64_t nfree = 0, total = 0;
  for (auto& kv : dev->slabs) {
    nfree += kv.second.nfree;
    total += kv.second.nblocks;
  }
  if (out_slab_bytes) *out_slab_bytes = dev->slab_bytes;
  if (out_free_blocks) *out_free_blocks = nfree;
  if (out_live_blocks) *out_live_blocks = total - nfree;
  return AGPU_OK;
}

bool agpu_internal_comm_wait_orphans(int64_t wait_ms);  // comm.hip
agpu_status agpu_device_destroy(agpu_device* dev) {
  if (!dev) return AGPU_OK;
  (void)hipSetDevice(dev->ordinal);
  // init helpers of one-rank communicators whose RCCL bootstrap came up late (or not yet): they must be out of RCCL before the runtime goes
  (void)agpu_internal_comm_wait_orphans(5000);
  // a stuck collective blocks every hipFree / hipStreamDestroy: leak the lot, the process is on its way out
  if (dev->poisoned.load(std::memory_order_acquire)) return AGPU_OK;
  {
    std::lock_guard<std::mutex> lock(dev->mu);
    device_trim_locked(dev);
    for (agpu_stream_slot* s : dev->slots) {
      ref_release_locked(dev, s->mark);
      ref_release_locked(dev, s->finish_ev);
      if (s->scratch) scratch_release_locked(s->scratch);
      if (s->owned) (void)hipStreamDestroy(s->stream);
      delete s;
    }
    dev->slots.clear();
    dev->idle.clear();
    for (auto& rf : dev->flag_retired) ref_release_locked(dev, rf.after);
    dev->flag_retired.clear();
    for (auto& kv : dev->slabs) (void)hipFree(kv.second.base);  // blocks the caller leaked
    dev->slabs.clear();
    for (agpu_device::Arena& a : dev->arenas)
      if (a.base) (void)hipFree(a.base);  // including arenas that still hold leaked blocks
    dev->arenas.clear();
    dev->arena_block.clear();
    {  // table groups whose columns the caller leaked (the blocks themselves are the caller's leak, like any other)
      std::vector<agpu_device::TableGroup*> groups;
      for (auto& kv : dev->table_member)
        if (std::find(groups.begin(), groups.end(), kv.second) == groups.end()) groups.push_back(kv.second);
      for (agpu_device::TableGroup* g : groups) delete g;
      dev->table_member.clear();
    }
    for (hipEvent_t e : dev->event_pool) (void)hipEventDestroy(e);
    dev->event_pool.clear();
    for (void* f : dev->flag_slabs) (void)hipHostFree(f);
    dev->flag_slabs.clear();
  }
  agpu_internal_free_staging(dev);
  delete dev;
  return AGPU_OK;
}

static agpu_status device_wait_all(agpu_device* dev, const void* src_dev, size_t bytes, void* dst_host);  // below, with the mailbox
agpu_status agpu_device_sync(agpu_device* dev) {
  AGPU_REQUIRE(dev, AGPU_ERR_ARG, "null device");
  AGPU_NOT_POISONED(dev);
  AGPU_HIP(hipSetDevice(dev->ordinal));
  return device_wait_all(dev, nullptr, 0, nullptr);
}

agpu_status agpu_device_name(agpu_device* dev, char* out, size_t out_cap) {
  AGPU_REQUIRE(dev && out && out_cap, AGPU_ERR_ARG, "null argument");
  snprintf(out, out_cap, "%s", dev->props.gcnArchName);
  return AGPU_OK;
}

agpu_status agpu_device_ordinal(agpu_device* dev, int32_t* out_ordinal) {
  AGPU_REQUIRE(dev && out_ordinal, AGPU_ERR_ARG, "null argument");
  *out_ordinal = dev->ordinal;
  return AGPU_OK;
}

agpu_status agpu_device_mem_info(agpu_device* dev, uint64_t* out_free, uint64_t* out_total) {
  AGPU_REQUIRE(dev && out_free && out_total, AGPU_ERR_ARG, "null argument");
  AGPU_HIP(hipSetDevice(dev->ordinal));
  size_t f = 0, t = 0;
  AGPU_HIP(hipMemGetInfo(&f, &t));
  *out_free = f;
  *out_total = t;
  return AGPU_OK;
}

// ---------------------------------------------------------------- buffers
// Two pools, both keyed by size.  Blocks > 512 KiB are rounded up to 2 MiB multiples and recycled through dev->cache.
// Smaller blocks — the reference's whole test-suite and examples/simple.rs live at 5–100 elements — come from 2 MiB
// slabs carved into one power-of-two size class each (256 B … 512 KiB), so a tiny array costs neither a hipMalloc nor
// hipFree's device-wide synchronisation.  A freed block may still be read or written by work queued on some stream (the
// host layers drop their references when a pipeline is released, not when the GPU is done; hipFree used to cover that
// by synchronising), so agpu_free attaches the marker event of every stream that has work outstanding and agpu_malloc
// waits for those before handing the block out again — idle streams cost nothing, and small blocks are reused oldest
// first so the markers have normally completed long ago.
static void collect_pending_locked(agpu_device* dev, std::vector<agpu_event_ref*>* pending, bool* ok) {
  *ok = true;
  for (agpu_stream_slot* s : dev->slots) {
    bool rec = true;
    agpu_event_ref* m = slot_mark_locked(dev, s, &rec);
    if (!rec) {
      *ok = false;
      return;
    }
    if (m && !ref_done_locked(m)) {
      m->refs++;
      pending->push_back(m);
    }
  }
}

static agpu_status wait_pending(agpu_device* dev, std::vector<agpu_event_ref*>& pending) {
  hipError_t e = hipSuccess;
  for (agpu_event_ref* r : pending)
    if (e == hipSuccess && !r->done) e = hipEventSynchronize(r->ev);
  {
    std::lock_guard<std::mutex> lock(dev->mu);
    for (agpu_event_ref* r : pending) {
      if (e == hipSuccess) r->done = true;
      ref_release_locked(dev, r);
    }
  }
  pending.clear();
  if (e != hipSuccess) {
    agpu_set_error("hipEventSynchronize failed: %s", hipGetErrorString(e));
    return AGPU_ERR_HIP;
  }
  return AGPU_OK;
}

static int small_class(size_t padded) {  // 256 B → 0 … 512 KiB → 11, 1 MiB → 12
  int c = 0;
  while (((size_t)AGPU_SMALL_MIN << c) < padded) c++;
  return c;
}

static bool all_done_locked(const std::vector<agpu_event_ref*>& v) {
  for (agpu_event_ref* r : v)
    if (!ref_done_locked(r)) return false;
  return true;
}

static agpu_status malloc_impl(agpu_device* dev, size_t bytes, int32_t zero_fill, const void* const* neighbours, int n_neighbours,
                               void** out_ptr);
// agpu_malloc_table lays its columns out itself inside ONE fresh hipMalloc block: keep that block out of the arenas (a table
// carved from an arena ran its compare at 0.80–0.83 of the roof where the same layout in its own block runs 0.85–0.89:
// tools/archive/r03_ab.sh, three boxes — inside a 32 GiB arena the physical placement of a 9 GiB span is not what a fresh 9 GiB
// allocation gets)
static thread_local bool t_no_arena = false;
agpu_status agpu_malloc(agpu_device* dev, size_t bytes, int32_t zero_fill, void** out_ptr) {
  return malloc_impl(dev, bytes, zero_fill, nullptr, 0, out_ptr);
}
agpu_status agpu_malloc_like(agpu_device* dev, size_t bytes, int32_t zero_fill, const void* const* neighbours, int32_t n_neighbours,
                             void** out_ptr) {
  AGPU_REQUIRE(n_neighbours >= 0 && (n_neighbours == 0 || neighbours), AGPU_ERR_ARG, "bad neighbour list");
  return malloc_impl(dev, bytes, zero_fill, neighbours, n_neighbours, out_ptr);
}

static agpu_status malloc_impl(agpu_device* dev, size_t bytes, int32_t zero_fill, const void* const* neighbours, int n_neighbours,
                               void** out_ptr) {
  AGPU_REQUIRE(dev && out_ptr, AGPU_ERR_ARG, "null argument");
  AGPU_NOT_POISONED(dev);  // reusing a block may wait for a marker behind the stuck collective
  AGPU_HIP(hipSetDevice(dev->ordinal));
  // pad to 16 B so vector tails of sub-word columns and bitmap words are always addressable
  size_t padded = (bytes + 15) & ~(size_t)15;
  if (padded == 0) padded = 16;
  const size_t requested = padded;  // what zero_fill covers (an arena block's rounding belongs to its neighbour's colour room)
  const bool pool = agpu_mem_pool_enabled();
  const bool large = pool && padded >= AGPU_POOL_MIN_BYTES;
  const bool small = pool && !large;
  void* p = nullptr;
  std::vector<agpu_event_ref*> pending;
  bool sync_all = false;
  if (large) {
    const bool placed = padded >= AGPU_ARENA_MIN_BLOCK && g_pool_arena.load(std::memory_order_relaxed) != 0 && !t_no_arena;
    padded = placed ? arena_padded(padded) : (padded + AGPU_POOL_GRANULE - 1) / AGPU_POOL_GRANULE * AGPU_POOL_GRANULE;
    std::lock_guard<std::mutex> lock(dev->mu);
    // accept up to 12.5 % slack.  Blocks of ≥ 1 GiB come in two kinds that do not stand in for each other (round 6): a request for a
    // PLACED block only takes a cached ARENA block, and a table's own block only a plain one.  Two freed one-column tables (plain
    // 4 GiB hipMallocs, colour 0, physical relation unknown) handed to an ordinary caller as the two inputs of an add ran it at 0.78 of
    // the roof instead of 0.84 — the whole of VERDICT r5 weak #1: bench.py's layout_pool leg drew exactly those two blocks
    // (profiles/r06_alloc_trace.txt) — and a table carved from an arena loses 4–6 points on the compare (above).
    const bool kinds = padded >= AGPU_ARENA_MIN_BLOCK && g_pool_arena.load(std::memory_order_relaxed) != 0;
    auto it = dev->cache.lower_bound(padded);
    while (kinds && it != dev->cache.end() && it->first <= padded + padded / 8 && (dev->arena_block.count(it->second.ptr) != 0) != placed) ++it;
    if (it != dev->cache.end() && it->first <= padded + padded / 8) {
      const bool was_arena = dev->arena_block.count(it->second.ptr) != 0;
      p = arena_recolour_locked(dev, it->second.ptr, placed, neighbours, n_neighbours);
      if (alloc_trace())
        fprintf(stderr, "agpu alloc: %zu B <- cache (%s block of %zu) %p nb=%d\n", requested, was_arena ? "arena" : "plain", it->first, p, n_neighbours);
      pending = std::move(it->second.pending);
      dev->cached_bytes -= it->first;
      dev->block_size[p] = it->first;
      padded = it->first;
      dev->cache.erase(it);
    } else if (placed) {  // carve a fresh one out of an arena (pool placement, above); no room → the plain hipMalloc below
      p = arena_carve_locked(dev, (uint32_t)(padded / AGPU_ARENA_UNIT), neighbours, n_neighbours);
      if (p) dev->block_size[p] = padded;
      if (p && alloc_trace()) fprintf(stderr, "agpu alloc: %zu B <- arena carve %p nb=%d\n", requested, p, n_neighbours);
    }
  } else if (small) {
    const int cls = small_class(padded < AGPU_SMALL_MIN ? AGPU_SMALL_MIN : padded);
    padded = (size_t)AGPU_SMALL_MIN << cls;
    std::lock_guard<std::mutex> lock(dev->mu);
    auto& fl = dev->small_free[cls];
    // oldest first; a block whose markers are still running goes to the back of the queue once — a fresh slab is
    // cheaper than stalling the host behind the GPU (bounded: at most AGPU_SLAB_CAP of slabs)
    if (!fl.empty() && !all_done_locked(fl.front().pending) && fl.size() > 1) {
      fl.push_back(std::move(fl.front()));
      fl.pop_front();
    }
    if (fl.empty() || (!all_done_locked(fl.front().pending) && dev->slab_bytes + AGPU_SLAB_BYTES <= AGPU_SLAB_CAP)) {
      void* base = nullptr;
      if (hipMalloc(&base, AGPU_SLAB_BYTES) == hipSuccess) {
        const uint32_t nb = (uint32_t)(AGPU_SLAB_BYTES / padded);
        dev->slabs[reinterpret_cast<uintptr_t>(base)] = agpu_device::Slab{base, cls, nb, nb};
        dev->slab_bytes += AGPU_SLAB_BYTES;
        for (uint32_t i = nb; i-- > 0;) fl.push_front(agpu_device::CachedBlock{static_cast<char*>(base) + (size_t)i * padded, {}, false});
      } else {
        (void)hipGetLastError();  // fall through: reuse a busy block, or a plain hipMalloc below
      }
    }
    if (!fl.empty()) {
      p = fl.front().ptr;
      pending = std::move(fl.front().pending);
      sync_all = fl.front().sync_all;
      fl.pop_front();
      auto sl = dev->slabs.upper_bound(reinterpret_cast<uintptr_t>(p));
      --sl;
      sl->second.nfree--;
      dev->block_size[p] = padded;
    }
  }
  if (p && sync_all) (void)hipDeviceSynchronize();
  if (p && !pending.empty()) {
    agpu_status ws = wait_pending(dev, pending);
    if (ws != AGPU_OK) return ws;
  }
  if (!p) {
    hipError_t e = dev_malloc(&p, padded);
    if (e == hipErrorOutOfMemory) {  // give the cached blocks back and retry once
      (void)hipGetLastError();
      std::lock_guard<std::mutex> lock(dev->mu);
      device_trim_locked(dev);
      e = dev_malloc(&p, padded);
    }
    if (e != hipSuccess) {
      (void)hipGetLastError();
      agpu_set_error("hipMalloc(%zu) failed: %s", padded, hipGetErrorString(e));
      return AGPU_ERR_HIP;
    }
    if (large) {
      std::lock_guard<std::mutex> lock(dev->mu);
      dev->block_size[p] = padded;
    }
    if (alloc_trace() && padded >= ((size_t)64 << 20)) fprintf(stderr, "agpu alloc: %zu B <- hipMalloc(%zu) %p\n", requested, padded, p);
  }
  if (zero_fill) {
    // hipMemset on device memory runs asynchronously on the NULL stream, and pipelines are non-blocking streams that
    // do not order against it: wait, or a later upload could be overwritten by the zero fill
    hipError_t e = hipMemset(p, 0, requested);
    if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
    if (e != hipSuccess) {
      (void)agpu_free(dev, p);
      agpu_set_error("hipMemset failed: %s", hipGetErrorString(e));
      return AGPU_ERR_HIP;
    }
  }
  *out_ptr = p;
  return AGPU_OK;
}

agpu_status agpu_free(agpu_device* dev, void* ptr) {
  AGPU_REQUIRE(dev, AGPU_ERR_ARG, "null device");
  if (!ptr) return AGPU_OK;
  if (dev->poisoned.load(std::memory_order_acquire)) return AGPU_OK;  // hipFree would wait for the device: leak (host destructors run this)
  AGPU_HIP(hipSetDevice(dev->ordinal));
  {  // a column of an agpu_malloc_table group: the block goes back when its last column does
    void* group_base = nullptr;
    bool member = false;
    {
      std::lock_guard<std::mutex> lock(dev->mu);
      auto tm = dev->table_member.find(ptr);
      if (tm != dev->table_member.end()) {
        member = true;
        agpu_device::TableGroup* g = tm->second;
        dev->table_member.erase(tm);
        if (--g->live == 0) {
          group_base = g->base;
          delete g;
        }
      }
    }
    if (member) return group_base ? agpu_free(dev, group_base) : AGPU_OK;
  }
  {
    std::unique_lock<std::mutex> lock(dev->mu);
    auto it = dev->block_size.find(ptr);
    if (it != dev->block_size.end()) {
      const size_t size = it->second;
      const bool small = size < AGPU_POOL_MIN_BYTES;
      if (small) {  // part of a slab: can only go back to its free list
        dev->block_size.erase(it);
        agpu_device::CachedBlock blk{ptr, {}, false};
        bool ok = true;
        collect_pending_locked(dev, &blk.pending, &ok);
        if (!ok) {  // a stream is being captured: no marker can be recorded now; the next owner waits for the device
          for (agpu_event_ref* r : blk.pending) ref_release_locked(dev, r);
          blk.pending.clear();
          blk.sync_all = true;
        }
        auto sl = dev->slabs.upper_bound(reinterpret_cast<uintptr_t>(ptr));
        --sl;
        sl->second.nfree++;
        dev->small_free[sl->second.cls].push_back(std::move(blk));
        return AGPU_OK;
      }
      dev->block_size.erase(it);
      if (agpu_mem_pool_enabled() && dev->cached_bytes + size <= dev->cache_cap) {
        agpu_device::CachedBlock blk{ptr, {}, false};
        bool ok = true;
        collect_pending_locked(dev, &blk.pending, &ok);
        if (ok) {
          dev->cached_bytes += size;
          dev->cache.emplace(size, std::move(blk));
          return AGPU_OK;
        }
        for (agpu_event_ref* r : blk.pending) ref_release_locked(dev, r);
      }
    }
    if (dev->arena_block.count(ptr)) {  // an arena block the cache has no room for: its units go back once the device is idle
      lock.unlock();
      AGPU_HIP(hipDeviceSynchronize());
      lock.lock();
      (void)arena_release_locked(dev, ptr);
      return AGPU_OK;
    }
  }
  AGPU_HIP(hipFree(ptr));
  return AGPU_OK;
}

// Columns of one table in ONE block, laid out for the memory-channel hash (include/arrow_gpu.h agpu_malloc_table; the
// measurements behind the two constants: tools/probe/hash_bits.py → profiles/r02_hash_bits.json).
//  * address bits {13, 21, 28} feed one bit of the hash, {12, 20, 27} a second, {14, 22, 23, 30, 31, 33}, {11, 19, 26},
//    {10, 25} weaker ones; bits 15–18, 24, 29 and 32 none.  Two streams whose element i sits in the same hash class
//    (distance 2^32: i32 eq at 0.78 of the HBM roof) collide; a distance that flips the first bit runs at 0.85–0.89.
//  * so big columns start a multiple of 512 MiB apart (no hash bit below 2^29 changes between element i of one column
//    and element i of the next, and none varies along the column through carries) plus a colour of 0 / 8 / 4 / 12 KiB by
//    column index: any two of four consecutive columns differ in the first or the second hash bit, adjacent ones in the first.
#define AGPU_TABLE_BIG_COLUMN ((size_t)1 << 30)
#define AGPU_TABLE_BIG_STRIDE ((size_t)512 << 20)
#define AGPU_TABLE_PLACED_MIN ((size_t)32 << 20)
agpu_status agpu_malloc_table(agpu_device* dev, int32_t n_columns, const uint64_t* bytes, int32_t zero_fill, void** out_ptrs) {
  AGPU_REQUIRE(dev && bytes && out_ptrs && n_columns > 0, AGPU_ERR_ARG, "bad argument");
  static const size_t colour[4] = {0, 8192, 4096, 12288};
  std::vector<size_t> off((size_t)n_columns);
  size_t total = 0;
  // three passes: big columns first (so they stay 512 MiB multiples apart), medium ones behind them on 2 MiB granules with
  // the same colours, and SMALL ones (< 32 MiB: validity bitmaps of modest batches, the columns of a 64 Ki-row record batch)
  // packed back to back at 256-byte alignment — placement buys nothing below tens of megabytes, and a 2 MiB granule per
  // 8 KiB bitmap made a file of many small batches cost 256× its size in HBM
  for (int pass = 0; pass < 3; pass++) {
    int pos = 0;
    for (int32_t k = 0; k < n_columns; k++) {
      const size_t b = bytes[k] ? (size_t)bytes[k] : 16;
      const int cls = b >= AGPU_TABLE_BIG_COLUMN ? 0 : b >= AGPU_TABLE_PLACED_MIN ? 1 : 2;
      if (cls != pass) continue;
      if (cls == 2) {
        off[(size_t)k] = total;
        total += (b + 255) / 256 * 256;
        continue;
      }
      const size_t gran = cls == 0 ? AGPU_TABLE_BIG_STRIDE : AGPU_POOL_GRANULE;
      off[(size_t)k] = total + colour[pos++ & 3];
      total += (b + 16384 + gran - 1) / gran * gran;  // 16 KiB: room for the colour
    }
  }
  void* base = nullptr;
  t_no_arena = true;
  agpu_status st = agpu_malloc(dev, total, zero_fill, &base);
  t_no_arena = false;
  if (st != AGPU_OK) return st;
  agpu_device::TableGroup* g = new agpu_device::TableGroup{base, (uint32_t)n_columns};
  {
    std::lock_guard<std::mutex> lock(dev->mu);
    for (int32_t k = 0; k < n_columns; k++) {
      out_ptrs[k] = static_cast<char*>(base) + off[(size_t)k];
      dev->table_member[out_ptrs[k]] = g;
    }
  }
  return AGPU_OK;
}

static void pipeline_mark_drained(agpu_pipeline* p);  // below, with the mailbox
static bool mailbox_enabled(const agpu_pipeline* p);
static agpu_status pipeline_wait_mailbox(agpu_pipeline* p, const void* src_dev, size_t bytes, void* dst_host, void* upload_dst_dev, const void* upload_src_host);
agpu_status agpu_upload(agpu_pipeline* p, void* dst_dev, const void* src_host, size_t bytes) {
  AGPU_BIND(p);
  if (!bytes) return AGPU_OK;
  AGPU_REQUIRE(dst_dev && src_host, AGPU_ERR_ARG, "null pointer");
  // small and medium sources, and anything that lives in the brk heap, go through the library's own page-locked slots
  // (arrow_cdata.hip agpu_internal_host_copy: why); big separate mappings straight from the caller's pageable memory
  // a small array (the reference's tests and examples live at 5–100 elements): through the pipeline's mailbox — the host fills the pinned payload,
  // a one-wave kernel moves it into place and posts; 7 µs instead of the 15 of hipMemcpyAsync + hipStreamSynchronize, complete on return all the same
  // (up to 1 KiB: at 3.6 KB the kernel's reads from host memory cost what the DMA's set-up did — 17.9–20 µs against 17.7, tools/probe/small_arrays.py)
  if (bytes <= 1024 && mailbox_enabled(p)) return pipeline_wait_mailbox(p, nullptr, bytes, nullptr, dst_dev, src_host);
  const agpu_status st = agpu_internal_host_copy(p, dst_dev, const_cast<void*>(src_host), bytes, true);
  if (st == AGPU_OK && !p->capturing) pipeline_mark_drained(p);  // complete on return: the stream has drained (device-level waits skip it)
  return st;
}

// The pipeline's mailbox (round 5, R5.10).  "One small kernel, one scalar back on the host" — the reference's own criterion shape
// (compare_sum.rs: a u32 sum of 1 Mi rows) — costs 15 µs with hipMemcpyAsync D2H + hipStreamSynchronize and 12 µs with the
// synchronize alone; a one-wave kernel that copies ≤ 64 bytes into pinned host memory and then posts a sequence number, with the host
// spinning on that number, is back in 6.7 µs (9.7 µs behind another kernel; tools/probe/latency_probe.hip).  The stream is in order, so
// the posted number also says that everything queued before it is over: agpu_pipeline_sync uses the same kernel with no payload.
// The spin is bounded (tuning "sync_spin", 200 µs by default); after that the blocking hipStreamSynchronize takes over — a long queue
// costs no core, a faulted queue reports its error there.
__global__ __launch_bounds__(AGPU_WAVE) void mailbox_post_kernel(const unsigned char* src, uint32_t bytes, unsigned char* dst, uint64_t* seq_word,
                                                              uint64_t seq) {
  // src → dst: device memory → the slot's payload (a download) or the payload → device memory (an upload); 16-byte vectors when both sides allow
  if (((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15u) == 0) {
    const uint32_t nv = bytes / 16;
    for (uint32_t i = threadIdx.x; i < nv; i += AGPU_WAVE) reinterpret_cast<u32x4*>(dst)[i] = reinterpret_cast<const u32x4*>(src)[i];
    for (uint32_t i = nv * 16 + threadIdx.x; i < bytes; i += AGPU_WAVE) dst[i] = src[i];
  } else {
    for (uint32_t i = threadIdx.x; i < bytes; i += AGPU_WAVE) dst[i] = src[i];
  }
  __threadfence_system();  // one wave: its stores are issued in order, the fence and the release below make them visible to the host first
  if (threadIdx.x == 0) __hip_atomic_store(seq_word, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// the host has just waited for everything on p's stream: remember it on the pipeline (its next sync is free) and on the stream slot (device-level
// waits skip it).  Only the outermost ABI call may say so — a nested one is followed by more work of its caller.
static void pipeline_mark_drained(agpu_pipeline* p) {
  p->dirty = false;
  if (p->scope_depth == 1) p->slot->clean_enq.store(p->slot->enq.load(std::memory_order_relaxed) + 1, std::memory_order_release);  // + 1: this call's own exit
}
static_assert(AGPU_MAILBOX_MAX_BYTES == AGPU_FLAG_SLOT_BYTES - AGPU_MBOX_PAYLOAD, "the public limit is the slot's payload room");
static bool mailbox_enabled(const agpu_pipeline* p) { return p->tune.sync_spin >= 0 && p->flags && !p->capturing; }
// waits for everything queued on the pipeline's stream; bytes ≤ AGPU_MAILBOX_MAX_BYTES travel on the way: device → host (download: the kernel
// fills the payload, the host copies it out after the post) or host → device (upload: the host fills the payload first, the kernel empties it)
static agpu_status pipeline_wait_mailbox(agpu_pipeline* p, const void* src_dev, size_t bytes, void* dst_host) {
  return pipeline_wait_mailbox(p, src_dev, bytes, dst_host, nullptr, nullptr);
}
static agpu_status pipeline_wait_mailbox(agpu_pipeline* p, const void* src_dev, size_t bytes, void* dst_host, void* upload_dst_dev,
                                         const void* upload_src_host) {
  char* slot = reinterpret_cast<char*>(p->flags);
  uint64_t* seq_word = reinterpret_cast<uint64_t*>(slot + AGPU_MBOX_SEQ);
  unsigned char* payload = reinterpret_cast<unsigned char*>(slot + AGPU_MBOX_PAYLOAD);
  const uint64_t seq = ++p->mbox_seq;
  if (upload_dst_dev) {
    memcpy(payload, upload_src_host, bytes);
    hipLaunchKernelGGL(mailbox_post_kernel, dim3(1), dim3(AGPU_WAVE), 0, p->stream, payload, (uint32_t)bytes, static_cast<unsigned char*>(upload_dst_dev),
                       seq_word, seq);
  } else {
    hipLaunchKernelGGL(mailbox_post_kernel, dim3(1), dim3(AGPU_WAVE), 0, p->stream, static_cast<const unsigned char*>(src_dev), (uint32_t)bytes, payload,
                       seq_word, seq);
  }
  AGPU_LAUNCH_CHECK();
  const int64_t budget_us = p->tune.sync_spin > 0 ? p->tune.sync_spin : 200;
  const auto t0 = std::chrono::steady_clock::now();
  bool arrived = false;
  for (;;) {
    for (int i = 0; i < 128 && !arrived; i++) {
      arrived = __atomic_load_n(seq_word, __ATOMIC_ACQUIRE) == seq;
      if (!arrived) __builtin_ia32_pause();
    }
    if (arrived || std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count() >= budget_us) break;
  }
  if (!arrived) {
    AGPU_HIP(hipStreamSynchronize(p->stream));
    if (__atomic_load_n(seq_word, __ATOMIC_ACQUIRE) != seq) {
      agpu_set_error("the pipeline's mailbox was not posted although its stream is idle");
      return AGPU_ERR_HIP;
    }
  }
  if (bytes && !upload_dst_dev) memcpy(dst_host, payload, bytes);
  pipeline_mark_drained(p);
  return AGPU_OK;
}

// Device-level wait (agpu_device_sync, agpu_device_download): everything THIS LIBRARY has queued on the device — every pipeline, pooled stream
// and wrapped stream.  hipDeviceSynchronize costs 11 µs behind one small kernel; the usual situation of the reference-style host API is ONE
// stream with work outstanding (the immediate ops of one thread draw the same pooled stream again and again), and then the mailbox kernel
// posted on THAT stream is back in 6–9 µs, with up to 64 bytes of payload on the way (`values()` of a reduction's result: one wait, not two).
// No stream outstanding: nothing to wait for (the payload, if any, travels on an idle owned stream).  Two or more: hipDeviceSynchronize as
// before — unless some pipeline is in graph capture: hipDeviceSynchronize would invalidate that capture (from any thread), so the outstanding
// streams are then waited for one by one through the mailbox and the capturing stream is left alone (what it had queued before its capture
// began is not waited for).  sync_spin < 0: hipDeviceSynchronize always.
static uint32_t* flag_get_locked(agpu_device* dev);  // below, with the pipelines
// in graph capture — by one of this library's pipelines, or (a wrapped stream) by whoever owns it, e.g. torch.cuda.graph
static bool slot_capturing(agpu_stream_slot* s) {
  if (s->capturing.load(std::memory_order_acquire)) return true;
  if (s->owned) return false;
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(s->stream, &st) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  return st != hipStreamCaptureStatusNone;
}
static bool slot_listed_locked(agpu_device* dev, agpu_stream_slot* s) {
  for (agpu_stream_slot* x : dev->slots)
    if (x == s) return true;
  return false;
}
// The device's mailbox has FOUR sequence words (slot bytes 128 … 159): up to three posts can be out at once — one per outstanding stream, waited
// for together — plus the word the payload's post uses.
// device_post: queues the post behind everything on `s` (payload optional: src → the slot's payload); *posted = false when the slot has left the
// device or entered a capture meanwhile (it then counts as done).  The launch happens under dev->mu: a stream leaves dev->slots, and a
// capture begins, under that mutex.
static agpu_status device_post(agpu_device* dev, agpu_stream_slot* s, int word, const void* src_dev, size_t bytes, uint64_t* seq_out, bool* posted) {
  char* mb = reinterpret_cast<char*>(dev->mbox);
  uint64_t* seq_word = reinterpret_cast<uint64_t*>(mb + AGPU_MBOX_SEQ) + word;
  std::lock_guard<std::mutex> lock(dev->mu);
  *posted = false;
  if (!slot_listed_locked(dev, s) || slot_capturing(s)) return AGPU_OK;
  *seq_out = ++dev->mbox_seq;
  hipLaunchKernelGGL(mailbox_post_kernel, dim3(1), dim3(AGPU_WAVE), 0, s->stream, static_cast<const unsigned char*>(src_dev), (uint32_t)bytes,
                     reinterpret_cast<unsigned char*>(mb + AGPU_MBOX_PAYLOAD), seq_word, *seq_out);
  AGPU_LAUNCH_CHECK();
  *posted = true;
  return AGPU_OK;
}
// device_await: spin, then sleep-poll, until word `word` shows `seq`
// *arrived_out = false with AGPU_OK: the post can no longer be waited for here (its slot left the device, or its stream entered a graph capture
// — asking a capturing stream anything invalidates the capture): the caller finishes with a runtime wait / copy instead
static agpu_status device_await(agpu_device* dev, agpu_stream_slot* s, int word, uint64_t seq, int64_t spin, bool* arrived_out) {
  uint64_t* seq_word = reinterpret_cast<uint64_t*>(reinterpret_cast<char*>(dev->mbox) + AGPU_MBOX_SEQ) + word;
  const int64_t budget_us = spin > 0 ? spin : 200;
  const auto t0 = std::chrono::steady_clock::now();
  bool arrived = false;
  for (;;) {
    for (int i = 0; i < 128 && !arrived; i++) {
      arrived = __atomic_load_n(seq_word, __ATOMIC_ACQUIRE) == seq;
      if (!arrived) __builtin_ia32_pause();
    }
    if (arrived || std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count() >= budget_us) break;
  }
  // a long queue: sleep-poll (no core burnt, no runtime wait that would hold dev->mu or touch a capture); every 5 ms ask the stream whether it
  // is still alive — a faulted queue never posts
  for (int polls = 0; !arrived; polls++) {
    std::this_thread::sleep_for(std::chrono::microseconds(polls < 20 ? 50 : 200));
    arrived = __atomic_load_n(seq_word, __ATOMIC_ACQUIRE) == seq;
    if (arrived || polls % 25 != 24) continue;
    std::lock_guard<std::mutex> lock(dev->mu);
    if (!slot_listed_locked(dev, s) || slot_capturing(s)) break;
    const hipError_t q = hipStreamQuery(s->stream);
    if (q == hipErrorNotReady) continue;
    arrived = __atomic_load_n(seq_word, __ATOMIC_ACQUIRE) == seq;
    if (q != hipSuccess || !arrived) {
      agpu_set_error("device wait: the stream stopped without posting the mailbox: %s", hipGetErrorString(q));
      return AGPU_ERR_HIP;
    }
  }
  *arrived_out = arrived;
  return AGPU_OK;
}

static agpu_status device_wait_all(agpu_device* dev, const void* src_dev, size_t bytes, void* dst_host) {
  struct Seen { agpu_stream_slot* s; uint64_t enq; };
  std::vector<Seen> seen;
  std::vector<agpu_stream_slot*> targets;
  const int64_t spin = g_tune_default[tune_index("sync_spin")].load(std::memory_order_relaxed);  // process-wide: a device has no tuning of its own
  bool any_capturing = false, have_mbox = false;
  {
    std::lock_guard<std::mutex> lock(dev->mu);
    if (spin >= 0 && !dev->mbox) {
      dev->mbox = flag_get_locked(dev);
      if (dev->mbox) dev->mbox_seq = *reinterpret_cast<volatile uint64_t*>(reinterpret_cast<char*>(dev->mbox) + AGPU_MBOX_SEQ);
    }
    have_mbox = spin >= 0 && dev->mbox;
    agpu_stream_slot* idle_owned = nullptr;
    for (agpu_stream_slot* s : dev->slots) {
      if (slot_capturing(s)) {
        any_capturing = true;
        continue;
      }
      const uint64_t e = s->enq.load(std::memory_order_acquire);
      seen.push_back(Seen{s, e});
      // drained: no call since the last completed wait (a wrapped stream also carries work this library never saw: never "drained")
      if (s->owned && !s->exposed.load(std::memory_order_acquire) && (e & ~1ull) == s->clean_enq.load(std::memory_order_acquire) && !(e & 1)) {
        if (!idle_owned) idle_owned = s;
        continue;
      }
      targets.push_back(s);
    }
    if (targets.empty() && spin >= 0) {  // (sync_spin < 0 = "always the runtime's device-wide wait": below, whatever the counters say)
      if (!bytes) return AGPU_OK;
      if (idle_owned && have_mbox) targets.push_back(idle_owned);
    }
  }
  // up to three streams: posts, out together and waited for together (tools/probe/two_streams.py).  More: the runtime's device-wide wait.
  const bool by_mailbox = have_mbox && !targets.empty() && (targets.size() <= 3 || any_capturing);
  if (by_mailbox) {
    std::lock_guard<std::mutex> one_wait(dev->mbox_mu);  // the device has ONE mailbox: one posted wait at a time (runtime waits below need no turn)
    // every outstanding stream but the last gets its post at once and they are waited for together; the LAST stream's post carries the bytes and
    // goes out only when the others have drained — whichever stream produced the bytes, they are final when that kernel copies them
    const size_t nt = targets.size();
    uint64_t seqs[3] = {0, 0, 0};
    bool posted[3] = {false, false, false};
    size_t early = bytes ? nt - 1 : nt;  // no payload: all of them at once
    if (early > 3) early = 3;
    for (size_t k = 0; k < early; k++) {
      const agpu_status st = device_post(dev, targets[k], (int)k, nullptr, 0, &seqs[k], &posted[k]);
      if (st != AGPU_OK) return st;
    }
    bool fallback = false;  // a post that could not be waited for: the runtime's device-wide wait finishes the job (no capture is open then, or
                            // the stream that opened one is simply skipped by it — its work is not ours to wait for mid-capture)
    for (size_t k = 0; k < early; k++)
      if (posted[k]) {
        bool got = false;
        const agpu_status st = device_await(dev, targets[k], (int)k, seqs[k], spin, &got);
        if (st != AGPU_OK) return st;
        if (!got) fallback = true;
      }
    for (size_t k = early; k < nt; k++) {  // the payload's post (and, with a capture open, streams beyond the third)
      const bool last = k + 1 == nt;
      uint64_t seq = 0;
      bool ok = false;
      agpu_status st = device_post(dev, targets[k], 3, last ? src_dev : nullptr, last ? bytes : 0, &seq, &ok);
      if (st != AGPU_OK) return st;
      bool got = false;
      if (ok) {
        st = device_await(dev, targets[k], 3, seq, spin, &got);
        if (st != AGPU_OK) return st;
        if (!got) fallback = true;
      }
      if (ok && got) {
        if (last && bytes) memcpy(dst_host, reinterpret_cast<char*>(dev->mbox) + AGPU_MBOX_PAYLOAD, bytes);
      } else if (last && bytes) {  // never posted, or the post never arrived: the payload is NOT in the mailbox — a blocking copy reads it
        if (!any_capturing) AGPU_HIP(hipDeviceSynchronize());  // whatever produced the bytes is through before they are read
        AGPU_HIP(hipMemcpy(dst_host, src_dev, bytes, hipMemcpyDeviceToHost));  // (rare: the stream went away between the scan and now)
      }
    }
    if (fallback && !any_capturing) AGPU_HIP(hipDeviceSynchronize());
  } else {
    AGPU_HIP(hipDeviceSynchronize());
    if (bytes) AGPU_HIP(hipMemcpy(dst_host, src_dev, bytes, hipMemcpyDeviceToHost));
  }
  // what was queued when this wait began is over (an odd count: the call that was in progress may still add work — clean up to the call before it)
  std::lock_guard<std::mutex> lock(dev->mu);
  for (agpu_stream_slot* s : dev->slots)  // (a slot may have left the device meanwhile: only those still listed are touched)
    for (const Seen& x : seen) {
      if (x.s != s) continue;
      const uint64_t upto = x.enq & ~1ull;
      uint64_t cur = s->clean_enq.load(std::memory_order_relaxed);
      while (cur < upto && !s->clean_enq.compare_exchange_weak(cur, upto, std::memory_order_release)) {}
    }
  return AGPU_OK;
}

agpu_status agpu_device_download(agpu_device* dev, void* dst_host, const void* src_dev, size_t bytes) {
  AGPU_REQUIRE(dev, AGPU_ERR_ARG, "null device");
  AGPU_REQUIRE(bytes <= AGPU_MAILBOX_MAX_BYTES, AGPU_ERR_ARG, "at most AGPU_MAILBOX_MAX_BYTES; bigger reads go through agpu_device_sync + agpu_download");
  AGPU_REQUIRE(!bytes || (dst_host && src_dev), AGPU_ERR_ARG, "null pointer");
  AGPU_NOT_POISONED(dev);
  AGPU_HIP(hipSetDevice(dev->ordinal));
  return device_wait_all(dev, src_dev, bytes, dst_host);
}

agpu_status agpu_download(agpu_pipeline* p, void* dst_host, const void* src_dev, size_t bytes) {
  const bool was_dirty = p && p->dirty;
  AGPU_BIND(p);
  if (!bytes) {
    if (was_dirty && mailbox_enabled(p)) return pipeline_wait_mailbox(p, nullptr, 0, nullptr);
    AGPU_HIP(hipStreamSynchronize(p->stream));
    pipeline_mark_drained(p);
    return AGPU_OK;
  }
  AGPU_REQUIRE(dst_host && src_dev, AGPU_ERR_ARG, "null pointer");
  if (bytes <= AGPU_MAILBOX_MAX_BYTES && mailbox_enabled(p)) return pipeline_wait_mailbox(p, src_dev, bytes, dst_host);  // a scalar, a small array
  const agpu_status st = agpu_internal_host_copy(p, const_cast<void*>(src_dev), dst_host, bytes, false);
  if (st == AGPU_OK && !p->capturing) pipeline_mark_drained(p);  // complete on return: the stream has drained
  return st;
}

agpu_status agpu_host_alloc(agpu_device* dev, size_t bytes, void** out_host_ptr) {
  AGPU_REQUIRE(dev && out_host_ptr, AGPU_ERR_ARG, "null argument");
  AGPU_HIP(hipSetDevice(dev->ordinal));
  void* h = nullptr;
  AGPU_HIP(hipHostMalloc(&h, bytes ? bytes : 1, hipHostMallocDefault));
  *out_host_ptr = h;
  return AGPU_OK;
}

agpu_status agpu_host_free(agpu_device* dev, void* host_ptr) {
  AGPU_REQUIRE(dev, AGPU_ERR_ARG, "null device");
  if (!host_ptr) return AGPU_OK;
  AGPU_HIP(hipSetDevice(dev->ordinal));
  AGPU_HIP(hipHostFree(host_ptr));
  return AGPU_OK;
}

agpu_status agpu_upload_async(agpu_pipeline* p, void* dst_dev, const void* src_pinned, size_t bytes) {
  AGPU_BIND(p);
  if (!bytes) return AGPU_OK;
  AGPU_REQUIRE(dst_dev && src_pinned, AGPU_ERR_ARG, "null pointer");
  AGPU_HIP(hipMemcpyAsync(dst_dev, src_pinned, bytes, hipMemcpyHostToDevice, p->stream));
  p->copy_tail = true;
  return AGPU_OK;
}

agpu_status agpu_download_async(agpu_pipeline* p, void* dst_pinned, const void* src_dev, size_t bytes) {
  AGPU_BIND(p);
  if (!bytes) return AGPU_OK;
  AGPU_REQUIRE(dst_pinned && src_dev, AGPU_ERR_ARG, "null pointer");
  AGPU_HIP(hipMemcpyAsync(dst_pinned, src_dev, bytes, hipMemcpyDeviceToHost, p->stream));
  p->copy_tail = true;
  return AGPU_OK;
}

// clone_buffer / clone_array [ref: gpu_device.rs:212-230, compute_pipeline.rs:275-299].  hipMemcpyAsync D2D runs the
// runtime's blit kernel: 4.6 TB/s of traffic on a 4 GB column (0.58 of the roof, tools/kernel_table.py "context" row)
// where a plain stream of nontemporal 16-byte loads and stores in one-wave blocks — the element-wise kernels' shape —
// moves the same bytes at 6.7.  Big, 16-byte-aligned, non-overlapping copies take that kernel; everything else (small
// copies: the launch dominates either way) stays with the runtime.
__global__ __launch_bounds__(AGPU_WAVE) void copy_kernel(const u32x4* src, u32x4* dst, uint64_t nvec, uint64_t nblocks, uint64_t half) {
  for (uint64_t b = blockIdx.x; b < nblocks; b += gridDim.x) {
    const uint64_t i = two_streams(b, half) * AGPU_WAVE + threadIdx.x;  // big copies: two lock-step streams (common.hpp)
    if (i < nvec) __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
  }
}

agpu_status agpu_copy(agpu_pipeline* p, void* dst_dev, const void* src_dev, size_t bytes) {
  AGPU_BIND(p);
  if (!bytes) return AGPU_OK;
  AGPU_REQUIRE(dst_dev && src_dev, AGPU_ERR_ARG, "null pointer");
  const uintptr_t d = reinterpret_cast<uintptr_t>(dst_dev), s = reinterpret_cast<uintptr_t>(src_dev);
  const bool disjoint = d + bytes <= s || s + bytes <= d;
  if (bytes >= ((size_t)1 << 20) && ((d | s) & 15u) == 0 && disjoint) {
    const uint64_t nvec = bytes / 16;
    const uint64_t blocks = (nvec + AGPU_WAVE - 1) / AGPU_WAVE;
    hipLaunchKernelGGL(copy_kernel, dim3((unsigned)(blocks < 0x3FFFFFFFull ? blocks : 0x3FFFFFFFull)), dim3(AGPU_WAVE), 0, p->stream,
                       static_cast<const u32x4*>(src_dev), static_cast<u32x4*>(dst_dev), nvec, blocks, two_streams_half(p, blocks, 1024));
    AGPU_LAUNCH_CHECK();
    const size_t done = (size_t)nvec * 16;
    if (done < bytes)
      AGPU_HIP(hipMemcpyAsync(static_cast<char*>(dst_dev) + done, static_cast<const char*>(src_dev) + done, bytes - done,
                              hipMemcpyDeviceToDevice, p->stream));
    return AGPU_OK;
  }
  AGPU_HIP(hipMemcpyAsync(dst_dev, src_dev, bytes, hipMemcpyDeviceToDevice, p->stream));
  return AGPU_OK;
}

// the same for fills: hipMemsetAsync 6.3 TB/s on 4 GB, the broadcast kernel (elementwise.hip fill_kernel) 6.9
// (tools/probe/memset_check.py; a one-wave-block variant with nontemporal stores measured no better than the runtime)
agpu_status agpu_memset(agpu_pipeline* p, void* dst_dev, int32_t byte_value, size_t bytes) {
  AGPU_BIND(p);
  if (!bytes) return AGPU_OK;
  AGPU_REQUIRE(dst_dev, AGPU_ERR_ARG, "null pointer");
  if (bytes >= ((size_t)1 << 20) && (reinterpret_cast<uintptr_t>(dst_dev) & 15u) == 0)
    return agpu_internal_fill_bytes(p, dst_dev, ((uint32_t)byte_value & 255u) * 0x01010101u, bytes);
  AGPU_HIP(hipMemsetAsync(dst_dev, byte_value, bytes, p->stream));
  return AGPU_OK;
}

// ---------------------------------------------------------------- pipeline
// Sticky error words: each LIVE pipeline owns one; a destroyed pipeline's word is handed to the next owner only after
// the stream has passed every kernel the old owner queued (its last marker completed), so a late take/put of the
// previous owner can neither raise an error in the new owner's sync nor be wiped by the new owner's reset.
static uint32_t* flag_get_locked(agpu_device* dev) {
  for (size_t i = 0; i < dev->flag_retired.size();) {
    if (ref_done_locked(dev->flag_retired[i].after)) {
      ref_release_locked(dev, dev->flag_retired[i].after);
      dev->flag_free.push_back(dev->flag_retired[i].word);
      dev->flag_retired[i] = dev->flag_retired.back();
      dev->flag_retired.pop_back();
    } else {
      i++;
    }
  }
  if (dev->flag_free.empty()) {
    void* slab = nullptr;  // pinned + device-visible under unified addressing; 16 slots of 4 KiB: error word, mailbox sequence word, mailbox payload
    if (hipHostMalloc(&slab, 16 * AGPU_FLAG_SLOT_BYTES, hipHostMallocDefault) != hipSuccess) {
      (void)hipGetLastError();
      return nullptr;
    }
    memset(slab, 0, 16 * AGPU_FLAG_SLOT_BYTES);
    dev->flag_slabs.push_back(slab);
    for (int i = 15; i >= 0; i--) dev->flag_free.push_back(reinterpret_cast<uint32_t*>(static_cast<char*>(slab) + AGPU_FLAG_SLOT_BYTES * i));
  }
  uint32_t* w = dev->flag_free.back();
  dev->flag_free.pop_back();
  *w = 0;
  return w;
}

static agpu_pipeline* pipeline_new(agpu_device* dev, agpu_stream_slot* slot, bool owns, uint32_t* flags) {
  agpu_pipeline* p = new agpu_pipeline();
  p->dev = dev;
  p->slot = slot;
  p->stream = slot->stream;
  p->owns_stream = owns;
  p->capturing = false;
  p->seen_gen = 0;  // first call orders the stream behind everything other pipelines have finished so far
  p->tune = agpu_tuning_defaults();
  p->flags = flags;
  // the slot's sequence word keeps counting across owners (a slot is only handed out again once its previous owner's kernels are over)
  p->mbox_seq = *reinterpret_cast<volatile uint64_t*>(reinterpret_cast<char*>(flags) + AGPU_MBOX_SEQ);
  p->dirty = true;
  p->profile = env_profile();
  p->scope_depth = 0;
  p->t0 = p->t1 = nullptr;
  p->t_valid = false;
  p->last_name = "";
  return p;
}

agpu_status agpu_pipeline_create(agpu_device* dev, agpu_pipeline** out_pipeline) {
  AGPU_REQUIRE(dev && out_pipeline, AGPU_ERR_ARG, "null argument");
  AGPU_HIP(hipSetDevice(dev->ordinal));
  agpu_stream_slot* slot = nullptr;
  uint32_t* flags = nullptr;
  {
    std::lock_guard<std::mutex> lock(dev->mu);
    if (!dev->idle.empty()) {  // work still queued on a recycled stream simply runs first: same ordering
      slot = dev->idle.back();
      dev->idle.pop_back();
    }
    flags = flag_get_locked(dev);
  }
  if (!flags) {
    agpu_set_error("hipHostMalloc of the pipeline error words failed");
    return AGPU_ERR_HIP;
  }
  if (!slot) {
    hipStream_t s = nullptr;
    AGPU_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    slot = new agpu_stream_slot();
    slot->stream = s;
    slot->owned = true;
    std::lock_guard<std::mutex> lock(dev->mu);
    dev->slots.push_back(slot);
  }
  *out_pipeline = pipeline_new(dev, slot, true, flags);
  return AGPU_OK;
}

agpu_status agpu_pipeline_wrap_stream(agpu_device* dev, void* hip_stream, agpu_pipeline** out_pipeline) {
  AGPU_REQUIRE(dev && out_pipeline, AGPU_ERR_ARG, "null argument");
  AGPU_HIP(hipSetDevice(dev->ordinal));
  agpu_stream_slot* slot = new agpu_stream_slot();
  slot->stream = reinterpret_cast<hipStream_t>(hip_stream);
  slot->owned = false;
  uint32_t* flags = nullptr;
  {
    std::lock_guard<std::mutex> lock(dev->mu);
    dev->slots.push_back(slot);
    flags = flag_get_locked(dev);
  }
  if (!flags) {
    agpu_set_error("hipHostMalloc of the pipeline error words failed");
    return AGPU_ERR_HIP;
  }
  *out_pipeline = pipeline_new(dev, slot, false, flags);
  return AGPU_OK;
}

// Cross-pipeline ordering.  The reference has ONE wgpu queue: everything a later submit records runs after everything
// an earlier `finish()` submitted [ref: compute_pipeline.rs:259-263 queue.submit].  Pipelines here are independent HIP
// streams, so `finish` PUBLISHES the stream's position (a marker event + a device-wide generation number) and every
// pipeline, before its next call enqueues anything, makes its stream wait for the positions other pipelines have
// published since it last looked (agpu_scope_enter).  Cost when nothing was published: one atomic load per call.
// Contract: a pipeline may consume buffers another pipeline produced once that pipeline has finished (or was
// destroyed) — exactly the reference's rule that results are defined after `finish()`.
static void publish_finish_locked(agpu_pipeline* p) {
  agpu_device* dev = p->dev;
  agpu_stream_slot* s = p->slot;
  bool ok = true;
  agpu_event_ref* m = slot_mark_locked(dev, s, &ok);
  if (!ok || !m || m == s->finish_ev) return;  // capturing, nothing ever ran, or nothing new since the last finish
  m->refs++;
  ref_release_locked(dev, s->finish_ev);
  s->finish_ev = m;
  s->finish_gen = dev->finish_gen.load(std::memory_order_relaxed) + 1;
  dev->finish_gen.store(s->finish_gen, std::memory_order_release);
}

agpu_status agpu_pipeline_finish(agpu_pipeline* p) {
  AGPU_REQUIRE(p && p->dev, AGPU_ERR_ARG, "null pipeline");
  AGPU_HIP(hipSetDevice(p->dev->ordinal));
  if (p->capturing) return AGPU_OK;
  // everything recorded so far is already enqueued in order; like the reference, do not wait
  std::lock_guard<std::mutex> lock(p->dev->mu);
  publish_finish_locked(p);
  return AGPU_OK;
}

agpu_status agpu_pipeline_sync(agpu_pipeline* p) {
  const bool was_dirty = p && p->dirty, copy_tail = p && p->copy_tail;
  AGPU_BIND(p);
  if (was_dirty && !copy_tail && mailbox_enabled(p)) {
    const agpu_status st = pipeline_wait_mailbox(p, nullptr, 0, nullptr);
    if (st != AGPU_OK) return st;
  } else {
    AGPU_HIP(hipStreamSynchronize(p->stream));
    pipeline_mark_drained(p);
  }
  if (p->flags && *p->flags) {  // sticky kernel-side errors surface here, once
    const uint32_t f = *p->flags;
    *p->flags = 0;
    if (f & AGPU_FLAG_INDEX_RANGE) {
      agpu_set_error("take/put: an index was out of range (the element was skipped / read as 0)");
      return AGPU_ERR_SHAPE;
    }
  }
  return AGPU_OK;
}

agpu_status agpu_pipeline_destroy(agpu_pipeline* p) {
  if (!p) return AGPU_OK;
  agpu_device* dev = p->dev;
  (void)hipSetDevice(dev->ordinal);
  if (dev->poisoned.load(std::memory_order_acquire)) {  // its stream may hold the stuck collective: no wait, no hipStreamDestroy
    delete p;
    return AGPU_OK;
  }
  if (p->t0) (void)hipEventDestroy(p->t0);
  if (p->t1) (void)hipEventDestroy(p->t1);
  agpu_stream_slot* s = p->slot;
  if (p->capturing) {  // abandon the capture so the stream is usable again
    hipGraph_t g = nullptr;
    (void)hipStreamEndCapture(p->stream, &g);
    if (g) (void)hipGraphDestroy(g);
    (void)hipGetLastError();
    p->capturing = false;
    p->slot->capturing.store(false, std::memory_order_release);
  }
  if (p->owns_stream && agpu_mem_pool_enabled()) {
    // back to the pool WITHOUT waiting: queued work keeps running, the next owner's launches are ordered behind it.
    // Destroying a pipeline is a submit point like finish(): other pipelines order themselves behind its work.
    std::lock_guard<std::mutex> lock(dev->mu);
    publish_finish_locked(p);
    bool ok = true;
    agpu_event_ref* m = slot_mark_locked(dev, s, &ok);
    if (m && !ref_done_locked(m)) {
      m->refs++;
      dev->flag_retired.push_back(agpu_device::RetiredFlag{p->flags, m});
    } else {
      dev->flag_free.push_back(p->flags);
    }
    s->exposed.store(false, std::memory_order_release);  // the handle dies with the pipeline; what was queued through it is behind the marker above
    dev->idle.push_back(s);
    delete p;
    return AGPU_OK;
  }
  (void)hipStreamSynchronize(p->stream);
  {
    std::lock_guard<std::mutex> lock(dev->mu);
    dev->flag_free.push_back(p->flags);
    if (s->scratch) scratch_release_locked(s->scratch);
    s->scratch = nullptr;
    ref_release_locked(dev, s->mark);
    ref_release_locked(dev, s->finish_ev);
    for (size_t i = 0; i < dev->slots.size(); i++)
      if (dev->slots[i] == s) {
        dev->slots.erase(dev->slots.begin() + (long)i);
        break;
      }
  }
  if (p->owns_stream) (void)hipStreamDestroy(p->stream);
  delete s;
  delete p;
  return AGPU_OK;
}

agpu_status agpu_pipeline_device(agpu_pipeline* p, agpu_device** out_device) {
  AGPU_REQUIRE(p && out_device, AGPU_ERR_ARG, "null argument");
  *out_device = p->dev;
  return AGPU_OK;
}

agpu_status agpu_pipeline_stream(agpu_pipeline* p, void** out_hip_stream) {
  AGPU_REQUIRE(p && out_hip_stream, AGPU_ERR_ARG, "null argument");
  *out_hip_stream = reinterpret_cast<void*>(p->stream);
  if (p->slot) p->slot->exposed.store(true, std::memory_order_release);  // the caller may queue work of its own from now on
  return AGPU_OK;
}

// Explicit cross-pipeline dependency for hosts that overlap pipelines on purpose (double-buffered staging): work
// enqueued on `p` after this call runs after everything enqueued on `other` before it.  Does not block the host.
agpu_status agpu_pipeline_wait_pipeline(agpu_pipeline* p, agpu_pipeline* other) {
  AGPU_BIND(p);
  AGPU_REQUIRE(other && other->dev == p->dev, AGPU_ERR_ARG, "pipelines must share a device");
  if (other->slot == p->slot) return AGPU_OK;
  AGPU_REQUIRE(!p->capturing && !other->capturing, AGPU_ERR_ARG, "not during graph capture");
  std::lock_guard<std::mutex> lock(p->dev->mu);
  bool ok = true;
  agpu_event_ref* m = slot_mark_locked(p->dev, other->slot, &ok);
  AGPU_REQUIRE(ok, AGPU_ERR_HIP, "could not record a marker on the other pipeline's stream");
  if (m && !ref_done_locked(m)) AGPU_HIP(hipStreamWaitEvent(p->stream, m->ev, 0));
  return AGPU_OK;
}

// ---------------------------------------------------------------- graphs
agpu_status agpu_pipeline_begin_capture(agpu_pipeline* p) {
  {
    AGPU_BIND(p);  // orders the stream behind other pipelines' finished work BEFORE the capture starts
  }
  AGPU_REQUIRE(!p->capturing, AGPU_ERR_ARG, "already capturing");
  hipError_t ce;
  {  // under dev->mu: a device-level wait on another thread checks the flag and launches its post kernel under the same mutex — never into a capture
    std::lock_guard<std::mutex> lock(p->dev->mu);
    p->slot->capturing.store(true, std::memory_order_release);
    ce = hipStreamBeginCapture(p->stream, hipStreamCaptureModeThreadLocal);
    if (ce != hipSuccess) p->slot->capturing.store(false, std::memory_order_release);
  }
  if (ce != hipSuccess) {
    agpu_set_error("hipStreamBeginCapture failed: %s", hipGetErrorString(ce));
    return AGPU_ERR_HIP;
  }
  p->capturing = true;
  return AGPU_OK;
}

agpu_status agpu_pipeline_end_capture(agpu_pipeline* p, agpu_graph** out_graph) {
  AGPU_REQUIRE(p && p->dev, AGPU_ERR_ARG, "null pipeline");
  AGPU_HIP(hipSetDevice(p->dev->ordinal));
  AGPU_REQUIRE(out_graph, AGPU_ERR_ARG, "null out_graph");
  AGPU_REQUIRE(p->capturing, AGPU_ERR_ARG, "not capturing");
  p->capturing = false;
  hipGraph_t g = nullptr;
  const hipError_t ee = hipStreamEndCapture(p->stream, &g);
  p->slot->capturing.store(false, std::memory_order_release);
  AGPU_HIP(ee);
  hipGraphExec_t ex = nullptr;
  hipError_t e = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
  if (e != hipSuccess) {
    (void)hipGraphDestroy(g);
    agpu_set_error("hipGraphInstantiate failed: %s", hipGetErrorString(e));
    return AGPU_ERR_HIP;
  }
  agpu_graph* gr = new agpu_graph();
  gr->dev = p->dev;
  gr->graph = g;
  gr->exec = ex;
  {
    // kernel nodes of reductions carry the scratch POINTER: keep that block alive as long as the graph
    std::lock_guard<std::mutex> lock(p->dev->mu);
    gr->scratch = p->slot->scratch;
    if (gr->scratch) gr->scratch->refs++;
  }
  *out_graph = gr;
  return AGPU_OK;
}

agpu_status agpu_graph_launch(agpu_graph* g, agpu_pipeline* p) {
  AGPU_BIND(p);
  AGPU_REQUIRE(g, AGPU_ERR_ARG, "null graph");
  AGPU_HIP(hipGraphLaunch(g->exec, p->stream));
  return AGPU_OK;
}

agpu_status agpu_graph_destroy(agpu_graph* g) {
  if (!g) return AGPU_OK;
  (void)hipSetDevice(g->dev->ordinal);
  (void)hipGraphExecDestroy(g->exec);
  (void)hipGraphDestroy(g->graph);
  if (g->scratch) {
    std::lock_guard<std::mutex> lock(g->dev->mu);
    if (g->scratch->refs == 1) (void)hipDeviceSynchronize();  // last owner: replays may still be running
    scratch_release_locked(g->scratch);
  }
  delete g;
  return AGPU_OK;
}

// ---------------------------------------------------------------- events
agpu_status agpu_event_create(agpu_device* dev, agpu_event** out_event) {
  AGPU_REQUIRE(dev && out_event, AGPU_ERR_ARG, "null argument");
  AGPU_HIP(hipSetDevice(dev->ordinal));
  hipEvent_t ev;
  AGPU_HIP(hipEventCreate(&ev));
  agpu_event* e = new agpu_event();
  e->dev = dev;
  e->ev = ev;
  *out_event = e;
  return AGPU_OK;
}

agpu_status agpu_event_record(agpu_event* e, agpu_pipeline* p) {
  AGPU_REQUIRE(p && p->dev && e, AGPU_ERR_ARG, "null argument");
  AGPU_HIP(hipSetDevice(p->dev->ordinal));
  AGPU_HIP(hipEventRecord(e->ev, p->stream));
  return AGPU_OK;
}

agpu_status agpu_event_elapsed_ms(agpu_event* start, agpu_event* stop, float* out_ms) {
  AGPU_REQUIRE(start && stop && out_ms, AGPU_ERR_ARG, "null argument");
  AGPU_HIP(hipSetDevice(stop->dev->ordinal));
  AGPU_HIP(hipEventSynchronize(stop->ev));
  AGPU_HIP(hipEventElapsedTime(out_ms, start->ev, stop->ev));
  return AGPU_OK;
}

agpu_status agpu_event_destroy(agpu_event* e) {
  if (!e) return AGPU_OK;
  (void)hipSetDevice(e->dev->ordinal);
  (void)hipEventDestroy(e->ev);
  delete e;
  return AGPU_OK;
}

// ---------------------------------------------------------------- per-launch profiling
// [ref: GpuDevice::compute_pass insert_debug_marker(entry_point) gpu_device.rs:132; CmpQuery compute_query.rs:7-89 —
//  a 2-slot timestamp query per pass + wait_for_results() logging "Time taken for compute pass"]
agpu_status agpu_pipeline_enable_timing(agpu_pipeline* p, int32_t profile_bits) {
  AGPU_REQUIRE(p && p->dev, AGPU_ERR_ARG, "null pipeline");
  AGPU_REQUIRE(p->scope_depth == 0, AGPU_ERR_ARG, "not from inside a call");
  uint32_t b = (uint32_t)profile_bits & (AGPU_PROF_ROCTX | AGPU_PROF_TIMING | AGPU_PROF_LOG);
  if (b & AGPU_PROF_LOG) b |= AGPU_PROF_TIMING;
  p->profile = b;
  p->t_valid = false;
  return AGPU_OK;
}

agpu_status agpu_pipeline_last_kernel_ns(agpu_pipeline* p, uint64_t* out_ns, const char** out_name) {
  AGPU_REQUIRE(p && p->dev && out_ns, AGPU_ERR_ARG, "null argument");
  AGPU_HIP(hipSetDevice(p->dev->ordinal));
  AGPU_REQUIRE(p->t_valid, AGPU_ERR_ARG, "no timed launch yet (agpu_pipeline_enable_timing / AGPU_PROFILE=2)");
  AGPU_HIP(hipEventSynchronize(p->t1));
  float ms = 0.0f;
  AGPU_HIP(hipEventElapsedTime(&ms, p->t0, p->t1));
  *out_ns = (uint64_t)((double)ms * 1e6 + 0.5);
  if (out_name) *out_name = p->last_name;
  return AGPU_OK;
}

// ---------------------------------------------------------------- tuning
agpu_status agpu_set_tuning(const char* key, int64_t value) {
  if (key && !strcmp(key, "pool_arena")) {
    g_pool_arena.store(value, std::memory_order_relaxed);
    return AGPU_OK;
  }
  if (key && !strcmp(key, "mem_pool")) {
    g_mem_pool.store(value, std::memory_order_relaxed);
    return AGPU_OK;
  }
  const int i = tune_index(key);
  AGPU_REQUIRE(i >= 0, AGPU_ERR_ARG, "unknown tuning key");
  g_tune_default[i].store(value, std::memory_order_relaxed);
  return AGPU_OK;
}

agpu_status agpu_get_tuning(const char* key, int64_t* out_value) {
  AGPU_REQUIRE(out_value, AGPU_ERR_ARG, "null out_value");
  if (key && !strcmp(key, "pool_arena")) {
    *out_value = g_pool_arena.load(std::memory_order_relaxed);
    return AGPU_OK;
  }
  if (key && !strcmp(key, "mem_pool")) {
    *out_value = g_mem_pool.load(std::memory_order_relaxed);
    return AGPU_OK;
  }
  const int i = tune_index(key);
  AGPU_REQUIRE(i >= 0, AGPU_ERR_ARG, "unknown tuning key");
  *out_value = g_tune_default[i].load(std::memory_order_relaxed);
  return AGPU_OK;
}

agpu_status agpu_pipeline_set_tuning(agpu_pipeline* p, const char* key, int64_t value) {
  AGPU_REQUIRE(p, AGPU_ERR_ARG, "null pipeline");
  const int i = tune_index(key);
  AGPU_REQUIRE(i >= 0, AGPU_ERR_ARG, "unknown tuning key");
  reinterpret_cast<int64_t*>(&p->tune)[i] = value;
  return AGPU_OK;
}

agpu_status agpu_pipeline_get_tuning(agpu_pipeline* p, const char* key, int64_t* out_value) {
  AGPU_REQUIRE(p && out_value, AGPU_ERR_ARG, "null argument");
  const int i = tune_index(key);
  AGPU_REQUIRE(i >= 0, AGPU_ERR_ARG, "unknown tuning key");
  *out_value = reinterpret_cast<const int64_t*>(&p->tune)[i];
  return AGPU_OK;
}

}  // extern "C"

// ---------------------------------------------------------------- call scope (AGPU_BIND)
agpu_status agpu_scope_enter(agpu_pipeline* p, const char* name) {
  if (!p || !p->dev) {
    agpu_set_error("%s: null pipeline", name);
    return AGPU_ERR_ARG;
  }
  AGPU_NOT_POISONED(p->dev);  // every pipeline call: uploads, downloads, syncs and launches alike would queue or wait behind it
  AGPU_HIP(hipSetDevice(p->dev->ordinal));
  agpu_device* dev = p->dev;
#ifndef AGPU_TEST_NO_ORDERING
  if (p->scope_depth == 0 && !p->capturing && p->seen_gen != dev->finish_gen.load(std::memory_order_acquire)) {
    std::lock_guard<std::mutex> lock(dev->mu);
    for (agpu_stream_slot* s : dev->slots) {
      if (s == p->slot || !s->finish_ev || s->finish_gen <= p->seen_gen || ref_done_locked(s->finish_ev)) continue;
      AGPU_HIP(hipStreamWaitEvent(p->stream, s->finish_ev->ev, 0));
    }
    p->seen_gen = dev->finish_gen.load(std::memory_order_relaxed);
  }
#endif  // AGPU_TEST_NO_ORDERING: tools/probe builds a variant without the wait to prove the ordering tests can fail
  if (p->profile && !p->capturing) {
    if (p->profile & AGPU_PROF_ROCTX) roctxRangePushA(name);
    if ((p->profile & AGPU_PROF_TIMING) && p->scope_depth == 0) {
      if (!p->t0) {
        AGPU_HIP(hipEventCreate(&p->t0));
        AGPU_HIP(hipEventCreate(&p->t1));
      }
      p->t_valid = false;
      p->last_name = name;
      AGPU_HIP(hipEventRecord(p->t0, p->stream));
    }
  }
  if (p->scope_depth++ == 0) p->slot->enq.fetch_add(1, std::memory_order_release);  // odd: call in progress
  p->dirty = true;
  p->copy_tail = false;
  return AGPU_OK;
}

void agpu_scope_exit(agpu_pipeline* p) {
  p->scope_depth--;
  if (p->scope_depth == 0) p->slot->enq.fetch_add(1, std::memory_order_release);  // even: everything it enqueued is counted
  if (p->profile && !p->capturing) {
    if ((p->profile & AGPU_PROF_TIMING) && p->scope_depth == 0 && p->t0) {
      if (hipEventRecord(p->t1, p->stream) == hipSuccess) p->t_valid = true;
      if (p->t_valid && (p->profile & AGPU_PROF_LOG) && hipEventSynchronize(p->t1) == hipSuccess) {
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, p->t0, p->t1) == hipSuccess)
          fprintf(stderr, "[arrow_gpu] Time taken for compute pass %s: %.0f ns\n", p->last_name, (double)ms * 1e6);
      }
    }
    if (p->profile & AGPU_PROF_ROCTX) roctxRangePop();
  }
}

// Name the innermost open range after the reference's entry point (agpu_launch_by_name).
void agpu_scope_label(agpu_pipeline* p, const char* label) {
  if (p->profile & AGPU_PROF_ROCTX) roctxMarkA(label);
  if (p->scope_depth == 1) p->last_name = label;
}

agpu_status agpu_scratch(agpu_pipeline* p, size_t bytes, void** out) {
  agpu_stream_slot* s = p->slot;
  if (!s->scratch || s->scratch->bytes < bytes) {
    AGPU_REQUIRE(!p->capturing, AGPU_ERR_ARG, "scratch growth during graph capture; run the op once before capturing");
    const size_t want = bytes < (1u << 20) ? (1u << 20) : bytes;
    void* ptr = nullptr;
    AGPU_HIP(hipMalloc(&ptr, want));
    agpu_scratch_block* old = s->scratch;
    if (old) AGPU_HIP(hipStreamSynchronize(p->stream));  // queued reductions still read the old block
    std::lock_guard<std::mutex> lock(p->dev->mu);
    s->scratch = new agpu_scratch_block{ptr, want, 1};
    if (old) scratch_release_locked(old);  // stays alive if a captured graph references it
  }
  *out = s->scratch->ptr;
  return AGPU_OK;
}

