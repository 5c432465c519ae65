// common.hpp — shared internals of libarrow_gpu_hip.so (gfx950 only; no CPU path, no CUDA shims).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <unordered_map>
#include <utility>
#include <vector>

#include "../../include/arrow_gpu.h"

#define AGPU_STR2(x) #x
#define AGPU_STR(x) AGPU_STR2(x)

// ---------------------------------------------------------------- handles
struct agpu_device {
  int ordinal;
  hipDeviceProp_t props;
  int num_cus;
  // 8 KiB of f64 {sin, cos} pairs for the 16-bit fused trig kernels (elementwise.hip: trig16_kernel), built once at
  // device creation: [l] = sincos(l), [256 + h] = sincos(256·h), l, h ∈ 0..255
  void* trig16_table;
  // 2 KiB of f64 {1/c, −log2(1/c)} pairs for f32 pow (elementwise.hip: pow_f32_dev); same allocation, + 8 KiB
  void* pow_table;

  // ---- resource pools (runtime.hip).  Measured on MI355X / ROCm 7: hipStreamCreate 4.3 ms + hipStreamDestroy 2.6 ms,
  // hipFree 0.2 ms (implicit device sync), hipMalloc of a 4 GB block 0.2–60 ms — against 0.19 ms for the kernel of a
  // 1e8-row add.  The reference's default API creates a pipeline and an output buffer PER OP, so both are pooled.
  struct StreamSlot {  // an idle owned stream with the reduction scratch and the error word that travel with it
    hipStream_t stream;
    void* scratch;
    size_t scratch_bytes;
    uint32_t* flags;
  };
  struct CachedBlock {  // a freed device block; `pending` = events recorded on every stream at free time
    void* ptr;
    std::vector<hipEvent_t> pending;
  };
  std::mutex mu;
  std::vector<StreamSlot> idle_streams;
  std::vector<hipStream_t> all_streams;             // every stream work may be queued on (owned + wrapped, live or idle)
  std::unordered_map<void*, size_t> block_size;     // live pooled-size blocks handed out by agpu_malloc
  std::multimap<size_t, CachedBlock> cache;         // size → freed blocks
  size_t cached_bytes = 0, cache_cap = 0;
  std::vector<hipEvent_t> event_pool;
};
#define AGPU_TABLE_BYTES (512 * 16 + 128 * 16)
agpu_status agpu_internal_build_tables(void* trig16_table, void* pow_table);  // elementwise.hip; synchronous

struct agpu_pipeline {
  agpu_device* dev;
  hipStream_t stream;
  bool owns_stream;
  bool capturing;
  // scratch for reductions / popcount partials (allocated on first use, reused; stream-ordered so one per pipeline)
  void* scratch;
  size_t scratch_bytes;
  // sticky error word in pinned host memory, written by kernels (bit 0: take/put index out of range), read and cleared
  // by agpu_pipeline_sync — no pre-pass over the index column, no readback, the pipeline stays asynchronous
  uint32_t* flags;
};
#define AGPU_FLAG_INDEX_RANGE 1u

struct agpu_event {
  agpu_device* dev;
  hipEvent_t ev;
};

struct agpu_graph {
  agpu_device* dev;
  hipGraph_t graph;
  hipGraphExec_t exec;
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

static inline agpu_status agpu_bind(agpu_pipeline* p) {
  if (!p || !p->dev) {
    agpu_set_error("null pipeline");
    return AGPU_ERR_ARG;
  }
  AGPU_HIP(hipSetDevice(p->dev->ordinal));
  return AGPU_OK;
}
#define AGPU_BIND(p)                         \
  do {                                       \
    agpu_status _s = agpu_bind(p);           \
    if (_s != AGPU_OK) return _s;            \
  } while (0)

agpu_status agpu_scratch(agpu_pipeline* p, size_t bytes, void** out);

// ---------------------------------------------------------------- tuning knobs (bench sweeps; see agpu_set_tuning)
struct agpu_tuning {
  int64_t stream_grid;    // blocks for streaming kernels (0 = auto: CUs * stream_blocks_per_cu)
  int64_t stream_bpc;     // blocks per CU when stream_grid == 0
  int64_t stream_unroll;  // 16-byte vectors in flight per lane per array: 1, 2, 4 or 8
  int64_t stream_nt;      // bit0: nontemporal loads, bit1: nontemporal stores
  int64_t cmp_variant;    // 0 = ballot (dword loads), 1 = vector loads + nibble shuffle
  int64_t reduce_grid;    // blocks for reductions (0 = auto)
  int64_t table_tiles;    // tiles per block for kernels that stage a lookup table in LDS (lut8 / trig16 / pow)
  int64_t mem_pool;       // 1 = cache freed device blocks ≥ 1 MiB and idle streams (default), 0 = hipMalloc/hipFree every time
};
extern agpu_tuning g_tune;

// Grid for a streaming kernel that owns `tiles` block-tiles.  Measured on MI355X at 1e9 rows (profiles/
// r01_sweep_add_f32_1e9.json): ONE tile per block beats every persistent grid (6.54 vs ≤6.52 TB/s at 32768 blocks,
// 5.3 TB/s at 2048), so the default is grid = tiles; stream_grid > 0 forces a persistent grid-stride launch and
// stream_bpc > 0 a blocks-per-CU one (both kept for sweeps).  Kernels still grid-stride, so any grid is correct.
static inline int stream_grid_for(const agpu_pipeline* p, uint64_t tiles) {
  uint64_t g = tiles;
  if (g_tune.stream_grid > 0) g = (uint64_t)g_tune.stream_grid;
  else if (g_tune.stream_bpc > 0) g = (uint64_t)p->dev->num_cus * (uint64_t)g_tune.stream_bpc;
  if (g > tiles) g = tiles;
  if (g > 0x3FFFFFFFull) g = 0x3FFFFFFFull;  // hipDim3.x limit headroom
  if (g < 1) g = 1;
  return (int)g;
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
