/*
 * arrow_gpu.h — C ABI of the MI355X-native Arrow compute-kernel library
 * (libarrow_gpu_hip.so, hand-written HIP kernels built for gfx950).
 *
 * This header is the drop-in seam for psvri/arrow-gpu's device runtime:
 * it replaces `arrow_gpu_array::gpu_utils::{GpuDevice, ArrowComputePipeline}`
 * and the `compute_shaders/ **.wgsl` entry points they dispatch.  Every entry
 * point cites the reference interface it replaces as
 *   [ref: <path under psvri/arrow-gpu>:<lines>].
 *
 * Conventions
 *   - plain C: opaque handles, raw DEVICE pointers (`void*` returned by
 *     agpu_malloc, or any hipMalloc'ed / torch-owned HBM pointer) and element
 *     counts.  No C++/torch types cross this boundary.
 *   - every call returns agpu_status; 0 = AGPU_OK.  Nothing aborts; the text of
 *     the last failure on the calling thread is agpu_last_error().
 *   - kernel calls are ASYNCHRONOUS on the pipeline's HIP stream and execute in
 *     call order (the reference records passes into one wgpu CommandEncoder and
 *     submits them in order; a HIP stream gives the same ordering eagerly).
 *     The only blocking calls are agpu_download, agpu_pipeline_sync,
 *     agpu_device_sync and agpu_event_elapsed_ms.
 *   - Arrow layout: values are contiguous little-endian; Boolean data and
 *     validity are LSB-first bitmaps, bit i at byte i/8 mask 1<<(i%8), bit set =
 *     true/valid [ref: crates/array/src/array/null_bit_buffer.rs:47-61].
 *     Bitmap buffers handed to kernels must be readable/writable up to the next
 *     multiple of 8 bytes (agpu_bitmap_bytes(n_bits)); kernels write 0 to the
 *     padding bits past n_bits unless stated (the reference leaves them
 *     unspecified).
 *   - there is NO CPU fallback in this library: without a gfx950 device
 *     agpu_device_create fails with AGPU_ERR_NO_DEVICE.
 */
#ifndef ARROW_GPU_H
#define ARROW_GPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AGPU_ABI_VERSION 2

typedef int32_t agpu_status;
enum {
  AGPU_OK = 0,
  AGPU_ERR_UNSUPPORTED = 1, /* op/dtype pair the reference panics on ("Operation not supported") */
  AGPU_ERR_SHAPE = 2,       /* length / alignment / index-range violation */
  AGPU_ERR_HIP = 3,         /* a HIP runtime call failed; see agpu_last_error() */
  AGPU_ERR_ARG = 4,         /* null handle / null pointer / bad enum */
  AGPU_ERR_NO_DEVICE = 5    /* no gfx950 device visible */
};

/* [ref: crates/array/src/array/mod.rs:40-50  enum ArrowType] */
typedef enum {
  AGPU_BOOL = 0,
  AGPU_F32 = 1,
  AGPU_U32 = 2,
  AGPU_U16 = 3,
  AGPU_U8 = 4,
  AGPU_I32 = 5,
  AGPU_I16 = 6,
  AGPU_I8 = 7,
  AGPU_DATE32 = 8 /* i32 storage [ref: crates/array/src/array/mod.rs:90] */
} agpu_dtype;

/* Binary / scalar ops.  [ref: crates/arithmetic/compute_shaders/{f32,i32,u32}/{array,scalar}.wgsl,
 * crates/compare/compute_shaders/ * /min_max.wgsl, crates/logical/compute_shaders/ * /{logical,shift}.wgsl,
 * crates/math/compute_shaders/{f32/floatbinary,i32/binary}.wgsl] */
typedef enum {
  AGPU_OP_ADD = 0,
  AGPU_OP_SUB = 1,
  AGPU_OP_MUL = 2,
  AGPU_OP_DIV = 3,
  AGPU_OP_REM = 4,
  AGPU_OP_MIN = 5,
  AGPU_OP_MAX = 6,
  AGPU_OP_AND = 7,
  AGPU_OP_OR = 8,
  AGPU_OP_XOR = 9,
  AGPU_OP_SHL = 10, /* rhs is always a u32 array */
  AGPU_OP_SHR = 11, /* rhs is always a u32 array; arithmetic for signed types */
  AGPU_OP_POW = 12
} agpu_binary_op;

/* Unary ops.  [ref: crates/arithmetic/compute_shaders/f32/neg.wgsl, crates/logical/compute_shaders/ * /not.wgsl,
 * crates/math/compute_shaders/f32/floatunary.wgsl, crates/math/compute_shaders/i32/unary.wgsl,
 * crates/trigonometry/compute_shaders/ * /{trigonometry,hyperbolic}.wgsl] */
typedef enum {
  AGPU_UN_NEG = 0,
  AGPU_UN_ABS = 1,
  AGPU_UN_NOT = 2,
  AGPU_UN_SQRT = 3,
  AGPU_UN_CBRT = 4,
  AGPU_UN_EXP = 5,
  AGPU_UN_EXP2 = 6,
  AGPU_UN_LOG = 7,
  AGPU_UN_LOG2 = 8,
  AGPU_UN_SIN = 9,   /* for u8/i8/u16/i16 inputs: fused cast+sin, output f32 */
  AGPU_UN_COS = 10,  /* idem */
  AGPU_UN_ACOS = 11, /* f32 only */
  AGPU_UN_SINH = 12, /* for small ints: fused, output f32 */
  AGPU_UN_POPCOUNT = 13 /* integer types: set bits per element [ref: crates/logical/compute_shaders/u32/countbitones.wgsl:9-15
                           `countob`, the first half of BooleanArrayGPU::all(), crates/logical/src/boolean.rs:120-146] */
} agpu_unary_op;

/* [ref: crates/compare/src/lib.rs:17-21 entry points "gt","gteq","lt","lteq","eq"] */
typedef enum { AGPU_CMP_GT = 0, AGPU_CMP_GTEQ = 1, AGPU_CMP_LT = 2, AGPU_CMP_LTEQ = 3, AGPU_CMP_EQ = 4 } agpu_cmp_op;

/* Whole-column reductions.  SUM is the reference's `Sum` [ref: crates/arithmetic/src/aggregate_kernels.rs:24-51];
 * MIN/MAX reductions are new (north_star config 5), Arrow min_max semantics: NaN is ignored unless all NaN. */
typedef enum { AGPU_RED_SUM = 0, AGPU_RED_MIN = 1, AGPU_RED_MAX = 2 } agpu_reduce_op;

typedef struct agpu_device agpu_device;     /* replaces GpuDevice  [ref: crates/array/src/gpu_utils/gpu_device.rs:29-33] */
typedef struct agpu_pipeline agpu_pipeline; /* replaces ArrowComputePipeline [ref: crates/array/src/gpu_utils/compute_pipeline.rs:8-12] */
typedef struct agpu_event agpu_event;       /* replaces CmpQuery timestamps [ref: crates/array/src/gpu_utils/compute_query.rs:7-52] */

/* ---------------------------------------------------------------- misc */
int32_t agpu_abi_version(void);
const char* agpu_last_error(void);  /* thread-local, never NULL */
const char* agpu_build_info(void);  /* e.g. "arrow_gpu_hip gfx950 hip-7.2" */
size_t agpu_dtype_size(agpu_dtype t); /* bytes per element; 0 for AGPU_BOOL (bit-packed) */
size_t agpu_bitmap_bytes(uint64_t n_bits); /* ceil(n_bits/64)*8 : allocation size every bitmap argument must have */

/* ---------------------------------------------------------------- device
 * [ref: GpuDevice::new / from_adapter, crates/array/src/gpu_utils/gpu_device.rs:46-106;
 *       global GPU_DEVICE, crates/array/src/lib.rs:17] */
agpu_status agpu_device_count(int32_t* out_count);
agpu_status agpu_device_create(int32_t ordinal, agpu_device** out_device);
agpu_status agpu_device_destroy(agpu_device* dev);
/* agpu_device_sync waits for everything THIS LIBRARY has queued on the device: every pipeline's stream, pooled streams of destroyed pipelines,
 * wrapped streams (work other runtimes queued on streams of their own is theirs to wait for).  With ONE stream outstanding — the usual
 * state behind the reference-style immediate ops of one thread — or up to three the wait goes through a
 * kernel per stream that posts into pinned host memory (tuning "sync_spin", docs/experiments.md R5.10: 6–9 µs instead of hipDeviceSynchronize's
 * 11); with more it IS hipDeviceSynchronize.  A stream with no call of this library since its last completed wait is known to be empty and is
 * not waited for again — unless its raw handle was handed out (agpu_pipeline_stream) or it is a wrapped stream: those may carry work the
 * library never saw and are waited for every time.  Tuning sync_spin < 0: always hipDeviceSynchronize, whatever the counters say.
 * agpu_device_download: the same wait with up to AGPU_MAILBOX_MAX_BYTES of device memory delivered on the way — `values()` of a reduction's result or of a small array in one
 * wait instead of two [ref: GpuDevice::retrive_data gpu_device.rs:232-265 polls the whole queue, then maps the staging buffer]. */
agpu_status agpu_device_sync(agpu_device* dev);
#define AGPU_MAILBOX_MAX_BYTES 3840 /* what a pinned mailbox carries: agpu_device_download's limit; agpu_download up to this size and agpu_upload up to 1 KiB take the same route */
agpu_status agpu_device_download(agpu_device* dev, void* dst_host, const void* src_dev, size_t bytes);
agpu_status agpu_device_name(agpu_device* dev, char* out, size_t out_cap); /* e.g. "gfx950:sramecc+:xnack-" */
agpu_status agpu_device_ordinal(agpu_device* dev, int32_t* out_ordinal);
agpu_status agpu_device_mem_info(agpu_device* dev, uint64_t* out_free, uint64_t* out_total);
/* Resource pools.  The reference allocates an output buffer and (in its default, non-`_op` API) a command encoder per
 * operation [ref: impl_arithmetic_op! crates/arithmetic/src/lib.rs:11-50 — `ArrowComputePipeline::new` … `finish()`
 * around every op].  On ROCm a stream costs 4.3 ms to create and 2.6 ms to destroy and hipFree synchronises the device,
 * so idle streams and freed blocks are recycled (tuning key "mem_pool", default 1): blocks > 1 MiB whole, in 2 MiB
 * granules; smaller blocks (the reference's tests and examples live at 5–100 elements) from 2 MiB slabs carved into
 * power-of-two size classes, 256 B … 1 MiB.  agpu_device_trim returns the cached blocks and fully free slabs to the
 * driver (also done automatically when hipMalloc runs out of memory); cached large blocks never exceed half of the
 * device memory. */
agpu_status agpu_device_trim(agpu_device* dev);
agpu_status agpu_device_pool_info(agpu_device* dev, uint64_t* out_cached_bytes, uint64_t* out_cached_blocks,
                                  uint64_t* out_idle_streams); /* the > 1 MiB cache and the idle streams */
agpu_status agpu_device_small_pool_info(agpu_device* dev, uint64_t* out_slab_bytes, uint64_t* out_free_blocks,
                                        uint64_t* out_live_blocks); /* the <= 1 MiB slab pool */

/* ---------------------------------------------------------------- buffers (raw HBM pointers)
 * agpu_malloc          [ref: GpuDevice::create_empty_buffer gpu_device.rs:183-192] — zero_fill!=0 reproduces wgpu's
 *                      zero-initialised buffers; kernels here never rely on it.
 * agpu_upload          [ref: create_gpu_buffer_with_data gpu_device.rs:171-181, create_scalar_buffer :203-210]
 * agpu_download        [ref: retrive_data gpu_device.rs:232-265] — blocks until the work queued on `p` (required, non-NULL) has drained.
 * agpu_copy            [ref: clone_buffer(_pass) gpu_device.rs:212-230; ArrowComputePipeline::{clone_buffer,
 *                      copy_buffer_to_buffer} compute_pipeline.rs:275-299] — ordered on the pipeline's stream. */
agpu_status agpu_malloc(agpu_device* dev, size_t bytes, int32_t zero_fill, void** out_ptr);
agpu_status agpu_free(agpu_device* dev, void* ptr);
/* Pool placement.  Blocks of ≥ 1 GiB are carved out of arenas (one hipMalloc of up to 32 GiB each) at multiples of 512 MiB
 * plus a colour of 0 / 8 / 4 / 12 KiB, so that the buffers an ordinary caller allocates one by one relate to each other the
 * way the columns of an agpu_malloc_table do (the HBM channel hash) — agpu_malloc rotates the colour over
 * successive allocations; agpu_malloc_like picks, for a buffer that will be used TOGETHER with `neighbours` (the output of
 * an op and its inputs: what every `*_op` of the reference's API allocates, [ref: apply_binary_function
 * compute_pipeline.rs:68-113 `create_empty_buffer(new_buffer_size)`]), the arena the neighbours live in and the colour that
 * differs from theirs in the strongest hash bit.  Smaller blocks and neighbours outside any arena: plain agpu_malloc.
 * Sizes round up to 512 MiB; tuning "pool_arena" = 0 switches the arenas off. */
agpu_status agpu_malloc_like(agpu_device* dev, size_t bytes, int32_t zero_fill, const void* const* neighbours,
                             int32_t n_neighbours, void** out_ptr);
/* The buffers of one table (the columns a kernel will read together) in ONE block, placed for the HBM channel hash:
 * 2 MiB-aligned allocations put element i of every column into the same hash class whenever their distance has no hash
 * bit set, and two read streams in the same class cost a compare 10 % of its bandwidth (0.78 → 0.85–0.89 of the roof;
 * f32 add 1.5 %: DESIGN.md §3, tools/probe/hash_bits.py).  Columns of 1 GiB and more start a multiple of 512 MiB apart
 * plus 0 / 8 / 4 / 12 KiB by column index; smaller ones a multiple of 2 MiB plus the same colours.  Every out_ptrs[k] is
 * an ordinary buffer: free each with agpu_free (any order); the block returns to the pool with the last one.
 * Not in the reference (wgpu places buffers). */
agpu_status agpu_malloc_table(agpu_device* dev, int32_t n_columns, const uint64_t* bytes, int32_t zero_fill, void** out_ptrs);
/* Blocking copies: complete on return, and everything enqueued on the pipeline before them is over.  [ref: create_gpu_buffer_with_data
 * gpu_device.rs:145-156 / retrive_data :232-265]  Small ones — uploads of <= 1 KiB, downloads of <= AGPU_MAILBOX_MAX_BYTES: the
 * reference's tests and examples live at 5-100 elements — never reach the runtime's copy engine or its stream wait: a one-wave kernel moves
 * the bytes between device memory and the pipeline's pinned mailbox and posts a sequence number the host spins on (7-9 us instead of 15;
 * tuning "sync_spin" below).  agpu_download of 0 bytes is a pipeline sync. */
agpu_status agpu_upload(agpu_pipeline* p, void* dst_dev, const void* src_host, size_t bytes);
agpu_status agpu_download(agpu_pipeline* p, void* dst_host, const void* src_dev, size_t bytes);
agpu_status agpu_copy(agpu_pipeline* p, void* dst_dev, const void* src_dev, size_t bytes);
agpu_status agpu_memset(agpu_pipeline* p, void* dst_dev, int32_t byte_value, size_t bytes);

/* Pinned-host staging for ingest/egress (the reference moves every array through pageable host Vecs,
 * [ref: PrimitiveArrayGpu::from_slice / raw_values, crates/array/src/array/primitive_array_gpu.rs:55-74]).
 * agpu_host_alloc returns page-locked memory; the *_async copies only ENQUEUE on the pipeline's stream (the host
 * buffer must stay untouched until agpu_pipeline_sync) so H2D, kernels and D2H of different pipelines overlap. */
agpu_status agpu_host_alloc(agpu_device* dev, size_t bytes, void** out_host_ptr);
agpu_status agpu_host_free(agpu_device* dev, void* host_ptr);
agpu_status agpu_upload_async(agpu_pipeline* p, void* dst_dev, const void* src_pinned, size_t bytes);
agpu_status agpu_download_async(agpu_pipeline* p, void* dst_pinned, const void* src_dev, size_t bytes);

/* ---------------------------------------------------------------- pipeline = one HIP stream
 * [ref: ArrowComputePipeline::new compute_pipeline.rs:15-22; finish :259-273 (submit, no wait)].
 * One pipeline per host thread; many pipelines may share a device concurrently.
 * agpu_pipeline_wrap_stream adopts an existing hipStream_t (e.g. torch's current stream); it is not destroyed. */
agpu_status agpu_pipeline_create(agpu_device* dev, agpu_pipeline** out_pipeline);
agpu_status agpu_pipeline_wrap_stream(agpu_device* dev, void* hip_stream, agpu_pipeline** out_pipeline);
/* finish = the reference's submit point.  Work is already enqueued, so it does NOT wait — but it PUBLISHES the stream's
 * position: every other pipeline of the device orders its next call behind it.  The reference has ONE queue, so whatever
 * is submitted later runs after whatever was submitted earlier [ref: compute_pipeline.rs:259-263]; with one HIP stream
 * per pipeline the same guarantee is: a pipeline may read what another pipeline produced once that pipeline has finished
 * (or was destroyed — destroy publishes too).  Cost when nobody finished anything: one atomic load per call. */
agpu_status agpu_pipeline_finish(agpu_pipeline* p);
agpu_status agpu_pipeline_sync(agpu_pipeline* p);   /* host waits for everything enqueued so far */
agpu_status agpu_pipeline_destroy(agpu_pipeline* p);
agpu_status agpu_pipeline_device(agpu_pipeline* p, agpu_device** out_device);
/* the pipeline's hipStream_t.  From this call on the stream counts as carrying foreign work (see agpu_device_sync), until the pipeline is destroyed */
agpu_status agpu_pipeline_stream(agpu_pipeline* p, void** out_hip_stream);
/* Explicit dependency for hosts that overlap pipelines on purpose (double-buffered staging): work enqueued on `p` after
 * this call runs after everything enqueued on `other` so far.  Does not block the host, does not need a finish. */
agpu_status agpu_pipeline_wait_pipeline(agpu_pipeline* p, agpu_pipeline* other);

/* hipGraph capture of a launch-bound op chain (examples/simple.rs-style `*_op` chains).
 * begin → enqueue ops on p → end (returns a replayable graph) → agpu_graph_launch any number of times.
 * Reductions and popcounts use per-stream scratch whose POINTER is baked into the captured kernel nodes: run such an op
 * once before capturing (scratch cannot grow during capture); the graph keeps that scratch block alive for its own
 * lifetime, and must be replayed on the pipeline that captured it or while that pipeline is idle. */
typedef struct agpu_graph agpu_graph;
agpu_status agpu_pipeline_begin_capture(agpu_pipeline* p);
agpu_status agpu_pipeline_end_capture(agpu_pipeline* p, agpu_graph** out_graph);
agpu_status agpu_graph_launch(agpu_graph* g, agpu_pipeline* p);
agpu_status agpu_graph_destroy(agpu_graph* g);

/* HIP events on the pipeline's own stream [ref: CmpQuery, compute_query.rs:15-75 — 2-slot timestamp query per pass] */
agpu_status agpu_event_create(agpu_device* dev, agpu_event** out_event);
agpu_status agpu_event_record(agpu_event* e, agpu_pipeline* p);
agpu_status agpu_event_elapsed_ms(agpu_event* start, agpu_event* stop, float* out_ms); /* syncs on `stop` */
agpu_status agpu_event_destroy(agpu_event* e);

/* Per-launch profiling [ref: GpuDevice::compute_pass → insert_debug_marker(entry_point) gpu_device.rs:132; CmpQuery
 * compute_query.rs:7-89 — timestamp pair per pass, wait_for_results() logs "Time taken for compute pass"].
 * profile_bits: 1 = a roctx range named after the ABI call (and a roctx mark with the reference's shader/entry-point for
 * agpu_launch_by_name*) around every launch — shows up in `rocprofv3 --marker-trace`; 2 = a HIP event pair around every
 * launch, read back with agpu_pipeline_last_kernel_ns (blocks on the stop event; out_name = the call's name, static
 * storage); 4 = additionally wait after every launch and log the reference's line to stderr.  The environment variable
 * AGPU_PROFILE=<bits> sets the default for every new pipeline; 0 (default) costs nothing. */
agpu_status agpu_pipeline_enable_timing(agpu_pipeline* p, int32_t profile_bits);
agpu_status agpu_pipeline_last_kernel_ns(agpu_pipeline* p, uint64_t* out_ns, const char** out_name);

/* Launch tuning (sweeps and tests; defaults are the measured best).  SEVEN keys (round 6 removed ten whose other values were never better,
 * the adaptive tiles-per-block policy of round 5 among them — "tile_auto": with the occupancy caps in place it decided "one tile" everywhere):
 *   "stream_grid"   blocks of the streaming kernels: 0 = one tile per block (default), > 0 = that many, grid-striding (tests: the loop paths)
 *   "cmp_variant"   4-byte compares: 0 = ballot (default), 1 = vector loads + nibble shuffle
 *   "gather_bucket" take / put: 0 = auto (size thresholds + the device-side locality probe), 1 = direct kernels, 2 = bucketed pipelines
 *                   whenever the shape qualifies, 4 = like 2 but with the probe (tests)
 *   "h2d_mode"      host staging of agpu_import_arrow / agpu_export_arrow: 0 = auto, 1 = pageable copy, 2 = threaded pinned staging, 3 = hipHostRegister
 *   "tiles"         tiles per block of the kernels that issue the NEXT tile's loads before they evaluate the current one — the VALU-heavy f32
 *                   unary kernels, the widening casts and cast-headed chains, the LDS-table kernels: 0 = each kernel's static default, > 0 = that many
 *   "wave_lds"      unused dynamic LDS per wave that caps the waves per CU of sin / cos f32, the widening casts to 32 bits and the 8-bit table
 *                   kernels: 0 = each kernel's measured default (≈ 24 or 16 waves per CU instead of 32: +3–9 % on those kernels,
 *                   docs/experiments.md R5.5, R6.2), < 0 = no cap, > 0 = that many bytes
 *   "sync_spin"     agpu_pipeline_sync, agpu_download of <= AGPU_MAILBOX_MAX_BYTES and agpu_upload of <= 1 KiB wait through the pipeline's pinned
 *                   MAILBOX — a one-wave kernel queued behind the pipeline's work copies the bytes into (out of) pinned host memory and posts a
 *                   sequence number the host spins on (one kernel + one scalar back: 6.7 µs instead of 15; docs/experiments.md R5.10): 0 (default) =
 *                   on, spinning for at most 200 µs before the blocking wait; > 0 = that many µs; < 0 = off (runtime copies and waits).
 *                   AGPU_SYNC_SPIN=<n> in the environment sets the process-wide default before the first device is created.
 * Unknown key → AGPU_ERR_ARG.  Results never depend on any of these.  Every pipeline carries its own copy: agpu_set_tuning changes the
 * process default that pipelines created AFTERWARDS start from (atomic, any thread), agpu_pipeline_set_tuning changes one pipeline only — a
 * sweep on one thread never changes the kernels another pipeline launches.  "mem_pool" (0/1, agpu_set_tuning only) switches the device-level
 * block and stream pools, "pool_arena" (0/1, agpu_set_tuning only) the placed arenas behind blocks of >= 1 GiB. */
agpu_status agpu_set_tuning(const char* key, int64_t value);
agpu_status agpu_get_tuning(const char* key, int64_t* out_value);
agpu_status agpu_pipeline_set_tuning(agpu_pipeline* p, const char* key, int64_t value);
agpu_status agpu_pipeline_get_tuning(agpu_pipeline* p, const char* key, int64_t* out_value);

/* ---------------------------------------------------------------- element-wise kernels
 * All take element counts `n` (rows).  in/out pointers must be aligned to the element size; 16-byte alignment
 * selects the vectorised path.  `out` may alias an input exactly (in place), never partially overlap. */

/* out[i] = a[i] op b[i].
 * [ref: ArrowComputePipeline::apply_binary_function compute_pipeline.rs:68-113 as called by
 *  impl_arithmetic_array_op! crates/arithmetic/src/lib.rs:54-94 (add_f32/sub_f32/mul_f32/div_f32, add_i32, add_u32),
 *  apply_function_min_max! crates/compare/src/lib.rs:113-140, Logical crates/logical/src/lib.rs:88-158, MathBinary
 *  crates/math/src/lib.rs].  Supported: f32 {ADD,SUB,MUL,DIV,REM,MIN,MAX,POW}; i32/u32/date32 {ADD,SUB,MUL,DIV,REM,MIN,MAX,
 *  AND,OR,XOR,SHL,SHR} (+POW i32); u16/i16/u8/i8 {ADD,SUB,MUL,MIN,MAX,AND,OR,XOR,SHL,SHR}.  For SHL/SHR `b` is u32[n]. */
agpu_status agpu_binary(agpu_pipeline* p, agpu_binary_op op, agpu_dtype dtype, const void* a, const void* b, void* out,
                        uint64_t n);

/* out[i] = a[i] op *scalar, `scalar` = DEVICE pointer to a 1-element array of `dtype` (the reference binds a
 * 1-element storage buffer) [ref: apply_scalar_function compute_pipeline.rs:167-213 as called by impl_arithmetic_op!
 * crates/arithmetic/src/lib.rs:11-50; entry points f32_add.. i32_rem.. u32_div.. u16_add]. */
agpu_status agpu_scalar(agpu_pipeline* p, agpu_binary_op op, agpu_dtype dtype, const void* a, const void* scalar,
                        void* out, uint64_t n);

/* out[i] = op(in[i]).  Output dtype = input dtype, except SIN/COS/SINH on u8/i8/u16/i16 which write f32[n]
 * (the reference's fused sin_u8-style kernels) — every one of them bit-identical to the cast followed by the f32 function.
 * Accuracy of the f32 transcendentals: within 1 ULP of f64 libm rounded once on EVERY bit pattern (SIN / COS / LOG are evaluated in
 * packed f32 since round 6 and proven exhaustively; the reference pins them to 0.01 absolute only).
 * [ref: apply_unary_function compute_pipeline.rs:24-66 as called by Neg arithmetic_kernels.rs:270-319, bitwise_not
 *  logical/src/lib.rs:135-158, apply_unary_function_op! crates/math/src/lib.rs:138-193 and
 *  crates/trigonometry/src/lib.rs:85-137]. */
agpu_status agpu_unary(agpu_pipeline* p, agpu_unary_op op, agpu_dtype dtype, const void* in, void* out, uint64_t n);
/* Self-test of an f32 function over a RANGE OF BIT PATTERNS: every pattern in [first_bits, first_bits + count) (count
 * up to 2^32 = all of f32) goes through the kernel's own arithmetic and through the f64 device library rounded once to
 * f32; *out_max_ulp is the largest distance in ULPs (±0 equal, NaN only against NaN, 0xFFFFFFFF if one side is NaN),
 * *out_worst_bits (optional) a pattern that attains it.  ops: SQRT CBRT EXP EXP2 LOG LOG2 SIN COS ACOS SINH.  Blocking.
 * Not in the reference, whose tests pin these functions to 0.01 absolute on a handful of points
 * [ref: crates/trigonometry/src/f32_kernel.rs:62-132, crates/math/src/f32.rs:84-271]. */
agpu_status agpu_selftest_unary_f32(agpu_pipeline* p, agpu_unary_op op, uint64_t first_bits, uint64_t count,
                                    uint32_t* out_max_ulp, uint32_t* out_worst_bits);
/* The same for f32 pow over `count` reproducible pseudo-random operand pairs (2^64 pairs cannot be enumerated):
 * domain 0 = x any positive bit pattern (denormals, inf, NaN), |y| < 2^8; 1 = x within 2^13 ULPs of 1, |y| up to 2^30;
 * 2 = x in [2^-3, 2^3), |y| < 2^7 (results across overflow / underflow).  2^32 pairs take about a second. */
agpu_status agpu_selftest_pow_f32(agpu_pipeline* p, uint64_t seed, uint64_t count, int32_t domain, uint32_t* out_max_ulp,
                                  uint32_t* out_worst_x_bits, uint32_t* out_worst_y_bits);

/* out[i] = (to)in[i].  Table = cast_dyn's [ref: crates/cast/src/lib.rs:135-161] plus identity-width sign
 * reinterprets (memcpy in the reference :69-86).  from=AGPU_BOOL,to=F32: `in` is a bitmap of n bits.
 * f32→u8: trunc toward 0, clamp to [0,2^32-1], then mod 256; NaN→0 [ref: crates/cast/compute_shaders/f32/cast_u8.wgsl].
 * f32→i8 / i16 / u16 / i32 / u32 / date32: REFERENCE-ABSENT (the reference implements f32→u8 only; north_star asks for
 * "i8/i16/u8/u16 <-> f32").  Defined by analogy with cast_u8.wgsl: WGSL's conversion to the 32-bit integer of the
 * target's signedness (u32(x) / i32(x): truncate toward 0, clamp to that range, NaN → 0), then the low bits of the
 * target width.  u32/i32 → f32 stay unsupported, as in the reference's table. */
agpu_status agpu_cast(agpu_pipeline* p, agpu_dtype from, agpu_dtype to, const void* in, void* out, uint64_t n);

/* out[i] = value (n elements of dtype); `value_bits` holds the scalar's little-endian bytes in its low bits.
 * [ref: apply_broadcast_function compute_pipeline.rs:215-256; crates/array/compute_shaders/{f32,i32,u32}/broadcast.wgsl]
 * dtype=AGPU_BOOL fills a bitmap of n bits (padding zero) [ref: boolean_gpu.rs:173-194]. */
agpu_status agpu_broadcast(agpu_pipeline* p, agpu_dtype dtype, uint32_t value_bits, void* out, uint64_t n);
/* Same, with the scalar read from a 1-element DEVICE buffer — the literal shape of the reference call
 * apply_broadcast_function(&scalar_buffer, ...) [ref: crates/array/src/array/f32_gpu.rs:14-37]; not for AGPU_BOOL. */
agpu_status agpu_broadcast_from_device(agpu_pipeline* p, agpu_dtype dtype, const void* scalar_dev, void* out, uint64_t n);

/* ---------------------------------------------------------------- fused element-wise chains
 * The reference's `*_op(&mut pipeline)` API lets callers record op chains ((a + s) * s in examples/simple.rs:45-72)
 * but still runs one dispatch — one full pass over HBM — per op.  agpu_fused_chain evaluates a LINEAR chain
 *     acc = in[i];  for each step: acc = op(acc, operand_i)   (operand = none | 1-element scalar buffer | array[n])
 * in ONE kernel: the column is read once and written once however long the chain is.  Every step applies exactly the
 * same scalar operation, in the same order and with the same rounding, as the stand-alone kernel, so the result is
 * bit-identical to running the ops one by one.  dtype ∈ {F32, I32, U32, DATE32}; all arrays have n rows.  Unary steps: neg abs (all),
 * not (integers), sqrt cbrt exp exp2 log log2 sin cos acos sinh (f32).
 * kind == AGPU_CHAIN_UNARY: `op` is an agpu_unary_op and `operand` is ignored; otherwise `op` is an agpu_binary_op. */
typedef enum { AGPU_CHAIN_UNARY = 0, AGPU_CHAIN_SCALAR = 1, AGPU_CHAIN_ARRAY = 2 } agpu_chain_kind;
typedef struct {
  int32_t op;          /* agpu_unary_op or agpu_binary_op, see kind */
  int32_t kind;        /* agpu_chain_kind */
  const void* operand; /* device pointer: 1 element (SCALAR) or n elements (ARRAY); NULL for UNARY */
} agpu_chain_step;
#define AGPU_CHAIN_MAX_STEPS 8
agpu_status agpu_fused_chain(agpu_pipeline* p, agpu_dtype dtype, const void* in, const agpu_chain_step* steps,
                             int32_t n_steps, void* out, uint64_t n);
/* The same chain ending in a compare — a predicate such as (a * b + c) > d in one pass: the chain's result is never
 * stored, out_bits bit i = chain(in)[i] cmp operand (operand_kind AGPU_CHAIN_SCALAR: 1 element, AGPU_CHAIN_ARRAY: n
 * elements), packed like agpu_compare (agpu_bitmap_bytes(n) bytes, padding bits 0).  n_steps ≤ 7 (0 = plain compare).
 * Bit-identical to agpu_fused_chain followed by agpu_compare [ref: Compare::*_op crates/compare/src/lib.rs:142-162]. */
agpu_status agpu_fused_chain_compare(agpu_pipeline* p, agpu_dtype dtype, const void* in, const agpu_chain_step* steps,
                                     int32_t n_steps, agpu_cmp_op cmp_op, int32_t operand_kind, const void* operand,
                                     void* out_bits, uint64_t n);

/* A chain with a WIDENING CAST at its head: acc = (float)in[i] for a u8 / i8 / u16 / i16 column, then the f32 chain, stored as
 * f32 — `cast → sin`, `cast → mul_scalar → add_scalar` in ONE pass: the narrow column is read once (1–2 B/row) and the result
 * written once; the 4 B/row intermediate of the unfused pair (written by the cast, re-read by the next op) never exists:
 * cast u8→f32 then sin is 5 B/row instead of 13.  Bit-identical to agpu_cast(from → F32) followed by the steps one by one.
 * Steps: f32 unary (neg abs sqrt cbrt exp exp2 log log2 sin cos acos sinh), f32 scalar / array (add sub mul div rem min max;
 * ≤ AGPU_CAST_CHAIN_MAX_ARRAYS array operands, f32 columns of n rows — more is AGPU_ERR_UNSUPPORTED, and a recording host must cut
 * the chain there).  `steps` must not be NULL when n_steps > 0, whatever n is.  n_steps == 0: the plain cast.  `cast → sin | cos | sinh` of an 8-bit column runs the
 * reference's own fused kernels [ref: crates/trigonometry/src/u8_kernel.rs:34-38, i8_kernel.rs; the chain API of
 * crates/arrow/examples/simple.rs:45-72]. */
#define AGPU_CAST_CHAIN_MAX_ARRAYS 4
agpu_status agpu_fused_cast_chain(agpu_pipeline* p, agpu_dtype from, const void* in, const agpu_chain_step* steps,
                                  int32_t n_steps, float* out, uint64_t n);

/* ---------------------------------------------------------------- compare → bitmap
 * out_bits bit i = a[i] cmp b[i], LSB-first, agpu_bitmap_bytes(n) bytes written, padding bits 0.
 * [ref: apply_function! crates/compare/src/lib.rs:85-111; crates/compare/compute_shaders/ * /cmp.wgsl
 *  (atomicOr into workgroup words, lanes gid%32==0 store)]. f32: IEEE compare, NaN ⇒ false. */
agpu_status agpu_compare(agpu_pipeline* p, agpu_cmp_op op, agpu_dtype dtype, const void* a, const void* b,
                         void* out_bits, uint64_t n);

/* Fused form of the reference's two steps (compare dispatch + NullBitBufferGpu::merge_null_bit_buffer_op,
 * [ref: crates/compare/src/lib.rs:98-103, null_bit_buffer.rs:206-243]): also writes out_validity = va & vb.
 * va / vb may each be NULL (= all valid): both NULL → out_validity untouched (may be NULL); one NULL → copy. */
agpu_status agpu_compare_validity(agpu_pipeline* p, agpu_cmp_op op, agpu_dtype dtype, const void* a, const void* b,
                                  const void* va, const void* vb, void* out_bits, void* out_validity, uint64_t n);
/* The same, with the NULL COUNT of the result as a by-product: *out_null_count_dev (device u64) = n − popcount of the first n
 * bits of out_validity (0 when both va and vb are NULL).  The waves that store the validity words add v_bcnt of what they
 * store (one u32 per wave in stream scratch, no atomics) and one small block folds them — no second pass over the bitmap.
 * The reference needs that pass [ref: crates/logical/src/boolean.rs:120-146 countob + Sum]; Arrow consumers want the
 * count with every array (ArrowArray.null_count).  out_null_count_dev may be NULL (= agpu_compare_validity). */
agpu_status agpu_compare_validity_count(agpu_pipeline* p, agpu_cmp_op op, agpu_dtype dtype, const void* a, const void* b,
                                        const void* va, const void* vb, void* out_bits, void* out_validity, uint64_t n,
                                        uint64_t* out_null_count_dev);

/* ---------------------------------------------------------------- bitmaps (Boolean data and validity)
 * op ∈ {AND, OR, XOR} word-wise over n_bits.  This is NullBitBufferGpu::merge_null_bit_buffer's kernel
 * [ref: crates/array/src/array/null_bit_buffer.rs:168-204 → crates/logical/compute_shaders/u32/logical.wgsl:13-29]
 * and BooleanArrayGPU's Logical impl [ref: crates/logical/src/boolean.rs:18-75]. Padding bits: op applied as-is. */
agpu_status agpu_bitmap_binary(agpu_pipeline* p, agpu_binary_op op, const void* a, const void* b, void* out,
                               uint64_t n_bits);
/* The same with the number of SET bits among the first n_bits of `out` as a by-product (*out_set_count_dev, device u64; padding
 * bits are not counted whatever they hold): for a validity AND that is the valid count, n_bits − it the null count
 * [ref: NullBitBufferGpu::merge_null_bit_buffer_op null_bit_buffer.rs:206-243; countob + Sum boolean.rs:120-146]. */
agpu_status agpu_bitmap_binary_count(agpu_pipeline* p, agpu_binary_op op, const void* a, const void* b, void* out,
                                     uint64_t n_bits, uint64_t* out_set_count_dev);
/* out = ~in on every whole word covering n_bits (padding flipped too, like the reference) [ref: u32/not.wgsl:9-13] */
agpu_status agpu_bitmap_not(agpu_pipeline* p, const void* in, void* out, uint64_t n_bits);
/* *out_count (device u64) = number of set bits among the first n_bits.  Null counts, `all()`
 * [ref: crates/logical/src/boolean.rs:120-146 countob + Sum].  One-wave blocks over 16 KiB chunks + one folding block (no
 * atomics, no memset); bitmaps that were just produced by agpu_compare_validity_count / agpu_bitmap_*_count need no call. */
agpu_status agpu_bitmap_popcount(agpu_pipeline* p, const void* bits, uint64_t n_bits, uint64_t* out_count_dev);
/* *out_any (device u32) = 1 if any of the first n_bits is set else 0 [ref: boolean.rs:106-118, u32/any.wgsl] */
agpu_status agpu_bitmap_any(agpu_pipeline* p, const void* bits, uint64_t n_bits, uint32_t* out_any_dev);
/* out bits [0, n_bits) = src bits [src_bit_offset, src_bit_offset + n_bits); padding bits of out are 0.  Re-aligns the
 * validity / Boolean bitmap of a SLICED Arrow array (Arrow C Data Interface `offset` ≠ 0) to the word-aligned layout the
 * kernels use.  `src` must be readable up to the 8-byte word holding the last addressed bit. */
agpu_status agpu_bitmap_copy_bits(agpu_pipeline* p, const void* src, uint64_t src_bit_offset, void* out, uint64_t n_bits);
/* merge validity: out = ((va & m) | (vb & ~m)) & vm; va/vb/vm may be NULL (= all ones); all three NULL → AGPU_ERR_ARG.
 * [ref: merge_null_buffers_op crates/routines/src/merge.rs:17-86, u32/merge_null_buffer.wgsl] */
agpu_status agpu_bitmap_merge_validity(agpu_pipeline* p, const void* va, const void* vb, const void* mask,
                                       const void* vmask, void* out, uint64_t n_bits);
agpu_status agpu_bitmap_merge_validity_count(agpu_pipeline* p, const void* va, const void* vb, const void* mask,
                                             const void* vmask, void* out, uint64_t n_bits, uint64_t* out_set_count_dev);

/* ---------------------------------------------------------------- reductions
 * out_dev → 1 element: SUM f32→f32, i32→i32 (wrapping), u32→u32 (wrapping); MIN/MAX → same dtype.
 * f32 SUM reproduces the reference's summation ORDER bit-exactly: adjacent-pair binary tree inside 256-element
 * blocks, then recursively over block sums [ref: crates/arithmetic/compute_shaders/f32/aggregate.wgsl:28-37,
 * aggregate_kernels.rs:26-43]; like the reference it ignores validity when `validity`==NULL.  With a validity
 * bitmap, null slots contribute the identity (0 / +inf / -inf) — an extension the reference lacks.
 * agpu_reduce_f64: SUM of f32 accumulated and returned in f64 (used for the multi-GPU final reduce). */
agpu_status agpu_reduce(agpu_pipeline* p, agpu_reduce_op op, agpu_dtype dtype, const void* in, const void* validity,
                        uint64_t n, void* out_dev);
agpu_status agpu_reduce_sum_f64(agpu_pipeline* p, const float* in, const void* validity, uint64_t n, double* out_dev);
/* Whole-column f32 statistics in ONE pass (no counterpart in the reference, whose only reduction is `sum` [aggregate_kernels.rs:24-51];
 * north_star config 5 wants sum / min / max of one column — three agpu_reduce calls read it three times).  out_dev receives one record:
 * sum = agpu_reduce(SUM) — the reference's tree order —, min / max = agpu_reduce(MIN / MAX) (Arrow's NaN rule), sum_f64 =
 * agpu_reduce_sum_f64, each BIT-IDENTICAL to the separate call on the same column (null-aware like them when `validity` is given: null
 * slots contribute the identities); the column is read once (4 B/row, + its bitmap) when it is 16-byte aligned (the bitmap 4-byte) and has
 * at least 2^20 rows, otherwise the four reductions run one after the other.  8-byte aligned out_dev. */
typedef struct agpu_f32_stats {
  float sum;
  float min;
  float max;
  uint32_t reserved; /* 0 */
  double sum_f64;
} agpu_f32_stats;    /* 24 bytes */
agpu_status agpu_reduce_stats_f32(agpu_pipeline* p, const float* in, const void* validity, uint64_t n, agpu_f32_stats* out_dev);

/* ---------------------------------------------------------------- swizzle: take / put / merge
 * width = bytes per element (1, 2 or 4; the reference implements 4 and Boolean only).
 * Index ranges are checked inside the kernels with the reference's robust-buffer-access outcome (WGSL): an
 * out-of-range read yields 0, an out-of-range write is dropped — and additionally a sticky bit is set in the pipeline,
 * which the next agpu_pipeline_sync reports once as AGPU_ERR_SHAPE.  No pre-pass over the indices, no readback.
 * take: out[i] = values[idx[i]], i < n_idx.
 * [ref: apply_take_op crates/routines/src/take.rs:9-55, 32bit/take.wgsl:13-17]
 * Large takes / puts (from 2^24 - 2^25 rows) are enqueued in SEVERAL forms: a bucketed pipeline that moves the indices to the data
 * (random indices: 1.8 - 3.5x the direct kernels) and the direct kernel behind it.  A locality probe over the index column(s)
 * decides ON THE DEVICE which form does the work — sorted, sequential, clustered or few-valued indices (a take after a filter)
 * stream through the direct kernel, random ones go through the pipeline; the other forms return at once.  A put of 2^26 rows or
 * more has two forms in between: source column local (the scatter of a contiguous or sorted selection) -> a destination-only
 * pipeline, destination column local (the gather into one) -> the take's pipeline storing through the destination column.  The
 * host waits at most 150 us for the probe's answer (an idle stream delivers it in ~25 us) and then enqueues only the chosen form;
 * without the answer every form is enqueued, gated on the device.  The call never blocks beyond that.  Tuning "gather_bucket" = 1 / 2 forces direct / pipelined (no probe), 3 = the round-2 pair pipeline for takes. */
agpu_status agpu_take(agpu_pipeline* p, int32_t width, const void* values, uint64_t n_values, const uint32_t* idx,
                      void* out, uint64_t n_idx);
/* take of the COLUMNS OF ONE TABLE by one index column: outs[c][i] = values[c][idx[i]] for c < n_cols (every column n_values elements of
 * widths[c] ∈ {1, 2, 4} bytes; at most 64 columns).  The reference takes array by array [Swizzle::take_op crates/routines/src/lib.rs:122-143];
 * here everything the merge-back pipeline does to the INDEX column (histogram, scans, partition: a third of a take) runs once for all
 * columns and each column costs its gather + merge passes only — 2^28 random rows: ≈ 2.0 ms per further column instead of 3.0.  Results are
 * those of n_cols agpu_take calls; below the pipeline's sizes, for local indices and for other widths that is what runs. */
agpu_status agpu_take_columns(agpu_pipeline* p, int32_t n_cols, const int32_t* widths, const void* const* values, uint64_t n_values,
                              const uint32_t* idx, void* const* outs, uint64_t n_idx);
/* … with validity bitmaps: validities[c] = column c's bitmap (n_values bits) or NULL; out_validities[c] receives bit i = validities[c] bit
 * idx[i] (agpu_bitmap_bytes(n_idx) bytes, padding bits 0), like agpu_take_validity — the bit travels with the value through the column's
 * gather + merge passes.  validities == NULL: agpu_take_columns. */
agpu_status agpu_take_columns_validity(agpu_pipeline* p, int32_t n_cols, const int32_t* widths, const void* const* values,
                                       const void* const* validities, uint64_t n_values, const uint32_t* idx, void* const* outs,
                                       void* const* out_validities, uint64_t n_idx);
/* take of an array WITH NULLS in one call: out[i] = values[idx[i]] and out_validity bit i = validity bit idx[i] (n_values
 * bits; out_validity: agpu_bitmap_bytes(n_idx) bytes, padding bits 0) [ref: Swizzle::take_op crates/routines/src/lib.rs:122-143
 * = apply_take_op (take.rs:9-55) for the values + take_null_buffer (bool.rs:33-46) for the validity: two dispatches].  For
 * 4-byte values at bucketed sizes the bit travels with the value through the merge-back pipeline (one extra word load per
 * gather, the 16 KiB of bitmap behind a 512 KiB region are L2-resident) instead of a second random pass over the index
 * column; otherwise = agpu_take + agpu_take_bits.  validity == NULL: plain agpu_take, out_validity untouched. */
agpu_status agpu_take_validity(agpu_pipeline* p, int32_t width, const void* values, uint64_t n_values, const void* validity,
                               const uint32_t* idx, void* out, void* out_validity, uint64_t n_idx);
/* out bit i = bits[idx[i]]  [ref: crates/routines/src/bool.rs:15-46, bool/take.wgsl:13-33] — data and validity.  From 2^25
 * rows out of a bitmap of 2^27 … 2^29 bits it runs through the merge-back pipeline with the bitmap's words as the elements
 * (1.8x the direct bit gather at 2^28 rows); tuning gather_bucket = 1 / 2 forces the direct / the pipelined form. */
agpu_status agpu_take_bits(agpu_pipeline* p, const void* bits, uint64_t n_bits, const uint32_t* idx, void* out_bits,
                           uint64_t n_idx);
/* dst[dst_idx[i]] = src[src_idx[i]] in place; duplicate dst_idx ⇒ unspecified winner
 * [ref: apply_put_op crates/routines/src/put.rs:9-56, 32bit/put.wgsl:17-23].  The *_bounded forms take the two array
 * lengths and range-check as described above; agpu_put / agpu_put_bits trust the caller (lengths unknown). */
agpu_status agpu_put_bounded(agpu_pipeline* p, int32_t width, const void* src, uint64_t n_src, const uint32_t* src_idx,
                             void* dst, uint64_t n_dst, const uint32_t* dst_idx, uint64_t n);
agpu_status agpu_put(agpu_pipeline* p, int32_t width, const void* src, const uint32_t* src_idx, void* dst,
                     const uint32_t* dst_idx, uint64_t n);
/* bit scatter: dst bit dst_idx[i] = src bit src_idx[i] [ref: crates/routines/src/bool.rs:48-128, bool/put.wgsl:17-34 — an
 * atomic and / or per row there].  The bounded form runs bucketed by destination region from 2^24 rows (the bits are gathered
 * by the Boolean take's pipeline, partitioned by 32 KiB bitmap regions and applied in LDS: no global atomics, 2.8x the direct
 * kernel at 2^28 rows); below that, with unknown lengths, or under tuning gather_bucket = 1: one atomic per row. */
agpu_status agpu_put_bits_bounded(agpu_pipeline* p, const void* src_bits, uint64_t n_src_bits, const uint32_t* src_idx,
                                  void* dst_bits, uint64_t n_dst_bits, const uint32_t* dst_idx, uint64_t n);
agpu_status agpu_put_bits(agpu_pipeline* p, const void* src_bits, const uint32_t* src_idx, void* dst_bits,
                          const uint32_t* dst_idx, uint64_t n);
/* out[i] = mask bit i ? a[i] : b[i]  [ref: Swizzle::merge_op crates/routines/src/lib.rs:82-120, {32,16,8}bit/merge.wgsl] */
agpu_status agpu_merge(agpu_pipeline* p, int32_t width, const void* a, const void* b, const void* mask_bits, void* out,
                       uint64_t n);
/* Boolean data merge: out = (a & m) | (b & ~m) [ref: bool/merge.wgsl:17-21] */
agpu_status agpu_merge_bits(agpu_pipeline* p, const void* a, const void* b, const void* mask_bits, void* out,
                            uint64_t n_bits);
/* *out_max (device u32) = max(idx[0..n)) — lets a host wrapper reject out-of-range indices (HIP has no robust access) */
agpu_status agpu_index_max(agpu_pipeline* p, const uint32_t* idx, uint64_t n, uint32_t* out_max_dev);

/* ---------------------------------------------------------------- multi-GPU: chunk-sharded columns, RCCL final reduce
 * Not in the reference (single device + single queue [ref: crates/array/src/gpu_utils/gpu_device.rs:29-33]); this is
 * north_star config 5.  One host thread or process per GPU, each with its own agpu_device + pipeline + communicator
 * rank; a column is sharded into contiguous row ranges (cut on multiples of 512 rows: whole bitmap words, 2 KiB-aligned
 * f32 spans); every kernel above runs shard-local with no collective; only whole-column statistics finish over RCCL.
 *   rank 0: agpu_comm_get_unique_id(id) → ship the 128 bytes to the other ranks (pipe / file / torch.distributed …)
 *   all   : agpu_comm_init_rank(dev, id, rank, world, &comm)      — collective, waits until all ranks arrived — at most
 *           AGPU_COMM_TIMEOUT_MS (default 120 000; 0 = for ever), then AGPU_ERR_HIP instead of hanging: exit the process
 *   all   : agpu_comm_reduce(comm, p, op, dtype, shard, validity, n_local, out_dev)   — collective, asynchronous on p
 * agpu_comm_reduce = shard-local agpu_reduce + an all-gather of ONE 16-byte record per rank + a single-workgroup
 * combine IN RANK ORDER on every rank (identical result everywhere, independent of RCCL's ring order):
 *   SUM f32  : the shard sums are combined by the reference's own adjacent-pair tree (zero-padded to 256), so shards of
 *              256^k rows reproduce the reference's whole-column tree bit for bit [ref: aggregate.wgsl:28-37];
 *   SUM ints : wrapping; MIN/MAX: Arrow semantics (NaN ignored unless all NaN); empty shards contribute the identity.
 * One statistic in flight per communicator: calls on one pipeline are stream-ordered, a call made on ANOTHER pipeline is
 * ordered behind the previous one by an event (the record buffers are shared), and a mutex serialises host threads —
 * every rank must still issue its collectives in the same order, as with any RCCL communicator. */
#define AGPU_COMM_ID_BYTES 128
typedef struct agpu_comm agpu_comm;
typedef enum { AGPU_COMM_F32 = 0, AGPU_COMM_F64 = 1, AGPU_COMM_I32 = 2, AGPU_COMM_U32 = 3, AGPU_COMM_I64 = 4, AGPU_COMM_U64 = 5 } agpu_comm_dtype;
agpu_status agpu_comm_get_unique_id(void* out_id /* AGPU_COMM_ID_BYTES */);
agpu_status agpu_comm_init_rank(agpu_device* dev, const void* unique_id, int32_t rank, int32_t world, agpu_comm** out_comm);
/* the same with an explicit deadline (timeout_ms <= 0: wait for ever).  On a timeout ncclCommInitRank is still pending on a
 * helper thread that cannot be cancelled — report, and exit the process (a fresh process is the only clean retry). */
agpu_status agpu_comm_init_rank_timeout(agpu_device* dev, const void* unique_id, int32_t rank, int32_t world,
                                        int64_t timeout_ms, agpu_comm** out_comm);
/* "rccl <version> (built against <version>) from <path of the loaded librccl>; hip runtime <version> from <path of the
 * loaded libamdhip64>" — a process that imported torch first runs on torch's bundled copies, any other on /opt/rocm's;
 * multi-GPU records print it so that they say which runtime they measured. */
agpu_status agpu_comm_runtime_info(char* out, size_t out_cap);
agpu_status agpu_comm_destroy(agpu_comm* c);
agpu_status agpu_comm_rank(agpu_comm* c, int32_t* out_rank, int32_t* out_world);
agpu_status agpu_comm_reduce(agpu_comm* c, agpu_pipeline* p, agpu_reduce_op op, agpu_dtype dtype, const void* in,
                             const void* validity, uint64_t n_local, void* out_dev);
/* f32 column summed in f64 per shard, shard sums added in rank order in f64 (order-robust statistic for huge columns) */
agpu_status agpu_comm_reduce_sum_f64(agpu_comm* c, agpu_pipeline* p, const float* in, const void* validity,
                                     uint64_t n_local, double* out_dev);
/* agpu_reduce_stats_f32 over this rank's shard + the four final reduces (one 16-byte record per rank and statistic, combined in rank
 * order): every field equals what agpu_comm_reduce / agpu_comm_reduce_sum_f64 give for that statistic; the shard is read ONCE. */
agpu_status agpu_comm_reduce_stats_f32(agpu_comm* c, agpu_pipeline* p, const float* in, const void* validity, uint64_t n_local,
                                       agpu_f32_stats* out_dev);
/* the same final reduce for a per-shard statistic the caller already holds on the device (1 element of dtype at
 * partial_dev; kind_f64 != 0: an f64 sum) */
agpu_status agpu_comm_final_reduce(agpu_comm* c, agpu_pipeline* p, agpu_reduce_op op, agpu_dtype dtype, int32_t kind_f64,
                                   const void* partial_dev, uint64_t n_local, void* out_dev);
/* The combine step on its own, for hosts that move the records by other means (MPI, a file, another collective):
 * records_dev = `world` 16-byte records {statistic in the low 4 bytes (8 for an f64 sum), u64 n_local} in RANK ORDER on
 * this device → out_dev (1 element).  Exactly what agpu_comm_reduce runs after its all-gather; no communicator needed. */
agpu_status agpu_reduce_combine(agpu_pipeline* p, agpu_reduce_op op, agpu_dtype dtype, int32_t kind_f64,
                                const void* records_dev, int32_t world, void* out_dev);
/* plain in-place ncclAllReduce (null counts, row counts: integer statistics whose order cannot matter) */
agpu_status agpu_comm_all_reduce(agpu_comm* c, agpu_pipeline* p, agpu_reduce_op op, agpu_comm_dtype ctype, void* buf_dev,
                                 uint64_t count);
/* collective + host wait on p's stream; gives up with AGPU_ERR_HIP after AGPU_COMM_TIMEOUT_MS when a peer never joins */
agpu_status agpu_comm_barrier(agpu_comm* c, agpu_pipeline* p);
/* The host wait that belongs behind agpu_comm_reduce / _all_reduce / _final_reduce (instead of agpu_pipeline_sync or a
 * download, which would block for ever behind a collective a dead peer never joins): waits for p's stream, gives up after
 * AGPU_COMM_TIMEOUT_MS.  The communicator is NOT held while the host polls: other threads may keep enqueueing on it.
 * A wait that gives up while no collective of this communicator is in flight (the stream was only behind long ordinary work)
 * returns AGPU_ERR_HIP and leaves the device alone — wait again.
 * POISONING: when this wait, agpu_comm_barrier, agpu_comm_peers or agpu_comm_init_rank(_timeout) gives up, a collective
 * nobody will join is still queued on the device.  The device is then marked poisoned: every pipeline call, agpu_malloc,
 * agpu_device_sync / _trim return AGPU_ERR_HIP at once; agpu_free, agpu_pipeline_destroy and agpu_device_destroy return
 * AGPU_OK without waiting (they leak); agpu_comm_destroy aborts the communicator (ncclCommAbort, bounded to 2 s) instead of
 * draining it — so a host unwinding through its destructors ends instead of hanging.  Exit the process afterwards. */
agpu_status agpu_comm_sync(agpu_comm* c, agpu_pipeline* p);
/* What RCCL itself reports for this communicator — ncclCommCount, ncclCommUserRank, ncclCommCuDevice — as opposed to what
 * the caller passed to init (agpu_comm_rank).  Any out pointer may be NULL. */
agpu_status agpu_comm_size(agpu_comm* c, int32_t* out_count, int32_t* out_user_rank, int32_t* out_device);
/* A communicator of ONE rank whose RCCL bootstrap did not come up within the init deadline is created as a LOCAL communicator instead of
 * failing (a single rank waits for nobody; what can hang is RCCL's own socket bootstrap): no ncclComm_t behind it, its collectives are
 * the device copies they amount to at world 1, agpu_comm_size reports {1, 0, this device}.  *out_local = 1 for such a communicator — a
 * record that claims "RCCL ran" must check it.  World > 1 never falls back: a missing peer is an error and poisons the device. */
agpu_status agpu_comm_is_local(agpu_comm* c, int32_t* out_local);
/* One identity record per rank, so that a multi-GPU record can PROVE which devices joined one communicator. */
typedef struct agpu_comm_peer {
  int32_t rank;           /* ncclCommUserRank on the rank the record came from (-1 from agpu_device_identity) */
  int32_t world;          /* ncclCommCount as that rank sees it */
  int32_t device_ordinal; /* HIP ordinal inside that rank's process (after HIP_VISIBLE_DEVICES) */
  int32_t nccl_device;    /* ncclCommCuDevice */
  int32_t pci_domain, pci_bus, pci_device; /* hipDeviceProp_t: the physical address of the GPU */
  int32_t pid;
  uint64_t host_hash;     /* FNV-1a 64 of hostname + kernel boot id: equal on one node */
  uint8_t uuid[16];       /* hipDeviceProp_t.uuid */
  char gcn_arch[16];      /* "gfx950…", truncated */
} agpu_comm_peer;         /* 72 bytes */
agpu_status agpu_device_identity(agpu_device* dev, agpu_comm_peer* out);
/* collective: all-gather of the records THROUGH the communicator + host wait with the deadline.  out_host[r] = rank r's
 * record (cap ≥ ncclCommCount, which must equal the world given to init: AGPU_ERR_SHAPE otherwise); the gather buffer belongs to
 * the communicator (no allocation per call; concurrent callers take turns); *out_distinct (may be NULL) = number of distinct (host_hash, PCI address) among them — equal to
 * world exactly when every rank drives its own GPU. */
agpu_status agpu_comm_peers(agpu_comm* c, agpu_pipeline* p, agpu_comm_peer* out_host, int32_t cap, int32_t* out_distinct);

/* ---------------------------------------------------------------- Arrow C Data Interface
 * The reference builds arrays from host Vecs and reads them back as Vecs
 * [ref: PrimitiveArrayGpu::from_slice / from_optional_slice / raw_values / values,
 *  crates/array/src/array/primitive_array_gpu.rs:22-104].  Here any producer of the Arrow C Data Interface (arrow-rs
 * `FFI_ArrowArray`, Arrow C++, pyarrow `_export_to_c`) hands its buffers over as they are.  The two structs are the ones
 * of the Arrow specification (https://arrow.apache.org/docs/format/CDataInterface.html), declared here so the header
 * has no dependency. */
#ifndef ARROW_C_DATA_INTERFACE
#define ARROW_C_DATA_INTERFACE
#define ARROW_FLAG_DICTIONARY_ORDERED 1
#define ARROW_FLAG_NULLABLE 2
#define ARROW_FLAG_MAP_KEYS_SORTED 4
struct ArrowSchema {
  const char* format;
  const char* name;
  const char* metadata;
  int64_t flags;
  int64_t n_children;
  struct ArrowSchema** children;
  struct ArrowSchema* dictionary;
  void (*release)(struct ArrowSchema*);
  void* private_data;
};
struct ArrowArray {
  int64_t length;
  int64_t null_count;
  int64_t offset;
  int64_t n_buffers;
  int64_t n_children;
  const void** buffers;
  struct ArrowArray** children;
  struct ArrowArray* dictionary;
  void (*release)(struct ArrowArray*);
  void* private_data;
};
#endif
#ifndef ARROW_C_STREAM_INTERFACE
#define ARROW_C_STREAM_INTERFACE
struct ArrowArrayStream {
  int (*get_schema)(struct ArrowArrayStream*, struct ArrowSchema* out);
  int (*get_next)(struct ArrowArrayStream*, struct ArrowArray* out);
  const char* (*get_last_error)(struct ArrowArrayStream*);
  void (*release)(struct ArrowArrayStream*);
  void* private_data;
};
#endif
/* A column resident in HBM: what PrimitiveArrayGpu<T> / BooleanArrayGPU hold
 * [ref: primitive_array_gpu.rs:12-20 {data, gpu_device, len, null_buffer}; boolean_gpu.rs:15-22]. */
typedef struct {
  agpu_dtype dtype;
  uint64_t length;
  int64_t null_count;    /* -1 = not computed */
  void* values;          /* device (agpu_malloc); for AGPU_BOOL a bitmap of `length` bits, word-aligned, padding 0 */
  void* validity;        /* device bitmap (bit set = valid), word-aligned, padding 0 — or NULL: no nulls */
  uint64_t values_bytes; /* allocation sizes */
  uint64_t validity_bytes;
} agpu_arrow_column;
/* format ∈ {"c","C","s","S","i","I","f","b","tdD"} (i8,u8,i16,u16,i32,u32,f32,bool,date32); anything else →
 * AGPU_ERR_UNSUPPORTED (the reference has no other array types [ref: crates/array/src/array/mod.rs:40-50]).
 * `offset` (sliced arrays) is honoured: values are copied from `offset` on, bitmaps are re-aligned on the GPU
 * (agpu_bitmap_copy_bits).  null_count == 0 or no validity buffer ⇒ out->validity = NULL.
 * Host→HBM movement: tuning "h2d_mode" = 1 pageable hipMemcpy (default: 56 GB/s measured, the link's rate), 2 = page-locked
 * 4 MiB chunks filled by up to eight host threads while earlier chunks are on the link (45 GB/s), 3 = hipHostRegister
 * in place (57 GB/s).  The device-side work is ordered on p's stream and the call returns as soon as the SOURCE has been
 * read completely — the caller may release `array` immediately and keeps ownership of it (this call never calls
 * array->release). */
agpu_status agpu_import_arrow(agpu_pipeline* p, const struct ArrowArray* array, const struct ArrowSchema* schema,
                              agpu_arrow_column* out_column);
/* The columns of one record batch at once: like n calls of agpu_import_arrow, but all device buffers come out of ONE
 * block laid out by agpu_malloc_table (value buffers in column order, validity bitmaps behind them), so kernels that read
 * several of the columns together find them in different HBM hash classes.  Each out_columns[k] is freed as usual
 * (agpu_arrow_column_free / agpu_free of its two pointers, any order). */
agpu_status agpu_import_arrow_table(agpu_pipeline* p, int32_t n_columns, const struct ArrowArray* const* arrays,
                                    const struct ArrowSchema* const* schemas, agpu_arrow_column* out_columns);
/* Arrow C Stream Interface (arrow-rs FFI_ArrowArrayStream, pyarrow RecordBatchReader._export_to_c): pull the NEXT record
 * batch of `stream` and import the columns listed in `columns` (indices into the stream's schema; all must have a GPU
 * array type) as one table-placed block (agpu_import_arrow_table).  *out_rows = the batch's row count, or −1 with
 * nothing imported when the stream has ended.  The stream stays the caller's (this call never releases it); the batch
 * pulled from it is released before returning. */
agpu_status agpu_import_arrow_stream_next(agpu_pipeline* p, struct ArrowArrayStream* stream, const int32_t* columns,
                                          int32_t n_columns, agpu_arrow_column* out_columns, int64_t* out_rows);
/* Device column → freshly allocated host buffers behind a released-by-consumer ArrowArray/ArrowSchema pair (both
 * `release` callbacks free everything this call allocated).  Blocks until the data has arrived. */
agpu_status agpu_export_arrow(agpu_pipeline* p, const agpu_arrow_column* column, struct ArrowArray* out_array,
                              struct ArrowSchema* out_schema);
agpu_status agpu_arrow_column_free(agpu_device* dev, agpu_arrow_column* column); /* agpu_free of both buffers */
/* The staging engine on its own: copy `bytes` between pageable host memory and HBM through the same path
 * (to_device != 0: host → device).  Returns when the host side is complete (H2D: source consumed; D2H: data arrived). */
agpu_status agpu_staged_copy(agpu_pipeline* p, void* dev_ptr, void* host_ptr, size_t bytes, int32_t to_device);

/* ---------------------------------------------------------------- Arrow IPC: streaming format and file format
 * Not in the reference (arrays exist only as host Vecs or wgpu buffers [ref: primitive_array_gpu.rs:22-104]); files and
 * sockets carry Arrow IPC (arrow-rs `arrow::ipc`, pyarrow `pa.ipc`), so columns can come from and go to that format
 * without a host-side Arrow library.  The reader BORROWS `data` (mmap the file: bytes go page cache → HBM with no copy in
 * between) — keep it mapped until agpu_ipc_close.  Little-endian V4/V5 metadata; bodies uncompressed or LZ4_FRAME-compressed
 * (Feather V2's default: decoded by the library, the view then owns the decompressed bytes); columns of the nine GPU array
 * types are readable — also when dictionary-encoded (indices int8 / 16 / 32, dictionaries without nulls; deltas and
 * replacements followed): agpu_ipc_column_view decodes on the host, agpu_ipc_read_column on the GPU (indices + dictionary
 * cross the link, agpu_take gathers) —; columns of any other type (utf8, int64, nested, string dictionaries …) are skipped correctly and report
 * AGPU_ERR_UNSUPPORTED when asked for; ZSTD-compressed bodies → AGPU_ERR_UNSUPPORTED; malformed or truncated input →
 * AGPU_ERR_SHAPE (every metadata access is bounds-checked).  Host-only calls (open … column_view, writer_create,
 * write_batch, finish) need no GPU. */
typedef struct agpu_ipc_reader agpu_ipc_reader;
typedef struct agpu_ipc_writer agpu_ipc_writer;
typedef struct {
  const char* name;   /* reader: owned by the reader */
  const char* format; /* Arrow C Data Interface format string ("f", "i", "b", "tdD", "u" …); "" when there is none; ignored by the writer */
  int32_t dtype;      /* agpu_dtype, or -1: no GPU array type */
  int32_t nullable;
} agpu_ipc_field;
agpu_status agpu_ipc_open(const void* data, uint64_t bytes, agpu_ipc_reader** out_reader);
void agpu_ipc_close(agpu_ipc_reader* r);
agpu_status agpu_ipc_num_fields(const agpu_ipc_reader* r, int32_t* out_n);
agpu_status agpu_ipc_field_info(const agpu_ipc_reader* r, int32_t i, agpu_ipc_field* out);
agpu_status agpu_ipc_num_batches(const agpu_ipc_reader* r, int64_t* out_n);
agpu_status agpu_ipc_batch_rows(const agpu_ipc_reader* r, int64_t batch, int64_t* out_rows);
/* column `column` of record batch `batch` as an ArrowArray / ArrowSchema pair whose buffers POINT INTO `data` (the
 * release callbacks free only the two small structs' private parts) */
agpu_status agpu_ipc_column_view(const agpu_ipc_reader* r, int64_t batch, int32_t column, struct ArrowArray* out_array,
                                 struct ArrowSchema* out_schema);
/* = agpu_ipc_column_view + agpu_import_arrow: the column in HBM, ordered on p's stream */
agpu_status agpu_ipc_read_column(const agpu_ipc_reader* r, int64_t batch, int32_t column, agpu_pipeline* p,
                                 agpu_arrow_column* out_column);
/* several columns of one record batch → one table-placed block (agpu_import_arrow_table) */
agpu_status agpu_ipc_read_batch(const agpu_ipc_reader* r, int64_t batch, const int32_t* columns, int32_t n_columns,
                                agpu_pipeline* p, agpu_arrow_column* out_columns);
/* Writer: fd ≥ 0 → bytes are written to that descriptor as they are produced (the caller opens and closes it);
 * fd < 0 → bytes accumulate in memory and agpu_ipc_writer_finish hands out the buffer (valid until destroy).
 * file_format != 0 → "ARROW1" file with footer; else the streaming format.  Body buffers are padded to 64 bytes. */
agpu_status agpu_ipc_writer_create(const agpu_ipc_field* fields, int32_t n_fields, int32_t file_format, int32_t fd,
                                   agpu_ipc_writer** out_writer);
/* codec 0: bodies as they are (default); 1: every buffer of every later batch as an LZ4 frame (BodyCompression LZ4_FRAME,
 * what Feather V2 defaults to; incompressible buffers are stored).  Compression runs on the host, one thread. */
agpu_status agpu_ipc_writer_set_compression(agpu_ipc_writer* w, int32_t codec);
/* one record batch from host arrays (one ArrowArray per field, in schema order; `offset` honoured, bitmaps re-packed,
 * null counts recomputed) */
agpu_status agpu_ipc_writer_write_batch(agpu_ipc_writer* w, const struct ArrowArray* const* columns);
/* one record batch from device columns (one agpu_arrow_column per field): HBM → sink; null_count −1 is counted on the GPU */
agpu_status agpu_ipc_writer_write_device_batch(agpu_ipc_writer* w, agpu_pipeline* p, const agpu_arrow_column* columns);
/* end-of-stream marker (+ footer for the file format).  out_data = the in-memory buffer (NULL for an fd sink); out_bytes =
 * total bytes produced */
agpu_status agpu_ipc_writer_finish(agpu_ipc_writer* w, const void** out_data, uint64_t* out_bytes);
void agpu_ipc_writer_destroy(agpu_ipc_writer* w);

/* ---------------------------------------------------------------- reference entry-point names
 * Keeps the reference's kernel identity for a thin shim.  `shader_key` is EITHER
 *   - the reference's own argument: the WGSL TEXT of one of its shader constants, NUL-terminated, exactly as the op crates
 *     build them — `include_str!("…/f32/array.wgsl")` or `concat!(include_str!("…/u8/utils.wgsl"), include_str!("…/u8/cmp.wgsl"))`
 *     [ref: crates/arithmetic/src/f32.rs:10-15, crates/compare/src/u8.rs:3-12; (text, entry point) is the pipeline-cache
 *     key, gpu_device.rs:145-168].  The text is recognised by its 64-bit FNV-1a hash + length in a table of the reference's
 *     76 constants (csrc/shader_hashes.inc, generated by tools/extract_entry_points.py: hashes only, no WGSL is shipped); an
 *     unknown text → AGPU_ERR_UNSUPPORTED.  One text exists under two names: compare/u32/min_max.wgsl is byte-identical to
 *     compare/i32/min_max.wgsl (array<i32>: the reference's u32 min / max compare as SIGNED) and resolves to the i32 kernel
 *     — the program the text is; the typed agpu_binary(MIN / MAX, AGPU_U32) compares unsigned;
 *   - or such a text named by its hash, "#<16 hex digits of FNV-1a 64>:<byte length>" (a shim may hash each constant once);
 *   - or the path key: the WGSL file's path under crates/ without "compute_shaders/" and ".wgsl" (e.g.
 *     "arithmetic/f32/array", "compare/i32/cmp", "logical/u32/logical").
 * entry_point = the @compute fn name ("add_f32", "eq", "bitwise_and", ...).  inputs[] are the read bindings in
 * binding order (for `put`: src, src_indexes, dst_indexes); out is the read_write binding; n = number of OUTPUT
 * elements the dispatch covers (bits for Boolean/bitmap kernels, index count for take/put, INPUT rows for "sum").
 * [ref: GpuDevice::create_compute_pipeline(shader, entry_point) gpu_device.rs:145-168 — (shader, entry) is the cache key] */
agpu_status agpu_launch_by_name(agpu_pipeline* p, const char* shader_key, const char* entry_point,
                                const void* const* inputs, int32_t n_inputs, void* out, uint64_t n);
/* The lookup on its own (host only, no GPU): path key of a shader text / of its (FNV-1a 64, byte length); out_cap ≥ 64.
 * A shim may resolve each `const *_SHADER` once and cache the key. */
agpu_status agpu_shader_key_for_source(const char* wgsl, size_t len, char* out_key, size_t out_cap);
agpu_status agpu_shader_key_for_hash(uint64_t fnv1a64_of_text, uint64_t text_bytes, char* out_key, size_t out_cap);
/* The reference's LITERAL call: apply_{unary,binary,scalar,ternary,broadcast}_function(buffers…, new_buffer_size, shader,
 * entry_point, dispatch_size) [ref: compute_pipeline.rs:24-66 (unary), 68-113 (binary), 115-165 (ternary), 167-213
 * (scalar), 215-256 (broadcast); the immediate forms gpu_device.rs:267-509; routines::apply_take_op take.rs:9-55,
 * apply_put_op put.rs:9-56; cast::apply_boolean_unary_function boolean_cast.rs:8-55] carries no element count — a
 * shim that keeps every call site unchanged only has buffers with their BYTE sizes and the dispatch size.  This form
 * takes exactly that: inputs[k] / input_bytes[k] = the read bindings in binding order (put: src, src_indexes,
 * dst_indexes), out / out_bytes = the read_write binding (the buffer the shim allocated with new_buffer_size; put: dst),
 * dispatch_size = workgroups of 256.  The element count is derived as the WGSL sees it: every lane that lies inside all
 * of its bindings and inside dispatch_size × 256 invocations is processed — including the padding lanes of sub-word
 * columns (a 5-element u8 column is 2 words = 8 lanes) — lanes outside are skipped (robust buffer access leaves them
 * unspecified).  "sum" runs ONE 256-ary level per call, like the shader (the loop over levels stays in Sum::sum_op).
 * All byte sizes are multiples of 4 (wgpu's COPY_BUFFER_ALIGNMENT); every buffer must come from agpu_malloc (blocks are
 * padded to 16 bytes, which the word-granular bitmap kernels rely on).  All 148 live (shader, entry point) pairs of the
 * reference are accepted (tests/golden/reference_entry_points.json, tests/test_gpu_by_name.py); the 5 dead ones — WGSL
 * files no Rust source includes — return AGPU_ERR_UNSUPPORTED. */
agpu_status agpu_launch_by_name_sized(agpu_pipeline* p, const char* shader_key, const char* entry_point,
                                      const void* const* inputs, const uint64_t* input_bytes, int32_t n_inputs,
                                      void* out, uint64_t out_bytes, uint32_t dispatch_size);

/* ---------------------------------------------------------------- synthetic columns (bench / parity inputs)
 * Counter-based: element i of a column depends only on (seed, row0+i), so any shard of a column can be generated
 * independently on any GPU and re-generated bit-identically by the CPU oracle (oracle/agpu_oracle.c: synth_*).
 *   h = splitmix64(seed ^ (row * 0x9E3779B97F4A7C15))
 *   f32 : lo + (hi-lo) * ((h >> 40) * 2^-24)            (24-bit uniform, exact in f32 arithmetic order as written)
 *   i32 : (int32)(h >> 32) mod `modulus` (modulus>0) else the raw 32 bits
 *   u8  : h >> 56
 *   bits: bit = ((h >> 11) * 2^-53) < p_set                (Bernoulli validity / Boolean data) */
agpu_status agpu_synth_f32(agpu_pipeline* p, float* out, uint64_t n, uint64_t seed, uint64_t row0, float lo, float hi);
agpu_status agpu_synth_i32(agpu_pipeline* p, int32_t* out, uint64_t n, uint64_t seed, uint64_t row0, uint32_t modulus);
agpu_status agpu_synth_u8(agpu_pipeline* p, uint8_t* out, uint64_t n, uint64_t seed, uint64_t row0);
agpu_status agpu_synth_bits(agpu_pipeline* p, void* out_bits, uint64_t n_bits, uint64_t seed, uint64_t row0,
                            double p_set);
/* Order-independent 64-bit checksum of a byte range (sum of splitmix64(word ^ index)) for full-size parity checks. */
agpu_status agpu_checksum(agpu_pipeline* p, const void* data, uint64_t bytes, uint64_t* out_sum_dev);

#ifdef __cplusplus
}
#endif
#endif /* ARROW_GPU_H */
