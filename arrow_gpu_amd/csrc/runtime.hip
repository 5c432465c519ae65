// runtime.hip — device / buffer / pipeline / event / graph half of the C ABI (include/arrow_gpu.h).
// Replaces arrow_gpu_array::gpu_utils::{GpuDevice, ArrowComputePipeline, CmpQuery}
// [ref: crates/array/src/gpu_utils/gpu_device.rs:29-514, compute_pipeline.rs:8-300, compute_query.rs:7-90].
// A pipeline is a HIP stream: launches are eager and ordered, `finish` is the (already satisfied) submit point.
#include "common.hpp"

static thread_local char g_err[512] = "";

void agpu_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

agpu_tuning g_tune = {
    /*stream_grid*/ 0, /*stream_bpc*/ 0, /*stream_unroll*/ 1, /*stream_nt*/ 1, /*cmp_variant*/ 0, /*reduce_grid*/ 0,
    /*table_tiles*/ 4,  // profiles/r01_sweep_table_tiles.json: sin_u8 5.74 → 6.02, sin_u16 5.85 → 6.16 TB/s vs 1 tile per block
    /*mem_pool*/ 1};

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
  agpu_device* d = new agpu_device();
  d->ordinal = ordinal;
  AGPU_HIP(hipSetDevice(ordinal));
  AGPU_HIP(hipGetDeviceProperties(&d->props, ordinal));
  d->num_cus = d->props.multiProcessorCount;
  if (strncmp(d->props.gcnArchName, "gfx950", 6) != 0) {
    agpu_set_error("device %d is %s; this library is built for gfx950 only", ordinal, d->props.gcnArchName);
    delete d;
    return AGPU_ERR_NO_DEVICE;
  }
  d->trig16_table = nullptr;
  hipError_t me = hipMalloc(&d->trig16_table, AGPU_TABLE_BYTES);
  d->pow_table = me == hipSuccess ? static_cast<char*>(d->trig16_table) + 512 * 16 : nullptr;
  agpu_status ts = me == hipSuccess ? agpu_internal_build_tables(d->trig16_table, d->pow_table) : AGPU_ERR_HIP;
  if (ts != AGPU_OK) {
    if (me != hipSuccess) agpu_set_error("hipMalloc of the function tables failed: %s", hipGetErrorString(me));
    if (d->trig16_table) (void)hipFree(d->trig16_table);
    delete d;
    return ts;
  }
  d->cache_cap = d->props.totalGlobalMem / 2;  // cached (idle) blocks never hold more than half of HBM
  *out_device = d;
  return AGPU_OK;
}

// release every cached block and the scratch of idle streams; blocks until the device is idle
static void device_trim_locked(agpu_device* dev) {
  (void)hipDeviceSynchronize();
  for (auto& kv : dev->cache) {
    for (hipEvent_t e : kv.second.pending) dev->event_pool.push_back(e);
    (void)hipFree(kv.second.ptr);
  }
  dev->cache.clear();
  dev->cached_bytes = 0;
  for (auto& s : dev->idle_streams) {
    if (s.scratch) (void)hipFree(s.scratch);
    s.scratch = nullptr;
    s.scratch_bytes = 0;
  }
}

agpu_status agpu_device_trim(agpu_device* dev) {
  AGPU_REQUIRE(dev, AGPU_ERR_ARG, "null device");
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
  if (out_idle_streams) *out_idle_streams = dev->idle_streams.size();
  return AGPU_OK;
}

agpu_status agpu_device_destroy(agpu_device* dev) {
  if (!dev) return AGPU_OK;
  (void)hipSetDevice(dev->ordinal);
  {
    std::lock_guard<std::mutex> lock(dev->mu);
    device_trim_locked(dev);
    for (auto& s : dev->idle_streams) {
      (void)hipStreamDestroy(s.stream);
      if (s.flags) (void)hipHostFree(s.flags);
    }
    dev->idle_streams.clear();
    for (hipEvent_t e : dev->event_pool) (void)hipEventDestroy(e);
    dev->event_pool.clear();
  }
  if (dev->trig16_table) (void)hipFree(dev->trig16_table);
  delete dev;
  return AGPU_OK;
}

agpu_status agpu_device_sync(agpu_device* dev) {
  AGPU_REQUIRE(dev, AGPU_ERR_ARG, "null device");
  AGPU_HIP(hipSetDevice(dev->ordinal));
  AGPU_HIP(hipDeviceSynchronize());
  return AGPU_OK;
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
// Blocks ≥ 1 MiB are rounded up to 2 MiB multiples and recycled through dev->cache.  A freed block may still be read or
// written by work queued on some stream (the host layers drop their references when a pipeline is released, not when
// the GPU is done; hipFree used to cover that with its implicit device sync), so agpu_free records one event per stream
// and agpu_malloc waits for them before handing the block out again — normally they completed long ago.
#define AGPU_POOL_MIN_BYTES ((size_t)1 << 20)
#define AGPU_POOL_GRANULE ((size_t)2 << 20)

agpu_status agpu_malloc(agpu_device* dev, size_t bytes, int32_t zero_fill, void** out_ptr) {
  AGPU_REQUIRE(dev && out_ptr, AGPU_ERR_ARG, "null argument");
  AGPU_HIP(hipSetDevice(dev->ordinal));
  // pad to 16 B so vector tails of sub-word columns and bitmap words are always addressable
  size_t padded = (bytes + 15) & ~(size_t)15;
  if (padded == 0) padded = 16;
  const bool pooled = g_tune.mem_pool != 0 && padded >= AGPU_POOL_MIN_BYTES;
  if (pooled) padded = (padded + AGPU_POOL_GRANULE - 1) / AGPU_POOL_GRANULE * AGPU_POOL_GRANULE;
  void* p = nullptr;
  if (pooled) {
    std::vector<hipEvent_t> pending;
    {
      std::lock_guard<std::mutex> lock(dev->mu);
      auto it = dev->cache.lower_bound(padded);
      if (it != dev->cache.end() && it->first <= padded + padded / 8) {  // accept up to 12.5 % slack
        p = it->second.ptr;
        pending = std::move(it->second.pending);
        dev->cached_bytes -= it->first;
        dev->block_size[p] = it->first;
        dev->cache.erase(it);
      }
    }
    if (p) {
      hipError_t e = hipSuccess;
      for (hipEvent_t ev : pending)
        if (e == hipSuccess) e = hipEventSynchronize(ev);
      {
        std::lock_guard<std::mutex> lock(dev->mu);
        for (hipEvent_t ev : pending) dev->event_pool.push_back(ev);
      }
      if (e != hipSuccess) {
        agpu_set_error("hipEventSynchronize failed: %s", hipGetErrorString(e));
        return AGPU_ERR_HIP;
      }
    }
  }
  if (!p) {
    hipError_t e = hipMalloc(&p, padded);
    if (e == hipErrorOutOfMemory) {  // give the cached blocks back and retry once
      (void)hipGetLastError();
      std::lock_guard<std::mutex> lock(dev->mu);
      device_trim_locked(dev);
      e = hipMalloc(&p, padded);
    }
    if (e != hipSuccess) {
      (void)hipGetLastError();
      agpu_set_error("hipMalloc(%zu) failed: %s", padded, hipGetErrorString(e));
      return AGPU_ERR_HIP;
    }
    if (pooled) {
      std::lock_guard<std::mutex> lock(dev->mu);
      dev->block_size[p] = padded;
    }
  }
  if (zero_fill) {
    // hipMemset on device memory runs asynchronously on the NULL stream, and pipelines are non-blocking streams that
    // do not order against it: wait, or a later upload could be overwritten by the zero fill
    hipError_t e = hipMemset(p, 0, padded);
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
  AGPU_HIP(hipSetDevice(dev->ordinal));
  {
    std::unique_lock<std::mutex> lock(dev->mu);
    auto it = dev->block_size.find(ptr);
    if (it != dev->block_size.end()) {
      const size_t size = it->second;
      dev->block_size.erase(it);
      if (g_tune.mem_pool != 0 && dev->cached_bytes + size <= dev->cache_cap) {
        agpu_device::CachedBlock blk{ptr, {}};
        bool ok = true;
        for (hipStream_t s : dev->all_streams) {
          hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
          if (hipStreamIsCapturing(s, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) {
            (void)hipGetLastError();  // an event recorded now would become a graph node: do not pool this block
            ok = false;
            break;
          }
          hipEvent_t ev = nullptr;
          if (!dev->event_pool.empty()) {
            ev = dev->event_pool.back();
            dev->event_pool.pop_back();
          } else if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) {
            ok = false;
            break;
          }
          if (hipEventRecord(ev, s) != hipSuccess) {  // e.g. the stream is being captured into a graph
            (void)hipGetLastError();
            dev->event_pool.push_back(ev);
            ok = false;
            break;
          }
          blk.pending.push_back(ev);
        }
        if (ok) {
          dev->cached_bytes += size;
          dev->cache.emplace(size, std::move(blk));
          return AGPU_OK;
        }
        for (hipEvent_t ev : blk.pending) dev->event_pool.push_back(ev);
      }
    }
  }
  AGPU_HIP(hipFree(ptr));
  return AGPU_OK;
}

agpu_status agpu_upload(agpu_pipeline* p, void* dst_dev, const void* src_host, size_t bytes) {
  AGPU_BIND(p);
  if (!bytes) return AGPU_OK;
  AGPU_REQUIRE(dst_dev && src_host, AGPU_ERR_ARG, "null pointer");
  // pageable source: hipMemcpyAsync stages synchronously w.r.t. the host buffer, ordered on the stream
  AGPU_HIP(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, p->stream));
  AGPU_HIP(hipStreamSynchronize(p->stream));
  return AGPU_OK;
}

agpu_status agpu_download(agpu_pipeline* p, void* dst_host, const void* src_dev, size_t bytes) {
  AGPU_BIND(p);
  if (!bytes) {
    AGPU_HIP(hipStreamSynchronize(p->stream));
    return AGPU_OK;
  }
  AGPU_REQUIRE(dst_host && src_dev, AGPU_ERR_ARG, "null pointer");
  AGPU_HIP(hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, p->stream));
  AGPU_HIP(hipStreamSynchronize(p->stream));
  return AGPU_OK;
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
  return AGPU_OK;
}

agpu_status agpu_download_async(agpu_pipeline* p, void* dst_pinned, const void* src_dev, size_t bytes) {
  AGPU_BIND(p);
  if (!bytes) return AGPU_OK;
  AGPU_REQUIRE(dst_pinned && src_dev, AGPU_ERR_ARG, "null pointer");
  AGPU_HIP(hipMemcpyAsync(dst_pinned, src_dev, bytes, hipMemcpyDeviceToHost, p->stream));
  return AGPU_OK;
}

agpu_status agpu_copy(agpu_pipeline* p, void* dst_dev, const void* src_dev, size_t bytes) {
  AGPU_BIND(p);
  if (!bytes) return AGPU_OK;
  AGPU_REQUIRE(dst_dev && src_dev, AGPU_ERR_ARG, "null pointer");
  AGPU_HIP(hipMemcpyAsync(dst_dev, src_dev, bytes, hipMemcpyDeviceToDevice, p->stream));
  return AGPU_OK;
}

agpu_status agpu_memset(agpu_pipeline* p, void* dst_dev, int32_t byte_value, size_t bytes) {
  AGPU_BIND(p);
  if (!bytes) return AGPU_OK;
  AGPU_REQUIRE(dst_dev, AGPU_ERR_ARG, "null pointer");
  AGPU_HIP(hipMemsetAsync(dst_dev, byte_value, bytes, p->stream));
  return AGPU_OK;
}

// ---------------------------------------------------------------- pipeline
static agpu_status pipeline_new(agpu_device* dev, hipStream_t s, bool owns, agpu_pipeline** out) {
  agpu_pipeline* p = new agpu_pipeline();
  p->dev = dev;
  p->stream = s;
  p->owns_stream = owns;
  p->capturing = false;
  p->scratch = nullptr;
  p->scratch_bytes = 0;
  p->flags = nullptr;
  *out = p;
  return AGPU_OK;
}

static agpu_status flags_new(uint32_t** out) {
  void* f = nullptr;
  AGPU_HIP(hipHostMalloc(&f, 64, hipHostMallocDefault));  // pinned + device-visible under unified addressing
  *static_cast<uint32_t*>(f) = 0;
  *out = static_cast<uint32_t*>(f);
  return AGPU_OK;
}

agpu_status agpu_pipeline_create(agpu_device* dev, agpu_pipeline** out_pipeline) {
  AGPU_REQUIRE(dev && out_pipeline, AGPU_ERR_ARG, "null argument");
  AGPU_HIP(hipSetDevice(dev->ordinal));
  agpu_device::StreamSlot slot{nullptr, nullptr, 0, nullptr};
  {
    std::lock_guard<std::mutex> lock(dev->mu);
    if (!dev->idle_streams.empty()) {  // work still queued on a recycled stream simply runs first: same ordering
      slot = dev->idle_streams.back();
      dev->idle_streams.pop_back();
    }
  }
  if (!slot.stream) {
    AGPU_HIP(hipStreamCreateWithFlags(&slot.stream, hipStreamNonBlocking));
    std::lock_guard<std::mutex> lock(dev->mu);
    dev->all_streams.push_back(slot.stream);
  }
  if (!slot.flags) {
    agpu_status fs = flags_new(&slot.flags);
    if (fs != AGPU_OK) return fs;
  }
  *slot.flags = 0;  // a new owner starts clean (its predecessor synchronised or gave up its right to the report)
  agpu_status st = pipeline_new(dev, slot.stream, true, out_pipeline);
  if (st == AGPU_OK) {
    (*out_pipeline)->scratch = slot.scratch;
    (*out_pipeline)->scratch_bytes = slot.scratch_bytes;
    (*out_pipeline)->flags = slot.flags;
  }
  return st;
}

agpu_status agpu_pipeline_wrap_stream(agpu_device* dev, void* hip_stream, agpu_pipeline** out_pipeline) {
  AGPU_REQUIRE(dev && out_pipeline, AGPU_ERR_ARG, "null argument");
  {
    std::lock_guard<std::mutex> lock(dev->mu);
    dev->all_streams.push_back(reinterpret_cast<hipStream_t>(hip_stream));
  }
  agpu_status st = pipeline_new(dev, reinterpret_cast<hipStream_t>(hip_stream), false, out_pipeline);
  if (st == AGPU_OK) st = flags_new(&(*out_pipeline)->flags);
  return st;
}

agpu_status agpu_pipeline_finish(agpu_pipeline* p) {
  AGPU_REQUIRE(p, AGPU_ERR_ARG, "null pipeline");
  return AGPU_OK;  // everything recorded so far is already enqueued in order; like the reference, do not wait
}

agpu_status agpu_pipeline_sync(agpu_pipeline* p) {
  AGPU_BIND(p);
  AGPU_HIP(hipStreamSynchronize(p->stream));
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
  if (p->owns_stream && g_tune.mem_pool != 0 && !p->capturing) {
    // back to the pool WITHOUT waiting: queued work keeps running, the next owner's launches are ordered behind it
    std::lock_guard<std::mutex> lock(dev->mu);
    dev->idle_streams.push_back(agpu_device::StreamSlot{p->stream, p->scratch, p->scratch_bytes, p->flags});
    delete p;
    return AGPU_OK;
  }
  if (p->scratch || p->flags) (void)hipStreamSynchronize(p->stream);
  if (p->scratch) (void)hipFree(p->scratch);
  if (p->flags) (void)hipHostFree(p->flags);
  {
    std::lock_guard<std::mutex> lock(dev->mu);
    for (size_t i = 0; i < dev->all_streams.size(); i++)
      if (dev->all_streams[i] == p->stream) {
        dev->all_streams.erase(dev->all_streams.begin() + (long)i);
        break;
      }
  }
  if (p->owns_stream) (void)hipStreamDestroy(p->stream);
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
  return AGPU_OK;
}

// ---------------------------------------------------------------- graphs
agpu_status agpu_pipeline_begin_capture(agpu_pipeline* p) {
  AGPU_BIND(p);
  AGPU_REQUIRE(!p->capturing, AGPU_ERR_ARG, "already capturing");
  AGPU_HIP(hipStreamBeginCapture(p->stream, hipStreamCaptureModeThreadLocal));
  p->capturing = true;
  return AGPU_OK;
}

agpu_status agpu_pipeline_end_capture(agpu_pipeline* p, agpu_graph** out_graph) {
  AGPU_BIND(p);
  AGPU_REQUIRE(out_graph, AGPU_ERR_ARG, "null out_graph");
  AGPU_REQUIRE(p->capturing, AGPU_ERR_ARG, "not capturing");
  p->capturing = false;
  hipGraph_t g = nullptr;
  AGPU_HIP(hipStreamEndCapture(p->stream, &g));
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
  AGPU_BIND(p);
  AGPU_REQUIRE(e, AGPU_ERR_ARG, "null event");
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

// ---------------------------------------------------------------- tuning
static int64_t* tune_slot(const char* key) {
  if (!key) return nullptr;
  if (!strcmp(key, "stream_grid")) return &g_tune.stream_grid;
  if (!strcmp(key, "stream_bpc")) return &g_tune.stream_bpc;
  if (!strcmp(key, "stream_unroll")) return &g_tune.stream_unroll;
  if (!strcmp(key, "stream_nt")) return &g_tune.stream_nt;
  if (!strcmp(key, "cmp_variant")) return &g_tune.cmp_variant;
  if (!strcmp(key, "reduce_grid")) return &g_tune.reduce_grid;
  if (!strcmp(key, "mem_pool")) return &g_tune.mem_pool;
  if (!strcmp(key, "table_tiles")) return &g_tune.table_tiles;
  return nullptr;
}

agpu_status agpu_set_tuning(const char* key, int64_t value) {
  int64_t* s = tune_slot(key);
  AGPU_REQUIRE(s, AGPU_ERR_ARG, "unknown tuning key");
  *s = value;
  return AGPU_OK;
}

agpu_status agpu_get_tuning(const char* key, int64_t* out_value) {
  int64_t* s = tune_slot(key);
  AGPU_REQUIRE(s && out_value, AGPU_ERR_ARG, "unknown tuning key");
  *out_value = *s;
  return AGPU_OK;
}

}  // extern "C"

agpu_status agpu_scratch(agpu_pipeline* p, size_t bytes, void** out) {
  if (p->scratch_bytes < bytes) {
    AGPU_REQUIRE(!p->capturing, AGPU_ERR_ARG, "scratch growth during graph capture; run the op once before capturing");
    if (p->scratch) {
      AGPU_HIP(hipStreamSynchronize(p->stream));
      AGPU_HIP(hipFree(p->scratch));
      p->scratch = nullptr;
      p->scratch_bytes = 0;
    }
    size_t want = bytes < (1u << 20) ? (1u << 20) : bytes;
    AGPU_HIP(hipMalloc(&p->scratch, want));
    p->scratch_bytes = want;
  }
  *out = p->scratch;
  return AGPU_OK;
}
