// arrow_cdata.hip — Arrow C Data Interface at the C ABI + the host↔HBM staging engine.
//
// The reference moves every array through pageable host Vecs: `from_slice` → `create_gpu_buffer_with_data`
// (wgpu create_buffer_init: a mapped-at-creation copy) and `raw_values` → `retrive_data` (copy to a MAP_READ buffer,
// map, memcpy into a Vec) [ref: crates/array/src/array/primitive_array_gpu.rs:22-104, gpu_device.rs:171-181,232-265].
// Here a producer's ArrowArray/ArrowSchema pair is consumed as it is: values from `offset` on, bitmaps re-aligned on
// the GPU.  Host↔HBM movement has three selectable engines (pageable hipMemcpy — the default, at link rate on this
// platform —, threaded page-locked staging, hipHostRegister in place); overlap of transfers with compute is the host
// layer's job (two pipelines + agpu_pipeline_wait_pipeline: arrow_gpu_amd/interop.py map_chunks).
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>

#include <thread>

#include "common.hpp"

// ---------------------------------------------------------------- staging engine
// mode 1: one hipMemcpy of the pageable range (the runtime stages internally, single-threaded).
// mode 2: T host threads; thread t owns two page-locked 4 MiB slots and moves chunks t, t+T, t+2T, …: memcpy into a
//         slot (host DRAM bandwidth, parallel), hipMemcpyAsync slot → HBM on the pipeline's stream (the DMA engine,
//         overlapped with the next memcpy), event per slot so a slot is refilled only after its DMA has finished.
//         Chunks land in disjoint destinations, so their order on the stream does not matter.
// mode 3: hipHostRegister the caller's range in place, one async DMA, unregister (no CPU copy; pays the pinning).
// Pieces of a slot's size at every transfer size.  1 MiB pieces below 64 MiB (several per thread, so that a thread's DMA runs under its
// memcpy) were measured in round 5 on thread-arena ranges and are WORSE: 8 / 32 MiB up 23 / 27 GB/s (4 MiB pieces: 23 / 36), down 6.4 / 6.5
// (21 / 16) — every piece costs an event wait of ≈ 150 µs on the shared stream, which is what bounds the engine, not the copies.
static size_t stage_chunk_for(size_t bytes) { (void)bytes; return AGPU_STAGE_CHUNK; }
static int stage_threads_for(const agpu_pipeline* p, size_t bytes) {
  (void)p;
  int64_t t = 8;
  const size_t chunks = (bytes + stage_chunk_for(bytes) - 1) / stage_chunk_for(bytes);
  if ((size_t)t > chunks) t = (int64_t)chunks;
  if (t > 32) t = 32;
  if (t < 1) t = 1;
  return (int)t;
}

static agpu_status stage_reserve_locked(agpu_device* dev, size_t slots) {
  while (dev->stage.size() < slots) {
    agpu_device::StageSlot s{nullptr, nullptr, false};
    AGPU_HIP(hipHostMalloc(&s.host, AGPU_STAGE_CHUNK, hipHostMallocDefault));
    hipError_t e = hipEventCreateWithFlags(&s.ev, hipEventDisableTiming);
    if (e != hipSuccess) {
      (void)hipHostFree(s.host);
      agpu_set_error("hipEventCreate failed: %s", hipGetErrorString(e));
      return AGPU_ERR_HIP;
    }
    dev->stage.push_back(s);
  }
  return AGPU_OK;
}

void agpu_internal_free_staging(agpu_device* dev) {
  for (int d = 0; d < 2; d++) {
    std::lock_guard<std::mutex> block(dev->bounce_mu[d]);
    if (dev->bounce[d].host) {
      (void)hipEventDestroy(dev->bounce[d].ev);
      (void)hipHostFree(dev->bounce[d].host);
      dev->bounce[d].host = nullptr;
    }
  }
  std::lock_guard<std::mutex> lock(dev->stage_mu);
  for (auto& s : dev->stage) {
    (void)hipEventSynchronize(s.ev);
    (void)hipEventDestroy(s.ev);
    (void)hipHostFree(s.host);
  }
  dev->stage.clear();
}

static agpu_status staged_copy_threads(agpu_pipeline* p, char* dev_ptr, char* host_ptr, size_t bytes, bool to_device) {
  agpu_device* dev = p->dev;
  const int T = stage_threads_for(p, bytes);
  std::lock_guard<std::mutex> lock(dev->stage_mu);  // the slots serve one transfer at a time
  agpu_status st = stage_reserve_locked(dev, (size_t)T * 2);
  if (st != AGPU_OK) return st;
  const size_t CH = stage_chunk_for(bytes);
  const size_t nchunks = (bytes + CH - 1) / CH;
  std::vector<hipError_t> errs((size_t)T, hipSuccess);
  auto worker = [&](int t) {
    hipError_t e = hipSetDevice(dev->ordinal);
    int k = 0;
    for (size_t c = (size_t)t; c < nchunks && e == hipSuccess; c += (size_t)T, k ^= 1) {
      agpu_device::StageSlot& s = dev->stage[(size_t)t * 2 + (size_t)k];
      const size_t off = c * CH;
      const size_t len = bytes - off < CH ? bytes - off : CH;
      if (to_device) {
        if (s.used) e = hipEventSynchronize(s.ev);  // the DMA that last read this slot
        if (e != hipSuccess) break;
        memcpy(s.host, host_ptr + off, len);
        e = hipMemcpyAsync(dev_ptr + off, s.host, len, hipMemcpyHostToDevice, p->stream);
        if (e == hipSuccess) e = hipEventRecord(s.ev, p->stream);
        s.used = true;
      } else {
        // two chunks in flight per thread: issue this one, then drain the other slot while it travels
        if (s.used) e = hipEventSynchronize(s.ev);
        if (e != hipSuccess) break;
        e = hipMemcpyAsync(s.host, dev_ptr + off, len, hipMemcpyDeviceToHost, p->stream);
        if (e == hipSuccess) e = hipEventRecord(s.ev, p->stream);
        s.used = true;
        if (c >= (size_t)T) {  // previous chunk of this thread sits in the other slot
          agpu_device::StageSlot& o = dev->stage[(size_t)t * 2 + (size_t)(k ^ 1)];
          const size_t poff = (c - (size_t)T) * CH;
          if (e == hipSuccess) e = hipEventSynchronize(o.ev);
          if (e == hipSuccess) memcpy(host_ptr + poff, o.host, CH);
        }
      }
    }
    if (!to_device && e == hipSuccess) {  // drain the last chunk this thread issued
      size_t last = nchunks;
      for (size_t c = (size_t)t; c < nchunks; c += (size_t)T) last = c;
      if (last < nchunks) {
        const int lk = (int)(((last - (size_t)t) / (size_t)T) & 1);
        agpu_device::StageSlot& o = dev->stage[(size_t)t * 2 + (size_t)lk];
        const size_t off = last * CH;
        const size_t len = bytes - off < CH ? bytes - off : CH;
        e = hipEventSynchronize(o.ev);
        if (e == hipSuccess) memcpy(host_ptr + off, o.host, len);
      }
    }
    errs[(size_t)t] = e;
  };
  std::vector<std::thread> threads;
  for (int t = 1; t < T; t++) threads.emplace_back(worker, t);
  worker(0);
  for (auto& th : threads) th.join();
  for (hipError_t e : errs)
    if (e != hipSuccess) {
      (void)hipGetLastError();
      agpu_set_error("staged copy failed: %s", hipGetErrorString(e));
      return AGPU_ERR_HIP;
    }
  return AGPU_OK;
}

// Small and medium transfers (≤ AGPU_BOUNCE_MAX_BYTES = 4 MiB) never hand the caller's pageable pointer to the runtime: the bytes go
// through a page-locked slot of the device (one per direction).  A pageable
// hipMemcpy makes the runtime pin the caller's pages for the duration (a KFD userptr mapping); with the C heap (numpy,
// std::vector — brk memory that glibc keeps extending and trimming) that ended, about once in thirty runs of the GPU test
// suite, in "Memory access fault by GPU … on address <a page of the brk heap>" raised from the runtime's event thread
// (native backtrace via tools/probe/abort_bt.c): nothing of ours dereferences host memory, and the fault is fatal to
// the process.  Big transfers keep the direct path — their buffers are separate mappings that live until the call
// returns — and the measured 56 GB/s.
agpu_status agpu_internal_bounce_copy(agpu_pipeline* p, void* dev_ptr, void* host_ptr, size_t bytes, bool to_device) {
  agpu_device* dev = p->dev;
  const int dir = to_device ? 0 : 1;
  std::lock_guard<std::mutex> lock(dev->bounce_mu[dir]);
  agpu_device::StageSlot& s = dev->bounce[dir];
  if (!s.host) {
    AGPU_HIP(hipHostMalloc(&s.host, AGPU_BOUNCE_MAX_BYTES, hipHostMallocDefault));
    hipError_t e = hipEventCreateWithFlags(&s.ev, hipEventDisableTiming);
    if (e != hipSuccess) {
      (void)hipHostFree(s.host);
      s.host = nullptr;
      agpu_set_error("hipEventCreate failed: %s", hipGetErrorString(e));
      return AGPU_ERR_HIP;
    }
  }
  if (to_device) {
    memcpy(s.host, host_ptr, bytes);
    AGPU_HIP(hipMemcpyAsync(dev_ptr, s.host, bytes, hipMemcpyHostToDevice, p->stream));
    AGPU_HIP(hipStreamSynchronize(p->stream));  // the slot is free again and, as before, the data is in place on return
  } else {
    AGPU_HIP(hipMemcpyAsync(s.host, dev_ptr, bytes, hipMemcpyDeviceToHost, p->stream));
    AGPU_HIP(hipStreamSynchronize(p->stream));
    memcpy(host_ptr, s.host, bytes);
  }
  return AGPU_OK;
}

// Does [ptr, ptr + bytes) touch the C heap proper — the brk segment, "[heap]" in /proc/self/maps?  glibc's dynamic mmap
// threshold lets malloc serve blocks of up to 32 MiB from there once bigger ones have been freed, and that is the one kind
// of host memory whose pages come and go under a running process (the heap top is trimmed and re-extended): a pageable
// hipMemcpy — or hipHostRegister — pins such pages through a KFD userptr mapping, and that is what the GPU memory faults
// of round 2 pointed at.  mmap'ed blocks (numpy arrays ≥ 128 KiB by default, Arrow buffers, files) are
// separate mappings that live until the caller frees them.  The heap's start never moves; its end is sbrk(0).
static bool host_range_in_brk_heap(const void* ptr, size_t bytes) {
  static std::atomic<uintptr_t> heap_start{0};
  static std::atomic<uintptr_t> no_heap_at_brk{0};  // "no [heap] line" was established while sbrk(0) had this value: re-parse only when it moved
  uintptr_t start = heap_start.load(std::memory_order_relaxed);
  const uintptr_t brk_now = reinterpret_cast<uintptr_t>(sbrk(0));
  if (!start) {
    if (no_heap_at_brk.load(std::memory_order_relaxed) == brk_now) return false;
    FILE* f = fopen("/proc/self/maps", "r");
    if (f) {
      char line[512];
      while (fgets(line, sizeof line, f))
        if (strstr(line, "[heap]")) {
          unsigned long long a = 0;
          if (sscanf(line, "%llx-", &a) == 1) start = (uintptr_t)a;
          break;
        }
      fclose(f);
    }
    if (!start) {  // no heap segment yet: nothing can live in it (remembered until the break moves)
      no_heap_at_brk.store(brk_now, std::memory_order_relaxed);
      return false;
    }
    heap_start.store(start, std::memory_order_relaxed);
  }
  const uintptr_t lo = reinterpret_cast<uintptr_t>(ptr), hi = lo + bytes;
  return lo < brk_now && hi > start;
}

// glibc serves malloc calls of NON-main threads out of mmap'ed thread arenas — 64 MiB-aligned "heaps" of at most 64 MiB
// (HEAP_MAX_SIZE) that are trimmed and shrunk exactly like the brk heap, so their pages are the same hazard.  A block
// inside one is necessarily smaller than 64 MiB and never crosses a 64 MiB boundary; whatever is bigger is either the brk heap
// (checked above) or a mapping of its own.
// Round 4 sent every 4–64 MiB range to the staging engine unless it was PROVEN to be a mapping of its own (starts at its mapping's
// beginning, mapping not 64 MiB-aligned).  Measured in round 5 (tools/probe/host_copy_routes.py → profiles/r05_host_copy_routes*.json):
// that proof fails for nearly everything real — fresh numpy arrays (their mmapped chunk merges with a neighbouring mapping), numpy
// views at an offset, every pyarrow pool (mimalloc / jemalloc carve buffers out of big regions), arrays built by pa.array — and the
// staged path moves them at 14–40 GB/s where the runtime does 40–56.  So the test now asks the question itself: IS THE RANGE INSIDE A
// GLIBC THREAD ARENA?  Such a heap starts at A = lo & ~(64 MiB − 1) with a `heap_info` header {mstate ar_ptr; heap_info* prev; size_t
// size; size_t mprotect_size; …} (malloc/arena.c, unchanged in these fields from glibc 2.26 to 2.39): the first heap of an arena keeps
// its malloc_state right behind the header (ar_ptr − A is 32 or 48), later heaps point back (prev is 64 MiB-aligned, ar_ptr lies a
// header behind ANOTHER 64 MiB boundary); size ≤ mprotect_size ≤ 64 MiB, both page multiples.  A range is staged when the header at A
// is readable (it is read through the kernel — a pipe write that fails with EFAULT instead of a load that faults) and looks like that; any
// other memory — numpy, Arrow pools, file mappings, mmap itself — goes to the runtime at the link's rate.  A false "arena" costs
// bandwidth only; a false "not an arena" needs a glibc whose heap_info no longer starts with these four words, heaps that are not 64 MiB
// (glibc.malloc.hugetlb = 2) or another allocator altogether — host_allocator_unknown() below sends those processes down the safe path.
#define AGPU_THREAD_ARENA_MAX ((size_t)64 << 20)
static bool host_range_in_thread_arena(const void* ptr, size_t bytes) {
  const uintptr_t lo = reinterpret_cast<uintptr_t>(ptr), hi = lo + bytes, mask = AGPU_THREAD_ARENA_MAX - 1;
  if (((lo ^ (hi - 1)) & ~mask) != 0) return false;  // crosses a 64 MiB boundary: no single heap holds it
  const uintptr_t A = lo & ~mask;
  // The would-be header is read THROUGH THE KERNEL: write(2) from address A into a pipe returns EFAULT for an unmapped or unreadable page
  // where a plain load would raise SIGSEGV (a PROT_NONE reservation of another allocator may well sit at the 64 MiB boundary below a live
  // buffer, and another thread may unmap it at any time).  Two system calls, ≈ 2 µs — the /proc/self/maps pass this replaces cost 50–200 µs
  // per copy in a process with a few hundred mappings, as much as a 5 MiB transfer itself.
  uint64_t w[4];
  {
    static std::mutex pipe_mu;
    static int fds[2] = {-1, -1};
    static pid_t owner = 0;
    std::lock_guard<std::mutex> lk(pipe_mu);
    const pid_t me = getpid();
    if (fds[0] >= 0 && owner != me) {  // a forked child must not share the parent's pipe (its reads would take the parent's bytes)
      (void)close(fds[0]);
      (void)close(fds[1]);
      fds[0] = fds[1] = -1;
    }
    if (fds[0] < 0) {
      if (pipe2(fds, O_CLOEXEC | O_NONBLOCK) != 0) return true;  // cannot tell: the safe path
      owner = me;
    }
    const ssize_t put = write(fds[1], reinterpret_cast<const void*>(A), sizeof w);
    if (put != (ssize_t)sizeof w) {
      if (put > 0) {  // (cannot happen for 32 bytes; keep the pipe clean anyway)
        char sink[sizeof w];
        (void)!read(fds[0], sink, (size_t)put);
      }
      return false;  // nothing readable at the would-be header: not a heap
    }
    if (read(fds[0], w, sizeof w) != (ssize_t)sizeof w) return true;
  }
  const uint64_t ar_ptr = w[0], prev = w[1], size = w[2], mprot = w[3];
  const uint64_t ar_off = ar_ptr & mask;
  const bool header = ar_ptr != 0 && (ar_ptr & 7) == 0 && ar_off >= 16 && ar_off <= 256 &&
                      (prev == 0 ? (ar_ptr & ~(uint64_t)mask) == A : ((prev & mask) == 0 && prev != A)) &&
                      size >= 4096 && (size & 4095) == 0 && mprot >= size && (mprot & 4095) == 0 && mprot <= AGPU_THREAD_ARENA_MAX;
  return header;
}
// The header test above knows glibc's DEFAULT heaps.  Two set-ups it cannot vouch for: glibc.malloc.hugetlb = 2 makes the heaps
// 4 × the huge page size instead of 64 MiB, and a preloaded allocator (jemalloc, tcmalloc, mimalloc as LD_PRELOAD) trims and unmaps extents of
// its own that carry no heap_info at all.  In such a process every range below 64 MiB takes the staged path — bandwidth, not safety, is what
// that costs (14–40 GB/s instead of 33–56).
static bool host_allocator_unknown() {
  static const bool unknown = [] {
    const char* t = getenv("GLIBC_TUNABLES");
    if (t) {
      const char* h = strstr(t, "glibc.malloc.hugetlb=");
      if (h && h[21] != '0') return true;
    }
    const char* pre = getenv("LD_PRELOAD");
    if (pre && (strstr(pre, "jemalloc") || strstr(pre, "tcmalloc") || strstr(pre, "mimalloc") || strstr(pre, "malloc"))) return true;
    return false;
  }();
  return unknown;
}
static bool host_range_needs_staging(const void* ptr, size_t bytes) {
  if (host_range_in_brk_heap(ptr, bytes)) return true;
  if (bytes >= AGPU_THREAD_ARENA_MAX) return false;
  if (host_allocator_unknown()) return true;
  return host_range_in_thread_arena(ptr, bytes);
}

// One host↔HBM copy of a caller's (possibly pageable) range that is COMPLETE on return, never handing heap pages to the
// runtime: ≤ 4 MiB through the bounce slot, brk-heap ranges of any size through the page-locked chunk engine (mode 2's),
// everything else — separate mappings — straight to the runtime at the link's rate.  agpu_upload / agpu_download and the
// default staged copy all end here.
agpu_status agpu_internal_host_copy(agpu_pipeline* p, void* dev_ptr, void* host_ptr, size_t bytes, bool to_device) {
  if (!bytes) return AGPU_OK;
  // AGPU_HOST_COPY_DIRECT=1: hand every range to the runtime as it is — ONLY for tools/probe/heap_copy_stress.py --direct, the
  // reproducer of the round-2 memory fault
  static const bool direct = [] { const char* e = getenv("AGPU_HOST_COPY_DIRECT"); return e && *e && *e != '0'; }();
  if (!p->capturing && !direct) {
    if (bytes <= AGPU_BOUNCE_MAX_BYTES) return agpu_internal_bounce_copy(p, dev_ptr, host_ptr, bytes, to_device);
    if (host_range_needs_staging(host_ptr, bytes)) {
      agpu_status st = staged_copy_threads(p, static_cast<char*>(dev_ptr), static_cast<char*>(host_ptr), bytes, to_device);
      if (st != AGPU_OK) return st;
      AGPU_HIP(hipStreamSynchronize(p->stream));
      return AGPU_OK;
    }
  }
  if (to_device) AGPU_HIP(hipMemcpyAsync(dev_ptr, host_ptr, bytes, hipMemcpyHostToDevice, p->stream));
  else AGPU_HIP(hipMemcpyAsync(host_ptr, dev_ptr, bytes, hipMemcpyDeviceToHost, p->stream));
  AGPU_HIP(hipStreamSynchronize(p->stream));
  return AGPU_OK;
}
// test hook (tests/test_gpu_host_copies.py): which path a host range would take — 0 bounce, 1 chunk engine (brk heap / a malloc arena / unproven), 2 direct
extern "C" int32_t agpu_internal_host_copy_path(const void* host_ptr, size_t bytes) {
  if (bytes <= AGPU_BOUNCE_MAX_BYTES) return 0;
  return host_range_needs_staging(host_ptr, bytes) ? 1 : 2;
}

static agpu_status staged_copy_impl(agpu_pipeline* p, void* dev_ptr, void* host_ptr, size_t bytes, bool to_device) {
  if (!bytes) return AGPU_OK;
  // auto = mode 1: measured on the MI355X box (tools/probe/h2d_sweep.py → profiles/r02_h2d_sweep.json, 1 GiB, EPYC 9575F host):
  // pageable hipMemcpy 56.3 / 56.1 GB/s (H2D / D2H) — the runtime already moves pageable memory at the rate of the link
  // (page-locked reference: 57 GB/s) — hipHostRegister in place 57.6 / 57.0, the threaded staging below 45–46 at 2–32
  // threads (29 with one).  The staging engine stays selectable (tuning "h2d_mode" = 2) for hosts whose runtime stages
  // pageable copies slowly; it is not the default anywhere.
  int64_t mode = p->tune.h2d_mode;
  if (mode <= 0 || mode > 3) mode = 1;
  if (mode == 1) return agpu_internal_host_copy(p, dev_ptr, host_ptr, bytes, to_device);
  if (mode == 3 && host_range_needs_staging(host_ptr, bytes)) mode = 2;  // registering heap pages in place is the same userptr pin
  if (mode == 2) return staged_copy_threads(p, static_cast<char*>(dev_ptr), static_cast<char*>(host_ptr), bytes, to_device);
  if (mode == 3) {
    hipError_t e = hipHostRegister(host_ptr, bytes, hipHostRegisterDefault);
    if (e == hipSuccess) {
      e = to_device ? hipMemcpyAsync(dev_ptr, host_ptr, bytes, hipMemcpyHostToDevice, p->stream)
                    : hipMemcpyAsync(host_ptr, dev_ptr, bytes, hipMemcpyDeviceToHost, p->stream);
      if (e == hipSuccess) e = hipStreamSynchronize(p->stream);
      (void)hipHostUnregister(host_ptr);
      if (e != hipSuccess) {
        agpu_set_error("registered copy failed: %s", hipGetErrorString(e));
        return AGPU_ERR_HIP;
      }
      return AGPU_OK;
    }
    (void)hipGetLastError();  // e.g. read-only mapping: fall back to the pageable copy
  }
  if (to_device) AGPU_HIP(hipMemcpyAsync(dev_ptr, host_ptr, bytes, hipMemcpyHostToDevice, p->stream));
  else AGPU_HIP(hipMemcpyAsync(host_ptr, dev_ptr, bytes, hipMemcpyDeviceToHost, p->stream));
  AGPU_HIP(hipStreamSynchronize(p->stream));
  return AGPU_OK;
}

// ---------------------------------------------------------------- Arrow C Data Interface
static bool dtype_of_format(const char* f, agpu_dtype* out) {
  if (!f) return false;
  if (!strcmp(f, "c")) *out = AGPU_I8;
  else if (!strcmp(f, "C")) *out = AGPU_U8;
  else if (!strcmp(f, "s")) *out = AGPU_I16;
  else if (!strcmp(f, "S")) *out = AGPU_U16;
  else if (!strcmp(f, "i")) *out = AGPU_I32;
  else if (!strcmp(f, "I")) *out = AGPU_U32;
  else if (!strcmp(f, "f")) *out = AGPU_F32;
  else if (!strcmp(f, "b")) *out = AGPU_BOOL;
  else if (!strcmp(f, "tdD")) *out = AGPU_DATE32;
  else return false;
  return true;
}
static const char* format_of_dtype(agpu_dtype t) {
  switch (t) {
    case AGPU_I8: return "c";
    case AGPU_U8: return "C";
    case AGPU_I16: return "s";
    case AGPU_U16: return "S";
    case AGPU_I32: return "i";
    case AGPU_U32: return "I";
    case AGPU_F32: return "f";
    case AGPU_BOOL: return "b";
    case AGPU_DATE32: return "tdD";
  }
  return nullptr;
}

// Arrow bitmap (byte-granular, arbitrary bit offset) → word-aligned device bitmap `out` of `out_b` bytes, padding bits 0
static agpu_status import_bitmap_into(agpu_pipeline* p, const uint8_t* host_bits, uint64_t bit_offset, uint64_t n_bits, void* out,
                                      size_t out_b) {
  agpu_device* dev = p->dev;
  if (!n_bits) return AGPU_OK;
  agpu_status st = AGPU_OK;
  const uint64_t first = bit_offset / 8, last = (bit_offset + n_bits + 7) / 8, span = last - first;
  if ((bit_offset & 7) == 0) {  // byte-aligned slice: the bytes are the bitmap; only the padding needs clearing
    AGPU_HIP(hipMemsetAsync(static_cast<char*>(out) + (out_b - 8), 0, 8, p->stream));
    st = staged_copy_impl(p, out, const_cast<uint8_t*>(host_bits + first), span, true);
    if (st == AGPU_OK && (n_bits & 7)) st = agpu_bitmap_copy_bits(p, out, 0, out, n_bits);  // mask the tail bits
  } else {
    void* tmp = nullptr;
    st = agpu_malloc(dev, agpu_bitmap_bytes(span * 8) + 8, 0, &tmp);
    if (st == AGPU_OK) {
      AGPU_HIP(hipMemsetAsync(static_cast<char*>(tmp) + agpu_bitmap_bytes(span * 8) - 8, 0, 16, p->stream));
      st = staged_copy_impl(p, tmp, const_cast<uint8_t*>(host_bits + first), span, true);
      if (st == AGPU_OK) st = agpu_bitmap_copy_bits(p, tmp, bit_offset & 7, out, n_bits);
      (void)agpu_free(dev, tmp);  // recycled only after the stream has passed the kernel above (runtime.hip markers)
    }
  }
  return st;
}
static size_t import_bitmap_bytes(uint64_t n_bits) { return agpu_bitmap_bytes(n_bits) ? agpu_bitmap_bytes(n_bits) : 8; }
static agpu_status import_bitmap(agpu_pipeline* p, const uint8_t* host_bits, uint64_t bit_offset, uint64_t n_bits,
                                 void** out_dev, uint64_t* out_bytes) {
  const size_t out_b = import_bitmap_bytes(n_bits);
  void* out = nullptr;
  agpu_status st = agpu_malloc(p->dev, out_b, 0, &out);
  if (st != AGPU_OK) return st;
  st = import_bitmap_into(p, host_bits, bit_offset, n_bits, out, out_b);
  if (st != AGPU_OK) {
    (void)agpu_free(p->dev, out);
    return st;
  }
  *out_dev = out;
  *out_bytes = out_b;
  return AGPU_OK;
}

// Host memory for an exported column.  A D2H copy into never-touched pages takes one page fault per 4 KiB inside the
// copy (measured: 7 GB/s for a 1 GiB column against 56 GB/s into touched memory), so large buffers are 2 MiB-aligned,
// offered to transparent huge pages and first-touched by a few threads in parallel before the DMA starts (12 GB/s end to
// end) — and when the consumer releases an exported array its big buffers go to a small process-wide cache instead of
// back to the OS, so the NEXT export of a similar size lands in pages that are already there (the link's rate).  The
// cache holds at most AGPU_EXPORT_CACHE_BYTES (4 GiB; environment AGPU_EXPORT_CACHE_MB overrides, 0 disables).
struct ExportCache {
  std::mutex mu;
  std::vector<std::pair<void*, size_t>> blocks;  // {pointer, padded size}
  size_t bytes = 0;
  size_t cap = (size_t)4 << 30;
  ExportCache() {
    if (const char* e = getenv("AGPU_EXPORT_CACHE_MB")) cap = (size_t)strtoull(e, nullptr, 10) << 20;
  }
};
static ExportCache& export_cache() {  // never destroyed: consumers may release exported arrays while the process exits
  static ExportCache* c = new ExportCache;
  return *c;
}
static const size_t kExportBig = (size_t)4 << 20;
static size_t export_padded(size_t bytes) {
  const size_t align = bytes >= kExportBig ? ((size_t)2 << 20) : 64;
  const size_t padded = (bytes + align - 1) / align * align;
  return padded ? padded : align;
}
static void* alloc_export_buffer(size_t bytes) {
  const size_t padded = export_padded(bytes);
  if (bytes >= kExportBig) {
    ExportCache& c = export_cache();
    std::lock_guard<std::mutex> lock(c.mu);
    for (size_t i = 0; i < c.blocks.size(); i++)
      if (c.blocks[i].second >= padded && c.blocks[i].second <= padded + padded / 4) {  // touched already: no faults in the copy
        void* ptr = c.blocks[i].first;
        c.bytes -= c.blocks[i].second;
        c.blocks[i] = c.blocks.back();
        c.blocks.pop_back();
        return ptr;
      }
  }
  const size_t align = bytes >= kExportBig ? ((size_t)2 << 20) : 64;
  void* ptr = nullptr;
  if (posix_memalign(&ptr, align, padded) != 0) return nullptr;
  if (bytes >= kExportBig) {
    (void)madvise(ptr, padded, MADV_HUGEPAGE);
    const int T = 8;
    const size_t per = (padded / T + 4095) & ~(size_t)4095;
    std::vector<std::thread> th;
    auto touch = [=](int t) {
      char* b = static_cast<char*>(ptr);
      const size_t lo = (size_t)t * per, hi = lo + per < padded ? lo + per : padded;
      for (size_t off = lo; off < hi; off += 4096) b[off] = 0;
    };
    for (int t = 0; t < T; t++) {
      try {
        th.emplace_back(touch, t);
      } catch (...) {  // no thread to be had: touch this slice here
        touch(t);
      }
    }
    for (auto& x : th) x.join();
  }
  return ptr;
}
// `bytes` = what the buffer was allocated for (its padded size is recomputed: cached blocks may be up to 25 % bigger,
// which only means the cache's accounting is conservative)
static void free_export_buffer(void* ptr, size_t bytes) {
  if (!ptr) return;
  if (bytes >= kExportBig) {
    ExportCache& c = export_cache();
    const size_t padded = export_padded(bytes);
    std::lock_guard<std::mutex> lock(c.mu);
    if (c.bytes + padded <= c.cap) {
      c.blocks.emplace_back(ptr, padded);
      c.bytes += padded;
      return;
    }
  }
  free(ptr);
}

struct ExportPrivate {
  void* values;
  void* validity;
  const void* buffers[2];
  size_t values_bytes, validity_bytes;
};

static void release_exported_array(struct ArrowArray* a) {
  if (!a || !a->release) return;
  ExportPrivate* pd = static_cast<ExportPrivate*>(a->private_data);
  if (pd) {
    free_export_buffer(pd->values, pd->values_bytes);
    free_export_buffer(pd->validity, pd->validity_bytes);
    delete pd;
  }
  a->release = nullptr;
}
static void release_exported_schema(struct ArrowSchema* s) {
  if (!s || !s->release) return;
  s->release = nullptr;  // format / name point at static storage
}

extern "C" {

agpu_status agpu_staged_copy(agpu_pipeline* p, void* dev_ptr, void* host_ptr, size_t bytes, int32_t to_device) {
  AGPU_BIND(p);
  if (!bytes) return AGPU_OK;
  AGPU_REQUIRE(dev_ptr && host_ptr, AGPU_ERR_ARG, "null pointer");
  AGPU_REQUIRE(!p->capturing, AGPU_ERR_ARG, "not during graph capture");
  return staged_copy_impl(p, dev_ptr, host_ptr, bytes, to_device != 0);
}

agpu_status agpu_import_arrow(agpu_pipeline* p, const struct ArrowArray* array, const struct ArrowSchema* schema,
                              agpu_arrow_column* out_column) {
  AGPU_BIND(p);
  AGPU_REQUIRE(array && schema && out_column, AGPU_ERR_ARG, "null argument");
  AGPU_REQUIRE(!p->capturing, AGPU_ERR_ARG, "not during graph capture");
  AGPU_REQUIRE(array->release && schema->release, AGPU_ERR_ARG, "released ArrowArray / ArrowSchema");
  memset(out_column, 0, sizeof(*out_column));
  agpu_dtype dt;
  if (!dtype_of_format(schema->format, &dt)) {
    agpu_set_error("Arrow format '%s' has no GPU array type (c C s S i I f b tdD)", schema->format ? schema->format : "(null)");
    return AGPU_ERR_UNSUPPORTED;
  }
  AGPU_REQUIRE(!array->dictionary && array->n_children == 0, AGPU_ERR_UNSUPPORTED, "nested / dictionary arrays are not supported");
  AGPU_REQUIRE(array->n_buffers == 2 && array->buffers, AGPU_ERR_SHAPE, "a primitive array has exactly 2 buffers");
  AGPU_REQUIRE(array->length >= 0 && array->offset >= 0, AGPU_ERR_SHAPE, "negative length / offset");
  const uint64_t n = (uint64_t)array->length, off = (uint64_t)array->offset;
  const uint8_t* vbits = static_cast<const uint8_t*>(array->buffers[0]);
  const uint8_t* data = static_cast<const uint8_t*>(array->buffers[1]);
  AGPU_REQUIRE(n == 0 || data, AGPU_ERR_SHAPE, "null data buffer");
  agpu_device* dev = p->dev;
  agpu_arrow_column col;
  memset(&col, 0, sizeof(col));
  col.dtype = dt;
  col.length = n;
  col.null_count = array->null_count;
  agpu_status st = AGPU_OK;
  if (vbits && array->null_count != 0 && n) {
    st = import_bitmap(p, vbits, off, n, &col.validity, &col.validity_bytes);
    if (st != AGPU_OK) return st;
  } else {
    col.null_count = 0;
  }
  if (dt == AGPU_BOOL) {
    st = import_bitmap(p, data, off, n, &col.values, &col.values_bytes);
  } else {
    const size_t w = agpu_dtype_size(dt);
    col.values_bytes = n * w ? n * w : 16;
    st = agpu_malloc(dev, col.values_bytes, 0, &col.values);
    if (st == AGPU_OK) st = staged_copy_impl(p, col.values, const_cast<uint8_t*>(data + off * w), n * w, true);
  }
  if (st != AGPU_OK) {
    (void)agpu_arrow_column_free(dev, &col);
    return st;
  }
  *out_column = col;
  return AGPU_OK;
}

// The columns of one record batch → ONE device block placed for the HBM channel hash (agpu_malloc_table): value buffers
// first (in column order), validity bitmaps behind them.
agpu_status agpu_import_arrow_table(agpu_pipeline* p, int32_t n_columns, const struct ArrowArray* const* arrays,
                                    const struct ArrowSchema* const* schemas, agpu_arrow_column* out_columns) {
  AGPU_BIND(p);
  AGPU_REQUIRE(n_columns > 0 && arrays && schemas && out_columns, AGPU_ERR_ARG, "bad argument");
  AGPU_REQUIRE(!p->capturing, AGPU_ERR_ARG, "not during graph capture");
  std::vector<agpu_arrow_column> cols((size_t)n_columns);
  std::vector<uint64_t> sizes;
  std::vector<int> has_v((size_t)n_columns, 0);
  for (int32_t k = 0; k < n_columns; k++) {
    const struct ArrowArray* a = arrays[k];
    const struct ArrowSchema* sc = schemas[k];
    AGPU_REQUIRE(a && sc && a->release && sc->release, AGPU_ERR_ARG, "null / released ArrowArray / ArrowSchema");
    agpu_dtype dt;
    if (!dtype_of_format(sc->format, &dt)) {
      agpu_set_error("Arrow format '%s' has no GPU array type (c C s S i I f b tdD)", sc->format ? sc->format : "(null)");
      return AGPU_ERR_UNSUPPORTED;
    }
    AGPU_REQUIRE(!a->dictionary && a->n_children == 0, AGPU_ERR_UNSUPPORTED, "nested / dictionary arrays are not supported");
    AGPU_REQUIRE(a->n_buffers == 2 && a->buffers && a->length >= 0 && a->offset >= 0, AGPU_ERR_SHAPE, "a primitive array has exactly 2 buffers");
    AGPU_REQUIRE(a->length == 0 || a->buffers[1], AGPU_ERR_SHAPE, "null data buffer");
    agpu_arrow_column& c = cols[(size_t)k];
    memset(&c, 0, sizeof(c));
    c.dtype = dt;
    c.length = (uint64_t)a->length;
    has_v[(size_t)k] = (a->buffers[0] && a->null_count != 0 && a->length) ? 1 : 0;
    c.null_count = has_v[(size_t)k] ? a->null_count : 0;
    c.values_bytes = dt == AGPU_BOOL ? import_bitmap_bytes(c.length) : (c.length * agpu_dtype_size(dt) ? c.length * agpu_dtype_size(dt) : 16);
    c.validity_bytes = has_v[(size_t)k] ? import_bitmap_bytes(c.length) : 0;
    sizes.push_back(c.values_bytes);
  }
  for (int32_t k = 0; k < n_columns; k++)
    if (has_v[(size_t)k]) sizes.push_back(cols[(size_t)k].validity_bytes);
  std::vector<void*> ptrs(sizes.size(), nullptr);
  agpu_status st = agpu_malloc_table(p->dev, (int32_t)sizes.size(), sizes.data(), 0, ptrs.data());
  if (st != AGPU_OK) return st;
  size_t nv = (size_t)n_columns;
  for (int32_t k = 0; k < n_columns; k++) {
    cols[(size_t)k].values = ptrs[(size_t)k];
    if (has_v[(size_t)k]) cols[(size_t)k].validity = ptrs[nv++];
  }
  for (int32_t k = 0; k < n_columns && st == AGPU_OK; k++) {
    const struct ArrowArray* a = arrays[k];
    agpu_arrow_column& c = cols[(size_t)k];
    const uint64_t n = c.length, off = (uint64_t)a->offset;
    if (has_v[(size_t)k]) st = import_bitmap_into(p, static_cast<const uint8_t*>(a->buffers[0]), off, n, c.validity, c.validity_bytes);
    if (st != AGPU_OK) break;
    if (c.dtype == AGPU_BOOL) st = import_bitmap_into(p, static_cast<const uint8_t*>(a->buffers[1]), off, n, c.values, c.values_bytes);
    else if (n) {
      const size_t w = agpu_dtype_size(c.dtype);
      st = staged_copy_impl(p, c.values, const_cast<uint8_t*>(static_cast<const uint8_t*>(a->buffers[1]) + off * w), n * w, true);
    }
  }
  if (st != AGPU_OK) {
    for (void* q : ptrs) (void)agpu_free(p->dev, q);
    return st;
  }
  for (int32_t k = 0; k < n_columns; k++) out_columns[k] = cols[(size_t)k];
  return AGPU_OK;
}

agpu_status agpu_import_arrow_stream_next(agpu_pipeline* p, struct ArrowArrayStream* stream, const int32_t* columns,
                                          int32_t n_columns, agpu_arrow_column* out_columns, int64_t* out_rows) {
  AGPU_REQUIRE(p && stream && stream->release && columns && out_columns && out_rows && n_columns > 0, AGPU_ERR_ARG, "bad argument");
  *out_rows = -1;
  struct ArrowSchema schema;
  struct ArrowArray batch;
  memset(&schema, 0, sizeof(schema));
  memset(&batch, 0, sizeof(batch));
  if (stream->get_schema(stream, &schema) != 0 || !schema.release) {
    const char* why = stream->get_last_error ? stream->get_last_error(stream) : nullptr;
    agpu_set_error("ArrowArrayStream.get_schema failed: %s", why ? why : "(no message)");
    return AGPU_ERR_ARG;
  }
  agpu_status st = AGPU_OK;
  if (stream->get_next(stream, &batch) != 0) {
    const char* why = stream->get_last_error ? stream->get_last_error(stream) : nullptr;
    agpu_set_error("ArrowArrayStream.get_next failed: %s", why ? why : "(no message)");
    st = AGPU_ERR_ARG;
  } else if (batch.release) {  // a released (all-zero) array marks the end of the stream
    // a record batch travels as a struct array: children = the columns, schema.children = their schemas
    if (!schema.format || strcmp(schema.format, "+s") != 0 || batch.n_children != schema.n_children || batch.offset != 0) {
      agpu_set_error("agpu_import_arrow_stream_next: the stream's items are not plain struct arrays (format '%s')", schema.format ? schema.format : "(null)");
      st = AGPU_ERR_UNSUPPORTED;
    } else {
      std::vector<const struct ArrowArray*> ap((size_t)n_columns);
      std::vector<const struct ArrowSchema*> sp((size_t)n_columns);
      for (int32_t k = 0; k < n_columns && st == AGPU_OK; k++) {
        if (columns[k] < 0 || columns[k] >= batch.n_children) {
          agpu_set_error("agpu_import_arrow_stream_next: column index %d out of range", (int)columns[k]);
          st = AGPU_ERR_ARG;
        } else {
          ap[(size_t)k] = batch.children[columns[k]];
          sp[(size_t)k] = schema.children[columns[k]];
        }
      }
      if (st == AGPU_OK) st = agpu_import_arrow_table(p, n_columns, ap.data(), sp.data(), out_columns);
      if (st == AGPU_OK) *out_rows = batch.length;
    }
    batch.release(&batch);
  }
  schema.release(&schema);
  return st;
}

agpu_status agpu_export_arrow(agpu_pipeline* p, const agpu_arrow_column* column, struct ArrowArray* out_array,
                              struct ArrowSchema* out_schema) {
  AGPU_BIND(p);
  AGPU_REQUIRE(column && out_array && out_schema, AGPU_ERR_ARG, "null argument");
  AGPU_REQUIRE(!p->capturing, AGPU_ERR_ARG, "not during graph capture");
  const char* fmt = format_of_dtype(column->dtype);
  AGPU_REQUIRE(fmt, AGPU_ERR_ARG, "bad dtype");
  const uint64_t n = column->length;
  AGPU_REQUIRE(n == 0 || column->values, AGPU_ERR_ARG, "null values");
  const size_t vbytes = column->dtype == AGPU_BOOL ? agpu_bitmap_bytes(n) : (size_t)n * agpu_dtype_size(column->dtype);
  const size_t nbytes = column->validity ? agpu_bitmap_bytes(n) : 0;
  ExportPrivate* pd = new ExportPrivate{nullptr, nullptr, {nullptr, nullptr}, vbytes, nbytes};
  // 64-byte aligned and padded, as the Arrow specification recommends
  pd->values = alloc_export_buffer(vbytes);
  if (nbytes) pd->validity = alloc_export_buffer(nbytes);
  agpu_status st = (!pd->values || (nbytes && !pd->validity)) ? AGPU_ERR_ARG : AGPU_OK;
  if (st != AGPU_OK) agpu_set_error("host allocation of %zu bytes failed", vbytes + nbytes);
  if (st == AGPU_OK) st = staged_copy_impl(p, column->values, pd->values, vbytes, false);
  if (st == AGPU_OK && nbytes) st = staged_copy_impl(p, column->validity, pd->validity, nbytes, false);
  if (st == AGPU_OK) {  // modes 1 and 3 have synchronised already; mode 2 drained its slots — make it unconditional
    hipError_t e = hipStreamSynchronize(p->stream);
    if (e != hipSuccess) {
      agpu_set_error("hipStreamSynchronize failed: %s", hipGetErrorString(e));
      st = AGPU_ERR_HIP;
    }
  }
  if (st != AGPU_OK) {
    free_export_buffer(pd->values, vbytes);
    free_export_buffer(pd->validity, nbytes);
    delete pd;
    return st;
  }
  pd->buffers[0] = pd->validity;
  pd->buffers[1] = pd->values;
  memset(out_array, 0, sizeof(*out_array));
  out_array->length = (int64_t)n;
  out_array->null_count = column->validity ? column->null_count : 0;
  out_array->offset = 0;
  out_array->n_buffers = 2;
  out_array->buffers = pd->buffers;
  out_array->release = release_exported_array;
  out_array->private_data = pd;
  memset(out_schema, 0, sizeof(*out_schema));
  out_schema->format = fmt;
  out_schema->name = "";
  out_schema->flags = ARROW_FLAG_NULLABLE;
  out_schema->release = release_exported_schema;
  return AGPU_OK;
}

agpu_status agpu_arrow_column_free(agpu_device* dev, agpu_arrow_column* column) {
  AGPU_REQUIRE(dev && column, AGPU_ERR_ARG, "null argument");
  agpu_status a = agpu_free(dev, column->values), b = agpu_free(dev, column->validity);
  column->values = column->validity = nullptr;
  return a != AGPU_OK ? a : b;
}

}  // extern "C"
