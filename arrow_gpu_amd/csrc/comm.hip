// comm.hip — multi-GPU half of the C ABI: an RCCL communicator and the FINAL REDUCE of whole-column statistics.
//
// Nothing here exists in the reference (one wgpu device + one queue: crates/array/src/gpu_utils/gpu_device.rs:29-33).
// north_star config 5: a column is chunk-sharded across the GPUs of one node (one host thread or process per GPU,
// contiguous row ranges), every element-wise / compare / cast / bitmap kernel runs shard-local with NO collective,
// and only sum / min / max (and null counts) finish with a collective of ONE record per rank over RCCL / xGMI.
//
// The final reduce is an all-gather of a 16-byte record {statistic, n_local} per rank followed by a single-wave
// kernel on every rank that combines the records IN RANK ORDER with exactly the functors of the shard-local kernels
// (reduce.hip).  Why not ncclAllReduce: (i) its association order over ranks is an implementation detail of the ring
// / tree RCCL picks, so an f32 sum would not be reproducible across world sizes or RCCL versions, while the gathered
// records can be summed in the reference's own adjacent-pair tree order — a sharded f32 Sum over shards of 256^k
// rows is then BIT-IDENTICAL to the reference's tree over the whole column [ref: aggregate.wgsl:28-37,
// aggregate_kernels.rs:26-43]; (ii) Arrow's min/max ignore NaN unless every value is NaN, which ncclMin/ncclMax do
// not promise; (iii) empty shards must contribute the identity.  The message is 16 B per rank: pure latency, the
// xGMI link rate is irrelevant, one collective per statistic.  agpu_comm_all_reduce is the plain ncclAllReduce for
// integer counts (null counts, row counts) where the order cannot matter.
#include <dlfcn.h>
#include <unistd.h>
#include <rccl/rccl.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <memory>
#include <thread>

#include "common.hpp"

static_assert(AGPU_COMM_ID_BYTES == sizeof(ncclUniqueId), "agpu_comm id blob must hold an ncclUniqueId");

struct comm_init_job;
struct agpu_comm {
  agpu_device* dev;
  ncclComm_t comm;
  int rank, world;
  char* send;      // device: one 16-byte record
  char* recv;      // device: world records
  int32_t* token;  // device: barrier word
  // One record in flight: every collective here writes send / recv asynchronously on the CALLER's stream.  Calls from one
  // pipeline are stream-ordered; a call from ANOTHER pipeline is ordered behind the previous call with this event, so two
  // pipelines sharing a communicator never race on the record (and RCCL sees its collectives in one order).  `mu` covers
  // the host side (two host threads).
  std::mutex mu;
  hipEvent_t last_done = nullptr;
  hipStream_t last_stream = nullptr;
  bool used = false;
  char* peers = nullptr;  // device: world + 1 identity records (agpu_comm_peers), part of the one allocation behind `send`
  // Collectives enqueued and not yet known to be over (a successful deadline wait on the stream of the last one clears it;
  // every call is ordered behind the previous one, so that stream drains last).  A wait that gives up while this is zero
  // was only behind ordinary work: it reports the timeout and leaves the device alone.
  std::atomic<uint32_t> in_flight{0};
  // world 1 only: RCCL's bootstrap did not come up within the deadline — no ncclComm_t; the collectives of a single rank are device copies
  bool local = false;
  // … and the helper that is still inside ncclCommInitRank: if it comes back after all, agpu_comm_destroy (or the helper itself, once the
  // communicator is gone) destroys the ncclComm_t nobody uses — a live RCCL communicator at process exit is a crash (comm_init_job below)
  std::shared_ptr<comm_init_job> late_init;
  std::mutex peers_mu;  // one agpu_comm_peers at a time (they share `peers`); never taken together with a wait on `mu`
};

// AGPU_COMM_TIMEOUT_MS: how long a collective rendezvous (communicator init, barrier) may wait for the other ranks
// before the call gives up with AGPU_ERR_HIP.  Default 120 s; 0 = wait for ever (RCCL's own behaviour).
static int64_t comm_timeout_ms() {
  const char* e = getenv("AGPU_COMM_TIMEOUT_MS");
  if (e && *e) return strtoll(e, nullptr, 10);
  return 120000;
}

// A collective whose peer never joined stays queued on its stream for ever: every later wait on that stream, every
// hipDeviceSynchronize, hipFree and hipStreamDestroy of the device would block behind it.  The deadline-aware waits below
// therefore POISON the device when they give up (agpu_device::poisoned): from then on every ABI call that could wait
// returns AGPU_ERR_HIP at once, the destroy calls leak instead of synchronising (runtime.hip), and agpu_comm_destroy
// aborts the communicator instead of draining it — so a host that unwinds through its destructors after the timeout
// (C++ exceptions, Python __del__) ends instead of hanging.  The process is expected to exit.
static void comm_poison(agpu_comm* c, const char* what, long long waited_ms) {
  c->dev->poisoned.store(true, std::memory_order_release);
  agpu_set_error("%s: rank %d of %d waited %lld ms for the other ranks (AGPU_COMM_TIMEOUT_MS); the device is poisoned — "
                 "every further call fails fast, destroy calls leak: exit this process",
                 what, c->rank, c->world, waited_ms);
}

// Host wait for everything queued on p's stream, giving up after AGPU_COMM_TIMEOUT_MS (0 = wait for ever).  Called WITHOUT
// c->mu: other host threads may enqueue on the communicator while this one polls (the mutex only covers the bookkeeping).
static agpu_status comm_wait_stream(agpu_comm* c, agpu_pipeline* p, const char* what) {
  const int64_t limit = comm_timeout_ms();
  uint32_t seen;
  bool last_here;
  {
    std::lock_guard<std::mutex> lk(c->mu);
    seen = c->in_flight.load(std::memory_order_relaxed);
    last_here = c->used && c->last_stream == p->stream;
  }
  // everything this communicator enqueued up to `seen` is over once the stream of its last call has drained
  auto drained = [&]() {
    if (!last_here || !seen) return;
    std::lock_guard<std::mutex> lk(c->mu);
    const uint32_t now = c->in_flight.load(std::memory_order_relaxed);
    c->in_flight.store(now >= seen ? now - seen : 0, std::memory_order_relaxed);
  };
  if (limit <= 0) {
    AGPU_HIP(hipStreamSynchronize(p->stream));
    drained();
    return AGPU_OK;
  }
  const auto t0 = std::chrono::steady_clock::now();
  for (uint32_t spin = 0;; spin++) {
    hipError_t q = hipStreamQuery(p->stream);
    if (q == hipSuccess) {
      drained();
      return AGPU_OK;
    }
    if (q != hipErrorNotReady) {
      agpu_set_error("%s: %s", what, hipGetErrorString(q));
      return AGPU_ERR_HIP;
    }
    if (spin > 2000) std::this_thread::sleep_for(std::chrono::microseconds(50));
    if ((spin & 255) == 255) {
      const long long waited = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count();
      if (waited > limit) {
        if (c->in_flight.load(std::memory_order_acquire) == 0) {  // no collective of this communicator anywhere: slow ordinary work
          agpu_set_error("%s: the stream did not drain within %lld ms (AGPU_COMM_TIMEOUT_MS); no collective of this communicator is in "
                         "flight, so the device is NOT poisoned — wait again or raise the limit", what, waited);
          return AGPU_ERR_HIP;
        }
        comm_poison(c, what, waited);
        return AGPU_ERR_HIP;
      }
    }
  }
}

// reduce.hip: combine `world` gathered records in rank order → out_dev (1 element); kind: 0..2 = agpu_reduce_op on
// `dtype`, 3 = f64 sum
agpu_status agpu_internal_comm_finish(agpu_pipeline* p, int kind, agpu_dtype dtype, const void* records, int world,
                                      void* out_dev);
agpu_status agpu_internal_comm_pack(agpu_pipeline* p, void* record, uint64_t n_local);

#define AGPU_NCCL(call)                                                                          \
  do {                                                                                           \
    ncclResult_t _r = (call);                                                                    \
    if (_r != ncclSuccess) {                                                                     \
      agpu_set_error("%s failed: %s (%s:%d)", #call, ncclGetErrorString(_r), __FILE__, __LINE__); \
      return AGPU_ERR_HIP;                                                                       \
    }                                                                                            \
  } while (0)

extern "C" {

agpu_status agpu_comm_get_unique_id(void* out_id) {
  AGPU_REQUIRE(out_id, AGPU_ERR_ARG, "null out_id");
  ncclUniqueId id;
  AGPU_NCCL(ncclGetUniqueId(&id));
  memcpy(out_id, &id, sizeof(id));
  return AGPU_OK;
}

// ncclCommInitRank blocks until all `world` ranks have called it — for ever when one never arrives (a worker that died
// before reaching it, a launcher that started fewer ranks).  The blocking call therefore runs on a helper thread and the
// caller waits for it with a deadline.  On a timeout the helper is still inside RCCL and cannot be cancelled: it is
// detached, the call reports AGPU_ERR_HIP, and the PROCESS should exit (a fresh process is the only clean retry; the
// communicator of a rank that did arrive is useless without its peers anyway).  The collectives themselves stay on the
// plain blocking-communicator path.
struct comm_init_job {
  std::mutex mu;
  std::condition_variable cv;
  bool done = false;
  ncclResult_t result = ncclSuccess;
  hipError_t hip = hipSuccess;
  ncclComm_t comm = nullptr;
  bool orphaned = false;  // one rank, bootstrap late, and the local communicator that stood in for it is gone: the helper cleans up after itself
};
// Seen with the test hook (AGPU_COMM_TEST_STALL_INIT_MS=19000, deadline 20 s: the helper's ncclCommInitRank returns a second AFTER the call gave
// up): the process ran to its end on the local communicator and died with SIGSEGV at exit — RCCL's communicator, with its proxy threads, was
// still alive.  Whoever sees the late ncclComm_t last destroys it.
// A helper that is STILL inside ncclCommInitRank when its communicator is destroyed is orphaned: it cleans up after itself when RCCL lets it
// go.  Its job stays on this list so that agpu_device_destroy can wait for it (bounded): a helper that creates and destroys an RCCL
// communicator while the process tears the runtime down is the exit crash described above.
static std::mutex g_orphans_mu;
static std::vector<std::shared_ptr<comm_init_job>> g_orphans;
static void comm_reap_late_init(const std::shared_ptr<comm_init_job>& job) {
  if (!job) return;
  ncclComm_t late = nullptr;
  bool orphan = false;
  {
    std::unique_lock<std::mutex> lk(job->mu);
    if (job->cv.wait_for(lk, std::chrono::milliseconds(2000), [&] { return job->done; })) {
      late = job->comm;
      job->comm = nullptr;
    } else {
      job->orphaned = true;
      orphan = true;
    }
  }
  if (late) (void)ncclCommDestroy(late);
  if (orphan) {
    std::lock_guard<std::mutex> lk(g_orphans_mu);
    g_orphans.push_back(job);
  }
}
// runtime.hip agpu_device_destroy: wait (at most wait_ms in all) for the orphaned init helpers to leave RCCL; true when none is left inside
bool agpu_internal_comm_wait_orphans(int64_t wait_ms) {
  std::vector<std::shared_ptr<comm_init_job>> jobs;
  {
    std::lock_guard<std::mutex> lk(g_orphans_mu);
    jobs.swap(g_orphans);
  }
  const auto deadline = std::chrono::steady_clock::now() + std::chrono::milliseconds(wait_ms);
  bool all = true;
  for (auto& job : jobs) {
    std::unique_lock<std::mutex> lk(job->mu);
    if (!job->cv.wait_until(lk, deadline, [&] { return job->done; })) {
      all = false;
      lk.unlock();
      std::lock_guard<std::mutex> g(g_orphans_mu);
      g_orphans.push_back(job);  // still pending: a later destroy may wait again
    }
  }
  if (!all)
    fprintf(stderr, "arrow_gpu_hip: an RCCL bootstrap that never came up is still pending on a helper thread at device teardown "
                    "(waited %lld ms); the process may not exit cleanly\n", (long long)wait_ms);
  return all;
}

agpu_status agpu_comm_init_rank_timeout(agpu_device* dev, const void* unique_id, int32_t rank, int32_t world,
                                        int64_t timeout_ms, agpu_comm** out_comm) {
  AGPU_REQUIRE(dev && unique_id && out_comm, AGPU_ERR_ARG, "null argument");
  AGPU_REQUIRE(world >= 1 && world <= 256 && rank >= 0 && rank < world, AGPU_ERR_ARG, "bad rank / world (1..256 ranks)");
  *out_comm = nullptr;
  AGPU_HIP(hipSetDevice(dev->ordinal));
  ncclUniqueId id;
  memcpy(&id, unique_id, sizeof(id));
  ncclComm_t comm = nullptr;
  bool is_local = false;
  std::shared_ptr<comm_init_job> late;
  // a one-rank communicator waits for nobody: its bootstrap takes a fraction of a second or is not going to come up — 20 s at most
  if (world == 1 && timeout_ms > 20000) timeout_ms = 20000;
  if (timeout_ms <= 0) {
    AGPU_NCCL(ncclCommInitRank(&comm, world, id, rank));
  } else {
    auto job = std::make_shared<comm_init_job>();
    const int ordinal = dev->ordinal;
    // test hook, compiled into the TEST build of the library only (csrc/Makefile `hooks`: libarrow_gpu_hip_hooks.so, -DAGPU_TEST_HOOKS;
    // tests/test_gpu_comm.py loads it through AGPU_LIB): AGPU_COMM_TEST_STALL_INIT_MS makes the helper sit that long before it calls RCCL — a
    // bootstrap that does not come up, reproducibly.  The product library has no such switch.
#ifdef AGPU_TEST_HOOKS
    static const long stall_ms = [] { const char* e = getenv("AGPU_COMM_TEST_STALL_INIT_MS"); return e && *e ? strtol(e, nullptr, 10) : 0L; }();
#else
    static const long stall_ms = 0;
#endif
    std::thread([job, id, rank, world, ordinal]() {
      if (stall_ms > 0) std::this_thread::sleep_for(std::chrono::milliseconds(stall_ms));
      ncclComm_t c = nullptr;
      hipError_t he = hipSetDevice(ordinal);
      ncclResult_t r = he == hipSuccess ? ncclCommInitRank(&c, world, id, rank) : ncclUnhandledCudaError;
      bool orphaned = false;
      {
        std::lock_guard<std::mutex> lk(job->mu);
        orphaned = job->orphaned;
        job->comm = orphaned ? nullptr : c;
        job->result = r;
        job->hip = he;
        job->done = true;
        job->cv.notify_all();
      }
      if (orphaned && c) (void)ncclCommDestroy(c);
    }).detach();
    std::unique_lock<std::mutex> lk(job->mu);
    bool local = false;
    if (!job->cv.wait_for(lk, std::chrono::milliseconds(timeout_ms), [&] { return job->done; })) {
      if (world == 1) {
        // A single rank waits for nobody: what did not come up is RCCL's own socket bootstrap (seen once in round 5: ncclCommInitRank of a
        // one-rank communicator pending for 60 s on a shared node, while the same call took 0.3 s minutes later).  A one-rank job must not
        // die of that — the communicator becomes LOCAL (agpu_comm_is_local): no ncclComm_t, its collectives are the device copies they
        // amount to at world 1.  The helper stays parked in RCCL's bootstrap (sockets, no device work yet) and is abandoned.
        agpu_set_error("agpu_comm_init_rank: RCCL's bootstrap of a ONE-rank communicator did not come up within %lld ms — local communicator",
                       (long long)timeout_ms);
        local = true;
      } else {
        agpu_set_error("agpu_comm_init_rank: rank %d of %d gave up after %lld ms waiting for the other ranks "
                       "(ncclCommInitRank is still pending on a helper thread: exit this process)",
                       (int)rank, (int)world, (long long)timeout_ms);
        dev->poisoned.store(true, std::memory_order_release);  // the helper may hold device work: nothing may wait for the device now
        return AGPU_ERR_HIP;
      }
    }
    if (!local) {
      if (job->hip != hipSuccess) {
        agpu_set_error("hipSetDevice(%d) failed on the init thread: %s", ordinal, hipGetErrorString(job->hip));
        return AGPU_ERR_HIP;
      }
      if (job->result != ncclSuccess) {
        agpu_set_error("ncclCommInitRank failed: %s", ncclGetErrorString(job->result));
        return AGPU_ERR_HIP;
      }
      comm = job->comm;
    }
    is_local = local;
    if (local) late = job;
  }
  char* mem = nullptr;
  const size_t rec_bytes = 16 + 16 * (size_t)world + 16, peer_bytes = sizeof(agpu_comm_peer) * ((size_t)world + 1);
  hipError_t e = hipMalloc(&mem, rec_bytes + peer_bytes);
  if (e != hipSuccess) {
    if (comm) (void)ncclCommDestroy(comm);
    comm_reap_late_init(late);
    agpu_set_error("hipMalloc of the communicator records failed: %s", hipGetErrorString(e));
    return AGPU_ERR_HIP;
  }
  (void)hipMemset(mem, 0, rec_bytes + peer_bytes);
  (void)hipStreamSynchronize(nullptr);
  agpu_comm* c = new agpu_comm();
  c->dev = dev;
  c->comm = comm;
  c->local = is_local;
  c->late_init = late;
  c->rank = rank;
  c->world = world;
  c->send = mem;
  c->recv = mem + 16;
  c->token = reinterpret_cast<int32_t*>(mem + 16 + 16 * (size_t)world);
  c->peers = mem + rec_bytes;  // 16-byte aligned: rec_bytes is a multiple of 16
  if (hipEventCreateWithFlags(&c->last_done, hipEventDisableTiming) != hipSuccess) c->last_done = nullptr;
  *out_comm = c;
  return AGPU_OK;
}

agpu_status agpu_comm_init_rank(agpu_device* dev, const void* unique_id, int32_t rank, int32_t world, agpu_comm** out_comm) {
  return agpu_comm_init_rank_timeout(dev, unique_id, rank, world, comm_timeout_ms(), out_comm);
}

// Which shared objects the collectives and the HIP runtime of THIS process come from (dladdr of one symbol of each): a
// process that imported torch first runs on torch's bundled librccl / libamdhip64, one that did not on /opt/rocm's — the
// bench prints this so a multi-GPU record says which runtime it measured.
agpu_status agpu_comm_runtime_info(char* out, size_t out_cap) {
  AGPU_REQUIRE(out && out_cap > 0, AGPU_ERR_ARG, "null out");
  Dl_info a{}, b{};
  const char* rccl = dladdr(reinterpret_cast<void*>(&ncclGetUniqueId), &a) && a.dli_fname ? a.dli_fname : "?";
  const char* hip = dladdr(reinterpret_cast<void*>(&hipGetDeviceCount), &b) && b.dli_fname ? b.dli_fname : "?";
  int v = 0, hv = 0;
  (void)ncclGetVersion(&v);
  (void)hipRuntimeGetVersion(&hv);
  snprintf(out, out_cap, "rccl %d (built against %d) from %s; hip runtime %d from %s", v, (int)NCCL_VERSION_CODE, rccl, hv, hip);
  return AGPU_OK;
}

agpu_status agpu_comm_destroy(agpu_comm* c) {
  if (!c) return AGPU_OK;
  (void)hipSetDevice(c->dev->ordinal);
  if (c->dev->poisoned.load(std::memory_order_acquire)) {
    // a collective is stuck on some stream: never wait for the device.  ncclCommAbort raises the abort flag the stuck
    // kernel polls (so the stream may drain after all); it runs on a helper thread because it may itself wait, and is
    // given two seconds.  The record buffers and the event leak — the process is about to exit.
    if (c->local) {
      comm_reap_late_init(c->late_init);
      delete c;
      return AGPU_OK;
    }
    auto done = std::make_shared<std::atomic<bool>>(false);
    ncclComm_t comm = c->comm;
    const int ordinal = c->dev->ordinal;
    std::thread([done, comm, ordinal]() {
      (void)hipSetDevice(ordinal);
      (void)ncclCommAbort(comm);
      done->store(true, std::memory_order_release);
    }).detach();
    for (int i = 0; i < 400 && !done->load(std::memory_order_acquire); i++) std::this_thread::sleep_for(std::chrono::milliseconds(5));
    delete c;
    return AGPU_OK;
  }
  (void)hipDeviceSynchronize();
  if (!c->local) (void)ncclCommDestroy(c->comm);
  else comm_reap_late_init(c->late_init);
  (void)hipFree(c->send);
  if (c->last_done) (void)hipEventDestroy(c->last_done);
  delete c;
  return AGPU_OK;
}

agpu_status agpu_comm_rank(agpu_comm* c, int32_t* out_rank, int32_t* out_world) {
  AGPU_REQUIRE(c, AGPU_ERR_ARG, "null communicator");
  if (out_rank) *out_rank = c->rank;
  if (out_world) *out_world = c->world;
  return AGPU_OK;
}

agpu_status agpu_comm_is_local(agpu_comm* c, int32_t* out_local) {
  AGPU_REQUIRE(c && out_local, AGPU_ERR_ARG, "null argument");
  *out_local = c->local ? 1 : 0;
  return AGPU_OK;
}

// what RCCL itself reports — not what the caller passed to init
agpu_status agpu_comm_size(agpu_comm* c, int32_t* out_count, int32_t* out_user_rank, int32_t* out_device) {
  AGPU_REQUIRE(c, AGPU_ERR_ARG, "null communicator");
  if (c->local) {  // one rank, this device (agpu_comm_is_local says that RCCL is not behind these numbers)
    if (out_count) *out_count = 1;
    if (out_user_rank) *out_user_rank = 0;
    if (out_device) *out_device = c->dev->ordinal;
    return AGPU_OK;
  }
  int v = 0;
  if (out_count) {
    AGPU_NCCL(ncclCommCount(c->comm, &v));
    *out_count = v;
  }
  if (out_user_rank) {
    AGPU_NCCL(ncclCommUserRank(c->comm, &v));
    *out_user_rank = v;
  }
  if (out_device) {
    AGPU_NCCL(ncclCommCuDevice(c->comm, &v));
    *out_device = v;
  }
  return AGPU_OK;
}

static uint64_t fnv1a64(const void* data, size_t n, uint64_t h = 1469598103934665603ull) {
  const unsigned char* b = static_cast<const unsigned char*>(data);
  for (size_t i = 0; i < n; i++) h = (h ^ b[i]) * 1099511628211ull;
  return h;
}

// this process's identity record (no communicator needed: also what a host that gathers by other means would ship)
agpu_status agpu_device_identity(agpu_device* dev, agpu_comm_peer* out) {
  AGPU_REQUIRE(dev && out, AGPU_ERR_ARG, "null argument");
  memset(out, 0, sizeof(*out));
  out->rank = -1;
  out->world = -1;
  out->nccl_device = -1;
  out->device_ordinal = dev->ordinal;
  out->pci_domain = dev->props.pciDomainID;
  out->pci_bus = dev->props.pciBusID;
  out->pci_device = dev->props.pciDeviceID;
  out->pid = (int32_t)getpid();
  char host[256] = {0};
  (void)gethostname(host, sizeof(host) - 1);
  uint64_t h = fnv1a64(host, strlen(host));
  if (FILE* f = fopen("/proc/sys/kernel/random/boot_id", "r")) {  // two containers of one name on different machines differ here
    char boot[64] = {0};
    if (fgets(boot, sizeof(boot), f)) h = fnv1a64(boot, strlen(boot), h);
    fclose(f);
  }
  out->host_hash = h;
  static_assert(sizeof(out->uuid) == sizeof(dev->props.uuid.bytes), "uuid is 16 bytes");
  memcpy(out->uuid, dev->props.uuid.bytes, sizeof(out->uuid));
  snprintf(out->gcn_arch, sizeof(out->gcn_arch), "%s", dev->props.gcnArchName);
  return AGPU_OK;
}

static agpu_status comm_check(agpu_comm* c, agpu_pipeline* p) {
  AGPU_REQUIRE(c, AGPU_ERR_ARG, "null communicator");
  AGPU_REQUIRE(c->dev == p->dev, AGPU_ERR_ARG, "communicator and pipeline belong to different devices");
  AGPU_REQUIRE(!p->capturing, AGPU_ERR_ARG, "collectives are not captured into graphs");
  return AGPU_OK;  // (a poisoned device never gets here: AGPU_BIND refuses it)
}

// Holds the communicator for one collective call: orders p's stream behind the previous call when that ran on another
// stream, and leaves the "done" event behind on the way out.
struct comm_use {
  agpu_comm* c;
  agpu_pipeline* p;
  std::unique_lock<std::mutex> lk;
  comm_use(agpu_comm* c_, agpu_pipeline* p_) : c(c_), p(p_), lk(c_->mu) {
    if (c->used && c->last_done && c->last_stream != p->stream) (void)hipStreamWaitEvent(p->stream, c->last_done, 0);
  }
  ~comm_use() {
    c->in_flight.fetch_add(1, std::memory_order_release);  // (a call that failed before it enqueued only makes a later timeout poison)
    if (c->last_done && hipEventRecord(c->last_done, p->stream) == hipSuccess) {
      c->last_stream = p->stream;
      c->used = true;
    }
  }
};

agpu_status agpu_comm_all_reduce(agpu_comm* c, agpu_pipeline* p, agpu_reduce_op op, agpu_comm_dtype ctype, void* buf_dev,
                                 uint64_t count) {
  AGPU_BIND(p);
  agpu_status st = comm_check(c, p);
  if (st != AGPU_OK) return st;
  comm_use use(c, p);
  AGPU_REQUIRE(buf_dev || count == 0, AGPU_ERR_ARG, "null buffer");
  ncclDataType_t dt;
  switch (ctype) {
    case AGPU_COMM_F32: dt = ncclFloat32; break;
    case AGPU_COMM_F64: dt = ncclFloat64; break;
    case AGPU_COMM_I32: dt = ncclInt32; break;
    case AGPU_COMM_U32: dt = ncclUint32; break;
    case AGPU_COMM_I64: dt = ncclInt64; break;
    case AGPU_COMM_U64: dt = ncclUint64; break;
    default: agpu_set_error("bad agpu_comm_dtype %d", (int)ctype); return AGPU_ERR_ARG;
  }
  ncclRedOp_t ro;
  switch (op) {
    case AGPU_RED_SUM: ro = ncclSum; break;
    case AGPU_RED_MIN: ro = ncclMin; break;
    case AGPU_RED_MAX: ro = ncclMax; break;
    default: agpu_set_error("bad reduce op %d", (int)op); return AGPU_ERR_ARG;
  }
  if (count == 0 || c->local) return AGPU_OK;  // (one rank: the buffer already holds the result)
  AGPU_NCCL(ncclAllReduce(buf_dev, buf_dev, (size_t)count, dt, ro, c->comm, p->stream));
  return AGPU_OK;
}

// gather the record every rank left in c->send, then combine in rank order (reduce.hip)
static agpu_status gather_and_finish(agpu_comm* c, agpu_pipeline* p, int kind, agpu_dtype dtype, void* out_dev) {
  if (c->local) AGPU_HIP(hipMemcpyAsync(c->recv, c->send, 16, hipMemcpyDeviceToDevice, p->stream));
  else AGPU_NCCL(ncclAllGather(c->send, c->recv, 16, ncclUint8, c->comm, p->stream));
  return agpu_internal_comm_finish(p, kind, dtype, c->recv, c->world, out_dev);
}

agpu_status agpu_comm_reduce(agpu_comm* c, agpu_pipeline* p, agpu_reduce_op op, agpu_dtype dtype, const void* in,
                             const void* validity, uint64_t n_local, void* out_dev) {
  AGPU_BIND(p);
  agpu_status st = comm_check(c, p);
  if (st != AGPU_OK) return st;
  comm_use use(c, p);
  AGPU_REQUIRE(out_dev, AGPU_ERR_ARG, "null out_dev");
  AGPU_REQUIRE((int)op >= 0 && (int)op <= 2, AGPU_ERR_ARG, "bad reduce op");
  st = agpu_internal_comm_pack(p, c->send, n_local);
  if (st != AGPU_OK) return st;
  st = agpu_reduce(p, op, dtype, in, validity, n_local, c->send);  // shard-local statistic → low bytes of the record
  if (st != AGPU_OK) return st;
  return gather_and_finish(c, p, (int)op, dtype == AGPU_DATE32 ? AGPU_I32 : dtype, out_dev);
}

agpu_status agpu_comm_reduce_sum_f64(agpu_comm* c, agpu_pipeline* p, const float* in, const void* validity,
                                     uint64_t n_local, double* out_dev) {
  AGPU_BIND(p);
  agpu_status st = comm_check(c, p);
  if (st != AGPU_OK) return st;
  comm_use use(c, p);
  AGPU_REQUIRE(out_dev, AGPU_ERR_ARG, "null out_dev");
  st = agpu_internal_comm_pack(p, c->send, n_local);
  if (st != AGPU_OK) return st;
  st = agpu_reduce_sum_f64(p, in, validity, n_local, reinterpret_cast<double*>(c->send));
  if (st != AGPU_OK) return st;
  return gather_and_finish(c, p, 3, AGPU_F32, out_dev);
}

// Final reduce of a statistic the caller already holds per shard (1 element of `dtype` at partial_dev — or an f64
// when kind_f64 != 0): the same gather + rank-ordered combine, without the shard-local pass.
agpu_status agpu_comm_final_reduce(agpu_comm* c, agpu_pipeline* p, agpu_reduce_op op, agpu_dtype dtype, int32_t kind_f64,
                                   const void* partial_dev, uint64_t n_local, void* out_dev) {
  AGPU_BIND(p);
  agpu_status st = comm_check(c, p);
  if (st != AGPU_OK) return st;
  comm_use use(c, p);
  AGPU_REQUIRE(partial_dev && out_dev, AGPU_ERR_ARG, "null pointer");
  AGPU_REQUIRE((int)op >= 0 && (int)op <= 2, AGPU_ERR_ARG, "bad reduce op");
  AGPU_REQUIRE(!kind_f64 || op == AGPU_RED_SUM, AGPU_ERR_UNSUPPORTED, "f64 partials are sums only");
  if (dtype == AGPU_DATE32) dtype = AGPU_I32;
  AGPU_REQUIRE(kind_f64 || dtype == AGPU_F32 || dtype == AGPU_I32 || dtype == AGPU_U32, AGPU_ERR_UNSUPPORTED,
               "32-bit statistics only (like agpu_reduce)");
  st = agpu_internal_comm_pack(p, c->send, n_local);
  if (st != AGPU_OK) return st;
  AGPU_HIP(hipMemcpyAsync(c->send, partial_dev, kind_f64 ? 8 : 4, hipMemcpyDeviceToDevice, p->stream));
  return gather_and_finish(c, p, kind_f64 ? 3 : (int)op, dtype, out_dev);
}

// One pass over the shard for all four statistics (reduce.hip launch_stats_f32 behind agpu_reduce_stats_f32), then the four final reduces
// — each exactly agpu_comm_final_reduce — in place on the record's fields.
agpu_status agpu_comm_reduce_stats_f32(agpu_comm* c, agpu_pipeline* p, const float* in, const void* validity, uint64_t n_local,
                                       agpu_f32_stats* out_dev) {
  agpu_status st = agpu_reduce_stats_f32(p, in, validity, n_local, out_dev);
  if (st != AGPU_OK) return st;
  st = agpu_comm_final_reduce(c, p, AGPU_RED_SUM, AGPU_F32, 0, &out_dev->sum, n_local, &out_dev->sum);
  if (st == AGPU_OK) st = agpu_comm_final_reduce(c, p, AGPU_RED_MIN, AGPU_F32, 0, &out_dev->min, n_local, &out_dev->min);
  if (st == AGPU_OK) st = agpu_comm_final_reduce(c, p, AGPU_RED_MAX, AGPU_F32, 0, &out_dev->max, n_local, &out_dev->max);
  if (st == AGPU_OK) st = agpu_comm_final_reduce(c, p, AGPU_RED_SUM, AGPU_F32, 1, &out_dev->sum_f64, n_local, &out_dev->sum_f64);
  return st;
}

agpu_status agpu_comm_barrier(agpu_comm* c, agpu_pipeline* p) {
  AGPU_BIND(p);
  agpu_status st = comm_check(c, p);
  if (st != AGPU_OK) return st;
  {
    comm_use use(c, p);
    if (!c->local) AGPU_NCCL(ncclAllReduce(c->token, c->token, 1, ncclInt32, ncclMax, c->comm, p->stream));
  }
  // a peer that died never joins the all-reduce and the stream would never drain: poll with a deadline instead (the
  // communicator is not held meanwhile)
  return comm_wait_stream(c, p, "agpu_comm_barrier");
}

// The host wait that belongs behind agpu_comm_reduce / _all_reduce / _final_reduce: like agpu_pipeline_sync, but it gives up
// after AGPU_COMM_TIMEOUT_MS (and poisons the device) instead of blocking for ever behind a collective a dead peer never joins.
agpu_status agpu_comm_sync(agpu_comm* c, agpu_pipeline* p) {
  AGPU_BIND(p);
  agpu_status st = comm_check(c, p);
  if (st != AGPU_OK) return st;
  return comm_wait_stream(c, p, "agpu_comm_sync");  // polls without c->mu: other threads keep using the communicator
}

// One identity record per rank, gathered THROUGH the communicator: proof of which devices joined it.  out_host[r] is rank
// r's record (rank / world as RCCL reports them there); *out_distinct = the number of distinct (host, PCI address) pairs.
agpu_status agpu_comm_peers(agpu_comm* c, agpu_pipeline* p, agpu_comm_peer* out_host, int32_t cap, int32_t* out_distinct) {
  AGPU_BIND(p);
  agpu_status st = comm_check(c, p);
  if (st != AGPU_OK) return st;
  AGPU_REQUIRE(out_host, AGPU_ERR_ARG, "null out_host");
  agpu_comm_peer mine;
  st = agpu_device_identity(c->dev, &mine);
  if (st != AGPU_OK) return st;
  st = agpu_comm_size(c, &mine.world, &mine.rank, &mine.nccl_device);
  if (st != AGPU_OK) return st;
  // the gather moves ncclCommCount records: that, not the world the caller passed to init, is what must fit
  AGPU_REQUIRE(mine.world == c->world, AGPU_ERR_SHAPE, "ncclCommCount differs from the world this communicator was created with");
  AGPU_REQUIRE(cap >= mine.world, AGPU_ERR_ARG, "out_host must hold one record per rank (ncclCommCount)");
  const size_t rec = sizeof(agpu_comm_peer);
  std::lock_guard<std::mutex> one_at_a_time(c->peers_mu);
  char* mem = c->peers;  // owned by the communicator: no allocation, no device-wide hipFree per call, nothing to leak on an error
  {
    comm_use use(c, p);
    hipError_t e = hipMemcpyAsync(mem, &mine, rec, hipMemcpyHostToDevice, p->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(p->stream);  // `mine` is pageable stack memory: the copy must be over before it goes
    if (e != hipSuccess) {
      agpu_set_error("agpu_comm_peers: %s", hipGetErrorString(e));
      return AGPU_ERR_HIP;
    }
    ncclResult_t r = ncclSuccess;
    if (c->local) {
      e = hipMemcpyAsync(mem + rec, mem, rec, hipMemcpyDeviceToDevice, p->stream);
      if (e != hipSuccess) {
        agpu_set_error("agpu_comm_peers: %s", hipGetErrorString(e));
        return AGPU_ERR_HIP;
      }
    } else {
      r = ncclAllGather(mem, mem + rec, rec, ncclUint8, c->comm, p->stream);
    }
    if (r != ncclSuccess) {
      agpu_set_error("ncclAllGather failed: %s", ncclGetErrorString(r));
      return AGPU_ERR_HIP;
    }
  }
  st = comm_wait_stream(c, p, "agpu_comm_peers");
  if (st != AGPU_OK) return st;
  hipError_t e = hipMemcpy(out_host, mem + rec, rec * (size_t)c->world, hipMemcpyDeviceToHost);
  if (e != hipSuccess) {
    agpu_set_error("agpu_comm_peers: %s", hipGetErrorString(e));
    return AGPU_ERR_HIP;
  }
  if (out_distinct) {
    int32_t d = 0;
    for (int i = 0; i < c->world; i++) {
      bool seen = false;
      for (int j = 0; j < i && !seen; j++)
        seen = out_host[j].host_hash == out_host[i].host_hash && out_host[j].pci_domain == out_host[i].pci_domain &&
               out_host[j].pci_bus == out_host[i].pci_bus && out_host[j].pci_device == out_host[i].pci_device;
      d += seen ? 0 : 1;
    }
    *out_distinct = d;
  }
  return AGPU_OK;
}

}  // extern "C"
