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
#include <rccl/rccl.h>

#include <chrono>
#include <condition_variable>
#include <memory>
#include <thread>

#include "common.hpp"

static_assert(AGPU_COMM_ID_BYTES == sizeof(ncclUniqueId), "agpu_comm id blob must hold an ncclUniqueId");

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
};

// AGPU_COMM_TIMEOUT_MS: how long a collective rendezvous (communicator init, barrier) may wait for the other ranks
// before the call gives up with AGPU_ERR_HIP.  Default 120 s; 0 = wait for ever (RCCL's own behaviour).
static int64_t comm_timeout_ms() {
  const char* e = getenv("AGPU_COMM_TIMEOUT_MS");
  if (e && *e) return strtoll(e, nullptr, 10);
  return 120000;
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
};

agpu_status agpu_comm_init_rank_timeout(agpu_device* dev, const void* unique_id, int32_t rank, int32_t world,
                                        int64_t timeout_ms, agpu_comm** out_comm) {
  AGPU_REQUIRE(dev && unique_id && out_comm, AGPU_ERR_ARG, "null argument");
  AGPU_REQUIRE(world >= 1 && world <= 256 && rank >= 0 && rank < world, AGPU_ERR_ARG, "bad rank / world (1..256 ranks)");
  *out_comm = nullptr;
  AGPU_HIP(hipSetDevice(dev->ordinal));
  ncclUniqueId id;
  memcpy(&id, unique_id, sizeof(id));
  ncclComm_t comm = nullptr;
  if (timeout_ms <= 0) {
    AGPU_NCCL(ncclCommInitRank(&comm, world, id, rank));
  } else {
    auto job = std::make_shared<comm_init_job>();
    const int ordinal = dev->ordinal;
    std::thread([job, id, rank, world, ordinal]() {
      ncclComm_t c = nullptr;
      hipError_t he = hipSetDevice(ordinal);
      ncclResult_t r = he == hipSuccess ? ncclCommInitRank(&c, world, id, rank) : ncclUnhandledCudaError;
      std::lock_guard<std::mutex> lk(job->mu);
      job->comm = c;
      job->result = r;
      job->hip = he;
      job->done = true;
      job->cv.notify_all();
    }).detach();
    std::unique_lock<std::mutex> lk(job->mu);
    if (!job->cv.wait_for(lk, std::chrono::milliseconds(timeout_ms), [&] { return job->done; })) {
      agpu_set_error("agpu_comm_init_rank: rank %d of %d gave up after %lld ms waiting for the other ranks "
                     "(ncclCommInitRank is still pending on a helper thread: exit this process)",
                     (int)rank, (int)world, (long long)timeout_ms);
      return AGPU_ERR_HIP;
    }
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
  char* mem = nullptr;
  hipError_t e = hipMalloc(&mem, 16 + 16 * (size_t)world + 16);
  if (e != hipSuccess) {
    (void)ncclCommDestroy(comm);
    agpu_set_error("hipMalloc of the communicator records failed: %s", hipGetErrorString(e));
    return AGPU_ERR_HIP;
  }
  (void)hipMemset(mem, 0, 16 + 16 * (size_t)world + 16);
  (void)hipStreamSynchronize(nullptr);
  agpu_comm* c = new agpu_comm();
  c->dev = dev;
  c->comm = comm;
  c->rank = rank;
  c->world = world;
  c->send = mem;
  c->recv = mem + 16;
  c->token = reinterpret_cast<int32_t*>(mem + 16 + 16 * (size_t)world);
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
  (void)hipDeviceSynchronize();
  (void)ncclCommDestroy(c->comm);
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

static agpu_status comm_check(agpu_comm* c, agpu_pipeline* p) {
  AGPU_REQUIRE(c, AGPU_ERR_ARG, "null communicator");
  AGPU_REQUIRE(c->dev == p->dev, AGPU_ERR_ARG, "communicator and pipeline belong to different devices");
  AGPU_REQUIRE(!p->capturing, AGPU_ERR_ARG, "collectives are not captured into graphs");
  return AGPU_OK;
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
  if (count == 0) return AGPU_OK;
  AGPU_NCCL(ncclAllReduce(buf_dev, buf_dev, (size_t)count, dt, ro, c->comm, p->stream));
  return AGPU_OK;
}

// gather the record every rank left in c->send, then combine in rank order (reduce.hip)
static agpu_status gather_and_finish(agpu_comm* c, agpu_pipeline* p, int kind, agpu_dtype dtype, void* out_dev) {
  AGPU_NCCL(ncclAllGather(c->send, c->recv, 16, ncclUint8, c->comm, p->stream));
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

agpu_status agpu_comm_barrier(agpu_comm* c, agpu_pipeline* p) {
  AGPU_BIND(p);
  agpu_status st = comm_check(c, p);
  if (st != AGPU_OK) return st;
  comm_use use(c, p);
  AGPU_NCCL(ncclAllReduce(c->token, c->token, 1, ncclInt32, ncclMax, c->comm, p->stream));
  const int64_t limit = comm_timeout_ms();
  if (limit <= 0) {
    AGPU_HIP(hipStreamSynchronize(p->stream));
    return AGPU_OK;
  }
  // a peer that died never joins the all-reduce and the stream would never drain: poll with a deadline instead
  const auto t0 = std::chrono::steady_clock::now();
  for (uint32_t spin = 0;; spin++) {
    hipError_t q = hipStreamQuery(p->stream);
    if (q == hipSuccess) return AGPU_OK;
    if (q != hipErrorNotReady) {
      agpu_set_error("agpu_comm_barrier: %s", hipGetErrorString(q));
      return AGPU_ERR_HIP;
    }
    if (spin > 2000) std::this_thread::sleep_for(std::chrono::microseconds(50));
    if ((spin & 255) == 255 &&
        std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count() > limit) {
      agpu_set_error("agpu_comm_barrier: rank %d of %d waited %lld ms for the other ranks (AGPU_COMM_TIMEOUT_MS): exit this process",
                     c->rank, c->world, (long long)limit);
      return AGPU_ERR_HIP;
    }
  }
}

}  // extern "C"
