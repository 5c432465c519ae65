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
#include <rccl/rccl.h>

#include "common.hpp"

static_assert(AGPU_COMM_ID_BYTES == sizeof(ncclUniqueId), "agpu_comm id blob must hold an ncclUniqueId");

struct agpu_comm {
  agpu_device* dev;
  ncclComm_t comm;
  int rank, world;
  char* send;      // device: one 16-byte record
  char* recv;      // device: world records
  int32_t* token;  // device: barrier word
};

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

agpu_status agpu_comm_init_rank(agpu_device* dev, const void* unique_id, int32_t rank, int32_t world, agpu_comm** out_comm) {
  AGPU_REQUIRE(dev && unique_id && out_comm, AGPU_ERR_ARG, "null argument");
  AGPU_REQUIRE(world >= 1 && world <= 256 && rank >= 0 && rank < world, AGPU_ERR_ARG, "bad rank / world (1..256 ranks)");
  *out_comm = nullptr;
  AGPU_HIP(hipSetDevice(dev->ordinal));
  ncclUniqueId id;
  memcpy(&id, unique_id, sizeof(id));
  ncclComm_t comm = nullptr;
  AGPU_NCCL(ncclCommInitRank(&comm, world, id, rank));  // blocks until all `world` ranks have called it
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
  *out_comm = c;
  return AGPU_OK;
}

agpu_status agpu_comm_destroy(agpu_comm* c) {
  if (!c) return AGPU_OK;
  (void)hipSetDevice(c->dev->ordinal);
  (void)hipDeviceSynchronize();
  (void)ncclCommDestroy(c->comm);
  (void)hipFree(c->send);
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

agpu_status agpu_comm_all_reduce(agpu_comm* c, agpu_pipeline* p, agpu_reduce_op op, agpu_comm_dtype ctype, void* buf_dev,
                                 uint64_t count) {
  AGPU_BIND(p);
  agpu_status st = comm_check(c, p);
  if (st != AGPU_OK) return st;
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
  AGPU_NCCL(ncclAllReduce(c->token, c->token, 1, ncclInt32, ncclMax, c->comm, p->stream));
  AGPU_HIP(hipStreamSynchronize(p->stream));
  return AGPU_OK;
}

}  // extern "C"
