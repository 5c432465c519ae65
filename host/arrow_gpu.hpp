// arrow_gpu.hpp — C++17 host layer over the C ABI (include/arrow_gpu.h), mirroring psvri/arrow-gpu's Rust API 1:1.
//
// The reference's host is Rust (crates/array, crates/arithmetic, crates/compare, crates/logical, crates/cast,
// crates/math, crates/trigonometry, crates/routines); no Rust toolchain exists in the build image, so the compiled-
// language host is this header.  Names are the reference's:
//   GpuDevice, ArrowComputePipeline                      crates/array/src/gpu_utils/{gpu_device,compute_pipeline}.rs
//   PrimitiveArrayGpu<T>, Float32ArrayGPU … Date32ArrayGPU, BooleanArrayGPU, NullBitBufferGpu, BooleanBufferBuilder,
//   ArrowArrayGPU (enum → std::variant), ArrowType       crates/array/src/array/*.rs
//   x.add(y) / x.add_op(y, pipeline) / add_dyn(a, b) / add_op_dyn(a, b, pipeline) …   the op traits + dyn_fn! tables
// Rust trait bounds become static_asserts (an op a type does not implement fails to compile, as in Rust); the *_dyn
// functions throw ArrowErrorGPU where the reference panics.  Header-only; link with -larrow_gpu_hip.
#pragma once
#include <array>
#include <cstdint>
#include <cstring>
#include <exception>
#include <memory>
#include <mutex>
#include <optional>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <variant>
#include <vector>

#include "../include/arrow_gpu.h"

namespace arrow_gpu {

// ------------------------------------------------------------------ errors  [crates/array/src/lib.rs:11-14]
struct ArrowErrorGPU : std::runtime_error {
  enum Kind { OperationNotSupported, CastingNotSupported, Runtime } kind;
  ArrowErrorGPU(Kind k, const std::string& m) : std::runtime_error(m), kind(k) {}
};
inline void check(agpu_status s, const char* what) {
  if (s == AGPU_OK) return;
  const std::string msg = std::string(what) + ": " + agpu_last_error();
  throw ArrowErrorGPU(s == AGPU_ERR_UNSUPPORTED ? ArrowErrorGPU::OperationNotSupported : ArrowErrorGPU::Runtime, msg);
}

enum class ArrowType { BooleanType, Float32Type, UInt32Type, UInt16Type, UInt8Type, Int32Type, Int16Type, Int8Type, Date32Type };

// ------------------------------------------------------------------ device, buffers, pipeline
class GpuDevice;
using DevicePtr = std::shared_ptr<GpuDevice>;

struct Buffer {  // Arc<wgpu::Buffer> stand-in [crates/array/src/array/buffer.rs:5-53]
  void* ptr = nullptr;
  uint64_t bytes = 0;
  DevicePtr dev;
  uint64_t size() const { return bytes; }
  ~Buffer();
};
using BufferPtr = std::shared_ptr<Buffer>;

class ArrowComputePipeline;

class GpuDevice : public std::enable_shared_from_this<GpuDevice> {
 public:
  agpu_device* raw = nullptr;
  agpu_pipeline* io = nullptr;  // uploads / read-backs / immediate clones; a pipeline serves one thread at a time:
  std::mutex io_mu;             // … so the device-level helpers below take this lock
  static DevicePtr create(int ordinal = 0) {  // GpuDevice::new() [gpu_device.rs:46-85]
    auto d = std::shared_ptr<GpuDevice>(new GpuDevice());
    check(agpu_device_create(ordinal, &d->raw), "agpu_device_create");
    check(agpu_pipeline_create(d->raw, &d->io), "agpu_pipeline_create");
    return d;
  }
  // GpuDevice::from_adapter(adapter) [gpu_device.rs:87-106] named a wgpu adapter; on a ROCm node the choice among the visible GPUs is
  // the device ordinal (after HIP_VISIBLE_DEVICES), so the adapter IS the ordinal
  static DevicePtr from_adapter(int adapter) { return create(adapter); }
  ~GpuDevice() {
    if (io) agpu_pipeline_destroy(io);
    if (raw) agpu_device_destroy(raw);
  }
  // `like`: the buffers this one will be read / written together with (an op's inputs, for its output) — big blocks are
  // then placed against them for the HBM channel hash (agpu_malloc_like, include/arrow_gpu.h "Pool placement")
  BufferPtr create_empty_buffer(uint64_t size, std::initializer_list<const Buffer*> like = {}) {  // [gpu_device.rs:183-192]
    auto b = std::make_shared<Buffer>();
    const void* nb[4];
    int n = 0;
    if (size >= (1ull << 30))
      for (const Buffer* x : like)
        if (x && x->ptr && n < 4) nb[n++] = x->ptr;
    if (n) check(agpu_malloc_like(raw, size, 0, nb, n, &b->ptr), "agpu_malloc_like");
    else check(agpu_malloc(raw, size ? size : 1, 0, &b->ptr), "agpu_malloc");
    b->bytes = size;
    b->dev = shared_from_this();
    return b;
  }
  // the buffers of one table in ONE block, placed for the HBM channel hash (agpu_malloc_table); each is freed on its own
  std::vector<BufferPtr> create_table_buffers(const std::vector<uint64_t>& sizes) {
    std::vector<void*> ptrs(sizes.size(), nullptr);
    check(agpu_malloc_table(raw, (int32_t)sizes.size(), sizes.data(), 0, ptrs.data()), "agpu_malloc_table");
    std::vector<BufferPtr> out;
    for (size_t k = 0; k < sizes.size(); k++) {
      auto b = std::make_shared<Buffer>();
      b->ptr = ptrs[k];
      b->bytes = sizes[k];
      b->dev = shared_from_this();
      out.push_back(b);
    }
    return out;
  }
  template <typename N>
  BufferPtr create_gpu_buffer_with_data(const N* data, size_t count) {  // [gpu_device.rs:171-181]
    auto b = create_empty_buffer(count * sizeof(N));
    if (count) {
      std::lock_guard<std::mutex> lock(io_mu);
      check(agpu_upload(io, b->ptr, data, count * sizeof(N)), "agpu_upload");
    }
    return b;
  }
  std::vector<uint8_t> retrive_data(const BufferPtr& b, uint64_t nbytes) {  // the only blocking call [gpu_device.rs:232-265]
    std::vector<uint8_t> out(nbytes);
    if (nbytes <= AGPU_MAILBOX_MAX_BYTES) {  // a scalar, a small array: the device-level wait delivers it — one wait instead of two
      check(agpu_device_download(raw, out.data(), nbytes ? b->ptr : nullptr, nbytes), "agpu_device_download");
      return out;
    }
    check(agpu_device_sync(raw), "agpu_device_sync");
    if (nbytes) {
      std::lock_guard<std::mutex> lock(io_mu);
      check(agpu_download(io, out.data(), b->ptr, nbytes), "agpu_download");
    }
    return out;
  }
  BufferPtr clone_buffer(const BufferPtr& b) {  // [gpu_device.rs:212-222]
    auto out = create_empty_buffer(b->bytes);
    std::lock_guard<std::mutex> lock(io_mu);
    if (b->bytes) check(agpu_copy(io, out->ptr, b->ptr, b->bytes), "agpu_copy");
    // record + submit, like the reference's queue.submit: `finish` publishes the copy to every other pipeline
    check(agpu_pipeline_finish(io), "agpu_pipeline_finish");
    return out;
  }

 private:
  GpuDevice() = default;
};
inline Buffer::~Buffer() {
  if (ptr && dev) agpu_free(dev->raw, ptr);
}
inline DevicePtr GPU_DEVICE() {  // static GPU_DEVICE: LazyLock<Arc<GpuDevice>> [crates/array/src/lib.rs:17]
  static DevicePtr* d = new DevicePtr(GpuDevice::create(0));  // intentionally leaked: outlives static destruction of HIP
  return *d;
}

class ArrowComputePipeline {  // [compute_pipeline.rs:8-300]; one HIP stream, ops run in recorded order
 public:
  DevicePtr device;
  agpu_pipeline* raw = nullptr;  // the stream handle; use h() for anything that must run after recorded ops
  std::vector<BufferPtr> keep;   // buffers referenced by in-flight work

  // fuse = true: same-width element-wise ops on f32 / i32 / u32 / Date32 columns are only RECORDED, like
  // commands in the reference's encoder, and issued at finish() / sync() / the next non-recordable op.  A run of ops
  // that each consume the previous result — whose array the caller already dropped, e.g.
  // `a.add_scalar_op(s, p).mul_scalar_op(s, p)` — becomes ONE agpu_fused_chain launch; intermediates that are still
  // referenced anywhere (use_count) are materialised as usual.
  bool fuse = false;
  struct Stats {
    size_t recorded = 0, kernels = 0, fused_chains = 0, fused_ops = 0;
  } stats;

  explicit ArrowComputePipeline(DevicePtr d, const char* /*label*/ = nullptr, bool fuse_ops = false)
      : device(std::move(d)), fuse(fuse_ops) {
    check(agpu_pipeline_create(device->raw, &raw), "agpu_pipeline_create");
  }
  ArrowComputePipeline(const ArrowComputePipeline&) = delete;
  ~ArrowComputePipeline() {
    if (raw) agpu_pipeline_destroy(raw);  // like an unsubmitted encoder, recorded-but-unfinished ops are dropped
  }
  agpu_pipeline* h() {  // handle for work that must be ordered after everything recorded so far
    flush();
    return raw;
  }
  void finish() { check(agpu_pipeline_finish(h()), "agpu_pipeline_finish"); }  // submit; does not wait
  void sync() {
    check(agpu_pipeline_sync(h()), "agpu_pipeline_sync");
    keep.clear();
  }
  BufferPtr clone_buffer(const BufferPtr& b, bool bitmap = false) {  // bitmap copies never depend on recorded value ops
    auto out = device->create_empty_buffer(b->bytes);
    if (b->bytes) check(agpu_copy(bitmap ? raw : h(), out->ptr, b->ptr, b->bytes), "agpu_copy");
    keep.push_back(b);
    keep.push_back(out);
    return out;
  }

  // ---- recording (fuse == true)
  static bool recordable(int kind, int op, agpu_dtype dtype) {
    const bool is_f = dtype == AGPU_F32;
    if (!(is_f || dtype == AGPU_I32 || dtype == AGPU_U32 || dtype == AGPU_DATE32)) return false;
    if (kind == AGPU_CHAIN_UNARY)
      return op == AGPU_UN_NEG || op == AGPU_UN_ABS || (is_f ? (op >= AGPU_UN_SQRT && op <= AGPU_UN_SINH) : op == AGPU_UN_NOT);
    return (op >= AGPU_OP_ADD && op <= AGPU_OP_MAX) || (!is_f && op >= AGPU_OP_AND && op <= AGPU_OP_XOR);
  }
  void record(int kind, int op, agpu_dtype dtype, BufferPtr a, BufferPtr operand, BufferPtr out, size_t n) {
    pending_.push_back(Node{kind, op, dtype, std::move(a), std::move(operand), std::move(out), n});
    stats.recorded++;
  }
  // a widening cast u8 / i8 / u16 / i16 → f32: may only START a chain (`cast → sin`); `op` carries the source dtype
  static constexpr int kCastNode = 4;
  void record_cast(agpu_dtype from, BufferPtr a, BufferPtr out, size_t n) {
    pending_.push_back(Node{kCastNode, (int)from, AGPU_F32, std::move(a), nullptr, std::move(out), n});
    stats.recorded++;
  }
  void flush() {
    if (pending_.empty()) return;
    std::vector<Node> nodes;
    nodes.swap(pending_);
    std::exception_ptr first_error;
    size_t i = 0;
    while (i < nodes.size()) {
      size_t len = 1, arrays = 0;
      while (len < AGPU_CHAIN_MAX_STEPS && i + len < nodes.size()) {
        const Node& last = nodes[i + len - 1];
        const Node& nxt = nodes[i + len];
        // dead intermediate: only `last.out` and `nxt.a` still reference the buffer (the caller dropped the array, no
        // later node reads it, nothing else keeps it alive)
        const bool dead = last.out.use_count() == 2;
        // behind a cast head the kernel reads at most AGPU_CAST_CHAIN_MAX_ARRAYS array operands: the chain is cut in front of
        // the next one (its intermediate is materialised, a plain chain starts there)
        const bool room = !(nodes[i].kind == kCastNode && nxt.kind == AGPU_CHAIN_ARRAY && arrays >= AGPU_CAST_CHAIN_MAX_ARRAYS);
        if (nxt.a == last.out && nxt.operand != last.out && nxt.n == last.n && nxt.dtype == last.dtype && nxt.kind != kCastNode && room &&
            dead) {
          if (nxt.kind == AGPU_CHAIN_ARRAY) arrays++;
          len++;
        } else {
          break;
        }
      }
      // a failing launch does not take the rest of the recording with it: the caller issued every op, so the remaining
      // chains still run and the first error is thrown once the list is empty
      try {
        launch_chain(nodes, i, len);
      } catch (...) {
        if (!first_error) first_error = std::current_exception();
      }
      for (size_t k = 0; k < len; k++) {
        keep.push_back(nodes[i + k].a);
        if (nodes[i + k].operand) keep.push_back(nodes[i + k].operand);
        keep.push_back(nodes[i + k].out);
      }
      i += len;
    }
    if (first_error) std::rethrow_exception(first_error);
  }

 private:
  struct Node;
  void launch_single(const Node& nd) {
    if (nd.kind == kCastNode)
      check(agpu_cast(raw, (agpu_dtype)nd.op, AGPU_F32, nd.a->ptr, nd.out->ptr, nd.n), "agpu_cast");
    else if (nd.kind == AGPU_CHAIN_UNARY)
      check(agpu_unary(raw, (agpu_unary_op)nd.op, nd.dtype, nd.a->ptr, nd.out->ptr, nd.n), "agpu_unary");
    else if (nd.kind == AGPU_CHAIN_SCALAR)
      check(agpu_scalar(raw, (agpu_binary_op)nd.op, nd.dtype, nd.a->ptr, nd.operand->ptr, nd.out->ptr, nd.n), "agpu_scalar");
    else
      check(agpu_binary(raw, (agpu_binary_op)nd.op, nd.dtype, nd.a->ptr, nd.operand->ptr, nd.out->ptr, nd.n), "agpu_binary");
    stats.kernels++;
  }
  void launch_chain(const std::vector<Node>& nodes, size_t i, size_t len) {
    if (len == 1) return launch_single(nodes[i]);
    const bool cast_head = nodes[i].kind == kCastNode;
    const size_t first = cast_head ? 1 : 0;
    std::vector<agpu_chain_step> steps(len - first);
    for (size_t k = first; k < len; k++)
      steps[k - first] = agpu_chain_step{nodes[i + k].op, nodes[i + k].kind, nodes[i + k].operand ? nodes[i + k].operand->ptr : nullptr};
    const agpu_status st =
        cast_head ? agpu_fused_cast_chain(raw, (agpu_dtype)nodes[i].op, nodes[i].a->ptr, steps.data(), (int32_t)steps.size(),
                                          static_cast<float*>(nodes[i + len - 1].out->ptr), nodes[i].n)
                  : agpu_fused_chain(raw, nodes[i].dtype, nodes[i].a->ptr, steps.data(), (int32_t)steps.size(), nodes[i + len - 1].out->ptr,
                                     nodes[i].n);
    if (st == AGPU_ERR_UNSUPPORTED) {  // a chain shape the fused kernels do not take: the recorded ops one by one
      for (size_t k = 0; k < len; k++) launch_single(nodes[i + k]);
      return;
    }
    check(st, cast_head ? "agpu_fused_cast_chain" : "agpu_fused_chain");
    stats.kernels++;
    stats.fused_chains++;
    stats.fused_ops += len;
  }
  struct Node {
    int kind, op;
    agpu_dtype dtype;
    BufferPtr a, operand, out;
    size_t n;
  };
  std::vector<Node> pending_;
};

// ------------------------------------------------------------------ bitmaps
inline uint64_t bitmap_bytes(uint64_t n_bits) { return (n_bits + 63) / 64 * 8; }

struct BooleanBufferBuilder {  // [crates/array/src/array/null_bit_buffer.rs:10-62]
  std::vector<uint8_t> data;
  size_t len = 0;
  bool contains_nulls = true;
  static BooleanBufferBuilder new_with_capacity(size_t size) {
    BooleanBufferBuilder b;
    b.data.assign((size + 7) / 8, 0);
    b.len = size;
    return b;
  }
  static BooleanBufferBuilder new_set_with_capacity(size_t size) {
    BooleanBufferBuilder b;
    b.data.assign((size + 7) / 8, 0xFF);
    if (size % 8) b.data.back() = (uint8_t)(0xFF >> (8 - size % 8));
    b.len = size;
    b.contains_nulls = false;
    return b;
  }
  void set_bit(size_t pos) { data[pos / 8] |= (uint8_t)(1u << (pos % 8)); }
  void unset_bit(size_t pos) { data[pos / 8] &= (uint8_t)~(1u << (pos % 8)); }
  bool is_set(size_t pos) const { return data[pos / 8] & (1u << (pos % 8)); }
  static bool is_set_in_slice(const uint8_t* d, size_t pos) { return d[pos / 8] & (1u << (pos % 8)); }
};

inline BufferPtr upload_bitmap(const DevicePtr& dev, const std::vector<uint8_t>& bytes, size_t n_bits) {
  std::vector<uint8_t> padded(bitmap_bytes(n_bits) ? bitmap_bytes(n_bits) : 8, 0);
  std::memcpy(padded.data(), bytes.data(), bytes.size());
  return dev->create_gpu_buffer_with_data(padded.data(), padded.size());
}

struct NullBitBufferGpu {  // [null_bit_buffer.rs:92-243]
  BufferPtr bit_buffer;
  size_t len = 0;
  DevicePtr gpu_device;
  // the kernel that produced the bitmap may have left a count behind (device u64; count_is_set_bits: set bits, else nulls):
  // null_count() is then an 8-byte read-back instead of a pass over the bitmap
  BufferPtr count_buf;
  bool count_is_set_bits = false;
  bool null_count_known() const { return (bool)count_buf; }
  uint64_t null_count() const {  // blocking; no reference counterpart (countob + Sum, boolean.rs:120-146, is its only bit count)
    BufferPtr c = count_buf;
    bool set_bits = count_is_set_bits;
    if (!c) {
      c = gpu_device->create_empty_buffer(8);
      set_bits = true;
      std::lock_guard<std::mutex> lk(gpu_device->io_mu);
      check(agpu_bitmap_popcount(gpu_device->io, bit_buffer->ptr, len, static_cast<uint64_t*>(c->ptr)), "agpu_bitmap_popcount");
    }
    check(agpu_device_sync(gpu_device->raw), "agpu_device_sync");
    const auto raw = gpu_device->retrive_data(c, 8);
    uint64_t v = 0;
    std::memcpy(&v, raw.data(), 8);
    return set_bits ? (uint64_t)len - v : v;
  }
  static std::optional<NullBitBufferGpu> make(const DevicePtr& dev, const BooleanBufferBuilder& b) {
    if (!b.contains_nulls) return std::nullopt;
    return NullBitBufferGpu{upload_bitmap(dev, b.data, b.len), b.len, dev};
  }
  std::vector<uint8_t> raw_values() const {
    auto raw = gpu_device->retrive_data(bit_buffer, (len + 7) / 8);
    return raw;
  }
  static std::optional<NullBitBufferGpu> clone_null_bit_buffer_op(const std::optional<NullBitBufferGpu>& d,
                                                                   ArrowComputePipeline& p) {
    if (!d) return std::nullopt;
    return NullBitBufferGpu{p.clone_buffer(d->bit_buffer, true), d->len, d->gpu_device};
  }
  // (None,None)→None; one side → copy; both → AND  [null_bit_buffer.rs:206-243]
  static std::optional<NullBitBufferGpu> merge_null_bit_buffer_op(const std::optional<NullBitBufferGpu>& l,
                                                                   const std::optional<NullBitBufferGpu>& r,
                                                                   ArrowComputePipeline& p) {
    if (!l && !r) return std::nullopt;
    if (!l || !r) return clone_null_bit_buffer_op(l ? l : r, p);
    auto out = l->gpu_device->create_empty_buffer(l->bit_buffer->bytes);
    auto cnt = l->gpu_device->create_empty_buffer(8);  // set bits of the result, counted by the waves that store it
    check(agpu_bitmap_binary_count(p.raw, AGPU_OP_AND, l->bit_buffer->ptr, r->bit_buffer->ptr, out->ptr, l->len,
                                   static_cast<uint64_t*>(cnt->ptr)), "bitmap and");
    p.keep.insert(p.keep.end(), {l->bit_buffer, r->bit_buffer, out, cnt});
    return NullBitBufferGpu{out, l->len, l->gpu_device, cnt, true};
  }
  static std::optional<NullBitBufferGpu> merge_null_bit_buffer(const std::optional<NullBitBufferGpu>& l,
                                                                const std::optional<NullBitBufferGpu>& r) {
    if (!l && !r) return std::nullopt;
    ArrowComputePipeline p((l ? l : r)->gpu_device);
    auto out = merge_null_bit_buffer_op(l, r, p);
    p.finish();
    return out;
  }
};

// ------------------------------------------------------------------ primitive types
struct Date32Type {};  // i32 storage [crates/array/src/array/date32_gpu.rs]
template <typename T> struct Prim;
#define AGPU_PRIM(T, NATIVE, CODE, TYPE)                       \
  template <> struct Prim<T> {                                 \
    using Native = NATIVE;                                     \
    static constexpr agpu_dtype dtype = CODE;                  \
    static constexpr ArrowType arrow_type = ArrowType::TYPE;   \
  };
AGPU_PRIM(float, float, AGPU_F32, Float32Type)
AGPU_PRIM(uint32_t, uint32_t, AGPU_U32, UInt32Type)
AGPU_PRIM(uint16_t, uint16_t, AGPU_U16, UInt16Type)
AGPU_PRIM(uint8_t, uint8_t, AGPU_U8, UInt8Type)
AGPU_PRIM(int32_t, int32_t, AGPU_I32, Int32Type)
AGPU_PRIM(int16_t, int16_t, AGPU_I16, Int16Type)
AGPU_PRIM(int8_t, int8_t, AGPU_I8, Int8Type)
AGPU_PRIM(Date32Type, int32_t, AGPU_DATE32, Date32Type)
#undef AGPU_PRIM

template <typename T, typename... Ts> inline constexpr bool is_one_of = (std::is_same_v<T, Ts> || ...);
template <typename T> inline constexpr bool is_int32ish = is_one_of<T, int32_t, uint32_t, Date32Type>;
template <typename T> inline constexpr bool is_small_int = is_one_of<T, int16_t, uint16_t, int8_t, uint8_t>;
template <typename T> inline constexpr bool is_int_type = is_int32ish<T> || is_small_int<T>;

class BooleanArrayGPU;

template <typename T>
class PrimitiveArrayGpu {  // [crates/array/src/array/primitive_array_gpu.rs:12-117]
 public:
  using Native = typename Prim<T>::Native;
  static constexpr agpu_dtype DTYPE = Prim<T>::dtype;
  BufferPtr data;
  DevicePtr gpu_device;
  size_t len = 0;
  std::optional<NullBitBufferGpu> null_buffer;

  PrimitiveArrayGpu() = default;
  PrimitiveArrayGpu(BufferPtr d, DevicePtr dev, size_t n, std::optional<NullBitBufferGpu> nb)
      : data(std::move(d)), gpu_device(std::move(dev)), len(n), null_buffer(std::move(nb)) {}

  static PrimitiveArrayGpu from_optional_slice(const std::vector<std::optional<Native>>& v, const DevicePtr& dev) {
    std::vector<Native> host(v.size(), Native());
    auto nulls = BooleanBufferBuilder::new_with_capacity(v.size());
    for (size_t i = 0; i < v.size(); i++)
      if (v[i]) {
        host[i] = *v[i];
        nulls.set_bit(i);
      }
    return PrimitiveArrayGpu(dev->create_gpu_buffer_with_data(host.data(), host.size()), dev, v.size(),
                             NullBitBufferGpu::make(dev, nulls));
  }
  static PrimitiveArrayGpu from_slice(const std::vector<Native>& v, const DevicePtr& dev) {
    return PrimitiveArrayGpu(dev->create_gpu_buffer_with_data(v.data(), v.size()), dev, v.size(), std::nullopt);
  }
  std::vector<Native> raw_values() const {
    auto raw = gpu_device->retrive_data(data, len * sizeof(Native));
    std::vector<Native> out(len);
    if (len) std::memcpy(out.data(), raw.data(), len * sizeof(Native));
    return out;
  }
  std::vector<std::optional<Native>> values() const {
    auto raw = raw_values();
    std::vector<std::optional<Native>> out(len);
    std::vector<uint8_t> nulls;
    if (null_buffer) nulls = null_buffer->raw_values();
    for (size_t i = 0; i < len; i++)
      if (!null_buffer || BooleanBufferBuilder::is_set_in_slice(nulls.data(), i)) out[i] = raw[i];
    return out;
  }
  // Arrow C Data Interface: any producer's ArrowArray / ArrowSchema pair → HBM, offsets and bit offsets of
  // sliced arrays resolved on the way in; the pair stays the caller's to release.  The reference can only copy through
  // host Vecs [primitive_array_gpu.rs:22-104].
  static PrimitiveArrayGpu from_arrow_c(const struct ArrowArray* array, const struct ArrowSchema* schema, const DevicePtr& dev) {
    ArrowComputePipeline p(dev);
    agpu_arrow_column col;
    check(agpu_import_arrow(p.h(), array, schema, &col), "agpu_import_arrow");
    auto take = [&](void* ptr, uint64_t bytes) {
      auto b = std::make_shared<Buffer>();
      b->ptr = ptr;
      b->bytes = bytes;
      b->dev = dev;
      return b;
    };
    BufferPtr values = take(col.values, col.values_bytes);
    std::optional<NullBitBufferGpu> nulls;
    if (col.validity) nulls = NullBitBufferGpu{take(col.validity, col.validity_bytes), (size_t)col.length, dev};
    const bool same = col.dtype == DTYPE || (col.dtype == AGPU_I32 && DTYPE == AGPU_DATE32) || (col.dtype == AGPU_DATE32 && DTYPE == AGPU_I32);
    if (!same) throw ArrowErrorGPU(ArrowErrorGPU::CastingNotSupported, "Arrow format does not match this array type");
    p.finish();
    p.sync();
    return PrimitiveArrayGpu(values, dev, (size_t)col.length, nulls);
  }
  // → host buffers behind an ArrowArray / ArrowSchema pair whose release callbacks free them (the consumer calls release)
  void to_arrow_c(struct ArrowArray* out_array, struct ArrowSchema* out_schema) const {
    ArrowComputePipeline p(gpu_device);
    agpu_arrow_column col{};
    col.dtype = DTYPE;
    col.length = len;
    col.null_count = null_buffer ? -1 : 0;
    col.values = data->ptr;
    col.values_bytes = data->bytes;
    if (null_buffer) {
      col.validity = null_buffer->bit_buffer->ptr;
      col.validity_bytes = null_buffer->bit_buffer->bytes;
    }
    check(agpu_device_sync(gpu_device->raw), "agpu_device_sync");  // other pipelines may still be writing this array
    if (null_buffer && null_buffer->null_count_known()) col.null_count = (int64_t)null_buffer->null_count();  // 8-byte read, no pass
    check(agpu_export_arrow(p.h(), &col, out_array, out_schema), "agpu_export_arrow");
  }
  PrimitiveArrayGpu clone_array() const {
    ArrowComputePipeline p(gpu_device);
    auto out = PrimitiveArrayGpu(p.clone_buffer(data), gpu_device, len, NullBitBufferGpu::clone_null_bit_buffer_op(null_buffer, p));
    p.finish();
    return out;
  }
  ArrowType get_dtype() const { return Prim<T>::arrow_type; }

  // Broadcast<T> [crates/array/src/kernels/broadcast.rs:6-17]
  static PrimitiveArrayGpu broadcast_op(Native value, size_t n, ArrowComputePipeline& p) {
    auto out = p.device->create_empty_buffer(n * sizeof(Native));
    uint32_t bits = 0;
    std::memcpy(&bits, &value, sizeof(Native));
    check(agpu_broadcast(p.h(), DTYPE, bits, out->ptr, n), "agpu_broadcast");
    p.keep.push_back(out);
    return PrimitiveArrayGpu(out, p.device, n, std::nullopt);
  }
  static PrimitiveArrayGpu broadcast(Native value, size_t n, const DevicePtr& dev) {
    ArrowComputePipeline p(dev);
    auto a = broadcast_op(value, n, p);
    p.finish();
    return a;
  }

  // ---- generic launch helpers (impl_arithmetic_array_op!, impl_arithmetic_op!, apply_unary_function_op!)
  template <typename Rhs>
  PrimitiveArrayGpu binary_op_(agpu_binary_op op, const PrimitiveArrayGpu<Rhs>& v, ArrowComputePipeline& p) const {
    if (len != v.len) throw ArrowErrorGPU(ArrowErrorGPU::Runtime, "binary op: arrays of different length");
    auto out = gpu_device->create_empty_buffer(len * sizeof(Native), {data.get(), v.data.get()});
    if (p.fuse && sizeof(typename Prim<Rhs>::Native) == sizeof(Native) && ArrowComputePipeline::recordable(AGPU_CHAIN_ARRAY, op, DTYPE)) {
      p.record(AGPU_CHAIN_ARRAY, op, DTYPE, data, v.data, out, len);
    } else {
      check(agpu_binary(p.h(), op, DTYPE, data->ptr, v.data->ptr, out->ptr, len), "agpu_binary");
      p.keep.insert(p.keep.end(), {data, v.data, out});
    }
    return PrimitiveArrayGpu(out, gpu_device, len, NullBitBufferGpu::merge_null_bit_buffer_op(null_buffer, v.null_buffer, p));
  }
  template <typename Rhs>
  PrimitiveArrayGpu scalar_op_(agpu_binary_op op, const PrimitiveArrayGpu<Rhs>& v, ArrowComputePipeline& p) const {
    auto out = gpu_device->create_empty_buffer(len * sizeof(Native), {data.get()});
    if (p.fuse && sizeof(typename Prim<Rhs>::Native) == sizeof(Native) && ArrowComputePipeline::recordable(AGPU_CHAIN_SCALAR, op, DTYPE)) {
      p.record(AGPU_CHAIN_SCALAR, op, DTYPE, data, v.data, out, len);
    } else {
      check(agpu_scalar(p.h(), op, DTYPE, data->ptr, v.data->ptr, out->ptr, len), "agpu_scalar");
      p.keep.insert(p.keep.end(), {data, v.data, out});
    }
    return PrimitiveArrayGpu(out, gpu_device, len, NullBitBufferGpu::clone_null_bit_buffer_op(null_buffer, p));
  }
  template <typename Out = T>
  PrimitiveArrayGpu<Out> unary_op_(agpu_unary_op op, ArrowComputePipeline& p) const {
    auto out = gpu_device->create_empty_buffer(len * sizeof(typename Prim<Out>::Native), {data.get()});
    if (p.fuse && std::is_same_v<Out, T> && ArrowComputePipeline::recordable(AGPU_CHAIN_UNARY, op, DTYPE)) {
      p.record(AGPU_CHAIN_UNARY, op, DTYPE, data, nullptr, out, len);
    } else {
      check(agpu_unary(p.h(), op, DTYPE, data->ptr, out->ptr, len), "agpu_unary");
      p.keep.insert(p.keep.end(), {data, out});
    }
    return PrimitiveArrayGpu<Out>(out, gpu_device, len, NullBitBufferGpu::clone_null_bit_buffer_op(null_buffer, p));
  }
  BooleanArrayGPU compare_op_(agpu_cmp_op op, const PrimitiveArrayGpu& v, ArrowComputePipeline& p) const;

#define AGPU_DEFAULT_IMPL(NAME, ARGS_DECL, ARGS_USE)  \
  auto NAME ARGS_DECL const {                         \
    ArrowComputePipeline p(gpu_device);               \
    auto out = NAME##_op ARGS_USE;                    \
    p.finish();                                       \
    return out;                                       \
  }
  // ---- arrow_gpu_arithmetic [crates/arithmetic/src/arithmetic_kernels.rs; f32.rs, i32.rs, u32.rs, u16.rs]
  template <typename R> auto add_op(const PrimitiveArrayGpu<R>& v, ArrowComputePipeline& p) const {
    static_assert((std::is_same_v<T, R> && (std::is_same_v<T, float> || is_int32ish<T>)) ||
                      (is_one_of<T, int32_t, Date32Type> && is_one_of<R, int32_t, Date32Type>),
                  "ArrowAdd is implemented for f32, u32, i32, Date32 (and i32<->Date32)");
    return binary_op_(AGPU_OP_ADD, v, p);
  }
  auto sub_op(const PrimitiveArrayGpu& v, ArrowComputePipeline& p) const {
    static_assert(std::is_same_v<T, float> || is_int32ish<T>, "ArrowSub: f32 (reference) + 32-bit ints");
    return binary_op_(AGPU_OP_SUB, v, p);
  }
  auto mul_op(const PrimitiveArrayGpu& v, ArrowComputePipeline& p) const {
    static_assert(std::is_same_v<T, float> || is_int32ish<T>, "ArrowMul: f32 (reference) + 32-bit ints");
    return binary_op_(AGPU_OP_MUL, v, p);
  }
  auto div_op(const PrimitiveArrayGpu& v, ArrowComputePipeline& p) const {
    static_assert(std::is_same_v<T, float>, "ArrowDiv is implemented for f32 only");
    return binary_op_(AGPU_OP_DIV, v, p);
  }
  template <typename R> auto add_scalar_op(const PrimitiveArrayGpu<R>& v, ArrowComputePipeline& p) const {
    static_assert(std::is_same_v<T, float> || is_int32ish<T> || std::is_same_v<T, uint16_t>, "ArrowScalarAdd: f32, i32, Date32, u32, u16");
    return scalar_op_(AGPU_OP_ADD, v, p);
  }
#define AGPU_SCALAR(NAME, OP)                                                                                   \
  template <typename R> auto NAME##_op(const PrimitiveArrayGpu<R>& v, ArrowComputePipeline& p) const {          \
    static_assert(std::is_same_v<T, float> || is_int32ish<T>, #NAME ": f32, i32, u32, Date32");                 \
    return scalar_op_(OP, v, p);                                                                                \
  }                                                                                                             \
  template <typename R> AGPU_DEFAULT_IMPL(NAME, (const PrimitiveArrayGpu<R>& v), (v, p))
  AGPU_SCALAR(sub_scalar, AGPU_OP_SUB)
  AGPU_SCALAR(mul_scalar, AGPU_OP_MUL)
  AGPU_SCALAR(div_scalar, AGPU_OP_DIV)
  AGPU_SCALAR(rem_scalar, AGPU_OP_REM)
#undef AGPU_SCALAR
  template <typename R> AGPU_DEFAULT_IMPL(add, (const PrimitiveArrayGpu<R>& v), (v, p))
  template <typename R> AGPU_DEFAULT_IMPL(add_scalar, (const PrimitiveArrayGpu<R>& v), (v, p))
  AGPU_DEFAULT_IMPL(sub, (const PrimitiveArrayGpu& v), (v, p))
  AGPU_DEFAULT_IMPL(mul, (const PrimitiveArrayGpu& v), (v, p))
  AGPU_DEFAULT_IMPL(div, (const PrimitiveArrayGpu& v), (v, p))
  auto neg_op(ArrowComputePipeline& p) const {
    static_assert(std::is_same_v<T, float>, "Neg is implemented for f32");
    return unary_op_(AGPU_UN_NEG, p);
  }
  AGPU_DEFAULT_IMPL(neg, (), (p))
  // Sum → 1-element array, validity ignored, f32 in the reference's summation order [aggregate_kernels.rs:24-51]
  auto sum_op(ArrowComputePipeline& p) const {
    static_assert(is_one_of<T, float, int32_t, uint32_t>, "Sum32Bit: f32, i32, u32");
    auto out = gpu_device->create_empty_buffer(16);
    check(agpu_reduce(p.h(), AGPU_RED_SUM, DTYPE, data->ptr, nullptr, len, out->ptr), "agpu_reduce");
    p.keep.insert(p.keep.end(), {data, out});
    return PrimitiveArrayGpu(out, gpu_device, 1, std::nullopt);
  }
  AGPU_DEFAULT_IMPL(sum, (), (p))
  // sum (the reference's tree order) / min / max (Arrow's NaN rule) / f64-accumulated sum in ONE pass over the column (agpu_reduce_stats_f32:
  // no counterpart in the reference, whose only reduction is Sum); null slots contribute the identities.  The record stays on the device.
  struct Stats {
    BufferPtr data;
    DevicePtr gpu_device;
    agpu_f32_stats values() const {  // blocking
      const auto raw = gpu_device->retrive_data(data, sizeof(agpu_f32_stats));
      agpu_f32_stats s;
      std::memcpy(&s, raw.data(), sizeof(s));
      return s;
    }
  };
  Stats stats_op(ArrowComputePipeline& p) const {
    static_assert(std::is_same_v<T, float>, "one-pass statistics: f32");
    auto out = gpu_device->create_empty_buffer(32);
    const void* validity = null_buffer ? null_buffer->bit_buffer->ptr : nullptr;
    check(agpu_reduce_stats_f32(p.h(), static_cast<const float*>(data->ptr), validity, len, static_cast<agpu_f32_stats*>(out->ptr)), "agpu_reduce_stats_f32");
    p.keep.insert(p.keep.end(), {data, out});
    return Stats{out, gpu_device};
  }
  Stats stats() const {
    ArrowComputePipeline p(gpu_device);
    auto out = stats_op(p);
    p.finish();
    return out;
  }

  // ---- arrow_gpu_compare [crates/compare/src/lib.rs:41-172]
#define AGPU_CMP(NAME, OP)                                                                                  \
  BooleanArrayGPU NAME##_op(const PrimitiveArrayGpu& v, ArrowComputePipeline& p) const;                     \
  BooleanArrayGPU NAME(const PrimitiveArrayGpu& v) const;
  AGPU_CMP(gt, AGPU_CMP_GT)
  AGPU_CMP(gteq, AGPU_CMP_GTEQ)
  AGPU_CMP(lt, AGPU_CMP_LT)
  AGPU_CMP(lteq, AGPU_CMP_LTEQ)
  AGPU_CMP(eq, AGPU_CMP_EQ)
#undef AGPU_CMP
  auto max_op(const PrimitiveArrayGpu& v, ArrowComputePipeline& p) const { return binary_op_(AGPU_OP_MAX, v, p); }
  auto min_op(const PrimitiveArrayGpu& v, ArrowComputePipeline& p) const { return binary_op_(AGPU_OP_MIN, v, p); }
  AGPU_DEFAULT_IMPL(max, (const PrimitiveArrayGpu& v), (v, p))
  AGPU_DEFAULT_IMPL(min, (const PrimitiveArrayGpu& v), (v, p))

  // ---- arrow_gpu_logical [crates/logical/src/lib.rs:44-187]
#define AGPU_LOGICAL(NAME, OP)                                                                       \
  auto NAME##_op(const PrimitiveArrayGpu& v, ArrowComputePipeline& p) const {                        \
    static_assert(is_int_type<T> && !std::is_same_v<T, Date32Type>, #NAME ": integer arrays");       \
    return binary_op_(OP, v, p);                                                                     \
  }                                                                                                  \
  AGPU_DEFAULT_IMPL(NAME, (const PrimitiveArrayGpu& v), (v, p))
  AGPU_LOGICAL(bitwise_and, AGPU_OP_AND)
  AGPU_LOGICAL(bitwise_or, AGPU_OP_OR)
  AGPU_LOGICAL(bitwise_xor, AGPU_OP_XOR)
#undef AGPU_LOGICAL
  auto bitwise_not_op(ArrowComputePipeline& p) const {
    static_assert(is_int_type<T> && !std::is_same_v<T, Date32Type>, "bitwise_not: integer arrays");
    return unary_op_(AGPU_UN_NOT, p);
  }
  AGPU_DEFAULT_IMPL(bitwise_not, (), (p))
  auto bitwise_shl_op(const PrimitiveArrayGpu<uint32_t>& v, ArrowComputePipeline& p) const {
    static_assert(is_int_type<T> && !std::is_same_v<T, Date32Type>, "bitwise_shl: integer arrays");
    return binary_op_(AGPU_OP_SHL, v, p);
  }
  auto bitwise_shr_op(const PrimitiveArrayGpu<uint32_t>& v, ArrowComputePipeline& p) const {
    static_assert(is_int_type<T> && !std::is_same_v<T, Date32Type>, "bitwise_shr: integer arrays");
    return binary_op_(AGPU_OP_SHR, v, p);
  }
  AGPU_DEFAULT_IMPL(bitwise_shl, (const PrimitiveArrayGpu<uint32_t>& v), (v, p))
  AGPU_DEFAULT_IMPL(bitwise_shr, (const PrimitiveArrayGpu<uint32_t>& v), (v, p))

  // ---- arrow_gpu_math / arrow_gpu_trigonometry [crates/math/src/lib.rs:37-236, crates/trigonometry/src/lib.rs:22-137]
#define AGPU_FLOAT_UNARY(NAME, OP)                                                \
  auto NAME##_op(ArrowComputePipeline& p) const {                                 \
    static_assert(std::is_same_v<T, float>, #NAME " is implemented for f32");     \
    return unary_op_(OP, p);                                                      \
  }                                                                               \
  AGPU_DEFAULT_IMPL(NAME, (), (p))
  AGPU_FLOAT_UNARY(sqrt, AGPU_UN_SQRT)
  AGPU_FLOAT_UNARY(cbrt, AGPU_UN_CBRT)
  AGPU_FLOAT_UNARY(exp, AGPU_UN_EXP)
  AGPU_FLOAT_UNARY(exp2, AGPU_UN_EXP2)
  AGPU_FLOAT_UNARY(log, AGPU_UN_LOG)
  AGPU_FLOAT_UNARY(log2, AGPU_UN_LOG2)
  AGPU_FLOAT_UNARY(acos, AGPU_UN_ACOS)
#undef AGPU_FLOAT_UNARY
  auto abs_op(ArrowComputePipeline& p) const {
    static_assert(is_one_of<T, float, int32_t>, "abs: f32, i32");
    return unary_op_(AGPU_UN_ABS, p);
  }
  AGPU_DEFAULT_IMPL(abs, (), (p))
  auto power_op(const PrimitiveArrayGpu& v, ArrowComputePipeline& p) const {
    static_assert(is_one_of<T, float, int32_t>, "power: f32, i32");
    return binary_op_(AGPU_OP_POW, v, p);
  }
  AGPU_DEFAULT_IMPL(power, (const PrimitiveArrayGpu& v), (v, p))
#define AGPU_TRIG(NAME, OP)                                                                                    \
  PrimitiveArrayGpu<float> NAME##_op(ArrowComputePipeline& p) const {                                          \
    static_assert(std::is_same_v<T, float> || is_small_int<T>, #NAME ": f32 and the fused u8/i8/u16/i16 kernels"); \
    return unary_op_<float>(OP, p);                                                                            \
  }                                                                                                            \
  AGPU_DEFAULT_IMPL(NAME, (), (p))
  AGPU_TRIG(sin, AGPU_UN_SIN)
  AGPU_TRIG(cos, AGPU_UN_COS)
  AGPU_TRIG(sinh, AGPU_UN_SINH)
#undef AGPU_TRIG

  // ---- arrow_gpu_cast: `a.cast<Float32ArrayGPU>()` ≙ `<A as Cast<Float32ArrayGPU>>::cast(&a)` [crates/cast/src/lib.rs]
  template <typename OutArray> OutArray cast_op(ArrowComputePipeline& p) const {
    using O = typename OutArray::ElemTag;
    auto out = gpu_device->create_empty_buffer(len * sizeof(typename Prim<O>::Native));
    if (p.fuse && std::is_same_v<O, float> && is_small_int<T>) {  // a fusing pipeline records the widening cast (the head of a chain)
      p.record_cast(DTYPE, data, out, len);
    } else {
      agpu_status s = agpu_cast(p.h(), DTYPE, Prim<O>::dtype, data->ptr, out->ptr, len);
      if (s == AGPU_ERR_UNSUPPORTED) throw ArrowErrorGPU(ArrowErrorGPU::CastingNotSupported, agpu_last_error());
      check(s, "agpu_cast");
      p.keep.insert(p.keep.end(), {data, out});
    }
    return OutArray(out, gpu_device, len, NullBitBufferGpu::clone_null_bit_buffer_op(null_buffer, p));
  }
  template <typename OutArray> OutArray cast() const {
    ArrowComputePipeline p(gpu_device);
    auto out = cast_op<OutArray>(p);
    p.finish();
    return out;
  }
  // BitCast<T>: reinterpret the values, a device copy of the buffer [crates/cast/src/lib.rs:90-107, table :187-192: u32 → f32]
  template <typename OutArray> OutArray bitcast_op(ArrowComputePipeline& p) const {
    static_assert(std::is_same_v<T, uint32_t> && std::is_same_v<typename OutArray::ElemTag, float>, "BitCast is implemented for UInt32ArrayGPU → Float32ArrayGPU");
    return OutArray(p.clone_buffer(data), gpu_device, len, NullBitBufferGpu::clone_null_bit_buffer_op(null_buffer, p));
  }
  template <typename OutArray> OutArray bitcast() const {
    ArrowComputePipeline p(gpu_device);
    auto out = bitcast_op<OutArray>(p);
    p.finish();
    return out;
  }
  // UInt32ArrayGPU::create_broadcast_buffer(_op)(value, len, …) -> Buffer [crates/array/src/array/u32_gpu.rs:36-64]: a bare device
  // buffer of `len` copies of `value` (the reference's helper for index / count columns)
  static BufferPtr create_broadcast_buffer_op(uint32_t value, uint64_t n, ArrowComputePipeline& p) {
    static_assert(std::is_same_v<T, uint32_t>, "create_broadcast_buffer is a UInt32ArrayGPU function");
    auto out = p.device->create_empty_buffer(4 * n);
    check(agpu_broadcast(p.h(), AGPU_U32, value, out->ptr, n), "agpu_broadcast");
    p.keep.push_back(out);
    return out;
  }
  static BufferPtr create_broadcast_buffer(uint32_t value, uint64_t n, const DevicePtr& dev) {
    ArrowComputePipeline p(dev);
    auto out = create_broadcast_buffer_op(value, n, p);
    p.finish();
    return out;
  }
  using ElemTag = T;

  // ---- arrow_gpu_routines: Swizzle [crates/routines/src/lib.rs:28-171]
  PrimitiveArrayGpu merge_op(const PrimitiveArrayGpu& other, const BooleanArrayGPU& mask, ArrowComputePipeline& p) const;
  PrimitiveArrayGpu merge(const PrimitiveArrayGpu& other, const BooleanArrayGPU& mask) const;
  PrimitiveArrayGpu take_op(const PrimitiveArrayGpu<uint32_t>& indexes, ArrowComputePipeline& p) const;
  PrimitiveArrayGpu take(const PrimitiveArrayGpu<uint32_t>& indexes) const {
    ArrowComputePipeline p(gpu_device);
    auto out = take_op(indexes, p);
    p.finish();
    p.sync();
    return out;
  }
  void put_op(const PrimitiveArrayGpu<uint32_t>& src_indexes, PrimitiveArrayGpu& dst,
              const PrimitiveArrayGpu<uint32_t>& dst_indexes, ArrowComputePipeline& p) const {
    if (null_buffer || dst.null_buffer)
      throw ArrowErrorGPU(ArrowErrorGPU::OperationNotSupported, "put with null buffers is todo!() in the reference");
    check(agpu_put_bounded(p.h(), (int)sizeof(Native), data->ptr, len, (const uint32_t*)src_indexes.data->ptr, dst.data->ptr,
                           dst.len, (const uint32_t*)dst_indexes.data->ptr, src_indexes.len), "agpu_put_bounded");
    p.keep.insert(p.keep.end(), {data, src_indexes.data, dst.data, dst_indexes.data});
  }
  void put(const PrimitiveArrayGpu<uint32_t>& si, PrimitiveArrayGpu& dst, const PrimitiveArrayGpu<uint32_t>& di) const {
    ArrowComputePipeline p(gpu_device);
    put_op(si, dst, di, p);
    p.finish();
    p.sync();
  }
};

using Float32ArrayGPU = PrimitiveArrayGpu<float>;
using UInt32ArrayGPU = PrimitiveArrayGpu<uint32_t>;
using UInt16ArrayGPU = PrimitiveArrayGpu<uint16_t>;
using UInt8ArrayGPU = PrimitiveArrayGpu<uint8_t>;
using Int32ArrayGPU = PrimitiveArrayGpu<int32_t>;
using Int16ArrayGPU = PrimitiveArrayGpu<int16_t>;
using Int8ArrayGPU = PrimitiveArrayGpu<int8_t>;
using Date32ArrayGPU = PrimitiveArrayGpu<Date32Type>;

// ------------------------------------------------------------------ BooleanArrayGPU [crates/array/src/array/boolean_gpu.rs]
class BooleanArrayGPU {
 public:
  BufferPtr data;
  DevicePtr gpu_device;
  size_t len = 0;
  std::optional<NullBitBufferGpu> null_buffer;
  BooleanArrayGPU() = default;
  BooleanArrayGPU(BufferPtr d, DevicePtr dev, size_t n, std::optional<NullBitBufferGpu> nb)
      : data(std::move(d)), gpu_device(std::move(dev)), len(n), null_buffer(std::move(nb)) {}
  static BooleanArrayGPU from_optional_slice(const std::vector<std::optional<bool>>& v, const DevicePtr& dev) {
    auto buf = BooleanBufferBuilder::new_with_capacity(v.size());
    auto nulls = BooleanBufferBuilder::new_with_capacity(v.size());
    for (size_t i = 0; i < v.size(); i++)
      if (v[i]) {
        nulls.set_bit(i);
        if (*v[i]) buf.set_bit(i);
      }
    return BooleanArrayGPU(upload_bitmap(dev, buf.data, v.size()), dev, v.size(), NullBitBufferGpu::make(dev, nulls));
  }
  static BooleanArrayGPU from_slice(const std::vector<bool>& v, const DevicePtr& dev) {
    auto buf = BooleanBufferBuilder::new_with_capacity(v.size());
    for (size_t i = 0; i < v.size(); i++)
      if (v[i]) buf.set_bit(i);
    return BooleanArrayGPU(upload_bitmap(dev, buf.data, v.size()), dev, v.size(), std::nullopt);
  }
  // NB: like the reference, `len` is the BYTE count of the slice [boolean_gpu.rs:72-82]
  static BooleanArrayGPU from_bytes_slice(const std::vector<uint8_t>& bytes, const DevicePtr& dev) {
    return BooleanArrayGPU(upload_bitmap(dev, bytes, bytes.size() * 8), dev, bytes.size(), std::nullopt);
  }
  std::vector<bool> raw_values() const {
    auto raw = gpu_device->retrive_data(data, (len + 7) / 8);
    std::vector<bool> out(len);
    for (size_t i = 0; i < len; i++) out[i] = BooleanBufferBuilder::is_set_in_slice(raw.data(), i);
    return out;
  }
  std::vector<std::optional<bool>> values() const {
    auto raw = raw_values();
    std::vector<std::optional<bool>> out(len);
    std::vector<uint8_t> nulls;
    if (null_buffer) nulls = null_buffer->raw_values();
    for (size_t i = 0; i < len; i++)
      if (!null_buffer || BooleanBufferBuilder::is_set_in_slice(nulls.data(), i)) out[i] = raw[i];
    return out;
  }
  static BooleanArrayGPU broadcast(bool value, size_t n, const DevicePtr& dev) {
    ArrowComputePipeline p(dev);
    auto out = dev->create_empty_buffer(bitmap_bytes(n) ? bitmap_bytes(n) : 8);
    check(agpu_broadcast(p.h(), AGPU_BOOL, value ? 1u : 0u, out->ptr, n), "agpu_broadcast");
    p.finish();
    return BooleanArrayGPU(out, dev, n, std::nullopt);
  }
  ArrowType get_dtype() const { return ArrowType::BooleanType; }
  BooleanArrayGPU clone_array() const {  // [boolean_gpu.rs / array/mod.rs:159-171]
    ArrowComputePipeline p(gpu_device);
    auto out = BooleanArrayGPU(p.clone_buffer(data, true), gpu_device, len, NullBitBufferGpu::clone_null_bit_buffer_op(null_buffer, p));
    p.finish();
    return out;
  }

  // Logical / LogicalContains [crates/logical/src/boolean.rs:12-147]
  BooleanArrayGPU logical_(agpu_binary_op op, const BooleanArrayGPU& v, ArrowComputePipeline& p) const {
    auto out = gpu_device->create_empty_buffer(bitmap_bytes(len) ? bitmap_bytes(len) : 8);
    check(agpu_bitmap_binary(p.h(), op, data->ptr, v.data->ptr, out->ptr, len), "agpu_bitmap_binary");
    p.keep.insert(p.keep.end(), {data, v.data, out});
    return BooleanArrayGPU(out, gpu_device, len, NullBitBufferGpu::merge_null_bit_buffer_op(null_buffer, v.null_buffer, p));
  }
  BooleanArrayGPU bitwise_and_op(const BooleanArrayGPU& v, ArrowComputePipeline& p) const { return logical_(AGPU_OP_AND, v, p); }
  BooleanArrayGPU bitwise_or_op(const BooleanArrayGPU& v, ArrowComputePipeline& p) const { return logical_(AGPU_OP_OR, v, p); }
  BooleanArrayGPU bitwise_xor_op(const BooleanArrayGPU& v, ArrowComputePipeline& p) const { return logical_(AGPU_OP_XOR, v, p); }
  BooleanArrayGPU bitwise_not_op(ArrowComputePipeline& p) const {
    auto out = gpu_device->create_empty_buffer(bitmap_bytes(len) ? bitmap_bytes(len) : 8);
    check(agpu_bitmap_not(p.h(), data->ptr, out->ptr, len), "agpu_bitmap_not");
    p.keep.insert(p.keep.end(), {data, out});
    return BooleanArrayGPU(out, gpu_device, len, NullBitBufferGpu::clone_null_bit_buffer_op(null_buffer, p));
  }
  AGPU_DEFAULT_IMPL(bitwise_and, (const BooleanArrayGPU& v), (v, p))
  AGPU_DEFAULT_IMPL(bitwise_or, (const BooleanArrayGPU& v), (v, p))
  AGPU_DEFAULT_IMPL(bitwise_xor, (const BooleanArrayGPU& v), (v, p))
  AGPU_DEFAULT_IMPL(bitwise_not, (), (p))
  bool any() const {
    ArrowComputePipeline p(gpu_device);
    auto out = gpu_device->create_empty_buffer(16);
    check(agpu_bitmap_any(p.h(), data->ptr, len, (uint32_t*)out->ptr), "agpu_bitmap_any");
    p.sync();
    uint32_t v = 0;
    std::memcpy(&v, gpu_device->retrive_data(out, 4).data(), 4);
    return v != 0;
  }
  bool all() const {
    ArrowComputePipeline p(gpu_device);
    auto out = gpu_device->create_empty_buffer(16);
    check(agpu_bitmap_popcount(p.h(), data->ptr, len, (uint64_t*)out->ptr), "agpu_bitmap_popcount");
    p.sync();
    uint64_t v = 0;
    std::memcpy(&v, gpu_device->retrive_data(out, 8).data(), 8);
    return v == len;
  }
  PrimitiveArrayGpu<float> cast_f32() const {  // Cast<Float32ArrayGPU> for BooleanArrayGPU [cast/src/boolean_cast.rs]
    ArrowComputePipeline p(gpu_device);
    auto out = gpu_device->create_empty_buffer(len * 4);
    check(agpu_cast(p.h(), AGPU_BOOL, AGPU_F32, data->ptr, out->ptr, len), "agpu_cast");
    auto nb = NullBitBufferGpu::clone_null_bit_buffer_op(null_buffer, p);
    p.finish();
    return PrimitiveArrayGpu<float>(out, gpu_device, len, nb);
  }
};
#undef AGPU_DEFAULT_IMPL

// ---- out-of-class definitions that need BooleanArrayGPU complete
template <typename T>
BooleanArrayGPU PrimitiveArrayGpu<T>::compare_op_(agpu_cmp_op op, const PrimitiveArrayGpu& v, ArrowComputePipeline& p) const {
  if (len != v.len) throw ArrowErrorGPU(ArrowErrorGPU::Runtime, "compare: arrays of different length");
  const uint64_t nb = bitmap_bytes(len) ? bitmap_bytes(len) : 8;
  auto out = gpu_device->create_empty_buffer(nb);
  std::optional<NullBitBufferGpu> nulls;
  if (!null_buffer && !v.null_buffer) {
    check(agpu_compare(p.h(), op, DTYPE, data->ptr, v.data->ptr, out->ptr, len), "agpu_compare");
  } else {  // fused validity AND (the reference: a second, separately submitted dispatch)
    auto outv = gpu_device->create_empty_buffer(nb);
    auto cnt = gpu_device->create_empty_buffer(8);  // the result's null count: a by-product of the launch's validity blocks
    check(agpu_compare_validity_count(p.h(), op, DTYPE, data->ptr, v.data->ptr,
                                      null_buffer ? null_buffer->bit_buffer->ptr : nullptr,
                                      v.null_buffer ? v.null_buffer->bit_buffer->ptr : nullptr, out->ptr, outv->ptr, len,
                                      static_cast<uint64_t*>(cnt->ptr)),
          "agpu_compare_validity_count");
    nulls = NullBitBufferGpu{outv, len, gpu_device, cnt, false};
    p.keep.push_back(cnt);
    if (null_buffer) p.keep.push_back(null_buffer->bit_buffer);
    if (v.null_buffer) p.keep.push_back(v.null_buffer->bit_buffer);
    p.keep.push_back(outv);
  }
  p.keep.insert(p.keep.end(), {data, v.data, out});
  return BooleanArrayGPU(out, gpu_device, len, nulls);
}
#define AGPU_CMP_DEF(NAME, OP)                                                                                           \
  template <typename T>                                                                                                  \
  BooleanArrayGPU PrimitiveArrayGpu<T>::NAME##_op(const PrimitiveArrayGpu& v, ArrowComputePipeline& p) const {           \
    return compare_op_(OP, v, p);                                                                                        \
  }                                                                                                                      \
  template <typename T> BooleanArrayGPU PrimitiveArrayGpu<T>::NAME(const PrimitiveArrayGpu& v) const {                   \
    ArrowComputePipeline p(gpu_device);                                                                                  \
    auto out = NAME##_op(v, p);                                                                                          \
    p.finish();                                                                                                          \
    return out;                                                                                                          \
  }
AGPU_CMP_DEF(gt, AGPU_CMP_GT)
AGPU_CMP_DEF(gteq, AGPU_CMP_GTEQ)
AGPU_CMP_DEF(lt, AGPU_CMP_LT)
AGPU_CMP_DEF(lteq, AGPU_CMP_LTEQ)
AGPU_CMP_DEF(eq, AGPU_CMP_EQ)
#undef AGPU_CMP_DEF

inline std::optional<NullBitBufferGpu> take_null_buffer(const std::optional<NullBitBufferGpu>& nb,
                                                        const UInt32ArrayGPU& indexes, ArrowComputePipeline& p) {
  if (!nb) return std::nullopt;
  auto out = indexes.gpu_device->create_empty_buffer(bitmap_bytes(indexes.len) ? bitmap_bytes(indexes.len) : 8);
  check(agpu_take_bits(p.h(), nb->bit_buffer->ptr, nb->len, (const uint32_t*)indexes.data->ptr, out->ptr, indexes.len), "agpu_take_bits");
  p.keep.insert(p.keep.end(), {nb->bit_buffer, indexes.data, out});
  return NullBitBufferGpu{out, indexes.len, indexes.gpu_device};
}
// Index ranges are checked inside the take / put kernels (out-of-range read -> 0, write dropped, like WGSL robust
// access); the pipeline's next sync() throws.  The *_op forms never block, the default forms sync before returning.
template <typename T>
PrimitiveArrayGpu<T> PrimitiveArrayGpu<T>::take_op(const UInt32ArrayGPU& indexes, ArrowComputePipeline& p) const {
  auto out = gpu_device->create_empty_buffer(indexes.len * sizeof(Native));
  if (!null_buffer) {
    check(agpu_take(p.h(), (int)sizeof(Native), data->ptr, len, (const uint32_t*)indexes.data->ptr, out->ptr, indexes.len), "agpu_take");
    p.keep.insert(p.keep.end(), {data, indexes.data, out});
    return PrimitiveArrayGpu(out, gpu_device, indexes.len, std::nullopt);
  }
  // values and validity in ONE call (the reference: two dispatches, take.rs:9-55 + bool.rs:33-46): at bucketed sizes the
  // validity bit travels with the value
  auto outv = gpu_device->create_empty_buffer(bitmap_bytes(indexes.len) ? bitmap_bytes(indexes.len) : 8);
  check(agpu_take_validity(p.h(), (int)sizeof(Native), data->ptr, len, null_buffer->bit_buffer->ptr, (const uint32_t*)indexes.data->ptr,
                           out->ptr, outv->ptr, indexes.len), "agpu_take_validity");
  p.keep.insert(p.keep.end(), {data, null_buffer->bit_buffer, indexes.data, out, outv});
  return PrimitiveArrayGpu(out, gpu_device, indexes.len, NullBitBufferGpu{outv, indexes.len, gpu_device});
}
// take of the columns of ONE table (equal length, no validity bitmaps) by one index column: what `c.take_op(indexes, p)` per column gives,
// through agpu_take_columns — at pipeline sizes the index column's part of the merge-back take runs once for all columns (no counterpart in
// the reference, which takes array by array [routines/src/lib.rs:122-143]).  Columns with nulls: take_op per column.
template <typename T>
std::vector<PrimitiveArrayGpu<T>> take_columns_op(const std::vector<PrimitiveArrayGpu<T>>& columns, const UInt32ArrayGPU& indexes,
                                                  ArrowComputePipeline& p) {
  using Native = typename PrimitiveArrayGpu<T>::Native;
  std::vector<PrimitiveArrayGpu<T>> out;
  if (columns.empty()) return out;
  std::vector<int32_t> widths;
  std::vector<const void*> vals;
  std::vector<void*> outs;
  std::vector<BufferPtr> bufs;
  for (const auto& c : columns) {
    if (c.null_buffer || c.len != columns[0].len)
      throw ArrowErrorGPU(ArrowErrorGPU::OperationNotSupported, "take_columns: columns of one length without validity bitmaps");
    bufs.push_back(c.gpu_device->create_empty_buffer(indexes.len * sizeof(Native)));
    widths.push_back((int32_t)sizeof(Native));
    vals.push_back(c.data->ptr);
    outs.push_back(bufs.back()->ptr);
  }
  check(agpu_take_columns(p.h(), (int32_t)columns.size(), widths.data(), vals.data(), columns[0].len, (const uint32_t*)indexes.data->ptr,
                          outs.data(), indexes.len), "agpu_take_columns");
  p.keep.push_back(indexes.data);
  for (size_t c = 0; c < columns.size(); c++) {
    p.keep.insert(p.keep.end(), {columns[c].data, bufs[c]});
    out.emplace_back(bufs[c], columns[c].gpu_device, indexes.len, std::nullopt);
  }
  return out;
}
template <typename T>
std::vector<PrimitiveArrayGpu<T>> take_columns(const std::vector<PrimitiveArrayGpu<T>>& columns, const UInt32ArrayGPU& indexes) {
  if (columns.empty()) return {};
  ArrowComputePipeline p(columns[0].gpu_device);
  auto out = take_columns_op(columns, indexes, p);
  p.finish();
  p.sync();
  return out;
}
// validity of merge: ((v1 & m) | (v2 & ~m)) & v_mask in one kernel [crates/routines/src/merge.rs:17-86]
inline std::optional<NullBitBufferGpu> merge_null_buffers_op(const std::optional<NullBitBufferGpu>& a,
                                                             const std::optional<NullBitBufferGpu>& b,
                                                             const BooleanArrayGPU& mask, ArrowComputePipeline& p, size_t n) {
  if (!a && !b && !mask.null_buffer) return std::nullopt;
  auto out = mask.gpu_device->create_empty_buffer(bitmap_bytes(n) ? bitmap_bytes(n) : 8);
  check(agpu_bitmap_merge_validity(p.h(), a ? a->bit_buffer->ptr : nullptr, b ? b->bit_buffer->ptr : nullptr, mask.data->ptr,
                                   mask.null_buffer ? mask.null_buffer->bit_buffer->ptr : nullptr, out->ptr, n),
        "agpu_bitmap_merge_validity");
  p.keep.push_back(out);
  return NullBitBufferGpu{out, n, mask.gpu_device};
}
template <typename T>
PrimitiveArrayGpu<T> PrimitiveArrayGpu<T>::merge_op(const PrimitiveArrayGpu& other, const BooleanArrayGPU& mask,
                                                    ArrowComputePipeline& p) const {
  auto out = gpu_device->create_empty_buffer(len * sizeof(Native));
  check(agpu_merge(p.h(), (int)sizeof(Native), data->ptr, other.data->ptr, mask.data->ptr, out->ptr, len), "agpu_merge");
  p.keep.insert(p.keep.end(), {data, other.data, mask.data, out});
  return PrimitiveArrayGpu(out, gpu_device, len, merge_null_buffers_op(null_buffer, other.null_buffer, mask, p, len));
}
template <typename T>
PrimitiveArrayGpu<T> PrimitiveArrayGpu<T>::merge(const PrimitiveArrayGpu& other, const BooleanArrayGPU& mask) const {
  ArrowComputePipeline p(gpu_device);
  auto out = merge_op(other, mask, p);
  p.finish();
  return out;
}

// ------------------------------------------------------------------ enum ArrowArrayGPU + *_dyn [crates/array/src/array/mod.rs:104-186]
using ArrowArrayGPU = std::variant<Float32ArrayGPU, UInt32ArrayGPU, UInt16ArrayGPU, UInt8ArrayGPU, Int32ArrayGPU, Int16ArrayGPU,
                                   Int8ArrayGPU, Date32ArrayGPU, BooleanArrayGPU>;

inline DevicePtr get_gpu_device(const ArrowArrayGPU& a) {
  return std::visit([](const auto& x) { return x.gpu_device; }, a);
}
inline size_t len(const ArrowArrayGPU& a) {
  return std::visit([](const auto& x) { return x.len; }, a);
}
inline ArrowType get_dtype(const ArrowArrayGPU& a) {
  return std::visit([](const auto& x) { return x.get_dtype(); }, a);
}
template <typename A> const A& try_from(const ArrowArrayGPU& a) {  // TryFrom<ArrowArrayGPU>
  if (auto p = std::get_if<A>(&a)) return *p;
  throw ArrowErrorGPU(ArrowErrorGPU::CastingNotSupported, "could not cast ArrowArrayGPU into the requested array type");
}
[[noreturn]] inline void not_supported(const char* fn) {
  throw ArrowErrorGPU(ArrowErrorGPU::OperationNotSupported, std::string("Operation ") + fn + " not supported for these types");
}

// dyn_fn! with a same-type list: dispatch when both operands hold the same listed alternative
#define AGPU_DYN_SAME(NAME, METHOD, ...)                                                                        \
  inline ArrowArrayGPU NAME##_op_dyn(const ArrowArrayGPU& a, const ArrowArrayGPU& b, ArrowComputePipeline& p) { \
    return std::visit(                                                                                          \
        [&](const auto& x, const auto& y) -> ArrowArrayGPU {                                                    \
          using X = std::decay_t<decltype(x)>;                                                                  \
          using Y = std::decay_t<decltype(y)>;                                                                  \
          if constexpr (std::is_same_v<X, Y> && is_one_of<X, __VA_ARGS__>) return ArrowArrayGPU(x.METHOD(y, p)); \
          else not_supported(#NAME "_dyn");                                                                     \
        },                                                                                                      \
        a, b);                                                                                                  \
  }                                                                                                             \
  inline ArrowArrayGPU NAME##_dyn(const ArrowArrayGPU& a, const ArrowArrayGPU& b) {                             \
    ArrowComputePipeline p(get_gpu_device(a));                                                                  \
    auto out = NAME##_op_dyn(a, b, p);                                                                          \
    p.finish();                                                                                                 \
    return out;                                                                                                 \
  }
// [crates/arithmetic/src/arithmetic_kernels.rs:122-175, 225-260] (the i32<->Date32 mixes are reachable through the typed API)
AGPU_DYN_SAME(add_scalar, add_scalar_op, Float32ArrayGPU, Int32ArrayGPU, Date32ArrayGPU, UInt32ArrayGPU, UInt16ArrayGPU)
AGPU_DYN_SAME(sub_scalar, sub_scalar_op, Float32ArrayGPU, Int32ArrayGPU, UInt32ArrayGPU)
AGPU_DYN_SAME(mul_scalar, mul_scalar_op, Float32ArrayGPU, Int32ArrayGPU, UInt32ArrayGPU)
AGPU_DYN_SAME(div_scalar, div_scalar_op, Float32ArrayGPU, Int32ArrayGPU, UInt32ArrayGPU)
AGPU_DYN_SAME(rem_scalar, rem_scalar_op, Float32ArrayGPU, Int32ArrayGPU, UInt32ArrayGPU, Date32ArrayGPU)
AGPU_DYN_SAME(add_array, add_op, Float32ArrayGPU, UInt32ArrayGPU, Int32ArrayGPU, Date32ArrayGPU)
AGPU_DYN_SAME(sub_array, sub_op, Float32ArrayGPU)
AGPU_DYN_SAME(mul_array, mul_op, Float32ArrayGPU)
AGPU_DYN_SAME(div_array, div_op, Float32ArrayGPU)
// [crates/compare/src/lib.rs:174-334]
#define AGPU_ALL_PRIMS Float32ArrayGPU, UInt32ArrayGPU, UInt16ArrayGPU, UInt8ArrayGPU, Int32ArrayGPU, Int16ArrayGPU, Int8ArrayGPU, Date32ArrayGPU
AGPU_DYN_SAME(gt, gt_op, AGPU_ALL_PRIMS)
AGPU_DYN_SAME(gteq, gteq_op, AGPU_ALL_PRIMS)
AGPU_DYN_SAME(lt, lt_op, AGPU_ALL_PRIMS)
AGPU_DYN_SAME(lteq, lteq_op, AGPU_ALL_PRIMS)
AGPU_DYN_SAME(eq, eq_op, AGPU_ALL_PRIMS)
AGPU_DYN_SAME(max, max_op, AGPU_ALL_PRIMS)
AGPU_DYN_SAME(min, min_op, AGPU_ALL_PRIMS)
// [crates/logical/src/lib.rs:189-349]
#define AGPU_LOGICAL_TYPES Int32ArrayGPU, UInt32ArrayGPU, UInt16ArrayGPU, Int16ArrayGPU, UInt8ArrayGPU, Int8ArrayGPU, BooleanArrayGPU
AGPU_DYN_SAME(bitwise_and, bitwise_and_op, AGPU_LOGICAL_TYPES)
AGPU_DYN_SAME(bitwise_or, bitwise_or_op, AGPU_LOGICAL_TYPES)
AGPU_DYN_SAME(bitwise_xor, bitwise_xor_op, AGPU_LOGICAL_TYPES)
AGPU_DYN_SAME(power, power_op, Int32ArrayGPU, Float32ArrayGPU)
#undef AGPU_DYN_SAME

// add_dyn & co: array∘array when both or neither have len 1, else scalar with the len-1 side as the scalar
// [crates/arithmetic/src/arithmetic_kernels.rs:101-119]
#define AGPU_DYN_LEN(NAME)                                                                                      \
  inline ArrowArrayGPU NAME##_op_dyn(const ArrowArrayGPU& a, const ArrowArrayGPU& b, ArrowComputePipeline& p) { \
    const size_t x = len(a), y = len(b);                                                                        \
    if ((x == 1 && y == 1) || (x != 1 && y != 1)) return NAME##_array_op_dyn(a, b, p);                          \
    if (y == 1) return NAME##_scalar_op_dyn(a, b, p);                                                           \
    return NAME##_scalar_op_dyn(b, a, p);                                                                       \
  }                                                                                                             \
  inline ArrowArrayGPU NAME##_dyn(const ArrowArrayGPU& a, const ArrowArrayGPU& b) {                             \
    ArrowComputePipeline p(get_gpu_device(a));                                                                  \
    auto out = NAME##_op_dyn(a, b, p);                                                                          \
    p.finish();                                                                                                 \
    return out;                                                                                                 \
  }
AGPU_DYN_LEN(add)
AGPU_DYN_LEN(sub)
AGPU_DYN_LEN(mul)
AGPU_DYN_LEN(div)
#undef AGPU_DYN_LEN

#define AGPU_DYN_UNARY(NAME, METHOD, ...)                                                      \
  inline ArrowArrayGPU NAME##_op_dyn(const ArrowArrayGPU& a, ArrowComputePipeline& p) {        \
    return std::visit(                                                                         \
        [&](const auto& x) -> ArrowArrayGPU {                                                  \
          using X = std::decay_t<decltype(x)>;                                                 \
          if constexpr (is_one_of<X, __VA_ARGS__>) return ArrowArrayGPU(x.METHOD(p));          \
          else not_supported(#NAME "_dyn");                                                    \
        },                                                                                     \
        a);                                                                                    \
  }                                                                                            \
  inline ArrowArrayGPU NAME##_dyn(const ArrowArrayGPU& a) {                                    \
    ArrowComputePipeline p(get_gpu_device(a));                                                 \
    auto out = NAME##_op_dyn(a, p);                                                            \
    p.finish();                                                                                \
    return out;                                                                                \
  }
AGPU_DYN_UNARY(neg, neg_op, Float32ArrayGPU)
AGPU_DYN_UNARY(abs, abs_op, Float32ArrayGPU, Int32ArrayGPU)
AGPU_DYN_UNARY(sqrt, sqrt_op, Float32ArrayGPU)
AGPU_DYN_UNARY(cbrt, cbrt_op, Float32ArrayGPU)
AGPU_DYN_UNARY(exp, exp_op, Float32ArrayGPU)
AGPU_DYN_UNARY(exp2, exp2_op, Float32ArrayGPU)
AGPU_DYN_UNARY(log, log_op, Float32ArrayGPU)
AGPU_DYN_UNARY(log2, log2_op, Float32ArrayGPU)
AGPU_DYN_UNARY(acos, acos_op, Float32ArrayGPU)
AGPU_DYN_UNARY(sin, sin_op, Float32ArrayGPU, UInt16ArrayGPU, UInt8ArrayGPU, Int16ArrayGPU, Int8ArrayGPU)
AGPU_DYN_UNARY(cos, cos_op, Float32ArrayGPU, UInt16ArrayGPU, UInt8ArrayGPU, Int16ArrayGPU, Int8ArrayGPU)
AGPU_DYN_UNARY(sinh, sinh_op, Float32ArrayGPU, UInt16ArrayGPU, UInt8ArrayGPU, Int16ArrayGPU, Int8ArrayGPU)
AGPU_DYN_UNARY(bitwise_not, bitwise_not_op, AGPU_LOGICAL_TYPES)
#undef AGPU_DYN_UNARY
#undef AGPU_ALL_PRIMS
#undef AGPU_LOGICAL_TYPES

inline ArrowArrayGPU take_dyn(const ArrowArrayGPU& a, const UInt32ArrayGPU& idx) {  // [crates/routines/src/take.rs:58-94]
  return std::visit(
      [&](const auto& x) -> ArrowArrayGPU {
        using X = std::decay_t<decltype(x)>;
        if constexpr (is_one_of<X, Date32ArrayGPU, UInt32ArrayGPU, Int32ArrayGPU, Float32ArrayGPU>) return ArrowArrayGPU(x.take(idx));
        else not_supported("take_dyn");
      },
      a);
}

// ---- the remaining names of the reference's public surface
// enum ScalarArray [crates/array/src/utils/mod.rs:2-11] — what ArrowArrayGPU::get_raw_values returns [array/src/array/mod.rs:145-157]
using ScalarArray = std::variant<std::vector<float>, std::vector<uint32_t>, std::vector<uint16_t>, std::vector<uint8_t>, std::vector<int32_t>,
                                 std::vector<int16_t>, std::vector<int8_t>, std::vector<bool>>;
inline ScalarArray get_raw_values(const ArrowArrayGPU& a) {
  return std::visit([](const auto& x) -> ScalarArray { return ScalarArray(x.raw_values()); }, a);  // Date32 → Vec<i32>, as in the reference
}
inline ArrowArrayGPU clone_array(const ArrowArrayGPU& a) {  // [array/src/array/mod.rs:159-171]
  return std::visit([](const auto& x) -> ArrowArrayGPU { return ArrowArrayGPU(x.clone_array()); }, a);
}
// enum ScalarValue / enum Operand [crates/array/src/kernels/mod.rs:5-24]
struct ScalarValue {
  std::variant<float, uint32_t, uint16_t, uint8_t, int32_t, int16_t, int8_t, bool> v;
  static ScalarValue F32(float x) { return {x}; }
  static ScalarValue U32(uint32_t x) { return {x}; }
  static ScalarValue U16(uint16_t x) { return {x}; }
  static ScalarValue U8(uint8_t x) { return {x}; }
  static ScalarValue I32(int32_t x) { return {x}; }
  static ScalarValue I16(int16_t x) { return {x}; }
  static ScalarValue I8(int8_t x) { return {x}; }
  static ScalarValue BOOL(bool x) { return {x}; }
};
// broadcast_dyn / broadcast_op_dyn [array/src/array/mod.rs:189-219]
inline ArrowArrayGPU broadcast_op_dyn(const ScalarValue& value, size_t n, ArrowComputePipeline& p) {
  return std::visit(
      [&](auto x) -> ArrowArrayGPU {
        using X = decltype(x);
        if constexpr (std::is_same_v<X, bool>) {
          auto out = p.device->create_empty_buffer(bitmap_bytes(n) ? bitmap_bytes(n) : 8);
          check(agpu_broadcast(p.h(), AGPU_BOOL, x ? 1u : 0u, out->ptr, n), "agpu_broadcast");
          p.keep.push_back(out);
          return ArrowArrayGPU(BooleanArrayGPU(out, p.device, n, std::nullopt));
        } else {
          return ArrowArrayGPU(PrimitiveArrayGpu<X>::broadcast_op(x, n, p));
        }
      },
      value.v);
}
inline ArrowArrayGPU broadcast_dyn(const ScalarValue& value, size_t n, const DevicePtr& dev) {
  ArrowComputePipeline p(dev);
  auto out = broadcast_op_dyn(value, n, p);
  p.finish();
  return out;
}
struct Operand {  // enum Operand { Scalar(ScalarValue), Array(ArrowArrayGPU) }
  std::variant<ScalarValue, ArrowArrayGPU> v;
  static Operand Scalar(ScalarValue s) { return Operand{std::move(s)}; }
  static Operand Array(ArrowArrayGPU a) { return Operand{std::move(a)}; }
  bool is_scalar() const { return v.index() == 0; }
  // the array a `*_dyn` kernel takes for this operand: the array itself, or the scalar as a 1-element array (a 1-element array IS
  // the scalar form there: crates/arithmetic/src/arithmetic_kernels.rs:101-119)
  ArrowArrayGPU as_array(const DevicePtr& dev) const {
    if (!is_scalar()) return std::get<ArrowArrayGPU>(v);
    return broadcast_dyn(std::get<ScalarValue>(v), 1, dev);
  }
};
// cast_dyn / cast_op_dyn(from, &ArrowType) [crates/cast/src/lib.rs:135-185] and bitcast_dyn / bitcast_op_dyn [:187-218]
inline ArrowArrayGPU cast_op_dyn(const ArrowArrayGPU& from, ArrowType into, ArrowComputePipeline& p) {
  return std::visit(
      [&](const auto& x) -> ArrowArrayGPU {
        using X = std::decay_t<decltype(x)>;
        if constexpr (std::is_same_v<X, BooleanArrayGPU>) {
          if (into == ArrowType::Float32Type) {
            auto out = x.gpu_device->create_empty_buffer(x.len * 4);
            check(agpu_cast(p.h(), AGPU_BOOL, AGPU_F32, x.data->ptr, out->ptr, x.len), "agpu_cast");
            p.keep.insert(p.keep.end(), {x.data, out});
            return ArrowArrayGPU(Float32ArrayGPU(out, x.gpu_device, x.len, NullBitBufferGpu::clone_null_bit_buffer_op(x.null_buffer, p)));
          }
        } else if constexpr (!std::is_same_v<X, Date32ArrayGPU>) {
          switch (into) {  // the C ABI owns the table (agpu_cast → AGPU_ERR_UNSUPPORTED → CastingNotSupported)
            case ArrowType::Float32Type: return ArrowArrayGPU(x.template cast_op<Float32ArrayGPU>(p));
            case ArrowType::UInt32Type: return ArrowArrayGPU(x.template cast_op<UInt32ArrayGPU>(p));
            case ArrowType::UInt16Type: return ArrowArrayGPU(x.template cast_op<UInt16ArrayGPU>(p));
            case ArrowType::UInt8Type: return ArrowArrayGPU(x.template cast_op<UInt8ArrayGPU>(p));
            case ArrowType::Int32Type: return ArrowArrayGPU(x.template cast_op<Int32ArrayGPU>(p));
            case ArrowType::Int16Type: return ArrowArrayGPU(x.template cast_op<Int16ArrayGPU>(p));
            case ArrowType::Int8Type: return ArrowArrayGPU(x.template cast_op<Int8ArrayGPU>(p));
            default: break;
          }
        }
        throw ArrowErrorGPU(ArrowErrorGPU::CastingNotSupported, "Casting not supported for these types");
      },
      from);
}
inline ArrowArrayGPU cast_dyn(const ArrowArrayGPU& from, ArrowType into) {
  ArrowComputePipeline p(get_gpu_device(from));
  auto out = cast_op_dyn(from, into, p);
  p.finish();
  return out;
}
inline ArrowArrayGPU bitcast_op_dyn(const ArrowArrayGPU& from, ArrowType into, ArrowComputePipeline& p) {
  if (auto u = std::get_if<UInt32ArrayGPU>(&from); u && into == ArrowType::Float32Type)
    return ArrowArrayGPU(u->template bitcast_op<Float32ArrayGPU>(p));
  throw ArrowErrorGPU(ArrowErrorGPU::CastingNotSupported, "Casting not supported for these types");
}
inline ArrowArrayGPU bitcast_dyn(const ArrowArrayGPU& from, ArrowType into) {
  ArrowComputePipeline p(get_gpu_device(from));
  auto out = bitcast_op_dyn(from, into, p);
  p.finish();
  return out;
}
// cast::apply_boolean_unary_function(gpu_device, original_values, new_buffer_size, output_item_size, shader, entry_point, pipeline)
// [crates/cast/src/boolean_cast.rs:8-55]: a Boolean bitmap in, one invocation per OUTPUT element — the reference's literal call
// shape through the by-name seam (`shader`: the WGSL text, its "#hash:len" name or its path key, e.g. "cast/boolean/cast_f32")
inline BufferPtr apply_boolean_unary_function(const DevicePtr& gpu_device, const BufferPtr& original_values, uint64_t new_buffer_size,
                                              uint64_t output_item_size, const char* shader, const char* entry_point,
                                              ArrowComputePipeline& pipeline) {
  auto out = gpu_device->create_empty_buffer(new_buffer_size);
  check(agpu_memset(pipeline.h(), out->ptr, 0, new_buffer_size), "agpu_memset");  // create_empty_buffer of the reference is zero-filled
  const void* ins[1] = {original_values->ptr};
  const uint64_t sizes[1] = {original_values->bytes};
  const uint64_t dispatch = ((new_buffer_size + output_item_size - 1) / output_item_size + 255) / 256;
  check(agpu_launch_by_name_sized(pipeline.h(), shader, entry_point, ins, sizes, 1, out->ptr, new_buffer_size, (uint32_t)dispatch),
        "agpu_launch_by_name_sized");
  pipeline.keep.insert(pipeline.keep.end(), {original_values, out});
  return out;
}

// ---- fused element-wise chains: the `*_op` chain of examples/simple.rs:45-72 as ONE kernel.
//   auto out = FusedChain(a).add_scalar(s).mul_scalar(s).finish();      // (a + s) * s, 8 B/row instead of 16
// Each step uses the same arithmetic as the stand-alone op (bit-identical result); validity follows the reference's
// rules step by step (clone for unary / scalar steps, AND with every array operand's validity).
template <typename T>
class FusedChain {
  static_assert(std::is_same_v<T, float> || is_int32ish<T>, "fused chains: f32 / i32 / u32 / Date32 columns");
  using Arr = PrimitiveArrayGpu<T>;
  Arr src_;
  std::vector<agpu_chain_step> steps_;
  std::vector<Arr> operands_;  // keeps operand buffers (and their validity) alive until finish

  FusedChain& push(int op, int kind, const Arr* operand) {
    if (steps_.size() >= AGPU_CHAIN_MAX_STEPS) throw ArrowErrorGPU(ArrowErrorGPU::Runtime, "a fused chain holds at most 8 steps");
    if (kind == AGPU_CHAIN_ARRAY && operand->len != src_.len)
      throw ArrowErrorGPU(ArrowErrorGPU::Runtime, "fused chain: arrays of different length");
    steps_.push_back(agpu_chain_step{op, kind, operand ? operand->data->ptr : nullptr});
    if (operand) operands_.push_back(*operand);
    return *this;
  }

 public:
  explicit FusedChain(const Arr& a) : src_(a) {}
#define AGPU_CHAIN_BIN(NAME, OP)                                                                         \
  FusedChain& NAME(const Arr& v) { return push(OP, v.len == 1 && src_.len != 1 ? AGPU_CHAIN_SCALAR : AGPU_CHAIN_ARRAY, &v); } \
  FusedChain& NAME##_scalar(const Arr& v) { return push(OP, AGPU_CHAIN_SCALAR, &v); }
  AGPU_CHAIN_BIN(add, AGPU_OP_ADD)
  AGPU_CHAIN_BIN(sub, AGPU_OP_SUB)
  AGPU_CHAIN_BIN(mul, AGPU_OP_MUL)
  AGPU_CHAIN_BIN(div, AGPU_OP_DIV)
  AGPU_CHAIN_BIN(rem, AGPU_OP_REM)
  AGPU_CHAIN_BIN(min, AGPU_OP_MIN)
  AGPU_CHAIN_BIN(max, AGPU_OP_MAX)
#undef AGPU_CHAIN_BIN
  FusedChain& neg() { return push(AGPU_UN_NEG, AGPU_CHAIN_UNARY, nullptr); }
  FusedChain& abs() { return push(AGPU_UN_ABS, AGPU_CHAIN_UNARY, nullptr); }
#define AGPU_CHAIN_FLOAT_UNARY(NAME, OP)                                                  \
  FusedChain& NAME() {                                                                    \
    static_assert(std::is_same_v<T, float>, #NAME " in a fused chain needs an f32 column"); \
    return push(OP, AGPU_CHAIN_UNARY, nullptr);                                           \
  }
  AGPU_CHAIN_FLOAT_UNARY(sqrt, AGPU_UN_SQRT)
  AGPU_CHAIN_FLOAT_UNARY(cbrt, AGPU_UN_CBRT)
  AGPU_CHAIN_FLOAT_UNARY(exp, AGPU_UN_EXP)
  AGPU_CHAIN_FLOAT_UNARY(exp2, AGPU_UN_EXP2)
  AGPU_CHAIN_FLOAT_UNARY(log, AGPU_UN_LOG)
  AGPU_CHAIN_FLOAT_UNARY(log2, AGPU_UN_LOG2)
  AGPU_CHAIN_FLOAT_UNARY(sin, AGPU_UN_SIN)
  AGPU_CHAIN_FLOAT_UNARY(cos, AGPU_UN_COS)
  AGPU_CHAIN_FLOAT_UNARY(acos, AGPU_UN_ACOS)
  AGPU_CHAIN_FLOAT_UNARY(sinh, AGPU_UN_SINH)
#undef AGPU_CHAIN_FLOAT_UNARY

  Arr finish_op(ArrowComputePipeline& p) const {
    auto out = src_.gpu_device->create_empty_buffer(src_.len * sizeof(typename Arr::Native));
    std::optional<NullBitBufferGpu> nulls = src_.null_buffer;
    bool merged = false;
    size_t k = 0;
    for (const auto& st : steps_) {
      if (st.kind == AGPU_CHAIN_UNARY) continue;
      const Arr& operand = operands_[k++];
      p.keep.push_back(operand.data);
      if (st.kind == AGPU_CHAIN_ARRAY && operand.null_buffer) {
        nulls = NullBitBufferGpu::merge_null_bit_buffer_op(nulls, operand.null_buffer, p);
        merged = true;
      }
    }
    if (!merged) nulls = NullBitBufferGpu::clone_null_bit_buffer_op(nulls, p);
    check(agpu_fused_chain(p.h(), Arr::DTYPE, src_.data->ptr, steps_.data(), (int32_t)steps_.size(), out->ptr, src_.len),
          "agpu_fused_chain");
    p.keep.insert(p.keep.end(), {src_.data, out});
    return Arr(out, src_.gpu_device, src_.len, nulls);
  }
  Arr finish() const {
    ArrowComputePipeline p(src_.gpu_device);
    auto out = finish_op(p);
    p.finish();
    return out;
  }

  // terminal compare: the predicate chain(a) cmp other → BooleanArrayGPU, the chain's value is never stored
  BooleanArrayGPU compare_op(agpu_cmp_op op, const Arr& other, ArrowComputePipeline& p) const {
    if (steps_.size() >= AGPU_CHAIN_MAX_STEPS) throw ArrowErrorGPU(ArrowErrorGPU::Runtime, "at most 7 steps before a compare");
    const bool scalar = other.len == 1 && src_.len != 1;
    if (!scalar && other.len != src_.len) throw ArrowErrorGPU(ArrowErrorGPU::Runtime, "compare: arrays of different length");
    auto out = src_.gpu_device->create_empty_buffer(bitmap_bytes(src_.len) ? bitmap_bytes(src_.len) : 8);
    std::optional<NullBitBufferGpu> nulls = src_.null_buffer;
    bool merged = false;
    size_t k = 0;
    for (const auto& st : steps_) {
      if (st.kind == AGPU_CHAIN_UNARY) continue;
      const Arr& operand = operands_[k++];
      p.keep.push_back(operand.data);
      if (st.kind == AGPU_CHAIN_ARRAY && operand.null_buffer) {
        nulls = NullBitBufferGpu::merge_null_bit_buffer_op(nulls, operand.null_buffer, p);
        merged = true;
      }
    }
    if (!scalar && other.null_buffer) {
      nulls = NullBitBufferGpu::merge_null_bit_buffer_op(nulls, other.null_buffer, p);
      merged = true;
    }
    if (!merged) nulls = NullBitBufferGpu::clone_null_bit_buffer_op(nulls, p);
    check(agpu_fused_chain_compare(p.h(), Arr::DTYPE, src_.data->ptr, steps_.data(), (int32_t)steps_.size(), op,
                                   scalar ? AGPU_CHAIN_SCALAR : AGPU_CHAIN_ARRAY, other.data->ptr, out->ptr, src_.len),
          "agpu_fused_chain_compare");
    p.keep.insert(p.keep.end(), {src_.data, other.data, out});
    return BooleanArrayGPU(out, src_.gpu_device, src_.len, nulls);
  }
#define AGPU_CHAIN_CMP(NAME, OP)                                                                                  \
  BooleanArrayGPU NAME##_op(const Arr& v, ArrowComputePipeline& p) const { return compare_op(OP, v, p); }         \
  BooleanArrayGPU NAME(const Arr& v) const {                                                                      \
    ArrowComputePipeline p(src_.gpu_device);                                                                      \
    auto out = compare_op(OP, v, p);                                                                              \
    p.finish();                                                                                                   \
    return out;                                                                                                   \
  }
  AGPU_CHAIN_CMP(gt, AGPU_CMP_GT)
  AGPU_CHAIN_CMP(gteq, AGPU_CMP_GTEQ)
  AGPU_CHAIN_CMP(lt, AGPU_CMP_LT)
  AGPU_CHAIN_CMP(lteq, AGPU_CMP_LTEQ)
  AGPU_CHAIN_CMP(eq, AGPU_CMP_EQ)
#undef AGPU_CHAIN_CMP
};
template <typename T> FusedChain(const PrimitiveArrayGpu<T>&) -> FusedChain<T>;

// A chain with a WIDENING CAST at its head (agpu_fused_cast_chain): a u8 / i8 / u16 / i16 column in, f32 steps, an f32 column out —
//   auto y = FusedCastChain(u8_col).sin().finish();                         // the reference's fused sin_u8 [trigonometry/src/u8_kernel.rs:34-38]
//   auto z = FusedCastChain(u8_col).mul_scalar(scale).add_scalar(off).finish();  // 1 B/row in, 4 B/row out, nothing in between
// bit-identical to col.cast<Float32ArrayGPU>() followed by the same ops one by one; validity as in FusedChain.
template <typename T>
class FusedCastChain {
  static_assert(is_small_int<T>, "cast-headed chains start from u8 / i8 / u16 / i16 (the reference's casts to f32)");
  using Src = PrimitiveArrayGpu<T>;
  using Arr = PrimitiveArrayGpu<float>;
  Src src_;
  std::vector<agpu_chain_step> steps_;
  std::vector<Arr> operands_;

  FusedCastChain& push(int op, int kind, const Arr* operand) {
    if (steps_.size() >= AGPU_CHAIN_MAX_STEPS) throw ArrowErrorGPU(ArrowErrorGPU::Runtime, "a fused chain holds at most 8 steps");
    if (kind == AGPU_CHAIN_ARRAY && operand->len != src_.len)
      throw ArrowErrorGPU(ArrowErrorGPU::Runtime, "fused chain: arrays of different length");
    steps_.push_back(agpu_chain_step{op, kind, operand ? operand->data->ptr : nullptr});
    if (operand) operands_.push_back(*operand);
    return *this;
  }

 public:
  explicit FusedCastChain(const Src& a) : src_(a) {}
#define AGPU_CCHAIN_BIN(NAME, OP)                                                                                \
  FusedCastChain& NAME(const Arr& v) { return push(OP, v.len == 1 && src_.len != 1 ? AGPU_CHAIN_SCALAR : AGPU_CHAIN_ARRAY, &v); } \
  FusedCastChain& NAME##_scalar(const Arr& v) { return push(OP, AGPU_CHAIN_SCALAR, &v); }
  AGPU_CCHAIN_BIN(add, AGPU_OP_ADD)
  AGPU_CCHAIN_BIN(sub, AGPU_OP_SUB)
  AGPU_CCHAIN_BIN(mul, AGPU_OP_MUL)
  AGPU_CCHAIN_BIN(div, AGPU_OP_DIV)
  AGPU_CCHAIN_BIN(rem, AGPU_OP_REM)
  AGPU_CCHAIN_BIN(min, AGPU_OP_MIN)
  AGPU_CCHAIN_BIN(max, AGPU_OP_MAX)
#undef AGPU_CCHAIN_BIN
#define AGPU_CCHAIN_UNARY(NAME, OP) \
  FusedCastChain& NAME() { return push(OP, AGPU_CHAIN_UNARY, nullptr); }
  AGPU_CCHAIN_UNARY(neg, AGPU_UN_NEG)
  AGPU_CCHAIN_UNARY(abs, AGPU_UN_ABS)
  AGPU_CCHAIN_UNARY(sqrt, AGPU_UN_SQRT)
  AGPU_CCHAIN_UNARY(cbrt, AGPU_UN_CBRT)
  AGPU_CCHAIN_UNARY(exp, AGPU_UN_EXP)
  AGPU_CCHAIN_UNARY(exp2, AGPU_UN_EXP2)
  AGPU_CCHAIN_UNARY(log, AGPU_UN_LOG)
  AGPU_CCHAIN_UNARY(log2, AGPU_UN_LOG2)
  AGPU_CCHAIN_UNARY(sin, AGPU_UN_SIN)
  AGPU_CCHAIN_UNARY(cos, AGPU_UN_COS)
  AGPU_CCHAIN_UNARY(acos, AGPU_UN_ACOS)
  AGPU_CCHAIN_UNARY(sinh, AGPU_UN_SINH)
#undef AGPU_CCHAIN_UNARY

  Arr finish_op(ArrowComputePipeline& p) const {
    auto out = src_.gpu_device->create_empty_buffer(src_.len * sizeof(float));
    std::optional<NullBitBufferGpu> nulls = src_.null_buffer;
    bool merged = false;
    size_t k = 0;
    for (const auto& st : steps_) {
      if (st.kind == AGPU_CHAIN_UNARY) continue;
      const Arr& operand = operands_[k++];
      p.keep.push_back(operand.data);
      if (st.kind == AGPU_CHAIN_ARRAY && operand.null_buffer) {
        nulls = NullBitBufferGpu::merge_null_bit_buffer_op(nulls, operand.null_buffer, p);
        merged = true;
      }
    }
    if (!merged) nulls = NullBitBufferGpu::clone_null_bit_buffer_op(nulls, p);
    check(agpu_fused_cast_chain(p.h(), Src::DTYPE, src_.data->ptr, steps_.data(), (int32_t)steps_.size(), static_cast<float*>(out->ptr),
                                src_.len),
          "agpu_fused_cast_chain");
    p.keep.insert(p.keep.end(), {src_.data, out});
    return Arr(out, src_.gpu_device, src_.len, nulls);
  }
  Arr finish() const {
    ArrowComputePipeline p(src_.gpu_device);
    auto out = finish_op(p);
    p.finish();
    return out;
  }
};
template <typename T> FusedCastChain(const PrimitiveArrayGpu<T>&) -> FusedCastChain<T>;

// ------------------------------------------------------------------ chunk-sharded columns (north_star config 5)
// Not in the reference (one device, one queue: crates/array/src/gpu_utils/gpu_device.rs:29-33).  One host thread per
// GPU, each with its own GpuDevice, pipeline and Communicator rank; a column is a contiguous row range per GPU (cut with
// shard_rows on 512-row boundaries); every op of this header runs shard-local; only whole-column statistics finish over
// RCCL — one 16-byte record per rank, combined in rank order on every rank (include/arrow_gpu.h "multi-GPU").
struct Shard {
  int rank, world;
  uint64_t row0, rows;
};
inline Shard shard_rows(uint64_t total_rows, int world, int rank, uint64_t align = 512) {
  const uint64_t chunks = (total_rows + align - 1) / align, per = chunks / (uint64_t)world, extra = chunks % (uint64_t)world;
  const uint64_t r = (uint64_t)rank, first = r * per + (r < extra ? r : extra), cnt = per + (r < extra ? 1 : 0);
  const uint64_t lo = first * align < total_rows ? first * align : total_rows;
  const uint64_t hi = (first + cnt) * align < total_rows ? (first + cnt) * align : total_rows;
  return Shard{rank, world, lo, hi - lo};
}

class Communicator {
 public:
  using Id = std::array<uint8_t, AGPU_COMM_ID_BYTES>;
  agpu_comm* raw = nullptr;
  int rank = 0, world = 1;
  static Id unique_id() {  // rank 0 makes it; every rank passes the same bytes to the constructor
    Id id{};
    check(agpu_comm_get_unique_id(id.data()), "agpu_comm_get_unique_id");
    return id;
  }
  // collective: returns once all `world` ranks have arrived; timeout_ms < 0 = the library default (AGPU_COMM_TIMEOUT_MS,
  // 120 s), 0 = wait for ever.  A timeout throws; the process should then exit (the pending rendezvous cannot be cancelled)
  Communicator(const DevicePtr& dev, const Id& id, int rank_, int world_, int64_t timeout_ms = -1) : rank(rank_), world(world_) {
    if (timeout_ms < 0) check(agpu_comm_init_rank(dev->raw, id.data(), rank_, world_, &raw), "agpu_comm_init_rank");
    else check(agpu_comm_init_rank_timeout(dev->raw, id.data(), rank_, world_, timeout_ms, &raw), "agpu_comm_init_rank_timeout");
  }
  static std::string runtime_info() {
    char buf[1024];
    check(agpu_comm_runtime_info(buf, sizeof buf), "agpu_comm_runtime_info");
    return buf;
  }
  Communicator(const Communicator&) = delete;
  ~Communicator() {
    if (raw) agpu_comm_destroy(raw);
  }
  void barrier(ArrowComputePipeline& p) { check(agpu_comm_barrier(raw, p.h()), "agpu_comm_barrier"); }
  // the host wait that belongs behind a collective (reduce_sharded_op, all-reduce): like p.sync() but with the collective deadline;
  // a timeout throws, the device is poisoned (include/arrow_gpu.h) and every destructor on the way out returns without waiting
  void sync(ArrowComputePipeline& p) { check(agpu_comm_sync(raw, p.h()), "agpu_comm_sync"); }
  // true for a ONE-rank communicator whose RCCL bootstrap did not come up in time: no RCCL behind it, its collectives are device copies
  // (include/arrow_gpu.h agpu_comm_is_local) — a record that says "RCCL ran" must check this
  bool is_local() const {
    int32_t v = 0;
    check(agpu_comm_is_local(raw, &v), "agpu_comm_is_local");
    return v != 0;
  }
  // what RCCL reports (ncclCommCount), not what the launcher said
  int size() const {
    int32_t n = 0;
    check(agpu_comm_size(raw, &n, nullptr, nullptr), "agpu_comm_size");
    return n;
  }
  // collective: one identity record per rank gathered through the communicator; .second = distinct (host, PCI address) pairs —
  // equal to world exactly when every rank drives its own GPU
  std::pair<std::vector<agpu_comm_peer>, int> peers(ArrowComputePipeline& p) {
    std::vector<agpu_comm_peer> v((size_t)world);
    int32_t distinct = 0;
    check(agpu_comm_peers(raw, p.h(), v.data(), world, &distinct), "agpu_comm_peers");
    return {std::move(v), distinct};
  }
};

// Whole-column Sum / min / max of a sharded column → 1-element array holding the SAME value on every rank.
// f32 Sum: each shard in the reference's tree order, shard sums combined by one more adjacent-pair level.
template <typename T>
PrimitiveArrayGpu<T> reduce_sharded_op(const PrimitiveArrayGpu<T>& shard, agpu_reduce_op op, Communicator& comm,
                                       ArrowComputePipeline& p, bool use_validity = false) {
  static_assert(is_one_of<T, float, int32_t, uint32_t>, "32-bit statistics: f32, i32, u32");
  auto out = shard.gpu_device->create_empty_buffer(16);
  const void* validity = use_validity && shard.null_buffer ? shard.null_buffer->bit_buffer->ptr : nullptr;
  check(agpu_comm_reduce(comm.raw, p.h(), op, Prim<T>::dtype, shard.data->ptr, validity, shard.len, out->ptr), "agpu_comm_reduce");
  p.keep.insert(p.keep.end(), {shard.data, out});
  return PrimitiveArrayGpu<T>(out, shard.gpu_device, 1, std::nullopt);
}
template <typename T> PrimitiveArrayGpu<T> sum_sharded_op(const PrimitiveArrayGpu<T>& s, Communicator& c, ArrowComputePipeline& p) {
  return reduce_sharded_op(s, AGPU_RED_SUM, c, p);
}
template <typename T> PrimitiveArrayGpu<T> min_sharded_op(const PrimitiveArrayGpu<T>& s, Communicator& c, ArrowComputePipeline& p) {
  return reduce_sharded_op(s, AGPU_RED_MIN, c, p);
}
template <typename T> PrimitiveArrayGpu<T> max_sharded_op(const PrimitiveArrayGpu<T>& s, Communicator& c, ArrowComputePipeline& p) {
  return reduce_sharded_op(s, AGPU_RED_MAX, c, p);
}
// sum / min / max / f64 sum of a sharded f32 column with ONE pass over this rank's shard (agpu_comm_reduce_stats_f32)
inline PrimitiveArrayGpu<float>::Stats stats_sharded_op(const PrimitiveArrayGpu<float>& shard, Communicator& comm, ArrowComputePipeline& p,
                                                        bool use_validity = false) {
  auto out = shard.gpu_device->create_empty_buffer(32);
  const void* validity = use_validity && shard.null_buffer ? shard.null_buffer->bit_buffer->ptr : nullptr;
  check(agpu_comm_reduce_stats_f32(comm.raw, p.h(), static_cast<const float*>(shard.data->ptr), validity, shard.len,
                                   static_cast<agpu_f32_stats*>(out->ptr)), "agpu_comm_reduce_stats_f32");
  p.keep.insert(p.keep.end(), {shard.data, out});
  return PrimitiveArrayGpu<float>::Stats{out, shard.gpu_device};
}
#define AGPU_SHARDED_DEFAULT(NAME)                                                                 \
  template <typename T> PrimitiveArrayGpu<T> NAME(const PrimitiveArrayGpu<T>& s, Communicator& c) { \
    ArrowComputePipeline p(s.gpu_device);                                                          \
    auto out = NAME##_op(s, c, p);                                                                 \
    p.finish();                                                                                    \
    return out;                                                                                    \
  }
AGPU_SHARDED_DEFAULT(sum_sharded)
AGPU_SHARDED_DEFAULT(min_sharded)
AGPU_SHARDED_DEFAULT(max_sharded)
#undef AGPU_SHARDED_DEFAULT

// ------------------------------------------------------------------ Arrow IPC files / streams (include/arrow_gpu.h "Arrow IPC")
// Not in the reference (arrays exist only as host Vecs [primitive_array_gpu.rs:22-104]).  The reader borrows the bytes
// (map_file(): the whole file memory-mapped, so a column goes page cache → HBM without a host copy in between).
struct IpcField {
  std::string name, format;
  int dtype;  // agpu_dtype or -1: no GPU array type (the column is skipped)
  bool nullable;
};
namespace detail {
inline ArrowArrayGPU array_of_column(const agpu_arrow_column& col, const DevicePtr& dev) {
  auto take = [&](void* ptr, uint64_t bytes) {
    auto b = std::make_shared<Buffer>();
    b->ptr = ptr;
    b->bytes = bytes;
    b->dev = dev;
    return b;
  };
  BufferPtr values = take(col.values, col.values_bytes);
  std::optional<NullBitBufferGpu> nulls;
  if (col.validity) nulls = NullBitBufferGpu{take(col.validity, col.validity_bytes), (size_t)col.length, dev};
  const size_t n = (size_t)col.length;
  switch (col.dtype) {
    case AGPU_F32: return Float32ArrayGPU(values, dev, n, nulls);
    case AGPU_U32: return UInt32ArrayGPU(values, dev, n, nulls);
    case AGPU_U16: return UInt16ArrayGPU(values, dev, n, nulls);
    case AGPU_U8: return UInt8ArrayGPU(values, dev, n, nulls);
    case AGPU_I32: return Int32ArrayGPU(values, dev, n, nulls);
    case AGPU_I16: return Int16ArrayGPU(values, dev, n, nulls);
    case AGPU_I8: return Int8ArrayGPU(values, dev, n, nulls);
    case AGPU_DATE32: return Date32ArrayGPU(values, dev, n, nulls);
    case AGPU_BOOL: return BooleanArrayGPU(values, dev, n, nulls);
  }
  throw ArrowErrorGPU(ArrowErrorGPU::CastingNotSupported, "unknown dtype in agpu_arrow_column");
}
inline agpu_arrow_column column_of_array(const ArrowArrayGPU& a) {
  return std::visit(
      [](const auto& x) {
        using A = std::decay_t<decltype(x)>;
        agpu_arrow_column col{};
        if constexpr (std::is_same_v<A, BooleanArrayGPU>) col.dtype = AGPU_BOOL;
        else col.dtype = A::DTYPE;
        col.length = x.len;
        col.null_count = x.null_buffer ? -1 : 0;
        col.values = x.data->ptr;
        col.values_bytes = x.data->bytes;
        if (x.null_buffer) {
          col.validity = x.null_buffer->bit_buffer->ptr;
          col.validity_bytes = x.null_buffer->bit_buffer->bytes;
        }
        return col;
      },
      a);
}
}  // namespace detail

class IpcReader {
 public:
  IpcReader(const void* data, size_t bytes) { open(data, bytes); }  // borrowed: keep `data` alive
  static std::unique_ptr<IpcReader> map_file(const std::string& path);
  IpcReader(const IpcReader&) = delete;
  ~IpcReader();
  const std::vector<IpcField>& fields() const { return fields_; }
  int64_t num_batches() const {
    int64_t n = 0;
    check(agpu_ipc_num_batches(raw_, &n), "agpu_ipc_num_batches");
    return n;
  }
  int64_t batch_rows(int64_t batch) const {
    int64_t n = 0;
    check(agpu_ipc_batch_rows(raw_, batch, &n), "agpu_ipc_batch_rows");
    return n;
  }
  int column_index(const std::string& name) const {
    for (size_t i = 0; i < fields_.size(); i++)
      if (fields_[i].name == name) return (int)i;
    throw ArrowErrorGPU(ArrowErrorGPU::OperationNotSupported, "no column named " + name);
  }
  ArrowArrayGPU read_column_op(int64_t batch, int column, ArrowComputePipeline& p) const {
    agpu_arrow_column col;
    check(agpu_ipc_read_column(raw_, batch, column, p.h(), &col), "agpu_ipc_read_column");
    return detail::array_of_column(col, p.device);
  }
  // several columns of one record batch, their device buffers out of ONE block placed for the HBM channel hash
  std::vector<ArrowArrayGPU> read_batch_op(int64_t batch, const std::vector<int32_t>& columns, ArrowComputePipeline& p) const {
    std::vector<agpu_arrow_column> cols(columns.size());
    check(agpu_ipc_read_batch(raw_, batch, columns.data(), (int32_t)columns.size(), p.h(), cols.data()), "agpu_ipc_read_batch");
    std::vector<ArrowArrayGPU> out;
    for (auto& c : cols) out.push_back(detail::array_of_column(c, p.device));
    return out;
  }
  ArrowArrayGPU read_column(int64_t batch, int column, const DevicePtr& dev) const {
    ArrowComputePipeline p(dev);
    auto out = read_column_op(batch, column, p);
    p.finish();
    p.sync();  // the upload has read the source; the caller may unmap it
    return out;
  }

 private:
  IpcReader() = default;
  void open(const void* data, size_t bytes) {
    check(agpu_ipc_open(data, bytes, &raw_), "agpu_ipc_open");
    int32_t n = 0;
    check(agpu_ipc_num_fields(raw_, &n), "agpu_ipc_num_fields");
    for (int32_t i = 0; i < n; i++) {
      agpu_ipc_field f;
      check(agpu_ipc_field_info(raw_, i, &f), "agpu_ipc_field_info");
      fields_.push_back(IpcField{f.name ? f.name : "", f.format ? f.format : "", f.dtype, f.nullable != 0});
    }
  }
  agpu_ipc_reader* raw_ = nullptr;
  std::vector<IpcField> fields_;
  void* map_ = nullptr;
  size_t map_bytes_ = 0;
};

class IpcWriter {
 public:
  // fd ≥ 0: bytes go to that descriptor as produced; fd < 0: in memory, finish() returns them
  IpcWriter(const std::vector<IpcField>& fields, bool file_format, int fd = -1) : n_fields_(fields.size()) {
    std::vector<agpu_ipc_field> f(fields.size());
    for (size_t i = 0; i < fields.size(); i++) f[i] = agpu_ipc_field{fields[i].name.c_str(), nullptr, fields[i].dtype, fields[i].nullable ? 1 : 0};
    check(agpu_ipc_writer_create(f.data(), (int32_t)f.size(), file_format ? 1 : 0, fd, &raw_), "agpu_ipc_writer_create");
  }
  IpcWriter(const IpcWriter&) = delete;
  ~IpcWriter() {
    if (raw_) agpu_ipc_writer_destroy(raw_);
  }
  void write_batch_op(const std::vector<ArrowArrayGPU>& columns, ArrowComputePipeline& p) {
    if (columns.size() != n_fields_) throw ArrowErrorGPU(ArrowErrorGPU::OperationNotSupported, "one column per schema field");
    std::vector<agpu_arrow_column> cols;
    for (auto& c : columns) cols.push_back(detail::column_of_array(c));
    check(agpu_device_sync(p.device->raw), "agpu_device_sync");  // other pipelines may still be writing the columns
    check(agpu_ipc_writer_write_device_batch(raw_, p.h(), cols.data()), "agpu_ipc_writer_write_device_batch");
  }
  void write_batch(const std::vector<ArrowArrayGPU>& columns) {
    if (columns.empty()) throw ArrowErrorGPU(ArrowErrorGPU::OperationNotSupported, "write_batch needs a column");
    ArrowComputePipeline p(get_gpu_device(columns[0]));
    write_batch_op(columns, p);
    p.finish();
  }
  std::vector<uint8_t> finish(uint64_t* total_bytes = nullptr) {
    const void* data = nullptr;
    uint64_t n = 0;
    check(agpu_ipc_writer_finish(raw_, &data, &n), "agpu_ipc_writer_finish");
    if (total_bytes) *total_bytes = n;
    std::vector<uint8_t> out;
    if (data) out.assign(static_cast<const uint8_t*>(data), static_cast<const uint8_t*>(data) + n);
    return out;
  }

 private:
  agpu_ipc_writer* raw_ = nullptr;
  size_t n_fields_;
};

}  // namespace arrow_gpu

// map_file needs POSIX mmap; kept at the end so the rest of the header stays free of system headers
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
namespace arrow_gpu {
inline std::unique_ptr<IpcReader> IpcReader::map_file(const std::string& path) {
  const int fd = ::open(path.c_str(), O_RDONLY);
  if (fd < 0) throw ArrowErrorGPU(ArrowErrorGPU::OperationNotSupported, "cannot open " + path);
  struct stat st;
  if (::fstat(fd, &st) != 0 || st.st_size <= 0) {
    ::close(fd);
    throw ArrowErrorGPU(ArrowErrorGPU::OperationNotSupported, "cannot stat / empty file: " + path);
  }
  void* m = ::mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
  ::close(fd);
  if (m == MAP_FAILED) throw ArrowErrorGPU(ArrowErrorGPU::OperationNotSupported, "mmap failed: " + path);
  std::unique_ptr<IpcReader> r(new IpcReader());
  r->map_ = m;
  r->map_bytes_ = (size_t)st.st_size;
  r->open(m, r->map_bytes_);  // on failure the destructor unmaps
  return r;
}
inline IpcReader::~IpcReader() {
  if (raw_) agpu_ipc_close(raw_);
  if (map_) ::munmap(map_, map_bytes_);
}
}  // namespace arrow_gpu
