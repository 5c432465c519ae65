// arrow_ipc.hip — Arrow IPC (streaming format and file format) at the C ABI (the "IPC import-export" half).
//
// The reference has no serialised form at all: arrays exist as host Vecs or wgpu buffers
// [ref: crates/array/src/array/primitive_array_gpu.rs:22-104].  Columns that live in files or arrive over a socket are
// Arrow IPC in practice (arrow-rs `arrow::ipc`, Arrow C++ / pyarrow `pa.ipc`), so this file reads and writes that
// format for the array types the reference has (i8/u8/i16/u16/i32/u32/f32/bool/date32 [ref: crates/array/src/array/
// mod.rs:40-50]) without any dependency: the metadata is Flatbuffers, and the two dozen lines of Flatbuffers one needs
// to read it (root offset → table → vtable → field) and to write it (front to back, offsets patched afterwards) are
// below.  Format facts follow the Arrow columnar specification (format/Message.fbs, Schema.fbs, File.fbs):
//   encapsulated message = 0xFFFFFFFF, int32 metadata size (padded so the body starts 8-byte aligned), Message
//   flatbuffer, body; a stream ends with 0xFFFFFFFF 0x00000000; a file is "ARROW1\0\0" + the same messages + a Footer
//   flatbuffer + int32 footer size + "ARROW1"; body buffers are listed flattened in depth-first field order as
//   {offset, length} relative to the body start.
// The reader BORROWS the caller's bytes (an mmap of the file is the intended use: buffers go from the page cache to
// HBM with no intermediate copy) and hands each column out as an ArrowArray view or imports it to the device through
// agpu_import_arrow.  Not supported, reported as AGPU_ERR_UNSUPPORTED: compressed bodies, big-endian files, and
// (per column) every type the GPU has no array for — such columns are skipped correctly, their neighbours stay readable.
#include <unistd.h>

#include <cerrno>
#include <cstring>
#include <memory>
#include <new>
#include <string>

#include "common.hpp"

#include "arrow_ipc_reader.inc"

// ================================================================ writer
struct agpu_ipc_writer {
  std::vector<FieldInfo> fields;
  bool file_format = false;
  int fd = -1;                 // ≥ 0: bytes go to this descriptor; else into `mem`
  std::vector<uint8_t> mem;
  uint64_t pos = 0;            // bytes emitted so far
  struct Block { int64_t offset; int32_t meta_len; int64_t body_len; };
  std::vector<Block> blocks;
  bool finished = false;
  bool failed = false;
  int codec = 0;               // 0: bodies as they are; 1: every buffer as an LZ4 frame (BodyCompression LZ4_FRAME / BUFFER)
  void* pin[2] = {nullptr, nullptr};  // page-locked bounce slots of the descriptor sink (device batches only)
  hipEvent_t pin_ev[2] = {nullptr, nullptr};
  int pin_device = -1;
};

namespace {

static agpu_status sink_write(agpu_ipc_writer* w, const void* src, size_t n) {
  if (!n) return AGPU_OK;
  if (w->fd >= 0) {
    const char* s = static_cast<const char*>(src);
    size_t left = n;
    while (left) {
      const ssize_t k = ::write(w->fd, s, left);
      if (k < 0) {
        if (errno == EINTR) continue;
        w->failed = true;
        agpu_set_error("agpu_ipc_writer: write() failed: %s", strerror(errno));
        return AGPU_ERR_ARG;
      }
      s += k;
      left -= (size_t)k;
    }
  } else {
    const uint8_t* s = static_cast<const uint8_t*>(src);
    w->mem.insert(w->mem.end(), s, s + n);
  }
  w->pos += n;
  return AGPU_OK;
}
static agpu_status sink_zeros(agpu_ipc_writer* w, size_t n) {
  static const uint8_t z[64] = {0};
  while (n) {
    const size_t k = n < 64 ? n : 64;
    agpu_status st = sink_write(w, z, k);
    if (st != AGPU_OK) return st;
    n -= k;
  }
  return AGPU_OK;
}

static void type_of_dtype(int32_t dt, uint8_t* tt, int32_t* bits, bool* sg) {
  *bits = 0;
  *sg = false;
  switch (dt) {
    case AGPU_I8: *tt = T_Int; *bits = 8; *sg = true; break;
    case AGPU_U8: *tt = T_Int; *bits = 8; break;
    case AGPU_I16: *tt = T_Int; *bits = 16; *sg = true; break;
    case AGPU_U16: *tt = T_Int; *bits = 16; break;
    case AGPU_I32: *tt = T_Int; *bits = 32; *sg = true; break;
    case AGPU_U32: *tt = T_Int; *bits = 32; break;
    case AGPU_F32: *tt = T_FloatingPoint; break;
    case AGPU_BOOL: *tt = T_Bool; break;
    case AGPU_DATE32: *tt = T_Date; break;
    default: *tt = T_NONE; break;
  }
}

constexpr size_t kPinChunk = (size_t)8 << 20;
static agpu_status device_to_fd(agpu_ipc_writer* w, agpu_pipeline* p, const char* dev_ptr, size_t bytes) {
  AGPU_BIND_AS(p, "agpu_ipc_writer_write_device_batch");
  AGPU_REQUIRE(!p->capturing, AGPU_ERR_ARG, "not during graph capture");
  if (!w->pin[0]) {
    for (int k = 0; k < 2; k++) {
      AGPU_HIP(hipHostMalloc(&w->pin[k], kPinChunk, hipHostMallocDefault));
      AGPU_HIP(hipEventCreateWithFlags(&w->pin_ev[k], hipEventDisableTiming));
    }
    w->pin_device = p->dev->ordinal;
  }
  const size_t nchunks = (bytes + kPinChunk - 1) / kPinChunk;
  agpu_status st = AGPU_OK;
  for (size_t k = 0; k <= nchunks && st == AGPU_OK; k++) {
    if (k < nchunks) {
      const size_t off = k * kPinChunk, len = bytes - off < kPinChunk ? bytes - off : kPinChunk;
      AGPU_HIP(hipMemcpyAsync(w->pin[k & 1], dev_ptr + off, len, hipMemcpyDeviceToHost, p->stream));
      AGPU_HIP(hipEventRecord(w->pin_ev[k & 1], p->stream));
    }
    if (k >= 1) {
      const size_t off = (k - 1) * kPinChunk, len = bytes - off < kPinChunk ? bytes - off : kPinChunk;
      AGPU_HIP(hipEventSynchronize(w->pin_ev[(k - 1) & 1]));
      st = sink_write(w, w->pin[(k - 1) & 1], len);
    }
  }
  if (st != AGPU_OK) (void)hipStreamSynchronize(p->stream);  // a DMA may still target the other slot
  return st;
}

// Schema table written at the current end of `o`; returns its position
static size_t write_schema_table(FbOut& o, const std::vector<FieldInfo>& fields) {
  std::vector<FbField> sf = {{0, 2, 0 /* Little */, false, 0}, {1, 4, 0, true, 0}};
  const size_t schema = fb_write_table(o, sf);
  o.pad_to(4);
  o.link(sf[1].at);
  o.u32((uint32_t)fields.size());
  const size_t refs = o.b.size();
  o.b.resize(refs + 4 * fields.size(), 0);
  for (size_t i = 0; i < fields.size(); i++) {
    const FieldInfo& fi = fields[i];
    uint8_t tt;
    int32_t bits;
    bool sg;
    type_of_dtype(fi.dtype, &tt, &bits, &sg);
    std::vector<FbField> ff = {{0, 4, 0, true, 0}, {1, 1, fi.nullable ? 1u : 0u, false, 0}, {2, 1, tt, false, 0},
                               {3, 4, 0, true, 0}, {5, 4, 0, true, 0}};
    const size_t ftab = fb_write_table(o, ff);
    o.patch32(refs + 4 * i, (uint32_t)(ftab - (refs + 4 * i)));
    fb_write_string(o, ff[0].at, fi.name);
    // the type table
    std::vector<FbField> tf;
    if (tt == T_Int) tf = {{0, 4, (uint64_t)(uint32_t)bits, false, 0}, {1, 1, sg ? 1u : 0u, false, 0}};
    else if (tt == T_FloatingPoint) tf = {{0, 2, 1 /* SINGLE */, false, 0}};
    else if (tt == T_Date) tf = {{0, 2, 0 /* DAY */, false, 0}};
    // link after fb_write_table has padded: the reference must point at the TABLE, not at its vtable
    const size_t ttab = fb_write_table(o, tf);
    o.patch32(ff[3].at, (uint32_t)(ttab - ff[3].at));
    // children: an empty vector (Arrow C++ rejects a null children pointer)
    o.pad_to(4);
    o.link(ff[4].at);
    o.u32(0);
  }
  return schema;
}

static agpu_status emit_message(agpu_ipc_writer* w, FbOut& o, int32_t* meta_len_out) {
  // prefix (8 bytes) + flatbuffer + padding: the body that follows must start on an 8-byte boundary
  while ((8 + o.b.size()) % 8) o.b.push_back(0);
  const uint32_t cont = kContinuation, size = (uint32_t)o.b.size();
  agpu_status st = sink_write(w, &cont, 4);
  if (st == AGPU_OK) st = sink_write(w, &size, 4);
  if (st == AGPU_OK) st = sink_write(w, o.b.data(), o.b.size());
  if (meta_len_out) *meta_len_out = (int32_t)(8 + o.b.size());
  return st;
}

static agpu_status write_schema_message(agpu_ipc_writer* w) {
  FbOut o;
  o.u32(0);  // root uoffset, linked below
  std::vector<FbField> mf = {{0, 2, (uint64_t)kV5, false, 0}, {1, 1, H_Schema, false, 0}, {2, 4, 0, true, 0}, {3, 8, 0, false, 0}};
  const size_t msg = fb_write_table(o, mf);
  o.patch32(0, (uint32_t)msg);
  const size_t schema = write_schema_table(o, w->fields);
  o.patch32(mf[2].at, (uint32_t)(schema - mf[2].at));
  return emit_message(w, o, nullptr);
}

struct HostColumn {
  const uint8_t* validity;  // Arrow bitmap or NULL
  const uint8_t* values;
  uint64_t offset;          // in elements / bits (C Data Interface `offset`)
  int64_t null_count;
};

// bits [off, off+n) of src → dst from bit 0, trailing bits of the last byte 0
static void copy_bits_host(const uint8_t* src, uint64_t off, uint64_t n, uint8_t* dst) {
  const size_t nb = bitmap_span_bytes(n);
  if (!nb) return;
  const unsigned sh = (unsigned)(off & 7);
  const uint8_t* s = src + off / 8;
  if (sh == 0) memcpy(dst, s, nb);
  else {
    const size_t src_bytes = bitmap_span_bytes(sh + n);
    for (size_t i = 0; i < nb; i++) {
      const unsigned lo = s[i] >> sh;
      const unsigned hi = i + 1 < src_bytes ? (unsigned)s[i + 1] << (8 - sh) : 0u;
      dst[i] = (uint8_t)(lo | hi);
    }
  }
  if (n & 7) dst[nb - 1] &= (uint8_t)((1u << (n & 7)) - 1u);
}
static int64_t count_zero_bits(const uint8_t* bits, uint64_t n) {
  int64_t ones = 0;
  const size_t full = (size_t)(n / 8);
  for (size_t i = 0; i < full; i++) ones += __builtin_popcount(bits[i]);
  if (n & 7) ones += __builtin_popcount(bits[full] & ((1u << (n & 7)) - 1u));
  return (int64_t)n - ones;
}

static size_t pad64(size_t n) { return (n + 63) / 64 * 64; }

// RecordBatch metadata for `rows` rows with the given (validity_bytes, values_bytes, null_count) per column
static void build_batch_message(FbOut& o, int64_t rows, const std::vector<int64_t>& null_counts,
                                const std::vector<size_t>& vbytes, const std::vector<size_t>& dbytes, int64_t* body_len,
                                bool lz4 = false) {
  o.u32(0);
  std::vector<FbField> mf = {{0, 2, (uint64_t)kV5, false, 0}, {1, 1, H_RecordBatch, false, 0}, {2, 4, 0, true, 0}, {3, 8, 0, false, 0}};
  const size_t msg = fb_write_table(o, mf);
  o.patch32(0, (uint32_t)msg);
  std::vector<FbField> rf = {{0, 8, (uint64_t)rows, false, 0}, {1, 4, 0, true, 0}, {2, 4, 0, true, 0}};
  if (lz4) rf.push_back({3, 4, 0, true, 0});  // compression: BodyCompression
  const size_t rb = fb_write_table(o, rf);
  o.patch32(mf[2].at, (uint32_t)(rb - mf[2].at));
  if (lz4) {  // {codec: LZ4_FRAME = 0, method: BUFFER = 0}, written out although both are the defaults
    std::vector<FbField> cf = {{0, 1, 0, false, 0}, {1, 1, 0, false, 0}};
    const size_t ct = fb_write_table(o, cf);
    o.patch32(rf[3].at, (uint32_t)(ct - rf[3].at));
  }
  const size_t ncol = null_counts.size();
  // nodes: vector of 16-byte structs, elements 8-byte aligned (so the length word sits at 4 mod 8)
  o.pad_to(8, 4);
  o.link(rf[1].at);
  o.u32((uint32_t)ncol);
  for (size_t c = 0; c < ncol; c++) {
    o.i64(rows);
    o.i64(null_counts[c]);
  }
  o.pad_to(8, 4);
  o.link(rf[2].at);
  o.u32((uint32_t)(2 * ncol));
  int64_t off = 0;
  for (size_t c = 0; c < ncol; c++) {
    o.i64(off);
    o.i64((int64_t)vbytes[c]);
    off += (int64_t)pad64(vbytes[c]);
    o.i64(off);
    o.i64((int64_t)dbytes[c]);
    off += (int64_t)pad64(dbytes[c]);
  }
  *body_len = off;
  memcpy(o.b.data() + mf[3].at, &off, 8);
}

// One record batch whose buffers are all in host memory, every buffer as {int64 uncompressed length, LZ4 frame} — or
// {−1, the bytes} where the frame would not be smaller (the format's escape for incompressible buffers).
struct HostPiece {
  const uint8_t* p;
  size_t n;
};
static agpu_status emit_batch_lz4(agpu_ipc_writer* w, int64_t rows, const std::vector<int64_t>& nulls, const std::vector<HostPiece>& validity,
                                 const std::vector<HostPiece>& values) {
  const size_t ncol = nulls.size();
  std::vector<std::vector<uint8_t>> enc(2 * ncol);
  std::vector<size_t> vbytes(ncol), dbytes(ncol);
  for (size_t c = 0; c < ncol; c++)
    for (int k = 0; k < 2; k++) {
      const HostPiece& src = k == 0 ? validity[c] : values[c];
      std::vector<uint8_t>& out = enc[2 * c + (size_t)k];
      if (src.n) {
        int64_t ulen = (int64_t)src.n;
        out.resize(8);
        lz4_frame_encode(src.p, src.n, &out);
        if (out.size() - 8 >= src.n) {
          ulen = -1;
          out.resize(8);
          out.insert(out.end(), src.p, src.p + src.n);
        }
        memcpy(out.data(), &ulen, 8);
      }
      (k == 0 ? vbytes[c] : dbytes[c]) = out.size();
    }
  FbOut o;
  int64_t body_len = 0;
  build_batch_message(o, rows, nulls, vbytes, dbytes, &body_len, true);
  const int64_t block_off = (int64_t)w->pos;
  int32_t meta_len = 0;
  agpu_status st = emit_message(w, o, &meta_len);
  for (size_t k = 0; k < enc.size() && st == AGPU_OK; k++) {
    st = sink_write(w, enc[k].data(), enc[k].size());
    if (st == AGPU_OK) st = sink_zeros(w, pad64(enc[k].size()) - enc[k].size());
  }
  if (st != AGPU_OK) {
    w->failed = true;
    return st;
  }
  w->blocks.push_back({block_off, meta_len, body_len});
  return AGPU_OK;
}

}  // namespace

extern "C" {

// A dictionary-encoded column is decoded ON THE GPU: only the indices (1, 2 or 4 bytes per row) and the dictionary cross
// the link; indices widen to u32 (agpu_cast), null slots — whose indices are unspecified — become index 0 (agpu_merge
// against a zero column), and agpu_take gathers from the dictionary (small: it lives in L2).  An index beyond the
// dictionary in a VALID slot surfaces like any bad take index: AGPU_ERR_SHAPE at the pipeline's next sync.
static void release_noop_array(struct ArrowArray*) {}
static void release_noop_schema(struct ArrowSchema*) {}
static agpu_status read_dict_column_device(const agpu_ipc_reader* r, int64_t batch, int32_t column, agpu_pipeline* p,
                                           agpu_arrow_column* out_column) {
  const FieldInfo& fi = r->fields[(size_t)column];
  const BatchInfo& bi = r->batches[(size_t)batch];
  std::vector<uint8_t> idx_owned[2], dict_values;
  const uint8_t *iv = nullptr, *ix = nullptr;
  int64_t rows = 0, nulls = 0;
  agpu_status st = dict_parts(r, bi, fi, (int)column, &rows, &nulls, &iv, &ix, idx_owned, &dict_values);
  if (st != AGPU_OK) return st;
  const uint64_t n = (uint64_t)rows;
  const size_t w = agpu_dtype_size((agpu_dtype)fi.dtype), iw = agpu_dtype_size((agpu_dtype)fi.index_dtype);
  const uint64_t dict_len = dict_values.size() / w;
  agpu_device* dev = p->dev;
  // 1. the indices as an ordinary primitive column (validity re-aligned and masked by the import)
  static const char* const fmt_of[] = {nullptr, nullptr, "I", "S", "C", "i", "s", "c", nullptr};  // by agpu_dtype: U32 U16 U8 I32 I16 I8
  const void* bufs[2] = {nulls > 0 ? iv : nullptr, ix};
  struct ArrowArray a;
  struct ArrowSchema sc;
  memset(&a, 0, sizeof(a));
  memset(&sc, 0, sizeof(sc));
  a.length = rows;
  a.null_count = nulls;
  a.n_buffers = 2;
  a.buffers = bufs;
  a.release = release_noop_array;
  sc.format = fmt_of[fi.index_dtype];
  sc.name = "";
  sc.release = release_noop_schema;
  agpu_arrow_column idx;
  st = agpu_import_arrow(p, &a, &sc, &idx);
  if (st != AGPU_OK) return st;
  void *idx32 = nullptr, *zeros = nullptr, *merged = nullptr, *dict_dev = nullptr, *vals = nullptr;
  const size_t n4 = n * 4 ? n * 4 : 16, nvb = n * w ? n * w : 16;
  const uint32_t* take_idx = static_cast<const uint32_t*>(idx.values);
  if (st == AGPU_OK && iw < 4) {  // 2. widen to u32
    st = agpu_malloc(dev, n4, 0, &idx32);
    if (st == AGPU_OK && n) st = agpu_cast(p, (agpu_dtype)fi.index_dtype, AGPU_U32, idx.values, idx32, n);
    take_idx = static_cast<const uint32_t*>(idx32);
  }
  if (st == AGPU_OK && idx.validity && n) {  // 3. null slots → index 0
    st = agpu_malloc(dev, n4, 0, &zeros);
    if (st == AGPU_OK) st = agpu_memset(p, zeros, 0, n4);
    if (st == AGPU_OK) st = agpu_malloc(dev, n4, 0, &merged);
    if (st == AGPU_OK) st = agpu_merge(p, 4, take_idx, zeros, idx.validity, merged, n);
    take_idx = static_cast<const uint32_t*>(merged);
  }
  if (st == AGPU_OK) st = agpu_malloc(dev, nvb, 0, &vals);
  if (st == AGPU_OK && n) {
    if (dict_len == 0) {  // an all-null column over an empty dictionary
      st = agpu_memset(p, vals, 0, nvb);
    } else {  // 4. the dictionary and the gather
      st = agpu_malloc(dev, dict_values.size(), 0, &dict_dev);
      if (st == AGPU_OK) st = agpu_upload(p, dict_dev, dict_values.data(), dict_values.size());
      if (st == AGPU_OK) st = agpu_take(p, (int32_t)w, dict_dev, dict_len, take_idx, vals, n);
    }
  }
  // temporaries: the pool hands them out again only after the stream has passed the kernels above
  if (dict_dev) (void)agpu_free(dev, dict_dev);
  if (merged) (void)agpu_free(dev, merged);
  if (zeros) (void)agpu_free(dev, zeros);
  if (idx32) (void)agpu_free(dev, idx32);
  (void)agpu_free(dev, idx.values);
  if (st != AGPU_OK) {
    if (vals) (void)agpu_free(dev, vals);
    if (idx.validity) (void)agpu_free(dev, idx.validity);
    return st;
  }
  memset(out_column, 0, sizeof(*out_column));
  out_column->dtype = (agpu_dtype)fi.dtype;
  out_column->length = n;
  out_column->null_count = idx.validity ? nulls : 0;
  out_column->values = vals;
  out_column->values_bytes = nvb;
  out_column->validity = idx.validity;
  out_column->validity_bytes = idx.validity_bytes;
  return AGPU_OK;
}

agpu_status agpu_ipc_read_column(const agpu_ipc_reader* r, int64_t batch, int32_t column, agpu_pipeline* p,
                                 agpu_arrow_column* out_column) {
  AGPU_REQUIRE(p && out_column, AGPU_ERR_ARG, "null argument");
  if (r && batch >= 0 && (size_t)batch < r->batches.size() && column >= 0 && (size_t)column < r->fields.size() &&
      r->fields[(size_t)column].dict && r->fields[(size_t)column].first_node >= 0) {
    try {
      return read_dict_column_device(r, batch, column, p, out_column);
    } catch (const std::bad_alloc&) {
      agpu_set_error("agpu_ipc_read_column: out of host memory");
      return AGPU_ERR_ARG;
    }
  }
  struct ArrowArray a;
  struct ArrowSchema s;
  agpu_status st = agpu_ipc_column_view(r, batch, column, &a, &s);
  if (st != AGPU_OK) return st;
  st = agpu_import_arrow(p, &a, &s, out_column);
  a.release(&a);
  s.release(&s);
  return st;
}

static agpu_status agpu_ipc_read_batch_impl(const agpu_ipc_reader* r, int64_t batch, const int32_t* columns, int32_t n_columns,
                                agpu_pipeline* p, agpu_arrow_column* out_columns) {
  AGPU_REQUIRE(p && columns && out_columns && n_columns > 0, AGPU_ERR_ARG, "bad argument");
  std::vector<struct ArrowArray> arrays((size_t)n_columns);
  std::vector<struct ArrowSchema> schemas((size_t)n_columns);
  std::vector<const struct ArrowArray*> ap((size_t)n_columns);
  std::vector<const struct ArrowSchema*> sp((size_t)n_columns);
  agpu_status st = AGPU_OK;
  int32_t made = 0;
  for (; made < n_columns && st == AGPU_OK; made++) {
    st = agpu_ipc_column_view(r, batch, columns[made], &arrays[(size_t)made], &schemas[(size_t)made]);
    if (st != AGPU_OK) break;
    ap[(size_t)made] = &arrays[(size_t)made];
    sp[(size_t)made] = &schemas[(size_t)made];
  }
  if (st == AGPU_OK) st = agpu_import_arrow_table(p, n_columns, ap.data(), sp.data(), out_columns);
  for (int32_t k = 0; k < made; k++) {
    arrays[(size_t)k].release(&arrays[(size_t)k]);
    schemas[(size_t)k].release(&schemas[(size_t)k]);
  }
  return st;
}

// ---------------------------------------------------------------- writer
static agpu_status agpu_ipc_writer_create_impl(const agpu_ipc_field* fields, int32_t n_fields, int32_t file_format, int32_t fd,
                                   agpu_ipc_writer** out_writer) {
  AGPU_REQUIRE(out_writer && (fields || n_fields == 0) && n_fields >= 0, AGPU_ERR_ARG, "bad argument");
  *out_writer = nullptr;
  std::unique_ptr<agpu_ipc_writer> w(new agpu_ipc_writer);
  for (int32_t i = 0; i < n_fields; i++) {
    FieldInfo fi;
    fi.name = fields[i].name ? fields[i].name : "";
    fi.dtype = fields[i].dtype;
    fi.nullable = fields[i].nullable != 0;
    uint8_t tt;
    int32_t bits;
    bool sg;
    type_of_dtype(fi.dtype, &tt, &bits, &sg);
    if (tt == T_NONE) {
      agpu_set_error("agpu_ipc_writer_create: field %d has no agpu_dtype", (int)i);
      return AGPU_ERR_UNSUPPORTED;
    }
    w->fields.push_back(std::move(fi));
  }
  w->file_format = file_format != 0;
  w->fd = fd;
  agpu_status st = AGPU_OK;
  if (w->file_format) st = sink_write(w.get(), "ARROW1\0\0", 8);
  if (st == AGPU_OK) st = write_schema_message(w.get());
  if (st != AGPU_OK) return st;
  *out_writer = w.release();
  return AGPU_OK;
}

static agpu_status writer_ready(agpu_ipc_writer* w) {
  AGPU_REQUIRE(w, AGPU_ERR_ARG, "null writer");
  AGPU_REQUIRE(!w->finished, AGPU_ERR_ARG, "writer already finished");
  AGPU_REQUIRE(!w->failed, AGPU_ERR_ARG, "writer failed earlier");
  return AGPU_OK;
}

static agpu_status agpu_ipc_writer_write_batch_impl(agpu_ipc_writer* w, const struct ArrowArray* const* columns) {
  agpu_status st = writer_ready(w);
  if (st != AGPU_OK) return st;
  const size_t ncol = w->fields.size();
  AGPU_REQUIRE(columns || ncol == 0, AGPU_ERR_ARG, "null columns");
  int64_t rows = ncol ? -1 : 0;
  std::vector<HostColumn> hc(ncol);
  std::vector<int64_t> nulls(ncol);
  std::vector<size_t> vbytes(ncol), dbytes(ncol);
  for (size_t c = 0; c < ncol; c++) {
    const struct ArrowArray* a = columns[c];
    AGPU_REQUIRE(a && a->release, AGPU_ERR_ARG, "null / released ArrowArray");
    AGPU_REQUIRE(a->n_buffers == 2 && a->buffers && a->n_children == 0 && !a->dictionary, AGPU_ERR_SHAPE, "a primitive array has exactly 2 buffers");
    AGPU_REQUIRE(a->length >= 0 && a->offset >= 0, AGPU_ERR_SHAPE, "negative length / offset");
    if (rows < 0) rows = a->length;
    AGPU_REQUIRE(a->length == rows, AGPU_ERR_SHAPE, "columns of one record batch differ in length");
    hc[c] = HostColumn{static_cast<const uint8_t*>(a->buffers[0]), static_cast<const uint8_t*>(a->buffers[1]), (uint64_t)a->offset, a->null_count};
    AGPU_REQUIRE(rows == 0 || hc[c].values, AGPU_ERR_SHAPE, "null data buffer");
    const bool has_v = hc[c].validity && a->null_count != 0 && rows > 0;
    if (!has_v) hc[c].validity = nullptr;
    vbytes[c] = has_v ? bitmap_span_bytes((uint64_t)rows) : 0;
    const int32_t dt = w->fields[c].dtype;
    dbytes[c] = dt == AGPU_BOOL ? bitmap_span_bytes((uint64_t)rows) : (size_t)rows * agpu_dtype_size((agpu_dtype)dt);
  }
  // validity bitmaps are re-packed to offset 0 first so that exact null counts can go into the metadata
  std::vector<std::vector<uint8_t>> vpacked(ncol);
  for (size_t c = 0; c < ncol; c++) {
    nulls[c] = 0;
    if (hc[c].validity) {
      vpacked[c].resize(vbytes[c]);
      copy_bits_host(hc[c].validity, hc[c].offset, (uint64_t)rows, vpacked[c].data());
      nulls[c] = count_zero_bits(vpacked[c].data(), (uint64_t)rows);
      if (nulls[c] == 0) {
        vbytes[c] = 0;
        vpacked[c].clear();
        hc[c].validity = nullptr;
      }
    }
  }
  if (w->codec == 1) {
    std::vector<HostPiece> pv(ncol), pd(ncol);
    std::vector<std::vector<uint8_t>> bpacked(ncol);
    for (size_t c = 0; c < ncol; c++) {
      pv[c] = HostPiece{vpacked[c].data(), vbytes[c]};
      const int32_t dt = w->fields[c].dtype;
      if (dt == AGPU_BOOL) {
        bpacked[c].resize(dbytes[c]);
        copy_bits_host(hc[c].values, hc[c].offset, (uint64_t)rows, bpacked[c].data());
        pd[c] = HostPiece{bpacked[c].data(), dbytes[c]};
      } else {
        pd[c] = HostPiece{dbytes[c] ? hc[c].values + hc[c].offset * agpu_dtype_size((agpu_dtype)dt) : nullptr, dbytes[c]};
      }
    }
    return emit_batch_lz4(w, rows, nulls, pv, pd);
  }
  FbOut o;
  int64_t body_len = 0;
  build_batch_message(o, rows, nulls, vbytes, dbytes, &body_len);
  const int64_t block_off = (int64_t)w->pos;
  int32_t meta_len = 0;
  st = emit_message(w, o, &meta_len);
  for (size_t c = 0; c < ncol && st == AGPU_OK; c++) {
    if (vbytes[c]) {
      st = sink_write(w, vpacked[c].data(), vbytes[c]);
      if (st == AGPU_OK) st = sink_zeros(w, pad64(vbytes[c]) - vbytes[c]);
    }
    if (st != AGPU_OK) break;
    const int32_t dt = w->fields[c].dtype;
    if (dt == AGPU_BOOL) {
      std::vector<uint8_t> packed(dbytes[c]);
      copy_bits_host(hc[c].values, hc[c].offset, (uint64_t)rows, packed.data());
      st = sink_write(w, packed.data(), dbytes[c]);
    } else {
      st = sink_write(w, hc[c].values + hc[c].offset * agpu_dtype_size((agpu_dtype)dt), dbytes[c]);
    }
    if (st == AGPU_OK) st = sink_zeros(w, pad64(dbytes[c]) - dbytes[c]);
  }
  if (st != AGPU_OK) {
    w->failed = true;
    return st;
  }
  w->blocks.push_back({block_off, meta_len, body_len});
  return AGPU_OK;
}

static agpu_status agpu_ipc_writer_write_device_batch_impl(agpu_ipc_writer* w, agpu_pipeline* p, const agpu_arrow_column* columns) {
  agpu_status st = writer_ready(w);
  if (st != AGPU_OK) return st;
  AGPU_REQUIRE(p, AGPU_ERR_ARG, "null pipeline");
  const size_t ncol = w->fields.size();
  AGPU_REQUIRE(columns || ncol == 0, AGPU_ERR_ARG, "null columns");
  const int64_t rows = ncol ? (int64_t)columns[0].length : 0;
  std::vector<int64_t> nulls(ncol);
  std::vector<size_t> vbytes(ncol), dbytes(ncol);
  void* cnt_dev = nullptr;
  for (size_t c = 0; c < ncol; c++) {
    const agpu_arrow_column& col = columns[c];
    AGPU_REQUIRE((int64_t)col.length == rows, AGPU_ERR_SHAPE, "columns of one record batch differ in length");
    AGPU_REQUIRE((int32_t)col.dtype == w->fields[c].dtype, AGPU_ERR_SHAPE, "column dtype differs from the writer's schema");
    AGPU_REQUIRE(rows == 0 || col.values, AGPU_ERR_ARG, "null values");
    nulls[c] = 0;
    vbytes[c] = 0;
    if (col.validity && rows) {
      nulls[c] = col.null_count;
      if (nulls[c] < 0) {  // not known: count on the device (one popcount launch + an 8-byte readback)
        if (!cnt_dev) {
          st = agpu_malloc(p->dev, 16, 0, &cnt_dev);
          if (st != AGPU_OK) return st;
        }
        uint64_t ones = 0;
        st = agpu_bitmap_popcount(p, col.validity, (uint64_t)rows, static_cast<uint64_t*>(cnt_dev));
        if (st == AGPU_OK) st = agpu_download(p, &ones, cnt_dev, 8);
        if (st != AGPU_OK) {
          (void)agpu_free(p->dev, cnt_dev);
          return st;
        }
        nulls[c] = rows - (int64_t)ones;
      }
      if (nulls[c] > 0) vbytes[c] = bitmap_span_bytes((uint64_t)rows);
    }
    dbytes[c] = col.dtype == AGPU_BOOL ? bitmap_span_bytes((uint64_t)rows) : (size_t)rows * agpu_dtype_size(col.dtype);
  }
  if (cnt_dev) (void)agpu_free(p->dev, cnt_dev);
  if (w->codec == 1) {  // compression runs on the host: HBM → host buffers → LZ4 frames → sink
    std::vector<std::vector<uint8_t>> hv(ncol), hd(ncol);
    std::vector<HostPiece> pv(ncol), pd(ncol);
    for (size_t c = 0; c < ncol; c++) {
      hv[c].resize(vbytes[c]);
      hd[c].resize(dbytes[c]);
      if (vbytes[c]) st = agpu_staged_copy(p, const_cast<void*>(columns[c].validity), hv[c].data(), vbytes[c], 0);
      if (st == AGPU_OK && dbytes[c]) st = agpu_staged_copy(p, const_cast<void*>(columns[c].values), hd[c].data(), dbytes[c], 0);
      if (st != AGPU_OK) return st;
      if (vbytes[c] && (rows & 7)) hv[c][vbytes[c] - 1] &= (uint8_t)((1u << (rows & 7)) - 1u);
      pv[c] = HostPiece{hv[c].data(), vbytes[c]};
      pd[c] = HostPiece{hd[c].data(), dbytes[c]};
    }
    return emit_batch_lz4(w, rows, nulls, pv, pd);
  }
  FbOut o;
  int64_t body_len = 0;
  build_batch_message(o, rows, nulls, vbytes, dbytes, &body_len);
  const int64_t block_off = (int64_t)w->pos;
  int32_t meta_len = 0;
  st = emit_message(w, o, &meta_len);
  if (st != AGPU_OK) {
    w->failed = true;
    return st;
  }
  // body: HBM → the sink.  In-memory sink: straight into the output buffer; descriptor sink: two page-locked 8 MiB
  // slots — the DMA of chunk k runs while write() drains chunk k−1.
  auto put = [&](const void* dev_ptr, size_t bytes) -> agpu_status {
    if (!bytes) return AGPU_OK;
    const size_t padded = pad64(bytes);
    if (w->fd < 0) {
      const size_t at = w->mem.size();
      w->mem.resize(at + padded, 0);
      agpu_status s2 = agpu_staged_copy(p, const_cast<void*>(dev_ptr), w->mem.data() + at, bytes, 0);
      if (s2 == AGPU_OK) w->pos += padded;
      return s2;
    }
    agpu_status s2 = device_to_fd(w, p, static_cast<const char*>(dev_ptr), bytes);
    if (s2 == AGPU_OK) s2 = sink_zeros(w, padded - bytes);
    return s2;
  };
  for (size_t c = 0; c < ncol && st == AGPU_OK; c++) {
    if (vbytes[c]) st = put(columns[c].validity, vbytes[c]);
    if (st == AGPU_OK) st = put(columns[c].values, dbytes[c]);
  }
  if (st != AGPU_OK) {
    w->failed = true;
    return st;
  }
  w->blocks.push_back({block_off, meta_len, body_len});
  return AGPU_OK;
}

static agpu_status agpu_ipc_writer_finish_impl(agpu_ipc_writer* w, const void** out_data, uint64_t* out_bytes) {
  agpu_status st = writer_ready(w);
  if (st != AGPU_OK) return st;
  const uint32_t eos[2] = {kContinuation, 0};
  st = sink_write(w, eos, 8);
  if (st == AGPU_OK && w->file_format) {
    FbOut o;
    o.u32(0);
    std::vector<FbField> ff = {{0, 2, (uint64_t)kV5, false, 0}, {1, 4, 0, true, 0}, {2, 4, 0, true, 0}, {3, 4, 0, true, 0}};
    const size_t footer = fb_write_table(o, ff);
    o.patch32(0, (uint32_t)footer);
    const size_t schema = write_schema_table(o, w->fields);
    o.patch32(ff[1].at, (uint32_t)(schema - ff[1].at));
    o.pad_to(8, 4);  // dictionaries: empty vector of Block
    o.link(ff[2].at);
    o.u32(0);
    o.pad_to(8, 4);
    o.link(ff[3].at);
    o.u32((uint32_t)w->blocks.size());
    for (auto& b : w->blocks) {
      o.i64(b.offset);
      o.u32((uint32_t)b.meta_len);
      o.u32(0);
      o.i64(b.body_len);
    }
    const int32_t fsize = (int32_t)o.b.size();
    st = sink_write(w, o.b.data(), o.b.size());
    if (st == AGPU_OK) st = sink_write(w, &fsize, 4);
    if (st == AGPU_OK) st = sink_write(w, "ARROW1", 6);
  }
  if (st != AGPU_OK) {
    w->failed = true;
    return st;
  }
  w->finished = true;
  if (out_data) *out_data = w->fd < 0 ? w->mem.data() : nullptr;
  if (out_bytes) *out_bytes = w->pos;
  return AGPU_OK;
}

agpu_status agpu_ipc_read_batch(const agpu_ipc_reader* r, int64_t batch, const int32_t* columns, int32_t n_columns,
                                agpu_pipeline* p, agpu_arrow_column* out_columns) {
  try {  // these allocate (metadata, in-memory sink): an allocation failure must not unwind through the C ABI
    return agpu_ipc_read_batch_impl(r, batch, columns, n_columns, p, out_columns);
  } catch (const std::bad_alloc&) {
    agpu_set_error("agpu_ipc_read_batch: out of host memory");
    return AGPU_ERR_ARG;
  }
}

agpu_status agpu_ipc_writer_create(const agpu_ipc_field* fields, int32_t n_fields, int32_t file_format, int32_t fd,
                                   agpu_ipc_writer** out_writer) {
  try {  // these allocate (metadata, in-memory sink): an allocation failure must not unwind through the C ABI
    return agpu_ipc_writer_create_impl(fields, n_fields, file_format, fd, out_writer);
  } catch (const std::bad_alloc&) {
    agpu_set_error("agpu_ipc_writer_create: out of host memory");
    return AGPU_ERR_ARG;
  }
}

agpu_status agpu_ipc_writer_write_batch(agpu_ipc_writer* w, const struct ArrowArray* const* columns) {
  try {  // these allocate (metadata, in-memory sink): an allocation failure must not unwind through the C ABI
    return agpu_ipc_writer_write_batch_impl(w, columns);
  } catch (const std::bad_alloc&) {
    agpu_set_error("agpu_ipc_writer_write_batch: out of host memory");
    return AGPU_ERR_ARG;
  }
}

agpu_status agpu_ipc_writer_write_device_batch(agpu_ipc_writer* w, agpu_pipeline* p, const agpu_arrow_column* columns) {
  try {  // these allocate (metadata, in-memory sink): an allocation failure must not unwind through the C ABI
    return agpu_ipc_writer_write_device_batch_impl(w, p, columns);
  } catch (const std::bad_alloc&) {
    agpu_set_error("agpu_ipc_writer_write_device_batch: out of host memory");
    return AGPU_ERR_ARG;
  }
}

agpu_status agpu_ipc_writer_finish(agpu_ipc_writer* w, const void** out_data, uint64_t* out_bytes) {
  try {  // these allocate (metadata, in-memory sink): an allocation failure must not unwind through the C ABI
    return agpu_ipc_writer_finish_impl(w, out_data, out_bytes);
  } catch (const std::bad_alloc&) {
    agpu_set_error("agpu_ipc_writer_finish: out of host memory");
    return AGPU_ERR_ARG;
  }
}

agpu_status agpu_ipc_writer_set_compression(agpu_ipc_writer* w, int32_t codec) {
  AGPU_REQUIRE(w, AGPU_ERR_ARG, "null writer");
  AGPU_REQUIRE(w->blocks.empty() && !w->finished, AGPU_ERR_ARG, "set the compression before the first record batch");
  if (codec != 0 && codec != 1) {
    agpu_set_error("agpu_ipc_writer_set_compression: codec %d (0 = none, 1 = LZ4 frame; ZSTD is not implemented)", (int)codec);
    return AGPU_ERR_UNSUPPORTED;
  }
  w->codec = codec;
  return AGPU_OK;
}

void agpu_ipc_writer_destroy(agpu_ipc_writer* w) {
  if (!w) return;
  if (w->pin[0] || w->pin[1]) {
    int cur = -1;
    (void)hipGetDevice(&cur);
    if (w->pin_device >= 0) (void)hipSetDevice(w->pin_device);
    for (int k = 0; k < 2; k++) {
      if (w->pin_ev[k]) {
        (void)hipEventSynchronize(w->pin_ev[k]);
        (void)hipEventDestroy(w->pin_ev[k]);
      }
      if (w->pin[k]) (void)hipHostFree(w->pin[k]);
    }
    if (cur >= 0) (void)hipSetDevice(cur);
  }
  delete w;
}

}  // extern "C"
