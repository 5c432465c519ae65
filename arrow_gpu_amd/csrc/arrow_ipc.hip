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
#include <string>

#include "common.hpp"

namespace {

// ---------------------------------------------------------------- Flatbuffers, reading (every access bounds-checked)
struct FbBuf {
  const uint8_t* p;
  size_t n;
  bool has(size_t pos, size_t len) const { return pos <= n && len <= n - pos; }
  uint16_t u16(size_t pos) const { uint16_t v; memcpy(&v, p + pos, 2); return v; }
  uint32_t u32(size_t pos) const { uint32_t v; memcpy(&v, p + pos, 4); return v; }
  int32_t i32(size_t pos) const { int32_t v; memcpy(&v, p + pos, 4); return v; }
  int64_t i64(size_t pos) const { int64_t v; memcpy(&v, p + pos, 8); return v; }
};
struct FbTable {
  const FbBuf* b = nullptr;
  size_t pos = 0, vt = 0;
  uint16_t vt_size = 0, tab_size = 0;
  bool ok = false;
};
struct FbVec {
  const FbBuf* b = nullptr;
  size_t first = 0;  // position of element 0
  uint32_t len = 0;
  bool ok = false;
};
static FbTable fb_table_at(const FbBuf& b, size_t pos) {
  FbTable t;
  t.b = &b;
  if (!b.has(pos, 4)) return t;
  const int64_t vt = (int64_t)pos - (int64_t)b.i32(pos);
  if (vt < 0 || !b.has((size_t)vt, 4)) return t;
  t.pos = pos;
  t.vt = (size_t)vt;
  t.vt_size = b.u16(t.vt);
  t.tab_size = b.u16(t.vt + 2);
  if (t.vt_size < 4 || (t.vt_size & 1) || !b.has(t.vt, t.vt_size) || t.tab_size < 4 || !b.has(pos, t.tab_size)) return t;
  t.ok = true;
  return t;
}
static FbTable fb_root(const FbBuf& b) {
  if (!b.has(0, 4)) return FbTable{};
  return fb_table_at(b, b.u32(0));
}
// position of field `id` inside the table, or 0 when absent (default value applies)
static size_t fb_field(const FbTable& t, int id, size_t size) {
  const size_t slot = 4 + 2 * (size_t)id;
  if (!t.ok || slot + 2 > t.vt_size) return 0;
  const uint16_t o = t.b->u16(t.vt + slot);
  if (!o || (size_t)o + size > t.tab_size) return 0;
  return t.pos + o;
}
static int64_t fb_i64(const FbTable& t, int id, int64_t dflt) { const size_t f = fb_field(t, id, 8); return f ? t.b->i64(f) : dflt; }
static int32_t fb_i32(const FbTable& t, int id, int32_t dflt) { const size_t f = fb_field(t, id, 4); return f ? t.b->i32(f) : dflt; }
static int16_t fb_i16(const FbTable& t, int id, int16_t dflt) { const size_t f = fb_field(t, id, 2); return f ? (int16_t)t.b->u16(f) : dflt; }
static uint8_t fb_u8(const FbTable& t, int id, uint8_t dflt) { const size_t f = fb_field(t, id, 1); return f ? t.b->p[f] : dflt; }
static bool fb_present(const FbTable& t, int id) { return fb_field(t, id, 4) != 0; }
static FbTable fb_sub(const FbTable& t, int id) {
  const size_t f = fb_field(t, id, 4);
  if (!f) return FbTable{};
  const uint64_t target = (uint64_t)f + t.b->u32(f);
  if (target > t.b->n) return FbTable{};
  return fb_table_at(*t.b, (size_t)target);
}
static FbVec fb_vec(const FbTable& t, int id, size_t elem) {
  FbVec v;
  v.b = t.b;
  const size_t f = fb_field(t, id, 4);
  if (!f) return v;
  const uint64_t target = (uint64_t)f + t.b->u32(f);
  if (target > t.b->n || !t.b->has((size_t)target, 4)) return v;
  v.len = t.b->u32((size_t)target);
  v.first = (size_t)target + 4;
  if ((uint64_t)v.len * elem > t.b->n || !t.b->has(v.first, (size_t)v.len * elem)) return v;
  v.ok = true;
  return v;
}
static FbTable fb_vec_table(const FbVec& v, uint32_t i) {
  const size_t at = v.first + 4 * (size_t)i;
  const uint64_t target = (uint64_t)at + v.b->u32(at);
  if (target > v.b->n) return FbTable{};
  return fb_table_at(*v.b, (size_t)target);
}
static bool fb_string(const FbTable& t, int id, std::string* out) {
  const FbVec v = fb_vec(t, id, 1);
  if (!v.ok) return false;
  out->assign(reinterpret_cast<const char*>(v.b->p + v.first), v.len);
  return true;
}

// ---------------------------------------------------------------- Flatbuffers, writing (front to back)
struct FbOut {
  std::vector<uint8_t> b;
  void pad_to(size_t align, size_t bias = 0) {  // make (size + bias) a multiple of align
    while ((b.size() + bias) % align) b.push_back(0);
  }
  void raw(const void* v, size_t n) { const uint8_t* s = static_cast<const uint8_t*>(v); b.insert(b.end(), s, s + n); }
  void u16(uint16_t v) { raw(&v, 2); }
  void u32(uint32_t v) { raw(&v, 4); }
  void i64(int64_t v) { raw(&v, 8); }
  void patch32(size_t at, uint32_t v) { memcpy(b.data() + at, &v, 4); }
  void link(size_t ref_at) { patch32(ref_at, (uint32_t)(b.size() - ref_at)); }  // uoffset: from the field to the target that starts HERE
};
struct FbField {
  int id;
  int size;      // 1, 2, 4 or 8
  uint64_t val;  // scalar value; ignored for references
  bool ref;      // a uoffset to be linked later
  size_t at;     // out: absolute position of the field
};
// vtable + table; the table starts 8-byte aligned.  Returns the table's position; fields[i].at = where each value sits.
static size_t fb_write_table(FbOut& o, std::vector<FbField>& fields) {
  int slots = 0;
  for (auto& f : fields) slots = f.id + 1 > slots ? f.id + 1 : slots;
  const size_t vt_size = 4 + 2 * (size_t)slots;
  o.pad_to(8, vt_size);  // the table follows the vtable immediately and must sit on an 8-byte boundary
  const size_t vt = o.b.size();
  std::vector<uint16_t> off((size_t)slots, 0);
  size_t cur = 4;  // after the soffset
  for (int size : {8, 4, 2, 1})
    for (auto& f : fields)
      if (f.size == size) {
        cur = (cur + (size_t)size - 1) / (size_t)size * (size_t)size;
        off[(size_t)f.id] = (uint16_t)cur;
        cur += (size_t)size;
      }
  const size_t tab_size = (cur + 3) / 4 * 4;
  o.u16((uint16_t)vt_size);
  o.u16((uint16_t)tab_size);
  for (uint16_t x : off) o.u16(x);
  const size_t tab = o.b.size();
  o.b.resize(tab + tab_size, 0);
  const int32_t so = (int32_t)(tab - vt);
  memcpy(o.b.data() + tab, &so, 4);
  for (auto& f : fields) {
    f.at = tab + off[(size_t)f.id];
    if (!f.ref) memcpy(o.b.data() + f.at, &f.val, (size_t)f.size);  // little-endian host
  }
  return tab;
}
static void fb_write_string(FbOut& o, size_t ref_at, const std::string& s) {
  o.pad_to(4);
  o.link(ref_at);
  o.u32((uint32_t)s.size());
  o.raw(s.data(), s.size());
  o.b.push_back(0);
}

// ---------------------------------------------------------------- Arrow metadata constants (format/Schema.fbs, Message.fbs)
enum : uint8_t {
  T_NONE = 0, T_Null = 1, T_Int = 2, T_FloatingPoint = 3, T_Binary = 4, T_Utf8 = 5, T_Bool = 6, T_Decimal = 7, T_Date = 8,
  T_Time = 9, T_Timestamp = 10, T_Interval = 11, T_List = 12, T_Struct = 13, T_Union = 14, T_FixedSizeBinary = 15,
  T_FixedSizeList = 16, T_Map = 17, T_Duration = 18, T_LargeBinary = 19, T_LargeUtf8 = 20, T_LargeList = 21,
  T_RunEndEncoded = 22, T_BinaryView = 23, T_Utf8View = 24, T_ListView = 25, T_LargeListView = 26
};
enum : uint8_t { H_NONE = 0, H_Schema = 1, H_DictionaryBatch = 2, H_RecordBatch = 3 };
constexpr int16_t kV4 = 3, kV5 = 4;
constexpr uint32_t kContinuation = 0xFFFFFFFFu;

struct FieldInfo {
  std::string name, format;
  int32_t dtype = -1;  // agpu_dtype or -1
  bool nullable = true;
  int64_t n_nodes = 0, n_buffers = 0;         // of the whole subtree; -1: layout unknown (view types)
  int64_t first_node = -1, first_buffer = -1; // in a record batch's flattened lists; -1: not locatable
};
struct BatchInfo {
  size_t meta_pos, meta_len;  // the Message flatbuffer
  size_t body_pos, body_len;
  int64_t rows;
};

// nodes / buffers one field contributes to a record batch (depth first); false = unknown layout
static bool field_layout(const FbTable& field, int16_t version, int64_t* nodes, int64_t* buffers, int depth) {
  if (!field.ok || depth > 64) return false;
  const uint8_t tt = fb_u8(field, 2, T_NONE);
  int64_t own = 0;
  bool with_children = false;
  if (fb_sub(field, 4).ok) {  // dictionary-encoded: the batch carries the indices (an Int column); children live in the dictionary
    *nodes += 1;
    *buffers += 2;
    return true;
  }
  switch (tt) {
    case T_Null: own = 0; break;
    case T_Int: case T_FloatingPoint: case T_Bool: case T_Decimal: case T_Date: case T_Time: case T_Timestamp:
    case T_Interval: case T_Duration: case T_FixedSizeBinary: own = 2; break;
    case T_Binary: case T_Utf8: case T_LargeBinary: case T_LargeUtf8: own = 3; break;
    case T_List: case T_LargeList: case T_Map: own = 2; with_children = true; break;
    case T_ListView: case T_LargeListView: own = 3; with_children = true; break;
    case T_Struct: case T_FixedSizeList: own = 1; with_children = true; break;
    case T_RunEndEncoded: own = 0; with_children = true; break;
    case T_Union: {
      const FbTable u = fb_sub(field, 3);
      const int16_t mode = u.ok ? fb_i16(u, 0, 0) : 0;  // Sparse = 0, Dense = 1
      own = (mode == 1 ? 2 : 1) + (version < kV5 ? 1 : 0);
      with_children = true;
      break;
    }
    default: return false;  // view types (variadic buffers) and anything newer than this reader
  }
  *nodes += 1;
  *buffers += own;
  if (with_children) {
    const FbVec ch = fb_vec(field, 5, 4);
    if (ch.ok)
      for (uint32_t i = 0; i < ch.len; i++)
        if (!field_layout(fb_vec_table(ch, i), version, nodes, buffers, depth + 1)) return false;
  }
  return true;
}

static void classify_type(const FbTable& field, FieldInfo* fi) {
  const uint8_t tt = fb_u8(field, 2, T_NONE);
  const FbTable ty = fb_sub(field, 3);
  fi->dtype = -1;
  fi->format = "";
  if (fb_sub(field, 4).ok) return;  // dictionary-encoded
  switch (tt) {
    case T_Int: {
      const int32_t bits = ty.ok ? fb_i32(ty, 0, 0) : 0;
      const bool sg = ty.ok && fb_u8(ty, 1, 0) != 0;
      if (bits == 8) { fi->dtype = sg ? AGPU_I8 : AGPU_U8; fi->format = sg ? "c" : "C"; }
      else if (bits == 16) { fi->dtype = sg ? AGPU_I16 : AGPU_U16; fi->format = sg ? "s" : "S"; }
      else if (bits == 32) { fi->dtype = sg ? AGPU_I32 : AGPU_U32; fi->format = sg ? "i" : "I"; }
      else if (bits == 64) fi->format = sg ? "l" : "L";
      break;
    }
    case T_FloatingPoint: {
      const int16_t prec = ty.ok ? fb_i16(ty, 0, 0) : 0;  // HALF, SINGLE, DOUBLE
      if (prec == 1) { fi->dtype = AGPU_F32; fi->format = "f"; }
      else fi->format = prec == 0 ? "e" : "g";
      break;
    }
    case T_Bool: fi->dtype = AGPU_BOOL; fi->format = "b"; break;
    case T_Date: {
      const int16_t unit = ty.ok ? fb_i16(ty, 0, 1) : 1;  // DAY = 0, MILLISECOND = 1 (the schema's default)
      if (unit == 0) { fi->dtype = AGPU_DATE32; fi->format = "tdD"; }
      else fi->format = "tdm";
      break;
    }
    case T_Utf8: fi->format = "u"; break;
    case T_Binary: fi->format = "z"; break;
    case T_LargeUtf8: fi->format = "U"; break;
    case T_LargeBinary: fi->format = "Z"; break;
    case T_Null: fi->format = "n"; break;
    default: break;
  }
}

static size_t bitmap_span_bytes(uint64_t n_bits) { return (size_t)((n_bits + 7) / 8); }

}  // namespace

struct agpu_ipc_reader {
  FbBuf data{nullptr, 0};
  bool is_file = false;
  int16_t version = kV5;
  std::vector<FieldInfo> fields;
  std::vector<BatchInfo> batches;
};

namespace {

static agpu_status parse_schema(agpu_ipc_reader* r, const FbTable& schema) {
  AGPU_REQUIRE(schema.ok, AGPU_ERR_SHAPE, "malformed Schema table");
  if (fb_i16(schema, 0, 0) != 0) {
    agpu_set_error("agpu_ipc_open: big-endian IPC data is not supported");
    return AGPU_ERR_UNSUPPORTED;
  }
  const FbVec fv = fb_vec(schema, 1, 4);
  AGPU_REQUIRE(fv.ok || !fb_present(schema, 1), AGPU_ERR_SHAPE, "malformed Schema.fields");
  int64_t node = 0, buf = 0;
  bool locatable = true;
  for (uint32_t i = 0; fv.ok && i < fv.len; i++) {
    const FbTable f = fb_vec_table(fv, i);
    AGPU_REQUIRE(f.ok, AGPU_ERR_SHAPE, "malformed Field table");
    FieldInfo fi;
    (void)fb_string(f, 0, &fi.name);
    fi.nullable = fb_u8(f, 1, 0) != 0;
    classify_type(f, &fi);
    int64_t nn = 0, nb = 0;
    const bool known = field_layout(f, r->version, &nn, &nb, 0);
    fi.n_nodes = known ? nn : -1;
    fi.n_buffers = known ? nb : -1;
    if (locatable) {
      fi.first_node = node;
      fi.first_buffer = buf;
    }
    if (!known) {
      fi.dtype = -1;
      locatable = false;  // the columns after a variadic-layout column cannot be located from the schema alone
    }
    node += nn;
    buf += nb;
    r->fields.push_back(std::move(fi));
  }
  return AGPU_OK;
}

// one encapsulated message at `pos`; *next = position after its body
static agpu_status parse_message(agpu_ipc_reader* r, size_t pos, bool* end, bool* have_schema, size_t* next) {
  const FbBuf& d = r->data;
  *end = false;
  if (!d.has(pos, 4)) {
    *end = true;  // a stream may simply stop (the end-of-stream marker is optional)
    return AGPU_OK;
  }
  uint32_t size = d.u32(pos);
  pos += 4;
  if (size == kContinuation) {
    AGPU_REQUIRE(d.has(pos, 4), AGPU_ERR_SHAPE, "truncated message prefix");
    size = d.u32(pos);
    pos += 4;
  }  // else: the pre-0.15 framing, the length alone
  if (size == 0) {
    *end = true;
    return AGPU_OK;
  }
  AGPU_REQUIRE(d.has(pos, size), AGPU_ERR_SHAPE, "truncated message metadata");
  FbBuf* mb = new FbBuf{d.p + pos, size};  // tables keep a pointer to their buffer: give it a stable home for this call
  std::unique_ptr<FbBuf> hold(mb);
  const FbTable msg = fb_root(*mb);
  AGPU_REQUIRE(msg.ok, AGPU_ERR_SHAPE, "malformed Message flatbuffer");
  const int16_t version = fb_i16(msg, 0, 0);
  const uint8_t htype = fb_u8(msg, 1, H_NONE);
  const int64_t body_len = fb_i64(msg, 3, 0);
  AGPU_REQUIRE(body_len >= 0 && d.has(pos + size, (size_t)body_len), AGPU_ERR_SHAPE, "truncated message body");
  const size_t body_pos = pos + size;
  *next = body_pos + (size_t)body_len;
  if (htype == H_Schema) {
    AGPU_REQUIRE(!*have_schema, AGPU_ERR_SHAPE, "second Schema message");
    if (version < kV4) {
      agpu_set_error("agpu_ipc_open: metadata version %d predates V4", (int)version);
      return AGPU_ERR_UNSUPPORTED;
    }
    r->version = version;
    agpu_status st = parse_schema(r, fb_sub(msg, 2));
    if (st != AGPU_OK) return st;
    *have_schema = true;
  } else if (htype == H_RecordBatch) {
    AGPU_REQUIRE(*have_schema, AGPU_ERR_SHAPE, "RecordBatch before Schema");
    const FbTable rb = fb_sub(msg, 2);
    AGPU_REQUIRE(rb.ok, AGPU_ERR_SHAPE, "malformed RecordBatch table");
    BatchInfo bi{pos, size, body_pos, (size_t)body_len, fb_i64(rb, 0, 0)};
    AGPU_REQUIRE(bi.rows >= 0, AGPU_ERR_SHAPE, "negative RecordBatch.length");
    r->batches.push_back(bi);
  }  // DictionaryBatch / Tensor / SparseTensor: skipped (dictionary columns are reported as unsupported per column)
  return AGPU_OK;
}

static agpu_status parse_stream(agpu_ipc_reader* r, size_t pos) {
  bool have_schema = false, end = false;
  while (!end) {
    size_t next = pos;
    agpu_status st = parse_message(r, pos, &end, &have_schema, &next);
    if (st != AGPU_OK) return st;
    pos = next;
  }
  AGPU_REQUIRE(have_schema, AGPU_ERR_SHAPE, "no Schema message");
  return AGPU_OK;
}

// File format: the Footer lists every record batch as a Block {offset, metaDataLength, bodyLength}
static agpu_status parse_file(agpu_ipc_reader* r) {
  const FbBuf& d = r->data;
  AGPU_REQUIRE(d.n >= 8 + 4 + 6 && !memcmp(d.p + d.n - 6, "ARROW1", 6), AGPU_ERR_SHAPE, "file does not end with the ARROW1 magic");
  const int32_t fsize = d.i32(d.n - 10);
  AGPU_REQUIRE(fsize > 0 && (size_t)fsize <= d.n - 18, AGPU_ERR_SHAPE, "bad footer size");
  FbBuf fb{d.p + d.n - 10 - (size_t)fsize, (size_t)fsize};
  const FbTable footer = fb_root(fb);
  AGPU_REQUIRE(footer.ok, AGPU_ERR_SHAPE, "malformed Footer flatbuffer");
  r->version = fb_i16(footer, 0, 0);
  if (r->version < kV4) {
    agpu_set_error("agpu_ipc_open: metadata version %d predates V4", (int)r->version);
    return AGPU_ERR_UNSUPPORTED;
  }
  agpu_status st = parse_schema(r, fb_sub(footer, 1));
  if (st != AGPU_OK) return st;
  const FbVec blocks = fb_vec(footer, 3, 24);
  AGPU_REQUIRE(blocks.ok || !fb_present(footer, 3), AGPU_ERR_SHAPE, "malformed Footer.recordBatches");
  for (uint32_t i = 0; blocks.ok && i < blocks.len; i++) {
    const size_t at = blocks.first + 24 * (size_t)i;
    const int64_t off = fb.i64(at);
    AGPU_REQUIRE(off >= 8 && (uint64_t)off < d.n, AGPU_ERR_SHAPE, "record batch block outside the file");
    bool have_schema = true, end = false;
    size_t next = 0;
    const size_t before = r->batches.size();
    st = parse_message(r, (size_t)off, &end, &have_schema, &next);
    if (st != AGPU_OK) return st;
    AGPU_REQUIRE(!end && r->batches.size() == before + 1, AGPU_ERR_SHAPE, "footer block does not point at a RecordBatch");
  }
  return AGPU_OK;
}

struct ViewPrivate {
  const void* buffers[2];
  std::string format, name;
};
static void release_view_array(struct ArrowArray* a) {
  if (!a || !a->release) return;
  delete static_cast<ViewPrivate*>(a->private_data);
  a->release = nullptr;
}
static void release_view_schema(struct ArrowSchema* s) {
  if (!s || !s->release) return;
  delete static_cast<ViewPrivate*>(s->private_data);
  s->release = nullptr;
}

static agpu_status column_buffers(const agpu_ipc_reader* r, int64_t batch, int32_t column, const FieldInfo** out_fi,
                                  int64_t* rows, int64_t* null_count, const uint8_t** validity, const uint8_t** values) {
  AGPU_REQUIRE(r, AGPU_ERR_ARG, "null reader");
  AGPU_REQUIRE(batch >= 0 && (size_t)batch < r->batches.size(), AGPU_ERR_ARG, "batch index out of range");
  AGPU_REQUIRE(column >= 0 && (size_t)column < r->fields.size(), AGPU_ERR_ARG, "column index out of range");
  const FieldInfo& fi = r->fields[(size_t)column];
  if (fi.dtype < 0 || fi.first_node < 0) {
    agpu_set_error("agpu_ipc: column %d ('%s', format '%s') has no GPU array type (i8 u8 i16 u16 i32 u32 f32 bool date32)",
                   (int)column, fi.name.c_str(), fi.format.c_str());
    return AGPU_ERR_UNSUPPORTED;
  }
  const BatchInfo& bi = r->batches[(size_t)batch];
  FbBuf mb{r->data.p + bi.meta_pos, bi.meta_len};
  const FbTable rb = fb_sub(fb_root(mb), 2);
  AGPU_REQUIRE(rb.ok, AGPU_ERR_SHAPE, "malformed RecordBatch table");
  if (fb_present(rb, 3)) {
    agpu_set_error("agpu_ipc: compressed record batch bodies (LZ4 / ZSTD) are not supported");
    return AGPU_ERR_UNSUPPORTED;
  }
  const FbVec nodes = fb_vec(rb, 1, 16), bufs = fb_vec(rb, 2, 16);
  AGPU_REQUIRE(nodes.ok && bufs.ok, AGPU_ERR_SHAPE, "malformed RecordBatch nodes / buffers");
  AGPU_REQUIRE((uint64_t)fi.first_node < nodes.len && (uint64_t)fi.first_buffer + 2 <= bufs.len, AGPU_ERR_SHAPE,
               "RecordBatch has fewer nodes / buffers than the schema requires");
  const size_t nat = nodes.first + 16 * (size_t)fi.first_node;
  const int64_t len = mb.i64(nat), nulls = mb.i64(nat + 8);
  AGPU_REQUIRE(len >= 0 && nulls >= 0 && nulls <= len, AGPU_ERR_SHAPE, "bad FieldNode");
  AGPU_REQUIRE((uint64_t)len <= (uint64_t)bi.body_len * 8, AGPU_ERR_SHAPE, "FieldNode.length exceeds what the body can hold");  // also keeps len × width from overflowing
  const size_t bat = bufs.first + 16 * (size_t)fi.first_buffer;
  const int64_t voff = mb.i64(bat), vlen = mb.i64(bat + 8), doff = mb.i64(bat + 16), dlen = mb.i64(bat + 24);
  AGPU_REQUIRE(voff >= 0 && vlen >= 0 && doff >= 0 && dlen >= 0, AGPU_ERR_SHAPE, "negative buffer offset / length");
  AGPU_REQUIRE((uint64_t)voff <= bi.body_len && (uint64_t)vlen <= bi.body_len - (uint64_t)voff && (uint64_t)doff <= bi.body_len &&
                   (uint64_t)dlen <= bi.body_len - (uint64_t)doff,
               AGPU_ERR_SHAPE, "buffer outside the message body");
  const size_t need = fi.dtype == AGPU_BOOL ? bitmap_span_bytes((uint64_t)len) : (size_t)len * agpu_dtype_size((agpu_dtype)fi.dtype);
  AGPU_REQUIRE((uint64_t)dlen >= need, AGPU_ERR_SHAPE, "values buffer shorter than the column");
  AGPU_REQUIRE(nulls == 0 || (uint64_t)vlen >= bitmap_span_bytes((uint64_t)len), AGPU_ERR_SHAPE, "validity buffer shorter than the column");
  *out_fi = &fi;
  *rows = len;
  *null_count = nulls;
  *validity = (nulls > 0 && vlen > 0) ? r->data.p + bi.body_pos + (size_t)voff : nullptr;
  *values = r->data.p + bi.body_pos + (size_t)doff;
  return AGPU_OK;
}

}  // namespace

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
                                const std::vector<size_t>& vbytes, const std::vector<size_t>& dbytes, int64_t* body_len) {
  o.u32(0);
  std::vector<FbField> mf = {{0, 2, (uint64_t)kV5, false, 0}, {1, 1, H_RecordBatch, false, 0}, {2, 4, 0, true, 0}, {3, 8, 0, false, 0}};
  const size_t msg = fb_write_table(o, mf);
  o.patch32(0, (uint32_t)msg);
  std::vector<FbField> rf = {{0, 8, (uint64_t)rows, false, 0}, {1, 4, 0, true, 0}, {2, 4, 0, true, 0}};
  const size_t rb = fb_write_table(o, rf);
  o.patch32(mf[2].at, (uint32_t)(rb - mf[2].at));
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

}  // namespace

extern "C" {

agpu_status agpu_ipc_open(const void* data, uint64_t bytes, agpu_ipc_reader** out_reader) {
  AGPU_REQUIRE(data && out_reader, AGPU_ERR_ARG, "null argument");
  *out_reader = nullptr;
  std::unique_ptr<agpu_ipc_reader> r(new agpu_ipc_reader);
  r->data = FbBuf{static_cast<const uint8_t*>(data), (size_t)bytes};
  agpu_status st;
  if (bytes >= 8 && !memcmp(data, "ARROW1\0\0", 8)) {
    r->is_file = true;
    st = parse_file(r.get());
  } else {
    st = parse_stream(r.get(), 0);
  }
  if (st != AGPU_OK) return st;
  *out_reader = r.release();
  return AGPU_OK;
}

void agpu_ipc_close(agpu_ipc_reader* r) { delete r; }

agpu_status agpu_ipc_num_fields(const agpu_ipc_reader* r, int32_t* out_n) {
  AGPU_REQUIRE(r && out_n, AGPU_ERR_ARG, "null argument");
  *out_n = (int32_t)r->fields.size();
  return AGPU_OK;
}

agpu_status agpu_ipc_field_info(const agpu_ipc_reader* r, int32_t i, agpu_ipc_field* out) {
  AGPU_REQUIRE(r && out, AGPU_ERR_ARG, "null argument");
  AGPU_REQUIRE(i >= 0 && (size_t)i < r->fields.size(), AGPU_ERR_ARG, "field index out of range");
  const FieldInfo& fi = r->fields[(size_t)i];
  out->name = fi.name.c_str();
  out->format = fi.format.c_str();
  out->dtype = fi.dtype;
  out->nullable = fi.nullable ? 1 : 0;
  return AGPU_OK;
}

agpu_status agpu_ipc_num_batches(const agpu_ipc_reader* r, int64_t* out_n) {
  AGPU_REQUIRE(r && out_n, AGPU_ERR_ARG, "null argument");
  *out_n = (int64_t)r->batches.size();
  return AGPU_OK;
}

agpu_status agpu_ipc_batch_rows(const agpu_ipc_reader* r, int64_t batch, int64_t* out_rows) {
  AGPU_REQUIRE(r && out_rows, AGPU_ERR_ARG, "null argument");
  AGPU_REQUIRE(batch >= 0 && (size_t)batch < r->batches.size(), AGPU_ERR_ARG, "batch index out of range");
  *out_rows = r->batches[(size_t)batch].rows;
  return AGPU_OK;
}

agpu_status agpu_ipc_column_view(const agpu_ipc_reader* r, int64_t batch, int32_t column, struct ArrowArray* out_array,
                                 struct ArrowSchema* out_schema) {
  AGPU_REQUIRE(out_array && out_schema, AGPU_ERR_ARG, "null argument");
  const FieldInfo* fi = nullptr;
  int64_t rows = 0, nulls = 0;
  const uint8_t *validity = nullptr, *values = nullptr;
  agpu_status st = column_buffers(r, batch, column, &fi, &rows, &nulls, &validity, &values);
  if (st != AGPU_OK) return st;
  ViewPrivate* ap = new ViewPrivate{{validity, values}, fi->format, fi->name};
  ViewPrivate* sp = new ViewPrivate{{nullptr, nullptr}, fi->format, fi->name};
  memset(out_array, 0, sizeof(*out_array));
  out_array->length = rows;
  out_array->null_count = nulls;
  out_array->n_buffers = 2;
  out_array->buffers = ap->buffers;
  out_array->release = release_view_array;
  out_array->private_data = ap;
  memset(out_schema, 0, sizeof(*out_schema));
  out_schema->format = sp->format.c_str();
  out_schema->name = sp->name.c_str();
  out_schema->flags = fi->nullable ? ARROW_FLAG_NULLABLE : 0;
  out_schema->release = release_view_schema;
  out_schema->private_data = sp;
  return AGPU_OK;
}

agpu_status agpu_ipc_read_column(const agpu_ipc_reader* r, int64_t batch, int32_t column, agpu_pipeline* p,
                                 agpu_arrow_column* out_column) {
  AGPU_REQUIRE(p && out_column, AGPU_ERR_ARG, "null argument");
  struct ArrowArray a;
  struct ArrowSchema s;
  agpu_status st = agpu_ipc_column_view(r, batch, column, &a, &s);
  if (st != AGPU_OK) return st;
  st = agpu_import_arrow(p, &a, &s, out_column);
  a.release(&a);
  s.release(&s);
  return st;
}

// ---------------------------------------------------------------- writer
agpu_status agpu_ipc_writer_create(const agpu_ipc_field* fields, int32_t n_fields, int32_t file_format, int32_t fd,
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

agpu_status agpu_ipc_writer_write_batch(agpu_ipc_writer* w, const struct ArrowArray* const* columns) {
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

agpu_status agpu_ipc_writer_write_device_batch(agpu_ipc_writer* w, agpu_pipeline* p, const agpu_arrow_column* columns) {
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

agpu_status agpu_ipc_writer_finish(agpu_ipc_writer* w, const void** out_data, uint64_t* out_bytes) {
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
