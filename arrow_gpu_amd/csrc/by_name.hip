// by_name.hip — agpu_launch_by_name: the reference's kernel identity (WGSL file, @compute entry point) → typed ABI.
//
// In the reference a kernel is identified by (shader text, entry-point name): that pair is the pipeline-cache key of
// GpuDevice::create_compute_pipeline (crates/array/src/gpu_utils/gpu_device.rs:145-168) and what every op macro passes
// to apply_{unary,binary,scalar,ternary,broadcast}_function.  A Rust shim that keeps those call sites can forward the
// shader's path ("<crate>/<dir>/<file>", e.g. "arithmetic/f32/array") and the entry point here unchanged.
#include <string>

#include "common.hpp"

agpu_status agpu_bitmap_andnot_internal(agpu_pipeline* p, const void* a, const void* b, void* out, uint64_t n_bits);

#include <mutex>
#include <set>

// "<shader_key>::<entry_point>" with static lifetime (agpu_pipeline_last_kernel_ns hands the pointer out)
static const char* intern_label(const char* shader_key, const char* entry_point) {
  static std::mutex mu;
  static std::set<std::string> names;
  std::lock_guard<std::mutex> lock(mu);
  return names.insert(std::string(shader_key) + "::" + entry_point).first->c_str();
}

agpu_status agpu_internal_sum_level(agpu_pipeline* p, agpu_dtype dtype, const void* in, uint64_t m, void* out, uint64_t groups);

static bool dtype_of(const std::string& d, agpu_dtype* out) {
  if (d == "f32") *out = AGPU_F32;
  else if (d == "i32") *out = AGPU_I32;
  else if (d == "u32") *out = AGPU_U32;
  else if (d == "i16") *out = AGPU_I16;
  else if (d == "u16") *out = AGPU_U16;
  else if (d == "i8") *out = AGPU_I8;
  else if (d == "u8") *out = AGPU_U8;
  else if (d == "boolean" || d == "bool") *out = AGPU_BOOL;
  else return false;
  return true;
}

static bool binop_of(const std::string& s, agpu_binary_op* out) {
  if (s == "add") *out = AGPU_OP_ADD;
  else if (s == "sub") *out = AGPU_OP_SUB;
  else if (s == "mul") *out = AGPU_OP_MUL;
  else if (s == "div") *out = AGPU_OP_DIV;
  else if (s == "rem") *out = AGPU_OP_REM;
  else return false;
  return true;
}

static bool cmp_of(const std::string& s, agpu_cmp_op* out) {
  if (s == "gt") *out = AGPU_CMP_GT;
  else if (s == "gteq") *out = AGPU_CMP_GTEQ;
  else if (s == "lt") *out = AGPU_CMP_LT;
  else if (s == "lteq") *out = AGPU_CMP_LTEQ;
  else if (s == "eq") *out = AGPU_CMP_EQ;
  else return false;
  return true;
}

static bool unary_of(const std::string& s, agpu_unary_op* out) {
  if (s == "sqrt") *out = AGPU_UN_SQRT;
  else if (s == "exp") *out = AGPU_UN_EXP;
  else if (s == "exp2") *out = AGPU_UN_EXP2;
  else if (s == "log") *out = AGPU_UN_LOG;
  else if (s == "log2") *out = AGPU_UN_LOG2;
  else if (s == "abs") *out = AGPU_UN_ABS;
  else if (s == "cbrt") *out = AGPU_UN_CBRT;
  else if (s == "sin") *out = AGPU_UN_SIN;
  else if (s == "cos") *out = AGPU_UN_COS;
  else if (s == "acos") *out = AGPU_UN_ACOS;
  else if (s == "sinh") *out = AGPU_UN_SINH;
  else return false;
  return true;
}

// ---------------------------------------------------------------- the reference's shader argument is the TEXT
// Every op crate holds `const X_SHADER: &str = include_str!("…wgsl")` or `concat!(include_str!("…/utils.wgsl"),
// include_str!("…wgsl"))` and hands that text to apply_*_function; (text, entry point) is the pipeline-cache key
// [ref: crates/arithmetic/src/f32.rs:10-15, crates/compare/src/u8.rs:3-12, gpu_device.rs:145-168].  So that a shim can pass
// those constants through UNCHANGED, the `shader_key` argument of the by-name entry points accepts either the path key or
// the text itself: a text is recognised by its 64-bit FNV-1a hash + byte length in a table generated from the reference's
// 76 constants (71 distinct texts) by tools/extract_entry_points.py — hashes are data, no WGSL is shipped.  Two files of
// the reference are byte-identical (compare/u32/min_max.wgsl = compare/i32/min_max.wgsl, both over array<i32>: its u32
// min / max compare as signed); such a text resolves to the program it IS — the i32 kernel.
namespace {
struct ShaderHash {
  uint64_t hash;
  uint32_t bytes;
  const char* key;
};
const ShaderHash kShaderHashes[] = {
#include "shader_hashes.inc"
};
uint64_t fnv1a64(const char* s, size_t n) {
  uint64_t h = 0xCBF29CE484222325ull;
  for (size_t i = 0; i < n; i++) h = (h ^ (uint8_t)s[i]) * 0x100000001B3ull;
  return h;
}
const char* key_of_hash(uint64_t h, uint64_t n) {
  for (const ShaderHash& e : kShaderHashes)
    if (e.hash == h && e.bytes == n) return e.key;
  return nullptr;
}
// a path key is short and has no white space; anything else is taken for shader text
bool looks_like_key(const char* s, size_t n) {
  if (n == 0 || n > 96) return false;
  int slashes = 0;
  for (size_t i = 0; i < n; i++) {
    const unsigned char c = (unsigned char)s[i];
    if (c <= ' ' || c == '@' || c == '{' || c == ';') return false;
    slashes += c == '/';
  }
  return slashes == 2;
}
// → the path key to dispatch on (static storage or the caller's own string), or nullptr with the error set
const char* resolve_shader(const char* shader) {
  const size_t n = strlen(shader);
  if (shader[0] == '#') {  // "#<16 hex digits of the FNV-1a 64>:<byte length>": a text named by its hash (a shim may hash its constants once)
    unsigned long long h = 0, len = 0;
    if (sscanf(shader + 1, "%16llx:%llu", &h, &len) == 2) {
      const char* key = key_of_hash(h, len);
      if (!key) agpu_set_error("no shader constant of the reference has FNV-1a %016llx / %llu bytes", h, len);
      return key;
    }
    agpu_set_error("malformed shader hash reference '%.40s' (want #<16 hex>:<bytes>)", shader);
    return nullptr;
  }
  if (looks_like_key(shader, n)) return shader;
  const char* key = key_of_hash(fnv1a64(shader, n), n);
  if (!key) agpu_set_error("shader text of %zu bytes is none of the reference's %zu shader constants (FNV-1a %016llx)", n,
                           sizeof(kShaderHashes) / sizeof(kShaderHashes[0]), (unsigned long long)fnv1a64(shader, n));
  return key;
}
}  // namespace

extern "C" agpu_status agpu_shader_key_for_hash(uint64_t fnv1a64_of_text, uint64_t text_bytes, char* out_key, size_t out_cap) {
  AGPU_REQUIRE(out_key && out_cap > 0, AGPU_ERR_ARG, "null out_key");
  const char* key = key_of_hash(fnv1a64_of_text, text_bytes);
  if (!key) {
    agpu_set_error("no shader constant of the reference has FNV-1a %016llx / %llu bytes", (unsigned long long)fnv1a64_of_text,
                   (unsigned long long)text_bytes);
    return AGPU_ERR_UNSUPPORTED;
  }
  AGPU_REQUIRE(strlen(key) < out_cap, AGPU_ERR_ARG, "out_key too small (64 bytes suffice)");
  strcpy(out_key, key);
  return AGPU_OK;
}

extern "C" agpu_status agpu_shader_key_for_source(const char* wgsl, size_t len, char* out_key, size_t out_cap) {
  AGPU_REQUIRE(wgsl, AGPU_ERR_ARG, "null shader text");
  return agpu_shader_key_for_hash(fnv1a64(wgsl, len), len, out_key, out_cap);
}

extern "C" agpu_status agpu_launch_by_name(agpu_pipeline* p, const char* shader_key, const char* entry_point,
                                           const void* const* inputs, int32_t n_inputs, void* out, uint64_t n) {
  AGPU_REQUIRE(p && shader_key && entry_point && (n_inputs == 0 || inputs), AGPU_ERR_ARG, "null argument");
  shader_key = resolve_shader(shader_key);  // the path key, or the reference's shader TEXT
  if (!shader_key) return AGPU_ERR_UNSUPPORTED;
  AGPU_BIND_AS(p, "agpu_launch_by_name");
  agpu_scope_label(p, intern_label(shader_key, entry_point));  // [ref: insert_debug_marker(entry_point) gpu_device.rs:132]
  const std::string key(shader_key), ep(entry_point);
  const size_t s1 = key.find('/'), s2 = key.rfind('/');
  AGPU_REQUIRE(s1 != std::string::npos && s2 != s1, AGPU_ERR_ARG, "shader_key must be <crate>/<dir>/<file>");
  const std::string crate = key.substr(0, s1), dir = key.substr(s1 + 1, s2 - s1 - 1), file = key.substr(s2 + 1);
  agpu_dtype dt = AGPU_U32;
  const bool has_dt = dtype_of(dir, &dt);
  auto need = [&](int k) { return n_inputs >= k; };
#define IN(k) inputs[k]
#define NEED(k) AGPU_REQUIRE(need(k), AGPU_ERR_ARG, "too few input bindings for this entry point")

  if (crate == "arithmetic" && has_dt) {
    if (file == "array") {
      NEED(2);
      if (ep == "bitwise_and") return agpu_binary(p, AGPU_OP_AND, dt, IN(0), IN(1), out, n);
      if (ep == "bitwise_or") return agpu_binary(p, AGPU_OP_OR, dt, IN(0), IN(1), out, n);
      const size_t us = ep.find('_');  // add_f32
      agpu_binary_op op;
      if (us != std::string::npos && binop_of(ep.substr(0, us), &op)) return agpu_binary(p, op, dt, IN(0), IN(1), out, n);
    } else if (file == "scalar") {
      NEED(2);
      const size_t us = ep.find('_');  // f32_add
      agpu_binary_op op;
      if (us != std::string::npos && binop_of(ep.substr(us + 1), &op)) return agpu_scalar(p, op, dt, IN(0), IN(1), out, n);
    } else if (file == "neg" && ep == "neg") {
      NEED(1);
      return agpu_unary(p, AGPU_UN_NEG, dt, IN(0), out, n);
    } else if (file == "aggregate" && ep == "sum") {
      NEED(1);  // n = INPUT rows here; the whole multi-level tree runs, out = 1 element
      return agpu_reduce(p, AGPU_RED_SUM, dt, IN(0), nullptr, n, out);
    }
  } else if (crate == "array" && has_dt && file == "broadcast" && ep == "broadcast") {
    NEED(1);
    return agpu_broadcast_from_device(p, dt, IN(0), out, n);
  } else if (crate == "logical" && has_dt) {
    if (file == "logical") {
      NEED(2);
      if (ep == "bitwise_and") return agpu_binary(p, AGPU_OP_AND, dt, IN(0), IN(1), out, n);
      if (ep == "bitwise_or") return agpu_binary(p, AGPU_OP_OR, dt, IN(0), IN(1), out, n);
      if (ep == "bitwise_xor") return agpu_binary(p, AGPU_OP_XOR, dt, IN(0), IN(1), out, n);
    } else if (file == "not" && ep == "bitwise_not") {
      NEED(1);
      return agpu_unary(p, AGPU_UN_NOT, dt, IN(0), out, n);
    } else if (file == "shift") {
      NEED(2);
      if (ep == "bitwise_shl") return agpu_binary(p, AGPU_OP_SHL, dt, IN(0), IN(1), out, n);
      if (ep == "bitwise_shr") return agpu_binary(p, AGPU_OP_SHR, dt, IN(0), IN(1), out, n);
    } else if (file == "any" && ep == "any") {
      NEED(1);  // n = bits
      return agpu_bitmap_any(p, IN(0), n, static_cast<uint32_t*>(out));
    } else if (file == "countbitones" && ep == "countob") {
      NEED(1);  // n = 32-bit words; out[i] = countOneBits(in[i])  [logical/u32/countbitones.wgsl:9-15]
      return agpu_unary(p, AGPU_UN_POPCOUNT, dt, IN(0), out, n);
    }
  } else if (crate == "compare" && has_dt) {
    NEED(2);
    if (file == "cmp") {
      agpu_cmp_op op;
      if (cmp_of(ep, &op)) return agpu_compare(p, op, dt, IN(0), IN(1), out, n);
    } else if (file == "min_max") {
      if (ep == "max_") return agpu_binary(p, AGPU_OP_MAX, dt, IN(0), IN(1), out, n);
      if (ep == "min_") return agpu_binary(p, AGPU_OP_MIN, dt, IN(0), IN(1), out, n);
    }
  } else if (crate == "cast" && has_dt && file.rfind("cast_", 0) == 0 && ep == file) {
    NEED(1);
    agpu_dtype to;
    if (dtype_of(file.substr(5), &to)) return agpu_cast(p, dt, to, IN(0), out, n);
  } else if (crate == "math" && has_dt) {
    if ((file == "floatunary" || file == "unary") && !ep.empty() && ep.back() == '_') {
      NEED(1);
      agpu_unary_op op;
      if (unary_of(ep.substr(0, ep.size() - 1), &op)) return agpu_unary(p, op, dt, IN(0), out, n);
    } else if ((file == "floatbinary" || file == "binary") && ep == "power_") {
      NEED(2);
      return agpu_binary(p, AGPU_OP_POW, dt, IN(0), IN(1), out, n);
    }
  } else if (crate == "trigonometry" && has_dt) {
    NEED(1);
    const size_t us = ep.rfind('_');  // sin_u8
    agpu_unary_op op;
    if (us != std::string::npos && unary_of(ep.substr(0, us), &op)) return agpu_unary(p, op, dt, IN(0), out, n);
  } else if (crate == "routines") {
    int width = 0;
    if (dir == "32bit") width = 4;
    else if (dir == "16bit") width = 2;
    else if (dir == "8bit") width = 1;
    if (width && file == "take" && ep == "take") {
      NEED(2);  // n = number of indexes
      return agpu_take(p, width, IN(0), UINT64_MAX, static_cast<const uint32_t*>(IN(1)), out, n);
    }
    if (width && file == "put" && ep == "put") {
      NEED(3);  // bindings: src, (dst = out), src_indexes, dst_indexes — pass inputs = {src, src_idx, dst_idx}
      return agpu_put(p, width, IN(0), static_cast<const uint32_t*>(IN(1)), out, static_cast<const uint32_t*>(IN(2)), n);
    }
    if (width && file == "merge" && ep == "merge_array") {
      NEED(3);
      return agpu_merge(p, width, IN(0), IN(1), IN(2), out, n);
    }
    if (dir == "bool") {
      if (file == "merge" && ep == "merge_array") {
        NEED(3);
        return agpu_merge_bits(p, IN(0), IN(1), IN(2), out, n);
      }
      if (file == "take" && ep == "take") {
        NEED(2);
        return agpu_take_bits(p, IN(0), UINT64_MAX, static_cast<const uint32_t*>(IN(1)), out, n);
      }
      if (file == "put" && ep == "put") {
        NEED(3);
        return agpu_put_bits(p, IN(0), static_cast<const uint32_t*>(IN(1)), out, static_cast<const uint32_t*>(IN(2)), n);
      }
    }
    if (dir == "u32" && file == "merge_null_buffer") {
      NEED(2);  // n = bits
      if (ep == "merge_selected" || ep == "merge_nulls") return agpu_bitmap_binary(p, AGPU_OP_AND, IN(0), IN(1), out, n);
      if (ep == "merge_or") return agpu_bitmap_binary(p, AGPU_OP_OR, IN(0), IN(1), out, n);
      if (ep == "merge_not_selected") return agpu_bitmap_andnot_internal(p, IN(0), IN(1), out, n);
    }
  }
#undef IN
#undef NEED
  agpu_set_error("no kernel for shader '%s' entry point '%s'", shader_key, entry_point);
  return AGPU_ERR_UNSUPPORTED;
}

// ---------------------------------------------------------------- the byte-sized form: the reference's literal call
// apply_{unary,binary,scalar,ternary,broadcast}_function(buffers…, new_buffer_size, shader, entry_point, dispatch_size)
// [ref: compute_pipeline.rs:24-256, gpu_device.rs:267-509; routines::apply_take_op take.rs:9-55, apply_put_op
//  put.rs:9-56; cast::apply_boolean_unary_function boolean_cast.rs:8-55] carries NO element count: the WGSL runs
// dispatch_size × 256 invocations and wgpu's robust buffer access drops whatever falls outside a binding.  The count is
// therefore derived here the way the shader would see it — from the byte sizes of the bound buffers and the dispatch —
// and every lane that is fully inside its bindings is processed, INCLUDING the padding lanes of sub-word columns
// (buffers are multiples of 4 bytes; a 5-element u8 column is 2 words = 8 lanes), exactly like the WGSL does.  Lanes
// whose inputs or output fall outside a binding are skipped (robust access leaves their value unspecified).
namespace {
struct Sized {
  uint64_t inv;         // invocations = dispatch_size * 256
  const uint64_t* ib;   // input byte sizes
  int n_in;
  uint64_t ob;          // output byte size
  uint64_t W(int k) const { return k < n_in ? ib[k] / 4 : 0; }
  uint64_t OW() const { return ob / 4; }
};
inline uint64_t min3(uint64_t a, uint64_t b, uint64_t c) { return a < b ? (a < c ? a : c) : (b < c ? b : c); }
inline uint64_t min2(uint64_t a, uint64_t b) { return a < b ? a : b; }
inline int lanes_of(agpu_dtype dt) {
  const size_t w = agpu_dtype_size(dt);
  return w ? (int)(4 / w) : 32;
}
}  // namespace

extern "C" agpu_status agpu_launch_by_name_sized(agpu_pipeline* p, const char* shader_key, const char* entry_point,
                                                 const void* const* inputs, const uint64_t* input_bytes, int32_t n_inputs,
                                                 void* out, uint64_t out_bytes, uint32_t dispatch_size) {
  AGPU_REQUIRE(p && shader_key && entry_point && (n_inputs == 0 || (inputs && input_bytes)), AGPU_ERR_ARG, "null argument");
  AGPU_REQUIRE(n_inputs >= 0 && n_inputs <= 4, AGPU_ERR_ARG, "0..4 read bindings");
  shader_key = resolve_shader(shader_key);  // the path key, or the reference's shader TEXT (its literal argument)
  if (!shader_key) return AGPU_ERR_UNSUPPORTED;
  for (int k = 0; k < n_inputs; k++)
    AGPU_REQUIRE(input_bytes[k] % 4 == 0, AGPU_ERR_SHAPE, "wgpu buffers are multiples of 4 bytes");
  AGPU_REQUIRE(out_bytes % 4 == 0, AGPU_ERR_SHAPE, "wgpu buffers are multiples of 4 bytes");
  AGPU_BIND_AS(p, "agpu_launch_by_name_sized");
  agpu_scope_label(p, intern_label(shader_key, entry_point));  // [ref: insert_debug_marker(entry_point) gpu_device.rs:132]
  const std::string key(shader_key), ep(entry_point);
  const size_t s1 = key.find('/'), s2 = key.rfind('/');
  AGPU_REQUIRE(s1 != std::string::npos && s2 != s1, AGPU_ERR_ARG, "shader_key must be <crate>/<dir>/<file>");
  const std::string crate = key.substr(0, s1), dir = key.substr(s1 + 1, s2 - s1 - 1), file = key.substr(s2 + 1);
  agpu_dtype dt = AGPU_U32;
  const bool has_dt = dtype_of(dir, &dt);
  const Sized z{(uint64_t)dispatch_size * 256, input_bytes, n_inputs, out_bytes};
  const int L = has_dt ? lanes_of(dt) : 1;
#define NEED(k) AGPU_REQUIRE(n_inputs >= (k), AGPU_ERR_ARG, "too few input bindings for this entry point")
  uint64_t n = 0;
  bool known = true;
  if (crate == "arithmetic" && has_dt && file == "aggregate" && ep == "sum") {
    // ONE level of the reference's tree per dispatch: workgroup g writes out[g] = tree sum of in[256g .. 256g+255],
    // rows past arrayLength(&input) count as 0 [ref: aggregate.wgsl:21-41; the loop over levels is the caller's,
    // aggregate_kernels.rs:26-43]
    NEED(1);
    return agpu_internal_sum_level(p, dt, inputs[0], z.W(0), out, min2((uint64_t)dispatch_size, z.OW()));
  } else if (crate == "arithmetic" && has_dt && file == "scalar") {
    NEED(2);  // binding 1 is the 1-element operand
    n = min3(z.inv, z.W(0), z.OW()) * (uint64_t)L;
  } else if (crate == "array" && has_dt && file == "broadcast") {
    NEED(1);
    n = min2(z.inv, z.OW());
  } else if (crate == "compare" && has_dt && file == "cmp") {
    NEED(2);  // one invocation per input word (L lanes), one output word per 32 lanes
    n = min2(min3(z.inv, z.W(0), z.W(1)) * (uint64_t)L, z.OW() * 32);
  } else if (crate == "logical" && has_dt && file == "shift" && L > 1) {
    NEED(2);  // one data word (L lanes) + L separate u32 shift amounts per invocation
    n = min3(min2(z.inv, z.W(0)), z.OW(), z.W(1) / (uint64_t)L) * (uint64_t)L;
  } else if (crate == "logical" && has_dt && file == "any") {
    NEED(1);
    n = min2(z.inv, z.W(0)) * 32;  // bits
  } else if (crate == "cast" && has_dt && dt == AGPU_F32) {
    NEED(1);  // cast_u8: one invocation per OUTPUT word, 4 f32 in
    n = min2(min2(z.inv, z.OW()) * 4, z.W(0));
  } else if (crate == "cast" && has_dt && dt == AGPU_BOOL) {
    NEED(1);  // one invocation per output f32, guarded by arrayLength(&new_values)
    n = min3(z.inv, z.OW(), z.W(0) * 32);
  } else if ((crate == "cast" || crate == "trigonometry") && has_dt && L > 1) {
    NEED(1);  // one invocation per input word → L outputs
    agpu_dtype to = AGPU_F32;
    if (crate == "cast") AGPU_REQUIRE(file.rfind("cast_", 0) == 0 && dtype_of(file.substr(5), &to), AGPU_ERR_UNSUPPORTED, "unknown cast shader");
    AGPU_REQUIRE(agpu_dtype_size(to) != 0, AGPU_ERR_UNSUPPORTED, "no sub-word cast to a bit-packed type");  // "cast/u8/cast_bool"
    n = min2(min2(z.inv, z.W(0)) * (uint64_t)L, out_bytes / agpu_dtype_size(to));
  } else if (crate == "routines" && (dir == "32bit" || dir == "16bit" || dir == "8bit") && file == "merge") {
    NEED(3);
    const int ml = dir == "32bit" ? 1 : dir == "16bit" ? 2 : 4;
    n = min2(min2(min3(z.inv, z.W(0), z.W(1)), z.OW()) * (uint64_t)ml, z.W(2) * 32);
  } else if (crate == "routines" && dir == "32bit" && file == "take") {
    NEED(2);  // values length comes with the binding: out-of-range indices read 0, like robust access
    return agpu_take(p, 4, inputs[0], z.W(0), static_cast<const uint32_t*>(inputs[1]), out, min3(z.inv, z.W(1), z.OW()));
  } else if (crate == "routines" && dir == "32bit" && file == "put") {
    NEED(3);  // inputs = {src, src_indexes, dst_indexes}, out = dst (binding 1, read_write)
    return agpu_put_bounded(p, 4, inputs[0], z.W(0), static_cast<const uint32_t*>(inputs[1]), out, z.OW(),
                            static_cast<const uint32_t*>(inputs[2]), min3(z.inv, z.W(1), z.W(2)));
  } else if (crate == "routines" && dir == "bool" && file == "take") {
    NEED(2);  // one invocation per OUTPUT word of 32 gathered bits
    return agpu_take_bits(p, inputs[0], z.W(0) * 32, static_cast<const uint32_t*>(inputs[1]), out,
                          min2(min2(z.inv, z.OW()) * 32, z.W(1)));
  } else if (crate == "routines" && dir == "bool" && file == "put") {
    NEED(3);  // one invocation per index pair
    return agpu_put_bits_bounded(p, inputs[0], z.W(0) * 32, static_cast<const uint32_t*>(inputs[1]), out, z.OW() * 32,
                                 static_cast<const uint32_t*>(inputs[2]), min3(z.inv, z.W(1), z.W(2)));
  } else if (crate == "routines" && ((dir == "bool" && file == "merge") || (dir == "u32" && file == "merge_null_buffer"))) {
    NEED(2);  // whole 32-bit words of bits
    uint64_t w = min3(z.inv, z.W(0), z.W(1));
    if (file == "merge") {
      NEED(3);
      w = min2(w, z.W(2));
    }
    n = min2(w, z.OW()) * 32;
  } else if (n_inputs >= 1 && (crate == "arithmetic" || crate == "logical" || crate == "compare" || crate == "math" || crate == "trigonometry") && has_dt) {
    // the word-map shape: invocation i touches word i of every binding (array.wgsl, neg.wgsl, logical.wgsl, not.wgsl,
    // 32-bit shift.wgsl, min_max.wgsl, floatunary/floatbinary.wgsl, i32 unary/binary.wgsl, f32 trigonometry.wgsl,
    // countbitones.wgsl); sub-word min_max (u16) carries L lanes per word
    uint64_t w = min3(z.inv, z.W(0), z.OW());
    const bool two = file == "array" || file == "logical" || file == "shift" || file == "min_max" || file == "floatbinary" || file == "binary";
    if (two) {
      NEED(2);
      w = min2(w, z.W(1));
    }
    n = w * (uint64_t)L;
  } else {
    known = false;
  }
#undef NEED
  if (!known) {
    agpu_set_error("no kernel for shader '%s' entry point '%s'", shader_key, entry_point);
    return AGPU_ERR_UNSUPPORTED;
  }
  return agpu_launch_by_name(p, shader_key, entry_point, inputs, n_inputs, out, n);
}
