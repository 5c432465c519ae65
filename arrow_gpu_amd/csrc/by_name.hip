// by_name.hip — agpu_launch_by_name: the reference's kernel identity (WGSL file, @compute entry point) → typed ABI.
//
// In the reference a kernel is identified by (shader text, entry-point name): that pair is the pipeline-cache key of
// GpuDevice::create_compute_pipeline (crates/array/src/gpu_utils/gpu_device.rs:145-168) and what every op macro passes
// to apply_{unary,binary,scalar,ternary,broadcast}_function.  A Rust shim that keeps those call sites can forward the
// shader's path ("<crate>/<dir>/<file>", e.g. "arithmetic/f32/array") and the entry point here unchanged.
#include <string>

#include "common.hpp"

agpu_status agpu_bitmap_andnot_internal(agpu_pipeline* p, const void* a, const void* b, void* out, uint64_t n_bits);

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

extern "C" agpu_status agpu_launch_by_name(agpu_pipeline* p, const char* shader_key, const char* entry_point,
                                           const void* const* inputs, int32_t n_inputs, void* out, uint64_t n) {
  AGPU_REQUIRE(p && shader_key && entry_point && (n_inputs == 0 || inputs), AGPU_ERR_ARG, "null argument");
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
