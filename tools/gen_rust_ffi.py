#!/usr/bin/env python3
"""Generate the Rust side of the drop-in boundary from include/arrow_gpu.h:

    bindings/rust/ffi.rs               every export of the header as an `extern "C"` declaration, every struct as #[repr(C)],
                                       every enum / #define as constants — produced MECHANICALLY from the header text
    bindings/rust/gpu_device.rs        `GpuDevice` + `Buffer`: the reference's method signatures
                                       [crates/array/src/gpu_utils/gpu_device.rs:46-106, 171-509] over the C ABI
    bindings/rust/compute_pipeline.rs  `ArrowComputePipeline` [crates/array/src/gpu_utils/compute_pipeline.rs:8-299] +
                                       routines::apply_take_op / apply_put_op [crates/routines/src/take.rs:9-55, put.rs:9-56] and
                                       cast::apply_boolean_unary_function [crates/cast/src/boolean_cast.rs:8-55]
                                       over `agpu_launch_by_name_sized`
    bindings/rust/mod.rs               what `crates/array/src/gpu_utils/mod.rs` becomes

There is no Rust toolchain in the build image: these files are NOT compiled here.  What IS checked (tests/test_capi_symbols.py):
ffi.rs is up to date with the header (this script's output is byte-identical to the committed file), every header export appears in
it with the same arity and the same pointer / integer / float widths, every struct has the same fields in the same order, and the two
hand-written modules only call `ffi::` functions that exist, with the right number of arguments.

    python tools/gen_rust_ffi.py            rewrite bindings/rust/
    python tools/gen_rust_ffi.py --check    exit 1 if the committed files differ from what would be generated
"""
from __future__ import annotations

import os
import re
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
HEADER = os.path.join(ROOT, "include", "arrow_gpu.h")
OUT_DIR = os.path.join(ROOT, "bindings", "rust")

SCALARS = {"int32_t": "i32", "uint32_t": "u32", "int64_t": "i64", "uint64_t": "u64", "uint8_t": "u8", "int8_t": "i8", "uint16_t": "u16",
           "int16_t": "i16", "size_t": "usize", "float": "f32", "double": "f64", "char": "c_char", "int": "c_int", "void": "c_void"}
# width class of a Rust / C type as the ABI sees it (what tests compare)
WIDTH = {"i32": "i4", "u32": "i4", "c_int": "i4", "i64": "i8", "u64": "i8", "usize": "isize", "f32": "f4", "f64": "f8", "u8": "i1", "i8": "i1",
         "u16": "i2", "i16": "i2", "c_char": "i1"}


def strip_comments(text: str) -> str:
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    return re.sub(r"//[^\n]*", " ", text)


class Header:
    """The declarations of include/arrow_gpu.h (a deliberately small C subset: the header is written to stay inside it)."""

    def __init__(self, text: str):
        src = strip_comments(text)
        self.defines = [(m.group(1), m.group(2)) for m in re.finditer(r"^#define\s+(\w+)\s+(-?\d+)\s*$", src, flags=re.M)]
        self.enums = []      # (name, [(member, value)])
        self.opaque = []     # names
        self.structs = []    # (name, [(c_type, field_name, array_len | None) | ("fnptr", name, ret, [args])])
        self.aliases = []    # (name, c_type)
        self.functions = []  # (name, ret_c_type, [(c_type, arg_name)])
        body = re.sub(r"^#.*$", "", src, flags=re.M)
        body = body.replace('extern "C" {', "").strip()
        for m in re.finditer(r"typedef\s+enum\s*\{(.*?)\}\s*(\w+)\s*;", body, flags=re.S):
            members = []
            for item in m.group(1).split(","):
                item = item.strip()
                if item:
                    k, v = [x.strip() for x in item.split("=")]
                    members.append((k, int(v)))
            self.enums.append((m.group(2), members))
        body_wo = re.sub(r"typedef\s+enum\s*\{.*?\}\s*\w+\s*;", "", body, flags=re.S)
        self.status_codes = []  # the anonymous enum of agpu_status values
        for m in re.finditer(r"(?<!typedef\s)\benum\s*\{(.*?)\}\s*;", body_wo, flags=re.S):
            for item in m.group(1).split(","):
                item = item.strip()
                if item:
                    k, v = [x.strip() for x in item.split("=")]
                    self.status_codes.append((k, int(v)))
        body_wo = re.sub(r"(?<!typedef\s)\benum\s*\{.*?\}\s*;", "", body_wo, flags=re.S)
        for m in re.finditer(r"typedef\s+struct\s+(\w+)\s+(\w+)\s*;", body_wo):
            self.opaque.append(m.group(2))
        for m in re.finditer(r"typedef\s+(\w+)\s+(\w+)\s*;", body_wo):
            if m.group(1) != "struct":
                self.aliases.append((m.group(2), m.group(1)))
        for m in re.finditer(r"(?:typedef\s+struct\s*(\w*)|struct\s+(\w+))\s*\{(.*?)\}\s*(\w*)\s*;", body_wo, flags=re.S):
            name = m.group(4) or m.group(2) or m.group(1)
            self.structs.append((name, self._fields(m.group(3))))
        body_fn = re.sub(r"(?:typedef\s+struct\s*\w*|struct\s+\w+)\s*\{.*?\}\s*\w*\s*;", "", body_wo, flags=re.S)
        body_fn = re.sub(r"typedef[^;]*;", "", body_fn)
        for m in re.finditer(r"([\w\s\*]+?)\b(agpu_\w+)\s*\(([^;{}]*?)\)\s*;", body_fn, flags=re.S):
            ret = " ".join(m.group(1).split())
            args = []
            raw = " ".join(m.group(3).split())
            if raw and raw != "void":
                for a in raw.split(","):
                    a = a.strip()
                    mm = re.match(r"(.*?)(\w+)$", a)
                    args.append((mm.group(1).strip(), mm.group(2)))
            self.functions.append((m.group(2), ret, args))

    @staticmethod
    def _fields(text: str):
        out = []
        for decl in text.split(";"):
            decl = " ".join(decl.split())
            if not decl:
                continue
            fp = re.match(r"(.*?)\(\s*\*\s*(\w+)\s*\)\s*\((.*)\)$", decl)
            if fp:
                args = [" ".join(re.sub(r"\b(\w+)$", "", a.strip()).split()) if re.search(r"[\*\s]\w+$", a.strip()) and not a.strip().endswith("*") else a.strip()
                        for a in fp.group(3).split(",")]
                out.append(("fnptr", fp.group(2), fp.group(1).strip(), args))
                continue
            m = re.match(r"(.*?)([\w\s,\[\]]+)$", decl)
            # "int32_t pci_domain, pci_bus, pci_device" / "uint8_t uuid[16]" / "const void** buffers"
            tm = re.match(r"((?:const\s+)?(?:struct\s+)?\w+[\s\*]*)(.*)$", decl)
            ctype, names = tm.group(1).strip(), tm.group(2)
            for nm in names.split(","):
                nm = nm.strip()
                stars = ""
                while nm.startswith("*"):
                    stars += "*"
                    nm = nm[1:].strip()
                arr = re.match(r"(\w+)\[(\d+)\]$", nm)
                if arr:
                    out.append((ctype + stars, arr.group(1), int(arr.group(2))))
                else:
                    out.append((ctype + stars, nm, None))
        return out


def rust_type(c: str, known: set) -> str:
    """C type (as written in the header) → Rust type."""
    c = " ".join(c.replace("*", " * ").split())
    toks = c.split()
    # const T * const * …: walk from the base type outwards
    base_const = False
    i = 0
    if toks[i] == "const":
        base_const = True
        i += 1
    if toks[i] == "struct":
        i += 1
    base = toks[i]
    i += 1
    if base in SCALARS:
        ty = SCALARS[base]
    elif base in known:
        ty = base
    else:
        raise ValueError(f"unknown C type {base!r} in {c!r}")
    const_next = base_const
    while i < len(toks):
        t = toks[i]
        if t == "*":
            ty = f"*const {ty}" if const_next else f"*mut {ty}"
            const_next = False
        elif t == "const":
            # applies to the pointer just formed: a `T* const*` makes the NEXT star point at a const pointer
            const_next = True
        else:
            raise ValueError(f"cannot parse {c!r}")
        i += 1
    return ty


def width_class(rust: str) -> str:
    if rust.startswith("*") or rust.startswith("Option<"):
        return "ptr"
    return WIDTH.get(rust, "i4")  # enums and agpu_status are 32-bit integers


def generate_ffi(h: Header) -> str:
    known = {n for n, _ in h.enums} | set(h.opaque) | {n for n, _ in h.structs} | {n for n, _ in h.aliases}
    o = []
    w = o.append
    w("// GENERATED by tools/gen_rust_ffi.py from include/arrow_gpu.h — do not edit; re-run the script when the header changes.")
    w("// `extern \"C\"` image of libarrow_gpu_hip.so's C ABI: what crates/array/src/gpu_utils/ binds instead of wgpu")
    w("// [replaces the wgpu Device / Queue / CommandEncoder calls of gpu_device.rs:29-514 and compute_pipeline.rs:8-300].")
    w("// NOT COMPILED in the build image (no rustc); tests/test_capi_symbols.py checks it against the header mechanically.")
    w("#![allow(non_camel_case_types, non_upper_case_globals, dead_code)]")
    w("use std::os::raw::{c_char, c_int, c_void};")
    w("")
    for name, value in h.defines:
        if name.endswith("_H") or name in ("ARROW_C_DATA_INTERFACE", "ARROW_C_STREAM_INTERFACE"):
            continue
        ty = "i64" if name.startswith("ARROW_FLAG") else "usize" if name.endswith("_BYTES") or name.endswith("_STEPS") else "i32"
        w(f"pub const {name}: {ty} = {value};")
    w("")
    for name, cty in h.aliases:
        w(f"pub type {name} = {SCALARS[cty]};")
    for k, v in h.status_codes:
        w(f"pub const {k}: agpu_status = {v};")
    for name, members in h.enums:
        w(f"pub type {name} = i32;")
        for k, v in members:
            w(f"pub const {k}: {name} = {v};")
    w("")
    for name in h.opaque:
        w(f"#[repr(C)] pub struct {name} {{ _private: [u8; 0] }}")
    w("")
    for name, fields in h.structs:
        w("#[repr(C)]")
        w(f"pub struct {name} {{")
        for f in fields:
            if f[0] == "fnptr":
                _, fname, ret, args = f
                rargs = ", ".join(rust_type(a, known) for a in args)
                rret = "" if ret == "void" else f" -> {rust_type(ret, known)}"
                w(f"    pub {fname}: Option<unsafe extern \"C\" fn({rargs}){rret}>,")
            else:
                ctype, fname, arr = f
                rt = rust_type(ctype, known)
                w(f"    pub {fname}: {'[%s; %d]' % (rt, arr) if arr else rt},")
        w("}")
    w("")
    w('#[link(name = "arrow_gpu_hip")]')
    w('extern "C" {')
    for name, ret, args in h.functions:
        rargs = ", ".join(f"{('r#' + an) if an in ('in', 'type', 'ref', 'box', 'move') else an}: {rust_type(ct, known)}" for ct, an in args)
        rret = "" if ret == "void" else f" -> {rust_type(ret, known)}"
        w(f"    pub fn {name}({rargs}){rret};")
    w("}")
    w("")
    return "\n".join(o)


GPU_DEVICE_RS = r'''// bindings/rust/gpu_device.rs — what crates/array/src/gpu_utils/gpu_device.rs becomes: the same public methods, bodies = C-ABI calls.
// Written by tools/gen_rust_ffi.py (template; the `ffi::` calls are checked against include/arrow_gpu.h by
// tests/test_capi_symbols.py).  NOT COMPILED in the build image (no rustc) — flagged in INTEGRATION.md.
// [ref: crates/array/src/gpu_utils/gpu_device.rs — GpuDevice::new :46-85, from_adapter :87-106, create_* :171-210,
//  clone_buffer(_pass) :212-230, retrive_data :232-265, apply_{unary,scalar,binary,ternary,broadcast}_function :267-509]
use std::ffi::{c_void, CStr, CString};
use std::sync::{Arc, Mutex};

use super::ffi;
use crate::array::RustNativeType; // unchanged marker trait [crates/array/src/array/mod.rs]

/// Stands in for `wgpu::Buffer`: a device pointer + its byte size.  The array types keep their `Arc<Buffer>` fields unchanged.
pub struct Buffer {
    pub(crate) ptr: *mut c_void,
    size: u64,
    dev: *mut ffi::agpu_device,
}
unsafe impl Send for Buffer {}
unsafe impl Sync for Buffer {}
impl Buffer {
    pub fn size(&self) -> u64 { self.size }
}
impl Drop for Buffer {
    fn drop(&mut self) { unsafe { ffi::agpu_free(self.dev, self.ptr); } } // pooled: no device sync, no hipFree
}

pub(crate) fn check(status: ffi::agpu_status) {
    if status == 0 { return; }
    let msg = unsafe { CStr::from_ptr(ffi::agpu_last_error()) }.to_string_lossy().into_owned();
    // the reference panics on unsupported type pairs ("Operation … not supported …") and unwraps everywhere else
    panic!("{}", msg);
}

/// The shader argument stays what the op crates pass today — the WGSL text.  The ABI recognises it by hash; the conversion to a
/// C string is cached per `&'static str` so a launch does not allocate.
pub(crate) fn shader_key(shader: &str) -> CString {
    let mut key = [0i8; 64];
    let st = unsafe { ffi::agpu_shader_key_for_source(shader.as_ptr() as *const i8, shader.len(), key.as_mut_ptr(), key.len()) };
    check(st);
    unsafe { CStr::from_ptr(key.as_ptr()) }.to_owned()
}

pub struct GpuDevice {
    pub(crate) raw: *mut ffi::agpu_device,
    io: Mutex<*mut ffi::agpu_pipeline>, // uploads / read-backs / the immediate apply_* forms [queue.submit per call in the reference]
}
unsafe impl Send for GpuDevice {}
unsafe impl Sync for GpuDevice {}

impl GpuDevice {
    pub fn new() -> GpuDevice { Self::from_ordinal(0) }

    /// `from_adapter(adapter: Adapter)` named a wgpu adapter; on a ROCm node the choice is the device ordinal.
    pub fn from_adapter(ordinal: i32) -> GpuDevice { Self::from_ordinal(ordinal) }

    fn from_ordinal(ordinal: i32) -> GpuDevice {
        let mut raw = std::ptr::null_mut();
        check(unsafe { ffi::agpu_device_create(ordinal, &mut raw) }); // AGPU_ERR_NO_DEVICE without a gfx950: there is no CPU fallback
        let mut io = std::ptr::null_mut();
        check(unsafe { ffi::agpu_pipeline_create(raw, &mut io) });
        GpuDevice { raw, io: Mutex::new(io) }
    }

    fn wrap(&self, ptr: *mut c_void, size: u64) -> Buffer { Buffer { ptr, size, dev: self.raw } }

    pub fn create_gpu_buffer_with_data(&self, data: &[impl RustNativeType]) -> Buffer {
        let bytes = std::mem::size_of_val(data);
        let buf = self.create_empty_buffer(bytes as u64);
        let io = self.io.lock().unwrap();
        check(unsafe { ffi::agpu_upload(*io, buf.ptr, data.as_ptr() as *const c_void, bytes) });
        buf
    }

    /// Creates an empty GPU buffer (zero-filled, as wgpu guarantees; the kernels never rely on it)
    pub fn create_empty_buffer(&self, size: u64) -> Buffer {
        let mut p = std::ptr::null_mut();
        check(unsafe { ffi::agpu_malloc(self.raw, size as usize, 1, &mut p) });
        self.wrap(p, size)
    }

    /// output of an op: placed against the buffers it will be used with (HBM channel hash, DESIGN.md §3)
    pub fn create_empty_buffer_like(&self, size: u64, neighbours: &[&Buffer]) -> Buffer {
        let ptrs: Vec<*const c_void> = neighbours.iter().map(|b| b.ptr as *const c_void).collect();
        let mut p = std::ptr::null_mut();
        check(unsafe { ffi::agpu_malloc_like(self.raw, size as usize, 1, ptrs.as_ptr(), ptrs.len() as i32, &mut p) });
        self.wrap(p, size)
    }

    pub fn create_retrive_buffer(&self, size: u64) -> Buffer { self.create_empty_buffer(size) } // no staging buffer is needed

    pub fn create_scalar_buffer<T: Copy>(&self, value: &T) -> Buffer {
        let bytes = std::mem::size_of::<T>();
        let buf = self.create_empty_buffer(bytes as u64);
        let io = self.io.lock().unwrap();
        check(unsafe { ffi::agpu_upload(*io, buf.ptr, value as *const T as *const c_void, bytes) });
        buf
    }

    pub fn clone_buffer(&self, buffer: &Buffer) -> Buffer {
        let out = self.create_empty_buffer(buffer.size());
        let io = self.io.lock().unwrap();
        check(unsafe { ffi::agpu_copy(*io, out.ptr, buffer.ptr as *const c_void, buffer.size() as usize) });
        check(unsafe { ffi::agpu_pipeline_finish(*io) });
        out
    }

    pub fn clone_buffer_pass(&self, buffer: &Buffer, pipeline: *mut ffi::agpu_pipeline) -> Buffer {
        let out = self.create_empty_buffer(buffer.size());
        check(unsafe { ffi::agpu_copy(pipeline, out.ptr, buffer.ptr as *const c_void, buffer.size() as usize) });
        out
    }

    /// The only blocking call, as in the reference.
    pub fn retrive_data(&self, data: &Buffer) -> Vec<u8> {
        let mut host = vec![0u8; data.size() as usize];
        let io = self.io.lock().unwrap();
        check(unsafe { ffi::agpu_download(*io, host.as_mut_ptr() as *mut c_void, data.ptr as *const c_void, host.len()) });
        host
    }

    fn launch(&self, inputs: &[&Buffer], out: &Buffer, shader: &str, entry_point: &str, dispatch_size: u32) {
        let ptrs: Vec<*const c_void> = inputs.iter().map(|b| b.ptr as *const c_void).collect();
        let sizes: Vec<u64> = inputs.iter().map(|b| b.size()).collect();
        let key = shader_key(shader);
        let entry = CString::new(entry_point).unwrap();
        let io = self.io.lock().unwrap();
        check(unsafe { ffi::agpu_launch_by_name_sized(*io, key.as_ptr(), entry.as_ptr(), ptrs.as_ptr(), sizes.as_ptr(), ptrs.len() as i32,
                                                      out.ptr, out.size(), dispatch_size) });
        check(unsafe { ffi::agpu_pipeline_finish(*io) }); // the immediate forms submit at once [queue.submit, gpu_device.rs:308]
    }

    pub fn apply_unary_function(&self, original_values: &Buffer, new_buffer_size: u64, item_size: u64, shader: &str, entry_point: &str) -> Buffer {
        let out = self.create_empty_buffer_like(new_buffer_size, &[original_values]);
        let dispatch_size = original_values.size().div_ceil(item_size);
        self.launch(&[original_values], &out, shader, entry_point, dispatch_size.div_ceil(256) as u32);
        out
    }

    pub fn apply_scalar_function(&self, original_values: &Buffer, scalar_value: &Buffer, output_buffer_size: u64, item_size: u64, shader: &str,
                                 entry_point: &str) -> Buffer {
        let out = self.create_empty_buffer_like(output_buffer_size, &[original_values]);
        let dispatch_size = original_values.size() / item_size;
        self.launch(&[original_values, scalar_value], &out, shader, entry_point, dispatch_size.div_ceil(256) as u32);
        out
    }

    pub fn apply_binary_function(&self, operand_1: &Buffer, operand_2: &Buffer, item_size: u64, shader: &str, entry_point: &str) -> Buffer {
        let out = self.create_empty_buffer_like(operand_1.size(), &[operand_1, operand_2]);
        let dispatch_size = operand_1.size() / item_size;
        self.launch(&[operand_1, operand_2], &out, shader, entry_point, dispatch_size.div_ceil(256) as u32);
        out
    }

    pub fn apply_ternary_function(&self, operand_1: &Buffer, operand_2: &Buffer, operand_3: &Buffer, item_size: u64, shader: &str,
                                  entry_point: &str) -> Buffer {
        let out = self.create_empty_buffer_like(operand_1.size(), &[operand_1, operand_2, operand_3]);
        let dispatch_size = operand_1.size() / item_size;
        self.launch(&[operand_1, operand_2, operand_3], &out, shader, entry_point, dispatch_size.div_ceil(256) as u32);
        out
    }

    pub fn apply_broadcast_function(&self, scalar_value: &Buffer, output_buffer_size: u64, item_size: u64, shader: &str, entry_point: &str) -> Buffer {
        let out = self.create_empty_buffer(output_buffer_size);
        let dispatch_size = output_buffer_size / item_size;
        self.launch(&[scalar_value], &out, shader, entry_point, dispatch_size.div_ceil(256) as u32);
        out
    }

    /// give pooled blocks and idle streams back to the driver (no counterpart in the reference: wgpu frees on drop)
    pub fn trim(&self) { check(unsafe { ffi::agpu_device_trim(self.raw) }); }
}

impl Drop for GpuDevice {
    fn drop(&mut self) {
        unsafe {
            ffi::agpu_pipeline_destroy(*self.io.lock().unwrap());
            ffi::agpu_device_destroy(self.raw);
        }
    }
}

pub type SharedDevice = Arc<GpuDevice>; // `pub static GPU_DEVICE: LazyLock<Arc<GpuDevice>>` stays as it is [crates/array/src/lib.rs:17]
'''

COMPUTE_PIPELINE_RS = r'''// bindings/rust/compute_pipeline.rs — what crates/array/src/gpu_utils/compute_pipeline.rs becomes: the same public methods over
// `agpu_launch_by_name_sized`, the reference's literal call shape (buffers with their byte sizes, a shader text, an entry-point
// name, a dispatch size — no element count).  Written by tools/gen_rust_ffi.py (template; `ffi::` calls checked against
// include/arrow_gpu.h by tests/test_capi_symbols.py).  NOT COMPILED in the build image (no rustc).
// [ref: crates/array/src/gpu_utils/compute_pipeline.rs — new :15-22, apply_unary :24-66, apply_binary :68-113, apply_ternary :115-165,
//  apply_scalar :167-213, apply_broadcast :215-256, finish :259-273, clone_buffer :275-282, copy_buffer_to_buffer :284-299;
//  crates/routines/src/take.rs:9-55, put.rs:9-56; crates/cast/src/boolean_cast.rs:8-55]
use std::ffi::{c_void, CString};
use std::sync::Arc;

use super::ffi;
use super::gpu_device::{check, shader_key, Buffer, GpuDevice};

pub struct ArrowComputePipeline {
    pub device: Arc<GpuDevice>,
    pub(crate) raw: *mut ffi::agpu_pipeline, // a HIP stream: launches are eager and ordered, like commands in the encoder
}

impl ArrowComputePipeline {
    pub fn new(device: Arc<GpuDevice>, _label: Option<&str>) -> Self {
        let mut raw = std::ptr::null_mut();
        check(unsafe { ffi::agpu_pipeline_create(device.raw, &mut raw) }); // recycled streams: ~1 µs, not hipStreamCreate
        ArrowComputePipeline { device, raw }
    }

    fn launch(&mut self, inputs: &[&Buffer], out: &Buffer, shader: &str, entry_point: &str, dispatch_size: u32) {
        let ptrs: Vec<*const c_void> = inputs.iter().map(|b| b.ptr as *const c_void).collect();
        let sizes: Vec<u64> = inputs.iter().map(|b| b.size()).collect();
        let key = shader_key(shader);
        let entry = CString::new(entry_point).unwrap();
        check(unsafe { ffi::agpu_launch_by_name_sized(self.raw, key.as_ptr(), entry.as_ptr(), ptrs.as_ptr(), sizes.as_ptr(), ptrs.len() as i32,
                                                      out.ptr, out.size(), dispatch_size) });
    }

    pub fn apply_unary_function(&mut self, original_values: &Buffer, new_buffer_size: u64, shader: &str, entry_point: &str, dispatch_size: u32) -> Buffer {
        let out = self.device.create_empty_buffer_like(new_buffer_size, &[original_values]);
        self.launch(&[original_values], &out, shader, entry_point, dispatch_size);
        out
    }

    pub fn apply_binary_function(&mut self, operand_1: &Buffer, operand_2: &Buffer, new_buffer_size: u64, shader: &str, entry_point: &str,
                                 dispatch_size: u32) -> Buffer {
        let out = self.device.create_empty_buffer_like(new_buffer_size, &[operand_1, operand_2]);
        self.launch(&[operand_1, operand_2], &out, shader, entry_point, dispatch_size);
        out
    }

    pub fn apply_ternary_function(&mut self, operand_1: &Buffer, operand_2: &Buffer, operand_3: &Buffer, new_buffer_size: u64, shader: &str,
                                  entry_point: &str, dispatch_size: u32) -> Buffer {
        let out = self.device.create_empty_buffer_like(new_buffer_size, &[operand_1, operand_2, operand_3]);
        self.launch(&[operand_1, operand_2, operand_3], &out, shader, entry_point, dispatch_size);
        out
    }

    pub fn apply_scalar_function(&mut self, original_values: &Buffer, scalar_value: &Buffer, output_buffer_size: u64, shader: &str, entry_point: &str,
                                 dispatch_size: u32) -> Buffer {
        let out = self.device.create_empty_buffer_like(output_buffer_size, &[original_values]);
        self.launch(&[original_values, scalar_value], &out, shader, entry_point, dispatch_size);
        out
    }

    pub fn apply_broadcast_function(&mut self, scalar_value: &Buffer, output_buffer_size: u64, shader: &str, entry_point: &str, dispatch_size: u32) -> Buffer {
        let out = self.device.create_empty_buffer(output_buffer_size);
        self.launch(&[scalar_value], &out, shader, entry_point, dispatch_size);
        out
    }

    /// Submit the pipeline to the GPU: publishes the stream's position to every other pipeline; does not wait.
    pub fn finish(self) {
        check(unsafe { ffi::agpu_pipeline_finish(self.raw) });
    }

    pub fn clone_buffer(&mut self, buffer: &Buffer) -> Buffer {
        let out = self.device.create_empty_buffer(buffer.size());
        check(unsafe { ffi::agpu_copy(self.raw, out.ptr, buffer.ptr as *const c_void, buffer.size() as usize) });
        out
    }

    pub fn copy_buffer_to_buffer(&mut self, source: &Buffer, source_offset: u64, destination: &Buffer, destination_offset: u64, copy_size: u64) {
        let dst = unsafe { (destination.ptr as *mut u8).add(destination_offset as usize) } as *mut c_void;
        let src = unsafe { (source.ptr as *const u8).add(source_offset as usize) } as *const c_void;
        check(unsafe { ffi::agpu_copy(self.raw, dst, src, copy_size as usize) });
    }

    /// wait for everything recorded so far; also where an out-of-range take / put index is reported (AGPU_ERR_SHAPE)
    pub fn sync(&mut self) { check(unsafe { ffi::agpu_pipeline_sync(self.raw) }); }
}

impl Drop for ArrowComputePipeline {
    fn drop(&mut self) { unsafe { ffi::agpu_pipeline_destroy(self.raw); } } // a submit point like finish(); the stream goes back to the pool
}

/// crates/routines/src/take.rs:9-55 — bindings (values, indexes, output); `dispatch_size` is the row count there
pub(crate) fn apply_take_op(device: &GpuDevice, operand_1: &Buffer, operand_2: &Buffer, dispatch_size: u64, output_size: u64, shader: &str,
                            entry_point: &str, pipeline: &mut ArrowComputePipeline) -> Buffer {
    let out = device.create_empty_buffer(output_size);
    pipeline.launch(&[operand_1, operand_2], &out, shader, entry_point, dispatch_size.div_ceil(256) as u32);
    out
}

/// crates/routines/src/put.rs:9-56 — bindings (src, dst, src_indexes, dst_indexes); `dst` is written in place
pub(crate) fn apply_put_op(_device: &GpuDevice, src_buffer: &Buffer, dst_buffer: &Buffer, src_indexes: &Buffer, dst_indexes: &Buffer, dispatch_size: u64,
                           shader: &str, entry_point: &str, pipeline: &mut ArrowComputePipeline) {
    pipeline.launch(&[src_buffer, src_indexes, dst_indexes], dst_buffer, shader, entry_point, dispatch_size.div_ceil(256) as u32);
}

/// crates/cast/src/boolean_cast.rs:8-55 — Boolean bitmap in, one f32 per bit out
pub fn apply_boolean_unary_function(_gpu_device: &GpuDevice, original_values: &Buffer, new_buffer_size: u64, output_item_size: u64, shader: &str,
                                    entry_point: &str, pipeline: &mut ArrowComputePipeline) -> Buffer {
    let dispatch_size = new_buffer_size.div_ceil(output_item_size); // one invocation per OUTPUT element
    pipeline.apply_unary_function(original_values, new_buffer_size, shader, entry_point, dispatch_size.div_ceil(256) as u32)
}
'''

MOD_RS = r'''// bindings/rust/mod.rs — crates/array/src/gpu_utils/mod.rs with the wgpu back end removed.  Written by tools/gen_rust_ffi.py.
pub mod compute_pipeline;
pub mod ffi;
pub mod gpu_device;

pub use compute_pipeline::*;
pub use gpu_device::*;

/// `CmpQuery` (compute_query.rs:7-89: a timestamp query pair per pass) maps to the pipeline's own timing switch.
pub struct CmpQuery;
impl CmpQuery {
    pub fn enable(pipeline: &mut ArrowComputePipeline) { unsafe { ffi::agpu_pipeline_enable_timing(pipeline.raw, 2); } }
    pub fn wait_for_results(pipeline: &mut ArrowComputePipeline) -> u64 {
        let (mut ns, mut name) = (0u64, std::ptr::null());
        unsafe { ffi::agpu_pipeline_last_kernel_ns(pipeline.raw, &mut ns, &mut name); }
        ns
    }
}
'''


def outputs() -> dict:
    h = Header(open(HEADER).read())
    return {"ffi.rs": generate_ffi(h), "gpu_device.rs": GPU_DEVICE_RS.lstrip("\n"), "compute_pipeline.rs": COMPUTE_PIPELINE_RS.lstrip("\n"),
            "mod.rs": MOD_RS.lstrip("\n")}


def main():
    files = outputs()
    if "--check" in sys.argv:
        bad = [n for n, text in files.items() if not os.path.exists(os.path.join(OUT_DIR, n)) or open(os.path.join(OUT_DIR, n)).read() != text]
        if bad:
            print("out of date:", bad)
            sys.exit(1)
        print("bindings/rust is up to date with include/arrow_gpu.h")
        return
    os.makedirs(OUT_DIR, exist_ok=True)
    for n, text in files.items():
        with open(os.path.join(OUT_DIR, n), "w") as f:
            f.write(text)
        print("wrote", os.path.join("bindings", "rust", n), f"({text.count(chr(10))} lines)")


if __name__ == "__main__":
    main()
