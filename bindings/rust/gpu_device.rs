// bindings/rust/gpu_device.rs — what crates/array/src/gpu_utils/gpu_device.rs becomes: the same public methods, bodies = C-ABI calls.
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

    /// output of an op: placed against the buffers it will be used with (HBM channel hash)
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
