// bindings/rust/compute_pipeline.rs — what crates/array/src/gpu_utils/compute_pipeline.rs becomes: the same public methods over
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
