// bindings/rust/mod.rs — crates/array/src/gpu_utils/mod.rs with the wgpu back end removed.  Written by tools/gen_rust_ffi.py.
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
