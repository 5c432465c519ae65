#!/usr/bin/env python3
"""The reference's crates/arrow/examples/simple.rs, line for line, on the MI355X-native library.

  run_basic_add            [simple.rs:10-41]   typed op, default API (one pipeline per op), and the *_dyn form
  run_compute_pipeline_ops [simple.rs:45-72]   two ops recorded on ONE pipeline, one finish()
  + the same chain as one fused kernel (FusedChain) and through a fusing pipeline.

Needs an MI355X (gfx950): there is no CPU fallback.   python examples/simple.py
"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import arrow_gpu_amd as ag  # noqa: E402


def run_basic_add(device):
    float_values = [float(i) for i in range(10)]
    gpu_float_array = ag.Float32ArrayGPU.from_slice(float_values, device)
    gpu_float_array_scalar = ag.Float32ArrayGPU.from_slice([20.0], device)
    add_scalar_result = gpu_float_array.add_scalar(gpu_float_array_scalar)
    assert add_scalar_result.values() == [v + 20.0 for v in float_values]
    dyn_result = ag.add_scalar_dyn(gpu_float_array, gpu_float_array_scalar)  # enum dispatch, like ArrowArrayGPU
    assert isinstance(dyn_result, ag.Float32ArrayGPU)
    assert dyn_result.values() == [v + 20.0 for v in float_values]
    print("basic add:", add_scalar_result.values())


def run_compute_pipeline_ops(device):
    pipeline = ag.ArrowComputePipeline(device, "example")
    float_values = [float(i) for i in range(100)]
    lhs = ag.Float32ArrayGPU.from_slice(float_values, device)
    rhs = ag.Float32ArrayGPU.from_slice([20.0], device)
    new_gpu_array = ag.add_scalar_op_dyn(lhs, rhs, pipeline)
    new_gpu_array = ag.mul_scalar_op_dyn(new_gpu_array, rhs, pipeline)
    pipeline.finish()
    expected = [(v + 20.0) * 20.0 for v in float_values]
    assert new_gpu_array.values() == expected

    # the same chain as ONE kernel: explicitly …
    fused = ag.FusedChain(lhs).add_scalar(rhs).mul_scalar(rhs).finish()
    assert fused.values() == expected
    # … or by letting the pipeline fuse what it recorded when finish() is called
    fusing = ag.ArrowComputePipeline(device, "example", fuse=True)
    result = ag.mul_scalar_op_dyn(ag.add_scalar_op_dyn(lhs, rhs, fusing), rhs, fusing)
    fusing.finish()
    assert result.values() == expected and fusing.stats["kernels"] == 1
    print("pipeline ops:", expected[:4], "…  (fusing pipeline:", fusing.stats, ")")


if __name__ == "__main__":
    dev = ag.GPU_DEVICE()
    run_basic_add(dev)
    run_compute_pipeline_ops(dev)
    print("OK on", dev)
