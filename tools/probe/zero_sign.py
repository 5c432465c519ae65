import numpy as np, sys
sys.path.insert(0, ".")
import arrow_gpu_amd as ag
x = np.array([-0.0, 0.0, -1e-40, 1e-40, -1e-30], np.float32)
a = ag.Float32ArrayGPU.from_slice(x, ag.GPU_DEVICE())
for name in ("sin", "cos", "sinh"):
    r = getattr(a, name)().raw_values()
    print(name, r, np.signbit(r))
