"""arrow_gpu_amd — MI355X-native Arrow compute kernels (HIP, gfx950) behind psvri/arrow-gpu's API surface."""
