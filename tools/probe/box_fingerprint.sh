#!/bin/bash
# DEV TOOL (GPU box): what distinguishes the boxes on which two tiles per block WIN (R5.4) from those on which they lose?  Memory / compute
# partition modes, clocks, firmware, driver parameters — next to the box's class (cast u8→f32 at 1e9 rows, one vs two tiles per block).
#   bash tools/probe/box_fingerprint.sh >> gpurun_out/r05_boxes.txt
echo "=== box $(date -u +%FT%TZ) host=$(hostname) boot_id=$(cat /proc/sys/kernel/random/boot_id 2>/dev/null)"
for f in current_memory_partition current_compute_partition available_memory_partition available_compute_partition mem_info_vram_total mem_info_vram_used \
         mem_info_vram_vendor vbios_version unique_id pp_dpm_mclk pp_dpm_sclk pp_dpm_fclk pp_dpm_socclk gpu_busy_percent mem_busy_percent power_dpm_force_performance_level \
         xgmi_device_id ras/features; do
  for d in /sys/class/drm/card*/device; do
    [ -r "$d/$f" ] && echo "$d/$f: $(tr '\n' '|' < "$d/$f" 2>/dev/null | head -c 400)"
  done
done
cat /sys/module/amdgpu/version 2>/dev/null | sed 's/^/amdgpu version: /'
for prm in noretry vm_fragment_size vm_block_size mtype_local sched_policy hws_max_conc_proc; do
  [ -r /sys/module/amdgpu/parameters/$prm ] && echo "amdgpu.$prm=$(cat /sys/module/amdgpu/parameters/$prm)"
done
cat /sys/kernel/mm/transparent_hugepage/enabled 2>/dev/null | sed 's/^/thp: /'
uname -r
timeout 20 rocm-smi --showmemorypartition --showcomputepartition --showclocks --showmeminfo vram --showperflevel 2>/dev/null | grep -v '^$' | head -40
timeout 20 /opt/rocm/bin/amd-smi static --partition --vram --limit 2>/dev/null | head -60
python3 - <<'PY'
import ctypes as C, os, sys
sys.path.insert(0, os.getcwd())
from arrow_gpu_amd import _capi as capi
from arrow_gpu_amd.gpu_utils import ArrowComputePipeline, CmpQuery, GpuDevice
n = 1_000_000_000
dev = GpuDevice(0); p = ArrowComputePipeline(dev, "fp"); q = CmpQuery(dev); h = p._handle
from arrow_gpu_amd.sharding import Peer
me = Peer(); capi.call("agpu_device_identity", dev._handle, C.byref(me)); print("our device:", me.as_dict())
u8, g = dev.create_table_buffers([n, 4 * n])
capi.call("agpu_synth_u8", h, C.c_void_p(u8.ptr), n, 6, 0); p.sync()
def med(k):
    p.set_tuning("tiles", k)
    f = lambda: capi.call("agpu_cast", h, capi.U8, capi.F32, C.c_void_p(u8.ptr), C.c_void_p(g.ptr), n)
    f(); f(); ts = []
    for _ in range(7):
        q.begin(p); f(); q.end(p); ts.append(q.wait_for_results())
    return 5.0 * n / sorted(ts)[3] / 1e6 / 8000
a, b, c, d = med(1), med(2), med(1), med(2)
print(f"class: cast u8->f32 one tile {a:.3f} {c:.3f}  two tiles {b:.3f} {d:.3f}  -> {'TWO TILES WIN' if min(b, d) > max(a, c) * 1.01 else 'two tiles lose' if max(b, d) < min(a, c) * 0.99 else 'tie'}")
PY
