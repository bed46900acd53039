"""Developer probe: the map-update chain of the bench (100 032-ray HIMM batches in rotation + fused compose on the 4096^2
map) by itself, nothing else on the GPU -- run under `rocprofv3 --kernel-trace --stats` for the per-kernel times."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ros_navigation_amd as R

n = 4096
L = n * 0.05
e = R.Engine(L, L, 0.05)
e.upload(R.capi.LAYER_LASER, R.synth.obstacles_rect(n, n, density=0.30, seed=2))
e.compose_master(1)
sets = [R.synth.rays(64, 1563, L, L, seed=4 + k) for k in range(4)]
d = [torch.from_numpy(np.frombuffer(r.tobytes(), dtype=np.uint8).copy()).cuda() for r in sets]
for k in range(8):
    e.update_map_device(d[k % 4].data_ptr(), len(sets[k % 4]), compose_mode=0)
e.synchronize()
t0 = time.perf_counter()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
for k in range(reps):
    e.update_map_device(d[k % 4].data_ptr(), len(sets[k % 4]), compose_mode=0)
e.synchronize()
print("map update chain alone: %.3f ms per batch" % ((time.perf_counter() - t0) / reps * 1e3))
e.close()
