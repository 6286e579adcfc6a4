#!/usr/bin/env python3
"""Times the intersection batch at reduced occupancy (RT_TUNE_LDS_EXTRA adds LDS per block)."""
import os
import subprocess
import sys

code = r'''
import sys
sys.path.insert(0, ".")
import torch, ray_tracer_amd as rta
bvh = rta.init_environment()
n = 1 << 23
dev = torch.device("cuda:0")
rays = torch.from_numpy(rta.make_ray_batch(bvh, n)).to(dev)
ids = torch.empty(n, dtype=torch.int32, device=dev); dist = torch.empty(n, dtype=torch.float64, device=dev)
lib = rta._lib.lib()
def run():
    rta._lib.check(lib.rt_intersect_batch_device(bvh.scene(), rays.data_ptr(), n, 0, ids.data_ptr(), dist.data_ptr(), None, None, None))
run(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(3): run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 3
print("%.3f ms (%.0f Mrays/s)" % (ms, n / ms / 1e3))
'''
for extra in [int(x) for x in sys.argv[1:]] or [0, 4096, 18000, 50000, 120000]:
    env = dict(os.environ, RT_TUNE_LDS_EXTRA=str(extra))
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    lds = 18 * 256 * 8 + extra
    print("extra LDS %6d -> %6d B/block, blocks/CU by LDS %d:" % (extra, lds, 163840 // lds),
          out.stdout.strip() or out.stderr.strip()[-300:], flush=True)
