"""Host-side cost of msm_dev_begin / msm_dev_end (development aid): the device is idle when end() is called."""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import halo_accumulation_amd as h

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 17
n = 1 << lg
ctx = h._lib.Context(urs_n=n)
d = torch.empty(n * 4, dtype=torch.int64, device="cuda")
ctx.rng_scalars_dev(2, n, d.data_ptr())
for _ in range(4):
    ctx.msm_dev(d.data_ptr(), n)
K = 50
tb = te = 0.0
for _ in range(K):
    t = time.perf_counter()
    ctx.msm_dev_begin(0, d.data_ptr(), n)
    tb += time.perf_counter() - t
    torch.cuda.synchronize()
    time.sleep(0.003)
    t = time.perf_counter()
    ctx.msm_dev_end(0)
    te += time.perf_counter() - t
print("lg=%d  begin %.1f us   end (device idle) %.1f us" % (lg, tb / K * 1e6, te / K * 1e6))
for depth in (4, 8):
    pass
