"""Per-kernel time of one window-shard launch (batch MSMs x 1/parts of the windows), one launch in flight (development aid)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import halo_accumulation_amd as h
lg, parts, batch = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
n = 1 << lg
ctx = h._lib.Context(urs_n=n)
ds = []
for i in range(batch):
    d = torch.empty(n * 4, dtype=torch.int64, device="cuda"); ctx.rng_scalars_dev(2 + i, n, d.data_ptr()); ds.append(d)
ptrs = [d.data_ptr() for d in ds]
for _ in range(3):
    ctx.msm_dev_batch_begin(0, ptrs, n, part=parts - 1, parts=parts); ctx.msm_dev_batch_end(0, batch)
ctx.prof_enable(True); ctx.prof_reset()
K = 5
for _ in range(K):
    ctx.msm_dev_batch_begin(0, ptrs, n, part=parts - 1, parts=parts); ctx.msm_dev_batch_end(0, batch)
tot = 0
for k, (ms, cnt) in sorted(ctx.prof().items(), key=lambda kv: -kv[1][0]):
    print("   %-22s %8.3f ms per launch (x%d)" % (k, ms / K, cnt // K)); tot += ms / K
print("sum %.3f ms per launch = %.3f ms per MSM share" % (tot, tot / batch))
