import sys, time, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/oracle")
import numpy as np, torch
import halo_accumulation_amd as h, orc
n = 1 << 20
ctx = h._lib.Context(urs_n=n)
k, _ = orc.rng_scalars(99, 1)
for name, arr in (("all-same", np.tile(k[0], (n, 1))), ("all-one", np.tile(orc.fr_to_mont(1), (n, 1))), ("all-zero", np.zeros((n, 4), dtype=np.uint64))):
    d = torch.from_numpy(np.ascontiguousarray(arr).view(np.int64)).cuda()
    for mode in (0, 1):
        ctx.set_sort_mode(mode)
        ctx.msm_dev(d.data_ptr(), n)
        t = time.time()
        for _ in range(5): ctx.msm_dev(d.data_ptr(), n)
        print("%-9s sort_mode=%d  %.3f ms per MSM" % (name, mode, (time.time() - t) / 5 * 1e3), flush=True)
