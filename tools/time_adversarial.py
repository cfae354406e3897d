import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import halo_accumulation_amd as h
n = 1 << 20
ctx = h._lib.Context(urs_n=n)
_d = torch.empty(8, dtype=torch.int64, device="cuda"); ctx.rng_scalars_dev(99, 2, _d.data_ptr())
k = _d.cpu().numpy().view(np.uint64).reshape(2, 4)
ONE_MONT = np.array([0x5b2b3e9cfffffffd, 0x992c350be3420567, 0xffffffffffffffff, 0x3fffffffffffffff], dtype=np.uint64)  # 2^256 mod r
for name, arr in (("all-same", np.tile(k[0], (n, 1))), ("all-one", np.tile(ONE_MONT, (n, 1))), ("all-zero", np.zeros((n, 4), dtype=np.uint64))):
    d = torch.from_numpy(np.ascontiguousarray(arr).view(np.int64)).cuda()
    for mode in (0, 1):
        ctx.set_sort_mode(mode)
        ctx.msm_dev(d.data_ptr(), n)
        t = time.time()
        for _ in range(5): ctx.msm_dev(d.data_ptr(), n)
        print("%-9s sort_mode=%d  %.3f ms per MSM" % (name, mode, (time.time() - t) / 5 * 1e3), flush=True)
