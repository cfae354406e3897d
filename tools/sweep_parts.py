"""Throughput of one window shard of an MSM (what one rank of a window-sharded MSM does): python tools/sweep_parts.py LG PARTS[,PARTS...]"""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import halo_accumulation_amd as h

lg = int(sys.argv[1])
n = 1 << lg
ctx = h._lib.Context(urs_n=n)
d = torch.empty(n * 4, dtype=torch.int64, device="cuda")
ctx.rng_scalars_dev(2, n, d.data_ptr())
kls = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [0]
for parts in [int(x) for x in sys.argv[2].split(",")]:
  for kl in kls:
    ctx.set_task_len(kl)
    for part in sorted({parts - 1}):
        for depth in (4,):
            def run(K):
                pend = []
                for i in range(K):
                    if len(pend) == depth:
                        ctx.msm_dev_end(pend.pop(0))
                    ctx.msm_dev_begin(i % depth, d.data_ptr(), n, part=part, parts=parts)
                    pend.append(i % depth)
                while pend:
                    ctx.msm_dev_end(pend.pop(0))
            run(3 * depth)
            torch.cuda.synchronize()
            K = 64
            t = time.time()
            run(K)
            dt = (time.time() - t) / K * 1e3
            print("lg=%d parts=%d part=%d depth=%d task_len=%d  %.3f ms per shard  (%.0f/s)" % (lg, parts, part, depth, kl, dt, 1e3 / dt), flush=True)
