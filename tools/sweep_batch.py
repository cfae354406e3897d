"""Throughput of batched MSM launches (development aid): python tools/sweep_batch.py LG "depth:batch,depth:batch,..." """
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import halo_accumulation_amd as h

lg = int(sys.argv[1])
combos = [tuple(int(x) for x in c.split(":")) for c in sys.argv[2].split(",")]
n = 1 << lg
ctx = h._lib.Context(urs_n=n)
ds = []
for i in range(8):
    d = torch.empty(n * 4, dtype=torch.int64, device="cuda")
    ctx.rng_scalars_dev(2 + i, n, d.data_ptr())
    ds.append(d)
ref = [ctx.msm_dev(d.data_ptr(), n).tolist() for d in ds]
parts = int(os.environ.get('PARTS', '1'))  # window shard parts - 1 of `parts` (what the last rank of a window-sharded MSM does)
for depth, batch in combos:
    ptrs = [d.data_ptr() for d in ds[:batch]]
    def run(K):
        pend = []; o = None
        for i in range(K):
            if len(pend) == depth:
                o = ctx.msm_dev_batch_end(pend.pop(0), batch)
            ctx.msm_dev_batch_begin(i % depth, ptrs, n, part=parts - 1, parts=parts)
            pend.append(i % depth)
        while pend:
            o = ctx.msm_dev_batch_end(pend.pop(0), batch)
        return o
    run(3 * depth)
    K = max(8, 64 // batch)
    torch.cuda.synchronize()
    t = time.time()
    o = run(K)
    dt = (time.time() - t) / (K * batch) * 1e3
    print("lg=%d parts=%d depth=%d batch=%d  %.3f ms per MSM share  (%.0f/s)%s" % (lg, parts, depth, batch, dt, 1e3 / dt, "  same=%s" % (o.tolist() == ref[:batch]) if parts == 1 else ""), flush=True)
