"""Batched MSMs over a small key (a rank's index shard), table plan against general: python tools/time_table_batch.py LG BATCH[,BATCH..]"""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import halo_accumulation_amd as h
import torch
lg = int(sys.argv[1]); n = 1 << lg
ctx = h._lib.Context(urs_n=n)
for batch in [int(x) for x in sys.argv[2].split(",")]:
    sets = []
    for j in range(batch):
        d = torch.empty(n * 4, dtype=torch.int64, device="cuda"); ctx.rng_scalars_dev(50 + j, n, d.data_ptr())
        if os.environ.get("HALF_ZERO"): d.view(n >> 10, 2, 512, 4)[:, j % 2] = 0  # the L / R pattern of an IPA round
        sets.append(d)
    ptrs = [d.data_ptr() for d in sets]
    for mode, name in ((0, "general"), (-1, "table")):
        ctx.set_table_mode(mode)
        depth, pend = 4, []
        def run(K):
            for i in range(K):
                if len(pend) == depth: ctx.msm_dev_batch_end(pend.pop(0), batch)
                ctx.msm_dev_batch_begin(i % depth, ptrs, n); pend.append(i % depth)
            while pend: ctx.msm_dev_batch_end(pend.pop(0), batch)
        run(3 * depth); torch.cuda.synchronize()
        K = 100; t = time.time(); run(K); dt = (time.time() - t) / K
        ctx.prof_enable(True); ctx.prof_reset()
        ctx.msm_dev_batch_begin(0, ptrs, n); ctx.msm_dev_batch_end(0, batch)
        pr = ctx.prof(); ctx.prof_enable(False)
        top = "  ".join("%s %.0f" % (k.replace("k_msm_", "").replace("k_tmsm_", "t:").replace("k_smsm_", "s:"), ms / cnt * 1e3) for k, (ms, cnt) in sorted(pr.items(), key=lambda kv: -kv[1][0])[:7])
        print("n=2^%d batch %d %-8s %.3f ms per launch  %.3f ms per MSM  (%.0f/s) | %s" % (lg, batch, name, dt * 1e3, dt * 1e3 / batch, batch / dt, top), flush=True)
ctx.close()
