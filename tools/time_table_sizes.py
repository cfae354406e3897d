"""Table pipeline against the general one at the key sizes of a rank's index shard: python tools/time_table_sizes.py [lg ...]"""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import halo_accumulation_amd as h
import torch
for lg in [int(a) for a in sys.argv[1:]] or [17, 18, 19]:
    n = 1 << lg
    ctx = h._lib.Context(urs_n=n)
    d = torch.empty(n * 4, dtype=torch.int64, device="cuda")
    ctx.rng_scalars_dev(2, n, d.data_ptr())
    for mode, name in ((0, "general"), (-1, "table")):
        ctx.set_table_mode(mode)
        ref = ctx.msm_dev(d.data_ptr(), n)
        for _ in range(3): ctx.msm_dev(d.data_ptr(), n)
        t = time.time(); K = 20
        for _ in range(K): ctx.msm_dev(d.data_ptr(), n)
        solo = (time.time() - t) / K
        depth, K2, pend = 4, 200, []
        for i in range(2 * depth):
            if len(pend) == depth: ctx.msm_dev_end(pend.pop(0))
            ctx.msm_dev_begin(i % depth, d.data_ptr(), n); pend.append(i % depth)
        while pend: ctx.msm_dev_end(pend.pop(0))
        t = time.time()
        for i in range(K2):
            if len(pend) == depth: out = ctx.msm_dev_end(pend.pop(0))
            ctx.msm_dev_begin(i % depth, d.data_ptr(), n); pend.append(i % depth)
        while pend: out = ctx.msm_dev_end(pend.pop(0))
        pipe = (time.time() - t) / K2
        ctx.prof_enable(True); ctx.prof_reset()
        for _ in range(3): ctx.msm_dev(d.data_ptr(), n)
        pr = ctx.prof(); ctx.prof_enable(False)
        top = "  ".join("%s %.0f" % (k.replace("k_msm_", "").replace("k_tmsm_", "t:").replace("k_smsm_", "s:"), ms / cnt * 1e3) for k, (ms, cnt) in sorted(pr.items(), key=lambda kv: -kv[1][0])[:6])
        print("n=2^%d %-8s solo %.3f ms  pipelined(4) %.3f ms  same %s | %s" % (lg, name, solo * 1e3, pipe * 1e3, out.tolist() == ref.tolist(), top), flush=True)
    ctx.close()
