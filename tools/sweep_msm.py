"""Sweep window bits x reduce span for the MSM: solo latency and depth-4 pipelined throughput (development aid).

    python tools/sweep_msm.py 17 13,14,15 0,4,8,16
"""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import halo_accumulation_amd as h

lg = int(sys.argv[1])
cs = [int(x) for x in sys.argv[2].split(",")]
spans = [int(x) for x in sys.argv[3].split(",")]
n = 1 << lg
ctx = h._lib.Context(urs_n=n)
ctx.set_task_len(int(os.environ.get('TASK_LEN', '0')))
ctx.set_sort_mode(int(os.environ.get('SORT_MODE', '-1')))
d = torch.empty(n * 4, dtype=torch.int64, device="cuda")
ctx.rng_scalars_dev(2, n, d.data_ptr())
ref = None
for c in cs:
    for span in spans:
        ctx.set_window_bits(c)
        ctx.set_reduce_span(span)
        for _ in range(3):
            out = ctx.msm_dev(d.data_ptr(), n)
        if ref is None:
            ref = out.tolist()
        ok = out.tolist() == ref
        K = 10
        t = time.time()
        for _ in range(K):
            out = ctx.msm_dev(d.data_ptr(), n)
        solo = (time.time() - t) / K * 1e3
        res = []
        for depth in (2, 4):
            def run(K2):
                pend = []
                for i in range(K2):
                    if len(pend) == depth:
                        ctx.msm_dev_end(pend.pop(0))
                    ctx.msm_dev_begin(i % depth, d.data_ptr(), n)
                    pend.append(i % depth)
                while pend:
                    o = ctx.msm_dev_end(pend.pop(0))
                return o
            run(3 * depth)
            K2 = 48
            t = time.time()
            o = run(K2)
            res.append((time.time() - t) / K2 * 1e3)
            ok = ok and o.tolist() == ref
        print("lg=%d c=%2d span=%3d  solo %.3f ms  depth2 %.3f ms  depth4 %.3f ms  same=%s" % (lg, c, span, solo, res[0], res[1], ok), flush=True)
