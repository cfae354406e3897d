"""Pipelined MSM loop only (for rocprofv3 --kernel-trace timelines): python tools/pipe_loop.py LG DEPTH K"""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import halo_accumulation_amd as h

lg, depth, K = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
n = 1 << lg
ctx = h._lib.Context(urs_n=n)
if len(sys.argv) > 4:
    ctx.set_window_bits(int(sys.argv[4]))
if len(sys.argv) > 5:
    ctx.set_reduce_span(int(sys.argv[5]))
if os.environ.get("TABLE_MODE"): ctx.set_table_mode(int(os.environ["TABLE_MODE"]))  # 0: the general (variable-base) pipeline
if os.environ.get("TASK_LEN"): ctx.set_task_len(int(os.environ["TASK_LEN"]))  # development sweep
parts = int(os.environ.get("PARTS", "1"))
d = torch.empty(n * 4, dtype=torch.int64, device="cuda")
ctx.rng_scalars_dev(2, n, d.data_ptr())
def run(K2):
    pend = []
    for i in range(K2):
        if len(pend) == depth:
            ctx.msm_dev_end(pend.pop(0))
        ctx.msm_dev_begin(i % depth, d.data_ptr(), n, part=parts - 1, parts=parts)
        pend.append(i % depth)
    while pend:
        ctx.msm_dev_end(pend.pop(0))
run(3 * depth)
torch.cuda.synchronize()
t = time.time()
run(K)
print("lg=%d depth=%d: %.3f ms per MSM" % (lg, depth, (time.time() - t) / K * 1e3))
