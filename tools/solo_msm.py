"""Latency of ONE halo_msm_dev call at n = 2^lg, nothing else in flight (median of K), and the result's first words.
Usage: solo_msm.py LG [K=12]   (HALO_PIECE_ALTERNATE=0: the pieces of a large MSM one after the other)"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import halo_accumulation_amd as h
lg = int(sys.argv[1]); K = int(sys.argv[2]) if len(sys.argv) > 2 else 12
n = 1 << lg
ctx = h._lib.Context(urs_n=n)
d = torch.empty(n * 4, dtype=torch.int64, device="cuda")
ctx.rng_scalars_dev(2, n, d.data_ptr())
for _ in range(4): r = ctx.msm_dev(d.data_ptr(), n)
ts = []
for _ in range(K):
    t0 = time.perf_counter(); r2 = ctx.msm_dev(d.data_ptr(), n); ts.append(time.perf_counter() - t0)
    assert r2.tolist() == r.tolist()  # (every repetition: the pieces of a large MSM publish their sums to one counter)
print("lg=%d solo %.3f ms (min %.3f)  result %016x" % (lg, sorted(ts)[len(ts) // 2] * 1e3, min(ts) * 1e3, int(r[0])))
ctx.close()
