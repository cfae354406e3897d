"""halo_h_accumulate (acc.rs:85-94 through ffi::h_accumulate) for m instances at n = 2^lg: into a fresh output array and into a reused one.
Usage: time_haccum.py [lg=20] [m=2]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import halo_accumulation_amd as h
L = h._lib
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 20
m = int(sys.argv[2]) if len(sys.argv) > 2 else 2
n = 1 << lg
ctx = L.Context(urs_n=n)
d = torch.empty(4096 * 4, dtype=torch.int64, device="cuda")
ctx.rng_scalars_dev(5, 4096, d.data_ptr())
co = np.ascontiguousarray(d.cpu().numpy().view(np.uint64).reshape(4096, 4))
xis = np.ascontiguousarray(co[: m * (lg + 1)].reshape(m, lg + 1, 4)); al = np.ascontiguousarray(co[2000:2000 + m]); h0 = np.ascontiguousarray(co[3000:3002])
want = ctx.h_accumulate(h0, xis, al)
lib = L.load()
out = np.zeros((n, 4), dtype=np.uint64); out[:] = 1  # touched
def call(o):
    L.check(lib.halo_h_accumulate(ctx.h, L.ptr(h0), L.ptr(xis), L.ptr(al), m, lg, L.ptr(o)))
ts_r, ts_f, ts_w = [], [], []
for _ in range(7):
    t0 = time.perf_counter(); call(out); ts_r.append(time.perf_counter() - t0)
    assert out.tolist() == want.tolist()
    t0 = time.perf_counter(); o2 = np.empty((n, 4), dtype=np.uint64); call(o2); ts_f.append(time.perf_counter() - t0)
    t0 = time.perf_counter(); o3 = ctx.h_accumulate(h0, xis, al); ts_w.append(time.perf_counter() - t0)
    del o2, o3
med = lambda v: sorted(v)[len(v) // 2] * 1e3
print("lg=%d m=%d  halo_h_accumulate: reused output %.3f ms, fresh np.empty output %.3f ms, Python wrapper (np.zeros) %.3f ms" % (lg, m, med(ts_r), med(ts_f), med(ts_w)))
ctx.close()
