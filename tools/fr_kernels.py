"""The bandwidth-side Fr kernels (K4-K9), each ALONE at n = 2^lg: `reps` back-to-back launches through the library's
measurement hook (halo_bench_fr_kernel), event-timed per launch, one JSON object (GB/s against each kernel's algorithmic
bytes).  Also the program to run under `rocprofv3 --kernel-trace --stats` and the `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE`
passes (tools/pmc_summary.py).   Usage: fr_kernels.py [lg=20] [reps=20]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import halo_accumulation_amd as h
from halo_accumulation_amd import _lib

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 20
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
n = 1 << lg
KERNELS = [  # which, kernel, algorithmic bytes, what
    (0, "k_powers", 32 * n, "construct_powers (group.rs:29-37): n x 32 B written; 1 product per element (+ 4 per chain of 4..16): VALU-bound"),
    (1, "k_poly_eval_partial", 32 * n, "p(z) (pcdl.rs:135): n x 32 B read; 1 product per element (+ 5 per chain): VALU-bound"),
    (2, "k_dot2_partial", 64 * n, "scalar_dot (group.rs:13-15), one pair of vectors: 2 x n x 32 B read; 1 product per 64 B"),
    (3, "k_dot2_partial", 128 * (n // 2), "the two dot products of an IPA round (pcdl.rs:203,207), m = n/2: 4 x m x 32 B read"),
    (4, "k_h_coeffs", 32 * n, "HPoly::get_poly (pcdl.rs:56-77): n x 32 B written; 1.25 products per element: VALU-bound"),
    (5, "k_fold_scalars", 192 * (n // 2), "c, z folds (pcdl.rs:222-223), m = n/2: 4 x 32 B read + 2 x 32 B written, 2 products per element"),
    (6, "k_axpy", 96 * n, "p + alpha p_bar (pcdl.rs:156): 2 x 32 B read + 32 B written, 1 product per element"),
]
ctx = _lib.Context(urs_n=n)
ctx.set_table_mode(0)
res = []
ctx.bench_fr_kernel(5, n, 3)  # warm-up
for which, name, alg, what in KERNELS:
    ctx.prof_enable(1)
    ctx.prof_reset()
    ctx.bench_fr_kernel(which, n, reps)
    ms, cnt = ctx.prof()[name]
    ctx.prof_enable(0)
    res.append({"kernel": name, "which": which, "launches": cnt, "ms": ms / cnt, "algorithmic_bytes": alg, "GB/s": alg / (ms / cnt) / 1e6,
                "frac_of_8TBs": alg / (ms / cnt) / 1e6 / 8000.0, "what": what})
print(json.dumps({"n": n, "reps": reps, "timing": "HIP events around each of `reps` back-to-back launches (halo_prof_enable 1), mean", "kernels": res}, indent=1))
ctx.close()
