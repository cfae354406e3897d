#!/usr/bin/env python3
"""Generates tests/golden/open_2_<lg>.json: pcdl::open proofs at n = 2^lg from the CPU oracle (lg = 19 and 20 are committed).

Why: the HIP fold kernel switches to two points per lane sharing one inversion once a round folds
m >= 2^18 points (ipa.hip ipa_fold_points), i.e. from n = 2^19 on; the oracle (double-and-add fold,
one into_affine inversion per point -- pcdl.rs:204-224, group.rs:19) needs minutes at that size, so
its output is committed as a fixture instead of being recomputed in the GPU test.  The fixture is
data: seeds, the proof blob's SHA-256 and a few of its fields for diagnosis.

n = 2^20 is BASELINE config 3's size: the only size at which an open runs the tagged L/R launch over the c = 20 table
(msm.hip MsmBatch::tagged, ipa.hip k_nofold_expand_tagged) and the c = 20 table plan.  With --acc the file also holds one
acc::prover step (acc.rs:190-220) over two random instances (benches/acc.rs:15-29) at the same size: SHA-256 of both
instance blobs and of the accumulator blob, plus the accumulator's fields.

Run in the build container:  python tests/golden/make_open_fixture.py [lg] [--acc]
(lg = 19: about 6 minutes on 3 cores; lg = 20 --acc: about 15 minutes on 5 cores)
"""
import hashlib
import json
import os
import sys
import threading
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import numpy as np

import orc

LG = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 19
WITH_ACC = "--acc" in sys.argv
N = 1 << LG
Q_SEEDS = (0x48414C4F00000041, 0x48414C4F00000042)
ACC_SEED = 0x48414C4F00000043
COEFF_SEED = 0x48414C4F00000003
OPEN_SEED = 4242


def hexw(a):
    return [format(int(x), "016x") for x in np.asarray(a).reshape(-1)]


def main():
    t0 = time.time()
    # URS in three threads (ctypes releases the GIL)
    parts = [None] * 4
    def urs(k):
        parts[k] = orc.urs_affine(2 + k * (N // 4), N // 4)
    th = [threading.Thread(target=urs, args=(k,)) for k in range(4)]
    [t.start() for t in th]; [t.join() for t in th]
    gs = np.ascontiguousarray(np.concatenate(parts))
    print("urs", time.time() - t0, flush=True)
    pp = orc.make_pp(gs)
    deg = N - 5  # not a full polynomial: exercises the zero padding (pcdl.rs:106-107,183-184)
    coeffs, s = orc.rng_scalars(COEFF_SEED, deg + 1)
    zw, _ = orc.rng_scalars(s, 2)
    z = zw[0]
    d = N - 1
    out = {"lg_n": LG, "coeff_seed": COEFF_SEED, "open_seed": OPEN_SEED, "deg": deg,
           "generator": "tests/golden/make_open_fixture.py (oracle/halo_cpu.c orc_pcdl_commit / orc_pcdl_open)", "cases": {}}
    res = {}

    def case(name, w):
        C = orc.pcdl_commit(pp, coeffs, d, w)
        pf, st = orc.pcdl_open(pp, OPEN_SEED, coeffs, C, d, z, w)
        v = orc.poly_eval(coeffs, z)
        orc.pcdl_check(pp, C, d, z, v, pf)  # the oracle's own verifier accepts it
        o = 2 + 24 * LG
        res[name] = {"hiding": w is not None, "C": hexw(C), "v": hexw(v), "rng_state_after": format(st, "016x"),
                     "proof_sha256": hashlib.sha256(pf.tobytes()).hexdigest(),
                     "L0": hexw(pf[2:14]), "R0": hexw(pf[2 + 12 * LG: 14 + 12 * LG]),
                     "L1": hexw(pf[14:26]), "L_last": hexw(pf[2 + 12 * (LG - 1): 2 + 12 * LG]),
                     "U": hexw(pf[o: o + 12]), "c": hexw(pf[o + 12: o + 16])}
        print(name, time.time() - t0, flush=True)

    qs = [None, None]

    def inst(k):
        qs[k] = orc.random_instance(pp, Q_SEEDS[k], d)
        print("instance", k, time.time() - t0, flush=True)

    th = [threading.Thread(target=case, args=("plain", None)), threading.Thread(target=case, args=("hiding", zw[1]))]
    if WITH_ACC:
        th += [threading.Thread(target=inst, args=(k,)) for k in range(2)]
    [t.start() for t in th]; [t.join() for t in th]
    out["cases"] = {k: res[k] for k in ("plain", "hiding")}
    if WITH_ACC:
        acc, st = orc.acc_prover(pp, ACC_SEED, d, [q for q, _ in qs])
        orc.acc_verifier(pp, d, [q for q, _ in qs], acc)  # the oracle's own verifier and decider accept it
        orc.acc_decider(pp, acc)
        iw = orc.instance_words(LG)
        out["acc"] = {"q_seeds": [format(x, "016x") for x in Q_SEEDS], "acc_seed": format(ACC_SEED, "016x"),
                      "q_sha256": [hashlib.sha256(q.tobytes()).hexdigest() for q, _ in qs],
                      "q_rng_state_after": [format(s_, "016x") for _, s_ in qs],
                      "q_head": [hexw(q[:21]) for q, _ in qs],  # C, d, z, v of each instance
                      "acc_sha256": hashlib.sha256(acc.tobytes()).hexdigest(), "acc_rng_state_after": format(st, "016x"),
                      "acc_head": hexw(acc[:21]), "acc_tail": hexw(acc[iw:])}  # instance head; (C_bar', alpha... ) words behind it
        print("acc", time.time() - t0, flush=True)
    with open(os.path.join(HERE, "open_2_%d.json" % LG), "w") as f:
        json.dump(out, f, indent=1)
    print("done", time.time() - t0)


if __name__ == "__main__":
    main()
