"""BASELINE config 4: ASDL prover + verifier + decider over K accumulated instances at n = 2^lg
(the shape of the reference's benches/acc.rs:64-98: K x (random_instance + prover), then
K x verifier + 1 x decider).  Prints one JSON line; results are checked by the scheme itself
(every verifier and the decider must accept)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import halo_accumulation_amd as h
from halo_accumulation_amd import acc as A

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 20
K = int(sys.argv[2]) if len(sys.argv) > 2 else 64
n = 1 << lg; d = n - 1
ctx = h._lib.Context(urs_n=n)
if os.environ.get("FOLD_TABLE"): ctx.set_fold_table(int(os.environ["FOLD_TABLE"]))  # -1 default, 0 never, 1 at the first open
rng = [0x48414C4F00000004]
accs, qss, acc = [], [], None
t0 = time.perf_counter(); t_inst = 0.0; t_prov = 0.0
inst_ms, prov_ms, status = [], [], []
for _ in range(K):
    t = time.perf_counter(); q = A.random_instance(ctx, rng, d); t_inst += time.perf_counter() - t; inst_ms.append((time.perf_counter() - t) * 1e3)
    qs = [q] if acc is None else [A.instance_from_accumulator(ctx, acc, d), q]
    t = time.perf_counter(); acc = A.prover(ctx, rng, d, qs); t_prov += time.perf_counter() - t; prov_ms.append((time.perf_counter() - t) * 1e3)
    status.append(int(ctx.info(5)))
    accs.append(acc); qss.append(qs)
t_chain = time.perf_counter() - t0
t = time.perf_counter()
for a, qs in zip(accs, qss):
    A.verifier(ctx, d, qs, a)
t_ver = time.perf_counter() - t
t = time.perf_counter(); A.decider(ctx, accs[-1]); t_dec = time.perf_counter() - t
t = time.perf_counter()
for a in accs[: min(K, 8)]:
    A.decider(ctx, a)
t_slow = (time.perf_counter() - t) / min(K, 8)
print(json.dumps({"config": "ASDL over %d accumulated instances, n=2^%d, 1 GPU" % (K, lg),
                  "prover_chain_s": t_chain, "random_instance_ms_each": t_inst / K * 1e3, "prover_ms_each": t_prov / K * 1e3,
                  "fast_check_s (K verifiers + 1 decider, benches/acc.rs:64-74)": t_ver + t_dec,
                  "verifier_ms_each": t_ver / K * 1e3, "decider_ms": t_dec * 1e3,
                  "slow_check_s (K deciders, benches/acc.rs:100-106, extrapolated from %d)" % min(K, 8): t_slow * K,
                  "fold_table_build_ms (inside the chain: the key's 8th full-size open asks for the table's memory on a helper thread, the first later open that finds it builds the table, once)": ctx.info(2) / 1e3,
                  "fold_table_first_requested_at_step": next((i for i, s_ in enumerate(status) if s_ >= 1), None),
                  "fold_table_built_at_step": next((i for i, s_ in enumerate(status) if s_ == 2), None),
                  "prover_ms_median": sorted(prov_ms)[len(prov_ms) // 2],
                  "slowest_provers (step, ms)": sorted([(round(m, 1), i) for i, m in enumerate(prov_ms)], reverse=True)[:3],
                  "note": "a step that runs while the helper thread sits in its 40 GB hipMalloc waits for the driver (0.1 - 2 s, once per process, on a box "
                          "whose memory the process has not had before): it shows in the slowest steps, not in the medians",
                  "prover_chain_s_without_the_table_build": t_chain - ctx.info(2) / 1e6,
                  "slowest_random_instances (step, ms)": sorted([(round(m, 1), i) for i, m in enumerate(inst_ms)], reverse=True)[:3],
                  "random_instance_ms_median": sorted(inst_ms)[len(inst_ms) // 2],
                  "all_accepted": True}))
