// tuning(): the environment is read here and nowhere else in the library (tests/test_abi_cpu.py greps for it).
#include "tuning.hpp"

#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace halo {
namespace {
size_t parse_bytes(const char *e) {  // "48G", "512M", plain bytes
    char *rest = nullptr;
    double v = strtod(e, &rest);
    if (rest && (*rest == 'K' || *rest == 'k')) v *= 1024.0;
    else if (rest && (*rest == 'M' || *rest == 'm')) v *= 1024.0 * 1024.0;
    else if (rest && (*rest == 'G' || *rest == 'g')) v *= 1024.0 * 1024.0 * 1024.0;
    return v > 0 ? (size_t)v : 0;
}
int env_int(const char *name, int dflt) { const char *e = getenv(name); return e ? atoi(e) : dflt; }
bool env_off(const char *name) { const char *e = getenv(name); return e && atoi(e) == 0; }  // "=0" switches a default-on path off
Tuning read_env() {
    Tuning t;
    t.trace = getenv("HALO_TRACE") != nullptr;
    t.ipa_timing = getenv("HALO_IPA_TIMING") != nullptr;
    if (const char *e = getenv("HALO_MEMORY_BUDGET")) { t.memory_budget_set = true; t.memory_budget = parse_bytes(e); }
    t.graphs = env_int("HALO_GRAPHS", -1);
    t.fold_async = env_int("HALO_FOLD_ASYNC", -2);
    if (const char *e = getenv("HALO_HOST_SPLIT")) {
        // "a,b[,c[,d]]": 1 to 4 positive numbers of sixteenths that add up to 16, nothing else in the string
        int v[4] = {0, 0, 0, 0}, k = 0, sum = 0;
        bool ok = true;
        const char *q = e;
        while (ok) {
            char *rest = nullptr;
            long x = strtol(q, &rest, 10);
            if (rest == q || x < 1 || x > 16 || k == 4) { ok = false; break; }
            v[k++] = (int)x;
            sum += (int)x;
            if (*rest == 0) break;
            if (*rest != ',') { ok = false; break; }
            q = rest + 1;
        }
        ok = ok && k >= 1 && sum == 16;
        if (ok) { t.host_split_set = true; t.host_pieces = k; for (int i = 0; i < 4; ++i) t.host_split[i] = v[i]; }
        else fprintf(stderr, "[halo] HALO_HOST_SPLIT=%s ignored: 1 to 4 positive numbers of sixteenths that add up to 16\n", e);
    }
    t.fold_table_after = env_int("HALO_FOLD_TABLE_AFTER", t.fold_table_after);
    t.plan = getenv("HALO_PLAN");
    t.dots_first = env_int("HALO_DOTS_FIRST", -1);
    t.ipa_c_hint = !env_off("HALO_IPA_C_HINT");
    t.u_from_last_round = !env_off("HALO_U_FROM_LAST_ROUND");
    t.pow_e = env_int("HALO_POW_E", 0);
    if (t.pow_e != 0 && (t.pow_e < 4 || t.pow_e > 64 || (t.pow_e & (t.pow_e - 1)) != 0)) {
        // k_h_coeffs relies on a chain length that is a power of two (its mid * high factor is wave-uniform and changes every
        // fourth step): another value would give wrong coefficients
        fprintf(stderr, "[halo] HALO_POW_E=%d ignored: the chain length must be a power of two in [4, 64]\n", t.pow_e);
        t.pow_e = 0;
    }
    t.dot_blocks = env_int("HALO_DOT_BLOCKS", 0);
    t.smsm_kmax = env_int("HALO_SMSM_KMAX", 0);
    t.late_kmax = env_int("HALO_LATE_KMAX", 0);
    t.piece_alternate = !env_off("HALO_PIECE_ALTERNATE");
    t.graph_cache = env_int("HALO_GRAPH_CACHE", t.graph_cache);
    if (t.graph_cache < 1) t.graph_cache = 1;
    if (t.graph_cache > 8) t.graph_cache = 8;
    t.direct_results = !env_off("HALO_DIRECT_RESULTS");
    t.tagged = !env_off("HALO_TAGGED");
    t.reduce_rc = !env_off("HALO_REDUCE_RC");
    t.reduce1_waves = env_int("HALO_REDUCE1_WAVES", t.reduce1_waves);
    t.smsm_wave_task = env_int("HALO_SMSM_WAVE_TASK", 0);
    t.smsm_waves = env_int("HALO_SMSM_WAVES", t.smsm_waves);
    t.smsm_fused = env_int("HALO_SMSM_FUSED", 0) != 0;
    t.host_inv_fermat = getenv("HALO_HOST_INV_FERMAT") != nullptr;
    t.spin_us = env_int("HALO_SPIN_US", t.spin_us);
    return t;
}
}  // namespace

const Tuning &tuning() {
    static const Tuning t = read_env();
    return t;
}
DevHooks &dev_hooks() {
    static DevHooks h;
    return h;
}

}  // namespace halo
