// Exercises the host-side arithmetic of the library (csrc/host_math.hpp: fields, group law, fixed-base table, GLV digit
// expansion, SHA3-based key scalars) under AddressSanitizer + UBSan; built and run by tests/test_host_sanitizers.py.
#include <cstdio>
#include <cstring>
#include <random>
#include "host_math.hpp"
using namespace halo::host;
int main() {
    std::mt19937_64 g(1);
    Point G = Point::generator();
    Point acc = Point::infinity();
    for (int i = 0; i < 200; ++i) {
        Fr k = urs_scalar(i);
        Point p = G.mul(k);
        acc = acc + p;
        GlvDigits d = glv_digits(k);
        if (d.n > 132) { printf("bad digits\n"); return 1; }
        FixedBaseTable t(p);
        Fr s = urs_scalar(1000 + i);
        Point a = t.mul(s), b = p.mul(s);
        uint64_t wa[12], wb[12];
        a.store_normalized(wa); b.store_normalized(wb);
        if (memcmp(wa, wb, 96)) { printf("table mismatch\n"); return 1; }
    }
    uint64_t w[12]; acc.store_normalized(w);
    printf("ok %016llx\n", (unsigned long long)w[0]);
    return 0;
}
