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
    // inv() (62 division steps at a time, host_math.hpp modinv) against the Fermat power, both fields: random values, small
    // integers and their negatives, powers of two, raw limb patterns (any value below the modulus is some element)
    {
        long bad = 0;
        Fr x = urs_scalar(7);
        Fq q = Point::generator().mul(urs_scalar(8)).to_affine().x;
        for (int i = 0; i < 3000; ++i) {
            x = x * x + Fr::from_u64(i + 3); q = q * q + Fq::from_u64(5);
            bad += !(x.inv() == x.inv_fermat()) + !(q.inv() == q.inv_fermat());
            bad += !(x * x.inv() == Fr::one()) + !(q * q.inv() == Fq::one());
        }
        for (u64 k = 1; k < 200; ++k) {
            Fr a = Fr::from_u64(k); Fq b = Fq::from_u64(k);
            bad += !(a.inv() == a.inv_fermat()) + !((-a).inv() == (-a).inv_fermat()) + !(b.inv() == b.inv_fermat()) + !((-b).inv() == (-b).inv_fermat());
            Fr r1{{k, 0, 0, 0}}, r2{{0, 0, 0, k}}, r3{{~(u64)0, ~(u64)0, ~(u64)0, k}};
            bad += !(r1.inv() == r1.inv_fermat()) + !(r2.inv() == r2.inv_fermat()) + !(r3.inv() == r3.inv_fermat());
            Fq s1{{k, k, k, k & 0xfff}};
            bad += !(s1.inv() == s1.inv_fermat());
        }
        Fr p2 = Fr::one();
        for (int s = 0; s < 260; ++s) { p2 = p2 + p2; bad += !(p2.inv() == p2.inv_fermat()) + !((-p2).inv() == (-p2).inv_fermat()); }
        if (!Fr::zero().inv().is_zero() || bad) { printf("inverse mismatch: %ld\n", bad); return 1; }
    }
    uint64_t w[12]; acc.store_normalized(w);
    printf("ok %016llx\n", (unsigned long long)w[0]);
    return 0;
}
