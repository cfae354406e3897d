// Wire format of EvalProof / Instance / Accumulator: what `#[derive(CanonicalSerialize)]` with
// compressed points would write for the reference's structs (pcdl.rs:22-30, acc.rs:21-28,43-59), field by field in
// declaration order, so that proofs can leave the process in arkworks' own encoding:
//   Fr                 32 bytes, little endian, canonical (not Montgomery)
//   Projective point   33 bytes: x little endian, flags in the top bits of byte 32 (255 + 2 flag bits > 256):
//                      bit 7 = y is the larger of {y, -y}, bit 6 = point at infinity (x = 0)     [group.rs:45-60 hashes the same bytes]
//   usize              u64 little endian
//   Vec<T>             u64 little-endian length, then the elements
//   Option<T>          one byte 0 / 1, then T
//   DensePolynomial    its coefficient Vec (ark-poly strips trailing zero coefficients)
// The reference derives no (de)serialisation for these three structs and no Rust toolchain exists here, so the layout is
// "best known", exactly like the transcript encoding of host_math.hpp (parity unpinned).  Decoding validates
// everything a checked arkworks deserialisation would: canonical field elements, flag bits, points on the curve.
// Host-only code: no device is needed or touched.
#include <cstring>

#include "internal.hpp"

namespace halo {
namespace {

using host::Fq;
using host::Fr;
using host::Point;

size_t proof_words(size_t lg) { return 2 + 24 * lg + 32; }
size_t instance_words(size_t lg) { return 21 + proof_words(lg); }

struct Writer {
    uint8_t *out;
    size_t cap, pos = 0;
    bool ok = true;
    void put(const void *p, size_t n) {
        if (pos + n > cap) { ok = false; return; }
        std::memcpy(out + pos, p, n);
        pos += n;
    }
    void u64le(uint64_t v) { put(&v, 8); }
    void byte(uint8_t v) { put(&v, 1); }
    void scalar(const uint64_t *mont) {
        Fr c = Fr::load(mont).from_mont();
        put(c.l, 32);
    }
    void point(const uint64_t *jac) {
        uint8_t b[33] = {0};
        Point p = Point::load(jac);
        if (p.is_inf()) {
            b[32] = 0x40;
        } else {
            host::Affine a = p.to_affine();
            Fq x = a.x.from_mont(), y = a.y.from_mont(), ny = (-a.y).from_mont();
            std::memcpy(b, x.l, 32);
            if (!Fq::geq(ny.l, y.l)) b[32] |= 0x80;  // y > -y
        }
        put(b, 33);
    }
};

// sqrt in Fq (p = 1 mod 2^32): Tonelli-Shanks with the generator 5 of ark-pallas' FqConfig
bool fq_sqrt(const Fq &a, Fq *root) {
    if (a.is_zero()) { *root = a; return true; }
    // p - 1 = 2^32 * t
    static const uint64_t T[4] = {0x094cf91b992d30edULL, 0x00000000224698fcULL, 0x0ULL, 0x40000000ULL};
    static const Fq Z = Fq::from_u64(5).pow(T);  // generator of the 2^32-torsion
    uint64_t tm1h[4];                            // (t - 1) / 2
    { uint64_t one[4] = {1, 0, 0, 0}, t1[4]; Fq::sub_limbs(t1, T, one); for (int i = 0; i < 4; ++i) tm1h[i] = (t1[i] >> 1) | (i < 3 ? t1[i + 1] << 63 : 0); }
    Fq w = a.pow(tm1h);
    Fq x = a * w, b = x * w, z = Z;  // x = a^((t+1)/2), b = a^t
    int v = 32;
    while (b != Fq::one()) {
        int k = 0;
        Fq b2k = b;
        while (b2k != Fq::one()) { b2k = b2k.sqr(); if (++k == v) return false; }  // not a square
        Fq wz = z;
        for (int j = 0; j < v - k - 1; ++j) wz = wz.sqr();
        z = wz.sqr();
        b = b * z;
        x = x * wz;
        v = k;
    }
    *root = x;
    return true;
}

struct Reader {
    const uint8_t *in;
    size_t len, pos = 0;
    bool ok = true;
    bool take(void *p, size_t n) {
        if (!ok || pos + n > len) { ok = false; return false; }
        std::memcpy(p, in + pos, n);
        pos += n;
        return true;
    }
    uint64_t u64le() { uint64_t v = 0; take(&v, 8); return v; }
    int byte() { uint8_t v = 0; take(&v, 1); return v; }
    void scalar(uint64_t *mont_out) {
        Fr c;
        if (!take(c.l, 32)) return;
        if (Fr::geq(c.l, host::FrP::M)) { ok = false; return; }  // not canonical
        c.to_mont().store(mont_out);
    }
    void point(uint64_t *jac_out) {
        uint8_t b[33];
        if (!take(b, 33)) return;
        uint8_t flags = b[32] & 0xC0;
        b[32] &= 0x3F;
        if (flags == 0xC0) { ok = false; return; }
        if (flags == 0x40) {
            for (int i = 0; i < 33; ++i) if (b[i]) { ok = false; return; }
            Point::infinity().store(jac_out);
            return;
        }
        if (b[32]) { ok = false; return; }  // x >= 2^256
        Fq xc;
        std::memcpy(xc.l, b, 32);
        if (Fq::geq(xc.l, host::FqP::M)) { ok = false; return; }
        Fq x = xc.to_mont(), y;
        if (!fq_sqrt(x.sqr() * x + Fq::from_u64(5), &y)) { ok = false; return; }  // not on the curve
        Fq yc = y.from_mont(), nyc = (-y).from_mont();
        bool y_larger = !Fq::geq(nyc.l, yc.l);
        if (y_larger != (flags == 0x80)) y = -y;
        if (y.is_zero() && flags == 0x80) { ok = false; return; }
        Point::from_affine(x, y).store(jac_out);
    }
};

void write_proof(Writer &w, const uint64_t *pf) {
    size_t lg = (size_t)pf[1];
    w.u64le(lg);
    for (size_t i = 0; i < lg; ++i) w.point(pf + 2 + 12 * i);
    w.u64le(lg);
    for (size_t i = 0; i < lg; ++i) w.point(pf + 2 + 12 * lg + 12 * i);
    const uint64_t *tail = pf + 2 + 24 * lg;
    w.point(tail);        // U
    w.scalar(tail + 12);  // c
    w.byte(pf[0] ? 1 : 0);
    if (pf[0]) w.point(tail + 16);  // C_bar
    w.byte(pf[0] ? 1 : 0);
    if (pf[0]) w.scalar(tail + 28);  // w'
}
// returns lg, or (size_t)-1
size_t read_proof(Reader &r, uint64_t *pf, size_t cap_words) {
    uint64_t lg = r.u64le();
    if (!r.ok || lg > 40 || proof_words(lg) > cap_words) { r.ok = false; return (size_t)-1; }
    std::memset(pf, 0, 8 * proof_words(lg));
    pf[1] = lg;
    for (size_t i = 0; i < lg; ++i) r.point(pf + 2 + 12 * i);
    if (r.u64le() != lg) { r.ok = false; return (size_t)-1; }  // Ls and Rs have one entry per round
    for (size_t i = 0; i < lg; ++i) r.point(pf + 2 + 12 * lg + 12 * i);
    uint64_t *tail = pf + 2 + 24 * lg;
    r.point(tail);
    r.scalar(tail + 12);
    int has_cbar = r.byte();
    if (has_cbar > 1) { r.ok = false; return (size_t)-1; }
    if (has_cbar) r.point(tail + 16); else Point::infinity().store(tail + 16);
    int has_w = r.byte();
    if (has_w > 1 || has_w != has_cbar) { r.ok = false; return (size_t)-1; }  // open() sets both or neither (pcdl.rs:137-173)
    if (has_w) r.scalar(tail + 28);
    pf[0] = (uint64_t)has_cbar;
    return r.ok ? (size_t)lg : (size_t)-1;
}
void write_instance(Writer &w, const uint64_t *q) {
    w.point(q);
    w.u64le(q[12]);
    w.scalar(q + 13);
    w.scalar(q + 17);
    write_proof(w, q + 21);
}
size_t read_instance(Reader &r, uint64_t *q, size_t cap_words) {
    if (cap_words < 21) { r.ok = false; return (size_t)-1; }
    r.point(q);
    q[12] = r.u64le();
    r.scalar(q + 13);
    r.scalar(q + 17);
    if (!r.ok) return (size_t)-1;
    size_t lg = read_proof(r, q + 21, cap_words - 21);
    if (lg == (size_t)-1) return lg;
    if (q[12] + 1 != ((uint64_t)1 << lg)) { r.ok = false; return (size_t)-1; }  // d + 1 = 2^lg: the proof has lg rounds
    return lg;
}

// a blob's own words say how long it is: a corrupted or foreign blob must not send the encoder out of bounds
// (the decoders cap lg at 40 as well)
bool blob_header_ok(const uint64_t *pf) { return pf[0] <= 1 && pf[1] <= 40; }
int bad_blob() { set_error("encode: not an EvalProof blob (hiding flag > 1 or lg > 40)"); return HALO_E_ARG; }

int finish_write(const Writer &w, size_t *len) {
    if (!w.ok) { set_error("encode: output buffer too small"); return HALO_E_ARG; }
    *len = w.pos;
    return HALO_OK;
}
int reject(const char *what) { set_error(std::string("decode: malformed ") + what); return HALO_E_REJECT; }

}  // namespace
}  // namespace halo

using namespace halo;

extern "C" {

size_t halo_proof_encoded_size(size_t lg_n, int hiding) { return 8 + 33 * lg_n + 8 + 33 * lg_n + 33 + 32 + 1 + (hiding ? 33 : 0) + 1 + (hiding ? 32 : 0); }
size_t halo_instance_encoded_size(size_t lg_n, int hiding) { return 33 + 8 + 32 + 32 + halo_proof_encoded_size(lg_n, hiding); }
// largest case: both h coefficients present
size_t halo_accumulator_encoded_size(size_t lg_n) { return halo_instance_encoded_size(lg_n, 1) + 8 + 64 + 33 + 32; }

int halo_proof_encode(const uint64_t *proof, uint8_t *out, size_t cap, size_t *len) {
    if (!proof || !out || !len) { set_error("encode: null pointer"); return HALO_E_ARG; }
    if (!blob_header_ok(proof)) return bad_blob();
    Writer w{out, cap};
    write_proof(w, proof);
    return finish_write(w, len);
}
int halo_proof_decode(const uint8_t *in, size_t len, uint64_t *proof_out, size_t cap_words, size_t *lg_n) {
    if (!in || !proof_out || !lg_n) { set_error("decode: null pointer"); return HALO_E_ARG; }
    Reader r{in, len};
    size_t lg = read_proof(r, proof_out, cap_words);
    if (lg == (size_t)-1 || !r.ok || r.pos != len) return reject("EvalProof");
    *lg_n = lg;
    return HALO_OK;
}
int halo_instance_encode(const uint64_t *inst, uint8_t *out, size_t cap, size_t *len) {
    if (!inst || !out || !len) { set_error("encode: null pointer"); return HALO_E_ARG; }
    if (!blob_header_ok(inst + 21)) return bad_blob();
    Writer w{out, cap};
    write_instance(w, inst);
    return finish_write(w, len);
}
int halo_instance_decode(const uint8_t *in, size_t len, uint64_t *inst_out, size_t cap_words, size_t *lg_n) {
    if (!in || !inst_out || !lg_n) { set_error("decode: null pointer"); return HALO_E_ARG; }
    Reader r{in, len};
    size_t lg = read_instance(r, inst_out, cap_words);
    if (lg == (size_t)-1 || !r.ok || r.pos != len) return reject("Instance");
    *lg_n = lg;
    return HALO_OK;
}
// Accumulator = Instance fields | pi_V { h: DensePolynomial (<= 2 coefficients, trailing zeros stripped), U, w }
int halo_accumulator_encode(const uint64_t *acc, uint8_t *out, size_t cap, size_t *len) {
    if (!acc || !out || !len) { set_error("encode: null pointer"); return HALO_E_ARG; }
    if (!blob_header_ok(acc + 21)) return bad_blob();
    Writer w{out, cap};
    write_instance(w, acc);
    size_t lg = (size_t)acc[22];
    const uint64_t *piV = acc + instance_words(lg);
    bool z1 = !(piV[4] | piV[5] | piV[6] | piV[7]), z0 = !(piV[0] | piV[1] | piV[2] | piV[3]);
    uint64_t hlen = z1 ? (z0 ? 0 : 1) : 2;
    w.u64le(hlen);
    for (uint64_t k = 0; k < hlen; ++k) w.scalar(piV + 4 * k);
    w.point(piV + 8);
    w.scalar(piV + 20);
    return finish_write(w, len);
}
int halo_accumulator_decode(const uint8_t *in, size_t len, uint64_t *acc_out, size_t cap_words, size_t *lg_n) {
    if (!in || !acc_out || !lg_n) { set_error("decode: null pointer"); return HALO_E_ARG; }
    Reader r{in, len};
    size_t lg = read_instance(r, acc_out, cap_words);
    if (lg == (size_t)-1 || !r.ok) return reject("Accumulator");
    if (cap_words < instance_words(lg) + 24) { set_error("decode: output buffer too small"); return HALO_E_ARG; }
    uint64_t *piV = acc_out + instance_words(lg);
    std::memset(piV, 0, 24 * 8);
    uint64_t hlen = r.u64le();
    if (!r.ok || hlen > 2) return reject("Accumulator (h has at most two coefficients, acc.rs:192)");
    for (uint64_t k = 0; k < hlen; ++k) r.scalar(piV + 4 * k);
    if (r.ok && hlen && !(piV[4 * (hlen - 1)] | piV[4 * (hlen - 1) + 1] | piV[4 * (hlen - 1) + 2] | piV[4 * (hlen - 1) + 3]))
        return reject("Accumulator (leading coefficient of h is zero)");
    r.point(piV + 8);
    r.scalar(piV + 20);
    if (!r.ok || r.pos != len) return reject("Accumulator");
    *lg_n = lg;
    return HALO_OK;
}

}  // extern "C"
