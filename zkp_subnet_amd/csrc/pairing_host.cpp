// Host-side KZG opening verifier (BLS12-381 optimal-ate pairing) -- product code, plain C++17, no GPU.
//
// Replaces what the reference reaches through client.worker_verify(i, proof, alpha, eval, commitment)
// (reference neurons/validator.py:77-86, used by reward() at :160-170 and tests/test_miner.py:101-111):
//     accept  <=>  e(C - y [L_i(tau_y)]_1, [1]_2) == e(pi, [tau_x - alpha]_2)
// checked as  e(C - y L_i, -G2) * e(pi, tau_G2 - alpha G2) == 1  (two Miller loops, one final exponentiation).
// Verification is one-off per proof (two pairings), so it runs on the host; the MSM / NTT
// hot path never touches this file.  Tower Fp2 = Fp[u]/(u^2+1), Fp6 = Fp2[v]/(v^3 - (1+u)), Fp12 = Fp6[w]/(w^2 - v);
// affine Miller loop on the twist with full Fp12 line values; final exponentiation = easy part by conjugation and
// inversion, hard part as five exponentiations by the curve parameter + Frobenius maps (final_exp_fast; the exact
// one-exponent form final_exp is kept for the test hook, which also cross-checks the two).  Checked against the independent pure-Python pairing of
// oracle/pairing.py (direct degree-12 extension) in tests/test_verify.py.
#include <cerrno>
#include <cstdint>
#include <cstring>
#include <mutex>
#include <new>
#include <system_error>
#include <thread>
#include <vector>

#include <sys/random.h>
#include <sys/types.h>

#include "../../include/kzg_mi355x.h"
#include "../../include/kzg_mi355x_test.h"

#include "fp_host.h"

namespace {

using namespace kzg_host;


// ------------------------------------------------------------------------------------------------ Fp2
struct Fp2 {
    Fp c0, c1;
};
inline Fp2 operator+(const Fp2& a, const Fp2& b) { return {a.c0 + b.c0, a.c1 + b.c1}; }
inline Fp2 operator-(const Fp2& a, const Fp2& b) { return {a.c0 - b.c0, a.c1 - b.c1}; }
inline Fp2 neg(const Fp2& a) { return {neg(a.c0), neg(a.c1)}; }
inline bool is_zero(const Fp2& a) { return is_zero(a.c0) && is_zero(a.c1); }
inline bool operator==(const Fp2& a, const Fp2& b) { return a.c0 == b.c0 && a.c1 == b.c1; }
inline Fp2 operator*(const Fp2& a, const Fp2& b) {
    Fp t0 = a.c0 * b.c0, t1 = a.c1 * b.c1;
    Fp m = (a.c0 + a.c1) * (b.c0 + b.c1);
    return {t0 - t1, m - t0 - t1};
}
inline Fp2 mul_xi(const Fp2& a) { return {a.c0 - a.c1, a.c0 + a.c1}; }  // * (1 + u)
Fp2 inv(const Fp2& a) {
    Fp d = inv(a.c0 * a.c0 + a.c1 * a.c1);
    return {a.c0 * d, neg(a.c1) * d};
}
Fp2 fp2_small(u64 v) { return {fp_small(v), Fp{}}; }
Fp2 fp2_one() { return {FP_R, Fp{}}; }

// ------------------------------------------------------------------------------------------------ Fp6, Fp12
struct Fp6 {
    Fp2 c0, c1, c2;
};
inline Fp6 operator+(const Fp6& a, const Fp6& b) { return {a.c0 + b.c0, a.c1 + b.c1, a.c2 + b.c2}; }
inline Fp6 operator-(const Fp6& a, const Fp6& b) { return {a.c0 - b.c0, a.c1 - b.c1, a.c2 - b.c2}; }
inline Fp6 neg(const Fp6& a) { return {neg(a.c0), neg(a.c1), neg(a.c2)}; }
inline Fp6 operator*(const Fp6& a, const Fp6& b) {
    Fp2 t0 = a.c0 * b.c0, t1 = a.c1 * b.c1, t2 = a.c2 * b.c2;
    Fp2 c0 = t0 + mul_xi((a.c1 + a.c2) * (b.c1 + b.c2) - t1 - t2);
    Fp2 c1 = (a.c0 + a.c1) * (b.c0 + b.c1) - t0 - t1 + mul_xi(t2);
    Fp2 c2 = (a.c0 + a.c2) * (b.c0 + b.c2) - t0 - t2 + t1;
    return {c0, c1, c2};
}
inline Fp6 mul_v(const Fp6& a) { return {mul_xi(a.c2), a.c0, a.c1}; }
Fp6 inv(const Fp6& a) {
    Fp2 c0 = a.c0 * a.c0 - mul_xi(a.c1 * a.c2);
    Fp2 c1 = mul_xi(a.c2 * a.c2) - a.c0 * a.c1;
    Fp2 c2 = a.c1 * a.c1 - a.c0 * a.c2;
    Fp2 t = inv(mul_xi(a.c2 * c1 + a.c1 * c2) + a.c0 * c0);
    return {c0 * t, c1 * t, c2 * t};
}
struct Fp12 {
    Fp6 c0, c1;
};
inline Fp12 operator*(const Fp12& a, const Fp12& b) {
    Fp6 t0 = a.c0 * b.c0, t1 = a.c1 * b.c1;
    return {t0 + mul_v(t1), (a.c0 + a.c1) * (b.c0 + b.c1) - t0 - t1};
}
inline Fp12 conj(const Fp12& a) { return {a.c0, neg(a.c1)}; }
Fp12 inv(const Fp12& a) {
    Fp6 t = inv(a.c0 * a.c0 - mul_v(a.c1 * a.c1));
    return {a.c0 * t, neg(a.c1) * t};
}
Fp12 fp12_one() {
    Fp12 r{};
    r.c0.c0 = fp2_one();
    return r;
}
bool is_one(const Fp12& a) {
    Fp12 o = fp12_one();
    return memcmp(&a, &o, sizeof(a)) == 0;
}

// ------------------------------------------------------------------------------------------------ curves
template <class F>
struct Aff {
    F x, y;
    bool inf;
};
template <class F>
struct Jac {
    F x, y, z;
};
template <class F>
F f_one();
template <>
Fp f_one<Fp>() { return FP_R; }
template <>
Fp2 f_one<Fp2>() { return fp2_one(); }
template <class F>
Jac<F> jac_inf() {
    Jac<F> r{};
    r.x = f_one<F>();
    r.y = f_one<F>();
    return r;
}
template <class F>
Jac<F> to_jac(const Aff<F>& a) {
    if (a.inf) return jac_inf<F>();
    return {a.x, a.y, f_one<F>()};
}
template <class F>
Jac<F> jac_double(const Jac<F>& p) {
    if (is_zero(p.z) || is_zero(p.y)) return jac_inf<F>();
    F A = p.x * p.x, B = p.y * p.y, C = B * B;
    F t = p.x + B;
    F D = t * t - A - C;
    D = D + D;
    F E = A + A + A, Fq = E * E;
    F x3 = Fq - D - D;
    F c8 = C + C;
    c8 = c8 + c8;
    c8 = c8 + c8;
    F y3 = E * (D - x3) - c8;
    F z3 = p.y * p.z;
    z3 = z3 + z3;
    return {x3, y3, z3};
}
template <class F>
Jac<F> jac_add(const Jac<F>& p, const Jac<F>& q) {
    if (is_zero(p.z)) return q;
    if (is_zero(q.z)) return p;
    F z1z1 = p.z * p.z, z2z2 = q.z * q.z;
    F u1 = p.x * z2z2, u2 = q.x * z1z1;
    F s1 = p.y * q.z * z2z2, s2 = q.y * p.z * z1z1;
    if (u1 == u2) {
        if (s1 == s2) return jac_double(p);
        return jac_inf<F>();
    }
    F h = u2 - u1, rr = s2 - s1;
    F hh = h * h, hhh = h * hh, v = u1 * hh;
    F x3 = rr * rr - hhh - v - v;
    F y3 = rr * (v - x3) - s1 * hhh;
    F z3 = p.z * q.z * h;
    return {x3, y3, z3};
}
template <class F>
Jac<F> jac_mul(const Aff<F>& p, const u64* k, int words) {
    Jac<F> acc = jac_inf<F>(), base = to_jac(p);
    for (int i = words * 64 - 1; i >= 0; i--) {
        acc = jac_double(acc);
        if ((k[i / 64] >> (i % 64)) & 1) acc = jac_add(acc, base);
    }
    return acc;
}
template <class F>
Aff<F> to_aff(const Jac<F>& p) {
    if (is_zero(p.z)) return {F{}, F{}, true};
    F zi = inv(p.z), zi2 = zi * zi;
    return {p.x * zi2, p.y * zi2 * zi, false};
}
template <class F>
Aff<F> aff_neg(const Aff<F>& a) {
    return {a.x, neg(a.y), a.inf};
}

typedef Aff<Fp> G1A;
typedef Aff<Fp2> G2A;

bool g1_on_curve(const G1A& p) { return p.inf || p.y * p.y == p.x * p.x * p.x + fp_small(4); }
bool g2_on_curve(const G2A& p) {
    Fp2 b{fp_small(4), fp_small(4)};
    return p.inf || p.y * p.y == p.x * p.x * p.x + b;
}
template <class F>
bool in_subgroup(const Aff<F>& p) {
    return is_zero(jac_mul(p, R_ORDER, 4).z);
}
// G1 only, ~4x cheaper: with sigma(x, y) = (beta x, y) and z = |BLS parameter|, the endomorphism sigma + z^2 has degree
// z^4 - z^2 + 1 = r and kills G1, so its kernel IS G1:  P in G1  <=>  [z^2] P == -sigma(P)  (two multiplications by the
// 64-bit z instead of one by the 255-bit r; the same test as k_g1_subgroup_check on the GPU).
// beta = 0x5f19672f...fffefffe is the cube root of unity that pairs with -z^2; it is re-derived here as 2^((p-1)/3) or its
// square, whichever satisfies the identity on the generator.
Aff<Fp> g1_generator();
static bool g1_endo_identity(const Aff<Fp>& p, const Fp& beta) {
    const Aff<Fp> zp = to_aff(jac_mul(p, &ATE_LOOP, 1));
    if (zp.inf) return false;
    const Jac<Fp> q = jac_mul(zp, &ATE_LOOP, 1);            // [z^2] P = (X / Z^2, Y / Z^3)
    if (is_zero(q.z)) return false;
    const Fp z2 = q.z * q.z, z3 = z2 * q.z;
    return q.x == beta * p.x * z2 && q.y == neg(p.y * z3);
}
static const Fp& g1_beta() {
    static const Fp beta = [] {
        // (p - 1) / 3
        static const u64 E[6] = {0x9354ffffffffe38eULL, 0x0a395554e5c6aaaaULL, 0xcd104635a790520cULL, 0xcc27c3d6fbd7063fULL,
                                 0x190937e76bc3e447ULL, 0x08ab05f8bdd54cdeULL};
        const Fp b1 = fp_pow(fp_small(2), E, 6), b2 = b1 * b1;
        return g1_endo_identity(g1_generator(), b1) ? b1 : b2;
    }();
    return beta;
}
bool g1_in_subgroup_fast(const Aff<Fp>& p) { return p.inf || g1_endo_identity(p, g1_beta()); }
G1A g1_generator() {
    static const uint8_t gx[48] = {0x17, 0xf1, 0xd3, 0xa7, 0x31, 0x97, 0xd7, 0x94, 0x26, 0x95, 0x63, 0x8c, 0x4f, 0xa9, 0xac, 0x0f,
                                   0xc3, 0x68, 0x8c, 0x4f, 0x97, 0x74, 0xb9, 0x05, 0xa1, 0x4e, 0x3a, 0x3f, 0x17, 0x1b, 0xac, 0x58,
                                   0x6c, 0x55, 0xe8, 0x3f, 0xf9, 0x7a, 0x1a, 0xef, 0xfb, 0x3a, 0xf0, 0x0a, 0xdb, 0x22, 0xc6, 0xbb};
    static const uint8_t gy[48] = {0x08, 0xb3, 0xf4, 0x81, 0xe3, 0xaa, 0xa0, 0xf1, 0xa0, 0x9e, 0x30, 0xed, 0x74, 0x1d, 0x8a, 0xe4,
                                   0xfc, 0xf5, 0xe0, 0x95, 0xd5, 0xd0, 0x0a, 0xf6, 0x00, 0xdb, 0x18, 0xcb, 0x2c, 0x04, 0xb3, 0xed,
                                   0xd0, 0x3c, 0xc7, 0x44, 0xa2, 0x88, 0x8a, 0xe4, 0x0c, 0xaa, 0x23, 0x29, 0x46, 0xc5, 0xe7, 0xe1};
    G1A g{};
    fp_from_be48(g.x, gx);
    fp_from_be48(g.y, gy);
    return g;
}
bool hex48(Fp& out, const char* hex) {
    uint8_t b[48];
    for (int i = 0; i < 48; i++) {
        auto nib = [](char c) { return (uint8_t)(c <= '9' ? c - '0' : (c | 32) - 'a' + 10); };
        b[i] = (uint8_t)((nib(hex[2 * i]) << 4) | nib(hex[2 * i + 1]));
    }
    return fp_from_be48(out, b);
}
G2A g2_generator() {
    G2A g{};
    hex48(g.x.c0, "024aa2b2f08f0a91260805272dc51051c6e47ad4fa403b02b4510b647ae3d1770bac0326a805bbefd48056c8c121bdb8");
    hex48(g.x.c1, "13e02b6052719f607dacd3a088274f65596bd0d09920b61ab5da61bbdc7f5049334cf11213945d57e5ac7d055d042b7e");
    hex48(g.y.c0, "0ce5d527727d6e118cc9cdc6da2e351aadfd9baa8cbdd3a76d429a695160d12c923ac9cc3baca289e193548608b82801");
    hex48(g.y.c1, "0606c4a02ea734cc32acd2b02bc28b99cb3e287e85a763af267492ab572e99ab3f370d275cec1da1aaa9075ff05f79be");
    return g;
}

// ZCash compressed G1 -> affine; false when malformed / not on the curve
bool g1_decompress(G1A& out, const uint8_t* c48) {
    if (!(c48[0] & 0x80)) return false;
    if (c48[0] & 0x40) {
        if (c48[0] != 0xC0) return false;
        for (int i = 1; i < 48; i++)
            if (c48[i]) return false;
        out = {Fp{}, Fp{}, true};
        return true;
    }
    uint8_t xb[48];
    memcpy(xb, c48, 48);
    xb[0] &= 0x1F;
    Fp x;
    if (!fp_from_be48(x, xb)) return false;
    Fp rhs = x * x * x + fp_small(4);
    Fp y = fp_pow(rhs, SQRT_EXP, 6);
    if (!(y * y == rhs)) return false;
    u64 yl[6], twice[6];
    fp_to_limbs(yl, y);
    u64 c = add6(twice, yl, yl);
    const bool larger = c || ge6(twice, PM);  // y > (p-1)/2
    if (larger != ((c48[0] & 0x20) != 0)) y = neg(y);
    out = {x, y, false};
    return true;
}
bool g1_from_be96(G1A& out, const uint8_t* b) {
    bool zero = true;
    for (int i = 0; i < 96; i++) zero &= b[i] == 0;
    if (zero) {
        out = {Fp{}, Fp{}, true};
        return true;
    }
    out.inf = false;
    return fp_from_be48(out.x, b) && fp_from_be48(out.y, b + 48) && g1_on_curve(out);
}
// uncompressed G2, ZCash order x.c1 || x.c0 || y.c1 || y.c0
bool g2_from_be192(G2A& out, const uint8_t* b) {
    bool zero = true;
    for (int i = 0; i < 192; i++) zero &= b[i] == 0;
    if (zero) {
        out = {Fp2{}, Fp2{}, true};
        return true;
    }
    out.inf = false;
    return fp_from_be48(out.x.c1, b) && fp_from_be48(out.x.c0, b + 48) && fp_from_be48(out.y.c1, b + 96) &&
           fp_from_be48(out.y.c0, b + 144) && g2_on_curve(out);
}
bool fr_from_be32(u64* k, const uint8_t* b) {  // canonical check
    for (int i = 0; i < 4; i++) {
        u64 v = 0;
        for (int j = 0; j < 8; j++) v = (v << 8) | b[(3 - i) * 8 + j];
        k[i] = v;
    }
    for (int i = 3; i >= 0; i--) {
        if (k[i] != R_ORDER[i]) return k[i] < R_ORDER[i];
    }
    return false;
}

// ------------------------------------------------------------------------------------------------ pairing
// Line through T (and Q, or tangent) on the twist, evaluated at P in G1, scaled by w^3 (killed by the final
// exponentiation):  l = (lam xT - yT)  -  lam xP * v  +  yP * v w
Fp12 line_value(const Fp2& lam, const G2A& t, const G1A& p) {
    Fp12 l{};
    l.c0.c0 = lam * t.x - t.y;
    l.c0.c1 = neg(Fp2{lam.c0 * p.x, lam.c1 * p.x});
    l.c1.c1 = Fp2{p.y, Fp{}};
    return l;
}
Fp12 miller_loop(const G1A& p, const G2A& q) {
    if (p.inf || q.inf) return fp12_one();
    Fp12 f = fp12_one();
    G2A t = q;
    for (int i = 62; i >= 0; i--) {  // bit 63 of |x| is the leading one
        Fp2 lam = (t.x * t.x) * fp2_small(3) * inv(t.y + t.y);
        f = f * f * line_value(lam, t, p);
        Fp2 x3 = lam * lam - t.x - t.x;
        t = {x3, lam * (t.x - x3) - t.y, false};
        if ((ATE_LOOP >> i) & 1) {
            Fp2 lam2 = (q.y - t.y) * inv(q.x - t.x);
            f = f * line_value(lam2, t, p);
            Fp2 x4 = lam2 * lam2 - t.x - q.x;
            t = {x4, lam2 * (t.x - x4) - t.y, false};
        }
    }
    return f;  // the sign of the BLS parameter would invert f: irrelevant for a product-equals-one check
}
Fp12 final_exp(const Fp12& f) {  // exact: f^((p^12 - 1) / r); the hard part as one 2040-bit exponent (~3000 Fp12 products)
    Fp12 f1 = conj(f) * inv(f);  // f^(p^6 - 1)
    Fp12 acc = fp12_one();
    for (int i = 32 * 64 - 1; i >= 0; i--) {
        acc = acc * acc;
        if ((HARD_EXP[i / 64] >> (i % 64)) & 1) acc = acc * f1;
    }
    return acc;
}

// ---- fast final exponentiation for the verifier: f^(3 (p^12 - 1) / r) -- the cube of the exact value, which is 1 exactly
// when the exact value is (r is prime, 3 does not divide it).  Easy part (p^6 - 1)(p^2 + 1) by conjugation, inversion and
// two Frobenius maps; hard part 3 (p^4 - p^2 + 1) / r as five exponentiations by the 64-bit curve parameter |x| (weight
// 6) and a few Frobenius maps / products (the published addition chain for BLS12 curves; the exponent this sequence
// realises was checked symbolically to be exactly 3 (p^4 - p^2 + 1) / r).  ~350 Fp12 products instead of ~3000.
// Frobenius: with w^6 = xi, a -> a^p maps  sum a_ij v^i w^j  to  sum conj(a_ij) gamma^(2i + j) v^i w^j,
// gamma = xi^((p - 1) / 6); the constants are derived at first use instead of being typed in.
inline Fp2 fp2_conj(const Fp2& a) { return {a.c0, neg(a.c1)}; }
struct FrobConsts {
    Fp2 g[6];
    FrobConsts() {
        u64 e[6], rem = 0;                       // (p - 1) / 6 by long division (p - 1 is even and divisible by 3)
        u64 pm1[6];
        memcpy(pm1, PM, sizeof(pm1));
        pm1[0] -= 1;
        for (int i = 5; i >= 0; i--) {
            const u128 cur = ((u128)rem << 64) | pm1[i];
            e[i] = (u64)(cur / 6);
            rem = (u64)(cur % 6);
        }
        const Fp2 xi = {FP_R, FP_R};             // 1 + u
        Fp2 acc = fp2_one();
        for (int i = 6 * 64 - 1; i >= 0; i--) {
            acc = acc * acc;
            if ((e[i / 64] >> (i % 64)) & 1) acc = acc * xi;
        }
        g[0] = fp2_one();
        for (int k = 1; k < 6; k++) g[k] = g[k - 1] * acc;
    }
};
const FrobConsts& frob_consts() {
    static const FrobConsts c;                   // thread-safe one-time initialisation (C++11)
    return c;
}
Fp12 frobenius(const Fp12& a) {
    const FrobConsts& k = frob_consts();
    Fp12 r;
    r.c0.c0 = fp2_conj(a.c0.c0);
    r.c0.c1 = fp2_conj(a.c0.c1) * k.g[2];
    r.c0.c2 = fp2_conj(a.c0.c2) * k.g[4];
    r.c1.c0 = fp2_conj(a.c1.c0) * k.g[1];
    r.c1.c1 = fp2_conj(a.c1.c1) * k.g[3];
    r.c1.c2 = fp2_conj(a.c1.c2) * k.g[5];
    return r;
}
Fp12 cyc_exp_x(const Fp12& f) {  // f^x for the (negative) BLS parameter x, f in the cyclotomic subgroup: f^|x|, conjugated
    Fp12 acc = f;                 // bit 63 of |x|
    for (int i = 62; i >= 0; i--) {
        acc = acc * acc;
        if ((ATE_LOOP >> i) & 1) acc = acc * f;
    }
    return conj(acc);
}
Fp12 final_exp_fast(const Fp12& f) {
    Fp12 t2 = conj(f) * inv(f);                          // f^(p^6 - 1)
    t2 = frobenius(frobenius(t2)) * t2;                  // ^(p^2 + 1): now in the cyclotomic subgroup (conj = inverse)
    Fp12 t1 = conj(t2 * t2);
    Fp12 t3 = cyc_exp_x(t2);
    Fp12 t4 = t3 * t3;
    Fp12 t5 = t1 * t3;
    t1 = cyc_exp_x(t5);
    Fp12 t0 = cyc_exp_x(t1);
    Fp12 t6 = cyc_exp_x(t0);
    t6 = t6 * t4;
    t4 = cyc_exp_x(t6);
    t5 = conj(t5);
    t4 = t4 * (t5 * t2);
    t5 = conj(t2);
    t1 = t1 * t2;
    t1 = frobenius(frobenius(frobenius(t1)));
    t6 = t6 * t5;
    t6 = frobenius(t6);
    t3 = t3 * t0;
    t3 = frobenius(frobenius(t3));
    t3 = t3 * t1;
    t3 = t3 * t6;
    return t3 * t4;
}

struct VerifierKey {
    G2A tau_g2;
    G2A g2;
    std::vector<G1A> li;  // [L_i(tau_y)]_1 per resident slice
};

}  // namespace

struct kzg_vk {
    VerifierKey k;
};

extern "C" {

int kzg_vk_create(const uint8_t tau_g2_be192[192], const uint8_t* li_g1_be96, uint32_t n_slices, kzg_vk** out) {
    if (!tau_g2_be192 || !li_g1_be96 || !n_slices || !out) return KZG_E_ARG;
    kzg_vk* vk = new kzg_vk();
    vk->k.g2 = g2_generator();
    if (!g2_from_be192(vk->k.tau_g2, tau_g2_be192) || !in_subgroup(vk->k.tau_g2)) {
        delete vk;
        return KZG_E_POINT;
    }
    for (uint32_t i = 0; i < n_slices; i++) {
        G1A p;
        if (!g1_from_be96(p, li_g1_be96 + 96 * (size_t)i) || !in_subgroup(p)) {
            delete vk;
            return KZG_E_POINT;
        }
        vk->k.li.push_back(p);
    }
    *out = vk;
    return KZG_OK;
}
/* synthetic setup with a known trapdoor: tau_G2 = [tau] G2, li_k = [s0_k] G1 (same s0 as kzg_gen_srs) */
int kzg_vk_create_synthetic(const uint8_t tau_be32[32], const uint8_t* s0_be32, uint32_t n_slices, kzg_vk** out) {
    if (!tau_be32 || !s0_be32 || !n_slices || !out) return KZG_E_ARG;
    u64 k[4];
    if (!fr_from_be32(k, tau_be32)) return KZG_E_SCALAR;
    kzg_vk* vk = new kzg_vk();
    vk->k.g2 = g2_generator();
    vk->k.tau_g2 = to_aff(jac_mul(vk->k.g2, k, 4));
    const G1A g1 = g1_generator();
    for (uint32_t i = 0; i < n_slices; i++) {
        if (!fr_from_be32(k, s0_be32 + 32 * (size_t)i)) {
            delete vk;
            return KZG_E_SCALAR;
        }
        vk->k.li.push_back(to_aff(jac_mul(g1, k, 4)));
    }
    *out = vk;
    return KZG_OK;
}
void kzg_vk_destroy(kzg_vk* vk) { delete vk; }

static void g2_to_be192(uint8_t* b, const G2A& p) {
    if (p.inf) { memset(b, 0, 192); return; }
    fp_to_be48(b, p.x.c1); fp_to_be48(b + 48, p.x.c0); fp_to_be48(b + 96, p.y.c1); fp_to_be48(b + 144, p.y.c0);
}
/* serialise a verifier key: 192 B [tau_x]_2 then 96 B per slice (the layout of a `<setup>.vk` file) */
int kzg_vk_export(const kzg_vk* vk, uint8_t* out, uint64_t out_len) {
    if (!vk || !out) return KZG_E_ARG;
    const uint64_t need = 192 + 96 * (uint64_t)vk->k.li.size();
    if (out_len < need) return KZG_E_ARG;
    g2_to_be192(out, vk->k.tau_g2);
    for (size_t i = 0; i < vk->k.li.size(); i++) {
        uint8_t* b = out + 192 + 96 * i;
        if (vk->k.li[i].inf) memset(b, 0, 96);
        else { fp_to_be48(b, vk->k.li[i].x); fp_to_be48(b + 48, vk->k.li[i].y); }
    }
    return (int)vk->k.li.size();
}

int kzg_vk_verify(const kzg_vk* vk, uint32_t i, const uint8_t proof48[48], const uint8_t alpha_be32[32],
                  const uint8_t eval_be32[32], const uint8_t commitment48[48], int* out_valid) {
    if (!vk || !proof48 || !alpha_be32 || !eval_be32 || !commitment48 || !out_valid) return KZG_E_ARG;
    *out_valid = 0;
    if (i >= vk->k.li.size()) return KZG_E_ARG;
    u64 alpha[4], y[4];
    if (!fr_from_be32(alpha, alpha_be32) || !fr_from_be32(y, eval_be32)) return KZG_E_SCALAR;
    G1A c, pi;
    // malformed or off-curve / out-of-subgroup group elements are an invalid proof, not a call failure
    if (!g1_decompress(c, commitment48) || !g1_decompress(pi, proof48)) return KZG_OK;
    if (!g1_in_subgroup_fast(c) || !g1_in_subgroup_fast(pi)) return KZG_OK;
    // lhs = C - y * L_i ; rhs_q = tau_G2 - alpha * G2
    Jac<Fp> yl = jac_mul(vk->k.li[i], y, 4);
    G1A lhs = to_aff(jac_add(to_jac(c), to_jac(aff_neg(to_aff(yl)))));
    Jac<Fp2> ag = jac_mul(vk->k.g2, alpha, 4);
    G2A rhs_q = to_aff(jac_add(to_jac(vk->k.tau_g2), to_jac(aff_neg(to_aff(ag)))));
    Fp12 f = miller_loop(lhs, aff_neg(vk->k.g2)) * miller_loop(pi, rhs_q);
    *out_valid = is_one(final_exp_fast(f)) ? 1 : 0;   // the cube of the exact value: 1 exactly when that is 1
    return KZG_OK;
}

/* All rows of a validator step in ONE pairing check.  The rows of a step share alpha (reference neurons/validator.py:
 * 106-120 draws one random point per challenge), so with random 128-bit weights r_i
 *     prod_i [ e(C_i - y_i L_i, G2) * e(pi_i, tau G2 - alpha G2)^-1 ]^(r_i) == 1
 * collapses to two Miller loops on A = sum_i r_i (C_i - y_i L_i) and B = sum_i r_i pi_i: an invalid row survives with
 * probability 2^-128.  Per row that leaves two decompressions, two G1 membership tests and three scalar multiplications
 * (one of them by the 255-bit y_i), spread over `threads` host threads; the pairing is paid once.  *out_all_valid = 1
 * only if EVERY row is valid; on 0 the caller checks row by row to find the culprits (kzg_vk_verify). */
int kzg_vk_verify_batch(const kzg_vk* vk, uint32_t n, const uint32_t* idx, const uint8_t* proofs48,
                        const uint8_t alpha_be32[32], const uint8_t* evals_be32, const uint8_t* commitments48, int threads,
                        int* out_all_valid) {
    if (!vk || !out_all_valid || (n && (!idx || !proofs48 || !alpha_be32 || !evals_be32 || !commitments48))) return KZG_E_ARG;
    *out_all_valid = 0;
    u64 alpha[4];
    if (!fr_from_be32(alpha, alpha_be32)) return KZG_E_SCALAR;
    for (uint32_t i = 0; i < n; i++) {
        u64 y[4];
        if (idx[i] >= vk->k.li.size()) return KZG_E_ARG;
        if (!fr_from_be32(y, evals_be32 + 32 * (size_t)i)) return KZG_E_SCALAR;
    }
    if (n == 0) { *out_all_valid = 1; return KZG_OK; }
    // nothing may cross the C boundary: a failed thread creation (std::system_error) or allocation (std::bad_alloc) is a
    // status code, not std::terminate of the validator process
    try {
    std::vector<u64> w(2 * (size_t)n);                       // the weights: 128 random bits per row
    {
        size_t need = w.size() * sizeof(u64), got = 0;
        while (got < need) {
            const ssize_t r = getrandom(reinterpret_cast<uint8_t*>(w.data()) + got, need - got, 0);
            if (r < 0 && errno == EINTR) continue;
            if (r <= 0) return KZG_E_NOMEM;                  // no randomness, no batch check (a host resource failure)
            got += (size_t)r;
        }
    }
    if (threads < 1) threads = 1;
    {
        unsigned hw = std::thread::hardware_concurrency();
        if (hw == 0) hw = 16;
        if (hw > 64) hw = 64;
        if ((unsigned)threads > hw) threads = (int)hw;       // a thread per row of a 256-row step helps nobody
    }
    if ((uint32_t)threads > n) threads = (int)n;
    std::vector<Jac<Fp>> accA((size_t)threads, jac_inf<Fp>()), accB((size_t)threads, jac_inf<Fp>());
    std::vector<int> bad((size_t)threads, 0);
    auto work = [&](int t) {
        for (uint32_t i = (uint32_t)t; i < n; i += (uint32_t)threads) {
            G1A c, pi;
            u64 y[4];
            (void)fr_from_be32(y, evals_be32 + 32 * (size_t)i);
            if (!g1_decompress(c, commitments48 + 48 * (size_t)i) || !g1_decompress(pi, proofs48 + 48 * (size_t)i) ||
                !g1_in_subgroup_fast(c) || !g1_in_subgroup_fast(pi)) {
                bad[(size_t)t] = 1;
                return;
            }
            const Jac<Fp> yl = jac_mul(vk->k.li[idx[i]], y, 4);
            const G1A d = to_aff(jac_add(to_jac(c), to_jac(aff_neg(to_aff(yl)))));
            accA[(size_t)t] = jac_add(accA[(size_t)t], jac_mul(d, &w[2 * (size_t)i], 2));
            accB[(size_t)t] = jac_add(accB[(size_t)t], jac_mul(pi, &w[2 * (size_t)i], 2));
        }
    };
    {
        std::vector<std::thread> th;
        th.reserve((size_t)threads);
        int started = 1;                                     // slice 0 runs on the caller
        try {
            for (int t = 1; t < threads; t++, started++) th.emplace_back(work, t);
        } catch (const std::system_error&) {                 // out of threads: the caller takes the slices nobody got
        }
        work(0);
        for (int t = started; t < threads; t++) work(t);
        for (auto& x : th) x.join();
    }
    Jac<Fp> A = jac_inf<Fp>(), B = jac_inf<Fp>();
    for (int t = 0; t < threads; t++) {
        if (bad[(size_t)t]) return KZG_OK;                   // a malformed / off-curve / out-of-subgroup element: not all valid
        A = jac_add(A, accA[(size_t)t]);
        B = jac_add(B, accB[(size_t)t]);
    }
    const Jac<Fp2> ag = jac_mul(vk->k.g2, alpha, 4);
    const G2A rhs_q = to_aff(jac_add(to_jac(vk->k.tau_g2), to_jac(aff_neg(to_aff(ag)))));
    const Fp12 f = miller_loop(to_aff(A), aff_neg(vk->k.g2)) * miller_loop(to_aff(B), rhs_q);
    *out_all_valid = is_one(final_exp_fast(f)) ? 1 : 0;
    return KZG_OK;
    } catch (const std::bad_alloc&) {
        return KZG_E_NOMEM;
    } catch (...) {
        return KZG_E_NOMEM;
    }
}

/* test hook: out = final_exp(miller(P, Q)) as 12 x 48 bytes in tower order
 * (c0.c0.c0, c0.c0.c1, c0.c1.c0, ..., c1.c2.c1); P affine be96, Q uncompressed be192 */
int kzg_vk_pairing(const uint8_t p_be96[96], const uint8_t q_be192[192], uint8_t out_fp12[576]) {
    if (!p_be96 || !q_be192 || !out_fp12) return KZG_E_ARG;
    G1A p;
    G2A q;
    if (!g1_from_be96(p, p_be96) || !g2_from_be192(q, q_be192)) return KZG_E_POINT;
    const Fp12 ml = miller_loop(p, q);
    Fp12 e = final_exp(ml);
    // self-check of the verifier's fast path against the exact one: fast(f) == exact(f)^3
    const Fp12 fast = final_exp_fast(ml), cube = e * e * e;
    if (memcmp(&fast, &cube, sizeof(fast)) != 0) return KZG_E_HIP;   // (no better code: an internal inconsistency)
    const Fp* c = reinterpret_cast<const Fp*>(&e);
    for (int k = 0; k < 12; k++) fp_to_be48(out_fp12 + 48 * k, c[k]);
    return KZG_OK;
}

}  // extern "C"
