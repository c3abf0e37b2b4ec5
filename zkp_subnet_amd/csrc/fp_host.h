// Host-side BLS12-381 base field Fp (6 x 64-bit limbs, CIOS Montgomery, R = 2^384) -- shared by the pairing verifier
// (pairing_host.cpp) and the result encoder (finish_host.cpp).  Plain C++17, no GPU.
#pragma once
#include <cstdint>
#include <cstring>

namespace kzg_host {

typedef uint64_t u64;
typedef unsigned __int128 u128;

// ------------------------------------------------------------------------------------------------ Fp
struct Fp {
    u64 l[6];
};
static const u64 PM[6] = {0xb9feffffffffaaabULL, 0x1eabfffeb153ffffULL, 0x6730d2a0f6b0f624ULL,
                   0x64774b84f38512bfULL, 0x4b1ba7b6434bacd7ULL, 0x1a0111ea397fe69aULL};
static const u64 P_INV = 0x89f3fffcfffcfffdULL;  // -p^-1 mod 2^64
static const Fp FP_R = {{0x760900000002fffdULL, 0xebf4000bc40c0002ULL, 0x5f48985753c758baULL, 0x77ce585370525745ULL,
                  0x5c071a97a256ec6dULL, 0x15f65ec3fa80e493ULL}};   // 2^384 mod p
static const Fp FP_R2 = {{0xf4df1f341c341746ULL, 0x0a76e6a609d104f1ULL, 0x8de5476c4c95b6d5ULL, 0x67eb88a9939d83c0ULL,
                   0x9a793e85b519952dULL, 0x11988fe592cae3aaULL}};  // 2^768 mod p
static const u64 R_ORDER[4] = {0xffffffff00000001ULL, 0x53bda402fffe5bfeULL, 0x3339d80809a1d805ULL, 0x73eda753299d7d48ULL};
static const u64 SQRT_EXP[6] = {0xee7fbfffffffeaabULL, 0x07aaffffac54ffffULL, 0xd9cc34a83dac3d89ULL,
                         0xd91dd2e13ce144afULL, 0x92c6e9ed90d2eb35ULL, 0x0680447a8e5ff9a6ULL};  // (p+1)/4
static const u64 HARD_EXP[32] = {  // (p^6 + 1) / r
    0x8739e1cdc0705d6aULL, 0x09a5256de0381a16ULL, 0x9cf0f70a61c791e2ULL, 0x3a09c4497903f76eULL,
    0x2d7271563890f133ULL, 0x224741b36fec7760ULL, 0x338259c22a12bd40ULL, 0x38ee1cd4778e0de7ULL,
    0xc3b5ef4b188a20b0ULL, 0x1d615d49e2764d7bULL, 0x816101ddd076117dULL, 0xf007c01e7ebe3afcULL,
    0x27d7bd90935021c3ULL, 0xc3b5e2f557c0b15fULL, 0x5e886c94c4f82384ULL, 0xee6a95db11e63f56ULL,
    0x2b822f514a9c4f6fULL, 0x12d6a874d21b73daULL, 0x1304275ef499dffbULL, 0x967878febcb95d1fULL,
    0x4744497f8b2f2922ULL, 0x85a2e707f0841855ULL, 0x9f0c50126c802eecULL, 0xfb46e197bd2fa489ULL,
    0x548ce0809bc5f61aULL, 0xcf56fb1573beaa8cULL, 0xad7375a3763bdf7cULL, 0xe0ec9031179bdeccULL,
    0x6579aea83c48c1daULL, 0xdbf85ae664cf5bb3ULL, 0x7b6f235c55ca7566ULL, 0x000028b314877503ULL};
static const u64 ATE_LOOP = 0xd201000000010000ULL;  // |x|

inline bool ge6(const u64* a, const u64* b) {
    for (int i = 5; i >= 0; i--) {
        if (a[i] != b[i]) return a[i] > b[i];
    }
    return true;
}
inline u64 sub6(u64* r, const u64* a, const u64* b) {
    u64 br = 0;
    for (int i = 0; i < 6; i++) {
        u128 t = (u128)a[i] - b[i] - br;
        r[i] = (u64)t;
        br = (u64)(t >> 64) & 1;
    }
    return br;
}
inline u64 add6(u64* r, const u64* a, const u64* b) {
    u128 c = 0;
    for (int i = 0; i < 6; i++) {
        c += (u128)a[i] + b[i];
        r[i] = (u64)c;
        c >>= 64;
    }
    return (u64)c;
}
inline Fp operator+(const Fp& a, const Fp& b) {
    Fp r;
    u64 c = add6(r.l, a.l, b.l);
    if (c || ge6(r.l, PM)) sub6(r.l, r.l, PM);
    return r;
}
inline Fp operator-(const Fp& a, const Fp& b) {
    Fp r;
    if (sub6(r.l, a.l, b.l)) add6(r.l, r.l, PM);
    return r;
}
inline bool is_zero(const Fp& a) {
    u64 t = 0;
    for (int i = 0; i < 6; i++) t |= a.l[i];
    return t == 0;
}
inline Fp neg(const Fp& a) {
    if (is_zero(a)) return a;
    Fp r;
    sub6(r.l, PM, a.l);
    return r;
}
inline bool operator==(const Fp& a, const Fp& b) { return memcmp(a.l, b.l, sizeof(a.l)) == 0; }
inline Fp operator*(const Fp& a, const Fp& b) {  // CIOS Montgomery product
    u64 t[8] = {0};
    for (int i = 0; i < 6; i++) {
        u128 c = 0;
        for (int j = 0; j < 6; j++) {
            c += (u128)a.l[j] * b.l[i] + t[j];
            t[j] = (u64)c;
            c >>= 64;
        }
        c += t[6];
        t[6] = (u64)c;
        t[7] = (u64)(c >> 64);
        u64 q = t[0] * P_INV;
        c = (u128)q * PM[0] + t[0];
        c >>= 64;
        for (int j = 1; j < 6; j++) {
            c += (u128)q * PM[j] + t[j];
            t[j - 1] = (u64)c;
            c >>= 64;
        }
        c += t[6];
        t[5] = (u64)c;
        t[6] = t[7] + (u64)(c >> 64);
    }
    Fp r;
    if (t[6] || ge6(t, PM)) sub6(r.l, t, PM);
    else memcpy(r.l, t, sizeof(r.l));
    return r;
}
inline Fp fp_pow(const Fp& a, const u64* e, int words) {
    Fp acc = FP_R;
    for (int i = words * 64 - 1; i >= 0; i--) {
        acc = acc * acc;
        if ((e[i / 64] >> (i % 64)) & 1) acc = acc * a;
    }
    return acc;
}
static const Fp FP_R3 = {{0xed48ac6bd94ca1e0ULL, 0x315f831e03a7adf8ULL, 0x9a53352a615e29ddULL, 0x34c04e5e921e1761ULL,
                          0x2512d43565724728ULL, 0x0aa6346091755d4dULL}};  // 2^1152 mod p
inline void shr1_6(u64* a) {
    for (int i = 0; i < 5; i++) a[i] = (a[i] >> 1) | (a[i + 1] << 63);
    a[5] >>= 1;
}
inline bool is_one6(const u64* a) { return a[0] == 1 && !(a[1] | a[2] | a[3] | a[4] | a[5]); }
// Inverse of a Montgomery residue (0 -> 0).  Binary extended Euclid on the plain integers (<= 2 x 381 halvings and
// subtractions of 6-limb numbers: a few microseconds, against ~570 Montgomery products for the Fermat ladder -- this
// runs once per request on the latency path of the result encoder, finish_host.cpp).  Not constant-time; nothing
// secret is inverted here (coordinates of public commitments / proofs).
inline Fp inv(const Fp& a) {
    if (is_zero(a)) return a;
    u64 u[6], v[6], x1[6] = {1, 0, 0, 0, 0, 0}, x2[6] = {0, 0, 0, 0, 0, 0};
    memcpy(u, a.l, sizeof(u));
    memcpy(v, PM, sizeof(v));
    while (!is_one6(u) && !is_one6(v)) {
        while (!(u[0] & 1)) {
            shr1_6(u);
            if (x1[0] & 1) add6(x1, x1, PM);   // x1 < p, so x1 + p < 2^382: no carry out of six limbs
            shr1_6(x1);
        }
        while (!(v[0] & 1)) {
            shr1_6(v);
            if (x2[0] & 1) add6(x2, x2, PM);
            shr1_6(x2);
        }
        if (ge6(u, v)) {
            sub6(u, u, v);
            if (sub6(x1, x1, x2)) add6(x1, x1, PM);
        } else {
            sub6(v, v, u);
            if (sub6(x2, x2, x1)) add6(x2, x2, PM);
        }
    }
    Fp r;
    memcpy(r.l, is_one6(u) ? x1 : x2, sizeof(r.l));   // (a R)^-1 as a plain integer in [0, p)
    return r * FP_R3;                                  // (a R)^-1 R^3 / R = a^-1 R
}
inline bool fp_from_be48(Fp& r, const uint8_t* b) {  // false when >= p
    Fp t;
    for (int i = 0; i < 6; i++) {
        u64 v = 0;
        for (int k = 0; k < 8; k++) v = (v << 8) | b[(5 - i) * 8 + k];
        t.l[i] = v;
    }
    if (ge6(t.l, PM)) return false;
    r = t * FP_R2;
    return true;
}
inline void fp_to_limbs(u64* out, const Fp& a) {
    Fp one{{1, 0, 0, 0, 0, 0}};
    Fp t = a * one;
    memcpy(out, t.l, sizeof(t.l));
}
inline void fp_to_be48(uint8_t* b, const Fp& a) {
    u64 t[6];
    fp_to_limbs(t, a);
    for (int i = 0; i < 6; i++)
        for (int k = 0; k < 8; k++) b[(5 - i) * 8 + k] = (uint8_t)(t[i] >> (56 - 8 * k));
}
inline Fp fp_small(u64 v) {
    Fp t{{v, 0, 0, 0, 0, 0}};
    return t * FP_R2;
}

}  // namespace kzg_host
