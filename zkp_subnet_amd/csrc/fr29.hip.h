// Fr (BLS12-381 scalar field, 255 bits) on 9 UNSATURATED 29-bit limbs, Montgomery radix R = 2^261 -- the working
// representation of every Fr kernel (NTT, opening scan, codecs, SRS scalars).  HBM keeps 8 x u32 words (a value < 2^256).
//
// Same reasoning as fp28.hip.h: on gfx950 a field product costs its instruction count, and saturated limbs pay a
// carry instruction per limb product (the 8 x 32-bit CIOS form of field.hip.h: ~380 instructions, 128 of them mads).
// With 29-bit limbs a 64-bit column accumulator takes 9 products of up to 60 bits with no carry handling: 81 + 72 mads
// and ~50 other instructions (~205).  Two gifts of this modulus: r = 1 mod 2^29, so the Montgomery quotient digit is
// q = -acc mod 2^29 (no multiplication) and q * r_0 is an addition.  R = 2^261 leaves 6 bits above r (2^261 / r = 70):
// sums of a few dozen residues are legal product inputs, so the butterflies and scans below add lazily.
//
// Classes (value taken mod r):
//   N    : limbs 0..7 < 2^29 (limb 8 holds the excess), value < 2r        -- every product output
//   lazy : limbs < 2^31, value < 64r                                      -- legal FIRST operand of fr9_mul
//   the SECOND operand of fr9_mul must have limbs < 2^29 (normalised) and first * second < 2^261 r.
// Modelled limb-exactly (bounds included) in scripts/models/fr29_model.py, which also generates the constants.
#pragma once
#include "bigint.hip.h"

struct fr9_t {
    uint32_t l[9];
};
#define FR9_MASK 0x1fffffffu

#define FR9_TABLE(name, ...)                                   \
    __host__ __device__ constexpr uint32_t name(int i) {       \
        constexpr uint32_t m[9] = {__VA_ARGS__};               \
        return m[i];                                           \
    }
FR9_TABLE(fr9_r, 0x00000001u, 0x1ffffff8u, 0x1f96ffbfu, 0x1b4805ffu, 0x1d80553bu, 0x0c0404d0u, 0x1520cce7u, 0x0a6533afu, 0x0073eda7u)
FR9_TABLE(fr9_one_c, 0x1fffffbau, 0x0000022fu, 0x1cb61180u, 0x0a4e5c00u, 0x0ee8b1a2u, 0x16e6aedfu, 0x1907f8bbu, 0x0853ddf7u, 0x004d043fu)  // R mod r
FR9_TABLE(fr9_r2, 0x0a71b3c0u, 0x1d32207eu, 0x1663d999u, 0x1c5abc93u, 0x03b58c44u, 0x0be37438u, 0x0829f771u, 0x1660139eu, 0x0027fd91u)     // R^2 mod r
// K*r with limbs 0..7 >= 2^29 - 1: limb-wise K - b never borrows for a normalised b < 2r (M4) / < 6r (M8)
FR9_TABLE(fr9_m4, 0x20000004u, 0x3fffffdfu, 0x3e5bfefeu, 0x2d2017feu, 0x360154eeu, 0x30101342u, 0x3483339cu, 0x2994cebdu, 0x01cfb69cu)
FR9_TABLE(fr9_m8, 0x20000008u, 0x3fffffbfu, 0x3cb7fdfeu, 0x3a402ffeu, 0x2c02a9ddu, 0x20202686u, 0x2906673au, 0x33299d7cu, 0x039f6d39u)

KZG_DEV void fr9_zero(fr9_t& r) {
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = 0;
}
KZG_DEV void fr9_one(fr9_t& r) {
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = fr9_one_c(i);
}
// lazy sum / difference: limbs add
KZG_DEV void fr9_add(fr9_t& r, const fr9_t& a, const fr9_t& b) {
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = a.l[i] + b.l[i];
}
KZG_DEV void fr9_sub4(fr9_t& r, const fr9_t& a, const fr9_t& b) {   // b normalised, < 2r; value grows by 4r
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = a.l[i] + (fr9_m4(i) - b.l[i]);
}
// carry propagation: same value, limbs 0..7 < 2^29
KZG_DEV void fr9_norm(fr9_t& r, const fr9_t& a) {
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const uint32_t v = a.l[i] + c;
        r.l[i] = v & FR9_MASK;
        c = v >> 29;
    }
    r.l[8] = a.l[8] + c;
}
// Montgomery product, product scanning, one 64-bit accumulator per column.  Bounds: header.
KZG_DEV void fr9_mul(fr9_t& r, const fr9_t& a, const fr9_t& b) {
    uint32_t q[9];
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 17; k++) {
#pragma unroll
        for (int i = 0; i < 9; i++) {
            const int j = k - i;
            if (j >= 0 && j < 9) acc += (uint64_t)a.l[i] * b.l[j];
        }
#pragma unroll
        for (int i = 0; i < 9; i++) {
            const int j = k - i;
            if (i < k && j >= 1 && j < 9) acc += (uint64_t)q[i] * fr9_r(j);
        }
        if (k < 9) {
            q[k] = (0u - (uint32_t)acc) & FR9_MASK;
            acc += q[k];                              // q * r_0, r_0 = 1: the low 29 bits are now zero
        } else {
            r.l[k - 9] = (uint32_t)acc & FR9_MASK;
        }
        acc >>= 29;
    }
    r.l[8] = (uint32_t)acc;
}
// N class (normalised, < 2r) -> canonical [0, r)
KZG_DEV void fr9_canon(fr9_t& r, const fr9_t& a) {
    uint32_t d[9];
    int32_t br = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const int32_t v = (int32_t)a.l[i] - (int32_t)fr9_r(i) + br;
        br = v >> 29;  // 0 or -1
        d[i] = (i < 8) ? ((uint32_t)v & FR9_MASK) : (uint32_t)v;
    }
    const bool neg = br < 0;  // a < r
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = neg ? a.l[i] : d[i];
}
// Cheap partial reduction: any normalised value < 64r -> the same residue below 2r, WITHOUT a product.  The quotient is
// estimated from limb 8 (bits >= 232): q = floor(l8 * floor(2^52 / (r8 + 1)) / 2^52) with r8 = r >> 232 never exceeds
// floor(v / r) and falls short of it by less than 1 (checked over 2 x 10^5 values up to 64r incl. the edges k r - 1), then
// v - q r with signed carries.  ~45 instructions against ~215 for the product by R mod r.
KZG_DEV void fr9_reduce_approx(fr9_t& r, const fr9_t& a) {
    constexpr uint64_t M = (1ull << 52) / (uint64_t)(fr9_r(8) + 1);
    const uint32_t q = (uint32_t)(((uint64_t)a.l[8] * M) >> 52);
    int64_t carry = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const int64_t acc = (int64_t)a.l[i] - (int64_t)((uint64_t)q * fr9_r(i)) + carry;
        r.l[i] = (i < 8) ? ((uint32_t)acc & FR9_MASK) : (uint32_t)acc;
        carry = acc >> 29;
    }
}
// any lazy value (limbs < 2^31, value < 64r) -> canonical, same residue.  No product: carry propagation, the estimated
// quotient of fr9_reduce_approx (exact or one short: the result is < 2r), one conditional subtraction -- ~100 instructions
// where the product by R mod r that used to do this took ~245.
KZG_DEV void fr9_reduce(fr9_t& r, const fr9_t& a) {
#ifdef KZG_FR_REDUCE_BY_PRODUCT
    fr9_t one, t;
    fr9_one(one);
    fr9_mul(t, a, one);
    fr9_canon(r, t);
#else
    fr9_t t;
    fr9_norm(t, a);
    fr9_reduce_approx(t, t);
    fr9_canon(r, t);
#endif
}
KZG_DEV void fr9_to_mont(fr9_t& r, const fr9_t& a_canon_int) {
    fr9_t r2, t;
#pragma unroll
    for (int i = 0; i < 9; i++) r2.l[i] = fr9_r2(i);
    fr9_mul(t, a_canon_int, r2);
    fr9_canon(r, t);
}
KZG_DEV void fr9_from_mont(fr9_t& r, const fr9_t& a_lazy) {  // -> canonical integer
    fr9_t one, t;
    fr9_zero(one);
    one.l[0] = 1;
    fr9_mul(t, a_lazy, one);
    fr9_canon(r, t);
}

// ---- HBM / wire format: 8 x u32 little-endian words of a value < 2^256
KZG_DEV void fr9_from_words(fr9_t& r, const uint32_t* w) {
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const int bit = 29 * i, wi = bit >> 5, sh = bit & 31;
        uint32_t v = w[wi] >> sh;
        if (sh > 3 && wi + 1 < 8) v |= w[wi + 1] << (32 - sh);
        r.l[i] = (i < 8) ? (v & FR9_MASK) : v;
    }
}
KZG_DEV void fr9_to_words(uint32_t* w, const fr9_t& a) {  // a normalised, value < 2^256
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const int bit = 32 * k, li = bit / 29, sh = bit - 29 * li;  // word k starts inside limb li at bit sh
        uint32_t v = a.l[li] >> sh;
        if (li + 1 < 9) v |= a.l[li + 1] << (29 - sh);
        if (li + 2 < 9 && 58 - sh < 32) v |= a.l[li + 2] << (58 - sh);
        w[k] = v;
    }
}
KZG_DEV void fr9_load(fr9_t& v, const uint32_t* p) {
    const uint4* q = reinterpret_cast<const uint4*>(p);
    const uint4 a = q[0], b = q[1];
    const uint32_t w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    fr9_from_words(v, w);
}
KZG_DEV void fr9_store(uint32_t* p, const fr9_t& v_canon) {
    uint32_t w[8];
    fr9_to_words(w, v_canon);
    uint4* q = reinterpret_cast<uint4*>(p);
    q[0] = make_uint4(w[0], w[1], w[2], w[3]);
    q[1] = make_uint4(w[4], w[5], w[6], w[7]);
}
// words of a canonical integer >= r ?
KZG_DEV bool fr_words_ge_r(const uint32_t* w) {
    constexpr uint32_t rw[8] = {0x00000001u, 0xffffffffu, 0xfffe5bfeu, 0x53bda402u,
                                0x09a1d805u, 0x3339d808u, 0x299d7d48u, 0x73eda753u};
    uint32_t rm[8];
#pragma unroll
    for (int i = 0; i < 8; i++) rm[i] = rw[i];
    return bi_ge<8>(w, rm);
}
