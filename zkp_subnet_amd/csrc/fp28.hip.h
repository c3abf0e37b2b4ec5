// Fp (BLS12-381 base field) on 14 unsaturated 28-bit limbs -- the MSM's working representation.
//
// Why not 12 saturated 32-bit limbs (field.hip.h): on gfx950 a v_mad_u64_u32 costs about the same issue time as
// any other VOP3 instruction (scripts/ubench/valu_rates.hip: 2.3 ns vs 2.0 ns per wave-instruction per SIMD), so
// the cost of a field product is its INSTRUCTION COUNT.  Saturated limbs need a carry instruction per limb product
// (857 instructions / product); with 28-bit limbs a 64-bit column accumulator absorbs 14 products of up to 60
// bits with no carry handling at all: 2 x 196 mads + ~100 bookkeeping, and additions / subtractions are 14
// independent 32-bit adds with no carry chain (gfx950 pads every VCC carry hop with two wait states).
//
// Representation classes (value is taken mod p; R = 2^392):
//   N     : limbs < 2^28 (limb 13 may hold the excess), value < 2p      -- every product output
//   loose : limbs < 2^30, value < 32p                                   -- legal product INPUT
// A product of two loose inputs has columns < 14 * 2^60 + 14 * 2^56 + carry < 2^64 and value
// a*b/R + p < (2^386)^2 / 2^392 + p < 2p.  Lazy sums / differences of N values stay loose; fp_norm restores 28-bit
// limbs without changing the value.  Subtraction adds a multiple of p whose limbs dominate the subtrahend's:
// M4 for subtrahends < 2p, M8 for < 6p, M16 for < 14p (all with normalised limbs).
#pragma once
#include "bigint.hip.h"

struct alignas(8) fp_t {
    uint32_t l[14];
};

#define FP28_MASK 0x0fffffffu
#define FP28_PINV 0x0ffcfffdu  // -p^-1 mod 2^28

#define FP28_TABLE(name, ...)                                  \
    __host__ __device__ constexpr uint32_t name(int i) {       \
        constexpr uint32_t m[14] = {__VA_ARGS__};              \
        return m[i];                                           \
    }
FP28_TABLE(fp28_p, 0x0fffaaabu, 0x0fefffffu, 0x03ffffb9u, 0x0fffeb15u, 0x06241eabu, 0x0a0f6b0fu, 0x0f6730d2u,
           0x0f38512bu, 0x04774b84u, 0x04bacd76u, 0x0ba7b643u, 0x0e69a4b1u, 0x01ea397fu, 0x0001a011u)
FP28_TABLE(fp28_one, 0x0347fcb8u, 0x0d800000u, 0x0002b119u, 0x00cde6d2u, 0x0c7212e0u, 0x083a2090u, 0x0037669fu,
           0x0da0f73eu, 0x09b09b42u, 0x01297bb0u, 0x0515d98fu, 0x0012ca7cu, 0x0659fcfau, 0x0000577au)  // R mod p
FP28_TABLE(fp28_r2, 0x010370edu, 0x06d1c345u, 0x0e243d62u, 0x0ec45c53u, 0x03b1d65au, 0x0093317du, 0x0b4f36a0u,
           0x05d74088u, 0x0c10ea72u, 0x0865d118u, 0x07320a75u, 0x0fd5cd50u, 0x0cc8a759u, 0x0000c8d4u)  // R^2 mod p
// K*p with limbs redistributed so that limb i >= 2^28 - 1 for i < 13 (no per-limb borrow against normalised limbs)
FP28_TABLE(fp28_m4, 0x1ffeaaacu, 0x1fbffffeu, 0x1ffffee6u, 0x1fffac53u, 0x18907aaeu, 0x183dac3cu, 0x1d9cc349u,
           0x1ce144aeu, 0x11dd2e12u, 0x12eb35d8u, 0x1e9ed90cu, 0x19a692c5u, 0x17a8e5feu, 0x00068043u)
FP28_TABLE(fp28_m8, 0x1ffd5558u, 0x1f7ffffeu, 0x1ffffdceu, 0x1fff58a8u, 0x1120f55eu, 0x107b587au, 0x1b398694u,
           0x19c2895eu, 0x13ba5c26u, 0x15d66bb1u, 0x1d3db219u, 0x134d258cu, 0x1f51cbfeu, 0x000d0087u)
FP28_TABLE(fp28_m16, 0x1ffaaab0u, 0x1efffffeu, 0x1ffffb9eu, 0x1ffeb152u, 0x1241eabeu, 0x10f6b0f5u, 0x16730d29u,
           0x138512beu, 0x1774b84eu, 0x1bacd763u, 0x1a7b6433u, 0x169a4b1au, 0x1ea397fdu, 0x001a0110u)

KZG_DEV void fp_zero(fp_t& r) {
#pragma unroll
    for (int i = 0; i < 14; i++) r.l[i] = 0;
}
KZG_DEV void fp_one(fp_t& r) {
#pragma unroll
    for (int i = 0; i < 14; i++) r.l[i] = fp28_one(i);
}
// exact all-zero limbs: the encoding of "absent" (infinity marker); NOT a test for value == 0 mod p
KZG_DEV bool fp_limbs_zero(const fp_t& a) {
    uint32_t t = 0;
#pragma unroll
    for (int i = 0; i < 14; i++) t |= a.l[i];
    return t == 0;
}
// value == 0 mod p for an N-class value (product output): the only representatives are 0 and p
KZG_DEV bool fp_is_zero_n(const fp_t& a) {
    uint32_t z = 0, e = 0;
#pragma unroll
    for (int i = 0; i < 14; i++) {
        z |= a.l[i];
        e |= a.l[i] ^ fp28_p(i);
    }
    return z == 0 || e == 0;
}
KZG_DEV void fp_select(fp_t& r, const fp_t& a, const fp_t& b, bool take_b) {
#pragma unroll
    for (int i = 0; i < 14; i++) r.l[i] = take_b ? b.l[i] : a.l[i];
}
// carry propagation: same value, limbs 0..12 < 2^28
KZG_DEV void fp_norm(fp_t& r, const fp_t& a) {
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < 13; i++) {
        uint32_t v = a.l[i] + c;
        r.l[i] = v & FP28_MASK;
        c = v >> 28;
    }
    r.l[13] = a.l[13] + c;
}
// lazy sum: limbs add
KZG_DEV void fp_add(fp_t& r, const fp_t& a, const fp_t& b) {
#pragma unroll
    for (int i = 0; i < 14; i++) r.l[i] = a.l[i] + b.l[i];
}
KZG_DEV void fp_dbl(fp_t& r, const fp_t& a) {
#pragma unroll
    for (int i = 0; i < 14; i++) r.l[i] = a.l[i] << 1;
}
// r = a - b as a + K*p - b; b must have normalised limbs and value < 2p (M4) / 6p (M8) / 14p (M16)
KZG_DEV void fp_sub4(fp_t& r, const fp_t& a, const fp_t& b) {
#pragma unroll
    for (int i = 0; i < 14; i++) r.l[i] = a.l[i] + (fp28_m4(i) - b.l[i]);
}
KZG_DEV void fp_sub8(fp_t& r, const fp_t& a, const fp_t& b) {
#pragma unroll
    for (int i = 0; i < 14; i++) r.l[i] = a.l[i] + (fp28_m8(i) - b.l[i]);
}
KZG_DEV void fp_sub16(fp_t& r, const fp_t& a, const fp_t& b) {
#pragma unroll
    for (int i = 0; i < 14; i++) r.l[i] = a.l[i] + (fp28_m16(i) - b.l[i]);
}
// -a for a canonical value (limbs < 2^28, value < p): p - a, 0 stays 0; result canonical
KZG_DEV void fp_neg_canon(fp_t& r, const fp_t& a) {
    const bool z = fp_limbs_zero(a);
    int32_t br = 0;
#pragma unroll
    for (int i = 0; i < 14; i++) {
        int32_t d = (int32_t)fp28_p(i) - (int32_t)a.l[i] + br;
        br = d >> 28;  // 0 or -1 (arithmetic)
        r.l[i] = z ? 0u : ((uint32_t)d & FP28_MASK);
    }
}

// (A/B knob -DKZG_FP_ONE_ACC: ONE accumulator per column instead of two -- 28 fewer 64-bit additions per product, a longer
// dependent chain of mads)
#ifdef KZG_FP_ONE_ACC
#define FP_ACC1 acc0
#define FP_ACC_SUM acc0
#else
#define FP_ACC1 acc1
#define FP_ACC_SUM (acc0 + acc1)
#endif
// Montgomery product, product scanning (FIPS), one 64-bit accumulator pair per column; inputs loose, output N.
KZG_DEV void fp_mul_inline(fp_t& r, const fp_t& a, const fp_t& b) {
    uint32_t q[14];
    uint64_t carry = 0;
#pragma unroll
    for (int k = 0; k < 28; k++) {
        uint64_t acc0 = carry, acc1 = 0;
#pragma unroll
        for (int i = 0; i < 14; i++) {
            const int j = k - i;
            if (j >= 0 && j < 14) {
                if (i & 1) FP_ACC1 += (uint64_t)a.l[i] * b.l[j];
                else acc0 += (uint64_t)a.l[i] * b.l[j];
            }
        }
#pragma unroll
        for (int i = 0; i < 14; i++) {
            const int j = k - i;
            if (i < k && i < 14 && j >= 0 && j < 14) {
                if (i & 1) acc0 += (uint64_t)q[i] * fp28_p(j);
                else FP_ACC1 += (uint64_t)q[i] * fp28_p(j);
            }
        }
        uint64_t acc = FP_ACC_SUM;
        if (k < 14) {
            q[k] = ((uint32_t)acc * FP28_PINV) & FP28_MASK;
            acc += (uint64_t)q[k] * fp28_p(0);
            carry = acc >> 28;
        } else {
            r.l[k - 14] = (k < 27) ? ((uint32_t)acc & FP28_MASK) : (uint32_t)acc;
            carry = acc >> 28;
        }
    }
}
KZG_DEV void fp_sqr_inline(fp_t& r, const fp_t& a) {
    uint32_t q[14], a2[14];
#pragma unroll
    for (int i = 0; i < 14; i++) a2[i] = a.l[i] << 1;  // < 2^31
    uint64_t carry = 0;
#pragma unroll
    for (int k = 0; k < 28; k++) {
        uint64_t acc0 = carry, acc1 = 0;
#pragma unroll
        for (int i = 0; i < 14; i++) {
            const int j = k - i;
            if (j >= 0 && j < 14 && i < j) {
                if (i & 1) FP_ACC1 += (uint64_t)a.l[i] * a2[j];
                else acc0 += (uint64_t)a.l[i] * a2[j];
            }
        }
        if (!(k & 1) && (k >> 1) < 14) FP_ACC1 += (uint64_t)a.l[k >> 1] * a.l[k >> 1];
#pragma unroll
        for (int i = 0; i < 14; i++) {
            const int j = k - i;
            if (i < k && i < 14 && j >= 0 && j < 14) {
                if (i & 1) acc0 += (uint64_t)q[i] * fp28_p(j);
                else FP_ACC1 += (uint64_t)q[i] * fp28_p(j);
            }
        }
        uint64_t acc = FP_ACC_SUM;
        if (k < 14) {
            q[k] = ((uint32_t)acc * FP28_PINV) & FP28_MASK;
            acc += (uint64_t)q[k] * fp28_p(0);
            carry = acc >> 28;
        } else {
            r.l[k - 14] = (k < 27) ? ((uint32_t)acc & FP28_MASK) : (uint32_t)acc;
            carry = acc >> 28;
        }
    }
}

// (a*b + c*d) / R with ONE Montgomery reduction (inline only: a 4-operand call would not fit the register ABI).
// Column bound: a, b loose (limbs <= 1.5 * 2^29: 14 * 2.25 * 2^58 < 2^63) plus c, d with limbs < 2^29 and < 2^28
// (14 * 2^57 < 2^61) plus the reduction (< 2^60): < 2^63.4.  Value: (ab + cd)/R + p < 2p for a < 10p, b < 18p,
// c < 8p, d < 2p (196 p^2 / R < p/12).
KZG_DEV void fp_mul2_inline(fp_t& r, const fp_t& a, const fp_t& b, const fp_t& c, const fp_t& d) {
    uint32_t q[14];
    uint64_t carry = 0;
#pragma unroll
    for (int k = 0; k < 28; k++) {
        uint64_t acc0 = carry, acc1 = 0;
#pragma unroll
        for (int i = 0; i < 14; i++) {
            const int j = k - i;
            if (j >= 0 && j < 14) {
                acc0 += (uint64_t)a.l[i] * b.l[j];
                FP_ACC1 += (uint64_t)c.l[i] * d.l[j];
            }
        }
#pragma unroll
        for (int i = 0; i < 14; i++) {
            const int j = k - i;
            if (i < k && i < 14 && j >= 0 && j < 14) {
                if (i & 1) acc0 += (uint64_t)q[i] * fp28_p(j);
                else FP_ACC1 += (uint64_t)q[i] * fp28_p(j);
            }
        }
        uint64_t acc = FP_ACC_SUM;
        if (k < 14) {
            q[k] = ((uint32_t)acc * FP28_PINV) & FP28_MASK;
            acc += (uint64_t)q[k] * fp28_p(0);
            carry = acc >> 28;
        } else {
            r.l[k - 14] = (k < 27) ? ((uint32_t)acc & FP28_MASK) : (uint32_t)acc;
            carry = acc >> 28;
        }
    }
}
// 8p - b limb-wise (b: normalised limbs, value < 6p): a non-negative representative of -b with limbs < 2^29
KZG_DEV void fp_neg8(fp_t& r, const fp_t& b) {
#pragma unroll
    for (int i = 0; i < 14; i++) r.l[i] = fp28_m8(i) - b.l[i];
}

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
struct fp_ret {
    u32x4 v0, v1, v2;
    u32x2 v3;
};
KZG_DEV void fp_unvec(fp_t& a, u32x4 a0, u32x4 a1, u32x4 a2, u32x2 a3) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
        a.l[i] = a0[i]; a.l[4 + i] = a1[i]; a.l[8 + i] = a2[i];
    }
    a.l[12] = a3[0]; a.l[13] = a3[1];
}
KZG_DEV fp_ret fp_vec(const fp_t& r) {
    fp_ret o;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        o.v0[i] = r.l[i]; o.v1[i] = r.l[4 + i]; o.v2[i] = r.l[8 + i];
    }
    o.v3[0] = r.l[12]; o.v3[1] = r.l[13];
    return o;
}
#ifndef KZG_FP_MUL_INLINE
// Real function calls (s_swappc) for everything except the hot mixed addition (which inlines fp_mul_inline, see
// g1.hip.h fpm<>): one copy of the product / square per translation unit keeps the many point formulas of the tail
// kernels small; vector-typed arguments stay in VGPRs across the call.
static __device__ __noinline__ fp_ret fp_mul_raw(u32x4 a0, u32x4 a1, u32x4 a2, u32x2 a3, u32x4 b0, u32x4 b1, u32x4 b2,
                                                  u32x2 b3) {
    fp_t a, b, r;
    fp_unvec(a, a0, a1, a2, a3);
    fp_unvec(b, b0, b1, b2, b3);
    fp_mul_inline(r, a, b);
    return fp_vec(r);
}
static __device__ __noinline__ fp_ret fp_sqr_raw(u32x4 a0, u32x4 a1, u32x4 a2, u32x2 a3) {
    fp_t a, r;
    fp_unvec(a, a0, a1, a2, a3);
    fp_sqr_inline(r, a);
    return fp_vec(r);
}
KZG_DEV void fp_mul(fp_t& r, const fp_t& a, const fp_t& b) {
    fp_ret x = fp_vec(a), y = fp_vec(b);
    fp_ret o = fp_mul_raw(x.v0, x.v1, x.v2, x.v3, y.v0, y.v1, y.v2, y.v3);
    fp_unvec(r, o.v0, o.v1, o.v2, o.v3);
}
KZG_DEV void fp_sqr(fp_t& r, const fp_t& a) {
    fp_ret x = fp_vec(a);
    fp_ret o = fp_sqr_raw(x.v0, x.v1, x.v2, x.v3);
    fp_unvec(r, o.v0, o.v1, o.v2, o.v3);
}
#else
KZG_DEV void fp_mul(fp_t& r, const fp_t& a, const fp_t& b) { fp_mul_inline(r, a, b); }
KZG_DEV void fp_sqr(fp_t& r, const fp_t& a) { fp_sqr_inline(r, a); }
#endif

// ---- conversions.  "packed" = 12 x u32 little-endian limbs of a value < 2^384 (HBM format of the window tables)
KZG_DEV void fp_unpack(fp_t& r, const uint32_t* w) {  // 12 words -> 14 limbs of 28 bits
#pragma unroll
    for (int i = 0; i < 14; i++) {
        const int bit = 28 * i, wi = bit >> 5, sh = bit & 31;
        uint32_t lo = w[wi] >> sh;
        uint32_t hi = (sh > 4 && wi + 1 < 12) ? (w[wi + 1] << (32 - sh)) : 0u;
        r.l[i] = (lo | hi) & FP28_MASK;
    }
}
KZG_DEV void fp_pack(uint32_t* w, const fp_t& a) {  // canonical limbs (< 2^28 each, value < 2^384) -> 12 words
#pragma unroll
    for (int k = 0; k < 12; k++) {
        const int bit = 32 * k, li = bit / 28, sh = bit - 28 * li;  // word k starts inside limb li at bit sh
        uint32_t v = a.l[li] >> sh;
        if (li + 1 < 14) v |= a.l[li + 1] << (28 - sh);
        if (li + 2 < 14 && 56 - sh < 32) v |= a.l[li + 2] << (56 - sh);
        w[k] = v;
    }
}
// N-class (< 2p, normalised limbs) -> canonical [0, p)
KZG_DEV void fp_canon(fp_t& r, const fp_t& a) {
    uint32_t d[14];
    int32_t br = 0;
#pragma unroll
    for (int i = 0; i < 14; i++) {
        int32_t v = (int32_t)a.l[i] - (int32_t)fp28_p(i) + br;
        br = v >> 28;
        d[i] = (i < 13) ? ((uint32_t)v & FP28_MASK) : (uint32_t)v;
    }
    const bool neg = br < 0;  // a < p
#pragma unroll
    for (int i = 0; i < 14; i++) r.l[i] = neg ? a.l[i] : d[i];
}
KZG_DEV void fp_to_mont(fp_t& r, const fp_t& a) {  // canonical integer -> Montgomery (N class)
    fp_t r2;
#pragma unroll
    for (int i = 0; i < 14; i++) r2.l[i] = fp28_r2(i);
    fp_mul(r, a, r2);
}
KZG_DEV void fp_from_mont(fp_t& r, const fp_t& a) {  // loose Montgomery -> canonical integer
    fp_t one, t;
    fp_zero(one);
    one.l[0] = 1;
    fp_mul(t, a, one);
    fp_canon(r, t);
}
// loose Montgomery residue -> canonical Montgomery residue in [0, p) (what the window tables hold):
// a * (R mod p) / R = a, reduced to N class by the product, then one conditional subtraction
KZG_DEV void fp_canon_mont(fp_t& r, const fp_t& a_loose) {
    fp_t one, t;
    fp_one(one);
    fp_mul(t, a_loose, one);
    fp_canon(r, t);
}
