// BLS12-381 G1 (y^2 = x^3 + 4 over Fp) for the MSM kernels, on the 28-bit-limb Fp of fp28.hip.h.
//   affine point (HBM)  : x, y canonical Montgomery residues packed as 12 x u32 each, 96 B; all-zero = infinity
//   bucket / sum (XYZZ) : X, Y, ZZ, ZZZ as 14-limb fp_t, 224 B, x = X/ZZ, y = Y/ZZZ; ZZ all-zero limbs = infinity
// Stored-coordinate classes (fp28.hip.h): X normalised limbs, value < 14p; Y normalised, < 6p; ZZ, ZZZ product outputs
// (< 2p).  Every formula below ends inside these classes, so they hold inductively; subtrahends use the multiple
// of p that dominates their class (X: M16, Y: M8, product outputs: M4).
// XYZZ because the bucket update is a *mixed* add (affine table point into a running bucket): 8M + 2S, no inversion
// (EFD madd-2008-s); bucket + bucket is add-2008-s (12M + 2S).
#pragma once
#include "field.hip.h"
#include "fp28.hip.h"
#include "fr29.hip.h"

// HBM form of a table point: one 128-byte line = x, y as 14 canonical 28-bit limbs each (the working representation:
// no unpacking in the hot loop) + 16 bytes of padding.  A 96-byte packed row straddled two 128-B lines for 3 rows in 4.
struct alignas(128) g1_affine_t {
    uint32_t x[14], y[14], pad[4];
};
struct alignas(16) g1_xyzz_t {
    fp_t x, y, zz, zzz;
};
struct g1_aff28 {  // affine point in registers
    fp_t x, y;
};

KZG_DEV bool g1_is_inf(const g1_xyzz_t& p) { return fp_limbs_zero(p.zz); }
KZG_DEV void g1_set_inf(g1_xyzz_t& p) {
    fp_zero(p.x); fp_zero(p.y); fp_zero(p.zz); fp_zero(p.zzz);
}
KZG_DEV bool g1_aff_is_inf(const g1_aff28& p) { return fp_limbs_zero(p.x) && fp_limbs_zero(p.y); }
KZG_DEV void g1_from_aff(g1_xyzz_t& r, const g1_aff28& p) {
    if (g1_aff_is_inf(p)) { g1_set_inf(r); return; }
    r.x = p.x; r.y = p.y; fp_one(r.zz); fp_one(r.zzz);
}
KZG_DEV void g1_neg_aff(g1_aff28& p, bool negate) {
    fp_t ny;
    fp_neg_canon(ny, p.y);
    fp_select(p.y, p.y, ny, negate);
}

// Fp product / square, inlined (INL) or through the shared function call.  Measured on MI355X: with the ten
// products of the hot mixed addition inlined the accumulate kernel runs 16 % faster than through s_swappc calls
// (2.38 vs 2.83 ms at 2^20: call marshalling plus lost scheduling across the calls); the latency-bound tail kernels
// (cooperative add / double) are a few percent faster with the call form (smaller code), so both exist.
template <bool INL>
KZG_DEV void fpm(fp_t& r, const fp_t& a, const fp_t& b) {
    if constexpr (INL) fp_mul_inline(r, a, b);
    else fp_mul(r, a, b);
}
template <bool INL>
KZG_DEV void fps(fp_t& r, const fp_t& a) {
    if constexpr (INL) fp_sqr_inline(r, a);
    else fp_sqr(r, a);
}
// 2*(x, y), affine non-infinity input (EFD mdbl-2008-s-1, a = 0)
template <bool INL = false>
KZG_DEV void g1_dbl_aff(g1_xyzz_t& r, const fp_t& x, const fp_t& y) {
    fp_t U, V, W, S, M, t, u;
    fp_dbl(U, y);                       // < 2p, limbs < 2^29
    fps<INL>(V, U);
    fpm<INL>(W, U, V);
    fpm<INL>(S, x, V);
    fps<INL>(t, x);
    fp_add(M, t, t); fp_add(M, M, t);   // 3x^2: limbs < 3*2^28, value < 6p
    fps<INL>(u, M);
    fp_sub4(u, u, S); fp_sub4(u, u, S);
    fp_norm(r.x, u);                    // < 2p + 8p
    fp_sub16(t, S, r.x);
    fpm<INL>(t, M, t);
    fpm<INL>(u, W, y);
    fp_sub4(t, t, u);
    fp_norm(r.y, t);                    // < 6p
    r.zz = V; r.zzz = W;
}
// r = 2*p (EFD dbl-2008-s-1, a = 0)
template <bool INL = false>
KZG_DEV void g1_dbl(g1_xyzz_t& r, const g1_xyzz_t& p) {
    if (g1_is_inf(p)) { g1_set_inf(r); return; }
    fp_t U, V, W, S, M, t, u, x3;
    fp_dbl(U, p.y);                     // Y < 6p -> < 12p, limbs < 2^29
    fps<INL>(V, U);
    fpm<INL>(W, U, V);
    fpm<INL>(S, p.x, V);
    fps<INL>(t, p.x);
    fp_add(M, t, t); fp_add(M, M, t);
    fps<INL>(u, M);
    fp_sub4(u, u, S); fp_sub4(u, u, S);
    fp_norm(x3, u);
    fp_sub16(t, S, x3);
    fpm<INL>(t, M, t);
    fpm<INL>(u, W, p.y);
    fp_sub4(t, t, u);
    fp_norm(r.y, t);
    r.x = x3;
    fpm<INL>(r.zz, V, p.zz);
    fpm<INL>(r.zzz, W, p.zzz);
}
// acc += (qx, qy): affine, canonical, non-infinity (EFD madd-2008-s: 8M + 2S).  The common path is branch-free:
// an empty accumulator is handled by a select at the end; the only branch is the rare acc == +-q case.
template <bool INL = false>
KZG_DEV void g1_madd(g1_xyzz_t& acc, const fp_t& qx, const fp_t& qy) {
    const bool acc_inf = g1_is_inf(acc);
    fp_t U2, S2, P, R, PP, PPP, Q, RR, t, u, x3, y3, zz3, zzz3;
    fpm<INL>(U2, qx, acc.zz);
    fpm<INL>(S2, qy, acc.zzz);
    fp_sub16(P, U2, acc.x);             // < 18p, limbs < 2^28 + 2^29
    fp_sub8(R, S2, acc.y);              // < 10p
    fps<INL>(PP, P);
    fps<INL>(RR, R);
    if (!acc_inf && fp_is_zero_n(PP)) {  // x coordinates agree: q == acc (double) or q == -acc (cancel)
        if (fp_is_zero_n(RR)) g1_dbl_aff<INL>(acc, qx, qy);  // the INL kernel stays free of calls
        else g1_set_inf(acc);
        return;
    }
    fpm<INL>(PPP, P, PP);
    fpm<INL>(Q, acc.x, PP);
    fp_sub4(t, RR, PPP); fp_sub4(t, t, Q); fp_sub4(t, t, Q);
    fp_norm(x3, t);                     // < 2p + 12p = 14p
    fp_sub16(t, Q, x3);                 // < 18p
    if constexpr (INL) {                // y3 = (R (Q - x3) - Y1 PPP): two products, ONE reduction; result < 2p
        fp_neg8(u, acc.y);
        fp_mul2_inline(y3, R, t, u, PPP);
    } else {
        fpm<INL>(t, R, t);
        fpm<INL>(u, acc.y, PPP);
        fp_sub4(t, t, u);
        fp_norm(y3, t);                 // < 6p
    }
    fpm<INL>(zz3, acc.zz, PP);
    fpm<INL>(zzz3, acc.zzz, PPP);
    fp_t one;
    fp_one(one);
    fp_select(acc.x, x3, qx, acc_inf);
    fp_select(acc.y, y3, qy, acc_inf);
    fp_select(acc.zz, zz3, one, acc_inf);
    fp_select(acc.zzz, zzz3, one, acc_inf);
}
template <bool INL = false>
KZG_DEV void g1_madd_checked(g1_xyzz_t& acc, const g1_aff28& q) {
    if (g1_aff_is_inf(q)) return;
    g1_madd<INL>(acc, q.x, q.y);
}
// r = p + q (EFD add-2008-s: 12M + 2S) with the exceptional cases
template <bool INL = false>
KZG_DEV void g1_add(g1_xyzz_t& r, const g1_xyzz_t& p, const g1_xyzz_t& q) {
    if (g1_is_inf(p)) { r = q; return; }
    if (g1_is_inf(q)) { r = p; return; }
    fp_t U1, U2, S1, S2, P, R, PP, PPP, Q, RR, t, u, x3;
    fpm<INL>(U1, p.x, q.zz);
    fpm<INL>(U2, q.x, p.zz);
    fpm<INL>(S1, p.y, q.zzz);
    fpm<INL>(S2, q.y, p.zzz);
    fp_sub4(P, U2, U1);
    fp_sub4(R, S2, S1);
    fps<INL>(PP, P);
    fps<INL>(RR, R);
    if (fp_is_zero_n(PP)) {
        if (fp_is_zero_n(RR)) { g1_dbl<INL>(r, p); return; }   // kept inline: a real call here costs +11 % (A/B)
        g1_set_inf(r);
        return;
    }
    fpm<INL>(PPP, P, PP);
    fpm<INL>(Q, U1, PP);
    fp_sub4(t, RR, PPP); fp_sub4(t, t, Q); fp_sub4(t, t, Q);
    fp_norm(x3, t);
    fp_sub16(t, Q, x3);
    if constexpr (INL) {                // y3 = R (Q - x3) - S1 PPP: two products, ONE reduction (as in g1_madd); < 2p
        fp_neg8(u, S1);
        fp_mul2_inline(r.y, R, t, u, PPP);
    } else {
        fpm<INL>(t, R, t);
        fpm<INL>(u, S1, PPP);
        fp_sub4(t, t, u);
        fp_norm(r.y, t);
    }
    r.x = x3;
    fpm<INL>(t, p.zz, q.zz); fpm<INL>(r.zz, t, PP);
    fpm<INL>(t, p.zzz, q.zzz); fpm<INL>(r.zzz, t, PPP);
}

// ---- inversion.  One lane's Fermat ladder (381 squarings) is > 1 ms of pure latency, and a bit-by-bit binary
// Euclid on multi-limb values is ~0.2 ms of carry chains, so the single inversion that ends every MSM uses the
// word-approximation binary GCD (Pornin, "Optimized Binary GCD for Modular Inversion", 2020) on the 28-bit limbs:
// 28 outer rounds; each runs 28 binary-GCD steps on 58-bit approximations (low limb exact + top 30 bits) that
// only track a 2x2 factor matrix (f0 g0 / f1 g1, |.| <= 2^28), then applies the matrix to the full values with
// signed 64-bit multiply-accumulates -- dividing by 2^28 is dropping one limb, and no carry flag is ever used.
// Invariants a*K = y*u, b*K = y*v (mod p) with K = R^2: for a Montgomery residue y = xR the result v = R^2/y = x^-1 R
// is again a Montgomery residue.  Prototyped limb-exactly in Python before transcription (3000 random + edge values).
KZG_DEV int fp28_bitlen(const uint32_t* l) {
    int n = 0;
#pragma unroll
    for (int i = 0; i < 14; i++) n = l[i] ? 28 * i + (32 - __builtin_clz(l[i])) : n;
    return n;
}
// low limb (28 bits, exact) | top 30 bits of the n-bit value << 28          (n > 58)
KZG_DEV uint64_t fp28_approx(const uint32_t* l, int n) {
    const int sh = n - 30, li = sh / 28, off = sh - 28 * li;
    uint64_t w0 = 0, w1 = 0, w2 = 0;
#pragma unroll
    for (int i = 0; i < 14; i++) {
        w0 = (i == li) ? l[i] : w0;
        w1 = (i == li + 1) ? l[i] : w1;
        w2 = (i == li + 2) ? l[i] : w2;
    }
    // bits [off, off + 30) of w0 | w1 << 28 | w2 << 56 : never needs more than 64 bits of the 84
    const uint64_t lo = (w0 | (w1 << 28)) >> off;                    // 56 - off valid bits
    const uint64_t hi = (off > 26) ? (w2 << (56 - off)) : 0;         // only when 56 - off < 30
    return (uint64_t)l[0] | (((lo | hi) & 0x3fffffffull) << 28);
}
KZG_DEV uint64_t fp28_small_value(const uint32_t* l) {  // value < 2^58 -> exact
    return (uint64_t)l[0] | ((uint64_t)l[1] << 28) | ((uint64_t)(l[2] & 3u) << 56);
}
// (x f + y g) / 2^28, exact; x, y unsigned 14-limb values; returns |result| limbs and whether it was negative
KZG_DEV bool fp28_lincomb_exact(uint32_t* out, const uint32_t* x, const uint32_t* y, int32_t f, int32_t g) {
    int64_t carry = 0;
    uint32_t t[14];
#pragma unroll
    for (int i = 0; i < 14; i++) {
        const int64_t acc = carry + (int64_t)x[i] * f + (int64_t)y[i] * g;
        t[i] = (uint32_t)acc & FP28_MASK;
        carry = acc >> 28;
    }
    // value / 2^28 = t[1..13] plus the signed top `carry` at limb 13
    const bool neg = carry < 0;
    uint32_t c = neg ? 1u : 0u;
    const uint32_t flip = neg ? FP28_MASK : 0u;
#pragma unroll
    for (int i = 0; i < 13; i++) {
        const uint32_t v = (t[i + 1] ^ flip) + c;
        out[i] = v & FP28_MASK;
        c = v >> 28;
    }
    const int64_t top = neg ? (-carry - 1 + (int64_t)c) : carry;
    out[13] = (uint32_t)top;
    return neg;
}
// (x f + y g) / 2^28 mod p into [0, p); x, y in [0, p)
KZG_DEV void fp28_lincomb_mod(uint32_t* out, const uint32_t* x, const uint32_t* y, int32_t f, int32_t g) {
    int64_t carry = 0;
    uint32_t t[14];
#pragma unroll
    for (int i = 0; i < 14; i++) {
        const int64_t acc = carry + (int64_t)x[i] * f + (int64_t)y[i] * g;
        t[i] = (uint32_t)acc & FP28_MASK;
        carry = acc >> 28;
    }
    const int64_t t14 = carry;
    const uint32_t q = (t[0] * FP28_PINV) & FP28_MASK;
    carry = 0;
    int32_t r[14];
#pragma unroll
    for (int i = 0; i < 14; i++) {
        const int64_t acc = carry + (int64_t)t[i] + (int64_t)((uint64_t)q * fp28_p(i));
        if (i > 0) r[i - 1] = (int32_t)((uint32_t)acc & FP28_MASK);
        carry = acc >> 28;
    }
    r[13] = (int32_t)(t14 + carry);  // signed top limb; value in (-2p, 3p)
#pragma unroll
    for (int pass = 0; pass < 2; pass++) {
        if (r[13] < 0) {
            int32_t cc = 0;
#pragma unroll
            for (int i = 0; i < 13; i++) {
                const int32_t v = r[i] + (int32_t)fp28_p(i) + cc;
                r[i] = v & (int32_t)FP28_MASK;
                cc = v >> 28;
            }
            r[13] += (int32_t)fp28_p(13) + cc;
        }
    }
#pragma unroll
    for (int pass = 0; pass < 2; pass++) {
        int32_t d[14], br = 0;
#pragma unroll
        for (int i = 0; i < 13; i++) {
            const int32_t v = r[i] - (int32_t)fp28_p(i) + br;
            d[i] = v & (int32_t)FP28_MASK;
            br = v >> 28;
        }
        d[13] = r[13] - (int32_t)fp28_p(13) + br;
        if (d[13] >= 0) {
#pragma unroll
            for (int i = 0; i < 14; i++) r[i] = d[i];
        }
    }
#pragma unroll
    for (int i = 0; i < 14; i++) out[i] = (uint32_t)r[i];
}
// a: any loose Montgomery residue != 0 mod p; r: canonical Montgomery residue of the inverse
KZG_DEV void fp_inv(fp_t& r, const fp_t& a_in) {
    fp_t ac;
    fp_canon_mont(ac, a_in);
    if (fp_limbs_zero(ac)) { fp_zero(r); return; }
    uint32_t a[14], b[14], u[14], v[14];
#pragma unroll
    for (int i = 0; i < 14; i++) {
        a[i] = ac.l[i];
        b[i] = fp28_p(i);
        u[i] = fp28_r2(i);
        v[i] = 0;
    }
    for (int round = 0; round < 28; round++) {  // ceil((2*381 - 1) / 28)
        const int na = fp28_bitlen(a), nb = fp28_bitlen(b);
        const int n = na > nb ? na : nb;
        uint64_t xa, xb;
        if (n <= 58) {
            xa = fp28_small_value(a);
            xb = fp28_small_value(b);
        } else {
            xa = fp28_approx(a, n);
            xb = fp28_approx(b, n);
        }
        int32_t f0 = 1, g0 = 0, f1 = 0, g1 = 1;
        for (int j = 0; j < 28; j++) {
            const bool odd = xa & 1u;
            const bool sw = odd && xa < xb;
            const uint64_t ta = sw ? xb : xa, tb = sw ? xa : xb;
            const int32_t tf0 = sw ? f1 : f0, tg0 = sw ? g1 : g0, tf1 = sw ? f0 : f1, tg1 = sw ? g0 : g1;
            xa = (odd ? ta - tb : ta) >> 1;
            xb = tb;
            f0 = odd ? tf0 - tf1 : tf0;
            g0 = odd ? tg0 - tg1 : tg0;
            f1 = tf1 << 1;
            g1 = tg1 << 1;
        }
        uint32_t na_l[14], nb_l[14], nu[14], nv[14];
        if (fp28_lincomb_exact(na_l, a, b, f0, g0)) { f0 = -f0; g0 = -g0; }
        if (fp28_lincomb_exact(nb_l, a, b, f1, g1)) { f1 = -f1; g1 = -g1; }
        fp28_lincomb_mod(nu, u, v, f0, g0);
        fp28_lincomb_mod(nv, u, v, f1, g1);
#pragma unroll
        for (int i = 0; i < 14; i++) { a[i] = na_l[i]; b[i] = nb_l[i]; u[i] = nu[i]; v[i] = nv[i]; }
    }
#pragma unroll
    for (int i = 0; i < 14; i++) r.l[i] = v[i];  // a == 0, b == 1: v = R^2 / y
}
// Fermat ladder a^(p-2): for the batch normalisations where thousands of lanes invert at once (no divergence)
KZG_DEV void fp_inv_fermat(fp_t& r, const fp_t& a) {
    fp_t acc;
    fp_one(acc);
    for (int i = 380; i >= 0; i--) {
        fp_sqr(acc, acc);
        uint32_t w = 0;
#pragma unroll
        for (int k = 0; k < 12; k++) w = (k == (i >> 5)) ? (k == 0 ? FpParams::mod(0) - 2u : FpParams::mod(k)) : w;
        if ((w >> (i & 31)) & 1u) fp_mul(acc, acc, a);
    }
    r = acc;
}
// affine (canonical Montgomery, in registers) from XYZZ; infinity -> zeros
KZG_DEV void g1_to_aff(g1_aff28& r, const g1_xyzz_t& p) {
    if (g1_is_inf(p)) { fp_zero(r.x); fp_zero(r.y); return; }
    fp_t i, t;
    fp_mul(t, p.zz, p.zzz);
    fp_inv(i, t);
    fp_mul(t, i, p.zzz);  // 1/ZZ
    fp_mul(t, p.x, t);
    fp_canon(r.x, t);
    fp_mul(t, i, p.zz);   // 1/ZZZ
    fp_mul(t, p.y, t);
    fp_canon(r.y, t);
}

// ---- HBM formats
KZG_DEV void g1_load_aff(g1_aff28& p, const g1_affine_t* src) {
    const uint4* q = reinterpret_cast<const uint4*>(src);
    uint32_t w[28];
#pragma unroll
    for (int i = 0; i < 7; i++) {
        uint4 v = q[i];
        w[4 * i] = v.x; w[4 * i + 1] = v.y; w[4 * i + 2] = v.z; w[4 * i + 3] = v.w;
    }
#pragma unroll
    for (int i = 0; i < 14; i++) { p.x.l[i] = w[i]; p.y.l[i] = w[14 + i]; }
}
KZG_DEV void g1_store_aff(g1_affine_t* dst, const g1_aff28& p) {  // p canonical
    uint32_t w[28];
#pragma unroll
    for (int i = 0; i < 14; i++) { w[i] = p.x.l[i]; w[14 + i] = p.y.l[i]; }
    uint4* q = reinterpret_cast<uint4*>(dst);
#pragma unroll
    for (int i = 0; i < 7; i++) q[i] = make_uint4(w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]);
}
KZG_DEV void store_xyzz(g1_xyzz_t* dst, const g1_xyzz_t& p) {
    uint4* q = reinterpret_cast<uint4*>(dst);
    const uint32_t* w = reinterpret_cast<const uint32_t*>(&p);
    uint32_t t[56];
    const fp_t* f[4] = {&p.x, &p.y, &p.zz, &p.zzz};
#pragma unroll
    for (int k = 0; k < 4; k++)
#pragma unroll
        for (int i = 0; i < 14; i++) t[14 * k + i] = f[k]->l[i];
    (void)w;
#pragma unroll
    for (int i = 0; i < 14; i++) q[i] = make_uint4(t[4 * i], t[4 * i + 1], t[4 * i + 2], t[4 * i + 3]);
}
KZG_DEV void load_xyzz(g1_xyzz_t& p, const g1_xyzz_t* src) {
    const uint4* q = reinterpret_cast<const uint4*>(src);
    uint32_t t[56];
#pragma unroll
    for (int i = 0; i < 14; i++) {
        uint4 v = q[i];
        t[4 * i] = v.x; t[4 * i + 1] = v.y; t[4 * i + 2] = v.z; t[4 * i + 3] = v.w;
    }
    fp_t* f[4] = {&p.x, &p.y, &p.zz, &p.zzz};
#pragma unroll
    for (int k = 0; k < 4; k++)
#pragma unroll
        for (int i = 0; i < 14; i++) f[k]->l[i] = t[14 * k + i];
}

// ---- byte codecs (big-endian wire <-> little-endian limbs)
KZG_DEV uint32_t bswap32(uint32_t v) { return __builtin_bswap32(v); }
template <int N>
KZG_DEV void limbs_from_be(uint32_t* l, const uint8_t* be) {  // be: 4N bytes, 4-byte aligned
    const uint32_t* w = reinterpret_cast<const uint32_t*>(be);
#pragma unroll
    for (int i = 0; i < N; i++) l[i] = bswap32(w[N - 1 - i]);
}
template <int N>
KZG_DEV void limbs_to_be(uint8_t* be, const uint32_t* l) {
    uint32_t* w = reinterpret_cast<uint32_t*>(be);
#pragma unroll
    for (int i = 0; i < N; i++) w[N - 1 - i] = bswap32(l[i]);
}
// 48 big-endian bytes of a canonical integer < p  <->  Montgomery residue
KZG_DEV bool fp_from_be48(fp_t& r, const uint8_t* be) {  // returns false when the integer is >= p
    uint32_t w[12], pm[12];
    limbs_from_be<12>(w, be);
#pragma unroll
    for (int i = 0; i < 12; i++) pm[i] = FpParams::mod(i);
    const bool ok = !bi_ge<12>(w, pm);
    fp_t a;
    fp_unpack(a, w);
    fp_to_mont(r, a);
    fp_canon(r, r);
    return ok;
}
KZG_DEV void fp_to_be48(uint8_t* be, const fp_t& a_mont) {
    fp_t c;
    fp_from_mont(c, a_mont);
    uint32_t w[12];
    fp_pack(w, c);
    limbs_to_be<12>(be, w);
}
// ZCash 48-byte compressed encoding of an affine canonical-Montgomery point
KZG_DEV void g1_compress(uint8_t* out48, const g1_aff28& p) {
    if (g1_aff_is_inf(p)) {
        uint32_t* w = reinterpret_cast<uint32_t*>(out48);
#pragma unroll
        for (int i = 0; i < 12; i++) w[i] = 0;
        out48[0] = 0xC0;
        return;
    }
    fp_t yc;
    fp_from_mont(yc, p.y);
    uint32_t y[12], t[12], pm[12];
    fp_pack(y, yc);
    // y > (p-1)/2  <=>  2y >= p (p odd)
    uint32_t c = bi_add<12>(t, y, y);
#pragma unroll
    for (int i = 0; i < 12; i++) pm[i] = FpParams::mod(i);
    const bool larger = c || bi_ge<12>(t, pm);
    fp_to_be48(out48, p.x);
    out48[0] |= larger ? 0xA0 : 0x80;
}
