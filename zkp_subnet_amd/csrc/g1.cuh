// BLS12-381 G1 (y^2 = x^3 + 4 over Fp) for the MSM kernels.
//   affine point  : (x, y) Montgomery form, 96 B, 16-B aligned; (0,0) encodes infinity ((0,0) is off-curve)
//   bucket / sum  : extended Jacobian "XYZZ" (X, Y, ZZ, ZZZ), x = X/ZZ, y = Y/ZZZ, ZZ == 0 encodes infinity.
// XYZZ is chosen because the bucket update is a *mixed* add (affine SRS point into a running bucket):
// 8M + 2S with no field inversion and no Z bookkeeping (EFD madd-2008-s); bucket+bucket is 12M + 2S.
// The Fp product is a real function call (s_swappc) by default: one 6.6 KB copy instead of ten inlined copies
// per point addition keeps the hot loop inside the instruction cache.
#pragma once
#include "field.cuh"

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
struct fp_ret {
    u32x4 v0, v1, v2;
};

#ifndef KZG_FP_MUL_INLINE
// operands travel in VGPRs (vector-typed arguments stay in registers under the AMDGPU calling convention,
// a 48-byte struct by value would go through scratch)
static __device__ __noinline__ fp_ret fp_mul_raw(u32x4 a0, u32x4 a1, u32x4 a2, u32x4 b0, u32x4 b1, u32x4 b2) {
    fp_t a, b, r;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        a.l[i] = a0[i]; a.l[4 + i] = a1[i]; a.l[8 + i] = a2[i];
        b.l[i] = b0[i]; b.l[4 + i] = b1[i]; b.l[8 + i] = b2[i];
    }
    f_mul(r, a, b);
    fp_ret o;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        o.v0[i] = r.l[i]; o.v1[i] = r.l[4 + i]; o.v2[i] = r.l[8 + i];
    }
    return o;
}
KZG_DEV void fp_mul(fp_t& r, const fp_t& a, const fp_t& b) {
    u32x4 a0 = {a.l[0], a.l[1], a.l[2], a.l[3]}, a1 = {a.l[4], a.l[5], a.l[6], a.l[7]},
          a2 = {a.l[8], a.l[9], a.l[10], a.l[11]};
    u32x4 b0 = {b.l[0], b.l[1], b.l[2], b.l[3]}, b1 = {b.l[4], b.l[5], b.l[6], b.l[7]},
          b2 = {b.l[8], b.l[9], b.l[10], b.l[11]};
    fp_ret o = fp_mul_raw(a0, a1, a2, b0, b1, b2);
#pragma unroll
    for (int i = 0; i < 4; i++) {
        r.l[i] = o.v0[i]; r.l[4 + i] = o.v1[i]; r.l[8 + i] = o.v2[i];
    }
}
#else
KZG_DEV void fp_mul(fp_t& r, const fp_t& a, const fp_t& b) { f_mul(r, a, b); }
#endif
KZG_DEV void fp_sqr(fp_t& r, const fp_t& a) { fp_mul(r, a, a); }
KZG_DEV void fp_add(fp_t& r, const fp_t& a, const fp_t& b) { f_add(r, a, b); }
KZG_DEV void fp_sub(fp_t& r, const fp_t& a, const fp_t& b) { f_sub(r, a, b); }
KZG_DEV void fp_dbl(fp_t& r, const fp_t& a) { f_add(r, a, a); }

struct alignas(16) g1_affine_t {
    fp_t x, y;
};
struct alignas(16) g1_xyzz_t {
    fp_t x, y, zz, zzz;
};

KZG_DEV bool g1_affine_is_inf(const g1_affine_t& p) { return f_is_zero(p.x) && f_is_zero(p.y); }
KZG_DEV bool g1_is_inf(const g1_xyzz_t& p) { return f_is_zero(p.zz); }
KZG_DEV void g1_set_inf(g1_xyzz_t& p) {
    f_zero(p.x); f_zero(p.y); f_zero(p.zz); f_zero(p.zzz);
}
KZG_DEV void g1_from_affine(g1_xyzz_t& r, const g1_affine_t& p) {
    if (g1_affine_is_inf(p)) { g1_set_inf(r); return; }
    r.x = p.x; r.y = p.y; f_one(r.zz); f_one(r.zzz);
}
KZG_DEV void g1_neg_affine(g1_affine_t& r, const g1_affine_t& p, bool negate) {
    fp_t ny;
    f_neg(ny, p.y);
    r.x = p.x;
    bi_select<12>(r.y.l, p.y.l, ny.l, negate);
}

// 2*(x, y) for an affine non-infinity point (EFD mdbl-2008-s-1, a = 0)
KZG_DEV void g1_dbl_affine(g1_xyzz_t& r, const fp_t& x, const fp_t& y) {
    fp_t U, V, W, S, M, t;
    fp_dbl(U, y);
    fp_sqr(V, U);
    fp_mul(W, U, V);
    fp_mul(S, x, V);
    fp_sqr(M, x);
    fp_dbl(t, M); fp_add(M, t, M);
    fp_sqr(r.x, M); fp_sub(r.x, r.x, S); fp_sub(r.x, r.x, S);
    fp_sub(t, S, r.x); fp_mul(t, M, t);
    fp_mul(U, W, y);
    fp_sub(r.y, t, U);
    r.zz = V; r.zzz = W;
}
// r = 2*p (EFD dbl-2008-s-1, a = 0)
KZG_DEV void g1_dbl(g1_xyzz_t& r, const g1_xyzz_t& p) {
    if (g1_is_inf(p)) { g1_set_inf(r); return; }
    fp_t U, V, W, S, M, t, x3;
    fp_dbl(U, p.y);
    fp_sqr(V, U);
    fp_mul(W, U, V);
    fp_mul(S, p.x, V);
    fp_sqr(M, p.x);
    fp_dbl(t, M); fp_add(M, t, M);
    fp_sqr(x3, M); fp_sub(x3, x3, S); fp_sub(x3, x3, S);
    fp_sub(t, S, x3); fp_mul(t, M, t);
    fp_mul(U, W, p.y);
    fp_sub(r.y, t, U);
    r.x = x3;
    fp_mul(r.zz, V, p.zz);
    fp_mul(r.zzz, W, p.zzz);
}
// acc += (qx, qy), affine non-infinity q (EFD madd-2008-s: 8M + 2S).  Branch-free on the common path; the
// only branches are the rare acc == q doubling and the "acc was empty" select.
KZG_DEV void g1_madd(g1_xyzz_t& acc, const fp_t& qx, const fp_t& qy) {
    const bool acc_inf = g1_is_inf(acc);
    fp_t U2, S2, P, R, PP, PPP, Q, t, x3, y3;
    fp_mul(U2, qx, acc.zz);
    fp_mul(S2, qy, acc.zzz);
    fp_sub(P, U2, acc.x);
    fp_sub(R, S2, acc.y);
    if (!acc_inf && f_is_zero(P) && f_is_zero(R)) {  // same point: the chord formula degenerates
        g1_dbl_affine(acc, qx, qy);
        return;
    }
    fp_sqr(PP, P);
    fp_mul(PPP, P, PP);
    fp_mul(Q, acc.x, PP);
    fp_sqr(x3, R); fp_sub(x3, x3, PPP); fp_sub(x3, x3, Q); fp_sub(x3, x3, Q);
    fp_sub(t, Q, x3); fp_mul(t, R, t);
    fp_mul(y3, acc.y, PPP);
    fp_sub(y3, t, y3);
    fp_mul(t, acc.zz, PP);      // P == 0, R != 0 (q == -acc) gives ZZ3 = 0: infinity, as it must
    fp_mul(U2, acc.zzz, PPP);
    fp_t one;
    f_one(one);
    bi_select<12>(acc.x.l, x3.l, qx.l, acc_inf);
    bi_select<12>(acc.y.l, y3.l, qy.l, acc_inf);
    bi_select<12>(acc.zz.l, t.l, one.l, acc_inf);
    bi_select<12>(acc.zzz.l, U2.l, one.l, acc_inf);
}
// acc += q, affine q that may be infinity
KZG_DEV void g1_madd_checked(g1_xyzz_t& acc, const g1_affine_t& q) {
    if (g1_affine_is_inf(q)) return;
    g1_madd(acc, q.x, q.y);
}
// r = p + q (EFD add-2008-s: 12M + 2S) with the exceptional cases
KZG_DEV void g1_add(g1_xyzz_t& r, const g1_xyzz_t& p, const g1_xyzz_t& q) {
    if (g1_is_inf(p)) { r = q; return; }
    if (g1_is_inf(q)) { r = p; return; }
    fp_t U1, U2, S1, S2, P, R, PP, PPP, Q, t, x3;
    fp_mul(U1, p.x, q.zz);
    fp_mul(U2, q.x, p.zz);
    fp_mul(S1, p.y, q.zzz);
    fp_mul(S2, q.y, p.zzz);
    fp_sub(P, U2, U1);
    fp_sub(R, S2, S1);
    if (f_is_zero(P)) {
        if (f_is_zero(R)) { g1_dbl(r, p); return; }
        g1_set_inf(r);
        return;
    }
    fp_sqr(PP, P);
    fp_mul(PPP, P, PP);
    fp_mul(Q, U1, PP);
    fp_sqr(x3, R); fp_sub(x3, x3, PPP); fp_sub(x3, x3, Q); fp_sub(x3, x3, Q);
    fp_sub(t, Q, x3); fp_mul(t, R, t);
    fp_mul(S1, S1, PPP);
    fp_sub(r.y, t, S1);
    r.x = x3;
    fp_mul(t, p.zz, q.zz); fp_mul(r.zz, t, PP);
    fp_mul(t, p.zzz, q.zzz); fp_mul(r.zzz, t, PPP);
}

// ---- Fp inversion.  One lane's Fermat ladder (381 squarings) costs > 1 ms of pure latency on a GPU, so the
// single inversion that ends every MSM uses the binary extended Euclid instead: only shifts, adds and compares on
// 12 limbs, ~760 halvings + ~380 subtractions.  Invariants x1*A = u*k, x2*A = v*k (mod p) with k = R^2, so for a
// Montgomery-form input A = aR the result x1 = R^2/A = a^-1 R is the Montgomery form of the inverse directly.
template <int N>
KZG_DEV void bi_shr1(uint32_t* a) {
#pragma unroll
    for (int i = 0; i < N - 1; i++) a[i] = __builtin_amdgcn_alignbit(a[i + 1], a[i], 1);
    a[N - 1] >>= 1;
}
KZG_DEV void fp_half(uint32_t* x) {  // x/2 mod p for x in [0, p)
    const uint32_t mask = 0u - (x[0] & 1u);
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) x[i] = __builtin_addc(x[i], FpParams::mod(i) & mask, c, &c);  // < 2^382: no carry out
    bi_shr1<12>(x);
}
KZG_DEV bool bi_is_one12(const uint32_t* a) {
    uint32_t t = a[0] ^ 1u;
#pragma unroll
    for (int i = 1; i < 12; i++) t |= a[i];
    return t == 0;
}
KZG_DEV void fp_inv(fp_t& r, const fp_t& a) {
    if (f_is_zero(a)) { f_zero(r); return; }
    uint32_t u[12], v[12];
    fp_t x1, x2;
#pragma unroll
    for (int i = 0; i < 12; i++) {
        u[i] = a.l[i];
        v[i] = FpParams::mod(i);
        x1.l[i] = FpParams::r2(i);
        x2.l[i] = 0;
    }
    for (int guard = 0; guard < 2000 && !bi_is_one12(u) && !bi_is_one12(v); guard++) {  // bounded: never hangs
        while (!(u[0] & 1u)) { bi_shr1<12>(u); fp_half(x1.l); }
        while (!(v[0] & 1u)) { bi_shr1<12>(v); fp_half(x2.l); }
        if (bi_ge<12>(u, v)) {
            bi_sub<12>(u, u, v);
            f_sub(x1, x1, x2);
            if (bi_is_zero<12>(u)) break;  // u == v == 1 before the subtraction
        } else {
            bi_sub<12>(v, v, u);
            f_sub(x2, x2, x1);
        }
    }
    const bool take_x1 = bi_is_one12(u);
    bi_select<12>(r.l, x2.l, x1.l, take_x1);
}
// Fermat ladder a^(p-2): used where thousands of lanes invert at once (throughput-bound batch normalisation)
struct FpInvExp {
    static constexpr int BITS = 381;
    __device__ static constexpr uint32_t limb(int i) { return i == 0 ? FpParams::mod(0) - 2u : FpParams::mod(i); }
};
KZG_DEV void fp_inv_fermat(fp_t& r, const fp_t& a) {
    fp_t acc;
    f_one(acc);
    for (int i = FpInvExp::BITS - 1; i >= 0; i--) {
        fp_sqr(acc, acc);
        uint32_t w = 0;
#pragma unroll
        for (int k = 0; k < 12; k++) w = (k == (i >> 5)) ? FpInvExp::limb(k) : w;
        if ((w >> (i & 31)) & 1u) fp_mul(acc, acc, a);
    }
    r = acc;
}
// affine (Montgomery) from XYZZ; infinity -> (0,0)
KZG_DEV void g1_to_affine(g1_affine_t& r, const g1_xyzz_t& p) {
    if (g1_is_inf(p)) { f_zero(r.x); f_zero(r.y); return; }
    fp_t i, t;
    fp_mul(t, p.zz, p.zzz);
    fp_inv(i, t);
    fp_mul(t, i, p.zzz);  // 1/ZZ
    fp_mul(r.x, p.x, t);
    fp_mul(t, i, p.zz);   // 1/ZZZ
    fp_mul(r.y, p.y, t);
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
// ZCash 48-byte compressed encoding of an affine Montgomery-form point
KZG_DEV void g1_compress(uint8_t* out48, const g1_affine_t& p) {
    if (g1_affine_is_inf(p)) {
        uint32_t* w = reinterpret_cast<uint32_t*>(out48);
#pragma unroll
        for (int i = 0; i < 12; i++) w[i] = 0;
        out48[0] = 0xC0;
        return;
    }
    fp_t x, y, two_y;
    f_from_mont(x, p.x);
    f_from_mont(y, p.y);
    // y > (p-1)/2  <=>  2y > p - 1  <=>  2y >= p (p odd, 2y != p)
    uint32_t t[13];
    uint32_t c = bi_add<12>(t, y.l, y.l);
    uint32_t pm[12];
#pragma unroll
    for (int i = 0; i < 12; i++) pm[i] = FpParams::mod(i);
    bool larger = c || bi_ge<12>(t, pm);
    (void)two_y;
    limbs_to_be<12>(out48, x.l);
    out48[0] |= larger ? 0xA0 : 0x80;
}
