// msm_dev.hip.h -- device helpers shared by the translation units of the MSM pipeline (msm_sort.hip, msm_accumulate.hip,
// msm_tree.hip, g1_kernels.hip, srs_kernels.hip): the launch-size helper, the issue priority of the latency-bound tail, the
// cooperative (4 waves per addition) point operations, and the LDS tree sum.  Declarations of the launchers: msm.hip.h.
#pragma once
#include "msm.hip.h"
#include "fp_lp.hip.h"

#define NONE_KEY 0xffffffffu
static inline uint32_t nblk(uint64_t n, uint32_t b) { return (uint32_t)((n + b - 1) / b); }

// ---- scalars: 8 x u32 little-endian; bits [lo, lo + c) of one
KZG_DEV uint32_t limb_at(const uint32_t* s, int i) {
    uint32_t r = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) r = (i == k) ? s[k] : r;
    return r;
}
KZG_DEV uint32_t window_bits(const uint32_t* s, int lo, int c) {
    if (lo >= 256) return 0;
    int wi = lo >> 5, b = lo & 31;
    uint64_t v = limb_at(s, wi) | ((uint64_t)limb_at(s, wi + 1) << 32);  // limb_at(.., 8) == 0
    return (uint32_t)(v >> b) & ((1u << c) - 1u);
}
KZG_DEV void load_scalar(uint32_t* s, const uint32_t* scalars, uint64_t j, int mont) {
    const uint4* p = reinterpret_cast<const uint4*>(scalars + 8 * j);
    uint4 a = p[0], b = p[1];
    s[0] = a.x; s[1] = a.y; s[2] = a.z; s[3] = a.w;
    s[4] = b.x; s[5] = b.y; s[6] = b.z; s[7] = b.w;
    if (mont) {  // Montgomery-form row (the coefficients of a commitment): back to the canonical integer
        fr9_t v;
        fr9_from_words(v, s);
        fr9_from_mont(v, v);
        fr9_to_words(s, v);
    }
}

// The kernels after the accumulate (fold, bucket tree, final combination, encoding) are chains of dependent point
// operations with little work.  When they share the GPU with another lane's accumulate kernel (two requests in flight,
// or the two MSMs of a long row's commit+open) they must (a) FIT next to it -- k_msm_accumulate is held to 256
// registers (launch bound 2 waves/SIMD; 280 with AGPR spill space otherwise) so that a 250-register tail wave can be
// resident on the same SIMD -- and (b) win the issue arbitration against a wave that saturates the integer pipe.
KZG_DEV void tail_priority() { __builtin_amdgcn_s_setprio(3); }

// ------------------------------------------------------------------------------------------------ cooperative ops
// The tail of the MSM (fold, bucket tree, final) is a chain of DEPENDENT point additions; one lane needs ~16 us per
// addition (14 Fp products back to back), so these phases are pure latency.  A cooperative addition spreads ONE
// addition over the 4 waves of a 256-thread workgroup: lane l of every wave works on operation l, wave w computes
// the w-th of the independent Fp products of each stage, results cross waves through LDS.  Dependent depth: 4
// products + 4 barriers instead of 14 products.  Exceptional operands (an infinity, or equal x: P == 0) are
// detected identically by every wave (same inputs) and redone by wave 0 with the ordinary g1_add.
// All 256 threads of the workgroup must call these functions (they contain __syncthreads).
struct CoopLds {
    fp_t v[12][64];  // U1 U2 S1 S2 | PP RR ZZ12 ZZZ12 | PPP Q t u
};
KZG_DEV void lds_put(fp_t* dst, const fp_t& v) {
#pragma unroll
    for (int i = 0; i < 14; i++) dst->l[i] = v.l[i];
}
KZG_DEV void lds_get(fp_t& v, const fp_t* src) {
#pragma unroll
    for (int i = 0; i < 14; i++) v.l[i] = src->l[i];
}
// *out = *p + *q for the operation of this lane (lane = threadIdx.x & 63; all four waves pass the same pointers).
// Operands are read field by field from memory (global or LDS) so that no wave holds both points in registers;
// wave 0 writes the complete result.  `active` = this lane has an operation at all (inactive lanes only keep the
// barriers company).  out may alias p or q.
KZG_DEV void load_fp(fp_t& v, const fp_t* src) {
#pragma unroll
    for (int i = 0; i < 14; i++) v.l[i] = src->l[i];
}
KZG_DEV void coop_add(CoopLds& sm, g1_xyzz_t* out, const g1_xyzz_t* p, const g1_xyzz_t* q, bool active) {
    const uint32_t w = threadIdx.x >> 6, l = threadIdx.x & 63;
    fp_t a, b, t;
    bool pinf = true, qinf = true;
    fp_zero(a);
    fp_zero(b);
    if (active) {
        fp_t z1, z2;
        load_fp(z1, &p->zz);
        load_fp(z2, &q->zz);
        pinf = fp_limbs_zero(z1);
        qinf = fp_limbs_zero(z2);
        // stage 1 operands: U1 = X1 ZZ2, U2 = X2 ZZ1, S1 = Y1 ZZZ2, S2 = Y2 ZZZ1
        if (w == 0) { load_fp(a, &p->x); b = z2; }
        else if (w == 1) { load_fp(a, &q->x); b = z1; }
        else if (w == 2) { load_fp(a, &p->y); load_fp(b, &q->zzz); }
        else { load_fp(a, &q->y); load_fp(b, &p->zzz); }
    }
    fp_mul(t, a, b);
    lds_put(&sm.v[w][l], t);
    __syncthreads();
    // stage 2: PP = (U2 - U1)^2 (w0), RR = (S2 - S1)^2 (w1), ZZ12 = ZZ1 ZZ2 (w2), ZZZ12 = ZZZ1 ZZZ2 (w3)
    if (w < 2) {
        fp_t x, y;
        lds_get(x, &sm.v[w == 0 ? 0 : 2][l]);
        lds_get(y, &sm.v[w == 0 ? 1 : 3][l]);
        fp_sub4(a, y, x);
        b = a;
    } else if (active) {
        load_fp(a, w == 2 ? &p->zz : &p->zzz);
        load_fp(b, w == 2 ? &q->zz : &q->zzz);
    }
    fp_mul(t, a, b);
    lds_put(&sm.v[4 + w][l], t);
    __syncthreads();
    fp_t PP;
    lds_get(PP, &sm.v[4][l]);
    const bool special = !active || pinf || qinf || fp_is_zero_n(PP);
    // stage 3: PPP = P PP (w0; a still holds P), Q = U1 PP (w1), zz3 = ZZ12 PP (w2); w3 repeats w2's product, unused
    if (w == 1) lds_get(a, &sm.v[0][l]);
    else if (w >= 2) lds_get(a, &sm.v[6][l]);
    fp_mul(t, a, PP);
    if (w < 2) lds_put(&sm.v[8 + w][l], t);
    if (w == 2) lds_put(&sm.v[10][l], t);  // zz3
    __syncthreads();
    // stage 4: x3 = RR - PPP - 2Q ; t = R (Q - x3) (w0) ; u = S1 PPP (w1) ; zzz3 = ZZZ12 PPP (w2, w3 idles alike)
    fp_t PPP, x3;
    lds_get(PPP, &sm.v[8][l]);
    if (w == 0) {
        fp_t Q, RR, s1, s2;
        lds_get(Q, &sm.v[9][l]);
        lds_get(RR, &sm.v[5][l]);
        fp_sub4(t, RR, PPP); fp_sub4(t, t, Q); fp_sub4(t, t, Q);
        fp_norm(x3, t);
        fp_sub16(b, Q, x3);
        lds_get(s1, &sm.v[2][l]);
        lds_get(s2, &sm.v[3][l]);
        fp_sub4(a, s2, s1);  // R
    } else if (w < 3) {
        lds_get(a, &sm.v[w == 1 ? 2 : 7][l]);
        b = PPP;
    } else {
        a = PPP;  // wave 3 has no product in this stage
        b = PPP;
    }
    fp_mul(t, a, b);
    if (w == 1) lds_put(&sm.v[11][l], t);  // u = S1 PPP
    if (w == 2) lds_put(&sm.v[7][l], t);   // zzz3 (ZZZ12 is dead now)
    __syncthreads();
    if (w == 0 && active) {
        if (special) {  // rare: an infinity or equal x coordinates -> the ordinary addition, one lane
            g1_xyzz_t pa, qa, r;
            load_xyzz(pa, p);
            load_xyzz(qa, q);
            g1_add(r, pa, qa);
            store_xyzz(out, r);
        } else {
            fp_t u, y3;
            lds_get(u, &sm.v[11][l]);
            fp_sub4(t, t, u);
            fp_norm(y3, t);
            g1_xyzz_t r;
            r.x = x3; r.y = y3;
            lds_get(r.zz, &sm.v[10][l]);
            lds_get(r.zzz, &sm.v[7][l]);
            store_xyzz(out, r);
        }
    }
    __syncthreads();  // result visible to the workgroup; sm free for the next cooperative call
}

// *out = 2 * *p, cooperative (EFD dbl-2008-s-1 spread over the waves: dependent depth 3 products instead of 9)
KZG_DEV void coop_dbl(CoopLds& sm, g1_xyzz_t* out, const g1_xyzz_t* p, bool active) {
    const uint32_t w = threadIdx.x >> 6, l = threadIdx.x & 63;
    fp_t a, b, t, X, Y;
    fp_zero(X);
    fp_zero(Y);
    bool inf = true;
    if (active) {
        fp_t z;
        load_fp(z, &p->zz);
        inf = fp_limbs_zero(z);
        load_fp(X, &p->x);
        load_fp(Y, &p->y);
    }
    const bool live = active && !inf;
    fp_t U;
    fp_dbl(U, Y);
    // stage 1: V = U^2 (w0), XX = X^2 (w1)
    a = (w == 0) ? U : X;
    fp_sqr(t, a);
    if (w < 2) lds_put(&sm.v[w][l], t);
    __syncthreads();
    // stage 2: W = U V (w0), S = X V (w1), MM = (3 XX)^2 (w2), zz3 = V ZZ (w3)
    fp_t V, M;
    lds_get(V, &sm.v[0][l]);
    {
        fp_t xx;
        lds_get(xx, &sm.v[1][l]);
        fp_add(M, xx, xx);
        fp_add(M, M, xx);
    }
    if (w == 0) { a = U; b = V; }
    else if (w == 1) { a = X; b = V; }
    else if (w == 2) { a = M; b = M; }
    else { a = V; fp_zero(b); if (active) load_fp(b, &p->zz); }
    fp_mul(t, a, b);
    lds_put(&sm.v[2 + w][l], t);  // W S MM zz3
    __syncthreads();
    // stage 3: x3 = MM - 2S ; t = M (S - x3) (w0) ; u = W Y (w1) ; zzz3 = W ZZZ (w2)
    fp_t x3;
    if (w == 0) {
        fp_t S, MM;
        lds_get(S, &sm.v[3][l]);
        lds_get(MM, &sm.v[4][l]);
        fp_sub4(t, MM, S); fp_sub4(t, t, S);
        fp_norm(x3, t);
        a = M;
        fp_sub16(b, S, x3);
    } else {
        lds_get(a, &sm.v[2][l]);  // W
        if (w == 1) b = Y;
        else { fp_zero(b); if (active) load_fp(b, &p->zzz); }
    }
    fp_mul(t, a, b);
    if (w == 1) lds_put(&sm.v[6][l], t);
    if (w == 2) lds_put(&sm.v[7][l], t);
    __syncthreads();
    if (w == 0 && live) {
        fp_t u, y3;
        lds_get(u, &sm.v[6][l]);
        fp_sub4(t, t, u);
        fp_norm(y3, t);
        g1_xyzz_t r;
        r.x = x3; r.y = y3;
        lds_get(r.zz, &sm.v[5][l]);
        lds_get(r.zzz, &sm.v[7][l]);
        store_xyzz(out, r);
    } else if (w == 0 && active && out != p) {
        g1_xyzz_t r;
        g1_set_inf(r);
        store_xyzz(out, r);
    }
    __syncthreads();
}


// sum of the values held by lanes [0, nthreads) (power of two <= blockDim), result in every lane's `mine`
KZG_DEV void lds_tree_sum(g1_xyzz_t* sm, g1_xyzz_t& mine, uint32_t tid, uint32_t nthreads) {
    store_xyzz(&sm[tid], mine);
    __syncthreads();
    for (uint32_t d = nthreads >> 1; d >= 1; d >>= 1) {
        if (tid < d) {
            g1_xyzz_t a, b, r;
            load_xyzz(a, &sm[tid]);
            load_xyzz(b, &sm[tid + d]);
            g1_add(r, a, b);
            store_xyzz(&sm[tid], r);
        }
        __syncthreads();
    }
    load_xyzz(mine, &sm[0]);
}

