// Lane-parallel Fp and G1 for the latency-bound tail of the MSM (narrow bucket-tree levels, final combination).
//
// One lane needs ~2500 cycles for an Fp product (505 instructions at ~5 cycles each) and a point addition is a chain
// of four such products deep, so a phase with only a handful of point operations is pure latency.  Here ONE field
// element lives in a DPP row: limb j (28 bits, fp28.hip.h) in lane j of the row's 16 lanes, lanes 14 / 15 zero.  The
// Montgomery product runs operand-scanning across the row -- per step: broadcast b_i (row_newbcast), one mad, q from
// lane 0 (row_newbcast:0), one mad, then every lane keeps its high part and takes its upper neighbour's low 28 bits
// (row_shl:1), which both shifts the window one limb down and keeps the accumulators below 2^37 -- 14 steps of 11
// instructions instead of 505 dependent-issue slots: ~900 cycles per product, and the four rows of a wave run four
// INDEPENDENT products at once, which is exactly the width of a point addition's stages (4 + 4 + 3 + 3 products).
// One wave = one point operation; values cross rows through a 1-KB slice of LDS private to the wave (no barriers).
// Modelled limb-exactly in scripts/models/lp_mul_model.py before transcription.  Used only where operations are
// few (<= LP_MAX_OPS): per operation it costs ~6x the wave-instructions of the one-lane-per-operation form.
#pragma once
#include "g1.hip.h"

#ifndef LP_MAX_OPS
#define LP_MAX_OPS 3072
#endif

template <int CTRL>
KZG_DEV uint32_t lp_dpp(uint32_t v) {  // out-of-row sources read as zero
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, true);
}
#define LP_BCAST(i) (0x150 + (i))  // row_newbcast:i  (gfx90a+)
#define LP_FROM_UPPER 0x101        // row_shl:1: lane j reads lane j + 1
#define LP_FROM_LOWER 0x111        // row_shr:1: lane j reads lane j - 1

struct LpLane {
    uint32_t j, row;            // limb index inside the row (0..15), row inside the wave (0..3)
    uint32_t p, m4, m8, m16;    // limb j of p and of the subtraction multiples 4p / 8p / 16p (fp28.hip.h); 0 for j >= 14
};
#define LP_PICK(dst, fn)                                                  \
    {                                                                     \
        uint32_t r_ = 0;                                                  \
        _Pragma("unroll") for (int i_ = 0; i_ < 14; i_++) r_ = (k.j == (uint32_t)i_) ? fn(i_) : r_; \
        dst = r_;                                                         \
    }
KZG_DEV LpLane lp_lane() {
    LpLane k;
    const uint32_t lane = threadIdx.x & 63u;
    k.j = lane & 15u;
    k.row = lane >> 4;
    LP_PICK(k.p, fp28_p)
    LP_PICK(k.m4, fp28_m4)
    LP_PICK(k.m8, fp28_m8)
    LP_PICK(k.m16, fp28_m16)
    return k;
}
// carry propagation inside every row: same value, limbs 0..12 < 2^28 (lane 13 keeps the excess).  Wave-uniform loop
// (all 64 lanes must call): one round almost always, a carry ripples further only through limbs equal to 2^28 - 1.
KZG_DEV uint32_t lp_norm(uint64_t t, const LpLane& k) {
    const bool low = k.j < 13;
    for (;;) {
        if (!__ballot(low && t > FP28_MASK)) break;
        const uint32_t lo = low ? ((uint32_t)t & FP28_MASK) : (uint32_t)t;
        const uint32_t hi = low ? (uint32_t)(t >> 28) : 0u;      // < 2^36 / 2^28: fits
        t = (uint64_t)lo + lp_dpp<LP_FROM_LOWER>(hi);
    }
    return (uint32_t)t;
}
// r = a * b / 2^392 (mod p) per row; a, b loose (limbs < 2^30 + 2^28, value < 32p); r is N class (normalised, < 2p)
#define LP_MUL_STEP(i)                                                  \
    {                                                                   \
        t += (uint64_t)a * lp_dpp<LP_BCAST(i)>(b);                      \
        const uint32_t q = ((uint32_t)t * FP28_PINV) & FP28_MASK;       \
        t += (uint64_t)lp_dpp<LP_BCAST(0)>(q) * k.p;                    \
        const uint32_t lo = (uint32_t)t & FP28_MASK;                    \
        t = (t >> 28) + lp_dpp<LP_FROM_UPPER>(lo);                      \
    }
KZG_DEV uint32_t lp_mul(uint32_t a, uint32_t b, const LpLane& k) {
    uint64_t t = 0;
    LP_MUL_STEP(0) LP_MUL_STEP(1) LP_MUL_STEP(2) LP_MUL_STEP(3) LP_MUL_STEP(4) LP_MUL_STEP(5) LP_MUL_STEP(6)
    LP_MUL_STEP(7) LP_MUL_STEP(8) LP_MUL_STEP(9) LP_MUL_STEP(10) LP_MUL_STEP(11) LP_MUL_STEP(12) LP_MUL_STEP(13)
    return lp_norm(t, k);
}
KZG_DEV uint32_t lp_sub4(uint32_t a, uint32_t b, const LpLane& k) { return a + (k.m4 - b); }     // b normalised, < 2p
KZG_DEV uint32_t lp_sub16(uint32_t a, uint32_t b, const LpLane& k) { return a + (k.m16 - b); }   // b normalised, < 14p

// limb j of a field element in memory (global or LDS)
KZG_DEV uint32_t lp_load(const fp_t* f, const LpLane& k) { return k.j < 14 ? f->l[k.j] : 0u; }
KZG_DEV void lp_store(fp_t* f, uint32_t v, const LpLane& k) {
    if (k.j < 14) f->l[k.j] = v;
}
// wave-private LDS exchange: slot s holds one field element (16 dwords).  All accesses of a slot come from the same
// wave, whose LDS operations execute in order; the fences keep the compiler from moving them across each other.
#define LP_SLOTS 12
struct LpScratch {
    uint32_t v[LP_SLOTS][16];
};
KZG_DEV void lp_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
KZG_DEV void lp_put(LpScratch& sm, int slot, uint32_t v, bool mine, const LpLane& k) {
    if (mine) sm.v[slot][k.j] = v;
}
KZG_DEV uint32_t lp_get(const LpScratch& sm, int slot, const LpLane& k) { return sm.v[slot][k.j]; }
// all limbs of the row-0 copy zero / all equal to p's (wave-uniform answers)
KZG_DEV bool lp_row0_all(bool pred_per_lane, const LpLane& k) {
    const uint64_t m = __ballot(pred_per_lane || k.j >= 14);
    return (m & 0xffffull) == 0xffffull;
}

// the rare exceptional addition (an infinity, equal x coordinates), one lane, out of line: its ~250 registers must not
// set the budget of the lane-parallel kernels
static __device__ __noinline__ void lp_add_fallback(g1_xyzz_t* out, const g1_xyzz_t* p, const g1_xyzz_t* q) {
    g1_xyzz_t pa, qa, o;
    load_xyzz(pa, p);
    load_xyzz(qa, q);
    g1_add(o, pa, qa);
    store_xyzz(out, o);
}
// *out = *p + *q (XYZZ, add-2008-s) by ONE wave; every lane of the wave must call with the same arguments.
// Exceptional operands (an infinity, equal x coordinates) fall back to the ordinary one-lane addition.
// out may alias p or q.
KZG_DEV void lp_add(LpScratch& sm, g1_xyzz_t* out, const g1_xyzz_t* p, const g1_xyzz_t* q, const LpLane& k) {
    const uint32_t r = k.row;
    // stage 1: U1 = X1 ZZ2 | U2 = X2 ZZ1 | S1 = Y1 ZZZ2 | S2 = Y2 ZZZ1
    const fp_t* fa = r == 0 ? &p->x : r == 1 ? &q->x : r == 2 ? &p->y : &q->y;
    const fp_t* fb = r == 0 ? &q->zz : r == 1 ? &p->zz : r == 2 ? &q->zzz : &p->zzz;
    uint32_t a = lp_load(fa, k), b = lp_load(fb, k);
    const uint64_t nz = __ballot(b != 0);
    const bool qinf = (nz & 0xffffull) == 0, pinf = ((nz >> 16) & 0xffffull) == 0;   // row 0 holds ZZ2, row 1 ZZ1
    // operands of stage 2's rows 2 / 3 are fetched before anything is written (out may alias p / q)
    const uint32_t z1 = lp_load(r == 3 ? &p->zzz : &p->zz, k), z2 = lp_load(r == 3 ? &q->zzz : &q->zz, k);
    if (pinf || qinf) {
        if ((threadIdx.x & 63u) == 0) lp_add_fallback(out, p, q);
        return;
    }
    uint32_t t = lp_mul(a, b, k);
    lp_put(sm, r, t, true, k);                    // slots 0..3 = U1 U2 S1 S2
    lp_sync();
    // stage 2: PP = (U2 - U1)^2 | RR = (S2 - S1)^2 | ZZ12 = ZZ1 ZZ2 | ZZZ12 = ZZZ1 ZZZ2
    uint32_t d = 0;
    if (r < 2) {
        d = lp_sub4(lp_get(sm, 2 * r + 1, k), lp_get(sm, 2 * r, k), k);     // P (row 0), R (row 1)
        a = b = d;
    } else {
        a = z1;
        b = z2;
    }
    t = lp_mul(a, b, k);
    lp_put(sm, 4 + r, t, true, k);                // slots 4..7 = PP RR ZZ12 ZZZ12
    lp_put(sm, 8, d, r == 1, k);                  // slot 8 = R
    lp_sync();
    const uint32_t PP = lp_get(sm, 4, k);
    const bool pp_zero = lp_row0_all(PP == 0, k) || lp_row0_all(PP == k.p, k);
    if (pp_zero) {                                // equal x: doubling or cancellation, one lane
        if ((threadIdx.x & 63u) == 0) lp_add_fallback(out, p, q);
        return;
    }
    // stage 3: PPP = P PP | Q = U1 PP | ZZ3 = ZZ12 PP | (row 3 repeats row 2's product; unused)
    a = r == 0 ? d : r == 1 ? lp_get(sm, 0, k) : lp_get(sm, 6, k);
    t = lp_mul(a, PP, k);
    lp_put(sm, 9, t, r == 0, k);                  // PPP
    lp_put(sm, 10, t, r == 1, k);                 // Q
    const uint32_t zz3 = t;                       // row 2
    lp_sync();
    // stage 4: x3 = RR - PPP - 2Q | T = R (Q - x3) | U = S1 PPP | ZZZ3 = ZZZ12 PPP
    const uint32_t PPP = lp_get(sm, 9, k), Q = lp_get(sm, 10, k), RR = lp_get(sm, 5, k);
    const uint32_t x3 = lp_norm((uint64_t)RR + (k.m4 - PPP) + (k.m4 - Q) + (k.m4 - Q), k);   // < 14p
    if (r == 0) {
        a = lp_get(sm, 8, k);
        b = lp_sub16(Q, x3, k);
    } else {
        a = lp_get(sm, r == 1 ? 2 : 7, k);
        b = PPP;
    }
    t = lp_mul(a, b, k);
    lp_put(sm, 11, t, r == 1, k);                 // U = S1 PPP
    lp_sync();
    const uint32_t y3 = lp_norm(r == 0 ? (uint64_t)t + (k.m4 - lp_get(sm, 11, k)) : 0ull, k);   // row 0: T - U, < 6p
    if (r == 0) {
        lp_store(&out->x, x3, k);
        lp_store(&out->y, y3, k);
    } else if (r == 2) {
        lp_store(&out->zz, zz3, k);
        lp_store(&out->zzz, t, k);
    }
}

// *out = 2 * *p (dbl-2008-s-1, a = 0) by one wave; out may alias p
KZG_DEV void lp_dbl(LpScratch& sm, g1_xyzz_t* out, const g1_xyzz_t* p, const LpLane& k) {
    const uint32_t r = k.row;
    const uint32_t X = lp_load(&p->x, k), Y = lp_load(&p->y, k);
    const uint32_t Z = lp_load(r == 2 ? &p->zzz : &p->zz, k);    // row 3: ZZ (stage 2), row 2: ZZZ (stage 3)
    const bool inf = (__ballot(lp_load(&p->zz, k) != 0) & 0xffffull) == 0;
    if (inf) {
        if (out != p && (threadIdx.x & 63u) < 56) reinterpret_cast<uint32_t*>(out)[threadIdx.x & 63u] = 0;
        return;
    }
    const uint32_t U = Y << 1;                                    // < 12p, limbs < 2^29
    // stage 1: V = U^2 | XX = X^2
    uint32_t a = r == 0 ? U : X;
    uint32_t t = lp_mul(a, a, k);
    lp_put(sm, r, t, r < 2, k);                                   // slots 0, 1 = V, XX
    lp_sync();
    const uint32_t V = lp_get(sm, 0, k), XX = lp_get(sm, 1, k);
    const uint32_t M = XX + XX + XX;                              // 3 X^2: limbs < 3 * 2^28, value < 6p
    // stage 2: W = U V | S = X V | MM = M^2 | ZZ3 = V ZZ
    uint32_t b;
    if (r == 0) { a = U; b = V; }
    else if (r == 1) { a = X; b = V; }
    else if (r == 2) { a = M; b = M; }
    else { a = V; b = Z; }
    t = lp_mul(a, b, k);
    lp_put(sm, 2 + r, t, true, k);                                // slots 2..5 = W S MM ZZ3
    lp_sync();
    // stage 3: x3 = MM - 2S | T = M (S - x3) | Uy = W Y | ZZZ3 = W ZZZ
    const uint32_t W = lp_get(sm, 2, k), S = lp_get(sm, 3, k), MM = lp_get(sm, 4, k);
    const uint32_t x3 = lp_norm((uint64_t)MM + (k.m4 - S) + (k.m4 - S), k);     // < 2p + 8p
    if (r == 0) { a = M; b = lp_sub16(S, x3, k); }
    else if (r == 1) { a = W; b = Y; }
    else { a = W; b = Z; }                                        // row 2: ZZZ (row 3 repeats with ZZ; unused)
    t = lp_mul(a, b, k);
    lp_put(sm, 6, t, r == 1, k);
    lp_sync();
    const uint32_t y3 = lp_norm(r == 0 ? (uint64_t)t + (k.m4 - lp_get(sm, 6, k)) : 0ull, k);    // row 0: < 6p
    const uint32_t zz3 = lp_get(sm, 5, k);
    if (r == 0) {
        lp_store(&out->x, x3, k);
        lp_store(&out->y, y3, k);
    } else if (r == 2) {
        lp_store(&out->zz, zz3, k);
        lp_store(&out->zzz, t, k);
    }
}
