/* TEST INFRASTRUCTURE ONLY -- plain-C CPU restatement of the KZG segment-prover hot path.
 *
 * Not product code: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker / reported CPU baseline ("kind": "port").
 *
 * PARITY UNPINNED against the real reference prover: the reference delegates every field /
 * curve operation to the external Rust binary apollozkp/fourier (reference requirements.txt:3,
 * un-pinned, not under /root/reference; call sites neurons/miner.py:39,48 worker_commit /
 * worker_open, neurons/validator.py:59-104 fft / eval / worker_verify).  This file restates the
 * published algorithms (BLS12-381, KZG10, Pianist worker basis) and is pinned
 * by (a) the reference's Fr known-answer vector tests/test_miner.py:33-55 and (b) bit-for-bit
 * agreement with the independent pure-Python oracle (oracle/bls12_381.py) on tests/golden/.
 *
 * Deliberately different from the GPU design: 64-bit limbs via unsigned __int128 (6 for Fp, 4 for
 * Fr), Booth-recoded signed windows chosen per call from a cost model, AFFINE buckets filled by
 * batched affine additions (one shared inversion per up to 4096 additions), one bucket set per
 * (window, point-chunk) task reduced by the classic running sum in XYZZ, windows combined by
 * doublings; no sort, no precomputed tables.
 * It is also the reported CPU baseline, so it is written the way a careful CPU implementation
 * is (tasks over windows x chunks spread on a thread pool, threaded NTT), not as a toy.
 *
 * Byte conventions: Fr = 32 B big-endian canonical; G1 affine = x||y 2x48 B big-endian
 * (96 zero bytes = infinity); G1 compressed = 48 B ZCash encoding.
 */
#define _GNU_SOURCE
#include <pthread.h>
#include <sched.h>
#include <stdio.h>
#include <stdatomic.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#if defined(__x86_64__)
#include <x86intrin.h>
#endif

typedef unsigned __int128 u128;
typedef uint64_t u64;

#define NP 6 /* Fp limbs */
#define NR 4 /* Fr limbs */

typedef struct { u64 l[NP]; } fp;
typedef struct { u64 l[NR]; } fr;
typedef struct { fp x, y; int inf; } g1a;   /* affine, Montgomery form */
typedef struct { fp x, y, z; } g1j;         /* Jacobian, z == 0 -> infinity */

static const u64 P_MOD[NP] = {0xb9feffffffffaaabULL, 0x1eabfffeb153ffffULL, 0x6730d2a0f6b0f624ULL,
                              0x64774b84f38512bfULL, 0x4b1ba7b6434bacd7ULL, 0x1a0111ea397fe69aULL};
static const u64 R_MOD[NR] = {0xffffffff00000001ULL, 0x53bda402fffe5bfeULL, 0x3339d80809a1d805ULL,
                              0x73eda753299d7d48ULL};

static u64 P_INV, R_INV;          /* -m^-1 mod 2^64 */
static fp FP_ONE, FP_R2;          /* R mod p, R^2 mod p */
static fr FR_ONE, FR_R2;
static g1a G1_GEN;
static int g_init = 0;

/* ------------------------------------------------------------------ generic limb helpers */
static inline int limbs_ge(const u64 *a, const u64 *b, int n) {
    for (int i = n - 1; i >= 0; i--) {
        if (a[i] > b[i]) return 1;
        if (a[i] < b[i]) return 0;
    }
    return 1;
}
static inline u64 limbs_add(u64 *r, const u64 *a, const u64 *b, int n) {
    u128 c = 0;
    for (int i = 0; i < n; i++) { c += (u128)a[i] + b[i]; r[i] = (u64)c; c >>= 64; }
    return (u64)c;
}
static inline u64 limbs_sub(u64 *r, const u64 *a, const u64 *b, int n) {
    u64 br = 0;
    for (int i = 0; i < n; i++) {
        u128 t = (u128)a[i] - b[i] - br;
        r[i] = (u64)t; br = (u64)(t >> 64) & 1;
    }
    return br;
}
static inline int limbs_is_zero(const u64 *a, int n) {
    u64 t = 0; for (int i = 0; i < n; i++) t |= a[i]; return t == 0;
}
/* modular add / sub (inputs and outputs fully reduced).  The carry chains go through the add-with-carry intrinsics where
 * the compiler has them: the portable 128-bit form below costs ~8x as much with gcc and dominated a point addition. */
#if defined(__x86_64__)
#define ADC(c, a, b, out) _addcarry_u64((c), (a), (b), (unsigned long long *)(out))
#define SBB(c, a, b, out) _subborrow_u64((c), (a), (b), (unsigned long long *)(out))
static inline void mod_add(u64 *r, const u64 *a, const u64 *b, const u64 *m, int n) {
    u64 s[NP], d[NP];
    unsigned char c = 0, br = 0;
    for (int i = 0; i < n; i++) c = ADC(c, a[i], b[i], &s[i]);
    for (int i = 0; i < n; i++) br = SBB(br, s[i], m[i], &d[i]);
    const int use_d = c | (br ^ 1);           /* carry out, or no borrow -> take s - m */
    for (int i = 0; i < n; i++) r[i] = use_d ? d[i] : s[i];
}
static inline void mod_sub(u64 *r, const u64 *a, const u64 *b, const u64 *m, int n) {
    u64 d[NP];
    unsigned char br = 0, c = 0;
    for (int i = 0; i < n; i++) br = SBB(br, a[i], b[i], &d[i]);
    const u64 mask = (u64)0 - (u64)br;
    for (int i = 0; i < n; i++) c = ADC(c, d[i], m[i] & mask, &r[i]);
}
#else
static inline void mod_add(u64 *r, const u64 *a, const u64 *b, const u64 *m, int n) {
    u64 s[NP], d[NP];
    u64 c = limbs_add(s, a, b, n);
    u64 br = limbs_sub(d, s, m, n);
    u64 use_d = (u64)0 - (u64)(c | (br ^ 1)); /* carry out, or no borrow -> take s - m */
    for (int i = 0; i < n; i++) r[i] = (d[i] & use_d) | (s[i] & ~use_d);
}
static inline void mod_sub(u64 *r, const u64 *a, const u64 *b, const u64 *m, int n) {
    u64 d[NP];
    u64 mask = (u64)0 - limbs_sub(d, a, b, n);
    u128 c = 0;
    for (int i = 0; i < n; i++) { c += (u128)d[i] + (m[i] & mask); r[i] = (u64)c; c >>= 64; }
}
#endif
/* CIOS Montgomery product; instantiated with a literal limb count so the loops fully unroll */
#define DEFINE_MONT_MUL(NAME, N)                                                                   \
    static inline void NAME(u64 *r, const u64 *a, const u64 *b, const u64 *m, u64 inv) {          \
        u64 t[N + 2];                                                                              \
        _Pragma("GCC unroll 8") for (int i = 0; i < N + 2; i++) t[i] = 0;                         \
        _Pragma("GCC unroll 8") for (int i = 0; i < N; i++) {                                     \
            u128 c = 0;                                                                            \
            _Pragma("GCC unroll 8") for (int j = 0; j < N; j++) {                                 \
                c += (u128)a[j] * b[i] + t[j]; t[j] = (u64)c; c >>= 64;                            \
            }                                                                                      \
            c += t[N]; t[N] = (u64)c; t[N + 1] = (u64)(c >> 64);                                   \
            u64 q = t[0] * inv;                                                                    \
            c = (u128)q * m[0] + t[0]; c >>= 64;                                                   \
            _Pragma("GCC unroll 8") for (int j = 1; j < N; j++) {                                 \
                c += (u128)q * m[j] + t[j]; t[j - 1] = (u64)c; c >>= 64;                           \
            }                                                                                      \
            c += t[N]; t[N - 1] = (u64)c; t[N] = t[N + 1] + (u64)(c >> 64);                        \
        }                                                                                          \
        if (t[N] || limbs_ge(t, m, N)) limbs_sub(r, t, m, N);                                      \
        else for (int i = 0; i < N; i++) r[i] = t[i];                                              \
    }
DEFINE_MONT_MUL(mont_mul6, 6)
DEFINE_MONT_MUL(mont_mul4, 4)
static u64 neg_inv64(u64 m0) {
    u64 x = 1;
    for (int i = 0; i < 6; i++) x *= 2 - m0 * x; /* Newton: x = m0^-1 mod 2^64 */
    return (u64)0 - x;
}
static void pow2_mod(u64 *r, int bits, const u64 *m, int n) { /* 2^bits mod m */
    for (int i = 0; i < n; i++) r[i] = 0;
    r[0] = 1;
    for (int i = 0; i < bits; i++) mod_add(r, r, r, m, n);
}
static void be_to_limbs(u64 *r, const uint8_t *b, int n) {
    for (int i = 0; i < n; i++) {
        u64 v = 0;
        for (int k = 0; k < 8; k++) v = (v << 8) | b[(n - 1 - i) * 8 + k];
        r[i] = v;
    }
}
static void limbs_to_be(uint8_t *b, const u64 *a, int n) {
    for (int i = 0; i < n; i++)
        for (int k = 0; k < 8; k++) b[(n - 1 - i) * 8 + k] = (uint8_t)(a[i] >> (56 - 8 * k));
}

/* ------------------------------------------------------------------ Fp */
static inline void fp_mul(fp *r, const fp *a, const fp *b) { mont_mul6(r->l, a->l, b->l, P_MOD, P_INV); }
static inline void fp_sqr(fp *r, const fp *a) { fp_mul(r, a, a); }
static inline void fp_add(fp *r, const fp *a, const fp *b) { mod_add(r->l, a->l, b->l, P_MOD, NP); }
static inline void fp_sub(fp *r, const fp *a, const fp *b) { mod_sub(r->l, a->l, b->l, P_MOD, NP); }
static inline void fp_dbl(fp *r, const fp *a) { fp_add(r, a, a); }
static inline int fp_is_zero(const fp *a) { return limbs_is_zero(a->l, NP); }
static inline int fp_eq(const fp *a, const fp *b) { return memcmp(a, b, sizeof(fp)) == 0; }
static void fp_neg(fp *r, const fp *a) {
    if (fp_is_zero(a)) *r = *a; else limbs_sub(r->l, P_MOD, a->l, NP);
}
static void fp_from_be(fp *r, const uint8_t *b) { fp t; be_to_limbs(t.l, b, NP); fp_mul(r, &t, &FP_R2); }
static void fp_to_limbs(u64 *out, const fp *a) { /* out of Montgomery form */
    fp one = {{1, 0, 0, 0, 0, 0}}, t; fp_mul(&t, a, &one); memcpy(out, t.l, sizeof(t.l));
}
static void fp_to_be(uint8_t *b, const fp *a) { u64 t[NP]; fp_to_limbs(t, a); limbs_to_be(b, t, NP); }
static void fp_pow(fp *r, const fp *a, const u64 *e, int n) {
    fp acc = FP_ONE;
    for (int i = n * 64 - 1; i >= 0; i--) {
        fp_sqr(&acc, &acc);
        if ((e[i / 64] >> (i % 64)) & 1) fp_mul(&acc, &acc, a);
    }
    *r = acc;
}
static void fp_inv(fp *r, const fp *a) {
    u64 e[NP]; u64 two[NP] = {2, 0, 0, 0, 0, 0};
    limbs_sub(e, P_MOD, two, NP);
    fp_pow(r, a, e, NP);
}

/* ------------------------------------------------------------------ Fr */
static inline void fr_mul(fr *r, const fr *a, const fr *b) { mont_mul4(r->l, a->l, b->l, R_MOD, R_INV); }
static inline void fr_add(fr *r, const fr *a, const fr *b) { mod_add(r->l, a->l, b->l, R_MOD, NR); }
static inline void fr_sub(fr *r, const fr *a, const fr *b) { mod_sub(r->l, a->l, b->l, R_MOD, NR); }
static int fr_from_be(fr *r, const uint8_t *b) { /* returns -1 when non-canonical */
    fr t; be_to_limbs(t.l, b, NR);
    if (limbs_ge(t.l, R_MOD, NR)) return -1;
    fr_mul(r, &t, &FR_R2); return 0;
}
static void fr_to_limbs(u64 *out, const fr *a) {
    fr one = {{1, 0, 0, 0}}, t; fr_mul(&t, a, &one); memcpy(out, t.l, sizeof(t.l));
}
static void fr_to_be(uint8_t *b, const fr *a) { u64 t[NR]; fr_to_limbs(t, a); limbs_to_be(b, t, NR); }
static void fr_pow(fr *r, const fr *a, const u64 *e, int n) {
    fr acc = FR_ONE;
    for (int i = n * 64 - 1; i >= 0; i--) {
        fr_mul(&acc, &acc, &acc);
        if ((e[i / 64] >> (i % 64)) & 1) fr_mul(&acc, &acc, a);
    }
    *r = acc;
}
static void fr_inv(fr *r, const fr *a) {
    u64 e[NR]; u64 two[NR] = {2, 0, 0, 0};
    limbs_sub(e, R_MOD, two, NR);
    fr_pow(r, a, e, NR);
}
static void fr_from_u64(fr *r, u64 v) { fr t = {{v, 0, 0, 0}}; fr_mul(r, &t, &FR_R2); }
/* primitive n-th root of unity 7^((r-1)/n), n = 2^logn */
static void fr_root_of_unity(fr *w, int logn) {
    u64 e[NR], one[NR] = {1, 0, 0, 0};
    limbs_sub(e, R_MOD, one, NR);
    for (int s = 0; s < logn; s++) { /* e >>= 1 */
        for (int i = 0; i < NR; i++) e[i] = (e[i] >> 1) | (i + 1 < NR ? e[i + 1] << 63 : 0);
    }
    fr g; fr_from_u64(&g, 7);
    fr_pow(w, &g, e, NR);
}

/* ------------------------------------------------------------------ G1 (Jacobian) */
static void g1j_set_inf(g1j *p) { p->x = FP_ONE; p->y = FP_ONE; memset(&p->z, 0, sizeof(fp)); }
static inline int g1j_is_inf(const g1j *p) { return fp_is_zero(&p->z); }
static void g1j_from_affine(g1j *r, const g1a *a) {
    if (a->inf) { g1j_set_inf(r); return; }
    r->x = a->x; r->y = a->y; r->z = FP_ONE;
}
static void g1j_double(g1j *r, const g1j *p) { /* dbl-2009-l, a = 0 */
    if (g1j_is_inf(p) || fp_is_zero(&p->y)) { g1j_set_inf(r); return; }
    fp A, B, C, D, E, F, t, z3;
    fp_sqr(&A, &p->x); fp_sqr(&B, &p->y); fp_sqr(&C, &B);
    fp_add(&t, &p->x, &B); fp_sqr(&t, &t); fp_sub(&t, &t, &A); fp_sub(&t, &t, &C); fp_dbl(&D, &t);
    fp_dbl(&E, &A); fp_add(&E, &E, &A);
    fp_sqr(&F, &E);
    fp_mul(&z3, &p->y, &p->z); fp_dbl(&z3, &z3);
    fp_sub(&t, &F, &D); fp_sub(&r->x, &t, &D);
    fp_sub(&t, &D, &r->x); fp_mul(&t, &E, &t);
    fp_dbl(&C, &C); fp_dbl(&C, &C); fp_dbl(&C, &C);
    fp_sub(&r->y, &t, &C);
    r->z = z3;
}
static void g1j_add_affine(g1j *r, const g1j *p, const g1a *q) { /* madd-2007-bl + special cases */
    if (q->inf) { if (r != p) *r = *p; return; }
    if (g1j_is_inf(p)) { g1j_from_affine(r, q); return; }
    fp Z1Z1, U2, S2, H, HH, I, J, rr, V, t, t2;
    fp_sqr(&Z1Z1, &p->z);
    fp_mul(&U2, &q->x, &Z1Z1);
    fp_mul(&S2, &q->y, &p->z); fp_mul(&S2, &S2, &Z1Z1);
    if (fp_eq(&U2, &p->x)) {
        if (fp_eq(&S2, &p->y)) { g1j_double(r, p); return; }
        g1j_set_inf(r); return;
    }
    fp_sub(&H, &U2, &p->x); fp_sqr(&HH, &H);
    fp_dbl(&I, &HH); fp_dbl(&I, &I);
    fp_mul(&J, &H, &I);
    fp_sub(&rr, &S2, &p->y); fp_dbl(&rr, &rr);
    fp_mul(&V, &p->x, &I);
    fp_sqr(&t, &rr); fp_sub(&t, &t, &J); fp_sub(&t, &t, &V); fp_sub(&t, &t, &V); /* X3 */
    fp_sub(&t2, &V, &t); fp_mul(&t2, &rr, &t2);
    fp_mul(&J, &p->y, &J); fp_dbl(&J, &J);
    fp z3; fp_add(&z3, &p->z, &H); fp_sqr(&z3, &z3); fp_sub(&z3, &z3, &Z1Z1); fp_sub(&z3, &z3, &HH);
    r->x = t; fp_sub(&r->y, &t2, &J); r->z = z3;
}
static void g1j_add(g1j *r, const g1j *p, const g1j *q) { /* add-2007-bl + special cases */
    if (g1j_is_inf(p)) { if (r != q) *r = *q; return; }
    if (g1j_is_inf(q)) { if (r != p) *r = *p; return; }
    fp Z1Z1, Z2Z2, U1, U2, S1, S2, H, I, J, rr, V, t, t2, z3;
    fp_sqr(&Z1Z1, &p->z); fp_sqr(&Z2Z2, &q->z);
    fp_mul(&U1, &p->x, &Z2Z2); fp_mul(&U2, &q->x, &Z1Z1);
    fp_mul(&S1, &p->y, &q->z); fp_mul(&S1, &S1, &Z2Z2);
    fp_mul(&S2, &q->y, &p->z); fp_mul(&S2, &S2, &Z1Z1);
    if (fp_eq(&U1, &U2)) {
        if (fp_eq(&S1, &S2)) { g1j_double(r, p); return; }
        g1j_set_inf(r); return;
    }
    fp_sub(&H, &U2, &U1);
    fp_dbl(&I, &H); fp_sqr(&I, &I);
    fp_mul(&J, &H, &I);
    fp_sub(&rr, &S2, &S1); fp_dbl(&rr, &rr);
    fp_mul(&V, &U1, &I);
    fp_sqr(&t, &rr); fp_sub(&t, &t, &J); fp_sub(&t, &t, &V); fp_sub(&t, &t, &V);
    fp_sub(&t2, &V, &t); fp_mul(&t2, &rr, &t2);
    fp_mul(&S1, &S1, &J); fp_dbl(&S1, &S1);
    fp_add(&z3, &p->z, &q->z); fp_sqr(&z3, &z3); fp_sub(&z3, &z3, &Z1Z1); fp_sub(&z3, &z3, &Z2Z2);
    fp_mul(&z3, &z3, &H);
    r->x = t; fp_sub(&r->y, &t2, &S1); r->z = z3;
}
/* ------------------------------------------------------------------ G1 buckets in XYZZ (x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2) */
typedef struct { fp x, y, zz, zzz; } g1x;   /* zz == 0 -> infinity */
static void g1x_set_inf(g1x *p) { memset(p, 0, sizeof(*p)); }
static inline int g1x_is_inf(const g1x *p) { return fp_is_zero(&p->zz); }
static void g1x_double_affine(g1x *r, const fp *x, const fp *y) { /* mdbl-2008-s-1, a = 0 */
    fp U, V, W, S, M, t;
    fp_dbl(&U, y); fp_sqr(&V, &U); fp_mul(&W, &U, &V); fp_mul(&S, x, &V);
    fp_sqr(&M, x); fp_dbl(&t, &M); fp_add(&M, &M, &t);
    fp_sqr(&r->x, &M); fp_sub(&r->x, &r->x, &S); fp_sub(&r->x, &r->x, &S);
    fp_sub(&t, &S, &r->x); fp_mul(&t, &M, &t);
    fp_mul(&U, &W, y); fp_sub(&r->y, &t, &U);
    r->zz = V; r->zzz = W;
}
/* r += q (neg: r -= q), q affine: madd-2008-s, 8M + 2S */
static void g1x_madd(g1x *r, const g1a *q, int neg) {
    if (q->inf) return;
    fp qy = q->y;
    if (neg) fp_neg(&qy, &q->y);
    if (g1x_is_inf(r)) { r->x = q->x; r->y = qy; r->zz = FP_ONE; r->zzz = FP_ONE; return; }
    fp U2, S2, P, R, PP, PPP, Q, t;
    fp_mul(&U2, &q->x, &r->zz); fp_mul(&S2, &qy, &r->zzz);
    fp_sub(&P, &U2, &r->x); fp_sub(&R, &S2, &r->y);
    if (fp_is_zero(&P)) {
        if (fp_is_zero(&R)) g1x_double_affine(r, &q->x, &qy); else g1x_set_inf(r);
        return;
    }
    fp_sqr(&PP, &P); fp_mul(&PPP, &P, &PP); fp_mul(&Q, &r->x, &PP);
    fp_sqr(&t, &R); fp_sub(&t, &t, &PPP); fp_sub(&t, &t, &Q); fp_sub(&t, &t, &Q);          /* X3 */
    fp_sub(&Q, &Q, &t); fp_mul(&Q, &R, &Q); fp_mul(&S2, &r->y, &PPP); fp_sub(&r->y, &Q, &S2);
    r->x = t;
    fp_mul(&r->zz, &r->zz, &PP); fp_mul(&r->zzz, &r->zzz, &PPP);
}
static void g1x_to_jac(g1j *r, const g1x *p) { /* (X ZZ, Y ZZZ, ZZ): X ZZ / ZZ^2 = x, Y ZZZ / ZZ^3 = Y / ZZZ = y */
    if (g1x_is_inf(p)) { g1j_set_inf(r); return; }
    fp_mul(&r->x, &p->x, &p->zz); fp_mul(&r->y, &p->y, &p->zzz); r->z = p->zz;
}
/* r += q, both XYZZ: add-2008-s, 12M + 2S */
static void g1x_add(g1x *r, const g1x *q) {
    if (g1x_is_inf(q)) return;
    if (g1x_is_inf(r)) { *r = *q; return; }
    fp U1, U2, S1, S2, P, R, PP, PPP, Q, t;
    fp_mul(&U1, &r->x, &q->zz); fp_mul(&U2, &q->x, &r->zz);
    fp_mul(&S1, &r->y, &q->zzz); fp_mul(&S2, &q->y, &r->zzz);
    fp_sub(&P, &U2, &U1); fp_sub(&R, &S2, &S1);
    if (fp_is_zero(&P)) {
        if (!fp_is_zero(&R)) { g1x_set_inf(r); return; }
        g1j a, d; g1x_to_jac(&a, r); g1j_double(&d, &a);                 /* rare: equal points */
        if (g1j_is_inf(&d)) { g1x_set_inf(r); return; }
        fp z2; fp_sqr(&z2, &d.z); r->x = d.x; r->y = d.y; r->zz = z2; fp_mul(&r->zzz, &z2, &d.z);
        return;
    }
    fp_sqr(&PP, &P); fp_mul(&PPP, &P, &PP); fp_mul(&Q, &U1, &PP);
    fp_sqr(&t, &R); fp_sub(&t, &t, &PPP); fp_sub(&t, &t, &Q); fp_sub(&t, &t, &Q);
    fp_sub(&Q, &Q, &t); fp_mul(&Q, &R, &Q); fp_mul(&S1, &S1, &PPP); fp_sub(&r->y, &Q, &S1);
    r->x = t;
    fp_mul(&t, &r->zz, &q->zz); fp_mul(&r->zz, &t, &PP);
    fp_mul(&t, &r->zzz, &q->zzz); fp_mul(&r->zzz, &t, &PPP);
}

static void g1j_to_affine(g1a *r, const g1j *p) {
    if (g1j_is_inf(p)) { memset(r, 0, sizeof(*r)); r->inf = 1; return; }
    fp zi, zi2;
    fp_inv(&zi, &p->z); fp_sqr(&zi2, &zi);
    fp_mul(&r->x, &p->x, &zi2);
    fp_mul(&zi2, &zi2, &zi); fp_mul(&r->y, &p->y, &zi2);
    r->inf = 0;
}
/* batch normalisation with Montgomery's trick */
static void g1j_batch_to_affine(g1a *out, const g1j *in, size_t n) {
    fp *pre = (fp *)malloc((n + 1) * sizeof(fp));
    pre[0] = FP_ONE;
    for (size_t i = 0; i < n; i++) {
        if (g1j_is_inf(&in[i])) pre[i + 1] = pre[i]; else fp_mul(&pre[i + 1], &pre[i], &in[i].z);
    }
    fp inv; fp_inv(&inv, &pre[n]);
    for (size_t i = n; i-- > 0;) {
        if (g1j_is_inf(&in[i])) { memset(&out[i], 0, sizeof(g1a)); out[i].inf = 1; continue; }
        fp zi, zi2;
        fp_mul(&zi, &inv, &pre[i]);
        fp_mul(&inv, &inv, &in[i].z);
        fp_sqr(&zi2, &zi);
        fp_mul(&out[i].x, &in[i].x, &zi2);
        fp_mul(&zi2, &zi2, &zi); fp_mul(&out[i].y, &in[i].y, &zi2);
        out[i].inf = 0;
    }
    free(pre);
}
static void g1a_from_be96(g1a *r, const uint8_t *b) {
    int z = 1; for (int i = 0; i < 96; i++) if (b[i]) { z = 0; break; }
    if (z) { memset(r, 0, sizeof(*r)); r->inf = 1; return; }
    fp_from_be(&r->x, b); fp_from_be(&r->y, b + 48); r->inf = 0;
}
static void g1a_to_be96(uint8_t *b, const g1a *a) {
    if (a->inf) { memset(b, 0, 96); return; }
    fp_to_be(b, &a->x); fp_to_be(b + 48, &a->y);
}
static void g1a_compress(uint8_t *out, const g1a *a) {
    if (a->inf) { memset(out, 0, 48); out[0] = 0xC0; return; }
    u64 y[NP], half[NP], one[NP] = {1, 0, 0, 0, 0, 0};
    fp_to_be(out, &a->x);
    fp_to_limbs(y, &a->y);
    limbs_sub(half, P_MOD, one, NP);
    for (int i = 0; i < NP; i++) half[i] = (half[i] >> 1) | (i + 1 < NP ? half[i + 1] << 63 : 0);
    out[0] |= 0x80;
    if (!limbs_ge(half, y, NP)) out[0] |= 0x20; /* y > (p-1)/2 */
}

/* ------------------------------------------------------------------ init */
static void orc_init(void) {
    if (g_init) return;
    P_INV = neg_inv64(P_MOD[0]); R_INV = neg_inv64(R_MOD[0]);
    pow2_mod(FP_ONE.l, 384, P_MOD, NP); pow2_mod(FP_R2.l, 768, P_MOD, NP);
    pow2_mod(FR_ONE.l, 256, R_MOD, NR); pow2_mod(FR_R2.l, 512, R_MOD, NR);
    static const uint8_t gx[48] = {0x17,0xf1,0xd3,0xa7,0x31,0x97,0xd7,0x94,0x26,0x95,0x63,0x8c,0x4f,0xa9,0xac,0x0f,
        0xc3,0x68,0x8c,0x4f,0x97,0x74,0xb9,0x05,0xa1,0x4e,0x3a,0x3f,0x17,0x1b,0xac,0x58,
        0x6c,0x55,0xe8,0x3f,0xf9,0x7a,0x1a,0xef,0xfb,0x3a,0xf0,0x0a,0xdb,0x22,0xc6,0xbb};
    static const uint8_t gy[48] = {0x08,0xb3,0xf4,0x81,0xe3,0xaa,0xa0,0xf1,0xa0,0x9e,0x30,0xed,0x74,0x1d,0x8a,0xe4,
        0xfc,0xf5,0xe0,0x95,0xd5,0xd0,0x0a,0xf6,0x00,0xdb,0x18,0xcb,0x2c,0x04,0xb3,0xed,
        0xd0,0x3c,0xc7,0x44,0xa2,0x88,0x8a,0xe4,0x0c,0xaa,0x23,0x29,0x46,0xc5,0xe7,0xe1};
    fp_from_be(&G1_GEN.x, gx); fp_from_be(&G1_GEN.y, gy); G1_GEN.inf = 0;
    g_init = 1;
}

static int usable_cpus(void);
/* ------------------------------------------------------------------ a tiny fork-join helper: fn(arg, t, nthreads) on t = 0..nthreads-1 */
typedef struct { void (*fn)(void *, int, int); void *arg; int t, n; } fj_item;
static void *fj_tramp(void *p) { fj_item *it = (fj_item *)p; it->fn(it->arg, it->t, it->n); return NULL; }
static void fork_join(void (*fn)(void *, int, int), void *arg, int threads) {
    if (threads <= 1) { fn(arg, 0, 1); return; }
    fj_item *it = (fj_item *)calloc(threads, sizeof(fj_item));
    pthread_t *th = (pthread_t *)calloc(threads, sizeof(pthread_t));
    int started = 0;
    for (int t = 1; t < threads; t++) {
        it[t].fn = fn; it[t].arg = arg; it[t].t = t; it[t].n = threads;
        if (pthread_create(&th[t], NULL, fj_tramp, &it[t])) break;
        started = t;
    }
    if (started + 1 < threads) {         /* could not start them all: run the missing shares here */
        for (int t = started + 1; t < threads; t++) fn(arg, t, threads);
    }
    fn(arg, 0, threads);
    for (int t = 1; t <= started; t++) pthread_join(th[t], NULL);
    free(it); free(th);
}

/* ------------------------------------------------------------------ NTT (radix-2 CT, natural order), stages split over threads */
typedef struct { fr *a; fr *tw; size_t n; int logn; fr wn; pthread_barrier_t bar; } ntt_ctx;
static void ntt_share(void *arg, int t, int nt) {
    ntt_ctx *c = (ntt_ctx *)arg;
    const size_t n = c->n, half_n = n / 2;
    /* twiddles w^k for this thread's range of k, started from w^lo by one exponentiation */
    size_t lo = half_n * (size_t)t / nt, hi = half_n * (size_t)(t + 1) / nt;
    if (lo < hi) {
        u64 e[NR] = {lo, 0, 0, 0};
        fr cur; fr_pow(&cur, &c->wn, e, NR);
        for (size_t k = lo; k < hi; k++) { c->tw[k] = cur; fr_mul(&cur, &cur, &c->wn); }
    }
    /* bit reversal: the pair (i, j) is swapped by the thread owning the smaller index */
    lo = n * (size_t)t / nt; hi = n * (size_t)(t + 1) / nt;
    for (size_t i = lo; i < hi; i++) {
        size_t j = 0;
        for (int b = 0; b < c->logn; b++) if (i >> b & 1) j |= (size_t)1 << (c->logn - 1 - b);
        if (i < j) { fr x = c->a[i]; c->a[i] = c->a[j]; c->a[j] = x; }
    }
    pthread_barrier_wait(&c->bar);
    lo = half_n * (size_t)t / nt; hi = half_n * (size_t)(t + 1) / nt;
    for (size_t len = 2; len <= n; len <<= 1) {
        const size_t half = len / 2, step = n / len;
        for (size_t bf = lo; bf < hi; bf++) {            /* butterfly bf of this stage */
            const size_t k = bf & (half - 1), s0 = (bf - k) * 2;
            fr u = c->a[s0 + k], v;
            fr_mul(&v, &c->a[s0 + k + half], &c->tw[k * step]);
            fr_add(&c->a[s0 + k], &u, &v); fr_sub(&c->a[s0 + k + half], &u, &v);
        }
        pthread_barrier_wait(&c->bar);
    }
}
typedef struct { fr *a; size_t n; fr f; } scale_ctx;
static void scale_share(void *arg, int t, int nt) {
    scale_ctx *c = (scale_ctx *)arg;
    for (size_t i = c->n * (size_t)t / nt; i < c->n * (size_t)(t + 1) / nt; i++) fr_mul(&c->a[i], &c->a[i], &c->f);
}
static void fr_ntt_inplace_mt(fr *a, size_t n, int inverse, int threads) {
    if (n <= 1) return;
    int logn = 0; while (((size_t)1 << logn) < n) logn++;
    if (threads < 1) threads = 1;
    if (threads > usable_cpus()) threads = usable_cpus();
    while (threads > 1 && (size_t)threads * 1024 > n) threads--;     /* at least ~1000 butterflies per thread and stage */
    ntt_ctx c; c.a = a; c.n = n; c.logn = logn;
    fr_root_of_unity(&c.wn, logn);
    if (inverse) fr_inv(&c.wn, &c.wn);
    c.tw = (fr *)malloc((n / 2) * sizeof(fr));
    pthread_barrier_init(&c.bar, NULL, (unsigned)threads);
    fork_join(ntt_share, &c, threads);
    pthread_barrier_destroy(&c.bar);
    free(c.tw);
    if (inverse) {
        scale_ctx sc; sc.a = a; sc.n = n;
        fr_from_u64(&sc.f, (u64)n); fr_inv(&sc.f, &sc.f);
        fork_join(scale_share, &sc, threads);
    }
}
static void fr_ntt_inplace(fr *a, size_t n, int inverse) { fr_ntt_inplace_mt(a, n, inverse, 1); }

/* ------------------------------------------------------------------ Pippenger
 * Booth-recoded signed windows of c bits: digit d in [-2^(c-1), 2^(c-1)], |d| selects the bucket, the sign negates the
 * point.  The work is cut into tasks (window w, chunk k of the points); a task fills its own 2^(c-1) XYZZ buckets with
 * mixed additions and reduces them by the running sum  sum_b b B_b = sum_b (sum_{b' >= b} B_b').  c and the number of
 * chunks come from a cost model (mixed additions n/K + 2^c bucket additions per task, ceil(tasks / threads) rounds), so
 * that few threads use wide windows and many threads still find >= `threads` tasks of useful size -- the old scheme gave
 * each thread a whole Pippenger on n/threads points and got SLOWER beyond 16 threads.  Window sums are added per window
 * and combined by c doublings per window at the end. */
static unsigned get_window(const u64 *s, int lo, int c) {   /* plain unsigned window (fixed-base table of orc_srs_gen) */
    int w = lo / 64, b = lo % 64;
    u64 v = s[w] >> b;
    if (b + c > 64 && w + 1 < NR) v |= s[w + 1] << (64 - b);
    return (unsigned)(v & (((u64)1 << c) - 1));
}
static int booth_digit(const u64 *s, int w, int c) {        /* bits [w c - 1, w c + c) of the scalar, bit -1 = 0 */
    const int lo = w * c - 1;
    u64 v;
    if (lo < 0) v = s[0] << 1;
    else {
        const int q = lo / 64, b = lo % 64;
        v = q < NR ? s[q] >> b : 0;
        if (b + c + 1 > 64 && q + 1 < NR) v |= s[q + 1] << (64 - b);
    }
    v &= ((u64)1 << (c + 1)) - 1;
    const int neg = (int)(v >> c) & 1;
    int d = (int)((v + 1) >> 1);
    return neg ? d - (1 << c) : d;                           /* = ((v + 1) >> 1) - 2^c * top_bit */
}
typedef struct {
    const g1a *pts; const u64 (*sc)[NR]; size_t n;
    int c, nwin, chunks;
    g1j *partial;               /* [nwin][chunks] */
    atomic_int next;
} msm_plan;
/* Bucket filling by BATCHED AFFINE additions (what the fast CPU provers do): buckets are affine points; a batch collects up
 * to `batch` additions bucket += point that touch DISTINCT buckets, inverts all their denominators x2 - x1 with one
 * field inversion (Montgomery's trick: 3 products per element), and finishes each addition with 2M + 1S -- ~6.6 products
 * per addition instead of the 10 of a mixed XYZZ addition.  A point whose bucket is already in the open batch waits in a
 * pending list; if that list outgrows a batch (skewed scalars: many points on few buckets) its entries go into a second,
 * XYZZ bucket set by ordinary mixed additions, so the worst case costs what the plain method costs.  The running-sum
 * reduction reads both sets. */
enum { BA_BATCH_MAX = 4096 };   /* a batch holds at most a quarter of the bucket count (few collisions), 256 .. 4096 */
typedef struct { fp x, y; } apoint;
typedef struct { uint32_t b; int kind; apoint q; fp den; } ba_op;    /* kind 0 add, 1 double, 2 cancel (P + -P) */
typedef struct {
    apoint *bk; unsigned char *full, *open;      /* affine buckets, occupancy, "in the open batch" */
    g1x *spill; int have_spill; size_t nb;
    ba_op *ops; int n_ops, batch;
    fp *pre;
    uint32_t *pend_b; apoint *pend_q; int n_pend, cap_pend;
} ba_state;
static void ba_flush(ba_state *st) {
    const int n = st->n_ops;
    if (!n) return;
    st->pre[0] = FP_ONE;
    for (int i = 0; i < n; i++) {
        ba_op *o = &st->ops[i];
        const apoint *p = &st->bk[o->b];
        if (fp_eq(&p->x, &o->q.x)) {
            if (fp_eq(&p->y, &o->q.y) && !fp_is_zero(&p->y)) { o->kind = 1; fp_dbl(&o->den, &p->y); }
            else { o->kind = 2; o->den = FP_ONE; }
        } else { o->kind = 0; fp_sub(&o->den, &o->q.x, &p->x); }
        fp_mul(&st->pre[i + 1], &st->pre[i], &o->den);
    }
    fp inv; fp_inv(&inv, &st->pre[n]);
    for (int i = n - 1; i >= 0; i--) {
        ba_op *o = &st->ops[i];
        apoint *p = &st->bk[o->b];
        fp dinv, lam, t, x3;
        fp_mul(&dinv, &inv, &st->pre[i]);
        fp_mul(&inv, &inv, &o->den);
        st->open[o->b] = 0;
        if (o->kind == 2) { st->full[o->b] = 0; continue; }
        if (o->kind == 1) { fp_sqr(&t, &p->x); fp_dbl(&lam, &t); fp_add(&t, &lam, &t); }   /* 3 x^2 */
        else fp_sub(&t, &o->q.y, &p->y);
        fp_mul(&lam, &t, &dinv);
        fp_sqr(&x3, &lam); fp_sub(&x3, &x3, &p->x); fp_sub(&x3, &x3, &o->q.x);
        fp_sub(&t, &p->x, &x3); fp_mul(&t, &lam, &t); fp_sub(&p->y, &t, &p->y);
        p->x = x3;
    }
    st->n_ops = 0;
}
static void ba_spill(ba_state *st, uint32_t b, const apoint *q) {
    if (!st->have_spill) {
        st->spill = (g1x *)calloc(st->nb + 1, sizeof(g1x));
        st->have_spill = 1;
    }
    g1a a; a.x = q->x; a.y = q->y; a.inf = 0;
    g1x_madd(&st->spill[b], &a, 0);
}
/* bucket b += q, now or in the open batch; returns 0 when it has to wait (bucket already in the batch) */
static int ba_try(ba_state *st, uint32_t b, const apoint *q) {
    if (st->open[b]) return 0;
    if (!st->full[b]) { st->bk[b] = *q; st->full[b] = 1; return 1; }
    ba_op *o = &st->ops[st->n_ops++];
    o->b = b; o->q = *q;
    st->open[b] = 1;
    return 1;
}
static void ba_drain_pending(ba_state *st) {      /* after a flush: retry what waited; what still collides keeps waiting */
    int kept = 0;
    for (int i = 0; i < st->n_pend; i++) {
        if (st->n_ops < st->batch && ba_try(st, st->pend_b[i], &st->pend_q[i])) continue;
        st->pend_b[kept] = st->pend_b[i]; st->pend_q[kept] = st->pend_q[i]; kept++;
    }
    st->n_pend = kept;
}
static void ba_add(ba_state *st, uint32_t b, const apoint *q) {
    if (!ba_try(st, b, q)) {
        if (st->n_pend == st->cap_pend) {                /* skew: stop batching these, add them the plain way */
            for (int i = 0; i < st->n_pend; i++) ba_spill(st, st->pend_b[i], &st->pend_q[i]);
            st->n_pend = 0;
        }
        st->pend_b[st->n_pend] = b; st->pend_q[st->n_pend] = *q; st->n_pend++;
        return;
    }
    if (st->n_ops == st->batch) { ba_flush(st); ba_drain_pending(st); }
}
static void msm_task(const msm_plan *pl, int task, ba_state *st) {
    const int w = task / pl->chunks, k = task % pl->chunks, c = pl->c;
    const size_t lo = pl->n * (size_t)k / pl->chunks, hi = pl->n * (size_t)(k + 1) / pl->chunks;
    const size_t nb = (size_t)1 << (c - 1);
    memset(st->full, 0, nb + 1);
    memset(st->open, 0, nb + 1);
    if (st->have_spill) memset(st->spill, 0, (nb + 1) * sizeof(g1x));
    st->n_ops = st->n_pend = 0;
    enum { AHEAD = 8 };           /* the bucket of point i + AHEAD is requested while point i is handled */
    int dq[AHEAD];
    for (size_t i = lo; i < hi && i < lo + AHEAD; i++) dq[i - lo] = booth_digit(pl->sc[i], w, c);
    for (size_t i = lo; i < hi; i++) {
        const int d = dq[(i - lo) % AHEAD];
        if (i + AHEAD < hi) {
            const int dn = booth_digit(pl->sc[i + AHEAD], w, c);
            dq[(i - lo) % AHEAD] = dn;
            __builtin_prefetch(&st->bk[dn < 0 ? -dn : dn]);
            __builtin_prefetch((const char *)&st->bk[dn < 0 ? -dn : dn] + 64);
        }
        if (!d || pl->pts[i].inf) continue;
        apoint q; q.x = pl->pts[i].x;
        if (d < 0) fp_neg(&q.y, &pl->pts[i].y); else q.y = pl->pts[i].y;
        ba_add(st, (uint32_t)(d < 0 ? -d : d), &q);
    }
    for (;;) {                                            /* close the last batch and whatever waited for it */
        ba_flush(st);
        if (!st->n_pend) break;
        const int before = st->n_pend;
        ba_drain_pending(st);
        if (st->n_pend == before && st->n_ops == 0) break;   /* cannot happen: a closed batch frees every bucket */
    }
    g1x run, acc; g1x_set_inf(&run); g1x_set_inf(&acc);
    for (size_t b = nb; b >= 1; b--) {
        if (st->full[b]) { g1a a; a.x = st->bk[b].x; a.y = st->bk[b].y; a.inf = 0; g1x_madd(&run, &a, 0); }
        if (st->have_spill) g1x_add(&run, &st->spill[b]);
        g1x_add(&acc, &run);
    }
    g1x_to_jac(&pl->partial[task], &acc);
}
static void msm_worker(void *arg, int t, int nt) {
    (void)t; (void)nt;
    msm_plan *pl = (msm_plan *)arg;
    const size_t nb = (size_t)1 << (pl->c - 1);
    ba_state st; memset(&st, 0, sizeof(st));
    st.nb = nb;
    st.bk = (apoint *)malloc((nb + 1) * sizeof(apoint));
    st.full = (unsigned char *)malloc(nb + 1);
    st.open = (unsigned char *)malloc(nb + 1);
    st.batch = (int)(nb / 4 < 256 ? 256 : nb / 4 > BA_BATCH_MAX ? BA_BATCH_MAX : nb / 4);
    st.ops = (ba_op *)malloc((size_t)st.batch * sizeof(ba_op));
    st.pre = (fp *)malloc(((size_t)st.batch + 1) * sizeof(fp));
    st.cap_pend = st.batch;
    st.pend_b = (uint32_t *)malloc(st.cap_pend * sizeof(uint32_t));
    st.pend_q = (apoint *)malloc(st.cap_pend * sizeof(apoint));
    const int ntask = pl->nwin * pl->chunks;
    for (;;) {
        const int task = atomic_fetch_add(&pl->next, 1);
        if (task >= ntask) break;
        msm_task(pl, task, &st);
    }
    free(st.bk); free(st.full); free(st.open); free(st.ops); free(st.pre); free(st.pend_b); free(st.pend_q);
    if (st.have_spill) free(st.spill);
}
static void msm_choose(size_t n, int threads, int *out_c, int *out_chunks) {
    double best = -1; int bc = 2, bk = 1;
    for (int c = 2; c <= 16; c++) {
        const int nwin = (256 + c - 1) / c;
        int k = (threads + nwin - 1) / nwin;                 /* enough tasks for every thread */
        if (k < 1) k = 1;
        if ((size_t)k > n) k = n ? (int)n : 1;
        /* per bucket: one mixed (10 M) + one full (14 M) addition; per point: one batched-affine addition (~6.6 M) */
        const double per_task = (double)n / k + (24.0 / 6.6) * (double)((size_t)1 << (c - 1));
        const int rounds = (nwin * k + threads - 1) / threads;
        const double cost = rounds * per_task;
        if (best < 0 || cost < best) { best = cost; bc = c; bk = k; }
    }
    *out_c = bc; *out_chunks = bk;
}
/* CPUs this process may actually use: the affinity mask, cut down to the cgroup's CPU quota when there is one (a container
 * that SEES 256 cores but is allowed 16 gains nothing from 256 threads -- and a plan made for 256-way parallelism, with
 * narrower windows and smaller chunks, is a worse plan for the 16 that run) */
static int usable_cpus(void) {
    static int cached = 0;
    if (cached) return cached;
    int n = 0;
#ifdef __linux__
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof(set), &set) == 0) n = CPU_COUNT(&set);
    FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r");                  /* cgroup v2: "<quota|max> <period>" */
    if (f) {
        char q[32]; long period = 0;
        if (fscanf(f, "%31s %ld", q, &period) == 2 && q[0] != 'm' && period > 0) {
            const long quota = atol(q);
            const int lim = (int)((quota + period - 1) / period);
            if (lim >= 1 && (n == 0 || lim < n)) n = lim;
        }
        fclose(f);
    } else {
        long quota = -1, period = 0;                                  /* cgroup v1 */
        if ((f = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r"))) { if (fscanf(f, "%ld", &quota) != 1) quota = -1; fclose(f); }
        if ((f = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r"))) { if (fscanf(f, "%ld", &period) != 1) period = 0; fclose(f); }
        if (quota > 0 && period > 0) {
            const int lim = (int)((quota + period - 1) / period);
            if (lim >= 1 && (n == 0 || lim < n)) n = lim;
        }
    }
#endif
    if (n < 1) n = 1;
    cached = n;
    return n;
}
int orc_usable_cpus(void) { return usable_cpus(); }
static void msm_mt(g1j *out, const g1a *pts, const u64 (*sc)[NR], size_t n, int threads) {
    g1j total; g1j_set_inf(&total);
    if (n == 0) { *out = total; return; }
    if (threads < 1) threads = 1;
    if (threads > usable_cpus()) threads = usable_cpus();
    msm_plan pl; pl.pts = pts; pl.sc = sc; pl.n = n;
    msm_choose(n, threads, &pl.c, &pl.chunks);
    pl.nwin = (256 + pl.c - 1) / pl.c;
    const int ntask = pl.nwin * pl.chunks;
    pl.partial = (g1j *)malloc((size_t)ntask * sizeof(g1j));
    atomic_init(&pl.next, 0);
    fork_join(msm_worker, &pl, threads < ntask ? threads : ntask);
    for (int w = pl.nwin - 1; w >= 0; w--) {
        for (int k = 0; k < pl.c; k++) g1j_double(&total, &total);
        for (int k = 0; k < pl.chunks; k++) g1j_add(&total, &total, &pl.partial[w * pl.chunks + k]);
    }
    free(pl.partial);
    *out = total;
}

/* ------------------------------------------------------------------ exported C API (ctypes) */
/* scalar multiple of the generator: out48 = compress([k]G) */
int orc_g1_mul_gen(const uint8_t k_be32[32], uint8_t out48[48]) {
    orc_init();
    u64 k[NR]; be_to_limbs(k, k_be32, NR);
    g1j acc; g1j_set_inf(&acc);
    for (int i = 255; i >= 0; i--) {
        g1j_double(&acc, &acc);
        if ((k[i / 64] >> (i % 64)) & 1) g1j_add_affine(&acc, &acc, &G1_GEN);
    }
    g1a a; g1j_to_affine(&a, &acc); g1a_compress(out48, &a);
    return 0;
}
int orc_g1_compress(const uint8_t in_be96[96], uint8_t out48[48]) {
    orc_init(); g1a a; g1a_from_be96(&a, in_be96); g1a_compress(out48, &a); return 0;
}
/* out = compress(sum of n affine points) */
int orc_g1_sum(const uint8_t *pts_be96, uint64_t n, uint8_t out48[48]) {
    orc_init();
    g1j acc; g1j_set_inf(&acc);
    for (uint64_t i = 0; i < n; i++) { g1a a; g1a_from_be96(&a, pts_be96 + 96 * i); g1j_add_affine(&acc, &acc, &a); }
    g1a r; g1j_to_affine(&r, &acc); g1a_compress(out48, &r);
    return 0;
}
int orc_fr_ntt(uint8_t *inout_be32, uint64_t n, int inverse) {
    orc_init();
    if (n & (n - 1)) return -1;
    fr *a = (fr *)malloc(n * sizeof(fr));
    for (uint64_t i = 0; i < n; i++) if (fr_from_be(&a[i], inout_be32 + 32 * i)) { free(a); return -2; }
    fr_ntt_inplace(a, n, inverse);
    for (uint64_t i = 0; i < n; i++) fr_to_be(inout_be32 + 32 * i, &a[i]);
    free(a);
    return 0;
}
int orc_fr_eval(const uint8_t *coeffs_be32, uint64_t n, const uint8_t x_be32[32], uint8_t out_be32[32]) {
    orc_init();
    fr x, acc, c; memset(&acc, 0, sizeof(acc));
    if (fr_from_be(&x, x_be32)) return -2;
    for (uint64_t i = n; i-- > 0;) {
        if (fr_from_be(&c, coeffs_be32 + 32 * i)) return -2;
        fr_mul(&acc, &acc, &x); fr_add(&acc, &acc, &c);
    }
    fr_to_be(out_be32, &acc);
    return 0;
}
/* worker i's slice U_{i,j} = tau_x^j L_i(tau_y) G, j < T = 2^(scale - machines_scale); out: T x 96 B */
int orc_srs_gen(const uint8_t tau_x_be32[32], const uint8_t tau_y_be32[32], int scale, int machines_scale,
                uint32_t i, uint8_t *out_be96) {
    orc_init();
    if (machines_scale < 0 || scale < machines_scale || scale - machines_scale > 30) return -1;
    size_t T = (size_t)1 << (scale - machines_scale), M = (size_t)1 << machines_scale;
    if (i >= M) return -1;
    fr tx, ty, li;
    if (fr_from_be(&tx, tau_x_be32) || fr_from_be(&ty, tau_y_be32)) return -2;
    /* L_i(tau_y) = w^i/M * (tau_y^M - 1)/(tau_y - w^i) */
    if (M == 1) li = FR_ONE;
    else {
        fr w, wi, num, den, minv; fr_root_of_unity(&w, machines_scale);
        u64 e[NR] = {i, 0, 0, 0}; fr_pow(&wi, &w, e, NR);
        fr_sub(&den, &ty, &wi);
        if (limbs_is_zero(den.l, NR)) li = FR_ONE;
        else {
            u64 em[NR] = {M, 0, 0, 0}; fr_pow(&num, &ty, em, NR); fr_sub(&num, &num, &FR_ONE);
            fr_from_u64(&minv, (u64)M); fr_inv(&minv, &minv); fr_inv(&den, &den);
            fr_mul(&li, &wi, &minv); fr_mul(&li, &li, &num); fr_mul(&li, &li, &den);
        }
    }
    /* fixed-base table: 32 windows x 255 multiples of G */
    enum { W = 8, NW = 32, TS = 255 };
    g1j *tj = (g1j *)malloc(sizeof(g1j) * NW * TS);
    g1j cur; g1j_from_affine(&cur, &G1_GEN);
    for (int w = 0; w < NW; w++) {
        g1j acc = cur;
        for (int d = 0; d < TS; d++) { tj[w * TS + d] = acc; g1j_add(&acc, &acc, &cur); }
        for (int s = 0; s < W; s++) g1j_double(&cur, &cur);
    }
    g1a *tab = (g1a *)malloc(sizeof(g1a) * NW * TS);
    g1j_batch_to_affine(tab, tj, NW * TS);
    free(tj);
    g1j *res = (g1j *)malloc(sizeof(g1j) * T);
    fr s = li;
    for (size_t j = 0; j < T; j++) {
        u64 k[NR]; fr_to_limbs(k, &s);
        g1j acc; g1j_set_inf(&acc);
        for (int w = 0; w < NW; w++) {
            unsigned d = get_window(k, w * W, W);
            if (d) g1j_add_affine(&acc, &acc, &tab[w * TS + d - 1]);
        }
        res[j] = acc;
        fr_mul(&s, &s, &tx);
    }
    g1a *aff = (g1a *)malloc(sizeof(g1a) * T);
    g1j_batch_to_affine(aff, res, T);
    for (size_t j = 0; j < T; j++) g1a_to_be96(out_be96 + 96 * j, &aff[j]);
    free(aff); free(res); free(tab);
    return 0;
}
static int load_inputs(g1a **pts, u64 (**sc)[NR], const uint8_t *p96, const uint8_t *s32, uint64_t n) {
    *pts = (g1a *)malloc((n ? n : 1) * sizeof(g1a));
    *sc = (u64(*)[NR])malloc((n ? n : 1) * sizeof(u64[NR]));
    for (uint64_t i = 0; i < n; i++) {
        g1a_from_be96(&(*pts)[i], p96 + 96 * i);
        be_to_limbs((*sc)[i], s32 + 32 * i, NR);
        if (limbs_ge((*sc)[i], R_MOD, NR)) { free(*pts); free(*sc); return -2; }
    }
    return 0;
}
/* out48 = compress(sum_j scalars[j] * points[j]) */
int orc_msm(const uint8_t *points_be96, const uint8_t *scalars_be32, uint64_t n, int threads, uint8_t out48[48]) {
    orc_init();
    g1a *pts; u64 (*sc)[NR];
    int rc = load_inputs(&pts, &sc, points_be96, scalars_be32, n);
    if (rc) return rc;
    g1j r; msm_mt(&r, pts, (const u64(*)[NR])sc, n, threads);
    g1a a; g1j_to_affine(&a, &r); g1a_compress(out48, &a);
    free(pts); free(sc);
    return 0;
}
/* opaque preloaded MSM instance so the bench can time the MSM alone (inputs already decoded) */
typedef struct { g1a *pts; u64 (*sc)[NR]; uint64_t n; } orc_msm_inst;
void *orc_msm_prepare(const uint8_t *points_be96, const uint8_t *scalars_be32, uint64_t n) {
    orc_init();
    orc_msm_inst *m = (orc_msm_inst *)calloc(1, sizeof(*m));
    if (load_inputs(&m->pts, &m->sc, points_be96, scalars_be32, n)) { free(m); return NULL; }
    m->n = n; return m;
}
int orc_msm_run(void *inst, int threads, uint8_t out48[48]) {
    orc_msm_inst *m = (orc_msm_inst *)inst;
    g1j r; msm_mt(&r, m->pts, (const u64(*)[NR])m->sc, m->n, threads);
    g1a a; g1j_to_affine(&a, &r); g1a_compress(out48, &a);
    return 0;
}
void orc_msm_free(void *inst) { orc_msm_inst *m = (orc_msm_inst *)inst; if (m) { free(m->pts); free(m->sc); free(m); } }

static int load_coeffs(fr **out, const uint8_t *row, uint64_t T, int evaluation_form, int threads) {
    fr *a = (fr *)malloc((T ? T : 1) * sizeof(fr));
    for (uint64_t i = 0; i < T; i++) if (fr_from_be(&a[i], row + 32 * i)) { free(a); return -2; }
    if (evaluation_form) fr_ntt_inplace_mt(a, T, 1, threads);
    *out = a; return 0;
}
/* KZG worker commit: commitment = MSM(slice[0..T), IFFT(row)) */
int orc_commit(const uint8_t *slice_be96, const uint8_t *row_be32, uint64_t T, int evaluation_form, int threads,
               uint8_t out48[48]) {
    orc_init();
    if (T == 0 || (T & (T - 1))) return -1;
    fr *a; int rc = load_coeffs(&a, row_be32, T, evaluation_form, threads); if (rc) return rc;
    g1a *pts = (g1a *)malloc(T * sizeof(g1a)); u64 (*sc)[NR] = (u64(*)[NR])malloc(T * sizeof(u64[NR]));
    for (uint64_t i = 0; i < T; i++) { g1a_from_be96(&pts[i], slice_be96 + 96 * i); fr_to_limbs(sc[i], &a[i]); }
    g1j r; msm_mt(&r, pts, (const u64(*)[NR])sc, T, threads);
    g1a af; g1j_to_affine(&af, &r); g1a_compress(out48, &af);
    free(a); free(pts); free(sc);
    return 0;
}
/* KZG worker open at alpha: y = f(alpha), proof = MSM(slice[0..T-1), (f - y)/(X - alpha)) */
int orc_open(const uint8_t *slice_be96, const uint8_t *row_be32, uint64_t T, int evaluation_form,
             const uint8_t alpha_be32[32], int threads, uint8_t out_eval32[32], uint8_t out_proof48[48]) {
    orc_init();
    if (T == 0 || (T & (T - 1))) return -1;
    fr alpha; if (fr_from_be(&alpha, alpha_be32)) return -2;
    fr *a; int rc = load_coeffs(&a, row_be32, T, evaluation_form, threads); if (rc) return rc;
    g1a *pts = (g1a *)malloc(T * sizeof(g1a)); u64 (*sc)[NR] = (u64(*)[NR])malloc(T * sizeof(u64[NR]));
    fr acc = a[T - 1];
    for (uint64_t j = T - 1; j >= 1; j--) {
        fr_to_limbs(sc[j - 1], &acc);
        fr_mul(&acc, &acc, &alpha); fr_add(&acc, &acc, &a[j - 1]);
    }
    fr_to_be(out_eval32, &acc);
    for (uint64_t i = 0; i + 1 < T; i++) g1a_from_be96(&pts[i], slice_be96 + 96 * i);
    g1j r; msm_mt(&r, pts, (const u64(*)[NR])sc, T - 1, threads);
    g1a af; g1j_to_affine(&af, &r); g1a_compress(out_proof48, &af);
    free(a); free(pts); free(sc);
    return 0;
}
