// Montgomery prime fields over SATURATED 32-bit limbs, values fully reduced in [0, m): Fr (scalar field, 8 limbs,
// R = 2^256) for the NTT / opening kernels, and Fp (12 limbs, R = 2^384) as the reference implementation for the
// unit tests and the packed HBM format.  The MSM's working Fp is the unsaturated 14 x 28-bit form of fp28.hip.h.
// Constants were re-derived in oracle/bls12_381.py.
#pragma once
#include "bigint.hip.h"

struct FpParams {
    static constexpr int N = 12;
    static constexpr uint32_t INV = 0xfffcfffdu;  // -p^-1 mod 2^32
    __host__ __device__ static constexpr uint32_t mod(int i) {
        constexpr uint32_t m[12] = {0xffffaaabu, 0xb9feffffu, 0xb153ffffu, 0x1eabfffeu, 0xf6b0f624u, 0x6730d2a0u,
                                    0xf38512bfu, 0x64774b84u, 0x434bacd7u, 0x4b1ba7b6u, 0x397fe69au, 0x1a0111eau};
        return m[i];
    }
    // R mod p (Montgomery one)
    __host__ __device__ static constexpr uint32_t one(int i) {
        constexpr uint32_t m[12] = {0x0002fffdu, 0x76090000u, 0xc40c0002u, 0xebf4000bu, 0x53c758bau, 0x5f489857u,
                                    0x70525745u, 0x77ce5853u, 0xa256ec6du, 0x5c071a97u, 0xfa80e493u, 0x15f65ec3u};
        return m[i];
    }
    // R^2 mod p
    __host__ __device__ static constexpr uint32_t r2(int i) {
        constexpr uint32_t m[12] = {0x1c341746u, 0xf4df1f34u, 0x09d104f1u, 0x0a76e6a6u, 0x4c95b6d5u, 0x8de5476cu,
                                    0x939d83c0u, 0x67eb88a9u, 0xb519952du, 0x9a793e85u, 0x92cae3aau, 0x11988fe5u};
        return m[i];
    }
};

struct FrParams {
    static constexpr int N = 8;
    static constexpr uint32_t INV = 0xffffffffu;  // -r^-1 mod 2^32
    __host__ __device__ static constexpr uint32_t mod(int i) {
        constexpr uint32_t m[8] = {0x00000001u, 0xffffffffu, 0xfffe5bfeu, 0x53bda402u,
                                   0x09a1d805u, 0x3339d808u, 0x299d7d48u, 0x73eda753u};
        return m[i];
    }
    __host__ __device__ static constexpr uint32_t one(int i) {
        constexpr uint32_t m[8] = {0xfffffffeu, 0x00000001u, 0x00034802u, 0x5884b7fau,
                                   0xecbc4ff5u, 0x998c4fefu, 0xacc5056fu, 0x1824b159u};
        return m[i];
    }
    __host__ __device__ static constexpr uint32_t r2(int i) {
        constexpr uint32_t m[8] = {0xf3f29c6du, 0xc999e990u, 0x87925c23u, 0x2b6cedcbu,
                                   0x7254398fu, 0x05d31496u, 0x9f59ff11u, 0x0748d9d9u};
        return m[i];
    }
};

template <class P>
struct alignas(16) field_t {
    static constexpr int N = P::N;
    uint32_t l[N];
};

template <class P>
KZG_DEV void f_zero(field_t<P>& r) {
#pragma unroll
    for (int i = 0; i < P::N; i++) r.l[i] = 0;
}
template <class P>
KZG_DEV void f_one(field_t<P>& r) {
#pragma unroll
    for (int i = 0; i < P::N; i++) r.l[i] = P::one(i);
}
template <class P>
KZG_DEV bool f_is_zero(const field_t<P>& a) {
    return bi_is_zero<P::N>(a.l);
}
template <class P>
KZG_DEV bool f_eq(const field_t<P>& a, const field_t<P>& b) {
    return bi_eq<P::N>(a.l, b.l);
}
// conditional subtract of the modulus: r = (carry || t >= m) ? t - m : t
template <class P>
KZG_DEV void f_final_sub(uint32_t* r, const uint32_t* t, uint32_t carry) {
    constexpr int N = P::N;
    uint32_t d[N];
    uint32_t br = 0;
#pragma unroll
    for (int i = 0; i < N; i++) d[i] = __builtin_subc(t[i], P::mod(i), br, &br);
    bool take_d = carry || !br;
#pragma unroll
    for (int i = 0; i < N; i++) r[i] = take_d ? d[i] : t[i];
}
template <class P>
KZG_DEV void f_add(field_t<P>& r, const field_t<P>& a, const field_t<P>& b) {
    constexpr int N = P::N;
    uint32_t t[N];
    uint32_t c = bi_add<N>(t, a.l, b.l);
    f_final_sub<P>(r.l, t, c);
}
template <class P>
KZG_DEV void f_sub(field_t<P>& r, const field_t<P>& a, const field_t<P>& b) {
    constexpr int N = P::N;
    uint32_t t[N];
    uint32_t br = bi_sub<N>(t, a.l, b.l);
    uint32_t mask = 0u - br;
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < N; i++) r.l[i] = __builtin_addc(t[i], P::mod(i) & mask, c, &c);
}
template <class P>
KZG_DEV void f_neg(field_t<P>& r, const field_t<P>& a) {
    constexpr int N = P::N;
    bool z = f_is_zero(a);
    uint32_t br = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
        uint32_t d = __builtin_subc(P::mod(i), a.l[i], br, &br);
        r.l[i] = z ? 0u : d;
    }
}
template <class P>
KZG_DEV void f_dbl(field_t<P>& r, const field_t<P>& a) {
    f_add(r, a, a);
}

// CIOS Montgomery product.  Inputs < m, output < m.  Both moduli leave >= 1 spare bit in the top limb, so
// the running value never needs more than N+1 limbs.
template <class P>
KZG_DEV void f_mul_inline(field_t<P>& r, const field_t<P>& a, const field_t<P>& b) {
    constexpr int N = P::N;
    uint32_t t[N + 1];
#pragma unroll
    for (int i = 0; i <= N; i++) t[i] = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
        uint64_t c = 0;
        const uint32_t bi = b.l[i];
#pragma unroll
        for (int j = 0; j < N; j++) {
            uint64_t v = (uint64_t)a.l[j] * bi + t[j] + c;
            t[j] = (uint32_t)v;
            c = v >> 32;
        }
        t[N] += (uint32_t)c;
        const uint32_t q = t[0] * P::INV;
        uint64_t v = (uint64_t)q * P::mod(0) + t[0];
        c = v >> 32;
#pragma unroll
        for (int j = 1; j < N; j++) {
            v = (uint64_t)q * P::mod(j) + t[j] + c;
            t[j - 1] = (uint32_t)v;
            c = v >> 32;
        }
        v = (uint64_t)t[N] + c;
        t[N - 1] = (uint32_t)v;
        t[N] = (uint32_t)(v >> 32);
    }
    f_final_sub<P>(r.l, t, t[N]);
}

// ---------------------------------------------------------------------------------------------------
// Even/odd Montgomery product built directly on v_mad_u64_u32 (the only wide multiply on the CDNA4 VALU).
// The running value is held redundantly as
//     t = sum E[k] 2^(64k) + sum O[k] 2^(64k+32) + sum CE[k] 2^(64(k+1)) + sum CO[k] 2^(64(k+1)+32)
// E/O are 64-bit slots at even / odd limb offsets, so a_j*b_i lands on one slot with ONE mad (64-bit
// addend) and its carry-out is counted into CE/CO with ONE v_addc: 2 instructions per limb product and
// no 64-bit operand shuffling (the plain C++ CIOS above compiles to ~4.3/product).  The one-limb shift
// of each Montgomery round is a pure renaming E<->O.  Checked against the big-int oracle in
// tests/test_gpu_field.py; the algorithm itself was validated limb-exactly in Python first.
// One asm statement covers a half row (H products).  Two gfx950 rules shape it:
//  * hipcc pads every asm statement with an s_nop, so per-product statements would waste an issue slot each;
//  * a VALU write of an SGPR/VCC needs 2 wait states before a VALU read of it (hipcc pads its own carry
//    chains with s_nop 1; inside asm nobody does).  So every mad gets its OWN carry SGPR pair and the
//    v_addc that consumes it is issued H instructions later: the distance covers the hazard with useful work.
//  * slots and counters are EARLY-CLOBBER ("+&v"): the statement writes them while it still reads x / y / q,
//    and without the '&' hipcc may place an input whose value it proves equal to a slot's initial value in the
//    slot's own register (seen with constant-folded operands: silently wrong products).
template <int H>
struct mad_row;

template <>
struct mad_row<6> {
    // S[k] += x[2k] * y ; C[k] += carry
    static KZG_DEV void vec(uint64_t* S, uint32_t* C, const uint32_t* x, uint32_t y) {
        uint64_t cy[6];
        asm("v_mad_u64_u32 %0, %12, %18, %24, %0\n\t"
            "v_mad_u64_u32 %1, %13, %19, %24, %1\n\t"
            "v_mad_u64_u32 %2, %14, %20, %24, %2\n\t"
            "v_mad_u64_u32 %3, %15, %21, %24, %3\n\t"
            "v_mad_u64_u32 %4, %16, %22, %24, %4\n\t"
            "v_mad_u64_u32 %5, %17, %23, %24, %5\n\t"
            "v_addc_co_u32 %6, vcc, 0, %6, %12\n\t"
            "v_addc_co_u32 %7, vcc, 0, %7, %13\n\t"
            "v_addc_co_u32 %8, vcc, 0, %8, %14\n\t"
            "v_addc_co_u32 %9, vcc, 0, %9, %15\n\t"
            "v_addc_co_u32 %10, vcc, 0, %10, %16\n\t"
            "v_addc_co_u32 %11, vcc, 0, %11, %17\n\t"
            : "+&v"(S[0]), "+&v"(S[1]), "+&v"(S[2]), "+&v"(S[3]), "+&v"(S[4]), "+&v"(S[5]), "+&v"(C[0]), "+&v"(C[1]), "+&v"(C[2]), "+&v"(C[3]), "+&v"(C[4]), "+&v"(C[5]), "=&s"(cy[0]), "=&s"(cy[1]), "=&s"(cy[2]), "=&s"(cy[3]), "=&s"(cy[4]), "=&s"(cy[5])
            : "v"(x[0]), "v"(x[2]), "v"(x[4]), "v"(x[6]), "v"(x[8]), "v"(x[10]), "v"(y)
            : "vcc");
    }
    // S[k] += q * m_k with wave-uniform m_k (modulus limbs, SGPRs)
    static KZG_DEV void uni(uint64_t* S, uint32_t* C, uint32_t q, uint32_t m0, uint32_t m1, uint32_t m2,
                            uint32_t m3, uint32_t m4, uint32_t m5) {
        uint64_t cy[6];
        asm("v_mad_u64_u32 %0, %12, %18, %19, %0\n\t"
            "v_mad_u64_u32 %1, %13, %18, %20, %1\n\t"
            "v_mad_u64_u32 %2, %14, %18, %21, %2\n\t"
            "v_mad_u64_u32 %3, %15, %18, %22, %3\n\t"
            "v_mad_u64_u32 %4, %16, %18, %23, %4\n\t"
            "v_mad_u64_u32 %5, %17, %18, %24, %5\n\t"
            "v_addc_co_u32 %6, vcc, 0, %6, %12\n\t"
            "v_addc_co_u32 %7, vcc, 0, %7, %13\n\t"
            "v_addc_co_u32 %8, vcc, 0, %8, %14\n\t"
            "v_addc_co_u32 %9, vcc, 0, %9, %15\n\t"
            "v_addc_co_u32 %10, vcc, 0, %10, %16\n\t"
            "v_addc_co_u32 %11, vcc, 0, %11, %17\n\t"
            : "+&v"(S[0]), "+&v"(S[1]), "+&v"(S[2]), "+&v"(S[3]), "+&v"(S[4]), "+&v"(S[5]), "+&v"(C[0]), "+&v"(C[1]), "+&v"(C[2]), "+&v"(C[3]), "+&v"(C[4]), "+&v"(C[5]), "=&s"(cy[0]), "=&s"(cy[1]), "=&s"(cy[2]), "=&s"(cy[3]), "=&s"(cy[4]), "=&s"(cy[5])
            : "v"(q), "s"(m0), "s"(m1), "s"(m2), "s"(m3), "s"(m4), "s"(m5)
            : "vcc");
    }
};
template <>
struct mad_row<4> {
    static KZG_DEV void vec(uint64_t* S, uint32_t* C, const uint32_t* x, uint32_t y) {
        uint64_t cy[4];
        asm("v_mad_u64_u32 %0, %8, %12, %16, %0\n\t"
            "v_mad_u64_u32 %1, %9, %13, %16, %1\n\t"
            "v_mad_u64_u32 %2, %10, %14, %16, %2\n\t"
            "v_mad_u64_u32 %3, %11, %15, %16, %3\n\t"
            "v_addc_co_u32 %4, vcc, 0, %4, %8\n\t"
            "v_addc_co_u32 %5, vcc, 0, %5, %9\n\t"
            "v_addc_co_u32 %6, vcc, 0, %6, %10\n\t"
            "v_addc_co_u32 %7, vcc, 0, %7, %11\n\t"
            : "+&v"(S[0]), "+&v"(S[1]), "+&v"(S[2]), "+&v"(S[3]), "+&v"(C[0]), "+&v"(C[1]), "+&v"(C[2]), "+&v"(C[3]), "=&s"(cy[0]), "=&s"(cy[1]), "=&s"(cy[2]), "=&s"(cy[3])
            : "v"(x[0]), "v"(x[2]), "v"(x[4]), "v"(x[6]), "v"(y)
            : "vcc");
    }
    static KZG_DEV void uni(uint64_t* S, uint32_t* C, uint32_t q, uint32_t m0, uint32_t m1, uint32_t m2,
                            uint32_t m3, uint32_t, uint32_t) {
        uint64_t cy[4];
        asm("v_mad_u64_u32 %0, %8, %12, %13, %0\n\t"
            "v_mad_u64_u32 %1, %9, %12, %14, %1\n\t"
            "v_mad_u64_u32 %2, %10, %12, %15, %2\n\t"
            "v_mad_u64_u32 %3, %11, %12, %16, %3\n\t"
            "v_addc_co_u32 %4, vcc, 0, %4, %8\n\t"
            "v_addc_co_u32 %5, vcc, 0, %5, %9\n\t"
            "v_addc_co_u32 %6, vcc, 0, %6, %10\n\t"
            "v_addc_co_u32 %7, vcc, 0, %7, %11\n\t"
            : "+&v"(S[0]), "+&v"(S[1]), "+&v"(S[2]), "+&v"(S[3]), "+&v"(C[0]), "+&v"(C[1]), "+&v"(C[2]), "+&v"(C[3]), "=&s"(cy[0]), "=&s"(cy[1]), "=&s"(cy[2]), "=&s"(cy[3])
            : "v"(q), "s"(m0), "s"(m1), "s"(m2), "s"(m3)
            : "vcc");
    }
};
KZG_DEV void add64_count(uint64_t& slot, uint32_t& cnt, uint32_t add) {
    uint64_t v = slot + add;
    cnt += (v < slot) ? 1u : 0u;
    slot = v;
}

template <class P>
KZG_DEV void f_mul_mad(field_t<P>& r, const field_t<P>& a, const field_t<P>& b) {
    constexpr int N = P::N, H = N / 2;
    static_assert(N % 2 == 0, "even limb count");
    uint64_t E[H], O[H];
    uint32_t CE[H], CO[H];
#pragma unroll
    for (int k = 0; k < H; k++) CE[k] = CO[k] = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
        const uint32_t bi = b.l[i];
        if (i == 0) {
#pragma unroll
            for (int k = 0; k < H; k++) {
                E[k] = (uint64_t)a.l[2 * k] * bi;
                O[k] = (uint64_t)a.l[2 * k + 1] * bi;
            }
        } else {
            mad_row<H>::vec(E, CE, a.l, bi);
            mad_row<H>::vec(O, CO, a.l + 1, bi);
        }
        const uint32_t q = (uint32_t)E[0] * P::INV;
        mad_row<H>::uni(E, CE, q, P::mod(0), P::mod(2), P::mod(4), P::mod(6), P::mod(H > 4 ? 8 : 0),
                        P::mod(H > 4 ? 10 : 0));
        mad_row<H>::uni(O, CO, q, P::mod(1), P::mod(3), P::mod(5), P::mod(7), P::mod(H > 4 ? 9 : 0),
                        P::mod(H > 4 ? 11 : 0));
        // limb 0 is now zero: shift down one limb (E <- O, O <- E shifted), folding the two stragglers
        const uint32_t e0hi = (uint32_t)(E[0] >> 32);
        const uint32_t ce0 = CE[0];
        uint64_t nE[H], nO[H];
        uint32_t nCE[H], nCO[H];
#pragma unroll
        for (int k = 0; k < H; k++) {
            nE[k] = O[k];
            nCE[k] = CO[k];
        }
#pragma unroll
        for (int k = 0; k < H - 1; k++) {
            nO[k] = E[k + 1];
            nCO[k] = CE[k + 1];
        }
        nO[H - 1] = 0;
        nCO[H - 1] = 0;
        add64_count(nE[0], nCE[0], e0hi);
        add64_count(nO[0], nCO[0], ce0);
#pragma unroll
        for (int k = 0; k < H; k++) {
            E[k] = nE[k];
            O[k] = nO[k];
            CE[k] = nCE[k];
            CO[k] = nCO[k];
        }
    }
    // merge the redundant form into N limbs (value < 2m < 2^(32N))
    uint32_t t[N];
    uint32_t c = 0;
    t[0] = (uint32_t)E[0];
#pragma unroll
    for (int i = 1; i < N; i++) {
        uint32_t e = (i & 1) ? (uint32_t)(E[i / 2] >> 32) : (uint32_t)E[i / 2];
        uint32_t o = ((i - 1) & 1) ? (uint32_t)(O[(i - 1) / 2] >> 32) : (uint32_t)O[(i - 1) / 2];
        t[i] = __builtin_addc(e, o, c, &c);
    }
    c = 0;
#pragma unroll
    for (int i = 2; i < N; i++) {
        uint32_t cnt = (i & 1) ? CO[(i - 3) / 2] : CE[(i - 2) / 2];
        t[i] = __builtin_addc(t[i], cnt, c, &c);
    }
    f_final_sub<P>(r.l, t, 0);
}

#ifndef KZG_MUL_PLAIN
#define f_mul f_mul_mad
#else
#define f_mul f_mul_inline
#endif

template <class P>
KZG_DEV void f_sqr(field_t<P>& r, const field_t<P>& a) {
    f_mul(r, a, a);
}
// to / from Montgomery form
template <class P>
KZG_DEV void f_to_mont(field_t<P>& r, const field_t<P>& a) {
    field_t<P> r2;
#pragma unroll
    for (int i = 0; i < P::N; i++) r2.l[i] = P::r2(i);
    f_mul(r, a, r2);
}
template <class P>
KZG_DEV void f_from_mont(field_t<P>& r, const field_t<P>& a) {
    field_t<P> one;
#pragma unroll
    for (int i = 0; i < P::N; i++) one.l[i] = i == 0 ? 1u : 0u;
    f_mul(r, a, one);
}
typedef field_t<FpParams> fp32_t;  // saturated 12 x 32-bit Fp: reference implementation for tests + inversion helper
typedef field_t<FrParams> fr_t;
