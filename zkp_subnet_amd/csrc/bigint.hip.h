// Multi-limb integer helpers for gfx950 (CDNA4): 32-bit limbs, little-endian limb order.
// v_mad_u64_u32 (32x32+64) is the widest integer multiply the VALU offers, so 381-bit Fp is 12 limbs
// and 255-bit Fr is 8 limbs.  Everything is fully unrolled: limbs live in VGPRs, never in scratch.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define KZG_DEV __device__ __forceinline__

template <int N>
struct alignas(16) bigint_t {
    uint32_t l[N];
};

template <int N>
KZG_DEV uint32_t bi_add(uint32_t* r, const uint32_t* a, const uint32_t* b) {
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < N; i++) r[i] = __builtin_addc(a[i], b[i], c, &c);
    return c;
}
template <int N>
KZG_DEV uint32_t bi_sub(uint32_t* r, const uint32_t* a, const uint32_t* b) {
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < N; i++) r[i] = __builtin_subc(a[i], b[i], c, &c);
    return c;
}
template <int N>
KZG_DEV bool bi_is_zero(const uint32_t* a) {
    uint32_t t = 0;
#pragma unroll
    for (int i = 0; i < N; i++) t |= a[i];
    return t == 0;
}
template <int N>
KZG_DEV bool bi_eq(const uint32_t* a, const uint32_t* b) {
    uint32_t t = 0;
#pragma unroll
    for (int i = 0; i < N; i++) t |= a[i] ^ b[i];
    return t == 0;
}
// a >= b
template <int N>
KZG_DEV bool bi_ge(const uint32_t* a, const uint32_t* b) {
    uint32_t t[N];
    return bi_sub<N>(t, a, b) == 0;
}
template <int N>
KZG_DEV void bi_select(uint32_t* r, const uint32_t* a, const uint32_t* b, bool take_b) {
#pragma unroll
    for (int i = 0; i < N; i++) r[i] = take_b ? b[i] : a[i];
}
