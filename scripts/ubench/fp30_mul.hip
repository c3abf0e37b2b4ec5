// Product-level A/B for the MSM's base field on gfx950 (VERDICT r2 task 8): the working representation of the round,
//   fp28 : 14 unsigned 28-bit limbs, R = 2^392, lazy additions (limbs < 2^30 are legal product inputs)  -- 2 x 196 mads
// against the candidate that needs fewer multiplies,
//   f30s : 13 SIGNED 30-bit limbs in [-2^29, 2^29], R = 2^390, v_mad_i64_i32 columns                    -- 2 x 169 mads
// Thirteen 30-bit limbs leave a 64-bit column no headroom: 26 products of up to 2^58 are < 2^62.7, so BOTH operands of
// every product must be normalised (an unsigned form does not fit at all: 26 x 2^60 > 2^64, and two accumulators per
// column cost more in carry handling than the 54 mads they save).  Sums and differences therefore pay a signed carry
// propagation (f30_norm) before they may enter a product; fp28 multiplies them as they are.
// Timed: dependent chains of products (two independent chains per lane, 2 waves per SIMD, as k_msm_accumulate runs):
//   mode 0  fp28   a <- a * b                      mode 1  f30s   a <- a * b
//   mode 2  fp28   a <- (a + b) * b   (lazy add)   mode 3  f30s   a <- norm(a + b) * b
// and one product of each kind is written out so that the host script can check f30s against big-integer arithmetic.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/ubench/fp30_mul.hip -o /tmp/fp30_mul && /tmp/fp30_mul
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <vector>

#include "../../zkp_subnet_amd/csrc/fp28.hip.h"

struct f30 {
    int32_t l[13];
};
__host__ __device__ constexpr int32_t f30_p(int i) {
    constexpr int32_t m[13] = {-21845, -402915328, 356515836, -352321620, -252304353, 55215067, 288093811,
                               316751073, -321428361, 517541167, -375082566, -91332614, 1704210};   // p, balanced digits
    return m[i];
}
#define F30_PINV 0x3ffcfffdu   // -p^-1 mod 2^30
KZG_DEV int32_t sext30(uint32_t v) { return (int32_t)(v << 2) >> 2; }
// Montgomery product, product scanning, signed columns; |a_i|, |b_j| <= 2^29 required, output digits in [-2^29, 2^29)
KZG_DEV void f30_mul(f30& r, const f30& a, const f30& b) {
    int32_t q[13];
    int64_t carry = 0;
#pragma unroll
    for (int k = 0; k < 25; k++) {
        int64_t acc0 = carry, acc1 = 0;
#pragma unroll
        for (int i = 0; i < 13; i++) {
            const int j = k - i;
            if (j >= 0 && j < 13) {
                if (i & 1) acc1 += (int64_t)a.l[i] * b.l[j];
                else acc0 += (int64_t)a.l[i] * b.l[j];
            }
        }
#pragma unroll
        for (int i = 0; i < 13; i++) {
            const int j = k - i;
            if (i < k && j >= 1 && j < 13) {
                if (i & 1) acc0 += (int64_t)q[i] * f30_p(j);
                else acc1 += (int64_t)q[i] * f30_p(j);
            }
        }
        int64_t acc = acc0 + acc1;
        if (k < 13) {
            q[k] = sext30((uint32_t)acc * F30_PINV);
            acc += (int64_t)q[k] * f30_p(0);       // the low 30 bits are now zero
            carry = acc >> 30;
        } else {
            const int32_t d = sext30((uint32_t)acc);
            r.l[k - 13] = d;
            carry = (acc - d) >> 30;
        }
    }
    r.l[12] = (int32_t)carry;
}
// signed carry propagation: digits back into [-2^29, 2^29) (the top one keeps the excess)
KZG_DEV void f30_norm(f30& r, const f30& a) {
    int32_t c = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) {
        const int32_t t = a.l[i] + c;
        const int32_t d = sext30((uint32_t)t);
        r.l[i] = d;
        c = (t - d) >> 30;
    }
    r.l[12] = a.l[12] + c;
}
KZG_DEV void f30_add(f30& r, const f30& a, const f30& b) {
#pragma unroll
    for (int i = 0; i < 13; i++) r.l[i] = a.l[i] + b.l[i];
}

#define ITERS 2000
template <int MODE>
__global__ void __launch_bounds__(256) k_chain(const uint32_t* __restrict__ seed, uint32_t* __restrict__ out) {
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    if constexpr (MODE == 0 || MODE == 2) {
        fp_t a0, a1, b;
#pragma unroll
        for (int i = 0; i < 14; i++) {
            a0.l[i] = (seed[i] + t * 7u) & FP28_MASK;
            a1.l[i] = (seed[14 + i] + t * 13u) & FP28_MASK;
            b.l[i] = (seed[28 + i] ^ t) & FP28_MASK;
        }
        a0.l[13] &= 0xffff; a1.l[13] &= 0xffff; b.l[13] &= 0xffff;
        for (int it = 0; it < ITERS; it++) {
            if constexpr (MODE == 2) {
                fp_t s0, s1;
                fp_add(s0, a0, b);
                fp_add(s1, a1, b);
                fp_mul_inline(a0, s0, b);
                fp_mul_inline(a1, s1, b);
            } else {
                fp_mul_inline(a0, a0, b);
                fp_mul_inline(a1, a1, b);
            }
        }
        uint32_t x = 0;
#pragma unroll
        for (int i = 0; i < 14; i++) x ^= a0.l[i] + a1.l[i];
        out[t] = x;
    } else {
        f30 a0, a1, b;
#pragma unroll
        for (int i = 0; i < 13; i++) {
            a0.l[i] = sext30(seed[i] + t * 7u);
            a1.l[i] = sext30(seed[14 + i] + t * 13u);
            b.l[i] = sext30(seed[28 + i] ^ t);
        }
        a0.l[12] >>= 8; a1.l[12] >>= 8; b.l[12] >>= 8;
        for (int it = 0; it < ITERS; it++) {
            if constexpr (MODE == 3) {
                f30 s0, s1;
                f30_add(s0, a0, b);
                f30_add(s1, a1, b);
                f30_norm(s0, s0);
                f30_norm(s1, s1);
                f30_mul(a0, s0, b);
                f30_mul(a1, s1, b);
            } else {
                f30_mul(a0, a0, b);
                f30_mul(a1, a1, b);
            }
        }
        uint32_t x = 0;
#pragma unroll
        for (int i = 0; i < 13; i++) x ^= (uint32_t)(a0.l[i] + a1.l[i]);
        out[t] = x;
    }
}
// one product of each kind on the same integers, written out limb by limb for the host-side check
__global__ void k_check(const int32_t* __restrict__ a30, const int32_t* __restrict__ b30, int32_t* __restrict__ r30,
                        int32_t* __restrict__ rn30) {
    if (threadIdx.x || blockIdx.x) return;
    f30 a, b, r, s;
    for (int i = 0; i < 13; i++) { a.l[i] = a30[i]; b.l[i] = b30[i]; }
    f30_mul(r, a, b);
    for (int i = 0; i < 13; i++) r30[i] = r.l[i];
    f30_add(s, a, b);
    f30_add(s, s, a);          // 2a + b: digits up to 3 x 2^29, then normalised
    f30_norm(s, s);
    f30_mul(r, s, b);
    for (int i = 0; i < 13; i++) rn30[i] = r.l[i];
}

template <int MODE>
static double run(const char* name, const uint32_t* d_seed, uint32_t* d_out) {
    const int blocks = 256 * 4 * 2 / 4;   // 2 waves per SIMD: 2048 waves = 512 workgroups of 4 waves
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k_chain<MODE><<<blocks, 256>>>(d_seed, d_out);
    hipDeviceSynchronize();
    float best = 1e9f;
    for (int rep = 0; rep < 5; rep++) {
        hipEventRecord(e0);
        k_chain<MODE><<<blocks, 256>>>(d_seed, d_out);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    // per SIMD: 2 waves x ITERS x 2 chains products
    const double ns_per_product = best * 1e6 / (2.0 * ITERS * 2);
    printf("{\"mode\": \"%s\", \"ms\": %.3f, \"ns_per_wave_product_per_simd\": %.2f}\n", name, best, ns_per_product);
    return ns_per_product;
}
int main() {
    uint32_t h_seed[42];
    uint64_t x = 0x9E3779B97F4A7C15ull;
    for (auto& v : h_seed) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; v = (uint32_t)x; }
    uint32_t *d_seed, *d_out;
    hipMalloc(&d_seed, sizeof(h_seed));
    hipMalloc(&d_out, 512 * 256 * 4);
    hipMemcpy(d_seed, h_seed, sizeof(h_seed), hipMemcpyHostToDevice);
    const double t0 = run<0>("fp28 a*b", d_seed, d_out);
    const double t1 = run<1>("f30s a*b", d_seed, d_out);
    const double t2 = run<2>("fp28 (a+b)*b lazy", d_seed, d_out);
    const double t3 = run<3>("f30s norm(a+b)*b", d_seed, d_out);
    printf("{\"f30s_over_fp28_plain\": %.4f, \"f30s_over_fp28_with_sum_operand\": %.4f}\n", t1 / t0, t3 / t2);
    // check vectors
    int32_t h_a[13], h_b[13], h_r[13], h_rn[13];
    for (int i = 0; i < 13; i++) {
        x ^= x << 13; x ^= x >> 7; x ^= x << 17;
        h_a[i] = (int32_t)((uint32_t)x << 2) >> 2;
        h_b[i] = (int32_t)((uint32_t)(x >> 32) << 2) >> 2;
    }
    h_a[12] >>= 9; h_b[12] >>= 9;     // values below 2^381
    int32_t *d_a, *d_b, *d_r, *d_rn;
    hipMalloc(&d_a, 52); hipMalloc(&d_b, 52); hipMalloc(&d_r, 52); hipMalloc(&d_rn, 52);
    hipMemcpy(d_a, h_a, 52, hipMemcpyHostToDevice);
    hipMemcpy(d_b, h_b, 52, hipMemcpyHostToDevice);
    k_check<<<1, 64>>>(d_a, d_b, d_r, d_rn);
    hipMemcpy(h_r, d_r, 52, hipMemcpyDeviceToHost);
    hipMemcpy(h_rn, d_rn, 52, hipMemcpyDeviceToHost);
    const char* names[4] = {"a", "b", "r", "rn"};
    const int32_t* vecs[4] = {h_a, h_b, h_r, h_rn};
    for (int v = 0; v < 4; v++) {
        printf("CHECK %s", names[v]);
        for (int i = 0; i < 13; i++) printf(" %d", vecs[v][i]);
        printf("\n");
    }
    return 0;
}
