// abi_test.hip -- test hooks of the library (include/kzg_mi355x_test.h): single field / group operations on the GPU and
// the host-side point encoder, so that tests/ can compare each of them with the oracle.  Not part of the serving surface.
#include "ctx.hip.h"

using namespace kzg_impl;

namespace {

// ---- unit-op test kernels
// Fr: saturated 32-bit Montgomery (field.hip.h).  Fp: op 0 mul / 1 add / 2 sub / 4 sqr on the 28-bit-limb working
// representation (fp28.hip.h); op 3 = the plain-C++ 12 x 32-bit CIOS reference product.
__global__ void __launch_bounds__(256) k_test_fr(int op, const uint8_t* a_be, const uint8_t* b_be, uint8_t* out_be,
                                                  uint64_t n) {
    uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    if (op == 3) {  // the saturated 8 x 32-bit CIOS reference (field.hip.h), self-contained
        fr_t a, b, r;
        limbs_from_be<8>(a.l, a_be + 32 * j);
        limbs_from_be<8>(b.l, b_be + 32 * j);
        f_to_mont(a, a);
        f_to_mont(b, b);
        f_mul_inline(r, a, b);
        f_from_mont(r, r);
        limbs_to_be<8>(out_be + 32 * j, r.l);
        return;
    }
    uint32_t wa[8], wb[8], wr[8];
    limbs_from_be<8>(wa, a_be + 32 * j);
    limbs_from_be<8>(wb, b_be + 32 * j);
    fr9_t a, b, r;
    fr9_from_words(a, wa);
    fr9_from_words(b, wb);
    fr9_to_mont(a, a);
    fr9_to_mont(b, b);
    if (op == 0) fr9_mul(r, a, b);
    else if (op == 1) fr9_add(r, a, b);
    else if (op == 2) fr9_sub4(r, a, b);
    else if (op == 5 || op == 6) {
        // k (a + b) summed lazily, k = 1 + (a mod 29) <= 29: a value up to 58 r with limbs up to 2^30.9, then the
        // product-free reductions: 5 = fr9_reduce (canonical), 6 = fr9_reduce_approx alone (< 2r)
        const uint32_t k = 1u + (uint32_t)((((uint64_t)wa[1] << 32) | wa[0]) % 29u);   // test side: (a mod 2^64) mod 29
        fr9_t s;
        fr9_add(s, a, b);
        r = s;
        for (uint32_t i = 1; i < k; i++) {
            fr9_add(r, r, s);
            if ((i & 1) == 0) fr9_norm(r, r);        // limbs stay below 2^31
        }
        if (op == 5) fr9_reduce(r, r);
        else {
            fr9_norm(r, r);
            fr9_reduce_approx(r, r);
        }
    }
    else fr9_mul(r, a, a);
    fr9_from_mont(r, r);
    fr9_to_words(wr, r);
    limbs_to_be<8>(out_be + 32 * j, wr);
}
__global__ void __launch_bounds__(256) k_test_fp(int op, const uint8_t* a_be, const uint8_t* b_be, uint8_t* out_be,
                                                  uint64_t n) {
    uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    if (op == 3) {
        fp32_t a, b, r;
        limbs_from_be<12>(a.l, a_be + 48 * j);
        limbs_from_be<12>(b.l, b_be + 48 * j);
        f_to_mont(a, a);
        f_to_mont(b, b);
        f_mul_inline(r, a, b);
        f_from_mont(r, r);
        limbs_to_be<12>(out_be + 48 * j, r.l);
        return;
    }
    fp_t a, b, r;
    fp_from_be48(a, a_be + 48 * j);
    fp_from_be48(b, b_be + 48 * j);
    if (op == 0) fp_mul(r, a, b);
    else if (op == 1) fp_add(r, a, b);
    else if (op == 2) fp_sub4(r, a, b);
    else fp_sqr(r, a);
    fp_to_be48(out_be + 48 * j, r);
}
__global__ void __launch_bounds__(256) k_test_g1(int op, const uint8_t* a_be, const uint8_t* b_be, uint8_t* out_be,
                                                  uint64_t n) {
    uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    g1_aff28 a, b, o;
    fp_from_be48(a.x, a_be + 96 * j); fp_from_be48(a.y, a_be + 96 * j + 48);
    fp_from_be48(b.x, b_be + 96 * j); fp_from_be48(b.y, b_be + 96 * j + 48);
    g1_xyzz_t pa, pb, r, t;
    g1_from_aff(pa, a);
    g1_from_aff(pb, b);
    if (op == 0) { r = pa; g1_madd_checked(r, b); }
    else if (op == 1) { g1_dbl(t, pa); g1_add(r, t, pb); }
    else if (op == 2) { g1_dbl(r, pa); }
    else if (op == 3) { g1_dbl(t, pa); g1_dbl(r, t); }
    else {  // long dependent chain: ((a + b) + b + ... ) exercising the class invariants across many mixed adds
        r = pa;
        for (int k = 0; k < 40; k++) g1_madd_checked<true>(r, (k & 1) ? a : b);  // the inlined-product variant
    }
    g1_to_aff(o, r);
    fp_to_be48(out_be + 96 * j, o.x);
    fp_to_be48(out_be + 96 * j + 48, o.y);
}

// lane-parallel point operations (fp_lp.hip.h), one wave per element: op 5 = 2a + b (full addition of two XYZZ points),
// 6 = 2 * (2a), 7 = ten rounds of r <- 2r + b starting from a (class invariants across a long dependent chain)
__global__ void __launch_bounds__(64) k_test_g1_lp(int op, const uint8_t* a_be, const uint8_t* b_be, uint8_t* out_be) {
    __shared__ LpScratch sm;
    __shared__ g1_xyzz_t pa, pb, r;
    const uint64_t j = blockIdx.x;
    const LpLane k = lp_lane();
    if (threadIdx.x == 0) {
        g1_aff28 a, b;
        fp_from_be48(a.x, a_be + 96 * j); fp_from_be48(a.y, a_be + 96 * j + 48);
        fp_from_be48(b.x, b_be + 96 * j); fp_from_be48(b.y, b_be + 96 * j + 48);
        g1_xyzz_t ta, tb, t2;
        g1_from_aff(ta, a);
        g1_from_aff(tb, b);
        g1_dbl(t2, ta);
        store_xyzz(&pa, op == 7 ? ta : t2);
        store_xyzz(&pb, tb);
    }
    __syncthreads();
    if (op == 5) lp_add(sm, &r, &pa, &pb, k);
    else if (op == 6) lp_dbl(sm, &r, &pa, k);
    else {
        for (int it = 0; it < 10; it++) {
            lp_dbl(sm, &pa, &pa, k);
            lp_add(sm, &pa, &pa, &pb, k);
        }
        if (threadIdx.x < 56) reinterpret_cast<uint32_t*>(&r)[threadIdx.x] = reinterpret_cast<uint32_t*>(&pa)[threadIdx.x];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        g1_xyzz_t v;
        load_xyzz(v, &r);
        g1_aff28 o;
        g1_to_aff(o, v);
        fp_to_be48(out_be + 96 * j, o.x);
        fp_to_be48(out_be + 96 * j + 48, o.y);
    }
}

}  // namespace

extern "C" {

// dev-only prototype hooks (scripts/proto/): compiled in only by `KZG_WITH_PROTO=1 python -m zkp_subnet_amd.build`;
// the shipped library and include/kzg_mi355x.h do not carry them
#ifdef KZG_WITH_PROTO
#include "../../scripts/proto/baff_hook.inc"
#endif

// test hooks for the host-side encoder (finish_host.cpp): no GPU involved
int kzg_host_xyzz_to_c48(const uint32_t xyzz_limbs28[56], uint8_t out48[48]) {
    if (!xyzz_limbs28 || !out48) return KZG_E_ARG;
    kzg_host::xyzz_to_c48(xyzz_limbs28, out48);
    return KZG_OK;
}
int kzg_host_xyzz_pair_to_c48(const uint32_t a_limbs28[56], const uint32_t b_limbs28[56], uint8_t out_a48[48],
                              uint8_t out_b48[48]) {
    if (!a_limbs28 || !b_limbs28 || !out_a48 || !out_b48) return KZG_E_ARG;
    kzg_host::xyzz_pair_to_c48(a_limbs28, b_limbs28, out_a48, out_b48);
    return KZG_OK;
}
int kzg_host_xyzz_to_partial192(const uint32_t xyzz_limbs28[56], uint8_t out192[192]) {
    if (!xyzz_limbs28 || !out192) return KZG_E_ARG;
    kzg_host::xyzz_to_partial192(xyzz_limbs28, out192);
    return KZG_OK;
}

int kzg_test_field(kzg_ctx* ctx, int field, int op, const uint8_t* a_be, const uint8_t* b_be, uint8_t* out_be,
                   uint64_t n) {
    if (!ctx || !a_be || !b_be || !out_be || !n) return KZG_E_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    LaneHold H(ctx);
    if (int rc = H.take()) return rc;
    Lane& L = H.L();
    const size_t w = field == 0 ? 48 : 32;
    HIPCHK(ctx, L.in_be.ensure(2 * n * w));
    HIPCHK(ctx, L.out_be.ensure(n * w));
    uint8_t* da = L.in_be.as<uint8_t>();
    uint8_t* db = da + n * w;
    HIPCHK(ctx, hipMemcpyAsync(da, a_be, n * w, hipMemcpyHostToDevice, L.stream));
    HIPCHK(ctx, hipMemcpyAsync(db, b_be, n * w, hipMemcpyHostToDevice, L.stream));
    uint32_t blocks = (uint32_t)((n + 255) / 256);
    if (field == 0) k_test_fp<<<blocks, 256, 0, L.stream>>>(op, da, db, L.out_be.as<uint8_t>(), n);
    else k_test_fr<<<blocks, 256, 0, L.stream>>>(op, da, db, L.out_be.as<uint8_t>(), n);
    HIPCHK(ctx, hipMemcpyAsync(out_be, L.out_be.p, n * w, hipMemcpyDeviceToHost, L.stream));
    HIPCHK(ctx, hipStreamSynchronize(L.stream));
    HIPCHK(ctx, hipGetLastError());
    H.clean = true;
    return KZG_OK;
}
int kzg_test_g1(kzg_ctx* ctx, int op, const uint8_t* a_be96, const uint8_t* b_be96, uint8_t* out_be96, uint64_t n) {
    if (!ctx || !a_be96 || !b_be96 || !out_be96 || !n) return KZG_E_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    LaneHold H(ctx);
    if (int rc = H.take()) return rc;
    Lane& L = H.L();
    HIPCHK(ctx, L.in_be.ensure(2 * n * 96));
    HIPCHK(ctx, L.out_be.ensure(n * 96));
    uint8_t* da = L.in_be.as<uint8_t>();
    uint8_t* db = da + n * 96;
    HIPCHK(ctx, hipMemcpyAsync(da, a_be96, n * 96, hipMemcpyHostToDevice, L.stream));
    HIPCHK(ctx, hipMemcpyAsync(db, b_be96, n * 96, hipMemcpyHostToDevice, L.stream));
    if (op >= 5) k_test_g1_lp<<<(uint32_t)n, 64, 0, L.stream>>>(op, da, db, L.out_be.as<uint8_t>());
    else k_test_g1<<<(uint32_t)((n + 255) / 256), 256, 0, L.stream>>>(op, da, db, L.out_be.as<uint8_t>(), n);
    HIPCHK(ctx, hipMemcpyAsync(out_be96, L.out_be.p, n * 96, hipMemcpyDeviceToHost, L.stream));
    HIPCHK(ctx, hipStreamSynchronize(L.stream));
    HIPCHK(ctx, hipGetLastError());
    H.clean = true;
    return KZG_OK;
}

}  // extern "C"
