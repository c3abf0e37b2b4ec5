// Result encoder on the host: the ONE point an MSM leaves behind, in the GPU's working form (XYZZ, 4 x 14 lazy 28-bit
// limbs, Montgomery residues with R = 2^392), becomes the 48-byte ZCash-compressed G1 encoding the wire carries
// (reference base/protocol.py:49-60: commitment / proof strings) or the 192-byte partial-sum record of the C-ABI.
//
// Why here and not in a kernel: affine conversion is one field inversion -- a ~100 us single-lane dependent chain
// on the GPU (k_g1_compress: 142 us measured) against a few microseconds of host time -- and the 224 bytes cross PCIe
// in the same copy that used to carry the 48.  It is an encoding step of O(1) work per request, not a fallback: the
// sums themselves never leave the GPU, and nothing here runs when the HIP path fails.  kzg_set_host_finish(ctx, 0)
// keeps the encoding on the GPU (the device-to-device entry points always do).
#include "fp_host.h"

namespace kzg_host {

// lazy 28-bit limbs (each < 2^32) -> the canonical integer of the residue class, 6 x u64
static void limbs28_mod_p(u64 out[6], const uint32_t* l) {
    u64 w[7] = {0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 14; i++) {
        const int bit = 28 * i, wi = bit >> 6, sh = bit & 63;
        u128 v = (u128)l[i] << sh;
        u128 c = (u128)w[wi] + (u64)v;
        w[wi] = (u64)c;
        c = (c >> 64) + (u64)(v >> 64);
        for (int k = wi + 1; k < 7 && c; k++) {
            c += w[k];
            w[k] = (u64)c;
            c >>= 64;
        }
    }
    // value < 2^397 < 2^17 p: conditional subtraction of p << k, k = 16 .. 0
    for (int k = 16; k >= 0; k--) {
        u64 m[7];
        m[0] = PM[0] << k;
        for (int i = 1; i < 6; i++) m[i] = (PM[i] << k) | (k ? PM[i - 1] >> (64 - k) : 0);
        m[6] = k ? PM[5] >> (64 - k) : 0;
        bool ge = true;
        for (int i = 6; i >= 0; i--) {
            if (w[i] != m[i]) {
                ge = w[i] > m[i];
                break;
            }
        }
        if (ge) {
            u64 br = 0;
            for (int i = 0; i < 7; i++) {
                u128 t = (u128)w[i] - m[i] - br;
                w[i] = (u64)t;
                br = (u64)(t >> 64) & 1;
            }
        }
    }
    for (int i = 0; i < 6; i++) out[i] = w[i];
}
static Fp fp_from_limbs28(const uint32_t* l) {
    Fp t;
    limbs28_mod_p(t.l, l);
    return t * FP_R2;
}
static bool limbs_all_zero(const uint32_t* l) {
    uint32_t t = 0;
    for (int i = 0; i < 14; i++) t |= l[i];
    return t == 0;
}

static void affine_to_c48(const Fp& x, const Fp& y, uint8_t out48[48]) {
    u64 yl[6], twice[6];
    fp_to_limbs(yl, y);
    const u64 c = add6(twice, yl, yl);
    const bool larger = c || ge6(twice, PM);  // y > (p - 1) / 2  <=>  2y >= p
    fp_to_be48(out48, x);
    out48[0] |= larger ? 0xA0 : 0x80;
}
static void infinity_c48(uint8_t out48[48]) {
    memset(out48, 0, 48);
    out48[0] = 0xC0;
}
// xyzz: 56 words (X, Y, ZZ, ZZZ); ZZ all-zero limbs = infinity
void xyzz_to_c48(const uint32_t* xyzz, uint8_t out48[48]) {
    if (limbs_all_zero(xyzz + 28)) {
        infinity_c48(out48);
        return;
    }
    // the common factor 2^392 of the four residues cancels in X/ZZ and Y/ZZZ
    const Fp X = fp_from_limbs28(xyzz), Y = fp_from_limbs28(xyzz + 14), ZZ = fp_from_limbs28(xyzz + 28),
             ZZZ = fp_from_limbs28(xyzz + 42);
    const Fp i = inv(ZZ * ZZZ);
    affine_to_c48(X * (i * ZZZ), Y * (i * ZZ), out48);
}
// the two points of a commit+open share ONE inversion (Montgomery's trick)
void xyzz_pair_to_c48(const uint32_t* xyzz0, const uint32_t* xyzz1, uint8_t out0[48], uint8_t out1[48]) {
    const bool inf0 = limbs_all_zero(xyzz0 + 28), inf1 = limbs_all_zero(xyzz1 + 28);
    if (inf0 || inf1) {
        xyzz_to_c48(xyzz0, out0);
        xyzz_to_c48(xyzz1, out1);
        return;
    }
    const Fp X0 = fp_from_limbs28(xyzz0), Y0 = fp_from_limbs28(xyzz0 + 14), ZZ0 = fp_from_limbs28(xyzz0 + 28),
             ZZZ0 = fp_from_limbs28(xyzz0 + 42);
    const Fp X1 = fp_from_limbs28(xyzz1), Y1 = fp_from_limbs28(xyzz1 + 14), ZZ1 = fp_from_limbs28(xyzz1 + 28),
             ZZZ1 = fp_from_limbs28(xyzz1 + 42);
    const Fp d0 = ZZ0 * ZZZ0, d1 = ZZ1 * ZZZ1;
    const Fp i = inv(d0 * d1);
    const Fp i0 = i * d1, i1 = i * d0;   // 1 / d0, 1 / d1
    affine_to_c48(X0 * (i0 * ZZZ0), Y0 * (i0 * ZZ0), out0);
    affine_to_c48(X1 * (i1 * ZZZ1), Y1 * (i1 * ZZ1), out1);
}
// 192-byte partial-sum record: X, Y, ZZ, ZZZ as canonical residues (still Montgomery, R = 2^392), 12 x u32 LE each;
// all zeros = infinity (include/kzg_mi355x.h, kzg_msm_partial)
void xyzz_to_partial192(const uint32_t* xyzz, uint8_t out192[192]) {
    if (limbs_all_zero(xyzz + 28)) {
        memset(out192, 0, 192);
        return;
    }
    for (int k = 0; k < 4; k++) {
        u64 v[6];
        limbs28_mod_p(v, xyzz + 14 * k);
        memcpy(out192 + 48 * k, v, 48);  // little-endian host: 6 x u64 LE == 12 x u32 LE
    }
}

}  // namespace kzg_host
