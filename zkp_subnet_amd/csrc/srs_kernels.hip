// srs_kernels.hip -- SRS plumbing on the GPU: setup points from 96-byte / 48-byte compressed records into the table format
// (reference base/miner.py:75-84 loads the setup file in the prover), read-back, the window tables 2^off[w] P_j by doubling +
// batched normalisation, and the synthetic tau-derived SRS of tests and benches (`fourier setup --generate-setup`,
// reference tests/conftest.py:50-65).
#include "msm_dev.hip.h"

// ------------------------------------------------------------------------------------------------ SRS plumbing
__global__ void __launch_bounds__(256) k_srs_from_be96(const uint8_t* __restrict__ be, g1_affine_t* __restrict__ out,
                                                        uint64_t n, uint32_t* __restrict__ bad) {
    uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const uint32_t* w = reinterpret_cast<const uint32_t*>(be + 96 * j);
    uint32_t any = 0;
#pragma unroll
    for (int i = 0; i < 24; i++) any |= w[i];
    g1_aff28 p;
    if (!any) {
        fp_zero(p.x); fp_zero(p.y);
        g1_store_aff(&out[j], p);
        return;
    }
    bool ok = fp_from_be48(p.x, be + 96 * j);
    ok &= fp_from_be48(p.y, be + 96 * j + 48);
    if (!ok) atomicOr(bad, 1u);
    // on-curve: y^2 == x^3 + 4
    fp_t y2, x3, four, t;
    fp_sqr(y2, p.y);
    fp_sqr(x3, p.x); fp_mul(x3, x3, p.x);
    fp_one(four); fp_dbl(four, four); fp_dbl(four, four);
    fp_add(t, x3, four);
    fp_sub4(t, t, y2);                 // == 0 mod p on the curve
    fp_t one, chk;
    fp_one(one);
    fp_mul(chk, t, one);
    if (!fp_is_zero_n(chk)) atomicOr(bad, 2u);
    g1_store_aff(&out[j], p);
}
// ---- ZCash-compressed SRS files (the reference's `uncompressed=False` setup files: base/miner.py:75-81,
// utils/config.py:131-150): 48 bytes per point, flags compressed 0x80 | infinity 0x40 | y-sign 0x20, x big-endian.
// y = (x^3 + 4)^((p+1)/4) (p = 3 mod 4); a non-residue, x >= p or malformed flags fail the load.
KZG_DEV void fp_pow_p_plus_1_over_4(fp_t& r, const fp_t& a) {
    fp_t acc;
    fp_one(acc);
    for (int i = 378; i >= 0; i--) {  // (p+1)/4 has 379 bits; bit i of it is bit i+2 of p+1
        fp_sqr(acc, acc);
        const int b = i + 2;
        uint32_t w = 0;
#pragma unroll
        for (int k = 0; k < 12; k++) w = (k == (b >> 5)) ? (k == 0 ? FpParams::mod(0) + 1u : FpParams::mod(k)) : w;
        if ((w >> (b & 31)) & 1u) fp_mul(acc, acc, a);
    }
    r = acc;
}
KZG_DEV bool fp_mont_is_larger(const fp_t& y_mont) {  // y > (p-1)/2 for the canonical integer behind y_mont
    fp_t yc;
    fp_from_mont(yc, y_mont);
    uint32_t y[12], t[12], pm[12];
    fp_pack(y, yc);
    const uint32_t c = bi_add<12>(t, y, y);
#pragma unroll
    for (int i = 0; i < 12; i++) pm[i] = FpParams::mod(i);
    return c || bi_ge<12>(t, pm);
}
__global__ void __launch_bounds__(256) k_srs_from_c48(const uint8_t* __restrict__ c48, g1_affine_t* __restrict__ out,
                                                       uint64_t n, uint32_t* __restrict__ bad) {
    uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    uint32_t w[12], pm[12];
    limbs_from_be<12>(w, c48 + 48 * j);  // w[11] holds the flag bits
    const uint32_t flags = w[11] >> 29;
    w[11] &= 0x1fffffffu;
    g1_aff28 p;
    fp_zero(p.x); fp_zero(p.y);
    if (!(flags & 4u)) atomicOr(bad, 1u);  // not a compressed encoding
    if (flags & 2u) {                      // infinity: every other bit must be clear
        uint32_t any = flags & 1u;
#pragma unroll
        for (int i = 0; i < 12; i++) any |= w[i];
        if (any) atomicOr(bad, 1u);
        g1_store_aff(&out[j], p);
        return;
    }
#pragma unroll
    for (int i = 0; i < 12; i++) pm[i] = FpParams::mod(i);
    if (bi_ge<12>(w, pm)) atomicOr(bad, 1u);
    fp_t xr, x3, four, y2, y, t, one, chk;
    fp_unpack(xr, w);
    fp_to_mont(p.x, xr);
    fp_canon(p.x, p.x);
    fp_sqr(x3, p.x); fp_mul(x3, x3, p.x);
    fp_one(one);
    fp_dbl(four, one); fp_dbl(four, four);
    fp_add(t, x3, four);
    fp_mul(y2, t, one);                    // normalised x^3 + 4
    fp_pow_p_plus_1_over_4(y, y2);
    fp_sqr(t, y);
    fp_sub4(t, t, y2);
    fp_mul(chk, t, one);
    if (!fp_is_zero_n(chk)) atomicOr(bad, 2u);  // x^3 + 4 is not a square: no such point
    fp_canon(p.y, y);
    if (fp_mont_is_larger(p.y) != ((flags & 1u) != 0)) fp_neg_canon(p.y, p.y);
    g1_store_aff(&out[j], p);
}
__global__ void __launch_bounds__(256) k_srs_to_c48(const g1_affine_t* __restrict__ in, uint8_t* __restrict__ c48,
                                                     uint64_t n) {
    uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    g1_aff28 p;
    g1_load_aff(p, &in[j]);
    g1_compress(c48 + 48 * j, p);
}
__global__ void __launch_bounds__(256) k_srs_to_be96(const g1_affine_t* __restrict__ in, uint8_t* __restrict__ be,
                                                      uint64_t n) {
    uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    g1_aff28 p;
    g1_load_aff(p, in + j);
    fp_to_be48(be + 96 * j, p.x);
    fp_to_be48(be + 96 * j + 48, p.y);
}

// window tables: tmp[(w-1)*count + j] = 2^off[w] P_{first+j} in XYZZ
__global__ void __launch_bounds__(256) k_precomp_dbl(const g1_affine_t* __restrict__ table, uint64_t first,
                                                      uint64_t count, const WinLayout lay,
                                                      g1_xyzz_t* __restrict__ tmp) {
    uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= count) return;
    g1_aff28 p;
    g1_load_aff(p, table + first + j);
    g1_xyzz_t cur, r;
    g1_from_aff(cur, p);
    for (int w = 1; w < lay.nwin; w++) {
        for (int k = lay.off[w - 1]; k < lay.off[w]; k++) {
            g1_dbl(r, cur);
            cur = r;
        }
        store_xyzz(&tmp[(uint64_t)(w - 1) * count + j], cur);
    }
}
// d = ZZ*ZZZ products chained per lane (Montgomery batch inversion); prefix products are parked in the
// destination slots (as packed canonical residues)
KZG_DEV void park_fp(g1_affine_t* slot, const fp_t& v_loose) {
    fp_t c;
    fp_canon_mont(c, v_loose);
#pragma unroll
    for (int i = 0; i < 14; i++) slot->x[i] = c.l[i];
}
KZG_DEV void unpark_fp(fp_t& v, const g1_affine_t* slot) {
#pragma unroll
    for (int i = 0; i < 14; i++) v.l[i] = slot->x[i];
}
KZG_DEV void xyzz_to_aff_with_inv(g1_aff28& o, const g1_xyzz_t& p, const fp_t& iw /* 1/(zz*zzz) */) {
    fp_t t;
    fp_mul(t, iw, p.zzz);   // 1/zz
    fp_mul(t, p.x, t);
    fp_canon(o.x, t);
    fp_mul(t, iw, p.zz);    // 1/zzz
    fp_mul(t, p.y, t);
    fp_canon(o.y, t);
}
__global__ void __launch_bounds__(256) k_precomp_norm(g1_affine_t* __restrict__ table, uint64_t stride,
                                                       uint64_t first, uint64_t count, int nwin,
                                                       const g1_xyzz_t* __restrict__ tmp) {
    uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= count) return;
    fp_t acc, d;
    fp_one(acc);
    for (int w = 1; w < nwin; w++) {
        g1_xyzz_t p;
        load_xyzz(p, &tmp[(uint64_t)(w - 1) * count + j]);
        park_fp(&table[(uint64_t)w * stride + first + j], acc);
        if (!g1_is_inf(p)) {
            fp_mul(d, p.zz, p.zzz);
            fp_mul(acc, acc, d);
        }
    }
    fp_t inv;
    fp_inv_fermat(inv, acc);
    for (int w = nwin - 1; w >= 1; w--) {
        g1_xyzz_t p;
        load_xyzz(p, &tmp[(uint64_t)(w - 1) * count + j]);
        g1_affine_t* dst = &table[(uint64_t)w * stride + first + j];
        g1_aff28 o;
        if (g1_is_inf(p)) {
            fp_zero(o.x); fp_zero(o.y);
        } else {
            fp_t pre, iw;
            unpark_fp(pre, dst);
            fp_mul(d, p.zz, p.zzz);
            fp_mul(iw, inv, pre);   // 1 / (zz*zzz)
            fp_mul(inv, inv, d);
            xyzz_to_aff_with_inv(o, p, iw);
        }
        g1_store_aff(dst, o);
    }
}

// ---- synthetic SRS (tests / benches): out[j] = [s0 tau^j] G via an 8-bit fixed-base table of G
KZG_DEV void g1_generator(g1_aff28& g) {
    constexpr uint32_t gx[12] = {0xdb22c6bbu, 0xfb3af00au, 0xf97a1aefu, 0x6c55e83fu, 0x171bac58u, 0xa14e3a3fu,
                                 0x9774b905u, 0xc3688c4fu, 0x4fa9ac0fu, 0x2695638cu, 0x3197d794u, 0x17f1d3a7u};
    constexpr uint32_t gy[12] = {0x46c5e7e1u, 0x0caa2329u, 0xa2888ae4u, 0xd03cc744u, 0x2c04b3edu, 0x00db18cbu,
                                 0xd5d00af6u, 0xfcf5e095u, 0x741d8ae4u, 0xa09e30edu, 0xe3aaa0f1u, 0x08b3f481u};
    uint32_t wx[12], wy[12];
#pragma unroll
    for (int i = 0; i < 12; i++) { wx[i] = gx[i]; wy[i] = gy[i]; }
    fp_t x, y;
    fp_unpack(x, wx);
    fp_unpack(y, wy);
    fp_to_mont(g.x, x); fp_canon(g.x, g.x);
    fp_to_mont(g.y, y); fp_canon(g.y, g.y);
}
// gtab[w*255 + d] = (d+1) * 2^(8w) * G, affine
__global__ void __launch_bounds__(256) k_gen_gtab(g1_affine_t* __restrict__ gtab) {
    uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= 32 * 255) return;
    uint32_t w = gid / 255, d = gid % 255 + 1;
    g1_aff28 g;
    g1_generator(g);
    g1_xyzz_t base, r, acc;
    g1_from_aff(base, g);
    for (uint32_t k = 0; k < 8 * w; k++) { g1_dbl(r, base); base = r; }
    g1_set_inf(acc);
    for (int b = 7; b >= 0; b--) {
        g1_dbl(r, acc); acc = r;
        if ((d >> b) & 1u) { g1_add(r, acc, base); acc = r; }
    }
    g1_aff28 o;
    fp_t t, iw;
    fp_mul(t, acc.zz, acc.zzz);
    fp_inv_fermat(iw, t);
    xyzz_to_aff_with_inv(o, acc, iw);
    g1_store_aff(&gtab[gid], o);
}
// scal[j] = s0 * tau^j (canonical limbs); 64 consecutive j per lane
__global__ void __launch_bounds__(256) k_srs_scalars(uint32_t* __restrict__ scal, uint64_t count, uint64_t j_base,
                                                      const uint32_t* __restrict__ tau_mont,
                                                      const uint32_t* __restrict__ s0_mont) {
    uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t j0 = t * 64;
    if (j0 >= count) return;
    fr9_t tau, cur, pw;
    fr9_load(tau, tau_mont);
    fr9_load(cur, s0_mont);
    pw = tau;  // cur *= tau^(j_base + j0)
    for (uint64_t e = j_base + j0; e; e >>= 1) {
        if (e & 1) fr9_mul(cur, cur, pw);
        fr9_mul(pw, pw, pw);
    }
    for (uint64_t j = j0; j < j0 + 64 && j < count; j++) {
        fr9_t c;
        fr9_from_mont(c, cur);
        fr9_store(scal + 8 * j, c);
        fr9_mul(cur, cur, tau);
    }
}
__global__ void __launch_bounds__(256) k_srs_fixed_mul(const uint32_t* __restrict__ scal, uint64_t count,
                                                        const g1_affine_t* __restrict__ gtab,
                                                        g1_xyzz_t* __restrict__ tmp) {
    uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= count) return;
    uint32_t s[8];
    load_scalar(s, scal, j, 0);
    g1_xyzz_t acc;
    g1_set_inf(acc);
    for (int w = 0; w < 32; w++) {
        uint32_t d = (limb_at(s, w >> 2) >> (8 * (w & 3))) & 0xffu;
        if (d) {
            g1_aff28 p;
            g1_load_aff(p, gtab + w * 255 + d - 1);
            g1_madd<true>(acc, p.x, p.y);
        }
    }
    store_xyzz(&tmp[j], acc);
}
// XYZZ -> affine for `count` points, 16 consecutive points per lane share one inversion
__global__ void __launch_bounds__(256) k_batch_affine(const g1_xyzz_t* __restrict__ tmp, g1_affine_t* __restrict__ out,
                                                       uint64_t count) {
    uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t j0 = t * 16;
    if (j0 >= count) return;
    uint64_t j1 = min(j0 + 16, count);
    fp_t acc, d, inv;
    fp_one(acc);
    for (uint64_t j = j0; j < j1; j++) {
        g1_xyzz_t p;
        load_xyzz(p, &tmp[j]);
        park_fp(&out[j], acc);
        if (!g1_is_inf(p)) {
            fp_mul(d, p.zz, p.zzz);
            fp_mul(acc, acc, d);
        }
    }
    fp_inv_fermat(inv, acc);
    for (uint64_t j = j1; j-- > j0;) {
        g1_xyzz_t p;
        load_xyzz(p, &tmp[j]);
        g1_aff28 o;
        if (g1_is_inf(p)) {
            fp_zero(o.x); fp_zero(o.y);
        } else {
            fp_t pre, iw;
            unpark_fp(pre, &out[j]);
            fp_mul(d, p.zz, p.zzz);
            fp_mul(iw, inv, pre);
            fp_mul(inv, inv, d);
            xyzz_to_aff_with_inv(o, p, iw);
        }
        g1_store_aff(&out[j], o);
    }
}


// ------------------------------------------------------------------------------------------------ launchers
void launch_srs_from_be96(hipStream_t s, const uint8_t* be96, g1_affine_t* out, uint64_t n, uint32_t* bad_flag) {
    if (!n) return;
    k_srs_from_be96<<<nblk(n, 256), 256, 0, s>>>(be96, out, n, bad_flag);
}
void launch_srs_from_c48(hipStream_t s, const uint8_t* c48, g1_affine_t* out, uint64_t n, uint32_t* bad_flag) {
    if (!n) return;
    k_srs_from_c48<<<nblk(n, 256), 256, 0, s>>>(c48, out, n, bad_flag);
}
void launch_srs_to_c48(hipStream_t s, const g1_affine_t* in, uint8_t* c48, uint64_t n) {
    if (!n) return;
    k_srs_to_c48<<<nblk(n, 256), 256, 0, s>>>(in, c48, n);
}
void launch_srs_to_be96(hipStream_t s, const g1_affine_t* in, uint8_t* be96, uint64_t n) {
    if (!n) return;
    k_srs_to_be96<<<nblk(n, 256), 256, 0, s>>>(in, be96, n);
}
void launch_srs_precompute(hipStream_t s, g1_affine_t* table, uint64_t stride, uint64_t first, uint64_t count,
                           const WinLayout& lay, g1_xyzz_t* tmp) {
    if (!count || lay.nwin < 2) return;
    k_precomp_dbl<<<nblk(count, 256), 256, 0, s>>>(table, first, count, lay, tmp);
    k_precomp_norm<<<nblk(count, 256), 256, 0, s>>>(table, stride, first, count, lay.nwin, tmp);
}
void launch_srs_generate(hipStream_t s, g1_affine_t* out, uint64_t count, uint64_t j_base, const uint32_t* tau_mont,
                         const uint32_t* s0_mont, g1_affine_t* gtab, g1_xyzz_t* tmp, bool build_gtab) {
    if (!count) return;
    if (build_gtab) k_gen_gtab<<<nblk(32 * 255, 256), 256, 0, s>>>(gtab);
    uint32_t* sbuf = reinterpret_cast<uint32_t*>(tmp + count);  // scalars staged behind the XYZZ scratch
    k_srs_scalars<<<nblk((count + 63) / 64, 256), 256, 0, s>>>(sbuf, count, j_base, tau_mont, s0_mont);
    k_srs_fixed_mul<<<nblk(count, 256), 256, 0, s>>>(sbuf, count, gtab, tmp);
    k_batch_affine<<<nblk((count + 15) / 16, 256), 256, 0, s>>>(tmp, out, count);
}

