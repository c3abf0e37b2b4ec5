// PROTOTYPE (measurement only, not on the product path): batched-affine pairwise addition rounds, the alternative to
// the XYZZ bucket accumulation of k_msm_accumulate that VERDICT r1 item 3 asked to be measured instead of estimated.
//
// A round adds npairs independent pairs of affine points (idx[2p], idx[2p+1]) of a table in the 128-byte row format:
//     lambda = (y2 - y1) / (x2 - x1),  x3 = lambda^2 - x1 - x2,  y3 = lambda (x1 - x3) - y1
// with ONE field inversion per lane shared by the lane's K pairs (Montgomery's trick): a forward sweep multiplies the
// denominators and parks the prefix products in HBM (64 B per pair), one inversion (word-approximation binary GCD,
// ~42k instructions per wave), a backward sweep re-reads the points and the prefixes and emits the sums as new table
// rows.  5M + 1S per addition instead of the 8M + 2S (+ fused y3) of the mixed XYZZ addition, at the price of reading
// every point twice and 128 B of prefix traffic per pair.  Round 1 gathers its operands through the sorted entry list of
// a real MSM (pairs of consecutive entries); later rounds read the previous round's output in order.
// Pairs with equal x (doubling / cancellation) are only counted: the prototype measures cost, it does not replace the
// accumulate kernel.  Checked against the XYZZ formulas by k_baff_check.
#include "../../zkp_subnet_amd/csrc/msm.hip.h"

KZG_DEV void baff_load_x(fp_t& x, const g1_affine_t* row) {
    const uint4* q = reinterpret_cast<const uint4*>(row);
    uint32_t w[16];
#pragma unroll
    for (int i = 0; i < 4; i++) { uint4 v = q[i]; w[4*i]=v.x; w[4*i+1]=v.y; w[4*i+2]=v.z; w[4*i+3]=v.w; }
#pragma unroll
    for (int i = 0; i < 14; i++) x.l[i] = w[i];
}
KZG_DEV void baff_park(uint4* slot, const fp_t& v) {   // 64-byte slot: 14 limbs + padding
    slot[0] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
    slot[1] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
    slot[2] = make_uint4(v.l[8], v.l[9], v.l[10], v.l[11]);
    slot[3] = make_uint4(v.l[12], v.l[13], 0u, 0u);
}
KZG_DEV void baff_unpark(fp_t& v, const uint4* slot) {
    uint4 a = slot[0], b = slot[1], c = slot[2], d = slot[3];
    v.l[0]=a.x; v.l[1]=a.y; v.l[2]=a.z; v.l[3]=a.w; v.l[4]=b.x; v.l[5]=b.y; v.l[6]=b.z; v.l[7]=b.w;
    v.l[8]=c.x; v.l[9]=c.y; v.l[10]=c.z; v.l[11]=c.w; v.l[12]=d.x; v.l[13]=d.y;
}
KZG_DEV bool baff_same(const fp_t& a, const fp_t& b) {
    uint32_t t = 0;
#pragma unroll
    for (int i = 0; i < 14; i++) t |= a.l[i] ^ b.l[i];
    return t == 0;
}
// idx == nullptr: pair p = rows (2p, 2p + 1) of `table`
__global__ void __launch_bounds__(256, 2) k_baff_round(const g1_affine_t* __restrict__ table,
                                                        const uint32_t* __restrict__ idx, uint32_t npairs,
                                                        uint32_t lanes, uint4* __restrict__ prefix,
                                                        g1_affine_t* __restrict__ out, uint32_t* __restrict__ n_equal_x) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= lanes || t >= npairs) return;
    const uint32_t K = (npairs - t + lanes - 1) / lanes;   // pairs t, t + lanes, t + 2 lanes, ...
    fp_t acc, one;
    fp_one(one);
    acc = one;
    uint32_t skipped = 0;
    for (uint32_t k = 0; k < K; k++) {
        const uint32_t p = k * lanes + t;
        const uint32_t i1 = idx ? idx[2 * p] & 0x7fffffffu : 2 * p, i2 = idx ? idx[2 * p + 1] & 0x7fffffffu : 2 * p + 1;
        fp_t x1, x2, d;
        baff_load_x(x1, table + i1);
        baff_load_x(x2, table + i2);
        baff_park(prefix + 4 * (uint64_t)p, acc);
        const bool same = baff_same(x1, x2);
        fp_sub4(d, x2, x1);
        fp_select(d, d, one, same);
        skipped += same;
        fp_mul_inline(acc, acc, d);
    }
    if (skipped) atomicAdd(n_equal_x, skipped);
    fp_t inv;
    fp_inv(inv, acc);
    for (uint32_t k = K; k-- > 0;) {
        const uint32_t p = k * lanes + t;
        const uint32_t e1 = idx ? idx[2 * p] : 2 * p, e2 = idx ? idx[2 * p + 1] : 2 * p + 1;
        g1_aff28 a, b;
        g1_load_aff(a, table + (e1 & 0x7fffffffu));
        g1_load_aff(b, table + (e2 & 0x7fffffffu));
        if (idx) {
            g1_neg_aff(a, e1 >> 31);
            g1_neg_aff(b, e2 >> 31);
        }
        fp_t pre, d, inv_d, lam, t1, t2, x3, y3;
        baff_unpark(pre, prefix + 4 * (uint64_t)p);
        const bool same = baff_same(a.x, b.x);
        fp_sub4(d, b.x, a.x);
        fp_select(d, d, one, same);
        fp_mul_inline(inv_d, inv, pre);          // 1 / d_k
        fp_mul_inline(inv, inv, d);              // 1 / (d_0 ... d_{k-1})
        fp_sub4(t1, b.y, a.y);
        fp_mul_inline(lam, t1, inv_d);
        fp_sqr_inline(t2, lam);
        fp_sub4(t2, t2, a.x);
        fp_sub4(t2, t2, b.x);                    // < 2p + 8p
        fp_norm(x3, t2);
        fp_sub16(t1, a.x, x3);
        fp_mul_inline(t1, lam, t1);
        fp_sub4(t1, t1, a.y);
        fp_norm(y3, t1);                         // < 6p
        g1_aff28 o;
        fp_canon_mont(o.x, x3);                  // table rows hold canonical residues (one more product each; a
        fp_canon_mont(o.y, y3);                  // production version would keep x3, y3 lazily reduced instead)
        g1_store_aff(out + p, o);
    }
}
// out[p] == table[idx[2p]] + table[idx[2p+1]] for pairs p = first + stride * j, via the XYZZ formulas
__global__ void __launch_bounds__(64) k_baff_check(const g1_affine_t* __restrict__ table, const uint32_t* __restrict__ idx,
                                                    uint32_t npairs, uint32_t stride, const g1_affine_t* __restrict__ out,
                                                    uint32_t* __restrict__ bad) {
    const uint32_t p = (blockIdx.x * blockDim.x + threadIdx.x) * stride;
    if (p >= npairs) return;
    const uint32_t e1 = idx ? idx[2 * p] : 2 * p, e2 = idx ? idx[2 * p + 1] : 2 * p + 1;
    g1_aff28 a, b, want, got;
    g1_load_aff(a, table + (e1 & 0x7fffffffu));
    g1_load_aff(b, table + (e2 & 0x7fffffffu));
    if (idx) {
        g1_neg_aff(a, e1 >> 31);
        g1_neg_aff(b, e2 >> 31);
    }
    if (baff_same(a.x, b.x)) return;
    g1_xyzz_t acc;
    g1_from_aff(acc, a);
    g1_madd_checked(acc, b);
    g1_to_aff(want, acc);
    g1_load_aff(got, out + p);
    if (!baff_same(want.x, got.x) || !baff_same(want.y, got.y)) atomicAdd(bad, 1u);
}

void launch_baff_round(hipStream_t s, const g1_affine_t* table, const uint32_t* idx, uint32_t npairs, uint32_t lanes,
                       void* prefix, g1_affine_t* out, uint32_t* n_equal_x) {
    if (!npairs) return;
    k_baff_round<<<(lanes + 255) / 256, 256, 0, s>>>(table, idx, npairs, lanes, reinterpret_cast<uint4*>(prefix), out,
                                                     n_equal_x);
}
void launch_baff_check(hipStream_t s, const g1_affine_t* table, const uint32_t* idx, uint32_t npairs, uint32_t stride,
                       const g1_affine_t* out, uint32_t* bad) {
    const uint32_t cnt = (npairs + stride - 1) / stride;
    if (cnt) k_baff_check<<<(cnt + 63) / 64, 64, 0, s>>>(table, idx, npairs, stride, out, bad);
}
