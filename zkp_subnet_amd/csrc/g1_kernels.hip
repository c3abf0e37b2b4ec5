// g1_kernels.hip -- small G1 kernels around the MSM: sums of a few points (partials of the SRS segments, the worker rows'
// commitments: reference neurons/validator.py:196-198), membership in the prime-order subgroup, affine conversion + ZCash
// compression on the GPU, the record publish into the lane's pinned page, the 192-byte partial format.
#include "msm_dev.hip.h"

__global__ void __launch_bounds__(64) k_g1_sum(const g1_xyzz_t* __restrict__ in, uint32_t count,
                                                g1_xyzz_t* __restrict__ out) {
    tail_priority();
    __shared__ g1_xyzz_t sm[64];
    const uint32_t tid = threadIdx.x;
    g1_xyzz_t acc, q, r;
    g1_set_inf(acc);
    for (uint32_t i = tid; i < count; i += 64) {
        load_xyzz(q, &in[i]);
        g1_add(r, acc, q);
        acc = r;
    }
    uint32_t lanes = 1;                      // tree only over the lanes that can hold a term
    while (lanes < count && lanes < 64) lanes <<= 1;
    lds_tree_sum(sm, acc, tid, lanes);
    if (tid == 0) store_xyzz(out, acc);
}

// sum of up to 32 XYZZ points by lane-parallel additions (one wave per addition, log2 depth of ~2 us steps): the sum of
// the all_gathered partials of an SRS-sharded MSM is on every step's critical path (8 ranks: 3 levels instead of the
// 4 x 16 us of dependent one-lane additions of k_g1_sum)
__global__ void __launch_bounds__(512) k_g1_sum_lp(const g1_xyzz_t* __restrict__ in, uint32_t count,
                                                    g1_xyzz_t* __restrict__ out) {
    tail_priority();
    __shared__ LpScratch sm[8];
    __shared__ g1_xyzz_t pts[32];
    const LpLane k = lp_lane();
    const int w = (int)__builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (uint32_t i = threadIdx.x; i < 32 * 56; i += 512)
        reinterpret_cast<uint32_t*>(pts)[i] = i < count * 56 ? reinterpret_cast<const uint32_t*>(in)[i] : 0u;   // zeros = infinity
    __syncthreads();
    for (int d = 16; d >= 1; d >>= 1) {
        if ((uint32_t)d < count)
            for (int l = w; l < d; l += 8)
                if ((uint32_t)(l + d) < count) lp_add(sm[w], &pts[l], &pts[l], &pts[l + d], k);
        __syncthreads();
    }
    if (threadIdx.x < 56) reinterpret_cast<uint32_t*>(out)[threadIdx.x] = reinterpret_cast<const uint32_t*>(&pts[0])[threadIdx.x];
}
// sum of `count` affine points (table-row format: the output of k_srs_from_c48): Pianist's master aggregation
// sum_i commit_i over the worker rows (reference neurons/validator.py:196-198, README.md:38)
__global__ void __launch_bounds__(64) k_g1_sum_affine(const g1_affine_t* __restrict__ in, uint32_t count,
                                                       g1_xyzz_t* __restrict__ out) {
    tail_priority();
    __shared__ g1_xyzz_t sm[64];
    const uint32_t tid = threadIdx.x;
    g1_xyzz_t acc;
    g1_set_inf(acc);
    for (uint32_t i = tid; i < count; i += 64) {
        g1_aff28 q;
        g1_load_aff(q, &in[i]);
        g1_madd_checked(acc, q);
    }
    uint32_t lanes = 1;
    while (lanes < count && lanes < 64) lanes <<= 1;
    lds_tree_sum(sm, acc, tid, lanes);
    if (tid == 0) store_xyzz(out, acc);
}

// ---- G1 membership of untrusted points (the miners' commitments entering the master aggregation): on the curve is not
// enough -- E(Fp) has cofactor h = (z-1)^2/3 ~ 2^126.  With sigma(x, y) = (beta x, y), beta a primitive cube root of unity,
// the endomorphism sigma + z^2 (z = |BLS parameter| = 0xd201000000010000) has degree N(sigma + z^2) = z^4 - z^2 + 1 = r,
// is separable, and kills G1 (sigma acts there as -z^2 mod r): its kernel IS G1.  So P is in G1 exactly when
// [z^2] P == -sigma(P), two multiplications by the 64-bit, weight-6 z: 126 doublings + 10 additions instead of the
// 255 + ~127 of [r]P.  One wave per point (lane-parallel point operations, ~1.5 us each): ~0.2 ms for any count up to
// the number of SIMDs.  beta below satisfies sigma(G) = -[z^2]G for the generator (checked in tests/test_oracle.py).
FP28_TABLE(g1_beta_mont, 0x0a75929au, 0x0681b798u, 0x022a3e9du, 0x0abc02bfu, 0x04e5bb45u, 0x055e6e7eu, 0x04814117u,
           0x06d04f1bu, 0x0ae3387du, 0x054acb0cu, 0x00a4c74bu, 0x056138b5u, 0x0b64e066u, 0x000076f2u)
#define BLS_Z 0xd201000000010000ull
__global__ void __launch_bounds__(64) k_g1_subgroup_check_lp(const g1_affine_t* __restrict__ in, uint32_t count,
                                                              uint32_t* __restrict__ bad) {
    tail_priority();
    __shared__ LpScratch sm;
    __shared__ g1_xyzz_t base, acc;
    const LpLane k = lp_lane();
    const uint32_t j = blockIdx.x;
    if (j >= count) return;
    if (threadIdx.x == 0) {
        g1_aff28 a;
        g1_load_aff(a, &in[j]);
        g1_xyzz_t t;
        g1_from_aff(t, a);
        store_xyzz(&base, t);
        store_xyzz(&acc, t);
    }
    __syncthreads();
    for (int pass = 0; pass < 2; pass++) {           // acc <- [z] base, most significant bit first; then base <- acc
        for (int b = 62; b >= 0; b--) {
            lp_dbl(sm, &acc, &acc, k);
            __syncthreads();
            if ((BLS_Z >> b) & 1ull) {
                lp_add(sm, &acc, &acc, &base, k);
                __syncthreads();
            }
        }
        if (threadIdx.x < 56) reinterpret_cast<uint32_t*>(&base)[threadIdx.x] = reinterpret_cast<const uint32_t*>(&acc)[threadIdx.x];
        __syncthreads();
    }
    if (threadIdx.x != 0) return;
    g1_aff28 a;
    g1_load_aff(a, &in[j]);
    if (g1_aff_is_inf(a)) return;                    // the identity is a member
    g1_xyzz_t q;
    load_xyzz(q, &acc);
    if (g1_is_inf(q)) { atomicOr(bad, 4u); return; } // order divides z^2: not in G1
    // [z^2]P = (X/ZZ, Y/ZZZ) == (beta x, -y)  <=>  beta x ZZ - X == 0  and  y ZZZ + Y == 0
    fp_t beta, one, t, d, chk;
#pragma unroll
    for (int i = 0; i < 14; i++) beta.l[i] = g1_beta_mont(i);
    fp_one(one);
    fp_mul(t, beta, a.x);
    fp_mul(t, t, q.zz);
    fp_sub16(d, t, q.x);                             // X: normalised, < 14p
    fp_mul(chk, d, one);
    bool ok = fp_is_zero_n(chk);
    fp_mul(t, a.y, q.zzz);
    fp_add(d, t, q.y);                               // Y: normalised, < 6p
    fp_mul(chk, d, one);
    ok &= fp_is_zero_n(chk);
    if (!ok) atomicOr(bad, 4u);
}

// the same test, one LANE per point, for whole setup files (throughput form: 2^24 points are ~2 x 10^9 point operations,
// ~0.3 s -- the lane-parallel form above is for a few hundred untrusted commitments where latency counts)
__global__ void __launch_bounds__(256) k_g1_subgroup_check(const g1_affine_t* __restrict__ in, uint64_t n,
                                                            uint32_t* __restrict__ bad) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    g1_aff28 a;
    g1_load_aff(a, &in[j]);
    if (g1_aff_is_inf(a)) return;
    g1_xyzz_t acc, t;
    g1_from_aff(acc, a);
    for (int b = 62; b >= 0; b--) {                    // acc <- [z] P : the base is affine, mixed additions
        g1_dbl(t, acc);
        acc = t;
        if ((BLS_Z >> b) & 1ull) g1_madd_checked(acc, a);
    }
    g1_xyzz_t base = acc, q = acc;
    for (int b = 62; b >= 0; b--) {                    // q <- [z] ([z] P)
        g1_dbl(t, q);
        q = t;
        if ((BLS_Z >> b) & 1ull) {
            g1_add(t, q, base);
            q = t;
        }
    }
    if (g1_is_inf(q)) { atomicOr(bad, 4u); return; }
    fp_t beta, one, u, d, chk;
#pragma unroll
    for (int i = 0; i < 14; i++) beta.l[i] = g1_beta_mont(i);
    fp_one(one);
    fp_mul(u, beta, a.x);
    fp_mul(u, u, q.zz);
    fp_sub16(d, u, q.x);
    fp_mul(chk, d, one);
    bool ok = fp_is_zero_n(chk);
    fp_mul(u, a.y, q.zzz);
    fp_add(d, u, q.y);
    fp_mul(chk, d, one);
    ok &= fp_is_zero_n(chk);
    if (!ok) atomicOr(bad, 4u);
}

__global__ void __launch_bounds__(64) k_g1_compress(const g1_xyzz_t* __restrict__ in, uint8_t* __restrict__ out48) {
    tail_priority();
    if (threadIdx.x != 0) return;
    g1_xyzz_t p;
    load_xyzz(p, in);
    g1_aff28 a;
    g1_to_aff(a, p);
    g1_compress(out48, a);
}
// two points, ONE inversion (Montgomery's trick): commit + open share the single-lane inversion latency
__global__ void __launch_bounds__(64) k_g1_compress_pair(const g1_xyzz_t* __restrict__ in0,
                                                          const g1_xyzz_t* __restrict__ in1,
                                                          uint8_t* __restrict__ out0, uint8_t* __restrict__ out1) {
    tail_priority();
    if (threadIdx.x != 0) return;
    g1_xyzz_t p0, p1;
    load_xyzz(p0, in0);
    load_xyzz(p1, in1);
    const bool i0 = g1_is_inf(p0), i1 = g1_is_inf(p1);
    fp_t d0, d1, one, t, inv, w0, w1;
    fp_one(one);
    fp_mul(d0, p0.zz, p0.zzz);
    fp_mul(d1, p1.zz, p1.zzz);
    if (i0) d0 = one;
    if (i1) d1 = one;
    fp_mul(t, d0, d1);
    fp_inv(inv, t);
    fp_mul(w0, inv, d1);  // 1 / d0
    fp_mul(w1, inv, d0);  // 1 / d1
    g1_aff28 a0, a1;
    fp_zero(a0.x); fp_zero(a0.y); fp_zero(a1.x); fp_zero(a1.y);
    if (!i0) {
        fp_mul(t, w0, p0.zzz); fp_mul(t, p0.x, t); fp_canon(a0.x, t);
        fp_mul(t, w0, p0.zz); fp_mul(t, p0.y, t); fp_canon(a0.y, t);
    }
    if (!i1) {
        fp_mul(t, w1, p1.zzz); fp_mul(t, p1.x, t); fp_canon(a1.x, t);
        fp_mul(t, w1, p1.zz); fp_mul(t, p1.y, t); fp_canon(a1.y, t);
    }
    g1_compress(out0, a0);
    g1_compress(out1, a1);
}
// the lane's tail record -> its pinned host page (device-visible host memory): a 1-wave store instead of a copy-engine
// transfer (~16 us on this stack for 832 bytes); the stream synchronisation that follows makes it visible to the host
// clear2 != null: the two input-error flags of the record are zeroed once copied, so the lane's next request finds them
// clean without a memset at its start
// seq_word != null: once every word is out (system-scope fence), the host page's sequence word is set to `seq` -- the
// host polls it instead of waiting for the stream's completion signal (kernel end + signal + wake-up: several us)
__global__ void __launch_bounds__(256) k_publish(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst_host, uint32_t words,
                                                 uint32_t* __restrict__ clear2, uint32_t* __restrict__ seq_word, uint32_t seq) {
    for (uint32_t i = threadIdx.x; i < words; i += 256) dst_host[i] = src[i];
    if (clear2 || seq_word) {
        __threadfence_system();
        __syncthreads();
        if (clear2 && threadIdx.x < 2) clear2[threadIdx.x] = 0;
        if (seq_word && threadIdx.x == 0) {
            __hip_atomic_store(seq_word, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}
// XYZZ working form <-> the 192-byte partial-sum format of the C-ABI (4 x 12 u32: canonical Montgomery residues)
__global__ void __launch_bounds__(64) k_xyzz_pack(const g1_xyzz_t* __restrict__ in, uint32_t* __restrict__ out48w,
                                                   uint32_t count) {
    tail_priority();
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    g1_xyzz_t p;
    load_xyzz(p, &in[i]);
    const bool inf = g1_is_inf(p);
    fp_t c;
    const fp_t* f[4] = {&p.x, &p.y, &p.zz, &p.zzz};
#pragma unroll
    for (int k = 0; k < 4; k++) {
        fp_canon_mont(c, *f[k]);
        if (inf) fp_zero(c);
        fp_pack(out48w + 48 * i + 12 * k, c);
    }
}
__global__ void __launch_bounds__(64) k_xyzz_unpack(const uint32_t* __restrict__ in48w, g1_xyzz_t* __restrict__ out,
                                                     uint32_t count) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    g1_xyzz_t p;
    fp_t* f[4] = {&p.x, &p.y, &p.zz, &p.zzz};
#pragma unroll
    for (int k = 0; k < 4; k++) fp_unpack(*f[k], in48w + 48 * i + 12 * k);
    store_xyzz(&out[i], p);
}


// ------------------------------------------------------------------------------------------------ launchers
void launch_g1_sum(hipStream_t s, const g1_xyzz_t* in, uint32_t count, g1_xyzz_t* out_xyzz) {
    if (count >= 2 && count <= 32) k_g1_sum_lp<<<1, 512, 0, s>>>(in, count, out_xyzz);
    else k_g1_sum<<<1, 64, 0, s>>>(in, count, out_xyzz);
}
void launch_g1_sum_affine(hipStream_t s, const g1_affine_t* in, uint32_t count, g1_xyzz_t* out_xyzz) {
    k_g1_sum_affine<<<1, 64, 0, s>>>(in, count, out_xyzz);
}
void launch_g1_subgroup_check(hipStream_t s, const g1_affine_t* in, uint32_t count, uint32_t* bad_flag) {
    if (count) k_g1_subgroup_check_lp<<<count, 64, 0, s>>>(in, count, bad_flag);
}
void launch_g1_subgroup_check_bulk(hipStream_t s, const g1_affine_t* in, uint64_t n, uint32_t* bad_flag) {
    if (n) k_g1_subgroup_check<<<nblk(n, 256), 256, 0, s>>>(in, n, bad_flag);
}
void launch_g1_compress(hipStream_t s, const g1_xyzz_t* in, uint8_t* out48) {
    k_g1_compress<<<1, 64, 0, s>>>(in, out48);
}
void launch_g1_compress_pair(hipStream_t s, const g1_xyzz_t* in0, const g1_xyzz_t* in1, uint8_t* out0, uint8_t* out1) {
    k_g1_compress_pair<<<1, 64, 0, s>>>(in0, in1, out0, out1);
}
void launch_publish(hipStream_t s, const void* src_dev, void* dst_host_devptr, uint32_t bytes, uint32_t* clear2,
                    uint32_t* seq_word_devptr, uint32_t seq) {
    k_publish<<<1, 256, 0, s>>>(reinterpret_cast<const uint32_t*>(src_dev), reinterpret_cast<uint32_t*>(dst_host_devptr), bytes / 4,
                                clear2, seq_word_devptr, seq);
}
void launch_xyzz_pack(hipStream_t s, const g1_xyzz_t* in, uint32_t* out48w, uint32_t count) {
    if (count) k_xyzz_pack<<<nblk(count, 64), 64, 0, s>>>(in, out48w, count);
}
void launch_xyzz_unpack(hipStream_t s, const uint32_t* in48w, g1_xyzz_t* out, uint32_t count) {
    if (count) k_xyzz_unpack<<<nblk(count, 64), 64, 0, s>>>(in48w, out, count);
}
