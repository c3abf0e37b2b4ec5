// Fr-side kernels of the KZG worker path: wire codec, radix-2 Cooley-Tukey NTT with LDS-staged butterflies,
// and the opening's evaluation + quotient as a chunked linear-recurrence scan.
// Replaces what the external prover does behind fft(poly,left,inverse) / eval / the implicit IFFT + synthetic
// division of worker_commit / worker_open (reference neurons/validator.py:59-65,98-104; neurons/miner.py:39,48).
// Domain convention: w_n = 7^((r-1)/n), natural order in and out, inverse carries 1/n.
#include "fr_kernels.cuh"

static inline uint32_t nblk(uint64_t n, uint32_t b) { return (uint32_t)((n + b - 1) / b); }

KZG_DEV void fr_load(fr_t& v, const uint32_t* p) {
    const uint4* q = reinterpret_cast<const uint4*>(p);
    uint4 a = q[0], b = q[1];
    v.l[0] = a.x; v.l[1] = a.y; v.l[2] = a.z; v.l[3] = a.w;
    v.l[4] = b.x; v.l[5] = b.y; v.l[6] = b.z; v.l[7] = b.w;
}
KZG_DEV void fr_store(uint32_t* p, const fr_t& v) {
    uint4* q = reinterpret_cast<uint4*>(p);
    q[0] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
    q[1] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
}

// ------------------------------------------------------------------------------------------------ codec
__global__ void __launch_bounds__(256) k_fr_from_be(const uint8_t* __restrict__ be, uint32_t* __restrict__ out,
                                                     uint64_t n, int to_mont, uint32_t* __restrict__ bad) {
    uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    fr_t v;
    limbs_from_be<8>(v.l, be + 32 * j);
    uint32_t rm[8];
#pragma unroll
    for (int i = 0; i < 8; i++) rm[i] = FrParams::mod(i);
    if (bi_ge<8>(v.l, rm)) atomicOr(bad, 1u);  // non-canonical scalar: the call fails (SURVEY 8b errors)
    if (to_mont) f_to_mont(v, v);
    fr_store(out + 8 * j, v);
}
__global__ void __launch_bounds__(256) k_fr_to_be(const uint32_t* __restrict__ in, uint8_t* __restrict__ be,
                                                   uint64_t n, int from_mont) {
    uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    fr_t v;
    fr_load(v, in + 8 * j);
    if (from_mont) f_from_mont(v, v);
    limbs_to_be<8>(be + 32 * j, v.l);
}
__global__ void __launch_bounds__(256) k_fr_from_mont(const uint32_t* __restrict__ in, uint32_t* __restrict__ out,
                                                       uint64_t n) {
    uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    fr_t v;
    fr_load(v, in + 8 * j);
    f_from_mont(v, v);
    fr_store(out + 8 * j, v);
}

// ------------------------------------------------------------------------------------------------ twiddles
// w_{2^32} = 7^((r-1)/2^32) and its inverse, canonical limbs (re-derived in oracle/bls12_381.py)
KZG_DEV void fr_root_2_32(fr_t& w, int inverse) {
    constexpr uint32_t W[8] = {0x439f0d2bu, 0x3829971fu, 0x8c2280b9u, 0xb6368350u,
                               0x22c813b4u, 0xd09b6819u, 0xdfe81f20u, 0x16a2a19eu};
    constexpr uint32_t WI[8] = {0x3cf19a78u, 0x0fb4d6e1u, 0xb566f833u, 0x6f67d4a2u,
                                0xa35d0168u, 0xed4f2f74u, 0x6e19c653u, 0x0538a6f6u};
#pragma unroll
    for (int i = 0; i < 8; i++) w.l[i] = inverse ? WI[i] : W[i];
    f_to_mont(w, w);
}
// tw[k] = w_n^(+-k), k < n/2, Montgomery form; 64 consecutive k per lane
__global__ void __launch_bounds__(256) k_fr_twiddles(uint32_t* __restrict__ tw, int log_n, int inverse) {
    const uint64_t half = (uint64_t)1 << (log_n - 1);
    uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t k0 = t * 64;
    if (k0 >= half) return;
    fr_t w, cur, pw;
    fr_root_2_32(w, inverse);
    for (int i = log_n; i < 32; i++) f_mul(w, w, w);
    f_one(cur);
    pw = w;
    for (uint64_t e = k0; e; e >>= 1) {
        if (e & 1) f_mul(cur, cur, pw);
        f_mul(pw, pw, pw);
    }
    for (uint64_t k = k0; k < k0 + 64 && k < half; k++) {
        fr_store(tw + 8 * k, cur);
        f_mul(cur, cur, w);
    }
}
// out[0] = 2^-log_n in Montgomery form
__global__ void k_fr_inv_pow2(uint32_t* __restrict__ out, int log_n) {
    if (threadIdx.x || blockIdx.x) return;
    constexpr uint32_t HALF[8] = {0x80000001u, 0x7fffffffu, 0x7fff2dffu, 0xa9ded201u,
                                  0x04d0ec02u, 0x199cec04u, 0x94cebea4u, 0x39f6d3a9u};  // (r+1)/2
    fr_t h, acc;
#pragma unroll
    for (int i = 0; i < 8; i++) h.l[i] = HALF[i];
    f_to_mont(h, h);
    f_one(acc);
    for (int i = 0; i < log_n; i++) f_mul(acc, acc, h);
    fr_store(out, acc);
}

// ------------------------------------------------------------------------------------------------ NTT
__global__ void __launch_bounds__(256) k_fr_bitrev(const uint32_t* __restrict__ in, uint32_t* __restrict__ out,
                                                    int log_n) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >> log_n) return;
    uint64_t j = log_n ? (__brevll(i) >> (64 - log_n)) : 0;
    fr_t v;
    fr_load(v, in + 8 * i);
    fr_store(out + 8 * j, v);
}

#define NTT_TILE_LOG 10
#define NTT_TILE (1 << NTT_TILE_LOG)
// Stages 0 .. S-1 (S = min(log_n, 10)) of the DIT network on a contiguous 2^S tile staged in LDS:
// one HBM read + one HBM write for up to ten butterfly stages.
__global__ void __launch_bounds__(256) k_fr_ntt_lds(uint32_t* __restrict__ data, int log_n,
                                                     const uint32_t* __restrict__ tw) {
    __shared__ uint4 sm[NTT_TILE * 2];  // 1024 x 32 B
    const int S = log_n < NTT_TILE_LOG ? log_n : NTT_TILE_LOG;
    const uint32_t tile = 1u << S;
    const uint64_t base = (uint64_t)blockIdx.x * tile;
    const uint4* src = reinterpret_cast<const uint4*>(data + 8 * base);
    for (uint32_t i = threadIdx.x; i < tile * 2; i += blockDim.x) sm[i] = src[i];
    __syncthreads();
    for (int s = 0; s < S; s++) {
        const uint32_t half = 1u << s;
        for (uint32_t b = threadIdx.x; b < tile / 2; b += blockDim.x) {
            const uint32_t k = b & (half - 1);
            const uint32_t i = ((b >> s) << (s + 1)) + k, j = i + half;
            fr_t u, v, w, t;
            fr_load(u, reinterpret_cast<const uint32_t*>(&sm[2 * i]));
            fr_load(v, reinterpret_cast<const uint32_t*>(&sm[2 * j]));
            fr_load(w, tw + 8 * ((uint64_t)k << (log_n - s - 1)));
            f_mul(t, v, w);
            f_add(v, u, t);
            f_sub(w, u, t);
            fr_store(reinterpret_cast<uint32_t*>(&sm[2 * i]), v);
            fr_store(reinterpret_cast<uint32_t*>(&sm[2 * j]), w);
        }
        __syncthreads();
    }
    uint4* dst = reinterpret_cast<uint4*>(data + 8 * base);
    for (uint32_t i = threadIdx.x; i < tile * 2; i += blockDim.x) dst[i] = sm[i];
}
// one global radix-2 stage s >= 10
__global__ void __launch_bounds__(256) k_fr_ntt_stage(uint32_t* __restrict__ data, int log_n, int s,
                                                       const uint32_t* __restrict__ tw) {
    const uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >> (log_n - 1)) return;
    const uint64_t half = (uint64_t)1 << s;
    const uint64_t k = b & (half - 1);
    const uint64_t i = ((b >> s) << (s + 1)) + k, j = i + half;
    fr_t u, v, w, t;
    fr_load(u, data + 8 * i);
    fr_load(v, data + 8 * j);
    fr_load(w, tw + 8 * (k << (log_n - s - 1)));
    f_mul(t, v, w);
    f_add(v, u, t);
    f_sub(w, u, t);
    fr_store(data + 8 * i, v);
    fr_store(data + 8 * j, w);
}
__global__ void __launch_bounds__(256) k_fr_scale(uint32_t* __restrict__ data, uint64_t n,
                                                   const uint32_t* __restrict__ factor) {
    uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    fr_t v, f;
    fr_load(v, data + 8 * j);
    fr_load(f, factor);
    f_mul(v, v, f);
    fr_store(data + 8 * j, v);
}

// ------------------------------------------------------------------------------------------------ eval + quotient
// Chunk length L = 2^lchunk coefficients per lane: 64 for large polynomials (throughput), down to 4 for small rows
// where the chunk loops are pure latency (a 2^12 row: 64-long serial Horner loops cost 0.2 ms, 8-long ones 0.03 ms).
static inline int poly_lchunk(uint64_t n) {
    int l = 2;
    while (l < 6 && (n >> (l + 1)) >= 16384) l++;
    return l;
}
// h[t] = sum_k f[t*L + k] alpha^k
__global__ void __launch_bounds__(256) k_poly_chunk_eval(const uint32_t* __restrict__ f, uint64_t n, int lchunk,
                                                          const uint32_t* __restrict__ alpha_mont,
                                                          uint32_t* __restrict__ h) {
    uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t L = (uint64_t)1 << lchunk;
    uint64_t lo = t * L;
    if (lo >= n) return;
    uint64_t hi = lo + L < n ? lo + L : n;
    fr_t a, s, c;
    fr_load(a, alpha_mont);
    f_zero(s);
    for (uint64_t j = hi; j-- > lo;) {
        fr_load(c, f + 8 * j);
        f_mul(s, s, a);
        f_add(s, s, c);
    }
    fr_store(h + 8 * t, s);
}
// Suffix recurrence over chunks, H_t = h_t + beta H_{t+1}, beta = alpha^L: one 1024-lane block; lane v serially
// folds m consecutive chunks, then a Hillis-Steele suffix scan whose multiplier (beta^m)^(2^step) is uniform.
// Writes hnext[t] = H_{t+1} and y = H_0 = f(alpha).
__global__ void __launch_bounds__(1024) k_poly_chunk_scan(const uint32_t* __restrict__ h, uint64_t nchunks, int lchunk,
                                                           const uint32_t* __restrict__ alpha_mont,
                                                           uint32_t* __restrict__ hnext, uint32_t* __restrict__ y_mont) {
    __shared__ uint4 sm[1024 * 2];
    const uint32_t v = threadIdx.x;
    const uint64_t m = (nchunks + 1023) / 1024;
    const uint64_t lo = (uint64_t)v * m;
    const uint64_t hi = lo + m < nchunks ? lo + m : nchunks;
    fr_t beta, g, c, mult;
    fr_load(beta, alpha_mont);
    for (int i = 0; i < lchunk; i++) f_mul(beta, beta, beta);  // alpha^L
    f_zero(g);
    for (uint64_t u = hi; u-- > lo && lo < nchunks;) {
        fr_load(c, h + 8 * u);
        f_mul(g, g, beta);
        f_add(g, g, c);
    }
    // mult = beta^m
    f_one(mult);
    {
        fr_t pw = beta;
        for (uint64_t e = m; e; e >>= 1) {
            if (e & 1) f_mul(mult, mult, pw);
            f_mul(pw, pw, pw);
        }
    }
    fr_store(reinterpret_cast<uint32_t*>(&sm[2 * v]), g);
    __syncthreads();
    for (uint32_t d = 1; d < 1024; d <<= 1) {
        fr_t other;
        f_zero(other);
        if (v + d < 1024) fr_load(other, reinterpret_cast<const uint32_t*>(&sm[2 * (v + d)]));
        __syncthreads();
        f_mul(other, other, mult);
        f_add(g, g, other);
        fr_store(reinterpret_cast<uint32_t*>(&sm[2 * v]), g);
        f_mul(mult, mult, mult);
        __syncthreads();
    }
    // g == H_{lo}; walk the lane's own chunks downward from H_{hi}
    fr_t s;
    f_zero(s);
    if (v + 1 < 1024) fr_load(s, reinterpret_cast<const uint32_t*>(&sm[2 * (v + 1)]));
    for (uint64_t u = hi; u-- > lo && lo < nchunks;) {
        fr_store(hnext + 8 * u, s);
        fr_load(c, h + 8 * u);
        f_mul(s, s, beta);
        f_add(s, s, c);
    }
    if (v == 0) fr_store(y_mont, s);
}
// q[j-1] = sum_{k>=j} f_k alpha^(k-j), written canonical (ready to be MSM scalars); q has n-1 entries
__global__ void __launch_bounds__(256) k_poly_quotient(const uint32_t* __restrict__ f, uint64_t n, int lchunk,
                                                        const uint32_t* __restrict__ alpha_mont,
                                                        const uint32_t* __restrict__ hnext,
                                                        uint32_t* __restrict__ q_canon) {
    uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t L = (uint64_t)1 << lchunk;
    uint64_t lo = t * L;
    if (lo >= n) return;
    uint64_t hi = lo + L < n ? lo + L : n;
    fr_t a, s, c, o;
    fr_load(a, alpha_mont);
    fr_load(s, hnext + 8 * t);
    for (uint64_t j = hi; j-- > lo;) {
        fr_load(c, f + 8 * j);
        f_mul(s, s, a);
        f_add(s, s, c);
        if (j >= 1) {
            f_from_mont(o, s);
            fr_store(q_canon + 8 * (j - 1), o);
        }
    }
}

// ------------------------------------------------------------------------------------------------ launchers
void launch_fr_from_be(hipStream_t s, const uint8_t* be, uint32_t* out, uint64_t n, int to_mont, uint32_t* bad) {
    if (n) k_fr_from_be<<<nblk(n, 256), 256, 0, s>>>(be, out, n, to_mont, bad);
}
void launch_fr_to_be(hipStream_t s, const uint32_t* in, uint8_t* be, uint64_t n, int from_mont) {
    if (n) k_fr_to_be<<<nblk(n, 256), 256, 0, s>>>(in, be, n, from_mont);
}
void launch_fr_from_mont(hipStream_t s, const uint32_t* in, uint32_t* out, uint64_t n) {
    if (n) k_fr_from_mont<<<nblk(n, 256), 256, 0, s>>>(in, out, n);
}
void launch_fr_twiddles(hipStream_t s, uint32_t* tw, int log_n, int inverse) {
    if (log_n < 1) return;
    uint64_t half = (uint64_t)1 << (log_n - 1);
    k_fr_twiddles<<<nblk((half + 63) / 64, 256), 256, 0, s>>>(tw, log_n, inverse);
}
void launch_fr_inv_pow2(hipStream_t s, uint32_t* out, int log_n) { k_fr_inv_pow2<<<1, 64, 0, s>>>(out, log_n); }
void launch_fr_ntt(hipStream_t s, const uint32_t* in, uint32_t* out, int log_n, const uint32_t* tw,
                   const uint32_t* scale_or_null) {
    const uint64_t n = (uint64_t)1 << log_n;
    k_fr_bitrev<<<nblk(n, 256), 256, 0, s>>>(in, out, log_n);
    if (log_n >= 1) {
        const int S = log_n < NTT_TILE_LOG ? log_n : NTT_TILE_LOG;
        k_fr_ntt_lds<<<(uint32_t)(n >> S), 256, 0, s>>>(out, log_n, tw);
        for (int st = S; st < log_n; st++) k_fr_ntt_stage<<<nblk(n / 2, 256), 256, 0, s>>>(out, log_n, st, tw);
    }
    if (scale_or_null) k_fr_scale<<<nblk(n, 256), 256, 0, s>>>(out, n, scale_or_null);
}
void launch_poly_open(hipStream_t s, const uint32_t* f_mont, uint64_t n, const uint32_t* alpha_mont, uint32_t* h,
                      uint32_t* hnext, uint32_t* y_mont, uint32_t* q_canon_or_null) {
    if (!n) return;
    const int lchunk = poly_lchunk(n);
    const uint64_t nchunks = (n + ((uint64_t)1 << lchunk) - 1) >> lchunk;
    k_poly_chunk_eval<<<nblk(nchunks, 256), 256, 0, s>>>(f_mont, n, lchunk, alpha_mont, h);
    k_poly_chunk_scan<<<1, 1024, 0, s>>>(h, nchunks, lchunk, alpha_mont, hnext, y_mont);
    if (q_canon_or_null)
        k_poly_quotient<<<nblk(nchunks, 256), 256, 0, s>>>(f_mont, n, lchunk, alpha_mont, hnext, q_canon_or_null);
}
