// fr_poly.hip -- Fr-side kernels, part 2: the opening's evaluation y = f(alpha) and quotient (f - y) / (X - alpha) as a
// chunked linear-recurrence scan (no inversion anywhere: alpha = 0 or a root of unity need no special case), and the word
// comparison behind the row cache's verification.  Replaces eval(poly, x) and the synthetic division of worker_open
// (reference neurons/validator.py:98-104; neurons/miner.py:48).
#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "fr_kernels.hip.h"

static inline uint32_t nblk(uint64_t n, uint32_t b) { return (uint32_t)((n + b - 1) / b); }

// ------------------------------------------------------------------------------------------------ eval + quotient
// Chunk length L = 2^lchunk coefficients per lane: 16 for large polynomials, down to 4 for small rows: the chunk loops
// are chains of dependent Fr products, so short chunks + more levels beat long ones.  Inside a chain the running value
// stays lazy (product output + one canonical coefficient: < 3r) and is canonicalised once, when it is stored.
static inline int poly_lchunk(uint64_t n) {
    int l = 2;
    while (l < 4 && (n >> (l + 1)) >= 16384) l++;
    return l;
}
KZG_DEV void fr9_pow2k(fr9_t& a, int k) {  // a <- a^(2^k), canonical in and out
    for (int i = 0; i < k; i++) fr9_mul(a, a, a);
    fr9_canon(a, a);
}
// h[t] = sum_k f[t*L + k] a^k with a = alpha^(2^sq)  (sq > 0: f is itself an array of chunk values, second level)
// ARG: alpha comes as the kernel ARGUMENT (its 32 big-endian bytes) instead of from memory: the first kernel of an
// opening converts it itself -- every lane, it is one product -- and lane 0 leaves the Montgomery form at alpha_out for
// the kernels behind it (and raises *bad for a value >= r); a 1-lane conversion kernel ahead of it was ~5 us of latency
template <bool ARG>
__global__ void __launch_bounds__(256) k_poly_chunk_eval(const uint32_t* __restrict__ f, uint64_t n, int lchunk,
                                                          const uint32_t* __restrict__ alpha_mont, int sq,
                                                          uint32_t* __restrict__ h, const FrArg arg,
                                                          uint32_t* __restrict__ alpha_out, uint32_t* __restrict__ bad) {
    uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t L = (uint64_t)1 << lchunk;
    uint64_t lo = t * L;
    fr9_t a, s, c;
    if constexpr (ARG) {
        uint32_t w[8];
#pragma unroll
        for (int i = 0; i < 8; i++) w[i] = bswap32(arg.w[7 - i]);
        fr9_from_words(a, w);
        fr9_to_mont(a, a);
        if (t == 0) {
            if (fr_words_ge_r(w)) atomicOr(bad, 1u);
            fr9_store(alpha_out, a);
        }
    }
    if (lo >= n) return;
    uint64_t hi = lo + L < n ? lo + L : n;
    if constexpr (!ARG) fr9_load(a, alpha_mont);
    fr9_pow2k(a, sq);
    fr9_zero(s);
    for (uint64_t j = hi; j-- > lo;) {
        fr9_load(c, f + 8 * j);
        fr9_mul(s, s, a);
        fr9_add(s, s, c);
    }
    fr9_reduce(s, s);
    fr9_store(h + 8 * t, s);
}
// Suffix recurrence over chunks, H_t = h_t + beta H_{t+1}, beta = alpha^L: one NT_-lane block; lane v serially
// folds m consecutive chunks, then a Hillis-Steele suffix scan whose multiplier (beta^m)^(2^step) is uniform.
// Writes hnext[t] = H_{t+1} and y = H_0 = f(alpha).
template <uint32_t NT_>
__global__ void __launch_bounds__(NT_) k_poly_chunk_scan(const uint32_t* __restrict__ h, uint64_t nchunks, int lchunk,
                                                           const uint32_t* __restrict__ alpha_mont,
                                                           uint32_t* __restrict__ hnext, uint32_t* __restrict__ y_mont,
                                                           uint8_t* __restrict__ y_be_or_null) {
    __shared__ uint32_t sm[9][NT_];
    const uint32_t v = threadIdx.x;
    const uint64_t m = (nchunks + NT_ - 1) / NT_;
    const uint64_t lo = (uint64_t)v * m;
    const uint64_t hi = lo + m < nchunks ? lo + m : nchunks;
    fr9_t beta, g, c, mult;
    fr9_load(beta, alpha_mont);
    fr9_pow2k(beta, lchunk);  // alpha^L
    fr9_zero(g);
    for (uint64_t u = hi; u-- > lo && lo < nchunks;) {
        fr9_load(c, h + 8 * u);
        fr9_mul(g, g, beta);
        fr9_add(g, g, c);
    }
    fr9_norm(g, g);           // < 3r, normalised
    // mult = beta^m (N class, renormalised products)
    fr9_one(mult);
    {
        fr9_t pw = beta;
        for (uint64_t e = m; e; e >>= 1) {
            if (e & 1) fr9_mul(mult, mult, pw);
            fr9_mul(pw, pw, pw);
        }
    }
#pragma unroll
    for (int i = 0; i < 9; i++) sm[i][v] = g.l[i];
    __syncthreads();
    for (uint32_t d = 1; d < NT_; d <<= 1) {   // g grows by < 2r per step: < 3r + 20r at the end, always normalised
        fr9_t other;
        fr9_zero(other);
        if (v + d < NT_) {
#pragma unroll
            for (int i = 0; i < 9; i++) other.l[i] = sm[i][v + d];
        }
        __syncthreads();
        fr9_mul(other, other, mult);
        fr9_add(g, g, other);
        fr9_norm(g, g);
#pragma unroll
        for (int i = 0; i < 9; i++) sm[i][v] = g.l[i];
        fr9_mul(mult, mult, mult);
        __syncthreads();
    }
    // g == H_{lo}; walk the lane's own chunks downward from H_{hi}
    fr9_t s;
    fr9_zero(s);
    if (v + 1 < NT_) {
#pragma unroll
        for (int i = 0; i < 9; i++) s.l[i] = sm[i][v + 1];
    }
    for (uint64_t u = hi; u-- > lo && lo < nchunks;) {
        fr9_t o;
        fr9_reduce(o, s);
        fr9_store(hnext + 8 * u, o);
        fr9_load(c, h + 8 * u);
        fr9_mul(s, s, beta);
        fr9_add(s, s, c);
    }
    if (v == 0) {
        fr9_reduce(s, s);
        fr9_store(y_mont, s);
        if (y_be_or_null) {   // the evaluation as the wire carries it (32 bytes big-endian): no separate 1-lane kernel
            fr9_t yc;
            fr9_from_mont(yc, s);
            uint32_t w[8];
            fr9_to_words(w, yc);
            limbs_to_be<8>(y_be_or_null, w);
        }
    }
}
// second level back down: hnext2[g] = H_{(g+1) * L2} over groups of L2 = 2^l2 first-level chunks -> hnext[u] = H_{u+1}
__global__ void __launch_bounds__(256) k_poly_chunk_expand(const uint32_t* __restrict__ h, uint64_t nchunks, int l2,
                                                            const uint32_t* __restrict__ alpha_mont, int sq,
                                                            const uint32_t* __restrict__ hnext2,
                                                            uint32_t* __restrict__ hnext) {
    uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t L = (uint64_t)1 << l2;
    uint64_t lo = g * L;
    if (lo >= nchunks) return;
    uint64_t hi = lo + L < nchunks ? lo + L : nchunks;
    fr9_t beta, s, c;
    fr9_load(beta, alpha_mont);
    fr9_pow2k(beta, sq);
    fr9_load(s, hnext2 + 8 * g);
    for (uint64_t u = hi; u-- > lo;) {
        fr9_t o;
        fr9_reduce(o, s);
        fr9_store(hnext + 8 * u, o);
        fr9_load(c, h + 8 * u);
        fr9_mul(s, s, beta);
        fr9_add(s, s, c);
    }
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
    fr9_t a, s, c, o;
    fr9_load(a, alpha_mont);
    fr9_load(s, hnext + 8 * t);
    for (uint64_t j = hi; j-- > lo;) {
        fr9_load(c, f + 8 * j);
        fr9_mul(s, s, a);
        fr9_add(s, s, c);
        if (j >= 1) {
            fr9_from_mont(o, s);
            fr9_store(q_canon + 8 * (j - 1), o);
        }
    }
    if (hi == n) {  // slot n - 1: a zero, so that the n - 1 coefficients can ride as a length-n scalar set (batched commit+open)
        fr9_zero(o);
        fr9_store(q_canon + 8 * (n - 1), o);
    }
}

// ---- long rows (16 coefficients per lane): the quotient (and, as an A/B form, the level-0 fold) with the coefficients
// staged through LDS.  A lane of the kernels above walks ITS 512-byte chunk, so one wave load touches 64 pieces of 32
// bytes at a 512-byte stride (k_poly_quotient: 2.3 TB/s for 256 MB, with the caches reassembling the lines).  Here ONE
// WAVE per workgroup takes 64 consecutive chunks (32 KB of the vector) in PHASES of PQ_CO coefficients per chunk: each
// part is moved between HBM and LDS by the whole wave in 16-byte units -- a wave instruction covers whole 128-byte
// segments -- and the lanes run their recurrences out of (and, for the quotient, back into) padded LDS rows.  With four
// coefficients per phase a wave holds 9 KB of LDS and 8 staged loads: enough waves per SIMD to overlap one wave's
// transfers with another's products (uncontended: 115 us strided -> 86 us with 8 per phase -> ~70 us with 4).
// PQ_CO coefficients of every chunk per phase (8: two phases, 17-unit rows; 4: four phases, 9-unit rows -- half the LDS
// and registers per wave, twice the phases).  Row stride = 2 PQ_CO + 1 units: odd, so consecutive rows start 4 banks apart.
#ifndef PQ_CO
#define PQ_CO 4   // same-box A/B at 2^22 (profiles/r04_ab_opening_lds_phases.log): opening 0.253 (8) -> 0.236 (4) -> 0.272 ms (2)
#endif
#define PQ_ROW (2 * PQ_CO + 1)
#define PQ_PHASES (16 / PQ_CO)
#define PQ_ITERS (2 * PQ_CO)        // 16-byte units per lane and phase: 64 rows x 2 PQ_CO units / 64 lanes
KZG_DEV void pq_load_part(uint4 (*sm)[PQ_ROW], const uint4* __restrict__ src, uint32_t uoff, uint32_t lane) {
    uint4 tmp[PQ_ITERS];
#pragma unroll
    for (int i = 0; i < PQ_ITERS; i++) {   // all loads in flight before the first LDS write
        const uint32_t u = (uint32_t)i * 64 + lane;
        tmp[i] = src[(uint64_t)(u / (2 * PQ_CO)) * 32 + uoff + (u % (2 * PQ_CO))];
    }
#pragma unroll
    for (int i = 0; i < PQ_ITERS; i++) {
        const uint32_t u = (uint32_t)i * 64 + lane;
        sm[u / (2 * PQ_CO)][u % (2 * PQ_CO)] = tmp[i];
    }
}
KZG_DEV void pq_row_load(fr9_t& v, const uint4* row, int k) {
    const uint4 a = row[2 * k], b = row[2 * k + 1];
    const uint32_t w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    fr9_from_words(v, w);
}
// q[j-1] = sum_{k>=j} f_k alpha^(k-j), canonical, for the 1024 coefficients of this workgroup; q[n-1] = 0
__global__ void __launch_bounds__(64) k_poly_quotient16_lds(const uint32_t* __restrict__ f, uint64_t n,
                                                             const uint32_t* __restrict__ alpha_mont,
                                                             const uint32_t* __restrict__ hnext,
                                                             uint32_t* __restrict__ q_canon) {
    __shared__ uint4 sm[64][PQ_ROW];
    const uint32_t lane = threadIdx.x;
    const uint64_t chunk0 = (uint64_t)blockIdx.x * 64;
    const uint4* src = reinterpret_cast<const uint4*>(f) + chunk0 * 32;
    uint4* dst = reinterpret_cast<uint4*>(q_canon);
    fr9_t a, s, c, o;
    fr9_load(a, alpha_mont);
    fr9_load(s, hnext + 8 * (chunk0 + lane));
    for (int ph = 0; ph < PQ_PHASES; ph++) {
        const uint32_t uoff = (uint32_t)(PQ_PHASES - 1 - ph) * 2 * PQ_CO;
        pq_load_part(sm, src, uoff, lane);
        __syncthreads();
#pragma unroll 2
        for (int k = PQ_CO - 1; k >= 0; k--) {
            pq_row_load(c, sm[lane], k);
            fr9_mul(s, s, a);
            fr9_add(s, s, c);
            fr9_from_mont(o, s);
            uint32_t w[8];
            fr9_to_words(w, o);
            sm[lane][2 * k] = make_uint4(w[0], w[1], w[2], w[3]);       // in place: the slot the coefficient came from
            sm[lane][2 * k + 1] = make_uint4(w[4], w[5], w[6], w[7]);
        }
        __syncthreads();
        // the value computed at coefficient j is q[j - 1]: the whole part moves down by one coefficient (two units)
#pragma unroll
        for (int i = 0; i < PQ_ITERS; i++) {
            const uint32_t u = (uint32_t)i * 64 + lane;
            const uint64_t gu = (chunk0 + u / (2 * PQ_CO)) * 32 + uoff + (u % (2 * PQ_CO));
            if (gu >= 2) dst[gu - 2] = sm[u / (2 * PQ_CO)][u % (2 * PQ_CO)];
        }
        __syncthreads();
    }
    if ((chunk0 + lane + 1) * 16 == n) {  // slot n - 1: a zero, so that the n - 1 coefficients ride as a length-n scalar set
        dst[2 * (n - 1)] = make_uint4(0u, 0u, 0u, 0u);
        dst[2 * (n - 1) + 1] = make_uint4(0u, 0u, 0u, 0u);
    }
}

// *flag |= 1 when the two word arrays differ anywhere (row-cache hits: the caller's row against the cached row's bytes)
__global__ void __launch_bounds__(256) k_words_differ(const uint4* __restrict__ a, const uint4* __restrict__ b, uint64_t n16,
                                                       uint32_t* __restrict__ flag) {
    uint32_t d = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint4 x = a[i], y = b[i];
        d |= (x.x ^ y.x) | (x.y ^ y.y) | (x.z ^ y.z) | (x.w ^ y.w);
    }
    if (__any(d != 0) && (threadIdx.x & 63) == 0) atomicOr(flag, 1u);
}
void launch_words_differ(hipStream_t s, const uint32_t* a, const uint32_t* b, uint64_t n_words, uint32_t* flag) {
    const uint64_t n16 = n_words / 4;   // rows are whole 32-byte elements
    if (!n16) return;
    const uint32_t blocks = (uint32_t)std::min<uint64_t>((n16 + 255) / 256, 2048);
    k_words_differ<<<blocks, 256, 0, s>>>(reinterpret_cast<const uint4*>(a), reinterpret_cast<const uint4*>(b), n16, flag);
}

void launch_poly_open(hipStream_t s, const uint32_t* f_mont, uint64_t n, uint32_t* alpha_mont, uint32_t* h,
                      uint32_t* hnext, uint32_t* y_mont, uint32_t* q_canon_or_null, const uint8_t* alpha_be32_host,
                      uint32_t* bad, uint8_t* y_be_or_null) {
    FrArg arg;
    memset(&arg, 0, sizeof(arg));
    if (alpha_be32_host) memcpy(arg.w, alpha_be32_host, 32);
    if (!n) {
        if (alpha_be32_host) launch_fr_from_host32(s, alpha_be32_host, alpha_mont, 1, bad);
        return;
    }
    // Level 0 folds 2^l0 coefficients per lane, every further level 16 values of the level below, until at most 2048
    // values are left for the single-workgroup scan; then the suffix values H are expanded back down level by
    // level.  Every serial loop is <= 16 long (each step is one dependent Fr product, ~1 us for a lone wave), and
    // all levels but the scan fill the GPU.  The level arrays are stacked in h / hnext (serve.hip / pipeline.hip size them for
    // (n+3)/4 * 3/2 + 64 entries; the levels above the first sum to < 1/3 of it).
    const int l0 = poly_lchunk(n);
    const int lup = 4;   // log2 chunk of the levels above the first (4-long chunks + more levels measured no faster on short rows)
    int lv_l[16], lv_sq[16];
    uint64_t lv_n[16], lv_off[16];
    int K = 1;
    lv_l[0] = l0; lv_sq[0] = 0; lv_n[0] = n; lv_off[0] = 0;           // level 0 = f itself (offset unused)
    lv_n[1] = (n + ((uint64_t)1 << l0) - 1) >> l0; lv_sq[1] = l0; lv_off[1] = 0;
    // long rows: the level-0 fold and the quotient with their coefficients staged through LDS (KZG_POLY_NO_LDS=1: the
    // strided forms, kept for the A/B and as the reference of test_poly_kernel_variants_agree)
    static const bool no_lds = getenv("KZG_POLY_NO_LDS") != nullptr;
    // from 2^21 coefficients: same-box A/Bs (profiles/r04_ab_opening_lds_staging.log, r04_ab_opening_lds_phases.log) of the
    // opening stage: 2^22 0.284 -> 0.236 ms, 2^21 0.196 -> 0.185, 2^20 0.160 -> 0.167 (one wave per SIMD there: the LDS hop
    // is pure latency).  KZG_POLY_LDS_MIN_LOG moves the threshold.
    static const int lds_min_log = getenv("KZG_POLY_LDS_MIN_LOG") ? atoi(getenv("KZG_POLY_LDS_MIN_LOG")) : 21;
    const bool lds = !no_lds && l0 == 4 && (n & 1023) == 0 && n >= ((uint64_t)1 << lds_min_log);
    // (the level-0 fold gains nothing from LDS staging: 53 against 49 us, profiles/r04_ab_opening_lds_staging.log -- strided)
    if (alpha_be32_host)
        k_poly_chunk_eval<true><<<nblk(lv_n[1], 256), 256, 0, s>>>(f_mont, n, l0, alpha_mont, 0, h, arg, alpha_mont, bad);
    else
        k_poly_chunk_eval<false><<<nblk(lv_n[1], 256), 256, 0, s>>>(f_mont, n, l0, alpha_mont, 0, h, arg, nullptr, nullptr);
    while (lv_n[K] > 2048 && K < 14) {
        lv_l[K] = lup;
        lv_n[K + 1] = (lv_n[K] + ((uint64_t)1 << lup) - 1) >> lup;
        lv_sq[K + 1] = lv_sq[K] + lup;
        lv_off[K + 1] = lv_off[K] + lv_n[K];
        k_poly_chunk_eval<false><<<nblk(lv_n[K + 1], 256), 256, 0, s>>>(h + 8 * lv_off[K], lv_n[K], lup, alpha_mont, lv_sq[K],
                                                                        h + 8 * lv_off[K + 1], arg, nullptr, nullptr);
        K++;
    }
    // the scan is one workgroup of dependent Fr products: 256 lanes (one wave per SIMD, <= 8 values each) run the chain
    // at a lone wave's issue rate; 1024 lanes (four waves per SIMD) only when there is more than that to fold
    if (lv_n[K] <= 1024)
        k_poly_chunk_scan<256><<<1, 256, 0, s>>>(h + 8 * lv_off[K], lv_n[K], lv_sq[K], alpha_mont, hnext + 8 * lv_off[K], y_mont,
                                                 y_be_or_null);
    else
        k_poly_chunk_scan<1024><<<1, 1024, 0, s>>>(h + 8 * lv_off[K], lv_n[K], lv_sq[K], alpha_mont, hnext + 8 * lv_off[K], y_mont,
                                                   y_be_or_null);
    for (int k = K - 1; k >= 1; k--)
        k_poly_chunk_expand<<<nblk(lv_n[k + 1], 256), 256, 0, s>>>(h + 8 * lv_off[k], lv_n[k], lv_l[k], alpha_mont,
                                                                   lv_sq[k], hnext + 8 * lv_off[k + 1],
                                                                   hnext + 8 * lv_off[k]);
    if (q_canon_or_null) {
        if (lds) k_poly_quotient16_lds<<<(uint32_t)(n >> 10), 64, 0, s>>>(f_mont, n, alpha_mont, hnext, q_canon_or_null);
        else k_poly_quotient<<<nblk(lv_n[1], 256), 256, 0, s>>>(f_mont, n, l0, alpha_mont, hnext, q_canon_or_null);
    }
}

