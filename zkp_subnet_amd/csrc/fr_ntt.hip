// fr_ntt.hip -- Fr-side kernels, part 1: the wire codec (32-byte big-endian <-> 8 x u32, canonical check) and the radix-2
// Cooley-Tukey NTT with LDS-staged butterflies.  Replaces what the external prover does behind fft(poly, left, inverse) and
// the implicit IFFT of worker_commit / worker_open (reference neurons/validator.py:59-65; neurons/miner.py:39,48).
// Domain convention: w_n = 7^((r-1)/n), natural order in and out, inverse carries 1/n.
#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "fr_kernels.hip.h"

static inline uint32_t nblk(uint64_t n, uint32_t b) { return (uint32_t)((n + b - 1) / b); }

// All arithmetic below is fr29.hip.h's: 9 x 29-bit limbs, Montgomery radix 2^261; memory holds 8-word values.
// "Montgomery form" in this file (rows, coefficients, twiddles, 1/n) therefore means a * 2^261 mod r, canonical.
KZG_DEV void words_from_be(uint32_t* w, const uint8_t* be) { limbs_from_be<8>(w, be); }

// ------------------------------------------------------------------------------------------------ codec
__global__ void __launch_bounds__(256) k_fr_from_be(const uint8_t* __restrict__ be, uint32_t* __restrict__ out,
                                                     uint64_t n, int to_mont, uint32_t* __restrict__ bad) {
    uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    uint32_t w[8];
    words_from_be(w, be + 32 * j);
    if (fr_words_ge_r(w)) atomicOr(bad, 1u);  // non-canonical scalar: the call fails (SURVEY 8b errors)
    fr9_t v;
    fr9_from_words(v, w);
    if (to_mont) fr9_to_mont(v, v);
    fr9_store(out + 8 * j, v);
}
// (FrArg -- a scalar handed over as a kernel argument -- is declared in fr_kernels.hip.h: the opening uses it too)
__global__ void __launch_bounds__(64) k_fr_from_arg(const FrArg a, uint32_t* __restrict__ out, int to_mont,
                                                     uint32_t* __restrict__ bad) {
    if (threadIdx.x || blockIdx.x) return;
    uint32_t w[8];
#pragma unroll
    for (int i = 0; i < 8; i++) w[i] = bswap32(a.w[7 - i]);
    if (fr_words_ge_r(w)) atomicOr(bad, 1u);
    fr9_t v;
    fr9_from_words(v, w);
    if (to_mont) fr9_to_mont(v, v);
    fr9_store(out, v);
}
__global__ void __launch_bounds__(256) k_fr_to_be(const uint32_t* __restrict__ in, uint8_t* __restrict__ be,
                                                   uint64_t n, int from_mont) {
    uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    fr9_t v;
    fr9_load(v, in + 8 * j);
    if (from_mont) fr9_from_mont(v, v);
    uint32_t w[8];
    fr9_to_words(w, v);
    limbs_to_be<8>(be + 32 * j, w);
}
__global__ void __launch_bounds__(256) k_fr_from_mont(const uint32_t* __restrict__ in, uint32_t* __restrict__ out,
                                                       uint64_t n) {
    uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    fr9_t v;
    fr9_load(v, in + 8 * j);
    fr9_from_mont(v, v);
    fr9_store(out + 8 * j, v);
}

// ------------------------------------------------------------------------------------------------ twiddles
// w_{2^32} = 7^((r-1)/2^32) and its inverse, canonical limbs (re-derived in oracle/bls12_381.py)
KZG_DEV void fr_root_2_32(fr9_t& w, int inverse) {
    constexpr uint32_t W[8] = {0x439f0d2bu, 0x3829971fu, 0x8c2280b9u, 0xb6368350u,
                               0x22c813b4u, 0xd09b6819u, 0xdfe81f20u, 0x16a2a19eu};
    constexpr uint32_t WI[8] = {0x3cf19a78u, 0x0fb4d6e1u, 0xb566f833u, 0x6f67d4a2u,
                                0xa35d0168u, 0xed4f2f74u, 0x6e19c653u, 0x0538a6f6u};
    uint32_t t[8];
#pragma unroll
    for (int i = 0; i < 8; i++) t[i] = inverse ? WI[i] : W[i];
    fr9_from_words(w, t);
    fr9_to_mont(w, w);
}
// tw = w_n^(+-k), k < n/2, Montgomery form, canonical, as the NINE 29-bit limbs the butterflies multiply by, in a 48-byte
// slot each: a butterfly reads its twiddle with three 16-byte loads and no conversion.
// 64 consecutive k per lane
__global__ void __launch_bounds__(256) k_fr_twiddles(uint32_t* __restrict__ tw, int log_n, int inverse) {
    const uint64_t half = (uint64_t)1 << (log_n - 1);
    uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t k0 = t * 64;
    if (k0 >= half) return;
    fr9_t w, cur, pw;
    fr_root_2_32(w, inverse);
    for (int i = log_n; i < 32; i++) { fr9_mul(w, w, w); }       // N class in, N class out
    fr9_canon(w, w);
    fr9_one(cur);
    pw = w;
    for (uint64_t e = k0; e; e >>= 1) {
        if (e & 1) fr9_mul(cur, cur, pw);
        fr9_mul(pw, pw, pw);
    }
    for (uint64_t k = k0; k < k0 + 64 && k < half; k++) {
        fr9_t c;
        fr9_canon(c, cur);
        {
            uint4* q = reinterpret_cast<uint4*>(tw + 12 * k);
            q[0] = make_uint4(c.l[0], c.l[1], c.l[2], c.l[3]);
            q[1] = make_uint4(c.l[4], c.l[5], c.l[6], c.l[7]);
            q[2] = make_uint4(c.l[8], 0u, 0u, 0u);
        }
        fr9_mul(cur, cur, w);
    }
}
// out[0] = 2^-log_n in Montgomery form
__global__ void k_fr_inv_pow2(uint32_t* __restrict__ out, int log_n) {
    if (threadIdx.x || blockIdx.x) return;
    constexpr uint32_t HALF[8] = {0x80000001u, 0x7fffffffu, 0x7fff2dffu, 0xa9ded201u,
                                  0x04d0ec02u, 0x199cec04u, 0x94cebea4u, 0x39f6d3a9u};  // (r+1)/2
    uint32_t t[8];
#pragma unroll
    for (int i = 0; i < 8; i++) t[i] = HALF[i];
    fr9_t h, acc;
    fr9_from_words(h, t);
    fr9_to_mont(h, h);
    fr9_one(acc);
    for (int i = 0; i < log_n; i++) fr9_mul(acc, acc, h);
    fr9_canon(acc, acc);
    fr9_store(out, acc);
}

// ------------------------------------------------------------------------------------------------ NTT
// Radix-2 DIT network (bit-reversed input order -> natural output), run as a few PASSES of up to 8 stages; a pass
// stages a tile of 2^S "rows" x C "columns" (at most 1024 elements = 32 KB) in LDS, so every pass costs one HBM read
// and one HBM write of the vector: 3 passes at 2^22 (8+7+7) instead of bit reversal + 1 LDS pass + 12 global stages.
//   * first pass (stages 0..S-1): the bit reversal is folded into its load.  Output tile t (2^S contiguous elements)
//     is in[brev_S(r) * NT + brev(t)], NT = n / 2^S tiles; a workgroup takes the C tiles whose brev(t) are adjacent,
//     so that each source row is C contiguous elements (C x 32 B = 128 B), runs C independent 2^S-point networks and
//     writes C contiguous tiles.
//   * later passes (stages s0..s0+S-1), in place: row r of a tile = bits [s0, s0+S) of the index, the C columns are
//     adjacent values of the low bits, the remaining high bits are fixed: again C x 32 B contiguous per row.
// Stage s pairs i and i + 2^s with twiddle w_n^((i mod 2^s) << (log_n - s - 1)) from the n/2-entry table.  The last
// pass also applies the 1/n of the inverse transform.
#define NTT_PASS_LOG 8      // stages per pass (tile rows 2^S <= 256)
#define NTT_TILE_ELEMS 1024 // rows x columns
KZG_DEV uint32_t brev_bits(uint32_t v, int bits) { return bits ? (__brev(v) >> (32 - bits)) : 0u; }

// LDS tile: limb-major (9 arrays of 1024 words), values kept as normalised 9-limb residues between stages; they are
// only canonicalised (one product by R mod r, or by the 1/n of the inverse transform) when the pass stores to HBM.
struct NttTile {
    uint32_t l[9][NTT_TILE_ELEMS];
};
KZG_DEV void tile_get(fr9_t& v, const NttTile& sm, uint32_t e) {
#pragma unroll
    for (int i = 0; i < 9; i++) v.l[i] = sm.l[i][e];
}
KZG_DEV void tile_put(NttTile& sm, uint32_t e, const fr9_t& v) {
#pragma unroll
    for (int i = 0; i < 9; i++) sm.l[i][e] = v.l[i];
}
// Between the passes of a transform the vector is kept in `mid`, values partially reduced (< 2r, fr9_reduce_approx) --
// round 2: the nine 29-bit limbs in a 48-byte slot; round 3: packed into 8 words (below): a pass that is not the last stores without
// the canonicalising product (27 % of the products of a three-pass transform), a pass that is not the first loads without
// unpacking.  LAST: this pass writes `out` (8-word canonical elements, 1/n applied).  The first pass reads `in`.
// nine limbs in a 48-byte slot (three 16-byte accesses; element-major, so a tile row stays C x 48 contiguous bytes)
KZG_DEV void limbs12_get(fr9_t& v, const uint32_t* __restrict__ base, uint64_t e) {
    const uint4* q = reinterpret_cast<const uint4*>(base + 12 * e);
    const uint4 a = q[0], b = q[1], c = q[2];
    v.l[0] = a.x; v.l[1] = a.y; v.l[2] = a.z; v.l[3] = a.w;
    v.l[4] = b.x; v.l[5] = b.y; v.l[6] = b.z; v.l[7] = b.w;
    v.l[8] = c.x;
}
KZG_DEV void limbs12_put(uint32_t* __restrict__ base, uint64_t e, const fr9_t& v) {
    uint4* q = reinterpret_cast<uint4*>(base + 12 * e);
    q[0] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
    q[1] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
    q[2] = make_uint4(v.l[8], 0u, 0u, 0u);
}
#ifndef KZG_NTT_MID48
// Round 3: the partially reduced value (< 2r < 2^256) is PACKED into 8 words between passes -- 2/3 of the bytes, at the
// price of a pack and an unpack (~20 instructions each) per element and pass boundary.  Same-box A/B against the 48-byte
// slots of round 2 (profiles/r03_ab_ntt_mid32.log): 2^22 0.580 / 0.572 / 0.580 -> 0.547 / 0.542 / 0.557 ms, 2^20 0.158 /
// 0.162 / 0.159 -> 0.153 / 0.151 / 0.152: -5 %.  The transform is partly bound by its HBM round trips, and
// instructions are cheaper than bytes here.  -DKZG_NTT_MID48 restores the slot form (the scratch is sized for it either way).
KZG_DEV void mid_get(fr9_t& v, const uint32_t* __restrict__ mid, uint64_t, uint64_t e) { fr9_load(v, mid + 8 * e); }
KZG_DEV void mid_put(uint32_t* __restrict__ mid, uint64_t, uint64_t e, const fr9_t& v_norm) {
    fr9_t t;
    fr9_reduce_approx(t, v_norm);
    fr9_store(mid + 8 * e, t);
}
#else
KZG_DEV void mid_get(fr9_t& v, const uint32_t* __restrict__ mid, uint64_t, uint64_t e) { limbs12_get(v, mid, e); }
KZG_DEV void mid_put(uint32_t* __restrict__ mid, uint64_t, uint64_t e, const fr9_t& v_norm) {
    fr9_t t;
    fr9_reduce_approx(t, v_norm);
    limbs12_put(mid, e, t);
}
#endif
template <uint32_t NT_, bool LAST>
__global__ void __launch_bounds__(NT_) k_fr_ntt_pass(const uint32_t* __restrict__ in, uint32_t* __restrict__ out,
                                                      uint32_t* __restrict__ mid, int log_n, int s0, int S, int logC,
                                                      const uint32_t* __restrict__ tw,
                                                      const uint32_t* __restrict__ scale_or_null) {
    __shared__ NttTile sm;
    const uint32_t R = 1u << S, C = 1u << logC, E = R << logC;
    const bool first = s0 == 0;
    const uint64_t n = (uint64_t)1 << log_n;
    const int log_nt = log_n - S;  // first pass: tiles
    // ---- load
    if (first) {
        const uint64_t NT = (uint64_t)1 << log_nt;
        const uint64_t u = blockIdx.x;
        for (uint32_t e = threadIdx.x; e < E; e += NT_) {
            const uint32_t rp = e >> logC, c = e & (C - 1);  // source row, tile
            fr9_t v;
            fr9_load(v, in + 8 * ((uint64_t)rp * NT + u * C + c));
            tile_put(sm, (c << S) + brev_bits(rp, S), v);    // LDS layout [tile][row]
        }
    } else {
        const uint64_t groups = ((uint64_t)1 << s0) >> logC;  // column groups per block of 2^(s0+S) elements
        const uint64_t h = blockIdx.x / groups, cg = blockIdx.x - h * groups;
        const uint64_t base = (h << (s0 + S)) + (cg << logC);
        for (uint32_t e = threadIdx.x; e < E; e += NT_) {
            const uint32_t r = e >> logC, c = e & (C - 1);
            fr9_t v;
            mid_get(v, mid, n, base + ((uint64_t)r << s0) + c);
            tile_put(sm, e, v);                              // LDS layout [row][column]
        }
    }
    __syncthreads();
    // ---- butterflies: t = v w (N class), u' = u + t, v' = u + 4r - t, both renormalised.  A value grows by at most 4r
    // per stage: < 33r after the 8 stages of a pass, inside what fr9_mul accepts as its first operand (< 64r).
    const uint64_t col0 = first ? 0 : ((uint64_t)(blockIdx.x % (((uint64_t)1 << s0) >> logC)) << logC);
    for (int l = 0; l < S; l++) {
        const uint32_t half = 1u << l;
        const int s = s0 + l;
        const bool renorm = ((S - 1 - l) & 1) == 0;
        for (uint32_t b = threadIdx.x; b < E / 2; b += NT_) {
            uint32_t ei, ej;
            uint64_t k;
            if (first) {
                const uint32_t c = b >> (S - 1), bb = b & ((R >> 1) - 1);
                const uint32_t kk = bb & (half - 1);
                const uint32_t i = ((bb >> l) << (l + 1)) + kk;
                ei = (c << S) + i;
                ej = ei + half;
                k = kk;
            } else {
                const uint32_t c = b & (C - 1), bb = b >> logC;
                const uint32_t kk = bb & (half - 1);
                const uint32_t i = ((bb >> l) << (l + 1)) + kk;
                ei = (i << logC) + c;
                ej = ei + (half << logC);
                k = ((uint64_t)kk << s0) + col0 + c;
            }
            fr9_t u, v, w, t;
            tile_get(u, sm, ei);
            tile_get(v, sm, ej);
            if (first && l == 0) {
                t = v;          // stage 0: every twiddle is w^0 = 1 and v is a canonical input -- no product (1/log2(n) of them all)
            } else {
                limbs12_get(w, tw, k << (log_n - s - 1));
                fr9_mul(t, v, w);
            }
            fr9_add(v, u, t);
            fr9_sub4(w, u, t);
            if (renorm) {   // every second stage, and always the last: in between the limbs stay below 2^31 (header of fr29.hip.h)
                fr9_norm(v, v);
                fr9_norm(w, w);
            }
            tile_put(sm, ei, v);
            tile_put(sm, ej, w);
        }
        __syncthreads();
    }
    // ---- store.  Last pass: canonical 8-word elements; the same product applies the 1/n of an inverse transform.
    // Earlier passes: the nine limbs, partially reduced, to `mid`.
    fr9_t f;
    fr9_zero(f);
    if constexpr (LAST) {
        if (scale_or_null) fr9_load(f, scale_or_null);
    }
    if (first) {
        const uint32_t u = blockIdx.x;
        const uint64_t tiles_per_c = ((uint64_t)1 << log_nt) >> logC;
        const uint64_t t_low = brev_bits(u, log_nt - logC);
        for (uint32_t e = threadIdx.x; e < E; e += NT_) {
            const uint32_t c = e >> S, r = e & (R - 1);
            const uint64_t t = (uint64_t)brev_bits(c, logC) * tiles_per_c + t_low;
            fr9_t v;
            tile_get(v, sm, e);
            if constexpr (LAST) {
                if (scale_or_null) { fr9_mul(v, v, f); fr9_canon(v, v); }
                else fr9_reduce(v, v);          // forward transform: nothing to multiply by
                fr9_store(out + 8 * ((t << S) + r), v);
            } else {
                mid_put(mid, n, (t << S) + r, v);
            }
        }
    } else {
        const uint64_t groups = ((uint64_t)1 << s0) >> logC;
        const uint64_t h = blockIdx.x / groups, cg = blockIdx.x - h * groups;
        const uint64_t base = (h << (s0 + S)) + (cg << logC);
        for (uint32_t e = threadIdx.x; e < E; e += NT_) {
            const uint32_t r = e >> logC, c = e & (C - 1);
            fr9_t v;
            tile_get(v, sm, e);
            if constexpr (LAST) {
                if (scale_or_null) { fr9_mul(v, v, f); fr9_canon(v, v); }
                else fr9_reduce(v, v);
                fr9_store(out + 8 * (base + ((uint64_t)r << s0) + c), v);
            } else {
                mid_put(mid, n, base + ((uint64_t)r << s0) + c, v);
            }
        }
    }
}

// ---- the register-blocked form: TWO stages per LDS round trip.  A lane takes the four elements i, i+h, i+2h, i+3h
// (h = 2^l) through stage l -- (x0, x1) and (x2, x3), both with twiddle A = w^(k) of that stage -- and stage l+1 --
// (y0, y2) with B0 and (y1, y3) with B1 -- without leaving its registers: 8 LDS reads + 8 writes of a limb per butterfly
// pair instead of 16 + 16 (the radix-2 kernel moved 36 LDS words per butterfly, this one 18), half the barriers, one index
// computation per four butterflies.  The multiplication count is that of radix 2 (a prime field has no cheap w_4): A, B0
// and B1 are three loads from the same table -- with table index t for A, B0 sits at t/2 and B1 at t/2 + n/4.  Tiles hold 2048
// elements (72 KB of LDS, 512 lanes, two workgroups per CU): up to 9 stages x >= 4 columns, i.e. rows of >= 192 contiguous
// bytes in HBM (an 11 + 11 split of 2^22 would need single-element rows: 48-byte gathers at a 98-KB stride).  Lazy bounds: inputs normalised (limbs < 2^29); after stage l limbs < 2^29 + 2^30, still a legal
// first operand of the product; after stage l+1 limbs < 2^29 + 2^31 fit a word and are normalised before they go back to
// LDS; values grow <= 4r per stage, < 2r + 36r over the 9 stages of a pass (< 64r).  An odd stage count starts with one radix-2 stage.
#define NTT4_PASS_LOG 9      // stages per pass: 2^9 rows x >= 4 columns, so that a tile row is >= 192 contiguous bytes of HBM
#define NTT4_TILE_ELEMS 2048
template <uint32_t ELEMS>
struct NttTile4 {            // four elements per lane: a 64-lane workgroup stages 256 elements (9 KB), a 512-lane one 2048 (72 KB)
    uint32_t l[9][ELEMS];
};
template <class TILE>
KZG_DEV void tile4_get(fr9_t& v, const TILE& sm, uint32_t e) {
#pragma unroll
    for (int i = 0; i < 9; i++) v.l[i] = sm.l[i][e];
}
template <class TILE>
KZG_DEV void tile4_put(TILE& sm, uint32_t e, const fr9_t& v) {
#pragma unroll
    for (int i = 0; i < 9; i++) sm.l[i][e] = v.l[i];
}
KZG_DEV void bfly(fr9_t& u, fr9_t& v, const fr9_t& w) {     // (u, v) <- (u + v w, u + 4r - v w), lazy
    fr9_t t;
    fr9_mul(t, v, w);
    fr9_sub4(v, u, t);
    fr9_add(u, u, t);
}
// A/B knob: -DKZG_NTT4_OCC4 asks for four waves per SIMD (<= 128 VGPRs; the non-first passes then spill 16 dwords)
#ifdef KZG_NTT4_OCC4
#define NTT4_BOUNDS(N) __launch_bounds__(N, 1024 / (N))
#else
#define NTT4_BOUNDS(N) __launch_bounds__(N)
#endif
template <uint32_t NT_, bool LAST>
__global__ void NTT4_BOUNDS(NT_) k_fr_ntt_pass4(const uint32_t* __restrict__ in, uint32_t* __restrict__ out,
                                                       uint32_t* __restrict__ mid, int log_n, int s0, int S, int logC,
                                                       const uint32_t* __restrict__ tw,
                                                       const uint32_t* __restrict__ scale_or_null) {
    __shared__ NttTile4<4 * NT_> sm;     // the launcher never gives a workgroup more than 4 elements per lane
    const uint32_t R = 1u << S, C = 1u << logC, E = R << logC;
    const bool first = s0 == 0;
    const uint64_t n = (uint64_t)1 << log_n;
    const int log_nt = log_n - S;
    // ---- load (as k_fr_ntt_pass: bit reversal folded into the first pass, 48-byte slots between passes)
    // The (at most four) elements of a lane are REQUESTED TOGETHER and unpacked afterwards: a rolled loop waits for one HBM
    // round trip per element -- four latencies in a row at the head of every workgroup's life, which the two other
    // workgroups of the CU cover only partly (28 % of a wave's resident cycles were waits).
    uint64_t base = 0, col0 = 0;
#if !defined(KZG_NTT_ROLLED_LOADS) && !defined(KZG_NTT_MID48)
    {
        const uint32_t* src = first ? in : mid;
        uint64_t idx[4];
        uint32_t dst[4];
        if (first) {
            const uint64_t NT = (uint64_t)1 << log_nt;
            const uint64_t u = blockIdx.x;
#pragma unroll
            for (int it = 0; it < 4; it++) {
                const uint32_t e = threadIdx.x + (uint32_t)it * NT_;
                const uint32_t rp = e >> logC, c = e & (C - 1);
                idx[it] = (uint64_t)rp * NT + u * C + c;
                dst[it] = (c << S) + brev_bits(rp, S);        // LDS layout [tile][row]
            }
        } else {
            const uint64_t groups = ((uint64_t)1 << s0) >> logC;
            const uint64_t h = blockIdx.x / groups, cg = blockIdx.x - h * groups;
            base = (h << (s0 + S)) + (cg << logC);
            col0 = cg << logC;
#pragma unroll
            for (int it = 0; it < 4; it++) {
                const uint32_t e = threadIdx.x + (uint32_t)it * NT_;
                const uint32_t r = e >> logC, c = e & (C - 1);
                idx[it] = base + ((uint64_t)r << s0) + c;
                dst[it] = e;                                  // LDS layout [row][column]
            }
        }
        uint4 lo[4], hi[4];
#pragma unroll
        for (int it = 0; it < 4; it++) {
            if (threadIdx.x + (uint32_t)it * NT_ < E) {
                const uint4* q = reinterpret_cast<const uint4*>(src + 8 * idx[it]);
                lo[it] = q[0];
                hi[it] = q[1];
            }
        }
#pragma unroll
        for (int it = 0; it < 4; it++) {
            if (threadIdx.x + (uint32_t)it * NT_ < E) {
                const uint32_t w[8] = {lo[it].x, lo[it].y, lo[it].z, lo[it].w, hi[it].x, hi[it].y, hi[it].z, hi[it].w};
                fr9_t v;
                fr9_from_words(v, w);
                tile4_put(sm, dst[it], v);
            }
        }
    }
#else
    if (first) {
        const uint64_t NT = (uint64_t)1 << log_nt;
        const uint64_t u = blockIdx.x;
        for (uint32_t e = threadIdx.x; e < E; e += NT_) {
            const uint32_t rp = e >> logC, c = e & (C - 1);
            fr9_t v;
            fr9_load(v, in + 8 * ((uint64_t)rp * NT + u * C + c));
            tile4_put(sm, (c << S) + brev_bits(rp, S), v);    // LDS layout [tile][row]
        }
    } else {
        const uint64_t groups = ((uint64_t)1 << s0) >> logC;
        const uint64_t h = blockIdx.x / groups, cg = blockIdx.x - h * groups;
        base = (h << (s0 + S)) + (cg << logC);
        col0 = cg << logC;
        for (uint32_t e = threadIdx.x; e < E; e += NT_) {
            const uint32_t r = e >> logC, c = e & (C - 1);
            fr9_t v;
            mid_get(v, mid, n, base + ((uint64_t)r << s0) + c);
            tile4_put(sm, e, v);                              // LDS layout [row][column]
        }
    }
#endif
    __syncthreads();
    int l = 0;
    if (S & 1) {      // one radix-2 stage first
        const int s = s0;
        for (uint32_t b = threadIdx.x; b < E / 2; b += NT_) {
            uint32_t ei, ej;
            uint64_t k;
            if (first) {
                const uint32_t c = b >> (S - 1), bb = b & ((R >> 1) - 1);
                ei = (c << S) + (bb << 1);
                ej = ei + 1;
                k = 0;
            } else {
                const uint32_t c = b & (C - 1), bb = b >> logC;
                ei = ((bb << 1) << logC) + c;
                ej = ei + C;
                k = col0 + c;
            }
            fr9_t u, v, w;
            tile4_get(u, sm, ei);
            tile4_get(v, sm, ej);
            if (first) {        // stage 0: w^0 = 1 everywhere, v canonical: the butterfly without its product
                w = v;
                fr9_sub4(v, u, w);
                fr9_add(u, u, w);
            } else {
                limbs12_get(w, tw, k << (log_n - s - 1));
                bfly(u, v, w);
            }
            fr9_norm(u, u);
            fr9_norm(v, v);
            tile4_put(sm, ei, u);
            tile4_put(sm, ej, v);
        }
        __syncthreads();
        l = 1;
    }
    for (; l < S; l += 2) {
        const uint32_t half = 1u << l;
        const int s = s0 + l;
        for (uint32_t b = threadIdx.x; b < E / 4; b += NT_) {
            uint32_t e0, st;
            uint64_t k;
            if (first) {
                const uint32_t c = b >> (S - 2), bb = b & ((R >> 2) - 1);
                const uint32_t kk = bb & (half - 1);
                e0 = (c << S) + ((bb >> l) << (l + 2)) + kk;
                st = half;
                k = kk;
            } else {
                const uint32_t c = b & (C - 1), bb = b >> logC;
                const uint32_t kk = bb & (half - 1);
                e0 = ((((bb >> l) << (l + 2)) + kk) << logC) + c;
                st = half << logC;
                k = ((uint64_t)kk << s0) + col0 + c;
            }
            const uint64_t ia = k << (log_n - s - 1), ib = ia >> 1;
            // all three twiddles are requested before anything else: ONE wait on global memory per four butterflies (loaded
            // one by one in front of their products, each product stalled on its own load: +12 % on the radix-2 kernel)
            fr9_t x0, x1, x2, x3, wa, wb0, wb1;
            limbs12_get(wa, tw, ia);
            limbs12_get(wb0, tw, ib);
            limbs12_get(wb1, tw, ib + (n >> 2));
            tile4_get(x1, sm, e0 + st);
            tile4_get(x3, sm, e0 + 3 * st);
            tile4_get(x0, sm, e0);
            tile4_get(x2, sm, e0 + 2 * st);
            if (first && l == 0) {           // stage 0 of the transform: twiddle 1, canonical inputs -- no products
                fr9_t t1 = x1, t3 = x3;
                fr9_sub4(x1, x0, t1); fr9_add(x0, x0, t1);
                fr9_sub4(x3, x2, t3); fr9_add(x2, x2, t3);
            } else {
                bfly(x0, x1, wa);            // stage l
                bfly(x2, x3, wa);
            }
            bfly(x0, x2, wb0);               // stage l + 1
            bfly(x1, x3, wb1);
            fr9_norm(x0, x0);
            fr9_norm(x1, x1);
            fr9_norm(x2, x2);
            fr9_norm(x3, x3);
            tile4_put(sm, e0, x0);
            tile4_put(sm, e0 + st, x1);
            tile4_put(sm, e0 + 2 * st, x2);
            tile4_put(sm, e0 + 3 * st, x3);
        }
        __syncthreads();
    }
    // ---- store
    fr9_t f;
    fr9_zero(f);
    if constexpr (LAST) {
        if (scale_or_null) fr9_load(f, scale_or_null);
    }
    if (first) {
        const uint32_t u = blockIdx.x;
        const uint64_t tiles_per_c = ((uint64_t)1 << log_nt) >> logC;
        const uint64_t t_low = brev_bits(u, log_nt - logC);
        for (uint32_t e = threadIdx.x; e < E; e += NT_) {
            const uint32_t c = e >> S, r = e & (R - 1);
            const uint64_t t = (uint64_t)brev_bits(c, logC) * tiles_per_c + t_low;
            fr9_t v;
            tile4_get(v, sm, e);
            if constexpr (LAST) {
                if (scale_or_null) { fr9_mul(v, v, f); fr9_canon(v, v); }
                else fr9_reduce(v, v);
                fr9_store(out + 8 * ((t << S) + r), v);
            } else {
                mid_put(mid, n, (t << S) + r, v);
            }
        }
    } else {
        for (uint32_t e = threadIdx.x; e < E; e += NT_) {
            const uint32_t r = e >> logC, c = e & (C - 1);
            fr9_t v;
            tile4_get(v, sm, e);
            if constexpr (LAST) {
                if (scale_or_null) { fr9_mul(v, v, f); fr9_canon(v, v); }
                else fr9_reduce(v, v);
                fr9_store(out + 8 * (base + ((uint64_t)r << s0) + c), v);
            } else {
                mid_put(mid, n, base + ((uint64_t)r << s0) + c, v);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ launchers
void launch_fr_from_be(hipStream_t s, const uint8_t* be, uint32_t* out, uint64_t n, int to_mont, uint32_t* bad) {
    if (n) k_fr_from_be<<<nblk(n, 256), 256, 0, s>>>(be, out, n, to_mont, bad);
}
void launch_fr_from_host32(hipStream_t s, const uint8_t be32[32], uint32_t* out, int to_mont, uint32_t* bad) {
    FrArg a;
    memcpy(a.w, be32, 32);
    k_fr_from_arg<<<1, 64, 0, s>>>(a, out, to_mont, bad);
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
static void launch_fr_ntt_radix2(hipStream_t s, const uint32_t* in, uint32_t* out, int log_n, const uint32_t* tw,
                                 const uint32_t* scale_or_null, uint32_t* mid) {
    const uint64_t n = (uint64_t)1 << log_n;
    // split log_n into ceil(log_n / 8) passes of near-equal depth (22 -> 8 + 7 + 7)
    const int passes = (log_n + NTT_PASS_LOG - 1) / NTT_PASS_LOG;
    int s0 = 0;
    for (int p = 0; p < passes; p++) {
        const int S = (log_n - s0 + (passes - p) - 1) / (passes - p);
        // columns: as many as the 1024-element tile allows, bounded by what exists (tiles / low-bit range)
        int logC = 10 - S;
        const int avail = p == 0 ? log_n - S : s0;
        if (logC > avail) logC = avail;
        if (p == 0 && logC > 2) logC = 2;  // first pass: 4 tiles (128 B source rows) keep the tiles' stores long
        const uint32_t blocks = (uint32_t)(n >> (S + logC));
        const bool last = p == passes - 1;
        const bool big = (1u << (S + logC)) >= 1024;
        if (big && last) k_fr_ntt_pass<512, true><<<blocks, 512, 0, s>>>(in, out, mid, log_n, s0, S, logC, tw, scale_or_null);
        else if (big) k_fr_ntt_pass<512, false><<<blocks, 512, 0, s>>>(in, out, mid, log_n, s0, S, logC, tw, nullptr);
        else if (last) k_fr_ntt_pass<256, true><<<blocks, 256, 0, s>>>(in, out, mid, log_n, s0, S, logC, tw, scale_or_null);
        else k_fr_ntt_pass<256, false><<<blocks, 256, 0, s>>>(in, out, mid, log_n, s0, S, logC, tw, nullptr);
        s0 += S;
    }
}
// KZG_NTT_RADIX2=1 selects the round-2 kernel (one stage per LDS round trip, 1024-element tiles) for same-box A/B runs
static int ntt_use_radix2() {
    static const int v = [] {
        const char* e = getenv("KZG_NTT_RADIX2");
        return (e && e[0] == '1') ? 1 : 0;
    }();
    return v;
}
void launch_fr_ntt(hipStream_t s, const uint32_t* in, uint32_t* out, int log_n, const uint32_t* tw,
                   const uint32_t* scale_or_null, uint32_t* mid) {
    const uint64_t n = (uint64_t)1 << log_n;
    if (log_n == 0) {  // length 1: the transform is the identity (1/1 = 1)
        (void)hipMemcpyAsync(out, in, 32, hipMemcpyDeviceToDevice, s);
        return;
    }
    // Same-box A/B of the two kernels (profiles/r03_ab_ntt_radix4.log; NTT time inside a commit+open, ms):
    //   size   radix-2 (1024-tile)   radix-2^2, 2048-element tiles   radix-2^2, 1024-element tiles
    //   2^22   0.554 / 0.561         0.580 / 0.585                   0.540 / 0.539
    //   2^20   0.156 / 0.156         0.157 / 0.156                   0.153 / 0.150
    //   2^18   0.059 / 0.060         0.076 / 0.076                   0.061 / 0.059
    //   2^16   0.042 / 0.042         0.051 / 0.050                   0.042 / 0.042
    //   2^12   0.028 / 0.027         0.038 / 0.038                   0.030 / 0.030
    // Halving the LDS traffic and the barriers buys 3 % at best: the kernel is bound by the products themselves (153 mads
    // + ~120 other instructions per butterfly whatever the blocking) and by the vector's three HBM round trips (1.07 GB at
    // 2^22, ~0.25 ms at the bandwidth such tiles reach), which only partly overlap.  Larger tiles lose (two workgroups per
    // CU instead of four hide less of the load / store phases).  So: the register-blocked kernel with 1024-element tiles
    // from 2^18 up, the radix-2 kernel (one butterfly per lane: more lanes for the latency-bound short rows) below.
    if (ntt_use_radix2() || log_n < 18) {
        launch_fr_ntt_radix2(s, in, out, log_n, tw, scale_or_null, mid);
        return;
    }
    static const int tile_log = [] {
        const char* e = getenv("KZG_NTT_TILE_LOG");
        const int v = e ? atoi(e) : 10;
        return v >= 8 && v <= 11 ? v : 10;
    }();
    const int max_s = tile_log - 2;
    const int passes = (log_n + max_s - 1) / max_s;
    int s0 = 0;
    for (int p = 0; p < passes; p++) {
        const int S = (log_n - s0 + (passes - p) - 1) / (passes - p);
        int logC = tile_log - S;
        const int avail = p == 0 ? log_n - S : s0;
        if (logC > avail) logC = avail;
        if (p == 0 && logC > 2) logC = 2;  // first pass: 4 tiles (128 B source rows) keep the tiles' stores long
        const uint32_t blocks = (uint32_t)(n >> (S + logC));
        const bool last = p == passes - 1;
        const uint32_t quads = (1u << (S + logC)) >> 2;     // one lane per four elements (the tile is sized 4 x lanes)
#define NTT4_LAUNCH(NT)                                                                                                      \
    do {                                                                                                                     \
        if (last) k_fr_ntt_pass4<NT, true><<<blocks, NT, 0, s>>>(in, out, mid, log_n, s0, S, logC, tw, scale_or_null);      \
        else k_fr_ntt_pass4<NT, false><<<blocks, NT, 0, s>>>(in, out, mid, log_n, s0, S, logC, tw, nullptr);               \
    } while (0)
        if (quads > 256) NTT4_LAUNCH(512);
        else if (quads > 128) NTT4_LAUNCH(256);
        else if (quads > 64) NTT4_LAUNCH(128);
        else NTT4_LAUNCH(64);
#undef NTT4_LAUNCH
        s0 += S;
    }
}
