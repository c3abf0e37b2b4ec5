// Fr-side kernel launchers (fr_ntt.hip: codec + NTT; fr_poly.hip: opening): wire codec, NTT, opening evaluation + quotient.
#pragma once
#include "g1.hip.h"

// one scalar as a kernel argument: the 32 big-endian bytes, as they lie in memory
struct FrArg {
    uint32_t w[8];
};
void launch_fr_from_be(hipStream_t s, const uint8_t* be, uint32_t* out, uint64_t n, int to_mont, uint32_t* bad);
// one scalar from HOST memory, handed over as a kernel argument (no copy on the stream)
void launch_fr_from_host32(hipStream_t s, const uint8_t be32[32], uint32_t* out, int to_mont, uint32_t* bad);
void launch_fr_to_be(hipStream_t s, const uint32_t* in, uint8_t* be, uint64_t n, int from_mont);
void launch_fr_from_mont(hipStream_t s, const uint32_t* in, uint32_t* out, uint64_t n);
// tw: 2^(log_n-1) Montgomery-form powers of w_n (inverse: of w_n^-1), 12 words each (nine limbs + padding)
void launch_fr_twiddles(hipStream_t s, uint32_t* tw, int log_n, int inverse);
void launch_fr_inv_pow2(hipStream_t s, uint32_t* out, int log_n);
// out-of-place natural-order NTT of 2^log_n Montgomery-form elements; scale (1/n, Montgomery) applied if given
// tw: the table launch_fr_twiddles fills (n/2 twiddles, nine 29-bit limbs in a 48-byte slot each).  mid: 12 * n words of
// scratch (the vector between passes, same slot format); unused when log_n <= 8 (one pass)
void launch_fr_ntt(hipStream_t s, const uint32_t* in, uint32_t* out, int log_n, const uint32_t* tw,
                   const uint32_t* scale_or_null, uint32_t* mid);
// y = f(alpha) (Montgomery) and, if q is given, the n-1 canonical coefficients of (f - y)/(X - alpha)
// h, hnext: ceil(n/4) * 3/2 + 64 Fr scratch each (level arrays stacked; the chunk length shrinks to 4 for small rows)
// alpha_be32_host given: alpha arrives as 32 big-endian HOST bytes (a kernel argument of the first kernel), its
// Montgomery form is left at alpha_mont, *bad is raised if it is >= r; otherwise alpha_mont is read.
// y_be_or_null: also write y as 32 big-endian bytes (device pointer)
void launch_poly_open(hipStream_t s, const uint32_t* f_mont, uint64_t n, uint32_t* alpha_mont, uint32_t* h,
                      uint32_t* hnext, uint32_t* y_mont, uint32_t* q_canon_or_null,
                      const uint8_t* alpha_be32_host = nullptr, uint32_t* bad = nullptr, uint8_t* y_be_or_null = nullptr);
// *flag |= 1 when a[0, n_words) and b[0, n_words) differ (n_words a multiple of 4): verification of a row-cache hit
void launch_words_differ(hipStream_t s, const uint32_t* a, const uint32_t* b, uint64_t n_words, uint32_t* flag);
