/* kzg_mi355x_test.h -- test hooks of libkzg_mi355x.so: the SAME library, a separate declaration.
 *
 * Nothing here is part of the serving surface (include/kzg_mi355x.h): these entry points expose single field / group /
 * pairing operations and the host-side point encoder so that tests/test_gpu_*.py, tests/test_abi.py and
 * tests/test_verify.py can compare each of them with the oracle.  A binding of the prover never needs
 * this file.
 */
#ifndef KZG_MI355X_TEST_H
#define KZG_MI355X_TEST_H
#include "kzg_mi355x.h"
#ifdef __cplusplus
extern "C" {
#endif

/* ---- unit-op hooks for the parity tests (tests/test_gpu_units.py); not part of the serving surface */
int kzg_test_field(kzg_ctx* ctx, int field /*0 Fp,1 Fr*/, int op /*0 mul,1 add,2 sub,3 mul(plain C ref),4 sqr*/,
                   const uint8_t* a_be, const uint8_t* b_be, uint8_t* out_be, uint64_t n);
int kzg_test_g1(kzg_ctx* ctx, int op /*0 a+b mixed,1 2a+b full,2 2a,3 4a,4 a+20a+20b chain; lane-parallel forms: 5 2a+b,
                                         6 4a, 7 ten rounds r <- 2r+b from a*/, const uint8_t* a_be96,
                const uint8_t* b_be96, uint8_t* out_be96, uint64_t n);

/* the host encoder itself (no GPU): 4 x 14 limbs of 28 bits (X, Y, ZZ, ZZZ; lazy limbs < 2^32; residues with
 * R = 2^392) -> 48-byte compressed point / 192-byte partial record.  Test hooks. */
int kzg_host_xyzz_to_c48(const uint32_t xyzz_limbs28[56], uint8_t out48[48]);
int kzg_host_xyzz_pair_to_c48(const uint32_t a_limbs28[56], const uint32_t b_limbs28[56], uint8_t out_a48[48],
                              uint8_t out_b48[48]); /* two points, one shared inversion (commit + open) */
int kzg_host_xyzz_to_partial192(const uint32_t xyzz_limbs28[56], uint8_t out192[192]);

/* test hook for the timeout path of kzg_msm_sharded: the NEXT sharded MSM on this context first queues a kernel that
 * spins for `ms` milliseconds (bounded: at most 2000) on its lane, so that a 1-rank communicator can be made to overrun
 * kzg_comm_set_timeout without a dead peer. */
int kzg_test_comm_stall(kzg_ctx* ctx, int ms);
/* the same for the next `count` sharded MSMs (two calls in flight on two lanes: one times out, the other must not return a
 * result computed under the aborted communicator) */
int kzg_test_comm_stall_n(kzg_ctx* ctx, int ms, int count);

/* test hook: final_exp(miller(P, Q)) as 12 x 48 B in tower order (Fp12 = Fp6[w], Fp6 = Fp2[v], Fp2 = Fp[u]) */
int kzg_vk_pairing(const uint8_t p_be96[96], const uint8_t q_be192[192], uint8_t out_fp12[576]);

#ifdef __cplusplus
}
#endif
#endif /* KZG_MI355X_TEST_H */
