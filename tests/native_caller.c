/* native_caller.c -- the C-ABI of libkzg_mi355x.so driven from plain C (C99, gcc, the public header only): what a cgo /
 * FFI binding of the prover does, with no Python and no torch in the process.  Built and run by tests/test_abi.py (no
 * GPU: kzg_create must fail with a status code) and tests/test_gpu_multi.py (on the MI355X: every line it prints is
 * compared with the CPU oracle by the test).  Mirrors INTEGRATION.md section 3.
 *
 *   native_caller <log2 n> <tau, 64 hex digits> <scalars file: n x 32 bytes big-endian, canonical>
 *
 * One process = one rank: the 1-rank communicator exercises the library's own collective path (kzg_comm_unique_id ->
 * kzg_comm_init -> kzg_msm_sharded: partial, ncclAllGather on the lane's stream, sum) end to end; with N ranks the only
 * difference is how the 128-byte id reaches the other processes.
 * exit codes: 0 ok, 2 usage / input, 3 no usable device (KZG_E_HIP from kzg_create), 4 a call failed, 5 results disagree */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "kzg_mi355x.h"

static void hex(const char* name, const uint8_t* p, int n) {
    printf("%s ", name);
    for (int i = 0; i < n; i++) printf("%02x", p[i]);
    printf("\n");
}
static int from_hex(const char* s, uint8_t* out, int n) {
    if ((int)strlen(s) != 2 * n) return -1;
    for (int i = 0; i < n; i++) {
        unsigned v;
        if (sscanf(s + 2 * i, "%2x", &v) != 1) return -1;
        out[i] = (uint8_t)v;
    }
    return 0;
}
#define CHECK(call)                                                                         \
    do {                                                                                    \
        int rc_ = (call);                                                                   \
        if (rc_ != KZG_OK) {                                                                \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, kzg_last_error(ctx));             \
            return 4;                                                                       \
        }                                                                                   \
    } while (0)

int main(int argc, char** argv) {
    if (argc != 4) {
        fprintf(stderr, "usage: native_caller <log2 n> <tau hex64> <scalars.bin>\n");
        return 2;
    }
    const int lg = atoi(argv[1]);
    uint8_t tau[32], one[32] = {0};
    one[31] = 1;
    if (lg < 1 || lg > 20 || from_hex(argv[2], tau, 32)) return 2;
    const uint64_t n = (uint64_t)1 << lg;
    uint8_t* scal = (uint8_t*)malloc(32 * n);
    FILE* f = fopen(argv[3], "rb");
    if (!scal || !f || fread(scal, 32, n, f) != n) {
        fprintf(stderr, "cannot read %llu scalars from %s\n", (unsigned long long)n, argv[3]);
        return 2;
    }
    fclose(f);
    printf("version %s\n", kzg_version());

    kzg_ctx* ctx = NULL;
    int rc = kzg_create(0, &ctx);
    if (rc != KZG_OK) {      /* no gfx950 device: a status code and a message, never an abort (there is no CPU fallback) */
        fprintf(stderr, "kzg_create -> %d: %s\n", rc, kzg_last_error(NULL));
        return rc == KZG_E_HIP ? 3 : 4;
    }
    /* one worker slice of n points [tau^j] G (the synthetic stand-in for the setup file, include/kzg_mi355x.h kzg_gen_srs) */
    CHECK(kzg_gen_srs(ctx, tau, one, 1, lg, 0));
    uint8_t msm_host[48], msm_res[48], msm_shard[48], c48[48], ev32[32], p48[48];
    CHECK(kzg_msm(ctx, scal, n, 0, msm_host));                       /* scalars from host memory */
    CHECK(kzg_upload_fr(ctx, 0, scal, n, 0));
    CHECK(kzg_msm_resident(ctx, 0, n, 0, msm_res));                  /* the bench's entry point */
    /* the SRS-sharded MSM with the collective inside the library: this process is rank 0 of 1 */
    uint8_t id[128];
    int32_t info[4];
    CHECK(kzg_comm_unique_id(id));      /* N ranks: rank 0 draws it, the others receive the 128 bytes out of band */
    CHECK(kzg_comm_init(ctx, id, 0, 1));
    CHECK(kzg_comm_set_timeout(ctx, 60000));
    CHECK(kzg_comm_info(ctx, info));
    printf("comm rank %d world %d rccl %d broken %d\n", info[0], info[1], info[2], info[3]);
    for (int k = 0; k < 3; k++) CHECK(kzg_msm_sharded(ctx, 0, n, 0, msm_shard));
    CHECK(kzg_comm_destroy(ctx));
    /* the miner's request: commit + open of the same vector as an evaluation-form row, alpha = its second element */
    CHECK(kzg_commit_open(ctx, 0, scal, n, 1, scal + 32, c48, ev32, p48));
    hex("msm", msm_host, 48);
    hex("msm_resident", msm_res, 48);
    hex("msm_sharded", msm_shard, 48);
    hex("commitment", c48, 48);
    hex("eval", ev32, 32);
    hex("proof", p48, 48);
    /* errors are status codes with a message */
    rc = kzg_msm_sharded(ctx, 0, n, 0, msm_shard);
    printf("sharded_without_comm %d\n", rc);
    rc = kzg_commit(ctx, 7, scal, n, 1, c48);
    printf("bad_worker_index %d\n", rc);
    kzg_destroy(ctx);
    /* several GPUs behind one handle (here: the same GPU twice -- one context each): worker index i -> device i mod 2 */
    {
        const int devs[2] = {0, 0};
        kzg_multi* m = NULL;
        uint8_t s0[64] = {0}, mc[48];
        s0[31] = 1;                                   /* two worker slices with factors 1 and 1: both equal the single slice */
        s0[63] = 1;
        if (kzg_multi_create(2, devs, &m) != KZG_OK || kzg_multi_gen_srs(m, tau, s0, lg + 1, 1) != KZG_OK ||
            kzg_multi_commit(m, 1, scal, n, 1, mc) != KZG_OK) {
            fprintf(stderr, "kzg_multi_*: %s\n", kzg_multi_last_error(m));
            return 4;
        }
        printf("multi devices %d device_of_1 %d\n", kzg_multi_count(m), kzg_multi_device_of(m, 1));
        hex("multi_commitment", mc, 48);
        printf("multi_bad_index %d\n", kzg_multi_commit(m, 2, scal, n, 1, mc));
        kzg_multi_destroy(m);
    }
    free(scal);
    if (memcmp(msm_host, msm_res, 48) || memcmp(msm_host, msm_shard, 48)) return 5;
    return 0;
}
