/* kzg_mi355x.h -- C-ABI of libkzg_mi355x.so, the MI355X (gfx950) KZG segment prover.
 *
 * This is the drop-in boundary for the reference's prover seam: the reference miner talks to an external
 * prover process through `fourier.Client` (constructed at reference base/miner.py:73-84, used at
 * neurons/miner.py:39,48 and neurons/validator.py:59-104).  Each entry point below names the reference
 * interface it replaces.  The Python class zkp_subnet_amd.client.Client binds these through ctypes and
 * reproduces the Client method surface; INTEGRATION.md shows the binding a maintainer would add.
 *
 * Conventions
 *   - plain pointers and sizes only; the caller owns every buffer; the library copies in and out.
 *   - Fr  : 32 bytes big-endian, canonical (< r).  A non-canonical scalar fails the call (KZG_E_SCALAR).
 *   - G1  : affine x||y, 2 x 48 bytes big-endian (96 zero bytes = infinity), or the 48-byte ZCash
 *           compressed encoding for results.  Partial sums cross the ABI as 192 opaque bytes (XYZZ).
 *   - return 0 on success, negative kzg_status otherwise; kzg_last_error(ctx) gives the message.
 *     No exception or abort crosses the boundary.
 *   - one ctx = one GPU, and it is thread-safe: the reference's axon runs Miner.forward on worker threads
 *     (neurons/miner.py:106-135), so concurrent calls on one ctx each take one of four internal lanes (own HIP stream
 *     + workspace) and run concurrently on the GPU -- one request's sort and latency-bound tail hide under another's
 *     accumulate.  A fifth concurrent call waits for a lane.  (Re)loading the SRS and writing a resident slot are
 *     exclusive: they wait until the lanes are idle.  kzg_last_error reports the calling THREAD's last failure
 *     (ctypes releases the GIL during a call).
 *   - there is NO CPU fallback: every compute entry point fails with KZG_E_HIP when no gfx950 device works.
 */
#ifndef KZG_MI355X_H
#define KZG_MI355X_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct kzg_ctx kzg_ctx;

typedef enum {
    KZG_OK = 0,
    KZG_E_ARG = -1,     /* bad length / index / state */
    KZG_E_SCALAR = -2,  /* non-canonical Fr (>= r) */
    KZG_E_POINT = -3,   /* G1 input not reduced or not on the curve */
    KZG_E_HIP = -4,     /* HIP runtime failure or no usable device */
    KZG_E_NOMEM = -5,
    KZG_E_BUSY = -6,    /* an MSM ticket is outstanding (see kzg_msm_submit) */
    KZG_E_COMM = -7     /* RCCL: the library cannot be loaded, a collective call failed or timed out (kzg_comm_*) */
} kzg_status;

/* ---- lifecycle: replaces Client(port, bin, ...) + Client.start()/stop()  (reference base/miner.py:73-84,155,181) */
int kzg_create(int device_id, kzg_ctx** out);
void kzg_destroy(kzg_ctx* ctx);
const char* kzg_last_error(kzg_ctx* ctx); /* last failure of the calling thread; valid until its next failing call */
const char* kzg_version(void);
/* What the process around the context looks like, measured once by kzg_create.  The four lanes of a context overlap only
 * when the HIP runtime gives their streams different hardware queues: it has GPU_MAX_HW_QUEUES of them (default 4, read ONCE
 * when the runtime initialises) for every stream of the process.  The library never writes the environment (the axon's
 * threads may be reading it: neurons/miner.py:106-135 runs forward on worker threads): a launcher exports
 * GPU_MAX_HW_QUEUES=8 before the process's first HIP call (zkp_subnet_amd does at import), and this
 * call reports what the context actually got:
 *   out[0] lanes of a context (4)
 *   out[1] lanes measured to run concurrently: one spinning single-wave kernel per lane, the largest number alive at one
 *          instant -- out[0] when every lane has its own queue, 1 when they execute one after the other (results are the
 *          same either way; two requests in flight then gain nothing), 0 if the probe itself failed
 *   out[2] 1 if a HIP runtime was already initialised in the process when this library was loaded (an export made after
 *          that point came too late)
 *   out[3] GPU_MAX_HW_QUEUES as the environment showed it at kzg_create (0 = unset: the runtime's default) */
int kzg_runtime_info(kzg_ctx* ctx, int32_t out[4]);
/* window bits c for the signed-digit Pippenger tables; 0 = choose from the slice length.  Call before the SRS. */
int kzg_set_window(kzg_ctx* ctx, int c);
int kzg_get_window(kzg_ctx* ctx);
/* bit offsets of the windows actually in use: out[w] = first bit of window w, out[nwin] = 256; returns nwin */
int kzg_get_window_layout(kzg_ctx* ctx, int32_t* out_offsets, int max);

/* ---- SRS: replaces the prover's setup / precompute file loading (reference base/miner.py:75-84,
 *      utils/config.py:124-164).  The flat SRS holds 2^machines_scale-or-fewer worker slices of
 *      T = 2^(scale-machines_scale) points each, slice k at points [k*T, (k+1)*T).  Points stay resident
 *      ("cached SRS") together with their window multiples 2^off[w] P. */
int kzg_load_srs(kzg_ctx* ctx, const uint8_t* g1_affine_be96, uint64_t n_points, int scale, int machines_scale);
/* Same, from a ZCash-compressed file (48 bytes per point: flags 0x80 compressed | 0x40 infinity | 0x20 y-sign, x
 * big-endian) -- the reference's `uncompressed=False` setup files (base/miner.py:75-81, utils/config.py:131-150).
 * The y coordinates are recovered on the GPU (one Fp square root per point); a non-residue, x >= p or malformed flags
 * fail the load with KZG_E_POINT. */
int kzg_load_srs_compressed(kzg_ctx* ctx, const uint8_t* g1_c48, uint64_t n_points, int scale, int machines_scale);
/* The reference's own start path: Client(setup_path=...).start(scale, machines_scale) gives the prover a FILE
 * (base/miner.py:75-84; Makefile:63-74 starts mainnet from setup_24_8.uncompressed: 2^24 points, 1.6 GB).  The file
 * (96-byte records, or 48-byte compressed ones with compressed=1) is read with pread(2) straight into two pinned
 * tiles: host read, upload and GPU decode overlap, and a file truncated or replaced during the load is KZG_E_ARG (a
 * mapping would have raised SIGBUS).  All three loaders build the new tables aside and swap them in only
 * when the whole load has succeeded: after a failure (bad point, I/O, memory) the previously loaded SRS keeps serving. */
int kzg_load_srs_file(kzg_ctx* ctx, const char* path, int compressed, int scale, int machines_scale);
/* The same for ONE device of a host that spreads the worker rows over G GPUs (worker i on device i mod G: kzg_multi_*,
 * MultiDeviceClient): only the slices this context serves are read (pread of just those byte ranges), decoded, checked and
 * tabulated -- resident slice k = file slice first_slice + k * slice_stride, so worker index i = first_slice + k * slice_stride
 * is served as slice k.  Mainnet 24 / 8 on G devices: 34 / G GB of tables and ~1 / G of the start time per device instead of
 * the whole file on each (reference Makefile:63-74 starts one prover per process from the whole file: base/miner.py:75-81). */
int kzg_load_srs_file_slices(kzg_ctx* ctx, const char* path, int compressed, int scale, int machines_scale,
                             uint32_t first_slice, uint32_t slice_stride);
/* One contiguous SEGMENT of a flat SRS: file points [first_point, first_point + n_points) become the resident points
 * [0, n_points) of a single slice (machines_scale 0; n_points <= 2^scale, `scale` sizes the Pippenger window).  What device g
 * holds when ONE MSM is sharded by SRS segment over the GPUs of a host (kzg_multi_msm; BASELINE.json configs[3]). */
int kzg_load_srs_file_range(kzg_ctx* ctx, const char* path, int compressed, uint64_t first_point, uint64_t n_points, int scale);
/* seconds spent by the last successful load: [0] host copies into the pinned tiles, [1] host waiting for upload + decode
 * (decompression), [2] window-table build, [3] whole call */
int kzg_get_load_stats(kzg_ctx* ctx, double out_s[4]);
/* The loaders check every point: coordinates reduced, on the curve, AND in the prime-order subgroup G1 (E(Fp) has a
 * cofactor of ~2^126; the test is the endomorphism identity [z^2]P == -sigma(P), ~0.3 s for 2^24 points).  A failure is
 * KZG_E_POINT.  enable = 0 skips the membership part for a file whose provenance is already established: it applies to
 * the NEXT load only (whether that load succeeds or not) and the check is armed again afterwards -- a sticky switch
 * would silently cover later loads from caller memory too. */
int kzg_set_srs_subgroup_check(kzg_ctx* ctx, int enable);

/* synthetic SRS with known discrete logs (tests / benches; stands in for `fourier setup --generate-setup`,
 * reference tests/conftest.py:50-65): slice k, point j = [s0_k * tau^j] G.  s0_be32: n_slices x 32 bytes. */
int kzg_gen_srs(kzg_ctx* ctx, const uint8_t tau_be32[32], const uint8_t* s0_be32, uint32_t n_slices, int scale,
                int machines_scale);
uint64_t kzg_srs_points(kzg_ctx* ctx);
/* read back resident points: window w multiple of points [first, first+count) as affine be96 */
int kzg_srs_read(kzg_ctx* ctx, int w, uint64_t first, uint64_t count, uint8_t* out_be96);
int kzg_srs_read_compressed(kzg_ctx* ctx, int w, uint64_t first, uint64_t count, uint8_t* out_c48);

/* ---- hot path.  worker index i selects slice i of the resident SRS. */
/* replaces Client.worker_commit(i, poly)            (reference neurons/miner.py:38-45) */
int kzg_commit(kzg_ctx* ctx, uint32_t i, const uint8_t* row_be32, uint64_t T, int evaluation_form,
               uint8_t out_commitment48[48]);
/* replaces Client.worker_open(i, poly, x)           (reference neurons/miner.py:47-54) */
int kzg_open(kzg_ctx* ctx, uint32_t i, const uint8_t* row_be32, uint64_t T, int evaluation_form,
             const uint8_t alpha_be32[32], uint8_t out_eval32[32], uint8_t out_proof48[48]);
/* fused Miner.rpc_commit_and_open (reference neurons/miner.py:56-61): one upload, one IFFT, the two MSMs as one
 * batched pass (rows <= 2^18) or on two streams */
int kzg_commit_open(kzg_ctx* ctx, uint32_t i, const uint8_t* row_be32, uint64_t T, int evaluation_form,
                    const uint8_t alpha_be32[32], uint8_t out_commitment48[48], uint8_t out_eval32[32],
                    uint8_t out_proof48[48]);
/* The UNCHANGED reference miner makes two calls per request with the same row -- worker_commit(i, poly), then
 * worker_open(i, poly, x) (neurons/miner.py:56-61).  These forms take a 128-bit content tag identifying the row's bytes
 * (the host codec computes it while decoding the text); the coefficient vectors of the last four rows stay on the
 * device, and a call whose (tag, T, evaluation_form) is cached skips the INTT and keeps the row's upload off its critical
 * path.  A miss behaves exactly like kzg_commit / kzg_open and leaves its own coefficients behind.  Results are
 * identical either way, whatever the tags: a tag is a HINT -- on a hit the library uploads row_be32 beside the request
 * and compares it bit for bit with the row the slot was filled from; two different rows under one tag each get their
 * own answer (the colliding slot is dropped and the call recomputed), so a fast non-cryptographic tag is safe. */
int kzg_commit_cached(kzg_ctx* ctx, uint32_t i, const uint8_t* row_be32, uint64_t T, int evaluation_form,
                      const uint8_t content_tag[16], uint8_t out_commitment48[48]);
int kzg_open_cached(kzg_ctx* ctx, uint32_t i, const uint8_t* row_be32, uint64_t T, int evaluation_form,
                    const uint8_t content_tag[16], const uint8_t alpha_be32[32], uint8_t out_eval32[32],
                    uint8_t out_proof48[48]);
int kzg_row_cache_stats(kzg_ctx* ctx, uint64_t out_hits_misses[2]);
/* plain MSM over resident points [srs_offset, srs_offset+n): the headline kernel (BASELINE.json metric) */
int kzg_msm(kzg_ctx* ctx, const uint8_t* scalars_be32, uint64_t n, uint64_t srs_offset, uint8_t out48[48]);
/* replaces Client.fft(poly, left, inverse)          (reference neurons/validator.py:58-65); in place */
int kzg_ntt(kzg_ctx* ctx, uint8_t* inout_be32, uint64_t n, int inverse);
/* replaces Client.eval(poly, x)                     (reference neurons/validator.py:97-104) */
int kzg_eval(kzg_ctx* ctx, const uint8_t* coeffs_be32, uint64_t n, const uint8_t x_be32[32], uint8_t out_be32[32]);
/* both in one call: y = (NTT / inverse NTT of vals)(x) -- the validator's per-row challenge step
 * eval(fft(poly[i], left=True, inverse=True), alpha) (reference neurons/validator.py:115-118) with the coefficient
 * vector staying on the device */
int kzg_ntt_eval(kzg_ctx* ctx, const uint8_t* vals_be32, uint64_t n, int inverse, const uint8_t x_be32[32],
                 uint8_t out_y32[32]);

/* ---- verification: replaces Client.worker_verify(i, proof, alpha, eval, commitment)
 *      (reference neurons/validator.py:77-86; tests/test_miner.py:101-111).  Host-side pairing check
 *      e(C - y [L_i]_1, [1]_2) == e(pi, [tau_x - alpha]_2): one-off per proof, no GPU involved, so the verifier key is
 *      its own object.  *out_valid = 1 / 0; malformed or off-curve proof / commitment bytes give valid = 0. */
typedef struct kzg_vk kzg_vk;
/* tau_g2: uncompressed G2 x.c1||x.c0||y.c1||y.c0 (4 x 48 B big-endian); li_g1: [L_i(tau_y)]_1 per resident slice */
int kzg_vk_create(const uint8_t tau_g2_be192[192], const uint8_t* li_g1_be96, uint32_t n_slices, kzg_vk** out);
/* synthetic setup with known trapdoor (same tau / s0 as kzg_gen_srs): tests and benches */
int kzg_vk_create_synthetic(const uint8_t tau_be32[32], const uint8_t* s0_be32, uint32_t n_slices, kzg_vk** out);
void kzg_vk_destroy(kzg_vk* vk);
/* serialise: 192 B [tau_x]_2 + 96 B per slice (a `<setup>.vk` file); returns the slice count or a negative status */
int kzg_vk_export(const kzg_vk* vk, uint8_t* out, uint64_t out_len);
int kzg_vk_verify(const kzg_vk* vk, uint32_t i, const uint8_t proof48[48], const uint8_t alpha_be32[32],
                  const uint8_t eval_be32[32], const uint8_t commitment48[48], int* out_valid);
/* Every row of a validator step in ONE pairing check (the rows share alpha: neurons/validator.py:106-120).  Random
 * 128-bit weights r_i (getrandom) fold the n checks into two Miller loops on sum r_i (C_i - y_i L_i) and sum r_i pi_i;
 * the per-row work (decompression, G1 membership, three scalar multiplications) runs on `threads` host threads.
 * *out_all_valid = 1 only when every row is valid (an invalid one slips through with probability 2^-128); on 0 call
 * kzg_vk_verify row by row to find which.  idx: the worker index of each row. */
int kzg_vk_verify_batch(const kzg_vk* vk, uint32_t n, const uint32_t* idx, const uint8_t* proofs48,
                        const uint8_t alpha_be32[32], const uint8_t* evals_be32, const uint8_t* commitments48, int threads,
                        int* out_all_valid);

/* ---- multi-GPU: each rank reduces its SRS shard to ONE partial sum; the 192-byte partials are exchanged by
 *      the caller (RCCL all_gather over xGMI in zkp_subnet_amd.distributed) and summed on any rank. */
/* (A partial is a projective XYZZ representative: two runs over the same input may return different bytes for the
 *  same group element.  Only kzg_g1_sum's 48-byte output is canonical.) */
int kzg_msm_partial(kzg_ctx* ctx, const uint8_t* scalars_be32, uint64_t n, uint64_t srs_offset,
                    uint8_t out_xyzz192[192]);
int kzg_g1_sum(kzg_ctx* ctx, const uint8_t* partials_xyzz192, uint32_t count, uint8_t out48[48]);
/* Pianist master aggregation: sum of `count` 48-byte compressed G1 points (the worker rows' commitments,
 * sum_i commit_i = commitment of the bivariate polynomial; reference neurons/validator.py:196-198, README.md:38).
 * Inputs are decompressed on the GPU (one Fp square root each) and checked for membership in G1 (the prime-order
 * subgroup: they come from untrusted miners); malformed / off-curve / out-of-subgroup input -> KZG_E_POINT. */
int kzg_g1_sum_compressed(kzg_ctx* ctx, const uint8_t* points_c48, uint32_t count, uint8_t out48[48]);
/* Device-pointer forms for the collective path: the partial is written into / the gathered partials are read from the
 * CALLER's device memory (the tensors of an RCCL all_gather), so a step makes no host round trip for them.
 * kzg_msm_partial_resident_dev returns after its stream has drained (dev_out is complete); the caller must have
 * completed the collective (stream synchronised) before kzg_g1_sum_dev. */
int kzg_msm_partial_resident_dev(kzg_ctx* ctx, int slot, uint64_t n, uint64_t srs_offset, void* dev_out_xyzz192);
int kzg_g1_sum_dev(kzg_ctx* ctx, const void* dev_partials_xyzz192, uint32_t count, uint8_t out48[48]);
/* The same step chained through streams, one host synchronisation per MSM.  _begin queues the MSM of this rank's SRS
 * segment on a free lane, writes its 192-byte partial to dev_out_xyzz192 and makes `consumer_stream` (a hipStream_t: the
 * stream the collective is enqueued on, e.g. torch's current stream) wait for it ON THE DEVICE; it returns a ticket
 * without blocking.  After the all_gather has been enqueued on that stream, _finish makes the lane wait for
 * `producer_stream` in turn, sums the `count` gathered partials on the lane and returns the compressed point.
 * (the exchange is 192 B per rank, latency-bound -- host round trips around it are what it costs.) */
int kzg_msm_sharded_begin(kzg_ctx* ctx, int slot, uint64_t n, uint64_t srs_offset, void* dev_out_xyzz192,
                          void* consumer_stream, int* out_ticket);
int kzg_msm_sharded_finish(kzg_ctx* ctx, int ticket, const void* dev_partials_xyzz192, uint32_t count,
                           void* producer_stream, uint8_t out48[48]);

/* ---- the collective INSIDE the library ("RCCL is used directly from C++"; one process per GPU, rank g
 *      holds SRS segment g).  The only exchange of an SRS-sharded MSM is one ncclAllGather of 192 bytes per rank; these
 *      entry points own the communicator and enqueue that all_gather on the LANE's own stream, between the partial and
 *      the sum: no foreign stream, no event hand-over, one host wait per MSM, no framework underneath (the reference
 *      seam is one client object per process, base/miner.py:73-84).  RCCL is bound at the first kzg_comm_* call
 *      (dlopen of librccl.so.1 -- the copy the process already maps, if any; KZG_RCCL_LIB names another file); without
 *      it these calls fail with KZG_E_COMM and everything else works as before.
 *  kzg_comm_unique_id   ncclGetUniqueId: ONE rank calls it and hands the 128 bytes to every rank by any means it has
 *                       (a file, a socket, torch.distributed's store, MPI): the rendezvous is the caller's.
 *  kzg_comm_init        ncclCommInitRank on the context's device; collective over the `world` ranks (blocks until all
 *                       have called it).  Waits for the lanes to be idle; KZG_E_BUSY while a ticket is out.
 *  kzg_comm_init_bounded the same with a deadline: the rendezvous AND a first checked 192-byte all_gather (which connects the
 *                       transports) run on a helper thread that holds nothing of the context; when the `world` ranks have
 *                       not joined and exchanged within init_timeout_ms (0 = wait for ever = kzg_comm_init) the call returns
 *                       KZG_E_COMM and the context keeps serving -- a peer that never arrives costs one timeout, never a
 *                       wedged context (a helper still blocked inside RCCL's rendezvous is left behind; it owns nothing).
 *  kzg_comm_set_timeout per-call budget in ms for kzg_msm_sharded (0 = wait for ever, the default).  When it expires the
 *                       communicator is ABORTED (ncclCommAbort: the stuck collective leaves the stream), the call and every
 *                       later one return KZG_E_COMM until kzg_comm_destroy + kzg_comm_init -- a dead peer costs one
 *                       timeout, never a parked axon thread.  A call on ANOTHER lane whose collective was in flight under
 *                       the aborted communicator returns KZG_E_COMM too, never a result (checked after its wait).
 *  kzg_comm_info        out[0] rank, out[1] world (0 = no communicator), out[2] RCCL version (e.g. 22707), out[3] 1 if broken.
 *  kzg_comm_selftest    one small all_gather with checked content (rank i sends 192 bytes of value i + 1): run it right
 *                       after kzg_comm_init, before tables are built -- ncclCommInitRank succeeding does not prove that bytes
 *                       move between these ranks.  Collective; honours the timeout.
 *  kzg_msm_sharded      the MSM of this rank's segment (resident scalars in `slot`, points [srs_offset, srs_offset + n)
 *                       of this rank's resident SRS) -> 192-byte partial -> ncclAllGather on the lane's stream -> sum of
 *                       the `world` partials -> 48-byte compressed point, the same on every rank.  Thread-safe like every
 *                       call (each takes a lane); ranks must issue their sharded MSMs in the same order (RCCL's rule for
 *                       collectives on one communicator). */
int kzg_comm_unique_id(uint8_t out_id128[128]);
int kzg_comm_init(kzg_ctx* ctx, const uint8_t unique_id128[128], int rank, int world);
int kzg_comm_init_bounded(kzg_ctx* ctx, const uint8_t unique_id128[128], int rank, int world, int init_timeout_ms);
int kzg_comm_destroy(kzg_ctx* ctx);
int kzg_comm_set_timeout(kzg_ctx* ctx, int timeout_ms);
int kzg_comm_info(kzg_ctx* ctx, int32_t out[4]);
int kzg_comm_selftest(kzg_ctx* ctx);
int kzg_msm_sharded(kzg_ctx* ctx, int slot, uint64_t n, uint64_t srs_offset, uint8_t out48[48]);

/* ---- several GPUs behind ONE handle (SURVEY 8b proposed kzg_create(device_count, device_ids); the reference builds one
 *      prover client per process, base/miner.py:73-84).  One context per GPU inside; host threads only; thread-safe like a
 *      context.  Two layouts, chosen by the load call:
 *  ROWS -- worker index i is served by device_ids[i mod G].  The in-process form of the reference's only distribution scheme:
 *      Pianist rows are independent, one row per miner (neurons/validator.py:194-222).  Nothing is exchanged.
 *  kzg_multi_load_srs_file   every device loads, in parallel, ONLY the slices of the worker indices it serves (i = g, g + G, ...:
 *                            kzg_load_srs_file_slices) -- 1 / G of the file, of the tables and of the start time each
 *  kzg_multi_gen_srs         synthetic SRS: s0_be32_all holds ALL 2^machines_scale slice factors; device g generates and
 *                            holds only the slices of the worker indices it serves
 *  kzg_multi_commit / _open / _commit_open        = kzg_commit / kzg_open / kzg_commit_open on the device of index i
 *  kzg_multi_commit_open_rows   the rows of one challenge fanned out over the devices from host threads (up to four rows per
 *                            device in flight); out_status[k] is row k's own status -- a bad row never costs the others
 *  SEGMENTS -- ONE MSM over all the GPUs of the handle (BASELINE.json configs[3] from behind the one-client seam): a flat SRS of
 *      n_points is cut into G contiguous segments (sizes differ by at most one), segment g resident on device g.
 *  kzg_multi_load_srs_file_segments   device g reads file points [lo_g, lo_g + n_g) only (kzg_load_srs_file_range)
 *  kzg_multi_gen_srs_segments         synthetic: point j = [tau^j] G; s0_be32_per_device[g] = tau^(lo_g) (G x 32 bytes, computed
 *                            by the caller: tests and benches only); kzg_multi_segment reports lo_g and n_g
 *  kzg_multi_msm             sum_j s_j P_(srs_offset + j), j < n: device g computes the partial of its part of the range on its
 *                            own lane (all devices concurrently, scalars uploaded straight to their device), the G 192-byte
 *                            partials come back through the host and are summed once -- no collective, the exchange is G x 192 B.
 *                            Bit-identical to kzg_msm over the same points on one device.
 *  kzg_multi_upload_fr / kzg_multi_msm_resident   the same with the scalars resident in HBM (segment g's scalars in slot
 *                            `slot` of device g): what a serving loop and bench.py time
 *  After a load that failed on ANY device the handle refuses every routed call (KZG_E_ARG) until a load has succeeded on all
 *  of them: the devices that did load already hold the new SRS, so no layout describes the handle in between.
 *  kzg_multi_ctx             the k-th per-GPU context: every other call of this header applies to it */
typedef struct kzg_multi kzg_multi;
int kzg_multi_create(int device_count, const int* device_ids, kzg_multi** out);
void kzg_multi_destroy(kzg_multi* m);
const char* kzg_multi_last_error(kzg_multi* m); /* last failure of the calling thread */
int kzg_multi_count(kzg_multi* m);
kzg_ctx* kzg_multi_ctx(kzg_multi* m, int k);
int kzg_multi_device_of(kzg_multi* m, uint32_t i);
int kzg_multi_load_srs_file(kzg_multi* m, const char* path, int compressed, int scale, int machines_scale);
int kzg_multi_gen_srs(kzg_multi* m, const uint8_t tau_be32[32], const uint8_t* s0_be32_all, int scale, int machines_scale);
int kzg_multi_load_srs_file_segments(kzg_multi* m, const char* path, int compressed, uint64_t n_points);
int kzg_multi_gen_srs_segments(kzg_multi* m, const uint8_t tau_be32[32], const uint8_t* s0_be32_per_device, uint64_t n_points);
int kzg_multi_segment(kzg_multi* m, int g, uint64_t out_first_count[2]);
int kzg_multi_msm(kzg_multi* m, const uint8_t* scalars_be32, uint64_t n, uint64_t srs_offset, uint8_t out48[48]);
int kzg_multi_upload_fr(kzg_multi* m, int slot, const uint8_t* scalars_be32, uint64_t n, uint64_t srs_offset);
int kzg_multi_msm_resident(kzg_multi* m, int slot, uint8_t out48[48]);
int kzg_multi_commit(kzg_multi* m, uint32_t i, const uint8_t* row_be32, uint64_t T, int evaluation_form,
                     uint8_t out_commitment48[48]);
int kzg_multi_open(kzg_multi* m, uint32_t i, const uint8_t* row_be32, uint64_t T, int evaluation_form,
                   const uint8_t alpha_be32[32], uint8_t out_eval32[32], uint8_t out_proof48[48]);
int kzg_multi_commit_open(kzg_multi* m, uint32_t i, const uint8_t* row_be32, uint64_t T, int evaluation_form,
                          const uint8_t alpha_be32[32], uint8_t out_commitment48[48], uint8_t out_eval32[32],
                          uint8_t out_proof48[48]);
int kzg_multi_commit_open_rows(kzg_multi* m, uint32_t n_rows, const uint32_t* indices, const uint8_t* rows_be32, uint64_t T,
                               int evaluation_form, const uint8_t alpha_be32[32], uint8_t* out_commitments48,
                               uint8_t* out_evals32, uint8_t* out_proofs48, int* out_status);

/* ---- device-resident inputs (what a serving loop and bench.py use: inputs already in HBM when timing starts).
 *      slot in [0, 4).  to_mont=1 stores Montgomery form (rows for commit/open), 0 canonical (MSM scalars). */
int kzg_upload_fr(kzg_ctx* ctx, int slot, const uint8_t* be32, uint64_t n, int to_mont);
int kzg_msm_resident(kzg_ctx* ctx, int slot, uint64_t n, uint64_t srs_offset, uint8_t out48[48]);
int kzg_msm_partial_resident(kzg_ctx* ctx, int slot, uint64_t n, uint64_t srs_offset, uint8_t out_xyzz192[192]);
/* Ticketed form of the two calls above: several requests in flight from ONE host thread.  submit queues the MSM on a
 * free lane and returns; wait blocks for that ticket and writes 48 (partial=0) or 192 bytes (exactly one waiter per
 * ticket).  MSM i+1's sort/accumulate then overlaps the latency-bound tail of MSM i.  Results are identical to the
 * blocking calls.  With every lane taken submit fails with KZG_E_BUSY; a blocking call made while tickets are
 * outstanding uses a free lane, or fails with KZG_E_BUSY when every lane is parked under a ticket; operations that
 * need the whole context (SRS load, kzg_upload_fr, kzg_ntt_resident) fail with KZG_E_BUSY while any ticket is out. */
int kzg_msm_submit(kzg_ctx* ctx, int slot, uint64_t n, uint64_t srs_offset, int partial, int* out_ticket);
int kzg_msm_wait(kzg_ctx* ctx, int ticket, uint8_t* out);
/* Gives up a ticket of kzg_msm_submit / kzg_msm_sharded_begin whose result will not be collected (the collective
 * between _begin and _finish raised, a peer died): drains the lane and frees it.  Without it the lane would stay parked. */
int kzg_msm_cancel(kzg_ctx* ctx, int ticket);
int kzg_commit_open_resident(kzg_ctx* ctx, uint32_t i, int slot, uint64_t T, int evaluation_form,
                             const uint8_t alpha_be32[32], uint8_t out_commitment48[48], uint8_t out_eval32[32],
                             uint8_t out_proof48[48]);
int kzg_ntt_resident(kzg_ctx* ctx, int slot, uint64_t n, int inverse); /* in place on the slot */

/* ---- pinned host staging.  acquire hands out one of four page-locked buffers of at least `bytes` bytes (it waits
 *      when all are held); release returns it.  A host that decodes the synapse's base64 text itself
 *      (zkp_subnet_amd/csrc/wire_py.c) writes the 32-byte scalars straight into it and passes the pointer as
 *      row_be32 / scalars_be32: the upload then runs at PCIe speed with no pageable bounce and no page faults, and
 *      concurrent requests each decode into their own buffer. */
int kzg_staging_acquire(kzg_ctx* ctx, uint64_t bytes, void** out_ptr, int* out_token);
int kzg_staging_release(kzg_ctx* ctx, int token);
/* Long rows: start the upload of bytes [offset, offset + bytes) of a held staging buffer NOW and return at once, so that
 * the copy engine moves tile k while the host decodes tile k + 1 of the synapse's text (the reference ships the whole
 * polynomial as text per call, neurons/miner.py:39,48).  Flushes are contiguous from offset 0 (multiples of 32 bytes).  A
 * compute call that is then handed the buffer's pointer finds the flushed prefix on the device and skips its own upload
 * (it waits for the copy on its stream, not on the host); whatever was not flushed is uploaded the ordinary way.
 * The flushes are ONE-SHOT: they serve the first compute call that is handed the pointer; a second call on the same held
 * buffer (whose bytes the holder may have rewritten) uploads the ordinary way, and the next flush starts again at
 * offset 0.  kzg_staging_release forgets the flushes. */
int kzg_staging_flush(kzg_ctx* ctx, int token, uint64_t offset, uint64_t bytes);

/* ---- where the result point is encoded.  1 (default): the XYZZ working form of the ONE point a request produces
 *      comes back in the request's single device-to-host copy and the host does the affine conversion (one Fp
 *      inversion) + ZCash compression -- a few microseconds instead of a ~140 us single-lane GPU kernel.
 *      0: encode on the GPU (k_g1_compress).  Results are identical; the *_dev entry points always stay on the GPU. */
int kzg_set_host_finish(kzg_ctx* ctx, int enable);

/* ---- per-stage HIP-event timings of the last hot-path call (events recorded on the ctx's own stream) */
enum {
    KZG_T_DECODE = 0, KZG_T_NTT, KZG_T_DIGITS, KZG_T_SCAN, KZG_T_SCATTER, KZG_T_ACCUMULATE, KZG_T_FIXUP,
    KZG_T_TREE, KZG_T_FINAL, KZG_T_POLY, KZG_T_TOTAL, KZG_T_COLLECTIVE /* kzg_msm_sharded: pack + all_gather + sum */,
    KZG_T_COUNT
};
/* enable: 0 off; 1 events around every stage (concurrent calls then serialise on one lane so that stage times stay
   attributable); 2 events around the accumulate kernel only (two per launch; nothing is serialised) */
int kzg_set_profiling(kzg_ctx* ctx, int enable);
int kzg_get_timings(kzg_ctx* ctx, float* out_ms, int count); /* accumulated over the last call's MSMs */
/* fixed-size MSM plan facts for roofline bookkeeping: entries per lane, lanes, buckets, windows */
int kzg_msm_plan(kzg_ctx* ctx, uint64_t n, int32_t out[4]);
/* The bound of the dominant kernel as a measurement of THIS device at THIS moment (SURVEY 8d "confirm on the box"):
 * k_msm_accumulate is bound by the issue rate of v_mad_u64_u32, and boxes of one pool differ by up to
 * 14 % in it.  Runs a chain of that one instruction with `waves_per_simd` waves on every SIMD for ~1.5 ms (exclusive:
 * waits for the lanes to be idle; KZG_E_BUSY while tickets are out).  out[0] ns per wave-instruction per SIMD, out[1]
 * G wave-instructions/s of the whole chip, out[2] s_memtime ticks per ns over the launch (the clock the SIMDs ran at, in
 * GHz, when s_memtime counts shader clocks), out[3] kernel ms, out[4] SIMDs, out[5] ticks per instruction of one wave.
 * bench.py calls it right after its timed region; nothing on the serving path does. */
int kzg_calibrate(kzg_ctx* ctx, int waves_per_simd, double out[6]);

/* ---- host-side wire codec (Prove.poly is a list of 43-char unpadded base64 strings, reference
 *      base/protocol.py:35-40).  packed: n x 43 chars, no separators.  Pure host code. */
int kzg_b64_decode_fr(const char* packed43, uint64_t n, uint8_t* out_be32);
int kzg_b64_encode_fr(const uint8_t* be32, uint64_t n, char* out_packed43);

#ifdef __cplusplus
}
#endif
#endif /* KZG_MI355X_H */
