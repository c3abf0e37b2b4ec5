// pipeline.hip -- the host-side sequencing of the HIP kernels: ONE Pippenger MSM on a lane (msm_core: sort -> accumulate ->
// fold -> tree -> final; kernels in msm_sort / msm_accumulate / msm_tree.hip) and one worker row's commit and / or open (commit_open_dev: INTT -> MSM ||
// evaluation + quotient -> MSM; kernels in fr_ntt.hip / fr_poly.hip).  What it computes is what the reference's prover computes behind
// Client.worker_commit / worker_open (reference neurons/miner.py:38-54); see ctx.hip.h for the map of the library.
#include "ctx.hip.h"

using namespace kzg_impl;

namespace kzg_impl {

// sorted entries per accumulate lane.  Large MSMs (throughput-bound): the grid is a whole number of "rounds" of 131072
// lanes (2 waves per SIMD on 256 CUs: the second wave hides the point loads) so that the last round is not a partially
// filled tail; chunks stay <= 512 entries.  Small MSMs (latency-bound: one wave already saturates a SIMD's integer
// issue, ~10.5 us per mixed addition): 65536 lanes = one wave per SIMD, which halves the number of carries to fold.
// (the shortest chunk: 8 until the short rows' fold became one launch that is linear in the carries per bucket; with it
// 6 is the optimum -- 2^12 row 0.334 -> 0.328 ms, 2^10 0.275 -> 0.268; 4 gives the fold back what the accumulate gains:
// `profiles/r03_ab_min_chunk.log`)
#ifndef KZG_MIN_CHUNK
#define KZG_MIN_CHUNK 6
#endif
int pick_chunk(uint64_t entries) {
    const uint64_t lanes = 131072;
    if (entries <= lanes * 16) {
        const uint64_t k = (entries + lanes / 2 - 1) / (lanes / 2);
        return (int)(k < KZG_MIN_CHUNK ? KZG_MIN_CHUNK : k);
    }
    const uint64_t rounds = (entries + lanes * 512 - 1) / (lanes * 512);
    const uint64_t k = (entries + lanes * rounds - 1) / (lanes * rounds);
    return (int)k;
}
// ---- the MSM pipeline on device-resident scalars -> one XYZZ point at out_xyzz (device), on lane L.
// With scalars2 != null: TWO MSMs over the same n points in one pass (the commitment and the opening of one row):
// set b is sorted into bucket set b, the sort / accumulate / fold / tree kernels simply see twice the buckets, the
// tree stops at two roots and out_xyzz[0..1] receive the two sums.  One kernel sequence, one latency-bound tail.
// The only host wait inside is on the 4-byte fold-depth read-back; the calling thread holds no lock meanwhile.
// The bucket tree on stream s: merges level arrays until `stop` nodes are left.  Three buffers in rotation (a level reads its
// own array and the P array of the level below, writes the next) plus a fourth for the two-level launches.  On return b.in is
// the last level array, b.prev the level below it, b.out a buffer nothing reads any more.
struct TreeBufs {
    g1_xyzz_t *in, *prev, *out, *extra;
};
static void run_tree(hipStream_t s, TreeBufs& b, uint32_t n_in, uint32_t stop) {
    for (int level = 0; n_in > stop;) {
        if ((n_in >> 2) >= stop && msm_tree_level2_ok(n_in, level)) {
            // two narrow levels per launch: the level + 1 P array goes to `out`, the level + 2 array to `extra`
            launch_msm_tree_level2(s, b.in, b.prev, b.out, b.extra, n_in, level);
            g1_xyzz_t *old_in = b.in, *old_prev = b.prev;
            b.in = b.extra;
            b.prev = b.out;
            b.out = old_prev;
            b.extra = old_in;
            level += 2;
            n_in >>= 2;
            continue;
        }
        launch_msm_tree_level(s, b.in, b.prev, b.out, n_in, level);
        g1_xyzz_t* recycled = b.prev;
        b.prev = b.in;
        b.in = b.out;
        b.out = recycled;
        level++;
        n_in >>= 1;
    }
}
int msm_core(kzg_ctx* ctx, Lane& L, const uint32_t* scalars, int mont, uint64_t n, uint64_t srs_offset,
             g1_xyzz_t* out_xyzz, const uint32_t* scalars2, int mont2) {
    hipStream_t s = L.stream;
    const int nbatch = scalars2 ? 2 : 1;
    if (n == 0) {
        HIPCHK(ctx, hipMemsetAsync(out_xyzz, 0, nbatch * sizeof(g1_xyzz_t), s));
        return KZG_OK;
    }
    if (srs_offset + n > ctx->stride) return fail(ctx, KZG_E_ARG, "MSM range exceeds the resident SRS");
    const uint64_t entries = n * (uint64_t)ctx->nwin * nbatch;
    if (entries >= ((uint64_t)1 << 32)) return fail(ctx, KZG_E_ARG, "MSM too large for 32-bit entry indices");
    MsmShape sh;
    sh.c = ctx->c; sh.nwin = ctx->nwin; sh.lay = ctx->lay; sh.nbuckets = ctx->nbuckets * nbatch; sh.n = n;
    sh.nbatch = nbatch;
    sh.srs_offset = srs_offset; sh.srs_stride = ctx->stride; sh.chunk = pick_chunk(entries);
    const uint32_t nchunks = (uint32_t)((entries + sh.chunk - 1) / sh.chunk);
    const size_t B = sh.nbuckets;
    L.expect_short = entries <= ((uint64_t)1 << 24);   // up to ~3 ms of GPU time (a 2^20-point MSM)
    L.expect_us += 400 + (uint32_t)(entries / 4096);     // ~0.2 ns per sorted entry + the latency-bound tail
    // Sort mode.  Fast: no count pass, fixed-capacity partition regions -- right for well-spread scalars (field elements
    // of a polynomial), wrong for skewed ones, where a region overflows: that is detected on the device, costs one wasted
    // sort (the queued accumulate sees an empty MSM), and is remembered for the lane's next few calls.
    bool fast = msm_sort_fast_ok(sh) && L.skew_hint == 0;
    if (L.skew_hint > 0) L.skew_hint--;
    HIPCHK(ctx, L.rank.ensure(msm_sort_parted_entries(sh, msm_sort_fast_ok(sh)) * 8));   // partitioned (key_low, value) pairs
    HIPCHK(ctx, L.sorted.ensure(entries * 4));
    HIPCHK(ctx, L.hist.ensure(16384 * 4));
    HIPCHK(ctx, L.offsets.ensure((B + 1) * 4));
    // (+ 16 KB each: whichever buffer is free after the last level also holds the 2 x 32 doubled components of the final)
    HIPCHK(ctx, L.bufA.ensure(B * sizeof(g1_xyzz_t) + 16384));
    HIPCHK(ctx, L.bufB.ensure(B * sizeof(g1_xyzz_t) / 2 + 16384));  // level arrays: n/2^L nodes x L components <= B/2
    HIPCHK(ctx, L.bufC.ensure(B * sizeof(g1_xyzz_t) / 2 + 16384));
    HIPCHK(ctx, L.bufD.ensure((size_t)(LP_MAX_OPS + 64) * sizeof(g1_xyzz_t) + 16384));   // fourth buffer of the two-level launches
    HIPCHK(ctx, L.carries.ensure((size_t)nchunks * sizeof(g1_xyzz_t)));
    HIPCHK(ctx, L.carry_key.ensure((size_t)nchunks * 4));
    uint32_t* max_len_d = L.flags() + 2;
    uint32_t* max_len_h = reinterpret_cast<uint32_t*>(L.pin + PIN_MAXLEN);
    auto sort_and_publish = [&](bool fast_mode, bool ws_clean) {
        const SortTail tail{(uint32_t)sh.chunk, L.bufA.as<g1_xyzz_t>(), reinterpret_cast<uint32_t*>(L.pin_dev + PIN_MAXLEN),
                            reinterpret_cast<uint32_t*>(L.pin_dev + PIN_SEQ_SORT), ++L.sort_seq};
        launch_msm_sort(s, sh, scalars, mont, scalars2, mont2, L.hist.as<uint32_t>(), ws_clean, L.rank.as<uint2>(),
                        L.offsets.as<uint32_t>(), L.sorted.as<uint32_t>(), max_len_d, fast_mode, max_len_d + 1,
                        &tail);
    };
    {
        Span sp(ctx, L, KZG_T_DIGITS);
        // the longest run of carries decides how many fold steps are launched; it depends on the offsets only, so
        // its read-back (with the sort's overflow word) completes while the accumulate kernel runs and costs no bubble.
        // The sort's last kernel takes that maximum, marks the empty buckets and publishes both words itself (SortTail).
        sort_and_publish(fast, L.sort_ws_clean);
        L.sort_ws_clean = true;
        HIPCHK(ctx, hipEventRecord(L.ev_sorted, s));
    }
    {
        Span sp(ctx, L, KZG_T_ACCUMULATE);
        launch_msm_accumulate(s, sh, ctx->table.as<g1_affine_t>(), L.offsets.as<uint32_t>(), L.sorted.as<uint32_t>(),
                              L.bufA.as<g1_xyzz_t>(), L.carries.as<g1_xyzz_t>(), L.carry_key.as<uint32_t>(), nchunks);
    }
    // short rows: the tail's ~20 launches must be queued while the (short) accumulate runs -- poll for the two words
    // instead of sleeping on the event
    auto wait_sorted = [&]() -> hipError_t {
#ifndef KZG_NO_POLL
        if (ctx->profiling != 1 && L.expect_short && poll_pinned(ctx, L, PIN_SEQ_SORT, L.sort_seq)) return hipSuccess;
#endif
        return hipEventSynchronize(L.ev_sorted);
    };
    HIPCHK(ctx, wait_sorted());
    if (fast && max_len_h[1]) {   // a region overflowed: skewed scalars.  Exact sort + accumulate once more.
        L.skew_hint = 16;
        {
            Span sp(ctx, L, KZG_T_DIGITS);
            sort_and_publish(false, true);
            HIPCHK(ctx, hipEventRecord(L.ev_sorted, s));
        }
        {
            Span sp(ctx, L, KZG_T_ACCUMULATE);
            launch_msm_accumulate(s, sh, ctx->table.as<g1_affine_t>(), L.offsets.as<uint32_t>(), L.sorted.as<uint32_t>(),
                                  L.bufA.as<g1_xyzz_t>(), L.carries.as<g1_xyzz_t>(), L.carry_key.as<uint32_t>(), nchunks);
        }
        HIPCHK(ctx, wait_sorted());
    }
    {
        Span sp(ctx, L, KZG_T_FIXUP);
        if (nchunks && msm_fold_bucket_ok(sh.nbuckets, *max_len_h)) {
            launch_fold_bucket(s, L.offsets.as<uint32_t>(), (uint32_t)sh.chunk, sh.nbuckets, L.carries.as<g1_xyzz_t>(),
                               L.bufA.as<g1_xyzz_t>());
        } else {
            for (uint32_t d = 1; d < *max_len_h; d <<= 1)
                launch_fold_step(s, L.offsets.as<uint32_t>(), L.carry_key.as<uint32_t>(), (uint32_t)sh.chunk, nchunks, d,
                                 L.carries.as<g1_xyzz_t>());
            launch_fold_heads(s, L.offsets.as<uint32_t>(), L.carry_key.as<uint32_t>(), (uint32_t)sh.chunk, nchunks,
                              L.carries.as<g1_xyzz_t>(), L.bufA.as<g1_xyzz_t>());
        }
    }
    TreeBufs tb{L.bufA.as<g1_xyzz_t>(), L.bufC.as<g1_xyzz_t>(), L.bufB.as<g1_xyzz_t>(), L.bufD.as<g1_xyzz_t>()};
    {
        Span sp(ctx, L, KZG_T_TREE);
        run_tree(s, tb, sh.nbuckets, (uint32_t)nbatch);
    }
    {
        Span sp(ctx, L, KZG_T_FINAL);
        // `out` (the buffer the last level did not write and no longer reads) holds the doubled components in between
        launch_msm_final(s, tb.in, tb.prev, ctx->c - 1, nbatch, out_xyzz, tb.out);
    }
    HIPCHK(ctx, hipGetLastError());
    return KZG_OK;
}

// twiddle / 1/n tables are shared by all lanes: built once under the ctx mutex, complete before the mutex is dropped
int ensure_twiddles(kzg_ctx* ctx, Lane& L, int log_n, int inverse, uint32_t** tw, uint32_t** invn) {
    std::lock_guard<std::mutex> lk(ctx->mu);
    auto& m = inverse ? ctx->tw_inv : ctx->tw_fwd;
    bool built = false;
    if (log_n >= 1 && !m.count(log_n)) {
        DevBuf b;
        HIPCHK(ctx, b.ensure(((size_t)1 << (log_n - 1)) * 48));   // nine 29-bit limbs per twiddle in a 48-byte slot
        launch_fr_twiddles(L.stream, b.as<uint32_t>(), log_n, inverse);
        m[log_n] = std::move(b);
        built = true;
    }
    *tw = log_n >= 1 ? m[log_n].as<uint32_t>() : nullptr;
    if (invn) {
        if (!ctx->inv_n.count(log_n)) {
            DevBuf b;
            HIPCHK(ctx, b.ensure(32));
            launch_fr_inv_pow2(L.stream, b.as<uint32_t>(), log_n);
            ctx->inv_n[log_n] = std::move(b);
            built = true;
        }
        *invn = ctx->inv_n[log_n].as<uint32_t>();
    }
    if (built) HIPCHK(ctx, hipStreamSynchronize(L.stream));
    return KZG_OK;
}
// coefficients (Montgomery) of the row; returns pointer in *coeffs.  row_dev: Montgomery-form row.
int row_to_coeffs(kzg_ctx* ctx, Lane& L, const uint32_t* row_dev, uint64_t T, int evaluation_form, const uint32_t** coeffs,
                  uint32_t* dst) {   // dst: where the coefficients go instead of the lane's own buffer (row cache)
    if (!evaluation_form || T == 1) {
        if (dst && dst != row_dev) HIPCHK(ctx, hipMemcpyAsync(dst, row_dev, T * 32, hipMemcpyDeviceToDevice, L.stream));
        *coeffs = dst ? dst : row_dev;
        return KZG_OK;
    }
    int lg = ilog2_exact(T);
    if (lg < 0) return fail(ctx, KZG_E_ARG, "evaluation-form row length must be a power of two");
    uint32_t *tw, *invn;
    int rc = ensure_twiddles(ctx, L, lg, 1, &tw, &invn);
    if (rc) return rc;
    if (!dst) {
        HIPCHK(ctx, L.coeffB.ensure(T * 32));
        dst = L.coeffB.as<uint32_t>();
    }
    HIPCHK(ctx, L.ntt_mid.ensure(T * 48));
    Span sp(ctx, L, KZG_T_NTT);
    launch_fr_ntt(L.stream, row_dev, dst, lg, tw, invn, L.ntt_mid.as<uint32_t>());
    *coeffs = dst;
    return KZG_OK;
}
int check_worker(kzg_ctx* ctx, uint32_t i, uint64_t T) {
    int rc = need_srs(ctx);
    if (rc) return rc;
    if (T == 0) return fail(ctx, KZG_E_ARG, "empty polynomial");
    if (T > ctx->T) return fail(ctx, KZG_E_ARG, "polynomial longer than the worker's SRS slice");
    // (the last resident slice may be shorter than T: a single truncated slice, see plan_table)
    if ((uint64_t)i * ctx->T >= ctx->stride || (uint64_t)i * ctx->T + T > ctx->stride)
        return fail(ctx, KZG_E_ARG, "worker index outside the resident SRS");
    return KZG_OK;
}
// upload BE scalars to `dst` (device limbs); dst must hold n*32 bytes
// the device twin of a staging buffer whose first `bytes` bytes have been flushed (kzg_staging_flush), or null.  The
// caller holds that buffer (it was handed its pointer), so nobody else touches the record meanwhile.
// One-shot: the twin serves the FIRST API call that asks for it (that call may ask more than once: its upload and its
// row-cache verification); any later call finds the flushes forgotten -- the holder may have rewritten the pinned buffer
// between two calls, and a stale twin would be a wrong answer with no error.
const uint8_t* flushed_twin(kzg_ctx* ctx, const uint8_t* host_ptr, uint64_t bytes, hipEvent_t* ev) {
    for (Stage& st : ctx->stage)
        if (host_ptr && st.p_pub.load(std::memory_order_acquire) == host_ptr) {
            if (st.consumed_by && st.consumed_by != call_id_current()) {
                st.flushed = 0;                // a second call on the same held buffer: ordinary upload from the host bytes
                return nullptr;
            }
            if (st.flushed >= bytes && st.flushed && st.twin.p) {
                st.consumed_by = call_id_current();
                *ev = st.ev;
                return static_cast<const uint8_t*>(st.twin.p);
            }
            return nullptr;
        }
    return nullptr;
}
int upload_fr(kzg_ctx* ctx, Lane& L, const uint8_t* be32, uint64_t n, uint32_t* dst, int to_mont) {
    if (!n) return KZG_OK;
    Span sp(ctx, L, KZG_T_DECODE);
    hipEvent_t ev = nullptr;
    if (const uint8_t* twin = flushed_twin(ctx, be32, n * 32, &ev)) {   // uploaded tile by tile while the host decoded
        HIPCHK(ctx, hipStreamWaitEvent(L.stream, ev, 0));
        L.in_be_src = twin;
    } else {
        HIPCHK(ctx, L.in_be.ensure(n * 32));
        HIPCHK(ctx, hipMemcpyAsync(L.in_be.p, be32, n * 32, hipMemcpyHostToDevice, L.stream));
        L.in_be_src = L.in_be.as<uint8_t>();
    }
    launch_fr_from_be(L.stream, L.in_be_src, dst, n, to_mont, L.flags());
    return KZG_OK;
}


// commit and/or open on a device-resident Montgomery row, on the lane(s) the call holds.  With both requested:
//  * rows up to 2^18 (latency-bound: dozens of small dependent kernels): the commitment MSM(U_i, f) and the opening
//    MSM(U_i, q) run as ONE batched pass over the slice's window tables (msm_core with two scalar sets) -- one sort,
//    one accumulate launch, one bucket tree with two roots: a single tail instead of two;
//  * longer rows (throughput-bound): when a second lane is free the opening (evaluation, quotient, MSM) runs there,
//    concurrently with the commitment MSM -- they share only the read-only coefficients -- so that each one's sort and
//    tail hide under the other's accumulate; otherwise (another host thread's request holds the other lanes, or
//    profiling is on) the two MSMs run back to back on this lane and the overlap comes from the other requests.
#ifndef KZG_BATCHED_ROW_MAX
#define KZG_BATCHED_ROW_MAX ((uint64_t)1 << 18)
#endif
int commit_open_dev(kzg_ctx* ctx, LaneHold& H, uint32_t i, const uint32_t* row_dev, uint64_t T, int evaluation_form,
                    const uint8_t* alpha_be32, uint8_t* out_c48, uint8_t* out_eval32, uint8_t* out_p48,
                    const uint32_t* coeffs_ready, uint32_t* coeffs_dst, const VerifyJob* verify) {
    Lane& A = H.L();
    hipStream_t s = A.stream;
    const uint32_t* coeffs = coeffs_ready;     // row cache hit: the coefficient vector is already on the device
    int rc = KZG_OK;
    if (!coeffs) rc = row_to_coeffs(ctx, A, row_dev, T, evaluation_form, &coeffs, coeffs_dst);
    if (rc) return rc;
    const uint64_t offset = (uint64_t)i * ctx->T;
    g1_xyzz_t* res = A.res();
    const bool both = out_c48 && out_p48;
    const bool batched = both && T <= KZG_BATCHED_ROW_MAX;
    Lane* B = (both && !batched) ? H.second() : nullptr;   // lane of the opening, when one is free
    Lane& O = B ? *B : A;
    hipStream_t so = O.stream;
    if (B) {
        HIPCHK(ctx, hipEventRecord(A.ev_coeffs, s));
        HIPCHK(ctx, hipStreamWaitEvent(so, A.ev_coeffs, 0));
    }
    if (out_c48 && !batched) {
        rc = msm_core(ctx, A, coeffs, 1, T, offset, res);
        if (rc) return rc;
    }
    if (out_p48) {
        uint32_t* alpha_m = reinterpret_cast<uint32_t*>(A.tail + TB_ALPHA_M);
        uint32_t* y_m = reinterpret_cast<uint32_t*>(A.tail + TB_Y_M);
        const uint64_t nchunks = (T + 3) / 4;
        HIPCHK(ctx, O.hbuf.ensure((nchunks + (nchunks >> 1) + 64) * 32));
        HIPCHK(ctx, O.hnext.ensure((nchunks + (nchunks >> 1) + 64) * 32));
        HIPCHK(ctx, O.qbuf.ensure(T * 32));
        {
            Span sp(ctx, A, KZG_T_POLY, so);
            // alpha rides in as an argument of the opening's first kernel, y leaves big-endian from its scan kernel
            launch_poly_open(so, coeffs, T, alpha_m, O.hbuf.as<uint32_t>(), O.hnext.as<uint32_t>(), y_m,
                             O.qbuf.as<uint32_t>(), alpha_be32, A.flags(), A.tail + TB_EVAL);
        }
        if (batched) {
            // the quotient has T - 1 coefficients; k_poly_quotient leaves a zero in slot T - 1, so it rides as a second
            // length-T scalar set
            rc = msm_core(ctx, A, coeffs, 1, T, offset, res, O.qbuf.as<uint32_t>(), 0);
        } else {
            rc = msm_core(ctx, O, O.qbuf.as<uint32_t>(), 0, T - 1, offset, res + 1);
        }
        if (rc) return rc;
        if (B) {
            HIPCHK(ctx, hipEventRecord(B->ev_done, so));
            HIPCHK(ctx, hipStreamWaitEvent(s, B->ev_done, 0));
        }
    }
    queue_encode(ctx, A, out_c48 != nullptr, out_p48 != nullptr);
    if (verify) {   // row-cache hit: queued LAST, so that its few runtime calls cost host time while the GPU is busy with the
        // request's own kernels; the copy and the comparison run beside them, the publish waits for the verdict
        uint32_t* vflag = reinterpret_cast<uint32_t*>(A.tail + TB_VERIFY);
        HIPCHK(ctx, hipMemsetAsync(vflag, 0, 4, A.vstream));
        hipEvent_t fev = nullptr;
        const uint32_t* mine = reinterpret_cast<const uint32_t*>(flushed_twin(ctx, verify->row_be32, verify->T * 32, &fev));
        if (mine) {      // the row is already on the device (flushed tile by tile during the decode)
            HIPCHK(ctx, hipStreamWaitEvent(A.vstream, fev, 0));
        } else {
            HIPCHK(ctx, A.vbuf.ensure(verify->T * 32));
            HIPCHK(ctx, hipMemcpyAsync(A.vbuf.p, verify->row_be32, verify->T * 32, hipMemcpyHostToDevice, A.vstream));
            mine = A.vbuf.as<uint32_t>();
        }
        launch_words_differ(A.vstream, mine, verify->cached_raw, verify->T * 8, vflag);
        HIPCHK(ctx, hipEventRecord(A.ev_verify, A.vstream));
        HIPCHK(ctx, hipStreamWaitEvent(s, A.ev_verify, 0));   // the record must carry TB_VERIFY's final value
    }
    rc = finish(ctx, A);
    if (rc) return rc;
    if (out_c48 && out_p48 && ctx->host_finish)     // both points, one shared inversion
        kzg_host::xyzz_pair_to_c48(reinterpret_cast<const uint32_t*>(A.pin + TB_RES0),
                                   reinterpret_cast<const uint32_t*>(A.pin + TB_RES1), out_c48, out_p48);
    else {
        if (out_c48) result_c48(ctx, A, 0, out_c48);
        if (out_p48) result_c48(ctx, A, 1, out_p48);
    }
    if (out_p48) memcpy(out_eval32, A.pin + TB_EVAL, 32);
    H.clean = true;
    return KZG_OK;
}

}  // namespace kzg_impl
