// C-ABI of libkzg_mi355x.so (declared in include/kzg_mi355x.h): context, resident SRS + window tables,
// workspace, and the host-side sequencing of the HIP kernels.  The seam it fills is the prover client of the
// reference miner (reference base/miner.py:73-84 lifecycle; neurons/miner.py:38-61 commit / open).
// No CPU arithmetic fallback exists here: if HIP fails, the call fails.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/kzg_mi355x.h"
#include "fr_kernels.hip.h"
#include "msm.hip.h"

#define KZG_VERSION "kzg_mi355x 0.1 (gfx950)"
#define N_SLOTS 4

namespace {

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        size_t want = bytes + (bytes >> 3) + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) {
            e = hipMalloc(&p, bytes);
            want = bytes;
        }
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    template <class T>
    T* as() const { return reinterpret_cast<T*>(p); }
};

struct StageSpan {
    int stage;
    hipEvent_t a, b;
};

// One MSM in flight: its own stream and sort / bucket workspace.  Lane 0 runs on the context's main stream; lane 1
// takes the opening MSM of a long row's commit+open and every second ticket of kzg_msm_submit, so that one MSM's sort
// and latency-bound tail (carry fold, bucket tree, final combination) hide under the other's accumulate.
struct MsmLane {
    hipStream_t stream = nullptr;
    DevBuf rank, sorted, hist, offsets, bufA, bufB, bufC, carries, carry_key;
    DevBuf res, small;           // result point + its encoded form, for the ticketed (asynchronous) MSM
    hipEvent_t ev_sorted = nullptr, ev_done = nullptr;
    bool busy = false;           // a ticket is outstanding on this lane
    bool partial = false;
    // profiling spans of the call running on this lane
    std::vector<hipEvent_t> ev_pool;
    size_t ev_used = 0;
    std::vector<StageSpan> spans;
};
#define N_LANES 2

}  // namespace

struct kzg_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::mutex mu;
    std::string err;
    int c_user = 0, c = 0, nwin = 0;
    WinLayout lay;
    uint32_t nbuckets = 0;
    // resident SRS + window tables: table[w*stride + j] = 2^(c*w) P_j
    DevBuf table;
    uint64_t stride = 0, T = 0;
    int scale = 0, mscale = 0;
    // workspace
    DevBuf in_be, scal, res, coeffA, coeffB, qbuf, hbuf, hnext, small, out_be;
    MsmLane lane[N_LANES];
    hipEvent_t ev_coeffs = nullptr;
    DevBuf slot[N_SLOTS];
    uint64_t slot_n[N_SLOTS] = {0, 0, 0, 0};
    int slot_mont[N_SLOTS] = {0, 0, 0, 0};
    std::map<int, DevBuf> tw_fwd, tw_inv, inv_n;
    uint32_t* flags = nullptr;   // device: [0] bad scalar, [1] bad point
    uint8_t* host_pin = nullptr; // pinned staging for small results
    void* stage_host = nullptr;  // pinned staging for a caller-decoded polynomial (kzg_staging_buffer)
    size_t stage_cap = 0;
    int next_lane = 0;
    hipStream_t aux = nullptr;   // kzg_g1_sum: independent of the MSM lanes
    DevBuf aux_in, aux_pts, aux_out;
    bool profiling = false;
    float tms[KZG_T_COUNT] = {0};  // stage times of the last completed hot-path call
};

namespace {

int fail(kzg_ctx* ctx, int code, const std::string& msg) {
    if (ctx) ctx->err = msg;
    return code;
}
#define HIPCHK(ctx, expr)                                                                                   \
    do {                                                                                                    \
        hipError_t _e = (expr);                                                                             \
        if (_e != hipSuccess)                                                                               \
            return fail(ctx, _e == hipErrorOutOfMemory ? KZG_E_NOMEM : KZG_E_HIP,                           \
                        std::string(#expr) + ": " + hipGetErrorString(_e));                                 \
    } while (0)

hipEvent_t prof_event(MsmLane& L) {
    if (L.ev_used == L.ev_pool.size()) {
        hipEvent_t e;
        (void)hipEventCreate(&e);
        L.ev_pool.push_back(e);
    }
    return L.ev_pool[L.ev_used++];
}
struct Span {
    MsmLane* lane = nullptr;
    int idx = -1;
    hipStream_t stream;
    Span(kzg_ctx* c, int stage, hipStream_t st = nullptr, int li = 0) : stream(st ? st : c->stream) {
        if (!c->profiling) return;
        lane = &c->lane[li];
        StageSpan s{stage, prof_event(*lane), prof_event(*lane)};
        (void)hipEventRecord(s.a, stream);
        lane->spans.push_back(s);
        idx = (int)lane->spans.size() - 1;
    }
    ~Span() {
        if (idx >= 0) (void)hipEventRecord(lane->spans[idx].b, stream);
    }
};
// opens the KZG_T_TOTAL span of the call on lane li; prof_close() ends it just before the last copy-back
void prof_begin(kzg_ctx* ctx, int li = 0) {
    MsmLane& L = ctx->lane[li];
    L.spans.clear();
    L.ev_used = 0;
    if (!ctx->profiling) return;
    StageSpan s{KZG_T_TOTAL, prof_event(L), prof_event(L)};
    (void)hipEventRecord(s.a, L.stream);
    L.spans.push_back(s);
}
void prof_close(kzg_ctx* ctx, int li = 0) {
    MsmLane& L = ctx->lane[li];
    if (ctx->profiling && !L.spans.empty() && L.spans[0].stage == KZG_T_TOTAL) (void)hipEventRecord(L.spans[0].b, L.stream);
}
void prof_end(kzg_ctx* ctx, int li = 0) {  // lane stream already synchronised
    MsmLane& L = ctx->lane[li];
    if (L.spans.empty()) return;
    for (float& t : ctx->tms) t = 0.f;
    for (auto& s : L.spans) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, s.a, s.b) == hipSuccess) ctx->tms[s.stage] += ms;
    }
    L.spans.clear();
}

int choose_window(uint64_t T) {
    int lg = 0;
    while (((uint64_t)1 << (lg + 1)) <= T) lg++;
    // measured on MI355X (bench.py --window sweep): the bucket tree costs ~log2(B) dependent point additions of
    // latency, the accumulate n*ceil(256/c) mixed additions of throughput
    if (lg <= 9) return 8;
    if (lg <= 11) return 10;
    if (lg <= 13) return 12;
    if (lg <= 15) return 14;
    if (lg <= 19) return 16;
    if (lg <= 23) return 20;
    return 22;
}
// nwin = ceil(256/c) windows of width base or base+1 (256 = nwin*base + extra): the widest is <= c bits
void set_window(kzg_ctx* ctx, int c) {
    const int nwin = (256 + c - 1) / c, base = 256 / nwin, extra = 256 % nwin;
    ctx->nwin = ctx->lay.nwin = nwin;
    int off = 0;
    for (int w = 0; w < nwin; w++) {
        ctx->lay.off[w] = (uint16_t)off;
        off += base + (w < extra ? 1 : 0);
    }
    ctx->lay.off[nwin] = 256;
    ctx->c = base + (extra ? 1 : 0);
    ctx->nbuckets = 1u << (ctx->c - 1);
}
// sorted entries per accumulate lane.  Large MSMs (throughput-bound): the grid is a whole number of "rounds" of 131072
// lanes (2 waves per SIMD on 256 CUs: the second wave hides the point loads) so that the last round is not a partially
// filled tail; chunks stay <= 512 entries.  Small MSMs (latency-bound: one wave already saturates a SIMD's integer
// issue, ~10.5 us per mixed addition): 65536 lanes = one wave per SIMD, which halves the number of carries to fold.
int pick_chunk(uint64_t entries) {
    const uint64_t lanes = 131072;
    if (entries <= lanes * 16) {
        const uint64_t k = (entries + lanes / 2 - 1) / (lanes / 2);
        return (int)(k < 8 ? 8 : k);
    }
    const uint64_t rounds = (entries + lanes * 512 - 1) / (lanes * 512);
    const uint64_t k = (entries + lanes * rounds - 1) / (lanes * rounds);
    return (int)k;
}
int ilog2_exact(uint64_t n) {
    if (!n || (n & (n - 1))) return -1;
    int l = 0;
    while (((uint64_t)1 << l) < n) l++;
    return l;
}

// ---- the MSM pipeline on device-resident scalars -> one XYZZ point at out_xyzz (device), on lane `li`.
// With scalars2 != null: TWO MSMs over the same n points in one pass (the commitment and the opening of one row):
// set b is sorted into bucket set b, the sort / accumulate / fold / tree kernels simply see twice the buckets, the
// tree stops at two roots and out_xyzz[0..1] receive the two sums.  One kernel sequence, one latency-bound tail.
int msm_core(kzg_ctx* ctx, int li, const uint32_t* scalars, int mont, uint64_t n, uint64_t srs_offset,
             g1_xyzz_t* out_xyzz, const uint32_t* scalars2 = nullptr, int mont2 = 0) {
    MsmLane& L = ctx->lane[li];
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
    HIPCHK(ctx, L.rank.ensure(entries * 8));          // partitioned (key_low, value) pairs
    HIPCHK(ctx, L.sorted.ensure(entries * 4));
    HIPCHK(ctx, L.hist.ensure(16384 * 4));
    HIPCHK(ctx, L.offsets.ensure((B + 1) * 4));
    HIPCHK(ctx, L.bufA.ensure(B * sizeof(g1_xyzz_t)));
    HIPCHK(ctx, L.bufB.ensure(B * sizeof(g1_xyzz_t) / 2 + 4096));   // level arrays: n/2^L nodes x L components <= B/2
    HIPCHK(ctx, L.bufC.ensure(B * sizeof(g1_xyzz_t) / 2 + 4096));
    HIPCHK(ctx, L.carries.ensure((size_t)nchunks * sizeof(g1_xyzz_t)));
    HIPCHK(ctx, L.carry_key.ensure((size_t)nchunks * 4));
    uint32_t* max_len_d = ctx->flags + 2 + li;
    uint32_t* max_len_h = reinterpret_cast<uint32_t*>(ctx->host_pin + 32) + li;
    {
        Span sp(ctx, KZG_T_DIGITS, s, li);
        HIPCHK(ctx, hipMemsetAsync(L.bufA.p, 0, B * sizeof(g1_xyzz_t), s));
        launch_msm_sort(s, sh, scalars, mont, scalars2, mont2, L.hist.as<uint32_t>(), L.rank.as<uint2>(),
                        L.offsets.as<uint32_t>(), L.sorted.as<uint32_t>());
        // the longest run of carries decides how many fold steps are launched; it depends on the offsets only, so
        // its 4-byte read-back completes while the accumulate kernel runs and costs no bubble
        HIPCHK(ctx, hipMemsetAsync(max_len_d, 0, 4, s));
        launch_fold_maxlen(s, L.offsets.as<uint32_t>(), sh.nbuckets, (uint32_t)sh.chunk, max_len_d);
        HIPCHK(ctx, hipMemcpyAsync(max_len_h, max_len_d, 4, hipMemcpyDeviceToHost, s));
        HIPCHK(ctx, hipEventRecord(L.ev_sorted, s));
    }
    {
        Span sp(ctx, KZG_T_ACCUMULATE, s, li);
        launch_msm_accumulate(s, sh, ctx->table.as<g1_affine_t>(), L.offsets.as<uint32_t>(), L.sorted.as<uint32_t>(),
                              L.bufA.as<g1_xyzz_t>(), L.carries.as<g1_xyzz_t>(), L.carry_key.as<uint32_t>(), nchunks);
    }
    HIPCHK(ctx, hipEventSynchronize(L.ev_sorted));
    {
        Span sp(ctx, KZG_T_FIXUP, s, li);
        for (uint32_t d = 1; d < *max_len_h; d <<= 1)
            launch_fold_step(s, L.offsets.as<uint32_t>(), L.carry_key.as<uint32_t>(), (uint32_t)sh.chunk, nchunks, d,
                             L.carries.as<g1_xyzz_t>());
        launch_fold_heads(s, L.offsets.as<uint32_t>(), L.carry_key.as<uint32_t>(), (uint32_t)sh.chunk, nchunks,
                          L.carries.as<g1_xyzz_t>(), L.bufA.as<g1_xyzz_t>());
    }
    // three buffers in rotation: a level reads its own array and the P array of the level below, writes the next
    g1_xyzz_t* in = L.bufA.as<g1_xyzz_t>();
    g1_xyzz_t* prev = L.bufC.as<g1_xyzz_t>();
    g1_xyzz_t* out = L.bufB.as<g1_xyzz_t>();
    {
        Span sp(ctx, KZG_T_TREE, s, li);
        uint32_t n_in = sh.nbuckets;
        for (int level = 0; n_in > (uint32_t)nbatch; level++, n_in >>= 1) {
            launch_msm_tree_level(s, in, prev, out, n_in, level);
            g1_xyzz_t* recycled = prev;
            prev = in;
            in = out;
            out = recycled;
        }
    }
    {
        Span sp(ctx, KZG_T_FINAL, s, li);
        launch_msm_final(s, in, prev, ctx->c - 1, nbatch, out_xyzz);
    }
    HIPCHK(ctx, hipGetLastError());
    return KZG_OK;
}

// every entry point except kzg_msm_submit / kzg_msm_wait / kzg_g1_sum shares lane 0 and the context buffers
int need_idle(kzg_ctx* ctx) {
    for (const MsmLane& L : ctx->lane)
        if (L.busy) return fail(ctx, KZG_E_BUSY, "an MSM ticket is outstanding: call kzg_msm_wait first");
    return KZG_OK;
}
int need_srs(kzg_ctx* ctx) {
    if (!ctx->table.p || !ctx->stride) return fail(ctx, KZG_E_ARG, "no SRS resident: call kzg_load_srs / kzg_gen_srs");
    return KZG_OK;
}
int clear_flags(kzg_ctx* ctx) {
    HIPCHK(ctx, hipMemsetAsync(ctx->flags, 0, 16, ctx->stream));
    return KZG_OK;
}
// results staged in host_pin: [0..16) flags, then payload
int finish(kzg_ctx* ctx) {
    prof_close(ctx);
    HIPCHK(ctx, hipMemcpyAsync(ctx->host_pin, ctx->flags, 16, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    prof_end(ctx);
    const uint32_t* f = reinterpret_cast<const uint32_t*>(ctx->host_pin);
    if (f[0]) return fail(ctx, KZG_E_SCALAR, "non-canonical Fr scalar (>= r)");
    if (f[1]) return fail(ctx, KZG_E_POINT, "G1 input not reduced or not on the curve");
    return KZG_OK;
}
int ensure_twiddles(kzg_ctx* ctx, int log_n, int inverse, uint32_t** tw, uint32_t** invn) {
    auto& m = inverse ? ctx->tw_inv : ctx->tw_fwd;
    if (log_n >= 1 && !m.count(log_n)) {
        DevBuf b;
        HIPCHK(ctx, b.ensure(((size_t)1 << (log_n - 1)) * 32));
        launch_fr_twiddles(ctx->stream, b.as<uint32_t>(), log_n, inverse);
        m[log_n] = b;
    }
    *tw = log_n >= 1 ? m[log_n].as<uint32_t>() : nullptr;
    if (invn) {
        if (!ctx->inv_n.count(log_n)) {
            DevBuf b;
            HIPCHK(ctx, b.ensure(32));
            launch_fr_inv_pow2(ctx->stream, b.as<uint32_t>(), log_n);
            ctx->inv_n[log_n] = b;
        }
        *invn = ctx->inv_n[log_n].as<uint32_t>();
    }
    return KZG_OK;
}
// coefficients (Montgomery) of the row; returns pointer in *coeffs.  row_dev: Montgomery-form row.
int row_to_coeffs(kzg_ctx* ctx, const uint32_t* row_dev, uint64_t T, int evaluation_form, const uint32_t** coeffs) {
    if (!evaluation_form || T == 1) {
        *coeffs = row_dev;
        return KZG_OK;
    }
    int lg = ilog2_exact(T);
    if (lg < 0) return fail(ctx, KZG_E_ARG, "evaluation-form row length must be a power of two");
    uint32_t *tw, *invn;
    int rc = ensure_twiddles(ctx, lg, 1, &tw, &invn);
    if (rc) return rc;
    HIPCHK(ctx, ctx->coeffB.ensure(T * 32));
    Span sp(ctx, KZG_T_NTT);
    launch_fr_ntt(ctx->stream, row_dev, ctx->coeffB.as<uint32_t>(), lg, tw, invn);
    *coeffs = ctx->coeffB.as<uint32_t>();
    return KZG_OK;
}
int check_worker(kzg_ctx* ctx, uint32_t i, uint64_t T) {
    int rc = need_srs(ctx);
    if (rc) return rc;
    if (T == 0) return fail(ctx, KZG_E_ARG, "empty polynomial");
    if (T > ctx->T) return fail(ctx, KZG_E_ARG, "polynomial longer than the worker's SRS slice");
    if ((uint64_t)i * ctx->T + ctx->T > ctx->stride) return fail(ctx, KZG_E_ARG, "worker index outside the resident SRS");
    return KZG_OK;
}
// upload BE scalars to `dst` (device limbs); dst must hold n*32 bytes
int upload_fr(kzg_ctx* ctx, const uint8_t* be32, uint64_t n, uint32_t* dst, int to_mont) {
    if (!n) return KZG_OK;
    HIPCHK(ctx, ctx->in_be.ensure(n * 32));
    Span sp(ctx, KZG_T_DECODE);
    HIPCHK(ctx, hipMemcpyAsync(ctx->in_be.p, be32, n * 32, hipMemcpyHostToDevice, ctx->stream));
    launch_fr_from_be(ctx->stream, ctx->in_be.as<uint8_t>(), dst, n, to_mont, ctx->flags);
    return KZG_OK;
}

// commit and/or open on a device-resident Montgomery row.  With both requested:
//  * rows up to 2^18 (latency-bound: dozens of small dependent kernels): the commitment MSM(U_i, f) and the opening
//    MSM(U_i, q) run as ONE batched pass over the slice's window tables (msm_core with two scalar sets) -- one sort,
//    one accumulate launch, one bucket tree with two roots, one shared inversion: a single tail instead of two;
//  * longer rows (throughput-bound): the opening (evaluation, quotient, MSM) runs on lane 1 concurrently with the
//    commitment MSM on lane 0 -- they share only the read-only coefficients -- so that each one's sort and tail hide
//    under the other's accumulate.  Profiling keeps everything on lane 0 so that stage times stay attributable.
#ifndef KZG_BATCHED_ROW_MAX
#define KZG_BATCHED_ROW_MAX ((uint64_t)1 << 18)
#endif
int commit_open_dev(kzg_ctx* ctx, uint32_t i, const uint32_t* row_dev, uint64_t T, int evaluation_form,
                    const uint8_t* alpha_be32, uint8_t* out_c48, uint8_t* out_eval32, uint8_t* out_p48) {
    hipStream_t s = ctx->stream;
    const uint32_t* coeffs = nullptr;
    int rc = row_to_coeffs(ctx, row_dev, T, evaluation_form, &coeffs);
    if (rc) return rc;
    HIPCHK(ctx, ctx->res.ensure(4 * sizeof(g1_xyzz_t)));
    HIPCHK(ctx, ctx->small.ensure(1024));
    uint8_t* small = ctx->small.as<uint8_t>();  // [0,48) commitment [64,112) proof [128,160) eval be [192..) alpha/y limbs
    const uint64_t offset = (uint64_t)i * ctx->T;
    g1_xyzz_t* res = ctx->res.as<g1_xyzz_t>();
    const bool both = out_c48 && out_p48;
    const bool batched = both && T <= KZG_BATCHED_ROW_MAX;
    const int lo = (both && !batched && !ctx->profiling) ? 1 : 0;  // lane of the opening
    hipStream_t so = ctx->lane[lo].stream;
    if (lo) {
        HIPCHK(ctx, hipEventRecord(ctx->ev_coeffs, s));
        HIPCHK(ctx, hipStreamWaitEvent(so, ctx->ev_coeffs, 0));
    }
    if (out_c48 && !batched) {
        rc = msm_core(ctx, 0, coeffs, 1, T, offset, res);
        if (rc) return rc;
        if (!out_p48) {  // commit only; with an opening the two points share one inversion below
            Span sp(ctx, KZG_T_FINAL);
            launch_g1_compress(s, res, small);
        }
    }
    if (out_p48) {
        uint32_t* alpha_m = reinterpret_cast<uint32_t*>(small + 192);
        uint32_t* y_m = reinterpret_cast<uint32_t*>(small + 256);
        HIPCHK(ctx, hipMemcpyAsync(small + 320, alpha_be32, 32, hipMemcpyHostToDevice, so));
        launch_fr_from_be(so, small + 320, alpha_m, 1, 1, ctx->flags);
        const uint64_t nchunks = (T + 3) / 4;
        HIPCHK(ctx, ctx->hbuf.ensure((nchunks + (nchunks >> 3) + 64) * 32));
        HIPCHK(ctx, ctx->hnext.ensure((nchunks + (nchunks >> 3) + 64) * 32));
        HIPCHK(ctx, ctx->qbuf.ensure(T * 32));
        {
            Span sp(ctx, KZG_T_POLY, so);
            launch_poly_open(so, coeffs, T, alpha_m, ctx->hbuf.as<uint32_t>(), ctx->hnext.as<uint32_t>(), y_m,
                             ctx->qbuf.as<uint32_t>());
            launch_fr_to_be(so, y_m, small + 128, 1, 1);
        }
        if (batched) {
            // the quotient has T - 1 coefficients; a zero in slot T - 1 lets it ride as a second length-T scalar set
            HIPCHK(ctx, hipMemsetAsync(ctx->qbuf.as<uint32_t>() + 8 * (T - 1), 0, 32, s));
            rc = msm_core(ctx, 0, coeffs, 1, T, offset, res, ctx->qbuf.as<uint32_t>(), 0);
        } else {
            rc = msm_core(ctx, lo, ctx->qbuf.as<uint32_t>(), 0, T - 1, offset, res + 1);
        }
        if (rc) return rc;
        if (lo) {
            HIPCHK(ctx, hipEventRecord(ctx->lane[lo].ev_done, so));
            HIPCHK(ctx, hipStreamWaitEvent(s, ctx->lane[lo].ev_done, 0));
        }
        Span sp(ctx, KZG_T_FINAL);
        if (out_c48) launch_g1_compress_pair(s, res, res + 1, small, small + 64);
        else launch_g1_compress(s, res + 1, small + 64);
    }
    HIPCHK(ctx, hipMemcpyAsync(ctx->host_pin + 64, small, 192, hipMemcpyDeviceToHost, s));
    rc = finish(ctx);
    if (rc) return rc;
    if (out_c48) memcpy(out_c48, ctx->host_pin + 64, 48);
    if (out_p48) {
        memcpy(out_p48, ctx->host_pin + 64 + 64, 48);
        memcpy(out_eval32, ctx->host_pin + 64 + 128, 32);
    }
    return KZG_OK;
}

int alloc_table(kzg_ctx* ctx, uint64_t n_points, int scale, int mscale) {
    if (mscale < 0 || scale < mscale || scale - mscale > 30) return fail(ctx, KZG_E_ARG, "bad scale / machines_scale");
    const uint64_t T = (uint64_t)1 << (scale - mscale);
    if (n_points == 0 || n_points % T) return fail(ctx, KZG_E_ARG, "SRS length must be a whole number of worker slices");
    set_window(ctx, ctx->c_user ? ctx->c_user : choose_window(T));
    if ((uint64_t)ctx->nwin * n_points >= ((uint64_t)1 << 31))
        return fail(ctx, KZG_E_ARG, "SRS x windows exceeds 2^31 table entries");
    ctx->table.release();
    HIPCHK(ctx, ctx->table.ensure((size_t)ctx->nwin * n_points * sizeof(g1_affine_t)));
    ctx->stride = n_points; ctx->T = T; ctx->scale = scale; ctx->mscale = mscale;
    return KZG_OK;
}
int precompute_tables(kzg_ctx* ctx) {
    const uint64_t tile = ctx->stride < ((uint64_t)1 << 20) ? ctx->stride : ((uint64_t)1 << 20);
    DevBuf tmp;
    HIPCHK(ctx, tmp.ensure((size_t)(ctx->nwin - 1) * tile * sizeof(g1_xyzz_t) + 256));
    for (uint64_t first = 0; first < ctx->stride; first += tile) {
        uint64_t cnt = ctx->stride - first < tile ? ctx->stride - first : tile;
        launch_srs_precompute(ctx->stream, ctx->table.as<g1_affine_t>(), ctx->stride, first, cnt, ctx->lay,
                              tmp.as<g1_xyzz_t>());
    }
    hipError_t e = hipStreamSynchronize(ctx->stream);
    tmp.release();
    HIPCHK(ctx, e);
    HIPCHK(ctx, hipGetLastError());
    return KZG_OK;
}

// ---- unit-op test kernels
// Fr: saturated 32-bit Montgomery (field.hip.h).  Fp: op 0 mul / 1 add / 2 sub / 4 sqr on the 28-bit-limb working
// representation (fp28.hip.h); op 3 = the plain-C++ 12 x 32-bit CIOS reference product.
__global__ void __launch_bounds__(256) k_test_fr(int op, const uint8_t* a_be, const uint8_t* b_be, uint8_t* out_be,
                                                  uint64_t n) {
    uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    fr_t a, b, r;
    limbs_from_be<8>(a.l, a_be + 32 * j);
    limbs_from_be<8>(b.l, b_be + 32 * j);
    f_to_mont(a, a);
    f_to_mont(b, b);
    if (op == 0) f_mul(r, a, b);
    else if (op == 1) f_add(r, a, b);
    else if (op == 2) f_sub(r, a, b);
    else if (op == 3) f_mul_inline(r, a, b);
    else f_mul(r, a, a);
    f_from_mont(r, r);
    limbs_to_be<8>(out_be + 32 * j, r.l);
}
__global__ void __launch_bounds__(256) k_test_fp(int op, const uint8_t* a_be, const uint8_t* b_be, uint8_t* out_be,
                                                  uint64_t n) {
    uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    if (op == 3) {
        fp32_t a, b, r;
        limbs_from_be<12>(a.l, a_be + 48 * j);
        limbs_from_be<12>(b.l, b_be + 48 * j);
        f_to_mont(a, a);
        f_to_mont(b, b);
        f_mul_inline(r, a, b);
        f_from_mont(r, r);
        limbs_to_be<12>(out_be + 48 * j, r.l);
        return;
    }
    fp_t a, b, r;
    fp_from_be48(a, a_be + 48 * j);
    fp_from_be48(b, b_be + 48 * j);
    if (op == 0) fp_mul(r, a, b);
    else if (op == 1) fp_add(r, a, b);
    else if (op == 2) fp_sub4(r, a, b);
    else fp_sqr(r, a);
    fp_to_be48(out_be + 48 * j, r);
}
__global__ void __launch_bounds__(256) k_test_g1(int op, const uint8_t* a_be, const uint8_t* b_be, uint8_t* out_be,
                                                  uint64_t n) {
    uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    g1_aff28 a, b, o;
    fp_from_be48(a.x, a_be + 96 * j); fp_from_be48(a.y, a_be + 96 * j + 48);
    fp_from_be48(b.x, b_be + 96 * j); fp_from_be48(b.y, b_be + 96 * j + 48);
    g1_xyzz_t pa, pb, r, t;
    g1_from_aff(pa, a);
    g1_from_aff(pb, b);
    if (op == 0) { r = pa; g1_madd_checked(r, b); }
    else if (op == 1) { g1_dbl(t, pa); g1_add(r, t, pb); }
    else if (op == 2) { g1_dbl(r, pa); }
    else if (op == 3) { g1_dbl(t, pa); g1_dbl(r, t); }
    else {  // long dependent chain: ((a + b) + b + ... ) exercising the class invariants across many mixed adds
        r = pa;
        for (int k = 0; k < 40; k++) g1_madd_checked<true>(r, (k & 1) ? a : b);  // the inlined-product variant
    }
    g1_to_aff(o, r);
    fp_to_be48(out_be + 96 * j, o.x);
    fp_to_be48(out_be + 96 * j + 48, o.y);
}

const char B64[] = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789+/";
int8_t b64_rev[256];
bool b64_init_done = false;
void b64_init() {
    if (b64_init_done) return;
    memset(b64_rev, -1, sizeof(b64_rev));
    for (int i = 0; i < 64; i++) b64_rev[(uint8_t)B64[i]] = (int8_t)i;
    b64_init_done = true;
}

}  // namespace

// =====================================================================================================
extern "C" {

const char* kzg_version(void) { return KZG_VERSION; }

int kzg_create(int device_id, kzg_ctx** out) {
    if (!out) return KZG_E_ARG;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device_id < 0 || device_id >= count) return KZG_E_HIP;
    if (hipSetDevice(device_id) != hipSuccess) return KZG_E_HIP;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) != hipSuccess) return KZG_E_HIP;
    if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0) return KZG_E_HIP;  // built for gfx950 only
    kzg_ctx* ctx = new kzg_ctx();
    ctx->device = device_id;
    bool ok = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) == hipSuccess &&
              hipMalloc((void**)&ctx->flags, 16) == hipSuccess &&
              hipHostMalloc((void**)&ctx->host_pin, 4096, hipHostMallocDefault) == hipSuccess &&
              hipEventCreateWithFlags(&ctx->ev_coeffs, hipEventDisableTiming) == hipSuccess &&
              hipStreamCreateWithFlags(&ctx->aux, hipStreamNonBlocking) == hipSuccess;
    for (int l = 0; ok && l < N_LANES; l++) {
        MsmLane& L = ctx->lane[l];
        if (l == 0) L.stream = ctx->stream;
        else ok = hipStreamCreateWithFlags(&L.stream, hipStreamNonBlocking) == hipSuccess;
        ok = ok && hipEventCreateWithFlags(&L.ev_sorted, hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&L.ev_done, hipEventDisableTiming) == hipSuccess;
    }
    if (!ok) {
        kzg_destroy(ctx);
        return KZG_E_HIP;
    }
    *out = ctx;
    return KZG_OK;
}

void kzg_destroy(kzg_ctx* ctx) {
    if (!ctx) return;
    {
        std::lock_guard<std::mutex> lk(ctx->mu);
        (void)hipSetDevice(ctx->device);
        for (MsmLane& L : ctx->lane)
            if (L.stream) (void)hipStreamSynchronize(L.stream);
        DevBuf* bufs[] = {&ctx->table, &ctx->in_be, &ctx->scal, &ctx->res, &ctx->coeffA,
                          &ctx->coeffB, &ctx->qbuf, &ctx->hbuf, &ctx->hnext, &ctx->small, &ctx->out_be};
        for (DevBuf* b : bufs) b->release();
        for (MsmLane& L : ctx->lane) {
            for (DevBuf* b : {&L.rank, &L.sorted, &L.hist, &L.offsets, &L.bufA, &L.bufB, &L.bufC, &L.carries, &L.carry_key})
                b->release();
            L.res.release();
            L.small.release();
            for (hipEvent_t e : L.ev_pool) (void)hipEventDestroy(e);
            if (L.ev_sorted) (void)hipEventDestroy(L.ev_sorted);
            if (L.ev_done) (void)hipEventDestroy(L.ev_done);
            if (L.stream && L.stream != ctx->stream) (void)hipStreamDestroy(L.stream);
        }
        for (auto& b : ctx->slot) b.release();
        for (auto* m : {&ctx->tw_fwd, &ctx->tw_inv, &ctx->inv_n})
            for (auto& kv : *m) kv.second.release();
        if (ctx->ev_coeffs) (void)hipEventDestroy(ctx->ev_coeffs);
        for (DevBuf* b : {&ctx->aux_in, &ctx->aux_pts, &ctx->aux_out}) b->release();
        if (ctx->aux) {
            (void)hipStreamSynchronize(ctx->aux);
            (void)hipStreamDestroy(ctx->aux);
        }
        if (ctx->flags) (void)hipFree(ctx->flags);
        if (ctx->host_pin) (void)hipHostFree(ctx->host_pin);
        if (ctx->stage_host) (void)hipHostFree(ctx->stage_host);
        if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    }
    delete ctx;
}

const char* kzg_last_error(kzg_ctx* ctx) { return ctx ? ctx->err.c_str() : "null ctx"; }

int kzg_set_window(kzg_ctx* ctx, int c) {
    if (!ctx) return KZG_E_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    if (c != 0 && (c < 4 || c > 22)) return fail(ctx, KZG_E_ARG, "window bits must be 0 (auto) or in [4, 22]");
    if (ctx->table.p) return fail(ctx, KZG_E_ARG, "window must be set before the SRS is loaded");
    ctx->c_user = c;
    return KZG_OK;
}
int kzg_get_window(kzg_ctx* ctx) { return ctx ? ctx->c : 0; }
int kzg_get_window_layout(kzg_ctx* ctx, int32_t* out_offsets, int max) {
    if (!ctx || !out_offsets || !ctx->c) return KZG_E_ARG;
    for (int w = 0; w <= ctx->nwin && w < max; w++) out_offsets[w] = ctx->lay.off[w];
    return ctx->nwin;
}
uint64_t kzg_srs_points(kzg_ctx* ctx) { return ctx ? ctx->stride : 0; }

static int load_srs_common(kzg_ctx* ctx, const uint8_t* data, uint64_t n_points, int scale, int machines_scale,
                           bool compressed) {
    if (!ctx || !data) return KZG_E_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (int busy = need_idle(ctx)) return busy;
    int rc = alloc_table(ctx, n_points, scale, machines_scale);
    if (rc) return rc;
    rc = clear_flags(ctx);
    if (rc) return rc;
    const uint64_t tile = (uint64_t)1 << 20;
    const size_t rec = compressed ? 48 : 96;
    HIPCHK(ctx, ctx->in_be.ensure((n_points < tile ? n_points : tile) * rec));
    for (uint64_t first = 0; first < n_points; first += tile) {
        uint64_t cnt = n_points - first < tile ? n_points - first : tile;
        HIPCHK(ctx, hipMemcpyAsync(ctx->in_be.p, data + rec * first, cnt * rec, hipMemcpyHostToDevice, ctx->stream));
        if (compressed)
            launch_srs_from_c48(ctx->stream, ctx->in_be.as<uint8_t>(), ctx->table.as<g1_affine_t>() + first, cnt,
                                ctx->flags + 1);
        else
            launch_srs_from_be96(ctx->stream, ctx->in_be.as<uint8_t>(), ctx->table.as<g1_affine_t>() + first, cnt,
                                 ctx->flags + 1);
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    }
    rc = finish(ctx);
    if (rc) {
        ctx->table.release();
        ctx->stride = 0;
        return rc;
    }
    return precompute_tables(ctx);
}
int kzg_load_srs(kzg_ctx* ctx, const uint8_t* g1_affine_be96, uint64_t n_points, int scale, int machines_scale) {
    return load_srs_common(ctx, g1_affine_be96, n_points, scale, machines_scale, false);
}
int kzg_load_srs_compressed(kzg_ctx* ctx, const uint8_t* g1_c48, uint64_t n_points, int scale, int machines_scale) {
    return load_srs_common(ctx, g1_c48, n_points, scale, machines_scale, true);
}

int kzg_gen_srs(kzg_ctx* ctx, const uint8_t tau_be32[32], const uint8_t* s0_be32, uint32_t n_slices, int scale,
                int machines_scale) {
    if (!ctx || !tau_be32 || !s0_be32 || !n_slices) return KZG_E_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (int busy = need_idle(ctx)) return busy;
    if (machines_scale < 0 || scale < machines_scale || scale - machines_scale > 30)
        return fail(ctx, KZG_E_ARG, "bad scale / machines_scale");
    const uint64_t T = (uint64_t)1 << (scale - machines_scale);
    int rc = alloc_table(ctx, (uint64_t)n_slices * T, scale, machines_scale);
    if (rc) return rc;
    rc = clear_flags(ctx);
    if (rc) return rc;
    DevBuf gtab, tmp, sc;
    HIPCHK(ctx, gtab.ensure(32 * 255 * sizeof(g1_affine_t)));
    const uint64_t tile = T < ((uint64_t)1 << 20) ? T : ((uint64_t)1 << 20);
    HIPCHK(ctx, tmp.ensure(tile * (sizeof(g1_xyzz_t) + 32) + 256));
    HIPCHK(ctx, sc.ensure(((size_t)n_slices + 1) * 32 + 64));
    // tau and the per-slice factors, Montgomery form, on device
    uint32_t* tau_m = sc.as<uint32_t>();
    HIPCHK(ctx, ctx->in_be.ensure(((size_t)n_slices + 1) * 32));
    HIPCHK(ctx, hipMemcpyAsync(ctx->in_be.p, tau_be32, 32, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(ctx->in_be.as<uint8_t>() + 32, s0_be32, (size_t)n_slices * 32, hipMemcpyHostToDevice,
                               ctx->stream));
    launch_fr_from_be(ctx->stream, ctx->in_be.as<uint8_t>(), tau_m, (uint64_t)n_slices + 1, 1, ctx->flags);
    for (uint32_t k = 0; k < n_slices; k++) {
        for (uint64_t first = 0; first < T; first += tile) {
            uint64_t cnt = T - first < tile ? T - first : tile;
            launch_srs_generate(ctx->stream, ctx->table.as<g1_affine_t>() + (uint64_t)k * T + first, cnt, first, tau_m,
                                tau_m + 8 * (1 + (uint64_t)k), gtab.as<g1_affine_t>(), tmp.as<g1_xyzz_t>(),
                                k == 0 && first == 0);
        }
    }
    rc = finish(ctx);
    gtab.release(); tmp.release(); sc.release();
    if (rc) return rc;
    HIPCHK(ctx, hipGetLastError());
    return precompute_tables(ctx);
}

static int srs_read_common(kzg_ctx* ctx, int w, uint64_t first, uint64_t count, uint8_t* out, bool compressed) {
    if (!ctx || !out) return KZG_E_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (int busy = need_idle(ctx)) return busy;
    int rc = need_srs(ctx);
    if (rc) return rc;
    if (w < 0 || w >= ctx->nwin || first + count > ctx->stride) return fail(ctx, KZG_E_ARG, "srs_read out of range");
    if (!count) return KZG_OK;
    const size_t rec = compressed ? 48 : 96;
    HIPCHK(ctx, ctx->out_be.ensure(count * rec));
    const g1_affine_t* src = ctx->table.as<g1_affine_t>() + (uint64_t)w * ctx->stride + first;
    if (compressed) launch_srs_to_c48(ctx->stream, src, ctx->out_be.as<uint8_t>(), count);
    else launch_srs_to_be96(ctx->stream, src, ctx->out_be.as<uint8_t>(), count);
    HIPCHK(ctx, hipMemcpyAsync(out, ctx->out_be.p, count * rec, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return KZG_OK;
}
int kzg_srs_read(kzg_ctx* ctx, int w, uint64_t first, uint64_t count, uint8_t* out_be96) {
    return srs_read_common(ctx, w, first, count, out_be96, false);
}
int kzg_srs_read_compressed(kzg_ctx* ctx, int w, uint64_t first, uint64_t count, uint8_t* out_c48) {
    return srs_read_common(ctx, w, first, count, out_c48, true);
}

static int msm_host_common(kzg_ctx* ctx, const uint8_t* scalars_be32, uint64_t n, uint64_t srs_offset, uint8_t* out,
                           bool partial) {
    if (!ctx || !out || (n && !scalars_be32)) return KZG_E_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (int busy = need_idle(ctx)) return busy;
    int rc = need_srs(ctx);
    if (rc) return rc;
    prof_begin(ctx);
    rc = clear_flags(ctx);
    if (rc) return rc;
    HIPCHK(ctx, ctx->scal.ensure(n * 32 + 32));
    HIPCHK(ctx, ctx->res.ensure(4 * sizeof(g1_xyzz_t)));
    HIPCHK(ctx, ctx->small.ensure(1024));
    rc = upload_fr(ctx, scalars_be32, n, ctx->scal.as<uint32_t>(), 0);
    if (rc) return rc;
    rc = msm_core(ctx, 0, ctx->scal.as<uint32_t>(), 0, n, srs_offset, ctx->res.as<g1_xyzz_t>());
    if (rc) return rc;
    if (partial) {
        launch_xyzz_pack(ctx->stream, ctx->res.as<g1_xyzz_t>(), ctx->small.as<uint32_t>(), 1);
        HIPCHK(ctx, hipMemcpyAsync(ctx->host_pin + 64, ctx->small.p, 192, hipMemcpyDeviceToHost, ctx->stream));
    } else {
        launch_g1_compress(ctx->stream, ctx->res.as<g1_xyzz_t>(), ctx->small.as<uint8_t>());
        HIPCHK(ctx, hipMemcpyAsync(ctx->host_pin + 64, ctx->small.p, 48, hipMemcpyDeviceToHost, ctx->stream));
    }
    rc = finish(ctx);
    if (rc) return rc;
    memcpy(out, ctx->host_pin + 64, partial ? 192 : 48);
    return KZG_OK;
}
int kzg_msm(kzg_ctx* ctx, const uint8_t* scalars_be32, uint64_t n, uint64_t srs_offset, uint8_t out48[48]) {
    return msm_host_common(ctx, scalars_be32, n, srs_offset, out48, false);
}
int kzg_msm_partial(kzg_ctx* ctx, const uint8_t* scalars_be32, uint64_t n, uint64_t srs_offset,
                    uint8_t out_xyzz192[192]) {
    return msm_host_common(ctx, scalars_be32, n, srs_offset, out_xyzz192, true);
}

// runs on its own stream and buffers: legal while MSM tickets are outstanding (a rank sums the gathered partials
// of step i while its step i+1 is already on the GPU).  `on_device`: the partials already sit in device memory
// (the output tensor of the all_gather) and every prior writer has completed.
static int g1_sum_common(kzg_ctx* ctx, const uint8_t* partials, uint32_t count, uint8_t out48[48], bool on_device) {
    if (!ctx || !out48 || (count && !partials)) return KZG_E_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, ctx->aux_in.ensure((size_t)count * 192 + 192));
    HIPCHK(ctx, ctx->aux_pts.ensure(((size_t)count + 2) * sizeof(g1_xyzz_t)));
    HIPCHK(ctx, ctx->aux_out.ensure(64));
    hipStream_t s = ctx->aux;
    g1_xyzz_t* pts = ctx->aux_pts.as<g1_xyzz_t>();
    const uint32_t* src = reinterpret_cast<const uint32_t*>(partials);
    if (!on_device) {
        if (count) HIPCHK(ctx, hipMemcpyAsync(ctx->aux_in.p, partials, (size_t)count * 192, hipMemcpyHostToDevice, s));
        src = ctx->aux_in.as<uint32_t>();
    }
    launch_xyzz_unpack(s, src, pts + 1, count);
    launch_g1_sum(s, pts + 1, count, pts);
    launch_g1_compress(s, pts, ctx->aux_out.as<uint8_t>());
    uint8_t* pin = ctx->host_pin + 1024;
    HIPCHK(ctx, hipMemcpyAsync(pin, ctx->aux_out.p, 48, hipMemcpyDeviceToHost, s));
    HIPCHK(ctx, hipStreamSynchronize(s));
    HIPCHK(ctx, hipGetLastError());
    memcpy(out48, pin, 48);
    return KZG_OK;
}
int kzg_g1_sum(kzg_ctx* ctx, const uint8_t* partials_xyzz192, uint32_t count, uint8_t out48[48]) {
    return g1_sum_common(ctx, partials_xyzz192, count, out48, false);
}
int kzg_g1_sum_dev(kzg_ctx* ctx, const void* dev_partials_xyzz192, uint32_t count, uint8_t out48[48]) {
    return g1_sum_common(ctx, reinterpret_cast<const uint8_t*>(dev_partials_xyzz192), count, out48, true);
}

static int commit_open_host(kzg_ctx* ctx, uint32_t i, const uint8_t* row_be32, uint64_t T, int evaluation_form,
                            const uint8_t* alpha, uint8_t* c48, uint8_t* e32, uint8_t* p48) {
    if (!ctx || !row_be32 || (p48 && (!alpha || !e32))) return KZG_E_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (int busy = need_idle(ctx)) return busy;
    int rc = check_worker(ctx, i, T);
    if (rc) return rc;
    prof_begin(ctx);
    rc = clear_flags(ctx);
    if (rc) return rc;
    HIPCHK(ctx, ctx->coeffA.ensure(T * 32));
    rc = upload_fr(ctx, row_be32, T, ctx->coeffA.as<uint32_t>(), 1);
    if (rc) return rc;
    return commit_open_dev(ctx, i, ctx->coeffA.as<uint32_t>(), T, evaluation_form, alpha, c48, e32, p48);
}
int kzg_commit(kzg_ctx* ctx, uint32_t i, const uint8_t* row_be32, uint64_t T, int evaluation_form,
               uint8_t out_commitment48[48]) {
    if (!out_commitment48) return KZG_E_ARG;
    return commit_open_host(ctx, i, row_be32, T, evaluation_form, nullptr, out_commitment48, nullptr, nullptr);
}
int kzg_open(kzg_ctx* ctx, uint32_t i, const uint8_t* row_be32, uint64_t T, int evaluation_form,
             const uint8_t alpha_be32[32], uint8_t out_eval32[32], uint8_t out_proof48[48]) {
    if (!out_proof48) return KZG_E_ARG;
    return commit_open_host(ctx, i, row_be32, T, evaluation_form, alpha_be32, nullptr, out_eval32, out_proof48);
}
int kzg_commit_open(kzg_ctx* ctx, uint32_t i, const uint8_t* row_be32, uint64_t T, int evaluation_form,
                    const uint8_t alpha_be32[32], uint8_t out_commitment48[48], uint8_t out_eval32[32],
                    uint8_t out_proof48[48]) {
    if (!out_commitment48 || !out_proof48) return KZG_E_ARG;
    return commit_open_host(ctx, i, row_be32, T, evaluation_form, alpha_be32, out_commitment48, out_eval32,
                            out_proof48);
}

static int ntt_dev(kzg_ctx* ctx, uint32_t* data, uint64_t n, int inverse) {  // in place via coeffB
    int lg = ilog2_exact(n);
    if (lg < 0) return fail(ctx, KZG_E_ARG, "NTT length must be a power of two");
    uint32_t *tw = nullptr, *invn = nullptr;
    int rc = ensure_twiddles(ctx, lg, inverse, &tw, inverse ? &invn : nullptr);
    if (rc) return rc;
    HIPCHK(ctx, ctx->coeffB.ensure(n * 32));
    {
        Span sp(ctx, KZG_T_NTT);
        launch_fr_ntt(ctx->stream, data, ctx->coeffB.as<uint32_t>(), lg, tw, inverse ? invn : nullptr);
        HIPCHK(ctx, hipMemcpyAsync(data, ctx->coeffB.p, n * 32, hipMemcpyDeviceToDevice, ctx->stream));
    }
    return KZG_OK;
}
int kzg_ntt(kzg_ctx* ctx, uint8_t* inout_be32, uint64_t n, int inverse) {
    if (!ctx || !inout_be32 || !n) return KZG_E_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (int busy = need_idle(ctx)) return busy;
    prof_begin(ctx);
    int rc = clear_flags(ctx);
    if (rc) return rc;
    HIPCHK(ctx, ctx->coeffA.ensure(n * 32));
    HIPCHK(ctx, ctx->out_be.ensure(n * 32));
    rc = upload_fr(ctx, inout_be32, n, ctx->coeffA.as<uint32_t>(), 1);
    if (rc) return rc;
    rc = ntt_dev(ctx, ctx->coeffA.as<uint32_t>(), n, inverse);
    if (rc) return rc;
    launch_fr_to_be(ctx->stream, ctx->coeffA.as<uint32_t>(), ctx->out_be.as<uint8_t>(), n, 1);
    rc = finish(ctx);
    if (rc) return rc;
    HIPCHK(ctx, hipMemcpy(inout_be32, ctx->out_be.p, n * 32, hipMemcpyDeviceToHost));
    return KZG_OK;
}
int kzg_eval(kzg_ctx* ctx, const uint8_t* coeffs_be32, uint64_t n, const uint8_t x_be32[32], uint8_t out_be32[32]) {
    if (!ctx || !x_be32 || !out_be32 || (n && !coeffs_be32)) return KZG_E_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (int busy = need_idle(ctx)) return busy;
    if (n == 0) {
        memset(out_be32, 0, 32);
        return KZG_OK;
    }
    prof_begin(ctx);
    int rc = clear_flags(ctx);
    if (rc) return rc;
    HIPCHK(ctx, ctx->coeffA.ensure(n * 32));
    HIPCHK(ctx, ctx->small.ensure(1024));
    const uint64_t nchunks = (n + 3) / 4;
    HIPCHK(ctx, ctx->hbuf.ensure((nchunks + (nchunks >> 3) + 64) * 32));
    HIPCHK(ctx, ctx->hnext.ensure((nchunks + (nchunks >> 3) + 64) * 32));
    rc = upload_fr(ctx, coeffs_be32, n, ctx->coeffA.as<uint32_t>(), 1);
    if (rc) return rc;
    uint8_t* small = ctx->small.as<uint8_t>();
    uint32_t* x_m = reinterpret_cast<uint32_t*>(small + 192);
    uint32_t* y_m = reinterpret_cast<uint32_t*>(small + 256);
    HIPCHK(ctx, hipMemcpyAsync(small + 320, x_be32, 32, hipMemcpyHostToDevice, ctx->stream));
    launch_fr_from_be(ctx->stream, small + 320, x_m, 1, 1, ctx->flags);
    launch_poly_open(ctx->stream, ctx->coeffA.as<uint32_t>(), n, x_m, ctx->hbuf.as<uint32_t>(),
                     ctx->hnext.as<uint32_t>(), y_m, nullptr);
    launch_fr_to_be(ctx->stream, y_m, small + 128, 1, 1);
    HIPCHK(ctx, hipMemcpyAsync(ctx->host_pin + 64, small + 128, 32, hipMemcpyDeviceToHost, ctx->stream));
    rc = finish(ctx);
    if (rc) return rc;
    memcpy(out_be32, ctx->host_pin + 64, 32);
    return KZG_OK;
}

int kzg_upload_fr(kzg_ctx* ctx, int slot, const uint8_t* be32, uint64_t n, int to_mont) {
    if (!ctx || slot < 0 || slot >= N_SLOTS || (n && !be32)) return KZG_E_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (int busy = need_idle(ctx)) return busy;
    prof_begin(ctx);
    int rc = clear_flags(ctx);
    if (rc) return rc;
    HIPCHK(ctx, ctx->slot[slot].ensure(n * 32 + 32));
    rc = upload_fr(ctx, be32, n, ctx->slot[slot].as<uint32_t>(), to_mont);
    if (rc) return rc;
    rc = finish(ctx);
    if (rc) return rc;
    ctx->slot_n[slot] = n;
    ctx->slot_mont[slot] = to_mont ? 1 : 0;
    return KZG_OK;
}
// dev_out != null: the 192-byte partial is left in the CALLER's device buffer (e.g. a torch tensor about to enter an
// RCCL all_gather) instead of coming back to the host
static int msm_resident_common(kzg_ctx* ctx, int slot, uint64_t n, uint64_t srs_offset, uint8_t* out, bool partial,
                               void* dev_out = nullptr) {
    if (!ctx || (!out && !dev_out) || slot < 0 || slot >= N_SLOTS) return KZG_E_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (int busy = need_idle(ctx)) return busy;
    int rc = need_srs(ctx);
    if (rc) return rc;
    if (n > ctx->slot_n[slot]) return fail(ctx, KZG_E_ARG, "slot holds fewer scalars than requested");
    prof_begin(ctx);
    rc = clear_flags(ctx);
    if (rc) return rc;
    HIPCHK(ctx, ctx->res.ensure(4 * sizeof(g1_xyzz_t)));
    HIPCHK(ctx, ctx->small.ensure(1024));
    rc = msm_core(ctx, 0, ctx->slot[slot].as<uint32_t>(), ctx->slot_mont[slot], n, srs_offset, ctx->res.as<g1_xyzz_t>());
    if (rc) return rc;
    if (dev_out) {
        launch_xyzz_pack(ctx->stream, ctx->res.as<g1_xyzz_t>(), reinterpret_cast<uint32_t*>(dev_out), 1);
    } else if (partial) {
        launch_xyzz_pack(ctx->stream, ctx->res.as<g1_xyzz_t>(), ctx->small.as<uint32_t>(), 1);
        HIPCHK(ctx, hipMemcpyAsync(ctx->host_pin + 64, ctx->small.p, 192, hipMemcpyDeviceToHost, ctx->stream));
    } else {
        Span sp(ctx, KZG_T_FINAL);
        launch_g1_compress(ctx->stream, ctx->res.as<g1_xyzz_t>(), ctx->small.as<uint8_t>());
        HIPCHK(ctx, hipMemcpyAsync(ctx->host_pin + 64, ctx->small.p, 48, hipMemcpyDeviceToHost, ctx->stream));
    }
    rc = finish(ctx);  // synchronises the stream: dev_out is complete when the call returns
    if (rc) return rc;
    if (out) memcpy(out, ctx->host_pin + 64, partial ? 192 : 48);
    return KZG_OK;
}
int kzg_msm_partial_resident_dev(kzg_ctx* ctx, int slot, uint64_t n, uint64_t srs_offset, void* dev_out_xyzz192) {
    if (!dev_out_xyzz192) return KZG_E_ARG;
    return msm_resident_common(ctx, slot, n, srs_offset, nullptr, true, dev_out_xyzz192);
}
int kzg_msm_resident(kzg_ctx* ctx, int slot, uint64_t n, uint64_t srs_offset, uint8_t out48[48]) {
    return msm_resident_common(ctx, slot, n, srs_offset, out48, false);
}
int kzg_msm_partial_resident(kzg_ctx* ctx, int slot, uint64_t n, uint64_t srs_offset, uint8_t out_xyzz192[192]) {
    return msm_resident_common(ctx, slot, n, srs_offset, out_xyzz192, true);
}
// ---- ticketed MSM: submit returns once the work is queued on a free lane, wait returns the result.  Two lanes, so
// MSM i+1 (sort, accumulate) overlaps the latency-bound tail (fold, bucket tree, final combination, inversion) of
// MSM i.  While a ticket is outstanding only kzg_msm_submit / kzg_msm_wait / kzg_g1_sum may be called on the ctx.
int kzg_msm_submit(kzg_ctx* ctx, int slot, uint64_t n, uint64_t srs_offset, int partial, int* out_ticket) {
    if (!ctx || !out_ticket || slot < 0 || slot >= N_SLOTS) return KZG_E_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int rc = need_srs(ctx);
    if (rc) return rc;
    if (n > ctx->slot_n[slot]) return fail(ctx, KZG_E_ARG, "slot holds fewer scalars than requested");
    int li = -1;
    for (int k = 0; k < N_LANES && li < 0; k++)
        if (!ctx->lane[(ctx->next_lane + k) % N_LANES].busy) li = (ctx->next_lane + k) % N_LANES;
    if (li < 0) return fail(ctx, KZG_E_BUSY, "both MSM lanes hold an outstanding ticket: call kzg_msm_wait first");
    MsmLane& L = ctx->lane[li];
    HIPCHK(ctx, L.res.ensure(sizeof(g1_xyzz_t)));
    HIPCHK(ctx, L.small.ensure(256));
    prof_begin(ctx, li);
    rc = msm_core(ctx, li, ctx->slot[slot].as<uint32_t>(), ctx->slot_mont[slot], n, srs_offset, L.res.as<g1_xyzz_t>());
    if (rc) return rc;
    uint8_t* pin = ctx->host_pin + 256 + 256 * li;
    if (partial) {
        launch_xyzz_pack(L.stream, L.res.as<g1_xyzz_t>(), L.small.as<uint32_t>(), 1);
    } else {
        Span sp(ctx, KZG_T_FINAL, L.stream, li);
        launch_g1_compress(L.stream, L.res.as<g1_xyzz_t>(), L.small.as<uint8_t>());
    }
    prof_close(ctx, li);
    HIPCHK(ctx, hipMemcpyAsync(pin, L.small.p, partial ? 192 : 48, hipMemcpyDeviceToHost, L.stream));
    HIPCHK(ctx, hipEventRecord(L.ev_done, L.stream));
    HIPCHK(ctx, hipGetLastError());
    L.busy = true;
    L.partial = partial != 0;
    ctx->next_lane = (li + 1) % N_LANES;
    *out_ticket = li;
    return KZG_OK;
}
int kzg_msm_wait(kzg_ctx* ctx, int ticket, uint8_t* out) {
    if (!ctx || !out || ticket < 0 || ticket >= N_LANES) return KZG_E_ARG;
    MsmLane& L = ctx->lane[ticket];
    {
        std::lock_guard<std::mutex> lk(ctx->mu);
        if (!L.busy) return fail(ctx, KZG_E_ARG, "no outstanding MSM on this ticket");
        HIPCHK(ctx, hipSetDevice(ctx->device));
    }
    hipError_t e = hipEventSynchronize(L.ev_done);  // not under the lock: another thread may submit meanwhile
    std::lock_guard<std::mutex> lk(ctx->mu);
    L.busy = false;
    HIPCHK(ctx, e);
    prof_end(ctx, ticket);
    memcpy(out, ctx->host_pin + 256 + 256 * ticket, L.partial ? 192 : 48);
    return KZG_OK;
}

int kzg_commit_open_resident(kzg_ctx* ctx, uint32_t i, int slot, uint64_t T, int evaluation_form,
                             const uint8_t alpha_be32[32], uint8_t out_commitment48[48], uint8_t out_eval32[32],
                             uint8_t out_proof48[48]) {
    if (!ctx || slot < 0 || slot >= N_SLOTS || !alpha_be32 || !out_commitment48 || !out_eval32 || !out_proof48)
        return KZG_E_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (int busy = need_idle(ctx)) return busy;
    int rc = check_worker(ctx, i, T);
    if (rc) return rc;
    if (T > ctx->slot_n[slot] || !ctx->slot_mont[slot])
        return fail(ctx, KZG_E_ARG, "slot must hold >= T Montgomery-form elements (kzg_upload_fr(.., to_mont=1))");
    prof_begin(ctx);
    rc = clear_flags(ctx);
    if (rc) return rc;
    return commit_open_dev(ctx, i, ctx->slot[slot].as<uint32_t>(), T, evaluation_form, alpha_be32, out_commitment48,
                           out_eval32, out_proof48);
}
int kzg_ntt_resident(kzg_ctx* ctx, int slot, uint64_t n, int inverse) {
    if (!ctx || slot < 0 || slot >= N_SLOTS || !n) return KZG_E_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (int busy = need_idle(ctx)) return busy;
    if (n > ctx->slot_n[slot] || !ctx->slot_mont[slot]) return fail(ctx, KZG_E_ARG, "slot must hold >= n Montgomery elements");
    prof_begin(ctx);
    int rc = clear_flags(ctx);
    if (rc) return rc;
    rc = ntt_dev(ctx, ctx->slot[slot].as<uint32_t>(), n, inverse);
    if (rc) return rc;
    return finish(ctx);
}

int kzg_staging_buffer(kzg_ctx* ctx, uint64_t bytes, void** out_ptr) {
    if (!ctx || !out_ptr) return KZG_E_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (bytes > ctx->stage_cap) {
        if (ctx->stage_host) (void)hipHostFree(ctx->stage_host);
        ctx->stage_host = nullptr;
        ctx->stage_cap = 0;
        const size_t want = (size_t)bytes + ((size_t)bytes >> 3) + 4096;
        hipError_t e = hipHostMalloc(&ctx->stage_host, want, hipHostMallocDefault);
        if (e != hipSuccess) {
            ctx->stage_host = nullptr;
            return fail(ctx, KZG_E_NOMEM, std::string("hipHostMalloc(staging): ") + hipGetErrorString(e));
        }
        ctx->stage_cap = want;
    }
    *out_ptr = ctx->stage_host;
    return KZG_OK;
}

int kzg_set_profiling(kzg_ctx* ctx, int enable) {
    if (!ctx) return KZG_E_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    ctx->profiling = enable != 0;
    return KZG_OK;
}
int kzg_get_timings(kzg_ctx* ctx, float* out_ms, int count) {
    if (!ctx || !out_ms) return KZG_E_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    for (int i = 0; i < count && i < KZG_T_COUNT; i++) out_ms[i] = ctx->tms[i];
    return KZG_OK;
}
int kzg_msm_plan(kzg_ctx* ctx, uint64_t n, int32_t out[4]) {
    if (!ctx || !out || !ctx->c) return KZG_E_ARG;
    const uint64_t entries = n * (uint64_t)ctx->nwin;
    const int chunk = pick_chunk(entries);
    out[0] = chunk;
    out[1] = (int32_t)((entries + chunk - 1) / chunk);
    out[2] = (int32_t)ctx->nbuckets;
    out[3] = ctx->nwin;
    return KZG_OK;
}

int kzg_b64_decode_fr(const char* packed43, uint64_t n, uint8_t* out_be32) {
    if ((n && !packed43) || (n && !out_be32)) return KZG_E_ARG;
    b64_init();
    for (uint64_t k = 0; k < n; k++) {
        const uint8_t* s = reinterpret_cast<const uint8_t*>(packed43) + 43 * k;
        uint8_t* o = out_be32 + 32 * k;
        int bad = 0;
        for (int g = 0; g < 10; g++) {
            int a = b64_rev[s[4 * g]], b = b64_rev[s[4 * g + 1]], c = b64_rev[s[4 * g + 2]], d = b64_rev[s[4 * g + 3]];
            bad |= (a | b | c | d) < 0;
            uint32_t v = ((uint32_t)a << 18) | ((uint32_t)b << 12) | ((uint32_t)c << 6) | (uint32_t)d;
            o[3 * g] = (uint8_t)(v >> 16); o[3 * g + 1] = (uint8_t)(v >> 8); o[3 * g + 2] = (uint8_t)v;
        }
        int a = b64_rev[s[40]], b = b64_rev[s[41]], c = b64_rev[s[42]];
        bad |= (a | b | c) < 0;
        uint32_t v = ((uint32_t)a << 12) | ((uint32_t)b << 6) | (uint32_t)c;  // 18 bits, low 2 must be zero
        bad |= (v & 3u) != 0;
        o[30] = (uint8_t)(v >> 10); o[31] = (uint8_t)(v >> 2);
        if (bad) return KZG_E_SCALAR;
    }
    return KZG_OK;
}
int kzg_b64_encode_fr(const uint8_t* be32, uint64_t n, char* out_packed43) {
    if ((n && !be32) || (n && !out_packed43)) return KZG_E_ARG;
    for (uint64_t k = 0; k < n; k++) {
        const uint8_t* i = be32 + 32 * k;
        char* o = out_packed43 + 43 * k;
        for (int g = 0; g < 10; g++) {
            uint32_t v = ((uint32_t)i[3 * g] << 16) | ((uint32_t)i[3 * g + 1] << 8) | i[3 * g + 2];
            o[4 * g] = B64[v >> 18]; o[4 * g + 1] = B64[(v >> 12) & 63]; o[4 * g + 2] = B64[(v >> 6) & 63];
            o[4 * g + 3] = B64[v & 63];
        }
        uint32_t v = (((uint32_t)i[30] << 8) | i[31]) << 2;
        o[40] = B64[v >> 12]; o[41] = B64[(v >> 6) & 63]; o[42] = B64[v & 63];
    }
    return KZG_OK;
}

int kzg_test_field(kzg_ctx* ctx, int field, int op, const uint8_t* a_be, const uint8_t* b_be, uint8_t* out_be,
                   uint64_t n) {
    if (!ctx || !a_be || !b_be || !out_be || !n) return KZG_E_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (int busy = need_idle(ctx)) return busy;
    const size_t w = field == 0 ? 48 : 32;
    HIPCHK(ctx, ctx->in_be.ensure(2 * n * w));
    HIPCHK(ctx, ctx->out_be.ensure(n * w));
    uint8_t* da = ctx->in_be.as<uint8_t>();
    uint8_t* db = da + n * w;
    HIPCHK(ctx, hipMemcpyAsync(da, a_be, n * w, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(db, b_be, n * w, hipMemcpyHostToDevice, ctx->stream));
    uint32_t blocks = (uint32_t)((n + 255) / 256);
    if (field == 0) k_test_fp<<<blocks, 256, 0, ctx->stream>>>(op, da, db, ctx->out_be.as<uint8_t>(), n);
    else k_test_fr<<<blocks, 256, 0, ctx->stream>>>(op, da, db, ctx->out_be.as<uint8_t>(), n);
    HIPCHK(ctx, hipMemcpyAsync(out_be, ctx->out_be.p, n * w, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    HIPCHK(ctx, hipGetLastError());
    return KZG_OK;
}
int kzg_test_g1(kzg_ctx* ctx, int op, const uint8_t* a_be96, const uint8_t* b_be96, uint8_t* out_be96, uint64_t n) {
    if (!ctx || !a_be96 || !b_be96 || !out_be96 || !n) return KZG_E_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (int busy = need_idle(ctx)) return busy;
    HIPCHK(ctx, ctx->in_be.ensure(2 * n * 96));
    HIPCHK(ctx, ctx->out_be.ensure(n * 96));
    uint8_t* da = ctx->in_be.as<uint8_t>();
    uint8_t* db = da + n * 96;
    HIPCHK(ctx, hipMemcpyAsync(da, a_be96, n * 96, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(db, b_be96, n * 96, hipMemcpyHostToDevice, ctx->stream));
    k_test_g1<<<(uint32_t)((n + 255) / 256), 256, 0, ctx->stream>>>(op, da, db, ctx->out_be.as<uint8_t>(), n);
    HIPCHK(ctx, hipMemcpyAsync(out_be96, ctx->out_be.p, n * 96, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    HIPCHK(ctx, hipGetLastError());
    return KZG_OK;
}

}  // extern "C"
