// ctx.hip.h -- INTERNAL to libkzg_mi355x.so (never installed, never included by a caller): the context, its lanes and
// the helpers the translation units of the C-ABI share.  The boundary itself is include/kzg_mi355x.h; the seam it fills
// is the prover client of the reference miner (reference base/miner.py:73-84 lifecycle; neurons/miner.py:38-61 commit / open).
//
//   lanes.hip     context lifecycle, lane / ticket bookkeeping glue, record publish + host wait, staging, profiling
//   srs.hip       setup loaders (memory, file, per-device slices), synthetic SRS, window tables, read-back
//   pipeline.hip  the host-side sequencing of the kernels: one MSM (msm_core), a row's commit / open (commit_open_dev)
//   serve.hip     the serving entry points: commit / open / msm / ntt / eval, resident slots, tickets, sums
//   comm.hip      the library's own RCCL collective (kzg_comm_*, kzg_msm_sharded, the stream-chained pair)
//   abi_test.hip  unit-op test hooks (include/kzg_mi355x_test.h)
// No CPU arithmetic fallback exists anywhere here: if HIP fails, the call fails.
#pragma once
#include <hip/hip_runtime.h>

#include <errno.h>
#include <stdlib.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <system_error>
#include <thread>
#include <utility>
#include <vector>

#include "../../include/kzg_mi355x.h"
#include "../../include/kzg_mi355x_test.h"
#include "fr_kernels.hip.h"
#include "msm.hip.h"
#include "fp_lp.hip.h"
#include "rccl_dl.h"
#include "lanebook.h"

#define KZG_VERSION "kzg_mi355x 0.6 (gfx950)"
#define N_SLOTS 4
#ifndef N_LANES
#define N_LANES 4
#endif
#define N_STAGE 4
#define KZG_MAX_GATHER 4096   // partials one kzg_msm_sharded_finish can sum (ranks of a job)

void launch_calibrate_mad(hipStream_t s, uint64_t* out, uint32_t blocks, uint32_t iters);   // csrc/calibrate.hip
int calibrate_unroll();
// two spinning single-wave kernels on two streams: do they overlap?  (csrc/calibrate.hip; kzg_runtime_info)
void launch_spin_probe(hipStream_t s, uint64_t* out2, uint64_t ticks);
namespace kzg_host {  // finish_host.cpp
void xyzz_to_c48(const uint32_t* xyzz, uint8_t out48[48]);
void xyzz_pair_to_c48(const uint32_t* xyzz0, const uint32_t* xyzz1, uint8_t out0[48], uint8_t out1[48]);
void xyzz_to_partial192(const uint32_t* xyzz, uint8_t out192[192]);
}  // namespace kzg_host

namespace kzg_impl {

// device allocation that frees itself: an early error return can no longer leak it (move-only)
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    DevBuf(DevBuf&& o) noexcept : p(o.p), cap(o.cap) { o.p = nullptr; o.cap = 0; }
    DevBuf& operator=(DevBuf&& o) noexcept {
        if (this != &o) {
            release();
            p = o.p; cap = o.cap;
            o.p = nullptr; o.cap = 0;
        }
        return *this;
    }
    ~DevBuf() { release(); }
    hipError_t ensure(size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        release();
        size_t want = bytes + (bytes >> 3) + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            e = hipMalloc(&p, bytes);
            want = bytes;
        }
        if (e == hipSuccess) cap = want;
        else p = nullptr;
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

// A lane = everything ONE request needs: its own HIP stream, MSM workspace (sort / bucket / carry buffers), request
// buffers (uploaded row, coefficients, quotient, scan scratch) and a 1-KB "tail" record whose first half comes back to
// the host in a single copy.  A call owns its lane from acquire to release, so host threads calling into one ctx (the
// reference's axon runs Miner.forward on worker threads, neurons/miner.py:106-135) run concurrently on different lanes:
// one request's sort and latency-bound tail hide under another's accumulate.  The SRS tables are shared, read-only.
using kzg_book::LANE_CALL;      // lane / ticket / staging / row-cache bookkeeping: csrc/lanebook.h (HIP-free, TSan-driven)
using kzg_book::LANE_TICKET;
// tail record (device, 1024 B).  [0, TB_COPY) is copied to the lane's pinned buffer when a request finishes.
enum {
    TB_RES0 = 0, TB_RES1 = 224,   // result points, XYZZ working form (2 x 224 B)
    TB_EVAL = 448,                // y = f(alpha), 32 B big-endian
    TB_FLAGS = 480,               // u32 x 4: [0] bad scalar, [1] bad point, [2] longest carry run, [3] sort overflow
    TB_VERIFY = 496,              // u32: a row-cache hit whose uploaded bytes differ from the cached row's (kzg_*_cached)
    TB_C48 = 512, TB_P48 = 576,   // GPU-side encodings (host_finish off)
    TB_PART = 640,                // 192-byte partial (GPU-side packing)
    TB_COPY = 832,
    TB_ALPHA_M = 832, TB_Y_M = 864, TB_ALPHA_BE = 896, TB_SIZE = 1024
};
#define PIN_MAXLEN 1024           // offset of the fold-depth read-back inside the lane's pinned page
#define PIN_SEQ 2048              // sequence word of the last published record (polled by finish())
#define PIN_SEQ_SORT 2052         // sequence word of the last published fold-depth / overflow pair (polled by msm_core)
struct Lane {
    int index = 0;
    hipStream_t stream = nullptr;
    DevBuf rank, sorted, hist, offsets, bufA, bufB, bufC, bufD, carries, carry_key;      // MSM workspace
    DevBuf ntt_mid;               // the vector between the passes of an NTT (9 words per element)
    DevBuf gather;                // kzg_msm_sharded_finish: the gathered partials, unpacked (own buffer: the MSM may still run)
    DevBuf comm_send, comm_recv;  // kzg_msm_sharded: this rank's packed 192-byte partial / the `world` gathered ones (sized by kzg_comm_init)
    DevBuf in_be, scal, coeffA, coeffB, qbuf, hbuf, hnext, out_be;                 // request buffers
    uint8_t* tail = nullptr;      // device, TB_SIZE
    uint8_t* pin = nullptr;       // host pinned, 4096
    uint8_t* pin_dev = nullptr;   // the same page as the GPU addresses it
    uint32_t pub_seq = 0;         // sequence number of the last record publish on this lane
    uint32_t sort_seq = 0;        // ... and of the last fold-depth publish
    bool expect_short = false;    // the request in flight is a short one (set by msm_core): finish() may poll for its record
    uint32_t expect_us = 0;       // ... and roughly how long its GPU work takes (bounds the polling)
    bool flags_clean = false;     // the tail record's flag words are zero (left so by the last request's publish)
    bool sort_ws_clean = false;   // the sort's partition counts are zero (left so by every completed sort)
    int skew_hint = 0;            // > 0: the last fast sort overflowed (skewed scalars): go straight to the exact sort
    hipEvent_t ev_sorted = nullptr, ev_done = nullptr, ev_coeffs = nullptr, ev_ext = nullptr;
    const uint8_t* in_be_src = nullptr;   // where upload_fr found the request's big-endian row on the device (in_be or a staging twin)
    hipStream_t vstream = nullptr;   // row-cache hits: upload of the caller's row + its comparison with the cached one,
    hipEvent_t ev_verify = nullptr;  // beside the request's own kernels (the lane's publish waits for this event)
    DevBuf vbuf;
    bool partial = false;         // outstanding ticket wants the 192-byte partial
    // profiling spans of the call running on this lane
    std::vector<hipEvent_t> ev_pool;
    size_t ev_used = 0;
    std::vector<StageSpan> spans;
    g1_xyzz_t* res() const { return reinterpret_cast<g1_xyzz_t*>(tail + TB_RES0); }
    uint32_t* flags() const { return reinterpret_cast<uint32_t*>(tail + TB_FLAGS); }
};
struct Stage {
    void* p = nullptr;
    size_t cap = 0;               // (who holds the buffer is the book's business: ctx->book.stage_*)
    // kzg_staging_flush: a device twin that receives the buffer's prefix WHILE the host is still decoding the rest
    DevBuf twin;
    uint64_t flushed = 0;         // bytes [0, flushed) of p are in (or on their way to) the twin
    uint64_t consumed_by = 0;     // id of the API call that was served from the twin (0: none yet).  The flushes are ONE-SHOT:
                                  // a later call handed the same pointer (the holder may have rewritten the buffer) uploads
                                  // the ordinary way, and the next flush starts again from offset 0
    hipEvent_t ev = nullptr;      // recorded on ctx->h2d behind the last flush
    // the pointer as OTHER threads may read it (flushed_twin scans every record; only the holder touches the rest)
    std::atomic<void*> p_pub{nullptr};
};

}  // namespace kzg_impl

struct kzg_ctx {
    int device = 0;
    kzg_book::LaneBook<N_LANES, N_STAGE> book;   // lanes, tickets, staging pool, row-cache slots: csrc/lanebook.h
    std::mutex mu;                 // guards the twiddle caches, config and timings
    int c_user = 0, c = 0, nwin = 0;
    int poll_timeout_ms = 200;   // finish(): how long the pinned page is polled before falling back to the stream
    WinLayout lay;
    uint32_t nbuckets = 0;
    // resident SRS + window tables: table[w*stride + j] = 2^off[w] P_j   (read-only while any lane is busy)
    kzg_impl::DevBuf table;
    uint64_t stride = 0, T = 0;
    int scale = 0, mscale = 0;
    kzg_impl::Lane lane[N_LANES];
    kzg_impl::DevBuf slot[N_SLOTS];
    uint64_t slot_n[N_SLOTS] = {0, 0, 0, 0};
    int slot_mont[N_SLOTS] = {0, 0, 0, 0};
    std::map<int, kzg_impl::DevBuf> tw_fwd, tw_inv, inv_n;
    kzg_impl::Stage stage[N_STAGE];
    hipStream_t h2d = nullptr;     // the copy stream of kzg_staging_flush (one for all staging buffers: they share the link)
    // kzg_g1_sum*: own stream and buffers, independent of the lanes
    std::mutex aux_mu;
    hipStream_t aux = nullptr;
    kzg_impl::DevBuf aux_in, aux_pts, aux_out;
    uint8_t* aux_pin = nullptr;
    // the library's own communicator (kzg_comm_*): one ncclAllGather of 192 B per rank per sharded MSM, on the lane's stream
    struct Comm {
        std::mutex mu;            // RCCL allows one thread at a time per communicator: guards every call that names `comm`
        ncclComm_t comm = nullptr;
        hipStream_t first_stream = nullptr;   // the stream kzg_comm_init's first (connecting) all_gather ran on; lives with `comm`
        int rank = 0, world = 0;
        int timeout_ms = 0;       // kzg_comm_set_timeout (0: wait for ever)
        bool broken = false;      // a collective failed or timed out and the communicator was aborted
        uint64_t gen = 0;         // bumped whenever `comm` is installed, aborted or dropped: a sharded MSM compares the value it
                                  // started under with the one it finds after its wait (another lane's timeout may have aborted
                                  // the communicator UNDER its collective: the stream then ran on over stale bytes)
        std::string why;
        std::atomic<int> stall_ms{0};   // kzg_test_comm_stall: the next `stall_left` sharded MSMs spin this long first
        std::atomic<int> stall_left{0};
    } comm;
    int profiling = 0;   // 0 off, 1 every stage (calls serialise on lane 0), 2 the accumulate kernel only (no serialisation)
    bool host_finish = true;
    bool srs_subgroup_check = true;  // kzg_load_srs*: G1 membership of every point (kzg_set_srs_subgroup_check)
    float tms[KZG_T_COUNT] = {0};  // stage times of the last completed hot-path call
    double load_stats[4] = {0, 0, 0, 0};   // kzg_get_load_stats
    int32_t rt_info[4] = {0, 0, 0, 0};     // kzg_runtime_info: measured once by kzg_create
    // coefficient vectors of the last few rows, keyed by the caller's 128-bit content tag (kzg_commit_cached /
    // kzg_open_cached): the reference miner sends the SAME row twice per request (neurons/miner.py:56-61)
    // (which slot holds which row, and who is using it: ctx->book.rcache_*)
    struct RowCache {
        kzg_impl::DevBuf coef;
        kzg_impl::DevBuf raw;        // the row's 32-byte big-endian elements as they were uploaded: what a hit is verified against
    } rcache[N_LANES];
};

namespace kzg_impl {

// ---- errors: the message of the last failing call ON THIS THREAD (calls run concurrently: a per-ctx string would be torn)
int fail(kzg_ctx* ctx, int code, const std::string& msg);
const char* last_error_cstr();
#define HIPCHK(ctx, expr)                                                                                   \
    do {                                                                                                    \
        hipError_t _e = (expr);                                                                             \
        if (_e != hipSuccess)                                                                               \
            return fail(ctx, _e == hipErrorOutOfMemory ? KZG_E_NOMEM : KZG_E_HIP,                           \
                        std::string(#expr) + ": " + hipGetErrorString(_e));                                 \
    } while (0)
// the API call running on this thread: a fresh id whenever a call takes its lane(s) (LaneHold::take*).  What a one-shot
// resource (the flushed twin of a staging buffer) remembers of the call it served.
uint64_t call_id_new();
uint64_t call_id_current();

// ---- lane ownership (lanes.hip; the state machine itself is csrc/lanebook.h)
int lane_acquire(kzg_ctx* ctx, int state, int* out_li);
int lane_try_second(kzg_ctx* ctx, int first);
void lane_release(kzg_ctx* ctx, int li);
int lanes_acquire_all(kzg_ctx* ctx);
void lanes_release_all(kzg_ctx* ctx);
int ticket_claim(kzg_ctx* ctx, int ticket);
// Owns one lane (optionally a second) for the duration of a call.  Unless the call reached its normal end (`clean`),
// the streams are drained before the lanes become reusable: a HIP failure midway leaves kernels queued that still read
// and write the lane's buffers.
struct LaneHold {
    kzg_ctx* ctx;
    int li = -1, li2 = -1;
    bool all = false, clean = false;
    explicit LaneHold(kzg_ctx* c) : ctx(c) {}
    LaneHold(const LaneHold&) = delete;
    LaneHold& operator=(const LaneHold&) = delete;
    int take() {
        (void)call_id_new();
        return lane_acquire(ctx, LANE_CALL, &li);
    }
    int take_all() {
        (void)call_id_new();
        int rc = lanes_acquire_all(ctx);
        if (rc == KZG_OK) { all = true; li = 0; }
        return rc;
    }
    Lane& L() { return ctx->lane[li]; }
    Lane* second() {
        if (li2 < 0) li2 = lane_try_second(ctx, li);
        return li2 >= 0 ? &ctx->lane[li2] : nullptr;
    }
    void drain();     // wait for everything this call queued (what the destructor does for a call that did not end cleanly)
    ~LaneHold();
};

// ---- per-stage HIP-event spans (kzg_set_profiling)
hipEvent_t prof_event(Lane& L);
struct Span {
    Lane* lane = nullptr;
    int idx = -1;
    hipStream_t stream;
    Span(kzg_ctx* c, Lane& L, int stage, hipStream_t st = nullptr) : stream(st ? st : L.stream) {
        if (!c->profiling || (c->profiling == 2 && stage != KZG_T_ACCUMULATE)) return;
        lane = &L;
        StageSpan s{stage, prof_event(L), prof_event(L)};
        (void)hipEventRecord(s.a, stream);
        L.spans.push_back(s);
        idx = (int)L.spans.size() - 1;
    }
    ~Span() {
        if (idx >= 0) (void)hipEventRecord(lane->spans[idx].b, stream);
    }
};
void prof_begin(kzg_ctx* ctx, Lane& L);   // opens the KZG_T_TOTAL span of the call on lane L
void prof_close(kzg_ctx* ctx, Lane& L);   // ... ends it just before the last copy-back
void prof_end(kzg_ctx* ctx, Lane& L);     // lane stream already synchronised: stage times -> ctx->tms

// ---- ending a request (lanes.hip)
// waits until the pinned word at `off` shows `seq` (a k_publish has landed); false if it does not within the budget
bool poll_pinned(const kzg_ctx* ctx, const Lane& L, uint32_t off, uint32_t seq);
int need_srs(kzg_ctx* ctx);
int clear_flags(kzg_ctx* ctx, Lane& L);
int finish(kzg_ctx* ctx, Lane& L, bool allow_poll = true);
void result_c48(kzg_ctx* ctx, Lane& L, int which, uint8_t out48[48]);
void result_partial(kzg_ctx* ctx, Lane& L, uint8_t out192[192]);
void queue_encode(kzg_ctx* ctx, Lane& L, bool first, bool second);
void queue_pack(kzg_ctx* ctx, Lane& L);

// ---- the kernel sequences (pipeline.hip)
inline int ilog2_exact(uint64_t n) {
    if (!n || (n & (n - 1))) return -1;
    int l = 0;
    while (((uint64_t)1 << l) < n) l++;
    return l;
}
int pick_chunk(uint64_t entries);
int msm_core(kzg_ctx* ctx, Lane& L, const uint32_t* scalars, int mont, uint64_t n, uint64_t srs_offset, g1_xyzz_t* out_xyzz,
             const uint32_t* scalars2 = nullptr, int mont2 = 0);
int ensure_twiddles(kzg_ctx* ctx, Lane& L, int log_n, int inverse, uint32_t** tw, uint32_t** invn);
int row_to_coeffs(kzg_ctx* ctx, Lane& L, const uint32_t* row_dev, uint64_t T, int evaluation_form, const uint32_t** coeffs,
                  uint32_t* dst = nullptr);
int check_worker(kzg_ctx* ctx, uint32_t i, uint64_t T);
const uint8_t* flushed_twin(kzg_ctx* ctx, const uint8_t* host_ptr, uint64_t bytes, hipEvent_t* ev);
int upload_fr(kzg_ctx* ctx, Lane& L, const uint8_t* be32, uint64_t n, uint32_t* dst, int to_mont);
struct VerifyJob {   // a row-cache hit's evidence: the caller's row (host) against the bytes the slot was filled from (device)
    const uint8_t* row_be32;
    uint64_t T;
    const uint32_t* cached_raw;
};
int commit_open_dev(kzg_ctx* ctx, LaneHold& H, uint32_t i, const uint32_t* row_dev, uint64_t T, int evaluation_form,
                    const uint8_t* alpha_be32, uint8_t* out_c48, uint8_t* out_eval32, uint8_t* out_p48,
                    const uint32_t* coeffs_ready = nullptr, uint32_t* coeffs_dst = nullptr, const VerifyJob* verify = nullptr);

// ---- the collective (comm.hip)
void comm_teardown(kzg_ctx* ctx);

}  // namespace kzg_impl
