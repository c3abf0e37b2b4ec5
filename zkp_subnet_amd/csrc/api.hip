// C-ABI of libkzg_mi355x.so (declared in include/kzg_mi355x.h): context, resident SRS + window tables,
// workspace, and the host-side sequencing of the HIP kernels.  The seam it fills is the prover client of the
// reference miner (reference base/miner.py:73-84 lifecycle; neurons/miner.py:38-61 commit / open).
// No CPU arithmetic fallback exists here: if HIP fails, the call fails.
#include <hip/hip_runtime.h>

#include <errno.h>
#include <stdlib.h>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <new>
#include <atomic>
#include <chrono>
#include <string>
#include <system_error>
#include <thread>
#include <algorithm>
#include <utility>
#include <vector>

#include "../../include/kzg_mi355x.h"
#include "../../include/kzg_mi355x_test.h"
#include "fr_kernels.hip.h"
#include "msm.hip.h"
#include "fp_lp.hip.h"
#include "rccl_dl.h"
#include "lanebook.h"

#define KZG_VERSION "kzg_mi355x 0.5 (gfx950)"
#define N_SLOTS 4
#ifndef N_LANES
#define N_LANES 4
#endif
#define N_STAGE 4
#define KZG_MAX_GATHER 4096   // partials one kzg_msm_sharded_finish can sum (ranks of a job)

void launch_calibrate_mad(hipStream_t s, uint64_t* out, uint32_t blocks, uint32_t iters);   // csrc/calibrate.hip
int calibrate_unroll();
namespace kzg_host {  // finish_host.cpp
void xyzz_to_c48(const uint32_t* xyzz, uint8_t out48[48]);
void xyzz_pair_to_c48(const uint32_t* xyzz0, const uint32_t* xyzz1, uint8_t out0[48], uint8_t out1[48]);
void xyzz_to_partial192(const uint32_t* xyzz, uint8_t out192[192]);
}  // namespace kzg_host

namespace {

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
    hipEvent_t ev_sorted = nullptr, ev_done = nullptr, ev_coeffs = nullptr, ev_ext = nullptr, ev_acc = nullptr;
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

}  // namespace

struct kzg_ctx {
    int device = 0;
    kzg_book::LaneBook<N_LANES, N_STAGE> book;   // lanes, tickets, staging pool, row-cache slots: csrc/lanebook.h
    std::mutex mu;                 // guards the twiddle caches, config and timings
    int c_user = 0, c = 0, nwin = 0;
    int poll_timeout_ms = 200;   // finish(): how long the pinned page is polled before falling back to the stream
    WinLayout lay;
    uint32_t nbuckets = 0;
    // resident SRS + window tables: table[w*stride + j] = 2^off[w] P_j   (read-only while any lane is busy)
    DevBuf table;
    uint64_t stride = 0, T = 0;
    int scale = 0, mscale = 0;
    Lane lane[N_LANES];
    DevBuf slot[N_SLOTS];
    uint64_t slot_n[N_SLOTS] = {0, 0, 0, 0};
    int slot_mont[N_SLOTS] = {0, 0, 0, 0};
    std::map<int, DevBuf> tw_fwd, tw_inv, inv_n;
    Stage stage[N_STAGE];
    hipStream_t h2d = nullptr;     // the copy stream of kzg_staging_flush (one for all staging buffers: they share the link)
    // kzg_g1_sum*: own stream and buffers, independent of the lanes
    std::mutex aux_mu;
    hipStream_t aux = nullptr;
    DevBuf aux_in, aux_pts, aux_out;
    uint8_t* aux_pin = nullptr;
    // the library's own communicator (kzg_comm_*): one ncclAllGather of 192 B per rank per sharded MSM, on the lane's stream
    struct Comm {
        std::mutex mu;            // RCCL allows one thread at a time per communicator: guards every call that names `comm`
        ncclComm_t comm = nullptr;
        int rank = 0, world = 0;
        int timeout_ms = 0;       // kzg_comm_set_timeout (0: wait for ever)
        bool broken = false;      // a collective failed or timed out and the communicator was aborted
        std::string why;
        std::atomic<int> stall_ms{0};   // kzg_test_comm_stall
    } comm;
    int profiling = 0;   // 0 off, 1 every stage (calls serialise on lane 0), 2 the accumulate kernel only (no serialisation)
    bool host_finish = true;
    bool srs_subgroup_check = true;  // kzg_load_srs*: G1 membership of every point (kzg_set_srs_subgroup_check)
    // two-lane commit+open: one accumulate at a time instead of two sharing the SIMDs.  Same-box A/B
    // (profiles/r03_ab_serial_accumulate.log): 19.22 / 19.22 / 19.09 ms against 19.34 / 19.20 / 19.21 at 2^22, 5.37-5.40 either
    // way at 2^20 -- no difference: what overlaps the accumulates (sort, opening, trees) is throughput-bound work on the
    // same SIMDs, not idle latency.  Off by default; KZG_SERIAL_ACC=1 turns it on.
    bool serial_accumulate = false;
    float tms[KZG_T_COUNT] = {0};  // stage times of the last completed hot-path call
    double load_stats[4] = {0, 0, 0, 0};   // kzg_get_load_stats
    // coefficient vectors of the last few rows, keyed by the caller's 128-bit content tag (kzg_commit_cached /
    // kzg_open_cached): the reference miner sends the SAME row twice per request (neurons/miner.py:56-61)
    // (which slot holds which row, and who is using it: ctx->book.rcache_*)
    struct RowCache {
        DevBuf coef;
        DevBuf raw;        // the row's 32-byte big-endian elements as they were uploaded: what a hit is verified against
    } rcache[N_LANES];
};

namespace {

// the message of the last failing call ON THIS THREAD (calls run concurrently: a per-ctx string would be torn)
thread_local std::string tl_err;
// the API call running on this thread: a fresh id whenever a call takes its lane(s) (LaneHold::take*).  What a one-shot
// resource (the flushed twin of a staging buffer) remembers of the call it served.
thread_local uint64_t tl_call_id = 0;
std::atomic<uint64_t> g_call_ids{0};
int fail(kzg_ctx*, int code, const std::string& msg) {
    tl_err = msg;
    return code;
}
#define HIPCHK(ctx, expr)                                                                                   \
    do {                                                                                                    \
        hipError_t _e = (expr);                                                                             \
        if (_e != hipSuccess)                                                                               \
            return fail(ctx, _e == hipErrorOutOfMemory ? KZG_E_NOMEM : KZG_E_HIP,                           \
                        std::string(#expr) + ": " + hipGetErrorString(_e));                                 \
    } while (0)

// ---- lane ownership
// blocking call: take the lowest free lane (keeps a single-threaded caller on lane 0 and its warm workspace); wait while
// lanes are merely busy with other calls; fail with KZG_E_BUSY when every lane is parked under an MSM ticket (only
// kzg_msm_wait can free those: waiting here could deadlock a single-threaded caller).  Profiling pins everything to
// lane 0 so that stage times stay attributable.
int lane_acquire(kzg_ctx* ctx, int state, int* out_li) {
    switch (ctx->book.acquire(state, out_li)) {
        case kzg_book::BOOK_OK: return KZG_OK;
        case kzg_book::BOOK_BUSY_NO_TICKET_LANE: return fail(ctx, KZG_E_BUSY, "every MSM lane is taken: call kzg_msm_wait first");
        default: return fail(ctx, KZG_E_BUSY, "every lane holds an outstanding MSM ticket: call kzg_msm_wait first");
    }
}
int lane_try_second(kzg_ctx* ctx, int first) { return ctx->book.try_second(first); }
void lane_release(kzg_ctx* ctx, int li) { ctx->book.release(li); }
// exclusive operations ((re)loading the SRS, uploading a resident slot, reading the SRS back): all lanes, on lane 0
int lanes_acquire_all(kzg_ctx* ctx) {
    if (ctx->book.acquire_all() != kzg_book::BOOK_OK) return fail(ctx, KZG_E_BUSY, "an MSM ticket is outstanding: call kzg_msm_wait first");
    return KZG_OK;
}
void lanes_release_all(kzg_ctx* ctx) { ctx->book.release_all(); }
// the one waiter (or canceller) of a ticket: TICKET -> WAITING under the book's lock
int ticket_claim(kzg_ctx* ctx, int ticket) {
    if (ctx->book.ticket_claim(ticket) != kzg_book::BOOK_OK)
        return fail(ctx, KZG_E_ARG, "no outstanding MSM on this ticket (or it is already being waited for)");
    return KZG_OK;
}
// Owns one lane (optionally a second) for the duration of a call.  Unless the call reached its normal end (`clean`),
// the streams are drained before the lanes become reusable: a HIP failure midway leaves kernels queued that still read
// and write the lane's buffers.
struct LaneHold {
    kzg_ctx* ctx;
    int li = -1, li2 = -1;
    bool all = false, clean = false;
    explicit LaneHold(kzg_ctx* c) : ctx(c) {}
    int take() {
        tl_call_id = ++g_call_ids;
        return lane_acquire(ctx, LANE_CALL, &li);
    }
    int take_all() {
        tl_call_id = ++g_call_ids;
        int rc = lanes_acquire_all(ctx);
        if (rc == KZG_OK) { all = true; li = 0; }
        return rc;
    }
    Lane& L() { return ctx->lane[li]; }
    Lane* second() {
        if (li2 < 0) li2 = lane_try_second(ctx, li);
        return li2 >= 0 ? &ctx->lane[li2] : nullptr;
    }
    void drain() {   // wait for everything this call queued (what the destructor does for a call that did not end cleanly)
        if (li >= 0) (void)hipStreamSynchronize(ctx->lane[li].stream);
        if (li >= 0) (void)hipStreamSynchronize(ctx->lane[li].vstream);
        if (li2 >= 0) (void)hipStreamSynchronize(ctx->lane[li2].stream);
        (void)hipGetLastError();
    }
    ~LaneHold() {
        if (li < 0) return;
        if (!clean) {
            (void)hipStreamSynchronize(ctx->lane[li].stream);
            (void)hipStreamSynchronize(ctx->lane[li].vstream);
            ctx->lane[li].sort_ws_clean = false;
            if (li2 >= 0) {
                (void)hipStreamSynchronize(ctx->lane[li2].stream);
                ctx->lane[li2].sort_ws_clean = false;
            }
            (void)hipGetLastError();
        }
        if (all) { lanes_release_all(ctx); return; }
        if (li2 >= 0) lane_release(ctx, li2);
        lane_release(ctx, li);
    }
};

hipEvent_t prof_event(Lane& L) {
    if (L.ev_used == L.ev_pool.size()) {
        hipEvent_t e;
        (void)hipEventCreate(&e);
        L.ev_pool.push_back(e);
    }
    return L.ev_pool[L.ev_used++];
}
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
// opens the KZG_T_TOTAL span of the call on lane L; prof_close() ends it just before the last copy-back
void prof_begin(kzg_ctx* ctx, Lane& L) {
    L.spans.clear();
    L.ev_used = 0;
    if (ctx->profiling != 1) return;
    StageSpan s{KZG_T_TOTAL, prof_event(L), prof_event(L)};
    (void)hipEventRecord(s.a, L.stream);
    L.spans.push_back(s);
}
void prof_close(kzg_ctx* ctx, Lane& L) {
    if (ctx->profiling == 1 && !L.spans.empty() && L.spans[0].stage == KZG_T_TOTAL) (void)hipEventRecord(L.spans[0].b, L.stream);
}
void prof_end(kzg_ctx* ctx, Lane& L) {  // lane stream already synchronised
    if (L.spans.empty()) return;
    float t[KZG_T_COUNT] = {0};
    for (auto& s : L.spans) {
        float ms = 0.f;
        hipError_t e = hipEventElapsedTime(&ms, s.a, s.b);
        if (e == hipErrorNotReady) {   // the host saw the published record before the runtime retired the event
            (void)hipGetLastError();
            (void)hipEventSynchronize(s.b);
            e = hipEventElapsedTime(&ms, s.a, s.b);
        }
        if (e == hipSuccess) t[s.stage] += ms;
    }
    L.spans.clear();
    std::lock_guard<std::mutex> lk(ctx->mu);
    memcpy(ctx->tms, t, sizeof(t));
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
    if (lg <= 22) return 20;
    if (lg <= 25) return 22;   // 2^23: 19.13 -> 18.89 ms against c = 20
    return 24;   // 2^26: 11 windows, 2^23 buckets -- 132.2 -> 128.4 ms (the tree grows 1.2 -> 4.4 ms, the accumulate drops 8 %)
}
// nwin = ceil(256/c) windows of width base or base+1 (256 = nwin*base + extra): the widest is <= c bits
// Everything that describes one resident table.  A (re)load builds table + spec ASIDE and installs both only when the
// whole load has succeeded: a failed reload leaves the previous SRS serving.
struct TableSpec {
    int c = 0, nwin = 0;
    WinLayout lay;
    uint32_t nbuckets = 0;
    uint64_t stride = 0, T = 0;
    int scale = 0, mscale = 0;
};
void spec_window(TableSpec& sp, int c) {
    const int nwin = (256 + c - 1) / c, base = 256 / nwin, extra = 256 % nwin;
    sp.nwin = sp.lay.nwin = nwin;
    int off = 0;
    for (int w = 0; w < nwin; w++) {
        sp.lay.off[w] = (uint16_t)off;
        off += base + (w < extra ? 1 : 0);
    }
    sp.lay.off[nwin] = 256;
    sp.c = base + (extra ? 1 : 0);
    sp.nbuckets = 1u << (sp.c - 1);
}
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
int ilog2_exact(uint64_t n) {
    if (!n || (n & (n - 1))) return -1;
    int l = 0;
    while (((uint64_t)1 << l) < n) l++;
    return l;
}

// ---- the MSM pipeline on device-resident scalars -> one XYZZ point at out_xyzz (device), on lane L.
// With scalars2 != null: TWO MSMs over the same n points in one pass (the commitment and the opening of one row):
// set b is sorted into bucket set b, the sort / accumulate / fold / tree kernels simply see twice the buckets, the
// tree stops at two roots and out_xyzz[0..1] receive the two sums.  One kernel sequence, one latency-bound tail.
// The only host wait inside is on the 4-byte fold-depth read-back; the calling thread holds no lock meanwhile.
// waits until the pinned word at `off` shows `seq` (a k_publish has landed); false if it does not within the budget
static bool poll_pinned(const kzg_ctx* ctx, const Lane& L, uint32_t off, uint32_t seq) {
    // The budget is the request's own expected duration (set by msm_core from its entry count), not a fixed 200 ms: a
    // request that overruns it is waited for in the runtime (which sleeps) instead of burning a host core.  After the
    // first ~50 us the loop yields between probes, so four axon threads waiting at once do not pin four cores.
    const volatile uint32_t* w = reinterpret_cast<const volatile uint32_t*>(L.pin + off);
    const auto t0 = std::chrono::steady_clock::now();
    const auto budget = std::chrono::microseconds(
        std::min<uint64_t>((uint64_t)ctx->poll_timeout_ms * 1000, 2 * (uint64_t)L.expect_us + 500));
    bool yielding = false;
    for (uint32_t spin = 0;; spin++) {
        if (*w == seq) {
            std::atomic_thread_fence(std::memory_order_acquire);
            return true;
        }
        if ((spin & 0xff) == 0xff) {
            const auto dt = std::chrono::steady_clock::now() - t0;
            if (dt > budget) return false;
            yielding = dt > std::chrono::microseconds(50);
        }
        if (yielding) std::this_thread::yield();
        else __builtin_ia32_pause();
    }
}
// acc_wait / acc_record (both optional): the accumulate kernel starts only after the event acc_wait (another lane's
// accumulate) and records acc_record when it ends -- the "one multiplier-bound kernel at a time" form of a two-lane
// commit+open (KZG_SERIAL_ACC=1).  Measured against two concurrent accumulates: no difference (kzg_ctx::serial_accumulate).
int msm_core(kzg_ctx* ctx, Lane& L, const uint32_t* scalars, int mont, uint64_t n, uint64_t srs_offset,
             g1_xyzz_t* out_xyzz, const uint32_t* scalars2 = nullptr, int mont2 = 0, hipEvent_t acc_wait = nullptr,
             hipEvent_t acc_record = nullptr) {
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
        static const bool separate = getenv("KZG_SORT_SEPARATE_TAIL") != nullptr;   // A/B knob: the two extra launches
        const SortTail tail{(uint32_t)sh.chunk, L.bufA.as<g1_xyzz_t>(), reinterpret_cast<uint32_t*>(L.pin_dev + PIN_MAXLEN),
                            reinterpret_cast<uint32_t*>(L.pin_dev + PIN_SEQ_SORT), ++L.sort_seq};
        launch_msm_sort(s, sh, scalars, mont, scalars2, mont2, L.hist.as<uint32_t>(), ws_clean, L.rank.as<uint2>(),
                        L.offsets.as<uint32_t>(), L.sorted.as<uint32_t>(), max_len_d, fast_mode, max_len_d + 1,
                        separate ? nullptr : &tail);
        if (separate) {
            launch_fold_maxlen(s, L.offsets.as<uint32_t>(), sh.nbuckets, (uint32_t)sh.chunk, max_len_d, L.bufA.as<g1_xyzz_t>());
            launch_publish(s, max_len_d, L.pin_dev + PIN_MAXLEN, 8, nullptr,
                           reinterpret_cast<uint32_t*>(L.pin_dev + PIN_SEQ_SORT), tail.seq);
        }
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
    if (acc_wait) HIPCHK(ctx, hipStreamWaitEvent(s, acc_wait, 0));
    {
        Span sp(ctx, L, KZG_T_ACCUMULATE);
        launch_msm_accumulate(s, sh, ctx->table.as<g1_affine_t>(), L.offsets.as<uint32_t>(), L.sorted.as<uint32_t>(),
                              L.bufA.as<g1_xyzz_t>(), L.carries.as<g1_xyzz_t>(), L.carry_key.as<uint32_t>(), nchunks);
    }
    if (acc_record) HIPCHK(ctx, hipEventRecord(acc_record, s));
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
        if (acc_record) HIPCHK(ctx, hipEventRecord(acc_record, s));   // re-recorded: a later waiter sees this launch
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
    // three buffers in rotation: a level reads its own array and the P array of the level below, writes the next
    g1_xyzz_t* in = L.bufA.as<g1_xyzz_t>();
    g1_xyzz_t* prev = L.bufC.as<g1_xyzz_t>();
    g1_xyzz_t* out = L.bufB.as<g1_xyzz_t>();
    {
        Span sp(ctx, L, KZG_T_TREE);
        uint32_t n_in = sh.nbuckets;
        g1_xyzz_t* extra = L.bufD.as<g1_xyzz_t>();
        for (int level = 0; n_in > (uint32_t)nbatch;) {
            if ((n_in >> 2) >= (uint32_t)nbatch && msm_tree_level2_ok(n_in, level)) {
                // two narrow levels per launch: the level + 1 P array goes to `out`, the level + 2 array to `extra`
                launch_msm_tree_level2(s, in, prev, out, extra, n_in, level);
                g1_xyzz_t *old_in = in, *old_prev = prev;
                in = extra;
                prev = out;
                out = old_prev;
                extra = old_in;
                level += 2;
                n_in >>= 2;
                continue;
            }
            launch_msm_tree_level(s, in, prev, out, n_in, level);
            g1_xyzz_t* recycled = prev;
            prev = in;
            in = out;
            out = recycled;
            level++;
            n_in >>= 1;
        }
    }
    {
        Span sp(ctx, L, KZG_T_FINAL);
        // `out` (the buffer the last level did not write and no longer reads) holds the doubled components in between
        launch_msm_final(s, in, prev, ctx->c - 1, nbatch, out_xyzz, out);
    }
    HIPCHK(ctx, hipGetLastError());
    return KZG_OK;
}

int need_srs(kzg_ctx* ctx) {
    if (!ctx->table.p || !ctx->stride) return fail(ctx, KZG_E_ARG, "no SRS resident: call kzg_load_srs / kzg_gen_srs");
    return KZG_OK;
}
// the request's flag words start at zero: left so by the publish that ended the lane's previous request (k_publish
// clears the two input-error flags it has copied; the fold-depth and overflow words are reset by the sort), by a memset
// only on a fresh lane or after a request that failed half-way
int clear_flags(kzg_ctx* ctx, Lane& L) {
    const bool was_clean = L.flags_clean;
    L.flags_clean = false;
    L.expect_short = false;
    L.expect_us = 0;
    if (!was_clean) HIPCHK(ctx, hipMemsetAsync(L.flags(), 0, 16, L.stream));
    return KZG_OK;
}
// ends a request: the lane's tail record comes back in ONE copy (result points, eval, flags, GPU-side encodings)
int finish(kzg_ctx* ctx, Lane& L, bool allow_poll = true) {
    prof_close(ctx, L);
    const uint32_t seq = ++L.pub_seq;
    launch_publish(L.stream, L.tail, L.pin_dev, TB_COPY, L.flags(), reinterpret_cast<uint32_t*>(L.pin_dev + PIN_SEQ), seq);
    // The record is complete when its sequence word arrives: poll the pinned page (the kernel's completion signal and the
    // runtime's wake-up come several microseconds later).  Whatever follows on this lane is stream-ordered behind the
    // publish anyway.  With stage events to read (profiling), on a HIP error, or if nothing arrives: the stream.
    bool seen = false;
#ifndef KZG_NO_POLL
    // only for requests expected to take well under the polling budget (short rows: that is where a few us count); a long
    // MSM waits in the runtime, which may sleep
    // (level-2 profiling keeps polling: its two events per accumulate launch precede the publish in stream order)
    if (allow_poll && ctx->profiling != 1 && L.expect_short) seen = poll_pinned(ctx, L, PIN_SEQ, seq);
#endif
    if (!seen) HIPCHK(ctx, hipStreamSynchronize(L.stream));
    L.flags_clean = true;
    prof_end(ctx, L);
    const uint32_t* f = reinterpret_cast<const uint32_t*>(L.pin + TB_FLAGS);
    if (f[0]) return fail(ctx, KZG_E_SCALAR, "non-canonical Fr scalar (>= r)");
    if (f[1]) return fail(ctx, KZG_E_POINT, (f[1] & 3u) ? "G1 input not reduced or not on the curve"
                                                        : "G1 input on the curve but outside the prime-order subgroup");
    return KZG_OK;
}
// encodes result point `which` (0 / 1) of a finished request: on the host from the XYZZ working form (default), or
// the bytes the GPU encoder left in the tail record
void result_c48(kzg_ctx* ctx, Lane& L, int which, uint8_t out48[48]) {
    if (ctx->host_finish) kzg_host::xyzz_to_c48(reinterpret_cast<const uint32_t*>(L.pin + (which ? TB_RES1 : TB_RES0)), out48);
    else memcpy(out48, L.pin + (which ? TB_P48 : TB_C48), 48);
}
void result_partial(kzg_ctx* ctx, Lane& L, uint8_t out192[192]) {
    if (ctx->host_finish) kzg_host::xyzz_to_partial192(reinterpret_cast<const uint32_t*>(L.pin + TB_RES0), out192);
    else memcpy(out192, L.pin + TB_PART, 192);
}
// queue the GPU-side encoders when the host does not finish (no-ops otherwise)
void queue_encode(kzg_ctx* ctx, Lane& L, bool first, bool second) {
    if (ctx->host_finish) return;
    Span sp(ctx, L, KZG_T_FINAL);
    if (first && second) launch_g1_compress_pair(L.stream, L.res(), L.res() + 1, L.tail + TB_C48, L.tail + TB_P48);
    else if (first) launch_g1_compress(L.stream, L.res(), L.tail + TB_C48);
    else if (second) launch_g1_compress(L.stream, L.res() + 1, L.tail + TB_P48);
}
void queue_pack(kzg_ctx* ctx, Lane& L) {
    if (!ctx->host_finish) launch_xyzz_pack(L.stream, L.res(), reinterpret_cast<uint32_t*>(L.tail + TB_PART), 1);
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
                  uint32_t* dst = nullptr) {   // dst: where the coefficients go instead of the lane's own buffer (row cache)
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
    if ((uint64_t)i * ctx->T + ctx->T > ctx->stride) return fail(ctx, KZG_E_ARG, "worker index outside the resident SRS");
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
            if (st.consumed_by && st.consumed_by != tl_call_id) {
                st.flushed = 0;                // a second call on the same held buffer: ordinary upload from the host bytes
                return nullptr;
            }
            if (st.flushed >= bytes && st.flushed && st.twin.p) {
                st.consumed_by = tl_call_id;
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
struct VerifyJob {   // a row-cache hit's evidence: the caller's row (host) against the bytes the slot was filled from (device)
    const uint8_t* row_be32;
    uint64_t T;
    const uint32_t* cached_raw;
};
int commit_open_dev(kzg_ctx* ctx, LaneHold& H, uint32_t i, const uint32_t* row_dev, uint64_t T, int evaluation_form,
                    const uint8_t* alpha_be32, uint8_t* out_c48, uint8_t* out_eval32, uint8_t* out_p48,
                    const uint32_t* coeffs_ready = nullptr, uint32_t* coeffs_dst = nullptr, const VerifyJob* verify = nullptr) {
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
    // two lanes.  Optionally (KZG_SERIAL_ACC=1) the opening's accumulate queues behind the commitment's -- measured: no gain
    const bool serial_acc = B && ctx->serial_accumulate;
    if (out_c48 && !batched) {
        rc = msm_core(ctx, A, coeffs, 1, T, offset, res, nullptr, 0, nullptr, serial_acc ? A.ev_acc : nullptr);
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
            rc = msm_core(ctx, O, O.qbuf.as<uint32_t>(), 0, T - 1, offset, res + 1, nullptr, 0,
                          (serial_acc && out_c48) ? A.ev_acc : nullptr, nullptr);
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

// A (re)load never touches the serving table until it has succeeded: the new table is built in its own allocation and
// swapped in at the end (288 GB of HBM hold both: mainnet's 34 GB twice is nothing).  Only when the second allocation does
// not fit is the old table given up first -- then, and only then, a failure leaves the context without an SRS.
int plan_table(kzg_ctx* ctx, uint64_t n_points, int scale, int mscale, TableSpec& sp) {
    if (mscale < 0 || scale < mscale || scale - mscale > 30) return fail(ctx, KZG_E_ARG, "bad scale / machines_scale");
    const uint64_t T = (uint64_t)1 << (scale - mscale);
    if (n_points == 0 || n_points % T) return fail(ctx, KZG_E_ARG, "SRS length must be a whole number of worker slices");
    spec_window(sp, ctx->c_user ? ctx->c_user : choose_window(T));
    if ((uint64_t)sp.nwin * n_points >= ((uint64_t)1 << 31))
        return fail(ctx, KZG_E_ARG, "SRS x windows exceeds 2^31 table entries");
    sp.stride = n_points; sp.T = T; sp.scale = scale; sp.mscale = mscale;
    return KZG_OK;
}
void drop_table(kzg_ctx* ctx) {
    ctx->table.release();
    ctx->stride = 0;
    ctx->T = 0;
}
int alloc_new_table(kzg_ctx* ctx, const TableSpec& sp, DevBuf& nt) {
    const size_t bytes = (size_t)sp.nwin * sp.stride * sizeof(g1_affine_t);
    hipError_t e = hipMalloc(&nt.p, bytes);
    if (e != hipSuccess && ctx->table.p) {      // both do not fit: give the old one up first
        (void)hipGetLastError();
        drop_table(ctx);
        e = hipMalloc(&nt.p, bytes);
    }
    if (e != hipSuccess) {
        (void)hipGetLastError();
        nt.p = nullptr;
        return fail(ctx, KZG_E_NOMEM, std::string("hipMalloc(window tables): ") + hipGetErrorString(e));
    }
    nt.cap = bytes;
    return KZG_OK;
}
void install_table(kzg_ctx* ctx, const TableSpec& sp, DevBuf& nt) {
    ctx->table = std::move(nt);   // frees the previous table
    ctx->c = sp.c; ctx->nwin = sp.nwin; ctx->lay = sp.lay; ctx->nbuckets = sp.nbuckets;
    ctx->stride = sp.stride; ctx->T = sp.T; ctx->scale = sp.scale; ctx->mscale = sp.mscale;
}
// drains a stream at scope exit unless disarmed: the new table (declared before it) must not be freed under queued kernels
struct DrainGuard {
    hipStream_t s;
    bool armed = true;
    ~DrainGuard() {
        if (armed) {
            (void)hipStreamSynchronize(s);
            (void)hipGetLastError();
        }
    }
};
int precompute_tables(kzg_ctx* ctx, const TableSpec& sp, g1_affine_t* table) {
    hipStream_t s = ctx->lane[0].stream;
    const uint64_t tile = sp.stride < ((uint64_t)1 << 20) ? sp.stride : ((uint64_t)1 << 20);
    DevBuf tmp;
    HIPCHK(ctx, tmp.ensure((size_t)(sp.nwin - 1) * tile * sizeof(g1_xyzz_t) + 256));
    for (uint64_t first = 0; first < sp.stride; first += tile) {
        uint64_t cnt = sp.stride - first < tile ? sp.stride - first : tile;
        launch_srs_precompute(s, table, sp.stride, first, cnt, sp.lay, tmp.as<g1_xyzz_t>());
    }
    HIPCHK(ctx, hipStreamSynchronize(s));
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
    if (op == 3) {  // the saturated 8 x 32-bit CIOS reference (field.hip.h), self-contained
        fr_t a, b, r;
        limbs_from_be<8>(a.l, a_be + 32 * j);
        limbs_from_be<8>(b.l, b_be + 32 * j);
        f_to_mont(a, a);
        f_to_mont(b, b);
        f_mul_inline(r, a, b);
        f_from_mont(r, r);
        limbs_to_be<8>(out_be + 32 * j, r.l);
        return;
    }
    uint32_t wa[8], wb[8], wr[8];
    limbs_from_be<8>(wa, a_be + 32 * j);
    limbs_from_be<8>(wb, b_be + 32 * j);
    fr9_t a, b, r;
    fr9_from_words(a, wa);
    fr9_from_words(b, wb);
    fr9_to_mont(a, a);
    fr9_to_mont(b, b);
    if (op == 0) fr9_mul(r, a, b);
    else if (op == 1) fr9_add(r, a, b);
    else if (op == 2) fr9_sub4(r, a, b);
    else if (op == 5 || op == 6) {
        // k (a + b) summed lazily, k = 1 + (a mod 29) <= 29: a value up to 58 r with limbs up to 2^30.9, then the
        // product-free reductions: 5 = fr9_reduce (canonical), 6 = fr9_reduce_approx alone (< 2r)
        const uint32_t k = 1u + (uint32_t)((((uint64_t)wa[1] << 32) | wa[0]) % 29u);   // test side: (a mod 2^64) mod 29
        fr9_t s;
        fr9_add(s, a, b);
        r = s;
        for (uint32_t i = 1; i < k; i++) {
            fr9_add(r, r, s);
            if ((i & 1) == 0) fr9_norm(r, r);        // limbs stay below 2^31
        }
        if (op == 5) fr9_reduce(r, r);
        else {
            fr9_norm(r, r);
            fr9_reduce_approx(r, r);
        }
    }
    else fr9_mul(r, a, a);
    fr9_from_mont(r, r);
    fr9_to_words(wr, r);
    limbs_to_be<8>(out_be + 32 * j, wr);
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

// lane-parallel point operations (fp_lp.hip.h), one wave per element: op 5 = 2a + b (full addition of two XYZZ points),
// 6 = 2 * (2a), 7 = ten rounds of r <- 2r + b starting from a (class invariants across a long dependent chain)
__global__ void __launch_bounds__(64) k_test_g1_lp(int op, const uint8_t* a_be, const uint8_t* b_be, uint8_t* out_be) {
    __shared__ LpScratch sm;
    __shared__ g1_xyzz_t pa, pb, r;
    const uint64_t j = blockIdx.x;
    const LpLane k = lp_lane();
    if (threadIdx.x == 0) {
        g1_aff28 a, b;
        fp_from_be48(a.x, a_be + 96 * j); fp_from_be48(a.y, a_be + 96 * j + 48);
        fp_from_be48(b.x, b_be + 96 * j); fp_from_be48(b.y, b_be + 96 * j + 48);
        g1_xyzz_t ta, tb, t2;
        g1_from_aff(ta, a);
        g1_from_aff(tb, b);
        g1_dbl(t2, ta);
        store_xyzz(&pa, op == 7 ? ta : t2);
        store_xyzz(&pb, tb);
    }
    __syncthreads();
    if (op == 5) lp_add(sm, &r, &pa, &pb, k);
    else if (op == 6) lp_dbl(sm, &r, &pa, k);
    else {
        for (int it = 0; it < 10; it++) {
            lp_dbl(sm, &pa, &pa, k);
            lp_add(sm, &pa, &pa, &pb, k);
        }
        if (threadIdx.x < 56) reinterpret_cast<uint32_t*>(&r)[threadIdx.x] = reinterpret_cast<uint32_t*>(&pa)[threadIdx.x];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        g1_xyzz_t v;
        load_xyzz(v, &r);
        g1_aff28 o;
        g1_to_aff(o, v);
        fp_to_be48(out_be + 96 * j, o.x);
        fp_to_be48(out_be + 96 * j + 48, o.y);
    }
}

// ---- the library's own collective (kzg_comm_*, kzg_msm_sharded)
// drops the communicator: destroyed when healthy, aborted when a collective failed or timed out on it (ncclCommDestroy
// would wait for operations that will never complete).  Callers hold every lane (or are tearing the context down).
void comm_teardown(kzg_ctx* ctx) {
    std::lock_guard<std::mutex> lk(ctx->comm.mu);
    if (ctx->comm.comm) {
        std::string err;
        if (const kzg_rccl::Api* r = kzg_rccl::api(&err)) {
            if (ctx->comm.broken) (void)r->CommAbort(ctx->comm.comm);
            else (void)r->CommDestroy(ctx->comm.comm);
        }
        ctx->comm.comm = nullptr;
    }
    ctx->comm.rank = ctx->comm.world = 0;
    ctx->comm.broken = false;
    ctx->comm.why.clear();
}
// test hook (kzg_test_comm_stall): keeps one wave busy for `ticks` of the constant-rate wall clock, or 2^31 polls at most
__global__ void __launch_bounds__(64) k_test_stall(uint64_t ticks) {
    if (threadIdx.x) return;
    const uint64_t t0 = wall_clock64();
    for (uint32_t i = 0; i < 0x7fffffffu && wall_clock64() - t0 < ticks; i++) __builtin_amdgcn_s_sleep(32);
}
// finish() with a bounded wait: the lane's stream holds a collective whose peers are not ours to trust.  Polls the pinned
// sequence word the publish sets (busy for the first 100 us, then yielding, then in 50-us sleeps).  The budget counts from
// the moment `coll_start` (an event recorded right in front of the collective) has fired: it bounds the COLLECTIVE, not
// this rank's own MSM in front of it -- so when it expires the collective is what the stream is executing, which is the
// state ncclCommAbort is made for.  *timed_out: the caller aborts the communicator, which releases the stream.
int finish_bounded(kzg_ctx* ctx, Lane& L, int timeout_ms, hipEvent_t coll_start, bool* timed_out) {
    *timed_out = false;
    if (timeout_ms <= 0) return finish(ctx, L, false);
    prof_close(ctx, L);
    const uint32_t seq = ++L.pub_seq;
    launch_publish(L.stream, L.tail, L.pin_dev, TB_COPY, L.flags(), reinterpret_cast<uint32_t*>(L.pin_dev + PIN_SEQ), seq);
    HIPCHK(ctx, hipGetLastError());
    const volatile uint32_t* w = reinterpret_cast<const volatile uint32_t*>(L.pin + PIN_SEQ);
    const auto t0 = std::chrono::steady_clock::now();
    auto t_coll = t0;
    bool coll_running = false;
    for (uint32_t spin = 0; *w != seq; spin++) {
        if ((spin & 0x3f) != 0x3f) {
            __builtin_ia32_pause();
            continue;
        }
        const auto now = std::chrono::steady_clock::now();
        if (!coll_running) {
            const hipError_t e = hipEventQuery(coll_start);
            if (e == hipSuccess) {
                coll_running = true;
                t_coll = now;
            } else if (e != hipErrorNotReady) {
                return fail(ctx, KZG_E_HIP, std::string("kzg_msm_sharded: ") + hipGetErrorString(e));
            }
            (void)hipGetLastError();
        } else if (now - t_coll > std::chrono::milliseconds(timeout_ms)) {
            *timed_out = true;
            return fail(ctx, KZG_E_COMM, "kzg_msm_sharded: the all_gather did not complete within " + std::to_string(timeout_ms) +
                                             " ms (a peer is dead or late); the communicator has been aborted");
        }
        const auto dt = now - t0;
        if (dt > std::chrono::milliseconds(5)) std::this_thread::sleep_for(std::chrono::microseconds(50));
        else if (dt > std::chrono::microseconds(100)) std::this_thread::yield();
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    L.flags_clean = true;
    if (ctx->profiling) HIPCHK(ctx, hipStreamSynchronize(L.stream));    // the stage events behind the publish
    prof_end(ctx, L);
    const uint32_t* f = reinterpret_cast<const uint32_t*>(L.pin + TB_FLAGS);
    if (f[0]) return fail(ctx, KZG_E_SCALAR, "non-canonical Fr scalar (>= r)");
    if (f[1]) return fail(ctx, KZG_E_POINT, "G1 input not reduced, not on the curve or outside the prime-order subgroup");
    return KZG_OK;
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

// The HIP runtime spreads a process's streams over GPU_MAX_HW_QUEUES hardware queues (default 4), and kernels of two
// streams that share a queue execute in order.  A context has four lanes (a stream + a verification side stream each), an
// auxiliary and a copy stream: on four queues lanes collide -- which lanes depends on the order in which every stream of the
// PROCESS was created (round 4's `pipelined` regression: DESIGN.md 3.3).  With eight queues the four lanes run concurrently
// whatever that order: requests/s with 4 host threads 4961 -> 6700 at 2^12 and 1640 -> 1967 at 2^16, two MSMs in flight
// overlap in either stream order, single requests unchanged (profiles/r05_ab_hw_queues.log).  The variable is read when the
// HIP runtime initialises, i.e. at the first HIP call of the process: setting it here -- when this library is loaded, only if
// the user has not set it -- is in time unless something else has already used HIP (then the caller exports it itself:
// INTEGRATION.md section 4; zkp_subnet_amd/_native.py and bench.py set it before they touch HIP).
__attribute__((constructor)) static void kzg_default_hw_queues() { setenv("GPU_MAX_HW_QUEUES", "8", /*overwrite=*/0); }

// =====================================================================================================
extern "C" {

const char* kzg_version(void) { return KZG_VERSION; }

int kzg_create(int device_id, kzg_ctx** out) {
    if (!out) return KZG_E_ARG;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device_id < 0 || device_id >= count)
        return fail(nullptr, KZG_E_HIP, "no usable HIP device with that index");
    if (hipSetDevice(device_id) != hipSuccess) return fail(nullptr, KZG_E_HIP, "hipSetDevice failed");
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) != hipSuccess) return fail(nullptr, KZG_E_HIP, "hipGetDeviceProperties failed");
    if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0)
        return fail(nullptr, KZG_E_HIP, "this library is built for gfx950 (MI355X) only");
    kzg_ctx* ctx = new kzg_ctx();
    ctx->device = device_id;
    if (const char* e = getenv("KZG_SERIAL_ACC")) ctx->serial_accumulate = e[0] != '0';
    bool ok = hipStreamCreateWithFlags(&ctx->aux, hipStreamNonBlocking) == hipSuccess &&
              hipHostMalloc((void**)&ctx->aux_pin, 256, hipHostMallocDefault) == hipSuccess;
    for (int k = 0; ok && k < N_STAGE; k++)
        ok = hipEventCreateWithFlags(&ctx->stage[k].ev, hipEventDisableTiming) == hipSuccess;
    for (int l = 0; ok && l < N_LANES; l++) {
        Lane& L = ctx->lane[l];
        L.index = l;
        ok = hipStreamCreateWithFlags(&L.stream, hipStreamNonBlocking) == hipSuccess &&
             hipMalloc((void**)&L.tail, TB_SIZE) == hipSuccess && hipMemset(L.tail, 0, TB_SIZE) == hipSuccess &&
             // coherent (fine-grained) and mapped: k_publish stores results straight into this page
             hipHostMalloc((void**)&L.pin, 4096, hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess &&
             hipHostGetDevicePointer((void**)&L.pin_dev, L.pin, 0) == hipSuccess &&
             // the host trusts this page on the sole basis of its sequence words (PIN_SEQ / PIN_SEQ_SORT == the lane's
             // counters, which start at 1): recycled host memory must not carry a stale equal word
             (memset(L.pin, 0, 4096), true) &&
             hipEventCreateWithFlags(&L.ev_sorted, hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&L.ev_done, hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&L.ev_coeffs, hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&L.ev_ext, hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&L.ev_acc, hipEventDisableTiming) == hipSuccess &&
             hipStreamCreateWithFlags(&L.vstream, hipStreamNonBlocking) == hipSuccess &&
             hipEventCreateWithFlags(&L.ev_verify, hipEventDisableTiming) == hipSuccess;
    }
    // The copy stream of kzg_staging_flush is created LAST.  The runtime spreads streams over its few hardware queues in
    // creation order; created between `aux` and the lanes' streams (as round 4's 1b467d4 did) it moved lanes 0 and 1 --
    // the two that carry two MSMs in flight -- onto one queue, and the second request's sort and accumulate no longer
    // overlapped the first one's latency-bound tail: `pipelined` 2.44-2.46 ms per MSM before that commit, 2.64-2.73 with
    // it, 2.44-2.46 again with the stream created here (same box, three rounds: profiles/r05_ab_pipelined_bisect.log).
    ok = ok && hipStreamCreateWithFlags(&ctx->h2d, hipStreamNonBlocking) == hipSuccess;
    if (!ok) {
        kzg_destroy(ctx);
        return fail(nullptr, KZG_E_HIP, "stream / event / buffer creation failed");
    }
    *out = ctx;
    return KZG_OK;
}

void kzg_destroy(kzg_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    comm_teardown(ctx);     // before the lanes' streams go: a communicator holds kernels and proxies on them
    for (Lane& L : ctx->lane) {
        if (L.stream) (void)hipStreamSynchronize(L.stream);
        if (L.vstream) (void)hipStreamSynchronize(L.vstream);
        for (DevBuf* b : {&L.vbuf, &L.rank, &L.sorted, &L.hist, &L.offsets, &L.bufA, &L.bufB, &L.bufC, &L.bufD, &L.carries, &L.carry_key,
                          &L.in_be, &L.scal, &L.coeffA, &L.coeffB, &L.qbuf, &L.hbuf, &L.hnext, &L.out_be, &L.ntt_mid, &L.gather,
                          &L.comm_send, &L.comm_recv})
            b->release();
        for (hipEvent_t e : L.ev_pool) (void)hipEventDestroy(e);
        for (hipEvent_t e : {L.ev_sorted, L.ev_done, L.ev_coeffs, L.ev_ext, L.ev_acc, L.ev_verify})
            if (e) (void)hipEventDestroy(e);
        if (L.vstream) (void)hipStreamDestroy(L.vstream);
        if (L.tail) (void)hipFree(L.tail);
        if (L.pin) (void)hipHostFree(L.pin);
        if (L.stream) (void)hipStreamDestroy(L.stream);
    }
    if (ctx->aux) {
        (void)hipStreamSynchronize(ctx->aux);
        (void)hipStreamDestroy(ctx->aux);
    }
    if (ctx->aux_pin) (void)hipHostFree(ctx->aux_pin);
    if (ctx->h2d) {
        (void)hipStreamSynchronize(ctx->h2d);
        (void)hipStreamDestroy(ctx->h2d);
    }
    for (Stage& st : ctx->stage) {
        if (st.p) (void)hipHostFree(st.p);
        if (st.ev) (void)hipEventDestroy(st.ev);
        st.twin.release();
    }
    delete ctx;  // the remaining DevBufs (table, slots, twiddles, aux) free themselves
}

// message of the last failing call made by THIS thread (valid until its next failing call)
const char* kzg_last_error(kzg_ctx*) { return tl_err.c_str(); }

int kzg_set_window(kzg_ctx* ctx, int c) {
    if (!ctx) return KZG_E_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    if (c != 0 && (c < 4 || c > 24)) return fail(ctx, KZG_E_ARG, "window bits must be 0 (auto) or in [4, 24]");
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
int kzg_set_host_finish(kzg_ctx* ctx, int enable) {
    if (!ctx) return KZG_E_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    ctx->host_finish = enable != 0;
    return KZG_OK;
}

// copies [src, src + bytes) with up to four threads: a setup file in the page cache is read at memory speed, not at one
// core's memcpy speed
static void copy_parallel(uint8_t* dst, const uint8_t* src, size_t bytes) {
    const size_t min_piece = (size_t)4 << 20;
    const unsigned parts = (unsigned)std::min<size_t>(4, std::max<size_t>(1, bytes / min_piece));
    if (parts <= 1) {
        memcpy(dst, src, bytes);
        return;
    }
    std::vector<std::thread> th;
    const size_t piece = ((bytes / parts) + 4095) & ~(size_t)4095;
    for (unsigned t = 1; t < parts; t++) {
        const size_t off = (size_t)t * piece;
        if (off >= bytes) break;
        const size_t len = std::min(piece, bytes - off);
        th.emplace_back([=] { memcpy(dst + off, src + off, len); });
    }
    memcpy(dst, src, std::min(piece, bytes));
    for (auto& t : th) t.join();
}
// reads [off, off + bytes) of `fd` into dst with up to four threads of pread(2): a setup file in the page cache arrives at
// memory speed, and a file truncated or replaced under the load is a short read / errno here -- a status code -- where
// a mapping would have raised SIGBUS in the miner process.  false: I/O error or end of file before `bytes`.
// 0: all of it arrived; > 0: the errno of the failing pread (each reader thread has its OWN errno: the value travels in
// the return code, never through the caller's thread-local); -1: end of file before `bytes` (the file shrank).
static int pread_full(int fd, uint8_t* dst, size_t bytes, off_t off) {
    while (bytes) {
        const ssize_t r = pread(fd, dst, bytes, off);
        if (r < 0 && errno == EINTR) continue;
        if (r < 0) return errno ? errno : EIO;
        if (r == 0) return -1;
        dst += r;
        off += r;
        bytes -= (size_t)r;
    }
    return 0;
}
static int read_parallel(int fd, uint8_t* dst, size_t bytes, off_t off) {   // same codes as pread_full: the first failure
    const size_t min_piece = (size_t)4 << 20;
    const unsigned parts = (unsigned)std::min<size_t>(4, std::max<size_t>(1, bytes / min_piece));
    if (parts <= 1) return pread_full(fd, dst, bytes, off);
    const size_t piece = ((bytes / parts) + 4095) & ~(size_t)4095;
    std::atomic<int> err{0};
    auto note = [&err](int rc) {
        int none = 0;
        if (rc) err.compare_exchange_strong(none, rc);
    };
    std::vector<std::thread> th;
    unsigned started = 1;
    try {
        for (unsigned t = 1; t < parts && (size_t)t * piece < bytes; t++, started++) {
            const size_t o = (size_t)t * piece, len = std::min(piece, bytes - o);
            th.emplace_back([=, &note] { note(pread_full(fd, dst + o, len, off + (off_t)o)); });
        }
    } catch (const std::system_error&) {   // no thread to be had: the caller reads the rest itself
    }
    note(pread_full(fd, dst, std::min(piece, bytes), off));
    for (unsigned t = started; t < parts && (size_t)t * piece < bytes; t++) {
        const size_t o = (size_t)t * piece;
        note(pread_full(fd, dst + o, std::min(piece, bytes - o), off + (off_t)o));
    }
    for (auto& t : th) t.join();
    return err.load();
}
// The points of a setup file / caller buffer -> a NEW window-0 table, tile by tile through two pinned staging buffers: the
// host fills buffer b (pread from the setup file, or a copy of the caller's memory) while the GPU still copies and decodes buffer 1 - b;
// the only host waits are for a buffer to come free.  Then the window tables; then the swap.
static int load_srs_common(kzg_ctx* ctx, const uint8_t* data, int fd, uint64_t n_points, int scale, int machines_scale,
                           bool compressed) {
    if (!ctx || (!data && fd < 0)) return KZG_E_ARG;
    bool subgroup_check;
    {   // the opt-out is for ONE load: taken (and re-armed) here, whatever becomes of this call
        std::lock_guard<std::mutex> lk(ctx->mu);
        subgroup_check = ctx->srs_subgroup_check;
        ctx->srs_subgroup_check = true;
    }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    using clk = std::chrono::steady_clock;
    const auto t_begin = clk::now();
    double host_copy_s = 0, wait_s = 0;
    LaneHold H(ctx);
    if (int rc = H.take_all()) return rc;
    Lane& L = H.L();
    TableSpec sp;
    int rc = plan_table(ctx, n_points, scale, machines_scale, sp);
    if (rc) return rc;
    DevBuf nt;
    rc = alloc_new_table(ctx, sp, nt);
    if (rc) return rc;
    DrainGuard drain{L.stream};
    rc = clear_flags(ctx, L);
    if (rc) return rc;
    const uint64_t tile = (uint64_t)1 << 18;
    const size_t rec = compressed ? 48 : 96;
    const uint64_t tile_pts = n_points < tile ? n_points : tile;
    struct PinPair {
        uint8_t* p[2] = {nullptr, nullptr};
        hipEvent_t ev[2] = {nullptr, nullptr};
        bool used[2] = {false, false};
        ~PinPair() {
            for (int b = 0; b < 2; b++) {
                if (ev[b]) (void)hipEventDestroy(ev[b]);
                if (p[b]) (void)hipHostFree(p[b]);
            }
        }
    } pin;   // (dies before `drain` runs: every path below that leaves with copies in flight drains the stream itself first)
    const int nbuf = n_points > tile ? 2 : 1;
    for (int b = 0; b < nbuf; b++) {
        HIPCHK(ctx, hipHostMalloc((void**)&pin.p[b], tile_pts * rec, hipHostMallocDefault));
        HIPCHK(ctx, hipEventCreateWithFlags(&pin.ev[b], hipEventDisableTiming));
    }
    DevBuf dev_in[2];
    for (int b = 0; b < nbuf; b++) HIPCHK(ctx, dev_in[b].ensure(tile_pts * rec));
    g1_affine_t* table = nt.as<g1_affine_t>();
    hipError_t err = hipSuccess;
    uint64_t t_idx = 0;
    for (uint64_t first = 0; first < n_points && err == hipSuccess; first += tile, t_idx++) {
        const int b = (int)(t_idx & 1) % nbuf;
        const uint64_t cnt = n_points - first < tile ? n_points - first : tile;
        if (pin.used[b]) {
            const auto w0 = clk::now();
            err = hipEventSynchronize(pin.ev[b]);     // the H2D copy that last read this buffer has completed
            wait_s += std::chrono::duration<double>(clk::now() - w0).count();
            if (err != hipSuccess) break;
        }
        const auto c0 = clk::now();
        if (data) copy_parallel(pin.p[b], data + rec * first, cnt * rec);
        else if (const int e = read_parallel(fd, pin.p[b], cnt * rec, (off_t)(rec * first))) {
            (void)hipStreamSynchronize(L.stream);     // copies of the other buffer may still be in flight
            return fail(ctx, KZG_E_ARG, e < 0 ? std::string("setup file: the file shrank during the load (end of file before the last point)")
                                              : std::string("setup file: read failed (") + strerror(e) + ")");
        }
        host_copy_s += std::chrono::duration<double>(clk::now() - c0).count();
        err = hipMemcpyAsync(dev_in[b].p, pin.p[b], cnt * rec, hipMemcpyHostToDevice, L.stream);
        if (err != hipSuccess) break;
        err = hipEventRecord(pin.ev[b], L.stream);
        pin.used[b] = true;
        if (compressed) launch_srs_from_c48(L.stream, dev_in[b].as<uint8_t>(), table + first, cnt, L.flags() + 1);
        else launch_srs_from_be96(L.stream, dev_in[b].as<uint8_t>(), table + first, cnt, L.flags() + 1);
        // on the curve is not in G1 (cofactor ~2^126): every point of the file is put through the endomorphism test
        if (subgroup_check) launch_g1_subgroup_check_bulk(L.stream, table + first, cnt, L.flags() + 1);
    }
    if (err != hipSuccess) {
        (void)hipStreamSynchronize(L.stream);
        return fail(ctx, KZG_E_HIP, std::string("SRS upload: ") + hipGetErrorString(err));
    }
    const auto w0 = clk::now();
    rc = finish(ctx, L);        // drains the stream; a bad point in ANY tile has raised the flag by now
    wait_s += std::chrono::duration<double>(clk::now() - w0).count();
    if (rc) return rc;          // the previous table (if any) keeps serving
    const auto p0 = clk::now();
    rc = precompute_tables(ctx, sp, table);
    if (rc) return rc;
    const double tables_s = std::chrono::duration<double>(clk::now() - p0).count();
    drain.armed = false;
    install_table(ctx, sp, nt);
    {
        std::lock_guard<std::mutex> lk(ctx->mu);
        ctx->load_stats[0] = host_copy_s;
        ctx->load_stats[1] = wait_s;
        ctx->load_stats[2] = tables_s;
        ctx->load_stats[3] = std::chrono::duration<double>(clk::now() - t_begin).count();
    }
    H.clean = true;
    return KZG_OK;
}
int kzg_load_srs(kzg_ctx* ctx, const uint8_t* g1_affine_be96, uint64_t n_points, int scale, int machines_scale) {
    return load_srs_common(ctx, g1_affine_be96, -1, n_points, scale, machines_scale, false);
}
int kzg_load_srs_compressed(kzg_ctx* ctx, const uint8_t* g1_c48, uint64_t n_points, int scale, int machines_scale) {
    return load_srs_common(ctx, g1_c48, -1, n_points, scale, machines_scale, true);
}
// The reference's start path: `Client(setup_path=...).start(scale, machines_scale)` hands the prover a FILE
// (base/miner.py:75-84, Makefile:63-74: setup_24_8.uncompressed = 2^24 points, 1.6 GB).  The file is read with pread(2)
// straight into the pinned tiles (page cache -> pinned memory, one copy, as a mapping would give) so that a file that is
// truncated or replaced while a multi-GB load runs fails the call instead of raising SIGBUS.
int kzg_load_srs_file(kzg_ctx* ctx, const char* path, int compressed, int scale, int machines_scale) {
    if (!ctx || !path) return KZG_E_ARG;
    const int fd = open(path, O_RDONLY | O_CLOEXEC);
    if (fd < 0) return fail(ctx, KZG_E_ARG, std::string("cannot open setup file ") + path + ": " + strerror(errno));
    struct stat st;
    if (fstat(fd, &st) != 0 || st.st_size <= 0) {
        close(fd);
        return fail(ctx, KZG_E_ARG, std::string("setup file ") + path + " is empty or unreadable");
    }
    const size_t rec = compressed ? 48 : 96;
    if ((size_t)st.st_size % rec) {
        close(fd);
        return fail(ctx, KZG_E_ARG, "setup file must be a whole number of " + std::to_string(rec) + "-byte G1 points");
    }
    (void)posix_fadvise(fd, 0, st.st_size, POSIX_FADV_SEQUENTIAL);
    (void)posix_fadvise(fd, 0, st.st_size, POSIX_FADV_WILLNEED);
    const int rc = load_srs_common(ctx, nullptr, fd, (uint64_t)st.st_size / rec, scale, machines_scale, compressed != 0);
    close(fd);
    return rc;
}
// seconds of the last successful kzg_load_srs*: [0] host copies file/buffer -> pinned tiles, [1] host waits for the GPU
// (upload + decode / decompression), [2] window-table build, [3] the whole call
int kzg_set_srs_subgroup_check(kzg_ctx* ctx, int enable) {
    if (!ctx) return KZG_E_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    ctx->srs_subgroup_check = enable != 0;
    return KZG_OK;
}
// The rate that bounds the accumulate kernel, measured now, on this device (csrc/calibrate.hip).  Exclusive: waits for
// the lanes to be idle so that nothing shares the SIMDs with the measurement.
int kzg_calibrate(kzg_ctx* ctx, int waves_per_simd, double out[6]) {
    if (!ctx || !out) return KZG_E_ARG;
    if (waves_per_simd < 1 || waves_per_simd > 8) return fail(ctx, KZG_E_ARG, "waves_per_simd must be in [1, 8]");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipDeviceProp_t prop;
    HIPCHK(ctx, hipGetDeviceProperties(&prop, ctx->device));
    const uint32_t cus = (uint32_t)prop.multiProcessorCount, simds = 4 * cus;
    LaneHold H(ctx);
    if (int rc = H.take_all()) return rc;
    Lane& L = H.L();
    const uint32_t blocks = cus * (uint32_t)waves_per_simd;      // 256 threads = 4 waves = one per SIMD of a CU
    HIPCHK(ctx, L.out_be.ensure((2 + (size_t)blocks * 256) * sizeof(uint64_t)));
    uint64_t* d = L.out_be.as<uint64_t>();
    hipEvent_t e0 = nullptr, e1 = nullptr;
    HIPCHK(ctx, hipEventCreate(&e0));
    if (hipEventCreate(&e1) != hipSuccess) { (void)hipEventDestroy(e0); return fail(ctx, KZG_E_HIP, "hipEventCreate"); }
    // ~1.5 ms at 2 waves per SIMD (2.3 ns per wave-instruction per SIMD); a short launch first pages the code in
    const uint32_t iters = 40960u * 2u / (uint32_t)std::max(2, waves_per_simd);
    launch_calibrate_mad(L.stream, d, blocks, iters / 16);
    (void)hipEventRecord(e0, L.stream);
    launch_calibrate_mad(L.stream, d, blocks, iters);
    (void)hipEventRecord(e1, L.stream);
    uint64_t ticks = 0;
    hipError_t err = hipMemcpyAsync(&ticks, d, sizeof(ticks), hipMemcpyDeviceToHost, L.stream);
    if (err == hipSuccess) err = hipStreamSynchronize(L.stream);
    float ms = 0;
    if (err == hipSuccess) err = hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (err != hipSuccess) return fail(ctx, KZG_E_HIP, std::string("kzg_calibrate: ") + hipGetErrorString(err));
    const double inst_per_wave = (double)iters * calibrate_unroll();
    const double inst_per_simd = inst_per_wave * waves_per_simd;
    out[0] = (double)ms * 1e6 / inst_per_simd;                   // ns per v_mad_u64_u32 wave-instruction per SIMD
    out[1] = (double)simds / out[0];                             // G wave-mads per second, whole chip
    out[2] = ms > 0 ? (double)ticks / ((double)ms * 1e6) : 0;    // s_memtime ticks per ns over the launch (wave 0)
    out[3] = (double)ms;
    out[4] = (double)simds;
    out[5] = (double)ticks / inst_per_wave;                      // ticks per instruction of ONE wave
    H.clean = true;
    return KZG_OK;
}
int kzg_get_load_stats(kzg_ctx* ctx, double out_s[4]) {
    if (!ctx || !out_s) return KZG_E_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    for (int i = 0; i < 4; i++) out_s[i] = ctx->load_stats[i];
    return KZG_OK;
}

int kzg_gen_srs(kzg_ctx* ctx, const uint8_t tau_be32[32], const uint8_t* s0_be32, uint32_t n_slices, int scale,
                int machines_scale) {
    if (!ctx || !tau_be32 || !s0_be32 || !n_slices) return KZG_E_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (machines_scale < 0 || scale < machines_scale || scale - machines_scale > 30)
        return fail(ctx, KZG_E_ARG, "bad scale / machines_scale");
    LaneHold H(ctx);
    if (int rc = H.take_all()) return rc;
    Lane& L = H.L();
    const uint64_t T = (uint64_t)1 << (scale - machines_scale);
    TableSpec sp;
    int rc = plan_table(ctx, (uint64_t)n_slices * T, scale, machines_scale, sp);
    if (rc) return rc;
    DevBuf nt;
    rc = alloc_new_table(ctx, sp, nt);
    if (rc) return rc;
    DrainGuard drain{L.stream};
    rc = clear_flags(ctx, L);
    if (rc) return rc;
    DevBuf gtab, tmp, sc;
    HIPCHK(ctx, gtab.ensure(32 * 255 * sizeof(g1_affine_t)));
    const uint64_t tile = T < ((uint64_t)1 << 20) ? T : ((uint64_t)1 << 20);
    HIPCHK(ctx, tmp.ensure(tile * (sizeof(g1_xyzz_t) + 32) + 256));
    HIPCHK(ctx, sc.ensure(((size_t)n_slices + 1) * 32 + 64));
    // tau and the per-slice factors, Montgomery form, on device
    uint32_t* tau_m = sc.as<uint32_t>();
    HIPCHK(ctx, L.in_be.ensure(((size_t)n_slices + 1) * 32));
    HIPCHK(ctx, hipMemcpyAsync(L.in_be.p, tau_be32, 32, hipMemcpyHostToDevice, L.stream));
    HIPCHK(ctx, hipMemcpyAsync(L.in_be.as<uint8_t>() + 32, s0_be32, (size_t)n_slices * 32, hipMemcpyHostToDevice, L.stream));
    launch_fr_from_be(L.stream, L.in_be.as<uint8_t>(), tau_m, (uint64_t)n_slices + 1, 1, L.flags());
    for (uint32_t k = 0; k < n_slices; k++) {
        for (uint64_t first = 0; first < T; first += tile) {
            uint64_t cnt = T - first < tile ? T - first : tile;
            launch_srs_generate(L.stream, nt.as<g1_affine_t>() + (uint64_t)k * T + first, cnt, first, tau_m,
                                tau_m + 8 * (1 + (uint64_t)k), gtab.as<g1_affine_t>(), tmp.as<g1_xyzz_t>(),
                                k == 0 && first == 0);
        }
    }
    rc = finish(ctx, L);  // synchronises: gtab / tmp / sc are idle when they go out of scope
    if (rc) return rc;
    HIPCHK(ctx, hipGetLastError());
    rc = precompute_tables(ctx, sp, nt.as<g1_affine_t>());
    if (rc) return rc;
    drain.armed = false;
    install_table(ctx, sp, nt);
    H.clean = true;
    return KZG_OK;
}

static int srs_read_common(kzg_ctx* ctx, int w, uint64_t first, uint64_t count, uint8_t* out, bool compressed) {
    if (!ctx || !out) return KZG_E_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    LaneHold H(ctx);
    if (int rc = H.take()) return rc;
    Lane& L = H.L();
    int rc = need_srs(ctx);
    if (rc) return rc;
    if (w < 0 || w >= ctx->nwin || first + count > ctx->stride) return fail(ctx, KZG_E_ARG, "srs_read out of range");
    if (!count) return KZG_OK;
    const size_t rec = compressed ? 48 : 96;
    HIPCHK(ctx, L.out_be.ensure(count * rec));
    const g1_affine_t* src = ctx->table.as<g1_affine_t>() + (uint64_t)w * ctx->stride + first;
    if (compressed) launch_srs_to_c48(L.stream, src, L.out_be.as<uint8_t>(), count);
    else launch_srs_to_be96(L.stream, src, L.out_be.as<uint8_t>(), count);
    HIPCHK(ctx, hipMemcpyAsync(out, L.out_be.p, count * rec, hipMemcpyDeviceToHost, L.stream));
    HIPCHK(ctx, hipStreamSynchronize(L.stream));
    H.clean = true;
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
    HIPCHK(ctx, hipSetDevice(ctx->device));
    LaneHold H(ctx);
    if (int rc = H.take()) return rc;
    Lane& L = H.L();
    int rc = need_srs(ctx);
    if (rc) return rc;
    prof_begin(ctx, L);
    rc = clear_flags(ctx, L);
    if (rc) return rc;
    HIPCHK(ctx, L.scal.ensure(n * 32 + 32));
    rc = upload_fr(ctx, L, scalars_be32, n, L.scal.as<uint32_t>(), 0);
    if (rc) return rc;
    rc = msm_core(ctx, L, L.scal.as<uint32_t>(), 0, n, srs_offset, L.res());
    if (rc) return rc;
    if (partial) queue_pack(ctx, L);
    else queue_encode(ctx, L, true, false);
    rc = finish(ctx, L);
    if (rc) return rc;
    if (partial) result_partial(ctx, L, out);
    else result_c48(ctx, L, 0, out);
    H.clean = true;
    return KZG_OK;
}
int kzg_msm(kzg_ctx* ctx, const uint8_t* scalars_be32, uint64_t n, uint64_t srs_offset, uint8_t out48[48]) {
    return msm_host_common(ctx, scalars_be32, n, srs_offset, out48, false);
}
int kzg_msm_partial(kzg_ctx* ctx, const uint8_t* scalars_be32, uint64_t n, uint64_t srs_offset,
                    uint8_t out_xyzz192[192]) {
    return msm_host_common(ctx, scalars_be32, n, srs_offset, out_xyzz192, true);
}

// Sums of a few points run on their own stream and buffers: legal while MSM tickets are outstanding (a rank sums the
// gathered partials of step i while its step i+1 is already on the GPU).  Three input forms:
//   host partials (192-byte XYZZ records), device partials (the output tensor of an all_gather; every prior writer has
//   completed), and 48-byte compressed points (the commitments of the worker rows: Pianist's master aggregation
//   sum_i commit_i, reference neurons/validator.py:196-198, README.md:38) which are decompressed on the GPU.
static int g1_sum_common(kzg_ctx* ctx, const uint8_t* in, uint32_t count, uint8_t out48[48], int form) {
    if (!ctx || !out48 || (count && !in)) return KZG_E_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    std::lock_guard<std::mutex> lk(ctx->aux_mu);
    const size_t rec = form == 2 ? 48 : 192;
    HIPCHK(ctx, ctx->aux_in.ensure((size_t)count * rec + 192));
    HIPCHK(ctx, ctx->aux_pts.ensure(((size_t)count + 2) * sizeof(g1_xyzz_t)));
    HIPCHK(ctx, ctx->aux_out.ensure(256));
    hipStream_t s = ctx->aux;
    g1_xyzz_t* pts = ctx->aux_pts.as<g1_xyzz_t>();
    uint32_t* bad = ctx->aux_out.as<uint32_t>() + 32;
    HIPCHK(ctx, hipMemsetAsync(bad, 0, 4, s));
    if (form == 2) {
        static_assert(sizeof(g1_affine_t) <= sizeof(g1_xyzz_t), "affine rows are staged in the XYZZ scratch");
        if (count) HIPCHK(ctx, hipMemcpyAsync(ctx->aux_in.p, in, (size_t)count * 48, hipMemcpyHostToDevice, s));
        g1_affine_t* aff = reinterpret_cast<g1_affine_t*>(pts + 1);
        launch_srs_from_c48(s, ctx->aux_in.as<uint8_t>(), aff, count, bad);
        // these are UNTRUSTED points (miners' commitments): on the curve is not enough, E(Fp) has a 2^126 cofactor
        launch_g1_subgroup_check(s, aff, count, bad);
        launch_g1_sum_affine(s, aff, count, pts);
    } else {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(in);
        if (form == 0) {
            if (count) HIPCHK(ctx, hipMemcpyAsync(ctx->aux_in.p, in, (size_t)count * 192, hipMemcpyHostToDevice, s));
            src = ctx->aux_in.as<uint32_t>();
        }
        launch_xyzz_unpack(s, src, pts + 1, count);
        launch_g1_sum(s, pts + 1, count, pts);
    }
    uint8_t* pin = ctx->aux_pin;
    if (ctx->host_finish) {
        HIPCHK(ctx, hipMemcpyAsync(pin, pts, sizeof(g1_xyzz_t), hipMemcpyDeviceToHost, s));
    } else {
        launch_g1_compress(s, pts, ctx->aux_out.as<uint8_t>());
        HIPCHK(ctx, hipMemcpyAsync(pin, ctx->aux_out.p, 48, hipMemcpyDeviceToHost, s));
    }
    HIPCHK(ctx, hipMemcpyAsync(pin + 224, bad, 4, hipMemcpyDeviceToHost, s));
    HIPCHK(ctx, hipStreamSynchronize(s));
    HIPCHK(ctx, hipGetLastError());
    if (const uint32_t b = *reinterpret_cast<const uint32_t*>(pin + 224))
        return fail(ctx, KZG_E_POINT, (b & 3u) ? "compressed G1 input malformed, not reduced or not on the curve"
                                                : "G1 input on the curve but outside the prime-order subgroup");
    if (ctx->host_finish) kzg_host::xyzz_to_c48(reinterpret_cast<const uint32_t*>(pin), out48);
    else memcpy(out48, pin, 48);
    return KZG_OK;
}
int kzg_g1_sum(kzg_ctx* ctx, const uint8_t* partials_xyzz192, uint32_t count, uint8_t out48[48]) {
    return g1_sum_common(ctx, partials_xyzz192, count, out48, 0);
}
int kzg_g1_sum_dev(kzg_ctx* ctx, const void* dev_partials_xyzz192, uint32_t count, uint8_t out48[48]) {
    return g1_sum_common(ctx, reinterpret_cast<const uint8_t*>(dev_partials_xyzz192), count, out48, 1);
}
int kzg_g1_sum_compressed(kzg_ctx* ctx, const uint8_t* points_c48, uint32_t count, uint8_t out48[48]) {
    return g1_sum_common(ctx, points_c48, count, out48, 2);
}

static int commit_open_host(kzg_ctx* ctx, uint32_t i, const uint8_t* row_be32, uint64_t T, int evaluation_form,
                            const uint8_t* alpha, uint8_t* c48, uint8_t* e32, uint8_t* p48) {
    if (!ctx || !row_be32 || (p48 && (!alpha || !e32))) return KZG_E_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    LaneHold H(ctx);
    if (int rc = H.take()) return rc;
    Lane& L = H.L();
    int rc = check_worker(ctx, i, T);
    if (rc) return rc;
    prof_begin(ctx, L);
    rc = clear_flags(ctx, L);
    if (rc) return rc;
    HIPCHK(ctx, L.coeffA.ensure(T * 32));
    rc = upload_fr(ctx, L, row_be32, T, L.coeffA.as<uint32_t>(), 1);
    if (rc) return rc;
    return commit_open_dev(ctx, H, i, L.coeffA.as<uint32_t>(), T, evaluation_form, alpha, c48, e32, p48);
}
// ---- the UNCHANGED reference miner (neurons/miner.py:56-61) calls worker_commit(i, poly) and then worker_open(i, poly, x)
// with the same row: the second call used to decode, upload and inverse-transform it all over again.  With a content tag
// (a 128-bit keyed hash the codec folds into its decode pass) the coefficient vector of the last few rows stays on the
// device: a call whose (tag, T, form) is cached skips the INTT and keeps the upload off its critical path; anything else
// behaves exactly like the untagged call and leaves its own coefficients behind.  The tag (zkp_subnet_amd/csrc/wire_py.c:
// a keyed multiply-fold over the decoded bytes, fast but with no cryptographic analysis) is a HINT, not a proof of
// identity: every hit is verified bit for bit on the GPU against the row the slot was filled from.
static int rcache_lookup(kzg_ctx* ctx, const uint8_t tag[16], uint64_t T, int ef) { return ctx->book.rcache_lookup(tag, T, ef); }
static void rcache_release(kzg_ctx* ctx, int k, bool valid, const uint8_t tag[16], uint64_t T, int ef) {
    ctx->book.rcache_release(k, valid, tag, T, ef);
}
static int commit_open_host_cached(kzg_ctx* ctx, uint32_t i, const uint8_t* row_be32, uint64_t T, int evaluation_form,
                                   const uint8_t tag[16], const uint8_t* alpha, uint8_t* c48, uint8_t* e32, uint8_t* p48) {
    if (!ctx || !row_be32 || !tag || (p48 && (!alpha || !e32))) return KZG_E_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    LaneHold H(ctx);
    if (int rc = H.take()) return rc;
    Lane& L = H.L();
    int rc = check_worker(ctx, i, T);
    if (rc) return rc;
    prof_begin(ctx, L);
    rc = clear_flags(ctx, L);
    if (rc) return rc;
    int look = rcache_lookup(ctx, tag, T, evaluation_form);
    if (look >= 0) {
        // hit: no INTT, and nothing of the row on the request's critical path.  The tag is only a HINT: the caller's row
        // is uploaded on a side stream while the MSM runs and compared, bit for bit, with the row this slot was filled
        // from; the publish waits for that verdict.  A colliding tag costs one wasted pass, never a wrong answer.
        auto& e = ctx->rcache[look];
        const VerifyJob job{row_be32, T, e.raw.as<uint32_t>()};
        rc = commit_open_dev(ctx, H, i, nullptr, T, evaluation_form, alpha, c48, e32, p48, e.coef.as<uint32_t>(), nullptr, &job);
        if (rc != KZG_OK) H.drain();            // queued kernels may still read the slot
        const bool same = rc != KZG_OK || *reinterpret_cast<const volatile uint32_t*>(L.pin + TB_VERIFY) == 0;
        rcache_release(ctx, look, same, tag, T, evaluation_form);   // a slot whose tag collided is dropped
        if (same) return rc;
        ctx->book.rcache_collision();   // equal tags, different rows: the answer just computed belongs to the OTHER row --
                                        // discard it, take the miss path
        H.clean = false;
        prof_begin(ctx, L);
        rc = clear_flags(ctx, L);
        if (rc) return rc;
        look = -1;                              // no caching for this call (its tag is known to be ambiguous)
    }
    const int slot = look <= -2 ? -2 - look : -1;
    uint32_t* dst = nullptr;
    if (slot >= 0) {
        auto& e = ctx->rcache[slot];
        if (e.coef.ensure(T * 32) == hipSuccess && e.raw.ensure(T * 32) == hipSuccess) dst = e.coef.as<uint32_t>();
        else (void)hipGetLastError();
    }
    rc = L.coeffA.ensure(T * 32) == hipSuccess ? KZG_OK : fail(ctx, KZG_E_NOMEM, "row buffer");
    if (!rc) rc = upload_fr(ctx, L, row_be32, T, L.coeffA.as<uint32_t>(), 1);
    if (!rc && dst)   // the uploaded bytes themselves stay with the slot: what a later hit is verified against
        rc = hipMemcpyAsync(ctx->rcache[slot].raw.p, L.in_be_src, T * 32, hipMemcpyDeviceToDevice, L.stream) == hipSuccess
                 ? KZG_OK : fail(ctx, KZG_E_HIP, "row cache: copy of the uploaded row");
    if (!rc) rc = commit_open_dev(ctx, H, i, L.coeffA.as<uint32_t>(), T, evaluation_form, alpha, c48, e32, p48, nullptr, dst);
    if (rc != KZG_OK && slot >= 0) H.drain();   // a failed call may have kernels queued that still write the slot: not reusable before
    if (slot >= 0) rcache_release(ctx, slot, rc == KZG_OK && dst != nullptr, tag, T, evaluation_form);
    return rc;
}
int kzg_commit_cached(kzg_ctx* ctx, uint32_t i, const uint8_t* row_be32, uint64_t T, int evaluation_form,
                      const uint8_t content_tag[16], uint8_t out_commitment48[48]) {
    if (!out_commitment48) return KZG_E_ARG;
    return commit_open_host_cached(ctx, i, row_be32, T, evaluation_form, content_tag, nullptr, out_commitment48, nullptr, nullptr);
}
int kzg_open_cached(kzg_ctx* ctx, uint32_t i, const uint8_t* row_be32, uint64_t T, int evaluation_form,
                    const uint8_t content_tag[16], const uint8_t alpha_be32[32], uint8_t out_eval32[32],
                    uint8_t out_proof48[48]) {
    if (!out_proof48) return KZG_E_ARG;
    return commit_open_host_cached(ctx, i, row_be32, T, evaluation_form, content_tag, alpha_be32, nullptr, out_eval32, out_proof48);
}
int kzg_row_cache_stats(kzg_ctx* ctx, uint64_t out_hits_misses[2]) {
    if (!ctx || !out_hits_misses) return KZG_E_ARG;
    uint64_t st[3];
    ctx->book.rcache_stats(st);
    out_hits_misses[0] = st[0];
    out_hits_misses[1] = st[1];
    return KZG_OK;
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

static int ntt_dev(kzg_ctx* ctx, Lane& L, uint32_t* data, uint64_t n, int inverse) {  // in place via coeffB
    int lg = ilog2_exact(n);
    if (lg < 0) return fail(ctx, KZG_E_ARG, "NTT length must be a power of two");
    uint32_t *tw = nullptr, *invn = nullptr;
    int rc = ensure_twiddles(ctx, L, lg, inverse, &tw, inverse ? &invn : nullptr);
    if (rc) return rc;
    HIPCHK(ctx, L.coeffB.ensure(n * 32));
    HIPCHK(ctx, L.ntt_mid.ensure(n * 48));
    {
        Span sp(ctx, L, KZG_T_NTT);
        launch_fr_ntt(L.stream, data, L.coeffB.as<uint32_t>(), lg, tw, inverse ? invn : nullptr, L.ntt_mid.as<uint32_t>());
        HIPCHK(ctx, hipMemcpyAsync(data, L.coeffB.p, n * 32, hipMemcpyDeviceToDevice, L.stream));
    }
    return KZG_OK;
}
int kzg_ntt(kzg_ctx* ctx, uint8_t* inout_be32, uint64_t n, int inverse) {
    if (!ctx || !inout_be32 || !n) return KZG_E_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    LaneHold H(ctx);
    if (int rc = H.take()) return rc;
    Lane& L = H.L();
    prof_begin(ctx, L);
    int rc = clear_flags(ctx, L);
    if (rc) return rc;
    HIPCHK(ctx, L.coeffA.ensure(n * 32));
    HIPCHK(ctx, L.out_be.ensure(n * 32));
    rc = upload_fr(ctx, L, inout_be32, n, L.coeffA.as<uint32_t>(), 1);
    if (rc) return rc;
    rc = ntt_dev(ctx, L, L.coeffA.as<uint32_t>(), n, inverse);
    if (rc) return rc;
    launch_fr_to_be(L.stream, L.coeffA.as<uint32_t>(), L.out_be.as<uint8_t>(), n, 1);
    rc = finish(ctx, L);
    if (rc) return rc;
    HIPCHK(ctx, hipMemcpy(inout_be32, L.out_be.p, n * 32, hipMemcpyDeviceToHost));
    H.clean = true;
    return KZG_OK;
}
int kzg_eval(kzg_ctx* ctx, const uint8_t* coeffs_be32, uint64_t n, const uint8_t x_be32[32], uint8_t out_be32[32]) {
    if (!ctx || !x_be32 || !out_be32 || (n && !coeffs_be32)) return KZG_E_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (n == 0) {
        memset(out_be32, 0, 32);
        return KZG_OK;
    }
    LaneHold H(ctx);
    if (int rc = H.take()) return rc;
    Lane& L = H.L();
    prof_begin(ctx, L);
    int rc = clear_flags(ctx, L);
    if (rc) return rc;
    HIPCHK(ctx, L.coeffA.ensure(n * 32));
    const uint64_t nchunks = (n + 3) / 4;
    HIPCHK(ctx, L.hbuf.ensure((nchunks + (nchunks >> 1) + 64) * 32));
    HIPCHK(ctx, L.hnext.ensure((nchunks + (nchunks >> 1) + 64) * 32));
    rc = upload_fr(ctx, L, coeffs_be32, n, L.coeffA.as<uint32_t>(), 1);
    if (rc) return rc;
    uint32_t* x_m = reinterpret_cast<uint32_t*>(L.tail + TB_ALPHA_M);
    uint32_t* y_m = reinterpret_cast<uint32_t*>(L.tail + TB_Y_M);
    launch_poly_open(L.stream, L.coeffA.as<uint32_t>(), n, x_m, L.hbuf.as<uint32_t>(), L.hnext.as<uint32_t>(), y_m, nullptr,
                     x_be32, L.flags(), L.tail + TB_EVAL);
    rc = finish(ctx, L);
    if (rc) return rc;
    memcpy(out_be32, L.pin + TB_EVAL, 32);
    H.clean = true;
    return KZG_OK;
}

// y = (NTT or inverse NTT of vals)(x): the validator's per-row challenge step -- fft(poly[i], left=True, inverse=True) then
// eval(coefficients, alpha) (reference neurons/validator.py:115-118) -- without the round trip of 2^16 coefficients through
// text between the two
int kzg_ntt_eval(kzg_ctx* ctx, const uint8_t* vals_be32, uint64_t n, int inverse, const uint8_t x_be32[32],
                 uint8_t out_y32[32]) {
    if (!ctx || !vals_be32 || !n || !x_be32 || !out_y32) return KZG_E_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int lg = ilog2_exact(n);
    if (lg < 0) return fail(ctx, KZG_E_ARG, "NTT length must be a power of two");
    LaneHold H(ctx);
    if (int rc = H.take()) return rc;
    Lane& L = H.L();
    prof_begin(ctx, L);
    int rc = clear_flags(ctx, L);
    if (rc) return rc;
    HIPCHK(ctx, L.coeffA.ensure(n * 32));
    HIPCHK(ctx, L.coeffB.ensure(n * 32));
    HIPCHK(ctx, L.ntt_mid.ensure(n * 48));
    const uint64_t nchunks = (n + 3) / 4;
    HIPCHK(ctx, L.hbuf.ensure((nchunks + (nchunks >> 1) + 64) * 32));
    HIPCHK(ctx, L.hnext.ensure((nchunks + (nchunks >> 1) + 64) * 32));
    rc = upload_fr(ctx, L, vals_be32, n, L.coeffA.as<uint32_t>(), 1);
    if (rc) return rc;
    const uint32_t* coeffs = L.coeffA.as<uint32_t>();
    if (lg > 0) {
        uint32_t *tw = nullptr, *invn = nullptr;
        rc = ensure_twiddles(ctx, L, lg, inverse, &tw, inverse ? &invn : nullptr);
        if (rc) return rc;
        Span sp(ctx, L, KZG_T_NTT);
        launch_fr_ntt(L.stream, L.coeffA.as<uint32_t>(), L.coeffB.as<uint32_t>(), lg, tw, inverse ? invn : nullptr,
                      L.ntt_mid.as<uint32_t>());
        coeffs = L.coeffB.as<uint32_t>();
    }
    uint32_t* x_m = reinterpret_cast<uint32_t*>(L.tail + TB_ALPHA_M);
    uint32_t* y_m = reinterpret_cast<uint32_t*>(L.tail + TB_Y_M);
    launch_poly_open(L.stream, coeffs, n, x_m, L.hbuf.as<uint32_t>(), L.hnext.as<uint32_t>(), y_m, nullptr, x_be32, L.flags(),
                     L.tail + TB_EVAL);
    rc = finish(ctx, L);
    if (rc) return rc;
    memcpy(out_y32, L.pin + TB_EVAL, 32);
    H.clean = true;
    return KZG_OK;
}

int kzg_upload_fr(kzg_ctx* ctx, int slot, const uint8_t* be32, uint64_t n, int to_mont) {
    if (!ctx || slot < 0 || slot >= N_SLOTS || (n && !be32)) return KZG_E_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    LaneHold H(ctx);
    if (int rc = H.take_all()) return rc;   // no request may be reading the slot
    Lane& L = H.L();
    prof_begin(ctx, L);
    int rc = clear_flags(ctx, L);
    if (rc) return rc;
    ctx->slot_n[slot] = 0;
    HIPCHK(ctx, ctx->slot[slot].ensure(n * 32 + 32));
    rc = upload_fr(ctx, L, be32, n, ctx->slot[slot].as<uint32_t>(), to_mont);
    if (rc) return rc;
    rc = finish(ctx, L);
    if (rc) return rc;
    ctx->slot_n[slot] = n;
    ctx->slot_mont[slot] = to_mont ? 1 : 0;
    H.clean = true;
    return KZG_OK;
}
// dev_out != null: the 192-byte partial is left in the CALLER's device buffer (e.g. a torch tensor about to enter an
// RCCL all_gather) instead of coming back to the host
static int msm_resident_common(kzg_ctx* ctx, int slot, uint64_t n, uint64_t srs_offset, uint8_t* out, bool partial,
                               void* dev_out = nullptr) {
    if (!ctx || (!out && !dev_out) || slot < 0 || slot >= N_SLOTS) return KZG_E_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    LaneHold H(ctx);
    if (int rc = H.take()) return rc;
    Lane& L = H.L();
    int rc = need_srs(ctx);
    if (rc) return rc;
    if (n > ctx->slot_n[slot]) return fail(ctx, KZG_E_ARG, "slot holds fewer scalars than requested");
    prof_begin(ctx, L);
    rc = clear_flags(ctx, L);
    if (rc) return rc;
    rc = msm_core(ctx, L, ctx->slot[slot].as<uint32_t>(), ctx->slot_mont[slot], n, srs_offset, L.res());
    if (rc) return rc;
    if (dev_out) launch_xyzz_pack(L.stream, L.res(), reinterpret_cast<uint32_t*>(dev_out), 1);
    else if (partial) queue_pack(ctx, L);
    else queue_encode(ctx, L, true, false);
    rc = finish(ctx, L);  // synchronises the stream: dev_out is complete when the call returns
    if (rc) return rc;
    if (out) {
        if (partial) result_partial(ctx, L, out);
        else result_c48(ctx, L, 0, out);
    }
    H.clean = true;
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
// ---- the multi-GPU step chained through streams, ONE host synchronisation per MSM: the partial is queued
// on a lane and written to the caller's device tensor, the caller's stream (torch's current stream: RCCL runs behind it)
// is made to wait for it on the device; after the collective has been enqueued there, kzg_msm_sharded_finish makes the
// lane wait for that stream in turn, sums the gathered partials on the lane and returns the encoded point.  Nothing
// blocks the host in between.  (The blocking pair kzg_msm_partial_resident_dev / kzg_g1_sum_dev costs three host
// synchronisations and two copy-engine transfers: +0.18 ms on a 2.6-ms step.)
int kzg_msm_sharded_begin(kzg_ctx* ctx, int slot, uint64_t n, uint64_t srs_offset, void* dev_out_xyzz192,
                          void* consumer_stream, int* out_ticket) {
    if (!ctx || !out_ticket || !dev_out_xyzz192 || slot < 0 || slot >= N_SLOTS) return KZG_E_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int li = -1;
    int rc = lane_acquire(ctx, LANE_TICKET, &li);
    if (rc) return rc;
    Lane& L = ctx->lane[li];
    // checked while the lane is held: an exclusive operation (SRS reload, kzg_upload_fr) cannot slip in between
    rc = need_srs(ctx);
    if (!rc && n > ctx->slot_n[slot]) rc = fail(ctx, KZG_E_ARG, "slot holds fewer scalars than requested");
    if (rc) {
        lane_release(ctx, li);
        return rc;
    }
    prof_begin(ctx, L);
    rc = clear_flags(ctx, L);
    // the gathered partials get their own buffer, allocated BEFORE anything is queued: _finish must not (re)allocate
    // while the lane's MSM may still be running (hipFree synchronises the whole device)
    if (!rc && L.gather.ensure((size_t)(KZG_MAX_GATHER + 2) * sizeof(g1_xyzz_t)) != hipSuccess)
        rc = fail(ctx, KZG_E_NOMEM, "gather buffer");
    if (!rc) rc = msm_core(ctx, L, ctx->slot[slot].as<uint32_t>(), ctx->slot_mont[slot], n, srs_offset, L.res());
    if (!rc) {
        launch_xyzz_pack(L.stream, L.res(), reinterpret_cast<uint32_t*>(dev_out_xyzz192), 1);
        hipError_t e = hipEventRecord(L.ev_ext, L.stream);
        if (e == hipSuccess) e = hipStreamWaitEvent(reinterpret_cast<hipStream_t>(consumer_stream), L.ev_ext, 0);
        if (e == hipSuccess) e = hipGetLastError();
        if (e != hipSuccess) rc = fail(ctx, KZG_E_HIP, std::string("kzg_msm_sharded_begin: ") + hipGetErrorString(e));
    }
    if (rc) {
        L.sort_ws_clean = false;
        (void)hipStreamSynchronize(L.stream);
        (void)hipGetLastError();
        lane_release(ctx, li);
        return rc;
    }
    *out_ticket = li;
    return KZG_OK;
}
int kzg_msm_sharded_finish(kzg_ctx* ctx, int ticket, const void* dev_partials_xyzz192, uint32_t count,
                           void* producer_stream, uint8_t out48[48]) {
    if (!ctx || !out48 || !dev_partials_xyzz192 || !count || count > KZG_MAX_GATHER || ticket < 0 || ticket >= N_LANES) return KZG_E_ARG;
    Lane& L = ctx->lane[ticket];
    if (int rc0 = ticket_claim(ctx, ticket)) return rc0;
    (void)hipSetDevice(ctx->device);
    int rc = KZG_OK;
    hipError_t e = hipEventRecord(L.ev_ext, reinterpret_cast<hipStream_t>(producer_stream));
    if (e == hipSuccess) e = hipStreamWaitEvent(L.stream, L.ev_ext, 0);
    if (e == hipSuccess) {
        g1_xyzz_t* pts = L.gather.as<g1_xyzz_t>();    // sized by _begin
        launch_xyzz_unpack(L.stream, reinterpret_cast<const uint32_t*>(dev_partials_xyzz192), pts, count);
        launch_g1_sum(L.stream, pts, count, L.res());
        queue_encode(ctx, L, true, false);
        // no polling: the lane waits on an external producer (the collective of ALL ranks), whose time is not ours to bound
        rc = finish(ctx, L, false);
        if (!rc) result_c48(ctx, L, 0, out48);
    } else {
        rc = fail(ctx, KZG_E_HIP, std::string("kzg_msm_sharded_finish: ") + hipGetErrorString(e));
    }
    if (rc) {
        L.sort_ws_clean = false;
        (void)hipStreamSynchronize(L.stream);
        (void)hipGetLastError();
    }
    lane_release(ctx, ticket);
    return rc;
}

// ---- the collective inside the library (include/kzg_mi355x.h "the collective INSIDE the library")
int kzg_comm_unique_id(uint8_t out_id128[128]) {
    if (!out_id128) return KZG_E_ARG;
    std::string err;
    const kzg_rccl::Api* r = kzg_rccl::api(&err);
    if (!r) return fail(nullptr, KZG_E_COMM, err);
    static_assert(sizeof(ncclUniqueId) == 128, "the ABI hands the unique id over as 128 bytes");
    ncclUniqueId id;
    const ncclResult_t e = r->GetUniqueId(&id);
    if (e != ncclSuccess) return fail(nullptr, KZG_E_COMM, std::string("ncclGetUniqueId: ") + r->GetErrorString(e));
    memcpy(out_id128, &id, 128);
    return KZG_OK;
}
int kzg_comm_init(kzg_ctx* ctx, const uint8_t unique_id128[128], int rank, int world) {
    if (!ctx || !unique_id128 || world < 1 || world > KZG_MAX_GATHER || rank < 0 || rank >= world) return KZG_E_ARG;
    std::string err;
    const kzg_rccl::Api* r = kzg_rccl::api(&err);
    if (!r) return fail(ctx, KZG_E_COMM, err);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    LaneHold H(ctx);
    if (int rc = H.take_all()) return rc;
    std::lock_guard<std::mutex> lk(ctx->comm.mu);
    if (ctx->comm.comm) return fail(ctx, KZG_E_ARG, "a communicator exists already: kzg_comm_destroy first");
    // every buffer a sharded MSM needs beyond the plain MSM's, now: nothing is (re)allocated while collectives are in flight
    for (Lane& L : ctx->lane) {
        HIPCHK(ctx, L.comm_send.ensure(256));
        HIPCHK(ctx, L.comm_recv.ensure((size_t)world * 192));
        HIPCHK(ctx, L.gather.ensure(((size_t)world + 2) * sizeof(g1_xyzz_t)));
    }
    ncclUniqueId id;
    memcpy(&id, unique_id128, 128);
    ncclComm_t c = nullptr;
    const ncclResult_t e = r->CommInitRank(&c, world, id, rank);      // collective: returns when every rank has joined
    if (e != ncclSuccess || !c)
        return fail(ctx, KZG_E_COMM, std::string("ncclCommInitRank(rank ") + std::to_string(rank) + " of " + std::to_string(world) +
                                         "): " + r->GetErrorString(e));
    ctx->comm.comm = c;
    ctx->comm.rank = rank;
    ctx->comm.world = world;
    ctx->comm.broken = false;
    ctx->comm.why.clear();
    H.clean = true;
    return KZG_OK;
}
int kzg_comm_destroy(kzg_ctx* ctx) {
    if (!ctx) return KZG_E_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    LaneHold H(ctx);
    if (int rc = H.take_all()) return rc;       // no sharded MSM is in flight while the communicator goes
    comm_teardown(ctx);
    H.clean = true;
    return KZG_OK;
}
int kzg_comm_set_timeout(kzg_ctx* ctx, int timeout_ms) {
    if (!ctx || timeout_ms < 0) return KZG_E_ARG;
    std::lock_guard<std::mutex> lk(ctx->comm.mu);
    ctx->comm.timeout_ms = timeout_ms;
    return KZG_OK;
}
int kzg_comm_info(kzg_ctx* ctx, int32_t out[4]) {
    if (!ctx || !out) return KZG_E_ARG;
    std::string err;
    const kzg_rccl::Api* r = kzg_rccl::api(&err);
    std::lock_guard<std::mutex> lk(ctx->comm.mu);
    out[0] = ctx->comm.rank;
    out[1] = ctx->comm.comm || ctx->comm.broken ? ctx->comm.world : 0;
    out[2] = r ? r->version : 0;
    out[3] = ctx->comm.broken ? 1 : 0;
    return KZG_OK;
}
// One small all_gather whose content is checked: rank i contributes 192 bytes of value (i + 1) & 0xff, every rank verifies
// all `world` pieces.  What a caller runs right after kzg_comm_init -- before it builds tables and uploads scalars -- to
// learn that the communicator really moves bytes between THESE ranks (ncclCommInitRank succeeding does not prove the
// data path: transports connect at the first collective).  Honours kzg_comm_set_timeout like kzg_msm_sharded.
int kzg_comm_selftest(kzg_ctx* ctx) {
    if (!ctx) return KZG_E_ARG;
    std::string err;
    const kzg_rccl::Api* r = kzg_rccl::api(&err);
    if (!r) return fail(ctx, KZG_E_COMM, err);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    LaneHold H(ctx);
    if (int rc = H.take()) return rc;
    Lane& L = H.L();
    int world = 0, rank = 0, timeout_ms = 0;
    {
        std::lock_guard<std::mutex> lk(ctx->comm.mu);
        if (ctx->comm.broken) return fail(ctx, KZG_E_COMM, "the communicator was aborted (" + ctx->comm.why + "): kzg_comm_destroy + kzg_comm_init");
        if (!ctx->comm.comm) return fail(ctx, KZG_E_ARG, "no communicator: call kzg_comm_init");
        world = ctx->comm.world;
        rank = ctx->comm.rank;
        timeout_ms = ctx->comm.timeout_ms;
    }
    std::unique_ptr<uint8_t[]> got(new (std::nothrow) uint8_t[(size_t)world * 192]);    // no exception crosses the C boundary
    if (!got) return fail(ctx, KZG_E_NOMEM, "kzg_comm_selftest: host buffer");
    HIPCHK(ctx, hipMemsetAsync(L.comm_send.p, (rank + 1) & 0xff, 192, L.stream));
    HIPCHK(ctx, hipMemsetAsync(L.comm_recv.p, 0, (size_t)world * 192, L.stream));
    {
        std::lock_guard<std::mutex> lk(ctx->comm.mu);
        if (!ctx->comm.comm) return fail(ctx, KZG_E_COMM, "the communicator went away during the call");
        const ncclResult_t e = r->AllGather(L.comm_send.p, L.comm_recv.p, 192, ncclUint8, ctx->comm.comm, L.stream);
        if (e != ncclSuccess) {
            ctx->comm.broken = true;
            ctx->comm.why = std::string("ncclAllGather: ") + r->GetErrorString(e);
            return fail(ctx, KZG_E_COMM, ctx->comm.why);
        }
    }
    HIPCHK(ctx, hipEventRecord(L.ev_done, L.stream));
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        const hipError_t e = hipEventQuery(L.ev_done);
        if (e == hipSuccess) break;
        if (e != hipErrorNotReady) return fail(ctx, KZG_E_HIP, std::string("kzg_comm_selftest: ") + hipGetErrorString(e));
        (void)hipGetLastError();
        if (timeout_ms > 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(timeout_ms)) {
            std::lock_guard<std::mutex> lk(ctx->comm.mu);
            if (ctx->comm.comm) {
                (void)r->CommAbort(ctx->comm.comm);
                ctx->comm.comm = nullptr;
            }
            ctx->comm.broken = true;
            ctx->comm.why = "the self-test all_gather timed out after " + std::to_string(timeout_ms) + " ms";
            return fail(ctx, KZG_E_COMM, "kzg_comm_selftest: " + ctx->comm.why);
        }
        std::this_thread::sleep_for(std::chrono::microseconds(100));
    }
    HIPCHK(ctx, hipMemcpy(got.get(), L.comm_recv.p, (size_t)world * 192, hipMemcpyDeviceToHost));
    for (int i = 0; i < world; i++)
        for (int b = 0; b < 192; b++)
            if (got[(size_t)i * 192 + b] != (uint8_t)((i + 1) & 0xff))
                return fail(ctx, KZG_E_COMM, "kzg_comm_selftest: the piece of rank " + std::to_string(i) + " arrived damaged on rank " +
                                                 std::to_string(rank));
    H.clean = true;
    return KZG_OK;
}
int kzg_test_comm_stall(kzg_ctx* ctx, int ms) {
    if (!ctx || ms < 0 || ms > 2000) return KZG_E_ARG;
    ctx->comm.stall_ms.store(ms);
    return KZG_OK;
}
int kzg_msm_sharded(kzg_ctx* ctx, int slot, uint64_t n, uint64_t srs_offset, uint8_t out48[48]) {
    if (!ctx || !out48 || slot < 0 || slot >= N_SLOTS) return KZG_E_ARG;
    std::string err;
    const kzg_rccl::Api* r = kzg_rccl::api(&err);
    if (!r) return fail(ctx, KZG_E_COMM, err);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    LaneHold H(ctx);
    if (int rc = H.take()) return rc;
    Lane& L = H.L();
    int rc = need_srs(ctx);
    if (rc) return rc;
    if (n > ctx->slot_n[slot]) return fail(ctx, KZG_E_ARG, "slot holds fewer scalars than requested");
    int world = 0, timeout_ms = 0;
    {
        std::lock_guard<std::mutex> lk(ctx->comm.mu);
        if (ctx->comm.broken) return fail(ctx, KZG_E_COMM, "the communicator was aborted (" + ctx->comm.why + "): kzg_comm_destroy + kzg_comm_init");
        if (!ctx->comm.comm) return fail(ctx, KZG_E_ARG, "no communicator: call kzg_comm_init");
        world = ctx->comm.world;
        timeout_ms = ctx->comm.timeout_ms;
    }
    prof_begin(ctx, L);
    rc = clear_flags(ctx, L);
    if (rc) return rc;
    rc = msm_core(ctx, L, ctx->slot[slot].as<uint32_t>(), ctx->slot_mont[slot], n, srs_offset, L.res());
    if (rc) return rc;
    {
        Span sp(ctx, L, KZG_T_COLLECTIVE);
        launch_xyzz_pack(L.stream, L.res(), L.comm_send.as<uint32_t>(), 1);
        if (timeout_ms > 0) HIPCHK(ctx, hipEventRecord(L.ev_ext, L.stream));   // from here on the timeout's clock runs
        if (const int ms = ctx->comm.stall_ms.exchange(0)) {      // test hook: a late "peer"
            int khz = 0;
            (void)hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, ctx->device);
            k_test_stall<<<1, 64, 0, L.stream>>>((uint64_t)(khz > 0 ? khz : 100000) * (uint64_t)ms);
        }
        {
            // enqueued on the LANE's stream, stream-ordered between the partial and the sum.  The lock only serialises the
            // enqueue (RCCL: one thread at a time per communicator); the transfer itself overlaps other lanes' work.
            std::lock_guard<std::mutex> lk(ctx->comm.mu);
            if (!ctx->comm.comm) return fail(ctx, KZG_E_COMM, "the communicator went away during the call");
            const ncclResult_t e = r->AllGather(L.comm_send.p, L.comm_recv.p, 192, ncclUint8, ctx->comm.comm, L.stream);
            if (e != ncclSuccess) {
                ctx->comm.broken = true;
                ctx->comm.why = std::string("ncclAllGather: ") + r->GetErrorString(e);
                return fail(ctx, KZG_E_COMM, ctx->comm.why);
            }
        }
        g1_xyzz_t* pts = L.gather.as<g1_xyzz_t>();
        launch_xyzz_unpack(L.stream, L.comm_recv.as<uint32_t>(), pts, (uint32_t)world);
        launch_g1_sum(L.stream, pts, (uint32_t)world, L.res());
    }
    queue_encode(ctx, L, true, false);
    bool timed_out = false;
    rc = finish_bounded(ctx, L, timeout_ms, L.ev_ext, &timed_out);
    if (timed_out) {
        std::lock_guard<std::mutex> lk(ctx->comm.mu);
        if (ctx->comm.comm) {
            (void)r->CommAbort(ctx->comm.comm);      // the stuck collective leaves the stream; LaneHold then drains the lane
            ctx->comm.comm = nullptr;
        }
        ctx->comm.broken = true;
        ctx->comm.why = "a sharded MSM timed out after " + std::to_string(timeout_ms) + " ms";
    }
    if (rc) return rc;
    result_c48(ctx, L, 0, out48);
    H.clean = true;
    return KZG_OK;
}

// ---- ticketed MSM: submit returns once the work is queued on a free lane, wait returns the result, so MSM i+1 (sort,
// accumulate) overlaps the latency-bound tail (fold, bucket tree, final combination) of MSM i from ONE host thread.
// (Several host threads get the same overlap from the blocking calls: each call runs on its own lane.)
int kzg_msm_submit(kzg_ctx* ctx, int slot, uint64_t n, uint64_t srs_offset, int partial, int* out_ticket) {
    if (!ctx || !out_ticket || slot < 0 || slot >= N_SLOTS) return KZG_E_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int li = -1;
    int rc = lane_acquire(ctx, LANE_TICKET, &li);
    if (rc) return rc;
    Lane& L = ctx->lane[li];
    // checked while the lane is held: an exclusive operation (SRS reload, kzg_upload_fr) cannot slip in between
    rc = need_srs(ctx);
    if (!rc && n > ctx->slot_n[slot]) rc = fail(ctx, KZG_E_ARG, "slot holds fewer scalars than requested");
    if (rc) {
        lane_release(ctx, li);
        return rc;
    }
    prof_begin(ctx, L);
    rc = clear_flags(ctx, L);
    if (!rc) rc = msm_core(ctx, L, ctx->slot[slot].as<uint32_t>(), ctx->slot_mont[slot], n, srs_offset, L.res());
    if (!rc) {
        L.partial = partial != 0;
        if (partial) queue_pack(ctx, L);
        else queue_encode(ctx, L, true, false);
        prof_close(ctx, L);
        launch_publish(L.stream, L.tail, L.pin_dev, TB_COPY, L.flags(), reinterpret_cast<uint32_t*>(L.pin_dev + PIN_SEQ),
                       ++L.pub_seq);
        hipError_t e = hipEventRecord(L.ev_done, L.stream);
        if (e == hipSuccess) e = hipGetLastError();
        if (e != hipSuccess) rc = fail(ctx, KZG_E_HIP, std::string("kzg_msm_submit: ") + hipGetErrorString(e));
    }
    if (rc) {  // nothing may still be running on the lane's buffers when it becomes reusable
        L.sort_ws_clean = false;
        (void)hipStreamSynchronize(L.stream);
        (void)hipGetLastError();
        lane_release(ctx, li);
        return rc;
    }
    *out_ticket = li;
    return KZG_OK;
}
int kzg_msm_wait(kzg_ctx* ctx, int ticket, uint8_t* out) {
    if (!ctx || !out || ticket < 0 || ticket >= N_LANES) return KZG_E_ARG;
    Lane& L = ctx->lane[ticket];
    if (int rc0 = ticket_claim(ctx, ticket)) return rc0;   // exactly one waiter per ticket
    (void)hipSetDevice(ctx->device);
    hipError_t e = hipSuccess;  // not under the lock: other threads submit / run meanwhile
#ifndef KZG_NO_POLL
    if (ctx->profiling == 1 || !L.expect_short || !poll_pinned(ctx, L, PIN_SEQ, L.pub_seq))
#endif
        e = hipEventSynchronize(L.ev_done);
    int rc = KZG_OK;
    if (e != hipSuccess) {
        rc = fail(ctx, KZG_E_HIP, std::string("hipEventSynchronize(ticket): ") + hipGetErrorString(e));
        (void)hipStreamSynchronize(L.stream);
    } else {
        L.flags_clean = true;
        prof_end(ctx, L);
        if (L.partial) result_partial(ctx, L, out);
        else result_c48(ctx, L, 0, out);
    }
    lane_release(ctx, ticket);
    return rc;
}

// gives up an outstanding ticket (kzg_msm_submit / kzg_msm_sharded_begin) whose result will never be collected -- e.g. the
// collective between _begin and _finish raised: drains the lane and frees it
int kzg_msm_cancel(kzg_ctx* ctx, int ticket) {
    if (!ctx || ticket < 0 || ticket >= N_LANES) return KZG_E_ARG;
    Lane& L = ctx->lane[ticket];
    if (int rc0 = ticket_claim(ctx, ticket)) return rc0;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(L.stream);
    (void)hipGetLastError();
    L.spans.clear();
    L.flags_clean = false;     // its publish may not have run: the next request clears the flag words itself
    L.sort_ws_clean = false;
    lane_release(ctx, ticket);
    return KZG_OK;
}

int kzg_commit_open_resident(kzg_ctx* ctx, uint32_t i, int slot, uint64_t T, int evaluation_form,
                             const uint8_t alpha_be32[32], uint8_t out_commitment48[48], uint8_t out_eval32[32],
                             uint8_t out_proof48[48]) {
    if (!ctx || slot < 0 || slot >= N_SLOTS || !alpha_be32 || !out_commitment48 || !out_eval32 || !out_proof48)
        return KZG_E_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    LaneHold H(ctx);
    if (int rc = H.take()) return rc;
    Lane& L = H.L();
    int rc = check_worker(ctx, i, T);
    if (rc) return rc;
    if (T > ctx->slot_n[slot] || !ctx->slot_mont[slot])
        return fail(ctx, KZG_E_ARG, "slot must hold >= T Montgomery-form elements (kzg_upload_fr(.., to_mont=1))");
    prof_begin(ctx, L);
    rc = clear_flags(ctx, L);
    if (rc) return rc;
    return commit_open_dev(ctx, H, i, ctx->slot[slot].as<uint32_t>(), T, evaluation_form, alpha_be32, out_commitment48,
                           out_eval32, out_proof48);
}
int kzg_ntt_resident(kzg_ctx* ctx, int slot, uint64_t n, int inverse) {
    if (!ctx || slot < 0 || slot >= N_SLOTS || !n) return KZG_E_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    LaneHold H(ctx);
    if (int rc = H.take_all()) return rc;   // rewrites the slot in place
    Lane& L = H.L();
    if (n > ctx->slot_n[slot] || !ctx->slot_mont[slot]) return fail(ctx, KZG_E_ARG, "slot must hold >= n Montgomery elements");
    prof_begin(ctx, L);
    int rc = clear_flags(ctx, L);
    if (rc) return rc;
    rc = ntt_dev(ctx, L, ctx->slot[slot].as<uint32_t>(), n, inverse);
    if (rc) return rc;
    rc = finish(ctx, L);
    if (rc) return rc;
    H.clean = true;
    return KZG_OK;
}

// ---- pinned host staging: a pool of N_STAGE page-locked buffers, one per request in flight
int kzg_staging_acquire(kzg_ctx* ctx, uint64_t bytes, void** out_ptr, int* out_token) {
    if (!ctx || !out_ptr || !out_token) return KZG_E_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const int k = ctx->book.stage_acquire(bytes);   // waits while all are held; prefers one that is already large enough
    Stage& st = ctx->stage[k];
    if (bytes > st.cap) {
        st.p_pub.store(nullptr, std::memory_order_release);
        if (st.p) (void)hipHostFree(st.p);
        st.p = nullptr;
        st.cap = 0;
        const size_t want = (size_t)bytes + ((size_t)bytes >> 3) + 4096;
        hipError_t e = hipHostMalloc(&st.p, want, hipHostMallocDefault);
        if (e != hipSuccess) {
            st.p = nullptr;
            ctx->book.stage_set_cap(k, 0);
            (void)ctx->book.stage_release(k);
            return fail(ctx, KZG_E_NOMEM, std::string("hipHostMalloc(staging): ") + hipGetErrorString(e));
        }
        st.cap = want;
        ctx->book.stage_set_cap(k, want);
    }
    st.flushed = 0;
    st.consumed_by = 0;
    st.p_pub.store(st.p, std::memory_order_release);
    *out_ptr = st.p;
    *out_token = k;
    return KZG_OK;
}
// Starts the upload of bytes [offset, offset + bytes) of a held staging buffer to its device twin and returns at once: the
// host goes on decoding the next tile of the row while the copy engine moves this one.  Flushes must be contiguous from
// offset 0.  A compute call that is later handed the buffer's pointer finds the prefix it needs already on the device (it
// waits for the copy stream's event on ITS stream, not on the host) and skips its own upload; anything not flushed is
// uploaded the ordinary way.  Releasing the buffer forgets the flushes.
int kzg_staging_flush(kzg_ctx* ctx, int token, uint64_t offset, uint64_t bytes) {
    if (!ctx || token < 0 || token >= N_STAGE) return KZG_E_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    Stage& st = ctx->stage[token];
    if (!ctx->book.stage_held(token)) return fail(ctx, KZG_E_ARG, "staging buffer is not held");
    if (st.consumed_by) {      // a compute call has been served from the twin: the flushes start over (one-shot, see flushed_twin)
        st.consumed_by = 0;
        st.flushed = 0;
    }
    if (offset != st.flushed || offset > st.cap || bytes > st.cap - offset || (bytes & 31))
        return fail(ctx, KZG_E_ARG, "staging flush: not the next contiguous piece");
    if (!bytes) return KZG_OK;
    if (st.twin.cap < st.cap) {
        // the twin may still be read by nothing: the previous holder's compute call returned before it released the buffer
        HIPCHK(ctx, hipStreamSynchronize(ctx->h2d));
        HIPCHK(ctx, st.twin.ensure(st.cap));
    }
    HIPCHK(ctx, hipMemcpyAsync(static_cast<uint8_t*>(st.twin.p) + offset, static_cast<const uint8_t*>(st.p) + offset, bytes,
                               hipMemcpyHostToDevice, ctx->h2d));
    HIPCHK(ctx, hipEventRecord(st.ev, ctx->h2d));
    st.flushed = offset + bytes;
    return KZG_OK;
}
int kzg_staging_release(kzg_ctx* ctx, int token) {
    if (!ctx || token < 0 || token >= N_STAGE) return KZG_E_ARG;
    if (!ctx->book.stage_held(token)) return fail(ctx, KZG_E_ARG, "staging buffer is not held");
    ctx->stage[token].flushed = 0;       // still the holder's: nobody else is handed this buffer before stage_release
    ctx->stage[token].consumed_by = 0;
    if (ctx->book.stage_release(token) != kzg_book::BOOK_OK) return fail(ctx, KZG_E_ARG, "staging buffer is not held");
    return KZG_OK;
}

int kzg_set_profiling(kzg_ctx* ctx, int enable) {
    if (!ctx) return KZG_E_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    ctx->profiling = enable == 2 ? 2 : (enable != 0);
    ctx->book.set_serial(ctx->profiling == 1);     // stage profiling pins every call to lane 0
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
// dev-only prototype hooks (scripts/proto/): compiled in only by `KZG_WITH_PROTO=1 python -m zkp_subnet_amd.build`;
// the shipped library and include/kzg_mi355x.h do not carry them
#ifdef KZG_WITH_PROTO
#include "../../scripts/proto/baff_hook.inc"
#endif

// test hooks for the host-side encoder (finish_host.cpp): no GPU involved
int kzg_host_xyzz_to_c48(const uint32_t xyzz_limbs28[56], uint8_t out48[48]) {
    if (!xyzz_limbs28 || !out48) return KZG_E_ARG;
    kzg_host::xyzz_to_c48(xyzz_limbs28, out48);
    return KZG_OK;
}
int kzg_host_xyzz_pair_to_c48(const uint32_t a_limbs28[56], const uint32_t b_limbs28[56], uint8_t out_a48[48],
                              uint8_t out_b48[48]) {
    if (!a_limbs28 || !b_limbs28 || !out_a48 || !out_b48) return KZG_E_ARG;
    kzg_host::xyzz_pair_to_c48(a_limbs28, b_limbs28, out_a48, out_b48);
    return KZG_OK;
}
int kzg_host_xyzz_to_partial192(const uint32_t xyzz_limbs28[56], uint8_t out192[192]) {
    if (!xyzz_limbs28 || !out192) return KZG_E_ARG;
    kzg_host::xyzz_to_partial192(xyzz_limbs28, out192);
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
    HIPCHK(ctx, hipSetDevice(ctx->device));
    LaneHold H(ctx);
    if (int rc = H.take()) return rc;
    Lane& L = H.L();
    const size_t w = field == 0 ? 48 : 32;
    HIPCHK(ctx, L.in_be.ensure(2 * n * w));
    HIPCHK(ctx, L.out_be.ensure(n * w));
    uint8_t* da = L.in_be.as<uint8_t>();
    uint8_t* db = da + n * w;
    HIPCHK(ctx, hipMemcpyAsync(da, a_be, n * w, hipMemcpyHostToDevice, L.stream));
    HIPCHK(ctx, hipMemcpyAsync(db, b_be, n * w, hipMemcpyHostToDevice, L.stream));
    uint32_t blocks = (uint32_t)((n + 255) / 256);
    if (field == 0) k_test_fp<<<blocks, 256, 0, L.stream>>>(op, da, db, L.out_be.as<uint8_t>(), n);
    else k_test_fr<<<blocks, 256, 0, L.stream>>>(op, da, db, L.out_be.as<uint8_t>(), n);
    HIPCHK(ctx, hipMemcpyAsync(out_be, L.out_be.p, n * w, hipMemcpyDeviceToHost, L.stream));
    HIPCHK(ctx, hipStreamSynchronize(L.stream));
    HIPCHK(ctx, hipGetLastError());
    H.clean = true;
    return KZG_OK;
}
int kzg_test_g1(kzg_ctx* ctx, int op, const uint8_t* a_be96, const uint8_t* b_be96, uint8_t* out_be96, uint64_t n) {
    if (!ctx || !a_be96 || !b_be96 || !out_be96 || !n) return KZG_E_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    LaneHold H(ctx);
    if (int rc = H.take()) return rc;
    Lane& L = H.L();
    HIPCHK(ctx, L.in_be.ensure(2 * n * 96));
    HIPCHK(ctx, L.out_be.ensure(n * 96));
    uint8_t* da = L.in_be.as<uint8_t>();
    uint8_t* db = da + n * 96;
    HIPCHK(ctx, hipMemcpyAsync(da, a_be96, n * 96, hipMemcpyHostToDevice, L.stream));
    HIPCHK(ctx, hipMemcpyAsync(db, b_be96, n * 96, hipMemcpyHostToDevice, L.stream));
    if (op >= 5) k_test_g1_lp<<<(uint32_t)n, 64, 0, L.stream>>>(op, da, db, L.out_be.as<uint8_t>());
    else k_test_g1<<<(uint32_t)((n + 255) / 256), 256, 0, L.stream>>>(op, da, db, L.out_be.as<uint8_t>(), n);
    HIPCHK(ctx, hipMemcpyAsync(out_be96, L.out_be.p, n * 96, hipMemcpyDeviceToHost, L.stream));
    HIPCHK(ctx, hipStreamSynchronize(L.stream));
    HIPCHK(ctx, hipGetLastError());
    H.clean = true;
    return KZG_OK;
}

}  // extern "C"
