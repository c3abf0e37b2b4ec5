// lanes.hip -- the context and its four lanes: lifecycle (kzg_create / kzg_destroy: replaces Client(...) + start / stop,
// reference base/miner.py:73-84,155,181), the glue between csrc/lanebook.h (who holds which lane, ticket, staging buffer --
// HIP-free, TSan-driven) and the HIP objects hanging on those slots, the end of a request (record publish, host wait,
// result encoding), the pinned staging pool, profiling read-back and kzg_runtime_info.  See ctx.hip.h for the map.
#include "ctx.hip.h"

#include <dirent.h>
#include <unistd.h>

using namespace kzg_impl;

namespace kzg_impl {

thread_local std::string tl_err;
thread_local uint64_t tl_call_id = 0;
std::atomic<uint64_t> g_call_ids{0};
int fail(kzg_ctx*, int code, const std::string& msg) {
    tl_err = msg;
    return code;
}
const char* last_error_cstr() { return tl_err.c_str(); }
uint64_t call_id_new() { return tl_call_id = ++g_call_ids; }
uint64_t call_id_current() { return tl_call_id; }

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
void LaneHold::drain() {
    if (li >= 0) (void)hipStreamSynchronize(ctx->lane[li].stream);
    if (li >= 0) (void)hipStreamSynchronize(ctx->lane[li].vstream);
    if (li2 >= 0) (void)hipStreamSynchronize(ctx->lane[li2].stream);
    (void)hipGetLastError();
}
LaneHold::~LaneHold() {
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

hipEvent_t prof_event(Lane& L) {
    if (L.ev_used == L.ev_pool.size()) {
        hipEvent_t e;
        (void)hipEventCreate(&e);
        L.ev_pool.push_back(e);
    }
    return L.ev_pool[L.ev_used++];
}
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

// waits until the pinned word at `off` shows `seq` (a k_publish has landed); false if it does not within the budget
bool poll_pinned(const kzg_ctx* ctx, const Lane& L, uint32_t off, uint32_t seq) {
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
int finish(kzg_ctx* ctx, Lane& L, bool allow_poll) {
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

}  // namespace kzg_impl


// ---- what the process around us looks like (kzg_runtime_info).  The four lanes only overlap when the HIP runtime gives
// their streams different hardware queues (GPU_MAX_HW_QUEUES, read once when the runtime initialises: eight let the lanes
// run concurrently whatever the creation order of the process's streams -- profiles/r05_ab_hw_queues.log).  The library
// NEVER writes the environment (round 5's load-time setenv raced with getenv in multi-threaded hosts and changed queue
// allocation for every HIP user of the process): launchers export the variable (zkp_subnet_amd/_native.py, bench.py,
// INTEGRATION.md section 4), and the library MEASURES what it got and says so.
namespace {
// was a HIP / HSA runtime already live in this process when the library was loaded?  Then whatever the caller exports
// afterwards comes too late.  Observed without touching HIP or the environment: an open descriptor on /dev/kfd.
int kfd_is_open() {
    DIR* d = opendir("/proc/self/fd");
    if (!d) return 0;
    int found = 0;
    char link[64], target[64];
    while (const dirent* e = readdir(d)) {
        if (e->d_name[0] == '.') continue;
        snprintf(link, sizeof(link), "/proc/self/fd/%s", e->d_name);
        const ssize_t n = readlink(link, target, sizeof(target) - 1);
        if (n == 8 && memcmp(target, "/dev/kfd", 8) == 0) {
            found = 1;
            break;
        }
    }
    closedir(d);
    return found;
}
int g_hip_live_at_load = 0;
__attribute__((constructor)) void kzg_note_load_state() { g_hip_live_at_load = kfd_is_open(); }   // reads only

// one spinning single-wave kernel per lane, all queued before any can finish: how many were running at the same instant?
// (N_LANES: every lane has its own hardware queue; 1: the lanes execute one after the other.)  ~0.5 ms, once per context.
int probe_lane_concurrency(kzg_ctx* ctx) {
    int khz = 0;
    (void)hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, ctx->device);
    if (khz <= 0) khz = 100000;
    const uint64_t ticks = (uint64_t)khz * 300 / 1000;       // 300 us: far above the ~10 us between two launches
    DevBuf buf;
    if (buf.ensure(N_LANES * 16 + 16) != hipSuccess) return 0;
    uint64_t* d = buf.as<uint64_t>();
    launch_spin_probe(ctx->lane[0].stream, d + 2 * N_LANES, 0);          // pages the kernel's code in
    if (hipStreamSynchronize(ctx->lane[0].stream) != hipSuccess) return 0;
    for (int l = 0; l < N_LANES; l++) launch_spin_probe(ctx->lane[l].stream, d + 2 * l, ticks);
    uint64_t t[2 * N_LANES];
    bool ok = true;
    for (int l = 0; l < N_LANES; l++) ok = hipStreamSynchronize(ctx->lane[l].stream) == hipSuccess && ok;
    ok = ok && hipMemcpy(t, d, sizeof(t), hipMemcpyDeviceToHost) == hipSuccess;
    (void)hipGetLastError();
    if (!ok) return 0;
    int best = 0;
    for (int a = 0; a < N_LANES; a++) {      // the maximum is attained at some interval's start
        int live = 0;
        for (int b = 0; b < N_LANES; b++) live += t[2 * b] <= t[2 * a] && t[2 * a] < t[2 * b + 1];
        best = std::max(best, live);
    }
    return best;
}
}  // namespace

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
    const char* q = getenv("GPU_MAX_HW_QUEUES");      // read, never written
    ctx->rt_info[0] = N_LANES;
    ctx->rt_info[1] = probe_lane_concurrency(ctx);
    ctx->rt_info[2] = g_hip_live_at_load;
    ctx->rt_info[3] = q && *q ? atoi(q) : 0;
    *out = ctx;
    return KZG_OK;
}
int kzg_runtime_info(kzg_ctx* ctx, int32_t out[4]) {
    if (!ctx || !out) return KZG_E_ARG;
    for (int i = 0; i < 4; i++) out[i] = ctx->rt_info[i];
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
        for (hipEvent_t e : {L.ev_sorted, L.ev_done, L.ev_coeffs, L.ev_ext, L.ev_verify})
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

}  // extern "C"
