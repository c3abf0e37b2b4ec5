// comm.hip -- the collective INSIDE the library ("RCCL is used directly from C++"): one process per GPU,
// rank g holds SRS segment g; the only exchange of an SRS-sharded MSM is one ncclAllGather of 192 bytes per rank, enqueued
// on the LANE's own stream between the partial and the sum.  Also the stream-chained pair for a caller-owned collective
// (kzg_msm_sharded_begin / _finish).  The reference has no device-level distribution (one row per miner:
// neurons/validator.py:194-222); this is BASELINE.json configs[3].
#include "ctx.hip.h"

using namespace kzg_impl;

namespace kzg_impl {

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
    if (ctx->comm.first_stream) {               // the stream the first collective ran on: it outlived every launch on it
        (void)hipStreamDestroy(ctx->comm.first_stream);
        ctx->comm.first_stream = nullptr;
    }
    ctx->comm.rank = ctx->comm.world = 0;
    ctx->comm.broken = false;
    ctx->comm.gen++;
    ctx->comm.why.clear();
}

}  // namespace kzg_impl

namespace {

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

// A sharded call's wait has come back with a complete record -- but was the record computed from a COMPLETE collective?
// Calls run concurrently on different lanes over one communicator: when another lane's call timed out it aborted the
// communicator under this call's all_gather too, and this lane's stream then ran on -- unpack, sum, publish -- over stale or
// partial bytes that look like valid points.  So: the communicator must still be the one the call started under
// (generation), not broken, and report no asynchronous error; anything else is KZG_E_COMM, never a result.
int comm_verdict(kzg_ctx* ctx, const kzg_rccl::Api* r, uint64_t gen, const char* who) {
    std::lock_guard<std::mutex> lk(ctx->comm.mu);
    if (ctx->comm.broken || !ctx->comm.comm || ctx->comm.gen != gen)
        return fail(ctx, KZG_E_COMM, std::string(who) + ": the communicator was aborted while this call's collective was in flight (" +
                                         (ctx->comm.why.empty() ? "replaced" : ctx->comm.why) + "); the result is discarded");
    ncclResult_t async = ncclSuccess;
    const ncclResult_t e = r->CommGetAsyncError(ctx->comm.comm, &async);
    if (e != ncclSuccess || async != ncclSuccess) {
        ctx->comm.broken = true;
        ctx->comm.gen++;
        ctx->comm.why = std::string("asynchronous RCCL error: ") + r->GetErrorString(e != ncclSuccess ? e : async);
        return fail(ctx, KZG_E_COMM, std::string(who) + ": " + ctx->comm.why);
    }
    return KZG_OK;
}
// test hook: one of the stalls kzg_test_comm_stall[_n] armed, or 0
int take_stall(kzg_ctx* ctx) {
    int left = ctx->comm.stall_left.load();
    while (left > 0 && !ctx->comm.stall_left.compare_exchange_weak(left, left - 1)) {
    }
    return left > 0 ? ctx->comm.stall_ms.load() : 0;
}
// a call's own timeout: abort the communicator (the stuck collective leaves the stream) and mark it for every lane
void comm_abort_locked_out(kzg_ctx* ctx, const kzg_rccl::Api* r, const std::string& why) {
    std::lock_guard<std::mutex> lk(ctx->comm.mu);
    if (ctx->comm.comm) {
        (void)r->CommAbort(ctx->comm.comm);
        ctx->comm.comm = nullptr;
    }
    ctx->comm.broken = true;
    ctx->comm.gen++;
    ctx->comm.why = why;
}

}  // namespace

extern "C" {

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
// Joining a communicator is a rendezvous with peers that are not ours to trust: ncclCommInitRank blocks in native code until
// every rank has arrived, and the transports between the ranks connect lazily inside the FIRST collective.  Both happen here
// on a helper thread that holds NO lane and NO lock of the context -- the join, then one checked 192-byte all_gather on
// buffers and a stream of its own -- while the caller waits with a deadline.  In time: the communicator, already connected,
// is installed (later enqueues never block the host; a device-side stall is what kzg_comm_set_timeout bounds).  Too late:
// the job is abandoned -- a communicator that exists by then is aborted, a helper still stuck in the rendezvous is left
// behind holding nothing of the context (it drops the communicator itself if it ever returns) -- the call returns KZG_E_COMM
// and the context serves on as before.
struct JoinJob {
    std::mutex mu;
    std::condition_variable cv;
    bool done = false, abandoned = false;
    ncclComm_t comm = nullptr;      // set as soon as ncclCommInitRank has returned (the waiter may abort it)
    hipStream_t first_stream = nullptr;   // the stream of the first collective: handed to the context with the communicator and
                                          // destroyed with it (never under a communicator that has launched on it)
    int rc = KZG_OK;
    std::string err;
};
static void join_steps(std::shared_ptr<JoinJob> job, const kzg_rccl::Api* r, int device, ncclUniqueId id, int rank, int world);
static void join_body(std::shared_ptr<JoinJob> job, const kzg_rccl::Api* r, int device, ncclUniqueId id, int rank, int world) {
    try {       // a detached thread: an exception that escaped it (a string that cannot be allocated) would end the process
        join_steps(job, r, device, id, rank, world);
    } catch (...) {
        std::lock_guard<std::mutex> lk(job->mu);
        if (job->comm) (void)r->CommAbort(job->comm);
        job->comm = nullptr;
        job->rc = KZG_E_NOMEM;
        job->done = true;
        job->cv.notify_all();
    }
}
static void join_steps(std::shared_ptr<JoinJob> job, const kzg_rccl::Api* r, int device, ncclUniqueId id, int rank, int world) {
    auto finish_job = [&](int rc, const std::string& why) {
        std::lock_guard<std::mutex> lk(job->mu);
        if (job->abandoned) {            // nobody is waiting any more: whatever exists goes away with this thread
            // (an aborted communicator was already dropped by the waiter: job->comm is null then)
            if (job->comm) (void)r->CommAbort(job->comm);
            job->comm = nullptr;
            if (job->first_stream) (void)hipStreamDestroy(job->first_stream);   // (handed over just before the waiter gave up)
            job->first_stream = nullptr;
        } else if (rc != KZG_OK && job->comm) {
            (void)r->CommAbort(job->comm);
            job->comm = nullptr;
        }
        job->rc = rc;
        job->err = why;
        job->done = true;
        job->cv.notify_all();
    };
    if (hipSetDevice(device) != hipSuccess) return finish_job(KZG_E_HIP, "hipSetDevice failed on the join thread");
    ncclComm_t c = nullptr;
    const ncclResult_t e = r->CommInitRank(&c, world, id, rank);      // returns when every rank has joined
    if (e != ncclSuccess || !c)
        return finish_job(KZG_E_COMM, std::string("ncclCommInitRank(rank ") + std::to_string(rank) + " of " + std::to_string(world) +
                                          "): " + r->GetErrorString(e));
    {
        std::lock_guard<std::mutex> lk(job->mu);
        job->comm = c;
        if (job->abandoned) {           // the waiter gave up while this thread sat in the rendezvous: nothing more to do here
            (void)r->CommAbort(c);
            job->comm = nullptr;
            job->done = true;
            return;
        }
    }
    // the first collective: connects the transports and proves that bytes move between THESE ranks
    hipStream_t st = nullptr;
    uint8_t* buf = nullptr;
    std::unique_ptr<uint8_t[]> got(new (std::nothrow) uint8_t[(size_t)world * 192]);
    int rc = KZG_OK;
    std::string why;
    if (!got || hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess ||
        hipMalloc((void**)&buf, ((size_t)world + 1) * 192) != hipSuccess) {
        rc = KZG_E_NOMEM;
        why = "kzg_comm_init: buffers of the first collective";
    } else {
        hipError_t he = hipMemsetAsync(buf, (rank + 1) & 0xff, 192, st);
        if (he == hipSuccess) he = hipMemsetAsync(buf + 192, 0, (size_t)world * 192, st);
        ncclResult_t ne = ncclSuccess;
        if (he == hipSuccess) ne = r->AllGather(buf, buf + 192, 192, ncclUint8, c, st);
        if (he == hipSuccess && ne == ncclSuccess) he = hipStreamSynchronize(st);
        if (he == hipSuccess && ne == ncclSuccess) he = hipMemcpy(got.get(), buf + 192, (size_t)world * 192, hipMemcpyDeviceToHost);
        if (ne != ncclSuccess) {
            rc = KZG_E_COMM;
            why = std::string("kzg_comm_init: first ncclAllGather: ") + r->GetErrorString(ne);
        } else if (he != hipSuccess) {
            rc = KZG_E_COMM;
            why = std::string("kzg_comm_init: first all_gather: ") + hipGetErrorString(he);
        } else {
            for (int i = 0; i < world && rc == KZG_OK; i++)
                for (int b = 0; b < 192; b++)
                    if (got[(size_t)i * 192 + b] != (uint8_t)((i + 1) & 0xff)) {
                        rc = KZG_E_COMM;
                        why = "kzg_comm_init: the piece of rank " + std::to_string(i) + " arrived damaged on rank " + std::to_string(rank);
                        break;
                    }
        }
    }
    (void)hipGetLastError();
    if (buf) (void)hipFree(buf);
    bool keep_stream = false;
    {
        std::lock_guard<std::mutex> lk(job->mu);
        keep_stream = rc == KZG_OK && !job->abandoned;
        if (keep_stream) job->first_stream = st;
    }
    finish_job(rc, why);                       // (aborts the communicator when the job failed or was abandoned ...)
    if (!keep_stream && st) (void)hipStreamDestroy(st);   // (... and only then does its stream go)
}
int kzg_comm_init_bounded(kzg_ctx* ctx, const uint8_t unique_id128[128], int rank, int world, int init_timeout_ms) {
    if (!ctx || !unique_id128 || world < 1 || world > KZG_MAX_GATHER || rank < 0 || rank >= world || init_timeout_ms < 0) return KZG_E_ARG;
    std::string err;
    const kzg_rccl::Api* r = kzg_rccl::api(&err);
    if (!r) return fail(ctx, KZG_E_COMM, err);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    {   // every buffer a sharded MSM needs beyond the plain MSM's, now: nothing is (re)allocated while collectives are in flight
        LaneHold H(ctx);
        if (int rc = H.take_all()) return rc;
        std::lock_guard<std::mutex> lk(ctx->comm.mu);
        if (ctx->comm.comm || ctx->comm.broken) return fail(ctx, KZG_E_ARG, "a communicator exists already: kzg_comm_destroy first");
        for (Lane& L : ctx->lane) {
            HIPCHK(ctx, L.comm_send.ensure(256));
            HIPCHK(ctx, L.comm_recv.ensure((size_t)world * 192));
            HIPCHK(ctx, L.gather.ensure(((size_t)world + 2) * sizeof(g1_xyzz_t)));
        }
        H.clean = true;
    }   // the lanes are free again: the context serves while the rendezvous runs
    ncclUniqueId id;
    memcpy(&id, unique_id128, 128);
    std::shared_ptr<JoinJob> job;
    try {
        job = std::make_shared<JoinJob>();
        std::thread(join_body, job, r, ctx->device, id, rank, world).detach();
    } catch (...) {
        return fail(ctx, KZG_E_NOMEM, "kzg_comm_init: cannot start the join thread");
    }
    int rc;
    std::string why;
    ncclComm_t c = nullptr;
    hipStream_t first_stream = nullptr;
    {
        std::unique_lock<std::mutex> lk(job->mu);
        const bool in_time = init_timeout_ms > 0
                                 ? job->cv.wait_for(lk, std::chrono::milliseconds(init_timeout_ms), [&] { return job->done; })
                                 : (job->cv.wait(lk, [&] { return job->done; }), true);
        if (!in_time) {
            job->abandoned = true;
            if (job->comm) {                    // joined, but stuck in the first collective: the abort releases the helper
                (void)r->CommAbort(job->comm);
                job->comm = nullptr;
            }
            return fail(ctx, KZG_E_COMM, "kzg_comm_init: the " + std::to_string(world) + " ranks did not all join and exchange within " +
                                             std::to_string(init_timeout_ms) + " ms (rank " + std::to_string(rank) + " gave up)");
        }
        rc = job->rc;
        why = job->err;
        c = job->comm;
        job->comm = nullptr;                    // ours now
        first_stream = job->first_stream;
        job->first_stream = nullptr;
    }
    if (rc != KZG_OK) return fail(ctx, rc, why);
    auto drop = [&]() {
        (void)r->CommAbort(c);
        if (first_stream) (void)hipStreamDestroy(first_stream);
    };
    LaneHold H(ctx);
    if (int rc2 = H.take_all()) {
        drop();
        return rc2;
    }
    std::lock_guard<std::mutex> lk(ctx->comm.mu);
    if (ctx->comm.comm) {                       // two concurrent inits: the second one loses
        drop();
        return fail(ctx, KZG_E_ARG, "a communicator exists already: kzg_comm_destroy first");
    }
    ctx->comm.first_stream = first_stream;
    ctx->comm.comm = c;
    ctx->comm.rank = rank;
    ctx->comm.world = world;
    ctx->comm.broken = false;
    ctx->comm.gen++;
    ctx->comm.why.clear();
    H.clean = true;
    return KZG_OK;
}
int kzg_comm_init(kzg_ctx* ctx, const uint8_t unique_id128[128], int rank, int world) {
    return kzg_comm_init_bounded(ctx, unique_id128, rank, world, 0);
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
    uint64_t gen = 0;
    {
        std::lock_guard<std::mutex> lk(ctx->comm.mu);
        if (ctx->comm.broken) return fail(ctx, KZG_E_COMM, "the communicator was aborted (" + ctx->comm.why + "): kzg_comm_destroy + kzg_comm_init");
        if (!ctx->comm.comm) return fail(ctx, KZG_E_ARG, "no communicator: call kzg_comm_init");
        world = ctx->comm.world;
        rank = ctx->comm.rank;
        timeout_ms = ctx->comm.timeout_ms;
        gen = ctx->comm.gen;
    }
    std::unique_ptr<uint8_t[]> got(new (std::nothrow) uint8_t[(size_t)world * 192]);    // no exception crosses the C boundary
    if (!got) return fail(ctx, KZG_E_NOMEM, "kzg_comm_selftest: host buffer");
    HIPCHK(ctx, hipMemsetAsync(L.comm_send.p, (rank + 1) & 0xff, 192, L.stream));
    HIPCHK(ctx, hipMemsetAsync(L.comm_recv.p, 0, (size_t)world * 192, L.stream));
    {
        std::lock_guard<std::mutex> lk(ctx->comm.mu);
        if (!ctx->comm.comm || ctx->comm.gen != gen) return fail(ctx, KZG_E_COMM, "the communicator went away during the call");
        const ncclResult_t e = r->AllGather(L.comm_send.p, L.comm_recv.p, 192, ncclUint8, ctx->comm.comm, L.stream);
        if (e != ncclSuccess) {
            ctx->comm.broken = true;
            ctx->comm.gen++;
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
            const std::string why = "the self-test all_gather timed out after " + std::to_string(timeout_ms) + " ms";
            comm_abort_locked_out(ctx, r, why);
            return fail(ctx, KZG_E_COMM, "kzg_comm_selftest: " + why);
        }
        std::this_thread::sleep_for(std::chrono::microseconds(100));
    }
    if (int rc = comm_verdict(ctx, r, gen, "kzg_comm_selftest")) return rc;
    HIPCHK(ctx, hipMemcpy(got.get(), L.comm_recv.p, (size_t)world * 192, hipMemcpyDeviceToHost));
    for (int i = 0; i < world; i++)
        for (int b = 0; b < 192; b++)
            if (got[(size_t)i * 192 + b] != (uint8_t)((i + 1) & 0xff))
                return fail(ctx, KZG_E_COMM, "kzg_comm_selftest: the piece of rank " + std::to_string(i) + " arrived damaged on rank " +
                                                 std::to_string(rank));
    H.clean = true;
    return KZG_OK;
}
int kzg_test_comm_stall_n(kzg_ctx* ctx, int ms, int count) {
    if (!ctx || ms < 0 || ms > 2000 || count < 0 || count > 16) return KZG_E_ARG;
    ctx->comm.stall_ms.store(ms);
    ctx->comm.stall_left.store(ms ? count : 0);
    return KZG_OK;
}
int kzg_test_comm_stall(kzg_ctx* ctx, int ms) { return kzg_test_comm_stall_n(ctx, ms, 1); }
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
    uint64_t gen = 0;
    {
        std::lock_guard<std::mutex> lk(ctx->comm.mu);
        if (ctx->comm.broken) return fail(ctx, KZG_E_COMM, "the communicator was aborted (" + ctx->comm.why + "): kzg_comm_destroy + kzg_comm_init");
        if (!ctx->comm.comm) return fail(ctx, KZG_E_ARG, "no communicator: call kzg_comm_init");
        world = ctx->comm.world;
        timeout_ms = ctx->comm.timeout_ms;
        gen = ctx->comm.gen;
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
        if (const int ms = take_stall(ctx)) {      // test hook: a late "peer"
            int khz = 0;
            (void)hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, ctx->device);
            k_test_stall<<<1, 64, 0, L.stream>>>((uint64_t)(khz > 0 ? khz : 100000) * (uint64_t)ms);
        }
        {
            // enqueued on the LANE's stream, stream-ordered between the partial and the sum.  The lock only serialises the
            // enqueue (RCCL: one thread at a time per communicator); the transfer itself overlaps other lanes' work.
            std::lock_guard<std::mutex> lk(ctx->comm.mu);
            if (!ctx->comm.comm || ctx->comm.gen != gen) return fail(ctx, KZG_E_COMM, "the communicator went away during the call");
            const ncclResult_t e = r->AllGather(L.comm_send.p, L.comm_recv.p, 192, ncclUint8, ctx->comm.comm, L.stream);
            if (e != ncclSuccess) {
                ctx->comm.broken = true;
                ctx->comm.gen++;
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
    // (the abort makes the stuck collective leave the stream; LaneHold then drains the lane)
    if (timed_out) comm_abort_locked_out(ctx, r, "a sharded MSM timed out after " + std::to_string(timeout_ms) + " ms");
    if (rc) return rc;
    // the record is complete -- was the collective?  (another lane's timeout may have aborted the communicator under it)
    if (int rc2 = comm_verdict(ctx, r, gen, "kzg_msm_sharded")) return rc2;
    result_c48(ctx, L, 0, out48);
    H.clean = true;
    return KZG_OK;
}

}  // extern "C"
