// multi_host.cpp -- kzg_multi_*: ONE handle for the G GPUs of a host (SURVEY 8b proposed kzg_create(device_count,
// device_ids); the reference builds one prover client per process: base/miner.py:73-84).  Written against the public C-ABI
// only; host threads only; nothing is exchanged between the devices but G x 192 bytes through the host.  Two layouts:
//   ROWS      worker index i is served by device i mod G, which holds exactly the slices of its workers -- the in-process
//             form of the reference's only distribution scheme (Pianist rows are independent, one row per miner: reference
//             neurons/validator.py:194-222).  No exchange at all.
//   SEGMENTS  one flat SRS cut into G contiguous segments, segment g on device g: ONE MSM runs as G partial MSMs, each on
//             its own device and lane, concurrently, and the G 192-byte partials are summed (BASELINE.json configs[3]; the
//             one-process form of what kzg_comm_* / kzg_msm_sharded do with one process per GPU).
// zkp_subnet_amd/multi.py is the same router one level up (with the text codec and the Client surface).
#include <string.h>

#include <algorithm>
#include <atomic>
#include <new>
#include <string>
#include <system_error>
#include <thread>
#include <vector>

#include "../../include/kzg_mi355x.h"

namespace {
enum Layout { LAYOUT_NONE = 0, LAYOUT_ROWS_WHOLE, LAYOUT_ROWS_SLICED, LAYOUT_SEGMENTS };
struct SlotRange {      // what kzg_multi_upload_fr left on the devices: scalars of global points [offset, offset + n)
    uint64_t offset = 0, n = 0;
    bool valid = false;
};
}  // namespace

struct kzg_multi {
    std::vector<kzg_ctx*> ctx;
    std::vector<int> device;
    // ROWS_WHOLE: worker i = slice i of its device (every device holds the whole SRS); ROWS_SLICED: slice i / G (every device
    // holds only the slices it serves); SEGMENTS: device g holds global points [seg_lo[g], seg_lo[g] + seg_n[g]).
    // LAYOUT_NONE after ANY failed load: a load that failed on one device has already replaced the SRS on the others, so no
    // routing rule describes the devices any more -- every call is refused until a load succeeds on all of them.
    int layout = LAYOUT_NONE;
    int machines_scale = 0;
    uint64_t total_points = 0;
    std::vector<uint64_t> seg_lo, seg_n;
    SlotRange slot[4];
};

namespace {
thread_local std::string tl_multi_err;
int mfail(int code, const std::string& msg) {
    tl_multi_err = msg;
    return code;
}
// Joins whatever was started, whatever happens: an exception that unwound past joinable threads would be std::terminate --
// the miner process gone instead of a status code.
struct Joiner {
    std::vector<std::thread> th;
    ~Joiner() {
        for (auto& t : th)
            if (t.joinable()) t.join();
    }
};
// starts up to `want` threads running work(t); returns how many really started (allocation or thread-creation failures
// simply mean fewer helpers: the caller does the rest itself)
template <class F>
unsigned spawn(Joiner& j, unsigned want, F work) {
    unsigned started = 0;
    try {
        j.th.reserve(want);
        for (; started < want; started++) j.th.emplace_back(work, started);
    } catch (...) {
    }
    return started;
}
// runs fn(g) for every device on its own host thread (the per-GPU calls block); returns the first failure
template <class F>
int each_device(kzg_multi* m, F fn) {
    const int G = (int)m->ctx.size();
    std::vector<int> rc(G, KZG_OK);
    std::vector<std::string> msg(G);
    auto body = [&](int g) {
        rc[g] = fn(g);
        if (rc[g] != KZG_OK) msg[g] = kzg_last_error(m->ctx[g]);      // the message is per THREAD: fetch it on this one
    };
    {
        Joiner j;
        const unsigned started = spawn(j, (unsigned)(G - 1), [&](unsigned t) { body((int)t + 1); });
        body(0);
        for (int g = (int)started + 1; g < G; g++) body(g);          // no thread to be had: the caller does the rest
    }
    for (int g = 0; g < G; g++)
        if (rc[g] != KZG_OK) return mfail(rc[g], "device " + std::to_string(m->device[g]) + ": " + msg[g]);
    return KZG_OK;
}
int route(kzg_multi* m, uint32_t i, kzg_ctx** ctx, uint32_t* slice) {
    if (!m || m->ctx.empty()) return mfail(KZG_E_ARG, "no devices");
    if (m->layout == LAYOUT_SEGMENTS) return mfail(KZG_E_ARG, "the handle holds SRS segments (kzg_multi_msm), not worker rows");
    if (m->layout == LAYOUT_NONE) return mfail(KZG_E_ARG, "no SRS resident: call kzg_multi_load_srs_file / kzg_multi_gen_srs");
    if (i >= (1u << m->machines_scale)) return mfail(KZG_E_ARG, "worker index outside [0, 2^machines_scale)");
    const uint32_t G = (uint32_t)m->ctx.size();
    *ctx = m->ctx[i % G];
    *slice = m->layout == LAYOUT_ROWS_SLICED ? i / G : i;
    return KZG_OK;
}
int relay(kzg_ctx* ctx, int rc) {      // a per-GPU failure becomes this handle's last error (same thread)
    if (rc != KZG_OK) tl_multi_err = kzg_last_error(ctx);
    return rc;
}
void forget(kzg_multi* m) {            // a load is about to replace the devices' SRS: nothing is routable until it has succeeded
    m->layout = LAYOUT_NONE;
    m->total_points = 0;
    for (auto& s : m->slot) s.valid = false;
}
// contiguous block partition of n points over G devices; sizes differ by at most one (zkp_subnet_amd.distributed.shard_range)
void shard_range(uint64_t n, uint64_t g, uint64_t G, uint64_t* lo, uint64_t* cnt) {
    const uint64_t base = n / G, extra = n % G;
    *lo = g * base + std::min(g, extra);
    *cnt = base + (g < extra ? 1 : 0);
}
int ceil_log2(uint64_t n) {
    int l = 0;
    while (((uint64_t)1 << l) < n) l++;
    return l;
}

int create_impl(int device_count, const int* device_ids, kzg_multi** out) {
    if (!out) return KZG_E_ARG;
    *out = nullptr;
    if (device_count < 1 || device_count > 64 || !device_ids) return mfail(KZG_E_ARG, "1..64 devices");
    kzg_multi* m = new kzg_multi();
    struct Guard {                      // an exception below (vector growth) must not leak the contexts created so far
        kzg_multi* m;
        ~Guard() {
            if (!m) return;
            for (kzg_ctx* d : m->ctx) kzg_destroy(d);
            delete m;
        }
    } guard{m};
    m->ctx.reserve(device_count);
    m->device.reserve(device_count);
    m->seg_lo.assign(device_count, 0);
    m->seg_n.assign(device_count, 0);
    for (int k = 0; k < device_count; k++) {
        kzg_ctx* c = nullptr;
        const int rc = kzg_create(device_ids[k], &c);
        if (rc != KZG_OK) {
            const std::string why = kzg_last_error(nullptr);
            return mfail(rc, "device " + std::to_string(device_ids[k]) + ": " + why);
        }
        m->ctx.push_back(c);            // reserved: cannot throw
        m->device.push_back(device_ids[k]);
    }
    guard.m = nullptr;
    *out = m;
    return KZG_OK;
}

int load_srs_file_impl(kzg_multi* m, const char* path, int compressed, int scale, int machines_scale) {
    if (!m || !path || machines_scale < 0 || machines_scale > 30) return mfail(KZG_E_ARG, "bad argument");
    const uint32_t G = (uint32_t)m->ctx.size();
    forget(m);
    // device g reads, checks and tabulates ONLY the slices of the worker indices it serves (i = g, g + G, ...): 1 / G of the
    // file, of the 34 GB of mainnet tables and of the start time each.  A device with no worker (G > slices) holds nothing.
    const int rc = each_device(m, [&](int g) {
        if ((uint64_t)g >= ((uint64_t)1 << machines_scale)) return (int)KZG_OK;
        return kzg_load_srs_file_slices(m->ctx[g], path, compressed, scale, machines_scale, (uint32_t)g, G);
    });
    if (rc != KZG_OK) return rc;        // the handle stays unloaded: some devices hold the new SRS, some the old one
    m->layout = LAYOUT_ROWS_SLICED;
    m->machines_scale = machines_scale;
    return KZG_OK;
}
int gen_srs_impl(kzg_multi* m, const uint8_t tau_be32[32], const uint8_t* s0_be32_all, int scale, int machines_scale) {
    if (!m || !tau_be32 || !s0_be32_all || machines_scale < 0 || machines_scale > 20) return mfail(KZG_E_ARG, "bad argument");
    const uint32_t M = 1u << machines_scale, G = (uint32_t)m->ctx.size();
    forget(m);
    const int rc = each_device(m, [&](int g) {
        std::vector<uint8_t> mine;     // the factors of the worker indices this device serves, in slice order: i = g, g + G, ...
        for (uint32_t i = (uint32_t)g; i < M; i += G) mine.insert(mine.end(), s0_be32_all + 32 * (size_t)i, s0_be32_all + 32 * (size_t)i + 32);
        if (mine.empty()) return (int)KZG_OK;                          // more devices than rows: nothing to hold
        return kzg_gen_srs(m->ctx[g], tau_be32, mine.data(), (uint32_t)(mine.size() / 32), scale, machines_scale);
    });
    if (rc != KZG_OK) return rc;
    m->layout = LAYOUT_ROWS_SLICED;
    m->machines_scale = machines_scale;
    return KZG_OK;
}
void install_segments(kzg_multi* m, uint64_t n_points) {
    const uint64_t G = m->ctx.size();
    for (uint64_t g = 0; g < G; g++) shard_range(n_points, g, G, &m->seg_lo[g], &m->seg_n[g]);
    m->total_points = n_points;
    m->machines_scale = 0;
    m->layout = LAYOUT_SEGMENTS;
}
int load_segments_impl(kzg_multi* m, const char* path, int compressed, uint64_t n_points) {
    if (!m || !path || !n_points) return mfail(KZG_E_ARG, "bad argument");
    const uint64_t G = m->ctx.size();
    if (n_points < G) return mfail(KZG_E_ARG, "fewer points than devices");
    forget(m);
    const int rc = each_device(m, [&](int g) {
        uint64_t lo, cnt;
        shard_range(n_points, (uint64_t)g, G, &lo, &cnt);
        return kzg_load_srs_file_range(m->ctx[g], path, compressed, lo, cnt, ceil_log2(cnt));
    });
    if (rc != KZG_OK) return rc;
    install_segments(m, n_points);
    return KZG_OK;
}
int gen_segments_impl(kzg_multi* m, const uint8_t tau_be32[32], const uint8_t* s0_be32_per_device, uint64_t n_points) {
    if (!m || !tau_be32 || !s0_be32_per_device || !n_points) return mfail(KZG_E_ARG, "bad argument");
    const uint64_t G = m->ctx.size();
    if (n_points < G) return mfail(KZG_E_ARG, "fewer points than devices");
    forget(m);
    const int rc = each_device(m, [&](int g) {
        uint64_t lo, cnt;
        shard_range(n_points, (uint64_t)g, G, &lo, &cnt);
        // one slice of 2^ceil(log2 cnt) >= cnt points from the caller's factor tau^lo (the tail beyond cnt is never addressed)
        return kzg_gen_srs(m->ctx[g], tau_be32, s0_be32_per_device + 32 * (size_t)g, 1, ceil_log2(cnt), 0);
    });
    if (rc != KZG_OK) return rc;
    install_segments(m, n_points);
    return KZG_OK;
}
// the part of global points [offset, offset + n) that lives on device g: [*lo, *lo + *cnt), possibly empty
void overlap(const kzg_multi* m, int g, uint64_t offset, uint64_t n, uint64_t* lo, uint64_t* cnt) {
    const uint64_t a = std::max(offset, m->seg_lo[g]), b = std::min(offset + n, m->seg_lo[g] + m->seg_n[g]);
    *lo = a;
    *cnt = b > a ? b - a : 0;
}
int need_segments(kzg_multi* m, uint64_t offset, uint64_t n) {
    if (!m || m->ctx.empty()) return mfail(KZG_E_ARG, "no devices");
    if (m->layout != LAYOUT_SEGMENTS)
        return mfail(KZG_E_ARG, "no SRS segments resident: call kzg_multi_load_srs_file_segments / kzg_multi_gen_srs_segments");
    if (offset > m->total_points || n > m->total_points - offset) return mfail(KZG_E_ARG, "MSM range exceeds the resident SRS");
    return KZG_OK;
}
// G partial MSMs, each on its own device (and lane), concurrently; then ONE sum of the 192-byte partials.  `one(g, lo, cnt, out192)`
// computes device g's partial over global points [lo, lo + cnt).
template <class F>
int msm_over_segments(kzg_multi* m, uint64_t offset, uint64_t n, uint8_t out48[48], F one) {
    const int G = (int)m->ctx.size();
    std::vector<uint8_t> partials((size_t)G * 192);
    std::vector<int> used(G, 0);
    const int rc = each_device(m, [&](int g) {
        uint64_t lo, cnt;
        overlap(m, g, offset, n, &lo, &cnt);
        if (!cnt) return (int)KZG_OK;
        used[g] = 1;
        return one(g, lo, cnt, partials.data() + (size_t)192 * g);
    });
    if (rc != KZG_OK) return rc;
    uint32_t count = 0;
    for (int g = 0; g < G; g++)
        if (used[g]) {
            if ((int)count != g) memmove(partials.data() + (size_t)192 * count, partials.data() + (size_t)192 * g, 192);
            count++;
        }
    // G x 192 bytes through the host: no collective is worth building for that (a single-process RCCL communicator would add
    // a group launch over G streams to save one ~10-us copy); the sum runs on device 0's auxiliary stream
    return relay(m->ctx[0], kzg_g1_sum(m->ctx[0], partials.data(), count, out48));
}
int msm_impl(kzg_multi* m, const uint8_t* scalars_be32, uint64_t n, uint64_t srs_offset, uint8_t out48[48]) {
    if (!out48 || (n && !scalars_be32)) return mfail(KZG_E_ARG, "bad argument");
    if (int rc = need_segments(m, srs_offset, n)) return rc;
    return msm_over_segments(m, srs_offset, n, out48, [&](int g, uint64_t lo, uint64_t cnt, uint8_t* out192) {
        return kzg_msm_partial(m->ctx[g], scalars_be32 + 32 * (size_t)(lo - srs_offset), cnt, lo - m->seg_lo[g], out192);
    });
}
int upload_impl(kzg_multi* m, int slot, const uint8_t* scalars_be32, uint64_t n, uint64_t srs_offset) {
    if (slot < 0 || slot >= 4 || (n && !scalars_be32)) return mfail(KZG_E_ARG, "bad argument");
    if (int rc = need_segments(m, srs_offset, n)) return rc;
    m->slot[slot].valid = false;
    const int rc = each_device(m, [&](int g) {
        uint64_t lo, cnt;
        overlap(m, g, srs_offset, n, &lo, &cnt);
        return kzg_upload_fr(m->ctx[g], slot, cnt ? scalars_be32 + 32 * (size_t)(lo - srs_offset) : scalars_be32, cnt, 0);
    });
    if (rc != KZG_OK) return rc;
    m->slot[slot].offset = srs_offset;
    m->slot[slot].n = n;
    m->slot[slot].valid = true;
    return KZG_OK;
}
int msm_resident_impl(kzg_multi* m, int slot, uint8_t out48[48]) {
    if (!m || slot < 0 || slot >= 4 || !out48) return mfail(KZG_E_ARG, "bad argument");
    if (!m->slot[slot].valid) return mfail(KZG_E_ARG, "the slot holds no scalars: kzg_multi_upload_fr first");
    const uint64_t offset = m->slot[slot].offset, n = m->slot[slot].n;
    if (int rc = need_segments(m, offset, n)) return rc;
    return msm_over_segments(m, offset, n, out48, [&](int g, uint64_t lo, uint64_t cnt, uint8_t* out192) {
        return kzg_msm_partial_resident(m->ctx[g], slot, cnt, lo - m->seg_lo[g], out192);
    });
}

// The rows of one challenge, all devices at once: row k (worker index indices[k], T x 32 bytes at rows_be32 + k * T * 32) runs
// on the device of its index, up to four rows per device in flight (a context has four lanes).  out_status[k] is that row's
// own status: one bad row never costs the others.  Returns KZG_OK when every row succeeded, else the first failing status.
int commit_open_rows_impl(kzg_multi* m, uint32_t n_rows, const uint32_t* indices, const uint8_t* rows_be32, uint64_t T,
                          int evaluation_form, const uint8_t alpha_be32[32], uint8_t* out_c48, uint8_t* out_e32, uint8_t* out_p48,
                          int* out_status) {
    if (!m || (n_rows && (!indices || !rows_be32 || !alpha_be32 || !out_c48 || !out_e32 || !out_p48 || !out_status)))
        return mfail(KZG_E_ARG, "bad argument");
    if (!n_rows) return KZG_OK;
    std::atomic<uint32_t> next{0};
    std::vector<std::string> msg(n_rows);
    auto work = [&](unsigned) {
        for (uint32_t k; (k = next.fetch_add(1)) < n_rows;) {
            out_status[k] = kzg_multi_commit_open(m, indices[k], rows_be32 + (size_t)k * T * 32, T, evaluation_form, alpha_be32,
                                                  out_c48 + 48 * (size_t)k, out_e32 + 32 * (size_t)k, out_p48 + 48 * (size_t)k);
            if (out_status[k] != KZG_OK) msg[k] = tl_multi_err;
        }
    };
    {
        Joiner j;
        const unsigned want = (unsigned)std::min<size_t>(n_rows, 4 * m->ctx.size());
        (void)spawn(j, want - 1, work);
        work(0);                       // (rows are claimed from one counter: fewer helpers only means fewer rows in flight)
    }
    for (uint32_t k = 0; k < n_rows; k++)
        if (out_status[k] != KZG_OK) return mfail(out_status[k], "row " + std::to_string(k) + " (worker " + std::to_string(indices[k]) + "): " + msg[k]);
    return KZG_OK;
}

// the entry points that allocate (vectors, strings, threads): no exception crosses the C boundary
template <class F>
int guarded(const char* who, F f) {
    try {
        return f();
    } catch (const std::bad_alloc&) {
        return mfail(KZG_E_NOMEM, std::string(who) + ": out of host memory");
    } catch (...) {
        return mfail(KZG_E_HIP, std::string(who) + ": unexpected host-side failure");
    }
}
}  // namespace

extern "C" {

void kzg_multi_destroy(kzg_multi* m) {
    if (!m) return;
    for (kzg_ctx* c : m->ctx) kzg_destroy(c);
    delete m;
}
const char* kzg_multi_last_error(kzg_multi*) { return tl_multi_err.c_str(); }
int kzg_multi_count(kzg_multi* m) { return m ? (int)m->ctx.size() : 0; }
kzg_ctx* kzg_multi_ctx(kzg_multi* m, int k) { return (m && k >= 0 && k < (int)m->ctx.size()) ? m->ctx[k] : nullptr; }
int kzg_multi_device_of(kzg_multi* m, uint32_t i) { return (m && !m->ctx.empty()) ? m->device[i % m->ctx.size()] : -1; }

int kzg_multi_commit(kzg_multi* m, uint32_t i, const uint8_t* row_be32, uint64_t T, int evaluation_form, uint8_t out48[48]) {
    kzg_ctx* c;
    uint32_t s;
    if (int rc = route(m, i, &c, &s)) return rc;
    return relay(c, kzg_commit(c, s, row_be32, T, evaluation_form, out48));
}
int kzg_multi_open(kzg_multi* m, uint32_t i, const uint8_t* row_be32, uint64_t T, int evaluation_form, const uint8_t alpha_be32[32],
                   uint8_t out_eval32[32], uint8_t out_proof48[48]) {
    kzg_ctx* c;
    uint32_t s;
    if (int rc = route(m, i, &c, &s)) return rc;
    return relay(c, kzg_open(c, s, row_be32, T, evaluation_form, alpha_be32, out_eval32, out_proof48));
}
int kzg_multi_commit_open(kzg_multi* m, uint32_t i, const uint8_t* row_be32, uint64_t T, int evaluation_form,
                          const uint8_t alpha_be32[32], uint8_t out48[48], uint8_t out_eval32[32], uint8_t out_proof48[48]) {
    kzg_ctx* c;
    uint32_t s;
    if (int rc = route(m, i, &c, &s)) return rc;
    return relay(c, kzg_commit_open(c, s, row_be32, T, evaluation_form, alpha_be32, out48, out_eval32, out_proof48));
}

int kzg_multi_create(int device_count, const int* device_ids, kzg_multi** out) {
    return guarded("kzg_multi_create", [&] { return create_impl(device_count, device_ids, out); });
}
int kzg_multi_load_srs_file(kzg_multi* m, const char* path, int compressed, int scale, int machines_scale) {
    return guarded("kzg_multi_load_srs_file", [&] { return load_srs_file_impl(m, path, compressed, scale, machines_scale); });
}
int kzg_multi_gen_srs(kzg_multi* m, const uint8_t tau_be32[32], const uint8_t* s0_be32_all, int scale, int machines_scale) {
    return guarded("kzg_multi_gen_srs", [&] { return gen_srs_impl(m, tau_be32, s0_be32_all, scale, machines_scale); });
}
int kzg_multi_load_srs_file_segments(kzg_multi* m, const char* path, int compressed, uint64_t n_points) {
    return guarded("kzg_multi_load_srs_file_segments", [&] { return load_segments_impl(m, path, compressed, n_points); });
}
int kzg_multi_gen_srs_segments(kzg_multi* m, const uint8_t tau_be32[32], const uint8_t* s0_be32_per_device, uint64_t n_points) {
    return guarded("kzg_multi_gen_srs_segments", [&] { return gen_segments_impl(m, tau_be32, s0_be32_per_device, n_points); });
}
int kzg_multi_segment(kzg_multi* m, int g, uint64_t out_first_count[2]) {
    if (!m || !out_first_count || g < 0 || g >= (int)m->ctx.size() || m->layout != LAYOUT_SEGMENTS) return mfail(KZG_E_ARG, "no such segment");
    out_first_count[0] = m->seg_lo[g];
    out_first_count[1] = m->seg_n[g];
    return KZG_OK;
}
int kzg_multi_msm(kzg_multi* m, const uint8_t* scalars_be32, uint64_t n, uint64_t srs_offset, uint8_t out48[48]) {
    return guarded("kzg_multi_msm", [&] { return msm_impl(m, scalars_be32, n, srs_offset, out48); });
}
int kzg_multi_upload_fr(kzg_multi* m, int slot, const uint8_t* scalars_be32, uint64_t n, uint64_t srs_offset) {
    return guarded("kzg_multi_upload_fr", [&] { return upload_impl(m, slot, scalars_be32, n, srs_offset); });
}
int kzg_multi_msm_resident(kzg_multi* m, int slot, uint8_t out48[48]) {
    return guarded("kzg_multi_msm_resident", [&] { return msm_resident_impl(m, slot, out48); });
}
int kzg_multi_commit_open_rows(kzg_multi* m, uint32_t n_rows, const uint32_t* indices, const uint8_t* rows_be32, uint64_t T,
                               int evaluation_form, const uint8_t alpha_be32[32], uint8_t* out_c48, uint8_t* out_e32, uint8_t* out_p48,
                               int* out_status) {
    return guarded("kzg_multi_commit_open_rows", [&] {
        return commit_open_rows_impl(m, n_rows, indices, rows_be32, T, evaluation_form, alpha_be32, out_c48, out_e32, out_p48, out_status);
    });
}

}  // extern "C"
