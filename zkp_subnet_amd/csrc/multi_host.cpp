// multi_host.cpp -- kzg_multi_*: ONE handle for the G GPUs of a host (SURVEY 8b proposed kzg_create(device_count,
// device_ids)).  A router over per-GPU contexts, written against the public C-ABI only: worker index i is served by device
// i mod G -- the in-process form of the reference's only distribution scheme (Pianist rows are independent, one row per
// miner: reference neurons/validator.py:194-222; one prover client per process: base/miner.py:73-84).  No collective;
// host threads only.  zkp_subnet_amd/multi.py is the same router one level up (with the text codec and the Client surface).
#include <string.h>

#include <algorithm>
#include <atomic>
#include <new>
#include <string>
#include <system_error>
#include <thread>
#include <vector>

#include "../../include/kzg_mi355x.h"

struct kzg_multi {
    std::vector<kzg_ctx*> ctx;
    std::vector<int> device;
    // how worker index i maps to a resident slice of its device: 0 = slice i (every device holds the whole SRS: file loads),
    // 1 = slice i / G (every device holds only the slices it serves: kzg_multi_gen_srs)
    int sliced = 0;
    int machines_scale = 0;
    bool loaded = false;
};

namespace {
thread_local std::string tl_multi_err;
int mfail(int code, const std::string& msg) {
    tl_multi_err = msg;
    return code;
}
// runs fn(g) for every device on its own host thread (the per-GPU calls block); returns the first failure
template <class F>
int each_device(kzg_multi* m, F fn) {
    const int G = (int)m->ctx.size();
    std::vector<int> rc(G, KZG_OK);
    std::vector<std::string> msg(G);
    std::vector<std::thread> th;
    auto body = [&](int g) {
        rc[g] = fn(g);
        if (rc[g] != KZG_OK) msg[g] = kzg_last_error(m->ctx[g]);      // the message is per THREAD: fetch it on this one
    };
    try {
        for (int g = 1; g < G; g++) th.emplace_back(body, g);
    } catch (const std::system_error&) {
    }
    body(0);
    for (int g = (int)th.size() + 1; g < G; g++) body(g);              // no thread to be had: the caller does the rest
    for (auto& t : th) t.join();
    for (int g = 0; g < G; g++)
        if (rc[g] != KZG_OK) return mfail(rc[g], "device " + std::to_string(m->device[g]) + ": " + msg[g]);
    return KZG_OK;
}
int route(kzg_multi* m, uint32_t i, kzg_ctx** ctx, uint32_t* slice) {
    if (!m || m->ctx.empty()) return mfail(KZG_E_ARG, "no devices");
    if (!m->loaded) return mfail(KZG_E_ARG, "no SRS resident: call kzg_multi_load_srs_file / kzg_multi_gen_srs");
    if (i >= (1u << m->machines_scale)) return mfail(KZG_E_ARG, "worker index outside [0, 2^machines_scale)");
    const uint32_t G = (uint32_t)m->ctx.size();
    *ctx = m->ctx[i % G];
    *slice = m->sliced ? i / G : i;
    return KZG_OK;
}
int relay(kzg_ctx* ctx, int rc) {      // a per-GPU failure becomes this handle's last error (same thread)
    if (rc != KZG_OK) tl_multi_err = kzg_last_error(ctx);
    return rc;
}
}  // namespace

extern "C" {

static int kzg_multi_create_impl(int device_count, const int* device_ids, kzg_multi** out) {
    if (!out) return KZG_E_ARG;
    *out = nullptr;
    if (device_count < 1 || device_count > 64 || !device_ids) return mfail(KZG_E_ARG, "1..64 devices");
    kzg_multi* m = new kzg_multi();
    for (int k = 0; k < device_count; k++) {
        kzg_ctx* c = nullptr;
        const int rc = kzg_create(device_ids[k], &c);
        if (rc != KZG_OK) {
            const std::string why = kzg_last_error(nullptr);
            for (kzg_ctx* d : m->ctx) kzg_destroy(d);
            delete m;
            return mfail(rc, "device " + std::to_string(device_ids[k]) + ": " + why);
        }
        m->ctx.push_back(c);
        m->device.push_back(device_ids[k]);
    }
    *out = m;
    return KZG_OK;
}
void kzg_multi_destroy(kzg_multi* m) {
    if (!m) return;
    for (kzg_ctx* c : m->ctx) kzg_destroy(c);
    delete m;
}
const char* kzg_multi_last_error(kzg_multi*) { return tl_multi_err.c_str(); }
int kzg_multi_count(kzg_multi* m) { return m ? (int)m->ctx.size() : 0; }
kzg_ctx* kzg_multi_ctx(kzg_multi* m, int k) { return (m && k >= 0 && k < (int)m->ctx.size()) ? m->ctx[k] : nullptr; }
int kzg_multi_device_of(kzg_multi* m, uint32_t i) { return (m && !m->ctx.empty()) ? m->device[i % m->ctx.size()] : -1; }

static int kzg_multi_load_srs_file_impl(kzg_multi* m, const char* path, int compressed, int scale, int machines_scale) {
    if (!m || !path || machines_scale < 0 || machines_scale > 30) return mfail(KZG_E_ARG, "bad argument");
    const int rc = each_device(m, [&](int g) { return kzg_load_srs_file(m->ctx[g], path, compressed, scale, machines_scale); });
    if (rc != KZG_OK) return rc;       // (a device whose load failed keeps serving its previous SRS; the handle stays as it was)
    m->sliced = 0;
    m->machines_scale = machines_scale;
    m->loaded = true;
    return KZG_OK;
}
static int kzg_multi_gen_srs_impl(kzg_multi* m, const uint8_t tau_be32[32], const uint8_t* s0_be32_all, int scale, int machines_scale) {
    if (!m || !tau_be32 || !s0_be32_all || machines_scale < 0 || machines_scale > 20) return mfail(KZG_E_ARG, "bad argument");
    const uint32_t M = 1u << machines_scale, G = (uint32_t)m->ctx.size();
    const int rc = each_device(m, [&](int g) {
        std::vector<uint8_t> mine;     // the factors of the worker indices this device serves, in slice order: i = g, g + G, ...
        for (uint32_t i = (uint32_t)g; i < M; i += G) mine.insert(mine.end(), s0_be32_all + 32 * (size_t)i, s0_be32_all + 32 * (size_t)i + 32);
        if (mine.empty()) return (int)KZG_OK;                          // more devices than rows: nothing to hold
        return kzg_gen_srs(m->ctx[g], tau_be32, mine.data(), (uint32_t)(mine.size() / 32), scale, machines_scale);
    });
    if (rc != KZG_OK) return rc;
    m->sliced = 1;
    m->machines_scale = machines_scale;
    m->loaded = true;
    return KZG_OK;
}

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
// The rows of one challenge, all devices at once: row k (worker index indices[k], T x 32 bytes at rows_be32 + k * T * 32) runs
// on the device of its index, up to four rows per device in flight (a context has four lanes).  out_status[k] is that row's
// own status: one bad row never costs the others.  Returns KZG_OK when every row succeeded, else the first failing status.
static int kzg_multi_commit_open_rows_impl(kzg_multi* m, uint32_t n_rows, const uint32_t* indices, const uint8_t* rows_be32, uint64_t T,
                               int evaluation_form, const uint8_t alpha_be32[32], uint8_t* out_c48, uint8_t* out_e32, uint8_t* out_p48,
                               int* out_status) {
    if (!m || (n_rows && (!indices || !rows_be32 || !alpha_be32 || !out_c48 || !out_e32 || !out_p48 || !out_status)))
        return mfail(KZG_E_ARG, "bad argument");
    if (!n_rows) return KZG_OK;
    std::atomic<uint32_t> next{0};
    std::vector<std::string> msg(n_rows);
    auto work = [&]() {
        for (uint32_t k; (k = next.fetch_add(1)) < n_rows;) {
            out_status[k] = kzg_multi_commit_open(m, indices[k], rows_be32 + (size_t)k * T * 32, T, evaluation_form, alpha_be32,
                                                  out_c48 + 48 * (size_t)k, out_e32 + 32 * (size_t)k, out_p48 + 48 * (size_t)k);
            if (out_status[k] != KZG_OK) msg[k] = tl_multi_err;
        }
    };
    const unsigned want = (unsigned)std::min<size_t>(n_rows, 4 * m->ctx.size());
    std::vector<std::thread> th;
    try {
        for (unsigned t = 1; t < want; t++) th.emplace_back(work);
    } catch (const std::system_error&) {
    }
    work();
    for (auto& t : th) t.join();
    for (uint32_t k = 0; k < n_rows; k++)
        if (out_status[k] != KZG_OK) return mfail(out_status[k], "row " + std::to_string(k) + " (worker " + std::to_string(indices[k]) + "): " + msg[k]);
    return KZG_OK;
}

// ---- the entry points above that allocate (vectors, strings, threads): no exception crosses the C boundary
int kzg_multi_create(int device_count, const int* device_ids, kzg_multi** out) {
    try {
        return kzg_multi_create_impl(device_count, device_ids, out);
    } catch (const std::bad_alloc&) {
        return mfail(KZG_E_NOMEM, "kzg_multi_create: out of host memory");
    } catch (...) {
        return mfail(KZG_E_HIP, "kzg_multi_create: unexpected host-side failure");
    }
}
int kzg_multi_load_srs_file(kzg_multi* m, const char* path, int compressed, int scale, int machines_scale) {
    try {
        return kzg_multi_load_srs_file_impl(m, path, compressed, scale, machines_scale);
    } catch (const std::bad_alloc&) {
        return mfail(KZG_E_NOMEM, "kzg_multi_load_srs_file: out of host memory");
    } catch (...) {
        return mfail(KZG_E_HIP, "kzg_multi_load_srs_file: unexpected host-side failure");
    }
}
int kzg_multi_gen_srs(kzg_multi* m, const uint8_t tau_be32[32], const uint8_t* s0_be32_all, int scale, int machines_scale) {
    try {
        return kzg_multi_gen_srs_impl(m, tau_be32, s0_be32_all, scale, machines_scale);
    } catch (const std::bad_alloc&) {
        return mfail(KZG_E_NOMEM, "kzg_multi_gen_srs: out of host memory");
    } catch (...) {
        return mfail(KZG_E_HIP, "kzg_multi_gen_srs: unexpected host-side failure");
    }
}
int kzg_multi_commit_open_rows(kzg_multi* m, uint32_t n_rows, const uint32_t* indices, const uint8_t* rows_be32, uint64_t T,
                               int evaluation_form, const uint8_t alpha_be32[32], uint8_t* out_c48, uint8_t* out_e32, uint8_t* out_p48,
                               int* out_status) {
    try {
        return kzg_multi_commit_open_rows_impl(m, n_rows, indices, rows_be32, T, evaluation_form, alpha_be32, out_c48, out_e32, out_p48, out_status);
    } catch (const std::bad_alloc&) {
        return mfail(KZG_E_NOMEM, "kzg_multi_commit_open_rows: out of host memory");
    } catch (...) {
        return mfail(KZG_E_HIP, "kzg_multi_commit_open_rows: unexpected host-side failure");
    }
}

}  // extern "C"
