// rccl_dl.cpp -- run-time binding of RCCL (see rccl_dl.h).  Host code only.
#include "rccl_dl.h"

#include <dlfcn.h>
#include <stdlib.h>

#include <mutex>

namespace kzg_rccl {
namespace {

std::once_flag g_once;
Api g_api;
bool g_ok = false;
std::string g_err;

template <typename F>
bool bind(void* h, const char* name, F& fn) {
    fn = reinterpret_cast<F>(dlsym(h, name));
    if (!fn) g_err = std::string("RCCL: symbol ") + name + " not found in " + g_api.path;
    return fn != nullptr;
}

void load_once() {
    // the soname first: if the process already maps an RCCL under it (torch's own copy), that copy is returned
    const char* env = getenv("KZG_RCCL_LIB");
    const char* names[] = {env && *env ? env : nullptr, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* h = nullptr;
    std::string tried;
    for (const char* n : names) {
        if (!n) continue;
        h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (h) {
            g_api.path = n;
            break;
        }
        const char* e = dlerror();
        tried += std::string(tried.empty() ? "" : "; ") + n + ": " + (e ? e : "?");
    }
    if (!h) {
        g_err = "RCCL cannot be loaded (" + tried + ")";
        return;
    }
    g_ok = bind(h, "ncclGetVersion", g_api.GetVersion) && bind(h, "ncclGetUniqueId", g_api.GetUniqueId) &&
           bind(h, "ncclCommInitRank", g_api.CommInitRank) && bind(h, "ncclCommDestroy", g_api.CommDestroy) &&
           bind(h, "ncclCommAbort", g_api.CommAbort) && bind(h, "ncclCommGetAsyncError", g_api.CommGetAsyncError) &&
           bind(h, "ncclAllGather", g_api.AllGather) && bind(h, "ncclGetErrorString", g_api.GetErrorString);
    if (g_ok && g_api.GetVersion(&g_api.version) != ncclSuccess) g_api.version = 0;
    // the handle is kept for the life of the process: communicators hold code and threads of that library
}

}  // namespace

const Api* api(std::string* err) {
    std::call_once(g_once, load_once);
    if (!g_ok) {
        if (err) *err = g_err;
        return nullptr;
    }
    return &g_api;
}

}  // namespace kzg_rccl
