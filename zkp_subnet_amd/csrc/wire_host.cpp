// wire_host.cpp -- host-side wire codec of the C-ABI: Prove.poly is a list of 43-char unpadded base64 strings (reference
// base/protocol.py:35-40).  Pure host code, scalar; the Python host uses the AVX2 codec of csrc/wire_py.c.
#include <stdint.h>
#include <string.h>

#include "../../include/kzg_mi355x.h"

namespace {

const char B64[] = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789+/";
struct B64Rev {
    int8_t v[256];
    B64Rev() {
        memset(v, -1, sizeof(v));
        for (int i = 0; i < 64; i++) v[(uint8_t)B64[i]] = (int8_t)i;
    }
};
const int8_t* b64_rev_table() {
    static const B64Rev t;      // initialised once, thread-safely (the axon decodes on several threads)
    return t.v;
}

}  // namespace

extern "C" {

int kzg_b64_decode_fr(const char* packed43, uint64_t n, uint8_t* out_be32) {
    if ((n && !packed43) || (n && !out_be32)) return KZG_E_ARG;
    const int8_t* b64_rev = b64_rev_table();
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

}  // extern "C"
