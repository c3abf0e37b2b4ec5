/* _wire: CPython extension for the text side of the `Prove` synapse (reference base/protocol.py:24-60: `poly` is a
 * List[str] of T base64 field elements, 43 characters each; neurons/miner.py:38-61 hands that list to the prover).
 *
 * The reference ships the list to its prover as JSON over localhost twice per request.  Here the list is decoded in
 * place (no join, no intermediate 43*T-byte copy): a small persistent thread pool walks disjoint ranges of the list, validates each str
 * and convert base64 -> 32 bytes big-endian while the calling thread keeps the GIL.  At T = 2^20 this replaces
 * ~130 ms of "".join + encode + single-thread decode with a few ms.  Canonicity (< r) is still checked on the device.
 *
 *   decode_fr_list(seq[, threads]) -> bytes   (32 * len(seq))
 *   decode_fr_list_into(seq, address, capacity[, threads]) -> n   (into the library's pinned staging buffer)
 *   decode_fr_list_into_tagged(seq, address, capacity[, threads]) -> (n, tag)   the same + a 128-bit content tag of the
 *                                                                  decoded bytes (kzg_commit_cached / kzg_open_cached)
 *   encode_fr_list(bytes[, threads]) -> list[str]
 *   random_fr_rows(rows, T[, threads]) -> list[list[str]]   uniform field elements (getrandom + rejection), as text
 *
 * Host-side codec only: no field or curve arithmetic happens here. */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <sys/random.h>
#include <unistd.h>

static const char B64[] = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789+/";
static int8_t REV[256];

static int decode43(const uint8_t* s, uint8_t* o) {
    int bad = 0;
    for (int g = 0; g < 10; g++) {
        int a = REV[s[4 * g]], b = REV[s[4 * g + 1]], c = REV[s[4 * g + 2]], d = REV[s[4 * g + 3]];
        bad |= (a | b | c | d);
        uint32_t v = ((uint32_t)a << 18) | ((uint32_t)b << 12) | ((uint32_t)c << 6) | (uint32_t)d;
        o[3 * g] = (uint8_t)(v >> 16);
        o[3 * g + 1] = (uint8_t)(v >> 8);
        o[3 * g + 2] = (uint8_t)v;
    }
    int a = REV[s[40]], b = REV[s[41]], c = REV[s[42]];
    bad |= (a | b | c);
    uint32_t v = ((uint32_t)a << 12) | ((uint32_t)b << 6) | (uint32_t)c; /* 18 bits; the low 2 must be zero */
    o[30] = (uint8_t)(v >> 10);
    o[31] = (uint8_t)(v >> 2);
    return (bad < 0) || (v & 3u);
}
/* ---- AVX2 form of decode43 (SIMD base64 decode of the T scalars).  The 43 characters are covered by two
 * overlapping 32-character blocks (characters 0..31 -> bytes 0..23, characters 8..39 -> bytes 6..29; a str's buffer
 * holds 43 characters + NUL, so both loads stay inside it) plus the 3-character tail in scalar code.  Per block: nibble
 * lookups classify every character (any byte outside the alphabet sets a bit in lo & hi) and give the offset that maps it
 * to its 6-bit value; two multiply-adds pack 4 x 6 bits into 3 bytes (the published vector base64 scheme: Mula & Lemire,
 * "Faster Base64 Encoding and Decoding using AVX2 Instructions").  Checked against the scalar decoder on every byte
 * value in every position (tests/test_abi.py). */
#if defined(__x86_64__)
#include <immintrin.h>
__attribute__((target("avx2"))) static inline int dec32_avx2(const uint8_t* s, __m256i* out24) {
    const __m256i lut_lo = _mm256_setr_epi8(0x15, 0x11, 0x11, 0x11, 0x11, 0x11, 0x11, 0x11, 0x11, 0x11, 0x13, 0x1A, 0x1B, 0x1B,
                                            0x1B, 0x1A, 0x15, 0x11, 0x11, 0x11, 0x11, 0x11, 0x11, 0x11, 0x11, 0x11, 0x13, 0x1A,
                                            0x1B, 0x1B, 0x1B, 0x1A);
    const __m256i lut_hi = _mm256_setr_epi8(0x10, 0x10, 0x01, 0x02, 0x04, 0x08, 0x04, 0x08, 0x10, 0x10, 0x10, 0x10, 0x10, 0x10,
                                            0x10, 0x10, 0x10, 0x10, 0x01, 0x02, 0x04, 0x08, 0x04, 0x08, 0x10, 0x10, 0x10, 0x10,
                                            0x10, 0x10, 0x10, 0x10);
    const __m256i lut_roll = _mm256_setr_epi8(0, 16, 19, 4, -65, -65, -71, -71, 0, 0, 0, 0, 0, 0, 0, 0, 0, 16, 19, 4, -65, -65,
                                              -71, -71, 0, 0, 0, 0, 0, 0, 0, 0);
    const __m256i mask_2f = _mm256_set1_epi8(0x2f);
    const __m256i in = _mm256_loadu_si256((const __m256i*)s);
    const __m256i hi_n = _mm256_and_si256(_mm256_srli_epi32(in, 4), mask_2f);
    const __m256i lo_n = _mm256_and_si256(in, mask_2f);
    const __m256i hi = _mm256_shuffle_epi8(lut_hi, hi_n);
    const __m256i lo = _mm256_shuffle_epi8(lut_lo, lo_n);
    const __m256i eq_2f = _mm256_cmpeq_epi8(in, mask_2f);
    const __m256i roll = _mm256_shuffle_epi8(lut_roll, _mm256_add_epi8(eq_2f, hi_n));
    const int bad = !_mm256_testz_si256(lo, hi);
    const __m256i vals = _mm256_add_epi8(in, roll);
    const __m256i m1 = _mm256_maddubs_epi16(vals, _mm256_set1_epi32(0x01400140));
    __m256i o = _mm256_madd_epi16(m1, _mm256_set1_epi32(0x00011000));
    o = _mm256_shuffle_epi8(o, _mm256_setr_epi8(2, 1, 0, 6, 5, 4, 10, 9, 8, 14, 13, 12, -1, -1, -1, -1, 2, 1, 0, 6, 5, 4, 10, 9,
                                                8, 14, 13, 12, -1, -1, -1, -1));
    *out24 = _mm256_permutevar8x32_epi32(o, _mm256_setr_epi32(0, 1, 2, 4, 5, 6, 7, 7));
    return bad;
}
/* w (optional): the 32 decoded bytes as four little-endian words, taken from the vector registers -- reading them back
 * from `o` right after these overlapping stores would stall on store-to-load forwarding (measured: +17 ns per element) */
__attribute__((target("avx2"))) static int decode43_avx2(const uint8_t* s, uint8_t* o, uint64_t* w) {
    __m256i a, b;
    int bad = dec32_avx2(s, &a);
    bad |= dec32_avx2(s + 8, &b);
    _mm_storeu_si128((__m128i*)o, _mm256_castsi256_si128(a));                       /* bytes 0..15 */
    _mm_storel_epi64((__m128i*)(o + 16), _mm256_extracti128_si256(a, 1));          /* bytes 16..23 */
    _mm_storeu_si128((__m128i*)(o + 6), _mm256_castsi256_si128(b));                /* bytes 6..21 (same values) */
    _mm_storel_epi64((__m128i*)(o + 22), _mm256_extracti128_si256(b, 1));          /* bytes 22..29 */
    const int x = REV[s[40]], y = REV[s[41]], z = REV[s[42]];
    bad |= (x | y | z) < 0;
    const uint32_t v = ((uint32_t)x << 12) | ((uint32_t)y << 6) | (uint32_t)z;     /* 18 bits; the low 2 must be zero */
    o[30] = (uint8_t)(v >> 10);
    o[31] = (uint8_t)(v >> 2);
    if (w) {
        w[0] = (uint64_t)_mm256_extract_epi64(a, 0);
        w[1] = (uint64_t)_mm256_extract_epi64(a, 1);
        w[2] = (uint64_t)_mm256_extract_epi64(a, 2);
        w[3] = ((uint64_t)_mm256_extract_epi64(b, 2) >> 16) | ((uint64_t)(uint8_t)(v >> 10) << 48) | ((uint64_t)(uint8_t)(v >> 2) << 56);
    }
    return bad || (v & 3u);
}
static int have_avx2 = 0;
#define DECODE43(s, o) (have_avx2 ? decode43_avx2((s), (o), NULL) : decode43((s), (o)))
#else
static const int have_avx2 = 0;
#define DECODE43(s, o) decode43((s), (o))
#endif
static void encode43(const uint8_t* i, uint8_t* o) {
    for (int g = 0; g < 10; g++) {
        uint32_t v = ((uint32_t)i[3 * g] << 16) | ((uint32_t)i[3 * g + 1] << 8) | i[3 * g + 2];
        o[4 * g] = B64[v >> 18];
        o[4 * g + 1] = B64[(v >> 12) & 63];
        o[4 * g + 2] = B64[(v >> 6) & 63];
        o[4 * g + 3] = B64[v & 63];
    }
    uint32_t v = (((uint32_t)i[30] << 8) | i[31]) << 2;
    o[40] = B64[v >> 12];
    o[41] = B64[(v >> 6) & 63];
    o[42] = B64[v & 63];
}

typedef struct {
    PyObject** items; /* borrowed: the main thread holds the GIL (and a reference to the sequence) throughout */
    uint8_t* dst;
    Py_ssize_t lo, hi;
    Py_ssize_t bad; /* first bad index or -1 */
    int bad_kind;   /* 1 = not a 43-char ASCII str, 2 = invalid base64 */
    int want_tag;
    uint64_t tag[2]; /* this range's share of the content tag */
    int kind;        /* 0 = decode items[lo, hi) into dst; 1 = fill dst[43*lo, 43*hi) with random field elements as text */
} __attribute__((aligned(64))) dec_job;

/* ---- content tag: identifies the decoded row for the prover's coefficient cache (the unchanged reference miner sends the
 * same row in worker_commit and worker_open, neurons/miner.py:56-61).  tag = sum over elements of a keyed 128-bit hash
 * of (index, 32 bytes), mod 2^128: independent of how the list is split over threads, bound to positions, and keyed with
 * 512 random bits drawn once per process.  It is a fast keyed multiply-fold with NO cryptographic analysis (a zero
 * multiplicand drops a word from a term, single lanes admit position swaps): the prover therefore treats it as a hint and
 * verifies every cache hit against the cached row's bytes on the GPU (csrc/serve.hip, commit_open_host_cached).  Four 64x64->128 multiplies
 * per element, folded into the decode pass (the bytes are still in the vector registers). */
static uint64_t TAG_KEY[8];
static inline uint64_t mum64(uint64_t a, uint64_t b) {
    const unsigned __int128 m = (unsigned __int128)a * b;
    return (uint64_t)m ^ (uint64_t)(m >> 64);
}
/* two independent 64-bit lanes, each a keyed multiply-fold of all four words and the index (four independent 64x64->128
 * multiplies, no dependent chain: ~1 ns per element) */
#define TAG_C1 0x9E3779B97F4A7C15ull
#define TAG_C2 0xCA5A826395121157ull
static inline void tag_add(uint64_t acc[2], uint64_t idx, const uint64_t w[4]) {   /* w: the 32 bytes as LE words */
    const uint64_t i1 = (idx + 1) * TAG_C1, i2 = idx * TAG_C2 + 0xD6E8FEB86659FD93ull;   /* position binding */
    acc[0] += mum64(w[0] ^ TAG_KEY[0] ^ i1, w[1] ^ TAG_KEY[1]) + mum64(w[2] ^ TAG_KEY[2], w[3] ^ TAG_KEY[3] ^ i2);
    acc[1] += mum64(w[0] ^ TAG_KEY[4] ^ i2, w[2] ^ TAG_KEY[5]) + mum64(w[1] ^ TAG_KEY[6], w[3] ^ TAG_KEY[7] ^ i1);
}
static void rand_worker(dec_job* j);
static void* dec_worker(void* p) {
    dec_job* j = (dec_job*)p;
    j->bad = -1;
    j->bad_kind = 0;
    if (j->kind == 1) {
        rand_worker(j);
        return NULL;
    }
    /* accumulated in locals: the job records of different threads share cache lines (measured on the 16-core box: +25 ns
     * per element of false sharing when every element's tag went straight into j->tag) */
    uint64_t tag[2] = {0, 0};
    for (Py_ssize_t k = j->lo; k < j->hi; k++) {
        if (k + 12 < j->hi) { /* the str objects are scattered over the heap: without this, one cache miss each */
            __builtin_prefetch(j->items[k + 12]);
            __builtin_prefetch((const char*)j->items[k + 12] + 64);
        }
        PyObject* it = j->items[k];
        int kind = 0;
        if (!PyUnicode_Check(it) || !PyUnicode_IS_READY(it) || !PyUnicode_IS_COMPACT_ASCII(it) ||
            PyUnicode_GET_LENGTH(it) != 43)
            kind = 1;
        else if (!j->want_tag) {
            if (DECODE43((const uint8_t*)PyUnicode_1BYTE_DATA(it), j->dst + 32 * k)) kind = 2;
        } else {
            uint64_t w[4];
#if defined(__x86_64__)
            if (have_avx2) {
                if (decode43_avx2((const uint8_t*)PyUnicode_1BYTE_DATA(it), j->dst + 32 * k, w)) kind = 2;
            } else
#endif
            {
                if (decode43((const uint8_t*)PyUnicode_1BYTE_DATA(it), j->dst + 32 * k)) kind = 2;
                memcpy(w, j->dst + 32 * k, 32);
            }
            if (!kind) tag_add(tag, (uint64_t)k, w);
        }
        if (kind && j->bad < 0) {
            j->bad = k;
            j->bad_kind = kind;
        }
    }
    j->tag[0] = tag[0];
    j->tag[1] = tag[1];
    return NULL;
}

/* ---- uniform random field elements as wire text (the validator's challenge: Client.random_poly() is 2^machines_scale
 * rows of 2^(scale - machines_scale) elements, reference neurons/validator.py:67-75 -- 2^24 strings at mainnet scale).
 * 32 bytes from getrandom(2), top bit cleared (a 255-bit candidate), accepted when below r (probability 0.906): exactly
 * uniform on [0, r).  Big-endian comparison against r's bytes. */
static const uint8_t R_BE[32] = {0x73, 0xed, 0xa7, 0x53, 0x29, 0x9d, 0x7d, 0x48, 0x33, 0x39, 0xd8, 0x08, 0x09, 0xa1, 0xd8, 0x05,
                                 0x53, 0xbd, 0xa4, 0x02, 0xff, 0xfe, 0x5b, 0xfe, 0xff, 0xff, 0xff, 0xff, 0x00, 0x00, 0x00, 0x01};
static int fill_random(uint8_t* p, size_t n) {
    while (n) {
        ssize_t got = getrandom(p, n > (1u << 24) ? (1u << 24) : n, 0);
        if (got <= 0) return -1;
        p += got;
        n -= (size_t)got;
    }
    return 0;
}
static void rand_worker(dec_job* j) {
    uint8_t pool[32 * 512];
    size_t have = 0, pos = 0;
    for (Py_ssize_t k = j->lo; k < j->hi;) {
        if (pos == have) {
            if (fill_random(pool, sizeof(pool))) {
                j->bad = k;
                j->bad_kind = 3;
                return;
            }
            have = 512;
            pos = 0;
        }
        uint8_t* c = pool + 32 * pos++;
        c[0] &= 0x7f;
        if (memcmp(c, R_BE, 32) >= 0) continue;   /* >= r: draw again */
        encode43(c, j->dst + 43 * k);
        k++;
    }
}

#define POOL_MAX 15
static int pick_threads(long want, Py_ssize_t n) {
    /* measured on the GPU box's EPYC: 2^12 4 threads, 2^14 and up 8; long rows (one cache miss per str object: the walk is
     * memory-latency bound) use the box's whole 16-core share */
    if (want <= 0) want = n < 8192 ? 4 : n < (1 << 18) ? 8 : 16;
    if (want > POOL_MAX + 1) want = POOL_MAX + 1;
    if (n < 1024) return 1;
    return (int)want;
}

/* ---- a small persistent worker pool: creating 7 threads per call costs ~0.12 ms, as much as decoding a 2^16 row.
 * Workers sleep on a condition variable between calls; the caller runs slice 0 itself and waits for the others.
 * Calls are serialised by the GIL (the caller keeps it for the whole decode), so one job slot is enough.  After a
 * fork() the child has no workers: the pool is rebuilt when the pid changes. */
static struct {
    pthread_mutex_t mu;
    pthread_cond_t wake, done;
    pthread_t th[POOL_MAX];
    int nthreads;            /* workers alive */
    long pid;
    unsigned long generation; /* bumped per job */
    int njobs, next, finished;
    dec_job* jobs;
} pool = {PTHREAD_MUTEX_INITIALIZER, PTHREAD_COND_INITIALIZER, PTHREAD_COND_INITIALIZER, {0}, 0, 0, 0, 0, 0, 0, NULL};

static void* pool_worker(void* arg) {
    (void)arg;
    unsigned long seen = 0;
    pthread_mutex_lock(&pool.mu);
    for (;;) {
        while (pool.generation == seen || pool.next >= pool.njobs) {
            seen = pool.generation;
            pthread_cond_wait(&pool.wake, &pool.mu);
        }
        while (pool.next < pool.njobs) {
            dec_job* j = &pool.jobs[pool.next++];
            pthread_mutex_unlock(&pool.mu);
            dec_worker(j);
            pthread_mutex_lock(&pool.mu);
            if (++pool.finished == pool.njobs) pthread_cond_signal(&pool.done);
        }
        seen = pool.generation;
    }
    return NULL;
}
/* runs jobs[0..n) (n >= 1): job 0 on the calling thread, the rest on pool workers (or inline if none could start, or if
 * the pool's job slot is held by an asynchronous batch -- creating Python objects between pool_submit and pool_join can
 * run arbitrary finalizers, which may decode) */
static int pool_async_out;
static void pool_run(dec_job* jobs, int n) {
    if (n > 1 && !pool_async_out) {
        pthread_mutex_lock(&pool.mu);
        if (pool.pid != (long)getpid()) { /* first use, or a forked child: no workers here */
            pool.nthreads = 0;
            pool.pid = (long)getpid();
        }
        while (pool.nthreads < n - 1 && pool.nthreads < POOL_MAX) {
            pthread_attr_t at;
            pthread_attr_init(&at);
            pthread_attr_setdetachstate(&at, PTHREAD_CREATE_DETACHED);
            int rc = pthread_create(&pool.th[pool.nthreads], &at, pool_worker, NULL);
            pthread_attr_destroy(&at);
            if (rc) break;
            pool.nthreads++;
        }
        if (pool.nthreads > 0) {
            pool.jobs = jobs;
            pool.njobs = n;
            pool.next = 1; /* job 0 is the caller's */
            pool.finished = 1;
            pool.generation++;
            pthread_cond_broadcast(&pool.wake);
            pthread_mutex_unlock(&pool.mu);
            dec_worker(&jobs[0]);
            pthread_mutex_lock(&pool.mu);
            /* help with whatever is still unclaimed, then wait for the claimed ones */
            while (pool.next < pool.njobs) {
                dec_job* j = &pool.jobs[pool.next++];
                pthread_mutex_unlock(&pool.mu);
                dec_worker(j);
                pthread_mutex_lock(&pool.mu);
                pool.finished++;
            }
            while (pool.finished < pool.njobs) pthread_cond_wait(&pool.done, &pool.mu);
            pool.njobs = 0;
            pthread_mutex_unlock(&pool.mu);
            return;
        }
        pthread_mutex_unlock(&pool.mu);
    }
    for (int t = 0; t < n; t++) dec_worker(&jobs[t]);
}

/* asynchronous form: pool_submit hands ALL jobs to the workers and returns (1), or returns 0 when no worker could be
 * started (the caller then runs the jobs itself); pool_join helps with unclaimed jobs and waits for the rest.  One
 * outstanding batch at a time (the GIL serialises callers). */
/* pool_async_out: an asynchronous batch is outstanding (random_fr_rows): the pool's one job slot is taken */
static int pool_submit(dec_job* jobs, int n) {
    if (pool_async_out) return 0;
    pthread_mutex_lock(&pool.mu);
    if (pool.pid != (long)getpid()) {
        pool.nthreads = 0;
        pool.pid = (long)getpid();
    }
    while (pool.nthreads < n && pool.nthreads < POOL_MAX) {
        pthread_attr_t at;
        pthread_attr_init(&at);
        pthread_attr_setdetachstate(&at, PTHREAD_CREATE_DETACHED);
        int rc = pthread_create(&pool.th[pool.nthreads], &at, pool_worker, NULL);
        pthread_attr_destroy(&at);
        if (rc) break;
        pool.nthreads++;
    }
    if (pool.nthreads == 0) {
        pthread_mutex_unlock(&pool.mu);
        return 0;
    }
    pool.jobs = jobs;
    pool.njobs = n;
    pool.next = 0;
    pool.finished = 0;
    pool.generation++;
    pool_async_out = 1;
    pthread_cond_broadcast(&pool.wake);
    pthread_mutex_unlock(&pool.mu);
    return 1;
}
static void pool_join(void) {
    pthread_mutex_lock(&pool.mu);
    while (pool.next < pool.njobs) {
        dec_job* j = &pool.jobs[pool.next++];
        pthread_mutex_unlock(&pool.mu);
        dec_worker(j);
        pthread_mutex_lock(&pool.mu);
        pool.finished++;
    }
    while (pool.finished < pool.njobs) pthread_cond_wait(&pool.done, &pool.mu);
    pool.njobs = 0;
    pool_async_out = 0;
    pthread_mutex_unlock(&pool.mu);
}

/* Decodes the sequence into dst (n * 32 bytes).  Returns n, or -1 with a Python error set.  `cap` = room at dst in
 * bytes (0: dst is NULL and a bytes object is created: *out_bytes).  The GIL stays with the calling thread for the
 * whole call (~2 ns per element on 8 threads), which is what makes the lock-free reads above legal. */
/* Range form (first / count, count < 0 = to the end): elements [first, first + count) are decoded to dst + 32 * first --
 * dst is always the address of element 0 -- so that a long row can be decoded tile by tile, each tile's upload started
 * (kzg_staging_flush) while the next is decoded.  The tag of a range is that range's share of the content tag; the
 * length term is added by the call that decodes the LAST element, so the shares of consecutive ranges sum (mod 2^64 per
 * lane) to the tag of the whole row.  Returns the number of elements decoded. */
static Py_ssize_t decode_core_range(PyObject* seq_in, long threads, uint8_t* dst, size_t cap, PyObject** out_bytes,
                                    uint64_t* out_tag, Py_ssize_t first, Py_ssize_t count);
static Py_ssize_t decode_core(PyObject* seq_in, long threads, uint8_t* dst, size_t cap, PyObject** out_bytes,
                              uint64_t* out_tag) {
    return decode_core_range(seq_in, threads, dst, cap, out_bytes, out_tag, 0, -1);
}
static Py_ssize_t decode_core_range(PyObject* seq_in, long threads, uint8_t* dst, size_t cap, PyObject** out_bytes,
                                    uint64_t* out_tag, Py_ssize_t first, Py_ssize_t count) {
    PyObject* seq = PySequence_Fast(seq_in, "polynomial must be a sequence of base64 strings");
    if (!seq) return -1;
    const Py_ssize_t n = PySequence_Fast_GET_SIZE(seq);
    if (count < 0) count = n - first;
    if (first < 0 || first > n || count < 0 || first + count > n || (out_bytes && (first || count != n))) {
        Py_DECREF(seq);
        PyErr_SetString(PyExc_ValueError, "bad element range");
        return -1;
    }
    PyObject* out = NULL;
    if (out_bytes) {
        out = PyBytes_FromStringAndSize(NULL, 32 * n);
        if (!out) {
            Py_DECREF(seq);
            return -1;
        }
        dst = (uint8_t*)PyBytes_AS_STRING(out);
    } else if ((size_t)(first + count) * 32 > cap) {
        Py_DECREF(seq);
        PyErr_SetString(PyExc_ValueError, "destination buffer too small for the polynomial");
        return -1;
    }
    const int T = pick_threads(threads, count);
    dec_job jobs[POOL_MAX + 1];
    for (int t = 0; t < T; t++) {
        jobs[t].items = PySequence_Fast_ITEMS(seq);
        jobs[t].dst = dst;
        jobs[t].lo = first + count * t / T;
        jobs[t].hi = first + count * (t + 1) / T;
        jobs[t].want_tag = out_tag != NULL;
        jobs[t].kind = 0;
    }
    pool_run(jobs, T);
    if (out_tag) {
        out_tag[0] = out_tag[1] = 0;
        if (first + count == n) {   /* the length is part of the content: added by the call that reaches the end */
            out_tag[0] = mum64((uint64_t)n ^ TAG_KEY[1], TAG_KEY[2] | 1);
            out_tag[1] = mum64((uint64_t)n ^ TAG_KEY[7], TAG_KEY[4] | 1);
        }
        for (int t = 0; t < T; t++) {
            out_tag[0] += jobs[t].tag[0];
            out_tag[1] += jobs[t].tag[1];
        }
    }
    Py_ssize_t bad = -1;
    int kind = 0;
    for (int t = T - 1; t >= 0; t--)
        if (jobs[t].bad >= 0) {
            bad = jobs[t].bad;
            kind = jobs[t].bad_kind;
        }
    Py_DECREF(seq);
    if (bad >= 0) {
        Py_XDECREF(out);
        PyErr_Format(PyExc_ValueError,
                     kind == 1 ? "polynomial entry %zd is not a 43-character base64 field element"
                               : "polynomial entry %zd is not valid unpadded base64 of 32 bytes", bad);
        return -1;
    }
    if (out_bytes) *out_bytes = out;
    return count;
}

static PyObject* decode_fr_list(PyObject* self, PyObject* args) {
    PyObject* seq_in;
    long threads = 0;
    if (!PyArg_ParseTuple(args, "O|l", &seq_in, &threads)) return NULL;
    PyObject* out = NULL;
    if (decode_core(seq_in, threads, NULL, 0, &out, NULL) < 0) return NULL;
    return out;
}

/* decode_fr_list_into(seq, address, capacity[, threads]) -> n: writes n*32 bytes at `address` (a caller-owned buffer,
 * e.g. the library's pinned staging area from kzg_staging_buffer): no bytes object, no page faults, no second copy */
static PyObject* decode_fr_list_into(PyObject* self, PyObject* args) {
    PyObject* seq_in;
    unsigned long long addr, cap;
    long threads = 0;
    Py_ssize_t first = 0, count = -1;   /* a range of the sequence (tile-by-tile decode of a long row) */
    if (!PyArg_ParseTuple(args, "OKK|lnn", &seq_in, &addr, &cap, &threads, &first, &count)) return NULL;
    if (!addr) {
        PyErr_SetString(PyExc_ValueError, "null destination");
        return NULL;
    }
    Py_ssize_t n = decode_core_range(seq_in, threads, (uint8_t*)(uintptr_t)addr, (size_t)cap, NULL, NULL, first, count);
    if (n < 0) return NULL;
    return PyLong_FromSsize_t(n);
}
/* decode_fr_list_into_tagged(seq, address, capacity[, threads]) -> (n, tag: 16 bytes) */
static PyObject* decode_fr_list_into_tagged(PyObject* self, PyObject* args) {
    PyObject* seq_in;
    unsigned long long addr, cap;
    long threads = 0;
    Py_ssize_t first = 0, count = -1;
    const char* base = NULL;            /* the tag shares of the ranges decoded so far (16 bytes), summed into the result */
    Py_ssize_t base_len = 0;
    if (!PyArg_ParseTuple(args, "OKK|lnnz#", &seq_in, &addr, &cap, &threads, &first, &count, &base, &base_len)) return NULL;
    if (!addr || (base && base_len != 16)) {
        PyErr_SetString(PyExc_ValueError, base ? "the running tag must be 16 bytes" : "null destination");
        return NULL;
    }
    uint64_t tag[2];
    Py_ssize_t n = decode_core_range(seq_in, threads, (uint8_t*)(uintptr_t)addr, (size_t)cap, NULL, tag, first, count);
    if (n < 0) return NULL;
    if (base) {
        uint64_t b[2];
        memcpy(b, base, 16);
        tag[0] += b[0];
        tag[1] += b[1];
    }
    return Py_BuildValue("ny#", n, (const char*)tag, (Py_ssize_t)16);
}

static PyObject* encode_fr_list(PyObject* self, PyObject* args) {
    Py_buffer buf;
    long threads = 0; /* accepted for symmetry; creating str objects needs the GIL, so this runs on one thread */
    if (!PyArg_ParseTuple(args, "y*|l", &buf, &threads)) return NULL;
    if (buf.len % 32) {
        PyBuffer_Release(&buf);
        PyErr_SetString(PyExc_ValueError, "expected a multiple of 32 bytes");
        return NULL;
    }
    const Py_ssize_t n = buf.len / 32;
    PyObject* out = PyList_New(n);
    if (!out) {
        PyBuffer_Release(&buf);
        return NULL;
    }
    const uint8_t* src = (const uint8_t*)buf.buf;
    for (Py_ssize_t k = 0; k < n; k++) {
        PyObject* s = PyUnicode_New(43, 127);
        if (!s) {
            Py_DECREF(out);
            PyBuffer_Release(&buf);
            return NULL;
        }
        encode43(src + 32 * k, (uint8_t*)PyUnicode_1BYTE_DATA(s));
        PyList_SET_ITEM(out, k, s);
    }
    PyBuffer_Release(&buf);
    return out;
}

/* random_fr_rows(rows, T[, threads]) -> list of `rows` lists of T 43-character strings.  The text of a batch of rows is
 * produced by the thread pool (the calling thread keeps the GIL and takes part), then the str objects are created --
 * that part needs the GIL and sets the pace (~50 ns per element). */
static PyObject* random_fr_rows(PyObject* self, PyObject* args) {
    Py_ssize_t rows, T;
    long threads = 0;
    if (!PyArg_ParseTuple(args, "nn|l", &rows, &T, &threads)) return NULL;
    if (rows < 0 || T < 0 || (T && rows > ((Py_ssize_t)1 << 40) / T)) {
        PyErr_SetString(PyExc_ValueError, "bad shape");
        return NULL;
    }
    PyObject* out = PyList_New(rows);
    if (!out) return NULL;
    Py_ssize_t batch = T ? ((Py_ssize_t)1 << 19) / T : rows;      /* ~2^19 elements (22 MB of text) per buffer */
    if (batch < 1) batch = 1;
    const size_t buf_bytes = (size_t)43 * (size_t)(batch * T) + 64;
    uint8_t* text[2] = {(uint8_t*)malloc(buf_bytes), (uint8_t*)malloc(buf_bytes)};
    dec_job jobs[2][POOL_MAX + 1];
    int njobs[2] = {0, 0}, async_[2] = {0, 0};
    int failed = !text[0] || !text[1];
    if (failed) PyErr_NoMemory();
    /* batch b+1's text is generated by the pool while this thread (which must keep the GIL) creates batch b's str objects */
    const Py_ssize_t nbatches = rows ? (rows + batch - 1) / batch : 0;
    for (Py_ssize_t b = 0; b <= nbatches && !failed; b++) {
        const int cur = (int)(b & 1), prev = cur ^ 1;
        if (b < nbatches) {
            const Py_ssize_t r0 = b * batch, nr = rows - r0 < batch ? rows - r0 : batch, n = nr * T;
            const int TH = pick_threads(threads ? threads : 16, n);
            for (int t = 0; t < TH; t++) {
                jobs[cur][t].items = NULL;
                jobs[cur][t].dst = text[cur];
                jobs[cur][t].lo = n * t / TH;
                jobs[cur][t].hi = n * (t + 1) / TH;
                jobs[cur][t].want_tag = 0;
                jobs[cur][t].kind = 1;
            }
            njobs[cur] = TH;
            async_[cur] = pool_submit(jobs[cur], TH);
            if (!async_[cur])
                for (int t = 0; t < TH; t++) dec_worker(&jobs[cur][t]);
        }
        if (b > 0) {      /* materialise batch b-1 (its generation was joined at the end of the previous iteration) */
            const Py_ssize_t r0 = (b - 1) * batch, nr = rows - r0 < batch ? rows - r0 : batch;
            for (Py_ssize_t r = 0; r < nr && !failed; r++) {
                PyObject* row = PyList_New(T);
                if (!row) { failed = 1; break; }
                PyList_SET_ITEM(out, r0 + r, row);
                const uint8_t* src = text[prev] + (size_t)43 * (size_t)(r * T);
                for (Py_ssize_t k = 0; k < T; k++) {
                    PyObject* sobj = PyUnicode_New(43, 127);
                    if (!sobj) { failed = 1; break; }
                    memcpy(PyUnicode_1BYTE_DATA(sobj), src + 43 * k, 43);
                    PyList_SET_ITEM(row, k, sobj);
                }
            }
        }
        if (b < nbatches) {
            if (async_[cur]) pool_join();
            for (int t = 0; t < njobs[cur]; t++)
                if (jobs[cur][t].bad >= 0 && !failed) {
                    failed = 1;
                    PyErr_SetString(PyExc_OSError, "getrandom failed");
                }
        }
    }
    free(text[0]);
    free(text[1]);
    if (failed) {
        Py_DECREF(out);     /* list deallocation tolerates the NULL slots of rows / elements never reached */
        return NULL;
    }
    return out;
}

static PyObject* simd_level(PyObject* self, PyObject* args) { return PyLong_FromLong(have_avx2 ? 2 : 0); }
/* set_simd(level) -> level in force: 0 forces the scalar decoder, 2 asks for AVX2 (granted only when the CPU has it).
 * For A/B runs and for the tests that compare the two decoders (and their content tags) inside one process. */
static PyObject* set_simd(PyObject* self, PyObject* args) {
    long level;
    if (!PyArg_ParseTuple(args, "l", &level)) return NULL;
#if defined(__x86_64__)
    have_avx2 = (level >= 2 && __builtin_cpu_supports("avx2")) ? 1 : 0;
#endif
    return PyLong_FromLong(have_avx2 ? 2 : 0);
}

static PyMethodDef methods[] = {
    {"simd_level", simd_level, METH_NOARGS, "2 = AVX2 decoder active, 0 = scalar"},
    {"set_simd", set_simd, METH_VARARGS, "0 = force the scalar decoder, 2 = AVX2 when available; returns the level in force"},
    {"decode_fr_list", decode_fr_list, METH_VARARGS, "sequence of 43-char base64 Fr -> n*32 bytes big-endian"},
    {"decode_fr_list_into", decode_fr_list_into, METH_VARARGS, "decode into a caller-owned buffer (address, capacity)"},
    {"decode_fr_list_into_tagged", decode_fr_list_into_tagged, METH_VARARGS,
     "decode into a caller-owned buffer -> (n, 16-byte keyed content tag of the decoded bytes)"},
    {"random_fr_rows", random_fr_rows, METH_VARARGS, "rows x T uniform random Fr as 43-char base64 (getrandom + rejection)"},
    {"encode_fr_list", encode_fr_list, METH_VARARGS, "n*32 bytes big-endian -> list of 43-char base64 Fr"},
    {NULL, NULL, 0, NULL}};
static struct PyModuleDef moddef = {PyModuleDef_HEAD_INIT, "_wire", "Prove synapse text codec", -1, methods};

PyMODINIT_FUNC PyInit__wire(void) {
    memset(REV, -1, sizeof(REV));
    for (int i = 0; i < 64; i++) REV[(uint8_t)B64[i]] = (int8_t)i;
#if defined(__x86_64__)
    __builtin_cpu_init();
    have_avx2 = __builtin_cpu_supports("avx2") ? 1 : 0;
    {
        const char* e = getenv("KZG_WIRE_NO_AVX2");   /* A/B and the scalar path's tests */
        if (e && *e == '1') have_avx2 = 0;
    }
#endif
    {   /* the tag key: 512 random bits per process (getrandom; /dev/urandom semantics) */
        ssize_t got = getrandom(TAG_KEY, sizeof(TAG_KEY), 0);
        if (got != (ssize_t)sizeof(TAG_KEY)) {
            PyErr_SetString(PyExc_OSError, "getrandom failed: cannot key the content tag");
            return NULL;
        }
    }
    return PyModule_Create(&moddef);
}
