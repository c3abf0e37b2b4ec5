// srs.hip -- the resident SRS: setup loaders (caller memory, the setup FILE of the reference's start path, per-device
// slices of it), the synthetic tau-derived SRS, the window tables 2^off[w] P_j and their read-back.  Replaces the prover's
// setup / precompute file loading (reference base/miner.py:75-84, utils/config.py:124-164, Makefile:63-74).
#include "ctx.hip.h"

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

using namespace kzg_impl;

namespace {

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
// A (re)load never touches the serving table until it has succeeded: the new table is built in its own allocation and
// swapped in at the end (288 GB of HBM hold both: mainnet's 34 GB twice is nothing).  Only when the second allocation does
// not fit is the old table given up first -- then, and only then, a failure leaves the context without an SRS.
int plan_table(kzg_ctx* ctx, uint64_t n_points, int scale, int mscale, TableSpec& sp) {
    if (mscale < 0 || scale < mscale || scale - mscale > 30) return fail(ctx, KZG_E_ARG, "bad scale / machines_scale");
    const uint64_t T = (uint64_t)1 << (scale - mscale);
    // (one slice may be SHORTER than 2^scale: a setup that supports lower degrees only, or one segment of a flat SRS)
    if (n_points == 0 || (n_points % T && !(mscale == 0 && n_points < T)))
        return fail(ctx, KZG_E_ARG, "SRS length must be a whole number of worker slices");
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

}  // namespace

extern "C" {

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
// first_slice / slice_stride: resident slice k is slice first_slice + k * slice_stride of the source (0 / 1: the source as it
// is) -- a device that serves the worker indices i = g (mod G) holds exactly those slices (kzg_load_srs_file_slices).
// base_point: the source starts at this point of the file (one contiguous SEGMENT of a flat SRS: kzg_load_srs_file_range).
static int load_srs_common(kzg_ctx* ctx, const uint8_t* data, int fd, uint64_t n_points, int scale, int machines_scale,
                           bool compressed, uint64_t first_slice = 0, uint64_t slice_stride = 1, uint64_t base_point = 0) {
    if (!ctx || (!data && fd < 0) || !slice_stride) return KZG_E_ARG;
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
        // resident points [first, first + cnt) -> the pinned tile, one piece per source slice (the whole tile in one piece
        // when the source is taken as it is): resident point p sits at source point ((p / T) * stride + first_slice) * T + p % T
        for (uint64_t done = 0; done < cnt;) {
            const uint64_t p0 = first + done, k = p0 / sp.T, o = p0 % sp.T;
            const uint64_t len = slice_stride == 1 ? cnt - done : std::min(cnt - done, sp.T - o);
            const uint64_t src = base_point + (k * slice_stride + first_slice) * sp.T + o;
            if (data) copy_parallel(pin.p[b] + rec * done, data + rec * src, len * rec);
            else if (const int e = read_parallel(fd, pin.p[b] + rec * done, len * rec, (off_t)(rec * src))) {
                (void)hipStreamSynchronize(L.stream);     // copies of the other buffer may still be in flight
                return fail(ctx, KZG_E_ARG, e < 0 ? std::string("setup file: the file shrank during the load (end of file before the last point)")
                                                  : std::string("setup file: read failed (") + strerror(e) + ")");
            }
            done += len;
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
static int load_srs_file_common(kzg_ctx* ctx, const char* path, int compressed, int scale, int machines_scale, bool sliced,
                                uint32_t first_slice, uint32_t slice_stride, bool ranged = false, uint64_t first_point = 0,
                                uint64_t range_points = 0) {
    if (!ctx || !path || (sliced && !slice_stride)) return KZG_E_ARG;
    if (machines_scale < 0 || scale < machines_scale || scale - machines_scale > 30) return fail(ctx, KZG_E_ARG, "bad scale / machines_scale");
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
    uint64_t n_points = (uint64_t)st.st_size / rec;
    if (sliced) {
        const uint64_t T = (uint64_t)1 << (scale - machines_scale);
        if (n_points % T) {
            close(fd);
            return fail(ctx, KZG_E_ARG, "SRS length must be a whole number of worker slices");
        }
        const uint64_t file_slices = n_points / T;
        if (first_slice >= file_slices) {
            close(fd);
            return fail(ctx, KZG_E_ARG, "no slice of the setup file falls to this context (first_slice beyond the file)");
        }
        n_points = ((file_slices - first_slice + slice_stride - 1) / slice_stride) * T;      // only what this context serves
        // (no read-ahead advice for the whole file: this context touches 1 / stride of it)
    } else if (ranged) {
        if (!range_points || first_point > n_points || range_points > n_points - first_point) {
            close(fd);
            return fail(ctx, KZG_E_ARG, "point range outside the setup file");
        }
        n_points = range_points;
        (void)posix_fadvise(fd, (off_t)(first_point * rec), (off_t)(range_points * rec), POSIX_FADV_WILLNEED);
    } else {
        (void)posix_fadvise(fd, 0, st.st_size, POSIX_FADV_SEQUENTIAL);
        (void)posix_fadvise(fd, 0, st.st_size, POSIX_FADV_WILLNEED);
    }
    const int rc = load_srs_common(ctx, nullptr, fd, n_points, scale, machines_scale, compressed != 0, sliced ? first_slice : 0,
                                   sliced ? slice_stride : 1, ranged ? first_point : 0);
    close(fd);
    return rc;
}
int kzg_load_srs_file(kzg_ctx* ctx, const char* path, int compressed, int scale, int machines_scale) {
    return load_srs_file_common(ctx, path, compressed, scale, machines_scale, false, 0, 1);
}
// Only the slices one device of a multi-GPU host serves: resident slice k = file slice first_slice + k * slice_stride (worker
// index i = g (mod G) on device g: first_slice = g, slice_stride = G).  pread touches just those byte ranges -- mainnet 24 / 8
// on G devices holds 34 / G GB of tables each and starts in ~1 / G of the single-device time (kzg_multi_load_srs_file).
int kzg_load_srs_file_slices(kzg_ctx* ctx, const char* path, int compressed, int scale, int machines_scale, uint32_t first_slice,
                             uint32_t slice_stride) {
    return load_srs_file_common(ctx, path, compressed, scale, machines_scale, true, first_slice, slice_stride);
}
// One contiguous SEGMENT of a flat SRS: file points [first_point, first_point + n_points) become resident points [0, n_points)
// of a single slice (machines_scale 0; `scale` >= log2(n_points) sizes the Pippenger window) -- what device g of a host holds
// when ONE MSM is sharded by SRS segment over its GPUs (kzg_multi_msm; BASELINE.json configs[3]).
int kzg_load_srs_file_range(kzg_ctx* ctx, const char* path, int compressed, uint64_t first_point, uint64_t n_points, int scale) {
    if (scale < 0 || scale > 30 || n_points > ((uint64_t)1 << scale)) return fail(ctx, KZG_E_ARG, "kzg_load_srs_file_range: n_points must be <= 2^scale");
    return load_srs_file_common(ctx, path, compressed, scale, 0, false, 0, 1, true, first_point, n_points);
}
// seconds of the last successful kzg_load_srs*: [0] host copies file/buffer -> pinned tiles, [1] host waits for the GPU
// (upload + decode / decompression), [2] window-table build, [3] the whole call
int kzg_set_srs_subgroup_check(kzg_ctx* ctx, int enable) {
    if (!ctx) return KZG_E_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    ctx->srs_subgroup_check = enable != 0;
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

}  // extern "C"
