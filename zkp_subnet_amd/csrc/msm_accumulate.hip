// msm_accumulate.hip -- stage 2 of the Pippenger MSM, THE HOT KERNEL (BASELINE.json metric): every lane sums a
// fixed chunk of the sorted entries with mixed XYZZ additions out of the window tables; then the carries of runs that span
// chunks are folded into their buckets.  k_msm_accumulate is bound by the issue rate of v_mad_u64_u32 (3546 per addition).
#include "msm_dev.hip.h"

// ------------------------------------------------------------------------------------------------ accumulate
// largest b with offsets[b] <= e  (then offsets[b+1] > e: b is non-empty and contains sorted entry e)
KZG_DEV uint32_t bucket_of(const uint32_t* __restrict__ offsets, uint32_t nbuckets, uint32_t e) {
    uint32_t b_lo = 0, b_hi = nbuckets - 1;
    while (b_lo < b_hi) {
        uint32_t mid = (b_lo + b_hi + 1) >> 1;
        if (offsets[mid] <= e) b_lo = mid; else b_hi = mid - 1;
    }
    return b_lo;
}
// Each lane owns sorted entries [t*K, (t+1)*K).  A bucket run that began in an earlier chunk is summed into
// carries[t] (at most one per chunk: only the FIRST run of a chunk can have begun earlier); every run that
// begins inside the chunk is stored straight to its bucket -- the lane that sees a run begin is its only writer.
#ifndef KZG_ACC_MIN_WAVES
#define KZG_ACC_MIN_WAVES 2
#endif
__global__ void __launch_bounds__(256, KZG_ACC_MIN_WAVES) k_msm_accumulate(const g1_affine_t* __restrict__ table,
                                                         const uint32_t* __restrict__ offsets,
                                                         const uint32_t* __restrict__ sorted, uint32_t nbuckets,
                                                         uint32_t chunk, uint32_t nchunks,
                                                         g1_xyzz_t* __restrict__ buckets,
                                                         g1_xyzz_t* __restrict__ carries,
                                                         uint32_t* __restrict__ carry_key) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nchunks) return;
    const uint32_t total = offsets[nbuckets];
    const uint32_t lo = t * chunk;
    if (lo >= total) {  // zero digits were dropped: fewer entries than the host-side bound
        carry_key[t] = NONE_KEY;
        return;
    }
    const uint32_t hi = min(lo + chunk, total);
    uint32_t cur = bucket_of(offsets, nbuckets, lo);
    uint32_t boundary = offsets[cur + 1];
    bool pending_carry = offsets[cur] < lo;  // first run began in an earlier chunk
    uint32_t my_carry_key = NONE_KEY;
    g1_xyzz_t acc;
    g1_set_inf(acc);
    uint32_t v_cur = sorted[lo];
    uint32_t w_cur[28];
    {
        const uint4* q = reinterpret_cast<const uint4*>(table + (v_cur & 0x7fffffffu));
#pragma unroll
        for (int i = 0; i < 7; i++) { uint4 t4 = q[i]; w_cur[4*i]=t4.x; w_cur[4*i+1]=t4.y; w_cur[4*i+2]=t4.z; w_cur[4*i+3]=t4.w; }
    }
    for (uint32_t e = lo; e < hi; e++) {
        if (e == boundary) {  // run of `cur` is complete
            if (pending_carry) {
                store_xyzz(&carries[t], acc);
                my_carry_key = cur;
                pending_carry = false;
            } else {
                store_xyzz(&buckets[cur], acc);
            }
            g1_set_inf(acc);
            // next non-empty bucket (empty ones keep the zero = infinity of the memset): a short linear probe, then a
            // binary search -- skewed inputs (all scalars equal ...) leave stretches of 10^4..10^5 empty buckets
            cur++;
            boundary = offsets[cur + 1];
            for (int g = 0; g < 3 && boundary == e; g++) {
                cur++;
                boundary = offsets[cur + 1];
            }
            if (boundary == e) {
                cur = bucket_of(offsets, nbuckets, e);
                boundary = offsets[cur + 1];
            }
        }
        // software pipeline: the packed words of the NEXT entry's point are requested before this entry's addition
        uint32_t wn[28];
        const uint32_t vn = (e + 1 < hi) ? sorted[e + 1] : v_cur;
        {
            const uint4* q = reinterpret_cast<const uint4*>(table + (vn & 0x7fffffffu));
#pragma unroll
            for (int i = 0; i < 7; i++) { uint4 t4 = q[i]; wn[4*i]=t4.x; wn[4*i+1]=t4.y; wn[4*i+2]=t4.z; wn[4*i+3]=t4.w; }
        }
        g1_aff28 p;
#pragma unroll
        for (int i = 0; i < 14; i++) { p.x.l[i] = w_cur[i]; p.y.l[i] = w_cur[14 + i]; }
        g1_neg_aff(p, v_cur >> 31);
        g1_madd_checked<true>(acc, p);
#pragma unroll
        for (int i = 0; i < 28; i++) w_cur[i] = wn[i];
        v_cur = vn;
    }
    if (pending_carry) {
        store_xyzz(&carries[t], acc);
        my_carry_key = cur;
    } else {
        store_xyzz(&buckets[cur], acc);
    }
    carry_key[t] = my_carry_key;
}

// ---- carries -> buckets.  The carries of bucket b sit at chunks t0+1 .. t1 with t0 = offsets[b] / K and
// t1 = (offsets[b+1] - 1) / K, so carry t knows its position i = t - t0 - 1 inside its run of len = t1 - t0 carries
// without any scan.  A per-run binary tree (step d: element i adds element i + d when i % 2d == 0) folds every run
// in ceil(log2(len)) fully parallel steps, for ANY scalar distribution; the heads (i == 0) then go into the buckets.
KZG_DEV bool carry_pos(const uint32_t* __restrict__ offsets, const uint32_t* __restrict__ carry_key, uint32_t chunk,
                       uint32_t t, uint32_t& key, uint32_t& i, uint32_t& len) {
    key = carry_key[t];
    if (key == NONE_KEY) return false;
    const uint32_t t0 = offsets[key] / chunk, t1 = (offsets[key + 1] - 1u) / chunk;
    i = t - t0 - 1u;
    len = t1 - t0;
    return true;
}
// longest carry run, from the bucket offsets alone (so it can be read back while the accumulate kernel runs).  The same
// pass marks the EMPTY buckets as infinity: the accumulate kernel stores every non-empty bucket exactly once (the lane in
// whose chunk its run begins), so nothing else needs clearing -- this replaces a memset of the whole bucket array
// (117 MB, ~25 us, at c = 20) by stores for the buckets that actually are empty (none for well-spread scalars).
__global__ void __launch_bounds__(256) k_fold_maxlen(const uint32_t* __restrict__ offsets, uint32_t nbuckets,
                                                      uint32_t chunk, uint32_t* __restrict__ max_len,
                                                      g1_xyzz_t* __restrict__ buckets) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nbuckets) return;
    const uint32_t lo = offsets[b], hi = offsets[b + 1];
    if (hi == lo) {
        uint4* q = reinterpret_cast<uint4*>(&buckets[b]);
#pragma unroll
        for (int i = 0; i < 14; i++) q[i] = make_uint4(0u, 0u, 0u, 0u);
        return;
    }
    const uint32_t len = (hi - 1u) / chunk - lo / chunk;
    if (len > 1) atomicMax(max_len, len);
}
__global__ void __launch_bounds__(256) k_fold_step(const uint32_t* __restrict__ offsets,
                                                    const uint32_t* __restrict__ carry_key, uint32_t chunk,
                                                    uint32_t nchunks, uint32_t d, g1_xyzz_t* __restrict__ carries) {
    tail_priority();
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nchunks) return;
    uint32_t key, i, len;
    if (!carry_pos(offsets, carry_key, chunk, t, key, i, len)) return;
    if ((i & (2u * d - 1u)) || i + d >= len) return;
    g1_xyzz_t a, b, r;
    load_xyzz(a, &carries[t]);
    load_xyzz(b, &carries[t + d]);
    g1_add(r, a, b);
    store_xyzz(&carries[t], r);
}
__global__ void __launch_bounds__(256) k_fold_heads(const uint32_t* __restrict__ offsets,
                                                     const uint32_t* __restrict__ carry_key, uint32_t chunk,
                                                     uint32_t nchunks, const g1_xyzz_t* __restrict__ carries,
                                                     g1_xyzz_t* __restrict__ buckets) {
    tail_priority();
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nchunks) return;
    uint32_t key, i, len;
    if (!carry_pos(offsets, carry_key, chunk, t, key, i, len) || i != 0) return;
    g1_xyzz_t a, b, r;
    load_xyzz(a, &buckets[key]);
    load_xyzz(b, &carries[t]);
    g1_add<true>(r, a, b);   // one full addition per lane, throughput-bound like the wide tree levels: inlined products
    store_xyzz(&buckets[key], r);
}


// cooperative variants of the fold kernels (64 carries per workgroup) for the small-slice, latency-bound regime
__global__ void __launch_bounds__(256) k_fold_step_coop(const uint32_t* __restrict__ offsets,
                                                         const uint32_t* __restrict__ carry_key, uint32_t chunk,
                                                         uint32_t nchunks, uint32_t d, g1_xyzz_t* __restrict__ carries) {
    tail_priority();
    __shared__ CoopLds sm;
    const uint32_t t = blockIdx.x * 64 + (threadIdx.x & 63);
    uint32_t key = 0, i = 0, len = 0;
    bool active = t < nchunks && carry_pos(offsets, carry_key, chunk, t, key, i, len);
    active = active && !(i & (2u * d - 1u)) && i + d < len;
    if (!__syncthreads_or(active)) return;
    coop_add(sm, &carries[t], &carries[t], &carries[active ? t + d : t], active);
}
__global__ void __launch_bounds__(256) k_fold_heads_coop(const uint32_t* __restrict__ offsets,
                                                          const uint32_t* __restrict__ carry_key, uint32_t chunk,
                                                          uint32_t nchunks, const g1_xyzz_t* __restrict__ carries,
                                                          g1_xyzz_t* __restrict__ buckets) {
    tail_priority();
    __shared__ CoopLds sm;
    const uint32_t t = blockIdx.x * 64 + (threadIdx.x & 63);
    uint32_t key = 0, i = 0, len = 0;
    const bool active = t < nchunks && carry_pos(offsets, carry_key, chunk, t, key, i, len) && i == 0;
    if (!__syncthreads_or(active)) return;
    g1_xyzz_t* dst = &buckets[active ? key : 0];
    coop_add(sm, dst, dst, &carries[active ? t : 0], active);
}

// the late fold steps of a short row: few pairs are left (about nchunks / 2d), so one WAVE per carry -- the inactive ones
// leave at once -- runs each addition lane-parallel (~6 us a step against ~13 for the cooperative form)
__global__ void __launch_bounds__(64) k_fold_step_lp(const uint32_t* __restrict__ offsets,
                                                      const uint32_t* __restrict__ carry_key, uint32_t chunk,
                                                      uint32_t d, g1_xyzz_t* __restrict__ carries) {
    tail_priority();
    __shared__ LpScratch sm;
    const uint32_t t = blockIdx.x;
    uint32_t key, i, len;
    if (!carry_pos(offsets, carry_key, chunk, t, key, i, len)) return;
    if ((i & (2u * d - 1u)) || i + d >= len) return;
    lp_add(sm, &carries[t], &carries[t], &carries[t + d], lp_lane());
}

// Short rows (few buckets, short carry runs): ONE launch instead of ceil(log2 max run) fold steps + the heads -- one wave
// per BUCKET adds its carries (chunks t0 + 1 .. t1 of its run: every one of them begins inside the run, so each holds a
// carry of this bucket) one after the other and then the sum to the bucket.  Serial in the run length, hence only
// behind the host's check of the fold-depth word (msm_fold_bucket_ok); a 2^12 row has <= 4096 buckets of ~5 carries.
__global__ void __launch_bounds__(64) k_fold_bucket_lp(const uint32_t* __restrict__ offsets, uint32_t chunk,
                                                        const g1_xyzz_t* __restrict__ carries,
                                                        g1_xyzz_t* __restrict__ buckets) {
    tail_priority();
    __shared__ LpScratch sm;
    __shared__ __align__(16) g1_xyzz_t acc;
    const uint32_t b = blockIdx.x;
    const uint32_t lo = offsets[b], hi = offsets[b + 1];
    if (hi == lo) return;
    const uint32_t t0 = lo / chunk, t1 = (hi - 1u) / chunk;
    if (t1 == t0) return;
    const LpLane k = lp_lane();
    const g1_xyzz_t* src = &carries[t0 + 1];
    for (uint32_t t = t0 + 2; t <= t1; t++) {
        lp_add(sm, &acc, src, &carries[t], k);
        lp_sync();
        src = &acc;
    }
    lp_add(sm, &buckets[b], &buckets[b], src, k);
}

// same merge as k_msm_tree_level, 64 operations per 256-thread workgroup, for the narrow (latency-bound) levels

// ------------------------------------------------------------------------------------------------ launchers
void launch_msm_accumulate(hipStream_t s, const MsmShape& sh, const g1_affine_t* table, const uint32_t* offsets,
                           const uint32_t* sorted, g1_xyzz_t* buckets, g1_xyzz_t* carries, uint32_t* carry_key,
                           uint32_t nchunks) {
    if (!nchunks) return;
    k_msm_accumulate<<<nblk(nchunks, 256), 256, 0, s>>>(table, offsets, sorted, sh.nbuckets, (uint32_t)sh.chunk,
                                                        nchunks, buckets, carries, carry_key);
}
void launch_fold_maxlen(hipStream_t s, const uint32_t* offsets, uint32_t nbuckets, uint32_t chunk, uint32_t* max_len,
                        g1_xyzz_t* buckets) {
    k_fold_maxlen<<<nblk(nbuckets, 256), 256, 0, s>>>(offsets, nbuckets, chunk, max_len, buckets);
}
// cooperative fold kernels (4 waves per 64 carries) up to this many chunks, one lane per carry above (A/B: 2^16 batched
// commit+open, 65536 chunks: fixup 0.090 -> 0.054 ms; no gain at 131072 chunks)
#ifndef KZG_FOLD_LP_MAX
#define KZG_FOLD_LP_MAX LP_MAX_OPS
#endif
#ifndef KZG_FOLD_COOP_MAX
#define KZG_FOLD_COOP_MAX 65536
#endif
void launch_fold_step(hipStream_t s, const uint32_t* offsets, const uint32_t* carry_key, uint32_t chunk,
                      uint32_t nchunks, uint32_t d, g1_xyzz_t* carries) {
    if (!nchunks) return;
    if (nchunks > KZG_FOLD_COOP_MAX) k_fold_step<<<nblk(nchunks, 256), 256, 0, s>>>(offsets, carry_key, chunk, nchunks, d, carries);
#if !defined(KZG_NO_LP) && !defined(KZG_NO_FOLD_LP)
    else if (nchunks / (2 * d) <= KZG_FOLD_LP_MAX) k_fold_step_lp<<<nchunks, 64, 0, s>>>(offsets, carry_key, chunk, d, carries);
#endif
    else k_fold_step_coop<<<nblk(nchunks, 64), 256, 0, s>>>(offsets, carry_key, chunk, nchunks, d, carries);
}
#ifndef KZG_FOLD_BUCKET_MAX
#define KZG_FOLD_BUCKET_MAX 4096   // buckets (one wave each)
#endif
#ifndef KZG_FOLD_BUCKET_RUN
#define KZG_FOLD_BUCKET_RUN 24     // carries of the longest run: the chain one wave walks
#endif
bool msm_fold_bucket_ok(uint32_t nbuckets, uint32_t max_run) {
#if defined(KZG_NO_LP) || defined(KZG_NO_FOLD_LP)
    (void)nbuckets; (void)max_run;
    return false;
#else
    return nbuckets <= KZG_FOLD_BUCKET_MAX && max_run <= KZG_FOLD_BUCKET_RUN;
#endif
}
void launch_fold_bucket(hipStream_t s, const uint32_t* offsets, uint32_t chunk, uint32_t nbuckets, const g1_xyzz_t* carries,
                        g1_xyzz_t* buckets) {
    if (nbuckets) k_fold_bucket_lp<<<nbuckets, 64, 0, s>>>(offsets, chunk, carries, buckets);
}
void launch_fold_heads(hipStream_t s, const uint32_t* offsets, const uint32_t* carry_key, uint32_t chunk,
                       uint32_t nchunks, const g1_xyzz_t* carries, g1_xyzz_t* buckets) {
    if (!nchunks) return;
    if (nchunks > KZG_FOLD_COOP_MAX)
        k_fold_heads<<<nblk(nchunks, 256), 256, 0, s>>>(offsets, carry_key, chunk, nchunks, carries, buckets);
    else
        k_fold_heads_coop<<<nblk(nchunks, 64), 256, 0, s>>>(offsets, carry_key, chunk, nchunks, carries, buckets);
}
