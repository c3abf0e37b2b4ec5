// msm_sort.hip -- stage 1 of the Pippenger MSM (pipeline: msm.hip.h): scalars -> signed window digits -> a two-level
// counting sort of the (bucket, table index) entries with every per-entry atomic in LDS.  Its last kernel marks the empty
// buckets, takes the fold depth and publishes it to the lane's pinned page (SortTail).  No reference source exists for this
// path (reference neurons/miner.py:39,48 only calls the external prover); checked bit-for-bit against oracle/ in tests/test_gpu_msm.py.
#include "msm_dev.hip.h"

#include <cstdlib>

// ------------------------------------------------------------------------------------------------ digits
// (limb_at / window_bits / load_scalar: msm_dev.hip.h -- the synthetic SRS generator reads scalars the same way)
// signed-digit recoding: digit in [-2^(c-1)+1, 2^(c-1)]; returns magnitude (0 = skip), sets neg, updates carry
KZG_DEV uint32_t signed_digit(const uint32_t* s, int w, const WinLayout& lay, uint32_t& carry, uint32_t& neg) {
    const int lo = lay.off[w], c = lay.off[w + 1] - lo;
    uint32_t d = window_bits(s, lo, c) + carry;
    const uint32_t half = 1u << (c - 1);
    neg = d > half;
    carry = neg;
    return neg ? (1u << c) - d : d;
}

// ---- counting sort of the (bucket key, table index) entries, two levels, every per-entry atomic in LDS ----------
// Level 1 splits the key's high bits into npart = 2^hbits partitions (per-block LDS histogram, ONE global atomic
// per block and partition to reserve room); level 2 gives each partition to one workgroup that histograms the low
// bits in LDS, emits the bucket offsets, and scatters inside its own (L2-resident) slice.  The previous version
// issued one global atomic per entry (24 G/s chip-wide: 0.9 ms at 2^20, 8 ms at 2^22).
#define SORT_MAXPART 4096
struct SortShape {
    uint64_t n, total, srs_offset, srs_stride;  // n scalars per set, total = n * sets
    const uint32_t* scalars2;                   // second scalar set (batch of two MSMs over the same points) or null
    int mont, mont2, keybits, hbits, lbits;     // key = set << keybits | digit magnitude - 1 = (part << lbits) | low
    uint32_t spb;                               // scalars per workgroup in the two level-1 kernels
};
// h[key]++ in LDS, returning the old value.  When every active lane of the wave holds the same key (all scalars
// equal, constant or sparse polynomials ...) one lane adds the whole count: same-address LDS atomics serialise.
KZG_DEV uint32_t lds_bump(uint32_t* h, uint32_t key) {
    const uint64_t act = __ballot(1);
    const uint32_t k0 = __builtin_amdgcn_readfirstlane(key);
    if (__ballot(key == k0) == act) {
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(act >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)act, 0u));
        uint32_t base = 0;
        if (rank == 0) base = atomicAdd(&h[k0], (uint32_t)__popcll(act));
        return __builtin_amdgcn_readfirstlane(base) + rank;
    }
    return atomicAdd(&h[key], 1u);
}
// inclusive scan of one value per thread over a 1024-thread workgroup: six shuffle steps inside each wave, the 16 wave
// totals through LDS (two barriers in all; the ten-step LDS scan this replaces had twenty -- 2-4 us of a short row's
// single-workgroup sort kernels).  Returns the inclusive prefix; `total` = sum over the workgroup.  wtot: 16 words of LDS.
KZG_DEV uint32_t block_scan_1024(uint32_t v, uint32_t* wtot, uint32_t& total) {
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t x = __shfl_up(inc, d, 64);
        if (lane >= (uint32_t)d) inc += x;
    }
    if (lane == 63) wtot[wave] = inc;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (uint32_t w = 0; w < 16; w++) {
        const uint32_t x = wtot[w];
        if (w < wave) base += x;
        tot += x;
    }
    total = tot;
    __syncthreads();   // wtot may be reused by the next scan
    return base + inc;
}
template <class F>
KZG_DEV void for_each_entry(const uint32_t* __restrict__ scalars, const SortShape& ss, const WinLayout& lay, F&& f) {
    const uint64_t base = (uint64_t)blockIdx.x * ss.spb;
    for (uint32_t r = 0; r < ss.spb / 256; r++) {
        const uint64_t g = base + r * 256 + threadIdx.x;
        if (g >= ss.total) break;
        const bool second = g >= ss.n;
        const uint64_t j = second ? g - ss.n : g;
        uint32_t s[8];
        load_scalar(s, second ? ss.scalars2 : scalars, j, second ? ss.mont2 : ss.mont);
        const uint32_t set_bit = second ? 1u << ss.keybits : 0u;
        uint32_t carry = 0, neg;
        for (int w = 0; w < lay.nwin; w++) {
            const uint32_t mag = signed_digit(s, w, lay, carry, neg);
            if (mag) f((mag - 1) | set_bit, (uint32_t)((uint64_t)w * ss.srs_stride + ss.srs_offset + j) | (neg << 31));
        }
    }
}
// partition sizes: 1024 lanes x R scalars each.  R = 4 for long inputs (all four loads in flight before the first digit is
// extracted); R = 1 for short ones, where 4 scalars x nwin LDS atomics per lane on a handful of workgroups is a ~50 us chain
template <int R>
__global__ void __launch_bounds__(1024) k_sort_count(const uint32_t* __restrict__ scalars, const SortShape ss,
                                                      const WinLayout lay, uint32_t* __restrict__ part_count) {
    __shared__ uint32_t h[SORT_MAXPART];
    const uint32_t npart = 1u << ss.hbits;
    for (uint32_t i = threadIdx.x; i < npart; i += 1024) h[i] = 0;
    __syncthreads();
    uint32_t sc[R][8];
    bool live[R], second[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
        const uint64_t g = (uint64_t)blockIdx.x * (1024 * R) + r * 1024 + threadIdx.x;
        live[r] = g < ss.total;
        second[r] = live[r] && g >= ss.n;
        const uint64_t j = second[r] ? g - ss.n : g;
        if (live[r]) load_scalar(sc[r], second[r] ? ss.scalars2 : scalars, j, second[r] ? ss.mont2 : ss.mont);
    }
#pragma unroll
    for (int r = 0; r < R; r++) {
        if (!live[r]) continue;
        const uint32_t set_bit = second[r] ? 1u << ss.keybits : 0u;
        uint32_t carry = 0, neg;
        for (int w = 0; w < lay.nwin; w++) {
            const uint32_t mag = signed_digit(sc[r], w, lay, carry, neg);
            if (mag) lds_bump(h, ((mag - 1) | set_bit) >> ss.lbits);
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < npart; i += 1024)
        if (h[i]) atomicAdd(&part_count[i], h[i]);
}
// part_base[0..npart] = exclusive scan of min(counts, clamp) (npart <= 4096).  `counts` is the count pass's histogram
// (exact mode) or the partition pass's cursors (fast mode, clamp = region capacity).  Housekeeping that would otherwise
// be more memsets on the stream (~4.5 us each, and short rows are nothing but such latencies): the counts are zeroed
// again once read (so the NEXT sort finds them clean), the fold-depth word of this MSM is reset, and the exact mode
// clears the overflow word a failed fast attempt left behind.
__global__ void __launch_bounds__(1024) k_sort_part_scan(uint32_t* __restrict__ counts, uint32_t npart, uint32_t clamp,
                                                          uint32_t* __restrict__ part_base,
                                                          uint32_t* __restrict__ max_len_word,
                                                          uint32_t* __restrict__ overflow_word_or_null) {
    __shared__ uint32_t part[16];
    const uint32_t t = threadIdx.x;
    const uint32_t per = (npart + 1023u) / 1024u;
    const uint32_t lo = t * per, hi = min(lo + per, npart);
    uint32_t v = 0;
    for (uint32_t i = lo; i < hi; i++) v += min(counts[i], clamp);
    uint32_t all;
    const uint32_t incl = block_scan_1024(v, part, all);
    uint32_t run = incl - v;
    for (uint32_t i = lo; i < hi; i++) {
        part_base[i] = run;
        run += min(counts[i], clamp);
        counts[i] = 0;
    }
    if (t == 1023) part_base[npart] = all;
    if (t == 0) {
        *max_len_word = 0;
        if (overflow_word_or_null) *overflow_word_or_null = 0;
    }
}
__global__ void __launch_bounds__(256) k_sort_partition(const uint32_t* __restrict__ scalars, const SortShape ss,
                                                         const WinLayout lay, const uint32_t* __restrict__ part_base,
                                                         uint32_t* __restrict__ part_cursor, uint2* __restrict__ parted) {
    __shared__ uint32_t h[SORT_MAXPART];
    __shared__ uint32_t base[SORT_MAXPART];
    const uint32_t npart = 1u << ss.hbits;
    for (uint32_t i = threadIdx.x; i < npart; i += 256) h[i] = 0;
    __syncthreads();
    for_each_entry(scalars, ss, lay, [&](uint32_t key, uint32_t) { lds_bump(h, key >> ss.lbits); });
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < npart; i += 256) {
        base[i] = h[i] ? part_base[i] + atomicAdd(&part_cursor[i], h[i]) : 0u;
        h[i] = 0;
    }
    __syncthreads();
    const uint32_t lmask = (1u << ss.lbits) - 1u;
    for_each_entry(scalars, ss, lay, [&](uint32_t key, uint32_t val) {
        const uint32_t q = key >> ss.lbits;
        const uint32_t pos = base[q] + lds_bump(h, q);
        parted[pos] = make_uint2(key & lmask, val);
    });
}
// ---- level 1, staged: one scalar per lane (its <= 32 digits stay in registers: no second decode), the workgroup's
// entries are ordered by partition inside LDS and leave as runs of consecutive addresses instead of 8-byte singles
// (the direct scatter wrote 3.5x the bytes it stored).  Used when nwin <= 32 (c >= 8); otherwise k_sort_partition.
#define SORT1_MAXW 32
#define SORT1_STAGE 13312  // entries per workgroup: 104 KB of (key_low | partition << 16, value); 1024 scalars at 13 windows
// region_cap != 0 (FAST mode, no count pass): partition q owns the fixed region [q * region_cap, (q + 1) * region_cap) of
// `parted`; a workgroup whose run would not fit raises *overflow and drops that partition's entries -- the host then
// reruns the sort in exact mode (count pass + exact bases), see msm_core.  region_cap == 0: exact bases from part_base.
__global__ void __launch_bounds__(1024) k_sort_partition_staged(const uint32_t* __restrict__ scalars, const SortShape ss,
                                                                 const WinLayout lay, uint32_t spb,
                                                                 const uint32_t* __restrict__ part_base,
                                                                 uint32_t* __restrict__ part_cursor,
                                                                 uint2* __restrict__ parted, uint32_t region_cap,
                                                                 uint32_t* __restrict__ overflow) {
    __shared__ uint32_t h[SORT_MAXPART];     // counts, then (global base - local offset) per partition
    __shared__ uint32_t skip[SORT_MAXPART / 32];
    __shared__ uint32_t loff[SORT_MAXPART];  // local exclusive offsets
    __shared__ uint32_t wsum[16];
    __shared__ uint2 stage[SORT1_STAGE];  // .x = key_low (<= 12 bits) | partition << 16
    const uint32_t t = threadIdx.x;
    const uint32_t npart = 1u << ss.hbits;
    for (uint32_t i = t; i < npart; i += 1024) h[i] = 0;
    if (t < SORT_MAXPART / 32) skip[t] = 0;
    __syncthreads();
    // 1. digits -> registers, rank inside (workgroup, partition)
    uint32_t keyn[SORT1_MAXW], rk[SORT1_MAXW];
    const uint64_t g = (uint64_t)blockIdx.x * spb + t;
    const bool live = t < spb && g < ss.total;
    const bool second = live && g >= ss.n;
    const uint64_t j = second ? g - ss.n : g;
#pragma unroll
    for (int w = 0; w < SORT1_MAXW; w++) keyn[w] = 0xffffffffu;
    if (live) {
        uint32_t sc[8];
        load_scalar(sc, second ? ss.scalars2 : scalars, j, second ? ss.mont2 : ss.mont);
        const uint32_t set_bit = second ? 1u << ss.keybits : 0u;
        uint32_t carry = 0, neg;
#pragma unroll
        for (int w = 0; w < SORT1_MAXW; w++) {
            if (w < lay.nwin) {
                const uint32_t mag = signed_digit(sc, w, lay, carry, neg);
                if (mag) {
                    const uint32_t key = (mag - 1) | set_bit;
                    keyn[w] = key | (neg << 31);
                    rk[w] = lds_bump(h, key >> ss.lbits);
                }
            }
        }
    }
    __syncthreads();
    // 2. exclusive scan of the counts
    const uint32_t per = (npart + 1023u) / 1024u;
    const uint32_t b0 = t * per, b1 = min(b0 + per, npart);
    uint32_t sum = 0;
    for (uint32_t i = b0; i < b1; i++) sum += h[i];
    uint32_t count;
    uint32_t run = block_scan_1024(sum, wsum, count) - sum;
    // 3. reserve the global run of every non-empty partition; h becomes (global position - local position)
    for (uint32_t i = b0; i < b1; i++) {
        const uint32_t c = h[i];
        loff[i] = run;
        if (c) {
            const uint32_t old = atomicAdd(&part_cursor[i], c);
            if (!region_cap) h[i] = part_base[i] + old - run;
            else if (old + c <= region_cap) h[i] = i * region_cap + old - run;
            else {
                atomicOr(&skip[i >> 5], 1u << (i & 31));
                atomicOr(overflow, 1u);
            }
        }
        run += c;
    }
    __syncthreads();
    // 4. place the entries at their local sorted position
    if (live) {
        const uint32_t lmask = (1u << ss.lbits) - 1u;
#pragma unroll
        for (int w = 0; w < SORT1_MAXW; w++) {
            if (keyn[w] != 0xffffffffu) {
                const uint32_t key = keyn[w] & 0x7fffffffu;
                const uint32_t q = key >> ss.lbits;
                const uint32_t i = loff[q] + rk[w];
                stage[i] = make_uint2((key & lmask) | (q << 16),
                                      (uint32_t)((uint64_t)w * ss.srs_stride + ss.srs_offset + j) | (keyn[w] & 0x80000000u));
            }
        }
    }
    __syncthreads();
    // 5. copy out: consecutive lanes -> consecutive addresses inside each partition's run
    for (uint32_t i = t; i < count; i += 1024) {
        const uint2 v = stage[i];
        const uint32_t q = v.x >> 16;
        if (!((skip[q >> 5] >> (q & 31)) & 1u)) parted[h[q] + i] = make_uint2(v.x & 0xffffu, v.y);
    }
}

// R scalars per lane (R rounds of `spb` scalars each, digits of all of them in registers): ONE reservation per partition
// for the whole workgroup, so its runs are R times longer (at 2^22 points and 4096 partitions a round leaves 3.25 entries
// = 26 bytes per partition, and 32-byte sectors written for 26 bytes cost twice the bytes) and the global atomics R times
// fewer; the R x spb x nwin entries pass through the same stage in R slices of the partition-ordered sequence.
template <int R, int MAXW>
__global__ void __launch_bounds__(1024) k_sort_partition_staged_multi(const uint32_t* __restrict__ scalars, const SortShape ss,
                                                                 const WinLayout lay, uint32_t spb,
                                                                 const uint32_t* __restrict__ part_base,
                                                                 uint32_t* __restrict__ part_cursor,
                                                                 uint2* __restrict__ parted, uint32_t region_cap,
                                                                 uint32_t* __restrict__ overflow) {
    __shared__ uint32_t h[SORT_MAXPART];     // counts, then (global base - local offset) per partition
    __shared__ uint32_t skip[SORT_MAXPART / 32];
    __shared__ uint32_t loff[SORT_MAXPART];  // local exclusive offsets
    __shared__ uint32_t wsum[16];
    __shared__ uint2 stage[SORT1_STAGE];  // .x = key_low (<= 12 bits) | partition << 16
    const uint32_t t = threadIdx.x;
    const uint32_t npart = 1u << ss.hbits;
    for (uint32_t i = t; i < npart; i += 1024) h[i] = 0;
    if (t < SORT_MAXPART / 32) skip[t] = 0;
    __syncthreads();
    // 1. digits -> registers, rank inside (workgroup, partition)
    static_assert(R * SORT1_STAGE <= 65536, "ranks are kept in 16 bits");
    static_assert(R == 1 || R % 2 == 0, "scalars are taken two at a time");
    uint32_t keyn[R][MAXW], rkp[(R + 1) / 2][MAXW];   // ranks < R * SORT1_STAGE < 2^16: two per register
#pragma unroll
    for (int r = 0; r < R; r++)
#pragma unroll
        for (int w = 0; w < MAXW; w++) {
            keyn[r][w] = 0xffffffffu;
            if (!(r & 1)) rkp[r / 2][w] = 0;
        }
    constexpr int PAIR = R == 1 ? 1 : 2;   // loads in flight together (all R of them would cost 8 registers each)
#pragma unroll
    for (int r0 = 0; r0 < R; r0 += PAIR) {
        uint32_t sc[PAIR][8];
        bool live[PAIR], second[PAIR];
#pragma unroll
        for (int u = 0; u < PAIR; u++) {
            const uint64_t g = ((uint64_t)blockIdx.x * R + (r0 + u)) * spb + t;
            live[u] = t < spb && g < ss.total;
            second[u] = live[u] && g >= ss.n;
            if (live[u]) load_scalar(sc[u], second[u] ? ss.scalars2 : scalars, second[u] ? g - ss.n : g, second[u] ? ss.mont2 : ss.mont);
        }
#pragma unroll
        for (int u = 0; u < PAIR; u++) {
            if (live[u]) {
                const int r = r0 + u;
                const uint32_t set_bit = second[u] ? 1u << ss.keybits : 0u;
                uint32_t carry = 0, neg;
#pragma unroll
                for (int w = 0; w < MAXW; w++) {
                    if (w < lay.nwin) {
                        const uint32_t mag = signed_digit(sc[u], w, lay, carry, neg);
                        if (mag) {
                            const uint32_t key = (mag - 1) | set_bit;
                            keyn[r][w] = key | (neg << 31);
                            rkp[r / 2][w] |= lds_bump(h, key >> ss.lbits) << (16 * (r & 1));
                        }
                    }
                }
            }
        }
    }
    __syncthreads();
    // 2. exclusive scan of the counts
    const uint32_t per = (npart + 1023u) / 1024u;
    const uint32_t b0 = t * per, b1 = min(b0 + per, npart);
    uint32_t sum = 0;
    for (uint32_t i = b0; i < b1; i++) sum += h[i];
    uint32_t count;
    uint32_t run = block_scan_1024(sum, wsum, count) - sum;
    // 3. reserve the global run of every non-empty partition; h becomes (global position - local position)
    for (uint32_t i = b0; i < b1; i++) {
        const uint32_t c = h[i];
        loff[i] = run;
        if (c) {
            const uint32_t old = atomicAdd(&part_cursor[i], c);
            if (!region_cap) h[i] = part_base[i] + old - run;
            else if (old + c <= region_cap) h[i] = i * region_cap + old - run;
            else {
                atomicOr(&skip[i >> 5], 1u << (i & 31));
                atomicOr(overflow, 1u);
            }
        }
        run += c;
    }
    __syncthreads();
    const uint32_t lmask = (1u << ss.lbits) - 1u;
    for (uint32_t p0 = 0; p0 < count; p0 += SORT1_STAGE) {   // one slice when R == 1
        // 4. place the slice's entries at their local sorted position
#pragma unroll
        for (int r = 0; r < R; r++) {   // (a scalar that is not live has no digits)
            const uint64_t g = ((uint64_t)blockIdx.x * R + r) * spb + t;
            const uint64_t j = g >= ss.n ? g - ss.n : g;
#pragma unroll
            for (int w = 0; w < MAXW; w++) {
                if (keyn[r][w] != 0xffffffffu) {
                    const uint32_t key = keyn[r][w] & 0x7fffffffu;
                    const uint32_t q = key >> ss.lbits;
                    const uint32_t i = loff[q] + ((rkp[r / 2][w] >> (16 * (r & 1))) & 0xffffu) - p0;
                    if (R == 1 || i < SORT1_STAGE)   // (unsigned: positions of earlier slices wrap far above)
                        stage[i] = make_uint2((key & lmask) | (q << 16),
                                              (uint32_t)((uint64_t)w * ss.srs_stride + ss.srs_offset + j) | (keyn[r][w] & 0x80000000u));
                }
            }
        }
        __syncthreads();
        // 5. copy out: consecutive lanes -> consecutive addresses inside each partition's run
        const uint32_t cnt = min(count - p0, (uint32_t)SORT1_STAGE);
        for (uint32_t i = t; i < cnt; i += 1024) {
            const uint2 v = stage[i];
            const uint32_t q = v.x >> 16;
            if (!((skip[q >> 5] >> (q & 31)) & 1u)) parted[h[q] + p0 + i] = make_uint2(v.x & 0xffffu, v.y);
        }
        if (R > 1) __syncthreads();
    }
}

// End of k_sort_buckets when it carries the MSM's SortTail (all threads of the workgroup call it): the workgroup's longest
// carry run goes into *max_len, and the LAST workgroup to arrive (device-scope ticket) hands {max_len, overflow} to the
// host page and then sets its sequence word -- what a k_publish launch behind the sort did.
KZG_DEV void sort_tail_publish(const SortTail& tail, uint32_t npart, uint32_t blk_len, uint32_t* max_len,
                               const uint32_t* overflow, uint32_t* done) {
    __syncthreads();   // every thread's part of this workgroup is done
    if (threadIdx.x == 0) {
        // No fences here: a release fence is a write-back of the XCD's whole L2, which the sort has just filled with
        // dirty lines -- once per workgroup that cost more than the two launches this replaces (+17 us on a 2^12 row).
        // Only atomics have to be ordered, and a RETURNED device-scope atomic has been performed: the maximum is waited
        // for before the ticket is taken, the ticket's value decides who publishes, and the last workgroup reads both
        // words with device-scope atomic loads.
        if (blk_len > 1) {
            const uint32_t old = atomicMax(max_len, blk_len);
            asm volatile("s_waitcnt vmcnt(0)" ::"v"(old) : "memory");
        }
        if (atomicAdd(done, 1u) == npart - 1u) {
            atomicExch(done, 0u);   // the next sort finds the ticket counter clean
            const uint32_t ml = __hip_atomic_load(max_len, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const uint32_t ov = __hip_atomic_load(overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // the host page is uncached on the device side: the two words are written through, acknowledged
            // (vmcnt), and only then the sequence word follows them over the same ordered path
            __hip_atomic_store(&tail.pin_dst[0], ml, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(&tail.pin_dst[1], ov, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(tail.seq_word, tail.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}
#define SORT_LONG_RUN 256
#define SORT_STAGE 28672  // 112 KB: one 1024-thread workgroup per CU; fewer, larger partitions keep level 1's runs longer (A/B)
// one workgroup per partition: LDS histogram of the low bits -> bucket offsets -> scatter inside the partition
// region_cap != 0: the partition's entries are read from its fixed region q * region_cap (fast mode), written at the
// contiguous part_base[q] as always.  A raised overflow word means the fast attempt failed: every offset becomes 0, so the
// accumulate kernel that is already queued behind this one sees an empty MSM and exits (nothing stale is dereferenced).
__global__ void __launch_bounds__(1024) k_sort_buckets(const uint2* __restrict__ parted_in, const uint32_t* __restrict__ part_base,
                                                        int lbits, uint32_t* __restrict__ offsets,
                                                        uint32_t* __restrict__ sorted, uint32_t npart, uint32_t region_cap,
                                                        const uint32_t* __restrict__ overflow,
                                                        uint32_t* __restrict__ part_cursor, const SortTail tail,
                                                        uint32_t* __restrict__ max_len, uint32_t* __restrict__ done) {
    __shared__ uint32_t h[4096];
    __shared__ uint32_t blk_len;   // longest carry run among this partition's buckets (fused k_fold_maxlen)
    __shared__ uint32_t th[4096];   // per-tile histogram / cursors of the oversized-partition path
    __shared__ uint32_t longb[SORT_STAGE / SORT_LONG_RUN + 1], nlong, maxc;
    __shared__ uint32_t wsum[16];
    // a partition of up to SORT_STAGE entries is scattered inside LDS and leaves as whole lines (the 4-byte scatter
    // straight to HBM wrote 3.7x the bytes: lines left L2 partly filled); larger (long inputs, skewed scalars) partitions go through it tile by tile
    __shared__ uint32_t stage[SORT_STAGE];
    const uint32_t q = blockIdx.x, t = threadIdx.x;
    const uint32_t nb = 1u << lbits;
    const uint32_t lo = part_base[q], hi = part_base[q + 1];
    if (t == 0) part_cursor[q] = 0;   // the next sort finds the cursors clean
    if (region_cap && *overflow) {
        for (uint32_t i = t; i < nb; i += 1024) offsets[((uint64_t)q << lbits) + i] = 0;
        if (q == npart - 1 && t == 0) offsets[(uint64_t)npart << lbits] = 0;
        if (tail.buckets) sort_tail_publish(tail, npart, 0u, max_len, overflow, done);
        return;
    }
    // entry e of the output range [lo, hi) lies at parted[e] (exact mode) or at its region's start + (e - lo)
    const uint2* parted = region_cap ? parted_in + ((uint64_t)q * region_cap - lo) : parted_in;
    for (uint32_t i = t; i < nb; i += 1024) h[i] = 0;
    if (t == 0) maxc = blk_len = 0;
    __syncthreads();
    // four independent loads in flight per lane: the loop is otherwise a chain of dependent global-load latencies
    {
        uint32_t e = lo + t;
        for (; e + 3 * 1024 < hi; e += 4 * 1024) {
            const uint32_t k0 = parted[e].x, k1 = parted[e + 1024].x, k2 = parted[e + 2048].x, k3 = parted[e + 3072].x;
            lds_bump(h, k0); lds_bump(h, k1); lds_bump(h, k2); lds_bump(h, k3);
        }
        for (; e < hi; e += 1024) lds_bump(h, parted[e].x);
    }
    __syncthreads();
    // exclusive scan of h[0..nb): each lane owns nb/1024 (>= 1 when nb >= 1024) consecutive bins
    const uint32_t per = (nb + 1023u) / 1024u;
    const uint32_t b0 = t * per, b1 = min(b0 + per, nb);
    uint32_t s = 0;
    for (uint32_t i = b0; i < b1; i++) s += h[i];
    uint32_t all;
    uint32_t run = lo + block_scan_1024(s, wsum, all) - s, cmax = 0, lmax = 0;
    for (uint32_t i = b0; i < b1; i++) {
        const uint32_t c = h[i];
        offsets[((uint64_t)q << lbits) + i] = run;
        h[i] = run;  // becomes the scatter cursor
        if (tail.buckets) {   // what k_fold_maxlen does for this bucket
            if (!c) {
                uint4* z = reinterpret_cast<uint4*>(&tail.buckets[((uint64_t)q << lbits) + i]);
#pragma unroll
                for (int k = 0; k < 14; k++) z[k] = make_uint4(0u, 0u, 0u, 0u);
            } else {
                lmax = max(lmax, (run + c - 1u) / tail.chunk - run / tail.chunk);
            }
        }
        run += c;
        cmax = max(cmax, c);
    }
    if (lmax > 1) atomicMax(&blk_len, lmax);
    if (cmax * 4u > hi - lo) atomicMax(&maxc, cmax);   // only a dominant bucket matters (see below)
    if (q == npart - 1 && t == 1023) offsets[(uint64_t)npart << lbits] = hi;
    __syncthreads();
    if (hi - lo <= SORT_STAGE) {
        uint32_t e = lo + t;
        for (; e + 3 * 1024 < hi; e += 4 * 1024) {
            const uint2 v0 = parted[e], v1 = parted[e + 1024], v2 = parted[e + 2048], v3 = parted[e + 3072];
            stage[lds_bump(h, v0.x) - lo] = v0.y;
            stage[lds_bump(h, v1.x) - lo] = v1.y;
            stage[lds_bump(h, v2.x) - lo] = v2.y;
            stage[lds_bump(h, v3.x) - lo] = v3.y;
        }
        for (; e < hi; e += 1024) {
            const uint2 v = parted[e];
            stage[lds_bump(h, v.x) - lo] = v.y;
        }
        __syncthreads();
        for (uint32_t i = t; i < hi - lo; i += 1024) sorted[lo + i] = stage[i];
    } else if (maxc * 4u > hi - lo) {
        // oversized because ONE bucket dominates (all-equal / sparse scalars): its entries are consecutive in entry order
        // too, so the direct scatter is already a stream of whole lines -- and spares the per-tile passes
        uint32_t e = lo + t;
        for (; e + 3 * 1024 < hi; e += 4 * 1024) {
            const uint2 v0 = parted[e], v1 = parted[e + 1024], v2 = parted[e + 2048], v3 = parted[e + 3072];
            sorted[lds_bump(h, v0.x)] = v0.y;
            sorted[lds_bump(h, v1.x)] = v1.y;
            sorted[lds_bump(h, v2.x)] = v2.y;
            sorted[lds_bump(h, v3.x)] = v3.y;
        }
        for (; e < hi; e += 1024) {
            const uint2 v = parted[e];
            sorted[lds_bump(h, v.x)] = v.y;
        }
    } else {
        // Oversized partition (long inputs: 2^26 points leave 180 k entries per partition; or skewed scalars): TILES of
        // SORT_STAGE entries are ordered by bucket inside LDS and leave as one run per bucket and tile.  The entry-order
        // 4-byte scatter this replaces wrote 6.6x the bytes it stored at 2^26 (rocprofv3 WRITE_SIZE: 19.4 GB for
        // 2.95 GB; every store found its line already evicted) and was two thirds of the sort's time there.
        const uint32_t lane = t & 63u, wave = t >> 6;
        for (uint32_t ts = lo; ts < hi; ts += SORT_STAGE) {
            const uint32_t te = min(ts + (uint32_t)SORT_STAGE, hi);
            for (uint32_t i = t; i < nb; i += 1024) th[i] = 0;
            __syncthreads();
            {
                uint32_t e = ts + t;
                for (; e + 3 * 1024 < te; e += 4 * 1024) {
                    const uint32_t k0 = parted[e].x, k1 = parted[e + 1024].x, k2 = parted[e + 2048].x, k3 = parted[e + 3072].x;
                    lds_bump(th, k0); lds_bump(th, k1); lds_bump(th, k2); lds_bump(th, k3);
                }
                for (; e < te; e += 1024) lds_bump(th, parted[e].x);
            }
            __syncthreads();
            uint32_t ts_sum = 0;
            for (uint32_t i = b0; i < b1; i++) ts_sum += th[i];
            uint32_t tile_all;
            uint32_t trun = block_scan_1024(ts_sum, wsum, tile_all) - ts_sum;
            if (t == 0) nlong = 0;
            __syncthreads();
            for (uint32_t i = b0; i < b1; i++) {
                const uint32_t c = th[i];
                th[i] = trun;  // the tile-local cursor; after the placement it is the END of the bucket's run in `stage`
                trun += c;
                if (c > SORT_LONG_RUN) longb[atomicAdd(&nlong, 1u)] = i;   // at most SORT_STAGE / SORT_LONG_RUN of them
            }
            __syncthreads();
            {
                uint32_t e = ts + t;
                for (; e + 3 * 1024 < te; e += 4 * 1024) {
                    const uint2 v0 = parted[e], v1 = parted[e + 1024], v2 = parted[e + 2048], v3 = parted[e + 3072];
                    stage[lds_bump(th, v0.x)] = v0.y;
                    stage[lds_bump(th, v1.x)] = v1.y;
                    stage[lds_bump(th, v2.x)] = v2.y;
                    stage[lds_bump(th, v3.x)] = v3.y;
                }
                for (; e < te; e += 1024) {
                    const uint2 v = parted[e];
                    stage[lds_bump(th, v.x)] = v.y;
                }
            }
            __syncthreads();
            // one wave per bucket: its run [end of bucket i-1, end of bucket i) goes to the bucket's global cursor ...
            for (uint32_t i = wave; i < nb; i += 16) {
                const uint32_t s0 = i ? th[i - 1] : 0u, s1 = th[i];
                if (s1 - s0 > SORT_LONG_RUN) continue;
                const uint32_t g = h[i];
                for (uint32_t k = s0 + lane; k < s1; k += 64) sorted[g + (k - s0)] = stage[k];
                if (lane == 0) h[i] = g + (s1 - s0);
            }
            // ... except the few long runs (skewed scalars: a tile may be ONE bucket), which the whole workgroup copies
            const uint32_t nl = nlong;
            for (uint32_t j = 0; j < nl; j++) {
                const uint32_t i = longb[j];
                const uint32_t s0 = i ? th[i - 1] : 0u, s1 = th[i];
                const uint32_t g = h[i];
                for (uint32_t k = s0 + t; k < s1; k += 1024) sorted[g + (k - s0)] = stage[k];
            }
            __syncthreads();
            if (t < nl) {
                const uint32_t i = longb[t];
                h[i] += th[i] - (i ? th[i - 1] : 0u);
            }
            __syncthreads();
        }
    }
    if (tail.buckets) sort_tail_publish(tail, npart, blk_len, max_len, overflow, done);
}


// ------------------------------------------------------------------------------------------------ launchers
uint64_t msm_sort_region_cap(uint64_t entries, uint32_t npart) {
    // a partition's share of the entries is not uniform: windows one bit narrower than the widest put their digits in the
    // lower half of the key space, and the top window is cut short by the field size -- up to ~2x the mean
    return (entries / npart) * 9 / 4 + 2048;
}
// which of the sort's two modes can run: the fast one needs the staged partition kernel and 32-bit region addressing
bool msm_sort_fast_ok(const MsmShape& sh) {
    const uint64_t entries = (uint64_t)sh.n * sh.nbatch * sh.nwin;
    return sh.nwin <= SORT1_MAXW && entries * 9 / 4 + ((uint64_t)SORT_MAXPART << 11) < ((uint64_t)1 << 32);
}
static void sort_shape(const MsmShape& sh, const uint32_t* scalars2, int scalars_mont, int scalars2_mont, SortShape& ss) {
    const int setbits = sh.nbatch > 1 ? 1 : 0;
    const int keybits = sh.c - 1 + setbits;
    ss.n = sh.n; ss.total = sh.n << setbits; ss.srs_offset = sh.srs_offset; ss.srs_stride = sh.srs_stride;
    ss.mont = scalars_mont; ss.scalars2 = scalars2; ss.mont2 = scalars2_mont; ss.keybits = sh.c - 1;
    // 1024 partitions (level 2 runs one workgroup per partition), up to 4096 when that brings a partition down to what
    // level 2 can stage in LDS (SORT_STAGE entries; ~13 k on average at 2^20 / 1024, ~27 k at 2^22 / 2048 and 2^23 / 4096)
    const uint64_t entries = ss.total * (uint64_t)sh.nwin;
    int hbits = 10;
    // short inputs: fewer partitions (down to 64) as long as one holds < 4096 entries --
    // 1024 workgroups of 1024 threads for ~200 entries each were four rounds of launch overhead
    const int min_hbits = 6;
    while (hbits > min_hbits && (entries >> hbits) < 4096) hbits--;
    while (hbits < 12 && (entries >> hbits) > 24576) hbits++;
    if (hbits > keybits) hbits = keybits;
    if (keybits - hbits > 12) hbits = keybits - 12;  // level 2 histograms at most 4096 buckets
    ss.hbits = hbits;
    ss.lbits = keybits - hbits;
    ss.spb = ss.total >= (1u << 21) ? 4096u : 1024u;
}
uint64_t msm_sort_parted_entries(const MsmShape& sh, bool fast) {  // capacity of `parted`, in entries
    const uint64_t entries = (uint64_t)sh.n * sh.nbatch * sh.nwin;
    if (!fast) return entries;
    SortShape ss;
    sort_shape(sh, nullptr, 0, 0, ss);
    return msm_sort_region_cap(entries, 1u << ss.hbits) << ss.hbits;
}
// rounds per workgroup of the staged level-1 partition (A/B knob KZG_SORT_ROUNDS = 1 | 2): two from 2^20 scalars up
// (sort 0.146 -> 0.138 ms at 2^20, 0.72 -> 0.58 at 2^22, 2.89 -> 2.25 at 2^24; four rounds spill their digits and lose:
// `profiles/r03_ab_sort_two_rounds.log`); needs the digits of both scalars in registers (<= 16 windows)
static int sort_rounds(uint64_t total, int nwin) {
    static const int forced = [] {
        const char* e = getenv("KZG_SORT_ROUNDS");
        return e ? atoi(e) : 0;
    }();
    if (nwin > 16) return 1;
    if (forced == 1 || forced == 2) return forced;
    return total >= (1u << 20) ? 2 : 1;
}
static void launch_partition_staged(hipStream_t s, const uint32_t* scalars, const SortShape& ss, const WinLayout& lay,
                                    uint32_t spb2, const uint32_t* part_base, uint32_t* part_cursor, uint2* parted,
                                    uint32_t cap, uint32_t* overflow_word) {
    const int rounds = sort_rounds(ss.total, lay.nwin);
    if (rounds == 2)
        k_sort_partition_staged_multi<2, 16><<<nblk(ss.total, 2 * spb2), 1024, 0, s>>>(scalars, ss, lay, spb2, part_base,
                                                                                        part_cursor, parted, cap, overflow_word);
    else
        k_sort_partition_staged<<<nblk(ss.total, spb2), 1024, 0, s>>>(scalars, ss, lay, spb2, part_base, part_cursor,
                                                                     parted, cap, overflow_word);
}
// FAST mode (uniform-ish scalars: the common case): no count pass -- partition straight into fixed-capacity regions, scan
// the cursors, level 2 reads the regions.  If a region overflows (skewed scalars) *overflow_word is raised, the offsets
// come out all zero (the queued accumulate sees an empty MSM) and the caller reruns in EXACT mode: count pass, exact
// bases, no overflow possible.  Both leave the partition counts / cursors zero for the next sort.
void launch_msm_sort(hipStream_t s, const MsmShape& sh, const uint32_t* scalars, int scalars_mont,
                     const uint32_t* scalars2, int scalars2_mont, uint32_t* part_ws, bool part_ws_clean, uint2* parted,
                     uint32_t* offsets, uint32_t* sorted, uint32_t* max_len_word, bool fast, uint32_t* overflow_word,
                     const SortTail* tail) {
    SortTail tl{0u, nullptr, nullptr, nullptr, 0u};
    if (tail) tl = *tail;
    uint32_t* done = part_ws + 3 * SORT_MAXPART + 16;   // k_sort_buckets' ticket counter (zero between sorts)
    SortShape ss;
    sort_shape(sh, scalars2, scalars_mont, scalars2_mont, ss);
    const uint64_t entries = ss.total * (uint64_t)sh.nwin;
    const uint32_t npart = 1u << ss.hbits;
    uint32_t* part_count = part_ws;                      // [npart]
    uint32_t* part_base = part_ws + SORT_MAXPART;        // [npart + 1]
    uint32_t* part_cursor = part_ws + 2 * SORT_MAXPART + 8;
    if (!part_ws_clean) (void)hipMemsetAsync(part_ws, 0, 16384 * 4, s);   // afterwards the kernels keep counts / cursors zero
    uint32_t spb2 = (SORT1_STAGE / (uint32_t)sh.nwin) & ~63u;  // staged partition: one scalar per lane, <= SORT1_STAGE entries
    if (spb2 > 1024) spb2 = 1024;
    if (fast) {
        const uint32_t cap = (uint32_t)msm_sort_region_cap(entries, npart);
        launch_partition_staged(s, scalars, ss, sh.lay, spb2, part_base, part_cursor, parted, cap, overflow_word);
        k_sort_part_scan<<<1, 1024, 0, s>>>(part_cursor, npart, cap, part_base, max_len_word, nullptr);
        k_sort_buckets<<<npart, 1024, 0, s>>>(parted, part_base, ss.lbits, offsets, sorted, npart, cap, overflow_word,
                                              part_cursor, tl, max_len_word, done);
        return;
    }
    const uint32_t blocks = nblk(ss.total, ss.spb);
    if (ss.total > (1u << 18)) k_sort_count<4><<<nblk(ss.total, 4096), 1024, 0, s>>>(scalars, ss, sh.lay, part_count);
    else k_sort_count<1><<<nblk(ss.total, 1024), 1024, 0, s>>>(scalars, ss, sh.lay, part_count);
    k_sort_part_scan<<<1, 1024, 0, s>>>(part_count, npart, 0xffffffffu, part_base, max_len_word, overflow_word);
    if (sh.nwin <= SORT1_MAXW)
        launch_partition_staged(s, scalars, ss, sh.lay, spb2, part_base, part_cursor, parted, 0u, overflow_word);
    else
        k_sort_partition<<<blocks, 256, 0, s>>>(scalars, ss, sh.lay, part_base, part_cursor, parted);
    k_sort_buckets<<<npart, 1024, 0, s>>>(parted, part_base, ss.lbits, offsets, sorted, npart, 0u, overflow_word, part_cursor,
                                          tl, max_len_word, done);
}
