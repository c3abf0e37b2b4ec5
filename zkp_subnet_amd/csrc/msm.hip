// Pippenger G1 MSM kernels for gfx950 (see msm.hip.h for the pipeline).  No reference source exists for this
// path (reference neurons/miner.py:39,48 only calls the external prover); the algorithm is restated from the
// published bucket method and checked bit-for-bit against oracle/ in tests/test_gpu_*.py.
#include "msm.hip.h"
#include "fp_lp.hip.h"
#include <cstdlib>

#define NONE_KEY 0xffffffffu

// ------------------------------------------------------------------------------------------------ digits
KZG_DEV uint32_t limb_at(const uint32_t* s, int i) {
    uint32_t r = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) r = (i == k) ? s[k] : r;
    return r;
}
KZG_DEV uint32_t window_bits(const uint32_t* s, int lo, int c) {
    if (lo >= 256) return 0;
    int wi = lo >> 5, b = lo & 31;
    uint64_t v = limb_at(s, wi) | ((uint64_t)limb_at(s, wi + 1) << 32);  // limb_at(.., 8) == 0
    return (uint32_t)(v >> b) & ((1u << c) - 1u);
}
KZG_DEV void load_scalar(uint32_t* s, const uint32_t* scalars, uint64_t j, int mont) {
    const uint4* p = reinterpret_cast<const uint4*>(scalars + 8 * j);
    uint4 a = p[0], b = p[1];
    s[0] = a.x; s[1] = a.y; s[2] = a.z; s[3] = a.w;
    s[4] = b.x; s[5] = b.y; s[6] = b.z; s[7] = b.w;
    if (mont) {  // Montgomery-form row (the coefficients of a commitment): back to the canonical integer
        fr9_t v;
        fr9_from_words(v, s);
        fr9_from_mont(v, v);
        fr9_to_words(s, v);
    }
}
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

// The kernels after the accumulate (fold, bucket tree, final combination, encoding) are chains of dependent point
// operations with little work.  When they share the GPU with another lane's accumulate kernel (two requests in flight,
// or the two MSMs of a long row's commit+open) they must (a) FIT next to it -- k_msm_accumulate is held to 256
// registers (launch bound 2 waves/SIMD; 280 with AGPR spill space otherwise) so that a 250-register tail wave can be
// resident on the same SIMD -- and (b) win the issue arbitration against a wave that saturates the integer pipe.
KZG_DEV void tail_priority() { __builtin_amdgcn_s_setprio(3); }

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

// ------------------------------------------------------------------------------------------------ bucket tree
// Node at level L covers 2^L consecutive buckets and holds [P, T_0 .. T_{L-1}] (component-major arrays):
//   P = sum of its buckets, T_k = sum of its buckets whose index has bit k set.
// Merging left (bit L = 0) and right (bit L = 1): P = P_l + P_r, T_k = T_k_l + T_k_r, T_L = P_r.
// Every add of a level is independent, so the serial depth of the whole reduction is log2(B) point adds,
// and sum_k (k+1) B_k = P + sum_i 2^i T_i at the root.
// T_L of a merged node is just the P of its right child: it is never copied.  A level-L array stores P, T_0 .. T_{L-2}
// (L components, one at levels 0 and 1); T_{L-1} of node m is read where it already lies, at P[2m + 1] of the
// level-(L-1) array `prev` -- so the merge L -> L+1 does (L + 1) additions per output node and no copy:
//   k = 0 .. L-1 : out[k][m] = in[k][2m] + in[k][2m+1]          (P and the stored T's)
//   k = L (L>=1) : out[L][m] = prev[0][4m+1] + prev[0][4m+3]    (T_{L-1} of the two children)
KZG_DEV void tree_operands(const g1_xyzz_t* in, const g1_xyzz_t* prev, uint32_t n_in, int level, uint32_t k, uint32_t m,
                           const g1_xyzz_t*& pa, const g1_xyzz_t*& pb) {
    if (level >= 1 && k == (uint32_t)level) {
        pa = &prev[4 * (uint64_t)m + 1];
        pb = &prev[4 * (uint64_t)m + 3];
    } else {
        pa = &in[(uint64_t)k * n_in + 2 * m];
        pb = pa + 1;
    }
}
// (A/B knob: waves per SIMD the wide tree kernel and the carry heads are compiled for; 3 costs 62 spilled dwords per lane)
#ifndef KZG_TREE_MIN_WAVES
#define KZG_TREE_MIN_WAVES 2
#endif
__global__ void __launch_bounds__(256, KZG_TREE_MIN_WAVES) k_msm_tree_level(const g1_xyzz_t* __restrict__ in,
                                                         const g1_xyzz_t* __restrict__ prev,
                                                         g1_xyzz_t* __restrict__ out, uint32_t n_in, int level) {
    tail_priority();
    const uint32_t n_out = n_in >> 1;
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= n_out * (uint32_t)(level + 1)) return;
    const uint32_t k = gid / n_out, m = gid - k * n_out;
    const g1_xyzz_t *pa, *pb;
    tree_operands(in, prev, n_in, level, k, m, pa, pb);
    g1_xyzz_t a, b, r;
    load_xyzz(a, pa);
    load_xyzz(b, pb);
    g1_add<true>(r, a, b);  // wide levels are throughput-bound: inlined products (-4 %, same-box A/B)
    // A wave's 64 results are one contiguous 14-KB run of `out`, but a lane's own 224 bytes make every store instruction
    // touch 64 different lines.  They go through a wave-private slice of LDS (240-byte slots: conflict-free b128 writes)
    // and leave as 14 fully coalesced rows: -9 % on the two-round levels (`profiles/r03_exp_tree_coalescing.log`; the same
    // treatment of the operand loads measured 0).  No barrier: a wave's LDS operations execute in order.
    if ((n_out & 63u) == 0) {   // wave-uniform: all 64 lanes are live and share k
        __shared__ uint4 stage[4][64 * 15];
        const uint32_t lane = threadIdx.x & 63u;
        uint4* mine = stage[threadIdx.x >> 6];
        uint32_t t[56];
        const fp_t* f[4] = {&r.x, &r.y, &r.zz, &r.zzz};
#pragma unroll
        for (int c = 0; c < 4; c++)
#pragma unroll
            for (int i = 0; i < 14; i++) t[14 * c + i] = f[c]->l[i];
#pragma unroll
        for (int i = 0; i < 14; i++) mine[lane * 15 + i] = make_uint4(t[4 * i], t[4 * i + 1], t[4 * i + 2], t[4 * i + 3]);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        uint4* wb = reinterpret_cast<uint4*>(&out[(uint64_t)k * n_out + (m - lane)]);
#pragma unroll
        for (int i = 0; i < 14; i++) {
            const uint32_t g = (uint32_t)i * 64u + lane;   // 16-byte piece of the run: piece g % 14 of result g / 14
            const uint32_t pt = g / 14u;
            wb[g] = mine[pt * 15 + (g - pt * 14u)];
        }
        return;
    }
    store_xyzz(&out[(uint64_t)k * n_out + m], r);
}

// ------------------------------------------------------------------------------------------------ cooperative ops
// The tail of the MSM (fold, bucket tree, final) is a chain of DEPENDENT point additions; one lane needs ~16 us per
// addition (14 Fp products back to back), so these phases are pure latency.  A cooperative addition spreads ONE
// addition over the 4 waves of a 256-thread workgroup: lane l of every wave works on operation l, wave w computes
// the w-th of the independent Fp products of each stage, results cross waves through LDS.  Dependent depth: 4
// products + 4 barriers instead of 14 products.  Exceptional operands (an infinity, or equal x: P == 0) are
// detected identically by every wave (same inputs) and redone by wave 0 with the ordinary g1_add.
// All 256 threads of the workgroup must call these functions (they contain __syncthreads).
struct CoopLds {
    fp_t v[12][64];  // U1 U2 S1 S2 | PP RR ZZ12 ZZZ12 | PPP Q t u
};
KZG_DEV void lds_put(fp_t* dst, const fp_t& v) {
#pragma unroll
    for (int i = 0; i < 14; i++) dst->l[i] = v.l[i];
}
KZG_DEV void lds_get(fp_t& v, const fp_t* src) {
#pragma unroll
    for (int i = 0; i < 14; i++) v.l[i] = src->l[i];
}
// *out = *p + *q for the operation of this lane (lane = threadIdx.x & 63; all four waves pass the same pointers).
// Operands are read field by field from memory (global or LDS) so that no wave holds both points in registers;
// wave 0 writes the complete result.  `active` = this lane has an operation at all (inactive lanes only keep the
// barriers company).  out may alias p or q.
KZG_DEV void load_fp(fp_t& v, const fp_t* src) {
#pragma unroll
    for (int i = 0; i < 14; i++) v.l[i] = src->l[i];
}
KZG_DEV void coop_add(CoopLds& sm, g1_xyzz_t* out, const g1_xyzz_t* p, const g1_xyzz_t* q, bool active) {
    const uint32_t w = threadIdx.x >> 6, l = threadIdx.x & 63;
    fp_t a, b, t;
    bool pinf = true, qinf = true;
    fp_zero(a);
    fp_zero(b);
    if (active) {
        fp_t z1, z2;
        load_fp(z1, &p->zz);
        load_fp(z2, &q->zz);
        pinf = fp_limbs_zero(z1);
        qinf = fp_limbs_zero(z2);
        // stage 1 operands: U1 = X1 ZZ2, U2 = X2 ZZ1, S1 = Y1 ZZZ2, S2 = Y2 ZZZ1
        if (w == 0) { load_fp(a, &p->x); b = z2; }
        else if (w == 1) { load_fp(a, &q->x); b = z1; }
        else if (w == 2) { load_fp(a, &p->y); load_fp(b, &q->zzz); }
        else { load_fp(a, &q->y); load_fp(b, &p->zzz); }
    }
    fp_mul(t, a, b);
    lds_put(&sm.v[w][l], t);
    __syncthreads();
    // stage 2: PP = (U2 - U1)^2 (w0), RR = (S2 - S1)^2 (w1), ZZ12 = ZZ1 ZZ2 (w2), ZZZ12 = ZZZ1 ZZZ2 (w3)
    if (w < 2) {
        fp_t x, y;
        lds_get(x, &sm.v[w == 0 ? 0 : 2][l]);
        lds_get(y, &sm.v[w == 0 ? 1 : 3][l]);
        fp_sub4(a, y, x);
        b = a;
    } else if (active) {
        load_fp(a, w == 2 ? &p->zz : &p->zzz);
        load_fp(b, w == 2 ? &q->zz : &q->zzz);
    }
    fp_mul(t, a, b);
    lds_put(&sm.v[4 + w][l], t);
    __syncthreads();
    fp_t PP;
    lds_get(PP, &sm.v[4][l]);
    const bool special = !active || pinf || qinf || fp_is_zero_n(PP);
    // stage 3: PPP = P PP (w0; a still holds P), Q = U1 PP (w1), zz3 = ZZ12 PP (w2); w3 repeats w2's product, unused
    if (w == 1) lds_get(a, &sm.v[0][l]);
    else if (w >= 2) lds_get(a, &sm.v[6][l]);
    fp_mul(t, a, PP);
    if (w < 2) lds_put(&sm.v[8 + w][l], t);
    if (w == 2) lds_put(&sm.v[10][l], t);  // zz3
    __syncthreads();
    // stage 4: x3 = RR - PPP - 2Q ; t = R (Q - x3) (w0) ; u = S1 PPP (w1) ; zzz3 = ZZZ12 PPP (w2, w3 idles alike)
    fp_t PPP, x3;
    lds_get(PPP, &sm.v[8][l]);
    if (w == 0) {
        fp_t Q, RR, s1, s2;
        lds_get(Q, &sm.v[9][l]);
        lds_get(RR, &sm.v[5][l]);
        fp_sub4(t, RR, PPP); fp_sub4(t, t, Q); fp_sub4(t, t, Q);
        fp_norm(x3, t);
        fp_sub16(b, Q, x3);
        lds_get(s1, &sm.v[2][l]);
        lds_get(s2, &sm.v[3][l]);
        fp_sub4(a, s2, s1);  // R
    } else if (w < 3) {
        lds_get(a, &sm.v[w == 1 ? 2 : 7][l]);
        b = PPP;
    } else {
        a = PPP;  // wave 3 has no product in this stage
        b = PPP;
    }
    fp_mul(t, a, b);
    if (w == 1) lds_put(&sm.v[11][l], t);  // u = S1 PPP
    if (w == 2) lds_put(&sm.v[7][l], t);   // zzz3 (ZZZ12 is dead now)
    __syncthreads();
    if (w == 0 && active) {
        if (special) {  // rare: an infinity or equal x coordinates -> the ordinary addition, one lane
            g1_xyzz_t pa, qa, r;
            load_xyzz(pa, p);
            load_xyzz(qa, q);
            g1_add(r, pa, qa);
            store_xyzz(out, r);
        } else {
            fp_t u, y3;
            lds_get(u, &sm.v[11][l]);
            fp_sub4(t, t, u);
            fp_norm(y3, t);
            g1_xyzz_t r;
            r.x = x3; r.y = y3;
            lds_get(r.zz, &sm.v[10][l]);
            lds_get(r.zzz, &sm.v[7][l]);
            store_xyzz(out, r);
        }
    }
    __syncthreads();  // result visible to the workgroup; sm free for the next cooperative call
}

// *out = 2 * *p, cooperative (EFD dbl-2008-s-1 spread over the waves: dependent depth 3 products instead of 9)
KZG_DEV void coop_dbl(CoopLds& sm, g1_xyzz_t* out, const g1_xyzz_t* p, bool active) {
    const uint32_t w = threadIdx.x >> 6, l = threadIdx.x & 63;
    fp_t a, b, t, X, Y;
    fp_zero(X);
    fp_zero(Y);
    bool inf = true;
    if (active) {
        fp_t z;
        load_fp(z, &p->zz);
        inf = fp_limbs_zero(z);
        load_fp(X, &p->x);
        load_fp(Y, &p->y);
    }
    const bool live = active && !inf;
    fp_t U;
    fp_dbl(U, Y);
    // stage 1: V = U^2 (w0), XX = X^2 (w1)
    a = (w == 0) ? U : X;
    fp_sqr(t, a);
    if (w < 2) lds_put(&sm.v[w][l], t);
    __syncthreads();
    // stage 2: W = U V (w0), S = X V (w1), MM = (3 XX)^2 (w2), zz3 = V ZZ (w3)
    fp_t V, M;
    lds_get(V, &sm.v[0][l]);
    {
        fp_t xx;
        lds_get(xx, &sm.v[1][l]);
        fp_add(M, xx, xx);
        fp_add(M, M, xx);
    }
    if (w == 0) { a = U; b = V; }
    else if (w == 1) { a = X; b = V; }
    else if (w == 2) { a = M; b = M; }
    else { a = V; fp_zero(b); if (active) load_fp(b, &p->zz); }
    fp_mul(t, a, b);
    lds_put(&sm.v[2 + w][l], t);  // W S MM zz3
    __syncthreads();
    // stage 3: x3 = MM - 2S ; t = M (S - x3) (w0) ; u = W Y (w1) ; zzz3 = W ZZZ (w2)
    fp_t x3;
    if (w == 0) {
        fp_t S, MM;
        lds_get(S, &sm.v[3][l]);
        lds_get(MM, &sm.v[4][l]);
        fp_sub4(t, MM, S); fp_sub4(t, t, S);
        fp_norm(x3, t);
        a = M;
        fp_sub16(b, S, x3);
    } else {
        lds_get(a, &sm.v[2][l]);  // W
        if (w == 1) b = Y;
        else { fp_zero(b); if (active) load_fp(b, &p->zzz); }
    }
    fp_mul(t, a, b);
    if (w == 1) lds_put(&sm.v[6][l], t);
    if (w == 2) lds_put(&sm.v[7][l], t);
    __syncthreads();
    if (w == 0 && live) {
        fp_t u, y3;
        lds_get(u, &sm.v[6][l]);
        fp_sub4(t, t, u);
        fp_norm(y3, t);
        g1_xyzz_t r;
        r.x = x3; r.y = y3;
        lds_get(r.zz, &sm.v[5][l]);
        lds_get(r.zzz, &sm.v[7][l]);
        store_xyzz(out, r);
    } else if (w == 0 && active && out != p) {
        g1_xyzz_t r;
        g1_set_inf(r);
        store_xyzz(out, r);
    }
    __syncthreads();
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
__global__ void __launch_bounds__(256) k_msm_tree_level_coop(const g1_xyzz_t* __restrict__ in,
                                                              const g1_xyzz_t* __restrict__ prev,
                                                              g1_xyzz_t* __restrict__ out, uint32_t n_in, int level) {
    tail_priority();
    __shared__ CoopLds sm;
    const uint32_t n_out = n_in >> 1;
    const uint32_t gid = blockIdx.x * 64 + (threadIdx.x & 63);
    const bool active = gid < n_out * (uint32_t)(level + 1);
    const uint32_t k = active ? gid / n_out : 0, m = active ? gid - k * n_out : 0;
    const g1_xyzz_t *pa, *pb;
    tree_operands(in, prev, n_in, level, k, m, pa, pb);
    coop_add(sm, &out[(uint64_t)k * n_out + m], pa, pb, active);
}

// ---- lane-parallel forms (fp_lp.hip.h): ONE wave per point operation, for phases with at most LP_MAX_OPS operations.
// Workgroups are single waves: every wave is a dependent instruction chain that owns its SIMD's issue port, so the
// operations must spread over as many SIMDs as possible (4 waves of one 256-thread workgroup would share one CU).
__global__ void __launch_bounds__(64) k_msm_tree_level_lp(const g1_xyzz_t* __restrict__ in,
                                                           const g1_xyzz_t* __restrict__ prev,
                                                           g1_xyzz_t* __restrict__ out, uint32_t n_in, int level) {
    tail_priority();
    __shared__ LpScratch sm;
    const LpLane k = lp_lane();
    const uint32_t n_out = n_in >> 1;
    const uint32_t gid = blockIdx.x;
    const uint32_t c = gid / n_out, m = gid - c * n_out;
    const g1_xyzz_t *pa, *pb;
    tree_operands(in, prev, n_in, level, c, m, pa, pb);
    lp_add(sm, &out[(uint64_t)c * n_out + m], pa, pb, k);
}
// TWO consecutive narrow levels in one launch (a short row's tail is a chain of ~6-us launches of which the addition is
// 2 us): a 2-wave workgroup per (component c, node j of level + 2).  Wave w merges the pair (2j + w) of level `level`
// into LDS, a barrier, wave 0 merges the two results.  Component level + 1 is born at the second merge (T_{level} of a
// level + 2 node = P[4j + 1] + P[4j + 3] of `in`): one addition by wave 0.  Only the P array of the skipped level is kept
// (mid_p, n_in / 2 nodes): the merge after the next reads its odd entries.
__global__ void __launch_bounds__(128) k_msm_tree_level2_lp(const g1_xyzz_t* __restrict__ in,
                                                             const g1_xyzz_t* __restrict__ prev,
                                                             g1_xyzz_t* __restrict__ mid_p, g1_xyzz_t* __restrict__ out,
                                                             uint32_t n_in, int level) {
    tail_priority();
    __shared__ LpScratch sm[2];
    __shared__ __align__(16) g1_xyzz_t mid[2];
    const LpLane k = lp_lane();
    const uint32_t w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t n2 = n_in >> 2;
    const uint32_t c = blockIdx.x / n2, j = blockIdx.x - c * n2;   // c in [0, level + 1]
    const bool late = c == (uint32_t)level + 1u;
    g1_xyzz_t* dst = &out[(uint64_t)c * n2 + j];
    if (late) {
        if (w == 0) lp_add(sm[0], dst, &in[4 * (uint64_t)j + 1], &in[4 * (uint64_t)j + 3], k);
    } else {
        const g1_xyzz_t *pa, *pb;
        tree_operands(in, prev, n_in, level, c, 2 * j + w, pa, pb);
        lp_add(sm[w], &mid[w], pa, pb, k);
    }
    __syncthreads();
    if (late) return;
    if (c == 0 && threadIdx.x < 112)
        reinterpret_cast<uint32_t*>(&mid_p[2 * (uint64_t)j])[threadIdx.x] = reinterpret_cast<const uint32_t*>(mid)[threadIdx.x];
    if (w == 0) lp_add(sm[0], dst, &mid[0], &mid[1], k);
}
// P + sum_i 2^i T_i for `nodes` roots, two launches.  (1) k_msm_final_dbl_lp: one wave per (root m, component l):
// pts[m][l] = 2^l T_l (l doublings; the P entry, index nbits, is copied) -- the chains run on different SIMDs, the
// longest (nbits - 1 doublings, ~1.4 us each) sets the time.  (2) k_msm_final_sum_lp: tree sum of the nbits + 1 points of
// a root, one wave per addition, one 512-thread workgroup per root.
#define FINAL_PTS 32
__global__ void __launch_bounds__(64) k_msm_final_dbl_lp(const g1_xyzz_t* __restrict__ node,
                                                          const g1_xyzz_t* __restrict__ prev, int nbits, int nodes,
                                                          g1_xyzz_t* __restrict__ pts) {
    tail_priority();
    __shared__ LpScratch sm;
    __shared__ g1_xyzz_t v;
    const LpLane k = lp_lane();
    const int l = (int)(blockIdx.x % (uint32_t)(nbits + 1));
    const uint32_t m = blockIdx.x / (uint32_t)(nbits + 1);
    const g1_xyzz_t* src = l < nbits - 1 ? &node[(uint64_t)(1 + l) * nodes + m] : l == nbits - 1 ? &prev[2 * m + 1] : &node[m];
    if (threadIdx.x < 56) reinterpret_cast<uint32_t*>(&v)[threadIdx.x] = reinterpret_cast<const uint32_t*>(src)[threadIdx.x];
    __syncthreads();
    if (l < nbits)
        for (int s = 0; s < l; s++) lp_dbl(sm, &v, &v, k);
    __syncthreads();
    if (threadIdx.x < 56)
        reinterpret_cast<uint32_t*>(&pts[(uint64_t)m * FINAL_PTS + l])[threadIdx.x] = reinterpret_cast<const uint32_t*>(&v)[threadIdx.x];
}
__global__ void __launch_bounds__(512) k_msm_final_sum_lp(const g1_xyzz_t* __restrict__ pts_g, int nbits,
                                                           g1_xyzz_t* __restrict__ out) {
    tail_priority();
    __shared__ LpScratch sm[8];
    __shared__ g1_xyzz_t pts[FINAL_PTS];
    const LpLane k = lp_lane();
    const int w = (int)__builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t m = blockIdx.x;
    const int count = nbits + 1;  // <= 23
    for (uint32_t i = threadIdx.x; i < (uint32_t)count * 56; i += 512)
        reinterpret_cast<uint32_t*>(pts)[i] = reinterpret_cast<const uint32_t*>(pts_g + (uint64_t)m * FINAL_PTS)[i];
    __syncthreads();
    for (int d = 16; d >= 1; d >>= 1) {
        for (int l = w; l < d; l += 8)
            if (l + d < count) lp_add(sm[w], &pts[l], &pts[l], &pts[l + d], k);
        __syncthreads();
    }
    if (threadIdx.x < 56) reinterpret_cast<uint32_t*>(&out[m])[threadIdx.x] = reinterpret_cast<const uint32_t*>(&pts[0])[threadIdx.x];
}

// sum of the values held by lanes [0, nthreads) (power of two <= blockDim), result in every lane's `mine`
KZG_DEV void lds_tree_sum(g1_xyzz_t* sm, g1_xyzz_t& mine, uint32_t tid, uint32_t nthreads) {
    store_xyzz(&sm[tid], mine);
    __syncthreads();
    for (uint32_t d = nthreads >> 1; d >= 1; d >>= 1) {
        if (tid < d) {
            g1_xyzz_t a, b, r;
            load_xyzz(a, &sm[tid]);
            load_xyzz(b, &sm[tid + d]);
            g1_add(r, a, b);
            store_xyzz(&sm[tid], r);
        }
        __syncthreads();
    }
    load_xyzz(mine, &sm[0]);
}

// node = [P, T_0..T_{nbits-1}]; operation i doubles T_i i times, then a 32- or 64-wide tree sum; every doubling and
// addition is cooperative (one 256-thread workgroup = 64 operations).
// node: the roots (level nbits: P, T_0 .. T_{nbits-2} stored); prev: the level below, whose P[2m+1] is T_{nbits-1}
__global__ void __launch_bounds__(256) k_msm_final(const g1_xyzz_t* __restrict__ node,
                                                    const g1_xyzz_t* __restrict__ prev, int nbits, int nodes,
                                                    g1_xyzz_t* __restrict__ out) {
    tail_priority();
    __shared__ CoopLds sm;
    __shared__ g1_xyzz_t pts[64];
    const uint32_t l = threadIdx.x & 63;
    const uint32_t m = blockIdx.x;  // root of this workgroup
    if (threadIdx.x < 64) {
        g1_xyzz_t v;
        g1_set_inf(v);
        if ((int)l < nbits - 1) load_xyzz(v, &node[(uint64_t)(1 + l) * nodes + m]);
        else if ((int)l == nbits - 1) load_xyzz(v, &prev[2 * m + 1]);
        else if ((int)l == nbits) load_xyzz(v, &node[m]);
        store_xyzz(&pts[l], v);
    }
    __syncthreads();
    for (int step = 0; step < nbits - 1; step++)  // operation l needs l doublings
        coop_dbl(sm, &pts[l], &pts[l], (int)l < nbits && step < (int)l);
    for (uint32_t d = (nbits + 1 <= 32 ? 16 : 32); d >= 1; d >>= 1)
        coop_add(sm, &pts[l], &pts[l], &pts[l + d < 64 ? l + d : l], l < d);
    if (threadIdx.x == 0) {
        g1_xyzz_t v;
        load_xyzz(v, &pts[0]);
        store_xyzz(&out[m], v);
    }
}

__global__ void __launch_bounds__(64) k_g1_sum(const g1_xyzz_t* __restrict__ in, uint32_t count,
                                                g1_xyzz_t* __restrict__ out) {
    tail_priority();
    __shared__ g1_xyzz_t sm[64];
    const uint32_t tid = threadIdx.x;
    g1_xyzz_t acc, q, r;
    g1_set_inf(acc);
    for (uint32_t i = tid; i < count; i += 64) {
        load_xyzz(q, &in[i]);
        g1_add(r, acc, q);
        acc = r;
    }
    uint32_t lanes = 1;                      // tree only over the lanes that can hold a term
    while (lanes < count && lanes < 64) lanes <<= 1;
    lds_tree_sum(sm, acc, tid, lanes);
    if (tid == 0) store_xyzz(out, acc);
}

// sum of up to 32 XYZZ points by lane-parallel additions (one wave per addition, log2 depth of ~2 us steps): the sum of
// the all_gathered partials of an SRS-sharded MSM is on every step's critical path (8 ranks: 3 levels instead of the
// 4 x 16 us of dependent one-lane additions of k_g1_sum)
__global__ void __launch_bounds__(512) k_g1_sum_lp(const g1_xyzz_t* __restrict__ in, uint32_t count,
                                                    g1_xyzz_t* __restrict__ out) {
    tail_priority();
    __shared__ LpScratch sm[8];
    __shared__ g1_xyzz_t pts[32];
    const LpLane k = lp_lane();
    const int w = (int)__builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (uint32_t i = threadIdx.x; i < 32 * 56; i += 512)
        reinterpret_cast<uint32_t*>(pts)[i] = i < count * 56 ? reinterpret_cast<const uint32_t*>(in)[i] : 0u;   // zeros = infinity
    __syncthreads();
    for (int d = 16; d >= 1; d >>= 1) {
        if ((uint32_t)d < count)
            for (int l = w; l < d; l += 8)
                if ((uint32_t)(l + d) < count) lp_add(sm[w], &pts[l], &pts[l], &pts[l + d], k);
        __syncthreads();
    }
    if (threadIdx.x < 56) reinterpret_cast<uint32_t*>(out)[threadIdx.x] = reinterpret_cast<const uint32_t*>(&pts[0])[threadIdx.x];
}
// sum of `count` affine points (table-row format: the output of k_srs_from_c48): Pianist's master aggregation
// sum_i commit_i over the worker rows (reference neurons/validator.py:196-198, README.md:38)
__global__ void __launch_bounds__(64) k_g1_sum_affine(const g1_affine_t* __restrict__ in, uint32_t count,
                                                       g1_xyzz_t* __restrict__ out) {
    tail_priority();
    __shared__ g1_xyzz_t sm[64];
    const uint32_t tid = threadIdx.x;
    g1_xyzz_t acc;
    g1_set_inf(acc);
    for (uint32_t i = tid; i < count; i += 64) {
        g1_aff28 q;
        g1_load_aff(q, &in[i]);
        g1_madd_checked(acc, q);
    }
    uint32_t lanes = 1;
    while (lanes < count && lanes < 64) lanes <<= 1;
    lds_tree_sum(sm, acc, tid, lanes);
    if (tid == 0) store_xyzz(out, acc);
}

// ---- G1 membership of untrusted points (the miners' commitments entering the master aggregation): on the curve is not
// enough -- E(Fp) has cofactor h = (z-1)^2/3 ~ 2^126.  With sigma(x, y) = (beta x, y), beta a primitive cube root of unity,
// the endomorphism sigma + z^2 (z = |BLS parameter| = 0xd201000000010000) has degree N(sigma + z^2) = z^4 - z^2 + 1 = r,
// is separable, and kills G1 (sigma acts there as -z^2 mod r): its kernel IS G1.  So P is in G1 exactly when
// [z^2] P == -sigma(P), two multiplications by the 64-bit, weight-6 z: 126 doublings + 10 additions instead of the
// 255 + ~127 of [r]P.  One wave per point (lane-parallel point operations, ~1.5 us each): ~0.2 ms for any count up to
// the number of SIMDs.  beta below satisfies sigma(G) = -[z^2]G for the generator (checked in tests/test_oracle.py).
FP28_TABLE(g1_beta_mont, 0x0a75929au, 0x0681b798u, 0x022a3e9du, 0x0abc02bfu, 0x04e5bb45u, 0x055e6e7eu, 0x04814117u,
           0x06d04f1bu, 0x0ae3387du, 0x054acb0cu, 0x00a4c74bu, 0x056138b5u, 0x0b64e066u, 0x000076f2u)
#define BLS_Z 0xd201000000010000ull
__global__ void __launch_bounds__(64) k_g1_subgroup_check_lp(const g1_affine_t* __restrict__ in, uint32_t count,
                                                              uint32_t* __restrict__ bad) {
    tail_priority();
    __shared__ LpScratch sm;
    __shared__ g1_xyzz_t base, acc;
    const LpLane k = lp_lane();
    const uint32_t j = blockIdx.x;
    if (j >= count) return;
    if (threadIdx.x == 0) {
        g1_aff28 a;
        g1_load_aff(a, &in[j]);
        g1_xyzz_t t;
        g1_from_aff(t, a);
        store_xyzz(&base, t);
        store_xyzz(&acc, t);
    }
    __syncthreads();
    for (int pass = 0; pass < 2; pass++) {           // acc <- [z] base, most significant bit first; then base <- acc
        for (int b = 62; b >= 0; b--) {
            lp_dbl(sm, &acc, &acc, k);
            __syncthreads();
            if ((BLS_Z >> b) & 1ull) {
                lp_add(sm, &acc, &acc, &base, k);
                __syncthreads();
            }
        }
        if (threadIdx.x < 56) reinterpret_cast<uint32_t*>(&base)[threadIdx.x] = reinterpret_cast<const uint32_t*>(&acc)[threadIdx.x];
        __syncthreads();
    }
    if (threadIdx.x != 0) return;
    g1_aff28 a;
    g1_load_aff(a, &in[j]);
    if (g1_aff_is_inf(a)) return;                    // the identity is a member
    g1_xyzz_t q;
    load_xyzz(q, &acc);
    if (g1_is_inf(q)) { atomicOr(bad, 4u); return; } // order divides z^2: not in G1
    // [z^2]P = (X/ZZ, Y/ZZZ) == (beta x, -y)  <=>  beta x ZZ - X == 0  and  y ZZZ + Y == 0
    fp_t beta, one, t, d, chk;
#pragma unroll
    for (int i = 0; i < 14; i++) beta.l[i] = g1_beta_mont(i);
    fp_one(one);
    fp_mul(t, beta, a.x);
    fp_mul(t, t, q.zz);
    fp_sub16(d, t, q.x);                             // X: normalised, < 14p
    fp_mul(chk, d, one);
    bool ok = fp_is_zero_n(chk);
    fp_mul(t, a.y, q.zzz);
    fp_add(d, t, q.y);                               // Y: normalised, < 6p
    fp_mul(chk, d, one);
    ok &= fp_is_zero_n(chk);
    if (!ok) atomicOr(bad, 4u);
}

// the same test, one LANE per point, for whole setup files (throughput form: 2^24 points are ~2 x 10^9 point operations,
// ~0.3 s -- the lane-parallel form above is for a few hundred untrusted commitments where latency counts)
__global__ void __launch_bounds__(256) k_g1_subgroup_check(const g1_affine_t* __restrict__ in, uint64_t n,
                                                            uint32_t* __restrict__ bad) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    g1_aff28 a;
    g1_load_aff(a, &in[j]);
    if (g1_aff_is_inf(a)) return;
    g1_xyzz_t acc, t;
    g1_from_aff(acc, a);
    for (int b = 62; b >= 0; b--) {                    // acc <- [z] P : the base is affine, mixed additions
        g1_dbl(t, acc);
        acc = t;
        if ((BLS_Z >> b) & 1ull) g1_madd_checked(acc, a);
    }
    g1_xyzz_t base = acc, q = acc;
    for (int b = 62; b >= 0; b--) {                    // q <- [z] ([z] P)
        g1_dbl(t, q);
        q = t;
        if ((BLS_Z >> b) & 1ull) {
            g1_add(t, q, base);
            q = t;
        }
    }
    if (g1_is_inf(q)) { atomicOr(bad, 4u); return; }
    fp_t beta, one, u, d, chk;
#pragma unroll
    for (int i = 0; i < 14; i++) beta.l[i] = g1_beta_mont(i);
    fp_one(one);
    fp_mul(u, beta, a.x);
    fp_mul(u, u, q.zz);
    fp_sub16(d, u, q.x);
    fp_mul(chk, d, one);
    bool ok = fp_is_zero_n(chk);
    fp_mul(u, a.y, q.zzz);
    fp_add(d, u, q.y);
    fp_mul(chk, d, one);
    ok &= fp_is_zero_n(chk);
    if (!ok) atomicOr(bad, 4u);
}

__global__ void __launch_bounds__(64) k_g1_compress(const g1_xyzz_t* __restrict__ in, uint8_t* __restrict__ out48) {
    tail_priority();
    if (threadIdx.x != 0) return;
    g1_xyzz_t p;
    load_xyzz(p, in);
    g1_aff28 a;
    g1_to_aff(a, p);
    g1_compress(out48, a);
}
// two points, ONE inversion (Montgomery's trick): commit + open share the single-lane inversion latency
__global__ void __launch_bounds__(64) k_g1_compress_pair(const g1_xyzz_t* __restrict__ in0,
                                                          const g1_xyzz_t* __restrict__ in1,
                                                          uint8_t* __restrict__ out0, uint8_t* __restrict__ out1) {
    tail_priority();
    if (threadIdx.x != 0) return;
    g1_xyzz_t p0, p1;
    load_xyzz(p0, in0);
    load_xyzz(p1, in1);
    const bool i0 = g1_is_inf(p0), i1 = g1_is_inf(p1);
    fp_t d0, d1, one, t, inv, w0, w1;
    fp_one(one);
    fp_mul(d0, p0.zz, p0.zzz);
    fp_mul(d1, p1.zz, p1.zzz);
    if (i0) d0 = one;
    if (i1) d1 = one;
    fp_mul(t, d0, d1);
    fp_inv(inv, t);
    fp_mul(w0, inv, d1);  // 1 / d0
    fp_mul(w1, inv, d0);  // 1 / d1
    g1_aff28 a0, a1;
    fp_zero(a0.x); fp_zero(a0.y); fp_zero(a1.x); fp_zero(a1.y);
    if (!i0) {
        fp_mul(t, w0, p0.zzz); fp_mul(t, p0.x, t); fp_canon(a0.x, t);
        fp_mul(t, w0, p0.zz); fp_mul(t, p0.y, t); fp_canon(a0.y, t);
    }
    if (!i1) {
        fp_mul(t, w1, p1.zzz); fp_mul(t, p1.x, t); fp_canon(a1.x, t);
        fp_mul(t, w1, p1.zz); fp_mul(t, p1.y, t); fp_canon(a1.y, t);
    }
    g1_compress(out0, a0);
    g1_compress(out1, a1);
}
// the lane's tail record -> its pinned host page (device-visible host memory): a 1-wave store instead of a copy-engine
// transfer (~16 us on this stack for 832 bytes); the stream synchronisation that follows makes it visible to the host
// clear2 != null: the two input-error flags of the record are zeroed once copied, so the lane's next request finds them
// clean without a memset at its start
// seq_word != null: once every word is out (system-scope fence), the host page's sequence word is set to `seq` -- the
// host polls it instead of waiting for the stream's completion signal (kernel end + signal + wake-up: several us)
__global__ void __launch_bounds__(256) k_publish(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst_host, uint32_t words,
                                                 uint32_t* __restrict__ clear2, uint32_t* __restrict__ seq_word, uint32_t seq) {
    for (uint32_t i = threadIdx.x; i < words; i += 256) dst_host[i] = src[i];
    if (clear2 || seq_word) {
        __threadfence_system();
        __syncthreads();
        if (clear2 && threadIdx.x < 2) clear2[threadIdx.x] = 0;
        if (seq_word && threadIdx.x == 0) {
            __hip_atomic_store(seq_word, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}
// XYZZ working form <-> the 192-byte partial-sum format of the C-ABI (4 x 12 u32: canonical Montgomery residues)
__global__ void __launch_bounds__(64) k_xyzz_pack(const g1_xyzz_t* __restrict__ in, uint32_t* __restrict__ out48w,
                                                   uint32_t count) {
    tail_priority();
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    g1_xyzz_t p;
    load_xyzz(p, &in[i]);
    const bool inf = g1_is_inf(p);
    fp_t c;
    const fp_t* f[4] = {&p.x, &p.y, &p.zz, &p.zzz};
#pragma unroll
    for (int k = 0; k < 4; k++) {
        fp_canon_mont(c, *f[k]);
        if (inf) fp_zero(c);
        fp_pack(out48w + 48 * i + 12 * k, c);
    }
}
__global__ void __launch_bounds__(64) k_xyzz_unpack(const uint32_t* __restrict__ in48w, g1_xyzz_t* __restrict__ out,
                                                     uint32_t count) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    g1_xyzz_t p;
    fp_t* f[4] = {&p.x, &p.y, &p.zz, &p.zzz};
#pragma unroll
    for (int k = 0; k < 4; k++) fp_unpack(*f[k], in48w + 48 * i + 12 * k);
    store_xyzz(&out[i], p);
}

// ------------------------------------------------------------------------------------------------ SRS plumbing
__global__ void __launch_bounds__(256) k_srs_from_be96(const uint8_t* __restrict__ be, g1_affine_t* __restrict__ out,
                                                        uint64_t n, uint32_t* __restrict__ bad) {
    uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const uint32_t* w = reinterpret_cast<const uint32_t*>(be + 96 * j);
    uint32_t any = 0;
#pragma unroll
    for (int i = 0; i < 24; i++) any |= w[i];
    g1_aff28 p;
    if (!any) {
        fp_zero(p.x); fp_zero(p.y);
        g1_store_aff(&out[j], p);
        return;
    }
    bool ok = fp_from_be48(p.x, be + 96 * j);
    ok &= fp_from_be48(p.y, be + 96 * j + 48);
    if (!ok) atomicOr(bad, 1u);
    // on-curve: y^2 == x^3 + 4
    fp_t y2, x3, four, t;
    fp_sqr(y2, p.y);
    fp_sqr(x3, p.x); fp_mul(x3, x3, p.x);
    fp_one(four); fp_dbl(four, four); fp_dbl(four, four);
    fp_add(t, x3, four);
    fp_sub4(t, t, y2);                 // == 0 mod p on the curve
    fp_t one, chk;
    fp_one(one);
    fp_mul(chk, t, one);
    if (!fp_is_zero_n(chk)) atomicOr(bad, 2u);
    g1_store_aff(&out[j], p);
}
// ---- ZCash-compressed SRS files (the reference's `uncompressed=False` setup files: base/miner.py:75-81,
// utils/config.py:131-150): 48 bytes per point, flags compressed 0x80 | infinity 0x40 | y-sign 0x20, x big-endian.
// y = (x^3 + 4)^((p+1)/4) (p = 3 mod 4); a non-residue, x >= p or malformed flags fail the load.
KZG_DEV void fp_pow_p_plus_1_over_4(fp_t& r, const fp_t& a) {
    fp_t acc;
    fp_one(acc);
    for (int i = 378; i >= 0; i--) {  // (p+1)/4 has 379 bits; bit i of it is bit i+2 of p+1
        fp_sqr(acc, acc);
        const int b = i + 2;
        uint32_t w = 0;
#pragma unroll
        for (int k = 0; k < 12; k++) w = (k == (b >> 5)) ? (k == 0 ? FpParams::mod(0) + 1u : FpParams::mod(k)) : w;
        if ((w >> (b & 31)) & 1u) fp_mul(acc, acc, a);
    }
    r = acc;
}
KZG_DEV bool fp_mont_is_larger(const fp_t& y_mont) {  // y > (p-1)/2 for the canonical integer behind y_mont
    fp_t yc;
    fp_from_mont(yc, y_mont);
    uint32_t y[12], t[12], pm[12];
    fp_pack(y, yc);
    const uint32_t c = bi_add<12>(t, y, y);
#pragma unroll
    for (int i = 0; i < 12; i++) pm[i] = FpParams::mod(i);
    return c || bi_ge<12>(t, pm);
}
__global__ void __launch_bounds__(256) k_srs_from_c48(const uint8_t* __restrict__ c48, g1_affine_t* __restrict__ out,
                                                       uint64_t n, uint32_t* __restrict__ bad) {
    uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    uint32_t w[12], pm[12];
    limbs_from_be<12>(w, c48 + 48 * j);  // w[11] holds the flag bits
    const uint32_t flags = w[11] >> 29;
    w[11] &= 0x1fffffffu;
    g1_aff28 p;
    fp_zero(p.x); fp_zero(p.y);
    if (!(flags & 4u)) atomicOr(bad, 1u);  // not a compressed encoding
    if (flags & 2u) {                      // infinity: every other bit must be clear
        uint32_t any = flags & 1u;
#pragma unroll
        for (int i = 0; i < 12; i++) any |= w[i];
        if (any) atomicOr(bad, 1u);
        g1_store_aff(&out[j], p);
        return;
    }
#pragma unroll
    for (int i = 0; i < 12; i++) pm[i] = FpParams::mod(i);
    if (bi_ge<12>(w, pm)) atomicOr(bad, 1u);
    fp_t xr, x3, four, y2, y, t, one, chk;
    fp_unpack(xr, w);
    fp_to_mont(p.x, xr);
    fp_canon(p.x, p.x);
    fp_sqr(x3, p.x); fp_mul(x3, x3, p.x);
    fp_one(one);
    fp_dbl(four, one); fp_dbl(four, four);
    fp_add(t, x3, four);
    fp_mul(y2, t, one);                    // normalised x^3 + 4
    fp_pow_p_plus_1_over_4(y, y2);
    fp_sqr(t, y);
    fp_sub4(t, t, y2);
    fp_mul(chk, t, one);
    if (!fp_is_zero_n(chk)) atomicOr(bad, 2u);  // x^3 + 4 is not a square: no such point
    fp_canon(p.y, y);
    if (fp_mont_is_larger(p.y) != ((flags & 1u) != 0)) fp_neg_canon(p.y, p.y);
    g1_store_aff(&out[j], p);
}
__global__ void __launch_bounds__(256) k_srs_to_c48(const g1_affine_t* __restrict__ in, uint8_t* __restrict__ c48,
                                                     uint64_t n) {
    uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    g1_aff28 p;
    g1_load_aff(p, &in[j]);
    g1_compress(c48 + 48 * j, p);
}
__global__ void __launch_bounds__(256) k_srs_to_be96(const g1_affine_t* __restrict__ in, uint8_t* __restrict__ be,
                                                      uint64_t n) {
    uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    g1_aff28 p;
    g1_load_aff(p, in + j);
    fp_to_be48(be + 96 * j, p.x);
    fp_to_be48(be + 96 * j + 48, p.y);
}

// window tables: tmp[(w-1)*count + j] = 2^off[w] P_{first+j} in XYZZ
__global__ void __launch_bounds__(256) k_precomp_dbl(const g1_affine_t* __restrict__ table, uint64_t first,
                                                      uint64_t count, const WinLayout lay,
                                                      g1_xyzz_t* __restrict__ tmp) {
    uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= count) return;
    g1_aff28 p;
    g1_load_aff(p, table + first + j);
    g1_xyzz_t cur, r;
    g1_from_aff(cur, p);
    for (int w = 1; w < lay.nwin; w++) {
        for (int k = lay.off[w - 1]; k < lay.off[w]; k++) {
            g1_dbl(r, cur);
            cur = r;
        }
        store_xyzz(&tmp[(uint64_t)(w - 1) * count + j], cur);
    }
}
// d = ZZ*ZZZ products chained per lane (Montgomery batch inversion); prefix products are parked in the
// destination slots (as packed canonical residues)
KZG_DEV void park_fp(g1_affine_t* slot, const fp_t& v_loose) {
    fp_t c;
    fp_canon_mont(c, v_loose);
#pragma unroll
    for (int i = 0; i < 14; i++) slot->x[i] = c.l[i];
}
KZG_DEV void unpark_fp(fp_t& v, const g1_affine_t* slot) {
#pragma unroll
    for (int i = 0; i < 14; i++) v.l[i] = slot->x[i];
}
KZG_DEV void xyzz_to_aff_with_inv(g1_aff28& o, const g1_xyzz_t& p, const fp_t& iw /* 1/(zz*zzz) */) {
    fp_t t;
    fp_mul(t, iw, p.zzz);   // 1/zz
    fp_mul(t, p.x, t);
    fp_canon(o.x, t);
    fp_mul(t, iw, p.zz);    // 1/zzz
    fp_mul(t, p.y, t);
    fp_canon(o.y, t);
}
__global__ void __launch_bounds__(256) k_precomp_norm(g1_affine_t* __restrict__ table, uint64_t stride,
                                                       uint64_t first, uint64_t count, int nwin,
                                                       const g1_xyzz_t* __restrict__ tmp) {
    uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= count) return;
    fp_t acc, d;
    fp_one(acc);
    for (int w = 1; w < nwin; w++) {
        g1_xyzz_t p;
        load_xyzz(p, &tmp[(uint64_t)(w - 1) * count + j]);
        park_fp(&table[(uint64_t)w * stride + first + j], acc);
        if (!g1_is_inf(p)) {
            fp_mul(d, p.zz, p.zzz);
            fp_mul(acc, acc, d);
        }
    }
    fp_t inv;
    fp_inv_fermat(inv, acc);
    for (int w = nwin - 1; w >= 1; w--) {
        g1_xyzz_t p;
        load_xyzz(p, &tmp[(uint64_t)(w - 1) * count + j]);
        g1_affine_t* dst = &table[(uint64_t)w * stride + first + j];
        g1_aff28 o;
        if (g1_is_inf(p)) {
            fp_zero(o.x); fp_zero(o.y);
        } else {
            fp_t pre, iw;
            unpark_fp(pre, dst);
            fp_mul(d, p.zz, p.zzz);
            fp_mul(iw, inv, pre);   // 1 / (zz*zzz)
            fp_mul(inv, inv, d);
            xyzz_to_aff_with_inv(o, p, iw);
        }
        g1_store_aff(dst, o);
    }
}

// ---- synthetic SRS (tests / benches): out[j] = [s0 tau^j] G via an 8-bit fixed-base table of G
KZG_DEV void g1_generator(g1_aff28& g) {
    constexpr uint32_t gx[12] = {0xdb22c6bbu, 0xfb3af00au, 0xf97a1aefu, 0x6c55e83fu, 0x171bac58u, 0xa14e3a3fu,
                                 0x9774b905u, 0xc3688c4fu, 0x4fa9ac0fu, 0x2695638cu, 0x3197d794u, 0x17f1d3a7u};
    constexpr uint32_t gy[12] = {0x46c5e7e1u, 0x0caa2329u, 0xa2888ae4u, 0xd03cc744u, 0x2c04b3edu, 0x00db18cbu,
                                 0xd5d00af6u, 0xfcf5e095u, 0x741d8ae4u, 0xa09e30edu, 0xe3aaa0f1u, 0x08b3f481u};
    uint32_t wx[12], wy[12];
#pragma unroll
    for (int i = 0; i < 12; i++) { wx[i] = gx[i]; wy[i] = gy[i]; }
    fp_t x, y;
    fp_unpack(x, wx);
    fp_unpack(y, wy);
    fp_to_mont(g.x, x); fp_canon(g.x, g.x);
    fp_to_mont(g.y, y); fp_canon(g.y, g.y);
}
// gtab[w*255 + d] = (d+1) * 2^(8w) * G, affine
__global__ void __launch_bounds__(256) k_gen_gtab(g1_affine_t* __restrict__ gtab) {
    uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= 32 * 255) return;
    uint32_t w = gid / 255, d = gid % 255 + 1;
    g1_aff28 g;
    g1_generator(g);
    g1_xyzz_t base, r, acc;
    g1_from_aff(base, g);
    for (uint32_t k = 0; k < 8 * w; k++) { g1_dbl(r, base); base = r; }
    g1_set_inf(acc);
    for (int b = 7; b >= 0; b--) {
        g1_dbl(r, acc); acc = r;
        if ((d >> b) & 1u) { g1_add(r, acc, base); acc = r; }
    }
    g1_aff28 o;
    fp_t t, iw;
    fp_mul(t, acc.zz, acc.zzz);
    fp_inv_fermat(iw, t);
    xyzz_to_aff_with_inv(o, acc, iw);
    g1_store_aff(&gtab[gid], o);
}
// scal[j] = s0 * tau^j (canonical limbs); 64 consecutive j per lane
__global__ void __launch_bounds__(256) k_srs_scalars(uint32_t* __restrict__ scal, uint64_t count, uint64_t j_base,
                                                      const uint32_t* __restrict__ tau_mont,
                                                      const uint32_t* __restrict__ s0_mont) {
    uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t j0 = t * 64;
    if (j0 >= count) return;
    fr9_t tau, cur, pw;
    fr9_load(tau, tau_mont);
    fr9_load(cur, s0_mont);
    pw = tau;  // cur *= tau^(j_base + j0)
    for (uint64_t e = j_base + j0; e; e >>= 1) {
        if (e & 1) fr9_mul(cur, cur, pw);
        fr9_mul(pw, pw, pw);
    }
    for (uint64_t j = j0; j < j0 + 64 && j < count; j++) {
        fr9_t c;
        fr9_from_mont(c, cur);
        fr9_store(scal + 8 * j, c);
        fr9_mul(cur, cur, tau);
    }
}
__global__ void __launch_bounds__(256) k_srs_fixed_mul(const uint32_t* __restrict__ scal, uint64_t count,
                                                        const g1_affine_t* __restrict__ gtab,
                                                        g1_xyzz_t* __restrict__ tmp) {
    uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= count) return;
    uint32_t s[8];
    load_scalar(s, scal, j, 0);
    g1_xyzz_t acc;
    g1_set_inf(acc);
    for (int w = 0; w < 32; w++) {
        uint32_t d = (limb_at(s, w >> 2) >> (8 * (w & 3))) & 0xffu;
        if (d) {
            g1_aff28 p;
            g1_load_aff(p, gtab + w * 255 + d - 1);
            g1_madd<true>(acc, p.x, p.y);
        }
    }
    store_xyzz(&tmp[j], acc);
}
// XYZZ -> affine for `count` points, 16 consecutive points per lane share one inversion
__global__ void __launch_bounds__(256) k_batch_affine(const g1_xyzz_t* __restrict__ tmp, g1_affine_t* __restrict__ out,
                                                       uint64_t count) {
    uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t j0 = t * 16;
    if (j0 >= count) return;
    uint64_t j1 = min(j0 + 16, count);
    fp_t acc, d, inv;
    fp_one(acc);
    for (uint64_t j = j0; j < j1; j++) {
        g1_xyzz_t p;
        load_xyzz(p, &tmp[j]);
        park_fp(&out[j], acc);
        if (!g1_is_inf(p)) {
            fp_mul(d, p.zz, p.zzz);
            fp_mul(acc, acc, d);
        }
    }
    fp_inv_fermat(inv, acc);
    for (uint64_t j = j1; j-- > j0;) {
        g1_xyzz_t p;
        load_xyzz(p, &tmp[j]);
        g1_aff28 o;
        if (g1_is_inf(p)) {
            fp_zero(o.x); fp_zero(o.y);
        } else {
            fp_t pre, iw;
            unpark_fp(pre, &out[j]);
            fp_mul(d, p.zz, p.zzz);
            fp_mul(iw, inv, pre);
            fp_mul(inv, inv, d);
            xyzz_to_aff_with_inv(o, p, iw);
        }
        g1_store_aff(&out[j], o);
    }
}

// ------------------------------------------------------------------------------------------------ launchers
static inline uint32_t nblk(uint64_t n, uint32_t b) { return (uint32_t)((n + b - 1) / b); }

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
#ifndef KZG_TREE_WIDE_MIN
#define KZG_TREE_WIDE_MIN 32768
#endif
void launch_msm_tree_level(hipStream_t s, const g1_xyzz_t* in, const g1_xyzz_t* prev, g1_xyzz_t* out,
                           uint32_t n_in_nodes, int level) {
    uint32_t ops = (n_in_nodes >> 1) * (uint32_t)(level + 1);
    // wide levels are throughput-bound (one lane per addition); narrow ones are latency-bound (4 waves per addition)
    // ... and the narrowest ones, where even that leaves the chip empty, run one WAVE per addition (fp_lp.hip.h)
    if (ops > KZG_TREE_WIDE_MIN) k_msm_tree_level<<<nblk(ops, 256), 256, 0, s>>>(in, prev, out, n_in_nodes, level);
#ifndef KZG_NO_LP
    else if (ops <= LP_MAX_OPS) k_msm_tree_level_lp<<<ops, 64, 0, s>>>(in, prev, out, n_in_nodes, level);
#endif
    else k_msm_tree_level_coop<<<nblk(ops, 64), 256, 0, s>>>(in, prev, out, n_in_nodes, level);
}
bool msm_tree_level2_ok(uint32_t n_in_nodes, int level) {
#if defined(KZG_NO_LP) || defined(KZG_NO_TREE_PAIRS)
    (void)n_in_nodes; (void)level;
    return false;
#else
    return n_in_nodes >= 4 && (n_in_nodes >> 1) * (uint32_t)(level + 1) <= LP_MAX_OPS;
#endif
}
void launch_msm_tree_level2(hipStream_t s, const g1_xyzz_t* in, const g1_xyzz_t* prev, g1_xyzz_t* mid_p, g1_xyzz_t* out,
                            uint32_t n_in_nodes, int level) {
    k_msm_tree_level2_lp<<<(n_in_nodes >> 2) * (uint32_t)(level + 2), 128, 0, s>>>(in, prev, mid_p, out, n_in_nodes, level);
}
void launch_msm_final(hipStream_t s, const g1_xyzz_t* node, const g1_xyzz_t* prev, int nbits, int nodes,
                      g1_xyzz_t* out_xyzz, g1_xyzz_t* scratch) {
#ifndef KZG_NO_LP
    k_msm_final_dbl_lp<<<nodes * (nbits + 1), 64, 0, s>>>(node, prev, nbits, nodes, scratch);
    k_msm_final_sum_lp<<<nodes, 512, 0, s>>>(scratch, nbits, out_xyzz);
#else
    (void)scratch;
    k_msm_final<<<nodes, 256, 0, s>>>(node, prev, nbits, nodes, out_xyzz);
#endif
}
void launch_g1_sum(hipStream_t s, const g1_xyzz_t* in, uint32_t count, g1_xyzz_t* out_xyzz) {
    if (count >= 2 && count <= 32) k_g1_sum_lp<<<1, 512, 0, s>>>(in, count, out_xyzz);
    else k_g1_sum<<<1, 64, 0, s>>>(in, count, out_xyzz);
}
void launch_g1_sum_affine(hipStream_t s, const g1_affine_t* in, uint32_t count, g1_xyzz_t* out_xyzz) {
    k_g1_sum_affine<<<1, 64, 0, s>>>(in, count, out_xyzz);
}
void launch_g1_subgroup_check(hipStream_t s, const g1_affine_t* in, uint32_t count, uint32_t* bad_flag) {
    if (count) k_g1_subgroup_check_lp<<<count, 64, 0, s>>>(in, count, bad_flag);
}
void launch_g1_subgroup_check_bulk(hipStream_t s, const g1_affine_t* in, uint64_t n, uint32_t* bad_flag) {
    if (n) k_g1_subgroup_check<<<nblk(n, 256), 256, 0, s>>>(in, n, bad_flag);
}
void launch_g1_compress(hipStream_t s, const g1_xyzz_t* in, uint8_t* out48) {
    k_g1_compress<<<1, 64, 0, s>>>(in, out48);
}
void launch_g1_compress_pair(hipStream_t s, const g1_xyzz_t* in0, const g1_xyzz_t* in1, uint8_t* out0, uint8_t* out1) {
    k_g1_compress_pair<<<1, 64, 0, s>>>(in0, in1, out0, out1);
}
void launch_publish(hipStream_t s, const void* src_dev, void* dst_host_devptr, uint32_t bytes, uint32_t* clear2,
                    uint32_t* seq_word_devptr, uint32_t seq) {
    k_publish<<<1, 256, 0, s>>>(reinterpret_cast<const uint32_t*>(src_dev), reinterpret_cast<uint32_t*>(dst_host_devptr), bytes / 4,
                                clear2, seq_word_devptr, seq);
}
void launch_xyzz_pack(hipStream_t s, const g1_xyzz_t* in, uint32_t* out48w, uint32_t count) {
    if (count) k_xyzz_pack<<<nblk(count, 64), 64, 0, s>>>(in, out48w, count);
}
void launch_xyzz_unpack(hipStream_t s, const uint32_t* in48w, g1_xyzz_t* out, uint32_t count) {
    if (count) k_xyzz_unpack<<<nblk(count, 64), 64, 0, s>>>(in48w, out, count);
}
void launch_srs_from_be96(hipStream_t s, const uint8_t* be96, g1_affine_t* out, uint64_t n, uint32_t* bad_flag) {
    if (!n) return;
    k_srs_from_be96<<<nblk(n, 256), 256, 0, s>>>(be96, out, n, bad_flag);
}
void launch_srs_from_c48(hipStream_t s, const uint8_t* c48, g1_affine_t* out, uint64_t n, uint32_t* bad_flag) {
    if (!n) return;
    k_srs_from_c48<<<nblk(n, 256), 256, 0, s>>>(c48, out, n, bad_flag);
}
void launch_srs_to_c48(hipStream_t s, const g1_affine_t* in, uint8_t* c48, uint64_t n) {
    if (!n) return;
    k_srs_to_c48<<<nblk(n, 256), 256, 0, s>>>(in, c48, n);
}
void launch_srs_to_be96(hipStream_t s, const g1_affine_t* in, uint8_t* be96, uint64_t n) {
    if (!n) return;
    k_srs_to_be96<<<nblk(n, 256), 256, 0, s>>>(in, be96, n);
}
void launch_srs_precompute(hipStream_t s, g1_affine_t* table, uint64_t stride, uint64_t first, uint64_t count,
                           const WinLayout& lay, g1_xyzz_t* tmp) {
    if (!count || lay.nwin < 2) return;
    k_precomp_dbl<<<nblk(count, 256), 256, 0, s>>>(table, first, count, lay, tmp);
    k_precomp_norm<<<nblk(count, 256), 256, 0, s>>>(table, stride, first, count, lay.nwin, tmp);
}
void launch_srs_generate(hipStream_t s, g1_affine_t* out, uint64_t count, uint64_t j_base, const uint32_t* tau_mont,
                         const uint32_t* s0_mont, g1_affine_t* gtab, g1_xyzz_t* tmp, bool build_gtab) {
    if (!count) return;
    if (build_gtab) k_gen_gtab<<<nblk(32 * 255, 256), 256, 0, s>>>(gtab);
    uint32_t* sbuf = reinterpret_cast<uint32_t*>(tmp + count);  // scalars staged behind the XYZZ scratch
    k_srs_scalars<<<nblk((count + 63) / 64, 256), 256, 0, s>>>(sbuf, count, j_base, tau_mont, s0_mont);
    k_srs_fixed_mul<<<nblk(count, 256), 256, 0, s>>>(sbuf, count, gtab, tmp);
    k_batch_affine<<<nblk((count + 15) / 16, 256), 256, 0, s>>>(tmp, out, count);
}
