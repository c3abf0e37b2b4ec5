// msm_tree.hip -- stages 3 and 4 of the Pippenger MSM: the bucket tree (pairwise merges keeping P = sum B_k and
// T_i = sum over the buckets whose index has bit i) and the final combination P + sum 2^i T_i.  Wide levels one lane per
// addition (multiplier-bound), narrow ones cooperative or lane-parallel (latency-bound).
#include "msm_dev.hip.h"

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


// ------------------------------------------------------------------------------------------------ launchers
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
