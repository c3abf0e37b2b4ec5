// Pippenger G1 MSM for gfx950 -- kernel declarations shared by the msm_*.hip kernel files and the host side (pipeline.hip, srs.hip, serve.hip, comm.hip).
//
// Design (DESIGN.md "MSM"): the SRS is fixed, so every point P_j is stored with its window multiples
// 2^off[w] P_j (table[w][j], affine, Montgomery; sized for 288 GB HBM).  All nwin signed digits of all
// scalars then fall into ONE set of B = 2^(c-1) buckets:
//   1-3. msm_sort        scalar -> signed digits; two-level counting sort with all per-entry atomics in LDS:
//                        partition by the key's high bits, then one workgroup per partition sorts the low bits
//                        -> offsets[bucket], sorted[] = table index | sign
//   4. msm_accumulate    HOT: every lane sums a fixed-size chunk of the sorted entries with mixed XYZZ adds
//                        (perfect load balance for any scalar distribution); bucket runs that span chunks
//                        leave "carry" partial sums
//   5. fold_step/heads   folds the carries into their buckets: per-bucket binary tree, log depth for ANY distribution
//   6. msm_tree_level    log2(B) pairwise-merge levels producing P = sum B_k and T_i = sum_{bit i of k} B_k
//   7. msm_final         sum = P + sum_i 2^i T_i  (-> XYZZ, optionally affine + 48-byte compression)
#pragma once
#include "g1.hip.h"

// Window w covers scalar bits [off[w], off[w+1]); off[nwin] = 256.  Widths differ by at most one bit
// (256 = nwin*base + extra), so no window is a short "top" window that would pile its digits on a few buckets.
struct WinLayout {
    int nwin;
    uint16_t off[66];
};
struct MsmShape {
    int c;               // widest window, bits
    int nwin;            // signed windows
    WinLayout lay;
    uint32_t nbuckets;   // 2^(c-1) per MSM; nbatch MSMs over the same points use nbatch consecutive bucket sets
    uint64_t n;          // number of (scalar, point) pairs in each MSM of the batch
    int nbatch;          // 1, or 2: commit + open as ONE pass (scalar set b -> bucket set b), see launch_msm_sort
    uint64_t srs_offset; // first point of the slice inside the resident SRS
    uint64_t srs_stride; // points per window table (= total resident SRS points)
    int chunk;           // sorted entries per lane in msm_accumulate
};

// signed-digit recode + two-level counting sort: fills offsets[0..nbuckets] and sorted[0..entries).
// part_ws: 16384 u32 scratch; parted: one uint2 per entry
// With sh.nbatch == 2 the second scalar set (scalars2, same length, same points) is sorted into the second bucket set:
// the key gets one more high bit, everything downstream just sees 2 * nbuckets buckets.
// part_ws_clean: the partition counts / cursors are known to be zero (a completed sort leaves them so); max_len_word: the
// MSM's fold-depth word, reset here (launch_fold_maxlen accumulates into it).  fast: no count pass, fixed-capacity
// partition regions (parted must hold msm_sort_parted_entries(sh, true) entries); when a region overflows,
// *overflow_word is raised, the offsets come out all zero, and the caller reruns with fast = false (which clears the word).
// tail != null: the sort's last kernel also does what launch_fold_maxlen + launch_publish would do behind it (two launches
// less in front of the accumulate): marks the empty buckets, takes the maximum of the carry-run lengths into
// *max_len_word, and the LAST workgroup to finish writes {max_len, overflow} to pin_dst (device-visible host memory) and
// then `seq` to seq_word.
struct SortTail {
    uint32_t chunk;          // sorted entries per accumulate lane
    g1_xyzz_t* buckets;      // [nbuckets * nbatch]
    uint32_t* pin_dst;       // two words of the lane's pinned page
    uint32_t* seq_word;      // the page's sequence word
    uint32_t seq;
};
void launch_msm_sort(hipStream_t s, const MsmShape& sh, const uint32_t* scalars, int scalars_mont,
                     const uint32_t* scalars2, int scalars2_mont, uint32_t* part_ws, bool part_ws_clean, uint2* parted,
                     uint32_t* offsets, uint32_t* sorted, uint32_t* max_len_word, bool fast, uint32_t* overflow_word,
                     const SortTail* tail = nullptr);
bool msm_sort_fast_ok(const MsmShape& sh);
uint64_t msm_sort_parted_entries(const MsmShape& sh, bool fast);
// bytes (multiple of 4) from device memory to a device-visible host pointer, by a kernel
// clear2: two device words zeroed after the copy (the record's input-error flags), or null
// seq_word_devptr: a word of the host page set to `seq` after everything else has landed (the host may poll it)
void launch_publish(hipStream_t s, const void* src_dev, void* dst_host_devptr, uint32_t bytes, uint32_t* clear2 = nullptr,
                    uint32_t* seq_word_devptr = nullptr, uint32_t seq = 0);
void launch_msm_accumulate(hipStream_t s, const MsmShape& sh, const g1_affine_t* table, const uint32_t* offsets,
                           const uint32_t* sorted, g1_xyzz_t* buckets, g1_xyzz_t* carries, uint32_t* carry_key,
                           uint32_t nchunks);
// carries -> buckets: per-bucket binary tree over the carries (positions derived from the bucket offsets).
// max_len: device word, maximum number of carries of one bucket (stays 0 when every run has <= 1 carry)
// (also marks the empty buckets as infinity: the bucket array needs no memset)
void launch_fold_maxlen(hipStream_t s, const uint32_t* offsets, uint32_t nbuckets, uint32_t chunk, uint32_t* max_len,
                        g1_xyzz_t* buckets);
// short rows: the whole fold (steps + heads) as ONE launch, one wave per bucket walking its run of carries; only when
// msm_fold_bucket_ok(buckets, longest run as published by the sort) -- the chain is serial in the run length
bool msm_fold_bucket_ok(uint32_t nbuckets, uint32_t max_run);
void launch_fold_bucket(hipStream_t s, const uint32_t* offsets, uint32_t chunk, uint32_t nbuckets, const g1_xyzz_t* carries,
                        g1_xyzz_t* buckets);
void launch_fold_step(hipStream_t s, const uint32_t* offsets, const uint32_t* carry_key, uint32_t chunk,
                      uint32_t nchunks, uint32_t d, g1_xyzz_t* carries);
void launch_fold_heads(hipStream_t s, const uint32_t* offsets, const uint32_t* carry_key, uint32_t chunk,
                       uint32_t nchunks, const g1_xyzz_t* carries, g1_xyzz_t* buckets);
// level i: in = nodes of (i+1) points, component-major; out = nodes of (i+2) points
// one merge level: `in` = the level-`level` array (n_in_nodes nodes; P, T_0 .. T_{level-2} stored component-major),
// `prev` = the level below it (its P array holds the T_{level-1} of `in`'s nodes; unused at level 0), `out` = the
// level + 1 array.  in, prev and out must be three different buffers.
void launch_msm_tree_level(hipStream_t s, const g1_xyzz_t* in, const g1_xyzz_t* prev, g1_xyzz_t* out,
                           uint32_t n_in_nodes, int level);
// two merge levels in one launch (narrow levels only: msm_tree_level2_ok): `out` = the level + 2 array, `mid_p` = the P
// array of level + 1 (n_in_nodes / 2 nodes; the next merge's `prev`).  in, prev, mid_p and out: four different buffers;
// mid_p and out need (LP_MAX_OPS + 64) points at most.
bool msm_tree_level2_ok(uint32_t n_in_nodes, int level);
void launch_msm_tree_level2(hipStream_t s, const g1_xyzz_t* in, const g1_xyzz_t* prev, g1_xyzz_t* mid_p, g1_xyzz_t* out,
                            uint32_t n_in_nodes, int level);
// node = [P, T_0 .. T_{nbits-1}] ; out_xyzz = P + sum 2^i T_i
// nodes: the bucket tree stopped at `nodes` roots (component-major: component k of root m at node[k * nodes + m];
// T_{nbits-1} of root m is P[2m + 1] of the level below, `prev`); out_xyzz[m] = P_m + sum_i 2^i T_{i,m}
// scratch: nodes * 32 XYZZ points (the doubled components between the two launches of the lane-parallel form)
void launch_msm_final(hipStream_t s, const g1_xyzz_t* node, const g1_xyzz_t* prev, int nbits, int nodes,
                      g1_xyzz_t* out_xyzz, g1_xyzz_t* scratch);
// sum `count` XYZZ points (count <= 1024) into out_xyzz[0]
void launch_g1_sum(hipStream_t s, const g1_xyzz_t* in, uint32_t count, g1_xyzz_t* out_xyzz);
// sum `count` affine table-format points into out_xyzz[0]
void launch_g1_sum_affine(hipStream_t s, const g1_affine_t* in, uint32_t count, g1_xyzz_t* out_xyzz);
// G1 membership (prime-order subgroup) of `count` affine table-format points, one wave each: *bad_flag |= 4 on a failure
void launch_g1_subgroup_check(hipStream_t s, const g1_affine_t* in, uint32_t count, uint32_t* bad_flag);
// the same test, one lane per point: whole setup files
void launch_g1_subgroup_check_bulk(hipStream_t s, const g1_affine_t* in, uint64_t n, uint32_t* bad_flag);
// affine + ZCash compression of one point
void launch_g1_compress(hipStream_t s, const g1_xyzz_t* in, uint8_t* out48);
// two points, one shared inversion
void launch_g1_compress_pair(hipStream_t s, const g1_xyzz_t* in0, const g1_xyzz_t* in1, uint8_t* out0, uint8_t* out1);
// working XYZZ (224 B) <-> ABI partial-sum format (4 x 48 B packed canonical Montgomery residues, zeros = infinity)
void launch_xyzz_pack(hipStream_t s, const g1_xyzz_t* in, uint32_t* out48w, uint32_t count);
void launch_xyzz_unpack(hipStream_t s, const uint32_t* in48w, g1_xyzz_t* out, uint32_t count);

// SRS plumbing
void launch_srs_from_be96(hipStream_t s, const uint8_t* be96, g1_affine_t* out, uint64_t n, uint32_t* bad_flag);
void launch_srs_from_c48(hipStream_t s, const uint8_t* c48, g1_affine_t* out, uint64_t n, uint32_t* bad_flag);
void launch_srs_to_c48(hipStream_t s, const g1_affine_t* in, uint8_t* c48, uint64_t n);
void launch_srs_to_be96(hipStream_t s, const g1_affine_t* in, uint8_t* be96, uint64_t n);
// window tables for points [first, first+count): tmp holds (nwin-1)*count XYZZ values
void launch_srs_precompute(hipStream_t s, g1_affine_t* table, uint64_t stride, uint64_t first, uint64_t count,
                           const WinLayout& lay, g1_xyzz_t* tmp);
// synthetic SRS: out[j] = [s0 * tau^(j_base+j)] G for j < count (discrete logs known -> tests / benches only)
// (global index j_base + j); gtab: 32*255 affine scratch (built when build_gtab); tmp: count XYZZ + count*32 B
void launch_srs_generate(hipStream_t s, g1_affine_t* out, uint64_t count, uint64_t j_base, const uint32_t* tau_mont,
                         const uint32_t* s0_mont, g1_affine_t* gtab, g1_xyzz_t* tmp, bool build_gtab);
