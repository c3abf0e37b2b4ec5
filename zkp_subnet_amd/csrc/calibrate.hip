// Same-run measurement of what bounds k_msm_accumulate: the issue rate of v_mad_u64_u32, the only
// wide integer multiply of gfx950 and 3546 of the ~4900 VALU instructions of one mixed XYZZ addition.  The kernel is a
// dependent-free chain of that ONE instruction (8 independent 64-bit accumulators per lane, so the 8-deep unroll never
// waits on its own result), run with `waves_per_simd` waves on every SIMD of the chip for ~1-2 ms.  bench.py calls it
// right after the timed region, on the same box, at the clocks the timed region left behind: mad_issue.frac is then a
// ratio of two measurements of one run (boxes of the pool differ by up to 14 % in this rate, profiles/r03_*).
// s_memtime ticks over the same span give the clock the SIMDs ran at.
#include <hip/hip_runtime.h>

#include <cstdint>

#define CAL_UNROLL 8

__global__ void __launch_bounds__(256) k_calibrate_mad(uint64_t* __restrict__ out, uint32_t seed, uint32_t iters) {
    uint32_t a = seed + threadIdx.x, b = seed * 3 + 1;
    uint64_t acc[CAL_UNROLL];
#pragma unroll
    for (int i = 0; i < CAL_UNROLL; i++) acc[i] = a + i;
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (uint32_t it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < CAL_UNROLL; i++)
            asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b) : "vcc");
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    uint64_t s = 0;
#pragma unroll
    for (int i = 0; i < CAL_UNROLL; i++) s += acc[i];
    out[2 + (size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;    // keeps the chain alive
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;     // ticks of wave 0 over its `iters` x 8 instructions
}

// blocks x 256 threads (one wave per SIMD of a CU per block); out: 2 + blocks * 256 words
void launch_calibrate_mad(hipStream_t s, uint64_t* out, uint32_t blocks, uint32_t iters) {
    hipLaunchKernelGGL(k_calibrate_mad, dim3(blocks), dim3(256), 0, s, out, 7u, iters);
}
int calibrate_unroll() { return CAL_UNROLL; }

// ---- do two lanes of a context really run side by side?  (kzg_runtime_info)  The HIP runtime maps a process's streams onto
// GPU_MAX_HW_QUEUES hardware queues (default 4) in creation order, and kernels of two streams that share a queue execute
// one after the other.  One single-wave kernel per lane spins for `ticks` of the constant-rate wall clock and leaves its
// first and last reading: overlapping intervals = concurrent lanes.  Bounded: at most 2^22 polls whatever the clock does.
__global__ void __launch_bounds__(64) k_spin_probe(uint64_t* __restrict__ out2, uint64_t ticks) {
    if (threadIdx.x) return;
    const uint64_t t0 = wall_clock64();
    uint64_t t1 = t0;
    for (uint32_t i = 0; i < (1u << 22) && t1 - t0 < ticks; i++) {
        __builtin_amdgcn_s_sleep(16);
        t1 = wall_clock64();
    }
    out2[0] = t0;
    out2[1] = t1;
}
void launch_spin_probe(hipStream_t s, uint64_t* out2, uint64_t ticks) {
    hipLaunchKernelGGL(k_spin_probe, dim3(1), dim3(64), 0, s, out2, ticks);
}
