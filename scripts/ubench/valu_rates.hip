// Instruction-throughput microbenchmark for gfx950 integer / fp64 VALU ops that matter for 381-bit Montgomery
// arithmetic.  Each kernel runs a long chain of ONE instruction kind with 8 independent accumulators per lane,
// at 1, 2 and 4 waves per SIMD; reports cycles per wave-instruction per SIMD (via wall time and s_memtime).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define ITERS 4096
#define UNROLL 8
template <int OP>
__global__ void __launch_bounds__(256) k(uint64_t* out, uint32_t seed) {
    uint32_t a = seed + threadIdx.x, b = seed * 3 + 1;
    uint64_t acc[UNROLL];
    double d[UNROLL];
    for (int i = 0; i < UNROLL; i++) { acc[i] = a + i; d[i] = (double)(a + i); }
    double da = (double)a * 1e-3, db = (double)b;
    uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < UNROLL; i++) {
            if (OP == 0) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b) : "vcc");
            if (OP == 1) { uint32_t x = (uint32_t)acc[i]; asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x) : "v"(b)); acc[i] = x; }
            if (OP == 2) { uint32_t x = (uint32_t)acc[i]; asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x) : "v"(b)); acc[i] = x; }
            if (OP == 3) { uint32_t x = (uint32_t)acc[i]; asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(x) : "v"(b)); acc[i] = x; }
            if (OP == 4) { uint32_t x = (uint32_t)acc[i]; asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(x) : "v"(b)); acc[i] = x; }
            if (OP == 5) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(d[i]) : "v"(da), "v"(db));
            if (OP == 6) { uint32_t x = (uint32_t)acc[i]; asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(b)); acc[i] = x; }
            if (OP == 7) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(acc[i]) : "v"(acc[(i + 1) % UNROLL]));
            if (OP == 8) { uint32_t x = (uint32_t)acc[i]; asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(x) : "v"(b) : "vcc"); acc[i] = x; }
            if (OP == 9) { uint32_t x = (uint32_t)acc[i]; asm volatile("v_add_co_u32 %0, s[10:11], %0, %1\n\tv_addc_co_u32 %0, vcc, 0, %0, s[10:11]" : "+v"(x) : "v"(b) : "vcc", "s10", "s11"); acc[i] = x; }
            if (OP == 10) { uint32_t x = (uint32_t)acc[i]; asm volatile("v_mad_i32_i24 %0, %0, %1, %0" : "+v"(x) : "v"(b)); acc[i] = x; }
            if (OP == 11) { uint32_t x = (uint32_t)acc[i]; asm volatile("v_dot4_u32_u8 %0, %0, %1, %0" : "+v"(x) : "v"(b)); acc[i] = x; }
            if (OP == 12) { uint32_t x = (uint32_t)acc[i]; asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(b) : ); acc[i] = x; }
            if (OP == 15) { uint32_t x = (uint32_t)acc[i]; asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[10:11]" : "+v"(x) : "v"(b) : "s10", "s11"); acc[i] = x; }
            if (OP == 16) { uint32_t x = (uint32_t)acc[i]; asm volatile("v_mov_b32 %0, %1" : "=v"(x) : "v"(b + i)); acc[i] += x; }
            if (OP == 17) { uint32_t x = (uint32_t)acc[i]; asm volatile("v_and_b32 %0, %0, %1" : "+v"(x) : "v"(b)); acc[i] = x; }
            if (OP == 18) { asm volatile("v_lshrrev_b64 %0, 28, %0" : "+v"(acc[i])); }
            if (OP == 13) { uint32_t x = (uint32_t)acc[i]; asm volatile("v_alignbit_b32 %0, %0, %1, 1" : "+v"(x) : "v"(b)); acc[i] = x; }
            if (OP == 14) { uint32_t x = (uint32_t)acc[i]; asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_addc_co_u32 %3, vcc, 0, %3, vcc" : "+v"(acc[i]), "+v"(x) : "v"(a), "v"(b), "v"(x) : "vcc"); }
        }
    }
    uint64_t t1 = __builtin_amdgcn_s_memtime();
    uint64_t s = 0;
    for (int i = 0; i < UNROLL; i++) s += acc[i] + (uint64_t)d[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[1 << 20] = t1 - t0;
}
template <int OP>
void run(const char* name, uint64_t* d_out) {
    for (int waves_per_simd : {1, 2, 4}) {
        int blocks = 256 * waves_per_simd;  // 256 CUs x (4 waves = 1 per SIMD) per block of 256 threads
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        k<OP><<<blocks, 256>>>(d_out, 7);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        k<OP><<<blocks, 256>>>(d_out, 7);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        uint64_t cyc; hipMemcpy(&cyc, d_out + (1 << 20), 8, hipMemcpyDeviceToHost);
        double n_inst = (double)ITERS * UNROLL * waves_per_simd;   // wave-instructions per SIMD
        printf("%-22s waves/SIMD=%d  wall %.3f ms  ns/inst/SIMD %.3f  (memtime clk %.2f per inst of this wave = %.2f per SIMD-inst)\n", name,
               waves_per_simd, ms, ms * 1e6 / n_inst, (double)cyc / (ITERS * UNROLL), (double)cyc / (ITERS * UNROLL) / waves_per_simd);
    }
}
int main() {
    uint64_t* d_out; hipMalloc(&d_out, ((1 << 20) + 16) * 8);
    run<0>("v_mad_u64_u32", d_out); run<14>("mad_u64_u32+addc", d_out);
    run<1>("v_mul_lo_u32", d_out); run<2>("v_mul_hi_u32", d_out);
    run<3>("v_mad_u32_u24", d_out); run<4>("v_mul_hi_u32_u24", d_out); run<10>("v_mad_i32_i24", d_out);
    run<5>("v_fma_f64", d_out); run<6>("v_add_u32", d_out); run<7>("v_lshl_add_u64", d_out);
    run<8>("v_addc_co_u32 chain", d_out); run<9>("add_co+addc (sgpr)", d_out); run<11>("v_dot4_u32_u8", d_out);
    run<12>("v_cndmask_b32 vcc", d_out); run<15>("v_cndmask_b32 sgpr", d_out); run<16>("v_mov_b32(+add)", d_out);
    run<17>("v_and_b32", d_out); run<18>("v_lshrrev_b64", d_out); run<13>("v_alignbit_b32", d_out);
    return 0;
}
