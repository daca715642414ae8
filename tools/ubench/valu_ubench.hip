// valu_ubench.hip -- integer VALU issue-rate micro-benchmark for gfx950 (measurement tool, not product).
// Each kernel runs ITER iterations of 8 independent dependency chains of one instruction kind with
// 8 waves/SIMD resident everywhere; prints wave64-instructions per SIMD per cycle-equivalent (ns).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define ITER 4096

template <int OP>
__global__ void __launch_bounds__(256) k(uint32_t *out, uint32_t a, uint32_t b)
{
    uint32_t x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 2654435761u + i * a;
    uint32_t sel = b | 0x03020100u;
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (OP == 0) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(x[i]) : "v"(sel));
            if (OP == 1) asm volatile("v_add_u32 %0, %1, %0" : "+v"(x[i]) : "v"(sel));
            if (OP == 2) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(sel), "v"(sel));
            if (OP == 3) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0xe4" : "+v"(x[i]) : "v"(sel), "v"(a));
            if (OP == 4) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x[i]) : "v"(sel));
            if (OP == 5) { uint64_t r; asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(r) : "v"(x[i]), "v"(sel) : "vcc"); x[i] = (uint32_t)(r >> 32) ^ (uint32_t)r; }
            if (OP == 6) asm volatile("v_lshrrev_b32 %0, 7, %0" : "+v"(x[i]));
            if (OP == 7) asm volatile("v_bcnt_u32_b32 %0, %0, %1" : "+v"(x[i]) : "v"(sel));
            if (OP == 8) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x[i]) : "v"(sel));
            if (OP == 9) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(sel), "v"(a));
            if (OP == 10) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(sel), "v"(a));
            if (OP == 11) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[i]) : "v"(sel) : "vcc");
            if (OP == 12) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(x[i]) : "v"(sel));
            if (OP == 13) asm volatile("v_sub_u32 %0, %1, %0" : "+v"(x[i]) : "v"(sel));
            if (OP == 14) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(x[i]) : "v"(sel));
            if (OP == 15) { uint64_t r = ((uint64_t)x[i] << 32) | sel; asm volatile("v_lshlrev_b64 %0, %1, %0" : "+v"(r) : "v"(a)); x[i] = (uint32_t)(r >> 32); }
            if (OP == 16) { uint64_t r = ((uint64_t)x[i] << 32) | sel; asm volatile("v_lshrrev_b64 %0, %1, %0" : "+v"(r) : "v"(a)); x[i] = (uint32_t)r; }
            if (OP == 17) asm volatile("v_alignbit_b32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(sel), "v"(a));
            if (OP == 18) asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(x[i]) : "v"(sel) : "vcc");
            if (OP == 19) asm volatile("v_lshlrev_b32 %0, %1, %0" : "+v"(x[i]) : "v"(a));
            if (OP == 20) asm volatile("v_bfe_u32 %0, %0, %1, 4" : "+v"(x[i]) : "v"(a));
        }
    }
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s ^= x[i];
    if (s == 0x12345u) out[threadIdx.x] = s;
}

template <int OP>
void run(const char *name, uint32_t *d)
{
    const int blocks = 256 * 8; // 8 blocks of 256 threads per CU = 8 waves/SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 3u, 0u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 3u, 0u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // per SIMD: 8 waves x ITER x 8 instructions
    double instr = 8.0 * ITER * 8.0;
    double ns_per = ms * 1e6 / instr;
    printf("%-16s %8.3f ms  %.3f ns per wave64-instruction per SIMD  (= %.2f cycles @2.4GHz, %.2f @2.1GHz)\n", name, ms,
           ns_per, ns_per * 2.4, ns_per * 2.1);
}

int main()
{
    uint32_t *d; hipMalloc(&d, 4096);
    run<0>("v_xor_b32", d); run<1>("v_add_u32", d); run<13>("v_sub_u32", d); run<2>("v_perm_b32", d);
    run<3>("v_bitop3_b32", d); run<9>("v_and_or_b32", d); run<14>("v_lshl_add_u32", d); run<6>("v_lshrrev_b32", d);
    run<7>("v_bcnt_u32_b32", d); run<11>("v_cndmask_b32", d); run<4>("v_mul_hi_u32", d); run<8>("v_mul_lo_u32", d);
    run<5>("v_mad_u64_u32", d); run<12>("v_pk_add_u16", d); run<10>("v_fma_f32", d);
    run<15>("v_lshlrev_b64", d); run<16>("v_lshrrev_b64", d); run<17>("v_alignbit_b32", d); run<18>("v_cndmask_e64", d);
    run<19>("v_lshlrev_b32 (var)", d); run<20>("v_bfe_u32", d);
    return 0;
}
