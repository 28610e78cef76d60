// Micro-benchmark: issue cost of a few VALU instructions on gfx950, W waves per SIMD.
// cycles per wave-instruction per SIMD = elapsed shader cycles / (instructions per wave * waves per SIMD)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define REP16(x) x x x x x x x x x x x x x x x x

template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* cyc, int iters, float seed) {
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float b = 1.000001f, c = 0.5f;
    unsigned m0 = threadIdx.x, m1 = 1;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) {
        if (KIND == 0) {   // 16 x 8 independent v_fma_f32 (VOP3, all VGPR)
            REP16(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                               "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));)
        } else if (KIND == 1) {   // v_fmac_f32 (VOP2)
            REP16(asm volatile("v_fmac_f32 %0, %8, %9\n v_fmac_f32 %1, %8, %9\n v_fmac_f32 %2, %8, %9\n v_fmac_f32 %3, %8, %9\n"
                               "v_fmac_f32 %4, %8, %9\n v_fmac_f32 %5, %8, %9\n v_fmac_f32 %6, %8, %9\n v_fmac_f32 %7, %8, %9\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));)
        } else if (KIND == 2) {   // v_rsq_f32
            REP16(asm volatile("v_rsq_f32 %0, %0\n v_rsq_f32 %1, %1\n v_rsq_f32 %2, %2\n v_rsq_f32 %3, %3\n"
                               "v_rsq_f32 %4, %4\n v_rsq_f32 %5, %5\n v_rsq_f32 %6, %6\n v_rsq_f32 %7, %7\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        } else if (KIND == 3) {   // dependent chain of v_fma_f32 (latency)
            REP16(asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                               "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                               : "+v"(a0) : "v"(b), "v"(c));)
        } else if (KIND == 4) {   // v_pk_fma_f32 (two fp32 FMAs per lane)
            REP16(asm volatile("v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n v_pk_fma_f32 %2, %2, %4, %4\n v_pk_fma_f32 %3, %3, %4, %4\n"
                               "v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n v_pk_fma_f32 %2, %2, %4, %4\n v_pk_fma_f32 %3, %3, %4, %4\n"
                               : "+v"(*(double*)&a0), "+v"(*(double*)&a2), "+v"(*(double*)&a4), "+v"(*(double*)&a6) : "v"(*(double*)&b));)
        } else if (KIND == 5) {   // v_cmp_ge_f32 (to SGPR pair) + v_addc_co_u32 pairs
            REP16(asm volatile("v_cmp_ge_f32 s[20:21], %2, %3\n v_addc_co_u32 %0, s[22:23], %0, %0, s[20:21]\n"
                               "v_cmp_ge_f32 s[20:21], %3, %2\n v_addc_co_u32 %1, s[22:23], %1, %1, s[20:21]\n"
                               "v_cmp_ge_f32 s[20:21], %2, %3\n v_addc_co_u32 %0, s[22:23], %0, %0, s[20:21]\n"
                               "v_cmp_ge_f32 s[20:21], %3, %2\n v_addc_co_u32 %1, s[22:23], %1, %1, s[20:21]\n"
                               : "+v"(m0), "+v"(m1) : "v"(a0), "v"(a1) : "s20", "s21", "s22", "s23");)
        } else if (KIND == 6) {   // v_sub_f32 / v_max_f32 / v_mul_f32 mix (VOP2)
            REP16(asm volatile("v_sub_f32 %0, %8, %0\n v_max_f32 %1, %9, %1\n v_mul_f32 %2, %8, %2\n v_sub_f32 %3, %8, %3\n"
                               "v_max_f32 %4, %9, %4\n v_mul_f32 %5, %8, %5\n v_sub_f32 %6, %8, %6\n v_max_f32 %7, %9, %7\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));)
        } else if (KIND == 7) {   // v_fma_f32 with an SGPR operand
            REP16(asm volatile("v_fma_f32 %0, -%0, %8, s30\n v_fma_f32 %1, -%1, %8, s30\n v_fma_f32 %2, -%2, %8, s30\n v_fma_f32 %3, -%3, %8, s30\n"
                               "v_fma_f32 %4, -%4, %8, s30\n v_fma_f32 %5, -%5, %8, s30\n v_fma_f32 %6, -%6, %8, s30\n v_fma_f32 %7, -%7, %8, s30\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));)
        } else if (KIND == 8) {   // v_pk_fma_f16 (two fp16 FMAs per lane): the mixed-precision density kernel's workhorse
            REP16(asm volatile("v_pk_fma_f16 %0, %0, %8, %9\n v_pk_fma_f16 %1, %1, %8, %9\n v_pk_fma_f16 %2, %2, %8, %9\n v_pk_fma_f16 %3, %3, %8, %9\n"
                               "v_pk_fma_f16 %4, %4, %8, %9\n v_pk_fma_f16 %5, %5, %8, %9\n v_pk_fma_f16 %6, %6, %8, %9\n v_pk_fma_f16 %7, %7, %8, %9\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));)
        } else if (KIND == 9) {   // v_pk_add_f16 / v_pk_max_f16 / v_pk_mul_f16 mix
            REP16(asm volatile("v_pk_add_f16 %0, %8, %0\n v_pk_max_f16 %1, %9, %1\n v_pk_mul_f16 %2, %8, %2\n v_pk_add_f16 %3, %8, %3\n"
                               "v_pk_max_f16 %4, %9, %4\n v_pk_mul_f16 %5, %8, %5\n v_pk_add_f16 %6, %8, %6\n v_pk_max_f16 %7, %9, %7\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));)
        } else if (KIND == 10) {  // v_fma_f16 (one fp16 FMA per lane, VOP3)
            REP16(asm volatile("v_fma_f16 %0, %0, %8, %9\n v_fma_f16 %1, %1, %8, %9\n v_fma_f16 %2, %2, %8, %9\n v_fma_f16 %3, %3, %8, %9\n"
                               "v_fma_f16 %4, %4, %8, %9\n v_fma_f16 %5, %5, %8, %9\n v_fma_f16 %6, %6, %8, %9\n v_fma_f16 %7, %7, %8, %9\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));)
        } else if (KIND == 11) {  // v_fma_mix_f32: fp16 sources (low halves) read in place, fp32 accumulator
            REP16(asm volatile("v_fma_mix_f32 %0, %8, %9, %0 op_sel_hi:[1,1,0]\n v_fma_mix_f32 %1, %8, %9, %1 op_sel_hi:[1,1,0]\n"
                               "v_fma_mix_f32 %2, %8, %9, %2 op_sel_hi:[1,1,0]\n v_fma_mix_f32 %3, %8, %9, %3 op_sel_hi:[1,1,0]\n"
                               "v_fma_mix_f32 %4, %8, %9, %4 op_sel_hi:[1,1,0]\n v_fma_mix_f32 %5, %8, %9, %5 op_sel_hi:[1,1,0]\n"
                               "v_fma_mix_f32 %6, %8, %9, %6 op_sel_hi:[1,1,0]\n v_fma_mix_f32 %7, %8, %9, %7 op_sel_hi:[1,1,0]\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));)
        } else if (KIND == 12) {  // v_fma_mix_f32 with one fp16 source (high half) and two fp32 sources
            REP16(asm volatile("v_fma_mix_f32 %0, %8, %9, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 %1, %8, %9, %1 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n"
                               "v_fma_mix_f32 %2, %8, %9, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 %3, %8, %9, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n"
                               "v_fma_mix_f32 %4, %8, %9, %4 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 %5, %8, %9, %5 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n"
                               "v_fma_mix_f32 %6, %8, %9, %6 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 %7, %8, %9, %7 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));)
        } else if (KIND == 14) {  // v_fmaak_f32: fma with a 32-bit literal addend (what hipcc picks for fmaf(a, a, 1e-30f))
            REP16(asm volatile("v_fmaak_f32 %0, %8, %8, 0xda24260\n v_fmaak_f32 %1, %9, %9, 0xda24260\n v_fmaak_f32 %2, %8, %9, 0xda24260\n v_fmaak_f32 %3, %9, %8, 0xda24260\n"
                               "v_fmaak_f32 %4, %8, %8, 0xda24260\n v_fmaak_f32 %5, %9, %9, 0xda24260\n v_fmaak_f32 %6, %8, %9, 0xda24260\n v_fmaak_f32 %7, %9, %8, 0xda24260\n"
                               : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3), "=v"(a4), "=v"(a5), "=v"(a6), "=v"(a7) : "v"(b), "v"(c));)
        } else if (KIND == 15) {  // v_alignbit_b32 (the collision mask's shift-in)
            REP16(asm volatile("v_alignbit_b32 %0, %0, %8, 31\n v_alignbit_b32 %1, %1, %9, 31\n v_alignbit_b32 %2, %2, %8, 31\n v_alignbit_b32 %3, %3, %9, 31\n"
                               "v_alignbit_b32 %4, %4, %8, 31\n v_alignbit_b32 %5, %5, %9, 31\n v_alignbit_b32 %6, %6, %8, 31\n v_alignbit_b32 %7, %7, %9, 31\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));)
        } else if (KIND == 16) {  // v_alignbit_b32 with the shift in a VGPR
            REP16(asm volatile("v_alignbit_b32 %0, %0, %8, %9\n v_alignbit_b32 %1, %1, %8, %9\n v_alignbit_b32 %2, %2, %8, %9\n v_alignbit_b32 %3, %3, %8, %9\n"
                               "v_alignbit_b32 %4, %4, %8, %9\n v_alignbit_b32 %5, %5, %8, %9\n v_alignbit_b32 %6, %6, %8, %9\n v_alignbit_b32 %7, %7, %8, %9\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(m1));)
        } else if (KIND == 17) {  // v_lshl_or_b32 v, v, 1, v  (near << 1 | bit)
            REP16(asm volatile("v_lshl_or_b32 %0, %0, 1, %8\n v_lshl_or_b32 %1, %1, 1, %8\n v_lshl_or_b32 %2, %2, 1, %8\n v_lshl_or_b32 %3, %3, 1, %8\n"
                               "v_lshl_or_b32 %4, %4, 1, %8\n v_lshl_or_b32 %5, %5, 1, %8\n v_lshl_or_b32 %6, %6, 1, %8\n v_lshl_or_b32 %7, %7, 1, %8\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));)
        } else if (KIND == 18) {  // v_lshl_or_b32 with the shift in a VGPR
            REP16(asm volatile("v_lshl_or_b32 %0, %0, %9, %8\n v_lshl_or_b32 %1, %1, %9, %8\n v_lshl_or_b32 %2, %2, %9, %8\n v_lshl_or_b32 %3, %3, %9, %8\n"
                               "v_lshl_or_b32 %4, %4, %9, %8\n v_lshl_or_b32 %5, %5, %9, %8\n v_lshl_or_b32 %6, %6, %9, %8\n v_lshl_or_b32 %7, %7, %9, %8\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(m1));)
        } else if (KIND == 19) {  // v_max_f32 with inline constant 0 (VOP2) and v_fma_f32 with an inline constant (VOP3)
            REP16(asm volatile("v_max_f32 %0, 0, %0\n v_fma_f32 %1, %1, %8, 1.0\n v_max_f32 %2, 0, %2\n v_fma_f32 %3, %3, %8, 1.0\n"
                               "v_max_f32 %4, 0, %4\n v_fma_f32 %5, %5, %8, 1.0\n v_max_f32 %6, 0, %6\n v_fma_f32 %7, %7, %8, 1.0\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));)
        } else if (KIND == 13) {  // v_cvt_f32_f16 (what a plain conversion of a staged half would cost)
            REP16(asm volatile("v_cvt_f32_f16 %0, %8\n v_cvt_f32_f16 %1, %9\n v_cvt_f32_f16 %2, %8\n v_cvt_f32_f16 %3, %9\n"
                               "v_cvt_f32_f16 %4, %8\n v_cvt_f32_f16 %5, %9\n v_cvt_f32_f16 %6, %8\n v_cvt_f32_f16 %7, %9\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));)
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(m0 + m1);
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int KIND>
void run(const char* name, int per_iter) {
    const int iters = 2000;
    for (int w : {1, 2, 3, 4, 6, 8}) {
        const int blocks = 256 * w;          // 256 CUs x w blocks of 4 waves: w waves per SIMD
        float* out; unsigned long long* cyc;
        hipMalloc(&out, blocks * 256 * sizeof(float));
        hipMalloc(&cyc, blocks * 4 * sizeof(unsigned long long));
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, cyc, 10, 1.0f);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters, 1.0f);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(blocks * 4);
        hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
        double mean = 0; for (auto v : h) mean += (double)v; mean /= h.size();
        const double instr_per_wave = (double)iters * per_iter;
        printf("%-34s waves/SIMD %d : %6.2f counter-ticks per wave-instr per wave, %6.2f per SIMD ; wall %.3f ms -> %.2f ns/instr/SIMD\n",
               name, w, mean / instr_per_wave, mean / instr_per_wave / w, ms, ms * 1e6 / (instr_per_wave * w));
        hipFree(out); hipFree(cyc);
    }
}

int main(int argc, char**) {
    if (argc > 1) {            // any argument: only the kinds added last
        run<14>("v_fmaak_f32 (32-bit literal)", 128);
        run<15>("v_alignbit_b32 v, v, v, 31", 128);
        run<16>("v_alignbit_b32 v, v, v, v", 128);
        run<17>("v_lshl_or_b32 v, v, 1, v", 128);
        run<18>("v_lshl_or_b32 v, v, v, v", 128);
        run<19>("v_max 0 / v_fma inline 1.0 mix", 128);
        return 0;
    }
    run<0>("v_fma_f32 (VOP3) independent", 128);
    run<1>("v_fmac_f32 (VOP2) independent", 128);
    run<6>("v_sub/v_max/v_mul (VOP2)", 128);
    run<7>("v_fma_f32 -v, v, SGPR", 128);
    run<2>("v_rsq_f32 independent", 128);
    run<3>("v_fma_f32 dependent chain", 128);
    run<4>("v_pk_fma_f32 independent", 128);
    run<5>("v_cmp_ge_f32 + v_addc_co_u32", 128);
    run<8>("v_pk_fma_f16 independent", 128);
    run<9>("v_pk_add/max/mul_f16 mix", 128);
    run<10>("v_fma_f16 independent", 128);
    run<11>("v_fma_mix_f32 (2 fp16 srcs)", 128);
    run<12>("v_fma_mix_f32 (1 fp16 hi src)", 128);
    run<13>("v_cvt_f32_f16", 128);
    run<14>("v_fmaak_f32 (32-bit literal)", 128);
    run<15>("v_alignbit_b32 v, v, v, 31", 128);
    run<16>("v_alignbit_b32 v, v, v, v", 128);
    run<17>("v_lshl_or_b32 v, v, 1, v", 128);
    run<18>("v_lshl_or_b32 v, v, v, v", 128);
    run<19>("v_max 0 / v_fma inline 1.0 mix", 128);
    return 0;
}
