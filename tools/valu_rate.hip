// valu_rate — issue rate of a few VALU instructions on one MI355X CU-SIMD (wave64): how many cycles does a
// SIMD spend per wave instruction? 4 = full rate. Used to price the byte -> float conversion of the 8-bit
// node boxes (v_cvt_f32_ubyteN vs v_perm_b32 + bias trick).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int OP>
__global__ __launch_bounds__(256) void k(float* out, int iters, uint32_t seed) {
    uint32_t x[8]; float f[8]; unsigned long long mask = 0x5555555555555555ull * seed; unsigned long long m2[4] = {mask, mask, mask, mask};
    for (int i = 0; i < 8; ++i) { x[i] = seed * (threadIdx.x + 1u + i); f[i] = (float)(i + 1) + seed; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(f[i]) : "v"(f[(i + 1) & 7]));
                if (OP == 1) asm volatile("v_cvt_f32_ubyte1 %0, %1" : "=v"(f[i]) : "v"(x[i]));
                if (OP == 2) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(x[i]) : "v"(x[(i + 1) & 7]), "v"(x[(i + 2) & 7]), "s"(0x07060501u));
                if (OP == 3) asm volatile("v_cvt_f32_u32 %0, %1" : "=v"(f[i]) : "v"(x[i]));
                if (OP == 4) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(f[i]) : "v"(f[(i + 1) & 7]), "v"(f[(i + 2) & 7]));
                if (OP == 5) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(f[i]) : "v"(f[(i + 1) & 7]), "v"(f[(i + 2) & 7]));
                if (OP == 6) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(*(double*)&f[i & 6]) : "v"(*(double*)&f[(i + 2) & 6]));
                if (OP == 7) asm volatile("v_bfe_u32 %0, %1, 8, 8" : "=v"(x[i]) : "v"(x[(i + 1) & 7]));
                if (OP == 8) asm volatile("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(f[i]) : "v"(f[(i + 1) & 7]), "v"(f[(i + 2) & 7]), "s"(mask));
                if (OP == 9) asm volatile("v_cmp_lt_f32_e64 %0, %1, %2" : "=s"(m2[i & 3]) : "v"(f[(i + 1) & 7]), "v"(f[(i + 2) & 7]));
                if (OP == 10) { asm volatile("v_cmp_lt_f32_e64 %0, %1, %2" : "=s"(m2[i & 3]) : "v"(f[(i + 1) & 7]), "v"(f[(i + 2) & 7]));
                                asm volatile("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(f[i]) : "v"(f[(i + 3) & 7]), "v"(f[(i + 4) & 7]), "s"(m2[(i + 2) & 3])); }
                if (OP == 14) asm volatile("v_cmp_lt_f32_e32 vcc, %1, %2\n s_nop 1\n v_cndmask_b32_e32 %0, %3, %4, vcc" : "=v"(f[i]) : "v"(f[(i + 1) & 7]), "v"(f[(i + 2) & 7]), "v"(f[(i + 3) & 7]), "v"(f[(i + 4) & 7]) : "vcc");
                if (OP == 15) asm volatile("v_cmp_lt_f32_e32 vcc, %1, %2\n v_mov_b32 %0, %3\n v_mov_b32 %0, %4\n v_cndmask_b32_e32 %0, %3, %4, vcc" : "=&v"(f[i]) : "v"(f[(i + 1) & 7]), "v"(f[(i + 2) & 7]), "v"(f[(i + 3) & 7]), "v"(f[(i + 4) & 7]) : "vcc");
                if (OP == 11) asm volatile("v_min_f32 %0, %1, %2" : "=v"(f[i]) : "v"(f[(i + 1) & 7]), "v"(f[(i + 2) & 7]));
                if (OP == 12) asm volatile("v_sub_f32 %0, %1, %2" : "=v"(f[i]) : "v"(f[(i + 1) & 7]), "v"(f[(i + 2) & 7]));
                if (OP == 13) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(f[i]) : "v"(f[(i + 1) & 7]), "v"(f[(i + 2) & 7]));
                if (OP == 16) asm volatile("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(f[i]) : "v"(x[(i + 1) & 7]), "v"(f[(i + 2) & 7]), "v"(f[(i + 3) & 7]));
            }
        }
    }
    float s = (float)(m2[0] ^ m2[1] ^ m2[2] ^ m2[3]); for (int i = 0; i < 8; ++i) s += f[i] + (float)x[i];
    if (s == 123.456f) out[threadIdx.x] = s;
}

int main() {
    float* out; CHECK(hipMalloc((void**)&out, 4096));
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount; const double ghz = prop.clockRate * 1e-6;
    const int iters = 4096, blocks = cus * 8;     // 8 blocks x 4 waves per CU = 8 waves per SIMD
    const char* names[] = {"v_fma_f32", "v_cvt_f32_ubyte1", "v_perm_b32", "v_cvt_f32_u32", "v_max3_f32", "v_cndmask_b32 vcc (vcc never written: artefact)", "v_pk_fma_f32", "v_bfe_u32", "v_cndmask_e64 sgpr", "v_cmp_e64", "v_cmp+v_cndmask", "v_min_f32", "v_sub_f32", "v_mul_f32", "cmp_e32;nop;cndmask_e32 (3 instr)", "cmp_e32;mov;mov;cndmask_e32 (4 instr)", "v_fma_mix_f32 (f16 hi, f32, f32)"};
    hipEvent_t t0, t1; CHECK(hipEventCreate(&t0)); CHECK(hipEventCreate(&t1));
    for (int op = 0; op < 17; ++op) {
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            CHECK(hipEventRecord(t0, 0));
            switch (op) {
            case 0: hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, out, iters, 3u); break;
            case 1: hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, out, iters, 3u); break;
            case 2: hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, out, iters, 3u); break;
            case 3: hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), 0, 0, out, iters, 3u); break;
            case 4: hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(256), 0, 0, out, iters, 3u); break;
            case 5: hipLaunchKernelGGL(k<5>, dim3(blocks), dim3(256), 0, 0, out, iters, 3u); break;
            case 6: hipLaunchKernelGGL(k<6>, dim3(blocks), dim3(256), 0, 0, out, iters, 3u); break;
            case 7: hipLaunchKernelGGL(k<7>, dim3(blocks), dim3(256), 0, 0, out, iters, 3u); break;
            case 8: hipLaunchKernelGGL(k<8>, dim3(blocks), dim3(256), 0, 0, out, iters, 3u); break;
            case 9: hipLaunchKernelGGL(k<9>, dim3(blocks), dim3(256), 0, 0, out, iters, 3u); break;
            case 10: hipLaunchKernelGGL(k<10>, dim3(blocks), dim3(256), 0, 0, out, iters, 3u); break;
            case 11: hipLaunchKernelGGL(k<11>, dim3(blocks), dim3(256), 0, 0, out, iters, 3u); break;
            case 12: hipLaunchKernelGGL(k<12>, dim3(blocks), dim3(256), 0, 0, out, iters, 3u); break;
            case 13: hipLaunchKernelGGL(k<13>, dim3(blocks), dim3(256), 0, 0, out, iters, 3u); break;
            case 14: hipLaunchKernelGGL(k<14>, dim3(blocks), dim3(256), 0, 0, out, iters, 3u); break;
            case 15: hipLaunchKernelGGL(k<15>, dim3(blocks), dim3(256), 0, 0, out, iters, 3u); break;
            case 16: hipLaunchKernelGGL(k<16>, dim3(blocks), dim3(256), 0, 0, out, iters, 3u); break;
            }
            CHECK(hipEventRecord(t1, 0)); CHECK(hipEventSynchronize(t1));
            float ms; CHECK(hipEventElapsedTime(&ms, t0, t1)); if (rep && ms < best) best = ms;
        }
        const double waveInstr = (double)blocks * 4.0 * iters * 64.0;     // per launch
        const double perSimd = waveInstr / (cus * 4.0);
        printf("%-18s %.3f ms -> %.2f cycles per wave instruction per SIMD\n", names[op], best, best * 1e-3 * ghz * 1e9 / perSimd);
    }
    return 0;
}
