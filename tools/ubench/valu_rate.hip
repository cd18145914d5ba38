// Microbenchmark: VALU issue rate of integer ops on gfx950 (is a wave64 v_and/v_alignbit 2 or 4 cycles?)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void k_valu(unsigned* out, int iters, unsigned seed) {
    unsigned a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            a0 = __builtin_amdgcn_alignbit(a1, a0, 3); a1 ^= a2; a2 = __builtin_amdgcn_alignbit(a3, a2, 5); a3 &= a4 | 0x55555555u;
            a4 = __builtin_amdgcn_alignbit(a5, a4, 7); a5 ^= a6; a6 = __builtin_amdgcn_alignbit(a7, a6, 9); a7 |= a0 & 0x33333333u;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
__global__ void k_salu_mix(unsigned* out, int iters, unsigned seed) {
    unsigned a0 = threadIdx.x + seed, a1 = a0 * 3;
    unsigned s = seed;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            a0 = __builtin_amdgcn_alignbit(a1, a0, 3); a1 ^= a0;
            s = s * 5 + 1; s ^= s >> 3;            // scalar work interleaved
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ s;
}
int main() {
    unsigned* d; hipMalloc(&d, 256 * 8 * 256 * 4 * sizeof(unsigned));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int iters = 20000;
    for (int wps = 1; wps <= 8; wps *= 2) {     // waves per SIMD
        int blocks = 256 * wps;                  // 256-thread blocks: 1 wave per SIMD per block
        hipLaunchKernelGGL(k_valu, dim3(blocks), dim3(256), 0, 0, d, 100, 1u);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_valu, dim3(blocks), dim3(256), 0, 0, d, iters, 1u);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double instr_per_wave = (double)iters * 64;   // 32 alignbit + 16 xor + 8 and_or + 8 bitop3 per iteration (checked in the .s)
        double waves_per_simd = wps;
        double cyc = ms * 1e-3 * 2.4e9;
        printf("valu: waves/SIMD=%d  ms=%.3f  cycles@2.4GHz per VALU instr per SIMD = %.2f\n", wps, ms, cyc / (instr_per_wave * waves_per_simd));
    }
    for (int wps = 1; wps <= 8; wps *= 2) {
        int blocks = 256 * wps;
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_salu_mix, dim3(blocks), dim3(256), 0, 0, d, iters, 1u);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("mix : waves/SIMD=%d  ms=%.3f  per iteration-unit (2 VALU + ~3 SALU) cycles per SIMD = %.2f\n", wps, ms, ms * 1e-3 * 2.4e9 / ((double)iters * 16 * wps));
    }
    return 0;
}
