// Microbenchmark (round 2): issue cost of individual VALU instructions on gfx950 at 8 waves/SIMD, in NANOSECONDS
// per wave-instruction per SIMD (clock-independent) and in shader cycles at the clock measured in-kernel
// (s_memtime / s_memrealtime).  Inline asm, so the compiler cannot fold or pack anything.
// Answers: are plain v_and/v_xor/v_lshl half-rate like v_alignbit, and what does FP32 do?
#include <hip/hip_runtime.h>
#include <cstdio>
#define OP3(I) asm volatile(I " %0, %1, %2" : "=v"(a) : "v"(b), "v"(c)); asm volatile(I " %0, %1, %2" : "=v"(b) : "v"(c), "v"(d)); \
               asm volatile(I " %0, %1, %2" : "=v"(c) : "v"(d), "v"(e)); asm volatile(I " %0, %1, %2" : "=v"(d) : "v"(e), "v"(f)); \
               asm volatile(I " %0, %1, %2" : "=v"(e) : "v"(f), "v"(g)); asm volatile(I " %0, %1, %2" : "=v"(f) : "v"(g), "v"(h)); \
               asm volatile(I " %0, %1, %2" : "=v"(g) : "v"(h), "v"(a)); asm volatile(I " %0, %1, %2" : "=v"(h) : "v"(a), "v"(b));
#define OP4(I, SUF) asm volatile(I " %0, %1, %2, %3" SUF : "=v"(a) : "v"(b), "v"(c), "v"(d)); asm volatile(I " %0, %1, %2, %3" SUF : "=v"(b) : "v"(c), "v"(d), "v"(e)); \
               asm volatile(I " %0, %1, %2, %3" SUF : "=v"(c) : "v"(d), "v"(e), "v"(f)); asm volatile(I " %0, %1, %2, %3" SUF : "=v"(d) : "v"(e), "v"(f), "v"(g)); \
               asm volatile(I " %0, %1, %2, %3" SUF : "=v"(e) : "v"(f), "v"(g), "v"(h)); asm volatile(I " %0, %1, %2, %3" SUF : "=v"(f) : "v"(g), "v"(h), "v"(a)); \
               asm volatile(I " %0, %1, %2, %3" SUF : "=v"(g) : "v"(h), "v"(a), "v"(b)); asm volatile(I " %0, %1, %2, %3" SUF : "=v"(h) : "v"(a), "v"(b), "v"(c));
#define DEF(NAME, BODY)                                                                      \
    __global__ void NAME(unsigned* out, unsigned long long* clk, int iters, unsigned seed) { \
        unsigned a = threadIdx.x + seed, b = a * 3, c = a * 5, d = a * 7, e = a * 11, f = a * 13, g = a * 17, h = a * 19;   \
        unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();   \
        for (int i = 0; i < iters; ++i) {                                                    \
            _Pragma("unroll") for (int u = 0; u < 8; ++u) { BODY }                           \
        }                                                                                    \
        unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();   \
        if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }     \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d + e + f + g + h;          \
    }
DEF(k_and, OP3("v_and_b32"))
DEF(k_xor, OP3("v_xor_b32"))
DEF(k_or, OP3("v_or_b32"))
DEF(k_shl, OP3("v_lshlrev_b32"))
DEF(k_add, OP3("v_add_u32"))
DEF(k_mul24, OP3("v_mul_u32_u24"))
DEF(k_bcnt, OP3("v_bcnt_u32_b32"))
DEF(k_fmul, OP3("v_mul_f32"))
DEF(k_alignbit, OP4("v_alignbit_b32", ""))
DEF(k_dot4, OP4("v_dot4_u32_u8", ""))
DEF(k_perm, OP4("v_perm_b32", ""))
DEF(k_bitop3, OP4("v_bitop3_b32", " bitop3:0x96"))
DEF(k_andor, OP4("v_and_or_b32", ""))
DEF(k_lshlor, OP4("v_lshl_or_b32", ""))
DEF(k_bfe, OP4("v_bfe_u32", ""))
DEF(k_fma, OP4("v_fma_f32", ""))
DEF(k_sad, OP4("v_sad_u8", ""))
DEF(k_msad, OP4("v_msad_u8", ""))
DEF(k_lerp, OP4("v_lerp_u8", ""))
typedef void (*kern_t)(unsigned*, unsigned long long*, int, unsigned);
int main() {
    unsigned* d; (void)hipMalloc(&d, 256 * 8 * 256 * sizeof(unsigned));
    unsigned long long* clk; (void)hipMalloc(&clk, 16);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    struct { const char* name; kern_t k; } tests[] = {{"v_and_b32", k_and}, {"v_xor_b32", k_xor}, {"v_or_b32", k_or}, {"v_lshlrev_b32", k_shl}, {"v_add_u32", k_add},
        {"v_mul_u32_u24", k_mul24}, {"v_bcnt_u32_b32", k_bcnt}, {"v_mul_f32", k_fmul}, {"v_alignbit_b32", k_alignbit}, {"v_dot4_u32_u8", k_dot4}, {"v_perm_b32", k_perm},
        {"v_bitop3_b32", k_bitop3}, {"v_and_or_b32", k_andor}, {"v_lshl_or_b32", k_lshlor}, {"v_bfe_u32", k_bfe}, {"v_fma_f32", k_fma}, {"v_sad_u8", k_sad}, {"v_msad_u8", k_msad}, {"v_lerp_u8", k_lerp}};
    const int iters = 20000, blocks_per_wps = 256;
    for (int wps : {8, 2, 1})
    for (auto& t : tests) {
        const int blocks = blocks_per_wps * wps;
        hipLaunchKernelGGL(t.k, dim3(blocks), dim3(256), 0, 0, d, clk, 2000, 1u);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(t.k, dim3(blocks), dim3(256), 0, 0, d, clk, iters, 1u);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h[2]; (void)hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
        double ghz = (double)h[0] / (double)h[1] * 0.1;      // s_memrealtime ticks at 100 MHz
        double ns = ms * 1e6 / ((double)iters * 64 * wps);
        printf("waves/SIMD=%d %-16s %.3f ns per instruction per SIMD = %.2f cycles at the measured %.2f GHz\n", wps, t.name, ns, ns * ghz, ghz);
    }
    return 0;
}
