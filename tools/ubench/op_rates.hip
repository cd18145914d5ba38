// Microbenchmark: issue cost (cycles per wave64 instruction per SIMD, 8 waves/SIMD) of the integer
// ops the scan kernels are built from, on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#define DEF(NAME, BODY)                                                                   \
    __global__ void NAME(unsigned* out, int iters, unsigned seed) {                        \
        unsigned a = threadIdx.x + seed, b = a * 3 + 1, c = a * 5 + 2, d = a * 7 + 3;      \
        unsigned e = a * 11 + 4, f = a * 13 + 5, g = a * 17 + 6, h = a * 19 + 7;           \
        for (int i = 0; i < iters; ++i) {                                                  \
            _Pragma("unroll") for (int u = 0; u < 8; ++u) { BODY }                         \
        }                                                                                  \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d ^ e ^ f ^ g ^ h;        \
    }
// every BODY = 8 instructions of the op under test on independent chains
DEF(k_alignbit, a = __builtin_amdgcn_alignbit(b, a, 3); b = __builtin_amdgcn_alignbit(c, b, 5); c = __builtin_amdgcn_alignbit(d, c, 7); d = __builtin_amdgcn_alignbit(e, d, 9);
                e = __builtin_amdgcn_alignbit(f, e, 11); f = __builtin_amdgcn_alignbit(g, f, 13); g = __builtin_amdgcn_alignbit(h, g, 15); h = __builtin_amdgcn_alignbit(a, h, 17);)
DEF(k_mul_lo, a *= b | 1; b *= c | 1; c *= d | 1; d *= e | 1; e *= f | 1; f *= g | 1; g *= h | 1; h *= a | 1;)
DEF(k_mul24, a = ((a & 0xFFFFFFu) * (b & 0xFFFFFFu)); b = ((b & 0xFFFFFFu) * (c & 0xFFFFFFu)); c = ((c & 0xFFFFFFu) * (d & 0xFFFFFFu)); d = ((d & 0xFFFFFFu) * (e & 0xFFFFFFu));
             e = ((e & 0xFFFFFFu) * (f & 0xFFFFFFu)); f = ((f & 0xFFFFFFu) * (g & 0xFFFFFFu)); g = ((g & 0xFFFFFFu) * (h & 0xFFFFFFu)); h = ((h & 0xFFFFFFu) * (a & 0xFFFFFFu));)
DEF(k_dot4, a = __builtin_amdgcn_udot4(b, c, a, false); b = __builtin_amdgcn_udot4(c, d, b, false); c = __builtin_amdgcn_udot4(d, e, c, false); d = __builtin_amdgcn_udot4(e, f, d, false);
            e = __builtin_amdgcn_udot4(f, g, e, false); f = __builtin_amdgcn_udot4(g, h, f, false); g = __builtin_amdgcn_udot4(h, a, g, false); h = __builtin_amdgcn_udot4(a, b, h, false);)
DEF(k_perm, a = __builtin_amdgcn_perm(b, a, c); b = __builtin_amdgcn_perm(c, b, d); c = __builtin_amdgcn_perm(d, c, e); d = __builtin_amdgcn_perm(e, d, f);
            e = __builtin_amdgcn_perm(f, e, g); f = __builtin_amdgcn_perm(g, f, h); g = __builtin_amdgcn_perm(h, g, a); h = __builtin_amdgcn_perm(a, h, b);)
DEF(k_popc, a = __popc(b) + a; b = __popc(c) + b; c = __popc(d) + c; d = __popc(e) + d; e = __popc(f) + e; f = __popc(g) + f; g = __popc(h) + g; h = __popc(a) + h;)
DEF(k_bitop3, a = (a & b) ^ (c | a); b = (b & c) ^ (d | b); c = (c & d) ^ (e | c); d = (d & e) ^ (f | d); e = (e & f) ^ (g | e); f = (f & g) ^ (h | f); g = (g & h) ^ (a | g); h = (h & a) ^ (b | h);)
DEF(k_ffs, a += __ffs((int)b); b += __ffs((int)c); c += __ffs((int)d); d += __ffs((int)e); e += __ffs((int)f); f += __ffs((int)g); g += __ffs((int)h); h += __ffs((int)a);)
DEF(k_brev, a = __brev(b) ^ a; b = __brev(c) ^ b; c = __brev(d) ^ c; d = __brev(e) ^ d; e = __brev(f) ^ e; f = __brev(g) ^ f; g = __brev(h) ^ g; h = __brev(a) ^ h;)
typedef void (*kern_t)(unsigned*, int, unsigned);
int main() {
    unsigned* d; (void)hipMalloc(&d, 256 * 8 * 256 * sizeof(unsigned));
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    struct { const char* name; kern_t k; double per_body; } tests[] = {
        {"v_alignbit_b32", k_alignbit, 8}, {"v_mul_lo_u32", k_mul_lo, 8}, {"v_mul_u32_u24", k_mul24, 8}, {"v_dot4_u32_u8", k_dot4, 8},
        {"v_perm_b32", k_perm, 8}, {"v_bcnt+add (popc)", k_popc, 8}, {"v_bitop3 (3-input)", k_bitop3, 8}, {"v_ffbl+add (ffs)", k_ffs, 8}, {"v_bfrev+xor", k_brev, 8}};
    const int iters = 10000, wps = 8, blocks = 256 * wps;
    for (auto& t : tests) {
        hipLaunchKernelGGL(t.k, dim3(blocks), dim3(256), 0, 0, d, 10, 1u);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(t.k, dim3(blocks), dim3(256), 0, 0, d, iters, 1u);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        double per = ms * 1e-3 * 2.4e9 / ((double)iters * 8 * t.per_body * wps);
        printf("%-22s %.2f cycles@2.4GHz per source-level op per SIMD (check the .s for how many instructions that is)\n", t.name, per);
    }
    return 0;
}
