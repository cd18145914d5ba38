// zero_copy.hip -- how fast can a kernel pull pinned host memory over the link, against the DMA engines?
// hipcc --offload-arch=gfx950 -O3 tools/ubench/zero_copy.hip -o /tmp/zc && /tmp/zc
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

struct Segs { int n; const char* src[64]; char* dst[64]; unsigned bytes[64]; };

__global__ __launch_bounds__(256) void gather(Segs S) {
    // blocks are dealt round-robin to segments' 64 KB tiles
    const unsigned tile = 65536;
    unsigned t = blockIdx.x;
    for (;;) {
        // find the tile t
        unsigned k = t; int s = 0; bool found = false;
        for (; s < S.n; ++s) { const unsigned nt = (S.bytes[s] + tile - 1) / tile; if (k < nt) { found = true; break; } k -= nt; }
        if (!found) return;
        const unsigned lo = k * tile, hi = min(lo + tile, S.bytes[s]);
        const char* src = S.src[s] + lo; char* dst = S.dst[s] + lo;
        const unsigned n = hi - lo;
        for (unsigned j = threadIdx.x * 16; j + 16 <= n; j += 256 * 16) {
            uint4 v; __builtin_memcpy(&v, src + j, 16); __builtin_memcpy(dst + j, &v, 16);
        }
        if (threadIdx.x < (n & 15)) dst[(n & ~15u) + threadIdx.x] = src[(n & ~15u) + threadIdx.x];
        t += gridDim.x;
    }
}

int main() {
    const size_t total = size_t(64) << 20;      // one window's worth of sequences
    char* h; CK(hipHostMalloc((void**)&h, total * 2 + 4096, 0));
    memset(h, 7, total * 2 + 4096);
    char* d; CK(hipMalloc((void**)&d, total + 4096));
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int misalign = 0; misalign < 2; ++misalign) {
        Segs S; S.n = 16;
        size_t at = 0;
        for (int i = 0; i < 16; ++i) { S.bytes[i] = (unsigned)(total / 16 - 37 * i); S.src[i] = h + (total / 8) * i + (misalign ? 3 + i : 0); S.dst[i] = d + at; at += S.bytes[i]; }
        for (int grid : {64, 128, 256, 512, 1024}) {
            hipLaunchKernelGGL(gather, dim3(grid), dim3(256), 0, st, S);
            CK(hipStreamSynchronize(st));
            CK(hipEventRecord(e0, st));
            for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(gather, dim3(grid), dim3(256), 0, st, S);
            CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("gather kernel misalign=%d grid=%4d: %.2f ms / window, %.1f GB/s\n", misalign, grid, ms / 10, at / 1e6 / (ms / 10));
        }
        // DMA: 16 copies
        auto c0 = std::chrono::steady_clock::now();
        CK(hipEventRecord(e0, st));
        for (int r = 0; r < 10; ++r) for (int i = 0; i < 16; ++i) CK(hipMemcpyAsync(S.dst[i], S.src[i], S.bytes[i], hipMemcpyHostToDevice, st));
        auto c1 = std::chrono::steady_clock::now();
        CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("hipMemcpyAsync x16 misalign=%d: %.2f ms / window, %.1f GB/s; host time per call %.1f us\n", misalign, ms / 10, at / 1e6 / (ms / 10),
               std::chrono::duration<double, std::micro>(c1 - c0).count() / 160);
    }
    // small copies (offsets: ~110 KB each)
    auto c0 = std::chrono::steady_clock::now();
    for (int r = 0; r < 160; ++r) CK(hipMemcpyAsync(d + r * 131072, h + r * 262144, 110000, hipMemcpyHostToDevice, st));
    auto c1 = std::chrono::steady_clock::now();
    CK(hipStreamSynchronize(st));
    printf("hipMemcpyAsync of 110 KB: host time per call %.1f us\n", std::chrono::duration<double, std::micro>(c1 - c0).count() / 160);
    // verify the gather
    std::vector<char> back(1 << 20); CK(hipMemcpy(back.data(), d, back.size(), hipMemcpyDeviceToHost));
    for (char c : back) if (c != 7) { printf("MISMATCH\n"); return 1; }
    printf("ok\n");
}
