#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <vector>
#include <thread>
#include <chrono>
#include <string>
#include <cstdint>
#include <sys/mman.h>
#include <immintrin.h>
// fused: copy the line to out while looking for '\n' (stop) or '+' (fail); returns the line length or -1
__attribute__((target("avx2"))) static inline long copy_line(const char* s, const char* e, char* out) {
    const __m256i nl = _mm256_set1_epi8('\n'), pl = _mm256_set1_epi8('+');
    const char* p = s;
    while (p + 32 <= e) {
        const __m256i v = _mm256_loadu_si256((const __m256i*)p);
        _mm256_storeu_si256((__m256i*)(out + (p - s)), v);
        const unsigned mn = (unsigned)_mm256_movemask_epi8(_mm256_cmpeq_epi8(v, nl));
        const unsigned mp = (unsigned)_mm256_movemask_epi8(_mm256_cmpeq_epi8(v, pl));
        if (mn | mp) {
            if (!mn) return -1;
            const int j = __builtin_ctz(mn);
            if (mp & ((1u << j) - 1)) return -1;
            return (p - s) + j;
        }
        p += 32;
    }
    for (; p < e; ++p) { if (*p == '\n') return p - s; if (*p == '+') return -1; out[p - s] = *p; }
    return -2;
}
// true if the first '\n' at or after q is exactly q + L
__attribute__((target("avx2"))) static inline bool line_is(const char* q, size_t L) {
    const __m256i nl = _mm256_set1_epi8('\n');
    size_t j = 0;
    unsigned m = 0;
    for (; j + 32 <= L; j += 32) m |= (unsigned)_mm256_movemask_epi8(_mm256_cmpeq_epi8(_mm256_loadu_si256((const __m256i*)(q + j)), nl));
    if (m) return false;
    for (; j < L; ++j) if (q[j] == '\n') return false;
    return q[L] == '\n';
}
__attribute__((target("avx2"))) static inline size_t rec2(const char* data, size_t size, size_t p, char*& out, uint32_t*& offs, uint32_t& at, uint32_t& mx) {
    if (data[p] != '@') return 0;
    const char* e = data + size;
    const char* l1 = (const char*)memchr(data + p, '\n', size - p);
    if (!l1) return 0;
    const char* s0 = l1 + 1;
    const long Ls = copy_line(s0, e, out);
    if (Ls < 0) return 0;
    const size_t L = Ls;
    const char* p0 = s0 + L + 1;
    if (p0 >= e || *p0 != '+') return 0;
    const char* l3 = (p0 + 1 < e && p0[1] == '\n') ? p0 + 1 : (const char*)memchr(p0, '\n', e - p0);
    if (!l3) return 0;
    const char* q0 = l3 + 1;
    if ((size_t)(e - q0) < L + 1) return 0;
    if (!line_is(q0, L)) return 0;
    out += L;
    *offs++ = at; at += L; if (L > mx) mx = L;
    return q0 + L + 1 - data;
}

static inline size_t rec(const char* data, size_t size, size_t p, char*& out, uint32_t*& offs, uint32_t& at, uint32_t& mx) {
    if (data[p] != '@') return 0;
    const char* e = data + size;
    const char* l1 = (const char*)memchr(data + p, '\n', size - p);
    if (!l1) return 0;
    const char* s0 = l1 + 1;
    const char* l2 = s0 < e ? (const char*)memchr(s0, '\n', e - s0) : nullptr;
    if (!l2) return 0;
    const size_t L = l2 - s0;
    if (memchr(s0, '+', L)) return 0;
    const char* p0 = l2 + 1;
    if (p0 >= e || *p0 != '+') return 0;
    const char* l3 = (p0 + 1 < e && p0[1] == '\n') ? p0 + 1 : (const char*)memchr(p0, '\n', e - p0);
    if (!l3) return 0;
    const char* q0 = l3 + 1;
    if ((size_t)(e - q0) < L + 1) return 0;
    const char* l4 = (const char*)memchr(q0, '\n', L + 1);
    if (l4 != q0 + L) return 0;
    memcpy(out, s0, L); out += L;
    *offs++ = at; at += L; if (L > mx) mx = L;
    return l4 + 1 - data;
}
int main(int argc, char** argv) {
    int T = argc > 1 ? atoi(argv[1]) : 8;
    size_t nrec = argc > 2 ? atol(argv[2]) : 4000000;
    std::string one;
    std::vector<char> text;
    text.reserve(nrec * 330);
    srand(1);
    for (size_t i = 0; i < nrec; ++i) {
        char name[64]; int n = snprintf(name, 64, "@read%zu/1 some:description\n", i);
        text.insert(text.end(), name, name + n);
        for (int j = 0; j < 150; ++j) text.push_back("ACGT"[rand() & 3]);
        text.push_back('\n'); text.push_back('+'); text.push_back('\n');
        for (int j = 0; j < 150; ++j) text.push_back('I');
        text.push_back('\n');
    }
    size_t size = text.size();
    printf("text %.2f GB\n", size / 1e9);
    std::vector<char> dst(size); std::vector<uint32_t> offs(nrec + 64);
    memset(dst.data(), 1, size);
    for (int rep = 0; rep < 4; ++rep) {
        // memcpy
        auto t0 = std::chrono::steady_clock::now();
        { std::vector<std::thread> th; for (int t = 0; t < T; ++t) th.emplace_back([&, t] { size_t a = size * t / T, b = size * (t + 1) / T; memcpy(dst.data() + a, text.data() + a, b - a); }); for (auto& x : th) x.join(); }
        auto t1 = std::chrono::steady_clock::now();
        // parse: slice boundaries at record starts (cheat: records equal-ish; find next "\n@" )
        std::vector<size_t> cut(T + 1); cut[0] = 0; cut[T] = size;
        for (int t = 1; t < T; ++t) { size_t g = size * t / T; // find record start: brute force using strict check
            for (;; ++g) { if (text[g] == '@' && text[g-1]=='\n') { char tmp[400]; char* o = tmp; uint32_t of[2]; uint32_t* op = of; uint32_t at=0,mx=0; if (rec(text.data(), size, g, o, op, at, mx)) break; } } cut[t] = g; }
        std::vector<size_t> cnt(T);
        auto t2 = std::chrono::steady_clock::now();
        { std::vector<std::thread> th; for (int t = 0; t < T; ++t) th.emplace_back([&, t] {
            size_t p = cut[t], b = cut[t + 1]; char* o = dst.data() + p; uint32_t* op = offs.data() + (nrec / T + 8) * t; uint32_t at = 0, mx = 0; size_t n = 0;
            while (p < b) { p = (rep==2? rec2(text.data(), size, p, o, op, at, mx) : rec(text.data(), size, p, o, op, at, mx)); if (!p) { printf("fail\n"); return; } ++n; }
            cnt[t] = n; }); for (auto& x : th) x.join(); }
        auto t3 = std::chrono::steady_clock::now();
        size_t tot = 0; for (auto c : cnt) tot += c;
        printf("memcpy %.1f ms (%.1f GB/s)  parse %.1f ms (%.1f GB/s text, %.1f Mrec/s) n=%zu\n", std::chrono::duration<double, std::milli>(t1 - t0).count(), size / 1e9 / std::chrono::duration<double>(t1 - t0).count(),
               std::chrono::duration<double, std::milli>(t3 - t2).count(), size / 1e9 / std::chrono::duration<double>(t3 - t2).count(), tot / 1e6 / std::chrono::duration<double>(t3 - t2).count(), tot);
    }
}
