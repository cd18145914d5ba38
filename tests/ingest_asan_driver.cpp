// ingest_asan_driver.cpp -- the host ingestion code (csrc/scg_ingest.cpp: raw-text windows, the host record scan) built with
// AddressSanitizer + UBSan and run over fuzzed FASTQ files with exact-size buffers: the raw windows must reassemble the
// file, every segment of the host scan must lie inside the buffers, nothing may be read or written out of bounds.
// Built and run by tests/test_ingest_asan.py:  ingest_asan_driver <seed> <rounds> <dir>
#include "scg_ingest.h"
#include "scg_host.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>
#include <zlib.h>
using namespace scg;
static std::string make(std::mt19937_64& rng, int& flaw) {
    std::string t;
    const int n = (int)(rng() % 3000);
    flaw = (int)(rng() % 8);
    for (int i = 0; i < n; ++i) {
        const int L = (int)(rng() % 200);
        std::string name = "@r" + std::to_string(i) + std::string(rng() % 5 == 0 ? rng() % 400 : 0, 'x');
        std::string s(L, 'A'), q(L, 'I');
        for (auto& c : s) c = "ACGTN"[rng() % 5];
        t += name + "\n" + s + "\n+\n" + q + "\n";
        if (flaw == 1 && i == n / 2) t += "\n";
        if (flaw == 2 && i == n / 3) t += "@m\nAC\nGT\n+\nII\nII\n";
        if (flaw == 3 && i == n / 2) t += "@p\nAC+GT\n+\nIIIII\n";
    }
    if (flaw == 4 && t.size() > 10) t.resize(t.size() - 1 - rng() % std::min<size_t>(t.size() - 1, 300));
    if (flaw == 5 && !t.empty()) t.pop_back();
    return t;
}
int main(int argc, char** argv) {
    const uint64_t seed = argc > 1 ? strtoull(argv[1], nullptr, 10) : 1;
    const int rounds = argc > 2 ? atoi(argv[2]) : 100;
    std::mt19937_64 rng(seed);
    long windows = 0, declined = 0;
    for (int r = 0; r < rounds; ++r) {
        int flaw;
        const std::string text = make(rng, flaw);
        const std::string file = std::string(argc > 3 ? argv[3] : "/tmp") + "/f.fastq";
        const char* path = file.c_str();
        FILE* f = fopen(path, "wb"); fwrite(text.data(), 1, text.size(), f); fclose(f);
        const size_t caps[5] = {4096, 9000, 70001, 300000, 1 << 20};
        const size_t cap = caps[rng() % 5];
        const int threads = 1 + (int)(rng() % 8);
        {   // raw windows
            auto src = TextSource::open(path, threads);
            std::vector<char> w(cap);      // exact size: ASan sees overruns
            std::string all;
            for (;;) { size_t got = src->next(w.data(), cap); if (src->unusual() || !got) break; all.append(w.data(), got); ++windows; }
            if (!src->unusual()) { std::string want = text; if (!want.empty() && want.back() != '\n') want += '\n'; if (all != want) { fprintf(stderr, "raw windows differ (round %d)\n", r); return 1; } }
        }
        {   // host record scan
            auto src = TextSource::open(path, threads);
            std::vector<char> seqs(cap);
            const size_t cap_off = cap / 16 / 4 + 257;
            std::vector<uint32_t> offs(cap_off);
            size_t reads = 0;
            for (;;) {
                ParsedWindow pw;
                size_t got = src->next_parsed(seqs.data(), cap, offs.data(), cap_off, pw);
                if (src->unusual() || !got) break;
                for (int i = 0; i < pw.n_segs; ++i) {
                    const ParsedSegment& g = pw.seg[i];
                    if (g.seq_at + g.seq_bytes > cap || g.off_at + g.n_records + 1 > cap_off) { fprintf(stderr, "segment out of range\n"); return 1; }
                    reads += g.n_records;
                }
                ++windows;
            }
            if (src->unusual()) ++declined;
            else if (flaw == 0 || flaw == 5) { size_t want = 0; for (size_t i = 0; i + 1 < text.size() || i < text.size(); ++i) if (text[i] == '\n') ++want; (void)want; }
        }
    }
    printf("ok: %d files, %ld windows, %ld declined by the host scan\n", rounds, windows, declined);
    return 0;
}
