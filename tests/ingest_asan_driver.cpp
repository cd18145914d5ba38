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
// The text as a BGZF file: members of `block` bytes of text each, bgzip's header, an empty end-of-file member.
static std::string bgzf(const std::string& text, size_t block) {
    std::string out;
    auto member = [&](const char* p, size_t n) {
        z_stream zs;
        memset(&zs, 0, sizeof(zs));
        deflateInit2(&zs, 6, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY);
        std::string payload(deflateBound(&zs, n) + 16, '\0');
        zs.next_in = reinterpret_cast<Bytef*>(const_cast<char*>(p));
        zs.avail_in = (uInt)n;
        zs.next_out = reinterpret_cast<Bytef*>(&payload[0]);
        zs.avail_out = (uInt)payload.size();
        deflate(&zs, Z_FINISH);
        payload.resize(zs.total_out);
        deflateEnd(&zs);
        const uint32_t bsize = (uint32_t)payload.size() + 25, crc = (uint32_t)crc32(crc32(0, nullptr, 0), reinterpret_cast<const Bytef*>(p), (uInt)n), isize = (uint32_t)n;
        const unsigned char head[18] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, (unsigned char)(bsize & 255), (unsigned char)(bsize >> 8)};
        out.append(reinterpret_cast<const char*>(head), 18);
        out += payload;
        for (uint32_t v : {crc, isize}) for (int k = 0; k < 4; ++k) out.push_back((char)((v >> (8 * k)) & 255));
    };
    for (size_t at = 0; at < text.size(); at += block) member(text.data() + at, std::min(block, text.size() - at));
    member("", 0);
    return out;
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
    // BGZF: windows of text inflated by the host threads, and the member batches the device inflater is handed
    long batches = 0;
    for (int r = 0; r < rounds / 2; ++r) {
        int flaw;
        std::string text = make(rng, flaw);
        if (flaw != 0 && flaw != 5) continue;                  // (ordinary text; what the scan makes of the rest is the GPU tests' business)
        const size_t blocks[4] = {300, 5000, 30000, 65280};
        const std::string gz = bgzf(text, blocks[rng() % 4]);
        const std::string file = std::string(argc > 3 ? argv[3] : "/tmp") + "/f.bgzf.gz";
        FILE* f = fopen(file.c_str(), "wb"); fwrite(gz.data(), 1, gz.size(), f); fclose(f);
        const int threads = 1 + (int)(rng() % 8);
        {
            auto src = TextSource::open(file.c_str(), threads);
            const size_t cap = 70000 + rng() % 400000;
            std::vector<char> w(cap);
            std::string all;
            for (;;) { size_t got = src->next(w.data(), cap); if (src->unusual() || !got) break; all.append(w.data(), got); }
            std::string want = text; if (!want.empty() && want.back() != '\n') want += '\n';
            if (!src->unusual() && all != want) { fprintf(stderr, "BGZF windows differ (round %d)\n", r); return 1; }
        }
        {
            auto src = TextSource::open(file.c_str(), threads);
            const size_t slack = 64, cap = 70000 + rng() % 200000, cap_text = 66000 + rng() % 300000;
            std::vector<char> staging(cap);
            std::vector<CompressedMember> members;
            std::string all;
            for (;;) {
                size_t text_bytes = 0; bool last = false;
                const size_t got = src->next_members(staging.data(), cap, slack, cap_text, members, text_bytes, last);
                if (src->unusual() || !got) break;
                ++batches;
                for (const CompressedMember& m : members) {
                    if ((size_t)m.in_off + m.in_len + slack > got + 0 || (size_t)m.out_off + m.out_len > text_bytes) { fprintf(stderr, "member out of range\n"); return 1; }
                    std::string piece(m.out_len, '\0');
                    z_stream zs; memset(&zs, 0, sizeof(zs)); inflateInit2(&zs, -15);
                    zs.next_in = reinterpret_cast<Bytef*>(staging.data() + m.in_off); zs.avail_in = m.in_len;
                    zs.next_out = reinterpret_cast<Bytef*>(&piece[0]); zs.avail_out = m.out_len + (m.out_len ? 0 : 1);
                    const int rc = inflate(&zs, Z_FINISH);
                    if (rc != Z_STREAM_END || zs.total_out != m.out_len || zs.avail_in != 0) { fprintf(stderr, "payload does not inflate (rc %d)\n", rc); return 1; }
                    inflateEnd(&zs);
                    all += piece;
                }
                if (last) break;
            }
            if (!src->unusual() && all != text) { fprintf(stderr, "member batches do not reassemble the text (round %d)\n", r); return 1; }
        }
    }
    printf("ok: %d files, %ld windows, %ld declined by the host scan, %ld member batches\n", rounds, windows, declined, batches);
    return 0;
}
