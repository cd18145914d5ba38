// pgzip_harness.cpp -- test driver for the parallel gzip decoder (screencounter_amd/csrc/scg_pgzip.*), host only.
//
//   pgzip_harness <file.gz> <threads> [read_cap_bytes]
//   pgzip_harness --time <file.gz> <threads>           decode only (no comparison): "time <bytes> <seconds>"
//
// Decodes the file with scg::ParallelGunzip and, independently, with zlib's gzread (the reference's reader,
// byteme/GzipFileReader.hpp:39-51) and prints one line:
//   "same <bytes> <seconds parallel> <seconds zlib>"     both agree on the text
//   "declined <zlib verdict>"                            the parallel decoder handed the file back (zlib: "ok <bytes>" or "error")
//   "DIFF ..."                                           the parallel decoder accepted the file with another text: a bug
// Built by tests/test_pgzip_cpu.py, plain and with -fsanitize=address,undefined.
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <zlib.h>

#include "../screencounter_amd/csrc/scg_pgzip.hpp"

static int time_only(const char* path, int threads) {
    std::vector<uint8_t> file;
    FILE* f = std::fopen(path, "rb");
    if (!f) { std::perror("open"); return 2; }
    uint8_t buf[1 << 16];
    size_t n;
    while ((n = std::fread(buf, 1, sizeof(buf), f)) > 0) file.insert(file.end(), buf, buf + n);
    std::fclose(f);
    const size_t size = file.size();
    file.resize(size + 64, 0);
    std::vector<char> window(size_t(128) << 20);
    for (int rep = 0; rep < 3; ++rep) {
        const auto t0 = std::chrono::steady_clock::now();
        size_t total = 0;
        bool failed = false;
        {
            scg::ParallelGunzip pg(file.data(), size, threads);
            for (;;) {
                const size_t got = pg.read(window.data(), window.size());
                if (!got) { failed = pg.failed(); break; }
                total += got;
            }
        }
        std::printf("time %zu %.4f%s\n", total, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(), failed ? " FAILED" : "");
    }
    return 0;
}

int main(int argc, char** argv) {
    if (argc >= 4 && !std::strcmp(argv[1], "--time")) return time_only(argv[2], std::atoi(argv[3]));
    if (argc < 3) { std::fprintf(stderr, "usage: %s file.gz threads [read_cap]\n", argv[0]); return 2; }
    const char* path = argv[1];
    const int threads = std::atoi(argv[2]);
    const size_t cap = argc > 3 ? static_cast<size_t>(std::atoll(argv[3])) : (size_t(64) << 20);
    std::vector<uint8_t> file;
    {
        FILE* f = std::fopen(path, "rb");
        if (!f) { std::perror("open"); return 2; }
        uint8_t buf[1 << 16];
        size_t n;
        while ((n = std::fread(buf, 1, sizeof(buf), f)) > 0) file.insert(file.end(), buf, buf + n);
        std::fclose(f);
    }
    const size_t size = file.size();
    file.resize(size + 64, 0);

    std::vector<char> text;
    bool failed = false;
    const auto t0 = std::chrono::steady_clock::now();
    {
        scg::ParallelGunzip pg(file.data(), size, threads);
        std::vector<char> window(cap);
        for (;;) {
            const size_t n = pg.read(window.data(), cap);
            if (!n) { failed = pg.failed(); break; }
            text.insert(text.end(), window.begin(), window.begin() + static_cast<long>(n));
        }
    }
    const double t_par = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();

    std::vector<char> ref;
    bool ref_error = false;
    const auto t1 = std::chrono::steady_clock::now();
    {
        gzFile gz = gzopen(path, "rb");
        if (!gz) { std::perror("gzopen"); return 2; }
        gzbuffer(gz, 1 << 20);
        std::vector<char> buf(size_t(1) << 22);
        for (;;) {
            const int n = gzread(gz, buf.data(), static_cast<unsigned>(buf.size()));
            if (n < 0) { ref_error = true; break; }
            if (n == 0) break;
            ref.insert(ref.end(), buf.begin(), buf.begin() + n);
        }
        gzclose(gz);
    }
    const double t_ref = std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count();

    if (failed) {
        if (ref_error) std::printf("declined error\n");
        else std::printf("declined ok %zu\n", ref.size());
        return 0;
    }
    if (ref_error) { std::printf("DIFF accepted %zu bytes of a file zlib rejects\n", text.size()); return 1; }
    if (text.size() != ref.size() || (text.size() && std::memcmp(text.data(), ref.data(), text.size()) != 0)) {
        size_t at = 0;
        while (at < text.size() && at < ref.size() && text[at] == ref[at]) ++at;
        std::printf("DIFF sizes %zu vs %zu, first difference at %zu\n", text.size(), ref.size(), at);
        return 1;
    }
    std::printf("same %zu %.4f %.4f\n", text.size(), t_par, t_ref);
    return 0;
}
