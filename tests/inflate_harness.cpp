// inflate_harness.cpp -- host-side differential test of csrc/scg_inflate.h (the per-lane DEFLATE decoder of the device
// inflate kernel) against zlib.  Built by tests/test_inflate_cpu.py with g++ -fsanitize=address,undefined.
//   inflate_harness <seed> <rounds>
// Valid streams of many shapes (levels 0-9, Z_FIXED / Z_HUFFMAN_ONLY / Z_RLE / Z_FILTERED, FASTQ-like, random, runs)
// must decode identically; corrupted streams must be rejected whenever zlib rejects them and, when accepted, give
// exactly zlib's output; nothing may read or write out of bounds (ASan) or run away.
#include "../screencounter_amd/csrc/scg_inflate.h"

#include <zlib.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

static std::vector<uint8_t> make_data(std::mt19937_64& rng, int kind, size_t n) {
    std::vector<uint8_t> d(n);
    switch (kind) {
        case 0: {   // FASTQ-like
            size_t i = 0;
            uint64_t id = rng();
            while (i < n) {
                char name[96];
                int m = snprintf(name, sizeof(name), "@INSTR:%llu:FLOW:1:%llu:%llu 1:N:0:ACGT\n", (unsigned long long)(id % 97), (unsigned long long)(id % 100000), (unsigned long long)(rng() % 100000));
                ++id;
                for (int k = 0; k < m && i < n; ++k) d[i++] = (uint8_t)name[k];
                const int L = 20 + (int)(rng() % 140);
                for (int k = 0; k < L && i < n; ++k) d[i++] = (uint8_t)"ACGTN"[(rng() % 100) < 99 ? rng() % 4 : 4];
                if (i < n) d[i++] = '\n';
                if (i < n) d[i++] = '+';
                if (i < n) d[i++] = '\n';
                for (int k = 0; k < L && i < n; ++k) d[i++] = (uint8_t)("FFFF:FF,F#"[rng() % 10]);
                if (i < n) d[i++] = '\n';
            }
            break;
        }
        case 1: for (auto& b : d) b = (uint8_t)rng(); break;                            // incompressible
        case 2: for (auto& b : d) b = 0; break;                                          // one long run
        case 3: for (size_t i = 0; i < n; ++i) d[i] = (uint8_t)("abcabcabd"[i % 9]); break;  // short periods: overlapping copies
        case 4: for (auto& b : d) b = (uint8_t)(rng() % 3 ? 'a' + rng() % 4 : rng()); break; // skewed alphabet: long and short codes
        case 5: {   // many distinct symbols with geometric frequencies: code lengths up to 15
            for (auto& b : d) { int k = 0; while (k < 200 && (rng() & 1)) ++k; b = (uint8_t)k; }
            break;
        }
        default: for (size_t i = 0; i < n; ++i) d[i] = (uint8_t)(i * 7 + (i >> 8)); break;
    }
    return d;
}

static std::vector<uint8_t> deflate_raw(const std::vector<uint8_t>& d, int level, int strategy, int flush_every) {
    z_stream zs;
    memset(&zs, 0, sizeof(zs));
    if (deflateInit2(&zs, level, Z_DEFLATED, -15, 8, strategy) != Z_OK) abort();
    std::vector<uint8_t> out(deflateBound(&zs, d.size()) + 64 + (flush_every ? d.size() / flush_every * 16 : 0));
    zs.next_out = out.data();
    zs.avail_out = (uInt)out.size();
    size_t at = 0;
    while (flush_every && at + flush_every < d.size()) {      // several blocks, incl. empty stored ones from Z_FULL_FLUSH
        zs.next_in = const_cast<Bytef*>(d.data() + at);
        zs.avail_in = (uInt)flush_every;
        if (deflate(&zs, (at / flush_every) % 2 ? Z_FULL_FLUSH : Z_SYNC_FLUSH) != Z_OK) abort();
        at += flush_every;
    }
    zs.next_in = const_cast<Bytef*>(d.data() + at);
    zs.avail_in = (uInt)(d.size() - at);
    if (deflate(&zs, Z_FINISH) != Z_STREAM_END) abort();
    out.resize(zs.total_out);
    deflateEnd(&zs);
    return out;
}

// zlib's verdict on a raw stream that must produce exactly n bytes from exactly the whole input.
static bool zlib_inflate(const std::vector<uint8_t>& c, size_t n, std::vector<uint8_t>& out) {
    z_stream zs;
    memset(&zs, 0, sizeof(zs));
    if (inflateInit2(&zs, -15) != Z_OK) abort();
    out.assign(n + 1, 0);
    zs.next_in = const_cast<Bytef*>(c.data());
    zs.avail_in = (uInt)c.size();
    zs.next_out = out.data();
    zs.avail_out = (uInt)out.size();
    const int rc = inflate(&zs, Z_FINISH);
    const bool ok = rc == Z_STREAM_END && zs.total_out == n && zs.avail_in == 0;
    inflateEnd(&zs);
    out.resize(n);
    return ok;
}

// The decoder with "wavefronts" of 1, 7 and 64 lanes (the device runs 64): all must agree, byte for byte and verdict
// for verdict -- the wider ones exercise the batched literals and the deferred stores of short matches.
static int ours(const std::vector<uint8_t>& c, size_t n, std::vector<uint8_t>& out) {
    // exact-size buffers so that ASan sees any access beyond the slack or the output
    std::vector<uint8_t> in(c.size() + scginf::IN_SLACK, 0xA5);
    memcpy(in.data(), c.data(), c.size());
    static scginf::LaneTables T;
    out.assign(n, 0xEE);
    const int rc = scginf::inflate_member(in.data(), (uint32_t)c.size(), out.data(), (uint32_t)n, T, scginf::HostWave<64>());
    std::vector<uint8_t> o1(n, 0xEE), o7(n, 0xEE);
    const int rc1 = scginf::inflate_member(in.data(), (uint32_t)c.size(), o1.data(), (uint32_t)n, T, scginf::HostWave<1>());
    const int rc7 = scginf::inflate_member(in.data(), (uint32_t)c.size(), o7.data(), (uint32_t)n, T, scginf::HostWave<7>());
    // (which of the two error codes a broken stream earns may depend on where a run of literals is cut: both mean "not this decoder's")
    const bool ok = rc == scginf::INFLATE_OK;
    if ((rc1 == scginf::INFLATE_OK) != ok || (rc7 == scginf::INFLATE_OK) != ok || (ok && (o1 != out || o7 != out))) {
        fprintf(stderr, "FAIL the lane widths disagree: rc %d / %d / %d\n", rc, rc1, rc7);
        exit(1);
    }
    return rc;
}

int main(int argc, char** argv) {
    const uint64_t seed = argc > 1 ? strtoull(argv[1], nullptr, 10) : 1;
    const int rounds = argc > 2 ? atoi(argv[2]) : 200;
    std::mt19937_64 rng(seed);
    long valid = 0, corrupt = 0, corrupt_accepted = 0, size_variants = 0;
    const int strategies[5] = {Z_DEFAULT_STRATEGY, Z_FIXED, Z_HUFFMAN_ONLY, Z_RLE, Z_FILTERED};
    for (int r = 0; r < rounds; ++r) {
        const int kind = (int)(rng() % 7);
        const size_t n = (r % 11 == 0) ? rng() % 40 : (r % 7 == 0 ? 65280 : 1 + rng() % 70000);
        const std::vector<uint8_t> d = make_data(rng, kind, n);
        const int level = (int)(rng() % 10);
        const int strategy = strategies[rng() % 5];
        const int flush_every = (rng() % 3 == 0 && n > 64) ? (int)(16 + rng() % (n / 2)) : 0;
        const std::vector<uint8_t> c = deflate_raw(d, level, strategy, flush_every);
        std::vector<uint8_t> got;
        const int rc = ours(c, n, got);
        if (rc != scginf::INFLATE_OK || got != d) {
            fprintf(stderr, "FAIL valid stream: round %d kind %d n %zu level %d strategy %d flush %d rc %d\n", r, kind, n, level, strategy, flush_every, rc);
            return 1;
        }
        ++valid;
        // wrong announced sizes must be rejected
        if (n > 0) {
            if (ours(c, n - 1, got) == scginf::INFLATE_OK) { fprintf(stderr, "FAIL short output accepted (round %d)\n", r); return 1; }
        }
        if (ours(c, n + 1, got) == scginf::INFLATE_OK) { fprintf(stderr, "FAIL long output accepted (round %d)\n", r); return 1; }
        {
            std::vector<uint8_t> c2 = c;
            c2.push_back(0);
            if (ours(c2, n, got) == scginf::INFLATE_OK) { fprintf(stderr, "FAIL trailing input accepted (round %d)\n", r); return 1; }
            if (c.size() > 1) {
                c2.assign(c.begin(), c.end() - 1);
                std::vector<uint8_t> z;
                if (ours(c2, n, got) == scginf::INFLATE_OK && !zlib_inflate(c2, n, z)) { fprintf(stderr, "FAIL truncated input accepted (round %d)\n", r); return 1; }
            }
            size_variants += 4;
        }
        // corrupted streams: bit flips, byte smashes, early in the header and anywhere
        for (int k = 0; k < 12 && !c.empty(); ++k) {
            std::vector<uint8_t> bad = c;
            const int flips = 1 + (int)(rng() % 3);
            for (int f = 0; f < flips; ++f) {
                const size_t where = (k % 3 == 0) ? rng() % std::min<size_t>(bad.size(), 24) : rng() % bad.size();
                if (rng() % 4 == 0) bad[where] = (uint8_t)rng(); else bad[where] ^= (uint8_t)(1u << (rng() % 8));
            }
            std::vector<uint8_t> zout, mine;
            const bool zok = zlib_inflate(bad, n, zout);
            const int mrc = ours(bad, n, mine);
            ++corrupt;
            if (mrc == scginf::INFLATE_OK) {
                ++corrupt_accepted;
                if (!zok) { fprintf(stderr, "FAIL accepted a stream zlib rejects: round %d variant %d\n", r, k); return 1; }
                if (mine != zout) { fprintf(stderr, "FAIL output differs from zlib on an accepted corrupted stream: round %d variant %d\n", r, k); return 1; }
            }
            // (rejecting a stream zlib accepts only costs a fall-back to the host path, but it should not happen either)
            if (zok && mrc != scginf::INFLATE_OK) { fprintf(stderr, "FAIL rejected a stream zlib accepts: round %d variant %d rc %d\n", r, k, mrc); return 1; }
        }
    }
    printf("ok: %ld valid streams, %ld size variants, %ld corrupted streams (%ld still valid and identical to zlib)\n", valid, size_variants, corrupt, corrupt_accepted);
    return 0;
}
