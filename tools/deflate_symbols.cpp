// deflate_symbols.cpp -- measurement aid: the number of DEFLATE symbols (literals, matches, end-of-block codes) and of
// text bytes in a gzip / BGZF file, for "instructions per symbol" figures of the device inflater (tools/prof_bgzf_pmc.sh).
//   g++ -O2 -std=c++17 -o /tmp/deflate_symbols tools/deflate_symbols.cpp && /tmp/deflate_symbols file.gz
#include <cstdio>
#include <memory>
#include <vector>

#include "../screencounter_amd/csrc/scg_pgzip.h"

using namespace scg::pgz;

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    std::vector<uint8_t> file;
    FILE* f = std::fopen(argv[1], "rb");
    if (!f) { std::perror("open"); return 2; }
    uint8_t buf[1 << 16];
    size_t n;
    while ((n = std::fread(buf, 1, sizeof(buf), f)) > 0) file.insert(file.end(), buf, buf + n);
    std::fclose(f);
    const size_t size = file.size();
    file.resize(size + 64, 0);
    std::vector<uint16_t> out(WINDOW + (size_t(1) << 26));
    std::unique_ptr<Tables> T(new Tables);
    uint8_t lens[320];
    size_t symbols = 0, text = 0, members = 0, off = 0;
    while (off + 18 <= size && file[off] == 0x1f && file[off + 1] == 0x8b) {
        size_t p = off + 10;
        const uint8_t flg = file[off + 3];
        if (flg & 4) p += 2 + (file[p] | (size_t(file[p + 1]) << 8));
        for (int bit : {8, 16}) if (flg & bit) { while (file[p]) ++p; ++p; }
        if (flg & 2) p += 2;
        Bits br;
        uint64_t bit = uint64_t(p) * 8;
        size_t op = 0;
        int rc;
        do {
            br.open(file.data(), size, bit);
            rc = decode_block<true>(br, *T, out.data() + WINDOW, op, size_t(1) << 26, WINDOW, lens, &symbols);    // (counting only: distances are not policed)
            bit = br.bitpos();
            if (op > (size_t(1) << 25)) {            // keep a window of history, drop the rest
                std::copy(out.begin() + WINDOW + op - WINDOW, out.begin() + WINDOW + op, out.begin());
                text += op; op = 0;
                // (matches reach at most WINDOW symbols back: the copied tail in front of out[WINDOW] serves them)
            }
        } while (rc == BLOCK_OK);
        if (rc != BLOCK_FINAL) { std::fprintf(stderr, "decode error in member %zu\n", members); return 1; }
        text += op;
        ++members;
        off = size_t((bit + 7) >> 3) + 8;
    }
    std::printf("members %zu text_bytes %zu symbols %zu bytes_per_symbol %.3f\n", members, text, symbols, double(text) / double(symbols));
    return 0;
}
