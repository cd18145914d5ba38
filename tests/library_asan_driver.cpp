// library_asan_driver.cpp -- the index builder (csrc/scg_library.cpp: IUPAC expansion, duplicate detection, chains and
// tables of every position group, built by one thread per group) under AddressSanitizer + UBSan, with a check that
// every entry can be found again the way the device looks for it: hash of its group key, linear probing to the slot
// whose head shares the key, then along the chain.  Built and run by tests/test_ingest_asan.py:  <seed> <rounds>
#include "scg_host.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <set>
#include <string>
#include <vector>

using namespace scg;

static bool reachable(const HostIndex& X, size_t e) {
    const size_t n = static_cast<size_t>(X.n_entries);
    const uint32_t cap = X.slot_mask + 1;
    for (int s = 0; s < X.nseg; ++s) {
        const uint32_t* node = X.nodes.data() + static_cast<size_t>(s) * n * 4;
        const uint32_t* table = X.tables.data() + static_cast<size_t>(s) * cap * 4;
        const uint64_t mask = X.segmask[s];
        const uint64_t key = (static_cast<uint64_t>(node[4 * e + 1]) << 32) | node[4 * e];
        const uint64_t sk = key & mask;
        uint32_t pos = scg_hash64(sk) & X.slot_mask;
        bool found = false;
        for (uint32_t step = 0; step <= cap; ++step, pos = (pos + 1) & X.slot_mask) {
            const uint32_t* slot = table + 4 * static_cast<size_t>(pos);
            if (slot[3] == SCG_SLOT_EMPTY) return false;
            const uint64_t hk = (static_cast<uint64_t>(slot[1]) << 32) | slot[0];
            if ((hk & mask) != sk) continue;
            // the slot holds a copy of the chain's head; the chain continues in the node array
            if (hk == key && slot[2] == node[4 * e + 2]) { found = true; break; }
            for (uint32_t nx = slot[3]; nx != 0xFFFFFFFFu; nx = node[4 * static_cast<size_t>(nx) + 3]) {
                if (nx >= n) return false;
                if (nx == e) { found = true; break; }
            }
            break;
        }
        if (!found) return false;
    }
    return true;
}

int main(int argc, char** argv) {
    const uint64_t seed = argc > 1 ? strtoull(argv[1], nullptr, 10) : 1;
    const int rounds = argc > 2 ? atoi(argv[2]) : 50;
    std::mt19937_64 rng(seed);
    long entries = 0;
    for (int r = 0; r < rounds; ++r) {
        const int len = 4 + static_cast<int>(rng() % 29);
        int n = 1 + static_cast<int>(rng() % (r % 5 == 0 ? 40000 : 600));
        if (len < 10) n = std::min(n, (1 << (2 * len)) / 4);                  // (there are only 4^len sequences)
        const int mm = static_cast<int>(rng() % 4);
        std::set<std::string> seen;
        std::vector<std::string> pool;
        while (static_cast<int>(pool.size()) < n) {
            std::string s(len, 'A');
            for (auto& c : s) c = "ACGT"[rng() % 4];
            if (rng() % 50 == 0) s[rng() % len] = "RYSWKMN"[rng() % 7];      // an ambiguous base now and then
            if (seen.insert(s).second) pool.push_back(s);
            if (seen.size() > 100000) break;
        }
        std::vector<const char*> p;
        for (auto& s : pool) p.push_back(s.c_str());
        try {
            const HostIndex X = build_index(p.data(), static_cast<int32_t>(p.size()), len, mm);
            for (size_t e = 0; e < static_cast<size_t>(X.n_entries); ++e) {
                if (!reachable(X, e)) { fprintf(stderr, "entry %zu not reachable (round %d: n %d len %d mm %d)\n", e, r, n, len, mm); return 1; }
            }
            entries += X.n_entries;
        } catch (const Error&) {
            // two barcodes whose expansions collide: the reference's "duplicate sequences" error, fine
        }
    }
    printf("ok: %d pools, %ld entries reachable through every table\n", rounds, entries);
    return 0;
}
