// scg_library.cpp -- host-side compilation of templates and barcode pools into the flat tables
// the kernels consume.  Replaces the constructors of kaori::ScanTemplate, SimpleBarcodeSearch,
// SegmentedBarcodeSearch and the mismatch trie build (see scg_host.h for citations).
#include "scg_host.h"

#include <algorithm>
#include <array>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <mutex>
#include <thread>
#include <unordered_map>

namespace scg {

// Slots per entry of the group tables.  A wavefront repeats a probe step as long as ANY of its 64 lanes has not found its
// chain, so collisions cost far more than their per-lookup rate suggests: at 50 % load nearly every wavefront probed twice
// or more.  Small libraries, whose tables stay cache-resident anyway, get very sparse tables; for 100 k barcodes a
// quarter load is the measured optimum (MI355X: config 3 -13 % at 64 slots per entry, configs 2 and 5 -4 % at 4, +1 % at
// 8 where the tables start missing the L2).  SCG_TABLE_FACTOR overrides (tuning aid).
static uint32_t table_factor(size_t n) {
    if (const char* e = std::getenv("SCG_TABLE_FACTOR")) { const int v = std::atoi(e); if (v >= 2) return static_cast<uint32_t>(v); }
    if (n <= 4096) return 64;
    if (n <= 32768) return 16;
    return 4;
}

namespace {

// code = (ascii >> 1) & 3 :  A0 C1 T2 G3
inline int base_code(char c) {
    switch (c) {
        case 'A': case 'a': return 0;
        case 'C': case 'c': return 1;
        case 'T': case 't': return 2;
        case 'G': case 'g': return 3;
        default: return -1;
    }
}

// Allowed codes of a library character in the reference's expansion order A, C, G, T
// (kaori/MismatchTrie.hpp:163-188).  Returns the count, 0 for an unknown character.
struct IupacTable {
    uint8_t n[256];
    uint8_t code[256][4];
    IupacTable() {
        std::memset(n, 0, sizeof(n));
        std::memset(code, 0, sizeof(code));
        const int A = 0, C = 1, T = 2, G = 3;
        auto def = [&](char c, std::initializer_list<int> codes) {
            for (int ch : {static_cast<int>(c), static_cast<int>(c) + ('a' - 'A')}) {
                int k = 0;
                for (int x : codes) code[ch][k++] = static_cast<uint8_t>(x);
                n[ch] = static_cast<uint8_t>(k);
            }
        };
        def('A', {A}); def('C', {C}); def('G', {G}); def('T', {T});
        def('R', {A, G}); def('Y', {C, T}); def('S', {C, G}); def('W', {A, T}); def('K', {G, T}); def('M', {A, C});
        def('B', {C, G, T}); def('D', {A, G, T}); def('H', {A, C, T}); def('V', {A, C, G});
        def('N', {A, C, G, T});
    }
};
const IupacTable& iupac() {
    static const IupacTable t;
    return t;
}
inline int iupac_codes(char c, int out[4]) {
    const IupacTable& t = iupac();
    const unsigned char u = static_cast<unsigned char>(c);
    for (int k = 0; k < 4; ++k) out[k] = t.code[u][k];
    return t.n[u];
}

const int64_t MAX_EXPANSIONS = int64_t(1) << 26;

// Calls f(key) for every concrete expansion of `s` in lexicographic (A,C,G,T) order.
template<class F>
void for_each_expansion(const char* s, int len, F f) {
    int cnt[SCG_MAX_BARCODE], codes[SCG_MAX_BARCODE][4], choice[SCG_MAX_BARCODE];
    bool plain = true;
    uint64_t only = 0;
    for (int p = 0; p < len; ++p) {
        cnt[p] = iupac_codes(s[p], codes[p]);
        choice[p] = 0;
        plain &= cnt[p] == 1;
        const uint64_t c = static_cast<uint64_t>(codes[p][0]);
        only |= ((c & 1) << p) | ((c >> 1) << (32 + p));
    }
    if (plain) {            // the usual library: A, C, G, T only
        f(only);
        return;
    }
    for (;;) {
        uint64_t key = 0;
        for (int p = 0; p < len; ++p) {
            uint64_t c = static_cast<uint64_t>(codes[p][choice[p]]);
            key |= ((c & 1) << p) | ((c >> 1) << (32 + p));    // plane-split (scg_common.h)
        }
        f(key);
        int p = len - 1;
        for (; p >= 0; --p) {
            if (++choice[p] < cnt[p]) break;
            choice[p] = 0;
        }
        if (p < 0) break;
    }
}

int64_t count_expansions(const char* const* pool, int32_t n, int32_t len) {
    int64_t total = 0;
    int codes[4];
    for (int32_t i = 0; i < n; ++i) {
        int64_t m = 1;
        for (int p = 0; p < len; ++p) {
            int c = iupac_codes(pool[i][p], codes);
            if (c == 0) {
                // kaori/MismatchTrie.hpp:187
                throw Error(SCG_ERR_INVALID, std::string("unknown base '") + pool[i][p] + "' detected when constructing the trie");
            }
            m *= c;
            if (m > MAX_EXPANSIONS) break;
        }
        total += m;
        if (total > MAX_EXPANSIONS) {
            throw Error(SCG_ERR_UNSUPPORTED, "barcode pool expands to more than 2^26 concrete sequences");
        }
    }
    return total;
}

uint32_t capacity_for(int64_t entries, int slots_per_entry = 2) {
    uint64_t cap = 16;
    while (cap < static_cast<uint64_t>(entries) * static_cast<uint64_t>(slots_per_entry)) cap <<= 1;
    return static_cast<uint32_t>(cap);
}

struct Builder {
    std::vector<uint64_t>& keys;
    std::vector<int32_t>& vals;
    uint32_t mask;
    Builder(std::vector<uint64_t>& k, std::vector<int32_t>& v, uint32_t capacity) : keys(k), vals(v), mask(capacity - 1) {
        keys.assign(capacity, SCG_EMPTY_KEY);
        vals.assign(capacity, -1);
    }
    // Returns the slot holding `key`, inserting it (with val) if absent; *existed tells which.
    uint32_t upsert(uint64_t key, int32_t val, bool* existed) {
        uint32_t h = scg_hash64(key) & mask;
        for (;;) {
            if (keys[h] == key) { *existed = true; return h; }
            if (keys[h] == SCG_EMPTY_KEY) { keys[h] = key; vals[h] = val; *existed = false; return h; }
            h = (h + 1) & mask;
        }
    }
};

void check_len(int32_t len) {
    if (len > SCG_MAX_BARCODE) {
        throw Error(SCG_ERR_UNSUPPORTED, "variable regions longer than 32 bp are not supported by this engine (got " + std::to_string(len) + ")");
    }
}

[[noreturn]] void throw_duplicate(int32_t a, int32_t b) {
    // kaori/MismatchTrie.hpp:119-122
    throw Error(SCG_ERR_INVALID, "duplicate sequences detected (" + std::to_string(a + 1) + ", " + std::to_string(b + 1) + ") when constructing the trie");
}

} // namespace

HostTemplate parse_template(const char* constant, int strand) {
    size_t len = std::strlen(constant);
    if (len > SCG_MAX_TEMPLATE) {
        // src/count_single_barcodes.cpp:46
        throw Error(SCG_ERR_INVALID, "lacking compile-time support for constant regions longer than 256 bp");
    }
    HostTemplate out;
    std::memset(&out.t, 0, sizeof(out.t));
    out.fwd = (strand != 1);   // src/utils.cpp:33-41: 0 forward, 1 reverse, anything else both
    out.rev = (strand != 0);   // kaori/utils.hpp:33-39
    ScgTemplate& t = out.t;
    t.len = static_cast<int32_t>(len);

    int L = t.len;
    int nreg = 0, nconst = 0;
    int starts[SCG_MAX_TEMPLATE], ends[SCG_MAX_TEMPLATE];
    for (int i = 0; i < L; ++i) {
        char b = constant[i];
        if (b == '-') {
            if (nreg && ends[nreg - 1] == i) {
                ++ends[nreg - 1];
            } else {
                starts[nreg] = i; ends[nreg] = i + 1; ++nreg;
            }
        } else {
            int c = base_code(b);
            if (c < 0) {
                if (out.fwd) {
                    throw Error(SCG_ERR_INVALID, std::string("unknown base '") + b + "'");          // kaori/utils.hpp:156-158
                } else {
                    throw Error(SCG_ERR_INVALID, std::string("cannot complement unknown base '") + b + "'");  // :117
                }
            }
            t.fpos[nconst] = static_cast<uint8_t>(i);
            t.fcode[nconst] = static_cast<uint8_t>(c);
            ++nconst;
        }
    }
    t.nconst = nconst;
    t.nreg = nreg;
    // reverse-complemented template: position i of it is the complement of position L-1-i
    for (int k = 0; k < nconst; ++k) {
        int src = nconst - 1 - k;
        t.rpos[k] = static_cast<uint8_t>(L - 1 - t.fpos[src]);
        t.rcode[k] = static_cast<uint8_t>(t.fcode[src] ^ 2);
    }
    // Callers reject any template whose region count is not the one they expect, so only
    // region counts the engine can use are materialised.
    if (nreg <= SCG_MAX_REGIONS) {
        for (int r = 0; r < nreg; ++r) {
            t.fstart[r] = starts[r];
            t.flen[r] = ends[r] - starts[r];
            int src = nreg - 1 - r;             // kaori/ScanTemplate.hpp:82-94
            t.rstart[r] = L - ends[src];
            t.rlen[r] = ends[src] - starts[src];
        }
    }
    return out;
}

int pool_length(const char* const* pool, int32_t n) {
    // src/utils.cpp:5-23
    size_t size = 0;
    for (int32_t i = 0; i < n; ++i) {
        size_t cur = std::strlen(pool[i]);
        if (i == 0) {
            size = cur;
        } else if (cur != size) {
            throw Error(SCG_ERR_INVALID, "variable regions should all have the same length (" + std::to_string(size) + ")");
        }
    }
    return static_cast<int>(size);
}

namespace {

// fn(0) .. fn(n - 1), each on its own thread (n <= 6 position groups).
template<class F>
void for_each_group(int n, F fn) {
    std::exception_ptr err;
    std::mutex mu;
    auto guarded = [&](int i) {
        try {
            fn(i);
        } catch (...) {
            std::lock_guard<std::mutex> g(mu);
            if (!err) err = std::current_exception();
        }
    };
    std::vector<std::thread> th;
    for (int i = 1; i < n; ++i) th.emplace_back(guarded, i);
    if (n > 0) guarded(0);
    for (auto& t : th) t.join();
    if (err) std::rethrow_exception(err);
}

// Lays the concrete entries (key, value) out as a segment index (scg_common.h: ScgIndex).
void finish_index(HostIndex& X, const std::vector<uint64_t>& keys, const std::vector<int32_t>& vals, int32_t len, int max_mm) {
    const size_t n = keys.size();
    X.len = len;
    X.n_entries = static_cast<int32_t>(n);
    // position groups (see ScgIndex): masks over `parts` equal slices of the barcode
    auto slice = [&](int part, int parts) -> uint64_t {
        int a = static_cast<int>(static_cast<int64_t>(part) * len / parts);
        int b = static_cast<int>(static_cast<int64_t>(part + 1) * len / parts);
        uint64_t m32 = (b - a >= 32) ? 0xFFFFFFFFull : (((1ull << (b - a)) - 1ull) << a);
        return m32 | (m32 << 32);
    };
    std::vector<uint64_t> groups;
    if (max_mm == 0) {
        groups.push_back(slice(0, 1));
        X.nwalk[0] = X.nwalk[1] = X.nwalk[2] = X.nwalk[3] = 1;
    } else if (max_mm == 1) {
        groups.push_back(slice(0, 2));
        groups.push_back(slice(1, 2));
        X.nwalk[0] = 1; X.nwalk[1] = X.nwalk[2] = X.nwalk[3] = 2;
    } else if (max_mm == 2) {
        static const int pairs[6][2] = {{0, 1}, {2, 3}, {0, 2}, {1, 3}, {0, 3}, {1, 2}};
        for (auto& pr : pairs) groups.push_back(slice(pr[0], 4) | slice(pr[1], 4));
        X.nwalk[0] = 1; X.nwalk[1] = 2; X.nwalk[2] = X.nwalk[3] = 6;
    } else if (max_mm == 3) {
        for (int q = 0; q < 4; ++q) groups.push_back(slice(q, 4));
        X.nwalk[0] = 1; X.nwalk[1] = 2; X.nwalk[2] = 3; X.nwalk[3] = 4;
    }                          // wider budgets: no tables (nseg = 0), dense scans of the node array
    const int nseg = static_cast<int>(groups.size());
    X.nseg = nseg;
    const int ncopies = nseg > 0 ? nseg : 1;
    X.nodes.resize(static_cast<size_t>(ncopies) * n * 4);
    for (int c = 0; c < ncopies; ++c) {
        uint32_t* node = X.nodes.data() + static_cast<size_t>(c) * n * 4;
        for (size_t e = 0; e < n; ++e) {
            node[4 * e] = static_cast<uint32_t>(keys[e]);
            node[4 * e + 1] = static_cast<uint32_t>(keys[e] >> 32);
            node[4 * e + 2] = static_cast<uint32_t>(vals[e]);
            node[4 * e + 3] = 0xFFFFFFFFu;          // chain end until linked below
        }
    }
    if (nseg == 0) return;

    // One table per group, open addressing with linear probing, one slot count for all (load: table_factor).  Entries
    // agreeing on the group's positions form a chain in ascending entry order whose head sits in the slot itself: going
    // through the entries back to front, each either claims a free slot or becomes the new head of the chain it finds.
    // The groups are independent and are built side by side (the build overlaps the first window of a file only).
    uint32_t cap = 16;
    while (cap < n * table_factor(n)) cap <<= 1;
    X.slot_mask = cap - 1;
    X.tables.resize(static_cast<size_t>(nseg) * cap * 4);
    for (int sgm = 0; sgm < nseg; ++sgm) X.segmask[sgm] = groups[sgm];
    for_each_group(nseg, [&](int sgm) {
        const uint64_t mask = groups[sgm];
        const uint32_t slot_mask = X.slot_mask;
        uint32_t* node = X.nodes.data() + static_cast<size_t>(sgm) * n * 4;
        uint32_t* table = X.tables.data() + static_cast<size_t>(sgm) * cap * 4;
        std::vector<int32_t> placed(cap, -1);
        for (size_t i = n; i-- > 0;) {
            const uint64_t sk = keys[i] & mask;
            uint32_t pos = scg_hash64(sk) & slot_mask;
            for (;;) {
                const int32_t e = placed[pos];
                if (e < 0) { placed[pos] = static_cast<int32_t>(i); break; }
                if ((keys[e] & mask) == sk) {
                    node[4 * i + 3] = static_cast<uint32_t>(e);
                    placed[pos] = static_cast<int32_t>(i);
                    break;
                }
                pos = (pos + 1) & slot_mask;
            }
        }
        for (uint32_t pos = 0; pos < cap; ++pos) {
            const int32_t e = placed[pos];
            if (e < 0) {
                table[4 * pos] = table[4 * pos + 1] = table[4 * pos + 2] = 0;
                table[4 * pos + 3] = SCG_SLOT_EMPTY;
            } else {
                for (int w = 0; w < 4; ++w) table[4 * pos + w] = node[4 * static_cast<size_t>(e) + w];
            }
        }
    });
}

} // namespace

HostIndex build_index(const char* const* pool, int32_t n, int32_t len, int max_mm) {
    check_len(len);
    int64_t total = count_expansions(pool, n, len);
    std::vector<uint64_t> hk;
    std::vector<int32_t> hv;
    Builder B(hk, hv, capacity_for(total));     // host-only exact map, for duplicate detection
    int32_t sentinel = -1;                      // owner of the one key equal to SCG_EMPTY_KEY (32 x G)
    std::vector<uint64_t> keys;
    std::vector<int32_t> vals;
    keys.reserve(total);
    vals.reserve(total);
    for (int32_t i = 0; i < n; ++i) {
        for_each_expansion(pool[i], len, [&](uint64_t key) {
            if (key == SCG_EMPTY_KEY) {
                if (sentinel >= 0) throw_duplicate(sentinel, i);
                sentinel = i;
            } else {
                bool existed;
                uint32_t slot = B.upsert(key, i, &existed);
                if (existed) throw_duplicate(hv[slot], i);
            }
            keys.push_back(key);
            vals.push_back(i);
        });
    }
    HostIndex X;
    finish_index(X, keys, vals, len, max_mm);
    return X;
}

// ---------------------------------------------------------------------------------------------
// Wide keys (33..64 bases, or several regions concatenated: N = 1 word of 64 bits per plane) and big keys (65..256
// bases: N = SCG_BIG_WORDS): the same index.  A node / slot is {lo words}{hi words}{value, next}, padded to whole uint4:
// 8 and 20 32-bit words.
// ---------------------------------------------------------------------------------------------
namespace {

template<int N>
struct KeyN {
    uint64_t lo[N], hi[N];
    bool operator==(const KeyN& o) const {
        for (int k = 0; k < N; ++k) if (lo[k] != o.lo[k] || hi[k] != o.hi[k]) return false;
        return true;
    }
};
template<int N>
struct KeyNHash {
    size_t operator()(const KeyN<N>& k) const {
        size_t h = 0;
        for (int w = 0; w < N; ++w) {
            h = h * 0x9E3779B97F4A7C15ull + ((static_cast<size_t>(scg_hash64(k.lo[w])) << 32) ^ scg_hash64(k.hi[w]) ^ (k.lo[w] * 0x9E3779B97F4A7C15ull));
        }
        return h;
    }
};
template<int N> constexpr int node_words() { return N == 1 ? 8 : 20; }
template<int N> uint32_t group_hash(const KeyN<N>& k);
template<> uint32_t group_hash<1>(const KeyN<1>& k) { return scg_hash128(k.lo[0], k.hi[0]); }
template<> uint32_t group_hash<SCG_BIG_WORDS>(const KeyN<SCG_BIG_WORDS>& k) { return scg_hash_big(k.lo, k.hi); }

template<int N, class F>
void for_each_expansion_n(const char* s, int len, F f) {
    constexpr int MAXLEN = 64 * N;
    int cnt[MAXLEN], codes[MAXLEN][4], choice[MAXLEN];
    for (int p = 0; p < len; ++p) {
        cnt[p] = iupac_codes(s[p], codes[p]);
        choice[p] = 0;
    }
    for (;;) {
        KeyN<N> key;
        std::memset(&key, 0, sizeof(key));
        for (int p = 0; p < len; ++p) {
            uint64_t c = static_cast<uint64_t>(codes[p][choice[p]]);
            key.lo[p >> 6] |= (c & 1) << (p & 63);
            key.hi[p >> 6] |= (c >> 1) << (p & 63);
        }
        f(key);
        int p = len - 1;
        for (; p >= 0; --p) {
            if (++choice[p] < cnt[p]) break;
            choice[p] = 0;
        }
        if (p < 0) break;
    }
}

template<int N>
void put_node(uint32_t* node, const KeyN<N>& k, uint32_t val, uint32_t next) {
    for (int w = 0; w < N; ++w) {
        node[2 * w] = static_cast<uint32_t>(k.lo[w]); node[2 * w + 1] = static_cast<uint32_t>(k.lo[w] >> 32);
        node[2 * N + 2 * w] = static_cast<uint32_t>(k.hi[w]); node[2 * N + 2 * w + 1] = static_cast<uint32_t>(k.hi[w] >> 32);
    }
    node[4 * N] = val; node[4 * N + 1] = next;
    for (int w = 4 * N + 2; w < node_words<N>(); ++w) node[w] = 0;
}

template<int N>
void check_len_n(int len) {
    if (len > 64 * N) {
        throw Error(SCG_ERR_UNSUPPORTED, "variable regions longer than " + std::to_string(64 * N) + " bp in total are not supported by this engine (got " +
                                         std::to_string(len) + ")");
    }
}

// A group of key positions: up to two ranges [a, b).
struct Group { int a0, b0, a1, b1; };

template<int N>
KeyN<N> masked(const KeyN<N>& k, const uint64_t mask[N]) {
    KeyN<N> r;
    for (int w = 0; w < N; ++w) { r.lo[w] = k.lo[w] & mask[w]; r.hi[w] = k.hi[w] & mask[w]; }
    return r;
}

// Tables and node copies of a wide / big index over concrete keys with their values.
template<int N>
void finish_index_n(HostIndex& X, const std::vector<KeyN<N> >& keys, const std::vector<int32_t>& vals, int32_t len, int max_mm) {
    constexpr int NW = node_words<N>();
    X.wide = N == 1 ? 1 : 2;
    X.len = len;
    const size_t cnt = keys.size();
    X.n_entries = static_cast<int32_t>(cnt);
    auto slice = [&](int part, int parts, int& a, int& b) {
        a = static_cast<int>(static_cast<int64_t>(part) * len / parts);
        b = static_cast<int>(static_cast<int64_t>(part + 1) * len / parts);
    };
    std::vector<Group> groups;              // same position groups as finish_index
    auto one = [&](int part, int parts) { Group g{0, 0, 0, 0}; slice(part, parts, g.a0, g.b0); groups.push_back(g); };
    if (max_mm == 0) {
        one(0, 1);
        X.nwalk[0] = X.nwalk[1] = X.nwalk[2] = X.nwalk[3] = 1;
    } else if (max_mm == 1) {
        one(0, 2);
        one(1, 2);
        X.nwalk[0] = 1; X.nwalk[1] = X.nwalk[2] = X.nwalk[3] = 2;
    } else if (max_mm == 2) {
        static const int pairs[6][2] = {{0, 1}, {2, 3}, {0, 2}, {1, 3}, {0, 3}, {1, 2}};
        for (auto& pr : pairs) {
            Group g;
            slice(pr[0], 4, g.a0, g.b0);
            slice(pr[1], 4, g.a1, g.b1);
            groups.push_back(g);
        }
        X.nwalk[0] = 1; X.nwalk[1] = 2; X.nwalk[2] = X.nwalk[3] = 6;
    } else if (max_mm == 3) {
        for (int q = 0; q < 4; ++q) one(q, 4);
        X.nwalk[0] = 1; X.nwalk[1] = 2; X.nwalk[2] = 3; X.nwalk[3] = 4;
    }
    const int nseg = static_cast<int>(groups.size());
    X.nseg = nseg;
    const int ncopies = nseg > 0 ? nseg : 1;
    X.nodes.resize(static_cast<size_t>(ncopies) * cnt * NW);
    for (int c = 0; c < ncopies; ++c) {
        uint32_t* node = X.nodes.data() + static_cast<size_t>(c) * cnt * NW;
        for (size_t e = 0; e < cnt; ++e) put_node<N>(node + NW * e, keys[e], static_cast<uint32_t>(vals[e]), 0xFFFFFFFFu);
    }
    if (nseg == 0) return;
    uint32_t cap = 16;
    while (cap < cnt * table_factor(cnt)) cap <<= 1;
    X.slot_mask = cap - 1;
    X.tables.resize(static_cast<size_t>(nseg) * cap * NW);
    std::vector<std::array<uint64_t, N> > masks(nseg);
    for (int sgm = 0; sgm < nseg; ++sgm) {
        const Group& g = groups[sgm];
        const uint64_t ranges = static_cast<uint64_t>(g.a0) | (static_cast<uint64_t>(g.b0) << 16) | (static_cast<uint64_t>(g.a1) << 32) | (static_cast<uint64_t>(g.b1) << 48);
        uint64_t m[SCG_BIG_WORDS];
        scg_big_group(ranges, m);
        for (int w = 0; w < N; ++w) masks[sgm][w] = m[w];
        // wide: the 64-bit position mask itself; big: the ranges (ScgIndex::segmask)
        X.segmask[sgm] = N == 1 ? m[0] : ranges;
    }
    for_each_group(nseg, [&](int sgm) {             // as finish_index
        const uint64_t* mask = masks[sgm].data();
        const uint32_t slot_mask = X.slot_mask;
        uint32_t* node = X.nodes.data() + static_cast<size_t>(sgm) * cnt * NW;
        uint32_t* table = X.tables.data() + static_cast<size_t>(sgm) * cap * NW;
        std::vector<int32_t> placed(cap, -1);
        for (size_t i = cnt; i-- > 0;) {
            const KeyN<N> sk = masked<N>(keys[i], mask);
            uint32_t pos = group_hash<N>(sk) & slot_mask;
            for (;;) {
                const int32_t e = placed[pos];
                if (e < 0) { placed[pos] = static_cast<int32_t>(i); break; }
                if (masked<N>(keys[e], mask) == sk) {
                    node[NW * i + 4 * N + 1] = static_cast<uint32_t>(e);
                    placed[pos] = static_cast<int32_t>(i);
                    break;
                }
                pos = (pos + 1) & slot_mask;
            }
        }
        for (uint32_t pos = 0; pos < cap; ++pos) {
            const int32_t e = placed[pos];
            if (e < 0) {
                for (int w = 0; w < NW; ++w) table[NW * pos + w] = 0;
                table[NW * pos + 4 * N + 1] = SCG_SLOT_EMPTY;
            } else {
                for (int w = 0; w < NW; ++w) table[NW * pos + w] = node[NW * static_cast<size_t>(e) + w];
            }
        }
    });
}

template<int N>
HostIndex build_index_n(const char* const* pool, int32_t n, int32_t len, int max_mm) {
    check_len_n<N>(len);
    count_expansions(pool, n, len);
    std::unordered_map<KeyN<N>, int32_t, KeyNHash<N> > owner;
    std::vector<KeyN<N> > keys;
    std::vector<int32_t> vals;
    for (int32_t i = 0; i < n; ++i) {
        for_each_expansion_n<N>(pool[i], len, [&](const KeyN<N>& key) {
            auto ins = owner.emplace(key, i);
            if (!ins.second) throw_duplicate(ins.first->second, i);
            keys.push_back(key);
            vals.push_back(i);
        });
    }
    HostIndex X;
    finish_index_n<N>(X, keys, vals, len, max_mm);
    return X;
}

template<int N>
HostIndex build_uid_index_n(const char* const* pool, int32_t n, int32_t len, int max_mm,
                            std::vector<std::vector<int32_t> >& expansions, size_t& n_uid) {
    check_len_n<N>(len);
    count_expansions(pool, n, len);
    std::unordered_map<KeyN<N>, int32_t, KeyNHash<N> > uid_of;
    std::vector<KeyN<N> > keys;
    expansions.assign(n, std::vector<int32_t>());
    for (int32_t i = 0; i < n; ++i) {
        for_each_expansion_n<N>(pool[i], len, [&](const KeyN<N>& key) {
            auto ins = uid_of.emplace(key, static_cast<int32_t>(keys.size()));
            if (ins.second) keys.push_back(key);
            expansions[i].push_back(ins.first->second);
        });
    }
    std::vector<int32_t> vals(keys.size());
    for (size_t u = 0; u < keys.size(); ++u) vals[u] = static_cast<int32_t>(u);
    n_uid = keys.size();
    HostIndex X;
    finish_index_n<N>(X, keys, vals, len, max_mm);
    return X;
}

} // namespace

HostIndex build_index_wide(const char* const* pool, int32_t n, int32_t len, int max_mm) {
    return len > SCG_MAX_WIDE_BARCODE ? build_index_n<SCG_BIG_WORDS>(pool, n, len, max_mm) : build_index_n<1>(pool, n, len, max_mm);
}

// value = uid of the concrete sequence, wide keys (build_uid_index for pools of 33..64 bases, or paired with one).
HostIndex build_uid_index_wide(const char* const* pool, int32_t n, int32_t len, int max_mm,
                               std::vector<std::vector<int32_t> >& expansions, size_t& n_uid) {
    return len > SCG_MAX_WIDE_BARCODE ? build_uid_index_n<SCG_BIG_WORDS>(pool, n, len, max_mm, expansions, n_uid)
                                      : build_uid_index_n<1>(pool, n, len, max_mm, expansions, n_uid);
}

HostIndex build_index_big(const char* const* pool, int32_t n, int32_t len, int max_mm) { return build_index_n<SCG_BIG_WORDS>(pool, n, len, max_mm); }
HostIndex build_uid_index_big(const char* const* pool, int32_t n, int32_t len, int max_mm,
                              std::vector<std::vector<int32_t> >& expansions, size_t& n_uid) {
    return build_uid_index_n<SCG_BIG_WORDS>(pool, n, len, max_mm, expansions, n_uid);
}

HostIndex build_uid_index(const char* const* pool, int32_t n, int32_t len, int max_mm,
                          std::vector<std::vector<int32_t> >& expansions,
                          std::vector<uint64_t>& uid_keys) {
    check_len(len);
    int64_t total = count_expansions(pool, n, len);
    std::vector<uint64_t> hk;
    std::vector<int32_t> hv;
    Builder B(hk, hv, capacity_for(total));
    int32_t sentinel = -1;
    expansions.assign(n, std::vector<int32_t>());
    uid_keys.clear();
    for (int32_t i = 0; i < n; ++i) {
        for_each_expansion(pool[i], len, [&](uint64_t key) {
            int32_t uid;
            if (key == SCG_EMPTY_KEY) {
                if (sentinel < 0) {
                    sentinel = static_cast<int32_t>(uid_keys.size());
                    uid_keys.push_back(key);
                }
                uid = sentinel;
            } else {
                bool existed;
                uint32_t slot = B.upsert(key, static_cast<int32_t>(uid_keys.size()), &existed);
                if (!existed) uid_keys.push_back(key);
                uid = hv[slot];
            }
            expansions[i].push_back(uid);
        });
    }
    std::vector<int32_t> vals(uid_keys.size());
    for (size_t u = 0; u < uid_keys.size(); ++u) vals[u] = static_cast<int32_t>(u);
    HostIndex X;
    finish_index(X, uid_keys, vals, len, max_mm);
    return X;
}

HostPairTable build_pair_table(const std::vector<std::vector<int32_t> >& exp1, const std::vector<uint64_t>& uid_keys1,
                               const std::vector<std::vector<int32_t> >& exp2, const std::vector<uint64_t>& uid_keys2) {
    HostPairTable P;
    int64_t total = 0;
    for (size_t i = 0; i < exp1.size(); ++i) {
        total += static_cast<int64_t>(exp1[i].size()) * static_cast<int64_t>(exp2[i].size());
        if (total > MAX_EXPANSIONS) {
            throw Error(SCG_ERR_UNSUPPORTED, "barcode pairs expand to more than 2^26 concrete sequences");
        }
    }
    Builder B(P.keys, P.vals, capacity_for(total, 4));      // this one is probed by the device: a quarter full
    P.mask = B.mask;
    for (size_t i = 0; i < exp1.size(); ++i) {
        // concatenated expansions in lexicographic order: first barcode major
        for (int32_t u1 : exp1[i]) {
            for (int32_t u2 : exp2[i]) {
                uint64_t key = (static_cast<uint64_t>(static_cast<uint32_t>(u1)) << 32) | static_cast<uint32_t>(u2);
                bool existed;
                uint32_t slot = B.upsert(key, static_cast<int32_t>(i), &existed);
                if (existed) throw_duplicate(P.vals[slot], static_cast<int32_t>(i));
                if (!uid_keys1.empty() && !uid_keys2.empty()) {      // (wide pools pass none: they have no dense-scan fallback)
                    P.list_key1.push_back(uid_keys1[u1]);
                    P.list_key2.push_back(uid_keys2[u2]);
                }
                P.list_vals.push_back(static_cast<int32_t>(i));
            }
        }
    }
    P.n_entries = static_cast<int32_t>(P.list_vals.size());
    return P;
}

ScgScan build_scan(const ScgTemplate& t, int max_mm) {
    ScgScan sc;
    std::memset(&sc, 0, sizeof(sc));
    sc.len = t.len;
    sc.nreg = t.nreg;
    for (int r = 0; r < SCG_MAX_REGIONS; ++r) {
        sc.fstart[r] = t.fstart[r];
        sc.rstart[r] = t.rstart[r];
        sc.flen[r] = t.flen[r];
        sc.rlen[r] = t.rlen[r];
    }
    for (int k = 0; k < t.nconst; ++k) {
        int fp = t.fpos[k], rp = t.rpos[k];
        sc.fmask[fp >> 5] |= 1u << (fp & 31);
        sc.fplane0[fp >> 5] |= static_cast<uint32_t>(t.fcode[k] & 1) << (fp & 31);
        sc.fplane1[fp >> 5] |= static_cast<uint32_t>(t.fcode[k] >> 1) << (fp & 31);
        sc.rmask[rp >> 5] |= 1u << (rp & 31);
        sc.rplane0[rp >> 5] |= static_cast<uint32_t>(t.rcode[k] & 1) << (rp & 31);
        sc.rplane1[rp >> 5] |= static_cast<uint32_t>(t.rcode[k] >> 1) << (rp & 31);
    }
    // k + 1 disjoint groups of constant positions for a budget of k mismatches; with fewer than
    // k + 1 constant bases (or k + 1 > SCG_MAX_SEEDS) no filter is possible and every position is
    // a candidate.
    // (9 of the SCG_SEED_LEN = 10 bases a seed may have: measured -1 % on config 2, -2 % on config 3, neutral elsewhere;
    // 8 costs +11 %: a false candidate then precedes the true one in a third of the wavefronts)
    int seed_max = SCG_SEED_LEN - 1;
    if (const char* e = std::getenv("SCG_SEED_MAX")) { const int v = std::atoi(e); if (v >= 4 && v <= SCG_SEED_LEN) seed_max = v; }   // tuning aid
    auto fill = [&](ScgSeeds& S, const uint8_t* pos, const uint8_t* code) {
        int want = max_mm + 1;
        if (want > SCG_MAX_SEEDS || t.nconst < want) {
            S.nseeds = 0;
            return;
        }
        // Any k + 1 disjoint groups of constant positions satisfy the pigeonhole argument.  Candidates: runs of up to
        // seed_max constant positions (consecutive in the list of constant positions) that stay inside one block of 32
        // template positions -- the scanners take a shifted plane word from two adjacent words, and blocks 0 and 1 keep
        // the compact scanner applicable.  The longest runs win (fewest false candidates), earlier ones first.
        struct Run { int first, n, blk; };
        std::vector<Run> runs;
        for (int k = 0; k < t.nconst;) {
            const int blk = pos[k] >> 5;
            int n = 1;
            while (k + n < t.nconst && n < seed_max && (pos[k + n] >> 5) == blk) ++n;
            runs.push_back(Run{k, n, blk});
            k += n;
        }
        if (static_cast<int>(runs.size()) < want) {      // fewer runs than seeds: split the longest until there are enough
            while (static_cast<int>(runs.size()) < want) {
                size_t big = 0;
                for (size_t i = 1; i < runs.size(); ++i) if (runs[i].n > runs[big].n) big = i;
                if (runs[big].n < 2) break;
                const Run r = runs[big];
                runs[big] = Run{r.first, r.n / 2, r.blk};
                runs.insert(runs.begin() + static_cast<long>(big) + 1, Run{r.first + r.n / 2, r.n - r.n / 2, r.blk});
            }
        }
        if (static_cast<int>(runs.size()) < want) { S.nseeds = 0; return; }
        std::stable_sort(runs.begin(), runs.end(), [](const Run& x, const Run& y) { return x.n > y.n; });
        runs.resize(static_cast<size_t>(want));
        std::sort(runs.begin(), runs.end(), [](const Run& x, const Run& y) { return x.first < y.first; });
        S.nseeds = want;
        for (int i = 0; i < want; ++i) {
            const int a = runs[static_cast<size_t>(i)].first;
            ScgSeed sd;
            std::memset(&sd, 0, sizeof(sd));
            sd.len = runs[static_cast<size_t>(i)].n;
            sd.blk = runs[static_cast<size_t>(i)].blk;
            for (int c = 0; c < 4; ++c) {
                uint8_t steps[64];
                int ns = 0;
                for (int j = 0; j < sd.len; ++j) {
                    if (code[a + j] != c) continue;
                    steps[ns++] = static_cast<uint8_t>(pos[a + j] - 32 * sd.blk);      // offset within the block, < 32
                }
                if (ns & 1) { steps[ns] = steps[ns - 1]; ++ns; }      // even step count: the compact scanner folds bases in pairs (a repeated AND is a no-op)
                for (int k = 0; k < ns; ++k) sd.walk[c].w[k >> 2] |= static_cast<uint32_t>(steps[k]) << (8 * (k & 3));
                sd.nsteps |= static_cast<uint32_t>(ns) << (8 * c);
            }
            S.seed[i] = sd;
        }
    };
    fill(sc.fseeds, t.fpos, t.fcode);
    fill(sc.rseeds, t.rpos, t.rcode);
    // The compact kernels keep only the candidate words and read shifted plane words straight from the planes: valid
    // while every seed lies in block 0 or 1.
    sc.compact_ok = 1;
    for (const ScgSeeds* S : {&sc.fseeds, &sc.rseeds}) {
        for (int i = 0; i < S->nseeds; ++i) if (S->seed[i].blk > 1) sc.compact_ok = 0;
    }
    return sc;
}

} // namespace scg
