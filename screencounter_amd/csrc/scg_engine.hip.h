// scg_engine.hip.h -- device-side building blocks shared by the counting kernels:
// base decoding, constant-region check, variable-region packing and the library matcher.
//
// These are the device counterparts of (paths relative to inst/include/kaori/ in the reference):
//   const_mismatches  <- ScanTemplate::strand_match            ScanTemplate.hpp:233-252
//   pack_region       <- std::string(start + first, ...)       SimpleSingleMatch.hpp:173-185
//   index_match       <- SimpleBarcodeSearch::search           BarcodeSearch.hpp:243-251
//                        + AnyMismatches::search               MismatchTrie.hpp:446-501
//   pair_match        <- SegmentedBarcodeSearch<2>::search     BarcodeSearch.hpp:478-487
//                        + SegmentedMismatches<2>::search      MismatchTrie.hpp:577-660
// The data structures are different by design (a pigeonhole segment index over flat arrays
// instead of a pointer trie and per-thread caches); the results are the same unique-minimum
// answers, cache-free.
#ifndef SCG_ENGINE_HIP_H
#define SCG_ENGINE_HIP_H

#include <hip/hip_runtime.h>
#include "scg_common.h"

namespace scgdev {

// ---------------------------------------------------------------------------------------------
// Bases
// ---------------------------------------------------------------------------------------------
// 2-bit code of a read byte, or -1 when it is not one of ACGTacgt (kaori/utils.hpp:122-133).
__device__ __forceinline__ int base_code(uint32_t c) {
    uint32_t k = (c >> 1) & 3u;
    uint32_t expect = (0x47544341u >> (8u * k)) & 0xFFu;   // "ACTG"[k]
    return ((c & 0xDFu) == expect) ? (int)k : -1;
}

struct Read {
    const uint8_t* p;
    int n;
};

// One increment of counter i (see ScgCounters).
__device__ __forceinline__ void count_one(const ScgCounters& C, int64_t i) {
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    atomicAdd(C.base + ((i << C.replica_shift) | (int64_t)(gid & C.replica_mask)), 1);
}

// The same counter for many lanes at once (the barcode1-only / barcode2-only tallies of the diagnostics paths,
// `which` = 0 / 1, global index i): one atomic per wavefront carries the number of lanes whose `flag` is set, and
// goes to the wavefront's slot of ScgCounters::hot when the plan provides it.  Call from converged code.
__device__ __forceinline__ void count_flagged(const ScgCounters& C, int which, int64_t i, bool flag) {
    const uint64_t m = __ballot(flag);
    if (m == 0) return;
    const int leader = __ffsll((long long)m) - 1;
    if ((int)(threadIdx.x & 63u) == leader) {
        const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
        int32_t* dst = C.hot ? C.hot + which * SCG_HOT_SLOTS + ((gid >> 6) & (SCG_HOT_SLOTS - 1))
                             : C.base + ((i << C.replica_shift) | (int64_t)(gid & C.replica_mask));
        atomicAdd(dst, (int)__popcll(m));
    }
}

__device__ __forceinline__ Read get_read(const ScgReads& R, int64_t i) {
    Read r;
    if (R.offsets) {
        uint32_t a = R.offsets[i], b = R.offsets[i + 1];
        r.p = R.seqs + a;
        r.n = (int)(b - a);
    } else {
        r.p = R.seqs + (size_t)i * (size_t)R.fixed_len;
        r.n = R.fixed_len;
    }
    return r;
}

// ---------------------------------------------------------------------------------------------
// Constant-region check at one position, byte-wise: number of mismatching constant bases;
// counting stops as soon as `limit` is exceeded (callers only ask "<= limit, and if so how many").
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int const_mismatches(const ScgTemplate* __restrict__ T, bool reverse,
                                                const uint8_t* __restrict__ read, int p, int limit) {
    const uint8_t* pos = reverse ? T->rpos : T->fpos;
    const uint8_t* code = reverse ? T->rcode : T->fcode;
    int mm = 0;
    int nconst = T->nconst;
    for (int k = 0; k < nconst; ++k) {
        int c = base_code(read[p + pos[k]]);
        mm += (c != (int)code[k]);
        if (mm > limit) break;
    }
    return mm;
}

// ---------------------------------------------------------------------------------------------
// A variable region as a plane-split key.  `other` has bit j set where the read holds a
// non-ACGT byte; such a position mismatches every library base (MismatchTrie.hpp:452-453).
// W = uint32_t for keys of up to 32 bases (every paired-end and combinatorial path), uint64_t for
// the wide single-end variants (barcodes of 33-64 bases, concatenated dual barcodes).
// ---------------------------------------------------------------------------------------------
// A 256-bit word for the planes of keys of 65..256 bases (ScgIndex::wide == 2): the operations the key code uses, word by
// word.  Only the byte-wise general kernels are instantiated for it: such barcodes are rare and this is about reach, not
// speed.
struct Big {
    uint64_t w[SCG_BIG_WORDS];
    __device__ __forceinline__ Big() {}
    __device__ __forceinline__ Big(uint64_t x) {
        w[0] = x;
#pragma unroll
        for (int k = 1; k < SCG_BIG_WORDS; ++k) w[k] = 0;
    }
    __device__ __forceinline__ explicit operator bool() const {
        uint64_t any = 0;
#pragma unroll
        for (int k = 0; k < SCG_BIG_WORDS; ++k) any |= w[k];
        return any != 0;
    }
};
#define SCG_BIG_BINARY(OP) \
    __device__ __forceinline__ Big operator OP(const Big& a, const Big& b) { \
        Big r; \
        _Pragma("unroll") for (int k = 0; k < SCG_BIG_WORDS; ++k) r.w[k] = a.w[k] OP b.w[k]; \
        return r; \
    } \
    __device__ __forceinline__ Big& operator OP##=(Big& a, const Big& b) { a = a OP b; return a; }
SCG_BIG_BINARY(&)
SCG_BIG_BINARY(|)
SCG_BIG_BINARY(^)
#undef SCG_BIG_BINARY
__device__ __forceinline__ Big operator~(const Big& a) {
    Big r;
#pragma unroll
    for (int k = 0; k < SCG_BIG_WORDS; ++k) r.w[k] = ~a.w[k];
    return r;
}
__device__ __forceinline__ bool operator==(const Big& a, const Big& b) {
    uint64_t d = 0;
#pragma unroll
    for (int k = 0; k < SCG_BIG_WORDS; ++k) d |= a.w[k] ^ b.w[k];
    return d == 0;
}
__device__ __forceinline__ bool operator!=(const Big& a, const Big& b) { return !(a == b); }
__device__ __forceinline__ Big operator<<(const Big& a, int n) {          // 0 <= n < 256
    Big r;
    const int q = n >> 6, sh = n & 63;
#pragma unroll
    for (int k = 0; k < SCG_BIG_WORDS; ++k) {
        uint64_t v = 0;
#pragma unroll
        for (int j = 0; j < SCG_BIG_WORDS; ++j) {
            if (j == k - q) v |= a.w[j] << sh;
            if (j == k - q - 1 && sh) v |= a.w[j] >> (64 - sh);
        }
        r.w[k] = v;
    }
    return r;
}
__device__ __forceinline__ Big operator>>(const Big& a, int n) {          // 0 <= n < 256
    Big r;
    const int q = n >> 6, sh = n & 63;
#pragma unroll
    for (int k = 0; k < SCG_BIG_WORDS; ++k) {
        uint64_t v = 0;
#pragma unroll
        for (int j = 0; j < SCG_BIG_WORDS; ++j) {
            if (j == k + q) v |= a.w[j] >> sh;
            if (j == k + q + 1 && sh) v |= a.w[j] << (64 - sh);
        }
        r.w[k] = v;
    }
    return r;
}

template<class W>
struct QueryT {
    W lo, hi;            // code bit planes
    W other;             // non-ACGT positions
    int n_other;
};
using Query = QueryT<uint32_t>;

__device__ __forceinline__ uint32_t low_mask(int len) { return len >= 32 ? 0xFFFFFFFFu : ((1u << len) - 1u); }
template<class W> __device__ __forceinline__ W low_mask_w(int len);
template<> __device__ __forceinline__ uint32_t low_mask_w<uint32_t>(int len) { return low_mask(len); }
template<> __device__ __forceinline__ uint64_t low_mask_w<uint64_t>(int len) { return len >= 64 ? ~0ull : ((1ull << len) - 1ull); }

template<> __device__ __forceinline__ Big low_mask_w<Big>(int len) {
    Big r;
#pragma unroll
    for (int k = 0; k < SCG_BIG_WORDS; ++k) {
        const int n = len - 64 * k;
        r.w[k] = n >= 64 ? ~0ull : (n <= 0 ? 0ull : ((1ull << n) - 1ull));
    }
    return r;
}

__device__ __forceinline__ int popcount_w(uint32_t x) { return __popc(x); }
__device__ __forceinline__ int popcount_w(uint64_t x) { return __popcll(x); }
__device__ __forceinline__ uint32_t bitrev_w(uint32_t x) { return __brev(x); }
__device__ __forceinline__ uint64_t bitrev_w(uint64_t x) { return __brevll(x); }
__device__ __forceinline__ int popcount_w(const Big& x) {
    int n = 0;
#pragma unroll
    for (int k = 0; k < SCG_BIG_WORDS; ++k) n += __popcll(x.w[k]);
    return n;
}
__device__ __forceinline__ Big bitrev_w(const Big& x) {
    Big r;
#pragma unroll
    for (int k = 0; k < SCG_BIG_WORDS; ++k) r.w[k] = __brevll(x.w[SCG_BIG_WORDS - 1 - k]);
    return r;
}

// Reverse complement of a plane-split region: reverse the bit order, complement = code ^ 2
// (flips plane 1 only).  Equivalent to kaori indexing reverse-complemented barcodes
// (BarcodeSearch.hpp:36-43): Hamming distance is invariant under joint reverse complement.
template<class W>
__device__ __forceinline__ QueryT<W> reverse_complement(const QueryT<W>& q, int len) {
    QueryT<W> r;
    const int sh = (int)(8 * sizeof(W)) - len;
    const W m = low_mask_w<W>(len);
    r.lo = bitrev_w(q.lo) >> sh;
    r.hi = (~(bitrev_w(q.hi) >> sh)) & m;
    r.other = bitrev_w(q.other) >> sh;
    r.hi &= ~r.other;
    r.n_other = q.n_other;
    return r;
}

// Byte-wise packing straight from global memory (general engine).
template<class W = uint32_t>
__device__ __forceinline__ QueryT<W> pack_region(const uint8_t* __restrict__ s, int len, bool reverse) {
    QueryT<W> q;
    q.lo = 0; q.hi = 0; q.other = 0;
    for (int j = 0; j < len; ++j) {
        int c = base_code(s[j]);
        if (c < 0) {
            q.other |= (W)1 << j;
        } else {
            q.lo |= (W)(c & 1) << j;
            q.hi |= (W)(c >> 1) << j;
        }
    }
    q.n_other = popcount_w(q.other);
    return reverse ? reverse_complement(q, len) : q;
}

template<class W>
__device__ __forceinline__ int query_distance(const QueryT<W>& q, W elo, W ehi, W lenmask) {
    W mism = ((q.lo ^ elo) | (q.hi ^ ehi)) & ~q.other & lenmask;
    return popcount_w(mism) + q.n_other;
}

// One 16-byte access.  The empty asm pins all four words right after the load: left alone, the
// compiler sinks the words it needs "later" (value, link) into separate dword loads behind the
// branches that use them, which triples the number of cache requests per lookup.
__device__ __forceinline__ uint4 load_node(const uint4* p) {
    uint4 v = *p;
    asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w));
    return v;
}

// ---------------------------------------------------------------------------------------------
// Node access per key width.  Narrow: one uint4 {lo, hi, value, next}.  Wide: two uint4,
// {lo.lo32, lo.hi32, hi.lo32, hi.hi32} and {value, next, 0, 0}.  segmask: plane-split 2 x 32 bits
// (narrow) or one 64-bit position mask for both planes (wide).
// ---------------------------------------------------------------------------------------------
template<class W> struct KeyOps;
template<> struct KeyOps<uint32_t> {
    struct Node { uint32_t lo, hi; int val, next; };
    static constexpr int STRIDE = 1;         // uint4 per node / slot
    static __device__ __forceinline__ Node load(const uint4* base, size_t i) {
        const uint4 v = load_node(base + i);
        return Node{v.x, v.y, (int)v.z, (int)v.w};
    }
    static __device__ __forceinline__ uint32_t plane_mask(uint64_t m, int plane) { return plane ? (uint32_t)(m >> 32) : (uint32_t)m; }
    static __device__ __forceinline__ uint32_t hash(uint32_t lo, uint32_t hi) { return scg_hash64(((uint64_t)hi << 32) | lo); }
};
template<> struct KeyOps<uint64_t> {
    struct Node { uint64_t lo, hi; int val, next; };
    static constexpr int STRIDE = 2;
    static __device__ __forceinline__ Node load(const uint4* base, size_t i) {
        const uint4 a = load_node(base + 2 * i), b = load_node(base + 2 * i + 1);
        return Node{((uint64_t)a.y << 32) | a.x, ((uint64_t)a.w << 32) | a.z, (int)b.x, (int)b.y};
    }
    static __device__ __forceinline__ uint64_t plane_mask(uint64_t m, int) { return m; }
    static __device__ __forceinline__ uint32_t hash(uint64_t lo, uint64_t hi) { return scg_hash128(lo, hi); }
};
template<> struct KeyOps<Big> {
    struct Node { Big lo, hi; int val, next; };
    static constexpr int STRIDE = 5;         // {lo x 4 words}{hi x 4 words}{value, next, 0, 0}
    static __device__ __forceinline__ Node load(const uint4* base, size_t i) {
        Node n;
        const uint4* p = base + 5 * i;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const uint4 a = load_node(p + k), b = load_node(p + 2 + k);
            n.lo.w[2 * k] = ((uint64_t)a.y << 32) | a.x; n.lo.w[2 * k + 1] = ((uint64_t)a.w << 32) | a.z;
            n.hi.w[2 * k] = ((uint64_t)b.y << 32) | b.x; n.hi.w[2 * k + 1] = ((uint64_t)b.w << 32) | b.z;
        }
        const uint4 c = load_node(p + 4);
        n.val = (int)c.x; n.next = (int)c.y;
        return n;
    }
    static __device__ __forceinline__ Big plane_mask(uint64_t ranges, int) {
        Big m;
        scg_big_group(ranges, m.w);
        return m;
    }
    static __device__ __forceinline__ uint32_t hash(const Big& lo, const Big& hi) { return scg_hash_big(lo.w, hi.w); }
};

// ---------------------------------------------------------------------------------------------
// Library index search.  Calls f(value, distance) for every entry within Hamming distance
// <= cap of the query (an entry may be reported more than once); f returns true to stop.
// ---------------------------------------------------------------------------------------------
//
// best (optional): the caller's running minimum distance (> cap while nothing is found).  When given, the walk is
// ADAPTIVE: once an entry at distance b is known, every entry that can still matter (distance <= b: a closer one,
// or a tie) is reachable through the first nwalk[b] tables, so the later tables are skipped -- a query one mismatch
// away from its barcode looks into two tables of a budget-2 index, not six.
template<class W, class F>
__device__ __forceinline__ void index_search(const ScgIndex& X, const QueryT<W>& q, int cap, F f,
                                             const int* best = nullptr) {
    typedef KeyOps<W> K;
    if (q.n_other > cap) return;
    const W lm = low_mask_w<W>(X.len);
    if (X.nseg == 0) {
        // budget wider than the index supports: dense scan (rare, any budget)
        for (int e = 0; e < X.n_entries; ++e) {
            const typename K::Node ent = K::load(X.nodes, (size_t)e);
            int d = query_distance<W>(q, ent.lo, ent.hi, lm);
            if (d <= cap && f(ent.val, d)) return;
        }
        return;
    }
    // the first nwalk[cap] position groups suffice for a budget of cap (ScgIndex)
    // (selected from SGPR copies: indexing the kernel-argument array with a per-lane cap would be a vector load)
    const int w0 = __builtin_amdgcn_readfirstlane(X.nwalk[0]), w1 = __builtin_amdgcn_readfirstlane(X.nwalk[1]);
    const int w2 = __builtin_amdgcn_readfirstlane(X.nwalk[2]), w3 = __builtin_amdgcn_readfirstlane(X.nwalk[3]);
    auto tables_for = [&](int c) { return c <= 0 ? w0 : (c == 1 ? w1 : (c == 2 ? w2 : w3)); };
    int nwalk = tables_for(cap);
    const uint32_t nslots = X.slot_mask + 1u;
#pragma unroll 1
    for (int s = 0; s < nwalk; ++s) {
        const uint64_t mask = X.segmask[s];
        const W mlo = K::plane_mask(mask, 0), mhi = K::plane_mask(mask, 1);
        if (q.other & mlo) continue;              // a non-ACGT byte spoils this group
        const W sklo = q.lo & mlo, skhi = q.hi & mhi;
        uint32_t pos = K::hash(sklo, skhi) & X.slot_mask;
        const uint4* table = X.tables + (size_t)s * nslots * K::STRIDE;
        // the slot of a group key holds the head node of its chain: an exact hit is one access
        // (tried: 4-byte slots -- hash tag over head index -- so that the tables of a 100 k-barcode library stay in an
        // XCD's L2, the node fetched behind a matching tag: +12 % on configs 2 and 5; the second, dependent gather costs
        // more than the L2 misses it saves, and fewer slots per entry made it worse still; 8-byte slots with key planes and
        // value in one word, no dependent access for one-member chains: +2 %, profiles/r3_packed_slots_ab.txt)
        typename K::Node ent;
        bool found = false;
        for (;;) {
            ent = K::load(table, (size_t)pos);
            if ((uint32_t)ent.next == SCG_SLOT_EMPTY) break;   // no entry shares this group with the query
            if ((ent.lo & mlo) == sklo && (ent.hi & mhi) == skhi) { found = true; break; }
            pos = (pos + 1) & X.slot_mask;
        }
        if (!found) continue;
        const uint4* nodes = X.nodes + (size_t)s * (size_t)X.n_entries * K::STRIDE;
        for (;;) {
            int d = query_distance<W>(q, ent.lo, ent.hi, lm);
            if (d <= cap && f(ent.val, d)) return;
            const int e = ent.next;
            if (e < 0) break;
            ent = K::load(nodes, (size_t)e);
        }
        if (best && *best < cap) {
            const int t = tables_for(*best);
            nwalk = t < nwalk ? t : nwalk;
        }
    }
}

// match(q, cap): the unique library entry at the minimum Hamming distance <= cap.
// index >= 0 on a hit; SCG_MISSING if nothing within cap; SCG_AMBIGUOUS on a tie between
// different entries.  `mm` receives the distance of a hit.
//
// keep_first = DuplicateAction::FIRST (MismatchTrie.hpp:109-110, :273-276, :311-314): a tie goes to
// the smallest value instead of being ambiguous.  Used by the include.invalid=TRUE path, where
// the values are sequence uids numbered in order of first appearance in the pool.
template<class W>
__device__ __forceinline__ void index_match(const ScgIndex& X, const QueryT<W>& q, int cap, int& index, int& mm,
                                            bool keep_first = false) {
    int best = cap + 1, cur = SCG_MISSING;
    index_search<W>(X, q, cap, [&](int v, int d) -> bool {
        if (d < best) { best = d; cur = v; }
        else if (d == best && cur != v) { cur = keep_first ? (v < cur ? v : cur) : SCG_AMBIGUOUS; }
        return d == 0;      // an exact entry is unique (one entry per concrete sequence)
    }, &best);
    index = cur; mm = best;
}

// ---------------------------------------------------------------------------------------------
// Pairs
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int pair_probe(const ScgPairTable& t, int u1, int u2) {
    uint64_t key = ((uint64_t)(uint32_t)u1 << 32) | (uint32_t)u2;
    uint32_t h = scg_hash64(key) & t.mask;
    for (;;) {
        uint64_t k = t.keys[h];
        if (k == key) return t.vals[h];
        if (k == SCG_EMPTY_KEY) return -1;
        h = (h + 1) & t.mask;
    }
}

// Up to K distinct neighbours (value, distance) of a query within cap; returns false on overflow.
template<int K, class W>
__device__ __forceinline__ bool index_neighbours(const ScgIndex& X, const QueryT<W>& q, int cap, int val[K], int dist[K], int& n) {
    n = 0;
    bool overflow = false;
    index_search<W>(X, q, cap, [&](int v, int d) -> bool {
        bool seen = false;
#pragma unroll
        for (int i = 0; i < K; ++i) seen |= (i < n && val[i] == v);
        if (!seen) {
            if (n == K) { overflow = true; return true; }
#pragma unroll
            for (int i = 0; i < K; ++i) {
                if (i == n) { val[i] = v; dist[i] = d; }
            }
            ++n;
        }
        return false;
    });
    return !overflow;
}

// pair_match((q1,q2),(cap1,cap2)): among valid pairs whose two halves are within their own caps,
// the unique one with the smallest total distance.
template<class W>
__device__ __forceinline__ void pair_match(const ScgIndex& X1, const ScgIndex& X2, const ScgPairTable& P,
                                           const QueryT<W>& q1, int cap1, const QueryT<W>& q2, int cap2,
                                           int& index, int& total) {
    index = SCG_MISSING; total = 0;
    if (q1.n_other > cap1 || q2.n_other > cap2) return;
    int best = cap1 + cap2 + 1, cur = SCG_MISSING;
    auto consider = [&](int u1, int d1, int u2, int d2) {
        int tot = d1 + d2;
        if (tot > best) return;
        int v = pair_probe(P, u1, u2);
        if (v >= 0) {
            if (tot < best) { best = tot; cur = v; }
            else if (cur != v) { cur = SCG_AMBIGUOUS; }
        }
    };
    // (wide keys have no dense pair list: their budgets beyond the tables fall through to index_search's dense scans)
    if (sizeof(W) >= 8 || (X1.nseg != 0 && X2.nseg != 0)) {
        // The neighbourhoods of the two halves are gathered once each (they hold one or two
        // sequences in practice) and crossed; only a pathological library overflows the small
        // arrays, in which case the halves are searched nested.
#ifndef SCG_PAIR_K
#define SCG_PAIR_K 4
#endif
        constexpr int K = SCG_PAIR_K;
        int v1[K], d1[K], n1, v2[K], d2[K], n2;
        const bool ok1 = index_neighbours<K, W>(X1, q1, cap1, v1, d1, n1);
        if (ok1 && n1 == 0) return;
        const bool ok2 = index_neighbours<K, W>(X2, q2, cap2, v2, d2, n2);
        if (ok2 && n2 == 0) return;
        if (ok1 && ok2) {
#pragma unroll
            for (int i = 0; i < K; ++i) {
#pragma unroll
                for (int j = 0; j < K; ++j) {
                    if (i < n1 && j < n2) consider(v1[i], d1[i], v2[j], d2[j]);
                }
            }
        } else {
            index_search<W>(X1, q1, cap1, [&](int u1, int e1) -> bool {
                index_search<W>(X2, q2, cap2, [&](int u2, int e2) -> bool {
                    consider(u1, e1, u2, e2);
                    return false;
                });
                return false;
            });
        }
    } else if constexpr (sizeof(W) == 4) {
        uint32_t lm1 = low_mask(X1.len), lm2 = low_mask(X2.len);
        for (int i = 0; i < P.n_entries; ++i) {
            uint64_t k1 = P.list_key1[i];
            int e1 = query_distance<uint32_t>(q1, (uint32_t)k1, (uint32_t)(k1 >> 32), lm1);
            if (e1 > cap1) continue;
            uint64_t k2 = P.list_key2[i];
            int e2 = query_distance<uint32_t>(q2, (uint32_t)k2, (uint32_t)(k2 >> 32), lm2);
            if (e2 > cap2) continue;
            int tot = e1 + e2, v = P.list_vals[i];
            if (tot < best) { best = tot; cur = v; }
            else if (tot == best && cur != v) { cur = SCG_AMBIGUOUS; }
        }
    }
    if (cur != SCG_MISSING) { index = cur; total = best; }
}

} // namespace scgdev

#endif
