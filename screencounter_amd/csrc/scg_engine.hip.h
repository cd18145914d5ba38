// scg_engine.hip.h -- device-side building blocks shared by the counting kernels:
// base decoding, constant-region scan, variable-region packing and the library matcher.
//
// These are the device counterparts of (paths relative to inst/include/kaori/ in the reference):
//   const_mismatches  <- ScanTemplate::strand_match            ScanTemplate.hpp:233-252
//   pack_region       <- std::string(start + first, ...)       SimpleSingleMatch.hpp:173-185
//   table_match       <- SimpleBarcodeSearch::search           BarcodeSearch.hpp:243-251
//                        + AnyMismatches::search               MismatchTrie.hpp:446-501
//   pair_match        <- SegmentedBarcodeSearch<2>::search     BarcodeSearch.hpp:478-487
//                        + SegmentedMismatches<2>::search      MismatchTrie.hpp:577-660
// The data structures are different by design (flat hash tables + neighbourhood enumeration
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
// Constant-region scan at one position: number of mismatching constant bases, counting stops as
// soon as `limit` is exceeded (the callers only ask "<= limit, and if so how many").
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
// Variable region -> packed key.  reverse => the region is reverse-complemented so that it can be
// looked up in the forward library (equivalent to kaori indexing the reverse-complemented
// barcodes, BarcodeSearch.hpp:36-43: Hamming distance is invariant under joint RC).
// nmask has 0b11 at every position holding a non-ACGT byte; such a position mismatches every
// library base (MismatchTrie.hpp:452-453).
// ---------------------------------------------------------------------------------------------
struct Query {
    uint64_t key;
    uint64_t nmask;
    int n_other;
};

__device__ __forceinline__ Query pack_region(const uint8_t* __restrict__ s, int len, bool reverse) {
    Query q;
    q.key = 0; q.nmask = 0; q.n_other = 0;
    for (int j = 0; j < len; ++j) {
        int c = base_code(s[j]);
        int pos = reverse ? (len - 1 - j) : j;
        if (c < 0) {
            q.nmask |= 3ull << (2 * pos);
            ++q.n_other;
        } else {
            if (reverse) c ^= 2;
            q.key |= (uint64_t)c << (2 * pos);
        }
    }
    return q;
}

// ---------------------------------------------------------------------------------------------
// Hash probes
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int table_probe(const ScgTable& t, uint64_t key) {
    if (key == SCG_EMPTY_KEY) return t.sentinel_val;
    uint32_t h = scg_hash64(key) & t.mask;
    for (;;) {
        uint64_t k = t.keys[h];
        if (k == key) return t.vals[h];
        if (k == SCG_EMPTY_KEY) return -1;
        h = (h + 1) & t.mask;
    }
}

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

// ---------------------------------------------------------------------------------------------
// Neighbourhood enumeration.  Calls f(variant_key) for every sequence at Hamming distance
// exactly d (d <= 2) from the query, where each non-ACGT query position is a forced mismatch that
// may take any of the four bases.  f returns true to stop early; the function returns true if
// it was stopped.
// ---------------------------------------------------------------------------------------------
template<class F>
__device__ bool for_each_neighbour(const Query& q, int len, int d, F f) {
    int e = d - q.n_other;          // substitutions among the proper bases
    if (e < 0) return false;
    // positions of the (at most two) non-ACGT bytes
    int j0 = 0, j1 = 0;
    if (q.n_other >= 1) {
        j0 = (__ffsll((unsigned long long)q.nmask) - 1) >> 1;
        if (q.n_other >= 2) {
            uint64_t rest = q.nmask & ~(3ull << (2 * j0));
            j1 = (__ffsll((unsigned long long)rest) - 1) >> 1;
        }
    }
    uint64_t base = q.key & ~q.nmask;
    int fills = 1 << (2 * q.n_other);
    for (int fl = 0; fl < fills; ++fl) {
        uint64_t kf = base;
        if (q.n_other >= 1) kf |= (uint64_t)(fl & 3) << (2 * j0);
        if (q.n_other >= 2) kf |= (uint64_t)((fl >> 2) & 3) << (2 * j1);
        if (e == 0) {
            if (f(kf)) return true;
        } else if (e == 1) {
            for (int j = 0; j < len; ++j) {
                if ((q.nmask >> (2 * j)) & 1) continue;
                for (uint64_t x = 1; x < 4; ++x) {
                    if (f(kf ^ (x << (2 * j)))) return true;
                }
            }
        } else {
            for (int ja = 0; ja < len; ++ja) {
                if ((q.nmask >> (2 * ja)) & 1) continue;
                for (int jb = ja + 1; jb < len; ++jb) {
                    if ((q.nmask >> (2 * jb)) & 1) continue;
                    for (uint64_t xa = 1; xa < 4; ++xa) {
                        uint64_t ka = kf ^ (xa << (2 * ja));
                        for (uint64_t xb = 1; xb < 4; ++xb) {
                            if (f(ka ^ (xb << (2 * jb)))) return true;
                        }
                    }
                }
            }
        }
    }
    return false;
}

// Hamming distance between a query and a packed concrete sequence; non-ACGT positions always count.
__device__ __forceinline__ int packed_distance(const Query& q, uint64_t entry, uint64_t lenmask) {
    uint64_t diff = (q.key ^ entry) & ~q.nmask & lenmask;
    uint64_t mism = (diff | (diff >> 1)) & 0x5555555555555555ull;
    return __popcll((unsigned long long)mism) + q.n_other;
}

__device__ __forceinline__ uint64_t len_mask(int len) {
    return len >= 32 ? ~0ull : ((1ull << (2 * len)) - 1ull);
}

// ---------------------------------------------------------------------------------------------
// match(q, cap): the unique library entry at the minimum Hamming distance <= cap.
// index >= 0 on a hit; SCG_MISSING if nothing within cap; SCG_AMBIGUOUS on a tie between
// different entries.  `mm` receives the distance of a hit.
// ---------------------------------------------------------------------------------------------
__device__ inline void table_match(const ScgTable& t, const Query& q, int cap, int& index, int& mm) {
    index = SCG_MISSING; mm = 0;
    if (q.n_other > cap) return;
    int top = cap < 2 ? cap : 2;
    for (int d = q.n_other; d <= top; ++d) {
        int cur = SCG_MISSING;
        for_each_neighbour(q, t.len, d, [&](uint64_t k) -> bool {
            int v = table_probe(t, k);
            if (v >= 0) {
                if (cur == SCG_MISSING) cur = v;
                else if (cur != v) { cur = SCG_AMBIGUOUS; return true; }
            }
            return false;
        });
        if (cur != SCG_MISSING) { index = cur; mm = d; return; }
    }
    if (cap > 2) {
        // Rare wide budgets: one pass over the dense list (nothing lies within distance 2).
        uint64_t lm = len_mask(t.len);
        int best = cap + 1, cur = SCG_MISSING;
        for (int i = 0; i < t.n_entries; ++i) {
            int d = packed_distance(q, t.list_keys[i], lm);
            if (d < best) { best = d; cur = t.list_vals[i]; }
            else if (d == best && d <= cap && cur != t.list_vals[i]) { cur = SCG_AMBIGUOUS; }
        }
        if (cur != SCG_MISSING) { index = cur; mm = best; }
    }
}

// ---------------------------------------------------------------------------------------------
// pair_match((q1,q2),(cap1,cap2)): among valid pairs whose two halves are within their own caps,
// the unique one with the smallest total distance.
// ---------------------------------------------------------------------------------------------
__device__ inline void pair_match(const ScgTable& t1, const ScgTable& t2, const ScgPairTable& P,
                                  const Query& q1, int cap1, const Query& q2, int cap2,
                                  int& index, int& total) {
    index = SCG_MISSING; total = 0;
    if (q1.n_other > cap1 || q2.n_other > cap2) return;
    int best = cap1 + cap2 + 1, cur = SCG_MISSING;
    if (cap1 <= 2 && cap2 <= 2) {
        for (int d1 = q1.n_other; d1 <= cap1; ++d1) {
            for_each_neighbour(q1, t1.len, d1, [&](uint64_t k1) -> bool {
                int u1 = table_probe(t1, k1);
                if (u1 < 0) return false;
                for (int d2 = q2.n_other; d2 <= cap2; ++d2) {
                    if (d1 + d2 > best) break;
                    for_each_neighbour(q2, t2.len, d2, [&](uint64_t k2) -> bool {
                        int u2 = table_probe(t2, k2);
                        if (u2 < 0) return false;
                        int v = pair_probe(P, u1, u2);
                        if (v >= 0) {
                            int tot = d1 + d2;
                            if (tot < best) { best = tot; cur = v; }
                            else if (tot == best && cur != v) { cur = SCG_AMBIGUOUS; }
                        }
                        return false;
                    });
                }
                return false;
            });
        }
    } else {
        uint64_t lm1 = len_mask(t1.len), lm2 = len_mask(t2.len);
        for (int i = 0; i < P.n_entries; ++i) {
            int d1 = packed_distance(q1, P.list_key1[i], lm1);
            if (d1 > cap1) continue;
            int d2 = packed_distance(q2, P.list_key2[i], lm2);
            if (d2 > cap2) continue;
            int tot = d1 + d2, v = P.list_vals[i];
            if (tot < best) { best = tot; cur = v; }
            else if (tot == best && cur != v) { cur = SCG_AMBIGUOUS; }
        }
    }
    if (cur != SCG_MISSING) { index = cur; total = best; }
}

} // namespace scgdev

#endif
