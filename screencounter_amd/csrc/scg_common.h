// scg_common.h -- structures shared by the host runtime and the HIP kernels.
//
// Base encoding used everywhere on the device: code = (ascii >> 1) & 3, i.e.
//   A/a -> 0, C/c -> 1, T/t -> 2, G/g -> 3;  complement(code) = code ^ 2.
// A byte is a "standard base" (kaori/utils.hpp:122-133) iff (byte & 0xDF) == "ACTG"[code].
//
// Packed barcodes ("keys") are PLANE-SPLIT: for a barcode of len <= 32 bases, bit j of the low
// 32-bit word is code bit 0 of base j and bit j of the high word is code bit 1 of base j.
// The same split is used for reads staged in LDS (one bit-plane per code bit plus a validity
// plane), so a variable region is cut out of a read with two funnel shifts, reverse-complemented
// with two bit reversals, and compared with XOR / OR / popcount.
#ifndef SCG_COMMON_H
#define SCG_COMMON_H

#include <stdint.h>
#include <hip/hip_vector_types.h>

#define SCG_MAX_TEMPLATE 256   // reference: src/count_single_barcodes.cpp:37-47
#define SCG_MAX_REGIONS 8      // variable regions per template (countDualBarcodesSingleEnd concatenates them: DualBarcodesSingleEnd.hpp:144-163)
#define SCG_COMBO_REGIONS 2    // pools of countComboBarcodes; reference: src/count_combo_barcodes_single.cpp:44-46
#define SCG_MAX_BARCODE 32     // bases per key of the narrow (2 x 32-bit plane) engine
#define SCG_MAX_WIDE_BARCODE 64   // bases per key of the wide (2 x 64-bit plane) engine
#define SCG_MAX_BIG_BARCODE 256   // bases per key of the big (2 x 256-bit plane) engine: as long as the longest template
#define SCG_BIG_WORDS 4           // 64-bit words per plane of a big key
#define SCG_MAX_SEGMENTS 6     // hash tables ("segment groups") of the library index (mismatch budgets <= 3)
#define SCG_MAX_SEEDS 4        // pigeonhole seeds of the constant-region scan (budgets <= 3)
#define SCG_SEED_LEN 10        // constant bases per seed (at most)
#define SCG_EMPTY_KEY (~0ull)
#define SCG_SLOT_EMPTY 0xFFFFFFFEu
#define SCG_HOT_SLOTS 4096

#define SCG_MISSING (-1)
#define SCG_AMBIGUOUS (-2)

// Seeds for the bit-parallel constant-region scan.  With at most k mismatching constant bases,
// at least one of k+1 disjoint groups of constant positions matches exactly (pigeonhole), so the
// union of the seeds' exact-match positions is a superset of the positions the reference's
// ScanTemplate reports (kaori/ScanTemplate.hpp:233-252); every candidate is then verified exactly.
// Packed into dwords so that the whole description travels in the kernel-argument segment and
// lives in SGPRs (gfx950 has no scalar byte loads).
// For one (seed, base code): the seed's bases with that code, as their offsets (0..31) from template position
// 32 * ScgSeed::blk, ascending, one per byte from byte 0 of w[0] up.  Plane E[code] shifted right by such an offset is
// ANDed into the seed's match mask.  The offsets are ABSOLUTE within the block, so that a funnel shift can take a walk
// word (shifted down by whole bytes) as its shift operand directly -- the instruction reads bits 4:0 only; the
// general scanner, which moves a running copy of the plane, takes the differences.  A seed never straddles a
// 32-position block, so that the compact scanner can take a shifted plane word with one funnel shift from two adjacent words.
#define SCG_SEED_STEPS 16
struct ScgSeedWalk {
    uint32_t w[SCG_SEED_STEPS / 4];
};

struct ScgSeed {
    int32_t len;                             // bases in the seed (0 => matches everywhere)
    uint32_t nsteps;                         // byte c: number of steps of walk[c]
    int32_t blk;                             // the seed's bases all lie in template positions [32 blk, 32 blk + 32); walks are relative to 32 blk
    int32_t pad;
    ScgSeedWalk walk[4];                     // per base code
};

struct ScgSeeds {
    int32_t nseeds;                          // 0 => every position is a candidate
    int32_t pad[3];
    ScgSeed seed[SCG_MAX_SEEDS];
};

// Everything the staged scan of one template needs, by value in the kernel arguments.
struct ScgScan {
    int32_t len;                              // template length T
    int32_t nreg;
    int32_t fstart[SCG_MAX_REGIONS];          // variable-region starts, forward template
    int32_t rstart[SCG_MAX_REGIONS];          // ... on the reverse-complemented template
    int32_t flen[SCG_MAX_REGIONS];            // region lengths, forward order
    int32_t rlen[SCG_MAX_REGIONS];            // ... in the order they appear on the reverse-complemented template
    ScgSeeds fseeds, rseeds;
    int32_t compact_ok;                       // every seed lies in block 0 or 1 (template positions < 64)
    int32_t pad[3];
    // bit planes of the constant bases (word w covers template positions 32w .. 32w+31)
    uint32_t fplane0[SCG_MAX_TEMPLATE / 32], fplane1[SCG_MAX_TEMPLATE / 32], fmask[SCG_MAX_TEMPLATE / 32];
    uint32_t rplane0[SCG_MAX_TEMPLATE / 32], rplane1[SCG_MAX_TEMPLATE / 32], rmask[SCG_MAX_TEMPLATE / 32];
};

// One template = constant bases + variable regions (kaori/ScanTemplate.hpp:53-95).
// Constant positions are listed explicitly so that the scanner touches nothing else.
struct ScgTemplate {
    int32_t len;                       // template length T
    int32_t nconst;                    // number of constant positions
    int32_t nreg;                      // number of variable regions
    int32_t fstart[SCG_MAX_REGIONS];   // region starts on the forward template, ascending
    int32_t rstart[SCG_MAX_REGIONS];   // region starts on the reverse-complemented template, ascending
    int32_t flen[SCG_MAX_REGIONS];     // length of forward region r
    int32_t rlen[SCG_MAX_REGIONS];     // length of reverse-scan region r (= flen[nreg-1-r])
    uint8_t fpos[SCG_MAX_TEMPLATE];    // constant positions, forward template
    uint8_t fcode[SCG_MAX_TEMPLATE];   // base code expected at fpos[k]
    uint8_t rpos[SCG_MAX_TEMPLATE];    // constant positions, reverse-complemented template
    uint8_t rcode[SCG_MAX_TEMPLATE];
};

// Library index: every concrete barcode (IUPAC codes expanded) once, reachable through several hash
// tables, each keyed by a GROUP of barcode positions (a plane-split mask).  The groups are chosen so
// that an entry within Hamming distance <= c of a query agrees with it exactly on at least one of
// the first nwalk[c] groups (pigeonhole):
//   budget 0: the whole key;            budget 1: two halves;
//   budget 2: the six pairs of four quarters, ordered {01},{23},{02},{13},{03},{12}
//             (c = 0 needs one, c = 1 the first two, c = 2 all six; 10-base keys keep chains ~1 long);
//   budget 3: four quarters (c + 1 of them);   wider budgets: no tables, dense scans.
// Walking those short chains and verifying each member by XOR + popcount finds every neighbour.
//
// Layout.  Entries that agree on group s form a chain in ascending entry order; `nodes` holds one
// copy of the entry array per table, each with that table's chain link, and the chain's head node
// is also stored IN the slot of the (open-addressed, linearly probed) hash table of group s.  One
// 16-byte access therefore reaches the head entry -- the whole answer for a read whose barcode
// matches exactly -- and following a chain is one access per further member.  The index is kept
// small and is touched as rarely as possible on purpose: measured on MI355X, a lane-divergent
// 16-byte lookup costs about as much as 64 vector instructions of one wavefront, and lookups that
// miss the 4 MB L2 of an XCD are the kernel's only HBM traffic beyond the reads themselves.
// Replaces the exact std::unordered_map + mismatch trie + per-thread caches of
// kaori/BarcodeSearch.hpp:243-251 and kaori/MismatchTrie.hpp:446-501 with the same
// unique-minimum semantics.
struct ScgIndex {
    const uint4* nodes;         // [max(nseg,1)][n_entries] x {key lo, key hi, value, next entry of table s's chain (-1 ends)}
    const uint4* tables;        // [nseg][slot_mask + 1] x head node of the chain keyed here; word 3 == SCG_SLOT_EMPTY: free slot
    uint32_t slot_mask;
    int32_t n_entries;
    int32_t len;                // bases per key
    int32_t nseg;               // number of tables; 0 => budget too wide: dense scan of `nodes`
    int32_t wide;               // 1: keys of 33..64 bases: nodes / slots are two uint4 {lo64, hi64}, {value, next, 0, 0},
                                // segmask[s] is a 64-bit position mask applied to both planes
                                // 2: keys of 65..256 bases ("big"): nodes / slots are five uint4 {lo x 4}{hi x 4}{value, next, 0, 0}
                                // (as 64-bit words), segmask[s] holds the group as two position ranges [a0, b0) and [a1, b1),
                                // 16 bits each from the low end (scg_big_group)
    int32_t nwalk[4];           // tables to walk for a query cap of 0..3
    uint64_t segmask[SCG_MAX_SEGMENTS];   // plane-split position mask of table s
};

// (uid1, uid2) -> valid pair index, for dual barcodes (kaori/handlers/DualBarcodesPairedEnd.hpp:138-178).
struct ScgPairTable {
    const uint64_t* keys;     // (uid1 << 32) | uid2, SCG_EMPTY_KEY where free
    const int32_t* vals;
    uint32_t mask;
    int32_t n_entries;
    // dense list of every concrete (seq1, seq2, pair index) for brute-force search at wide budgets
    const uint64_t* list_key1;
    const uint64_t* list_key2;
    const int32_t* list_vals;
};

// Where a kernel's atomic increments go.  Hot counters (small libraries, skewed screens, the
// barcode1-only tally) would serialise tens of millions of atomics on a few addresses, so the
// plan keeps `replicas` privatised copies of the counter array; a lane adds to the copy picked by
// its global id and a fold kernel sums the copies into the plan's counters after every launch.
// Copies of one counter are adjacent (element (i, r) at i * replicas + r), so the lanes of a wave
// that hit the same hot counter add into one contiguous 256-byte run -- the shape global atomics
// run fastest at -- and the fold reads each counter's copies as one coalesced segment.
// replica_shift == 0 means the counter array is large enough to be used directly.
struct ScgCounters {
    int32_t* base;
    uint32_t replica_mask;      // replicas - 1 (power of two)
    uint32_t replica_shift;     // log2(replicas)
    // Tally mode (single-barcode kernels, mid-sized libraries): instead of one memory-side atomic per
    // mapped read the kernel stores the barcode index of read r (or -1) in unit_index[r], and
    // tally_kernel turns the index stream into counts through LDS histograms (scg_kernels.hip).
    int32_t* unit_index;        // nullptr: count with atomics
    // Sparse mode (combination spaces beyond the dense limit: scg_sparse.hip): the combination of read / pair r, or ~0
    // for none, goes to unit_pair[r] as (first << 32) | second; nothing is added to `base` for it.
    uint64_t* unit_pair;        // nullptr: dense cells
    // Diagnostics paths: partial sums of the two single-address tallies (barcode1-only, barcode2-only), one slot
    // per wavefront modulo SCG_HOT_SLOTS, folded into their counters after the launch (hot_fold_kernel).  Even one
    // atomic per wavefront on a single address serialises (measured: 5 ms per 20 M pairs).
    int32_t* hot;               // [2][SCG_HOT_SLOTS] or nullptr
};

struct ScgReads {
    const uint8_t* seqs;
    const uint32_t* offsets;  // n + 1 entries, or nullptr for fixed-length reads
    int32_t fixed_len;
    int32_t max_len;          // hint: upper bound on the read lengths (0 = unknown); picks the LDS tile shape only
    int32_t ablate;           // measurement builds compiled with -DSCG_ABLATE only (tools/ablate_build.sh): 1 = skip phase C,
                              // 2 = skip phases B and C, 3 = everything but the counter atomics; the product library ignores it
};

struct ScgSingleParams {
    ScgScan scan;
    const ScgTemplate* tmpl;   // byte-wise form, for the general engine
    ScgIndex index;
    int32_t max_mm;
    int32_t use_first;
    int32_t fwd, rev;
};

struct ScgComboParams {
    ScgScan scan;
    const ScgTemplate* tmpl;
    ScgIndex index[SCG_COMBO_REGIONS];
    int32_t n_pool[SCG_COMBO_REGIONS];
    int32_t max_mm;
    int32_t use_first;
    int32_t fwd, rev;
    // countDualBarcodesSingleEnd(include.invalid=TRUE) runs this kernel as a second pass: only reads whose
    // entry of `only_if_negative` is < 0 (no valid combination found by the first pass) are searched, and ties
    // between barcodes go to the first (DuplicateAction::FIRST; index values are sequence uids).
    const int32_t* only_if_negative;
    int32_t keep_first;
    int32_t pad;
};

struct ScgDualParams {
    ScgScan scan1, scan2;
    const ScgTemplate* tmpl1;
    const ScgTemplate* tmpl2;
    ScgIndex index1, index2;
    ScgPairTable pairs;
    int32_t rev1, rev2;
    int32_t max_mm1, max_mm2;
    int32_t randomized;
    int32_t use_first;
    // include.invalid=TRUE (handlers/DualBarcodesPairedEndWithDiagnostics.hpp): pairs without a valid
    // combination are searched mate by mate; counters = [n_pool valid][barcode1-only][barcode2-only]
    // [n_uid1 x n_uid2 invalid combinations by sequence uid]
    // diagnostics == 2: countPairedComboBarcodes (handlers/CombinatorialBarcodesPairedEnd.hpp) -- there is no
    // list of valid pairs at all (n_pool = 0), every pair is searched mate by mate, and a tie between
    // different barcodes is ambiguous (DuplicateAction::ERROR) instead of going to the first.
    int32_t diagnostics;
    int32_t n_pool;
    int32_t n_uid2;
    int32_t keep_first;                  // mate searches: ties go to the first barcode (DuplicateAction::FIRST) instead of being ambiguous
    const int32_t* only_if_negative;     // diagnostics == 2 as a second pass: search only pairs whose entry is < 0
    // The staged pair search keeps ONE verified template hit per mate and template (dual_passes_kernel); a pair with more
    // is left to the byte-wise search of a second, small launch: overflow[0] counts such pairs, overflow[1..] lists them.
    // Room for every pair of the batch; the launcher clears the count.
    int32_t* overflow;
};

static inline
#ifdef __HIPCC__
__host__ __device__
#endif
uint32_t scg_hash64(uint64_t key) {
    // Group keys are sparse (a few bit runs of two 32-bit planes, often with the low bits clear), so
    // each plane gets a multiply and a fold of its own before the final avalanche.
    uint32_t h = (uint32_t)key * 0x9E3779B1u;
    h ^= h >> 16;
    h = (h ^ (uint32_t)(key >> 32)) * 0x85EBCA6Bu;
    h ^= h >> 13;
    h *= 0xC2B2AE35u;
    h ^= h >> 16;
    return h;
}

// Hash of a wide (2 x 64-bit plane) group key.
static inline
#ifdef __HIPCC__
__host__ __device__
#endif
uint32_t scg_hash128(uint64_t lo, uint64_t hi) {
    uint32_t a = scg_hash64(lo), b = scg_hash64(hi);
    uint32_t h = (a ^ ((b << 15) | (b >> 17))) * 0x9E3779B1u;
    return h ^ (h >> 15);
}

// Hash of a big (2 x 256-bit plane) group key.
static inline
#ifdef __HIPCC__
__host__ __device__
#endif
uint32_t scg_hash_big(const uint64_t* lo, const uint64_t* hi) {
    uint32_t h = scg_hash128(lo[0], hi[0]);
    for (int k = 1; k < SCG_BIG_WORDS; ++k) {
        h = (h ^ scg_hash128(lo[k] + k, hi[k])) * 0x85EBCA6Bu;
        h ^= h >> 13;
    }
    return h;
}

// The position mask of a big index's group (ScgIndex::segmask) as 64-bit words.
static inline
#ifdef __HIPCC__
__host__ __device__
#endif
void scg_big_group(uint64_t ranges, uint64_t mask[SCG_BIG_WORDS]) {
    for (int k = 0; k < SCG_BIG_WORDS; ++k) mask[k] = 0;
    for (int r = 0; r < 2; ++r) {
        const int a = (int)((ranges >> (32 * r)) & 0xFFFFu), b = (int)((ranges >> (32 * r + 16)) & 0xFFFFu);
        for (int k = 0; k < SCG_BIG_WORDS; ++k) {
            const int lo = a > 64 * k ? a : 64 * k, hi = b < 64 * (k + 1) ? b : 64 * (k + 1);
            if (hi <= lo) continue;
            const int n = hi - lo;
            mask[k] |= (n >= 64 ? ~0ull : ((1ull << n) - 1ull)) << (lo - 64 * k);
        }
    }
}

#endif
