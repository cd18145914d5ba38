// scg_common.h -- structures shared by the host runtime and the HIP kernels.
//
// Base encoding used everywhere on the device: code = (ascii >> 1) & 3, i.e.
//   A/a -> 0, C/c -> 1, T/t -> 2, G/g -> 3;  complement(code) = code ^ 2.
// A byte is a "standard base" (kaori/utils.hpp:122-133) iff (byte & 0xDF) == "ACTG"[code].
// Barcodes of up to 32 bases are packed 2 bits per base into one uint64 (base j at bits 2j, 2j+1).
#ifndef SCG_COMMON_H
#define SCG_COMMON_H

#include <stdint.h>

#define SCG_MAX_TEMPLATE 256   // reference: src/count_single_barcodes.cpp:37-47
#define SCG_MAX_REGIONS 2      // reference: src/count_combo_barcodes_single.cpp:44-46
#define SCG_MAX_BARCODE 32     // bases per variable region handled by the packed-key engine
#define SCG_EMPTY_KEY (~0ull)

#define SCG_MISSING (-1)
#define SCG_AMBIGUOUS (-2)

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

// Open-addressing hash table: packed concrete barcode -> value (barcode index, or sequence uid
// for dual pools).  Replaces the exact std::unordered_map + mismatch trie of
// kaori/BarcodeSearch.hpp:243-251; mismatch search enumerates the Hamming neighbourhood
// of the query instead of walking a trie (same unique-minimum semantics).
struct ScgTable {
    const uint64_t* keys;     // capacity entries, SCG_EMPTY_KEY where free
    const int32_t* vals;
    uint32_t mask;            // capacity - 1
    int32_t len;              // bases per key
    int32_t sentinel_val;     // value of the one key that equals SCG_EMPTY_KEY (32 x G), else -1
    int32_t n_entries;        // concrete sequences stored (after IUPAC expansion)
    const uint64_t* list_keys;  // the same entries as a dense list, for brute-force search
    const int32_t* list_vals;
};

// (uid1, uid2) -> valid pair index, for dual barcodes (kaori/handlers/DualBarcodesPairedEnd.hpp:138-178).
struct ScgPairTable {
    const uint64_t* keys;     // (uid1 << 32) | uid2
    const int32_t* vals;
    uint32_t mask;
    int32_t n_entries;
    // dense list of every concrete (seq1, seq2, pair index) for brute-force search at caps > 2
    const uint64_t* list_key1;
    const uint64_t* list_key2;
    const int32_t* list_vals;
};

struct ScgReads {
    const uint8_t* seqs;
    const uint32_t* offsets;  // n + 1 entries, or nullptr for fixed-length reads
    int32_t fixed_len;
};

struct ScgSingleParams {
    const ScgTemplate* tmpl;
    ScgTable table;
    int32_t max_mm;
    int32_t use_first;
    int32_t fwd, rev;
};

struct ScgComboParams {
    const ScgTemplate* tmpl;
    ScgTable table[SCG_MAX_REGIONS];
    int32_t n_pool[SCG_MAX_REGIONS];
    int32_t max_mm;
    int32_t use_first;
    int32_t fwd, rev;
};

struct ScgDualParams {
    const ScgTemplate* tmpl1;
    const ScgTemplate* tmpl2;
    ScgTable table1, table2;
    ScgPairTable pairs;
    int32_t rev1, rev2;
    int32_t max_mm1, max_mm2;
    int32_t randomized;
    int32_t use_first;
};

static inline
#ifdef __HIPCC__
__host__ __device__
#endif
uint32_t scg_hash64(uint64_t key) {
    uint32_t lo = (uint32_t)key, hi = (uint32_t)(key >> 32);
    uint32_t h = lo * 0x9E3779B1u ^ hi * 0x85EBCA77u;
    h ^= h >> 15;
    h *= 0x2C1B3C6Du;
    h ^= h >> 13;
    return h;
}

#endif
