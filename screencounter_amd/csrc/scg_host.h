// scg_host.h -- internal C++ interface of the host runtime (not part of the C ABI).
#ifndef SCG_HOST_H
#define SCG_HOST_H

#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/scg.h"
#include "scg_common.h"

namespace scg {

// Carries an SCG_ERR_* code to the C boundary, where it becomes (code, message).
struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& msg) : std::runtime_error(msg), code(c) {}
};

// ---------------------------------------------------------------------------------------------
// Template: kaori/ScanTemplate.hpp:53-95.
// ---------------------------------------------------------------------------------------------
struct HostTemplate {
    ScgTemplate t;
    bool fwd, rev;
};
// strand: 0 forward, 1 reverse, 2 both.  Throws Error(SCG_ERR_INVALID) like the reference.
HostTemplate parse_template(const char* constant, int strand);

// ---------------------------------------------------------------------------------------------
// Pools: src/utils.cpp:5-23 (equal lengths), kaori/BarcodeSearch.hpp:23-60 (fill_library),
// kaori/MismatchTrie.hpp:93-205 (IUPAC expansion, DuplicateAction::ERROR).
// ---------------------------------------------------------------------------------------------
int pool_length(const char* const* pool, int32_t n);   // throws if lengths differ

struct HostIndex {
    int32_t len = 0;
    int32_t n_entries = 0;
    int32_t nseg = 0;
    int wide = 0;                    // 1: 2 x 64-bit planes, 8 words per node / slot; 2: 2 x 256-bit planes, 20 words (ScgIndex::wide)
    uint32_t slot_mask = 0;
    int32_t nwalk[4] = {0, 0, 0, 0};
    uint64_t segmask[SCG_MAX_SEGMENTS] = {0, 0, 0, 0, 0, 0};
    std::vector<uint32_t> nodes;     // [max(nseg,1)][n_entries] x 4 words: key lo, key hi, value, next in table s's chain
    std::vector<uint32_t> tables;    // [nseg][slot_mask + 1] x 4 words: head node of the chain keyed in that slot
};

// max_mm is the plan's mismatch budget for this pool: the index gets max_mm + 1 segments
// (none, i.e. dense scans, when that exceeds SCG_MAX_SEGMENTS).
// value = barcode index; two barcodes sharing one concrete sequence => Error("duplicate sequences
// detected (a, b) when constructing the trie").
HostIndex build_index(const char* const* pool, int32_t n, int32_t len, int max_mm);

// Same for keys of up to 64 bases (wide index; always used for concatenated multi-region keys) and, beyond, of up to 256
// bases (big index) -- the longest a template can be (src/count_single_barcodes.cpp:37-47).
HostIndex build_index_wide(const char* const* pool, int32_t n, int32_t len, int max_mm);
// The big index whatever the length (a pool that shares a kernel with a pool of more than 64 bases).
HostIndex build_index_big(const char* const* pool, int32_t n, int32_t len, int max_mm);
HostIndex build_uid_index_big(const char* const* pool, int32_t n, int32_t len, int max_mm,
                              std::vector<std::vector<int32_t> >& expansions, size_t& n_uid);

// value = uid of the concrete sequence (duplicates within the pool merge).  expansions[i] receives
// the uids of barcode i's concrete expansions, in lexicographic (A,C,G,T) order.
HostIndex build_uid_index(const char* const* pool, int32_t n, int32_t len, int max_mm,
                          std::vector<std::vector<int32_t> >& expansions,
                          std::vector<uint64_t>& uid_keys);

// The same with wide keys (either pool of a pair longer than 32 bases); n_uid receives the number of distinct sequences.
HostIndex build_uid_index_wide(const char* const* pool, int32_t n, int32_t len, int max_mm,
                               std::vector<std::vector<int32_t> >& expansions, size_t& n_uid);

struct HostPairTable {
    int32_t n_entries = 0;
    uint32_t mask = 0;
    std::vector<uint64_t> keys;
    std::vector<int32_t> vals;
    std::vector<uint64_t> list_key1, list_key2;
    std::vector<int32_t> list_vals;
};
HostPairTable build_pair_table(const std::vector<std::vector<int32_t> >& exp1, const std::vector<uint64_t>& uid_keys1,
                               const std::vector<std::vector<int32_t> >& exp2, const std::vector<uint64_t>& uid_keys2);

// The staged-scan description of a template for a mismatch budget: bit planes of the constant
// bases and the pigeonhole seeds of both strands.
ScgScan build_scan(const ScgTemplate& t, int max_mm);


// ---------------------------------------------------------------------------------------------
// FASTQ: kaori/FastqReader.hpp:42-110 over byteme readers.
// ---------------------------------------------------------------------------------------------
struct ReadBatch {
    std::vector<char> seqs;
    std::vector<uint64_t> offsets;   // n + 1
    int64_t size() const { return static_cast<int64_t>(offsets.size()) - 1; }
    void clear() { seqs.clear(); offsets.assign(1, 0); }
};

class FastqStream {
public:
    explicit FastqStream(const char* path);
    ~FastqStream();
    FastqStream(const FastqStream&) = delete;
    FastqStream& operator=(const FastqStream&) = delete;

    // Appends up to max_reads records / max_bytes sequence bytes to `out` (which is cleared first).
    // Returns false once the file is exhausted and nothing was appended.
    bool next_batch(ReadBatch& out, int64_t max_reads, int64_t max_bytes);

private:
    struct Impl;
    Impl* impl;
};

// ---------------------------------------------------------------------------------------------
// Parallel staging of plain (uncompressed) FASTQ files made of ordinary 4-line records.
//
// The file is mapped and cut into windows; inside a window every worker thread starts at a
// position that *looks like* a record start and parses strict 4-line records up to the next
// worker's start.  A worker's start is only trusted once its predecessor has landed on it exactly,
// and the first worker starts on a verified boundary, so by induction the result equals the
// sequential reference parse (kaori/FastqReader.hpp:42-110) whenever every record satisfies:
// '@' first, no '+' inside the sequence line, '+' line third, quality as long as the sequence.
// Anything else (multi-line records, malformed input, a missed landing) makes `unusual()` return
// true, and the caller redoes the file with the sequential FastqStream, which reproduces the
// reference's behaviour and error messages exactly.
// ---------------------------------------------------------------------------------------------
class ParallelFastq {
public:
    // Throws Error(SCG_ERR_IO) if the file cannot be opened / mapped.
    ParallelFastq(const char* path, int nthreads);
    ~ParallelFastq();
    ParallelFastq(const ParallelFastq&) = delete;
    ParallelFastq& operator=(const ParallelFastq&) = delete;

    static bool is_plain_file(const char* path);   // exists, regular, not gzip

    // Parses the next window into one batch per worker (empty batches possible).
    // Returns false when the file is exhausted or the input turned out to be unusual.
    bool next_window(std::vector<ReadBatch>& out);
    bool unusual() const;

private:
    struct Impl;
    Impl* impl;
};

int default_host_threads(int requested, int devices = 1);

} // namespace scg

#endif
