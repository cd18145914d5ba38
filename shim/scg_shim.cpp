// shim/scg_shim.cpp -- the Rcpp glue a screenCounter maintainer drops into src/ in place of the bodies of
// src/count_single_barcodes.cpp, src/count_combo_barcodes_single.cpp, src/count_dual_barcodes.cpp,
// src/count_combo_barcodes_paired.cpp, src/count_dual_barcodes_single_end.cpp, src/count_random_barcodes.cpp and
// src/match_barcodes.cpp of the reference.  The exported signatures are the reference's own
// (src/RcppExports.cpp:136-145: arity 7 / 7 / 14 / 13 / 8 / 6 / 4), so R/RcppExports.R and every R/*.R file stay
// unchanged.  Each function: borrow the CHARSXP pointers, call the C ABI of include/scg.h, turn a non-zero
// return into Rcpp::stop (what END_RCPP does with the reference's std::runtime_error).
//
// R is not present in this repository's build image, so the file is compiled only for syntax there, against
// the type-level stand-in tests/fake_rcpp/Rcpp.h (tests/test_shim.py); with the real Rcpp it builds as part
// of the package: PKG_CPPFLAGS=-I<repo>/include, PKG_LIBS=-L<repo>/screencounter_amd -lscg.
#include "Rcpp.h"
#include "scg.h"
#include <algorithm>
#include <string>
#include <vector>

namespace {
// R CHARSXPs are NUL-terminated; libscg checks "all the same length" itself (src/utils.cpp:5-23).
std::vector<const char*> borrow(const Rcpp::CharacterVector& x) {
    std::vector<const char*> p(x.size());
    for (R_xlen_t i = 0; i < x.size(); ++i) p[i] = CHAR(STRING_ELT(x, i));
    return p;
}
void check(int rc, const char* err) { if (rc != SCG_OK) Rcpp::stop(err); }
}

//[[Rcpp::export(rng=false)]]
Rcpp::List count_single_barcodes(std::string path, std::string constant, int strand, Rcpp::CharacterVector pool,
                                 int mismatches, bool use_first, int nthreads) {
    auto p = borrow(pool);
    Rcpp::IntegerVector counts(pool.size());
    int total = 0;
    char err[1024];
    check(scg_count_single_barcodes(path.c_str(), constant.c_str(), strand, p.data(), (int32_t)p.size(),
                                    mismatches, use_first, nthreads, counts.begin(), &total, err, sizeof(err)), err);
    return Rcpp::List::create(counts, total);          // same shape as src/count_single_barcodes.cpp:49
}

//[[Rcpp::export(rng=false)]]
Rcpp::List count_combo_barcodes_single(std::string path, std::string constant, int strand, Rcpp::List pool,
                                       int mismatches, bool use_first, int nthreads) {
    if (pool.size() != 2) Rcpp::stop("currently expecting only 2 variable regions for single-end combinatorial barcodes");
    Rcpp::CharacterVector c0(pool[0]), c1(pool[1]);
    auto p0 = borrow(c0), p1 = borrow(c1);
    int32_t *idx = nullptr, *freq = nullptr; int64_t k = 0; int total = 0;
    char err[1024];
    check(scg_count_combo_barcodes_single(path.c_str(), constant.c_str(), strand, p0.data(), (int32_t)p0.size(),
                                          p1.data(), (int32_t)p1.size(), mismatches, use_first, nthreads,
                                          &idx, &freq, &k, &total, err, sizeof(err)), err);
    Rcpp::IntegerMatrix indices(2, k);                  // column-major 2 x K, 0-based, sorted by (first, second)
    std::copy(idx, idx + 2 * k, indices.begin());
    Rcpp::IntegerVector counts(freq, freq + k);
    scg_free(idx); scg_free(freq);
    return Rcpp::List::create(indices, counts, Rcpp::IntegerVector::create(total));   // src/count_combo_barcodes_single.cpp:32-36
}

//[[Rcpp::export(rng=false)]]
Rcpp::List count_dual_barcodes(std::string path1, std::string constant1, bool reverse1, int mismatches1, Rcpp::CharacterVector pool1,
                               std::string path2, std::string constant2, bool reverse2, int mismatches2, Rcpp::CharacterVector pool2,
                               bool randomized, bool use_first, bool diagnostics, int nthreads) {
    if (pool1.size() != pool2.size()) Rcpp::stop("both barcode pools should be of the same length");
    auto p1 = borrow(pool1), p2 = borrow(pool2);
    Rcpp::IntegerVector counts(pool1.size());
    int total = 0;
    char err[1024];
    if (diagnostics) {                                  // include.invalid=TRUE: 5-list of src/count_dual_barcodes.cpp:64-70
        int32_t *idx = nullptr, *freq = nullptr; int64_t k = 0; int b1 = 0, b2 = 0;
        check(scg_count_dual_barcodes_diagnostics(path1.c_str(), constant1.c_str(), reverse1, mismatches1, p1.data(),
                                                  path2.c_str(), constant2.c_str(), reverse2, mismatches2, p2.data(), (int32_t)p1.size(),
                                                  randomized, use_first, nthreads, counts.begin(), &idx, &freq, &k, &total, &b1, &b2,
                                                  err, sizeof(err)), err);
        Rcpp::IntegerMatrix indices(2, k);
        std::copy(idx, idx + 2 * k, indices.begin());
        Rcpp::IntegerVector invalid(freq, freq + k);
        scg_free(idx); scg_free(freq);
        return Rcpp::List::create(counts, Rcpp::List::create(indices, invalid), Rcpp::IntegerVector::create(total),
                                  Rcpp::IntegerVector::create(b1), Rcpp::IntegerVector::create(b2));
    }
    check(scg_count_dual_barcodes(path1.c_str(), constant1.c_str(), reverse1, mismatches1, p1.data(),
                                  path2.c_str(), constant2.c_str(), reverse2, mismatches2, p2.data(), (int32_t)p1.size(),
                                  randomized, use_first, diagnostics, nthreads, counts.begin(), &total, err, sizeof(err)), err);
    return Rcpp::List::create(counts, Rcpp::IntegerVector::create(total));            // src/count_dual_barcodes.cpp:47-50
}

//[[Rcpp::export(rng=false)]]
Rcpp::List count_random_barcodes(std::string path, std::string constant, int strand, int mismatches, bool use_first, int nthreads) {
    char* seqs = nullptr; int32_t* freq = nullptr; int64_t k = 0; int32_t len = 0; int total = 0;
    char err[1024];
    check(scg_count_random_barcodes(path.c_str(), constant.c_str(), strand, mismatches, use_first, nthreads,
                                    &seqs, &freq, &k, &len, &total, err, sizeof(err)), err);
    Rcpp::CharacterVector sequences(k);
    for (int64_t i = 0; i < k; ++i) sequences[i] = std::string(seqs + i * (len + 1), len);
    Rcpp::IntegerVector frequencies(freq, freq + k);
    scg_free(seqs); scg_free(freq);
    return Rcpp::List::create(Rcpp::List::create(sequences, frequencies), total);           // src/count_random_barcodes.cpp:61
}

//[[Rcpp::export(rng=false)]]
Rcpp::List count_dual_barcodes_single_end(std::string path, std::string constant, Rcpp::List pools, int strand, int mismatches,
                                          bool use_first, bool diagnostics, int nthreads) {
    std::vector<std::vector<const char*> > cols;           // one pool per variable region
    std::vector<const char* const*> rows;
    std::vector<int32_t> sizes;
    for (R_xlen_t p = 0; p < pools.size(); ++p) {
        cols.push_back(borrow(Rcpp::CharacterVector(pools[p])));
        sizes.push_back((int32_t)cols.back().size());
    }
    for (auto& c : cols) rows.push_back(c.data());
    Rcpp::IntegerVector counts(sizes.empty() ? 0 : sizes[0]);
    int total = 0;
    char err[1024];
    if (diagnostics) {                                  // 3-list of src/count_dual_barcodes_single_end.cpp:44-48
        int32_t *idx = nullptr, *freq = nullptr; int64_t k = 0;
        check(scg_count_dual_barcodes_single_end_diagnostics(path.c_str(), constant.c_str(), rows.data(), sizes.data(), (int32_t)rows.size(),
                                                             strand, mismatches, use_first, nthreads, counts.begin(), &idx, &freq, &k, &total,
                                                             err, sizeof(err)), err);
        Rcpp::IntegerMatrix indices(2, k);
        std::copy(idx, idx + 2 * k, indices.begin());
        Rcpp::IntegerVector invalid(freq, freq + k);
        scg_free(idx); scg_free(freq);
        return Rcpp::List::create(counts, Rcpp::List::create(indices, invalid), Rcpp::IntegerVector::create(total));
    }
    check(scg_count_dual_barcodes_single_end(path.c_str(), constant.c_str(), rows.data(), sizes.data(), (int32_t)rows.size(),
                                             strand, mismatches, use_first, diagnostics, nthreads, counts.begin(), &total, err, sizeof(err)), err);
    return Rcpp::List::create(counts, Rcpp::IntegerVector::create(total));     // src/count_dual_barcodes_single_end.cpp:31-34
}

//[[Rcpp::export(rng=false)]]
Rcpp::List count_combo_barcodes_paired(std::string path1, std::string constant1, bool reverse1, int mismatches1, Rcpp::CharacterVector pool1,
                                       std::string path2, std::string constant2, bool reverse2, int mismatches2, Rcpp::CharacterVector pool2,
                                       bool randomized, bool use_first, int nthreads) {
    auto p1 = borrow(pool1), p2 = borrow(pool2);
    int32_t *idx = nullptr, *freq = nullptr; int64_t k = 0; int total = 0, b1 = 0, b2 = 0;
    char err[1024];
    check(scg_count_combo_barcodes_paired(path1.c_str(), constant1.c_str(), reverse1, mismatches1, p1.data(), (int32_t)p1.size(),
                                          path2.c_str(), constant2.c_str(), reverse2, mismatches2, p2.data(), (int32_t)p2.size(),
                                          randomized, use_first, nthreads, &idx, &freq, &k, &total, &b1, &b2, err, sizeof(err)), err);
    Rcpp::IntegerMatrix indices(2, k);
    std::copy(idx, idx + 2 * k, indices.begin());
    Rcpp::IntegerVector counts(freq, freq + k);
    scg_free(idx); scg_free(freq);
    return Rcpp::List::create(indices, counts, Rcpp::IntegerVector::create(total),          // src/count_combo_barcodes_paired.cpp:47-54
                              Rcpp::IntegerVector::create(b1), Rcpp::IntegerVector::create(b2));
}

//[[Rcpp::export(rng=false)]]
Rcpp::List match_barcodes(Rcpp::CharacterVector sequences, Rcpp::CharacterVector choices, int substitutions, bool reverse) {
    auto s = borrow(sequences), c = borrow(choices);
    Rcpp::IntegerVector id(sequences.size()), mm(sequences.size());
    char err[1024];
    check(scg_match_barcodes(s.data(), (int32_t)s.size(), c.data(), (int32_t)c.size(), substitutions, reverse,
                             id.begin(), mm.begin(), err, sizeof(err)), err);
    for (R_xlen_t i = 0; i < id.size(); ++i) {          // 0-based / -1  ->  1-based / NA (src/match_barcodes.cpp:24-31)
        if (id[i] < 0) { id[i] = NA_INTEGER; mm[i] = NA_INTEGER; } else { id[i] += 1; }
    }
    return Rcpp::List::create(id, mm);
}

// ---------------------------------------------------------------------------------------------------------------
// Many files in one native call: optional additions for the matrixOf* functions.  With these three exported, e.g.
// matrixOfSingleBarcodes (R/countSingleBarcodes.R:112-126) replaces its
//     out <- bplapply(files, FUN=countSingleBarcodes, ..., BPPARAM=BPPARAM)
// by one call whose result it unpacks into the same per-file list; libscg schedules the files over the GPUs itself
// (one file at a time per device, library compiled once), so no BiocParallel worker processes are needed.
// ---------------------------------------------------------------------------------------------------------------

//[[Rcpp::export(rng=false)]]
Rcpp::List count_single_barcodes_files(Rcpp::CharacterVector paths, std::string constant, int strand, Rcpp::CharacterVector pool,
                                       int mismatches, bool use_first, int nthreads) {
    auto f = borrow(paths), p = borrow(pool);
    Rcpp::IntegerMatrix counts((int)pool.size(), paths.size());     // column f = file f
    Rcpp::IntegerVector totals(paths.size());
    char err[1024];
    check(scg_count_single_barcodes_files(f.data(), (int32_t)f.size(), constant.c_str(), strand, p.data(), (int32_t)p.size(),
                                          mismatches, use_first, nthreads, counts.begin(), totals.begin(), err, sizeof(err)), err);
    return Rcpp::List::create(counts, totals);
}

//[[Rcpp::export(rng=false)]]
Rcpp::List count_combo_barcodes_single_files(Rcpp::CharacterVector paths, std::string constant, int strand, Rcpp::List pool,
                                             int mismatches, bool use_first, int nthreads) {
    if (pool.size() != 2) Rcpp::stop("currently expecting only 2 variable regions for single-end combinatorial barcodes");
    Rcpp::CharacterVector c0(pool[0]), c1(pool[1]);
    auto f = borrow(paths), p0 = borrow(c0), p1 = borrow(c1);
    std::vector<int32_t*> idx(f.size(), nullptr), freq(f.size(), nullptr);
    std::vector<int64_t> k(f.size(), 0);
    Rcpp::IntegerVector totals(paths.size());
    char err[1024];
    check(scg_count_combo_barcodes_single_files(f.data(), (int32_t)f.size(), constant.c_str(), strand, p0.data(), (int32_t)p0.size(),
                                                p1.data(), (int32_t)p1.size(), mismatches, use_first, nthreads,
                                                idx.data(), freq.data(), k.data(), totals.begin(), err, sizeof(err)), err);
    Rcpp::List out(paths.size());                        // one count_combo_barcodes_single() result per file
    for (R_xlen_t i = 0; i < paths.size(); ++i) {
        Rcpp::IntegerMatrix indices(2, k[i]);
        std::copy(idx[i], idx[i] + 2 * k[i], indices.begin());
        Rcpp::IntegerVector counts(freq[i], freq[i] + k[i]);
        scg_free(idx[i]); scg_free(freq[i]);
        out[i] = Rcpp::List::create(indices, counts, Rcpp::IntegerVector::create(totals[i]));
    }
    return out;
}

//[[Rcpp::export(rng=false)]]
Rcpp::List count_dual_barcodes_files(Rcpp::CharacterVector paths1, std::string constant1, bool reverse1, int mismatches1, Rcpp::CharacterVector pool1,
                                     Rcpp::CharacterVector paths2, std::string constant2, bool reverse2, int mismatches2, Rcpp::CharacterVector pool2,
                                     bool randomized, bool use_first, int nthreads) {
    if (pool1.size() != pool2.size()) Rcpp::stop("both barcode pools should be of the same length");
    if (paths1.size() != paths2.size()) Rcpp::stop("'paths1' and 'paths2' should be of the same length");
    auto f1 = borrow(paths1), f2 = borrow(paths2), p1 = borrow(pool1), p2 = borrow(pool2);
    Rcpp::IntegerMatrix counts((int)pool1.size(), paths1.size());
    Rcpp::IntegerVector totals(paths1.size());
    char err[1024];
    check(scg_count_dual_barcodes_files(f1.data(), constant1.c_str(), reverse1, mismatches1, p1.data(),
                                        f2.data(), constant2.c_str(), reverse2, mismatches2, p2.data(), (int32_t)p1.size(), (int32_t)f1.size(),
                                        randomized, use_first, nthreads, counts.begin(), totals.begin(), err, sizeof(err)), err);
    return Rcpp::List::create(counts, totals);
}
