/*
 * kaori_ref.cpp -- thin C-ABI driver around the REAL reference implementation.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/README.md).  This file contains no algorithm of its
 * own: it instantiates the reference's kaori handlers exactly the way the reference's Rcpp
 * glue does (src/count_single_barcodes.cpp:11-50, src/count_combo_barcodes_single.cpp:12-70,
 * src/count_dual_barcodes.cpp:11-117, src/match_barcodes.cpp:6-37) minus the Rcpp types, so
 * that golden vectors and the "reference" CPU baseline come from kaori itself.
 *
 * It is compiled by oracle/Makefile with -I/root/reference/inst/include, reading the reference
 * headers where they lie; the only output is oracle/_ref/libkaori_ref.so (git-ignored).
 * No reference source is copied into this repository.
 */
#include <stdexcept>   // kaori/utils.hpp uses std::runtime_error without including this
#include <string>
#include <algorithm>
#include <numeric>
#include <vector>
#include <array>
#include <cstring>
#include <cstdint>
#include <cstdlib>
#include <type_traits>

#include "kaori/kaori.hpp"
#include "byteme/SomeFileReader.hpp"

namespace {

int set_err(char* err, size_t cap, const char* msg) {
    if (err && cap) {
        std::strncpy(err, msg, cap - 1);
        err[cap - 1] = '\0';
    }
    return 1;
}

kaori::BarcodePool make_pool(const char* const* pool, int n) {
    // src/utils.cpp:5-23 (format_pointers): all entries share the first one's length
    std::vector<const char*> ptrs(pool, pool + n);
    size_t len = 0;
    for (int i = 0; i < n; ++i) {
        size_t cur = std::strlen(pool[i]);
        if (i == 0) {
            len = cur;
        } else if (cur != len) {
            throw std::runtime_error("variable regions should all have the same length (" + std::to_string(len) + ")");
        }
    }
    return kaori::BarcodePool(std::move(ptrs), len);
}

kaori::SearchStrand to_strand(int strand) {
    // src/utils.cpp:33-41
    if (strand == 0) return kaori::SearchStrand::FORWARD;
    if (strand == 1) return kaori::SearchStrand::REVERSE;
    return kaori::SearchStrand::BOTH;
}

template<size_t N>
void single_(byteme::SomeFileReader& reader, const std::string& constant, int strand, const kaori::BarcodePool& pool,
             int mm, bool use_first, int nthreads, int32_t* counts, int32_t* total) {
    typename kaori::SingleBarcodeSingleEnd<N>::Options options;
    options.strand = to_strand(strand);
    options.max_mismatches = mm;
    options.use_first = use_first;
    kaori::SingleBarcodeSingleEnd<N> handler(constant.c_str(), constant.size(), pool, options);
    kaori::process_single_end_data(&reader, handler, nthreads);
    const auto& c = handler.get_counts();
    std::copy(c.begin(), c.end(), counts);
    *total = handler.get_total();
}

template<size_t N>
void combo_(byteme::SomeFileReader& reader, const std::string& constant, int strand,
            const std::array<kaori::BarcodePool, 2>& pools, int mm, bool use_first, int nthreads,
            std::vector<std::array<int, 2> >& out, int32_t* total) {
    typename kaori::CombinatorialBarcodesSingleEnd<N, 2>::Options options;
    options.strand = to_strand(strand);
    options.max_mismatches = mm;
    options.use_first = use_first;
    kaori::CombinatorialBarcodesSingleEnd<N, 2> handler(constant.c_str(), constant.size(), pools, options);
    kaori::process_single_end_data(&reader, handler, nthreads);
    handler.sort();
    out = handler.get_combinations();
    *total = handler.get_total();
}

template<size_t N>
void dual_(byteme::SomeFileReader& r1, const std::string& c1, bool rev1, const kaori::BarcodePool& p1, int mm1,
           byteme::SomeFileReader& r2, const std::string& c2, bool rev2, const kaori::BarcodePool& p2, int mm2,
           bool randomized, bool use_first, int nthreads, int32_t* counts, int32_t* total) {
    typename kaori::DualBarcodesPairedEnd<N>::Options options;
    options.strand1 = rev1 ? kaori::SearchStrand::REVERSE : kaori::SearchStrand::FORWARD;
    options.max_mismatches1 = mm1;
    options.strand2 = rev2 ? kaori::SearchStrand::REVERSE : kaori::SearchStrand::FORWARD;
    options.max_mismatches2 = mm2;
    options.random = randomized;
    options.use_first = use_first;
    kaori::DualBarcodesPairedEnd<N> handler(c1.c_str(), c1.size(), p1, c2.c_str(), c2.size(), p2, options);
    kaori::process_paired_end_data(&r1, &r2, handler, nthreads);
    const auto& c = handler.get_counts();
    std::copy(c.begin(), c.end(), counts);
    *total = handler.get_total();
}

#define DISPATCH_N(len, CALL)                                                                        \
    if ((len) <= 32) { CALL(32); }                                                                   \
    else if ((len) <= 64) { CALL(64); }                                                              \
    else if ((len) <= 128) { CALL(128); }                                                            \
    else if ((len) <= 256) { CALL(256); }                                                            \
    else { throw std::runtime_error("lacking compile-time support for constant regions longer than 256 bp"); }

} // namespace

extern "C" {

int kref_count_single(const char* path, const char* tmpl, int strand, const char* const* pool, int n_pool,
                      int mm, int use_first, int nthreads, int32_t* counts, int32_t* total,
                      char* err, size_t errcap) {
    try {
        byteme::SomeFileReader reader(path);
        auto bp = make_pool(pool, n_pool);
        std::string constant(tmpl);
#define CALL(N) single_<N>(reader, constant, strand, bp, mm, use_first != 0, nthreads, counts, total)
        DISPATCH_N(constant.size(), CALL)
#undef CALL
    } catch (std::exception& e) {
        return set_err(err, errcap, e.what());
    }
    return 0;
}

/* On success *idx_out is a malloc'd 2xK column-major matrix of 0-based indices sorted by
 * (first, second), *freq_out the K frequencies (src/utils.h:14-45); free with kref_free. */
int kref_count_combo(const char* path, const char* tmpl, int strand,
                     const char* const* pool0, int n0, const char* const* pool1, int n1,
                     int mm, int use_first, int nthreads,
                     int32_t** idx_out, int32_t** freq_out, int64_t* k_out, int32_t* total,
                     char* err, size_t errcap) {
    try {
        byteme::SomeFileReader reader(path);
        std::array<kaori::BarcodePool, 2> pools{ make_pool(pool0, n0), make_pool(pool1, n1) };
        std::string constant(tmpl);
        std::vector<std::array<int, 2> > sorted;
#define CALL(N) combo_<N>(reader, constant, strand, pools, mm, use_first != 0, nthreads, sorted, total)
        DISPATCH_N(constant.size(), CALL)
#undef CALL
        std::vector<int32_t> idx, freq;
        for (size_t i = 0; i < sorted.size(); ++i) {
            if (i && sorted[i] == sorted[i - 1]) {
                ++freq.back();
            } else {
                idx.push_back(sorted[i][0]);
                idx.push_back(sorted[i][1]);
                freq.push_back(1);
            }
        }
        *k_out = static_cast<int64_t>(freq.size());
        *idx_out = static_cast<int32_t*>(std::malloc(sizeof(int32_t) * (idx.size() + 1)));
        *freq_out = static_cast<int32_t*>(std::malloc(sizeof(int32_t) * (freq.size() + 1)));
        std::copy(idx.begin(), idx.end(), *idx_out);
        std::copy(freq.begin(), freq.end(), *freq_out);
    } catch (std::exception& e) {
        return set_err(err, errcap, e.what());
    }
    return 0;
}

int kref_count_dual(const char* path1, const char* tmpl1, int reverse1, int mm1, const char* const* pool1,
                    const char* path2, const char* tmpl2, int reverse2, int mm2, const char* const* pool2,
                    int n_pool, int randomized, int use_first, int nthreads,
                    int32_t* counts, int32_t* total, char* err, size_t errcap) {
    try {
        byteme::SomeFileReader r1(path1);
        auto p1 = make_pool(pool1, n_pool);
        byteme::SomeFileReader r2(path2);
        auto p2 = make_pool(pool2, n_pool);
        std::string c1(tmpl1), c2(tmpl2);
        size_t len = std::max(c1.size(), c2.size());
#define CALL(N) dual_<N>(r1, c1, reverse1 != 0, p1, mm1, r2, c2, reverse2 != 0, p2, mm2, randomized != 0, use_first != 0, nthreads, counts, total)
        DISPATCH_N(len, CALL)
#undef CALL
    } catch (std::exception& e) {
        return set_err(err, errcap, e.what());
    }
    return 0;
}

/* include.invalid=TRUE branch, src/count_dual_barcodes.cpp:52-71: counts, sorted invalid combinations
 * (2 x K, 0-based) with frequencies, total, barcode1-only, barcode2-only. */
int kref_count_dual_diag(const char* path1, const char* tmpl1, int reverse1, int mm1, const char* const* pool1,
                         const char* path2, const char* tmpl2, int reverse2, int mm2, const char* const* pool2,
                         int n_pool, int randomized, int use_first, int nthreads,
                         int32_t* counts, int32_t** idx_out, int32_t** freq_out, int64_t* k_out,
                         int32_t* total, int32_t* b1_only, int32_t* b2_only, char* err, size_t errcap) {
    try {
        byteme::SomeFileReader r1(path1);
        auto p1 = make_pool(pool1, n_pool);
        byteme::SomeFileReader r2(path2);
        auto p2 = make_pool(pool2, n_pool);
        std::string c1(tmpl1), c2(tmpl2);
        size_t len = std::max(c1.size(), c2.size());
        std::vector<std::array<int, 2> > sorted;
        auto run = [&](auto tag) {
            constexpr size_t N = decltype(tag)::value;
            typename kaori::DualBarcodesPairedEnd<N>::Options options;
            options.strand1 = reverse1 ? kaori::SearchStrand::REVERSE : kaori::SearchStrand::FORWARD;
            options.max_mismatches1 = mm1;
            options.strand2 = reverse2 ? kaori::SearchStrand::REVERSE : kaori::SearchStrand::FORWARD;
            options.max_mismatches2 = mm2;
            options.random = randomized != 0;
            options.use_first = use_first != 0;
            kaori::DualBarcodesPairedEndWithDiagnostics<N> handler(c1.c_str(), c1.size(), p1, c2.c_str(), c2.size(), p2, options);
            kaori::process_paired_end_data(&r1, &r2, handler, nthreads);
            handler.sort();
            const auto& c = handler.get_counts();
            std::copy(c.begin(), c.end(), counts);
            sorted = handler.get_combinations();
            *total = handler.get_total();
            *b1_only = handler.get_barcode1_only();
            *b2_only = handler.get_barcode2_only();
        };
        if (len <= 32) run(std::integral_constant<size_t, 32>());
        else if (len <= 64) run(std::integral_constant<size_t, 64>());
        else if (len <= 128) run(std::integral_constant<size_t, 128>());
        else if (len <= 256) run(std::integral_constant<size_t, 256>());
        else throw std::runtime_error("lacking compile-time support for constant regions longer than 256 bp");
        std::vector<int32_t> idx, freq;
        for (size_t i = 0; i < sorted.size(); ++i) {
            if (i && sorted[i] == sorted[i - 1]) {
                ++freq.back();
            } else {
                idx.push_back(sorted[i][0]);
                idx.push_back(sorted[i][1]);
                freq.push_back(1);
            }
        }
        *k_out = static_cast<int64_t>(freq.size());
        *idx_out = static_cast<int32_t*>(std::malloc(sizeof(int32_t) * (idx.size() + 1)));
        *freq_out = static_cast<int32_t*>(std::malloc(sizeof(int32_t) * (freq.size() + 1)));
        std::copy(idx.begin(), idx.end(), *idx_out);
        std::copy(freq.begin(), freq.end(), *freq_out);
    } catch (std::exception& e) {
        return set_err(err, errcap, e.what());
    }
    return 0;
}

/* src/count_dual_barcodes_single_end.cpp:36-50 (include.invalid=TRUE): counts, sorted combinations (2 x K,
 * 0-based) with frequencies, total. */
int kref_count_dual_single_end_diag(const char* path, const char* tmpl, int strand,
                                    const char* const* const* pools, const int* n_pools, int n_regions,
                                    int mm, int use_first, int nthreads, int32_t* counts,
                                    int32_t** idx_out, int32_t** freq_out, int64_t* k_out, int32_t* total,
                                    char* err, size_t errcap) {
    try {
        byteme::SomeFileReader reader(path);
        std::vector<kaori::BarcodePool> ptr_pools;
        for (int r = 0; r < n_regions; ++r) ptr_pools.push_back(make_pool(pools[r], n_pools[r]));
        std::string constant(tmpl);
        std::vector<std::array<int, 2> > sorted;
        auto run = [&](auto tag) {
            constexpr size_t N = decltype(tag)::value;
            typename kaori::DualBarcodesSingleEnd<N>::Options options;
            options.strand = to_strand(strand);
            options.max_mismatches = mm;
            options.use_first = use_first != 0;
            kaori::DualBarcodesSingleEndWithDiagnostics<N, 2> handler(constant.c_str(), constant.size(), ptr_pools, options);
            kaori::process_single_end_data(&reader, handler, nthreads);
            handler.sort();
            const auto& c = handler.get_counts();
            std::copy(c.begin(), c.end(), counts);
            sorted = handler.get_combinations();
            *total = handler.get_total();
        };
        size_t len = constant.size();
        if (len <= 32) run(std::integral_constant<size_t, 32>());
        else if (len <= 64) run(std::integral_constant<size_t, 64>());
        else if (len <= 128) run(std::integral_constant<size_t, 128>());
        else if (len <= 256) run(std::integral_constant<size_t, 256>());
        else throw std::runtime_error("lacking compile-time support for constant regions longer than 256 bp");
        std::vector<int32_t> idx, freq;
        for (size_t i = 0; i < sorted.size(); ++i) {
            if (i && sorted[i] == sorted[i - 1]) {
                ++freq.back();
            } else {
                idx.push_back(sorted[i][0]);
                idx.push_back(sorted[i][1]);
                freq.push_back(1);
            }
        }
        *k_out = static_cast<int64_t>(freq.size());
        *idx_out = static_cast<int32_t*>(std::malloc(sizeof(int32_t) * (idx.size() + 1)));
        *freq_out = static_cast<int32_t*>(std::malloc(sizeof(int32_t) * (freq.size() + 1)));
        std::copy(idx.begin(), idx.end(), *idx_out);
        std::copy(freq.begin(), freq.end(), *freq_out);
    } catch (std::exception& e) {
        return set_err(err, errcap, e.what());
    }
    return 0;
}

/* src/count_random_barcodes.cpp:41-62: *seq_out = K strings of *len_out chars, NUL-terminated, sorted
 * byte-wise (the reference's unordered_map order is unspecified; its R caller sorts); free with kref_free. */
int kref_count_random(const char* path, const char* tmpl, int strand, int mm, int use_first, int nthreads,
                      char** seq_out, int32_t** freq_out, int64_t* k_out, int32_t* len_out, int32_t* total,
                      char* err, size_t errcap) {
    try {
        byteme::SomeFileReader reader(path);
        std::string constant(tmpl);
        std::vector<std::pair<std::string, int> > rows;
        auto run = [&](auto tag) {
            constexpr size_t N = decltype(tag)::value;
            typename kaori::RandomBarcodeSingleEnd<N>::Options options;
            options.strand = to_strand(strand);
            options.max_mismatches = mm;
            options.use_first = use_first != 0;
            kaori::RandomBarcodeSingleEnd<N> handler(constant.c_str(), constant.size(), options);
            kaori::process_single_end_data(&reader, handler, nthreads);
            rows.assign(handler.get_counts().begin(), handler.get_counts().end());
            *total = handler.get_total();
        };
        size_t len = constant.size();
        if (len <= 32) run(std::integral_constant<size_t, 32>());
        else if (len <= 64) run(std::integral_constant<size_t, 64>());
        else if (len <= 128) run(std::integral_constant<size_t, 128>());
        else if (len <= 256) run(std::integral_constant<size_t, 256>());
        else throw std::runtime_error("lacking compile-time support for constant regions longer than 256 bp");
        std::sort(rows.begin(), rows.end());
        size_t vlen = rows.empty() ? 0 : rows[0].first.size();
        *len_out = static_cast<int32_t>(vlen);
        *k_out = static_cast<int64_t>(rows.size());
        *seq_out = static_cast<char*>(std::malloc(rows.size() * (vlen + 1) + 1));
        *freq_out = static_cast<int32_t*>(std::malloc(sizeof(int32_t) * (rows.size() + 1)));
        for (size_t i = 0; i < rows.size(); ++i) {
            std::memcpy(*seq_out + i * (vlen + 1), rows[i].first.data(), vlen);
            (*seq_out)[i * (vlen + 1) + vlen] = 0;
            (*freq_out)[i] = rows[i].second;
        }
    } catch (std::exception& e) {
        return set_err(err, errcap, e.what());
    }
    return 0;
}

/* src/count_dual_barcodes_single_end.cpp:53-87, non-diagnostic branch. */
int kref_count_dual_single_end(const char* path, const char* tmpl, int strand,
                               const char* const* const* pools, const int* n_pools, int n_regions,
                               int mm, int use_first, int nthreads, int32_t* counts, int32_t* total,
                               char* err, size_t errcap) {
    try {
        byteme::SomeFileReader reader(path);
        std::vector<kaori::BarcodePool> ptr_pools;
        for (int r = 0; r < n_regions; ++r) ptr_pools.push_back(make_pool(pools[r], n_pools[r]));
        std::string constant(tmpl);
        auto run = [&](auto tag) {
            constexpr size_t N = decltype(tag)::value;
            typename kaori::DualBarcodesSingleEnd<N>::Options options;
            options.strand = to_strand(strand);
            options.max_mismatches = mm;
            options.use_first = use_first != 0;
            kaori::DualBarcodesSingleEnd<N> handler(constant.c_str(), constant.size(), ptr_pools, options);
            kaori::process_single_end_data(&reader, handler, nthreads);
            const auto& c = handler.get_counts();
            std::copy(c.begin(), c.end(), counts);
            *total = handler.get_total();
        };
        size_t len = constant.size();
        if (len <= 32) run(std::integral_constant<size_t, 32>());
        else if (len <= 64) run(std::integral_constant<size_t, 64>());
        else if (len <= 128) run(std::integral_constant<size_t, 128>());
        else if (len <= 256) run(std::integral_constant<size_t, 256>());
        else throw std::runtime_error("lacking compile-time support for constant regions longer than 256 bp");
    } catch (std::exception& e) {
        return set_err(err, errcap, e.what());
    }
    return 0;
}

/* src/count_combo_barcodes_paired.cpp:57-95: sorted combinations (2 x K, 0-based) with frequencies,
 * total, barcode1-only, barcode2-only. */
int kref_count_combo_paired(const char* path1, const char* tmpl1, int reverse1, int mm1, const char* const* pool1, int n_pool1,
                            const char* path2, const char* tmpl2, int reverse2, int mm2, const char* const* pool2, int n_pool2,
                            int randomized, int use_first, int nthreads,
                            int32_t** idx_out, int32_t** freq_out, int64_t* k_out,
                            int32_t* total, int32_t* b1_only, int32_t* b2_only, char* err, size_t errcap) {
    try {
        byteme::SomeFileReader r1(path1);
        auto p1 = make_pool(pool1, n_pool1);
        byteme::SomeFileReader r2(path2);
        auto p2 = make_pool(pool2, n_pool2);
        std::string c1(tmpl1), c2(tmpl2);
        size_t len = std::max(c1.size(), c2.size());
        std::vector<std::array<int, 2> > sorted;
        auto run = [&](auto tag) {
            constexpr size_t N = decltype(tag)::value;
            typename kaori::CombinatorialBarcodesPairedEnd<N>::Options options;
            options.strand1 = reverse1 ? kaori::SearchStrand::REVERSE : kaori::SearchStrand::FORWARD;
            options.max_mismatches1 = mm1;
            options.strand2 = reverse2 ? kaori::SearchStrand::REVERSE : kaori::SearchStrand::FORWARD;
            options.max_mismatches2 = mm2;
            options.random = randomized != 0;
            options.use_first = use_first != 0;
            kaori::CombinatorialBarcodesPairedEnd<N> handler(c1.c_str(), c1.size(), p1, c2.c_str(), c2.size(), p2, options);
            kaori::process_paired_end_data(&r1, &r2, handler, nthreads);
            handler.sort();
            sorted = handler.get_combinations();
            *total = handler.get_total();
            *b1_only = handler.get_barcode1_only();
            *b2_only = handler.get_barcode2_only();
        };
        if (len <= 32) run(std::integral_constant<size_t, 32>());
        else if (len <= 64) run(std::integral_constant<size_t, 64>());
        else if (len <= 128) run(std::integral_constant<size_t, 128>());
        else if (len <= 256) run(std::integral_constant<size_t, 256>());
        else throw std::runtime_error("lacking compile-time support for constant regions longer than 256 bp");
        std::vector<int32_t> idx, freq;
        for (size_t i = 0; i < sorted.size(); ++i) {
            if (i && sorted[i] == sorted[i - 1]) {
                ++freq.back();
            } else {
                idx.push_back(sorted[i][0]);
                idx.push_back(sorted[i][1]);
                freq.push_back(1);
            }
        }
        *k_out = static_cast<int64_t>(freq.size());
        *idx_out = static_cast<int32_t*>(std::malloc(sizeof(int32_t) * (idx.size() + 1)));
        *freq_out = static_cast<int32_t*>(std::malloc(sizeof(int32_t) * (freq.size() + 1)));
        std::copy(idx.begin(), idx.end(), *idx_out);
        std::copy(freq.begin(), freq.end(), *freq_out);
    } catch (std::exception& e) {
        return set_err(err, errcap, e.what());
    }
    return 0;
}

/* src/match_barcodes.cpp:6-37; index 0-based, -1 where R would report NA. */
int kref_match_barcodes(const char* const* sequences, int nseq, const char* const* choices, int nchoices,
                        int substitutions, int reverse, int32_t* index_out, int32_t* mm_out,
                        char* err, size_t errcap) {
    try {
        kaori::SimpleBarcodeSearch::Options opt;
        opt.max_mismatches = substitutions;
        opt.reverse = reverse != 0;
        auto pool = make_pool(choices, nchoices);
        kaori::SimpleBarcodeSearch searcher(pool, opt);
        auto state = searcher.initialize();
        auto x = make_pool(sequences, nseq);
        for (int i = 0; i < nseq; ++i) {
            searcher.search(x.pool[i], state);
            if (state.index >= 0) {
                index_out[i] = state.index;
                mm_out[i] = state.mismatches;
            } else {
                index_out[i] = -1;
                mm_out[i] = -1;
            }
        }
    } catch (std::exception& e) {
        return set_err(err, errcap, e.what());
    }
    return 0;
}

/* Sequences of a FASTQ file exactly as kaori::FastqReader yields them. */
int kref_parse_fastq(const char* path, char** seqs_out, uint64_t** offsets_out, int64_t* n_out,
                     char* err, size_t errcap) {
    try {
        byteme::SomeFileReader reader(path);
        kaori::FastqReader fq(&reader);
        std::vector<char> seqs;
        std::vector<uint64_t> offs(1, 0);
        while (fq()) {
            const auto& s = fq.get_sequence();
            seqs.insert(seqs.end(), s.begin(), s.end());
            offs.push_back(seqs.size());
        }
        *n_out = static_cast<int64_t>(offs.size()) - 1;
        *seqs_out = static_cast<char*>(std::malloc(seqs.size() + 1));
        *offsets_out = static_cast<uint64_t*>(std::malloc(sizeof(uint64_t) * offs.size()));
        std::copy(seqs.begin(), seqs.end(), *seqs_out);
        std::copy(offs.begin(), offs.end(), *offsets_out);
    } catch (std::exception& e) {
        return set_err(err, errcap, e.what());
    }
    return 0;
}

void kref_free(void* p) { std::free(p); }

} // extern "C"
