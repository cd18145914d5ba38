// scg_plan.cpp -- plans: host compilation of the templates and libraries (every argument check of the reference's
// constructors lives here), the combination streams of the sparse mode, and the launch of one batch.
//
// Host-side counterpart of the handler construction in the reference's Rcpp glue (src/count_single_barcodes.cpp,
// src/count_combo_barcodes_single.cpp, src/count_dual_barcodes.cpp, src/count_dual_barcodes_single_end.cpp).
#include "scg_internal.hpp"

namespace scgapi {
// Key width classes: 0 = up to 32 bases (2 x 32-bit planes), 1 = up to 64, 2 = up to 256 ("big").  Pools that meet in one
// kernel (both regions of a combination, both mates of a pair) are built in the widest class among them.
int key_class(int len) { return len > SCG_MAX_WIDE_BARCODE ? 2 : (len > SCG_MAX_BARCODE ? 1 : 0); }

scg::HostIndex build_index_class(int cls, const char* const* pool, int32_t n, int32_t len, int max_mm) {
    if (cls == 2) return scg::build_index_big(pool, n, len, max_mm);
    return cls == 1 ? scg::build_index_wide(pool, n, len, max_mm) : scg::build_index(pool, n, len, max_mm);
}
scg::HostIndex build_uid_index_class(int cls, const char* const* pool, int32_t n, int32_t len, int max_mm,
                                     std::vector<std::vector<int32_t> >& expansions, size_t& n_uid) {
    if (cls == 2) return scg::build_uid_index_big(pool, n, len, max_mm, expansions, n_uid);
    return scg::build_uid_index_wide(pool, n, len, max_mm, expansions, n_uid);
}

// Big keys have the byte-wise general kernels only.
bool general_only(const scg_plan* P) {
    return scg::force_general() || P->tab[0].view.wide == 2 || (P->kind == scg_plan::DUAL_SE_DIAG && P->tab_combined.view.wide == 2);
}

// 256 MB of int32 cells: beyond that, combinations are sorted and run-length encoded.  $SCG_DENSE_CELLS moves the limit (the
// tests run every combination case both ways).
int64_t dense_cells() {
    const char* e = getenv("SCG_DENSE_CELLS");
    if (e && *e) return std::min<int64_t>(std::max<int64_t>(atoll(e), 0), int64_t(1) << 30);
    return int64_t(1) << 26;
}

ScgReads make_reads(const char* d_seqs, const uint32_t* d_offsets, int32_t fixed_len, int32_t max_len) {
    ScgReads r;
    r.seqs = reinterpret_cast<const uint8_t*>(d_seqs);
    r.offsets = d_offsets;
    r.fixed_len = d_offsets ? 0 : fixed_len;
    r.max_len = d_offsets ? max_len : fixed_len;
#ifdef SCG_ABLATE
    // measurement builds only (make EXTRA=-DSCG_ABLATE OUT=...): the product library has no such switch
    const char* ab = std::getenv("SCG_ABLATE");
    r.ablate = ab ? std::atoi(ab) : 0;
#else
    r.ablate = 0;
#endif
    return r;
}

void check_reads_args(const char* d_seqs, const uint32_t* d_offsets, int32_t fixed_len, int64_t n) {
    if (n < 0) throw Error(SCG_ERR_INVALID, "negative read count");
    if (n > 0 && !d_seqs && !(d_offsets == nullptr && fixed_len == 0)) throw Error(SCG_ERR_INVALID, "null read buffer");
    if (!d_offsets && fixed_len < 0) throw Error(SCG_ERR_INVALID, "negative fixed read length");
}

// ---- host compilation of the three plan kinds (all reference argument checks live here) ----

std::unique_ptr<scg_plan> compile_single(const char* constant, int strand, const char* const* pool, int32_t n_pool,
                                         int mismatches, int use_first) {
    if (!constant || (n_pool > 0 && !pool) || n_pool < 0) throw Error(SCG_ERR_INVALID, "null argument");
    std::unique_ptr<scg_plan> P(new scg_plan);
    P->kind = scg_plan::SINGLE;
    int plen = scg::pool_length(pool, n_pool);                 // src/utils.cpp:15-17
    P->ht1 = scg::parse_template(constant, strand);            // src/count_single_barcodes.cpp:37-47, ScanTemplate.hpp:53-95
    if (P->ht1.t.nreg != 1) {
        throw Error(SCG_ERR_INVALID, "expected one variable region in the constant template");   // SimpleSingleMatch.hpp:75-77
    }
    int vlen = P->ht1.t.flen[0];
    if (vlen != plen) {                                        // SimpleSingleMatch.hpp:79-83
        throw Error(SCG_ERR_INVALID, "length of barcode_pool sequences (" + std::to_string(plen) +
                    ") should be the same as the barcode_pool region (" + std::to_string(vlen) + ")");
    }
    if (mismatches < 0) throw Error(SCG_ERR_INVALID, "negative number of mismatches");
    // BarcodeSearch.hpp:23-60; barcodes of 33..64 bases take the wide (2 x 64-bit plane) index and kernels
    P->htab[0] = plen > SCG_MAX_BARCODE ? scg::build_index_wide(pool, n_pool, plen, mismatches) : scg::build_index(pool, n_pool, plen, mismatches);
    P->scan1 = scg::build_scan(P->ht1.t, mismatches);
    P->n_pool[0] = n_pool;
    P->n_counters = n_pool;
    P->max_mm1 = mismatches;
    P->use_first = use_first != 0;
    return P;
}

// countDualBarcodesSingleEnd (kaori::DualBarcodesSingleEnd, handlers/DualBarcodesSingleEnd.hpp:66-123): one read
// holds every variable region; pools[r][c] over r spells valid combination c, and the concatenation of a window's
// regions is matched against the concatenated library with one shared mismatch budget.  Same kernels as the single
// barcode with a wide key assembled from several regions.
std::unique_ptr<scg_plan> compile_dual_single_end(const char* constant, int strand, const char* const* const* pools, const int32_t* n_pools,
                                                  int32_t n_regions, int mismatches, int use_first) {
    if (!constant || n_regions < 0 || (n_regions > 0 && (!pools || !n_pools))) throw Error(SCG_ERR_INVALID, "null argument");
    std::unique_ptr<scg_plan> P(new scg_plan);
    P->kind = scg_plan::SINGLE;
    std::vector<int> plen(n_regions);
    for (int r = 0; r < n_regions; ++r) {
        if (n_pools[r] < 0 || (n_pools[r] > 0 && !pools[r])) throw Error(SCG_ERR_INVALID, "null argument");
        plen[r] = scg::pool_length(pools[r], n_pools[r]);         // src/utils.cpp:15-17 (format_pointers per pool)
    }
    P->ht1 = scg::parse_template(constant, strand);
    const ScgTemplate& t = P->ht1.t;
    if (t.nreg != n_regions) throw Error(SCG_ERR_INVALID, "length of 'barcode_pools' should equal the number of variable regions");   // :76-78
    if (n_regions < 1 || n_regions > SCG_MAX_REGIONS) {
        throw Error(SCG_ERR_UNSUPPORTED, "this engine counts dual barcodes in single-end reads with 1 to " + std::to_string(SCG_MAX_REGIONS) +
                    " variable regions (got " + std::to_string(n_regions) + ")");
    }
    int total = 0;
    for (int r = 0; r < n_regions; ++r) {                         // :80-87
        if (plen[r] != t.flen[r]) {
            throw Error(SCG_ERR_INVALID, "length of variable region " + std::to_string(r + 1) + " (" + std::to_string(t.flen[r]) +
                        ") should be the same as its sequences (" + std::to_string(plen[r]) + ")");
        }
        total += plen[r];
    }
    const int32_t n_choices = n_pools[0];
    for (int r = 1; r < n_regions; ++r) {                         // :89-97
        if (n_pools[r] != n_choices) throw Error(SCG_ERR_INVALID, "all entries of 'barcode_pools' should have the same length");
    }
    if (mismatches < 0) throw Error(SCG_ERR_INVALID, "negative number of mismatches");
    std::vector<std::string> combined(n_choices);                // :100-109
    std::vector<const char*> ptrs(n_choices);
    for (int32_t c = 0; c < n_choices; ++c) {
        for (int r = 0; r < n_regions; ++r) combined[c].append(pools[r][c], plen[r]);
        ptrs[c] = combined[c].c_str();
    }
    P->htab[0] = scg::build_index_wide(ptrs.data(), n_choices, total, mismatches);     // duplicates => error (:111-113)
    P->scan1 = scg::build_scan(t, mismatches);
    P->n_pool[0] = n_choices;
    P->n_counters = n_choices;
    P->max_mm1 = mismatches;
    P->use_first = use_first != 0;
    return P;
}

std::unique_ptr<scg_plan> compile_combo(const char* constant, int strand,
                                        const char* const* pool0, int32_t n0, const char* const* pool1, int32_t n1,
                                        int mismatches, int use_first) {
    if (!constant || (n0 > 0 && !pool0) || (n1 > 0 && !pool1) || n0 < 0 || n1 < 0) throw Error(SCG_ERR_INVALID, "null argument");
    std::unique_ptr<scg_plan> P(new scg_plan);
    P->kind = scg_plan::COMBO;
    int len0 = scg::pool_length(pool0, n0);
    int len1 = scg::pool_length(pool1, n1);
    P->ht1 = scg::parse_template(constant, strand);
    if (P->ht1.t.nreg != 2) {                                  // CombinatorialBarcodesSingleEnd.hpp:79-81
        throw Error(SCG_ERR_INVALID, "expected 2 variable regions in the constant template");
    }
    int lens[2] = {len0, len1};
    for (int r = 0; r < 2; ++r) {                              // :86-93
        if (P->ht1.t.flen[r] != lens[r]) {
            throw Error(SCG_ERR_INVALID, "length of variable region " + std::to_string(r + 1) + " (" + std::to_string(P->ht1.t.flen[r]) +
                        ") should be the same as its sequences (" + std::to_string(lens[r]) + ")");
        }
    }
    if (mismatches < 0) throw Error(SCG_ERR_INVALID, "negative number of mismatches");
    // pools of 33..64 bases take the wide (2 x 64-bit plane) index and kernels, longer ones the big one; both pools then, one
    // key width per kernel
    const int cls = std::max(key_class(len0), key_class(len1));
    P->htab[0] = build_index_class(cls, pool0, n0, len0, mismatches);
    P->htab[1] = build_index_class(cls, pool1, n1, len1, mismatches);
    P->scan1 = scg::build_scan(P->ht1.t, mismatches);
    P->n_pool[0] = n0; P->n_pool[1] = n1;
    int64_t cells = static_cast<int64_t>(n0) * static_cast<int64_t>(n1);
    P->sparse = cells > dense_cells();                           // beyond the dense limit: sort + run-length encode, like the reference
    P->n_counters = P->sparse ? 0 : cells;
    P->max_mm1 = mismatches;
    P->use_first = use_first != 0;
    return P;
}

std::unique_ptr<scg_plan> compile_dual(const char* constant1, int reverse1, int mismatches1, const char* const* pool1,
                                       const char* constant2, int reverse2, int mismatches2, const char* const* pool2,
                                       int32_t n_pool, int randomized, int use_first, int diagnostics) {
    if (!constant1 || !constant2 || (n_pool > 0 && (!pool1 || !pool2)) || n_pool < 0) throw Error(SCG_ERR_INVALID, "null argument");
    std::unique_ptr<scg_plan> P(new scg_plan);
    P->kind = scg_plan::DUAL;
    int len1 = scg::pool_length(pool1, n_pool);                // src/count_dual_barcodes.cpp:93-97
    int len2 = scg::pool_length(pool2, n_pool);
    P->ht1 = scg::parse_template(constant1, reverse1 ? 1 : 0); // DualBarcodesPairedEnd.hpp:99-100
    P->ht2 = scg::parse_template(constant2, reverse2 ? 1 : 0);
    if (P->ht1.t.nreg != 1) throw Error(SCG_ERR_INVALID, "expected one variable region in the first constant template");    // :115-117
    if (P->ht1.t.flen[0] != len1) {                            // :119-122
        throw Error(SCG_ERR_INVALID, "length of variable sequences (" + std::to_string(len1) + ") should be the same as the variable region (" +
                    std::to_string(P->ht1.t.flen[0]) + ")");
    }
    if (P->ht2.t.nreg != 1) throw Error(SCG_ERR_INVALID, "expected one variable region in the second constant template");   // :128-130
    if (P->ht2.t.flen[0] != len2) {
        throw Error(SCG_ERR_INVALID, "length of variable sequences (" + std::to_string(len2) + ") should be the same as the variable region (" +
                    std::to_string(P->ht2.t.flen[0]) + ")");
    }
    if (mismatches1 < 0 || mismatches2 < 0) throw Error(SCG_ERR_INVALID, "negative number of mismatches");
    std::vector<std::vector<int32_t> > exp1, exp2;
    std::vector<uint64_t> uk1, uk2;
    size_t n_uid1 = 0, n_uid2 = 0;
    if (const int cls = std::max(key_class(len1), key_class(len2))) {     // barcodes of more than 32 bases on either mate: wide / big indexes and kernels for both
        P->htab[0] = build_uid_index_class(cls, pool1, n_pool, len1, mismatches1, exp1, n_uid1);
        P->htab[1] = build_uid_index_class(cls, pool2, n_pool, len2, mismatches2, exp2, n_uid2);
    } else {
        P->htab[0] = scg::build_uid_index(pool1, n_pool, len1, mismatches1, exp1, uk1);
        P->htab[1] = scg::build_uid_index(pool2, n_pool, len2, mismatches2, exp2, uk2);
        n_uid1 = uk1.size(); n_uid2 = uk2.size();
    }
    P->scan1 = scg::build_scan(P->ht1.t, mismatches1);
    P->scan2 = scg::build_scan(P->ht2.t, mismatches2);
    P->hpairs = scg::build_pair_table(exp1, uk1, exp2, uk2);   // :138-178 (duplicate pairs => error)
    P->n_pool[0] = P->n_pool[1] = n_pool;
    P->n_counters = n_pool;
    if (diagnostics) {
        // uid -> index of the first barcode that contains the sequence
        auto firsts = [&](const std::vector<std::vector<int32_t> >& exp, size_t n_uid) {
            std::vector<int32_t> f(n_uid, -1);
            for (size_t i = 0; i < exp.size(); ++i) {
                for (int32_t u : exp[i]) if (f[u] < 0) f[u] = static_cast<int32_t>(i);
            }
            return f;
        };
        P->first1 = firsts(exp1, n_uid1);
        P->first2 = firsts(exp2, n_uid2);
        int64_t cells = static_cast<int64_t>(n_uid1) * static_cast<int64_t>(n_uid2);
        P->sparse = cells > dense_cells();
        P->diagnostics = 1;
        P->n_counters = static_cast<int64_t>(n_pool) + 2 + (P->sparse ? 0 : cells);
    }
    P->max_mm1 = mismatches1; P->max_mm2 = mismatches2;
    P->rev1 = reverse1 != 0; P->rev2 = reverse2 != 0;
    P->randomized = randomized != 0;
    P->use_first = use_first != 0;
    return P;
}

// countDualBarcodesSingleEnd(include.invalid=TRUE): DualBarcodesSingleEndWithDiagnostics<N, 2>
// (handlers/DualBarcodesSingleEndWithDiagnostics.hpp:35-60) = the valid-combination handler plus
// CombinatorialBarcodesSingleEnd<N, 2> over the same pools with DuplicateAction::FIRST.
std::unique_ptr<scg_plan> compile_dual_single_end_diag(const char* constant, int strand, const char* const* const* pools, const int32_t* n_pools,
                                                       int32_t n_regions, int mismatches, int use_first) {
    auto P = compile_dual_single_end(constant, strand, pools, n_pools, n_regions, mismatches, use_first);   // its constructor runs first
    const ScgTemplate& t = P->ht1.t;
    if (t.nreg != 2) throw Error(SCG_ERR_INVALID, "expected 2 variable regions in the constant template");   // CombinatorialBarcodesSingleEnd.hpp:84-86
    P->kind = scg_plan::DUAL_SE_DIAG;
    P->htab_combined = std::move(P->htab[0]);
    std::vector<std::vector<int32_t> > exp0, exp1;
    std::vector<uint64_t> uk0, uk1;
    size_t n_uid0 = 0, n_uid1 = 0;
    if (const int cls = std::max(key_class(t.flen[0]), key_class(t.flen[1]))) {
        P->htab[0] = build_uid_index_class(cls, pools[0], n_pools[0], t.flen[0], mismatches, exp0, n_uid0);
        P->htab[1] = build_uid_index_class(cls, pools[1], n_pools[1], t.flen[1], mismatches, exp1, n_uid1);
    } else {
        P->htab[0] = scg::build_uid_index(pools[0], n_pools[0], t.flen[0], mismatches, exp0, uk0);
        P->htab[1] = scg::build_uid_index(pools[1], n_pools[1], t.flen[1], mismatches, exp1, uk1);
        n_uid0 = uk0.size(); n_uid1 = uk1.size();
    }
    auto firsts = [&](const std::vector<std::vector<int32_t> >& exp, size_t n_uid) {
        std::vector<int32_t> f(n_uid, -1);
        for (size_t i = 0; i < exp.size(); ++i) {
            for (int32_t u : exp[i]) if (f[u] < 0) f[u] = static_cast<int32_t>(i);
        }
        return f;
    };
    P->first1 = firsts(exp0, n_uid0);
    P->first2 = firsts(exp1, n_uid1);
    int64_t cells = static_cast<int64_t>(n_uid0) * static_cast<int64_t>(n_uid1);
    P->sparse = cells > dense_cells();
    P->n_pool[1] = P->n_pool[0];
    P->n_counters = static_cast<int64_t>(P->n_pool[0]) + 2 + (P->sparse ? 0 : cells);
    return P;
}

// countPairedComboBarcodes: two independent SimpleSingleMatch matchers (CombinatorialBarcodesPairedEnd.hpp:85-118).
std::unique_ptr<scg_plan> compile_paired_combo(const char* constant1, int reverse1, int mismatches1, const char* const* pool1, int32_t n1,
                                               const char* constant2, int reverse2, int mismatches2, const char* const* pool2, int32_t n2,
                                               int randomized, int use_first) {
    if (!constant1 || !constant2 || (n1 > 0 && !pool1) || (n2 > 0 && !pool2) || n1 < 0 || n2 < 0) throw Error(SCG_ERR_INVALID, "null argument");
    std::unique_ptr<scg_plan> P(new scg_plan);
    P->kind = scg_plan::DUAL;
    int len1 = scg::pool_length(pool1, n1);                    // src/utils.cpp:15-17
    int len2 = scg::pool_length(pool2, n2);
    P->ht1 = scg::parse_template(constant1, reverse1 ? 1 : 0);
    P->ht2 = scg::parse_template(constant2, reverse2 ? 1 : 0);
    auto check = [](const scg::HostTemplate& ht, int plen) {   // SimpleSingleMatch.hpp:75-83
        if (ht.t.nreg != 1) throw Error(SCG_ERR_INVALID, "expected one variable region in the constant template");
        if (ht.t.flen[0] != plen) {
            throw Error(SCG_ERR_INVALID, "length of barcode_pool sequences (" + std::to_string(plen) +
                        ") should be the same as the barcode_pool region (" + std::to_string(ht.t.flen[0]) + ")");
        }
    };
    check(P->ht1, len1);
    check(P->ht2, len2);
    if (mismatches1 < 0 || mismatches2 < 0) throw Error(SCG_ERR_INVALID, "negative number of mismatches");
    const int cls = std::max(key_class(len1), key_class(len2));
    P->htab[0] = build_index_class(cls, pool1, n1, len1, mismatches1);     // values = pool indices; duplicates => error
    P->htab[1] = build_index_class(cls, pool2, n2, len2, mismatches2);
    P->scan1 = scg::build_scan(P->ht1.t, mismatches1);
    P->scan2 = scg::build_scan(P->ht2.t, mismatches2);
    int64_t cells = static_cast<int64_t>(n1) * static_cast<int64_t>(n2);
    P->sparse = cells > dense_cells();
    P->first1.resize(n1);
    P->first2.resize(n2);
    for (int32_t i = 0; i < n1; ++i) P->first1[i] = i;
    for (int32_t i = 0; i < n2; ++i) P->first2[i] = i;
    P->n_pool[0] = P->n_pool[1] = 0;                           // no list of valid pairs
    P->diagnostics = 2;
    P->n_counters = 2 + (P->sparse ? 0 : cells);               // [barcode1-only][barcode2-only][n1 x n2]
    P->max_mm1 = mismatches1; P->max_mm2 = mismatches2;
    P->rev1 = reverse1 != 0; P->rev2 = reverse2 != 0;
    P->randomized = randomized != 0;
    P->use_first = use_first != 0;
    return P;
}

ScgCounters plan_counters(const scg_plan* P) {
    ScgCounters c;
    if (P->replica_shift > 0) {
        c.base = P->replicas.as<int32_t>();
        c.replica_shift = static_cast<uint32_t>(P->replica_shift);
        c.replica_mask = (1u << P->replica_shift) - 1u;
    } else {
        c.base = P->counters; c.replica_mask = 0; c.replica_shift = 0;
    }
    c.unit_index = nullptr;
    c.unit_pair = nullptr;
    c.hot = P->hot.p ? P->hot.as<int32_t>() : nullptr;
    return c;
}

// ---- sparse mode: combination streams ----

// The runs of the batch last counted on `stream` -> the plan's map.
void retire_pairs(scg_plan* P, hipStream_t stream, scg_plan::PairStream& ps) {
    if (!ps.pending) return;
    uint32_t runs = 0;
    HIP_CHECK(hipEventSynchronize(ps.done));
    HIP_CHECK(hipMemcpy(&runs, ps.runs.p, sizeof(runs), hipMemcpyDeviceToHost));
    std::vector<uint64_t> keys(runs);
    std::vector<uint32_t> counts(runs);
    if (runs) {
        HIP_CHECK(hipMemcpy(keys.data(), ps.unique.p, sizeof(uint64_t) * runs, hipMemcpyDeviceToHost));
        HIP_CHECK(hipMemcpy(counts.data(), ps.counts.p, sizeof(uint32_t) * runs, hipMemcpyDeviceToHost));
    }
    for (uint32_t i = 0; i < runs; ++i) {
        if (keys[i] != ~uint64_t(0)) P->sparse_counts[keys[i]] += counts[i];
    }
    ps.pending = 0;
}

// A stream of n keys for the next batch on `stream`, every slot "none" (kernels that skip a read leave it so).
uint64_t* begin_pairs(scg_plan* P, hipStream_t stream, int64_t n) {
    scg_plan::PairStream& ps = P->pair_stream[stream];
    retire_pairs(P, stream, ps);
    const size_t m = static_cast<size_t>(std::max<int64_t>(n, 1));
    ps.keys.ensure(m * sizeof(uint64_t));
    ps.sorted.ensure(m * sizeof(uint64_t));
    ps.unique.ensure(m * sizeof(uint64_t));
    ps.counts.ensure(m * sizeof(uint32_t));
    if (!ps.runs.p) ps.runs.alloc(sizeof(uint32_t));
    ps.scratch.ensure(scg::sort_rle_scratch_bytes(m));
    HIP_CHECK(hipMemsetAsync(ps.keys.p, 0xFF, m * sizeof(uint64_t), stream));
    return ps.keys.as<uint64_t>();
}

// Behind the counting kernels of the batch: sort + run-length encode, still asynchronous.
void finish_pairs(scg_plan* P, hipStream_t stream, int64_t n) {
    scg_plan::PairStream& ps = P->pair_stream[stream];
    HIP_CHECK(scg::launch_sort_rle(ps.keys.as<uint64_t>(), ps.sorted.as<uint64_t>(), static_cast<size_t>(n), ps.unique.as<uint64_t>(),
                                   ps.counts.as<uint32_t>(), ps.runs.as<uint32_t>(), ps.scratch.p, ps.scratch.bytes, stream));
    if (!ps.done) HIP_CHECK(hipEventCreateWithFlags(&ps.done, hipEventDisableTiming));
    HIP_CHECK(hipEventRecord(ps.done, stream));
    ps.pending = n;
}

void retire_all_pairs(scg_plan* P) {
    DeviceGuard g(P->device);
    for (auto& kv : P->pair_stream) retire_pairs(P, kv.first, kv.second);
}

// Tally mode pays off when the library is large enough that block-level aggregation finds no repeats
// (small libraries are served by the replicas) and small enough for a few LDS passes, on batches
// large enough to amortise the second kernel.  SCG_TALLY=0/1 overrides (measurement aid).
bool use_tally(const scg_plan* P, int64_t n) {
    if (P->diagnostics) return false;      // the diagnostics kernels count several things per pair
    if (const char* e = std::getenv("SCG_TALLY")) { if (*e) return *e != '0'; }
    return P->n_counters >= 4096 && P->n_counters <= 4 * 80 * 1024 && n >= (int64_t(1) << 20);
}

void fold_replicas(scg_plan* P, hipStream_t stream) {
    if (P->replica_shift > 0) {
        HIP_CHECK(scg::launch_fold(P->replicas.as<int32_t>(), P->replica_shift, P->n_counters, P->counters, stream));
    }
}

// countDualBarcodesSingleEnd(include.invalid=TRUE) in two passes over the batch: the valid-combination search
// writes its per-read result as an index stream (tallied into counters[0 .. n_pool)), then the combinatorial
// search runs on the reads that found nothing (ScgComboParams::only_if_negative) with DuplicateAction::FIRST
// and counts (uid1, uid2) cells behind the two unused diagnostics slots: [n_pool][2][n_uid1 x n_uid2].
void launch_batch_se_diag(scg_plan* P, const ScgReads& R, int64_t n, hipStream_t stream) {
    scg_plan::Timer timer(P, stream);
    DevBuf& buf = P->unit_index[stream];
    buf.ensure(static_cast<size_t>(n) * sizeof(int32_t));
    ScgSingleParams sp;
    sp.scan = P->scan1;
    sp.tmpl = P->d_tmpl1.as<ScgTemplate>();
    sp.index = P->tab_combined.view;
    sp.max_mm = P->max_mm1; sp.use_first = P->use_first;
    sp.fwd = P->ht1.fwd; sp.rev = P->ht1.rev;
    ScgCounters c1;
    c1.base = P->counters; c1.replica_mask = 0; c1.replica_shift = 0; c1.hot = nullptr; c1.unit_pair = nullptr;
    c1.unit_index = buf.as<int32_t>();
    HIP_CHECK(scg::launch_single(sp, P->ht1.t.len, R, n, c1, P->error_flag.as<int32_t>(), stream));
    HIP_CHECK(scg::launch_tally(c1.unit_index, n, P->counters, P->n_pool[0], stream));
    ScgComboParams cp;
    cp.scan = P->scan1;
    cp.tmpl = P->d_tmpl1.as<ScgTemplate>();
    cp.index[0] = P->tab[0].view; cp.index[1] = P->tab[1].view;
    cp.n_pool[0] = static_cast<int32_t>(P->first1.size()); cp.n_pool[1] = static_cast<int32_t>(P->first2.size());
    cp.max_mm = P->max_mm1; cp.use_first = P->use_first;
    cp.fwd = P->ht1.fwd; cp.rev = P->ht1.rev;
    cp.only_if_negative = c1.unit_index; cp.keep_first = 1; cp.pad = 0;
    ScgCounters c2;
    c2.base = P->counters + P->n_pool[0] + 2; c2.replica_mask = 0; c2.replica_shift = 0; c2.unit_index = nullptr; c2.hot = nullptr;
    c2.unit_pair = P->sparse ? begin_pairs(P, stream, n) : nullptr;
    HIP_CHECK(scg::launch_combo(cp, P->ht1.t.len, R, n, c2, P->error_flag.as<int32_t>(), stream));
    timer.stop();
    if (P->sparse) finish_pairs(P, stream, n);
    P->total += n;
}

void launch_batch(scg_plan* P, const ScgReads& R, int64_t n, hipStream_t stream) {
    if (P->kind == scg_plan::DUAL_SE_DIAG) { launch_batch_se_diag(P, R, n, stream); return; }
    scg_plan::Timer timer(P, stream);
    if (P->kind == scg_plan::SINGLE) {
        ScgSingleParams sp;
        sp.scan = P->scan1;
        sp.tmpl = P->d_tmpl1.as<ScgTemplate>();
        sp.index = P->tab[0].view;
        sp.max_mm = P->max_mm1; sp.use_first = P->use_first;
        sp.fwd = P->ht1.fwd; sp.rev = P->ht1.rev;
            ScgCounters counts = plan_counters(P);
        const bool tally = use_tally(P, n) && R.max_len > 0 && R.max_len <= 320 && !general_only(P);
        if (tally) {
            DevBuf& buf = P->unit_index[stream];              // batches on different streams may be in flight together
            buf.ensure(static_cast<size_t>(n) * sizeof(int32_t));
            counts.unit_index = buf.as<int32_t>();
        }
        HIP_CHECK(scg::launch_single(sp, P->ht1.t.len, R, n, counts, P->error_flag.as<int32_t>(), stream));
        if (tally) {
            timer.stop();                                      // kernel statistics cover the counting kernel, as in rocprof
            HIP_CHECK(scg::launch_tally(counts.unit_index, n, P->counters, P->n_counters, stream));
            P->total += n;
            return;
        }
    } else {
        ScgComboParams cp;
        cp.scan = P->scan1;
        cp.tmpl = P->d_tmpl1.as<ScgTemplate>();
        cp.index[0] = P->tab[0].view; cp.index[1] = P->tab[1].view;
        cp.n_pool[0] = P->n_pool[0]; cp.n_pool[1] = P->n_pool[1];
        cp.max_mm = P->max_mm1; cp.use_first = P->use_first;
        cp.fwd = P->ht1.fwd; cp.rev = P->ht1.rev;
        cp.only_if_negative = nullptr; cp.keep_first = 0; cp.pad = 0;
        ScgCounters counts = plan_counters(P);
        if (P->sparse) {
            counts.unit_pair = begin_pairs(P, stream, n);
            HIP_CHECK(scg::launch_combo(cp, P->ht1.t.len, R, n, counts, P->error_flag.as<int32_t>(), stream));
            timer.stop();
            finish_pairs(P, stream, n);
            P->total += n;
            return;
        }
        const bool tally = use_tally(P, n) && R.max_len > 0 && R.max_len <= 320 && !general_only(P);
        if (tally) {
            DevBuf& buf = P->unit_index[stream];
            buf.ensure(static_cast<size_t>(n) * sizeof(int32_t));
            counts.unit_index = buf.as<int32_t>();
        }
        HIP_CHECK(scg::launch_combo(cp, P->ht1.t.len, R, n, counts, P->error_flag.as<int32_t>(), stream));
        if (tally) {
            timer.stop();
            HIP_CHECK(scg::launch_tally(counts.unit_index, n, P->counters, P->n_counters, stream));
            P->total += n;
            return;
        }
    }
    timer.stop();
    fold_replicas(P, stream);
    P->total += n;
}

// Paired-end kernels search each template on ONE strand, fixed per plan: they get the scan description with that strand in
// the forward fields, so that they neither carry both strands' seeds and planes in SGPRs nor select between them at run time.
ScgScan searched_strand_first(const ScgScan& t, bool reverse) {
    if (!reverse) return t;
    ScgScan o = t;
    o.fseeds = t.rseeds; o.rseeds = t.fseeds;
    for (int r = 0; r < SCG_MAX_REGIONS; ++r) {
        o.fstart[r] = t.rstart[r]; o.rstart[r] = t.fstart[r];
        o.flen[r] = t.rlen[r]; o.rlen[r] = t.flen[r];
    }
    for (int w = 0; w < SCG_MAX_TEMPLATE / 32; ++w) {
        o.fplane0[w] = t.rplane0[w]; o.rplane0[w] = t.fplane0[w];
        o.fplane1[w] = t.rplane1[w]; o.rplane1[w] = t.fplane1[w];
        o.fmask[w] = t.rmask[w]; o.rmask[w] = t.fmask[w];
    }
    return o;
}

void launch_batch_paired(scg_plan* P, const ScgReads& R1, const ScgReads& R2, int64_t n, hipStream_t stream) {
    scg_plan::Timer timer(P, stream);
    ScgDualParams dp;
    dp.scan1 = searched_strand_first(P->scan1, P->rev1); dp.scan2 = searched_strand_first(P->scan2, P->rev2);
    dp.tmpl1 = P->d_tmpl1.as<ScgTemplate>(); dp.tmpl2 = P->d_tmpl2.as<ScgTemplate>();
    dp.index1 = P->tab[0].view; dp.index2 = P->tab[1].view; dp.pairs = P->pairs.view;
    dp.rev1 = P->rev1; dp.rev2 = P->rev2; dp.max_mm1 = P->max_mm1; dp.max_mm2 = P->max_mm2;
    dp.randomized = P->randomized; dp.use_first = P->use_first;
    dp.diagnostics = P->diagnostics; dp.n_pool = P->diagnostics == 2 ? 0 : P->n_pool[0]; dp.n_uid2 = static_cast<int32_t>(P->first2.size());
    dp.keep_first = P->diagnostics == 1; dp.only_if_negative = nullptr;
    ScgCounters counts = plan_counters(P);
    if (P->sparse) counts.unit_pair = begin_pairs(P, stream, n);      // (the invalid / all combinations of the diagnostics passes)
    const int lo_len = std::min(R1.max_len, R2.max_len), hi_len = std::max(R1.max_len, R2.max_len);
    const bool staged = lo_len > 0 && hi_len <= 320 && !general_only(P);
    const int tmpl_len = std::max(P->ht1.t.len, P->ht2.t.len);
    dp.overflow = nullptr;
    if (staged && P->diagnostics != 2 && n < INT32_MAX) {
        DevBuf& buf = P->overflow[stream];
        buf.ensure((static_cast<size_t>(n) + 1) * sizeof(int32_t));
        dp.overflow = buf.as<int32_t>();
    }
    if (P->diagnostics == 1 && staged) {
        // include.invalid=TRUE in two lean passes: valid pairs as an index stream (tallied), then the mate-by-mate
        // search on the pairs that found none
        DevBuf& buf = P->unit_index[stream];
        buf.ensure(static_cast<size_t>(n) * sizeof(int32_t));
        ScgCounters c1 = counts;
        c1.unit_index = buf.as<int32_t>();
        c1.unit_pair = nullptr;
        dp.diagnostics = 0;
        HIP_CHECK(scg::launch_dual(dp, tmpl_len, R1, R2, n, c1, P->error_flag.as<int32_t>(), stream));
        HIP_CHECK(scg::launch_tally(c1.unit_index, n, P->counters, P->n_pool[0], stream));
        dp.diagnostics = 2;
        dp.only_if_negative = c1.unit_index;
        HIP_CHECK(scg::launch_dual(dp, tmpl_len, R1, R2, n, counts, P->error_flag.as<int32_t>(), stream));
        timer.stop();
        if (P->sparse) finish_pairs(P, stream, n);
        fold_replicas(P, stream);
        HIP_CHECK(scg::launch_hot_fold(P->hot.as<int32_t>(), P->counters + P->n_pool[0], stream));
        P->total += n;
        return;
    }
    const bool tally = use_tally(P, n) && staged;
    if (tally) {
        DevBuf& buf = P->unit_index[stream];
        buf.ensure(static_cast<size_t>(n) * sizeof(int32_t));
        counts.unit_index = buf.as<int32_t>();
    }
    HIP_CHECK(scg::launch_dual(dp, tmpl_len, R1, R2, n, counts, P->error_flag.as<int32_t>(), stream));
    timer.stop();
    if (P->sparse) finish_pairs(P, stream, n);
    if (tally) HIP_CHECK(scg::launch_tally(counts.unit_index, n, P->counters, P->n_counters, stream));
    else fold_replicas(P, stream);
    if (P->hot.p) HIP_CHECK(scg::launch_hot_fold(P->hot.as<int32_t>(), P->counters + (P->diagnostics == 2 ? 0 : P->n_pool[0]), stream));
    P->total += n;
}

} // namespace scgapi
