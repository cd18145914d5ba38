// scg_api.cpp -- the C ABI (include/scg.h): every entry point, over the plans (scg_plan.cpp) and the pipelines
// (scg_pipelines.cpp).
//
// Host-side counterpart of the reference's Rcpp glue (src/count_single_barcodes.cpp,
// src/count_combo_barcodes_single.cpp, src/count_dual_barcodes.cpp, src/match_barcodes.cpp): same argument order,
// same checks in the same order, status + message instead of exceptions.
#include "scg_internal.hpp"

// -------------------------------------------------------------------------------------------------
// C ABI
// -------------------------------------------------------------------------------------------------
#pragma GCC visibility push(default)     // the C ABI is the library's whole export list (csrc/Makefile: -fvisibility=hidden)
extern "C" {

const char* scg_version(void) { return "scg 0.3.0 (gfx950)"; }

int scg_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

void scg_free(void* p) { std::free(p); }

void scg_release_buffers(void) {
    release_cached_slots();
    scg::ParallelGunzip::release_cached();
    scg::release_device_gunzip_scratch();
}

int scg_parse_fastq(const char* path, char** seqs_out, uint64_t** offsets_out, int64_t* n_reads_out, char* err, size_t errcap) {
    return guarded(err, errcap, [&] {
        if (!path || !seqs_out || !offsets_out || !n_reads_out) throw Error(SCG_ERR_INVALID, "null argument");
        scg::FastqStream fq(path);
        scg::ReadBatch all, b;
        all.clear();
        bool done = false;
        const int threads = scg::default_host_threads(1);
        if (threads > 1 && scg::ParallelFastq::is_plain_file(path)) {
            scg::ParallelFastq pf(path, threads);
            std::vector<scg::ReadBatch> window;
            while (pf.next_window(window)) {
                for (auto& w : window) {
                    uint64_t base = all.seqs.size();
                    all.seqs.insert(all.seqs.end(), w.seqs.begin(), w.seqs.end());
                    for (int64_t i = 1; i <= w.size(); ++i) all.offsets.push_back(base + w.offsets[i]);
                }
            }
            done = !pf.unusual();
            if (!done) all.clear();
        }
        while (!done && fq.next_batch(b, BATCH_READS, BATCH_BYTES)) {
            uint64_t base = all.seqs.size();
            all.seqs.insert(all.seqs.end(), b.seqs.begin(), b.seqs.end());
            for (int64_t i = 1; i <= b.size(); ++i) all.offsets.push_back(base + b.offsets[i]);
        }
        char* s = static_cast<char*>(std::malloc(all.seqs.size() + 1));
        uint64_t* o = static_cast<uint64_t*>(std::malloc(sizeof(uint64_t) * all.offsets.size()));
        if (!s || !o) { std::free(s); std::free(o); throw std::bad_alloc(); }
        if (!all.seqs.empty()) std::memcpy(s, all.seqs.data(), all.seqs.size());
        std::memcpy(o, all.offsets.data(), sizeof(uint64_t) * all.offsets.size());
        *seqs_out = s; *offsets_out = o; *n_reads_out = all.size();
    });
}

int scg_fastq_text_windows(const char* path, int64_t window_bytes, int nthreads, char** text_out, int64_t* n_bytes_out,
                           int64_t** cuts_out, int64_t* n_windows_out, char* kind_out, char* err, size_t errcap) {
    return guarded(err, errcap, [&] {
        if (!path || !text_out || !n_bytes_out || !cuts_out || !n_windows_out || window_bytes < 64) throw Error(SCG_ERR_INVALID, "null argument");
        std::vector<char> all, window(static_cast<size_t>(window_bytes));
        std::vector<int64_t> cuts(1, 0);
        for (int attempt = 0; attempt < 2; ++attempt) {
            std::unique_ptr<scg::TextSource> src = scg::TextSource::open(path, scg::default_host_threads(nthreads), attempt == 0);
            if (kind_out) { std::strncpy(kind_out, src->kind(), 15); kind_out[15] = 0; }
            bool again = false;
            all.clear(); cuts.assign(1, 0);
            for (;;) {
                const size_t got = src->next(window.data(), window.size());
                if (src->unusual()) {
                    // (a gzip file the parallel decoder hands back is read again through one inflate stream, as the pipelines do)
                    if (attempt == 0 && is_parallel_gzip(src.get())) { again = true; break; }
                    throw Error(SCG_ERR_UNSUPPORTED, "the text cannot be cut into windows of whole 4-line records");
                }
                if (!got) break;
                all.insert(all.end(), window.begin(), window.begin() + got);
                cuts.push_back(static_cast<int64_t>(all.size()));
            }
            if (!again) break;
        }
        char* t = static_cast<char*>(std::malloc(all.size() + 1));
        int64_t* c = static_cast<int64_t*>(std::malloc(sizeof(int64_t) * cuts.size()));
        if (!t || !c) { std::free(t); std::free(c); throw std::bad_alloc(); }
        if (!all.empty()) std::memcpy(t, all.data(), all.size());
        std::memcpy(c, cuts.data(), sizeof(int64_t) * cuts.size());
        *text_out = t; *n_bytes_out = static_cast<int64_t>(all.size());
        *cuts_out = c; *n_windows_out = static_cast<int64_t>(cuts.size()) - 1;
    });
}

int scg_fastq_scan_windows(const char* path, int64_t window_bytes, int nthreads, char** seqs_out, uint64_t** offsets_out,
                           int64_t* n_reads_out, int64_t* n_windows_out, char* err, size_t errcap) {
    return guarded(err, errcap, [&] {
        if (!path || !seqs_out || !offsets_out || !n_reads_out || !n_windows_out || window_bytes < 64) throw Error(SCG_ERR_INVALID, "null argument");
        std::unique_ptr<scg::TextSource> src = scg::TextSource::open(path, scg::default_host_threads(nthreads));
        if (!src->parses()) throw Error(SCG_ERR_UNSUPPORTED, std::string("no host-side record scan for ") + src->kind() + " input");
        const size_t cap = static_cast<size_t>(window_bytes), cap_offsets = cap / 4 + 64;
        std::vector<char> all, window(cap);
        std::vector<uint32_t> offs(cap_offsets);
        std::vector<uint64_t> offsets(1, 0);
        int64_t n_windows = 0;
        for (;;) {
            scg::ParsedWindow w;
            const size_t got = src->next_parsed(window.data(), cap, offs.data(), cap_offsets, w);
            if (src->unusual()) throw Error(SCG_ERR_UNSUPPORTED, "the text is not a run of ordinary 4-line records");
            if (!got) break;
            ++n_windows;
            for (int i = 0; i < w.n_segs; ++i) {
                const scg::ParsedSegment& g = w.seg[i];
                const uint64_t base = all.size();
                all.insert(all.end(), window.begin() + static_cast<long>(g.seq_at), window.begin() + static_cast<long>(g.seq_at + g.seq_bytes));
                for (uint32_t r = 0; r < g.n_records; ++r) offsets.push_back(base + offs[g.off_at + r + 1]);
            }
        }
        char* t = static_cast<char*>(std::malloc(all.size() + 1));
        uint64_t* o = static_cast<uint64_t*>(std::malloc(sizeof(uint64_t) * offsets.size()));
        if (!t || !o) { std::free(t); std::free(o); throw std::bad_alloc(); }
        if (!all.empty()) std::memcpy(t, all.data(), all.size());
        std::memcpy(o, offsets.data(), sizeof(uint64_t) * offsets.size());
        *seqs_out = t; *offsets_out = o;
        *n_reads_out = static_cast<int64_t>(offsets.size()) - 1;
        *n_windows_out = n_windows;
    });
}

int scg_bgzf_member_batches(const char* path, int64_t staging_bytes, int64_t text_bytes, int nthreads, uint32_t** table_out, int64_t* n_members_out,
                            char** payloads_out, int64_t* n_payload_bytes_out, int64_t* n_batches_out, char* err, size_t errcap) {
    return guarded(err, errcap, [&] {
        if (!path || !table_out || !n_members_out || !payloads_out || !n_payload_bytes_out || !n_batches_out || staging_bytes < 64 || text_bytes < 1) {
            throw Error(SCG_ERR_INVALID, "null argument");
        }
        std::unique_ptr<scg::TextSource> src = scg::TextSource::open(path, scg::default_host_threads(nthreads));
        if (!src->has_members()) throw Error(SCG_ERR_UNSUPPORTED, std::string("no member batches for ") + src->kind() + " input");
        const size_t slack = scg::inflate_input_slack();
        std::vector<char> staging(static_cast<size_t>(staging_bytes) + slack), all;
        std::vector<uint32_t> table;
        std::vector<scg::CompressedMember> members;
        int64_t batches = 0;
        for (;;) {
            size_t text = 0;
            bool last = false;
            const size_t got = src->next_members(staging.data(), staging.size(), slack, static_cast<size_t>(text_bytes), members, text, last);
            if (src->unusual()) throw Error(SCG_ERR_UNSUPPORTED, "a member the device inflater does not take (not BGZF throughout, extra header fields, or larger than a batch)");
            if (!got) break;
            for (const scg::CompressedMember& m : members) {
                const uint32_t row[6] = {static_cast<uint32_t>(batches), static_cast<uint32_t>(all.size() + m.in_off), m.in_len, m.out_off, m.out_len, m.crc};
                table.insert(table.end(), row, row + 6);
            }
            all.insert(all.end(), staging.begin(), staging.begin() + static_cast<long>(got - slack));
            ++batches;
            if (last) break;
        }
        uint32_t* t = static_cast<uint32_t*>(std::malloc(sizeof(uint32_t) * (table.size() + 1)));
        char* p = static_cast<char*>(std::malloc(all.size() + 1));
        if (!t || !p) { std::free(t); std::free(p); throw std::bad_alloc(); }
        if (!table.empty()) std::memcpy(t, table.data(), sizeof(uint32_t) * table.size());
        if (!all.empty()) std::memcpy(p, all.data(), all.size());
        *table_out = t; *n_members_out = static_cast<int64_t>(table.size() / 6);
        *payloads_out = p; *n_payload_bytes_out = static_cast<int64_t>(all.size());
        *n_batches_out = batches;
    });
}

int scg_set_device(int device, char* err, size_t errcap) {
    return guarded(err, errcap, [&] {
        int n = 0;
        if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) throw Error(SCG_ERR_DEVICE, "no HIP device available: libscg has no CPU fallback");
        if (device < 0 || device >= n) {
            throw Error(SCG_ERR_DEVICE, "HIP device " + std::to_string(device) + " out of range (" + std::to_string(n) + " visible)");
        }
        HIP_CHECK(hipSetDevice(device));
    });
}

int scg_set_devices(const int* devices, int n, char* err, size_t errcap) {
    return guarded(err, errcap, [&] {
        if (n < 0 || (n > 0 && !devices)) throw Error(SCG_ERR_INVALID, "null argument");
        set_thread_devices(devices, n);
    });
}

int scg_plan_single(scg_plan** plan_out, const char* constant, int strand, const char* const* pool, int32_t n_pool,
                    int mismatches, int use_first, int device, char* err, size_t errcap) {
    return guarded(err, errcap, [&] {
        if (!plan_out) throw Error(SCG_ERR_INVALID, "null argument");
        auto P = compile_single(constant, strand, pool, n_pool, mismatches, use_first);
        P->to_device(device);
        *plan_out = P.release();
    });
}

int scg_plan_combo(scg_plan** plan_out, const char* constant, int strand, const char* const* pool0, int32_t n_pool0,
                   const char* const* pool1, int32_t n_pool1, int mismatches, int use_first,
                   int device, char* err, size_t errcap) {
    return guarded(err, errcap, [&] {
        if (!plan_out) throw Error(SCG_ERR_INVALID, "null argument");
        auto P = compile_combo(constant, strand, pool0, n_pool0, pool1, n_pool1, mismatches, use_first);
        P->to_device(device);
        *plan_out = P.release();
    });
}

int scg_plan_dual(scg_plan** plan_out, const char* constant1, int reverse1, int mismatches1, const char* const* pool1,
                  const char* constant2, int reverse2, int mismatches2, const char* const* pool2,
                  int32_t n_pool, int randomized, int use_first, int diagnostics, int device, char* err, size_t errcap) {
    return guarded(err, errcap, [&] {
        if (!plan_out) throw Error(SCG_ERR_INVALID, "null argument");
        auto P = compile_dual(constant1, reverse1, mismatches1, pool1, constant2, reverse2, mismatches2, pool2, n_pool, randomized, use_first, diagnostics);
        P->to_device(device);
        *plan_out = P.release();
    });
}

int scg_plan_dual_single_end(scg_plan** plan_out, const char* constant, int strand, const char* const* const* pools, const int32_t* n_pools,
                             int32_t n_regions, int mismatches, int use_first, int device, char* err, size_t errcap) {
    return guarded(err, errcap, [&] {
        if (!plan_out) throw Error(SCG_ERR_INVALID, "null argument");
        *plan_out = nullptr;
        auto P = compile_dual_single_end(constant, strand, pools, n_pools, n_regions, mismatches, use_first);
        P->to_device(device);
        *plan_out = P.release();
    });
}

int scg_plan_paired_combo(scg_plan** plan_out,
                          const char* constant1, int reverse1, int mismatches1, const char* const* pool1, int32_t n_pool1,
                          const char* constant2, int reverse2, int mismatches2, const char* const* pool2, int32_t n_pool2,
                          int randomized, int use_first, int device, char* err, size_t errcap) {
    return guarded(err, errcap, [&] {
        if (!plan_out) throw Error(SCG_ERR_INVALID, "null argument");
        *plan_out = nullptr;
        auto P = compile_paired_combo(constant1, reverse1, mismatches1, pool1, n_pool1, constant2, reverse2, mismatches2, pool2, n_pool2,
                                      randomized, use_first);
        P->to_device(device);
        *plan_out = P.release();
    });
}

void scg_plan_destroy(scg_plan* plan) {
    if (!plan) return;
    int prev = -1;
    if (hipGetDevice(&prev) == hipSuccess && prev != plan->device) (void)hipSetDevice(plan->device); else prev = -1;
    delete plan;
    if (prev >= 0) (void)hipSetDevice(prev);
}

int64_t scg_plan_num_counters(const scg_plan* plan) { return plan ? plan->n_counters : 0; }

int32_t* scg_plan_device_counters(scg_plan* plan) { return plan ? plan->counters : nullptr; }

int scg_plan_bind_counters(scg_plan* plan, int32_t* d_counters, char* err, size_t errcap) {
    return guarded(err, errcap, [&] {
        if (!plan) throw Error(SCG_ERR_INVALID, "null plan");
        plan->counters = d_counters ? d_counters : plan->own_counters.as<int32_t>();
    });
}

int scg_plan_reset(scg_plan* plan, void* stream, char* err, size_t errcap) {
    return guarded(err, errcap, [&] {
        if (!plan) throw Error(SCG_ERR_INVALID, "null plan");
        DeviceGuard g(plan->device);
        HIP_CHECK(hipMemsetAsync(plan->counters, 0, static_cast<size_t>(plan->n_counters) * sizeof(int32_t), static_cast<hipStream_t>(stream)));
        HIP_CHECK(hipMemsetAsync(plan->error_flag.p, 0, sizeof(int32_t), static_cast<hipStream_t>(stream)));
        for (auto& kv : plan->pair_stream) {
            if (kv.second.pending) { HIP_CHECK(hipEventSynchronize(kv.second.done)); kv.second.pending = 0; }
        }
        plan->sparse_counts.clear();
        plan->total = 0;
    });
}

int scg_count_batch(scg_plan* plan, const char* d_seqs, const uint32_t* d_offsets, int32_t fixed_len, int32_t max_len,
                    int64_t n_reads, void* stream, char* err, size_t errcap) {
    return guarded(err, errcap, [&] {
        if (!plan) throw Error(SCG_ERR_INVALID, "null plan");
        if (plan->kind == scg_plan::DUAL) throw Error(SCG_ERR_INVALID, "scg_count_batch called on a dual plan; use scg_count_batch_paired");
        check_reads_args(d_seqs, d_offsets, fixed_len, n_reads);
        DeviceGuard g(plan->device);
        launch_batch(plan, make_reads(d_seqs, d_offsets, fixed_len, max_len), n_reads, static_cast<hipStream_t>(stream));
    });
}

int scg_count_batch_paired(scg_plan* plan, const char* d_seqs1, const uint32_t* d_offsets1, int32_t fixed_len1,
                           const char* d_seqs2, const uint32_t* d_offsets2, int32_t fixed_len2, int32_t max_len,
                           int64_t n_pairs, void* stream, char* err, size_t errcap) {
    return guarded(err, errcap, [&] {
        if (!plan) throw Error(SCG_ERR_INVALID, "null plan");
        if (plan->kind != scg_plan::DUAL) throw Error(SCG_ERR_INVALID, "scg_count_batch_paired needs a dual plan");
        check_reads_args(d_seqs1, d_offsets1, fixed_len1, n_pairs);
        check_reads_args(d_seqs2, d_offsets2, fixed_len2, n_pairs);
        DeviceGuard g(plan->device);
        launch_batch_paired(plan, make_reads(d_seqs1, d_offsets1, fixed_len1, max_len), make_reads(d_seqs2, d_offsets2, fixed_len2, max_len),
                            n_pairs, static_cast<hipStream_t>(stream));
    });
}

int scg_plan_read(scg_plan* plan, int32_t* counts_out, int64_t* total_out, void* stream, char* err, size_t errcap) {
    return guarded(err, errcap, [&] {
        if (!plan) throw Error(SCG_ERR_INVALID, "null plan");
        DeviceGuard g(plan->device);
        HIP_CHECK(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
        read_counters(plan, counts_out);
        if (total_out) *total_out = plan->total;
    });
}

int scg_plan_read_combinations(scg_plan* plan, int32_t** indices_out, int32_t** freq_out, int64_t* k_out, int64_t* total_out,
                               void* stream, char* err, size_t errcap) {
    return guarded(err, errcap, [&] {
        if (!plan || !indices_out || !freq_out || !k_out) throw Error(SCG_ERR_INVALID, "null argument");
        if (plan->kind != scg_plan::COMBO) throw Error(SCG_ERR_INVALID, "not a combination plan");
        DeviceGuard g(plan->device);
        HIP_CHECK(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
        std::vector<int32_t> cells(static_cast<size_t>(plan->n_counters) + 1);
        read_counters(plan, cells.data());
        if (plan->sparse) { retire_all_pairs(plan); combos_from_sparse(plan->sparse_counts, indices_out, freq_out, k_out); }
        else combo_compact(cells.data(), plan->n_pool[0], plan->n_pool[1], indices_out, freq_out, k_out);
        if (total_out) *total_out = plan->total;
    });
}

int scg_combo_compact(const int32_t* cells, int32_t n_pool0, int32_t n_pool1,
                      int32_t** indices_out, int32_t** freq_out, int64_t* k_out, char* err, size_t errcap) {
    return guarded(err, errcap, [&] {
        if (!cells || !indices_out || !freq_out || !k_out) throw Error(SCG_ERR_INVALID, "null argument");
        combo_compact(cells, n_pool0, n_pool1, indices_out, freq_out, k_out);
    });
}

int scg_plan_set_profiling(scg_plan* plan, int enabled) {
    if (!plan) return SCG_ERR_INVALID;
    plan->profiling = enabled != 0;
    plan->events_used = 0;   // (re)starting a measurement window
    return SCG_OK;
}

int scg_plan_kernel_stats(scg_plan* plan, double* total_ms_out, int64_t* launches_out, char* err, size_t errcap) {
    return guarded(err, errcap, [&] {
        if (!plan) throw Error(SCG_ERR_INVALID, "null plan");
        DeviceGuard g(plan->device);
        double ms = 0;
        for (size_t i = 0; i < plan->events_used; ++i) {
            HIP_CHECK(hipEventSynchronize(plan->events[i].second));
            float t = 0;
            HIP_CHECK(hipEventElapsedTime(&t, plan->events[i].first, plan->events[i].second));
            ms += t;
        }
        if (total_ms_out) *total_ms_out = ms;
        if (launches_out) *launches_out = static_cast<int64_t>(plan->events_used);
    });
}

// ---- file-level entry points -------------------------------------------------------------------

int scg_count_single_barcodes(const char* path, const char* constant, int strand, const char* const* pool, int32_t n_pool,
                              int mismatches, int use_first, int nthreads, int32_t* counts_out, int32_t* total_out,
                              char* err, size_t errcap) {
    return guarded(err, errcap, [&] {
        if (!path || !counts_out || !total_out) throw Error(SCG_ERR_INVALID, "null argument");
        Trace tr;
        scg::FastqStream fq(path);                                           // src/count_single_barcodes.cpp:30
        tr.mark("open");
        auto set = compile_and_count_single_end(path, fq, nthreads, [&] {
            return compile_single(constant, strand, pool, n_pool, mismatches, use_first);   // :31-47
        });
        tr.mark("compile + count");
        set->read(counts_out);
        *total_out = narrow_total(set->total());
        tr.mark("read counters");
        set.reset();
        tr.mark("release the plan");
    });
}

int scg_count_combo_barcodes_single(const char* path, const char* constant, int strand,
                                    const char* const* pool0, int32_t n_pool0, const char* const* pool1, int32_t n_pool1,
                                    int mismatches, int use_first, int nthreads,
                                    int32_t** indices_out, int32_t** freq_out, int64_t* k_out, int32_t* total_out,
                                    char* err, size_t errcap) {
    return guarded(err, errcap, [&] {
        if (!path || !indices_out || !freq_out || !k_out || !total_out) throw Error(SCG_ERR_INVALID, "null argument");
        scg::FastqStream fq(path);
        auto set = compile_and_count_single_end(path, fq, nthreads, [&] {
            return compile_combo(constant, strand, pool0, n_pool0, pool1, n_pool1, mismatches, use_first);
        });
        std::vector<int32_t> cells(static_cast<size_t>(set->first()->n_counters) + 1);
        set->read(cells.data());
        const int32_t total = narrow_total(set->total());
        if (set->first()->sparse) combos_from_sparse(set->sparse_merged(), indices_out, freq_out, k_out);
        else combo_compact(cells.data(), n_pool0, n_pool1, indices_out, freq_out, k_out);
        *total_out = total;
    });
}

int scg_count_dual_barcodes(const char* path1, const char* constant1, int reverse1, int mismatches1, const char* const* pool1,
                            const char* path2, const char* constant2, int reverse2, int mismatches2, const char* const* pool2,
                            int32_t n_pool, int randomized, int use_first, int diagnostics, int nthreads,
                            int32_t* counts_out, int32_t* total_out, char* err, size_t errcap) {
    return guarded(err, errcap, [&] {
        if (!path1 || !path2 || !counts_out || !total_out) throw Error(SCG_ERR_INVALID, "null argument");
        scg::FastqStream fq1(path1);                                         // src/count_dual_barcodes.cpp:93-97
        scg::FastqStream fq2(path2);
        if (diagnostics) {
            throw Error(SCG_ERR_INVALID, "diagnostics requested: call scg_count_dual_barcodes_diagnostics, which returns the extra outputs");
        }
        auto set = compile_and_count_paired(path1, path2, fq1, fq2, nthreads, [&] {
            return compile_dual(constant1, reverse1, mismatches1, pool1, constant2, reverse2, mismatches2, pool2, n_pool, randomized, use_first);
        });
        set->read(counts_out);
        *total_out = narrow_total(set->total());
    });
}

int scg_count_dual_barcodes_diagnostics(const char* path1, const char* constant1, int reverse1, int mismatches1, const char* const* pool1,
                                        const char* path2, const char* constant2, int reverse2, int mismatches2, const char* const* pool2,
                                        int32_t n_pool, int randomized, int use_first, int nthreads,
                                        int32_t* counts_out, int32_t** invalid_indices_out, int32_t** invalid_freq_out, int64_t* k_out,
                                        int32_t* total_out, int32_t* barcode1_only_out, int32_t* barcode2_only_out,
                                        char* err, size_t errcap) {
    return guarded(err, errcap, [&] {
        if (!path1 || !path2 || !counts_out || !invalid_indices_out || !invalid_freq_out || !k_out || !total_out ||
            !barcode1_only_out || !barcode2_only_out) {
            throw Error(SCG_ERR_INVALID, "null argument");
        }
        scg::FastqStream fq1(path1);
        scg::FastqStream fq2(path2);
        auto set = compile_and_count_paired(path1, path2, fq1, fq2, nthreads, [&] {
            return compile_dual(constant1, reverse1, mismatches1, pool1, constant2, reverse2, mismatches2, pool2, n_pool, randomized, use_first, 1);
        });
        std::vector<int32_t> all(static_cast<size_t>(set->first()->n_counters) + 1);
        set->read(all.data());
        const auto sparse = set->first()->sparse ? set->sparse_merged() : std::unordered_map<uint64_t, int64_t>();
        diagnostics_from_counters(set->first(), all, counts_out, invalid_indices_out, invalid_freq_out, k_out, barcode1_only_out, barcode2_only_out,
                                  set->first()->sparse ? &sparse : nullptr);
        *total_out = narrow_total(set->total());
    });
}

int scg_count_dual_barcodes_single_end(const char* path, const char* constant, const char* const* const* pools, const int32_t* n_pools,
                                       int32_t n_regions, int strand, int mismatches, int use_first, int diagnostics, int nthreads,
                                       int32_t* counts_out, int32_t* total_out, char* err, size_t errcap) {
    return guarded(err, errcap, [&] {
        if (!path || !total_out || (n_regions > 0 && n_pools && n_pools[0] > 0 && !counts_out)) throw Error(SCG_ERR_INVALID, "null argument");
        scg::FastqStream fq(path);                             // src/count_dual_barcodes_single_end.cpp:64: reader first
        auto set = compile_and_count_single_end(path, fq, nthreads, [&] {
            auto P = compile_dual_single_end(constant, strand, pools, n_pools, n_regions, mismatches, use_first);
            if (diagnostics) {
                throw Error(SCG_ERR_INVALID, "diagnostics requested: call scg_count_dual_barcodes_single_end_diagnostics, which returns the extra outputs");
            }
            return P;
        });
        set->read(counts_out);
        *total_out = narrow_total(set->total());
    });
}

// countRandomBarcodes (src/count_random_barcodes.cpp:41-62, kaori::RandomBarcodeSingleEnd): the device
// locates the template in every read (same scan kernels, no library), the host cuts the variable region
// out of its copy of the batch and tallies the strings.  Reproduced quirks of the reference:
//  * the forward-strand string is the raw read bytes (case preserved);
//  * on the reverse strand the region is taken at the FORWARD template's offset inside the window
//    (RandomBarcodeSingleEnd.hpp:103-105 reads variable_regions()[0], not the reverse regions) and then
//    reverse-complemented with complement_base<true>: ACGTN in either case -> upper case, anything
//    else is the error "cannot complement unknown base".
// Output order: byte-wise ascending (the reference iterates an unordered_map; its R caller sorts).
int scg_count_random_barcodes(const char* path, const char* constant, int strand, int mismatches, int use_first, int nthreads,
                              char** sequences_out, int32_t** freq_out, int64_t* k_out, int32_t* length_out, int32_t* total_out,
                              char* err, size_t errcap) {
    return guarded(err, errcap, [&] {
        if (!path || !constant || !sequences_out || !freq_out || !k_out || !length_out || !total_out) throw Error(SCG_ERR_INVALID, "null argument");
        scg::FastqStream fq(path);                             // reader first (src/count_random_barcodes.cpp:42)
        std::unique_ptr<scg_plan> P(new scg_plan);
        P->kind = scg_plan::SINGLE;
        P->ht1 = scg::parse_template(constant, strand);
        const ScgTemplate& t = P->ht1.t;
        if (t.nreg < 1) throw Error(SCG_ERR_INVALID, "expected one variable region in the constant template");
        if (t.nreg > SCG_MAX_REGIONS) throw Error(SCG_ERR_UNSUPPORTED, "this engine handles templates with at most " + std::to_string(SCG_MAX_REGIONS) + " variable regions");
        if (mismatches < 0) throw Error(SCG_ERR_INVALID, "negative number of mismatches");
        P->scan1 = scg::build_scan(t, mismatches);
        P->max_mm1 = mismatches;
        P->use_first = use_first != 0;
        P->n_counters = 0;
        P->to_device(-1);
        DeviceGuard g(P->device);
        const int vstart = t.fstart[0], vlen = t.flen[0];      // forward coordinates on both strands (see above)
        std::unordered_map<std::string, int32_t> tally;
        std::string key(static_cast<size_t>(vlen), ' ');
        ScgSingleParams sp;
        sp.scan = P->scan1;
        sp.tmpl = P->d_tmpl1.as<ScgTemplate>();
        std::memset(&sp.index, 0, sizeof(sp.index));
        sp.max_mm = mismatches; sp.use_first = P->use_first;
        sp.fwd = P->ht1.fwd; sp.rev = P->ht1.rev;
        auto launch = [&](Stager::Slot& s, const ScgReads& R, int64_t n) {
            s.d_aux.ensure(static_cast<size_t>(n) * sizeof(int32_t));
            s.h_aux.ensure(static_cast<size_t>(n) * sizeof(int32_t));
            HIP_CHECK(scg::launch_random(sp, t.len, R, n, s.d_aux.as<int32_t>(), P->error_flag.as<int32_t>(), s.stream));
            HIP_CHECK(hipMemcpyAsync(s.h_aux.p, s.d_aux.p, static_cast<size_t>(n) * sizeof(int32_t), hipMemcpyDeviceToHost, s.stream));
            P->total += n;
        };
        auto retire = [&](Stager::Slot& s) {
            const int32_t* hits = s.h_aux.as<int32_t>();
            const char* seqs = s.h_seqs[0].as<char>();
            const uint32_t* offs = s.h_offs[0].as<uint32_t>();
            for (int64_t i = 0; i < s.n_reads; ++i) {
                const int32_t h = hits[i];
                if (h < 0) continue;
                const char* start = seqs + offs[i] + (h >> 1) + vstart;
                if (!(h & 1)) {
                    key.assign(start, static_cast<size_t>(vlen));
                } else {
                    for (int j = 0; j < vlen; ++j) {
                        char b = start[vlen - j - 1], o;
                        switch (b) {                            // kaori/utils.hpp:41-120, complement_base<true>
                            case 'A': case 'a': o = 'T'; break;
                            case 'C': case 'c': o = 'G'; break;
                            case 'G': case 'g': o = 'C'; break;
                            case 'T': case 't': o = 'A'; break;
                            case 'N': case 'n': o = 'N'; break;
                            default: throw Error(SCG_ERR_INVALID, std::string("cannot complement unknown base '") + b + "'");
                        }
                        key[static_cast<size_t>(j)] = o;
                    }
                }
                ++tally[key];
            }
        };
        auto restart = [&] { tally.clear(); };
        count_single_end_file(P.get(), path, fq, nthreads, launch, retire, restart);
        read_counters(P.get(), nullptr);                       // surfaces the oversize-read flag
        std::vector<std::pair<std::string, int32_t> > rows(tally.begin(), tally.end());
        std::sort(rows.begin(), rows.end());
        const size_t stride = static_cast<size_t>(vlen) + 1;
        char* so = static_cast<char*>(std::malloc(rows.size() * stride + 1));
        int32_t* fo = static_cast<int32_t*>(std::malloc(sizeof(int32_t) * (rows.size() + 1)));
        if (!so || !fo) { std::free(so); std::free(fo); throw std::bad_alloc(); }
        for (size_t i = 0; i < rows.size(); ++i) {
            std::memcpy(so + i * stride, rows[i].first.data(), static_cast<size_t>(vlen));
            so[i * stride + vlen] = 0;
            fo[i] = rows[i].second;
        }
        *sequences_out = so; *freq_out = fo; *k_out = static_cast<int64_t>(rows.size()); *length_out = vlen;
        *total_out = narrow_total(P->total);
    });
}

int scg_count_dual_barcodes_single_end_diagnostics(const char* path, const char* constant, const char* const* const* pools, const int32_t* n_pools,
                                                   int32_t n_regions, int strand, int mismatches, int use_first, int nthreads,
                                                   int32_t* counts_out, int32_t** invalid_indices_out, int32_t** invalid_freq_out, int64_t* k_out,
                                                   int32_t* total_out, char* err, size_t errcap) {
    return guarded(err, errcap, [&] {
        if (!path || !invalid_indices_out || !invalid_freq_out || !k_out || !total_out) throw Error(SCG_ERR_INVALID, "null argument");
        scg::FastqStream fq(path);
        auto set = compile_and_count_single_end(path, fq, nthreads, [&] {
            return compile_dual_single_end_diag(constant, strand, pools, n_pools, n_regions, mismatches, use_first);
        });
        std::vector<int32_t> all(static_cast<size_t>(set->first()->n_counters) + 1);
        set->read(all.data());
        const int32_t total = narrow_total(set->total());
        int32_t b1 = 0, b2 = 0;
        const auto sparse = set->first()->sparse ? set->sparse_merged() : std::unordered_map<uint64_t, int64_t>();
        diagnostics_from_counters(set->first(), all, counts_out, invalid_indices_out, invalid_freq_out, k_out, &b1, &b2, set->first()->sparse ? &sparse : nullptr);
        *total_out = total;
    });
}

int scg_count_combo_barcodes_paired(const char* path1, const char* constant1, int reverse1, int mismatches1,
                                    const char* const* pool1, int32_t n_pool1,
                                    const char* path2, const char* constant2, int reverse2, int mismatches2,
                                    const char* const* pool2, int32_t n_pool2,
                                    int randomized, int use_first, int nthreads,
                                    int32_t** indices_out, int32_t** freq_out, int64_t* k_out,
                                    int32_t* total_out, int32_t* barcode1_only_out, int32_t* barcode2_only_out,
                                    char* err, size_t errcap) {
    return guarded(err, errcap, [&] {
        if (!path1 || !path2 || !indices_out || !freq_out || !k_out || !total_out || !barcode1_only_out || !barcode2_only_out) {
            throw Error(SCG_ERR_INVALID, "null argument");
        }
        scg::FastqStream fq1(path1);                           // src/count_combo_barcodes_paired.cpp:75-79: readers first
        scg::FastqStream fq2(path2);
        auto set = compile_and_count_paired(path1, path2, fq1, fq2, nthreads, [&] {
            return compile_paired_combo(constant1, reverse1, mismatches1, pool1, n_pool1, constant2, reverse2, mismatches2, pool2, n_pool2,
                                        randomized, use_first);
        });
        std::vector<int32_t> all(static_cast<size_t>(set->first()->n_counters) + 1);
        set->read(all.data());
        const auto sparse = set->first()->sparse ? set->sparse_merged() : std::unordered_map<uint64_t, int64_t>();
        diagnostics_from_counters(set->first(), all, nullptr, indices_out, freq_out, k_out, barcode1_only_out, barcode2_only_out,
                                  set->first()->sparse ? &sparse : nullptr);
        *total_out = narrow_total(set->total());
    });
}

// ---- many files in one call (matrixOf*) ---------------------------------------------------------------------

int scg_count_single_barcodes_files(const char* const* paths, int32_t n_files, const char* constant, int strand,
                                    const char* const* pool, int32_t n_pool, int mismatches, int use_first, int nthreads,
                                    int32_t* counts_out, int32_t* totals_out, char* err, size_t errcap) {
    return guarded(err, errcap, [&] {
        if (n_files < 0 || (n_files > 0 && (!paths || !totals_out || (n_pool > 0 && !counts_out)))) throw Error(SCG_ERR_INVALID, "null argument");
        for (int32_t f = 0; f < n_files; ++f) if (!paths[f]) throw Error(SCG_ERR_INVALID, "null argument");
        if (n_files == 0) return;
        if (n_files == 1) {      // one file: all devices share it
            const int rc = scg_count_single_barcodes(paths[0], constant, strand, pool, n_pool, mismatches, use_first, nthreads,
                                                     counts_out, totals_out, err, errcap);
            if (rc != SCG_OK) throw Error(rc, err ? err : "");
            return;
        }
        scg::FastqStream probe(paths[0]);                          // the first file's reader comes before the argument checks, as in a loop over files
        std::vector<int> devices = device_list();
        if (devices.size() > static_cast<size_t>(n_files)) devices.resize(static_cast<size_t>(n_files));
        PlanSet set(compile_single(constant, strand, pool, n_pool, mismatches, use_first), devices);
        const size_t stride = static_cast<size_t>(n_pool);
        schedule_files(n_files, set, [&](scg_plan* P, int32_t f) {
            scg::FastqStream fq(paths[f]);
            reset_plan(P);
            count_single_end(std::vector<scg_plan*>(1, P), paths[f], fq, nthreads);
            read_counters(P, counts_out + stride * static_cast<size_t>(f));
            totals_out[f] = narrow_total(P->total);
        });
    });
}

int scg_count_combo_barcodes_single_files(const char* const* paths, int32_t n_files, const char* constant, int strand,
                                          const char* const* pool0, int32_t n_pool0, const char* const* pool1, int32_t n_pool1,
                                          int mismatches, int use_first, int nthreads,
                                          int32_t** indices_out, int32_t** freq_out, int64_t* k_out, int32_t* totals_out,
                                          char* err, size_t errcap) {
    return guarded(err, errcap, [&] {
        if (n_files < 0 || (n_files > 0 && (!paths || !indices_out || !freq_out || !k_out || !totals_out))) throw Error(SCG_ERR_INVALID, "null argument");
        for (int32_t f = 0; f < n_files; ++f) {
            if (!paths[f]) throw Error(SCG_ERR_INVALID, "null argument");
            indices_out[f] = nullptr; freq_out[f] = nullptr; k_out[f] = 0;
        }
        if (n_files == 0) return;
        try {
            scg::FastqStream probe(paths[0]);
            std::vector<int> devices = device_list();
            if (devices.size() > static_cast<size_t>(n_files)) devices.resize(static_cast<size_t>(n_files));
            PlanSet set(compile_combo(constant, strand, pool0, n_pool0, pool1, n_pool1, mismatches, use_first), devices);
            const size_t cells = static_cast<size_t>(set.first()->n_counters);
            schedule_files(n_files, set, [&](scg_plan* P, int32_t f) {
                scg::FastqStream fq(paths[f]);
                reset_plan(P);
                count_single_end(std::vector<scg_plan*>(1, P), paths[f], fq, nthreads);
                std::vector<int32_t> dense(cells + 1);
                read_counters(P, dense.data());
                totals_out[f] = narrow_total(P->total);
                if (P->sparse) { retire_all_pairs(P); combos_from_sparse(P->sparse_counts, &indices_out[f], &freq_out[f], &k_out[f]); }
                else combo_compact(dense.data(), n_pool0, n_pool1, &indices_out[f], &freq_out[f], &k_out[f]);
            });
        } catch (...) {
            for (int32_t f = 0; f < n_files; ++f) { std::free(indices_out[f]); std::free(freq_out[f]); indices_out[f] = nullptr; freq_out[f] = nullptr; k_out[f] = 0; }
            throw;
        }
    });
}

int scg_count_dual_barcodes_files(const char* const* paths1, const char* constant1, int reverse1, int mismatches1, const char* const* pool1,
                                  const char* const* paths2, const char* constant2, int reverse2, int mismatches2, const char* const* pool2,
                                  int32_t n_pool, int32_t n_files, int randomized, int use_first, int nthreads,
                                  int32_t* counts_out, int32_t* totals_out, char* err, size_t errcap) {
    return guarded(err, errcap, [&] {
        if (n_files < 0 || (n_files > 0 && (!paths1 || !paths2 || !totals_out || (n_pool > 0 && !counts_out)))) throw Error(SCG_ERR_INVALID, "null argument");
        for (int32_t f = 0; f < n_files; ++f) if (!paths1[f] || !paths2[f]) throw Error(SCG_ERR_INVALID, "null argument");
        if (n_files == 0) return;
        { scg::FastqStream probe1(paths1[0]); scg::FastqStream probe2(paths2[0]); }
        std::vector<int> devices = device_list();
        if (devices.size() > static_cast<size_t>(n_files)) devices.resize(static_cast<size_t>(n_files));
        PlanSet set(compile_dual(constant1, reverse1, mismatches1, pool1, constant2, reverse2, mismatches2, pool2, n_pool, randomized, use_first), devices);
        const size_t stride = static_cast<size_t>(n_pool);
        schedule_files(n_files, set, [&](scg_plan* P, int32_t f) {
            scg::FastqStream fq1(paths1[f]);
            scg::FastqStream fq2(paths2[f]);
            reset_plan(P);
            count_paired_files(P, paths1[f], paths2[f], fq1, fq2, nthreads);
            read_counters(P, counts_out + stride * static_cast<size_t>(f));
            totals_out[f] = narrow_total(P->total);
        });
    });
}

int scg_plan_read_diagnostics(scg_plan* plan, int32_t* counts_out, int32_t** invalid_indices_out, int32_t** invalid_freq_out,
                              int64_t* k_out, int64_t* total_out, int32_t* barcode1_only_out, int32_t* barcode2_only_out,
                              void* stream, char* err, size_t errcap) {
    return guarded(err, errcap, [&] {
        if (!plan || !invalid_indices_out || !invalid_freq_out || !k_out || !barcode1_only_out || !barcode2_only_out) {
            throw Error(SCG_ERR_INVALID, "null argument");
        }
        if (plan->kind != scg_plan::DUAL || !plan->diagnostics) throw Error(SCG_ERR_INVALID, "not a dual plan with diagnostics");
        DeviceGuard g(plan->device);
        HIP_CHECK(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
        std::vector<int32_t> all(static_cast<size_t>(plan->n_counters) + 1);
        read_counters(plan, all.data());
        if (plan->sparse) retire_all_pairs(plan);
        diagnostics_from_counters(plan, all, counts_out, invalid_indices_out, invalid_freq_out, k_out, barcode1_only_out, barcode2_only_out,
                                  plan->sparse ? &plan->sparse_counts : nullptr);
        if (total_out) *total_out = plan->total;
    });
}

int scg_match_barcodes(const char* const* sequences, int32_t n_sequences, const char* const* choices, int32_t n_choices,
                       int substitutions, int reverse, int32_t* index_out, int32_t* mismatches_out, char* err, size_t errcap) {
    return guarded(err, errcap, [&] {
        if ((n_sequences > 0 && (!sequences || !index_out || !mismatches_out)) || (n_choices > 0 && !choices)) {
            throw Error(SCG_ERR_INVALID, "null argument");
        }
        int clen = scg::pool_length(choices, n_choices);                     // src/match_barcodes.cpp:12
        scg::HostIndex ht = clen > SCG_MAX_BARCODE ? scg::build_index_wide(choices, n_choices, clen, substitutions < 0 ? 0 : substitutions)
                                                   : scg::build_index(choices, n_choices, clen, substitutions < 0 ? 0 : substitutions);   // :13
        int slen = scg::pool_length(sequences, n_sequences);                 // :20
        if (n_sequences > 0 && slen != clen) {
            throw Error(SCG_ERR_INVALID, "sequences should have the same length as the choices (" + std::to_string(clen) + ")");
        }
        if (substitutions < 0) throw Error(SCG_ERR_INVALID, "negative number of mismatches");
        if (n_sequences == 0) return;
        int device = resolve_device(-1);
        DeviceGuard g(device);
        DevIndex tab;
        tab.upload(ht);
        std::vector<uint8_t> flat(static_cast<size_t>(n_sequences) * clen);
        for (int32_t i = 0; i < n_sequences; ++i) std::memcpy(flat.data() + static_cast<size_t>(i) * clen, sequences[i], clen);
        DevBuf d_seqs, d_idx, d_mm;
        d_seqs.upload(flat);
        d_idx.alloc(sizeof(int32_t) * n_sequences);
        d_mm.alloc(sizeof(int32_t) * n_sequences);
        HIP_CHECK(scg::launch_match(tab.view, d_seqs.as<uint8_t>(), n_sequences, substitutions, reverse, d_idx.as<int32_t>(), d_mm.as<int32_t>(), nullptr));
        HIP_CHECK(hipMemcpy(index_out, d_idx.p, sizeof(int32_t) * n_sequences, hipMemcpyDeviceToHost));
        HIP_CHECK(hipMemcpy(mismatches_out, d_mm.p, sizeof(int32_t) * n_sequences, hipMemcpyDeviceToHost));
    });
}

int scg_synth_reads(const scg_synth_spec* spec, char* d_seqs_out, int64_t n_reads, void* stream, char* err, size_t errcap) {
    return guarded(err, errcap, [&] {
        if (!spec || (!d_seqs_out && n_reads > 0)) throw Error(SCG_ERR_INVALID, "null argument");
        if (spec->read_len <= 0 || spec->template_len < 0 || spec->n_regions < 0 || spec->n_regions > 2) {
            throw Error(SCG_ERR_INVALID, "bad synthetic read specification");
        }
        HIP_CHECK(scg::launch_synth(*spec, d_seqs_out, n_reads, static_cast<hipStream_t>(stream)));
    });
}

} // extern "C"
#pragma GCC visibility pop
