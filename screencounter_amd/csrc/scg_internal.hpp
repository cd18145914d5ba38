// scg_internal.hpp -- what the three parts of the host side share: error plumbing, device buffers, the plan object, and the
// functions one part calls in another.  scg_plan.cpp: plan compilation and batch launches; scg_pipelines.cpp: FASTQ files
// to counts (staging, windows, devices); scg_api.cpp: the C entry points of include/scg.h.
#ifndef SCG_INTERNAL_HPP
#define SCG_INTERNAL_HPP
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <chrono>
#include <climits>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <atomic>
#include <sys/stat.h>
#include <memory>
#include <string>
#include <functional>
#include <thread>
#include <unordered_map>
#include <vector>

#include "scg_host.h"
#include "scg_ingest.h"
#include "scg_pgzip.hpp"
#include "scg_launch.h"
#include "scg_textscan.h"

using scg::Error;

namespace scgapi {

#define HIP_CHECK(expr)                                                                              \
    do {                                                                                             \
        hipError_t e_ = (expr);                                                                      \
        if (e_ != hipSuccess) {                                                                      \
            throw Error(SCG_ERR_DEVICE, std::string(#expr) + " failed: " + hipGetErrorString(e_));   \
        }                                                                                            \
    } while (0)

inline void copy_err(char* err, size_t cap, const char* msg) {
    if (err && cap) {
        std::strncpy(err, msg, cap - 1);
        err[cap - 1] = '\0';
    }
}

template<class F>
inline int guarded(char* err, size_t cap, F f) {
    try {
        f();
        if (err && cap) err[0] = '\0';
        return SCG_OK;
    } catch (const Error& e) {
        copy_err(err, cap, e.what());
        return e.code;
    } catch (const std::bad_alloc&) {
        copy_err(err, cap, "out of host memory");
        return SCG_ERR_DEVICE;
    } catch (const std::exception& e) {
        copy_err(err, cap, e.what());
        return SCG_ERR_INVALID;
    } catch (...) {
        copy_err(err, cap, "unknown error");
        return SCG_ERR_INVALID;
    }
}

// Opt-in stage timings on stderr (SCG_TRACE=1): where a file-level call spends its wall time.
struct Trace {
    bool on;
    std::chrono::steady_clock::time_point t0, last;
    Trace() : on(false) {
        const char* e = std::getenv("SCG_TRACE");
        on = e && *e && *e != '0';
        t0 = last = std::chrono::steady_clock::now();
    }
    void mark(const char* what) {
        if (!on) return;
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[scg] %-28s %8.2f ms  (total %8.2f ms)\n", what,
                     std::chrono::duration<double, std::milli>(now - last).count(), std::chrono::duration<double, std::milli>(now - t0).count());
        last = now;
    }
};

inline int resolve_device(int device) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        throw Error(SCG_ERR_DEVICE, "no HIP device available: libscg has no CPU fallback");
    }
    if (device < 0) {
        const char* env = std::getenv("SCG_DEVICE");
        if (env && *env) {
            device = std::atoi(env);
        } else {
            HIP_CHECK(hipGetDevice(&device));
        }
    }
    if (device < 0 || device >= n) {
        throw Error(SCG_ERR_DEVICE, "HIP device " + std::to_string(device) + " out of range (" + std::to_string(n) + " visible)");
    }
    return device;
}

// Makes `device` current for the calling thread for the lifetime of the guard.
struct DeviceGuard {
    int prev = -1;
    explicit DeviceGuard(int device) {
        HIP_CHECK(hipGetDevice(&prev));
        if (prev != device) HIP_CHECK(hipSetDevice(device)); else prev = -1;
    }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    DevBuf() {}
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { release(); }
    void release() { if (p) { (void)hipFree(p); p = nullptr; bytes = 0; } }
    void alloc(size_t n) {
        release();
        if (n == 0) n = 16;
        HIP_CHECK(hipMalloc(&p, n));
        bytes = n;
    }
    void ensure(size_t n) { if (n > bytes) alloc(n + n / 4); }
    template<class T> void upload(const std::vector<T>& v) {
        alloc(v.size() * sizeof(T));
        if (!v.empty()) HIP_CHECK(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    }
    template<class T> T* as() const { return static_cast<T*>(p); }
};

struct PinnedBuf {
    void* p = nullptr;
    size_t bytes = 0;
    PinnedBuf() {}
    PinnedBuf(const PinnedBuf&) = delete;
    PinnedBuf& operator=(const PinnedBuf&) = delete;
    ~PinnedBuf() { if (p) (void)hipHostFree(p); }
    void ensure(size_t n) {
        if (n <= bytes) return;
        if (p) { (void)hipHostFree(p); p = nullptr; }
        n += n / 4;
        HIP_CHECK(hipHostMalloc(&p, n, hipHostMallocDefault));
        bytes = n;
    }
    template<class T> T* as() const { return static_cast<T*>(p); }
};

struct DevIndex {
    DevBuf nodes, tables;
    ScgIndex view;
    void upload(const scg::HostIndex& h) {
        nodes.upload(h.nodes); tables.upload(h.tables);
        view.nodes = nodes.as<uint4>(); view.tables = tables.as<uint4>();
        view.wide = h.wide; view.slot_mask = h.slot_mask; view.n_entries = h.n_entries; view.len = h.len; view.nseg = h.nseg;
        for (int s = 0; s < SCG_MAX_SEGMENTS; ++s) view.segmask[s] = h.segmask[s];
        for (int c = 0; c < 4; ++c) view.nwalk[c] = h.nwalk[c];
    }
};

struct DevPairTable {
    DevBuf keys, vals, l1, l2, lv;
    ScgPairTable view;
    void upload(const scg::HostPairTable& h) {
        keys.upload(h.keys); vals.upload(h.vals); l1.upload(h.list_key1); l2.upload(h.list_key2); lv.upload(h.list_vals);
        view.keys = keys.as<uint64_t>(); view.vals = vals.as<int32_t>(); view.mask = h.mask; view.n_entries = h.n_entries;
        view.list_key1 = l1.as<uint64_t>(); view.list_key2 = l2.as<uint64_t>(); view.list_vals = lv.as<int32_t>();
    }
};

} // namespace scgapi

using namespace scgapi;

// -------------------------------------------------------------------------------------------------
// Plans
// -------------------------------------------------------------------------------------------------
struct scg_plan {
    enum Kind { SINGLE, COMBO, DUAL, DUAL_SE_DIAG } kind = SINGLE;   // DUAL_SE_DIAG: single-end dual barcodes, include.invalid=TRUE
    int device = 0;

    // host-compiled pieces (valid before any device work)
    scg::HostTemplate ht1, ht2;
    ScgScan scan1, scan2;
    scg::HostIndex htab[2];
    scg::HostPairTable hpairs;
    int32_t n_pool[2] = {0, 0};
    int max_mm1 = 0, max_mm2 = 0;
    bool rev1 = false, rev2 = false, randomized = false, use_first = true;
    int diagnostics = 0;         // 0 none, 1 include.invalid=TRUE, 2 paired combinations (ScgDualParams::diagnostics)
    std::vector<int32_t> first1, first2;   // sequence uid -> first pool index (DuplicateAction::FIRST)

    // device state
    DevBuf d_tmpl1, d_tmpl2;
    DevIndex tab[2];
    DevPairTable pairs;
    DevBuf own_counters;
    DevBuf hot;          // diagnostics plans: per-wavefront slots of the two single-address tallies (ScgCounters::hot)
    DevBuf replicas;     // privatised counter copies (ScgCounters); empty when n_counters is large
    scg::HostIndex htab_combined;   // DUAL_SE_DIAG: wide index of the concatenated combinations (tab[0..1] = the per-region pools)
    DevIndex tab_combined;
    std::map<hipStream_t, DevBuf> overflow;     // pair search: the batch's pairs left to the byte-wise search (ScgDualParams::overflow)
    std::map<hipStream_t, DevBuf> unit_index;   // tally mode: barcode index per read of the batch in flight on each stream (ScgCounters::unit_index)
    // Sparse mode: the combination space (n0 x n1 pools, or the invalid pairs of include.invalid=TRUE) is beyond the dense
    // limit, so combinations travel as a stream of 64-bit keys per batch (ScgCounters::unit_pair), are sorted and
    // run-length encoded on the device (scg_sparse.hip) and merged here -- the reference's own algorithm
    // (kaori/utils.hpp:173-198, src/utils.h:14-45).
    bool sparse = false;
    struct PairStream {
        DevBuf keys, sorted, unique, counts, runs, scratch;
        int64_t pending = 0;                    // reads of the batch whose runs have not been merged yet
        hipEvent_t done = nullptr;              // behind the batch's sort + run-length encode (the stream may be gone when the runs are read)
        PairStream() {}
        PairStream(const PairStream&) = delete;
        PairStream& operator=(const PairStream&) = delete;
        ~PairStream() { if (done) (void)hipEventDestroy(done); }
    };
    std::map<hipStream_t, PairStream> pair_stream;
    std::unordered_map<uint64_t, int64_t> sparse_counts;
    int replica_shift = 0;   // log2(replicas)
    DevBuf error_flag;   // set by a staged kernel that met a read longer than the declared maximum
    int32_t* counters = nullptr;
    int64_t n_counters = 0;
    int64_t total = 0;

    bool profiling = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t> > events;
    size_t events_used = 0;

    ~scg_plan() {
        for (auto& e : events) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
    }

    void to_device(int dev) {
        device = resolve_device(dev);
        DeviceGuard g(device);
        std::vector<ScgTemplate> t1(1, ht1.t);
        d_tmpl1.upload(t1);
        if (kind == DUAL) {
            std::vector<ScgTemplate> t2(1, ht2.t);
            d_tmpl2.upload(t2);
        }
        tab[0].upload(htab[0]);
        if (kind != SINGLE) tab[1].upload(htab[1]);
        if (kind == DUAL_SE_DIAG) { tab_combined.upload(htab_combined); htab_combined = scg::HostIndex(); }
        if (kind == DUAL) pairs.upload(hpairs);
        own_counters.alloc(static_cast<size_t>(std::max<int64_t>(n_counters, 1)) * sizeof(int32_t));   // (plans without counters: random barcodes)
        counters = own_counters.as<int32_t>();
        HIP_CHECK(hipMemset(counters, 0, static_cast<size_t>(std::max<int64_t>(n_counters, 1)) * sizeof(int32_t)));
        // enough replicas that ~2^20 distinct addresses take the atomics
        replica_shift = 0;
        int addr_log2 = 20;
        if (const char* e = std::getenv("SCG_REPLICA_ADDR_LOG2")) addr_log2 = std::atoi(e);     // tuning aid
        while (n_counters > 0 && replica_shift < 12 && (n_counters << (replica_shift + 1)) <= (int64_t(1) << addr_log2)) ++replica_shift;
        if (replica_shift > 0) {
            replicas.alloc((static_cast<size_t>(n_counters) << replica_shift) * sizeof(int32_t));
            HIP_CHECK(hipMemset(replicas.p, 0, replicas.bytes));
        }
        if (kind == DUAL && diagnostics) {
            hot.alloc(2 * SCG_HOT_SLOTS * sizeof(int32_t));
            HIP_CHECK(hipMemset(hot.p, 0, hot.bytes));
        }
        error_flag.alloc(sizeof(int32_t));
        HIP_CHECK(hipMemset(error_flag.p, 0, sizeof(int32_t)));
        // hipMemset of device memory returns when the fill is ENQUEUED on the null stream, and the streams the kernels run on
        // are non-blocking ones, which the null stream does not hold back: a counting kernel launched right behind this
        // function (the first window of a file is inflated and scanned by the time the plan exists) added to counters and
        // replicas that the fill then cleared -- every counter a few reads short, once in 30-70 calls of the bench's BGZF leg
        // (tools/repro_bgzf_counts.py)
        HIP_CHECK(hipStreamSynchronize(nullptr));
        // host copies are no longer needed
        for (auto& h : htab) { h = scg::HostIndex(); }
        hpairs = scg::HostPairTable();
    }

    struct Timer {
        scg_plan* plan; hipStream_t stream; size_t slot = 0; bool on;
        Timer(scg_plan* p, hipStream_t s) : plan(p), stream(s), on(p->profiling) {
            if (!on) return;
            if (plan->events_used == plan->events.size()) {
                hipEvent_t a, b;
                HIP_CHECK(hipEventCreate(&a));
                HIP_CHECK(hipEventCreate(&b));
                plan->events.emplace_back(a, b);
            }
            slot = plan->events_used++;
            HIP_CHECK(hipEventRecord(plan->events[slot].first, stream));
        }
        void stop() { if (on) HIP_CHECK(hipEventRecord(plan->events[slot].second, stream)); }
    };
};


namespace scgapi {

typedef std::function<std::unique_ptr<scg_plan>()> Compile;

// ---- scg_plan.cpp ----
bool general_only(const scg_plan* P);
int64_t dense_cells();
ScgReads make_reads(const char* d_seqs, const uint32_t* d_offsets, int32_t fixed_len, int32_t max_len);
void check_reads_args(const char* d_seqs, const uint32_t* d_offsets, int32_t fixed_len, int64_t n);
std::unique_ptr<scg_plan> compile_single(const char* constant, int strand, const char* const* pool, int32_t n_pool, int mismatches, int use_first);
std::unique_ptr<scg_plan> compile_dual_single_end(const char* constant, int strand, const char* const* const* pools, const int32_t* n_pools, int32_t n_regions,
                                                  int mismatches, int use_first);
std::unique_ptr<scg_plan> compile_combo(const char* constant, int strand, const char* const* pool0, int32_t n0, const char* const* pool1, int32_t n1, int mismatches,
                                        int use_first);
std::unique_ptr<scg_plan> compile_dual(const char* constant1, int reverse1, int mismatches1, const char* const* pool1, const char* constant2, int reverse2,
                                       int mismatches2, const char* const* pool2, int32_t n_pool, int randomized, int use_first, int diagnostics = 0);
std::unique_ptr<scg_plan> compile_dual_single_end_diag(const char* constant, int strand, const char* const* const* pools, const int32_t* n_pools, int32_t n_regions,
                                                       int mismatches, int use_first);
std::unique_ptr<scg_plan> compile_paired_combo(const char* constant1, int reverse1, int mismatches1, const char* const* pool1, int32_t n1, const char* constant2,
                                               int reverse2, int mismatches2, const char* const* pool2, int32_t n2, int randomized, int use_first);
void retire_all_pairs(scg_plan* P);
void launch_batch(scg_plan* P, const ScgReads& R, int64_t n, hipStream_t stream);
void launch_batch_paired(scg_plan* P, const ScgReads& R1, const ScgReads& R2, int64_t n, hipStream_t stream);

// ---- scg_pipelines.cpp ----
// ---- FASTQ -> device staging: two slots, each with its own stream, pinned and device buffers ----
struct Stager {
    static const int SLOTS = 2;
    struct Slot {
        hipStream_t stream = nullptr;
        PinnedBuf h_seqs[2], h_offs[2], h_aux;
        DevBuf d_seqs[2], d_offs[2], d_aux;
        int64_t n_reads = 0;         // reads of the batch in flight (for `retire`)
        bool busy = false;
    } slot[SLOTS];
    int next = 0;
    // Called with a slot whose stream has just been synchronised, before its buffers are reused:
    // pipelines that bring per-read results back to the host consume them here.
    std::function<void(Slot&)> retire;

    Stager() {
        for (auto& s : slot) HIP_CHECK(hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking));
    }
    ~Stager() {
        for (auto& s : slot) if (s.stream) { (void)hipStreamSynchronize(s.stream); (void)hipStreamDestroy(s.stream); }
    }

    Slot& acquire() {
        Slot& s = slot[next];
        next = (next + 1) % SLOTS;
        if (s.busy) { HIP_CHECK(hipStreamSynchronize(s.stream)); s.busy = false; if (retire) retire(s); }
        return s;
    }

    // Copies one host batch into lane `which` of the slot; returns the device view.
    ScgReads stage(Slot& s, int which, const scg::ReadBatch& b) {
        size_t nbytes = b.seqs.size();
        size_t n = static_cast<size_t>(b.size());
        if (nbytes >= (size_t(1) << 32)) throw Error(SCG_ERR_INVALID, "internal: batch exceeds 4 GiB");
        s.h_seqs[which].ensure(nbytes + 16);
        s.h_offs[which].ensure((n + 1) * sizeof(uint32_t));
        s.d_seqs[which].ensure(nbytes + 16);
        s.d_offs[which].ensure((n + 1) * sizeof(uint32_t));
        if (nbytes) std::memcpy(s.h_seqs[which].p, b.seqs.data(), nbytes);
        uint32_t* ho = s.h_offs[which].as<uint32_t>();
        uint64_t max_len = 0;
        ho[0] = static_cast<uint32_t>(b.offsets[0]);
        for (size_t i = 1; i <= n; ++i) {
            ho[i] = static_cast<uint32_t>(b.offsets[i]);
            max_len = std::max<uint64_t>(max_len, b.offsets[i] - b.offsets[i - 1]);
        }
        if (nbytes) HIP_CHECK(hipMemcpyAsync(s.d_seqs[which].p, s.h_seqs[which].p, nbytes, hipMemcpyHostToDevice, s.stream));
        HIP_CHECK(hipMemcpyAsync(s.d_offs[which].p, ho, (n + 1) * sizeof(uint32_t), hipMemcpyHostToDevice, s.stream));
        return make_reads(s.d_seqs[which].as<char>(), s.d_offs[which].as<uint32_t>(), 0,
                          static_cast<int32_t>(std::min<uint64_t>(max_len, 1u << 30)));
    }

    void drain() {
        // oldest batch first, so that `retire` sees the batches in file order
        for (int k = 0; k < SLOTS; ++k) {
            Slot& s = slot[(next + k) % SLOTS];
            HIP_CHECK(hipStreamSynchronize(s.stream));
            if (s.busy && retire) { s.busy = false; retire(s); }
            s.busy = false;
        }
    }
};

const int64_t BATCH_READS = int64_t(1) << 22;
const int64_t BATCH_BYTES = int64_t(1) << 30;

// One compiled plan on each of `devices` (the reference's counterpart: one handler state per worker thread,
// process_data.hpp:126-147; the per-device results are summed like its serial reduce(), :115-124).
struct PlanSet {
    std::vector<std::unique_ptr<scg_plan> > plans;

    PlanSet(std::unique_ptr<scg_plan> compiled, const std::vector<int>& devices);
    std::vector<scg_plan*> all() const;
    scg_plan* first() const { return plans[0].get(); }
    int64_t total() const;
    // Sum of the devices' counters (every count is bounded by the total, which the callers check against int32).
    void read(int32_t* counts_out) const;
    void reset() const;
    // Sparse mode: the combinations of all devices (every batch's runs merged).
    std::unordered_map<uint64_t, int64_t> sparse_merged() const;
};

void count_single_end_file(scg_plan* P, const char* path, scg::FastqStream& fq, int nthreads,
                           const std::function<void(Stager::Slot&, const ScgReads&, int64_t)>& launch = nullptr,
                           const std::function<void(Stager::Slot&)>& retire = nullptr,
                           const std::function<void()>& restart = nullptr);
void release_cached_slots();
void reset_plan(scg_plan* P);
bool is_parallel_gzip(const scg::TextSource* s);
void count_single_end(const std::vector<scg_plan*>& plans, const char* path, scg::FastqStream& fq, int nthreads);
int32_t narrow_total(int64_t total);
void read_counters(scg_plan* P, int32_t* counts_out);
std::vector<int> device_list(bool* explicit_list = nullptr);
void set_thread_devices(const int* devices, int32_t n);      // scg_set_devices()
std::unique_ptr<scg_plan> clone_compiled(const scg_plan& a);
void schedule_files(int32_t n_files, const PlanSet& set, const std::function<void(scg_plan*, int32_t)>& per_file);
std::unique_ptr<PlanSet> compile_and_count_single_end(const char* path, scg::FastqStream& fq, int nthreads, Compile compile);
void combo_compact(const int32_t* cells, int32_t n0, int32_t n1, int32_t** indices_out, int32_t** freq_out, int64_t* k_out);
void combos_from_sparse(const std::unordered_map<uint64_t, int64_t>& m, int32_t** indices_out, int32_t** freq_out, int64_t* k_out);
void diagnostics_from_counters(const scg_plan* P, const std::vector<int32_t>& all, int32_t* counts_out, int32_t** idx_out, int32_t** freq_out, int64_t* k_out,
                               int32_t* b1, int32_t* b2, const std::unordered_map<uint64_t, int64_t>* sparse = nullptr);
void count_paired_files(scg_plan* P, const char* path1, const char* path2, scg::FastqStream& fq1, scg::FastqStream& fq2, int nthreads,
                        bool try_device_inflate = true, bool parallel_gzip = true);
std::unique_ptr<PlanSet> compile_and_count_paired(const char* path1, const char* path2, scg::FastqStream& fq1, scg::FastqStream& fq2, int nthreads, Compile compile);

} // namespace scgapi

#endif
