// scg_pipelines.cpp -- FASTQ files to counts: staging of parsed batches, the windowed pipelines (plain text scanned on
// the host or the device, BGZF members inflated on the device, paired files), devices and plan sets of a call.
//
// Host-side counterpart of the reference's chunked drivers (inst/include/kaori/process_data.hpp:105-190, :224-340):
// instead of handing 100 000-read chunks to std::threads, windows of the file go through pinned buffers into HBM on
// several HIP streams and are counted by the kernels of scg_kernels.hip.
#include "scg_internal.hpp"

namespace scgapi {


// FASTQ file -> counters for single-end plans.  Plain 4-line FASTQ is parsed by several host
// threads (ParallelFastq); gzip input, and any file the parallel reader finds unusual, goes
// through the sequential reader, which reproduces the reference's parse and errors exactly.
void count_single_end_file(scg_plan* P, const char* path, scg::FastqStream& fq, int nthreads,
                           const std::function<void(Stager::Slot&, const ScgReads&, int64_t)>& launch,
                           const std::function<void(Stager::Slot&)>& retire,
                           const std::function<void()>& restart) {
    Stager st;
    st.retire = retire;
    auto run = [&](Stager::Slot& s, const ScgReads& R, int64_t n) {
        s.n_reads = n;
        if (launch) launch(s, R, n); else launch_batch(P, R, n, s.stream);
    };
    const int threads = scg::default_host_threads(nthreads);
    if (threads > 1 && scg::ParallelFastq::is_plain_file(path)) {
        scg::ParallelFastq pf(path, threads);
        // window k + 1 is parsed by the workers while window k is copied to the device and counted
        std::vector<scg::ReadBatch> window, ahead;
        bool have = pf.next_window(window);
        while (have) {
            bool have_next = false;
            std::thread prefetch([&] { have_next = pf.next_window(ahead); });
            try {
                for (auto& b : window) {
                    if (b.size() == 0) continue;
                    auto& s = st.acquire();
                    ScgReads R = st.stage(s, 0, b);
                    run(s, R, b.size());
                    s.busy = true;
                }
            } catch (...) {
                prefetch.join();
                throw;
            }
            prefetch.join();
            window.swap(ahead);
            have = have_next;
        }
        st.drain();
        if (!pf.unusual()) return;
        // start over with the reference-exact sequential reader
        if (P->n_counters) HIP_CHECK(hipMemset(P->counters, 0, static_cast<size_t>(P->n_counters) * sizeof(int32_t)));
        HIP_CHECK(hipStreamSynchronize(nullptr));           // (the fill is only enqueued: scg_plan::upload)
        P->total = 0;
        if (restart) restart();
    }
    scg::ReadBatch b;
    while (fq.next_batch(b, BATCH_READS, BATCH_BYTES)) {
        auto& s = st.acquire();
        ScgReads R = st.stage(s, 0, b);
        run(s, R, b.size());
        s.busy = true;
    }
    st.drain();
}

// -------------------------------------------------------------------------------------------------
// Device-scan pipeline (single-end): raw FASTQ text -> pinned window -> HBM -> record scan -> counting kernels.
//
// The host moves bytes only (scg_ingest.cpp: file pages or inflated gzip blocks, cut at record boundaries); the
// records are found and validated on the GPU (scg_textscan.hip), so the text crosses PCIe once and no host thread
// parses it.  Windows go round-robin over the plans (one per device), each device working through a few slots with
// their own streams: while window k is copied and scanned, the host fills window k + 1 and the counting kernels of
// window k - 1 run.  Replaces kaori::process_single_end_data (process_data.hpp:105-190).  Anything the scan reports
// as out of the ordinary raises UnusualInput and the caller redoes the file with the sequential reader.
// -------------------------------------------------------------------------------------------------
struct UnusualInput {};

struct ScanSlot {
    scg_plan* plan = nullptr;
    int plan_device = -1;
    hipStream_t stream = nullptr;
    PinnedBuf text, h_result, h_offsets;
    DevBuf d_text, d_counts, d_nl, d_offsets, d_seqs, d_result, d_scan;
    scg::TextScanBuffers B;
    size_t cap = 0;
    bool pending = false;      // scan enqueued; the counting kernels still have to be launched
    bool parsed = false;       // the host did the record scan of the pending window: `host_result` holds its outcome
    scg::TextScanResult host_result{};
    bool busy = false;         // work of an earlier window may still be running on the stream

    // device-side inflate (InflatePipeline) only:
    DevBuf d_in, d_status;     // compressed members + their table; failure flags of the inflate / carry kernels
    PinnedBuf h_status;
    hipEvent_t scanned = nullptr, carried = nullptr;
    size_t pinned_cap = 0;     // bytes of `text` (the pinned staging buffer): the window, or less when only compressed bytes pass through
    uint32_t text_bytes = 0;   // text in d_text for the pending window
    bool last = false;         // the pending window is the input's last

    void init(int device, size_t window, size_t pinned_bytes) {
        plan_device = device;
        cap = window;
        pinned_cap = pinned_bytes;
        DeviceGuard g(device);
        HIP_CHECK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
        text.ensure(pinned_cap);
        h_result.ensure(sizeof(scg::TextScanResult));
        B.cap_blocks = scg::text_scan_blocks(cap) + 1;
        B.cap_lines = cap / 16 + 1024;              // lines shorter than 16 bytes on average: left to the sequential reader
        B.cap_records = B.cap_lines / 4 + 1;
        B.cap_seq_bytes = cap / 2 + 64;
        h_offsets.ensure((B.cap_records + 1) * sizeof(uint32_t));
        d_text.alloc(scg::text_scan_padded(cap) + 16);
        d_counts.alloc(B.cap_blocks * sizeof(uint32_t));
        d_nl.alloc(B.cap_lines * sizeof(uint32_t));
        d_offsets.alloc((B.cap_records + 1) * sizeof(uint32_t));
        d_seqs.alloc(B.cap_seq_bytes + 64);
        d_result.alloc(sizeof(scg::TextScanResult));
        d_scan.alloc(scg::text_scan_scratch(B.cap_blocks, B.cap_records) * sizeof(uint32_t));
        B.scan_scratch = d_scan.as<uint32_t>();
        B.block_counts = d_counts.as<uint32_t>();
        B.nl = d_nl.as<uint32_t>();
        B.offsets = d_offsets.as<uint32_t>();
        B.seqs = d_seqs.as<char>();
        B.result = d_result.as<scg::TextScanResult>();
    }
    // The extras of the inflate pipeline, on first use.
    void ensure_inflate() {
        if (scanned) return;
        DeviceGuard g(plan_device);
        d_in.alloc(pinned_cap);
        d_status.alloc(sizeof(uint32_t));
        h_status.ensure(sizeof(uint32_t));
        HIP_CHECK(hipEventCreateWithFlags(&scanned, hipEventDisableTiming));
        HIP_CHECK(hipEventCreateWithFlags(&carried, hipEventDisableTiming));
    }
    ~ScanSlot() {
        if (stream) {
            int prev = -1;
            if (plan_device >= 0 && hipGetDevice(&prev) == hipSuccess && prev != plan_device) (void)hipSetDevice(plan_device); else prev = -1;
            (void)hipStreamSynchronize(stream);
            if (scanned) (void)hipEventDestroy(scanned);
            if (carried) (void)hipEventDestroy(carried);
            (void)hipStreamDestroy(stream);
            if (prev >= 0) (void)hipSetDevice(prev);
        }
    }
};

// The sequences and offsets the host threads found in a window (pinned, in segments) go to the slot's HBM buffers,
// back to back: one kernel pulls them over the link.  Returns the number of records.
uint32_t enqueue_gather(ScanSlot& s, const scg::ParsedWindow& w) {
    if (w.seq_bytes > s.B.cap_seq_bytes || w.seq_bytes > 0xFFFFFFFFull || w.n_records > s.B.cap_records) throw UnusualInput();
    scg::GatherSegments G;
    G.n = static_cast<uint32_t>(w.n_segs);
    uint32_t rec = 0, at = 0;
    for (int i = 0; i < w.n_segs; ++i) {
        const scg::ParsedSegment& g = w.seg[i];
        G.seq_src[i] = s.text.as<char>() + g.seq_at;
        G.off_src[i] = s.h_offsets.as<uint32_t>() + g.off_at;
        G.seq_at[i] = at;
        G.first[i] = rec;
        G.off_base[i] = 0;
        rec += g.n_records;
        at += g.seq_bytes;
    }
    G.seq_at[G.n] = at;
    G.first[G.n] = rec;
    HIP_CHECK(scg::launch_gather_segments(s.B.seqs, s.B.offsets, G, s.stream));
    return rec;
}

size_t scan_window_bytes(uint64_t hint) {
    size_t w = size_t(128) << 20;
    if (const char* e = std::getenv("SCG_WINDOW_KB")) {          // test hook: tiny windows force many hand-overs
        const long kb = std::atol(e);
        if (kb > 0) w = static_cast<size_t>(kb) << 10;
    }
    const size_t floor = std::getenv("SCG_WINDOW_KB") ? size_t(4) << 10 : scg::TextSource::min_capacity();
    const uint64_t need = hint + (hint >> 4) + 4096;             // the whole input in one window when it is small
    if (need < w) w = static_cast<size_t>(need);
    return std::max(w, floor);
}

// Idle scan slots are kept for the next call (pinning and unpinning 3 x 128 MB of host memory costs ~120 ms, a third
// of the time a 10 GB file takes): at most four per device (the paired pipeline uses four), released by scg_release_buffers() or with the process.
struct SlotPool {
    std::mutex mu;
    std::vector<std::unique_ptr<ScanSlot> > idle;
    // pinned_bytes = 0: as much pinned staging as text (the raw-text and host-scan pipelines)
    std::unique_ptr<ScanSlot> take(int device, size_t window, size_t pinned_bytes = 0) {
        const bool whole = pinned_bytes == 0;          // these pipelines fill the pinned buffer up to the slot's capacity
        if (whole) pinned_bytes = window;
        {
            std::lock_guard<std::mutex> g(mu);
            for (size_t i = 0; i < idle.size(); ++i) {
                if (idle[i]->plan_device == device && idle[i]->cap >= window && idle[i]->cap <= 2 * window + (size_t(8) << 20) &&
                    idle[i]->pinned_cap >= (whole ? idle[i]->cap : pinned_bytes)) {
                    std::unique_ptr<ScanSlot> s = std::move(idle[i]);
                    idle.erase(idle.begin() + static_cast<long>(i));
                    s->plan = nullptr;
                    return s;
                }
            }
        }
        std::unique_ptr<ScanSlot> s(new ScanSlot);
        s->init(device, window, pinned_bytes);
        return s;
    }
    void give(std::unique_ptr<ScanSlot> s) {
        const char* e = std::getenv("SCG_BUFFER_CACHE");
        if (e && *e == '0') return;
        s->plan = nullptr;
        std::lock_guard<std::mutex> g(mu);
        // At most 4 idle slots per device and size class (slots that could serve one another's windows), 8 per device:
        // a process that alternates between input forms -- plain files, then BGZF -- keeps both kinds instead of
        // allocating 2.5 GB anew on every call of the second kind (25 ms per call, measured in bench.py's BGZF leg).
        int same = 0, on_device = 0;
        for (auto& x : idle) {
            if (x->plan_device != s->plan_device) continue;
            ++on_device;
            same += 2 * x->cap <= 3 * s->cap && 2 * s->cap <= 3 * x->cap;       // (within a factor of 1.5: 128 MB text windows and 257 MB inflate windows are two classes)
        }
        if (same < 4 && on_device < 8) idle.push_back(std::move(s));
    }
    void clear() {
        std::lock_guard<std::mutex> g(mu);
        idle.clear();
    }
};

SlotPool& slot_pool() {
    static SlotPool* pool = new SlotPool;      // deliberately never destroyed: the HIP runtime may be gone by then
    return *pool;
}

bool host_scan_enabled() {
    const char* e = std::getenv("SCG_HOST_SCAN");            // test hook: 0 ships the raw text of plain files too
    return !(e && *e == '0');
}

// The single-end pipeline.  Slots belong to devices, not to plans, so the first window can be read, copied and
// scanned (start) while the template and the library are still being compiled on another thread; run() then binds
// device d's slots to plans[d] and carries on.
class TextPipeline {
public:
    TextPipeline(scg::TextSource& source, const std::vector<int>& devs)
        : src(source), devices(devs), window(scan_window_bytes(source.size_hint())), host_scan(source.parses() && host_scan_enabled()) {
        const int slots_per_device = 3;
        for (int k = 0; k < slots_per_device; ++k) {
            for (int d : devices) slots.push_back(slot_pool().take(d, window));
        }
        tr.mark("  scan slots (pinned + HBM)");
    }
    ~TextPipeline() {
        if (ok) for (auto& s : slots) if (s) slot_pool().give(std::move(s));
    }

    // Before the plans exist (the library is still being compiled on another thread): every slot takes a window -- parse
    // or copy, the link and the device's record scan need no plan, only the counting does.
    void start() {
        for (size_t k = 0; k < slots.size() && !ended && filled == k; ++k) fill_next();
    }

    void run(const std::vector<scg_plan*>& plans) {
        for (size_t i = 0; i < slots.size(); ++i) slots[i]->plan = plans[i % plans.size()];
        // Window k is filled and put on the wire; the counting kernels of window k - lag are launched afterwards, by
        // which time its copy and scan have normally finished: the host thread does not wait on the link.
        // (Windows scanned by the host need no such wait.)
        const size_t lag = host_scan ? 0 : devices.size() * 2 < slots.size() ? devices.size() * 2 : slots.size() - 1;
        while (finished + lag < filled) finish_next();              // (the windows start() has taken while there was no plan)
        while (!ended) {
            fill_next();
            if (filled > lag && finished < filled - lag) finish_next();
        }
        while (finished < filled) finish_next();
        for (auto& s : slots) {
            DeviceGuard g(s->plan_device);
            HIP_CHECK(hipStreamSynchronize(s->stream));
            s->busy = false;
        }
        ok = true;
        if (tr.on) {
            std::fprintf(stderr, "[scg]   windows of %zu MB (%s scan): host fill %.2f ms, waiting for scans %.2f ms, for free slots %.2f ms\n", window >> 20,
                         host_scan ? "host" : "device", t_fill, t_finish, t_busy);
        }
        tr.mark("  windows");
    }

private:
    scg::TextSource& src;
    std::vector<int> devices;
    size_t window;
    bool host_scan;
    std::vector<std::unique_ptr<ScanSlot> > slots;
    size_t filled = 0, finished = 0;     // windows put on the wire / windows whose counting kernels have been launched
    bool ended = false, ok = false;
    Trace tr;
    double t_fill = 0, t_finish = 0, t_busy = 0;

    void fill_next() {
        ScanSlot& s = *slots[filled % slots.size()];
        DeviceGuard g(s.plan_device);
        if (s.pending) finish_next();                              // (only when there are fewer slots than the lag needs)
        const auto b0 = std::chrono::steady_clock::now();
        if (s.busy) { HIP_CHECK(hipStreamSynchronize(s.stream)); s.busy = false; }
        const auto f0 = std::chrono::steady_clock::now();
        t_busy += std::chrono::duration<double, std::milli>(f0 - b0).count();
        scg::ParsedWindow w;
        const bool on_device = src.device_resident() && src.device() == s.plan_device;       // text that lies in this device's HBM already
        const size_t bytes = on_device ? src.next_device(s.d_text.as<char>(), s.cap, s.stream)
                           : host_scan ? src.next_parsed(s.text.as<char>(), s.cap, s.h_offsets.as<uint32_t>(), s.B.cap_records + 1, w)
                                       : src.next(s.text.as<char>(), s.cap);
        t_fill += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - f0).count();
        if (src.unusual()) throw UnusualInput();
        if (bytes == 0) { ended = true; return; }
        if (on_device) {
            s.parsed = false;
            HIP_CHECK(scg::launch_text_scan(s.d_text.as<char>(), bytes, s.B, s.stream));
            HIP_CHECK(hipMemcpyAsync(s.h_result.p, s.d_result.p, sizeof(scg::TextScanResult), hipMemcpyDeviceToHost, s.stream));
            s.pending = true;
            ++filled;
            return;
        }
        if (host_scan) {
            const uint32_t rec = enqueue_gather(s, w);
            s.host_result = scg::TextScanResult{0, rec, w.max_len, 0, w.seq_bytes, 0, 0};
            s.parsed = true;
            s.pending = true;
            ++filled;
            return;
        }
        s.parsed = false;
        HIP_CHECK(hipMemcpyAsync(s.d_text.p, s.text.p, bytes, hipMemcpyHostToDevice, s.stream));
        HIP_CHECK(scg::launch_text_scan(s.d_text.as<char>(), bytes, s.B, s.stream));
        HIP_CHECK(hipMemcpyAsync(s.h_result.p, s.d_result.p, sizeof(scg::TextScanResult), hipMemcpyDeviceToHost, s.stream));
        s.pending = true;
        ++filled;
    }

    void finish_next() {
        ScanSlot& s = *slots[finished % slots.size()];
        const auto f1 = std::chrono::steady_clock::now();
        DeviceGuard g(s.plan_device);
        if (!s.parsed) HIP_CHECK(hipStreamSynchronize(s.stream)); // copy + scan + result are in
        t_finish += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - f1).count();
        s.pending = false;
        ++finished;
        const scg::TextScanResult r = s.parsed ? s.host_result : *s.h_result.as<scg::TextScanResult>();
        if (r.flags) throw UnusualInput();
        if (r.n_records) {
            launch_batch(s.plan, make_reads(s.B.seqs, s.B.offsets, 0, static_cast<int32_t>(std::min<uint32_t>(r.max_len, 1u << 30))),
                         static_cast<int64_t>(r.n_records), s.stream);
        }
        s.busy = true;
    }
};

bool device_inflate_enabled() {
    const char* e = std::getenv("SCG_DEVICE_INFLATE");       // test hook: 0 inflates BGZF members on the host threads
    return !(e && *e == '0');
}
// Test hook SCG_DEVICE_INFLATE=2: a file the device inflater hands back is an error instead of a quiet second try on the
// host (so that a test on a well-formed file cannot pass on the fall-back).
[[noreturn]] void inflate_declined() {
    throw Error(SCG_ERR_UNSUPPORTED, "the device inflater handed the file back (SCG_DEVICE_INFLATE=2 forbids the fall-back)");
}
bool device_inflate_strict() {
    const char* e = std::getenv("SCG_DEVICE_INFLATE");
    return e && *e == '2';
}

// ---- BGZF members inflated on the device: what the single-end (InflatePipeline) and paired (PairedPipeline) paths share ----
constexpr size_t INFLATE_GAP = size_t(1) << 20;      // room in front of a window's text for the previous window's partial record

// Text capacity of a device-inflate window: 256 MB (4 000 members in flight), less for small inputs.
size_t inflate_window_text(uint64_t hint) {
    size_t w = size_t(256) << 20;
    if (const char* e = std::getenv("SCG_WINDOW_KB")) { const long kb = std::atol(e); if (kb > 0) w = static_cast<size_t>(kb) << 10; }
    const uint64_t need = hint + (hint >> 4) + 4096;
    if (need < w) w = static_cast<size_t>(need);
    return std::max<size_t>(w, std::getenv("SCG_WINDOW_KB") ? size_t(4) << 10 : scg::TextSource::min_capacity());
}
size_t inflate_window_staging(size_t cap_text) { return cap_text / 2 + (size_t(1) << 20); }     // compressed bytes + member table
size_t inflate_window_slot(size_t cap_text) { return INFLATE_GAP + cap_text + 64; }

// Takes the next members of `src` into slot `s` (staging = its pinned buffer) and enqueues, on the slot's stream:
// members + table -> HBM, inflate + CRC check behind the gap, then -- once `prev` (the window before, if any) has its
// record structure -- the carry of prev's partial record into the gap, the record scan, and the copies back of the
// scan result and the status word.  `reader` is the slot whose carry read this slot's previous text (its `carried`
// event is waited for before the text is overwritten).  Returns false at the end of the input.
bool enqueue_inflate_window(scg::TextSource& src, ScanSlot& s, const ScanSlot* prev, const ScanSlot& reader, size_t cap_text, size_t cap_in,
                            std::vector<scg::CompressedMember>& members, double* t_fill) {
    const auto f0 = std::chrono::steady_clock::now();
    // staging: [member table | payloads]; room for one member per 32 bytes of compressed input is never short
    const size_t slack = scg::inflate_input_slack();
    const size_t table_cap = (cap_in / 32 / sizeof(scg::InflateMember)) * sizeof(scg::InflateMember);
    char* const stage = s.text.as<char>();
    size_t text_bytes = 0;
    bool last = false;
    const size_t in_bytes = src.next_members(stage + table_cap, cap_in - table_cap, slack, cap_text, members, text_bytes, last);
    if (src.unusual()) throw UnusualInput();
    if (in_bytes == 0) return false;
    if (members.size() * sizeof(scg::InflateMember) > table_cap) throw UnusualInput();        // (members of < 32 bytes: not a real file)
    static_assert(sizeof(scg::InflateMember) == sizeof(scg::CompressedMember), "same layout");
    scg::InflateMember* table = reinterpret_cast<scg::InflateMember*>(stage);
    for (size_t i = 0; i < members.size(); ++i) {
        table[i].in_off = static_cast<uint32_t>(table_cap) + members[i].in_off;
        table[i].in_len = members[i].in_len;
        table[i].out_off = static_cast<uint32_t>(INFLATE_GAP) + members[i].out_off;
        table[i].out_len = members[i].out_len;
        table[i].crc = members[i].crc;
    }
    if (t_fill) *t_fill += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - f0).count();
    const uint32_t n = static_cast<uint32_t>(members.size());
    HIP_CHECK(hipStreamWaitEvent(s.stream, reader.carried, 0));
    HIP_CHECK(hipMemsetAsync(s.d_status.p, 0, sizeof(uint32_t), s.stream));
    HIP_CHECK(hipMemcpyAsync(s.d_in.p, stage, n * sizeof(scg::InflateMember), hipMemcpyHostToDevice, s.stream));
    HIP_CHECK(hipMemcpyAsync(s.d_in.as<char>() + table_cap, stage + table_cap, in_bytes, hipMemcpyHostToDevice, s.stream));
    HIP_CHECK(scg::launch_inflate_members(s.d_in.as<uint8_t>(), s.d_in.as<scg::InflateMember>(), n, s.d_text.as<char>(), s.d_status.as<uint32_t>(), s.stream));
    s.text_bytes = static_cast<uint32_t>(INFLATE_GAP + text_bytes);
    if (last) {
        // the reference accepts a final record without its newline: one is appended (a second one is harmless, see inflate_window_records)
        HIP_CHECK(hipMemsetAsync(s.d_text.as<char>() + s.text_bytes, '\n', 1, s.stream));
        s.text_bytes += 1;
    }
    s.last = last;
    if (prev) HIP_CHECK(hipStreamWaitEvent(s.stream, prev->scanned, 0));
    HIP_CHECK(scg::launch_carry_tail(prev ? prev->d_text.as<char>() : nullptr, prev ? prev->B.result : nullptr, prev ? prev->text_bytes : 0u,
                                     s.d_text.as<char>(), static_cast<uint32_t>(INFLATE_GAP), s.d_status.as<uint32_t>(), s.stream));
    HIP_CHECK(hipEventRecord(s.carried, s.stream));
    HIP_CHECK(scg::launch_text_scan(s.d_text.as<char>(), s.text_bytes, s.B, s.stream, true, s.scanned));
    HIP_CHECK(hipMemcpyAsync(s.h_result.p, s.d_result.p, sizeof(scg::TextScanResult), hipMemcpyDeviceToHost, s.stream));
    HIP_CHECK(hipMemcpyAsync(s.h_status.p, s.d_status.p, sizeof(uint32_t), hipMemcpyDeviceToHost, s.stream));
    return true;
}

// After the slot's stream has been synchronised: the window's records (record 0 is the gap's dummy: count from
// offsets + 1).  Throws UnusualInput for whatever the host paths have to redo: a member zlib has to look at (corrupt, or
// in a form the device's decoder declines), a record longer than the gap, anything but ordinary records.
scg::TextScanResult inflate_window_records(const ScanSlot& s) {
    if (*s.h_status.as<uint32_t>()) throw UnusualInput();
    const scg::TextScanResult r = *s.h_result.as<scg::TextScanResult>();
    if (r.flags || r.n_records == 0) throw UnusualInput();
    // behind the last whole record of the input: nothing, or the newline appended above
    if (s.last && s.text_bytes - r.cut > 1) throw UnusualInput();
    return r;
}

// Kernels of one device read another device's memory (the previous window's tail, when the windows of a BGZF file go
// round-robin over the devices of a call): peer access, once per ordered pair.  False if the hardware does not offer it.
bool enable_peer_access(const std::vector<int>& devices) {
    for (int a : devices) {
        for (int b : devices) {
            if (a == b) continue;
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, a, b) != hipSuccess || !can) return false;
            DeviceGuard g(a);
            const hipError_t e = hipDeviceEnablePeerAccess(b, 0);
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) { (void)hipGetLastError(); return false; }
            (void)hipGetLastError();
        }
    }
    return true;
}

// The single-end pipeline for BGZF input with the members inflated on the device (scg_inflate.hip).  Per window, on
// the slot's stream: compressed members + their table -> HBM; inflate + CRC check into the text buffer behind a gap of
// GAP bytes; then -- once the previous window has been scanned -- the gap receives that window's partial last record
// (launch_carry_tail), the text is scanned for records (allow_tail) and the result comes back.  The windows go round-robin
// over the devices of the call, three slots each: the inflate kernels -- the expensive part -- of all of them overlap;
// only the carry chains the windows, and where the previous window lies on another device the carry kernel reads its
// tail (<= 1 MB) and its scan result over xGMI (peer access; the streams wait on each other's events).
class InflatePipeline {
public:
    InflatePipeline(scg::TextSource& source, const std::vector<int>& devs) : src(source), devices(devs) {
        if (devices.size() > 1 && !enable_peer_access(devices)) devices.resize(1);
        cap_text = inflate_window_text(source.size_hint());
        window = inflate_window_slot(cap_text);
        cap_in = inflate_window_staging(cap_text);
        for (int k = 0; k < 3; ++k) {
            for (int d : devices) {
                slots.push_back(slot_pool().take(d, window, cap_in));
                slots.back()->ensure_inflate();
            }
        }
        tr.mark("  scan slots (pinned + HBM)");
    }
    ~InflatePipeline() {
        for (auto& s : slots) {
            if (!s) continue;
            DeviceGuard g(s->plan_device);
            (void)hipStreamSynchronize(s->stream);
            s->busy = false; s->pending = false;
            if (ok) slot_pool().give(std::move(s));
        }
    }
    size_t n_devices() const { return devices.size(); }

    // Before the plans exist (the library is still being compiled on another thread): every slot takes a window --
    // members over the link, inflate, CRC and record scan need no plan, only the counting does.
    void start() {
        for (size_t k = 0; k < slots.size() && !ended && filled == k; ++k) fill_next();
    }

    // plans[i] counts what device i of the list was given (fewer plans than devices: the list was cut down at construction)
    void run(const std::vector<scg_plan*>& plans) {
        for (size_t i = 0; i < slots.size(); ++i) slots[i]->plan = plans[(i % devices.size()) % plans.size()];
        const size_t lag = std::min(2 * devices.size(), slots.size() - 1);
        while (finished + lag < filled) finish_next();              // (the windows start() has taken while there was no plan)
        while (!ended) {
            fill_next();
            if (filled > lag && finished < filled - lag) finish_next();
        }
        while (finished < filled) finish_next();
        for (auto& s : slots) {
            DeviceGuard g(s->plan_device);
            HIP_CHECK(hipStreamSynchronize(s->stream));
            s->busy = false;
        }
        ok = true;
        if (tr.on) {
            std::fprintf(stderr, "[scg]   windows of %zu MB (device inflate, %zu device(s)): host fill %.2f ms, waiting for the device %.2f ms, for free slots %.2f ms\n",
                         cap_text >> 20, devices.size(), t_fill, t_finish, t_busy);
        }
        tr.mark("  windows");
    }

private:
    scg::TextSource& src;
    std::vector<int> devices;
    size_t cap_text = 0, window = 0, cap_in = 0;
    std::vector<std::unique_ptr<ScanSlot> > slots;
    std::vector<scg::CompressedMember> members;
    size_t filled = 0, finished = 0;
    bool ended = false, ok = false;
    Trace tr;
    double t_fill = 0, t_finish = 0, t_busy = 0;

    void fill_next() {
        ScanSlot& s = *slots[filled % slots.size()];
        DeviceGuard g(s.plan_device);
        if (s.pending) finish_next();
        const auto b0 = std::chrono::steady_clock::now();
        if (s.busy) { HIP_CHECK(hipStreamSynchronize(s.stream)); s.busy = false; }
        t_busy += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - b0).count();
        const ScanSlot* prev = filled ? slots[(filled - 1) % slots.size()].get() : nullptr;
        const ScanSlot& next = *slots[(filled + 1) % slots.size()];  // the window after this slot's previous one read its tail from here
        if (!enqueue_inflate_window(src, s, prev, next, cap_text, cap_in, members, &t_fill)) { ended = true; return; }
        const bool last = s.last;
        s.parsed = false;
        s.pending = true;
        ++filled;
        if (last) ended = true;
    }

    void finish_next() {
        ScanSlot& s = *slots[finished % slots.size()];
        const auto f1 = std::chrono::steady_clock::now();
        DeviceGuard g(s.plan_device);
        HIP_CHECK(hipStreamSynchronize(s.stream));
        t_finish += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - f1).count();
        s.pending = false;
        ++finished;
        const scg::TextScanResult r = inflate_window_records(s);
        if (r.n_records > 1) {                                       // record 0 is the gap's dummy
            launch_batch(s.plan, make_reads(s.B.seqs, s.B.offsets + 1, 0, static_cast<int32_t>(std::min<uint32_t>(r.max_len, 1u << 30))),
                         static_cast<int64_t>(r.n_records - 1), s.stream);
        }
        s.busy = true;
    }
};

void count_text_stream(const std::vector<scg_plan*>& plans, scg::TextSource& src) {
    std::vector<int> devices;
    for (scg_plan* P : plans) devices.push_back(P->device);
    TextPipeline pipe(src, devices);
    pipe.run(plans);
}

// -------------------------------------------------------------------------------------------------
// The paired-end pipeline.  Each file is taken in windows like single-end input -- plain files scanned for records by
// the host threads, compressed ones shipped as text and scanned on the device -- and the two streams of sequences are
// brought into step on the device without moving them again: windows of the two files hold different numbers of
// records, so each mate keeps a cursor into its current window (sequences + offsets in HBM); the kernels count
// min(remaining, remaining) pairs from the two cursors, and the mate whose window is used up takes its next one.
// Replaces kaori::process_paired_end_data (process_data.hpp:224-340).  One device: pair i needs read i of both files.
// -------------------------------------------------------------------------------------------------
struct MateWindows {
    scg::TextSource* src = nullptr;
    std::unique_ptr<ScanSlot> slot[2];      // double buffer: the host fills one while the kernels read the other
    hipEvent_t used[2] = {nullptr, nullptr};// the last kernel reading slot k has been enqueued before this event
    hipEvent_t ready = nullptr;             // the current window has arrived in HBM
    bool host_scan = false, done = false, fresh = false;
    bool inflate = false;                   // BGZF mate, members inflated on the device (enqueue_inflate_window)
    bool any = false;                       // (inflate) a window has been taken before: its partial last record is carried on
    size_t cap_text = 0, cap_in = 0;        // (inflate) window sizes
    int cur = 1;
    uint32_t first = 0;                     // records of the current window start at offsets[first] (1 behind a gap's dummy record)
    uint32_t n = 0, k = 0, max_len = 0;     // records in the current window, of which k have been paired
    uint32_t remaining() const { return n - k; }
    ~MateWindows() {
        for (hipEvent_t e : used) if (e) (void)hipEventDestroy(e);
        if (ready) (void)hipEventDestroy(ready);
    }
};

class PairedPipeline {
public:
    // device_inflate: BGZF mates may have their members inflated on the device (false: by the host threads)
    PairedPipeline(int dev, scg::TextSource& src1, scg::TextSource& src2, bool device_inflate)
        : device(dev), window(std::max(scan_window_bytes(src1.size_hint()), scan_window_bytes(src2.size_hint()))) {
        DeviceGuard g(device);
        mate[0].src = &src1; mate[1].src = &src2;
        HIP_CHECK(hipStreamCreateWithFlags(&compute, hipStreamNonBlocking));
        for (auto& m : mate) {
            m.host_scan = m.src->parses() && host_scan_enabled();
            m.inflate = device_inflate && m.src->has_members() && device_inflate_enabled();
            if (m.inflate) {
                m.cap_text = inflate_window_text(m.src->size_hint());
                m.cap_in = inflate_window_staging(m.cap_text);
            }
            for (int k = 0; k < 2; ++k) {
                if (m.inflate) {
                    m.slot[k] = slot_pool().take(device, inflate_window_slot(m.cap_text), m.cap_in);
                    m.slot[k]->ensure_inflate();
                } else {
                    m.slot[k] = slot_pool().take(device, window);
                }
                HIP_CHECK(hipEventCreateWithFlags(&m.used[k], hipEventDisableTiming));
            }
            HIP_CHECK(hipEventCreateWithFlags(&m.ready, hipEventDisableTiming));
            any_inflate |= m.inflate;
        }
        tr.mark("  scan slots (pinned + HBM)");
    }
    bool inflates() const { return any_inflate; }
    ~PairedPipeline() {
        int prev = -1;
        if (hipGetDevice(&prev) == hipSuccess && prev != device) (void)hipSetDevice(device); else prev = -1;
        if (compute) { (void)hipStreamSynchronize(compute); (void)hipStreamDestroy(compute); }
        for (auto& m : mate) for (auto& s : m.slot) {
            if (!s) continue;
            (void)hipStreamSynchronize(s->stream);
            s->busy = false; s->pending = false;
            if (ok) slot_pool().give(std::move(s));
        }
        if (prev >= 0) (void)hipSetDevice(prev);
    }

    // The first window of each file on its way (no plan needed yet).
    void start() {
        DeviceGuard g(device);
        advance();
        advanced = true;
    }

    void run(scg_plan* P) {
        DeviceGuard g(device);
        for (;;) {
            if (advanced) advanced = false; else advance();
            for (auto& m : mate) {
                if (!m.fresh) continue;                               // device scan: the record count comes back from the card
                m.fresh = false;
                ScanSlot& s = *m.slot[m.cur];
                const auto w0 = std::chrono::steady_clock::now();
                HIP_CHECK(hipStreamSynchronize(s.stream));
                t_wait += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - w0).count();
                if (m.inflate) {
                    const scg::TextScanResult r = inflate_window_records(s);
                    m.n = r.n_records - 1;                            // record 0 is the gap's dummy
                    m.first = 1;
                    m.max_len = r.max_len;
                    continue;
                }
                const scg::TextScanResult r = *s.h_result.as<scg::TextScanResult>();
                if (r.flags) throw UnusualInput();
                m.n = r.n_records;
                m.max_len = r.max_len;
            }
            // one file is exhausted and fully paired while the other still holds reads (process_data.hpp:284-285)
            for (int i = 0; i < 2; ++i) {
                if (mate[i].done && mate[i].remaining() == 0 && mate[1 - i].remaining() > 0) {
                    throw Error(SCG_ERR_IO, "different number of reads in paired FASTQ files");
                }
            }
            if (mate[0].done && mate[1].done) break;
            const uint32_t np = std::min(mate[0].remaining(), mate[1].remaining());
            if (np == 0) continue;
            const int32_t max_len = static_cast<int32_t>(std::min<uint32_t>(std::max(mate[0].max_len, mate[1].max_len), 1u << 30));
            ScgReads R[2];
            for (int i = 0; i < 2; ++i) {
                MateWindows& m = mate[i];
                const ScanSlot& s = *m.slot[m.cur];
                HIP_CHECK(hipStreamWaitEvent(compute, m.ready, 0));
                R[i] = make_reads(s.B.seqs, s.B.offsets + m.first + m.k, 0, max_len);
            }
            launch_batch_paired(P, R[0], R[1], static_cast<int64_t>(np), compute);
            for (auto& m : mate) {
                m.k += np;
                HIP_CHECK(hipEventRecord(m.used[m.cur], compute));
            }
        }
        HIP_CHECK(hipStreamSynchronize(compute));
        ok = true;
        if (tr.on) std::fprintf(stderr, "[scg]   paired windows of %zu MB: host fill %.2f ms, waiting for the device %.2f ms\n", window >> 20, t_fill, t_wait);
        tr.mark("  windows");
    }

private:
    int device;
    size_t window;
    MateWindows mate[2];
    hipStream_t compute = nullptr;
    bool ok = false, advanced = false, any_inflate = false;
    std::vector<scg::CompressedMember> members;
    Trace tr;
    double t_fill = 0, t_wait = 0;

    // A mate whose window is used up takes its next one.
    void advance() {
        for (auto& m : mate) {
            m.fresh = false;
            if (m.done || m.remaining() > 0) continue;
            m.cur ^= 1;
            ScanSlot& s = *m.slot[m.cur];
            const auto w0 = std::chrono::steady_clock::now();
            HIP_CHECK(hipEventSynchronize(m.used[m.cur]));        // its previous content is no longer being read
            const auto f0 = std::chrono::steady_clock::now();
            t_wait += std::chrono::duration<double, std::milli>(f0 - w0).count();
            if (m.inflate) {
                // the window before lies in the mate's other slot: its partial last record is carried over on the device
                const ScanSlot& other = *m.slot[m.cur ^ 1];
                m.n = m.k = 0;
                m.first = 1;
                if (!enqueue_inflate_window(*m.src, s, m.any ? &other : nullptr, other, m.cap_text, m.cap_in, members, &t_fill)) {
                    m.done = true;
                    continue;
                }
                m.any = true;
                m.fresh = true;
                HIP_CHECK(hipEventRecord(m.ready, s.stream));
                continue;
            }
            scg::ParsedWindow w;
            const bool on_device = m.src->device_resident() && m.src->device() == device;        // (an ordinary gzip mate decoded by this device)
            const size_t bytes = on_device ? m.src->next_device(s.d_text.as<char>(), s.cap, s.stream)
                               : m.host_scan ? m.src->next_parsed(s.text.as<char>(), s.cap, s.h_offsets.as<uint32_t>(), s.B.cap_records + 1, w)
                                             : m.src->next(s.text.as<char>(), s.cap);
            t_fill += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - f0).count();
            if (m.src->unusual()) throw UnusualInput();
            m.n = m.k = 0;
            if (bytes == 0) { m.done = true; continue; }
            if (m.host_scan && !on_device) {
                m.n = enqueue_gather(s, w);
                m.max_len = w.max_len;
            } else {
                if (!on_device) HIP_CHECK(hipMemcpyAsync(s.d_text.p, s.text.p, bytes, hipMemcpyHostToDevice, s.stream));
                HIP_CHECK(scg::launch_text_scan(s.d_text.as<char>(), bytes, s.B, s.stream));
                HIP_CHECK(hipMemcpyAsync(s.h_result.p, s.d_result.p, sizeof(scg::TextScanResult), hipMemcpyDeviceToHost, s.stream));
                m.fresh = true;
            }
            HIP_CHECK(hipEventRecord(m.ready, s.stream));
        }
    }
};

// -------------------------------------------------------------------------------------------------
// Paired plain files over SEVERAL devices.  Pair i needs read i of both files, so the work is handed out by record
// index: the host threads scan each mate's file in windows as they do for one device (PlainSource::next_parsed: sequences
// and offsets in pinned memory) and keep a cursor into each mate's current window; a round takes the
// min(remaining, remaining) pairs the two cursors have in common, and ONE device -- round-robin -- gathers exactly those
// records of both mates over its own PCIe link and counts them.  Every record crosses a link once, the devices work on
// different rounds at the same time, the per-device counters are summed at the end (PlanSet::read).  Replaces
// kaori::process_paired_end_data (process_data.hpp:224-340) for calls with more than one device; compressed mates keep
// the one-device pipeline above (their text exists only in one device's memory).
// -------------------------------------------------------------------------------------------------
class PairedRounds {
public:
    PairedRounds(const std::vector<int>& devs, scg::TextSource& src1, scg::TextSource& src2)
        : devices(devs), window(std::max(scan_window_bytes(src1.size_hint()), scan_window_bytes(src2.size_hint()))) {
        mate[0].src = &src1; mate[1].src = &src2;
        cap_lines = window / 16 + 1024;
        cap_records = cap_lines / 4 + 1;
        cap_seq = window / 2 + 64;
        for (auto& m : mate) {
            for (auto& hw : m.win) {
                hw.text.ensure(window);
                hw.offs.ensure((cap_records + 1) * sizeof(uint32_t));
            }
        }
        for (size_t i = 0; i < devices.size() * 2; ++i) {
            rounds.emplace_back(new Round);
            Round& R = *rounds.back();
            R.device = devices[i / 2];
            DeviceGuard g(R.device);
            HIP_CHECK(hipStreamCreateWithFlags(&R.stream, hipStreamNonBlocking));
            HIP_CHECK(hipEventCreateWithFlags(&R.done, hipEventDisableTiming));
            for (int k = 0; k < 2; ++k) {
                R.seqs[k].alloc(cap_seq + 64);
                R.offs[k].alloc((cap_records + 1) * sizeof(uint32_t));
            }
        }
        tr.mark("  round buffers (pinned + HBM)");
    }
    ~PairedRounds() {
        for (auto& rp : rounds) {
            Round& R = *rp;
            int prev = -1;
            if (hipGetDevice(&prev) == hipSuccess && prev != R.device) (void)hipSetDevice(R.device); else prev = -1;
            if (R.stream) { (void)hipStreamSynchronize(R.stream); (void)hipStreamDestroy(R.stream); }
            if (R.done) (void)hipEventDestroy(R.done);
            R.seqs[0].release(); R.seqs[1].release(); R.offs[0].release(); R.offs[1].release();
            if (prev >= 0) (void)hipSetDevice(prev);
        }
    }

    // plans[d] belongs to devices[d]
    void run(const std::vector<scg_plan*>& plans) {
        const size_t D = devices.size();
        for (size_t r = 0;; ++r) {
            for (auto& m : mate) advance(m);
            const uint64_t rem0 = remaining(mate[0]), rem1 = remaining(mate[1]);
            if (rem0 == 0 && rem1 == 0) break;                                  // both files used up
            if (rem0 == 0 || rem1 == 0) throw Error(SCG_ERR_IO, "different number of reads in paired FASTQ files");   // process_data.hpp:284-285
            const uint64_t np = std::min(rem0, rem1);
            Round& R = *rounds[(r % D) * 2 + (r / D) % 2];
            DeviceGuard g(R.device);
            if (R.busy) { HIP_CHECK(hipStreamSynchronize(R.stream)); R.busy = false; }
            ScgReads reads[2];
            uint32_t max_len = 0;
            for (int i = 0; i < 2; ++i) {
                HostWindow& hw = mate[i].win[mate[i].cur];
                enqueue_range(R, i, hw, mate[i].k, np);
                max_len = std::max(max_len, hw.w.max_len);
            }
            for (int i = 0; i < 2; ++i) {
                reads[i] = make_reads(R.seqs[i].as<char>(), R.offs[i].as<uint32_t>(), 0, static_cast<int32_t>(std::min<uint32_t>(max_len, 1u << 30)));
            }
            launch_batch_paired(plans[r % D], reads[0], reads[1], static_cast<int64_t>(np), R.stream);
            HIP_CHECK(hipEventRecord(R.done, R.stream));
            R.busy = true;
            for (auto& m : mate) {
                m.win[m.cur].readers.push_back(std::make_pair(R.device, R.done));
                m.k += np;
            }
            ++n_rounds;
        }
        for (auto& rp : rounds) {
            DeviceGuard g(rp->device);
            HIP_CHECK(hipStreamSynchronize(rp->stream));
            rp->busy = false;
        }
        if (tr.on) std::fprintf(stderr, "[scg]   paired rounds over %zu device(s): %zu rounds of <= %zu MB windows, host scan %.2f ms\n", D, n_rounds, window >> 20, t_fill);
        tr.mark("  rounds");
    }

private:
    struct HostWindow {
        PinnedBuf text, offs;
        scg::ParsedWindow w;
        std::vector<std::pair<int, hipEvent_t> > readers;      // rounds whose gathers read this window
    };
    struct Mate {
        scg::TextSource* src = nullptr;
        HostWindow win[3];
        int cur = -1;
        uint64_t k = 0;              // records of the current window that have been paired
        bool done = false;
    };
    struct Round {
        int device = 0;
        hipStream_t stream = nullptr;
        hipEvent_t done = nullptr;
        DevBuf seqs[2], offs[2];
        bool busy = false;
    };
    std::vector<int> devices;
    size_t window, cap_lines = 0, cap_records = 0, cap_seq = 0, n_rounds = 0;
    Mate mate[2];
    std::vector<std::unique_ptr<Round> > rounds;
    Trace tr;
    double t_fill = 0;

    static uint64_t remaining(const Mate& m) { return m.cur < 0 ? 0 : m.win[m.cur].w.n_records - m.k; }

    // A mate whose window is used up takes its next one (into the buffer whose readers have long finished).
    void advance(Mate& m) {
        if (m.done || remaining(m) > 0) return;
        const int next = (m.cur + 1) % 3;
        HostWindow& hw = m.win[next];
        for (auto& rd : hw.readers) {
            DeviceGuard g(rd.first);
            HIP_CHECK(hipEventSynchronize(rd.second));
        }
        hw.readers.clear();
        const auto f0 = std::chrono::steady_clock::now();
        const size_t bytes = m.src->next_parsed(hw.text.as<char>(), window, hw.offs.as<uint32_t>(), cap_records + 1, hw.w);
        t_fill += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - f0).count();
        if (m.src->unusual()) throw UnusualInput();
        if (bytes == 0) { m.done = true; m.cur = -1; return; }
        if (hw.w.seq_bytes > cap_seq || hw.w.n_records > cap_records) throw UnusualInput();
        m.cur = next;
        m.k = 0;
    }

    // Records [k, k + n) of a parsed window -> the round's device buffers of mate i.
    void enqueue_range(Round& R, int i, const HostWindow& hw, uint64_t k, uint64_t n) {
        scg::GatherSegments G;
        G.n = 0;
        uint32_t rec = 0, at = 0;
        uint64_t base = 0;
        const uint32_t* offs = hw.offs.as<uint32_t>();
        for (int sgm = 0; sgm < hw.w.n_segs && base < k + n; ++sgm) {
            const scg::ParsedSegment& g = hw.w.seg[sgm];
            const uint64_t lo = std::max<uint64_t>(k, base), hi = std::min<uint64_t>(k + n, base + g.n_records);
            if (hi > lo) {
                const uint32_t j0 = static_cast<uint32_t>(lo - base), j1 = static_cast<uint32_t>(hi - base);
                const uint32_t* so = offs + g.off_at;
                G.seq_src[G.n] = hw.text.as<char>() + g.seq_at + so[j0];
                G.off_src[G.n] = so + j0;
                G.off_base[G.n] = so[j0];
                G.seq_at[G.n] = at;
                G.first[G.n] = rec;
                at += so[j1] - so[j0];
                rec += j1 - j0;
                ++G.n;
            }
            base += g.n_records;
        }
        G.seq_at[G.n] = at;
        G.first[G.n] = rec;
        HIP_CHECK(scg::launch_gather_segments(R.seqs[i].as<char>(), R.offs[i].as<uint32_t>(), G, R.stream));
    }
};

void reset_plan(scg_plan* P) {
    DeviceGuard g(P->device);
    if (P->n_counters) HIP_CHECK(hipMemset(P->counters, 0, static_cast<size_t>(P->n_counters) * sizeof(int32_t)));
    if (P->replica_shift > 0) HIP_CHECK(hipMemset(P->replicas.p, 0, P->replicas.bytes));
    HIP_CHECK(hipStreamSynchronize(nullptr));               // (the fills are only enqueued: scg_plan::upload)
    for (auto& kv : P->pair_stream) {                       // (sparse mode: batches in flight are let finish and dropped)
        if (kv.second.pending) { HIP_CHECK(hipEventSynchronize(kv.second.done)); kv.second.pending = 0; }
    }
    P->sparse_counts.clear();
    P->total = 0;
}

// A gzip file the parallel decoder (scg_pgzip.h) handed back gets a second try with one inflate stream before the
// host readers take it.
bool is_parallel_gzip(const scg::TextSource* s) { return s && std::strcmp(s->kind(), "gzip-parallel") == 0; }

bool device_scan_enabled() {
    const char* e = std::getenv("SCG_DEVICE_SCAN");          // test hook: 0 keeps the host parsers
    return !(e && *e == '0');
}

// One single-end file on a set of plans (one per device).  Ordinary files go through the device scan; whatever it
// declines is redone on the first plan by the host readers (count_single_end_file), which end in the sequential
// reference-exact parser.
void count_single_end(const std::vector<scg_plan*>& plans, const char* path, scg::FastqStream& fq, int nthreads) {
    if (device_scan_enabled()) {
        bool done = false;
        if (device_inflate_enabled()) {
            // BGZF: members inflated on the device; whatever that declines gets the host threads' zlib next
            try {
                std::unique_ptr<scg::TextSource> src = scg::TextSource::open(path, scg::default_host_threads(nthreads));
                if (src->has_members()) {
                    std::vector<int> devs;
                    for (scg_plan* P : plans) devs.push_back(P->device);
                    InflatePipeline pipe(*src, devs);
                    pipe.run(plans);
                    done = true;
                }
            } catch (const UnusualInput&) {
                if (device_inflate_strict()) inflate_declined();
                for (scg_plan* P : plans) reset_plan(P);
            }
            if (done) return;
        }
        for (int attempt = 0; attempt < 2 && !done; ++attempt) {
            bool again = false;
            try {
                std::unique_ptr<scg::TextSource> src = scg::TextSource::open(path, scg::default_host_threads(nthreads), attempt == 0);
                again = attempt == 0 && is_parallel_gzip(src.get());
                count_text_stream(plans, *src);
                done = true;
            } catch (const UnusualInput&) {
                for (scg_plan* P : plans) reset_plan(P);
            }
            if (!again) break;
        }
        if (done) return;
    }
    DeviceGuard g(plans[0]->device);
    count_single_end_file(plans[0], path, fq, nthreads, nullptr, nullptr, nullptr);
}

// The reference's totals and counters are 32-bit `int`s (SingleBarcodeSingleEnd.hpp:132-133) and R integers
// are 32-bit; a file with more reads than that would overflow them silently there.  Here the total is kept
// in 64 bits and narrowing at the ABI is checked; no counter can exceed the total.
int32_t narrow_total(int64_t total) {
    if (total > static_cast<int64_t>(INT32_MAX)) {
        throw Error(SCG_ERR_INVALID, "number of reads (" + std::to_string(total) + ") exceeds the 32-bit range of the count vectors");
    }
    return static_cast<int32_t>(total);
}

void read_counters(scg_plan* P, int32_t* counts_out) {
    int32_t flag = 0;
    HIP_CHECK(hipMemcpy(&flag, P->error_flag.p, sizeof(flag), hipMemcpyDeviceToHost));
    if (flag) {
        throw Error(SCG_ERR_INVALID, "a read is longer than the max_len declared for its batch: counts are incomplete");
    }
    if (counts_out && P->n_counters) {
        HIP_CHECK(hipMemcpy(counts_out, P->counters, static_cast<size_t>(P->n_counters) * sizeof(int32_t), hipMemcpyDeviceToHost));
    }
}

// ---- devices and plan sets ---------------------------------------------------------------------------------
// Which devices a file-level call may use: $SCG_DEVICES ("all", or a comma list in which an id may repeat: several
// pipelines on one card) if set; else every visible device, the calling thread's current one ($SCG_DEVICE) first.
thread_local std::vector<int> tl_devices;       // scg_set_devices(): overrides $SCG_DEVICES for the calling thread

std::vector<int> device_list(bool* explicit_list) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        throw Error(SCG_ERR_DEVICE, "no HIP device available: libscg has no CPU fallback");
    }
    if (!tl_devices.empty()) {
        for (int v : tl_devices) {
            if (v < 0 || v >= n) throw Error(SCG_ERR_DEVICE, "HIP device " + std::to_string(v) + " out of range (" + std::to_string(n) + " visible)");
        }
        if (explicit_list) *explicit_list = true;
        return tl_devices;
    }
    std::vector<int> out;
    const char* env = std::getenv("SCG_DEVICES");
    if (explicit_list) *explicit_list = env && *env && std::strcmp(env, "all") != 0;
    if (env && *env && std::strcmp(env, "all") != 0) {
        const char* p = env;
        while (*p) {
            char* end = nullptr;
            const long v = std::strtol(p, &end, 10);
            if (end == p) throw Error(SCG_ERR_DEVICE, std::string("cannot parse SCG_DEVICES='") + env + "'");
            if (v < 0 || v >= n) throw Error(SCG_ERR_DEVICE, "HIP device " + std::to_string(v) + " out of range (" + std::to_string(n) + " visible)");
            out.push_back(static_cast<int>(v));
            p = end;
            while (*p == ',' || *p == ' ') ++p;
        }
        if (out.empty()) throw Error(SCG_ERR_DEVICE, "SCG_DEVICES lists no device");
        return out;
    }
    const int first = resolve_device(-1);
    const bool only_one = std::getenv("SCG_DEVICE") && *std::getenv("SCG_DEVICE") && !(env && *env);
    out.push_back(first);
    if (!only_one) for (int d = 0; d < n; ++d) if (d != first) out.push_back(d);
    return out;
}

// Devices for ONE input of about `text_bytes` of FASTQ text: an explicit $SCG_DEVICES is taken as given; otherwise one
// more device per four windows of text, so that small files do not pay for contexts and pinned buffers they cannot use.
std::vector<int> devices_for_input(uint64_t text_bytes) {
    bool given = false;
    std::vector<int> all = device_list(&given);
    if (given) return all;
    const uint64_t per_device = uint64_t(4) * scan_window_bytes(~uint64_t(0) >> 8);
    const size_t want = static_cast<size_t>(std::max<uint64_t>(1, text_bytes / per_device));
    if (all.size() > want) all.resize(want);
    return all;
}

uint64_t text_bytes_hint(const char* path) {
    struct stat st;
    if (!path || ::stat(path, &st) != 0) return 0;
    unsigned char h[2] = {0, 0};
    FILE* f = std::fopen(path, "rb");
    size_t got = 0;
    if (f) { got = std::fread(h, 1, 2, f); std::fclose(f); }
    const bool gz = got == 2 && h[0] == 0x1f && h[1] == 0x8b;
    return static_cast<uint64_t>(st.st_size) * (gz ? 5 : 1);
}

// A second plan with the same compiled (host-side) content, for another device.  Call before to_device().
std::unique_ptr<scg_plan> clone_compiled(const scg_plan& a) {
    std::unique_ptr<scg_plan> b(new scg_plan);
    b->kind = a.kind;
    b->ht1 = a.ht1; b->ht2 = a.ht2;
    b->scan1 = a.scan1; b->scan2 = a.scan2;
    b->htab[0] = a.htab[0]; b->htab[1] = a.htab[1];
    b->hpairs = a.hpairs;
    b->htab_combined = a.htab_combined;
    b->n_pool[0] = a.n_pool[0]; b->n_pool[1] = a.n_pool[1];
    b->max_mm1 = a.max_mm1; b->max_mm2 = a.max_mm2;
    b->rev1 = a.rev1; b->rev2 = a.rev2; b->randomized = a.randomized; b->use_first = a.use_first;
    b->diagnostics = a.diagnostics;
    b->first1 = a.first1; b->first2 = a.first2;
    b->n_counters = a.n_counters;
    b->sparse = a.sparse;
    return b;
}


// Files over devices inside one call (the matrixOf* functions: R/countSingleBarcodes.R:112-126, R/countComboBarcodes.R:149-164,
// R/countDualBarcodes.R:205-254 hand the files to BiocParallel workers): every device runs one pipeline at a time and takes
// the next unprocessed file when it is done; per_file(plan, i) counts file i and stores its column.  The error of the
// lowest-numbered failing file is reported, as a serial loop over the files would.
void schedule_files(int32_t n_files, const PlanSet& set, const std::function<void(scg_plan*, int32_t)>& per_file) {
    std::atomic<int32_t> next(0), first_bad(n_files);
    std::mutex mu;
    int32_t bad = n_files;
    int bad_code = 0;
    std::string bad_msg;
    auto worker = [&](scg_plan* P) {
        for (;;) {
            const int32_t i = next.fetch_add(1);
            if (i >= n_files) return;
            if (i > first_bad.load()) return;      // a file before this one has failed: the call reports that error, whatever comes after
            int code = 0;
            std::string msg;
            try {
                DeviceGuard g(P->device);
                per_file(P, i);
                continue;
            } catch (const Error& e) { code = e.code; msg = e.what();
            } catch (const std::bad_alloc&) { code = SCG_ERR_DEVICE; msg = "out of host memory";
            } catch (const std::exception& e) { code = SCG_ERR_INVALID; msg = e.what(); }
            std::lock_guard<std::mutex> g(mu);
            if (i < bad) { bad = i; bad_code = code; bad_msg = msg; first_bad.store(i); }
        }
    };
    std::vector<std::thread> th;
    for (size_t d = 1; d < set.plans.size(); ++d) th.emplace_back(worker, set.plans[d].get());
    worker(set.plans[0].get());
    for (auto& t : th) t.join();
    if (bad < n_files) throw Error(bad_code, bad_msg);
}

// One single-end file for a file-level entry point.  `compile` (template + pools -> plan: host work only) runs on a second
// thread while the first window of text is read, copied to the first device and scanned; then the plans go to the devices
// and the file is counted.  Errors keep the reference's order: the reader was opened by the caller, the handler's
// constructor (compile) comes before anything met while reading.
std::unique_ptr<PlanSet> compile_and_count_single_end(const char* path, scg::FastqStream& fq, int nthreads, Compile compile) {
    Trace tr;
    std::unique_ptr<scg_plan> compiled;
    std::exception_ptr compile_err, early;
    std::thread th([&] {
        try { compiled = compile(); } catch (...) { compile_err = std::current_exception(); }
    });
    std::vector<int> devices;
    std::unique_ptr<scg::TextSource> src;
    std::unique_ptr<TextPipeline> pipe;
    std::unique_ptr<InflatePipeline> inflate;
    bool parallel_gzip_declined = false, inflate_declined_early = false;
    try {
        devices = devices_for_input(text_bytes_hint(path));
        if (device_scan_enabled()) {
            const int host_threads = scg::default_host_threads(nthreads, static_cast<int>(devices.size()));
            // an ordinary gzip file of some size: decoded by the device when it is of the plain kind (one member, or several large ones), its text
            // left in HBM; whatever that decoder declines goes to the host threads' decoder
            if (scg::TextSource::ordinary_gzip(path, host_threads)) {
                src = scg::TextSource::open_on_device(path, devices[0], host_threads);
                // (test hook SCG_DEVICE_GUNZIP=2: a file the device decoder hands back is an error, so that a test on a
                // well-formed file cannot pass on the host decoders)
                const char* e = std::getenv("SCG_DEVICE_GUNZIP");
                if (!src && e && *e == '2') throw Error(SCG_ERR_UNSUPPORTED, "the device gzip decoder handed the file back (SCG_DEVICE_GUNZIP=2 forbids the fall-back)");
            }
            if (src) devices.resize(1);
            else src = scg::TextSource::open(path, host_threads);
            if (src->has_members() && device_inflate_enabled()) {
                inflate.reset(new InflatePipeline(*src, devices));
                devices.resize(inflate->n_devices());  // (one device when the others cannot be reached over xGMI)
                inflate->start();
            } else {
                pipe.reset(new TextPipeline(*src, devices));
                pipe->start();
            }
        }
    } catch (const UnusualInput&) {
        parallel_gzip_declined = is_parallel_gzip(src.get());
        inflate_declined_early = src && src->has_members() && device_inflate_enabled();      // the first window already: same second chance as any later one
        pipe.reset();
        inflate.reset();
    } catch (...) {
        early = std::current_exception();
        pipe.reset();
        inflate.reset();
    }
    th.join();
    if (compile_err) std::rethrow_exception(compile_err);
    if (early) std::rethrow_exception(early);
    tr.mark("compile + first window");
    std::unique_ptr<PlanSet> set(new PlanSet(std::move(compiled), devices));
    tr.mark("upload to device(s)");
    bool done = false;
    if (inflate || inflate_declined_early) {
        if (inflate_declined_early && device_inflate_strict()) inflate_declined();
        if (inflate) {
            try {
                inflate->run(set->all());
                done = true;
            } catch (const UnusualInput&) {
                if (device_inflate_strict()) inflate_declined();
                inflate.reset();                       // (its kernels have finished before the counters are cleared)
                set->reset();
            }
            inflate.reset();
        }
        if (!done) {
            // second chance for BGZF: members inflated by the host threads' zlib, records scanned on the device
            try {
                src = scg::TextSource::open(path, scg::default_host_threads(nthreads));
                pipe.reset(new TextPipeline(*src, devices));
            } catch (const UnusualInput&) {
                pipe.reset();
            }
        }
    }
    bool retry_gzip = parallel_gzip_declined;
    if (pipe) {
        const bool parallel = is_parallel_gzip(src.get());
        const bool was_on_device = src && src->device_resident();
        try {
            pipe->run(set->all());
            done = true;
        } catch (const UnusualInput&) {
            pipe.reset();
            set->reset();
            retry_gzip = parallel;
        }
        pipe.reset();
        if (!done && was_on_device) {
            // the device's gzip decoder gave up on a later group of the file (or the text is not ordinary records): the host
            // threads' decoder next, with its own fall-backs behind it
            try {
                src = scg::TextSource::open(path, scg::default_host_threads(nthreads));
                retry_gzip = is_parallel_gzip(src.get());
                pipe.reset(new TextPipeline(*src, devices));
                pipe->run(set->all());
                done = true;
            } catch (const UnusualInput&) {
                pipe.reset();
                set->reset();
            }
            pipe.reset();
        }
    }
    if (!done && retry_gzip) {
        // the parallel gzip decoder handed the file back: one inflate stream, records scanned on the device
        try {
            src = scg::TextSource::open(path, scg::default_host_threads(nthreads), false);
            pipe.reset(new TextPipeline(*src, devices));
            pipe->run(set->all());
            done = true;
        } catch (const UnusualInput&) {
            pipe.reset();
            set->reset();
        }
        pipe.reset();
    }
    if (!done) {
        DeviceGuard g(set->first()->device);
        count_single_end_file(set->first(), path, fq, nthreads, nullptr, nullptr, nullptr);
    }
    tr.mark("count file");
    return set;
}

void combo_compact(const int32_t* cells, int32_t n0, int32_t n1, int32_t** indices_out, int32_t** freq_out, int64_t* k_out) {
    int64_t total = static_cast<int64_t>(n0) * n1, k = 0;
    for (int64_t c = 0; c < total; ++c) k += cells[c] != 0;
    int32_t* idx = static_cast<int32_t*>(std::malloc(sizeof(int32_t) * static_cast<size_t>(2 * k + 1)));
    int32_t* freq = static_cast<int32_t*>(std::malloc(sizeof(int32_t) * static_cast<size_t>(k + 1)));
    if (!idx || !freq) { std::free(idx); std::free(freq); throw std::bad_alloc(); }
    int64_t j = 0;
    // cell order = (first, second) lexicographic order = the reference's sorted column order
    for (int64_t c = 0; c < total; ++c) {
        if (cells[c]) {
            idx[2 * j] = static_cast<int32_t>(c / n1);
            idx[2 * j + 1] = static_cast<int32_t>(c % n1);
            freq[j] = cells[c];
            ++j;
        }
    }
    *indices_out = idx; *freq_out = freq; *k_out = k;
}

// Sparse mode: (first << 32 | second) -> count, as the reference's sorted run-length form (src/utils.h:14-45).
void combos_from_sparse(const std::unordered_map<uint64_t, int64_t>& m, int32_t** indices_out, int32_t** freq_out, int64_t* k_out) {
    std::vector<std::pair<uint64_t, int64_t> > rows(m.begin(), m.end());
    std::sort(rows.begin(), rows.end());                    // key order = (first, second) order
    const size_t k = rows.size();
    int32_t* idx = static_cast<int32_t*>(std::malloc(sizeof(int32_t) * (2 * k + 1)));
    int32_t* freq = static_cast<int32_t*>(std::malloc(sizeof(int32_t) * (k + 1)));
    if (!idx || !freq) { std::free(idx); std::free(freq); throw std::bad_alloc(); }
    for (size_t j = 0; j < k; ++j) {
        if (rows[j].second > static_cast<int64_t>(INT32_MAX)) { std::free(idx); std::free(freq); throw Error(SCG_ERR_INVALID, "a count exceeds the 32-bit range of the count vectors"); }
        idx[2 * j] = static_cast<int32_t>(rows[j].first >> 32);
        idx[2 * j + 1] = static_cast<int32_t>(rows[j].first & 0xFFFFFFFFu);
        freq[j] = static_cast<int32_t>(rows[j].second);
    }
    *indices_out = idx; *freq_out = freq; *k_out = static_cast<int64_t>(k);
}

// [n_pool valid][b1][b2][uid1 x uid2] -> the reference's outputs: invalid combinations by first pool
// index, merged (several uids of one IUPAC barcode share an index), sorted by (first, second).
// (sparse: the plan is in sparse mode and these are its combinations by sequence uid, in place of the dense cells)
void diagnostics_from_counters(const scg_plan* P, const std::vector<int32_t>& all, int32_t* counts_out,
                               int32_t** idx_out, int32_t** freq_out, int64_t* k_out, int32_t* b1, int32_t* b2,
                               const std::unordered_map<uint64_t, int64_t>* sparse) {
    const int32_t n_pool = P->n_pool[0];
    if (counts_out) std::copy(all.begin(), all.begin() + n_pool, counts_out);
    *b1 = all[n_pool];
    *b2 = all[n_pool + 1];
    const int32_t* cells = all.data() + n_pool + 2;
    const size_t nu1 = P->first1.size(), nu2 = P->first2.size();
    std::vector<std::pair<std::pair<int32_t, int32_t>, int32_t> > found;
    if (sparse) {
        for (auto& kv : *sparse) {
            const size_t u1 = static_cast<size_t>(kv.first >> 32), u2 = static_cast<size_t>(kv.first & 0xFFFFFFFFu);
            if (u1 >= nu1 || u2 >= nu2) throw Error(SCG_ERR_DEVICE, "internal: combination out of range");
            if (kv.second > static_cast<int64_t>(INT32_MAX)) throw Error(SCG_ERR_INVALID, "a count exceeds the 32-bit range of the count vectors");
            found.push_back(std::make_pair(std::make_pair(P->first1[u1], P->first2[u2]), static_cast<int32_t>(kv.second)));
        }
    } else {
        for (size_t u1 = 0; u1 < nu1; ++u1) {
            for (size_t u2 = 0; u2 < nu2; ++u2) {
                int32_t c = cells[u1 * nu2 + u2];
                if (c) found.push_back(std::make_pair(std::make_pair(P->first1[u1], P->first2[u2]), c));
            }
        }
    }
    std::sort(found.begin(), found.end());
    std::vector<int32_t> idx, freq;
    for (size_t i = 0; i < found.size(); ++i) {
        if (i && found[i].first == found[i - 1].first) {
            freq.back() += found[i].second;
        } else {
            idx.push_back(found[i].first.first);
            idx.push_back(found[i].first.second);
            freq.push_back(found[i].second);
        }
    }
    int32_t* oi = static_cast<int32_t*>(std::malloc(sizeof(int32_t) * (idx.size() + 1)));
    int32_t* of = static_cast<int32_t*>(std::malloc(sizeof(int32_t) * (freq.size() + 1)));
    if (!oi || !of) { std::free(oi); std::free(of); throw std::bad_alloc(); }
    std::copy(idx.begin(), idx.end(), oi);
    std::copy(freq.begin(), freq.end(), of);
    *idx_out = oi; *freq_out = of; *k_out = static_cast<int64_t>(freq.size());
}

// Appends the reads [from, to) of `src` to `dst`.
void append_reads(scg::ReadBatch& dst, const scg::ReadBatch& src, int64_t from, int64_t to) {
    if (to <= from) return;
    const uint64_t b0 = src.offsets[from], b1 = src.offsets[to];
    const uint64_t base = dst.seqs.size();
    dst.seqs.insert(dst.seqs.end(), src.seqs.begin() + b0, src.seqs.begin() + b1);
    for (int64_t i = from + 1; i <= to; ++i) dst.offsets.push_back(base + (src.offsets[i] - b0));
}

// Both FASTQ files of a paired-end run (process_data.hpp:224-340).  Plain files are parsed by the
// multi-threaded reader, each file on its own; the two read streams are re-cut into batches of
// equal read counts (pair i = read i of both files).  gzip input or anything unusual falls back to
// the sequential readers in lock-step.  Unequal read counts => the reference's error.
void count_paired_host(scg_plan* P, const char* path1, const char* path2, scg::FastqStream& fq1, scg::FastqStream& fq2, int nthreads) {
    Stager st;
    auto launch_pair = [&](const scg::ReadBatch& x, const scg::ReadBatch& y) {
        auto& s = st.acquire();
        ScgReads R1 = st.stage(s, 0, x);
        ScgReads R2 = st.stage(s, 1, y);
        launch_batch_paired(P, R1, R2, x.size(), s.stream);
        s.busy = true;
    };
    const int threads = scg::default_host_threads(nthreads);
    if (threads > 1 && scg::ParallelFastq::is_plain_file(path1) && scg::ParallelFastq::is_plain_file(path2)) {
        const int half = threads > 3 ? threads / 2 : 2;
        scg::ParallelFastq pf1(path1, half), pf2(path2, half);
        scg::ReadBatch q1, q2;          // reads parsed but not yet paired
        q1.clear(); q2.clear();
        std::vector<scg::ReadBatch> w1, w2;
        bool more1 = true, more2 = true, odd = false;
        while (more1 || more2) {
            // advance whichever stream is behind (both at first)
            const bool need1 = more1 && q1.size() <= q2.size();
            const bool need2 = more2 && q2.size() <= q1.size();
            std::thread t;
            bool got2 = false;
            if (need2) t = std::thread([&] { got2 = pf2.next_window(w2); });
            bool got1 = need1 ? pf1.next_window(w1) : false;
            if (t.joinable()) t.join();
            if (need1) { if (got1) for (auto& b : w1) append_reads(q1, b, 0, b.size()); else more1 = false; }
            if (need2) { if (got2) for (auto& b : w2) append_reads(q2, b, 0, b.size()); else more2 = false; }
            if (pf1.unusual() || pf2.unusual()) { odd = true; break; }
            const int64_t n = std::min(q1.size(), q2.size());
            if (n > 0) {
                scg::ReadBatch a, b2, r1, r2;
                a.clear(); b2.clear(); r1.clear(); r2.clear();
                append_reads(a, q1, 0, n); append_reads(r1, q1, n, q1.size());
                append_reads(b2, q2, 0, n); append_reads(r2, q2, n, q2.size());
                launch_pair(a, b2);
                q1.seqs.swap(r1.seqs); q1.offsets.swap(r1.offsets);
                q2.seqs.swap(r2.seqs); q2.offsets.swap(r2.offsets);
            }
            if (!need1 && !need2) break;
        }
        st.drain();
        if (!odd) {
            if (q1.size() != q2.size()) {
                throw Error(SCG_ERR_IO, "different number of reads in paired FASTQ files");   // process_data.hpp:284-285
            }
            return;
        }
        HIP_CHECK(hipMemset(P->counters, 0, static_cast<size_t>(P->n_counters) * sizeof(int32_t)));
        HIP_CHECK(hipStreamSynchronize(nullptr));
        P->total = 0;
    }
    scg::ReadBatch b1, b2;
    for (;;) {
        bool more1 = fq1.next_batch(b1, BATCH_READS / 4, INT64_MAX);
        bool more2 = fq2.next_batch(b2, BATCH_READS / 4, INT64_MAX);
        if (b1.size() != b2.size()) {
            throw Error(SCG_ERR_IO, "different number of reads in paired FASTQ files");   // process_data.hpp:284-285
        }
        if (!more1 && !more2) break;
        launch_pair(b1, b2);
    }
    st.drain();
}

// One mate of a paired run.  An ordinary gzip mate is decoded by the device when it is of the plain kind, one
// mate after the other; what the device hands back, and every other form, is opened the ordinary way (two parallel gzip
// decoders share the host threads).
std::unique_ptr<scg::TextSource> open_paired_mate(const char* path, int device, int threads, bool parallel_gzip) {
    std::unique_ptr<scg::TextSource> s;
    if (parallel_gzip && scg::TextSource::ordinary_gzip(path, threads)) {
        s = scg::TextSource::open_on_device(path, device, threads);
        const char* e = std::getenv("SCG_DEVICE_GUNZIP");          // (test hook, as for single-end input)
        if (!s && e && *e == '2') throw Error(SCG_ERR_UNSUPPORTED, "the device gzip decoder handed the file back (SCG_DEVICE_GUNZIP=2 forbids the fall-back)");
    }
    if (!s) s = scg::TextSource::open(path, threads, parallel_gzip, std::max(2, threads / 2));
    return s;
}

void count_paired_files(scg_plan* P, const char* path1, const char* path2, scg::FastqStream& fq1, scg::FastqStream& fq2, int nthreads,
                        bool try_device_inflate, bool parallel_gzip) {
    if (device_scan_enabled()) {
        // ordinary files: windows of sequences paired on the device (PairedPipeline); anything else: the host readers
        bool done = false, declined_gzip = false;
        try {
            const int threads = scg::default_host_threads(nthreads);
            // (two parallel gzip decoders share the host threads)
            std::unique_ptr<scg::TextSource> s1 = open_paired_mate(path1, P->device, threads, parallel_gzip),
                                             s2 = open_paired_mate(path2, P->device, threads, parallel_gzip);
            declined_gzip = is_parallel_gzip(s1.get()) || is_parallel_gzip(s2.get()) || s1->device_resident() || s2->device_resident();
            bool device_inflate = try_device_inflate;
            for (;;) {
                {
                    PairedPipeline pipe(P->device, *s1, *s2, device_inflate);
                    const bool retry = pipe.inflates();
                    try {
                        pipe.run(P);
                        done = true;
                    } catch (const UnusualInput&) {
                        if (!retry) throw;
                        if (device_inflate_strict()) inflate_declined();
                    }
                }
                if (done) break;
                // a BGZF mate the device handed back: once more with the host threads' zlib (the pipeline is gone: its
                // kernels have finished)
                reset_plan(P);
                s1 = scg::TextSource::open(path1, threads, parallel_gzip, std::max(2, threads / 2));
                s2 = scg::TextSource::open(path2, threads, parallel_gzip, std::max(2, threads / 2));
                device_inflate = false;
            }
        } catch (const UnusualInput&) {
            reset_plan(P);
        }
        if (done) return;
        if (declined_gzip) {       // a gzip mate the parallel decoder handed back: once more with one inflate stream per mate
            count_paired_files(P, path1, path2, fq1, fq2, nthreads, try_device_inflate, false);
            return;
        }
    }
    count_paired_host(P, path1, path2, fq1, fq2, nthreads);
}

// One pair of files, one call: as compile_and_count_single_end, the templates and libraries are compiled on a second
// thread while the first window of each file is read and sent on its way.  Plain mates and more than one device: the
// pairs go round-robin over all of them (PairedRounds); otherwise one device.
std::unique_ptr<PlanSet> compile_and_count_paired(const char* path1, const char* path2, scg::FastqStream& fq1, scg::FastqStream& fq2, int nthreads,
                                                  Compile compile) {
    Trace tr;
    std::unique_ptr<scg_plan> P;
    std::exception_ptr compile_err, early;
    std::thread th([&] {
        try { P = compile(); } catch (...) { compile_err = std::current_exception(); }
    });
    std::vector<int> devices;
    std::unique_ptr<scg::TextSource> s1, s2;
    std::unique_ptr<PairedPipeline> pipe;
    std::unique_ptr<PairedRounds> rounds;
    bool gzip_parallel = false, gzip_declined = false;
    try {
        devices = devices_for_input(text_bytes_hint(path1) + text_bytes_hint(path2));
        if (device_scan_enabled()) {
            const int threads = scg::default_host_threads(nthreads, static_cast<int>(devices.size()));
            s1 = open_paired_mate(path1, devices[0], threads, true);
            s2 = open_paired_mate(path2, devices[0], threads, true);
            gzip_parallel = is_parallel_gzip(s1.get()) || is_parallel_gzip(s2.get());
            if (devices.size() > 1 && s1->parses() && s2->parses() && host_scan_enabled()) {
                rounds.reset(new PairedRounds(devices, *s1, *s2));
            } else {
                devices.resize(1);
                pipe.reset(new PairedPipeline(devices[0], *s1, *s2, true));
                pipe->start();
            }
        } else {
            devices.resize(1);
        }
    } catch (const UnusualInput&) {
        pipe.reset();
        rounds.reset();
        gzip_declined = gzip_parallel;
        if (!devices.empty()) devices.resize(1);
    } catch (...) {
        early = std::current_exception();
        pipe.reset();
        rounds.reset();
    }
    th.join();
    if (compile_err) std::rethrow_exception(compile_err);
    if (early) std::rethrow_exception(early);
    tr.mark("compile + first windows");
    std::unique_ptr<PlanSet> set(new PlanSet(std::move(P), devices));
    scg_plan* const first = set->first();
    DeviceGuard g(first->device);
    tr.mark("upload to device(s)");
    bool done = false, inflate_declined_it = false;
    if (rounds) {
        try {
            rounds->run(set->all());
            done = true;
        } catch (const UnusualInput&) {
            rounds.reset();                        // (its kernels have finished before the counters are cleared)
            set->reset();
        }
        rounds.reset();
    }
    if (pipe) {
        try {
            pipe->run(first);
            done = true;
        } catch (const UnusualInput&) {
            inflate_declined_it = pipe->inflates();
            if (inflate_declined_it && device_inflate_strict()) inflate_declined();
            pipe.reset();                          // (its kernels have finished before the counters are cleared)
            set->reset();
            gzip_declined = gzip_parallel;
        }
        pipe.reset();
    }
    if (!done) {
        // a BGZF mate the device handed back gets the host threads' zlib next, a gzip mate the parallel decoder handed back
        // one inflate stream; everything else the host readers
        if (inflate_declined_it || gzip_declined) count_paired_files(first, path1, path2, fq1, fq2, nthreads, !inflate_declined_it, !gzip_declined);
        else count_paired_host(first, path1, path2, fq1, fq2, nthreads);
    }
    tr.mark("count files");
    return set;
}

PlanSet::PlanSet(std::unique_ptr<scg_plan> compiled, const std::vector<int>& devices) {
    for (size_t i = 1; i < devices.size(); ++i) plans.push_back(clone_compiled(*compiled));
    plans.insert(plans.begin(), std::move(compiled));
    for (size_t i = 0; i < plans.size(); ++i) plans[i]->to_device(devices[i]);
}
std::vector<scg_plan*> PlanSet::all() const {
    std::vector<scg_plan*> v;
    for (auto& p : plans) v.push_back(p.get());
    return v;
}
int64_t PlanSet::total() const {
    int64_t t = 0;
    for (auto& p : plans) t += p->total;
    return t;
}
void PlanSet::read(int32_t* counts_out) const {
    if (plans.size() == 1) {
        DeviceGuard g(plans[0]->device);
        read_counters(plans[0].get(), counts_out);
        return;
    }
    const size_t n = static_cast<size_t>(plans[0]->n_counters);
    std::vector<int64_t> acc(n, 0);
    std::vector<int32_t> part(n + 1);
    for (auto& p : plans) {
        DeviceGuard g(p->device);
        read_counters(p.get(), part.data());
        for (size_t i = 0; i < n; ++i) acc[i] += part[i];
    }
    if (counts_out) {
        for (size_t i = 0; i < n; ++i) {
            if (acc[i] > static_cast<int64_t>(INT32_MAX)) throw Error(SCG_ERR_INVALID, "a count exceeds the 32-bit range of the count vectors");
            counts_out[i] = static_cast<int32_t>(acc[i]);
        }
    }
}
void PlanSet::reset() const { for (auto& p : plans) reset_plan(p.get()); }
std::unordered_map<uint64_t, int64_t> PlanSet::sparse_merged() const {
    std::unordered_map<uint64_t, int64_t> all;
    for (auto& p : plans) {
        retire_all_pairs(p.get());
        if (all.empty()) all = p->sparse_counts;
        else for (auto& kv : p->sparse_counts) all[kv.first] += kv.second;
    }
    return all;
}

void release_cached_slots() { slot_pool().clear(); }
void set_thread_devices(const int* devices, int32_t n) { tl_devices.assign(devices, devices + n); }

} // namespace scgapi
