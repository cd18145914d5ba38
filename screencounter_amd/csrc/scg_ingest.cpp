// scg_ingest.cpp -- sources of raw FASTQ text for the device-side record scan (see scg_ingest.h).
#include "scg_ingest.h"
#include "scg_host.h"
#include "scg_pgzip.hpp"

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <dlfcn.h>
#include <exception>
#include <fcntl.h>
#include <mutex>
#include <string>
#include <sys/mman.h>
#include <sys/stat.h>
#include <thread>
#include <unistd.h>
#include <vector>
#include <zlib.h>

namespace scg {

struct WorkerPool::State {
    std::mutex mu;
    std::condition_variable wake, done;
    std::vector<std::thread> th;
    const std::function<void(int)>* fn = nullptr;
    int n = 0;
    std::atomic<int> next{0};
    int running = 0;             // workers still inside the current job
    uint64_t job = 0;
    bool stop = false;
    std::exception_ptr err;

    void share() {
        for (;;) {
            const int i = next.fetch_add(1);
            if (i >= n) return;
            try {
                (*fn)(i);
            } catch (...) {
                std::lock_guard<std::mutex> g(mu);
                if (!err) err = std::current_exception();
                next.store(n);
            }
        }
    }
    void worker() {
        uint64_t seen = 0;
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            wake.wait(lk, [&] { return stop || job != seen; });
            if (stop) return;
            seen = job;
            lk.unlock();
            share();
            lk.lock();
            if (--running == 0) done.notify_one();
        }
    }
};

WorkerPool::WorkerPool(int threads) : st(new State), n_threads(std::max(1, threads)) {
    for (int t = 1; t < n_threads; ++t) st->th.emplace_back([this] { st->worker(); });
}

WorkerPool::~WorkerPool() {
    {
        std::lock_guard<std::mutex> g(st->mu);
        st->stop = true;
    }
    st->wake.notify_all();
    for (auto& t : st->th) t.join();
    delete st;
}

void WorkerPool::run(int n, const std::function<void(int)>& fn) {
    if (n <= 0) return;
    if (n == 1 || st->th.empty()) {
        for (int i = 0; i < n; ++i) fn(i);
        return;
    }
    {
        std::lock_guard<std::mutex> g(st->mu);
        st->fn = &fn;
        st->n = n;
        st->next.store(0);
        st->running = static_cast<int>(st->th.size());
        st->err = nullptr;
        ++st->job;
    }
    st->wake.notify_all();
    st->share();
    std::unique_lock<std::mutex> lk(st->mu);
    st->done.wait(lk, [&] { return st->running == 0; });
    if (st->err) std::rethrow_exception(st->err);
}

size_t strict_record_end(const char* data, size_t size, size_t p, const char** seq_out, size_t* seq_len_out) {
    if (p >= size || data[p] != '@') return 0;
    const char* e = data + size;
    const char* l1 = static_cast<const char*>(std::memchr(data + p, '\n', size - p));
    if (!l1) return 0;
    const char* s0 = l1 + 1;
    const char* l2 = s0 < e ? static_cast<const char*>(std::memchr(s0, '\n', e - s0)) : nullptr;
    if (!l2) return 0;
    if (std::memchr(s0, '+', l2 - s0)) return 0;            // '+' would end the sequence early
    const char* p0 = l2 + 1;
    if (p0 >= e || *p0 != '+') return 0;
    const char* l3 = static_cast<const char*>(std::memchr(p0, '\n', e - p0));
    if (!l3) return 0;
    const char* q0 = l3 + 1;
    const size_t seq_len = static_cast<size_t>(l2 - s0);
    const char* l4 = q0 < e ? static_cast<const char*>(std::memchr(q0, '\n', e - q0)) : nullptr;
    const size_t qual_len = l4 ? static_cast<size_t>(l4 - q0) : static_cast<size_t>(e - q0);
    if (qual_len != seq_len) return 0;
    if (seq_len == 0 && !l4) return 0;                      // leave EOF corner cases to the sequential parser
    if (seq_out) { *seq_out = s0; *seq_len_out = seq_len; }
    return l4 ? static_cast<size_t>(l4 + 1 - data) : size;
}

namespace {

// The largest c <= len such that two consecutive ordinary records end exactly at c, searched among the line starts of
// the last `slack` bytes; 0 if there is none.  data[0] is a record start, so c is then a record boundary as long as
// the window holds ordinary records only -- which the device scan verifies for every window (scg_textscan.hip): a
// wrong guess here surfaces there as a line count that is not a multiple of four.
} // namespace

size_t find_cut(const char* data, size_t len, size_t slack) {
    const size_t floor = len > slack ? len - slack : 0;
    size_t pos = len;
    while (pos > floor) {
        // candidate: the line start that follows the last newline before pos - 1
        const char* nl = pos >= 2 ? static_cast<const char*>(memrchr(data + floor, '\n', pos - 1 - floor)) : nullptr;
        const size_t p = nl ? static_cast<size_t>(nl + 1 - data) : floor;
        if (nl || floor == 0) {
            const size_t e1 = strict_record_end(data, len, p);
            if (e1 && e1 < len) {
                const size_t e2 = strict_record_end(data, len, e1);
                if (e2 && e2 <= len && data[e2 - 1] == '\n') return e2;
            }
        }
        if (!nl) break;
        pos = p;      // next candidate: the line before
    }
    return 0;
}

namespace {

struct MappedFile {
    int fd = -1;
    const char* data = nullptr;
    size_t size = 0;          // bytes of content
    size_t mapped = 0;        // bytes of the mapping (>= size for an adopted anonymous one)
    explicit MappedFile(const char* path) {
        fd = ::open(path, O_RDONLY);
        if (fd < 0) throw Error(SCG_ERR_IO, std::string("failed to open file at '") + path + "'");
        struct stat st;
        if (::fstat(fd, &st) != 0) { ::close(fd); throw Error(SCG_ERR_IO, "failed to stat the FASTQ file"); }
        size = static_cast<size_t>(st.st_size);
        if (size) {
            void* m = ::mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
            if (m == MAP_FAILED) { ::close(fd); throw Error(SCG_ERR_IO, "failed to map the FASTQ file"); }
            ::madvise(m, size, MADV_SEQUENTIAL);
            data = static_cast<const char*>(m);
            mapped = size;
        }
    }
    // Adopts an anonymous mapping of `reserved` bytes holding `content` bytes of text.
    MappedFile(char* anonymous, size_t reserved, size_t content) : data(anonymous), size(content), mapped(reserved) {}
    MappedFile(MappedFile&& o) noexcept : fd(o.fd), data(o.data), size(o.size), mapped(o.mapped) { o.fd = -1; o.data = nullptr; o.size = o.mapped = 0; }
    ~MappedFile() {
        if (data && mapped) ::munmap(const_cast<char*>(data), mapped);
        if (fd >= 0) ::close(fd);
    }
    MappedFile(const MappedFile&) = delete;
    MappedFile& operator=(const MappedFile&) = delete;
};

// ---- plain file: windows are copied out of a mapping of the file by several threads.  Copying from the mapping
//      runs at 85-95 GB/s on tmpfs with 16 threads (pread() into the same pinned buffers: 50-58 GB/s, below the PCIe
//      link the windows go over next), but a 10 GB mapping that has been touched takes ~100 ms to unmap, single-threaded
//      and under the address-space write lock.  So every thread drops the page-table entries of the slice it has just
//      copied with madvise(MADV_DONTNEED) -- a read-lock operation that runs in parallel and leaves the page cache
//      alone -- and the final munmap finds nothing left to tear down ----
class PlainSource : public TextSource {
    MappedFile f;
    size_t pos = 0;
    WorkerPool pool;

    // The part of the mapping behind [a, b) is not needed again: whole pages inside it lose their entries.
    static void drop(const char* a, const char* b) {
        const uintptr_t page = 4096;
        const uintptr_t lo = (reinterpret_cast<uintptr_t>(a) + page - 1) & ~(page - 1);
        const uintptr_t hi = reinterpret_cast<uintptr_t>(b) & ~(page - 1);
        if (hi > lo) (void)::madvise(reinterpret_cast<void*>(lo), hi - lo, MADV_DONTNEED);
    }
    // How much text the next window takes (0: none, or odd set) and whether the file ends in it without a newline.
    size_t window(size_t cap, bool& pad) {
        pad = false;
        if (odd || pos >= f.size) return 0;
        const size_t left = f.size - pos;
        if (left + 1 <= cap) {
            pad = f.data[f.size - 1] != '\n';
            return left;
        }
        const size_t take = find_cut(f.data + pos, cap - 1);
        if (!take) odd = true;
        return take;
    }
    const char* what = "plain";
public:
    PlainSource(const char* path, int nthreads) : f(path), pool(nthreads) { threads = nthreads; }
    // text that already lies in memory (a gzip file inflated whole): consumed pages are given back as the windows go by
    PlainSource(MappedFile&& text, int nthreads, const char* kind_name) : f(std::move(text)), pool(nthreads), what(kind_name) { threads = nthreads; }
    const char* kind() const override { return what; }
    uint64_t size_hint() const override { return f.size - pos; }
    size_t next(char* dst, size_t cap) override {
        bool pad;
        size_t take = window(cap, pad);
        if (!take) return 0;
        const int parts = static_cast<int>(std::min<size_t>(static_cast<size_t>(threads), (take >> 22) + 1));
        const char* src = f.data + pos;
        pool.run(parts, [&](int i) {
            const size_t a = take * i / parts, b = take * (i + 1) / parts;
            std::memcpy(dst + a, src + a, b - a);
            drop(src + a, src + b);
        });
        pos += take;
        if (pad) dst[take++] = '\n';
        return take;
    }

    // The record scan on the host: the window is split at record boundaries (guessed by find_cut, confirmed by the
    // thread of the slice before, whose chain of records has to end exactly there); every thread walks the records of
    // its slice with the same definition of an ordinary record as the device scan (strict_record_end) and copies the
    // sequence lines out.  The threads read the mapping at the speed they would copy it, and write half as much.
    bool parses() const override { return true; }
    size_t next_parsed(char* seqs, size_t cap, uint32_t* offsets, size_t cap_offsets, ParsedWindow& out) override {
        out = ParsedWindow();
        bool pad;
        const size_t take = window(cap, pad);       // a final record without its newline is an ordinary record here
        if (!take) return 0;
        const char* src = f.data + pos;
        const size_t size = f.size - pos;           // records may be checked against text beyond the window
        const int max_parts = static_cast<int>(sizeof(out.seg) / sizeof(out.seg[0]));
        const size_t slice = std::min(size_t(4) << 20, std::max<size_t>(256, cap / 8));     // (small windows: tests)
        int parts = static_cast<int>(std::min<size_t>(static_cast<size_t>(threads), take / slice + 1));
        parts = std::max(1, std::min(parts, max_parts));
        // Slice i runs from boundary(i) to boundary(i + 1): each thread works out both ends of its own slice (the
        // guesses are deterministic, so neighbours agree); 0 means "no boundary found here, the slice before carries on".
        auto boundary = [&](int i) -> size_t {
            if (i <= 0) return 0;
            if (i >= parts) return take;
            return find_cut(src, take * i / parts, size_t(1) << 18);
        };
        const size_t per = cap_offsets / parts;
        std::atomic<bool> bad(false);
        pool.run(parts, [&](int i) {
            ParsedSegment& g = out.seg[i];
            g = ParsedSegment();
            // where the slice really starts and ends: a missing boundary hands the stretch to the slice before
            const size_t lo = boundary(i);
            size_t hi = take;
            if (i > 0 && lo == 0) { g.seq_at = take * i / parts; g.off_at = per * i; offsets[g.off_at] = 0; return; }
            for (int k = i + 1; k < parts; ++k) {
                const size_t c = boundary(k);
                if (c) { hi = c; break; }
            }
            g.seq_at = lo;
            g.off_at = per * i;
            char* o = seqs + g.seq_at;
            uint32_t* off = offsets + g.off_at;
            size_t p = lo, n = 0, at = 0;
            const size_t end = hi;
            while (p < end) {
                const char* s;
                size_t len;
                const size_t e = strict_record_end(src, size, p, &s, &len);
                if (!e || n + 2 > per) { bad.store(true); return; }
                std::memcpy(o + at, s, len);
                off[n++] = static_cast<uint32_t>(at);
                at += len;
                if (len > g.max_len) g.max_len = static_cast<uint32_t>(len);
                p = e;
            }
            if (p != end) { bad.store(true); return; }
            off[n] = static_cast<uint32_t>(at);
            g.n_records = static_cast<uint32_t>(n);
            g.seq_bytes = static_cast<uint32_t>(at);
        });
        if (bad.load()) { odd = true; return 0; }
        // Only now may the window's pages go: every thread works out its slice's ends from text that lies in its
        // neighbours' slices, and a page dropped from an ANONYMOUS mapping (a gzip file inflated whole) reads as zeros
        // afterwards -- a thread that dropped its slice early made a late neighbour find "no boundary here" and stand
        // back for records nobody then parsed (found by tools/gpu_ingest_fuzz.py, 1 file in 800; file-backed mappings
        // fault the same bytes in again and never showed it).
        pool.run(parts, [&](int i) { drop(src + take * i / parts, src + take * (i + 1) / parts); });
        out.n_segs = parts;
        for (int i = 0; i < parts; ++i) {
            ParsedSegment& g = out.seg[i];
            out.max_len = std::max(out.max_len, g.max_len);
            out.n_records += g.n_records;
            out.seq_bytes += g.seq_bytes;
        }
        pos += take;
        return take;
    }
};

// Text that arrives as a stream (inflated gzip): the part of a window behind its last record boundary is carried over
// to the front of the next one.
class CarrySource : public TextSource {
protected:
    std::vector<char> carry;
    bool exhausted = false;
    // Appends up to cap - have fresh bytes at dst + have; returns the new fill level; sets exhausted at the end of the stream.
    virtual size_t fill(char* dst, size_t have, size_t cap) = 0;
public:
    size_t next(char* dst, size_t cap) override {
        if (odd) return 0;
        size_t have = carry.size();
        if (have >= cap) { odd = true; return 0; }
        if (have) std::memcpy(dst, carry.data(), have);
        carry.clear();
        if (!exhausted) have = fill(dst, have, cap - 1);
        if (have == 0) return 0;
        if (exhausted) {
            if (dst[have - 1] != '\n') dst[have++] = '\n';
            return have;
        }
        const size_t cut = find_cut(dst, have);
        if (!cut) { odd = true; return 0; }
        carry.assign(dst + cut, dst + have);
        return cut;
    }
};

// ---- BGZF and other blocked gzip files whose members announce their size (the 'BC' extra subfield of the SAM/BAM
//      specification, written by bgzip): members are located from their headers alone and inflated in parallel,
//      each straight into its place in the window ----
struct BgzfMember {
    size_t off, csize;
    uint32_t isize;
};

// Parses the member header at data[off]; false if it is not a BGZF member.
bool bgzf_member(const char* data, size_t size, size_t off, BgzfMember& m) {
    const unsigned char* p = reinterpret_cast<const unsigned char*>(data) + off;
    if (size - off < 28) return false;
    if (p[0] != 0x1f || p[1] != 0x8b || p[2] != 8 || !(p[3] & 4)) return false;
    const size_t xlen = p[10] | (static_cast<size_t>(p[11]) << 8);
    if (size - off < 12 + xlen + 8) return false;
    size_t x = 12;
    size_t bsize = 0;
    while (x + 4 <= 12 + xlen) {
        const size_t slen = p[x + 2] | (static_cast<size_t>(p[x + 3]) << 8);
        if (p[x] == 'B' && p[x + 1] == 'C' && slen == 2 && x + 6 <= 12 + xlen) {
            bsize = (p[x + 4] | (static_cast<size_t>(p[x + 5]) << 8)) + 1;
            break;
        }
        x += 4 + slen;
    }
    if (!bsize || bsize < 12 + xlen + 8 || off + bsize > size) return false;
    const unsigned char* t = p + bsize - 4;
    m.off = off;
    m.csize = bsize;
    m.isize = t[0] | (static_cast<uint32_t>(t[1]) << 8) | (static_cast<uint32_t>(t[2]) << 16) | (static_cast<uint32_t>(t[3]) << 24);
    return true;
}

// libdeflate (whole-buffer DEFLATE, 2-3 x zlib's inflate) is part of this image as a shared library without headers;
// it is looked up at run time and used for the members it accepts (CRC and size checked by it, all input consumed);
// anything else -- and everything when it is absent or SCG_LIBDEFLATE=0 -- is zlib's, whose verdict and message count.
struct Libdeflate {
    void* (*alloc)() = nullptr;
    void (*release)(void*) = nullptr;
    int (*gzip_ex)(void*, const void*, size_t, void*, size_t, size_t*, size_t*) = nullptr;
    Libdeflate() {
        void* h = ::dlopen("libdeflate.so.0", RTLD_NOW | RTLD_LOCAL);
        if (!h) return;
        alloc = reinterpret_cast<void* (*)()>(::dlsym(h, "libdeflate_alloc_decompressor"));
        release = reinterpret_cast<void (*)(void*)>(::dlsym(h, "libdeflate_free_decompressor"));
        gzip_ex = reinterpret_cast<int (*)(void*, const void*, size_t, void*, size_t, size_t*, size_t*)>(::dlsym(h, "libdeflate_gzip_decompress_ex"));
        if (!alloc || !release || !gzip_ex) alloc = nullptr;
    }
    bool usable() const {
        const char* e = std::getenv("SCG_LIBDEFLATE");
        return alloc != nullptr && !(e && *e == '0');
    }
};
const Libdeflate& libdeflate() {
    static const Libdeflate* L = new Libdeflate;      // (never unloaded)
    return *L;
}
struct ThreadDecompressor {
    void* d = nullptr;
    ~ThreadDecompressor() { if (d) libdeflate().release(d); }
};

void inflate_member(const char* src, size_t csize, char* dst, uint32_t isize) {
    const Libdeflate& L = libdeflate();
    if (L.usable()) {
        thread_local ThreadDecompressor td;
        if (!td.d) td.d = L.alloc();
        if (td.d) {
            size_t used = 0, made = 0;
            if (L.gzip_ex(td.d, src, csize, dst, isize, &used, &made) == 0 && used == csize && made == isize) return;
        }
    }
    z_stream zs;
    std::memset(&zs, 0, sizeof(zs));
    if (inflateInit2(&zs, 15 + 16) != Z_OK) throw Error(SCG_ERR_IO, "failed to initialise zlib");
    zs.next_in = reinterpret_cast<Bytef*>(const_cast<char*>(src));
    zs.avail_in = static_cast<uInt>(csize);
    zs.next_out = reinterpret_cast<Bytef*>(dst);
    zs.avail_out = isize;
    const int rc = inflate(&zs, Z_FINISH);
    const bool ok = rc == Z_STREAM_END && zs.total_out == isize && zs.avail_in == 0;
    std::string msg = zs.msg ? zs.msg : "corrupt gzip member";
    inflateEnd(&zs);
    if (!ok) throw Error(SCG_ERR_IO, msg);      // byteme/GzipFileReader.hpp:45-48 reports zlib's message as well
}

class BgzfSource : public CarrySource {
    MappedFile f;
    size_t off = 0;
    WorkerPool pool;
public:
    BgzfSource(const char* path, int nthreads) : f(path), pool(nthreads) { threads = nthreads; }
    const char* kind() const override { return "bgzf"; }
    uint64_t size_hint() const override { return static_cast<uint64_t>(f.size - off) * 6 + carry.size(); }
    static bool looks_like(const char* path) {
        MappedFile g(path);
        BgzfMember m;
        return g.size && bgzf_member(g.data, g.size, 0, m);
    }
    bool has_members() const override { return true; }
    size_t next_members(char* staging, size_t cap, size_t slack, size_t cap_text, std::vector<CompressedMember>& members, size_t& text_bytes,
                        bool& last) override {
        members.clear();
        text_bytes = 0;
        last = false;
        if (odd || off >= f.size) return 0;
        {
            // The walk below touches one header every ~14 KB of a mapping whose pages have no entries yet: a page
            // fault every few members, 20-40 ms per GB on the calling thread.  The threads fill the entries of the
            // stretch the window can take in one go each (MADV_POPULATE_READ, Linux 5.14+; older kernels: ignored).
            const size_t span = std::min(cap, f.size - off);
            const int parts = static_cast<int>(std::min<size_t>(static_cast<size_t>(threads), (span >> 22) + 1));
            const char* base = f.data + off;
            pool.run(parts, [&](int i) {
                const uintptr_t page = 4096;
                const uintptr_t lo = reinterpret_cast<uintptr_t>(base + span * i / parts) & ~(page - 1);
                const uintptr_t hi = reinterpret_cast<uintptr_t>(base + span * (i + 1) / parts);
                if (hi > lo) (void)::madvise(reinterpret_cast<void*>(lo), hi - lo, 22 /* MADV_POPULATE_READ */);
            });
        }
        std::vector<size_t> from;          // payload offsets within the file
        size_t in = 0, out = 0;
        const size_t window_start = off;
        while (off < f.size) {
            BgzfMember m;
            if (!bgzf_member(f.data, f.size, off, m)) { odd = true; return 0; }
            const unsigned char* p = reinterpret_cast<const unsigned char*>(f.data) + off;
            const size_t xlen = p[10] | (static_cast<size_t>(p[11]) << 8);
            // FEXTRA only (what bgzip writes): names, comments and header CRCs are for zlib to judge
            if (p[3] != 4 || m.csize < 12 + xlen + 8) { odd = true; return 0; }
            const size_t payload = m.csize - 12 - xlen - 8;
            if (in + payload + slack > cap || out + m.isize > cap_text) break;
            const unsigned char* t = p + m.csize - 8;
            CompressedMember c;
            c.in_off = static_cast<uint32_t>(in);
            c.in_len = static_cast<uint32_t>(payload);
            c.out_off = static_cast<uint32_t>(out);
            c.out_len = m.isize;
            c.crc = t[0] | (static_cast<uint32_t>(t[1]) << 8) | (static_cast<uint32_t>(t[2]) << 16) | (static_cast<uint32_t>(t[3]) << 24);
            members.push_back(c);
            from.push_back(off + 12 + xlen);
            in += payload;
            out += m.isize;
            off += m.csize;
        }
        if (members.empty()) { odd = true; return 0; }            // one member larger than a window
        last = off >= f.size;
        text_bytes = out;
        // the payloads, back to back (a fifth of the text: the copy is cheap next to inflating it)
        const int n = static_cast<int>(members.size());
        const int parts = std::min(n, threads * 4);
        pool.run(parts, [&](int i) {
            const int k0 = static_cast<int>(static_cast<int64_t>(n) * i / parts), k1 = static_cast<int>(static_cast<int64_t>(n) * (i + 1) / parts);
            for (int k = k0; k < k1; ++k) std::memcpy(staging + members[k].in_off, f.data + from[k], members[k].in_len);
            // these members are not read again: their pages lose their entries here, in parallel, instead of in the final
            // munmap (20 ms per GB, single-threaded, on the caller's way out)
            if (k1 > k0) {
                const size_t a = k0 == 0 ? window_start : from[k0], b = k1 == n ? off : from[k1];
                const uintptr_t page = 4096;
                const uintptr_t lo = (reinterpret_cast<uintptr_t>(f.data + a) + page - 1) & ~(page - 1);
                const uintptr_t hi = reinterpret_cast<uintptr_t>(f.data + b) & ~(page - 1);
                if (hi > lo) (void)::madvise(reinterpret_cast<void*>(lo), hi - lo, MADV_DONTNEED);
            }
        });
        std::memset(staging + in, 0, slack);
        return in + slack;
    }
protected:
    size_t fill(char* dst, size_t have, size_t cap) override {
        std::vector<BgzfMember> batch;
        std::vector<size_t> where;
        size_t at = have;
        while (off < f.size) {
            BgzfMember m;
            if (!bgzf_member(f.data, f.size, off, m)) {
                // a member without the size subfield in the middle of the file: not a format this source understands
                odd = true;
                return have;
            }
            if (at + m.isize > cap) break;
            batch.push_back(m);
            where.push_back(at);
            at += m.isize;
            off += m.csize;
        }
        if (off >= f.size) exhausted = true;
        if (batch.empty() && !exhausted) { odd = true; return have; }     // one member larger than a window
        try {
            pool.run(static_cast<int>(batch.size()), [&](int i) {
                if (batch[i].isize) inflate_member(f.data + batch[i].off, batch[i].csize, dst + where[i], batch[i].isize);
            });
        } catch (const Error&) {
            // a corrupt member: the sequential reader reports it the way the reference does -- zlib's message with the
            // path in front, and only after whatever the text before it holds
            odd = true;
            return have;
        }
        return at;
    }
};

// ---- any other gzip file: one inflate stream (zlib's gzread handles concatenated members like the reference's
//      byteme::GzipFileReader), bounded by that one thread ----
class GzipSource : public CarrySource {
    gzFile gz = nullptr;
    uint64_t csize = 0;
public:
    explicit GzipSource(const char* path) {
        struct stat st;
        if (::stat(path, &st) == 0) csize = static_cast<uint64_t>(st.st_size);
        gz = gzopen(path, "rb");
        if (!gz) throw Error(SCG_ERR_IO, std::string("failed to open file at '") + path + "'");
        gzbuffer(gz, 1 << 20);
    }
    ~GzipSource() override { if (gz) gzclose(gz); }
    const char* kind() const override { return "gzip"; }
    uint64_t size_hint() const override { return csize * 8; }
protected:
    size_t fill(char* dst, size_t have, size_t cap) override {
        while (have < cap) {
            const size_t want = std::min<size_t>(cap - have, size_t(1) << 30);
            const int got = gzread(gz, dst + have, static_cast<unsigned>(want));
            if (got < 0) {
                int dummy;
                throw Error(SCG_ERR_IO, gzerror(gz, &dummy));      // byteme/GzipFileReader.hpp:47
            }
            if (got == 0) { exhausted = true; break; }
            have += static_cast<size_t>(got);
        }
        return have;
    }
};

// ---- any other gzip file, when it is small enough to be inflated whole: libdeflate needs the complete member and the
//      complete output buffer, but runs at three times zlib's speed; the text then lies in (anonymous, lazily
//      committed) memory and is treated like a mapped plain file, record scan by the host threads included.  The limit
//      ($SCG_GZIP_WHOLE_GB, default 8, 0 = never) bounds the memory this takes; larger files, files libdeflate
//      declines, and files with anything but whole gzip members in them are streamed through zlib as before ----
std::unique_ptr<MappedFile> inflate_whole(const char* path) {
    const Libdeflate& L = libdeflate();
    if (!L.usable()) return nullptr;
    size_t limit = size_t(8) << 30;
    if (const char* e = std::getenv("SCG_GZIP_WHOLE_GB")) limit = static_cast<size_t>(std::max(0.0, std::atof(e)) * double(size_t(1) << 30));
    MappedFile in(path);
    // (FASTQ rarely compresses by less than three: a file that large would hit the limit after seconds of wasted work)
    if (limit == 0 || in.size < 18 || in.size > limit / 3) return nullptr;
    // address space is free: reserve what the most compressible FASTQ could need, commit what is written
    const size_t cap = std::min(limit, in.size * 24 + (size_t(1) << 20)) + 4096;
    void* m = ::mmap(nullptr, cap, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (m == MAP_FAILED) return nullptr;
    void* d = L.alloc();
    size_t at = 0, made = 0;
    bool ok = d != nullptr;
    while (ok && at < in.size) {
        size_t used = 0, got = 0;
        const unsigned char* p = reinterpret_cast<const unsigned char*>(in.data) + at;
        ok = in.size - at >= 18 && p[0] == 0x1f && p[1] == 0x8b &&
             L.gzip_ex(d, p, in.size - at, static_cast<char*>(m) + made, cap - 4096 - made, &used, &got) == 0 && used > 0;
        at += used;
        made += got;
    }
    if (d) L.release(d);
    if (!ok) {                     // (too large, corrupt, trailing bytes ...: zlib's stream decides)
        ::munmap(m, cap);
        return nullptr;
    }
    return std::unique_ptr<MappedFile>(new MappedFile(static_cast<char*>(m), cap, made));
}

// ---- any other gzip file of some size: decoded by all host threads at once (scg_pgzip.h: chunks decoded
//      speculatively with an unknown window, stitched in order, checked against every member's CRC-32 and length), the
//      text written straight into the pinned windows the device scans.  Whatever the decoder hands back -- a corrupt
//      stream, a header CRC, a ratio beyond its buffers -- sets unusual(): the caller redoes the file with the
//      streams above, whose verdict is zlib's ----
class PgzipSource : public CarrySource {
    int fd = -1;
    uint8_t* map = nullptr;
    size_t mapped = 0, csize = 0;
    std::unique_ptr<ParallelGunzip> pg;
    uint64_t produced = 0;
public:
    PgzipSource(const char* path, int nthreads) {
        threads = nthreads;
        fd = ::open(path, O_RDONLY);
        if (fd < 0) throw Error(SCG_ERR_IO, std::string("failed to open file at '") + path + "'");
        struct stat st;
        if (::fstat(fd, &st) != 0) { ::close(fd); throw Error(SCG_ERR_IO, "failed to stat the FASTQ file"); }
        csize = static_cast<size_t>(st.st_size);
        // the file, followed by a page of zeros: the decoder's bit reader looks a few bytes beyond the last one
        mapped = ((csize + 4095) & ~size_t(4095)) + 4096;
        void* m = ::mmap(nullptr, mapped, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (m == MAP_FAILED) { ::close(fd); throw Error(SCG_ERR_IO, "failed to map the FASTQ file"); }
        if (csize && ::mmap(m, csize, PROT_READ, MAP_PRIVATE | MAP_FIXED, fd, 0) == MAP_FAILED) {
            ::munmap(m, mapped); ::close(fd);
            throw Error(SCG_ERR_IO, "failed to map the FASTQ file");
        }
        map = static_cast<uint8_t*>(m);
        pg.reset(new ParallelGunzip(map, csize, nthreads));
    }
    ~PgzipSource() override {
        pg.reset();
        if (map) ::munmap(map, mapped);
        if (fd >= 0) ::close(fd);
    }
    static bool wanted(const char* path, int nthreads) {
        const char* e = std::getenv("SCG_PGZIP");             // test hook: 0 keeps the single inflate streams
        if (e && *e == '0') return false;
        struct stat st;
        if (::stat(path, &st) != 0) return false;
        return ParallelGunzip::chunk_size_for(static_cast<size_t>(st.st_size), nthreads) != 0;
    }
    const char* kind() const override { return "gzip-parallel"; }
    uint64_t size_hint() const override { return static_cast<uint64_t>(csize) * 8 > produced ? static_cast<uint64_t>(csize) * 8 - produced : 4096; }
protected:
    size_t fill(char* dst, size_t have, size_t cap) override {
        const size_t got = pg->read(dst + have, cap - have);
        if (pg->failed()) { odd = true; return have; }
        if (got < cap - have) exhausted = true;
        produced += got;
        return have + got;
    }
};

} // namespace

// A gzip file that is not BGZF and large enough for the chunked decoders (the host threads', or the device's).
bool TextSource::ordinary_gzip(const char* path, int threads) {
    unsigned char h[2] = {0, 0};
    FILE* fp = std::fopen(path, "rb");
    if (!fp) return false;
    const size_t got = std::fread(h, 1, 2, fp);
    std::fclose(fp);
    if (got != 2 || h[0] != 0x1f || h[1] != 0x8b) return false;
    return !BgzfSource::looks_like(path) && PgzipSource::wanted(path, threads);
}

std::unique_ptr<TextSource> TextSource::open(const char* path, int threads, bool parallel_gzip, int gzip_threads) {
    if (gzip_threads <= 0) gzip_threads = threads;
    unsigned char h[2] = {0, 0};
    size_t got = 0;
    {
        FILE* fp = std::fopen(path, "rb");
        if (!fp) throw Error(SCG_ERR_IO, std::string("failed to open file at '") + path + "'");
        got = std::fread(h, 1, 2, fp);
        std::fclose(fp);
    }
    const bool gz = got == 2 && h[0] == 0x1f && h[1] == 0x8b;      // byteme/magic_numbers.hpp:19-22
    if (!gz) return std::unique_ptr<TextSource>(new PlainSource(path, threads));
    if (BgzfSource::looks_like(path)) return std::unique_ptr<TextSource>(new BgzfSource(path, threads));
    if (parallel_gzip && PgzipSource::wanted(path, gzip_threads)) return std::unique_ptr<TextSource>(new PgzipSource(path, gzip_threads));
    if (std::unique_ptr<MappedFile> text = inflate_whole(path)) {
        return std::unique_ptr<TextSource>(new PlainSource(std::move(*text), threads, "gzip"));
    }
    return std::unique_ptr<TextSource>(new GzipSource(path));
}

} // namespace scg
