// scg_fastq.cpp -- host FASTQ staging: file -> batches of concatenated sequence bytes + offsets.
//
// Replaces kaori::FastqReader (inst/include/kaori/FastqReader.hpp:42-110) over
// byteme::SomeFileReader (inst/include/byteme/SomeFileReader.hpp:31-44).  Same record grammar,
// including its quirks:
//   - a record starts with '@'; the name line runs to the first '\n';
//   - the sequence is every byte up to the first '+', newlines removed (multi-line allowed,
//     '\r' is kept as a base);
//   - the '+' line is skipped; qualities are consumed until a newline is met with at least as
//     many quality bytes as sequence bytes (or EOF), and the two lengths must then agree;
//   - the last record may lack its trailing newline; an empty file holds zero reads;
//   - "line numbers" in messages advance by exactly four per record, as the reference counts them.
// Unlike the reference's byte-at-a-time virtual-call loop this scanner works on large buffers
// with memchr, and only the sequence bytes are kept (names and qualities never leave this file).
#include "scg_host.h"
#include "scg_ingest.h"

#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <thread>
#include <unistd.h>
#include <zlib.h>

namespace scg {

struct FastqStream::Impl {
    std::string path;
    bool gz = false;
    gzFile gzf = nullptr;
    int fd = -1;
    std::vector<unsigned char> buf;
    const unsigned char* cur = nullptr;
    const unsigned char* end = nullptr;
    bool eof = false;
    bool okay = false;       // a record starts at *cur
    int64_t line_count = 0;

    bool started = false;

    // Opens the file; the first read happens with the first record (the reference opens its reader before it builds
    // the handler and reads only once processing starts: src/count_single_barcodes.cpp:30-49).
    explicit Impl(const char* p) : path(p) {
        unsigned char header[3];
        size_t got = 0;
        {
            FILE* f = std::fopen(p, "rb");
            if (!f) {
                // byteme/SelfClosingFILE.hpp / SelfClosingGzFile.hpp:13
                throw Error(SCG_ERR_IO, "failed to open file at '" + path + "'");
            }
            got = std::fread(header, 1, 3, f);
            std::fclose(f);
        }
        // byteme/magic_numbers.hpp:19-22
        gz = (got >= 2 && header[0] == 0x1f && header[1] == 0x8b);
        if (gz) {
            gzf = gzopen(p, "rb");
            if (!gzf) throw Error(SCG_ERR_IO, "failed to open file at '" + path + "'");
            gzbuffer(gzf, 1 << 20);
        } else {
            fd = ::open(p, O_RDONLY);
            if (fd < 0) throw Error(SCG_ERR_IO, "failed to open file at '" + path + "'");
        }
    }

    void start() {
        started = true;
        buf.resize(size_t(8) << 20);
        refill();
        okay = cur < end;    // kaori/FastqReader.hpp:32
    }

    ~Impl() {
        if (gzf) gzclose(gzf);
        if (fd >= 0) ::close(fd);
    }

    bool refill() {
        if (eof) { cur = end = buf.data(); return false; }
        long got;
        if (gz) {
            got = gzread(gzf, buf.data(), static_cast<unsigned>(buf.size()));
            if (got < 0) {
                int dummy;
                throw Error(SCG_ERR_IO, gzerror(gzf, &dummy));   // byteme/GzipFileReader.hpp:47
            }
        } else {
            got = ::read(fd, buf.data(), buf.size());
            if (got < 0) throw Error(SCG_ERR_IO, "failed to read raw binary file");   // byteme/RawFileReader.hpp:60
        }
        cur = buf.data();
        end = cur + got;
        if (got == 0) { eof = true; return false; }
        return true;
    }

    // Moves past the current byte; false at end of input (PerByte::advance).
    bool advance() {
        ++cur;
        if (cur < end) return true;
        return refill();
    }

    void advance_or_throw() {
        if (!advance()) {
            throw Error(SCG_ERR_IO, "premature end of the file at line " + std::to_string(line_count + 1));   // FastqReader.hpp:115-120
        }
    }

    // Skips to the next '\n' (which becomes the current byte); throws on EOF.
    void skip_to_newline() {
        for (;;) {
            const void* nl = std::memchr(cur, '\n', static_cast<size_t>(end - cur));
            if (nl) { cur = static_cast<const unsigned char*>(nl); return; }
            cur = end - 1;
            advance_or_throw();
        }
    }

    // Parses one record starting at *cur, appending its sequence to `seqs`.
    void record(std::vector<char>& seqs) {
        int64_t init_line = line_count;
        if (*cur != '@') {
            throw Error(SCG_ERR_IO, "read name should start with '@' (starting line " + std::to_string(init_line + 1) + ")");
        }
        advance_or_throw();
        skip_to_newline();          // name up to the first whitespace, rest of the line ignored
        ++line_count;

        size_t start = seqs.size();
        advance_or_throw();
        for (;;) {                  // sequence: up to the first '+', newlines dropped
            size_t avail = static_cast<size_t>(end - cur);
            const unsigned char* nl = static_cast<const unsigned char*>(std::memchr(cur, '\n', avail));
            size_t span = nl ? static_cast<size_t>(nl - cur) : avail;
            const unsigned char* plus = static_cast<const unsigned char*>(std::memchr(cur, '+', span));
            if (plus) {
                seqs.insert(seqs.end(), cur, plus);
                cur = plus;
                break;
            }
            seqs.insert(seqs.end(), cur, cur + span);
            if (nl) {
                cur = nl;           // on the newline: step over it
            } else {
                cur = end - 1;
            }
            advance_or_throw();
        }
        ++line_count;

        advance_or_throw();         // rest of the '+' line
        skip_to_newline();
        ++line_count;

        size_t seq_len = seqs.size() - start, qual_len = 0;
        okay = false;
        while (advance()) {
            // consume quality bytes up to the next newline in one step where possible
            size_t avail = static_cast<size_t>(end - cur);
            const unsigned char* nl = static_cast<const unsigned char*>(std::memchr(cur, '\n', avail));
            if (!nl) {
                qual_len += avail;
                cur = end - 1;
                continue;
            }
            qual_len += static_cast<size_t>(nl - cur);
            cur = nl;
            if (qual_len >= seq_len) {
                okay = advance();   // sneak past the newline
                break;
            }
        }
        if (qual_len != seq_len) {
            throw Error(SCG_ERR_IO, "non-equal lengths for quality and sequence strings (starting line " + std::to_string(init_line + 1) + ")");
        }
        ++line_count;
    }
};

FastqStream::FastqStream(const char* path) : impl(new Impl(path)) {}

FastqStream::~FastqStream() { delete impl; }

bool FastqStream::next_batch(ReadBatch& out, int64_t max_reads, int64_t max_bytes) {
    out.clear();
    if (!impl->started) impl->start();
    while (impl->okay && out.size() < max_reads && static_cast<int64_t>(out.seqs.size()) < max_bytes) {
        impl->record(out.seqs);
        out.offsets.push_back(out.seqs.size());
    }
    return out.size() > 0;
}

// ---------------------------------------------------------------------------------------------
// ParallelFastq
// ---------------------------------------------------------------------------------------------
// CPUs this process may keep busy: the hardware threads, or less under a cgroup CPU quota (cgroup v2 cpu.max,
// v1 cpu.cfs_quota_us / cpu.cfs_period_us) -- threads beyond the quota are only throttled.
static int cpu_share() {
    unsigned hw = std::thread::hardware_concurrency();
    int n = hw ? static_cast<int>(hw) : 1;
    long quota = -1, period = -1;
    if (FILE* f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char q[32] = {0};
        if (std::fscanf(f, "%31s %ld", q, &period) == 2 && std::strcmp(q, "max") != 0) quota = std::atol(q);
        std::fclose(f);
    } else {
        if (FILE* a = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (std::fscanf(a, "%ld", &quota) != 1) quota = -1; std::fclose(a); }
        if (FILE* b = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (std::fscanf(b, "%ld", &period) != 1) period = -1; std::fclose(b); }
    }
    if (quota > 0 && period > 0) n = std::min<long>(n, std::max<long>(1, (quota + period - 1) / period));
    return n;
}

int default_host_threads(int requested, int devices) {
    // R's num.threads defaults to 1 and is advisory; the stager sizes itself from the machine -- sixteen threads per
    // device the call feeds, within the CPUs the process may use -- unless SCG_HOST_THREADS says otherwise.
    const char* env = std::getenv("SCG_HOST_THREADS");
    if (env && *env) {
        int v = std::atoi(env);
        if (v >= 1) return v;
    }
    static const int share = cpu_share();
    int n = std::min(share, 16 * std::max(1, devices));
    if (n > 64) n = 64;
    if (requested > n) n = requested > 64 ? 64 : requested;
    return n < 1 ? 1 : n;
}

struct ParallelFastq::Impl {
    int fd = -1;
    const char* data = nullptr;
    size_t size = 0;
    size_t pos = 0;          // verified record start (or EOF)
    int nthreads = 1;
    bool odd = false;
    size_t window = 0;

    // If a strict 4-line record starts at p, returns the offset just past it (== size for a final
    // record without trailing newline); otherwise 0.
    size_t record_end(size_t p, const char** seq_out = nullptr, size_t* seq_len_out = nullptr) const {
        return strict_record_end(data, size, p, seq_out, seq_len_out);
    }

    // First position >= from that starts two consecutive strict records (or one ending at EOF).
    size_t find_start(size_t from, size_t limit) const {
        size_t p = from;
        if (p > 0 && data[p - 1] != '\n') {
            const char* nl = static_cast<const char*>(std::memchr(data + p, '\n', size - p));
            if (!nl) return size;
            p = static_cast<size_t>(nl + 1 - data);
        }
        while (p < limit) {
            size_t e1 = record_end(p);
            if (e1 && (e1 == size || record_end(e1))) return p;
            const char* nl = static_cast<const char*>(std::memchr(data + p, '\n', size - p));
            if (!nl) return size;
            p = static_cast<size_t>(nl + 1 - data);
        }
        return p < size ? p : size;
    }
};

bool ParallelFastq::is_plain_file(const char* path) {
    struct stat st;
    if (::stat(path, &st) != 0 || !S_ISREG(st.st_mode) || st.st_size < 4) return false;
    unsigned char h[2] = {0, 0};
    FILE* f = std::fopen(path, "rb");
    if (!f) return false;
    size_t got = std::fread(h, 1, 2, f);
    std::fclose(f);
    return !(got == 2 && h[0] == 0x1f && h[1] == 0x8b);
}

ParallelFastq::ParallelFastq(const char* path, int nthreads) : impl(new Impl) {
    impl->fd = ::open(path, O_RDONLY);
    if (impl->fd < 0) { delete impl; throw Error(SCG_ERR_IO, std::string("failed to open file at '") + path + "'"); }
    struct stat st;
    if (::fstat(impl->fd, &st) != 0) { ::close(impl->fd); delete impl; throw Error(SCG_ERR_IO, "failed to stat the FASTQ file"); }
    impl->size = static_cast<size_t>(st.st_size);
    if (impl->size) {
        void* m = ::mmap(nullptr, impl->size, PROT_READ, MAP_PRIVATE, impl->fd, 0);
        if (m == MAP_FAILED) { ::close(impl->fd); delete impl; throw Error(SCG_ERR_IO, "failed to map the FASTQ file"); }
        ::madvise(m, impl->size, MADV_SEQUENTIAL);
        impl->data = static_cast<const char*>(m);
    }
    impl->nthreads = nthreads < 1 ? 1 : nthreads;
    size_t piece = size_t(32) << 20;
    if (const char* env = std::getenv("SCG_FASTQ_PIECE_KB")) {      // test hook: tiny pieces force many hand-overs
        long kb = std::atol(env);
        if (kb > 0) piece = static_cast<size_t>(kb) << 10;
    }
    impl->window = static_cast<size_t>(impl->nthreads) * piece;
    // the file must open with a strict record for this reader to apply at all
    if (impl->size && !impl->record_end(0)) impl->odd = true;
}

ParallelFastq::~ParallelFastq() {
    if (impl->data) ::munmap(const_cast<char*>(impl->data), impl->size);
    if (impl->fd >= 0) ::close(impl->fd);
    delete impl;
}

bool ParallelFastq::unusual() const { return impl->odd; }

bool ParallelFastq::next_window(std::vector<ReadBatch>& out) {
    Impl& I = *impl;
    out.assign(static_cast<size_t>(I.nthreads), ReadBatch());
    for (auto& b : out) b.clear();
    if (I.odd || I.pos >= I.size) return false;
    const size_t a = I.pos;
    const size_t b = std::min(I.size, a + I.window);
    const int T = I.nthreads;
    const size_t piece = (b - a + T - 1) / T;
    // tentative starts: worker 0 at the verified boundary, the others wherever two records line up
    std::vector<size_t> start(T + 1);
    start[0] = a;
    for (int k = 1; k < T; ++k) start[k] = 0;
    std::vector<size_t> landed(T, 0);
    std::vector<char> bad(T, 0);
    auto work = [&](int k) {
        size_t lo = (k == 0) ? a : I.find_start(std::min(b, a + k * piece), b);
        start[k] = lo;
        // parse until the next worker's nominal piece begins (exact hand-over is checked afterwards)
        const size_t nominal_end = (k == T - 1) ? b : std::min(b, a + (k + 1) * piece);
        ReadBatch& rb = out[k];
        size_t p = lo;
        while (p < I.size && p < nominal_end) {
            const char* sq = nullptr;
            size_t sl = 0;
            size_t e = I.record_end(p, &sq, &sl);
            if (!e) break;
            if (k == T - 1 && e > b && b < I.size) break;       // record crosses the window: next window
            rb.seqs.insert(rb.seqs.end(), sq, sq + sl);         // sequence = line 2
            rb.offsets.push_back(rb.seqs.size());
            p = e;
        }
        landed[k] = p;
        // stopping early inside the piece (other than at EOF) means a record failed the strict test
        if (p < nominal_end && p < I.size && !(k == T - 1)) bad[k] = 1;
        if (k == T - 1 && p < nominal_end && p < I.size) {
            // the last worker may stop before b only because the next record crosses the window
            size_t e = I.record_end(p);
            if (!e || e <= b) bad[k] = 1;
        }
    };
    std::vector<std::thread> th;
    for (int k = 1; k < T; ++k) th.emplace_back(work, k);
    work(0);
    for (auto& t : th) t.join();
    // hand-over check: worker k must land exactly where worker k+1 started
    for (int k = 0; k < T; ++k) {
        if (bad[k]) { I.odd = true; break; }
        if (k + 1 < T && landed[k] != start[k + 1]) { I.odd = true; break; }
    }
    if (I.odd) {
        for (auto& rb : out) rb.clear();
        return false;
    }
    if (landed[T - 1] == a) {
        // no progress: a single record larger than the window, or an unparseable tail
        I.odd = true;
        for (auto& rb : out) rb.clear();
        return false;
    }
    I.pos = landed[T - 1];
    return true;
}

} // namespace scg
