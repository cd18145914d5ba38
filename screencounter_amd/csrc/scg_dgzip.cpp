// scg_dgzip.cpp -- an ordinary gzip file decoded by the DEVICE: the host side of the gunzip_* kernels (scg_inflate.hip).
//
// What a sequencer or `gzip` writes is one DEFLATE stream without entry points; scg_pgzip.h explains the two-stage scheme
// that breaks the chain, and decodes it on the host threads (46 Mreads/s on 16 of them).  Here the same scheme runs where
// the BGZF members are inflated: the compressed bytes cross the link as they are, one wavefront per 64 KB chunk finds a
// block start and decodes to the next chunk's start into 16-bit symbols (markers for the unknown 32 KiB in front), the
// host checks that the chunks chain up exactly -- the only thing it has to know about them -- and two kernels turn the
// symbols into text that never leaves HBM; the windows the record scan takes are device-to-device copies.
// Replaces byteme::GzipFileReader (inst/include/byteme/GzipFileReader.hpp:39-51) for the files it accepts; anything
// unusual -- many small members, a header CRC, a stored block at a chunk start, a ratio beyond the symbol buffers, a chunk
// that does not end where the next begins, a CRC-32 or length mismatch -- makes open_on_device() return null, and the
// file goes to the host decoders, whose last resort is zlib itself.
#include <hip/hip_runtime_api.h>
#include <zlib.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include "scg_host.h"
#include "scg_ingest.h"
#include "scg_inflate.h"
#include "scg_textscan.h"

namespace scg {
namespace {

// CRC-32 of a text from the CRCs of its pieces: crc(A B) = crc(A) * x^(8 |B|) + crc(B) over GF(2) modulo the CRC polynomial
// (reflected: bit 31 is x^0) -- zlib's crc32_combine, whose 1.2.11 form squares a 32 x 32 matrix per call; thousands of
// pieces of one length need the power only once.
uint32_t gf2_mult(uint32_t a, uint32_t b) {
    uint32_t p = 0;
    for (int i = 0; i < 32 && a; ++i) {
        if (a & 0x80000000u) p ^= b;
        a <<= 1;
        b = (b & 1u) ? (b >> 1) ^ 0xEDB88320u : b >> 1;
    }
    return p;
}
uint32_t x_pow_bytes(uint64_t len) {                   // x^(8 len)
    uint32_t sq = 0x40000000u, x = 0x80000000u;         // x^1, x^0
    for (int k = 0; k < 3; ++k) sq = gf2_mult(sq, sq);
    for (; len; len >>= 1) {
        if (len & 1u) x = gf2_mult(sq, x);
        sq = gf2_mult(sq, sq);
    }
    return x;
}

struct Declined {};                                     // (thrown inside decode(); never leaves this file)
#define DGZ_CHECK(expr) do { if ((expr) != hipSuccess) throw Declined(); } while (0)

struct DevMem {
    void* p = nullptr;
    ~DevMem() { if (p) (void)hipFree(p); }
    void alloc(size_t n) { DGZ_CHECK(hipMalloc(&p, n)); }
    template<class T> T* as() const { return static_cast<T*>(p); }
};
struct View {                                            // (memory owned by the scratch set)
    void* p;
    template<class T> T* as() const { return static_cast<T*>(p); }
};
struct PinnedMem {
    void* p = nullptr;
    ~PinnedMem() { if (p) (void)hipHostFree(p); }
    void alloc(size_t n) { DGZ_CHECK(hipHostMalloc(&p, n, hipHostMallocDefault)); }
};
struct Mapping {
    int fd = -1;
    const uint8_t* data = nullptr;
    size_t size = 0;
    explicit Mapping(const char* path) {
        fd = ::open(path, O_RDONLY);
        if (fd < 0) return;
        struct stat st;
        if (::fstat(fd, &st) != 0 || st.st_size <= 0) return;
        void* m = ::mmap(nullptr, static_cast<size_t>(st.st_size), PROT_READ, MAP_PRIVATE, fd, 0);
        if (m == MAP_FAILED) return;
        data = static_cast<const uint8_t*>(m);
        size = static_cast<size_t>(st.st_size);
    }
    ~Mapping() {
        if (data) ::munmap(const_cast<uint8_t*>(data), size);
        if (fd >= 0) ::close(fd);
    }
};

bool trace_on() { const char* e = std::getenv("SCG_TRACE"); return e && *e && *e != '0'; }

// The decoder's scratch -- the compressed file and the chunks' symbols in HBM (33 bytes per compressed byte), two pinned
// buffers for the way there -- is kept for the next file: allocating and freeing 13 GB per call cost 20 ms of the 100 a
// 0.4 GB file takes.  One set per process; a second call at the same time allocates its own.  scg_release_buffers() frees.
struct Scratch {
    int device = -1;
    void* d_in = nullptr; size_t in_bytes = 0;
    void* d_syms = nullptr; size_t sym_bytes = 0;
    void* bounce[2] = {nullptr, nullptr};
    bool busy = false;
    void drop() {
        if (device >= 0) {
            int before = 0;
            (void)hipGetDevice(&before);
            (void)hipSetDevice(device);
            if (d_in) (void)hipFree(d_in);
            if (d_syms) (void)hipFree(d_syms);
            for (auto& b : bounce) if (b) (void)hipHostFree(b);
            (void)hipSetDevice(before);
        }
        d_in = d_syms = bounce[0] = bounce[1] = nullptr;
        in_bytes = sym_bytes = 0;
        device = -1;
    }
};
std::mutex scratch_mu;
Scratch scratch_cache;
constexpr size_t BOUNCE_BYTES = size_t(32) << 20;

// The cached set if it is free (grown to the sizes asked for), else a fresh one; `own` says which.
struct ScratchLease {
    Scratch local;
    Scratch* s = nullptr;
    bool cached = false;
    ScratchLease(int device, size_t in_bytes, size_t sym_bytes) {
        {
            std::lock_guard<std::mutex> g(scratch_mu);
            const char* e = std::getenv("SCG_BUFFER_CACHE");
            if (!scratch_cache.busy && !(e && *e == '0')) { scratch_cache.busy = true; cached = true; }
        }
        s = cached ? &scratch_cache : &local;
        try {
            grow(device, in_bytes, sym_bytes);
        } catch (...) {
            release();
            throw;
        }
    }
    void release() {
        if (cached) {
            std::lock_guard<std::mutex> g(scratch_mu);
            scratch_cache.busy = false;
            cached = false;
        } else {
            local.drop();
        }
    }
    void grow(int device, size_t in_bytes, size_t sym_bytes) {
        if (s->device != device) s->drop();
        s->device = device;
        if (s->in_bytes < in_bytes) {
            if (s->d_in) { (void)hipFree(s->d_in); s->d_in = nullptr; s->in_bytes = 0; }
            DGZ_CHECK(hipMalloc(&s->d_in, in_bytes));
            s->in_bytes = in_bytes;
        }
        if (s->sym_bytes < sym_bytes) {
            if (s->d_syms) { (void)hipFree(s->d_syms); s->d_syms = nullptr; s->sym_bytes = 0; }
            DGZ_CHECK(hipMalloc(&s->d_syms, sym_bytes));
            s->sym_bytes = sym_bytes;
        }
        for (auto& b : s->bounce) if (!b) DGZ_CHECK(hipHostMalloc(&b, BOUNCE_BYTES, hipHostMallocDefault));
    }
    ~ScratchLease() { release(); }
};

// The stream is decoded in GROUPS of chunks (512 MB of compressed bytes by default: the symbols of a group take 33 bytes per
// compressed byte while it is decoded), each group like a file of its own except for what it takes over from the group
// before: the bit position its first chunk starts at, and the end of the text before it -- the last 32 KiB (the window its
// markers point into) or more (the records the scan has not taken yet), copied to the front of the new group's text.  The
// CRC-32 and the length are known to be right only when the last group is through: a file that fails them then has been
// counted in part already, and is redone by the host decoders like any other input that turns out unusual late.
class DeviceGunzipSource : public TextSource {
    int dev;
    Mapping f;
    uint64_t member_start = 0, first_byte = 0, stream_end = 0;
    uint32_t tail_group = 32;                    // chunks per group of the tails' scan (scg_inflate.hip, "The tails")
    uint64_t min_member_chunks = 32;             // a member that is not the file's last is decoded here when it has this many chunks
    size_t chunk_bytes = 0;
    uint64_t n_chunks = 0, cap_syms = 0, group_chunks = 0;
    int host_threads = 1;
    bool tr = false;
    // progress through the stream
    uint64_t next_chunk = 0, expect_bit = 0;
    bool ended = false;
    uLong crc_acc = 0;
    uint64_t text_total = 0;
    // the current group's text: [prefix: the end of the text before][the group's own], of which [0, pos) has been handed out
    DevMem text;
    uint64_t text_bytes = 0, pos = 0;
    std::vector<char> tail;        // host copy of a window's last stretch (where the cut is looked for)

public:
    DeviceGunzipSource(const char* path, int device, int threads) : dev(device), f(path), host_threads(threads), tr(trace_on()) {}
    const char* kind() const override { return "gzip-device"; }
    uint64_t size_hint() const override { return (text_bytes - pos) + (ended ? 0 : (stream_end - std::min<uint64_t>(stream_end, first_byte + next_chunk * chunk_bytes)) * 8); }
    bool device_resident() const override { return true; }
    int device() const override { return dev; }

    // The header, the sizes, and the first group.  False: not a file for this decoder.
    bool begin() {
        const bool test_hook = std::getenv("SCG_DGZIP_CHUNK_KB") != nullptr;       // (tiny chunks, tiny files)
        if (!f.data || f.size < (test_hook ? size_t(64) : size_t(2) << 20)) return false;
        chunk_bytes = size_t(64) << 10;          // (MI355X, 8 M reads: 128 KB 131 Mreads/s, 64 KB 139, 32 KB 120)
        if (const char* e = std::getenv("SCG_DGZIP_CHUNK_KB")) { const long kb = std::atol(e); if (kb >= 4) chunk_bytes = static_cast<size_t>(kb) << 10; }
        if (const char* e = std::getenv("SCG_DGZIP_TAIL_GROUP")) tail_group = static_cast<uint32_t>(std::max(1L, std::atol(e)));   // (tests: many groups in small files)
        if (const char* e = std::getenv("SCG_DGZIP_MIN_MEMBER_CHUNKS")) min_member_chunks = static_cast<uint64_t>(std::max(0L, std::atol(e)));   // (tests: members of any size)
        if (!member_at(0)) return false;
        // a chunk decodes from its block start to the next chunk's: up to two chunks of input when a neighbour holds no block start
        // (a DEFLATE block is 30-60 KB of compressed bytes as a rule: small chunks -- the tests' -- mostly hold no block start at all)
        cap_syms = std::max<uint64_t>(chunk_bytes * 16, uint64_t(1) << 20) + 65536;
        uint64_t group_bytes = uint64_t(512) << 20;                     // (its text must stay below 4 GB: the CRC pieces are indexed with 32 bits)
        if (const char* e = std::getenv("SCG_DGZIP_GROUP_KB")) { const long kb = std::atol(e); if (kb >= 4) group_bytes = static_cast<uint64_t>(kb) << 10; }
        group_chunks = std::max<uint64_t>(1, group_bytes / chunk_bytes);
        return next_group();
    }

    // A member's header at byte `at` of the file (RFC 1952: deflate, no header CRC; name / comment / extra fields are
    // skipped): the chunk grid, the chain and the checksums start afresh behind it.  False: not a header this decoder takes.
    bool member_at(uint64_t at0) {
        const uint8_t* p = f.data + at0;
        if (at0 + 18 + 2 > f.size) return false;
        if (p[0] != 0x1f || p[1] != 0x8b || p[2] != 8 || (p[3] & 0xE2)) return false;
        uint64_t at = at0 + 10;
        if (p[3] & 4) { if (at + 2 > f.size) return false; at += 2 + (f.data[at] | (static_cast<uint64_t>(f.data[at + 1]) << 8)); }
        for (int field = 0; field < 2; ++field) {
            if (p[3] & (field == 0 ? 8 : 16)) {
                while (at < f.size && f.data[at]) ++at;
                ++at;
            }
        }
        if (at + 8 + 2 > f.size) return false;
        member_start = at0;
        first_byte = at;
        stream_end = f.size - 8;                  // (where the LAST member's stream ends; an earlier member's end is found by decoding)
        n_chunks = (stream_end - first_byte + chunk_bytes - 1) / chunk_bytes;
        next_chunk = 0;
        expect_bit = first_byte * 8;
        crc_acc = crc32(0L, Z_NULL, 0);
        text_total = 0;
        return true;
    }

    size_t next(char* dst, size_t cap) override {
        bool pad;
        size_t take = window(cap, pad);
        if (!take) return 0;
        if (hipMemcpy(dst, text.as<char>() + pos, take, hipMemcpyDeviceToHost) != hipSuccess) { odd = true; return 0; }
        pos += take;
        if (pad) dst[take++] = '\n';
        return take;
    }
    size_t next_device(char* d_dst, size_t cap, void* stream) override {
        bool pad;
        size_t take = window(cap, pad);
        if (!take) return 0;
        hipStream_t s = static_cast<hipStream_t>(stream);
        if (hipMemcpyAsync(d_dst, text.as<char>() + pos, take, hipMemcpyDeviceToDevice, s) != hipSuccess) { odd = true; return 0; }
        pos += take;
        if (pad) {
            if (hipMemsetAsync(d_dst + take, '\n', 1, s) != hipSuccess) { odd = true; return 0; }
            ++take;
        }
        return take;
    }

private:
    // How much text the next window of at most cap bytes takes (0: none, or odd set); pad: the input ends in it without a newline.
    size_t window(size_t cap, bool& pad) {
        pad = false;
        if (odd) return 0;
        // (the text in HBM is still being copied from by the windows handed out: they are stream-ordered copies, and the
        // next group's decoding synchronises the device before the old text is let go)
        while (text_bytes - pos + 1 <= cap && !ended) {
            if (!next_group()) { odd = true; return 0; }
        }
        if (pos >= text_bytes) return 0;
        const uint64_t left = text_bytes - pos;
        const size_t slack = size_t(1) << 20;
        if (left + 1 <= cap) {                                          // (the stream has ended: the rest)
            char c = 0;
            if (hipMemcpy(&c, text.as<char>() + text_bytes - 1, 1, hipMemcpyDeviceToHost) != hipSuccess) { odd = true; return 0; }
            pad = c != '\n';
            return static_cast<size_t>(left);
        }
        const size_t n = cap - 1;
        const size_t look = std::min(n, 2 * slack);
        tail.resize(look);
        if (hipMemcpy(tail.data(), text.as<char>() + pos + (n - look), look, hipMemcpyDeviceToHost) != hipSuccess) { odd = true; return 0; }
        const size_t cut = find_cut(tail.data(), look, slack);
        if (!cut) { odd = true; return 0; }
        return n - look + cut;
    }

    // Decodes the next group of chunks into a new text buffer (behind what the old one still holds).  False: declined.
    bool next_group() {
        const auto t0 = std::chrono::steady_clock::now();
        auto ms = [&] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
        int before = 0;
        (void)hipGetDevice(&before);
        struct Restore { int d; ~Restore() { (void)hipSetDevice(d); } } restore{before};
        try {
            DGZ_CHECK(hipSetDevice(dev));
            const uint64_t lo = next_chunk, hi = std::min(n_chunks, lo + group_chunks);
            const bool last_group = hi == n_chunks;
            // a few chunks of the next group are searched as well: the first of them with a block start says where this group ends
            const uint64_t extra = last_group ? 0 : std::min<uint64_t>(8, n_chunks - hi);
            const uint32_t n = static_cast<uint32_t>(hi - lo + extra), n_decode = static_cast<uint32_t>(hi - lo);
            const uint64_t a = expect_bit >> 3;                                     // bytes [a, b) of the file go to the device
            const uint64_t b = last_group ? f.size : std::min<uint64_t>(f.size, first_byte + (hi + extra) * chunk_bytes + 8192);
            size_t free_bytes = 0, total_bytes = 0;
            DGZ_CHECK(hipMemGetInfo(&free_bytes, &total_bytes));
            const size_t in_bytes = static_cast<size_t>(b - a) + 64, sym_bytes = cap_syms * n_decode * sizeof(uint16_t);
            ScratchLease lease(dev, in_bytes, sym_bytes);
            if (in_bytes + sym_bytes + (b - a) * 12 > free_bytes + lease.s->in_bytes + lease.s->sym_bytes) return false;
            const View d_in{lease.s->d_in}, d_syms{lease.s->d_syms};
            DevMem d_chunks;
            d_chunks.alloc(sizeof(GunzipChunk) * n);
            // the bytes into HBM through two pinned buffers filled by a few threads each
            {
                const size_t piece = BOUNCE_BYTES;
                hipEvent_t done[2];
                for (int k = 0; k < 2; ++k) DGZ_CHECK(hipEventCreateWithFlags(&done[k], hipEventDisableTiming));
                const int nt = std::max(1, std::min(host_threads, 8));
                uint64_t off = a;
                for (int k = 0; off < b; ++k, off += piece) {
                    const size_t len = static_cast<size_t>(std::min<uint64_t>(piece, b - off));
                    char* const dst = static_cast<char*>(lease.s->bounce[k & 1]);
                    if (k >= 2) DGZ_CHECK(hipEventSynchronize(done[k & 1]));
                    std::vector<std::thread> th;
                    for (int i = 0; i < nt; ++i) {
                        th.emplace_back([&, i] {
                            const size_t x = len * i / nt, y = len * (i + 1) / nt;
                            std::memcpy(dst + x, f.data + off + x, y - x);
                            // (pages that are not read again lose their entries here, in parallel, not in the final munmap;
                            // the stretch the next group starts in, and the header and trailer, stay)
                            const uintptr_t page = 4096;
                            const uintptr_t l = (reinterpret_cast<uintptr_t>(f.data + off + x) + page - 1) & ~(page - 1);
                            const uintptr_t h = reinterpret_cast<uintptr_t>(f.data + off + y) & ~(page - 1);
                            const uint64_t keep_from = last_group ? f.size - 4096 : first_byte + hi * chunk_bytes - 4096;
                            if (h > l && off + x >= 4096 && off + y <= keep_from) (void)::madvise(reinterpret_cast<void*>(l), h - l, MADV_DONTNEED);
                        });
                    }
                    for (auto& t : th) t.join();
                    DGZ_CHECK(hipMemcpyAsync(d_in.as<char>() + (off - a), dst, len, hipMemcpyHostToDevice, nullptr));
                    DGZ_CHECK(hipEventRecord(done[k & 1], nullptr));
                }
                DGZ_CHECK(hipMemsetAsync(d_in.as<char>() + (b - a), 0, 64, nullptr));
                DGZ_CHECK(hipStreamSynchronize(nullptr));
                for (int k = 0; k < 2; ++k) (void)hipEventDestroy(done[k]);
            }
            if (tr) std::fprintf(stderr, "[scg]   gzip on the device: chunks %llu-%llu of %llu (%zu KB each), their bytes in HBM after %.2f ms\n",
                                 (unsigned long long)lo, (unsigned long long)hi, (unsigned long long)n_chunks, chunk_bytes >> 10, ms());
            std::vector<GunzipChunk> chunks(n);
            std::memset(chunks.data(), 0, sizeof(GunzipChunk) * n);
            chunks[0].start_bit = expect_bit;
            DGZ_CHECK(hipMemcpy(d_chunks.p, chunks.data(), sizeof(GunzipChunk) * n, hipMemcpyHostToDevice));
            DGZ_CHECK(launch_gunzip_find(d_in.as<uint8_t>(), a, b, d_chunks.as<GunzipChunk>(), n, lo, first_byte, chunk_bytes, stream_end, nullptr));
            DGZ_CHECK(launch_gunzip_decode(d_in.as<uint8_t>(), a, b, d_chunks.as<GunzipChunk>(), n, n_decode, d_syms.as<uint16_t>(), cap_syms, nullptr));
            DGZ_CHECK(hipMemcpy(chunks.data(), d_chunks.p, sizeof(GunzipChunk) * n, hipMemcpyDeviceToHost));
            if (tr) std::fprintf(stderr, "[scg]   gzip on the device: chunks decoded after %.2f ms\n", ms());
            // the chain: every chunk ends where the next one (that found a block start) begins; the last group's last ends the stream
            const uint64_t keep = text_bytes - pos;                                   // what the scan has not taken yet
            const uint64_t prefix = text_bytes == 0 ? 0 : std::max<uint64_t>(keep, std::min<uint64_t>(text_bytes, 32768));
            std::vector<uint64_t> text_at(n_decode);
            uint64_t expect = expect_bit, made = 0;
            bool final_seen = false;
            uint32_t n_use = 0;                                                        // chunks of this group that belong to the member
            for (uint32_t c = 0; c < n_decode && !final_seen; ++c) {
                text_at[c] = prefix + made;
                n_use = c + 1;
                if (chunks[c].start_bit == ~uint64_t(0)) continue;
                if (chunks[c].status != scginf::INFLATE_OK || chunks[c].start_bit != expect) return false;
                expect = chunks[c].end_bit;
                made += chunks[c].made;
                final_seen = chunks[c].final_block != 0;
            }
            if ((made == 0 && !final_seen) || prefix + made + 64 >= (uint64_t(1) << 32)) return false;   // (the scan's windows and the CRC pieces index the text with 32 bits)
            uint64_t resume = n_chunks;
            uint64_t trailer = 0;
            bool file_ends = false;
            if (final_seen) {
                // the member's trailer; behind it the file ends or the next member begins (what was decoded of this group behind
                // the member's last block, on the old member's grid, is dropped: the next group starts at the new member's header)
                trailer = (expect + 7) >> 3;
                if (trailer + 8 > f.size) return false;
                file_ends = trailer + 8 == f.size;
                // (a file of many small members -- BGZF without its size fields -- would take a group per member: the host's)
                if (!file_ends && trailer - first_byte < min_member_chunks * chunk_bytes) return false;
            } else {
                if (last_group) return false;
                resume = 0;
                for (uint32_t c = n_decode; c < n; ++c) {
                    if (chunks[c].start_bit != ~uint64_t(0)) { resume = lo + c; break; }
                }
                if (!resume || chunks[resume - lo].start_bit != expect) return false;   // (the group did not end where the next begins)
            }
            // the new text: the end of the old one in front, then this group's
            DevMem fresh, d_at, d_status, d_pieces, d_crcs;
            fresh.alloc(prefix + made + 64);
            if (prefix) DGZ_CHECK(hipMemcpy(fresh.p, text.as<char>() + (text_bytes - prefix), prefix, hipMemcpyDeviceToDevice));
            d_at.alloc(sizeof(uint64_t) * n_decode);
            d_status.alloc(sizeof(uint32_t));
            DGZ_CHECK(hipMemcpy(d_at.p, text_at.data(), sizeof(uint64_t) * n_decode, hipMemcpyHostToDevice));
            DGZ_CHECK(hipMemset(d_status.p, 0, sizeof(uint32_t)));
            // the tails' scratch: per group of chunks a map (64 KB), a window (32 KB) and a count
            GunzipTailScratch tails;
            tails.group = tail_group;
            const size_t tail_groups = (n_use + tail_group - 1) / tail_group;
            DevMem d_tails;
            d_tails.alloc(tail_groups * (size_t(65536) + 32768 + 64));
            tails.maps = d_tails.as<uint16_t>();
            tails.wins = d_tails.as<uint8_t>() + tail_groups * 65536;
            tails.avails = reinterpret_cast<uint32_t*>(d_tails.as<uint8_t>() + tail_groups * (size_t(65536) + 32768));
            DGZ_CHECK(launch_gunzip_text(d_syms.as<uint16_t>(), cap_syms, d_chunks.as<GunzipChunk>(), d_at.as<uint64_t>(), n_use, fresh.as<char>(),
                                         prefix - std::min<uint64_t>(prefix, text_total), tails, d_status.as<uint32_t>(), nullptr));
            // CRC-32: pieces of 512 KB on the device (thousands of workgroups: the byte-serial sums hide one another's
            // latency), combined here like zlib's crc32_combine
            const uint64_t piece = uint64_t(512) << 10;
            const uint32_t np = static_cast<uint32_t>((made + piece - 1) / piece);
            std::vector<InflateMember> pieces(np);
            for (uint32_t i = 0; i < np; ++i) {
                pieces[i].in_off = pieces[i].in_len = 0;
                pieces[i].out_off = static_cast<uint32_t>(prefix + piece * i);
                pieces[i].out_len = static_cast<uint32_t>(std::min<uint64_t>(piece, made - piece * i));
                pieces[i].crc = 0;
            }
            d_pieces.alloc(sizeof(InflateMember) * np);
            d_crcs.alloc(sizeof(uint32_t) * np);
            DGZ_CHECK(hipMemcpy(d_pieces.p, pieces.data(), sizeof(InflateMember) * np, hipMemcpyHostToDevice));
            DGZ_CHECK(launch_crc_pieces(fresh.as<char>(), d_pieces.as<InflateMember>(), np, d_crcs.as<uint32_t>(), nullptr));
            std::vector<uint32_t> crcs(np);
            uint32_t status = 0;
            DGZ_CHECK(hipMemcpy(crcs.data(), d_crcs.p, sizeof(uint32_t) * np, hipMemcpyDeviceToHost));
            DGZ_CHECK(hipMemcpy(&status, d_status.p, sizeof(uint32_t), hipMemcpyDeviceToHost));
            if (status) return false;
            const uint32_t whole_piece = x_pow_bytes(piece);
            for (uint32_t i = 0; i < np; ++i) {
                const uint32_t shift = pieces[i].out_len == piece ? whole_piece : x_pow_bytes(pieces[i].out_len);
                crc_acc = gf2_mult(shift, static_cast<uint32_t>(crc_acc)) ^ crcs[i];
            }
            text_total += made;
            if (final_seen) {
                const uint8_t* t = f.data + trailer;
                const uint32_t want_crc = t[0] | (uint32_t(t[1]) << 8) | (uint32_t(t[2]) << 16) | (uint32_t(t[3]) << 24);
                const uint32_t want_size = t[4] | (uint32_t(t[5]) << 8) | (uint32_t(t[6]) << 16) | (uint32_t(t[7]) << 24);
                if (static_cast<uint32_t>(crc_acc) != want_crc || static_cast<uint32_t>(text_total) != want_size) return false;
            }
            // (every copy out of the old text was enqueued before this point and the device has been synchronised since)
            DGZ_CHECK(hipDeviceSynchronize());
            if (text.p) { (void)hipFree(text.p); text.p = nullptr; }
            text.p = fresh.p; fresh.p = nullptr;
            pos = prefix - keep;
            text_bytes = prefix + made;
            ended = final_seen && file_ends;
            next_chunk = resume;
            expect_bit = expect;
            if (final_seen && !file_ends && !member_at(trailer + 8)) return false;      // (trailing bytes that are no member: zlib's to judge)
            if (tr) std::fprintf(stderr, "[scg]   gzip on the device: %.2f GB of text in HBM%s after %.2f ms\n", made / 1e9, ended ? ", CRC-32 checked," : "", ms());
            return true;
        } catch (const Declined&) {
            return false;
        }
    }
};

} // namespace

void release_device_gunzip_scratch() {
    std::lock_guard<std::mutex> g(scratch_mu);
    if (!scratch_cache.busy) scratch_cache.drop();
}

std::unique_ptr<TextSource> TextSource::open_on_device(const char* path, int device, int threads) {
    const char* e = std::getenv("SCG_DEVICE_GUNZIP");
    if (e && *e == '0') return nullptr;
    std::unique_ptr<DeviceGunzipSource> s(new DeviceGunzipSource(path, device, threads));
    if (!s->begin()) return nullptr;
    return std::unique_ptr<TextSource>(s.release());
}

} // namespace scg
