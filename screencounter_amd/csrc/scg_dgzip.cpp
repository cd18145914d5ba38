// scg_dgzip.cpp -- an ordinary gzip file decoded by the DEVICE: the host side of the gunzip_* kernels (scg_inflate.hip).
//
// What a sequencer or `gzip` writes is one DEFLATE stream without entry points; scg_pgzip.h explains the two-stage scheme
// that breaks the chain, and decodes it on the host threads (46 Mreads/s on 16 of them).  Here the same scheme runs where
// the BGZF members are inflated: the compressed bytes cross the link as they are, one wavefront per 128 KB chunk finds a
// block start and decodes to the next chunk's start into 16-bit symbols (markers for the unknown 32 KiB in front), the
// host checks that the chunks chain up exactly -- the only thing it has to know about them -- and two kernels turn the
// symbols into text that never leaves HBM; the windows the record scan takes are device-to-device copies.
// Replaces byteme::GzipFileReader (inst/include/byteme/GzipFileReader.hpp:39-51) for the files it accepts; anything
// unusual -- several members, a header CRC, a stored block at a chunk start, a ratio beyond the symbol buffers, a chunk
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

class DeviceGunzipSource : public TextSource {
    int dev;
    DevMem text;                   // the whole text (+ 64 bytes)
    uint64_t total = 0, pos = 0;
    std::vector<char> tail;        // host copy of a window's last stretch (where the cut is looked for)
public:
    DeviceGunzipSource(int device, DevMem&& t, uint64_t n) : dev(device), total(n) { text.p = t.p; t.p = nullptr; }
    const char* kind() const override { return "gzip-device"; }
    uint64_t size_hint() const override { return total - pos; }
    bool device_resident() const override { return true; }
    int device() const override { return dev; }

    // How much text the next window of at most cap bytes takes; `last`: the input ends with it.
    size_t window(size_t cap, bool& last, bool& pad) {
        last = pad = false;
        if (odd || pos >= total) return 0;
        const uint64_t left = total - pos;
        const size_t slack = size_t(1) << 20;
        if (left + 1 <= cap) {
            char c = 0;
            if (hipMemcpy(&c, text.as<char>() + total - 1, 1, hipMemcpyDeviceToHost) != hipSuccess) { odd = true; return 0; }
            last = true;
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
    size_t next(char* dst, size_t cap) override {
        bool last, pad;
        size_t take = window(cap, last, pad);
        if (!take) return 0;
        if (hipMemcpy(dst, text.as<char>() + pos, take, hipMemcpyDeviceToHost) != hipSuccess) { odd = true; return 0; }
        pos += take;
        if (pad) dst[take++] = '\n';
        return take;
    }
    size_t next_device(char* d_dst, size_t cap, void* stream) override {
        bool last, pad;
        size_t take = window(cap, last, pad);
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
};

std::unique_ptr<TextSource> decode(const char* path, int device, int threads) {
    const bool tr = trace_on();
    const auto t0 = std::chrono::steady_clock::now();
    auto ms = [&] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
    Mapping f(path);
    const bool test_hook = std::getenv("SCG_DGZIP_CHUNK_KB") != nullptr;       // (tiny chunks, tiny files)
    if (!f.data || f.size < (test_hook ? size_t(64) : size_t(2) << 20) || f.size > (size_t(2) << 30)) return nullptr;
    const uint8_t* p = f.data;
    // RFC 1952 header: deflate, no header CRC; name / comment / extra fields are skipped
    if (p[0] != 0x1f || p[1] != 0x8b || p[2] != 8 || (p[3] & 0xE2)) return nullptr;
    size_t at = 10;
    if (p[3] & 4) { if (at + 2 > f.size) return nullptr; at += 2 + (p[at] | (static_cast<size_t>(p[at + 1]) << 8)); }
    for (int field = 0; field < 2; ++field) {
        if (p[3] & (field == 0 ? 8 : 16)) {
            while (at < f.size && p[at]) ++at;
            ++at;
        }
    }
    if (at + 8 + 2 >= f.size) return nullptr;
    const uint64_t first_byte = at, stream_end = f.size - 8;
    const uint8_t* t = p + stream_end;
    const uint32_t want_crc = t[0] | (uint32_t(t[1]) << 8) | (uint32_t(t[2]) << 16) | (uint32_t(t[3]) << 24);
    const uint32_t want_size = t[4] | (uint32_t(t[5]) << 8) | (uint32_t(t[6]) << 16) | (uint32_t(t[7]) << 24);

    size_t chunk_bytes = size_t(128) << 10;
    if (const char* e = std::getenv("SCG_DGZIP_CHUNK_KB")) { const long kb = std::atol(e); if (kb >= 4) chunk_bytes = static_cast<size_t>(kb) << 10; }
    const uint32_t n = static_cast<uint32_t>((stream_end - first_byte + chunk_bytes - 1) / chunk_bytes);
    // a chunk decodes from its block start to the next chunk's: up to two chunks of input when a neighbour holds no block start
    // (a DEFLATE block is 30-60 KB of compressed bytes as a rule: small chunks -- the tests' -- mostly hold no block start at all)
    const uint64_t cap_syms = std::max<uint64_t>(chunk_bytes * 16, uint64_t(1) << 20) + 65536;

    int before = 0;
    (void)hipGetDevice(&before);
    struct Restore { int d; ~Restore() { (void)hipSetDevice(d); } } restore{before};
    try {
        DGZ_CHECK(hipSetDevice(device));
        // (not more than the card has to spare: the symbols take 32 bytes per compressed byte)
        size_t free_bytes = 0, total_bytes = 0;
        DGZ_CHECK(hipMemGetInfo(&free_bytes, &total_bytes));
        const size_t need = f.size + 64 + cap_syms * n * sizeof(uint16_t) + f.size * 12;
        ScratchLease lease(device, f.size + 64, cap_syms * n * sizeof(uint16_t));
        if (need > free_bytes + lease.s->in_bytes + lease.s->sym_bytes) return nullptr;
        const View d_in{lease.s->d_in}, d_syms{lease.s->d_syms};
        DevMem d_chunks;
        d_chunks.alloc(sizeof(GunzipChunk) * n);
        // the file into HBM through two pinned buffers filled by a few threads each
        {
            const size_t piece = BOUNCE_BYTES;
            struct { void* p; } bounce[2] = {{lease.s->bounce[0]}, {lease.s->bounce[1]}};
            hipEvent_t done[2];
            for (int k = 0; k < 2; ++k) DGZ_CHECK(hipEventCreateWithFlags(&done[k], hipEventDisableTiming));
            const int nt = std::max(1, std::min(threads, 8));
            size_t off = 0;
            for (int k = 0; off < f.size; ++k, off += piece) {
                const size_t len = std::min(piece, f.size - off);
                const int b = k & 1;
                if (k >= 2) DGZ_CHECK(hipEventSynchronize(done[b]));
                std::vector<std::thread> th;
                for (int i = 0; i < nt; ++i) {
                    th.emplace_back([&, i] {
                        const size_t a = len * i / nt, e = len * (i + 1) / nt;
                        std::memcpy(static_cast<char*>(bounce[b].p) + a, f.data + off + a, e - a);
                        // (the pages are not read again: their entries go here, in parallel, not in the final munmap)
                        const uintptr_t page = 4096;
                        const uintptr_t lo = (reinterpret_cast<uintptr_t>(f.data + off + a) + page - 1) & ~(page - 1);
                        const uintptr_t hi = reinterpret_cast<uintptr_t>(f.data + off + e) & ~(page - 1);
                        if (hi > lo && off + a >= 4096 && off + e + 4096 <= f.size) (void)::madvise(reinterpret_cast<void*>(lo), hi - lo, MADV_DONTNEED);
                    });
                }
                for (auto& x : th) x.join();
                DGZ_CHECK(hipMemcpyAsync(d_in.as<char>() + off, bounce[b].p, len, hipMemcpyHostToDevice, nullptr));
                DGZ_CHECK(hipEventRecord(done[b], nullptr));
            }
            DGZ_CHECK(hipMemsetAsync(d_in.as<char>() + f.size, 0, 64, nullptr));
            DGZ_CHECK(hipStreamSynchronize(nullptr));
            for (int k = 0; k < 2; ++k) (void)hipEventDestroy(done[k]);
        }
        if (tr) std::fprintf(stderr, "[scg]   gzip on the device: %u chunks of %zu KB, file in HBM after %.2f ms\n", n, chunk_bytes >> 10, ms());
        std::vector<GunzipChunk> chunks(n);
        std::memset(chunks.data(), 0, sizeof(GunzipChunk) * n);
        chunks[0].start_bit = first_byte * 8;
        DGZ_CHECK(hipMemcpy(d_chunks.p, chunks.data(), sizeof(GunzipChunk) * n, hipMemcpyHostToDevice));
        DGZ_CHECK(launch_gunzip_find(d_in.as<uint8_t>(), f.size, d_chunks.as<GunzipChunk>(), n, first_byte, chunk_bytes, stream_end, nullptr));
        DGZ_CHECK(launch_gunzip_decode(d_in.as<uint8_t>(), f.size, d_chunks.as<GunzipChunk>(), n, d_syms.as<uint16_t>(), cap_syms, nullptr));
        DGZ_CHECK(hipMemcpy(chunks.data(), d_chunks.p, sizeof(GunzipChunk) * n, hipMemcpyDeviceToHost));
        if (tr) std::fprintf(stderr, "[scg]   gzip on the device: chunks decoded after %.2f ms\n", ms());
        // the chain: every chunk ends where the next one (that found a block start) begins, the last one ends the stream
        std::vector<uint64_t> text_at(n);
        uint64_t expect = first_byte * 8, total = 0;
        bool ended = false;
        for (uint32_t c = 0; c < n; ++c) {
            text_at[c] = total;
            if (chunks[c].start_bit == ~uint64_t(0)) continue;
            if (ended || chunks[c].status != scginf::INFLATE_OK || chunks[c].start_bit != expect) return nullptr;
            expect = chunks[c].end_bit;
            total += chunks[c].made;
            ended = chunks[c].final_block != 0;
        }
        if (!ended || ((expect + 7) >> 3) != stream_end) return nullptr;            // (more members, or trailing bytes: not for this decoder)
        if (static_cast<uint32_t>(total) != want_size || total == 0 || total >= (uint64_t(1) << 32) - (uint64_t(1) << 26)) return nullptr;
        DevMem d_text, d_at, d_status, d_pieces, d_crcs;
        d_text.alloc(total + 64);
        d_at.alloc(sizeof(uint64_t) * n);
        d_status.alloc(sizeof(uint32_t));
        if (tr) { DGZ_CHECK(hipDeviceSynchronize()); std::fprintf(stderr, "[scg]   gzip on the device: text buffer allocated after %.2f ms\n", ms()); }
        DGZ_CHECK(hipMemcpy(d_at.p, text_at.data(), sizeof(uint64_t) * n, hipMemcpyHostToDevice));
        DGZ_CHECK(hipMemset(d_status.p, 0, sizeof(uint32_t)));
        DGZ_CHECK(launch_gunzip_text(d_syms.as<uint16_t>(), cap_syms, d_chunks.as<GunzipChunk>(), d_at.as<uint64_t>(), n, d_text.as<char>(),
                                     d_status.as<uint32_t>(), nullptr));
        if (tr) { DGZ_CHECK(hipDeviceSynchronize()); std::fprintf(stderr, "[scg]   gzip on the device: symbols turned into text after %.2f ms\n", ms()); }
        // CRC-32: pieces of 4 MB on the device, combined here like zlib's crc32_combine
        const uint64_t piece = uint64_t(4) << 20;
        const uint32_t np = static_cast<uint32_t>((total + piece - 1) / piece);
        std::vector<InflateMember> pieces(np);
        for (uint32_t i = 0; i < np; ++i) {
            pieces[i].in_off = pieces[i].in_len = 0;
            pieces[i].out_off = static_cast<uint32_t>(piece * i);
            pieces[i].out_len = static_cast<uint32_t>(std::min<uint64_t>(piece, total - piece * i));
            pieces[i].crc = 0;
        }
        d_pieces.alloc(sizeof(InflateMember) * np);
        d_crcs.alloc(sizeof(uint32_t) * np);
        DGZ_CHECK(hipMemcpy(d_pieces.p, pieces.data(), sizeof(InflateMember) * np, hipMemcpyHostToDevice));
        DGZ_CHECK(launch_crc_pieces(d_text.as<char>(), d_pieces.as<InflateMember>(), np, d_crcs.as<uint32_t>(), nullptr));
        std::vector<uint32_t> crcs(np);
        uint32_t status = 0;
        DGZ_CHECK(hipMemcpy(crcs.data(), d_crcs.p, sizeof(uint32_t) * np, hipMemcpyDeviceToHost));
        DGZ_CHECK(hipMemcpy(&status, d_status.p, sizeof(uint32_t), hipMemcpyDeviceToHost));
        if (status) return nullptr;
        uLong crc = crc32(0L, Z_NULL, 0);
        for (uint32_t i = 0; i < np; ++i) crc = crc32_combine(crc, crcs[i], static_cast<z_off_t>(pieces[i].out_len));
        if (static_cast<uint32_t>(crc) != want_crc) return nullptr;
        if (tr) std::fprintf(stderr, "[scg]   gzip on the device: %.2f GB of text in HBM, CRC-32 checked, after %.2f ms\n", total / 1e9, ms());
        return std::unique_ptr<TextSource>(new DeviceGunzipSource(device, std::move(d_text), total));
    } catch (const Declined&) {
        return nullptr;
    }
}

} // namespace

void release_device_gunzip_scratch() {
    std::lock_guard<std::mutex> g(scratch_mu);
    if (!scratch_cache.busy) scratch_cache.drop();
}

std::unique_ptr<TextSource> TextSource::open_on_device(const char* path, int device, int threads) {
    const char* e = std::getenv("SCG_DEVICE_GUNZIP");
    if (e && *e == '0') return nullptr;
    return decode(path, device, threads);
}

} // namespace scg
