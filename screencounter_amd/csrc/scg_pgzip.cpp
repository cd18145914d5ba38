// scg_pgzip.cpp -- the threads around scg_pgzip.h: chunks decoded speculatively in parallel, stitched in order,
// turned into bytes in parallel (the header explains the scheme).  Host code only.
#include "scg_pgzip.hpp"

#include <algorithm>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>
#include <zlib.h>

#include "scg_pgzip.h"

namespace scg {

namespace {

using namespace pgz;

// CRC-32 of a buffer: libdeflate's (carry-less multiplication, ~10 GB/s) when the image has the library, else zlib's.
typedef uint32_t (*crc_fn)(uint32_t, const void*, size_t);
uint32_t zlib_crc(uint32_t c, const void* p, size_t n) {
    const unsigned char* b = static_cast<const unsigned char*>(p);
    while (n) {
        const size_t k = std::min<size_t>(n, size_t(1) << 30);
        c = static_cast<uint32_t>(crc32(c, b, static_cast<uInt>(k)));
        b += k; n -= k;
    }
    return c;
}
crc_fn pick_crc() {
    const char* e = std::getenv("SCG_LIBDEFLATE");
    if (!(e && *e == '0')) {
        if (void* h = ::dlopen("libdeflate.so.0", RTLD_NOW | RTLD_LOCAL)) {
            if (void* f = ::dlsym(h, "libdeflate_crc32")) return reinterpret_cast<crc_fn>(f);
        }
    }
    return zlib_crc;
}

// A gzip member header at d[off ...) (RFC 1952): the offset of its DEFLATE stream, or 0 for anything but a header in
// the form `gzip` writes (optional name, comment and extra field; a header CRC or reserved flags are left to zlib).
size_t member_payload(const uint8_t* d, size_t size, size_t off) {
    if (off > size || size - off < 18 || d[off] != 0x1f || d[off + 1] != 0x8b || d[off + 2] != 8) return 0;
    const uint8_t flg = d[off + 3];
    if (flg & 0xE2) return 0;
    size_t p = off + 10;
    if (flg & 4) {
        if (p + 2 > size) return 0;
        p += 2 + (d[p] | (static_cast<size_t>(d[p + 1]) << 8));
    }
    for (int bit : {8, 16}) {                                   // FNAME, FCOMMENT: zero-terminated
        if (flg & bit) {
            while (p < size && d[p]) ++p;
            ++p;
        }
    }
    return p + 8 <= size ? p : 0;
}

// A run of a chunk's symbols that belongs to one member; `ends`: the member ends with it (trailer values attached).
struct Segment {
    size_t from, to;
    bool ends;
    uint32_t crc, isize;
};

struct Chunk {
    // stage 1
    uint64_t start = ~uint64_t(0), end = 0;      // bit positions; start == ~0: no block was found in the chunk's range
    uint16_t* buf = nullptr;                     // WINDOW marker symbols, then the chunk's symbols (from the pool)
    size_t n = 0;                                // symbols decoded
    std::vector<Segment> segs;
    bool started_at_header = false;              // decoded from a member header (nothing before it can be referred to)
    bool ends_at_header = false;                 // stopped in front of a member header, or at the end of the file
    bool reaches_eof = false;
    bool failed = false;                         // something invalid met on the way (not necessarily the stream's fault)
    bool s1_done = false;
    // stitch
    uint8_t* lut = nullptr;                      // 64 KiB: symbol -> byte (from the pool)
    uint32_t first_marker = MARKER;              // markers below this one point before the member's first byte
    // consumer
    size_t seg_i = 0, seg_off = 0;
};

struct Piece {
    const uint16_t* sym;
    const uint8_t* lut;
    char* dst;
    size_t n;
    uint32_t first_marker;       // > MARKER: symbols in [MARKER, first_marker) are "invalid distance too far back"
    bool ends;                   // the member ends with this piece
    uint32_t want_crc, want_isize;
    uint32_t crc;
    bool bad;
};

// Symbol buffers outlive a decoder: a fresh 32 MB buffer costs a page fault per 4 KB written and as much again when it is
// unmapped -- 60 ms of a 300 ms call on an 8 M-read file.  Buffers of the standard size go back here when a decoder is
// done and serve the next one; ParallelGunzip::release_cached() (scg_release_buffers) frees them.
constexpr size_t STANDARD_SYMBOLS = size_t(16) << 20;        // 1 MB chunks x 16
struct BufferCache {
    std::mutex mu;
    std::vector<uint16_t*> bufs;
    uint16_t* take() {
        std::lock_guard<std::mutex> g(mu);
        if (bufs.empty()) return nullptr;
        uint16_t* b = bufs.back();
        bufs.pop_back();
        return b;
    }
    void give(uint16_t* b) {
        {
            std::lock_guard<std::mutex> g(mu);
            if (bufs.size() < 40) { bufs.push_back(b); return; }
        }
        std::free(b);
    }
    void clear() {
        std::lock_guard<std::mutex> g(mu);
        for (uint16_t* b : bufs) std::free(b);
        bufs.clear();
    }
};
BufferCache& buffer_cache() {
    static BufferCache* c = new BufferCache;      // (never destroyed: worker threads of a late call may still be around at exit)
    return *c;
}

}  // namespace

void ParallelGunzip::release_cached() { buffer_cache().clear(); }

size_t ParallelGunzip::chunk_size_for(size_t size, int threads) {
    if (const char* e = std::getenv("SCG_PGZIP_CHUNK_KB")) {         // test hook: tiny chunks
        const long kb = std::atol(e);
        if (kb >= 1) return static_cast<size_t>(kb) << 10;
    }
    if (threads < 2 || size < (size_t(2) << 20)) return 0;          // one libdeflate stream is as fast on such a file
    size_t c = size / (static_cast<size_t>(threads) * 8);
    c = std::max<size_t>(c, size_t(256) << 10);
    c = std::min<size_t>(c, size_t(1) << 20);
    return c;
}

struct ParallelGunzip::Impl {
    const uint8_t* data;
    size_t size;
    int n_threads;
    size_t chunk_bytes, cap_symbols, n_chunks, lookahead;
    crc_fn crc;

    std::mutex mu;
    std::condition_variable cv_work, cv_ready;
    std::vector<Chunk> chunks;
    std::vector<uint16_t*> free_bufs;
    std::vector<uint8_t*> free_luts;
    size_t next_s1 = 0;                // next chunk whose stage 1 has not been claimed
    size_t stitched_upto = 0;          // chunks below are stitched
    size_t consume_at = 0;             // first chunk the consumer has not finished
    uint64_t cur = 0;                  // bit position where the stitched stream ends
    bool cur_at_header = true;         // ... and a member header (or the end of the file) is what comes there
    bool stream_done = false;          // the stitching reached the end of the file
    bool fail = false, stop = false, stitching = false;
    size_t redone = 0;                 // chunks the stitching pass had to decode again
    std::vector<Piece>* pieces = nullptr;
    size_t piece_next = 0, piece_done = 0;
    uint8_t tail[WINDOW];              // the text in front of the next chunk to stitch, right-aligned
    size_t tail_n = 0;                 // how much of it exists (the member may be younger than a window)
    uint32_t run_crc = 0;              // the current member, as far as it has been handed out
    uint64_t run_len = 0;
    std::vector<std::thread> workers;

    Impl(const uint8_t* d, size_t n, int threads, size_t chunk) : data(d), size(n), n_threads(std::max(1, threads)), chunk_bytes(chunk) {
        cap_symbols = std::max<size_t>(chunk_bytes * 16, size_t(1) << 16);       // a chunk that inflates more than 16-fold ends the attempt
        if (chunk_bytes >= (size_t(256) << 10) && cap_symbols <= STANDARD_SYMBOLS) cap_symbols = STANDARD_SYMBOLS;   // (interchangeable buffers; untouched pages cost nothing)
        if (const char* e = std::getenv("SCG_PGZIP_CAP_KB")) { const long kb = std::atol(e); if (kb >= 1) cap_symbols = static_cast<size_t>(kb) << 10; }   // test hook
        n_chunks = std::max<size_t>(1, (size + chunk_bytes - 1) / chunk_bytes);
        chunks.resize(n_chunks);
        lookahead = static_cast<size_t>(n_threads) * 2 + 2;
        crc = pick_crc();
        for (int t = 0; t < n_threads; ++t) workers.emplace_back([this] { worker(); });
    }

    ~Impl() {
        {
            std::lock_guard<std::mutex> g(mu);
            stop = true;
        }
        cv_work.notify_all();
        for (auto& t : workers) t.join();
        auto done_with = [&](uint16_t* b) {
            if (!b) return;
            if (cap_symbols == STANDARD_SYMBOLS) buffer_cache().give(b); else std::free(b);
        };
        for (Chunk& c : chunks) { done_with(c.buf); std::free(c.lut); }
        for (uint16_t* b : free_bufs) done_with(b);
        for (uint8_t* l : free_luts) std::free(l);
    }

    // Symbol buffers are recycled: a fresh one costs a page fault per 4 KB written, a third of the decoding time.
    uint16_t* take_buf() {
        {
            std::lock_guard<std::mutex> g(mu);
            if (!free_bufs.empty()) { uint16_t* b = free_bufs.back(); free_bufs.pop_back(); return b; }
        }
        uint16_t* b = cap_symbols == STANDARD_SYMBOLS ? buffer_cache().take() : nullptr;
        if (b) return b;                                  // (its marker prefix is never overwritten)
        b = static_cast<uint16_t*>(std::malloc((WINDOW + cap_symbols + 16) * sizeof(uint16_t)));
        if (!b) throw std::bad_alloc();
        for (uint32_t k = 0; k < WINDOW; ++k) b[k] = static_cast<uint16_t>(MARKER + k);
        return b;
    }

    // ---- stage 1 -----------------------------------------------------------------------------------------------
    // Decodes chunk j from bit `start` -- a block start, or a member header when `at_header` -- to the first block
    // boundary (or member header) at or beyond the nominal start of chunk j + 1.  The lock is not held.
    void decode_chunk(size_t j, uint64_t start, bool at_header) {
        Chunk& c = chunks[j];
        c.n = 0; c.segs.clear(); c.reaches_eof = false; c.failed = false; c.ends_at_header = false;
        c.start = start; c.started_at_header = at_header;
        if (!c.buf) c.buf = take_buf();
        uint16_t* out = c.buf + WINDOW;
        const uint64_t stop_at = static_cast<uint64_t>(std::min(size, (j + 1) * chunk_bytes)) * 8u;
        std::unique_ptr<Tables> T(new Tables);
        uint8_t lens[320];
        Bits br;
        uint64_t bit = start;
        size_t member_from = 0;              // where the current member's symbols begin (0: before the chunk, or at its start)
        bool in_member = !at_header, member_began_here = false;
        for (;;) {
            if (!in_member) {
                const size_t off = static_cast<size_t>(bit >> 3);
                if (off == size) { c.reaches_eof = true; c.ends_at_header = true; break; }
                if (bit >= stop_at && bit != start) { c.ends_at_header = true; break; }       // the next chunk starts with this header
                const size_t p = member_payload(data, size, off);
                if (!p) { c.failed = true; break; }
                bit = static_cast<uint64_t>(p) * 8u;
                in_member = true;
                member_began_here = true;
                member_from = c.n;
            }
            br.open(data, size, bit);
            // matches reach back to the member's first byte: WINDOW symbols of unknown text when it lies before the chunk
            size_t rel = c.n - member_from;
            const int rc = decode_block(br, *T, out + member_from, rel, cap_symbols - member_from, member_began_here ? 0 : WINDOW, lens);
            if (rc == BLOCK_BAD || rc == BLOCK_FULL) { c.failed = true; break; }
            c.n = member_from + rel;
            bit = br.bitpos();
            if (rc == BLOCK_FINAL) {
                const size_t t = static_cast<size_t>((bit + 7) >> 3);          // trailer at the next byte boundary: CRC-32, ISIZE
                if (t + 8 > size) { c.failed = true; break; }
                Segment s;
                s.from = member_from; s.to = c.n; s.ends = true;
                s.crc = data[t] | (uint32_t(data[t + 1]) << 8) | (uint32_t(data[t + 2]) << 16) | (uint32_t(data[t + 3]) << 24);
                s.isize = data[t + 4] | (uint32_t(data[t + 5]) << 8) | (uint32_t(data[t + 6]) << 16) | (uint32_t(data[t + 7]) << 24);
                c.segs.push_back(s);
                bit = static_cast<uint64_t>(t + 8) * 8u;
                in_member = false;
                member_from = c.n;
                continue;
            }
            if (bit >= stop_at) break;
        }
        if (in_member && !c.failed) c.segs.push_back(Segment{member_from, c.n, false, 0, 0});
        c.end = bit;
    }

    void stage1(size_t j) {
        Chunk& c = chunks[j];
        if (j == 0) {
            decode_chunk(0, 0, true);
            return;
        }
        const uint64_t from = static_cast<uint64_t>(j * chunk_bytes) * 8u;
        const uint64_t to = static_cast<uint64_t>(std::min(size, (j + 1) * chunk_bytes)) * 8u;
        std::unique_ptr<Tables> T(new Tables);
        uint8_t lens[320];
        uint64_t at = from;
        for (int tries = 0; tries < 4; ++tries) {
            const uint64_t s = find_dynamic_block(data, size, at, to, *T, lens);
            if (s == ~uint64_t(0)) { c.start = s; c.failed = true; return; }
            decode_chunk(j, s, false);
            if (!c.failed) return;
            at = s + 1;                                      // a header that parsed but was none: look on
        }
    }

    // ---- stitching: one thread at a time, in chunk order.  Called with the lock held; releases it around the work.
    void advance(std::unique_lock<std::mutex>& lk) {
        if (stitching) return;
        stitching = true;
        while (!fail && !stream_done && stitched_upto < n_chunks && chunks[stitched_upto].s1_done) {
            const size_t j = stitched_upto;
            Chunk& c = chunks[j];
            uint8_t* lut = nullptr;
            if (!free_luts.empty()) { lut = free_luts.back(); free_luts.pop_back(); }
            lk.unlock();
            bool ok = true;
            const uint64_t nominal_end = static_cast<uint64_t>(std::min(size, (j + 1) * chunk_bytes)) * 8u;
            if (j > 0 && cur >= nominal_end && j + 1 < n_chunks) {
                // the chunk before ran through this one's whole range (one long block): nothing of it is left
                c.n = 0; c.segs.clear(); c.end = cur; c.reaches_eof = false; c.failed = false;
                c.ends_at_header = cur_at_header; c.started_at_header = cur_at_header;
            } else {
                // accepted as decoded only if it began where the stream really continues; a chunk found behind a
                // member header that the one before stopped in front of began at that member's first block
                bool match = !c.failed && c.start == cur && c.started_at_header == cur_at_header;
                if (!match && !c.failed && cur_at_header && !c.started_at_header) {
                    const size_t p = member_payload(data, size, static_cast<size_t>(cur >> 3));
                    if (p && c.start == static_cast<uint64_t>(p) * 8u) match = true;      // (its markers, if any, are void: first_marker below)
                }
                if (!match) {
                    // (a stream in which the guesses keep failing -- stored or fixed blocks throughout -- is decoded
                    // here, by one thread: not what this decoder is for)
                    if (++redone >= 8 && redone * 2 > j) ok = false;
                    else {
                        try { decode_chunk(j, cur, cur_at_header); } catch (...) { c.failed = true; }
                        if (c.failed) ok = false;
                    }
                }
            }
            if (ok) {
                if (!lut) { lut = static_cast<uint8_t*>(std::malloc(65536)); if (!lut) ok = false; }
            }
            if (ok) {
                c.lut = lut; lut = nullptr;
                if (cur_at_header) tail_n = 0;
                for (int b = 0; b < 256; ++b) c.lut[b] = static_cast<uint8_t>(b);
                std::memset(c.lut + MARKER, 0, WINDOW - tail_n);
                std::memcpy(c.lut + MARKER + (WINDOW - tail_n), tail + (WINDOW - tail_n), tail_n);
                c.first_marker = static_cast<uint32_t>(MARKER + (WINDOW - tail_n));
                // the text in front of the next chunk: what follows the last member start, at most a window of it
                const uint16_t* sym = c.buf ? c.buf + WINDOW : nullptr;
                size_t member_from = 0;
                bool fresh_member = false;
                for (const Segment& s : c.segs) if (s.ends) { member_from = s.to; fresh_member = true; }
                if (fresh_member) tail_n = 0;
                const size_t fresh = c.n - member_from;
                if (fresh >= WINDOW) {
                    for (size_t i = 0; i < WINDOW; ++i) tail[i] = c.lut[sym[c.n - WINDOW + i]];
                    tail_n = WINDOW;
                } else if (fresh > 0) {
                    const size_t keep = std::min(tail_n, WINDOW - fresh);
                    std::memmove(tail + WINDOW - fresh - keep, tail + WINDOW - keep, keep);
                    for (size_t i = 0; i < fresh; ++i) tail[WINDOW - fresh + i] = c.lut[sym[member_from + i]];
                    tail_n = keep + fresh;
                }
            }
            lk.lock();
            if (lut) free_luts.push_back(lut);
            if (!ok) { fail = true; break; }
            cur = c.end;
            cur_at_header = c.ends_at_header;
            ++stitched_upto;
            if (c.reaches_eof) stream_done = true;
            cv_ready.notify_all();
        }
        if (!fail && !stream_done && stitched_upto == n_chunks) fail = true;      // the last chunk did not end the stream
        stitching = false;
        if (fail || stream_done) { cv_ready.notify_all(); cv_work.notify_all(); }
    }

    void resolve(Piece& p) const {
        const uint16_t* s = p.sym;
        const uint8_t* lut = p.lut;
        uint8_t* d = reinterpret_cast<uint8_t*>(p.dst);
        size_t i = 0;
        for (; i + 8 <= p.n; i += 8) {                 // eight look-ups, one store
            uint64_t v = static_cast<uint64_t>(lut[s[i]]) | (static_cast<uint64_t>(lut[s[i + 1]]) << 8) |
                         (static_cast<uint64_t>(lut[s[i + 2]]) << 16) | (static_cast<uint64_t>(lut[s[i + 3]]) << 24) |
                         (static_cast<uint64_t>(lut[s[i + 4]]) << 32) | (static_cast<uint64_t>(lut[s[i + 5]]) << 40) |
                         (static_cast<uint64_t>(lut[s[i + 6]]) << 48) | (static_cast<uint64_t>(lut[s[i + 7]]) << 56);
            std::memcpy(d + i, &v, 8);
        }
        for (; i < p.n; ++i) d[i] = lut[s[i]];
        p.bad = false;
        if (p.first_marker > MARKER) {
            // the member is younger than a window: a marker that points before its first byte is zlib's
            // "invalid distance too far back" (only here is the age of the window known)
            for (size_t k = 0; k < p.n; ++k) if (s[k] >= MARKER && s[k] < p.first_marker) { p.bad = true; break; }
        }
        p.crc = crc(0, d, p.n);
    }

    void worker() {
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            if (stop) return;
            if (pieces && piece_next < pieces->size()) {
                Piece& p = (*pieces)[piece_next++];
                lk.unlock();
                resolve(p);
                lk.lock();
                if (++piece_done == pieces->size()) cv_ready.notify_all();
                continue;
            }
            if (!fail && !stream_done && next_s1 < n_chunks && next_s1 < consume_at + lookahead) {
                const size_t j = next_s1++;
                lk.unlock();
                bool threw = false;
                try { stage1(j); } catch (...) { threw = true; }
                lk.lock();
                if (threw) { fail = true; cv_ready.notify_all(); }
                chunks[j].s1_done = true;
                advance(lk);
                continue;
            }
            cv_work.wait(lk);
        }
    }

    size_t read(char* dst, size_t cap) {
        size_t have = 0;
        std::unique_lock<std::mutex> lk(mu);
        while (have < cap && !fail) {
            while (!fail && consume_at >= stitched_upto && !stream_done) {
                advance(lk);
                if (fail || consume_at < stitched_upto || stream_done) break;
                cv_work.notify_all();
                cv_ready.wait(lk);
            }
            if (fail) break;
            if (consume_at >= stitched_upto) break;                 // the end of the stream
            // pieces of the stitched chunks, as many as fit; they end at member ends, so that every CRC belongs to one member
            std::vector<Piece> ps;
            size_t at = have;
            const size_t piece_max = size_t(1) << 20;
            bool full = false;
            for (size_t j = consume_at; j < stitched_upto && !full; ++j) {
                Chunk& c = chunks[j];
                while (c.seg_i < c.segs.size()) {
                    const Segment& s = c.segs[c.seg_i];
                    const size_t left = (s.to - s.from) - c.seg_off;
                    if (left > 0 && at >= cap) { full = true; break; }
                    const size_t n = std::min(std::min(left, piece_max), cap - at);       // (0: an empty member's end is still checked)
                    Piece p;
                    p.sym = c.buf + WINDOW + s.from + c.seg_off; p.lut = c.lut; p.dst = dst + at; p.n = n;
                    p.first_marker = c.seg_i == 0 ? c.first_marker : MARKER;
                    p.ends = s.ends && n == left;
                    p.want_crc = s.crc; p.want_isize = s.isize;
                    p.crc = 0; p.bad = false;
                    ps.push_back(p);
                    at += n;
                    c.seg_off += n;
                    if (n == left) { ++c.seg_i; c.seg_off = 0; }
                }
            }
            if (!ps.empty()) {
                pieces = &ps; piece_next = 0; piece_done = 0;
                cv_work.notify_all();
                while (piece_next < ps.size()) {                    // the reader lends a hand
                    Piece& p = ps[piece_next++];
                    lk.unlock();
                    resolve(p);
                    lk.lock();
                    ++piece_done;
                }
                while (piece_done < ps.size()) cv_ready.wait(lk);
                pieces = nullptr;
                // CRCs in order; every member is checked against its trailer as it ends, like zlib's inflate does
                for (const Piece& p : ps) {
                    if (p.bad) { fail = true; break; }
                    run_crc = p.n ? static_cast<uint32_t>(crc32_combine(run_crc, p.crc, static_cast<z_off_t>(p.n))) : run_crc;
                    run_len += p.n;
                    if (p.ends) {
                        if (run_crc != p.want_crc || static_cast<uint32_t>(run_len) != p.want_isize) { fail = true; break; }
                        run_crc = 0; run_len = 0;
                    }
                }
                if (fail) break;
            }
            have = at;
            while (consume_at < stitched_upto && chunks[consume_at].seg_i == chunks[consume_at].segs.size()) {
                Chunk& c = chunks[consume_at];
                if (c.buf) { free_bufs.push_back(c.buf); c.buf = nullptr; }
                if (c.lut) { free_luts.push_back(c.lut); c.lut = nullptr; }
                ++consume_at;
            }
            cv_work.notify_all();
            if (stream_done && consume_at >= stitched_upto) break;
        }
        if (fail) { cv_work.notify_all(); return 0; }
        return have;
    }
};

ParallelGunzip::ParallelGunzip(const uint8_t* data, size_t size, int threads) {
    size_t chunk = chunk_size_for(size, threads);
    if (!chunk) chunk = std::max<size_t>(size, 1);
    impl = new Impl(data, size, threads, chunk);
}
ParallelGunzip::~ParallelGunzip() { delete impl; }
size_t ParallelGunzip::read(char* dst, size_t cap) { return impl->read(dst, cap); }
bool ParallelGunzip::failed() const {
    std::lock_guard<std::mutex> g(impl->mu);
    return impl->fail;
}

}  // namespace scg
