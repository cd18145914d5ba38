// scg_inflate.h -- DEFLATE (RFC 1951) decoder for one gzip member, written to run one member per GPU wavefront.
//
// BGZF files ("blocked gzip", what bgzip writes) are a concatenation of small independent gzip members (<= 64 KiB of
// text each) that announce their compressed size; the reference inflates them one after the other on the caller
// thread (byteme/GzipFileReader.hpp:39-51 -> zlib).  Here the compressed bytes cross the PCIe link as they are and
// every wavefront of scg_inflate.hip's kernel inflates one member, its Huffman tables in LDS; the text never exists
// on the host.  Decoding is a chain of dependent steps, so all lanes of the wavefront walk it together (every value
// below is wave-uniform: the compiler keeps the decoder's state in scalar registers) and the lanes only differ where
// bytes move: a match is copied 64 bytes at a time, one per lane.  The function is plain C++ so that the very same code
// can be compiled for the host by the tests (tests/inflate_harness.cpp: differential runs against zlib, corrupted
// streams under ASan) with a "wavefront" of one lane; the product only ever runs it on the device.
//
// Accept / reject behaviour follows zlib's inflate (inflate.c, inftrees.c) rule by rule -- over-subscribed and
// incomplete code sets, missing end-of-block code, too many length / distance symbols, invalid stored block lengths,
// distances beyond the start of the member -- because a member this decoder accepts is not looked at by zlib again.
// Whatever it rejects is redone by the host path, whose errors are zlib's own.  Every loop is bounded by the member's
// compressed and inflated sizes: a corrupt stream ends in an error, never in a runaway lane.
#ifndef SCG_INFLATE_H
#define SCG_INFLATE_H

#include <stdint.h>

#if defined(__HIPCC__)
#define SCG_HD __host__ __device__ __forceinline__
#else
#define SCG_HD inline
#endif

namespace scginf {

enum : int {
    INFLATE_OK = 0,
    INFLATE_BAD_DATA = 1,      // not a valid DEFLATE stream
    INFLATE_BAD_SIZE = 2,      // valid so far, but it does not fill / overruns the announced sizes
};

constexpr int LIT_BITS = 9;       // primary table of the literal/length code
constexpr int DIST_BITS = 7;      // primary table of the distance code
constexpr uint32_t IN_SLACK = 64; // bytes readable beyond a member's payload (the caller pads its buffer)

// One member's tables (LDS on the device, one set per wavefront): 2016 bytes.
struct LaneTables {
    uint16_t lit[1 << LIT_BITS];      // (code length << 12) | symbol, 0 = the code is longer than LIT_BITS (or invalid);
                                      // while a dynamic header is read, the code lengths live here (bytes)
    uint16_t dtab[1 << DIST_BITS];    // the same for distance codes; during the header: the code-length code's table
    uint16_t lsym[288];               // literal/length symbols sorted by (code length, symbol)
    uint16_t dsym[32];
    uint16_t lcount[16];              // number of codes of each length
    uint16_t dcount[16];
    uint16_t offs[16];                // scratch of the table builder
};

// The lanes that walk a member together ("Wave" below).  The decoder's own state is the same in every lane; the lanes
// differ only inside a Wave::Vec, a vector with one byte per lane, which the decoder moves around but never looks into:
//   width()                          lanes
//   set(v, j, byte)                  v[j] = byte
//   load(v, src, n, dist)            v[j] = src[j mod dist], j < n <= width()
//   store(dst, v, n)                 dst[j] = v[j], j < n
//   copy(dst, src, n, dist)          dst[j] = src[j mod dist], j < n, any n (src + dist <= dst: nothing it reads is written by it)
// On the device a Vec is one register and each operation one instruction per lane (scg_inflate.hip); the host tests
// run the same decoder with arrays (tests/inflate_harness.cpp), both one lane wide and 64 lanes wide.
template<int WIDTH>
struct HostWave {
    struct Vec { uint8_t b[WIDTH]; };
    SCG_HD uint32_t width() const { return WIDTH; }
    SCG_HD void set(Vec& v, uint32_t j, uint32_t byte) const { v.b[j] = static_cast<uint8_t>(byte); }
    SCG_HD void load(Vec& v, const uint8_t* src, uint32_t n, uint32_t dist) const { for (uint32_t j = 0; j < n; ++j) v.b[j] = src[j % dist]; }
    SCG_HD void store(uint8_t* dst, const Vec& v, uint32_t n) const { for (uint32_t j = 0; j < n; ++j) dst[j] = v.b[j]; }
    SCG_HD void copy(uint8_t* dst, const uint8_t* src, uint32_t n, uint32_t dist) const {
        // all loads of a round of WIDTH bytes before its stores, as the lanes of a wavefront do it
        for (uint32_t at = 0; at < n; at += WIDTH) {
            Vec v;
            const uint32_t m = n - at < static_cast<uint32_t>(WIDTH) ? n - at : static_cast<uint32_t>(WIDTH);
            for (uint32_t j = 0; j < m; ++j) v.b[j] = src[(at + j) % dist];
            for (uint32_t j = 0; j < m; ++j) dst[at + j] = v.b[j];
        }
    }
};

SCG_HD uint32_t load32(const uint8_t* p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }
SCG_HD uint64_t load64(const uint8_t* p) { uint64_t v; __builtin_memcpy(&v, p, 8); return v; }
SCG_HD void store64(uint8_t* p, uint64_t v) { __builtin_memcpy(p, &v, 8); }

struct BitReader {
    const uint8_t* in;
    uint32_t pos;        // bytes of the stream that have entered buf
    uint32_t len;        // payload bytes; in[0 .. len + IN_SLACK) is readable
    uint64_t buf;        // LSB first
    uint32_t cnt;        // valid bits in buf
    uint32_t a0, a1;     // the stream's next two words, loaded ahead of their use: in[pos .. pos + 8)

    SCG_HD void open(const uint8_t* p, uint32_t n) {
        in = p; pos = 0; len = n; buf = 0; cnt = 0;
        a0 = load32(in);
        a1 = load32(in + 4);
    }
    // Continues at byte `at` of the stream (bit buffer empty).
    SCG_HD void seek(uint32_t at) {
        pos = at; buf = 0; cnt = 0;
        a0 = load32(in + pos);
        a1 = load32(in + pos + 4);
    }
    // At least 33 bits afterwards.  The word that enters the buffer was requested two refills ago.
    SCG_HD void refill() {
        if (cnt <= 32) {
            buf |= static_cast<uint64_t>(a0) << cnt;
            cnt += 32;
            pos += 4;
            a0 = a1;
            a1 = load32(in + pos + 4);
        }
    }
    SCG_HD uint32_t bits(uint32_t n) {
        const uint32_t v = static_cast<uint32_t>(buf) & ((1u << n) - 1u);
        buf >>= n;
        cnt -= n;
        return v;
    }
    SCG_HD bool overrun() const { return pos > len + 8; }     // more than the refill look-ahead beyond the payload
};

// The member's text.  Literals are collected, one per lane, and written a wavefront's width at a time.  A match of
// at most that many bytes is loaded at once but stored only when the next match comes along (or the text ends): the
// wavefront does not need the bytes to go on decoding, so the load's latency overlaps the next symbols.  Whatever reads
// text that is still pending -- a match whose source reaches into the pending literals or the pending match -- has it
// written out first.
template<class Wave>
struct TextWriter {
    const Wave& wave;
    uint8_t* out;
    uint32_t op;             // bytes produced
    uint32_t n_lit;          // of which the last n_lit are literals still in `lit`
    typename Wave::Vec lit;
    uint32_t m_at, m_n;      // a match of m_n bytes for out[m_at ...) still in `held`
    typename Wave::Vec held;

    SCG_HD TextWriter(const Wave& w, uint8_t* o) : wave(w), out(o), op(0), n_lit(0), m_at(0), m_n(0) {}
    SCG_HD void flush_literals() {
        if (n_lit) wave.store(out + op - n_lit, lit, n_lit);
        n_lit = 0;
    }
    SCG_HD void flush_match() {
        if (m_n) wave.store(out + m_at, held, m_n);
        m_n = 0;
    }
    SCG_HD void literal(uint32_t byte) {
        wave.set(lit, n_lit, byte);
        ++n_lit;
        ++op;
        if (n_lit == wave.width()) flush_literals();
    }
    // out[op + j] = out[op - dist + j mod dist], j < n; dist <= op.
    SCG_HD void match(uint32_t n, uint32_t dist) {
        const uint32_t src = op - dist, reach = src + (n < dist ? n : dist);      // reads out[src .. reach)
        flush_literals();                       // (pending literals are always the text's last bytes)
        if (m_n && reach > m_at && src < m_at + m_n) flush_match();
        if (n <= wave.width()) {
            typename Wave::Vec v;
            wave.load(v, out + src, n, dist);
            flush_match();                      // (the one before: its bytes have had a symbol or more to arrive)
            held = v;
            m_at = op;
            m_n = n;
        } else {
            flush_match();
            wave.copy(out + op, out + src, n, dist);
        }
        op += n;
    }
    // n bytes from elsewhere (a stored block)
    SCG_HD void raw(const uint8_t* from, uint32_t n) {
        flush_literals();
        flush_match();
        wave.copy(out + op, from, n, n ? n : 1u);
        op += n;
    }
    SCG_HD void finish() {
        flush_literals();
        flush_match();
    }
};

SCG_HD uint32_t reverse_bits(uint32_t code, int len) {
    uint32_t r = 0;
    for (int i = 0; i < len; ++i) { r = (r << 1) | (code & 1u); code >>= 1; }
    return r;
}

// Builds the decoding tables of one canonical Huffman code from lens[0 .. n) (inftrees.c: inflate_table).
// kind: 0 = code-length code (incomplete sets rejected), 1 = literal/length or distance code (an incomplete set is
// accepted only when its longest code has one bit).  Returns false for a set zlib rejects.
// `table` has 1 << tbits entries; `count`, `sym`, `offs` as in LaneTables.  lens may lie inside `table`: it is not read
// any more once the table is being written.
SCG_HD bool build_code(const uint8_t* lens, int n, int kind, uint16_t* table, int tbits, uint16_t* count, uint16_t* sym, uint16_t* offs) {
    for (int l = 0; l < 16; ++l) count[l] = 0;
    for (int s = 0; s < n; ++s) count[lens[s]] = static_cast<uint16_t>(count[lens[s]] + 1);
    int max = 15;
    while (max >= 1 && count[max] == 0) --max;
    if (max >= 1) {
        int left = 1;
        for (int l = 1; l <= 15; ++l) {
            left <<= 1;
            left -= count[l];
            if (left < 0) return false;                       // over-subscribed
        }
        if (left > 0 && (kind == 0 || max != 1)) return false;   // incomplete set
    }
    // (max == 0: no codes at all -- every lookup below fails, which is the error zlib raises when such a code is used)
    offs[1] = 0;
    for (int l = 1; l < 15; ++l) offs[l + 1] = static_cast<uint16_t>(offs[l] + count[l]);
    for (int s = 0; s < n; ++s) {
        const int l = lens[s];
        if (l) { sym[offs[l]] = static_cast<uint16_t>(s); offs[l] = static_cast<uint16_t>(offs[l] + 1); }
    }
    count[0] = 0;
    // lens is dead from here on
    for (int i = 0; i < (1 << tbits); ++i) table[i] = 0;
    uint32_t code = 0;
    int idx = 0;
    for (int l = 1; l <= tbits; ++l) {
        for (int k = 0; k < count[l]; ++k) {
            const uint16_t entry = static_cast<uint16_t>((l << 12) | sym[idx++]);
            for (uint32_t i = reverse_bits(code, l); i < (1u << tbits); i += 1u << l) table[i] = entry;
            ++code;
        }
        code <<= 1;
    }
    return true;
}

// A code that the primary table does not resolve: canonical decoding one bit at a time.  Returns the symbol, or -1
// for a bit pattern that is no code.  Needs 15 bits in the reader.
SCG_HD int decode_slow(BitReader& br, const uint16_t* count, const uint16_t* sym) {
    uint32_t code = 0, first = 0, index = 0;
    uint32_t b = static_cast<uint32_t>(br.buf);
    for (int l = 1; l <= 15; ++l) {
        code |= b & 1u;
        b >>= 1;
        const uint32_t c = count[l];
        if (code < first + c) {
            br.buf >>= l;
            br.cnt -= static_cast<uint32_t>(l);
            return sym[index + (code - first)];
        }
        index += c;
        first = (first + c) << 1;
        code <<= 1;
    }
    return -1;
}

SCG_HD int decode_symbol(BitReader& br, const uint16_t* table, int tbits, const uint16_t* count, const uint16_t* sym) {
    const uint32_t e = table[static_cast<uint32_t>(br.buf) & ((1u << tbits) - 1u)];
    if (e) {
        const uint32_t l = e >> 12;
        br.buf >>= l;
        br.cnt -= l;
        return static_cast<int>(e & 0xFFFu);
    }
    return decode_slow(br, count, sym);
}

// Inflates the raw DEFLATE stream in[0 .. in_len) into out[0 .. out_len): INFLATE_OK only if the stream is valid, ends
// exactly at in_len and produces exactly out_len bytes.  in must be readable up to in_len + IN_SLACK.
template<class Wave>
SCG_HD int inflate_member(const uint8_t* in, uint32_t in_len, uint8_t* out, uint32_t out_len, LaneTables& T, const Wave& wave) {
    BitReader br;
    br.open(in, in_len);
    TextWriter<Wave> w(wave, out);
    uint32_t& op = w.op;
    uint8_t* const lens = reinterpret_cast<uint8_t*>(T.lit);             // 320 code lengths fit the 1 KiB of T.lit
    uint32_t last;
    do {
        if (br.overrun()) return INFLATE_BAD_DATA;
        br.refill();
        last = br.bits(1);
        const uint32_t type = br.bits(2);
        if (type == 0) {
            // stored block: LEN, ~LEN at the next byte boundary, then LEN bytes
            br.bits(br.cnt & 7u);
            br.refill();
            br.refill();
            const uint32_t n = br.bits(16), nn = br.bits(16);
            if ((n ^ 0xFFFFu) != nn) return INFLATE_BAD_DATA;           // "invalid stored block lengths"
            if (n > out_len - op) return INFLATE_BAD_SIZE;
            uint32_t left = n;
            while (left && br.cnt) {
                w.literal(br.bits(8));
                --left;
            }
            if (left) {
                // the bit buffer is empty: br.pos is the next byte of the stream
                if (br.pos > in_len || left > in_len - br.pos) return INFLATE_BAD_DATA;
                w.raw(in + br.pos, left);
                br.seek(br.pos + left);
            }
            continue;
        }
        if (type == 3) return INFLATE_BAD_DATA;                          // "invalid block type"
        int nlen, ndist;
        if (type == 1) {
            nlen = 288; ndist = 32;                                      // the fixed code (inflate.c: fixedtables)
            for (int s = 0; s < 144; ++s) lens[s] = 8;
            for (int s = 144; s < 256; ++s) lens[s] = 9;
            for (int s = 256; s < 280; ++s) lens[s] = 7;
            for (int s = 280; s < 288; ++s) lens[s] = 8;
            for (int s = 0; s < 32; ++s) lens[288 + s] = 5;
        } else {
            nlen = static_cast<int>(br.bits(5)) + 257;
            ndist = static_cast<int>(br.bits(5)) + 1;
            const int ncode = static_cast<int>(br.bits(4)) + 4;
            if (nlen > 286 || ndist > 30) return INFLATE_BAD_DATA;      // "too many length or distance symbols"
            // the code-length code: its 19 lengths sit behind the 320 bytes reserved for the lengths proper
            uint8_t* const cl = lens + 320;
            const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
            for (int i = 0; i < 19; ++i) cl[i] = 0;
            for (int i = 0; i < ncode; ++i) {
                br.refill();
                cl[order[i]] = static_cast<uint8_t>(br.bits(3));
            }
            // (its symbols and counts borrow the distance code's arrays, which are built afterwards)
            if (!build_code(cl, 19, 0, T.dtab, DIST_BITS, T.dcount, T.dsym, T.offs)) return INFLATE_BAD_DATA;   // "invalid code lengths set"
            int have = 0;
            while (have < nlen + ndist) {
                if (br.overrun()) return INFLATE_BAD_DATA;
                br.refill();
                const int s = decode_symbol(br, T.dtab, DIST_BITS, T.dcount, T.dsym);
                if (s < 0) return INFLATE_BAD_DATA;
                if (s < 16) {
                    lens[have++] = static_cast<uint8_t>(s);
                    continue;
                }
                uint8_t fill = 0;
                int rep;
                if (s == 16) {
                    if (have == 0) return INFLATE_BAD_DATA;             // "invalid bit length repeat"
                    fill = lens[have - 1];
                    rep = 3 + static_cast<int>(br.bits(2));
                } else if (s == 17) {
                    rep = 3 + static_cast<int>(br.bits(3));
                } else {
                    rep = 11 + static_cast<int>(br.bits(7));
                }
                if (have + rep > nlen + ndist) return INFLATE_BAD_DATA;  // "invalid bit length repeat"
                while (rep--) lens[have++] = fill;
            }
            if (lens[256] == 0) return INFLATE_BAD_DATA;                 // "invalid code -- missing end-of-block"
        }
        // distance code first: its lengths lie in the area the literal/length table is about to take
        if (!build_code(lens + nlen, ndist, 1, T.dtab, DIST_BITS, T.dcount, T.dsym, T.offs)) return INFLATE_BAD_DATA;   // "invalid distances set"
        {
            // the literal/length table overwrites its own lengths: the builder reads them before it writes
            if (!build_code(lens, nlen, 1, T.lit, LIT_BITS, T.lcount, T.lsym, T.offs)) return INFLATE_BAD_DATA;         // "invalid literal/lengths set"
        }
        for (;;) {
            if (br.overrun()) return INFLATE_BAD_DATA;
            br.refill();
            int s = decode_symbol(br, T.lit, LIT_BITS, T.lcount, T.lsym);
            if (s < 0) return INFLATE_BAD_DATA;                          // "invalid literal/length code"
            if (s < 256) {
                if (op >= out_len) return INFLATE_BAD_SIZE;
                w.literal(static_cast<uint32_t>(s));
                continue;
            }
            if (s == 256) break;
            if (s > 285) return INFLATE_BAD_DATA;
            s -= 257;
            uint32_t n;
            if (s < 8) {
                n = static_cast<uint32_t>(s) + 3u;
            } else if (s == 28) {
                n = 258;
            } else {
                const uint32_t eb = static_cast<uint32_t>(s - 4) >> 2;
                n = ((4u + (static_cast<uint32_t>(s) & 3u)) << eb) + 3u + br.bits(eb);
            }
            br.refill();
            const int d = decode_symbol(br, T.dtab, DIST_BITS, T.dcount, T.dsym);
            if (d < 0 || d >= 30) return INFLATE_BAD_DATA;               // "invalid distance code"
            uint32_t dist;
            if (d < 4) {
                dist = static_cast<uint32_t>(d) + 1u;
            } else {
                const uint32_t eb = (static_cast<uint32_t>(d) >> 1) - 1u;
                dist = ((2u + (static_cast<uint32_t>(d) & 1u)) << eb) + 1u + br.bits(eb);
            }
            if (dist > op) return INFLATE_BAD_DATA;                      // "invalid distance too far back"
            if (n > out_len - op) return INFLATE_BAD_SIZE;
            w.match(n, dist);
        }
    } while (!last);
    w.finish();
    // the stream must end where the member's payload ends, and fill the announced size
    const uint32_t consumed = br.pos - (br.cnt >> 3);
    if (consumed != in_len || op != out_len) return INFLATE_BAD_SIZE;
    return INFLATE_OK;
}

}  // namespace scginf

#endif
