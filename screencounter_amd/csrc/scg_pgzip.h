// scg_pgzip.h -- parallel decoding of ORDINARY gzip files (one or a few members: what a sequencer or `gzip` writes).
//
// The reference inflates such a file on the caller thread, 64 KiB at a time (byteme/GzipFileReader.hpp:39-51 over
// zlib's gzread; byteme/SomeFileReader.hpp:31-44 picks that reader from the magic bytes).  A DEFLATE stream has no
// index: a block can only be decoded after everything before it, because matches reach up to 32 KiB back into the
// text.  The two-stage scheme used here (the one pugz and rapidgzip made known) breaks that chain:
//
//  stage 1  the compressed stream is cut into chunks of a fixed compressed size; a host thread takes a chunk, finds
//           the first DEFLATE block that starts inside it (trial parsing of dynamic-Huffman block headers at every
//           bit offset: find_dynamic_block) and decodes from there to the first block boundary at or beyond the next
//           chunk's nominal start -- WITHOUT the 32 KiB of text before the chunk.  The output is 16-bit symbols:
//           0..255 a known byte, 0x8000 + k "byte k of the unknown window"; the output buffer starts with the
//           32 768 marker symbols themselves, so that a match which reaches into the window copies markers like any
//           other symbols and the decoder needs no special case.
//  stitch   in stream order, serially and cheaply: chunk j's result is accepted only if it begins exactly where the
//           decoding of chunk j - 1 ended (so every accepted chunk starts on a true block boundary, by induction from
//           the stream's known first block); the text window in front of chunk j + 1 is the last 32 KiB of chunk j's
//           symbols looked up in chunk j's window.  A chunk whose guess was wrong is decoded again from the right bit.
//  stage 2  symbols -> bytes through a 64 KiB table (the window behind the marker values), in parallel pieces that
//           write straight into the consumer's buffer, with the CRC-32 of every piece; the pieces' CRCs are combined
//           in order and checked against each member's trailer, as zlib does.
//
// Whatever does not fit -- a member that does not end the way RFC 1952 says, a code zlib would reject, reserved
// header bits, a ratio beyond the chunk buffers, a CRC or length mismatch -- ends the parallel attempt; the caller
// redoes the file with the sequential zlib path, whose verdict and message are the reference's.
//
// This header holds the single-threaded pieces (bit reader, table builder, marker-mode block decoder, block finder);
// plain C++ so that tests/pgzip_harness.cpp can run them against zlib under AddressSanitizer.  scg_pgzip.cpp holds
// the threads.
#ifndef SCG_PGZIP_H
#define SCG_PGZIP_H

#include <cstddef>
#include <cstdint>
#include <cstring>

namespace scg {
namespace pgz {

constexpr uint32_t WINDOW = 32768;              // DEFLATE's history
constexpr uint16_t MARKER = 0x8000;             // symbol MARKER + k = byte k of the unknown window (k = 0 oldest)
constexpr int LIT_BITS = 11;                    // primary table of the literal/length code
constexpr int DIST_BITS = 8;                    // primary table of the distance code
constexpr size_t IN_SLACK = 16;                 // bytes that must be readable beyond the end of the compressed data

// Table entry (uint32):  bits 0-4 bits to consume, bits 5-7 kind, bits 8-12 extra bits (SUB: bits of the subtable
// index), bits 16-31 payload: literal byte / base length / base distance / first entry of the subtable.
enum : uint32_t { K_INVALID = 0, K_LITERAL = 1, K_LENGTH = 2, K_END = 3, K_SUB = 4, K_DIST = 5 };
inline uint32_t make_entry(uint32_t nbits, uint32_t kind, uint32_t extra, uint32_t payload) {
    return nbits | (kind << 5) | (extra << 8) | (payload << 16);
}
inline uint32_t e_bits(uint32_t e) { return e & 31u; }
inline uint32_t e_kind(uint32_t e) { return (e >> 5) & 7u; }
inline uint32_t e_extra(uint32_t e) { return (e >> 8) & 31u; }
inline uint32_t e_payload(uint32_t e) { return e >> 16; }

// Worst cases: every subtable has 2^(15 - primary bits) entries and there are at most as many subtables as codes
// longer than the primary table.
constexpr size_t LIT_TABLE = (size_t(1) << LIT_BITS) + 288 * (size_t(1) << (15 - LIT_BITS));
constexpr size_t DIST_TABLE = (size_t(1) << DIST_BITS) + 32 * (size_t(1) << (15 - DIST_BITS));

struct Tables {
    uint32_t lit[LIT_TABLE];
    uint32_t dist[DIST_TABLE];
};

inline uint64_t load64(const uint8_t* p) { uint64_t v; std::memcpy(&v, p, 8); return v; }

// LSB-first bit reader over in[0 .. size + IN_SLACK).  `bitpos()` is the position of the next unread bit.
struct Bits {
    const uint8_t* in;
    size_t size;          // bytes of real data
    size_t pos;           // byte position of the next byte to enter buf
    uint64_t buf;
    uint32_t cnt;

    void open(const uint8_t* p, size_t n, uint64_t bit) {
        in = p; size = n;
        pos = static_cast<size_t>(bit >> 3);
        buf = 0; cnt = 0;
        refill();
        const uint32_t skip = static_cast<uint32_t>(bit & 7u);
        buf >>= skip; cnt -= skip;
    }
    // At least 56 bits afterwards (zeros beyond the data: overrun() tells).
    void refill() {
        if (pos + 8 <= size + IN_SLACK) {
            buf |= load64(in + pos) << cnt;
            const uint32_t take = (63u - cnt) >> 3;
            pos += take;
            cnt += take * 8u;
        }
    }
    uint32_t peek(uint32_t n) const { return static_cast<uint32_t>(buf) & ((1u << n) - 1u); }
    void drop(uint32_t n) { buf >>= n; cnt -= n; }
    uint32_t take(uint32_t n) { const uint32_t v = peek(n); drop(n); return v; }
    uint64_t bitpos() const { return static_cast<uint64_t>(pos) * 8u - cnt; }
    bool overrun() const { return bitpos() > static_cast<uint64_t>(size) * 8u; }
};

// Canonical Huffman code -> two-level decoding table.  Rules as in zlib's inflate_table: an over-subscribed set is
// rejected; an incomplete one too, except a literal/length or distance code whose only code has one bit
// (`allow_single`).  Returns false for a set zlib rejects.
//   kind_of(symbol, &extra, &payload) gives the entry kind of a symbol.
template<class KindOf>
inline bool build_table(const uint8_t* lens, int n, int tbits, uint32_t* table, size_t cap, bool allow_single, KindOf kind_of) {
    uint16_t count[16] = {0};
    for (int s = 0; s < n; ++s) ++count[lens[s]];
    count[0] = 0;
    int maxlen = 15;
    while (maxlen >= 1 && count[maxlen] == 0) --maxlen;
    const size_t primary = size_t(1) << tbits;
    for (size_t i = 0; i < primary; ++i) table[i] = 0;
    if (maxlen < 1) return true;                        // no codes: every lookup fails, which is zlib's error when one is used
    int left = 1;
    for (int l = 1; l <= 15; ++l) {
        left <<= 1;
        left -= count[l];
        if (left < 0) return false;
    }
    if (left > 0 && !(allow_single && maxlen == 1)) return false;
    uint16_t next[16];
    {
        uint32_t code = 0;
        for (int l = 1; l <= 15; ++l) { code = (code + count[l - 1]) << 1; next[l] = static_cast<uint16_t>(code); }
    }
    const int subbits = maxlen > tbits ? maxlen - tbits : 0;
    size_t used = primary;
    for (int s = 0; s < n; ++s) {
        const int l = lens[s];
        if (!l) continue;
        uint32_t code = next[l]++;
        uint32_t rev = 0;                               // DEFLATE packs Huffman codes MSB first into an LSB-first stream
        for (int i = 0; i < l; ++i) { rev = (rev << 1) | (code & 1u); code >>= 1; }
        uint32_t extra = 0, payload = 0;
        const uint32_t kind = kind_of(s, extra, payload);
        if (l <= tbits) {
            const uint32_t e = make_entry(static_cast<uint32_t>(l), kind, extra, payload);
            for (size_t i = rev; i < primary; i += size_t(1) << l) table[i] = e;
        } else {
            const uint32_t low = rev & static_cast<uint32_t>(primary - 1);
            if (e_kind(table[low]) != K_SUB) {
                if (used + (size_t(1) << subbits) > cap) return false;
                table[low] = make_entry(static_cast<uint32_t>(tbits), K_SUB, static_cast<uint32_t>(subbits), static_cast<uint32_t>(used));
                for (size_t i = 0; i < (size_t(1) << subbits); ++i) table[used + i] = 0;
                used += size_t(1) << subbits;
            }
            const uint32_t base = e_payload(table[low]);
            const uint32_t e = make_entry(static_cast<uint32_t>(l - tbits), kind, extra, payload);
            for (size_t i = rev >> tbits; i < (size_t(1) << subbits); i += size_t(1) << (l - tbits)) table[base + i] = e;
        }
    }
    return true;
}

inline uint32_t litlen_kind(int s, uint32_t& extra, uint32_t& payload) {
    static const uint16_t base[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    static const uint8_t ext[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    if (s < 256) { payload = static_cast<uint32_t>(s); return K_LITERAL; }
    if (s == 256) return K_END;
    if (s > 285) return K_INVALID;                      // 286, 287 take part in the fixed code but never occur ("invalid literal/length code")
    extra = ext[s - 257]; payload = base[s - 257];
    return K_LENGTH;
}
inline uint32_t dist_kind(int s, uint32_t& extra, uint32_t& payload) {
    static const uint16_t base[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145,
                                      8193, 12289, 16385, 24577};
    static const uint8_t ext[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
    if (s > 29) return K_INVALID;                       // "invalid distance code"
    extra = ext[s]; payload = base[s];
    return K_DIST;
}

enum : int { BLOCK_OK = 0, BLOCK_FINAL = 1, BLOCK_BAD = -1, BLOCK_FULL = -2 };

// Reads a dynamic block's code lengths (the bits behind the 3-bit block header) and builds both tables.
// lens: scratch of 320 bytes.  Every rule of zlib's inflate for this part of the stream (inflate.c: TABLE .. CODELENS).
inline bool read_dynamic_header(Bits& br, Tables& T, uint8_t* lens) {
    br.refill();
    const int nlen = static_cast<int>(br.take(5)) + 257;
    const int ndist = static_cast<int>(br.take(5)) + 1;
    const int ncode = static_cast<int>(br.take(4)) + 4;
    if (nlen > 286 || ndist > 30) return false;                          // "too many length or distance symbols"
    static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    uint8_t cl[19] = {0};
    for (int i = 0; i < ncode; ++i) {
        if ((i & 15) == 0) br.refill();
        cl[order[i]] = static_cast<uint8_t>(br.take(3));
    }
    // the code-length code: 7-bit table, never longer
    uint32_t cltab[128];
    int left = 1;
    {
        uint16_t count[8] = {0};
        for (int i = 0; i < 19; ++i) ++count[cl[i]];
        count[0] = 0;
        for (int l = 1; l <= 7; ++l) { left <<= 1; left -= count[l]; if (left < 0) return false; }
        if (left > 0) return false;                                      // "invalid code lengths set" (incomplete sets are not allowed here)
    }
    if (!build_table(cl, 19, 7, cltab, 128, false, [](int s, uint32_t&, uint32_t& payload) { payload = static_cast<uint32_t>(s); return uint32_t(K_LITERAL); })) return false;
    int have = 0;
    while (have < nlen + ndist) {
        br.refill();
        if (br.overrun()) return false;
        const uint32_t e = cltab[br.peek(7)];
        if (e_kind(e) != K_LITERAL) return false;
        br.drop(e_bits(e));
        const int s = static_cast<int>(e_payload(e));
        if (s < 16) { lens[have++] = static_cast<uint8_t>(s); continue; }
        uint8_t fill = 0;
        int rep;
        if (s == 16) {
            if (have == 0) return false;                                 // "invalid bit length repeat"
            fill = lens[have - 1];
            rep = 3 + static_cast<int>(br.take(2));
        } else if (s == 17) {
            rep = 3 + static_cast<int>(br.take(3));
        } else {
            rep = 11 + static_cast<int>(br.take(7));
        }
        if (have + rep > nlen + ndist) return false;                     // "invalid bit length repeat"
        while (rep--) lens[have++] = fill;
    }
    if (lens[256] == 0) return false;                                    // "invalid code -- missing end-of-block"
    if (!build_table(lens, nlen, LIT_BITS, T.lit, LIT_TABLE, true, litlen_kind)) return false;          // "invalid literal/lengths set"
    if (!build_table(lens + nlen, ndist, DIST_BITS, T.dist, DIST_TABLE, true, dist_kind)) return false;  // "invalid distances set"
    return true;
}

inline void fixed_tables(Tables& T) {
    uint8_t lens[320];
    for (int s = 0; s < 144; ++s) lens[s] = 8;
    for (int s = 144; s < 256; ++s) lens[s] = 9;
    for (int s = 256; s < 280; ++s) lens[s] = 7;
    for (int s = 280; s < 288; ++s) lens[s] = 8;
    for (int s = 0; s < 32; ++s) lens[288 + s] = 5;
    build_table(lens, 288, LIT_BITS, T.lit, LIT_TABLE, true, litlen_kind);
    build_table(lens + 288, 32, DIST_BITS, T.dist, DIST_TABLE, true, dist_kind);     // (codes 30, 31 decode to K_INVALID)
}

// out[0 .. n) = out[-dist .. ), overlapping the way LZ77 means it; may write up to 15 symbols beyond n.
inline void copy_match(uint16_t* out, uint32_t dist, uint32_t n) {
    const uint16_t* src = out - dist;
    if (dist >= 8) {                                     // 16-byte steps never read what they are about to write
        uint32_t k = 0;
        do {
            std::memcpy(out + k, src + k, 16);
            k += 8;
        } while (k < n);
    } else if (dist >= 4) {
        for (uint32_t k = 0; k < n; k += 4) std::memcpy(out + k, src + k, 8);
    } else if (dist == 1) {
        uint64_t v = src[0];
        v |= v << 16; v |= v << 32;
        for (uint32_t k = 0; k < n; k += 4) std::memcpy(out + k, &v, 8);
    } else {
        for (uint32_t k = 0; k < n; ++k) out[k] = src[k];
    }
}

// Decodes ONE block whose 3 header bits are next in `br`, appending 16-bit symbols at out[op ...).
//   cap       symbols that fit the buffer (a block that would not fit: BLOCK_FULL)
//   reach     how far back a match may reach from position op = 0 (WINDOW when the text before the chunk is unknown,
//             0 at the start of a member: "invalid distance too far back")
// Returns BLOCK_OK / BLOCK_FINAL (the member's last block) / BLOCK_BAD / BLOCK_FULL.
//   COUNT / n_symbols: measurement aid (tools/deflate_symbols.cpp): the number of literal, match and end-of-block symbols
template<bool COUNT = false>
inline int decode_block(Bits& br, Tables& T, uint16_t* out, size_t& op, size_t cap, size_t reach, uint8_t* lens, size_t* n_symbols = nullptr) {
    br.refill();
    if (br.overrun()) return BLOCK_BAD;
    const uint32_t last = br.take(1);
    const uint32_t type = br.take(2);
    if (type == 3) return BLOCK_BAD;                                     // "invalid block type"
    if (type == 0) {
        br.drop(br.cnt & 7u);                                            // to the byte boundary
        br.refill();
        const uint32_t n = br.take(16), nn = br.take(16);
        if ((n ^ 0xFFFFu) != nn) return BLOCK_BAD;                       // "invalid stored block lengths"
        if (n > cap - op) return BLOCK_FULL;
        const uint64_t at = br.bitpos();                                 // byte aligned
        const size_t from = static_cast<size_t>(at >> 3);
        if (from > br.size || n > br.size - from) return BLOCK_BAD;
        for (uint32_t k = 0; k < n; ++k) out[op + k] = br.in[from + k];
        op += n;
        br.open(br.in, br.size, at + static_cast<uint64_t>(n) * 8u);
        return last ? BLOCK_FINAL : BLOCK_OK;
    }
    if (type == 1) fixed_tables(T);
    else if (!read_dynamic_header(br, T, lens)) return BLOCK_BAD;
    const uint32_t lit_mask = (1u << LIT_BITS) - 1u, dist_mask = (1u << DIST_BITS) - 1u;
    // The reader's state in locals for the loop: the compiler keeps them in registers and the stores of symbols cannot
    // be taken to alias them.
    const uint8_t* const in = br.in;
    const size_t in_limit = br.size + IN_SLACK;
    size_t pos = br.pos;
    uint64_t buf = br.buf;
    uint32_t cnt = br.cnt;
    size_t o = op;
    int rc = BLOCK_BAD;
#define SCG_PGZ_REFILL()                                         \
    do {                                                          \
        if (pos + 8 <= in_limit) {                                \
            buf |= load64(in + pos) << cnt;                       \
            const uint32_t take_ = (63u - cnt) >> 3;              \
            pos += take_;                                         \
            cnt += take_ * 8u;                                    \
        }                                                         \
    } while (0)
#define SCG_PGZ_OVERRUN() (static_cast<uint64_t>(pos) * 8u - cnt > static_cast<uint64_t>(br.size) * 8u)
    // (tried: a second table giving up to four literals per look-up for the runs of two- and three-bit codes of sequence
    // lines -- 10 % slower: 16 KB more of tables to rebuild and keep in the L1 for every block)
    for (;;) {
        SCG_PGZ_REFILL();                                                // >= 56 bits: a whole length/distance pair (<= 48)
        uint32_t e = T.lit[static_cast<uint32_t>(buf) & lit_mask];
        if (e_kind(e) == K_SUB) {
            buf >>= LIT_BITS; cnt -= LIT_BITS;
            e = T.lit[e_payload(e) + (static_cast<uint32_t>(buf) & ((1u << e_extra(e)) - 1u))];
        }
        buf >>= e_bits(e); cnt -= e_bits(e);
        const uint32_t kind = e_kind(e);
        if (kind == K_LITERAL) {
            if (o + 4 > cap) {
                if (o >= cap) { rc = BLOCK_FULL; break; }
                out[o++] = static_cast<uint16_t>(e_payload(e));
                if (COUNT) ++*n_symbols;
                continue;
            }
            out[o++] = static_cast<uint16_t>(e_payload(e));
            if (COUNT) ++*n_symbols;
            // up to three more literals from the same refill (11 primary bits each at most, 56 in the buffer)
            e = T.lit[static_cast<uint32_t>(buf) & lit_mask];
            if (e_kind(e) != K_LITERAL) continue;
            buf >>= e_bits(e); cnt -= e_bits(e);
            out[o++] = static_cast<uint16_t>(e_payload(e));
            if (COUNT) ++*n_symbols;
            e = T.lit[static_cast<uint32_t>(buf) & lit_mask];
            if (e_kind(e) != K_LITERAL) continue;
            buf >>= e_bits(e); cnt -= e_bits(e);
            out[o++] = static_cast<uint16_t>(e_payload(e));
            if (COUNT) ++*n_symbols;
            e = T.lit[static_cast<uint32_t>(buf) & lit_mask];
            if (e_kind(e) != K_LITERAL) continue;
            buf >>= e_bits(e); cnt -= e_bits(e);
            out[o++] = static_cast<uint16_t>(e_payload(e));
            if (COUNT) ++*n_symbols;
            continue;
        }
        if (kind == K_LENGTH) {
            const uint32_t n = e_payload(e) + (static_cast<uint32_t>(buf) & ((1u << e_extra(e)) - 1u));
            buf >>= e_extra(e); cnt -= e_extra(e);
            uint32_t d = T.dist[static_cast<uint32_t>(buf) & dist_mask];
            if (e_kind(d) == K_SUB) {
                buf >>= DIST_BITS; cnt -= DIST_BITS;
                d = T.dist[e_payload(d) + (static_cast<uint32_t>(buf) & ((1u << e_extra(d)) - 1u))];
            }
            if (e_kind(d) != K_DIST) { rc = BLOCK_BAD; break; }          // "invalid distance code"
            buf >>= e_bits(d); cnt -= e_bits(d);
            const uint32_t dist = e_payload(d) + (static_cast<uint32_t>(buf) & ((1u << e_extra(d)) - 1u));
            buf >>= e_extra(d); cnt -= e_extra(d);
            if (dist > o + reach) { rc = BLOCK_BAD; break; }             // "invalid distance too far back"
            if (n + 16 > cap - o) {
                if (n > cap - o) { rc = BLOCK_FULL; break; }
                for (uint32_t k = 0; k < n; ++k) out[o + k] = out[o + k - dist];
            } else {
                copy_match(out + o, dist, n);
            }
            o += n;
            if (COUNT) ++*n_symbols;
            if (SCG_PGZ_OVERRUN()) { rc = BLOCK_BAD; break; }
            continue;
        }
        if (kind == K_END) {
            if (COUNT) ++*n_symbols;
            rc = SCG_PGZ_OVERRUN() ? BLOCK_BAD : (last ? BLOCK_FINAL : BLOCK_OK);
            break;
        }
        rc = BLOCK_BAD;                                                  // "invalid literal/length code"
        break;
    }
#undef SCG_PGZ_REFILL
#undef SCG_PGZ_OVERRUN
    br.pos = pos; br.buf = buf; br.cnt = cnt;
    op = o;
    return rc;
}

// The first bit position in [from, to) at which a non-final dynamic-Huffman block header parses without error, or
// ~0 if there is none.  (Stored and fixed blocks carry nothing to recognise them by: a chunk that begins with them
// is found one block later, or handed to the stitching pass.)
inline uint64_t find_dynamic_block(const uint8_t* in, size_t size, uint64_t from, uint64_t to, Tables& T, uint8_t* lens) {
    const uint64_t end = static_cast<uint64_t>(size) * 8u;
    if (to > end) to = end;
    for (uint64_t bit = from; bit + 17 <= to; ++bit) {
        const size_t byte = static_cast<size_t>(bit >> 3);
        const uint64_t v = load64(in + byte) >> (bit & 7u);              // >= 57 bits
        // BFINAL = 0, BTYPE = 10 (LSB first: 0, then 0 1), HLIT <= 29, HDIST <= 29
        if ((v & 7u) != 4u) continue;
        if (((v >> 3) & 31u) > 29u || ((v >> 8) & 31u) > 29u) continue;
        // the code-length code must be complete: Kraft sum over up to 19 three-bit lengths (the first 13 sit in v)
        const int ncode = static_cast<int>((v >> 13) & 15u) + 4;
        uint32_t kraft = 0;
        {
            uint64_t w = v >> 17;
            int i = 0;
            for (; i < ncode && i < 13; ++i) { const uint32_t l = static_cast<uint32_t>(w & 7u); w >>= 3; if (l) kraft += 128u >> l; }
            if (kraft > 128u) continue;
            if (i < ncode) {
                const uint64_t bit2 = bit + 17 + 39;
                uint64_t w2 = load64(in + static_cast<size_t>(bit2 >> 3)) >> (bit2 & 7u);
                for (; i < ncode; ++i) { const uint32_t l = static_cast<uint32_t>(w2 & 7u); w2 >>= 3; if (l) kraft += 128u >> l; }
            }
        }
        if (kraft != 128u) continue;
        Bits br;
        br.open(in, size, bit + 3);
        if (read_dynamic_header(br, T, lens) && !br.overrun()) return bit;
    }
    return ~uint64_t(0);
}

}  // namespace pgz
}  // namespace scg

#endif
