// scg_inflate.hip -- BGZF members inflated on the device: one member per wavefront (scg_inflate.h), Huffman tables in LDS.
//
// A window of a BGZF file is a few thousand independent gzip members of <= 64 KiB of text each.  Their compressed
// bytes are shipped as they are (a fifth of the text), every wavefront of inflate_members_kernel decodes one member straight
// into its place in the window's text buffer in HBM, crc_members_kernel checks every member's CRC-32 against its
// trailer (one workgroup per member; the pieces are combined in GF(2) like zlib's crc32_combine), and the record scan
// (scg_textscan.hip) runs on the text where it lies.  The host never sees the text: its threads -- sixteen zlib streams
// were the bound of BGZF input, 23 Mreads/s -- only copy compressed bytes.
//
// Windows are cut at member boundaries, not at record boundaries, so the text of a window ends in a partial record.
// The scan reports where the last whole record ends (TextScanResult::cut); carry_tail_kernel moves the rest to the
// front of the next window's text.  That front is a gap of fixed size: what the tail does not fill is laid out as one
// dummy record with an empty sequence ("@xxx...\n\n+\n\n"), so that the text still starts at a record start, the first
// real line is checked like every other, and the counting kernels simply skip record 0.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "scg_inflate.h"
#include "scg_textscan.h"

namespace {

// One wavefront per member.  DEFLATE decoding is a chain of dependent steps full of data-dependent branches: with one
// member per LANE every lane that took a different path made the other 63 wait, and a window of 4 000 members took
// 44 ms (1.5 MB/s per lane).  A wavefront that walks one member together keeps the decoder's state in scalar registers,
// never diverges, and uses its lanes where bytes move (64-byte copies); the machine has room for 8 192 wavefronts, more
// than a window has members.
constexpr int INFLATE_BLOCK = 64;

// The wavefront as scg_inflate.h's decoder sees it: a Vec is one register, lane j holds byte j.
struct WaveLanes {
    typedef uint32_t Vec;
    uint32_t id;
    __device__ __forceinline__ uint32_t width() const { return INFLATE_BLOCK; }
    __device__ __forceinline__ void set(Vec& v, uint32_t j, uint32_t byte) const { if (id == j) v = byte; }
    // j mod dist for this lane (dist is wave-uniform; matches that reach into their own output are the rare case)
    __device__ __forceinline__ uint32_t wrap(uint32_t j, uint32_t n, uint32_t dist) const { return dist >= n ? j : (dist == 1 ? 0u : j % dist); }
    // The lanes hand bytes to one another through the text itself: what one lane stores, another may load a few symbols
    // later.  gfx9 issues a wavefront's vector memory operations in order, so that works as it stands; the
    // wavefront-scope fences (no instructions on this target) keep the COMPILER from moving a load of the text across
    // an earlier store of it.
    static __device__ __forceinline__ void lanes_published() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); }
    static __device__ __forceinline__ void lanes_may_read() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }
    __device__ __forceinline__ void load(Vec& v, const uint8_t* src, uint32_t n, uint32_t dist) const {
        lanes_may_read();
        if (id < n) v = src[wrap(id, n, dist)];
    }
    __device__ __forceinline__ void store(uint8_t* dst, const Vec& v, uint32_t n) const {
        if (id < n) dst[id] = static_cast<uint8_t>(v);
        lanes_published();
    }
    __device__ __forceinline__ void copy(uint8_t* dst, const uint8_t* src, uint32_t n, uint32_t dist) const {
        lanes_may_read();
        for (uint32_t j = id; j < n; j += INFLATE_BLOCK) dst[j] = src[wrap(j, n, dist)];
        lanes_published();
    }
};

__global__ __launch_bounds__(INFLATE_BLOCK) void inflate_members_kernel(const uint8_t* __restrict__ in, const scg::InflateMember* __restrict__ members,
                                                                        uint32_t n, uint8_t* text, uint32_t* __restrict__ status) {
    __shared__ scginf::LaneTables tables;
    const uint32_t m = blockIdx.x;
    if (m >= n) return;
    const scg::InflateMember M = members[m];
    const int rc = scginf::inflate_member(in + M.in_off, M.in_len, text + M.out_off, M.out_len, tables, WaveLanes{threadIdx.x});
    if (rc != scginf::INFLATE_OK && threadIdx.x == 0) atomicOr(status, scg::INFLATE_STATUS_BAD);
}

// ---------------------------------------------------------------------------------------------------------------
// The lane-parallel decoder (default).  One wavefront per member as before, but the symbols of a block are no longer
// decoded one after the other by the wavefront as a whole -- 60 scalar instructions a symbol, and a CU issues one
// scalar instruction per cycle for all its wavefronts -- but 128 bit offsets at a time, two per lane:
//
//   1. lane i decodes the literal/length code (and, for a length, the distance code) that WOULD start at bit
//      `bitpos + i` of the stream: two look-ups in the LDS tables and some vector arithmetic, the same for every lane;
//      most of these offsets are no code starts and their results are thrown away;
//   2. the true code starts are the chain 0 -> next(0) -> next(next(0)) ...: one v_readlane per symbol walks it (8 scalar
//      instructions a symbol, one loop exit), and on the way every symbol's place in the text is fixed;
//   3. all literals of the batch are stored at once, one byte per lane; the matches are only NOTED (place, length,
//      distance, in LDS) and the next batch is decoded at once: the decoder never needs a match's bytes to go on, and
//      waiting for each copy's load -- a round trip to HBM per match, three matches a batch -- was 70 % of the
//      wavefront's time.  When 64 or more are noted (and at block ends) resolve_matches copies them, one match per
//      lane, in rounds: whatever reads only text that is final goes in this round.
// The compressed bytes come through a 1 KB ring in LDS, fetched 512 bytes at a time one fetch ahead (InRing): the 64
// overlapping 8-byte reads of a batch were a round trip to HBM / L2 per batch.
// Measured (profiles/r3_bgzf_pmc.txt; 16 M reads of the benchmark stream, 256 MB windows): 7.6 ms -> 3.5 ms per window,
// 66 -> 24 scalar and 35 -> 31 vector instructions per symbol; random-sequence reads 13.0 -> 6.4 ms.  (64 offsets a
// batch: 3.9 / 6.8 ms; the per-symbol work -- one chain step, its share of 14 offsets' decoding -- is what is left.)
// Codes longer than the primary tables, an end-of-block or an invalid pattern stop the chain; that one symbol is
// decoded the old way (scginf::decode_symbol: canonical bit-by-bit decoding) and the batches go on behind it.
// Block headers (once per ~30 KB of text) are read by the wavefront as a whole, as before.  Accept / reject rules are
// scg_inflate.h's; what this decoder gets wrong on a corrupt stream the CRC kernel catches, and zlib judges the file.
// ---------------------------------------------------------------------------------------------------------------
constexpr int LANES_LIT_BITS = 10, LANES_DIST_BITS = 8;      // (11 / 9 bits: no gain -- long codes are rare in FASTQ)
struct WaveTables {                   // 3.3 KB of LDS per wavefront
    uint16_t lit[1 << LANES_LIT_BITS];
    uint16_t dtab[1 << LANES_DIST_BITS];
    uint16_t lsym[288];
    uint16_t dsym[32];
    uint16_t lcount[16];
    uint16_t dcount[16];
    uint16_t offs[16];
};

// What a wavefront keeps next to its tables (2 KB): a ring of the compressed bytes around the read position, so that a
// batch's 64 overlapping 8-byte reads come from LDS and the stream is fetched from HBM 512 bytes at a time, one fetch
// ahead of its use; and the matches that have been decoded but not copied yet (see resolve_matches).
constexpr uint32_t RING_BYTES = 1024, RING_HALF = 512, MATCH_SLOTS = 128;
struct WaveStage {
    uint32_t ring[RING_BYTES / 4];
    uint32_t tok_at[MATCH_SLOTS];      // place in the text
    uint32_t tok_dist[MATCH_SLOTS];    // distance | length << 16
};

__device__ __forceinline__ uint32_t rdlane(uint32_t v, uint32_t lane) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)lane); }
__device__ __forceinline__ uint64_t ballot64(bool p) { return __ballot(p); }
__device__ __forceinline__ uint32_t rank_below(uint64_t m) {      // set bits of m below this lane
    return __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(m >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(m), 0u));
}

// The ring of compressed bytes.  [lo, hi) of the member's payload is in LDS (hi - lo == RING_BYTES, both multiples of
// RING_HALF); `ahead` is this lane's 8 bytes of [hi, hi + RING_HALF), requested when the half before it was committed.
struct InRing {
    const uint8_t* in;
    uint32_t limit;                    // bytes of `in` that may be read (payload + IN_SLACK)
    uint32_t lo, hi;
    uint64_t ahead;
    __device__ __forceinline__ uint64_t fetch(uint32_t at) const {
        uint64_t w = 0;
        if (at + 8u <= limit) __builtin_memcpy(&w, in + at, 8);
        return w;
    }
    __device__ __forceinline__ void put(WaveStage& S, uint32_t at, uint64_t w) const {
        const uint32_t q = (at >> 2) & (RING_BYTES / 4 - 1);
        S.ring[q] = static_cast<uint32_t>(w);
        S.ring[q + 1] = static_cast<uint32_t>(w >> 32);
    }
    // Ring around byte `at` (the block's first symbol, or wherever the bitwise decoder left off).
    __device__ __forceinline__ void reset(WaveStage& S, uint32_t at, uint32_t lane) {
        lo = at & ~(RING_HALF - 1);
        hi = lo + RING_BYTES;
        const uint64_t a = fetch(lo + 8 * lane), b = fetch(lo + RING_HALF + 8 * lane);
        ahead = fetch(hi + 8 * lane);
        put(S, lo + 8 * lane, a);
        put(S, lo + RING_HALF + 8 * lane, b);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    }
    // Bytes [at, at + 28) are in the ring afterwards (a batch of 128 bit offsets reads that far); the position only moves forward.
    __device__ __forceinline__ void ensure(WaveStage& S, uint32_t at, uint32_t lane) {
        if (at + 28u <= hi && at >= lo) return;
        if (at < lo || at + 28u > hi + RING_HALF) { reset(S, at, lane); return; }
        put(S, hi + 8 * lane, ahead);                          // replaces the half the position has left behind
        lo += RING_HALF;
        hi += RING_HALF;
        ahead = fetch(hi + 8 * lane);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    }
    // 64 bits of the stream from bit `b` on (>= 57 of them valid).
    __device__ __forceinline__ uint64_t bits_at(const WaveStage& S, uint32_t b) const {
        const uint32_t byte = b >> 3, q = byte >> 2;
        const uint32_t x0 = S.ring[q & (RING_BYTES / 4 - 1)], x1 = S.ring[(q + 1) & (RING_BYTES / 4 - 1)], x2 = S.ring[(q + 2) & (RING_BYTES / 4 - 1)];
        const uint32_t sh = (byte & 3u) * 8u + (b & 7u);       // < 32
        const uint64_t low = (static_cast<uint64_t>(x1) << 32) | x0;
        return (low >> sh) | ((static_cast<uint64_t>(x2) << 1) << (63u - sh));     // (no branch for sh == 0)
    }
};

// The decoder writes bytes (BGZF members: the text itself) or 16-bit symbols (chunks of an ordinary gzip stream, decoded
// without the 32 KiB of text in front of them: a symbol is a byte, or MARKER + k for "byte k of the window I do not have";
// scg_pgzip.h).  In symbol mode a match may reach in front of the chunk's first symbol: those positions read as markers.
constexpr uint32_t MARKER = 0x8000u, MARKER_WINDOW = 32768u;

// One match, copied by one lane.  A source at least 8 bytes back is moved in 8-byte words (up to four loads in flight);
// a closer one is a pattern of period `d`, built once in a register and stored over and over.
__device__ __forceinline__ void copy_match(uint8_t* out, uint32_t at, uint32_t len, uint32_t d) {
    uint8_t* dst = out + at;
    const uint8_t* src = dst - d;
    uint32_t j = 0;
    uint64_t w;
    if (d >= 8) {
        if (d >= 32) {
            for (; j + 32 <= len; j += 32) {
                uint64_t x[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) __builtin_memcpy(&x[k], src + j + 8 * k, 8);
#pragma unroll
                for (int k = 0; k < 4; ++k) __builtin_memcpy(dst + j + 8 * k, &x[k], 8);
            }
        }
        for (; j + 8 <= len; j += 8) {
            __builtin_memcpy(&w, src + j, 8);
            __builtin_memcpy(dst + j, &w, 8);
        }
        if (j < len) __builtin_memcpy(&w, src + j, 8);          // (ends at most where this match's own text begins)
    } else {
        __builtin_memcpy(&w, src, 8);                           // the d bytes of the pattern and 8 - d that are not text yet
        w &= ~0ull >> (64u - 8u * d);
        for (uint32_t have = d; have < 8; have <<= 1) w |= w << (8u * have);
        const uint32_t step = d == 3 ? 6u : (d >= 5 ? d : 8u);  // the largest multiple of d within a word
        for (; j + 8 <= len; j += step) __builtin_memcpy(dst + j, &w, 8);
    }
    for (uint32_t k = 0; j + k < len; ++k) dst[j + k] = static_cast<uint8_t>(w >> (8u * k));
}
// The same in symbols (four to a word); a match that begins in front of the chunk goes symbol by symbol.
__device__ __forceinline__ void copy_match(uint16_t* out, uint32_t at, uint32_t len, uint32_t d) {
    uint16_t* dst = out + at;
    if (at < d) {
        for (uint32_t j = 0; j < len; ++j) {
            const int32_t from = static_cast<int32_t>(at + j) - static_cast<int32_t>(d);
            dst[j] = from < 0 ? static_cast<uint16_t>(MARKER + MARKER_WINDOW + from) : out[from];
        }
        return;
    }
    const uint16_t* src = dst - d;
    uint32_t j = 0;
    uint64_t w;
    if (d >= 4) {
        if (d >= 16) {
            for (; j + 16 <= len; j += 16) {
                uint64_t x[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) __builtin_memcpy(&x[k], src + j + 4 * k, 8);
#pragma unroll
                for (int k = 0; k < 4; ++k) __builtin_memcpy(dst + j + 4 * k, &x[k], 8);
            }
        }
        for (; j + 4 <= len; j += 4) {
            __builtin_memcpy(&w, src + j, 8);
            __builtin_memcpy(dst + j, &w, 8);
        }
        if (j < len) __builtin_memcpy(&w, src + j, 8);
    } else {
        __builtin_memcpy(&w, src, 8);
        w &= ~0ull >> (64u - 16u * d);
        for (uint32_t have = d; have < 4; have <<= 1) w |= w << (16u * have);
        const uint32_t step = d == 3 ? 3u : 4u;
        for (; j + 4 <= len; j += step) __builtin_memcpy(dst + j, &w, 8);
    }
    for (uint32_t k = 0; j + k < len; ++k) dst[j + k] = static_cast<uint16_t>(w >> (16u * k));
}

// Copies the nm matches of S.tok_* in rounds of up to 64, one match per lane.  A match may go when everything it reads
// is final: the text in front of the earliest match that has not been copied yet (all literals are in place already,
// and matches before that one are done).  The earliest pending match always goes (it reads only text in front of
// itself), so every round retires at least one; FASTQ's typical sources -- the record before, or far back -- retire
// 30 or so per round, i.e. per memory round trip, where copying them one by one took a round trip each.
template<class OutT>
__device__ __forceinline__ void resolve_matches(OutT* out, const WaveStage& S, uint32_t nm, uint32_t lane) {
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (uint32_t g = 0; g < nm; g += INFLATE_BLOCK) {
        const uint32_t idx = g + lane;
        const bool have = idx < nm;
        const uint32_t at = have ? S.tok_at[idx] : 0u, ld = have ? S.tok_dist[idx] : 1u;
        const uint32_t d = ld & 0xFFFFu, len = ld >> 16;
        // (in symbol mode a source may begin in front of the chunk: what lies there is final by definition)
        const int32_t src_end = static_cast<int32_t>(at) - static_cast<int32_t>(d) + static_cast<int32_t>(len < d ? len : d);
        uint64_t pending = ballot64(have);
        while (pending) {
            const uint32_t first = static_cast<uint32_t>(__builtin_ctzll(pending));
            const uint32_t final_to = rdlane(at, first);
            const bool go = ((pending >> lane) & 1ull) && (src_end <= static_cast<int32_t>(final_to) || lane == first);
            if (go) copy_match(out, at, len, d);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            pending &= ~ballot64(go);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    }
}

// n elements to out[at ...) from dist back, by all lanes (a match met by the bitwise decoder).
template<class OutT>
__device__ __forceinline__ void wave_copy(OutT* out, uint32_t at, uint32_t n, uint32_t dist, uint32_t lane) {
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (uint32_t j = lane; j < n; j += INFLATE_BLOCK) {
        const uint32_t k = dist >= n ? j : (dist == 1 ? 0u : j % dist);
        const int32_t from = static_cast<int32_t>(at + k) - static_cast<int32_t>(dist);
        out[at + j] = (sizeof(OutT) == 2 && from < 0) ? static_cast<OutT>(MARKER + MARKER_WINDOW + from) : out[from];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
}
// n bytes of a stored block, by all lanes.
template<class OutT>
__device__ __forceinline__ void wave_store(OutT* out, uint32_t at, const uint8_t* src, uint32_t n, uint32_t lane) {
    for (uint32_t j = lane; j < n; j += INFLATE_BLOCK) out[at + j] = src[j];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
}

// What the code starting at the low end of `w` would be (>= 57 valid bits), were it a code start.
// kind: 0 literal, 1 match, 2 end of block, 3 "not decodable here" (long code / invalid: the chain stops)
struct LaneCode {
    uint32_t kind, adv, outlen, n, dist, sym;
};
__device__ __forceinline__ LaneCode decode_here(const WaveTables& T, uint64_t w) {
    // Straight-line: every lane goes through the length and the distance arithmetic, whatever its code is, and the
    // results are selected at the end.  As nested branches this cost 13 exec-mask save / restore pairs and 40 register
    // moves per 64 offsets -- scalar instructions the whole wavefront waits for, most of them for offsets that are no code
    // starts anyway.
    LaneCode c;
    const uint32_t e = T.lit[static_cast<uint32_t>(w) & ((1u << LANES_LIT_BITS) - 1u)];
    const uint32_t l = e >> 12;
    c.sym = e & 0xFFFu;
    const uint32_t t0 = c.sym - 257u;                                   // (wraps for literals: clamped, and not used then)
    const uint32_t t = t0 < 28u ? t0 : 28u;
    const bool plain = t < 8u || t == 28u;
    const uint32_t eb = plain ? 0u : (t - 4u) >> 2;
    const uint32_t base = t < 8u ? t + 3u : (t == 28u ? 258u : ((4u + (t & 3u)) << eb) + 3u);
    uint64_t w2 = w >> l;
    c.n = base + (static_cast<uint32_t>(w2) & ((1u << eb) - 1u));
    w2 >>= eb;
    const uint32_t de = T.dtab[static_cast<uint32_t>(w2) & ((1u << LANES_DIST_BITS) - 1u)];
    const uint32_t dl = de >> 12, ds0 = de & 0xFFFu;
    const uint32_t ds = ds0 < 29u ? ds0 : 29u;
    const uint32_t deb = ds < 4u ? 0u : (ds >> 1) - 1u;
    const uint32_t dbase = ds < 4u ? ds + 1u : ((2u + (ds & 1u)) << deb) + 1u;
    w2 >>= dl;
    c.dist = dbase + (static_cast<uint32_t>(w2) & ((1u << deb) - 1u));
    const bool is_length = c.sym > 256u && c.sym <= 285u;
    const bool is_match = is_length && de != 0u && ds0 < 30u;
    c.kind = e == 0u ? 3u : (c.sym < 256u ? 0u : (c.sym == 256u ? 2u : (is_match ? 1u : 3u)));
    c.adv = c.kind == 1u ? l + eb + dl + deb : l;
    c.outlen = c.kind == 0u ? 1u : (c.kind == 1u ? c.n : 0u);
    return c;
}

// The code lengths of a dynamic-Huffman block header (br stands behind BFINAL and BTYPE) and the block's two tables.
// False for anything zlib would reject (too many codes, an over-subscribed or incomplete set, no end-of-block code).
__device__ __forceinline__ bool read_dynamic_tables(scginf::BitReader& br, WaveTables& T, uint8_t* lens) {
    using namespace scginf;
    const int nlen = static_cast<int>(br.bits(5)) + 257;
    const int ndist = static_cast<int>(br.bits(5)) + 1;
    const int ncode = static_cast<int>(br.bits(4)) + 4;
    if (nlen > 286 || ndist > 30) return false;
    uint8_t* const cl = lens + 320;
    // the order of the code-length code's lengths (16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15), five bits each
    const uint64_t order_lo = 0x22caa324e804a30ull, order_hi = 0x3c2e1346cull;
    for (int i = 0; i < 19; ++i) cl[i] = 0;
    for (int i = 0; i < ncode; ++i) {
        br.refill();
        const uint32_t at = static_cast<uint32_t>((i < 12 ? order_lo >> (5 * i) : order_hi >> (5 * (i - 12))) & 31u);
        cl[at] = static_cast<uint8_t>(br.bits(3));
    }
    if (!build_code(cl, 19, 0, T.dtab, 7, T.dcount, T.dsym, T.offs)) return false;
    int have = 0;
    while (have < nlen + ndist) {
        if (br.overrun()) return false;
        br.refill();
        const int s = decode_symbol(br, T.dtab, 7, T.dcount, T.dsym);
        if (s < 0) return false;
        if (s < 16) { lens[have++] = static_cast<uint8_t>(s); continue; }
        uint8_t fill = 0;
        int rep;
        if (s == 16) {
            if (have == 0) return false;
            fill = lens[have - 1];
            rep = 3 + static_cast<int>(br.bits(2));
        } else if (s == 17) {
            rep = 3 + static_cast<int>(br.bits(3));
        } else {
            rep = 11 + static_cast<int>(br.bits(7));
        }
        if (have + rep > nlen + ndist) return false;
        while (rep--) lens[have++] = fill;
    }
    if (lens[256] == 0) return false;
    if (!build_code(lens + nlen, ndist, 1, T.dtab, LANES_DIST_BITS, T.dcount, T.dsym, T.offs)) return false;
    if (!build_code(lens, nlen, 1, T.lit, LANES_LIT_BITS, T.lcount, T.lsym, T.offs)) return false;
    return true;
}

// Decodes DEFLATE blocks from bit `start_bit` of in[0 .. in_len) until a final block ends or a block ends at or beyond
// `stop_bit`, into out[0 .. out_len).  `end` receives the bit position behind the last block, `made` the elements
// written, `final_block` whether the last block was the stream's last.
struct LanesResult {
    uint32_t end_bit, made, final_block;
};
template<class OutT>
__device__ __forceinline__ int inflate_lanes(const uint8_t* __restrict__ in, uint32_t in_len, uint32_t start_bit, uint32_t stop_bit, OutT* out,
                                             uint32_t out_len, WaveTables& T, WaveStage& S, const uint32_t lane, LanesResult& res) {
    using namespace scginf;
    constexpr bool SYMBOLS = sizeof(OutT) == 2;
    uint8_t* const lens = reinterpret_cast<uint8_t*>(T.lit);             // 320 + 19 code lengths fit the 2 KiB of T.lit
    uint32_t bitpos = start_bit, op = 0;
    const uint32_t in_bits = in_len * 8u;
    uint32_t last;
    uint32_t nm = 0;                                                     // matches decoded, not copied yet
    InRing ring;
    ring.in = in; ring.limit = in_len + IN_SLACK; ring.lo = ring.hi = 0; ring.ahead = 0;
    do {
        if (bitpos > in_bits) return INFLATE_BAD_DATA;
        // ---- block header: the wavefront as a whole ----
        BitReader br;
        br.open(in, in_len);
        br.seek(bitpos >> 3);
        br.refill();
        br.bits(bitpos & 7u);
        br.refill();
        last = br.bits(1);
        const uint32_t type = br.bits(2);
        if (type == 0) {
            br.bits(br.cnt & 7u);
            br.refill();
            br.refill();
            const uint32_t n = br.bits(16), nn = br.bits(16);
            if ((n ^ 0xFFFFu) != nn) return INFLATE_BAD_DATA;
            if (n > out_len - op) return INFLATE_BAD_SIZE;
            const uint32_t from = br.pos - (br.cnt >> 3);                // byte position of the next unread byte
            if (from > in_len || n > in_len - from) return INFLATE_BAD_DATA;
            resolve_matches(out, S, nm, lane);
            nm = 0;
            wave_store(out, op, in + from, n, lane);
            op += n;
            bitpos = (from + n) * 8u;
            continue;
        }
        if (type == 3) return INFLATE_BAD_DATA;
        if (type == 1) {
            for (int s = 0; s < 144; ++s) lens[s] = 8;
            for (int s = 144; s < 256; ++s) lens[s] = 9;
            for (int s = 256; s < 280; ++s) lens[s] = 7;
            for (int s = 280; s < 288; ++s) lens[s] = 8;
            for (int s = 0; s < 32; ++s) lens[288 + s] = 5;
            if (!build_code(lens + 288, 32, 1, T.dtab, LANES_DIST_BITS, T.dcount, T.dsym, T.offs)) return INFLATE_BAD_DATA;
            if (!build_code(lens, 288, 1, T.lit, LANES_LIT_BITS, T.lcount, T.lsym, T.offs)) return INFLATE_BAD_DATA;
        } else if (!read_dynamic_tables(br, T, lens)) {
            return INFLATE_BAD_DATA;
        }
        bitpos = (br.pos * 8u) - br.cnt;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");          // the tables, written through one lane's eyes, are read per lane below

        // ---- the block's symbols, 64 bit offsets at a time ----
        ring.reset(S, bitpos >> 3, lane);
        bool block_done = false;
        while (!block_done) {
            if (bitpos > in_bits) return INFLATE_BAD_DATA;
            ring.ensure(S, bitpos >> 3, lane);
            // 128 bit offsets a batch, two per lane (a batch of 64 holds ~4.5 symbols of FASTQ: the chain walk, the LDS
            // look-ups' latency and the batch's fixed work are shared by twice as many this way)
            const LaneCode A = decode_here(T, ring.bits_at(S, bitpos + lane));
            const LaneCode B = decode_here(T, ring.bits_at(S, bitpos + INFLATE_BLOCK + lane));
            // The chain of code starts 0 -> next(0) -> ...: one v_readlane per symbol.  A lane's word holds the offset of
            // the code behind its own (< 128 + 48) and its output length; a lane whose successor cannot be walked over (end
            // of block, undecodable here) points to 256 + that offset instead, so the walk has ONE exit test per half --
            // "left these 64 offsets" -- and no second branch per symbol.
            const uint32_t nxtA = lane + A.adv, nxtB = INFLATE_BLOCK + lane + B.adv;
            const uint32_t behindA_a = static_cast<uint32_t>(__builtin_amdgcn_ds_bpermute(static_cast<int>((nxtA & 63u) << 2), static_cast<int>(A.kind)));
            const uint32_t behindA_b = static_cast<uint32_t>(__builtin_amdgcn_ds_bpermute(static_cast<int>((nxtA & 63u) << 2), static_cast<int>(B.kind)));
            const uint32_t behindB_b = static_cast<uint32_t>(__builtin_amdgcn_ds_bpermute(static_cast<int>((nxtB & 63u) << 2), static_cast<int>(B.kind)));
            const bool haltsA = (nxtA < INFLATE_BLOCK ? behindA_a : behindA_b) >= 2;          // (nxtA < 128 always)
            const bool haltsB = nxtB < 2 * INFLATE_BLOCK && behindB_b >= 2;
            const uint32_t infoA = (haltsA ? 256u + nxtA : nxtA) | (A.outlen << 16);
            const uint32_t infoB = (haltsB ? 256u + nxtB : nxtB) | (B.outlen << 16);
            uint32_t outposA = 0, outposB = 0;
            uint64_t chainA = 0, chainB = 0;                             // the offsets that are code starts
            uint32_t pos = rdlane(A.kind, 0) >= 2 ? 256u : 0u;           // (the batch may begin with such a code)
            while (pos < INFLATE_BLOCK) {
                const uint32_t inf = rdlane(infoA, pos);
                outposA = lane == pos ? op : outposA;                    // (v_writelane needs its lane number in m0 on gfx9: two plain vector instructions instead)
                chainA |= 1ull << pos;
                op += inf >> 16;
                pos = inf & 0xFFFFu;
            }
            while (pos < 2 * INFLATE_BLOCK) {
                const uint32_t at = pos - INFLATE_BLOCK;
                const uint32_t inf = rdlane(infoB, at);
                outposB = lane == at ? op : outposB;
                chainB |= 1ull << at;
                op += inf >> 16;
                pos = inf & 0xFFFFu;
            }
            uint32_t stop = 0;                                           // 0 none, 2 end of block, 3 undecodable here
            uint32_t stop_adv = 0;
            if (pos >= 256u) {
                pos -= 256u;
                stop = pos < INFLATE_BLOCK ? rdlane(A.kind, pos) : rdlane(B.kind, pos - INFLATE_BLOCK);
                stop_adv = pos < INFLATE_BLOCK ? rdlane(A.adv, pos) : rdlane(B.adv, pos - INFLATE_BLOCK);
            }
            if (op > out_len) return INFLATE_BAD_SIZE;                   // (nothing of the batch has been written yet)
            const bool mineA = (chainA >> lane) & 1ull, mineB = (chainB >> lane) & 1ull;
            // literals: one byte per lane and half, all at once
            if (mineA && A.kind == 0) out[outposA] = static_cast<OutT>(A.sym);
            if (mineB && B.kind == 0) out[outposB] = static_cast<OutT>(B.sym);
            // matches: noted, copied later -- the decoder does not need their bytes to go on, so it does not wait for them
            const bool matchA = mineA && A.kind == 1, matchB = mineB && B.kind == 1;
            const uint64_t matchesA = ballot64(matchA), matchesB = ballot64(matchB);
            if (matchesA | matchesB) {
                if (!SYMBOLS && ballot64((matchA && A.dist > outposA) || (matchB && B.dist > outposB))) return INFLATE_BAD_DATA;       // "invalid distance too far back"
                const uint32_t nA = static_cast<uint32_t>(__builtin_popcountll(matchesA));
                if (matchA) {
                    const uint32_t slot = nm + rank_below(matchesA);
                    S.tok_at[slot] = outposA;
                    S.tok_dist[slot] = A.dist | (A.n << 16);
                }
                if (matchB) {                                            // (behind the first half's: the list stays in text order)
                    const uint32_t slot = nm + nA + rank_below(matchesB);
                    S.tok_at[slot] = outposB;
                    S.tok_dist[slot] = B.dist | (B.n << 16);
                }
                nm += nA + static_cast<uint32_t>(__builtin_popcountll(matchesB));
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                if (nm > MATCH_SLOTS - INFLATE_BLOCK) {                  // (a batch holds at most 64 matches: two bits each at least)
                    resolve_matches(out, S, nm, lane);
                    nm = 0;
                }
            }
            bitpos += pos;                                               // (stop != 0: the symbol at `pos` was not consumed)
            if (stop == 0) continue;
            if (stop == 2) {
                bitpos += stop_adv;                                      // the end-of-block code itself
                block_done = true;
                continue;
            }
            // one symbol the old way: a code longer than the primary tables, or an error to be named
            BitReader sr;
            sr.open(in, in_len);
            sr.seek(bitpos >> 3);
            sr.refill();
            sr.bits(bitpos & 7u);
            sr.refill();
            int s = decode_symbol(sr, T.lit, LANES_LIT_BITS, T.lcount, T.lsym);
            if (s < 0) return INFLATE_BAD_DATA;
            if (s < 256) {
                if (op >= out_len) return INFLATE_BAD_SIZE;
                if (lane == 0) out[op] = static_cast<OutT>(s);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                ++op;
            } else if (s == 256) {
                block_done = true;
            } else {
                if (s > 285) return INFLATE_BAD_DATA;
                s -= 257;
                uint32_t len;
                if (s < 8) len = static_cast<uint32_t>(s) + 3u;
                else if (s == 28) len = 258;
                else {
                    const uint32_t eb = static_cast<uint32_t>(s - 4) >> 2;
                    len = ((4u + (static_cast<uint32_t>(s) & 3u)) << eb) + 3u + sr.bits(eb);
                }
                sr.refill();
                const int dcode = decode_symbol(sr, T.dtab, LANES_DIST_BITS, T.dcount, T.dsym);
                if (dcode < 0 || dcode >= 30) return INFLATE_BAD_DATA;
                uint32_t d;
                if (dcode < 4) d = static_cast<uint32_t>(dcode) + 1u;
                else {
                    const uint32_t eb = (static_cast<uint32_t>(dcode) >> 1) - 1u;
                    d = ((2u + (static_cast<uint32_t>(dcode) & 1u)) << eb) + 1u + sr.bits(eb);
                }
                if (!SYMBOLS && d > op) return INFLATE_BAD_DATA;
                if (len > out_len - op) return INFLATE_BAD_SIZE;
                resolve_matches(out, S, nm, lane);                       // (its source may be one of them)
                nm = 0;
                wave_copy(out, op, len, d, lane);
                op += len;
            }
            bitpos = (sr.pos * 8u) - sr.cnt;
        }
    } while (!last && bitpos < stop_bit);
    resolve_matches(out, S, nm, lane);
    res.end_bit = bitpos; res.made = op; res.final_block = last;
    return INFLATE_OK;
}

// A BGZF member: one DEFLATE stream that fills its payload and yields exactly the announced text.
__device__ __forceinline__ int inflate_member_lanes(const uint8_t* __restrict__ in, uint32_t in_len, uint8_t* out, uint32_t out_len,
                                                    WaveTables& T, WaveStage& S, const uint32_t lane) {
    LanesResult r;
    const int rc = inflate_lanes<uint8_t>(in, in_len, 0u, ~0u, out, out_len, T, S, lane, r);
    if (rc != scginf::INFLATE_OK) return rc;
    if (!r.final_block || ((r.end_bit + 7u) >> 3) != in_len || r.made != out_len) return scginf::INFLATE_BAD_SIZE;
    return scginf::INFLATE_OK;
}

#ifndef SCG_INFLATE_WAVES
#define SCG_INFLATE_WAVES 7          /* 72 VGPRs, no spills (8 = 64 VGPRs spills 9 dwords; A/B within noise) */
#endif
__global__ __launch_bounds__(INFLATE_BLOCK, SCG_INFLATE_WAVES) void inflate_members_lanes_kernel(const uint8_t* __restrict__ in, const scg::InflateMember* __restrict__ members,
                                                                              uint32_t n, uint8_t* text, uint32_t* __restrict__ status) {
    __shared__ WaveTables tables;
    __shared__ WaveStage stage;
    const uint32_t m = blockIdx.x;
    if (m >= n) return;
    const scg::InflateMember M = members[m];
    const int rc = inflate_member_lanes(in + M.in_off, M.in_len, text + M.out_off, M.out_len, tables, stage, threadIdx.x);
    if (rc != scginf::INFLATE_OK && threadIdx.x == 0) atomicOr(status, scg::INFLATE_STATUS_BAD);
}

// ---------------------------------------------------------------------------------------------------------------
// Ordinary gzip on the device (scg_pgzip.h explains the two-stage scheme; here its stage 1 runs one wavefront per chunk):
//   gunzip_find_kernel     every chunk but the first looks for the first bit position at or behind its nominal start at
//                          which a non-final dynamic-Huffman block header parses (64 positions tested per step by a
//                          cheap filter -- block type, code counts, a complete code-length code -- the survivors parsed
//                          in full, tables and all);
//   gunzip_decode_kernel   every chunk decodes from its start to the next chunk's start into 16-bit symbols, without the
//                          32 KiB of text in front of it (markers); the host then checks that every chunk ended exactly
//                          where the next began -- by induction from the stream's known first block every chunk then
//                          started on a true block boundary -- and that the last one ended the stream;
//   gunzip_tails_kernel    in stream order, one workgroup: the last 32 KiB of every chunk's symbols become text (their
//                          markers point into the 32 KiB before the chunk, which are text by then);
//   gunzip_resolve_kernel  everything else becomes text, all chunks at once.
// ---------------------------------------------------------------------------------------------------------------
// `in` holds the file's bytes [origin, size); chunks[k] is chunk number chunk0 + k of the stream (a group of a long file).
__global__ __launch_bounds__(INFLATE_BLOCK) void gunzip_find_kernel(const uint8_t* __restrict__ in_slice, uint64_t origin, uint64_t size,
                                                                    scg::GunzipChunk* __restrict__ chunks, uint32_t n, uint64_t chunk0, uint64_t first_byte,
                                                                    uint64_t chunk_bytes, uint64_t stream_end_byte) {
    __shared__ WaveTables tables;
    using namespace scginf;
    const uint8_t* in = in_slice - origin;       // (indexed by the file's own byte positions, all >= origin)
    const uint32_t c = blockIdx.x + 1;           // the group's first chunk starts where the one before ended
    if (c >= n) return;
    const uint32_t lane = threadIdx.x;
    uint8_t* const lens = reinterpret_cast<uint8_t*>(tables.lit);
    const uint64_t from = (first_byte + chunk_bytes * (chunk0 + c)) * 8u;
    uint64_t to = from + chunk_bytes * 8u;
    if (to > stream_end_byte * 8u) to = stream_end_byte * 8u;
    uint64_t found = ~0ull;
    for (uint64_t base = from; base + 80 <= to && found == ~0ull; base += INFLATE_BLOCK) {
        const uint64_t bit = base + lane;
        uint64_t v, v2;
        __builtin_memcpy(&v, in + (bit >> 3), 8);                       // (the buffer is readable IN_SLACK bytes beyond the stream)
        v >>= (bit & 7u);                                               // >= 57 bits
        const uint64_t bit2 = bit + 56;
        __builtin_memcpy(&v2, in + (bit2 >> 3), 8);
        v2 >>= (bit2 & 7u);
        // BFINAL = 0, BTYPE = 10 (LSB first), HLIT <= 29, HDIST <= 29, and the code-length code complete (Kraft sum)
        bool ok = (v & 7u) == 4u && ((v >> 3) & 31u) <= 29u && ((v >> 8) & 31u) <= 29u;
        const uint32_t ncode = static_cast<uint32_t>((v >> 13) & 15u) + 4u;
        uint32_t kraft = 0;
        uint64_t w = v >> 17;
#pragma unroll
        for (uint32_t i = 0; i < 13; ++i) { const uint32_t l = static_cast<uint32_t>(w & 7u); w >>= 3; if (i < ncode && l) kraft += 128u >> l; }
        w = v2;
#pragma unroll
        for (uint32_t i = 13; i < 19; ++i) { const uint32_t l = static_cast<uint32_t>(w & 7u); w >>= 3; if (i < ncode && l) kraft += 128u >> l; }
        ok = ok && kraft == 128u && bit + 80 <= to;
        uint64_t cand = ballot64(ok);
        while (cand) {
            const uint32_t k = static_cast<uint32_t>(__builtin_ctzll(cand));
            cand &= cand - 1;
            const uint64_t at = base + k;
            // the header in full, by the wavefront as a whole: relative to a base that keeps positions in 32 bits
            const uint64_t byte0 = at >> 3;
            const uint64_t left = size - byte0;
            BitReader br;
            br.open(in + byte0, static_cast<uint32_t>(left < (1u << 28) ? left : (1u << 28)));
            br.refill();
            br.bits(static_cast<uint32_t>(at & 7u) + 3u);               // BFINAL, BTYPE
            br.refill();
            if (read_dynamic_tables(br, tables, lens) && !br.overrun()) { found = at; break; }
        }
    }
    if (lane == 0) chunks[c].start_bit = found;
}

// (chunks[0 .. n) are searched; the first n_decode of them are decoded: the rest belong to the next group and only say where
// this group's last chunk has to stop)
__global__ __launch_bounds__(INFLATE_BLOCK, SCG_INFLATE_WAVES) void gunzip_decode_kernel(const uint8_t* __restrict__ in_slice, uint64_t origin, uint64_t size,
                                                                                        scg::GunzipChunk* __restrict__ chunks, uint32_t n, uint32_t n_decode,
                                                                                        uint16_t* __restrict__ syms, uint64_t cap_syms) {
    __shared__ WaveTables tables;
    __shared__ WaveStage stage;
    const uint8_t* in = in_slice - origin;
    const uint32_t c = blockIdx.x;
    if (c >= n_decode) return;
    const uint64_t start = chunks[c].start_bit;
    if (start == ~0ull) {                                               // no block starts here: the chunk before decodes through
        if (threadIdx.x == 0) { chunks[c].made = 0; chunks[c].status = 0; chunks[c].end_bit = ~0ull; chunks[c].final_block = 0; }
        return;
    }
    uint64_t stop = ~0ull;
    for (uint32_t k = c + 1; k < n; ++k) {
        const uint64_t s = chunks[k].start_bit;
        if (s != ~0ull) { stop = s; break; }
    }
    const uint64_t byte0 = start >> 3;
    const uint64_t left = size - byte0;
    const uint32_t in_len = static_cast<uint32_t>(left < (1u << 28) ? left : (1u << 28));
    const uint64_t rel_stop = stop == ~0ull ? ~0ull : stop - byte0 * 8u;
    LanesResult r;
    r.end_bit = 0; r.made = 0; r.final_block = 0;
    const int rc = inflate_lanes<uint16_t>(in + byte0, in_len, static_cast<uint32_t>(start & 7u), rel_stop < 0x7FFFFFFFull ? static_cast<uint32_t>(rel_stop) : 0x7FFFFFFFu,
                                           syms + cap_syms * c, static_cast<uint32_t>(cap_syms), tables, stage, threadIdx.x, r);
    if (threadIdx.x == 0) {
        chunks[c].status = static_cast<uint32_t>(rc);
        chunks[c].made = r.made;
        chunks[c].end_bit = byte0 * 8u + r.end_bit;
        chunks[c].final_block = r.final_block;
    }
}

constexpr int TAILS_BLOCK = 1024;
// The tails.  The last 32 KiB of every chunk's symbols become text first: a marker's byte lies in the 32 KiB in front of
// its chunk -- the tail of the chunk before -- so the tails form a chain through the whole stream, and in FASTQ nearly
// every tail is full of markers (a flank that every read repeats is copied from the read before, never spelled out
// again).  A chain of functions, though: a chunk turns the window in front of it into the window behind it, every byte
// of the new window either a literal or a copy of one byte of the old -- and such maps compose.  So, like a scan:
//   gunzip_tail_maps_kernel     one workgroup per GROUP of chunks composes its chunks' maps: the window behind the group
//                               as a function of the window in front of it (32 Ki 16-bit entries: literal or marker);
//   gunzip_tail_windows_kernel  one workgroup walks the groups -- not the chunks -- and leaves every group's window;
//   gunzip_tails_kernel         one workgroup per group walks its chunks with that window and writes the tails.
// The windows live in LDS throughout (two buffers: look-ups read one while the other is filled).
//
// avail: how many bytes of a window, from its end, belong to the member (a reference in front of the member is not a
// valid file, whatever text lies there).
__device__ __forceinline__ uint32_t group_made(const scg::GunzipChunk* __restrict__ chunks, uint32_t c0, uint32_t c1) {
    uint64_t sum = 0;
    for (uint32_t c = c0; c < c1; ++c) sum += chunks[c].made;
    return static_cast<uint32_t>(sum < MARKER_WINDOW ? sum : MARKER_WINDOW);
}

__global__ __launch_bounds__(TAILS_BLOCK) void gunzip_tail_maps_kernel(const uint16_t* __restrict__ syms, uint64_t cap_syms, const scg::GunzipChunk* __restrict__ chunks,
                                                                       uint32_t n, uint32_t group, uint16_t* __restrict__ maps) {
    __shared__ uint16_t map[2][MARKER_WINDOW];
    const uint32_t g = blockIdx.x, c0 = g * group, c1 = min(n, c0 + group);
    uint32_t cur = 0;
    for (uint32_t i = threadIdx.x; i < MARKER_WINDOW; i += TAILS_BLOCK) map[0][i] = static_cast<uint16_t>(MARKER + i);
    __syncthreads();
    for (uint32_t c = c0; c < c1; ++c) {
        const uint32_t made = chunks[c].made;
        if (made == 0) continue;                 // (no block began in this chunk: the one before decoded through it)
        const uint16_t* s = syms + cap_syms * c;
        // the window behind the chunk: the last 32 Ki of (the window in front ++ the chunk's symbols)
        for (uint32_t i = threadIdx.x; i < MARKER_WINDOW; i += TAILS_BLOCK) {
            const uint64_t j = static_cast<uint64_t>(i) + made;
            uint16_t v;
            if (j < MARKER_WINDOW) v = map[cur][j];
            else {
                v = s[j - MARKER_WINDOW];
                if (v >= MARKER) v = map[cur][v - MARKER];
            }
            map[cur ^ 1][i] = v;
        }
        __syncthreads();
        cur ^= 1;
    }
    uint16_t* out = maps + static_cast<size_t>(g) * MARKER_WINDOW;
    for (uint32_t i = threadIdx.x; i < MARKER_WINDOW; i += TAILS_BLOCK) out[i] = map[cur][i];
}

// wins[g], avails[g]: the window in front of group g's first chunk.  text_at[0] > floor: a later group of a long member,
// the end of the text before it lies in front.
__global__ __launch_bounds__(TAILS_BLOCK) void gunzip_tail_windows_kernel(const scg::GunzipChunk* __restrict__ chunks, const uint64_t* __restrict__ text_at, uint32_t n,
                                                                          uint32_t group, const uint16_t* __restrict__ maps, const uint8_t* __restrict__ text,
                                                                          uint64_t floor, uint8_t* __restrict__ wins, uint32_t* __restrict__ avails) {
    __shared__ uint8_t win[2][MARKER_WINDOW];
    uint32_t cur = 0, avail = 0;
    if (n && text_at[0] > floor) {
        const uint64_t at = text_at[0];
        avail = static_cast<uint32_t>(at - floor < MARKER_WINDOW ? at - floor : MARKER_WINDOW);
        for (uint32_t i = threadIdx.x; i < avail; i += TAILS_BLOCK) win[0][MARKER_WINDOW - avail + i] = text[at - avail + i];
    }
    __syncthreads();
    const uint32_t groups = (n + group - 1) / group;
    for (uint32_t g = 0; g < groups; ++g) {
        uint8_t* out = wins + static_cast<size_t>(g) * MARKER_WINDOW;
        for (uint32_t i = threadIdx.x; i < MARKER_WINDOW; i += TAILS_BLOCK) out[i] = win[cur][i];
        if (threadIdx.x == 0) avails[g] = avail;
        if (g + 1 == groups) break;
        const uint16_t* m = maps + static_cast<size_t>(g) * MARKER_WINDOW;
        for (uint32_t i = threadIdx.x; i < MARKER_WINDOW; i += TAILS_BLOCK) {
            const uint16_t v = m[i];
            win[cur ^ 1][i] = v >= MARKER ? win[cur][v - MARKER] : static_cast<uint8_t>(v);   // (bytes in front of the member: never looked at)
        }
        const uint32_t made = group_made(chunks, g * group, min(n, (g + 1) * group));
        avail = min(avail + made, static_cast<uint32_t>(MARKER_WINDOW));
        __syncthreads();
        cur ^= 1;
    }
}

// text_at[c] = where chunk c's text begins.
__global__ __launch_bounds__(TAILS_BLOCK) void gunzip_tails_kernel(const uint16_t* __restrict__ syms, uint64_t cap_syms, const scg::GunzipChunk* __restrict__ chunks,
                                                                   const uint64_t* __restrict__ text_at, uint32_t n, uint32_t group, const uint8_t* __restrict__ wins,
                                                                   const uint32_t* __restrict__ avails, uint8_t* text, uint32_t* __restrict__ status) {
    __shared__ uint8_t win[2][MARKER_WINDOW];
    const uint32_t g = blockIdx.x, c0 = g * group, c1 = min(n, c0 + group);
    uint32_t cur = 0;
    uint32_t avail = avails[g];
    {
        const uint8_t* w0 = wins + static_cast<size_t>(g) * MARKER_WINDOW;
        for (uint32_t i = threadIdx.x; i < MARKER_WINDOW; i += TAILS_BLOCK) win[0][i] = w0[i];
        __syncthreads();
    }
    bool bad = false;
    for (uint32_t c = c0; c < c1; ++c) {
        const uint32_t made = chunks[c].made;
        if (made == 0) continue;
        const uint64_t at = text_at[c];
        const uint32_t first_valid = MARKER_WINDOW - avail;
        if (made >= MARKER_WINDOW) {
            const uint16_t* s = syms + cap_syms * c + (made - MARKER_WINDOW);
            uint8_t* t = text + at + (made - MARKER_WINDOW);
            constexpr int PER = MARKER_WINDOW / TAILS_BLOCK;             // 32 symbols a thread
            uint16_t v[PER];
#pragma unroll
            for (int k = 0; k < PER; ++k) v[k] = s[threadIdx.x + k * TAILS_BLOCK];
#pragma unroll
            for (int k = 0; k < PER; ++k) {
                uint8_t byte = static_cast<uint8_t>(v[k]);
                if (v[k] >= MARKER) {
                    const uint32_t w = v[k] - MARKER;
                    if (w >= first_valid) byte = win[cur][w];
                    else { bad = true; byte = 0; }                       // (a reference in front of the stream)
                }
                win[cur ^ 1][threadIdx.x + k * TAILS_BLOCK] = byte;
                t[threadIdx.x + k * TAILS_BLOCK] = byte;
            }
            avail = MARKER_WINDOW;
        } else {
            // a short chunk (the stream's last, as a rule): all of it is tail, and the window keeps the end of the old one
            const uint16_t* s = syms + cap_syms * c;
            uint8_t* t = text + at;
            const uint32_t keep = MARKER_WINDOW - made;
            for (uint32_t i = threadIdx.x; i < keep; i += TAILS_BLOCK) win[cur ^ 1][i] = win[cur][i + made];
            for (uint32_t i = threadIdx.x; i < made; i += TAILS_BLOCK) {
                const uint32_t x = s[i];
                uint8_t byte = static_cast<uint8_t>(x);
                if (x >= MARKER) {
                    const uint32_t w = x - MARKER;
                    if (w >= first_valid) byte = win[cur][w];
                    else { bad = true; byte = 0; }
                }
                win[cur ^ 1][keep + i] = byte;
                t[i] = byte;
            }
            avail = min(avail + made, static_cast<uint32_t>(MARKER_WINDOW));
        }
        __syncthreads();
        cur ^= 1;
    }
    if (bad) atomicOr(status, 1u);
}

constexpr int RESOLVE_BLOCK = 256, RESOLVE_SLICES = 16;
// Everything in front of a chunk's tail becomes text, one slice of a chunk per workgroup.  A marker's byte lies in the
// 32 KiB in front of the chunk -- tails, final since gunzip_tails_kernel -- which every workgroup first copies to LDS;
// then eight symbols a thread and step: one 16-byte load, look-ups in LDS, one 8-byte store (cap_syms is a multiple of 8,
// so a chunk's symbols begin on a 16-byte boundary; the text's position is whatever the chunks before made it).
__global__ __launch_bounds__(RESOLVE_BLOCK) void gunzip_resolve_kernel(const uint16_t* __restrict__ syms, uint64_t cap_syms, const scg::GunzipChunk* __restrict__ chunks,
                                                                       const uint64_t* __restrict__ text_at, uint32_t n, uint8_t* text, uint64_t floor,
                                                                       uint32_t* __restrict__ status) {
    __shared__ __attribute__((aligned(16))) uint8_t win[MARKER_WINDOW];
    const uint32_t c = blockIdx.x / RESOLVE_SLICES, slice = blockIdx.x % RESOLVE_SLICES;
    if (c >= n) return;
    const uint32_t made = chunks[c].made;
    if (made <= MARKER_WINDOW) return;                                   // (all tail: text already)
    const uint32_t body = made - MARKER_WINDOW;
    const uint64_t at = text_at[c];
    // the window: text[at - 32 KiB + w] for the w that lie in the member (a reference in front of it is not a valid file)
    const uint32_t avail = static_cast<uint32_t>(at - floor < MARKER_WINDOW ? at - floor : MARKER_WINDOW);
    const uint32_t first_valid = MARKER_WINDOW - avail;
    {
        const uint8_t* src = text + at - MARKER_WINDOW;                  // (only src[first_valid ..] is touched)
        const uint32_t whole = (first_valid + 15u) & ~15u;               // 16-byte groups from here on
        for (uint32_t w = first_valid + threadIdx.x; w < whole && w < MARKER_WINDOW; w += RESOLVE_BLOCK) win[w] = src[w];
        for (uint32_t w = whole + threadIdx.x * 16u; w < MARKER_WINDOW; w += RESOLVE_BLOCK * 16u) {
            uint4 v;
            __builtin_memcpy(&v, src + w, 16);
            *reinterpret_cast<uint4*>(win + w) = v;
        }
    }
    __syncthreads();
    const uint16_t* s = syms + cap_syms * c;
    uint8_t* t = text + at;
    const uint32_t groups = (body + 7u) >> 3;
    const uint32_t ga = static_cast<uint32_t>(static_cast<uint64_t>(groups) * slice / RESOLVE_SLICES);
    const uint32_t gb = static_cast<uint32_t>(static_cast<uint64_t>(groups) * (slice + 1) / RESOLVE_SLICES);
    bool bad = false;
    auto byte_of = [&](uint32_t x) -> uint32_t {
        if (x < MARKER) return x & 0xFFu;
        const uint32_t w = x - MARKER;
        if (w >= first_valid) return win[w];
        bad = true;
        return 0u;
    };
    for (uint32_t g = ga + threadIdx.x; g < gb; g += RESOLVE_BLOCK) {
        const uint32_t i = g << 3;
        if (i + 8u <= body) {
            const uint4 v = *reinterpret_cast<const uint4*>(s + i);
            const uint32_t q[4] = {v.x, v.y, v.z, v.w};
            uint64_t out = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                out |= static_cast<uint64_t>(byte_of(q[k] & 0xFFFFu)) << (16 * k);
                out |= static_cast<uint64_t>(byte_of(q[k] >> 16)) << (16 * k + 8);
            }
            __builtin_memcpy(t + i, &out, 8);
        } else {
            for (uint32_t j = i; j < body; ++j) t[j] = static_cast<uint8_t>(byte_of(s[j]));   // (the tail behind is text already)
        }
    }
    if (bad) atomicOr(status, 1u);
}

// zlib's crc32.c: a * b mod p over GF(2), reflected (bit 31 is x^0).
__device__ __forceinline__ uint32_t multmodp(uint32_t a, uint32_t b) {
    uint32_t p = 0;
    for (int i = 0; i < 32 && a; ++i) {        // bit 31 of a first; done when no bits of a are left
        if (a & 0x80000000u) p ^= b;
        a <<= 1;
        b = (b & 1u) ? (b >> 1) ^ 0xEDB88320u : b >> 1;
    }
    return p;
}

constexpr int CRC_BLOCK = 256;

// One workgroup per member: every lane takes a contiguous piece of the member's text, computes its CRC-32 byte by
// byte through a 256-entry table in LDS, multiplies it by x^(8 * bytes behind the piece) and the products are XORed:
// crc(A || B) = crc(A) * x^(8 |B|) + crc(B)  (crc32_combine).
__global__ __launch_bounds__(CRC_BLOCK) void crc_members_kernel(const uint8_t* __restrict__ text, const scg::InflateMember* __restrict__ members, uint32_t n,
                                                                scg::CrcPowers P, uint32_t* __restrict__ status, uint32_t* __restrict__ crcs_out) {
    __shared__ uint32_t table[256];
    __shared__ uint32_t part[CRC_BLOCK / 64];
    {
        uint32_t c = threadIdx.x;
        for (int k = 0; k < 8; ++k) c = (c & 1u) ? (c >> 1) ^ 0xEDB88320u : c >> 1;
        table[threadIdx.x] = c;
    }
    __syncthreads();
    for (uint32_t m = blockIdx.x; m < n; m += gridDim.x) {
        const scg::InflateMember M = members[m];
        // pieces of whole 16-byte loads (a byte at a time, the 64 lanes of a load would touch 64 cache lines for 64 bytes)
        const uint32_t piece = ((M.out_len + CRC_BLOCK - 1) / CRC_BLOCK + 15u) & ~15u;
        const uint32_t a = min(threadIdx.x * piece, M.out_len), b = min(a + piece, M.out_len);
        uint32_t c = 0xFFFFFFFFu;
        const uint8_t* p = text + M.out_off;
        uint32_t i = a;
        for (; i + 16 <= b; i += 16) {
            uint4 v;
            __builtin_memcpy(&v, p + i, 16);
            const uint32_t d[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                c ^= d[k];
#pragma unroll
                for (int j = 0; j < 4; ++j) c = table[c & 0xFFu] ^ (c >> 8);
            }
        }
        for (; i < b; ++i) c = table[(c ^ p[i]) & 0xFFu] ^ (c >> 8);
        c = (b > a) ? ~c : 0u;                                 // (the CRC of nothing is 0)
        // x^(8 * (out_len - b)): product of the precomputed x^(2^k) over the set bits of the exponent
        uint32_t behind = M.out_len - b, x = 1u << 31;
        for (int k = 3; behind; ++k, behind >>= 1) if (behind & 1u) x = multmodp(P.x2n[k & 31], x);
        c = c ? multmodp(x, c) : 0u;
        for (int off = 32; off > 0; off >>= 1) c ^= __shfl_down(c, off, 64);
        if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = c;
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t crc = 0;
            for (int w = 0; w < CRC_BLOCK / 64; ++w) crc ^= part[w];
            if (crcs_out) crcs_out[m] = crc;                              // (pieces of one long text: the host combines them)
            else if (crc != M.crc) atomicOr(status, scg::INFLATE_STATUS_CRC);
        }
        __syncthreads();
    }
}

// text[0 .. gap) = one dummy record with an empty sequence, then the tail of the previous window's text (the bytes
// behind its last whole record).  prev_text == nullptr: no previous window.
__global__ __launch_bounds__(256) void carry_tail_kernel(const char* __restrict__ prev_text, const scg::TextScanResult* __restrict__ prev_result,
                                                         uint32_t prev_bytes, char* __restrict__ text, uint32_t gap, uint32_t* __restrict__ status) {
    uint32_t cut = prev_bytes;
    if (prev_text) {
        cut = prev_result->cut;
        if (cut > prev_bytes) cut = prev_bytes;
    }
    const uint32_t tail = prev_bytes - cut;
    if (tail + 6 > gap) {
        if (blockIdx.x == 0 && threadIdx.x == 0) atomicOr(status, scg::INFLATE_STATUS_TAIL);
        return;
    }
    const uint32_t dummy = gap - tail;       // "@" + (dummy - 6) x "x" + "\n\n+\n\n"
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < gap; i += gridDim.x * 256) {
        char c;
        if (i >= dummy) c = prev_text[cut + (i - dummy)];
        else if (i == 0) c = '@';
        else if (i < dummy - 5) c = 'x';
        else c = (i == dummy - 3) ? '+' : '\n';
        text[i] = c;
    }
}

} // namespace

namespace scg {

size_t inflate_input_slack() { return scginf::IN_SLACK; }

const CrcPowers& crc_powers() {
    static const CrcPowers P = [] {
        auto mult = [](uint32_t a, uint32_t b) {
            uint32_t m = 1u << 31, p = 0;
            for (;;) {
                if (a & m) {
                    p ^= b;
                    if ((a & (m - 1)) == 0) break;
                }
                m >>= 1;
                b = (b & 1u) ? (b >> 1) ^ 0xEDB88320u : b >> 1;
            }
            return p;
        };
        CrcPowers t;
        uint32_t p = 1u << 30;                 // x^1
        t.x2n[0] = p;
        for (int k = 1; k < 32; ++k) t.x2n[k] = p = mult(p, p);
        return t;
    }();
    return P;
}

hipError_t launch_inflate_members(const uint8_t* d_in, const InflateMember* d_members, uint32_t n, char* d_text, uint32_t* d_status, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    // SCG_INFLATE_LANES=0: the decoder that walks every symbol with the whole wavefront (measurement / test aid)
    const char* old_decoder = std::getenv("SCG_INFLATE_LANES");
    if (old_decoder && *old_decoder == '0') {
        hipLaunchKernelGGL(inflate_members_kernel, dim3(n), dim3(INFLATE_BLOCK), 0, stream, d_in, d_members, n,
                           reinterpret_cast<uint8_t*>(d_text), d_status);
    } else {
        hipLaunchKernelGGL(inflate_members_lanes_kernel, dim3(n), dim3(INFLATE_BLOCK), 0, stream, d_in, d_members, n,
                           reinterpret_cast<uint8_t*>(d_text), d_status);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(crc_members_kernel, dim3(n < 4096 ? n : 4096), dim3(CRC_BLOCK), 0, stream, reinterpret_cast<const uint8_t*>(d_text), d_members, n,
                       crc_powers(), d_status, static_cast<uint32_t*>(nullptr));
    return hipGetLastError();
}

hipError_t launch_gunzip_find(const uint8_t* d_in, uint64_t origin, uint64_t size, GunzipChunk* d_chunks, uint32_t n, uint64_t chunk0, uint64_t first_byte,
                              uint64_t chunk_bytes, uint64_t stream_end_byte, hipStream_t stream) {
    if (n <= 1) return hipSuccess;
    hipLaunchKernelGGL(gunzip_find_kernel, dim3(n - 1), dim3(INFLATE_BLOCK), 0, stream, d_in, origin, size, d_chunks, n, chunk0, first_byte, chunk_bytes,
                       stream_end_byte);
    return hipGetLastError();
}
hipError_t launch_gunzip_decode(const uint8_t* d_in, uint64_t origin, uint64_t size, GunzipChunk* d_chunks, uint32_t n, uint32_t n_decode, uint16_t* d_syms,
                                uint64_t cap_syms, hipStream_t stream) {
    if (n_decode == 0) return hipSuccess;
    hipLaunchKernelGGL(gunzip_decode_kernel, dim3(n_decode), dim3(INFLATE_BLOCK), 0, stream, d_in, origin, size, d_chunks, n, n_decode, d_syms, cap_syms);
    return hipGetLastError();
}
hipError_t launch_gunzip_text(const uint16_t* d_syms, uint64_t cap_syms, const GunzipChunk* d_chunks, const uint64_t* d_text_at, uint32_t n, char* d_text,
                              uint64_t floor, const GunzipTailScratch& scratch, uint32_t* d_status, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    const uint32_t group = scratch.group, groups = (n + group - 1) / group;
    uint8_t* text = reinterpret_cast<uint8_t*>(d_text);
    if (groups > 1)
        hipLaunchKernelGGL(gunzip_tail_maps_kernel, dim3(groups - 1), dim3(TAILS_BLOCK), 0, stream, d_syms, cap_syms, d_chunks, n, group, scratch.maps);
    hipLaunchKernelGGL(gunzip_tail_windows_kernel, dim3(1), dim3(TAILS_BLOCK), 0, stream, d_chunks, d_text_at, n, group, scratch.maps, text, floor, scratch.wins,
                       scratch.avails);
    hipLaunchKernelGGL(gunzip_tails_kernel, dim3(groups), dim3(TAILS_BLOCK), 0, stream, d_syms, cap_syms, d_chunks, d_text_at, n, group, scratch.wins, scratch.avails,
                       text, d_status);
    hipLaunchKernelGGL(gunzip_resolve_kernel, dim3(n * RESOLVE_SLICES), dim3(RESOLVE_BLOCK), 0, stream, d_syms, cap_syms, d_chunks, d_text_at, n, text, floor, d_status);
    return hipGetLastError();
}
// CRC-32 of the pieces members[0 .. n) of d_text (out_off, out_len; their crc fields are not looked at) -> d_crcs.
hipError_t launch_crc_pieces(const char* d_text, const InflateMember* d_members, uint32_t n, uint32_t* d_crcs, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(crc_members_kernel, dim3(n < 4096 ? n : 4096), dim3(CRC_BLOCK), 0, stream, reinterpret_cast<const uint8_t*>(d_text), d_members, n,
                       crc_powers(), static_cast<uint32_t*>(nullptr), d_crcs);
    return hipGetLastError();
}

hipError_t launch_carry_tail(const char* prev_text, const TextScanResult* prev_result, uint32_t prev_bytes, char* text, uint32_t gap, uint32_t* d_status,
                             hipStream_t stream) {
    if (gap < 6) return hipErrorInvalidValue;
    hipLaunchKernelGGL(carry_tail_kernel, dim3((gap + 255) / 256 < 1024 ? (gap + 255) / 256 : 1024), dim3(256), 0, stream, prev_text, prev_result,
                       prev_text ? prev_bytes : 0u, text, gap, d_status);
    return hipGetLastError();
}

} // namespace scg
