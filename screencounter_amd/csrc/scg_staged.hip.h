// scg_staged.hip.h -- the LDS-staged bit-plane scanner (the fast path of every counting kernel).
//
// Work decomposition for one workgroup of 256 lanes = 256 consecutive reads:
//
//  Phase A  (cooperative, coalesced)   the workgroup's contiguous byte span of ASCII reads is
//           loaded once, 16 B per lane per load, and transposed on the fly into three bit planes
//           in LDS: code bit 0, code bit 1 and validity (1 = ACGTacgt), one bit per base.
//           This is the only pass over the read bytes: HBM traffic = the algorithmic bytes.
//  Phase B  (one read per lane)        each lane pulls its read's plane bits out of LDS with
//           funnel shifts and runs a bit-parallel (Shift-And) exact search for k+1 pigeonhole
//           seeds of the template's constant bases on each strand; the union of the hits is a
//           superset of the positions where the reference's ScanTemplate reports <= k constant
//           mismatches (kaori/ScanTemplate.hpp:183-252).
//  Phase C  (one read per lane)        candidates are visited in the reference's order
//           (position-major, forward before reverse: kaori/SimpleSingleMatch.hpp:226-242); each is
//           verified exactly against the template planes (XOR/AND/popcount over the window taken
//           from LDS), its variable region is cut out of the planes and matched through the
//           segment index (scg_engine.hip.h).
//
// The host picks NW / NT from the batch's maximum read length and the template length, so every
// read fits its tile row; batches with reads longer than 320 bases (or of unknown maximum length)
// are counted by the byte-wise general kernels instead.  A staged kernel that nevertheless meets
// an oversize read raises the plan's error flag rather than producing a wrong count.
#ifndef SCG_STAGED_HIP_H
#define SCG_STAGED_HIP_H

#include <hip/hip_runtime.h>
#include <type_traits>
#include "scg_engine.hip.h"

namespace scgdev {

#ifndef SCG_STAGE_BLOCK
#define SCG_STAGE_BLOCK 256
#endif
constexpr int STAGE_BLOCK = SCG_STAGE_BLOCK;      // lanes = reads per workgroup

// Marks a wave-uniform value as such (keeps it in an SGPR).
__device__ __forceinline__ uint32_t uniform(uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); }
__device__ __forceinline__ int uniform(int x) { return __builtin_amdgcn_readfirstlane(x); }

template<int NW>
struct Tile {
    // 256 reads of up to 32*NW bases plus the <= 15 bytes the span starts before its first read
    static constexpr int CAP_BYTES = STAGE_BLOCK * 32 * NW + 16;
    static constexpr int WORDS = CAP_BYTES / 32 + NW + 5;            // plane words incl. read-ahead slack
    uint32_t p0[WORDS];
    uint32_t p1[WORDS];
    uint32_t v[WORDS];
};

// 16 ASCII bytes -> 16 bits of each plane.
//
// Per dword: code k = (byte >> 1) & 3; "ACTG"[k] is fetched with one byte permute (selected by the code bits in
// place) and XORed with the byte stripped of its case bit: the result z is zero exactly for ACGTacgt.  Plane bits are
// gathered with v_dot4_u32_u8 against power-of-two weights, straight from ASCII bits 1 and 2 (so
// scaled by 2 and 4).  When every lane of the wavefront holds only standard bases -- the rule in
// real data, where N calls are rare -- the validity plane is all ones and its gather (a second byte
// permute per dword plus a third dot product) is skipped.
__device__ __forceinline__ void transpose_chunk(const uint4& x, uint32_t& p0, uint32_t& p1, uint32_t& v) {
    const uint32_t d[4] = {x.x, x.y, x.z, x.w};
    uint32_t a0[2] = {0, 0}, a1[2] = {0, 0};
    uint32_t z[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t w = (i & 1) ? 0x80402010u : 0x08040201u;     // bit weights of the 4 bytes
        // "ACTG"[code] per byte without shifting the code down: ASCII bits 1-2 in place (0, 2, 4, 6) select bytes
        // 0 and 2 of each table word (a shift is a half-rate instruction on gfx950, the mask is not)
        const uint32_t expect = __builtin_amdgcn_perm(0x00470054u, 0x00430041u, d[i] & 0x06060606u);
        z[i] = expect ^ (d[i] & 0xDFDFDFDFu);                       // zero byte <=> standard base
        a0[i >> 1] = __builtin_amdgcn_udot4(d[i] & 0x02020202u, w, a0[i >> 1], false);
        a1[i >> 1] = __builtin_amdgcn_udot4(d[i] & 0x04040404u, w, a1[i >> 1], false);
    }
    p0 = (a0[0] >> 1) | (a0[1] << 7);
    p1 = (a1[0] >> 2) | (a1[1] << 6);
    if (__builtin_amdgcn_ballot_w64((z[0] | z[1] | z[2] | z[3]) != 0) == 0) {     // wave-uniform
        v = 0xFFFFu;
    } else {
        // z has no bit in positions 1, 2 (the table byte repeats the code bits) or 5, so z + 0x0C stays inside its byte:
        // 0x0C for a standard base, >= 0x0D otherwise -- the byte permute's selectors for the constants 0x00 and 0xFF.
        // A dot product of those bytes with the bit weights is 255 x (the invalid bits); starting from 0xFF its low
        // byte is their complement: the validity bits of 8 bases from three instructions per dword.
        uint32_t ai[2] = {0xFFu, 0xFFu};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t w = (i & 1) ? 0x80402010u : 0x08040201u;
            const uint32_t bad = __builtin_amdgcn_perm(0u, 0u, z[i] + 0x0C0C0C0Cu);
            ai[i >> 1] = __builtin_amdgcn_udot4(bad, w, ai[i >> 1], false);
        }
        v = __builtin_amdgcn_perm(ai[1], ai[0], 0x0C0C0400u);      // low bytes of the two halves side by side
    }
}

// Phase A.  Returns false when the workgroup's span does not fit the tile (caller falls back to
// the byte-wise engine for the whole workgroup).  On success `span0` is the byte offset (from
// R.seqs) that bit 0 of the planes corresponds to; it may be negative by up to 15.
template<int NW>
__device__ __forceinline__ bool stage_reads(const ScgReads& R, int64_t n_reads, int64_t r0, int nr,
                                            Tile<NW>& tile, int64_t& span0) {
    uint64_t b0, b1, total;
    if (R.offsets) {
        b0 = R.offsets[r0];
        b1 = R.offsets[r0 + nr];
        total = R.offsets[n_reads];
    } else {
        b0 = (uint64_t)r0 * (uint64_t)R.fixed_len;
        b1 = (uint64_t)(r0 + nr) * (uint64_t)R.fixed_len;
        total = (uint64_t)n_reads * (uint64_t)R.fixed_len;
    }
    const int delta = (int)((uintptr_t)(R.seqs + b0) & 15u);
    const uint64_t span = (b1 - b0) + (uint64_t)delta;
    if (span > (uint64_t)Tile<NW>::CAP_BYTES) return false;
    span0 = (int64_t)b0 - delta;
    const int nchunks = (int)((span + 15) >> 4);
    uint16_t* h0 = reinterpret_cast<uint16_t*>(tile.p0);
    uint16_t* h1 = reinterpret_cast<uint16_t*>(tile.p1);
    uint16_t* hv = reinterpret_cast<uint16_t*>(tile.v);
    // All of a lane's loads are issued before any is consumed, so that it has 2*NW independent
    // 16-byte requests in flight instead of one HBM round trip per chunk.  Only the first and the
    // last workgroup of a buffer can touch bytes outside [0, total); they take the guarded loop.
#ifdef SCG_STAGE_BATCH
    constexpr int BATCH = SCG_STAGE_BATCH;
#else
    constexpr int BATCH = NW;
#endif
    const bool edge = span0 < 0 || (uint64_t)(span0 + 16 * (int64_t)nchunks) > total;
    for (int c0 = threadIdx.x; c0 < nchunks; c0 += STAGE_BLOCK * BATCH) {
        uint4 x[BATCH];
        const uint8_t* base = R.seqs + span0 + 16 * (int64_t)c0;
        if (!edge) {
#pragma unroll
            for (int k = 0; k < BATCH; ++k) {
                if (c0 + k * STAGE_BLOCK < nchunks) {      // (x[k] stays unread otherwise)
                    // streamed once: non-temporal, so the read bytes do not evict the library index from L2
                    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                    u32x4 t = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(base + (size_t)k * STAGE_BLOCK * 16));
                    x[k] = make_uint4(t.x, t.y, t.z, t.w);
                }
            }
        } else {
#pragma unroll
            for (int k = 0; k < BATCH; ++k) {       // static k: x[] must stay in registers
                x[k] = make_uint4(0, 0, 0, 0);
                const int c = c0 + k * STAGE_BLOCK;
                if (c < nchunks) {
                    int64_t cb = span0 + 16 * (int64_t)c;
                    uint32_t t[4] = {0, 0, 0, 0};
                    for (int j = 0; j < 16; ++j) {
                        int64_t o = cb + j;
                        uint32_t byte = (o >= 0 && (uint64_t)o < total) ? R.seqs[o] : 0u;
                        t[j >> 2] |= byte << (8 * (j & 3));
                    }
                    x[k] = make_uint4(t[0], t[1], t[2], t[3]);
                }
            }
        }
#pragma unroll
        for (int k = 0; k < BATCH; ++k) {
            const int c = c0 + k * STAGE_BLOCK;
            if (c < nchunks) {
                uint32_t p0, p1, v;
                transpose_chunk(x[k], p0, p1, v);
                h0[c] = (uint16_t)p0;
                h1[c] = (uint16_t)p1;
                hv[c] = (uint16_t)v;
            }
        }
    }
    return true;
}

// NW words of a bit vector starting at bit `bit` of an LDS plane.
template<int NW>
__device__ __forceinline__ void load_bits(const uint32_t* __restrict__ plane, int bit, uint32_t out[NW]) {
    const int w0 = bit >> 5, sh = bit & 31;
    uint32_t lo = plane[w0];
#pragma unroll
    for (int i = 0; i < NW; ++i) {
        uint32_t hi = plane[w0 + i + 1];
        out[i] = __builtin_amdgcn_alignbit(hi, lo, sh);
        lo = hi;
    }
}

// x >>= g over NW little-endian words, g wave-uniform and < 32.
template<int NW>
__device__ __forceinline__ void shift_right_small(uint32_t x[NW], int g) {
#pragma unroll
    for (int i = 0; i < NW - 1; ++i) x[i] = __builtin_amdgcn_alignbit(x[i + 1], x[i], g);
    x[NW - 1] >>= g;
}

template<int NW>
__device__ __forceinline__ void shift_right(uint32_t x[NW], int g) {
    while (g > 0) {
        int step = g < 31 ? g : 31;
        shift_right_small<NW>(x, step);
        g -= step;
    }
}

// The four base-match planes of one read.
template<int NW>
struct BasePlanes {
    uint32_t e[4][NW];     // e[code][word]: bit j set iff base j is a standard base with that code
};

// Phase B for one strand: positions p (bit p of cand) where at least one seed matches exactly
// and p + T <= n.  The seed description is wave-uniform (kernel arguments, SGPRs).
//
// A seed with bases (t_j, c_j) matches at p iff  AND_j E[c_j][p + t_j].  The product is
// evaluated one base CODE at a time: for a fixed (compile-time) code c a running copy of plane
// E[c] is shifted along the seed's offsets and ANDed in wherever the seed asks for c (the walk is
// precomputed on the host as step bytes).  Every register index is static, each seed base costs
// one multi-word funnel shift and one multi-word 3-input bit op, branch-free.
//
// NC = words of candidate positions kept (positions 0 .. 32*NC-1).  With NC < NW (the host
// guarantees that every seed then starts below bit 32 and spans < 32 positions) only NC + 1
// words of the running plane are ever needed.
//
// The masks are NOT cut off at the read's last window position n - T: bits beyond it come from whatever follows
// the read in the tile and mean nothing.  Every consumer walks the candidates in ascending order and stops at the
// first position > n - T (StagedRead::last), which costs it nothing -- it replaces the "no bits left" test -- where
// building and applying per-lane length masks cost ~60 instructions a read.
template<int NW, int NC>
__device__ __forceinline__ void seed_candidates(const BasePlanes<NW>& E, const ScgSeeds& S, uint32_t cand[NC]) {
    constexpr int NS = (NC < NW) ? NC + 1 : NW;     // words of the running plane
    if (S.nseeds == 0) {
#pragma unroll
        for (int i = 0; i < NC; ++i) cand[i] = 0xFFFFFFFFu;
        return;
    }
#pragma unroll
    for (int i = 0; i < NC; ++i) cand[i] = 0;
    for (int s = 0; s < S.nseeds; ++s) {
        uint32_t g[NC];
#pragma unroll
        for (int i = 0; i < NC; ++i) g[i] = 0xFFFFFFFFu;
        const uint32_t counts = S.seed[s].nsteps;
        if constexpr (NC < NW) {
            // Compact form: a seed's bases all lie in one 32-position block (blk 0 or 1, guaranteed by the host), so the
            // plane shifted to a seed base is one funnel shift per word straight from two adjacent words of E (no running
            // copy), and two bases are folded into g with a single 3-input AND.  The host pads every walk to an even step count.
            static_assert(NC + 2 <= NW, "compact scanner: candidate words + two plane words of look-ahead");
            auto walk = [&](auto blk_tag) {
                constexpr int B = decltype(blk_tag)::value;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    int left = (int)((counts >> (8 * c)) & 0xFFu);
#pragma unroll
                    for (int wi = 0; wi < SCG_SEED_STEPS / 4; ++wi) {
                        if (left <= 0) break;
                        const uint32_t word = S.seed[s].walk[c].w[wi];
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            if (left <= 2 * h) break;
                            // (the funnel shift reads bits 4:0 of its shift operand: no masking of the offsets needed)
                            const uint32_t o1 = word >> (16 * h);
                            const uint32_t o2 = word >> (16 * h + 8);
#pragma unroll
                            for (int i = 0; i < NC; ++i) {
                                const uint32_t a = __builtin_amdgcn_alignbit(E.e[c][i + B + 1], E.e[c][i + B], o1);
                                const uint32_t b = __builtin_amdgcn_alignbit(E.e[c][i + B + 1], E.e[c][i + B], o2);
                                g[i] &= a & b;
                            }
                        }
                        left -= 4;
                    }
                }
            };
            if (uniform(S.seed[s].blk)) walk(std::integral_constant<int, 1>()); else walk(std::integral_constant<int, 0>());
#pragma unroll
            for (int i = 0; i < NC; ++i) cand[i] |= g[i];
            continue;
        }
        const int blk = uniform(S.seed[s].blk);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            int left = (int)((counts >> (8 * c)) & 0xFFu);
            if (left == 0) continue;
            // running copy of plane c, moved down by the seed's block (whole words), then by the walk's steps
            uint32_t cur[NS];
#pragma unroll
            for (int i = 0; i < NS; ++i) cur[i] = E.e[c][i];
            for (int b = 0; b < blk; ++b) {
#pragma unroll
                for (int i = 0; i + 1 < NS; ++i) cur[i] = cur[i + 1];
                cur[NS - 1] = 0;
            }
            int at = 0;                                   // offset the running copy stands at
#pragma unroll
            for (int wi = 0; wi < SCG_SEED_STEPS / 4; ++wi) {
                if (left <= 0) break;
                uint32_t word = S.seed[s].walk[c].w[wi];
                const int cnt = left < 4 ? left : 4;
                left -= 4;
                for (int k = 0; k < cnt; ++k) {
                    const int o = (int)(word & 31u);
                    word >>= 8;
                    shift_right_small<NS>(cur, o - at);      // offsets ascend (a padding byte repeats the last one: shift by 0)
                    at = o;
#pragma unroll
                    for (int i = 0; i < NC; ++i) g[i] &= cur[i];
                }
            }
        }
#pragma unroll
        for (int i = 0; i < NC; ++i) cand[i] |= g[i];
    }
}

// Lowest set bit of an NW-word mask, or 1 << 30 when empty.
template<int NW>
__device__ __forceinline__ int first_bit(const uint32_t m[NW]) {
    // v_ffbl_b32 returns 0xFFFFFFFF for an empty word, which survives the OR and loses every min
    uint32_t pos = 0xFFFFFFFFu;
#pragma unroll
    for (int i = 0; i < NW; ++i) {
        uint32_t r;
        asm("v_ffbl_b32 %0, %1" : "=v"(r) : "v"(m[i]));
        r |= 32u * i;
        pos = r < pos ? r : pos;
    }
    return pos >= (1u << 30) ? (1 << 30) : (int)pos;
}

template<int NW>
__device__ __forceinline__ void clear_bit(uint32_t m[NW], int pos) {
    const int w = pos >> 5;
    const uint32_t b = 1u << (pos & 31);
#pragma unroll
    for (int i = 0; i < NW; ++i) m[i] &= ~(w == i ? b : 0u);
}

// Exact number of constant-region mismatches of the template placed at plane bit `bit`
// (kaori/ScanTemplate.hpp:233-252: a non-standard base mismatches a constant position).
template<int NW, int NT>
__device__ __forceinline__ int window_mismatches(const Tile<NW>& tile, int bit, const ScgScan& T, bool reverse) {
    uint32_t w0[NT], w1[NT], wv[NT];
    load_bits<NT>(tile.p0, bit, w0);
    load_bits<NT>(tile.p1, bit, w1);
    load_bits<NT>(tile.v, bit, wv);
    int mm = 0;
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        // both strands' words are pinned in SGPRs first: selecting between two kernel-argument
        // fields by a per-lane condition otherwise becomes a per-lane LOAD from the argument segment
        const uint32_t f0 = uniform(T.fplane0[i]), r0 = uniform(T.rplane0[i]);
        const uint32_t f1 = uniform(T.fplane1[i]), r1 = uniform(T.rplane1[i]);
        const uint32_t fm = uniform(T.fmask[i]), rm = uniform(T.rmask[i]);
        uint32_t t0 = reverse ? r0 : f0;
        uint32_t t1 = reverse ? r1 : f1;
        uint32_t cm = reverse ? rm : fm;
        uint32_t same = wv[i] & ~((w0[i] ^ t0) | (w1[i] ^ t1));
        mm += __popc(cm & ~same);
    }
    return mm;
}

// Both strands' template words side by side in LDS, for kernels where the strand differs from lane
// to lane (single / combo): choosing between two kernel-argument words per lane costs three vector
// instructions a word (two SGPR->VGPR moves and a select), reading the chosen strand's row costs none.
template<int NT>
struct StrandTable {
    static constexpr int STRIDE = 3 * NT + 2 * SCG_MAX_REGIONS;   // plane0[NT] | plane1[NT] | mask[NT] | region starts | region lengths
    uint32_t w[2 * STRIDE];
};

template<int NT>
__device__ __forceinline__ void fill_strand_table(StrandTable<NT>& st, const ScgScan& T) {
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            st.w[i] = T.fplane0[i];
            st.w[NT + i] = T.fplane1[i];
            st.w[2 * NT + i] = T.fmask[i];
            st.w[StrandTable<NT>::STRIDE + i] = T.rplane0[i];
            st.w[StrandTable<NT>::STRIDE + NT + i] = T.rplane1[i];
            st.w[StrandTable<NT>::STRIDE + 2 * NT + i] = T.rmask[i];
        }
#pragma unroll
        for (int r = 0; r < SCG_MAX_REGIONS; ++r) {
            st.w[3 * NT + r] = (uint32_t)T.fstart[r];
            st.w[StrandTable<NT>::STRIDE + 3 * NT + r] = (uint32_t)T.rstart[r];
            st.w[3 * NT + SCG_MAX_REGIONS + r] = (uint32_t)T.flen[r];
            st.w[StrandTable<NT>::STRIDE + 3 * NT + SCG_MAX_REGIONS + r] = (uint32_t)T.rlen[r];
        }
    }
}

template<int NW, int NT>
__device__ __forceinline__ int window_mismatches(const Tile<NW>& tile, int bit, const StrandTable<NT>& st, bool reverse) {
    uint32_t w0[NT], w1[NT], wv[NT];
    load_bits<NT>(tile.p0, bit, w0);
    load_bits<NT>(tile.p1, bit, w1);
    load_bits<NT>(tile.v, bit, wv);
    const uint32_t* row = st.w + (reverse ? StrandTable<NT>::STRIDE : 0);
    int mm = 0;
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        uint32_t same = wv[i] & ~((w0[i] ^ row[i]) | (w1[i] ^ row[NT + i]));
        mm += __popc(row[2 * NT + i] & ~same);
    }
    return mm;
}

template<int NT>
__device__ __forceinline__ int region_start(const StrandTable<NT>& st, int r, bool reverse) {
    return (int)st.w[(reverse ? StrandTable<NT>::STRIDE : 0) + 3 * NT + r];
}

template<int NT>
__device__ __forceinline__ int region_length(const StrandTable<NT>& st, int r, bool reverse) {
    return (int)st.w[(reverse ? StrandTable<NT>::STRIDE : 0) + 3 * NT + SCG_MAX_REGIONS + r];
}

// `len` plane bits starting at bit `bit`, as a W-wide word (len <= bits of W).
template<class W> __device__ __forceinline__ W plane_bits(const uint32_t* __restrict__ plane, int bit);
template<> __device__ __forceinline__ uint32_t plane_bits<uint32_t>(const uint32_t* __restrict__ plane, int bit) {
    uint32_t a[1];
    load_bits<1>(plane, bit, a);
    return a[0];
}
template<> __device__ __forceinline__ uint64_t plane_bits<uint64_t>(const uint32_t* __restrict__ plane, int bit) {
    uint32_t a[2];
    load_bits<2>(plane, bit, a);
    return ((uint64_t)a[1] << 32) | a[0];
}

// Variable region of `len` bases (<= bits of W) starting at plane bit `bit`.
template<int NW, class W = uint32_t>
__device__ __forceinline__ QueryT<W> region_query(const Tile<NW>& tile, int bit, int len, bool reverse) {
    const W a = plane_bits<W>(tile.p0, bit), b = plane_bits<W>(tile.p1, bit), c = plane_bits<W>(tile.v, bit);
    const W m = low_mask_w<W>(len);
    QueryT<W> q;
    q.other = ~c & m;
    q.lo = a & m & ~q.other;
    q.hi = b & m & ~q.other;
    q.n_other = popcount_w(q.other);
    return reverse ? reverse_complement(q, len) : q;
}

// The template's variable regions at window position `bit`, concatenated in read order into one
// key (DualBarcodesSingleEnd.hpp:149-166 builds the same string; one region is the ordinary single
// barcode).  On the reverse strand the regions come in the reverse template's order and the whole
// key is reverse-complemented, which equals kaori searching its reverse-complemented library.
template<int NW, int NT, class W>
__device__ __forceinline__ QueryT<W> regions_query(const Tile<NW>& tile, int bit, const StrandTable<NT>& st, int nreg, int total_len, bool reverse) {
    QueryT<W> q;
    q.lo = 0; q.hi = 0; q.other = 0;
    int off = 0;
    for (int r = 0; r < nreg; ++r) {
        const int start = region_start<NT>(st, r, reverse), len = region_length<NT>(st, r, reverse);
        const W a = plane_bits<W>(tile.p0, bit + start), b = plane_bits<W>(tile.p1, bit + start), c = plane_bits<W>(tile.v, bit + start);
        const W m = low_mask_w<W>(len);
        const W other = ~c & m;
        q.other |= other << off;
        q.lo |= (a & m & ~other) << off;
        q.hi |= (b & m & ~other) << off;
        off += len;
    }
    q.n_other = popcount_w(q.other);
    return reverse ? reverse_complement(q, total_len) : q;
}

// Per-lane view of one staged read.
struct StagedRead {
    int bit;      // plane bit of base 0
    int n;        // length
};

// The last window position of a read of n bases for a template of tlen: candidates beyond it are not candidates
// (seed_candidates leaves them in its masks).  Negative when the read is shorter than the template.
__device__ __forceinline__ int last_position(int n, int tlen) { return n - tlen; }

// Phase B for both strands of one template.  The candidate masks may hold bits beyond the read's last window position
// (last_position): walk them in ascending order and stop there.
template<int NW, int NC = NW>
__device__ __forceinline__ void scan_read(const Tile<NW>& tile, const StagedRead& sr, const ScgScan& T,
                                          bool fwd, bool rev, uint32_t candF[NC], uint32_t candR[NC]) {
    BasePlanes<NW> E;
    {
        // (bits beyond the read's end belong to the next read: no seed looks at them from a position <= n - T)
        uint32_t p0[NW], p1[NW], v[NW];
        load_bits<NW>(tile.p0, sr.bit, p0);
        load_bits<NW>(tile.p1, sr.bit, p1);
        load_bits<NW>(tile.v, sr.bit, v);
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            E.e[0][i] = v[i] & ~p0[i] & ~p1[i];
            E.e[1][i] = v[i] & p0[i] & ~p1[i];
            E.e[2][i] = v[i] & ~p0[i] & p1[i];
            E.e[3][i] = v[i] & p0[i] & p1[i];
        }
    }
    if (fwd) {
        seed_candidates<NW, NC>(E, T.fseeds, candF);
    } else {
#pragma unroll
        for (int i = 0; i < NC; ++i) candF[i] = 0;
    }
    if (rev) {
        seed_candidates<NW, NC>(E, T.rseeds, candR);
    } else {
#pragma unroll
        for (int i = 0; i < NC; ++i) candR[i] = 0;
    }
}

} // namespace scgdev

#endif
