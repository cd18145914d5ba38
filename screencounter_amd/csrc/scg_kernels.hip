// scg_kernels.hip -- counting kernels (gfx950) and their launchers.
//
// Two engines compute the same function.  The *staged* kernels (scg_staged.hip.h: reads transposed
// into LDS bit planes, bit-parallel seed scan, exact verification) are what normally runs; the
// byte-wise *general* engine handles whatever does not fit the staged tiles (very long reads or
// templates) from inside the same kernels, and can be forced with SCG_FORCE_GENERAL=1 for tests.
// One read (or read pair) per lane.  Each kernel is the device counterpart of one
// kaori handler's process() (paths relative to inst/include/kaori/handlers/ in the reference):
//   single_kernel <- SingleBarcodeSingleEnd::process          SingleBarcodeSingleEnd.hpp:93-104
//                    + SimpleSingleMatch::search_first/best    ../SimpleSingleMatch.hpp:200-306
//   combo_kernel  <- CombinatorialBarcodesSingleEnd::process   CombinatorialBarcodesSingleEnd.hpp:149-258
//   dual_kernel   <- DualBarcodesPairedEnd::process            DualBarcodesPairedEnd.hpp:228-381
// Per-handler vector<int> counters + serial reduce() become device atomics on one int32 array.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdlib>
#include <mutex>

#include "scg_engine.hip.h"
#include "scg_staged.hip.h"
#include "scg_launch.h"

using namespace scgdev;

namespace {

constexpr int BLOCK = 256;

// Occupancy of the staged kernels (A/B-measured, tools/abx.sh): left to itself the compiler takes 78 VGPRs for the
// combination kernel = 6 waves per SIMD; asking for 8 (64 VGPRs, no spills) made it 11 % faster.  The dual kernel is
// limited by its two 15 KB tiles (5 workgroups per CU), not by registers.
#ifndef SCG_SINGLE_WAVES
#define SCG_SINGLE_WAVES 8
#endif
#ifndef SCG_COMBO_WAVES
#define SCG_COMBO_WAVES 8
#endif

// ---------------------------------------------------------------------------------------------
// single
// ---------------------------------------------------------------------------------------------
// Key of the template's variable regions at window position p, concatenated in read order (one
// region: the ordinary single barcode; several: DualBarcodesSingleEnd.hpp:149-166).
template<class W>
__device__ __forceinline__ QueryT<W> pack_regions(const ScgTemplate* T, const uint8_t* __restrict__ window, bool reverse, int total_len) {
    QueryT<W> q;
    q.lo = 0; q.hi = 0; q.other = 0;
    int off = 0;
    for (int r = 0; r < T->nreg; ++r) {
        const int start = reverse ? T->rstart[r] : T->fstart[r], len = reverse ? T->rlen[r] : T->flen[r];
        for (int j = 0; j < len; ++j) {
            int c = base_code(window[start + j]);
            if (c < 0) {
                q.other |= (W)1 << (off + j);
            } else {
                q.lo |= (W)(c & 1) << (off + j);
                q.hi |= (W)(c >> 1) << (off + j);
            }
        }
        off += len;
    }
    q.n_other = popcount_w(q.other);
    return reverse ? reverse_complement(q, total_len) : q;
}

template<class W = uint32_t>
__device__ __forceinline__ int single_read(const ScgSingleParams& P, const Read& rd) {
    const ScgTemplate* T = P.tmpl;
    const int len = T->len;
    const int max_mm = P.max_mm;
    int found = 0, index = -1, best = max_mm + 1;
    for (int p = 0; p + len <= rd.n; ++p) {
        for (int s = 0; s < 2; ++s) {                 // forward before reverse at each position
            if (s == 0 ? !P.fwd : !P.rev) continue;
            int c = const_mismatches(T, s != 0, rd.p, p, max_mm);
            if (c > max_mm) continue;
            QueryT<W> q = pack_regions<W>(T, rd.p + p, s != 0, P.index.len);
            int idx, d;
            index_match<W>(P.index, q, max_mm - c, idx, d);
            if (idx < 0) continue;
            int tot = c + d;
            if (P.use_first) {
                return idx;                           // SimpleSingleMatch.hpp:207-224 (tot <= max_mm by construction)
            } else if (tot == best) {                 // :265-289
                if (index != idx) { found = 0; index = -1; }
            } else if (tot < best) {
                found = 1; best = tot; index = idx;
            }
        }
    }
    return found ? index : -1;
}

template<class W>
__global__ __launch_bounds__(BLOCK) void single_kernel(ScgSingleParams P, ScgReads R, int64_t n_reads,
                                                        ScgCounters counts) {
    int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n_reads) return;
    Read rd = get_read(R, i);
    int idx = single_read<W>(P, rd);
    if (counts.unit_index) counts.unit_index[i] = idx;       // index-stream output (ScgCounters::unit_index)
    else if (idx >= 0) count_one(counts, idx);
}

// ---------------------------------------------------------------------------------------------
// random barcodes: where does the template sit?  (RandomBarcodeSingleEnd.hpp:117-175)
// hit = (position << 1) | reverse, or -1.  use_first: first position / strand within the budget;
// otherwise the unique minimum of constant mismatches (any tie, also across strands => none).
// ---------------------------------------------------------------------------------------------
struct BestHit {
    int best, code;
    bool tied;
    __device__ __forceinline__ void offer(int c, int p, bool reverse) {
        if (c < best) { best = c; code = (p << 1) | (reverse ? 1 : 0); tied = false; }
        else if (c == best) { tied = true; }
    }
};

__global__ __launch_bounds__(BLOCK) void random_kernel(ScgSingleParams P, ScgReads R, int64_t n_reads, int32_t* __restrict__ hits) {
    int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n_reads) return;
    Read rd = get_read(R, i);
    const ScgTemplate* T = P.tmpl;
    BestHit h{P.max_mm + 1, -1, false};
    int out = -1;
    for (int p = 0; p + T->len <= rd.n && out < 0; ++p) {
        for (int s = 0; s < 2; ++s) {
            if (s == 0 ? !P.fwd : !P.rev) continue;
            int c = const_mismatches(T, s != 0, rd.p, p, P.max_mm);
            if (c > P.max_mm) continue;
            if (P.use_first) { out = (p << 1) | s; break; }
            h.offer(c, p, s != 0);
        }
    }
    if (!P.use_first && !h.tied && h.best <= P.max_mm) out = h.code;
    hits[i] = out;
}

// ---------------------------------------------------------------------------------------------
// combo (two variable regions in one template)
// ---------------------------------------------------------------------------------------------
template<class W>
__device__ __forceinline__ bool combo_candidate(const ScgComboParams& P, const Read& rd, int p, bool reverse, int c,
                                                int out[SCG_COMBO_REGIONS], int& total) {
    const ScgTemplate* T = P.tmpl;
    int obs = c;
    for (int r = 0; r < SCG_COMBO_REGIONS; ++r) {
        // reverse scan order meets the pools back to front (CombinatorialBarcodesSingleEnd.hpp:111-116)
        int slot = reverse ? (SCG_COMBO_REGIONS - 1 - r) : r;
        int start = reverse ? T->rstart[r] : T->fstart[r];
        const ScgIndex& tab = P.index[slot];
        QueryT<W> q = pack_region<W>(rd.p + p + start, tab.len, reverse);
        int idx, d;
        index_match<W>(tab, q, P.max_mm - obs, idx, d, P.keep_first != 0);   // :168
        if (idx < 0) return false;
        obs += d;
        if (obs > P.max_mm) return false;               // :173-176
        out[slot] = idx;                                // :178-182
    }
    total = obs;
    return true;
}

// Returns 1 and fills best_id when the read yields a combination.
template<class W>
__device__ __forceinline__ int combo_read(const ScgComboParams& P, const Read& rd, int best_id[SCG_COMBO_REGIONS]) {
    const ScgTemplate* T = P.tmpl;
    const int len = T->len;
    int found = 0, best = P.max_mm + 1;
    for (int p = 0; p + len <= rd.n && !(found && P.use_first); ++p) {
        for (int s = 0; s < 2; ++s) {
            if (s == 0 ? !P.fwd : !P.rev) continue;
            int c = const_mismatches(T, s != 0, rd.p, p, P.max_mm);
            if (c > P.max_mm) continue;
            int cand[SCG_COMBO_REGIONS], tot;
            if (!combo_candidate<W>(P, rd, p, s != 0, c, cand, tot)) continue;
            if (P.use_first) {                          // :197-217
                found = 1; best_id[0] = cand[0]; best_id[1] = cand[1];
                break;
            } else if (tot <= best) {                   // :225-241
                if (tot == best) {
                    if (best_id[0] != cand[0] || best_id[1] != cand[1]) found = 0;
                } else {
                    found = 1; best = tot; best_id[0] = cand[0]; best_id[1] = cand[1];
                }
            }
        }
    }
    return found;
}

template<class W>
__global__ __launch_bounds__(BLOCK) void combo_kernel(ScgComboParams P, ScgReads R, int64_t n_reads,
                                                       ScgCounters cells) {
    int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n_reads) return;
    if (P.only_if_negative && P.only_if_negative[i] >= 0) return;      // second pass of the single-end dual diagnostics
    Read rd = get_read(R, i);
    int best_id[SCG_COMBO_REGIONS] = {0, 0};
    const bool found = combo_read<W>(P, rd, best_id) != 0;
    if (cells.unit_pair) cells.unit_pair[i] = found ? (((uint64_t)(uint32_t)best_id[0] << 32) | (uint32_t)best_id[1]) : ~0ull;
    else if (found) count_one(cells, (int64_t)best_id[0] * P.n_pool[1] + best_id[1]);
}

// ---------------------------------------------------------------------------------------------
// dual (paired-end)
// ---------------------------------------------------------------------------------------------
// One orientation: template 1 on read a, template 2 on read b.
// use_first: index of the first valid (hit1, hit2) combination or -1.
template<class W>
__device__ __forceinline__ int dual_first(const ScgDualParams& P, const Read& a, const Read& b) {
    const ScgTemplate* T1 = P.tmpl1;
    const ScgTemplate* T2 = P.tmpl2;
    int s1 = P.rev1 ? T1->rstart[0] : T1->fstart[0];
    int s2 = P.rev2 ? T2->rstart[0] : T2->fstart[0];
    for (int p1 = 0; p1 + T1->len <= a.n; ++p1) {
        int c1 = const_mismatches(T1, P.rev1 != 0, a.p, p1, P.max_mm1);
        if (c1 > P.max_mm1) continue;
        QueryT<W> q1 = pack_region<W>(a.p + p1 + s1, P.index1.len, P.rev1 != 0);
        for (int p2 = 0; p2 + T2->len <= b.n; ++p2) {
            int c2 = const_mismatches(T2, P.rev2 != 0, b.p, p2, P.max_mm2);
            if (c2 > P.max_mm2) continue;
            QueryT<W> q2 = pack_region<W>(b.p + p2 + s2, P.index2.len, P.rev2 != 0);
            int idx, tot;
            pair_match<W>(P.index1, P.index2, P.pairs, q1, P.max_mm1 - c1, q2, P.max_mm2 - c2, idx, tot);
            if (idx >= 0) return idx;                   // DualBarcodesPairedEnd.hpp:264-276
        }
    }
    return -1;
}

// best: (chosen, best_mismatches) as DualBarcodesPairedEnd.hpp:310-347
template<class W>
__device__ __forceinline__ void dual_best(const ScgDualParams& P, const Read& a, const Read& b, int& chosen, int& best) {
    const ScgTemplate* T1 = P.tmpl1;
    const ScgTemplate* T2 = P.tmpl2;
    int s1 = P.rev1 ? T1->rstart[0] : T1->fstart[0];
    int s2 = P.rev2 ? T2->rstart[0] : T2->fstart[0];
    chosen = -1;
    best = P.max_mm1 + P.max_mm2 + 1;
    for (int p1 = 0; p1 + T1->len <= a.n; ++p1) {
        int c1 = const_mismatches(T1, P.rev1 != 0, a.p, p1, P.max_mm1);
        if (c1 > P.max_mm1) continue;
        QueryT<W> q1 = pack_region<W>(a.p + p1 + s1, P.index1.len, P.rev1 != 0);
        for (int p2 = 0; p2 + T2->len <= b.n; ++p2) {
            int c2 = const_mismatches(T2, P.rev2 != 0, b.p, p2, P.max_mm2);
            if (c2 > P.max_mm2) continue;
            QueryT<W> q2 = pack_region<W>(b.p + p2 + s2, P.index2.len, P.rev2 != 0);
            int idx, tot;
            pair_match<W>(P.index1, P.index2, P.pairs, q1, P.max_mm1 - c1, q2, P.max_mm2 - c2, idx, tot);
            if (idx >= 0) {                             // :333-341
                int cur = tot + c1 + c2;
                if (cur < best) { chosen = idx; best = cur; }
                else if (cur == best && chosen != idx) { chosen = -1; }
            }
        }
    }
}

// SimpleSingleMatch::search_first / search_best of one mate on one strand (byte-wise), with the
// FIRST duplicate policy of the diagnostics path.  Returns found; index = sequence uid.
template<class W>
__device__ __forceinline__ bool mate_search(const ScgTemplate* T, const ScgIndex& X, bool reverse, int max_mm, bool use_first,
                                            bool keep_first, const Read& rd, int& index, int& mism) {
    bool found = false;
    index = -1; mism = 0;
    int best = max_mm + 1;
    const int start = reverse ? T->rstart[0] : T->fstart[0];
    for (int p = 0; p + T->len <= rd.n; ++p) {
        int c = const_mismatches(T, reverse, rd.p, p, max_mm);
        if (c > max_mm) continue;
        QueryT<W> q = pack_region<W>(rd.p + p + start, X.len, reverse);
        int idx, d;
        index_match<W>(X, q, max_mm - c, idx, d, keep_first);
        if (idx < 0) continue;
        int tot = c + d;
        if (use_first) { index = idx; mism = tot; return true; }
        if (tot == best) {
            if (index != idx) { found = false; index = -1; }
        } else if (tot < best) {
            found = true; best = tot; index = idx; mism = tot;
        }
    }
    return found;
}

// CombinatorialBarcodesPairedEnd::process (handlers/CombinatorialBarcodesPairedEnd.hpp:167-242) on a
// pair the dual search rejected.  S1(x)/S2(x) = search of template 1 / 2 on mate x (0 = a, 1 = b).
// `M` supplies the two searches as M::search1 / search2(P, mate, index, mismatches).  They take P explicitly:
// a closure holding a reference to the kernel-argument struct makes the compiler copy all of it (2 KB) to
// scratch at kernel entry, which cost the diagnostics kernels a factor of 8.
template<class M>
__device__ __forceinline__ void diagnose_pair(const ScgDualParams& P, const M& m, const ScgCounters& counters) {
    const int64_t b1_only = P.n_pool, b2_only = b1_only + 1, cells = b2_only + 1;
    // (sparse mode: the combination goes to the pair's slot of the stream, which the host has filled with ~0 beforehand)
    auto emit = [&](int u1, int u2) {
        if (counters.unit_pair) counters.unit_pair[(int64_t)blockIdx.x * blockDim.x + threadIdx.x] = ((uint64_t)(uint32_t)u1 << 32) | (uint32_t)u2;
        else count_one(counters, cells + (int64_t)u1 * P.n_uid2 + u2);
    };
    // Each of the four searches appears exactly once, in straight-line order: when the same search sits in
    // several divergent branches the compiler merges the copies and selects the template per lane, which
    // forces the whole argument struct into scratch.
    int i1 = -1, m1 = 0, i2 = -1, m2 = 0, j1 = -1, n1 = 0, j2 = -1, n2 = 0;
    const bool f1 = m.search1(P, 0, i1, m1);
    const bool f2 = m.search2(P, 1, i2, m2);
    // the swapped orientation is consulted by every randomized branch except "first match found both"
    const bool swapped = P.randomized && !(P.use_first && f1 && f2);
    bool g1 = false, g2 = false;
    if (swapped) {
        g1 = m.search1(P, 1, j1, n1);
        g2 = m.search2(P, 0, j2, n2);
    }
    // outcome per lane: 0 nothing, 1 combination (u1, u2), 2 barcode 1 only, 3 barcode 2 only
    int what = 0, u1 = 0, u2 = 0;
    auto pick = [&](int x, int y) { what = 1; u1 = x; u2 = y; };
    if (P.use_first) {                                   // :170-193
        if (f1 && f2) pick(i1, i2);
        else if (P.randomized) {
            if (g1 && g2) pick(j1, j2);
            else if (f1 || g1) what = 2;
            else if (f2 || g2) what = 3;
        } else {
            if (f1) what = 2;
            else if (f2) what = 3;
        }
    } else if (!P.randomized) {                          // :195-205
        if (f1 && f2) pick(i1, i2);
        else if (f1) what = 2;
        else if (f2) what = 3;
    } else if (f1 && f2) {                               // :207-226
        if (g1 && g2) {
            int mism = m1 + m2, rmism = n1 + n2;
            if (mism > rmism) pick(j1, j2);
            else if (mism < rmism) pick(i1, i2);
            else if (i1 == j1 && i2 == j2) pick(i1, i2);
        } else {
            pick(i1, i2);
        }
    } else {                                             // :227-239
        if (g1 && g2) pick(j1, j2);
        else if (f1 || g1) what = 2;
        else if (f2 || g2) what = 3;
    }
    if (what == 1) emit(u1, u2);
    count_flagged(counters, 0, b1_only, what == 2);      // single hot addresses: one atomic per wavefront, spread over slots
    count_flagged(counters, 1, b2_only, what == 3);
}

template<class W>
__device__ __forceinline__ int dual_pair(const ScgDualParams& P, const Read& a, const Read& b) {
    int idx;
    if (P.use_first) {                                  // :356-360
        idx = dual_first<W>(P, a, b);
        if (idx < 0 && P.randomized) idx = dual_first<W>(P, b, a);
    } else {                                            // :362-376
        int best;
        dual_best<W>(P, a, b, idx, best);
        if (P.randomized) {
            int idx2, best2;
            dual_best<W>(P, b, a, idx2, best2);
            if (idx < 0 || best > best2) { idx = idx2; best = best2; }
            else if (best == best2 && idx != idx2) { idx = -1; }
        }
    }
    return idx;
}

template<class W>
struct GeneralMates {
    Read a, b;
    __device__ __forceinline__ bool search1(const ScgDualParams& P, int which, int& index, int& mism) const {
        return mate_search<W>(P.tmpl1, P.index1, P.rev1 != 0, P.max_mm1, P.use_first != 0, P.keep_first != 0, which ? b : a, index, mism);
    }
    __device__ __forceinline__ bool search2(const ScgDualParams& P, int which, int& index, int& mism) const {
        return mate_search<W>(P.tmpl2, P.index2, P.rev2 != 0, P.max_mm2, P.use_first != 0, P.keep_first != 0, which ? b : a, index, mism);
    }
};

template<class W>
__global__ __launch_bounds__(BLOCK) void dual_kernel(ScgDualParams P, ScgReads R1, ScgReads R2, int64_t n_pairs,
                                                      ScgCounters counts) {
    int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n_pairs) return;
    if (P.diagnostics == 2 && P.only_if_negative && P.only_if_negative[i] >= 0) return;   // second pass of include.invalid=TRUE
    Read a = get_read(R1, i), b = get_read(R2, i);
    int idx = P.diagnostics == 2 ? -1 : dual_pair<W>(P, a, b);
    if (idx >= 0) {
        count_one(counts, idx);
    } else if (P.diagnostics) {
        diagnose_pair(P, GeneralMates<W>{a, b}, counts);
    }
}

// =============================================================================================
// Staged kernels
// =============================================================================================
// W = uint32_t: one variable region of <= 32 bases (the hot configuration).  W = uint64_t: keys of up to
// 64 bases, one region or several concatenated (wide barcodes, countDualBarcodesSingleEnd).
template<int NW, int NT, int NC, class W>
__device__ __forceinline__ int single_read_staged(const ScgSingleParams& P, const Tile<NW>& tile, const StrandTable<NT>& st, const StagedRead& sr, int ablate = 0) {
    const ScgScan& T = P.scan;
    const int max_mm = P.max_mm;
    uint32_t candF[NC], candR[NC];
    scan_read<NW, NC>(tile, sr, T, P.fwd != 0, P.rev != 0, candF, candR);
#ifdef SCG_ABLATE
    if (ablate == 1) return (candF[0] ^ candR[NC - 1]) == 0x12345u ? 0 : -1;
#endif
    int found = 0, index = -1, best = max_mm + 1;
    const int last = last_position(sr.n, T.len);
    for (;;) {
        int pf = first_bit<NC>(candF), pr = first_bit<NC>(candR);
        bool rev = pr < pf;                           // forward first on ties
        int p = rev ? pr : pf;
        if (p > last) break;                          // no candidate left inside the read (an empty mask gives 1 << 30)
        clear_bit<NC>(candF, rev ? -1 : p);
        clear_bit<NC>(candR, rev ? p : -1);
        int c = window_mismatches<NW, NT>(tile, sr.bit + p, st, rev);
        if (c > max_mm) continue;
        QueryT<W> q;
        if constexpr (sizeof(W) == 4) {
            const int start = region_start<NT>(st, 0, rev);
            q = region_query<NW>(tile, sr.bit + p + start, P.index.len, rev);
        } else {
            q = regions_query<NW, NT, W>(tile, sr.bit + p, st, uniform(T.nreg), P.index.len, rev);
        }
        int idx, d;
        index_match<W>(P.index, q, max_mm - c, idx, d);
        if (idx < 0) continue;
        int tot = c + d;
        if (P.use_first) {
            return idx;
        } else if (tot == best) {
            if (index != idx) { found = 0; index = -1; }
        } else if (tot < best) {
            found = 1; best = tot; index = idx;
        }
    }
    return found ? index : -1;
}

// NC < NW ("compact"): candidate positions fit 32*NC bits and the plan's seeds allow it (ScgScan::compact_ok).
template<int NW, int NT, int NC, class W>
__global__ __launch_bounds__(STAGE_BLOCK, SCG_SINGLE_WAVES) __attribute__((amdgpu_num_sgpr(80))) void single_staged_kernel(ScgSingleParams P, ScgReads R, int64_t n_reads,
                                                                   ScgCounters counts, int32_t* __restrict__ error_flag) {
    __shared__ Tile<NW> tile;
    __shared__ StrandTable<NT> strands;
    fill_strand_table<NT>(strands, P.scan);
    const int64_t r0 = (int64_t)blockIdx.x * STAGE_BLOCK;
    const int nr = (int)((n_reads - r0) < STAGE_BLOCK ? (n_reads - r0) : STAGE_BLOCK);
    int64_t span0 = 0;
    const bool staged = (P.scan.len <= 32 * NT) && stage_reads<NW>(R, n_reads, r0, nr, tile, span0);
    __syncthreads();
    if ((int)threadIdx.x >= nr) return;
    Read rd = get_read(R, r0 + threadIdx.x);
    if (!staged || rd.n > 32 * NW) {
        // The host picks the tile shape from the batch's maximum read length; landing here means
        // that bound was wrong.  Flag it (reported by scg_plan_read) instead of guessing.
        *error_flag = 1;
        return;
    }
    StagedRead sr;
    sr.bit = (int)((int64_t)(rd.p - R.seqs) - span0);
    sr.n = rd.n;
    int idx;
#ifdef SCG_ABLATE      // measurement builds only (phase ablation, tools/ablate_build.sh); never in the product library
    if (R.ablate == 2) {
        idx = (tile.p0[threadIdx.x] == 0xdeadbeefu) ? 0 : -1;
    } else {
        idx = single_read_staged<NW, NT, NC, W>(P, tile, strands, sr, R.ablate);
    }
    if (R.ablate == 3 && idx != 0x7ffffff0 && !counts.unit_index) return;     // keep idx live, skip the atomic
#else
    idx = single_read_staged<NW, NT, NC, W>(P, tile, strands, sr);
#endif
    if (counts.unit_index) counts.unit_index[r0 + threadIdx.x] = idx;
    else if (idx >= 0) count_one(counts, idx);
}

template<int NW, int NT, int NC>
__global__ __launch_bounds__(STAGE_BLOCK) __attribute__((amdgpu_num_sgpr(80))) void random_staged_kernel(ScgSingleParams P, ScgReads R, int64_t n_reads,
                                                                   int32_t* __restrict__ hits, int32_t* __restrict__ error_flag) {
    __shared__ Tile<NW> tile;
    __shared__ StrandTable<NT> strands;
    fill_strand_table<NT>(strands, P.scan);
    const int64_t r0 = (int64_t)blockIdx.x * STAGE_BLOCK;
    const int nr = (int)((n_reads - r0) < STAGE_BLOCK ? (n_reads - r0) : STAGE_BLOCK);
    int64_t span0 = 0;
    const bool staged = (P.scan.len <= 32 * NT) && stage_reads<NW>(R, n_reads, r0, nr, tile, span0);
    __syncthreads();
    if ((int)threadIdx.x >= nr) return;
    Read rd = get_read(R, r0 + threadIdx.x);
    if (!staged || rd.n > 32 * NW) {
        *error_flag = 1;
        return;
    }
    StagedRead sr;
    sr.bit = (int)((int64_t)(rd.p - R.seqs) - span0);
    sr.n = rd.n;
    uint32_t candF[NC], candR[NC];
    scan_read<NW, NC>(tile, sr, P.scan, P.fwd != 0, P.rev != 0, candF, candR);
    BestHit h{P.max_mm + 1, -1, false};
    int out = -1;
    const int last = last_position(sr.n, P.scan.len);
    for (;;) {
        int pf = first_bit<NC>(candF), pr = first_bit<NC>(candR);
        bool rev = pr < pf;                           // forward first on ties
        int p = rev ? pr : pf;
        if (p > last) break;
        clear_bit<NC>(candF, rev ? -1 : p);
        clear_bit<NC>(candR, rev ? p : -1);
        int c = window_mismatches<NW, NT>(tile, sr.bit + p, strands, rev);
        if (c > P.max_mm) continue;
        if (P.use_first) { out = (p << 1) | (rev ? 1 : 0); break; }
        h.offer(c, p, rev);
    }
    if (!P.use_first && !h.tied && h.best <= P.max_mm) out = h.code;
    hits[r0 + threadIdx.x] = out;
}

// SECOND: the masked DuplicateAction::FIRST pass of the single-end dual diagnostics -- a compile-time variant, so that
// the ordinary combination kernel keeps its code (a run-time flag here cost it 9 %).
// `pools`: the two pools' index descriptors in LDS.  Which pool a region is looked up in depends on the strand, which
// differs from lane to lane; picking between the two descriptors in the kernel arguments made every field of the
// chosen one (masks, table base, sizes: ~20 per look-up) a per-lane LOAD from the argument segment -- two thirds of
// the kernel's vector-memory instructions, each a round trip in the middle of the look-up's dependent chain.  From LDS
// the same selection is an address offset.
template<int NW, int NT, bool SECOND, class W>
__device__ __forceinline__ bool combo_candidate_staged(const ScgComboParams& P, const ScgIndex* pools, const Tile<NW>& tile, const StrandTable<NT>& st, const StagedRead& sr,
                                                       int p, bool reverse, int c, int out[SCG_COMBO_REGIONS], int& total) {
    int obs = c;
#pragma unroll
    for (int r = 0; r < SCG_COMBO_REGIONS; ++r) {
        int slot = reverse ? (SCG_COMBO_REGIONS - 1 - r) : r;
        const int start = region_start<NT>(st, r, reverse);
        const ScgIndex& tab = pools[slot];
        QueryT<W> q = region_query<NW, W>(tile, sr.bit + p + start, tab.len, reverse);
        int idx, d;
        index_match<W>(tab, q, P.max_mm - obs, idx, d, SECOND);
        if (idx < 0) return false;
        obs += d;
        if (obs > P.max_mm) return false;
        out[slot] = idx;
    }
    total = obs;
    return true;
}

template<int NW, int NT, int NC, bool SECOND, class W>
__global__ __launch_bounds__(STAGE_BLOCK, SCG_COMBO_WAVES) __attribute__((amdgpu_num_sgpr(80))) void combo_staged_kernel(ScgComboParams P, ScgReads R, int64_t n_reads,
                                                                  ScgCounters cells, int32_t* __restrict__ error_flag) {
    __shared__ Tile<NW> tile;
    __shared__ StrandTable<NT> strands;
    __shared__ ScgIndex pools[SCG_COMBO_REGIONS];
    fill_strand_table<NT>(strands, P.scan);
    if (threadIdx.x < SCG_COMBO_REGIONS * (sizeof(ScgIndex) / 4)) {
        static_assert(sizeof(ScgIndex) % 4 == 0, "copied by dwords");
        const uint32_t* src = reinterpret_cast<const uint32_t*>(&P.index[0]);
        reinterpret_cast<uint32_t*>(&pools[0])[threadIdx.x] = src[threadIdx.x];
    }
    const int64_t r0 = (int64_t)blockIdx.x * STAGE_BLOCK;
    const int nr = (int)((n_reads - r0) < STAGE_BLOCK ? (n_reads - r0) : STAGE_BLOCK);
    int64_t span0 = 0;
    const bool staged = (P.scan.len <= 32 * NT) && stage_reads<NW>(R, n_reads, r0, nr, tile, span0);
    __syncthreads();
    if ((int)threadIdx.x >= nr) return;
    Read rd = get_read(R, r0 + threadIdx.x);
    int found = 0, best = P.max_mm + 1;
    int best_id[SCG_COMBO_REGIONS] = {0, 0};
    if (!staged || rd.n > 32 * NW) {
        *error_flag = 1;
        return;
    }
    if (!(SECOND && P.only_if_negative[r0 + threadIdx.x] >= 0)) {   // (second pass: only reads the first pass left without a combination)
        const ScgScan& T = P.scan;
        StagedRead sr;
        sr.bit = (int)((int64_t)(rd.p - R.seqs) - span0);
        sr.n = rd.n;
        uint32_t candF[NC], candR[NC];
        scan_read<NW, NC>(tile, sr, T, P.fwd != 0, P.rev != 0, candF, candR);
        const int last = last_position(sr.n, T.len);
        for (;;) {
            int pf = first_bit<NC>(candF), pr = first_bit<NC>(candR);
            bool rev = pr < pf;
            int p = rev ? pr : pf;
            if (p > last) break;
            clear_bit<NC>(candF, rev ? -1 : p);
            clear_bit<NC>(candR, rev ? p : -1);
            int c = window_mismatches<NW, NT>(tile, sr.bit + p, strands, rev);
            if (c > P.max_mm) continue;
            int cand[SCG_COMBO_REGIONS], tot;
            if (!combo_candidate_staged<NW, NT, SECOND, W>(P, pools, tile, strands, sr, p, rev, c, cand, tot)) continue;
            if (P.use_first) {
                found = 1; best_id[0] = cand[0]; best_id[1] = cand[1];
                break;
            } else if (tot <= best) {
                if (tot == best) {
                    if (best_id[0] != cand[0] || best_id[1] != cand[1]) found = 0;
                } else {
                    found = 1; best = tot; best_id[0] = cand[0]; best_id[1] = cand[1];
                }
            }
        }
    }
    if (cells.unit_index) cells.unit_index[r0 + threadIdx.x] = found ? best_id[0] * P.n_pool[1] + best_id[1] : -1;     // tally mode
    else if (cells.unit_pair) cells.unit_pair[r0 + threadIdx.x] = found ? (((uint64_t)(uint32_t)best_id[0] << 32) | (uint32_t)best_id[1]) : ~0ull;   // sparse mode
    else if (found) count_one(cells, (int64_t)best_id[0] * P.n_pool[1] + best_id[1]);
}

// Staged counterpart of mate_search.
template<int NW, int NT, int NC, class W>
__device__ __forceinline__ bool mate_search_staged(const Tile<NW>& tile, const StagedRead& sr, const ScgScan& T, const ScgIndex& X,
                                                   bool reverse, int max_mm, bool use_first, bool keep_first, int& index, int& mism) {
    bool found = false;
    index = -1; mism = 0;
    int best = max_mm + 1;
    uint32_t cand[NC], unused[NC];
    // T carries the searched strand in its forward fields (searched_strand_first, scg_api.cpp); `reverse` only says how
    // the extracted region is to be read
    scan_read<NW, NC>(tile, sr, T, true, false, cand, unused);
    const int start = T.fstart[0];
    const int last = last_position(sr.n, T.len);
    for (;;) {
        int p = first_bit<NC>(cand);
        if (p > last) break;
        clear_bit<NC>(cand, p);
        int c = window_mismatches<NW, NT>(tile, sr.bit + p, T, false);
        if (c > max_mm) continue;
        QueryT<W> q = region_query<NW, W>(tile, sr.bit + p + start, X.len, reverse);
        int idx, d;
        index_match<W>(X, q, max_mm - c, idx, d, keep_first);
        if (idx < 0) continue;
        int tot = c + d;
        if (use_first) { index = idx; mism = tot; return true; }
        if (tot == best) {
            if (index != idx) { found = false; index = -1; }
        } else if (tot < best) {
            found = true; best = tot; index = idx; mism = tot;
        }
    }
    return found;
}

// One orientation of a staged pair: template 1 on (ta, a), template 2 on (tb, b).
template<int NW, int NT, int NC, class W>
__device__ __forceinline__ void dual_orientation_staged(const ScgDualParams& P, const bool BEST,
                                                        const Tile<NW>& ta, const StagedRead& a,
                                                        const Tile<NW>& tb, const StagedRead& b,
                                                        int& chosen, int& best) {
    const ScgScan& T1 = P.scan1;
    const ScgScan& T2 = P.scan2;
    const bool rev1 = P.rev1 != 0, rev2 = P.rev2 != 0;
    // both scans carry the searched strand in their forward fields (searched_strand_first, scg_api.cpp)
    const int s1 = T1.fstart[0];
    const int s2 = T2.fstart[0];
    chosen = -1;
    best = P.max_mm1 + P.max_mm2 + 1;
    uint32_t c1[NC], c2[NC], unused[NC];
    scan_read<NW, NC>(ta, a, T1, true, false, c1, unused);
    scan_read<NW, NC>(tb, b, T2, true, false, c2, unused);
    const int last1 = last_position(a.n, T1.len), last2 = last_position(b.n, T2.len);
    for (;;) {
        int p1 = first_bit<NC>(c1);
        if (p1 > last1) break;
        clear_bit<NC>(c1, p1);
        int m1 = window_mismatches<NW, NT>(ta, a.bit + p1, T1, false);
        if (m1 > P.max_mm1) continue;
        QueryT<W> q1 = region_query<NW, W>(ta, a.bit + p1 + s1, P.index1.len, rev1);
        uint32_t w2[NC];
#pragma unroll
        for (int i = 0; i < NC; ++i) w2[i] = c2[i];
        for (;;) {
            int p2 = first_bit<NC>(w2);
            if (p2 > last2) break;
            clear_bit<NC>(w2, p2);
            int m2 = window_mismatches<NW, NT>(tb, b.bit + p2, T2, false);
            if (m2 > P.max_mm2) {
                clear_bit<NC>(c2, p2);      // never a hit of mate 2: drop it for later outer iterations
                continue;
            }
            QueryT<W> q2 = region_query<NW, W>(tb, b.bit + p2 + s2, P.index2.len, rev2);
            int idx, tot;
            pair_match<W>(P.index1, P.index2, P.pairs, q1, P.max_mm1 - m1, q2, P.max_mm2 - m2, idx, tot);
            if (idx >= 0) {
                if (!BEST) { chosen = idx; return; }
                int cur = tot + m1 + m2;
                if (cur < best) { chosen = idx; best = cur; }
                else if (cur == best && chosen != idx) { chosen = -1; }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Pairs in two passes over ONE tile.  Two tiles (both mates' planes resident) cost 31 KB of LDS per workgroup: 5
// workgroups = 20 waves per CU, and the search is latency-bound (profiles/r3_occupancy.txt: taking 16 KB more LDS from
// each workgroup slows every staged kernel by 40 %).  Here mate 1 is staged, searched and reduced to a few registers,
// then mate 2 takes the same tile: 15.5 KB, 8 workgroups = 32 waves per CU.
//
// What a mate leaves behind, per template searched on it: the FIRST window that passes the constant bases (its variable
// region and its mismatches) and the number of windows that pass.  With at most one per mate the nested loops of
// DualBarcodesPairedEnd.hpp:264-276 / :310-347 have one iteration; a pair with more goes to the overflow list
// (ScgDualParams::overflow) and is searched byte-wise by dual_overflow_kernel right after.
// ---------------------------------------------------------------------------------------------
template<class W>
struct MateHits {
    QueryT<W> q;
    int m;
    int count;
};

template<int NW, int NT, int NC, class W>
__device__ __forceinline__ void collect_hits(const Tile<NW>& tile, const StagedRead& sr, const ScgScan& T, int key_len, bool reverse,
                                             int max_mm, MateHits<W>& h) {
    uint32_t cand[NC], unused[NC];
    scan_read<NW, NC>(tile, sr, T, true, false, cand, unused);
    const int last = last_position(sr.n, T.len);
    h.count = 0; h.m = 0;
    h.q = QueryT<W>{};
    for (;;) {
        int p = first_bit<NC>(cand);
        if (p > last) break;
        clear_bit<NC>(cand, p);
        int m = window_mismatches<NW, NT>(tile, sr.bit + p, T, false);
        if (m > max_mm) continue;
        if (h.count == 0) {
            h.q = region_query<NW, W>(tile, sr.bit + p + T.fstart[0], key_len, reverse);
            h.m = m;
        }
        ++h.count;
    }
}

// One orientation from what the passes kept: template 1's hit `x`, template 2's hit `y`.
template<class W>
__device__ __forceinline__ void dual_orientation_hits(const ScgDualParams& P, const bool BEST, const MateHits<W>& x, const MateHits<W>& y,
                                                      int& chosen, int& best) {
    chosen = -1;
    best = P.max_mm1 + P.max_mm2 + 1;
    if (x.count == 0 || y.count == 0) return;
    int idx, tot;
    pair_match<W>(P.index1, P.index2, P.pairs, x.q, P.max_mm1 - x.m, y.q, P.max_mm2 - y.m, idx, tot);
    if (idx >= 0) { chosen = idx; best = tot + x.m + y.m; }
}

// The four mate searches of diagnose_pair, done while each mate was staged.
struct SearchedMates {
    int index[4], mism[4];       // [2 * mate + template]
    bool found[4];
    __device__ __forceinline__ bool search1(const ScgDualParams&, int which, int& i, int& m) const { i = index[2 * which]; m = mism[2 * which]; return found[2 * which]; }
    __device__ __forceinline__ bool search2(const ScgDualParams&, int which, int& i, int& m) const { i = index[2 * which + 1]; m = mism[2 * which + 1]; return found[2 * which + 1]; }
};

// What the passes leave in registers.
template<class W>
struct PairState {
    MateHits<W> h1a, h2a, h1b, h2b;          // template 1 / 2 on mate a / b
    SearchedMates sm;
    bool bad;
};

// One pass: mate `pass` of the workgroup's pairs staged and searched.  Template 1 belongs on mate a, template 2 on mate b;
// the other two searches are the swapped orientation of randomized designs.
template<int NW, int NT, int NC, bool MATES_ONLY, bool RANDOMIZED, class W>
__device__ __forceinline__ void dual_pass(const int pass, const ScgDualParams& P, const ScgReads& R1, const ScgReads& R2, int64_t n_pairs,
                                          int64_t r0, int nr, bool fits, bool wanted, Tile<NW>& tile, PairState<W>& S) {
    ScgReads R;                              // (selected field by field: wave-uniform scalars)
    R.seqs = pass ? R2.seqs : R1.seqs; R.offsets = pass ? R2.offsets : R1.offsets;
    R.fixed_len = pass ? R2.fixed_len : R1.fixed_len; R.max_len = pass ? R2.max_len : R1.max_len;
    int64_t span = 0;
    if (pass) __syncthreads();               // every lane is done with mate 1's planes
    const bool ok = fits && stage_reads<NW>(R, n_pairs, r0, nr, tile, span);
    __syncthreads();
    S.bad = S.bad || !ok;
    if (!ok || !wanted) return;
    Read rd = get_read(R, r0 + threadIdx.x);
    if (rd.n > 32 * NW) { S.bad = true; return; }
    StagedRead sr;
    sr.bit = (int)((int64_t)(rd.p - R.seqs) - span); sr.n = rd.n;
    bool do1 = pass == 0 || RANDOMIZED, do2 = pass == 1 || RANDOMIZED;
    if (MATES_ONLY) {
        SearchedMates& sm = S.sm;
        // (as in diagnose_pair: the swapped orientation is not consulted when the first match found both; mate a's
        // search of template 2 comes before that is known)
        if (RANDOMIZED && pass == 1 && P.use_first && sm.found[0]) {
            int i2, m2;
            const bool f2 = mate_search_staged<NW, NT, NC, W>(tile, sr, P.scan2, P.index2, P.rev2 != 0, P.max_mm2, P.use_first != 0, P.keep_first != 0, i2, m2);
            sm.found[3] = f2; sm.index[3] = i2; sm.mism[3] = m2;
            do2 = false;
            do1 = !f2;
        }
        int i1 = -1, m1 = 0, i2 = -1, m2 = 0;
        bool f1 = false, f2 = false;
        if (do1) f1 = mate_search_staged<NW, NT, NC, W>(tile, sr, P.scan1, P.index1, P.rev1 != 0, P.max_mm1, P.use_first != 0, P.keep_first != 0, i1, m1);
        if (do2) f2 = mate_search_staged<NW, NT, NC, W>(tile, sr, P.scan2, P.index2, P.rev2 != 0, P.max_mm2, P.use_first != 0, P.keep_first != 0, i2, m2);
        if (pass == 0) { sm.found[0] = f1; sm.index[0] = i1; sm.mism[0] = m1; sm.found[1] = f2; sm.index[1] = i2; sm.mism[1] = m2; }
        else {
            sm.found[2] = f1; sm.index[2] = i1; sm.mism[2] = m1;
            if (do2) { sm.found[3] = f2; sm.index[3] = i2; sm.mism[3] = m2; }
        }
    } else {
        MateHits<W> t1, t2;
        t1.count = t2.count = 0; t1.m = t2.m = 0;
        t1.q = t2.q = QueryT<W>{};
        if (do1) collect_hits<NW, NT, NC, W>(tile, sr, P.scan1, P.index1.len, P.rev1 != 0, P.max_mm1, t1);
        if (do2) collect_hits<NW, NT, NC, W>(tile, sr, P.scan2, P.index2.len, P.rev2 != 0, P.max_mm2, t2);
        if (pass == 0) { S.h1a = t1; S.h2a = t2; } else { S.h1b = t1; S.h2b = t2; }
    }
}

// RANDOMIZED (both orientations: four searches): the two passes are ONE loop body, not unrolled, so that each template's
// search exists once in the code.  Two copies of a search (one per mate) get merged by the compiler into one that selects
// the template per lane, which moves the whole argument struct (2 KB) to scratch -- see diagnose_pair.  Otherwise each
// template is searched on its own mate only and the passes are straight-line code (20 registers fewer).
template<int NW, int NT, int NC, bool MATES_ONLY, bool RANDOMIZED, class W>
__global__ __launch_bounds__(STAGE_BLOCK) void dual_passes_kernel(ScgDualParams P, ScgReads R1, ScgReads R2, int64_t n_pairs,
                                                                 ScgCounters counts, int32_t* __restrict__ error_flag) {
    __shared__ Tile<NW> tile;
    const int64_t r0 = (int64_t)blockIdx.x * STAGE_BLOCK;
    const int nr = (int)((n_pairs - r0) < STAGE_BLOCK ? (n_pairs - r0) : STAGE_BLOCK);
    const bool fits = P.scan1.len <= 32 * NT && P.scan2.len <= 32 * NT;
    const bool lane = (int)threadIdx.x < nr;
    // (second pass of include.invalid=TRUE: a valid pair was counted by the first)
    const bool wanted = lane && !(MATES_ONLY && P.only_if_negative && P.only_if_negative[r0 + threadIdx.x] >= 0);
    PairState<W> S;
    S.bad = !fits;
    S.h1a.count = S.h2a.count = S.h1b.count = S.h2b.count = 0;
    S.h1a.m = S.h2a.m = S.h1b.m = S.h2b.m = 0;
    S.h1a.q = S.h2a.q = S.h1b.q = S.h2b.q = QueryT<W>{};
#pragma unroll
    for (int k = 0; k < 4; ++k) { S.sm.index[k] = -1; S.sm.mism[k] = 0; S.sm.found[k] = false; }

    if (RANDOMIZED) {
#pragma unroll 1
        for (int pass = 0; pass < 2; ++pass) dual_pass<NW, NT, NC, MATES_ONLY, RANDOMIZED, W>(pass, P, R1, R2, n_pairs, r0, nr, fits, wanted, tile, S);
    } else {
        dual_pass<NW, NT, NC, MATES_ONLY, RANDOMIZED, W>(0, P, R1, R2, n_pairs, r0, nr, fits, wanted, tile, S);
        dual_pass<NW, NT, NC, MATES_ONLY, RANDOMIZED, W>(1, P, R1, R2, n_pairs, r0, nr, fits, wanted, tile, S);
    }
    if (!lane) return;
    if (S.bad) { *error_flag = 1; return; }
    if (!wanted) return;
    if (MATES_ONLY) {
        diagnose_pair(P, S.sm, counts);
        return;
    }
    int idx = -1;
    if (S.h1a.count > 1 || S.h2b.count > 1 || S.h1b.count > 1 || S.h2a.count > 1) {
        const int slot = atomicAdd(P.overflow, 1);
        P.overflow[1 + slot] = (int32_t)(r0 + threadIdx.x);          // (batches are below 2^31 pairs)
        if (counts.unit_index) counts.unit_index[r0 + threadIdx.x] = -1;
        return;
    }
    // One body for both orientations and both policies: DualBarcodesPairedEnd.hpp:353-381.
    const bool best_mode = !P.use_first;
    int best;
    dual_orientation_hits<W>(P, best_mode, S.h1a, S.h2b, idx, best);
    if (RANDOMIZED && (best_mode || idx < 0)) {                      // :356-360, :363-371
        int ci, cb;
        dual_orientation_hits<W>(P, best_mode, S.h1b, S.h2a, ci, cb);
        if (!best_mode) idx = ci;
        else if (idx < 0 || best > cb) { idx = ci; best = cb; }
        else if (best == cb && idx != ci) { idx = -1; }
    }
    if (counts.unit_index) counts.unit_index[r0 + threadIdx.x] = idx;       // index stream (ScgCounters::unit_index)
    else if (idx >= 0) count_one(counts, idx);
}

// The pairs dual_passes_kernel listed, searched byte-wise like the general kernel does.  A fixed, small grid: the list is
// short (or empty) for any template with enough constant bases to be found by.
template<class W>
__global__ __launch_bounds__(BLOCK) void dual_overflow_kernel(ScgDualParams P, ScgReads R1, ScgReads R2, ScgCounters counts) {
    const int n = P.overflow[0];
    for (int k = blockIdx.x * BLOCK + threadIdx.x; k < n; k += gridDim.x * BLOCK) {
        const int64_t i = P.overflow[1 + k];
        Read a = get_read(R1, i), b = get_read(R2, i);
        const int idx = dual_pair<W>(P, a, b);
        if (counts.unit_index) counts.unit_index[i] = idx;
        else if (idx >= 0) count_one(counts, idx);
    }
}

// Both mates' tiles resident: for templates that pass their constant bases at many windows of a read (short flanks), where
// dual_passes_kernel would hand most pairs to the byte-wise search.
template<int NW, int NT, int NC, class W>
__global__ __launch_bounds__(STAGE_BLOCK) void dual_staged_kernel(ScgDualParams P, ScgReads R1, ScgReads R2, int64_t n_pairs,
                                                                 ScgCounters counts, int32_t* __restrict__ error_flag) {
    __shared__ Tile<NW> tile1;
    __shared__ Tile<NW> tile2;
    const int64_t r0 = (int64_t)blockIdx.x * STAGE_BLOCK;
    const int nr = (int)((n_pairs - r0) < STAGE_BLOCK ? (n_pairs - r0) : STAGE_BLOCK);
    int64_t span1 = 0, span2 = 0;
    const bool fits = P.scan1.len <= 32 * NT && P.scan2.len <= 32 * NT;
    const bool ok1 = fits && stage_reads<NW>(R1, n_pairs, r0, nr, tile1, span1);
    const bool ok2 = ok1 && stage_reads<NW>(R2, n_pairs, r0, nr, tile2, span2);
    __syncthreads();
    if ((int)threadIdx.x >= nr) return;
    Read a = get_read(R1, r0 + threadIdx.x), b = get_read(R2, r0 + threadIdx.x);
    if (!ok2 || a.n > 32 * NW || b.n > 32 * NW) {
        *error_flag = 1;
        return;
    }
    StagedRead sa, sb;
    sa.bit = (int)((int64_t)(a.p - R1.seqs) - span1); sa.n = a.n;
    sb.bit = (int)((int64_t)(b.p - R2.seqs) - span2); sb.n = b.n;
    // One body for both orientations (template 1 on mate 1 / on mate 2 when randomized) and
    // both policies: DualBarcodesPairedEnd.hpp:353-381.
    const bool best_mode = !P.use_first;
    const int norient = P.randomized ? 2 : 1;
    int best = 0, idx = -1;
    for (int o = 0; o < norient; ++o) {
        const Tile<NW>* ta = o ? &tile2 : &tile1;
        const Tile<NW>* tb = o ? &tile1 : &tile2;
        const StagedRead ra = o ? sb : sa, rb = o ? sa : sb;
        int ci, cb;
        dual_orientation_staged<NW, NT, NC, W>(P, best_mode, *ta, ra, *tb, rb, ci, cb);
        if (!best_mode) {
            idx = ci;
            if (ci >= 0) break;                      // :356-360
        } else if (o == 0) {
            idx = ci; best = cb;
        } else {                                     // :363-371
            if (idx < 0 || best > cb) { idx = ci; best = cb; }
            else if (best == cb && idx != ci) { idx = -1; }
        }
    }
    if (counts.unit_index) counts.unit_index[r0 + threadIdx.x] = idx;       // index stream (ScgCounters::unit_index)
    else if (idx >= 0) count_one(counts, idx);
}

// ---------------------------------------------------------------------------------------------
// matchBarcodes: one packed query per lane.
// ---------------------------------------------------------------------------------------------
template<class W>
__global__ __launch_bounds__(BLOCK) void match_kernel(ScgIndex tab, const uint8_t* __restrict__ seqs, int32_t n, int cap, int reverse,
                                                       int32_t* __restrict__ index_out, int32_t* __restrict__ mm_out) {
    int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    QueryT<W> q = pack_region<W>(seqs + (size_t)i * tab.len, tab.len, reverse != 0);
    int idx, d;
    index_match<W>(tab, q, cap, idx, d);
    index_out[i] = idx >= 0 ? idx : -1;
    mm_out[i] = idx >= 0 ? d : -1;
}

// ---------------------------------------------------------------------------------------------
// synthetic reads
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t splitmix64(uint64_t& s) {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__device__ __forceinline__ float u01(uint64_t& s) { return (float)(splitmix64(s) >> 40) * (1.0f / 16777216.0f); }
__device__ __forceinline__ uint32_t below(uint64_t& s, uint32_t n) { return (uint32_t)(((splitmix64(s) >> 32) * (uint64_t)n) >> 32); }

__global__ __launch_bounds__(BLOCK) void synth_kernel(scg_synth_spec S, char* __restrict__ out, int64_t n_reads) {
    int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n_reads) return;
    const char B[4] = {'A', 'C', 'G', 'T'};
    uint64_t g = (uint64_t)(S.first_read + i);
    // one stream for the choices every mate of a pair must share, one for this buffer's bases
    uint64_t shared = S.seed ^ (g * 0xD1342543DE82EF95ull);
    uint64_t priv = (S.seed + 0x632BE59BD9B4E019ull * (uint64_t)(S.pair_column + 1)) ^ (g * 0xA0761D6478BD642Full);
    char* dst = out + (size_t)i * (size_t)S.read_len;
    const int L = S.read_len, T = S.template_len;

    bool junk = u01(shared) < S.p_junk || T > L;
    bool flip = u01(shared) < S.p_reverse;
    int k0 = 0, k1 = 0;
    if (S.d_pair_index) {
        bool invalid = u01(shared) < S.p_invalid_pair;
        uint32_t row = below(shared, (uint32_t)S.n_pairs);
        uint32_t row2 = below(shared, (uint32_t)S.n_pairs);
        // an "invalid" pair takes its two barcodes from two independent rows
        int r = (invalid && S.pair_column == 1) ? (int)row2 : (int)row;
        k0 = S.d_pair_index[2 * (size_t)r + S.pair_column];
    } else {
        k0 = (int)below(shared, (uint32_t)(S.n_pool[0] > 0 ? S.n_pool[0] : 1));
        k1 = (int)below(shared, (uint32_t)(S.n_pool[1] > 0 ? S.n_pool[1] : 1));
    }
    int offset = junk ? 0 : (int)below(shared, (uint32_t)(L - T + 1));

    for (int j = 0; j < L; ++j) {
        char c = B[below(priv, 4)];
        if (!junk && j >= offset && j < offset + T) {
            int t = j - offset;
            char tc = S.d_template[t];
            if (tc != '-') {
                c = tc;
            } else {
                for (int r = 0; r < S.n_regions; ++r) {
                    int rel = t - S.region_start[r];
                    if (rel >= 0 && rel < S.region_len[r]) {
                        int k = r == 0 ? k0 : k1;
                        c = S.d_pool[r][(size_t)k * S.region_len[r] + rel];
                    }
                }
            }
            if (u01(priv) < S.p_sub) {               // substitution to a different base
                uint32_t cur = c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : 3;
                c = B[(cur + 1 + below(priv, 3)) & 3];
            }
        }
        if (u01(priv) < S.p_n) c = 'N';
        dst[j] = c;
    }
    if (flip) {
        for (int a = 0, b = L - 1; a <= b; ++a, --b) {
            char x = dst[a], y = dst[b];
            auto comp = [](char ch) { return ch == 'A' ? 'T' : ch == 'C' ? 'G' : ch == 'G' ? 'C' : ch == 'T' ? 'A' : ch; };
            dst[a] = comp(y);
            dst[b] = comp(x);
        }
    }
}

// Measurement aid ($SCG_EXTRA_LDS_KB, tools/abx.sh): unused dynamic LDS per workgroup, to see what a kernel loses when fewer
// workgroups fit a CU.  No effect on results.
inline size_t extra_lds() {
    static const size_t kb = [] { const char* e = std::getenv("SCG_EXTRA_LDS_KB"); return e ? (size_t)std::atoi(e) : (size_t)0; }();
    return kb << 10;
}
inline unsigned grid_for(int64_t n) { return (unsigned)((n + BLOCK - 1) / BLOCK); }
inline unsigned staged_grid(int64_t n) { return (unsigned)((n + STAGE_BLOCK - 1) / STAGE_BLOCK); }

} // namespace

namespace scg {

// Test hook: SCG_FORCE_GENERAL=1 runs the byte-wise engine alone (read at every launch so that a
// test process can flip it).  One definition for kernels and host pipelines alike (scg_launch.h).
bool force_general() {
    const char* e = std::getenv("SCG_FORCE_GENERAL");
    return e && *e && *e != '0';
}

namespace {

// NW in {5, 10}: plane words per read; NT in {2, 4, 8}: plane words per template window.
template<template<int, int> class Launch, class... Args>
hipError_t dispatch_shape(int max_len, int tmpl_len, Args&&... args) {
    const bool wide = max_len > 160;
    if (tmpl_len <= 64) return wide ? Launch<10, 2>::go(args...) : Launch<5, 2>::go(args...);
    if (tmpl_len <= 128) return wide ? Launch<10, 4>::go(args...) : Launch<5, 4>::go(args...);
    return wide ? Launch<10, 8>::go(args...) : Launch<5, 8>::go(args...);
}

template<int NW, int NT> struct LaunchSingle {
    static hipError_t go(const ScgSingleParams& P, const ScgReads& R, int64_t n, const ScgCounters& counts, int32_t* flag, hipStream_t stream) {
        // compact variant: all candidate positions (0 .. max_len - T) fit 3 words of a 5-word read
        const bool compact = NW == 5 && P.scan.compact_ok && R.max_len - P.scan.len + 1 <= 96;
        if (P.index.wide || P.scan.nreg != 1) {          // wide / concatenated keys
            if (compact) hipLaunchKernelGGL((single_staged_kernel<NW, NT, (NW == 5 ? 3 : NW), uint64_t>), dim3(staged_grid(n)), dim3(STAGE_BLOCK), extra_lds(), stream, P, R, n, counts, flag);
            else hipLaunchKernelGGL((single_staged_kernel<NW, NT, NW, uint64_t>), dim3(staged_grid(n)), dim3(STAGE_BLOCK), extra_lds(), stream, P, R, n, counts, flag);
        } else if (compact) {
            hipLaunchKernelGGL((single_staged_kernel<NW, NT, (NW == 5 ? 3 : NW), uint32_t>), dim3(staged_grid(n)), dim3(STAGE_BLOCK), extra_lds(), stream, P, R, n, counts, flag);
        } else {
            hipLaunchKernelGGL((single_staged_kernel<NW, NT, NW, uint32_t>), dim3(staged_grid(n)), dim3(STAGE_BLOCK), extra_lds(), stream, P, R, n, counts, flag);
        }
        return hipGetLastError();
    }
};
template<int NW, int NT> struct LaunchCombo {
    template<int NC, class W>
    static void run(const ScgComboParams& P, const ScgReads& R, int64_t n, const ScgCounters& cells, int32_t* flag, hipStream_t stream) {
        if (P.only_if_negative) hipLaunchKernelGGL((combo_staged_kernel<NW, NT, NC, true, W>), dim3(staged_grid(n)), dim3(STAGE_BLOCK), extra_lds(), stream, P, R, n, cells, flag);
        else hipLaunchKernelGGL((combo_staged_kernel<NW, NT, NC, false, W>), dim3(staged_grid(n)), dim3(STAGE_BLOCK), extra_lds(), stream, P, R, n, cells, flag);
    }
    static hipError_t go(const ScgComboParams& P, const ScgReads& R, int64_t n, const ScgCounters& cells, int32_t* flag, hipStream_t stream) {
        const bool compact = NW == 5 && P.scan.compact_ok && R.max_len - P.scan.len + 1 <= 96;
        constexpr int NCC = NW == 5 ? 3 : NW;
        const bool wide = P.index[0].wide != 0;          // pools of 33..64 bases (both indexes are then built wide)
        if (compact) { if (wide) run<NCC, uint64_t>(P, R, n, cells, flag, stream); else run<NCC, uint32_t>(P, R, n, cells, flag, stream); }
        else { if (wide) run<NW, uint64_t>(P, R, n, cells, flag, stream); else run<NW, uint32_t>(P, R, n, cells, flag, stream); }
        return hipGetLastError();
    }
};
// Expected number of windows of a random read that pass a template's constant bases: with few, the pair search keeps
// one hit per mate (dual_passes_kernel); a template that is found all over the place keeps both mates' tiles resident.
static double expected_chance_hits(const ScgScan& T, int max_mm, int read_len) {
    int c = 0;
    for (int w = 0; w < SCG_MAX_TEMPLATE / 32; ++w) c += __builtin_popcount(T.fmask[w]);
    double ways = 0, choose = 1, pow3 = 1;
    for (int k = 0; k <= max_mm && k <= c; ++k) {
        ways += choose * pow3;
        choose = choose * (c - k) / (k + 1);
        pow3 *= 3;
    }
    return (double)(read_len > T.len ? read_len - T.len + 1 : 1) * ways / std::pow(4.0, c);
}

static int two_tiles() {           // $SCG_DUAL_TWO_TILES=1 / 0 forces one or the other
    static const int v = [] { const char* e = getenv("SCG_DUAL_TWO_TILES"); return e && *e ? atoi(e) : -1; }();
    return v;
}

template<int NW, int NT> struct LaunchDual {
    template<int NC, class W>
    static void run(const ScgDualParams& P, const ScgReads& R1, const ScgReads& R2, int64_t n, const ScgCounters& counts, int32_t* flag, hipStream_t stream, bool passes) {
        if (P.diagnostics) {                             // 2 (mates only): independent searches of the two mates
            if (P.randomized) hipLaunchKernelGGL((dual_passes_kernel<NW, NT, NC, true, true, W>), dim3(staged_grid(n)), dim3(STAGE_BLOCK), extra_lds(), stream, P, R1, R2, n, counts, flag);
            else hipLaunchKernelGGL((dual_passes_kernel<NW, NT, NC, true, false, W>), dim3(staged_grid(n)), dim3(STAGE_BLOCK), extra_lds(), stream, P, R1, R2, n, counts, flag);
        } else if (passes) {
            (void)hipMemsetAsync(P.overflow, 0, sizeof(int32_t), stream);
            if (P.randomized) hipLaunchKernelGGL((dual_passes_kernel<NW, NT, NC, false, true, W>), dim3(staged_grid(n)), dim3(STAGE_BLOCK), extra_lds(), stream, P, R1, R2, n, counts, flag);
            else hipLaunchKernelGGL((dual_passes_kernel<NW, NT, NC, false, false, W>), dim3(staged_grid(n)), dim3(STAGE_BLOCK), extra_lds(), stream, P, R1, R2, n, counts, flag);
            hipLaunchKernelGGL((dual_overflow_kernel<W>), dim3(256), dim3(BLOCK), 0, stream, P, R1, R2, counts);
        } else {
            hipLaunchKernelGGL((dual_staged_kernel<NW, NT, NC, W>), dim3(staged_grid(n)), dim3(STAGE_BLOCK), extra_lds(), stream, P, R1, R2, n, counts, flag);
        }
    }
    static hipError_t go(const ScgDualParams& P, const ScgReads& R1, const ScgReads& R2, int64_t n, const ScgCounters& counts, int32_t* flag, hipStream_t stream) {
        const int max_len = R1.max_len > R2.max_len ? R1.max_len : R2.max_len;
        const int min_t = P.scan1.len < P.scan2.len ? P.scan1.len : P.scan2.len;
        const bool compact = NW == 5 && P.scan1.compact_ok && P.scan2.compact_ok && max_len - min_t + 1 <= 96;
        constexpr int NCC = NW == 5 ? 3 : NW;
        if (P.diagnostics == 1) return hipErrorInvalidValue;   // the host runs include.invalid=TRUE as a plain pass followed by a masked pass of 2
        const bool wide = P.index1.wide != 0;            // barcodes of 33..64 bases on either mate (both indexes are then built wide)
        const int forced = two_tiles();
        const bool passes = P.overflow && (forced >= 0 ? forced == 0
                                           : expected_chance_hits(P.scan1, P.max_mm1, max_len) + expected_chance_hits(P.scan2, P.max_mm2, max_len) < 0.01);
        if (compact) { if (wide) run<NCC, uint64_t>(P, R1, R2, n, counts, flag, stream, passes); else run<NCC, uint32_t>(P, R1, R2, n, counts, flag, stream, passes); }
        else { if (wide) run<NW, uint64_t>(P, R1, R2, n, counts, flag, stream, passes); else run<NW, uint32_t>(P, R1, R2, n, counts, flag, stream, passes); }
        return hipGetLastError();
    }
};

} // namespace

// The staged kernels need every read to fit a tile row (<= 320 bases); batches with longer reads
// or an unknown maximum length take the byte-wise general kernels.
static bool use_general(int max_len) { return force_general() || max_len <= 0 || max_len > 320; }

hipError_t launch_single(const ScgSingleParams& P, int tmpl_len, const ScgReads& R, int64_t n, const ScgCounters& counts, int32_t* flag, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    if (use_general(R.max_len) || P.index.wide == 2) {
        if (P.index.wide == 2) hipLaunchKernelGGL(single_kernel<Big>, dim3(grid_for(n)), dim3(BLOCK), 0, stream, P, R, n, counts);     // keys of 65..256 bases
        else if (P.index.wide || P.scan.nreg != 1) hipLaunchKernelGGL(single_kernel<uint64_t>, dim3(grid_for(n)), dim3(BLOCK), 0, stream, P, R, n, counts);
        else hipLaunchKernelGGL(single_kernel<uint32_t>, dim3(grid_for(n)), dim3(BLOCK), 0, stream, P, R, n, counts);
        return hipGetLastError();
    }
    return dispatch_shape<LaunchSingle>(R.max_len, tmpl_len, P, R, n, counts, flag, stream);
}

template<int NW, int NT> struct LaunchRandom {
    static hipError_t go(const ScgSingleParams& P, const ScgReads& R, int64_t n, int32_t* hits, int32_t* flag, hipStream_t stream) {
        if (NW == 5 && P.scan.compact_ok && R.max_len - P.scan.len + 1 <= 96) {
            hipLaunchKernelGGL((random_staged_kernel<NW, NT, (NW == 5 ? 3 : NW)>), dim3(staged_grid(n)), dim3(STAGE_BLOCK), extra_lds(), stream, P, R, n, hits, flag);
        } else {
            hipLaunchKernelGGL((random_staged_kernel<NW, NT, NW>), dim3(staged_grid(n)), dim3(STAGE_BLOCK), extra_lds(), stream, P, R, n, hits, flag);
        }
        return hipGetLastError();
    }
};

hipError_t launch_random(const ScgSingleParams& P, int tmpl_len, const ScgReads& R, int64_t n, int32_t* d_hits, int32_t* flag, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    if (use_general(R.max_len)) {
        hipLaunchKernelGGL(random_kernel, dim3(grid_for(n)), dim3(BLOCK), 0, stream, P, R, n, d_hits);
        return hipGetLastError();
    }
    return dispatch_shape<LaunchRandom>(R.max_len, tmpl_len, P, R, n, d_hits, flag, stream);
}

hipError_t launch_combo(const ScgComboParams& P, int tmpl_len, const ScgReads& R, int64_t n, const ScgCounters& cells, int32_t* flag, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    if (use_general(R.max_len) || P.index[0].wide == 2) {
        if (P.index[0].wide == 2) hipLaunchKernelGGL(combo_kernel<Big>, dim3(grid_for(n)), dim3(BLOCK), 0, stream, P, R, n, cells);
        else if (P.index[0].wide) hipLaunchKernelGGL(combo_kernel<uint64_t>, dim3(grid_for(n)), dim3(BLOCK), 0, stream, P, R, n, cells);
        else hipLaunchKernelGGL(combo_kernel<uint32_t>, dim3(grid_for(n)), dim3(BLOCK), 0, stream, P, R, n, cells);
        return hipGetLastError();
    }
    return dispatch_shape<LaunchCombo>(R.max_len, tmpl_len, P, R, n, cells, flag, stream);
}

hipError_t launch_dual(const ScgDualParams& P, int tmpl_len, const ScgReads& R1, const ScgReads& R2, int64_t n, const ScgCounters& counts, int32_t* flag, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    const int lo_len = R1.max_len < R2.max_len ? R1.max_len : R2.max_len;
    if (use_general(lo_len) || use_general(R1.max_len > R2.max_len ? R1.max_len : R2.max_len) || P.index1.wide == 2) {
        if (P.index1.wide == 2) hipLaunchKernelGGL(dual_kernel<Big>, dim3(grid_for(n)), dim3(BLOCK), 0, stream, P, R1, R2, n, counts);
        else if (P.index1.wide) hipLaunchKernelGGL(dual_kernel<uint64_t>, dim3(grid_for(n)), dim3(BLOCK), 0, stream, P, R1, R2, n, counts);
        else hipLaunchKernelGGL(dual_kernel<uint32_t>, dim3(grid_for(n)), dim3(BLOCK), 0, stream, P, R1, R2, n, counts);
        return hipGetLastError();
    }
    const int max_len = R1.max_len > R2.max_len ? R1.max_len : R2.max_len;
    return dispatch_shape<LaunchDual>(max_len, tmpl_len, P, R1, R2, n, counts, flag, stream);
}

// counters[i] += sum of counter i's replicas, replicas cleared (see ScgCounters).  Few replicas: one
// lane per counter; many (small, hot counter arrays): one wave per counter with a shuffle reduction.
// Exchanges, not load + store: kernels of another stream may be adding to the replicas right now.
// ---------------------------------------------------------------------------------------------
// Tally: barcode index per read -> counts, through LDS histograms.
// Memory-side atomics cost ~6.7 ps each on this part (90 M per 100 M-read launch = 0.6 ms, DESIGN.md
// section 4); a coalesced 4-byte store per read plus this kernel costs about a third of that.
// grid = (slices, passes): pass y owns counters [y * bins, (y + 1) * bins), slice x a contiguous range
// of reads.  Bins are 16-bit, two per LDS dword, so 80 K bins fit one CU's 160 KB; a bin is flushed to
// its global counter when it reaches 0x8000: the lane that sees the old value 0x7FFF subtracts 0x8000
// again, and since a workgroup has at most 1024 increments in flight the field cannot run over
// before that lands.  What is left is added to the global counters at the end (one atomic per
// non-empty bin and workgroup instead of one per read).
// ---------------------------------------------------------------------------------------------
constexpr int TALLY_BLOCK = 1024;

__global__ __launch_bounds__(TALLY_BLOCK) void tally_kernel(const int32_t* __restrict__ unit_index, int64_t n, int32_t* __restrict__ counters,
                                                            int64_t n_counters, int bins) {
    extern __shared__ uint32_t tally_bins[];       // bins / 2 dwords
    const int64_t lo = (int64_t)blockIdx.y * bins;
    const int nb = (int)((n_counters - lo) < bins ? (n_counters - lo) : bins);
    for (int i = threadIdx.x; i < (bins + 1) / 2; i += TALLY_BLOCK) tally_bins[i] = 0;
    __syncthreads();
    // slice boundaries in units of 4 indices so that 16-byte loads stay aligned
    const int64_t quads = (n + 3) / 4;
    const int64_t q0 = quads * blockIdx.x / gridDim.x, q1 = quads * (blockIdx.x + 1) / gridDim.x;
    auto add = [&](int32_t v) {
        const uint32_t b = (uint32_t)(v - (int32_t)lo);           // also rejects v = -1 and other passes' bins
        if (b >= (uint32_t)nb) return;
        const uint32_t sh = (b & 1u) * 16u;
        const uint32_t old = atomicAdd(&tally_bins[b >> 1], 1u << sh);
        if (((old >> sh) & 0xFFFFu) == 0x7FFFu) {
            atomicSub(&tally_bins[b >> 1], 0x8000u << sh);
            atomicAdd(&counters[lo + b], 0x8000);
        }
    };
    typedef int32_t i32x4 __attribute__((ext_vector_type(4)));
    const i32x4* quad = reinterpret_cast<const i32x4*>(unit_index);
    const int64_t q_full = n / 4;                  // quads below this are complete
    constexpr int U = 4;                           // loads in flight per lane: the stream is latency-bound otherwise
    int64_t q = q0 + threadIdx.x;
    for (; q + (U - 1) * TALLY_BLOCK < q1 && q + (U - 1) * TALLY_BLOCK < q_full; q += U * TALLY_BLOCK) {
        i32x4 v[U];
#pragma unroll
        for (int k = 0; k < U; ++k) v[k] = __builtin_nontemporal_load(quad + q + k * TALLY_BLOCK);
#pragma unroll
        for (int k = 0; k < U; ++k) { add(v[k].x); add(v[k].y); add(v[k].z); add(v[k].w); }
    }
    for (; q < q1; q += TALLY_BLOCK) {
        if (q < q_full) {
            const i32x4 v = __builtin_nontemporal_load(quad + q);
            add(v.x); add(v.y); add(v.z); add(v.w);
        } else {
            for (int64_t i = 4 * q; i < n; ++i) add(unit_index[i]);
        }
    }
    __syncthreads();
    for (int b = threadIdx.x; b < nb; b += TALLY_BLOCK) {
        const uint32_t c = (tally_bins[b >> 1] >> ((b & 1) * 16)) & 0xFFFFu;
        if (c) atomicAdd(&counters[lo + b], (int32_t)c);
    }
}

// Sums the per-wavefront slots of ScgCounters::hot into the two counters they stand for and clears them.
__global__ __launch_bounds__(BLOCK) void hot_fold_kernel(int32_t* __restrict__ hot, int32_t* __restrict__ pair_of_counters) {
    __shared__ int sums[2][BLOCK / 64];
    for (int which = 0; which < 2; ++which) {
        int acc = 0;
        for (int i = threadIdx.x; i < SCG_HOT_SLOTS; i += BLOCK) {
            // (exchanged, not loaded and stored: the diagnostics kernel of another stream may be adding to the slot right now;
            // and looked at through L2: see fold_kernel)
            int32_t* p = hot + which * SCG_HOT_SLOTS + i;
            if (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) acc += atomicExch(p, 0);
        }
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
        if ((threadIdx.x & 63) == 0) sums[which][threadIdx.x >> 6] = acc;
    }
    __syncthreads();
    if (threadIdx.x < 2) {
        int tot = 0;
        for (int w = 0; w < BLOCK / 64; ++w) tot += sums[threadIdx.x][w];
        if (tot) atomicAdd(&pair_of_counters[threadIdx.x], tot);
    }
}

// The "is there anything in this replica" look goes to L2 (an agent-scope atomic load), not through the CU's vector cache:
// the folds and counting kernels of a file's windows overlap on several streams, and a line of zeros that an earlier
// wavefront of another launch left in this CU's L1 hid the counts a kernel had added since -- they stayed in the replicas,
// and when it was the call's last fold they were missing from the result: four neighbouring counters (one 64-byte line of
// a 4-replica array) a few reads short, once in 30 000 calls with 100 KB windows (tools/stress_bgzf.py).
__global__ __launch_bounds__(BLOCK) void fold_kernel(int32_t* __restrict__ replicas, int shift, int64_t n,
                                                     int32_t* __restrict__ counters) {
    const int R = 1 << shift;
    if (R < 64) {
        int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
        if (i >= n) return;
        int32_t* p = replicas + (i << shift);
        int32_t sum = 0;
        for (int r = 0; r < R; ++r) {
            if (__hip_atomic_load(p + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) sum += atomicExch(p + r, 0);
        }
        if (sum) atomicAdd(&counters[i], sum);
    } else {
        int64_t i = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) >> 6;      // one wave per counter
        if (i >= n) return;
        const int lane = threadIdx.x & 63;
        int32_t* p = replicas + (i << shift);
        int32_t sum = 0;
        for (int r = lane; r < R; r += 64) {
            if (__hip_atomic_load(p + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) sum += atomicExch(p + r, 0);
        }
        for (int off = 32; off > 0; off >>= 1) sum += __shfl_down(sum, off, 64);
        if (lane == 0 && sum) atomicAdd(&counters[i], sum);
    }
}

hipError_t launch_hot_fold(int32_t* hot, int32_t* pair_of_counters, hipStream_t stream) {
    hipLaunchKernelGGL(hot_fold_kernel, dim3(1), dim3(BLOCK), 0, stream, hot, pair_of_counters);
    return hipGetLastError();
}

hipError_t launch_tally(const int32_t* unit_index, int64_t n, int32_t* counters, int64_t n_counters, hipStream_t stream) {
    if (n <= 0 || n_counters <= 0) return hipSuccess;
    const int max_bins = 80 * 1024;                       // 160 KB of LDS as 16-bit bins
    const int passes = (int)((n_counters + max_bins - 1) / max_bins);
    int bins = (int)((n_counters + passes - 1) / passes);
    bins = (bins + 1) & ~1;
    // Per device, once (several host threads launch on several devices at the same time: schedule_files, PlanSet): the
    // kernel's dynamic-LDS limit and the device's CU count.
    constexpr int MAX_DEVICES = 64;
    static std::once_flag once[MAX_DEVICES];
    static int cu_count[MAX_DEVICES];
    static hipError_t setup[MAX_DEVICES];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEVICES) return hipErrorInvalidDevice;
    std::call_once(once[dev], [dev] {
        setup[dev] = hipFuncSetAttribute(reinterpret_cast<const void*>(tally_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024 * 2);
        hipDeviceProp_t prop;
        cu_count[dev] = hipGetDeviceProperties(&prop, dev) == hipSuccess ? prop.multiProcessorCount : 256;
    });
    if (setup[dev] != hipSuccess) return setup[dev];
    // one workgroup (160 KB of LDS) per CU at a time: aim at one full wave of workgroups, each with
    // >= 256 K reads so that the final flush (one atomic per non-empty bin and workgroup) stays small
    const int cus = cu_count[dev];
    int64_t slices = n / (int64_t(1) << 18);
    if (slices < 1) slices = 1;
    if (slices * passes > cus) slices = cus / passes > 0 ? cus / passes : 1;
    hipLaunchKernelGGL(tally_kernel, dim3((unsigned)slices, (unsigned)passes), dim3(TALLY_BLOCK), (size_t)bins * 2, stream,
                       unit_index, n, counters, n_counters, bins);
    return hipGetLastError();
}

hipError_t launch_fold(int32_t* replicas, int shift, int64_t n, int32_t* counters, hipStream_t stream) {
    if (n <= 0 || shift <= 0) return hipSuccess;
    const int64_t threads = (1 << shift) < 64 ? n : n * 64;
    hipLaunchKernelGGL(fold_kernel, dim3(grid_for(threads)), dim3(BLOCK), 0, stream, replicas, shift, n, counters);
    return hipGetLastError();
}

hipError_t launch_match(const ScgIndex& tab, const uint8_t* d_seqs, int32_t n, int cap, int reverse,
                        int32_t* d_index, int32_t* d_mm, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    if (tab.wide == 2) hipLaunchKernelGGL(match_kernel<Big>, dim3(grid_for(n)), dim3(BLOCK), 0, stream, tab, d_seqs, n, cap, reverse, d_index, d_mm);
    else if (tab.wide) hipLaunchKernelGGL(match_kernel<uint64_t>, dim3(grid_for(n)), dim3(BLOCK), 0, stream, tab, d_seqs, n, cap, reverse, d_index, d_mm);
    else hipLaunchKernelGGL(match_kernel<uint32_t>, dim3(grid_for(n)), dim3(BLOCK), 0, stream, tab, d_seqs, n, cap, reverse, d_index, d_mm);
    return hipGetLastError();
}

hipError_t launch_synth(const scg_synth_spec& S, char* d_out, int64_t n, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(synth_kernel, dim3(grid_for(n)), dim3(BLOCK), 0, stream, S, d_out, n);
    return hipGetLastError();
}

} // namespace scg
