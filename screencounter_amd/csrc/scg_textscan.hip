// scg_textscan.hip -- device-side FASTQ record scan: raw FASTQ text in HBM -> concatenated sequences + offsets.
//
// The host ships windows of raw text (plain file bytes, or the output of the inflate threads) across PCIe exactly once
// and never looks at the records itself; these kernels find the line structure, check that every record of the window
// is an ordinary 4-line record and cut the sequence lines out into the layout the counting kernels consume
// (ScgReads: bytes back to back + uint32 offsets).  They implement the common case of kaori::FastqReader
// (inst/include/kaori/FastqReader.hpp:42-110) -- '@' line, one sequence line without '+', '+' line, one quality line
// as long as the sequence; '\r' stays a base like any other byte -- and REPORT everything else (multi-line records,
// malformed input, stray blank lines, a line count that is not a multiple of four) through TextScanResult::flags, on
// which the host redoes the whole file with the sequential reader that reproduces the reference's behaviour and error
// messages exactly (scg_fastq.cpp).  A window always starts at a record start (scg_ingest.cpp), so the line grouping
// found here is the sequential parser's grouping whenever no flag is raised.
#include <hip/hip_runtime.h>

#include "scg_textscan.h"

namespace {

constexpr int TS_BLOCK = 256;
constexpr int TS_BYTES = 16;                       // text bytes per lane
constexpr int TS_TILE = TS_BLOCK * TS_BYTES;       // text bytes per workgroup

// Bit 7 of byte j set iff byte j of x is '\n' (exact: no borrow between bytes).
__device__ __forceinline__ uint32_t newline_bytes(uint32_t x) {
    const uint32_t y = x ^ 0x0A0A0A0Au;
    return ~(((y & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | y) & 0x80808080u;
}

// 16-bit mask of the newlines among the 16 text bytes at `pos` (bytes at or beyond n are ignored).
__device__ __forceinline__ uint32_t newline_mask(const char* __restrict__ text, uint64_t pos, uint64_t n) {
    if (pos >= n) return 0;
    const uint4 x = *reinterpret_cast<const uint4*>(text + pos);      // the buffer is padded to a whole tile
    const uint32_t d[4] = {x.x, x.y, x.z, x.w};
    uint32_t m = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t t = newline_bytes(d[i]);
        // bits 7, 15, 23, 31 -> bits 0..3
        m |= (((t >> 7) & 1u) | ((t >> 14) & 2u) | ((t >> 21) & 4u) | ((t >> 28) & 8u)) << (4 * i);
    }
    const uint64_t left = n - pos;
    if (left < 16) m &= (1u << left) - 1u;
    return m;
}

__global__ __launch_bounds__(TS_BLOCK) void count_newlines_kernel(const char* __restrict__ text, uint64_t n, uint32_t* __restrict__ block_counts) {
    __shared__ uint32_t part[TS_BLOCK / 64];
    const uint64_t pos = ((uint64_t)blockIdx.x * TS_BLOCK + threadIdx.x) * TS_BYTES;
    uint32_t c = __popc(newline_mask(text, pos, n));
    for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t s = 0;
        for (int w = 0; w < TS_BLOCK / 64; ++w) s += part[w];
        block_counts[blockIdx.x] = s;
    }
}

// In-place exclusive scan of data[0 .. n); data[n] receives the total.  n comes from *n_ptr when n_ptr is given (a
// count produced by an earlier kernel; the grids are sized for `cap`).  Three small kernels: every workgroup scans a tile
// of 8 192 elements in place and notes its total; one workgroup scans the totals; every element receives its tile's
// offset.  (One workgroup doing it all took 1 ms for the 860 k sequence lengths of a 256 MB window -- on the chain
// that links the windows of a BGZF file.)
constexpr int SCAN_BLOCK = 1024;
constexpr int SCAN_PER_LANE = 8;
constexpr uint32_t SCAN_TILE = SCAN_BLOCK * SCAN_PER_LANE;

__device__ __forceinline__ uint32_t scan_count(uint32_t n_fixed, const uint32_t* n_ptr, uint32_t cap) {
    const uint32_t n = n_ptr ? *n_ptr : n_fixed;
    return n > cap ? cap : n;
}

// Exclusive scan over the workgroup of one value per lane; *total = the sum.
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t* total) {
    __shared__ uint32_t wave_sum[SCAN_BLOCK / 64];
    __shared__ uint32_t all;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t incl = v;
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = __shfl_up(incl, off, 64);
        if (lane >= off) incl += o;
    }
    if (lane == 63) wave_sum[wave] = incl;
    __syncthreads();
    if (wave == 0) {
        uint32_t w = lane < SCAN_BLOCK / 64 ? wave_sum[lane] : 0u, wi = w;
        for (int off = 1; off < SCAN_BLOCK / 64; off <<= 1) {
            const uint32_t o = __shfl_up(wi, off, 64);
            if (lane >= off) wi += o;
        }
        if (lane < SCAN_BLOCK / 64) wave_sum[lane] = wi - w;            // exclusive over the wavefronts
        if (lane == SCAN_BLOCK / 64 - 1) all = wi;
    }
    __syncthreads();
    *total = all;
    return wave_sum[wave] + incl - v;
}

__global__ __launch_bounds__(SCAN_BLOCK) void scan_tiles_kernel(uint32_t* __restrict__ data, uint32_t n_fixed, const uint32_t* __restrict__ n_ptr, uint32_t cap,
                                                                uint32_t* __restrict__ tile_sums) {
    const uint32_t n = scan_count(n_fixed, n_ptr, cap);
    const uint32_t base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_PER_LANE;
    if (blockIdx.x * SCAN_TILE >= n) return;                             // (whole workgroup)
    uint32_t v[SCAN_PER_LANE], sum = 0;
#pragma unroll
    for (int k = 0; k < SCAN_PER_LANE; ++k) {
        v[k] = base + k < n ? data[base + k] : 0u;
        sum += v[k];
    }
    uint32_t total;
    uint32_t run = block_exclusive_scan(sum, &total);
#pragma unroll
    for (int k = 0; k < SCAN_PER_LANE; ++k) {
        if (base + k < n) data[base + k] = run;
        run += v[k];
    }
    if (threadIdx.x == 0) tile_sums[blockIdx.x] = total;
}

// Exclusive scan of the tile totals (one workgroup; a window has a few hundred tiles), the grand total to data[n] and *total_out.
__global__ __launch_bounds__(SCAN_BLOCK) void scan_sums_kernel(uint32_t* __restrict__ data, uint32_t n_fixed, const uint32_t* __restrict__ n_ptr, uint32_t cap,
                                                               uint32_t* __restrict__ tile_sums, uint32_t* __restrict__ total_out) {
    const uint32_t n = scan_count(n_fixed, n_ptr, cap);
    const uint32_t tiles = (n + SCAN_TILE - 1) / SCAN_TILE;
    uint32_t carry = 0;
    for (uint32_t t0 = 0; t0 < tiles; t0 += SCAN_BLOCK) {               // (uniform trip count)
        const uint32_t t = t0 + threadIdx.x;
        const uint32_t v = t < tiles ? tile_sums[t] : 0u;
        uint32_t total;
        const uint32_t ex = block_exclusive_scan(v, &total);
        if (t < tiles) tile_sums[t] = carry + ex;
        carry += total;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        data[n] = carry;
        if (total_out) *total_out = carry;
    }
}

__global__ __launch_bounds__(SCAN_BLOCK) void add_sums_kernel(uint32_t* __restrict__ data, uint32_t n_fixed, const uint32_t* __restrict__ n_ptr, uint32_t cap,
                                                              const uint32_t* __restrict__ tile_sums) {
    const uint32_t n = scan_count(n_fixed, n_ptr, cap);
    if (blockIdx.x == 0 || blockIdx.x * SCAN_TILE >= n) return;          // (tile 0 has nothing to add)
    const uint32_t add = tile_sums[blockIdx.x];
    const uint32_t base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_PER_LANE;
#pragma unroll
    for (int k = 0; k < SCAN_PER_LANE; ++k) if (base + k < n) data[base + k] += add;
}

// (stream-ordered) data[0 .. n) -> exclusive prefix sums, data[n] = total; n = *n_ptr (clamped to cap) or n_fixed.
hipError_t launch_exclusive_scan(uint32_t* data, uint32_t n_fixed, const uint32_t* n_ptr, uint32_t cap, uint32_t* tile_sums, uint32_t* total_out,
                                 hipStream_t stream) {
    const uint32_t bound = n_ptr ? cap : (n_fixed < cap ? n_fixed : cap);
    const unsigned tiles = bound ? (bound + SCAN_TILE - 1) / SCAN_TILE : 1;
    hipLaunchKernelGGL(scan_tiles_kernel, dim3(tiles), dim3(SCAN_BLOCK), 0, stream, data, n_fixed, n_ptr, cap, tile_sums);
    hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(SCAN_BLOCK), 0, stream, data, n_fixed, n_ptr, cap, tile_sums, total_out);
    hipLaunchKernelGGL(add_sums_kernel, dim3(tiles), dim3(SCAN_BLOCK), 0, stream, data, n_fixed, n_ptr, cap, (const uint32_t*)tile_sums);
    return hipGetLastError();
}

// nl[k] = byte position of the k-th newline of the window.
__global__ __launch_bounds__(TS_BLOCK) void newline_positions_kernel(const char* __restrict__ text, uint64_t n, const uint32_t* __restrict__ block_base,
                                                                     uint32_t* __restrict__ nl, uint32_t cap_lines) {
    __shared__ uint32_t wave_sum[TS_BLOCK / 64];
    const uint64_t pos = ((uint64_t)blockIdx.x * TS_BLOCK + threadIdx.x) * TS_BYTES;
    uint32_t m = newline_mask(text, pos, n);
    const uint32_t c = __popc(m);
    // exclusive scan of c over the workgroup
    uint32_t incl = c;
    const int lane = threadIdx.x & 63;
    for (int off = 1; off < 64; off <<= 1) {
        uint32_t v = __shfl_up(incl, off, 64);
        if (lane >= off) incl += v;
    }
    if (lane == 63) wave_sum[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint32_t base = block_base[blockIdx.x];
    for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) base += wave_sum[w];
    uint32_t k = base + incl - c;
    while (m) {
        const int j = __ffs((int)m) - 1;
        m &= m - 1;
        if (k < cap_lines) nl[k] = (uint32_t)(pos + j);
        ++k;
    }
}

// One lane per record: structure checks, sequence length (written to lens[i], scanned into offsets afterwards).
__global__ __launch_bounds__(TS_BLOCK) void records_kernel(const char* __restrict__ text, const uint32_t* __restrict__ nl, uint32_t cap_lines,
                                                           uint32_t cap_records, uint32_t* __restrict__ lens, scg::TextScanResult* __restrict__ res, int allow_tail) {
    const uint32_t n_lines = res->n_lines;
    const uint32_t n_rec = n_lines / 4;
    const uint32_t i = blockIdx.x * TS_BLOCK + threadIdx.x;
    if (i == 0) {
        uint32_t f = 0;
        if (n_lines % 4 != 0 && !allow_tail) f |= scg::TEXTSCAN_NOT_FOUR_LINES;
        if (n_lines > cap_lines || n_rec > cap_records) f |= scg::TEXTSCAN_CAPACITY;
        if (f) atomicOr(&res->flags, f);
        res->n_records = n_rec;
        res->cut = (n_rec && !(f & scg::TEXTSCAN_CAPACITY)) ? nl[4 * n_rec - 1] + 1 : 0;
    }
    uint32_t len = 0;
    if (i < n_rec && n_lines <= cap_lines && n_rec <= cap_records) {
        const uint32_t l0 = i ? nl[4 * i - 1] + 1 : 0;     // '@' line
        const uint32_t e0 = nl[4 * i];
        const uint32_t s = e0 + 1, e1 = nl[4 * i + 1];      // sequence line
        const uint32_t p = e1 + 1, e2 = nl[4 * i + 2];      // '+' line
        const uint32_t q = e2 + 1, e3 = nl[4 * i + 3];      // quality line
        len = e1 - s;
        const bool ok = text[l0] == '@' && text[p] == '+' && (e3 - q) == len;
        lens[i] = len;
        if (!ok) atomicOr(&res->flags, scg::TEXTSCAN_MALFORMED);
    }
    // longest read of the window (one atomic per wavefront; every lane takes part in the reduction)
    uint32_t mx = len;
    for (int off = 32; off > 0; off >>= 1) { const uint32_t o = __shfl_down(mx, off, 64); mx = o > mx ? o : mx; }
    if ((threadIdx.x & 63) == 0 && mx) atomicMax(&res->max_len, mx);
}

// One wavefront per record: copies the sequence line to seqs[offsets[i] ...); a '+' inside it would have ended the
// sequence early for the reference's reader (FastqReader.hpp:66-84), which is not an ordinary record.
__global__ __launch_bounds__(TS_BLOCK) void compact_kernel(const char* __restrict__ text, const uint32_t* __restrict__ nl, const uint32_t* __restrict__ offsets,
                                                           char* __restrict__ seqs, uint64_t cap_seq_bytes, scg::TextScanResult* __restrict__ res) {
    if (res->flags & (scg::TEXTSCAN_CAPACITY | scg::TEXTSCAN_NOT_FOUR_LINES)) return;
    const uint32_t n_rec = res->n_records;
    const int lane = threadIdx.x & 63;
    const uint32_t waves = gridDim.x * (TS_BLOCK / 64);
    bool plus = false, over = false;
    for (uint32_t i = blockIdx.x * (TS_BLOCK / 64) + (threadIdx.x >> 6); i < n_rec; i += waves) {
        const uint32_t s = nl[4 * i] + 1, len = nl[4 * i + 1] - s;
        const uint64_t dst = offsets[i];
        if (dst + len > cap_seq_bytes) { over = true; continue; }
        for (uint32_t j = lane; j < len; j += 64) {
            const char c = text[s + j];
            plus |= c == '+';
            seqs[dst + j] = c;
        }
    }
    if (__ballot(plus) && lane == 0) atomicOr(&res->flags, scg::TEXTSCAN_MALFORMED);
    if (__ballot(over) && lane == 0) atomicOr(&res->flags, scg::TEXTSCAN_CAPACITY);
    if (blockIdx.x == 0 && threadIdx.x == 0) res->seq_bytes = offsets[n_rec];
}

// Tiles of 64 KB (sequence bytes) or 16 Ki offsets are dealt to the workgroups round-robin; the sources are pinned host
// memory, read in 16-byte pieces whatever their alignment.
constexpr uint32_t GATHER_TILE = 65536;

__global__ __launch_bounds__(TS_BLOCK) void gather_segments_kernel(char* __restrict__ seqs, uint32_t* __restrict__ offsets, scg::GatherSegments G) {
    if (blockIdx.x == 0 && threadIdx.x == 0) offsets[G.first[G.n]] = G.seq_at[G.n];
    for (uint32_t t = blockIdx.x;; t += gridDim.x) {
        uint32_t k = t, s = 0;
        bool seq_tile = true, found = false;
        for (; s < G.n; ++s) {
            const uint32_t nt = (G.seq_at[s + 1] - G.seq_at[s] + GATHER_TILE - 1) / GATHER_TILE;
            if (k < nt) { found = true; break; }
            k -= nt;
        }
        if (!found) {
            seq_tile = false;
            for (s = 0; s < G.n; ++s) {
                const uint32_t nt = ((G.first[s + 1] - G.first[s]) * 4 + GATHER_TILE - 1) / GATHER_TILE;
                if (k < nt) { found = true; break; }
                k -= nt;
            }
            if (!found) return;
        }
        if (seq_tile) {
            const uint32_t bytes = G.seq_at[s + 1] - G.seq_at[s];
            const uint32_t lo = k * GATHER_TILE, n = min(GATHER_TILE, bytes - lo);
            const char* src = G.seq_src[s] + lo;
            char* dst = seqs + G.seq_at[s] + lo;
            for (uint32_t j = threadIdx.x * 16; j + 16 <= n; j += TS_BLOCK * 16) {
                uint4 v;
                __builtin_memcpy(&v, src + j, 16);
                __builtin_memcpy(dst + j, &v, 16);
            }
            if (threadIdx.x < (n & 15u)) dst[(n & ~15u) + threadIdx.x] = src[(n & ~15u) + threadIdx.x];
        } else {
            const uint32_t count = G.first[s + 1] - G.first[s];
            const uint32_t lo = k * (GATHER_TILE / 4), n = min(GATHER_TILE / 4, count - lo);
            const uint32_t* src = G.off_src[s] + lo;
            uint32_t* dst = offsets + G.first[s] + lo;
            const uint32_t add = G.seq_at[s] - G.off_base[s];
            for (uint32_t j = threadIdx.x; j < n; j += TS_BLOCK) dst[j] = src[j] + add;
        }
    }
}

} // namespace

namespace scg {

hipError_t launch_gather_segments(char* seqs, uint32_t* offsets, const GatherSegments& G, hipStream_t stream) {
    if (G.n == 0 || G.n > 64) return hipErrorInvalidValue;
    // enough workgroups to keep the link busy, few enough to leave the counting kernels of earlier windows their CUs
    hipLaunchKernelGGL(gather_segments_kernel, dim3(128), dim3(TS_BLOCK), 0, stream, seqs, offsets, G);
    return hipGetLastError();
}

size_t text_scan_scratch(size_t cap_blocks, size_t cap_records) {
    const size_t n = cap_blocks > cap_records ? cap_blocks : cap_records;
    return (n + SCAN_TILE - 1) / SCAN_TILE + 2;
}
size_t text_scan_blocks(size_t n_bytes) { return (n_bytes + TS_TILE - 1) / TS_TILE; }
size_t text_scan_padded(size_t n_bytes) { return text_scan_blocks(n_bytes) * TS_TILE; }

hipError_t launch_text_scan(const char* d_text, size_t n_bytes, const TextScanBuffers& B, hipStream_t stream, bool allow_tail, hipEvent_t structure_known) {
    hipError_t e = hipMemsetAsync(B.result, 0, sizeof(TextScanResult), stream);
    if (e != hipSuccess) return e;
    if (n_bytes == 0) return hipSuccess;
    const unsigned blocks = (unsigned)text_scan_blocks(n_bytes);
    if (blocks + 1 > B.cap_blocks) return hipErrorInvalidValue;
    hipLaunchKernelGGL(count_newlines_kernel, dim3(blocks), dim3(TS_BLOCK), 0, stream, d_text, (uint64_t)n_bytes, B.block_counts);
    e = launch_exclusive_scan(B.block_counts, blocks, nullptr, blocks, B.scan_scratch, &B.result->n_lines, stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(newline_positions_kernel, dim3(blocks), dim3(TS_BLOCK), 0, stream, d_text, (uint64_t)n_bytes, B.block_counts, B.nl, (uint32_t)B.cap_lines);
    const unsigned rec_blocks = (unsigned)((B.cap_records + TS_BLOCK - 1) / TS_BLOCK);
    // a window holds at most n_bytes / 6 records (six bytes is the shortest 4-line record): never launch more lanes than that
    const unsigned need = (unsigned)((n_bytes / 6 + TS_BLOCK) / TS_BLOCK);
    hipLaunchKernelGGL(records_kernel, dim3(need < rec_blocks ? need : rec_blocks), dim3(TS_BLOCK), 0, stream, d_text, B.nl, (uint32_t)B.cap_lines,
                       (uint32_t)B.cap_records, B.offsets, B.result, allow_tail ? 1 : 0);
    if (structure_known) {
        e = hipEventRecord(structure_known, stream);
        if (e != hipSuccess) return e;
    }
    e = launch_exclusive_scan(B.offsets, 0u, &B.result->n_records, (uint32_t)B.cap_records, B.scan_scratch, nullptr, stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(compact_kernel, dim3(2048), dim3(TS_BLOCK), 0, stream, d_text, B.nl, B.offsets, B.seqs, (uint64_t)B.cap_seq_bytes, B.result);
    return hipGetLastError();
}

} // namespace scg
