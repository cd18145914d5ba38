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
                                                                scg::CrcPowers P, uint32_t* __restrict__ status) {
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
            if (crc != M.crc) atomicOr(status, scg::INFLATE_STATUS_CRC);
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
    hipLaunchKernelGGL(inflate_members_kernel, dim3(n), dim3(INFLATE_BLOCK), 0, stream, d_in, d_members, n,
                       reinterpret_cast<uint8_t*>(d_text), d_status);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(crc_members_kernel, dim3(n < 4096 ? n : 4096), dim3(CRC_BLOCK), 0, stream, reinterpret_cast<const uint8_t*>(d_text), d_members, n,
                       crc_powers(), d_status);
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
