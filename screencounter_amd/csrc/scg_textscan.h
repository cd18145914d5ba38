// scg_textscan.h -- device-side FASTQ record scan (scg_textscan.hip): raw text window -> sequences + offsets.
#ifndef SCG_TEXTSCAN_H
#define SCG_TEXTSCAN_H

#include <hip/hip_runtime_api.h>
#include <stddef.h>
#include <stdint.h>

namespace scg {

enum : uint32_t {
    TEXTSCAN_MALFORMED = 1,        // some record is not an ordinary 4-line record
    TEXTSCAN_NOT_FOUR_LINES = 2,   // the window's line count is not a multiple of four
    TEXTSCAN_CAPACITY = 4,         // more lines / records / sequence bytes than the slot's buffers hold (very short lines)
};

// Written by the device, copied back by the host before the counting kernels of the window are launched.
struct TextScanResult {
    uint32_t n_lines;
    uint32_t n_records;
    uint32_t max_len;      // longest sequence of the window
    uint32_t flags;        // TEXTSCAN_*: non-zero => the host falls back to the sequential reference-exact reader
    uint64_t seq_bytes;    // total sequence bytes (= offsets[n_records])
    uint32_t cut;          // allow_tail scans: the byte behind the last whole record (everything from there on is carried over)
    uint32_t reserved;
};

// Device scratch and outputs of one window.
struct TextScanBuffers {
    uint32_t* block_counts;   // [cap_blocks]: newline count per 4 KiB text tile, then their exclusive scan
    uint32_t* nl;             // [cap_lines]: byte position of every newline
    uint32_t* offsets;        // [cap_records + 1]: out, byte offset of every sequence in `seqs`
    char* seqs;               // [cap_seq_bytes]: out, the sequences back to back
    TextScanResult* result;
    uint32_t* scan_scratch;   // [text_scan_scratch(cap_blocks, cap_records)]: tile totals of the prefix scans
    size_t cap_blocks, cap_lines, cap_records, cap_seq_bytes;
};

size_t text_scan_blocks(size_t n_bytes);    // text tiles of a window
size_t text_scan_padded(size_t n_bytes);    // device text buffers must be readable up to this many bytes
size_t text_scan_scratch(size_t cap_blocks, size_t cap_records);   // entries of TextScanBuffers::scan_scratch

// Asynchronous on `stream`.  The text must start at a record start; the last byte of the final window of a file must be
// a newline (the host appends one when the file lacks it, as the reference accepts a final record without it).
// allow_tail: the text may end anywhere (windows cut at gzip member boundaries): the records are the whole groups of
// four lines, and `cut` says where the rest begins.
// `structure_known`, if given, is recorded once n_records, cut and the structure flags are in B.result (the sequences
// are still being cut out then): what the next window of a chained pipeline waits for.
hipError_t launch_text_scan(const char* d_text, size_t n_bytes, const TextScanBuffers& B, hipStream_t stream, bool allow_tail = false,
                            hipEvent_t structure_known = nullptr);

// Host-side record scan (scg_ingest.h): the sequences and offsets of a window lie in pinned host memory in segments,
// one per host thread, the offsets of each relative to its own first sequence.  One kernel pulls them over the link
// (zero-copy reads run at link speed, 56 GB/s, where sixteen hipMemcpyAsync calls of 4 MB reach 32-37 GB/s and cost the
// host 48 us each: tools/ubench/zero_copy.hip) and lays them out back to back:
//   seqs[seq_at[s] ...) = seq_src[s][0 .. seq_at[s+1] - seq_at[s]),
//   offsets[first[s] + j] = off_src[s][j] - off_base[s] + seq_at[s] for j < first[s+1] - first[s],   offsets[first[n]] = seq_at[n].
// (off_base[s] = off_src[s][0] when a segment is a stretch from the middle of what a host thread parsed: its offsets
// count from that thread's first sequence)
struct GatherSegments {
    uint32_t n;                  // segments (<= 64)
    const char* seq_src[64];     // device-visible host pointers
    const uint32_t* off_src[64];
    uint32_t seq_at[65];
    uint32_t first[65];
    uint32_t off_base[64];
};
hipError_t launch_gather_segments(char* seqs, uint32_t* offsets, const GatherSegments& G, hipStream_t stream);

// ---- BGZF members inflated on the device (scg_inflate.hip) ----
struct InflateMember {
    uint32_t in_off, in_len;      // the member's raw DEFLATE payload within the compressed buffer
    uint32_t out_off, out_len;    // where its text goes within the window's text buffer, and how much (ISIZE)
    uint32_t crc;                 // CRC-32 of the text (gzip trailer)
};
enum : uint32_t {
    INFLATE_STATUS_BAD = 1,       // some member is not a valid DEFLATE stream of the announced sizes
    INFLATE_STATUS_CRC = 2,       // some member's text does not match its CRC-32
    INFLATE_STATUS_TAIL = 4,      // the partial record carried over from the previous window does not fit the gap
};
struct CrcPowers { uint32_t x2n[32]; };       // x^(2^k) mod the CRC-32 polynomial
size_t inflate_input_slack();                 // bytes that must be readable behind the compressed buffer's last payload
// Inflates members[0 .. n) from d_in into d_text and checks their CRCs; failures are ORed into *d_status.
hipError_t launch_inflate_members(const uint8_t* d_in, const InflateMember* d_members, uint32_t n, char* d_text, uint32_t* d_status, hipStream_t stream);
// text[0 .. gap) = a dummy record with an empty sequence + the bytes of prev_text[0 .. prev_bytes) behind prev_result->cut
// (prev_text == nullptr: the first window, nothing to carry).
hipError_t launch_carry_tail(const char* prev_text, const TextScanResult* prev_result, uint32_t prev_bytes, char* text, uint32_t gap, uint32_t* d_status,
                             hipStream_t stream);

// ---- ordinary gzip decoded on the device (scg_inflate.hip; host side: scg_dgzip.cpp) ----
struct GunzipChunk {
    uint64_t start_bit;           // where the chunk's first block begins (absolute bit in the file; ~0: none found, merged into the chunk before)
    uint64_t end_bit;             // where its last block ended
    uint32_t status;              // scginf::INFLATE_* of its decoding
    uint32_t made;                // symbols written
    uint32_t final_block;         // its last block was the stream's last
    uint32_t pad;
};
// d_in holds the file's bytes [origin, size); d_chunks[k] is chunk chunk0 + k of the stream; all n are searched for their block
// starts (but the first, whose start is given), the first n_decode decoded.
hipError_t launch_gunzip_find(const uint8_t* d_in, uint64_t origin, uint64_t size, GunzipChunk* d_chunks, uint32_t n, uint64_t chunk0, uint64_t first_byte,
                              uint64_t chunk_bytes, uint64_t stream_end_byte, hipStream_t stream);
hipError_t launch_gunzip_decode(const uint8_t* d_in, uint64_t origin, uint64_t size, GunzipChunk* d_chunks, uint32_t n, uint32_t n_decode, uint16_t* d_syms,
                                uint64_t cap_syms, hipStream_t stream);
// The tails' scratch (scg_inflate.hip, "The tails"): chunks are taken in groups of `group`; per group a map of 32 Ki 16-bit
// entries (maps; the last group needs none), a window of 32 KiB (wins) and a count (avails).
struct GunzipTailScratch { uint32_t group; uint16_t* maps; uint8_t* wins; uint32_t* avails; };
hipError_t launch_gunzip_text(const uint16_t* d_syms, uint64_t cap_syms, const GunzipChunk* d_chunks, const uint64_t* d_text_at, uint32_t n, char* d_text,
                              uint64_t floor, const GunzipTailScratch& scratch, uint32_t* d_status, hipStream_t stream);   // (d_text[floor]: the member's first byte, or later)
hipError_t launch_crc_pieces(const char* d_text, const InflateMember* d_members, uint32_t n, uint32_t* d_crcs, hipStream_t stream);

} // namespace scg

#endif
