// scg_ingest.h -- sources of raw FASTQ text for the device-side record scan (scg_textscan.hip).
//
// Replaces, for ordinary files, the byte-at-a-time reader stack of the reference (kaori::FastqReader over
// byteme::PerByte over byteme::RawFileReader / GzipFileReader: inst/include/kaori/FastqReader.hpp:42-110,
// inst/include/byteme/RawFileReader.hpp, GzipFileReader.hpp:39-51, SomeFileReader.hpp:31-44): the host only moves
// bytes -- file pages or inflated blocks -- into pinned windows that start and end on record boundaries; the GPU
// finds and validates the records.  Anything that is not a run of ordinary 4-line records makes the pipeline fall
// back to the sequential reader of scg_fastq.cpp, which reproduces the reference exactly.
#ifndef SCG_INGEST_H
#define SCG_INGEST_H

#include <cstddef>
#include <cstdint>
#include <functional>
#include <memory>
#include <vector>

namespace scg {

// run(n, fn): fn(0) .. fn(n - 1) on a set of host threads that lives as long as the object (the caller runs one share
// itself); exceptions are rethrown.  A window of text is a millisecond or two of work, of which starting sixteen
// threads would be a third.
class WorkerPool {
public:
    explicit WorkerPool(int threads);       // threads - 1 workers; the caller of run() takes a share
    ~WorkerPool();
    void run(int n, const std::function<void(int)>& fn);
    int size() const { return n_threads; }
    WorkerPool(const WorkerPool&) = delete;
    WorkerPool& operator=(const WorkerPool&) = delete;
private:
    struct State;
    State* st;
    int n_threads;
};

// If an ordinary 4-line record starts at data[p], the offset just past it (== size for a final record without its
// trailing newline); otherwise 0.  Optionally reports where its sequence line lies.
size_t strict_record_end(const char* data, size_t size, size_t p, const char** seq = nullptr, size_t* seq_len = nullptr);

// Frees what the device gzip decoder keeps between files (scg_dgzip.cpp; scg_release_buffers()).
void release_device_gunzip_scratch();

// The largest cut <= len such that data[.. cut) ends with two consecutive ordinary records (searched for in the last
// `slack` bytes); 0 if there is none.
size_t find_cut(const char* data, size_t len, size_t slack = size_t(1) << 20);

// The sequences of one window, found by the host (TextSource::next_parsed): each segment is the work of one thread,
// `seq_bytes` bytes of sequences back to back at seqs[seq_at ...) and n_records + 1 byte offsets, relative to the
// segment's first sequence, at offsets[off_at ...).
struct ParsedSegment {
    size_t seq_at = 0, off_at = 0;
    uint32_t n_records = 0, seq_bytes = 0, max_len = 0;
};
struct ParsedWindow {
    ParsedSegment seg[64];
    int n_segs = 0;
    uint64_t n_records = 0, seq_bytes = 0;
    uint32_t max_len = 0;
};

// One gzip member of a batch handed to the device inflater (TextSource::next_members; the layout of
// scg::InflateMember in scg_textscan.h).
struct CompressedMember {
    uint32_t in_off, in_len;      // raw DEFLATE payload within the staging buffer
    uint32_t out_off, out_len;    // its text within the window (relative to the window's first inflated byte)
    uint32_t crc;
};

class TextSource {
public:
    // plain file (mapped), BGZF / blocked gzip (members inflated in parallel) or any other gzip: decoded by all threads
    // at once (scg_pgzip.h; kind() "gzip-parallel") unless the file is small or parallel_gzip is false -- then, and for
    // the second try at a file the parallel decoder handed back, one inflate stream
    // (gzip_threads: threads of the parallel gzip decoder, 0 = `threads`; two mates decoded at once share the host)
    static std::unique_ptr<TextSource> open(const char* path, int threads, bool parallel_gzip = true, int gzip_threads = 0);
    virtual ~TextSource() {}

    // Writes the next window of text into dst[0 .. cap): whole records, starting where the previous window ended.
    // Returns its size; 0 at the end of the input.  When the input ends without a newline one is appended (the
    // reference accepts a final record without it).  cap must be at least min_capacity().
    virtual size_t next(char* dst, size_t cap) = 0;
    // Sources whose text is addressable on the host (plain files) can do the record scan there instead, which halves
    // what goes over the PCIe link: the sequences of the next window of (at most cap bytes of) text go to
    // seqs[0 .. cap), sparsely (see ParsedSegment), their offsets to offsets[0 .. cap_offsets).  Returns the text
    // bytes consumed, 0 at the end of the input; sets unusual() for anything but ordinary 4-line records.
    virtual bool parses() const { return false; }
    virtual size_t next_parsed(char*, size_t, uint32_t*, size_t, ParsedWindow&) { return 0; }
    // BGZF: the members can be inflated independently, which the device does better than sixteen host threads.  The
    // payloads of the next members -- as many as fit `cap` staging bytes (incl. `slack` readable bytes behind the last)
    // and `cap_text` bytes of text -- are copied to staging[0 ...) back to back and described in `members`; `last` says
    // that the input ends with them.  Returns the staging bytes used, 0 at the end of the input (or with unusual()
    // set: a member in a form only zlib should judge).
    virtual bool has_members() const { return false; }
    virtual size_t next_members(char*, size_t, size_t, size_t, std::vector<CompressedMember>&, size_t&, bool&) { return 0; }
    // Text that already lies in HBM (an ordinary gzip file decoded by the device, scg_dgzip.cpp): the next window is
    // copied device to device into d_dst (device `device()`, on `stream`: a hipStream_t) instead of crossing the link twice.
    virtual bool device_resident() const { return false; }
    virtual int device() const { return -1; }
    virtual size_t next_device(char*, size_t, void*) { return 0; }
    // The text could not be cut at a verified record boundary: the caller must redo the file sequentially.
    bool unusual() const { return odd; }
    // An upper estimate of the text bytes still to come (window sizing only).
    virtual uint64_t size_hint() const = 0;
    virtual const char* kind() const = 0;
    static size_t min_capacity() { return size_t(4) << 20; }
    // An ordinary gzip file (one member, or several large ones) decoded on `device`, its text left in HBM; null when the file is not of that
    // kind, too small to bother, too large for the symbol buffers, or anything about it is unusual -- the caller then
    // opens it the ordinary way (scg_dgzip.cpp).
    static std::unique_ptr<TextSource> open_on_device(const char* path, int device, int threads);
    static bool ordinary_gzip(const char* path, int threads);      // a gzip file, not BGZF, of a size the chunked decoders take

protected:
    bool odd = false;
    int threads = 1;
};

} // namespace scg

#endif
