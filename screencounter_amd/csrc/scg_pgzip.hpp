// scg_pgzip.hpp -- ParallelGunzip: the text of an ordinary gzip file, decoded by several host threads (scg_pgzip.h
// explains how).  Replaces, for files it accepts, the one inflate stream of byteme::GzipFileReader
// (inst/include/byteme/GzipFileReader.hpp:39-51); anything it declines is left to that stream.
#ifndef SCG_PGZIP_HPP
#define SCG_PGZIP_HPP

#include <cstddef>
#include <cstdint>

namespace scg {

class ParallelGunzip {
public:
    // data[0 .. size) is the whole file; data must be readable up to size + 16.  threads >= 1 worker threads are
    // started at once and decode ahead of the reader, a bounded number of chunks.
    ParallelGunzip(const uint8_t* data, size_t size, int threads);
    ~ParallelGunzip();
    // The next bytes of text, as many as fit cap (fewer only at the end of the input); 0 at the end, or when the
    // file turned out to be one this decoder does not take (failed()): then everything read so far is void.
    size_t read(char* dst, size_t cap);
    bool failed() const;
    // Compressed bytes per chunk that a file of `size` bytes is cut into; 0: too small to bother.
    static size_t chunk_size_for(size_t size, int threads);
    // Decoders leave their symbol buffers (up to 40 x 32 MB of address space, of which the pages a file's chunks filled
    // are resident: ~12 MB each) for the next one; this frees them.
    static void release_cached();
    ParallelGunzip(const ParallelGunzip&) = delete;
    ParallelGunzip& operator=(const ParallelGunzip&) = delete;
private:
    struct Impl;
    Impl* impl;
};

}  // namespace scg

#endif
