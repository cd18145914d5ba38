// scg_fastq.cpp -- host FASTQ staging: file -> batches of concatenated sequence bytes + offsets.
//
// Replaces kaori::FastqReader (inst/include/kaori/FastqReader.hpp:42-110) over
// byteme::SomeFileReader (inst/include/byteme/SomeFileReader.hpp:31-44).  Same record grammar,
// including its quirks:
//   - a record starts with '@'; the name line runs to the first '\n';
//   - the sequence is every byte up to the first '+', newlines removed (multi-line allowed,
//     '\r' is kept as a base);
//   - the '+' line is skipped; qualities are consumed until a newline is met with at least as
//     many quality bytes as sequence bytes (or EOF), and the two lengths must then agree;
//   - the last record may lack its trailing newline; an empty file holds zero reads;
//   - "line numbers" in messages advance by exactly four per record, as the reference counts them.
// Unlike the reference's byte-at-a-time virtual-call loop this scanner works on large buffers
// with memchr, and only the sequence bytes are kept (names and qualities never leave this file).
#include "scg_host.h"

#include <cctype>
#include <cstdio>
#include <cstring>
#include <fcntl.h>
#include <unistd.h>
#include <zlib.h>

namespace scg {

struct FastqStream::Impl {
    std::string path;
    bool gz = false;
    gzFile gzf = nullptr;
    int fd = -1;
    std::vector<unsigned char> buf;
    const unsigned char* cur = nullptr;
    const unsigned char* end = nullptr;
    bool eof = false;
    bool okay = false;       // a record starts at *cur
    int64_t line_count = 0;

    explicit Impl(const char* p) : path(p), buf(size_t(8) << 20) {
        unsigned char header[3];
        size_t got = 0;
        {
            FILE* f = std::fopen(p, "rb");
            if (!f) {
                // byteme/SelfClosingFILE.hpp / SelfClosingGzFile.hpp:13
                throw Error(SCG_ERR_IO, "failed to open file at '" + path + "'");
            }
            got = std::fread(header, 1, 3, f);
            std::fclose(f);
        }
        // byteme/magic_numbers.hpp:19-22
        gz = (got >= 2 && header[0] == 0x1f && header[1] == 0x8b);
        if (gz) {
            gzf = gzopen(p, "rb");
            if (!gzf) throw Error(SCG_ERR_IO, "failed to open file at '" + path + "'");
            gzbuffer(gzf, 1 << 20);
        } else {
            fd = ::open(p, O_RDONLY);
            if (fd < 0) throw Error(SCG_ERR_IO, "failed to open file at '" + path + "'");
        }
        refill();
        okay = cur < end;    // kaori/FastqReader.hpp:32
    }

    ~Impl() {
        if (gzf) gzclose(gzf);
        if (fd >= 0) ::close(fd);
    }

    bool refill() {
        if (eof) { cur = end = buf.data(); return false; }
        long got;
        if (gz) {
            got = gzread(gzf, buf.data(), static_cast<unsigned>(buf.size()));
            if (got < 0) {
                int dummy;
                throw Error(SCG_ERR_IO, gzerror(gzf, &dummy));   // byteme/GzipFileReader.hpp:47
            }
        } else {
            got = ::read(fd, buf.data(), buf.size());
            if (got < 0) throw Error(SCG_ERR_IO, "failed to read raw binary file");   // byteme/RawFileReader.hpp:60
        }
        cur = buf.data();
        end = cur + got;
        if (got == 0) { eof = true; return false; }
        return true;
    }

    // Moves past the current byte; false at end of input (PerByte::advance).
    bool advance() {
        ++cur;
        if (cur < end) return true;
        return refill();
    }

    void advance_or_throw() {
        if (!advance()) {
            throw Error(SCG_ERR_IO, "premature end of the file at line " + std::to_string(line_count + 1));   // FastqReader.hpp:115-120
        }
    }

    // Skips to the next '\n' (which becomes the current byte); throws on EOF.
    void skip_to_newline() {
        for (;;) {
            const void* nl = std::memchr(cur, '\n', static_cast<size_t>(end - cur));
            if (nl) { cur = static_cast<const unsigned char*>(nl); return; }
            cur = end - 1;
            advance_or_throw();
        }
    }

    // Parses one record starting at *cur, appending its sequence to `seqs`.
    void record(std::vector<char>& seqs) {
        int64_t init_line = line_count;
        if (*cur != '@') {
            throw Error(SCG_ERR_IO, "read name should start with '@' (starting line " + std::to_string(init_line + 1) + ")");
        }
        advance_or_throw();
        skip_to_newline();          // name up to the first whitespace, rest of the line ignored
        ++line_count;

        size_t start = seqs.size();
        advance_or_throw();
        for (;;) {                  // sequence: up to the first '+', newlines dropped
            size_t avail = static_cast<size_t>(end - cur);
            const unsigned char* nl = static_cast<const unsigned char*>(std::memchr(cur, '\n', avail));
            size_t span = nl ? static_cast<size_t>(nl - cur) : avail;
            const unsigned char* plus = static_cast<const unsigned char*>(std::memchr(cur, '+', span));
            if (plus) {
                seqs.insert(seqs.end(), cur, plus);
                cur = plus;
                break;
            }
            seqs.insert(seqs.end(), cur, cur + span);
            if (nl) {
                cur = nl;           // on the newline: step over it
            } else {
                cur = end - 1;
            }
            advance_or_throw();
        }
        ++line_count;

        advance_or_throw();         // rest of the '+' line
        skip_to_newline();
        ++line_count;

        size_t seq_len = seqs.size() - start, qual_len = 0;
        okay = false;
        while (advance()) {
            // consume quality bytes up to the next newline in one step where possible
            size_t avail = static_cast<size_t>(end - cur);
            const unsigned char* nl = static_cast<const unsigned char*>(std::memchr(cur, '\n', avail));
            if (!nl) {
                qual_len += avail;
                cur = end - 1;
                continue;
            }
            qual_len += static_cast<size_t>(nl - cur);
            cur = nl;
            if (qual_len >= seq_len) {
                okay = advance();   // sneak past the newline
                break;
            }
        }
        if (qual_len != seq_len) {
            throw Error(SCG_ERR_IO, "non-equal lengths for quality and sequence strings (starting line " + std::to_string(init_line + 1) + ")");
        }
        ++line_count;
    }
};

FastqStream::FastqStream(const char* path) : impl(new Impl(path)) {}

FastqStream::~FastqStream() { delete impl; }

bool FastqStream::next_batch(ReadBatch& out, int64_t max_reads, int64_t max_bytes) {
    out.clear();
    while (impl->okay && out.size() < max_reads && static_cast<int64_t>(out.seqs.size()) < max_bytes) {
        impl->record(out.seqs);
        out.offsets.push_back(out.seqs.size());
    }
    return out.size() > 0;
}

} // namespace scg
