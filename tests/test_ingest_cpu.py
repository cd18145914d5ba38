"""Host side of the FASTQ staging (csrc/scg_ingest.cpp): the raw-text windows the file-level entry points ship to the GPU,
and the host-side record scan of plain files (sequences + offsets only go over the link).
Plain files, BGZF (members inflated in parallel) and ordinary gzip (one inflate stream, multi-member included) must all
yield the decompressed text cut into windows of whole 4-line records; text that cannot be cut that way is declined
(the counting calls then take the sequential reader).  No device needed."""
import ctypes as C
import gzip
import os
import random

import pytest

from tests import gen


def text_windows(sc, path, window, threads=4):
    from screencounter_amd import _lib
    L = sc.load()
    text, cuts = C.c_void_p(), C.c_void_p()
    n, nw = C.c_int64(0), C.c_int64(0)
    kind = C.create_string_buffer(16)
    err = _lib.errbuf()
    rc = L.scg_fastq_text_windows(os.fspath(path).encode(), int(window), int(threads), C.byref(text), C.byref(n), C.byref(cuts), C.byref(nw),
                                  kind, err, _lib.ERRCAP)
    if rc:
        raise _lib.ScgError(rc, err.value.decode())
    try:
        data = C.string_at(text, n.value)
        c = list((C.c_int64 * (nw.value + 1)).from_address(cuts.value))
    finally:
        L.scg_free(text)
        L.scg_free(cuts)
    return data, c, kind.value.decode()


def scan_windows(sc, path, window, threads=4):
    """scg_fastq_scan_windows -> (list of sequences, number of windows)."""
    from screencounter_amd import _lib
    import numpy as np
    L = sc.load()
    seqs, offs = C.c_void_p(), C.c_void_p()
    n, nw = C.c_int64(0), C.c_int64(0)
    err = _lib.errbuf()
    rc = L.scg_fastq_scan_windows(os.fspath(path).encode(), int(window), int(threads), C.byref(seqs), C.byref(offs), C.byref(n), C.byref(nw),
                                  err, _lib.ERRCAP)
    if rc:
        raise _lib.ScgError(rc, err.value.decode())
    try:
        o = np.ctypeslib.as_array(C.cast(offs, C.POINTER(C.c_uint64)), shape=(n.value + 1,)).copy()
        data = C.string_at(seqs, int(o[-1]))
    finally:
        L.scg_free(seqs)
        L.scg_free(offs)
    return [data[int(a):int(b)] for a, b in zip(o[:-1], o[1:])], nw.value


def strict_records(chunk: bytes):
    """Parses whole ordinary 4-line records; returns the sequences (asserts the chunk holds nothing else)."""
    lines = chunk.split(b"\n")
    assert lines[-1] == b"", "window does not end with a newline"
    lines = lines[:-1]
    assert len(lines) % 4 == 0
    seqs = []
    for i in range(0, len(lines), 4):
        assert lines[i].startswith(b"@") and lines[i + 2].startswith(b"+") and len(lines[i + 1]) == len(lines[i + 3])
        seqs.append(lines[i + 1])
    return seqs


def random_reads(rng, n, lo=20, hi=160):
    return ["".join(rng.choice("ACGTN") for _ in range(rng.randint(lo, hi))) for _ in range(n)]


@pytest.mark.parametrize("form", ["plain", "bgzf", "bgzf_no_eof", "gzip", "gzip_multi"])
@pytest.mark.parametrize("window", [4096, 70_001, 1 << 20])
def test_windows_reassemble_the_text(sc, tmp_path, form, window):
    rng = random.Random(hash((form, window)) & 0xFFFF)
    reads = random_reads(rng, 3000)
    text = gen.fastq_text(reads, trailing_newline=(window != 70_001))      # one size also covers "no final newline"
    path = str(tmp_path / ("x.fastq" + ("" if form == "plain" else ".gz")))
    if form == "plain":
        open(path, "wb").write(text)
    elif form.startswith("bgzf"):
        gen.write_bgzf(path, text, block=(900 if window == 4096 else rng.choice([900, 5000, 60000])), eof_block=(form == "bgzf"))
    elif form == "gzip":
        with gzip.open(path, "wb") as f:
            f.write(text)
    else:
        with open(path, "wb") as f:                                       # concatenated members without size fields
            third = len(text) // 3
            for part in (text[:third], text[third:2 * third], text[2 * third:]):
                f.write(gzip.compress(part))
    data, cuts, kind = text_windows(sc, path, window)
    assert kind == {"plain": "plain", "bgzf": "bgzf", "bgzf_no_eof": "bgzf", "gzip": "gzip", "gzip_multi": "gzip"}[form]
    assert data == (text if text.endswith(b"\n") else text + b"\n")
    assert cuts[0] == 0 and cuts[-1] == len(data) and all(b > a for a, b in zip(cuts, cuts[1:]))
    assert all(b - a <= window for a, b in zip(cuts, cuts[1:]))
    got = []
    for a, b in zip(cuts, cuts[1:]):
        got += strict_records(data[a:b])
    assert got == [r.encode() for r in reads]
    if window == 4096:
        assert len(cuts) > 50                                              # really cut into many windows


def test_empty_inputs(sc, tmp_path):
    p = tmp_path / "empty.fastq"
    p.write_bytes(b"")
    data, cuts, kind = text_windows(sc, p, 4096)
    assert data == b"" and cuts == [0] and kind == "plain"
    g = tmp_path / "empty.fastq.gz"
    g.write_bytes(gzip.compress(b""))
    data, cuts, kind = text_windows(sc, g, 4096)
    assert data == b"" and cuts == [0]
    b = tmp_path / "empty_bgzf.gz"
    gen.write_bgzf(str(b), b"")
    data, cuts, kind = text_windows(sc, b, 4096)
    assert data == b"" and kind == "bgzf"


def test_text_that_cannot_be_cut_is_declined(sc, tmp_path):
    from screencounter_amd import _lib
    # multi-line records (legal for the reference's reader, FastqReader.hpp:66-84): no window boundary can be verified
    rec = b"@r\nACGT\nACGT\n+\nIIII\nIIII\n"
    p = tmp_path / "multi.fastq"
    p.write_bytes(rec * 2000)
    with pytest.raises(_lib.ScgError) as e:
        text_windows(sc, p, 4096)
    assert e.value.code == _lib.SCG_ERR_UNSUPPORTED
    # ... but a file that fits one window is handed over whole (the device scan then declines it)
    data, cuts, _ = text_windows(sc, p, 1 << 20)
    assert data == rec * 2000 and len(cuts) == 2


def test_corrupt_gzip_member_is_an_io_error(sc, tmp_path):
    from screencounter_amd import _lib
    text = gen.fastq_text(random_reads(random.Random(5), 500))
    p = str(tmp_path / "bad.gz")
    gen.write_bgzf(p, text, block=5000)
    raw = bytearray(open(p, "rb").read())
    raw[len(raw) // 2] ^= 0x55
    open(p, "wb").write(bytes(raw))
    with pytest.raises(_lib.ScgError) as e:
        text_windows(sc, p, 1 << 20)
    assert e.value.code in (_lib.SCG_ERR_IO, _lib.SCG_ERR_UNSUPPORTED)


def test_missing_file(sc, tmp_path):
    from screencounter_amd import _lib
    with pytest.raises(_lib.ScgError) as e:
        text_windows(sc, tmp_path / "nope.fastq", 4096)
    assert e.value.code == _lib.SCG_ERR_IO and "failed to open file" in str(e.value)


@pytest.mark.parametrize("window", [4096, 70_001, 1 << 20])
@pytest.mark.parametrize("threads", [1, 4, 7])
def test_host_record_scan_of_plain_files(sc, tmp_path, window, threads):
    rng = random.Random(window * 31 + threads)
    reads = random_reads(rng, 4000, lo=0 if window != 4096 else 1, hi=160)
    text = gen.fastq_text(reads, trailing_newline=(window != 70_001))
    p = tmp_path / "x.fastq"
    p.write_bytes(text)
    got, nw = scan_windows(sc, p, window, threads)
    assert got == [r.encode() for r in reads]
    assert nw >= len(text) // window
    if window == 4096:
        assert nw > 50


def test_host_record_scan_declines_what_the_device_scan_declines(sc, tmp_path, monkeypatch):
    from screencounter_amd import _lib
    good = gen.fastq_text(random_reads(random.Random(3), 50))
    cases = {
        "multi_line": b"@r\nACGT\nACGT\n+\nIIII\nIIII\n" * 50,
        "plus_in_sequence": good + b"@r\nAC+GT\n+\nIIIII\n" + good,
        "quality_length": good + b"@r\nACGT\n+\nIII\n" + good,
        "no_at": good + b"r\nACGT\n+\nIIII\n" + good,
        "blank_line": good + b"\n" + good,
        "truncated": good + b"@r\nACGT\n+\n",
        "crlf_quality_longer": good + b"@r\nACGT\n+\nIIII\r\n" + good,
    }
    for name, text in cases.items():
        p = tmp_path / (name + ".fastq")
        p.write_bytes(text)
        for window in (1 << 20, 2048):
            with pytest.raises(_lib.ScgError) as e:
                scan_windows(sc, p, window)
            assert e.value.code == _lib.SCG_ERR_UNSUPPORTED, name
    # gzip: inflated whole (libdeflate, when the image has it) and scanned like a plain file; streamed through zlib
    # otherwise, which leaves the record scan to the device
    g = tmp_path / "x.fastq.gz"
    g.write_bytes(gzip.compress(good[:1000]) + gzip.compress(good[1000:]))
    import ctypes.util
    if ctypes.util.find_library("deflate"):
        assert b"\n".join(scan_windows(sc, g, 2048)[0]) == b"\n".join(strict_records(good))
    monkeypatch.setenv("SCG_LIBDEFLATE", "0")
    with pytest.raises(_lib.ScgError) as e:
        scan_windows(sc, g, 1 << 20)
    assert e.value.code == _lib.SCG_ERR_UNSUPPORTED and "gzip" in str(e.value)
    monkeypatch.delenv("SCG_LIBDEFLATE")
    e0 = tmp_path / "empty.fastq"
    e0.write_bytes(b"")
    assert scan_windows(sc, e0, 4096) == ([], 0)
    # CRLF files whose lines all carry the '\r' are ordinary records for the reference too ('\r' is part of each line)
    crlf = tmp_path / "crlf.fastq"
    crlf.write_bytes(b"@r1\r\nACGT\r\n+\r\nIIII\r\n@r2\r\nGG\r\n+\r\nII\r\n")
    assert scan_windows(sc, crlf, 4096)[0] == [b"ACGT\r", b"GG\r"]


def member_batches(sc, path, staging, text, threads=4):
    """scg_bgzf_member_batches -> (rows of (batch, payload offset, payload length, text offset, text length, crc), payload bytes, batches)."""
    from screencounter_amd import _lib
    import numpy as np
    L = sc.load()
    table, payloads = C.c_void_p(), C.c_void_p()
    n, nb, nbatch = C.c_int64(0), C.c_int64(0), C.c_int64(0)
    err = _lib.errbuf()
    rc = L.scg_bgzf_member_batches(os.fspath(path).encode(), int(staging), int(text), int(threads), C.byref(table), C.byref(n), C.byref(payloads),
                                   C.byref(nb), C.byref(nbatch), err, _lib.ERRCAP)
    if rc:
        raise _lib.ScgError(rc, err.value.decode())
    try:
        rows = np.ctypeslib.as_array(C.cast(table, C.POINTER(C.c_uint32)), shape=(max(n.value, 1) * 6,))[: n.value * 6].reshape(-1, 6).copy()
        data = C.string_at(payloads, nb.value)
    finally:
        L.scg_free(table)
        L.scg_free(payloads)
    return rows, data, nbatch.value


@pytest.mark.parametrize("block", [700, 20000, 65280])
def test_bgzf_member_batches_for_the_device_inflater(sc, tmp_path, block):
    """What the GPU is handed for a BGZF file: every member's raw DEFLATE payload, its place in the batch's text, its
    size and CRC; batches within the staging and text limits; the text reassembled from the payloads is the file's."""
    import zlib
    from screencounter_amd import _lib
    rng = random.Random(block)
    text = gen.fastq_text(random_reads(rng, 4000), trailing_newline=False)
    p = str(tmp_path / "x.bgzf.gz")
    gen.write_bgzf(p, text, block=block, eof_block=True)
    staging, limit = 150_000, 200_000
    rows, data, batches = member_batches(sc, p, staging, limit)
    assert batches > 3 and len(rows) >= len(text) // block
    out, at_batch, batch_text = [], -1, 0
    for b, off, n, toff, tlen, crc in rows.tolist():
        if b != at_batch:
            assert b == at_batch + 1 and toff == 0
            at_batch, batch_text, batch_in = b, 0, 0
        assert toff == batch_text
        d = zlib.decompressobj(-15)
        piece = d.decompress(data[off:off + n]) + d.flush()
        assert d.eof and d.unused_data == b"" and len(piece) == tlen and (zlib.crc32(piece) & 0xFFFFFFFF) == crc
        out.append(piece)
        batch_text += tlen
        batch_in += n
        assert batch_text <= limit and batch_in <= staging
    assert b"".join(out) == text
    # input the device inflater does not take
    g = tmp_path / "plain.gz"
    g.write_bytes(gzip.compress(text))
    for path, lim in ((g, limit), (p, block // 2 if block > 1000 else 10)):
        with pytest.raises(_lib.ScgError) as e:
            member_batches(sc, path, staging, lim)
        assert e.value.code == _lib.SCG_ERR_UNSUPPORTED


def test_host_scan_of_gzip_text_inflated_whole_is_stable(sc, tmp_path, monkeypatch):
    """A gzip file inflated whole lies in anonymous memory, where dropping a page (MADV_DONTNEED) zeroes it: the host
    scan's threads work out their slices' ends from text in their neighbours' slices, so a window's pages may only go when
    all threads are done.  Dropped early, one run in a few lost the records of a slice (27 of 3 000 reads in the GPU
    fuzz).  Many small windows, more threads than cores, repeated."""
    rng = random.Random(90790)
    reads = random_reads(rng, 3000)
    text = gen.fastq_text(reads)
    p = str(tmp_path / "whole.fastq.gz")
    import zlib
    c = zlib.compressobj(6, zlib.DEFLATED, 31)
    open(p, "wb").write(c.compress(text) + c.flush())
    monkeypatch.setenv("SCG_PGZIP", "0")
    want = [r.encode() for r in reads]
    for rep in range(40):
        got, windows = scan_windows(sc, p, 100 << 10, threads=8)
        assert got == want, (rep, len(got))
        assert windows > 5
