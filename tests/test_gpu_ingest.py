"""The ingestion paths behind the file-level entry points -- plain files scanned for records by the host threads
(sequences + offsets -> HBM) or, like BGZF and gzip always, shipped as raw text and scanned on the device
(csrc/scg_textscan.hip + scg_ingest.cpp): every input form (plain, BGZF, gzip), tiny windows that force many hand-overs and carries,
several pipelines (devices) sharing one file, the host-parser path as a cross-check, the multi-file entries against
per-file calls, and the fall-back to the sequential reader for everything that is not a run of ordinary records."""
import gzip
import os
import random

import numpy as np
import pytest

from tests import gen

pytestmark = pytest.mark.gpu

TEMPLATE = "ACGTACGA" + "-" * 12 + "TGCATGCA"


@pytest.fixture(params=["host_scan", "device_scan"])
def plain_scan(request, monkeypatch):
    """Who scans plain files for records: the host threads (default) or the device (SCG_HOST_SCAN=0)."""
    if request.param == "device_scan":
        monkeypatch.setenv("SCG_HOST_SCAN", "0")
    return request.param


@pytest.fixture(params=["device_inflate", "host_inflate"])
def bgzf_inflate(request, monkeypatch):
    """Who inflates the members of BGZF files in single-end calls: the device (default) or the host threads' zlib."""
    # 2 = the device, and handing a file back to the host is an error (these files are well-formed)
    monkeypatch.setenv("SCG_DEVICE_INFLATE", "0" if request.param == "host_inflate" else "2")
    return request.param


def make_case(seed, n=6000):
    rng = random.Random(seed)
    pool = gen.make_pool(rng, 60, 12, "ACGT")
    reads = gen.make_reads(rng, TEMPLATE, [pool], n, 2, 0.03, 0.01, 0.02, 0.1, 40)
    return pool, reads


def write_forms(tmp_path, reads, trailing_newline=True):
    text = gen.fastq_text(reads, trailing_newline=trailing_newline)
    paths = {}
    paths["plain"] = str(tmp_path / "r.fastq")
    open(paths["plain"], "wb").write(text)
    paths["bgzf"] = str(tmp_path / "r.bgzf.gz")
    gen.write_bgzf(paths["bgzf"], text, block=3000)
    paths["gzip"] = str(tmp_path / "r.fastq.gz")
    with gzip.open(paths["gzip"], "wb") as f:
        f.write(text)
    return paths


@pytest.mark.parametrize("window_kb", [None, 8])
@pytest.mark.parametrize("trailing_newline", [True, False])
def test_every_input_form_counts_like_the_oracle(sc, oracle, gpu, tmp_path, monkeypatch, plain_scan, bgzf_inflate, window_kb, trailing_newline):
    pool, reads = make_case(11)
    exp, total = oracle.count_single(reads, TEMPLATE, 2, pool, 1, True)
    paths = write_forms(tmp_path, reads, trailing_newline)
    if window_kb:
        monkeypatch.setenv("SCG_WINDOW_KB", str(window_kb))            # ~1 MB of text in 8 KB windows: > 100 hand-overs
    for form, path in paths.items():
        for use_first in (True, False):
            e, t = (exp, total) if use_first else oracle.count_single(reads, TEMPLATE, 2, pool, 1, False)
            got, n = sc.count_single_barcodes(path, TEMPLATE, 2, pool, 1, use_first, 4)
            assert n == t == len(reads), (form, n)
            assert np.array_equal(got, e), form
    # the host-parser path (device scan switched off) must agree
    monkeypatch.setenv("SCG_DEVICE_SCAN", "0")
    got, n = sc.count_single_barcodes(paths["plain"], TEMPLATE, 2, pool, 1, True, 4)
    assert n == total and np.array_equal(got, exp)


def test_several_pipelines_share_one_file(sc, oracle, gpu, tmp_path, monkeypatch, plain_scan, bgzf_inflate):
    """$SCG_DEVICES lists the devices one call may use; an id may repeat.  Windows go round-robin over the plans and the
    per-device counters are summed at the end: the result must not depend on the list."""
    pool, reads = make_case(12, n=20000)
    exp, total = oracle.count_single(reads, TEMPLATE, 2, pool, 1, True)
    paths = write_forms(tmp_path, reads)
    monkeypatch.setenv("SCG_WINDOW_KB", "64")
    for devices in ("0", "0,0", "0,0,0,0", "all"):
        monkeypatch.setenv("SCG_DEVICES", devices)
        for form in ("plain", "bgzf"):
            got, n = sc.count_single_barcodes(paths[form], TEMPLATE, 2, pool, 1, True, 4)
            assert n == total and np.array_equal(got, exp), (devices, form)
    monkeypatch.setenv("SCG_DEVICES", "0,7")
    if sc.load().scg_device_count() < 8:
        from screencounter_amd import _lib
        with pytest.raises(_lib.ScgError) as e:
            sc.count_single_barcodes(paths["plain"], TEMPLATE, 2, pool, 1, True, 4)
        assert e.value.code == _lib.SCG_ERR_DEVICE and "out of range" in str(e.value)


def test_combo_and_dual_single_end_through_the_scan(sc, oracle, gpu, tmp_path, monkeypatch, bgzf_inflate):
    rng = random.Random(13)
    t = "ACGTAC" + "-" * 8 + "GGATCC" + "-" * 6 + "TGCATG"
    p0, p1 = gen.make_pool(rng, 20, 8, "ACGT"), gen.make_pool(rng, 15, 6, "ACGT")
    reads = gen.make_reads(rng, t, [p0, p1], 5000, 2, 0.03, 0.01, 0.02, 0.1, 30)
    path = str(tmp_path / "c.bgzf.gz")
    gen.write_bgzf(path, gen.fastq_text(reads), block=2000)
    monkeypatch.setenv("SCG_WINDOW_KB", "16")
    monkeypatch.setenv("SCG_DEVICES", "0,0")
    idx, freq, total = sc.count_combo_barcodes_single(path, t, 2, [p0, p1], 1, True, 2)
    eidx, efreq, etotal = oracle.count_combo(reads, t, 2, p0, p1, 1, True)
    assert total == etotal and np.array_equal(idx, eidx) and np.array_equal(freq, efreq)
    pools = [[p0[i % len(p0)] for i in range(25)], [p1[(3 * i) % len(p1)] for i in range(25)]]
    pools = [list(x) for x in zip(*sorted(set(zip(*pools))))]            # distinct combinations
    c, n = sc.count_dual_barcodes_single_end(path, t, pools, 2, 1, True, False, 2)
    ec, en = oracle.count_dual_single_end(reads, t, 2, pools, 1, True)
    assert n == en and np.array_equal(c, ec)


def test_unordinary_files_fall_back_to_the_sequential_reader(sc, oracle, gpu, tmp_path, monkeypatch, plain_scan):
    from screencounter_amd import _lib
    pool, reads = make_case(14, n=400)
    exp, total = oracle.count_single(reads, TEMPLATE, 2, pool, 1, True)
    monkeypatch.setenv("SCG_WINDOW_KB", "8")
    # multi-line sequences and qualities: legal for the reference (FastqReader.hpp:66-84), declined by the scan
    multi = b"".join(b"@r%d\n" % i + r[:len(r) // 2].encode() + b"\n" + r[len(r) // 2:].encode() + b"\n+\n" +
                     b"I" * (len(r) // 2) + b"\n" + b"I" * (len(r) - len(r) // 2) + b"\n" for i, r in enumerate(reads) if len(r) >= 2)
    exp, total = oracle.count_single([r for r in reads if len(r) >= 2], TEMPLATE, 2, pool, 1, True)
    p = str(tmp_path / "multi.fastq")
    open(p, "wb").write(multi)
    got, n = sc.count_single_barcodes(p, TEMPLATE, 2, pool, 1, True, 4)
    assert n == total and np.array_equal(got, exp)
    exp, total = oracle.count_single(reads, TEMPLATE, 2, pool, 1, True)
    # a '+' inside a sequence line ends the sequence there for the reference; here: the same counts via the fallback
    q = str(tmp_path / "crlf.fastq")
    open(q, "wb").write(gen.fastq_text(reads).replace(b"\n", b"\r\n"))     # '\r' stays a base on every line
    r2 = [r + "\r" for r in reads]
    e2, t2 = oracle.count_single(r2, TEMPLATE, 2, pool, 1, True)
    got, n = sc.count_single_barcodes(q, TEMPLATE, 2, pool, 1, True, 4)
    assert n == t2 and np.array_equal(got, e2)
    # malformed input deep inside a large file: the reference's message, with its line number
    bad = gen.fastq_text(reads[:300]) + b"@broken\nACGT\n+\nIII\n" + gen.fastq_text(reads[300:])
    b = str(tmp_path / "bad.fastq")
    open(b, "wb").write(bad)
    with pytest.raises(_lib.ScgError) as e:
        sc.count_single_barcodes(b, TEMPLATE, 2, pool, 1, True, 4)
    assert e.value.code == _lib.SCG_ERR_IO
    assert str(e.value) == "non-equal lengths for quality and sequence strings (starting line 1201)"
    # a stray blank line at the end
    s = str(tmp_path / "blank.fastq")
    open(s, "wb").write(gen.fastq_text(reads) + b"\n")
    with pytest.raises(_lib.ScgError) as e:
        sc.count_single_barcodes(s, TEMPLATE, 2, pool, 1, True, 4)
    assert "read name should start with '@'" in str(e.value)


def test_bgzf_members_on_the_device(sc, oracle, gpu, tmp_path, monkeypatch):
    """BGZF members inflated by the device: members of every size up to the format's 64 KiB (one lane each), stored and
    fixed-Huffman members, empty members in the middle, windows that end in the middle of a record (the partial record is
    carried to the next window on the device), a corrupt member (the host's zlib gets the last word: its error), a
    member whose text was changed but still inflates (only the CRC tells)."""
    import struct
    import zlib
    from screencounter_amd import _lib
    pool, reads = make_case(21, n=30000)
    exp, total = oracle.count_single(reads, TEMPLATE, 2, pool, 1, True)
    text = gen.fastq_text(reads, trailing_newline=False)

    def member(chunk, level=6, strategy=zlib.Z_DEFAULT_STRATEGY):
        c = zlib.compressobj(level, zlib.DEFLATED, -15, 8, strategy)
        payload = c.compress(chunk) + c.flush()
        head = b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", len(payload) + 25)
        return head + payload + struct.pack("<II", zlib.crc32(chunk) & 0xFFFFFFFF, len(chunk))

    rng = random.Random(22)
    parts, at = [], 0
    while at < len(text):
        n = rng.choice([1, 17, 300, 5000, 65280, 65280, 40000])
        chunk = text[at:at + n]
        at += n
        kind = rng.randrange(6)
        if kind == 0:
            parts.append(member(chunk, 0))                              # stored blocks
        elif kind == 1:
            parts.append(member(chunk, 6, zlib.Z_FIXED))                # fixed Huffman code
        elif kind == 2:
            parts.append(member(b"") + member(chunk, 9))                # an empty member in the middle
        else:
            parts.append(member(chunk, rng.choice([1, 6, 9])))
    parts.append(member(b""))                                           # bgzip's end-of-file marker
    p = str(tmp_path / "mixed.bgzf.gz")
    open(p, "wb").write(b"".join(parts))
    monkeypatch.setenv("SCG_DEVICE_INFLATE", "2")                       # no quiet fall-back to the host
    for kb in (None, 200, 1000):
        if kb:
            monkeypatch.setenv("SCG_WINDOW_KB", str(kb))
        got, n = sc.count_single_barcodes(p, TEMPLATE, 2, pool, 1, True, 4)
        assert n == total and np.array_equal(got, exp), kb
    monkeypatch.delenv("SCG_WINDOW_KB")
    monkeypatch.delenv("SCG_DEVICE_INFLATE")
    # a flipped bit in the middle of a member
    raw = bytearray(b"".join(parts))
    raw[len(raw) // 2] ^= 0x10
    bad = str(tmp_path / "flipped.bgzf.gz")
    open(bad, "wb").write(bytes(raw))
    with pytest.raises(_lib.ScgError) as e:
        sc.count_single_barcodes(bad, TEMPLATE, 2, pool, 1, True, 4)
    assert e.value.code == _lib.SCG_ERR_IO
    # text changed, CRC kept: still a valid DEFLATE stream of the right size
    k = len(parts) // 2
    chunk = bytearray(text[:5000])
    good = member(bytes(chunk), 6)
    i = chunk.index(b"\n") + 3
    chunk[i] = ord("A") if chunk[i] != ord("A") else ord("C")
    forged = member(bytes(chunk), 6)
    forged = forged[:-8] + good[-8:]                                     # the original's CRC and size
    f2 = str(tmp_path / "forged.bgzf.gz")
    open(f2, "wb").write(forged + member(b""))
    with pytest.raises(_lib.ScgError) as e:
        sc.count_single_barcodes(f2, TEMPLATE, 2, pool, 1, True, 4)
    assert e.value.code == _lib.SCG_ERR_IO and "incorrect data check" in str(e.value)


def test_multi_file_entries_equal_per_file_calls(sc, oracle, gpu, tmp_path, monkeypatch, plain_scan):
    rng = random.Random(15)
    pool = gen.make_pool(rng, 40, 12, "ACGT")
    files, exp, totals = [], [], []
    for i in range(7):
        reads = gen.make_reads(rng, TEMPLATE, [pool], 500 + 300 * i, 2, 0.03, 0.01, 0.02, 0.1, 40)
        p = str(tmp_path / f"f{i}.fastq")
        if i % 3 == 1:
            p += ".gz"
            gen.write_bgzf(p, gen.fastq_text(reads), block=4000)
        else:
            open(p, "wb").write(gen.fastq_text(reads))
        files.append(p)
        c, t = oracle.count_single(reads, TEMPLATE, 2, pool, 1, True)
        exp.append(c)
        totals.append(t)
    for devices in (None, [0], [0, 0, 0]):
        mat, tot = sc.count_single_barcodes_files(files, TEMPLATE, 2, pool, 1, True, 2, devices)
        assert tot == totals and np.array_equal(mat, np.stack(exp, axis=1)), devices
    # one file, no files
    mat, tot = sc.count_single_barcodes_files(files[:1], TEMPLATE, 2, pool, 1, True, 2)
    assert tot == totals[:1] and np.array_equal(mat[:, 0], exp[0])
    mat, tot = sc.count_single_barcodes_files([], TEMPLATE, 2, pool, 1, True, 2)
    assert tot == [] and mat.shape == (len(pool), 0)
    # the error of the lowest-numbered failing file is reported
    from screencounter_amd import _lib
    with pytest.raises(_lib.ScgError) as e:
        sc.count_single_barcodes_files(files[:2] + [str(tmp_path / "missing.fastq")] + files[2:], TEMPLATE, 2, pool, 1, True, 2, [0, 0])
    assert e.value.code == _lib.SCG_ERR_IO and "missing.fastq" in str(e.value)
    # combinations: per file exactly what the single-file entry returns
    t = "ACGTAC" + "-" * 8 + "GGATCC" + "-" * 6 + "TGCATG"
    p0, p1 = gen.make_pool(rng, 12, 8, "ACGT"), gen.make_pool(rng, 9, 6, "ACGT")
    cfiles = []
    for i in range(4):
        reads = gen.make_reads(rng, t, [p0, p1], 400 + 100 * i, 2, 0.03, 0.01, 0.02, 0.1, 30)
        p = str(tmp_path / f"c{i}.fastq")
        open(p, "wb").write(gen.fastq_text(reads))
        cfiles.append(p)
    per = sc.count_combo_barcodes_single_files(cfiles, t, 2, [p0, p1], 1, True, 2, [0, 0])
    for f, (idx, freq, total) in zip(cfiles, per):
        i1, f1, t1 = sc.count_combo_barcodes_single(f, t, 2, [p0, p1], 1, True, 2)
        assert total == t1 and np.array_equal(idx, i1) and np.array_equal(freq, f1)
    # pairs
    pairs1, pairs2 = files[:3], files[:3]
    mat, tot = sc.count_dual_barcodes_files(pairs1, TEMPLATE, False, 1, pool, pairs2, TEMPLATE, False, 1, pool, False, True, 2, [0, 0])
    for c, f in enumerate(pairs1):
        cc, tt = sc.count_dual_barcodes(f, TEMPLATE, False, 1, pool, f, TEMPLATE, False, 1, pool, False, True, False, 2)
        assert tot[c] == tt and np.array_equal(mat[:, c], cc)


def test_paired_files_through_the_scan(sc, oracle, gpu, tmp_path, monkeypatch, plain_scan, bgzf_inflate):
    """The windows of the two mates hold different numbers of records (names and read lengths differ), so the cursors that
    bring the two streams into step on the device are exercised; mate 1 is plain (scanned by the host threads or the
    device) or BGZF, mate 2 BGZF (members inflated on the device or by the host threads) or plain."""
    from screencounter_amd import _lib
    rng = random.Random(16)
    t1, t2 = "ACGTAC" + "-" * 10 + "TGCATG", "GGATCC" + "-" * 8 + "AAGCTT"
    u1, u2 = gen.make_pool(rng, 12, 10, "ACGT", min_dist=3), gen.make_pool(rng, 10, 8, "ACGT", min_dist=3)
    pairs = [(a, b) for a in u1 for b in u2]
    rng.shuffle(pairs)
    pairs = pairs[:60]
    pool1, pool2 = [a for a, _ in pairs], [b for _, b in pairs]
    r1, r2 = [], []
    for i in range(9000):
        a, b = rng.choice(pairs) if rng.random() < 0.85 else (rng.choice(u1), rng.choice(u2))
        x = gen.mutate(rng, gen.fill_template(t1, [a]), 0.02, 0.01, 0.02)
        y = gen.mutate(rng, gen.fill_template(t2, [b]), 0.02, 0.01, 0.02)
        r1.append(gen.rand_seq(rng, rng.randint(0, 60)) + x + gen.rand_seq(rng, rng.randint(0, 10)))     # mate 1 records are longer
        r2.append(gen.rand_seq(rng, rng.randint(0, 5)) + y)
    exp, total = oracle.count_dual(r1, r2, t1, False, 1, pool1, t2, False, 1, pool2, False, True)
    p1 = str(tmp_path / "m1.fastq")
    open(p1, "wb").write(gen.fastq_text(r1, name_prefix="a_rather_long_read_name_"))
    p2 = str(tmp_path / "m2.fastq.gz")
    gen.write_bgzf(p2, gen.fastq_text(r2, trailing_newline=False), block=2500)
    p2_plain = str(tmp_path / "m2.fastq")
    open(p2_plain, "wb").write(gen.fastq_text(r2, trailing_newline=False))
    p1_bgzf = str(tmp_path / "m1.fastq.gz")
    gen.write_bgzf(p1_bgzf, gen.fastq_text(r1, name_prefix="a_rather_long_read_name_"), block=7000)
    for kb in (None, 16, 64):
        if kb:
            monkeypatch.setenv("SCG_WINDOW_KB", str(kb))
        for first in (True, False):
            e, t = (exp, total) if first else oracle.count_dual(r1, r2, t1, False, 1, pool1, t2, False, 1, pool2, False, False)
            for one, second in ((p1, p2), (p1, p2_plain), (p1_bgzf, p2)):
                got, n = sc.count_dual_barcodes(one, t1, False, 1, pool1, second, t2, False, 1, pool2, False, first, False, 4)
                assert n == t == len(r1) and np.array_equal(got, e), (kb, first, one, second)
    # include.invalid = TRUE and randomized through the same pipeline
    d = oracle.count_dual_diag(r1, r2, t1, False, 1, pool1, t2, False, 1, pool2, True, True)
    counts, (idx, freq), tot, b1, b2 = sc.count_dual_barcodes(p1, t1, False, 1, pool1, p2, t2, False, 1, pool2, True, True, True, 4)
    assert tot == d["total"] and np.array_equal(counts, d["counts"]) and np.array_equal(idx, d["indices"]) and np.array_equal(freq, d["freq"])
    assert (b1, b2) == (d["barcode1_only"], d["barcode2_only"])
    # host-parser path agrees
    monkeypatch.setenv("SCG_DEVICE_SCAN", "0")
    got, n = sc.count_dual_barcodes(p1, t1, False, 1, pool1, p2, t2, False, 1, pool2, False, True, False, 4)
    assert n == total and np.array_equal(got, exp)
    monkeypatch.delenv("SCG_DEVICE_SCAN")
    # a corrupt member in a BGZF mate: zlib's error, whoever inflated first
    monkeypatch.setenv("SCG_DEVICE_INFLATE", "0" if bgzf_inflate == "host_inflate" else "1")
    raw = bytearray(open(p2, "rb").read())
    raw[len(raw) // 2] ^= 0x04
    pbad = str(tmp_path / "m2_bad.fastq.gz")
    open(pbad, "wb").write(bytes(raw))
    with pytest.raises(_lib.ScgError) as e:
        sc.count_dual_barcodes(p1, t1, False, 1, pool1, pbad, t2, False, 1, pool2, False, True, False, 4)
    assert e.value.code == _lib.SCG_ERR_IO
    # unequal numbers of reads: the reference's error
    p3 = str(tmp_path / "short.fastq")
    open(p3, "wb").write(gen.fastq_text(r2[:-7]))
    for a, b in ((p1, p3), (p3, p1)):
        with pytest.raises(_lib.ScgError) as e:
            sc.count_dual_barcodes(a, t1, False, 1, pool1, b, t2, False, 1, pool2, False, True, False, 4)
        assert e.value.code == _lib.SCG_ERR_IO and "different number of reads in paired FASTQ files" in str(e.value)
    # one mate with multi-line records: both files go through the host readers
    multi = b"".join(b"@r%d\n" % i + r[:len(r) // 2].encode() + b"\n" + r[len(r) // 2:].encode() + b"\n+\n" +
                     b"I" * (len(r) // 2) + b"\n" + b"I" * (len(r) - len(r) // 2) + b"\n" for i, r in enumerate(r2))
    p4 = str(tmp_path / "multi2.fastq")
    open(p4, "wb").write(multi)
    got, n = sc.count_dual_barcodes(p1, t1, False, 1, pool1, p4, t2, False, 1, pool2, False, True, False, 4)
    assert n == total and np.array_equal(got, exp)


def test_error_order_and_buffer_cache(sc, oracle, gpu, tmp_path, monkeypatch, plain_scan):
    """The library is compiled while the first window is already on its way to the GPU; the reference's error order must
    survive that: missing file, then the handler's argument errors, then whatever the file holds.  And the staging buffers
    kept between calls never change results."""
    from screencounter_amd import _lib
    pool, reads = make_case(17, n=300)
    good = str(tmp_path / "good.fastq")
    open(good, "wb").write(gen.fastq_text(reads))
    bad = str(tmp_path / "bad.fastq")
    open(bad, "wb").write(b"@r0\nACGT\n+\nIII\n")
    with pytest.raises(_lib.ScgError) as e:                       # file error first
        sc.count_single_barcodes(str(tmp_path / "none.fastq"), "ACXT----TGCA", 2, ["AAAA"], 0, True, 1)
    assert e.value.code == _lib.SCG_ERR_IO and "failed to open file" in str(e.value)
    with pytest.raises(_lib.ScgError) as e:                       # argument error before the file's content is looked at
        sc.count_single_barcodes(bad, "ACXT----TGCA", 2, ["AAAA"], 0, True, 1)
    assert e.value.code == _lib.SCG_ERR_INVALID and "unknown base 'X'" in str(e.value)
    with pytest.raises(_lib.ScgError) as e:
        sc.count_single_barcodes(bad, TEMPLATE, 2, pool, 1, True, 1)
    assert e.value.code == _lib.SCG_ERR_IO and "non-equal lengths" in str(e.value)
    exp, total = oracle.count_single(reads, TEMPLATE, 2, pool, 1, True)
    prepared = sc.prepare_pool(pool)
    for cache in ("1", "0", "1"):
        monkeypatch.setenv("SCG_BUFFER_CACHE", cache)
        for p in (pool, prepared):
            got, n = sc.count_single_barcodes(good, TEMPLATE, 2, p, 1, True, 2)
            assert n == total and np.array_equal(got, exp)
        sc.load().scg_release_buffers()
    got, n = sc.count_single_barcodes(good, TEMPLATE, 2, prepared, 1, True, 2)
    assert n == total and np.array_equal(got, exp)


def test_cached_slots_serve_every_pipeline(sc, oracle, gpu, tmp_path, monkeypatch):
    """Idle staging slots are kept between calls and shared by all pipelines, whose needs differ: the device-inflate
    pipeline pins only what the compressed bytes take, the others fill the pinned buffer up to the slot's text capacity.
    Alternating the pipelines over files larger than a slot must never hand a pipeline a slot it overruns (found by
    tools/gpu_ingest_fuzz.py as heap corruption)."""
    pool, reads = make_case(31, n=30000)
    exp, total = oracle.count_single(reads, TEMPLATE, 2, pool, 1, True)
    paths = write_forms(tmp_path, reads)
    for kb in (700, 100, 700, 16):
        monkeypatch.setenv("SCG_WINDOW_KB", str(kb))
        for form, env in (("bgzf", {"SCG_DEVICE_INFLATE": "2"}), ("plain", {}), ("bgzf", {"SCG_DEVICE_INFLATE": "0"}), ("plain", {"SCG_HOST_SCAN": "0"})):
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            got, n = sc.count_single_barcodes(paths[form], TEMPLATE, 2, pool, 1, True, 4)
            assert n == total and np.array_equal(got, exp), (kb, form, env)
            for k in env:
                monkeypatch.delenv(k)


def test_ordinary_gzip_decoded_by_all_host_threads(sc, oracle, gpu, tmp_path, monkeypatch):
    """A .fastq.gz that is not BGZF goes through the parallel decoder (csrc/scg_pgzip.h: chunks decoded with an unknown
    window, stitched, CRC-checked); SCG_PGZIP_CHUNK_KB makes these small files take it.  Single-end and paired, one and
    several members, tiny windows; and a file the decoder hands back (corrupt: the reference's error, through zlib)."""
    import zlib
    from screencounter_amd import _lib
    pool, reads = make_case(21, n=12000)
    exp, total = oracle.count_single(reads, TEMPLATE, 2, pool, 1, True)
    text = gen.fastq_text(reads)

    def gz(data, level=6):
        c = zlib.compressobj(level, zlib.DEFLATED, 31)
        return c.compress(data) + c.flush()

    one = str(tmp_path / "one.fastq.gz")
    open(one, "wb").write(gz(text, 4))
    several = str(tmp_path / "several.fastq.gz")
    third = (len(text) // 3)
    cut1, cut2 = text.index(b"\n@", third) + 1, text.index(b"\n@", 2 * third) + 1
    open(several, "wb").write(gz(text[:cut1], 6) + gz(text[cut1:cut2], 1) + gz(text[cut2:], 9))
    monkeypatch.setenv("SCG_PGZIP_CHUNK_KB", "32")
    for kb in (None, 16):
        if kb:
            monkeypatch.setenv("SCG_WINDOW_KB", str(kb))
        for path in (one, several):
            got, n = sc.count_single_barcodes(path, TEMPLATE, 2, pool, 1, True, 4)
            assert n == total == len(reads) and np.array_equal(got, exp), (kb, path)
    monkeypatch.delenv("SCG_WINDOW_KB")
    # the same file with the decoder switched off must agree (and is what a declined file falls back to)
    monkeypatch.setenv("SCG_PGZIP", "0")
    got, n = sc.count_single_barcodes(one, TEMPLATE, 2, pool, 1, True, 4)
    assert n == total and np.array_equal(got, exp)
    monkeypatch.delenv("SCG_PGZIP")
    # both mates gzip: two decoders share the host threads
    rng = random.Random(22)
    t1, t2 = "ACGTAC" + "-" * 10 + "TGCATG", "GGATCC" + "-" * 8 + "AAGCTT"
    u1, u2 = gen.make_pool(rng, 12, 10, "ACGT", min_dist=3), gen.make_pool(rng, 10, 8, "ACGT", min_dist=3)
    pairs = [(a, b) for a in u1 for b in u2][:50]
    pool1, pool2 = [a for a, _ in pairs], [b for _, b in pairs]
    r1, r2 = [], []
    for i in range(8000):
        a, b = rng.choice(pairs)
        r1.append(gen.rand_seq(rng, rng.randint(0, 40)) + gen.mutate(rng, gen.fill_template(t1, [a]), 0.02, 0.01, 0.02))
        r2.append(gen.mutate(rng, gen.fill_template(t2, [b]), 0.02, 0.01, 0.02) + gen.rand_seq(rng, rng.randint(0, 9)))
    e, t = oracle.count_dual(r1, r2, t1, False, 1, pool1, t2, False, 1, pool2, False, True)
    p1, p2 = str(tmp_path / "m1.fastq.gz"), str(tmp_path / "m2.fastq.gz")
    open(p1, "wb").write(gz(gen.fastq_text(r1), 6))
    open(p2, "wb").write(gz(gen.fastq_text(r2), 6))
    got, n = sc.count_dual_barcodes(p1, t1, False, 1, pool1, p2, t2, False, 1, pool2, False, True, False, 4)
    assert n == t == len(r1) and np.array_equal(got, e)
    # a corrupt stream: handed back by the decoder, reported the way the sequential reader (zlib) reports it
    raw = bytearray(open(one, "rb").read())
    raw[len(raw) // 2] ^= 0x10
    bad = str(tmp_path / "bad.fastq.gz")
    open(bad, "wb").write(bytes(raw))
    with pytest.raises(_lib.ScgError) as e1:
        sc.count_single_barcodes(bad, TEMPLATE, 2, pool, 1, True, 4)
    monkeypatch.setenv("SCG_PGZIP", "0")
    with pytest.raises(_lib.ScgError) as e2:
        sc.count_single_barcodes(bad, TEMPLATE, 2, pool, 1, True, 4)
    assert e1.value.code == e2.value.code == _lib.SCG_ERR_IO and str(e1.value) == str(e2.value)


def test_paired_plain_files_over_several_devices(sc, oracle, gpu, tmp_path, monkeypatch):
    """More than one device in the list (an id may repeat): paired plain files go through PairedRounds -- the host threads
    scan both mates, every round of pairs the two cursors share is gathered and counted by one device, round-robin, and
    the per-device counters are summed.  The mates' windows hold different numbers of records (names and read lengths
    differ), so rounds begin and end inside windows all the time; tiny windows make hundreds of them."""
    from screencounter_amd import _lib
    rng = random.Random(31)
    t1, t2 = "ACGTAC" + "-" * 10 + "TGCATG", "GGATCC" + "-" * 8 + "AAGCTT"
    u1, u2 = gen.make_pool(rng, 12, 10, "ACGT", min_dist=3), gen.make_pool(rng, 10, 8, "ACGT", min_dist=3)
    pairs = [(a, b) for a in u1 for b in u2]
    rng.shuffle(pairs)
    pairs = pairs[:60]
    pool1, pool2 = [a for a, _ in pairs], [b for _, b in pairs]
    r1, r2 = [], []
    for i in range(12000):
        a, b = rng.choice(pairs) if rng.random() < 0.85 else (rng.choice(u1), rng.choice(u2))
        x = gen.mutate(rng, gen.fill_template(t1, [a]), 0.02, 0.01, 0.02)
        y = gen.mutate(rng, gen.fill_template(t2, [b]), 0.02, 0.01, 0.02)
        r1.append(gen.rand_seq(rng, rng.randint(0, 60)) + x + gen.rand_seq(rng, rng.randint(0, 10)))
        r2.append(gen.rand_seq(rng, rng.randint(0, 5)) + y)
    p1, p2 = str(tmp_path / "m1.fastq"), str(tmp_path / "m2.fastq")
    open(p1, "wb").write(gen.fastq_text(r1, name_prefix="a_rather_long_read_name_"))
    open(p2, "wb").write(gen.fastq_text(r2, trailing_newline=False))
    exp, total = oracle.count_dual(r1, r2, t1, False, 1, pool1, t2, False, 1, pool2, False, True)
    expb, _ = oracle.count_dual(r1, r2, t1, False, 1, pool1, t2, False, 1, pool2, True, False)
    d = oracle.count_dual_diag(r1, r2, t1, False, 1, pool1, t2, False, 1, pool2, True, True)
    for kb in (None, 16):
        if kb:
            monkeypatch.setenv("SCG_WINDOW_KB", str(kb))
        for devices in ("0,0", "0,0,0"):
            monkeypatch.setenv("SCG_DEVICES", devices)
            got, n = sc.count_dual_barcodes(p1, t1, False, 1, pool1, p2, t2, False, 1, pool2, False, True, False, 4)
            assert n == total == len(r1) and np.array_equal(got, exp), (kb, devices)
            got, n = sc.count_dual_barcodes(p1, t1, False, 1, pool1, p2, t2, False, 1, pool2, True, False, False, 4)
            assert n == total and np.array_equal(got, expb), (kb, devices, "best, randomized")
        # include.invalid = TRUE: the diagnostics counters are summed over the devices as well
        counts, (idx, freq), tot, b1, b2 = sc.count_dual_barcodes(p1, t1, False, 1, pool1, p2, t2, False, 1, pool2, True, True, True, 4)
        assert tot == d["total"] and np.array_equal(counts, d["counts"]) and np.array_equal(idx, d["indices"]) and np.array_equal(freq, d["freq"])
        assert (b1, b2) == (d["barcode1_only"], d["barcode2_only"])
    # unequal numbers of reads: the reference's error, whichever file is the longer one
    short = str(tmp_path / "short.fastq")
    open(short, "wb").write(gen.fastq_text(r2[:-7]))
    for a, b, ta, tb, pa, pb in ((p1, short, t1, t2, pool1, pool2), (short, p1, t2, t1, pool2, pool1)):
        with pytest.raises(_lib.ScgError) as e:
            sc.count_dual_barcodes(a, ta, False, 1, pa, b, tb, False, 1, pb, False, True, False, 4)
        assert e.value.code == _lib.SCG_ERR_IO and "different number of reads" in str(e.value)


def test_ordinary_gzip_decoded_by_the_device(sc, oracle, gpu, tmp_path, monkeypatch):
    """Ordinary gzip files (what `gzip` writes; several members too, when they are large) go to the device: chunks decoded into symbols by one wavefront each, the
    chain checked on the host, symbols turned into text that stays in HBM (csrc/scg_dgzip.cpp).  SCG_DEVICE_GUNZIP=2 makes
    a hand-back an error, so these well-formed files cannot pass on the host decoders; files of a kind the device decoder
    does not take (many small members, a header CRC, trailing bytes, a flipped bit) must come out as the host readers have
    them -- same counts, or the same error."""
    import gzip as gz_mod
    import io
    import zlib
    rng = random.Random(4242)
    template = "ACGTACGA" + "-" * 12 + "TGCATGCA"
    pool = gen.make_pool(rng, 60, 12, "ACGT")
    reads = gen.make_reads(rng, template, [pool], 30000, 2, 0.03, 0.01, 0.02, 0.1, 60)
    text = gen.fastq_text(reads)
    exp = oracle.count_single(reads, template, 2, pool, 1, True)

    def gz(data, level):
        c = zlib.compressobj(level, zlib.DEFLATED, 31)
        return c.compress(data) + c.flush()

    def count(path):
        return sc.count_single_barcodes(path, template, 2, pool, 1, True, 4)

    monkeypatch.setenv("SCG_PGZIP_CHUNK_KB", "64")                 # (these files are small: let the chunked decoders have them)
    p = str(tmp_path / "one.fastq.gz")
    turn = 0
    for level in (1, 4, 6, 9):
        for chunk_kb, group_kb in ((4, None), (16, 256), (128, None), (32, 128)):    # (group_kb: the stream decoded in several groups of chunks)
            for window_kb in (None, 100):
                open(p, "wb").write(gz(text, level))
                # the tails' scan over groups of 32 chunks (default), of 1, 3 or 7: the 250 chunks of 4 KB make up to 250 groups
                tail_group = (None, 1, 3, 7)[turn % 4]
                turn += 1
                if tail_group:
                    monkeypatch.setenv("SCG_DGZIP_TAIL_GROUP", str(tail_group))
                else:
                    monkeypatch.delenv("SCG_DGZIP_TAIL_GROUP", raising=False)
                monkeypatch.setenv("SCG_DGZIP_CHUNK_KB", str(chunk_kb))
                if group_kb:
                    monkeypatch.setenv("SCG_DGZIP_GROUP_KB", str(group_kb))
                else:
                    monkeypatch.delenv("SCG_DGZIP_GROUP_KB", raising=False)
                monkeypatch.setenv("SCG_DEVICE_GUNZIP", "2")
                if window_kb:
                    monkeypatch.setenv("SCG_WINDOW_KB", str(window_kb))
                else:
                    monkeypatch.delenv("SCG_WINDOW_KB", raising=False)
                try:
                    c, t = count(p)
                except sc.ScgError as e:
                    raise AssertionError((level, chunk_kb, group_kb, window_kb, tail_group, str(e)))
                assert t == exp[1] and np.array_equal(c, exp[0]), (level, chunk_kb, group_kb, window_kb, tail_group)
    monkeypatch.delenv("SCG_WINDOW_KB", raising=False)
    monkeypatch.setenv("SCG_DGZIP_TAIL_GROUP", "5")
    monkeypatch.delenv("SCG_DGZIP_GROUP_KB", raising=False)
    monkeypatch.setenv("SCG_DGZIP_CHUNK_KB", "16")
    # a named file (FNAME) is taken; the final record without its newline too
    b = io.BytesIO()
    with gz_mod.GzipFile(filename="reads.fastq", mode="wb", fileobj=b, compresslevel=5) as f:
        f.write(text[:-1])
    open(p, "wb").write(b.getvalue())
    c, t = count(p)
    assert t == exp[1] and np.array_equal(c, exp[0])
    # handed back: the strict switch says so, the default quietly takes the host decoders
    half = len(text) // 2
    cut = text.rfind(b"\n@", 0, half) + 1
    raw = gz(text, 6)
    hdr = bytearray(raw[:10]); hdr[3] = 2
    with_hcrc = bytes(hdr) + (zlib.crc32(bytes(hdr)) & 0xFFFF).to_bytes(2, "little") + raw[10:]
    # several members (`cat a.gz b.gz`, or a writer that flushes now and then): each one's chain, CRC-32 and ISIZE checked on
    # its own, the next one's chunk grid starting behind its header; a member may end anywhere in a group of chunks
    third = text.rfind(b"\n@", 0, len(text) // 3) + 1
    two_thirds = text.rfind(b"\n@", 0, 2 * len(text) // 3) + 1
    near_end = text.rfind(b"\n@", 0, len(text) - 3000) + 1
    monkeypatch.setenv("SCG_DEVICE_GUNZIP", "2")
    for name, data, chunk_kb in (("two members", gz(text[:cut], 6) + gz(text[cut:], 6), 8),         # (32 chunks and more: taken)
                                 ("three members, levels", gz(text[:third], 1) + gz(text[third:two_thirds], 9) + gz(text[two_thirds:], 4), 4),
                                 ("an empty member last", gz(text, 6) + gz(b"", 6), 16),
                                 ("a small member last", gz(text[:near_end], 6) + gz(text[near_end:], 6), 4)):
        open(p, "wb").write(data)
        monkeypatch.setenv("SCG_DGZIP_CHUNK_KB", str(chunk_kb))
        for group_kb in (None, 64, 400):
            if group_kb:
                monkeypatch.setenv("SCG_DGZIP_GROUP_KB", str(group_kb))
            else:
                monkeypatch.delenv("SCG_DGZIP_GROUP_KB", raising=False)
            try:
                c, t = count(p)
            except sc.ScgError as e:
                raise AssertionError((name, group_kb, str(e)))
            assert t == exp[1] and np.array_equal(c, exp[0]), (name, group_kb)
    monkeypatch.delenv("SCG_DGZIP_GROUP_KB", raising=False)
    monkeypatch.setenv("SCG_DGZIP_CHUNK_KB", "16")
    many = b"".join(gz(text[i:i + 50000], 6) for i in range(0, len(text), 50000))      # (members too small to give each a launch)
    for name, data in (("many small members", many), ("header crc", with_hcrc), ("trailing bytes", raw + b"\0\0\0\0")):
        open(p, "wb").write(data)
        monkeypatch.setenv("SCG_DEVICE_GUNZIP", "2")
        with pytest.raises(sc.ScgError):
            count(p)
        monkeypatch.setenv("SCG_DEVICE_GUNZIP", "1")
        c, t = count(p)
        assert t == exp[1] and np.array_equal(c, exp[0]), name
    # flipped bits: whatever the sequential reader makes of the file (SCG_DEVICE_SCAN=0) is what every path must give
    for k in range(12):
        data = bytearray(raw)
        data[rng.randrange(12, len(data) - 8)] ^= 1 << rng.randrange(8)
        open(p, "wb").write(bytes(data))
        monkeypatch.setenv("SCG_DEVICE_SCAN", "0")
        try:
            want = ("ok",) + tuple(x if isinstance(x, int) else x.tobytes() for x in count(p)[::-1])
        except sc.ScgError as e:
            want = ("error", e.code, str(e))
        monkeypatch.delenv("SCG_DEVICE_SCAN")
        try:
            got = ("ok",) + tuple(x if isinstance(x, int) else x.tobytes() for x in count(p)[::-1])
        except sc.ScgError as e:
            got = ("error", e.code, str(e))
        assert got == want, k


def test_paired_ordinary_gzip_mates_decoded_by_the_device(sc, oracle, gpu, tmp_path, monkeypatch):
    """R1.fastq.gz / R2.fastq.gz as `gzip` writes them: each mate decoded on the device (one after the other), both texts
    in HBM, windows paired by cursors as for any other paired input.  Also one gzip mate next to a plain one."""
    import zlib
    rng = random.Random(4343)
    case = gen.random_dual_case(rng, hazard_free=True, sizes=(20000,), max_mm=1)
    exp = oracle.count_dual(case["reads1"], case["reads2"], case["template1"], case["reverse1"], case["mismatches1"], case["pool1"],
                            case["template2"], case["reverse2"], case["mismatches2"], case["pool2"], case["randomized"], case["use_first"])

    def gz(data, level):
        c = zlib.compressobj(level, zlib.DEFLATED, 31)
        return c.compress(data) + c.flush()

    t1, t2 = gen.fastq_text(case["reads1"]), gen.fastq_text(case["reads2"], name_prefix="a_longer_name_for_the_second_mate_")
    p1, p2, plain2 = str(tmp_path / "r1.fastq.gz"), str(tmp_path / "r2.fastq.gz"), str(tmp_path / "r2.fastq")
    open(p1, "wb").write(gz(t1, 6))
    open(p2, "wb").write(gz(t2, 1))
    open(plain2, "wb").write(t2)
    monkeypatch.setenv("SCG_PGZIP_CHUNK_KB", "64")
    monkeypatch.setenv("SCG_DGZIP_CHUNK_KB", "16")
    monkeypatch.setenv("SCG_DEVICE_GUNZIP", "2")
    for second in (p2, plain2):
        for window_kb in (None, 100):
            if window_kb:
                monkeypatch.setenv("SCG_WINDOW_KB", str(window_kb))
            else:
                monkeypatch.delenv("SCG_WINDOW_KB", raising=False)
            c, t = sc.count_dual_barcodes(p1, case["template1"], case["reverse1"], case["mismatches1"], case["pool1"],
                                          second, case["template2"], case["reverse2"], case["mismatches2"], case["pool2"],
                                          case["randomized"], case["use_first"], False, 4)
            assert t == exp[1] and np.array_equal(c, exp[0]), (second, window_kb)
