"""The parallel decoder of ordinary gzip files (csrc/scg_pgzip.h / .cpp: chunks decoded speculatively with an unknown
window, stitched in order, CRC-checked per member) against zlib's gzread -- the reader the reference uses
(inst/include/byteme/GzipFileReader.hpp:39-51).  Host code only: run natively and under AddressSanitizer + UBSan.

The contract: whatever the decoder ACCEPTS must be byte-identical to zlib's text; anything else it has to hand back
("declined"), and the sequential path -- zlib itself -- then decides.  A "DIFF" is a bug."""
import gzip
import io
import os
import random
import subprocess
import zlib

import pytest

from tests import gen
from tests.test_ingest_cpu import text_windows, random_reads, strict_records

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = [os.path.join(ROOT, "tests", "pgzip_harness.cpp"), os.path.join(ROOT, "screencounter_amd", "csrc", "scg_pgzip.cpp")]


@pytest.fixture(scope="module", params=["native", "asan"])
def harness(request, tmp_path_factory):
    out = str(tmp_path_factory.mktemp("pgzip") / f"harness_{request.param}")
    flags = ["-O2"] if request.param == "native" else ["-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer"]
    subprocess.check_call(["g++", "-std=c++17", *flags, "-o", out, *SRC, "-lz", "-ldl", "-lpthread"])
    return out


def run(harness, path, threads=4, chunk_kb=64, cap=None, extra_env=None):
    env = dict(os.environ, SCG_PGZIP_CHUNK_KB=str(chunk_kb))
    env.update(extra_env or {})
    cmd = [harness, path, str(threads)] + ([str(cap)] if cap else [])
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout, r.stderr[-2000:])
    return r.stdout.strip().splitlines()[-1]


def fastq(n, seed):
    rng = random.Random(seed)
    out = []
    for i in range(n):
        L = rng.randint(50, 150)
        s = "".join(rng.choice("ACGT") for _ in range(L))
        q = "".join(rng.choice("FFFFF:,#") for _ in range(L))
        out.append(f"@read{i} lane:{rng.randint(1, 8)}\n{s}\n+\n{q}\n")
    return "".join(out).encode()


def gz(data, level=6):
    c = zlib.compressobj(level, zlib.DEFLATED, 31)
    return c.compress(data) + c.flush()


TEXT = fastq(20000, 1)


@pytest.mark.parametrize("level", [1, 4, 6, 9])
@pytest.mark.parametrize("threads,chunk_kb", [(1, 64), (3, 32), (8, 200)])
def test_same_text_as_zlib(harness, tmp_path, level, threads, chunk_kb):
    p = str(tmp_path / "x.gz")
    open(p, "wb").write(gz(TEXT, level))
    assert run(harness, p, threads, chunk_kb).startswith(f"same {len(TEXT)} ")


def test_small_reads_of_the_consumer(harness, tmp_path):
    p = str(tmp_path / "x.gz")
    open(p, "wb").write(gz(TEXT, 6))
    for cap in (1, 4097, 1 << 20):                   # the consumer's buffer may be smaller than a piece, a chunk, a window
        if cap == 1 and "asan" in harness:
            continue
        assert run(harness, p, 4, 16, cap=cap if cap > 1 else 333).startswith(f"same {len(TEXT)} ")


def test_members_concatenated(harness, tmp_path):
    third = len(TEXT) // 3
    data = gz(TEXT[:third], 6) + gz(b"", 6) + gz(TEXT[third:2 * third], 1) + gz(b"", 9) + gz(TEXT[2 * third:], 9)
    p = str(tmp_path / "multi.gz")
    open(p, "wb").write(data)
    for chunk_kb in (16, 64, 4096):
        assert run(harness, p, 4, chunk_kb).startswith(f"same {len(TEXT)} ")
    # members of a few blocks each: member ends and starts fall next to chunk boundaries all the time
    parts = [TEXT[i:i + 200011] for i in range(0, len(TEXT), 200011)]
    open(p, "wb").write(b"".join(gz(x, 6) for x in parts))
    assert run(harness, p, 4, 16).startswith(f"same {len(TEXT)} ")
    # single-block members hold nothing a chunk could start at (the block finder looks for non-final blocks): handed back, or
    # decoded by the stitching pass -- never wrong
    parts = [TEXT[i:i + 20011] for i in range(0, len(TEXT), 20011)]
    open(p, "wb").write(b"".join(gz(x, 6) for x in parts))
    assert run(harness, p, 4, 16).split()[:2] in (["same", str(len(TEXT))], ["declined", "ok"])


def test_header_fields(harness, tmp_path):
    b = io.BytesIO()
    with gzip.GzipFile(filename="some_name.fastq", mode="wb", fileobj=b, compresslevel=5) as f:
        f.write(TEXT)
    p = str(tmp_path / "named.gz")
    open(p, "wb").write(b.getvalue())
    assert run(harness, p).startswith(f"same {len(TEXT)} ")
    # FEXTRA + FCOMMENT by hand
    raw = gz(TEXT, 6)
    hdr = bytearray(raw[:10]); hdr[3] = 4 | 16
    data = bytes(hdr) + (5).to_bytes(2, "little") + b"extra" + b"a comment\0" + raw[10:]
    open(p, "wb").write(data)
    assert run(harness, p).startswith(f"same {len(TEXT)} ")
    # a header CRC (FHCRC) is for zlib to check: declined
    hdr = bytearray(raw[:10]); hdr[3] = 2
    crc16 = zlib.crc32(bytes(hdr)) & 0xFFFF
    open(p, "wb").write(bytes(hdr) + crc16.to_bytes(2, "little") + raw[10:])
    assert run(harness, p).startswith("declined ok")


def test_tiny_and_empty(harness, tmp_path):
    p = str(tmp_path / "t.gz")
    open(p, "wb").write(gz(b"@r\nACGT\n+\nIIII\n"))       # one fixed-Huffman block
    assert run(harness, p) .startswith("same 15 ")
    open(p, "wb").write(gz(b""))
    assert run(harness, p).startswith("same 0 ")


def test_stored_and_incompressible(harness, tmp_path):
    p = str(tmp_path / "s.gz")
    open(p, "wb").write(gz(TEXT[:1_000_000], 0))            # stored blocks only: nothing to find; decoded by the stitching pass, or declined
    assert run(harness, p).split()[0] in ("same", "declined")
    open(p, "wb").write(gz(os.urandom(700_000), 6))
    assert run(harness, p).split()[0] in ("same", "declined")
    # a chunk that inflates beyond its buffer ends the attempt (zlib is fine with it)
    open(p, "wb").write(gz(b"\0" * 30_000_000, 6))
    assert run(harness, p).startswith("declined ok 30000000")


def test_what_zlib_rejects_is_never_accepted(harness, tmp_path):
    base = gz(TEXT, 6)
    p = str(tmp_path / "bad.gz")
    rng = random.Random(11)
    outcomes = set()
    for i in range(40 if "asan" not in harness else 15):
        b = bytearray(base)
        for _ in range(rng.choice([1, 1, 2, 5])):
            pos = rng.randrange(len(b))
            b[pos] ^= 1 << rng.randrange(8)
        open(p, "wb").write(bytes(b))
        line = run(harness, p, 4, rng.choice([16, 64]))
        assert not line.startswith("DIFF"), (i, line)
        outcomes.add(line.split()[0])
    assert "declined" in outcomes
    # truncated, and trailing bytes that are no member
    open(p, "wb").write(base[:-777])
    assert run(harness, p).startswith("declined")
    open(p, "wb").write(base + b"trailing bytes")
    assert run(harness, p).startswith("declined")


def test_through_the_text_source(sc, tmp_path, monkeypatch):
    """The ingestion's TextSource takes such a file through the parallel decoder (kind "gzip-parallel"), cuts the text
    into windows of whole records, and reads a file the decoder hands back through one inflate stream."""
    rng = random.Random(3)
    reads = random_reads(rng, 20000)
    text = gen.fastq_text(reads)
    p = str(tmp_path / "x.fastq.gz")
    open(p, "wb").write(gz(text, 4))
    monkeypatch.setenv("SCG_PGZIP_CHUNK_KB", "64")
    data, cuts, kind = text_windows(sc, p, 1 << 20, threads=4)
    assert kind == "gzip-parallel" and data == text and len(cuts) > 3
    got = []
    for a, b in zip(cuts, cuts[1:]):
        got += strict_records(data[a:b])
    assert got == [r.encode() for r in reads]
    monkeypatch.setenv("SCG_PGZIP", "0")
    data, cuts, kind = text_windows(sc, p, 1 << 20, threads=4)
    assert kind == "gzip" and data == text
    monkeypatch.delenv("SCG_PGZIP")
    # handed back (trailing garbage): the same text through the fall-back, as gzread gives it
    open(p, "wb").write(gz(text, 4) + b"\0\0\0\0")
    data, cuts, kind = text_windows(sc, p, 1 << 20, threads=4)
    assert kind == "gzip" and data == text
