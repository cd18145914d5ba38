"""Random workload generators shared by the oracle fuzzers and the parity tests (test infrastructure)."""
from __future__ import annotations

import random

BASES = "ACGT"
COMP = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N", "a": "t", "c": "g", "g": "c", "t": "a", "n": "n"}
IUPAC = "RYSWKMBDHVN"


def rc(s: str) -> str:
    return "".join(COMP.get(c, "N") for c in reversed(s))


def rand_seq(rng: random.Random, n: int, alphabet: str = BASES) -> str:
    return "".join(rng.choice(alphabet) for _ in range(n))


def mutate(rng: random.Random, s: str, p_sub: float, p_n: float, p_lower: float) -> str:
    out = []
    for c in s:
        if rng.random() < p_sub:
            c = rng.choice([b for b in BASES if b != c.upper()])
        if rng.random() < p_n:
            c = rng.choice("NnRX.")  # any non-ACGT byte is "other" to the scanner
        if rng.random() < p_lower:
            c = c.lower()
        out.append(c)
    return "".join(out)


def make_pool(rng: random.Random, n: int, length: int, alphabet: str, min_dist: int = 1, iupac_rate: float = 0.0):
    """distinct sequences with pairwise Hamming distance >= min_dist (on the concrete bases)"""
    pool: list[str] = []
    tries = 0
    while len(pool) < n and tries < 20000:
        tries += 1
        s = rand_seq(rng, length, alphabet)
        if all(sum(a != b for a, b in zip(s, t)) >= min_dist for t in pool):
            pool.append(s)
    if iupac_rate > 0:
        pool = ["".join(rng.choice(IUPAC) if rng.random() < iupac_rate else c for c in s) for s in pool]
    return pool


def make_template(rng: random.Random, nvar: int, var_lens: list[int], flank_lo: int, flank_hi: int) -> str:
    parts = [rand_seq(rng, rng.randint(flank_lo, flank_hi))]
    for v in range(nvar):
        parts.append("-" * var_lens[v])
        lo = max(flank_lo, 1) if v < nvar - 1 else flank_lo  # keep regions separate
        parts.append(rand_seq(rng, rng.randint(lo, flank_hi)))
    t = "".join(parts)
    if rng.random() < 0.2:
        t = t.lower()
    return t


def fill_template(template: str, inserts: list[str]) -> str:
    out = []
    it = iter(inserts)
    i = 0
    while i < len(template):
        if template[i] == "-":
            j = i
            while j < len(template) and template[j] == "-":
                j += 1
            out.append(next(it))
            i = j
        else:
            out.append(template[i].upper())
            i += 1
    return "".join(out)


def concrete(rng: random.Random, s: str) -> str:
    """pick one concrete expansion of an IUPAC library string"""
    table = {"R": "AG", "Y": "CT", "S": "CG", "W": "AT", "K": "GT", "M": "AC", "B": "CGT", "D": "AGT", "H": "ACT", "V": "ACG", "N": "ACGT"}
    return "".join(rng.choice(table[c]) if c in table else c for c in s.upper())


def make_reads(rng, template, pools, n, strand, p_sub, p_n, p_lower, p_junk, pad_hi, valid_pairs=None):
    reads = []
    for _ in range(n):
        if rng.random() < p_junk:
            reads.append(rand_seq(rng, rng.randint(0, len(template) + pad_hi)))
            continue
        if valid_pairs is not None:
            raise AssertionError
        ins = [concrete(rng, rng.choice(p)) for p in pools]
        core = fill_template(template, ins)
        core = mutate(rng, core, p_sub, p_n, p_lower)
        read = rand_seq(rng, rng.randint(0, pad_hi)) + core + rand_seq(rng, rng.randint(0, pad_hi))
        if rng.random() < 0.1:  # two constructs in one read
            ins2 = [concrete(rng, rng.choice(p)) for p in pools]
            read += rand_seq(rng, rng.randint(0, 3)) + mutate(rng, fill_template(template, ins2), p_sub, p_n, p_lower)
        if strand == 1 or (strand == 2 and rng.random() < 0.5):
            read = rc(read)
        reads.append(read)
    return reads




# ---------------------------------------------------------------------------------------------
# Whole random cases (inputs only) for the three entry points and the matcher.
# ---------------------------------------------------------------------------------------------
def random_single_case(rng: random.Random, max_vlen: int = 33, sizes=(1, 30, 200), min_vlen: int = 0) -> dict:
    vlen = rng.choice([v for v in (3, 4, 6, 8, 10, 20, 33, 40, 57, 64, 65, 100, 128, 129, 200, 240) if min_vlen <= v <= max_vlen])
    alphabet = rng.choice(["AC", "ACG", BASES, BASES])
    npool = rng.choice([1, 2, 5, 20, 100])
    pool = make_pool(rng, npool, vlen, alphabet, min_dist=1, iupac_rate=rng.choice([0, 0, 0.05 if vlen <= 64 else 0.01]))
    template = make_template(rng, 1, [vlen], rng.choice([0, 1, 3]), rng.choice([4, 8, 12, 40]) if vlen <= 128 else 8)    # (templates: at most 256)
    strand = rng.choice([0, 1, 2])
    mm = rng.choice([0, 1, 1, 2, 3])
    first = rng.random() < 0.5
    reads = make_reads(rng, template, [pool], rng.choice(sizes), strand, rng.choice([0, 0.02, 0.08]), rng.choice([0, 0.01, 0.05]),
                       rng.choice([0, 0.3]), 0.1, rng.choice([0, 5, 30]))
    return dict(kind="single", template=template, strand=strand, pool=pool, mismatches=mm, use_first=first, reads=reads)


def random_combo_case(rng: random.Random, sizes=(1, 30, 200), wide: bool = False) -> dict:
    v0, v1 = rng.choice([3, 5, 8, 14]), rng.choice([3, 6, 14])
    if wide == "big":  # a pool of 65..200 bases (big keys)
        v0, v1 = rng.choice([(65, 6), (8, 100), (100, 110), (70, 40), (200, 20)])
    elif wide:        # a pool of 33..64 bases (wide keys), the other short or long
        v0, v1 = rng.choice([(33, 6), (8, 40), (64, 64), (36, 33)])
    alphabet = rng.choice(["AC", BASES, BASES])
    p0 = make_pool(rng, rng.choice([1, 4, 30]), v0, alphabet, iupac_rate=rng.choice([0, 0, 0.05 if v0 <= 64 else 0.01]))
    p1 = make_pool(rng, rng.choice([1, 4, 30]), v1, alphabet, iupac_rate=rng.choice([0, 0, 0.05 if v1 <= 64 else 0.01]))
    template = make_template(rng, 2, [v0, v1], rng.choice([0, 1, 3]), rng.choice([4, 8, 12]))
    strand = rng.choice([0, 1, 2])
    mm = rng.choice([0, 1, 2, 3])
    first = rng.random() < 0.5
    reads = make_reads(rng, template, [p0, p1], rng.choice(sizes), strand, rng.choice([0, 0.03, 0.08]), rng.choice([0, 0.02]),
                       rng.choice([0, 0.3]), 0.1, rng.choice([0, 5, 30]))
    return dict(kind="combo", template=template, strand=strand, pool0=p0, pool1=p1, mismatches=mm, use_first=first, reads=reads)


def random_dual_case(rng: random.Random, hazard_free: bool = True, sizes=(1, 30, 150), max_mm: int = 2, wide: bool = False) -> dict:
    l1, l2 = rng.choice([4, 6, 9, 12]), rng.choice([4, 7, 12])
    if wide == "big":  # a barcode of 65..240 bases on either mate (big keys for both)
        l1, l2 = rng.choice([(65, 7), (9, 100), (128, 129), (70, 34), (240, 240)])
    elif wide:        # a barcode of 33..64 bases on either mate (wide keys for both)
        l1, l2 = rng.choice([(33, 7), (9, 40), (64, 64), (35, 34)])
    mm1, mm2 = rng.randint(0, max_mm), rng.randint(0, max_mm)
    # pools whose members are >= 2*cap+1 apart cannot trigger the reference's order-dependent
    # segmented-search cache
    d1 = 2 * mm1 + 1 if hazard_free else 1
    d2 = 2 * mm2 + 1 if hazard_free else 1
    u1 = make_pool(rng, rng.choice([1, 3, 8]), l1, BASES, min_dist=d1)
    u2 = make_pool(rng, rng.choice([1, 3, 8]), l2, BASES, min_dist=d2)
    allpairs = [(a, b) for a in u1 for b in u2]
    rng.shuffle(allpairs)
    pairs = allpairs[: rng.randint(1, len(allpairs))]
    pool1 = [a for a, _ in pairs]
    pool2 = [b for _, b in pairs]
    t1 = make_template(rng, 1, [l1], rng.choice([0, 2, 4]), rng.choice([4, 8]))
    t2 = make_template(rng, 1, [l2], rng.choice([0, 2, 4]), rng.choice([4, 8]))
    rev1, rev2 = rng.random() < 0.3, rng.random() < 0.3
    randomized = rng.random() < 0.4
    first = rng.random() < 0.5
    n = rng.choice(sizes)
    r1s, r2s = [], []
    p_sub, p_n = rng.choice([0, 0.03, 0.08]), rng.choice([0, 0.02])
    for _ in range(n):
        u = rng.random()
        if u < 0.1:
            a, b = rand_seq(rng, rng.randint(0, 30)), rand_seq(rng, rng.randint(0, 30))
        else:
            if u < 0.8:
                x, y = rng.choice(pairs)
            else:
                x, y = rng.choice(u1), rng.choice(u2)
            a = mutate(rng, fill_template(t1, [x]), p_sub, p_n, 0.05)
            b = mutate(rng, fill_template(t2, [y]), p_sub, p_n, 0.05)
            pad = rng.choice([0, 4, 20])
            a = rand_seq(rng, rng.randint(0, pad)) + a + rand_seq(rng, rng.randint(0, pad))
            b = rand_seq(rng, rng.randint(0, pad)) + b + rand_seq(rng, rng.randint(0, pad))
            if rng.random() < 0.1:
                x2, _ = rng.choice(pairs)
                a += mutate(rng, fill_template(t1, [x2]), p_sub, p_n, 0.0)
            if rev1:
                a = rc(a)
            if rev2:
                b = rc(b)
            if randomized and rng.random() < 0.5:
                a, b = b, a
        r1s.append(a)
        r2s.append(b)
    return dict(kind="dual", template1=t1, reverse1=rev1, mismatches1=mm1, pool1=pool1,
                template2=t2, reverse2=rev2, mismatches2=mm2, pool2=pool2,
                randomized=randomized, use_first=first, reads1=r1s, reads2=r2s)


def random_paired_combo_case(rng: random.Random, sizes=(1, 30, 150), max_mm: int = 2, wide: bool = False) -> dict:
    """countPairedComboBarcodes: the reads of a dual case against the two pools taken independently
    (distinct barcodes, no list of valid pairs; close barcodes allowed so that ties occur)."""
    c = random_dual_case(rng, hazard_free=rng.random() < 0.5, sizes=sizes, max_mm=max_mm, wide=wide)
    pool1 = list(dict.fromkeys(c["pool1"]))
    pool2 = list(dict.fromkeys(c["pool2"]))
    return dict(kind="paired_combo", template1=c["template1"], reverse1=c["reverse1"], mismatches1=c["mismatches1"], pool1=pool1,
                template2=c["template2"], reverse2=c["reverse2"], mismatches2=c["mismatches2"], pool2=pool2,
                randomized=c["randomized"], use_first=c["use_first"], reads1=c["reads1"], reads2=c["reads2"])


def random_dual_single_end_case(rng: random.Random, sizes=(1, 30, 150), wide: bool = None, diag: bool = False, nreg: int = None) -> dict:
    """countDualBarcodesSingleEnd: the variable regions of one read, pools aligned by row (row c = valid combination c);
    `wide` forces a combined key longer than 32 bases; nreg >= 3 exercises templates with many regions."""
    if nreg is None:
        nreg = 2 if diag else rng.choice([1, 2, 2])
    if wide is None:
        wide = rng.random() < 0.4
    if wide == "big":      # a combined key of 65..256 bases
        lens = rng.choice({1: [[65], [100], [230]], 2: [[40, 40], [20, 100], [100, 100], [70, 5]], 3: [[40, 40, 40], [30, 5, 90]], 4: [[20, 30, 40, 50]],
                           5: [[30, 30, 30, 30, 30]]}[nreg])
    elif nreg >= 3:
        lens = rng.choice({3: [[4, 6, 5], [8, 8, 8], [12, 20, 10], [20, 20, 20], [3, 30, 7]], 4: [[4, 4, 4, 4], [10, 10, 10, 10], [16, 16, 16, 16]],
                           5: [[3, 4, 5, 6, 7], [12, 12, 12, 12, 12]]}[nreg])
    elif nreg == 1:
        lens = [rng.choice([33, 40, 64] if wide else [4, 9, 20])]
    else:
        # (include.invalid=TRUE searches each region on its own with the narrow index: regions <= 32 bases there)
        lens = rng.choice(([[20, 20], [17, 30], [32, 32]] if diag else [[20, 20], [17, 30], [32, 32], [5, 40]]) if wide else [[4, 6], [8, 8], [12, 20]])
    alphabet = rng.choice(["AC", BASES])
    n = min(rng.choice([1, 4, 25]), len(alphabet) ** min(sum(lens), 8) // 2)      # distinct rows must exist
    seen, rows = set(), []
    while len(rows) < n:
        row = tuple(rand_seq(rng, l, alphabet) for l in lens)
        if diag and rows and rng.random() < 0.4:      # a barcode shared by several combinations (duplicates within one column)
            row = (rng.choice(rows)[0], row[1]) if rng.random() < 0.5 else (row[0], rng.choice(rows)[1])
        if "".join(row) not in seen:
            seen.add("".join(row))
            rows.append(row)
    pools = [[row[r] for row in rows] for r in range(nreg)]
    template = make_template(rng, nreg, lens, rng.choice([0, 2, 5]), rng.choice([5, 9]))
    strand = rng.choice([0, 1, 2])
    mm = rng.randint(0, 3)
    first = rng.random() < 0.5
    reads = []
    p_sub, p_n = rng.choice([0, 0.01, 0.04]), rng.choice([0, 0.01])
    for _ in range(rng.choice(sizes)):
        u = rng.random()
        if u < 0.1:
            reads.append(rand_seq(rng, rng.randint(0, len(template) + 20)))
            continue
        row = rng.choice(rows) if u < 0.85 else tuple(rng.choice(rows)[r] for r in range(nreg))   # some mixed (invalid) rows
        core = mutate(rng, fill_template(template, list(row)), p_sub, p_n, 0.05)
        read = rand_seq(rng, rng.randint(0, 12)) + core + rand_seq(rng, rng.randint(0, 12))
        if strand == 1 or (strand == 2 and rng.random() < 0.5):
            read = rc(read)
        reads.append(read)
    return dict(kind="dual_single_end", template=template, strand=strand, pools=pools, mismatches=mm, use_first=first, reads=reads)


def random_big_match_case(rng: random.Random) -> dict:
    """matchBarcodes with choices of 65..256 bases (big keys)."""
    vlen = rng.choice([65, 100, 128, 200, 256])
    pool = make_pool(rng, rng.choice([1, 5, 30]), vlen, "ACGT")
    seqs = [mutate(rng, rng.choice(pool), 0.01, 0.003, 0.1) for _ in range(30)]
    subs, rev = rng.choice([0, 1, 2, 3]), rng.random() < 0.5
    if rev:
        seqs = [rc(s) if set(s.upper()) <= set("ACGT") else s for s in seqs]
    return dict(kind="match", sequences=seqs, choices=pool, substitutions=subs, reverse=rev)


def large_grid_case(seed: int = 77, n_pool: int = 40000, n_reads: int = 21000) -> dict:
    """countComboBarcodes with 2 x 40 000 barcodes (1.6e9 possible combinations: beyond any dense histogram), the inputs
    regenerated from a seed wherever they are needed (tests/golden/kaori_large_grid.json holds a digest of them)."""
    rng = random.Random(seed)

    def pool(length):
        seen = set()
        while len(seen) < n_pool:
            seen.add(rand_seq(rng, length))
        out = sorted(seen)
        rng.shuffle(out)
        return out
    pool0, pool1 = pool(12), pool(10)
    template = "ACGT" + "-" * 12 + "GGTACC" + "-" * 10 + "TTGA"
    pairs = [(rng.randrange(n_pool), rng.randrange(n_pool)) for _ in range(3000)]
    reads = []
    for _ in range(n_reads):
        a, b = rng.choice(pairs) if rng.random() < 0.7 else (rng.randrange(n_pool), rng.randrange(n_pool))
        s = fill_template(template, [pool0[a], pool1[b]])
        s = rand_seq(rng, rng.randrange(0, 20)) + s + rand_seq(rng, rng.randrange(0, 20))
        if rng.random() < 0.3:
            s = rc(s)
        reads.append(mutate(rng, s, 0.01, 0.002, 0.0))
    return dict(kind="combo", template=template, strand=2, pool0=pool0, pool1=pool1, mismatches=1, use_first=True, reads=reads)


def case_digest(case: dict) -> str:
    import hashlib
    h = hashlib.sha256()
    for key in sorted(case):
        v = case[key]
        h.update(key.encode())
        h.update(("\n".join(v) if isinstance(v, list) else repr(v)).encode())
    return h.hexdigest()


def random_random_barcode_case(rng: random.Random, sizes=(1, 30, 150)) -> dict:
    """countRandomBarcodes: unknown sequences in the variable region; asymmetric flanks exercise the
    reference's use of forward coordinates on the reverse strand, lower case / N its string handling."""
    vlen = rng.choice([3, 6, 10, 20, 40])
    template = make_template(rng, 1, [vlen], rng.choice([0, 2, 5]), rng.choice([5, 9, 14]))
    strand = rng.choice([0, 1, 2])
    mm = rng.randint(0, 2)
    first = rng.random() < 0.5
    alphabet = rng.choice(["AC", BASES])
    some = [rand_seq(rng, vlen, alphabet) for _ in range(rng.choice([1, 3, 10]))]
    reads = []
    p_sub, p_n, p_low = rng.choice([0, 0.02, 0.06]), rng.choice([0, 0.02]), rng.choice([0, 0.1])
    for _ in range(rng.choice(sizes)):
        if rng.random() < 0.1:
            reads.append(rand_seq(rng, rng.randint(0, len(template) + 20)))
            continue
        core = mutate(rng, fill_template(template, [rng.choice(some)]), p_sub, p_n, p_low)
        read = rand_seq(rng, rng.randint(0, 12)) + core + rand_seq(rng, rng.randint(0, 12))
        if rng.random() < 0.15:
            read += rand_seq(rng, rng.randint(0, 3)) + mutate(rng, fill_template(template, [rng.choice(some)]), p_sub, p_n, p_low)
        if strand == 1 or (strand == 2 and rng.random() < 0.5):
            read = rc(read)
        reads.append(read)
    return dict(kind="random", template=template, strand=strand, mismatches=mm, use_first=first, reads=reads)


def random_match_case(rng: random.Random) -> dict:
    vlen = rng.choice([3, 5, 8, 12])
    alphabet = rng.choice(["AC", BASES])
    pool = make_pool(rng, rng.choice([1, 4, 30]), vlen, alphabet, iupac_rate=rng.choice([0, 0.1]))
    seqs = [mutate(rng, concrete(rng, rng.choice(pool)), 0.15, 0.03, 0.1) for _ in range(40)]
    return dict(kind="match", sequences=seqs, choices=pool, substitutions=rng.choice([0, 1, 2, 3]), reverse=rng.random() < 0.5)


def write_bgzf(path: str, data: bytes, block: int = 60000, level: int = 6, eof_block: bool = True) -> None:
    """BGZF ("blocked gzip", SAM/BAM specification section 4.1; what bgzip writes): a series of gzip members of at most
    64 KiB, each carrying its own compressed size in a 'BC' extra subfield, optionally ended by the empty EOF member."""
    import struct
    import zlib

    def member(chunk: bytes) -> bytes:
        co = zlib.compressobj(level, zlib.DEFLATED, -15)
        body = co.compress(chunk) + co.flush()
        bsize = 12 + 6 + len(body) + 8
        assert bsize <= 65536
        head = b"\x1f\x8b\x08\x04" + b"\x00\x00\x00\x00" + b"\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, bsize - 1)
        return head + body + struct.pack("<II", zlib.crc32(chunk) & 0xFFFFFFFF, len(chunk) & 0xFFFFFFFF)

    with open(path, "wb") as f:
        for a in range(0, len(data), block):
            f.write(member(data[a:a + block]))
        if eof_block:
            f.write(member(b""))


def fastq_text(reads, name_prefix: str = "r", trailing_newline: bool = True) -> bytes:
    out = b"".join(b"@%s%d some comment\n" % (name_prefix.encode(), i) + (r.encode() if isinstance(r, str) else bytes(r)) + b"\n+\n" +
                   b"I" * len(r) + b"\n" for i, r in enumerate(reads))
    return out if trailing_newline else out[:-1]
