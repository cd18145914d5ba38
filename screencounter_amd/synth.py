"""Synthetic workloads of SURVEY.md section 8(d), generated on the GPU.

Reads are produced directly in HBM by ``scg_synth_reads`` (counter-based splitmix64 keyed by the
global read index, so any shard of any size can be regenerated on any rank).  Libraries are drawn
on the host with numpy from the same base seed.  Nothing here is part of the counted hot path.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _lib
from ._lib import SynthSpec, check, errbuf

FLANK5 = "CAGCTACGTACGATCGGTCA"
FLANK3 = "CCAGCTCGATCGTAGCATGC"
BASE_SEED = 0x5C0DE000


def random_library(n: int, length: int, seed: int, min_dist: int = 1) -> List[str]:
    """n distinct uniform-random k-mers (rejection on exact duplicates; optionally on Hamming
    distance < min_dist, used for the dual pools of config 4)."""
    rng = np.random.default_rng(seed)
    letters = np.frombuffer(b"ACGT", dtype=np.uint8)
    if min_dist <= 1:
        out: dict = {}
        while len(out) < n:
            block = letters[rng.integers(0, 4, size=(n, length))]
            for row in block:
                s = row.tobytes().decode()
                if s not in out:
                    out[s] = None
                    if len(out) == n:
                        break
        return list(out.keys())
    kept = np.zeros((0, length), dtype=np.uint8)
    while kept.shape[0] < n:
        cand = letters[rng.integers(0, 4, size=(1, length))]
        if kept.shape[0] == 0 or int((kept != cand).sum(axis=1).min()) >= min_dist:
            kept = np.concatenate([kept, cand])
    return [row.tobytes().decode() for row in kept]


@dataclass
class Workload:
    """One of BASELINE.json's configs, in the vocabulary of the reference's entry points."""
    config_id: int
    entry: str                      # "single" | "combo" | "dual"
    n_reads: int                    # reads (or pairs) per GPU
    read_len: int
    template: str                   # '-' marks variable bases
    pools: List[List[str]]
    mismatches: int
    strand: int = 2                 # 0 forward, 1 reverse, 2 both (single/combo)
    use_first: bool = True
    # dual only
    template2: Optional[str] = None
    pair_index: Optional[np.ndarray] = None     # int32 [n_pairs, 2] rows into pools[0], pools[1]
    p_sub: float = 0.005
    p_n: float = 0.001
    p_junk: float = 0.05
    p_invalid_pair: float = 0.0
    seed: int = field(default=0)

    @property
    def p_reverse(self) -> float:
        return 0.5 if (self.entry != "dual" and self.strand == 2) else (1.0 if self.strand == 1 and self.entry != "dual" else 0.0)

    @property
    def bytes_per_unit(self) -> int:
        """Algorithmic bytes per read (pair): the sequence bytes that must be fetched once."""
        return self.read_len * (2 if self.entry == "dual" else 1)

    def describe(self) -> str:
        unit = "pairs" if self.entry == "dual" else "reads"
        lib = " x ".join(str(len(p)) for p in self.pools)
        return (f"config{self.config_id}: count{self.entry.capitalize()}Barcodes {self.n_reads} {unit} x {self.read_len}bp, "
                f"library {lib}, <={self.mismatches}mm, use_first={self.use_first}")


def _regions(template: str) -> List[Tuple[int, int]]:
    out, i = [], 0
    while i < len(template):
        if template[i] == "-":
            j = i
            while j < len(template) and template[j] == "-":
                j += 1
            out.append((i, j - i))
            i = j
        else:
            i += 1
    return out


def workload(config_id: int, n_reads: Optional[int] = None, n_library: Optional[int] = None) -> Workload:
    """The five configurations of BASELINE.json (shapes from SURVEY.md 8d).  n_reads / n_library
    override the size for tests and bounded samples; everything else stays as specified."""
    seed = BASE_SEED + config_id
    if config_id == 1:
        t = FLANK5[-12:] + "-" * 20 + FLANK3[:12]
        return Workload(1, "single", n_reads or 1_000_000, 75, t, [random_library(n_library or 1000, 20, seed)], 0, strand=0, seed=seed)
    if config_id in (2, 5):
        t = FLANK5 + "-" * 20 + FLANK3
        return Workload(config_id, "single", n_reads or 100_000_000, 150, t, [random_library(n_library or 100_000, 20, BASE_SEED + 2)],
                        1 if config_id == 2 else 2, strand=2, seed=seed)
    if config_id == 3:
        t = FLANK5[:12] + "-" * 14 + FLANK5[12:] + "-" * 14 + FLANK3[:12]
        n = n_library or 500
        return Workload(3, "combo", n_reads or 100_000_000, 150, t,
                        [random_library(n, 14, seed), random_library(n, 14, seed + 1000)], 1, strand=2, seed=seed)
    if config_id == 4:
        t1 = FLANK5 + "-" * 20 + FLANK3
        t2 = FLANK3 + "-" * 20 + FLANK5          # mate 2 uses the two flanks swapped
        n_unique = 2000 if n_library is None else max(2, int(round(n_library ** 0.5)) * 2)
        n_pairs = n_library or 50_000
        u1 = random_library(n_unique, 20, seed, min_dist=3)
        u2 = random_library(n_unique, 20, seed + 1000, min_dist=3)
        rng = np.random.default_rng(seed + 2000)
        flat = rng.choice(n_unique * n_unique, size=n_pairs, replace=False)
        pair_index = np.stack([flat // n_unique, flat % n_unique], axis=1).astype(np.int32)
        pool1 = [u1[i] for i in pair_index[:, 0]]
        pool2 = [u2[j] for j in pair_index[:, 1]]
        w = Workload(4, "dual", n_reads or 50_000_000, 150, t1, [pool1, pool2], 1, strand=0, template2=t2,
                     pair_index=np.stack([np.arange(n_pairs), np.arange(n_pairs)], axis=1).astype(np.int32),
                     p_invalid_pair=0.05 / 0.95, seed=seed)
        return w
    raise ValueError(f"unknown config {config_id}")


class DeviceWorkload:
    """A Workload's pools/template uploaded once, able to fill read buffers on its device."""

    def __init__(self, w: Workload, device):
        import torch
        self.w = w
        self.device = torch.device(device)
        self._lib = _lib.load()
        self._keep = []

        def up(b: bytes):
            t = torch.from_numpy(np.frombuffer(b, dtype=np.uint8).copy()).to(self.device)
            self._keep.append(t)
            return t

        self.d_template = [up(w.template.encode())]
        if w.template2:
            self.d_template.append(up(w.template2.encode()))
        self.d_pools = [up("".join(p).encode()) for p in w.pools]
        self.d_pair_index = None
        if w.pair_index is not None:
            self.d_pair_index = torch.from_numpy(np.ascontiguousarray(w.pair_index, dtype=np.int32)).to(self.device)

    def _spec(self, mate: int, first_read: int) -> SynthSpec:
        w = self.w
        s = SynthSpec()
        s.seed = w.seed
        s.first_read = first_read
        s.read_len = w.read_len
        tmpl = w.template if mate == 0 else w.template2
        s.template_len = len(tmpl)
        s.d_template = self.d_template[mate].data_ptr()
        regs = _regions(tmpl)
        s.n_regions = len(regs)
        for r, (st, ln) in enumerate(regs):
            s.region_start[r] = st
            s.region_len[r] = ln
        if w.entry == "dual":
            s.d_pool[0] = self.d_pools[mate].data_ptr()
            s.n_pool[0] = len(w.pools[mate])
            s.d_pair_index = self.d_pair_index.data_ptr()
            s.n_pairs = int(self.d_pair_index.shape[0])
            s.pair_column = mate
            s.p_invalid_pair = w.p_invalid_pair
        else:
            for r in range(len(regs)):
                s.d_pool[r] = self.d_pools[r].data_ptr()
                s.n_pool[r] = len(w.pools[r])
            s.d_pair_index = None
            s.n_pairs = 0
            s.pair_column = 0
            s.p_invalid_pair = 0.0
        s.p_sub, s.p_n, s.p_junk, s.p_reverse = w.p_sub, w.p_n, w.p_junk, w.p_reverse
        return s

    def generate(self, n_reads: int, first_read: int = 0, mate: int = 0, out=None):
        """uint8 CUDA tensor of n_reads * read_len bytes (reads back to back)."""
        import torch
        nbytes = n_reads * self.w.read_len
        if out is None:
            out = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=self.device)
        spec = self._spec(mate, first_read)
        err = errbuf()
        with torch.cuda.device(self.device):
            stream = torch.cuda.current_stream().cuda_stream
            check(self._lib.scg_synth_reads(C.byref(spec), C.c_void_p(out.data_ptr()), int(n_reads), C.c_void_p(stream), err, _lib.ERRCAP), err)
        return out

    def plan(self):
        from .engine import Plan
        w = self.w
        dev = self.device.index if self.device.index is not None else 0
        if w.entry == "single":
            return Plan.single(w.template, w.strand, w.pools[0], w.mismatches, w.use_first, device=dev)
        if w.entry == "combo":
            return Plan.combo(w.template, w.strand, w.pools[0], w.pools[1], w.mismatches, w.use_first, device=dev)
        return Plan.dual(w.template, False, w.mismatches, w.pools[0], w.template2, False, w.mismatches, w.pools[1],
                         randomized=False, use_first=w.use_first, device=dev)


# ---------------------------------------------------------------------------------------------
# Host generator: a numpy restatement of synth_kernel (csrc/scg_kernels.hip), byte-identical to the
# device generator for the same (workload, first_read, mate).  It exists so that CPU-only tiers (the
# config-1 fixture, the gloo rehearsal, oracle tests) can produce the benchmark stream without a GPU;
# tests/test_gpu_config_parity.py checks the two generators against each other.
# ---------------------------------------------------------------------------------------------
_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


class _SplitMix:
    """splitmix64 streams, one per read, advanced only where `mask` is set (the device generator's draws
    are data-dependent)."""

    def __init__(self, state: np.ndarray):
        self.s = state.astype(np.uint64)

    def next(self, mask=None) -> np.ndarray:
        with np.errstate(over="ignore"):
            s = self.s + np.uint64(0x9E3779B97F4A7C15)
            if mask is not None:
                s = np.where(mask, s, self.s)
            self.s = s
            z = s
            z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
            z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
            return z ^ (z >> np.uint64(31))

    def u01(self, mask=None) -> np.ndarray:
        return (self.next(mask) >> np.uint64(40)).astype(np.float32) * np.float32(1.0 / 16777216.0)

    def below(self, n: int, mask=None) -> np.ndarray:
        return (((self.next(mask) >> np.uint64(32)) * np.uint64(n)) >> np.uint64(32)).astype(np.int64)


def generate_host(w: Workload, n_reads: int, first_read: int = 0, mate: int = 0) -> np.ndarray:
    """uint8 array of n_reads * read_len bytes: the same stream DeviceWorkload.generate produces."""
    L = w.read_len
    tmpl = w.template if mate == 0 else w.template2
    T = len(tmpl)
    regs = _regions(tmpl)
    B = np.frombuffer(b"ACGT", dtype=np.uint8)
    g = (np.arange(n_reads, dtype=np.uint64) + np.uint64(first_read))
    with np.errstate(over="ignore"):
        seed = np.uint64(w.seed)
        shared = _SplitMix(seed ^ (g * np.uint64(0xD1342543DE82EF95)))
        pair_column = mate if w.entry == "dual" else 0
        priv = _SplitMix((seed + np.uint64(0x632BE59BD9B4E019) * np.uint64(pair_column + 1)) ^ (g * np.uint64(0xA0761D6478BD642F)))
    junk = (shared.u01() < np.float32(w.p_junk)) | (T > L)
    flip = shared.u01() < np.float32(w.p_reverse)
    if w.entry == "dual":
        pools = [np.frombuffer("".join(w.pools[mate]).encode(), dtype=np.uint8).reshape(len(w.pools[mate]), -1)]
        invalid = shared.u01() < np.float32(w.p_invalid_pair)
        row = shared.below(int(w.pair_index.shape[0]))
        row2 = shared.below(int(w.pair_index.shape[0]))
        r = np.where(invalid & (pair_column == 1), row2, row)
        k = [w.pair_index[r, pair_column].astype(np.int64), None]
    else:
        pools = [np.frombuffer("".join(p).encode(), dtype=np.uint8).reshape(len(p), -1) for p in w.pools]
        n0 = len(w.pools[0]) if len(w.pools) > 0 and len(w.pools[0]) > 0 else 1
        n1 = len(w.pools[1]) if len(w.pools) > 1 and len(w.pools[1]) > 0 else 1
        k = [shared.below(n0), shared.below(n1)]
    offset = np.where(junk, 0, shared.below(max(L - T + 1, 1)))
    tb = np.frombuffer(tmpl.encode(), dtype=np.uint8)
    out = np.empty((n_reads, L), dtype=np.uint8)
    code_of = np.zeros(256, dtype=np.int64)
    code_of[ord("C")] = 1
    code_of[ord("G")] = 2
    code_of[ord("T")] = 3
    p_sub, p_n = np.float32(w.p_sub), np.float32(w.p_n)
    for j in range(L):
        c = B[priv.below(4)]
        t = j - offset
        inside = ~junk & (t >= 0) & (t < T)
        tt = np.clip(t, 0, T - 1)
        tc = tb[tt]
        cc = np.where(tc != ord("-"), tc, c)
        for r, (st, ln) in enumerate(regs):
            rel = tt - st
            hit = inside & (rel >= 0) & (rel < ln)
            if hit.any():
                cc = np.where(hit, pools[r][np.where(hit, k[r], 0), np.clip(rel, 0, ln - 1)], cc)
        c = np.where(inside, cc, c)
        sub = inside & (priv.u01(inside) < p_sub)
        if sub.any():
            alt = B[(code_of[c] + 1 + priv.below(3, sub)) & 3]
            c = np.where(sub, alt, c)
        c = np.where(priv.u01() < p_n, np.uint8(ord("N")), c)
        out[:, j] = c
    if flip.any():
        comp = np.arange(256, dtype=np.uint8)
        for a, b in zip(b"ACGT", b"TGCA"):
            comp[a] = b
        out[flip] = comp[out[flip][:, ::-1]]
    return out.reshape(-1)


def reads_to_fastq(path: str, seqs: np.ndarray, read_len: int, start_index: int = 0, chunk: int = 4_000_000) -> int:
    """Write fixed-length reads (uint8 array of n*read_len bytes) as a 4-line FASTQ; returns n."""
    n = seqs.size // read_len
    if n > chunk:                                   # bounded memory for multi-GB files
        open(path, "wb").close()
        for a in range(0, n, chunk):
            b = min(a + chunk, n)
            _reads_to_fastq(path, seqs[a * read_len: b * read_len], read_len, start_index + a, "ab")
        return n
    return _reads_to_fastq(path, seqs, read_len, start_index, "wb")


def _bgzf_member(chunk: bytes) -> bytes:
    import struct
    import zlib
    co = zlib.compressobj(1, zlib.DEFLATED, -15)
    body = co.compress(chunk) + co.flush()
    return (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, 12 + 6 + len(body) + 8 - 1) +
            body + struct.pack("<II", zlib.crc32(chunk) & 0xFFFFFFFF, len(chunk)))


def _bgzf_span(args) -> bytes:
    path, a, b = args
    with open(path, "rb") as f:
        f.seek(a)
        data = f.read(b - a)
    return b"".join(_bgzf_member(data[i:i + 65280]) for i in range(0, len(data), 65280))


def fastq_to_bgzf(src: str, dst: str, workers: int = 8) -> None:
    """Compress a FASTQ file into BGZF (blocked gzip, what bgzip writes: 64 KiB members that carry their size) with a
    pool of worker processes; bench / test infrastructure."""
    import multiprocessing as mp
    size = os.path.getsize(src)
    span = 65280 * 256
    jobs = [(src, a, min(a + span, size)) for a in range(0, size, span)]
    with mp.get_context("fork").Pool(max(1, workers)) as pool, open(dst, "wb") as out:
        for blob in pool.imap(_bgzf_span, jobs):
            out.write(blob)
        out.write(_bgzf_member(b""))


def _reads_to_fastq(path: str, seqs: np.ndarray, read_len: int, start_index: int, mode: str) -> int:
    n = seqs.size // read_len
    width = 10
    rec = 2 + width + 1 + read_len + 1 + 2 + read_len + 1
    buf = np.empty((n, rec), dtype=np.uint8)
    buf[:, 0] = ord("@")
    buf[:, 1] = ord("r")
    idx = np.arange(start_index, start_index + n, dtype=np.int64)
    for d in range(width):
        buf[:, 2 + width - 1 - d] = (idx // (10 ** d)) % 10 + ord("0")
    p = 2 + width
    buf[:, p] = ord("\n")
    buf[:, p + 1:p + 1 + read_len] = seqs[:n * read_len].reshape(n, read_len)
    p += 1 + read_len
    buf[:, p] = ord("\n")
    buf[:, p + 1] = ord("+")
    buf[:, p + 2] = ord("\n")
    buf[:, p + 3:p + 3 + read_len] = ord("I")
    buf[:, p + 3 + read_len] = ord("\n")
    with open(path, mode) as f:
        f.write(buf.tobytes())
    return n
