#!/usr/bin/env python3
"""Time-boxed fuzz of the ingestion paths on the GPU: random FASTQ text -- ordinary records of random shapes, and now and
then something the fast paths must hand back (multi-line records, blank lines, missing '@' / '+', truncated ends, CRLF,
a flipped bit in a BGZF member) -- counted through every path:
  plain file   host record scan | device record scan
  BGZF         members inflated on the device | by the host threads
  gzip         ordinary members (one or several, any level), decoded by the device in chunks (members of any size under
               SCG_DGZIP_MIN_MEMBER_CHUNKS=0; what it does not take is handed back) | by the host threads in chunks | by one stream
with random window sizes; every path must give what the sequential reference-exact reader gives (SCG_DEVICE_SCAN=0):
the same counts, or the same error.
usage: python3 tools/gpu_ingest_fuzz.py [seconds] [first_seed]"""
import os
import random
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import screencounter_amd as sc  # noqa: E402
from screencounter_amd import _lib  # noqa: E402
from tests import gen  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 50_000
TEMPLATE = "ACGTACGA" + "-" * 12 + "TGCATGCA"
MODE_VARS = ("SCG_DEVICE_SCAN", "SCG_HOST_SCAN", "SCG_DEVICE_INFLATE", "SCG_WINDOW_KB", "SCG_PGZIP", "SCG_PGZIP_CHUNK_KB", "SCG_DEVICE_GUNZIP", "SCG_DGZIP_CHUNK_KB", "SCG_DGZIP_GROUP_KB", "SCG_DGZIP_MIN_MEMBER_CHUNKS", "SCG_DGZIP_TAIL_GROUP")


def set_mode(**kw):
    for v in MODE_VARS:
        os.environ.pop(v, None)
    for k, v in kw.items():
        if v is not None:
            os.environ[k] = str(v)


def run(path, pool):
    try:
        c, t = sc.count_single_barcodes(path, TEMPLATE, 2, pool, 1, True, 4)
        return ("ok", t, c.tobytes())
    except _lib.ScgError as e:
        return ("error", e.code, str(e))


def run_paired(p1, p2, pool1, pool2):
    try:
        c, t = sc.count_dual_barcodes(p1, T1, False, 1, pool1, p2, T2, False, 1, pool2, False, True, False, 4)
        return ("ok", t, c.tobytes())
    except _lib.ScgError as e:
        return ("error", e.code, str(e))


T1, T2 = "ACGTAC" + "-" * 10 + "TGCATG", "GGATCC" + "-" * 8 + "AAGCTT"


def make_pair(rng):
    """Two mates of the same (or, now and then, not the same) number of records; ordinary text."""
    u1, u2 = gen.make_pool(rng, 10, 10, "ACGT", min_dist=3), gen.make_pool(rng, 8, 8, "ACGT", min_dist=3)
    pairs = [(a, b) for a in u1 for b in u2]
    rng.shuffle(pairs)
    pairs = pairs[:40]
    n = rng.choice([1, 50, 2000, 9000])
    r1, r2 = [], []
    for _ in range(n):
        a, b = rng.choice(pairs) if rng.random() < 0.8 else (rng.choice(u1), rng.choice(u2))
        r1.append(gen.rand_seq(rng, rng.randint(0, 80)) + gen.mutate(rng, gen.fill_template(T1, [a]), 0.02, 0.01, 0.02))
        r2.append(gen.rand_seq(rng, rng.randint(0, 8)) + gen.mutate(rng, gen.fill_template(T2, [b]), 0.02, 0.01, 0.02))
    if rng.random() < 0.15 and n > 3:
        r2 = r2[:-rng.randint(1, 3)]
    t1 = gen.fastq_text(r1, name_prefix=rng.choice(["r", "a_much_longer_read_name_"]), trailing_newline=rng.random() < 0.8)
    t2 = gen.fastq_text(r2, trailing_newline=rng.random() < 0.8)
    return [a for a, _ in pairs], [b for _, b in pairs], t1, t2


def make_text(rng):
    pool = gen.make_pool(rng, 40, 12, "ACGT")
    n = rng.choice([0, 1, 5, 200, 3000, 12000])
    reads = gen.make_reads(rng, TEMPLATE, [pool], n, 2, 0.03, 0.01, 0.02, 0.1, rng.choice([0, 10, 60, 200])) if n else []
    if reads and rng.random() < 0.3:
        reads[rng.randrange(len(reads))] = ""                      # an empty read
    recs = []
    for i, r in enumerate(reads):
        name = rng.choice(["r%d" % i, "@odd+name %d" % i, "x" * rng.randint(1, 300), ""])
        plus = rng.choice(["", "", "", "r%d" % i])
        recs.append(b"@" + name.encode() + b"\n" + r.encode() + b"\n+" + plus.encode() + b"\n" + bytes(rng.choice(b"FI:,#") for _ in r) + b"\n")
    flaw = rng.choice(["none"] * 5 + ["multiline", "blank", "no_at", "no_plus", "short_quality", "truncated", "crlf", "no_final_newline", "plus_in_seq"])
    if recs:
        k = rng.randrange(len(recs))
        r = reads[k] or "ACGT"
        if flaw == "multiline" and len(r) >= 2:
            recs[k] = b"@m\n" + r[:len(r) // 2].encode() + b"\n" + r[len(r) // 2:].encode() + b"\n+\n" + b"I" * len(r) + b"\n"
        elif flaw == "blank":
            recs[k] = recs[k] + b"\n"
        elif flaw == "no_at":
            recs[k] = recs[k][1:]
        elif flaw == "no_plus":
            recs[k] = recs[k].replace(b"\n+", b"\n-", 1)
        elif flaw == "short_quality":
            recs[k] = b"@q\n" + r.encode() + b"\n+\n" + b"I" * max(len(r) - 1, 0) + b"\n"
        elif flaw == "plus_in_seq":
            recs[k] = b"@p\nAC+GT\n+\nIIIII\n"
    text = b"".join(recs)
    if flaw == "truncated" and len(text) > 10:
        text = text[:-rng.randint(1, min(len(text) - 1, 200))]
    elif flaw == "crlf":
        text = text.replace(b"\n", b"\r\n")
    elif flaw == "no_final_newline" and text.endswith(b"\n"):
        text = text[:-1]
    return pool, text, flaw


tally = {}
t0 = time.time()
it = 0
with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as tmp:
    last = t0
    while time.time() - t0 < budget:
        seed = seed0 + it
        it += 1
        rng = random.Random(seed)
        if it % 4 == 0:
            pool1, pool2, t1, t2 = make_pair(rng)
            files = {}
            for name, text in (("m1", t1), ("m2", t2)):
                files[name] = os.path.join(tmp, name + ".fastq")
                open(files[name], "wb").write(text)
                files[name + "z"] = os.path.join(tmp, name + ".bgzf.gz")
                gen.write_bgzf(files[name + "z"], text, block=rng.choice([900, 9000, 65280]))
            set_mode(SCG_DEVICE_SCAN=0)
            want = run_paired(files["m1"], files["m2"], pool1, pool2)
            kb = rng.choice([None, 16, 100, 700])
            for name, a, b, env in (("plain+plain", "m1", "m2", {}), ("bgzf+bgzf device", "m1z", "m2z", {}), ("bgzf+plain host", "m1z", "m2", dict(SCG_DEVICE_INFLATE=0)),
                                    ("plain+bgzf device scan", "m1", "m2z", dict(SCG_HOST_SCAN=0))):
                set_mode(SCG_WINDOW_KB=kb, **env)
                got = run_paired(files[a], files[b], pool1, pool2)
                if got != want:
                    print(f"MISMATCH seed {seed} paired mode {name} window_kb {kb}: got {got[:2]} want {want[:2]}", flush=True)
                    sys.exit(1)
            key = "paired:" + want[0]
            tally[key] = tally.get(key, 0) + 1
            continue
        pool, text, flaw = make_text(rng)
        plain = os.path.join(tmp, "f.fastq")
        open(plain, "wb").write(text)
        bg = os.path.join(tmp, "f.bgzf.gz")
        gen.write_bgzf(bg, text, block=rng.choice([300, 3000, 30000, 65280]), eof_block=rng.random() < 0.8)
        bitflip = rng.random() < 0.1 and os.path.getsize(bg) > 100
        if bitflip:
            raw = bytearray(open(bg, "rb").read())
            raw[rng.randrange(30, len(raw) - 10)] ^= 1 << rng.randrange(8)
            open(bg, "wb").write(bytes(raw))
        # the same text as ordinary gzip: one member or a few, any level, now and then a flipped bit
        import zlib
        gz = os.path.join(tmp, "f.plain.gz")
        cuts = sorted(rng.sample(range(len(text) + 1), min(rng.choice([0, 0, 1, 3]), len(text) + 1))) if text else []
        parts = [text[a:b] for a, b in zip([0] + cuts, cuts + [len(text)])]
        raw = b""
        for part in parts:
            c = zlib.compressobj(rng.choice([1, 4, 6, 9]), zlib.DEFLATED, 31)
            raw += c.compress(part) + c.flush()
        gzflip = rng.random() < 0.15 and len(raw) > 100
        if gzflip:
            raw = bytearray(raw)
            raw[rng.randrange(12, len(raw) - 8)] ^= 1 << rng.randrange(8)
        open(gz, "wb").write(bytes(raw))
        set_mode(SCG_DEVICE_SCAN=0)
        want = run(plain, pool)
        want_bg = run(bg, pool) if bitflip else want
        want_gz = run(gz, pool) if gzflip else want
        kb = rng.choice([None, None, 4, 16, 100, 700])
        modes = [("host_scan", plain, dict(SCG_WINDOW_KB=kb)), ("device_scan", plain, dict(SCG_HOST_SCAN=0, SCG_WINDOW_KB=kb)),
                 ("device_inflate", bg, dict(SCG_WINDOW_KB=kb)), ("host_inflate", bg, dict(SCG_DEVICE_INFLATE=0, SCG_WINDOW_KB=kb)),
                 ("gzip_parallel", gz, dict(SCG_PGZIP_CHUNK_KB=rng.choice([4, 16, 64]), SCG_DEVICE_GUNZIP=0, SCG_WINDOW_KB=kb)),
                 ("gzip_device", gz, dict(SCG_PGZIP_CHUNK_KB=16, SCG_DGZIP_CHUNK_KB=rng.choice([4, 16, 64]), SCG_DGZIP_GROUP_KB=rng.choice([None, 128, 512]), SCG_DGZIP_MIN_MEMBER_CHUNKS=rng.choice([None, 0, 0, 1]), SCG_DGZIP_TAIL_GROUP=rng.choice([None, 1, 2, 5]),
                                         SCG_WINDOW_KB=kb)),
                 ("gzip_stream", gz, dict(SCG_PGZIP=0, SCG_WINDOW_KB=kb))]
        for name, path, env in modes:
            set_mode(**env)
            if os.environ.get("FUZZ_VERBOSE"):
                print(f"seed {seed} flaw {flaw} bitflip {bitflip} mode {name} window_kb {kb} bytes {len(text)}", file=sys.stderr, flush=True)
            got = run(path, pool)
            exp = want_bg if path == bg else (want_gz if path == gz else want)
            if got != exp:
                print(f"MISMATCH seed {seed} file {it} flaw {flaw} bitflip {bitflip} gzflip {gzflip} mode {name} env {env} window_kb {kb}: got {got[:2]} {got[2][:80] if got[0] == 'error' else ''} "
                      f"want {exp[:2]} {exp[2][:80] if exp[0] == 'error' else ''}", flush=True)
                again = [run(path, pool) for _ in range(3)]                  # a race, or state left behind by an earlier call?
                print(f"  the same call three more times: {[a[:2] + (a == exp,) for a in again]}", flush=True)
                if got[0] == "ok" and exp[0] == "ok":
                    g, w = np.frombuffer(got[2], dtype=np.int32), np.frombuffer(exp[2], dtype=np.int32)
                    print(f"  counts differ at {[(int(i), int(g[i]), int(w[i])) for i in np.nonzero(g != w)[0][:12]]}", flush=True)
                set_mode(SCG_DEVICE_SCAN=0)
                print(f"  the sequential reader once more: {run(path, pool) == exp}", flush=True)
                import shutil
                keep = os.path.join(ROOT, "gpurun_out", f"fuzz_mismatch_{seed}" + os.path.splitext(path)[1])
                os.makedirs(os.path.dirname(keep), exist_ok=True)
                shutil.copy(path, keep)
                sys.exit(1)
        key = flaw + ("+bitflip" if bitflip else "") + ("+gzflip" if gzflip else "") + ":" + want[0]
        tally[key] = tally.get(key, 0) + 1
        if time.time() - last > 10:
            last = time.time()
            print(f"{it} files, {time.time() - t0:.0f} s", flush=True)
set_mode()
print(f"ingest fuzz: {it} files x 7 paths, no mismatch; {dict(sorted(tally.items()))}")
