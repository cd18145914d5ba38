"""GPU box: hammer the BGZF device path with small windows on a few texts (CRLF and plain), other input forms in between to
vary what the slot cache hands out; every run must give the sequential reader's counts.  usage: python3 tools/stress_bgzf.py [seconds]"""
import os, random, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import screencounter_amd as sc
from tests import gen

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
TEMPLATE = "ACGTACGA" + "-" * 12 + "TGCATGCA"
VARS = ("SCG_DEVICE_SCAN", "SCG_HOST_SCAN", "SCG_DEVICE_INFLATE", "SCG_WINDOW_KB")


def mode(**kw):
    for v in VARS:
        os.environ.pop(v, None)
    for k, v in kw.items():
        if v is not None:
            os.environ[k] = str(v)


def run(path, pool):
    c, t = sc.count_single_barcodes(path, TEMPLATE, 2, pool, 1, True, 4)
    return t, c.copy()


rng = random.Random(7)
cases = []
with tempfile.TemporaryDirectory(dir="/dev/shm") as tmp:
    for k in range(8):
        pool = gen.make_pool(rng, 40, 12, "ACGT")
        reads = gen.make_reads(rng, TEMPLATE, [pool], 12000, 2, 0.03, 0.01, 0.02, 0.1, rng.choice([0, 10, 60, 200]))
        text = gen.fastq_text(reads)
        if k % 2 == 0:
            text = text.replace(b"\n", b"\r\n")
        plain = os.path.join(tmp, f"p{k}.fastq")
        open(plain, "wb").write(text)
        bg = os.path.join(tmp, f"b{k}.gz")
        gen.write_bgzf(bg, text, block=rng.choice([300, 3000, 30000, 65280]))
        mode(SCG_DEVICE_SCAN=0)
        want = run(plain, pool)
        cases.append((pool, plain, bg, want, k % 2 == 0))
    t0 = last = time.time()
    n = bad = 0
    while time.time() - t0 < budget:
        if time.time() - last > 30:
            last = time.time()
            print(f"{n} runs, {bad} mismatches, {time.time() - t0:.0f} s", flush=True)
        pool, plain, bg, want, crlf = rng.choice(cases)
        kb = rng.choice([100, 100, 16, 700, None])
        which = rng.random()
        if which < 0.7:
            mode(SCG_WINDOW_KB=kb); path = bg; name = "device_inflate"
        elif which < 0.8:
            mode(SCG_WINDOW_KB=kb, SCG_DEVICE_INFLATE=0); path = bg; name = "host_inflate"
        elif which < 0.9:
            mode(SCG_WINDOW_KB=kb); path = plain; name = "host_scan"
        else:
            mode(SCG_WINDOW_KB=kb, SCG_HOST_SCAN=0); path = plain; name = "device_scan"
        got = run(path, pool)
        n += 1
        if got[0] != want[0] or not np.array_equal(got[1], want[1]):
            bad += 1
            d = np.nonzero(got[1] != want[1])[0]
            print(f"MISMATCH run {n} {name} window_kb {kb} crlf {crlf} file {os.path.basename(path)}: total {got[0]} vs {want[0]}, counts differ at "
                  f"{[(int(i), int(got[1][i]), int(want[1][i])) for i in d[:8]]}", flush=True)
            again = [run(path, pool) for _ in range(3)]
            print("   again:", [bool(a[0] == want[0] and np.array_equal(a[1], want[1])) for a in again], flush=True)
            if bad >= 5:
                break
    mode()
    print(f"stress: {n} runs, {bad} mismatches", flush=True)
