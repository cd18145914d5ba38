"""GPU box: count one FASTQ file under several settings (debug aid).  usage: python3 tools/repro_gz_count.py <file> "VAR=val ..." ..."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = '''
import sys
sys.path.insert(0, %r)
import random
import screencounter_amd as sc
from tests import gen
TEMPLATE = "ACGTACGA" + "-" * 12 + "TGCATGCA"
rng = random.Random(70860)
pool = gen.make_pool(rng, 40, 12, "ACGT")
c, t = sc.count_single_barcodes(sys.argv[1], TEMPLATE, 2, pool, 1, True, 4)
print(t, int(c.sum()))
''' % ROOT
for setting in sys.argv[2:]:
    env = dict(os.environ)
    for kv in setting.split():
        if "=" in kv:
            k, v = kv.split("=", 1)
            env[k] = v
    r = subprocess.run([sys.executable, "-c", code, sys.argv[1]], env=env, capture_output=True, text=True)
    print("[%s] %s %s" % (setting, r.stdout.strip(), r.stderr.strip()[-300:]), flush=True)
