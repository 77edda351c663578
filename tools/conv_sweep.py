"""Sweep of the conv micro-benchmark over env-switched kernel variants (one subprocess per variant: the switches are read once
per process).  usage: python tools/conv_sweep.py OUT 'shape-filter ...' 'ENV=1 ENV2=2' 'ENV=..' ...   ('' = defaults)
Extra shapes beyond tools/bench_conv.py's list can be given as SG_BENCH_SHAPES='name:N:Cin:H:Cout:KS:stride:pad:refl:ups;...'"""
import os, subprocess, sys
out, filt, variants = sys.argv[1], sys.argv[2], sys.argv[3:] or ['']
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
with open(out, 'w') as f:
    for v in variants:
        env = dict(os.environ)
        for kv in v.split():
            k, val = kv.split('=', 1)
            env[k] = val
        r = subprocess.run([sys.executable, os.path.join(root, 'tools', 'bench_conv.py')] + filt.split(), env=env,
                           stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
        f.write("== variant '%s'\n%s\n" % (v, r.stdout))
        f.flush()
