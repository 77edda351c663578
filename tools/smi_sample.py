"""Sample rocm-smi clocks / power while a command runs; prints a small JSON summary (median / max sclk under load).
usage: python tools/smi_sample.py <out.json> -- <command ...>"""
import json
import re
import subprocess
import sys
import threading
import time

out, cmd = sys.argv[1], sys.argv[sys.argv.index('--') + 1:]
samples = []
stop = False


def loop():
    while not stop:
        try:
            t = subprocess.run(['rocm-smi', '--showclocks', '--showpower', '--json'], capture_output=True, text=True, timeout=10).stdout
            m = re.search(r'"sclk clock speed:?": "\((\d+)Mhz\)"', t)
            p = re.search(r'Power \(W\)": "([0-9.]+)"', t)
            mc = re.search(r'"mclk clock speed:?": "\((\d+)Mhz\)"', t)
            samples.append((time.time(), int(m.group(1)) if m else None, float(p.group(1)) if p else None,
                            int(mc.group(1)) if mc else None))
        except Exception:
            pass
        time.sleep(0.1)


th = threading.Thread(target=loop, daemon=True)
th.start()
rc = subprocess.call(cmd)
stop = True
th.join(timeout=5)
sclk = sorted(s[1] for s in samples if s[1])
pw = sorted(s[2] for s in samples if s[2])
busy = [s for s in samples if s[2] and s[2] > 0.6 * (pw[-1] if pw else 1)]
bs = sorted(s[1] for s in busy if s[1])
json.dump({'samples': len(samples), 'sclk_mhz_all': {'min': sclk[0] if sclk else None, 'median': sclk[len(sclk) // 2] if sclk else None,
                                                      'max': sclk[-1] if sclk else None},
           'sclk_mhz_under_load(power>60%max)': {'n': len(bs), 'min': bs[0] if bs else None, 'median': bs[len(bs) // 2] if bs else None,
                                                 'max': bs[-1] if bs else None},
           'power_w': {'median': pw[len(pw) // 2] if pw else None, 'max': pw[-1] if pw else None},
           'mclk_mhz': sorted(set(s[3] for s in samples if s[3]))}, open(out, 'w'), indent=1)
sys.exit(rc)
