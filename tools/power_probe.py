#!/usr/bin/env python3
"""Board power and shader clock while the brute-force search runs (rocm-smi sampled from a second thread), against an
idle GPU and against the grid path: direct evidence for 'power bound' beside the clock derived from GRBM_GUI_ACTIVE."""
import os, sys, subprocess, threading, time, re
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import icp_amd
from icp_amd import synth

samples = []
stop = threading.Event()
def sampler():
    while not stop.is_set():
        try:
            out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "-d", "0"], capture_output=True, text=True, timeout=5).stdout
        except Exception as ex:
            out = "ERR %r" % (ex,)
        pw = re.search(r"Power \(W\):\s*([0-9.]+)", out)
        sc = re.search(r"sclk clock level:.*\((\d+)Mhz\)", out)
        samples.append((time.perf_counter(), float(pw.group(1)) if pw else None, int(sc.group(1)) if sc else None, None if (pw or sc) else out[:200]))
        time.sleep(0.05)

def phase(name, fn):
    samples.clear()
    t0 = time.perf_counter()
    fn()
    t1 = time.perf_counter()
    mine = [s for s in samples if t0 + 0.3 <= s[0] <= t1]
    pw = [s[1] for s in mine if s[1] is not None]
    sc = [s[2] for s in mine if s[2] is not None]
    raw = [s[3] for s in mine if s[3]]
    print("%-28s %5.1f s, %3d samples: power avg %s W (max %s), sclk avg %s MHz (min %s)%s" % (
        name, t1 - t0, len(mine), "%.0f" % (sum(pw) / len(pw)) if pw else "n/a", "%.0f" % max(pw) if pw else "n/a",
        "%.0f" % (sum(sc) / len(sc)) if sc else "n/a", min(sc) if sc else "n/a", ("  raw: " + raw[0]) if raw and not pw else ""), flush=True)

th = threading.Thread(target=sampler, daemon=True); th.start()
P, Q = synth.make_pair(1000000, 1000000)
phase("idle", lambda: time.sleep(2.0))
for search, iters in (("f32", 120), ("exact", 14), ("grid", 20000)):
    e = icp_amd.Engine(search=search); e.set_target(Q); e.set_source(P); e.iterate(3); e.sync()
    phase("search=%s x %d iterations" % (search, iters), lambda: (e.iterate(iters), e.sync()))
    e.close()
stop.set()
