#!/usr/bin/env python3
"""Eight host threads, each running one-shot registrations, engines with replaced targets and stage-level searches of
random sizes at the same time (ctypes releases the GIL): every result must equal the one the same input gives alone."""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import icp_amd
from icp_amd import synth

sizes = [300, 1000, 2903, 10000, 48485, 120000]
inputs = {}
for n in sizes:
    for seed in range(2):
        inputs[(n, seed)] = synth.make_pair(n, max(500, n // 2 + 17 * seed), seed=700 + seed)

def one(kind, key):
    P, Q = inputs[key]
    if kind == 0:
        return icp_amd.icp_gpu_optimized(P, Q, 3)
    if kind == 1:
        e = icp_amd.Engine(search=["auto", "f32", "grid"][key[1] + (key[0] % 2)])
        e.set_target(Q); e.set_source(P); e.iterate(2)
        e.set_target(Q[: max(1, len(Q) // 2)]); e.iterate(1)
        out = e.get_source(); e.close()
        return out
    return icp_amd.correspondences(P[:20000], P.mean(0), Q - Q.mean(0), search="f32", q_splits=1 + key[1])

want = {(kind, key): one(kind, key) for kind in range(3) for key in inputs}
print("references done (%d cases)" % len(want), flush=True)
errs, done = [], [0]
def work(t, rounds):
    rng = np.random.RandomState(t)
    try:
        for _ in range(rounds):
            kind, key = int(rng.randint(3)), list(inputs)[int(rng.randint(len(inputs)))]
            got = one(kind, key)
            if not np.array_equal(got, want[(kind, key)]):
                errs.append(("mismatch", t, kind, key))
            done[0] += 1
    except Exception as exc:   # noqa: BLE001
        errs.append(repr(exc))
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 60
threads = [threading.Thread(target=work, args=(t, rounds)) for t in range(8)]
t0 = time.time()
for th in threads: th.start()
while any(th.is_alive() for th in threads):
    time.sleep(10)
    print("... %d calls done, %d problems" % (done[0], len(errs)), flush=True)
for th in threads: th.join()
print("%d concurrent calls in %.1f s, problems: %s" % (done[0], time.time() - t0, errs[:5]))
sys.exit(1 if errs else 0)
