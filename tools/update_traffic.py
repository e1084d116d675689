#!/usr/bin/env python3
"""Writes icp_amd/measured_traffic.json from the two rocprofv3 --pmc summaries of a bench run (tools/collect_profiles.sh,
tools/rocprof_summary.py --pmc): FETCH_SIZE and WRITE_SIZE per launch of the brute-force search kernel, with the git blob
hash of the kernel's source (icp_amd/csrc/search_mfma.hip) so that bench.py can tell whether the figure belongs to the
kernel it is running (it reports `traffic: null` otherwise).

    python tools/update_traffic.py profiles/r2_v1_pmc_fetch.txt profiles/r2_v1_pmc_write.txt [n m n_gpus]
"""
import hashlib
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL_SOURCE = os.path.join("icp_amd", "csrc", "search_mfma.hip")


def blob_hash(path):
    """The hash `git hash-object` gives the file (sha1 over "blob <size>\\0" + contents); needs no git on the GPU box."""
    data = open(path, "rb").read()
    return hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()


def per_dispatch(path, kernel_prefix, counter):
    """(value per dispatch, dispatches) of `counter` for the seeded launches (`<G, false>`) of the kernel."""
    lines = open(path).read().splitlines()
    for i, ln in enumerate(lines):
        if ln.startswith(kernel_prefix) and "false>" in ln:
            n = int(re.search(r"dispatches=(\d+)", ln).group(1))
            for nxt in lines[i + 1:]:
                if not nxt.startswith(" "):
                    break
                if nxt.split()[0] == counter:
                    return float(re.search(r"per dispatch ([0-9.]+)", nxt).group(1)), n
    raise SystemExit("%s: no %s line for %s<.., false>" % (path, counter, kernel_prefix))


def per_dispatch_any(path, kernel_prefix, counter):
    """(value per dispatch, dispatches) of `counter` summed over every instantiation line of the kernel."""
    lines = open(path).read().splitlines()
    tot, nd = 0.0, 0
    for i, ln in enumerate(lines):
        if ln.startswith(kernel_prefix):
            n = int(re.search(r"dispatches=(\d+)", ln).group(1))
            for nxt in lines[i + 1:]:
                if not nxt.startswith(" "):
                    break
                if nxt.split()[0] == counter:
                    tot += float(re.search(r"per dispatch ([0-9.]+)", nxt).group(1)) * n
                    nd += n
    if not nd:
        raise SystemExit("%s: no %s line for %s" % (path, counter, kernel_prefix))
    return tot / nd, nd


def main_batch(argv):
    """--batch fetch.txt write.txt batch n iters: the multi-registration kernel (icp_amd/csrc/kernels.hip)"""
    fetch_txt, write_txt = argv[0], argv[1]
    batch, n, iters = (int(x) for x in argv[2:5])
    src = os.path.join("icp_amd", "csrc", "kernels.hip")
    fetch_kb, nd = per_dispatch_any(fetch_txt, "icpk::batch_icp_kernel", "FETCH_SIZE")
    write_kb, _ = per_dispatch_any(write_txt, "icpk::batch_icp_kernel", "WRITE_SIZE")
    path = os.path.join(ROOT, "icp_amd", "measured_traffic.json")
    rec = json.load(open(path))
    rec["batch_icp_kernel"] = {
        "n": n, "m": n, "n_gpus": 1, "batch": batch, "iters": iters, "fetch_size_kb": fetch_kb, "write_size_kb": write_kb,
        "traffic_bytes": int(round((2.0 * fetch_kb + write_kb) * 1024)),
        "source": "%s, %s (%d dispatches)" % (fetch_txt, write_txt, nd),
        "kernel_source": src, "kernel_source_blob": blob_hash(os.path.join(ROOT, src))}
    json.dump(rec, open(path, "w"), indent=2)
    print(json.dumps(rec["batch_icp_kernel"], indent=2))


def main_grid(argv):
    """--grid fetch.txt write.txt n: the settled grid search (icp_amd/csrc/grid.hip: search_grid_settled_kernel)"""
    fetch_txt, write_txt, n = argv[0], argv[1], int(argv[2])
    src = os.path.join("icp_amd", "csrc", "grid.hip")
    fetch_kb, nd = per_dispatch_any(fetch_txt, "icpk::search_grid_settled_kernel", "FETCH_SIZE")
    write_kb, _ = per_dispatch_any(write_txt, "icpk::search_grid_settled_kernel", "WRITE_SIZE")
    path = os.path.join(ROOT, "icp_amd", "measured_traffic.json")
    rec = json.load(open(path))
    rec["search_grid_settled_kernel"] = {
        "n": n, "m": n, "n_gpus": 1, "fetch_size_kb": fetch_kb, "write_size_kb": write_kb,
        "traffic_bytes": int(round((2.0 * fetch_kb + write_kb) * 1024)),
        "source": "%s, %s (%d dispatches)" % (fetch_txt, write_txt, nd),
        "kernel_source": src, "kernel_source_blob": blob_hash(os.path.join(ROOT, src))}
    json.dump(rec, open(path, "w"), indent=2)
    print(json.dumps(rec["search_grid_settled_kernel"], indent=2))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--batch":
        return main_batch(sys.argv[2:])
    if len(sys.argv) > 1 and sys.argv[1] == "--grid":
        return main_grid(sys.argv[2:])
    fetch_txt, write_txt = sys.argv[1], sys.argv[2]
    n, m, g = (int(x) for x in sys.argv[3:6]) if len(sys.argv) >= 6 else (1000000, 1000000, 1)
    fetch_kb, nd = per_dispatch(fetch_txt, "icpk::search_mfma16_kernel", "FETCH_SIZE")
    write_kb, _ = per_dispatch(write_txt, "icpk::search_mfma16_kernel", "WRITE_SIZE")
    path = os.path.join(ROOT, "icp_amd", "measured_traffic.json")
    rec = json.load(open(path))
    rec["search_mfma16_kernel"] = {
        "n": n, "m": m, "n_gpus": g, "fetch_size_kb": fetch_kb, "write_size_kb": write_kb,
        # FETCH_SIZE counts 64 B per 128-B request of a wide streaming read on gfx950: doubled (MI355X_MICROARCH.md, HBM)
        "traffic_bytes": int(round((2.0 * fetch_kb + write_kb) * 1024)),
        "source": "%s, %s (%d seeded dispatches)" % (fetch_txt, write_txt, nd),
        "kernel_source": KERNEL_SOURCE, "kernel_source_blob": blob_hash(os.path.join(ROOT, KERNEL_SOURCE))}
    json.dump(rec, open(path, "w"), indent=2)
    print(json.dumps(rec["search_mfma16_kernel"], indent=2))


if __name__ == "__main__":
    main()
