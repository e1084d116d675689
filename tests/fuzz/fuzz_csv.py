#!/usr/bin/env python3
"""Differential fuzz of the CSV reader (icp_amd_read_csv / _mt / read_cloud, hostio.cpp) against the reference's own
readCSV, compiled unmodified into oracle/_ref/ref_trace (build container only: CPU, no GPU).  Random small files made of
numeric and non-numeric tokens, blanks, headers, ragged rows and both line endings; both readers must agree on
success / failure and, on success, on shape and every bit of every value.
    python tests/fuzz/fuzz_csv.py [cases] [seed]"""
import os, struct, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import icp_amd

REF = os.path.join(ROOT, "oracle", "_ref", "ref_trace")
TOKENS = ["0", "1", "-2.5", "1e3", ".5", "+7", "3.", "1e-320", "1e400", "-0.0", "0x1p3", "0x10", "nan", "inf", "-inf", "NAN", "Infinity",
          "abc", "", " ", " 3 ", "4 5", "1.5abc", "1,2".replace(",", ";"), "\t6", "7\t", "1e", "--1", "1.2.3", "12345678901234567890",
          "0.1", "2.2250738585072014e-308", "1d3", "１"]

def make_case(rng):
    cols = int(rng.choice([1, 2, 3, 3, 3, 4, 5]))
    rows = int(rng.randint(0, 6))
    eol = "\r\n" if rng.rand() < 0.2 else "\n"
    lines = []
    if rng.rand() < 0.5:
        lines.append(",".join(rng.choice(["x", "y", "Points_0", "1", "a b", ""]) for _ in range(int(rng.choice([cols, cols, cols + 1, 1])))))
    clean = rng.rand() < 0.5      # half of the files are mostly well-formed so that success paths are compared too
    for _ in range(rows):
        c = cols if rng.rand() < 0.85 else int(rng.randint(0, 7))
        toks = [("%.17g" % rng.uniform(-100, 100)) if (clean and rng.rand() < 0.9) else str(rng.choice(TOKENS)) for _ in range(c)]
        line = ",".join(toks)
        if rng.rand() < 0.05: line += ","
        if rng.rand() < 0.05: line = "," + line
        lines.append(line)
        if rng.rand() < 0.05: lines.append("")
    text = eol.join(lines)
    if lines and rng.rand() < 0.8: text += eol
    return text

def reference_read(path, qpath, tmp):
    out = os.path.join(tmp, "out.bin")
    if os.path.exists(out): os.remove(out)
    r = subprocess.run([REF, path, qpath, "0", out], capture_output=True, timeout=20)
    if r.returncode != 0: return ("err",)
    b = open(out, "rb").read()
    n = struct.unpack("<Q", b[8:16])[0]
    body = b[8 + 32 + 24:]
    vals = np.frombuffer(body, dtype=np.float64)
    cols = vals.size // n if n else 0
    return ("ok", (n, cols), vals.tobytes())

def ours_read(path):
    outs = []
    for fn in (lambda: icp_amd.read_csv(path), lambda: icp_amd.read_csv_mt(path, 3), lambda: icp_amd.read_cloud(path)):
        try:
            a, _ = fn()
            outs.append(("ok", tuple(a.shape), a.tobytes()))
        except icp_amd.IcpAmdError:
            outs.append(("err",))
    assert all(o == outs[0] for o in outs[1:]), "our readers disagree among themselves"
    return outs[0]

_libc = __import__("ctypes").CDLL(None)
_libc.strtod.restype = __import__("ctypes").c_double
_libc.strtod.argtypes = [__import__("ctypes").c_char_p, __import__("ctypes").POINTER(__import__("ctypes").c_char_p)]

def _numeric(tok):
    """strtod consumes a prefix of the token (the reference's and the library's notion of 'a number')"""
    import ctypes
    buf = ctypes.create_string_buffer(tok)
    end = ctypes.c_char_p()
    _libc.strtod(ctypes.cast(buf, ctypes.c_char_p), ctypes.byref(end))
    return ctypes.cast(end, ctypes.c_void_p).value != ctypes.addressof(buf)

def regular(data):
    """Independent model of which files the library promises to read like the reference: after an optional header line
    without a single number, every line has as many fields as the first one and every field is a number.  Fields as
    strtok sees them: split on ',', empty fields skipped, the newline is a delimiter after the first field only."""
    lines = data.split(b"\n")
    lines = [l + b"\n" for l in lines[:-1]] + ([lines[-1]] if lines[-1] else [])
    shape = []
    for line in lines:
        toks, pos, first = [], 0, True
        while True:
            delims = b"," if first else b",\n"
            while pos < len(line) and line[pos:pos + 1] in [bytes([d]) for d in delims]: pos += 1
            if pos >= len(line): break
            end = pos
            while end < len(line) and line[end:end + 1] not in [bytes([d]) for d in delims]: end += 1
            toks.append(line[pos:end]); pos = end; first = False
        shape.append((len(toks), sum(_numeric(t) for t in toks)))
    if not shape: return True
    n0, k0 = shape[0]
    if k0 not in (0, n0): return False
    return all(n == n0 and k == n0 for n, k in shape[1:])

def run(cases, seed, verbose=True):
    rng = np.random.RandomState(seed)
    bad = []
    stats = {"regular ok": 0, "regular err": 0, "irregular": 0}
    with tempfile.TemporaryDirectory() as tmp:
        qpath = os.path.join(tmp, "q.csv")
        open(qpath, "w").write("1,2,3\n4,5,6\n")
        for k in range(cases):
            text = make_case(rng)
            path = os.path.join(tmp, "p.csv")
            open(path, "wb").write(text.encode())
            ref, got = reference_read(path, qpath, tmp), ours_read(path)
            # files outside the promise (ragged rows, a first line mixing numbers and text): the reference returns an array
            # whose size disagrees with its own row count; the library must refuse them, whatever the reference did
            reg = regular(text.encode())
            stats["irregular" if not reg else "regular ok" if ref[0] == "ok" else "regular err"] += 1
            ok = (ref == got) if reg else (got == ("err",))
            if not ok:
                bad.append((text, ref[:2], got[:2]))
                if verbose: print("MISMATCH %r\n   reference %s\n   ours      %s" % (text, ref[:2], got[:2]), flush=True)
    if verbose: print(stats)
    return bad

if __name__ == "__main__":
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    bad = run(cases, seed)
    print("%d cases, %d mismatches" % (cases, len(bad)))
