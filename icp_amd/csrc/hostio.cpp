// icp_amd/csrc/hostio.cpp -- host-only pieces of the C ABI kept from the reference: the CSV point-cloud
// loader, the result writer and the final distance report.  Written from the behaviour of
// src/libCSV/csv.cpp:94-158 and src/libalg/CPUMatrix.cpp:81-94,291-295,323-327 (reference tree), not from
// their text.
#include "../../include/icp_amd.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <thread>
#include <vector>

// The library's thread-local error text lives here (host-only object); engine.hip reports through
// icp_amd_fail_().
#include <stdarg.h>
static thread_local char g_err[512] = "";
extern "C" const char *icp_amd_last_error(void) { return g_err; }
extern "C" int icp_amd_fail_(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

static int io_fail(int code, const std::string &msg) { return icp_amd_fail_(code, "%s", msg.c_str()); }

// Behaviour kept from readCSV: the first line is a header iff its first field is not a number (strtod
// consumes nothing); the column count is the field count of the first line; fields are separated by ','
// with empty fields skipped (strtok semantics); a field counts as a number as soon as strtod consumes a
// prefix of it; a non-numeric field after the first line -- including the field of a trailing blank line --
// is the error "Not a valid double !".  Two quirks of readCSV are kept as well, because they decide what a caller gets
// from a well-formed file (tests/fuzz/fuzz_csv.py compares against the reference's own reader, file by file):
//  * a file WITHOUT a header line and with two or more lines loses its last row: the reference stores the second line as
//    the "header" text once the first line turned out to be numeric (csv.cpp:113-114,128-129) and then counts one line
//    less (csv.cpp:147-148).  nblines, the values and the header text returned here are the reference's;
//  * a file without a single number (empty, or a header only) is the error "Alloc error (realloc) !" (csv.cpp:151-156,
//    realloc to size 0).
// Unlike the reference, files it would turn into an array whose size disagrees with nblines x nbcols (ragged rows, a
// first line that mixes numbers and text) are rejected.
static int finish_csv(std::vector<double> &&vals, size_t lines, size_t cols, bool has_header, const std::string &second_line,
                      double **out, size_t *nblines, size_t *nbcols, char *header, size_t header_cap)
{
    if (vals.empty()) return io_fail(ICP_AMD_EIO, "Alloc error (realloc) !");
    size_t rows = has_header ? lines - 1 : lines;
    if (vals.size() != rows * cols) return io_fail(ICP_AMD_EIO, "CSV size mismatch");
    if (!has_header && lines >= 2) {   // the reference's header bookkeeping: the second line is reported as the header, the last row is lost
        --rows;
        vals.resize(rows * cols);
        if (header && header_cap) snprintf(header, header_cap, "%s", second_line.c_str());
    }
    double *r = (double *)malloc(vals.size() * sizeof(double));
    if (!r) return io_fail(ICP_AMD_ENOMEM, "Alloc error !");
    memcpy(r, vals.data(), vals.size() * sizeof(double));
    *out = r;
    *nblines = rows;
    *nbcols = cols;
    return ICP_AMD_OK;
}

extern "C" int icp_amd_read_csv(const char *path, double **out, size_t *nblines, size_t *nbcols, char *header, size_t header_cap)
{
    if (!path || !out || !nblines || !nbcols) return io_fail(ICP_AMD_EINVAL, "icp_amd_read_csv: NULL argument");
    *out = nullptr;
    *nblines = *nbcols = 0;
    if (header && header_cap) header[0] = '\0';
    FILE *f = fopen(path, "r");
    if (!f) return io_fail(ICP_AMD_EIO, "File could not be opened !");
    std::vector<double> vals;
    char *line = nullptr;
    size_t cap = 0, lines = 0, cols = 0;
    bool has_header = false;
    std::string second;
    int rc = ICP_AMD_OK;
    ssize_t len;
    while (rc == ICP_AMD_OK && (len = getline(&line, &cap, f)) != -1) {
        size_t fields = 0, numeric = 0;
        const std::string raw(line, (size_t)len);
        if (lines == 1) second = raw;
        char *cur = line;
        const bool first = (lines == 0);
        while (*cur) {
            // split on ',' ; '\n' also ends a field except while looking for the very first field of a line
            const char *delims = (fields == 0) ? "," : ",\n";
            cur += strspn(cur, delims);
            if (!*cur) break;
            char *end = cur + strcspn(cur, delims);
            const char saved = *end;
            *end = '\0';
            ++fields;
            char *stop = nullptr;
            const double v = strtod(cur, &stop);
            if (stop == cur) {
                if (!first) { rc = io_fail(ICP_AMD_EIO, "Not a valid double !"); break; }
            } else {
                vals.push_back(v);
                ++numeric;
            }
            *end = saved;
            cur = end;
        }
        if (rc != ICP_AMD_OK) break;
        if (first) {
            cols = fields;
            if (numeric == 0 && fields > 0) {
                has_header = true;
                if (header && header_cap) snprintf(header, header_cap, "%s", raw.c_str());
            } else if (numeric != fields) {
                rc = io_fail(ICP_AMD_EIO, "Not a valid double !");
            }
        } else if (numeric != cols) {
            rc = io_fail(ICP_AMD_EIO, "CSV line " + std::to_string(lines + 1) + " has " + std::to_string(numeric) +
                                          " fields, expected " + std::to_string(cols));
        }
        ++lines;
    }
    free(line);
    fclose(f);
    if (rc != ICP_AMD_OK) return rc;
    return finish_csv(std::move(vals), lines, cols, has_header, second, out, nblines, nbcols, header, header_cap);
}

extern "C" void icp_amd_free(void *p) { free(p); }

// ------------------------------------------------------------------------------------------------
// f-3: the same CSV semantics parsed by several threads, and binary clouds (.f64 / .npy)
// ------------------------------------------------------------------------------------------------
namespace {

// One line [b, e) of the file buffer (e points behind its '\n', or at the end of the buffer, which is NUL
// terminated).  Same rules as the loop of icp_amd_read_csv; the buffer is not modified, so a conversion only counts
// when it ends inside its field (strtod skips leading blanks, including the newline of a blank field).
struct LineResult { size_t fields, numeric; bool bad_number; };
LineResult parse_line(const char *b, const char *e, bool first, std::vector<double> &vals)
{
    LineResult r{0, 0, false};
    const char *cur = b;
    while (cur < e) {
        const bool more = r.fields != 0;   // '\n' ends a field except while looking for the first field of a line
        while (cur < e && (*cur == ',' || (more && *cur == '\n'))) ++cur;
        if (cur >= e) break;
        const char *end = cur;
        while (end < e && *end != ',' && !(more && *end == '\n')) ++end;
        ++r.fields;
        char *stop = nullptr;
        const double v = strtod(cur, &stop);
        if (stop == cur || stop > end) {
            if (!first) { r.bad_number = true; return r; }
        } else {
            vals.push_back(v);
            ++r.numeric;
        }
        cur = end;
    }
    return r;
}

struct ChunkResult {
    std::vector<double> vals;
    size_t lines = 0;
    int err = 0;             // 0 none, 1 "Not a valid double !", 2 ragged
    size_t err_line = 0;     // chunk-relative
    size_t err_numeric = 0;
};

void parse_chunk(const char *b, const char *e, size_t cols, ChunkResult *out)
{
    out->vals.reserve((size_t)(e - b) / 8);
    while (b < e) {
        const char *nl = (const char *)memchr(b, '\n', (size_t)(e - b));
        const char *le = nl ? nl + 1 : e;
        const LineResult r = parse_line(b, le, false, out->vals);
        if (r.bad_number) { out->err = 1; out->err_line = out->lines; return; }
        if (r.numeric != cols) { out->err = 2; out->err_line = out->lines; out->err_numeric = r.numeric; return; }
        ++out->lines;
        b = le;
    }
}

int read_whole(const char *path, std::vector<char> &buf)
{
    FILE *f = fopen(path, "rb");
    if (!f) return io_fail(ICP_AMD_EIO, "File could not be opened !");
    if (fseek(f, 0, SEEK_END) != 0) { fclose(f); return io_fail(ICP_AMD_EIO, "File could not be read !"); }
    const long sz = ftell(f);
    rewind(f);
    if (sz < 0) { fclose(f); return io_fail(ICP_AMD_EIO, "File could not be read !"); }
    buf.resize((size_t)sz + 1);
    const size_t got = fread(buf.data(), 1, (size_t)sz, f);
    fclose(f);
    if (got != (size_t)sz) return io_fail(ICP_AMD_EIO, "File could not be read !");
    buf[(size_t)sz] = '\0';
    return ICP_AMD_OK;
}

bool ends_with(const char *s, const char *suffix)
{
    const size_t n = strlen(s), k = strlen(suffix);
    return n >= k && strcmp(s + n - k, suffix) == 0;
}

}  // namespace

// Same result and same errors as icp_amd_read_csv (tests compare them file by file); the lines after the first are
// parsed by `threads` threads on newline-aligned pieces of the file (0 = one per core, at most 16).
extern "C" int icp_amd_read_csv_mt(const char *path, int threads, double **out, size_t *nblines, size_t *nbcols, char *header,
                                   size_t header_cap)
{
    if (!path || !out || !nblines || !nbcols) return io_fail(ICP_AMD_EINVAL, "icp_amd_read_csv_mt: NULL argument");
    *out = nullptr;
    *nblines = *nbcols = 0;
    if (header && header_cap) header[0] = '\0';
    std::vector<char> buf;
    const int rc = read_whole(path, buf);
    if (rc != ICP_AMD_OK) return rc;
    const char *b = buf.data(), *e = b + buf.size() - 1;
    // getline() semantics end at an embedded NUL as far as strtok/strtod are concerned; such files take the serial path
    if (memchr(b, '\0', (size_t)(e - b))) return icp_amd_read_csv(path, out, nblines, nbcols, header, header_cap);
    if (b == e) return io_fail(ICP_AMD_EIO, "Alloc error (realloc) !");
    // first line: header detection and the column count
    const char *nl = (const char *)memchr(b, '\n', (size_t)(e - b));
    const char *le = nl ? nl + 1 : e;
    std::vector<double> head;
    const LineResult fr = parse_line(b, le, true, head);
    const size_t cols = fr.fields;
    bool has_header = false;
    if (fr.numeric == 0 && fr.fields > 0) {
        has_header = true;
        if (header && header_cap) snprintf(header, header_cap, "%.*s", (int)(le - b), b);
    } else if (fr.numeric != fr.fields) {
        return io_fail(ICP_AMD_EIO, "Not a valid double !");
    }
    if (threads <= 0) threads = (int)std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 16u);
    const size_t body = (size_t)(e - le);
    if ((size_t)threads > body / 65536 + 1) threads = (int)(body / 65536 + 1);
    std::vector<const char *> cut((size_t)threads + 1);
    cut[0] = le;
    cut[(size_t)threads] = e;
    for (int t = 1; t < threads; ++t) {
        const char *p = le + body / (size_t)threads * (size_t)t;
        if (p < cut[(size_t)t - 1]) p = cut[(size_t)t - 1];
        const char *q = (const char *)memchr(p, '\n', (size_t)(e - p));
        cut[(size_t)t] = q ? q + 1 : e;
    }
    std::vector<ChunkResult> res((size_t)threads);
    std::vector<std::thread> pool;
    for (int t = 1; t < threads; ++t) pool.emplace_back(parse_chunk, cut[(size_t)t], cut[(size_t)t + 1], cols, &res[(size_t)t]);
    parse_chunk(cut[0], cut[1], cols, &res[0]);
    for (std::thread &th : pool) th.join();
    size_t lines = 1, total = head.size();
    for (const ChunkResult &c : res) {
        if (c.err == 1) return io_fail(ICP_AMD_EIO, "Not a valid double !");
        if (c.err == 2)
            return io_fail(ICP_AMD_EIO, "CSV line " + std::to_string(lines + c.err_line + 1) + " has " +
                                            std::to_string(c.err_numeric) + " fields, expected " + std::to_string(cols));
        lines += c.lines;
        total += c.vals.size();
    }
    std::vector<double> vals;
    vals.reserve(total);
    vals.insert(vals.end(), head.begin(), head.end());
    for (ChunkResult &c : res) vals.insert(vals.end(), c.vals.begin(), c.vals.end());
    std::string second;
    if (!has_header && le < e) {
        const char *nl2 = (const char *)memchr(le, '\n', (size_t)(e - le));
        second.assign(le, nl2 ? nl2 + 1 : e);
    }
    return finish_csv(std::move(vals), lines, cols, has_header, second, out, nblines, nbcols, header, header_cap);
}

// Raw little-endian fp64, row-major, `cols` columns (the GPUICP clouds have 3).
static int read_f64(const char *path, size_t cols, double **out, size_t *nblines, size_t *nbcols)
{
    FILE *f = fopen(path, "rb");
    if (!f) return io_fail(ICP_AMD_EIO, "File could not be opened !");
    fseek(f, 0, SEEK_END);
    const long sz = ftell(f);
    rewind(f);
    if (sz < 0 || (size_t)sz % (cols * sizeof(double)) != 0) {
        fclose(f);
        return io_fail(ICP_AMD_EIO, "raw fp64 cloud: file size is not a multiple of " + std::to_string(cols * sizeof(double)) + " bytes");
    }
    double *r = (double *)malloc(sz ? (size_t)sz : 1);
    if (!r) { fclose(f); return io_fail(ICP_AMD_ENOMEM, "Alloc error !"); }
    const size_t got = fread(r, 1, (size_t)sz, f);
    fclose(f);
    if (got != (size_t)sz) { free(r); return io_fail(ICP_AMD_EIO, "File could not be read !"); }
    *out = r;
    *nblines = (size_t)sz / (cols * sizeof(double));
    *nbcols = cols;
    return ICP_AMD_OK;
}

// NumPy .npy, format 1.0 - 3.0: little-endian fp64 ('<f8'), C order, shape (rows, cols) or (rows,).
static int read_npy(const char *path, double **out, size_t *nblines, size_t *nbcols)
{
    FILE *f = fopen(path, "rb");
    if (!f) return io_fail(ICP_AMD_EIO, "File could not be opened !");
    unsigned char pre[12];
    int rc = ICP_AMD_OK;
    std::string hdr;
    size_t rows = 0, cols = 1;
    do {
        if (fread(pre, 1, 10, f) != 10 || memcmp(pre, "\x93NUMPY", 6) != 0) { rc = io_fail(ICP_AMD_EIO, "not a .npy file"); break; }
        size_t hlen = pre[8] | ((size_t)pre[9] << 8);
        if (pre[6] >= 2) {
            if (fread(pre + 10, 1, 2, f) != 2) { rc = io_fail(ICP_AMD_EIO, "truncated .npy header"); break; }
            hlen |= ((size_t)pre[10] << 16) | ((size_t)pre[11] << 24);
        }
        if (hlen > (1u << 20)) { rc = io_fail(ICP_AMD_EIO, "unreasonable .npy header"); break; }
        hdr.resize(hlen);
        if (fread(&hdr[0], 1, hlen, f) != hlen) { rc = io_fail(ICP_AMD_EIO, "truncated .npy header"); break; }
        if (hdr.find("'<f8'") == std::string::npos && hdr.find("'=f8'") == std::string::npos) {
            rc = io_fail(ICP_AMD_EIO, ".npy cloud must be little-endian float64 ('<f8')");
            break;
        }
        if (hdr.find("'fortran_order': False") == std::string::npos) { rc = io_fail(ICP_AMD_EIO, ".npy cloud must be C-ordered"); break; }
        const size_t sp = hdr.find("'shape':");
        const size_t lp = sp == std::string::npos ? sp : hdr.find('(', sp), rp = lp == std::string::npos ? lp : hdr.find(')', lp);
        if (rp == std::string::npos) { rc = io_fail(ICP_AMD_EIO, "bad .npy shape"); break; }
        std::vector<size_t> dims;
        const char *c = hdr.c_str() + lp + 1, *ce = hdr.c_str() + rp;
        while (c < ce) {
            char *stop = nullptr;
            const unsigned long long v = strtoull(c, &stop, 10);
            if (stop == c) { ++c; continue; }
            dims.push_back((size_t)v);
            c = stop;
        }
        if (dims.empty() || dims.size() > 2) { rc = io_fail(ICP_AMD_EIO, ".npy cloud must be 1- or 2-dimensional"); break; }
        rows = dims[0];
        cols = dims.size() == 2 ? dims[1] : 1;
        if (cols != 0 && rows > ((size_t)-1) / sizeof(double) / cols) { rc = io_fail(ICP_AMD_EIO, ".npy shape overflows"); break; }
        const size_t bytes = rows * cols * sizeof(double);
        double *r = (double *)malloc(bytes ? bytes : 1);
        if (!r) { rc = io_fail(ICP_AMD_ENOMEM, "Alloc error !"); break; }
        if (fread(r, 1, bytes, f) != bytes) { free(r); rc = io_fail(ICP_AMD_EIO, "truncated .npy data"); break; }
        *out = r;
        *nblines = rows;
        *nbcols = cols;
    } while (0);
    fclose(f);
    return rc;
}

// Point cloud by file name: *.npy, *.f64 (raw little-endian fp64 x 3 columns), anything else = CSV with the reference's
// semantics (parsed in parallel when the file is large).
extern "C" int icp_amd_read_cloud(const char *path, double **out, size_t *nblines, size_t *nbcols, char *header, size_t header_cap)
{
    if (!path || !out || !nblines || !nbcols) return io_fail(ICP_AMD_EINVAL, "icp_amd_read_cloud: NULL argument");
    *out = nullptr;
    *nblines = *nbcols = 0;
    if (header && header_cap) header[0] = '\0';
    if (ends_with(path, ".npy")) return read_npy(path, out, nblines, nbcols);
    if (ends_with(path, ".f64")) return read_f64(path, 3, out, nblines, nbcols);
    const char *tv = getenv("ICP_AMD_CSV_THREADS");
    const int threads = tv ? atoi(tv) : 0;
    if (threads == 1) return icp_amd_read_csv(path, out, nblines, nbcols, header, header_cap);
    return icp_amd_read_csv_mt(path, threads, out, nblines, nbcols, header, header_cap);
}

// Binary result writer, by file name like icp_amd_read_cloud (*.npy or *.f64); full fp64 precision, unlike the
// six-digit text of icp_amd_write_cloud.
extern "C" int icp_amd_write_cloud_binary(const char *path, const double *A, size_t rows, size_t cols)
{
    if (!path || (!A && rows != 0 && cols != 0)) return io_fail(ICP_AMD_EINVAL, "icp_amd_write_cloud_binary: NULL argument");
    const bool npy = ends_with(path, ".npy");
    if (!npy && !ends_with(path, ".f64")) return io_fail(ICP_AMD_EINVAL, "binary clouds are written as *.npy or *.f64");
    FILE *f = fopen(path, "wb");
    if (!f) return io_fail(ICP_AMD_EIO, "File could not be opened !");
    if (npy) {
        char dict[128];
        int len = snprintf(dict, sizeof dict, "{'descr': '<f8', 'fortran_order': False, 'shape': (%zu, %zu), }", rows, cols);
        const int pad = 64 - (10 + len + 1) % 64;
        const unsigned hlen = (unsigned)(len + pad % 64 + 1);
        const unsigned char pre[10] = {0x93, 'N', 'U', 'M', 'P', 'Y', 1, 0, (unsigned char)(hlen & 255), (unsigned char)(hlen >> 8)};
        fwrite(pre, 1, 10, f);
        fwrite(dict, 1, (size_t)len, f);
        for (int k = 0; k < pad % 64; ++k) fputc(' ', f);
        fputc('\n', f);
    }
    const size_t count = rows * cols;
    if (count && fwrite(A, sizeof(double), count, f) != count) { fclose(f); return io_fail(ICP_AMD_EIO, "write failed"); }
    if (fclose(f) != 0) return io_fail(ICP_AMD_EIO, "write failed");
    return ICP_AMD_OK;
}

// " dim0: N dim1: C array: " then one line per row, every value followed by a blank, default ostream
// formatting (= %g, 6 significant digits).
static void format_rows(const double *A, size_t r0, size_t r1, size_t cols, std::string *out)
{
    char buf[64];
    out->reserve((r1 - r0) * (cols * 12 + 1));
    for (size_t i = r0; i < r1; ++i) {
        for (size_t j = 0; j < cols; ++j) {
            const int k = snprintf(buf, sizeof buf, "%g ", A[i * cols + j]);
            out->append(buf, (size_t)k);
        }
        out->push_back('\n');
    }
}

extern "C" int icp_amd_write_cloud(const char *path, const double *A, size_t rows, size_t cols)
{
    if (!path || (!A && rows != 0 && cols != 0)) return io_fail(ICP_AMD_EINVAL, "icp_amd_write_cloud: NULL argument");
    FILE *f = fopen(path, "w");
    if (!f) return io_fail(ICP_AMD_EIO, "File could not be opened !");
    fprintf(f, " dim0: %zu dim1: %zu array: \n", rows, cols);
    // the rows are formatted by several threads (same snprintf, so the same bytes) and written in order
    size_t threads = std::min<size_t>(std::max(1u, std::thread::hardware_concurrency()), 16);
    if (rows * cols < 65536) threads = 1;
    std::vector<std::string> part(threads);
    std::vector<std::thread> pool;
    const size_t per = (rows + threads - 1) / threads;
    for (size_t t = 1; t < threads; ++t) {
        const size_t r0 = std::min(rows, t * per), r1 = std::min(rows, r0 + per);
        pool.emplace_back(format_rows, A, r0, r1, cols, &part[t]);
    }
    format_rows(A, 0, std::min(rows, per), cols, &part[0]);
    for (std::thread &th : pool) th.join();
    bool ok = true;
    for (const std::string &p : part) ok = ok && (p.empty() || fwrite(p.data(), 1, p.size(), f) == p.size());
    if (fclose(f) != 0 || !ok) return io_fail(ICP_AMD_EIO, "write failed");
    return ICP_AMD_OK;
}

// sqrt( sum_rows ( sum_cols (a-b)^2 ) ), each row summed on its own first (element_wise_reduce order)
extern "C" double icp_amd_euclidian_distance(const double *A, const double *B, size_t rows, size_t cols)
{
    double r = 0;
    for (size_t i = 0; i < rows; ++i) {
        double tmp = 0;
        for (size_t j = 0; j < cols; ++j) {
            const double d = A[i * cols + j] - B[i * cols + j];
            tmp = tmp + d * d;
        }
        r = r + tmp;
    }
    return sqrt(r);
}
