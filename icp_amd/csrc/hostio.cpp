// icp_amd/csrc/hostio.cpp -- host-only pieces of the C ABI kept from the reference: the CSV point-cloud
// loader, the result writer and the final distance report.  Written from the behaviour of
// src/libCSV/csv.cpp:94-158 and src/libalg/CPUMatrix.cpp:81-94,291-295,323-327 (reference tree), not from
// their text.
#include "../../include/icp_amd.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
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
// is the error "Not a valid double !".  Unlike the reference, ragged files are rejected instead of
// producing an array whose size disagrees with nblines x nbcols.
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
    int rc = ICP_AMD_OK;
    ssize_t len;
    while (rc == ICP_AMD_OK && (len = getline(&line, &cap, f)) != -1) {
        size_t fields = 0, numeric = 0;
        const std::string raw(line, (size_t)len);
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
    const size_t rows = has_header ? lines - 1 : lines;
    if (vals.size() != rows * cols) return io_fail(ICP_AMD_EIO, "CSV size mismatch");
    double *r = (double *)malloc((vals.empty() ? 1 : vals.size()) * sizeof(double));
    if (!r) return io_fail(ICP_AMD_ENOMEM, "Alloc error !");
    if (!vals.empty()) memcpy(r, vals.data(), vals.size() * sizeof(double));
    *out = r;
    *nblines = rows;
    *nbcols = cols;
    return ICP_AMD_OK;
}

extern "C" void icp_amd_free(void *p) { free(p); }

// " dim0: N dim1: C array: " then one line per row, every value followed by a blank, default ostream
// formatting (= %g, 6 significant digits).
extern "C" int icp_amd_write_cloud(const char *path, const double *A, size_t rows, size_t cols)
{
    if (!path || (!A && rows != 0 && cols != 0)) return io_fail(ICP_AMD_EINVAL, "icp_amd_write_cloud: NULL argument");
    FILE *f = fopen(path, "w");
    if (!f) return io_fail(ICP_AMD_EIO, "File could not be opened !");
    fprintf(f, " dim0: %zu dim1: %zu array: \n", rows, cols);
    for (size_t i = 0; i < rows; ++i) {
        for (size_t j = 0; j < cols; ++j) fprintf(f, "%g ", A[i * cols + j]);
        fputc('\n', f);
    }
    if (fclose(f) != 0) return io_fail(ICP_AMD_EIO, "write failed");
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
