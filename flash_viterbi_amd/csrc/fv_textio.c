/*
 * fv_textio.c — host-side text I/O for the generate_data file format.
 *
 * The reference programs read their model from four whitespace-separated text
 * files written by numpy.savetxt(fmt='%.16f') (reference
 * generate_data/data_script.py:98-101) and parse every value with
 * fscanf("%f") straight into a float (reference
 * src/FLASH_Viterbi_multithread.c:82-93).  The float a decoder sees is
 * therefore strtof() of the 16-decimal text, NOT (float) of the generator's
 * double.  Everything here reproduces that chain so that a model generated
 * in memory is bit-identical to one that went through the files.
 *
 * Plain C, no GPU: built into libfvhost.so by gcc.
 */
#define _POSIX_C_SOURCE 200809L
#include <errno.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>

#include "flashvit_host.h"

#define FVH_IOBUF (1u << 22)

/* "%.16f" of one double into buf, returns length.  Exact zeros dominate a
 * generate_data transition matrix (1-p of all entries), so they take a
 * memcpy fast path. */
static inline int fmt16(char *buf, double v)
{
    if (v == 0.0 && !(1.0 / v < 0)) {
        memcpy(buf, "0.0000000000000000", 18);
        return 18;
    }
    return snprintf(buf, 64, "%.16f", v);
}

int fvh_quantize_text16(const double *in, float *out, size_t n)
{
    char buf[512];
    for (size_t i = 0; i < n; ++i) {
        double v = in[i];
        if (v == 0.0) { out[i] = 0.0f; continue; }
        int len = snprintf(buf, sizeof buf, "%.16f", v);
        if (len <= 0 || len >= (int)sizeof buf) return FVH_ERR_FORMAT;
        out[i] = strtof(buf, NULL);
    }
    return 0;
}

int fvh_write_matrix_text16(const char *path, const double *a, size_t rows, size_t cols,
                            int row_newline)
{
    return fvh_write_matrix_text16_ex(path, a, rows, cols, row_newline, 0);
}

int fvh_write_matrix_text16_ex(const char *path, const double *a, size_t rows, size_t cols,
                               int row_newline, int append)
{
    FILE *fp = fopen(path, append ? "ab" : "wb");
    if (!fp) return FVH_ERR_OPEN;
    char *io = (char *)malloc(FVH_IOBUF);
    if (!io) { fclose(fp); return FVH_ERR_NOMEM; }
    setvbuf(fp, io, _IOFBF, FVH_IOBUF);
    char buf[512];
    int rc = 0;
    for (size_t r = 0; r < rows && !rc; ++r) {
        for (size_t c = 0; c < cols; ++c) {
            int len = fmt16(buf, a[r * cols + c]);
            if (c + 1 < cols) buf[len++] = ' ';
            if (fwrite(buf, 1, (size_t)len, fp) != (size_t)len) { rc = FVH_ERR_WRITE; break; }
        }
        /* numpy.savetxt terminates every row with `newline`; the generator
         * uses '\n' for matrices and ' ' for the 1-D Pi / ob vectors. */
        if (!rc && fputc(row_newline ? '\n' : ' ', fp) == EOF) rc = FVH_ERR_WRITE;
    }
    if (fclose(fp) != 0 && !rc) rc = FVH_ERR_WRITE;
    free(io);
    return rc;
}

int fvh_write_ints_text(const char *path, const int *v, size_t n)
{
    FILE *fp = fopen(path, "wb");
    if (!fp) return FVH_ERR_OPEN;
    int rc = 0;
    for (size_t i = 0; i < n; ++i)
        if (fprintf(fp, "%d ", v[i]) < 0) { rc = FVH_ERR_WRITE; break; }
    if (fclose(fp) != 0 && !rc) rc = FVH_ERR_WRITE;
    return rc;
}

/* Slurp a whole file; the largest BASELINE text input (A at K=3965) is 299 MB. */
static char *slurp(const char *path, size_t *len_out, int *rc)
{
    FILE *fp = fopen(path, "rb");
    if (!fp) { *rc = FVH_ERR_OPEN; return NULL; }
    if (fseek(fp, 0, SEEK_END) != 0) { fclose(fp); *rc = FVH_ERR_READ; return NULL; }
    long sz = ftell(fp);
    if (sz < 0) { fclose(fp); *rc = FVH_ERR_READ; return NULL; }
    rewind(fp);
    char *buf = (char *)malloc((size_t)sz + 1);
    if (!buf) { fclose(fp); *rc = FVH_ERR_NOMEM; return NULL; }
    size_t got = fread(buf, 1, (size_t)sz, fp);
    fclose(fp);
    if (got != (size_t)sz) { free(buf); *rc = FVH_ERR_READ; return NULL; }
    buf[sz] = '\0';
    *len_out = (size_t)sz;
    *rc = 0;
    return buf;
}

int fvh_read_floats_text(const char *path, float *out, size_t n)
{
    int rc;
    size_t len;
    char *buf = slurp(path, &len, &rc);
    if (!buf) return rc;
    char *p = buf;
    size_t i = 0;
    for (; i < n; ++i) {
        char *end;
        errno = 0;
        float v = strtof(p, &end);   /* same conversion fscanf("%f") performs */
        if (end == p) break;
        out[i] = v;
        p = end;
    }
    free(buf);
    return i == n ? 0 : FVH_ERR_SHORT;
}

int fvh_read_ints_text(const char *path, int *out, size_t n)
{
    int rc;
    size_t len;
    char *buf = slurp(path, &len, &rc);
    if (!buf) return rc;
    char *p = buf;
    size_t i = 0;
    for (; i < n; ++i) {
        char *end;
        long v = strtol(p, &end, 10);
        if (end == p) break;
        out[i] = (int)v;
        p = end;
    }
    free(buf);
    return i == n ? 0 : FVH_ERR_SHORT;
}

/* Raw little-endian f32 / i32 cache files ("binary model cache"):
 * 16-byte header {magic, dtype, rows, cols} then the payload. */
#define FVH_MAGIC 0x31425646u  /* "FVB1" */
#define FVH_MAGIC2 0x32425646u /* "FVB2": + {uint64 size, uint64 mtime in ns} of the text file the payload was parsed from */

int fvh_write_bin(const char *path, const void *data, uint32_t dtype, uint32_t rows, uint32_t cols)
{
    FILE *fp = fopen(path, "wb");
    if (!fp) return FVH_ERR_OPEN;
    uint32_t hdr[4] = { FVH_MAGIC, dtype, rows, cols };
    size_t n = (size_t)rows * cols;
    int rc = 0;
    if (fwrite(hdr, sizeof hdr, 1, fp) != 1) rc = FVH_ERR_WRITE;
    if (!rc && n && fwrite(data, 4, n, fp) != n) rc = FVH_ERR_WRITE;
    if (fclose(fp) != 0 && !rc) rc = FVH_ERR_WRITE;
    return rc;
}

/* Header of either version; *src_size / *src_mtime_ns stay 0 for an unbound (FVB1) file. */
static int read_bin_header(FILE *fp, uint32_t dtype, uint32_t rows, uint32_t cols, uint64_t *src_size, uint64_t *src_mtime_ns)
{
    uint32_t hdr[4];
    *src_size = *src_mtime_ns = 0;
    if (fread(hdr, sizeof hdr, 1, fp) != 1) return FVH_ERR_READ;
    if ((hdr[0] != FVH_MAGIC && hdr[0] != FVH_MAGIC2) || hdr[1] != dtype || hdr[2] != rows || hdr[3] != cols)
        return FVH_ERR_FORMAT;
    if (hdr[0] == FVH_MAGIC2) {
        uint64_t src[2];
        if (fread(src, sizeof src, 1, fp) != 1) return FVH_ERR_READ;
        *src_size = src[0]; *src_mtime_ns = src[1];
    }
    return 0;
}

int fvh_read_bin(const char *path, void *data, uint32_t dtype, uint32_t rows, uint32_t cols)
{
    FILE *fp = fopen(path, "rb");
    if (!fp) return FVH_ERR_OPEN;
    size_t n = (size_t)rows * cols;
    uint64_t a, b;
    int rc = read_bin_header(fp, dtype, rows, cols, &a, &b);
    if (!rc && n && fread(data, 4, n, fp) != n) rc = FVH_ERR_SHORT;
    fclose(fp);
    return rc;
}

/* Size and modification time of the text file a cache was parsed from (0, 0 when it does not exist). */
static int stat_source(const char *src, uint64_t *size, uint64_t *mtime_ns)
{
    struct stat sb;
    *size = *mtime_ns = 0;
    if (!src || stat(src, &sb) != 0) return -1;
    *size = (uint64_t)sb.st_size;
    *mtime_ns = (uint64_t)sb.st_mtim.tv_sec * 1000000000ull + (uint64_t)sb.st_mtim.tv_nsec;
    return 0;
}

int fvh_write_bin_src(const char *path, const void *data, uint32_t dtype, uint32_t rows, uint32_t cols,
                      const char *src_text_path)
{
    FILE *fp = fopen(path, "wb");
    if (!fp) return FVH_ERR_OPEN;
    uint32_t hdr[4] = { FVH_MAGIC2, dtype, rows, cols };
    uint64_t src[2];
    (void)stat_source(src_text_path, &src[0], &src[1]);
    size_t n = (size_t)rows * cols;
    int rc = 0;
    if (fwrite(hdr, sizeof hdr, 1, fp) != 1 || fwrite(src, sizeof src, 1, fp) != 1) rc = FVH_ERR_WRITE;
    if (!rc && n && fwrite(data, 4, n, fp) != n) rc = FVH_ERR_WRITE;
    if (fclose(fp) != 0 && !rc) rc = FVH_ERR_WRITE;
    return rc;
}

int fvh_read_bin_src(const char *path, void *data, uint32_t dtype, uint32_t rows, uint32_t cols,
                     const char *src_text_path)
{
    FILE *fp = fopen(path, "rb");
    if (!fp) return FVH_ERR_OPEN;
    size_t n = (size_t)rows * cols;
    uint64_t have_size, have_mtime, want_size, want_mtime;
    int rc = read_bin_header(fp, dtype, rows, cols, &have_size, &have_mtime);
    if (!rc && stat_source(src_text_path, &want_size, &want_mtime) == 0 &&
        (have_size != want_size || have_mtime != want_mtime))
        rc = FVH_ERR_STALE;          /* the text exists and is not the file this cache was parsed from */
    if (!rc && n && fread(data, 4, n, fp) != n) rc = FVH_ERR_SHORT;
    fclose(fp);
    return rc;
}

const char *fvh_strerror(int rc)
{
    switch (rc) {
    case 0: return "ok";
    case FVH_ERR_OPEN: return "cannot open file";
    case FVH_ERR_READ: return "read error";
    case FVH_ERR_WRITE: return "write error";
    case FVH_ERR_SHORT: return "file holds fewer values than requested";
    case FVH_ERR_FORMAT: return "bad format";
    case FVH_ERR_NOMEM: return "out of host memory";
    case FVH_ERR_STALE: return "cache does not belong to the text file next to it";
    default: return "unknown fvh error";
    }
}
