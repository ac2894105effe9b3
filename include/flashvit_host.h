/*
 * flashvit_host.h — C ABI of libfvhost.so: CPU-only text/binary I/O for the
 * generate_data input format.  No GPU, no torch types.
 *
 * Replaces, for the build's host programs and tools, the loader half of the
 * reference programs: getAddress / InitElement / create_vit
 * (reference src/FLASH_Viterbi_multithread.c:48-107, identical in
 * src/FLASH_BS_Viterbi_multithread.c:217-276), and the four np.savetxt calls of
 * the generator (reference generate_data/data_script.py:98-101).
 */
#ifndef FLASHVIT_HOST_H
#define FLASHVIT_HOST_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
    FVH_ERR_OPEN = -101,
    FVH_ERR_READ = -102,
    FVH_ERR_WRITE = -103,
    FVH_ERR_SHORT = -104,
    FVH_ERR_FORMAT = -105,
    FVH_ERR_NOMEM = -106,
    FVH_ERR_STALE = -107,
};

enum { FVH_DTYPE_F32 = 1, FVH_DTYPE_I32 = 2 };

/* out[i] = strtof(sprintf("%.16f", in[i])): the float the reference's
 * fscanf("%f") loader (FLASH_Viterbi_multithread.c:85-91) obtains from the text
 * data_script.py:98-100 writes for the double in[i]. */
int fvh_quantize_text16(const double *in, float *out, size_t n);

/* np.savetxt(path, a, fmt='%.16f') for a rows x cols matrix (row_newline=1,
 * data_script.py:98-99) or, with rows=n, cols=1, row_newline=0, for a 1-D vector
 * written with newline=' ' (data_script.py:100). */
int fvh_write_matrix_text16(const char *path, const double *a, size_t rows, size_t cols,
                            int row_newline);
/* Same, appending to an existing file when append != 0: lets a generator emit a K x K matrix in row
 * blocks without ever holding it (K = 65536 would be 34 GB of float64). */
int fvh_write_matrix_text16_ex(const char *path, const double *a, size_t rows, size_t cols,
                               int row_newline, int append);
/* np.savetxt(path, v, fmt='%d', newline=' ') (data_script.py:101). */
int fvh_write_ints_text(const char *path, const int *v, size_t n);

/* Whitespace-separated text -> n floats / ints; FVH_ERR_SHORT if the file ends
 * early (the reference would read garbage; InitElement has no such check). */
int fvh_read_floats_text(const char *path, float *out, size_t n);
int fvh_read_ints_text(const char *path, int *out, size_t n);

/* Raw cache of an already-parsed array: 16-byte header + little-endian payload. */
int fvh_write_bin(const char *path, const void *data, uint32_t dtype, uint32_t rows, uint32_t cols);
int fvh_read_bin(const char *path, void *data, uint32_t dtype, uint32_t rows, uint32_t cols);

/* The same cache bound to the text file it was parsed from: the header ("FVB2") also holds that file's size
 * and modification time.  fvh_read_bin_src refuses (FVH_ERR_STALE) a cache whose recorded source differs
 * from the text file now at src_text_path — regenerated inputs keep their names (the names encode only K, T
 * and prob, reference README.md:105-114), so a cache must never outlive its text; an unbound cache
 * (fvh_write_bin, the generator's --bin output) is refused the same way whenever the text exists.  When
 * src_text_path does not exist (cache-only data sets: K = 65536 has no text) any well-formed cache is taken. */
int fvh_write_bin_src(const char *path, const void *data, uint32_t dtype, uint32_t rows, uint32_t cols,
                      const char *src_text_path);
int fvh_read_bin_src(const char *path, void *data, uint32_t dtype, uint32_t rows, uint32_t cols,
                     const char *src_text_path);

const char *fvh_strerror(int rc);

#ifdef __cplusplus
}
#endif
#endif
