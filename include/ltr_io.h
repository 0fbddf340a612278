/* ltr_io.h -- C ABI of libltr_io.so: host-side dataset ingestion for the MI355X ranking path.
 *
 * SURVEY.md section 8 f-3.  Replaces the reference's only native component, the single-threaded
 * SVMrank text parser behind `pytorchltr.datasets.svmrank.parser.parse_svmrank_file`
 * (pytorchltr/datasets/svmrank/parser/svmrank_parser.h:174 `parse_svmrank_file`, bound by
 * svmrank_parser.pyx:22-58).  Plain C types only; no GPU, no torch.  The library parses with
 * `n_threads` threads (chunks cut at line boundaries) and fills caller-owned arrays, so a
 * binding can hand in numpy / pinned-host buffers directly.
 *
 * Results are identical to the reference's on every input it parses consistently: same accepted
 * language, same value arithmetic (long mantissa, pow(10, exp - decimals) in double), columns
 * shifted so the smallest index seen becomes column 0, dense row-major matrix.  The differences
 * (inputs on which the reference returns arrays of inconsistent length or uninitialised
 * entries) are listed in DESIGN.md section 9.
 */
#ifndef LTR_IO_H
#define LTR_IO_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Status codes; 0..3 are the reference's PARSE_* constants (svmrank_parser.h:21-24). */
#define LTR_IO_OK 0
#define LTR_IO_FILE_ERROR 1
#define LTR_IO_FORMAT_ERROR 2
#define LTR_IO_MEMORY_ERROR 3
#define LTR_IO_ARG_ERROR 4

/* Parses the file at `path`.  n_threads <= 0 picks the hardware concurrency (capped at 64 and
 * at one thread per MiB of input).  On LTR_IO_OK, *handle owns the parsed data and *rows /
 * *cols give the shape of the dense matrix (cols == 0 when no feature was seen).  On any error
 * *handle is NULL and nothing needs to be released. */
int ltr_svmrank_open(const char *path, int n_threads, void **handle, size_t *rows, size_t *cols);

/* Copies the parsed data into caller-owned arrays; any pointer may be NULL to skip that output.
 *   xs      (rows, cols) float64 row-major -- the reference's dtype
 *   xs_f32  (rows, cols) float32 row-major -- each value rounded once from its float64
 *   ys      (rows,) int32 labels
 *   qids    (rows,) int64 query ids
 * May be called more than once. */
int ltr_svmrank_read(void *handle, double *xs, float *xs_f32, int32_t *ys, int64_t *qids);

/* Releases a handle (NULL is allowed). */
void ltr_svmrank_close(void *handle);

const char *ltr_io_error_string(int code);

#ifdef __cplusplus
}
#endif

#endif /* LTR_IO_H */
