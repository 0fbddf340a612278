/* svmrank_ref_shim.c -- TEST INFRASTRUCTURE ONLY.
 *
 * A thin export layer over the reference's own SVMrank parser, compiled FROM THE SOURCES WHERE
 * THEY LIE under /root/reference (pytorchltr/datasets/svmrank/parser/svmrank_parser.h is a
 * self-contained C header: no cmake, no generated code, libc + libm only) by
 * oracle/build_ref.py into oracle/_ref/libsvmrank_ref.so.  Nothing of the reference is copied
 * into this repository; this file only includes the header through the -I path given on the
 * command line and re-exports its entry point with a ctypes-friendly signature.
 *
 * Used by tests/test_svmrank_parser.py as the checker for pytorchltr_amd's own parser
 * (pytorchltr_amd/csrc/svmrank_parser.cpp) and by tests/golden/generate_svmrank_golden.py to
 * produce the committed expected outputs.  The product never loads it.
 */
#include "svmrank_parser.h"

int ref_svmrank_parse(const char *path, double **xs, size_t *rows, size_t *cols, int **ys,
                      long **qids)
{
    shape sh;
    sh.rows = 0;
    sh.cols = 0;
    init_svmrank_parser();
    int rc = parse_svmrank_file((char *)path, xs, &sh, ys, qids);
    *rows = sh.rows;
    *cols = sh.cols;
    return rc;
}

void ref_svmrank_free(void *p)
{
    free(p);
}
