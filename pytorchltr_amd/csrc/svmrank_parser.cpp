// svmrank_parser.cpp -- multi-threaded SVMrank text parser (host side, C++17, no GPU).
//
// the on-disk format of every dataset the reference loads, and the
// reference's only native component (pytorchltr/datasets/svmrank/parser/svmrank_parser.h, a
// single-threaded table-driven DFA wrapped with Cython).  This is a from-scratch parser for the
// same language with the same results:
//
//   line     := ' '* ( comment | record )
//   comment  := '#' any* '\n'
//   record   := digits ' '+ "qid:" digits tail
//   tail     := ( ' '+ feature )* ' '* ( '#' any* | '\r' any* )? '\n'
//   feature  := digits ':' '-'* digits ( '.' digits+ ( [eE] [+-]? digits* )? )?
//
// (no empty lines, labels and ids are non-negative integers, an exponent needs a fraction,
// exactly as the reference's transition table accepts -- svmrank_parser.h:80-132).  A value is
// (double)(sign * mantissa) * pow(10, exponent - fraction_digits) (svmrank_parser.h:397-400), so
// results are bit-identical to the reference; columns are shifted so the smallest index seen
// becomes column 0 and the matrix is dense (svmrank_parser.h:474-485).
//
// Structure: the file is read once, cut into chunks at line boundaries, every chunk is scanned by
// its own thread into local (label, qid, (row, col, value)) arrays, then rows are renumbered by a
// prefix sum and the dense matrix is filled in parallel.  Lines are independent in this format
// (the reference's automaton returns to its start state at every '\n'), so the split is exact.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include "ltr_io.h"

namespace {

struct Chunk {
    const char *begin = nullptr;
    const char *end = nullptr;
    bool last = false;                 // ends at end-of-file (no trailing newline handling)
    std::vector<int32_t> ys;
    std::vector<int64_t> qids;
    std::vector<uint32_t> row;         // local row of every feature
    std::vector<uint32_t> col;
    std::vector<double> val;
    uint32_t min_col = UINT32_MAX;
    uint32_t max_col = 0;
    bool any_col = false;
    int error = LTR_IO_OK;
};

inline bool is_digit(char c) { return c >= '0' && c <= '9'; }

// Skips to just after the next '\n' (or to `end`).
inline const char *skip_line(const char *p, const char *end)
{
    const void *nl = memchr(p, '\n', (size_t)(end - p));
    return nl ? (const char *)nl + 1 : end;
}

// pow(10, k) for the exponents real files use, computed by the same libm call the reference makes
// per value (svmrank_parser.h:399), so a table hit returns the identical double.
constexpr int kPowRange = 40;
struct PowTable {
    double v[2 * kPowRange + 1];
    PowTable() { for (int k = -kPowRange; k <= kPowRange; ++k) v[k + kPowRange] = std::pow(10.0, (double)k); }
};
inline double pow10_exact(long k)
{
    static const PowTable table;
    if (k >= -kPowRange && k <= kPowRange) return table.v[k + kPowRange];
    return std::pow(10.0, (double)k);
}

// Parses one chunk.  Every line must be complete except possibly the last line of the file.
void parse_chunk(Chunk &c)
{
    const char *p = c.begin;
    const char *const end = c.end;
    const size_t guess = (size_t)(end - p) / 10 + 16;         // ~10 bytes per "col:value "
    c.row.reserve(guess);
    c.col.reserve(guess);
    c.val.reserve(guess);
    while (p < end) {
        // ---- start of line ----
        while (p < end && *p == ' ') ++p;
        if (p == end) break;                                   // trailing blanks at end of file
        if (*p == '#') { p = skip_line(p, end); continue; }    // comment line
        if (!is_digit(*p)) { c.error = LTR_IO_FORMAT_ERROR; return; }
        int32_t y = 0;
        while (p < end && is_digit(*p)) y = y * 10 + (*p++ - '0');
        if (p == end || *p != ' ') { c.error = LTR_IO_FORMAT_ERROR; return; }
        c.ys.push_back(y);
        const uint32_t row = (uint32_t)(c.ys.size() - 1);
        while (p < end && *p == ' ') ++p;
        if (end - p < 5 || memcmp(p, "qid:", 4) != 0 || !is_digit(p[4])) { c.error = LTR_IO_FORMAT_ERROR; return; }
        p += 4;
        int64_t qid = 0;
        while (p < end && is_digit(*p)) qid = qid * 10 + (*p++ - '0');
        // The reference stores a qid only when a blank follows it (its STORE_QID action sits on
        // ' ' alone, svmrank_parser.h:143), so for a record without features or trailing blank
        // ("3 qid:7\n") it returns an uninitialised id.  Here the id is stored in every case.
        if (p < end && *p != ' ' && *p != '\n' && *p != '#' && *p != '\r') { c.error = LTR_IO_FORMAT_ERROR; return; }
        c.qids.push_back(qid);
        // ---- features ----
        for (;;) {
            while (p < end && *p == ' ') ++p;
            if (p == end) return;                              // file ends inside a record: done
            if (*p == '\n') { ++p; break; }
            if (*p == '#' || *p == '\r') { p = skip_line(p, end); break; }
            if (!is_digit(*p)) { c.error = LTR_IO_FORMAT_ERROR; return; }
            uint32_t col = 0;
            while (p < end && is_digit(*p)) col = col * 10 + (uint32_t)(*p++ - '0');
            if (p == end || *p != ':') { c.error = LTR_IO_FORMAT_ERROR; return; }
            ++p;
            long sign = 1;
            while (p < end && *p == '-') { sign = -1; ++p; }
            if (p == end || !is_digit(*p)) { c.error = LTR_IO_FORMAT_ERROR; return; }
            long mant = 0, decimals = 0, expv = 0, expsign = 1;
            while (p < end && is_digit(*p)) mant = mant * 10 + (*p++ - '0');
            if (p < end && *p == '.') {
                ++p;
                if (p == end || !is_digit(*p)) { c.error = LTR_IO_FORMAT_ERROR; return; }
                while (p < end && is_digit(*p)) { mant = mant * 10 + (*p++ - '0'); ++decimals; }
                if (p < end && (*p == 'e' || *p == 'E')) {
                    ++p;
                    if (p < end && (*p == '-' || *p == '+')) { if (*p == '-') expsign = -1; ++p; }
                    else if (p == end || !is_digit(*p)) { c.error = LTR_IO_FORMAT_ERROR; return; }
                    while (p < end && is_digit(*p)) expv = expv * 10 + (*p++ - '0');
                }
            }
            // a value must be followed by a blank, a comment, CR, LF or the end of the file
            if (p < end && *p != ' ' && *p != '#' && *p != '\r' && *p != '\n') { c.error = LTR_IO_FORMAT_ERROR; return; }
            // At end-of-file without a newline the reference finishes the pending value WITHOUT
            // its sign (svmrank_parser.h:449); reproduced so results stay identical.
            const bool eof_value = (p == end) && c.last;
            double v = eof_value ? (double)mant : (double)(sign * mant);
            v = v * pow10_exact(expv * expsign - decimals);
            c.row.push_back(row);
            c.col.push_back(col);
            c.val.push_back(v);
            if (!c.any_col || col < c.min_col) c.min_col = col;
            if (!c.any_col || col > c.max_col) c.max_col = col;
            c.any_col = true;
        }
    }
}

struct Parsed {
    std::vector<Chunk> chunks;
    std::vector<size_t> row_base;      // first global row of every chunk
    size_t rows = 0;
    size_t cols = 0;
    uint32_t min_col = 0;
    const char *data = nullptr;        // the file, mapped read-only
    size_t bytes = 0;
    ~Parsed() { if (data && bytes) munmap(const_cast<char *>(data), bytes); }
};

int clamp_threads(int n, size_t bytes)
{
    if (n <= 0) n = (int)std::thread::hardware_concurrency();
    if (n <= 0) n = 1;
    if (n > 64) n = 64;
    const size_t by_size = bytes / (1u << 20) + 1;            // at least ~1 MiB per thread
    if ((size_t)n > by_size) n = (int)by_size;
    return n;
}

template <typename F>
void run_parallel(int n, F fn)
{
    if (n <= 1) { fn(0); return; }
    std::vector<std::thread> th;
    th.reserve((size_t)n);
    for (int t = 0; t < n; ++t) th.emplace_back(fn, t);
    for (auto &t : th) t.join();
}

}  // namespace

extern "C" {

int ltr_svmrank_open(const char *path, int n_threads, void **handle, size_t *rows, size_t *cols)
{
    if (!path || !handle || !rows || !cols) return LTR_IO_ARG_ERROR;
    *handle = nullptr;
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return LTR_IO_FILE_ERROR;
    struct stat st;
    if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode)) { close(fd); return LTR_IO_FILE_ERROR; }
    Parsed *ps = new (std::nothrow) Parsed();
    if (!ps) { close(fd); return LTR_IO_MEMORY_ERROR; }
    try {
        ps->bytes = (size_t)st.st_size;
        if (ps->bytes) {
            void *m = mmap(nullptr, ps->bytes, PROT_READ, MAP_PRIVATE, fd, 0);
            if (m == MAP_FAILED) { ps->bytes = 0; close(fd); delete ps; return LTR_IO_FILE_ERROR; }
            madvise(m, ps->bytes, MADV_SEQUENTIAL | MADV_WILLNEED);
            ps->data = static_cast<const char *>(m);
        }
        close(fd);
        const char *base = ps->data;
        const char *end = base + ps->bytes;
        const int nt = clamp_threads(n_threads, ps->bytes);
        // cut at line boundaries
        ps->chunks.resize((size_t)nt);
        const char *cur = base;
        for (int t = 0; t < nt; ++t) {
            const char *stop = (t == nt - 1) ? end : base + ps->bytes / (size_t)nt * (size_t)(t + 1);
            if (stop < cur) stop = cur;
            if (t != nt - 1) stop = skip_line(stop, end);
            ps->chunks[(size_t)t].begin = cur;
            ps->chunks[(size_t)t].end = stop;
            cur = stop;
        }
        for (int t = nt - 1; t >= 0; --t)                        // the chunk that reaches EOF
            if (ps->chunks[(size_t)t].begin < ps->chunks[(size_t)t].end || t == 0) { ps->chunks[(size_t)t].last = true; break; }
        run_parallel(nt, [&](int t) { parse_chunk(ps->chunks[(size_t)t]); });
        bool any = false;
        uint32_t lo = 0, hi = 0;
        ps->row_base.resize((size_t)nt);
        for (int t = 0; t < nt; ++t) {
            Chunk &c = ps->chunks[(size_t)t];
            if (c.error != LTR_IO_OK) { const int e = c.error; delete ps; return e; }
            if (c.ys.size() != c.qids.size()) { delete ps; return LTR_IO_FORMAT_ERROR; }
            ps->row_base[(size_t)t] = ps->rows;
            ps->rows += c.ys.size();
            if (c.any_col) {
                if (!any || c.min_col < lo) lo = c.min_col;
                if (!any || c.max_col > hi) hi = c.max_col;
                any = true;
            }
        }
        ps->min_col = any ? lo : 0;
        ps->cols = any ? (size_t)(hi - lo) + 1 : 0;
    } catch (const std::bad_alloc &) {
        delete ps;
        return LTR_IO_MEMORY_ERROR;
    }
    *rows = ps->rows;
    *cols = ps->cols;
    *handle = ps;
    return LTR_IO_OK;
}

int ltr_svmrank_read(void *handle, double *xs, float *xs_f32, int32_t *ys, int64_t *qids)
{
    if (!handle) return LTR_IO_ARG_ERROR;
    Parsed *ps = static_cast<Parsed *>(handle);
    const size_t cols = ps->cols;
    const int nt = (int)ps->chunks.size();
    run_parallel(nt, [&](int t) {
        const Chunk &c = ps->chunks[(size_t)t];
        const size_t r0 = ps->row_base[(size_t)t];
        if (c.ys.empty()) return;                        // (a chunk without rows: its vectors' data() may be null)
        if (xs) memset(xs + r0 * cols, 0, sizeof(double) * c.ys.size() * cols);
        if (xs_f32) memset(xs_f32 + r0 * cols, 0, sizeof(float) * c.ys.size() * cols);
        if (ys) memcpy(ys + r0, c.ys.data(), sizeof(int32_t) * c.ys.size());
        if (qids) memcpy(qids + r0, c.qids.data(), sizeof(int64_t) * c.qids.size());
        for (size_t i = 0; i < c.val.size(); ++i) {
            const size_t at = (r0 + c.row[i]) * cols + (size_t)(c.col[i] - ps->min_col);
            if (xs) xs[at] = c.val[i];
            if (xs_f32) xs_f32[at] = (float)c.val[i];
        }
    });
    return LTR_IO_OK;
}

void ltr_svmrank_close(void *handle)
{
    delete static_cast<Parsed *>(handle);
}

const char *ltr_io_error_string(int code)
{
    switch (code) {
    case LTR_IO_OK: return "ok";
    case LTR_IO_FILE_ERROR: return "could not open or read the file";
    case LTR_IO_FORMAT_ERROR: return "not in SVMrank format";
    case LTR_IO_MEMORY_ERROR: return "could not allocate memory";
    case LTR_IO_ARG_ERROR: return "invalid argument";
    default: return "unknown error";
    }
}

}  // extern "C"
