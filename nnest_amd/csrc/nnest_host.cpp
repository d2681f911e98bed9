// nnest_host.cpp -- host-side helpers of libnnest_hip.so that are not kernels.
//
// nnest_format_rows_e5: the text format of the reference's chain files (Sampler._save_samples, nnest/sampler.py:494-511:
// np.savetxt(..., fmt='%.5E'): one row per sample, "weight minusloglike params...", '%.5E' numbers separated by one space,
// '\n' after every row).  np.savetxt formats row by row in Python (9 us per 52-column row: 1.8 s for a config-2 chain of 2e5
// rows, 15 % of the run); here blocks of rows are formatted with snprintf("%.5E") -- the C library's correctly rounded
// conversion, the same digits Python's '%' produces -- on a few threads.
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

#include "../../include/nnest_hip.h"

extern "C" long nnest_format_rows_e5(const double *rows, long n_rows, int n_cols, char *out, long out_cap, int threads) {
    if (!rows || !out || n_rows < 0 || n_cols < 1) return -1;
    const long per_row = (long)n_cols * 14;   // "-1.23457E+123 " is 14 characters
    if (out_cap < n_rows * per_row + 1) return -1;
    if (threads < 1) threads = 1;
    if (threads > 16) threads = 16;
    if (n_rows < 4096) threads = 1;
    std::vector<long> used(threads, 0);
    const long chunk = (n_rows + threads - 1) / threads;
    auto work = [&](int t) {
        const long r0 = t * chunk, r1 = r0 + chunk < n_rows ? r0 + chunk : n_rows;
        char *p = out + r0 * per_row;   // every thread writes into its own slice of the worst-case layout
        for (long r = r0; r < r1; ++r) {
            const double *x = rows + r * n_cols;
            for (int c = 0; c < n_cols; ++c) {
                p += snprintf(p, 16, "%.5E", x[c]);
                *p++ = c + 1 < n_cols ? ' ' : '\n';
            }
        }
        used[t] = r0 < r1 ? p - (out + r0 * per_row) : 0;
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < threads; ++t) pool.emplace_back(work, t);
    work(0);
    for (auto &th : pool) th.join();
    long total = used[0];   // compact the slices
    for (int t = 1; t < threads; ++t) {
        if (used[t] > 0) memmove(out + total, out + (long)t * chunk * per_row, (size_t)used[t]);
        total += used[t];
    }
    return total;
}
