/* dib_ctw.h - C ABI of the infinite-depth context-tree-weighting (CTW) entropy-rate estimator.
 *
 * Replaces the reference's only native component, the Cython extension `ctw`
 * (reference chaos/ctw.pyx:2 `estimate_entropy(seq, alphabet_size)` -> chaos/cppctw.hpp:21 `ctw::cpp_ctw`
 *  -> chaos/cppctw.cpp:160-171 `estimate_entropy(const vector<char>&, char)`), which the chaos notebook calls
 * 75 times per partition.  Host C++ (the algorithm is a serial suffix-tree build - not a
 * GPU kernel): plain pointers and sizes, no Python / torch types, thread-safe (the reference keeps alphabet size and
 * beta in static members, chaos/cppctw.cpp:86-87, so it is not), and batchable across host threads.
 *
 * Results are bit-identical to the reference build on the same libm, including its float32 rounding of the final
 * rate (chaos/cppctw.cpp:101 returns `float`) and its MAX_DEPTH = 512 context cut-off (chaos/cppctw.cpp:13).
 */
#ifndef DIB_CTW_H
#define DIB_CTW_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DIB_CTW_OK 0
#define DIB_CTW_E_ARG (-1)     /* null pointer, n < 0, alphabet outside [1, 127], symbol outside [0, alphabet) */
#define DIB_CTW_E_NOMEM (-2)

const char* dib_ctw_version(void);

/* Entropy rate (bits per symbol) of seq[0..n) over {0..alphabet_size-1}; KT/Dirichlet estimator with
 * beta = 1/alphabet_size (chaos/cppctw.cpp:164).  n == 0 gives NaN like the reference (0/0).
 * Replaces ctw.estimate_entropy (chaos/ctw.pyx:2). */
int dib_ctw_estimate_entropy(const int8_t* seq, int64_t n, int alphabet_size, double* rate_out);

/* Many sequences at once: sequence i is seqs[offsets[i] .. offsets[i+1]) (offsets has n_seq+1 entries), all over the
 * same alphabet; spread over `threads` host threads (<= 0: hardware concurrency).  rates_out[n_seq].
 * The first error (if any) is returned; all valid sequences are still computed. */
int dib_ctw_estimate_entropy_batch(const int8_t* seqs, const int64_t* offsets, int n_seq, int alphabet_size, int threads,
                                   double* rates_out);

/* Number of suffix-tree nodes the estimator built for seq (diagnostics / memory sizing: the reference README warns
 * that periodic sequences exhaust memory; here a node costs 8*alphabet_size + 8 bytes in one arena). */
int dib_ctw_node_count(const int8_t* seq, int64_t n, int alphabet_size, int64_t* nodes_out);

#ifdef __cplusplus
}
#endif
#endif /* DIB_CTW_H */
