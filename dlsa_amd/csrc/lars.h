// Arguments and workspace of the LARS path kernels (lars.hip: R^{-1} form, one workgroup or a grid; lars_q.hip: carried
// Cholesky rows, one workgroup).  Reference: lars_lsa, dlsa/lsa.py:90-212.
#pragma once
#include "common.h"
#include <mutex>

namespace dlsa {

struct LarsArgs {
    const double* Sigma0;   // p x p
    const double* b0;       // p
    int64_t lds0;
    int p, intercept, type, max_steps;
    double n, eps;
    // workspace
    double* S;        // m x ld scaled Sigma (ld = m rounded up to even; the pad column is zero)
    double* Rinv;     // m x ld upper triangular inverse factor (active order), rows
    double* RinvT;    // m x ld its transpose, rows
    double* vec;      // 12 vectors of length m (see kernel)
    int* ivec;        // 4 int vectors of length m
    // outputs
    double* beta_path; double* beta0; double* aic; double* bic;
    int* n_steps;     // device scalar
    // multi-workgroup kernel only
    double* rbuf;     // m: r = R^{-T} x of the current append, gathered from the row owners
    double* wbuf;     // m: equiangular weights, gathered from the row owners
    double* upart;    // G x ld: per-workgroup partial sums of Sigma[:,active] w
    unsigned* bar;    // grid barrier: [0] arrival counter, [1] abort word (both zero at launch)
    long long bar_timeout;   // ticks of the 100 MHz wall clock a workgroup waits at a grid barrier before it aborts the launch
    int nwg;          // lars_q.hip: workgroups sharing the rows of the fused pass (1: none)
};

extern std::mutex g_lars_grid_mu;      // lars.hip: serialises this process's multi-workgroup path launches (launch .. completion)

// lars_q.hip: the path kernel for narrow problems (m = p - intercept <= LARS_Q_MAX_M)
constexpr int LARS_Q_MAX_M = 1020;
bool lars_q_eligible(int p, int intercept);
// returns DLSA_OK with *aborted = 1 when a clustered launch gave up at its barrier (nothing usable was written: run again with max_wgs = 1)
int lars_q_run(LarsArgs& a, int p, int intercept, hipStream_t s, int max_wgs, int* wgs_used);

// lars_c.hip (round 6): the same carried rows for WIDE problems (LARS_C_MIN_M < m <= LARS_C_MAX_M), the fused pass split by COLUMNS over
// up to LARS_C_MAX_WGS workgroups that meet at one bounded grid barrier per append; *n_steps = -1 when a barrier gave up (the caller
// reruns the path on lars.hip's single-workgroup kernel)
constexpr int LARS_C_MAX_M = 2044, LARS_C_MAX_WGS = 64;
constexpr int LARS_C_MIN_M = 448;                   // narrower problems: lars_q.hip (same box, lasso paths, lars_q.hip / lars_c.hip: p = 400 3.70 / 3.59 ms, 500 5.10 / 4.61,
                                                    // 640 9.31 / 6.26, 1020 20.8 / 12.1: profiles/r06_lars_column_split.txt)
bool lars_c_eligible(int p, int intercept);
int lars_c_run(LarsArgs& a, int p, int intercept, hipStream_t s, int* wgs_used);

}  // namespace dlsa
