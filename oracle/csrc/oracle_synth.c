/* TEST INFRASTRUCTURE ONLY -- C restatement of the oracle's seeded row generator (oracle/dlsa_oracle.py:
 * philox4x32_10, synth_features, synth_label_uniforms), which itself restates simulate_logistic
 * (reference dlsa/models.py:6-40: x ~ U(-0.5,0.5) :22, label ~ Bernoulli(sigmoid(x.beta)) :23,30) with a counter RNG.
 * Used by bench.py's cpu_baseline leg and by tests/ to produce the CPU sample quickly (the numpy generator needs
 * ~1.3 us per element); never linked into or called by the product library.
 * The integer pipeline is identical to the numpy and HIP generators: uniform rows are bit-identical; Gaussian rows
 * differ from numpy's only through libm's log / sin / cos (last-ulp).
 * Build: gcc -O2 -fPIC -shared -fopenmp oracle/csrc/oracle_synth.c -o oracle/_build/liboracle_synth.so -lm */
#include <math.h>
#include <stdint.h>

static inline void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}

static inline double u53(uint32_t hi, uint32_t lo) {
    return ((double)(hi >> 5) * 67108864.0 + (double)(lo >> 6)) * (1.0 / 9007199254740992.0);
}

/* X[i, j], i in [row0, row0+n), j < p, row pitch ldx; kind 0 uniform(-0.5,0.5), 1 N(0,1/12) */
void oracle_synth_features(uint64_t seed, int64_t row0, int64_t n, int p, int kind, double* X, int64_t ldx) {
    const int npair = (p + 1) / 2;
    const double sd = sqrt(1.0 / 12.0);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        uint64_t row = (uint64_t)(row0 + i);
        double* x = X + i * ldx;
        for (int j = 0; j < npair; ++j) {
            uint32_t c[4] = {(uint32_t)row, (uint32_t)(row >> 32), (uint32_t)j, 0u};
            philox4x32_10(c, (uint32_t)seed, 0u);
            double ua = u53(c[0], c[1]), ub = u53(c[2], c[3]), a, b;
            if (kind == 0) { a = ua - 0.5; b = ub - 0.5; }
            else {
                double rad = sqrt(-2.0 * log(1.0 - ua)) * sd, ang = 2.0 * M_PI * ub;
                a = rad * cos(ang); b = rad * sin(ang);
            }
            x[2 * j] = a;
            if (2 * j + 1 < p) x[2 * j + 1] = b;
        }
    }
}

/* u_i ~ U(0,1) of the Bernoulli draw: counter (i_lo, i_hi, 0, 1), key (seed + 1, 0) */
void oracle_synth_label_uniforms(uint64_t seed, int64_t row0, int64_t n, double* u) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        uint64_t row = (uint64_t)(row0 + i);
        uint32_t c[4] = {(uint32_t)row, (uint32_t)(row >> 32), 0u, 1u};
        philox4x32_10(c, (uint32_t)(seed + 1), 0u);
        u[i] = u53(c[0], c[1]);
    }
}
