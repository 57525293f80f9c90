// Lean fp64 transcendentals of the logistic terms (device code), shared by the logit pass (logit.hip) and the fused Newton
// pass (irls_pass.hip).  Included INSIDE namespace dlsa.
#pragma once
// ---------------------------------------------------------------------------------------------
// Lean fp64 transcendentals for the logistic terms.  The library exp / log1p / division cost ~130 extra
// VGPRs in this kernel (175 VGPRs at NC=1 -> two waves per SIMD, latency-bound at small p); these keep the
// same accuracy class (<= 2 ulp on e, mu, w; softplus to ~1e-16 absolute) in ~60 instructions:
//   e = exp(-|eta|):  k = rint(|eta| log2 e), r = k ln2 - |eta| (two-part ln2, |r| <= 0.347), degree-13
//       polynomial, ldexp;
//   1/(1+e), 1/den: v_rcp_f64 seed + two Newton steps;
//   log1p(e) = log t, t = 1+e in (1,2]: halve t above sqrt 2, s = (t-1)/(t+1) (|s| <= 0.172), 2 atanh(s)
//       as an odd polynomial with 10 terms, + ln2 if halved.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double rcp_newton(double d) {
    double x = __builtin_amdgcn_rcp(d);
    x = fma(fma(-d, x, 1.0), x, x);
    x = fma(fma(-d, x, 1.0), x, x);
    return x;
}

__device__ __forceinline__ double exp_neg(double a) {      // exp(-a), a >= 0
    a = fmin(a, 745.2);
    const double kf = rint(a * 1.4426950408889634);
    double r = fma(kf, 6.93147180369123816490e-01, -a);
    r = fma(kf, 1.90821492927058770002e-10, r);
    double q = 1.6059043836821613e-10;                      // 1/13!
    q = fma(q, r, 2.08767569878681e-09);
    q = fma(q, r, 2.505210838544172e-08);
    q = fma(q, r, 2.755731922398589e-07);
    q = fma(q, r, 2.7557319223985893e-06);
    q = fma(q, r, 2.48015873015873e-05);
    q = fma(q, r, 1.984126984126984e-04);
    q = fma(q, r, 1.388888888888889e-03);
    q = fma(q, r, 8.333333333333333e-03);
    q = fma(q, r, 4.1666666666666664e-02);
    q = fma(q, r, 1.6666666666666666e-01);
    q = fma(q, r, 0.5);
    q = fma(q, r, 1.0);
    q = fma(q, r, 1.0);
    return ldexp(q, -(int)kf);
}

// e = exp(-|eta|) -> mu = sigmoid(eta), wgt = mu(1-mu) = e/(1+e)^2, softplus(eta) = max(eta,0) + log1p(e)
// (three pieces, so that a caller can interleave other work between them: irls_wide.hip; logistic_terms is their composition)
__device__ __forceinline__ void logistic_mu_w(double eta, double e, double& mu, double& wgt) {
    const double inv = rcp_newton(1.0 + e);
    mu = eta >= 0.0 ? inv : e * inv;
    wgt = e * inv * inv;
}
__device__ __forceinline__ double logistic_softplus(double eta, double e) {
    const bool big = e > 0.41421356237309503;               // t = 1 + e > sqrt(2)
    const double num = big ? fma(0.5, e, -0.5) : e;         // t' - 1 with t' = t/2 or t
    const double den = big ? fma(0.5, e, 1.5) : 2.0 + e;    // t' + 1
    const double sv = num * rcp_newton(den);
    const double z = sv * sv;
    double q = 1.0 / 21.0;
    q = fma(q, z, 1.0 / 19.0);
    q = fma(q, z, 1.0 / 17.0);
    q = fma(q, z, 1.0 / 15.0);
    q = fma(q, z, 1.0 / 13.0);
    q = fma(q, z, 1.0 / 11.0);
    q = fma(q, z, 1.0 / 9.0);
    q = fma(q, z, 1.0 / 7.0);
    q = fma(q, z, 1.0 / 5.0);
    q = fma(q, z, 1.0 / 3.0);
    q = fma(q, z, 1.0);
    const double l1p = fma(2.0 * sv, q, big ? 6.931471805599453094e-01 : 0.0);
    return fmax(eta, 0.0) + l1p;
}
template <bool WANT_MU>
__device__ __forceinline__ void logistic_terms(double eta, double& mu, double& wgt, double& softplus) {
    const double e = exp_neg(fabs(eta));
    if (WANT_MU) logistic_mu_w(eta, e, mu, wgt);
    softplus = logistic_softplus(eta, e);
}
