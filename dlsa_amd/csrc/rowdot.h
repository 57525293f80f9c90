// Row dot products across a wave (device code): a wave holds RB rows, lane l the columns 128c + 2l + {0,1}; the RB partial
// dot products are reduced with ONE merged butterfly and the per-row results are broadcast back through SGPRs.
// Shared by the logit pass (logit.hip) and the wide Newton pass (irls_wide.hip).  Included INSIDE namespace dlsa.
#pragma once
// Cross-lane exchanges without LDS round trips: dpp_xor_f64 / swap_f64 of common.h.
// (lane & M ? hi : lo) of this lane + the same quantity of lane ^ M
template <int M>
__device__ __forceinline__ double exch_add(double lo, double hi, int lane) {
    if constexpr (M >= 16) {
        double a2, b2;
        swap_f64<M>(lo, hi, a2, b2);        // no select: a2 + b2 is the kept value plus the partner's copy in both halves
        return a2 + b2;
    } else {
        const double s0 = lo + dpp_xor_f64<M>(lo), s1 = hi + dpp_xor_f64<M>(hi);
        return (lane & M) ? s1 : s0;
    }
}
template <int M>
__device__ __forceinline__ double xor_add(double s) {           // s + s of lane ^ M
    if constexpr (M >= 16) return exch_add<M>(s, s, 0);
    else return s + dpp_xor_f64<M>(s);
}
template <int L>
__device__ __forceinline__ double read_lane_f64(double v) {     // broadcast of lane L through SGPRs
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), L), __builtin_amdgcn_readlane(__double2loint(v), L));
}

template <int RB>
__device__ __forceinline__ double merged_reduce(double (&v)[RB], int lane) {
    // after this, every lane holds the wave-sum of row rsel(lane): log2(RB) halving steps (masks 32, 16, ...), then
    // plain xor-sums over the remaining masks
    if constexpr (RB >= 2) {
#pragma unroll
        for (int i = 0; i < RB / 2; ++i) v[i] = exch_add<32>(v[i], v[i + RB / 2], lane);
    }
    if constexpr (RB >= 4) {
#pragma unroll
        for (int i = 0; i < RB / 4; ++i) v[i] = exch_add<16>(v[i], v[i + RB / 4], lane);
    }
    if constexpr (RB >= 8) {
#pragma unroll
        for (int i = 0; i < RB / 8; ++i) v[i] = exch_add<8>(v[i], v[i + RB / 8], lane);
    }
    if constexpr (RB >= 16) {
#pragma unroll
        for (int i = 0; i < RB / 16; ++i) v[i] = exch_add<4>(v[i], v[i + RB / 16], lane);
    }
    if constexpr (RB >= 32) v[0] = exch_add<2>(v[0], v[1], lane);
    static_assert(RB <= 32, "merged_reduce: at most 32 values");
    double s = v[0];
    if constexpr (RB < 2) s = xor_add<32>(s);
    if constexpr (RB < 4) s = xor_add<16>(s);
    if constexpr (RB < 8) s = xor_add<8>(s);
    if constexpr (RB < 16) s = xor_add<4>(s);
    if constexpr (RB < 32) s = xor_add<2>(s);
    s = xor_add<1>(s);
    return s;
}

// row handled by `lane` after merged_reduce, and the representative lane of row i
template <int RB>
__device__ __forceinline__ int row_of_lane(int lane) {
    int r = 0, m = 32;
#pragma unroll
    for (int cnt = RB; cnt > 1; cnt >>= 1, m >>= 1) r += ((lane & m) ? 1 : 0) * (cnt / 2);
    return r;
}
template <int RB>
__host__ __device__ constexpr int lane_of_row(int i) {
    int lane = 0, m = 32;
    for (int cnt = RB; cnt > 1; cnt >>= 1, m >>= 1) {
        if (i >= cnt / 2) { lane |= m; i -= cnt / 2; }
    }
    return lane;
}
template <int RB>
__host__ __device__ constexpr int rep_mask() {   // lane bits that must be zero for a representative lane
    int used = 0, m = 32;
    for (int cnt = RB; cnt > 1; cnt >>= 1, m >>= 1) used |= m;
    return 63 & ~used;
}

// g += sum_i resid(row i) * x_i : the residual of row i sits in lane lane_of_row(i) and is broadcast through SGPRs
template <int RB, int NC, int I>
__device__ __forceinline__ void rank1_update(double resid, const double2 (&x)[RB][NC], double2 (&g)[NC]) {
    if constexpr (I < RB) {
        const double ri = read_lane_f64<lane_of_row<RB>(I)>(resid);
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            g[c].x = fma(ri, x[I][c].x, g[c].x);
            g[c].y = fma(ri, x[I][c].y, g[c].y);
        }
        rank1_update<RB, NC, I + 1>(resid, x, g);
    }
}

