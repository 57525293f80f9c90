// Shared declarations of the plan-driven fp64 Gram kernel (gram_plan.hip, gram_plan_unit.hip, gram_plan_kernel.inc).
#pragma once
#include "common.h"

namespace dlsa {

typedef double dlsa_d2 __attribute__((ext_vector_type(2)));

constexpr int PLAN_NT_MIN = 8, PLAN_NT_MAX = 35;
constexpr int PLAN_KC = 8, PLAN_NST = 4;          // rows per chunk in multi-CU groups (two k-steps of 4), LDS ring stages
constexpr int PLAN_KC1 = 16;                      // rows per chunk of the single-CU plans (four k-steps per barrier: their ring fits)
constexpr int plan_kc(int C) { return C == 1 ? PLAN_KC1 : PLAN_KC; }
constexpr int64_t PLAN_MIN_ROWS = 32768;
#ifndef PLAN_AHEAD_CHUNKS
#define PLAN_AHEAD_CHUNKS 8
#endif
constexpr int PLAN_AHEAD = PLAN_AHEAD_CHUNKS;                     // chunks a workgroup may run ahead of the slowest of its group
constexpr int PLAN_PACE_SPINS = 512;              // pacing sleeps (~2 us each) per launch and workgroup: 512 + chunks / 4, then it runs on unpaced

struct PlanArgs {
    const double* X;
    const double* w;      // the row weights, or PLAN_KC ones when w_step == 0 (unweighted Gram)
    int w_step;           // 1: w advances with the rows; 0: every chunk reads the same PLAN_KC doubles
    double* partial;      // [nslab][PP][PP]
    int* progress;        // [nslab][4] chunk counts of the workgroups of a slab group, zero at launch (C > 1)
    int64_t ldx, n, rows_per_slab;
    int p;                // columns loaded (even)
    int PP;
    unsigned long long* clk;   // clock probe: wave 0 of workgroup 0 stores its s_memtime delta (dlsa_gram_last_kernel)
};

// LDS row pitch in doubles: the tile columns rounded up to whole 32-column groups (a multiple of 256 bytes keeps the two rows
// a ds_read_b128 lane group spans on disjoint banks)
constexpr int plan_pitch(int ntc) { return 32 * ((ntc + 1) / 2); }
constexpr int plan_buf(int ntc, int kc) { return kc * plan_pitch(ntc) + kc; }             // a chunk + its w

// workgroups (CUs of one XCD) that share a slab, by the number of full tiles; tools/gen_gram_plan_asm.py groups_for()
constexpr int plan_group(int nt) { return nt <= 17 ? 1 : nt <= 24 ? 2 : 4; }

// one per translation unit (Makefile PLAN_UNITS): widths LO .. HI
int gram_plan_launch_8(const PlanArgs& a, int nt, int g, int nslab, hipStream_t stream);
int gram_plan_launch_18(const PlanArgs& a, int nt, int g, int nslab, hipStream_t stream);
int gram_plan_launch_25(const PlanArgs& a, int nt, int g, int nslab, hipStream_t stream);
int gram_plan_launch_29(const PlanArgs& a, int nt, int g, int nslab, hipStream_t stream);
int gram_plan_launch_33(const PlanArgs& a, int nt, int g, int nslab, hipStream_t stream);

}  // namespace dlsa
