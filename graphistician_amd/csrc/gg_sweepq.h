// gg_sweepq.h - the quiet chain's fused launch (k_sweep_q, gg_resolve_kernels.h) lives in a translation unit of its own.
//
// Measured (round 3): the register allocation and the schedule the compiler picks for the rider loop inside k_sweep_q move
// with ANY change to the code compiled next to it - a four-line, functionally void edit of this library took the weight
// model's fused launch from 32 to 42 us (same instructions, different allocation: profiles/README.md).  Compiled alone,
// from gg_sweepq.hip, the kernel changes only when its own code does.
#pragma once
#include "gg_resolve_kernels.h"

// NT: 2 (adjacency), 3, 6, 10, 15 (weights, B = 1..4); ntl: riders with non-temporal loads; split: the K > 32 conditional
// without staged tables.  Returns GG_OK or GG_ERR_*.
int gg_launch_sweep_q(int NT, int adj, int ntl, int split, int grid, size_t lds, hipStream_t stream, const gg_resolve_args& g,
                      int stage, int32_t* ap, const gg_params_args& pg, const gg_ride_args& r, int nspec, unsigned vtag);
