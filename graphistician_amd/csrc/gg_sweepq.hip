// gg_sweepq.hip - the one kernel of the quiet chain's steady state, k_sweep_q (scan of sweep s | gated parameter step of
// sweep s+1 | riders counting the statistics of sweep s+2), compiled on its own: see gg_sweepq.h.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <math.h>
#include "../../include/graphgibbs.h"
#define GG_ONLY_SWEEPQ 1
#define GG_FIND_SLOT_SERIAL_W true   // (gg_block_kernels.h: the walk over the labels keeps the weight riders' schedule)
#include "gg_common.h"
#include "gg_block_kernels.h"
#include "gg_resolve_kernels.h"
#include "gg_sweepq.h"

template <int NT, bool ADJ, int NTL, int SPLIT>
static int launch1(int grid, size_t lds, hipStream_t stream, const gg_resolve_args& g, int stage, int32_t* ap, const gg_params_args& pg,
                   const gg_ride_args& r, int nspec, unsigned vtag) {
  static bool attr_set = false;
  if (!attr_set && lds > 40 * 1024) {
    GG_HIP(hipFuncSetAttribute((const void*)k_sweep_q<NT, ADJ, NTL, SPLIT>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    attr_set = true;
  }
  k_sweep_q<NT, ADJ, NTL, SPLIT><<<grid, 256, lds, stream>>>(g, stage, ap, pg, r, nspec, vtag);
  return GG_OK;
}

template <int NT, bool ADJ>
static int launch2(int ntl, int split, int grid, size_t lds, hipStream_t stream, const gg_resolve_args& g, int stage, int32_t* ap,
                   const gg_params_args& pg, const gg_ride_args& r, int nspec, unsigned vtag) {
  if (split) return ntl ? launch1<NT, ADJ, 1, 1>(grid, lds, stream, g, stage, ap, pg, r, nspec, vtag)
                        : launch1<NT, ADJ, 0, 1>(grid, lds, stream, g, stage, ap, pg, r, nspec, vtag);
  return ntl ? launch1<NT, ADJ, 1, 0>(grid, lds, stream, g, stage, ap, pg, r, nspec, vtag)
             : launch1<NT, ADJ, 0, 0>(grid, lds, stream, g, stage, ap, pg, r, nspec, vtag);
}

int gg_launch_sweep_q(int NT, int adj, int ntl, int split, int grid, size_t lds, hipStream_t stream, const gg_resolve_args& g,
                      int stage, int32_t* ap, const gg_params_args& pg, const gg_ride_args& r, int nspec, unsigned vtag) {
  if (adj) return launch2<2, true>(ntl, split, grid, lds, stream, g, stage, ap, pg, r, nspec, vtag);
  switch (NT) {
    case 3: return launch2<3, false>(ntl, split, grid, lds, stream, g, stage, ap, pg, r, nspec, vtag);
    case 6: return launch2<6, false>(ntl, split, grid, lds, stream, g, stage, ap, pg, r, nspec, vtag);
    case 10: return launch2<10, false>(ntl, split, grid, lds, stream, g, stage, ap, pg, r, nspec, vtag);
    case 15: return launch2<15, false>(ntl, split, grid, lds, stream, g, stage, ap, pg, r, nspec, vtag);
  }
  return gg_fail(GG_ERR_UNSUPPORTED, "gg_launch_sweep_q: no instantiation for NT=%d", NT);
}
