// gq_tail.h -- the last launch of a fused arg-max call: everything that only runs for rows the first filter +
// re-rank could not decide, as ONE device-dispatched kernel (the host cannot know the list length without a sync).
//
//   list A empty                      -> every block returns at once (the common case: a few microseconds)
//   cascade && |A| > kCascadeMin      -> phase A: fp32 MFMA filter on the listed rows (gq_filter.h, row-list mode)
//                                        grid barrier
//                                        phase B: level-2 re-rank of list A (16x tighter margin) -> list B
//                                        grid barrier
//   then                              -> fp64 second stage on list B (or on list A when the cascade did not run):
//                                        spread over 32 blocks per row for <= 64 rows, 8 rows per block otherwise.
// The phases walk virtual blocks, so the grid is whatever is co-resident (host: occupancy API x CU count), which
// the two barriers want.  The barriers are only ever executed on the ill-conditioned path, and they may fail
// (gq_rerank.h:grid_barrier: spin limit, or another block reported one): then -- and in every block that starts
// after that -- list A is finished by exhaustive_rows(), one row per block, no dependence on any other block.
// Rows a block had already decided in phase B were decided from complete phase-A records (a barrier only returns
// true after every block has arrived), so whichever path writes a row writes the reference's arg-max.
#pragma once
#include "gq_filter.h"
#include "gq_rerank.h"

namespace gqhip {

template <int MODE, int DIM>
__global__ __launch_bounds__(256, 1) void gq_tail_kernel(const RerankParams p, const FilterParams f2) {
  constexpr int GT2 = DIM <= 8 ? 4 : 2;      // tiles per candidate group of the fp32 filter
  constexpr int CT2 = DIM == 32 ? 4 : 8;
  const int count_a = p.hdr->fb_count;       // final: written by the previous launch
  if (count_a == 0) return;
  if (p.cascade && count_a > kCascadeMin) {
    bool ok = !barrier_aborted(p.hdr);   // a block that starts after an abort goes straight to the barrier-free finish
    if (ok) {
      const int nvb_f = ((count_a + 127) / 128) * f2.nsplit;
      for (int vb = blockIdx.x; vb < nvb_f; vb += gridDim.x) {
        filter_block<DIM, 1, CT2, MODE, GT2>(f2, vb);
        __syncthreads();
      }
      ok = grid_barrier(p.hdr, gridDim.x, p.bar_spin_limit);
    }
    if (ok) {
      constexpr int RPB = 4 * (64 / kRerankLanes);
      const int nvb_r = (count_a + RPB - 1) / RPB;
      for (int vb = blockIdx.x; vb < nvb_r; vb += gridDim.x) {
        rerank_block<MODE, DIM, GT2>(p, vb, count_a);
        __syncthreads();
      }
      ok = grid_barrier(p.hdr, gridDim.x, p.bar_spin_limit);
    }
    if (!ok) {
      exhaustive_rows<MODE>(p, p.fb_list, count_a, (int)blockIdx.x, (int)gridDim.x);
      return;
    }
  }
  second_stage<MODE, DIM>(p);
}

}  // namespace gqhip
