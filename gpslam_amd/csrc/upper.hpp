// upper.hpp -- the solver hierarchy above level 0 as LDS-resident block cyclic reduction (upper.hip).
//
// Level 0 of the partitioned block Gauss-Jordan (kernels.hpp: k_fused_level0 / k_chunk_forward_rows / k_chunk_forward)
// leaves one separator record [D | O | g] per chunk plus the addend [RD | Rg] it owes the next separator.  Everything
// above that is latency, not bandwidth (2 % of the data): one launch per level of chunks of four cost 141 us of a 470 us
// iteration on the 1e5-state Pose3 chain (profiles/round2_v4).  Here a workgroup takes a GROUP of kUpG = 32 consecutive
// blocks into LDS and reduces it to its first block by five sub-levels of cyclic reduction (pairs (s, j = s + 2^q): j is
// eliminated into its left neighbour s and into the block 2^q to its right), the panel of a pair spread over two
// 16-lane DPP rows in the first sub-level and over the four rows of a wave from the second on (cr_step.hpp, cr_quad.hpp),
// the waves of the workgroup working independent pairs; sub-levels are separated by LDS-only barriers instead of kernel
// boundaries.  n blocks -> ceil(n / 32) per launch; the launch that finds <= 32 blocks
// also solves the last one and back-substitutes the whole group in LDS (TOP).
#pragma once

#include <hip/hip_runtime.h>

namespace gps {

constexpr int kUpG = 32;    // blocks per group
constexpr int kUpQ = 5;     // sub-levels: 2^kUpQ == kUpG

struct UpFwdArgs {
  double *blk;        // n records [D (B x B) | O (B x B) | g (B)] of this level; the eliminated ones become [V | U | Y]
  const double *add;  // n + 1 addends [RD (B x B) | Rg (B)]: entry j joins block j, entry n is owed to the block beyond
                      // the level (the next rank's separator), or null
  double *up_blk;     // !TOP: one record per group = the group's first block after the reduction
  double *up_add;     // !TOP: ngroups + 1 addends of the next level (entry g + 1 is written by group g)
  double *x;          // TOP: the n solutions of this level (B doubles each)
  int n;
  int ext;            // a block exists beyond the level's last one (sharded chains: the next rank's separator)
  int *flag;          // set to 1 when a pivot is not positive
};

struct UpBwdArgs {
  const double *blk;  // the factor records the forward launch left behind
  double *x;          // n solutions of this level (+ one slot)
  const double *xup;  // solutions of the next level: entry g = first block of group g
  int n;
  int ext;
};

// B in {4, 6, 12}; G = blocks per group: kUpG, or 4 for the level that the row-layout level-0 kernels fold into their tail
// (then only its back-substitution, and the forward launch of the unfused path, come through here).
// Return a hipError_t (as int); the launches are asynchronous on `st`.
int upper_forward(int B, int G, bool top, const UpFwdArgs &a, hipStream_t st);
int upper_backward(int B, int G, const UpBwdArgs &a, hipStream_t st);

}  // namespace gps
