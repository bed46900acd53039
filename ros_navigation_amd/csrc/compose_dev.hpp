// compose_dev.hpp -- device code of engine.hip's compose_nbr_tiles_kernel (a header of its own since round 5's fused map-update
// kernel, which was measured slower and removed, shared it):
// MapProvider::updateMap's compose step and the neighbour-mask refresh of ONE 64 x 64 tile.
#pragma once
#include "engine.hpp"

namespace rna {

// GlobalPlanner::ifBlocked predicate (mc/include/move_control/map_global_planner.h:47-50):
// blocked iff the master value is finite-or-inf (not NaN) and > 0.
__device__ __forceinline__ bool cell_blocked(float v) { return !(v != v) && v > 0.0f; }

constexpr int COMPOSE_BLK_BYTES = (TILE + 2) * (TILE + 2);   // LDS of a workgroup: the tile's blocked bytes with a 1-cell ring

// One tile tt, all threads of the workgroup (any size): the block of a dirty tile copies laser -> master
// (composeMasterMapFromLayerdMap, mc/src/map_provider.cpp:216-223, restricted to where the two layers differ), and every
// block whose tile is dirty or touches a dirty tile recomputes its cells' masks.  A cell of a DIRTY tile is read from the
// laser layer -- what master holds there once the launch has finished; its own block may still be copying --, a cell of
// a clean tile from master.  The block also clears its byte of `next_dirty`.  `blk`: COMPOSE_BLK_BYTES of LDS, [jj][ii],
// ii fastest; out of map = blocked.  Two workgroup barriers inside (all threads must call, tt uniform).
__device__ __forceinline__ void compose_nbr_tile(uint8_t* __restrict__ blk, int tt, uint8_t* __restrict__ nbr, float* __restrict__ master,
                                                 const float* __restrict__ laser, const unsigned* __restrict__ dirty,
                                                 unsigned* __restrict__ next_dirty, int rows, int cols, int tiles_i, int tiles_j) {
  const int ti = tt % tiles_i, tj = tt / tiles_i;
  const unsigned char* dflag = reinterpret_cast<const unsigned char*>(dirty);
  unsigned dmask = 0u;   // bit (dj + 1) * 3 + (di + 1): that neighbouring tile is dirty
  for (int dj = -1; dj <= 1; ++dj)
    for (int di = -1; di <= 1; ++di) {
      const int a = ti + di, b = tj + dj;
      if (a < 0 || b < 0 || a >= tiles_i || b >= tiles_j) continue;
      if (dflag[b * tiles_i + a]) dmask |= 1u << ((dj + 1) * 3 + di + 1);
    }
  if (threadIdx.x == 0) reinterpret_cast<volatile unsigned char*>(next_dirty)[tj * tiles_i + ti] = 0;
  if (!dmask) return;   // (uniform across the workgroup)
  const bool own = (dmask >> 4) & 1u;
  __syncthreads();        // the previous tile's block has been read
  const int i0 = ti * TILE - 1, j0 = tj * TILE - 1;
  for (int k = threadIdx.x; k < (TILE + 2) * (TILE + 2); k += blockDim.x) {
    const int ii = k % (TILE + 2), jj = k / (TILE + 2);
    const int i = i0 + ii, j = j0 + jj;
    uint8_t b = 1;
    if (i >= 0 && j >= 0 && i < rows && j < cols) {
      const int di = ii == 0 ? 0 : (ii == TILE + 1 ? 2 : 1), dj = jj == 0 ? 0 : (jj == TILE + 1 ? 2 : 1);
      const size_t lin = (size_t)j * rows + i;
      const bool from_laser = (dmask >> (dj * 3 + di)) & 1u;
      const float v = from_laser ? laser[lin] : master[lin];
      if (own && di == 1 && dj == 1) master[lin] = v;   // the compose itself
      b = cell_blocked(v) ? 1 : 0;
    }
    blk[k] = b;
  }
  __syncthreads();
  for (int k = threadIdx.x; k < TILE * TILE; k += blockDim.x) {
    const int li = k & (TILE - 1), lj = k >> 6;
    const int i = ti * TILE + li, j = tj * TILE + lj;
    if (i >= rows || j >= cols) continue;
    const uint8_t* c = &blk[(lj + 1) * (TILE + 2) + (li + 1)];
    constexpr int S = TILE + 2;
    unsigned m = 0;
    if (!c[0]) {
      const bool up = !c[-1], dn = !c[1], lf = !c[-S], rt = !c[S];
      if (lf && up && !c[-S - 1]) m |= 1u;        // (-1,-1)
      if (lf) m |= 2u;                            // ( 0,-1)
      if (lf && dn && !c[-S + 1]) m |= 4u;        // ( 1,-1)
      if (up) m |= 8u;                            // (-1, 0)
      if (dn) m |= 16u;                           // ( 1, 0)
      if (rt && up && !c[S - 1]) m |= 32u;        // (-1, 1)
      if (rt) m |= 64u;                           // ( 0, 1)
      if (rt && dn && !c[S + 1]) m |= 128u;       // ( 1, 1)
    }
    nbr[(size_t)j * rows + i] = (uint8_t)m;
  }
}

}  // namespace rna
