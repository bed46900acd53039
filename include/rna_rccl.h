/*
 * rna_rccl.h -- C ABI of the tiled single-map exchange over RCCL (librna_rccl.so; links librna.so and librccl.so).
 *
 * SURVEY.md 8e mode 2 / BASELINE config 5: one map cut into ti x tj windows, one per GPU (rank = a*tj + b owns
 * window (a, b)); every GPU holds the whole layer but is authoritative for its window only.  The reference has no
 * multi-GPU path (single process, single map) -- these calls are what a C++ host (one process per GPU, one
 * ncclComm_t) runs after each rna_update_map to keep the replicas consistent:
 *   rna_rccl_exchange_halo   the halo-cell frame around the window, from its owners, for VFH+ submaps that reach past
 *                            the window (two rounds of ncclSend/ncclRecv: row strips, then column strips that carry
 *                            the corners -- 4 messages per rank instead of 8);
 *   rna_rccl_gather_dirty    every rank's 64 x 64 tiles that its last map update changed, all-gathered and written
 *                            into layer_a (and layer_b) of every other rank -- what grid A* needs before the queries
 *                            are sharded; the tile lists are built and consumed on the device, only `world` counters
 *                            (4 B each) visit the host to size the collective;
 *   rna_rccl_gather_layer    the whole owner windows instead (first synchronisation, or after a wholesale upload).
 * Collectives run on the engine's stream; `comm` is the caller's ncclComm_t passed as void*.  Python hosts use
 * ros_navigation_amd/dist.py (torch.distributed) for the same exchange.
 */
#ifndef RNA_RCCL_H
#define RNA_RCCL_H

#include "rna.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { int32_t rows, cols, ti, tj; } rna_tile_layout;

/* BASELINE config 5's shape: 2 x world/2 windows (8 GPUs -> 2 x 4); odd world sizes -> 1 x world */
int rna_tile_layout_for_world(int rows, int cols, int world, rna_tile_layout* out);
/* window[4] = {i0, ni, j0, nj} (buffer indices) of `rank`: balanced contiguous shares, as dist.TileLayout.window */
int rna_tile_window(const rna_tile_layout* layout, int rank, int32_t window[4]);
/* Steerer's 1.5 m submap (mc/src/steerer.cpp:128-135) reaches ceil(0.75 m / resolution) + 1 cells past its centre */
int rna_vfh_halo_cells(double resolution);

int rna_rccl_exchange_halo(rna_engine* e, void* comm, const rna_tile_layout* layout, int rank, int layer, int halo, int tracked,
                           size_t* bytes_received);
int rna_rccl_gather_dirty(rna_engine* e, void* comm, const rna_tile_layout* layout, int rank, int layer_a, int layer_b,
                          size_t* bytes_received);
int rna_rccl_gather_layer(rna_engine* e, void* comm, const rna_tile_layout* layout, int rank, int layer, size_t* bytes_received);
/* The staging buffers the three calls above grow on demand (per calling thread) go back to the device allocator. */
int rna_rccl_release(void);

#ifdef __cplusplus
}
#endif
#endif
