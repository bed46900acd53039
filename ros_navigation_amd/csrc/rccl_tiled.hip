// rccl_tiled.hip -- the tiled single-map exchange of include/rna_rccl.h on RCCL (librna_rccl.so).  One process per
// GPU; collectives and the pack / unpack kernels run on the engine's stream, so an exchange is ordered after the map
// update that produced the data and before the VFH+ / A* work that consumes it without any host synchronisation
// other than the one 4*world-byte read that sizes the dirty-tile all-gather.
// xGMI is point-to-point: the halo strips go to their (up to 4) neighbours directly (ncclSend/ncclRecv in one group
// per round), the dirty tiles as one all-gather of equal-sized contributions (RCCL's ring over the 7 links).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cmath>
#include <string>
#include <vector>

#include "../../include/rna_rccl.h"

namespace {

constexpr int TILE = 64;

void shard_bounds(int n, int rank, int world, int& lo, int& hi) {   // dist.shard_bounds
  const int base = n / world, rem = n % world;
  lo = rank * base + std::min(rank, rem);
  hi = lo + base + (rank < rem ? 1 : 0);
}
int rank_of(const rna_tile_layout* L, int a, int b) { return (a >= 0 && a < L->ti && b >= 0 && b < L->tj) ? a * L->tj + b : -1; }

struct DevBuf {   // grows on demand; freed by rna_rccl_release (or when the next larger request arrives)
  void* p = nullptr;
  size_t cap = 0;
  int ensure(size_t bytes) {
    if (bytes <= cap) return RNA_OK;
    if (p) (void)hipFree(p);
    p = nullptr; cap = 0;
    if (hipMalloc(&p, bytes) != hipSuccess) return RNA_ENOMEM;
    cap = bytes;
    return RNA_OK;
  }
  void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};
thread_local DevBuf t_send[2], t_recv[2], t_list, t_lists, t_counts, t_data, t_all;

#define RCCL_OK(call) do { if ((call) != ncclSuccess) return RNA_EHIP; } while (0)
#define HIP_OK(call) do { if ((call) != hipSuccess) return RNA_EHIP; } while (0)
#define RNA_TRY(call) do { const int _rc = (call); if (_rc != RNA_OK) return _rc; } while (0)

struct Strip { int peer; int send[4]; int recv[4]; };   // boxes: i0, ni, j0, nj

int one_round(rna_engine* e, ncclComm_t comm, hipStream_t st, int layer, const Strip* strips, int n, bool tracked, size_t* got) {
  for (int k = 0; k < n; ++k) {
    if (strips[k].peer < 0) continue;
    const Strip& s = strips[k];
    RNA_TRY(t_send[k].ensure((size_t)s.send[1] * s.send[3] * sizeof(float)));
    RNA_TRY(t_recv[k].ensure((size_t)s.recv[1] * s.recv[3] * sizeof(float)));
    RNA_TRY(rna_layer_pack_region(e, layer, s.send[0], s.send[1], s.send[2], s.send[3], (float*)t_send[k].p));
  }
  RCCL_OK(ncclGroupStart());
  for (int k = 0; k < n; ++k) {
    if (strips[k].peer < 0) continue;
    const Strip& s = strips[k];
    RCCL_OK(ncclSend(t_send[k].p, (size_t)s.send[1] * s.send[3], ncclFloat, s.peer, comm, st));
    RCCL_OK(ncclRecv(t_recv[k].p, (size_t)s.recv[1] * s.recv[3], ncclFloat, s.peer, comm, st));
  }
  RCCL_OK(ncclGroupEnd());
  for (int k = 0; k < n; ++k) {
    if (strips[k].peer < 0) continue;
    const Strip& s = strips[k];
    if (tracked) RNA_TRY(rna_layer_unpack_region_tracked(e, layer, s.recv[0], s.recv[1], s.recv[2], s.recv[3], (const float*)t_recv[k].p));
    else RNA_TRY(rna_layer_unpack_region(e, layer, s.recv[0], s.recv[1], s.recv[2], s.recv[3], (const float*)t_recv[k].p));
    *got += (size_t)s.recv[1] * s.recv[3] * sizeof(float);
  }
  return RNA_OK;
}

}  // namespace

extern "C" int rna_tile_layout_for_world(int rows, int cols, int world, rna_tile_layout* out) {
  if (!out || rows <= 0 || cols <= 0 || world <= 0) return RNA_EINVAL;
  out->rows = rows; out->cols = cols;
  out->ti = world % 2 == 0 ? 2 : 1;
  out->tj = world / out->ti;
  if (out->ti > rows || out->tj > cols) return RNA_EINVAL;
  return RNA_OK;
}

extern "C" int rna_tile_window(const rna_tile_layout* L, int rank, int32_t w[4]) {
  if (!L || !w || rank < 0 || rank >= L->ti * L->tj) return RNA_EINVAL;
  int i0, i1, j0, j1;
  shard_bounds(L->rows, rank / L->tj, L->ti, i0, i1);
  shard_bounds(L->cols, rank % L->tj, L->tj, j0, j1);
  w[0] = i0; w[1] = i1 - i0; w[2] = j0; w[3] = j1 - j0;
  return RNA_OK;
}

extern "C" int rna_vfh_halo_cells(double resolution) { return (int)std::ceil(0.5 * 1.5 / resolution) + 1; }

extern "C" int rna_rccl_exchange_halo(rna_engine* e, void* comm_, const rna_tile_layout* L, int rank, int layer, int halo,
                                      int tracked, size_t* bytes_received) {
  if (!e || !comm_ || !L || halo <= 0) return RNA_EINVAL;
  ncclComm_t comm = (ncclComm_t)comm_;
  hipStream_t st = (hipStream_t)rna_stream(e);
  int32_t w[4];
  RNA_TRY(rna_tile_window(L, rank, w));
  const int a = rank / L->tj, b = rank % L->tj, i0 = w[0], ni = w[1], j0 = w[2], nj = w[3];
  size_t got = 0;
  const int up = rank_of(L, a - 1, b), down = rank_of(L, a + 1, b), left = rank_of(L, a, b - 1), right = rank_of(L, a, b + 1);
  const int peers[4] = {up, down, left, right};
  for (int k = 0; k < 4; ++k)
    if (peers[k] >= 0) {
      int32_t pw[4];
      RNA_TRY(rna_tile_window(L, peers[k], pw));
      if ((k < 2 ? pw[1] : pw[3]) < halo) return RNA_EINVAL;   // halo wider than a neighbouring window
    }
  if (ni < halo || nj < halo) return RNA_EINVAL;
  // round 1: row strips (i direction) of the own columns
  const Strip r1[2] = {{up, {i0, halo, j0, nj}, {i0 - halo, halo, j0, nj}}, {down, {i0 + ni - halo, halo, j0, nj}, {i0 + ni, halo, j0, nj}}};
  RNA_TRY(one_round(e, comm, st, layer, r1, 2, tracked != 0, &got));
  // round 2: column strips that include the rows just received (they carry the corners)
  const int ie0 = i0 - (up >= 0 ? halo : 0), ie1 = i0 + ni + (down >= 0 ? halo : 0);
  const Strip r2[2] = {{left, {ie0, ie1 - ie0, j0, halo}, {ie0, ie1 - ie0, j0 - halo, halo}},
                       {right, {ie0, ie1 - ie0, j0 + nj - halo, halo}, {ie0, ie1 - ie0, j0 + nj, halo}}};
  RNA_TRY(one_round(e, comm, st, layer, r2, 2, tracked != 0, &got));
  if (bytes_received) *bytes_received = got;
  return RNA_OK;
}

extern "C" int rna_rccl_gather_layer(rna_engine* e, void* comm_, const rna_tile_layout* L, int rank, int layer, size_t* bytes_received) {
  if (!e || !comm_ || !L) return RNA_EINVAL;
  ncclComm_t comm = (ncclComm_t)comm_;
  hipStream_t st = (hipStream_t)rna_stream(e);
  const int world = L->ti * L->tj;
  if (bytes_received) *bytes_received = 0;
  if (world == 1) return RNA_OK;
  std::vector<int32_t> wins(4 * (size_t)world);
  size_t cap = 0;
  for (int r = 0; r < world; ++r) {
    RNA_TRY(rna_tile_window(L, r, &wins[4 * r]));
    cap = std::max(cap, (size_t)wins[4 * r + 1] * wins[4 * r + 3]);
  }
  RNA_TRY(t_data.ensure(cap * sizeof(float)));
  RNA_TRY(t_all.ensure(cap * sizeof(float) * world));
  const int32_t* w = &wins[4 * rank];
  RNA_TRY(rna_layer_pack_region(e, layer, w[0], w[1], w[2], w[3], (float*)t_data.p));
  RCCL_OK(ncclAllGather(t_data.p, t_all.p, cap, ncclFloat, comm, st));
  size_t got = 0;
  for (int r = 0; r < world; ++r) {
    if (r == rank) continue;
    const int32_t* pw = &wins[4 * r];
    RNA_TRY(rna_layer_unpack_region(e, layer, pw[0], pw[1], pw[2], pw[3], (const float*)t_all.p + (size_t)r * cap));
    got += (size_t)pw[1] * pw[3] * sizeof(float);
  }
  if (bytes_received) *bytes_received = got;
  return RNA_OK;
}

extern "C" int rna_rccl_gather_dirty(rna_engine* e, void* comm_, const rna_tile_layout* L, int rank, int layer_a, int layer_b,
                                     size_t* bytes_received) {
  if (!e || !comm_ || !L) return RNA_EINVAL;
  ncclComm_t comm = (ncclComm_t)comm_;
  hipStream_t st = (hipStream_t)rna_stream(e);
  const int world = L->ti * L->tj;
  if (bytes_received) *bytes_received = 0;
  if (world == 1) return RNA_OK;
  const size_t ntile = (size_t)((L->rows + TILE - 1) / TILE) * ((L->cols + TILE - 1) / TILE);
  int32_t w[4];
  RNA_TRY(rna_tile_window(L, rank, w));
  // 1. this rank's changed tiles, compacted on the device; 2. everybody's counts
  RNA_TRY(t_list.ensure(ntile * sizeof(int32_t)));
  RNA_TRY(t_counts.ensure(sizeof(int) * (size_t)(world + 1)));
  int* count_dev = (int*)t_counts.p + world;
  RNA_TRY(rna_last_dirty_tiles_device(e, w[0], w[1], w[2], w[3], (int32_t*)t_list.p, count_dev));
  RCCL_OK(ncclAllGather(count_dev, t_counts.p, 1, ncclInt32, comm, st));
  std::vector<int> counts(world);
  HIP_OK(hipMemcpyAsync(counts.data(), t_counts.p, sizeof(int) * (size_t)world, hipMemcpyDeviceToHost, st));
  HIP_OK(hipStreamSynchronize(st));   // 4 * world bytes: the one host visit, it sizes the collective below
  const size_t cap = (size_t)*std::max_element(counts.begin(), counts.end());
  if (cap == 0) return RNA_OK;
  // 3. tile lists and tile data, padded to the largest contribution
  RNA_TRY(t_lists.ensure(cap * sizeof(int32_t) * world));
  RNA_TRY(t_data.ensure(cap * TILE * TILE * sizeof(float)));
  RNA_TRY(t_all.ensure(cap * TILE * TILE * sizeof(float) * world));
  RCCL_OK(ncclAllGather(t_list.p, t_lists.p, cap, ncclInt32, comm, st));
  RNA_TRY(rna_layer_pack_tiles_device(e, layer_a, (const int32_t*)t_list.p, counts[rank], w[0], w[1], w[2], w[3], (float*)t_data.p));
  RCCL_OK(ncclAllGather(t_data.p, t_all.p, cap * TILE * TILE, ncclFloat, comm, st));
  size_t got = 0;
  for (int r = 0; r < world; ++r) {
    if (r == rank || counts[r] == 0) continue;
    int32_t pw[4];
    RNA_TRY(rna_tile_window(L, r, pw));
    RNA_TRY(rna_layers_unpack_tiles_device(e, layer_a, layer_b, (const int32_t*)t_lists.p + (size_t)r * cap, counts[r], pw[0], pw[1],
                                           pw[2], pw[3], (const float*)t_all.p + (size_t)r * cap * TILE * TILE));
    got += (size_t)counts[r] * TILE * TILE * sizeof(float);
  }
  if (bytes_received) *bytes_received = got;
  return RNA_OK;
}

// the calling thread's staging buffers (strips, tile lists, gathered tiles) back to the device allocator: a host calls
// it when it stops exchanging (before ncclCommDestroy), every thread that drove an exchange for itself
extern "C" int rna_rccl_release(void) {
  for (int k = 0; k < 2; ++k) { t_send[k].release(); t_recv[k].release(); }
  t_list.release(); t_lists.release(); t_counts.release(); t_data.release(); t_all.release();
  return RNA_OK;
}
