// tiled_host.cpp -- a C++ host for the tiled single-map mode (SURVEY.md 8e mode 2, BASELINE config 5): one process
// per GPU, RCCL over xGMI through include/rna_rccl.h.  Every rank applies the whole ray batch to its window
// (rna_himm_set_window), refreshes the halo frame for VFH+, steps its poses, hands its changed tiles to everybody and
// serves its shard of the A* queries.
//
//   tiled_host <rank> <world> <id-file> [grid=2048] [rounds=3]
// Rank 0 writes the ncclUniqueId to <id-file>, the others wait for it (a launcher with a shared directory is all the
// bootstrap needs).  world = 1 runs the same code path on one GPU (every exchange degenerates to a no-op).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "../include/rna_rccl.h"

#define OK(call) do { const int _rc = (call); if (_rc != 0) { std::fprintf(stderr, "%s failed: %d (line %d)\n", #call, _rc, __LINE__); return 1; } } while (0)

int main(int argc, char** argv) {
  if (argc < 4) { std::fprintf(stderr, "usage: tiled_host <rank> <world> <id-file> [grid] [rounds]\n"); return 2; }
  const int rank = std::atoi(argv[1]), world = std::atoi(argv[2]);
  const char* id_file = argv[3];
  const int n = argc > 4 ? std::atoi(argv[4]) : 2048, rounds = argc > 5 ? std::atoi(argv[5]) : 3;
  int ndev = 0;
  OK(hipGetDeviceCount(&ndev));
  if (ndev <= 0) { std::fprintf(stderr, "tiled_host needs a GPU\n"); return 1; }
  const int dev = rank % ndev;
  OK(hipSetDevice(dev));

  ncclUniqueId id;
  if (rank == 0) {
    OK(ncclGetUniqueId(&id));
    FILE* f = std::fopen(id_file, "wb");
    if (!f || std::fwrite(&id, sizeof(id), 1, f) != 1) return 1;
    std::fclose(f);
  } else {
    for (int tries = 0;; ++tries) {
      FILE* f = std::fopen(id_file, "rb");
      if (f && std::fread(&id, sizeof(id), 1, f) == 1) { std::fclose(f); break; }
      if (f) std::fclose(f);
      if (tries > 600) { std::fprintf(stderr, "no ncclUniqueId in %s\n", id_file); return 1; }
      std::this_thread::sleep_for(std::chrono::milliseconds(100));
    }
  }
  ncclComm_t comm;
  OK(ncclCommInitRank(&comm, world, id, rank));

  const double res = 0.05, len = n * res;
  rna_engine* e = nullptr;
  OK(rna_create(&e, len, len, res, 0.0, 0.0, dev));
  rna_tile_layout L;
  OK(rna_tile_layout_for_world(n, n, world, &L));
  int32_t w[4];
  OK(rna_tile_window(&L, rank, w));
  const int halo = rna_vfh_halo_cells(res);
  OK(rna_layer_fill(e, RNA_LAYER_LASER, 0.0f));
  OK(rna_compose_master(e, 1));
  OK(rna_himm_set_window(e, w[0], w[2], w[1], w[3]));

  rna_vfh_params vp;
  rna_vfh_default_params(&vp);
  // this rank's robots: a row of poses through the middle of its window
  std::vector<rna_pose> poses;
  for (int k = 0; k < 16; ++k) {
    double pos[2];
    rna_get_position(e, w[0] + w[1] / 2, w[2] + (k + 1) * w[3] / 18, pos);
    rna_pose p = {pos[0], pos[1], 0.1 * k, 0.2, 0, 90.0f, 3000.0f, 250.0f};
    poses.push_back(p);
  }
  OK(rna_vfh_init(e, &vp, (int)poses.size()));
  std::vector<rna_vfh_out> vout(poses.size());
  std::vector<float> origin(poses.size() * 72), hist(poses.size() * 72);

  size_t halo_bytes = 0, gather_bytes = 0;
  int found = 0;
  std::srand(5);   // the same ray batch on every rank
  for (int r = 0; r < rounds; ++r) {
    std::vector<rna_ray> rays(20000);
    for (size_t k = 0; k < rays.size(); ++k) {
      const double ox = (std::rand() / (double)RAND_MAX - 0.5) * (len - 14.0), oy = (std::rand() / (double)RAND_MAX - 0.5) * (len - 14.0);
      const double a = 6.283185 * std::rand() / RAND_MAX, l = 1.0 + 5.0 * std::rand() / RAND_MAX;
      rna_ray ray = {ox, oy, ox + l * std::cos(a), oy + l * std::sin(a), (k % 5) == 0, 0};
      rays[k] = ray;
    }
    OK(rna_update_map(e, rays.data(), (int)rays.size(), 0));                       // windowed HIMM + fused compose
    size_t got = 0;
    OK(rna_rccl_exchange_halo(e, comm, &L, rank, RNA_LAYER_MASTER, halo, 1, &got));
    halo_bytes += got;
    OK(rna_vfh_step_batch(e, poses.data(), (int)poses.size(), vout.data(), origin.data(), hist.data()));
    OK(rna_rccl_gather_dirty(e, comm, &L, rank, RNA_LAYER_LASER, RNA_LAYER_MASTER, &got));
    gather_bytes += got;
    OK(rna_compose_master(e, 0));                                                   // refresh the masks of the received tiles
    // this rank's shard of 32 queries across the whole map
    std::vector<rna_astar_query> q;
    for (int k = rank; k < 32; k += world) {
      rna_astar_query qq = {(k * 37 + 11) % n + ((k * 53 + 7) % n) * n, ((k * 91 + 300) % n) + ((k * 17 + 900) % n) * n};
      q.push_back(qq);
    }
    std::vector<int32_t> paths(q.size() * 16384);
    std::vector<rna_astar_result> res_(q.size());
    OK(rna_astar_configure(e, 32, 0, 0));
    OK(rna_astar_batch(e, q.data(), (int)q.size(), paths.data(), 16384, res_.data()));
    for (size_t k = 0; k < q.size(); ++k) found += res_[k].status == 0;
  }
  std::printf("tiled_host rank %d/%d OK: window (%d+%d, %d+%d), halo %d cells, %zu halo bytes, %zu gathered bytes, %d paths\n", rank, world,
              w[0], w[1], w[2], w[3], halo, halo_bytes, gather_bytes, found);
  rna_rccl_release();   // the exchange's staging buffers
  rna_destroy(e);
  ncclCommDestroy(comm);
  return 0;
}
