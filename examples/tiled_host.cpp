// tiled_host.cpp -- a C++ host for the tiled single-map mode (SURVEY.md 8e mode 2, BASELINE config 5): one process
// per GPU, RCCL over xGMI through include/rna_rccl.h.  Every rank applies the whole ray batch to its window
// (rna_himm_set_window), refreshes the halo frame for VFH+, steps its poses, hands its changed tiles to everybody and
// serves its shard of the A* queries.
//
//   tiled_host <rank> <world> <id-file> [grid=2048] [rounds=3] [session]
// Rank 0 publishes the ncclUniqueId in <id-file>, the others wait for the file of THEIR job (id_bootstrap.hpp: a file
// left by an earlier job carries another session token and is ignored).  world = 1 runs the same code path on one GPU
// (every exchange degenerates to a no-op).
//
// The program says where it is: every phase goes to stderr, unbuffered, and a watchdog thread ends the process with
// exit code 3 and the name of the phase when a phase makes no progress for RNA_TILED_WATCHDOG_S seconds (default 60; five
// times that while the HIP and RCCL runtimes start up)
// -- a hang names its phase instead of eating a test's timeout.  RNA_TILED_DUMP=<file> writes every input and output
// (rays, poses, VFH+ results, queries, paths, the final layers) so that a test can check them against the oracle.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include <unistd.h>

#include "../include/rna_rccl.h"
#include "id_bootstrap.hpp"

namespace {

std::atomic<unsigned long long> g_beat{0};
std::atomic<int> g_slack{1};   // the runtimes' own start-up phases get 5 x the watchdog limit, see phase()
const char* volatile g_phase = "start";
int g_rank = 0;
int g_round = -1;
const std::chrono::steady_clock::time_point g_t0 = std::chrono::steady_clock::now();

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now() - g_t0).count(); }

// `slow_start`: a phase that first touches the HIP / RCCL runtimes (on a box whose image was just pulled, paging in
// librccl.so's 570 MB of code objects takes minutes -- slow, not stuck)
void phase(const char* name, bool slow_start = false) {
  g_phase = name;
  g_slack.store(slow_start ? 5 : 1);
  g_beat.fetch_add(1);
  std::fprintf(stderr, "[tiled_host %d %8.3f s] round %d: %s\n", g_rank, now_s(), g_round, name);
}

void watchdog(double limit_s) {
  unsigned long long seen = g_beat.load();
  double since = now_s();
  for (;;) {
    std::this_thread::sleep_for(std::chrono::milliseconds(250));
    const unsigned long long b = g_beat.load();
    if (b != seen) { seen = b; since = now_s(); continue; }
    const double limit = limit_s * g_slack.load();
    if (now_s() - since > limit) {
      std::fprintf(stderr, "[tiled_host %d] WATCHDOG: no progress for %.0f s in phase \"%s\" of round %d -- giving up (exit 3)\n", g_rank,
                   limit, g_phase, g_round);
      _exit(3);   // (never a re-exec: the process has initialised the GPU)
    }
  }
}

FILE* g_dump = nullptr;
void dump(int tag, int round, const void* p, size_t bytes) {
  if (!g_dump) return;
  const int32_t head[2] = {tag, round};
  const int64_t n = (int64_t)bytes;
  std::fwrite(head, sizeof(head), 1, g_dump);
  std::fwrite(&n, sizeof(n), 1, g_dump);
  if (bytes) std::fwrite(p, bytes, 1, g_dump);
}

}  // namespace

#define OK(call) do { const int _rc = (call); if (_rc != 0) { std::fprintf(stderr, "[tiled_host %d] %s failed: %d (line %d, phase %s)\n", g_rank, #call, _rc, __LINE__, g_phase); return 1; } } while (0)

int main(int argc, char** argv) {
  if (argc < 4) { std::fprintf(stderr, "usage: tiled_host <rank> <world> <id-file> [grid] [rounds] [session]\n"); return 2; }
  std::setvbuf(stdout, nullptr, _IONBF, 0);
  std::setvbuf(stderr, nullptr, _IONBF, 0);
  const int rank = g_rank = std::atoi(argv[1]), world = std::atoi(argv[2]);
  const char* id_file = argv[3];
  const int n = argc > 4 ? std::atoi(argv[4]) : 2048, rounds = argc > 5 ? std::atoi(argv[5]) : 3;
  const char* session = argc > 6 ? argv[6] : std::getenv("RNA_TILED_SESSION");
  if (world > 1 && !(session && *session)) {   // (id_bootstrap.hpp: the default token cannot tell two jobs of one launcher apart)
    std::fprintf(stderr, "tiled_host: a job of %d ranks needs a session token that is fresh per job (7th argument or RNA_TILED_SESSION), the same for all its ranks\n", world);
    return 2;
  }
  // The engine pipelines A* batches over CU-masked streams (rna_astar_set_pipeline_depth): each wants a hardware queue
  // of its own, and HIP multiplexes all streams over GPU_MAX_HW_QUEUES (default 4) of them.  Must be in the environment
  // before the first HIP call (INTEGRATION.md "Runtime notes").
  if (!std::getenv("RNA_TILED_KEEP_HWQ")) setenv("GPU_MAX_HW_QUEUES", "8", 0);
  const double wd = std::getenv("RNA_TILED_WATCHDOG_S") ? std::atof(std::getenv("RNA_TILED_WATCHDOG_S")) : 60.0;
  if (wd > 0) std::thread(watchdog, wd).detach();
  if (const char* d = std::getenv("RNA_TILED_DUMP")) {
    g_dump = std::fopen(d, "wb");
    if (!g_dump) { std::fprintf(stderr, "cannot write %s\n", d); return 1; }
  }

  phase("hipGetDeviceCount", true);
  int ndev = 0;
  OK(hipGetDeviceCount(&ndev));
  if (ndev <= 0) { std::fprintf(stderr, "tiled_host needs a GPU\n"); return 1; }
  const int dev = rank % ndev;
  OK(hipSetDevice(dev));

  phase("ncclUniqueId", true);
  const uint64_t token = rna_bootstrap::session_token(session);
  ncclUniqueId id;
  if (rank == 0) {
    OK(ncclGetUniqueId(&id));
    if (!rna_bootstrap::publish(id_file, token, &id, sizeof(id))) { std::fprintf(stderr, "cannot write %s\n", id_file); return 1; }
  } else if (!rna_bootstrap::fetch(id_file, token, &id, sizeof(id), wd > 0 ? wd - 5 : 60.0)) {
    std::fprintf(stderr, "[tiled_host %d] no ncclUniqueId of this session in %s\n", rank, id_file);
    return 1;
  }
  phase("ncclCommInitRank", true);
  ncclComm_t comm;
  OK(ncclCommInitRank(&comm, world, id, rank));

  phase("rna_create", true);
  const double res = 0.05, len = n * res;
  rna_engine* e = nullptr;
  OK(rna_create(&e, len, len, res, 0.0, 0.0, dev));
  if (const char* d = std::getenv("RNA_TILED_DEPTH")) OK(rna_astar_set_pipeline_depth(e, std::atoi(d)));
  rna_tile_layout L;
  OK(rna_tile_layout_for_world(n, n, world, &L));
  int32_t w[4];
  OK(rna_tile_window(&L, rank, w));
  const int halo = rna_vfh_halo_cells(res);
  OK(rna_layer_fill(e, RNA_LAYER_LASER, 0.0f));
  OK(rna_compose_master(e, 1));
  OK(rna_himm_set_window(e, w[0], w[2], w[1], w[3]));
  {
    const int32_t head[10] = {n, rounds, rank, world, w[0], w[1], w[2], w[3], halo, 0};
    dump(0, -1, head, sizeof(head));
  }

  phase("rna_vfh_init");
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
  OK(rna_astar_configure(e, 32, 0, 0));

  size_t halo_bytes = 0, gather_bytes = 0;
  int found = 0;
  std::srand(5);   // the same ray batch on every rank
  for (int r = 0; r < rounds; ++r) {
    g_round = r;
    std::vector<rna_ray> rays(20000);
    for (size_t k = 0; k < rays.size(); ++k) {
      const double ox = (std::rand() / (double)RAND_MAX - 0.5) * (len - 14.0), oy = (std::rand() / (double)RAND_MAX - 0.5) * (len - 14.0);
      const double a = 6.283185 * std::rand() / RAND_MAX, l = 1.0 + 5.0 * std::rand() / RAND_MAX;
      rna_ray ray = {ox, oy, ox + l * std::cos(a), oy + l * std::sin(a), (k % 5) == 0, 0};
      rays[k] = ray;
    }
    dump(1, r, rays.data(), rays.size() * sizeof(rna_ray));
    phase("rna_update_map");
    OK(rna_update_map(e, rays.data(), (int)rays.size(), 0));                       // windowed HIMM + fused compose
    size_t got = 0;
    phase("rna_rccl_exchange_halo");
    OK(rna_rccl_exchange_halo(e, comm, &L, rank, RNA_LAYER_MASTER, halo, 1, &got));
    halo_bytes += got;
    phase("rna_vfh_step_batch");
    OK(rna_vfh_step_batch(e, poses.data(), (int)poses.size(), vout.data(), origin.data(), hist.data()));
    dump(2, r, poses.data(), poses.size() * sizeof(rna_pose));
    dump(3, r, vout.data(), vout.size() * sizeof(rna_vfh_out));
    dump(4, r, origin.data(), origin.size() * sizeof(float));
    dump(5, r, hist.data(), hist.size() * sizeof(float));
    phase("rna_rccl_gather_dirty");
    OK(rna_rccl_gather_dirty(e, comm, &L, rank, RNA_LAYER_LASER, RNA_LAYER_MASTER, &got));
    gather_bytes += got;
    phase("rna_compose_master");
    OK(rna_compose_master(e, 0));                                                   // refresh the masks of the received tiles
    // this rank's shard of 32 queries across the whole map
    std::vector<rna_astar_query> q;
    for (int k = rank; k < 32; k += world) {
      rna_astar_query qq = {(k * 37 + 11) % n + ((k * 53 + 7) % n) * n, ((k * 91 + 300) % n) + ((k * 17 + 900) % n) * n};
      q.push_back(qq);
    }
    const int max_len = 16384;
    std::vector<int32_t> paths(q.size() * max_len);
    std::vector<rna_astar_result> res_(q.size());
    phase("rna_astar_batch");
    if (std::getenv("RNA_TILED_RECONFIGURE")) OK(rna_astar_configure(e, 32, 0, 0));   // (developer knob: tear the stages down every round)
    OK(rna_astar_batch(e, q.data(), (int)q.size(), paths.data(), max_len, res_.data()));
    dump(6, r, q.data(), q.size() * sizeof(rna_astar_query));
    dump(7, r, res_.data(), res_.size() * sizeof(rna_astar_result));
    for (size_t k = 0; k < q.size(); ++k) {
      found += res_[k].status == 0;
      dump(8, r, paths.data() + k * max_len, res_[k].status == 0 ? (size_t)res_[k].path_len * sizeof(int32_t) : 0);
    }
  }
  g_round = rounds;
  if (g_dump) {
    phase("rna_layer_download");
    std::vector<float> layer((size_t)n * n);
    OK(rna_layer_download(e, RNA_LAYER_LASER, layer.data(), layer.size()));
    dump(9, rounds, layer.data(), layer.size() * sizeof(float));
    OK(rna_layer_download(e, RNA_LAYER_MASTER, layer.data(), layer.size()));
    dump(10, rounds, layer.data(), layer.size() * sizeof(float));
    std::fclose(g_dump);
    g_dump = nullptr;
  }
  phase("release");
  rna_rccl_release();   // the exchange's staging buffers
  rna_destroy(e);
  ncclCommDestroy(comm);
  std::printf("tiled_host rank %d/%d OK: window (%d+%d, %d+%d), halo %d cells, %zu halo bytes, %zu gathered bytes, %d paths\n", rank, world,
              w[0], w[1], w[2], w[3], halo, halo_bytes, gather_bytes, found);
  return 0;
}
