// engine.hip -- engine lifetime, GridMap layers, compose-master, neighbour masks, GridMap::move.
// Reference call sites replaced (mc/ = move_control, gmc/ = grid_map-master/grid_map_core):
//   GridMap ctor/setGeometry/operator[]   gmc/src/GridMap.cpp:27-70,125-151
//   MapProvider::composeMasterMapFromLayerdMap  mc/src/map_provider.cpp:216-223
//   GridMap::move                          gmc/src/GridMap.cpp:346-412
#include "engine.hpp"
#include "compose_dev.hpp"

#include <cmath>
#include <dirent.h>
#include <unistd.h>
#include <cstdlib>
#include <cstring>
#include <limits>

using namespace rna;

// ------------------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------------------
__global__ void fill_f32_kernel(float* __restrict__ p, size_t n, float v) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) p[i] = v;
}

// master = laser for the tiles whose dirty bit is set (mode 0).  One 256-thread block per tile,
// lanes run along Index(0), the contiguous dimension of the column-major layer.
__global__ void compose_dirty_tiles_kernel(float* __restrict__ master, const float* __restrict__ laser,
                                           const unsigned* __restrict__ dirty, int rows, int cols, int tiles_i) {
  const int ti = blockIdx.x, tj = blockIdx.y;
  const int t = tj * tiles_i + ti;
  if (!reinterpret_cast<const unsigned char*>(dirty)[t]) return;
  const int i0 = ti * TILE, j0 = tj * TILE;
  const int li = threadIdx.x & (TILE - 1);
  for (int c = threadIdx.x >> 6; c < TILE; c += 4) {
    const int i = i0 + li, j = j0 + c;
    if (i < rows && j < cols) {
      const size_t lin = (size_t)j * rows + i;
      master[lin] = laser[lin];
    }
  }
}

// (cell_blocked -- GlobalPlanner::ifBlocked's predicate -- lives in compose_dev.hpp)

// Per-cell 8-bit traversable-neighbour mask (DESIGN.md "Grid A* contract"): bit k <-> neighbour
// (di,dj) in the order (-1,-1),(0,-1),(1,-1),(-1,0),(1,0),(-1,1),(0,1),(1,1); a diagonal needs the
// target and both orthogonal cells free.  One block per 64x64 tile with a 1-cell halo in LDS.
// `all` != 0 recomputes every tile, otherwise only tiles that are dirty or touch a dirty tile.
// Tiles and adjacency are taken in MAP space (unwrapped indices): on a moved map the neighbours of a
// cell wrap around the circular buffer but not around the map edge; (s0, s1) is the buffer start
// index and the mask is stored at the cell's buffer index.  The dirty-tile shortcut is in buffer
// space, so the host passes all != 0 whenever the start index is non-zero.
__global__ void nbr_mask_tiles_kernel(uint8_t* __restrict__ nbr, const float* __restrict__ master,
                                      const unsigned* __restrict__ dirty, int all, int rows, int cols,
                                      int tiles_i, int tiles_j, int s0, int s1) {
  const int ti = blockIdx.x, tj = blockIdx.y;
  if (!all) {
    bool need = false;
    for (int dj = -1; dj <= 1 && !need; ++dj)
      for (int di = -1; di <= 1; ++di) {
        const int a = ti + di, b = tj + dj;
        if (a < 0 || b < 0 || a >= tiles_i || b >= tiles_j) continue;
        const int t = b * tiles_i + a;
        if (reinterpret_cast<const unsigned char*>(dirty)[t]) { need = true; break; }
      }
    if (!need) return;
  }
  __shared__ uint8_t blk[(TILE + 2) * (TILE + 2)];  // [jj][ii], ii fastest; out of map = blocked
  const int i0 = ti * TILE - 1, j0 = tj * TILE - 1;
  for (int k = threadIdx.x; k < (TILE + 2) * (TILE + 2); k += blockDim.x) {
    const int ii = k % (TILE + 2), jj = k / (TILE + 2);
    const int i = i0 + ii, j = j0 + jj;
    uint8_t b = 1;
    if (i >= 0 && j >= 0 && i < rows && j < cols) {
      int bi = i + s0, bj = j + s1;
      if (bi >= rows) bi -= rows;
      if (bj >= cols) bj -= cols;
      b = cell_blocked(master[(size_t)bj * rows + bi]) ? 1 : 0;
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
    int bi = i + s0, bj = j + s1;
    if (bi >= rows) bi -= rows;
    if (bj >= cols) bj -= cols;
    nbr[(size_t)bj * rows + bi] = (uint8_t)m;
  }
}

// MapProvider::updateMap's compose step and the mask refresh in ONE launch (unmoved map, masks valid before the update):
// the block of a dirty tile copies laser -> master (composeMasterMapFromLayerdMap, mc/src/map_provider.cpp:216-223,
// restricted to where the two layers differ), and every block whose tile is dirty or touches a dirty tile recomputes
// its cells' masks.  A cell of a DIRTY tile is read from the laser layer -- what master holds there once this launch
// has finished; its own block may still be copying --, a cell of a clean tile from master.  The block also clears its
// byte of `next_dirty`, the flag array the next update will mark (the two arrays swap roles on the host: the flags
// this launch consumed stay readable as "last dirty" until the next compose, rna_last_dirty_tiles).
__global__ void __launch_bounds__(256) compose_nbr_tiles_kernel(uint8_t* __restrict__ nbr, float* __restrict__ master, const float* __restrict__ laser,
                                                                const unsigned* __restrict__ dirty, unsigned* __restrict__ next_dirty, int rows,
                                                                int cols, int tiles_i, int tiles_j) {
  // A workgroup serves tiles in turn (grid-stride; at most 16 workgroups per CU are launched: one tile each on a 4096^2
  // map.  Fewer, longer workgroups -- 4 per CU, four tiles each -- are faster alone, 10 -> 8 us, and slower next to the
  // searches, 0.27 -> 0.38 ms per pass: there a kernel's time is its workgroups waiting for wave slots, and many short
  // ones find them sooner.  RNA_COMPOSE_WGS_PER_CU = 2 / 4 / 8 / 12 / 16 in the loop: 0.44 / 0.37 / 0.30 / 0.29 / 0.29 ms,
  // profiles/r04_sweep_compose_wgs_per_cu.txt).
  __shared__ uint8_t blk[(TILE + 2) * (TILE + 2)];  // [jj][ii], ii fastest; out of map = blocked
  for (int tt = blockIdx.x; tt < tiles_i * tiles_j; tt += gridDim.x)
    compose_nbr_tile(blk, tt, nbr, master, laser, dirty, next_dirty, rows, cols, tiles_i, tiles_j);
}

// GridMap::clearRows / clearCols on every layer (gmc/src/GridMap.cpp:590-606)
__global__ void clear_region_kernel(float* __restrict__ p, int rows, int i0, int ni, int j0, int nj) {
  const size_t n = (size_t)ni * nj;
  size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const float nanv = __int_as_float(0x7fc00000);
  for (; k < n; k += stride) {
    const int i = i0 + (int)(k % ni), j = j0 + (int)(k / ni);
    p[(size_t)j * rows + i] = nanv;
  }
}

// Tiled single map, incremental hand-over: the listed 64 x 64 tiles, clipped to a window, <-> fixed 4096-float
// slots of a dense buffer (slot k = tile k of the list; cells outside the clip are skipped / left alone).
__global__ void __launch_bounds__(256) pack_tiles_kernel(const float* __restrict__ layer, int rows, int tiles_i,
                                                         const int32_t* __restrict__ tiles, int wi0, int wi1, int wj0,
                                                         int wj1, float* __restrict__ dense) {
  const int t = tiles[blockIdx.x];
  const int i0 = (t % tiles_i) * TILE, j0 = (t / tiles_i) * TILE;
  float* out = dense + (size_t)blockIdx.x * (TILE * TILE);
  for (int k = threadIdx.x; k < TILE * TILE; k += 256) {
    const int i = i0 + (k & (TILE - 1)), j = j0 + (k >> 6);
    out[k] = (i >= wi0 && i < wi1 && j >= wj0 && j < wj1) ? layer[(size_t)j * rows + i] : 0.0f;
  }
}

__global__ void __launch_bounds__(256) unpack_tiles_kernel(float* __restrict__ layer_a, float* __restrict__ layer_b,
                                                           int rows, int tiles_i, const int32_t* __restrict__ tiles,
                                                           int wi0, int wi1, int wj0, int wj1,
                                                           const float* __restrict__ dense,
                                                           unsigned* __restrict__ dirty_tiles) {
  const int t = tiles[blockIdx.x];
  const int i0 = (t % tiles_i) * TILE, j0 = (t / tiles_i) * TILE;
  const float* in = dense + (size_t)blockIdx.x * (TILE * TILE);
  for (int k = threadIdx.x; k < TILE * TILE; k += 256) {
    const int i = i0 + (k & (TILE - 1)), j = j0 + (k >> 6);
    if (i >= wi0 && i < wi1 && j >= wj0 && j < wj1) {
      const float v = in[k];
      layer_a[(size_t)j * rows + i] = v;
      if (layer_b) layer_b[(size_t)j * rows + i] = v;
    }
  }
  if (threadIdx.x == 0) reinterpret_cast<volatile unsigned char*>(dirty_tiles)[t] = 1;
}

// the flagged tiles that intersect a window, as a device list (order arbitrary) + count
__global__ void dirty_list_kernel(const unsigned* __restrict__ flags, int tiles_i, int ntile, int wi0, int wi1, int wj0, int wj1,
                                  int32_t* __restrict__ list, int* __restrict__ count) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= ntile || !reinterpret_cast<const unsigned char*>(flags)[t]) return;
  const int i0 = (t % tiles_i) * TILE, j0 = (t / tiles_i) * TILE;
  if (i0 < wi1 && i0 + TILE > wi0 && j0 < wj1 && j0 + TILE > wj0) list[atomicAdd(count, 1)] = t;
}

__global__ void mark_tiles_kernel(unsigned* __restrict__ dirty_tiles, int tiles_i, int ta0, int ta1, int tb0, int tb1) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  const int na = ta1 - ta0, n = na * (tb1 - tb0);
  if (k < n) reinterpret_cast<volatile unsigned char*>(dirty_tiles)[(tb0 + k / na) * tiles_i + ta0 + k % na] = 1;
}

// GridMap::getSubmap (gmc/src/GridMap.cpp:287-339): submap cell (r, c) is parent buffer cell
// ((tl0 + r) mod rows, (tl1 + c) mod cols) -- the reference's <= 4 quadrant blocks are this wrap.
__global__ void submap_gather_kernel(const float* __restrict__ layer, int rows, int cols, int tl0, int tl1, int sr,
                                     int sc, float* __restrict__ out) {
  const size_t n = (size_t)sr * sc;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += stride) {
    int bi = tl0 + (int)(k % sr), bj = tl1 + (int)(k / sr);
    if (bi >= rows) bi -= rows;
    if (bj >= cols) bj -= cols;
    out[k] = layer[(size_t)bj * rows + bi];
  }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
static int grid_for(size_t n, int block) {
  size_t b = (n + block - 1) / block;
  if (b > 8192) b = 8192;
  if (b == 0) b = 1;
  return (int)b;
}

static int fill_layer(rna_engine* e, float* p, float v) {
  hipLaunchKernelGGL(fill_f32_kernel, dim3(grid_for(e->ncell, 256)), dim3(256), 0, e->stream, p, e->ncell, v);
  RNA_HIP(e, hipGetLastError());
  return RNA_OK;
}

extern "C" int rna_abi_version(void) { return RNA_ABI_VERSION; }

extern "C" const char* rna_last_error(const rna_engine* e) { return e ? e->err.c_str() : "null engine"; }

extern "C" const char* rna_kernel_name(int id) {
  static const char* names[RNA_K_COUNT] = {"himm_prep", "himm_raster", "himm_apply", "compose_master", "nbr_mask",
                                           "vfh_step", "astar_search", "astar_init", "rrt", "to_occupancy_grid", "astar_reset"};
  return (id >= 0 && id < RNA_K_COUNT) ? names[id] : "?";
}

extern "C" int rna_create(rna_engine** out, double length_x, double length_y, double resolution, double px,
                          double py, int device_id) {
  if (!out) return RNA_EINVAL;
  *out = nullptr;
  if (!(length_x > 0.0) || !(length_y > 0.0) || !(resolution > 0.0)) return RNA_EINVAL;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return RNA_ENODEVICE;
  if (device_id < 0 || device_id >= ndev) return RNA_EINVAL;
  rna_engine* e = new rna_engine();
  e->device = device_id;
  {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device_id) == hipSuccess && cus > 0) e->cu_count = cus;
  }
  set_geometry(e->geom, length_x, length_y, resolution, px, py);
  if (e->geom.size[0] <= 0 || e->geom.size[1] <= 0 ||
      (double)e->geom.size[0] * (double)e->geom.size[1] > 2.0e9) {
    delete e;
    return RNA_EINVAL;
  }
  e->ncell = (size_t)e->geom.size[0] * (size_t)e->geom.size[1];
  e->tiles_i = (e->geom.size[0] + TILE - 1) / TILE;
  e->tiles_j = (e->geom.size[1] + TILE - 1) / TILE;
  int rc = RNA_OK;
  auto bail = [&](int code) { rna_destroy(e); return code; };
  if (hipSetDevice(device_id) != hipSuccess) return bail(RNA_EHIP);
  {
    int prio_lo = 0, prio_hi = 0;   // numerically lower = higher priority
    (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
    if (getenv("RNA_NO_STREAM_PRIORITY")) prio_lo = prio_hi = 0;
    // developer knob RNA_ENGINE_CU_MASK=n: the engine stream may only use the first n CUs of the mask order (the CUs
    // the pipelined search streams leave free, astar.hip) -- its short kernels then never land next to search
    // workgroups on a CU whose issue slots they fill
    int only = 0;
    if (const char* m = getenv("RNA_ENGINE_CU_MASK")) only = atoi(m);
    hipDeviceProp_t prop;
    if (only > 0 && hipGetDeviceProperties(&prop, device_id) == hipSuccess && only < prop.multiProcessorCount) {
      uint32_t mask[16] = {};
      for (int c = 0; c < only && c < 512; ++c) mask[c >> 5] |= 1u << (c & 31);
      const int words = (prop.multiProcessorCount + 31) / 32 < 16 ? (prop.multiProcessorCount + 31) / 32 : 16;
      if (hipExtStreamCreateWithCUMask(&e->stream, (uint32_t)words, mask) != hipSuccess) return bail(RNA_EHIP);
    } else if (hipStreamCreateWithPriority(&e->stream, hipStreamNonBlocking, prio_hi) != hipSuccess) return bail(RNA_EHIP);
  }
  for (int l = 0; l < RNA_NUM_LAYERS; ++l) {
    if ((rc = dev_alloc(e, &e->layer[l], e->ncell)) != RNA_OK) return bail(rc);
    // GridMap::setGeometry -> clearAll(): every layer starts as NaN (gmc/src/GridMap.cpp:62)
    if ((rc = fill_layer(e, e->layer[l], std::numeric_limits<float>::quiet_NaN())) != RNA_OK) return bail(rc);
  }
  const size_t words = ((size_t)e->tiles_i * e->tiles_j + 3) / 4;  // one byte per tile, rounded to words
  if ((rc = dev_alloc(e, &e->dirty_tiles, words)) != RNA_OK) return bail(rc);
  if (hipMemsetAsync(e->dirty_tiles, 0, words * sizeof(unsigned), e->stream) != hipSuccess) return bail(RNA_EHIP);
  if ((rc = dev_alloc(e, &e->last_dirty, words)) != RNA_OK) return bail(rc);
  if (hipMemsetAsync(e->last_dirty, 0, words * sizeof(unsigned), e->stream) != hipSuccess) return bail(RNA_EHIP);
  if ((rc = dev_alloc(e, &e->nbr, e->ncell)) != RNA_OK) return bail(rc);
  e->nbr_all_dirty = true;
  if (hipStreamSynchronize(e->stream) != hipSuccess) return bail(RNA_EHIP);
  *out = e;
  return RNA_OK;
}

extern "C" void rna_destroy(rna_engine* e) {
  if (!e) return;
  (void)hipSetDevice(e->device);
  if (e->stream) (void)sync_all(e);
  himm_release(e);
  vfh_release(e);
  astar_release(e);
  for (int l = 0; l < RNA_NUM_LAYERS; ++l) dev_free(&e->layer[l]);
  dev_free(&e->dirty_tiles);
  dev_free(&e->last_dirty);
  dev_free(&e->tile_list);
  dev_free(&e->nbr);
  (void)profile_flush(e);
  for (hipEvent_t ev : e->free_events) (void)hipEventDestroy(ev);
  e->free_events.clear();
  if (e->vfh_stream) (void)hipStreamDestroy(e->vfh_stream);
  if (e->ev_vfh_go) (void)hipEventDestroy(e->ev_vfh_go);
  if (e->ev_vfh_done) (void)hipEventDestroy(e->ev_vfh_done);
  if (e->stream) (void)hipStreamDestroy(e->stream);
  delete e;
}

extern "C" int rna_get_geometry(const rna_engine* e, rna_geometry* o) {
  if (!e || !o) return RNA_EINVAL;
  for (int a = 0; a < 2; ++a) {
    o->length[a] = e->geom.len[a];
    o->position[a] = e->geom.pos[a];
    o->size[a] = e->geom.size[a];
    o->start_index[a] = e->geom.start[a];
  }
  o->resolution = e->geom.res;
  return RNA_OK;
}

static void layer_changed(rna_engine* e, int layer) {
  // master written directly (upload, fill, fromOccupancyGrid, a caller's device pointer): it no longer equals laser
  // outside the dirty tiles, so the next compose has to be the reference's whole-layer copy (map_provider.cpp:221)
  if (layer == RNA_LAYER_MASTER) { e->nbr_all_dirty = true; e->master_diverged = true; }
  if (layer == RNA_LAYER_LASER) e->laser_all_dirty = true;
}

extern "C" int rna_layer_upload(rna_engine* e, int layer, const float* host, size_t n) {
  if (!e || !host || layer < 0 || layer >= RNA_NUM_LAYERS || n != e->ncell) return RNA_EINVAL;
  RNA_ENTER(e);
  RNA_HIP(e, hipMemcpyAsync(e->layer[layer], host, n * sizeof(float), hipMemcpyHostToDevice, e->stream));
  RNA_HIP(e, hipStreamSynchronize(e->stream));
  layer_changed(e, layer);
  return RNA_OK;
}

extern "C" int rna_layer_download(rna_engine* e, int layer, float* host, size_t n) {
  if (!e || !host || layer < 0 || layer >= RNA_NUM_LAYERS || n != e->ncell) return RNA_EINVAL;
  RNA_ENTER(e);
  RNA_HIP(e, hipMemcpyAsync(host, e->layer[layer], n * sizeof(float), hipMemcpyDeviceToHost, e->stream));
  RNA_HIP(e, hipStreamSynchronize(e->stream));
  return RNA_OK;
}

extern "C" int rna_layer_fill(rna_engine* e, int layer, float value) {
  if (!e || layer < 0 || layer >= RNA_NUM_LAYERS) return RNA_EINVAL;
  RNA_ENTER(e);
  int rc = fill_layer(e, e->layer[layer], value);
  layer_changed(e, layer);
  return rc;
}

extern "C" void* rna_layer_device_ptr(rna_engine* e, int layer) {
  if (!e || layer < 0 || layer >= RNA_NUM_LAYERS) return nullptr;
  layer_changed(e, layer);  // the caller may write through the pointer
  return e->layer[layer];
}

// Rectangular block of a layer <-> a dense column-major device buffer (ni fastest): the halo
// strips and owner tiles of the tiled single-map mode travel through these (ros_navigation_amd/dist.py).
static int region_copy(rna_engine* e, int layer, int i0, int ni, int j0, int nj, float* dense, bool pack,
                       bool tracked = false) {
  if (!e || !dense || layer < 0 || layer >= RNA_NUM_LAYERS) return RNA_EINVAL;
  if (ni <= 0 || nj <= 0) return RNA_OK;
  if (i0 < 0 || j0 < 0 || i0 + ni > e->geom.size[0] || j0 + nj > e->geom.size[1])
    return rna::fail(e, RNA_EINVAL, "region outside the map");
  RNA_ENTER(e);
  float* blk = e->layer[layer] + (size_t)j0 * e->geom.size[0] + i0;
  const size_t lpitch = (size_t)e->geom.size[0] * sizeof(float), dpitch = (size_t)ni * sizeof(float);
  if (pack) {
    RNA_HIP(e, hipMemcpy2DAsync(dense, dpitch, blk, lpitch, dpitch, (size_t)nj, hipMemcpyDeviceToDevice, e->stream));
  } else {
    RNA_HIP(e, hipMemcpy2DAsync(blk, lpitch, dense, dpitch, dpitch, (size_t)nj, hipMemcpyDeviceToDevice, e->stream));
    if (tracked) {   // per-tile bookkeeping instead of "everything changed": the next compose refreshes these tiles
      const int ta0 = i0 / TILE, ta1 = (i0 + ni - 1) / TILE + 1, tb0 = j0 / TILE, tb1 = (j0 + nj - 1) / TILE + 1;
      const int nt = (ta1 - ta0) * (tb1 - tb0);
      hipLaunchKernelGGL(mark_tiles_kernel, dim3((nt + 255) / 256), dim3(256), 0, e->stream, e->dirty_tiles, e->tiles_i, ta0,
                         ta1, tb0, tb1);
      RNA_HIP(e, hipGetLastError());
    } else {
      layer_changed(e, layer);
    }
  }
  RNA_HIP(e, hipStreamSynchronize(e->stream));
  return RNA_OK;
}

extern "C" int rna_layer_pack_region(rna_engine* e, int layer, int i0, int ni, int j0, int nj, float* dense_device) {
  return region_copy(e, layer, i0, ni, j0, nj, dense_device, true);
}

extern "C" int rna_layer_unpack_region(rna_engine* e, int layer, int i0, int ni, int j0, int nj,
                                       const float* dense_device) {
  return region_copy(e, layer, i0, ni, j0, nj, const_cast<float*>(dense_device), false);
}

extern "C" int rna_layer_unpack_region_tracked(rna_engine* e, int layer, int i0, int ni, int j0, int nj,
                                               const float* dense_device) {
  return region_copy(e, layer, i0, ni, j0, nj, const_cast<float*>(dense_device), false, true);
}

static int get_submap(rna_engine* e, int layer, double px, double py, double lx, double ly, float* out, size_t cap,
                      rna_submap_info* info, bool to_host) {
  if (!e || !info || layer < 0 || layer >= RNA_NUM_LAYERS || (cap > 0 && !out)) return RNA_EINVAL;
  const double rp[2] = {px, py}, rl[2] = {lx, ly};
  SubmapInfo si;
  if (!submap_information(e->geom, rp, rl, si)) return 0;
  for (int a = 0; a < 2; ++a) {
    info->length[a] = si.len[a]; info->position[a] = si.pos[a];
    info->size[a] = si.size[a]; info->top_left[a] = si.top_left[a];
  }
  if (si.size[0] <= 0 || si.size[1] <= 0 || si.size[0] > e->geom.size[0] || si.size[1] > e->geom.size[1]) return 0;
  const size_t n = (size_t)si.size[0] * si.size[1];
  if (n > cap) return rna::fail(e, RNA_ECAPACITY, "rna_get_submap: output buffer smaller than the submap");
  RNA_ENTER(e);
  float* dst = out;
  float* staging = nullptr;
  if (to_host) {
    RNA_HIP(e, hipMalloc(&staging, n * sizeof(float)));
    dst = staging;
  }
  hipLaunchKernelGGL(submap_gather_kernel, dim3(grid_for(n, 256)), dim3(256), 0, e->stream, e->layer[layer],
                     e->geom.size[0], e->geom.size[1], si.top_left[0], si.top_left[1], si.size[0], si.size[1], dst);
  hipError_t err = hipGetLastError();
  if (err == hipSuccess && to_host)
    err = hipMemcpyAsync(out, staging, n * sizeof(float), hipMemcpyDeviceToHost, e->stream);
  if (err == hipSuccess && to_host) err = hipStreamSynchronize(e->stream);
  if (staging) (void)hipFree(staging);
  RNA_HIP(e, err);
  return 1;
}

extern "C" int rna_last_dirty_tiles(rna_engine* e, uint8_t* flags_host, size_t n_tiles) {
  if (!e || !flags_host || n_tiles != (size_t)e->tiles_i * e->tiles_j) return RNA_EINVAL;
  RNA_ENTER(e);
  RNA_HIP(e, hipMemcpyAsync(flags_host, e->last_dirty, n_tiles, hipMemcpyDeviceToHost, e->stream));
  RNA_HIP(e, hipStreamSynchronize(e->stream));
  return RNA_OK;
}

static int stage_tile_list(rna_engine* e, const int32_t* tiles_host, int n) {
  const int ntile = e->tiles_i * e->tiles_j;
  for (int k = 0; k < n; ++k)
    if (tiles_host[k] < 0 || tiles_host[k] >= ntile) return rna::fail(e, RNA_EINVAL, "tile index outside the map");
  if (n > e->tile_list_cap) {
    int cap = 256;
    while (cap < n) cap <<= 1;
    int rc = dev_alloc(e, &e->tile_list, (size_t)cap);
    if (rc != RNA_OK) { e->tile_list_cap = 0; return rc; }
    e->tile_list_cap = cap;
  }
  RNA_HIP(e, hipMemcpyAsync(e->tile_list, tiles_host, (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, e->stream));
  return RNA_OK;
}

static bool window_ok(const rna_engine* e, int i0, int ni, int j0, int nj) {
  return i0 >= 0 && j0 >= 0 && ni > 0 && nj > 0 && i0 + ni <= e->geom.size[0] && j0 + nj <= e->geom.size[1];
}

extern "C" int rna_layer_pack_tiles(rna_engine* e, int layer, const int32_t* tiles_host, int n, int i0, int ni, int j0,
                                    int nj, float* dense_device) {
  if (!e || layer < 0 || layer >= RNA_NUM_LAYERS || n < 0 || (n > 0 && (!tiles_host || !dense_device))) return RNA_EINVAL;
  if (n == 0) return RNA_OK;
  if (!window_ok(e, i0, ni, j0, nj)) return rna::fail(e, RNA_EINVAL, "window outside the map");
  RNA_ENTER(e);
  int rc = stage_tile_list(e, tiles_host, n);
  if (rc != RNA_OK) return rc;
  hipLaunchKernelGGL(pack_tiles_kernel, dim3(n), dim3(256), 0, e->stream, e->layer[layer], e->geom.size[0], e->tiles_i,
                     e->tile_list, i0, i0 + ni, j0, j0 + nj, dense_device);
  RNA_HIP(e, hipGetLastError());
  RNA_HIP(e, hipStreamSynchronize(e->stream));
  return RNA_OK;
}

extern "C" int rna_layers_unpack_tiles(rna_engine* e, int layer_a, int layer_b, const int32_t* tiles_host, int n, int i0,
                                       int ni, int j0, int nj, const float* dense_device) {
  if (!e || layer_a < 0 || layer_a >= RNA_NUM_LAYERS || layer_b >= RNA_NUM_LAYERS || layer_b == layer_a || n < 0 ||
      (n > 0 && (!tiles_host || !dense_device)))
    return RNA_EINVAL;
  if (n == 0) return RNA_OK;
  if (!window_ok(e, i0, ni, j0, nj)) return rna::fail(e, RNA_EINVAL, "window outside the map");
  RNA_ENTER(e);
  int rc = stage_tile_list(e, tiles_host, n);
  if (rc != RNA_OK) return rc;
  hipLaunchKernelGGL(unpack_tiles_kernel, dim3(n), dim3(256), 0, e->stream, e->layer[layer_a],
                     layer_b >= 0 ? e->layer[layer_b] : (float*)nullptr, e->geom.size[0], e->tiles_i, e->tile_list, i0,
                     i0 + ni, j0, j0 + nj, dense_device, e->dirty_tiles);
  RNA_HIP(e, hipGetLastError());
  RNA_HIP(e, hipStreamSynchronize(e->stream));
  return RNA_OK;
}

// Device-side forms of the three calls above for hosts that drive the exchange themselves (csrc/rccl_tiled.hip): no
// tile flags or lists cross PCIe.  All asynchronous on the engine's stream.
extern "C" int rna_last_dirty_tiles_device(rna_engine* e, int i0, int ni, int j0, int nj, int32_t* list_device, int* count_device) {
  if (!e || !list_device || !count_device) return RNA_EINVAL;
  if (!window_ok(e, i0, ni, j0, nj)) return rna::fail(e, RNA_EINVAL, "window outside the map");
  RNA_ENTER(e);
  const int ntile = e->tiles_i * e->tiles_j;
  RNA_HIP(e, hipMemsetAsync(count_device, 0, sizeof(int), e->stream));
  hipLaunchKernelGGL(dirty_list_kernel, dim3((ntile + 255) / 256), dim3(256), 0, e->stream, e->last_dirty, e->tiles_i, ntile, i0,
                     i0 + ni, j0, j0 + nj, list_device, count_device);
  RNA_HIP(e, hipGetLastError());
  return RNA_OK;
}

extern "C" int rna_layer_pack_tiles_device(rna_engine* e, int layer, const int32_t* tiles_device, int n, int i0, int ni, int j0,
                                           int nj, float* dense_device) {
  if (!e || layer < 0 || layer >= RNA_NUM_LAYERS || n < 0 || (n > 0 && (!tiles_device || !dense_device))) return RNA_EINVAL;
  if (n == 0) return RNA_OK;
  if (!window_ok(e, i0, ni, j0, nj)) return rna::fail(e, RNA_EINVAL, "window outside the map");
  RNA_ENTER(e);
  hipLaunchKernelGGL(pack_tiles_kernel, dim3(n), dim3(256), 0, e->stream, e->layer[layer], e->geom.size[0], e->tiles_i,
                     tiles_device, i0, i0 + ni, j0, j0 + nj, dense_device);
  RNA_HIP(e, hipGetLastError());
  return RNA_OK;
}

extern "C" int rna_layers_unpack_tiles_device(rna_engine* e, int layer_a, int layer_b, const int32_t* tiles_device, int n, int i0,
                                              int ni, int j0, int nj, const float* dense_device) {
  if (!e || layer_a < 0 || layer_a >= RNA_NUM_LAYERS || layer_b >= RNA_NUM_LAYERS || layer_b == layer_a || n < 0 ||
      (n > 0 && (!tiles_device || !dense_device)))
    return RNA_EINVAL;
  if (n == 0) return RNA_OK;
  if (!window_ok(e, i0, ni, j0, nj)) return rna::fail(e, RNA_EINVAL, "window outside the map");
  RNA_ENTER(e);
  hipLaunchKernelGGL(unpack_tiles_kernel, dim3(n), dim3(256), 0, e->stream, e->layer[layer_a],
                     layer_b >= 0 ? e->layer[layer_b] : (float*)nullptr, e->geom.size[0], e->tiles_i, tiles_device, i0,
                     i0 + ni, j0, j0 + nj, dense_device, e->dirty_tiles);
  RNA_HIP(e, hipGetLastError());
  return RNA_OK;
}

// GridMap GridMap::getSubmap(position, length, isSuccess) (gmc/src/GridMap.cpp:287-339) as a GridMap of its own: a new
// engine with the submap's geometry (startIndex (0,0)) whose layers are gathered from the parent on the device.
extern "C" int rna_create_submap(rna_engine* parent, double px, double py, double lx, double ly, rna_engine** out) {
  if (!parent || !out) return RNA_EINVAL;
  *out = nullptr;
  const double rp[2] = {px, py}, rl[2] = {lx, ly};
  SubmapInfo si;
  if (!submap_information(parent->geom, rp, rl, si)) return 0;
  if (si.size[0] <= 0 || si.size[1] <= 0 || si.size[0] > parent->geom.size[0] || si.size[1] > parent->geom.size[1]) return 0;
  rna_engine* c = nullptr;
  int rc = rna_create(&c, si.len[0], si.len[1], parent->geom.res, si.pos[0], si.pos[1], parent->device);
  if (rc != RNA_OK) return rna::fail(parent, rc, "rna_create_submap: rna_create failed");
  if (c->geom.size[0] != si.size[0] || c->geom.size[1] != si.size[1]) {
    rna_destroy(c);
    return rna::fail(parent, RNA_EINVAL, "rna_create_submap: geometry does not reproduce the submap size");
  }
  hipError_t err = hipSetDevice(parent->device);
  for (int l = 0; l < RNA_NUM_LAYERS && err == hipSuccess; ++l) {
    hipLaunchKernelGGL(submap_gather_kernel, dim3(grid_for(c->ncell, 256)), dim3(256), 0, parent->stream, parent->layer[l],
                       parent->geom.size[0], parent->geom.size[1], si.top_left[0], si.top_left[1], si.size[0], si.size[1],
                       c->layer[l]);
    err = hipGetLastError();
  }
  if (err == hipSuccess) err = hipStreamSynchronize(parent->stream);
  if (err != hipSuccess) {
    rna_destroy(c);
    RNA_HIP(parent, err);
  }
  c->nbr_all_dirty = true;
  c->laser_all_dirty = true;
  *out = c;
  return 1;
}

// GridMap copy (`map = map_` of MapProvider::getMap, mc/src/map_provider.cpp:120-125): a new engine on the same
// device with the same geometry, circular-buffer start index and layer contents.
extern "C" int rna_clone(rna_engine* src, rna_engine** out) {
  if (!src || !out) return RNA_EINVAL;
  *out = nullptr;
  rna_engine* c = nullptr;
  const Geom& g = src->geom;
  int rc = rna_create(&c, g.len[0], g.len[1], g.res, g.pos[0], g.pos[1], src->device);
  if (rc != RNA_OK) return rna::fail(src, rc, "rna_clone: rna_create failed");
  if (c->geom.size[0] != g.size[0] || c->geom.size[1] != g.size[1]) {
    rna_destroy(c);
    return rna::fail(src, RNA_EINVAL, "rna_clone: geometry does not reproduce the size");
  }
  c->geom = g;   // position / length / start index exactly as they are
  hipError_t err = hipSetDevice(src->device);
  for (int l = 0; l < RNA_NUM_LAYERS && err == hipSuccess; ++l)
    err = hipMemcpyAsync(c->layer[l], src->layer[l], src->ncell * sizeof(float), hipMemcpyDeviceToDevice, src->stream);
  if (err == hipSuccess) err = hipStreamSynchronize(src->stream);
  if (err != hipSuccess) {
    rna_destroy(c);
    RNA_HIP(src, err);
  }
  c->nbr_all_dirty = true;
  c->laser_all_dirty = true;
  *out = c;
  return RNA_OK;
}

extern "C" int rna_get_submap(rna_engine* e, int layer, double px, double py, double lx, double ly, float* out_host,
                              size_t cap_cells, rna_submap_info* info) {
  return get_submap(e, layer, px, py, lx, ly, out_host, cap_cells, info, true);
}

extern "C" int rna_get_submap_device(rna_engine* e, int layer, double px, double py, double lx, double ly,
                                     float* out_device, size_t cap_cells, rna_submap_info* info) {
  return get_submap(e, layer, px, py, lx, ly, out_device, cap_cells, info, false);
}

extern "C" void* rna_stream(rna_engine* e) { return e ? (void*)e->stream : nullptr; }

extern "C" int rna_synchronize(rna_engine* e) {
  if (!e) return RNA_EINVAL;
  RNA_ENTER(e);
  return sync_all(e);
}

extern "C" int rna_synchronize_map(rna_engine* e) {
  if (!e) return RNA_EINVAL;
  RNA_ENTER_NOJOIN(e);
  RNA_HIP(e, hipStreamSynchronize(e->stream));
  if (e->vfh_stream) RNA_HIP(e, hipStreamSynchronize(e->vfh_stream));
  return RNA_OK;
}

// GPU_MAX_HW_QUEUES is read by the HIP runtime at its first call: set it when the library is loaded (a host links or
// dlopens librna.so before it touches the GPU) unless the host has chosen a value itself.  Priority 101: before the
// constructors that register this library's code objects with the runtime.  What the constructor found is kept for
// rna_hw_queue_advice: the value as it was then (not as the environment reads later), whether this library had to set it,
// and whether the process had ALREADY opened the GPU by then (/dev/kfd among its file descriptors: the runtime has latched
// its settings and a value set now comes too late -- a host that initialises HIP, e.g. through torch.cuda, before it
// loads librna.so must export the variable itself).  (setenv from a constructor is not thread-safe against other threads'
// getenv: a host that dlopens the library from a threaded process should set the variable itself, then nothing is written.)
static int g_hwq_at_load = 0;            // GPU_MAX_HW_QUEUES after the constructor ran (0: unset)
static bool g_hwq_set_by_us = false, g_gpu_open_at_load = false;
static bool process_has_kfd_open() {
  // (the descriptors the process HAS, whatever their numbers: round 5 probed 0..4095 with 4 096 readlink calls per dlopen)
  DIR* d = opendir("/proc/self/fd");
  if (!d) return false;
  bool found = false;
  char link[300], target[256];
  while (const dirent* ent = readdir(d)) {
    if (ent->d_name[0] == '.') continue;
    snprintf(link, sizeof(link), "/proc/self/fd/%s", ent->d_name);
    const ssize_t n = readlink(link, target, sizeof(target) - 1);
    if (n <= 0) continue;
    target[n] = 0;
    if (strcmp(target, "/dev/kfd") == 0) { found = true; break; }
  }
  closedir(d);
  return found;
}
__attribute__((constructor(101))) static void rna_default_hw_queues() {
  g_gpu_open_at_load = process_has_kfd_open();
  if (!getenv("RNA_KEEP_HW_QUEUES") && !getenv("GPU_MAX_HW_QUEUES")) {
    setenv("GPU_MAX_HW_QUEUES", "8", 0);
    g_hwq_set_by_us = true;
  }
  const char* v = getenv("GPU_MAX_HW_QUEUES");
  g_hwq_at_load = v ? atoi(v) : 0;
}

extern "C" int rna_hw_queue_advice(int pipeline_depth, char* buf, size_t cap) {
  if (buf && cap) buf[0] = 0;
  if (pipeline_depth <= 2) return 0;
  const int have = g_hwq_at_load > 0 ? g_hwq_at_load : 4;   // (unset: the runtime's default)
  if (have < 8) {
    if (buf && cap)
      snprintf(buf, cap, "A* pipeline depth %d needs GPU_MAX_HW_QUEUES >= 8 in the environment before the first HIP call (%s%d): "
               "the stages' streams share %d hardware queues and their searches overlap less", pipeline_depth,
               g_hwq_at_load > 0 ? "it was " : "unset, the runtime's default is ", have, have);
    return 1;
  }
  if (g_hwq_set_by_us && g_gpu_open_at_load) {
    if (buf && cap)
      snprintf(buf, cap, "librna.so set GPU_MAX_HW_QUEUES=8 when it was loaded, but this process had opened the GPU before that: the HIP "
               "runtime has probably latched its default of 4 hardware queues, and a pipeline of %d A* stages shares them.  Export "
               "GPU_MAX_HW_QUEUES=8 before the first HIP call (e.g. before importing torch)", pipeline_depth);
    return 2;
  }
  return 0;
}

extern "C" int rna_get_index(const rna_engine* e, double x, double y, int32_t index[2]) {
  if (!e || !index) return RNA_EINVAL;
  int idx[2];
  if (!index_from_position(e->geom, x, y, idx)) return 0;
  index[0] = idx[0];
  index[1] = idx[1];
  return 1;
}

extern "C" int rna_get_position(const rna_engine* e, int32_t i, int32_t j, double p[2]) {
  if (!e || !p) return RNA_EINVAL;
  if (i < 0 || j < 0 || i >= e->geom.size[0] || j >= e->geom.size[1]) return 0;
  const int idx[2] = {i, j};
  position_from_index(e->geom, idx, p);
  return 1;
}

namespace rna {

int side_stream(rna_engine* e, hipStream_t* out) {
  if (!e->vfh_stream) {
    int prio_lo = 0, prio_hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
    // developer knob RNA_SIDE_CU_MASK=n: the side stream (VFH+, mask snapshots, launch orders) may only use the first n CUs of
    // the mask order, like RNA_ENGINE_CU_MASK for the engine stream
    int only = 0;
    if (const char* m = getenv("RNA_SIDE_CU_MASK")) only = atoi(m);
    if (only > 0 && only < e->cu_count) {
      uint32_t mask[16] = {};
      const int words = (e->cu_count + 31) / 32 < 16 ? (e->cu_count + 31) / 32 : 16;
      for (int c = 0; c < only && c < 512; ++c) mask[c >> 5] |= 1u << (c & 31);
      RNA_HIP(e, hipExtStreamCreateWithCUMask(&e->vfh_stream, (uint32_t)words, mask));
    } else {
      RNA_HIP(e, hipStreamCreateWithPriority(&e->vfh_stream, hipStreamNonBlocking, prio_hi));
    }
    RNA_HIP(e, hipEventCreateWithFlags(&e->ev_vfh_go, hipEventDisableTiming));
    RNA_HIP(e, hipEventCreateWithFlags(&e->ev_vfh_done, hipEventDisableTiming));
  }
  *out = e->vfh_stream;
  return RNA_OK;
}

int side_join(rna_engine* e) {
  if (e->vfh_pending) {
    RNA_HIP(e, hipStreamWaitEvent(e->stream, e->ev_vfh_done, 0));
    e->vfh_pending = false;
  }
  AstarDevice& a = e->astar;
  for (int d = 0; d < AstarDevice::MAX_DEPTH; ++d)
    if (a.snap_pending[d]) {
      RNA_HIP(e, hipStreamWaitEvent(e->stream, a.snap_done[d], 0));
      a.snap_pending[d] = false;
    }
  return RNA_OK;
}

int sync_all(rna_engine* e) {
  RNA_HIP(e, hipStreamSynchronize(e->stream));
  if (e->vfh_stream) RNA_HIP(e, hipStreamSynchronize(e->vfh_stream));
  for (int d = 0; d < AstarDevice::MAX_DEPTH; ++d)
    if (e->astar.side[d]) RNA_HIP(e, hipStreamSynchronize(e->astar.side[d]));
  return astar_settle(e);
}

int profile_flush(rna_engine* e) {
  if (e->pending_events == 0) return RNA_OK;
  int src = sync_all(e);
  if (src != RNA_OK) return src;
  for (auto& slot : e->prof) {
    for (auto& pr : slot.pending) {
      float ms = 0;
      if (hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) { slot.total_ms += ms; slot.launches += 1; }
      e->free_events.push_back(pr.first);
      e->free_events.push_back(pr.second);
    }
    slot.pending.clear();
  }
  e->pending_events = 0;
  return RNA_OK;
}

// Recompute A* neighbour masks where the master layer changed.
int map_prepare_nbr(rna_engine* e) {
  if (!e->nbr_all_dirty) return RNA_OK;
  { const int rc = side_join(e); if (rc != RNA_OK) return rc; }   // snapshots in flight read the masks this rebuilds
  KernelTimer kt(e, RNA_K_NBRMASK);
  hipLaunchKernelGGL(nbr_mask_tiles_kernel, dim3(e->tiles_i, e->tiles_j), dim3(256), 0, e->stream, e->nbr,
                     e->layer[RNA_LAYER_MASTER], e->dirty_tiles, 1, e->geom.size[0], e->geom.size[1], e->tiles_i,
                     e->tiles_j, e->geom.start[0], e->geom.start[1]);
  RNA_HIP(e, hipGetLastError());
  e->nbr_all_dirty = false;
  return RNA_OK;
}

}  // namespace rna

extern "C" int rna_compose_master(rna_engine* e, int mode) {
  if (!e || (mode != 0 && mode != 1)) return RNA_EINVAL;
  RNA_ENTER(e);
  const size_t words = ((size_t)e->tiles_i * e->tiles_j + 3) / 4;  // one byte per tile, rounded to words
  const bool full = (mode == 1) || e->laser_all_dirty || e->master_diverged;
  const bool moved = e->geom.start[0] != 0 || e->geom.start[1] != 0;
  if (!full && !moved && !e->nbr_all_dirty) {
    // the usual case of the replan loop: dirty tiles only, masks valid before -- one launch, and the two flag arrays
    // swap roles (what this compose consumed = rna_last_dirty_tiles; the other one, cleared by the launch, is marked next)
    KernelTimer kt(e, RNA_K_COMPOSE);
    static const int wgs_per_cu = [] { const char* v = getenv("RNA_COMPOSE_WGS_PER_CU"); const int n = v ? atoi(v) : 0; return n > 0 ? n : 16; }();   // developer knob
    hipLaunchKernelGGL(compose_nbr_tiles_kernel, dim3(std::min(e->tiles_i * e->tiles_j, wgs_per_cu * e->cu_count)), dim3(256), 0, e->stream, e->nbr,
                       e->layer[RNA_LAYER_MASTER], e->layer[RNA_LAYER_LASER], e->dirty_tiles, e->last_dirty, e->geom.size[0],
                       e->geom.size[1], e->tiles_i, e->tiles_j);
    RNA_HIP(e, hipGetLastError());
    std::swap(e->dirty_tiles, e->last_dirty);
    return RNA_OK;
  }
  {
    KernelTimer kt(e, RNA_K_COMPOSE);
    if (full) {
      RNA_HIP(e, hipMemcpyAsync(e->layer[RNA_LAYER_MASTER], e->layer[RNA_LAYER_LASER], e->ncell * sizeof(float),
                                hipMemcpyDeviceToDevice, e->stream));
    } else {
      hipLaunchKernelGGL(compose_dirty_tiles_kernel, dim3(e->tiles_i, e->tiles_j), dim3(256), 0, e->stream,
                         e->layer[RNA_LAYER_MASTER], e->layer[RNA_LAYER_LASER], e->dirty_tiles, e->geom.size[0],
                         e->geom.size[1], e->tiles_i);
      RNA_HIP(e, hipGetLastError());
    }
  }
  if (e->laser_all_dirty || e->master_diverged || moved) {
    // a layer was replaced wholesale: nothing is known about which masks are still valid; moved map: dirty tiles (buffer
    // space) do not line up with mask tiles (map space) -- the masks are rebuilt before the next search (map_prepare_nbr)
    e->nbr_all_dirty = true;
  } else if (!e->nbr_all_dirty) {
    // masks were valid before this update: refresh only tiles that are dirty or touch a dirty tile
    KernelTimer kt(e, RNA_K_NBRMASK);
    hipLaunchKernelGGL(nbr_mask_tiles_kernel, dim3(e->tiles_i, e->tiles_j), dim3(256), 0, e->stream, e->nbr,
                       e->layer[RNA_LAYER_MASTER], e->dirty_tiles, 0, e->geom.size[0], e->geom.size[1], e->tiles_i,
                       e->tiles_j, 0, 0);
    RNA_HIP(e, hipGetLastError());
  }
  // what this compose consumed: the tiles a tiled deployment has to hand to the other GPUs (rna_last_dirty_tiles)
  if (full) {
    RNA_HIP(e, hipMemsetAsync(e->last_dirty, 1, words * sizeof(unsigned), e->stream));
  } else {
    RNA_HIP(e, hipMemcpyAsync(e->last_dirty, e->dirty_tiles, words * sizeof(unsigned), hipMemcpyDeviceToDevice, e->stream));
  }
  RNA_HIP(e, hipMemsetAsync(e->dirty_tiles, 0, words * sizeof(unsigned), e->stream));
  e->laser_all_dirty = false;
  e->master_diverged = false;
  return RNA_OK;
}

// GridMap::move (gmc/src/GridMap.cpp:346-412): shift the circular-buffer origin, NaN the dropped
// rows/cols of every layer, keep position_ aligned to the grid.
extern "C" int rna_move(rna_engine* e, double nx, double ny, int* moved) {
  if (!e) return RNA_EINVAL;
  RNA_ENTER(e);
  Geom& g = e->geom;
  const double pshift[2] = {nx - g.pos[0], ny - g.pos[1]};
  int ishift[2];
  double aligned[2];
  for (int a = 0; a < 2; ++a) {  // getIndexShiftFromPositionShift / getPositionShiftFromIndexShift
    const double t = pshift[a] / g.res;
    ishift[a] = -(int)(t + 0.5 * (t > 0 ? 1 : -1));
    aligned[a] = (double)(-ishift[a]) * g.res;
  }
  auto clear = [&](int i0, int ni, int j0, int nj) -> int {
    if (ni <= 0 || nj <= 0) return RNA_OK;
    for (int l = 0; l < RNA_NUM_LAYERS; ++l) {
      hipLaunchKernelGGL(clear_region_kernel, dim3(grid_for((size_t)ni * nj, 256)), dim3(256), 0, e->stream,
                         e->layer[l], g.size[0], i0, ni, j0, nj);
      RNA_HIP(e, hipGetLastError());
    }
    return RNA_OK;
  };
  int rc = RNA_OK;
  for (int a = 0; a < 2 && rc == RNA_OK; ++a) {
    if (ishift[a] == 0) continue;
    const int asz = g.size[a];
    const int ash = ishift[a] < 0 ? -ishift[a] : ishift[a];
    if (ash >= asz) {
      rc = clear(0, g.size[0], 0, g.size[1]);
    } else {
      const int sign = ishift[a] > 0 ? 1 : -1;
      const int start_index = g.start[a] - (sign < 0 ? 1 : 0);
      const int end_index = start_index - sign + ishift[a];
      const int index = wrap_index(sign > 0 ? start_index : end_index, asz);
      const int first = (index + ash <= asz) ? ash : asz - index;
      const int second = ash - first;
      if (a == 0) {
        rc = clear(index, first, 0, g.size[1]);
        if (rc == RNA_OK && second > 0) rc = clear(0, second, 0, g.size[1]);
      } else {
        rc = clear(0, g.size[0], index, first);
        if (rc == RNA_OK && second > 0) rc = clear(0, g.size[0], 0, second);
      }
    }
  }
  if (rc != RNA_OK) return rc;
  g.start[0] = wrap_index(g.start[0] + ishift[0], g.size[0]);
  g.start[1] = wrap_index(g.start[1] + ishift[1], g.size[1]);
  g.pos[0] += aligned[0];
  g.pos[1] += aligned[1];
  if (ishift[0] != 0 || ishift[1] != 0) e->nbr_all_dirty = true;
  if (moved) *moved = (ishift[0] != 0 || ishift[1] != 0);
  return RNA_OK;
}

extern "C" int rna_profile_enable(rna_engine* e, int on) {
  if (!e) return RNA_EINVAL;
  e->profiling = on < 0 ? 0 : (on > 2 ? 1 : on);
  return RNA_OK;
}

extern "C" int rna_profile_reset(rna_engine* e) {
  if (!e) return RNA_EINVAL;
  RNA_ENTER(e);
  int rc = profile_flush(e);
  if (rc != RNA_OK) return rc;
  for (auto& p : e->prof) { p.total_ms = 0; p.launches = 0; }
  return RNA_OK;
}

extern "C" int rna_profile_get(rna_engine* e, int id, double* total_ms, int64_t* launches) {
  if (!e || id < 0 || id >= RNA_K_COUNT) return RNA_EINVAL;
  RNA_ENTER(e);
  int rc = profile_flush(e);
  if (rc != RNA_OK) return rc;
  if (total_ms) *total_ms = e->prof[id].total_ms;
  if (launches) *launches = e->prof[id].launches;
  return RNA_OK;
}
