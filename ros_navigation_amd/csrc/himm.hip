// himm.hip -- order-faithful HIMM ray batches on gfx950.
//
// Replaces MapUpdater::lineOnMap / clearCell / markCell applied to every buffered RangeSample in
// arrival order (mc/include/move_control/map_updater.h:38-71, mc/src/laser_map_updater.cpp:7-21,
// mc/src/range_map_updater.cpp:7-21) with LineIterator's clipping walk and integer Bresenham
// (gmc/src/iterators/LineIterator.cpp:60-70,92-150).
//
// clearCell and markCell do not commute, so a plain atomic scatter is not parity-safe.  Design:
//   * clears commute with each other (they are the same function applied k times), so every cell
//     that receives NO mark in this batch takes its clears as order-free compare-and-swap updates;
//   * cells that hold >= 1 mark (ray end points; at most n of them) are found through a 1-bit-per-
//     cell bitmap + an open-addressing hash; their clears are only COUNTED, per interval between
//     consecutive marks of that cell (marks sorted by ray sequence number), and one thread per
//     marked cell then replays  clear^k0 mark clear^k1 mark ... clear^kn  exactly.
// The result equals the sequential reference for any float contents of the layer.
//
// Kernels: himm_prep (1 thread/ray: double-precision clipping, mark registration),
//          himm_collect (1 thread/hash slot: gather + sort that cell's marks),
//          himm_raster (16 lanes/ray, closed-form Bresenham cell k, HBM read-modify-write),
//          himm_apply (1 thread/hash slot: ordered replay on marked cells).
#include "engine.hpp"

using namespace rna;

namespace {

constexpr int LANES_PER_RAY = 16;

__device__ __forceinline__ float himm_clear(float v) {  // map_updater.h:61-71
  if (v <= 0.0f || v != v) v = 0.0f;
  else v = v - 10.0f;
  if (v < 0.0f) v = 0.0f;
  return v;
}

__device__ __forceinline__ float himm_mark(float v) {  // map_updater.h:52-59
  if (v <= 0.0f || v != v) return 30.0f;
  if (v <= 150.0f) return v + 30.0f;
  return v;
}

__device__ __forceinline__ float himm_clear_n(float v, unsigned k) {
  for (unsigned i = 0; i < k; ++i) {
    v = himm_clear(v);
    if (__float_as_int(v) == 0) break;  // +0.0 is a fixed point of clearCell
  }
  return v;
}

__device__ __forceinline__ unsigned hash_cell(unsigned c) {
  c ^= c >> 16; c *= 0x7feb352dU; c ^= c >> 15; c *= 0x846ca68bU; c ^= c >> 16;
  return c;
}

// A ray the reference could not digest either: a non-finite coordinate makes its clipping march
// (LineIterator.cpp:92-104) spin forever, a start millions of cells away makes it spin for minutes.  Defined here
// and in oracle/himm.c: such a ray is dropped whole (no clears, no mark).  HIMM_MAX_RAY_CELLS cells is 52 km at
// the reference's 0.05 m resolution; the march below needs at most length / step + 1 iterations.
constexpr double HIMM_MAX_RAY_CELLS = 1048576.0;
__device__ __forceinline__ bool ray_well_formed(const Geom& g, const rna_ray& r) {
  if (!(isfinite(r.sx) && isfinite(r.sy) && isfinite(r.ex) && isfinite(r.ey))) return false;
  const double vx = r.ex - r.sx, vy = r.ey - r.sy;
  return sqrt(vx * vx + vy * vy) <= HIMM_MAX_RAY_CELLS * g.res;
}

// LineIterator::getIndexLimitedToMapRange (LineIterator.cpp:92-104): march `start` towards `end`
// in (res - eps) steps until it is inside the map.
__device__ bool index_limited_to_map(const Geom& g, double sx, double sy, double ex, double ey, int idx[2]) {
  double px = sx, py = sy;
  const double vx = ex - sx, vy = ey - sy;
  const double nrm = sqrt(vx * vx + vy * vy);
  const double dx = vx / nrm, dy = vy / nrm;
  const double step = g.res - DBL_EPSILON;
  while (!index_from_position(g, px, py, idx)) {
    if (!(nrm > 0.0)) return false;  // zero-length ray outside the map (reference: undefined)
    px += step * dx;
    py += step * dy;
    const double rx = ex - px, ry = ey - py;
    if (sqrt(rx * rx + ry * ry) < step) return false;
  }
  return true;
}

__global__ void himm_init_slots_kernel(HimmSlot* __restrict__ slots, int n_slots, int* __restrict__ total) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_slots) slots[i] = HimmSlot{-1, -1, 0, 0};
  if (i == 0) *total = 0;
}

__global__ void himm_prep_kernel(Geom g, const rna_ray* __restrict__ rays, int n, int4* __restrict__ desc,
                                 int* __restrict__ ncells, int* __restrict__ next, HimmSlot* __restrict__ slots,
                                 int slot_mask, unsigned* __restrict__ mark_bitmap, int4 win) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  const rna_ray ray = rays[r];
  int s[2], t[2];
  int nc = 0;
  if (!ray_well_formed(g, ray)) {
    ncells[r] = 0;
    next[r] = -1;
    return;
  }
  if (index_limited_to_map(g, ray.sx, ray.sy, ray.ex, ray.ey, s) &&
      index_limited_to_map(g, ray.ex, ray.ey, ray.sx, ray.sy, t)) {
    const int dx = abs(t[0] - s[0]), dy = abs(t[1] - s[1]);
    nc = (dx >= dy ? dx : dy) + 1;
    desc[r] = make_int4(s[0], s[1], t[0], t[1]);
  }
  ncells[r] = nc;
  next[r] = -1;
  if (!ray.clear_end) {  // map_updater.h:44-49
    int ei[2];
    if (index_from_position(g, ray.ex, ray.ey, ei) && ei[0] >= win.x && ei[0] < win.y && ei[1] >= win.z &&
        ei[1] < win.w) {  // marks outside the owner window belong to another tile's GPU
      const int cell = ei[1] * g.size[0] + ei[0];
      unsigned h = hash_cell((unsigned)cell) & (unsigned)slot_mask;
      for (;;) {
        const int prev = atomicCAS(&slots[h].cell, -1, cell);
        if (prev == -1 || prev == cell) break;
        h = (h + 1) & (unsigned)slot_mask;
      }
      next[r] = atomicExch(&slots[h].head, r);
      atomicAdd(&slots[h].len, 1);
      atomicOr(&mark_bitmap[cell >> 5], 1u << (cell & 31));
    }
  }
}

__device__ void sift_down(int* a, int start, int end) {
  int root = start;
  for (;;) {
    int child = 2 * root + 1;
    if (child > end) break;
    if (child + 1 <= end && a[child] < a[child + 1]) child++;
    if (a[root] < a[child]) { const int t = a[root]; a[root] = a[child]; a[child] = t; root = child; }
    else break;
  }
}

__global__ void himm_collect_kernel(HimmSlot* __restrict__ slots, int n_slots, const int* __restrict__ next,
                                    int* __restrict__ seqs, unsigned* __restrict__ before,
                                    unsigned* __restrict__ after, int* __restrict__ total) {
  const int h = blockIdx.x * blockDim.x + threadIdx.x;
  if (h >= n_slots) return;
  HimmSlot sl = slots[h];
  if (sl.cell < 0) return;
  const int off = atomicAdd(total, sl.len);
  slots[h].offset = off;
  int* a = seqs + off;
  int k = 0;
  for (int r = sl.head; r >= 0; r = next[r]) a[k++] = r;
  const int len = k;
  if (len <= 24) {  // insertion sort
    for (int i = 1; i < len; ++i) {
      const int v = a[i];
      int j = i - 1;
      while (j >= 0 && a[j] > v) { a[j + 1] = a[j]; --j; }
      a[j + 1] = v;
    }
  } else {  // heap sort: many hits on one cell (e.g. a wall seen at close range)
    for (int st = (len - 2) / 2; st >= 0; --st) sift_down(a, st, len - 1);
    for (int end = len - 1; end > 0; --end) {
      const int t = a[end]; a[end] = a[0]; a[0] = t;
      sift_down(a, 0, end - 1);
    }
  }
  for (int i = 0; i < len; ++i) { before[off + i] = 0; after[off + i] = 0; }
}

__global__ void himm_raster_kernel(int rows, const int4* __restrict__ desc, const int* __restrict__ ncells, int n,
                                   float* __restrict__ layer, const unsigned* __restrict__ mark_bitmap,
                                   const HimmSlot* __restrict__ slots, int slot_mask,
                                   const int* __restrict__ seqs, unsigned* __restrict__ before,
                                   unsigned* __restrict__ after, unsigned* __restrict__ dirty_tiles, int tiles_i,
                                   int4 win) {
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  const int r = gid / LANES_PER_RAY;
  const int lane = gid % LANES_PER_RAY;
  if (r >= n) return;
  const int nc = ncells[r];
  if (nc == 0) return;
  const int4 d = desc[r];
  const int adx = abs(d.z - d.x), ady = abs(d.w - d.y);
  const int sx = d.z >= d.x ? 1 : -1, sy = d.w >= d.y ? 1 : -1;
  const bool xmajor = adx >= ady;           // LineIterator.cpp:133-149
  const int den = xmajor ? adx : ady;
  const int add = xmajor ? ady : adx;
  const int num0 = den / 2;
  int last_tile = -1;
  for (int k = lane; k < nc; k += LANES_PER_RAY) {
    // cell k of the Bresenham walk in closed form: the minor axis has stepped
    // floor((num0 + k*add) / den) times after k increments (LineIterator.cpp:60-70).
    const int m = den > 0 ? (int)(((long long)num0 + (long long)k * add) / den) : 0;
    const int i = d.x + (xmajor ? k : m) * sx;
    const int j = d.y + (xmajor ? m : k) * sy;
    // tiled single map (SURVEY 8e mode 2): the line is rasterised on the GLOBAL geometry and only
    // the cells of this GPU's window are written, so every cell sees exactly the full batch's ops
    if (i < win.x || i >= win.y || j < win.z || j >= win.w) continue;
    const int cell = j * rows + i;
    const int tile = (j >> 6) * tiles_i + (i >> 6);
    if (tile != last_tile && dirty_tiles) {   // flags exist for the laser layer only: "laser differs from master here"
      // one byte per tile, plain load + plain store of the same value by every first toucher: no
      // atomics (a bit-packed atomicOr here serialised the whole batch on four cache lines)
      volatile unsigned char* flag = reinterpret_cast<volatile unsigned char*>(dirty_tiles) + tile;
      if (!*flag) *flag = 1;
      last_tile = tile;
    }
    if ((mark_bitmap[cell >> 5] >> (cell & 31)) & 1u) {
      // marked cell: count this clear in the interval before the first mark with seq >= r
      unsigned h = hash_cell((unsigned)cell) & (unsigned)slot_mask;
      while (slots[h].cell != cell) h = (h + 1) & (unsigned)slot_mask;
      const int off = slots[h].offset, len = slots[h].len;
      int lo = 0, hi = len;
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (seqs[off + mid] < r) lo = mid + 1; else hi = mid;
      }
      if (lo < len) atomicAdd(&before[off + lo], 1u);
      else atomicAdd(&after[off + len - 1], 1u);
    } else {
      // unmarked cell: clears commute -> lock-free read-modify-write on the float bits
      int* p = reinterpret_cast<int*>(&layer[cell]);
      int old = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      for (;;) {
        const int nv = __float_as_int(himm_clear(__int_as_float(old)));
        if (nv == old) break;
        const int seen = atomicCAS(p, old, nv);
        if (seen == old) break;
        old = seen;
      }
    }
  }
}

__global__ void himm_apply_kernel(int rows, const HimmSlot* __restrict__ slots, int n_slots,
                                  const unsigned* __restrict__ before, const unsigned* __restrict__ after,
                                  float* __restrict__ layer, unsigned* __restrict__ mark_bitmap,
                                  unsigned* __restrict__ dirty_tiles, int tiles_i) {
  const int h = blockIdx.x * blockDim.x + threadIdx.x;
  if (h >= n_slots) return;
  const HimmSlot sl = slots[h];
  if (sl.cell < 0) return;
  float v = layer[sl.cell];
  for (int k = 0; k < sl.len; ++k) {
    v = himm_clear_n(v, before[sl.offset + k]);
    v = himm_mark(v);
  }
  v = himm_clear_n(v, after[sl.offset + sl.len - 1]);
  layer[sl.cell] = v;
  atomicAnd(&mark_bitmap[sl.cell >> 5], ~(1u << (sl.cell & 31)));  // leave the bitmap all-zero
  const int i = sl.cell % rows, j = sl.cell / rows;
  const int tile = (j >> 6) * tiles_i + (i >> 6);
  if (dirty_tiles) reinterpret_cast<volatile unsigned char*>(dirty_tiles)[tile] = 1;
}

int ensure_scratch(rna_engine* e, int n) {
  HimmScratch& s = e->himm;
  int rc;
  if (!s.mark_bitmap) {
    const size_t words = (e->ncell + 31) / 32;
    if ((rc = dev_alloc(e, &s.mark_bitmap, words)) != RNA_OK) return rc;
    RNA_HIP(e, hipMemsetAsync(s.mark_bitmap, 0, words * sizeof(unsigned), e->stream));
    if ((rc = dev_alloc(e, &s.total, 1)) != RNA_OK) return rc;
  }
  if (n <= s.cap_rays) return RNA_OK;
  int cap = 1024;
  while (cap < n) cap <<= 1;
  RNA_HIP(e, hipStreamSynchronize(e->stream));
  // a failed regrow leaves no half-sized scratch behind: the next call starts from nothing again
  s.cap_rays = 0;
  s.n_slots = 0;
  if ((rc = dev_alloc(e, &s.rays_dev, (size_t)cap)) != RNA_OK || (rc = dev_alloc(e, &s.desc, (size_t)cap)) != RNA_OK ||
      (rc = dev_alloc(e, &s.ncells, (size_t)cap)) != RNA_OK || (rc = dev_alloc(e, &s.next, (size_t)cap)) != RNA_OK ||
      (rc = dev_alloc(e, &s.seqs, (size_t)cap)) != RNA_OK || (rc = dev_alloc(e, &s.before, (size_t)cap)) != RNA_OK ||
      (rc = dev_alloc(e, &s.after, (size_t)cap)) != RNA_OK || (rc = dev_alloc(e, &s.slots, (size_t)cap * 2)) != RNA_OK) {
    const std::string why = e->err;
    (void)himm_release(e);
    e->err = why;
    return rc;
  }
  s.n_slots = cap * 2;
  s.cap_rays = cap;
  return RNA_OK;
}

int himm_launch(rna_engine* e, int layer, const rna_ray* rays_dev, int n) {
  HimmScratch& s = e->himm;
  const Geom g = e->geom;
  // hash sized to the batch: a power of two >= 2n keeps probe sequences short
  int n_slots = 1024;
  while (n_slots < 2 * n) n_slots <<= 1;
  if (n_slots > s.n_slots) n_slots = s.n_slots;
  const int4 win = s.win[1] > 0 ? make_int4(s.win[0], s.win[1], s.win[2], s.win[3])
                                : make_int4(0, g.size[0], 0, g.size[1]);
  {
    KernelTimer kt(e, RNA_K_HIMM_PREP);
    hipLaunchKernelGGL(himm_init_slots_kernel, dim3((n_slots + 255) / 256), dim3(256), 0, e->stream, s.slots,
                       n_slots, s.total);
    hipLaunchKernelGGL(himm_prep_kernel, dim3((n + 255) / 256), dim3(256), 0, e->stream, g, rays_dev, n, s.desc,
                       s.ncells, s.next, s.slots, n_slots - 1, s.mark_bitmap, win);
    hipLaunchKernelGGL(himm_collect_kernel, dim3((n_slots + 255) / 256), dim3(256), 0, e->stream, s.slots, n_slots,
                       s.next, s.seqs, s.before, s.after, s.total);
    RNA_HIP(e, hipGetLastError());
  }
  {
    KernelTimer kt(e, RNA_K_HIMM_RASTER);
    const long long threads = (long long)n * LANES_PER_RAY;
    hipLaunchKernelGGL(himm_raster_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, e->stream,
                       g.size[0], s.desc, s.ncells, n, e->layer[layer], s.mark_bitmap, s.slots, n_slots - 1,
                       s.seqs, s.before, s.after, layer == RNA_LAYER_LASER ? e->dirty_tiles : nullptr, e->tiles_i, win);
    RNA_HIP(e, hipGetLastError());
  }
  {
    KernelTimer kt(e, RNA_K_HIMM_APPLY);
    hipLaunchKernelGGL(himm_apply_kernel, dim3((n_slots + 255) / 256), dim3(256), 0, e->stream, g.size[0], s.slots,
                       n_slots, s.before, s.after, e->layer[layer], s.mark_bitmap, layer == RNA_LAYER_LASER ? e->dirty_tiles : nullptr, e->tiles_i);
    RNA_HIP(e, hipGetLastError());
  }
  if (layer == RNA_LAYER_MASTER) { e->nbr_all_dirty = true; e->master_diverged = true; }
  return RNA_OK;
}

}  // namespace

namespace rna {
int himm_release(rna_engine* e) {
  HimmScratch& s = e->himm;
  dev_free(&s.rays_dev); dev_free(&s.desc); dev_free(&s.ncells); dev_free(&s.next); dev_free(&s.slots);
  dev_free(&s.seqs); dev_free(&s.before); dev_free(&s.after); dev_free(&s.total); dev_free(&s.mark_bitmap);
  s.cap_rays = 0;
  s.n_slots = 0;
  return RNA_OK;
}
}  // namespace rna

extern "C" int rna_himm_set_window(rna_engine* e, int i0, int j0, int ni, int nj) {
  if (!e) return RNA_EINVAL;
  if (ni <= 0 || nj <= 0) {  // back to the whole map
    e->himm.win[0] = e->himm.win[1] = e->himm.win[2] = e->himm.win[3] = 0;
    return RNA_OK;
  }
  if (i0 < 0 || j0 < 0 || i0 + ni > e->geom.size[0] || j0 + nj > e->geom.size[1])
    return fail(e, RNA_EINVAL, "rna_himm_set_window: window outside the map");
  e->himm.win[0] = i0; e->himm.win[1] = i0 + ni; e->himm.win[2] = j0; e->himm.win[3] = j0 + nj;
  return RNA_OK;
}

extern "C" int rna_himm_update_device(rna_engine* e, int layer, const rna_ray* rays_device, int n) {
  if (!e || layer < 0 || layer >= RNA_NUM_LAYERS || n < 0 || (n > 0 && !rays_device)) return RNA_EINVAL;
  if (n == 0) return RNA_OK;
  RNA_HIP(e, hipSetDevice(e->device));
  int rc = ensure_scratch(e, n);
  if (rc != RNA_OK) return rc;
  return himm_launch(e, layer, rays_device, n);
}

extern "C" int rna_himm_update(rna_engine* e, int layer, const rna_ray* rays_host, int n) {
  if (!e || layer < 0 || layer >= RNA_NUM_LAYERS || n < 0 || (n > 0 && !rays_host)) return RNA_EINVAL;
  if (n == 0) return RNA_OK;
  RNA_HIP(e, hipSetDevice(e->device));
  int rc = ensure_scratch(e, n);
  if (rc != RNA_OK) return rc;
  RNA_HIP(e, hipMemcpyAsync(e->himm.rays_dev, rays_host, (size_t)n * sizeof(rna_ray), hipMemcpyHostToDevice,
                            e->stream));
  rc = himm_launch(e, layer, e->himm.rays_dev, n);
  if (rc != RNA_OK) return rc;
  RNA_HIP(e, hipStreamSynchronize(e->stream));
  return RNA_OK;
}

extern "C" int rna_update_map_device(rna_engine* e, const rna_ray* rays_device, int n, int compose_mode) {
  int rc = rna_himm_update_device(e, RNA_LAYER_LASER, rays_device, n);
  if (rc != RNA_OK) return rc;
  return rna_compose_master(e, compose_mode);
}

extern "C" int rna_update_map(rna_engine* e, const rna_ray* rays_host, int n, int compose_mode) {
  int rc = rna_himm_update(e, RNA_LAYER_LASER, rays_host, n);
  if (rc != RNA_OK) return rc;
  rc = rna_compose_master(e, compose_mode);
  if (rc != RNA_OK) return rc;
  return rna_synchronize(e);
}
