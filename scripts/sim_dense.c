/* sim_dense.c -- developer tool: CPU model of the dense (register-resident sweep) tile job of the
 * grid A* kernel.  Counts tile jobs, sweeps, rounds and buckets for a batch of queries so that tile
 * shape, bucket width and job order can be chosen before the kernel is written; checks cost and E
 * against the values the driver script passes in (from the oracle).
 *
 *   sim_dense <workload.bin> TI TJ bucket mode [nq]
 *     TI   tile cells along i (the lane axis), TJ tile cells along j (the register axis)
 *     mode 0 = rounds (every active tile of a query runs once per round), 1 = best-first (one job at a time,
 *          lowest activation key first)
 * workload.bin: int32 rows, cols, nq; nbr[rows*cols] u8; nq x {int32 start, goal, cost, settled}
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define INF 0x3fffffff
static int rows, cols, TI, TJ, tiles_i, tiles_j, bucket_w, mode, hclose;
static uint8_t* nbr;
static int32_t* g;
static int gi, gj;

static inline int octile(int i, int j) {
  int dx = abs(i - gi), dy = abs(j - gj);
  int mx = dx > dy ? dx : dy, mn = dx > dy ? dy : dx;
  return 1000 * mx + 414 * mn;
}

typedef struct { long jobs, sweeps, rounds, buckets, tiles, maxjobs_round, thin_rounds, cells_changed; } stats;

/* per-query scheduler state */
static uint8_t *act_cur, *act_far, *ran, *touched;
static int32_t* key_cur; /* best-first: activation key */
static int best;
static long long bend;

static inline int prop(int v, int i, int j) {
  if (v >= INF) return INF;
  long long f = (long long)v + octile(i, j);
  return (f < bend && f <= best) ? v : INF;
}

static void activate(int ti, int tj, int key) {
  if (ti < 0 || tj < 0 || ti >= tiles_i || tj >= tiles_j) return;
  int t = tj * tiles_i + ti;
  if (!act_cur[t] || key < key_cur[t]) key_cur[t] = key;
  act_cur[t] = 1;
}

/* one dense job on tile (ti,tj); returns number of sweeps */
static int job(int ti, int tj, stats* st) {
  const int i0 = ti * TI, j0 = tj * TJ;
  const int W = TI + 2, H = TJ + 2;
  static int32_t *cur = NULL, *old = NULL, *pp = NULL;
  static uint8_t* mk = NULL;
  if (!cur) { cur = malloc(sizeof(int32_t) * 70 * 70); old = malloc(sizeof(int32_t) * 70 * 70); pp = malloc(sizeof(int32_t) * 70 * 70); mk = malloc(70 * 70); }
  const int t = tj * tiles_i + ti;
  const int first = !ran[t];
  ran[t] = 1;
  touched[t] = 1;
  /* load tile + halo */
  for (int b = 0; b < H; ++b)
    for (int a = 0; a < W; ++a) {
      int i = i0 + a - 1, j = j0 + b - 1;
      int v = INF; uint8_t m = 0;
      if (i >= 0 && j >= 0 && i < rows && j < cols) { v = g[(size_t)j * rows + i]; m = nbr[(size_t)j * rows + i]; }
      cur[b * W + a] = v; old[b * W + a] = v; mk[b * W + a] = m;
      pp[b * W + a] = prop(v, i, j);
    }
  int sweeps = 0, quiet = 0, dir = 0;
  /* goal inside this tile?  its g is the bound */
  while (quiet < 2) {
    int changed = 0;
    for (int s = 0; s < TJ; ++s) {
      const int b = dir == 0 ? 1 + s : TJ - s;      /* row inside the (halo-padded) tile */
      const int bp = dir == 0 ? b - 1 : b + 1;      /* source row */
      int32_t tmp[70];
      for (int a = 1; a <= TI; ++a) {
        const int i = i0 + a - 1, j = j0 + b - 1;
        if (i >= rows || j >= cols) { tmp[a] = INF; continue; }
        int v = cur[b * W + a];
        const uint8_t m = mk[b * W + a];
        /* neighbour bits: k0=(-1,-1) k1=(0,-1) k2=(1,-1) k3=(-1,0) k4=(1,0) k5=(-1,1) k6=(0,1) k7=(1,1) */
        const int kd0 = dir == 0 ? 0 : 5, kd1 = dir == 0 ? 1 : 6, kd2 = dir == 0 ? 2 : 7;
        if ((m >> kd1) & 1) { int c = pp[bp * W + a] + 1000; if (c < v) v = c; }
        if ((m >> kd0) & 1) { int c = pp[bp * W + a - 1] + 1414; if (c < v) v = c; }
        if ((m >> kd2) & 1) { int c = pp[bp * W + a + 1] + 1414; if (c < v) v = c; }
        tmp[a] = v;
      }
      /* horizontal (Jacobi, from the p of the vertically updated row) */
      int32_t ph[70];
      ph[0] = pp[b * W + 0]; ph[TI + 1] = pp[b * W + TI + 1];
      for (int a = 1; a <= TI; ++a) ph[a] = prop(tmp[a], i0 + a - 1, j0 + b - 1);
      if (hclose) {   /* exact horizontal closure of the row (segmented min-plus scan on the GPU) */
        for (int a = 1; a <= TI; ++a) {
          const int i = i0 + a - 1, j = j0 + b - 1;
          if (i >= rows || j >= cols) continue;
          if ((mk[b * W + a] >> 3) & 1) { int c = ph[a - 1] + 1000; if (c < tmp[a]) { tmp[a] = c; ph[a] = prop(c, i, j); } }
        }
        for (int a = TI; a >= 1; --a) {
          const int i = i0 + a - 1, j = j0 + b - 1;
          if (i >= rows || j >= cols) continue;
          if ((mk[b * W + a] >> 4) & 1) { int c = ph[a + 1] + 1000; if (c < tmp[a]) { tmp[a] = c; ph[a] = prop(c, i, j); } }
        }
      }
      for (int a = 1; a <= TI; ++a) {
        const int i = i0 + a - 1, j = j0 + b - 1;
        if (i >= rows || j >= cols) continue;
        int v = tmp[a];
        const uint8_t m = mk[b * W + a];
        if (!hclose && ((m >> 3) & 1)) { int c = ph[a - 1] + 1000; if (c < v) v = c; }
        if (!hclose && ((m >> 4) & 1)) { int c = ph[a + 1] + 1000; if (c < v) v = c; }
        if (v < cur[b * W + a]) {
          cur[b * W + a] = v; changed = 1;
          if (i == gi && j == gj && v < best) best = v;
        }
        pp[b * W + a] = prop(cur[b * W + a], i, j);
      }
    }
    ++sweeps;
    quiet = changed ? 0 : quiet + 1;
    dir ^= 1;
  }
  st->sweeps += sweeps;
  /* write back, activation */
  int far = 0;
  for (int b = 1; b <= TJ; ++b)
    for (int a = 1; a <= TI; ++a) {
      const int i = i0 + a - 1, j = j0 + b - 1;
      if (i >= rows || j >= cols) continue;
      const int v = cur[b * W + a];
      const int ch = v < old[b * W + a];
      if (ch) { g[(size_t)j * rows + i] = v; st->cells_changed++; }
      if (v >= INF) continue;
      const long long f = (long long)v + octile(i, j);
      if (f > best) continue;
      if (f >= bend) { far = 1; continue; }
      const int newly = first && f >= bend - bucket_w;
      if (!(ch || newly)) continue;
      const int ea = a == 1 ? -1 : (a == TI ? 1 : 0), eb = b == 1 ? -1 : (b == TJ ? 1 : 0);
      const int key = (int)(mode == 2 ? v : f);
      if (ea) activate(ti + ea, tj, key);
      if (eb) activate(ti, tj + eb, key);
      if (ea && eb) activate(ti + ea, tj + eb, key);
    }
  if (far) act_far[t] = 1;
  return sweeps;
}

int main(int argc, char** argv) {
  if (argc < 6) { fprintf(stderr, "usage\n"); return 2; }
  FILE* f = fopen(argv[1], "rb");
  TI = atoi(argv[2]); TJ = atoi(argv[3]); bucket_w = atoi(argv[4]); mode = atoi(argv[5]);
  int32_t hdr[3];
  if (fread(hdr, 4, 3, f) != 3) return 1;
  rows = hdr[0]; cols = hdr[1];
  int nq = hdr[2];
  if (argc > 6 && atoi(argv[6]) < nq) nq = atoi(argv[6]);
  hclose = argc > 7 ? atoi(argv[7]) : 0;
  nbr = malloc((size_t)rows * cols);
  if (fread(nbr, 1, (size_t)rows * cols, f) != (size_t)rows * cols) return 1;
  int32_t* qs = malloc(sizeof(int32_t) * 4 * hdr[2]);
  if (fread(qs, 16, hdr[2], f) != (size_t)hdr[2]) return 1;
  fclose(f);
  tiles_i = (rows + TI - 1) / TI; tiles_j = (cols + TJ - 1) / TJ;
  const int ntile = tiles_i * tiles_j;
  g = malloc(sizeof(int32_t) * (size_t)rows * cols);
  act_cur = calloc(ntile, 1); act_far = calloc(ntile, 1); ran = calloc(ntile, 1); touched = calloc(ntile, 1);
  key_cur = malloc(sizeof(int32_t) * ntile);
  stats tot = {0};
  long totE = 0, bad = 0;
  long hist[8] = {0};
  for (int q = 0; q < nq; ++q) {
    const int start = qs[4 * q], goal = qs[4 * q + 1], want_cost = qs[4 * q + 2], want_E = qs[4 * q + 3];
    for (size_t c = 0; c < (size_t)rows * cols; ++c) g[c] = INF;
    memset(act_cur, 0, ntile); memset(act_far, 0, ntile); memset(ran, 0, ntile); memset(touched, 0, ntile);
    gi = goal % rows; gj = goal / rows;
    const int si = start % rows, sj = start / rows;
    g[start] = 0;
    best = INF;
    if (start == goal) best = 0;
    if (getenv("SIM_KNOWN")) best = want_cost + atoi(getenv("SIM_KNOWN"));
    stats st = {0};
    long long bucket = octile(si, sj) / bucket_w;
    bend = (bucket + 1) * (long long)bucket_w;
    activate(si / TI, sj / TJ, 0);
    st.buckets = 1;
    for (;;) {
      /* collect active tiles */
      int n = 0;
      static int* list = NULL;
      if (!list) list = malloc(sizeof(int) * ntile);
      if (mode == 0) {
        for (int t = 0; t < ntile; ++t) if (act_cur[t]) { list[n++] = t; act_cur[t] = 0; }
      } else {
        int bt = -1;
        for (int t = 0; t < ntile; ++t) if (act_cur[t] && (bt < 0 || key_cur[t] < key_cur[bt])) bt = t;
        if (bt >= 0) { list[n++] = bt; act_cur[bt] = 0; }
      }
      if (n == 0) {
        if (best != INF && best < bend) break;
        int any = 0;
        for (int t = 0; t < ntile; ++t) { if (act_far[t]) { act_cur[t] = 1; key_cur[t] = 0; any = 1; } act_far[t] = 0; ran[t] = 0; }
        if (!any) break;
        bucket += 1; bend = (bucket + 1) * (long long)bucket_w;
        st.buckets++;
        continue;
      }
      st.rounds++;
      if (n > st.maxjobs_round) st.maxjobs_round = n;
      { int b = n <= 1 ? 0 : n <= 2 ? 1 : n <= 4 ? 2 : n <= 8 ? 3 : n <= 16 ? 4 : n <= 32 ? 5 : n <= 64 ? 6 : 7; hist[b] += n; }
      for (int k = 0; k < n; ++k) { job(list[k] % tiles_i, list[k] / tiles_i, &st); st.jobs++; }
    }
    long E = 0, reached = 0;
    for (int t = 0; t < ntile; ++t) st.tiles += touched[t];
    for (int j = 0; j < cols; ++j)
      for (int i = 0; i < rows; ++i) {
        int v = g[(size_t)j * rows + i];
        if (v < INF) { reached++; if (best != INF && v + octile(i, j) <= best) E++; }
      }
    const int ok = (best == want_cost || (best == INF && want_cost >= 0x7fffffff)) && E == want_E;
    if (!ok) { bad++; fprintf(stderr, "query %d MISMATCH cost %d want %d E %ld want %d\n", q, best, want_cost, E, want_E); }
    totE += E;
    printf("q%-3d E %7ld reached %7ld tiles %5ld jobs %6ld (%.2f/tile) sweeps %7ld (%.2f/job) rounds %5ld buckets %3ld cost %d\n", q, E, reached, st.tiles,
           st.jobs, (double)st.jobs / (st.tiles ? st.tiles : 1), st.sweeps, (double)st.sweeps / (st.jobs ? st.jobs : 1), st.rounds, st.buckets, best);
    tot.jobs += st.jobs; tot.sweeps += st.sweeps; tot.rounds += st.rounds; tot.buckets += st.buckets; tot.tiles += st.tiles; tot.cells_changed += st.cells_changed;
  }
  const double cells_per_tile = (double)TI * TJ;
  /* cost model: per job ~300 wave instructions of load / store / activation, per sweep 21 * TJ */
  const double instr = tot.jobs * 300.0 + tot.sweeps * 21.0 * TJ * (TI > 64 ? TI / 64.0 : 1.0);
  printf("TOTAL nq %d TI %d TJ %d bucket %d mode %d: E %ld tiles %ld jobs %ld (%.2f/tile) sweeps %ld (%.2f/job) rounds %ld buckets %ld  writes/E %.2f\n", nq, TI, TJ,
         bucket_w, mode, totE, tot.tiles, tot.jobs, (double)tot.jobs / tot.tiles, tot.sweeps, (double)tot.sweeps / tot.jobs, tot.rounds, tot.buckets,
         (double)tot.cells_changed / totE);
  printf("  model: %.3g wave-instr per query, %.1f per settled cell; dense cell-updates per settled cell %.1f; jobs by round size 1:%ld 2:%ld 3-4:%ld 5-8:%ld 9-16:%ld 17-32:%ld 33-64:%ld 65+:%ld mismatches %ld\n",
         instr / nq, instr / totE, tot.sweeps * cells_per_tile / totE, hist[0], hist[1], hist[2], hist[3], hist[4], hist[5], hist[6], hist[7], bad);
  return bad ? 1 : 0;
}
