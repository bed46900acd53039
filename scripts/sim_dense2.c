/* sim_dense2.c -- developer tool: CPU model of the register-resident dense tile job as the GPU would run it:
 * 64 x 16-cell tiles (lane = i, register = j), rounds of jobs that read their halo from the state at the START of the
 * round (jobs of a round run concurrently on the GPU), halo contributions applied once at load, alternating down / up
 * sweeps with per-row dirty flags (a row is re-evaluated only if one of its sources changed), bucketed by f.
 * Checks cost and E against the oracle's values and prints the counts the instruction model needs.
 *
 *   sim_dense2 <workload.bin> bucket [nq] [variant] [TJ]
 *     variant 0: horizontal step from the row's values after this sweep's vertical step ("fresh")
 *             1: horizontal step from the row's values before this sweep touched it ("stale": one max3 chain per row)
 *             + 10 x the number of extra passes a changed row makes along itself (the kernel: 161)
 *   environment: SIM_REDBLACK=1  rounds alternate between the two checkerboard colours of the tiles
 *                SIM_FILTER=1    a tile wakes a neighbour only if one of its edge cells beats, by an open step, what the
 *                                neighbour's cell held when the job loaded its halo (astar_tile.hip, section 7)
 *   With both set and variant 161 this is the schedule of tsa_search_kernel (tests/test_dense_model.py runs it).
 * workload.bin: see sim_dense.py
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define INF 0x3fffffff
static int rows, cols, TI = 64, TJ = 16, tiles_i, tiles_j, bucket_w, variant, extra_h, redblack, filter;
static uint8_t* nbr;
static int32_t* g;
static int gi, gj;

static inline int octile(int i, int j) {
  int dx = abs(i - gi), dy = abs(j - gj);
  int mx = dx > dy ? dx : dy, mn = dx > dy ? dy : dx;
  return 1000 * mx + 414 * mn;
}
typedef struct { long jobs, noop, sweeps, row_evals, row_skips, rounds, buckets, tiles, cells_changed, maxjobs, hextra; } stats;
static uint8_t *act_cur, *act_far, *first_f, *touched;
static int best;
static long long bend;
static long long lim;   /* propagate iff f < lim */

static inline int gat(int i, int j) { return (i >= 0 && j >= 0 && i < rows && j < cols) ? g[(size_t)j * rows + i] : INF; }
static inline int prop_at(int v, int i, int j) {
  if (v >= INF) return INF;
  return ((long long)v + octile(i, j) < lim) ? v : INF;
}
static void activate(int ti, int tj) {
  if (ti < 0 || tj < 0 || ti >= tiles_i || tj >= tiles_j) return;
  act_cur[tj * tiles_i + ti] = 1;
}

typedef struct { int t; int32_t* val; } pending_write;

/* one job; writes its new tile values to out (TI*TJ) and returns 1 if anything changed */
static int job(int t, stats* st, int32_t* out, int* goal_best) {
  const int ti = t % tiles_i, tj = t / tiles_i;
  const int i0 = ti * TI, j0 = tj * TJ;
  static int32_t cur[64][64], old[64][64], pp[64][64];
  static uint8_t mk[64][64];
  const int first = first_f[t];
  first_f[t] = 0;
  touched[t] = 1;
  for (int b = 0; b < TJ; ++b)
    for (int a = 0; a < TI; ++a) {
      const int i = i0 + a, j = j0 + b;
      const int in = i < rows && j < cols;
      cur[b][a] = old[b][a] = in ? g[(size_t)j * rows + i] : INF;
      mk[b][a] = in ? nbr[(size_t)j * rows + i] : 0;
    }
  /* halo contributions, once (the halo does not change during the job) */
  unsigned hz = 0, fa = 0, fb = 0;   /* per-row flags: own row changed / source row above changed / below changed */
  for (int b = 0; b < TJ; ++b)
    for (int a = 0; a < TI; ++a) {
      if (a != 0 && a != TI - 1 && b != 0 && b != TJ - 1) continue;
      const uint8_t m = mk[b][a];
      if (!m) continue;
      int v = cur[b][a];
      static const int di[8] = {-1, 0, 1, -1, 1, -1, 0, 1}, dj[8] = {-1, -1, -1, 0, 0, 1, 1, 1};
      for (int k = 0; k < 8; ++k) {
        if (!((m >> k) & 1)) continue;
        const int na = a + di[k], nb = b + dj[k];
        if (na >= 0 && na < TI && nb >= 0 && nb < TJ) continue;   /* in-tile source: the sweeps */
        const int p = prop_at(gat(i0 + na, j0 + nb), i0 + na, j0 + nb);
        if (p >= INF) continue;
        const int c = p + ((di[k] && dj[k]) ? 1414 : 1000);
        if (c < v) v = c;
      }
      if (v < cur[b][a]) { cur[b][a] = v; hz |= 1u << b; if (b + 1 < TJ) fa |= 1u << (b + 1); if (b > 0) fb |= 1u << (b - 1); }
    }
  const unsigned all = TJ >= 32 ? 0xffffffffu : ((1u << TJ) - 1);
  if (first) { hz = fa = fb = all; }
  if (!(hz | fa | fb)) {
    st->noop++;
    for (int b = 0; b < TJ; ++b) for (int a = 0; a < TI; ++a) out[b * TI + a] = cur[b][a];
    return 0;
  }
  for (int b = 0; b < TJ; ++b)
    for (int a = 0; a < TI; ++a) pp[b][a] = prop_at(cur[b][a], i0 + a, j0 + b);
  int dir = 0;
  while (hz | fa | fb) {
    const unsigned need = dir == 0 ? (fa | hz) : (fb | hz);
    if (need) {
      st->sweeps++;
      for (int s = 0; s < TJ; ++s) {
        const int b = dir == 0 ? s : TJ - 1 - s;
        const unsigned bit = 1u << b;
        const unsigned mine = dir == 0 ? (fa | hz) : (fb | hz);
        if (!(mine & bit)) { st->row_skips++; continue; }
        st->row_evals++;
        if (dir == 0) fa &= ~bit; else fb &= ~bit;
        hz &= ~bit;
        const int bp = dir == 0 ? b - 1 : b + 1;
        const int kd0 = dir == 0 ? 0 : 5, kd2 = dir == 0 ? 2 : 7;
        int32_t tmp[64], ph[66];
        for (int a = 0; a < TI; ++a) {
          int v = cur[b][a];
          const uint8_t m = mk[b][a];
          if (m && bp >= 0 && bp < TJ) {
            { int c = pp[bp][a] + 1000; if (pp[bp][a] < INF && c < v) v = c; }   /* straight: target free is enough */
            if (a > 0 && ((m >> kd0) & 1) && pp[bp][a - 1] < INF) { int c = pp[bp][a - 1] + 1414; if (c < v) v = c; }
            if (a < TI - 1 && ((m >> kd2) & 1) && pp[bp][a + 1] < INF) { int c = pp[bp][a + 1] + 1414; if (c < v) v = c; }
          }
          tmp[a] = v;
        }
        for (int a = 0; a < TI; ++a) ph[a + 1] = variant == 0 ? prop_at(tmp[a], i0 + a, j0 + b) : pp[b][a];
        ph[0] = ph[TI + 1] = INF;
        int changed = 0;
        for (int a = 0; a < TI; ++a) {
          int v = tmp[a];
          if (mk[b][a]) {
            if (ph[a] < INF && ph[a] + 1000 < v) v = ph[a] + 1000;
            if (ph[a + 2] < INF && ph[a + 2] + 1000 < v) v = ph[a + 2] + 1000;
          }
          if (v < cur[b][a]) { cur[b][a] = v; changed = 1; }
          pp[b][a] = prop_at(cur[b][a], i0 + a, j0 + b);
        }
        /* extra horizontal-only passes over a row that just changed (fresh values) */
        for (int e = 0; e < extra_h && changed; ++e) {
          int ch2 = 0;
          int32_t nv[64];
          st->hextra++;
          for (int a = 0; a < TI; ++a) {
            int v = cur[b][a];
            if (mk[b][a]) {
              if (a > 0 && pp[b][a - 1] < INF && pp[b][a - 1] + 1000 < v) v = pp[b][a - 1] + 1000;
              if (a < TI - 1 && pp[b][a + 1] < INF && pp[b][a + 1] + 1000 < v) v = pp[b][a + 1] + 1000;
            }
            nv[a] = v;
          }
          for (int a = 0; a < TI; ++a) if (nv[a] < cur[b][a]) { cur[b][a] = nv[a]; ch2 = 1; pp[b][a] = prop_at(nv[a], i0 + a, j0 + b); }
          if (!ch2) break;
        }
        if (changed) { hz |= bit; if (b + 1 < TJ) fa |= bit << 1; if (b > 0) fb |= bit >> 1; }
      }
    }
    dir ^= 1;
  }
  /* results, activation */
  int far = 0, any = 0;
  static uint8_t chg[64][64];
  for (int b = 0; b < TJ; ++b)
    for (int a = 0; a < TI; ++a) {
      const int i = i0 + a, j = j0 + b;
      const int v = cur[b][a];
      out[b * TI + a] = v;
      chg[b][a] = 0;
      if (i >= rows || j >= cols || v >= INF) continue;
      const int ch = v < old[b][a];
      chg[b][a] = (uint8_t)ch;
      if (ch) { any = 1; st->cells_changed++; if (i == gi && j == gj && v < *goal_best) *goal_best = v; }
      const long long f = (long long)v + octile(i, j);
      if (f > best) continue;
      if (f >= bend) { far = 1; continue; }
      if (filter) continue;
      const int newly = first && f >= bend - bucket_w;
      if (!(ch || newly)) continue;
      const int ea = a == 0 ? -1 : (a == TI - 1 ? 1 : 0), eb = b == 0 ? -1 : (b == TJ - 1 ? 1 : 0);
      if (ea) activate(ti + ea, tj);
      if (eb) activate(ti, tj + eb);
      if (ea && eb) activate(ti + ea, tj + eb);
    }
  if (filter) {
    /* the kernel's wake tests: the halo as loaded is g itself (a round's results become visible together) */
    static const int di[8] = {-1, 0, 1, -1, 1, -1, 0, 1}, dj[8] = {-1, -1, -1, 0, 0, 1, 1, 1};
#define PASSES(b, a) (pp[b][a] < INF)
    /* edge rows: cells that changed and may pass on (all that may pass on in a first job); lane 0 / 63 leave their
       outward diagonal step to the column test */
    for (int side = 0; side < 2; ++side) {
      const int b = side ? TJ - 1 : 0;
      for (int a = 0; a < TI; ++a) {
        if (!PASSES(b, a) || !(chg[b][a] || first)) continue;
        for (int k = (side ? 5 : 0); k < (side ? 8 : 3); ++k) {
          if (!((mk[b][a] >> k) & 1)) continue;
          const int na = a + di[k];
          if (na < 0 || na >= TI) continue;
          if (pp[b][a] + ((di[k] && dj[k]) ? 1414 : 1000) < gat(i0 + na, j0 + b + dj[k])) activate(ti, tj + dj[k]);
        }
      }
    }
    /* edge columns and corners: run when a cell of lane 0 / 63 changed and may pass on (or a first job); every cell of
       the two columns that may pass on takes part */
    int trig = first;
    for (int b = 0; b < TJ && !trig; ++b)
      if ((chg[b][0] && PASSES(b, 0)) || (chg[b][TI - 1] && PASSES(b, TI - 1))) trig = 1;
    if (trig)
      for (int side = 0; side < 2; ++side) {
        const int a = side ? TI - 1 : 0, da = side ? 1 : -1;
        for (int b = 0; b < TJ; ++b) {
          if (!PASSES(b, a)) continue;
          for (int k = 0; k < 8; ++k) {
            if (di[k] != da || !((mk[b][a] >> k) & 1)) continue;
            const int nb = b + dj[k];
            if (pp[b][a] + (dj[k] ? 1414 : 1000) < gat(i0 + a + da, j0 + nb)) activate(ti + da, tj + (nb < 0 ? -1 : (nb >= TJ ? 1 : 0)));
          }
        }
      }
#undef PASSES
  }
  if (far) act_far[t] = 1;
  return any;
}

int main(int argc, char** argv) {
  if (argc < 3) { fprintf(stderr, "usage: sim_dense2 workload.bin bucket [nq] [variant] [TJ]\n"); return 2; }
  FILE* f = fopen(argv[1], "rb");
  bucket_w = atoi(argv[2]);
  int32_t hdr[3];
  if (!f || fread(hdr, 4, 3, f) != 3) return 1;
  rows = hdr[0]; cols = hdr[1];
  int nq = hdr[2];
  if (argc > 3 && atoi(argv[3]) < nq) nq = atoi(argv[3]);
  variant = argc > 4 ? atoi(argv[4]) % 10 : 0;
  extra_h = argc > 4 ? atoi(argv[4]) / 10 : 0;
  if (argc > 5) TJ = atoi(argv[5]);
  if (argc > 6) TI = atoi(argv[6]);
  redblack = getenv("SIM_REDBLACK") != NULL;
  filter = getenv("SIM_FILTER") != NULL;
  nbr = malloc((size_t)rows * cols);
  if (fread(nbr, 1, (size_t)rows * cols, f) != (size_t)rows * cols) return 1;
  int32_t* qs = malloc(sizeof(int32_t) * 4 * hdr[2]);
  if (fread(qs, 16, hdr[2], f) != (size_t)hdr[2]) return 1;
  fclose(f);
  tiles_i = (rows + TI - 1) / TI; tiles_j = (cols + TJ - 1) / TJ;
  const int ntile = tiles_i * tiles_j;
  g = malloc(sizeof(int32_t) * (size_t)rows * cols);
  act_cur = calloc(ntile, 1); act_far = calloc(ntile, 1); first_f = calloc(ntile, 1); touched = calloc(ntile, 1);
  int* list = malloc(sizeof(int) * ntile);
  int32_t* outbuf = NULL; size_t outcap = 0;
  stats tot = {0};
  long totE = 0, bad = 0;
  for (int q = 0; q < nq; ++q) {
    const int start = qs[4 * q], goal = qs[4 * q + 1], want_cost = qs[4 * q + 2], want_E = qs[4 * q + 3];
    for (size_t c = 0; c < (size_t)rows * cols; ++c) g[c] = INF;
    memset(act_cur, 0, ntile); memset(act_far, 0, ntile); memset(first_f, 0, ntile); memset(touched, 0, ntile);
    gi = goal % rows; gj = goal / rows;
    const int si = start % rows, sj = start / rows;
    g[start] = 0;
    best = INF;
    if (start == goal) best = 0;
    stats st = {0};
    long long bucket = octile(si, sj) / bucket_w;
    bend = (bucket + 1) * (long long)bucket_w;
    act_cur[(sj / TJ) * tiles_i + si / TI] = 1;
    first_f[(sj / TJ) * tiles_i + si / TI] = 1;
    st.buckets = 1;
    int phase = 0;
    for (;;) {
      int n = 0, other = 0;
      for (int t = 0; t < ntile; ++t)
        if (act_cur[t]) {
          if (redblack && (((t % tiles_i) + (t / tiles_i)) & 1) != phase) { other = 1; continue; }
          list[n++] = t; act_cur[t] = 0;
        }
      if (n == 0 && other) { phase ^= 1; continue; }
      if (n == 0) {
        if (best != INF && best < bend) break;
        int any = 0;
        for (int t = 0; t < ntile; ++t) { if (act_far[t]) { act_cur[t] = 1; first_f[t] = 1; any = 1; } act_far[t] = 0; }
        if (!any) break;
        bucket += 1; bend = (bucket + 1) * (long long)bucket_w;
        st.buckets++;
        continue;
      }
      st.rounds++;
      if (n > st.maxjobs) st.maxjobs = n;
      lim = bend < (long long)best + 1 ? bend : (long long)best + 1;
      if ((size_t)n * TI * TJ > outcap) { outcap = (size_t)n * TI * TJ; outbuf = realloc(outbuf, outcap * sizeof(int32_t)); }
      int gb = best;
      for (int k = 0; k < n; ++k) { job(list[k], &st, outbuf + (size_t)k * TI * TJ, &gb); st.jobs++; }
      for (int k = 0; k < n; ++k) {   /* the round's results become visible together */
        const int t = list[k], i0 = (t % tiles_i) * TI, j0 = (t / tiles_i) * TJ;
        for (int b = 0; b < TJ; ++b)
          for (int a = 0; a < TI; ++a)
            if (i0 + a < rows && j0 + b < cols) g[(size_t)(j0 + b) * rows + i0 + a] = outbuf[(size_t)k * TI * TJ + b * TI + a];
      }
      best = gb;
      phase ^= 1;
    }
    long E = 0;
    for (int t = 0; t < ntile; ++t) st.tiles += touched[t];
    for (int j = 0; j < cols; ++j)
      for (int i = 0; i < rows; ++i) {
        int v = g[(size_t)j * rows + i];
        if (v < INF && best != INF && v + octile(i, j) <= best) E++;
      }
    const int ok = (best == want_cost || (best == INF && want_cost >= 0x7fffffff)) && E == want_E;
    if (!ok) { bad++; fprintf(stderr, "query %d MISMATCH cost %d want %d E %ld want %d\n", q, best, want_cost, E, want_E); }
    totE += E;
    tot.jobs += st.jobs; tot.noop += st.noop; tot.sweeps += st.sweeps; tot.row_evals += st.row_evals; tot.row_skips += st.row_skips;
    tot.rounds += st.rounds; tot.buckets += st.buckets; tot.tiles += st.tiles; tot.cells_changed += st.cells_changed; tot.hextra += st.hextra;
  }
  const double valu = (tot.jobs - tot.noop) * 350.0 + tot.noop * 200.0 + tot.row_evals * 23.0;
  printf("nq %d tile %dx%d bucket %d variant %d: E %ld tiles %ld jobs %ld (%.2f/tile, %.1f%% no-op) sweeps %.2f/job row evals %.1f/job skipped %.1f/job rounds %ld buckets %ld writes/E %.2f mismatches %ld\n",
         nq, TI, TJ, bucket_w, variant, totE, tot.tiles, tot.jobs, (double)tot.jobs / tot.tiles, 100.0 * tot.noop / tot.jobs,
         (double)tot.sweeps / tot.jobs, (double)tot.row_evals / tot.jobs, (double)tot.row_skips / tot.jobs, tot.rounds, tot.buckets,
         (double)tot.cells_changed / totE, bad);
  printf("  extra horizontal passes per job %.1f\n", (double)tot.hextra / tot.jobs);
  printf("  model: VALU wave-instr per settled cell %.1f (jobs %.1f + rows %.1f); per query: %.0f jobs, %.0f rounds\n", valu / totE,
         ((tot.jobs - tot.noop) * 350.0 + tot.noop * 200.0) / totE, tot.row_evals * 23.0 / totE, (double)tot.jobs / nq, (double)tot.rounds / nq);
  return bad ? 1 : 0;
}
