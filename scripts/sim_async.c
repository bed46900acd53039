/* sim_async.c -- developer tool: discrete-event CPU model of the tile kernel with W wavefronts per query and a choice of
 * schedulers, built on the exact tile job of sim_dense2.c (64 x 16 tiles, halo applied once at load, alternating sweeps
 * with row flags, extra horizontal passes, masked wake tests).  A job reads g as it is when the job STARTS and its
 * results (the tile, the wake-ups) become visible when it ENDS, `cost` microseconds later; W wavefronts run jobs
 * concurrently.  Answers, before a kernel is written: how many jobs / row evaluations a scheduling policy costs and how
 * long the longest query takes (makespan), each checked for exactness (cost and E of the oracle).
 *
 *   sim_async <workload.bin> bucket nq policy W [subbins]
 *     policy 0: red-black rounds separated by barriers (the round-2 kernel); round time = list build + longest wave
 *            1: asynchronous FIFO inside an f-bucket: a woken tile is appended to the queue, a free wavefront takes the
 *               head; a tile that is running when it is woken runs again afterwards
 *            2: as 1, but a free wavefront takes the queued tile with the lowest key (key = lowest f of a cell the
 *               waker improved, quantised to `subbins` classes per bucket)
 *            3: as 2 and a tile does not start while one of its four edge neighbours runs (it stays queued)
 *            4: the protocol of tsa_search_kernel: every wake-up adds a (key, tile) entry unless the tile is running (then
 *               its own wavefront queues it again with the lowest key when it ends); a free wavefront takes the entry
 *               with the lowest key and drops it if the tile has nothing pending (its wake-ups were consumed already)
 *   round 5: SIM_FIRSTROWS (a tile's first job in a bucket evaluates only the rows that hold a cell the new bound releases --
 *   the rule the kernel's RNA_TSA_FIRST_ROWS implements), SIM_FIRSTWAKE (... and its wake tests take released cells only: no
 *   difference), SIM_AGAINKEY (a tile woken while it ran is queued with its waker's key instead of the lowest); the last lines
 *   count the re-runs of tiles woken while they ran and how often a wavefront's next tile is the one it just finished -- the
 *   figures behind the kernel's sticky tiles.
 *   round 6: SIM_GKEY (keys are g instead of f = g + h: Dijkstra order inside a bucket), SIM_HW16=n (key = g + n/16 h), and a line
 *   with the working jobs (those that get past the halo step), the rows they write and the cells reached over the cells settled --
 *   the runs behind profiles/r06_sim_schedules.txt.
 *   job cost model (microseconds, from the RNA_TSA_STATS phase timers under load): load+halo 3.8, rows 0.2 each,
 *   horizontal passes 0.065 each, results 2.6; a job that finds nothing in its halo 3.8.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define INF 0x3fffffff
static int rows, cols, TI = 64, TJ = 16, tiles_i, tiles_j, bucket_w, extra_h = 16;
static int gkey = 0, hw16 = 16;
static uint8_t* nbr;
static int32_t* g;
static int gi, gj;
static int excl4 = 0, excl8 = 0, keymode = 1, kshift = -1, nodec = 0, dirtykey = 0, maxlive = 0;
static long long* popkey;
static double C_LOAD = 3.8, C_ROW = 0.2, C_HP = 0.065, C_RES = 2.6, C_ROUND = 2.4, C_POP = 0.3;

static inline int octile(int i, int j) {
  int dx = abs(i - gi), dy = abs(j - gj);
  int mx = dx > dy ? dx : dy, mn = dx > dy ? dy : dx;
  return 1000 * mx + 414 * mn;
}
typedef struct { long jobs, noop, row_evals, rounds, buckets, hextra; double makespan, busy; } stats;
static uint8_t *first_f;
static int best;
static long long bend, lim;

static inline int gat(int i, int j) { return (i >= 0 && j >= 0 && i < rows && j < cols) ? g[(size_t)j * rows + i] : INF; }
static inline int prop_at(int v, int i, int j) {
  if (v >= INF) return INF;
  return ((long long)v + octile(i, j) < lim) ? v : INF;
}

typedef struct {
  int t;
  int32_t val[64 * 16];
  int wake[8];          /* neighbour tile or -1 */
  long long wkey[8];    /* lowest f the waker offers that neighbour */
  int far, any, goal_best, rowsn, hp, noop;
  double cost;
  int32_t ppv[64 * 16];     /* what each cell may pass on (INF: nothing) */
  uint8_t chgv[64 * 16];    /* the cell changed in this job */
  int first;
} jobres;

/* one job on the CURRENT g; nothing is written: the caller applies `r` when the job ends */
static void job(int t, int first, jobres* r) {
  const int ti = t % tiles_i, tj = t / tiles_i;
  const int i0 = ti * TI, j0 = tj * TJ;
  static int32_t cur[16][64], old[16][64], pp[16][64];
  static uint8_t mk[16][64];
  static const int di[8] = {-1, 0, 1, -1, 1, -1, 0, 1}, dj[8] = {-1, -1, -1, 0, 0, 1, 1, 1};
  r->t = t; r->far = 0; r->any = 0; r->goal_best = INF; r->rowsn = 0; r->hp = 0; r->noop = 0;
  for (int k = 0; k < 8; ++k) { r->wake[k] = -1; r->wkey[k] = (long long)1 << 60; }
  for (int b = 0; b < TJ; ++b)
    for (int a = 0; a < TI; ++a) {
      const int i = i0 + a, j = j0 + b;
      const int in = i < rows && j < cols;
      cur[b][a] = old[b][a] = in ? g[(size_t)j * rows + i] : INF;
      mk[b][a] = in ? nbr[(size_t)j * rows + i] : 0;
    }
  unsigned hz = 0, fa = 0, fb = 0;
  for (int b = 0; b < TJ; ++b)
    for (int a = 0; a < TI; ++a) {
      if (a != 0 && a != TI - 1 && b != 0 && b != TJ - 1) continue;
      const uint8_t m = mk[b][a];
      if (!m) continue;
      int v = cur[b][a];
      for (int k = 0; k < 8; ++k) {
        if (!((m >> k) & 1)) continue;
        const int na = a + di[k], nb = b + dj[k];
        if (na >= 0 && na < TI && nb >= 0 && nb < TJ) continue;
        const int p = prop_at(gat(i0 + na, j0 + nb), i0 + na, j0 + nb);
        if (p >= INF) continue;
        const int c = p + ((di[k] && dj[k]) ? 1414 : 1000);
        if (c < v) v = c;
      }
      if (v < cur[b][a]) { cur[b][a] = v; hz |= 1u << b; if (b + 1 < TJ) fa |= 1u << (b + 1); if (b > 0) fb |= 1u << (b - 1); }
    }
  const unsigned all = (1u << TJ) - 1; int first_noskip = 0;
  if (first && getenv("SIM_FIRSTROWS")) {
    /* only the rows that hold a cell the new bound releases: it passes on now and f >= the bucket's start */
    for (int b = 0; b < TJ; ++b)
      for (int a = 0; a < TI; ++a) {
        const int i = i0 + a, j = j0 + b;
        if (i >= rows || j >= cols) continue;
        const int v = cur[b][a];
        if (v >= INF) continue;
        const long long f = (long long)v + octile(i, j);
        if (f < lim && f >= bend - bucket_w) { hz |= 1u << b; if (b + 1 < TJ) fa |= 1u << (b + 1); if (b > 0) fb |= 1u << (b - 1); }
      }
    first_noskip = 1;
  } else if (first) { hz = fa = fb = all; }
  if (!(hz | fa | fb) && !first) {
    r->noop = 1; r->cost = C_LOAD;
    for (int b = 0; b < TJ; ++b) for (int a = 0; a < TI; ++a) r->val[b * TI + a] = cur[b][a];
    return;
  }
  for (int b = 0; b < TJ; ++b)
    for (int a = 0; a < TI; ++a) pp[b][a] = prop_at(cur[b][a], i0 + a, j0 + b);
  int dir = 0;
  while (hz | fa | fb) {
    const unsigned need = dir == 0 ? (fa | hz) : (fb | hz);
    if (need) {
      for (int s = 0; s < TJ; ++s) {
        const int b = dir == 0 ? s : TJ - 1 - s;
        const unsigned bit = 1u << b;
        const unsigned mine = dir == 0 ? (fa | hz) : (fb | hz);
        if (!(mine & bit)) continue;
        r->rowsn++;
        if (dir == 0) fa &= ~bit; else fb &= ~bit;
        hz &= ~bit;
        const int bp = dir == 0 ? b - 1 : b + 1;
        const int kd0 = dir == 0 ? 0 : 5, kd2 = dir == 0 ? 2 : 7;
        int32_t tmp[64], ph[66];
        for (int a = 0; a < TI; ++a) {
          int v = cur[b][a];
          const uint8_t m = mk[b][a];
          if (m && bp >= 0 && bp < TJ) {
            { int c = pp[bp][a] + 1000; if (pp[bp][a] < INF && c < v) v = c; }
            if (a > 0 && ((m >> kd0) & 1) && pp[bp][a - 1] < INF) { int c = pp[bp][a - 1] + 1414; if (c < v) v = c; }
            if (a < TI - 1 && ((m >> kd2) & 1) && pp[bp][a + 1] < INF) { int c = pp[bp][a + 1] + 1414; if (c < v) v = c; }
          }
          tmp[a] = v;
        }
        for (int a = 0; a < TI; ++a) ph[a + 1] = pp[b][a];
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
        for (int e = 0; e < extra_h && changed; ++e) {
          int ch2 = 0;
          int32_t nv[64];
          r->hp++;
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
  static uint8_t chg[16][64]; static uint8_t rel[16][64]; const int firstw = getenv("SIM_FIRSTWAKE") != NULL;
  r->first = first;
  for (int b = 0; b < TJ; ++b)
    for (int a = 0; a < TI; ++a) { r->ppv[b * TI + a] = pp[b][a]; r->chgv[b * TI + a] = 0; }
  for (int b = 0; b < TJ; ++b)
    for (int a = 0; a < TI; ++a) {
      const int i = i0 + a, j = j0 + b;
      const int v = cur[b][a];
      r->val[b * TI + a] = v;
      chg[b][a] = 0; rel[b][a] = 0;
      if (i >= rows || j >= cols || v >= INF) continue;
      { const long long f_ = (long long)v + octile(i, j); rel[b][a] = (uint8_t)(first && f_ < lim && f_ >= bend - bucket_w); }
      const int ch = v < old[b][a];
      chg[b][a] = (uint8_t)ch;
      r->chgv[b * TI + a] = (uint8_t)ch;
      if (ch) { r->any = 1; if (i == gi && j == gj && v < r->goal_best) r->goal_best = v; }
      const long long f = (long long)v + octile(i, j);
      if (f > best) continue;
      if (f >= bend) { r->far = 1; continue; }
    }
#define PASSES(b, a) (pp[b][a] < INF)
#define WAKE(nt_i, nt_j, f_)                                                                      \
  do {                                                                                            \
    const int wi_ = (nt_i), wj_ = (nt_j);                                                         \
    if (wi_ >= 0 && wj_ >= 0 && wi_ < tiles_i && wj_ < tiles_j) {                                 \
      const int kk_ = (wj_ - tj + 1) * 3 + (wi_ - ti + 1);                                        \
      const int k8_ = kk_ < 4 ? kk_ : kk_ - 1;                                                    \
      r->wake[k8_] = wj_ * tiles_i + wi_;                                                         \
      if ((f_) < r->wkey[k8_]) r->wkey[k8_] = (f_);                                               \
    }                                                                                             \
  } while (0)
  for (int side = 0; side < 2; ++side) {
    const int b = side ? TJ - 1 : 0;
    for (int a = 0; a < TI; ++a) {
      if (!PASSES(b, a) || !(chg[b][a] || (firstw ? rel[b][a] : first))) continue;
      for (int k = (side ? 5 : 0); k < (side ? 8 : 3); ++k) {
        if (!((mk[b][a] >> k) & 1)) continue;
        const int na = a + di[k];
        if (na < 0 || na >= TI) continue;
        const int c = pp[b][a] + ((di[k] && dj[k]) ? 1414 : 1000);
        if (c < gat(i0 + na, j0 + b + dj[k])) WAKE(ti, tj + dj[k], (long long)c + (gkey ? 0 : (long long)octile(i0 + na, j0 + b + dj[k]) * hw16 / 16));
      }
    }
  }
  int trig = firstw ? 0 : first;
  for (int b = 0; b < TJ && !trig; ++b)
    if (((chg[b][0] || rel[b][0]) && PASSES(b, 0)) || ((chg[b][TI - 1] || rel[b][TI - 1]) && PASSES(b, TI - 1))) trig = 1;
  if (trig)
    for (int side = 0; side < 2; ++side) {
      const int a = side ? TI - 1 : 0, da = side ? 1 : -1;
      for (int b = 0; b < TJ; ++b) {
        if (!PASSES(b, a)) continue;
        for (int k = 0; k < 8; ++k) {
          if (di[k] != da || !((mk[b][a] >> k) & 1)) continue;
          const int nb = b + dj[k];
          const int c = pp[b][a] + (dj[k] ? 1414 : 1000);
          if (c < gat(i0 + a + da, j0 + nb)) WAKE(ti + da, tj + (nb < 0 ? -1 : (nb >= TJ ? 1 : 0)), (long long)c + (gkey ? 0 : (long long)octile(i0 + a + da, j0 + nb) * hw16 / 16));
        }
      }
    }
  if (keymode == 2) {
    long long m = (long long)1 << 60;
    for (int k = 0; k < 8; ++k) if (r->wake[k] >= 0 && r->wkey[k] < m) m = r->wkey[k];
    for (int k = 0; k < 8; ++k) r->wkey[k] = m;
  }
  r->cost = C_LOAD + C_ROW * r->rowsn + C_HP * r->hp + C_RES;
}

/* SIM_FRESH: the wake tests of a job evaluated when it ENDS against what the neighbours hold THEN (the kernel would
   re-read their edge rows / columns after its stores), instead of against the halo it loaded when it started */
static long n_work = 0, n_rows_written = 0, n_reached = 0; static int fresh = 0; static long n_again=0, n_again_noop=0, n_sticky=0, n_nbr=0; static int last_t[64]; static uint8_t *isagain;
static void wake_fresh(jobres* r) {
  static const int di[8] = {-1, 0, 1, -1, 1, -1, 0, 1}, dj[8] = {-1, -1, -1, 0, 0, 1, 1, 1};
  const int t = r->t, ti = t % tiles_i, tj = t / tiles_i, i0 = ti * TI, j0 = tj * TJ;
  for (int k = 0; k < 8; ++k) { r->wake[k] = -1; r->wkey[k] = (long long)1 << 60; }
  if (r->noop) return;
  for (int b = 0; b < TJ; ++b)
    for (int a = 0; a < TI; ++a) {
      if (a != 0 && a != TI - 1 && b != 0 && b != TJ - 1) continue;
      const int i = i0 + a, j = j0 + b;
      if (i >= rows || j >= cols) continue;
      const int p = r->ppv[b * TI + a];
      if (p >= INF) continue;
      const uint8_t m = nbr[(size_t)j * rows + i];
      for (int k = 0; k < 8; ++k) {
        if (!((m >> k) & 1)) continue;
        const int na = a + di[k], nb = b + dj[k];
        if (na >= 0 && na < TI && nb >= 0 && nb < TJ) continue;
        const int c = p + ((di[k] && dj[k]) ? 1414 : 1000);
        if (c < gat(i0 + na, j0 + nb)) {
          const int wi = ti + (na < 0 ? -1 : (na >= TI ? 1 : 0)), wj = tj + (nb < 0 ? -1 : (nb >= TJ ? 1 : 0));
          if (wi < 0 || wj < 0 || wi >= tiles_i || wj >= tiles_j) continue;
          const int kk = (wj - tj + 1) * 3 + (wi - ti + 1), k8 = kk < 4 ? kk : kk - 1;
          const long long f = (long long)c + (gkey ? 0 : octile(i0 + na, j0 + nb));
          r->wake[k8] = wj * tiles_i + wi;
          if (f < r->wkey[k8]) r->wkey[k8] = f;
        }
      }
    }
}

static void apply(const jobres* r) {
  const int t = r->t, i0 = (t % tiles_i) * TI, j0 = (t / tiles_i) * TJ;
  if (!r->noop) { n_work++; for (int b = 0; b < TJ; ++b) { int c = 0; for (int a = 0; a < TI; ++a) c |= r->chgv[b * TI + a]; n_rows_written += c; } }
  for (int b = 0; b < TJ; ++b)
    for (int a = 0; a < TI; ++a)
      if (i0 + a < rows && j0 + b < cols) g[(size_t)(j0 + b) * rows + i0 + a] = r->val[b * TI + a];
  if (r->goal_best < best) best = r->goal_best;
}

/* ---- scheduler state ---- */
static uint8_t *queued, *running, *dirty, *farflag;
static long long* qkey;   /* key of a queued tile */
static int* qlist; static int qn;   /* queued tiles in arrival order (holes = -1) */
static long long* dkey;

static void push(int t, long long key) {
  if (queued[t]) { if (key < qkey[t] && !nodec) qkey[t] = key; return; }
  queued[t] = 1; qkey[t] = key; qlist[qn++] = t;
  { int live = 0; for (int k = 0; k < qn; ++k) live += qlist[k] >= 0; if (live > maxlive) maxlive = live; }
}
static int nb_running(int t) {
  const int ti = t % tiles_i, tj = t / tiles_i;
  if (ti > 0 && running[t - 1]) return 1;
  if (ti + 1 < tiles_i && running[t + 1]) return 1;
  if (tj > 0 && running[t - tiles_i]) return 1;
  if (tj + 1 < tiles_j && running[t + tiles_i]) return 1;
  if (excl8) {
    if (ti > 0 && tj > 0 && running[t - 1 - tiles_i]) return 1;
    if (ti + 1 < tiles_i && tj > 0 && running[t + 1 - tiles_i]) return 1;
    if (ti > 0 && tj + 1 < tiles_j && running[t - 1 + tiles_i]) return 1;
    if (ti + 1 < tiles_i && tj + 1 < tiles_j && running[t + 1 + tiles_i]) return 1;
  }
  return 0;
}
/* policy 4: the queue is a multiset of (key, tile) entries -- a wake-up always adds one; a tile has a "pending" bit D
   and a "running" bit R.  An entry whose tile has nothing pending when it is taken is stale and dropped. */
typedef struct { long long key; int t; long seq; } entry;
static int clsn[4096], clsmax = 0, ringcap = 0, ringnc = 32; static long ringspill = 0;
static int ringmode = 0, nodup = 0; static long eseq = 0; static long long* qcls;
static entry* ents; static int nent = 0, maxent = 0; static long stale_pops = 0, pushes = 0;
static uint8_t *Dbit;
static void epush(int t, long long key) {
  long long k = key;
  if (kshift >= 0) { const long long lo = gkey ? 0 : bend - bucket_w; k = key < lo ? 0 : (key - lo) >> kshift; }
  if (ringcap > 0) { while (k < ringnc - 1 && clsn[k] >= ringcap) k++; if (k > ringnc - 1) k = ringnc - 1; if (clsn[k] >= ringcap) ringspill++; }
  ents[nent].key = k; ents[nent].t = t; ents[nent].seq = eseq++; nent++; pushes++;
  if (k >= 0 && k < 4096) { if (++clsn[k] > clsmax) clsmax = clsn[k]; }
  if (nent > maxent) maxent = nent;
}
static long long cls_of(long long key) { if (kshift < 0) return key; const long long lo = gkey ? 0 : bend - bucket_w; return key < lo ? 0 : (key - lo) >> kshift; }
static int againkey=0; static long long *akey;
static void wake4(int t, long long key) {
  if (running[t]) { if (!Dbit[t] || key < akey[t]) akey[t] = key; }
  if (nodup && !running[t]) {
    /* nodup 1: a pending tile keeps its first entry; 2: a duplicate only when the key class gets lower */
    if (Dbit[t]) { if (nodup == 2 && cls_of(key) < qcls[t]) { qcls[t] = cls_of(key); epush(t, key); } return; }
    Dbit[t] = 1; qcls[t] = cls_of(key); epush(t, key); return;
  }
  Dbit[t] = 1;
  if (!running[t]) epush(t, key);   /* a running tile is queued again (lowest key) by the wavefront that runs it */
}
/* returns a tile to run, -1 if the queue is empty; *spent = stale entries dropped on the way */
static int pop4(int* spent) {
  *spent = 0;
  for (;;) {
    if (nent == 0) return -1;
    int bi = -1;
    for (int k = 0; k < nent; ++k) {
      if (excl4 && nb_running(ents[k].t)) continue;
      if (bi < 0 || ents[k].key < ents[bi].key || (ents[k].key == ents[bi].key && (ringmode == 1 ? ents[k].seq < ents[bi].seq : (ringmode == 2 ? ents[k].seq > ents[bi].seq : ents[k].t < ents[bi].t)))) bi = k;
    }
    if (bi < 0) return -1;   /* everything queued sits next to a running tile */
    const int t = ents[bi].t;
    if (ents[bi].key >= 0 && ents[bi].key < 4096) clsn[ents[bi].key]--;
    ents[bi] = ents[--nent];
    if (running[t]) { (*spent)++; stale_pops++; continue; }   /* its runner sees D when it ends */
    if (!Dbit[t]) { (*spent)++; stale_pops++; continue; }
    Dbit[t] = 0;
    return t;
  }
}
/* take the next tile for a free wavefront; -1 if none can start now */
static int pop(int policy, int subbins) {
  int bi = -1;
  long long bk = 0;
  for (int k = 0; k < qn; ++k) {
    const int t = qlist[k];
    if (t < 0) continue;
    if (policy == 3 && nb_running(t)) continue;
    if (policy == 1) { bi = k; break; }
    long long key = qkey[t];
    if (kshift >= 0) {
      const long long lo = bend - bucket_w;
      key = key < lo ? 0 : (key - lo) >> kshift;
    } else if (subbins > 0) {   /* quantised: classes of bucket_w / subbins, ties in arrival order */
      const long long lo = bend - bucket_w;
      long long c = key < lo ? 0 : (key - lo) * subbins / bucket_w;
      if (c >= subbins) c = subbins - 1;
      key = c;
    }
    if (bi < 0 || key < bk) { bi = k; bk = key; }
  }
  if (bi < 0) return -1;
  const int t = qlist[bi];
  qlist[bi] = -1;
  queued[t] = 0;
  popkey[t] = qkey[t];
  /* compact now and then */
  if (qn > 4096) { int m = 0; for (int k = 0; k < qn; ++k) if (qlist[k] >= 0) qlist[m++] = qlist[k]; qn = m; }
  return t;
}

int main(int argc, char** argv) {
  if (argc < 6) { fprintf(stderr, "usage: sim_async workload.bin bucket nq policy W [subbins]\n"); return 2; }
  FILE* f = fopen(argv[1], "rb");
  bucket_w = atoi(argv[2]);
  int32_t hdr[3];
  if (!f || fread(hdr, 4, 3, f) != 3) return 1;
  rows = hdr[0]; cols = hdr[1];
  int nq = hdr[2];
  if (atoi(argv[3]) < nq) nq = atoi(argv[3]);
  const int policy = atoi(argv[4]), W = atoi(argv[5]);
  const int subbins = argc > 6 ? atoi(argv[6]) : 0;
  if (getenv("SIM_HPASS")) extra_h = atoi(getenv("SIM_HPASS"));
  if (getenv("SIM_EXCL8")) excl8 = 1;
  if (getenv("SIM_RINGCAP")) ringcap = atoi(getenv("SIM_RINGCAP"));   /* entries a class can hold: a push to a full class goes to the next one */
  if (getenv("SIM_RINGNC")) ringnc = atoi(getenv("SIM_RINGNC"));
  if (getenv("SIM_RING")) ringmode = atoi(getenv("SIM_RING"));   /* 2: last in, first out */     /* policy 4: entries of one key class leave in the order they came (a ring per class) */
  if (getenv("SIM_NODUP")) nodup = atoi(getenv("SIM_NODUP"));
  if (getenv("SIM_FRESH")) fresh = 1; if (getenv("SIM_GKEY")) gkey = 1; if (getenv("SIM_HW16")) hw16 = atoi(getenv("SIM_HW16"));
  if (getenv("SIM_EXCL")) excl4 = 1;   /* policy 4: a tile does not start while one of its four edge neighbours runs */
  if (getenv("SIM_NODEC")) nodec = 1;                            /* a queued tile keeps the key of its first wake-up */
  if (getenv("SIM_DIRTYKEY")) dirtykey = atoi(getenv("SIM_DIRTYKEY"));   /* key of a tile woken while it ran: 0 the waker's, 1 the key it was taken with, 2 lowest, 3 highest */
  if (getenv("SIM_KEY")) keymode = atoi(getenv("SIM_KEY"));     /* 1: lowest f offered to that neighbour; 2: lowest f offered to any neighbour by this job */
  if (getenv("SIM_KSHIFT")) kshift = atoi(getenv("SIM_KSHIFT")); /* keys compared as (f - bucket start) >> shift */
  nbr = malloc((size_t)rows * cols);
  if (fread(nbr, 1, (size_t)rows * cols, f) != (size_t)rows * cols) return 1;
  int32_t* qs = malloc(sizeof(int32_t) * 4 * hdr[2]);
  if (fread(qs, 16, hdr[2], f) != (size_t)hdr[2]) return 1;
  fclose(f);
  tiles_i = (rows + TI - 1) / TI; tiles_j = (cols + TJ - 1) / TJ;
  const int ntile = tiles_i * tiles_j;
  g = malloc(sizeof(int32_t) * (size_t)rows * cols);
  first_f = calloc(ntile, 1); queued = calloc(ntile, 1); running = calloc(ntile, 1); dirty = calloc(ntile, 1); farflag = calloc(ntile, 1);
  popkey = calloc(ntile, sizeof(long long));
  ents = malloc(sizeof(entry) * (1 << 22)); isagain=calloc(ntile,1); akey=calloc(ntile,sizeof(long long)); if(getenv("SIM_AGAINKEY")) againkey=1; for(int w=0;w<64;++w) last_t[w]=-1; Dbit = calloc(ntile, 1); qcls = calloc(ntile, sizeof(long long));
  qkey = malloc(sizeof(long long) * ntile); dkey = malloc(sizeof(long long) * ntile);
  qlist = malloc(sizeof(int) * (ntile + 8192));
  jobres* slot = malloc(sizeof(jobres) * (W > 4096 ? W : 4096));
  double* tend = malloc(sizeof(double) * W);
  int* busy = malloc(sizeof(int) * W);
  stats tot = {0};
  long totE = 0, bad = 0;
  double max_makespan = 0, sum_makespan = 0;
  for (int q = 0; q < nq; ++q) {
    const int start = qs[4 * q], goal = qs[4 * q + 1], want_cost = qs[4 * q + 2], want_E = qs[4 * q + 3];
    for (size_t c = 0; c < (size_t)rows * cols; ++c) g[c] = INF;
    memset(first_f, 0, ntile); memset(queued, 0, ntile); memset(running, 0, ntile); memset(dirty, 0, ntile); memset(farflag, 0, ntile);
    qn = 0;
    gi = goal % rows; gj = goal / rows;
    const int si = start % rows, sj = start / rows;
    g[start] = 0;
    best = INF;
    if (start == goal) best = 0;
    stats st = {0};
    long long bucket = octile(si, sj) / bucket_w;
    bend = (bucket + 1) * (long long)bucket_w;
    const int ts = (sj / TJ) * tiles_i + si / TI;
    first_f[ts] = 1;
    memset(Dbit, 0, ntile); nent = 0; memset(clsn, 0, sizeof(clsn));
    if (policy == 4) wake4(ts, 0); else push(ts, 0);
    st.buckets = 1;
    double now = 0;
    if (policy == 0) {
      /* red-black rounds with barriers */
      int phase = 0;
      for (;;) {
        int n = 0, other = 0;
        static int list[1 << 16];
        for (int k = 0; k < qn; ++k) {
          const int t = qlist[k];
          if (t < 0) continue;
          if ((((t % tiles_i) + (t / tiles_i)) & 1) != phase) { other = 1; continue; }
          list[n++] = t; qlist[k] = -1; queued[t] = 0;
        }
        { int m = 0; for (int k = 0; k < qn; ++k) if (qlist[k] >= 0) qlist[m++] = qlist[k]; qn = m; }
        if (n == 0 && other) { phase ^= 1; now += 0.5; continue; }
        if (n == 0) {
          if (best != INF && best < bend) break;
          int any = 0;
          for (int t = 0; t < ntile; ++t) if (farflag[t]) { farflag[t] = 0; first_f[t] = 1; push(t, 0); any = 1; }
          if (!any) break;
          bucket += 1; bend = (bucket + 1) * (long long)bucket_w;
          st.buckets++;
          now += C_ROUND;
          continue;
        }
        st.rounds++;
        lim = bend < (long long)best + 1 ? bend : (long long)best + 1;
        if (n > 4096) { fprintf(stderr, "round too large\n"); return 1; }
        for (int w = 0; w < W; ++w) tend[w] = 0;
        for (int k = 0; k < n; ++k) {
          const int fst = first_f[list[k]]; first_f[list[k]] = 0;
          job(list[k], fst, &slot[k]);
          st.jobs++; st.noop += slot[k].noop; st.row_evals += slot[k].rowsn; st.hextra += slot[k].hp;
          int wmin = 0;
          for (int w = 1; w < W; ++w) if (tend[w] < tend[wmin]) wmin = w;
          tend[wmin] += slot[k].cost;
          st.busy += slot[k].cost;
        }
        double rmax = 0;
        for (int w = 0; w < W; ++w) if (tend[w] > rmax) rmax = tend[w];
        now += rmax + C_ROUND;
        for (int k = 0; k < n; ++k) {
          apply(&slot[k]);
          if (slot[k].far) farflag[slot[k].t] = 1;
          for (int d = 0; d < 8; ++d) if (slot[k].wake[d] >= 0) push(slot[k].wake[d], slot[k].wkey[d]);
        }
        phase ^= 1;
      }
    } else {
      /* asynchronous: event loop over W wavefronts */
      for (int w = 0; w < W; ++w) busy[w] = 0;
      int nrun = 0;
      for (;;) {
        /* start jobs on free wavefronts */
        int started = 1;
        while (started) {
          started = 0;
          for (int w = 0; w < W && (qn > 0 || nent > 0); ++w) {
            if (busy[w]) continue;
            int spent = 0;
            const int t = policy == 4 ? pop4(&spent) : pop(policy, subbins);
            now += 0;   /* (stale entries cost the popping wavefront C_POP each, charged to its job below) */
            if (t < 0) break;
            lim = bend < (long long)best + 1 ? bend : (long long)best + 1;
            const int fst = first_f[t]; first_f[t] = 0;
            job(t, fst, &slot[w]);
            if (isagain[t]) { isagain[t]=0; n_again_noop += slot[w].noop; }
            if (last_t[w]==t) n_sticky++; else if (last_t[w]>=0) { int d=abs(last_t[w]%tiles_i - t%tiles_i), e=abs(last_t[w]/tiles_i - t/tiles_i); if (d<=1&&e<=1) n_nbr++; } last_t[w]=t;
            running[t] = 1; dirty[t] = 0;
            busy[w] = 1; tend[w] = now + C_POP * (1 + spent) + slot[w].cost;
            st.jobs++; st.noop += slot[w].noop; st.row_evals += slot[w].rowsn; st.hextra += slot[w].hp;
            st.busy += slot[w].cost;
            nrun++; started = 1;
          }
        }
        if (nrun == 0) {
          int left = 0;
          for (int k = 0; k < qn; ++k) if (qlist[k] >= 0) left = 1;
          if (left || nent) { fprintf(stderr, "stuck\n"); return 1; }
          /* bucket at its fixed point */
          if (best != INF && best < bend) break;
          int any = 0;
          qn = 0;
          for (int t = 0; t < ntile; ++t) if (farflag[t]) { farflag[t] = 0; first_f[t] = 1; if (policy == 4) wake4(t, 0); else push(t, 0); any = 1; }
          if (!any) break;
          bucket += 1; bend = (bucket + 1) * (long long)bucket_w;
          st.buckets++;
          now += C_ROUND;
          continue;
        }
        /* next completion */
        int wd = -1;
        for (int w = 0; w < W; ++w) if (busy[w] && (wd < 0 || tend[w] < tend[wd])) wd = w;
        now = tend[wd];
        busy[wd] = 0; nrun--;
        jobres* r = &slot[wd];
        apply(r);
        if (fresh) wake_fresh(r);
        running[r->t] = 0;
        if (r->far) farflag[r->t] = 1;
        if (policy == 4) {
          if (Dbit[r->t]) { epush(r->t, againkey ? akey[r->t] : 0); n_again++; isagain[r->t]=1; }
          for (int d = 0; d < 8; ++d) if (r->wake[d] >= 0) wake4(r->wake[d], r->wkey[d]);
          continue;
        }
        if (dirty[r->t]) {
          dirty[r->t] = 0;
          push(r->t, dirtykey == 0 ? dkey[r->t] : (dirtykey == 1 ? popkey[r->t] : (dirtykey == 2 ? 0 : (long long)1 << 40)));
        }
        for (int d = 0; d < 8; ++d) {
          const int nt = r->wake[d];
          if (nt < 0) continue;
          if (running[nt]) { if (!dirty[nt] || r->wkey[d] < dkey[nt]) dkey[nt] = r->wkey[d]; dirty[nt] = 1; }
          else push(nt, r->wkey[d]);
        }
      }
    }
    long E = 0;
    for (int j = 0; j < cols; ++j)
      for (int i = 0; i < rows; ++i) {
        int v = g[(size_t)j * rows + i];
        if (v < INF && best != INF && v + octile(i, j) <= best) E++;
        if (v < INF) n_reached++;
      }
    const int ok = (best == want_cost || (best == INF && want_cost >= 0x7fffffff)) && E == want_E;
    if (!ok) { bad++; fprintf(stderr, "query %d MISMATCH cost %d want %d E %ld want %d\n", q, best, want_cost, E, want_E); }
    totE += E;
    tot.jobs += st.jobs; tot.noop += st.noop; tot.row_evals += st.row_evals; tot.rounds += st.rounds; tot.buckets += st.buckets;
    tot.hextra += st.hextra; tot.busy += st.busy;
    sum_makespan += now;
    if (now > max_makespan) max_makespan = now;
    if (getenv("SIM_VERBOSE")) printf("  q%d E %ld jobs %ld makespan %.2f ms busy %.1f wave-ms eff %.0f%%\n", q, E, st.jobs, now * 1e-3, st.busy * 1e-3, 100.0 * st.busy / (now * W));
  }
  printf("policy %d W %d bucket %d subbins %d: E %ld jobs/query %.0f (%.1f%% no-op) rows/job %.1f hpass/job %.1f | busy %.2f wave-ms/query, makespan mean %.2f ms max %.2f ms, wave efficiency %.0f%%, mismatches %ld\n",
         policy, W, bucket_w, subbins, totE, (double)tot.jobs / nq, 100.0 * tot.noop / tot.jobs, (double)tot.row_evals / tot.jobs, (double)tot.hextra / tot.jobs,
         tot.busy * 1e-3 / nq, sum_makespan * 1e-3 / nq, max_makespan * 1e-3, 100.0 * tot.busy / (sum_makespan * W), bad);
  printf("  working jobs/query %.0f, rows written per working job %.1f, cells reached / E %.3f\n", (double)n_work / nq, (double)n_rows_written / (n_work ? n_work : 1), (double)n_reached / totE);
  printf("  most entries queued at once: %d (most in one key class: %d)\n", policy == 4 ? maxent : maxlive, clsmax);
  if (ringcap) printf("  pushes that found every class from theirs on full: %ld\n", ringspill);
  printf("  again re-queues %ld (%.3f of jobs), of which no-op %.3f; next job same tile %.3f of jobs, a neighbour tile %.3f\n", n_again, (double)n_again/tot.jobs, (double)n_again_noop/(n_again?n_again:1), (double)n_sticky/tot.jobs, (double)n_nbr/tot.jobs);
  if (policy == 4) printf("  entries pushed per job %.2f, stale entries dropped per job %.2f\n", (double)pushes / tot.jobs, (double)stale_pops / tot.jobs);
  return bad ? 1 : 0;
}
