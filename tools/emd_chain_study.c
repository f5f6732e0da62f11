/* Counting study for "run the eviction chains ahead and reconcile" (round 6).
 *
 *   gcc -O3 -march=native -fopenmp -o /tmp/emd_chain_study tools/emd_chain_study.c -lm
 *   /tmp/emd_chain_study x1.f32 x2.f32 B N R0 R1
 *
 * The tail of the auction is ~40-180 eviction chains per cloud: a bidder wins an object, the evicted owner bids in the
 * next round, ... until a free object is hit.  A chain could run k rounds ahead on its own workgroup without the
 * cluster-wide wait of every round iff no OTHER chain's price rise lands, within those k rounds, on an object that is the
 * best or the second-best of one of its searches (prices only rise, so a rise elsewhere changes neither its increment
 * nor its object), and no two chains bid for the same object.  This tool runs the exact auction (emd_cuda.cu:95-215,
 * exhaustive bids) and reports, for the rounds R0 <= r < R1:
 *   * windows: from every window start, the largest k <= 16 such that the sets {best, second best} of different chains
 *     over rounds r .. r+k-1 are pairwise disjoint (greedy tiling: the next window starts where this one ends);
 *   * per window length k = 2, 4, 8: the fraction of chains with no conflict at all (what a partial roll-back could keep).
 * Test infrastructure only. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static inline float sqdist3(float dx, float dy, float dz) { return fmaf(dz, dz, fmaf(dy, dy, dx * dx)); }
static inline float value(float s, float p) { return (float)(3.0 - (double)sqrtf(s) - (double)p); }

typedef struct { int chain, best, second; } Rec;

int main(int argc, char **argv) {
  if (argc < 7) { fprintf(stderr, "usage\n"); return 2; }
  const int B = atoi(argv[3]), n = atoi(argv[4]), R0 = atoi(argv[5]), R1 = atoi(argv[6]);
  const float eps = 0.004f;
  float *X1 = malloc(sizeof(float) * (size_t)B * n * 3), *X2 = malloc(sizeof(float) * (size_t)B * n * 3);
  FILE *f = fopen(argv[1], "rb"); if (!f || fread(X1, 4, (size_t)B * n * 3, f) != (size_t)B * n * 3) return 1; fclose(f);
  f = fopen(argv[2], "rb"); if (!f || fread(X2, 4, (size_t)B * n * 3, f) != (size_t)B * n * 3) return 1; fclose(f);
  double t_windows = 0, t_rounds = 0, t_hist[17] = {0}, t_keep[3] = {0}, t_keepn[3] = {0}, t_bidders = 0;
#pragma omp parallel for schedule(dynamic, 1)
  for (int cl = 0; cl < B; ++cl) {
    const float *x1 = X1 + (size_t)cl * n * 3, *x2 = X2 + (size_t)cl * n * 3;
    float *price = calloc(n, sizeof(float)), *binc = calloc(n, sizeof(float)), *maxinc = calloc(n, sizeof(float));
    int *ass = malloc(sizeof(int) * n), *assinv = malloc(sizeof(int) * n), *bid = calloc(n, sizeof(int)), *maxidx = calloc(n, sizeof(int));
    int *unass = malloc(sizeof(int) * n), *chain_of = malloc(sizeof(int) * n), *sec = malloc(sizeof(int) * n);
    for (int j = 0; j < n; ++j) { ass[j] = assinv[j] = -1; chain_of[j] = -1; }
    int nchain = 0;
    const int NR = R1 - R0;
    Rec **recs = calloc(NR, sizeof(Rec *)); int *nrec = calloc(NR, sizeof(int));
    for (int it = 0; it < R1; ++it) {
      int cnt = 0; for (int j = 0; j < n; ++j) if (ass[j] == -1) unass[cnt++] = j;
      if (!cnt) break;
      if (it >= R0) { recs[it - R0] = malloc(sizeof(Rec) * cnt); nrec[it - R0] = cnt; }
      for (int u = 0; u < cnt; ++u) {
        const int j = unass[u]; const float *q = x1 + j * 3;
        float best = -1e9f, better = -1e9f; int bi = -1, b2i = -1;
        for (int k = 0; k < n; ++k) {
          const float v = value(sqdist3(x2[k * 3] - q[0], x2[k * 3 + 1] - q[1], x2[k * 3 + 2] - q[2]), price[k]);
          if (v > best) { better = best; b2i = bi; best = v; bi = k; } else if (v > better) { better = v; b2i = k; }
        }
        bid[j] = bi; sec[j] = b2i; binc[j] = best - better + eps;
        if (binc[j] > maxinc[bi]) maxinc[bi] = binc[j];
        if (it >= R0) {
          if (chain_of[j] < 0) chain_of[j] = nchain++;
          recs[it - R0][u] = (Rec){chain_of[j], bi, b2i};
        }
      }
      for (int u = 0; u < cnt; ++u) { const int j = unass[u]; const int o = bid[j]; const float bi = binc[j], mi = maxinc[o]; if ((double)bi - 1e-6 <= (double)mi && (double)mi <= (double)bi + 1e-6) maxidx[o] = j; }
      for (int u = 0; u < cnt; ++u) {
        const int j = unass[u]; const int o = bid[j];
        if (maxidx[o] == j) {
          const int ai = assinv[o];
          if (ai != -1) { ass[ai] = -1; chain_of[ai] = chain_of[j]; }   /* the evicted owner carries the chain on */
          chain_of[j] = -1;
          assinv[o] = j; ass[j] = o; price[o] += binc[j]; maxinc[o] = -1e9f;
        }
      }
    }
    /* ---- windows */
    int *owner_chain = malloc(sizeof(int) * n), *stamp = calloc(n, sizeof(int));
    double windows = 0, rounds = 0, hist[17] = {0}, keep[3] = {0}, keepn[3] = {0}, bidders = 0;
    int gen = 0;
    for (int r = 0; r < NR && nrec[r];) {
      ++gen;
      int k = 0;
      for (; k < 16 && r + k < NR && nrec[r + k]; ++k) {
        int ok = 1;
        /* a round's objects must not collide with another chain's earlier objects of the window -- nor within the round */
        for (int u = 0; u < nrec[r + k] && ok; ++u) {
          const Rec e = recs[r + k][u]; const int ob[2] = {e.best, e.second};
          for (int a = 0; a < 2; ++a) if (ob[a] >= 0 && stamp[ob[a]] == gen && owner_chain[ob[a]] != e.chain) ok = 0;
        }
        if (!ok) break;
        for (int u = 0; u < nrec[r + k]; ++u) {
          const Rec e = recs[r + k][u]; const int ob[2] = {e.best, e.second};
          for (int a = 0; a < 2; ++a) if (ob[a] >= 0) {
            if (stamp[ob[a]] == gen && owner_chain[ob[a]] != e.chain) ok = 0;   /* two chains in the same round */
            stamp[ob[a]] = gen; owner_chain[ob[a]] = e.chain;
          }
        }
        if (!ok) break;
      }
      if (k == 0) k = 1;   /* a round that conflicts with itself is run synchronously */
      windows += 1; rounds += k; hist[k] += 1;
      r += k;
    }
    /* ---- fixed windows of 2 / 4 / 8 rounds: chains without any conflict */
    const int KS[3] = {2, 4, 8};
    char *bad = malloc(nchain + 1);
    for (int w = 0; w < 3; ++w)
      for (int r = 0; r + KS[w] <= NR && nrec[r + KS[w] - 1]; r += KS[w]) {
        ++gen; memset(bad, 0, nchain + 1);
        int maxc = 0;
        for (int k = 0; k < KS[w]; ++k) for (int u = 0; u < nrec[r + k]; ++u) {
          const Rec e = recs[r + k][u]; const int ob[2] = {e.best, e.second};
          for (int a = 0; a < 2; ++a) if (ob[a] >= 0) {
            if (stamp[ob[a]] == gen && owner_chain[ob[a]] != e.chain) { bad[e.chain] = 1; bad[owner_chain[ob[a]]] = 1; }
            stamp[ob[a]] = gen; owner_chain[ob[a]] = e.chain;
          }
          if (e.chain > maxc) maxc = e.chain;
        }
        /* chains alive in the window = those that bid in its first round */
        int alive = 0, good = 0;
        for (int u = 0; u < nrec[r]; ++u) { ++alive; good += !bad[recs[r][u].chain]; }
        keep[w] += good; keepn[w] += alive;
      }
    for (int r = 0; r < NR; ++r) bidders += nrec[r];
#pragma omp critical
    {
      t_windows += windows; t_rounds += rounds; t_bidders += bidders;
      for (int k = 0; k < 17; ++k) t_hist[k] += hist[k];
      for (int w = 0; w < 3; ++w) { t_keep[w] += keep[w]; t_keepn[w] += keepn[w]; }
    }
  }
  printf("clouds %d n %d rounds %d..%d: %.1f bidders (chains) per round\n", B, n, R0, R1 - 1, t_bidders / (t_rounds + 1e-9));
  printf("  greedy conflict-free windows: %.0f rounds in %.0f windows = %.2f rounds per cluster-wide synchronisation\n", t_rounds, t_windows, t_rounds / (t_windows + 1e-9));
  printf("  window length histogram (share of windows):");
  for (int k = 1; k <= 16; ++k) if (t_hist[k] > 0) printf(" %d: %.1f%%", k, 100 * t_hist[k] / t_windows);
  printf("\n  chains untouched by any conflict in fixed windows of 2 / 4 / 8 rounds: %.1f%% / %.1f%% / %.1f%%\n",
         100 * t_keep[0] / (t_keepn[0] + 1e-9), 100 * t_keep[1] / (t_keepn[1] + 1e-9), 100 * t_keep[2] / (t_keepn[2] + 1e-9));
  return 0;
}
