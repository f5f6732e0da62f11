/* Counting study for the auction's spatial index (round 6): "count before build".
 *
 *   gcc -O2 -fopenmp -o /tmp/emd_index_study tools/emd_index_study.c -lm
 *   /tmp/emd_index_study x1.f32 x2.f32 B N [R0] [curve: 0 Morton, 1 Hilbert] [R1]
 *
 * Runs the exact auction (emd_cuda.cu:95-215; exhaustive bids, same arithmetic as oracle/mvp_oracle.c, ties to the
 * lowest index -- the counts do not depend on the tie rule) and, for every bid of rounds >= R0, counts what a lossless
 * pruned search would LOAD under two indices over the objects:
 *   G  the uniform grid of csrc/emd_common.h (g^3 cells, g <= 12, ~12 objects per cell of a uniform volume; exact box
 *      and exact cheapest member per cell; a passing cell is loaded in chunks of 16 members; seed = home cell + hints)
 *   B  a two-level box hierarchy over a space-filling-curve order: leaf = 16 consecutive slots, node = 16 leaves,
 *      exact box + exact cheapest member each; seed = the leaves of the two hints + the person's home leaf (+ next)
 * A visit step of the kernels loads 16 chunks; the round lasts as long as its slowest search, so the per-round MAXIMUM
 * of the steps is what the tail round pays.  Test infrastructure only (tools/), not part of the product. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static inline float sqdist3(float dx, float dy, float dz) { return fmaf(dz, dz, fmaf(dy, dy, dx * dx)); }
static inline float value(float s, float p) { return (float)(3.0 - (double)sqrtf(s) - (double)p); }
#define MARGIN 1e-5f

typedef struct { float lo[3], hi[3], pmin; } Box;

static void box_reset(Box *b) { for (int a = 0; a < 3; ++a) { b->lo[a] = INFINITY; b->hi[a] = -INFINITY; } b->pmin = INFINITY; }
static void box_add(Box *b, const float *p, float price) {
  for (int a = 0; a < 3; ++a) { if (p[a] < b->lo[a]) b->lo[a] = p[a]; if (p[a] > b->hi[a]) b->hi[a] = p[a]; }
  if (price < b->pmin) b->pmin = price;
}
static int box_pass(const Box *b, const float *q, float tm, int use_price) {
  float d[3];
  for (int a = 0; a < 3; ++a) d[a] = fmaxf(fmaxf(b->lo[a] - q[a], q[a] - b->hi[a]), 0.f);
  const float tq = tm - (use_price ? b->pmin : 0.f);
  return tq >= 0.f && sqdist3(d[0], d[1], d[2]) <= tq * tq;
}

static uint32_t part3(uint32_t v) {  /* 10 bits -> every third bit */
  v &= 1023; v = (v | (v << 16)) & 0x30000FF; v = (v | (v << 8)) & 0x300F00F; v = (v | (v << 4)) & 0x30C30C3; v = (v | (v << 2)) & 0x9249249; return v;
}
static uint32_t morton(uint32_t x, uint32_t y, uint32_t z) { return part3(x) | (part3(y) << 1) | (part3(z) << 2); }
/* Hilbert index in 3-D, `bits` bits per axis (Skilling's transpose algorithm) */
static uint32_t hilbert(uint32_t x, uint32_t y, uint32_t z, int bits) {
  uint32_t X[3] = {x, y, z};
  uint32_t M = 1u << (bits - 1), P, Q, t;
  for (Q = M; Q > 1; Q >>= 1) {
    P = Q - 1;
    for (int i = 0; i < 3; ++i) {
      if (X[i] & Q) X[0] ^= P;
      else { t = (X[0] ^ X[i]) & P; X[0] ^= t; X[i] ^= t; }
    }
  }
  for (int i = 1; i < 3; ++i) X[i] ^= X[i - 1];
  t = 0;
  for (Q = M; Q > 1; Q >>= 1) if (X[2] & Q) t ^= Q - 1;
  for (int i = 0; i < 3; ++i) X[i] ^= t;
  uint32_t h = 0;
  for (int b = bits - 1; b >= 0; --b) for (int i = 0; i < 3; ++i) h = (h << 1) | ((X[i] >> b) & 1u);
  return h;
}

typedef struct { uint32_t key; int idx; } KV;
static int kv_cmp(const void *a, const void *b) {
  const KV *x = a, *y = b;
  if (x->key != y->key) return x->key < y->key ? -1 : 1;
  return x->idx - y->idx;
}

typedef struct {
  double searches, g_nsub, g_cells, g_chunks, g_steps, g_multi, b_nodes, b_leaftests, b_leaves, b_steps, b_multi, ideal;
  double b_leaves_np, b_nodes_np;        /* without price bounds */
  double rounds, g_maxsteps, b_maxsteps, g_sumchunks_round, b_sumleaves_round, bidders;
  double g_seedgap, b_seedgap;           /* seed threshold - final threshold, in grid cell widths */
  double b2_leaves, b2_steps, b2_maxsteps, ideal_n;
  double p_rem[2], p_zero[2], p_maxsteps[2], p_pref[2];   /* prefetch variants: leaves still to visit, searches with none, per-round max extra steps, groups prefetched */  /* variant: seed without the home leaf's neighbour */
} Stats;

int main(int argc, char **argv) {
  if (argc < 5) { fprintf(stderr, "usage: %s x1.f32 x2.f32 B N [R0] [curve]\n", argv[0]); return 2; }
  const int B = atoi(argv[3]), n = atoi(argv[4]);
  const int R0 = argc > 5 ? atoi(argv[5]) : 300;
  const int curve = argc > 6 ? atoi(argv[6]) : 0;
  const int R1 = argc > 7 ? atoi(argv[7]) : 3000;   /* the simulation stops after round R1 - 1 */
  const int iters = 3000;
  const float eps = 0.004f;
  float *X1 = malloc(sizeof(float) * (size_t)B * n * 3), *X2 = malloc(sizeof(float) * (size_t)B * n * 3);
  FILE *f = fopen(argv[1], "rb"); if (!f || fread(X1, 4, (size_t)B * n * 3, f) != (size_t)B * n * 3) { fprintf(stderr, "x1?\n"); return 1; } fclose(f);
  f = fopen(argv[2], "rb"); if (!f || fread(X2, 4, (size_t)B * n * 3, f) != (size_t)B * n * 3) { fprintf(stderr, "x2?\n"); return 1; } fclose(f);
  Stats tot; memset(&tot, 0, sizeof tot);
#pragma omp parallel for schedule(dynamic, 1)
  for (int cl = 0; cl < B; ++cl) {
    const float *x1 = X1 + (size_t)cl * n * 3, *x2 = X2 + (size_t)cl * n * 3;
    Stats st; memset(&st, 0, sizeof st);
    /* ---- geometry shared by both indices: the box of both clouds (emd.hip grid build) */
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int k = 0; k < n; ++k) for (int a = 0; a < 3; ++a) {
      lo[a] = fminf(lo[a], fminf(x1[k * 3 + a], x2[k * 3 + a])); hi[a] = fmaxf(hi[a], fmaxf(x1[k * 3 + a], x2[k * 3 + a]));
    }
    float ext = fmaxf(hi[0] - lo[0], fmaxf(hi[1] - lo[1], hi[2] - lo[2]));
    int g = 2; while (g < 12 && (g + 1) * (g + 1) * (g + 1) * 12 <= n) ++g;
    const float invh = (float)g / ext;
    const int ncell = g * g * g;
    /* ---- G: cells */
    int *cell_of = malloc(sizeof(int) * n), *cstart = calloc(ncell + 1, sizeof(int)), *cmem = malloc(sizeof(int) * n);
    for (int k = 0; k < n; ++k) {
      int c[3]; for (int a = 0; a < 3; ++a) { int v = (int)((x2[k * 3 + a] - lo[a]) * invh); c[a] = v < 0 ? 0 : v > g - 1 ? g - 1 : v; }
      cell_of[k] = (c[2] * g + c[1]) * g + c[0]; cstart[cell_of[k] + 1]++;
    }
    for (int c = 0; c < ncell; ++c) cstart[c + 1] += cstart[c];
    { int *fill = calloc(ncell, sizeof(int)); for (int k = 0; k < n; ++k) cmem[cstart[cell_of[k]] + fill[cell_of[k]]++] = k; free(fill); }
    Box *cbox = malloc(sizeof(Box) * ncell);
    /* ---- B: curve order, leaves of 16, nodes of 16 leaves */
    const int bits = 9;
    KV *kv = malloc(sizeof(KV) * n);
    const float qs = (float)(1 << bits) / ext;
    for (int k = 0; k < n; ++k) {
      uint32_t c[3]; for (int a = 0; a < 3; ++a) { int v = (int)((x2[k * 3 + a] - lo[a]) * qs); c[a] = v < 0 ? 0 : v > (1 << bits) - 1 ? (1 << bits) - 1 : v; }
      kv[k].key = curve ? hilbert(c[0], c[1], c[2], bits) : morton(c[0], c[1], c[2]); kv[k].idx = k;
    }
    qsort(kv, n, sizeof(KV), kv_cmp);
    int *slot_of = malloc(sizeof(int) * n);  /* object -> slot of the curve order */
    for (int s = 0; s < n; ++s) slot_of[kv[s].idx] = s;
    const int nleaf = n / 16, nnode = (nleaf + 15) / 16;
    Box *lbox = malloc(sizeof(Box) * nleaf), *nbox = malloc(sizeof(Box) * nnode);
    int *home = malloc(sizeof(int) * n);     /* person -> home leaf (position of its key in the order) */
    for (int j = 0; j < n; ++j) {
      uint32_t c[3]; for (int a = 0; a < 3; ++a) { int v = (int)((x1[j * 3 + a] - lo[a]) * qs); c[a] = v < 0 ? 0 : v > (1 << bits) - 1 ? (1 << bits) - 1 : v; }
      const uint32_t key = curve ? hilbert(c[0], c[1], c[2], bits) : morton(c[0], c[1], c[2]);
      int a = 0, b = n; while (a < b) { int m = (a + b) / 2; if (kv[m].key < key) a = m + 1; else b = m; }
      home[j] = (a >= n ? n - 1 : a) >> 4;
    }
    /* ---- auction state */
    float *price = calloc(n, sizeof(float)), *binc = calloc(n, sizeof(float)), *maxinc = calloc(n, sizeof(float));
    int *ass = malloc(sizeof(int) * n), *assinv = malloc(sizeof(int) * n), *bid = calloc(n, sizeof(int)), *maxidx = calloc(n, sizeof(int));
    int *unass = malloc(sizeof(int) * n), *p1 = malloc(sizeof(int) * n), *p2 = malloc(sizeof(int) * n);
    for (int j = 0; j < n; ++j) { ass[j] = assinv[j] = -1; p1[j] = p2[j] = -1; }
    char *cdirty = calloc(ncell, 1), *ldirty = calloc(nleaf, 1);
    int first = 1;
    for (int it = 0; it < iters; ++it) {
      int cnt = 0; for (int j = 0; j < n; ++j) if (ass[j] == -1) unass[cnt++] = j;
      if (!cnt || it >= R1) break;
      const int counting = it >= R0;
      if (counting) {
        /* exact boxes / cheapest members (the kernels keep them exact by re-scanning what a winner touched) */
        if (first) {
          for (int c = 0; c < ncell; ++c) { box_reset(&cbox[c]); for (int s = cstart[c]; s < cstart[c + 1]; ++s) box_add(&cbox[c], x2 + cmem[s] * 3, price[cmem[s]]); }
          for (int l = 0; l < nleaf; ++l) { box_reset(&lbox[l]); for (int s = 16 * l; s < 16 * l + 16; ++s) box_add(&lbox[l], x2 + kv[s].idx * 3, price[kv[s].idx]); }
          first = 0;
        } else {
          for (int c = 0; c < ncell; ++c) if (cdirty[c]) { cbox[c].pmin = INFINITY; for (int s = cstart[c]; s < cstart[c + 1]; ++s) cbox[c].pmin = fminf(cbox[c].pmin, price[cmem[s]]); }
          for (int l = 0; l < nleaf; ++l) if (ldirty[l]) { lbox[l].pmin = INFINITY; for (int s = 16 * l; s < 16 * l + 16; ++s) lbox[l].pmin = fminf(lbox[l].pmin, price[kv[s].idx]); }
        }
        memset(cdirty, 0, ncell); memset(ldirty, 0, nleaf);
        for (int m = 0; m < nnode; ++m) {
          box_reset(&nbox[m]);
          for (int l = 16 * m; l < 16 * m + 16 && l < nleaf; ++l) {
            for (int a = 0; a < 3; ++a) { nbox[m].lo[a] = fminf(nbox[m].lo[a], lbox[l].lo[a]); nbox[m].hi[a] = fmaxf(nbox[m].hi[a], lbox[l].hi[a]); }
            nbox[m].pmin = fminf(nbox[m].pmin, lbox[l].pmin);
          }
        }
      }
      int g_max = 0, b_max = 0, b2_max = 0, p_max[2] = {0, 0}; double g_sum = 0, b_sum = 0;
      for (int u = 0; u < cnt; ++u) {
        const int j = unass[u];
        const float *q = x1 + j * 3;
        float best = -1e9f, better = -1e9f; int bi = -1, b2i = -1;
        for (int k = 0; k < n; ++k) {
          const float v = value(sqdist3(x2[k * 3] - q[0], x2[k * 3 + 1] - q[1], x2[k * 3 + 2] - q[2]), price[k]);
          if (v > best) { better = best; b2i = bi; best = v; bi = k; } else if (v > better) { better = v; b2i = k; }
        }
        if (counting) {
          const float tm_final = (3.0f - better) + MARGIN;
          /* ideal: objects the exact filter lets through at the final threshold */
          int ideal = 0;
          if ((u & 7) == 0) for (int k = 0; k < n; ++k) { const float tq = tm_final - price[k]; if (tq >= 0.f && sqdist3(x2[k * 3] - q[0], x2[k * 3 + 1] - q[1], x2[k * 3 + 2] - q[2]) <= tq * tq) ++ideal; }
          st.ideal += ideal; st.ideal_n += (u & 7) == 0; st.searches += 1;
          /* ---- G */
          {
            int c[3]; for (int a = 0; a < 3; ++a) { int v = (int)((q[a] - lo[a]) * invh); c[a] = v < 0 ? 0 : v > g - 1 ? g - 1 : v; }
            const int c0 = (c[2] * g + c[1]) * g + c[0];
            float a1 = -1e9f, a2 = -1e9f; int have = 0;
            for (int s = cstart[c0]; s < cstart[c0 + 1]; ++s) { const int k = cmem[s]; const float v = value(sqdist3(x2[k * 3] - q[0], x2[k * 3 + 1] - q[1], x2[k * 3 + 2] - q[2]), price[k]); if (v > a1) { a2 = a1; a1 = v; } else if (v > a2) a2 = v; ++have; }
            const int hs[2] = {p1[j], p2[j]};
            for (int h = 0; h < 2; ++h) if (hs[h] >= 0 && cell_of[hs[h]] != c0) { const int k = hs[h]; const float v = value(sqdist3(x2[k * 3] - q[0], x2[k * 3 + 1] - q[1], x2[k * 3 + 2] - q[2]), price[k]); if (v > a1) { a2 = a1; a1 = v; } else if (v > a2) a2 = v; ++have; }
            if (have < 2) { a1 = a2 = -1e9f; for (int s = 0; s < 64; ++s) { const int k = cmem[s]; const float v = value(sqdist3(x2[k * 3] - q[0], x2[k * 3 + 1] - q[1], x2[k * 3 + 2] - q[2]), price[k]); if (v > a1) { a2 = a1; a1 = v; } else if (v > a2) a2 = v; } }
            const float tm = (3.0f - a2) + MARGIN;
            st.g_seedgap += (tm - tm_final) * invh;
            const float r = tm * invh + 1e-3f;
            int i0[3], i1[3];
            for (int a = 0; a < 3; ++a) { const float fa = (q[a] - lo[a]) * invh; i0[a] = (int)fminf(fmaxf(floorf(fa - r), 0.f), (float)(g - 1)); i1[a] = (int)fminf(fmaxf(floorf(fa + r), 0.f), (float)(g - 1)); }
            int cells = 0, chunks = 0;
            const int nsub = (i1[0] - i0[0] + 1) * (i1[1] - i0[1] + 1) * (i1[2] - i0[2] + 1);
            if (2 * nsub > ncell) { chunks = n / 16; cells = ncell; }   /* linear scan */
            else for (int z = i0[2]; z <= i1[2]; ++z) for (int y = i0[1]; y <= i1[1]; ++y) for (int x = i0[0]; x <= i1[0]; ++x) {
              const int cc = (z * g + y) * g + x; const int m = cstart[cc + 1] - cstart[cc];
              if (m && box_pass(&cbox[cc], q, tm, 1)) { ++cells; chunks += (m + 15) / 16; }
            }
            const int steps = (chunks + 15) / 16;
            st.g_nsub += nsub; st.g_cells += cells; st.g_chunks += chunks; st.g_steps += steps; st.g_multi += steps > 1;
            if (steps > g_max) g_max = steps; g_sum += chunks;
          }
          /* ---- B (variant 0: seed = leaves of p1, p2, home, home's neighbour; variant 1: without the neighbour) */
          for (int var = 0; var < 2; ++var) {
            int ls[4], nl = 0;
            const int cand[4] = {p1[j] >= 0 ? slot_of[p1[j]] >> 4 : -1, p2[j] >= 0 ? slot_of[p2[j]] >> 4 : -1, home[j], home[j] + 1 < nleaf ? home[j] + 1 : home[j] - 1};
            for (int c = 0; c < (var ? 3 : 4); ++c) { int dup = cand[c] < 0; for (int e = 0; e < nl; ++e) dup |= ls[e] == cand[c]; if (!dup) ls[nl++] = cand[c]; }
            float a1 = -1e9f, a2 = -1e9f;
            for (int e = 0; e < nl; ++e) for (int s = 16 * ls[e]; s < 16 * ls[e] + 16; ++s) { const int k = kv[s].idx; const float v = value(sqdist3(x2[k * 3] - q[0], x2[k * 3 + 1] - q[1], x2[k * 3 + 2] - q[2]), price[k]); if (v > a1) { a2 = a1; a1 = v; } else if (v > a2) a2 = v; }
            const float tm = (3.0f - a2) + MARGIN;
            int nodes = 0, leaves = 0, nodes_np = 0, leaves_np = 0;
            for (int m = 0; m < nnode; ++m) {
              const int pn = box_pass(&nbox[m], q, tm, 1), pn0 = box_pass(&nbox[m], q, tm, 0);
              nodes += pn; nodes_np += pn0;
              if (pn0) for (int l = 16 * m; l < 16 * m + 16 && l < nleaf; ++l) { if (pn && box_pass(&lbox[l], q, tm, 1)) ++leaves; if (box_pass(&lbox[l], q, tm, 0)) ++leaves_np; }
            }
            const int steps = (leaves + 15) / 16;
            if (var == 0) {
              st.b_seedgap += (tm - tm_final) * invh;
              st.b_nodes += nodes; st.b_leaftests += 16 * nodes; st.b_leaves += leaves; st.b_steps += steps; st.b_multi += steps > 1;
              st.b_nodes_np += nodes_np; st.b_leaves_np += leaves_np;
              if (steps > b_max) b_max = steps; b_sum += leaves;
            } else {
              st.b2_leaves += leaves; st.b2_steps += steps; if (steps > b2_max) b2_max = steps;
            }
          }
        }
        if (counting) {
          /* ---- P: ONE round trip over up to four aligned 64-slot groups (home, previous best, previous second best and
             a sibling group), exact top-2 of those <= 256 objects -> threshold; what is left to visit afterwards? */
          for (int var = 0; var < 2; ++var) {
            const int gh = home[j] >> 2, g1 = p1[j] >= 0 ? slot_of[p1[j]] >> 6 : -1, g2 = p2[j] >= 0 ? slot_of[p2[j]] >> 6 : -1;
            const int cand[4] = {g1, g2, gh, var == 0 ? (gh ^ 1) : (g1 >= 0 ? (g1 ^ 1) : (gh ^ 1))};
            int gs[4], ng = 0;
            for (int c = 0; c < 4; ++c) { int dup = cand[c] < 0 || cand[c] >= n / 64; for (int e = 0; e < ng; ++e) dup |= gs[e] == cand[c]; if (!dup) gs[ng++] = cand[c]; }
            float a1 = -1e9f, a2 = -1e9f;
            for (int e = 0; e < ng; ++e) for (int s = 64 * gs[e]; s < 64 * gs[e] + 64; ++s) { const int k = kv[s].idx; const float v = value(sqdist3(x2[k * 3] - q[0], x2[k * 3 + 1] - q[1], x2[k * 3 + 2] - q[2]), price[k]); if (v > a1) { a2 = a1; a1 = v; } else if (v > a2) a2 = v; }
            const float tm = (3.0f - a2) + MARGIN;
            int rem = 0;
            for (int m = 0; m < nnode; ++m) if (box_pass(&nbox[m], q, tm, 1))
              for (int l = 16 * m; l < 16 * m + 16 && l < nleaf; ++l) if (box_pass(&lbox[l], q, tm, 1)) {
                int pre = 0; for (int e = 0; e < ng; ++e) pre |= (l >> 2) == gs[e];
                rem += !pre;
              }
            st.p_rem[var] += rem; st.p_zero[var] += rem == 0; st.p_pref[var] += ng;
            const int steps = (rem + 15) / 16;
            if (steps > p_max[var]) p_max[var] = steps;
          }
        }
        p1[j] = bi; p2[j] = b2i;
        bid[j] = bi; binc[j] = best - better + eps;
        if (binc[j] > maxinc[bi]) maxinc[bi] = binc[j];
      }
      if (counting) { st.rounds += 1; st.g_maxsteps += g_max; st.b_maxsteps += b_max; st.b2_maxsteps += b2_max; st.p_maxsteps[0] += p_max[0]; st.p_maxsteps[1] += p_max[1]; st.g_sumchunks_round += g_sum; st.b_sumleaves_round += b_sum; st.bidders += cnt; }
      for (int u = 0; u < cnt; ++u) { const int j = unass[u]; const int o = bid[j]; const float bi = binc[j], mi = maxinc[o]; if ((double)bi - 1e-6 <= (double)mi && (double)mi <= (double)bi + 1e-6) maxidx[o] = j; }
      const int last = it == iters - 1;
      for (int u = 0; u < cnt; ++u) {
        const int j = unass[u]; const int o = bid[j];
        if (last || maxidx[o] == j) {
          const int ai = assinv[o];
          if (!last && ai != -1) ass[ai] = -1;
          assinv[o] = j; ass[j] = o; price[o] += binc[j]; maxinc[o] = -1e9f;
          cdirty[cell_of[o]] = 1; ldirty[slot_of[o] >> 4] = 1;
        }
      }
    }
#pragma omp critical
    {
      double *a = (double *)&tot, *b = (double *)&st;
      for (size_t i = 0; i < sizeof(Stats) / sizeof(double); ++i) a[i] += b[i];
    }
    free(cell_of); free(cstart); free(cmem); free(cbox); free(kv); free(slot_of); free(lbox); free(nbox); free(home);
    free(price); free(binc); free(maxinc); free(ass); free(assinv); free(bid); free(maxidx); free(unass); free(p1); free(p2); free(cdirty); free(ldirty);
  }
  const double S = tot.searches + 1e-9, R = tot.rounds + 1e-9;
  printf("clouds %d n %d rounds >= %d curve %s: %.0f searches, %.1f bidders / round, ideal %.1f objects pass the exact filter\n", B, n, R0, curve ? "hilbert" : "morton", tot.searches, tot.bidders / R, tot.ideal / (tot.ideal_n + 1e-9));
  printf("  G grid   : seed gap %.3f cells | %.0f cells tested, %.1f pass, %.1f chunks loaded | steps %.2f (>1: %.1f %%) | per round: max steps %.2f, chunks %.0f\n",
         tot.g_seedgap / S, tot.g_nsub / S, tot.g_cells / S, tot.g_chunks / S, tot.g_steps / S, 100 * tot.g_multi / S, tot.g_maxsteps / R, tot.g_sumchunks_round / R);
  printf("  B leaves : seed gap %.3f cells | %.1f of %d nodes pass (%.1f without price), %.0f leaves tested, %.1f pass (%.1f without price) | steps %.2f (>1: %.1f %%) | per round: max steps %.2f, chunks %.0f\n",
         tot.b_seedgap / S, tot.b_nodes / S, (n / 16 + 15) / 16, tot.b_nodes_np / S, tot.b_leaftests / S, tot.b_leaves / S, tot.b_leaves_np / S, tot.b_steps / S, 100 * tot.b_multi / S, tot.b_maxsteps / R, tot.b_sumleaves_round / R);
  printf("  B, seed without the neighbour leaf: %.1f leaves pass, steps %.2f, per-round max %.2f\n", tot.b2_leaves / S, tot.b2_steps / S, tot.b2_maxsteps / R);
  for (int v = 0; v < 2; ++v)
    printf("  P%d one round trip over the 64-slot groups of {home, best, second, %s}: %.2f groups, then %.2f leaves left to visit; nothing left in %.1f %% of the searches; per round: max extra steps %.2f\n",
           v, v ? "best's sibling" : "home's sibling", tot.p_pref[v] / S, tot.p_rem[v] / S, 100 * tot.p_zero[v] / S, tot.p_maxsteps[v] / R);
  return 0;
}
