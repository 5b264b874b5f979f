/*
 * dis_oracle.c -- CPU ORACLE (test infrastructure, see dis_oracle.h).
 *
 * Plain scalar C restatement of the reference's kroeger/ DIS optical-flow path.  Every function
 * cites the reference file:line it follows (paths relative to the reference root).
 * Build with -ffp-contract=off: the reference build (-O3 -msse4, no FMA target) evaluates every
 * product and sum separately, and the expression order below is the reference's.
 *
 * Deliberate definitions where the reference leaves behaviour open:
 *   (D1) per-patch reductions use dis_sum()'s fixed order (reference: Eigen packet order,
 *        version dependent; Eigen is not part of the reference tree).
 *   (D2) a patch whose start position is outside the valid region keeps an all-zero pweight
 *        (reference: pweight is left as it was allocated -- uninitialised, patch.cpp:135-141).
 *   (D3) a non-finite LK update is treated like an outlier (reset to p_in, stop).  The reference
 *        would convert NaN to int (undefined behaviour, patch.cpp:345-348).
 *   (D5) InitializeFromCoarserOF clamps the half-resolution index into the coarser array (the reference reads
 *        one row / column out of bounds when the level size is odd, which only `initflow` can make happen).
 *   (D4) 2x2 half-resolution = ((a+c)+(b+d))*0.25 (rows first); for 8-bit valued input every
 *        order is exact up to level 7, so this only matters for non-integer input.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <stdint.h>
#include <pthread.h>
#include <time.h>
#include "dis_oracle.h"

/* ------------------------------------------------------------------------------------------- */
/* allocation: every buffer of the port goes through dis_alloc / dis_release.  Outside          */
/* dis_flow_many() they are malloc / free.  Inside it every worker thread keeps the blocks it    */
/* releases in a thread-private cache keyed by size: a pair asks for the same sequence of sizes   */
/* as the pair before it, so from a thread's second pair on nothing reaches the C library (no    */
/* mmap / munmap / page faults per pair, which is what bound the all-cores leg of round 2).      */
/* ------------------------------------------------------------------------------------------- */
typedef struct dis_blk { size_t size; struct dis_blk *next; } dis_blk;      /* 16-byte header */
typedef struct { size_t size[96]; dis_blk *head[96]; int n; } dis_pool;
static __thread dis_pool *tl_pool = NULL;

static void *dis_alloc(size_t n, int zero)
{
  dis_blk *b = NULL;
  if (tl_pool) {
    for (int i = 0; i < tl_pool->n; ++i)
      if (tl_pool->size[i] == n && tl_pool->head[i]) { b = tl_pool->head[i]; tl_pool->head[i] = b->next; break; }
  }
  if (!b) { b = (dis_blk *)malloc(sizeof(dis_blk) + n); if (!b) return NULL; b->size = n; }
  if (zero) memset(b + 1, 0, n);
  return b + 1;
}
static void dis_release(void *p)
{
  if (!p) return;
  dis_blk *b = (dis_blk *)p - 1;
  if (tl_pool) {
    int i = 0;
    for (; i < tl_pool->n; ++i) if (tl_pool->size[i] == b->size) break;
    if (i == tl_pool->n && tl_pool->n < 96) { tl_pool->size[i] = b->size; tl_pool->head[i] = NULL; tl_pool->n++; }
    if (i < tl_pool->n) { b->next = tl_pool->head[i]; tl_pool->head[i] = b; return; }
  }
  free(b);
}
static void dis_pool_drain(dis_pool *pl)
{
  for (int i = 0; i < pl->n; ++i) while (pl->head[i]) { dis_blk *b = pl->head[i]; pl->head[i] = b->next; free(b); }
  pl->n = 0;
}
#define malloc(n) dis_alloc((n), 0)
#define calloc(a, b) dis_alloc((size_t)(a) * (size_t)(b), 1)
#define free(p) dis_release(p)

/* order switches of the parity-sensitivity test (tests/test_oracle.py): what the reference leaves to Eigen / OpenCV */
static int g_sum_order = 0;     /* D1: 0 = 16 partials + xor tree (default), 1 sequential, 2 Eigen-style 4-lane packets, 3 pairwise */
static int g_mean_order = 0;    /* D4: 0 = ((a+c)+(b+d))/4 (default), 1 ((a+b)+(c+d))/4, 2 (((a+b)+c)+d)/4, 3 a/4+b/4+c/4+d/4 */
void dis_set_sum_order(int o) { g_sum_order = o; }
void dis_set_mean_order(int o) { g_mean_order = o; }

/* ------------------------------------------------------------------------------------------- */
/* parameters / operating points                                                               */
/* ------------------------------------------------------------------------------------------- */

/* kroeger/run_dense.cpp:180-183 */
int dis_auto_first_scale(int imgwidth, int fratio, int patchsize)
{
  float v = (2.0f * (float)imgwidth) / ((float)fratio * (float)patchsize);
  int r = (int)floor(log2(v));
  return r > 0 ? r : 0;
}

/* kroeger/run_dense.cpp:225-268 */
void dis_op_point(int op, int width_org, int noc, dis_params *p)
{
  const int fratio = 5;
  memset(p, 0, sizeof(*p));
  p->dp_thresh = 0.05f; p->dr_thresh = 0.95f; p->res_thresh = 0.0f;
  p->patnorm = 1; p->noc = noc;
  p->tv_alpha = 10.0f; p->tv_gamma = 10.0f; p->tv_delta = 5.0f;
  p->tv_innerit = 1; p->tv_solverit = 3; p->tv_sor = 1.6f;
  p->costfct = 0; p->normoutlier = 5.0f;                          /* run_dense.cpp:228, oflow.h:63 */
  p->usefbcon = 0;                                                /* run_dense.cpp:228 */
  int sub;
  switch (op) {
    case 1: p->ps = 8;  p->patove = 0.3f;  sub = 2; p->max_iter = 16;  p->usetvref = 0; break;
    case 3: p->ps = 12; p->patove = 0.75f; sub = 4; p->max_iter = 16;  p->usetvref = 1; break;
    case 4: p->ps = 12; p->patove = 0.75f; sub = 5; p->max_iter = 128; p->usetvref = 1; break;
    case 2:
    default: p->ps = 8; p->patove = 0.4f;  sub = 2; p->max_iter = 12;  p->usetvref = 1; break;
  }
  p->min_iter = p->max_iter;
  p->sc_f = dis_auto_first_scale(width_org, fratio, p->ps);
  p->sc_l = p->sc_f - sub > 0 ? p->sc_f - sub : 0;
}

/* kroeger/run_dense.cpp:298-305 */
void dis_padded_size(int w, int h, int sc_f, int *wp, int *hp, int *padw, int *padh)
{
  int scfct = 1 << sc_f, pw = 0, ph = 0;
  int div = w % scfct; if (div > 0) pw = scfct - div;
  div = h % scfct;     if (div > 0) ph = scfct - div;
  *wp = w + pw; *hp = h + ph; *padw = pw; *padh = ph;
}

/* kroeger/run_dense.cpp:306-310: copyMakeBorder(top=floor(padh/2), bottom=ceil, left=floor(padw/2), right=ceil, REPLICATE) */
void dis_pad_frame(const float *in, int w, int h, int noc, int sc_f, float *out)
{
  int wp, hp, padw, padh;
  dis_padded_size(w, h, sc_f, &wp, &hp, &padw, &padh);
  int top = padh / 2, left = padw / 2;
  for (int y = 0; y < hp; ++y) {
    int sy = y - top; if (sy < 0) sy = 0; if (sy > h - 1) sy = h - 1;
    for (int x = 0; x < wp; ++x) {
      int sx = x - left; if (sx < 0) sx = 0; if (sx > w - 1) sx = w - 1;
      for (int c = 0; c < noc; ++c) out[((size_t)y * wp + x) * noc + c] = in[((size_t)sy * w + sx) * noc + c];
    }
  }
}

/* kroeger/run_dense.cpp:138-147 (the SELECTCHANNEL==2 build): the pyramid's level 0 is the gradient magnitude of the padded
 * frame -- cv::Sobel(ksize 1, BORDER_DEFAULT) = I(x+1) - I(x-1) / I(y+1) - I(y-1) per channel with REFLECT_101 at the frame's
 * edge (index -1 -> 1, n -> n-2), then sqrt(dx*dx + dy*dy), all in f32.  in / out: w x h x noc, w, h >= 2. */
void dis_gradient_magnitude(const float *in, int w, int h, int noc, float *out)
{
  for (int y = 0; y < h; ++y) {
    int ym = y > 0 ? y - 1 : 1, yp = y < h - 1 ? y + 1 : h - 2;
    for (int x = 0; x < w; ++x) {
      int xm = x > 0 ? x - 1 : 1, xp = x < w - 1 ? x + 1 : w - 2;
      for (int c = 0; c < noc; ++c) {
        float dx = in[((size_t)y * w + xp) * noc + c] - in[((size_t)y * w + xm) * noc + c];
        float dy = in[((size_t)yp * w + x) * noc + c] - in[((size_t)ym * w + x) * noc + c];
        float dx2 = dx * dx, dy2 = dy * dy;
        out[((size_t)y * w + x) * noc + c] = sqrtf(dx2 + dy2);
      }
    }
  }
}

/* ------------------------------------------------------------------------------------------- */
/* pyramid: kroeger/run_dense.cpp:130-178                                                      */
/* ------------------------------------------------------------------------------------------- */

int dis_level_w(const dis_pyramid *p, int l) { return p->w0 >> l; }
int dis_level_h(const dis_pyramid *p, int l) { return p->h0 >> l; }

static int reflect101(int i, int n) { if (i < 0) return -i; if (i >= n) return 2 * n - 2 - i; return i; }

dis_pyramid *dis_pyramid_build(const float *img, int wp, int hp, int noc, int sc_f, int ps)
{
  dis_pyramid *P = (dis_pyramid *)calloc(1, sizeof(dis_pyramid));
  P->nlev = sc_f + 1; P->noc = noc; P->ps = ps; P->w0 = wp; P->h0 = hp;
  P->im = (float **)calloc(P->nlev, sizeof(float *));
  P->dx = (float **)calloc(P->nlev, sizeof(float *));
  P->dy = (float **)calloc(P->nlev, sizeof(float *));
  float *prev = NULL; int pw = 0;
  for (int l = 0; l <= sc_f; ++l) {
    int w = wp >> l, h = hp >> l;
    float *cur = (float *)malloc(sizeof(float) * (size_t)w * h * noc);
    if (l == 0) memcpy(cur, img, sizeof(float) * (size_t)w * h * noc);       /* :137 clone */
    else {
      /* :150 cv::resize(.5,.5,INTER_LINEAR) == exact 2x2 mean (OpenCV maps it to INTER_AREA) */
      for (int y = 0; y < h; ++y) for (int x = 0; x < w; ++x) for (int c = 0; c < noc; ++c) {
        const float a = prev[((size_t)(2 * y) * pw + 2 * x) * noc + c];
        const float b = prev[((size_t)(2 * y) * pw + 2 * x + 1) * noc + c];
        const float cc = prev[((size_t)(2 * y + 1) * pw + 2 * x) * noc + c];
        const float d = prev[((size_t)(2 * y + 1) * pw + 2 * x + 1) * noc + c];
        float m;
        switch (g_mean_order) {
          case 1: m = ((a + b) + (cc + d)) * 0.25f; break;
          case 2: m = (((a + b) + cc) + d) * 0.25f; break;
          case 3: m = a * 0.25f + b * 0.25f + cc * 0.25f + d * 0.25f; break;
          default: m = ((a + cc) + (b + d)) * 0.25f; break;             /* definition D4 */
        }
        cur[((size_t)y * w + x) * noc + c] = m;
      }
    }
    /* :156-157 Sobel ksize=1 -> [-1 0 1], BORDER_DEFAULT = REFLECT_101; :166-175 pad */
    int tw = w + 2 * ps, th = h + 2 * ps;
    float *pim = (float *)malloc(sizeof(float) * (size_t)tw * th * noc);
    float *pdx = (float *)calloc((size_t)tw * th * noc, sizeof(float));
    float *pdy = (float *)calloc((size_t)tw * th * noc, sizeof(float));
    for (int y = 0; y < th; ++y) {
      int sy = y - ps; if (sy < 0) sy = 0; if (sy > h - 1) sy = h - 1;
      for (int x = 0; x < tw; ++x) {
        int sx = x - ps; if (sx < 0) sx = 0; if (sx > w - 1) sx = w - 1;
        for (int c = 0; c < noc; ++c) pim[((size_t)y * tw + x) * noc + c] = cur[((size_t)sy * w + sx) * noc + c];
      }
    }
    for (int y = 0; y < h; ++y) for (int x = 0; x < w; ++x) for (int c = 0; c < noc; ++c) {
      int xm = reflect101(x - 1, w), xq = reflect101(x + 1, w);
      int ym = reflect101(y - 1, h), yq = reflect101(y + 1, h);
      size_t o = ((size_t)(y + ps) * tw + (x + ps)) * noc + c;
      pdx[o] = cur[((size_t)y * w + xq) * noc + c] - cur[((size_t)y * w + xm) * noc + c];
      pdy[o] = cur[((size_t)yq * w + x) * noc + c] - cur[((size_t)ym * w + x) * noc + c];
    }
    P->im[l] = pim; P->dx[l] = pdx; P->dy[l] = pdy;
    free(prev); prev = cur; pw = w;
  }
  free(prev);
  return P;
}

void dis_pyramid_free(dis_pyramid *p)
{
  if (!p) return;
  for (int l = 0; l < p->nlev; ++l) { free(p->im[l]); free(p->dx[l]); free(p->dy[l]); }
  free(p->im); free(p->dx); free(p->dy); free(p);
}

/* ------------------------------------------------------------------------------------------- */
/* reductions (definition D1)                                                                  */
/* ------------------------------------------------------------------------------------------- */

static float dis_sum_pairwise(const float *v, int n)
{
  if (n <= 2) return n == 2 ? v[0] + v[1] : v[0];
  const int h = n / 2;
  return dis_sum_pairwise(v, h) + dis_sum_pairwise(v + h, n - h);
}

float dis_sum(const float *v, int n, int noc)
{
  if (g_sum_order == 1) {                       /* sequential, the order of a scalar loop */
    float acc = v[0];
    for (int e = 1; e < n; ++e) acc = acc + v[e];
    return acc;
  }
  if (g_sum_order == 2) {
    /* Eigen-style vectorised redux with 4-float packets (SSE, the reference's -msse4 build): two packet accumulators over
     * alternating packets, added, then the horizontal sum (a0 + a2) + (a1 + a3) (movehl / shuffle form of predux) */
    float p0[4], p1[4];
    for (int k = 0; k < 4; ++k) { p0[k] = v[k]; p1[k] = n >= 8 ? v[4 + k] : 0.0f; }
    int e = 8;
    for (; e + 8 <= n; e += 8) for (int k = 0; k < 4; ++k) { p0[k] = p0[k] + v[e + k]; p1[k] = p1[k] + v[e + 4 + k]; }
    if (e + 4 <= n) { for (int k = 0; k < 4; ++k) p0[k] = p0[k] + v[e + k]; e += 4; }
    float q[4];
    for (int k = 0; k < 4; ++k) q[k] = n >= 8 ? p0[k] + p1[k] : p0[k];
    float r = (q[0] + q[2]) + (q[1] + q[3]);
    for (; e < n; ++e) r = r + v[e];
    return r;
  }
  if (g_sum_order == 3) return dis_sum_pairwise(v, n);
  /* 16 partial sums: pixel q (element e / noc) goes to partial q % 16, elements in ascending order; then the balanced
   * tree over the 16 partials (xor 8, 4, 2, 1).  Every supported patch has a multiple of 16 pixels. */
  float lane[16]; int used[16];
  for (int i = 0; i < 16; ++i) { lane[i] = 0.0f; used[i] = 0; }
  for (int e = 0; e < n; ++e) {
    int l = (e / noc) & 15;
    if (!used[l]) { lane[l] = v[e]; used[l] = 1; } else lane[l] = lane[l] + v[e];
  }
  for (int k = 8; k >= 1; k >>= 1) {
    float t[16];
    for (int i = 0; i < 16; ++i) t[i] = lane[i] + lane[i ^ k];
    memcpy(lane, t, sizeof(t));
  }
  return lane[0];
}

static float dis_dot(const float *a, const float *b, int n, int noc, float *scratch)
{
  for (int i = 0; i < n; ++i) scratch[i] = a[i] * b[i];
  return dis_sum(scratch, n, noc);
}

/* ------------------------------------------------------------------------------------------- */
/* patch grid: kroeger/patchgrid.cpp, kroeger/patch.cpp                                        */
/* ------------------------------------------------------------------------------------------- */

/* kroeger/oflow.cpp:138-157 (camparam) + patchgrid.cpp:31-76 (grid) */
dis_grid *dis_grid_new(int w, int h, int lvl, const dis_params *p)
{
  dis_grid *g = (dis_grid *)calloc(1, sizeof(dis_grid));
  g->w = w; g->h = h; g->ps = p->ps; g->noc = p->noc; g->lvl = lvl;
  g->depth = p->depth; g->camlr = 0;
  int steps = (int)floor(p->ps * (1 - p->patove));               /* oflow.cpp:91 */
  g->steps = steps > 1 ? steps : 1;
  g->pad = p->ps; g->tmp_w = w + 2 * p->ps;
  g->lb = -(float)p->ps / 2;                                       /* oflow.cpp:147 */
  g->ubw = (float)(w + p->ps / 2 - 2);                             /* :148 */
  g->ubh = (float)(h + p->ps / 2 - 2);                             /* :149 */
  g->nopw = (int)ceil((float)w / (float)g->steps);                 /* patchgrid.cpp:43 */
  g->noph = (int)ceil((float)h / (float)g->steps);
  const int offw = (int)floor((w - (g->nopw - 1) * g->steps) / 2); /* :45 (integer division) */
  const int offh = (int)floor((h - (g->noph - 1) * g->steps) / 2);
  g->nop = g->nopw * g->noph;
  int nv = p->ps * p->ps * p->noc;
  g->pt_ref = (float *)calloc((size_t)g->nop * 2, sizeof(float));
  g->p_init = (float *)calloc((size_t)g->nop * 2, sizeof(float));
  g->p_iter = (float *)calloc((size_t)g->nop * 2, sizeof(float));
  g->tmpl = (float *)calloc((size_t)g->nop * nv, sizeof(float));
  g->tdx = (float *)calloc((size_t)g->nop * nv, sizeof(float));
  g->tdy = (float *)calloc((size_t)g->nop * nv, sizeof(float));
  g->pweight = (float *)calloc((size_t)g->nop * nv, sizeof(float));
  g->hes = (float *)calloc((size_t)g->nop * 3, sizeof(float));
  g->cnt = (int *)calloc((size_t)g->nop, sizeof(int));
  for (int x = 0; x < g->nopw; ++x) for (int y = 0; y < g->noph; ++y) {
    int i = x * g->noph + y;                                       /* :66 column-major ids */
    g->pt_ref[2 * i] = (float)(x * g->steps + offw);
    g->pt_ref[2 * i + 1] = (float)(y * g->steps + offh);
  }
  return g;
}

void dis_grid_free(dis_grid *g)
{
  if (!g) return;
  free(g->pt_ref); free(g->p_init); free(g->p_iter); free(g->tmpl); free(g->tdx); free(g->tdy);
  free(g->pweight); free(g->hes); free(g->cnt); free(g);
}

/* patchgrid.cpp:98-116 -> patch.cpp:57-69 InitializePatch; :287-332 getPatchStaticNNGrad; :71-88 ComputeHessian */
void dis_grid_init(dis_grid *g, const dis_params *p, const float *I0, const float *I0x, const float *I0y)
{
  const int ps = g->ps, noc = g->noc, nv = ps * ps * noc;
  float *scr = (float *)malloc(sizeof(float) * nv);
  for (int ip = 0; ip < g->nop; ++ip) {
    float *T = g->tmpl + (size_t)ip * nv, *Tx = g->tdx + (size_t)ip * nv, *Ty = g->tdy + (size_t)ip * nv;
    int px = (int)round(g->pt_ref[2 * ip]) + g->pad;
    int py = (int)round(g->pt_ref[2 * ip + 1]) + g->pad;
    int lb = -ps / 2, ub = ps / 2 - 1, k = 0;
    for (int j = lb; j <= ub; ++j) for (int i = lb; i <= ub; ++i) {
      size_t idx = ((size_t)(px + i) + (size_t)(py + j) * g->tmp_w) * noc;
      for (int c = 0; c < noc; ++c, ++k) { T[k] = I0[idx + c]; Tx[k] = I0x[idx + c]; Ty[k] = I0y[idx + c]; }
    }
    if (p->patnorm > 0) {                                           /* patch.cpp:330-331 */
      float m = dis_sum(T, nv, noc) / nv;
      for (int e = 0; e < nv; ++e) T[e] -= m;
    }
    float h00 = dis_dot(Tx, Tx, nv, noc, scr);                      /* patch.cpp:74-77 */
    float h01 = 0.0f, h11 = 0.0f;
    if (g->depth) {                                                 /* :83-87: 1x1 Hessian */
      if (h00 == 0) h00 += 1e-10;
    } else {
      h01 = dis_dot(Tx, Ty, nv, noc, scr);
      h11 = dis_dot(Ty, Ty, nv, noc, scr);
      if (h00 * h11 - h01 * h01 == 0) { h00 += 1e-10; h11 += 1e-10; } /* :78-82 (float += double -> float) */
    }
    g->hes[3 * ip] = h00; g->hes[3 * ip + 1] = h01; g->hes[3 * ip + 2] = h11;
    g->p_init[2 * ip] = 0; g->p_init[2 * ip + 1] = 0;               /* patchgrid.cpp:113 */
  }
  free(scr);
}

/* patchgrid.cpp:195-211 */
void dis_grid_init_from_coarser(dis_grid *g, const float *flow_prev)
{
  for (int ip = 0; ip < g->nop; ++ip) {
    int x = (int)floor(g->pt_ref[2 * ip] / 2);
    int y = (int)floor(g->pt_ref[2 * ip + 1] / 2);
    /* (D5) a level of odd size has patches at x = w-1 or y = h-1, whose half coordinate is one past the (w/2) x (h/2)
       array: the reference reads out of bounds there (only reachable with `initflow`, which no reference caller passes --
       between scales the sizes are exact halves).  Clamp to the last row / column. */
    if (x > g->w / 2 - 1) x = g->w / 2 - 1;
    if (y > g->h / 2 - 1) y = g->h / 2 - 1;
    int i = y * (g->w / 2) + x;
    if (g->depth) { g->p_init[2 * ip] = flow_prev[i] * 2; continue; }   /* :207-208 */
    g->p_init[2 * ip] = flow_prev[2 * i] * 2;
    g->p_init[2 * ip + 1] = flow_prev[2 * i + 1] * 2;
  }
}

/* patch.cpp:335-402 getPatchStaticBil (+ mean normalisation :400-401) */
static void patch_bil(const dis_grid *g, const dis_params *p, const float *img, float mx, float my, float *out)
{
  const int ps = g->ps, noc = g->noc, nv = ps * ps * noc;
  int pos0 = (int)ceil(mx + .00001f), pos1 = (int)ceil(my + .00001f);
  int pos2 = (int)floor(mx), pos3 = (int)floor(my);
  float r0 = mx - (float)pos2, r1 = my - (float)pos3;
  float we0 = r0 * r1, we1 = (1 - r0) * r1, we2 = r0 * (1 - r1), we3 = (1 - r0) * (1 - r1);
  pos0 += g->pad; pos1 += g->pad;
  int lb = -ps / 2, ub = ps / 2 - 1, k = 0;
  for (int yy = pos1 + lb; yy <= pos1 + ub; ++yy) {
    for (int xx = pos0 + lb; xx <= pos0 + ub; ++xx) {
      const float *a = img + ((size_t)yy * g->tmp_w + xx) * noc;
      const float *b = a - noc;
      const float *c = img + ((size_t)(yy - 1) * g->tmp_w + xx) * noc;
      const float *d = c - noc;
      for (int ch = 0; ch < noc; ++ch, ++k)
        out[k] = we0 * a[ch] + we1 * b[ch] + we2 * c[ch] + we3 * d[ch];
    }
  }
  if (p->patnorm > 0) {
    float m = dis_sum(out, nv, noc) / nv;
    for (int e = 0; e < nv; ++e) out[e] -= m;
  }
}

/* patchgrid.cpp:134-141 Optimize -> patch.cpp:159-212 OptimizeIter, :120-156 OptimizeStart,
 * :264-284 OptimizeComputeErrImg, :223-261 LossComputeErrorImage (costfct 0 L2, 1 L1, 2 pseudo-Huber) */
void dis_grid_optimize(dis_grid *g, const dis_params *p, const float *I1, float *trace)
{
  const int ps = g->ps, noc = g->noc, nv = ps * ps * noc;
  const float dp_thresh = p->dp_thresh * p->dp_thresh;               /* oflow.cpp:88 */
  const float outlier = (float)ps / 2;                               /* oflow.cpp:82 */
  float *pdiff = (float *)malloc(sizeof(float) * nv), *scr = (float *)malloc(sizeof(float) * nv);
  const int trow = p->max_iter + 1;
  for (int ip = 0; ip < g->nop; ++ip) {
    const float *T = g->tmpl + (size_t)ip * nv, *Tx = g->tdx + (size_t)ip * nv, *Ty = g->tdy + (size_t)ip * nv;
    float *pw = g->pweight + (size_t)ip * nv;
    const float h00 = g->hes[3 * ip], h01 = g->hes[3 * ip + 1], h11 = g->hes[3 * ip + 2];
    const float rx = g->pt_ref[2 * ip], ry = g->pt_ref[2 * ip + 1];
    const float pin0 = g->p_init[2 * ip], pin1 = g->p_init[2 * ip + 1];
    float p0 = pin0, p1 = pin1;
    float ptx = rx + p0, pty = ry + p1;                              /* paramtopt :214-221 */
    const float stx = ptx, sty = pty;
    int conv = 0, cnt = 0;
    float dp0 = 0, dp1 = 0, dpn = 1e-10f, dpn_init = 1e-10f, mares = 1e20f, mares_old = 1e20f;
    if (trace) for (int t = 0; t < trow * 4; ++t) trace[((size_t)ip * trow) * 4 + t] = 0;
    if (ptx < g->lb || pty < g->lb || ptx > g->ubw || pty > g->ubh) {   /* :135-141 */
      conv = 1;
      for (int e = 0; e < nv; ++e) pw[e] = 0.0f;                     /* (D2) */
    } else {
      mares = 1e5f; mares_old = 1e20f;
      goto compute_err;
    }
    while (!conv) {
      cnt++;
      dp0 = dis_dot(Tx, pdiff, nv, noc, scr);                        /* :178-179 */
      if (g->depth) {                                                /* :181, :184 with the 1x1 Hessian: L = sqrt(H) */
        float l00 = sqrtf(h00);
        float y0 = dp0 / l00;
        dp0 = y0 / l00; dp1 = 0.0f;
      } else {
      dp1 = dis_dot(Ty, pdiff, nv, noc, scr);
      {                                                              /* :184 Hes.llt().solve() */
        float l00 = sqrtf(h00);
        float l10 = h01 / l00;
        float l11 = sqrtf(h11 - l10 * l10);
        float y0 = dp0 / l00;
        float y1 = (dp1 - l10 * y0) / l11;
        float x1 = y1 / l11;
        float x0 = (y0 - l10 * x1) / l00;
        dp0 = x0; dp1 = x1;
      }
      }
      p0 -= dp0; p1 -= dp1;                                          /* :186 */
      if (g->depth) {                                                /* :188-193: std::min / std::max with 0 */
        if (g->camlr == 0) p0 = (0.0f < p0) ? 0.0f : p0;
        else p0 = (p0 < 0.0f) ? 0.0f : p0;
      }
      ptx = rx + p0; pty = ry + p1;
      {
        float ddx = stx - ptx, ddy = sty - pty;
        int bad = !(isfinite(dp0) && isfinite(dp1));                 /* (D3) */
        if (bad || sqrtf(ddx * ddx + ddy * ddy) > outlier ||         /* :199-208 */
            ptx < g->lb || pty < g->lb || ptx > g->ubw || pty > g->ubh) {
          p0 = pin0; p1 = pin1; ptx = rx + p0; pty = ry + p1;
          conv = 1;
          if (bad) { dp0 = 0; dp1 = 0; }
        }
      }
    compute_err:
      patch_bil(g, p, I1, ptx, pty, pdiff);                          /* :266 */
      for (int e = 0; e < nv; ++e) {
        float d = pdiff[e] - T[e];                                   /* :230-236 L2: the difference image itself */
        if (p->costfct == 1) d = copysignf(sqrtf(fabsf(d)), d);      /* :238-246 L1: sign(d) * sqrt(|d|) */
        else if (p->costfct == 2) {                                  /* :247-261 pseudo-Huber, b = normoutlier (oflow.cpp:106-107) */
          const float bsq = p->normoutlier * p->normoutlier, bsq2 = bsq * 2.0f;
          d = copysignf(sqrtf((sqrtf(1.0f + (d * d) / bsq) - 1.0f) * bsq2), d);
        }
        pdiff[e] = d; pw[e] = fabsf(d);
      }
      dpn = dp0 * dp0 + dp1 * dp1;                                   /* :272 */
      if (cnt == 1) dpn_init = dpn;
      mares_old = mares;
      mares = dis_sum(pw, nv, noc) / nv;                             /* :278 */
      if (!((cnt < p->max_iter) & (mares > p->res_thresh) &
            ((cnt < p->min_iter) | (dpn / dpn_init >= dp_thresh)) &
            ((cnt < p->min_iter) | (mares / mares_old <= p->dr_thresh))))
        conv = 1;
      if (trace && cnt <= p->max_iter) {
        float *tr = trace + ((size_t)ip * trow + cnt) * 4;
        tr[0] = p0; tr[1] = p1; tr[2] = mares; tr[3] = (float)cnt;
      }
    }
    g->p_iter[2 * ip] = p0; g->p_iter[2 * ip + 1] = p1; g->cnt[ip] = cnt;
  }
  free(pdiff); free(scr);
}

/* patchgrid.cpp:213-275 + :377-397 (usefbcon = 0) */
void dis_grid_aggregate(const dis_grid *g, const dis_params *p, float *flowout)
{
  dis_grid_aggregate_fb(g, NULL, p, flowout);
}

void dis_grid_aggregate_fb(const dis_grid *g, const dis_grid *cg, const dis_params *p, float *flowout)
{
  const int ps = g->ps, noc = g->noc, nv = ps * ps * noc, w = g->w, h = g->h;
  const float minerr = 2.0f;                                         /* oflow.h:62 */
  const int np = g->depth ? 1 : 2;                                   /* op->nop (oflow.cpp:76-80) */
  float *we = (float *)calloc((size_t)w * h, sizeof(float));
  memset(flowout, 0, sizeof(float) * np * (size_t)w * h);
  (void)p;
  for (int ip = 0; ip < g->nop; ++ip) {
    const float f0 = g->p_iter[2 * ip], f1 = g->p_iter[2 * ip + 1];
    const float *pw = g->pweight + (size_t)ip * nv;
    int lb = -ps / 2, ub = ps / 2 - 1;
    for (int y = lb; y <= ub; ++y) for (int x = lb; x <= ub; ++x, ++pw) {
      int yt = (int)(y + g->pt_ref[2 * ip + 1]);
      int xt = (int)(x + g->pt_ref[2 * ip]);
      if (xt >= 0 && yt >= 0 && xt < w && yt < h) {
        int i = yt * w + xt;
        float absw;
        if (noc == 1) absw = 1.0f / (float)(*pw > minerr ? *pw : minerr);
        else {
          /* :253-258: the pointer advances 3 per in-image pixel but only 1 per skipped pixel
             (reference behaviour, kept as is) */
          absw = (float)(*pw > minerr ? *pw : minerr); ++pw;
          absw += (float)(*pw > minerr ? *pw : minerr); ++pw;
          absw += (float)(*pw > minerr ? *pw : minerr);
          absw = 1.0f / absw;
        }
        we[i] += absw;
        if (np == 1) { flowout[i] += f0 * absw; continue; }          /* :268 */
        flowout[2 * i] += f0 * absw;
        flowout[2 * i + 1] += f1 * absw;
      }
    }
  }
  /* patchgrid.cpp:278-375: the complementary grid's patches, at their position after optimisation (pt_iter), with
     bilinear weights and reversed flow; same serial order (patch id, window row, window column, taps cc fc cf ff) */
  for (int ip = 0; cg && ip < cg->nop; ++ip) {
    const float f0 = cg->p_iter[2 * ip], f1 = cg->p_iter[2 * ip + 1];
    const float *pw = cg->pweight + (size_t)ip * nv;
    const float rx = cg->pt_ref[2 * ip] + f0, ry = cg->pt_ref[2 * ip + 1] + f1;      /* GetPointPos() = pt_iter (patch.cpp:214-221) */
    const int pos0 = (int)ceil(rx + .00001), pos1 = (int)ceil(ry + .00001);        /* :302-305 (double arithmetic) */
    const int pos2 = (int)floorf(rx), pos3 = (int)floorf(ry);
    const float r0 = rx - pos2, r1 = ry - pos3;
    const float wb0 = r0 * r1, wb1 = (1 - r0) * r1, wb2 = r0 * (1 - r1), wb3 = (1 - r0) * (1 - r1);
    int lb = -ps / 2, ub = ps / 2 - 1;
    for (int y = lb; y <= ub; ++y) for (int x = lb; x <= ub; ++x, ++pw) {
      int yt = y + pos1, xt = x + pos0;
      if (xt >= 1 && yt >= 1 && xt < (w - 1) && yt < (h - 1)) {
        float absw;
        if (noc == 1) absw = 1.0f / (float)(*pw > minerr ? *pw : minerr);
        else {
          absw = (float)(*pw > minerr ? *pw : minerr); ++pw;
          absw += (float)(*pw > minerr ? *pw : minerr); ++pw;
          absw += (float)(*pw > minerr ? *pw : minerr);
          absw = 1.0f / absw;
        }
        const float n0 = f0 * absw, n1 = f1 * absw;
        const int cc = xt + yt * w, fc = (xt - 1) + yt * w, cf = xt + (yt - 1) * w, ff = (xt - 1) + (yt - 1) * w;
        we[cc] += wb0 * absw; we[fc] += wb1 * absw; we[cf] += wb2 * absw; we[ff] += wb3 * absw;
        if (np == 1) {                                               /* :365-368 */
          flowout[cc] -= wb0 * n0; flowout[fc] -= wb1 * n0; flowout[cf] -= wb2 * n0; flowout[ff] -= wb3 * n0;
          continue;
        }
        flowout[2 * cc] -= wb0 * n0; flowout[2 * cc + 1] -= wb0 * n1;
        flowout[2 * fc] -= wb1 * n0; flowout[2 * fc + 1] -= wb1 * n1;
        flowout[2 * cf] -= wb2 * n0; flowout[2 * cf + 1] -= wb2 * n1;
        flowout[2 * ff] -= wb3 * n0; flowout[2 * ff + 1] -= wb3 * n1;
      }
    }
  }
  for (int i = 0; i < w * h; ++i) if (we[i] > 0) {
    if (np == 1) flowout[i] /= we[i];
    else { flowout[2 * i] /= we[i]; flowout[2 * i + 1] /= we[i]; }
  }
  free(we);
}

/* ------------------------------------------------------------------------------------------- */
/* FDF1.0.1 stages                                                                             */
/* ------------------------------------------------------------------------------------------- */

int dis_stride(int w) { return ((w + 3) / 4) * 4; }                  /* image.c:22 */

static int clampi(int v, int n) { return v < 0 ? 0 : (v > n - 1 ? n - 1 : v); }

/* FDF1.0.1/opticalflow_aux.c:18-60 */
void dis_image_warp(float *dst, float *mask, const float *src, const float *wx, const float *wy,
                    int w, int h, int noc)
{
  const int st = dis_stride(w);
  for (int j = 0; j < h; ++j) for (int i = 0; i < w; ++i) {
    const int o = j * st + i;
    float xx = i + wx[o], yy = j + wy[o];
    int x = (int)floor(xx), y = (int)floor(yy);
    float dx = xx - x, dy = yy - y;
    mask[o] = (xx >= 0 && xx <= w - 1 && yy >= 0 && yy <= h - 1);
    int x1 = clampi(x, w), x2 = clampi(x + 1, w), y1 = clampi(y, h), y2 = clampi(y + 1, h);
    for (int c = 0; c < noc; ++c) {
      const float *s = src + (size_t)c * st * h;
      dst[(size_t)c * st * h + o] =
          s[y1 * st + x1] * (1.0f - dx) * (1.0f - dy) +
          s[y1 * st + x2] * dx * (1.0f - dy) +
          s[y2 * st + x1] * (1.0f - dx) * dy +
          s[y2 * st + x2] * dx * dy;
    }
  }
}

/* convolution_new(2,{0,-8/12,1/12},0): image.c:326-349 -> coeffs {1/12,-8/12,-0,8/12,-1/12}
 * (refine_variational.cpp:45-46) */
static void deriv5(float c[5])
{
  const float h1 = -8.0f / 12.0f, h2 = 1.0f / 12.0f;
  c[0] = h2; c[1] = h1; c[2] = -0.0f; c[3] = -h1; c[4] = -h2;
}
/* convolve_horiz_fast_5 image.c:466-502 (replicate borders) */
static void conv_h5(float *dst, const float *src, int w, int h)
{
  const int st = dis_stride(w); float c[5]; deriv5(c);
  for (int j = 0; j < h; ++j) for (int i = 0; i < w; ++i) {
    const float *r = src + j * st;
    dst[j * st + i] = c[0] * r[clampi(i - 2, w)] + c[1] * r[clampi(i - 1, w)] + c[2] * r[i] +
                      c[3] * r[clampi(i + 1, w)] + c[4] * r[clampi(i + 2, w)];
  }
}
/* convolve_vert_fast_5 image.c:401-434 (border rows use summed coefficients) */
static void conv_v5(float *dst, const float *src, int w, int h)
{
  const int st = dis_stride(w); float c[5]; deriv5(c);
  for (int j = 0; j < h; ++j) for (int i = 0; i < w; ++i) {
    const float *s = src + i; float v;
#define S(r) s[(r) * st]
    if (j == 0) v = (c[0] + c[1] + c[2]) * S(0) + c[3] * S(1) + c[4] * S(2);
    else if (j == 1) v = (c[0] + c[1]) * S(0) + c[2] * S(1) + c[3] * S(2) + c[4] * S(3);
    else if (j == h - 2) v = c[0] * S(j - 2) + c[1] * S(j - 1) + c[2] * S(j) + (c[3] + c[4]) * S(j + 1);
    else if (j == h - 1) v = c[0] * S(j - 2) + c[1] * S(j - 1) + (c[2] + c[3] + c[4]) * S(j);
    else v = c[0] * S(j - 2) + c[1] * S(j - 1) + c[2] * S(j) + c[3] * S(j + 1) + c[4] * S(j + 2);
#undef S
    dst[j * st + i] = v;
  }
}
/* convolution_new(1,{0,-0.5},0) -> {-0.5,-0,0.5}; convolve_horiz_fast_3 image.c:436-464 */
static void conv_h3(float *dst, const float *src, int w, int h)
{
  const int st = dis_stride(w); const float c0 = -0.5f, c1 = -0.0f, c2 = 0.5f;
  for (int j = 0; j < h; ++j) for (int i = 0; i < w; ++i) {
    const float *r = src + j * st;
    dst[j * st + i] = c0 * r[clampi(i - 1, w)] + c1 * r[i] + c2 * r[clampi(i + 1, w)];
  }
}
/* convolve_vert_fast_3 image.c:376-399 */
static void conv_v3(float *dst, const float *src, int w, int h)
{
  const int st = dis_stride(w); const float c0 = -0.5f, c1 = -0.0f, c2 = 0.5f;
  for (int j = 0; j < h; ++j) for (int i = 0; i < w; ++i) {
    const float *s = src + i; float v;
    if (j == 0) v = (c0 + c1) * s[0] + c2 * s[st];
    else if (j == h - 1) v = c0 * s[(j - 1) * st] + (c1 + c2) * s[j * st];
    else v = c0 * s[(j - 1) * st] + c1 * s[j * st] + c2 * s[(j + 1) * st];
    dst[j * st + i] = v;
  }
}

/* FDF1.0.1/opticalflow_aux.c:65-116 */
void dis_get_derivatives(const float *im1, const float *im2w, int w, int h, int noc,
                         float *Ix, float *Iy, float *Iz, float *Ixx, float *Ixy, float *Iyy,
                         float *Ixz, float *Iyz)
{
  const int st = dis_stride(w); const size_t pl = (size_t)st * h;
  float *avg = (float *)calloc(pl * noc, sizeof(float));
  for (int c = 0; c < noc; ++c) for (int j = 0; j < h; ++j) for (int i = 0; i < w; ++i) {
    size_t o = c * pl + j * st + i;
    avg[o] = 0.5f * (im2w[o] + im1[o]);
    Iz[o] = im2w[o] - im1[o];
  }
  for (int c = 0; c < noc; ++c) {
    size_t o = c * pl;
    conv_h5(Ix + o, avg + o, w, h);
    conv_v5(Iy + o, avg + o, w, h);
    conv_h5(Ixx + o, Ix + o, w, h);
    conv_v5(Ixy + o, Ix + o, w, h);
    conv_v5(Iyy + o, Iy + o, w, h);
    conv_h5(Ixz + o, Iz + o, w, h);
    conv_v5(Iyz + o, Iz + o, w, h);
  }
  free(avg);
}

/* FDF1.0.1/opticalflow_aux.c:123-165 */
void dis_compute_smoothness(float *horiz, float *vert, const float *uu, const float *vv,
                            int w, int h, float quarter_alpha)
{
  const int st = dis_stride(w); const size_t pl = (size_t)st * h;
  const float eps = 0.001f * 0.001f;
  float *ux = (float *)calloc(pl * 5, sizeof(float)), *vx = ux + pl, *uy = vx + pl, *vy = uy + pl, *s = vy + pl;
  conv_h3(ux, uu, w, h); conv_h3(vx, vv, w, h); conv_v3(uy, uu, w, h); conv_v3(vy, vv, w, h);
  for (int j = 0; j < h; ++j) for (int i = 0; i < w; ++i) {
    int o = j * st + i;
    s[o] = quarter_alpha / sqrtf(ux[o] * ux[o] + uy[o] * uy[o] + vx[o] * vx[o] + vy[o] * vy[o] + eps);
  }
  for (int j = 0; j < h; ++j) for (int i = 0; i < w; ++i) {
    int o = j * st + i;
    horiz[o] = (i < w - 1) ? s[o] + s[o + 1] : 0.0f;
    vert[o] = (j < h - 1) ? s[o] + s[o + st] : 0.0f;
  }
  free(ux);
}

/* FDF1.0.1/opticalflow_aux.c:310-438 */
void dis_compute_data(float *a11, float *a12, float *a22, float *b1, float *b2,
                      const float *mask, const float *du, const float *dv,
                      const float *Ix, const float *Iy, const float *Iz, const float *Ixx,
                      const float *Ixy, const float *Iyy, const float *Ixz, const float *Iyz,
                      int w, int h, int noc, float half_delta_over3, float half_gamma_over3)
{
  const int st = dis_stride(w); const size_t pl = (size_t)st * h;
  const float dnorm = 0.1f * 0.1f, epsc = 0.001f * 0.001f, epsg = 0.001f * 0.001f;
  for (int j = 0; j < h; ++j) for (int i = 0; i < w; ++i) {
    const int o = j * st + i;
    float A11 = 0, A12 = 0, A22 = 0, B1 = 0, B2 = 0;
    const float u = du[o], v = dv[o], m = mask[o];
    if (noc == 1) {
      float tmp, tmp2, n1, n2;
      if (half_delta_over3) {
        tmp = Iz[o] + Ix[o] * u + Iy[o] * v;
        n1 = Ix[o] * Ix[o] + Iy[o] * Iy[o] + dnorm;
        tmp = m * half_delta_over3 / sqrtf(3 * tmp * tmp / n1 + epsc);
        tmp /= n1;
        A11 += tmp * Ix[o] * Ix[o];
        A12 += tmp * Ix[o] * Iy[o];
        A22 += tmp * Iy[o] * Iy[o];
        B1 -= tmp * Iz[o] * Ix[o];
        B2 -= tmp * Iz[o] * Iy[o];
      }
      n1 = Ixx[o] * Ixx[o] + Ixy[o] * Ixy[o] + dnorm;
      n2 = Iyy[o] * Iyy[o] + Ixy[o] * Ixy[o] + dnorm;
      tmp = Ixz[o] + Ixx[o] * u + Ixy[o] * v;
      tmp2 = Iyz[o] + Ixy[o] * u + Iyy[o] * v;
      tmp = m * half_gamma_over3 / sqrtf(3 * tmp * tmp / n1 + 3 * tmp2 * tmp2 / n2 + epsg);
      tmp2 = tmp / n2; tmp /= n1;
      A11 += tmp * Ixx[o] * Ixx[o] + tmp2 * Ixy[o] * Ixy[o];
      A12 += tmp * Ixx[o] * Ixy[o] + tmp2 * Ixy[o] * Iyy[o];
      A22 += tmp2 * Iyy[o] * Iyy[o] + tmp * Ixy[o] * Ixy[o];
      B1 -= tmp * Ixx[o] * Ixz[o] + tmp2 * Ixy[o] * Iyz[o];
      B2 -= tmp2 * Iyy[o] * Iyz[o] + tmp * Ixy[o] * Ixz[o];
      A11 *= 3; A12 *= 3; A22 *= 3; B1 *= 3; B2 *= 3;               /* :420-426 */
    } else {
      const float *ix1 = Ix, *ix2 = Ix + pl, *ix3 = Ix + 2 * pl, *iy1 = Iy, *iy2 = Iy + pl, *iy3 = Iy + 2 * pl;
      const float *iz1 = Iz, *iz2 = Iz + pl, *iz3 = Iz + 2 * pl;
      const float *ixx1 = Ixx, *ixx2 = Ixx + pl, *ixx3 = Ixx + 2 * pl, *ixy1 = Ixy, *ixy2 = Ixy + pl, *ixy3 = Ixy + 2 * pl;
      const float *iyy1 = Iyy, *iyy2 = Iyy + pl, *iyy3 = Iyy + 2 * pl, *ixz1 = Ixz, *ixz2 = Ixz + pl, *ixz3 = Ixz + 2 * pl;
      const float *iyz1 = Iyz, *iyz2 = Iyz + pl, *iyz3 = Iyz + 2 * pl;
      float tmp, tmp2, tmp3, tmp4, tmp5, tmp6, n1, n2, n3, n4, n5, n6;
      if (half_delta_over3) {
        tmp = iz1[o] + ix1[o] * u + iy1[o] * v;  n1 = ix1[o] * ix1[o] + iy1[o] * iy1[o] + dnorm;
        tmp2 = iz2[o] + ix2[o] * u + iy2[o] * v; n2 = ix2[o] * ix2[o] + iy2[o] * iy2[o] + dnorm;
        tmp3 = iz3[o] + ix3[o] * u + iy3[o] * v; n3 = ix3[o] * ix3[o] + iy3[o] * iy3[o] + dnorm;
        tmp = m * half_delta_over3 / sqrtf(tmp * tmp / n1 + tmp2 * tmp2 / n2 + tmp3 * tmp3 / n3 + epsc);
        tmp3 = tmp / n3; tmp2 = tmp / n2; tmp /= n1;
        A11 += tmp * ix1[o] * ix1[o]; A12 += tmp * ix1[o] * iy1[o]; A22 += tmp * iy1[o] * iy1[o];
        B1 -= tmp * iz1[o] * ix1[o];  B2 -= tmp * iz1[o] * iy1[o];
        A11 += tmp2 * ix2[o] * ix2[o]; A12 += tmp2 * ix2[o] * iy2[o]; A22 += tmp2 * iy2[o] * iy2[o];
        B1 -= tmp2 * iz2[o] * ix2[o];  B2 -= tmp2 * iz2[o] * iy2[o];
        A11 += tmp3 * ix3[o] * ix3[o]; A12 += tmp3 * ix3[o] * iy3[o]; A22 += tmp3 * iy3[o] * iy3[o];
        B1 -= tmp3 * iz3[o] * ix3[o];  B2 -= tmp3 * iz3[o] * iy3[o];
      }
      n1 = ixx1[o] * ixx1[o] + ixy1[o] * ixy1[o] + dnorm; n2 = iyy1[o] * iyy1[o] + ixy1[o] * ixy1[o] + dnorm;
      tmp = ixz1[o] + ixx1[o] * u + ixy1[o] * v;          tmp2 = iyz1[o] + ixy1[o] * u + iyy1[o] * v;
      n3 = ixx2[o] * ixx2[o] + ixy2[o] * ixy2[o] + dnorm; n4 = iyy2[o] * iyy2[o] + ixy2[o] * ixy2[o] + dnorm;
      tmp3 = ixz2[o] + ixx2[o] * u + ixy2[o] * v;         tmp4 = iyz2[o] + ixy2[o] * u + iyy2[o] * v;
      n5 = ixx3[o] * ixx3[o] + ixy3[o] * ixy3[o] + dnorm; n6 = iyy3[o] * iyy3[o] + ixy3[o] * ixy3[o] + dnorm;
      tmp5 = ixz3[o] + ixx3[o] * u + ixy3[o] * v;         tmp6 = iyz3[o] + ixy3[o] * u + iyy3[o] * v;
      tmp = m * half_gamma_over3 / sqrtf(tmp * tmp / n1 + tmp2 * tmp2 / n2 + tmp3 * tmp3 / n3 +
                                         tmp4 * tmp4 / n4 + tmp5 * tmp5 / n5 + tmp6 * tmp6 / n6 + epsg);
      tmp6 = tmp / n6; tmp5 = tmp / n5; tmp4 = tmp / n4; tmp3 = tmp / n3; tmp2 = tmp / n2; tmp /= n1;
      A11 += tmp * ixx1[o] * ixx1[o] + tmp2 * ixy1[o] * ixy1[o];
      A12 += tmp * ixx1[o] * ixy1[o] + tmp2 * ixy1[o] * iyy1[o];
      A22 += tmp2 * iyy1[o] * iyy1[o] + tmp * ixy1[o] * ixy1[o];
      B1 -= tmp * ixx1[o] * ixz1[o] + tmp2 * ixy1[o] * iyz1[o];
      B2 -= tmp2 * iyy1[o] * iyz1[o] + tmp * ixy1[o] * ixz1[o];
      A11 += tmp3 * ixx2[o] * ixx2[o] + tmp4 * ixy2[o] * ixy2[o];
      A12 += tmp3 * ixx2[o] * ixy2[o] + tmp4 * ixy2[o] * iyy2[o];
      A22 += tmp4 * iyy2[o] * iyy2[o] + tmp3 * ixy2[o] * ixy2[o];
      B1 -= tmp3 * ixx2[o] * ixz2[o] + tmp4 * ixy2[o] * iyz2[o];
      B2 -= tmp4 * iyy2[o] * iyz2[o] + tmp3 * ixy2[o] * ixz2[o];
      A11 += tmp5 * ixx3[o] * ixx3[o] + tmp6 * ixy3[o] * ixy3[o];
      A12 += tmp5 * ixx3[o] * ixy3[o] + tmp6 * ixy3[o] * iyy3[o];
      A22 += tmp6 * iyy3[o] * iyy3[o] + tmp5 * ixy3[o] * ixy3[o];
      B1 -= tmp5 * ixx3[o] * ixz3[o] + tmp6 * ixy3[o] * iyz3[o];
      B2 -= tmp6 * iyy3[o] * iyz3[o] + tmp5 * ixy3[o] * ixz3[o];
    }
    a11[o] = A11; a12[o] = A12; a22[o] = A22; b1[o] = B1; b2[o] = B2;
  }
}

/* FDF1.0.1/opticalflow_aux.c:172-199.  Scatter form of the reference turned into its per-pixel
 * gather with the same order of the four updates: -left, +right (horizontal pass, :177-190),
 * then -top, +bottom (vertical pass, :192-198). */
void dis_sub_laplacian(float *dst, const float *src, const float *horiz, const float *vert, int w, int h)
{
  const int st = dis_stride(w);
  for (int j = 0; j < h; ++j) for (int i = 0; i < w; ++i) {
    const int o = j * st + i;
    float v = dst[o];
    if (i > 0)     v -= horiz[o - 1] * (src[o] - src[o - 1]);
    if (i < w - 1) v += horiz[o] * (src[o + 1] - src[o]);
    if (j > 0)     v -= vert[o - st] * (src[o] - src[o - st]);
    if (j < h - 1) v += vert[o] * (src[o + st] - src[o]);
    dst[o] = v;
  }
}

/* FDF1.0.1/solver.c:77-421 sor_coupled: lexicographic Gauss-Seidel with exact 2x2 block inverse.
 * First sweep inverts the blocks in place (:115-120, note a11/a22 swap = adjugate). */
static void sor_invert_blocks(float *a11, float *a12, float *a22, const float *horiz, const float *vert, int w, int h)
{
  const int st = dis_stride(w);
  for (int j = 0; j < h; ++j) for (int i = 0; i < w; ++i) {
    const int o = j * st + i;
    const float hl = i > 0 ? horiz[o - 1] : 0.0f, hr = horiz[o];
    float dps = hl + hr;                                              /* (*hpl)+(*hp) */
    if (j > 0) dps = dps + vert[o - st];                              /* +(*vpt) */
    if (j < h - 1) dps = dps + vert[o];                               /* +(*vp) */
    const float A11 = a22[o] + dps, A22 = a11[o] + dps;
    const float det = A11 * A22 - a12[o] * a12[o];
    a11[o] = A11 / det; a22[o] = A22 / det; a12[o] = a12[o] / -det;
  }
}

static inline void sor_pixel(float *du, float *dv, const float *a11, const float *a12, const float *a22,
                             const float *b1, const float *b2, const float *horiz, const float *vert,
                             int i, int j, int w, int h, int st, float omega)
{
  const int o = j * st + i;
  const float hr = horiz[o];
  const float dur = (i < w - 1) ? du[o + 1] : 0.0f, dvr = (i < w - 1) ? dv[o + 1] : 0.0f;
  float s1 = hr * dur, s2 = hr * dvr;                                 /* (*hp)*(*dur) */
  if (j > 0) { s1 = s1 + vert[o - st] * du[o - st]; s2 = s2 + vert[o - st] * dv[o - st]; }
  if (j < h - 1) { s1 = s1 + vert[o] * du[o + st]; s2 = s2 + vert[o] * dv[o + st]; }
  s1 = s1 + b1[o]; s2 = s2 + b2[o];
  float B1 = s1, B2 = s2;
  if (i > 0) { B1 = horiz[o - 1] * du[o - 1] + s1; B2 = horiz[o - 1] * dv[o - 1] + s2; }
  du[o] += omega * (a11[o] * B1 + a12[o] * B2 - du[o]);
  dv[o] += omega * (a12[o] * B1 + a22[o] * B2 - dv[o]);
}

void dis_sor_coupled(float *du, float *dv, float *a11, float *a12, float *a22, const float *b1,
                     const float *b2, const float *horiz, const float *vert, int w, int h,
                     int iterations, float omega)
{
  const int st = dis_stride(w);
  if (iterations < 1) return;
  sor_invert_blocks(a11, a12, a22, horiz, vert, w, h);
  for (int it = 0; it < iterations; ++it)
    for (int j = 0; j < h; ++j) for (int i = 0; i < w; ++i)
      sor_pixel(du, dv, a11, a12, a22, b1, b2, horiz, vert, i, j, w, h, st, omega);
}

void dis_sor_coupled_redblack(float *du, float *dv, float *a11, float *a12, float *a22,
                              const float *b1, const float *b2, const float *horiz,
                              const float *vert, int w, int h, int iterations, float omega)
{
  const int st = dis_stride(w);
  if (iterations < 1) return;
  sor_invert_blocks(a11, a12, a22, horiz, vert, w, h);
  for (int it = 0; it < iterations; ++it)
    for (int col = 0; col < 2; ++col)
      for (int j = 0; j < h; ++j) for (int i = 0; i < w; ++i)
        if (((i + j) & 1) == col)
          sor_pixel(du, dv, a11, a12, a22, b1, b2, horiz, vert, i, j, w, h, st, omega);
}

/* FDF1.0.1/solver.c:19-72 sor_coupled_slow_but_readable (SURVEY 8a row a17'): the solver an OpenMP build of the reference
 * selects (refine_variational.cpp:202-206).  Point update -- du from the old dv, then dv from the NEW du, no block inverse --
 * in the row-major order of the serial loop (without OpenMP the `parallel for` over rows is a plain loop; with it the rows race,
 * SURVEY 8a probe).  Neighbour order of the sums: top, left, bottom, right.  a11 / a12 / a22 are NOT modified. */
void dis_sor_coupled_slow(float *du, float *dv, const float *a11, const float *a12, const float *a22, const float *b1,
                          const float *b2, const float *horiz, const float *vert, int w, int h, int iterations, float omega)
{
  const int st = dis_stride(w);
  for (int it = 0; it < iterations; ++it)
    for (int j = 0; j < h; ++j) for (int i = 0; i < w; ++i) {
      const int o = j * st + i;
      float sigma_u = 0.0f, sigma_v = 0.0f, sum_dpsis = 0.0f;
      if (j > 0)     { sigma_u -= vert[o - st] * du[o - st]; sigma_v -= vert[o - st] * dv[o - st]; sum_dpsis += vert[o - st]; }
      if (i > 0)     { sigma_u -= horiz[o - 1] * du[o - 1];  sigma_v -= horiz[o - 1] * dv[o - 1];  sum_dpsis += horiz[o - 1]; }
      if (j < h - 1) { sigma_u -= vert[o] * du[o + st];      sigma_v -= vert[o] * dv[o + st];      sum_dpsis += vert[o]; }
      if (i < w - 1) { sigma_u -= horiz[o] * du[o + 1];      sigma_v -= horiz[o] * dv[o + 1];      sum_dpsis += horiz[o]; }
      const float A11 = a11[o] + sum_dpsis, A12 = a12[o], A22 = a22[o] + sum_dpsis;
      const float B1 = b1[o] - sigma_u, B2 = b2[o] - sigma_v;
      du[o] = (1.0f - omega) * du[o] + omega / A11 * (B1 - A12 * dv[o]);        /* :63 */
      dv[o] = (1.0f - omega) * dv[o] + omega / A22 * (B2 - A12 * du[o]);        /* :64, with the du just written */
    }
}

/* kroeger/refine_variational.cpp:25-116 (ctor), :118-149 copyimage, :153-241 RefLevelOF */
void dis_varref(const float *I0, const float *I1, int w, int h, int lvl, const dis_params *p,
                float *flow, int sor_mode)
{
  const int st = dis_stride(w), noc = p->noc, pad = p->ps, tmp_w = w + 2 * pad;
  const size_t pl = (size_t)st * h;
  const float quarter_alpha = 0.25f * p->tv_alpha;
  const float half_gamma_over3 = p->tv_gamma * 0.5f / 3.0f;
  const float half_delta_over3 = p->tv_delta * 0.5f / 3.0f;
  const int inner = p->tv_innerit * (lvl + 1);
  float *buf = (float *)calloc(pl * (13 + 11 * noc), sizeof(float));
  float *wx = buf, *wy = wx + pl, *du = wy + pl, *dv = du + pl, *mask = dv + pl, *sh = mask + pl,
        *sv = sh + pl, *uu = sv + pl, *vv = uu + pl, *a11 = vv + pl, *a12 = a11 + pl, *a22 = a12 + pl,
        *b1 = a22 + pl;
  float *b2 = (float *)calloc(pl, sizeof(float));
  float *im1 = b1 + pl, *im2 = im1 + pl * noc, *w2 = im2 + pl * noc, *Ix = w2 + pl * noc,
        *Iy = Ix + pl * noc, *Iz = Iy + pl * noc, *Ixx = Iz + pl * noc, *Ixy = Ixx + pl * noc,
        *Iyy = Ixy + pl * noc, *Ixz = Iyy + pl * noc, *Iyz = Ixz + pl * noc;
  for (int j = 0; j < h; ++j) for (int i = 0; i < w; ++i) {
    wx[j * st + i] = flow[2 * (j * w + i)]; wy[j * st + i] = flow[2 * (j * w + i) + 1];
    for (int c = 0; c < noc; ++c) {
      size_t s = ((size_t)(j + pad) * tmp_w + (i + pad)) * noc + c;
      im1[c * pl + j * st + i] = I0[s]; im2[c * pl + j * st + i] = I1[s];
    }
  }
  dis_image_warp(w2, mask, im2, wx, wy, w, h, noc);
  dis_get_derivatives(im1, w2, w, h, noc, Ix, Iy, Iz, Ixx, Ixy, Iyy, Ixz, Iyz);
  memcpy(uu, wx, pl * sizeof(float)); memcpy(vv, wy, pl * sizeof(float));
  for (int it = 0; it < inner; ++it) {
    dis_compute_smoothness(sh, sv, uu, vv, w, h, quarter_alpha);
    dis_compute_data(a11, a12, a22, b1, b2, mask, du, dv, Ix, Iy, Iz, Ixx, Ixy, Iyy, Ixz, Iyz, w, h, noc,
                     half_delta_over3, half_gamma_over3);
    dis_sub_laplacian(b1, wx, sh, sv, w, h);
    dis_sub_laplacian(b2, wy, sh, sv, w, h);
    if (sor_mode == 0) dis_sor_coupled(du, dv, a11, a12, a22, b1, b2, sh, sv, w, h, p->tv_solverit, p->tv_sor);
    else if (sor_mode == 2) dis_sor_coupled_slow(du, dv, a11, a12, a22, b1, b2, sh, sv, w, h, p->tv_solverit, p->tv_sor);
    else dis_sor_coupled_redblack(du, dv, a11, a12, a22, b1, b2, sh, sv, w, h, p->tv_solverit, p->tv_sor);
    for (int j = 0; j < h; ++j) for (int i = 0; i < w; ++i) {
      int o = j * st + i; uu[o] = wx[o] + du[o]; vv[o] = wy[o] + dv[o];
    }
  }
  for (int j = 0; j < h; ++j) for (int i = 0; i < w; ++i) {
    flow[2 * (j * w + i)] = uu[j * st + i]; flow[2 * (j * w + i) + 1] = vv[j * st + i];
  }
  free(buf); free(b2);
}


/* ------------------------------------------------------------------------------------------- */
/* stereo depth (SELECTMODE 2) variants                                                        */
/* ------------------------------------------------------------------------------------------- */

/* FDF1.0.1/opticalflow_aux.c:446-540 compute_data_DE: only the horizontal increment du enters */
void dis_compute_data_de(float *a11, float *b1, const float *mask, const float *du,
                         const float *Ix, const float *Iy, const float *Iz, const float *Ixx,
                         const float *Ixy, const float *Iyy, const float *Ixz, const float *Iyz,
                         int w, int h, int noc, float half_delta_over3, float half_gamma_over3)
{
  const int st = dis_stride(w); const size_t pl = (size_t)st * h;
  const float dnorm = 0.1f * 0.1f, epsc = 0.001f * 0.001f, epsg = 0.001f * 0.001f;
  for (int j = 0; j < h; ++j) for (int i = 0; i < w; ++i) {
    const int o = j * st + i;
    float A11 = 0, B1 = 0;
    const float u = du[o], m = mask[o];
    if (noc == 1) {
      float tmp, tmp2, n1, n2;
      if (half_delta_over3) {                                          /* :481-506 */
        tmp = Iz[o] + Ix[o] * u;
        n1 = Ix[o] * Ix[o] + Iy[o] * Iy[o] + dnorm;
        tmp = m * half_delta_over3 / sqrtf(3 * tmp * tmp / n1 + epsc);
        tmp /= n1;
        A11 += tmp * Ix[o] * Ix[o];
        B1 -= tmp * Iz[o] * Ix[o];
      }
      n1 = Ixx[o] * Ixx[o] + Ixy[o] * Ixy[o] + dnorm;                  /* :508-511 */
      n2 = Iyy[o] * Iyy[o] + Ixy[o] * Ixy[o] + dnorm;
      tmp = Ixz[o] + Ixx[o] * u;
      tmp2 = Iyz[o] + Ixy[o] * u;
      tmp = m * half_gamma_over3 / sqrtf(3 * tmp * tmp / n1 + 3 * tmp2 * tmp2 / n2 + epsg);   /* :524 */
      tmp2 = tmp / n2; tmp /= n1;
      A11 += tmp * Ixx[o] * Ixx[o] + tmp2 * Ixy[o] * Ixy[o];           /* :527-528 */
      B1 -= tmp * Ixx[o] * Ixz[o] + tmp2 * Ixy[o] * Iyz[o];
      A11 *= 3; B1 *= 3;                                               /* :537-540 */
    } else {
      float t[3], n[3];
      if (half_delta_over3) {
        for (int c = 0; c < 3; ++c) {
          const size_t q = c * pl + o;
          t[c] = Iz[q] + Ix[q] * u;
          n[c] = Ix[q] * Ix[q] + Iy[q] * Iy[q] + dnorm;
        }
        float tmp = m * half_delta_over3 / sqrtf(t[0] * t[0] / n[0] + t[1] * t[1] / n[1] + t[2] * t[2] / n[2] + epsc);   /* :493 */
        const float k3 = tmp / n[2], k2 = tmp / n[1]; tmp /= n[0];
        const float k[3] = {tmp, k2, k3};
        for (int c = 0; c < 3; ++c) {                                  /* :499-506: a11 += ..; b1 -= ..; channel by channel */
          const size_t q = c * pl + o;
          A11 += k[c] * Ix[q] * Ix[q];
          B1 -= k[c] * Iz[q] * Ix[q];
        }
      }
      float n1[3], n2[3], t1[3], t2[3];
      for (int c = 0; c < 3; ++c) {
        const size_t q = c * pl + o;
        n1[c] = Ixx[q] * Ixx[q] + Ixy[q] * Ixy[q] + dnorm; n2[c] = Iyy[q] * Iyy[q] + Ixy[q] * Ixy[q] + dnorm;
        t1[c] = Ixz[q] + Ixx[q] * u;                       t2[c] = Iyz[q] + Ixy[q] * u;
      }
      const float tmp = m * half_gamma_over3 / sqrtf(t1[0] * t1[0] / n1[0] + t2[0] * t2[0] / n2[0] + t1[1] * t1[1] / n1[1] +
                                                     t2[1] * t2[1] / n2[1] + t1[2] * t1[2] / n1[2] + t2[2] * t2[2] / n2[2] + epsg);   /* :521 */
      for (int c = 0; c < 3; ++c) {                                    /* :527-535 */
        const size_t q = c * pl + o;
        const float ka = tmp / n1[c], kb = tmp / n2[c];
        A11 += ka * Ixx[q] * Ixx[q] + kb * Ixy[q] * Ixy[q];
        B1 -= ka * Ixx[q] * Ixz[q] + kb * Ixy[q] * Iyz[q];
      }
    }
    a11[o] = A11; b1[o] = B1;
  }
}

/* FDF1.0.1/solver.c:428-466 sor_coupled_slow_but_readable_DE; the default build has no OpenMP, so the row loop is serial:
 * lexicographic Gauss-Seidel.  Neighbour order of the sums: top, left, bottom, right. */
void dis_sor_de(float *du, const float *a11, const float *b1, const float *horiz, const float *vert,
                int w, int h, int iterations, float omega)
{
  const int st = dis_stride(w);
  for (int it = 0; it < iterations; ++it)
    for (int j = 0; j < h; ++j) for (int i = 0; i < w; ++i) {
      const int o = j * st + i;
      float sigma_u = 0.0f, sum_dpsis = 0.0f;
      if (j > 0)     { sigma_u -= vert[o - st] * du[o - st]; sum_dpsis += vert[o - st]; }
      if (i > 0)     { sigma_u -= horiz[o - 1] * du[o - 1];  sum_dpsis += horiz[o - 1]; }
      if (j < h - 1) { sigma_u -= vert[o] * du[o + st];      sum_dpsis += vert[o]; }
      if (i < w - 1) { sigma_u -= horiz[o] * du[o + 1];      sum_dpsis += horiz[o]; }
      const float A11 = a11[o] + sum_dpsis;
      const float B1 = b1[o] - sigma_u;
      du[o] = (1.0f - omega) * du[o] + omega * (B1 / A11);
    }
}

/* kroeger/refine_variational.cpp:25-116 (ctor with noparam = 1), :243-330 RefLevelDE */
void dis_varref_depth(const float *I0, const float *I1, int w, int h, int lvl, const dis_params *p,
                      float *flow, int camlr)
{
  const int st = dis_stride(w), noc = p->noc, pad = p->ps, tmp_w = w + 2 * pad;
  const size_t pl = (size_t)st * h;
  const float quarter_alpha = 0.25f * p->tv_alpha;
  const float half_gamma_over3 = p->tv_gamma * 0.5f / 3.0f;
  const float half_delta_over3 = p->tv_delta * 0.5f / 3.0f;
  const int inner = p->tv_innerit * (lvl + 1);
  float *buf = (float *)calloc(pl * (9 + 11 * noc), sizeof(float));
  float *wx = buf, *wy0 = wx + pl, *du = wy0 + pl, *mask = du + pl, *sh = mask + pl, *sv = sh + pl, *uu = sv + pl,
        *a11 = uu + pl, *b1 = a11 + pl;
  float *im1 = b1 + pl, *im2 = im1 + pl * noc, *w2 = im2 + pl * noc, *Ix = w2 + pl * noc,
        *Iy = Ix + pl * noc, *Iz = Iy + pl * noc, *Ixx = Iz + pl * noc, *Ixy = Ixx + pl * noc,
        *Iyy = Ixy + pl * noc, *Ixz = Iyy + pl * noc, *Iyz = Ixz + pl * noc;
  for (int j = 0; j < h; ++j) for (int i = 0; i < w; ++i) {
    wx[j * st + i] = flow[j * w + i];                                 /* :58-66, noparam = 1 */
    for (int c = 0; c < noc; ++c) {
      size_t s = ((size_t)(j + pad) * tmp_w + (i + pad)) * noc + c;
      im1[c * pl + j * st + i] = I0[s]; im2[c * pl + j * st + i] = I1[s];
    }
  }
  dis_image_warp(w2, mask, im2, wx, wy0, w, h, noc);                  /* :273, wy_dummy = 0 */
  dis_get_derivatives(im1, w2, w, h, noc, Ix, Iy, Iz, Ixx, Ixy, Iyy, Ixz, Iyz);
  memcpy(uu, wx, pl * sizeof(float));
  for (int it = 0; it < inner; ++it) {
    dis_compute_smoothness(sh, sv, uu, wy0, w, h, quarter_alpha);     /* :288 */
    dis_compute_data_de(a11, b1, mask, du, Ix, Iy, Iz, Ixx, Ixy, Iyy, Ixz, Iyz, w, h, noc, half_delta_over3, half_gamma_over3);
    dis_sub_laplacian(b1, wx, sh, sv, w, h);
    dis_sor_de(du, a11, b1, sh, sv, w, h, p->tv_solverit, p->tv_sor);
    for (int j = 0; j < h; ++j) for (int i = 0; i < w; ++i) {         /* :299-314 minps / maxps with zero */
      const int o = j * st + i;
      const float s = wx[o] + du[o];
      uu[o] = camlr == 0 ? (s < 0.0f ? s : 0.0f) : (s > 0.0f ? s : 0.0f);
    }
  }
  for (int j = 0; j < h; ++j) for (int i = 0; i < w; ++i) flow[j * w + i] = uu[j * st + i];
  free(buf);
}

/* ------------------------------------------------------------------------------------------- */
/* whole flow: kroeger/oflow.cpp:184-337                                                       */
/* ------------------------------------------------------------------------------------------- */

void dis_flow_pyr(const dis_pyramid *P0, const dis_pyramid *P1, const dis_params *p,
                  const float *initflow, float *outflow, int sor_mode, float *level_dump)
{
  const int ns = p->sc_f - p->sc_l + 1, fb = p->usefbcon != 0, np = p->depth ? 1 : 2;
  float **flow = (float **)calloc(ns, sizeof(float *)), **flow_bw = (float **)calloc(ns, sizeof(float *));
  size_t dump_off = 0;
  for (int sl = p->sc_f; sl >= p->sc_l; --sl) {
    const int ii = sl - p->sc_l, w = dis_level_w(P0, sl), h = dis_level_h(P0, sl);
    dis_grid *g = dis_grid_new(w, h, sl, p), *gb = fb ? dis_grid_new(w, h, sl, p) : NULL;   /* oflow.cpp:160-170 */
    flow[ii] = (float *)malloc(sizeof(float) * np * (size_t)w * h);
    if (gb) gb->camlr = 1;                                            /* oflow.cpp:157,165 */
    dis_grid_init(g, p, P0->im[sl], P0->dx[sl], P0->dy[sl]);
    if (fb) { flow_bw[ii] = (float *)malloc(sizeof(float) * np * (size_t)w * h); dis_grid_init(gb, p, P1->im[sl], P1->dx[sl], P1->dy[sl]); }   /* :193-197 */
    if (sl < p->sc_f) { dis_grid_init_from_coarser(g, flow[ii + 1]); if (fb) dis_grid_init_from_coarser(gb, flow_bw[ii + 1]); }   /* :209-216 */
    else if (initflow) dis_grid_init_from_coarser(g, initflow);
    dis_grid_optimize(g, p, P1->im[sl], NULL);
    if (fb) dis_grid_optimize(gb, p, P0->im[sl], NULL);                                   /* :233-235 */
    float *out = (sl == p->sc_l) ? outflow : flow[ii];
    dis_grid_aggregate_fb(g, gb, p, out);
    if (fb && sl > p->sc_l) dis_grid_aggregate_fb(gb, g, p, flow_bw[ii]);                 /* :269-270 */
    if (level_dump) { memcpy(level_dump + dump_off, out, sizeof(float) * np * (size_t)w * h); dump_off += np * (size_t)w * h; }
    if (p->usetvref && p->depth) {                                    /* :287-294 with RefLevelDE */
      dis_varref_depth(P0->im[sl], P1->im[sl], w, h, sl, p, out, 0);
      if (fb && sl > p->sc_l) dis_varref_depth(P1->im[sl], P0->im[sl], w, h, sl, p, flow_bw[ii], 1);
    } else if (p->usetvref) {
      dis_varref(P0->im[sl], P1->im[sl], w, h, sl, p, out, sor_mode);
      if (fb && sl > p->sc_l) dis_varref(P1->im[sl], P0->im[sl], w, h, sl, p, flow_bw[ii], sor_mode);   /* :291-294 */
    }
    if (level_dump) { memcpy(level_dump + dump_off, out, sizeof(float) * np * (size_t)w * h); dump_off += np * (size_t)w * h; }
    dis_grid_free(g);
    if (gb) dis_grid_free(gb);
  }
  for (int i = 0; i < ns; ++i) { free(flow[i]); free(flow_bw[i]); }
  free(flow); free(flow_bw);
}

void dis_flow(const float *I0, const float *I1, int wp, int hp, const dis_params *p, float *outflow, int sor_mode)
{
  dis_pyramid *P0 = dis_pyramid_build(I0, wp, hp, p->noc, p->sc_f, p->ps);
  dis_pyramid *P1 = dis_pyramid_build(I1, wp, hp, p->noc, p->sc_f, p->ps);
  dis_flow_pyr(P0, P1, p, NULL, outflow, sor_mode, NULL);
  dis_pyramid_free(P0); dis_pyramid_free(P1);
}

/* ------------------------------------------------------------------------------------------- */
/* all-cores leg of bench.py's cpu_baseline: frame-parallel over `nthreads` pthreads, one pair per */
/* thread at a time (the CPU analogue of frame-pair sharding).  Every thread first     */
/* runs ONE untimed pair (which also fills its block cache, see dis_alloc), all threads meet at a  */
/* barrier, then they draw pair indices from a shared counter until n_total pairs are done.       */
/* frames: nsrc unpadded w x h x noc pairs (pair k of the run is source pair k % nsrc).           */
/* with_pyramid = 1: padding + both pyramids + flow per pair (what a reference driver does per     */
/* pair); 0: every thread builds the pyramids of its first pair once and times dis_flow_pyr only  */
/* (what the reference prints as O.Flow Run-Time, kroeger/oflow.cpp:355-360).                     */
/* out (optional): flows of the first min(n_total, nsrc) pairs.  Returns the timed seconds.       */
/* ------------------------------------------------------------------------------------------- */
typedef struct {
  const float *I0, *I1; long pair_stride; int nsrc, w, h, wp, hp; const dis_params *p; int n_total, with_pyramid;
  float *out; long out_stride; long next; pthread_barrier_t bar; pthread_mutex_t mu; struct timespec t0, t1; int tid_seq;
} dis_many;

static void dis_many_pair(dis_many *m, int k, int tid, int first, float *pad0, float *pad1, float *flow, const dis_pyramid *P0, const dis_pyramid *P1)
{
  const dis_params *p = m->p;
  /* source pair: k % nsrc with the pyramid in the loop; without it every thread keeps the pyramids of pair tid % nsrc */
  const int src = m->with_pyramid ? ((k % m->nsrc) + m->nsrc) % m->nsrc : tid % m->nsrc;
  if (m->with_pyramid) {
    dis_pad_frame(m->I0 + (size_t)src * m->pair_stride, m->w, m->h, p->noc, p->sc_f, pad0);
    dis_pad_frame(m->I1 + (size_t)src * m->pair_stride, m->w, m->h, p->noc, p->sc_f, pad1);
    dis_flow(pad0, pad1, m->wp, m->hp, p, flow, 0);
  } else {
    dis_flow_pyr(P0, P1, p, NULL, flow, 0, NULL);
  }
  /* every output slot is written exactly once, by one thread: with the pyramid in the loop by pair k < nsrc (= its first
   * occurrence), without it by thread tid < nsrc on its first timed pair (`first`) */
  const int store = m->with_pyramid ? (k >= 0 && k < m->nsrc) : (k >= 0 && tid < m->nsrc && first);
  if (m->out && store && src < m->n_total) memcpy(m->out + (size_t)src * m->out_stride, flow, sizeof(float) * (size_t)m->out_stride);
}

static void *dis_many_worker(void *arg)
{
  dis_many *m = (dis_many *)arg;
  dis_pool pool; memset(&pool, 0, sizeof(pool));
  tl_pool = &pool;
  const dis_params *p = m->p;
  pthread_mutex_lock(&m->mu); const int tid = m->tid_seq++; pthread_mutex_unlock(&m->mu);
  float *pad0 = (float *)malloc(sizeof(float) * (size_t)m->wp * m->hp * p->noc);
  float *pad1 = (float *)malloc(sizeof(float) * (size_t)m->wp * m->hp * p->noc);
  float *flow = (float *)malloc(sizeof(float) * (size_t)m->out_stride);
  dis_pyramid *P0 = NULL, *P1 = NULL;
  if (!m->with_pyramid) {
    const int src = tid % m->nsrc;
    dis_pad_frame(m->I0 + (size_t)src * m->pair_stride, m->w, m->h, p->noc, p->sc_f, pad0);
    dis_pad_frame(m->I1 + (size_t)src * m->pair_stride, m->w, m->h, p->noc, p->sc_f, pad1);
    P0 = dis_pyramid_build(pad0, m->wp, m->hp, p->noc, p->sc_f, p->ps);
    P1 = dis_pyramid_build(pad1, m->wp, m->hp, p->noc, p->sc_f, p->ps);
  }
  dis_many_pair(m, -1 - tid, tid, 0, pad0, pad1, flow, P0, P1);           /* warm-up pair (source pair |k| % nsrc), not stored */
  if (pthread_barrier_wait(&m->bar) == PTHREAD_BARRIER_SERIAL_THREAD) clock_gettime(CLOCK_MONOTONIC, &m->t0);
  pthread_barrier_wait(&m->bar);
  for (int first = 1;; first = 0) {
    const long k = __atomic_fetch_add(&m->next, 1, __ATOMIC_RELAXED);
    if (k >= m->n_total) break;
    dis_many_pair(m, (int)k, tid, first, pad0, pad1, flow, P0, P1);
  }
  if (pthread_barrier_wait(&m->bar) == PTHREAD_BARRIER_SERIAL_THREAD) clock_gettime(CLOCK_MONOTONIC, &m->t1);
  pthread_barrier_wait(&m->bar);
  if (P0) dis_pyramid_free(P0);
  if (P1) dis_pyramid_free(P1);
  free(pad0); free(pad1); free(flow);
  tl_pool = NULL;
  dis_pool_drain(&pool);
  return NULL;
}

double dis_flow_many(const float *I0, const float *I1, long pair_stride, int nsrc, int w, int h, const dis_params *p,
                     int n_total, int nthreads, int with_pyramid, float *out)
{
  if (nsrc < 1 || nthreads < 1 || n_total < 0) return -1.0;
  dis_many m; memset(&m, 0, sizeof(m));
  int padw, padh;
  dis_padded_size(w, h, p->sc_f, &m.wp, &m.hp, &padw, &padh);
  m.I0 = I0; m.I1 = I1; m.pair_stride = pair_stride; m.nsrc = nsrc; m.w = w; m.h = h; m.p = p;
  m.n_total = n_total; m.with_pyramid = with_pyramid; m.out = out;
  m.out_stride = (long)(m.wp >> p->sc_l) * (m.hp >> p->sc_l) * (p->depth ? 1 : 2);
  pthread_mutex_init(&m.mu, NULL);
  pthread_t *th = (pthread_t *)(malloc)(sizeof(pthread_t) * (size_t)nthreads);
  /* The workers meet at a barrier sized for the threads that really exist: they are created holding the mutex (their first
   * action is to take it), the barrier is initialised for the number that could be created, then they are let go.  Fewer
   * threads than asked for is reported through the return value (-1), never by killing the host process. */
  pthread_mutex_lock(&m.mu);
  int started = 0;
  for (; started < nthreads; ++started) if (pthread_create(&th[started], NULL, dis_many_worker, &m) != 0) break;
  if (started < nthreads) {
    fprintf(stderr, "dis_flow_many: only %d of %d threads could be created\n", started, nthreads);
    m.n_total = 0;                                                     /* the threads that exist run their warm-up pair and leave */
  }
  if (started > 0) pthread_barrier_init(&m.bar, NULL, (unsigned)started);
  pthread_mutex_unlock(&m.mu);
  for (int i = 0; i < started; ++i) pthread_join(th[i], NULL);
  if (started < nthreads) {
    (free)(th);
    if (started > 0) pthread_barrier_destroy(&m.bar);
    pthread_mutex_destroy(&m.mu);
    return -1.0;
  }
  (free)(th);
  pthread_barrier_destroy(&m.bar); pthread_mutex_destroy(&m.mu);
  return (double)(m.t1.tv_sec - m.t0.tv_sec) + 1e-9 * (double)(m.t1.tv_nsec - m.t0.tv_nsec);
}

/* kroeger/run_dense.cpp:407-414.  cv::resize(INTER_LINEAR) upscaling: source coordinate
 * (d+0.5)/s-0.5, clamped (fx=0 at the borders), horizontal pass then vertical pass. */
void dis_upsample_crop(const float *flow, int wl, int hl, int sc_l, int padw, int padh,
                       int w_org, int h_org, float *out)
{
  dis_upsample_crop_n(flow, wl, hl, sc_l, padw, padh, w_org, h_org, 2, out);
}

void dis_upsample_crop_n(const float *flow, int wl, int hl, int sc_l, int padw, int padh,
                         int w_org, int h_org, int nch, float *out)
{
  const int s = 1 << sc_l, W = wl * s;
  const int x0 = padw / 2, y0 = padh / 2;
  const float scf = (float)s;
  const double scale = 1.0 / (double)s;
  for (int y = 0; y < h_org; ++y) {
    int dy = y + y0;
    float fy = (float)((dy + 0.5) * scale - 0.5);
    int sy = (int)floor(fy); fy -= sy;
    if (sy < 0) { fy = 0; sy = 0; }
    if (sy >= hl - 1) { fy = 0; sy = hl - 1; }
    int sy1 = sy + 1 < hl ? sy + 1 : hl - 1;
    for (int x = 0; x < w_org; ++x) {
      int dx = x + x0;
      float fx = (float)((dx + 0.5) * scale - 0.5);
      int sx = (int)floor(fx); fx -= sx;
      if (sx < 0) { fx = 0; sx = 0; }
      if (sx >= wl - 1) { fx = 0; sx = wl - 1; }
      int sx1 = sx + 1 < wl ? sx + 1 : wl - 1;
      (void)W;
      for (int c = 0; c < nch; ++c) {
        float v00 = flow[nch * (sy * wl + sx) + c], v01 = flow[nch * (sy * wl + sx1) + c];
        float v10 = flow[nch * (sy1 * wl + sx) + c], v11 = flow[nch * (sy1 * wl + sx1) + c];
        if (sc_l != 0) { v00 *= scf; v01 *= scf; v10 *= scf; v11 *= scf; }
        float r0 = v00 * (1.f - fx) + v01 * fx;
        float r1 = v10 * (1.f - fx) + v11 * fx;
        out[nch * ((size_t)y * w_org + x) + c] = r0 * (1.f - fy) + r1 * fy;
      }
    }
  }
}
