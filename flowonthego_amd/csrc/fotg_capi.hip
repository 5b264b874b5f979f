// fotg_capi.hip -- host side of libfotg.so: context, workspace arenas, launch sequencing, C-ABI (include/fotg.h).
// gfx950 only.  One context = one fixed (size, parameters, max_batch) configuration, reusable across calls;
// every per-call state is re-initialised by the kernels (the reference's CUDA port is only correct for the first
// calc(), src/kernels/extract.cu:139-140).  No hipMalloc / sync inside the launch path (graph-capturable).
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <atomic>
#include <mutex>
#include <new>
#include "common.h"
#include "pyramid.hip.h"
#include "lk.hip.h"
#include "lk_fast.hip.h"
#include "densify.hip.h"
#include "varref.hip.h"
#include "varref_tiles.hip.h"
#include "varref_levelpipe.hip.h"
#include "varref_depth.hip.h"
#include "upsample.hip.h"

using namespace fotg;

static thread_local int g_last_hip = 0;

#define HIPCHK(x)                                   \
  do {                                              \
    hipError_t e_ = (x);                            \
    if (e_ != hipSuccess) { g_last_hip = (int)e_; return FOTG_ERR_HIP; } \
  } while (0)
#define LAUNCHCHK() HIPCHK(hipGetLastError())

// every entry point runs on the context's device and leaves the caller's current device as it found it
struct DevGuard {
  int prev = -1;
  bool ok = false;
  explicit DevGuard(int dev)
  {
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess) return;
    if (cur == dev) { ok = true; return; }
    if (hipSetDevice(dev) != hipSuccess) return;
    prev = cur; ok = true;
  }
  ~DevGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};
#define ON_DEVICE(dev)                                                              \
  DevGuard dev_guard_(dev);                                                         \
  if (!dev_guard_.ok) { g_last_hip = (int)hipGetLastError(); return FOTG_ERR_HIP; }

struct GridState {
  const float *I0 = nullptr, *I0x = nullptr, *I0y = nullptr, *I1 = nullptr;
  long stride = 0;
  const float *flow_prev = nullptr;
  float *trace_host = nullptr;
  int camlr = 0;                     // depth mode: camera side of this grid (kroeger/oflow.cpp:153,157)
};

// Switches read ONCE from the environment at fotg_create (tests force the kernel variants through them; nothing under
// fotg_calc_batch touches the environment).  Every variant computes the same bits.
struct FotgTune {
  int vr_path;      // FOTG_VR_PATH: 0 automatic, 1 single-wave global-memory solver only, 2 no fused per-level kernel
  int vr_stream;    // FOTG_VR_STREAM: 0 = resident-D kernel instead of the streaming solver
  int vr_levelpipe; // FOTG_VR_LEVELPIPE: 1 (default) = tall levels run all inner iterations as one pipeline launch (varref_levelpipe.hip.h); 0 = one tile-solver launch per sor_coupled call; + 16 x diagnosis bits
  int lp_max_pairs; // pairs per launch up to which the level pipeline is used (contexts of a pipe: 4 / depth -- with more pairs resident the launch-per-stage path has the higher THROUGHPUT, the level pipeline the lower latency: measured)
  int vr_first_data; // FOTG_VR_FIRST_DATA: 0 = the first inner iteration's data term in a launch of its own (not in the set-up launch)
  int pyr_split;    // FOTG_PYR_SPLIT: launches the base pyramid kernel of a batch is cut into (1 = one launch)
  int lk_shw;       // FOTG_LK_SHW: shared LDS window of a wave's four patches: -1 automatic (RGB patches of 8 x 8 and more), 0 off, 1 on; 2 / 3 with FOTG_TEST_TAPS: on + some / all rows on the global-memory path
  int lk_lpp;       // FOTG_LK_LPP: lanes per patch of the LK kernel: 0 automatic, 8, 16
  int lk_banded;    // FOTG_LK_BANDED: 0 = plain workgroup order for launches of 1..7 pairs (1: XCD-banded, xcd_banded_x)
  int lk_fast_r;    // FOTG_LK_FAST_R: fast_math, gray 8 x 8 / 12 x 12 patches: radius of the staged window (2), 0 = the whole reachable region
  int lk_lpp_min_waves;   // FOTG_LK_LPP_MIN_WAVES: automatic: eight lanes per patch from this many waves per launch on
  int test_taps;    // FOTG_TEST_TAPS: 1 = fotg_ctx_counter(ctx, "inject_stall") is live (tests of the FOTG_ERR_STALL reporting)
};
static int env_int(const char *name, int dflt) { const char *e = getenv(name); return e ? atoi(e) : dflt; }

#ifdef FOTG_DEBUG
#include <map>
#include <mutex>
static std::map<void *, void *> g_dbg_base;      // user pointer -> allocation base (FOTG_DEBUG_GUARD)
static std::mutex g_dbg_mu;
static hipError_t fotg_dbg_free(void *p)
{
  void *base = p;
  { std::lock_guard<std::mutex> g(g_dbg_mu); auto it = g_dbg_base.find(p); if (it != g_dbg_base.end()) { base = it->second; g_dbg_base.erase(it); } }
  return (hipFree)(base);
}
#define hipFree(p) fotg_dbg_free((void *)(p))
#endif
struct fotg_ctx {
#ifdef FOTG_DEBUG
  struct Guard { const char *name; char *begin, *end; size_t size; } guards[256];      // FOTG_DEBUG_GUARD: a 64 KiB pattern behind every allocation
  int nguards;
#endif
  fotg_params p;
  FotgTune tune;
  int w_org, h_org, Wp, Hp, padw, padh, device, max_batch, noc, ps;
  int nch;                           // flow channels: 2, or 1 in stereo depth mode (op.nop, kroeger/oflow.cpp:76-80)
  int base_lv;                       // first level the pyramid materialises: min(sc_l, 4)
  LevelGeom geom[FOTG_MAXLEV];
  float *im[2][FOTG_MAXLEV];         // padded level images  [B][th][tw][noc]
  float *dx0[FOTG_MAXLEV], *dy0[FOTG_MAXLEV];
  long lev_stride[FOTG_MAXLEV];      // floats per pair in a level buffer
  float *flow[FOTG_MAXLEV];          // [B][h][w][2]
  float *p_iter[FOTG_MAXLEV];        // [B][nop][2]
  float *pweight[FOTG_MAXLEV];       // [B][nop][nv]
  // forward-backward merge (usefbcon): frame-1 gradients and the backward grid / flow (kroeger/oflow.cpp:133-170)
  float *dx1[FOTG_MAXLEV], *dy1[FOTG_MAXLEV], *p_iter_bw[FOTG_MAXLEV], *pweight_bw[FOTG_MAXLEV], *flow_bw[FOTG_MAXLEV];
  float *tap_t[FOTG_MAXLEV], *tap_tx[FOTG_MAXLEV], *tap_ty[FOTG_MAXLEV], *tap_hes[FOTG_MAXLEV];
  int *tap_cnt[FOTG_MAXLEV];
  float *trace_dev[FOTG_MAXLEV];
  bool taps;
  float *vr;                         // refinement workspace (planes)
  long vr_pair_stride;
  float4 *vrC[FOTG_MAXLEV];          // skewed system per level (cells outside the image stay zero forever)
  float2 *vrD[FOTG_MAXLEV];          // skewed (du,dv) per level
  VrArgs vra[FOTG_MAXLEV];
  // verbosity (the reference's op.verbosity, src/oflow.cpp:246-365 / kroeger/oflow.cpp:298-360): > 0 makes the flow calls
  // synchronous and prints the reference's timing lines from HIP-event times of the stages
  int verbosity;
  hipEvent_t tev[4 * FOTG_MAXLEV + 4];
  float tt[FOTG_MAXLEV][5];          // last measured ms per level: pconst, pinit, poptim, cflow, tvopt
  float tt_pyr, tt_total;
  float2 *vrX[FOTG_MAXLEV];          // tile pipeline of tall levels (varref_tiles.hip.h): per-sweep skewed arrays
  long x_pair_stride[FOTG_MAXLEV];
  int x_rt[FOTG_MAXLEV];
  int *tileSync;
  long sync_total;                   // ints in tileSync in front of the time-out counter: max(tile_sync_words, level-pipe words) for max_batch pairs
  int tile_nbs;                      // bands per (pair, sweep) in the progress words = the band count of the tallest tiled level (TileArgs::NBS)
  int lp_ntr;                        // tile rows of the data term the level-pipe words are sized for (the tallest tile level)
  // a bounded inter-workgroup wait that gave up (tile solver pipeline) sets this word of pinned host memory from the device;
  // fotg_calc, fotg_pipe_wait(host_wait) and fotg_pipe_sync read it after their synchronisation and return FOTG_ERR_STALL
  int *stall_host, *stall_dev;
  long stalls;                       // host-side count of the times the word was found set
  unsigned long long *stamps;        // -DFOTG_TILE_STATS builds only
  GridState gs[FOTG_MAXLEV];
};

static void fill_geom(const fotg_params &p, int Wp, int Hp, int l, LevelGeom &g)
{
  g.lvl = l;
  g.w = Wp >> l; g.h = Hp >> l;                                    // kroeger/oflow.cpp:144-145
  g.tw = g.w + 2 * p.ps; g.th = g.h + 2 * p.ps;                    // :150-151
  g.st = ((g.w + 3) / 4) * 4;
  int steps = (int)floor(p.ps * (1 - p.patove));                   // :91
  g.steps = steps > 1 ? steps : 1;
  g.lb = -(float)p.ps / 2;                                         // :147
  g.ubw = (float)(g.w + p.ps / 2 - 2);                             // :148
  g.ubh = (float)(g.h + p.ps / 2 - 2);
  g.nopw = (int)ceil((float)g.w / (float)g.steps);                 // patchgrid.cpp:43-44
  g.noph = (int)ceil((float)g.h / (float)g.steps);
  g.offw = (int)floor((g.w - (g.nopw - 1) * g.steps) / 2);         // :45-46
  g.offh = (int)floor((g.h - (g.noph - 1) * g.steps) / 2);
  g.nop = g.nopw * g.noph;
}

extern "C" {

const char *fotg_version(void) { return "fotg-mi355x 0.1 (gfx950)"; }
int fotg_last_hip_error(void) { return g_last_hip; }

const char *fotg_strerror(int s)
{
  switch (s) {
    case FOTG_OK: return "ok";
    case FOTG_ERR_ARG: return "invalid argument";
    case FOTG_ERR_HIP: return "HIP runtime error";
    case FOTG_ERR_BATCH: return "batch larger than max_batch";
    case FOTG_ERR_UNSUPPORTED: return "unsupported configuration";
    case FOTG_ERR_STALL: return "a bounded inter-workgroup wait timed out: the flow of this call is not valid";
    default: return "unknown status";
  }
}

/* kroeger/run_dense.cpp:180-183, 225-268 */
int fotg_op_point(int op, int width_org, int channels, fotg_params *p)
{
  if (!p || width_org <= 0 || (channels != 1 && channels != 3)) return FOTG_ERR_ARG;
  memset(p, 0, sizeof(*p));
  p->dp_thresh = 0.05f; p->dr_thresh = 0.95f; p->res_thresh = 0.0f; p->patnorm = 1; p->noc = channels;
  p->tv_alpha = 10.0f; p->tv_gamma = 10.0f; p->tv_delta = 5.0f; p->tv_innerit = 1; p->tv_solverit = 3; p->tv_sor = 1.6f;
  p->costfct = 0; p->normoutlier = 5.0f; p->usefbcon = 0;
  p->sor_mode = FOTG_SOR_LEXICOGRAPHIC;
  int sub;
  switch (op) {
    case 1: p->ps = 8;  p->patove = 0.3f;  sub = 2; p->max_iter = 16;  p->usetvref = 0; break;
    case 3: p->ps = 12; p->patove = 0.75f; sub = 4; p->max_iter = 16;  p->usetvref = 1; break;
    case 4: p->ps = 12; p->patove = 0.75f; sub = 5; p->max_iter = 128; p->usetvref = 1; break;
    case 2:
    default: p->ps = 8; p->patove = 0.4f;  sub = 2; p->max_iter = 12;  p->usetvref = 1; break;
  }
  p->min_iter = p->max_iter;
  const int fratio = 5;
  float v = (2.0f * (float)width_org) / ((float)fratio * (float)p->ps);
  int f = (int)floor(log2(v));
  p->sc_f = f > 0 ? f : 0;
  p->sc_l = p->sc_f - sub > 0 ? p->sc_f - sub : 0;
  return FOTG_OK;
}

int fotg_padded_size(int w, int h, int sc_f, int *wp, int *hp, int *padw, int *padh)
{
  if (w <= 0 || h <= 0 || sc_f < 0 || sc_f >= FOTG_MAXLEV) return FOTG_ERR_ARG;
  int scfct = 1 << sc_f, pw = 0, ph = 0;
  int div = w % scfct; if (div > 0) pw = scfct - div;
  div = h % scfct;     if (div > 0) ph = scfct - div;
  if (wp) *wp = w + pw; if (hp) *hp = h + ph; if (padw) *padw = pw; if (padh) *padh = ph;
  return FOTG_OK;
}

void fotg_destroy(fotg_ctx *c)
{
  if (!c) return;
#ifdef FOTG_DEBUG
  if (c->nguards > 0) (void)fotg_ctx_counter(c, "guard_violations");      // (prints what it finds)
#endif
  for (int l = 0; l < FOTG_MAXLEV; ++l) {
    (void)hipFree(c->im[0][l]); (void)hipFree(c->im[1][l]); (void)hipFree(c->dx0[l]); (void)hipFree(c->dy0[l]);
    (void)hipFree(c->flow[l]); (void)hipFree(c->p_iter[l]); (void)hipFree(c->pweight[l]);
    (void)hipFree(c->dx1[l]); (void)hipFree(c->dy1[l]); (void)hipFree(c->p_iter_bw[l]); (void)hipFree(c->pweight_bw[l]); (void)hipFree(c->flow_bw[l]);
    (void)hipFree(c->tap_t[l]); (void)hipFree(c->tap_tx[l]); (void)hipFree(c->tap_ty[l]); (void)hipFree(c->tap_hes[l]); (void)hipFree(c->tap_cnt[l]);
    (void)hipFree(c->trace_dev[l]);
    (void)hipFree(c->vrC[l]); (void)hipFree(c->vrD[l]); (void)hipFree(c->vrX[l]);
  }
  if (c->stall_host) (void)hipHostFree(c->stall_host);
  (void)hipFree(c->vr); (void)hipFree(c->tileSync); (void)hipFree(c->stamps);
  for (auto &e : c->tev) if (e) (void)hipEventDestroy(e);
  delete c;
}

int fotg_create(const fotg_params *p, int w_org, int h_org, int device, int max_batch, fotg_ctx **out)
{
  if (!p || !out || w_org <= 0 || h_org <= 0 || max_batch <= 0) return FOTG_ERR_ARG;
  if (p->noc != 1 && p->noc != 3) return FOTG_ERR_ARG;
  if (p->ps != 4 && p->ps != 8 && p->ps != 12 && p->ps != 16) return FOTG_ERR_UNSUPPORTED;   // op-points use 8 and 12 (run_dense.cpp:242-261); 4 and 16 for custom parameter sets
  if (p->sc_l < 0 || p->sc_f < p->sc_l || p->sc_f >= FOTG_MAXLEV) return FOTG_ERR_ARG;
  if (p->max_iter < 0 || p->tv_solverit < 0 || p->tv_innerit < 0) return FOTG_ERR_ARG;
  if (p->costfct < 0 || p->costfct > 2 || (p->costfct == 2 && !(p->normoutlier > 0))) return FOTG_ERR_ARG;
  if (p->depth && p->usetvref && p->sor_mode != FOTG_SOR_LEXICOGRAPHIC) return FOTG_ERR_UNSUPPORTED;
  if (p->sor_mode < FOTG_SOR_LEXICOGRAPHIC || p->sor_mode > FOTG_SOR_POINT) return FOTG_ERR_ARG;      // (an unknown value would run the lexicographic arithmetic without its buffers)
  if (p->u8_color < 0 || p->u8_color > 2 || (p->u8_color && p->noc != 1)) return FOTG_ERR_ARG;
  ON_DEVICE(device);
  fotg_ctx *c = new (std::nothrow) fotg_ctx();
  if (!c) return FOTG_ERR_ARG;
  memset((void *)c, 0, sizeof(*c));
  c->p = *p; c->w_org = w_org; c->h_org = h_org; c->device = device; c->max_batch = max_batch;
  c->noc = p->noc; c->ps = p->ps; c->nch = p->depth ? 1 : 2;
  c->tune.vr_path = env_int("FOTG_VR_PATH", 0);
  c->tune.vr_stream = env_int("FOTG_VR_STREAM", 1);
  c->tune.vr_levelpipe = env_int("FOTG_VR_LEVELPIPE", 1);
  c->tune.lp_max_pairs = env_int("FOTG_VR_LEVELPIPE_MAX_PAIRS", 1 << 20);
  c->tune.vr_first_data = env_int("FOTG_VR_FIRST_DATA", 1);
  c->tune.pyr_split = env_int("FOTG_PYR_SPLIT", 1);
  c->tune.test_taps = env_int("FOTG_TEST_TAPS", 0);
  c->tune.lk_shw = env_int("FOTG_LK_SHW", -1);
  c->tune.lk_lpp = env_int("FOTG_LK_LPP", 0);
  c->tune.lk_banded = env_int("FOTG_LK_BANDED", 1);
  c->tune.lk_lpp_min_waves = env_int("FOTG_LK_LPP_MIN_WAVES", 2048);
  c->tune.lk_fast_r = env_int("FOTG_LK_FAST_R", 2);
  if (hipHostMalloc((void **)&c->stall_host, 64, hipHostMallocMapped) != hipSuccess ||
      hipHostGetDevicePointer((void **)&c->stall_dev, c->stall_host, 0) != hipSuccess) { g_last_hip = (int)hipGetLastError(); fotg_destroy(c); return FOTG_ERR_HIP; }
  memset(c->stall_host, 0, 64);
  fotg_padded_size(w_org, h_org, p->sc_f, &c->Wp, &c->Hp, &c->padw, &c->padh);
  c->base_lv = p->sc_l < 4 ? p->sc_l : 4;
  const size_t B = (size_t)max_batch;
  for (int l = c->base_lv; l <= p->sc_f; ++l) {
    LevelGeom &g = c->geom[l];
    fill_geom(*p, c->Wp, c->Hp, l, g);
    if (g.w < 1 || g.h < 1) { fotg_destroy(c); return FOTG_ERR_ARG; }
    c->lev_stride[l] = (long)g.tw * g.th * c->noc;
    const size_t bytes = B * c->lev_stride[l] * sizeof(float);
#ifdef FOTG_DEBUG
    // FOTG_DEBUG builds: FOTG_DEBUG_POISON=1 fills every allocation with 0xFF bytes (float NaN) so that a read of memory the engine
    // never wrote shows up as a wrong result in the parity tests instead of depending on what the allocator hands out;
    // FOTG_DEBUG_GUARD=1 puts 64 KiB of 0xA5 behind every allocation, fotg_ctx_counter("guard_violations") counts the guards
    // that no longer hold the pattern (writes past the end of a buffer) and prints which
#define FOTG_GUARD_BYTES 65536
#define ALLOC(ptr, nbytes) do { const bool gd_ = getenv("FOTG_DEBUG_GUARD") != nullptr; const size_t nb_ = ((nbytes) + 255) & ~(size_t)255; char *base_ = nullptr; \
    if (hipMalloc((void **)&base_, nb_ + (gd_ ? 2 * FOTG_GUARD_BYTES : 0)) != hipSuccess) { g_last_hip = (int)hipGetLastError(); fotg_destroy(c); return FOTG_ERR_HIP; } \
    *(void **)&(ptr) = gd_ ? base_ + FOTG_GUARD_BYTES : base_; \
    if (getenv("FOTG_DEBUG_POISON")) (void)hipMemset((ptr), getenv("FOTG_DEBUG_POISON")[0] == '2' ? 0x4B : 0xFF, nb_); \
    if (gd_ && c->nguards < 256) { (void)hipMemset(base_, 0xA5, FOTG_GUARD_BYTES); (void)hipMemset(base_ + FOTG_GUARD_BYTES + nb_, 0xA5, FOTG_GUARD_BYTES); \
      { std::lock_guard<std::mutex> g_(g_dbg_mu); g_dbg_base[(void *)(ptr)] = base_; } \
      c->guards[c->nguards].name = #ptr; c->guards[c->nguards].begin = base_; c->guards[c->nguards].end = base_ + FOTG_GUARD_BYTES + nb_; c->guards[c->nguards++].size = nb_; } } while (0)
#else
#define ALLOC(ptr, nbytes) do { if (hipMalloc((void **)&(ptr), (nbytes)) != hipSuccess) { g_last_hip = (int)hipGetLastError(); fotg_destroy(c); return FOTG_ERR_HIP; } } while (0)
#endif
    const size_t bytes1 = (B + 1) * c->lev_stride[l] * sizeof(float);       // sequence mode: max_batch pairs = max_batch + 1 frames
    ALLOC(c->im[0][l], bytes1);
    ALLOC(c->im[1][l], bytes);
    if (l >= p->sc_l) {
      ALLOC(c->dx0[l], bytes1);
      ALLOC(c->dy0[l], bytes1);
      ALLOC(c->flow[l], B * g.w * g.h * 2 * sizeof(float));
      ALLOC(c->p_iter[l], B * g.nop * 2 * sizeof(float));
      ALLOC(c->pweight[l], B * g.nop * (size_t)(p->ps * p->ps * c->noc) * sizeof(float));
      if (p->usefbcon) {
        ALLOC(c->dx1[l], bytes);
        ALLOC(c->dy1[l], bytes);
        ALLOC(c->flow_bw[l], B * g.w * g.h * 2 * sizeof(float));
        ALLOC(c->p_iter_bw[l], B * g.nop * 2 * sizeof(float));
        ALLOC(c->pweight_bw[l], B * g.nop * (size_t)(p->ps * p->ps * c->noc) * sizeof(float));
      }
    }
  }
  if (p->usetvref) {
    const LevelGeom &g = c->geom[p->sc_l];
    // lexicographic solver: up to 1024 rows any kernel, up to 16384 rows the tile pipeline (at most four sweeps per launch); the
    // depth solver has one thread per row of a workgroup up to 1024 rows and loops over the rows beyond (one sweep per launch)
    if (g.h > 16384 && p->sor_mode == FOTG_SOR_LEXICOGRAPHIC) { fotg_destroy(c); return FOTG_ERR_UNSUPPORTED; }
    if (g.h > 16384 && p->sor_mode == FOTG_SOR_POINT) { fotg_destroy(c); return FOTG_ERR_UNSUPPORTED; }
    if (g.w < 3 || c->geom[p->sc_f].h < 5 || c->geom[p->sc_f].w < 3) { fotg_destroy(c); return FOTG_ERR_UNSUPPORTED; }
    c->vr_pair_stride = (long)g.st * g.h * (P_NSINGLE + C_NCOLOR * c->noc + (p->depth ? (int)DE_NPLANE : 0));
    ALLOC(c->vr, B * c->vr_pair_stride * sizeof(float));
    static const int ks[] = {1, 2, 3, 4, 6, 8, 12, 16};
    for (int l = p->sc_l; l <= p->sc_f; ++l) {
      const LevelGeom &gl = c->geom[l];
      VrArgs &a = c->vra[l];
      memset(&a, 0, sizeof(a));
      a.base = c->vr; a.pair_stride = c->vr_pair_stride; a.w = gl.w; a.h = gl.h; a.st = gl.st; a.noc = c->noc;
      a.pl = (long)gl.st * gl.h;
      a.point = p->sor_mode == FOTG_SOR_POINT;
      int K = 16;
      for (int k : ks) if (k * 64 >= gl.h) { K = k; break; }
      a.K = K; a.nlanes = (gl.h + K - 1) / K; a.RP = a.nlanes * K; a.RPD = ((a.RP + K + 1 + 1) / 2) * 2;      // + K padding rows for idle lanes, + 1 for the bottom neighbour
      a.S = gl.w + gl.h - 1; a.SC = a.S + 1;        // one spare (zero) row: idle lanes read past the last row
      a.c_pair_stride = (long)a.SC * a.RP * 2;
      a.d_pair_stride = (long)(a.S + 1) * a.RPD;
      if (p->depth) continue;
      // + slack: idle lanes of the solver read K cells past the row they are parked on, i.e. past the last pair's
      // last (spare) row
      const size_t cbytes = B * a.c_pair_stride * sizeof(float4) + 64 * 16 * 2 * sizeof(float4);
      ALLOC(c->vrC[l], cbytes);
      ALLOC(c->vrD[l], B * a.d_pair_stride * sizeof(float2) + 4096);
      if (hipMemset(c->vrC[l], 0, cbytes) != hipSuccess) { fotg_destroy(c); return FOTG_ERR_HIP; }
      a.C = c->vrC[l]; a.D = c->vrD[l];
      // tall levels (beyond the LDS solvers' 96 rows): per-sweep arrays of the tile pipeline, zero outside the image for good
      // (more than four sweeps -- the operating points use three -- run as consecutive launches of at most four: only levels of
      // more than 1024 rows, which no other solver reaches, pay for that)
      if (p->sor_mode == FOTG_SOR_LEXICOGRAPHIC && gl.h > 96 && p->tv_solverit >= 1 && (p->tv_solverit <= 4 || gl.h > 1024) &&
          (gl.h + FOTG_TILE_ROWS - 1) / FOTG_TILE_ROWS <= 256) {
        {
          // every lane of every band has a cell of its own in a row (no two lanes share a store target)
          const int nbr = ((gl.h + FOTG_TILE_ROWS - 1) / FOTG_TILE_ROWS) * FOTG_TILE_ROWS, need = ((gl.h + 2 + 15) / 16) * 16;
          c->x_rt[l] = nbr > need ? nbr : need;
        }
        // (the tile kernels address a pair's arrays through buffer resources: 32-bit byte offsets)
        if ((double)(a.S + 1) * a.RP * 32.0 >= 4294967296.0 || (double)(a.S + 1 + FOTG_TILE_DUMP) * c->x_rt[l] * 8.0 >= 4294967296.0) { fotg_destroy(c); return FOTG_ERR_UNSUPPORTED; }
        c->x_pair_stride[l] = (long)(p->tv_solverit < 4 ? p->tv_solverit : 4) * (a.S + 1 + FOTG_TILE_DUMP) * c->x_rt[l];
        const size_t xb = B * c->x_pair_stride[l] * sizeof(float2);
        ALLOC(c->vrX[l], xb);
        if (hipMemset(c->vrX[l], 0, xb) != hipSuccess) { fotg_destroy(c); return FOTG_ERR_HIP; }
        if (!c->tileSync) {
          // (the first level that gets here is the finest = tallest one: its tile rows size the level-pipe words)
          c->lp_ntr = (gl.h + FOTG_LP_TH - 1) / FOTG_LP_TH;
          c->tile_nbs = (gl.h + FOTG_TILE_ROWS - 1) / FOTG_TILE_ROWS;
          const long lp = lp_tile_words((int)B, c->tile_nbs) + lp_data_words((int)B, c->lp_ntr);
          c->sync_total = lp > tile_sync_words((int)B, c->tile_nbs) ? lp : tile_sync_words((int)B, c->tile_nbs);
          ALLOC(c->tileSync, (c->sync_total + 32) * sizeof(int));
          if (hipMemset(c->tileSync, 0, (c->sync_total + 32) * sizeof(int)) != hipSuccess) { fotg_destroy(c); return FOTG_ERR_HIP; }
        }
      }
    }
  }
#undef ALLOC
  *out = c;
  return FOTG_OK;
}

int fotg_enable_taps(fotg_ctx *c, int on)
{
  if (!c) return FOTG_ERR_ARG;
  if (on && !c->taps) {
    ON_DEVICE(c->device);
    for (int l = c->p.sc_l; l <= c->p.sc_f; ++l) {
      const size_t n = (size_t)c->max_batch * c->geom[l].nop, nv = (size_t)c->ps * c->ps * c->noc;
      HIPCHK(hipMalloc((void **)&c->tap_t[l], n * nv * 4)); HIPCHK(hipMalloc((void **)&c->tap_tx[l], n * nv * 4));
      HIPCHK(hipMalloc((void **)&c->tap_ty[l], n * nv * 4)); HIPCHK(hipMalloc((void **)&c->tap_hes[l], n * 3 * 4));
      HIPCHK(hipMalloc((void **)&c->tap_cnt[l], n * 4));
      HIPCHK(hipMalloc((void **)&c->trace_dev[l], (size_t)c->geom[l].nop * (c->p.max_iter + 1) * 4 * 4));
    }
    c->taps = true;
  }
  return FOTG_OK;
}

int fotg_level_size(const fotg_ctx *c, int l, int *w, int *h)
{
  if (!c || l < c->base_lv || l > c->p.sc_f) return FOTG_ERR_ARG;
  if (w) *w = c->geom[l].w; if (h) *h = c->geom[l].h;
  return FOTG_OK;
}
int fotg_out_size(const fotg_ctx *c, int *w, int *h) { return c ? fotg_level_size(c, c->p.sc_l, w, h) : FOTG_ERR_ARG; }
int fotg_num_patches(const fotg_ctx *c, int l, int *nopw, int *noph)
{
  if (!c || l < c->p.sc_l || l > c->p.sc_f) return FOTG_ERR_ARG;
  if (nopw) *nopw = c->geom[l].nopw; if (noph) *noph = c->geom[l].noph;
  return FOTG_OK;
}

/* ------------------------------------------------------------------------------------------------ */
/* pyramid                                                                                          */
/* ------------------------------------------------------------------------------------------------ */
}  // extern "C"
// I0 and/or I1 may be given; both frames of a batch share the launches
// SRCC: channels of the source frames (3 with NOC = 1: 8-bit colour frames, gray on load -- fotg_params::u8_color)
template <int NOC, typename T = float, int SRCC = NOC>
static int pyramid_impl(fotg_ctx *c, int n, const T *I0, const T *I1, hipStream_t s, int stages = 3)
{
  const int lv = c->base_lv, ps = c->ps;
  const LevelGeom &g0 = c->geom[lv];
  const int strips = (c->Wp + 255) >> 8, tiles = strips * (c->Hp >> lv);
  const long fstride = (long)c->w_org * c->h_org * SRCC;
  const int coef0 = c->p.u8_color == 2 ? 4899 : 1868, coef2 = c->p.u8_color == 2 ? 1868 : 4899;      // first / third byte of a pixel: B, R (cv::imread order) or R, B
  const T *A = I0 ? I0 : I1, *B = (I0 && I1) ? I1 : nullptr;
  float *dA = c->im[I0 ? 0 : 1][lv], *dB = c->im[1][lv];
  const int nimg = B ? 2 * n : n;
  // fast path: no horizontal padding, rows and frames aligned for the wide loads (16 B for f32, 4 B for u8)
  const uintptr_t amask = sizeof(T) == 4 ? 15 : 3;
  const int fast = (c->padw == 0) && ((c->w_org * SRCC) % 4 == 0) && (((uintptr_t)A & amask) == 0) && (!B || ((uintptr_t)B & amask) == 0) && ((fstride % 4) == 0);
  const int groups = (tiles + 3) / 4;
  dim3 grid(groups, nimg), block(256);
#define BASE(LV) do { if (fast) pyr_base_kernel<T, NOC, LV, true, SRCC><<<grid, block, 0, s>>>(A, B, n, fstride, c->w_org, c->h_org, c->padw / 2, c->padh / 2, c->Wp, c->Hp, dA, dB, c->lev_stride[lv], g0.tw, ps, coef0, coef2); \
    else pyr_base_kernel<T, NOC, LV, false, SRCC><<<grid, block, 0, s>>>(A, B, n, fstride, c->w_org, c->h_org, c->padw / 2, c->padh / 2, c->Wp, c->Hp, dA, dB, c->lev_stride[lv], g0.tw, ps, coef0, coef2); } while (0)
  if (stages & 1) {
    // FOTG_PYR_SPLIT > 1: the batch's images in that many launches, one after the other.  The launch is the path's only HBM-bound
    // kernel and fills every wave slot of the chip for its whole duration; with several batches in flight the kernels of the
    // other slots (dispatched oldest first) then get in at every launch boundary instead of after the whole pyramid.
    int split = 1;
    if (B && c->tune.pyr_split > 1)                     // the largest cut <= the target into groups of a multiple of 4 pairs (XCD-local placement)
      for (int k = sizeof(T) == 1 && c->tune.pyr_split > 8 ? 8 : c->tune.pyr_split; k > 1; --k)      // (8-bit frames: a quarter of the bytes per launch)
        if (n % k == 0 && (n / k) % 4 == 0) { split = k; break; }
    const int gs = n / split;
    const T *A0 = A, *B0 = B;
    float *dA0 = dA, *dB0 = dB;
    const int n_all = n;
    for (int part = 0; part < split; ++part) {
      if (split > 1) {
        A = A0 + (size_t)part * gs * fstride; B = B0 + (size_t)part * gs * fstride;
        dA = dA0 + (size_t)part * gs * c->lev_stride[lv]; dB = dB0 + (size_t)part * gs * c->lev_stride[lv];
        n = gs; grid.y = 2 * gs;
      }
      switch (lv) {
        case 0: BASE(0); break;
        case 1: BASE(1); break;
        case 2: BASE(2); break;
        case 3: BASE(3); break;
        default: BASE(4); break;
      }
      LAUNCHCHK();
    }
    n = n_all;
  }
#undef BASE
  if (!(stages & 2)) return FOTG_OK;
  const int nlev = c->p.sc_f - lv + 1;
  if ((long)c->geom[c->p.sc_l].tw * c->geom[c->p.sc_l].th * NOC <= 32768 && (long)g0.w * g0.h * NOC <= 32768) {
    // small levels: one fused launch, one workgroup per image
    PyrFinishArgs fa;
    memset(&fa, 0, sizeof(fa));
    for (int k = 0; k < nlev; ++k) {
      const int l = lv + k;
      fa.im[0][k] = c->im[0][l]; fa.im[1][k] = c->im[1][l]; fa.dx[k] = c->dx0[l]; fa.dy[k] = c->dy0[l];
      fa.stride[k] = c->lev_stride[l]; fa.w[k] = c->geom[l].w; fa.h[k] = c->geom[l].h;
    }
    fa.nlev = nlev; fa.first_used = c->p.sc_l - lv; fa.ps = ps;
    if (I0 && I1) { fa.n_per_src = n; }
    else if (I0) { fa.n_per_src = n; }                       // only `which` 0 blocks exist
    else { fa.n_per_src = 0; }                               // every block is `which` 1
    pyr_finish_kernel<NOC><<<dim3(nimg, fa.first_used == 0 ? 2 : 1), 1024, 0, s>>>(fa);
    LAUNCHCHK();
    return FOTG_OK;
  }
  if (I0 && I1) {
    // both frames of the batch per launch: one halving launch per level (they depend on each other), ONE launch for the borders
    // and gradients of all levels (4K operating point 4: 6 launches instead of 22)
    for (int l = lv + 1; l <= c->p.sc_f; ++l) {
      const LevelGeom &gs = c->geom[l - 1], &gd = c->geom[l];
      const int tot = gd.w * gd.h * NOC;
      pyr_halve_kernel<NOC><<<dim3((tot + 255) / 256, 2 * n), 256, 0, s>>>(c->im[0][l - 1], c->lev_stride[l - 1], gs.tw, c->im[0][l], c->lev_stride[l], gd.tw,
                                                                           gd.w, gd.h, ps, c->im[1][l - 1], c->im[1][l], n);
      LAUNCHCHK();
    }
    PyrBorderArgs ba;
    memset(&ba, 0, sizeof(ba));
    int nl = 0, maxtot = 0;
    for (int l = c->p.sc_l; l <= c->p.sc_f; ++l, ++nl) {
      const LevelGeom &g = c->geom[l];
      ba.im[0][nl] = c->im[0][l]; ba.im[1][nl] = c->im[1][l]; ba.dx[nl] = c->dx0[l]; ba.dy[nl] = c->dy0[l];
      ba.stride[nl] = c->lev_stride[l]; ba.w[nl] = g.w; ba.h[nl] = g.h;
      const int tot = g.tw * g.th * NOC;
      maxtot = tot > maxtot ? tot : maxtot;
    }
    ba.n_a = n; ba.ps = ps;
    pyr_border_grad_multi_kernel<NOC><<<dim3((maxtot + 255) / 256, 2 * n, nl), 256, 0, s>>>(ba);
    LAUNCHCHK();
    return FOTG_OK;
  }
  for (int which = 0; which < 2; ++which) {
    if (!(which == 0 ? I0 : I1)) continue;
    for (int l = lv + 1; l <= c->p.sc_f; ++l) {
      const LevelGeom &gs = c->geom[l - 1], &gd = c->geom[l];
      const int tot = gd.w * gd.h * NOC;
      pyr_halve_kernel<NOC><<<dim3((tot + 255) / 256, n), 256, 0, s>>>(c->im[which][l - 1], c->lev_stride[l - 1], gs.tw,
                                                                       c->im[which][l], c->lev_stride[l], gd.tw, gd.w, gd.h, ps);
      LAUNCHCHK();
    }
    for (int l = c->p.sc_l; l <= c->p.sc_f; ++l) {
      const LevelGeom &g = c->geom[l];
      const int tot = g.tw * g.th * NOC;
      pyr_border_grad_kernel<NOC><<<dim3((tot + 255) / 256, n), 256, 0, s>>>(
          c->im[which][l], which == 0 ? c->dx0[l] : nullptr, which == 0 ? c->dy0[l] : nullptr, c->lev_stride[l], g.w, g.h, ps);
      LAUNCHCHK();
    }
  }
  return FOTG_OK;
}

// the pyramid of a flow call: channels of the context; 8-bit colour frames of a gray context (u8_color) are converted on load
template <typename T>
static int pyramid_any(fotg_ctx *c, int n, const T *I0, const T *I1, hipStream_t s)
{
  if constexpr (sizeof(T) == 1) { if (c->p.u8_color) return pyramid_impl<1, T, 3>(c, n, I0, I1, s); }
  return c->noc == 1 ? pyramid_impl<1, T>(c, n, I0, I1, s) : pyramid_impl<3, T>(c, n, I0, I1, s);
}

extern "C" {
int fotg_pyramid(fotg_ctx *c, int n, const float *I, int which, void *stream)
{
  if (!c || !I || (which != 0 && which != 1)) return FOTG_ERR_ARG;
  if (n < 1 || n > c->max_batch) return FOTG_ERR_BATCH;
  ON_DEVICE(c->device);
  const float *I0 = which == 0 ? I : nullptr, *I1 = which == 1 ? I : nullptr;
  return c->noc == 1 ? pyramid_impl<1>(c, n, I0, I1, (hipStream_t)stream) : pyramid_impl<3>(c, n, I0, I1, (hipStream_t)stream);
}

int fotg_pyramid_pair(fotg_ctx *c, int n, const float *I0, const float *I1, int stages, void *stream)
{
  if (!c || !I0 || !I1 || !(stages & 3)) return FOTG_ERR_ARG;
  if (n < 1 || n > c->max_batch) return FOTG_ERR_BATCH;
  ON_DEVICE(c->device);
  return c->noc == 1 ? pyramid_impl<1>(c, n, I0, I1, (hipStream_t)stream, stages) : pyramid_impl<3>(c, n, I0, I1, (hipStream_t)stream, stages);
}

int fotg_pyramid_pair_u8(fotg_ctx *c, int n, const unsigned char *I0, const unsigned char *I1, int stages, void *stream)
{
  if (!c || !I0 || !I1 || !(stages & 3)) return FOTG_ERR_ARG;
  if (n < 1 || n > c->max_batch) return FOTG_ERR_BATCH;
  ON_DEVICE(c->device);
  if (c->p.u8_color) return pyramid_impl<1, unsigned char, 3>(c, n, I0, I1, (hipStream_t)stream, stages);
  return c->noc == 1 ? pyramid_impl<1, unsigned char>(c, n, I0, I1, (hipStream_t)stream, stages) : pyramid_impl<3, unsigned char>(c, n, I0, I1, (hipStream_t)stream, stages);
}

int fotg_level_ptr(fotg_ctx *c, int which, int l, int kind, float **ptr, long *pair_stride)
{
  if (!c || !ptr || l < c->p.sc_l || l > c->p.sc_f || (which != 0 && which != 1)) return FOTG_ERR_ARG;
  float *p = nullptr;
  if (kind == 0) p = c->im[which][l];
  else if (which == 0 && kind == 1) p = c->dx0[l];
  else if (which == 0 && kind == 2) p = c->dy0[l];
  if (!p) return FOTG_ERR_ARG;
  *ptr = p;
  if (pair_stride) *pair_stride = c->lev_stride[l];
  return FOTG_OK;
}

/* ------------------------------------------------------------------------------------------------ */
/* patch grid                                                                                       */
/* ------------------------------------------------------------------------------------------------ */
static int check_level(fotg_ctx *c, int l, int n)
{
  if (!c || l < c->p.sc_l || l > c->p.sc_f) return FOTG_ERR_ARG;
  if (n < 1 || n > c->max_batch) return FOTG_ERR_BATCH;
  return FOTG_OK;
}

int fotg_grid_init(fotg_ctx *c, int l, int n, const float *I0, const float *I0x, const float *I0y, long pair_stride, void *stream)
{
  (void)stream;
  int st = check_level(c, l, n); if (st) return st;
  if (!I0 || !I0x || !I0y) return FOTG_ERR_ARG;
  GridState &g = c->gs[l];
  g.I0 = I0; g.I0x = I0x; g.I0y = I0y; g.stride = pair_stride;
  g.flow_prev = nullptr;                                            // p_init.setZero() (patchgrid.cpp:113)
  return FOTG_OK;
}
int fotg_grid_set_target(fotg_ctx *c, int l, const float *I1, long pair_stride)
{
  int st = check_level(c, l, 1); if (st) return st;
  if (!I1) return FOTG_ERR_ARG;
  c->gs[l].I1 = I1;
  if (c->gs[l].stride && c->gs[l].stride != pair_stride) return FOTG_ERR_ARG;
  c->gs[l].stride = pair_stride;
  return FOTG_OK;
}
int fotg_grid_init_from_coarser(fotg_ctx *c, int l, int n, const float *flow_prev, void *stream)
{
  (void)stream;
  int st = check_level(c, l, n); if (st) return st;
  if (!flow_prev) return FOTG_ERR_ARG;
  c->gs[l].flow_prev = flow_prev;
  return FOTG_OK;
}
int fotg_grid_set_trace(fotg_ctx *c, int l, float *trace_host)
{
  int st = check_level(c, l, 1); if (st) return st;
  if (trace_host) { st = fotg_enable_taps(c, 1); if (st) return st; }
  c->gs[l].trace_host = trace_host;
  return FOTG_OK;
}

int fotg_grid_optimize(fotg_ctx *c, int l, int n, void *stream)
{
  int st = check_level(c, l, n); if (st) return st;
  ON_DEVICE(c->device);
  GridState &gs = c->gs[l];
  if (!gs.I0 || !gs.I1) return FOTG_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  const LevelGeom &g = c->geom[l];
  LkArgs a;
  memset(&a, 0, sizeof(a));
  a.I0 = gs.I0; a.I0x = gs.I0x; a.I0y = gs.I0y; a.I1 = gs.I1; a.img_stride = gs.stride;
  a.flow_prev = gs.flow_prev;
  a.flow_prev_stride = (long)(g.w / 2) * (g.h / 2) * c->nch;
  a.camlr = gs.camlr;
  a.p_iter = c->p_iter[l]; a.pweight = c->pweight[l];
  if (c->taps) { a.tmpl = c->tap_t[l]; a.tdx = c->tap_tx[l]; a.tdy = c->tap_ty[l]; a.hes = c->tap_hes[l]; a.cnt = c->tap_cnt[l]; }
  a.trace = gs.trace_host ? c->trace_dev[l] : nullptr;
  a.g = g;
  a.max_iter = c->p.max_iter; a.min_iter = c->p.min_iter; a.patnorm = c->p.patnorm;
  a.shw_test = (c->tune.test_taps && c->tune.lk_shw >= 2) ? c->tune.lk_shw - 1 : 0;     // (fast_math: 2 = every evaluation reads the level image instead of the staged window)
  a.costfct = c->p.costfct; a.huber_bsq = c->p.normoutlier * c->p.normoutlier; a.huber_2bsq = a.huber_bsq * 2.0f;   // kroeger/oflow.cpp:106-107
  a.dp_thresh_sq = c->p.dp_thresh * c->p.dp_thresh;                 // kroeger/oflow.cpp:88
  a.dr_thresh = c->p.dr_thresh; a.res_thresh = c->p.res_thresh;
  a.outlier = (float)c->ps / 2;                                     // :82
  {
    // the largest float whose (correctly rounded) square root is <= outlier: sqrtf(s) > outlier  <=>  s > outlier_sq
    float sq = a.outlier * a.outlier;
    while (sqrtf(sq) > a.outlier) sq = nextafterf(sq, 0.0f);
    while (sqrtf(nextafterf(sq, INFINITY)) <= a.outlier) sq = nextafterf(sq, INFINITY);
    a.outlier_sq = sq;
  }
  // four patches per wave, one wave per workgroup (lk.hip.h); eight (eight lanes per patch) for launches of the operating points'
  // L2 cost with enough waves to stay throughput-bound at three waves per SIMD (FOTG_LK_LPP: 0 automatic, 8, 16)
  const long waves8 = ((long)g.nop + 7) / 8 * n;
  const bool lpp8 = !c->p.depth && c->p.costfct == 0 && (c->ps == 8 || c->ps == 12) && c->noc == 1 &&
                    (c->tune.lk_lpp == 8 || (c->tune.lk_lpp == 0 && c->ps == 8 && waves8 >= c->tune.lk_lpp_min_waves));
  const int ppw = lpp8 ? 8 : 4;
  dim3 block(64), grid((g.nop + ppw - 1) / ppw, n);
  // pair counts that are not a multiple of 8 (xcd_local_wg() keeps the plain order then): XCD-banded placement for launches that span the
  // chip (FOTG_LK_BANDED=0: plain order; tests)
  const bool banded = (n & 7) != 0 && c->tune.lk_banded;
  a.nwg = 0;
  if (banded && grid.x >= 64) { a.nwg = (int)grid.x; grid.x = (grid.x + 7) & ~7u; }
  // the operating points' L2 cost: specialised kernels, with one shared LDS area per wave where private windows limit occupancy
  // (measured: RGB patches -- two to three waves per SIMD with private windows -- gain 9-19 % per level; gray ones are bound by
  // the issue rate of their instruction stream at any occupancy and lose the time of the packing plan: docs/EXPERIMENTS.md)
  const bool shw = c->tune.lk_shw < 0 ? (c->noc == 3 && c->ps >= 8) : c->tune.lk_shw != 0;
  // fotg_params::fast_math: the tolerance-mode kernel (lk_fast.hip.h) for what every operating point runs -- L2 cost, optical
  // flow, min_iter == max_iter, res_thresh <= 0; anything else (and the per-iteration trace tap) stays on the exact kernel
  if (c->p.fast_math && !c->p.depth && c->p.costfct == 0 && c->p.min_iter == c->p.max_iter && !(c->p.res_thresh > 0.0f) && !gs.trace_host) {
    // gray 8 x 8 / 12 x 12 patches stage a window of radius 2 around the start (FOTG_LK_FAST_R=0: the whole reachable region).
    // Sixteen lanes per patch throughout: eight (3 x 6 / 2 x 4 blocks) and four (4 x 4 blocks at ps 8) were measured slower or
    // equal at every launch size (docs/EXPERIMENTS.md, round 5).
    const bool small = c->noc == 1 && (c->ps == 8 || c->ps == 12) && c->tune.lk_fast_r > 0;
    dim3 gridf((g.nop + 3) / 4, n);
    a.nwg = 0;
    if (banded && gridf.x >= 64) { a.nwg = (int)gridf.x; gridf.x = (gridf.x + 7) & ~7u; }
#define LKF(PS_, NOC_, R_) lk_fast_kernel<PS_, NOC_, 16, R_><<<gridf, block, 0, s>>>(a)
    switch (c->ps * 10 + c->noc) {
      case 41: LKF(4, 1, 0); break;   case 43: LKF(4, 3, 0); break;
      case 81: if (small) LKF(8, 1, 2); else LKF(8, 1, 0); break;
      case 83: LKF(8, 3, 0); break;
      case 121: if (small) LKF(12, 1, 2); else LKF(12, 1, 0); break;
      case 123: LKF(12, 3, 0); break;
      case 161: LKF(16, 1, 0); break; default: LKF(16, 3, 0); break;
    }
#undef LKF
    LAUNCHCHK();
    return FOTG_OK;
  }
#define LK(PS_, NOC_) do { if (lpp8 && (PS_ == 8 || PS_ == 12) && NOC_ == 1) lk_kernel<(PS_ == 8 || PS_ == 12) ? PS_ : 8, 1, false, true, true, 8><<<grid, block, 0, s>>>(a); \
                           else if (a.costfct == 0 && shw) lk_kernel<PS_, NOC_, false, true, true><<<grid, block, 0, s>>>(a); \
                           else if (a.costfct == 0) lk_kernel<PS_, NOC_, false, true><<<grid, block, 0, s>>>(a); \
                           else lk_kernel<PS_, NOC_, false><<<grid, block, 0, s>>>(a); } while (0)
#define LKD(PS_, NOC_) lk_kernel<PS_, NOC_, true><<<grid, block, 0, s>>>(a)
  if (c->p.depth) {
    switch (c->ps * 10 + c->noc) {
      case 41: LKD(4, 1); break;   case 43: LKD(4, 3); break;
      case 81: LKD(8, 1); break;   case 83: LKD(8, 3); break;
      case 121: LKD(12, 1); break; case 123: LKD(12, 3); break;
      case 161: LKD(16, 1); break; default: LKD(16, 3); break;
    }
  } else {
    switch (c->ps * 10 + c->noc) {
      case 41: LK(4, 1); break;   case 43: LK(4, 3); break;
      case 81: LK(8, 1); break;   case 83: LK(8, 3); break;
      case 121: LK(12, 1); break; case 123: LK(12, 3); break;
      case 161: LK(16, 1); break; default: LK(16, 3); break;
    }
  }
#undef LK
#undef LKD
  LAUNCHCHK();
  if (gs.trace_host) {
    HIPCHK(hipStreamSynchronize(s));
    HIPCHK(hipMemcpy(gs.trace_host, c->trace_dev[l], (size_t)g.nop * (c->p.max_iter + 1) * 16, hipMemcpyDeviceToHost));
  }
  return FOTG_OK;
}

// AggregateFlowDense of the grid (p_iter, pweight); cg_*: the complementary grid set by SetComplGrid or nullptr
static int aggregate_impl(fotg_ctx *c, int l, int n, const float *p_iter, const float *pweight, const float *cg_p_iter,
                          const float *cg_pweight, float *flowout, hipStream_t s)
{
  const LevelGeom &g = c->geom[l];
  const long fs = (long)g.w * g.h * c->nch;
  const int nch = c->nch;
  if (cg_p_iter) {
    dim3 grid(((g.w + 15) / 16) * ((g.h + 15) / 16), n), block(256);
#define DFB(PS_, NOC_) densify_fb_kernel<PS_, NOC_><<<grid, block, 0, s>>>(p_iter, pweight, cg_p_iter, cg_pweight, flowout, fs, g, nch)
    if (c->ps == 4) { if (c->noc == 1) DFB(4, 1); else DFB(4, 3); }
    else if (c->ps == 16) { if (c->noc == 1) DFB(16, 1); else DFB(16, 3); }
#undef DFB
    else if (c->ps == 8 && c->noc == 1) densify_fb_kernel<8, 1><<<grid, block, 0, s>>>(p_iter, pweight, cg_p_iter, cg_pweight, flowout, fs, g, nch);
    else if (c->ps == 8) densify_fb_kernel<8, 3><<<grid, block, 0, s>>>(p_iter, pweight, cg_p_iter, cg_pweight, flowout, fs, g, nch);
    else if (c->noc == 1) densify_fb_kernel<12, 1><<<grid, block, 0, s>>>(p_iter, pweight, cg_p_iter, cg_pweight, flowout, fs, g, nch);
    else densify_fb_kernel<12, 3><<<grid, block, 0, s>>>(p_iter, pweight, cg_p_iter, cg_pweight, flowout, fs, g, nch);
    LAUNCHCHK();
    return FOTG_OK;
  }
  dim3 grid((g.w * g.h + 255) / 256, n), block(256);
  // (n not a multiple of 8: XCD-banded placement, see xcd_banded_x; only worth it for launches that span the chip)
  int nwg = 0;
  if ((n & 7) != 0 && grid.x >= 64) { nwg = (int)grid.x; grid.x = (grid.x + 7) & ~7u; }
#define DF(PS_, NOC_) densify_kernel<PS_, NOC_><<<grid, block, 0, s>>>(p_iter, pweight, flowout, fs, g, nch, nwg)
  if (c->ps == 4) { if (c->noc == 1) DF(4, 1); else DF(4, 3); }
  else if (c->ps == 16) { if (c->noc == 1) DF(16, 1); else DF(16, 3); }
#undef DF
  else if (c->ps == 8 && c->noc == 1) densify_kernel<8, 1><<<grid, block, 0, s>>>(p_iter, pweight, flowout, fs, g, nch, nwg);
  else if (c->ps == 8) densify_kernel<8, 3><<<grid, block, 0, s>>>(p_iter, pweight, flowout, fs, g, nch, nwg);
  else if (c->noc == 1) densify_kernel<12, 1><<<grid, block, 0, s>>>(p_iter, pweight, flowout, fs, g, nch, nwg);
  else densify_kernel<12, 3><<<grid, block, 0, s>>>(p_iter, pweight, flowout, fs, g, nch, nwg);
  LAUNCHCHK();
  return FOTG_OK;
}

int fotg_grid_aggregate(fotg_ctx *c, int l, int n, float *flowout, void *stream)
{
  int st = check_level(c, l, n); if (st) return st;
  if (!flowout) return FOTG_ERR_ARG;
  ON_DEVICE(c->device);
  return aggregate_impl(c, l, n, c->p_iter[l], c->pweight[l], nullptr, nullptr, flowout, (hipStream_t)stream);
}

int fotg_grid_read(fotg_ctx *c, int l, int pair, float *p_iter, float *pweight, float *tmpl, float *tdx, float *tdy,
                   float *hes, int *cnt)
{
  int st = check_level(c, l, 1); if (st) return st;
  if (pair < 0 || pair >= c->max_batch) return FOTG_ERR_ARG;
  ON_DEVICE(c->device);
  HIPCHK(hipDeviceSynchronize());
  const size_t nop = c->geom[l].nop, nv = (size_t)c->ps * c->ps * c->noc, pb = (size_t)pair * nop;
  if (p_iter) HIPCHK(hipMemcpy(p_iter, c->p_iter[l] + pb * 2, nop * 2 * 4, hipMemcpyDeviceToHost));
  if (pweight) HIPCHK(hipMemcpy(pweight, c->pweight[l] + pb * nv, nop * nv * 4, hipMemcpyDeviceToHost));
  if (tmpl || tdx || tdy || hes || cnt) {
    if (!c->taps) return FOTG_ERR_ARG;
    if (tmpl) HIPCHK(hipMemcpy(tmpl, c->tap_t[l] + pb * nv, nop * nv * 4, hipMemcpyDeviceToHost));
    if (tdx) HIPCHK(hipMemcpy(tdx, c->tap_tx[l] + pb * nv, nop * nv * 4, hipMemcpyDeviceToHost));
    if (tdy) HIPCHK(hipMemcpy(tdy, c->tap_ty[l] + pb * nv, nop * nv * 4, hipMemcpyDeviceToHost));
    if (hes) HIPCHK(hipMemcpy(hes, c->tap_hes[l] + pb * 3, nop * 3 * 4, hipMemcpyDeviceToHost));
    if (cnt) HIPCHK(hipMemcpy(cnt, c->tap_cnt[l] + pb, nop * 4, hipMemcpyDeviceToHost));
  }
  return FOTG_OK;
}

/* ------------------------------------------------------------------------------------------------ */
/* variational refinement                                                                           */
/* ------------------------------------------------------------------------------------------------ */
}  // extern "C"
template <int K, int P>
static void launch_sor(const VrArgs &a, int n, int sweeps, float omega, hipStream_t s)
{
  constexpr int U = (K <= 2) ? (P >= 8 ? 64 : 8 * P) : (K <= 4 ? 4 * P : P);     // steps per loop trip
  vr_sor_kernel<K, P, U><<<n, 64, 0, s>>>(a, sweeps, omega);
}

// Row bands of the barrier-stepped solver waves (sor_sync_wave): bands of <= 64 rows, one lane per row, one wave per (sweep, band).
static void set_bands(VrArgs &b, int sweeps)
{
  b.nbands = 0; b.band_rows = 0; b.band_mode = 0;
  const int nb = (b.h + 63) / 64;
  if (sweeps * nb > 8 || sweeps > 4) return;
  b.nbands = nb;
  b.band_rows = (b.h + nb - 1) / nb;
  b.band_mode = 3;
}

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-device property of a kernel: cache what was set per device
static std::mutex g_lds_mu;             // contexts / pipes driven from different host threads share the per-kernel caches below
static bool ensure_dyn_lds(const void *fn, int lds, int (&set)[32])
{
  std::lock_guard<std::mutex> lock(g_lds_mu);
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 32) { (void)hipGetLastError(); dev = 0; }
  if (lds <= set[dev]) return true;
  if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) { (void)hipGetLastError(); return false; }
  set[dev] = lds;
  return true;
}

static std::atomic<long> g_fused_cglobal_launches{0};   // fotg_debug_counter("fused_cglobal"): fused per-level launches with the system in global memory
static std::atomic<long> g_pipe_launches{0};            // fotg_debug_counter("sor_pipe"): vr_sor_pipe_kernel launches
static bool launch_sor_pipe(int n, float omega, hipStream_t s, const VrArgs &b)
{
  constexpr int P = 8, U = 32;
  const int lds = 128 + (b.S + 2) * b.RPD * (int)sizeof(float2);
  static int lds_set[32] = {0};
  if (!ensure_dyn_lds(reinterpret_cast<const void *>(&vr_sor_pipe_kernel<P, U>), lds, lds_set)) return false;
  vr_sor_pipe_kernel<P, U><<<n, 1024, lds, s>>>(b, omega);          // all 16 waves copy D in and out
  ++g_pipe_launches;
  return true;
}

static std::atomic<long> g_tile_launches{0};        // fotg_debug_counter("sor_tiles")
static std::atomic<long> g_tall_launches{0};        // fotg_debug_counter("sor_tall")
static std::atomic<long> g_levelpipe_launches{0};   // fotg_debug_counter("level_pipe")
#ifndef FOTG_TILE_P
#define FOTG_TILE_P 8                    // prefetch depth (diagonals) of the tile solver
#endif
static int a_level(const fotg_ctx *c, const VrArgs &a) { for (int l = c->p.sc_l; l <= c->p.sc_f; ++l) if (c->vra[l].w == a.w && c->vra[l].h == a.h) return l; return c->p.sc_l; }
static std::atomic<long> g_stream_launches{0};      // fotg_debug_counter("sor_stream"): tests assert the kernel really ran
// streaming solver (vr_sor_stream_kernel): diagonals travel through LDS rings, two rows per lane.
template <int RD, int RCW, bool FMA = false>
static bool launch_sor_stream_k(const VrArgs &b, int n, float omega, hipStream_t s)
{
  constexpr int M = FOTG_SYNC_M, U = 32;
  using GEO = StreamGeom<RD, RCW>;
  if (b.RP < 64 || b.RP + 1 > RCW || b.RPD > RD || (b.RPD & 1) || b.S < 2 * U) return false;
  if (b.nsweeps > 3 || b.h + 2 > b.RPD) return false;
  const int DS = ((M + 2 + M - 1) / M) * M;
  const int omax = b.nsweeps > 0 ? (b.nsweeps - 1) * DS : 0;
  const int RDN = M * (GEO::LI + omax / M + 1 + 1), RCN = RDN - M;
  const int lds = RCN * GEO::CSLOT + RDN * GEO::DB + GEO::DB;
  if (lds > 160 * 1024) return false;
  static int lds_set[32] = {0};
  if (!ensure_dyn_lds(reinterpret_cast<const void *>(&vr_sor_stream_kernel<RD, RCW, M, U, FMA>), lds, lds_set)) return false;
  vr_sor_stream_kernel<RD, RCW, M, U, FMA><<<n, 1024, lds, s>>>(b, omega);
  ++g_stream_launches;
  return true;
}

// Levels of 65..96 rows (1080p level 4: 68).  Two geometries: <= 69 rows and <= 97 rows per diagonal.
// tune.vr_stream = 0 (FOTG_VR_STREAM=0 at context creation; tests): the resident-D kernel instead.
static bool launch_sor_stream(const fotg_ctx *c, const VrArgs &b, int n, float omega, hipStream_t s)
{
  if (!c->tune.vr_stream) return false;
  if (c->p.fast_math) return launch_sor_stream_k<72, 70, true>(b, n, omega, s) || launch_sor_stream_k<100, 98, true>(b, n, omega, s);
  return launch_sor_stream_k<72, 70>(b, n, omega, s) || launch_sor_stream_k<100, 98>(b, n, omega, s);
}

// sweep-pipelined LDS solver when it applies: <= 4 sweeps (one wave each per band)
static bool dispatch_sor_pipe(const fotg_ctx *c, const VrArgs &a, int n, int sweeps, float omega, hipStream_t s)
{
#ifdef FOTG_DEBUG
  if (const char *e = getenv("FOTG_DEBUG_SWEEPS")) sweeps = atoi(e);                    // timing experiments only (wrong results)
#endif
  const int lds = 128 + (a.S + 2) * a.RPD * (int)sizeof(float2);
  if (sweeps < 1 || sweeps > 4 || a.S < 24) return false;
  VrArgs b = a;
  b.nsweeps = sweeps;
#ifdef FOTG_DEBUG
  if (getenv("FOTG_DEBUG_NOSOR")) b.nsweeps = 0;                // timing experiments only
#endif
  set_bands(b, sweeps);
  if (b.band_mode != 3) return false;
  if (launch_sor_stream(c, b, n, omega, s)) return true;        // 65..96 rows, any width: diagonals stream through LDS rings
  if (lds > 150 * 1024) return false;                           // the kernel below keeps the whole (du,dv) in LDS
  return launch_sor_pipe(n, omega, s, b);
}

static int fused_lds_bytes(const VrArgs &b, bool with_c)
{
  return 128 + (b.S + 2) * b.RPD * (int)sizeof(float2) + ((b.w * b.h + 3) / 4) * 16 + (with_c ? (b.SC * b.RP + 1) * 32 : 0);
}

template <int NOC, bool CL = false, bool RES = false, int NT = 512, bool FM = false>
static bool launch_inner_fused(const VrArgs &b, int n, int inner, float qa, float hd, float hg, float omega, float *flow, long fs, hipStream_t s,
                               const float *I0, const float *I1, long img_stride, int tw, int pad)
{
  constexpr int P = 8, U = 32;
  const int lds = fused_lds_bytes(b, CL);
  static int lds_set[32] = {0};
  if (!ensure_dyn_lds(reinterpret_cast<const void *>(&vr_inner_fused_kernel<NOC, P, U, CL, RES, NT, FM>), lds, lds_set)) return false;
  vr_inner_fused_kernel<NOC, P, U, CL, RES, NT, FM><<<n, NT, lds, s>>>(b, inner, qa, hd, hg, omega, flow, fs, I0, I1, img_stride, tw, pad);
  return true;
}

// whole fixed-point loop in one launch when (du,dv) + the smoothness plane fit in LDS, sweeps <= 4 and the level has <= 64 rows
template <int NOC>
static bool dispatch_inner_fused(const fotg_ctx *c, const VrArgs &a, int n, int sweeps, int inner, float qa, float hd, float hg, float omega, float *flow,
                                 long fs, hipStream_t s, const float *I0, const float *I1, long img_stride, int tw, int pad, int taps)
{
  // (fast_math: the data term in the tolerance mode's arithmetic, varref_dataterm.inc.h)
#define FOTG_FUSED(...) (c->p.fast_math ? launch_inner_fused<__VA_ARGS__, true>(b, n, inner, qa, hd, hg, omega, flow, fs, s, I0, I1, img_stride, tw, pad) \
                                        : launch_inner_fused<__VA_ARGS__, false>(b, n, inner, qa, hd, hg, omega, flow, fs, s, I0, I1, img_stride, tw, pad))
  const int lds = fused_lds_bytes(a, false);
  // one workgroup does the per-pixel phases of its pair: only worth it for small levels (measured: 60x34 yes, 120x68 no)
  // (red-black has no dependency chain: its half-sweeps use all the workgroup's threads, and one launch per level beats
  // 2 + inner (1 + 2 sweeps) launches at any level whose (du,dv) and smoothness plane fit in LDS -- 1080p level 4 included)
  const bool rb = c->p.sor_mode == FOTG_SOR_REDBLACK;
  if (sweeps < 1 || sweeps > 4 || inner < 1 || lds > 156 * 1024 || a.S < 24 || (a.w * a.h > 3000 && !rb)) return false;
  VrArgs b = a;
  b.taps = taps;
  b.nsweeps = sweeps;
  b.redblack = rb;
#ifdef FOTG_DEBUG
  if (getenv("FOTG_DEBUG_NOSOR")) b.nsweeps = 0;                // timing experiments only
#endif
  set_bands(b, sweeps);
  if (b.band_mode != 3 || b.nbands > 1) return false;           // the fused kernel's barrier-stepped waves assume a single band (<= 64 rows)
  // system cells in LDS as well when they fit
  if (fused_lds_bytes(a, true) <= 160 * 1024) {
    // gray levels of <= 4 pixels per thread also keep their per-pixel inputs in registers over the loop
    if constexpr (NOC == 1) {
      // (levels of more than 1024 pixels: 1024 threads, two pixels each -- 4 waves per SIMD hide the latencies of the per-pixel phases)
      if (a.w * a.h > 1024 && a.w * a.h <= 2048 && FOTG_FUSED(1, true, true, 1024)) return true;
      if (a.w * a.h <= 4 * 512 && FOTG_FUSED(1, true, true, 512)) return true;
    }
    if (FOTG_FUSED(NOC, true, false, 512)) return true;
  }
  // wide, short levels whose skewed system (w + h) x h x 32 B does not fit beside (du,dv) and the smoothness plane -- e.g. 100 x 30: 133 KB --
  // keep it in global memory (fotg_debug_counter("fused_cglobal") counts these launches)
  if (!FOTG_FUSED(NOC, false, false, 512)) return false;
  ++g_fused_cglobal_launches;
  return true;
#undef FOTG_FUSED
}

// FOTG_SOR_POINT (sor_coupled_slow_but_readable, a compatibility mode): the single-wave wavefront solver with the point update
static void dispatch_sor_point(const VrArgs &a, int n, int sweeps, float omega, hipStream_t s)
{
  if (a.h > 1024) {                  // more rows than 64 lanes x 16: a sweep per launch, rows looped (vr_sor_tall_kernel)
    for (int k = 0; k < sweeps; ++k) vr_sor_tall_kernel<true><<<n, 1024, 0, s>>>(a, omega);
    ++g_tall_launches;
    return;
  }
  switch (a.K) {
#define PT(K_) case K_: vr_sor_kernel<K_, 1, (K_ <= 4 ? 4 : 1), true><<<n, 64, 0, s>>>(a, sweeps, omega); break
    PT(1); PT(2); PT(3); PT(4); PT(6); PT(8); PT(12);
    default: vr_sor_kernel<16, 1, 1, true><<<n, 64, 0, s>>>(a, sweeps, omega); break;
#undef PT
  }
}

static void dispatch_sor(const fotg_ctx *c, const VrArgs &a, int n, int sweeps, float omega, hipStream_t s, bool sync_zeroed = false)
{
  const int path = c->tune.vr_path;    // 0 = automatic, 1 = single-wave global-memory solver only, 2 = no fused inner loop (tests)
  if (path != 1 && dispatch_sor_pipe(c, a, n, sweeps, omega, s)) return;
  // FOTG_VR_PATH=1 -- tests, and the recompute of a stalled tile pipeline -- uses no inter-workgroup waits anywhere: levels of more than
  // 1024 rows take the one-workgroup-per-pair wavefront (a sweep per launch), shorter ones the single-wave kernel below
  if (path == 1 && a.h > 1024) {
    for (int k = 0; k < sweeps; ++k) vr_sor_tall_kernel<false><<<n, 1024, 0, s>>>(a, omega);
    ++g_tall_launches;
    return;
  }
  // levels too tall for the LDS solvers: tiles = (sweep, band of 64 rows), one workgroup each, pipelined through global memory
  // (varref_tiles.hip.h)
  if (path != 1 && c->vrX[a_level(c, a)] && c->tileSync) {
    const int l = a_level(c, a);
    TileArgs g;
    g.X = c->vrX[l] + (size_t)(a.C - c->vrC[l]) / a.c_pair_stride * c->x_pair_stride[l];      // (views: same pair offset as C)
    g.x_pair_stride = c->x_pair_stride[l];
    g.x_buf_stride = (long)(a.S + 1 + FOTG_TILE_DUMP) * c->x_rt[l];
    g.RT = c->x_rt[l];
    g.NB = (a.h + FOTG_TILE_ROWS - 1) / FOTG_TILE_ROWS;
    g.npairs = n;
    g.sync = c->tileSync;
    g.NBS = c->tile_nbs;
    g.timeouts = g.sync + c->sync_total;
    g.stall_flag = c->stall_dev;
#ifdef FOTG_TILE_STATS
    if (!c->stamps) { if (hipMalloc((void **)&c->stamps, 4096 * 32 * 8) != hipSuccess) return; }
    (void)hipMemsetAsync(c->stamps, 0, 4096 * 32 * 8, s);
    g.stats = (long long *)c->stamps;
#endif
    // the sweeps are sequential passes over the same system, every launch starts from and ends in the level's D: more than four
    // sweeps = consecutive launches of at most four (one wave per sweep and band, X buffers for four), the same bits
    for (int done = 0; done < sweeps; done += 4) {
      const int sw = sweeps - done < 4 ? sweeps - done : 4;
      // (the data-term launch in front of the call has cleared the words already when the caller arranged that: VrArgs::zsync)
      if (!(sync_zeroed && done == 0)) (void)hipMemsetAsync(g.sync, 0, (size_t)tile_sync_words(n, c->tile_nbs) * sizeof(int), s);
      if (c->p.fast_math) vr_sor_tile_kernel<FOTG_TILE_P, true><<<n * g.NB * sw, FOTG_TILE_THREADS, 0, s>>>(a, g, sw, omega);
      else vr_sor_tile_kernel<FOTG_TILE_P><<<n * g.NB * sw, FOTG_TILE_THREADS, 0, s>>>(a, g, sw, omega);
      ++g_tile_launches;
    }
    return;
  }
  // prefetch depth: as deep as the register budget of K rows per lane allows, and 2P+2 <= S (ring never
  // runs ahead into rows the current sweep has not rewritten yet)
  const int cap = (a.S - 2) / 2;
  auto pick = [&](int pmax) { int p = pmax; while (p > 1 && p > cap) p >>= 1; return p; };
  switch (a.K) {
    case 1: switch (pick(16)) { case 16: launch_sor<1, 16>(a, n, sweeps, omega, s); break; case 8: launch_sor<1, 8>(a, n, sweeps, omega, s); break; case 4: launch_sor<1, 4>(a, n, sweeps, omega, s); break;
                               case 2: launch_sor<1, 2>(a, n, sweeps, omega, s); break; default: launch_sor<1, 1>(a, n, sweeps, omega, s); } break;
    case 2: switch (pick(8)) { case 8: launch_sor<2, 8>(a, n, sweeps, omega, s); break; case 4: launch_sor<2, 4>(a, n, sweeps, omega, s); break;
                               case 2: launch_sor<2, 2>(a, n, sweeps, omega, s); break; default: launch_sor<2, 1>(a, n, sweeps, omega, s); } break;
    case 3: switch (pick(4)) { case 4: launch_sor<3, 4>(a, n, sweeps, omega, s); break; case 2: launch_sor<3, 2>(a, n, sweeps, omega, s); break;
                               default: launch_sor<3, 1>(a, n, sweeps, omega, s); } break;
    case 4: switch (pick(4)) { case 4: launch_sor<4, 4>(a, n, sweeps, omega, s); break; case 2: launch_sor<4, 2>(a, n, sweeps, omega, s); break;
                               default: launch_sor<4, 1>(a, n, sweeps, omega, s); } break;
    case 6: launch_sor<6, 1>(a, n, sweeps, omega, s); break;
    case 8: launch_sor<8, 1>(a, n, sweeps, omega, s); break;
    case 12: launch_sor<12, 1>(a, n, sweeps, omega, s); break;
    default: launch_sor<16, 1>(a, n, sweeps, omega, s); break;
  }
}

template <int NOC>
static int varref_impl(fotg_ctx *c, int l, int n, const float *I0, const float *I1, long img_stride, float *flow, hipStream_t s)
{
  const LevelGeom &g = c->geom[l];
  const VrArgs &a = c->vra[l];
  const long fs = (long)g.w * g.h * 2;
  dim3 grid((g.w * g.h + 255) / 256, n), block(256);
  // kroeger/refine_variational.cpp:31-43
  const float quarter_alpha = 0.25f * c->p.tv_alpha;
  const float half_gamma_over3 = c->p.tv_gamma * 0.5f / 3.0f;
  const float half_delta_over3 = c->p.tv_delta * 0.5f / 3.0f;
  const int inner = c->p.tv_innerit * (l + 1);
  // small levels: the whole level (set-up stages, fixed-point loop, final w + d) in one launch, one workgroup per pair
  if (c->p.sor_mode != FOTG_SOR_POINT && c->p.tv_solverit > 0 && c->tune.vr_path == 0 &&
      dispatch_inner_fused<NOC>(c, a, n, c->p.tv_solverit, inner, quarter_alpha, half_delta_over3, half_gamma_over3, c->p.tv_sor, flow, fs, s,
                                I0, I1, img_stride, g.tw, c->ps, c->taps ? 1 : 0)) {
    LAUNCHCHK();
    return FOTG_OK;
  }
  // levels that go through the tile pipeline: the launch in front of every sor_coupled call (data term; set-up with the first data
  // term) clears the pipeline's sync words
  VrArgs az = a;
  const bool tiles = c->p.sor_mode == FOTG_SOR_LEXICOGRAPHIC && c->vrX[l] && c->tileSync && c->p.tv_solverit > 0;
  if (tiles) { az.zsync = c->tileSync; az.zsync_n = (int)tile_sync_words(n, c->tile_nbs); }
  // FOTG_VR_LEVELPIPE=1: the level's whole fixed-point loop as ONE pipeline launch behind the set-up launch (varref_levelpipe.hip.h).
  // At least two sweeps per call (what keeps a band's last sweep behind the data term of its neighbours), at most four (X buffers).
  const int ntr = (g.h + FOTG_LP_TH - 1) / FOTG_LP_TH;
  const bool levelpipe = tiles && c->tune.vr_levelpipe && n <= c->tune.lp_max_pairs && c->tune.vr_path == 0 && c->tune.vr_first_data && inner >= 1 && inner <= FOTG_LP_KMAX &&
                         c->p.tv_solverit >= 2 && c->p.tv_solverit <= 4 && ntr <= c->lp_ntr &&
                         lp_tile_words(n, c->tile_nbs) + lp_data_words(n, ntr) <= c->sync_total;
  if (levelpipe) { az.zsync_n = (int)(lp_tile_words(n, c->tile_nbs) + lp_data_words(n, ntr)); }
  // warp + first + second derivatives in one tiled launch, which also zeroes (du,dv) (refine_variational.cpp:185-186) and builds the
  // system of the first inner iteration (unless tune.vr_first_data = 0: a data-term launch of its own)
  const bool merged_first = c->tune.vr_first_data && inner > 0;
  {
    dim3 gs_(((g.w + 31) / 32) * ((g.h + 7) / 8), n);
    // (1..7 pairs, a launch that spans the chip: XCD-banded tiles like the LK launches; FOTG_LK_BANDED=0: plain order)
    if ((n & 7) != 0 && gs_.x >= 256 && c->tune.lk_banded) { az.nwg = (int)gs_.x; gs_.x = (gs_.x + 7) & ~7u; }
    if (c->p.fast_math)
      vr_setup_kernel<NOC, 2, true><<<gs_, 256, 0, s>>>(az, I0, I1, img_stride, g.tw, c->ps, flow, fs, 1, merged_first ? 1 : 0, quarter_alpha, half_delta_over3, half_gamma_over3);
    else
      vr_setup_kernel<NOC><<<gs_, 256, 0, s>>>(az, I0, I1, img_stride, g.tw, c->ps, flow, fs, 1, merged_first ? 1 : 0, quarter_alpha, half_delta_over3, half_gamma_over3);
    LAUNCHCHK();
  }
  if (levelpipe) {
    TileArgs tg;
    tg.X = c->vrX[l] + (size_t)(a.C - c->vrC[l]) / a.c_pair_stride * c->x_pair_stride[l];
    tg.x_pair_stride = c->x_pair_stride[l];
    tg.x_buf_stride = (long)(a.S + 1 + FOTG_TILE_DUMP) * c->x_rt[l];
    tg.RT = c->x_rt[l];
    tg.NB = (a.h + FOTG_TILE_ROWS - 1) / FOTG_TILE_ROWS;
    tg.npairs = n;
    tg.sync = c->tileSync;
    tg.NBS = c->tile_nbs;
    tg.timeouts = tg.sync + c->sync_total;
    tg.stall_flag = c->stall_dev;
    LevelPipeArgs q;
    q.K = inner; q.ntr = ntr; q.tiles_x = (g.w + FOTG_TW - 1) / FOTG_TW;
    q.dprog = c->tileSync + lp_tile_words(n, c->tile_nbs);
    q.quarter_alpha = quarter_alpha; q.half_delta_over3 = half_delta_over3; q.half_gamma_over3 = half_gamma_over3;
    q.dbg = c->tune.vr_levelpipe >> 4;       // (FOTG_VR_LEVELPIPE = 1 + 16 * dbg)
    q.stamps = nullptr;
    if (q.dbg & 4) {                         // diagnosis only: per-role time stamps (tools/levelpipe_stamps.py)
      if (!c->stamps && hipMalloc((void **)&c->stamps, 8 * 8 * 8192) != hipSuccess) return FOTG_ERR_HIP;
      (void)hipMemsetAsync(c->stamps, 0, 8 * 8 * 8192, s);
      q.stamps = (long long *)c->stamps;
    }
    const int sw = c->p.tv_solverit;
    const unsigned nwg = (unsigned)n * (unsigned)(inner * tg.NB * sw + (inner - 1) * ntr * FOTG_LP_DW);
    // a launch that fits the chip with one workgroup per CU asks for enough LDS to get exactly that: a solver wave that shares its
    // SIMD with a data-term wave of another workgroup runs up to 25 % slower (measured)
    const int excl = (nwg <= 256 && !(q.dbg & 8)) ? 72 * 1024 : 0;
    if (c->p.fast_math) {
      static int lds_set[32] = {0};
      if (excl && !ensure_dyn_lds(reinterpret_cast<const void *>(&vr_level_pipe_kernel<NOC, FOTG_TILE_P, true>), excl, lds_set)) return FOTG_ERR_HIP;
      vr_level_pipe_kernel<NOC, FOTG_TILE_P, true><<<nwg, 256, excl, s>>>(a, tg, q, sw, c->p.tv_sor);
    } else {
      static int lds_set[32] = {0};
      if (excl && !ensure_dyn_lds(reinterpret_cast<const void *>(&vr_level_pipe_kernel<NOC, FOTG_TILE_P, false>), excl, lds_set)) return FOTG_ERR_HIP;
      vr_level_pipe_kernel<NOC, FOTG_TILE_P, false><<<nwg, 256, excl, s>>>(a, tg, q, sw, c->p.tv_sor);
    }
    LAUNCHCHK();
    ++g_levelpipe_launches;
    vr_finish_kernel<<<grid, block, 0, s>>>(a, flow, fs);
    LAUNCHCHK();
    return FOTG_OK;
  }
  for (int it = 0; it < inner; ++it) {
    if (!(it == 0 && merged_first)) {
      if (c->p.fast_math) vr_data_kernel<NOC, true><<<dim3(((g.w + FOTG_TW - 1) / FOTG_TW) * ((g.h + FOTG_TH - 1) / FOTG_TH), n), 256, 0, s>>>(az, quarter_alpha, half_delta_over3, half_gamma_over3);
      else vr_data_kernel<NOC><<<dim3(((g.w + FOTG_TW - 1) / FOTG_TW) * ((g.h + FOTG_TH - 1) / FOTG_TH), n), 256, 0, s>>>(az, quarter_alpha, half_delta_over3, half_gamma_over3);
      LAUNCHCHK();
    }
    if (c->p.tv_solverit > 0) {
      if (c->p.sor_mode == FOTG_SOR_REDBLACK) {
        // one launch per half-sweep: every cell of the even, then of the odd diagonals, the whole batch at once
        for (int sw = 0; sw < c->p.tv_solverit; ++sw)
          for (int col = 0; col < 2; ++col) {
            const int nd = (a.S - col + 1) / 2;
            vr_rb_halfsweep_kernel<<<dim3((nd * a.RP + 255) / 256, n), 256, 0, s>>>(a, col, c->p.tv_sor);
          }
      }
      else if (c->p.sor_mode == FOTG_SOR_POINT) dispatch_sor_point(a, n, c->p.tv_solverit, c->p.tv_sor, s);
      else dispatch_sor(c, a, n, c->p.tv_solverit, c->p.tv_sor, s, tiles);
      LAUNCHCHK();
    }
  }
  vr_finish_kernel<<<grid, block, 0, s>>>(a, flow, fs);
  LAUNCHCHK();
  return FOTG_OK;
}

// stereo depth mode: RefLevelDE (kroeger/refine_variational.cpp:243-330), flow has one channel
static bool g_de_lds_set[32];
template <int NOC>
static int varref_depth_impl(fotg_ctx *c, int l, int n, const float *I0, const float *I1, long img_stride, float *flow, hipStream_t s, int camlr)
{
  const LevelGeom &g = c->geom[l];
  const VrArgs &a = c->vra[l];
  const long fs = (long)g.w * g.h;
  dim3 grid((g.w * g.h + 255) / 256, n), block(256);
  const float quarter_alpha = 0.25f * c->p.tv_alpha;
  const float half_gamma_over3 = c->p.tv_gamma * 0.5f / 3.0f;
  const float half_delta_over3 = c->p.tv_delta * 0.5f / 3.0f;
  const int inner = c->p.tv_innerit * (l + 1);
  vr_setup_kernel<NOC, 1><<<dim3(((g.w + 31) / 32) * ((g.h + 7) / 8), n), 256, 0, s>>>(a, I0, I1, img_stride, g.tw, c->ps, flow, fs);
  LAUNCHCHK();
  const int threads = ((g.h + 63) / 64) * 64;
  const int du_bytes = g.st * g.h * (int)sizeof(float);
  // levels up to 8192 cells with the operating points' three sweeps: everything after the set-up in one launch per level
  // (FOTG_VR_PATH != 0 forces the launch-per-stage path below; tests)
  if (5 * du_bytes <= 160 * 1024 && c->p.tv_solverit == 3 && 3 * threads <= 1024 && c->tune.vr_path == 0) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    static bool set[32];
    std::lock_guard<std::mutex> lock(g_lds_mu);
    if (dev >= 0 && dev < 32 && !set[dev]) {
      HIPCHK(hipFuncSetAttribute((const void *)vr_de_inner_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      HIPCHK(hipFuncSetAttribute((const void *)vr_de_inner_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      set[dev] = true;
    }
    vr_de_inner_kernel<NOC><<<n, 1024, 5 * du_bytes, s>>>(a, inner, quarter_alpha, half_delta_over3, half_gamma_over3, c->p.tv_sor, camlr,
                                                          flow, fs, threads, c->taps ? 1 : 0);
    LAUNCHCHK();
    return FOTG_OK;
  }
  vr_de_init_kernel<<<grid, block, 0, s>>>(a);
  LAUNCHCHK();
  // du + the four coefficient planes in LDS when they fit in the CU's 160 KiB, else du alone, else global memory
  const int lds = 5 * du_bytes <= 160 * 1024 ? 2 : du_bytes <= 128 * 1024 ? 1 : 0;
  const int lds_bytes = lds == 2 ? 5 * du_bytes : lds == 1 ? du_bytes : 0;
  if (lds_bytes > 64 * 1024) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lock(g_lds_mu);
    if (dev >= 0 && dev < 32 && !g_de_lds_set[dev]) {
      HIPCHK(hipFuncSetAttribute((const void *)vr_de_sor_kernel<3, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      HIPCHK(hipFuncSetAttribute((const void *)vr_de_sor_kernel<3, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      HIPCHK(hipFuncSetAttribute((const void *)vr_de_sor_kernel<3, 1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
      HIPCHK(hipFuncSetAttribute((const void *)vr_de_sor_kernel<1, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      HIPCHK(hipFuncSetAttribute((const void *)vr_de_sor_kernel<3, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
      HIPCHK(hipFuncSetAttribute((const void *)vr_de_sor_kernel<1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
      g_de_lds_set[dev] = true;
    }
  }
  for (int it = 0; it < inner; ++it) {
    vr_de_smooth_kernel<<<grid, block, 0, s>>>(a, quarter_alpha);
    LAUNCHCHK();
    vr_de_data_kernel<NOC><<<grid, block, 0, s>>>(a, half_delta_over3, half_gamma_over3);
    LAUNCHCHK();
    // the sweeps are sequential passes over du, so `k` single-sweep launches equal one k-sweep launch bit for bit; the
    // operating points use 3.  (With 0 sweeps the clamped update still runs: uu = min/max(wx + du, 0).)
    if (g.h > 1024) {                                             // more rows than a workgroup has threads: one sweep per launch, rows looped
      for (int k = 0; k < c->p.tv_solverit; ++k)
        vr_de_sor_tall_kernel<<<n, 1024, 0, s>>>(a, c->p.tv_sor, camlr, k == c->p.tv_solverit - 1 ? 1 : 0);
    } else if (c->p.tv_solverit == 3 && 3 * threads <= 1024 && lds) {     // one wave group per sweep
      if (lds == 2) vr_de_sor_kernel<3, 2, true><<<n, 3 * threads, lds_bytes, s>>>(a, c->p.tv_sor, camlr);
      else vr_de_sor_kernel<3, 1, true><<<n, 3 * threads, lds_bytes, s>>>(a, c->p.tv_sor, camlr);
    } else if (c->p.tv_solverit == 3) {
      if (lds == 2) vr_de_sor_kernel<3, 2><<<n, threads, lds_bytes, s>>>(a, c->p.tv_sor, camlr);
      else if (lds == 1) vr_de_sor_kernel<3, 1><<<n, threads, lds_bytes, s>>>(a, c->p.tv_sor, camlr);
      else vr_de_sor_kernel<3, 0><<<n, threads, 0, s>>>(a, c->p.tv_sor, camlr);
    } else {
      for (int k = 0; k < c->p.tv_solverit; ++k) {
        if (lds == 2) vr_de_sor_kernel<1, 2><<<n, threads, lds_bytes, s>>>(a, c->p.tv_sor, camlr);
        else if (lds == 1) vr_de_sor_kernel<1, 1><<<n, threads, lds_bytes, s>>>(a, c->p.tv_sor, camlr);
        else vr_de_sor_kernel<1, 0><<<n, threads, 0, s>>>(a, c->p.tv_sor, camlr);
      }
    }
    LAUNCHCHK();
  }
  vr_de_finish_kernel<<<grid, block, 0, s>>>(a, flow, fs);
  LAUNCHCHK();
  return FOTG_OK;
}

static int varref_dispatch(fotg_ctx *c, int l, int n, const float *I0, const float *I1, long pair_stride, float *flow, hipStream_t stream, int camlr)
{
  int st = check_level(c, l, n); if (st) return st;
  if (!I0 || !I1 || !flow || !c->vr) return FOTG_ERR_ARG;
  if (c->geom[l].h < 5 || c->geom[l].w < 3) return FOTG_ERR_UNSUPPORTED;
  ON_DEVICE(c->device);
  if (c->p.depth)
    return c->noc == 1 ? varref_depth_impl<1>(c, l, n, I0, I1, pair_stride, flow, stream, camlr)
                       : varref_depth_impl<3>(c, l, n, I0, I1, pair_stride, flow, stream, camlr);
  return c->noc == 1 ? varref_impl<1>(c, l, n, I0, I1, pair_stride, flow, stream)
                     : varref_impl<3>(c, l, n, I0, I1, pair_stride, flow, stream);
}

extern "C" {
/* measurement tap (bench.py's roofline of the time-dominant kernel): ONE sor_coupled call (tv_solverit sweeps) of level l on
 * whatever system the last fotg_varref left in the workspace -- the launch the refinement issues once per inner iteration.
 * It advances (du,dv) of the workspace by three more sweeps; no product entry point reads that state across calls. */
int fotg_bench_sor_call(fotg_ctx *c, int l, int n, void *stream)
{
  int st = check_level(c, l, n); if (st) return st;
  if (!c->vr || c->p.depth || !c->vrC[l] || c->p.tv_solverit < 1 || c->p.sor_mode != FOTG_SOR_LEXICOGRAPHIC) return FOTG_ERR_UNSUPPORTED;
  ON_DEVICE(c->device);
  dispatch_sor(c, c->vra[l], n, c->p.tv_solverit, c->p.tv_sor, (hipStream_t)stream);
  LAUNCHCHK();
  return FOTG_OK;
}

int fotg_varref(fotg_ctx *c, int l, int n, const float *I0, const float *I1, long pair_stride, float *flow, void *stream)
{
  if (!c || l < 0 || l >= FOTG_MAXLEV) return FOTG_ERR_ARG;
  return varref_dispatch(c, l, n, I0, I1, pair_stride, flow, (hipStream_t)stream, c->gs[l].camlr);
}

int fotg_grid_set_camera(fotg_ctx *c, int l, int camlr)
{
  int st = check_level(c, l, 1); if (st) return st;
  if (camlr != 0 && camlr != 1) return FOTG_ERR_ARG;
  c->gs[l].camlr = camlr;
  return FOTG_OK;
}

int fotg_varref_plane(fotg_ctx *c, int pair, const char *name, int l, float *host_out)
{
  int st = check_level(c, l, 1); if (st) return st;
  if (!name || !host_out || !c->vr || pair < 0 || pair >= c->max_batch) return FOTG_ERR_ARG;
  static const char *singles[] = {"wx", "wy", "mask"};
  static const char *colors[] = {"avg", "Iz", "Ix", "Iy", "Ixx", "Ixy", "Iyy", "Ixz", "Iyz"};
  static const char *sys[] = {"a11", "a12", "b1", "b2", "a22", "sh", "sv", "svt"};      // cell layout of data_term_cell()
  const LevelGeom &g = c->geom[l];
  const VrArgs &a = c->vra[l];
  const size_t pl = (size_t)g.st * g.h;
  ON_DEVICE(c->device);
  HIPCHK(hipDeviceSynchronize());
  if (c->p.depth) {
    static const char *de[] = {"du", "uu", "s", "a11", "b1", "sh", "sv"};        // VrDePlane order
    for (int k = 0; k < DE_NPLANE; ++k)
      if (!strcmp(name, de[k])) {
        HIPCHK(hipMemcpy(host_out, de_plane(a, pair, k), pl * 4, hipMemcpyDeviceToHost));
        return FOTG_OK;
      }
  }
  for (int i = 0; i < P_NSINGLE; ++i)
    if (!strcmp(name, singles[i])) {
      HIPCHK(hipMemcpy(host_out, c->vr + (size_t)pair * c->vr_pair_stride + i * pl, pl * 4, hipMemcpyDeviceToHost));
      return FOTG_OK;
    }
  for (int i = 0; i < C_NCOLOR; ++i)
    if (!strcmp(name, colors[i])) {
      HIPCHK(hipMemcpy(host_out, c->vr + (size_t)pair * c->vr_pair_stride + (P_NSINGLE + (size_t)i * c->noc) * pl,
                       pl * c->noc * 4, hipMemcpyDeviceToHost));
      return FOTG_OK;
    }
  if (c->p.depth) return FOTG_ERR_ARG;
  // planes that live in the skewed solver arrays: copy and de-skew on the host
  for (int k = 0; k < 8; ++k)
    if (!strcmp(name, sys[k])) {
      float *tmp = (float *)malloc((size_t)a.c_pair_stride * sizeof(float4));
      if (!tmp) return FOTG_ERR_ARG;
      hipError_t e = hipMemcpy(tmp, a.C + (size_t)pair * a.c_pair_stride, (size_t)a.c_pair_stride * sizeof(float4), hipMemcpyDeviceToHost);
      if (e != hipSuccess) { free(tmp); g_last_hip = (int)e; return FOTG_ERR_HIP; }
      memset(host_out, 0, pl * 4);
      for (int j = 0; j < g.h; ++j) for (int i = 0; i < g.w; ++i) host_out[(size_t)j * g.st + i] = tmp[a.cidx(i, j) * 4 + k];
      free(tmp);
      return FOTG_OK;
    }
  if (!strcmp(name, "du") || !strcmp(name, "dv")) {
    float *tmp = (float *)malloc((size_t)a.d_pair_stride * sizeof(float2));
    if (!tmp) return FOTG_ERR_ARG;
    hipError_t e = hipMemcpy(tmp, a.D + (size_t)pair * a.d_pair_stride, (size_t)a.d_pair_stride * sizeof(float2), hipMemcpyDeviceToHost);
    if (e != hipSuccess) { free(tmp); g_last_hip = (int)e; return FOTG_ERR_HIP; }
    memset(host_out, 0, pl * 4);
    const int comp = name[1] == 'v';
    for (int j = 0; j < g.h; ++j) for (int i = 0; i < g.w; ++i) host_out[(size_t)j * g.st + i] = tmp[a.didx(i, j) * 2 + comp];
    free(tmp);
    return FOTG_OK;
  }
  return FOTG_ERR_ARG;
}

/* ------------------------------------------------------------------------------------------------ */
/* whole flow: OFClass::calc (src/oflow.cpp:211-368) with kroeger numerics (kroeger/oflow.cpp:184-337)   */
/* ------------------------------------------------------------------------------------------------ */
// the scale loop for pairs [0, n) of context (view) c on one stream
}  // extern "C"
// I1 == nullptr: sequence mode -- I0 holds n+1 consecutive frames, pair k is (frame k, frame k+1); every frame's pyramid
// is built once (with gradients) and serves as the target of pair k-1 and the template source of pair k.
template <typename T>
static int calc_range(fotg_ctx *c, int n, const T *I0, const T *I1, const float *initflow, float *outflow, hipStream_t stream)
{
  int st;
  const bool seq = I1 == nullptr, fb = c->p.usefbcon != 0;
  const int nimg = seq ? n + 1 : n;
  // stage timing for the reference's verbosity output: events on the launch stream (not while a graph is being captured)
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  (void)hipStreamIsCapturing(stream, &cap);
  const bool timing = c->verbosity > 0 && cap == hipStreamCaptureStatusNone;
  int nev = 0;
  auto mark = [&]() {
    if (!timing || nev >= (int)(sizeof(c->tev) / sizeof(c->tev[0]))) return;
    if (!c->tev[nev] && hipEventCreate(&c->tev[nev]) != hipSuccess) return;
    (void)hipEventRecord(c->tev[nev++], stream);
  };
  mark();
  // the backward grid lives in a view of the context whose grid arrays are the *_bw ones (nothing is owned by the view)
  fotg_ctx *vb = nullptr;
  struct ViewGuard { fotg_ctx *&v; ~ViewGuard() { free(v); } } guard{vb};
  if (fb) {
    vb = (fotg_ctx *)malloc(sizeof(fotg_ctx));
    if (!vb) return FOTG_ERR_ARG;
    memcpy((void *)vb, (const void *)c, sizeof(fotg_ctx));
    vb->taps = false;
    for (int l = c->p.sc_l; l <= c->p.sc_f; ++l) {
      vb->p_iter[l] = c->p_iter_bw[l]; vb->pweight[l] = c->pweight_bw[l];
      memset((void *)&vb->gs[l], 0, sizeof(GridState));
      vb->gs[l].camlr = 1;                                         // kroeger/oflow.cpp:157,165: the backward grid is the right camera
    }
  }
  if (fb && !seq) {
    // both frames need gradients: two template-type pyramids (the second one into the frame-1 buffers)
    if ((st = pyramid_any<T>(c, n, I0, (const T *)nullptr, stream))) return st;
    fotg_ctx *v1 = (fotg_ctx *)malloc(sizeof(fotg_ctx));
    if (!v1) return FOTG_ERR_ARG;
    memcpy((void *)v1, (const void *)c, sizeof(fotg_ctx));
    for (int l = c->base_lv; l <= c->p.sc_f; ++l) { v1->im[0][l] = c->im[1][l]; v1->dx0[l] = c->dx1[l]; v1->dy0[l] = c->dy1[l]; }
    st = pyramid_any<T>(v1, n, I1, (const T *)nullptr, stream);
    free(v1);
    if (st) return st;
  } else if ((st = pyramid_any<T>(c, nimg, I0, I1, stream))) return st;
  mark();
  for (int l = c->p.sc_f; l >= c->p.sc_l; --l) {
    const long ls = c->lev_stride[l];
    const float *tgt = seq ? c->im[0][l] + ls : c->im[1][l];
    if (fb) {
      // kroeger/oflow.cpp:190-235 and :262-295 with usefbcon: both grids, each one's densification merges the other's patches
      const float *tx = seq ? c->dx0[l] + ls : c->dx1[l], *ty = seq ? c->dy0[l] + ls : c->dy1[l];
      if ((st = fotg_grid_init(c, l, n, c->im[0][l], c->dx0[l], c->dy0[l], ls, stream))) return st;
      if ((st = fotg_grid_set_target(c, l, tgt, ls))) return st;
      if ((st = fotg_grid_init(vb, l, n, tgt, tx, ty, ls, stream))) return st;
      if ((st = fotg_grid_set_target(vb, l, c->im[0][l], ls))) return st;
      if (l < c->p.sc_f) {
        if ((st = fotg_grid_init_from_coarser(c, l, n, c->flow[l + 1], stream))) return st;
        if ((st = fotg_grid_init_from_coarser(vb, l, n, c->flow_bw[l + 1], stream))) return st;
      } else if (initflow) { if ((st = fotg_grid_init_from_coarser(c, l, n, initflow, stream))) return st; }
      if ((st = fotg_grid_optimize(c, l, n, stream))) return st;
      if ((st = fotg_grid_optimize(vb, l, n, stream))) return st;
      mark();
      float *out = (l == c->p.sc_l) ? outflow : c->flow[l];
      if ((st = aggregate_impl(c, l, n, c->p_iter[l], c->pweight[l], c->p_iter_bw[l], c->pweight_bw[l], out, stream))) return st;
      if (l > c->p.sc_l && (st = aggregate_impl(c, l, n, c->p_iter_bw[l], c->pweight_bw[l], c->p_iter[l], c->pweight[l], c->flow_bw[l], stream))) return st;
      mark();
      if (c->p.usetvref) {
        if ((st = varref_dispatch(c, l, n, c->im[0][l], tgt, ls, out, stream, 0))) return st;
        if (l > c->p.sc_l && (st = varref_dispatch(c, l, n, tgt, c->im[0][l], ls, c->flow_bw[l], stream, 1))) return st;
      }
      mark();
      continue;
    }
    if ((st = fotg_grid_init(c, l, n, c->im[0][l], c->dx0[l], c->dy0[l], c->lev_stride[l], stream))) return st;
    if ((st = fotg_grid_set_target(c, l, tgt, c->lev_stride[l]))) return st;
    if (l < c->p.sc_f) { if ((st = fotg_grid_init_from_coarser(c, l, n, c->flow[l + 1], stream))) return st; }
    else if (initflow) { if ((st = fotg_grid_init_from_coarser(c, l, n, initflow, stream))) return st; }
    if ((st = fotg_grid_optimize(c, l, n, stream))) return st;
    mark();
    float *out = (l == c->p.sc_l) ? outflow : c->flow[l];
    if ((st = fotg_grid_aggregate(c, l, n, out, stream))) return st;
    mark();
    if (c->p.usetvref)
      if ((st = varref_dispatch(c, l, n, c->im[0][l], tgt, c->lev_stride[l], out, stream, 0))) return st;
    mark();
  }
  if (timing && nev == 2 + 3 * (c->p.sc_f - c->p.sc_l + 1)) {
    // The reference's lines (src/oflow.cpp:343, :356; kroeger/oflow.cpp:303, :358), from the GPU times of the stages.  Patch
    // construction and initialisation from the coarser flow are part of the LK launch here (pconst = pinit = 0 by construction).
    HIPCHK(hipEventSynchronize(c->tev[nev - 1]));
    auto ms = [&](int a_, int b_) { float t = 0.f; (void)hipEventElapsedTime(&t, c->tev[a_], c->tev[b_]); return t; };
    c->tt_pyr = ms(0, 1);
    c->tt_total = ms(1, nev - 1);
    int k = 1;
    for (int l = c->p.sc_f; l >= c->p.sc_l; --l, k += 3) {
      float *t = c->tt[l];
      t[0] = 0.f; t[1] = 0.f; t[2] = ms(k, k + 1); t[3] = ms(k + 1, k + 2); t[4] = ms(k + 2, k + 3);
      if (c->verbosity > 1)
        printf("TIME (Sc: %i, #p:%6i, pconst, pinit, poptim, cflow, tvopt, total): %8.2f %8.2f %8.2f %8.2f %8.2f -> %8.2f ms.\n", l, c->geom[l].nop * n,
               t[0], t[1], t[2], t[3], t[4], t[2] + t[3] + t[4]);
    }
    printf("TIME (O.Flow Run-Time   ) (ms): %3g\n", c->tt_total);
    fflush(stdout);
  }
  return FOTG_OK;
}

extern "C" {
int fotg_calc_batch(fotg_ctx *c, int n, const float *I0, const float *I1, const float *initflow, float *outflow, void *stream)
{
  if (!c || !I0 || !I1 || !outflow) return FOTG_ERR_ARG;
  if (n < 1 || n > c->max_batch) return FOTG_ERR_BATCH;
  ON_DEVICE(c->device);
  return calc_range<float>(c, n, I0, I1, initflow, outflow, (hipStream_t)stream);
}

/* 8-bit frames (SURVEY 8f "next" row 2): same path, the pyramid base kernel converts on load (exact) and reads a
 * quarter of the bytes.  Eager single-stream launch sequence. */
int fotg_calc_batch_u8(fotg_ctx *c, int n, const unsigned char *I0, const unsigned char *I1, const float *initflow, float *outflow, void *stream)
{
  if (!c || !I0 || !I1 || !outflow) return FOTG_ERR_ARG;
  if (n < 1 || n > c->max_batch) return FOTG_ERR_BATCH;
  ON_DEVICE(c->device);
  return calc_range<unsigned char>(c, n, I0, I1, initflow, outflow, (hipStream_t)stream);
}

/* ---- batches in flight ---------------------------------------------------------------------------------------------
 * The path is latency-bound at batch 64 (the refinement keeps a quarter of the CUs busy), and consecutive batches are
 * independent.  A pipe owns `depth` engine contexts, each with a non-blocking stream of its own; batch k goes to context
 * k % depth and overlaps with the batches before it.  No host synchronisation anywhere: a slot's stream orders the reuse of
 * its context, events order it against the caller's streams. */
// Host sync points of the product API call this AFTER they have synchronised with the context's work: a time-out of a bounded
// inter-workgroup wait (a stalled producer: preemption, a debugger, a starved queue) means the flow of that call is wrong.
static int stall_status(fotg_ctx *c)
{
  if (!c->stall_host) return FOTG_OK;
  volatile int *f = c->stall_host;
  if (*f == 0) return FOTG_OK;
  *f = 0;
  ++c->stalls;
  return FOTG_ERR_STALL;
}

struct fotg_pipe {
  int device, depth;
  fotg_ctx *ctx[FOTG_PIPE_MAX_DEPTH];
  hipStream_t stream[FOTG_PIPE_MAX_DEPTH];
  hipEvent_t ready[FOTG_PIPE_MAX_DEPTH];
  // completion events per TICKET, not per slot: ticket t records done[t % nring], nring = 4 * depth.  (Round 3 kept one event per
  // slot; a device-side wait for ticket t then waited for whatever batch the slot carried by now -- the chunked scatter waits for
  // ticket t - depth right after submitting ticket t into the same slot, i.e. it waited for the chunk it had just submitted and
  // the transfer of the next chunk never overlapped with compute.)  nring is a multiple of depth, so an event that has been
  // re-recorded belongs to a later batch of the SAME slot's stream, which still covers the older ticket.
  hipEvent_t done[4 * FOTG_PIPE_MAX_DEPTH];
  int nring;
  long submitted;
  // Self-healing host waits.  The tile solver's bounded waits (varref_tiles.hip.h) raise ONE word per context; a host wait that finds
  // it set cannot tell which of the context's batches raised it, so it recomputes every batch of that context that has not been
  // verified yet -- on the solver path without inter-workgroup waits (FOTG_VR_PATH=1's) -- from the arguments kept here.  The caller's
  // contract (frames and outflow untouched until the ticket has been waited for) is what makes that legal.
  // A ticket is HEALABLE only while its buffers are contractually still in place: submitted without FOTG_SUBMIT_NO_RECOMPUTE and not yet
  // handed out through fotg_pipe_wait(host_wait = 0) / fotg_pipe_ticket_event (whoever waits that way may free or reuse the frames and the
  // outflow as soon as THEIR wait returns, without the pipe knowing).  Suspects that are not healable are reported, never recomputed.
  struct Args { int n, u8, healable; const void *I0, *I1; const float *initflow; float *out; } args[4 * FOTG_PIPE_MAX_DEPTH];
  signed char tstatus[4 * FOTG_PIPE_MAX_DEPTH];      // per ticket (of the last nring): 0 unknown, 1 good, 2 stalled and not recomputed
  long verified[FOTG_PIPE_MAX_DEPTH];                // per slot: tickets below this one are known good (or have their status in tstatus / lost_*)
  long lost_lo[FOTG_PIPE_MAX_DEPTH], lost_hi[FOTG_PIPE_MAX_DEPTH];   // per slot: tickets of [lost_lo, lost_hi) were suspects of a flagged stall when the ring
                                                     // (the last 4 * depth submissions) no longer described them: never recomputed, FOTG_ERR_STALL on every wait
  long healed;                                       // batches recomputed so far
  std::mutex *mu;                                    // submit / verification (fotg_node waits from another thread than the one that submits)
};

void fotg_pipe_destroy(fotg_pipe *q)
{
  if (!q) return;
  DevGuard dg(q->device);
  for (int k = 0; k < q->depth; ++k) {
    if (q->stream[k]) (void)hipStreamSynchronize(q->stream[k]);
    if (q->ctx[k]) fotg_destroy(q->ctx[k]);
    if (q->ready[k]) (void)hipEventDestroy(q->ready[k]);
    if (q->stream[k]) (void)hipStreamDestroy(q->stream[k]);
  }
  for (auto &e : q->done) if (e) (void)hipEventDestroy(e);
  delete q->mu;
  delete q;
}

int fotg_pipe_create(const fotg_params *p, int w_org, int h_org, int device, int max_batch, int depth, fotg_pipe **out)
{
  if (!out || depth < 1 || depth > FOTG_PIPE_MAX_DEPTH) return FOTG_ERR_ARG;
  ON_DEVICE(device);
  {
    // HIP deals its streams to GPU_MAX_HW_QUEUES hardware queues (default 4, the null stream included) and two busy streams on
    // one queue run one after the other: more than three slots need the variable set BEFORE the HIP runtime is loaded
    // (INTEGRATION.md section 4).  The library cannot set it any more at this point; say so once.
    static std::atomic<bool> warned{false};
    if (depth > 3 && env_int("GPU_MAX_HW_QUEUES", 4) < depth + 1 && !warned.exchange(true))      // (creation time, not the launch path)
      fprintf(stderr, "fotg_pipe_create: %d batches in flight need a hardware queue each (+ one for the null stream): GPU_MAX_HW_QUEUES "
                      "must be >= %d in the environment BEFORE libamdhip64 is loaded (a setenv after that is not seen by the runtime, and "
                      "not by this check either); otherwise slots share queues and do not overlap\n", depth, depth + 1);
  }
  fotg_pipe *q = new (std::nothrow) fotg_pipe();
  if (!q) return FOTG_ERR_ARG;
  memset((void *)q, 0, sizeof(*q));
  q->device = device; q->depth = depth; q->nring = 4 * depth;
  q->mu = new (std::nothrow) std::mutex();
  if (!q->mu) { delete q; return FOTG_ERR_ARG; }
  // the slots' streams first and back to back, so that the runtime spreads them over its hardware queues
  for (int k = 0; k < depth; ++k)
    if (hipStreamCreateWithFlags(&q->stream[k], hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&q->ready[k], hipEventDisableTiming) != hipSuccess) {
      g_last_hip = (int)hipGetLastError();
      fotg_pipe_destroy(q);
      return FOTG_ERR_HIP;
    }
  for (int k = 0; k < q->nring; ++k)
    if (hipEventCreateWithFlags(&q->done[k], hipEventDisableTiming) != hipSuccess) {
      g_last_hip = (int)hipGetLastError();
      fotg_pipe_destroy(q);
      return FOTG_ERR_HIP;
    }
  for (int k = 0; k < depth; ++k) {
    const int st = fotg_create(p, w_org, h_org, device, max_batch, &q->ctx[k]);
    if (st != FOTG_OK) { fotg_pipe_destroy(q); return st; }
    // several batches in flight: the base pyramid launch of a batch in up to 16 parts (pyramid_impl; measured 170 -> 181 k pairs/s
    // at batch 64 with four in flight, at the price of ~5 % on a batch that runs alone -- which is why only pipes do it)
    if (depth > 1) q->ctx[k]->tune.pyr_split = env_int("FOTG_PIPE_PYR_SPLIT", 16);
    // several batches in flight: the level pipeline (latency) only while few pairs are resident, the launch-per-stage path (throughput) beyond
    if (depth > 1) q->ctx[k]->tune.lp_max_pairs = env_int("FOTG_PIPE_LEVELPIPE_MAX_PAIRS", 4 / depth > 1 ? 4 / depth : 1);
  }
  *out = q;
  return FOTG_OK;
}

extern "C++" {
// one batch again on context c, synchronously, on the solver paths that have no inter-workgroup waits (FOTG_VR_PATH = 1: the single-wave
// solver; levels of more than 1024 rows run vr_sor_tall_kernel, one workgroup per pair); FOTG_OK = the flow is in place and valid
template <typename T>
static int recompute_safe(fotg_ctx *c, int n, const T *I0, const T *I1, const float *initflow, float *outflow, hipStream_t s)
{
  const int keep = c->tune.vr_path;
  int st = FOTG_ERR_STALL;
  for (int attempt = 0; attempt < 3 && st == FOTG_ERR_STALL; ++attempt) {
    c->tune.vr_path = 1;
    st = calc_range<T>(c, n, I0, I1, initflow, outflow, s);
    c->tune.vr_path = keep;
    if (st != FOTG_OK) return st;
    HIPCHK(hipStreamSynchronize(s));
    st = (c->stall_host && *(volatile int *)c->stall_host) ? FOTG_ERR_STALL : FOTG_OK;
    if (c->stall_host) *(volatile int *)c->stall_host = 0;
  }
  return st;
}

// After the host has synchronised with ticket t (slot k): classify the unverified tickets of the slot.  heal = recompute the
// suspects (1) or only mark them (0).  Called with the pipe's mutex held.
static int pipe_verify(fotg_pipe *q, long t, int heal, int *newly_stalled = nullptr)
{
  const int k = (int)(t % q->depth);
  fotg_ctx *c = q->ctx[k];
  auto known = [&](long u) { return u >= q->submitted - q->nring; };       // (the ring still describes ticket u)
  // status of a ticket that has been classified: from the ring while it is there, afterwards from the slot's range of lost suspects
  auto status_of = [&](long u) {
    if (known(u)) return q->tstatus[u % q->nring] == 2 ? FOTG_ERR_STALL : FOTG_OK;
    return u >= q->lost_lo[k] && u < q->lost_hi[k] ? FOTG_ERR_STALL : FOTG_OK;
  };
  if (t < q->verified[k]) return status_of(t);
  const bool flagged = c->stall_host && *(volatile int *)c->stall_host != 0;
  if (!flagged) {
    // everything of this slot that has completed so far is good: at least the tickets up to t
    for (long u = q->verified[k]; u <= t; u += 1) if (u % q->depth == k && known(u)) q->tstatus[u % q->nring] = 1;
    q->verified[k] = t + 1;
    return FOTG_OK;
  }
  // the word does not say which batch of this context raised it: all of them that are not verified yet are suspects
  HIPCHK(hipStreamSynchronize(q->stream[k]));
  *(volatile int *)c->stall_host = 0;
  ++c->stalls;
  for (long u = q->verified[k]; u < q->submitted; ++u) {
    if (u % q->depth != k) continue;
    if (!known(u)) {
      // more than 4 * depth submissions ago: its arguments are gone, so it can be neither recomputed nor cleared -- it stays a suspect
      // (a node submits an unbounded number of pieces per job; ADVICE round 5: such a ticket used to be waited for as FOTG_OK)
      if (q->lost_hi[k] <= q->lost_lo[k]) { q->lost_lo[k] = u; q->lost_hi[k] = u + 1; }
      else { if (u < q->lost_lo[k]) q->lost_lo[k] = u; if (u + 1 > q->lost_hi[k]) q->lost_hi[k] = u + 1; }
      if (newly_stalled) ++*newly_stalled;
      continue;
    }
    int st = FOTG_ERR_STALL;
    const fotg_pipe::Args &ar = q->args[u % q->nring];
    if (heal && ar.healable) {
      st = ar.u8 ? recompute_safe<unsigned char>(c, ar.n, (const unsigned char *)ar.I0, (const unsigned char *)ar.I1, ar.initflow, ar.out, q->stream[k])
                 : recompute_safe<float>(c, ar.n, (const float *)ar.I0, (const float *)ar.I1, ar.initflow, ar.out, q->stream[k]);
      if (st == FOTG_OK) ++q->healed;
      else if (st != FOTG_ERR_STALL) return st;
    }
    q->tstatus[u % q->nring] = st == FOTG_OK ? 1 : 2;
    if (st != FOTG_OK && newly_stalled) ++*newly_stalled;
  }
  q->verified[k] = q->submitted;
  return status_of(t);
}

template <typename T>
static int pipe_submit(fotg_pipe *q, int n, const T *I0, const T *I1, const float *initflow, float *outflow, void *after_stream, long *ticket, int flags = 0)
{
  if (!q || !I0 || !I1 || !outflow) return FOTG_ERR_ARG;
  std::lock_guard<std::mutex> lock(*q->mu);
  const int k = (int)(q->submitted % q->depth);
  fotg_ctx *c = q->ctx[k];
  if (n < 1 || n > c->max_batch) return FOTG_ERR_BATCH;
  ON_DEVICE(q->device);
  // the frames (and the reuse of outflow) are ordered behind what the caller has enqueued on `after_stream` so far
  if (after_stream != FOTG_NO_STREAM) {
    HIPCHK(hipEventRecord(q->ready[k], (hipStream_t)after_stream));
    HIPCHK(hipStreamWaitEvent(q->stream[k], q->ready[k], 0));
  }
  const int st = calc_range<T>(c, n, I0, I1, initflow, outflow, q->stream[k]);
  if (st != FOTG_OK) return st;
  HIPCHK(hipEventRecord(q->done[q->submitted % q->nring], q->stream[k]));
  {
    fotg_pipe::Args &ar = q->args[q->submitted % q->nring];
    ar.n = n; ar.u8 = sizeof(T) == 1; ar.I0 = I0; ar.I1 = I1; ar.initflow = initflow; ar.out = outflow;
    ar.healable = !(flags & FOTG_SUBMIT_NO_RECOMPUTE);
    q->tstatus[q->submitted % q->nring] = 0;
  }
  if (ticket) *ticket = q->submitted;
  ++q->submitted;
  return FOTG_OK;
}
}  // extern "C++"

int fotg_pipe_submit(fotg_pipe *q, int n, const float *I0, const float *I1, const float *initflow, float *outflow, void *after_stream, long *ticket)
{
  return pipe_submit<float>(q, n, I0, I1, initflow, outflow, after_stream, ticket);
}

int fotg_pipe_submit_u8(fotg_pipe *q, int n, const unsigned char *I0, const unsigned char *I1, const float *initflow, float *outflow, void *after_stream, long *ticket)
{
  return pipe_submit<unsigned char>(q, n, I0, I1, initflow, outflow, after_stream, ticket);
}

int fotg_pipe_submit_ex(fotg_pipe *q, int n, const void *I0, const void *I1, int u8, const float *initflow, float *outflow, void *after_stream,
                        int flags, long *ticket)
{
  if (flags & ~FOTG_SUBMIT_NO_RECOMPUTE) return FOTG_ERR_ARG;
  return u8 ? pipe_submit<unsigned char>(q, n, (const unsigned char *)I0, (const unsigned char *)I1, initflow, outflow, after_stream, ticket, flags)
            : pipe_submit<float>(q, n, (const float *)I0, (const float *)I1, initflow, outflow, after_stream, ticket, flags);
}

int fotg_pipe_wait(fotg_pipe *q, long ticket, void *stream, int host_wait)
{
  if (!q || ticket < 0) return FOTG_ERR_ARG;
  { std::lock_guard<std::mutex> lock(*q->mu); if (ticket >= q->submitted) return FOTG_ERR_ARG; }
  ON_DEVICE(q->device);
  // (an event re-recorded since -- more than 4 * depth tickets ago -- belongs to a later batch of the same slot's stream: waiting for
  // that one covers the ticket)
  const int e = (int)(ticket % q->nring);
  if (host_wait) {
    HIPCHK(hipEventSynchronize(q->done[e]));           // (not under the mutex: submits go on while this thread waits)
    std::lock_guard<std::mutex> lock(*q->mu);
    return pipe_verify(q, ticket, host_wait != 2);
  }
  HIPCHK(hipStreamWaitEvent((hipStream_t)stream, q->done[e], 0));
  {
    // the caller's stream owns the result from here on and may free / reuse the buffers behind this wait: never recompute into them
    std::lock_guard<std::mutex> lock(*q->mu);
    if (ticket >= q->submitted - q->nring) q->args[ticket % q->nring].healable = 0;
  }
  return FOTG_OK;
}

/* the completion event (hipEvent_t) of batch `ticket`: for callers that wait from another thread than the one that submits, or
 * on several pipes at once (fotg_node_wait) -- hipEventSynchronize / hipStreamWaitEvent on it touch no state of the pipe.  Valid
 * for the next 4 * depth submissions; after that it belongs to a later batch of the same slot (waiting for it still covers the
 * ticket). */
int fotg_pipe_ticket_event(fotg_pipe *q, long ticket, void **event)
{
  if (!q || !event || ticket < 0) return FOTG_ERR_ARG;
  std::lock_guard<std::mutex> lock(*q->mu);
  if (ticket >= q->submitted) return FOTG_ERR_ARG;      // (an event of the ring that was never recorded, or belongs to an older batch)
  *event = (void *)q->done[ticket % q->nring];
  if (ticket >= q->submitted - q->nring) q->args[ticket % q->nring].healable = 0;       // (handed out: the pipe cannot know when its buffers go)
  return FOTG_OK;
}

int fotg_pipe_sync(fotg_pipe *q)
{
  if (!q) return FOTG_ERR_ARG;
  ON_DEVICE(q->device);
  for (int k = 0; k < q->depth; ++k) HIPCHK(hipStreamSynchronize(q->stream[k]));
  std::lock_guard<std::mutex> lock(*q->mu);
  int st = FOTG_OK;
  for (int k = 0; k < q->depth; ++k) {
    // the last ticket of slot k (if any): verifying it covers every earlier one of the slot
    long last = q->submitted - 1;
    while (last >= 0 && last % q->depth != k) --last;
    if (last < 0) continue;
    if (last < q->verified[k]) continue;                  // (all verified; a ticket that could not be recomputed keeps its status for whoever waits for it)
    int bad = 0;
    const int sk = pipe_verify(q, last, 1, &bad);
    if (sk != FOTG_OK) st = sk;
    else if (bad) st = FOTG_ERR_STALL;
  }
  return st;
}

int fotg_pipe_context(fotg_pipe *q, int slot, fotg_ctx **ctx)
{
  if (!q || !ctx || slot < 0 || slot >= q->depth) return FOTG_ERR_ARG;
  *ctx = q->ctx[slot];
  return FOTG_OK;
}

/* sequence mode (SURVEY 8f "next" row 2): n_frames consecutive frames -> n_frames - 1 flows (frame k -> k+1) */
int fotg_calc_sequence(fotg_ctx *c, int n_frames, const float *frames, const float *initflow, float *outflow, void *stream)
{
  if (!c || !frames || !outflow) return FOTG_ERR_ARG;
  if (n_frames < 2 || n_frames - 1 > c->max_batch) return FOTG_ERR_BATCH;
  ON_DEVICE(c->device);
  return calc_range<float>(c, n_frames - 1, frames, nullptr, initflow, outflow, (hipStream_t)stream);
}
int fotg_calc_sequence_u8(fotg_ctx *c, int n_frames, const unsigned char *frames, const float *initflow, float *outflow, void *stream)
{
  if (!c || !frames || !outflow) return FOTG_ERR_ARG;
  if (n_frames < 2 || n_frames - 1 > c->max_batch) return FOTG_ERR_BATCH;
  ON_DEVICE(c->device);
  return calc_range<unsigned char>(c, n_frames - 1, frames, nullptr, initflow, outflow, (hipStream_t)stream);
}

long fotg_debug_counter(const char *name)
{
  if (name && !strcmp(name, "sor_stream")) return g_stream_launches;
  if (name && !strcmp(name, "sor_pipe")) return g_pipe_launches;
  if (name && !strcmp(name, "fused_cglobal")) return g_fused_cglobal_launches;
  if (name && !strcmp(name, "sor_tiles")) return g_tile_launches;
  if (name && !strcmp(name, "sor_tall")) return g_tall_launches;
  if (name && !strcmp(name, "level_pipe")) return g_levelpipe_launches;
  return -1;
}

int fotg_set_verbosity(fotg_ctx *c, int verbosity)
{
  if (!c) return FOTG_ERR_ARG;
  c->verbosity = verbosity;
  return FOTG_OK;
}

int fotg_level_timings(fotg_ctx *c, int l, float *ms5)
{
  if (!c || !ms5 || l < c->p.sc_l || l > c->p.sc_f) return FOTG_ERR_ARG;
  memcpy(ms5, c->tt[l], 5 * sizeof(float));
  return FOTG_OK;
}

long fotg_ctx_counter(fotg_ctx *c, const char *name)
{
  if (!c || !name) return -1;
  if (!strcmp(name, "stamps_ptr")) return (long)(size_t)c->stamps;        // -DFOTG_TILE_STATS builds (tools/tile_stats.py)
#ifdef FOTG_DEBUG
  if (!strcmp(name, "guard_violations")) {
    DevGuard dg(c->device);
    if (!dg.ok || hipDeviceSynchronize() != hipSuccess) return -1;
    long bad = 0;
    static unsigned char host[FOTG_GUARD_BYTES];
    for (int k = 0; k < c->nguards; ++k)
      for (int side = 0; side < 2; ++side) {
        if (hipMemcpy(host, side ? c->guards[k].end : c->guards[k].begin, FOTG_GUARD_BYTES, hipMemcpyDeviceToHost) != hipSuccess) return -1;
        int first = -1, last = -1;
        for (int b = 0; b < FOTG_GUARD_BYTES; ++b) if (host[b] != 0xA5) { if (first < 0) first = b; last = b; }
        if (first >= 0) {
          ++bad;
          if (side) printf("guard violated behind %s (%zu bytes): bytes %d..%d past the end\n", c->guards[k].name, c->guards[k].size, first, last);
          else printf("guard violated in front of %s (%zu bytes): bytes %d..%d before the start\n", c->guards[k].name, c->guards[k].size, FOTG_GUARD_BYTES - last, FOTG_GUARD_BYTES - first);
        }
      }
    fflush(stdout);
    return bad;
  }
#endif
  // non-synchronising: host-side count of stalls reported so far + whether the device has flagged one since (meaningful once
  // the caller has synchronised with the context's stream); "inject_stall" sets the word like a timed-out wait would (tests)
  if (!strcmp(name, "stalls")) return c->stalls + (c->stall_host && *(volatile int *)c->stall_host ? 1 : 0);
  // the consuming query for callers of the asynchronous entry points (fotg_calc_batch on their own stream, fotg_pipe_wait with
  // host_wait = 0): AFTER their own synchronisation, 1 = a wait of this context timed out since the last query (the flows
  // computed since then are not valid; counted in "stalls"), 0 = none.  Clears the flag, so a later stall is seen again and an
  // old one is never blamed on a later call.
  if (!strcmp(name, "take_stall")) return stall_status(c) == FOTG_ERR_STALL ? 1 : 0;
  // test tap, only in contexts created with FOTG_TEST_TAPS=1 in the environment
  if (!strcmp(name, "inject_stall")) { if (!c->tune.test_taps) return -1; if (c->stall_host) *(volatile int *)c->stall_host = 1; return 0; }
  if (!strcmp(name, "tile_timeouts")) {
    if (!c->tileSync) return 0;
    DevGuard dg(c->device);
    long tot = 0;
    if (!dg.ok || hipDeviceSynchronize() != hipSuccess) return -1;
    {
      int v = 0;
      if (hipMemcpy(&v, c->tileSync + c->sync_total, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return -1;
      tot += v;
    }
    return tot;
  }
  return -1;
}

int fotg_calc(fotg_ctx *c, const float *I0, const float *I1, const float *initflow, float *outflow_host)
{
  if (!c || !outflow_host) return FOTG_ERR_ARG;
  const LevelGeom &g = c->geom[c->p.sc_l];
  int st = fotg_calc_batch(c, 1, I0, I1, initflow, c->flow[c->p.sc_l], nullptr);
  if (st) return st;
  ON_DEVICE(c->device);
  HIPCHK(hipStreamSynchronize(nullptr));
  if (stall_status(c) == FOTG_ERR_STALL) {
    // a bounded inter-workgroup wait of the tile solver gave up: this call knows its result is wrong and its inputs are still in
    // place -- compute it again on the solver path that has no such waits instead of handing an error to a valid call
    st = recompute_safe<float>(c, 1, I0, I1, initflow, c->flow[c->p.sc_l], nullptr);
    if (st != FOTG_OK) return st;
  }
  HIPCHK(hipMemcpy(outflow_host, c->flow[c->p.sc_l], (size_t)g.w * g.h * c->nch * sizeof(float), hipMemcpyDeviceToHost));
  return FOTG_OK;
}

int fotg_upsample_crop(fotg_ctx *c, int n, const float *flow, float *out, void *stream)
{
  if (!c || !flow || !out) return FOTG_ERR_ARG;
  if (n < 1 || n > c->max_batch) return FOTG_ERR_BATCH;
  ON_DEVICE(c->device);
  const LevelGeom &g = c->geom[c->p.sc_l];
  // four pixels per thread on a (columns, row, pair) grid
  const int tpr = (c->w_org + 3) / 4;                                 // threads per row
  const int bx = tpr >= 256 ? 256 : ((tpr + 63) / 64) * 64;
  dim3 grid((tpr + bx - 1) / bx, c->h_org, n), block(bx);
  if (c->h_org <= 65535 && n <= 65535) {
    if (c->nch == 2) upsample_crop4_kernel<2><<<grid, block, 0, (hipStream_t)stream>>>(flow, (long)g.w * g.h * 2, g.w, g.h, c->p.sc_l, c->padw / 2, c->padh / 2,
                                                                                        c->w_org, c->h_org, out, (long)c->w_org * c->h_org * 2);
    else upsample_crop4_kernel<1><<<grid, block, 0, (hipStream_t)stream>>>(flow, (long)g.w * g.h, g.w, g.h, c->p.sc_l, c->padw / 2, c->padh / 2,
                                                                            c->w_org, c->h_org, out, (long)c->w_org * c->h_org);
  } else {
    dim3 grid1((c->w_org * c->h_org + 255) / 256, n), block1(256);
    upsample_crop_kernel<<<grid1, block1, 0, (hipStream_t)stream>>>(flow, (long)g.w * g.h * c->nch, g.w, g.h, c->p.sc_l, c->padw / 2, c->padh / 2,
                                                                     c->w_org, c->h_org, out, (long)c->w_org * c->h_org * c->nch, c->nch);
  }
  LAUNCHCHK();
  return FOTG_OK;
}

extern "C++" {
template <typename T>
static int gradmag_impl(int device, int n, const T *frames, int w_org, int h_org, int channels, int sc_f, float *out, void *stream)
{
  if (!frames || !out || n < 1 || (channels != 1 && channels != 3)) return FOTG_ERR_ARG;
  int Wp, Hp, padw, padh;
  const int st = fotg_padded_size(w_org, h_org, sc_f, &Wp, &Hp, &padw, &padh);
  if (st != FOTG_OK) return st;
  if (Wp < 2 || Hp < 2) return FOTG_ERR_ARG;
  ON_DEVICE(device);
  const long per = (long)Wp * Hp * channels;
  gradmag_kernel<T><<<dim3((unsigned)((per + 255) / 256), n), 256, 0, (hipStream_t)stream>>>(frames, (long)w_org * h_org * channels, w_org, h_org, channels,
                                                                                            padw / 2, padh / 2, Wp, Hp, out);
  LAUNCHCHK();
  return FOTG_OK;
}
}  // extern "C++"

int fotg_gradient_magnitude(int device, int n, const float *frames, int w_org, int h_org, int channels, int sc_f, float *out, void *stream)
{
  return gradmag_impl<float>(device, n, frames, w_org, h_org, channels, sc_f, out, stream);
}
int fotg_gradient_magnitude_u8(int device, int n, const unsigned char *frames, int w_org, int h_org, int channels, int sc_f, float *out, void *stream)
{
  return gradmag_impl<unsigned char>(device, n, frames, w_org, h_org, channels, sc_f, out, stream);
}

}  // extern "C"
