// include/fotg/patchgrid.h -- PatGridClass of the reference (src/patchgrid.h:13-86) over the C-ABI, with the reference's
// constructor signature: PatGridClass(const img_params*, const opt_params*) (src/patchgrid.h:16), as src/oflow.cpp:101 calls it.
//
// A grid belongs to one scale of one engine context.  The reference passes no context: the owning OFClass registers the
// address of ITS opt_params (the pointer src/oflow.cpp:101 and :332 hand to every grid and to VarRefClass) in a process-wide
// registry, and the grid finds its context there.  A grid built from parameters no OFClass owns (a unit test of one scale)
// creates a private single-scale context for that level.
// All pointers are device pointers to padded level images ((h+2*padding) x (w+2*padding) x channels), exactly what the
// reference passes (src/oflow.cpp:250-251).
#ifndef FOTG_PATGRID_HEADER
#define FOTG_PATGRID_HEADER
#include <cstdio>
#include <cstdlib>
#include <map>
#include <mutex>
#include <vector>
#include "params.h"
#include "patch.h"

namespace OFC {

// reference behaviour on a device error: print, reset, exit (src/common/cuda_helper.h:286-299)
inline void fotgCheck(int status, const char *what)
{
  if (status != FOTG_OK) {
    fprintf(stderr, "fotg error in %s: %s (status %d, hip error %d)\n", what, fotg_strerror(status), status, fotg_last_hip_error());
    exit(EXIT_FAILURE);
  }
}

// opt_params* (as handed out by an OFClass) -> engine context
class ContextRegistry {
 public:
  static ContextRegistry &get() { static ContextRegistry r; return r; }
  void add(const opt_params *op, fotg_ctx *ctx) { std::lock_guard<std::mutex> g(m_); map_[op] = ctx; }
  void remove(const opt_params *op) { std::lock_guard<std::mutex> g(m_); map_.erase(op); }
  fotg_ctx *find(const opt_params *op) { std::lock_guard<std::mutex> g(m_); auto it = map_.find(op); return it == map_.end() ? nullptr : it->second; }
 private:
  std::mutex m_;
  std::map<const opt_params *, fotg_ctx *> map_;
};

// the context of (i_params, op): the registered one, or a new single-scale context for this level (owned by the caller)
inline fotg_ctx *fotgContextFor(const img_params *ip, const opt_params *op, bool *owned)
{
  *owned = false;
  if (fotg_ctx *c = ContextRegistry::get().find(op)) return c;
  opt_params o = *op;
  o.coarsest_scale = o.finest_scale = ip->curr_lvl;
  fotg_params p = to_fotg(o);
  fotg_ctx *c = nullptr;
  fotgCheck(fotg_create(&p, ip->width << ip->curr_lvl, ip->height << ip->curr_lvl, 0, 1, &c), "single-scale context");
  *owned = true;
  return c;
}

class PatGridClass {
 public:
  PatGridClass(const img_params *_i_params, const opt_params *_op) : i_params(_i_params), op(_op)
  {
    ctx_ = fotgContextFor(i_params, op, &owned_);
    fotgCheck(fotg_num_patches(ctx_, i_params->curr_lvl, &n_patches_width, &n_patches_height), "PatGridClass");
    n_patches = n_patches_width * n_patches_height;
    stride_ = (long)i_params->width_pad * i_params->height_pad * op->channels;
  }
  ~PatGridClass() { if (owned_) fotg_destroy(ctx_); }
  PatGridClass(const PatGridClass &) = delete;
  PatGridClass &operator=(const PatGridClass &) = delete;

  void InitializeGrid(const float *_I0, const float *_I0x, const float *_I0y)
  { fotgCheck(fotg_grid_init(ctx_, i_params->curr_lvl, 1, _I0, _I0x, _I0y, stride_, nullptr), "InitializeGrid"); }
  void SetTargetImage(const float *_I1) { fotgCheck(fotg_grid_set_target(ctx_, i_params->curr_lvl, _I1, stride_), "SetTargetImage"); }
  void InitializeFromCoarserOF(const float *flow_prev)
  { fotgCheck(fotg_grid_init_from_coarser(ctx_, i_params->curr_lvl, 1, flow_prev, nullptr), "InitializeFromCoarserOF"); }
  void Optimize() { fotgCheck(fotg_grid_optimize(ctx_, i_params->curr_lvl, 1, nullptr), "Optimize"); }
  void AggregateFlowDense(float *flowout) { fotgCheck(fotg_grid_aggregate(ctx_, i_params->curr_lvl, 1, flowout, nullptr), "AggregateFlowDense"); }

  inline int GetNumPatches() const { return n_patches; }
  inline int GetNumPatchesW() const { return n_patches_width; }
  inline int GetNumPatchesH() const { return n_patches_height; }
  // src/patchgrid.cpp:54-63 (the reference returns an Eigen::Vector2f; plain floats here)
  inline void GetRefPatchPos(int i, float *x, float *y) const
  {
    const int offw = (i_params->width - (n_patches_width - 1) * op->steps) / 2;
    const int offh = (i_params->height - (n_patches_height - 1) * op->steps) / 2;
    *x = (float)((i / n_patches_height) * op->steps + offw);
    *y = (float)((i % n_patches_height) * op->steps + offh);
  }
  // the reference's per-patch record (src/patch.h:15-36) after Optimize(), on the host: reference position, displacement and --
  // if fotg_enable_taps(ctx, 1) was called before -- Hessian and iteration count
  void GetPatchStates(std::vector<dev_patch_state> &st, int pair = 0) const
  {
    st.assign(n_patches, dev_patch_state());
    std::vector<float> p(2 * (size_t)n_patches), hes(3 * (size_t)n_patches, 0.f);
    std::vector<int> cnt(n_patches, 0);
    if (fotg_grid_read(ctx_, i_params->curr_lvl, pair, p.data(), nullptr, nullptr, nullptr, nullptr, hes.data(), cnt.data()) != FOTG_OK)
      fotgCheck(fotg_grid_read(ctx_, i_params->curr_lvl, pair, p.data(), nullptr, nullptr, nullptr, nullptr, nullptr, nullptr), "GetPatchStates");
    for (int i = 0; i < n_patches; ++i) {
      dev_patch_state &s = st[i];
      float x, y;
      GetRefPatchPos(i, &x, &y);
      s.has_converged = true; s.has_opt_started = true;
      s.H00 = hes[3 * i]; s.H01 = hes[3 * i + 1]; s.H11 = hes[3 * i + 2];
      s.p_curx = p[2 * i]; s.p_cury = p[2 * i + 1];
      s.midpoint_orgx = x; s.midpoint_orgy = y;
      s.midpoint_curx = x + s.p_curx; s.midpoint_cury = y + s.p_cury;
      s.count = cnt[i];
    }
  }
  // src/patchgrid.cpp:334-345, from the GPU times of the last flow call with verbosity > 0 (fotg_set_verbosity): patch
  // extraction and the initialisation from the coarser flow are part of the LK launch
  void printTimings()
  {
    float t[5] = {0, 0, 0, 0, 0};
    (void)fotg_level_timings(ctx_, i_params->curr_lvl, t);
    printf("\n===============Timings (ms)===============\n");
    printf("[extract]      %g\n[coarse]       %g\n[optiTime]      %g\n[aggregate]    %g\n[flow norm]    %g\n", t[0], t[1], t[2], t[3], 0.0);
    printf("==========================================\n");
  }
  fotg_ctx *handle() const { return ctx_; }

 private:
  fotg_ctx *ctx_;
  bool owned_ = false;
  const img_params *i_params;
  const opt_params *op;
  int n_patches_width, n_patches_height, n_patches;
  long stride_;
};

}  // namespace OFC
#endif
