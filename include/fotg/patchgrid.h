// include/fotg/patchgrid.h -- PatGridClass of the reference (src/patchgrid.h:13-86) over the C-ABI.
// A grid belongs to one scale of one OFClass; all pointers are device pointers to padded level images
// ((h+2*padding) x (w+2*padding) x channels), exactly what the reference passes (src/oflow.cpp:250-251).
#ifndef FOTG_PATGRID_HEADER
#define FOTG_PATGRID_HEADER
#include <cstdio>
#include <cstdlib>
#include "params.h"

namespace OFC {

// reference behaviour on a device error: print, reset, exit (src/common/cuda_helper.h:286-299)
inline void fotgCheck(int status, const char *what)
{
  if (status != FOTG_OK) {
    fprintf(stderr, "fotg error in %s: %s (status %d, hip error %d)\n", what, fotg_strerror(status), status, fotg_last_hip_error());
    exit(EXIT_FAILURE);
  }
}

class PatGridClass {
 public:
  PatGridClass(fotg_ctx *ctx, const img_params *_i_params, const opt_params *_op)
      : ctx_(ctx), i_params(_i_params), op(_op)
  {
    fotgCheck(fotg_num_patches(ctx_, i_params->curr_lvl, &n_patches_width, &n_patches_height), "PatGridClass");
    n_patches = n_patches_width * n_patches_height;
    stride_ = (long)i_params->width_pad * i_params->height_pad * op->channels;
  }
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
  void printTimings() { printf("[timings] per-kernel times: rocprofv3 --kernel-trace --stats\n"); }

 private:
  fotg_ctx *ctx_;
  const img_params *i_params;
  const opt_params *op;
  int n_patches_width, n_patches_height, n_patches;
  long stride_;
};

}  // namespace OFC
#endif
