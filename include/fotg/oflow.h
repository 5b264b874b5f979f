// include/fotg/oflow.h -- OFClass of the reference (src/oflow.h:22-46) over the C-ABI of libfotg.so.
//
//   OFC::OFClass ofc(op, iparams);                       // src/run_dense.cpp:277
//   ofc.calc(I0, I1, iparams, nullptr, outflow);         // src/run_dense.cpp:286
//
// I0/I1: device pointers, interleaved float32, iparams.width x iparams.height x op.channels (padded by the
// caller as in src/run_dense.cpp:231-253, or unpadded -- the library folds the padding into its pyramid
// kernel).  outflow: HOST buffer of 2*(W/2^finest)*(H/2^finest) floats like the reference's mapped buffer
// (src/run_dense.cpp:280-289).  Errors print and exit like checkCudaErrors (src/common/cuda_helper.h:286-299).
#ifndef FOTG_OFC_HEADER
#define FOTG_OFC_HEADER
#include <algorithm>
#include <cmath>
#include <vector>
#include "params.h"
#include "patchgrid.h"
#include "refine_variational.h"

namespace OFC {

class OFClass {
 public:
  OFClass(opt_params _op, img_params _i_params, int max_batch = 1, int device = 0) : op(_op)
  {
    op.outlier_thresh = (float)op.patch_size / 2;                                         // src/oflow.cpp:45-48
    op.steps = std::max(1, (int)floor(op.patch_size * (1 - op.patch_stride)));
    op.n_vals = op.channels * op.patch_size * op.patch_size;
    op.n_scales = op.coarsest_scale - op.finest_scale + 1;
    fotg_params p = to_fotg(op);
    fotgCheck(fotg_create(&p, _i_params.width, _i_params.height, device, max_batch, &ctx), "OFClass");
    fotgCheck(fotg_set_verbosity(ctx, op.verbosity), "OFClass");
    ContextRegistry::get().add(&op, ctx);                 // every grid / VarRefClass built from &op finds this context
    int Wp, Hp;
    fotg_padded_size(_i_params.width, _i_params.height, op.coarsest_scale, &Wp, &Hp, nullptr, nullptr);
    iparams.resize(op.n_scales);
    grid.resize(op.n_scales);
    for (int sl = op.coarsest_scale; sl >= op.finest_scale; --sl) {                       // src/oflow.cpp:80-102
      int i = sl - op.finest_scale;
      iparams[i].scale_fact = (float)pow(2, -sl);
      iparams[i].height = Hp >> sl;
      iparams[i].width = Wp >> sl;
      iparams[i].padding = op.patch_size;
      iparams[i].l_bound = -(float)op.patch_size / 2;
      iparams[i].u_bound_width = (float)(iparams[i].width + op.patch_size / 2 - 2);
      iparams[i].u_bound_height = (float)(iparams[i].height + op.patch_size / 2 - 2);
      iparams[i].width_pad = iparams[i].width + 2 * op.patch_size;
      iparams[i].height_pad = iparams[i].height + 2 * op.patch_size;
      iparams[i].curr_lvl = sl;
    }
    for (int i = 0; i < op.n_scales; ++i) grid[i] = new PatGridClass(&iparams[i], &op);   // src/oflow.cpp:101
  }
  ~OFClass()
  {
    for (auto g : grid) delete g;
    ContextRegistry::get().remove(&op);
    fotg_destroy(ctx);
  }
  OFClass(const OFClass &) = delete;
  OFClass &operator=(const OFClass &) = delete;

  // src/oflow.cpp:211-368 (with op.verbosity > 0 the library prints the reference's TIME lines; > 1 also the grids' tables,
  // src/oflow.cpp:362-366)
  void calc(const float *_I0, const float *_I1, img_params /*_iparams*/, const float *initflow, float *outflow)
  {
    fotgCheck(fotg_calc(ctx, _I0, _I1, initflow, outflow), "OFClass::calc");
    if (op.verbosity > 1)
      for (auto &g : grid) g->printTimings();
  }
  // n pairs, device output, asynchronous on `stream` (hipStream_t)
  void calc_batch(int n, const float *_I0, const float *_I1, const float *initflow, float *outflow_dev, void *stream = nullptr)
  { fotgCheck(fotg_calc_batch(ctx, n, _I0, _I1, initflow, outflow_dev, stream), "OFClass::calc_batch"); }
  // video: n_frames consecutive frames -> n_frames - 1 flows, every frame's pyramid built once
  void calc_sequence(int n_frames, const float *frames, const float *initflow, float *outflow_dev, void *stream = nullptr)
  { fotgCheck(fotg_calc_sequence(ctx, n_frames, frames, initflow, outflow_dev, stream), "OFClass::calc_sequence"); }

  fotg_ctx *handle() { return ctx; }
  PatGridClass *GetGrid(int scale) { return grid[scale - op.finest_scale]; }
  const img_params &GetImgParams(int scale) const { return iparams[scale - op.finest_scale]; }
  const opt_params &GetOptParams() const { return op; }     // the opt_params every grid / VarRefClass of this object is built from

 private:
  opt_params op;
  std::vector<img_params> iparams;
  std::vector<PatGridClass *> grid;
  fotg_ctx *ctx = nullptr;
};

}  // namespace OFC
#endif
