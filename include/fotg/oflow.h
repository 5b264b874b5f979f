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
    for (int i = 0; i < op.n_scales; ++i) grid[i] = new PatGridClass(ctx, &iparams[i], &op);
  }
  ~OFClass()
  {
    for (auto g : grid) delete g;
    fotg_destroy(ctx);
  }
  OFClass(const OFClass &) = delete;
  OFClass &operator=(const OFClass &) = delete;

  // src/oflow.cpp:211-368
  void calc(const float *_I0, const float *_I1, img_params /*_iparams*/, const float *initflow, float *outflow)
  { fotgCheck(fotg_calc(ctx, _I0, _I1, initflow, outflow), "OFClass::calc"); }
  // n pairs, device output, asynchronous on `stream` (hipStream_t)
  void calc_batch(int n, const float *_I0, const float *_I1, const float *initflow, float *outflow_dev, void *stream = nullptr)
  { fotgCheck(fotg_calc_batch(ctx, n, _I0, _I1, initflow, outflow_dev, stream), "OFClass::calc_batch"); }
  // video: n_frames consecutive frames -> n_frames - 1 flows, every frame's pyramid built once
  void calc_sequence(int n_frames, const float *frames, const float *initflow, float *outflow_dev, void *stream = nullptr)
  { fotgCheck(fotg_calc_sequence(ctx, n_frames, frames, initflow, outflow_dev, stream), "OFClass::calc_sequence"); }

  fotg_ctx *handle() { return ctx; }
  PatGridClass *GetGrid(int scale) { return grid[scale - op.finest_scale]; }
  const img_params &GetImgParams(int scale) const { return iparams[scale - op.finest_scale]; }

 private:
  opt_params op;
  std::vector<img_params> iparams;
  std::vector<PatGridClass *> grid;
  fotg_ctx *ctx = nullptr;
};

// src/refine_variational.h:35-57: the constructor does all the work, in place on flowout (device pointer)
class VarRefClass {
 public:
  VarRefClass(OFClass &ofc, const float *_I0, const float *_I1, const img_params *_i_params, const opt_params *_op, float *flowout)
  {
    const long stride = (long)_i_params->width_pad * _i_params->height_pad * _op->channels;
    fotgCheck(fotg_varref(ofc.handle(), _i_params->curr_lvl, 1, _I0, _I1, stride, flowout, nullptr), "VarRefClass");
  }
};

}  // namespace OFC
#endif
