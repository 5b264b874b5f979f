// include/fotg/refine_variational.h -- VarRefClass of the reference (src/refine_variational.h:35-57) over the C-ABI, with the
// reference's constructor signature (src/refine_variational.h:38-39), as src/oflow.cpp:332 calls it: the constructor does all
// the work, in place on flowout.  _I0 / _I1: padded level images ((h+2*padding) x (w+2*padding) x channels), flowout
// (h x w x 2) -- DEVICE pointers (the reference copies the level images to the host first, src/oflow.cpp:320-330, because its
// refinement runs on the CPU; here it runs on the GPU).  The context comes from the registry of patchgrid.h.
#ifndef FOTG_VARREF_HEADER
#define FOTG_VARREF_HEADER
#include "patchgrid.h"

namespace OFC {

class VarRefClass {
 public:
  VarRefClass(const float *_I0, const float *_I1, const img_params *_i_params, const opt_params *_op, float *flowout)
  {
    bool owned;
    fotg_ctx *ctx = fotgContextFor(_i_params, _op, &owned);
    const long stride = (long)_i_params->width_pad * _i_params->height_pad * _op->channels;
    const int st = fotg_varref(ctx, _i_params->curr_lvl, 1, _I0, _I1, stride, flowout, nullptr);
    if (owned) {                                  // a private context: wait for the launch before it goes away
      fotg_grid_read(ctx, _i_params->curr_lvl, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
      fotg_destroy(ctx);
    }
    fotgCheck(st, "VarRefClass");
  }
  ~VarRefClass() {}
};

}  // namespace OFC
#endif
