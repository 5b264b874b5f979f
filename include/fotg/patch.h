// include/fotg/patch.h -- dev_patch_state of the reference (src/patch.h:15-36), same fields and defaults.
// The engine keeps the per-patch state of a whole grid in registers for the life of one LK launch (one wave per
// 1 / 2 / 4 patches, flowonthego_amd/csrc/lk.hip.h); PatGridClass::GetPatchStates() fills this struct on the host from
// what the launch leaves behind (displacement, Hessian, iteration count) for callers that inspect it.
#ifndef FOTG_PAT_HEADER
#define FOTG_PAT_HEADER
#include "params.h"

namespace OFC {

typedef struct {
  bool has_converged;
  bool has_opt_started;

  float H00, H01, H11;
  float p_orgx, p_orgy;
  float p_curx, p_cury;
  float delta_px, delta_py;

  // start positions, current point position, patch norm
  float midpoint_curx, midpoint_cury;
  float midpoint_orgx, midpoint_orgy;

  float delta_p_sq_norm = 1e-10;
  float delta_p_sq_norm_init = 1e-10;
  float mares = 1e20;  // mares: Mean Absolute RESidual
  float mares_old = 1e20;
  int count = 0;
  bool invalid = false;

  float cost = 0.0;
} dev_patch_state;

}  // namespace OFC
#endif
