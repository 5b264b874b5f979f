// include/fotg/params.h -- the reference's parameter structs (src/params.h:9-65), same names and meaning.
// Differences: cublasHandle_t is gone (the reference creates the handle and never issues a BLAS call,
// src/oflow.cpp:58,149); `channels` (1 gray / 3 interleaved RGB) and `sor_mode` are added.
#ifndef FOTG_OFC_PARAMS_HEADER
#define FOTG_OFC_PARAMS_HEADER
#include "../fotg.h"

namespace OFC {

typedef struct {
  int width;              // image width, without '2 * padding' (src/params.h:10)
  int height;
  int padding;            // = patch_size (src/run_dense.cpp:263)
  float l_bound;
  float u_bound_width;
  float u_bound_height;
  int width_pad;
  int height_pad;
  float scale_fact;
  int curr_lvl;
} img_params;

typedef struct {
  int coarsest_scale;
  int finest_scale;
  int patch_size;
  float patch_stride;
  bool use_mean_normalization;
  int grad_descent_iter;
  float dp_thresh;        // 0.05 (src/oflow.cpp:53 stores the square; squared inside the library)
  float dr_thresh;        // 0.95
  float res_thresh;       // 0.0
  int verbosity;
  bool use_var_ref;
  int var_ref_iter;
  float var_ref_alpha;
  float var_ref_gamma;
  float var_ref_delta;
  float var_ref_sor_weight;
  // automatically set (src/oflow.cpp:45-48)
  float outlier_thresh;
  int steps;
  int n_vals;
  int n_scales;
  float min_errval = 2.0f;
  float norm_outlier = 5.0f;
  // additions
  int channels = 3;       // the reference's CUDA port is RGB only (src/run_dense.cpp:147)
  int sor_mode = FOTG_SOR_LEXICOGRAPHIC;
  int cost_func = 0;      // kroeger/oflow.h:45: 0 L2, 1 L1, 2 pseudo-Huber (threshold norm_outlier)
  bool use_fbcon = false; // kroeger/oflow.h:44 usefbcon: forward-backward merge in the densification
  bool depth_mode = false; // kroeger SELECTMODE=2 build (run_DE_*): stereo depth, flow arrays have ONE channel
  int min_iter = -1; // kroeger optparam.min_iter (oflow.h:38); < 0: = grad_descent_iter (src/ and the operating points)
  int u8_color = 0;  // channels = 1 only: the 8-bit entry points take 3-channel frames (1: B,G,R as cv::imread delivers, 2: R,G,B) and
                     // convert to gray on load like cv::imread(IMREAD_GRAYSCALE) (kroeger/run_dense.cpp:199-209); fotg_params::u8_color
  int var_ref_inner_iter = 1; // kroeger tv_innerit (oflow.h:50, run_dense.cpp:288): inner fixed-point iterations = var_ref_inner_iter * (level + 1)
                     // (refine_variational.cpp:36); src/ hard-codes 1 (src/refine_variational.cpp:41)
  bool fast_math = false; // tolerance mode of the patch loop and the refinement's arithmetic (fotg_params::fast_math): flows within 1e-3 px (mean) of the parity mode
} opt_params;

inline fotg_params to_fotg(const opt_params &op)
{
  fotg_params p = fotg_params();
  p.sc_f = op.coarsest_scale; p.sc_l = op.finest_scale; p.ps = op.patch_size;
  p.max_iter = p.min_iter = op.grad_descent_iter;          // src/kernels/optimize.cu:225-229
  if (op.min_iter >= 0 && op.min_iter <= op.grad_descent_iter) p.min_iter = op.min_iter;   // kroeger early termination (patch.cpp:279-282)
  p.dp_thresh = op.dp_thresh > 0 ? op.dp_thresh : 0.05f;
  p.dr_thresh = op.dr_thresh > 0 ? op.dr_thresh : 0.95f;
  p.res_thresh = op.res_thresh;
  p.patove = op.patch_stride; p.patnorm = op.use_mean_normalization; p.noc = op.channels;
  p.usetvref = op.use_var_ref; p.tv_alpha = op.var_ref_alpha; p.tv_gamma = op.var_ref_gamma; p.tv_delta = op.var_ref_delta;
  p.tv_innerit = op.var_ref_inner_iter; p.tv_solverit = op.var_ref_iter; p.tv_sor = op.var_ref_sor_weight; p.sor_mode = op.sor_mode;
  p.costfct = op.cost_func; p.normoutlier = op.norm_outlier; p.usefbcon = op.use_fbcon; p.depth = op.depth_mode;
  p.u8_color = op.u8_color;
  p.fast_math = op.fast_math;
  return p;
}

}  // namespace OFC
#endif
