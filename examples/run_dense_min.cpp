// examples/run_dense_min.cpp -- the shape of the reference's run_dense (src/run_dense.cpp:120-305) over the C++ shim:
// two raw float32 frames in, one Middlebury .flo out.  No OpenCV: frames are raw interleaved float32 (w*h*channels), the way
// the reference holds them after cv::imread + convertTo(CV_32F) (src/run_dense.cpp:137-145).
//
//   hipcc -O2 -Iinclude examples/run_dense_min.cpp -Lflowonthego_amd -lfotg -Wl,-rpath,$PWD/flowonthego_amd -o examples/run_dense_min
//   examples/run_dense_min frame0.raw frame1.raw W H C out.flo [op-point 1..4] [depth] [innerit=N]
// innerit=N: kroeger's tv_innerit command-line parameter (kroeger/run_dense.cpp:288; inner iterations = N * (level + 1)), default 1
// With the 8th argument "depth" it is the reference's run_DE_* binary instead (kroeger SELECTMODE=2): a rectified stereo pair in,
// one displacement channel out, written as a PFM file (SavePFMFile, kroeger/run_dense.cpp:60-81).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string.h>
#include "fotg/oflow.h"
#include "fotg/flowio.h"

static std::vector<float> read_raw(const char *path, size_t n)
{
  std::vector<float> v(n);
  FILE *f = fopen(path, "rb");
  if (!f || fread(v.data(), sizeof(float), n, f) != n) { fprintf(stderr, "cannot read %zu floats from %s\n", n, path); exit(1); }
  fclose(f);
  return v;
}

int main(int argc, char **argv)
{
  if (argc < 7) { fprintf(stderr, "usage: %s frame0.raw frame1.raw W H C out.flo|out.pfm [op-point] [depth] [innerit=N]\n", argv[0]); return 2; }
  const int W = atoi(argv[3]), H = atoi(argv[4]), C = atoi(argv[5]), oppt = argc > 7 ? atoi(argv[7]) : 2;
  bool depth = false;
  int innerit = 1;
  for (int k = 8; k < argc; ++k) {
    if (!strcmp(argv[k], "depth")) depth = true;
    else if (!strncmp(argv[k], "innerit=", 8)) innerit = atoi(argv[k] + 8);
  }
  const int nch = depth ? 1 : 2;
  const size_t n = (size_t)W * H * C;
  const std::vector<float> f0 = read_raw(argv[1], n), f1 = read_raw(argv[2], n);

  // operating point (src/run_dense.cpp:166-227) -- evaluated by the library, copied into the reference's struct
  fotg_params p;
  OFC::fotgCheck(fotg_op_point(oppt, W, C, &p), "fotg_op_point");
  OFC::opt_params op;
  op.coarsest_scale = p.sc_f; op.finest_scale = p.sc_l; op.patch_size = p.ps; op.patch_stride = p.patove;
  op.use_mean_normalization = p.patnorm != 0; op.grad_descent_iter = p.max_iter;
  op.dp_thresh = p.dp_thresh; op.dr_thresh = p.dr_thresh; op.res_thresh = p.res_thresh;
  op.use_var_ref = p.usetvref != 0; op.var_ref_iter = p.tv_solverit; op.var_ref_alpha = p.tv_alpha; op.var_ref_gamma = p.tv_gamma;
  op.var_ref_delta = p.tv_delta; op.var_ref_sor_weight = p.tv_sor; op.verbosity = 0; op.channels = C;
  op.depth_mode = depth;
  op.var_ref_inner_iter = innerit;                                                  // kroeger/run_dense.cpp:288 -> refine_variational.cpp:36
  OFC::img_params iparams;
  iparams.width = W; iparams.height = H; iparams.padding = op.patch_size;       // unpadded: the library pads inside its pyramid kernel

  float *d0 = nullptr, *d1 = nullptr, *dflow = nullptr, *dfull = nullptr;
  if (hipMalloc(&d0, n * 4) != hipSuccess || hipMalloc(&d1, n * 4) != hipSuccess) { fprintf(stderr, "hipMalloc failed\n"); return 1; }
  hipMemcpy(d0, f0.data(), n * 4, hipMemcpyHostToDevice);
  hipMemcpy(d1, f1.data(), n * 4, hipMemcpyHostToDevice);

  OFC::OFClass ofc(op, iparams);                                                    // src/run_dense.cpp:277
  int ow, oh;
  OFC::fotgCheck(fotg_out_size(ofc.handle(), &ow, &oh), "fotg_out_size");
  std::vector<float> coarse((size_t)nch * ow * oh);
  ofc.calc(d0, d1, iparams, nullptr, coarse.data());                                // src/run_dense.cpp:286

  // post-processing of src/run_dense.cpp:293-303 on the device: x 2^finest, bilinear upsample, crop the padding
  hipMalloc(&dflow, coarse.size() * 4);
  hipMalloc(&dfull, (size_t)nch * W * H * 4);
  hipMemcpy(dflow, coarse.data(), coarse.size() * 4, hipMemcpyHostToDevice);
  OFC::fotgCheck(fotg_upsample_crop(ofc.handle(), 1, dflow, dfull, nullptr), "fotg_upsample_crop");
  std::vector<float> full((size_t)nch * W * H);
  hipMemcpy(full.data(), dfull, full.size() * 4, hipMemcpyDeviceToHost);
  // SavePFMFile / SaveFlowFile of the reference (kroeger/run_dense.cpp:16-81), include/fotg/flowio.h
  if (!(depth ? OFC::SavePFMFile(full.data(), W, H, argv[6]) : OFC::SaveFlowFile(full.data(), W, H, argv[6]))) { fprintf(stderr, "cannot write %s\n", argv[6]); return 1; }
  printf("%s: %dx%d %s written (finest scale %dx%d, op-point %d)\n", argv[6], W, H, depth ? "disparity" : "flow", ow, oh, oppt);
  hipFree(d0); hipFree(d1); hipFree(dflow); hipFree(dfull);
  return 0;
}
