// examples/oflow_scale_loop.cpp -- the reference's scale loop (src/oflow.cpp:240-345) written against the shim classes, with the
// constructors called EXACTLY as the reference calls them:
//     new OFC::PatGridClass(&(iparams[i]), &op)                                    src/oflow.cpp:101
//     OFC::VarRefClass var_ref(I0, I1, &(iparams[ii]), &op, out_ptr)               src/oflow.cpp:332
// i.e. a reference-side file that builds its own grids and refinement objects compiles and runs unchanged against
// include/fotg/.  The result is compared with OFClass::calc on the same frames (the test also compares with the oracle).
//
//   hipcc -O2 -Iinclude examples/oflow_scale_loop.cpp -Lflowonthego_amd -lfotg -Wl,-rpath,$PWD/flowonthego_amd -o oflow_scale_loop
//   oflow_scale_loop frame0.raw frame1.raw W H C out_coarse.raw [verbosity]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "fotg/oflow.h"
#include "fotg/patch.h"

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
  if (argc < 7) { fprintf(stderr, "usage: %s frame0.raw frame1.raw W H C out.raw [verbosity]\n", argv[0]); return 2; }
  const int W = atoi(argv[3]), H = atoi(argv[4]), C = atoi(argv[5]);
  const size_t n = (size_t)W * H * C;
  const std::vector<float> f0 = read_raw(argv[1], n), f1 = read_raw(argv[2], n);
  fotg_params p;
  OFC::fotgCheck(fotg_op_point(2, W, C, &p), "fotg_op_point");
  OFC::opt_params op;
  op.coarsest_scale = p.sc_f; op.finest_scale = p.sc_l; op.patch_size = p.ps; op.patch_stride = p.patove;
  op.use_mean_normalization = p.patnorm != 0; op.grad_descent_iter = p.max_iter;
  op.dp_thresh = p.dp_thresh; op.dr_thresh = p.dr_thresh; op.res_thresh = p.res_thresh;
  op.use_var_ref = p.usetvref != 0; op.var_ref_iter = p.tv_solverit; op.var_ref_alpha = p.tv_alpha; op.var_ref_gamma = p.tv_gamma;
  op.var_ref_delta = p.tv_delta; op.var_ref_sor_weight = p.tv_sor; op.verbosity = argc > 7 ? atoi(argv[7]) : 0; op.channels = C;
  OFC::img_params ip0;
  ip0.width = W; ip0.height = H; ip0.padding = op.patch_size;

  float *d0 = nullptr, *d1 = nullptr;
  if (hipMalloc(&d0, n * 4) != hipSuccess || hipMalloc(&d1, n * 4) != hipSuccess) return 1;
  hipMemcpy(d0, f0.data(), n * 4, hipMemcpyHostToDevice);
  hipMemcpy(d1, f1.data(), n * 4, hipMemcpyHostToDevice);

  OFC::OFClass ofc(op, ip0);
  int ow, oh;
  OFC::fotgCheck(fotg_out_size(ofc.handle(), &ow, &oh), "fotg_out_size");
  std::vector<float> whole((size_t)2 * ow * oh);
  ofc.calc(d0, d1, ip0, nullptr, whole.data());                                     // the library's own scale loop

  // ---- the same loop by hand, with the reference's object constructions.  `opr` is the opt_params the OFClass handed out
  // (src/oflow.cpp keeps it as the member `op`; here the accessor of the shim).
  const OFC::opt_params &opr = ofc.GetOptParams();
  const int ns = opr.n_scales;
  std::vector<OFC::PatGridClass *> grid(ns);
  for (int i = 0; i < ns; ++i) grid[i] = new OFC::PatGridClass(&ofc.GetImgParams(opr.finest_scale + i), &opr);          // src/oflow.cpp:101
  OFC::fotgCheck(fotg_pyramid_pair(ofc.handle(), 1, d0, d1, 3, nullptr), "pyramid");                                   // ConstructImgPyramids
  std::vector<float *> flow(ns, nullptr);
  for (int sl = opr.coarsest_scale; sl >= opr.finest_scale; --sl) {
    const int ii = sl - opr.finest_scale;
    const OFC::img_params &ipl = ofc.GetImgParams(sl);
    float *I0, *I0x, *I0y, *I1;
    long stride;
    OFC::fotgCheck(fotg_level_ptr(ofc.handle(), 0, sl, 0, &I0, &stride), "level");
    OFC::fotgCheck(fotg_level_ptr(ofc.handle(), 0, sl, 1, &I0x, &stride), "level");
    OFC::fotgCheck(fotg_level_ptr(ofc.handle(), 0, sl, 2, &I0y, &stride), "level");
    OFC::fotgCheck(fotg_level_ptr(ofc.handle(), 1, sl, 0, &I1, &stride), "level");
    hipMalloc(&flow[ii], (size_t)2 * ipl.width * ipl.height * 4);
    grid[ii]->InitializeGrid(I0, I0x, I0y);                                         // src/oflow.cpp:250
    grid[ii]->SetTargetImage(I1);                                                   // :251
    if (sl < opr.coarsest_scale) grid[ii]->InitializeFromCoarserOF(flow[ii + 1]);   // :266
    grid[ii]->Optimize();                                                           // :281
    grid[ii]->AggregateFlowDense(flow[ii]);                                         // :304
    if (opr.use_var_ref) OFC::VarRefClass var_ref(I0, I1, &ipl, &opr, flow[ii]);    // :332
  }
  std::vector<float> byhand((size_t)2 * ow * oh);
  hipMemcpy(byhand.data(), flow[0], byhand.size() * 4, hipMemcpyDeviceToHost);
  std::vector<OFC::dev_patch_state> st;
  grid[0]->GetPatchStates(st);                                                      // src/patch.h:15-36 on the host
  double moved = 0;
  for (auto &s : st) moved += (double)s.p_curx * s.p_curx + (double)s.p_cury * s.p_cury;
  const bool same = memcmp(byhand.data(), whole.data(), byhand.size() * 4) == 0;
  printf("scale loop by hand %s OFClass::calc; %zu patches at the finest scale, mean |p|^2 = %g\n", same ? "==" : "!=", st.size(), moved / st.size());
  FILE *f = fopen(argv[6], "wb");
  if (!f) return 1;
  fwrite(byhand.data(), 4, byhand.size(), f);
  fclose(f);
  for (auto g : grid) delete g;
  for (auto q : flow) hipFree(q);
  hipFree(d0); hipFree(d1);
  return same ? 0 : 3;
}
