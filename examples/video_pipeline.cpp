// examples/video_pipeline.cpp -- a video loop with several frame pairs in flight (include/fotg/pipeline.h over fotg_pipe_*).
// The reference's run_dense handles one pair per process and its OFClass::calc is synchronous (src/oflow.cpp:211-368); a
// caller with a stream of frames keeps several pairs in flight instead.
//
//   hipcc -O2 -Iinclude examples/video_pipeline.cpp -Lflowonthego_amd -lfotg -Wl,-rpath,$PWD/flowonthego_amd -o examples/video_pipeline
//   examples/video_pipeline frames.raw W H N out_flows.raw [depth] [op-point]
// frames.raw: N consecutive gray float32 frames (W*H each); out_flows.raw: the N-1 finest-scale flows (u,v interleaved), pair k =
// (frame k, frame k+1), in order.  Every pair is submitted as it "arrives"; up to `depth` of them overlap on the GPU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "fotg/pipeline.h"

int main(int argc, char **argv)
{
  if (argc < 6) { fprintf(stderr, "usage: %s frames.raw W H N out_flows.raw [depth] [op-point]\n", argv[0]); return 2; }
  const int W = atoi(argv[2]), H = atoi(argv[3]), N = atoi(argv[4]), depth = argc > 6 ? atoi(argv[6]) : 3, oppt = argc > 7 ? atoi(argv[7]) : 2;
  const size_t npx = (size_t)W * H;
  std::vector<float> frames(npx * N);
  FILE *f = fopen(argv[1], "rb");
  if (!f || fread(frames.data(), sizeof(float), frames.size(), f) != frames.size()) { fprintf(stderr, "cannot read %d frames from %s\n", N, argv[1]); return 1; }
  fclose(f);

  fotg_params p;
  OFC::fotgCheck(fotg_op_point(oppt, W, 1, &p), "fotg_op_point");
  OFC::opt_params op;
  op.coarsest_scale = p.sc_f; op.finest_scale = p.sc_l; op.patch_size = p.ps; op.patch_stride = p.patove;
  op.use_mean_normalization = p.patnorm != 0; op.grad_descent_iter = p.max_iter;
  op.dp_thresh = p.dp_thresh; op.dr_thresh = p.dr_thresh; op.res_thresh = p.res_thresh;
  op.use_var_ref = p.usetvref != 0; op.var_ref_iter = p.tv_solverit; op.var_ref_alpha = p.tv_alpha; op.var_ref_gamma = p.tv_gamma;
  op.var_ref_delta = p.tv_delta; op.var_ref_sor_weight = p.tv_sor; op.verbosity = 0; op.channels = 1;
  OFC::img_params iparams;
  iparams.width = W; iparams.height = H; iparams.padding = op.patch_size;

  OFC::FlowPipeline pipe(op, iparams, /*max_batch*/1, depth);
  fotg_ctx *ctx0 = nullptr;
  OFC::fotgCheck(fotg_pipe_context(pipe.handle(), 0, &ctx0), "fotg_pipe_context");
  int ow, oh;
  OFC::fotgCheck(fotg_out_size(ctx0, &ow, &oh), "fotg_out_size");
  const size_t nflow = (size_t)2 * ow * oh;

  // all frames on the device (a real caller uploads frame k+1 on `upload` while pair k-1 is being computed)
  float *dframes = nullptr, *dflows = nullptr;
  hipStream_t upload;
  if (hipMalloc(&dframes, frames.size() * 4) != hipSuccess || hipMalloc(&dflows, nflow * (N - 1) * 4) != hipSuccess ||
      hipStreamCreateWithFlags(&upload, hipStreamNonBlocking) != hipSuccess) { fprintf(stderr, "hip allocation failed\n"); return 1; }
  std::vector<long> tickets;
  for (int k = 0; k < N; ++k) {
    hipMemcpyAsync(dframes + npx * k, frames.data() + npx * k, npx * 4, hipMemcpyHostToDevice, upload);
    if (k > 0)       // pair (k-1, k): starts behind the upload of frame k, overlaps with the pairs before it
      tickets.push_back(pipe.submit(1, dframes + npx * (k - 1), dframes + npx * k, nullptr, dflows + nflow * (k - 1), upload));
  }
  pipe.synchronize();
  std::vector<float> flows(nflow * (N - 1));
  hipMemcpy(flows.data(), dflows, flows.size() * 4, hipMemcpyDeviceToHost);
  f = fopen(argv[5], "wb");
  if (!f || fwrite(flows.data(), sizeof(float), flows.size(), f) != flows.size()) { fprintf(stderr, "cannot write %s\n", argv[5]); return 1; }
  fclose(f);
  printf("%s: %d flows of %dx%d written, %d pairs in flight, last ticket %ld\n", argv[5], N - 1, ow, oh, depth, tickets.back());
  hipFree(dframes); hipFree(dflows); hipStreamDestroy(upload);
  return 0;
}
