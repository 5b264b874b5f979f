// include/fotg/pipeline.h -- several batches in flight (fotg_pipe_* of include/fotg.h); no reference equivalent: the reference's
// OFClass::calc (src/oflow.cpp:211-368) is synchronous, one pair at a time.
//
//   OFC::FlowPipeline pipe(op, iparams, /*max_batch*/64, /*depth*/4);
//   long t = pipe.submit(n, I0, I1, nullptr, outflow_dev, upload_stream);   // returns at once; starts behind upload_stream's work
//   ...                                                                      // submit the next batches (other buffers)
//   pipe.wait(t, consumer_stream);                                           // consumer_stream waits on the device
//
// Same parameters and the same bits as OFClass::calc_batch.  Errors print and exit like checkCudaErrors.
#ifndef FOTG_PIPELINE_HEADER
#define FOTG_PIPELINE_HEADER
#include <algorithm>
#include <cmath>
#include "params.h"
#include "patchgrid.h"

namespace OFC {

class FlowPipeline {
 public:
  FlowPipeline(opt_params op, img_params i_params, int max_batch, int depth, int device = 0)
  {
    op.outlier_thresh = (float)op.patch_size / 2;                                         // src/oflow.cpp:45-48
    op.steps = std::max(1, (int)floor(op.patch_size * (1 - op.patch_stride)));
    op.n_vals = op.channels * op.patch_size * op.patch_size;
    op.n_scales = op.coarsest_scale - op.finest_scale + 1;
    fotg_params p = to_fotg(op);
    fotgCheck(fotg_pipe_create(&p, i_params.width, i_params.height, device, max_batch, depth, &pipe), "FlowPipeline");
  }
  ~FlowPipeline() { fotg_pipe_destroy(pipe); }
  FlowPipeline(const FlowPipeline &) = delete;
  FlowPipeline &operator=(const FlowPipeline &) = delete;

  // after_stream: the hipStream_t that produced the frames (nullptr = default stream), or FOTG_NO_STREAM: start at once
  long submit(int n, const float *I0, const float *I1, const float *initflow, float *outflow_dev, void *after_stream = nullptr)
  {
    long ticket = -1;
    fotgCheck(fotg_pipe_submit(pipe, n, I0, I1, initflow, outflow_dev, after_stream, &ticket), "FlowPipeline::submit");
    return ticket;
  }
  long submit_u8(int n, const unsigned char *I0, const unsigned char *I1, const float *initflow, float *outflow_dev, void *after_stream = nullptr)
  {
    long ticket = -1;
    fotgCheck(fotg_pipe_submit_u8(pipe, n, I0, I1, initflow, outflow_dev, after_stream, &ticket), "FlowPipeline::submit_u8");
    return ticket;
  }
  // frames / outflow in staging buffers that are recycled behind a device-side wait: a flagged stall of this batch is reported by
  // wait_host() / synchronize() instead of being recomputed from buffers that may be gone (fotg.h: RECOMPUTE CONTRACT)
  long submit_no_recompute(int n, const void *I0, const void *I1, bool u8, const float *initflow, float *outflow_dev, void *after_stream = nullptr)
  {
    long ticket = -1;
    fotgCheck(fotg_pipe_submit_ex(pipe, n, I0, I1, u8 ? 1 : 0, initflow, outflow_dev, after_stream, FOTG_SUBMIT_NO_RECOMPUTE, &ticket), "FlowPipeline::submit_no_recompute");
    return ticket;
  }
  // (after wait(ticket, stream) the waiting stream owns the result: the buffers of that batch may be freed or reused behind it, and the
  // pipe will not recompute into them)
  void wait(long ticket, void *stream = nullptr) { fotgCheck(fotg_pipe_wait(pipe, ticket, stream, 0), "FlowPipeline::wait"); }
  void wait_host(long ticket) { fotgCheck(fotg_pipe_wait(pipe, ticket, nullptr, 1), "FlowPipeline::wait_host"); }
  void synchronize() { fotgCheck(fotg_pipe_sync(pipe), "FlowPipeline::synchronize"); }
  fotg_pipe *handle() { return pipe; }

 private:
  fotg_pipe *pipe = nullptr;
};

}  // namespace OFC
#endif
