// include/fotg/node.h -- one process, several GPUs (fotg_node_* of include/fotg.h).  No reference equivalent: the
// reference's driver creates one OFClass on one device (src/run_dense.cpp:277-289).
//
//   int devs[] = {0, 1, 2, 3, 4, 5, 6, 7};
//   OFC::FlowNode node(op, iparams, devs, 8, /*pairs per submission and device*/64, /*batches in flight per device*/4);
//   long t = node.submit(512, I0, I1, out);      // I0[d], I1[d], out[d]: shard d (OFC::FlowNode::shard) in device d's memory
//   node.wait(t);
//
// Same parameters and the same bits as OFClass::calc_batch.  Errors print and exit like checkCudaErrors.
#ifndef FOTG_NODE_HEADER
#define FOTG_NODE_HEADER
#include <algorithm>
#include <cmath>
#include "params.h"
#include "patchgrid.h"

namespace OFC {

class FlowNode {
 public:
  FlowNode(opt_params op, img_params i_params, const int *devices, int ndev, int max_batch, int depth)
  {
    op.outlier_thresh = (float)op.patch_size / 2;                                         // src/oflow.cpp:45-48
    op.steps = std::max(1, (int)floor(op.patch_size * (1 - op.patch_stride)));
    op.n_vals = op.channels * op.patch_size * op.patch_size;
    op.n_scales = op.coarsest_scale - op.finest_scale + 1;
    fotg_params p = to_fotg(op);
    fotgCheck(fotg_node_create(&p, i_params.width, i_params.height, devices, ndev, max_batch, depth, &node), "FlowNode");
    fotgCheck(fotg_node_info(node, &n_dev, &out_w, &out_h, &flow_channels), "FlowNode");
  }
  ~FlowNode() { fotg_node_destroy(node); }
  FlowNode(const FlowNode &) = delete;
  FlowNode &operator=(const FlowNode &) = delete;

  // [begin, begin + count) of the n pairs that slot d computes
  static void shard(int n, int ndev, int d, int *begin, int *count) { fotgCheck(fotg_node_shard(n, ndev, d, begin, count), "FlowNode::shard"); }
  // resident frames (per-slot device pointers); returns at once
  long submit(int n, const float *const *I0, const float *const *I1, float *const *outflow)
  {
    long t = -1;
    fotgCheck(fotg_node_submit(node, n, I0, I1, outflow, &t), "FlowNode::submit");
    return t;
  }
  long submit_u8(int n, const unsigned char *const *I0, const unsigned char *const *I1, float *const *outflow)
  {
    long t = -1;
    fotgCheck(fotg_node_submit_u8(node, n, I0, I1, outflow, &t), "FlowNode::submit_u8");
    return t;
  }
  // whole batch on devices[0]: the other slots pull their shards over xGMI in chunks while computing, flows return to outflow
  long submit_scatter(int n, const float *I0, const float *I1, float *outflow, int chunk)
  {
    long t = -1;
    fotgCheck(fotg_node_submit_scatter(node, n, I0, I1, outflow, chunk, &t), "FlowNode::submit_scatter");
    return t;
  }
  // the same with 8-bit frames: the shards travel as bytes (a quarter of the link traffic; three bytes per pixel with u8_color)
  long submit_scatter_u8(int n, const unsigned char *I0, const unsigned char *I1, float *outflow, int chunk)
  {
    long t = -1;
    fotgCheck(fotg_node_submit_scatter_u8(node, n, I0, I1, outflow, chunk, &t), "FlowNode::submit_scatter_u8");
    return t;
  }
  void wait(long ticket) { fotgCheck(fotg_node_wait(node, ticket), "FlowNode::wait"); }
  void synchronize() { fotgCheck(fotg_node_sync(node), "FlowNode::synchronize"); }
  fotg_node *handle() { return node; }
  int n_dev = 0, out_w = 0, out_h = 0, flow_channels = 2;

 private:
  fotg_node *node = nullptr;
};

}  // namespace OFC
#endif
