// examples/multi_gpu.cpp -- a batch of frame pairs over every GPU of the node from ONE process (include/fotg/node.h over
// fotg_node_*).  The reference's driver owns one device (src/run_dense.cpp:277-289); pairs are independent, so the
// batch is cut into contiguous shards, one per GPU, with no exchange on the data path.
//
//   hipcc -O2 -Iinclude examples/multi_gpu.cpp -Lflowonthego_amd -lfotg -Wl,-rpath,$PWD/flowonthego_amd -o examples/multi_gpu
//   examples/multi_gpu frames0.raw frames1.raw W H N out_flows.raw [devices, e.g. 0,1,2,3 | all] [mode resident|scatter] [chunk] [repeat]
// frames0.raw / frames1.raw: N gray float32 frames each (pair k = frame k of both); out_flows.raw: the N finest-scale flows in pair
// order.  resident: every shard is uploaded to its own GPU first (what a server with per-GPU decoders has); scatter: everything
// is uploaded to the first GPU and the others pull their shards over xGMI in chunks under compute.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "fotg/node.h"

static std::vector<float> read_raw(const char *path, size_t n)
{
  std::vector<float> v(n);
  FILE *f = fopen(path, "rb");
  if (!f || fread(v.data(), sizeof(float), n, f) != n) { fprintf(stderr, "cannot read %zu floats from %s\n", n, path); exit(1); }
  fclose(f);
  return v;
}

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

int main(int argc, char **argv)
{
  if (argc < 7) { fprintf(stderr, "usage: %s frames0.raw frames1.raw W H N out_flows.raw [devices|all] [resident|scatter] [chunk] [repeat]\n", argv[0]); return 2; }
  const int W = atoi(argv[3]), H = atoi(argv[4]), N = atoi(argv[5]);
  std::vector<int> devs;
  int have = 0;
  HIP_OK(hipGetDeviceCount(&have));
  if (argc > 7 && strcmp(argv[7], "all")) { for (char *t = strtok(argv[7], ","); t; t = strtok(nullptr, ",")) devs.push_back(atoi(t)); }
  else for (int d = 0; d < have; ++d) devs.push_back(d);
  const bool scatter = argc > 8 && !strcmp(argv[8], "scatter");
  const int ndev = (int)devs.size(), per = (N + ndev - 1) / ndev;
  const int chunk = argc > 9 ? atoi(argv[9]) : (per < 16 ? per : 16), repeat = argc > 10 ? atoi(argv[10]) : 1;
  const size_t npx = (size_t)W * H;
  const std::vector<float> f0 = read_raw(argv[1], npx * N), f1 = read_raw(argv[2], npx * N);

  fotg_params p;
  OFC::fotgCheck(fotg_op_point(2, W, 1, &p), "fotg_op_point");
  OFC::opt_params op;
  op.coarsest_scale = p.sc_f; op.finest_scale = p.sc_l; op.patch_size = p.ps; op.patch_stride = p.patove;
  op.use_mean_normalization = p.patnorm != 0; op.grad_descent_iter = p.max_iter;
  op.dp_thresh = p.dp_thresh; op.dr_thresh = p.dr_thresh; op.res_thresh = p.res_thresh;
  op.use_var_ref = p.usetvref != 0; op.var_ref_iter = p.tv_solverit; op.var_ref_alpha = p.tv_alpha; op.var_ref_gamma = p.tv_gamma;
  op.var_ref_delta = p.tv_delta; op.var_ref_sor_weight = p.tv_sor; op.verbosity = 0; op.channels = 1;
  OFC::img_params iparams;
  iparams.width = W; iparams.height = H; iparams.padding = op.patch_size;

  OFC::FlowNode node(op, iparams, devs.data(), ndev, /*max_batch*/scatter ? chunk : per, /*depth*/scatter ? 3 : 2);
  const size_t nflow = (size_t)node.flow_channels * node.out_w * node.out_h;
  std::vector<float> flows(nflow * N);
  double best_ms = 1e30;

  if (!scatter) {
    std::vector<float *> I0(ndev, nullptr), I1(ndev, nullptr), out(ndev, nullptr);
    for (int d = 0; d < ndev; ++d) {
      int b, c;
      OFC::FlowNode::shard(N, ndev, d, &b, &c);
      if (!c) continue;
      HIP_OK(hipSetDevice(devs[d]));
      HIP_OK(hipMalloc(&I0[d], npx * c * 4)); HIP_OK(hipMalloc(&I1[d], npx * c * 4)); HIP_OK(hipMalloc(&out[d], nflow * c * 4));
      HIP_OK(hipMemcpy(I0[d], f0.data() + npx * b, npx * c * 4, hipMemcpyHostToDevice));
      HIP_OK(hipMemcpy(I1[d], f1.data() + npx * b, npx * c * 4, hipMemcpyHostToDevice));
    }
    for (int r = 0; r < repeat; ++r) {
      const auto t0 = std::chrono::steady_clock::now();
      node.wait(node.submit(N, I0.data(), I1.data(), out.data()));
      const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      if (ms < best_ms) best_ms = ms;
    }
    for (int d = 0; d < ndev; ++d) {
      int b, c;
      OFC::FlowNode::shard(N, ndev, d, &b, &c);
      if (!c) continue;
      HIP_OK(hipSetDevice(devs[d]));
      HIP_OK(hipMemcpy(flows.data() + nflow * b, out[d], nflow * c * 4, hipMemcpyDeviceToHost));
      hipFree(I0[d]); hipFree(I1[d]); hipFree(out[d]);
    }
  } else {
    float *G0 = nullptr, *G1 = nullptr, *GO = nullptr;
    HIP_OK(hipSetDevice(devs[0]));
    HIP_OK(hipMalloc(&G0, npx * N * 4)); HIP_OK(hipMalloc(&G1, npx * N * 4)); HIP_OK(hipMalloc(&GO, nflow * N * 4));
    HIP_OK(hipMemcpy(G0, f0.data(), npx * N * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(G1, f1.data(), npx * N * 4, hipMemcpyHostToDevice));
    for (int r = 0; r < repeat; ++r) {
      const auto t0 = std::chrono::steady_clock::now();
      node.wait(node.submit_scatter(N, G0, G1, GO, chunk));
      const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      if (ms < best_ms) best_ms = ms;
    }
    HIP_OK(hipMemcpy(flows.data(), GO, flows.size() * 4, hipMemcpyDeviceToHost));
    hipFree(G0); hipFree(G1); hipFree(GO);
  }
  FILE *f = fopen(argv[6], "wb");
  if (!f || fwrite(flows.data(), sizeof(float), flows.size(), f) != flows.size()) { fprintf(stderr, "cannot write %s\n", argv[6]); return 1; }
  fclose(f);
  printf("%s: %d flows of %dx%d over %d device slot(s), %s, best of %d: %.3f ms = %.0f pairs/s\n", argv[6], N, node.out_w, node.out_h, ndev,
         scatter ? "scatter from the first GPU" : "resident shards", repeat, best_ms, N / best_ms * 1e3);
  return 0;
}
