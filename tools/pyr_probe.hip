// pyr_probe.hip -- times the real pyr_base_kernel against the raw read pattern (tools only)
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../flowonthego_amd/csrc/pyramid.hip.h"
using namespace fotg;
int main()
{
  const int W = 1920, H = 1080, Hp = 1088, N = 64, ps = 8;
  const size_t img = (size_t)W * H;
  float *d0, *d1, *o0, *o1;
  hipMalloc(&d0, img * N * 4); hipMalloc(&d1, img * N * 4);
  hipMemset(d0, 0, img * N * 4); hipMemset(d1, 0, img * N * 4);
  const int tw = 120 + 16, th = 68 + 16; const long ls = (long)tw * th;
  hipMalloc(&o0, ls * N * 4); hipMalloc(&o1, ls * N * 4);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const int tiles = 8 * 68;
  auto run = [&](const char *name, auto launch) {
    launch(); hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < 10; ++i) launch();
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= 10;
    printf("%-40s %8.1f us  %7.2f TB/s\n", name, ms * 1e3, 2.0 * img * N * 4 / (ms * 1e-3) / 1e12);
  };
  run("pyr_base<1,4> fast both", [&] { pyr_base_kernel<float, 1, 4, true><<<dim3((tiles + 3) / 4, 2 * N), 256>>>(d0, d1, N, img, W, H, 0, 4, W, Hp, o0, o1, ls, tw, ps); });
  run("pyr_base<1,4> slow both", [&] { pyr_base_kernel<float, 1, 4, false><<<dim3((tiles + 3) / 4, 2 * N), 256>>>(d0, d1, N, img, W, H, 0, 4, W, Hp, o0, o1, ls, tw, ps); });
  run("pyr_base<1,3> fast both", [&] { pyr_base_kernel<float, 1, 3, true><<<dim3((8 * 136 + 3) / 4, 2 * N), 256>>>(d0, d1, N, img, W, H, 0, 4, W, Hp, o0, o1, ls, tw, ps); });
  run("pyr_base<1,2> fast both", [&] { pyr_base_kernel<float, 1, 2, true><<<dim3((8 * 272 + 3) / 4, 2 * N), 256>>>(d0, d1, N, img, W, H, 0, 4, W, Hp, o0, o1, ls, tw, ps); });
  return 0;
}
