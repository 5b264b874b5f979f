#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(float *o) {
  float B = (float)threadIdx.x;
  float lo = B, hi = B;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(lo), "+v"(hi));
  o[threadIdx.x] = lo;
  o[64 + threadIdx.x] = hi;
}
int main() {
  float *d, h[128]; hipMalloc(&d, 512);
  k<<<1, 64>>>(d); hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
  printf("sw[0]: lane0=%g lane31=%g lane32=%g lane63=%g\n", h[0], h[31], h[32], h[63]);
  printf("sw[1]: lane0=%g lane31=%g lane32=%g lane63=%g\n", h[64], h[95], h[96], h[127]);
  return 0;
}
