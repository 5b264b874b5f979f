// does an out-of-range lane of a buffer load to LDS write zeros or leave the LDS as it was?  (hipcc --offload-arch=gfx950 oob_probe.hip -o oob_probe)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const float4 *p, float4 *o, int nrec, int lo)
{
  __shared__ float4 lds[64];
  lds[threadIdx.x] = make_float4(-7.f, -7.f, -7.f, -7.f);
  __syncthreads();
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)p, 0, nrec, 0x00020000);
  typedef __attribute__((address_space(3))) void lvoid;
  const unsigned voff = (unsigned)((int)threadIdx.x - lo) * 16u;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lvoid *)lds, 16, voff, 0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  o[threadIdx.x] = lds[threadIdx.x];
}
int main()
{
  float4 h[64], r[64];
  for (int i = 0; i < 64; ++i) h[i] = make_float4(i + 1.f, i + 1.f, i + 1.f, i + 1.f);
  float4 *d, *o;
  hipMalloc(&d, sizeof(h)); hipMalloc(&o, sizeof(r));
  hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  k<<<1, 64>>>(d, o, 20 * 16 + 8, 3);            // lanes 0..2 below the range (huge offsets), lanes 3..22 inside, lane 23 half inside, lanes 24.. beyond
  hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
  for (int i = 0; i < 64; ++i) printf("%d:(%g %g %g %g) ", i, r[i].x, r[i].y, r[i].z, r[i].w);
  printf("\n");
  return 0;
}
