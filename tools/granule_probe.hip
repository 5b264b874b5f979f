// does a workgroup that polls a 16-byte cell with sc1 loads see another workgroup's sc1 store to it -- also when it has read
// (and cached) the line BEFORE the store?   hipcc --offload-arch=gfx950 -O2 tools/granule_probe.hip -o /tmp/granule_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned v4u __attribute__((ext_vector_type(4)));
__global__ void k(v4u *cell, int *out, int delay_us, int mode)
{
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)cell, 0, 64 * 16 * 64, 0x00020000);
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  if (blockIdx.x == 0) {                       // producer: wait, then store (sc1, like the tile writer)
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < (long long)delay_us * 100) __builtin_amdgcn_s_sleep(8);
    __builtin_amdgcn_raw_buffer_store_b128(v4u{1u + threadIdx.x, 2u, 3u, 4u}, rs, threadIdx.x * 16u, 0, 16);
    if (threadIdx.x == 0) out[0] = (int)xcc;
  } else if (blockIdx.x == gridDim.x - 1) {    // consumer: poll
    int spins = 0;
    v4u v;
    const long long t0 = wall_clock64();
    do {
      if (mode == 0) v = __builtin_amdgcn_raw_buffer_load_b128(rs, threadIdx.x * 16u, 0, 16);
      else { const v4u *p = cell + threadIdx.x; asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory"); }
      if (v.x != 0xffffffffu) break;
      __builtin_amdgcn_s_sleep(4);
    } while (++spins < (1 << 16));
    if (threadIdx.x == 0) { out[1] = (int)xcc; out[2] = spins; out[3] = (int)v.x; out[4] = (int)((wall_clock64() - t0) / 100); }
  }
}
int main()
{
  v4u *cell; int *out; int h[8];
  (void)hipMalloc(&cell, 64 * 16 * 64); (void)hipMalloc(&out, 32);
  for (int mode = 0; mode < 2; ++mode)
    for (int rep = 0; rep < 6; ++rep) {
      (void)hipMemset(cell, 0xff, 64 * 16 * 64); (void)hipMemset(out, 0, 32);
      k<<<2 + rep, 64>>>(cell, out, 20, mode);      // first and last workgroup: different XCDs for most grid sizes
      (void)hipDeviceSynchronize();
      (void)hipMemcpy(h, out, 32, hipMemcpyDeviceToHost);
      printf("mode %d grid %d: producer on XCD %d, consumer on XCD %d: %d spins, saw %d after %d us\n", mode, 2 + rep, h[0], h[1], h[2], h[3], h[4]);
    }
  return 0;
}
