// bw_probe.hip -- read-bandwidth ceilings on this GPU for the access patterns of pyr_base_kernel (tools only).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float vf4 __attribute__((ext_vector_type(4)));
__device__ inline float4 ntload(const float4 *q) { vf4 t = __builtin_nontemporal_load(reinterpret_cast<const vf4 *>(q)); return make_float4(t.x, t.y, t.z, t.w); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// A: linear grid-stride float4 read, U loads in flight per lane
template <int U, bool NT>
__global__ __launch_bounds__(256) void k_linear(const float4 *__restrict__ p, size_t n4, float *out)
{
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
  float acc = 0.f;
  for (; i + (U - 1) * stride < n4; i += U * stride) {
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = NT ? ntload(&p[i + u * stride]) : p[i + u * stride];
#pragma unroll
    for (int u = 0; u < U; ++u) acc += v[u].x + v[u].y + v[u].z + v[u].w;
  }
  if (acc == 123.456f) out[0] = acc;
}

// B: the pyramid tile pattern: wave = 256 px x R rows of a W-wide image (row pitch W floats), 4 waves per block
template <int R, bool NT>
__global__ __launch_bounds__(256) void k_tile(const float *__restrict__ p, int W, int H, size_t img_stride, float *out)
{
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int strips = (W + 255) >> 8, tile = blockIdx.x * 4 + wave, oh = H / R;
  if (tile >= strips * oh) return;
  const int strip = tile % strips, oy = tile / strips, x0 = strip * 256 + lane * 4;
  if (x0 >= W) return;
  const float *src = p + (size_t)blockIdx.y * img_stride;
  float4 v[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const float4 *q = reinterpret_cast<const float4 *>(src + (size_t)(oy * R + r) * W + x0);
    v[r] = NT ? ntload(q) : *q;
  }
  float acc = 0.f;
#pragma unroll
  for (int r = 0; r < R; ++r) acc += v[r].x + v[r].y + v[r].z + v[r].w;
  if (acc == 123.456f) out[0] = acc;
}

// C: band pattern: one block streams a band of R full rows (contiguous R*W floats), lanes interleaved
template <int R, int U>
__global__ __launch_bounds__(256) void k_band(const float *__restrict__ p, int W, int H, size_t img_stride, float *out)
{
  const float4 *src = reinterpret_cast<const float4 *>(p + (size_t)blockIdx.y * img_stride + (size_t)blockIdx.x * R * W);
  const int n4 = R * W / 4;
  float acc = 0.f;
  for (int i = threadIdx.x; i < n4; i += 256 * U) {
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { int k = i + u * 256; v[u] = k < n4 ? src[k] : make_float4(0, 0, 0, 0); }
#pragma unroll
    for (int u = 0; u < U; ++u) acc += v[u].x + v[u].y + v[u].z + v[u].w;
  }
  if (acc == 123.456f) out[0] = acc;
}

int main()
{
  const int W = 1920, H = 1088, N = 128;
  const size_t img = (size_t)W * H, tot = img * N;      // 1.07 GB
  float *d, *out;
  CK(hipMalloc(&d, tot * 4)); CK(hipMalloc(&out, 4));
  CK(hipMemset(d, 0, tot * 4));
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  auto run = [&](const char *name, auto launch) {
    launch(); hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < 10; ++i) launch();
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= 10;
    printf("%-40s %8.1f us  %7.2f TB/s\n", name, ms * 1e3, tot * 4 / (ms * 1e-3) / 1e12);
  };
  run("linear U=4 grid=256*8", [&] { k_linear<4, false><<<256 * 8, 256>>>((const float4 *)d, tot / 4, out); });
  run("linear U=8 grid=256*8", [&] { k_linear<8, false><<<256 * 8, 256>>>((const float4 *)d, tot / 4, out); });
  run("linear U=8 grid=256*16", [&] { k_linear<8, false><<<256 * 16, 256>>>((const float4 *)d, tot / 4, out); });
  run("linear U=16 grid=256*8", [&] { k_linear<16, false><<<256 * 8, 256>>>((const float4 *)d, tot / 4, out); });
  run("linear U=8 NT grid=256*8", [&] { k_linear<8, true><<<256 * 8, 256>>>((const float4 *)d, tot / 4, out); });
  run("tile R=16", [&] { k_tile<16, false><<<dim3((8 * (H / 16) + 3) / 4, N), 256>>>(d, W, H, img, out); });
  run("tile R=16 NT", [&] { k_tile<16, true><<<dim3((8 * (H / 16) + 3) / 4, N), 256>>>(d, W, H, img, out); });
  run("tile R=8", [&] { k_tile<8, false><<<dim3((8 * (H / 8) + 3) / 4, N), 256>>>(d, W, H, img, out); });
  run("tile R=4", [&] { k_tile<4, false><<<dim3((8 * (H / 4) + 3) / 4, N), 256>>>(d, W, H, img, out); });
  run("band R=16 U=8", [&] { k_band<16, 8><<<dim3(H / 16, N), 256>>>(d, W, H, img, out); });
  run("band R=16 U=16", [&] { k_band<16, 16><<<dim3(H / 16, N), 256>>>(d, W, H, img, out); });
  run("band R=4 U=8", [&] { k_band<4, 8><<<dim3(H / 4, N), 256>>>(d, W, H, img, out); });
  return 0;
}
