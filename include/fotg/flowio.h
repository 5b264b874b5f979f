// include/fotg/flowio.h -- the reference's output writers, same names and byte layout:
//   SaveFlowFile  kroeger/run_dense.cpp:16-57 == src/run_dense.cpp:26-67   Middlebury .flo: "PIEH", int32 width, int32 height, then
//                 height x width x 2 float32 (u, v) row-major
//   SavePFMFile   kroeger/run_dense.cpp:60-81                              stereo depth build: "Pf\n<w> <h>\n-1.000000\n" (negative
//                 scale = little endian), rows BOTTOM-UP, every value NEGATED (the file holds the positive disparity)
// The reference takes cv::Mat; here the flow is the plain host array the engine's callers hold (what fotg_upsample_crop produced,
// copied to the host): h x w x 2 (flow) or h x w (disparity) float32.  Host-side only, nothing here is on the timed path.
// Returns false when the file cannot be written (the reference prints and exits; callers of the shim decide).
#ifndef FOTG_OFC_FLOWIO_HEADER
#define FOTG_OFC_FLOWIO_HEADER
#include <cstdio>
#include <cstddef>

namespace OFC {

inline bool SaveFlowFile(const float *uv, int width, int height, const char *filename)
{
  if (!uv || !filename || width <= 0 || height <= 0) return false;
  FILE *f = fopen(filename, "wb");
  if (!f) return false;
  bool ok = fwrite("PIEH", 1, 4, f) == 4 && fwrite(&width, sizeof(int), 1, f) == 1 && fwrite(&height, sizeof(int), 1, f) == 1;
  const size_t n = (size_t)2 * width * height;
  ok = ok && fwrite(uv, sizeof(float), n, f) == n;
  return (fclose(f) == 0) && ok;
}

inline bool SavePFMFile(const float *disp, int width, int height, const char *filename)
{
  if (!disp || !filename || width <= 0 || height <= 0) return false;
  FILE *f = fopen(filename, "wb");
  if (!f) return false;
  bool ok = fprintf(f, "Pf\n%d %d\n%f\n", width, height, -1.0f) > 0;
  for (int y = height - 1; y >= 0 && ok; --y)
    for (int x = 0; x < width; ++x) {
      const float v = -disp[(size_t)y * width + x];
      if (fwrite(&v, sizeof(float), 1, f) != 1) { ok = false; break; }
    }
  return (fclose(f) == 0) && ok;
}

}  // namespace OFC
#endif
