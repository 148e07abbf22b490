/* C caller of the ABI in include/gan_amd.h (no Python, no torch): Conv2D(k4, s2, 'same') + LeakyReLU(0.3) of
 * GAN.downsample (base_gan.py:77-79,87) on a small fp32 tensor, checked against a scalar loop in this file.
 * Built by `make -C gan_amd/csrc smoke` (hipcc compiles it as C++ only for the HIP runtime calls); run by
 * tests/test_gpu_api.py::test_c_caller_of_the_abi.  Exit 0 = ok. */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../include/gan_amd.h"

#define N 2
#define H 8
#define CI 8
#define CO 16
#define HO (H / 2)

int main(void) {
  static float x[N][H][H][CI], w[4][4][CI][CO], y[N][HO][HO][CO], ref[N][HO][HO][CO];
  srand(7);
  for (size_t i = 0; i < sizeof x / 4; ++i) ((float*)x)[i] = (float)(rand() % 201 - 100) / 100.f;
  for (size_t i = 0; i < sizeof w / 4; ++i) ((float*)w)[i] = (float)(rand() % 201 - 100) / 500.f;
  for (int n = 0; n < N; ++n) for (int oy = 0; oy < HO; ++oy) for (int ox = 0; ox < HO; ++ox) for (int co = 0; co < CO; ++co) {
    double s = 0;
    for (int kh = 0; kh < 4; ++kh) for (int kw = 0; kw < 4; ++kw) {
      const int iy = 2 * oy + kh - 1, ix = 2 * ox + kw - 1;            /* TF 'same', even size: pad 1 on every side */
      if (iy < 0 || iy >= H || ix < 0 || ix >= H) continue;
      for (int ci = 0; ci < CI; ++ci) s += (double)x[n][iy][ix][ci] * w[kh][kw][ci][co];
    }
    ref[n][oy][ox][co] = (float)(s > 0 ? s : 0.3 * s);
  }
  float *dx, *dw, *dtr, *dy; void* ws;
  const size_t ws_bytes = 1 << 20;
  if (hipMalloc((void**)&dx, sizeof x) || hipMalloc((void**)&dw, sizeof w) || hipMalloc((void**)&dtr, sizeof w) ||
      hipMalloc((void**)&dy, sizeof y) || hipMalloc(&ws, ws_bytes)) { fprintf(stderr, "hipMalloc failed\n"); return 2; }
  hipMemcpy(dx, x, sizeof x, hipMemcpyHostToDevice);
  hipMemcpy(dw, w, sizeof w, hipMemcpyHostToDevice);
  hipMemset(dy, 0, sizeof y);
  int rc = gan_weights_prepare(dw, CI, CO, GAN_F32, NULL, dtr, NULL);     /* HWIO master -> [tap][cout][cin] */
  if (rc) { fprintf(stderr, "gan_weights_prepare rc=%d\n", rc); return 3; }
  GanConvDesc d;
  memset(&d, 0, sizeof d);
  d.struct_size = sizeof d; d.dtype = GAN_F32; d.stride = 2;
  d.x.ptr = dx; d.x.n = N; d.x.h = H; d.x.w = H; d.x.c = CI; d.x.pitch = CI;
  d.y.ptr = dy; d.y.n = N; d.y.h = HO; d.y.w = HO; d.y.c = CO; d.y.pitch = CO;
  d.w = dtr; d.w_rows = CO; d.act = GAN_ACT_LRELU; d.slope = 0.3f; d.workspace = ws; d.workspace_bytes = ws_bytes;
  GanConvDesc bad = d;
  bad.struct_size -= 8;
  if (gan_conv2d_fwd(&bad, NULL) != GAN_E_ARG) { fprintf(stderr, "a short descriptor was accepted\n"); return 4; }
  rc = gan_conv2d_fwd(&d, NULL);
  if (rc) { fprintf(stderr, "gan_conv2d_fwd rc=%d\n", rc); return 5; }
  if (hipDeviceSynchronize() != hipSuccess) return 6;
  hipMemcpy(y, dy, sizeof y, hipMemcpyDeviceToHost);
  double err = 0, mx = 0;
  for (size_t i = 0; i < sizeof y / 4; ++i) {
    const double e = fabs((double)((float*)y)[i] - ((float*)ref)[i]);
    if (e > err) err = e;
    if (fabs(((float*)ref)[i]) > mx) mx = fabs(((float*)ref)[i]);
  }
  printf("c_abi_smoke: %s, max-abs err %.3e (max |ref| %.3f)\n", gan_version(), err, mx);
  return err <= 2e-5 * mx ? 0 : 1;
}
