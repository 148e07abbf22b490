// Store-shape probe (MI355X, gfx950): what does a streaming kernel lose when one wave-instruction writes sixteen 64-byte half lines
// (conv_thin_k_kernel's epilogue: lane (pixel n, q) stores 16 B at pixel*128 + q*16, a second instruction the other half) instead of
// eight full 128-byte lines?  Writes a 64-MiB buffer R times in each shape, optionally beside a 32-MiB read stream of 16 B per lane.
// hipcc --offload-arch=gfx950 -O3 tools/probes/store_shape_probe.hip -o gan_amd/probes_bin/store_shape_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { if ((x) != hipSuccess) { printf("HIP error at %s\n", #x); return 1; } } while (0)
#define CKV(x) do { if ((x) != hipSuccess) printf("HIP error at %s\n", #x); } while (0)

template <int SHAPE, bool READ>
__global__ __launch_bounds__(256) void store_kernel(uint4* __restrict__ dst, const uint4* __restrict__ src, long long pixels) {
  const int lane = threadIdx.x & 63;
  const long long wave = ((long long)blockIdx.x * 256 + threadIdx.x) >> 6, nwaves = ((long long)gridDim.x * 256) >> 6;
  for (long long t = wave; t * 16 < pixels; t += nwaves) {          // a wave's tile = 16 pixels x 128 B (2 KB)
    uint4 v = make_uint4(lane, (unsigned)t, 3u, 4u);
    if (READ) { const uint4 r = src[t * 64 + lane]; v.x ^= r.x; v.y ^= r.w; }
    uint4* base = dst + t * 128;                                       // 128 x 16 B
    if (SHAPE == 0) {                   // half lines: pixel n = lane & 15, q = lane >> 4: 16 B at pixel * 8 + q (+4 for the second instruction)
      const int n = lane & 15, q = lane >> 4;
      base[n * 8 + q] = v;
      base[n * 8 + 4 + q] = v;
    } else {                            // full lines: 64 lanes x 16 B contiguous, twice
      base[lane] = v;
      base[64 + lane] = v;
    }
  }
}

int main() {
  const long long pixels = 524288;       // x 128 B = 64 MiB (D.down0's output at batch 32)
  uint4 *dst, *src;
  CK(hipMalloc(&dst, pixels * 128)); CK(hipMalloc(&src, pixels * 64));
  CK(hipMemset(src, 1, pixels * 64));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto run = [&](auto kern, const char* name) {
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(kern, dim3(2048), dim3(256), 0, 0, dst, src, pixels);
    CKV(hipEventRecord(e0));
    const int R = 20;
    for (int r = 0; r < R; ++r) hipLaunchKernelGGL(kern, dim3(2048), dim3(256), 0, 0, dst, src, pixels);
    CKV(hipEventRecord(e1)); CKV(hipEventSynchronize(e1));
    float ms = 0.f; CKV(hipEventElapsedTime(&ms, e0, e1));
    printf("%-34s %7.1f us per launch  %.2f TB/s written\n", name, ms / R * 1e3, pixels * 128.0 / (ms / R * 1e-3) / 1e12);
  };
  run(store_kernel<0, false>, "half lines (16 x 64 B), no read");
  run(store_kernel<1, false>, "full lines (8 x 128 B), no read");
  run(store_kernel<0, true>, "half lines, + 32 MiB read");
  run(store_kernel<1, true>, "full lines, + 32 MiB read");
  return 0;
}
