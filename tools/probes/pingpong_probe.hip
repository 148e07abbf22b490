// What the skeleton of an 8-wave "ping-pong" loop costs on MI355X (gfx950), without any real work around it: a 512-thread workgroup per
// CU runs PHASES phases of  [LOAD segment | s_barrier | MATH segment | s_barrier], waves 4..7 one barrier behind waves 0..3.
//   mode 0  s_barrier only (2 per phase)
//   mode 1  + s_waitcnt lgkmcnt(0), s_setprio 1 / 0 around an empty MATH segment
//   mode 2  + 16 MFMAs (16x16x32 bf16) in the MATH segment
//   mode 3  + 12 ds_read_b128 in the LOAD segment (conflict-free rows)
//   mode 4  mode 2 without the stagger (both waves of a SIMD multiply in the same interval)
//   mode 5  mode 3 with 32 MFMAs and 12 reads per phase (a 128x64 wave tile per phase)
//   mode 6  one barrier per phase: [reads for the NEXT phase issued in front of the MFMAs of this one | s_barrier], no stagger
//   dma<NP, ROWS>: mode 3 + NP LDS-DMA pieces (1 KiB each) per wave in the LOAD segment, 2*NP left in flight across the barrier
//           (counted vmcnt), from a 32 MB buffer that stays in the L2s / Infinity Cache; ROWS = 1: a piece is 1 KiB contiguous,
//           ROWS = 8: 8 rows of 128 B, 4 KB apart (the im2col gather of a 2048-channel... of a pixel pitch of 4 KB)
// hipcc --offload-arch=gfx950 -O3 tools/probes/pingpong_probe.hip -o gan_amd/probes_bin/pingpong_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

template <int MODE>
__global__ __launch_bounds__(512) void probe(int phases, float* out, unsigned long long* cyc) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2;
  const int r = lane & 15, q = lane >> 4;
  for (int i = tid; i < 16384; i += 512) ((unsigned*)smem)[i] = 0x3f803f80u;
  __syncthreads();
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
  const unsigned addr = lds_base + (wave & 3) * 16384 + r * 128 + ((q ^ (r & 7)) << 4);
  f32x4 acc[8][2];
  for (int i = 0; i < 8; ++i) for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
  uint4 fr[12];
  for (int i = 0; i < 12; ++i) fr[i] = uint4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
  constexpr bool stagger = MODE != 4 && MODE != 6;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  __builtin_amdgcn_s_barrier();
  if (stagger && wr == 1) __builtin_amdgcn_s_barrier();
  for (int p = 0; p < phases; ++p) {
    if constexpr (MODE == 3 || MODE == 5 || MODE == 6) {
#pragma unroll
      for (int i = 0; i < 12; ++i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fr[i]) : "v"(addr), "n"((i & 7) * 2048 + (i >> 3) * 64));
    }
    if constexpr (MODE != 6) __builtin_amdgcn_s_barrier();
    if constexpr (MODE >= 1) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
    }
    if constexpr (MODE >= 2) {
      constexpr int NM = MODE == 5 || MODE == 6 ? 32 : 16;
#pragma unroll
      for (int k = 0; k < NM; ++k)
        acc[k & 7][(k >> 3) & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)&fr[k % 4], *(const bf16x8*)&fr[4 + k % 8], acc[k & 7][(k >> 3) & 1], 0, 0, 0);
    }
    if constexpr (MODE >= 1) __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_s_barrier();
  }
  if (stagger && wr == 0) __builtin_amdgcn_s_barrier();
  __syncthreads();
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < 8; ++i) for (int j = 0; j < 2; ++j) s += acc[i][j][0] + acc[i][j][3];
  if (s == 12345.f) out[tid] = s;
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NP, int ROWS, int NM>
__global__ __launch_bounds__(512) void probe_dma(int phases, float* out, unsigned long long* cyc, const unsigned char* src, unsigned srcbytes) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2;
  const int r = lane & 15, q = lane >> 4;
  for (int i = tid; i < 16384; i += 512) ((unsigned*)smem)[i] = 0x3f803f80u;
  __syncthreads();
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
  const unsigned addr = lds_base + (wave & 3) * 16384 + r * 128 + ((q ^ (r & 7)) << 4);
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, srcbytes, 0x00020000);
  const int lrow = lane >> 3, slot = lane & 7;
  const unsigned lane_off = ROWS == 8 ? (unsigned)(lrow * 4096 + slot * 16) : (unsigned)(lane * 16);
  f32x4 acc[8][2];
  for (int i = 0; i < 8; ++i) for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
  uint4 fr[12];
  for (int i = 0; i < 12; ++i) fr[i] = uint4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
  unsigned pos = (unsigned)(blockIdx.x * 8 + wave) * 65536u;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  __builtin_amdgcn_s_barrier();
  if (wr == 1) __builtin_amdgcn_s_barrier();
  for (int p = 0; p < phases; ++p) {
#pragma unroll
    for (int k = 0; k < NP; ++k) {
      pos = (pos + 32768u * 37u) & (srcbytes - 1) & ~32767u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(smem + 65536 + ((p * NP + k) % 8) * 8192 + wave * 1024), 16,
                                               pos + lane_off, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 12; ++i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fr[i]) : "v"(addr), "n"((i & 7) * 2048 + (i >> 3) * 64));
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NP) : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int k = 0; k < NM; ++k)
      acc[k & 7][(k >> 3) & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)&fr[k % 4], *(const bf16x8*)&fr[4 + k % 8], acc[k & 7][(k >> 3) & 1], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_s_barrier();
  }
  if (wr == 0) __builtin_amdgcn_s_barrier();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < 8; ++i) for (int j = 0; j < 2; ++j) s += acc[i][j][0] + acc[i][j][3];
  if (s == 12345.f) out[tid] = s;
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NP, int ROWS, int NM>
static void run_dma(int phases, float* out, unsigned long long* cyc, const unsigned char* src, unsigned srcbytes) {
  constexpr int smem = 65536 + 8 * 8192;
  hipFuncSetAttribute((const void*)probe_dma<NP, ROWS, NM>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  probe_dma<NP, ROWS, NM><<<256, 512, smem>>>(phases, out, cyc, src, srcbytes);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  probe_dma<NP, ROWS, NM><<<256, 512, smem>>>(phases, out, cyc, src, srcbytes);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  printf("dma<%d pieces/wave/phase, %d rows/piece, %d MFMAs>: %.1f ns per phase = %.0f cycles at 2.4 GHz; %.1f GB/s per CU, %.2f TB/s chip; MFMA pipe %.0f %%\n", NP, ROWS, NM,
         ms * 1e6 / phases, ms * 1e6 / phases * 2.4, 8.0 * NP * 1024 / (ms * 1e6 / phases), 256 * 8.0 * NP * 1024 / (ms * 1e6 / phases) / 1e3,
         2.0 * NM * 16 / (ms * 1e6 / phases * 2.4) * 100);
}

template <int MODE>
static void run(int phases, float* out, unsigned long long* cyc) {
  hipFuncSetAttribute((const void*)probe<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  probe<MODE><<<256, 512, 65536>>>(phases, out, cyc);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  probe<MODE><<<256, 512, 65536>>>(phases, out, cyc);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[256];
  hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double m = 0;
  for (int i = 0; i < 256; ++i) m += (double)h[i];
  m /= 256;
  printf("mode %d: %d phases  kernel %.1f us  %.1f ns per phase  %.0f shader cycles per phase (mean over blocks)\n", MODE, phases, ms * 1e3,
         ms * 1e6 / phases, m / phases);
}

int main(int argc, char** argv) {
  const int phases = argc > 1 ? atoi(argv[1]) : 2000;
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 4096); hipMalloc(&cyc, 256 * 8);
  run<0>(phases, out, cyc); run<1>(phases, out, cyc); run<2>(phases, out, cyc); run<3>(phases, out, cyc);
  run<4>(phases, out, cyc); run<5>(phases, out, cyc); run<6>(phases, out, cyc);
  unsigned char* src; const unsigned srcbytes = 32u << 20;
  hipMalloc(&src, srcbytes); hipMemset(src, 0, srcbytes);
  run_dma<1, 1, 16>(phases, out, cyc, src, srcbytes); run_dma<2, 1, 16>(phases, out, cyc, src, srcbytes); run_dma<3, 1, 16>(phases, out, cyc, src, srcbytes);
  run_dma<4, 1, 16>(phases, out, cyc, src, srcbytes);
  run_dma<1, 8, 16>(phases, out, cyc, src, srcbytes); run_dma<2, 8, 16>(phases, out, cyc, src, srcbytes); run_dma<3, 8, 16>(phases, out, cyc, src, srcbytes);
  run_dma<4, 8, 16>(phases, out, cyc, src, srcbytes);
  run_dma<3, 8, 32>(phases, out, cyc, src, srcbytes); run_dma<6, 8, 32>(phases, out, cyc, src, srcbytes); run_dma<4, 8, 32>(phases, out, cyc, src, srcbytes);
  return 0;
}
