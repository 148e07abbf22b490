// What normalise-on-read would cost inside the 8-wave ping-pong loop (review item J2 / 7): the skeleton of tools/probes/pingpong_probe.hip
// (512-thread workgroup per CU, [LOAD | s_barrier | MATH | s_barrier], waves 4..7 one barrier behind; 3 LDS-DMA pieces + 12 ds_read_b128
// per wave and 16-MFMA phase = the table-driven 256x128 kernel's segments) with, per phase, the affine + LeakyReLU transform
// a = max(z, 0.3 z), z = y * s[c] + t[c] applied to NXF of the wave's 8 A fragments (8 bf16 each: unpack, 8 FMA, 8 mul, 8 max, pack -
// gfx950 has no packed bf16 arithmetic) in front of the MFMAs.  A 128x32 wave tile reads 8 A fragments per phase, and each of the 4
// waves of a tile row would transform the same rows again.
// hipcc --offload-arch=gfx950 -O3 tools/probes/norm_on_read_probe.hip -o gan_amd/probes_bin/norm_on_read_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { if ((x) != hipSuccess) { printf("HIP error at %s\n", #x); exit(1); } } while (0)

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
  __bf16 a = (__bf16)lo, b = (__bf16)hi;
  return (uint32_t)(*(uint16_t*)&a) | ((uint32_t)(*(uint16_t*)&b) << 16);
}
__device__ __forceinline__ uint4 xform(uint4 v, const float* s, const float* t) {
  float x[8];
  x[0] = __uint_as_float(v.x << 16); x[1] = __uint_as_float(v.x & 0xffff0000u);
  x[2] = __uint_as_float(v.y << 16); x[3] = __uint_as_float(v.y & 0xffff0000u);
  x[4] = __uint_as_float(v.z << 16); x[5] = __uint_as_float(v.z & 0xffff0000u);
  x[6] = __uint_as_float(v.w << 16); x[7] = __uint_as_float(v.w & 0xffff0000u);
#pragma unroll
  for (int e = 0; e < 8; ++e) { const float z = fmaf(x[e], s[e], t[e]); x[e] = fmaxf(z, 0.3f * z); }
  return make_uint4(pack_bf2(x[0], x[1]), pack_bf2(x[2], x[3]), pack_bf2(x[4], x[5]), pack_bf2(x[6], x[7]));
}

template <int NP, int NM, int NXF>
__global__ __launch_bounds__(512) void probe(int phases, float* out, const unsigned char* src, unsigned srcbytes, const float* coef) {
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
  const unsigned lane_off = (unsigned)(lrow * 4096 + slot * 16);
  float s[2][8], t[2][8];               // the lane's 2 x 8 channels of a K tile (both k steps): loop-invariant registers
  for (int k = 0; k < 2; ++k) for (int e = 0; e < 8; ++e) { s[k][e] = coef[(k * 4 + q) * 8 + e]; t[k][e] = coef[64 + (k * 4 + q) * 8 + e]; }
  f32x4 acc[8][2];
  for (int i = 0; i < 8; ++i) for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
  uint4 fr[12];
  for (int i = 0; i < 12; ++i) fr[i] = uint4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
  unsigned pos = (unsigned)(blockIdx.x * 8 + wave) * 65536u;
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
#pragma unroll
    for (int i = 0; i < NXF; ++i) fr[4 + i] = xform(fr[4 + i], s[i & 1], t[i & 1]);      // the A fragments of the phase
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
  float sum = 0;
  for (int i = 0; i < 8; ++i) for (int j = 0; j < 2; ++j) sum += acc[i][j][0] + acc[i][j][3];
  if (sum == 12345.f) out[tid] = sum;
}

template <int NP, int NM, int NXF>
static void run(int phases, float* out, const unsigned char* src, unsigned srcbytes, const float* coef) {
  constexpr int smem = 65536 + 8 * 8192;
  CK(hipFuncSetAttribute((const void*)probe<NP, NM, NXF>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  probe<NP, NM, NXF><<<256, 512, smem>>>(phases, out, src, srcbytes, coef);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  probe<NP, NM, NXF><<<256, 512, smem>>>(phases, out, src, srcbytes, coef);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  printf("%d pieces, %d MFMAs, %d of 8 A fragments transformed per wave and phase: %.1f ns per phase; MFMA pipe %.0f %% (at 2.4 GHz)\n", NP, NM, NXF,
         ms * 1e6 / phases, 2.0 * NM * 16 / (ms * 1e6 / phases * 2.4) * 100);
}

int main(int argc, char** argv) {
  const int phases = argc > 1 ? atoi(argv[1]) : 2000;
  float *out, *coef;
  CK(hipMalloc(&out, 4096)); CK(hipMalloc(&coef, 128 * 4));
  float h[128];
  for (int i = 0; i < 128; ++i) h[i] = i < 64 ? 1.0f + 0.01f * i : 0.001f * i;
  CK(hipMemcpy(coef, h, sizeof(h), hipMemcpyHostToDevice));
  unsigned char* src; const unsigned srcbytes = 32u << 20;
  CK(hipMalloc(&src, srcbytes)); CK(hipMemset(src, 0, srcbytes));
  run<3, 16, 0>(phases, out, src, srcbytes, coef);
  run<3, 16, 2>(phases, out, src, srcbytes, coef);
  run<3, 16, 4>(phases, out, src, srcbytes, coef);
  run<3, 16, 8>(phases, out, src, srcbytes, coef);
  return 0;
}
