// Cost and correctness of an in-kernel grid barrier + cross-workgroup data exchange on MI355X (gfx950): a persistent grid runs NB
// phases; in every phase a block writes 16 KB that ANOTHER block (another XCD) reads back and checks in the next phase.
//   barrier flavour  0: one counter, atomics only           1: + agent-scope release/acquire fences by one wave per block
//                    2: __threadfence() by every thread     3: hierarchical (8 group counters + 8 release flags), atomics only
//   data path        plain: ordinary loads / stores (write-back L2, not coherent across XCDs without the fences)
//                    coh:   global_load / global_store with sc0 sc1 (write-through / L2-bypassing), no cache maintenance at all
// hipcc --offload-arch=gfx950 -O3 tools/probes/gridbar_probe.hip -o gan_amd/probes_bin/gridbar_probe
#include <hip/hip_runtime.h>
#include <cstdio>

typedef __attribute__((ext_vector_type(4))) float f32x4;

// (the data registers of an asm store must stay untouched until the store has read them: the compiler re-used them for the next
// vector right behind the asm statement and 3 of 16 checks failed even with full fences - hence the wait inside the statement)
__device__ __forceinline__ void st_coh(f32x4* p, f32x4 v) { asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ f32x4 ld_coh(const f32x4* p) {
  f32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}

struct Bar { unsigned long long ctr; unsigned long long pad0[15]; unsigned long long grp[8][16]; unsigned long long rel[8][16]; unsigned long long master; };

__device__ __forceinline__ bool spin_until(const unsigned long long* p, unsigned long long target, int* err) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
    __builtin_amdgcn_s_sleep(1);
    if (__builtin_amdgcn_s_memrealtime() - t0 > 200000000ull) { *err = 1; return false; }   // 2 s at 100 MHz: give up, never hang
  }
  return true;
}

__device__ __forceinline__ bool grid_barrier(Bar* b, unsigned nblocks, int flavour, unsigned long long phase, int* err) {
  if (flavour == 2) __threadfence();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's stores have been acknowledged
  __syncthreads();
  bool ok = true;
  if (threadIdx.x == 0) {
    if (flavour == 1) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    if (flavour != 3) {
      const unsigned long long old = __hip_atomic_fetch_add(&b->ctr, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      ok = spin_until(&b->ctr, (old / nblocks + 1) * nblocks, err);
    } else {
      const unsigned g = blockIdx.x & 7, per = nblocks >> 3;               // nblocks % 8 == 0
      const unsigned long long old = __hip_atomic_fetch_add(&b->grp[g][0], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (old % per == per - 1) {                                         // last of the group: tell the master
        const unsigned long long m = __hip_atomic_fetch_add(&b->master, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (m % 8 == 7)                                                   // last group: release everyone (one flag per group)
          for (int k = 0; k < 8; ++k) __hip_atomic_store(&b->rel[k][0], phase + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      ok = spin_until(&b->rel[g][0], phase + 1, err);
    }
    if (flavour == 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
  if (flavour == 2) __threadfence();
  return ok;
}

template <bool COH>
__global__ __launch_bounds__(256) void bar_kernel(Bar* bar, int nb, int flavour, int* err, int* bad, f32x4* data) {
  const unsigned G = gridDim.x;
  int nbad = 0;
  for (int i = 0; i < nb; ++i) {
    // phase i: write my 16 KB slot (4 x 4 KB), tagged (block, phase)
    for (int k = 0; k < 4; ++k) {
      f32x4* dst = data + ((size_t)blockIdx.x * 4 + k) * 256 + threadIdx.x;
      const f32x4 v = f32x4{(float)blockIdx.x, (float)i, (float)k, (float)threadIdx.x};
      if (COH) st_coh(dst, v); else *dst = v;
    }
    if (!grid_barrier(bar, G, flavour, (unsigned long long)(2 * i), err)) break;
    // read the slot of a block 3 XCDs away (round-robin dispatch: block % 8 = XCD) and check the tag
    const unsigned src = (blockIdx.x + 3 + 8 * (i % 5)) % G;
    for (int k = 0; k < 4; ++k) {
      const f32x4* p = data + ((size_t)src * 4 + k) * 256 + threadIdx.x;
      const f32x4 v = COH ? ld_coh(p) : *p;
      if (v[0] != (float)src || v[1] != (float)i || v[2] != (float)k) ++nbad;
    }
    if (!grid_barrier(bar, G, flavour, (unsigned long long)(2 * i + 1), err)) break;     // WAR: everyone has read before the next phase overwrites
  }
  if (nbad) atomicAdd(bad, nbad);
}

int main() {
  Bar* bar; int* err; int* bad; f32x4* data;
  (void)hipMalloc(&bar, sizeof(Bar)); (void)hipMalloc(&err, 4); (void)hipMalloc(&bad, 4); (void)hipMalloc(&data, (size_t)512 * 4 * 256 * 16);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int nb = 1000;
  for (int grid : {128, 256, 512}) {
    for (int coh = 0; coh < 2; ++coh)
      for (int fl = 0; fl < 4; ++fl) {
        (void)hipMemset(bar, 0, sizeof(Bar)); (void)hipMemset(err, 0, 4); (void)hipMemset(bad, 0, 4); (void)hipMemset(data, 0, (size_t)512 * 4 * 256 * 16);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        if (coh) hipLaunchKernelGGL(bar_kernel<true>, dim3(grid), dim3(256), 0, 0, bar, nb, fl, err, bad, data);
        else hipLaunchKernelGGL(bar_kernel<false>, dim3(grid), dim3(256), 0, 0, bar, nb, fl, err, bad, data);
        (void)hipEventRecord(e1);
        (void)hipDeviceSynchronize();
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        int herr = 0, hbad = 0;
        (void)hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost); (void)hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost);
        printf("grid %4d data %-5s barrier flavour %d: %6.3f us per barrier (+ 16 KB write / read per phase), stale reads %d, timeout %d\n", grid,
               coh ? "coh" : "plain", fl, ms * 1e3 / (2 * nb), hbad, herr);
      }
  }
  return 0;
}
