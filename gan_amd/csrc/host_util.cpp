// Host-side utility of the C ABI: CRC-32C (Castagnoli), the checksum of TensorFlow's TensorBundle / table files
// (gan_amd/tfbundle.py writes tf.train.Checkpoint-compatible checkpoints, pix2pix.py:400-420; a few hundred MB per
// save, so not a job for Python loops).  SSE4.2 crc32 instruction, 8 bytes per step.
#include <nmmintrin.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>
#include "../../include/gan_amd.h"

extern "C" uint32_t gan_crc32c(uint32_t crc, const void* data, size_t n) {
  const unsigned char* p = (const unsigned char*)data;
  uint64_t c = crc ^ 0xffffffffu;
  while (n && ((uintptr_t)p & 7)) { c = _mm_crc32_u8((uint32_t)c, *p++); --n; }
  while (n >= 8) { uint64_t v; memcpy(&v, p, 8); c = _mm_crc32_u64(c, v); p += 8; n -= 8; }
  while (n) { c = _mm_crc32_u8((uint32_t)c, *p++); --n; }
  return (uint32_t)c ^ 0xffffffffu;
}

// ---- planner options -------------------------------------------------------------------------------------------------
// The launch planners (conv_gemm.hip, wgrad.hip, thin.hip) take their tunable constants from this table, which only
// gan_set_option() changes: the library reads no environment variable.  Values are read at PLAN time (every entry point plans
// per call), so a change applies to the calls that follow it.  Keys and defaults: include/gan_amd.h.
#include <atomic>
namespace {
struct Opt { const char* key; std::atomic<int> value; };
Opt g_opts[] = {
    {"conv.big_tiles", {1}},        // 256-row tiles for layers with >= 128 output channels
    {"conv.q128", {55}},            // relative quality (percent) of the 128x128 tile in the tile choice
    {"conv.q256n", {80}},           //   ... of the 256x128 tile (256x256 = 100)
    {"conv.big_min_blocks", {64}},  // a 256-row tile needs at least this many blocks (128 until the table-driven kernels: +0.5 % on the step)
    {"conv.tall64", {1}},           // 256x64 tiles for 64-channel outputs on big maps
    {"conv.lean_epilogue", {1}},    // ping-pong launches without a fused backward epilogue use the instantiation compiled without it (fewer registers)
    {"conv.tap_share", {7}},        // tap-shared ping-pong kernel: bit 0 on the 256x128 tiles, bit 1 on the 256x256 tiles, bit 2 the table-driven form (256x128)
    {"conv.pingpong", {1}},         // 256-row tiles on the ping-pong kernel
    {"conv.parity_patch", {1}},     // parity-patch kernel for stride-2 transposed convs with 64 output channels
    {"conv.parity_patch_max_n", {64}},
    {"conv.parity_patch_min_blocks", {192}},
    {"conv.split_target", {256}},   // split K until this many blocks (128-row tiles and smaller; 512 before the slab-reduce kernels took over the normalisation)
    {"conv.split_target_skinny", {1024}},
    {"conv.split_target_big", {256}},
    {"conv.split_target_256", {128}},   // 256-row tiles with a short reduction: split K until this many blocks
    {"conv.split_min_ktiles", {4}},
    {"conv.split_max", {64}},
    {"conv.bwd_fuse_tile", {1}},    // fused backward epilogue on tile epilogues: 0 never, 1 always (fastest step: round 3 measured +0.5 % / +1 %), 2 not on 64-column tiles, 3 only on them
    {"conv.stack", {0}},            // runs of small split-K layers in one persistent launch (gan_conv_stack_*): OFF - measured 13-16 % slower than the launches it replaces
    {"conv.stack_blocks", {256}},   // ... its resident grid (multiple of 8; at most 2 workgroups per CU fit)
    {"conv.reduce_stats_rg", {16}}, // split-K slab reduce emitting statistics partials: at most this many row groups per workgroup (1: one chunk per group, as before round 5)
    {"conv.own_max_kb", {192}},     // ... and at most this many KB of operands per workgroup (live taps x (8 weight rows + the layer's rows))
    {"conv.own_max_rows", {16}},    // column-owner kernel (conv_own.hip) for GanNormFuse layers with at most this many rows per parity (<= 64; 0: never)
    {"conv.skn512_min_rows", {0}},  // finishing slab reduce (GanNormFuse) on 512 threads for statistics groups of at least this many rows (> 256; 0: never)
    {"conv.norm_fuse", {1}},        // GanNormFuse: small split-K layers finished by their slab-reduce kernel
    {"conv.thin_fused", {1}},       // thin-N layers with <= 2 output channels in one kernel (Z through LDS instead of memory)
    {"conv.thin_k_blocks", {2048}}, // thin-K streaming kernel: grid cap (workgroups over all 64-channel groups)
    {"conv.thin", {7}},             // bit 0: streaming kernels at all, bit 1: thin-N, bit 2: thin-K
    {"wgrad.tile256", {0}},         // 256-row tiles in the 128x128 kernel family
    {"wgrad.pingpong", {1}},
    {"wgrad.row_table", {1}},       // ping-pong wgrad: the block decodes its reduction rows once into an LDS table instead of per K tile in the loop
    {"wgrad.pingpong_min_rows", {0}},   // 0: 1024 rows per split (2048 when the launch shares the chip)
    {"wgrad.pingpong_128", {1}},       // 128-channel SMALL tensors on the 256-column ping-pong tile (half the columns dropped): +0.5 % on the step
    {"wgrad.pingpong_min_gflop", {30}},
    {"wgrad.split_target", {512}},
    {"wgrad.fold_split_target", {512}},    // tap-folded (8-channel) layers: blocks wanted (each writes a 32 KB slab tile; 1024 before round 4: 46 -> 40 us per step for the two launches, 256: 57)
    {"wgrad.reduce_adam_min_params", {1 << 20}},
    {"wgrad.reduce_adam", {1}},     // GanAdamFuse on split launches: the slab reduce ends in the optimiser step
    {"wgrad.adam_halves_max_rows", {0}},   // GanAdamFuse epilogue: launches with at most this many reduction rows put the two halves of a tile on two workgroups
    {"wgrad.dead_taps", {1}},       // GanAdamFuse epilogue: blocks of taps that never meet the map (2x2 -> 1x1 layers) skip the update where m == v == 0
    {"norm.fin_in_apply", {0}},     // descriptors with a sync area: the finalize of a normalisation layer inside its apply launch (one launch less per layer and direction). OFF: measured 2 % slower on the step (the in-launch publish / wait / re-read chain is three cold round trips, 8-9 us against 5 + a boundary)
    {"diag.launch_log", {0}},       // record the kernel symbol of every launch (gan_launch_log; profiling tools)
};
Opt* find_opt(const char* key) {
  if (!key) return nullptr;
  for (Opt& o : g_opts)
    if (!strcmp(o.key, key)) return &o;
  return nullptr;
}
}  // namespace

int gan_opt(const char* key) {              // internal: a key the table does not hold is a programming error
  Opt* o = find_opt(key);
  return o ? o->value.load(std::memory_order_relaxed) : 0;
}
// ---- diagnostic launch log (include/gan_amd.h: gan_launch_log) ---------------------------------------------------------
#include <mutex>
#include <vector>
namespace {
std::atomic<int> g_log_on{0};
std::mutex g_log_mu;
std::vector<const void*> g_log;
}  // namespace
void gan_launch_note(const void* kernel_host_fn) {       // every launch site (GAN_LAUNCH, common.h)
  if (!g_log_on.load(std::memory_order_relaxed)) return;
  std::lock_guard<std::mutex> lk(g_log_mu);
  g_log.push_back(kernel_host_fn);
}
size_t gan_launch_log_ptrs(const void** out, size_t cap) {     // elementwise.hip resolves the names (needs the HIP runtime)
  std::lock_guard<std::mutex> lk(g_log_mu);
  for (size_t i = 0; i < g_log.size() && i < cap; ++i) out[i] = g_log[i];
  return g_log.size();
}

extern "C" int gan_set_option(const char* key, int32_t value) {
  Opt* o = find_opt(key);
  if (!o) return GAN_E_ARG;
  o->value.store(value, std::memory_order_relaxed);
  if (!strcmp(key, "diag.launch_log")) {
    std::lock_guard<std::mutex> lk(g_log_mu);
    if (value) g_log.clear();
    g_log_on.store(value ? 1 : 0, std::memory_order_relaxed);
  }
  return 0;
}
extern "C" int gan_get_option(const char* key, int32_t* value) {
  Opt* o = find_opt(key);
  if (!o || !value) return GAN_E_ARG;
  *value = o->value.load(std::memory_order_relaxed);
  return 0;
}
