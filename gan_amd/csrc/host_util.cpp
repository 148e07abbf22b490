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
