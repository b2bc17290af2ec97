// steps_s256.hpp -- launch arguments shared by the two S = 256 fused-step kernels
// (steps_s256.hip: split-bf16 three-product parity kernel; steps_s256_b16.hip: single-product bf16 kernel).
#pragma once
#include "common.hpp"

namespace ctdd {

constexpr int S256 = 256;
constexpr int S256_CHUNK_BYTES = 16384;     // one K-step (16 s0) of the A image: [plane 2][g 2][s 256][8 bf16]
constexpr size_t S256_INVQ16_OFFSET = (size_t)S256 * S256 * 4 /*invq f32*/ + 16 * (size_t)S256_CHUNK_BYTES /*A image*/;
constexpr size_t S256_STEP_TABLE_BYTES = S256_INVQ16_OFFSET + (size_t)S256 * S256 * 2 /*invq bf16 (steps_s256_b16.hip)*/;

struct S256Args {
  const float* logits;
  const int32_t* x;
  const int32_t* x_base;
  const unsigned char* tables;   // this step's derived tables (S256_STEP_TABLE_BYTES)
  const float* RT0;
  const float* R0;
  float beta, h;
  uint32_t flags;
  uint64_t seed, offset;
  int64_t R;                     // number of rows N*D
  float* out_rates;              // optional (R,256): masked reverse rates (validation / unfused use)
  int32_t* out_x;
  int32_t* out_changed;
};

// steps_s256_b16.hip: the CTDD_STEP_BF16 variant of the launch (same arguments, same tables)
int launch_tauleap_s256_b16(const S256Args& a, hipStream_t stream);

}  // namespace ctdd
