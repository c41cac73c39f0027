// Arguments of the MFMA flash attention kernel (flash.hip) and its launcher; shared with attn.hip (srgpt_attention).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "common.h"

struct AttnArgs {
  const bf16_t *q, *k, *v;
  bf16_t* o;
  int Tq, Tk, Hq, Hkv, D;
  int64_t q_bs, q_ts, q_hs, k_bs, k_ts, k_hs, v_bs, v_ts, v_hs;
  float scale;
  const int* kv_len;
};

int64_t srgpt_flash_slice_span_limit();  // byte offsets inside a (batch, head) K / V slice are 32-bit in the kernel
void srgpt_flash_bf16_launch(const AttnArgs& a, int B, bool causal, hipStream_t s);
