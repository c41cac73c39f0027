// Attention kernels.
//  1. flash_bf16_kernel  -- prefill / ViT: MFMA flash attention, in flash.hip (its own translation unit: it is compiled with
//     MFMA accumulators in arch VGPRs, this file is not); launched from srgpt_attention below.
//  2. simple_attn_kernel -- one wave per (query, head); any dtype / head_dim; fp32 parity path.
//  3. decode_split/combine -- single new token against the static KV cache, flash-decoding split over
//     keys, fused RoPE of the new q/k and cache append.  HBM/latency bound.
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "flash.h"

namespace {

// ================================================================================================
// 2. simple attention: one wave per (query row, head)
// ================================================================================================
template <typename T>
__global__ __launch_bounds__(64) void simple_attn_kernel(const T* __restrict__ q, const T* __restrict__ k,
                                                         const T* __restrict__ v, T* __restrict__ o, int Tq, int Tk,
                                                         int Hq, int Hkv, int D, int64_t q_bs, int64_t q_ts,
                                                         int64_t q_hs, int64_t k_bs, int64_t k_ts, int64_t k_hs,
                                                         int64_t v_bs, int64_t v_ts, int64_t v_hs, float scale,
                                                         int causal, const int* __restrict__ kv_len) {
  __shared__ float qs[256];
  const int t = blockIdx.x, h = blockIdx.y, b = blockIdx.z, lane = threadIdx.x;
  const int hk = h / (Hq / Hkv);
  const int klen = kv_len ? min(kv_len[b], Tk) : Tk;
  const T* qp = q + b * q_bs + (int64_t)t * q_ts + h * q_hs;
  for (int d = lane; d < D; d += 64) qs[d] = to_f(qp[d]);
  __syncthreads();
  const T* kb = k + b * k_bs + hk * k_hs;
  const T* vb = v + b * v_bs + hk * v_hs;
  const int last = causal ? min(klen - 1, t + (Tk - Tq)) : klen - 1;
  float m = -INFINITY, l = 0.f;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};  // d = lane + 64 * i, D <= 256
  for (int k0 = 0; k0 <= last; k0 += 64) {
    const int key = k0 + lane;
    float s = -INFINITY;
    if (key <= last) {
      const T* kr = kb + (int64_t)key * k_ts;
      float dot = 0.f;
      for (int d = 0; d < D; ++d) dot = fmaf(qs[d], to_f(kr[d]), dot);
      s = dot * scale;
    }
    const float m_new = fmaxf(m, wave_max(s));
    const float alpha = __expf(m - m_new);
    const float p = (key <= last) ? __expf(s - m_new) : 0.f;
    l = l * alpha + wave_sum(p);
    m = m_new;
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] *= alpha;
    const int nk = min(64, last + 1 - k0);
    for (int j = 0; j < nk; ++j) {
      const float pj = __shfl(p, j);
      const T* vr = vb + (int64_t)(k0 + j) * v_ts;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int d = lane + 64 * i;
        if (d < D) acc[i] = fmaf(pj, to_f(vr[d]), acc[i]);
      }
    }
  }
  const float inv = l > 0.f ? 1.f / l : 0.f;
  T* op = o + (((int64_t)b * Tq + t) * Hq + h) * D;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int d = lane + 64 * i;
    if (d < D) op[d] = from_f<T>(acc[i] * inv);
  }
}

// ================================================================================================
// 3. decode attention: split over keys + combine
// ================================================================================================
constexpr int DEC_CHUNK_MAX = 256;
constexpr int DEC_SPLIT_MAX = 64;

#ifdef SRGPT_TUNING_KNOBS
// phase stamps of the decode attention kernel (tuning build only; scripts/experiments/ubench_decode_stamps.py): block 0 -> slots 0..15, the
// block that merges (kv head 0, sequence 0) -> slots 16..31
__device__ unsigned long long srgpt_dbg_stamps[32];
#define DEC_STAMP(i) do { if (stamp_base >= 0 && threadIdx.x == 0) srgpt_dbg_stamps[stamp_base + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define DEC_STAMP(i) do { } while (0)
#endif
typedef SrgptPrefetch DecodePrefetch;  // common.h: L2 prefetch blocks appended to the launch (here: o_proj's weights)

// splits per (sequence, kv head): enough blocks for ~2 per CU, capped at 16 for a single sequence (it needs them to spread its
// K/V rows over the chip) and at 8 from two sequences up (the batch already spreads; more splits only multiply the merge work --
// decode step at 4 sequences 3.44 ms with 8 splits vs 3.47 with 16, at 8 sequences 3.60 / 3.66, profiles/r02_decode_splits.txt),
// never fewer than the score buffer requires (DEC_CHUNK_MAX keys per split)
static inline int decode_nsplit(int max_pos, int B, int Hkv) {
  const int force = SRGPT_KNOB("SRGPT_DECODE_MIN_SPLITS", 0);  // tuning build: fixed split count
  int want = cdiv(2 * srgpt_device_cus(), Hkv * B);
  const int cap = B == 1 ? 16 : 8;
  if (want > cap) want = cap;
  if (want < 1) want = 1;
  if (force > 0) want = force;
  int n = cdiv(max_pos, DEC_CHUNK_MAX);
  if (n < want) n = want;
  if (n > DEC_SPLIT_MAX) n = DEC_SPLIT_MAX;
  return n;
}

// ---- the tail both decode kernels share: arrival ticket of the (sequence, kv head) group, merge by the last arriver ----
// the MFMA kernel (bf16, head_dim 128): a block owns a FIXED range of 64 keys (4 waves x 16), so ceil(capacity / 64) blocks per
// (sequence, kv head) -- those past the sequence end leave at once and the merge skips them; beyond 64 splits the ranges grow in steps of 64 keys
// and the waves loop
static inline int decode_nsplit_mfma(int max_pos) {
  const int force = SRGPT_KNOB("SRGPT_DECODE_MIN_SPLITS", 0);  // tuning build: fixed split count
  int n = cdiv(max_pos, 64);
  if (n < 1) n = 1;
  if (force > 0) n = force;
  if (n > DEC_SPLIT_MAX) n = DEC_SPLIT_MAX;
  return n;
}
static inline bool decode_use_mfma(int dtype_is_bf16, int D, int G) {
  return dtype_is_bf16 && D == 128 && (G == 1 || G == 2 || G == 4 || G == 8) && SRGPT_KNOB("SRGPT_DECODE_MFMA", 1);
}

template <typename T, int D, int G, int SCLD>
__device__ __forceinline__ void decode_ticket_merge(float* __restrict__ wbase, int* __restrict__ ticket, int nsplit,
                                                    T* __restrict__ outp, float* __restrict__ sc, float* __restrict__ stat_m,
                                                    float* __restrict__ stat_l, int tid, int lane, int wave, bool first_group,
                                                    int stamp_base_in) {
#ifdef SRGPT_TUNING_KNOBS
  int stamp_base = stamp_base_in;
#endif
  // ---- arrival ticket: every storing wave drains its write-through stores, one lane takes the ticket; the block that draws
  //      the last one merges the nsplit partials of its G query heads (fixed split order: the result does not depend on
  //      which block came last) and re-arms the ticket for the next launch on this stream ----
  DEC_STAMP(7);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  DEC_STAMP(8);
  if (tid == 0) {
    const int t = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    stat_l[0] = (t == nsplit - 1) ? 1.f : 0.f;  // stat_l is free again: broadcast "I am last" through the existing LDS
  }
  __syncthreads();
  DEC_STAMP(9);
  if (stat_l[0] == 0.f) return;
#ifdef SRGPT_TUNING_KNOBS
  if (first_group) stamp_base = 16;
#endif
  DEC_STAMP(0);
  __syncthreads();
  // The first 16 splits' partials of this thread's pair of output dims AND the per-split statistics are requested back to
  // back: the merge pays one memory latency.  (G * D / 2 <= 256 * MAXW items; one or two per thread for the shipped shapes.)
  constexpr int PRE = 16;
  constexpr int MAXW = (G * (D / 2) + 255) / 256;
  unsigned long long v[MAXW][PRE];
#pragma unroll
  for (int wi = 0; wi < MAXW; ++wi) {
    const int w = min(tid + 256 * wi, G * (D / 2) - 1);
    const int gq = w / (D / 2), d = 2 * (w - gq * (D / 2));
    const float* wp = wbase + (size_t)gq * DEC_SPLIT_MAX * (D + 2) + d;
#pragma unroll
    for (int j = 0; j < PRE; ++j)
      v[wi][j] = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(wp + (size_t)min(j, nsplit - 1) * (D + 2)),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  // per-head merge weights: wave gq handles head gq (G <= 8 heads over 4 waves), one split per lane (nsplit <= 64)
  for (int gq = wave; gq < G; gq += 4) {
    const float* wp = wbase + (size_t)gq * DEC_SPLIT_MAX * (D + 2);
    const bool ok = lane < nsplit;
    const int sidx = ok ? lane : nsplit - 1;
    const unsigned long long ml = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(wp + (size_t)sidx * (D + 2) + D),
                                                    __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const float ms_raw = __uint_as_float((unsigned)ml), ls_raw = __uint_as_float((unsigned)(ml >> 32));
    const float ms = ok ? ms_raw : -INFINITY;
    const float ls = ok ? ls_raw : 0.f;
    const float M = lanes_max<64>(ms);
    const float wv = (ms > -INFINITY) ? __expf(ms - M) : 0.f;
    const float den = lanes_sum<64>(wv * ls);
    sc[gq * SCLD + lane] = wv;
    if (lane == 0) stat_m[gq] = den > 0.f ? 1.f / den : 0.f;
  }
  DEC_STAMP(1);
  __syncthreads();
#pragma unroll
  for (int wi = 0; wi < MAXW; ++wi) {
    const int w = tid + 256 * wi;
    if (w < G * (D / 2)) {
      const int gq = w / (D / 2), d = 2 * (w - gq * (D / 2));
      const float* wp = wbase + (size_t)gq * DEC_SPLIT_MAX * (D + 2) + d;
      float n0 = 0.f, n1 = 0.f;
#pragma unroll
      for (int j = 0; j < PRE; ++j)
        if (j < nsplit) {
          n0 = fmaf(sc[gq * SCLD + j], __uint_as_float((unsigned)v[wi][j]), n0);
          n1 = fmaf(sc[gq * SCLD + j], __uint_as_float((unsigned)(v[wi][j] >> 32)), n1);
        }
      // long contexts (> 16 live splits = > 1024 keys on the MFMA kernel): the remaining partials in batches of 16 loads in flight,
      // consumed in split order -- the same fma chain as one load at a time (48 dependent L2 round trips at 4096 keys before)
      for (int s0 = PRE; s0 < nsplit; s0 += PRE) {
        unsigned long long u[PRE];
#pragma unroll
        for (int j = 0; j < PRE; ++j)
          u[j] = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(wp + (size_t)min(s0 + j, nsplit - 1) * (D + 2)),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int j = 0; j < PRE; ++j)
          if (s0 + j < nsplit) {
            n0 = fmaf(sc[gq * SCLD + s0 + j], __uint_as_float((unsigned)u[j]), n0);
            n1 = fmaf(sc[gq * SCLD + s0 + j], __uint_as_float((unsigned)(u[j] >> 32)), n1);
          }
      }
      T* op = outp + (size_t)gq * D + d;
      op[0] = from_f<T>(n0 * stat_m[gq]);
      op[1] = from_f<T>(n1 * stat_m[gq]);
    }
  }
  DEC_STAMP(2);
  if (tid == 0) __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <typename T, int D, int G>
__global__ __launch_bounds__(256) void decode_split_kernel(const T* __restrict__ qkv, T* __restrict__ kcache,
                                                           T* __restrict__ vcache, const int* __restrict__ pos,
                                                           const T* __restrict__ cos_tab, const T* __restrict__ sin_tab,
                                                           float* __restrict__ ws, int* __restrict__ tickets,
                                                           T* __restrict__ out, int Hq, int Hkv, int max_pos,
                                                           int nsplit, float scale, int n_attn, DecodePrefetch pf) {
  constexpr int VEC = Vec16<T>::N;
  constexpr int LPK = D / VEC;   // lanes per key
  constexpr int KPW = 64 / LPK;  // keys per wave-instruction
  constexpr int HALF = D / 2;
  constexpr int STRIDE = 4 * KPW;  // keys per block iteration
  constexpr int PF = 2;            // key iterations whose K/V rows are fetched up front (3 and 4 measured: no difference)
  __shared__ float qs[G][D];
  __shared__ float knew[D], vnew[D];
  __shared__ float sc[G][DEC_CHUNK_MAX];
  __shared__ float red[4][G][D];
  __shared__ float stat_m[G], stat_l[G];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if ((int)blockIdx.x >= n_attn) {  // ---- prefetch block (common.h) ----
    srgpt_prefetch_block(pf, (int)blockIdx.x - n_attn, wave, lane);
    return;
  }
  const int hk = (int)blockIdx.x % Hkv, split = ((int)blockIdx.x / Hkv) % nsplit, b = (int)blockIdx.x / (Hkv * nsplit);
#ifdef SRGPT_TUNING_KNOBS
  int stamp_base = blockIdx.x == 0 ? 0 : -1;
#endif
  DEC_STAMP(0);
  const int P = pos[b];
  const int total = P + 1;
  // the key ranges depend on the sequence length only (NOT on the cache capacity: fixed capacity-based ranges were measured --
  // no faster -- and make the summation order, hence the bits, depend on how large a cache the caller happened to allocate)
  const int chunk = (total + nsplit - 1) / nsplit;
  const int kbeg = split * chunk, kend = min(kbeg + chunk, total);
  const T* row = qkv + (size_t)b * (Hq + 2 * Hkv) * D;
  T* kc = kcache + ((size_t)b * Hkv + hk) * (size_t)max_pos * D;
  T* vc = vcache + ((size_t)b * Hkv + hk) * (size_t)max_pos * D;
  const int sub = lane / LPK, dl = (lane % LPK) * VEC;
  const int key0 = kbeg + wave * KPW + sub;

  // ---- all cache rows of the first PF iterations go out before anything depends on q: one memory latency ----
  Vec16<T> kreg[PF], vreg[PF];
#pragma unroll
  for (int j = 0; j < PF; ++j) {
    const int key = key0 + j * STRIDE;
    if (key < kend && key != P) {
      kreg[j] = *reinterpret_cast<const Vec16<T>*>(kc + (size_t)key * D + dl);
      vreg[j] = *reinterpret_cast<const Vec16<T>*>(vc + (size_t)key * D + dl);
    }
  }

  DEC_STAMP(1);
  // ---- rotate q (G heads) and the new k; stage v.  Every global load of this stage is issued before any of
  //      them is consumed (unrolled, index wrapped instead of branched): one L2 latency instead of three. ----
  {
    constexpr int NITEM = (G + 1) * HALF;
    constexpr int ITEMS = (NITEM + 255) / 256;
    float x1v[ITEMS], x2v[ITEMS], cv[ITEMS], sv[ITEMS];
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
      const int w = (tid + 256 * it) % NITEM;
      const int hh = w / HALF, i = w - hh * HALF;
      const T* p = (hh < G) ? row + (size_t)(hk * G + hh) * D : row + (size_t)(Hq + hk) * D;
      x1v[it] = to_f(p[i]);
      x2v[it] = to_f(p[i + HALF]);
      cv[it] = to_f(cos_tab[(size_t)P * HALF + i]);
      sv[it] = to_f(sin_tab[(size_t)P * HALF + i]);
    }
    const float vraw = to_f(row[(size_t)(Hq + Hkv + hk) * D + (tid % D)]);
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
      const int w0 = tid + 256 * it;
      if (w0 < NITEM) {
        const int hh = w0 / HALF, i = w0 - hh * HALF;
        const float o1 = rnd<T>(rnd<T>(x1v[it] * cv[it]) + rnd<T>(-x2v[it] * sv[it]));
        const float o2 = rnd<T>(rnd<T>(x2v[it] * cv[it]) + rnd<T>(x1v[it] * sv[it]));
        if (hh < G) {
          qs[hh][i] = o1;
          qs[hh][i + HALF] = o2;
        } else {
          knew[i] = o1;
          knew[i + HALF] = o2;
        }
      }
    }
    if (tid < D) vnew[tid] = vraw;
  }
  DEC_STAMP(2);
  __syncthreads();
  DEC_STAMP(3);
  if (split == 0) {  // exactly one block per (b, hk) appends; nobody reads position P from the cache
    for (int d = tid; d < D; d += 256) {
      kc[(size_t)P * D + d] = from_f<T>(knew[d]);
      vc[(size_t)P * D + d] = from_f<T>(vnew[d]);
    }
  }

  float* wbase = ws + (((size_t)b * Hkv + hk) * G) * (size_t)DEC_SPLIT_MAX * (D + 2);
  // Partials are published WRITE-THROUGH (agent-scope relaxed atomic stores = global_store ... sc1) and read back with
  // agent-scope relaxed loads by whichever block of this (sequence, kv head) arrives last: no second launch for the merge.
  auto publish = [&](float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
  auto publish2 = [&](float* p, float a, float b2) {
    const unsigned long long u = (unsigned long long)__float_as_uint(a) | ((unsigned long long)__float_as_uint(b2) << 32);
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  const bool empty = kbeg >= kend;
  if (empty) {  // empty split: neutral partial (zeros, m = -inf, l = 0)
    for (int w = tid; w < G * (D + 2); w += 256) {
      const int gq = w / (D + 2), d = w - gq * (D + 2);
      publish(wbase + ((size_t)gq * DEC_SPLIT_MAX + split) * (D + 2) + d, (d == D) ? -INFINITY : 0.f);
    }
  }
  if (!empty) {

  // ---- pass 1: scores ----
  auto score = [&](int key, const float* kv) {
#pragma unroll
    for (int gq = 0; gq < G; ++gq) {
      float part = 0.f;
#pragma unroll
      for (int i = 0; i < VEC; ++i) part = fmaf(kv[i], qs[gq][dl + i], part);
      part = lanes_sum<LPK>(part);  // DPP / permlane: a ds_bpermute chain here cost ~2k cycles per key batch
      if ((lane % LPK) == 0) sc[gq][key - kbeg] = part * scale;
    }
  };
#pragma unroll
  for (int j = 0; j < PF; ++j) {
    const int key = key0 + j * STRIDE;
    if (key < kend) {
      float kv[VEC];
#pragma unroll
      for (int i = 0; i < VEC; ++i) kv[i] = (key == P) ? knew[dl + i] : kreg[j].get(i);
      score(key, kv);
    }
  }
  for (int key = key0 + PF * STRIDE; key < kend; key += STRIDE) {
    float kv[VEC];
    if (key == P) {
#pragma unroll
      for (int i = 0; i < VEC; ++i) kv[i] = knew[dl + i];
    } else {
      const Vec16<T> t = *reinterpret_cast<const Vec16<T>*>(kc + (size_t)key * D + dl);
#pragma unroll
      for (int i = 0; i < VEC; ++i) kv[i] = t.get(i);
    }
    score(key, kv);
  }
  DEC_STAMP(4);
  __syncthreads();
  // ---- softmax statistics of the chunk, one wave per q head ----
  const int n = kend - kbeg;
  for (int gq = wave; gq < G; gq += 4) {
    float mx = -INFINITY;
    for (int i = lane; i < n; i += 64) mx = fmaxf(mx, sc[gq][i]);
    mx = lanes_max<64>(mx);
    float sum = 0.f;
    for (int i = lane; i < n; i += 64) {
      const float pv = __expf(sc[gq][i] - mx);
      sc[gq][i] = pv;
      sum += pv;
    }
    sum = lanes_sum<64>(sum);
    if (lane == 0) {
      stat_m[gq] = mx;
      stat_l[gq] = sum;
    }
  }
  DEC_STAMP(5);
  __syncthreads();
  // ---- pass 2: O = P V ----
  float acc[G][VEC];
#pragma unroll
  for (int gq = 0; gq < G; ++gq)
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc[gq][i] = 0.f;
  auto accum = [&](int key, const float* vv) {
#pragma unroll
    for (int gq = 0; gq < G; ++gq) {
      const float pg = sc[gq][key - kbeg];
#pragma unroll
      for (int i = 0; i < VEC; ++i) acc[gq][i] = fmaf(pg, vv[i], acc[gq][i]);
    }
  };
#pragma unroll
  for (int j = 0; j < PF; ++j) {
    const int key = key0 + j * STRIDE;
    if (key < kend) {
      float vv[VEC];
#pragma unroll
      for (int i = 0; i < VEC; ++i) vv[i] = (key == P) ? vnew[dl + i] : vreg[j].get(i);
      accum(key, vv);
    }
  }
  for (int key = key0 + PF * STRIDE; key < kend; key += STRIDE) {
    float vv[VEC];
    if (key == P) {
#pragma unroll
      for (int i = 0; i < VEC; ++i) vv[i] = vnew[dl + i];
    } else {
      const Vec16<T> t = *reinterpret_cast<const Vec16<T>*>(vc + (size_t)key * D + dl);
#pragma unroll
      for (int i = 0; i < VEC; ++i) vv[i] = t.get(i);
    }
    accum(key, vv);
  }
#pragma unroll
  for (int gq = 0; gq < G; ++gq)
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      const float x = strided_sum<LPK>(acc[gq][i]);  // over the wave's KPW key sub-groups
      if (sub == 0) red[wave][gq][dl + i] = x;
    }
  DEC_STAMP(6);
  __syncthreads();
  // 8-byte write-through stores (pairs of floats; rows of D + 2 floats are 8-byte aligned because D is even)
  for (int w = tid; w < G * (D / 2); w += 256) {
    const int gq = w / (D / 2), d = 2 * (w - gq * (D / 2));
    float* wp = wbase + ((size_t)gq * DEC_SPLIT_MAX + split) * (D + 2);
    publish2(wp + d, red[0][gq][d] + red[1][gq][d] + red[2][gq][d] + red[3][gq][d],
             red[0][gq][d + 1] + red[1][gq][d + 1] + red[2][gq][d + 1] + red[3][gq][d + 1]);
    if (d == 0) publish2(wp + D, stat_m[gq], stat_l[gq]);
  }
  }  // !empty

  decode_ticket_merge<T, D, G, DEC_CHUNK_MAX>(wbase, tickets + (size_t)b * Hkv + hk, nsplit,
                                              out + ((size_t)b * Hq + (size_t)hk * G) * D, &sc[0][0], stat_m, stat_l, tid, lane, wave,
                                              hk == 0 && b == 0,
#ifdef SRGPT_TUNING_KNOBS
                                              stamp_base
#else
                                              -1
#endif
  );
}

// ------------------------------------------------------------------------------------------------
// decode attention, bf16 / head_dim 128 (every LLM geometry of the reference's recipes): round 3.
// The VALU kernel above spends ~10k of its ~23k cycles in three dependent-chain phases (scores with a cross-lane reduction per key
// and head, block-wide statistics, P.V with 32 cross-lane reductions per thread: profiles/r02_decode_attention_stamps.txt).  Here
//   * scores are ONE chain of four v_mfma_f32_16x16x32_bf16 per wave and 16 keys: A = K rows exactly as they lie in the cache
//     (lane = key, 16-byte d slices -- loaded straight from global in fragment shape), B = the G roped query heads (columns >= G
//     zero); the result leaves lane (head, key quad) with its 4 scores -- softmax statistics are two row swaps, no LDS, no barrier;
//   * P.V runs with lane = a pair of output dims over the wave's 16 keys (V rows as 256-byte wave loads, probabilities broadcast
//     from LDS): no cross-lane reduction at all;
//   * the four waves of a block (64 keys) meet once, through LDS, with per-wave (m, l) statistics -- one barrier;
// a block covers 64 keys (8 splits at <= 512 cached positions instead of 16): half as many partials for the merging block.
// All K / V row loads of a wave's first key group go out before anything depends on q.  Same partial format, ticket and merge.
// ------------------------------------------------------------------------------------------------
constexpr int DM_QLD = 136;  // bf16 row stride of the staged query heads (272 bytes: rows land on different bank groups)

template <int G>
__global__ __launch_bounds__(256) void decode_mfma_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ kcache,
                                                          bf16_t* __restrict__ vcache, const int* __restrict__ pos,
                                                          const bf16_t* __restrict__ cos_tab, const bf16_t* __restrict__ sin_tab,
                                                          float* __restrict__ ws, int* __restrict__ tickets,
                                                          bf16_t* __restrict__ out, int Hq, int Hkv, int max_pos, int nsplit,
                                                          int kpb, float scale, int n_attn, DecodePrefetch pf) {
  typedef bf16_t T;
  constexpr int D = 128, HALF = 64;
  __shared__ __attribute__((aligned(16))) bf16_t qs[16 * DM_QLD];  // B operand source: rows = query heads (>= G: zero)
  __shared__ __attribute__((aligned(16))) bf16_t knew[D], vnew[D];
  __shared__ __attribute__((aligned(16))) float pw[4][16][G];      // a wave's probabilities [key][head]
  __shared__ __attribute__((aligned(16))) float accs[4][G][D];     // per-wave P.V
  __shared__ float wm[4][G], wl[4][G];
  __shared__ float sc[G][64];                                      // merge weights (decode_ticket_merge)
  __shared__ float stat_m[G], stat_l[G];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if ((int)blockIdx.x >= n_attn) {  // ---- prefetch block (common.h) ----
    srgpt_prefetch_block(pf, (int)blockIdx.x - n_attn, wave, lane);
    return;
  }
  const int hk = (int)blockIdx.x % Hkv, split = ((int)blockIdx.x / Hkv) % nsplit, b = (int)blockIdx.x / (Hkv * nsplit);
#ifdef SRGPT_TUNING_KNOBS
  int stamp_base = blockIdx.x == 0 ? 0 : -1;
#endif
  DEC_STAMP(0);
  const int P = pos[b];
  const int total = P + 1;
  // FIXED key ranges: block `split` owns keys [split * kpb, (split + 1) * kpb), kpb = 64 for caches up to 4096 positions -- the
  // partition, hence the summation order and the bits, depend on the sequence length only, never on how large a cache the caller
  // happened to allocate (a pooled state serves requests of different sizes: scripts/soak.py caught exactly that); splits past the
  // sequence publish the neutral partial, which the merge adds as exact zeros
  const int kbeg = split * kpb, kend = min(kbeg + kpb, total);
  // Only the first ceil(total / kpb) splits own keys.  The blocks behind them leave at once -- no fetch, no RoPE, no partial, no
  // ticket: a state pooled for a long request (or max_length = 4096 out of a generation_config.json) costs a short request nothing
  // (ADVICE r3).  The merge covers the `nlive` partials only; the neutral partials the dead splits used to publish entered it as
  // exact zeros (weight exp(-inf) = 0), so the bits are the ones of the all-splits merge and stay independent of the capacity.
  const int nlive = min(nsplit, (total + kpb - 1) / kpb);
  if (split >= nlive) return;  // uniform per block
  const T* row = qkv + (size_t)b * (Hq + 2 * Hkv) * D;
  T* kc = kcache + ((size_t)b * Hkv + hk) * (size_t)max_pos * D;
  T* vc = vcache + ((size_t)b * Hkv + hk) * (size_t)max_pos * D;
  const int lq = lane & 15, g4 = lane >> 4;
  const int lastrow = max(P - 1, 0);  // rows >= P are not in the cache yet (row P is written by the split-0 block of THIS launch)

  // ---- K rows (fragment shape) and V rows (256-byte wave loads) of the wave's first 16-key group: one memory latency ----
  u32x4 kreg[4];
  unsigned int vreg[16];
  auto fetch = [&](int kb) {
    const T* kr = kc + (size_t)min(kb + lq, lastrow) * D + g4 * 8;
#pragma unroll
    for (int j = 0; j < 4; ++j) kreg[j] = *reinterpret_cast<const u32x4*>(kr + 32 * j);
#pragma unroll
    for (int i = 0; i < 16; ++i)
      vreg[i] = *reinterpret_cast<const unsigned int*>(vc + (size_t)min(kb + i, lastrow) * D + 2 * lane);
  };
  int kb = kbeg + 16 * wave;
  fetch(kb);

  DEC_STAMP(1);
  // ---- rotate q (G heads) and the new k; stage v.  Every global load of this stage is issued before any is consumed ----
  {
    constexpr int NITEM = (G + 1) * HALF;
    constexpr int ITEMS = (NITEM + 255) / 256;
    float x1v[ITEMS], x2v[ITEMS], cv[ITEMS], sv[ITEMS];
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
      const int w = (tid + 256 * it) % NITEM;
      const int hh = w / HALF, i = w - hh * HALF;
      const T* p = (hh < G) ? row + (size_t)(hk * G + hh) * D : row + (size_t)(Hq + hk) * D;
      x1v[it] = to_f(p[i]);
      x2v[it] = to_f(p[i + HALF]);
      cv[it] = to_f(cos_tab[(size_t)P * HALF + i]);
      sv[it] = to_f(sin_tab[(size_t)P * HALF + i]);
    }
    const T vraw = row[(size_t)(Hq + Hkv + hk) * D + (tid % D)];
    // query rows G .. 15 of the B operand are zero
    for (int i = tid; i < (16 - G) * (D / 2); i += 256) {
      const int r = G + i / (D / 2), c = 2 * (i % (D / 2));
      *reinterpret_cast<unsigned int*>(qs + r * DM_QLD + c) = 0u;
    }
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
      const int w0 = tid + 256 * it;
      if (w0 < NITEM) {
        const int hh = w0 / HALF, i = w0 - hh * HALF;
        const T o1 = from_f<T>(rnd<T>(x1v[it] * cv[it]) + rnd<T>(-x2v[it] * sv[it]));
        const T o2 = from_f<T>(rnd<T>(x2v[it] * cv[it]) + rnd<T>(x1v[it] * sv[it]));
        if (hh < G) {
          qs[hh * DM_QLD + i] = o1;
          qs[hh * DM_QLD + i + HALF] = o2;
        } else {
          knew[i] = o1;
          knew[i + HALF] = o2;
        }
      }
    }
    if (tid < D) vnew[tid] = vraw;
  }
  DEC_STAMP(2);
  __syncthreads();
  DEC_STAMP(3);
  if (split == 0) {  // exactly one block per (b, hk) appends; nobody reads position P from the cache
    if (tid < D / 2) {
      *reinterpret_cast<unsigned int*>(kc + (size_t)P * D + 2 * tid) = *reinterpret_cast<const unsigned int*>(knew + 2 * tid);
      *reinterpret_cast<unsigned int*>(vc + (size_t)P * D + 2 * tid) = *reinterpret_cast<const unsigned int*>(vnew + 2 * tid);
    }
  }
  // B operand: lane (head lq, d slice g4) -- the same four fragments for every key group
  bf16x8 qf[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) qf[j] = *reinterpret_cast<const bf16x8*>(qs + lq * DM_QLD + 32 * j + g4 * 8);
  u32x4 knf[4];  // the new key in fragment shape (for the lane whose key is P)
#pragma unroll
  for (int j = 0; j < 4; ++j) knf[j] = *reinterpret_cast<const u32x4*>(knew + 32 * j + g4 * 8);
  const unsigned int vnw = *reinterpret_cast<const unsigned int*>(vnew + 2 * lane);

  float m_run = -INFINITY, l_run = 0.f;  // per column (head lq): identical in the four lanes of a column
  float acc[G][2];
#pragma unroll
  for (int h = 0; h < G; ++h) acc[h][0] = acc[h][1] = 0.f;

  for (; kb < kend; kb += 64) {
    // ---- S^T = K Q^T: lane (head lq, quad g4) ends with the scores of keys kb + 4 g4 + r ----
    f32x4 sacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      u32x4 kv = kreg[j];
      if (kb + lq == P) kv = knf[j];
      sacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, kv), qf[j], sacc, 0, 0, 0);
    }
    float sv4[4], mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int key = kb + 4 * g4 + r;
      sv4[r] = key < kend ? sacc[r] * scale : -INFINITY;
      mx = fmaxf(mx, sv4[r]);
    }
    mx = rows_pair(mx, [](float a, float b2) { return fmaxf(a, b2); });
    mx = halves_pair(mx, [](float a, float b2) { return fmaxf(a, b2); });
    const float m_new = fmaxf(m_run, mx);  // kb < kend: at least one valid key, m_new is finite
    const float alpha = __expf(m_run - m_new);  // first group: exp(-inf) = 0
    float rs = 0.f;
    f32x4 pv;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      pv[r] = __expf(sv4[r] - m_new);
      rs += pv[r];
    }
    rs = rows_pair(rs, [](float a, float b2) { return a + b2; });
    rs = halves_pair(rs, [](float a, float b2) { return a + b2; });
    l_run = l_run * alpha + rs;
    m_run = m_new;
    // probabilities -> LDS [key][head] (a wave's own slab: in-order DS ops of ONE wave, no block barrier)
    if (lq < G) {
#pragma unroll
      for (int r = 0; r < 4; ++r) pw[wave][4 * g4 + r][lq] = pv[r];
    }
    // the rescale factor of the running P.V is per head: broadcast it the same way
    if (lq < G && g4 == 0) wm[wave][lq] = alpha;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int h = 0; h < G; ++h) {
      const float a_h = wm[wave][h];
      acc[h][0] *= a_h;
      acc[h][1] *= a_h;
    }
    // ---- P V: lane = output dims (2 lane, 2 lane + 1) ----
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int key = kb + i;
      unsigned int vv = vreg[i];
      if (key == P) vv = vnw;
      if (key >= kend) vv = 0u;  // clamped rows may hold anything (0 * NaN)
      const float v0 = bf16lo(vv), v1 = bf16hi(vv);
#pragma unroll
      for (int h = 0; h < G; ++h) {
        const float ph = pw[wave][i][h];
        acc[h][0] = fmaf(ph, v0, acc[h][0]);
        acc[h][1] = fmaf(ph, v1, acc[h][1]);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    if (kb + 64 < kend) fetch(kb + 64);  // contexts beyond 64 keys per split (max_pos > 64 * DEC_SPLIT_MAX never; > 512 here)
  }
  DEC_STAMP(4);
  // ---- the four waves meet: per-wave (m, l) and P.V through LDS ----
  if (lq < G && g4 == 0) {
    wm[wave][lq] = m_run;
    wl[wave][lq] = l_run;
  }
#pragma unroll
  for (int h = 0; h < G; ++h) *reinterpret_cast<f32x2*>(&accs[wave][h][2 * lane]) = f32x2{acc[h][0], acc[h][1]};
  DEC_STAMP(5);
  __syncthreads();
  DEC_STAMP(6);
  float* wbase = ws + (((size_t)b * Hkv + hk) * G) * (size_t)DEC_SPLIT_MAX * (D + 2);
  auto publish2 = [&](float* p, float a, float b2) {
    const unsigned long long u = (unsigned long long)__float_as_uint(a) | ((unsigned long long)__float_as_uint(b2) << 32);
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  for (int w = tid; w < G * (D / 2); w += 256) {
    const int gq = w / (D / 2), d = 2 * (w - gq * (D / 2));
    const float M = fmaxf(fmaxf(wm[0][gq], wm[1][gq]), fmaxf(wm[2][gq], wm[3][gq]));
    float n0 = 0.f, n1 = 0.f, L = 0.f;
#pragma unroll
    for (int wv = 0; wv < 4; ++wv) {  // fixed wave order
      const float e = (wm[wv][gq] > -INFINITY) ? __expf(wm[wv][gq] - M) : 0.f;
      n0 = fmaf(e, accs[wv][gq][d], n0);
      n1 = fmaf(e, accs[wv][gq][d + 1], n1);
      L = fmaf(e, wl[wv][gq], L);
    }
    float* wp = wbase + ((size_t)gq * DEC_SPLIT_MAX + split) * (D + 2);
    publish2(wp + d, n0, n1);
    if (d == 0) publish2(wp + D, M, L);
  }
  decode_ticket_merge<T, D, G, 64>(wbase, tickets + (size_t)b * Hkv + hk, nlive, out + ((size_t)b * Hq + (size_t)hk * G) * D,
                                   &sc[0][0], stat_m, stat_l, tid, lane, wave, hk == 0 && b == 0,
#ifdef SRGPT_TUNING_KNOBS
                                   stamp_base
#else
                                   -1
#endif
  );
}

template <typename T, int D>
int launch_decode_d(int G, const void* qkv, void* kc, void* vc, const int* pos, const void* ct, const void* st,
                    float* ws, int* tickets, void* out, int B, int Hq, int Hkv, int max_pos, int nsplit, float scale,
                    const DecodePrefetch& pf, hipStream_t s) {
  const int n_attn = Hkv * nsplit * B;
  dim3 grid(n_attn + (pf.base ? pf.nblocks : 0));
  if constexpr (std::is_same<T, bf16_t>::value && D == 128) {
    if (decode_use_mfma(1, D, G)) {
      const int kpb = cdiv(cdiv(max_pos, nsplit), 64) * 64;  // 64 keys per block up to 64 x 64 cached positions
#define LM(GG)                                                                                                              \
  hipLaunchKernelGGL((decode_mfma_kernel<GG>), grid, dim3(256), 0, s, (const bf16_t*)qkv, (bf16_t*)kc, (bf16_t*)vc, pos, \
                     (const bf16_t*)ct, (const bf16_t*)st, ws, tickets, (bf16_t*)out, Hq, Hkv, max_pos, nsplit, kpb, scale, n_attn, pf)
      switch (G) {
        case 1: LM(1); return SRGPT_OK;
        case 2: LM(2); return SRGPT_OK;
        case 4: LM(4); return SRGPT_OK;
        case 8: LM(8); return SRGPT_OK;
        default: break;
      }
#undef LM
    }
  }
#define LD(GG)                                                                                                     \
  hipLaunchKernelGGL((decode_split_kernel<T, D, GG>), grid, dim3(256), 0, s, (const T*)qkv, (T*)kc, (T*)vc, pos, \
                     (const T*)ct, (const T*)st, ws, tickets, (T*)out, Hq, Hkv, max_pos, nsplit, scale, n_attn, pf)
  switch (G) {
    case 1: LD(1); break;
    case 2: LD(2); break;
    case 4: LD(4); break;
    case 8: LD(8); break;
    default:
      srgpt_set_error("srgpt_decode_attention: heads/kv_heads = %d not supported (1,2,4,8)", G);
      return SRGPT_ERR_UNSUPPORTED;
  }
#undef LD
  return SRGPT_OK;
}

template <typename T>
int launch_decode(const void* qkv, void* kc, void* vc, const int* pos, const void* ct, const void* st, void* out,
                  float* ws, int B, int Hq, int Hkv, int D, int max_pos, const DecodePrefetch& pf, hipStream_t s) {
  const int G = Hq / Hkv;
  const bool mfma = decode_use_mfma(std::is_same<T, bf16_t>::value ? 1 : 0, D, G);
  const int nsplit = mfma ? decode_nsplit_mfma(max_pos) : decode_nsplit(max_pos, B, Hkv);
  // VALU kernel: a split's scores live in LDS (sc[G][DEC_CHUNK_MAX]): the longest chunk is ceil(max_pos / nsplit) keys
  SRGPT_CHECK(mfma || cdiv(max_pos, nsplit) <= DEC_CHUNK_MAX, SRGPT_ERR_UNSUPPORTED,
              "srgpt_decode_attention: max_pos %d exceeds %d cached positions", max_pos, DEC_SPLIT_MAX * DEC_CHUNK_MAX);
  const float scale = 1.0f / sqrtf((float)D);
  int* tickets = reinterpret_cast<int*>(ws + (size_t)B * Hq * DEC_SPLIT_MAX * (D + 2));  // int[B * Hkv] behind the partials
  int rc;
  switch (D) {
    case 16: rc = launch_decode_d<T, 16>(G, qkv, kc, vc, pos, ct, st, ws, tickets, out, B, Hq, Hkv, max_pos, nsplit, scale, pf, s); break;
    case 32: rc = launch_decode_d<T, 32>(G, qkv, kc, vc, pos, ct, st, ws, tickets, out, B, Hq, Hkv, max_pos, nsplit, scale, pf, s); break;
    case 64: rc = launch_decode_d<T, 64>(G, qkv, kc, vc, pos, ct, st, ws, tickets, out, B, Hq, Hkv, max_pos, nsplit, scale, pf, s); break;
    case 128: rc = launch_decode_d<T, 128>(G, qkv, kc, vc, pos, ct, st, ws, tickets, out, B, Hq, Hkv, max_pos, nsplit, scale, pf, s); break;
    default:
      srgpt_set_error("srgpt_decode_attention: head_dim %d not supported (16,32,64,128)", D);
      return SRGPT_ERR_UNSUPPORTED;
  }
  if (rc) return rc;
  SRGPT_LAUNCH_CHECK();
  return SRGPT_OK;
}

}  // namespace

#ifdef SRGPT_TUNING_KNOBS
extern "C" int srgpt_debug_stamps(unsigned long long* host, int n) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(srgpt_dbg_stamps), sizeof(unsigned long long) * (n < 32 ? n : 32));
}
#endif

extern "C" int64_t srgpt_decode_attn_ws_floats(int B, int Hq, int D) {
  return (int64_t)B * Hq * DEC_SPLIT_MAX * (D + 2) + (int64_t)B * Hq;  // partials + arrival tickets (<= B * Hkv ints)
}

// internal entry (model.hip): decode attention + L2 prefetch of the weight matrix the next GEMV streams.
// next_w = NULL -> no prefetch.  batch > 1 goes through the skinny kernel: no prefetch there (measured, common.h).
int srgpt_decode_attention_pf(const void* qkv, void* kcache, void* vcache, const int* pos, const void* cos_tab,
                              const void* sin_tab, void* out, float* ws, int B, int Hq, int Hkv, int D, int max_pos, int dtype,
                              const void* next_w, int next_n, int next_k, int next_fp8, int next_packed_rows, srgpt_stream_t stream) {
  SRGPT_CHECK(qkv && kcache && vcache && pos && cos_tab && sin_tab && out && ws, SRGPT_ERR_ARG,
              "srgpt_decode_attention: null pointer");
  SRGPT_CHECK(B > 0 && Hq > 0 && Hkv > 0 && Hq % Hkv == 0, SRGPT_ERR_ARG, "srgpt_decode_attention: bad heads");
  // 0 = off; 2 = all of o_proj (measured 3.189 / 3.169 / 3.138 ms per token at 0 / 1 / 2 rounds)
  const int pf_rounds = SRGPT_KNOB("SRGPT_DECODE_PREFETCH_ROUNDS", 2);
  // batched decode on packed fp8 weights (round 6): o_proj's 16-row tiles are contiguous and tile p belongs to block p of the next
  // launch -- the whole 16.8-MB matrix fits the 32 MB of L2; SRGPT_DECODE_PREFETCH_TILES = loads a prefetch wave keeps in flight (0: off)
  const int pf_tiles = SRGPT_KNOB("SRGPT_DECODE_PREFETCH_TILES", 0);
  const DecodePrefetch pf = dtype != SRGPT_BF16 ? srgpt_prefetch_for_gemv(nullptr, 0, 0, 0, 0, B, 0, 0)
                            // (16 consecutive rows of a ROW-MAJOR matrix are one contiguous tile too: block p of the batched product
                            //  streams rows 16 p .. 16 p + 15)
                            : ((next_packed_rows == 16 || next_packed_rows == 0) && B > 1) ? srgpt_prefetch_for_packed_tiles(next_w, next_n, next_k, next_fp8 ? 1 : 2, pf_tiles)
                                                                : srgpt_prefetch_for_gemv(next_w, next_n, next_k, 0, next_fp8, B, pf_rounds, 0);
  if (dtype == SRGPT_BF16)
    return launch_decode<bf16_t>(qkv, kcache, vcache, pos, cos_tab, sin_tab, out, ws, B, Hq, Hkv, D, max_pos, pf,
                                 as_stream(stream));
  if (dtype == SRGPT_F32)
    return launch_decode<float>(qkv, kcache, vcache, pos, cos_tab, sin_tab, out, ws, B, Hq, Hkv, D, max_pos, pf,
                                as_stream(stream));
  srgpt_set_error("srgpt_decode_attention: bad dtype %d", dtype);
  return SRGPT_ERR_ARG;
}

// the arrival tickets behind the partials (internal; model.hip zeroes them in every prefill so that a state whose workspace was
// never zeroed, or whose last launch was aborted, heals -- ADVICE r2 -- and srgpt_llm_decode_sync_state reads them back)
void* srgpt_decode_attn_sync_words(float* ws, int B, int Hq, int D, size_t* bytes) {
  if (bytes) *bytes = (size_t)B * Hq * sizeof(int);
  return ws + (size_t)B * Hq * DEC_SPLIT_MAX * (D + 2);
}

extern "C" int srgpt_decode_attention(const void* qkv, void* kcache, void* vcache, const int* pos, const void* cos_tab,
                                      const void* sin_tab, void* out, float* ws, int B, int Hq, int Hkv, int D,
                                      int max_pos, int dtype, srgpt_stream_t stream) {
  return srgpt_decode_attention_pf(qkv, kcache, vcache, pos, cos_tab, sin_tab, out, ws, B, Hq, Hkv, D, max_pos, dtype, nullptr, 0,
                                   0, 0, 0, stream);
}

extern "C" int srgpt_attention(const void* q, const void* k, const void* v, void* o, int B, int Tq, int Tk, int Hq,
                               int Hkv, int D, int64_t q_bs, int64_t q_ts, int64_t q_hs, int64_t k_bs, int64_t k_ts,
                               int64_t k_hs, int64_t v_bs, int64_t v_ts, int64_t v_hs, float scale, int causal,
                               const int* kv_len, int dtype, srgpt_stream_t stream) {
  SRGPT_CHECK(q && k && v && o, SRGPT_ERR_ARG, "srgpt_attention: null pointer");
  SRGPT_CHECK(B > 0 && Tq > 0 && Tk > 0 && Hq > 0 && Hkv > 0 && Hq % Hkv == 0 && D > 0 && D <= 256, SRGPT_ERR_ARG,
              "srgpt_attention: bad shape B=%d Tq=%d Tk=%d Hq=%d Hkv=%d D=%d", B, Tq, Tk, Hq, Hkv, D);
  hipStream_t s = as_stream(stream);
  const bool vec_ok = D % 8 == 0 && D <= 128 && q_ts % 8 == 0 && q_hs % 8 == 0 && q_bs % 8 == 0 && k_ts % 8 == 0 &&
                      k_hs % 8 == 0 && k_bs % 8 == 0 && v_ts % 8 == 0 && v_hs % 8 == 0 && v_bs % 8 == 0 &&
                      ((uintptr_t)q % 16 == 0) && ((uintptr_t)k % 16 == 0) && ((uintptr_t)v % 16 == 0) &&
                      ((uintptr_t)o % 8 == 0);
  // the MFMA kernel takes the row maximum before scaling (scale > 0) and addresses a (batch, head) slice with 32-bit byte offsets;
  // anything else goes to the one-wave-per-row kernel
  const int64_t span = ((int64_t)Tk + 4 * 64) * (k_ts > v_ts ? k_ts : v_ts) * 2;
  if (dtype == SRGPT_BF16 && vec_ok && scale > 0.f && span < srgpt_flash_slice_span_limit()) {
    AttnArgs a{(const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (bf16_t*)o, Tq, Tk, Hq, Hkv, D,
               q_bs, q_ts, q_hs, k_bs, k_ts, k_hs, v_bs, v_ts, v_hs, scale, kv_len};
    srgpt_flash_bf16_launch(a, B, causal != 0, s);
  } else if (dtype == SRGPT_BF16) {
    hipLaunchKernelGGL(simple_attn_kernel<bf16_t>, dim3(Tq, Hq, B), dim3(64), 0, s, (const bf16_t*)q, (const bf16_t*)k,
                       (const bf16_t*)v, (bf16_t*)o, Tq, Tk, Hq, Hkv, D, q_bs, q_ts, q_hs, k_bs, k_ts, k_hs, v_bs, v_ts,
                       v_hs, scale, causal, kv_len);
  } else if (dtype == SRGPT_F32) {
    hipLaunchKernelGGL(simple_attn_kernel<float>, dim3(Tq, Hq, B), dim3(64), 0, s, (const float*)q, (const float*)k,
                       (const float*)v, (float*)o, Tq, Tk, Hq, Hkv, D, q_bs, q_ts, q_hs, k_bs, k_ts, k_hs, v_bs, v_ts,
                       v_hs, scale, causal, kv_len);
  } else {
    srgpt_set_error("srgpt_attention: bad dtype %d", dtype);
    return SRGPT_ERR_ARG;
  }
  SRGPT_LAUNCH_CHECK();
  return SRGPT_OK;
}
