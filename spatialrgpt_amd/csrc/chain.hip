// EXPERIMENT -- compiled into the TUNING build only (Makefile), enabled there by SRGPT_DECODE_CHAIN=1; the product library does not
// contain it.  Status (round 3): bit-identical to the four launches it replaces (scripts/check_chain.py), weights stream at the
// full rate inside a phase, but each of the three in-launch edges costs ~10 us (tail of the last rows + store drain + grid barrier +
// reading the vector back + RMSNorm) against the 5 us the LDS ring can run ahead: 3.52 ms per token vs 2.96 for the launches
// (profiles/r03_gemv_chain.txt).  Next: tagged-granule hand-off instead of barrier + read-back.
//
// Persistent GEMV chain for the batch-1 decode step (round 3): o_proj -> gate/up (+RMSNorm, SwiGLU) -> down -> next layer's qkv
// (+RMSNorm) as ONE launch per layer instead of four.
//
// Why: the decode step is a chain of weight-streaming GEMVs, each bounded by HBM, separated by launch boundaries; every launch
// pays ~3 us of ramp / prologue / drain on top of its bytes, and pairwise fusion does not help because an
// in-launch all-to-all edge costs as much as the boundary it replaces -- unless the weight stream RUNS AHEAD of the edge.  The
// weights of the next phase depend on nothing, so here one loader wave per CU keeps streaming them (LDS-DMA,
// `global_load_lds_dwordx4`, no registers) into a 128-KiB ring while the consumers of the CU are still waiting for the previous
// phase's vector: 256 CUs x 128 KiB = 33.5 MB = ~5 us of HBM stream, more than the edge (grid barrier + reading the vector back).
// (guide: MI355X_MICROARCH.md "prefetch-credit", "engine-vs-launches": 0.87 - 0.89 x of the launches baseline.)
//
// One workgroup per CU (160 KB of LDS), 1 loader wave + 4 consumer waves:
//   loader    walks the block's granules (<= 8 KiB pieces of its weight rows, phase after phase, never waiting for a phase edge),
//             8 DMA instructions per granule (short granules are padded with re-reads so that `s_waitcnt vmcnt(56)` after a
//             granule means "the granule issued 7 granules ago has landed"), publishes a monotonic `ready` count in LDS, and
//             reuses a slot when its consumer has marked it free;
//   consumers each own whole output rows (round-robin), walk a row's granules in order and multiply from LDS with EXACTLY the
//             per-lane fma order of gemv.hip (lane l: 16-byte chunks l, l + 64, ... of the row; then the same wave_sum), so the
//             chain is bit-identical to the four launches it replaces; fused RMSNorm prologue (same 256-thread statistic as
//             gemv_kernel), SwiGLU / residual epilogues with gemv.hip's rounding points;
//   edges     a phase's outputs are stored write-through, every storing wave drains, the block arrives at an XCD-grouped,
//             generation-counted grid barrier (bounded spins; all blocks are resident by construction: grid = CUs, 1 block per CU),
//             then the consumers read the whole vector back with agent-scope loads and stage it (normalised) in LDS.
#include <type_traits>

#include "common.h"

namespace {

constexpr int CH_SLOT = 8192;   // bytes per ring slot = 8 wave-instructions of 1 KiB
constexpr int CH_NSLOT = 16;    // 128 KiB ring
constexpr int CH_INFLIGHT = 7;  // granules in flight: 7 * 8 = 56 DMA instructions <= 63 (vmcnt)
constexpr int CH_NC = 8;        // consumer waves: a row is ONE dependent fma chain per lane (bit-exactness), so the SIMDs need several
                                // waves each to hide it (4 consumers: 2.5k cycles per granule and wave, 10 GB/s per CU)
constexpr int CH_NS = 4;        // of which the first 4 (256 threads) stage the vector: gemv_kernel's statistic, thread for thread
constexpr int CH_THREADS = 64 * (1 + CH_NC);
constexpr int CH_XBYTES = 28672;  // staged input vector: K <= 14336 bf16
constexpr int CH_MAXPH = 4;
constexpr int CH_RING = CH_SLOT * CH_NSLOT;
constexpr int CH_LDS = CH_RING + CH_XBYTES + 1024;
constexpr int CH_SPIN = 1 << 22;  // bounded spins: far beyond any legitimate wait; an expiry sets the error word and lets the launch end

struct ChainPhase {
  const bf16_t* W;       // [N (2N if swiglu)][K]
  const bf16_t* norm_w;  // RMSNorm gain [K] or NULL
  const bf16_t* xin;     // [K]
  bf16_t* xout;          // [N]
  const bf16_t* resid;   // [N] or NULL (may alias xout)
  int N, K, swiglu;
};
struct ChainArgs {
  ChainPhase ph[CH_MAXPH];
  int nph;
  float eps;
  unsigned int* bar;  // grid-barrier words (zero between launches): [g * 16] group arrivals, [128] top, [144 + g * 16] group generation,
                      // [272] exit counter, [273] error
};

struct Ctrl {  // block-local control words in LDS
  unsigned int ready;             // granules landed (monotonic; single writer: the loader)
  unsigned int freed[CH_NSLOT];   // freed[s] = the next sequence number slot s may be filled for
  unsigned int cbar;              // consumer-only barrier (monotonic arrivals)
  unsigned int err;
  float red[8];
};

__device__ __forceinline__ unsigned int lds_load(const unsigned int* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void lds_store(unsigned int* p, unsigned int v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// The loader's own LDS traffic (poll of freed[], publish of ready) as raw instructions: the compiler orders every LDS access of a
// wave behind its pending LDS-DMA (`s_waitcnt vmcnt(0)` in front of each ds_read / ds_write it emits) -- the loader would drain
// its whole window once per granule (measured: 10 GB/s per CU).  These words are never DMA targets.
__device__ __forceinline__ unsigned int lds_addr(const void* p) {
  return (unsigned int)(uintptr_t)(const __attribute__((address_space(3))) void*)p;
}
__device__ __forceinline__ unsigned int lds_load_raw(const unsigned int* p) {
  unsigned int v;
  asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(lds_addr(p)) : "memory");
  return v;
}
__device__ __forceinline__ void lds_store_raw(unsigned int* p, unsigned int v) {
  asm volatile("ds_write_b32 %0, %1" ::"v"(lds_addr(p)), "v"(v) : "memory");
}

// rows of block b in a phase: units b, b + G, ...; a SwiGLU unit is the (gate row u, up row u + N) pair
__device__ __forceinline__ int units_of_block(int N, int b, int G) { return b < N ? (N - b + G - 1) / G : 0; }
__device__ __forceinline__ int granules_per_row(int K) { return (2 * K + CH_SLOT - 1) / CH_SLOT; }

// consumer-only barrier: the loader never takes part (it must keep streaming across phase edges), so no s_barrier here
__device__ __forceinline__ void consumer_barrier(Ctrl* c, unsigned int& epoch, int lane) {
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  epoch += CH_NC;
  if (lane == 0) {
    __hip_atomic_fetch_add(&c->cbar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    int it = 0;
    while (lds_load(&c->cbar) < epoch) {
      if (++it > CH_SPIN) {
        lds_store(&c->err, 1u);
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
  }
  __builtin_amdgcn_wave_barrier();
  asm volatile("" ::: "memory");
}

// grid barrier, edge e (0-based): groups of G / 8 blocks (block b -> group b % 8: the XCD it runs on, for speed only)
__device__ __forceinline__ void grid_arrive_wait(unsigned int* bar, int e, int b, int G, Ctrl* c) {
  const int g = b & 7;
  const unsigned int per = (unsigned int)(G >> 3);
  const unsigned int old = __hip_atomic_fetch_add(bar + g * 16, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (old == per * (unsigned int)(e + 1) - 1u) {
    const unsigned int t = __hip_atomic_fetch_add(bar + 128, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (t == 8u * (unsigned int)(e + 1) - 1u)
      for (int q = 0; q < 8; ++q) __hip_atomic_store(bar + 144 + q * 16, (unsigned int)(e + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  int it = 0;
  while (__hip_atomic_load(bar + 144 + g * 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned int)(e + 1)) {
    if (++it > CH_SPIN) {
      __hip_atomic_store(bar + 273, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      lds_store(&c->err, 1u);
      break;
    }
    __builtin_amdgcn_s_sleep(2);
  }
}

#ifdef SRGPT_TUNING_KNOBS
// phase stamps of block 0 (tuning build; scripts/ubench_chain_stamps.py): [0] entry; consumer wave 0: [1 + 4p .. 4 + 4p] = vector
// staged / rows done / arrived at the grid barrier / released, per phase p; loader: [20 + p] = phase p's last granule issued
__device__ unsigned long long srgpt_chain_stamps[32];
#define CH_STAMP(i) do { if (blockIdx.x == 0 && lane == 0) srgpt_chain_stamps[(i)] = __builtin_amdgcn_s_memtime(); } while (0)
extern "C" int srgpt_chain_debug_stamps(unsigned long long* host, int n) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(srgpt_chain_stamps), sizeof(unsigned long long) * (n < 32 ? n : 32));
}
#else
#define CH_STAMP(i) do { } while (0)
#endif

__global__ __launch_bounds__(CH_THREADS, 1) void gemv_chain_kernel(ChainArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* ring = smem;
  bf16_t* xs = reinterpret_cast<bf16_t*>(smem + CH_RING);
  Ctrl* ctrl = reinterpret_cast<Ctrl*>(smem + CH_RING + CH_XBYTES);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x, G = gridDim.x;

  if (tid < CH_NSLOT) ctrl->freed[tid] = (unsigned int)tid;
  if (tid == 0) {
    ctrl->ready = 0;
    ctrl->cbar = 0;
    ctrl->err = 0;
  }
  __syncthreads();  // the only block-wide barrier: before the roles diverge
  if (wave == 1) CH_STAMP(0);

  if (wave == 0) {
    // ================================ loader ================================
    unsigned int k = 0;  // granule sequence number
    for (int p = 0; p < a.nph; ++p) {
      const ChainPhase& ph = a.ph[p];
      const int RB = 2 * ph.K, gpr = granules_per_row(ph.K);
      const int nu = units_of_block(ph.N, b, G), R = ph.swiglu ? 2 : 1;
      for (int j = 0; j < nu; ++j)
        for (int r = 0; r < R; ++r) {
          const unsigned char* rowp = reinterpret_cast<const unsigned char*>(ph.W + (size_t)(b + j * G + r * ph.N) * ph.K);
          for (int g = 0; g < gpr; ++g, ++k) {
            const int slot = (int)(k % CH_NSLOT);
            int it = 0;
            while (lds_load_raw(&ctrl->freed[slot]) != k) {  // its previous granule has been consumed
              if (++it > CH_SPIN) {
                lds_store_raw(&ctrl->err, 1u);
                break;
              }
              __builtin_amdgcn_s_sleep(1);
            }
#pragma unroll
            for (int t = 0; t < 8; ++t) {
              // bytes past the row's end (short last granule, padding instructions) re-read the row's last 16 bytes
              const int off = min(g * CH_SLOT + t * 1024 + lane * 16, RB - 16);
              __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(rowp + off),
                                               (__attribute__((address_space(3))) void*)(ring + slot * CH_SLOT + t * 1024), 16, 0, 2);
            }
            // 8 instructions per granule, always: at most 56 outstanding <=> the granule issued 7 granules ago has landed
            asm volatile("s_waitcnt vmcnt(56)" ::: "memory");
            if (k >= (unsigned int)CH_INFLIGHT && lane == 0) lds_store_raw(&ctrl->ready, k - CH_INFLIGHT + 1);
          }
        }
      CH_STAMP(20 + p);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) lds_store_raw(&ctrl->ready, k);
    return;
  }

  // ================================ consumers ================================
  const int cw = wave - 1;              // consumer wave 0 .. CH_NC - 1
  const int ctid = cw * 64 + lane;      // 0 .. 255: gemv_kernel's thread id
  unsigned int cb_epoch = 0;
  unsigned int base = 0;                // sequence number of the phase's first granule (of this block)
  for (int p = 0; p < a.nph; ++p) {
    const ChainPhase& ph = a.ph[p];
    const int K = ph.K, nchunks = K >> 3, gpr = granules_per_row(K);
    const int nu = units_of_block(ph.N, b, G), R = ph.swiglu ? 2 : 1;
    // ---- stage the input vector (RMSNorm: gemv_kernel's statistic, thread for thread) ----
    {
      const bool fresh = p > 0;  // produced by other CUs inside this launch: agent-scope loads
      const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(ph.xin), 0, 2 * K, 0x00020000);
      float ss = 0.f;
      for (int c = ctid; c < nchunks && cw < CH_NS; c += 256) {
        const u32x4 v = fresh ? __builtin_amdgcn_raw_buffer_load_b128(xr, c * 16, 0, 16) : __builtin_amdgcn_raw_buffer_load_b128(xr, c * 16, 0, 0);
        *reinterpret_cast<u32x4*>(xs + (size_t)c * 8) = v;
        float sq = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float lo = bf16lo(v[q]), hi = bf16hi(v[q]);
          sq += lo * lo;
          sq += hi * hi;
        }
        ss += sq;
      }
      if (ph.norm_w) {
        // block_sum of gemv_kernel (common.h): wave_sum, then the waves' sums added in wave order
        float wsum = wave_sum(ss);
        consumer_barrier(ctrl, cb_epoch, lane);  // (also: every consumer is done with the previous phase's xs / red)
        if (lane == 0 && cw < CH_NS) ctrl->red[cw] = wsum;
        consumer_barrier(ctrl, cb_epoch, lane);
        float tot = 0.f;
        for (int i = 0; i < CH_NS; ++i) tot += ctrl->red[i];
        const float rs = rsqrtf(tot / (float)K + a.eps);
        for (int c = ctid; c < nchunks && cw < CH_NS; c += 256) {
          bf16x8 xv = *reinterpret_cast<const bf16x8*>(xs + (size_t)c * 8);
          const bf16x8 gv = *reinterpret_cast<const bf16x8*>(ph.norm_w + (size_t)c * 8);
#pragma unroll
          for (int i = 0; i < 8; ++i) xv[i] = (bf16_t)((float)gv[i] * rnd<bf16_t>((float)xv[i] * rs));  // weight * h.to(dtype)
          *reinterpret_cast<bf16x8*>(xs + (size_t)c * 8) = xv;
        }
      }
      consumer_barrier(ctrl, cb_epoch, lane);
    }
    if (cw == 0) CH_STAMP(1 + 4 * p);
    // ---- the wave's units ----
    for (int j = cw; j < nu; j += CH_NC) {
      const int unit = b + j * G;
      float acc[2] = {0.f, 0.f};
      for (int r = 0; r < R; ++r) {
        const unsigned int seq0 = base + (unsigned int)((j * R + r) * gpr);
        for (int g = 0; g < gpr; ++g) {
          const unsigned int seq = seq0 + g;
          int it = 0;
          while (lds_load(&ctrl->ready) <= seq) {
            if (++it > CH_SPIN) {
              lds_store(&ctrl->err, 1u);
              break;
            }
            __builtin_amdgcn_s_sleep(1);
          }
          asm volatile("" ::: "memory");
          const int slot = (int)(seq % CH_NSLOT);
          const unsigned char* sp = ring + slot * CH_SLOT;
          float accr = acc[r];
#pragma unroll
          for (int t = 0; t < 8; ++t) {
            const int ch = g * (CH_SLOT / 16) + t * 64 + lane;  // chunk of the row
            const bool valid = ch < nchunks;
            const u32x4 wv = *reinterpret_cast<const u32x4*>(sp + t * 1024 + lane * 16);
            const u32x4 xv = *reinterpret_cast<const u32x4*>(xs + (size_t)(valid ? ch : 0) * 8);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const unsigned int wq = valid ? wv[q] : 0u;
              accr = fmaf(bf16lo(wq), bf16lo(xv[q]), accr);
              accr = fmaf(bf16hi(wq), bf16hi(xv[q]), accr);
            }
          }
          acc[r] = accr;
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the slot's bytes are in registers: the loader may refill it
          if (lane == 0) lds_store(&ctrl->freed[slot], seq + CH_NSLOT);
        }
      }
      const float a0 = wave_sum(acc[0]);
      const float a1 = ph.swiglu ? wave_sum(acc[1]) : 0.f;
      if (lane == 0) {
        float v;
        if (ph.swiglu) {
          const float gq = rnd<bf16_t>(a0), uq = rnd<bf16_t>(a1);
          v = rnd<bf16_t>(silu(gq)) * uq;
        } else {
          v = rnd<bf16_t>(a0);
          if (ph.resid) {
            // the residual element may have been written by THIS block in an earlier phase of this launch (o_proj -> down):
            // agent-scope load, not an L1 hit on the old line
            const unsigned short rb = __hip_atomic_load(reinterpret_cast<const unsigned short*>(ph.resid + unit), __ATOMIC_RELAXED,
                                                        __HIP_MEMORY_SCOPE_AGENT);
            v = rnd<bf16_t>((float)__builtin_bit_cast(bf16_t, rb) + v);
          }
        }
        const bf16_t o = (bf16_t)v;
        // write-through: the next phase's consumers on every CU read it inside this launch
        __hip_atomic_store(reinterpret_cast<unsigned short*>(ph.xout + unit), __builtin_bit_cast(unsigned short, o), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    base += (unsigned int)(nu * R * gpr);
    if (cw == 0) CH_STAMP(2 + 4 * p);
    // ---- edge: every storing wave drains, the block arrives at the grid barrier, everyone waits for the generation ----
    if (p + 1 < a.nph) {
      consumer_barrier(ctrl, cb_epoch, lane);  // (drains vmcnt)
      if (cw == 0) CH_STAMP(3 + 4 * p);
      if (cw == 0 && lane == 0) grid_arrive_wait(a.bar, p, b, G, ctrl);
      consumer_barrier(ctrl, cb_epoch, lane);
      if (cw == 0) CH_STAMP(4 + 4 * p);
    }
  }
  // ---- exit: the last block out re-arms the barrier words for the next launch on this stream ----
  consumer_barrier(ctrl, cb_epoch, lane);
  if (cw == 0 && lane == 0) {
    if (lds_load(&ctrl->err)) __hip_atomic_store(a.bar + 273, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned int t = __hip_atomic_fetch_add(a.bar + 272, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (t == (unsigned int)G - 1u) {
      for (int q = 0; q < 8; ++q) {
        __hip_atomic_store(a.bar + q * 16, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(a.bar + 144 + q * 16, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      __hip_atomic_store(a.bar + 128, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(a.bar + 272, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

}  // namespace

// internal (model.hip): `nph` dependent batch-1 bf16 GEMVs (phase p + 1 reads what phase p wrote) as one persistent launch.
// ph[i] = {W, norm_w | NULL, xin, xout, resid | NULL, N, K, swiglu}.  Returns SRGPT_ERR_UNSUPPORTED for shapes it does not take
// (the caller then launches the GEMVs one by one).
int srgpt_gemv_chain(const void* const* W, const void* const* norm_w, const void* const* xin, void* const* xout,
                     const void* const* resid, const int* N, const int* K, const int* swiglu, int nph, float eps, void* bar,
                     srgpt_stream_t stream) {
  SRGPT_CHECK(nph >= 1 && nph <= CH_MAXPH && bar, SRGPT_ERR_ARG, "srgpt_gemv_chain: bad arguments");
  const int cus = srgpt_device_cus();
  if (cus % 8 != 0) return SRGPT_ERR_UNSUPPORTED;
  ChainArgs a;
  a.nph = nph;
  a.eps = eps;
  a.bar = reinterpret_cast<unsigned int*>(bar);
  for (int i = 0; i < nph; ++i) {
    if (K[i] % 8 != 0 || 2 * K[i] > CH_XBYTES || K[i] < 64 || N[i] < 1) return SRGPT_ERR_UNSUPPORTED;
    a.ph[i] = ChainPhase{reinterpret_cast<const bf16_t*>(W[i]), reinterpret_cast<const bf16_t*>(norm_w[i]),
                         reinterpret_cast<const bf16_t*>(xin[i]), reinterpret_cast<bf16_t*>(xout[i]),
                         reinterpret_cast<const bf16_t*>(resid[i]), N[i], K[i], swiglu[i]};
  }
  static std::atomic<uint64_t> attr_done{0};
  SRGPT_TRY(srgpt_ensure_dyn_lds(attr_done, (const void*)gemv_chain_kernel, CH_LDS));
  hipLaunchKernelGGL(gemv_chain_kernel, dim3(cus), dim3(CH_THREADS), CH_LDS, as_stream(stream), a);
  SRGPT_LAUNCH_CHECK();
  return SRGPT_OK;
}
