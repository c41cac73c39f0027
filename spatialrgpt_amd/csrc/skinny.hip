// Batched decode-path weight-streaming product for 2..16 rows: out[b, n] = W[n, :] . x[b, :]   (bf16, HBM-bound)
//
// Same contract and fusions as the GEMV (gemv.hip: RMSNorm prologue, SwiGLU epilogue, residual add, fp32 logits), but
// above 4 rows the VALU formulation runs out of issue slots and LDS bandwidth (B LDS reads + 8 B FMAs per 16 weight
// bytes), so the 16-row batch becomes the M side of v_mfma_f32_16x16x32_bf16 and 16 weight rows the N side.
//
// Roofline: weight bytes read exactly once (non-temporal, each wave-instruction = 2 rows x 512 contiguous bytes).
// MFMA operand fragments need lane (r = l & 15, g = l >> 4) to hold 16 bytes of ROW r -- loading them straight from
// HBM puts 16 different rows in one instruction (64-byte pieces; measured slow, scripts/experiments/ubench_stream.hip), so each
// wave stages its 16 x 256 weight tile through a PRIVATE 8 KiB LDS region: 8 coalesced global loads -> 8 ds_write_b128
// -> 8 conflict-free ds_read_b128 fragments (row stride padded to 544 B).  The matching [rows][256] slice of the
// (normalised) activations goes through a second private region the same way (it comes from L2).  Nothing in the K loop
// is shared between waves, so there is NO block barrier in it (LDS ops of one wave execute in order) and the 8-12 waves
// of a CU drift apart like the GEMV's do: some stream while others multiply.
//
// Work decomposition: block b owns the `cw` consecutive output columns [b cw, (b + 1) cw), cut into tiles of 16 (the last one
// partial: its missing rows are not loaded); cw = ceil(N / blocks) makes every CU stream the same number of weight rows
// (6144 q/k/v columns = 24 per CU, 14336 gate/up pairs = 56 -- whole 16-column units left a quarter of the CUs with half the
// work).  A tile is one sub-unit (SwiGLU: the 16 gate rows + the 16 matching up rows = 2 sub-units); up to 4 sub-units per
// pass keep their 16x16 fp32 accumulators in registers (2 x 4 VGPRs each).  The waves of a block split K into one contiguous
// range of 256-wide slices each (round 3; interleaved slices before: see SRGPT_SKINNY_CONTIG below), so every block uses all its
// waves even when it owns a single tile (o_proj / down_proj); their partial sums are added in a fixed order through LDS at the end
// of the pass (deterministic).
//
// The K loop is written for counted waits: every load in it is issued unconditionally, so the compiler can leave the
// prefetched stage in flight across the LDS writes (a branch around a load makes it wait for vmcnt(0) -- the prefetch was being
// drained every stage); the fragment reads of 4 k steps go out together in front of their MFMAs, which alternate between two
// accumulators (the compiler's own order was read -> wait -> MFMA per step: a full LDS latency each).
#include <stdlib.h>

#include <type_traits>

#include "common.h"

int srgpt_gemv_w8_valu(const void* x, const void* W8, const float* wscale, const void* norm_w, float eps,
                       const void* residual, void* out, int batch, int N, int K, int swiglu, int out_f32, hipStream_t s);  // gemv_w8.hip

namespace {

#ifdef SRGPT_TUNING_KNOBS
// phase stamps of block 0 / wave 0 of the last skinny launch (tuning build only; scripts/ubench_skinny_stamps.py)
__device__ unsigned long long srgpt_skinny_stamps[16];
#define SK_STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) srgpt_skinny_stamps[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define SK_STAMP(i) do { } while (0)
#endif

// Compile-time variants (-DSRGPT_SKINNY_*) and the timing probes below exist for the tuning build only: a product build that
// defines one of them is refused (one stray flag must not ship a different -- or wrong -- product).
#ifndef SRGPT_TUNING_KNOBS
#if defined(SRGPT_SKINNY_DEPTH) || defined(SRGPT_SKINNY_DEPTH_W8) || defined(SRGPT_SKINNY_XREUSE) || defined(SRGPT_SKINNY_PRE) || \
    defined(SRGPT_SKINNY_PRE_W8) || defined(SRGPT_SKINNY_FS) || defined(SRGPT_SKINNY_PROBE) || defined(SRGPT_SKINNY_CONTIG) || \
    defined(SRGPT_SKINNY_PK_WPS) || defined(SRGPT_SKINNY_ONE_ACC) || defined(SRGPT_SKINNY_FIX_NSU)
#error "skinny.hip: -DSRGPT_SKINNY_* variants need the tuning build (make TUNING=1)"
#endif
#endif

constexpr int WROWB = 512 + 32;        // bytes per staged weight row: 136 dwords = 8 mod 64 banks -> the lane groups of ds_read_b128
                                       // (MI355X guide, LDS table) hit distinct banks; 528 measured 30 % conflict cycles
constexpr int WSTAGEB = 16 * WROWB;    // weight stage per wave
constexpr int MAXSU = 4;               // sub-units (16-row weight tiles) accumulated per pass (8 for one 8-wave SwiGLU block per CU that
                                       // walks K once: measured equal, profiles/r06_skinny_gateup_8wave.txt)

#ifndef SRGPT_SKINNY_DEPTH
#define SRGPT_SKINNY_DEPTH 2           // register ring depth of weight stages (tuning builds override)
#endif
#ifndef SRGPT_SKINNY_DEPTH_W8
#define SRGPT_SKINNY_DEPTH_W8 2        // the same for fp8 weights (a stage is half the registers and half the bytes in flight)
#endif
#ifndef SRGPT_SKINNY_XREUSE
#define SRGPT_SKINNY_XREUSE 1          // activation fragments of a slice kept in registers across its sub-units
#endif
#ifndef SRGPT_SKINNY_PRE
#define SRGPT_SKINNY_PRE 1            // first weight stage of a block requested before its RMSNorm statistics are reduced
#endif
#ifndef SRGPT_SKINNY_PRE_W8
#define SRGPT_SKINNY_PRE_W8 0         // the same with fp8 weights (measured slower in round 2; re-measured in round 3, see profiles/)
#endif
#ifndef SRGPT_SKINNY_FS
#define SRGPT_SKINNY_FS 4              // MFMA k steps per fragment batch (8 LDS reads in flight per batch)
#endif

// NI = x rows staged per wave / 2: 2 (batch <= 4), 4 (batch <= 8) or 8 (batch <= 16).  NW = waves per block (the K split): two 4-wave
// blocks per CU, or one 8-wave block per CU (the launcher's rule, below).
// W8: the weights are OCP fp8 e4m3fn bytes with one fp32 scale per weight row (W8A16): a stage is 8 loads of 8 bytes per lane
// (2 rows x 256 bytes each), widened to bf16 (exact) on the way into LDS; the row scale multiplies the fp32 dot product.
// (Measured and dropped: 512-k fp8 slices with the raw bytes in LDS and the widening behind the fragment read -- a stage of as
// many bytes as a bf16 one -- 69.2 vs 63.7 us per layer at 8 rows: this kernel is not bound by bytes in flight.)
//
// Row statistics hand-off (round 5).  The RMSNorm in front of q/k/v, gate/up and lm_head needs sum(x^2) of every batch row; computed
// in the consumer, EVERY block re-read all rows (8 x 8 KiB at 8 rows) and reduced them through two block barriers before its first
// weight stage could even be requested: 7.7k of the q/k/v launch's 22k cycles at 8 fp8 rows (profiles/r02_skinny_stamps.txt).  The
// product that WRITES x (o_proj / down_proj with the residual add) has every element of it in a register: with `ss_out` it publishes,
// per block, the sum of squares of the columns it owns -- slot [row][block] of a [rows][SRGPT_ROWSS_STRIDE] fp32 table, slots past
// the grid zeroed, plain stores (the kernel boundary makes them visible).  A consumer instantiated with PUB reads its rows' 512 slots
// (one wave per row, two 16-byte loads per lane), adds them in a fixed order (lane-local, then the DPP / permlane tree) -- the
// result does not depend on which block finished first -- and goes straight to its weight stream: the statistics' loads, the first
// activation slice and the first weight stage are all requested back to back at kernel entry, one memory latency for the three.
// PK (round 6): the weights are read from the PACKED decode layout (srgpt_pack_decode_weights, include/srgpt.h): granules of `gr`
// rows (4 or 16) whose bytes are stored in MFMA-operand order -- [granule][k block][g = k / 8 % 4][row in granule][16 bytes] -- so
// that ONE coalesced 16-byte load per lane IS the B fragment of a k step (bf16: a k block = 32 k; fp8: 64 k = the fragments of two
// k steps); a wave-instruction reads 16 / gr pieces of gr x 64 contiguous bytes and a wave walks its K range through each granule
// sequentially.  The weights then never touch LDS: no ds_write / wave barrier / fragment ds_read per stage (the timing probe of
// round 5 priced them at 4 - 7 % of the fp8 layer, profiles/r05_skinny_probes.txt) and a block needs LDS only for its activation
// slices.  4-row granules keep every column split of the row-major kernel (24 q/k/v columns or 28 gate / up pairs per block);
// 16-row granules (1 KiB contiguous per instruction) stream the single-tile products faster (profiles/r06_skinny_packed.txt).
template <bool SWIGLU, int NI, int NW, bool W8, bool PUB, bool PK>
#ifndef SRGPT_SKINNY_PK_WPS
#define SRGPT_SKINNY_PK_WPS 2  // waves per SIMD the packed fp8 4-wave kernel is compiled for (tuning builds: 3 / 4 = 3 / 4 blocks per CU)
#endif
__global__ __launch_bounds__(64 * NW, NW == 4 ? (PK && W8 ? SRGPT_SKINNY_PK_WPS : 2) : 1) void skinny_kernel(const bf16_t* __restrict__ x, const void* __restrict__ Wv,
                                                                          const float* __restrict__ wscale,
                                                                          const bf16_t* __restrict__ norm_w, float norm_eps,
                                                                          const bf16_t* __restrict__ residual, void* __restrict__ out,
                                                                          int B, int N, int K, int out_f32, int cw,
                                                                          const float* __restrict__ ss_in, float* __restrict__ ss_out,
                                                                          int gr_shift) {
  constexpr int SK = 256;                   // k per wave slice
  constexpr int XROWB = WROWB;              // bytes per staged activation row
  constexpr int XL = NI;                    // activation loads per slice: 2 rows x 512 B each
  constexpr int R = SWIGLU ? 2 : 1;  // weight tiles (sub-units) per 16-column output tile
  constexpr int MAXU = MAXSU / R;
  constexpr int CPT = 16;            // output columns per tile
  constexpr int RS = R;              // weight rows (and fp8 row scales) per output
  constexpr int XSTAGEB = 2 * NI * XROWB;
  constexpr int NT = 64 * NW;
  using WReg = typename std::conditional<W8 && !PK, u32x2, u32x4>::type;  // one weight load per lane
  constexpr int WEB = W8 ? 1 : 2;                                       // bytes per weight element
  constexpr int WEPL = 8;                                               // weight elements per lane-load (row-major layout)
  constexpr int NWL = PK && W8 ? 4 : 8;                                 // loads per weight stage (packed fp8: 64 k per load)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // per wave: [x stage | w stage]; reused for the reduction
  __shared__ float rs_s[16];
  __shared__ float ss_s[16][NW / 2];
  __shared__ float pub_s[NW][4];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  unsigned char* xst = smem + wave * (XSTAGEB + (PK ? 0 : WSTAGEB));
  unsigned char* wst = xst + XSTAGEB;  // (row-major layout only)
  const int nsl = (K + SK - 1) / SK;  // K slices; wave w takes w, w + NW, ...
  const int c0 = (int)blockIdx.x * cw;                 // first output column of this block
  const int cwb = min(cw, N - c0);                     // its column count (the last block may own fewer)
  const int ntile = (cwb + CPT - 1) / CPT;             // tiles, the last one possibly partial
  const int npass = (ntile + MAXU - 1) / MAXU;
  const bool do_norm = PUB || norm_w != nullptr;  // (published statistics are the RMSNorm's: a compile-time fact for those instances)
  const int lrow = lane >> 5, lchunk = lane & 31;  // staging loads: lane -> (row parity, 16-byte chunk of the 512-byte row piece)
  const int xchunk = lchunk;
  auto xrow_of = [&](int j) { return 2 * j + lrow; };

  // activation slice: XL loads covering 2*NI rows x SK k (rows past the batch re-read row B-1: their outputs are never
  // stored), plus the RMSNorm gains of the lane's chunk
  // The slices live in a register ring of XS slots (slot of slice index j: j % XS); x of slice j + XS is requested right after slice
  // j has been staged.  Vector loads return IN ORDER: with one slot, waiting for the next slice's activations also waits for every
  // weight stage requested before them -- a product whose slices are ONE weight stage (o_proj / down_proj: a single 16-column tile)
  // then never has more than one stage in flight whatever the weight ring's depth (round 6; run_pass picks XS per sub-unit count).
  constexpr int DEPTH = W8 ? SRGPT_SKINNY_DEPTH_W8 : SRGPT_SKINNY_DEPTH;
  constexpr int XSMAX = DEPTH;
  u32x4 xr[XSMAX][XL];
  u32x4 gr[XSMAX];
  auto load_x = [&](auto slot_c, int sl) {
    constexpr int slot = decltype(slot_c)::value < XSMAX ? decltype(slot_c)::value : 0;  // (slots >= XS are never requested: the clamp only keeps -Warray-bounds quiet about discarded calls)
    int kg = min(sl * SK + xchunk * 8, K - 8);
    asm volatile("" : "+v"(kg));  // keep the row products out of loop-invariant registers (see issue_w)
#pragma unroll
    for (int j = 0; j < XL; ++j) {
      const unsigned off = (unsigned)min(xrow_of(j), B - 1) * (unsigned)K + (unsigned)kg;
      xr[slot][j] = *reinterpret_cast<const u32x4*>(x + off);
    }
    gr[slot] = *reinterpret_cast<const u32x4*>((do_norm ? norm_w : x) + kg);  // unconditional: see the note on counted waits below
  };
  // PUB instances (the decode step's RMSNorm products): 1 / rms of the rows this lane stages (rows 2 j + lrow) in registers once the
  // statistics are known -- read from LDS inside stage_x they cost an LDS round trip + lgkmcnt(0) per row pair and slice in the K
  // loop -- and the two roundings on element PAIRS (v_pk_mul_f32, one v_cvt_pk_bf16_f32 per rounding of a pair: 8 instead of 11
  // VALU per two elements; same operations, same bits).  The other instances keep the scalar form: with bf16 weights their 8-row
  // residual kernels are at the register limit and the extra live values spilled (o / down 9.2 / 22.6 -> 9.9 / 24.4 us, round 6).
  constexpr bool REGNORM = PUB && NI <= 4;  // (16 staged rows: the registers are not there either)
  float rsr[REGNORM ? XL : 1];
  auto load_rs = [&]() {
    if constexpr (REGNORM) {
#pragma unroll
      for (int j = 0; j < XL; ++j) rsr[j] = rs_s[xrow_of(j)];
    }
  };
  auto stage_x = [&](auto slot_c, int sl, bool valid) {
    constexpr int slot = decltype(slot_c)::value;
    const bool kvalid = valid && sl * SK + xchunk * 8 < K;
#pragma unroll
    for (int j = 0; j < XL; ++j) {
      u32x4 v = xr[slot][j];
      if (do_norm) {
        if constexpr (REGNORM) {
          const float rsj = rsr[j];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            // weight * hidden.to(dtype): two roundings, like the GEMV prologue
            f32x2 xv = {bf16lo(v[q]), bf16hi(v[q])};
            xv = xv * f32x2{rsj, rsj};
            const unsigned int hb = __builtin_bit_cast(unsigned int, __builtin_convertvector(xv, bf16x2));
            f32x2 o = f32x2{bf16lo(gr[slot][q]), bf16hi(gr[slot][q])} * f32x2{bf16lo(hb), bf16hi(hb)};
            v[q] = __builtin_bit_cast(unsigned int, __builtin_convertvector(o, bf16x2));
          }
        } else {
          const float rsj = rs_s[xrow_of(j)];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float lo = bf16lo(gr[slot][q]) * rnd<bf16_t>(bf16lo(v[q]) * rsj);
            const float hi = bf16hi(gr[slot][q]) * rnd<bf16_t>(bf16hi(v[q]) * rsj);
            bf16x2 p;
            p[0] = (bf16_t)lo;
            p[1] = (bf16_t)hi;
            v[q] = __builtin_bit_cast(unsigned int, p);
          }
        }
      }
      if (!kvalid) v = u32x4{0u, 0u, 0u, 0u};  // k past K contributes zeros (the weight loads there are clamped)
      *reinterpret_cast<u32x4*>(xst + xrow_of(j) * XROWB + xchunk * 16) = v;
    }
  };

  // weight stage of sub-unit su of (pass, slice): 8 loads, each 2 rows x 512 contiguous bytes (fp8: 2 rows x 256 bytes)
  // Every load inside the K loop is issued UNCONDITIONALLY: behind a branch the compiler cannot count the loads in flight, waits
  // for vmcnt(0) in front of the LDS writes and so drains the prefetched stage every step -- one full memory latency per stage,
  // which is what bounded this kernel before.  The loads are BUFFER loads on a descriptor of the tile's valid rows: what must not
  // be fetched (a stage past the wave's last slice, rows past the tile's last one, k past K) is pushed out of the descriptor's
  // range and comes back as zeros without a memory access -- no per-load select, and the per-lane part of the address is eight
  // kernel-constant offsets (row 2j + lrow, chunk lchunk) plus one slice offset per stage: 12 VALU per stage where the
  // flat-address form spent 45 (the ISA of round 2's kernel: 126 v_cndmask + 104 shifts / adds per 8 stages).
  unsigned wrow_off[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) wrow_off[j] = ((unsigned)(2 * j + lrow) * (unsigned)K + (unsigned)lchunk * WEPL) * WEB;
  constexpr unsigned W_OUT_OF_RANGE = 0x10000000u;  // > any descriptor range (16 rows x K x 2 bytes), no 32-bit wrap when added
  auto issue_w = [&](WReg* w, int pass, int sl, int su, bool ok) {
    const int cb = c0 + (pass * MAXU + su / R) * CPT;  // first column of the tile
    const int nrow = min(c0 + cwb - cb, 16);  // valid rows of the tile (16 except in the block's last tile)
    if constexpr (PK) {
      // granule (cb + (su % R) N) / gr of the packed array (tile starts, block widths and the stacked half's row count are multiples
      // of gr; rows past the matrix inside its last granule are zeros in the array).  Lane (c = lane & 15, g = lane >> 4) reads row
      // c % gr of the tile's granule c / gr: 16 bytes per k block, the k blocks of a granule gr x 64 bytes apart
      const unsigned gstride = (unsigned)K * WEB << gr_shift;  // bytes of a granule
      const unsigned char* base = reinterpret_cast<const unsigned char*>(Wv) + ((size_t)cb + (size_t)(su % R) * N) * K * WEB;
      const int ngran = (nrow + (1 << gr_shift) - 1) >> gr_shift;
      const __amdgpu_buffer_rsrc_t rs =
          __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(base), 0, nrow > 0 ? ngran * (int)gstride : 0, 0x00020000);
      constexpr int KBLK = W8 ? 64 : 32;  // k per load
      const unsigned c = lane & 15, g = lane >> 4, kbs = 64u << gr_shift;  // bytes of a (granule, k block)
      const unsigned lane_off = (c >> gr_shift) * gstride + ((g << gr_shift) + (c & ((1u << gr_shift) - 1))) * 16u;
      const unsigned so = ok ? lane_off + (unsigned)(sl * (SK / KBLK)) * kbs : W_OUT_OF_RANGE;
#pragma unroll
      for (int j = 0; j < NWL; ++j) {
        // k blocks past K (last slice of a K that is no multiple of 256) leave the range
        const unsigned oj = sl * SK + j * KBLK < K ? so + j * kbs : W_OUT_OF_RANGE;
        w[j] = __builtin_bit_cast(WReg, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)oj, 0, 2));  // aux 2 = nt
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
      const unsigned char* base = reinterpret_cast<const unsigned char*>(Wv) + ((size_t)cb + (size_t)(su % R) * N) * K * WEB;
      const __amdgpu_buffer_rsrc_t rs =
          __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(base), 0, nrow > 0 ? nrow * K * WEB : 0, 0x00020000);
      // slice offset; lanes whose k lies past K (last slice of a K that is no multiple of 256) leave the range too
      const unsigned so = (ok ? (unsigned)(sl * SK) * WEB : W_OUT_OF_RANGE) + (sl * SK + lchunk * WEPL < K ? 0u : W_OUT_OF_RANGE);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if constexpr (W8) w[j] = __builtin_bit_cast(WReg, __builtin_amdgcn_raw_buffer_load_b64(rs, (int)(wrow_off[j] + so), 0, 2));
        else w[j] = __builtin_bit_cast(WReg, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(wrow_off[j] + so), 0, 2));  // aux 2 = nt
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };
  // 8 fp8 -> 8 bf16 (every e4m3 value is exactly representable in bf16)
  auto widen = [&](const u32x2& r) -> u32x4 {
    u32x4 o;
#if defined(SRGPT_SKINNY_PROBE) && SRGPT_SKINNY_PROBE == 1  // timing probe (WRONG results): what would the K loop cost without the fp8 -> bf16 conversions?
    o[0] = r[0]; o[1] = r[1]; o[2] = r[0] ^ 0x3c003c00u; o[3] = r[1] ^ 0x3c003c00u;
    return o;
#endif
#pragma unroll
    for (int h = 0; h < 2; ++h) {  // v_cvt_scalef32_pk_bf16_fp8 (gfx950): two fp8 -> packed bf16x2 in one instruction, scale 1
      o[2 * h] = __builtin_bit_cast(unsigned int, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(r[h], 1.0f, false));
      o[2 * h + 1] = __builtin_bit_cast(unsigned int, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(r[h], 1.0f, true));
    }
    return o;
  };
  // what a staged weight load leaves in LDS: bf16 (row-major layout)
  auto staged = [&](const WReg& r) -> u32x4 {
    if constexpr (W8 && !PK) return widen(r);
    else return r;
  };
  // packed layout: the B fragment of k step s of a stage, straight from the stage's registers
  auto pk_frag = [&](const WReg* w, int s) -> bf16x8 {
    if constexpr (!PK) return bf16x8{};
    else if constexpr (W8) return __builtin_bit_cast(bf16x8, widen(u32x2{w[s >> 1][2 * (s & 1)], w[s >> 1][2 * (s & 1) + 1]}));
    else return __builtin_bit_cast(bf16x8, w[s]);
  };

  // A wave's K slices are ONE contiguous range of the rows (round 3): interleaved over the waves (wave, wave + NW, ...: the first
  // version) every 512-byte piece of a weight row was fetched by a different wave at a different time -- the same DRAM pages
  // opened again and again; contiguous ranges stream 5 - 6 % faster at 4 / 8 bf16 rows, 2.5 % at 8 fp8 rows
  // (profiles/r03_skinny_contig.txt).  -DSRGPT_SKINNY_CONTIG=0 restores the interleaved split (tuning builds).
#ifndef SRGPT_SKINNY_CONTIG
#define SRGPT_SKINNY_CONTIG 1
#endif
#if SRGPT_SKINNY_CONTIG
  const int per_wave = (nsl + NW - 1) / NW, first_sl = wave * per_wave;
  const int cnt = max(0, min(per_wave, nsl - first_sl));  // slices of this wave: [first_sl, first_sl + cnt)
  auto sl_of = [&](int i) { return first_sl + i; };
#else
  const int cnt = wave < nsl ? (nsl - wave + NW - 1) / NW : 0;  // slices of this wave: wave, wave + NW, ...
  auto sl_of = [&](int i) { return wave + NW * i; };
#endif
  SK_STAMP(0);
  load_x(std::integral_constant<int, 0>{}, sl_of(0));
  bool x0_ready = true;  // slot 0 holds (a request for) the first slice of the pass about to start

  // measured per decode step (profiles/r02_skinny_ab.txt, section 6): 5-8 rows bf16 -1.8 %, 3-4 rows bf16 +-0, fp8 +0.8..1 % -> bf16 only;
  // 16 staged rows: the registers are not there
  constexpr bool PRE = (W8 ? SRGPT_SKINNY_PRE_W8 != 0 : SRGPT_SKINNY_PRE != 0) && NI == 4;

  // ---- RMSNorm statistics of every batch row (LlamaRMSNorm: fp32 mean of squares over K) ----
  // `between` runs once, after the statistics' first batch of loads has been issued and before it is reduced
  auto rms_stats = [&](auto&& between) {
    constexpr int HT = NT / 2;                      // threads per row parity
    constexpr int UF = (PRE && NI >= 4 ? 8 : 16) / NI;  // loads in flight per thread (8 when a weight stage is in flight too)
    const int half = tid / HT, kc = tid % HT;       // thread -> rows {2i + half}, chunks kc, kc + HT, ...
    float ss[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) ss[i] = 0.f;
    const int nch = K >> 3;
    for (int c0 = 0; c0 < nch; c0 += UF * HT) {
      u32x4 v[UF][NI];
#pragma unroll
      for (int u = 0; u < UF; ++u)
#pragma unroll
        for (int i = 0; i < NI; ++i) {
          const int b = min(2 * i + half, B - 1);
          const int c = min(c0 + u * HT + kc, nch - 1);
          v[u][i] = *reinterpret_cast<const u32x4*>(x + (size_t)b * K + (size_t)c * 8);
        }
      if (c0 == 0) {
        __builtin_amdgcn_sched_barrier(0);
        between();
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int u = 0; u < UF; ++u) {
        const bool ok = c0 + u * HT + kc < nch;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
          float s = 0.f;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float lo = bf16lo(v[u][i][q]), hi = bf16hi(v[u][i][q]);
            s += lo * lo + hi * hi;
          }
          ss[i] += ok ? s : 0.f;
        }
      }
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const float t = wave_sum(ss[i]);
      if (lane == 0) ss_s[2 * i + half][wave % (NW / 2)] = t;
    }
    __syncthreads();
    if (tid < 2 * NI) {
      float t = 0.f;
#pragma unroll
      for (int k = 0; k < NW / 2; ++k) t += ss_s[tid][k];
      rs_s[tid] = rsqrtf(t / (float)K + norm_eps);
    }
    __syncthreads();
  };

  // ---- the same statistics from the producer's published partial sums (PUB) ----
  // wave w owns rows w, w + NW, ...; `between` runs after the loads have been requested (the first weight stage queues behind them)
  auto pub_stats = [&](auto&& between) {
    constexpr int RPW = (2 * NI + NW - 1) / NW;
    f32x4 sv[RPW][2];
#pragma unroll
    for (int j = 0; j < RPW; ++j) {
      const float* p = ss_in + (size_t)min(wave + j * NW, B - 1) * SRGPT_ROWSS_STRIDE + lane * 8;
      sv[j][0] = *reinterpret_cast<const f32x4*>(p);
      sv[j][1] = *reinterpret_cast<const f32x4*>(p + 4);
    }
    __builtin_amdgcn_sched_barrier(0);
    between();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < RPW; ++j) {
      float t = ((sv[j][0][0] + sv[j][0][1]) + (sv[j][0][2] + sv[j][0][3])) + ((sv[j][1][0] + sv[j][1][1]) + (sv[j][1][2] + sv[j][1][3]));
      t = wave_sum(t);
      if (lane == 0 && wave + j * NW < 2 * NI) rs_s[wave + j * NW] = rsqrtf(t / (float)K + norm_eps);
    }
    __syncthreads();
  };
  float pub_acc = 0.f;  // ss_out: sum of squares of this thread's output elements (all of one batch row: see the epilogue)

  // one pass over K for NSU sub-units
  auto run_pass = [&](auto nsu_c, int pass, int nu) {
    constexpr int NSU = decltype(nsu_c)::value;
    constexpr int NS = SK / 32;           // MFMA k steps per stage
    constexpr int FS = NI == 8 ? 2 : SRGPT_SKINNY_FS;  // k steps whose fragments are read together (16 staged rows: fewer, registers)
#ifndef SRGPT_SKINNY_ONE_ACC
#define SRGPT_SKINNY_ONE_ACC 0
#endif
    constexpr bool TWO_ACC = NI < 8 && !SRGPT_SKINNY_ONE_ACC;  // even / odd k steps on separate accumulators (16 staged rows: the registers are not there)
    // slots of the activation ring (see load_x): as many slices ahead as the weight ring reaches
    // (fp8 only: with bf16 weights a stage is twice the registers and the 8-row single-tile kernel spills under a second slot --
    //  o / down 9.2 / 22.6 -> 9.9 / 24.4 us at 8 bf16 rows, profiles/r06_skinny_norm_staging.txt)
    constexpr int XS = !W8 ? 1 : NSU == 1 ? DEPTH : (NSU == 2 && DEPTH == 4 ? 2 : 1);
    static_assert(DEPTH % XS == 0, "the slot of a slice must be a compile-time function of its place in the trip");
    if (!x0_ready) load_x(std::integral_constant<int, 0>{}, sl_of(0));
    if constexpr (XS > 1) load_x(std::integral_constant<int, 1>{}, sl_of(min(1, max(cnt - 1, 0))));
    if constexpr (XS > 2) load_x(std::integral_constant<int, 2>{}, sl_of(min(2, max(cnt - 1, 0))));
    if constexpr (XS > 3) load_x(std::integral_constant<int, 3>{}, sl_of(min(3, max(cnt - 1, 0))));
    x0_ready = XS == 1;  // a one-slot pass leaves the next pass's first slice requested
    f32x4 acc[NSU], acc2[TWO_ACC ? NSU : 1];
#pragma unroll
    for (int su = 0; su < NSU; ++su) acc[su] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int su = 0; su < (TWO_ACC ? NSU : 1); ++su) acc2[su] = f32x4{0.f, 0.f, 0.f, 0.f};
    // register ring of DEPTH weight stages with static slot indices (a trip = DEPTH slices x NSU stages, a multiple of
    // DEPTH): the loads of stage t+DEPTH-1 are issued before stage t is multiplied.  A stage is 8 KiB per wave (fp8: 4 KiB).
    WReg wb[DEPTH][NWL];
    // epilogue operands that do not depend on the products -- row scales (W8) and residual elements -- are requested up front
    // (fetched at their use they would each add a memory round trip after the last barrier of the pass), but BEHIND the first
    // weight stages and without touching their values: round 4's form, `residual ? (float)residual[i] : 0` in front of the first
    // weight request, compiled to branch -> load -> s_waitcnt vmcnt(0) -> convert per element -- o_proj and down_proj went
    // through up to four serialised memory round trips (2.7k cycles, profiles/r05_skinny_stamps.txt) before asking for a weight
    constexpr int EIT0 = ((NSU / R) * 256 + NT - 1) / NT;  // epilogue iterations per thread
    constexpr int EIT = EIT0 > 0 ? EIT0 : 1;               // (odd NSU never occurs with SwiGLU; keep the type valid)
    float pre_sc[EIT][RS];
    unsigned short pre_res[EIT];  // bf16 bits
    auto load_epilogue_operands = [&]() {
      const bool has_res = !SWIGLU && residual != nullptr;
      const bf16_t* rp = has_res ? residual : x;
#pragma unroll
      for (int i = 0; i < EIT; ++i) {
        const int e = tid + i * NT;
        const int u = e >> 8, l2 = e & 63, q = (e >> 6) & 3;
        const int b = min(4 * (l2 >> 4) + q, B - 1);
        const int n = min(c0 + (pass * MAXU + u) * CPT + (l2 & (CPT - 1)), c0 + cwb - 1);
#pragma unroll
        for (int r = 0; r < RS; ++r) pre_sc[i][r] = W8 ? wscale[n + r * N] : 1.f;
        pre_res[i] = *reinterpret_cast<const unsigned short*>(rp + (has_res ? (size_t)b * N + n : (size_t)0));
      }
    };
    // pass 0: the first weight stage goes out behind the statistics' own loads (in-order return: the reduction waits for its
    // activations only), so its HBM latency overlaps the reduction and the two block barriers -- as in the GEMV's prologue
    if (PUB && pass == 0) {
      pub_stats([&]() { issue_w(wb[0], 0, sl_of(0), 0, cnt > 0); });
      load_rs();
    } else if (PRE && pass == 0 && do_norm) {
      rms_stats([&]() { issue_w(wb[0], 0, sl_of(0), 0, cnt > 0); });
      load_rs();
    } else {
      issue_w(wb[0], pass, sl_of(0), 0, cnt > 0);
    }
#pragma unroll
    for (int f = 1; f < DEPTH - 1; ++f)
      issue_w(wb[f % DEPTH], pass, sl_of(f / NSU), f % NSU, f / NSU < cnt);
    __builtin_amdgcn_sched_barrier(0);
    load_epilogue_operands();
    __builtin_amdgcn_sched_barrier(0);
    if (pass == 0) SK_STAMP(1);
    // one slice (h-th of the trip that starts at slice index i): NSU stages
    // activation fragments of a slice held across its sub-units where the registers are there (fp8 weights: the stage ring is
    // half as wide; bf16 weights: the 4-row variant only)
    constexpr bool XREUSE = SRGPT_SKINNY_XREUSE != 0 && NSU >= 2 && (W8 ? NI <= 4 : NI == 2);
    auto slice = [&](int i, auto h_c) {
      constexpr int h = decltype(h_c)::value;
      const int sl = sl_of(i + h);
      bf16x8 xall[XREUSE ? NS : 1];
      {
#pragma unroll
          for (int su = 0; su < NSU; ++su) {
            const int cur = (h * NSU + su) % DEPTH;
            const int fn = h * NSU + su + DEPTH - 1;  // stage to prefetch, relative to this trip
            issue_w(wb[fn % DEPTH], pass, sl_of(i + fn / NSU), fn % NSU, i + fn / NSU < cnt);
            if (su == 0) {
              stage_x(std::integral_constant<int, h % XS>{}, sl, true);
              if constexpr (XS == 1) load_x(std::integral_constant<int, 0>{}, i + h + 1 < cnt ? sl_of(i + h + 1) : sl_of(0));  // next slice, or the first one of the next pass
              else load_x(std::integral_constant<int, h % XS>{}, sl_of(min(i + h + XS, cnt - 1)));  // XS slices ahead (past the end: a valid slice, never staged)
            }
#if !(defined(SRGPT_SKINNY_PROBE) && SRGPT_SKINNY_PROBE == 2)
            if constexpr (!PK) {
#pragma unroll
              for (int j = 0; j < 8; ++j) *reinterpret_cast<u32x4*>(wst + (2 * j + lrow) * WROWB + lchunk * 16) = staged(wb[cur][j]);
            }
#endif
            __builtin_amdgcn_wave_barrier();
            // The fragment reads of FS k steps are issued together in front of their MFMAs, which alternate between two
            // accumulators.  Left to the compiler the unrolled loop became read -> wait -> MFMA per k step on ONE accumulator: a
            // full LDS latency per step, ~1.5k cycles per stage and sub-unit -- the kernel was bound by that, not by HBM.
            const int xrow = lane & (2 * NI - 1);  // rows past the staged ones alias valid rows: their outputs are never stored
            if constexpr (XREUSE) {
              // the activation fragments of the slice are the same for all of its sub-units: read once, kept in registers
              if (su == 0) {
#pragma unroll
                for (int s = 0; s < NS; ++s) xall[s] = *reinterpret_cast<const bf16x8*>(xst + xrow * XROWB + (4 * s + (lane >> 4)) * 16);
              }
            }
#pragma unroll
            for (int s0 = 0; s0 < NS; s0 += FS) {
              bf16x8 xf[FS];
              bf16x8 wfr[FS];
#pragma unroll
              for (int t = 0; t < FS; ++t) {
                const int s = s0 + t;
                if constexpr (XREUSE) xf[t] = xall[s];
                else xf[t] = *reinterpret_cast<const bf16x8*>(xst + xrow * XROWB + (4 * s + (lane >> 4)) * 16);
#if defined(SRGPT_SKINNY_PROBE) && SRGPT_SKINNY_PROBE == 2  // timing probe (WRONG results): weights never pass through LDS
                wfr[t] = __builtin_bit_cast(bf16x8, staged(wb[cur][s & 7]));
#else
                if constexpr (PK) wfr[t] = pk_frag(wb[cur], s);
                else wfr[t] = *reinterpret_cast<const bf16x8*>(wst + (lane & 15) * WROWB + (4 * s + (lane >> 4)) * 16);
#endif
              }
              __builtin_amdgcn_sched_barrier(0);
#pragma unroll
              for (int t = 0; t < FS; ++t) {
                const bf16x8 wf = wfr[t];
                // D[batch row][weight row] += x[batch row][k] * W[weight row][k]
#if defined(SRGPT_SKINNY_PROBE) && SRGPT_SKINNY_PROBE == 3  // timing probe (WRONG results): no matrix instructions
                acc[su][0] += (float)xf[t][0] * (float)wf[0];
                acc[su][1] += (float)xf[t][7] * (float)wf[7];
#else
                if (TWO_ACC && (t & 1)) acc2[TWO_ACC ? su : 0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[t], wf, acc2[TWO_ACC ? su : 0], 0, 0, 0);
                else acc[su] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[t], wf, acc[su], 0, 0, 0);
#endif
              }
              __builtin_amdgcn_sched_barrier(0);
            }
            __builtin_amdgcn_wave_barrier();
          }
      }
    };
    // full trips run branch-free; the last cnt % DEPTH slices are guarded (nothing is left to prefetch behind them, so the
    // conservative waits the guard brings cost nothing)
    int i = 0;
    for (; i + DEPTH <= cnt; i += DEPTH) {
      slice(i, std::integral_constant<int, 0>{});
      if (pass == 0 && i == 0) SK_STAMP(2);
      if constexpr (DEPTH > 1) slice(i, std::integral_constant<int, 1>{});
      if constexpr (DEPTH > 2) slice(i, std::integral_constant<int, 2>{});
      if constexpr (DEPTH > 3) slice(i, std::integral_constant<int, 3>{});
    }
    if (i < cnt) slice(i, std::integral_constant<int, 0>{});
    if constexpr (DEPTH > 2)
      if (i + 1 < cnt) slice(i, std::integral_constant<int, 1>{});
    if constexpr (DEPTH > 3)
      if (i + 2 < cnt) slice(i, std::integral_constant<int, 2>{});
    static_assert(DEPTH >= 2 && DEPTH <= 4, "ring depth");

    if (pass == 0) SK_STAMP(3);
    // ---- cross-wave reduction (fixed order) + epilogue ----
    __syncthreads();
    if (pass == 0) SK_STAMP(4);
    float* redf = reinterpret_cast<float*>(smem);  // [NW waves][MAXSU][64 lanes][4]
#pragma unroll
    for (int su = 0; su < NSU; ++su) *reinterpret_cast<f32x4*>(redf + ((wave * MAXSU + su) * 64 + lane) * 4) = TWO_ACC ? acc[su] + acc2[TWO_ACC ? su : 0] : acc[su];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < EIT; ++i) {
      const int e = tid + i * NT;
      if (e >= (NSU / R) * 256) break;
      const int u = e >> 8, l2 = e & 63, q = (e >> 6) & 3;
      const int b = 4 * (l2 >> 4) + q;
      const int n = c0 + (pass * MAXU + u) * CPT + (l2 & 15);
      float a[RS];
#pragma unroll
      for (int r = 0; r < RS; ++r) {
        float t = 0.f;
#pragma unroll
        for (int wv = 0; wv < NW; ++wv) t += redf[((wv * MAXSU + u * R + r) * 64 + l2) * 4 + q];
        if (W8) t *= pre_sc[i][r];
        a[r] = t;
      }
      if (b < B && n < c0 + cwb) {
        if (SWIGLU) {
          const float g = rnd<bf16_t>(a[0]), up = rnd<bf16_t>(a[RS - 1]);
          reinterpret_cast<bf16_t*>(out)[(size_t)b * N + n] = (bf16_t)(rnd<bf16_t>(silu(g)) * up);
        } else {
          float v = rnd<bf16_t>(a[0]);
          if (residual) v = rnd<bf16_t>(__uint_as_float((unsigned int)pre_res[i] << 16) + v);
          pub_acc = fmaf(v, v, pub_acc);
          if (out_f32)
            reinterpret_cast<float*>(out)[(size_t)b * N + n] = v;
          else
            reinterpret_cast<bf16_t*>(out)[(size_t)b * N + n] = (bf16_t)v;
        }
      }
    }
    if (pass == 0) SK_STAMP(5);
    __syncthreads();  // the reduction buffer aliases the wave-private stages of the next pass
  };

  if (!PUB && !PRE && do_norm) {
    rms_stats([]() {});
    load_rs();
  }
  for (int pass = 0; pass < npass; ++pass) {
    int nu = 0;
#pragma unroll
    for (int u = 0; u < MAXU; ++u) nu += (pass * MAXU + u < ntile) ? 1 : 0;
    if (nu == 0) break;  // uniform per block; later passes are empty too
#ifdef SRGPT_SKINNY_FIX_NSU  // register / occupancy probe (tuning builds): ONE sub-unit count compiled in -- only shapes with that count compute correctly
    run_pass(std::integral_constant<int, SRGPT_SKINNY_FIX_NSU>{}, pass, nu);
    continue;
#endif
    switch (nu * R) {
      case 1: run_pass(std::integral_constant<int, 1>{}, pass, nu); break;
      case 2: run_pass(std::integral_constant<int, 2>{}, pass, nu); break;
      case 3: run_pass(std::integral_constant<int, 3>{}, pass, nu); break;
      default: run_pass(std::integral_constant<int, 4>{}, pass, nu); break;
    }
  }
  if constexpr (!SWIGLU) {
    if (ss_out != nullptr) {
      // epilogue element e = tid + i NT sits in batch row 4 (lane >> 4) + (wave & 3) whatever i and the pass are: a thread's pub_acc
      // belongs to ONE row, and the 16 lanes of a DPP row hold that row's 16 columns of every tile; waves w and w + 4 (8-wave blocks)
      // hold different tiles of the same rows: added in wave order through LDS
      float tot = lanes_sum<16>(pub_acc);
      if constexpr (NW > 4) {
        if ((lane & 15) == 0) pub_s[wave][lane >> 4] = tot;
        __syncthreads();
        if (wave < 4) {
          tot = 0.f;
#pragma unroll
          for (int wv = 0; wv < NW; wv += 4) tot += pub_s[wave + wv][lane >> 4];
        }
      }
      const int b = 4 * (lane >> 4) + wave;
      if (wave < 4 && (lane & 15) == 0 && b < B) {
        float* row = ss_out + (size_t)b * SRGPT_ROWSS_STRIDE;
        row[blockIdx.x] = tot;
        for (int sidx = (int)blockIdx.x + (int)gridDim.x; sidx < SRGPT_ROWSS_STRIDE; sidx += (int)gridDim.x) row[sidx] = 0.f;
      }
    }
  }
}

template <bool SWIGLU, int NI, int NW, bool W8, bool PUB, bool PK>
int launch_skinny(const void* x, const void* W, const float* wscale, const void* norm_w, float eps, const void* residual,
                  void* out, int batch, int N, int K, int out_f32, int grid, int cw, const float* ss_in, float* ss_out,
                  int gr_shift, hipStream_t s) {
  constexpr int lds_k = NW * (2 * NI * WROWB + (PK ? 0 : WSTAGEB)), lds_r = NW * MAXSU * 64 * 4 * 4;  // K loop stages | reduction buffer
  constexpr int lds = lds_k > lds_r ? lds_k : lds_r;
  static_assert(lds <= 160 * 1024, "LDS");
  static_assert(NW == 8 || 2 * lds <= 160 * 1024, "two 4-wave blocks per CU");
  auto kfn = skinny_kernel<SWIGLU, NI, NW, W8, PUB, PK>;
  static std::atomic<uint64_t> attr_done{0};
  SRGPT_TRY(srgpt_ensure_dyn_lds(attr_done, (const void*)kfn, lds));
  hipLaunchKernelGGL(kfn, dim3(grid), dim3(64 * NW), lds, s, (const bf16_t*)x, W, wscale, (const bf16_t*)norm_w, eps,
                     (const bf16_t*)residual, out, batch, N, K, out_f32, cw, ss_in, ss_out, gr_shift);
  SRGPT_LAUNCH_CHECK();
  return SRGPT_OK;
}

template <bool SWIGLU, int NI, bool W8>
int launch_skinny_nw(const void* x, const void* W, const float* wscale, const void* norm_w, float eps, const void* residual,
                     void* out, int batch, int N, int K, int out_f32, const float* ss_in, float* ss_out, int packed, hipStream_t s) {
  // Two 4-wave blocks per CU, or one 8-wave block per CU; both split the columns evenly over their blocks.  Measured per decode
  // step (profiles/r02_skinny_ab.txt): 4-wave blocks win (o_proj 8.8 vs 9.7 us, fp8 gate/up 27.4 vs 30.4) except where a block
  // would own few columns AND has the RMSNorm statistics to compute first (q/k/v: 24 columns per CU, 14.1 vs 14.3 us bf16,
  // 12.3 vs 13.7 fp8) -- there the prologue is shared by twice the threads.
  const int cus = srgpt_device_cus();
  int waves = SRGPT_KNOB("SRGPT_SKINNY_WAVES", 0);
  if (waves != 4 && waves != 8) {
    const int ncol = (N + cus - 1) / cus;  // output columns per CU
    waves = (norm_w != nullptr && ncol <= 32) ? 8 : 4;
    if (norm_w == nullptr && ncol <= SRGPT_KNOB("SRGPT_SKINNY_W8_RES_COLS", 0)) waves = 8;
  }
  const int blocks = waves == 8 ? cus : SRGPT_KNOB("SRGPT_SKINNY_BPC", 2) * cus;
  int cw = (N + blocks - 1) / blocks;
  if (cw < 16) cw = 16;
  const int gr_shift = packed == 16 ? 4 : packed == 8 ? 3 : 2;
  if (packed) cw = ((cw + packed - 1) / packed) * packed;  // whole granules
  const int grid = (N + cw - 1) / cw;
  SRGPT_CHECK(!ss_out || grid <= SRGPT_ROWSS_STRIDE, SRGPT_ERR_UNSUPPORTED, "skinny: %d blocks do not fit the %d row-statistics slots",
              grid, SRGPT_ROWSS_STRIDE);
#define SRGPT_SKINNY_GO(NWV, PUBV, PKV) \
  return launch_skinny<SWIGLU, NI, NWV, W8, PUBV, PKV>(x, W, wscale, norm_w, eps, residual, out, batch, N, K, out_f32, grid, cw, ss_in, ss_out, gr_shift, s)
  if (packed) {
    SRGPT_CHECK(K % (W8 ? 64 : 32) == 0 && (!SWIGLU || N % packed == 0), SRGPT_ERR_UNSUPPORTED,
                "skinny: the packed weight layout needs K %% %d == 0 (K = %d)%s", W8 ? 64 : 32, K, SWIGLU ? " and whole granules per half" : "");
    if (ss_in != nullptr) {
      if (waves == 8) SRGPT_SKINNY_GO(8, true, true);
      SRGPT_SKINNY_GO(4, true, true);
    }
    if (waves == 8) SRGPT_SKINNY_GO(8, false, true);
    SRGPT_SKINNY_GO(4, false, true);
  }
  if (ss_in != nullptr) {
    if (waves == 8) SRGPT_SKINNY_GO(8, true, false);
    SRGPT_SKINNY_GO(4, true, false);
  }
  if (waves == 8) SRGPT_SKINNY_GO(8, false, false);
  SRGPT_SKINNY_GO(4, false, false);
#undef SRGPT_SKINNY_GO
}

template <bool W8>
int skinny_dispatch(const void* x, const void* W, const float* wscale, const void* norm_w, float eps, const void* residual,
                    void* out, int batch, int N, int K, int swiglu, int out_f32, const float* ss_in, float* ss_out, int packed,
                    hipStream_t s) {
  // packed: 0 = row-major W [N][K]; 4 / 16 = the packed decode layout with granules of that many rows (srgpt_pack_decode_weights)
  if (SRGPT_KNOB("SRGPT_SKINNY_PACKED_TIMING", 0)) {  // tuning build: fetch-pattern timing on row-major data (WRONG results)
    const int ncol = (N + srgpt_device_cus() - 1) / srgpt_device_cus();
    packed = ncol <= 16 ? SRGPT_KNOB("SRGPT_SKINNY_PACKED_T1", 16) : SRGPT_KNOB("SRGPT_SKINNY_PACKED_TN", 4);
  }
  SRGPT_CHECK(packed == 0 || packed == 4 || packed == 8 || packed == 16, SRGPT_ERR_ARG, "skinny: packed layout granule %d (0, 4, 8 or 16)", packed);
  SRGPT_CHECK(batch >= 1 && batch <= 16, SRGPT_ERR_ARG, "skinny: batch %d outside 1..16", batch);
  SRGPT_CHECK(K % 8 == 0 && K >= 8, SRGPT_ERR_ARG, "skinny: K=%d must be a multiple of 8", K);
  SRGPT_CHECK(!ss_in || norm_w, SRGPT_ERR_ARG, "skinny: published row statistics are the RMSNorm's input (norm_w is NULL)");
  SRGPT_CHECK(!ss_out || (!swiglu && !out_f32), SRGPT_ERR_ARG, "skinny: row statistics are published for plain bf16 outputs only");
  if (batch <= 4)
    return swiglu ? launch_skinny_nw<true, 2, W8>(x, W, wscale, norm_w, eps, residual, out, batch, N, K, out_f32, ss_in, nullptr, packed, s)
                  : launch_skinny_nw<false, 2, W8>(x, W, wscale, norm_w, eps, residual, out, batch, N, K, out_f32, ss_in, ss_out, packed, s);
  if (batch <= 8)
    return swiglu ? launch_skinny_nw<true, 4, W8>(x, W, wscale, norm_w, eps, residual, out, batch, N, K, out_f32, ss_in, nullptr, packed, s)
                  : launch_skinny_nw<false, 4, W8>(x, W, wscale, norm_w, eps, residual, out, batch, N, K, out_f32, ss_in, ss_out, packed, s);
  return swiglu ? launch_skinny_nw<true, 8, W8>(x, W, wscale, norm_w, eps, residual, out, batch, N, K, out_f32, ss_in, nullptr, packed, s)
                : launch_skinny_nw<false, 8, W8>(x, W, wscale, norm_w, eps, residual, out, batch, N, K, out_f32, ss_in, ss_out, packed, s);
}

}  // namespace

#ifdef SRGPT_TUNING_KNOBS
extern "C" int srgpt_skinny_debug_stamps(unsigned long long* host, int n) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(srgpt_skinny_stamps), sizeof(unsigned long long) * (n < 16 ? n : 16));
}
#endif

// host entry used by srgpt_gemv (gemv.hip) for batches of up to 16 rows, bf16 weights
int srgpt_skinny_launch(const void* x, const void* W, const void* norm_w, float eps, const void* residual, void* out,
                        int batch, int N, int K, int swiglu, int out_f32, const float* ss_in, float* ss_out, int packed, hipStream_t s) {
  return skinny_dispatch<false>(x, W, nullptr, norm_w, eps, residual, out, batch, N, K, swiglu, out_f32, ss_in, ss_out, packed, s);
}

// fp8 weights, any batch size (16 rows per weight pass); ss_in / ss_out: the rows' statistics tables (see skinny_kernel)
int srgpt_skinny_w8_launch(const void* x, const void* W8, const float* wscale, const void* norm_w, float norm_eps,
                           const void* residual, void* out, int batch, int N, int K, int swiglu, int out_f32,
                           const float* ss_in, float* ss_out, int packed, hipStream_t s) {
  const size_t on = out_f32 ? sizeof(float) : 2;
  for (int b0 = 0; b0 < batch; b0 += 16) {
    const int nb = batch - b0 < 16 ? batch - b0 : 16;
    SRGPT_TRY(skinny_dispatch<true>((const char*)x + (size_t)b0 * K * 2, W8, wscale, norm_w, norm_eps,
                                    residual ? (const char*)residual + (size_t)b0 * N * 2 : nullptr,
                                    (char*)out + (size_t)b0 * N * on, nb, N, K, swiglu, out_f32,
                                    ss_in ? ss_in + (size_t)b0 * SRGPT_ROWSS_STRIDE : nullptr,
                                    ss_out ? ss_out + (size_t)b0 * SRGPT_ROWSS_STRIDE : nullptr, packed, s));
  }
  return SRGPT_OK;
}

// rows at or below this count take the one-row VALU kernel of gemv_w8.hip (which neither reads nor publishes row statistics)
int srgpt_w8_valu_max_batch() { return SRGPT_KNOB("SRGPT_W8_VALU_MAX_BATCH", 1); }  // 2 rows: the MFMA kernel is 7 % faster per step (round 3)

// Decode-path product with fp8 (OCP e4m3fn) weights and one fp32 scale per weight row, bf16 activations (W8A16):
// out[b, n] = bf16( (sum_k x[b, k] * fp8(W8[n, k])) * wscale[n] ), same fusions as srgpt_gemv.  Any batch size
// (16 rows per weight pass).
extern "C" int srgpt_gemv_w8(const void* x, const void* W8, const float* wscale, const void* norm_w, float norm_eps,
                             const void* residual, void* out, int batch, int N, int K, int swiglu, int out_f32,
                             srgpt_stream_t stream) {
  SRGPT_CHECK(x && W8 && wscale && out, SRGPT_ERR_ARG, "srgpt_gemv_w8: null pointer");
  SRGPT_CHECK(N > 0 && K > 0 && batch > 0, SRGPT_ERR_ARG, "srgpt_gemv_w8: bad shape");
  SRGPT_CHECK(K % 8 == 0, SRGPT_ERR_ARG, "srgpt_gemv_w8: K=%d must be a multiple of 8", K);
  SRGPT_CHECK(!(swiglu && (residual || out_f32)), SRGPT_ERR_ARG, "srgpt_gemv_w8: swiglu excludes residual/out_f32");
  hipStream_t s = as_stream(stream);
  const int valu_max = srgpt_w8_valu_max_batch();
  if (batch <= valu_max && batch <= 2 && K % 16 == 0)  // one row: VALU kernel (gemv_w8.hip), like the bf16 path
    return srgpt_gemv_w8_valu(x, W8, wscale, norm_w, norm_eps, residual, out, batch, N, K, swiglu, out_f32, s);
  return srgpt_skinny_w8_launch(x, W8, wscale, norm_w, norm_eps, residual, out, batch, N, K, swiglu, out_f32, nullptr, nullptr, 0, s);
}

// ---- the packed decode layout (include/srgpt.h, ABI 9) ----
namespace {
// one thread per 16 output bytes: (granule, k block, g, row in granule)
__global__ __launch_bounds__(256) void pack_decode_kernel(const unsigned char* __restrict__ W, unsigned char* __restrict__ out, int N, int K,
                                                          int eb, int rows, long long n16) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n16) return;
  const int kblock = 32 * (2 / eb);           // k per 16-byte lane load: 64 (fp8) / 32 (bf16)
  const int nkb = K / kblock;
  const int r = (int)(i % rows), g = (int)(i / rows % 4);
  const long long t = i / (4 * rows);
  const int kb = (int)(t % nkb);
  const long long n = t / nkb * rows + r;
  u32x4 v = {0u, 0u, 0u, 0u};
  if (n < N) {
    const unsigned char* src = W + ((size_t)n * K + (size_t)kb * kblock + 8 * g) * eb;
    if (eb == 2) {
      v = *reinterpret_cast<const u32x4*>(src);                                    // 8 bf16: k = 32 kb + 8 g .. + 8
    } else {
      const u32x2 lo = *reinterpret_cast<const u32x2*>(src), hi = *reinterpret_cast<const u32x2*>(src + 32);  // k = 64 kb (+ 32) + 8 g .. + 8
      v = u32x4{lo[0], lo[1], hi[0], hi[1]};
    }
  }
  *reinterpret_cast<u32x4*>(out + (size_t)i * 16) = v;
}
}  // namespace

extern "C" size_t srgpt_packed_bytes(int N, int K, int elem_bytes, int rows) {
  if (N <= 0 || K <= 0 || rows <= 0 || (elem_bytes != 1 && elem_bytes != 2)) return 0;
  return (size_t)((N + rows - 1) / rows) * rows * K * elem_bytes;
}

extern "C" int srgpt_pack_decode_weights(const void* W, void* out, int N, int K, int elem_bytes, int rows, srgpt_stream_t stream) {
  SRGPT_CHECK(W && out, SRGPT_ERR_ARG, "srgpt_pack_decode_weights: null pointer");
  SRGPT_CHECK(elem_bytes == 1 || elem_bytes == 2, SRGPT_ERR_ARG, "srgpt_pack_decode_weights: %d bytes per element (1: fp8, 2: bf16)", elem_bytes);
  SRGPT_CHECK(rows == 4 || rows == 8 || rows == 16, SRGPT_ERR_ARG, "srgpt_pack_decode_weights: %d rows per granule (4, 8 or 16)", rows);
  SRGPT_CHECK(N > 0 && K > 0 && K % (elem_bytes == 1 ? 64 : 32) == 0, SRGPT_ERR_ARG,
              "srgpt_pack_decode_weights: K=%d must be a positive multiple of %d", K, elem_bytes == 1 ? 64 : 32);
  const long long n16 = (long long)(srgpt_packed_bytes(N, K, elem_bytes, rows) / 16);
  hipLaunchKernelGGL(pack_decode_kernel, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, as_stream(stream), (const unsigned char*)W,
                     (unsigned char*)out, N, K, elem_bytes, rows, n16);
  SRGPT_LAUNCH_CHECK();
  return SRGPT_OK;
}
