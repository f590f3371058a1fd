// Wide backward chain of the activation-stationary SIREN kernels (bf16 MFMA, fp32 accumulate): k_fused_bwd of fused.hip
// for workgroups of up to 112 samples.  Reference arithmetic: models/siren.py:26-36 (SineLayer.forward) differentiated.
//
// k_fused_bwd's time is one weight stream per workgroup (2.6 MB through L2, independent of the workgroup's sample count)
// plus the epilogues of its samples; with 64-sample workgroups a cfg-D batch (25 088 rows) fills the chip 1.5 times, i.e.
// every CU pays two streams.  The backward needs no latent rows in LDS, so its g tile can hold 112 samples -- what stops
// the 64-feature x 112-sample accumulator block of a wave is the register file (112 + 64 ring registers + the epilogue's
// temporaries: 90-225 spills in fused_nodes.hip's 7-tile variants, whose scratch traffic drains the ring every step).
// Here a wave takes its 64 output features in TWO passes of 32 (one fragment pair each; feature map: wide_bwd_body):
// 56 accumulator registers, the ring holds eight steps of two fragments (the same 16 KB in flight per wave), and the
// registers this frees carry (a) the first pass's dz chunks until the g tile may be overwritten and (b) the saved rows of
// the next two sample tiles, requested before the current tile's stores are issued: gfx9 retires loads and stores through
// ONE in-order counter, so a wait for a tile's rows used to be a wait for every older store as well (DESIGN.md section 7,
// round 2).  The first pass's epilogue runs before the layer's barrier, i.e. beside the other waves' MFMA phases.
// Column sums (bias and first-layer weight gradients) go through the replicas of nodegrad.h when the caller has them;
// the first layer's d coords / dW0 are a tail phase on the dz_0 tile in LDS.
#include "fused.h"
#include "gemm_engine.h"
#include "nodegrad.h"
#include <algorithm>
#include <cstdlib>

namespace {

#include "fused_dev.h"
#include "dw_device.h"

#include "wide_dev.h"

#define WIDE_LA_MOD 2             // sample tiles whose saved rows (c, z, s) are requested ahead of the one being processed
#define WIDE_LA_PLAIN 4           // the same for unmodulated layers (c only)

// optional phase trace of one wave (tools/trace_fused.py MODE=wide): 100 MHz timestamps
__device__ long long g_wide_trace[128];
#ifdef FUSED_TRACE_BUILD   // perturbs register allocation: only in the libcinema_hip_trace.so developer build
#define WIDE_TRACE(ID) \
  if (a.trace_block == (int)blockIdx.x && (int)threadIdx.x == a.trace_wave * 64) g_wide_trace[ID] = (long long)wall_clock64();
#else
#define WIDE_TRACE(ID)
#endif

// (non-temporal stores, __builtin_nontemporal_store, measured here: 240 us against 235)
#define WIDE_STORE(P, V) *reinterpret_cast<bf16x8 *>(P) = V

struct WideBwdArgs {
  FinalBwdArgs fin; bf16_t *dout16; int g_chain, g_fin;   // tail workgroups: dOut -> [Npad x 16] bf16 + final bias gradient
  int g_big, nst_big;                // workgroups [0, g_big) own nst_big sample tiles of 16, the rest nst_big - 1
  const bf16_t *stream; long slab_elems;
  const float *side_w0;              // [HP][4] first-layer weights, then [16][HP] final_linear.weight[:, :H]
  const float *dout; const float *coords_t;
  const bf16_t *saved; long slot; int sv_c[CINEMA_MAX_LAYERS], sv_z[CINEMA_MAX_LAYERS], sv_s[CINEMA_MAX_LAYERS];
  bf16_t *gbuf; int gb_z[CINEMA_MAX_LAYERS], gb_s[CINEMA_MAX_LAYERS], gb_t[CINEMA_MAX_LAYERS];
  float *gp; long off_b[CINEMA_MAX_LAYERS], off_bl[CINEMA_MAX_LAYERS], off_w0; int wgrads;
  float *dcoords;
  unsigned mod_mask; int n_layers, in_dim, H, HP, out_dim, xrow, xsteps;
  float om_first, om_hidden;
  int trace_block, trace_wave;
  float *rep; int rep_w;             // column-sum replicas (nodegrad.h) or NULL: atomics straight into gp
};

// Epilogue of one pass (fragment pair FP: features fb + 64 FP .. + 7 of this lane, 64 contiguous bytes per row and lane
// group, the other half of the 128-byte line belongs to the neighbouring wave): dz = omega * dy * c (* s),
// ds = omega * dy * c * z, dt = dy * c.  FP == 0 leaves its dz chunks in `hold` (the g tile is still being read); FP == 1
// runs behind the layer's barrier and writes both passes' chunks.
template <bool MOD, int FP, int NST, bool LATB>
__device__ __forceinline__ void wide_epilogue(const WideBwdArgs &a, f32x4 (&acc)[2][NST], bf16x8 (&hold)[NST], int l,
                                              float omega, long n0, unsigned toff_l, unsigned tstep, unsigned fb_l, int i_l,
                                              int lane, lds_char *gs) {
  const char *C = (const char *)(a.saved + a.slot * a.sv_c[l] + n0 * a.HP) + 128 * FP;
  const char *Z = MOD ? (const char *)(a.saved + a.slot * a.sv_z[l] + n0 * a.HP) + 128 * FP : C;
  const char *S = MOD ? (const char *)(a.saved + a.slot * a.sv_s[l] + n0 * a.HP) + 128 * FP : C;
  char *DZ = (char *)(a.gbuf + a.slot * a.gb_z[l] + n0 * a.HP) + 128 * FP;
  char *DS = MOD ? (char *)(a.gbuf + a.slot * a.gb_s[l] + n0 * a.HP) + 128 * FP : DZ;
  char *DT = MOD ? (char *)(a.gbuf + a.slot * a.gb_t[l] + n0 * a.HP) + 128 * FP : DZ;
  const int c0 = (fb_l >> 3) + 8 * FP;
  float ts = 0.f, tt = 0.f;
  float tzv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // per-lane column sums of dz: ONE transpose-reduce per pass
  // saved rows of the first WIDE_LA tiles, then the drain (one wait for them and the ring's resident steps together)
  constexpr int WIDE_LA = MOD ? WIDE_LA_MOD : WIDE_LA_PLAIN;
  bf16x8 cP[NST], zP[NST], sP[NST];
#pragma unroll
  for (int q = 0; q < WIDE_LA && q < NST; ++q) {
    const unsigned o = toff_l + q * tstep;
    cP[q] = *reinterpret_cast<const bf16x8 *>(C + o);
    if constexpr (MOD) { zP[q] = *reinterpret_cast<const bf16x8 *>(Z + o); sP[q] = *reinterpret_cast<const bf16x8 *>(S + o); }
  }
  __builtin_amdgcn_sched_barrier(0);
  DRAIN_VMEM();
#pragma unroll
  for (int st = 0; st < NST; ++st) {
    const unsigned o = toff_l + st * tstep;
    if (st + WIDE_LA < NST) {
      const unsigned on = toff_l + (st + WIDE_LA) * tstep;
      cP[st + WIDE_LA] = *reinterpret_cast<const bf16x8 *>(C + on);
      if constexpr (MOD) {
        zP[st + WIDE_LA] = *reinterpret_cast<const bf16x8 *>(Z + on);
        sP[st + WIDE_LA] = *reinterpret_cast<const bf16x8 *>(S + on);
      }
    }
    float gq[8], dz[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) gq[j] = acc[j >> 2][st][j & 3] * (float)cP[st][j];
    if constexpr (MOD) {
      float ds[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        dz[j] = omega * gq[j] * (float)sP[st][j];
        ds[j] = omega * gq[j] * (float)zP[st][j];
      }
      if constexpr (LATB) {      // compiled out when the node-space reductions take the linear_lats bias sums
        if (a.wgrads) {
          ts += reduce8_scatter(ds, lane);
          tt += reduce8_scatter(gq, lane);
        }
      }
      bf16x8 pk;
#pragma unroll
      for (int j = 0; j < 8; ++j) pk[j] = (bf16_t)ds[j];
      WIDE_STORE(DS + o, pk);
#pragma unroll
      for (int j = 0; j < 8; ++j) pk[j] = (bf16_t)gq[j];
      WIDE_STORE(DT + o, pk);
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) dz[j] = omega * gq[j];
    }
    bf16x8 zp;
#pragma unroll
    for (int j = 0; j < 8; ++j) zp[j] = (bf16_t)dz[j];
    WIDE_STORE(DZ + o, zp);
    if constexpr (FP == 0) {
      hold[st] = zp;
    } else {
      lds_char *xr = gs + (16 * st + i_l) * a.xrow;
      *(LDS_AS bf16x8 *)(xr + (((c0 - 8) ^ i_l) << 4)) = hold[st];
      *(LDS_AS bf16x8 *)(xr + ((c0 ^ i_l) << 4)) = zp;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) tzv[j] += dz[j];
    __builtin_amdgcn_sched_barrier(0);   // one chunk at a time: bounds the live temporaries, keeps the request order
  }
  // column sums over the samples of this wave -> bias gradients; lanes i and i ^ 8 hold the same totals: the lower one adds
  if (a.wgrads) {
    const float tz = reduce8_scatter(tzv, lane);
    const int fi = fb_l + 64 * FP + reduce8_index(i_l);
    if (fi < a.H && !(i_l & 8)) {
      float *bsum = a.rep != nullptr ? a.rep + (long)(blockIdx.x & (NG_REP - 1)) * a.rep_w + l * a.HP : a.gp + a.off_b[l];
      atomicAdd(bsum + fi, tz);
      if constexpr (MOD && LATB) {
        atomicAdd(a.gp + a.off_bl[l] + fi, ts);
        atomicAdd(a.gp + a.off_bl[l] + a.H + fi, tt);
      }
    }
  }
}

template <int NST, int XS, bool LATB>
__device__ __forceinline__ void wide_bwd_body(const WideBwdArgs &a, const long n0) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  lds_char *gs = (lds_char *)smem;                       // g tile [NST*16][xrow] bf16, swizzled
  lds_char *w0s = gs + NST * 16 * a.xrow;                // [HP][4] fp32
  lds_char *cs = w0s + a.HP * 16;                        // coords tile [NST*16][4] fp32
  // final_linear rows [16][HP] fp32 + dOut tile [NST*16][16] fp32: read at the top of the chain only, before the first
  // write of g -- they alias the g tile when it has no pad columns to keep zero (the host sizes LDS the same way)
  const bool alias_top = a.xrow == a.HP * 2 && 16 * a.HP * 4 + NST * 16 * 64 <= NST * 16 * a.xrow;
  lds_char *wfs = alias_top ? gs : cs + NST * 16 * 16;   // [16][HP] fp32
  lds_char *ds_ = wfs + 16 * a.HP * 4;                   // dOut tile [NST*16][16] fp32
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 15, g = lane >> 4, gk = g;
  const int slab = wave;
  const bool active = slab * 64 < a.HP;
  // wide feature map (pack_stream_body, pairmap bit 2): pass p of waves 2k and 2k + 1 covers features 128 k + 64 p .. + 63,
  // i.e. the two waves request the halves of every 128-byte line of a saved row at the same time (with nodes.h's pair-major
  // map each line was fetched once per pass: 1.8 x the read traffic); this lane: features fb + 64 p .. + 7
  const unsigned fb = 128 * (slab >> 1) + 32 * (slab & 1) + 8 * g;
  const unsigned lane16 = lane * 16;
  const unsigned toff = (i * a.HP + fb) * 2u;
  const unsigned tstep = 16u * a.HP * 2u;
  const int top = a.n_layers - 1;
  const long seg_bytes = (long)XS * 4096;
  const char *sbase = (const char *)(a.stream + (long)slab * a.slab_elems);

  WIDE_TRACE(125)
  WRing2 r;
  if (active && top > 0) {
#pragma unroll
    for (int k = 0; k < WIDE_RING; ++k) ring2_fetch(r.s[k], sbase + k * 4096, lane16);
  }
  {
    for (int q = tid; q < a.HP; q += 512) *(LDS_AS f32x4 *)(w0s + q * 16) = *reinterpret_cast<const f32x4 *>(a.side_w0 + q * 4);
    for (int q = tid; q < 4 * a.HP; q += 512) *(LDS_AS f32x4 *)(wfs + q * 16) = *reinterpret_cast<const f32x4 *>(a.side_w0 + 4 * a.HP + q * 4);
    for (int q = tid; q < NST * 16 * 16; q += 512) {
      const int row = q >> 4, o = q & 15;
      ((LDS_AS float *)ds_)[q] = o < a.out_dim ? a.dout[(n0 + row) * a.out_dim + o] : 0.f;
    }
    if (tid < NST * 16) {
      const float *cp = a.coords_t + (n0 + tid) * a.in_dim;
      f32x4 c = {cp[0], a.in_dim > 1 ? cp[1] : 0.f, a.in_dim > 2 ? cp[2] : 0.f, 0.f};
      *(LDS_AS f32x4 *)(cs + tid * 16) = c;
    }
    if (a.xrow != a.HP * 2) {
      const u32x4 z4 = {0u, 0u, 0u, 0u};
      for (int q = tid; q < NST * 16 * a.xrow / 16; q += 512) *(LDS_AS u32x4 *)(gs + q * 16) = z4;
    }
  }
  __syncthreads();
  WIDE_TRACE(126)

  f32x4 acc[2][NST];
  bf16x8 hold[NST];
#pragma nounroll
  for (int l = top; l >= 0; --l) {
    const bool mod = (a.mod_mask >> l) & 1u;
    const float omega = l == 0 ? a.om_first : a.om_hidden;
    unsigned toff_l = toff, fb_l = fb;
    int i_l = i;
    asm volatile("" : "+v"(toff_l), "+v"(fb_l), "+v"(i_l));   // see k_fused_fwd: keeps LICM from hoisting address arithmetic
    const char *seg = sbase + (long)(top - 1 - l) * seg_bytes;  // this layer's segment of the stream (l < top)
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int fp = 0; fp < 2; ++fp) {
      if (active) {
#pragma unroll
        for (int ft = 0; ft < 2; ++ft)
#pragma unroll
          for (int st = 0; st < NST; ++st) acc[ft][st] = zero4;
        if (l == top) {
          // top of the chain: dy[n][f] = sum_o dOut[n][o] * Wf[o][f] on the VALU
          for (int o = 0; o < a.out_dim; ++o) {
            float d[NST];
#pragma unroll
            for (int st = 0; st < NST; ++st) d[st] = ((const LDS_AS float *)ds_)[(16 * st + i_l) * 16 + o];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
              const f32x4 w = *(const LDS_AS f32x4 *)(wfs + ((o * a.HP + fb_l + 64 * fp + 4 * q) * 4u));
#pragma unroll
              for (int st = 0; st < NST; ++st) acc[q][st] += w * d[st];
            }
          }
        } else {
          const char *start = seg + 2048 * fp;
          const char *next = fp == 0 ? seg + 2048 : (l > 0 ? seg + seg_bytes : seg + 2048);   // past the last pass: a harmless re-read
          run_pass<NST, XS>(acc, r, gs, a.xrow, i_l, gk, lane16, start, next);
        }
      }
      WIDE_TRACE(8 * l + 2 * fp)
      if (fp == 1) __syncthreads();  // every wave has finished reading g
      if (fp == 1) { WIDE_TRACE(8 * l + 3) }
      if (active) {
        // opaque copies again: otherwise the epilogue's per-tile addresses are computed in front of the GEMM and stay live
        // through it (60 registers at 7 tiles, spilled inside the peeled group)
        unsigned toff_e = toff, fb_e = fb;
        int i_e = i;
        asm volatile("" : "+v"(toff_e), "+v"(fb_e), "+v"(i_e));
        if (fp == 0) {
          if (mod) wide_epilogue<true, 0, NST, LATB>(a, acc, hold, l, omega, n0, toff_e, tstep, fb_e, i_e, lane, gs);
          else wide_epilogue<false, 0, NST, LATB>(a, acc, hold, l, omega, n0, toff_e, tstep, fb_e, i_e, lane, gs);
        } else {
          if (mod) wide_epilogue<true, 1, NST, LATB>(a, acc, hold, l, omega, n0, toff_e, tstep, fb_e, i_e, lane, gs);
          else wide_epilogue<false, 1, NST, LATB>(a, acc, hold, l, omega, n0, toff_e, tstep, fb_e, i_e, lane, gs);
        }
      }
      WIDE_TRACE(8 * l + 1 + 3 * fp)
    }
    __syncthreads();  // g of the next (lower) layer is complete
    WIDE_TRACE(8 * l + 5)
  }

  // first layer (siren.py:26-36 with is_first): the g tile now holds dz_0 (bf16, as the weight-gradient GEMMs of the other
  // layers read theirs).  d coords[n][k] = sum_f dz_0[n][f] W0[f][k]: four threads per sample, 128 features each;
  // dW0[f][k] = sum_n dz_0[n][f] x[n][k]: one thread per feature.  fp32 sums.
  {
    const int n = tid & 127, qf = tid >> 7;
    if (n < NST * 16) {
      const lds_char *row = gs + n * a.xrow;
      float dc[3] = {0.f, 0.f, 0.f};
      const int cq = a.HP >> 5;                            // 16-byte chunks per quarter row
      for (int c = qf * cq; c < (qf + 1) * cq; ++c) {
        const bf16x8 v = *(const LDS_AS bf16x8 *)(row + ((c ^ (n & 15)) << 4));
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const f32x4 w = *(const LDS_AS f32x4 *)(w0s + (8 * c + j) * 16);
          const float x = (float)v[j];
          dc[0] += x * w[0]; dc[1] += x * w[1]; dc[2] += x * w[2];
        }
      }
#pragma unroll
      for (int k = 0; k < 3; ++k)
        if (k < a.in_dim) atomicAdd(a.dcoords + (n0 + n) * a.in_dim + k, dc[k]);
    }
    if (a.wgrads) {
      for (int f = tid; f < a.H; f += 512) {
        float tw[3] = {0.f, 0.f, 0.f};
        for (int m = 0; m < NST * 16; ++m) {
          const float x = (float)*(const LDS_AS bf16_t *)(gs + m * a.xrow + (((f >> 3) ^ (m & 15)) << 4) + (f & 7) * 2);
          const f32x4 xc = *(const LDS_AS f32x4 *)(cs + m * 16);
          tw[0] += x * xc[0]; tw[1] += x * xc[1]; tw[2] += x * xc[2];
        }
        float *wsum = a.rep != nullptr ? a.rep + (long)(blockIdx.x & (NG_REP - 1)) * a.rep_w + a.n_layers * a.HP + f * 4
                                       : a.gp + a.off_w0 + (long)f * a.in_dim;
#pragma unroll
        for (int k = 0; k < 3; ++k)
          if (k < a.in_dim) atomicAdd(wsum + k, tw[k]);
      }
    }
  }
  WIDE_TRACE(127)
}

template <int BIG, int XS, bool LATB>
__global__ __launch_bounds__(512) void k_fused_bwdw(WideBwdArgs a) {
  // behind the chain workgroups: the dOut preparation of the weight-gradient launch (reads dOut only)
  if ((int)blockIdx.x >= a.g_chain) { dout_prep_body<bf16_t>(a.fin, a.dout16, blockIdx.x - a.g_chain, a.g_fin); return; }
  if ((int)blockIdx.x < a.g_big) wide_bwd_body<BIG, XS, LATB>(a, 16L * BIG * blockIdx.x);
  else wide_bwd_body<BIG - 1, XS, LATB>(a, 16L * BIG * a.g_big + 16L * (BIG - 1) * ((int)blockIdx.x - a.g_big));
}

}  // namespace

// ========================================================================================
// host side
// ========================================================================================
static inline int ksteps_w(int k) { return ((k / 32) + FUSED_RING - 1) / FUSED_RING * FUSED_RING; }

// The wide kernel reads the backward stream in its own feature map (wide_bwd_body), so the choice is a function of the descriptor
// (and the process-wide switch) only: fused_pack_args and fused_backward_chain both ask here.  CINEMA_BWD_WIDE=0 keeps
// k_fused_bwd (A/B switch).
bool fused_bwd_wide(const cinema_siren_desc *d) {
  static const bool on = [] { const char *e = getenv("CINEMA_BWD_WIDE"); return !(e && e[0] == '0'); }();
  return on && fused_eligible(d) && (d->HP == 256 || d->HP == 512);   // 8 or 16 k32 steps, whole slab pairs
}

// developer hook (not part of the C ABI in include/): choose the workgroup / wave whose phase timestamps the next launches
// record, and read them back
static int g_wtrace_block = -1, g_wtrace_wave = 0;
extern "C" int cinema_debug_wide_trace(int block, int wave, long long *host_out128) {
  g_wtrace_block = block; g_wtrace_wave = wave;
  if (host_out128) CINEMA_CHECK_HIP(hipMemcpyFromSymbol(host_out128, HIP_SYMBOL(g_wide_trace), sizeof(long long) * 128));
  return 0;
}

static int wide_max_tiles() {
  static const int v = [] { const char *e = getenv("CINEMA_BWD_WIDE_TILES"); const int x = e ? atoi(e) : 7; return x < 4 ? 4 : (x > 7 ? 7 : x); }();
  return v;
}

// `units` 16-sample tiles over workgroups of BIG or BIG - 1 tiles, ONE launch, the fewest chip fillings
static void wide_launch_plan(int64_t Npad, int bigmax, int *big, int *g_big, int *g_small) {
  const long units = Npad / 16;                                  // Npad is a multiple of 64
  const long ncu = cinema_cu_count();
  const long rounds = (units + bigmax * ncu - 1) / (bigmax * ncu);
  long nwg = rounds * ncu;
  if (units < 4 * nwg) nwg = units / 4;                          // small batches: tiles of 4
  const int b = (int)std::max<long>(4, (units + nwg - 1) / nwg);
  *big = b;
  *g_small = (int)((long)b * nwg - units);
  *g_big = (int)(nwg - *g_small);
}

static int wide_lds_bytes(const cinema_siren_desc *d, int nst) {
  const int xrow = 64 * ksteps_w(d->HP);
  const int top = 16 * d->HP * 4 + nst * 16 * 64;                // final_linear rows + dOut tile: alias the g tile when it has no pad columns
  return nst * 16 * xrow + d->HP * 16 + nst * 16 * 16 + (xrow == d->HP * 2 && top <= nst * 16 * xrow ? 0 : top);
}

template <int BIG, int XS>
static int launch_wide_x(const WideBwdArgs &a, bool latb, int grid, int smem, hipStream_t s) {
  if (latb) {
    CINEMA_SMEM_ATTR((k_fused_bwdw<BIG, XS, true>), smem);
    hipLaunchKernelGGL((k_fused_bwdw<BIG, XS, true>), dim3((unsigned)grid), dim3(512), smem, s, a);
  } else {
    CINEMA_SMEM_ATTR((k_fused_bwdw<BIG, XS, false>), smem);
    hipLaunchKernelGGL((k_fused_bwdw<BIG, XS, false>), dim3((unsigned)grid), dim3(512), smem, s, a);
  }
  CINEMA_CHECK_HIP(hipGetLastError());
  return 0;
}
template <int BIG>
static int launch_wide(const WideBwdArgs &a, bool latb, int grid, int smem, hipStream_t s) {
  return a.xsteps == 16 ? launch_wide_x<BIG, 16>(a, latb, grid, smem, s) : launch_wide_x<BIG, 8>(a, latb, grid, smem, s);
}

int wide_backward_chain(const cinema_siren_desc *d, const void *packed, const float *coords_t, const void *saved,
                        const float *dout, int64_t Npad, void *gbuf, float *gp, float *dcoords, bool wgrads,
                        hipStream_t s, void *dout16, float *dbf, bool lat_bias, float *colsum_rep, int rep_w) {
  CINEMA_REQUIRE(fused_bwd_wide(d) && d->pk_bs >= 0, "wide_backward_chain: descriptor not planned for the wide backward");
  CINEMA_REQUIRE(Npad % FUSED_TS == 0 && Npad >= FUSED_TS, "wide_backward_chain: Npad %ld not a positive multiple of %d", (long)Npad, FUSED_TS);
  WideBwdArgs a = {};
  a.stream = (const bf16_t *)packed + d->pk_bs;
  a.slab_elems = (long)d->bs_steps * FUSED_STEP_ELEMS;
  int bias_floats = 0;
  for (int i = 0; i < d->n_layers; ++i) {
    a.sv_c[i] = d->sv_c[i]; a.sv_z[i] = d->sv_z[i]; a.sv_s[i] = d->sv_s[i];
    a.gb_z[i] = d->gb_z[i]; a.gb_s[i] = d->gb_s[i]; a.gb_t[i] = d->gb_t[i];
    a.off_b[i] = d->off_b[i]; a.off_bl[i] = d->off_bl[i];
    bias_floats += d->HP * (((d->modulated_mask >> i) & 1u) ? 3 : 1);
  }
  a.side_w0 = (const float *)((const bf16_t *)packed + d->pk_side) + bias_floats;
  a.dout = dout; a.coords_t = coords_t;
  a.saved = (const bf16_t *)saved; a.slot = (long)Npad * d->HP; a.gbuf = (bf16_t *)gbuf;
  a.off_w0 = d->off_w[0];
  a.gp = gp; a.wgrads = wgrads ? 1 : 0; a.dcoords = dcoords;
  a.mod_mask = d->modulated_mask; a.n_layers = d->n_layers; a.in_dim = d->in_dim; a.H = d->hidden; a.HP = d->HP;
  a.out_dim = d->out_dim; a.xsteps = ksteps_w(d->HP); a.xrow = 64 * a.xsteps;
  a.om_first = d->omega_first; a.om_hidden = d->omega_hidden;
  a.trace_block = g_wtrace_block; a.trace_wave = g_wtrace_wave;
  a.rep = wgrads ? colsum_rep : nullptr; a.rep_w = rep_w;
  int big, g_small;
  wide_launch_plan(Npad, wide_max_tiles(), &big, &a.g_big, &g_small);
  a.nst_big = big;
  const int smem = wide_lds_bytes(d, big);
  CINEMA_REQUIRE(smem <= 160 * 1024, "wide_backward_chain: LDS budget exceeded (%d bytes, %d tiles)", smem, big);
  a.g_chain = a.g_big + g_small;
  if (dout16 != nullptr) {
    a.fin = FinalBwdArgs{dout, d->out_dim, dbf, (int)Npad};
    a.dout16 = (bf16_t *)dout16;
    a.g_fin = (int)std::min<int64_t>((Npad + 511) / 512, 128);
  }
  const int grid = a.g_chain + a.g_fin;
  static const bool no_latb = getenv("CINEMA_BWD_NO_LATB") != nullptr;   // developer timing of the node-gradient variant alone
  const bool latb = wgrads && lat_bias && !no_latb;
  switch (big) {
    case 4: return launch_wide<4>(a, latb, grid, smem, s);
    case 5: return launch_wide<5>(a, latb, grid, smem, s);
    case 6: return launch_wide<6>(a, latb, grid, smem, s);
    default: return launch_wide<7>(a, latb, grid, smem, s);
  }
}
