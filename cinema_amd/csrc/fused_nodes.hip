// Activation-stationary SIREN chain kernels, "nodes" path (bf16 MFMA, fp32 accumulate): fused.hip's kernels with the
// FiLM modulation taken from per-subject node tables instead of a per-sample latent GEMM (nodes.h).
// Reference arithmetic: models/siren.py:26-36 (SineLayer.forward), :59-61 (Siren.forward).
//
// Per modulated layer and half (scale, shift) the latent GEMM  [64 features x LP] . lat[n]  of fused.hip (37 % of its
// weight stream) becomes ONE fp16 k32 MFMA step in the layer's epilogue:  s[f][n] = sum_slot U[f][slot] * w[slot][n];
// the linear_lats bias rides in slot 31.  The latent rows never exist per sample: no lat tile in LDS, no K1 gather,
// and the final linear takes its latent part from V the same way.  A workgroup (8 waves) owns NST * 16 samples; with
// the LDS the lat tile leaves behind a tile holds up to 112 samples, so one weight stream serves 1.5 - 1.75 x the
// samples of fused.hip's 64.
#include "fused.h"
#include "gemm_engine.h"
#include "nodes.h"
#include <algorithm>
#include <cstdlib>

namespace {

#include "fused_dev.h"
#include "dw_device.h"

// run_seg of fused_dev.h with its first group of four steps peeled off.  gfx950 counts loads AND stores in vmcnt, so the
// first counted wait after an epilogue also drains the epilogue's stores.  The epilogues drain the ring's prefetches at
// their START (they were issued a whole GEMM ago: free) -- hipcc then knows that the four resident steps need no wait,
// the peeled group runs them at once, and the stores have four steps of MFMA time to retire before the loop's first
// counted wait.  (Without the drain hipcc merges the epilogue's pending loads / stores / atomics into the loop header's
// state and waits vmcnt(0) in EVERY group of four steps: k_fused_bwd of round 1 ran that way.)
template <int NST>
__device__ __forceinline__ void run_seg_peeled(f32x4 (&acc)[4][NST], WRing &r, const lds_char *tile, int rowbytes, int nsteps,
                                               int i, int gk, unsigned lane16) {
  const int rowoff = i * rowbytes;
  { FUSED_STEP(0, 0) FUSED_STEP(1, 1) FUSED_STEP(2, 2) FUSED_STEP(3, 3) }
  for (int k = 4; k < nsteps; k += 4) {
    FUSED_STEP(0, k) FUSED_STEP(1, k + 1) FUSED_STEP(2, k + 2) FUSED_STEP(3, k + 3)
  }
}
// s_waitcnt vmcnt(0) only (expcnt, lgkmcnt left at their maxima): gfx9 encoding
#define DRAIN_VMEM() __builtin_amdgcn_s_waitcnt(0x0F70)

// The U fragments of one half (scale or shift) of a layer: four MFMA A fragments in registers of their own, fetched at the
// start of the epilogue pass that uses them (their latency hides behind the pass's first stores).
__device__ __forceinline__ void u_fetch(bf16x8 (&u)[4], const char *p, unsigned lane16) {
#pragma unroll
  for (int ft = 0; ft < 4; ++ft) u[ft] = *reinterpret_cast<const bf16x8 *>(p + ft * 1024 + lane16);
}
__device__ __forceinline__ f16x8 as_f16(const bf16x8 &v) { return __builtin_bit_cast(f16x8, v); }

// Sample tiles.  A batch grouped by subject is cut into tiles of at most 16 consecutive rows of ONE subject
// (cinema_node_tiles: entry = {subject, first row, rows, 0}); a workgroup owns NST consecutive tiles and keeps each in 16
// LDS row slots, so every MFMA sample tile has one subject -- one U fragment set per step, no masking -- while the
// tensors in HBM stay indexed by sample row.  Slots beyond a tile's rows compute finite garbage that is never stored.
struct TileInfo { int sub[8], row[8], nr[8]; };
template <int NST>
__device__ __forceinline__ void load_tiles(TileInfo &t, const int4 *tiles, const int *n_tiles, long t0) {
  const int T = *n_tiles;
#pragma unroll
  for (int st = 0; st < NST; ++st) {
    const bool ok = t0 + st < T;
    const int4 e = tiles[ok ? t0 + st : 0];
    t.sub[st] = __builtin_amdgcn_readfirstlane(ok ? e.x : 0);
    t.row[st] = __builtin_amdgcn_readfirstlane(ok ? e.y : 0);
    t.nr[st] = __builtin_amdgcn_readfirstlane(ok ? e.z : 0);
  }
}
#define NODES_MAX_NST 8

// ----------------------------------------------------------------------------------------
// forward
// ----------------------------------------------------------------------------------------
struct NodesFwdArgs {
  int g_big, nst_big;                // workgroups [0, g_big) own nst_big sample tiles, the rest nst_big - 1
  const bf16_t *stream; long slab_elems;      // main-weight stream: per 64-feature slab the layers 1.. in order
  const char *U; long u_layer_bytes, u_slab_bytes; const char *V; int S;
  int umod[CINEMA_MAX_LAYERS];       // modulated ordinal of a layer
  const float *side; int lb_b[CINEMA_MAX_LAYERS], lb_total;
  unsigned mod_mask; int n_layers, in_dim, H, HP, LP, out_dim;
  float om_first, om_hidden;
  const float *coords_t; const f16_t *wrows; const int4 *tiles; const int *n_tiles;
  bf16_t *saved; long slot; int sv_y[CINEMA_MAX_LAYERS], sv_c[CINEMA_MAX_LAYERS], sv_z[CINEMA_MAX_LAYERS], sv_s[CINEMA_MAX_LAYERS];
  int round_s;                       // 1: s is rounded to bf16 (and saved) as fused.hip does -- its backward reads it
  const bf16_t *wf; const float *bf; float *out;
  int xrow, xsteps;
};

template <int NST, bool KEEP>
__device__ __forceinline__ void nodes_fwd_body(const NodesFwdArgs &a, const long t0) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  lds_char *xs = (lds_char *)smem;                         // x tile [NST*16][xrow] bf16, swizzled
  lds_char *wt = xs + NST * 16 * a.xrow;                   // w rows [NST*16][32] fp16
  lds_char *cs = wt + NST * 16 * 64;                       // registered coordinates [NST*16][4] fp32
  lds_char *bs = cs + NST * 16 * 16;                       // side block: biases, first-layer weights
  TileInfo ti;
  load_tiles<NST>(ti, a.tiles, a.n_tiles, t0);
  if (ti.nr[0] == 0) return;                               // beyond the last tile (the grid is sized for the worst case)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 15, g = lane >> 4, gk = g;
  const int slab = wave;
  const bool active = slab * 64 < a.HP;
  const unsigned fb = slab * 64 + 8 * g;                   // fragment pair p of this lane: features fb + 32 p .. + 7 (nodes.h)
  const unsigned lane16 = lane * 16;
  const unsigned toff = (i * a.HP + fb) * 2u;
  const unsigned rowb = a.HP * 2u;
  constexpr bool keep = KEEP;              // false (dense inference): no cos, no conversions, no store addresses
  {
    const u32x4 z4 = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int st = 0; st < NST; ++st) {
      if (tid < 64) {                                        // w rows of the tile: 16 slots x 4 chunks
        const int ii = tid >> 2, c = tid & 3;
        u32x4 v = z4;
        if (ii < ti.nr[st]) v = *reinterpret_cast<const u32x4 *>((const char *)(a.wrows + (long)(ti.row[st] + ii) * NODE_SLOTS) + c * 16);
        *(LDS_AS u32x4 *)(wt + (16 * st + ii) * 64 + c * 16) = v;
      } else if (tid < 80) {
        const int ii = tid - 64;
        f32x4 c = {0.f, 0.f, 0.f, 0.f};
        if (ii < ti.nr[st]) {
          const float *cp = a.coords_t + (long)(ti.row[st] + ii) * a.in_dim;
          c[0] = cp[0]; c[1] = a.in_dim > 1 ? cp[1] : 0.f; c[2] = a.in_dim > 2 ? cp[2] : 0.f;
        }
        *(LDS_AS f32x4 *)(cs + (16 * st + ii) * 16) = c;
      }
    }
    if (a.xrow != a.HP * 2)
      for (int q = tid; q < NST * 16 * a.xrow / 16; q += 512) *(LDS_AS u32x4 *)(xs + q * 16) = z4;
    const int nb4 = (a.lb_total + 4 * a.HP) / 4;
    for (int q = tid; q < nb4; q += 512) *(LDS_AS f32x4 *)(bs + q * 16) = *reinterpret_cast<const f32x4 *>(a.side + q * 4);
  }
  __syncthreads();

  WRing r;
  if (active) ring_init(r, (const char *)(a.stream + (long)slab * a.slab_elems), lane16);
  bf16x8 uu[4];

  f32x4 acc[4][NST];
#pragma nounroll
  for (int l = 0; l < a.n_layers; ++l) {
    const bool mod = (a.mod_mask >> l) & 1u;
    const float omega = l == 0 ? a.om_first : a.om_hidden;
    unsigned toff_l = toff, fb_l = fb;
    int i_l = i;
    asm volatile("" : "+v"(toff_l), "+v"(fb_l), "+v"(i_l));   // see k_fused_fwd: keeps LICM from hoisting address arithmetic
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    if (active) {
#pragma unroll
      for (int ft = 0; ft < 4; ++ft)
#pragma unroll
        for (int st = 0; st < NST; ++st) acc[ft][st] = zero4;      // the bias is added in the epilogue
      if (l == 0) {
        // first layer: K = in_dim <= 3 on the VALU in fp32, operands from LDS
#pragma unroll
        for (int st = 0; st < NST; ++st) {
          const f32x4 c = *(const LDS_AS f32x4 *)(cs + (16 * st + i_l) * 16);
#pragma unroll
          for (int ft = 0; ft < 4; ++ft) {
            f32x4 v;
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
              const f32x4 w = *(const LDS_AS f32x4 *)(bs + (a.lb_total + (fb_l + 32 * (ft >> 1) + 4 * (ft & 1) + rr) * 4) * 4);
              v[rr] = w[0] * c[0] + w[1] * c[1] + w[2] * c[2];
            }
            acc[ft][st] = v;
          }
        }
      } else {
        run_seg_peeled(acc, r, xs, a.xrow, a.xsteps, i_l, gk, lane16);
      }
    }
    __syncthreads();  // every wave has finished reading x
    if (active) {
      DRAIN_VMEM();                                           // the ring's four resident steps (see run_seg_peeled)
      char *Y = (char *)(a.saved + a.slot * a.sv_y[l]);
      char *C = (char *)(a.saved + a.slot * a.sv_c[l]);
      char *Z = mod ? (char *)(a.saved + a.slot * a.sv_z[l]) : Y;
      char *S = (mod && a.round_s) ? (char *)(a.saved + a.slot * a.sv_s[l]) : Y;
      const int c0 = fb_l >> 3;
      const lds_char *bl = bs + (a.lb_b[l] + fb_l) * 4u;       // this lane's biases: 8 per fragment pair, 32 features apart
      // The epilogue walks 16-byte chunks (one sample tile x 8 features = two accumulator fragments) so that few
      // temporaries are alive next to the accumulators and the ring; modulated layers walk the tiles once per fragment
      // pair fp, with the scale and shift U fragments of that pair only (4 fragments = 16 registers) resident.
      if (mod) {
        const char *ubase = a.U + (long)a.umod[l] * a.u_layer_bytes + (long)slab * a.u_slab_bytes + lane16;
#pragma unroll
        for (int fp = 0; fp < 2; ++fp) {
          const f32x4 b0 = *(const LDS_AS f32x4 *)(bl + 128 * fp), b1 = *(const LDS_AS f32x4 *)(bl + 128 * fp + 16);
          int cur_s = ti.sub[0];
#pragma unroll
          for (int q = 0; q < 4; ++q)      // q = 2 * half + fragment of the pair
            uu[q] = *reinterpret_cast<const bf16x8 *>(ubase + (long)cur_s * 8192 + (q >> 1) * 4096 + (2 * fp + (q & 1)) * 1024);
#pragma unroll
          for (int st = 0; st < NST; ++st) {
            const f16x8 bw = *(const LDS_AS f16x8 *)(wt + (16 * st + i_l) * 64 + g * 16);
            if (ti.sub[st] != cur_s) {                           // the next tile belongs to another subject
              cur_s = ti.sub[st];
#pragma unroll
              for (int q = 0; q < 4; ++q)
                uu[q] = *reinterpret_cast<const bf16x8 *>(ubase + (long)cur_s * 8192 + (q >> 1) * 4096 + (2 * fp + (q & 1)) * 1024);
            }
            const bool live = keep && i_l < ti.nr[st];
            const unsigned o = toff_l + (unsigned)ti.row[st] * rowb + 64u * fp;
            f32x4 sa[2], ta[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
              sa[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(as_f16(uu[q]), bw, zero4, 0, 0, 0);
              ta[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(as_f16(uu[2 + q]), bw, zero4, 0, 0, 0);
            }
            bf16x8 pz, ps, yp, cp;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              const float z = acc[2 * fp + (j >> 2)][st][j & 3] + (j < 4 ? b0[j & 3] : b1[j & 3]);
              float sv = sa[j >> 2][j & 3];
              pz[j] = (bf16_t)z;
              if (a.round_s) { ps[j] = (bf16_t)sv; sv = (float)ps[j]; }
              const float ph = omega * z * sv + ta[j >> 2][j & 3];
              yp[j] = (bf16_t)__sinf(ph);
              cp[j] = (bf16_t)__cosf(ph);
            }
            if (live) {
              *reinterpret_cast<bf16x8 *>(Z + o) = pz;
              if (a.round_s) *reinterpret_cast<bf16x8 *>(S + o) = ps;
              *reinterpret_cast<bf16x8 *>(Y + o) = yp;
              *reinterpret_cast<bf16x8 *>(C + o) = cp;
            }
            *(LDS_AS bf16x8 *)(xs + (16 * st + i_l) * a.xrow + (((c0 + 4 * fp) ^ i_l) << 4)) = yp;
            __builtin_amdgcn_sched_barrier(0);   // one chunk at a time: bounds the live temporaries
          }
        }
      } else {
#pragma unroll
        for (int fp = 0; fp < 2; ++fp) {
          const f32x4 b0 = *(const LDS_AS f32x4 *)(bl + 128 * fp), b1 = *(const LDS_AS f32x4 *)(bl + 128 * fp + 16);
#pragma unroll
          for (int st = 0; st < NST; ++st) {
            const bool live = keep && i_l < ti.nr[st];
            const unsigned o = toff_l + (unsigned)ti.row[st] * rowb + 64u * fp;
            bf16x8 yp, cp;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              const float ph = omega * (acc[2 * fp + (j >> 2)][st][j & 3] + (j < 4 ? b0[j & 3] : b1[j & 3]));
              yp[j] = (bf16_t)__sinf(ph);
              cp[j] = (bf16_t)__cosf(ph);
            }
            if (live) {
              *reinterpret_cast<bf16x8 *>(Y + o) = yp;
              *reinterpret_cast<bf16x8 *>(C + o) = cp;
            }
            *(LDS_AS bf16x8 *)(xs + (16 * st + i_l) * a.xrow + (((c0 + 4 * fp) ^ i_l) << 4)) = yp;
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
    }
    __syncthreads();  // x of the next layer is complete
  }

  // final linear (siren.py:61): out[n][o] = bf[o] + sum_k x_L[n][k] Wf[o][k] + sum_slot V[o][slot] w[slot][n]
  if (wave < NST) {
    const int st = wave, KT = a.HP + a.LP;
    f32x4 o4 = {0.f, 0.f, 0.f, 0.f};
    const char *wrow = (const char *)a.wf;
    const unsigned woff = (i * KT + 8 * gk) * 2u;
    const lds_char *xrp = xs + (16 * st + i) * a.xrow;
    const int nx = a.HP >> 5;
    for (int k0 = 0; k0 < nx; k0 += 8) {
      bf16x8 w[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int kk = k0 + k < nx ? k0 + k : nx - 1;
        w[k] = *reinterpret_cast<const bf16x8 *>(wrow + (woff + 64u * kk));
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int kk = k0 + k;
        if (kk < nx) {
          const bf16x8 b = *(const LDS_AS bf16x8 *)(xrp + (((((kk) << 2) + gk) ^ i) << 4));
          o4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[k], b, o4, 0, 0, 0);
        }
      }
    }
    int sj = 0, row0 = 0, nr = 0;
#pragma unroll
    for (int q = 0; q < NST; ++q)
      if (q == st) { sj = ti.sub[q]; row0 = ti.row[q]; nr = ti.nr[q]; }
    {
      const f16x8 bw = *(const LDS_AS f16x8 *)(wt + (16 * st + i) * 64 + gk * 16);
      const f16x8 v = *reinterpret_cast<const f16x8 *>(a.V + (long)sj * 1024 + lane16);
      o4 = __builtin_amdgcn_mfma_f32_16x16x32_f16(v, bw, o4, 0, 0, 0);
    }
    float *ot = a.out + (long)row0 * a.out_dim;
    if (i < nr) {
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int o = 4 * g + rr;
        if (o < a.out_dim) ot[(unsigned)(i * a.out_dim + o)] = o4[rr] + a.bf[o];
      }
    }
  }
}

template <int BIG, bool KEEP>
__global__ __launch_bounds__(512) void k_nodes_fwd(NodesFwdArgs a) {
  if ((int)blockIdx.x < a.g_big) nodes_fwd_body<BIG, KEEP>(a, (long)BIG * blockIdx.x);
  else nodes_fwd_body<BIG - 1, KEEP>(a, (long)BIG * a.g_big + (long)(BIG - 1) * ((int)blockIdx.x - a.g_big));
}

// ----------------------------------------------------------------------------------------
// backward chain (mirror of the forward; see k_fused_bwd).  s is recomputed from the node table with the same MFMA step
// the forward used (the same bits), so no s tensor is saved; ds, dt stream out for the node-space reduction.
// ----------------------------------------------------------------------------------------
struct NodesBwdArgs {
  FinalBwdArgs fin; bf16_t *dout16; int g_chain, g_fin;
  int g_big, nst_big;
  const bf16_t *stream; long slab_elems;
  const char *U; long u_layer_bytes, u_slab_bytes; int S;
  int umod[CINEMA_MAX_LAYERS];
  const float *side_w0;
  const float *dout; const float *coords_t; const f16_t *wrows; const int4 *tiles; const int *n_tiles;
  const bf16_t *saved; long slot; int sv_c[CINEMA_MAX_LAYERS], sv_z[CINEMA_MAX_LAYERS], sv_s[CINEMA_MAX_LAYERS];
  int round_s;
  bf16_t *gbuf; int gb_z[CINEMA_MAX_LAYERS], gb_s[CINEMA_MAX_LAYERS], gb_t[CINEMA_MAX_LAYERS];
  float *gp; long off_b[CINEMA_MAX_LAYERS], off_bl[CINEMA_MAX_LAYERS], off_w0; int wgrads;
  float *dcoords;
  unsigned mod_mask; int n_layers, in_dim, H, HP, out_dim, xrow, xsteps;
  float om_first, om_hidden;
};

template <bool MOD, bool FIRST, int NST>
__device__ __forceinline__ void nodes_bwd_epilogue(const NodesBwdArgs &a, f32x4 (&acc)[4][NST], const TileInfo &ti, int l,
                                                   float omega, unsigned toff_l, unsigned rowb, unsigned fb_l, int i_l, int g,
                                                   int lane, unsigned lane16, const char *ubase, lds_char *gs,
                                                   const lds_char *w0s, const lds_char *cs, const lds_char *wt) {
  const char *C = (const char *)(a.saved + a.slot * a.sv_c[l]);
  const char *Z = MOD ? (const char *)(a.saved + a.slot * a.sv_z[l]) : C;
  char *DZ = (char *)(a.gbuf + a.slot * a.gb_z[l]);
  char *DS = MOD ? (char *)(a.gbuf + a.slot * a.gb_s[l]) : DZ;
  char *DT = MOD ? (char *)(a.gbuf + a.slot * a.gb_t[l]) : DZ;
  const int c0 = fb_l >> 3;
  // column sums (bias / first-layer weight gradients) per 8-feature chunk fp: one scalar per tensor and chunk stays live
  float tz[2] = {0.f, 0.f}, ts_[2] = {0.f, 0.f}, tt[2] = {0.f, 0.f}, tw[2][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  // as in the forward: tiles are walked once per fragment pair fp, with that pair's scale fragments only resident
#pragma unroll
  for (int fp = 0; fp < 2; ++fp) {
    bf16x8 uu[2];
    int cur_s = 0;
    if constexpr (MOD) {
      cur_s = ti.sub[0];
#pragma unroll
      for (int q = 0; q < 2; ++q) uu[q] = *reinterpret_cast<const bf16x8 *>(ubase + (long)cur_s * 8192 + (2 * fp + q) * 1024 + lane16);
    }
#pragma unroll
    for (int st = 0; st < NST; ++st) {
      f16x8 bw = {0, 0, 0, 0, 0, 0, 0, 0};
      if constexpr (MOD) {
        bw = *(const LDS_AS f16x8 *)(wt + (16 * st + i_l) * 64 + g * 16);
        if (ti.sub[st] != cur_s) {
          cur_s = ti.sub[st];
#pragma unroll
          for (int q = 0; q < 2; ++q) uu[q] = *reinterpret_cast<const bf16x8 *>(ubase + (long)cur_s * 8192 + (2 * fp + q) * 1024 + lane16);
        }
      }
      const bool live = i_l < ti.nr[st];                   // slots beyond the tile's rows: zero gradients, nothing stored
      const unsigned o = toff_l + (unsigned)ti.row[st] * rowb + 64u * fp;
      float gq[8], dz[8];
      {
        bf16x8 cc = {0, 0, 0, 0, 0, 0, 0, 0};
        if (live) cc = *reinterpret_cast<const bf16x8 *>(C + o);
#pragma unroll
        for (int j = 0; j < 8; ++j) gq[j] = acc[2 * fp + (j >> 2)][st][j & 3] * (float)cc[j];
      }
      if constexpr (MOD) {
        bf16x8 zz = {0, 0, 0, 0, 0, 0, 0, 0};
        if (live) zz = *reinterpret_cast<const bf16x8 *>(Z + o);
        // s of this chunk: the forward's MFMA step again
        f32x4 sa[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) sa[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(as_f16(uu[q]), bw, zero4, 0, 0, 0);
        float ds[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float sj = sa[j >> 2][j & 3];
          if (a.round_s) sj = (float)(bf16_t)sj;
          dz[j] = omega * gq[j] * sj;
          ds[j] = omega * gq[j] * (float)zz[j];
        }
        bf16x8 pk;
#pragma unroll
        for (int j = 0; j < 8; ++j) pk[j] = (bf16_t)ds[j];
        if (live) *reinterpret_cast<bf16x8 *>(DS + o) = pk;
#pragma unroll
        for (int j = 0; j < 8; ++j) pk[j] = (bf16_t)gq[j];
        if (live) *reinterpret_cast<bf16x8 *>(DT + o) = pk;
        if (a.wgrads) {
          ts_[fp] += reduce8_scatter(ds, lane);
          tt[fp] += reduce8_scatter(gq, lane);
        }
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) dz[j] = omega * gq[j];
      }
      bf16x8 zp;
#pragma unroll
      for (int j = 0; j < 8; ++j) zp[j] = (bf16_t)dz[j];
      if (live) *reinterpret_cast<bf16x8 *>(DZ + o) = zp;
      if constexpr (!FIRST) {
        *(LDS_AS bf16x8 *)(gs + (16 * st + i_l) * a.xrow + (((c0 + 4 * fp) ^ i_l) << 4)) = zp;
      } else {
        // first layer: d coords = dz W0, dW0 = dz^T coords
        const f32x4 xc = *(const LDS_AS f32x4 *)(cs + (16 * st + i_l) * 16);
        float dc[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const f32x4 w = *(const LDS_AS f32x4 *)(w0s + (fb_l + 32 * fp + j) * 16);
#pragma unroll
          for (int k = 0; k < 3; ++k) dc[k] += dz[j] * w[k];
        }
        if (a.wgrads) {
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            float p[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) p[j] = dz[j] * xc[k];
            tw[fp][k] += reduce8_scatter(p, lane);
          }
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          dc[k] += __shfl_xor(dc[k], 16, 64);
          dc[k] += __shfl_xor(dc[k], 32, 64);
        }
        if (g == 0 && live) {
#pragma unroll
          for (int k = 0; k < 3; ++k)
            if (k < a.in_dim) atomicAdd(a.dcoords + (long)(ti.row[st] + i_l) * a.in_dim + k, dc[k]);
        }
      }
      if (a.wgrads) tz[fp] += reduce8_scatter(dz, lane);
      __builtin_amdgcn_sched_barrier(0);   // one chunk at a time: bounds the live temporaries
    }
  }
  // column sums over the samples of this wave -> bias (and first-layer weight) gradients; lanes i and i ^ 8 hold the
  // same totals: the lower one adds them
  if (a.wgrads && !(i_l & 8)) {
#pragma unroll
    for (int fp = 0; fp < 2; ++fp) {
      const int fi = fb_l + 32 * fp + reduce8_index(i_l);
      if (fi < a.H) {
        atomicAdd(a.gp + a.off_b[l] + fi, tz[fp]);
        if constexpr (MOD) {
          atomicAdd(a.gp + a.off_bl[l] + fi, ts_[fp]);
          atomicAdd(a.gp + a.off_bl[l] + a.H + fi, tt[fp]);
        }
        if constexpr (FIRST) {
#pragma unroll
          for (int k = 0; k < 3; ++k)
            if (k < a.in_dim) atomicAdd(a.gp + a.off_w0 + (long)fi * a.in_dim + k, tw[fp][k]);
        }
      }
    }
  }
}

template <int NST>
__device__ __forceinline__ void nodes_bwd_body(const NodesBwdArgs &a, const long t0) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  lds_char *gs = (lds_char *)smem;                         // g tile [NST*16][xrow] bf16, swizzled
  lds_char *wt = gs + NST * 16 * a.xrow;                   // w rows [NST*16][32] fp16
  lds_char *cs = wt + NST * 16 * 64;                       // coords tile [NST*16][4] fp32
  lds_char *w0s = cs + NST * 16 * 16;                      // [HP][4] fp32
  // final_linear rows [16][HP] fp32 + dOut tile [NST*16][16] fp32: read at the top of the chain only, before the first
  // epilogue writes g -- they alias the g tile when it has no pad columns to keep zero (the host sizes LDS the same way)
  const bool alias_top = a.xrow == a.HP * 2 && 16 * a.HP * 4 + NST * 16 * 64 <= NST * 16 * a.xrow;
  lds_char *wfs = alias_top ? gs : w0s + a.HP * 16;        // [16][HP] fp32
  lds_char *ds_ = wfs + 16 * a.HP * 4;                     // dOut tile [NST*16][16] fp32
  TileInfo ti;
  load_tiles<NST>(ti, a.tiles, a.n_tiles, t0);
  if (ti.nr[0] == 0) return;                               // beyond the last tile
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 15, g = lane >> 4, gk = g;
  const int slab = wave;
  const bool active = slab * 64 < a.HP;
  const unsigned fb = slab * 64 + 8 * g;                   // fragment pair p of this lane: features fb + 32 p .. + 7 (nodes.h)
  const unsigned lane16 = lane * 16;
  const unsigned toff = (i * a.HP + fb) * 2u;
  const unsigned rowb = a.HP * 2u;
  {
    const u32x4 z4 = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int st = 0; st < NST; ++st) {
      if (tid < 64) {
        const int ii = tid >> 2, c = tid & 3;
        u32x4 v = z4;
        if (ii < ti.nr[st]) v = *reinterpret_cast<const u32x4 *>((const char *)(a.wrows + (long)(ti.row[st] + ii) * NODE_SLOTS) + c * 16);
        *(LDS_AS u32x4 *)(wt + (16 * st + ii) * 64 + c * 16) = v;
      } else if (tid < 80) {
        const int ii = tid - 64;
        f32x4 c = {0.f, 0.f, 0.f, 0.f};
        if (ii < ti.nr[st]) {
          const float *cp = a.coords_t + (long)(ti.row[st] + ii) * a.in_dim;
          c[0] = cp[0]; c[1] = a.in_dim > 1 ? cp[1] : 0.f; c[2] = a.in_dim > 2 ? cp[2] : 0.f;
        }
        *(LDS_AS f32x4 *)(cs + (16 * st + ii) * 16) = c;
      } else if (tid >= 256) {                               // dOut tile: 16 slots x 16 outputs
        const int q = tid - 256, ii = q >> 4, o = q & 15;
        ((LDS_AS float *)ds_)[(16 * st + ii) * 16 + o] =
            (ii < ti.nr[st] && o < a.out_dim) ? a.dout[(long)(ti.row[st] + ii) * a.out_dim + o] : 0.f;
      }
    }
    for (int q = tid; q < a.HP; q += 512) *(LDS_AS f32x4 *)(w0s + q * 16) = *reinterpret_cast<const f32x4 *>(a.side_w0 + q * 4);
    for (int q = tid; q < 4 * a.HP; q += 512) *(LDS_AS f32x4 *)(wfs + q * 16) = *reinterpret_cast<const f32x4 *>(a.side_w0 + 4 * a.HP + q * 4);
    if (a.xrow != a.HP * 2)
      for (int q = tid; q < NST * 16 * a.xrow / 16; q += 512) *(LDS_AS u32x4 *)(gs + q * 16) = z4;
  }
  __syncthreads();

  WRing r;
  if (active && a.n_layers > 1) ring_init(r, (const char *)(a.stream + (long)slab * a.slab_elems), lane16);
  const int top = a.n_layers - 1;

  f32x4 acc[4][NST];
#pragma nounroll
  for (int l = top; l >= 0; --l) {
    const bool mod = (a.mod_mask >> l) & 1u;
    const float omega = l == 0 ? a.om_first : a.om_hidden;
    unsigned toff_l = toff, fb_l = fb;
    int i_l = i;
    asm volatile("" : "+v"(toff_l), "+v"(fb_l), "+v"(i_l));
    if (active) {
      const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ft = 0; ft < 4; ++ft)
#pragma unroll
        for (int st = 0; st < NST; ++st) acc[ft][st] = zero4;
      if (l == top) {
        // top of the chain: dy[n][f] = sum_o dOut[n][o] * Wf[o][f] on the VALU
        for (int o = 0; o < a.out_dim; ++o) {
          float d[NST];
#pragma unroll
          for (int st = 0; st < NST; ++st) d[st] = ((const LDS_AS float *)ds_)[(16 * st + i_l) * 16 + o];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x4 w = *(const LDS_AS f32x4 *)(wfs + ((o * a.HP + fb_l + 32 * (q >> 1) + 4 * (q & 1)) * 4u));
#pragma unroll
            for (int st = 0; st < NST; ++st) acc[q][st] += w * d[st];
          }
        }
      } else {
        run_seg_peeled(acc, r, gs, a.xrow, a.xsteps, i_l, gk, lane16);
      }
    }
    __syncthreads();  // every wave has finished reading g
    if (active) {
      DRAIN_VMEM();                                           // the ring's four resident steps (see run_seg_peeled)
      const char *ub = mod ? a.U + (long)a.umod[l] * a.u_layer_bytes + (long)slab * a.u_slab_bytes : nullptr;
      if (l == 0) {
        if (mod) nodes_bwd_epilogue<true, true, NST>(a, acc, ti, l, omega, toff_l, rowb, fb_l, i_l, g, lane, lane16, ub, gs, w0s, cs, wt);
        else nodes_bwd_epilogue<false, true, NST>(a, acc, ti, l, omega, toff_l, rowb, fb_l, i_l, g, lane, lane16, ub, gs, w0s, cs, wt);
      } else {
        if (mod) nodes_bwd_epilogue<true, false, NST>(a, acc, ti, l, omega, toff_l, rowb, fb_l, i_l, g, lane, lane16, ub, gs, w0s, cs, wt);
        else nodes_bwd_epilogue<false, false, NST>(a, acc, ti, l, omega, toff_l, rowb, fb_l, i_l, g, lane, lane16, ub, gs, w0s, cs, wt);
      }
    }
    __syncthreads();  // g of the next (lower) layer is complete
  }
}

template <int BIG>
__global__ __launch_bounds__(512) void k_nodes_bwd(NodesBwdArgs a) {
  if ((int)blockIdx.x >= a.g_chain) { dout_prep_body<bf16_t>(a.fin, a.dout16, blockIdx.x - a.g_chain, a.g_fin); return; }
  if ((int)blockIdx.x < a.g_big) nodes_bwd_body<BIG>(a, (long)BIG * blockIdx.x);
  else nodes_bwd_body<BIG - 1>(a, (long)BIG * a.g_big + (long)(BIG - 1) * ((int)blockIdx.x - a.g_big));
}

}  // namespace

// ========================================================================================
// host side
// ========================================================================================
static inline int ksteps_h(int k) { return ((k / 32) + FUSED_RING - 1) / FUSED_RING * FUSED_RING; }

// Launch plan: `units` 16-sample tiles over workgroups of BIG or BIG - 1 tiles, ONE launch.  A workgroup's time is its
// weight stream (independent of its sample count) plus the epilogues of its samples, so the plan aims at the fewest
// chip fillings: measured at cfg-D: 4-tile workgroups (0 spills) beat 6/7-tile ones (fewer weight streams, but spills): default 4, CINEMA_NODES_BIG=4..7.
static void nodes_launch_plan(int64_t Npad, int64_t S, int bigmax, int *big, int *g_big, int *g_small) {
  const long units = Npad / 16 + std::min<long>(S, Npad);      // upper bound of the tile count (cinema_node_tiles)
  const long ncu = cinema_cu_count();
  const long rounds = (units + bigmax * ncu - 1) / (bigmax * ncu);
  long nwg = rounds * ncu;
  if (units < 4 * nwg) nwg = (units + 3) / 4;                 // small batches: tiles of 4 and 3
  int b = (int)((units + nwg - 1) / nwg);
  if (b < 4) b = 4;
  *big = b;
  *g_small = (int)((long)b * nwg - units);
  *g_big = (int)(nwg - *g_small);
}

static int nodes_big_default() {
  static const int v = [] { const char *e = getenv("CINEMA_NODES_BIG"); const int x = e ? atoi(e) : 4; return x < 4 ? 4 : (x > 7 ? 7 : x); }();
  return v;
}
static int nodes_lds_fwd(const cinema_siren_desc *d, int nst, int lb_total) {
  return nst * 16 * 64 * ksteps_h(d->HP) + nst * 16 * (64 + 16) + 4 * (lb_total + 4 * d->HP);
}
static int nodes_lds_bwd(const cinema_siren_desc *d, int nst) {
  const int xrow = 64 * ksteps_h(d->HP);
  const int top = 16 * d->HP * 4 + nst * 16 * 64;            // final_linear rows + dOut tile: alias the g tile when it has no pad columns
  return nst * 16 * xrow + nst * 16 * (64 + 16) + 16 * d->HP + (xrow == d->HP * 2 && top <= nst * 16 * xrow ? 0 : top);
}

template <int BIG, bool KEEP>
static int launch_nodes_fwd_k(const NodesFwdArgs &a, int grid, int smem, hipStream_t s) {
  CINEMA_SMEM_ATTR((k_nodes_fwd<BIG, KEEP>), smem);
  hipLaunchKernelGGL((k_nodes_fwd<BIG, KEEP>), dim3((unsigned)grid), dim3(512), smem, s, a);
  CINEMA_CHECK_HIP(hipGetLastError());
  return 0;
}
template <int BIG>
static int launch_nodes_fwd(const NodesFwdArgs &a, int grid, int smem, hipStream_t s) {
  return a.saved != nullptr ? launch_nodes_fwd_k<BIG, true>(a, grid, smem, s) : launch_nodes_fwd_k<BIG, false>(a, grid, smem, s);
}
template <int BIG>
static int launch_nodes_bwd(const NodesBwdArgs &a, int grid, int smem, hipStream_t s) {
  CINEMA_SMEM_ATTR(k_nodes_bwd<BIG>, smem);
  hipLaunchKernelGGL(k_nodes_bwd<BIG>, dim3((unsigned)grid), dim3(512), smem, s, a);
  CINEMA_CHECK_HIP(hipGetLastError());
  return 0;
}

static int fill_umod(const cinema_siren_desc *d, int *umod) {
  int nm = 0;
  for (int i = 0; i < d->n_layers; ++i) umod[i] = ((d->modulated_mask >> i) & 1u) ? nm++ : 0;
  return nm;
}

int nodes_forward(const cinema_siren_desc *d, const float *params, const void *packed, const float *coords_t, const void *wrows,
                  const void *tiles, const void *U, const void *V, int64_t S, int64_t N, int64_t Npad, void *saved, float *out,
                  int round_s, hipStream_t s) {
  CINEMA_REQUIRE(fused_eligible(d) && d->pk_ns >= 0, "nodes_forward: descriptor not planned for the activation-stationary path");
  CINEMA_REQUIRE(Npad % 64 == 0, "nodes_forward: Npad %ld not a multiple of 64", (long)Npad);
  NodesFwdArgs a = {};
  a.stream = (const bf16_t *)packed + d->pk_ns;
  a.slab_elems = (long)d->ns_steps * FUSED_STEP_ELEMS;
  const int nm = fill_umod(d, a.umod);
  (void)nm;
  a.U = (const char *)U; a.u_slab_bytes = (long)S * 8192; a.u_layer_bytes = (long)(d->HP / 64) * a.u_slab_bytes;
  a.V = (const char *)V; a.S = (int)S;
  for (int i = 0; i < d->n_layers; ++i) {
    a.sv_y[i] = d->sv_y[i]; a.sv_c[i] = d->sv_c[i]; a.sv_z[i] = d->sv_z[i]; a.sv_s[i] = d->sv_s[i];
    a.lb_b[i] = a.lb_total; a.lb_total += d->HP;
    if ((d->modulated_mask >> i) & 1u) a.lb_total += 2 * d->HP;   // the side block keeps fused.hip's layout (bl unused here)
  }
  a.side = (const float *)((const bf16_t *)packed + d->pk_side);
  a.mod_mask = d->modulated_mask; a.n_layers = d->n_layers; a.in_dim = d->in_dim;
  a.H = d->hidden; a.HP = d->HP; a.LP = d->LP; a.out_dim = d->out_dim;
  a.om_first = d->omega_first; a.om_hidden = d->omega_hidden;
  a.coords_t = coords_t; a.wrows = (const f16_t *)wrows;
  a.tiles = (const int4 *)((const int *)tiles + 4); a.n_tiles = (const int *)tiles;
  a.saved = (bf16_t *)saved; a.slot = (long)Npad * d->HP; a.round_s = round_s;
  a.wf = (const bf16_t *)packed + d->pk_wf; a.bf = params + d->off_bf; a.out = out;
  a.xsteps = ksteps_h(d->HP); a.xrow = 64 * a.xsteps;
  int big, g_small;
  (void)N;
  nodes_launch_plan(Npad, S, nodes_big_default(), &big, &a.g_big, &g_small);
  a.nst_big = big;
  const int smem = nodes_lds_fwd(d, big, a.lb_total);
  CINEMA_REQUIRE(smem <= 160 * 1024, "nodes_forward: LDS budget exceeded (%d bytes, tiles of %d)", smem, big);
  const int grid = a.g_big + g_small;
  switch (big) {
    case 4: return launch_nodes_fwd<4>(a, grid, smem, s);
    case 5: return launch_nodes_fwd<5>(a, grid, smem, s);
    case 6: return launch_nodes_fwd<6>(a, grid, smem, s);
    default: return launch_nodes_fwd<7>(a, grid, smem, s);
  }
}

int nodes_backward_chain(const cinema_siren_desc *d, const void *packed, const float *coords_t, const void *wrows,
                         const void *tiles, const void *U, int64_t S, const void *saved, const float *dout, int64_t N,
                         int64_t Npad, void *gbuf, float *gp, float *dcoords, bool wgrads, int round_s, hipStream_t s,
                         void *dout16, float *dbf) {
  CINEMA_REQUIRE(fused_eligible(d) && d->pk_nb >= 0, "nodes_backward_chain: descriptor not planned for the activation-stationary path");
  CINEMA_REQUIRE(Npad % 64 == 0, "nodes_backward_chain: Npad %ld not a multiple of 64", (long)Npad);
  NodesBwdArgs a = {};
  a.stream = (const bf16_t *)packed + d->pk_nb;
  a.slab_elems = (long)d->nb_steps * FUSED_STEP_ELEMS;
  fill_umod(d, a.umod);
  a.U = (const char *)U; a.u_slab_bytes = (long)S * 8192; a.u_layer_bytes = (long)(d->HP / 64) * a.u_slab_bytes; a.S = (int)S;
  int bias_floats = 0;
  for (int i = 0; i < d->n_layers; ++i) {
    a.sv_c[i] = d->sv_c[i]; a.sv_z[i] = d->sv_z[i]; a.sv_s[i] = d->sv_s[i];
    a.gb_z[i] = d->gb_z[i]; a.gb_s[i] = d->gb_s[i]; a.gb_t[i] = d->gb_t[i];
    a.off_b[i] = d->off_b[i]; a.off_bl[i] = d->off_bl[i];
    bias_floats += d->HP * (((d->modulated_mask >> i) & 1u) ? 3 : 1);
  }
  a.side_w0 = (const float *)((const bf16_t *)packed + d->pk_side) + bias_floats;
  a.dout = dout; a.coords_t = coords_t; a.wrows = (const f16_t *)wrows;
  a.tiles = (const int4 *)((const int *)tiles + 4); a.n_tiles = (const int *)tiles;
  (void)N;
  a.saved = (const bf16_t *)saved; a.slot = (long)Npad * d->HP; a.gbuf = (bf16_t *)gbuf; a.round_s = round_s;
  a.off_w0 = d->off_w[0];
  a.gp = gp; a.wgrads = wgrads ? 1 : 0; a.dcoords = dcoords;
  a.mod_mask = d->modulated_mask; a.n_layers = d->n_layers; a.in_dim = d->in_dim; a.H = d->hidden; a.HP = d->HP;
  a.out_dim = d->out_dim; a.xsteps = ksteps_h(d->HP); a.xrow = 64 * a.xsteps;
  a.om_first = d->omega_first; a.om_hidden = d->omega_hidden;
  int big, g_small;
  nodes_launch_plan(Npad, S, nodes_big_default(), &big, &a.g_big, &g_small);
  a.nst_big = big;
  const int smem = nodes_lds_bwd(d, big);
  CINEMA_REQUIRE(smem <= 160 * 1024, "nodes_backward_chain: LDS budget exceeded (%d bytes, tiles of %d)", smem, big);
  a.g_chain = a.g_big + g_small;
  if (dout16 != nullptr) {
    a.fin = FinalBwdArgs{dout, d->out_dim, dbf, (int)Npad};
    a.dout16 = (bf16_t *)dout16;
    a.g_fin = (int)std::min<int64_t>((Npad + 511) / 512, 128);
  }
  const int grid = a.g_chain + a.g_fin;
  switch (big) {
    case 4: return launch_nodes_bwd<4>(a, grid, smem, s);
    case 5: return launch_nodes_bwd<5>(a, grid, smem, s);
    case 6: return launch_nodes_bwd<6>(a, grid, smem, s);
    default: return launch_nodes_bwd<7>(a, grid, smem, s);
  }
}
