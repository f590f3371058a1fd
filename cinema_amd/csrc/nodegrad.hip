// Node-space backward of the latent side (see nodegrad.h): interpolation-weight blocks, centred node projections, the
// one-pass reduction kernel k_node_grad and the tail of the backward.
// Reference arithmetic: models/inr_decoder.py:64-85 (transform, [idcs_df] gather, grid_sample), models/siren.py:32
// (linear_lats), :61 (final_linear on cat[x, latent]); utils.py:126-166 (embed2affine) for the embedding gradient.
#include "nodegrad.h"
#include "nodes.h"
#include "gemm_engine.h"
#include <algorithm>

namespace {

#include "k1_device.h"
#include "node_device.h"
#include "step_device.h"
#include "dw_device.h"

typedef short s16x4_t __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

// ---- interpolation weight of one grid node as a function of the registered coordinate, and its derivative ----
__device__ __forceinline__ float axis_w(const Axis &a, int i) {
  return (i == a.i0 ? a.w0 : 0.f) + ((i == a.i1 && a.i1 != a.i0) ? a.w1 : 0.f);
}
__device__ __forceinline__ float axis_d(const Axis &a, int i) {     // d axis_w / d u  (times a.dmul = d u / d coordinate)
  return ((i == a.i1) ? 1.f : 0.f) - ((i == a.i0) ? 1.f : 0.f);
}
struct SlotIdx { int ix, iy, iz; bool node; };
__device__ __forceinline__ SlotIdx slot_index(int slot, int lx, int ly, int lz) {
  SlotIdx q;
  q.node = slot < lx * ly * lz;
  q.iz = slot % lz;
  q.iy = (slot / lz) % ly;
  q.ix = slot / (lz * ly);
  return q;
}
// coordinate column 0 walks the LAST grid axis (lz), column 1 ly, column 2 lx (k1_device.h)
__device__ __forceinline__ float slot_weight(float t0, float t1, float t2, int lx, int ly, int lz, int slot) {
  const SlotIdx q = slot_index(slot, lx, ly, lz);
  if (!q.node) return 0.f;
  const Axis az = axis_nodes(t0, lz), ay = axis_nodes(t1, ly), ax = axis_nodes(t2, lx);
  return axis_w(ax, q.ix) * axis_w(ay, q.iy) * axis_w(az, q.iz);
}
__device__ __forceinline__ void slot_jacobian(float t0, float t1, float t2, int lx, int ly, int lz, int slot, float (&j)[3]) {
  const SlotIdx q = slot_index(slot, lx, ly, lz);
  j[0] = j[1] = j[2] = 0.f;
  if (!q.node) return;
  const Axis az = axis_nodes(t0, lz), ay = axis_nodes(t1, ly), ax = axis_nodes(t2, lx);
  const float wx = axis_w(ax, q.ix), wy = axis_w(ay, q.iy), wz = axis_w(az, q.iz);
  j[0] = wx * wy * axis_d(az, q.iz) * az.dmul;
  j[1] = wx * axis_d(ay, q.iy) * ay.dmul * wz;
  j[2] = axis_d(ax, q.ix) * ax.dmul * wy * wz;
}

// rows [lo, hi) of subject s in a batch grouped by subject (idx non-decreasing; ids outside [0, S) clamp as everywhere)
__device__ __forceinline__ int ng_lower(const int32_t *idx, long N, long key, long S) {
  if (key <= 0) return 0;
  if (key >= S) return (int)N;
  long lo = 0, hi = N;
  while (lo < hi) {
    const long mid = (lo + hi) >> 1;
    if (idx[mid] < key) lo = mid + 1; else hi = mid;
  }
  return (int)lo;
}

// ----------------------------------------------------------------------------------------
// wt [Npad/16][32 slots][16 samples] bf16: workgroup = one block of 16 rows, thread = (slot, sample)
// ----------------------------------------------------------------------------------------
struct NgWeightArgs {
  const float *coords; const int32_t *idx; const float *embed; int tf_dim; long N, S; int lx, ly, lz; bf16_t *wt;
  int32_t *offs; int nblk;       // the workgroup behind the nblk row blocks writes offs[0..S]: first row of every subject
};
__global__ __launch_bounds__(512) void k_ng_weights(NgWeightArgs a) {
  __shared__ float ct[16][4];
  const int tid = threadIdx.x;
  const long blk = blockIdx.x;
  if (blk >= a.nblk) {
    for (long t = tid; t <= a.S; t += 512) a.offs[t] = ng_lower(a.idx, a.N, t, a.S);
    return;
  }
  if (tid < 16) {
    const long n = blk * 16 + tid;
    float t0 = 0.f, t1 = 0.f, t2 = 0.f, ok = 0.f;
    if (n < a.N) {
      const float cx = a.coords[n * 3], cy = a.coords[n * 3 + 1], cz = a.coords[n * 3 + 2];
      long s = a.idx[n];
      s = s < 0 ? 0 : (s >= a.S ? a.S - 1 : s);
      t0 = cx; t1 = cy; t2 = cz; ok = 1.f;
      if (a.embed != nullptr) {
        Embed12 ev;
        float (&e)[12] = ev.v;
#pragma unroll
        for (int k = 0; k < 12; ++k) e[k] = k < a.tf_dim ? a.embed[s * a.tf_dim + k] : 0.f;
        const M3 A = affine_from_embed(ev, a.tf_dim);
        t0 = affine_row(A.m[0][0], A.m[0][1], A.m[0][2], e[3], cx, cy, cz);
        t1 = affine_row(A.m[1][0], A.m[1][1], A.m[1][2], e[4], cx, cy, cz);
        t2 = affine_row(A.m[2][0], A.m[2][1], A.m[2][2], e[5], cx, cy, cz);
      }
    }
    ct[tid][0] = t0; ct[tid][1] = t1; ct[tid][2] = t2; ct[tid][3] = ok;
  }
  __syncthreads();
  const int slot = tid >> 4, i = tid & 15;
  float w = 0.f;
  if (ct[i][3] != 0.f) w = slot == NG_ONE_SLOT ? 1.f : slot_weight(ct[i][0], ct[i][1], ct[i][2], a.lx, a.ly, a.lz, slot);
  a.wt[(blk * NG_SLOTS + slot) * 16 + i] = (bf16_t)w;
}

__global__ __launch_bounds__(256) void k_ng_rows(NodeRowArgs a) { node_rows_body(a, blockIdx.x, gridDim.x); }

// ----------------------------------------------------------------------------------------
// U'[s][slot][f] = sum_c Wl[f][c] * (table[s][c][slot] - mean_slot table[s][c][.])   and V' through final_linear[:, H:].
// The mean over the G nodes is taken out in fp32 BEFORE the rows are rounded to bf16: d coords only sees differences
// between the nodes of a cell (the derivative weights of a sample sum to zero), so the rounding error must scale with
// the spread of a subject's nodes, not with their common part.  Conditions are per subject: their centred rows are zero.
// Workgroup (4 waves) = (subject, modulated layer, quarter of the 2 HP features); C[m = feature][n = slot].
// ----------------------------------------------------------------------------------------
struct NgProjArgs {
  const float *latents; long S; int C, G, LP, HP, F, n_mod;
  const bf16_t *packed; long pk_wls[CINEMA_MAX_LAYERS], pk_wlt[CINEMA_MAX_LAYERS]; long pk_wf;
  int layer_of[CINEMA_MAX_LAYERS];
  bf16_t *Up, *Vp;
};
#define NG_PROJ_Q 4
__global__ __launch_bounds__(256) void k_ng_project(NgProjArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  lds_char *tc = (lds_char *)smem;                       // [32 slots][LP] bf16, 16-B chunk c of row r at c ^ (r & 7)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 15, g = lane >> 4;
  int b = blockIdx.x;
  const int quarter = b % NG_PROJ_Q; b /= NG_PROJ_Q;
  const int jm = b % a.n_mod;
  const long s = b / a.n_mod;
  const int rowb = a.LP * 2;
  for (int c = tid; c < a.LP; c += 256) {
    float v[NG_SLOTS];
    float mean = 0.f;
#pragma unroll
    for (int k = 0; k < NG_SLOTS; ++k) {
      v[k] = (c < a.C && k < a.G) ? a.latents[(s * a.C + c) * a.G + k] : 0.f;
      mean += v[k];
    }
    mean /= (float)a.G;
#pragma unroll
    for (int k = 0; k < NG_SLOTS; ++k) {
      const float x = (c < a.C && k < a.G) ? v[k] - mean : 0.f;
      *(LDS_AS bf16_t *)(tc + k * rowb + ((((c >> 3) ^ (k & 7)) << 4) | ((c & 7) * 2))) = (bf16_t)x;
    }
  }
  __syncthreads();
  const int ksteps = a.LP >> 5;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  const int layer = a.layer_of[jm];
  // features of this workgroup: [quarter * HP/2, (quarter + 1) * HP/2) of the 2 HP (scale half, then shift half)
  const int fq = a.HP >> 1, nmt = fq >> 4;                // m-tiles of 16 features in the quarter
  for (int mt = wave; mt < nmt; mt += 4) {
    const int f = quarter * fq + mt * 16;                 // first feature of the tile in [0, 2 HP)
    const int half = f >= a.HP ? 1 : 0, hf = f - half * a.HP;
    const bf16_t *wrow = a.packed + (half ? a.pk_wlt[layer] : a.pk_wls[layer]) + (long)(hf + i) * a.LP + 8 * g;
    f32x4 acc[2] = {zero4, zero4};
#pragma unroll 4
    for (int kk = 0; kk < ksteps; ++kk) {
      const bf16x8 af = *reinterpret_cast<const bf16x8 *>(wrow + 32 * kk);
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const int r = 16 * nt + i;
        const bf16x8 bfr = *(const LDS_AS bf16x8 *)(tc + r * rowb + (((4 * kk + g) ^ (r & 7)) << 4));
        acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bfr, acc[nt], 0, 0, 0);
      }
    }
    // lane (i = slot, g): features f + 4g .. + 3
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      bf16x4 o = {(bf16_t)acc[nt][0], (bf16_t)acc[nt][1], (bf16_t)acc[nt][2], (bf16_t)acc[nt][3]};
      *reinterpret_cast<bf16x4 *>(a.Up + (s * NG_SLOTS + 16 * nt + i) * a.F + (long)jm * 2 * a.HP + f + 4 * g) = o;
    }
  }
  if (jm == 0 && quarter == 0 && wave == 3) {
    // V'[slot][o] = sum_c Wf[o][H + c] * Tc[slot][c]   (padded [16][HP + LP] copy of final_linear)
    const bf16_t *wrow = a.packed + a.pk_wf + (long)i * (a.HP + a.LP) + a.HP + 8 * g;
    f32x4 acc[2] = {zero4, zero4};
    for (int kk = 0; kk < ksteps; ++kk) {
      const bf16x8 af = *reinterpret_cast<const bf16x8 *>(wrow + 32 * kk);
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const int r = 16 * nt + i;
        const bf16x8 bfr = *(const LDS_AS bf16x8 *)(tc + r * rowb + (((4 * kk + g) ^ (r & 7)) << 4));
        acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bfr, acc[nt], 0, 0, 0);
      }
    }
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      bf16x4 o = {(bf16_t)acc[nt][0], (bf16_t)acc[nt][1], (bf16_t)acc[nt][2], (bf16_t)acc[nt][3]};
      *reinterpret_cast<bf16x4 *>(a.Vp + (s * NG_SLOTS + 16 * nt + i) * 16 + 4 * g) = o;
    }
  }
}

// ----------------------------------------------------------------------------------------
// k_node_grad.  Workgroup (8 waves) = (subject s, tensor t) with t = 2 * modulated ordinal + half (ds, dt), plus one
// workgroup per subject for dOut.  It walks the subject's rows in blocks of 16 and reads every X row once, straight into
// registers in the MFMA operand layout (lane (i, g): row i, 8 features 8g..8g+7 of a 32-feature block) -- that register
// is the B operand of   dwn[slot][row] += U'[slot][f] X[row][f]   and, as the A operand against a 0/1 selector, leaves
// X^T in the accumulator layout (lane = feature, 4 consecutive rows), which after one conversion is the A operand of
//   dU[f][slot] += X^T[f][row] w[row][slot]   (k = 16 rows).  No LDS staging, no transposing reads.
// The partial dwn of the 8 waves (each owns features) meet in LDS once per block; there each wave takes 4 slots, applies
// d w / d coords and leaves 3 numbers per row, summed in wave order (deterministic).
// ----------------------------------------------------------------------------------------
struct NgGradArgs {
  const bf16_t *gbuf; long slot_elems; int gb_s[CINEMA_MAX_LAYERS], gb_t[CINEMA_MAX_LAYERS], layer_of[CINEMA_MAX_LAYERS];
  int n_mod, dout_slot, ncg;
  const bf16_t *wt, *Up, *Vp; const float *coords_t; const int32_t *offs;
  long N, Npad, S; int HP, F, lx, ly, lz, in_dim;
  bf16_t *dU, *dUo16; float *dUo32, *dcn;
  int dbg;      // developer ablations (cinema_debug_node_grad): 1 = no exchange / d coords, 2 = no dU products, 4 = no dwn products
};

__device__ __forceinline__ bf16x4 cvt4(const f32x4 &v) {
  bf16x4 o = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
  return o;
}
__device__ __forceinline__ f32x4 mfma16(const bf16x4 &a, const bf16x4 &b, const f32x4 &c) {
  return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4_t, a), __builtin_bit_cast(s16x4_t, b), c, 0, 0, 0);
}
// the 4 weights (rows 16b + 4g .. + 3) of slot 16 nt + i; ng_wmask zeroes the rows outside [beg, end) -- at the point of
// USE: touching the value where it is loaded would make the load wait there and undo the prefetch
__device__ __forceinline__ bf16x4 ng_wload(const bf16_t *wt, long b, int nt, int i, int g) {
  return *reinterpret_cast<const bf16x4 *>(wt + ((b * NG_SLOTS + 16 * nt + i) * 16 + 4 * g));
}
__device__ __forceinline__ bf16x4 ng_wmask(bf16x4 w, long b, int g, int beg, int end) {
  if (16 * b < beg || 16 * b + 16 > end) {          // workgroup-uniform: only a subject's first and last block
    const int n0 = (int)(16 * b) + 4 * g;
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (n0 + r < beg || n0 + r >= end) w[r] = (bf16_t)0.f;
  }
  return w;
}

// One tensor t = (modulated layer, half) and one group cg of 8 x 32 columns: wave w owns the 32-column block 8 cg + w.
// Loads run three row blocks ahead through a ring of four register sets with STATIC slots (a rotating copy would
// make every load wait at the copy); one barrier per row block (P, R and the axis table are double-buffered, the last
// sum of a block is taken one block later).  The interpolation axes of the block's 16 rows (3 per row: nodes, weights,
// d u / d coordinate) are worked out once, by 48 lanes of wave 1, and shared through LDS.
struct NgAxis { int nodes; float w0, w1, dmul; };      // nodes = i0 | i1 << 8
__device__ __forceinline__ float ngx_w(const NgAxis &a, int i) {
  const int i0 = a.nodes & 255, i1 = a.nodes >> 8;
  return (i == i0 ? a.w0 : 0.f) + ((i == i1 && i1 != i0) ? a.w1 : 0.f);
}
__device__ __forceinline__ float ngx_d(const NgAxis &a, int i) {
  const int i0 = a.nodes & 255, i1 = a.nodes >> 8;
  return (((i == i1) ? 1.f : 0.f) - ((i == i0) ? 1.f : 0.f)) * a.dmul;
}

__device__ __forceinline__ void ng_body_tensor(const NgGradArgs &a, const long s, const int t, const int cg, const int part,
                                               float (*P)[8][64][8], float (*R)[8][16][4], NgAxis (*AX)[16][4], const int beg,
                                               const int end) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 15, g = lane >> 4;
  const int jm = t >> 1, half = t & 1, layer = a.layer_of[jm];
  const bf16_t *X = a.gbuf + a.slot_elems * (half ? a.gb_t[layer] : a.gb_s[layer]);
  const int nkb = a.HP >> 5;
  const bool kv = 8 * cg + wave < nkb;                       // wave-uniform: this wave's column block exists
  const int kb = kv ? 8 * cg + wave : 0;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  bf16x8 sel[2];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    sel[0][e] = (bf16_t)((8 * g + e == i) ? 1.f : 0.f);
    sel[1][e] = (bf16_t)((8 * g + e == 16 + i) ? 1.f : 0.f);
  }
  bf16x8 up[2];
  f32x4 dUa[2][2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    up[mt] = *reinterpret_cast<const bf16x8 *>(a.Up + (s * NG_SLOTS + 16 * mt + i) * a.F + (long)t * a.HP + 32 * kb + 8 * g);
    dUa[0][mt] = zero4; dUa[1][mt] = zero4;
  }
  const long b0 = beg >> 4, b1 = ((long)end + 15) >> 4;
#ifndef NG_RING
#define NG_RING 4
#endif
  bf16x8 xr[NG_RING];
  bf16x4 wr[NG_RING][2];
  float cr[NG_RING];                  // wave 1, lane (i, g < 3): coordinate column g of row i
  const bool ax_lane = wave == 1 && g < 3;
  auto load = [&](long b, bf16x8 &x, bf16x4 (&w)[2], float &c) {
    const long bb = b < b1 ? b : b1 - 1;
    if (ax_lane) c = a.coords_t[(bb * 16 + i) * 3 + g];
    x = __builtin_nontemporal_load(reinterpret_cast<const bf16x8 *>(X + (bb * 16 + i) * a.HP + 32 * kb + 8 * g));
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) w[nt] = ng_wload(a.wt, bb, nt, i, g);
  };
#pragma unroll
  for (int q = 0; q < NG_RING - 1; ++q) load(b0 + q, xr[q], wr[q], cr[q]);
  // the slot this lane finishes after the exchange: wave -> (slot tile, lane group), lane group g -> accumulator register
  const int my_slot = 16 * (wave >> 2) + 4 * (wave & 3) + g;
  const SlotIdx sq = slot_index(my_slot, a.lx, a.ly, a.lz);
  const int ax_size = g == 0 ? a.lz : (g == 1 ? a.ly : a.lx);      // coordinate column 0 walks lz, 1 ly, 2 lx
  auto finish = [&](long b) {                                 // wave 0: the 3 numbers of every row of block b
    if (wave == 0 && g < 3) {
      float t3 = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) t3 += R[b & 1][w][i][g];
      const long n = b * 16 + i;
      if (n >= beg && n < end) a.dcn[((long)part * a.Npad + n) * 4 + g] = t3;
    }
  };
#define NG_STEP(J)                                                                                                  \
  {                                                                                                                 \
    const long b = bb + (J);                                                                                        \
    if (b < b1) {                                                                                                   \
      load(b + NG_RING - 1, xr[((J) + NG_RING - 1) % NG_RING], wr[((J) + NG_RING - 1) % NG_RING],                   \
           cr[((J) + NG_RING - 1) % NG_RING]);                                                                      \
      f32x4 cb[2] = {zero4, zero4};                                                                                 \
      if (kv) {                                                                                                     \
        if (!(a.dbg & 4)) _Pragma("unroll") for (int mt = 0; mt < 2; ++mt)                                          \
          cb[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(up[mt], xr[J], cb[mt], 0, 0, 0);                         \
        if (!(a.dbg & 2)) _Pragma("unroll") for (int ft = 0; ft < 2; ++ft) {                                        \
          const bf16x4 xt = cvt4(__builtin_amdgcn_mfma_f32_16x16x32_bf16(xr[J], sel[ft], zero4, 0, 0, 0));          \
          _Pragma("unroll") for (int nt = 0; nt < 2; ++nt)                                                          \
            dUa[ft][nt] = mfma16(xt, ng_wmask(wr[J][nt], b, g, beg, end), dUa[ft][nt]);                            \
        }                                                                                                           \
      }                                                                                                             \
      if (a.dbg & 1) { dUa[0][0] += cb[0]; dUa[0][1] += cb[1]; } else {                                             \
      if (ax_lane) {                                                                                                \
        const Axis ax = axis_nodes(cr[J], ax_size);                                                                 \
        NgAxis q = {ax.i0 | (ax.i1 << 8), ax.w0, ax.w1, ax.dmul};                                                   \
        AX[b & 1][i][g] = q;                                                                                        \
      }                                                                                                             \
      {                                                                                                             \
        f32x4 *pp = reinterpret_cast<f32x4 *>(&P[b & 1][wave][lane][0]);                                            \
        pp[0] = cb[0]; pp[1] = cb[1];                                                                               \
      }                                                                                                             \
      __syncthreads();                                                                                              \
      if (b > b0) finish(b - 1);                                                                                    \
      float v = 0.f;                                                                                                \
      _Pragma("unroll") for (int w = 0; w < 8; ++w) v += P[b & 1][w][(wave & 3) * 16 + i][(wave >> 2) * 4 + g];     \
      const NgAxis qz = AX[b & 1][i][0], qy = AX[b & 1][i][1], qx = AX[b & 1][i][2];                                \
      const float wx = ngx_w(qx, sq.ix), wy = ngx_w(qy, sq.iy), wz = ngx_w(qz, sq.iz);                              \
      if (!sq.node) v = 0.f;                                                                                        \
      float jc[3] = {v * wx * wy * ngx_d(qz, sq.iz), v * wx * ngx_d(qy, sq.iy) * wz, v * ngx_d(qx, sq.ix) * wy * wz}; \
      _Pragma("unroll") for (int k = 0; k < 3; ++k) {                                                               \
        float p = jc[k];                                                                                            \
        p += __shfl_xor(p, 16, 64);                                                                                 \
        p += __shfl_xor(p, 32, 64);                                                                                 \
        if (g == 0) R[b & 1][wave][i][k] = p;                                                                       \
      }                                                                                                             \
      }                                                                                                             \
    }                                                                                                               \
  }
#if NG_RING == 4
  for (long bb = b0; bb < b1; bb += 4) { NG_STEP(0) NG_STEP(1) NG_STEP(2) NG_STEP(3) }
#else
  for (long bb = b0; bb < b1; bb += 8) { NG_STEP(0) NG_STEP(1) NG_STEP(2) NG_STEP(3) NG_STEP(4) NG_STEP(5) NG_STEP(6) NG_STEP(7) }
#endif
#undef NG_STEP
  __syncthreads();
  finish(b1 - 1);
  // dU rows of this subject: lane (i = slot, g): features 32 kb + 16 ft + 4 g .. + 3
  if (kv) {
#pragma unroll
    for (int ft = 0; ft < 2; ++ft)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
        *reinterpret_cast<bf16x4 *>(a.dU + (s * NG_SLOTS + 16 * nt + i) * a.F + (long)t * a.HP + 32 * kb + 16 * ft + 4 * g) =
            cvt4(dUa[ft][nt]);
  }
}

// dOut part: every wave takes every 8th block with all 16 columns (no exchange); the dUo partials meet in LDS at the end
__device__ __forceinline__ void ng_body_dout(const NgGradArgs &a, const long s, const int part, float (*P)[64][8], const int beg,
                                             const int end) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 15, g = lane >> 4;
  const bf16_t *X = a.gbuf + a.slot_elems * a.dout_slot;      // [Npad][16] bf16
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
  bf16x8 sel0;
#pragma unroll
  for (int e = 0; e < 8; ++e) sel0[e] = (bf16_t)((8 * g + e == i) ? 1.f : 0.f);
  bf16x8 vp[2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
    vp[mt] = g < 2 ? *reinterpret_cast<const bf16x8 *>(a.Vp + (s * NG_SLOTS + 16 * mt + i) * 16 + 8 * g) : zero8;
  f32x4 acc[2] = {zero4, zero4};
  const long b0 = beg >> 4, b1 = ((long)end + 15) >> 4;
  for (long b = b0 + wave; b < b1; b += 8) {
    const bf16x8 x = g < 2 ? *reinterpret_cast<const bf16x8 *>(X + (b * 16 + i) * 16 + 8 * g) : zero8;
    bf16x4 w[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) w[nt] = ng_wmask(ng_wload(a.wt, b, nt, i, g), b, g, beg, end);
    const float *cp = a.coords_t + (b * 16 + i) * a.in_dim;
    const float t0 = cp[0], t1 = a.in_dim > 1 ? cp[1] : 0.f, t2 = a.in_dim > 2 ? cp[2] : 0.f;
    const bf16x4 xt = cvt4(__builtin_amdgcn_mfma_f32_16x16x32_bf16(x, sel0, zero4, 0, 0, 0));
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) acc[nt] = mfma16(xt, w[nt], acc[nt]);
    float dc[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const f32x4 cb = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vp[mt], x, zero4, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float jc[3];
        slot_jacobian(t0, t1, t2, a.lx, a.ly, a.lz, 16 * mt + 4 * g + r, jc);
#pragma unroll
        for (int k = 0; k < 3; ++k) dc[k] += cb[r] * jc[k];
      }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      dc[k] += __shfl_xor(dc[k], 16, 64);
      dc[k] += __shfl_xor(dc[k], 32, 64);
    }
    const long n = b * 16 + i;
    if (g == 0 && n >= beg && n < end) {
#pragma unroll
      for (int k = 0; k < 3; ++k) a.dcn[((long)part * a.Npad + n) * 4 + k] = dc[k];
    }
  }
  {
    f32x4 *pp = reinterpret_cast<f32x4 *>(&P[wave][lane][0]);
    pp[0] = acc[0]; pp[1] = acc[1];
  }
  __syncthreads();
  if (wave == 0) {
    // lane (i = slot, g): outputs o = 4g + r
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      f32x4 t4 = zero4;
#pragma unroll
      for (int w = 0; w < 8; ++w) t4 += *reinterpret_cast<const f32x4 *>(&P[w][lane][4 * nt]);
      const long row = s * NG_SLOTS + 16 * nt + i;
      *reinterpret_cast<f32x4 *>(a.dUo32 + row * 16 + 4 * g) = t4;
      *reinterpret_cast<bf16x4 *>(a.dUo16 + row * 16 + 4 * g) = cvt4(t4);
    }
  }
}

__global__ __launch_bounds__(512, 4) void k_node_grad(NgGradArgs a) {
  __shared__ __attribute__((aligned(16))) float P[2][8][64][8];
  __shared__ float R[2][8][16][4];
  __shared__ __attribute__((aligned(16))) NgAxis AX[2][16][4];
  const int per = 2 * a.n_mod * a.ncg + 1;
  const long s = blockIdx.x / per;
  const int part = blockIdx.x - (int)s * per;
  const int beg = a.offs[s], end = a.offs[s + 1];
  if (part < per - 1) {
    const int t = part / a.ncg, cg = part - t * a.ncg;
    if (beg >= end) {                         // subject without samples in this batch: its dU rows are zero
      const int cols = 256 < a.HP - 256 * cg ? 256 : a.HP - 256 * cg;
      for (int q = threadIdx.x; q < NG_SLOTS * (cols >> 3); q += 512) {
        const int slot = q / (cols >> 3), ch = q - slot * (cols >> 3);
        const u32x4 z = {0u, 0u, 0u, 0u};
        *reinterpret_cast<u32x4 *>(a.dU + (s * NG_SLOTS + slot) * a.F + (long)t * a.HP + 256 * cg + ch * 8) = z;
      }
      return;
    }
    ng_body_tensor(a, s, t, cg, part, P, R, AX, beg, end);
  } else {
    ng_body_dout(a, s, part, P[0], beg, end);
  }
}

// ----------------------------------------------------------------------------------------
// tail of the backward: block ranges of one launch
// ----------------------------------------------------------------------------------------
struct NgTailArgs {
  Dw256Args red; int g_red;
  // linear_lats.bias gradients: column sums of the constant-1 slot rows of dU
  const bf16_t *dU; float *gp; long off_bl[CINEMA_MAX_LAYERS]; int layer_of[CINEMA_MAX_LAYERS]; int n_mod, H, HP, F, g_bias;
  // d table / d cond from the dT parts
  const float *dT; long part_stride; int nseg, LP, C, G, cond_dims; float *dlatents, *dcond; int g_tab, tab_cs;
  // embedding-table gradient
  const float *coords, *dcoords, *dcn, *embed; float *dembed; const int32_t *offs; long N, Npad, S; int tf_dim, nparts, g_tf;
};

__device__ __forceinline__ void ng_tail_bias(const NgTailArgs &a, int vb) {
  const int col = vb * 256 + threadIdx.x;             // column of dU: (jm, half, feature)
  if (col >= a.F) return;
  const int t = col / a.HP, f = col - t * a.HP;
  if (f >= a.H) return;
  float acc = 0.f;
  for (long s = 0; s < a.S; ++s) acc += (float)a.dU[(s * NG_SLOTS + NG_ONE_SLOT) * a.F + col];
  a.gp[a.off_bl[a.layer_of[t >> 1]] + (long)(t & 1) * a.H + f] = acc;
}

// workgroup = (subject, slice of 64 columns of the node rows): dlatents[s][c][k] = sum over the parts of dT[s*32 + k][c]
__device__ __forceinline__ void ng_tail_table(const NgTailArgs &a, int vb) {
  __shared__ float tile[NG_SLOTS][65];
  const long s = vb / a.tab_cs;
  const int c0 = (vb - (int)s * a.tab_cs) * 64;
  const int tid = threadIdx.x;
  for (int q = tid; q < NG_SLOTS * 64; q += 256) {
    const int k = q >> 6, cc = q & 63;
    float v = 0.f;
    if (k < a.G && c0 + cc < a.LP)
      for (int p = 0; p < a.nseg; ++p) v += a.dT[(long)p * a.part_stride + (s * NG_SLOTS + k) * a.LP + c0 + cc];
    tile[k][cc] = v;
  }
  __syncthreads();
  for (int q = tid; q < 64 * a.G; q += 256) {
    const int cc = q / a.G, k = q - cc * a.G;
    if (c0 + cc < a.C) a.dlatents[(s * a.C + c0 + cc) * a.G + k] = tile[k][cc];
  }
  if (a.dcond != nullptr && tid < 64) {
    const int j = c0 + tid - a.C;
    if (j >= 0 && j < a.cond_dims) {
      float v = 0.f;
      for (int k = 0; k < a.G; ++k) v += tile[k][tid];
      a.dcond[s * a.cond_dims + j] = v;
    }
  }
}

// one workgroup per subject: sums of d c^T and d over its rows (d = the chain's d coords + the interpolation parts), then
// one evaluation of the embedding gradient (embed_grad is linear in them)
__device__ __forceinline__ void ng_tail_tf(const NgTailArgs &a, int vb) {
  __shared__ float red[4][12];
  const long s = vb;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int rng[2] = {a.offs[s], a.offs[s + 1]};
  float acc[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) acc[k] = 0.f;
  for (int n = rng[0] + tid; n < rng[1]; n += 256) {
    float d[3] = {a.dcoords[(long)n * 3], a.dcoords[(long)n * 3 + 1], a.dcoords[(long)n * 3 + 2]};
    for (int p = 0; p < a.nparts; ++p) {
      const f32x4 q = *reinterpret_cast<const f32x4 *>(a.dcn + ((long)p * a.Npad + n) * 4);
      d[0] += q[0]; d[1] += q[1]; d[2] += q[2];
    }
    const float c[3] = {a.coords[(long)n * 3], a.coords[(long)n * 3 + 1], a.coords[(long)n * 3 + 2]};
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
      for (int j = 0; j < 3; ++j) acc[r * 3 + j] += d[r] * c[j];
      acc[9 + r] += d[r];
    }
  }
#pragma unroll
  for (int k = 0; k < 12; ++k) {
    const float v = wave_sum(acc[k]);
    if (lane == 0) red[wave][k] = v;
  }
  __syncthreads();
  if (tid == 0) {
    M3 G;
    float d[3], e[12], g[12], t[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) t[k] = red[0][k] + red[1][k] + red[2][k] + red[3][k];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
      for (int j = 0; j < 3; ++j) G.m[r][j] = t[r * 3 + j];
      d[r] = t[9 + r];
    }
    for (int k = 0; k < 12; ++k) e[k] = k < a.tf_dim ? a.embed[s * a.tf_dim + k] : 0.f;
    embed_grad(e, a.tf_dim, G, d, g);
    for (int k = 0; k < a.tf_dim; ++k) a.dembed[s * a.tf_dim + k] = g[k];
  }
}

__global__ __launch_bounds__(256) void k_ng_tail(NgTailArgs a) {
  int b = blockIdx.x;
  if (b < a.g_red) { dw256_reduce_body(a.red, b); return; }
  b -= a.g_red;
  if (b < a.g_tab) { ng_tail_table(a, b); return; }
  b -= a.g_tab;
  if (b < a.g_tf) { ng_tail_tf(a, b); return; }
  b -= a.g_tf;
  ng_tail_bias(a, b);
}

inline size_t up256(size_t x) { return (x + 255) / 256 * 256; }

}  // namespace

// ========================================================================================
// host side
// ========================================================================================
bool fused_eligible(const cinema_siren_desc *d);

bool ng_eligible(const cinema_siren_desc *d, int lx, int ly, int lz) {
  if (d->precision != CINEMA_PREC_BF16 || d->pk_fs < 0 || !fused_eligible(d)) return false;
  if (lx < 1 || ly < 1 || lz < 1 || lx * ly * lz >= NG_ONE_SLOT) return false;
  if (d->in_dim != 3) return false;
  return d->modulated_mask != 0;
}

NgPlan ng_plan(const cinema_siren_desc *d, int64_t S, int64_t Npad) {
  NgPlan p = {};
  p.S = (long)S; p.Npad = (long)Npad; p.Rp = (long)((S * NG_SLOTS + 127) / 128 * 128);
  p.HP = d->HP; p.LP = d->LP;
  for (int i = 0; i < d->n_layers; ++i)
    if ((d->modulated_mask >> i) & 1u) p.layer_of[p.n_mod++] = i;
  p.F = p.n_mod * 2 * d->HP;
  p.ncg = (d->HP + 255) / 256;               // groups of 8 x 32 columns per tensor (k_node_grad)
  p.nparts = 2 * p.n_mod * p.ncg + 1;
  size_t o = 0;
  auto take = [&](size_t bytes) { const size_t at = o; o += up256(bytes); return at; };
  p.offs = take((size_t)(S + 1) * 4);
  p.wt = take((size_t)(Npad / 16) * NG_SLOTS * 16 * 2);
  p.T = take((size_t)p.Rp * d->LP * 2);
  p.Up = take((size_t)p.Rp * p.F * 2);
  p.Vp = take((size_t)p.Rp * 16 * 2);
  p.dU = take((size_t)p.Rp * p.F * 2);
  p.dUo16 = take((size_t)p.Rp * 16 * 2);
  p.dUo32 = take((size_t)p.Rp * 16 * 4);
  p.dcn = take((size_t)p.nparts * Npad * 4 * 4);
  p.dT = take((size_t)2 * p.n_mod * p.Rp * d->LP * 4);
  p.bytes = o;
  return p;
}

extern "C" int64_t cinema_nodegrad_work_bytes(const cinema_siren_desc *d, int64_t S, int64_t Npad) {
  if (!d || S < 1 || Npad < 128 || Npad % 128 != 0) {
    cinema_set_error("cinema_nodegrad_work_bytes: need S >= 1 and Npad a positive multiple of 128");
    return -1;
  }
  return (int64_t)ng_plan(d, S, Npad).bytes;
}

extern "C" int cinema_nodegrad_eligible(const cinema_siren_desc *d, int32_t lx, int32_t ly, int32_t lz) {
  return d != nullptr && ng_eligible(d, lx, ly, lz) ? 1 : 0;
}

extern "C" int cinema_nodegrad_prepare(const cinema_siren_desc *d, const void *packed, const float *coords, const int32_t *idx,
                                       const float *embed, int32_t tf_dim, const float *latents, const float *cond_subject,
                                       int64_t N, int64_t Npad, int64_t S, int32_t C, int32_t lx, int32_t ly, int32_t lz,
                                       int32_t cond_dims, void *work, void *stream) {
  CINEMA_REQUIRE(d && packed && coords && idx && latents && work, "cinema_nodegrad_prepare: null argument");
  CINEMA_REQUIRE(ng_eligible(d, lx, ly, lz), "cinema_nodegrad_prepare: needs a bf16 stack on the activation-stationary path with a "
                 "modulated layer and a latent grid of at most %d nodes (got %d x %d x %d)", NG_ONE_SLOT - 1, lx, ly, lz);
  CINEMA_REQUIRE(N >= 1 && Npad >= N && Npad % 128 == 0 && S >= 1, "cinema_nodegrad_prepare: bad sizes N=%ld Npad=%ld S=%ld", (long)N, (long)Npad, (long)S);
  CINEMA_REQUIRE(C >= 1 && C + cond_dims == d->lat, "cinema_nodegrad_prepare: C + cond_dims = %d != lat_size %d", C + cond_dims, d->lat);
  CINEMA_REQUIRE(cond_dims == 0 || cond_subject != nullptr, "cond_dims %d but cond_subject is null", cond_dims);
  CINEMA_REQUIRE(embed == nullptr || tf_dim == 6 || tf_dim == 9 || tf_dim == 12, "tf embedding width %d unsupported (6, 9 or 12)", tf_dim);
  hipStream_t s = (hipStream_t)stream;
  const NgPlan p = ng_plan(d, S, Npad);
  char *w = (char *)work;
  const int G = lx * ly * lz;
  NgWeightArgs wa = {coords, idx, embed, tf_dim, (long)N, (long)S, lx, ly, lz, (bf16_t *)(w + p.wt), (int32_t *)(w + p.offs),
                     (int)(Npad / 16)};
  hipLaunchKernelGGL(k_ng_weights, dim3((unsigned)(Npad / 16 + 1)), dim3(512), 0, s, wa);
  CINEMA_CHECK_HIP(hipGetLastError());
  NodeRowArgs r = {latents, cond_subject, (long)S, C, G, cond_dims, d->LP, (bf16_t *)(w + p.T)};
  const long total = (long)S * NODE_SLOTS * (d->LP / 8);
  hipLaunchKernelGGL(k_ng_rows, dim3((unsigned)std::min<long>((total + 255) / 256, 4096)), dim3(256), 0, s, r);
  CINEMA_CHECK_HIP(hipGetLastError());
  NgProjArgs pa = {};
  pa.latents = latents; pa.S = (long)S; pa.C = C; pa.G = G; pa.LP = d->LP; pa.HP = d->HP; pa.F = p.F; pa.n_mod = p.n_mod;
  pa.packed = (const bf16_t *)packed; pa.pk_wf = d->pk_wf;
  for (int j = 0; j < p.n_mod; ++j) {
    pa.layer_of[j] = p.layer_of[j];
    pa.pk_wls[p.layer_of[j]] = d->pk_wls[p.layer_of[j]];
    pa.pk_wlt[p.layer_of[j]] = d->pk_wlt[p.layer_of[j]];
  }
  pa.Up = (bf16_t *)(w + p.Up); pa.Vp = (bf16_t *)(w + p.Vp);
  const int smem = NG_SLOTS * d->LP * 2;
  CINEMA_REQUIRE(smem <= 64 * 1024, "cinema_nodegrad_prepare: lat_size %d too wide for the LDS staging", d->lat);
  CINEMA_REQUIRE((long)S * p.n_mod * NG_PROJ_Q < (1L << 30), "cinema_nodegrad_prepare: too many subjects");
  CINEMA_SMEM_ATTR(k_ng_project, smem);
  hipLaunchKernelGGL(k_ng_project, dim3((unsigned)(S * p.n_mod * NG_PROJ_Q)), dim3(256), smem, s, pa);
  CINEMA_CHECK_HIP(hipGetLastError());
  return 0;
}

static int g_ng_dbg = 0;
extern "C" int cinema_debug_node_grad(const cinema_siren_desc *d, void *work, const void *gbuf, const float *coords_t, int64_t N,
                                      int64_t Npad, int64_t S, int32_t lx, int32_t ly, int32_t lz, int32_t dbg, void *stream) {
  g_ng_dbg = dbg;       // developer hook (tools/time_nodegrad.py): k_node_grad alone, with ablation bits
  const int rc = ng_launch_grad(d, ng_plan(d, S, Npad), work, gbuf, coords_t, nullptr, N, lx, ly, lz, (hipStream_t)stream);
  g_ng_dbg = 0;
  return rc;
}

int ng_launch_grad(const cinema_siren_desc *d, const NgPlan &p, void *work, const void *gbuf, const float *coords_t,
                   const int32_t *idx, int64_t N, int lx, int ly, int lz, hipStream_t s) {
  char *w = (char *)work;
  (void)idx;
  NgGradArgs a = {};
  a.dbg = g_ng_dbg;
  a.gbuf = (const bf16_t *)gbuf; a.slot_elems = p.Npad * d->HP;
  for (int i = 0; i < d->n_layers; ++i) { a.gb_s[i] = d->gb_s[i]; a.gb_t[i] = d->gb_t[i]; }
  for (int j = 0; j < p.n_mod; ++j) a.layer_of[j] = p.layer_of[j];
  a.n_mod = p.n_mod; a.dout_slot = d->n_gbuf - 1;
  a.wt = (const bf16_t *)(w + p.wt); a.Up = (const bf16_t *)(w + p.Up); a.Vp = (const bf16_t *)(w + p.Vp);
  a.coords_t = coords_t; a.offs = (const int32_t *)(w + p.offs); a.ncg = p.ncg;
  a.N = (long)N; a.Npad = p.Npad; a.S = p.S; a.HP = d->HP; a.F = p.F; a.lx = lx; a.ly = ly; a.lz = lz; a.in_dim = d->in_dim;
  a.dU = (bf16_t *)(w + p.dU); a.dUo16 = (bf16_t *)(w + p.dUo16); a.dUo32 = (float *)(w + p.dUo32); a.dcn = (float *)(w + p.dcn);
  CINEMA_REQUIRE(p.S * p.nparts < (1L << 30), "node gradients: too many subjects");
  hipLaunchKernelGGL(k_node_grad, dim3((unsigned)(p.S * p.nparts)), dim3(512), 0, s, a);
  CINEMA_CHECK_HIP(hipGetLastError());
  return 0;
}

int ng_launch_tail(const cinema_siren_desc *d, const NgPlan &p, void *work, const NgTailArgsHost &t, hipStream_t s) {
  char *w = (char *)work;
  NgTailArgs a = {};
  if (t.reduce != nullptr) {
    a.red = *(const Dw256Args *)t.reduce;
    a.g_red = a.red.g.nprob > 0 ? 32 * a.red.g.tile_start[a.red.g.nprob] : 0;
  }
  a.dU = (const bf16_t *)(w + p.dU); a.n_mod = p.n_mod; a.H = d->hidden; a.HP = d->HP; a.F = p.F;
  for (int j = 0; j < p.n_mod; ++j) a.layer_of[j] = p.layer_of[j];
  for (int i = 0; i < d->n_layers; ++i) a.off_bl[i] = d->off_bl[i];
  a.S = p.S; a.Npad = p.Npad; a.N = (long)t.N;
  if (t.grad_params != nullptr) { a.gp = t.grad_params; a.g_bias = (p.F + 255) / 256; }
  if (t.dlatents != nullptr) {
    a.dT = (const float *)(w + p.dT); a.part_stride = p.Rp * (long)d->LP; a.nseg = 2 * p.n_mod; a.LP = d->LP;
    a.C = t.C; a.G = t.G; a.cond_dims = t.cond_dims; a.dlatents = t.dlatents; a.dcond = t.dcond;
    a.tab_cs = (d->LP + 63) / 64;
    a.g_tab = (int)(p.S * a.tab_cs);
  }
  if (t.dembed != nullptr) {
    a.coords = t.coords; a.dcoords = t.dcoords; a.dcn = (const float *)(w + p.dcn); a.embed = t.embed; a.dembed = t.dembed;
    a.offs = (const int32_t *)(w + p.offs); a.tf_dim = t.tf_dim; a.nparts = p.nparts; a.g_tf = (int)p.S;
  }
  const int grid = a.g_red + a.g_tab + a.g_tf + a.g_bias;
  if (grid == 0) return 0;
  hipLaunchKernelGGL(k_ng_tail, dim3((unsigned)grid), dim3(256), 0, s, a);
  CINEMA_CHECK_HIP(hipGetLastError());
  return 0;
}
