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
#include "noderows_device.h"
#include "ngprep_device.h"
#include "step_device.h"
#include "dw_device.h"

// everything the node-space backward needs before it can start (behind the node rows), in ONE launch (block ranges)
__global__ __launch_bounds__(256) void k_ng_rows(NodeRowsCArgs a) {
  __shared__ __attribute__((aligned(16))) char pool[NODE_ROWS_LDS_BYTES];
  node_rows_centred_body(a, blockIdx.x, pool);
}
struct NgPrepArgs { NgWeightArgs w; NgProjArgs p; int g_w; };
__global__ __launch_bounds__(512) void k_ng_prepare(NgPrepArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  int b = blockIdx.x;
  if (b < a.g_w) { ng_weights_body<512>(a.w, b, (float (*)[4])smem_raw); return; }
  ng_project_body(a.p, b - a.g_w, (lds_char *)smem_raw);
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
  const bf16_t *wt, *wrow, *Up, *Vp; const int32_t *axt; const float *coords_t; const int32_t *offs;
  long N, Npad, S; int HP, F, lx, ly, lz, in_dim;
  bf16_t *dU, *dUo16; float *dUo32, *dcn;
  // d (node rows) parts: part (t, cg) = dU tile x linear_lats^T (packed [LP][HP] copies), last part = dUo x final_linear[:, H:]
  const bf16_t *packed; long pk_wlsT[CINEMA_MAX_LAYERS], pk_wltT[CINEMA_MAX_LAYERS]; const float *wf_lat; int wf_ld, out_dim, lat, LP;
  float *dT; long dT_stride;
  int dbg;      // developer ablations (cinema_debug_node_grad): 1 = no exchange / d coords, 2 = no dU products, 4 = no dwn products
};

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

// One tensor t = (modulated layer, half) and one group cg of 256 columns.  The X rows go global -> LDS by LDS-DMA in
// tiles of 32 rows x 512 B (whole 512-B row pieces per wave-instruction: coalesced, no staging registers), three tiles in
// a ring, two in flight; the tile's interpolation weights and axes ride along.  Per tile and ONE barrier: every wave
// multiplies its 32 columns into dU (operands by transposing LDS reads, k = 32 rows), and the dwn products of the tile
// (2 x 16 rows, 2 x 16 slots, all 256 columns -- no partial sums to exchange) are pipelined over the eight waves: four
// take the MFMA stage of this tile, the other four apply d w / d coords to what they multiplied one tile earlier; the two
// slot halves of a row meet in LDS and are stored one tile later.  DMA completion is counted by hand (siren.hip, k_dw256).
#define NG_TR 32                               // rows per tile
#define NG_D 3
#define NG_XT (NG_TR * 512)                    // X tile
#define NG_WT (NG_TR * 64)                     // weights of the tile, row-major [row][32 slots] bf16
#define NG_SLOT (NG_XT + NG_WT)
#define NG_UP 16384                            // U' image: 32 rows x 512 B
#define NG_AT 1024                             // axes of the tile: 32 rows x 8 words
#define NG_AD 4                                // axis ring (read one tile later than the X ring)
#define NG_RB (2 * 2 * 2 * 16 * 4 * 4)         // [parity][row half][slot half][row][4] fp32
#define NG_SMEM (NG_UP + NG_D * NG_SLOT + NG_AD * NG_AT + NG_RB)
// 16-B chunk c of row r of a 512-B-row image sits at position c ^ key(r): 16 rows reading one chunk (ds_read_b128) and
// 8 rows x 2 chunks of a transposing read both cover every bank once
__device__ __forceinline__ int ng_key(int r) { return ((r & 7) << 1) | ((r >> 3) & 1); }
__device__ __forceinline__ void ng_glds16(const void *gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void ng_wait(int n) {      // s_waitcnt vmcnt(n), n wave-uniform
  switch (n) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
  }
}
// rows row0, row0 + 1 (image rows ir0, ir0 + 1) of src [rows][ld], columns c0 .. c0 + 255 -> 1 KB of a swizzled image
__device__ __forceinline__ void ng_stage2(const bf16_t *src, long ld, long row0, int ir0, int c0, int climit, lds_char *image, int lane) {
  const int r = lane >> 5, p = lane & 31;
  int col = c0 + ((p ^ ng_key(ir0 + r)) << 3);
  if (col + 8 > climit) col = c0;
  ng_glds16(src + (row0 + r) * ld + col, (unsigned)(__SIZE_TYPE__)(image + ir0 * 512));
}
// transposing read of 16 columns x 32 rows (k = row) of a swizzled 512-B-row image: an MFMA operand with k = rows
__device__ __forceinline__ bf16x8 ng_tr_frag(const lds_char *image, int colbase, int lane) {
  typedef LDS_AS s16x4 lds_s16x4;
  const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
  const int r0 = 8 * g + q, colbyte = (colbase + 4 * p) * 2;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (lds_s16x4 *)(image + r0 * 512 + ((((colbyte >> 4) ^ ng_key(r0)) << 4) | (colbyte & 15))));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (lds_s16x4 *)(image + (r0 + 4) * 512 + ((((colbyte >> 4) ^ ng_key(r0 + 4)) << 4) | (colbyte & 15))));
  s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}
// the same from the tile's weight rows [32 rows][32 slots] (64-B rows, not swizzled)
__device__ __forceinline__ bf16x8 ng_tr_wfrag(const lds_char *wt, int slotbase, int lane) {
  typedef LDS_AS s16x4 lds_s16x4;
  const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
  const int r0 = 8 * g + q;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(wt + r0 * 64 + (slotbase + 4 * p) * 2));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(wt + (r0 + 4) * 64 + (slotbase + 4 * p) * 2));
  s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}

__device__ __forceinline__ void ng_body_tensor(const NgGradArgs &a, const long s, const int t, const int cg, const int part,
                                               lds_char *smem, const int beg, const int end) {
  lds_char *UP = smem, *XR = UP + NG_UP, *AR = XR + NG_D * NG_SLOT;
  LDS_AS float *RR = (LDS_AS float *)(AR + NG_AD * NG_AT);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 15, g = lane >> 4;
  const int jm = t >> 1, half = t & 1, layer = a.layer_of[jm];
  const bf16_t *X = a.gbuf + a.slot_elems * (half ? a.gb_t[layer] : a.gb_s[layer]);
  const int c0 = 256 * cg;
  const int cols = a.HP - c0 < 256 ? a.HP - c0 : 256;
  const int nkk = cols >> 5;
  const bool kv = wave < nkk;                                // wave-uniform: this wave's 32-column block exists
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  const long b0 = beg / NG_TR, b1 = ((long)end + NG_TR - 1) / NG_TR;
  const int nt = (int)(b1 - b0);
  const int per_tile = wave == 0 ? 4 : (wave == 1 ? 3 : 2); // DMA instructions of this wave per tile
  auto stage_tile = [&](int k) {
    const long bb = b0 + k;
    lds_char *sl = XR + (k % NG_D) * NG_SLOT;
#pragma unroll
    for (int j = 0; j < 2; ++j) ng_stage2(X, a.HP, bb * NG_TR + 4 * wave + 2 * j, 4 * wave + 2 * j, c0, a.HP, sl, lane);
    if (wave == 0) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
        ng_glds16(a.wrow + (bb * NG_TR + 16 * j) * NG_SLOTS + lane * 8, (unsigned)(__SIZE_TYPE__)(sl + NG_XT + j * 1024));
    }
    if (wave == 1) ng_glds16(a.axt + bb * NG_TR * 8 + lane * 4, (unsigned)(__SIZE_TYPE__)(AR + (k % NG_AD) * NG_AT));
  };
  // U' rows of this subject and tensor: 32 rows x 256 columns, two instructions per wave
#pragma unroll
  for (int j = 0; j < 2; ++j) ng_stage2(a.Up + (long)t * a.HP, a.F, s * NG_SLOTS + 4 * wave + 2 * j, 4 * wave + 2 * j, c0, a.HP, UP, lane);
  for (int k = 0; k < NG_D - 1 && k < nt; ++k) stage_tile(k);
  // dwn roles of this wave: rows 16 h .. + 15 of a tile, slots 16 mt .. + 15, tiles of parity tp in the MFMA stage
  const int h = wave & 1, mt = (wave >> 1) & 1, tp = wave >> 2;
  int sl4[4];                                                // the lane's 4 slots 16 mt + 4 g + r -> (ix, iy, iz, is a node)
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const SlotIdx sq = slot_index(16 * mt + 4 * g + r, a.lx, a.ly, a.lz);
    sl4[r] = sq.ix | (sq.iy << 8) | (sq.iz << 16) | (sq.node ? 1 << 24 : 0);
  }
  const float dmv[3] = {0.5f * (float)(a.lz - 1), 0.5f * (float)(a.ly - 1), 0.5f * (float)(a.lx - 1)};
  f32x4 dUa[2][2] = {{zero4, zero4}, {zero4, zero4}};
  f32x4 cb = zero4;                                          // dwn of the tile this wave multiplied last
  for (int k = 0; k < nt + 2; ++k) {
    if (k < nt) {
      const int rem = nt - 1 - k < NG_D - 2 ? nt - 1 - k : NG_D - 2;    // tiles that may stay in flight behind tile k
      ng_wait(rem * per_tile);
    }
    __builtin_amdgcn_s_barrier();
    if (k + NG_D - 1 < nt) stage_tile(k + NG_D - 1);         // its X slot held tile k - 1: every wave is past it
    const lds_char *xt = XR + (k % NG_D) * NG_SLOT;
    if (k < nt && kv && !(a.dbg & 2)) {
      const long b = b0 + k;
      bf16x8 wf[2];
#pragma unroll
      for (int nt2 = 0; nt2 < 2; ++nt2) {
        wf[nt2] = ng_tr_wfrag(xt + NG_XT, 16 * nt2, lane);
        if (NG_TR * b < beg || NG_TR * b + NG_TR > end) {    // a subject's first and last tile: rows of its neighbours
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const long n = NG_TR * b + 8 * g + e;
            if (n < beg || n >= end) wf[nt2][e] = (bf16_t)0.f;
          }
        }
      }
#pragma unroll
      for (int ft = 0; ft < 2; ++ft) {
        const bf16x8 xf = ng_tr_frag(xt, 32 * wave + 16 * ft, lane);
#pragma unroll
        for (int nt2 = 0; nt2 < 2; ++nt2) dUa[ft][nt2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf, wf[nt2], dUa[ft][nt2], 0, 0, 0);
      }
    }
    if (a.dbg & 1) continue;
    if ((k & 1) == tp) {
      // finish the tile of two steps ago (its slot halves were left in LDS one step ago), then multiply tile k
      if (mt == 0 && k >= 2) {
        const long n = (b0 + k - 2) * NG_TR + 16 * h + i;
        const LDS_AS float *r0 = RR + (((k & 1) * 2 + h) * 2) * 64 + i * 4;
        if (g < 3 && n >= beg && n < end) a.dcn[((long)part * a.Npad + n) * 4 + g] = r0[g] + r0[64 + g];
      }
      if (k < nt) {
        f32x4 cb0 = zero4, cb1 = zero4;
#pragma unroll
        for (int kk = 0; kk < 8; kk += 2) {
          if (kk < nkk) {
            const int pos0 = ((4 * kk + g) ^ ng_key(i)) << 4, pos1 = ((4 * kk + 4 + g) ^ ng_key(i)) << 4;
            const bf16x8 xb0 = *(const LDS_AS bf16x8 *)(xt + (16 * h + i) * 512 + pos0);
            const bf16x8 xb1 = *(const LDS_AS bf16x8 *)(xt + (16 * h + i) * 512 + pos1);
            const bf16x8 ua0 = *(const LDS_AS bf16x8 *)(UP + (16 * mt + i) * 512 + pos0);
            const bf16x8 ua1 = *(const LDS_AS bf16x8 *)(UP + (16 * mt + i) * 512 + pos1);
            cb0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ua0, xb0, cb0, 0, 0, 0);
            cb1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ua1, xb1, cb1, 0, 0, 0);
          }
        }
        cb = cb0 + cb1;
      }
    } else if (k >= 1 && k - 1 < nt) {
      // d w / d coords on the dwn of tile k - 1 (this wave's rows and slots): axes of row 16 h + i from the axis ring
      const lds_char *at = AR + ((k - 1) % NG_AD) * NG_AT + (16 * h + i) * 32;
      const u32x4 r01 = *(const LDS_AS u32x4 *)at;
      const unsigned r2n = *(const LDS_AS unsigned *)(at + 16), r2w = *(const LDS_AS unsigned *)(at + 20);
      const unsigned nd[3] = {r01[0], r01[2], r2n};          // 0: lz axis (coordinate column 0), 1: ly, 2: lx
      const float w0v[3] = {__uint_as_float(r01[1]), __uint_as_float(r01[3]), __uint_as_float(r2w)};
      int n0[3], n1[3]; float w0[3], w1[3], dm[3];
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        n0[q] = nd[q] & 255; n1[q] = (nd[q] >> 8) & 255;
        const bool two = n1[q] != n0[q];
        w0[q] = w0v[q];
        w1[q] = two ? 1.f - w0v[q] : 0.f;
        dm[q] = (two && (nd[q] >> 16)) ? dmv[q] : 0.f;
      }
      float dc[3] = {0.f, 0.f, 0.f};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int id[3] = {(sl4[r] >> 16) & 255, (sl4[r] >> 8) & 255, sl4[r] & 255};     // iz, iy, ix
        const float v = (sl4[r] >> 24) ? cb[r] : 0.f;
        float wv[3], dv[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          const bool e0 = id[q] == n0[q], e1 = id[q] == n1[q];
          wv[q] = (e0 ? w0[q] : 0.f) + (e1 ? w1[q] : 0.f);
          dv[q] = (e1 ? dm[q] : 0.f) - (e0 ? dm[q] : 0.f);
        }
        dc[0] += v * wv[2] * wv[1] * dv[0];
        dc[1] += v * wv[2] * dv[1] * wv[0];
        dc[2] += v * dv[2] * wv[1] * wv[0];
      }
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        dc[q] += __shfl_xor(dc[q], 16, 64);
        dc[q] += __shfl_xor(dc[q], 32, 64);
      }
      if (g < 3) RR[((((k - 1) & 1) * 2 + h) * 2 + mt) * 64 + i * 4 + g] = dc[g];
    }
  }
  // dU rows of this subject: lane (i = slot, g): features c0 + 32 wave + 16 ft + 4 g .. + 3 -- to global memory (operand of
  // the weight-gradient problems) and, as a swizzled image over the U' rows, to LDS
  __syncthreads();                                           // every wave is out of the loop: U' and the rings are free
#pragma unroll
  for (int ft = 0; ft < 2; ++ft)
#pragma unroll
    for (int nt2 = 0; nt2 < 2; ++nt2) {
      const bf16x4 v = kv ? cvt4(dUa[ft][nt2]) : cvt4(zero4);
      const int row = 16 * nt2 + i, colbyte = (32 * wave + 16 * ft + 4 * g) * 2;
      if (kv) *reinterpret_cast<bf16x4 *>(a.dU + (s * NG_SLOTS + row) * a.F + (long)t * a.HP + c0 + 32 * wave + 16 * ft + 4 * g) = v;
      *(LDS_AS bf16x4 *)(UP + row * 512 + ((((colbyte >> 4) ^ ng_key(row)) << 4) | (colbyte & 15))) = v;
    }
  __syncthreads();
  // this workgroup's part of d (node rows): dT[slot][c] = sum_{f in its columns} dU[slot][f] * Wl[f][c]; C[m = c][n = slot],
  // A = 16 rows of the transposed packed copy (all k steps requested before the first product), B = the LDS image
  const bf16_t *WT = a.packed + (half ? a.pk_wltT[layer] : a.pk_wlsT[layer]) + c0;
  float *dTp = a.dT + (long)part * a.dT_stride + s * NG_SLOTS * a.LP;
  for (int ct = wave; ct < (a.LP >> 4); ct += 8) {
    bf16x8 af[8];
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) af[kk] = *reinterpret_cast<const bf16x8 *>(WT + (long)(16 * ct + i) * a.HP + 32 * (kk < nkk ? kk : 0) + 8 * g);
    f32x4 acc[2] = {zero4, zero4};
#pragma unroll
    for (int kk = 0; kk < 8; ++kk)
      if (kk < nkk) {
#pragma unroll
        for (int nt2 = 0; nt2 < 2; ++nt2) {
          const bf16x8 bfr = *(const LDS_AS bf16x8 *)(UP + (16 * nt2 + i) * 512 + (((4 * kk + g) ^ ng_key(16 * nt2 + i)) << 4));
          acc[nt2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[kk], bfr, acc[nt2], 0, 0, 0);
        }
      }
    // lane (i = slot, g): latent columns 16 ct + 4 g .. + 3
#pragma unroll
    for (int nt2 = 0; nt2 < 2; ++nt2) *reinterpret_cast<f32x4 *>(dTp + (16 * nt2 + i) * a.LP + 16 * ct + 4 * g) = acc[nt2];
  }
}

// dOut part: every wave takes every 8th block with all 16 columns (no exchange); the dUo partials meet in LDS at the end
__device__ __forceinline__ void ng_body_dout(const NgGradArgs &a, const long s, const int part, float (*P)[64][8], const int beg,
                                             const int end) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 15, g = lane >> 4;
  const bf16_t *X = a.gbuf + a.slot_elems * a.dout_slot;      // [Npad][16] bf16
  float *Uo = &P[0][0][0] + 8 * 64 * 8;                       // [32 slots][16] behind the partials
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
      *reinterpret_cast<f32x4 *>(Uo + (16 * nt + i) * 16 + 4 * g) = t4;
    }
  }
  __syncthreads();
  // the last part of d (node rows): dT[slot][c] = sum_o dUo[slot][o] * final_linear.weight[o][H + c], fp32; thread = (column,
  // half of the slots)
  float *dTp = a.dT + (long)part * a.dT_stride + s * NG_SLOTS * a.LP;
  for (int c = tid & 255; c < a.LP; c += 256) {
    float wv[16];
#pragma unroll
    for (int o = 0; o < 16; ++o) wv[o] = (o < a.out_dim && c < a.lat) ? a.wf_lat[(long)o * a.wf_ld + c] : 0.f;
    const int sl0 = (tid >> 8) * 16;
#pragma unroll 4
    for (int q = 0; q < 16; ++q) {
      float v = 0.f;
#pragma unroll
      for (int o = 0; o < 16; ++o) v += Uo[(sl0 + q) * 16 + o] * wv[o];
      dTp[(sl0 + q) * a.LP + c] = v;
    }
  }
}

__global__ __launch_bounds__(512, 4) void k_node_grad(NgGradArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  lds_char *smem = (lds_char *)smem_raw;
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
      for (int q = threadIdx.x; q < NG_SLOTS * a.LP / 4; q += 512) {
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        *reinterpret_cast<f32x4 *>(a.dT + (long)part * a.dT_stride + s * NG_SLOTS * a.LP + 4 * q) = z;
      }
      return;
    }
    ng_body_tensor(a, s, t, cg, part, smem, beg, end);
  } else {
    ng_body_dout(a, s, part, (float (*)[64][8])smem_raw, beg, end);
  }
}

// ----------------------------------------------------------------------------------------
// tail of the backward: block ranges of one launch
// ----------------------------------------------------------------------------------------
struct NgTailArgs {
  Dw256Args red; int g_red, skip;
  // linear_lats.bias gradients: column sums of the constant-1 slot rows of dU
  const bf16_t *dU; float *gp; long off_bl[CINEMA_MAX_LAYERS]; int layer_of[CINEMA_MAX_LAYERS]; int n_mod, H, HP, F, g_bias;
  // d table / d cond from the dT parts
  const float *dT; long part_stride; int nseg, LP, C, G, cond_dims; float *dlatents, *dcond; int g_tab, tab_cs;
  // embedding-table gradient
  const float *coords, *dcoords, *dcn, *embed; float *dembed; const int32_t *offs; long N, Npad, S; int tf_dim, nparts, g_tf;
  // column-sum replicas of the chain kernel (nodegrad.h: rep)
  float *rep; int rep_w, n_layers, in_dim, g_rep; long off_b[CINEMA_MAX_LAYERS], off_w0;
};

// thread = one b_l or first-layer weight gradient element: the sum of its NG_REP replicas joins the flat gradient, the
// replicas are cleared for the next step
__device__ __forceinline__ void ng_tail_rep(const NgTailArgs &a, int vb) {
  const int e = vb * 256 + threadIdx.x, nb = a.n_layers * a.H;
  if (e >= nb + a.H * a.in_dim) return;
  int src; long dst;
  if (e < nb) { const int l = e / a.H, f = e - l * a.H; src = l * a.HP + f; dst = a.off_b[l] + f; }
  else { const int q = e - nb, f = q / a.in_dim, k = q - f * a.in_dim; src = a.n_layers * a.HP + f * 4 + k; dst = a.off_w0 + q; }
  float v[NG_REP];
#pragma unroll
  for (int r = 0; r < NG_REP; ++r) v[r] = a.rep[(long)r * a.rep_w + src];
  float t = 0.f;
#pragma unroll
  for (int r = 0; r < NG_REP; ++r) { t += v[r]; a.rep[(long)r * a.rep_w + src] = 0.f; }
  a.gp[dst] += t;
}

// workgroup = 32 columns of dU x 8 groups of subjects: a thread requests its subjects' rows eight at a time (one thread
// per column walked S dependent-latency loads in a row and paced the whole launch)
__device__ __forceinline__ void ng_tail_bias(const NgTailArgs &a, int vb) {
  __shared__ float part[8][33];
  const int tid = threadIdx.x, cl = tid & 31, sg = tid >> 5;
  const int col = vb * 32 + cl;                       // column of dU: (jm, half, feature)
  float acc = 0.f;
  if (col < a.F) {
    for (long s0 = sg; s0 < a.S; s0 += 64) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const long s = s0 + 8 * u;
        v[u] = (float)a.dU[((s < a.S ? s : a.S - 1) * NG_SLOTS + NG_ONE_SLOT) * a.F + col];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += s0 + 8 * u < a.S ? v[u] : 0.f;
    }
  }
  part[sg][cl] = acc;
  __syncthreads();
  if (tid < 32 && col < a.F) {
    const int t = col / a.HP, f = col - t * a.HP;
    if (f >= a.H) return;
    float r = 0.f;
#pragma unroll
    for (int g = 0; g < 8; ++g) r += part[g][cl];
    a.gp[a.off_bl[a.layer_of[t >> 1]] + (long)(t & 1) * a.H + f] = r;
  }
}

// workgroup = (subject, slice of 64 columns of the node rows): dlatents[s][c][k] = sum over the parts of dT[s*32 + k][c]
__device__ __forceinline__ void ng_tail_table(const NgTailArgs &a, int vb) {
  __shared__ float tile[NG_SLOTS][65];
  const long s = vb / a.tab_cs;
  const int c0 = (vb - (int)s * a.tab_cs) * 64;
  const int tid = threadIdx.x;
  float acc[NG_SLOTS * 64 / 256];
#pragma unroll
  for (int u = 0; u < NG_SLOTS * 64 / 256; ++u) acc[u] = 0.f;
  const int cc = tid & 63, k0 = tid >> 6;                // rows k0 + 4 u; rows >= G of the parts are zero (never written)
  const float *src = a.dT + (s * NG_SLOTS + k0) * a.LP + (c0 + cc < a.LP ? c0 + cc : 0);
  for (int p = 0; p < a.nseg; p += 2) {                  // the loads of two parts (16) are in flight together
    float v[2][NG_SLOTS * 64 / 256];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int u = 0; u < NG_SLOTS * 64 / 256; ++u)
        v[h][u] = src[(long)(p + h < a.nseg ? p + h : p) * a.part_stride + (long)4 * u * a.LP];
#pragma unroll
    for (int u = 0; u < NG_SLOTS * 64 / 256; ++u) acc[u] += v[0][u];
    if (p + 1 < a.nseg) {
#pragma unroll
      for (int u = 0; u < NG_SLOTS * 64 / 256; ++u) acc[u] += v[1][u];
    }
  }
#pragma unroll
  for (int u = 0; u < NG_SLOTS * 64 / 256; ++u) tile[k0 + 4 * u][cc] = (k0 + 4 * u < a.G && c0 + cc < a.LP) ? acc[u] : 0.f;
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
  float ev[12];                                           // requested now, used by thread 0 at the end
#pragma unroll
  for (int k = 0; k < 12; ++k) ev[k] = (tid == 0 && k < a.tf_dim) ? a.embed[s * a.tf_dim + k] : 0.f;
  float acc[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) acc[k] = 0.f;
  for (int n0 = rng[0] + tid; n0 < rng[1]; n0 += 512) {   // two rows per thread and pass: all their loads in flight together
    float d[2][3], c[2][3];
    f32x4 q[2][8];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const long n = n0 + 256 * h < rng[1] ? n0 + 256 * h : n0;
#pragma unroll
      for (int r = 0; r < 3; ++r) { d[h][r] = a.dcoords[n * 3 + r]; c[h][r] = a.coords[n * 3 + r]; }
#pragma unroll
      for (int u = 0; u < 8; ++u) q[h][u] = *reinterpret_cast<const f32x4 *>(a.dcn + ((long)(u < a.nparts ? u : 0) * a.Npad + n) * 4);
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const long n = n0 + 256 * h;
      if (n >= rng[1]) break;
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (u < a.nparts) { d[h][0] += q[h][u][0]; d[h][1] += q[h][u][1]; d[h][2] += q[h][u][2]; }
      for (int p = 8; p < a.nparts; ++p) {                 // more than eight parts (hidden > 512 or > 3 modulated layers)
        const f32x4 x = *reinterpret_cast<const f32x4 *>(a.dcn + ((long)p * a.Npad + n) * 4);
        d[h][0] += x[0]; d[h][1] += x[1]; d[h][2] += x[2];
      }
#pragma unroll
      for (int r = 0; r < 3; ++r) {
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[r * 3 + j] += d[h][r] * c[h][j];
        acc[9 + r] += d[h][r];
      }
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
#pragma unroll
    for (int k = 0; k < 12; ++k) e[k] = ev[k];
    embed_grad(e, a.tf_dim, G, d, g);
    for (int k = 0; k < a.tf_dim; ++k) a.dembed[s * a.tf_dim + k] = g[k];
  }
}

__global__ __launch_bounds__(256) void k_ng_tail(NgTailArgs a) {
  int b = blockIdx.x;
  if (b < a.g_tf) { if (!(a.skip & 4)) ng_tail_tf(a, b); return; }      // the longest chain of dependent steps first
  b -= a.g_tf;
  if (b < a.g_red) { if (!(a.skip & 1)) dw256_reduce_body(a.red, b); return; }
  b -= a.g_red;
  if (b < a.g_tab) { if (!(a.skip & 2)) ng_tail_table(a, b); return; }
  b -= a.g_tab;
  if (b < a.g_bias) { if (!(a.skip & 8)) ng_tail_bias(a, b); return; }
  b -= a.g_bias;
  ng_tail_rep(a, b);
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
  p.T = take((size_t)p.Rp * d->LP * 2);       // T and Tc first: their places do not depend on Npad (the step head writes them)
  p.Tc = take((size_t)p.Rp * d->LP * 2);
  p.wt = take((size_t)(Npad / 16) * NG_SLOTS * 16 * 2);
  p.wrow = take((size_t)Npad * NG_SLOTS * 2 + 1024);
  p.axt = take((size_t)Npad * 8 * 4 + 1024);
  p.Up = take((size_t)p.Rp * p.F * 2);
  p.Vp = take((size_t)p.Rp * 16 * 2);
  p.dU = take((size_t)p.Rp * p.F * 2);
  p.dUo16 = take((size_t)p.Rp * 16 * 2);
  p.dUo32 = take((size_t)p.Rp * 16 * 4);
  p.dcn = take((size_t)p.nparts * Npad * 4 * 4);
  p.dt_parts = p.nparts;                     // k_node_grad leaves one part of d (node rows) per workgroup kind
  p.dT = take((size_t)p.dt_parts * p.Rp * d->LP * 4);
  p.rep_w = (d->n_layers + 4) * d->HP;
  p.rep = take((size_t)NG_REP * p.rep_w * 4);
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

// node rows T and Tc of the work buffer (cinema_step_head writes them itself when it is given the buffer)
int ng_launch_rows(const cinema_siren_desc *d, void *work, const float *latents, const float *cond_subject, int64_t S, int C,
                   int lx, int ly, int lz, int cond_dims, hipStream_t s) {
  CINEMA_REQUIRE(d->LP % 32 == 0, "node rows: padded latent width %d is not a multiple of 32", d->LP);
  const NgPlan p = ng_plan(d, S, 128);                   // the places of T and Tc do not depend on Npad
  NodeRowsCArgs r = {latents, cond_subject, (long)S, C, lx * ly * lz, cond_dims, d->LP, (bf16_t *)((char *)work + p.T),
                     (bf16_t *)((char *)work + p.Tc)};
  hipLaunchKernelGGL(k_ng_rows, dim3((unsigned)(S * (d->LP / 32))), dim3(256), 0, s, r);
  CINEMA_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int cinema_nodegrad_prepare(const cinema_siren_desc *d, const void *packed, const float *coords, const int32_t *idx,
                                       const float *embed, int32_t tf_dim, const float *latents, const float *cond_subject,
                                       int64_t N, int64_t Npad, int64_t S, int32_t C, int32_t lx, int32_t ly, int32_t lz,
                                       int32_t cond_dims, void *work, int32_t flags, void *stream) {
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
  if (!(flags & 1)) {                          // node rows not written by the step head: a launch of their own
    const int rc = ng_launch_rows(d, work, latents, cond_subject, S, C, lx, ly, lz, cond_dims, s);
    if (rc) return rc;
  }
  // flags & 2: the interpolation weights of this batch are in `work` already (cinema_step_head wrote them);
  // flags & 4: the projections are left to cinema_siren_forward_node_prep (tail workgroups of the forward launch)
  NgPrepArgs pa = {};
  pa.w = ng_weight_args(p, work, coords, idx, embed, tf_dim, (long)N, lx, ly, lz);
  pa.g_w = (flags & 2) ? 0 : ng_weight_grid(p);
  pa.p = ng_proj_args(d, p, work, packed);
  const long g_p = (flags & 4) ? 0 : ng_proj_grid(p);
  if (pa.g_w + g_p == 0) return 0;
  const int smem = std::max(ng_proj_smem(d), 16 * NG_WB * 16);
  CINEMA_REQUIRE(smem <= 160 * 1024, "cinema_nodegrad_prepare: lat_size %d too wide for the LDS staging", d->lat);
  CINEMA_REQUIRE(pa.g_w + g_p < (1L << 30), "cinema_nodegrad_prepare: too many subjects");
  CINEMA_SMEM_ATTR(k_ng_prepare, smem);
  hipLaunchKernelGGL(k_ng_prepare, dim3((unsigned)(pa.g_w + g_p)), dim3(512), smem, s, pa);
  CINEMA_CHECK_HIP(hipGetLastError());
  return 0;
}

static int g_ng_dbg = 0;
extern "C" int cinema_debug_node_grad(const cinema_siren_desc *d, void *work, const void *gbuf, const float *coords_t, int64_t N,
                                      int64_t Npad, int64_t S, int32_t lx, int32_t ly, int32_t lz, int32_t dbg,
                                      const void *packed, const float *params, void *stream) {
  g_ng_dbg = dbg;       // developer hook (tools/time_nodegrad.py): k_node_grad alone, with ablation bits
  const int rc = ng_launch_grad(d, ng_plan(d, S, Npad), work, gbuf, coords_t, nullptr, N, lx, ly, lz, packed, params, (hipStream_t)stream);
  g_ng_dbg = 0;
  return rc;
}

int ng_launch_grad(const cinema_siren_desc *d, const NgPlan &p, void *work, const void *gbuf, const float *coords_t,
                   const int32_t *idx, int64_t N, int lx, int ly, int lz, const void *packed, const float *params, hipStream_t s) {
  char *w = (char *)work;
  (void)idx;
  NgGradArgs a = {};
  a.dbg = g_ng_dbg;
  a.gbuf = (const bf16_t *)gbuf; a.slot_elems = p.Npad * d->HP;
  for (int i = 0; i < d->n_layers; ++i) { a.gb_s[i] = d->gb_s[i]; a.gb_t[i] = d->gb_t[i]; }
  for (int j = 0; j < p.n_mod; ++j) a.layer_of[j] = p.layer_of[j];
  a.n_mod = p.n_mod; a.dout_slot = d->n_gbuf - 1;
  a.wt = (const bf16_t *)(w + p.wt); a.wrow = (const bf16_t *)(w + p.wrow); a.axt = (const int32_t *)(w + p.axt); a.Up = (const bf16_t *)(w + p.Up); a.Vp = (const bf16_t *)(w + p.Vp);
  a.coords_t = coords_t; a.offs = (const int32_t *)(w + p.offs); a.ncg = p.ncg;
  a.N = (long)N; a.Npad = p.Npad; a.S = p.S; a.HP = d->HP; a.F = p.F; a.lx = lx; a.ly = ly; a.lz = lz; a.in_dim = d->in_dim;
  a.dU = (bf16_t *)(w + p.dU); a.dUo16 = (bf16_t *)(w + p.dUo16); a.dUo32 = (float *)(w + p.dUo32); a.dcn = (float *)(w + p.dcn);
  a.packed = (const bf16_t *)packed;
  for (int i = 0; i < d->n_layers; ++i) { a.pk_wlsT[i] = d->pk_wlsT[i]; a.pk_wltT[i] = d->pk_wltT[i]; }
  a.wf_lat = params + d->off_wf + d->hidden; a.wf_ld = d->hidden + d->lat; a.out_dim = d->out_dim; a.lat = d->lat; a.LP = d->LP;
  a.dT = (float *)(w + p.dT); a.dT_stride = p.Rp * (long)d->LP;
  CINEMA_REQUIRE(d->out_dim <= 16 && d->LP % 16 == 0, "node gradients: out_dim %d > 16 or padded latent width %d", d->out_dim, d->LP);
  CINEMA_REQUIRE(p.S * p.nparts < (1L << 30), "node gradients: too many subjects");
  CINEMA_SMEM_ATTR(k_node_grad, NG_SMEM);
  hipLaunchKernelGGL(k_node_grad, dim3((unsigned)(p.S * p.nparts)), dim3(512), NG_SMEM, s, a);
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
  if (t.grad_params != nullptr) {
    a.gp = t.grad_params; a.g_bias = (p.F + 31) / 32;
    a.rep = (float *)(w + p.rep); a.rep_w = p.rep_w; a.n_layers = d->n_layers; a.in_dim = d->in_dim; a.off_w0 = d->off_w[0];
    for (int i = 0; i < d->n_layers; ++i) a.off_b[i] = d->off_b[i];
    a.g_rep = (d->n_layers * d->hidden + d->hidden * d->in_dim + 255) / 256;
  }
  if (t.dlatents != nullptr) {
    a.dT = (const float *)(w + p.dT); a.part_stride = p.Rp * (long)d->LP; a.nseg = p.dt_parts; a.LP = d->LP;
    a.C = t.C; a.G = t.G; a.cond_dims = t.cond_dims; a.dlatents = t.dlatents; a.dcond = t.dcond;
    a.tab_cs = (d->LP + 63) / 64;
    a.g_tab = (int)(p.S * a.tab_cs);
  }
  if (t.dembed != nullptr) {
    a.coords = t.coords; a.dcoords = t.dcoords; a.dcn = (const float *)(w + p.dcn); a.embed = t.embed; a.dembed = t.dembed;
    a.offs = (const int32_t *)(w + p.offs); a.tf_dim = t.tf_dim; a.nparts = p.nparts; a.g_tf = (int)p.S;
  }
  static const int tail_skip = getenv("CINEMA_NG_TAIL_SKIP") ? atoi(getenv("CINEMA_NG_TAIL_SKIP")) : 0;   // developer timing only
  a.skip = tail_skip;
  const int grid = a.g_red + a.g_tab + a.g_tf + a.g_bias + a.g_rep;
  if (grid == 0) return 0;
  hipLaunchKernelGGL(k_ng_tail, dim3((unsigned)grid), dim3(256), 0, s, a);
  CINEMA_CHECK_HIP(hipGetLastError());
  return 0;
}
