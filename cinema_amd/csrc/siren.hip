// SIREN stack kernels (forward, backward, weight gradients, packing) for gfx950.
// Reference arithmetic: models/siren.py:26-36 (SineLayer.forward), :59-61 (Siren.forward).
#include "gemm_engine.h"
#include "fused.h"
#include "nodegrad.h"
#include <algorithm>
#include <cstdlib>
#include <type_traits>

namespace {

// ----------------------------------------------------------------------------------------
// small helpers
// ----------------------------------------------------------------------------------------
template <typename T> __device__ __forceinline__ void sincos_t(float a, float &s, float &c);
template <> __device__ __forceinline__ void sincos_t<float>(float a, float &s, float &c) {
  // fp32-accurate sin/cos without the library's scratch-backed Payne-Hanek path: three-term Cody-Waite
  // reduction by pi/2 (exact products for |a| < ~1e5; SIREN phases are O(100)) + minimax polynomials on
  // [-pi/4, pi/4] (Cephes sinf/cosf coefficients).  |error| <= ~2 ulp.
  const float k = rintf(a * 0.636619772367581343f);
  float r = fmaf(-k, 1.5703125f, a);
  r = fmaf(-k, 4.837512969970703125e-4f, r);
  r = fmaf(-k, 7.54978995489188216e-8f, r);
  const float r2 = r * r;
  const float ps = fmaf(r * r2, fmaf(r2, fmaf(r2, -1.9515295891e-4f, 8.3321608736e-3f), -1.6666654611e-1f), r);
  const float pc = fmaf(r2 * r2, fmaf(r2, fmaf(r2, 2.443315711809948e-5f, -1.388731625493765e-3f), 4.166664568298827e-2f),
                        fmaf(-0.5f, r2, 1.0f));
  const int q = (int)k & 3;
  const float sv = (q & 1) ? pc : ps, cv = (q & 1) ? ps : pc;
  s = (q & 2) ? -sv : sv;
  c = ((q + 1) & 2) ? -cv : cv;
}
template <> __device__ __forceinline__ void sincos_t<bf16_t>(float a, float &s, float &c) {
  s = __sinf(a);
  c = __cosf(a);
}

struct TileCtx {
  int f0, n0;          // workgroup tile origin (features, samples)
  int lane, wf, wn, g, i;
  int fb;              // first of this lane's 16 consecutive features
  int nrow[4];         // this lane's sample row for each sample tile st
};

__device__ __forceinline__ TileCtx make_ctx_id(int id, int tiles_f) {
  TileCtx c;
  c.f0 = (id % tiles_f) * 128;
  c.n0 = (id / tiles_f) * 128;
  const int tid = threadIdx.x;
  c.lane = tid & 63;
  const int wave = tid >> 6;
  c.wf = wave >> 1;
  c.wn = wave & 1;
  c.g = c.lane >> 4;
  c.i = c.lane & 15;
  c.fb = c.f0 + c.wf * 64 + 16 * c.g;
#pragma unroll
  for (int st = 0; st < 4; ++st) c.nrow[st] = c.n0 + c.wn * 64 + 16 * st + c.i;
  return c;
}
__device__ __forceinline__ TileCtx make_ctx(int tiles_f) { return make_ctx_id(xcd_remap(blockIdx.x, gridDim.x), tiles_f); }

__device__ __forceinline__ void acc_fill(f32x4 (&acc)[4][4], const float (&v)[16]) {
#pragma unroll
  for (int ft = 0; ft < 4; ++ft)
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      f32x4 x = {v[4 * ft], v[4 * ft + 1], v[4 * ft + 2], v[4 * ft + 3]};
      acc[ft][st] = x;
    }
}

// v[j] = p[fb + j] for fb + j < limit, else 0
__device__ __forceinline__ void load_feature_vec(const float *p, int fb, int limit, float (&v)[16]) {
#pragma unroll
  for (int j = 0; j < 16; ++j) v[j] = (p != nullptr && fb + j < limit) ? p[fb + j] : 0.f;
}

// acc[f][n] += sum_{k<ks} Ws[f*sf + k*sk] * Xs[n*ldx + k]   (fp32 VALU path for K <= 16)
__device__ __forceinline__ void smallk_accumulate(f32x4 (&acc)[4][4], const TileCtx &c, const float *Xs, int ldx,
                                                  int ks, const float *Ws, int sf, int sk, int flimit) {
  for (int k = 0; k < ks; ++k) {
    float xv[4];
#pragma unroll
    for (int st = 0; st < 4; ++st) xv[st] = Xs[(size_t)c.nrow[st] * ldx + k];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const float w = (c.fb + j < flimit) ? Ws[(size_t)(c.fb + j) * sf + (size_t)k * sk] : 0.f;
#pragma unroll
      for (int st = 0; st < 4; ++st) acc[j >> 2][st][j & 3] += w * xv[st];
    }
  }
}

// ----------------------------------------------------------------------------------------
// forward: one sine layer (optionally FiLM-modulated)        siren.py:26-36
// ----------------------------------------------------------------------------------------
struct FwdArgs {
  const void *X; int ldx;                                     // [Npad x K] T          (MFMA input)
  const float *Xs; int ldxs; int ks; const float *Ws; int ws_sf, ws_sk;   // small-K fp32 input
  const void *W; int K; int rowsW;                            // packed [HP x K] T
  const float *bias;                                          // [H]
  const void *LAT; int LP;                                    // [Npad x LP] T
  const void *Wls; const void *Wlt;                           // packed [HP x LP] T
  const float *bias_l;                                        // [2H]
  void *Y; void *C; void *Z; void *S; int ldo; int HP;        // outputs [Npad x HP] T
  int H; float omega; int tiles_f;
};

template <typename T, bool MOD, bool SMALLK>
__global__ __launch_bounds__(WG_THREADS) void k_layer_fwd(FwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const TileCtx c = make_ctx(a.tiles_f);
  f32x4 accZ[4][4];
  float v[16];
  load_feature_vec(a.bias, c.fb, a.H, v);
  acc_fill(accZ, v);
  if constexpr (SMALLK) {
    smallk_accumulate(accZ, c, a.Xs, a.ldxs, a.ks, a.Ws, a.ws_sf, a.ws_sk, a.H);
  } else {
    nt_mainloop<T>(accZ, (const T *)a.W + (size_t)c.f0 * a.K, a.K, a.rowsW - c.f0, (const T *)a.X + (size_t)c.n0 * a.ldx,
                   a.ldx, a.K, smem);
  }
  const bool fvalid = c.fb < a.HP;
  if constexpr (MOD) {
    f32x4 accS[4][4];
    load_feature_vec(a.bias_l, c.fb, a.H, v);
    acc_fill(accS, v);
    nt_mainloop<T>(accS, (const T *)a.Wls + (size_t)c.f0 * a.LP, a.LP, a.rowsW - c.f0,
                   (const T *)a.LAT + (size_t)c.n0 * a.LP, a.LP, a.LP, smem);
    float bt[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) bt[j] = (c.fb + j < a.H) ? a.bias_l[a.H + c.fb + j] : 0.f;
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      float zv[16], sv[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) { zv[j] = accZ[j >> 2][st][j & 3]; sv[j] = accS[j >> 2][st][j & 3]; }
      if (fvalid) {
        store16((T *)a.Z + (size_t)c.nrow[st] * a.ldo + c.fb, zv);
        store16((T *)a.S + (size_t)c.nrow[st] * a.ldo + c.fb, sv);
      }
#pragma unroll
      for (int j = 0; j < 16; ++j) accZ[j >> 2][st][j & 3] = a.omega * zv[j] * sv[j] + bt[j];
    }
    nt_mainloop<T>(accZ, (const T *)a.Wlt + (size_t)c.f0 * a.LP, a.LP, a.rowsW - c.f0,
                   (const T *)a.LAT + (size_t)c.n0 * a.LP, a.LP, a.LP, smem);
  }
#pragma unroll
  for (int st = 0; st < 4; ++st) {
    float yv[16], cv[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      float ph = accZ[j >> 2][st][j & 3];
      if constexpr (!MOD) ph *= a.omega;
      sincos_t<T>(ph, yv[j], cv[j]);
    }
    if (fvalid) {
      store16((T *)a.Y + (size_t)c.nrow[st] * a.ldo + c.fb, yv);
      store16((T *)a.C + (size_t)c.nrow[st] * a.ldo + c.fb, cv);
    }
  }
}

// ----------------------------------------------------------------------------------------
// backward chain: dy = G * W (or the small-K product with final_linear), then the elementwise
// backward of the layer whose output dy refers to; writes dz (, ds, dt) and the bias gradients.
// ----------------------------------------------------------------------------------------
struct BwdArgs {
  const void *G; int ldg;                 // [Npad x Kred] T : dz of the layer above
  const void *WT; int Kred; int rowsWT;   // packed W^T [HP x Kred] T
  const float *Xs; int ldxs; int ks; const float *Ws; int ws_sf, ws_sk;   // top of the chain: dOut, final_linear
  const void *C; const void *Z; const void *S; int ldo; int HP;
  void *DZ; void *DS; void *DT;
  float omega; int H;
  float *db; float *dbl;                  // grads of linear.bias [H], linear_lats.bias [2H] (atomics); may be NULL
  const float *coords; int in_dim; const float *W0; float *dW0; float *dcoords;   // FIRST only
  int tiles_f;
};

template <typename T, bool MOD, bool SMALLK, bool FIRST>
__global__ __launch_bounds__(WG_THREADS) void k_layer_bwd(BwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const TileCtx c = make_ctx(a.tiles_f);
  f32x4 acc[4][4];
  {
    float zero[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) zero[j] = 0.f;
    acc_fill(acc, zero);
  }
  if constexpr (SMALLK) {
    smallk_accumulate(acc, c, a.Xs, a.ldxs, a.ks, a.Ws, a.ws_sf, a.ws_sk, a.H);
  } else {
    nt_mainloop<T>(acc, (const T *)a.WT + (size_t)c.f0 * a.Kred, a.Kred, a.rowsWT - c.f0,
                   (const T *)a.G + (size_t)c.n0 * a.ldg, a.ldg, a.Kred, smem);
  }
  const bool fvalid = c.fb < a.HP;
  float sum_z[16], sum_s[16], sum_t[16], sum_w[3][16];
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    sum_z[j] = sum_s[j] = sum_t[j] = 0.f;
    sum_w[0][j] = sum_w[1][j] = sum_w[2][j] = 0.f;
  }
  float w0[3][16];
  if constexpr (FIRST) {
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
      for (int j = 0; j < 16; ++j)
        w0[k][j] = (k < a.in_dim && c.fb + j < a.H) ? a.W0[(size_t)(c.fb + j) * a.in_dim + k] : 0.f;
  }
#pragma unroll
  for (int st = 0; st < 4; ++st) {
    const size_t row = (size_t)c.nrow[st] * a.ldo + c.fb;
    float cv[16], dz[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) { cv[j] = 0.f; }
    if (fvalid) load16((const T *)a.C + row, cv);
    if constexpr (MOD) {
      float zv[16], sv[16], ds[16], dt[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) { zv[j] = 0.f; sv[j] = 0.f; }
      if (fvalid) {
        load16((const T *)a.Z + row, zv);
        load16((const T *)a.S + row, sv);
      }
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const float gq = acc[j >> 2][st][j & 3] * cv[j];
        dt[j] = gq;
        dz[j] = a.omega * gq * sv[j];
        ds[j] = a.omega * gq * zv[j];
        sum_s[j] += ds[j];
        sum_t[j] += dt[j];
      }
      if (fvalid) {
        store16((T *)a.DS + row, ds);
        store16((T *)a.DT + row, dt);
      }
    } else {
#pragma unroll
      for (int j = 0; j < 16; ++j) dz[j] = a.omega * acc[j >> 2][st][j & 3] * cv[j];
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) sum_z[j] += dz[j];
    if (fvalid) store16((T *)a.DZ + row, dz);
    if constexpr (FIRST) {
      float xk[3], dc[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        xk[k] = (k < a.in_dim) ? a.coords[(size_t)c.nrow[st] * a.in_dim + k] : 0.f;
        float p = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          sum_w[k][j] += dz[j] * xk[k];
          p += dz[j] * w0[k][j];
        }
        p += __shfl_xor(p, 16, 64);
        p += __shfl_xor(p, 32, 64);
        dc[k] = p;
      }
      if (c.g == 0) {
#pragma unroll
        for (int k = 0; k < 3; ++k)
          if (k < a.in_dim) atomicAdd(a.dcoords + (size_t)c.nrow[st] * a.in_dim + k, dc[k]);
      }
    }
  }
  // column sums over the 64 samples of this wave -> bias gradients
  const int fi = c.fb + reduce16_index(c.i);
  if (a.db != nullptr) {
    const float tz = reduce16_scatter(sum_z, c.lane);
    if (fi < a.H) atomicAdd(a.db + fi, tz);
    if constexpr (MOD) {
      const float ts = reduce16_scatter(sum_s, c.lane);
      const float tt = reduce16_scatter(sum_t, c.lane);
      if (fi < a.H) {
        atomicAdd(a.dbl + fi, ts);
        atomicAdd(a.dbl + a.H + fi, tt);
      }
    }
    if constexpr (FIRST) {
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const float tw = reduce16_scatter(sum_w[k], c.lane);
        if (k < a.in_dim && fi < a.H) atomicAdd(a.dW0 + (size_t)fi * a.in_dim + k, tw);
      }
    }
  }
}

// ----------------------------------------------------------------------------------------
// d lat = sum over modulated layers (ds * Wl_scale + dt * Wl_shift) + dOut * Wf[:, H:]
// ----------------------------------------------------------------------------------------
struct DlatArgs {
  NtSeg seg[2 * CINEMA_MAX_LAYERS];
  int nseg;
  const float *Xs; int ldxs; int ks; const float *Ws; int ws_sf, ws_sk;
  int lat; int LP; int ldg;
  float *out; int tiles_f;
};

template <typename T>
__global__ __launch_bounds__(WG_THREADS) void k_dlat(DlatArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const TileCtx c = make_ctx(a.tiles_f);
  f32x4 acc[4][4];
  {
    float zero[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) zero[j] = 0.f;
    acc_fill(acc, zero);
  }
  smallk_accumulate(acc, c, a.Xs, a.ldxs, a.ks, a.Ws, a.ws_sf, a.ws_sk, a.lat);
  nt_mainloop_segs<T>(acc, a.seg, a.nseg, c.f0, a.LP, c.n0, a.ldg, smem);
  if (c.fb < a.LP) {
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      float v[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) v[j] = acc[j >> 2][st][j & 3];
      store16(a.out + (size_t)c.nrow[st] * a.LP + c.fb, v);
    }
  }
}

// ----------------------------------------------------------------------------------------
// grouped weight-gradient GEMMs: dW[h][k] += sum_n P[n][h] * Q[n][k]
// ----------------------------------------------------------------------------------------
#include "dw_device.h"

template <typename T>
__global__ __launch_bounds__(WG_THREADS) void k_dw_grouped(DwArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using E = Engine<T>;
  // XCD-aware order: blocks that share an XCD get a contiguous range of logical ids, and logical ids walk
  // problem -> sample split -> tile, so the tiles that read the same operand slabs (same problem, same sample range)
  // run together behind one L2 instead of being dealt round-robin over the eight XCDs.
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  int pi = 0;
  while (pi + 1 < a.nprob && lid >= a.tile_start[pi + 1] * a.splitk) ++pi;
  const DwProb pr = a.p[pi];
  const int ntp = a.tile_start[pi + 1] - a.tile_start[pi];
  const int rel = lid - a.tile_start[pi] * a.splitk;
  const int split = rel / ntp, t = rel - split * ntp;
  const int h0 = (t / pr.tk) * 128, c0 = (t % pr.tk) * 128;
  const int ntile = (pr.rows ? pr.rows : a.Npad) / E::BN;
  const int nbeg = (int)((long)ntile * split / a.splitk) * E::BN;
  const int nend = (int)((long)ntile * (split + 1) / a.splitk) * E::BN;
  f32x4 acc[4][4];
#pragma unroll
  for (int x = 0; x < 4; ++x)
#pragma unroll
    for (int y = 0; y < 4; ++y) { f32x4 z = {0.f, 0.f, 0.f, 0.f}; acc[x][y] = z; }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wh = wave >> 1, wk = wave & 1;
  const int rows_left = pr.M - h0 - wh * 64;                          // valid output rows of this wave
  const int nht = __builtin_amdgcn_readfirstlane(rows_left <= 0 ? 0 : (rows_left >= 64 ? 4 : (rows_left + 15) >> 4));
  tn_mainloop<T>(acc, (const T *)pr.P, pr.ldp, h0, (const T *)pr.Q, pr.ldq, c0, nbeg, nend, nht, smem);
  const int g = lane >> 4, i = lane & 15;
#pragma unroll
  for (int ht = 0; ht < 4; ++ht)
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      const int col = c0 + wk * 64 + kt * 16 + i;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = h0 + wh * 64 + ht * 16 + 4 * g + r;
        if (row < pr.M && col < pr.K) atomicAdd(pr.out + (size_t)row * pr.ldo + col, acc[ht][kt][r]);
      }
    }
}

// ----------------------------------------------------------------------------------------
// bf16, wide stacks: the same grouped weight-gradient GEMMs on 256 x 256 output tiles.
// With 128 x 128 tiles a 512 x 512 problem re-reads both operand slabs four times through L2 -> CU (1.3 GB per step at
// cfg-D, the chip-wide L2 rate is ~17 TB/s) and every split adds its tile with float atomics (~1.3 TB/s chip-wide).
// Here: 8 waves (2 x 4), wave tile 128 x 64 (acc = 128 registers), operands [n][feature] staged global -> LDS by
// LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write pass; the XOR swizzle of the LDS image goes on
// the per-lane SOURCE address, the destination is lane-linear), two 64-sample buffers, fragments by
// ds_read_b64_tr_b16.  Splits write plain fp32 slabs; k_dw256_reduce sums them (deterministic, no atomics).
// ----------------------------------------------------------------------------------------
#define DW256_THREADS 512
#define DW256_BN 32                               // samples per staged tile
#define DW256_ROWB 512                            // LDS row: 256 bf16 columns
#define DW256_TILE (DW256_BN * DW256_ROWB)        // 16 KB per operand tile
#define DW256_NBUF 4                              // tiles k+1..k+3 are in flight while tile k is multiplied
#define DW256_SMEM (DW256_NBUF * 2 * DW256_TILE)  // 128 KB
#define DW256_MAX_SPLIT 12

typedef __attribute__((address_space(1))) const void gmem_cvoid;

__device__ __forceinline__ int dw256_key(int row) { return (row & 3) | (((row >> 3) & 1) << 2); }
__device__ __forceinline__ int dw256_off(int row, int colbyte) {
  return row * DW256_ROWB + ((((colbyte >> 4) ^ (dw256_key(row) << 1)) << 4) | (colbyte & 15));
}
// fragment of 16 columns x 32 rows (k = sample): lanes 16g..16g+15 fetch rows 8g+{0..7}
__device__ __forceinline__ bf16x8 dw256_frag(const lds_char *tile, int colbase, int lane) {
  const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
  const int r0 = 8 * g + q;
  typedef LDS_AS s16x4 lds_s16x4;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(tile + dw256_off(r0, (colbase + 4 * p) * 2)));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(tile + dw256_off(r0 + 4, (colbase + 4 * p) * 2)));
  s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}
// LDS-DMA issued from an asm statement: hipcc then knows nothing of it.  Through the builtin its waitcnt pass must
// assume that every later LDS read depends on the DMA and puts vmcnt(0) in front of the first fragment read of each
// tile, which leaves one tile in flight instead of three.  The price: completion is counted by hand (below).
__device__ __forceinline__ void glds16_asm(const void *gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
// rows n0..n0+31 of X [n][ld], columns col0..col0+255 -> LDS image [32][512 B]; one wave-instruction fills two rows
// (destination = wave-uniform base + lane * 16: the XOR swizzle of the image is applied to the per-lane SOURCE address)
__device__ __forceinline__ void dw256_stage(const bf16_t *X, int ld, int col0, long n0, lds_char *dst, int wave, int lane) {
  const int pos = lane & 31, rsub = lane >> 5;
#pragma unroll
  for (int rr = 0; rr < 2; ++rr) {
    const int row = 16 * rr + 2 * wave + rsub;
    int col = col0 + ((pos ^ (dw256_key(row) << 1)) << 3);
    if (col + 8 > ld) col = 0;
    glds16_asm(X + (n0 + row) * ld + col, (unsigned)(__SIZE_TYPE__)(dst + (16 * rr + 2 * wave) * DW256_ROWB));
  }
}

template <int NHT>
__device__ __forceinline__ void dw256_mma(f32x4 (&acc)[8][4], const lds_char *sP, const lds_char *sQ, int wh, int wk, int lane) {
  bf16x8 b[4];
#pragma unroll
  for (int kt = 0; kt < 4; ++kt) b[kt] = dw256_frag(sQ, wk * 64 + kt * 16, lane);
#pragma unroll
  for (int ht = 0; ht < NHT; ++ht) {
    const bf16x8 af = dw256_frag(sP, wh * 128 + ht * 16, lane);
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) acc[ht][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, b[kt], acc[ht][kt], 0, 0, 0);
  }
}

__global__ __launch_bounds__(DW256_THREADS) void k_dw256(Dw256Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  lds_char *base = (lds_char *)smem;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), wh = wave >> 2, wk = wave & 3;
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  int pi = 0;
  while (pi + 1 < a.g.nprob && lid >= a.g.p[pi + 1].wg0) ++pi;
  const DwProb pr = a.g.p[pi];
  const int ntp = a.g.tile_start[pi + 1] - a.g.tile_start[pi];
  const int rel = lid - pr.wg0;
  const int split = rel / ntp, t = rel - split * ntp;
  const int h0 = (t / pr.tk) * 256, c0 = (t % pr.tk) * 256;
  const int ntile = pr.rows / DW256_BN;
  const int tbeg = (int)((long)ntile * split / pr.nsplit), tend = (int)((long)ntile * (split + 1) / pr.nsplit);
  f32x4 acc[8][4];
#pragma unroll
  for (int x = 0; x < 8; ++x)
#pragma unroll
    for (int y = 0; y < 4; ++y) { f32x4 z = {0.f, 0.f, 0.f, 0.f}; acc[x][y] = z; }
  const int rows_left = pr.M - h0 - wh * 128, cols_left = pr.K - c0 - wk * 64;
  // rows beyond M multiply clamped (valid) columns and are masked at the store: only the one-subtile case of the
  // final_linear rows (out_dim <= 16) is worth a variant of its own
  const int nht = __builtin_amdgcn_readfirstlane(
      (rows_left <= 0 || cols_left <= 0) ? 0 : (rows_left >= 128 ? 8 : (rows_left + 15) >> 4));
  const bf16_t *P = (const bf16_t *)pr.P, *Q = (const bf16_t *)pr.Q;
  const long nbeg = (long)tbeg * DW256_BN;
  const int nt = tend - tbeg;
  // Tiles k+1..k+3 are in flight while tile k is multiplied.  Per tile and thread: 4 DMA instructions, retired in order.
#pragma unroll
  for (int j = 0; j < DW256_NBUF - 1; ++j)
    if (j < nt) {
      dw256_stage(P, pr.ldp, h0, nbeg + j * DW256_BN, base + j * 2 * DW256_TILE, wave, lane);
      dw256_stage(Q, pr.ldq, c0, nbeg + j * DW256_BN, base + j * 2 * DW256_TILE + DW256_TILE, wave, lane);
    }
  // slot J of a group of four works on LDS buffer J (static addresses: a run-time buffer index costs ~50 registers of
  // per-buffer offsets and spills -- and scratch reloads are VMEM loads, which would corrupt the counted waits below)
#define DW256_SLOT(J)                                                                                         \
  {                                                                                                           \
    const int k = k0 + (J);                                                                                   \
    if (k < nt) {                                                                                             \
      /* tile k has landed: this wave's DMAs by the counted wait (tiles k+1, k+2 stay in flight), everyone's by the */ \
      /* barrier, which also says every wave is done reading tile k-1 -- whose buffer the DMA of tile k+3 overwrites */ \
      if (k + 2 < nt) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                                        \
      else if (k + 1 < nt) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");                                   \
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                   \
      __builtin_amdgcn_s_barrier();                                                                           \
      if (k + 3 < nt) {                                                                                       \
        lds_char *nx = base + (((J) + 3) & 3) * 2 * DW256_TILE;                                               \
        dw256_stage(P, pr.ldp, h0, nbeg + (long)(k + 3) * DW256_BN, nx, wave, lane);                          \
        dw256_stage(Q, pr.ldq, c0, nbeg + (long)(k + 3) * DW256_BN, nx + DW256_TILE, wave, lane);             \
      }                                                                                                       \
      const lds_char *cur = base + (J) * 2 * DW256_TILE;                                                      \
      if (nht >= 2) dw256_mma<8>(acc, cur, cur + DW256_TILE, wh, wk, lane);                                   \
      else if (nht == 1) dw256_mma<1>(acc, cur, cur + DW256_TILE, wh, wk, lane);                              \
    }                                                                                                         \
  }
  for (int k0 = 0; k0 < nt; k0 += 4) { DW256_SLOT(0) DW256_SLOT(1) DW256_SLOT(2) DW256_SLOT(3) }
#undef DW256_SLOT
  float *slab = a.slab + ((size_t)pr.slab0 + (size_t)t * pr.nsplit + split) * (256 * 256);
  const int g = lane >> 4, i = lane & 15;
#pragma unroll
  for (int ht = 0; ht < 8; ++ht)
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      const int col = wk * 64 + kt * 16 + i;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = wh * 128 + ht * 16 + 4 * g + r;
        if (h0 + row < pr.M && c0 + col < pr.K) slab[row * 256 + col] = acc[ht][kt][r];
      }
    }
}

// ----------------------------------------------------------------------------------------
// bf16 d-latent GEMM on 256 (latent features) x 128 (samples) tiles.  k_dlat's 128 x 128 tiles make every workgroup
// pull 0.8 MB of ds / dt rows AND 0.8 MB of weights per 100 MFLOP, two workgroups per CU: 35 GB/s per CU, the rate one
// CU gets from L2 / Infinity Cache, for 89 us.  Here one workgroup owns ALL latent features of its 128 samples (the
// ds / dt rows are read once), operands go global -> LDS by LDS-DMA into three 48 KB buffers (two tiles in flight,
// counted vmcnt + raw barrier as in k_dw256), fragments are plain ds_read_b128 (both operands are k-contiguous).
// ----------------------------------------------------------------------------------------
#define DL256_THREADS 768                         // 8 multiply waves + 4 loader waves
#define DL256_BK 64
#define DL256_ROWB 128
#define DL256_A_TILE (256 * DL256_ROWB)
#define DL256_B_TILE (128 * DL256_ROWB)
#define DL256_STAGE (DL256_A_TILE + DL256_B_TILE)
#define DL256_SMEM (3 * DL256_STAGE)

// A: rows f0.. of the packed [rowsA x K] weights, LDS row pr holds feature row perm(pr) (see gemm_engine.h: a lane then
// owns 16 consecutive output features); B: rows n0.. of [n][ldB].  16-B chunk c of row r sits at position c ^ ((r>>1)&7).
// One tile = 48 DMA instructions of 1 KB (8 rows): 32 for the A rows, 16 for the B rows; loader wave d issues the
// instructions d, d + 4, ... (12 per tile).
struct Dl256Src { const bf16_t *A, *B; int K, f0, rowsA, ldB, k0; long n0; };

__device__ __forceinline__ void dl256_load_tile(const Dl256Src &q, lds_char *dst, int d, int lane) {
  const int rin = lane >> 3, pos = lane & 7;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int blk = d + 4 * j;                       // 8-row block 0..31 of the A tile
    const int pr = 8 * blk + rin;
    int f = q.f0 + ((pr & 128) | perm_feature_row(pr & 127));
    f = f < q.rowsA ? f : q.rowsA - 1;
    glds16_asm(q.A + (size_t)f * q.K + q.k0 + ((pos ^ ((pr >> 1) & 7)) << 3), (unsigned)(__SIZE_TYPE__)(dst + 8 * blk * DL256_ROWB));
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int blk = d + 4 * j;                       // 8-row block 0..15 of the B tile
    const int row = 8 * blk + rin;
    glds16_asm(q.B + (size_t)(q.n0 + row) * q.ldB + q.k0 + ((pos ^ ((row >> 1) & 7)) << 3),
               (unsigned)(__SIZE_TYPE__)(dst + DL256_A_TILE + 8 * blk * DL256_ROWB));
  }
}

__device__ __forceinline__ void dl256_frags(bf16x8 (&af)[4], bf16x8 (&bfr)[4], const lds_char *sA, const lds_char *sB, int kk,
                                            int wf, int wn, int lane) {
  const int i = lane & 15, ch = kk * 4 + (lane >> 4);
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int ra = wf * 64 + t * 16 + i, rb = wn * 64 + t * 16 + i;
    af[t] = *(const LDS_AS bf16x8 *)(sA + ra * DL256_ROWB + ((ch ^ ((ra >> 1) & 7)) << 4));
    bfr[t] = *(const LDS_AS bf16x8 *)(sB + rb * DL256_ROWB + ((ch ^ ((rb >> 1) & 7)) << 4));
  }
}

// Waves 0..7 multiply (4 x 2 wave tiles of 64 features x 64 samples), waves 8..11 only move data: an LDS-DMA
// instruction holds its wave for ~100 cycles (the CU's address path takes 64 B per clock), and eight waves that all
// stage and then all multiply after the same barrier add the two times up (measured: 21 us of MFMA + 19 us of DMA issue
// side by side).  With the loaders on their own waves the multiply waves never wait for an address.
__global__ __launch_bounds__(DL256_THREADS) void k_dlat256(DlatArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  lds_char *base = (lds_char *)smem;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool loader = wave >= 8;
  const int wf = (wave >> 1) & 3, wn = wave & 1;
  const int id = xcd_remap(blockIdx.x, gridDim.x);
  TileCtx c;
  c.f0 = (id % a.tiles_f) * 256;
  c.n0 = (id / a.tiles_f) * 128;
  c.lane = lane; c.wf = wf; c.wn = wn; c.g = lane >> 4; c.i = lane & 15;
  c.fb = c.f0 + wf * 64 + 16 * c.g;
#pragma unroll
  for (int st = 0; st < 4; ++st) c.nrow[st] = c.n0 + wn * 64 + 16 * st + c.i;
  const int K = a.seg[0].K, kps = K / DL256_BK;       // the host takes this path only when every segment has this K
  const int nt = a.nseg * kps;
  f32x4 acc[4][4];
#pragma unroll
  for (int x = 0; x < 4; ++x)
#pragma unroll
    for (int y = 0; y < 4; ++y) { f32x4 z = {0.f, 0.f, 0.f, 0.f}; acc[x][y] = z; }
  if (loader) {
    const int d = wave - 8;
    auto src_of = [&](int t) {
      const int sg = t / kps;
      Dl256Src q = {(const bf16_t *)a.seg[sg].A, (const bf16_t *)a.seg[sg].B, K, c.f0, a.LP, a.ldg, (t - sg * kps) * DL256_BK,
                    (long)c.n0};
      return q;
    };
    if (nt > 0) dl256_load_tile(src_of(0), base, d, lane);
    if (nt > 1) dl256_load_tile(src_of(1), base + DL256_STAGE, d, lane);
    // 12 DMA instructions per tile and loader wave, retired in order: at the wait of tile k the tiles k and k+1 are
    // outstanding.  The barrier tells the multiply waves that tile k is complete and tells this wave that they are
    // done with tile k-1, whose buffer tile k+2 overwrites.
    for (int k = 0; k < nt; ++k) {
      if (k + 1 < nt) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (k + 2 < nt) dl256_load_tile(src_of(k + 2), base + ((k + 2) % 3) * DL256_STAGE, d, lane);
    }
  } else {
    const bool wave_active = c.f0 + wf * 64 < a.LP;   // wave-uniform: feature rows of this wave exist
#define DL256_SLOT(J)                                                                                         \
  if (k0 + (J) < nt) {                                                                                        \
    __builtin_amdgcn_s_barrier();                                                                             \
    if (wave_active) {                                                                                        \
      const lds_char *sA = base + (J) * DL256_STAGE, *sB = sA + DL256_A_TILE;                                 \
      bf16x8 a0[4], b0[4], a1[4], b1[4];                                                                      \
      dl256_frags(a0, b0, sA, sB, 0, wf, wn, lane);                                                           \
      dl256_frags(a1, b1, sA, sB, 1, wf, wn, lane);                                                           \
      _Pragma("unroll") for (int ft = 0; ft < 4; ++ft)                                                        \
        _Pragma("unroll") for (int st = 0; st < 4; ++st)                                                      \
          acc[ft][st] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0[ft], b0[st], acc[ft][st], 0, 0, 0);        \
      _Pragma("unroll") for (int ft = 0; ft < 4; ++ft)                                                        \
        _Pragma("unroll") for (int st = 0; st < 4; ++st)                                                      \
          acc[ft][st] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1[ft], b1[st], acc[ft][st], 0, 0, 0);        \
    }                                                                                                         \
  }
    for (int k0 = 0; k0 < nt; k0 += 3) { DL256_SLOT(0) DL256_SLOT(1) DL256_SLOT(2) }
#undef DL256_SLOT
  }
  // The fp32 small-K term (dOut x final_linear.weight[:, H:], K = out_dim <= 16) goes last -- its ordinary loads must
  // not sit between the hand-counted DMAs above -- and through LDS: read per lane straight from global memory it is a
  // chain of dependent, uncoalesced round trips.
  __syncthreads();
  {
    LDS_AS float *wl = (LDS_AS float *)base;            // [ks][256]: weight of latent feature f0 + l against output k
    LDS_AS float *xl = wl + 16 * 256;                   // [128][16]: dOut rows of the tile
    for (int q = tid; q < a.ks * 256; q += DL256_THREADS) {
      const int k = q >> 8, f = c.f0 + (q & 255);
      wl[q] = f < a.lat ? a.Ws[(size_t)f * a.ws_sf + (size_t)k * a.ws_sk] : 0.f;
    }
    for (int q = tid; q < 128 * a.ks; q += DL256_THREADS) {
      const int n = q / a.ks, k = q - n * a.ks;
      xl[n * 16 + k] = a.Xs[(size_t)(c.n0 + n) * a.ldxs + k];
    }
    __syncthreads();
    if (loader) return;
    if (c.fb < a.LP)
      for (int k = 0; k < a.ks; ++k) {
        float xv[4];
#pragma unroll
        for (int st = 0; st < 4; ++st) xv[st] = xl[(wn * 64 + 16 * st + c.i) * 16 + k];
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
          const f32x4 w = *(const LDS_AS f32x4 *)(wl + k * 256 + wf * 64 + 16 * c.g + 4 * q4);
#pragma unroll
          for (int st = 0; st < 4; ++st)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[q4][st][r] += w[r] * xv[st];
        }
      }
  }
  if (c.fb < a.LP) {
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      float v[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) v[j] = acc[j >> 2][st][j & 3];
      store16(a.out + (size_t)c.nrow[st] * a.LP + c.fb, v);
    }
  }
}

__global__ __launch_bounds__(256) void k_dw256_reduce(Dw256Args a) { dw256_reduce_body(a, (int)blockIdx.x); }

// ----------------------------------------------------------------------------------------
// final linear (siren.py:61):  out[n][o] = bf[o] + sum_k [x_L | lat][n][k] * Wf[o][k]
// one wave per 16 samples, weights (padded to 16 rows) staged once per workgroup in LDS
// ----------------------------------------------------------------------------------------
struct FinalFwdArgs {
  const void *X; int ldx; int HP; const void *LAT; int LP; const void *Wf;  // Wf packed [16 x (HP+LP)] T
  const float *bias; int out_dim; float *out; int Npad;
};

template <typename T> struct FinalK;
template <> struct FinalK<bf16_t> { static constexpr int STEP = 32; static constexpr int PADB = 16; };
template <> struct FinalK<float> { static constexpr int STEP = 16; static constexpr int PADB = 16; };

template <typename T>
__global__ __launch_bounds__(WG_THREADS) void k_final_fwd(FinalFwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int KT = a.HP + a.LP;
  const int rowbytes = KT * (int)sizeof(T) + FinalK<T>::PADB;
  // stage weights: 16 rows x KT elements, 16-B chunks
  const int chunks_per_row = KT * (int)sizeof(T) / 16;
  for (int q = threadIdx.x; q < 16 * chunks_per_row; q += WG_THREADS) {
    const int r = q / chunks_per_row, ch = q - r * chunks_per_row;
    *reinterpret_cast<u32x4 *>(smem + r * rowbytes + ch * 16) =
        *reinterpret_cast<const u32x4 *>((const char *)a.Wf + ((size_t)r * KT * sizeof(T)) + ch * 16);
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 15, g = lane >> 4;
  const int ntiles = a.Npad / 16;
  for (int tile = blockIdx.x * 4 + wave; tile < ntiles; tile += gridDim.x * 4) {
    const int n = tile * 16 + i;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int part = 0; part < 2; ++part) {
      const T *src = part == 0 ? (const T *)a.X + (size_t)n * a.ldx : (const T *)a.LAT + (size_t)n * a.LP;
      const int klen = part == 0 ? a.HP : a.LP;
      const int kofs = part == 0 ? 0 : a.HP;
      if constexpr (sizeof(T) == 2) {
#pragma unroll 4
        for (int k = 0; k < klen; k += 32) {
          const bf16x8 b = *reinterpret_cast<const bf16x8 *>(src + k + 8 * g);
          const bf16x8 w = *reinterpret_cast<const bf16x8 *>(smem + i * rowbytes + (kofs + k + 8 * g) * 2);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, b, acc, 0, 0, 0);
        }
      } else {
#pragma unroll 2
        for (int k = 0; k < klen; k += 16) {
          const f32x4 b = *reinterpret_cast<const f32x4 *>(src + k + 4 * g);
          const f32x4 w = *reinterpret_cast<const f32x4 *>(smem + i * rowbytes + (kofs + k + 4 * g) * 4);
#pragma unroll
          for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w[j], b[j], acc, 0, 0, 0);
        }
      }
    }
    // acc[r] <-> output o = 4g + r of sample n
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int o = 4 * g + r;
      if (o < a.out_dim) a.out[(size_t)n * a.out_dim + o] = acc[r] + a.bias[o];
    }
  }
}

template <typename T>
__global__ __launch_bounds__(WG_THREADS) void k_dout_prep(FinalBwdArgs a, T *dout16) {
  dout_prep_body<T>(a, dout16, blockIdx.x, gridDim.x);
}

// ----------------------------------------------------------------------------------------
// packing: fp32 master weights -> zero-padded T copies (optionally transposed)
// ----------------------------------------------------------------------------------------
#include "step_device.h"
#define PACK_MAX_JOBS 64
struct PackArgs { PackJob job[PACK_MAX_JOBS]; int njobs; void *packed; };

template <typename T>
__global__ __launch_bounds__(256) void k_pack(PackArgs a) {
  pack_job_body<T>(a.job[blockIdx.y], a.packed, blockIdx.x, gridDim.x);
}

// stand-alone sine layer helpers: identity operand, padded T / fp32 rows -> dense fp32 rows
template <typename T>
__global__ __launch_bounds__(256) void k_eye(T *dst, int n) {
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < (long)n * n; e += (long)gridDim.x * 256)
    dst[e] = (T)((e / n == e % n) ? 1.f : 0.f);
}

template <typename T>
__global__ __launch_bounds__(256) void k_unpad(const T *src, int ld, float *dst, long rows, int cols) {
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < rows * cols; e += (long)gridDim.x * 256) {
    const long r = e / cols;
    dst[e] = (float)src[r * ld + (e - r * cols)];
  }
}

inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

}  // namespace

// ========================================================================================
// host side
// ========================================================================================
extern "C" int cinema_siren_plan(cinema_siren_desc *d) {
  CINEMA_REQUIRE(d != nullptr, "cinema_siren_plan: null descriptor");
  CINEMA_REQUIRE(d->n_layers >= 1 && d->n_layers <= CINEMA_MAX_LAYERS, "n_layers %d out of range [1,%d]", d->n_layers,
                 CINEMA_MAX_LAYERS);
  CINEMA_REQUIRE(d->in_dim >= 1 && d->in_dim <= 3, "in_dim %d unsupported (1..3)", d->in_dim);
  CINEMA_REQUIRE(d->hidden >= 16 && d->hidden % 16 == 0, "hidden_size %d must be a positive multiple of 16", d->hidden);
  CINEMA_REQUIRE(d->lat >= 1, "lat_size %d must be >= 1", d->lat);
  CINEMA_REQUIRE(d->out_dim >= 1 && d->out_dim <= 16, "sum(out_dim) = %d unsupported (1..16)", d->out_dim);
  CINEMA_REQUIRE(d->precision == CINEMA_PREC_F32 || d->precision == CINEMA_PREC_BF16, "bad precision %d", d->precision);
  const int H = d->hidden, L = d->lat;
  d->HP = round_up(H, 64);
  d->LP = round_up(L, 64);
  const long HP = d->HP, LP = d->LP;
  // flat fp32 parameter layout = state_dict order of the reference Siren
  long off = 0;
  d->off_wf = off; off += (long)d->out_dim * (H + L);
  d->off_bf = off; off += d->out_dim;
  long pk = 0;
  int sv = 0, gb = 0;
  for (int i = 0; i < CINEMA_MAX_LAYERS; ++i) {
    d->off_w[i] = d->off_b[i] = d->off_wl[i] = d->off_bl[i] = -1;
    d->pk_w[i] = d->pk_wt[i] = d->pk_wls[i] = d->pk_wlt[i] = d->pk_wlsT[i] = d->pk_wltT[i] = -1;
    d->sv_y[i] = d->sv_c[i] = d->sv_z[i] = d->sv_s[i] = -1;
    d->gb_z[i] = d->gb_s[i] = d->gb_t[i] = -1;
  }
  for (int i = 0; i < d->n_layers; ++i) {
    const int fan_in = i == 0 ? d->in_dim : H;
    const bool mod = (d->modulated_mask >> i) & 1u;
    d->off_w[i] = off; off += (long)H * fan_in;
    d->off_b[i] = off; off += H;
    if (mod) {
      d->off_wl[i] = off; off += 2L * H * L;
      d->off_bl[i] = off; off += 2L * H;
    }
    if (i > 0) {
      d->pk_w[i] = pk; pk += HP * HP;
      d->pk_wt[i] = pk; pk += HP * HP;
    }
    if (mod) {
      d->pk_wls[i] = pk; pk += HP * LP;
      d->pk_wlt[i] = pk; pk += HP * LP;
      d->pk_wlsT[i] = pk; pk += LP * HP;
      d->pk_wltT[i] = pk; pk += LP * HP;
    }
    d->sv_y[i] = sv++;
    d->sv_c[i] = sv++;
    d->gb_z[i] = gb++;
    if (mod) {
      d->sv_z[i] = sv++;
      d->sv_s[i] = sv++;
      d->gb_s[i] = gb++;
      d->gb_t[i] = gb++;
    }
  }
  d->pk_wf = pk; pk += 16L * (HP + LP);
  fused_plan(d, &pk);
  d->n_params = off;
  d->pk_total = pk;
  d->n_saved = sv;
  d->n_gbuf = gb + 1;   // last slot: dOut as [Npad x 16] T for the final_linear weight gradient
  // bf16 stacks with hidden >= 256: the weight-gradient GEMMs run on 256 x 256 tiles whose sample splits leave fp32
  // partial tiles in a slab region the caller appends to gbuf (k_dw256)
  d->gb_extra_bytes = 0;
  if (d->precision == CINEMA_PREC_BF16 && H >= 256) {
    auto t256 = [](long M, long K) { return ((M + 255) / 256) * ((K + 255) / 256); };
    long tiles = t256(d->out_dim, H) + t256(d->out_dim, L);
    for (int i = 1; i < d->n_layers; ++i) tiles += t256(H, H);
    for (int i = 0; i < d->n_layers; ++i)
      if ((d->modulated_mask >> i) & 1u) tiles += 2 * t256(H, L);
    d->gb_extra_bytes = tiles * 12 /* DW256_MAX_SPLIT */ * 256 * 256 * 4 + 256;
  }
  return 0;
}

// The activation-stationary kernels serve bf16 stacks up to hidden 512; CINEMA_SIREN_LAYERED=1 forces the
// layer-by-layer kernels (the only path for fp32 and wider stacks) so tests can cover both.
static int fused_mode(const cinema_siren_desc *d) {   // 0 = layered kernels, 64 = samples per workgroup of fused.hip
  static const bool layered = [] { const char *e = getenv("CINEMA_SIREN_LAYERED"); return e && e[0] == '1'; }();
  return (!layered && d->pk_fs >= 0 && fused_eligible(d)) ? 64 : 0;
}

template <typename T>
static int pack_impl(const cinema_siren_desc *d, const float *params, void *packed, int part, hipStream_t s) {
  // part bit 0: what the forward reads; bit 1: what only the backward reads (fused mode: the transposed linear_lats
  // copies of the d-latent GEMM; the layer-by-layer mode packs everything under bit 0)
  PackArgs a;
  int nj = 0;
  const int H = d->hidden, L = d->lat, HP = d->HP, LP = d->LP;
  a.packed = packed;
  auto flush = [&]() -> int {
    if (nj == 0) return 0;
    a.njobs = nj;
    hipLaunchKernelGGL((k_pack<T>), dim3(64, nj), dim3(256), 0, s, a);
    CINEMA_CHECK_HIP(hipGetLastError());
    nj = 0;
    return 0;
  };
  auto add = [&](const float *src, long dst, int rows, int cols, int ld, int rowsP, int colsP, int tr) -> int {
    PackJob j = {src, dst, rows, cols, ld, rowsP, colsP, tr};
    a.job[nj++] = j;
    return nj == PACK_MAX_JOBS ? flush() : 0;
  };
  // W, W^T and the row-major linear_lats copies are only read by the layer-by-layer kernels
  const bool fwd_layered = fused_mode(d) == 0;
  const bool nodes = (part & 4) != 0;       // nodes path: forward stream without the linear_lats segments + row-major linear_lats
  const bool rowmajor = (part & 8) != 0;    // node gradients: row-major linear_lats only (cinema_nodegrad_prepare)
  if (nodes && fwd_layered) { cinema_set_error("cinema_siren_pack_part: part 4 needs the activation-stationary bf16 path"); return -1; }
  if (!fwd_layered && (part & 1))
    if (int rc = fused_pack(d, params, packed, s)) return rc;     // weight streams, side block, padded final_linear
  if (nodes)
    if (int rc = fused_pack_nodes(d, params, packed, s)) return rc;
  if (!fwd_layered && !(part & 2) && !nodes && !rowmajor) return 0;
  if (fwd_layered && !(part & 1)) return 0;
  for (int i = 0; i < d->n_layers; ++i) {
    if (i > 0) {
      if (fwd_layered) {   // the fused chain kernels read the weight streams instead
        if (int rc = add(params + d->off_w[i], d->pk_w[i], H, H, H, HP, HP, 0)) return rc;
        if (int rc = add(params + d->off_w[i], d->pk_wt[i], H, H, H, HP, HP, 1)) return rc;
      }
    }
    if ((d->modulated_mask >> i) & 1u) {
      const float *wl = params + d->off_wl[i];
      if (fwd_layered || nodes || rowmajor) {
        if (int rc = add(wl, d->pk_wls[i], H, L, L, HP, LP, 0)) return rc;
        if (int rc = add(wl + (size_t)H * L, d->pk_wlt[i], H, L, L, HP, LP, 0)) return rc;
      }
      if (fwd_layered || (part & 2)) {
        if (int rc = add(wl, d->pk_wlsT[i], H, L, L, LP, HP, 1)) return rc;
        if (int rc = add(wl + (size_t)H * L, d->pk_wltT[i], H, L, L, LP, HP, 1)) return rc;
      }
    }
  }
  // final linear: one job per (output row, x | lat part), padded to 16 rows (fused mode: k_pack_side writes it)
  for (int o = 0; o < 16 && fwd_layered; ++o) {
    const bool live = o < d->out_dim;
    const float *row = params + d->off_wf + (size_t)(live ? o : 0) * (H + L);
    if (int rc = add(row, d->pk_wf + (long)o * (HP + LP), live ? 1 : 0, H, H + L, 1, HP, 0)) return rc;
    if (int rc = add(row + H, d->pk_wf + (long)o * (HP + LP) + HP, live ? 1 : 0, L, H + L, 1, LP, 0)) return rc;
  }
  return flush();
}

extern "C" int cinema_siren_pack_part(const cinema_siren_desc *d, const float *params, void *packed, int32_t part,
                                      void *stream) {
  CINEMA_REQUIRE(d && params && packed, "cinema_siren_pack: null argument");
  CINEMA_REQUIRE(part >= 1 && part <= 15, "cinema_siren_pack_part: part is a mask of 1 (forward), 2 (backward-only), 4 (nodes path), "
                 "8 (row-major linear_lats for cinema_nodegrad_prepare)");
  hipStream_t s = (hipStream_t)stream;
  return d->precision == CINEMA_PREC_BF16 ? pack_impl<bf16_t>(d, params, packed, part, s)
                                          : pack_impl<float>(d, params, packed, part, s);
}

extern "C" int cinema_siren_pack(const cinema_siren_desc *d, const float *params, void *packed, void *stream) {
  return cinema_siren_pack_part(d, params, packed, 3, stream);
}

template <typename T, bool MOD, bool SMALLK>
static int launch_fwd(const FwdArgs &a, int grid, hipStream_t s) {
  constexpr int smem = Engine<T>::NT_SMEM;
  CINEMA_SMEM_ATTR((k_layer_fwd<T, MOD, SMALLK>), smem);
  hipLaunchKernelGGL((k_layer_fwd<T, MOD, SMALLK>), dim3(grid), dim3(WG_THREADS), smem, s, a);
  CINEMA_CHECK_HIP(hipGetLastError());
  return 0;
}

template <typename T>
static int forward_impl(const cinema_siren_desc *d, const float *params, const void *packed, const float *coords_t,
                        const void *lat, int64_t N, int64_t Npad, void *saved, float *out, hipStream_t s) {
  const int H = d->hidden, HP = d->HP, LP = d->LP;
  const T *pk = (const T *)packed;
  T *sv = (T *)saved;
  const size_t slot = (size_t)Npad * HP;
  const int tiles_f = (HP + 127) / 128;
  const int grid = tiles_f * (int)(Npad / 128);
  for (int i = 0; i < d->n_layers; ++i) {
    const bool mod = (d->modulated_mask >> i) & 1u;
    FwdArgs a = {};
    a.bias = params + d->off_b[i];
    a.H = H; a.HP = HP; a.ldo = HP; a.tiles_f = tiles_f;
    a.omega = i == 0 ? d->omega_first : d->omega_hidden;
    a.Y = sv + slot * d->sv_y[i];
    a.C = sv + slot * d->sv_c[i];
    a.rowsW = HP;
    if (mod) {
      a.LAT = lat; a.LP = LP;
      a.Wls = pk + d->pk_wls[i];
      a.Wlt = pk + d->pk_wlt[i];
      a.bias_l = params + d->off_bl[i];
      a.Z = sv + slot * d->sv_z[i];
      a.S = sv + slot * d->sv_s[i];
    }
    int rc;
    if (i == 0) {
      a.Xs = coords_t; a.ldxs = d->in_dim; a.ks = d->in_dim;
      a.Ws = params + d->off_w[0]; a.ws_sf = d->in_dim; a.ws_sk = 1;
      rc = mod ? launch_fwd<T, true, true>(a, grid, s) : launch_fwd<T, false, true>(a, grid, s);
    } else {
      a.X = sv + slot * d->sv_y[i - 1]; a.ldx = HP;
      a.W = pk + d->pk_w[i]; a.K = HP;
      rc = mod ? launch_fwd<T, true, false>(a, grid, s) : launch_fwd<T, false, false>(a, grid, s);
    }
    if (rc) return rc;
  }
  // final linear
  FinalFwdArgs f = {};
  f.X = sv + slot * d->sv_y[d->n_layers - 1]; f.ldx = HP; f.HP = HP;
  f.LAT = lat; f.LP = LP;
  f.Wf = pk + d->pk_wf;
  f.bias = params + d->off_bf;
  f.out_dim = d->out_dim; f.out = out; f.Npad = (int)Npad;
  const int smem = 16 * ((HP + LP) * (int)sizeof(T) + FinalK<T>::PADB);
  CINEMA_SMEM_ATTR(k_final_fwd<T>, 160 * 1024 - 1024);
  CINEMA_REQUIRE(smem <= 159 * 1024, "final_linear too wide for LDS staging: hidden %d + lat %d", H, d->lat);
  const int fgrid = (int)std::min<int64_t>(Npad / 64, 1024);
  hipLaunchKernelGGL((k_final_fwd<T>), dim3(fgrid), dim3(WG_THREADS), smem, s, f);
  CINEMA_CHECK_HIP(hipGetLastError());
  (void)N;
  return 0;
}

extern "C" int cinema_siren_forward(const cinema_siren_desc *d, const float *params, const void *packed,
                                    const float *coords_t, const void *lat, int64_t N, int64_t Npad, void *saved,
                                    float *out, void *stream) {
  CINEMA_REQUIRE(d && params && packed && coords_t && lat && out, "cinema_siren_forward: null argument");
  CINEMA_REQUIRE(saved != nullptr || fused_mode(d) == 64,
                 "cinema_siren_forward: saved == NULL (inference, nothing kept) needs the activation-stationary path "
                 "(bf16, hidden <= 512); the layer-by-layer kernels pass activations through `saved`");
  CINEMA_REQUIRE(N >= 1 && Npad >= N && Npad % 128 == 0, "cinema_siren_forward: need 1 <= N <= Npad, Npad %% 128 == 0 (N=%ld Npad=%ld)",
                 (long)N, (long)Npad);
  CINEMA_REQUIRE(Npad / 128 * ((d->HP + 127) / 128) < (1 << 30), "grid too large");
  hipStream_t s = (hipStream_t)stream;
  if (fused_mode(d) == 64) return fused_forward(d, params, packed, coords_t, lat, N, Npad, saved, out, s);
  return d->precision == CINEMA_PREC_BF16
             ? forward_impl<bf16_t>(d, params, packed, coords_t, lat, N, Npad, saved, out, s)
             : forward_impl<float>(d, params, packed, coords_t, lat, N, Npad, saved, out, s);
}

extern "C" int cinema_siren_forward_node_prep(const cinema_siren_desc *d, const float *params, const void *packed,
                                              const float *coords_t, const void *lat, int64_t N, int64_t Npad, void *saved,
                                              float *out, void *node_work, int64_t S, int32_t lx, int32_t ly, int32_t lz,
                                              void *stream) {
  CINEMA_REQUIRE(d && params && packed && coords_t && lat && out && node_work, "cinema_siren_forward_node_prep: null argument");
  CINEMA_REQUIRE(fused_mode(d) == 64 && ng_eligible(d, lx, ly, lz),
                 "cinema_siren_forward_node_prep: needs a stack cinema_nodegrad_eligible accepts (bf16 chain kernels)");
  CINEMA_REQUIRE(N >= 1 && Npad >= N && Npad % 128 == 0 && S >= 1,
                 "cinema_siren_forward_node_prep: need 1 <= N <= Npad, Npad %% 128 == 0, S >= 1 (N=%ld Npad=%ld S=%ld)", (long)N,
                 (long)Npad, (long)S);
  return fused_forward(d, params, packed, coords_t, lat, N, Npad, saved, out, (hipStream_t)stream, node_work, S);
}

template <typename T, bool MOD, bool SMALLK, bool FIRST>
static int launch_bwd(const BwdArgs &a, int grid, hipStream_t s) {
  constexpr int smem = Engine<T>::NT_SMEM;
  CINEMA_SMEM_ATTR((k_layer_bwd<T, MOD, SMALLK, FIRST>), smem);
  hipLaunchKernelGGL((k_layer_bwd<T, MOD, SMALLK, FIRST>), dim3(grid), dim3(WG_THREADS), smem, s, a);
  CINEMA_CHECK_HIP(hipGetLastError());
  return 0;
}

struct NodesCtx { const void *wrows; const void *tiles; const void *U; int64_t S; int round_s; };
// node-space backward of the latent side (nodegrad.h): what the K1 backward needs beside the SIREN's own arguments
struct NgCtx {
  void *work; const float *coords; const int32_t *idx; const float *embed; int tf_dim; int64_t S; int C, lx, ly, lz, cond_dims;
  float *dlatents, *dembed, *dcond;
};

template <typename T>
static int backward_impl(const cinema_siren_desc *d, const float *params, const void *packed, const float *coords_t,
                         const void *lat, const void *saved, const float *dout, int64_t N, int64_t Npad, void *gbuf,
                         float *gp, float *dlat, float *dcoords, int32_t flags, hipStream_t s, Dw256Args *pending = nullptr,
                         const NodesCtx *nodes = nullptr, const NgCtx *ng = nullptr) {
  // pending != NULL: the split reduction of the 256 x 256 weight-gradient tiles is NOT launched; its arguments go to
  // *pending for the caller to run (cinema_affine_trilerp_bwd_sorted_pending: beside the K1 coordinate gradient)
  if (pending) { pending->g.nprob = 0; pending->slab = nullptr; }
  const int H = d->hidden, L = d->lat, HP = d->HP, LP = d->LP, nl = d->n_layers;
  const bool wgrads = !(flags & CINEMA_BWD_SKIP_WEIGHT_GRADS);
  const bool weights_first = flags & CINEMA_BWD_WEIGHTS_FIRST, dlat_only = flags & CINEMA_BWD_DLAT_ONLY;
  CINEMA_REQUIRE(!(weights_first && dlat_only) && !((weights_first || dlat_only) && (flags & CINEMA_BWD_PHASE_MASK)),
                 "cinema_siren_backward: WEIGHTS_FIRST, DLAT_ONLY and the phase bits exclude each other");
  const int phases = weights_first ? (CINEMA_BWD_PHASE_MASK & ~CINEMA_BWD_PHASE_DLAT) : dlat_only ? CINEMA_BWD_PHASE_DLAT
                     : (flags & CINEMA_BWD_PHASE_MASK) ? (flags & CINEMA_BWD_PHASE_MASK) : CINEMA_BWD_PHASE_MASK;
  const bool all_phases = phases == CINEMA_BWD_PHASE_MASK || weights_first;   // a two-call backward keeps full semantics
  const T *pk = (const T *)packed;
  const T *sv = (const T *)saved;
  T *gb = (T *)gbuf;
  const size_t slot = (size_t)Npad * HP;
  const int tiles_f = (HP + 127) / 128;
  const int grid = tiles_f * (int)(Npad / 128);
  if (wgrads && all_phases && !(flags & CINEMA_BWD_GRADS_CLEARED))
    CINEMA_CHECK_HIP(hipMemsetAsync(gp, 0, sizeof(float) * (size_t)d->n_params, s));
  if ((phases & CINEMA_BWD_PHASE_CHAIN) && !(flags & CINEMA_BWD_DCOORDS_CLEARED))
    CINEMA_CHECK_HIP(hipMemsetAsync(dcoords, 0, sizeof(float) * (size_t)Npad * d->in_dim, s));

  // ---- chain ----
  const bool fused_chain = std::is_same<T, bf16_t>::value && fused_mode(d) == 64;
  bool dout_prep_done = false;
  if (fused_chain && (phases & CINEMA_BWD_PHASE_CHAIN))
  {
    // with weight gradients the dOut preparation rides as tail workgroups of the chain launch (it only reads dOut)
    const bool prep_rides = wgrads && all_phases;
    void *d16 = prep_rides ? (void *)(gb + slot * (d->n_gbuf - 1)) : nullptr;
    float *dbf = prep_rides ? gp + d->off_bf : nullptr;
    if (nodes != nullptr) {
      if (int rc = nodes_backward_chain(d, packed, coords_t, nodes->wrows, nodes->tiles, nodes->U, nodes->S, saved, dout, N, Npad,
                                        gbuf, gp, dcoords, wgrads, nodes->round_s, s, d16, dbf))
        return rc;
    } else {
      // node gradients: the chain kernel's column sums go through the replicas of the work buffer (k_ng_tail adds them up)
      NgPlan rp = {};
      if (ng != nullptr) rp = ng_plan(d, ng->S, Npad);
      float *rep = (ng != nullptr && wgrads) ? (float *)((char *)ng->work + rp.rep) : nullptr;
      if (int rc = fused_backward_chain(d, packed, coords_t, saved, dout, Npad, gbuf, gp, dcoords, wgrads, s, d16, dbf,
                                        /*lat_bias=*/ng == nullptr, rep, rp.rep_w))
        return rc;
    }
    dout_prep_done = prep_rides;
  }
  // ---- node-space reductions of ds / dt / dOut (replace the d-latent GEMM, the linear_lats weight-gradient problems on
  //      sample rows and the K1 backward's per-sample passes) ----
  NgPlan ngp = {};
  if (ng != nullptr) {
    CINEMA_REQUIRE(fused_chain, "node gradients need the activation-stationary bf16 chain kernels");
    ngp = ng_plan(d, ng->S, Npad);
    if (phases & CINEMA_BWD_PHASE_CHAIN) {
      if (!dout_prep_done) {     // frozen weights: the dOut16 rows are still an operand of the node reduction
        FinalBwdArgs a = {};
        a.dout = dout; a.out_dim = d->out_dim; a.dbf = nullptr; a.Npad = (int)Npad;
        const int grid_f = (int)std::min<int64_t>((Npad + WG_THREADS - 1) / WG_THREADS, 256);
        hipLaunchKernelGGL((k_dout_prep<T>), dim3(grid_f), dim3(WG_THREADS), 0, s, a, gb + slot * (d->n_gbuf - 1));
        CINEMA_CHECK_HIP(hipGetLastError());
        dout_prep_done = wgrads;   // (its bias gradient was not taken: a weight-gradient pass would have to redo it)
      }
      if (int rc = ng_launch_grad(d, ngp, ng->work, gbuf, coords_t, ng->idx, N, ng->lx, ng->ly, ng->lz, packed, params, s)) return rc;
    }
  }
  auto ng_tail = [&](const Dw256Args *red, bool weights_side, bool table_side) -> int {
    NgTailArgsHost t = {};
    t.reduce = red;
    if (weights_side && wgrads) t.grad_params = gp;
    if (table_side) {
      t.dlatents = ng->dlatents; t.dcond = ng->dcond; t.dembed = ng->dembed; t.embed = ng->embed; t.tf_dim = ng->tf_dim;
      t.coords = ng->coords; t.dcoords = dcoords; t.idx = ng->idx;
    }
    t.N = N; t.C = ng->C; t.G = ng->lx * ng->ly * ng->lz; t.cond_dims = ng->cond_dims;
    return ng_launch_tail(d, ngp, ng->work, t, s);
  };
  for (int i = nl - 1; i >= 0 && (phases & CINEMA_BWD_PHASE_CHAIN) && !fused_chain; --i) {
    const bool mod = (d->modulated_mask >> i) & 1u;
    const bool top = i == nl - 1, first = i == 0;
    BwdArgs a = {};
    a.C = sv + slot * d->sv_c[i];
    a.ldo = HP; a.HP = HP; a.H = H; a.tiles_f = tiles_f;
    a.omega = first ? d->omega_first : d->omega_hidden;
    a.DZ = gb + slot * d->gb_z[i];
    if (mod) {
      a.Z = sv + slot * d->sv_z[i];
      a.S = sv + slot * d->sv_s[i];
      a.DS = gb + slot * d->gb_s[i];
      a.DT = gb + slot * d->gb_t[i];
    }
    if (wgrads) {
      a.db = gp + d->off_b[i];
      a.dbl = mod ? gp + d->off_bl[i] : nullptr;
    }
    if (top) {
      a.Xs = dout; a.ldxs = d->out_dim; a.ks = d->out_dim;
      a.Ws = params + d->off_wf; a.ws_sf = 1; a.ws_sk = H + L;   // element (f, o) = Wf[o][f]
    } else {
      a.G = gb + slot * d->gb_z[i + 1]; a.ldg = HP;
      a.WT = pk + d->pk_wt[i + 1]; a.Kred = HP; a.rowsWT = HP;
    }
    if (first) {
      a.coords = coords_t; a.in_dim = d->in_dim;
      a.W0 = params + d->off_w[0];
      a.dW0 = wgrads ? gp + d->off_w[0] : nullptr;
      a.dcoords = dcoords;
      if (!wgrads) { a.db = nullptr; }
    }
    int rc;
    // FIRST kernels need db != NULL to enter the reduction block that also produces dW0; with frozen weights
    // the whole block is skipped (dcoords is accumulated in the per-sample loop above it).
    if (top && first) rc = mod ? launch_bwd<T, true, true, true>(a, grid, s) : launch_bwd<T, false, true, true>(a, grid, s);
    else if (top) rc = mod ? launch_bwd<T, true, true, false>(a, grid, s) : launch_bwd<T, false, true, false>(a, grid, s);
    else if (first) rc = mod ? launch_bwd<T, true, false, true>(a, grid, s) : launch_bwd<T, false, false, true>(a, grid, s);
    else rc = mod ? launch_bwd<T, true, false, false>(a, grid, s) : launch_bwd<T, false, false, false>(a, grid, s);
    if (rc) return rc;
  }

  // ---- d lat ----
  if (ng != nullptr && (phases & CINEMA_BWD_PHASE_DLAT)) {
    // d (node rows): k_node_grad left its parts (one per workgroup kind); the tail sums them into d table / d cond
    if (!wgrads || dlat_only) return ng_tail(nullptr, false, true);
  } else if (phases & CINEMA_BWD_PHASE_DLAT) {
    DlatArgs a = {};
    int ns = 0;
    for (int i = 0; i < nl; ++i)
      if ((d->modulated_mask >> i) & 1u) {
        a.seg[ns].A = pk + d->pk_wlsT[i]; a.seg[ns].B = gb + slot * d->gb_s[i]; a.seg[ns].K = HP; ++ns;
        a.seg[ns].A = pk + d->pk_wltT[i]; a.seg[ns].B = gb + slot * d->gb_t[i]; a.seg[ns].K = HP; ++ns;
      }
    a.nseg = ns;
    a.Xs = dout; a.ldxs = d->out_dim; a.ks = d->out_dim;
    a.Ws = params + d->off_wf + H; a.ws_sf = 1; a.ws_sk = H + L;
    a.lat = L; a.LP = LP; a.ldg = HP; a.out = dlat;
    bool wide = false;
    if constexpr (std::is_same<T, bf16_t>::value) {
      if (LP >= 192 && Npad >= 4096 && ns > 0 && HP % DL256_BK == 0) {
        wide = true;
        a.tiles_f = (LP + 255) / 256;
        CINEMA_SMEM_ATTR(k_dlat256, DL256_SMEM);
        hipLaunchKernelGGL(k_dlat256, dim3(a.tiles_f * (int)(Npad / 128)), dim3(DL256_THREADS), DL256_SMEM, s, a);
        CINEMA_CHECK_HIP(hipGetLastError());
      }
    }
    if (!wide) {
      a.tiles_f = (LP + 127) / 128;
      const int g2 = a.tiles_f * (int)(Npad / 128);
      constexpr int smem = Engine<T>::NT_SMEM;
      CINEMA_SMEM_ATTR(k_dlat<T>, smem);
      hipLaunchKernelGGL((k_dlat<T>), dim3(g2), dim3(WG_THREADS), smem, s, a);
      CINEMA_CHECK_HIP(hipGetLastError());
    }
  }
  if (!wgrads) return 0;
  if (ng != nullptr && dlat_only) return 0;

  // ---- final_linear: dOut -> [Npad x 16] T copy (operand of the grouped launch) + bias gradient ----
  T *dout16 = gb + slot * (d->n_gbuf - 1);
  if ((phases & CINEMA_BWD_PHASE_FINAL_DW) && !dout_prep_done) {
    FinalBwdArgs a = {};
    a.dout = dout; a.out_dim = d->out_dim; a.dbf = all_phases ? gp + d->off_bf : nullptr; a.Npad = (int)Npad;
    const int grid_f = (int)std::min<int64_t>((Npad + WG_THREADS - 1) / WG_THREADS, 256);
    hipLaunchKernelGGL((k_dout_prep<T>), dim3(grid_f), dim3(WG_THREADS), 0, s, a, dout16);
    CINEMA_CHECK_HIP(hipGetLastError());
  }
  // ---- weight gradients: one grouped launch ----
  if (phases & CINEMA_BWD_PHASE_DW) {
    DwArgs a = {};
    int np = 0, tiles = 0;
    auto add = [&](const void *P, const void *Q, float *out, int ldp, int ldq, int ldo, int M, int K, int rows = 0) {
      DwProb p = {P, Q, out, ldp, ldq, ldo, M, K, (K + 127) / 128, rows ? rows : (int)Npad};
      a.p[np] = p;
      a.tile_start[np] = tiles;
      tiles += ((M + 127) / 128) * p.tk;
      ++np;
    };
    // node gradients: the linear_lats (and final_linear latent-column) problems reduce over the S*32 node rows
    // (P = dU, Q = node rows of the table) instead of the N sample rows (P = ds / dt, Q = lat)
    const T *ngT = ng ? (const T *)((const char *)ng->work + ngp.T) : nullptr;
    const T *ngU = ng ? (const T *)((const char *)ng->work + ngp.dU) : nullptr;
    const int ngF = ngp.F, ngR = (int)ngp.Rp;
    int jm = ngp.n_mod;
    for (int i = nl - 1; i >= 0; --i) {
      if (i > 0) add(gb + slot * d->gb_z[i], sv + slot * d->sv_y[i - 1], gp + d->off_w[i], HP, HP, H, H, H);
      if ((d->modulated_mask >> i) & 1u) {
        if (ng != nullptr) {
          --jm;
          add(ngU + (size_t)jm * 2 * HP, ngT, gp + d->off_wl[i], ngF, LP, L, H, L, ngR);
          add(ngU + (size_t)jm * 2 * HP + HP, ngT, gp + d->off_wl[i] + (size_t)H * L, ngF, LP, L, H, L, ngR);
        } else {
          add(gb + slot * d->gb_s[i], lat, gp + d->off_wl[i], HP, LP, L, H, L);
          add(gb + slot * d->gb_t[i], lat, gp + d->off_wl[i] + (size_t)H * L, HP, LP, L, H, L);
        }
      }
    }
    // final_linear.weight [out x (H + L)]: columns [0,H) against x_L, columns [H,H+L) against lat
    add(dout16, sv + slot * d->sv_y[nl - 1], gp + d->off_wf, 16, HP, H + L, d->out_dim, H);
    if (ng != nullptr)
      add((const T *)((const char *)ng->work + ngp.dUo16), ngT, gp + d->off_wf + H, 16, LP, H + L, d->out_dim, L, ngR);
    else
      add(dout16, lat, gp + d->off_wf + H, 16, LP, H + L, d->out_dim, L);
    a.tile_start[np] = tiles;
    a.nprob = np;
    a.Npad = (int)Npad;
    if constexpr (std::is_same<T, bf16_t>::value) {
      if (d->gb_extra_bytes > 0 && Npad >= 4096) {
        // 256 x 256 tiles, one workgroup per CU; partial tiles go to the slab region behind the gbuf tensors
        Dw256Args b = {};
        int t256 = 0, t_main = 0, t_node = 0;
        for (int q = 0; q < np; ++q) {
          b.g.p[q] = a.p[q];
          b.g.p[q].tk = (a.p[q].K + 255) / 256;
          b.g.tile_start[q] = t256;
          const int tq = ((a.p[q].M + 255) / 256) * b.g.p[q].tk;
          t256 += tq;
          (a.p[q].rows == (int)Npad ? t_main : t_node) += tq;
        }
        b.g.tile_start[np] = t256;
        b.g.nprob = np;
        b.g.Npad = (int)Npad;
        // one workgroup per CU: the problems over the sample rows are split, the (12 x shorter) ones over node rows are not
        const int ncu = cinema_cu_count();
        int sk = std::max(1, std::min(DW256_MAX_SPLIT, (ncu - t_node) / std::max(t_main, 1)));
        sk = std::min(sk, std::max(1, (int)(Npad / DW256_BN) / 4));
        b.g.splitk = sk;
        int nwg = 0;
        for (int q = 0; q < np; ++q) {
          DwProb &pq = b.g.p[q];
          pq.nsplit = pq.rows == (int)Npad ? sk : 1;
          pq.wg0 = nwg; pq.slab0 = nwg;
          nwg += (b.g.tile_start[q + 1] - b.g.tile_start[q]) * pq.nsplit;
        }
        CINEMA_REQUIRE((int64_t)nwg * 256 * 256 * 4 + 256 <= d->gb_extra_bytes,
                       "cinema_siren_backward: gb_extra_bytes of the descriptor is smaller than the plan's");
        const size_t gb_bytes = ((size_t)d->n_gbuf * Npad * HP * sizeof(T) + 255) / 256 * 256;
        b.slab = (float *)((char *)gbuf + gb_bytes);
        CINEMA_SMEM_ATTR(k_dw256, DW256_SMEM);
        hipLaunchKernelGGL(k_dw256, dim3(nwg), dim3(DW256_THREADS), DW256_SMEM, s, b);
        CINEMA_CHECK_HIP(hipGetLastError());
        if (ng != nullptr) return ng_tail(&b, true, !weights_first);
        if (pending) {
          *pending = b;
        } else {
          hipLaunchKernelGGL(k_dw256_reduce, dim3(t256 * 32), dim3(256), 0, s, b);
          CINEMA_CHECK_HIP(hipGetLastError());
        }
        (void)N;
        return 0;
      }
    }
    const int ntile = (int)(Npad / Engine<T>::BN);
    int splitk = std::max(1, 512 / std::max(tiles, 1));
    splitk = std::min(splitk, std::max(1, ntile / 4));
    a.splitk = splitk;
    constexpr int smem = Engine<T>::TN_SMEM;
    CINEMA_SMEM_ATTR(k_dw_grouped<T>, smem);
    hipLaunchKernelGGL((k_dw_grouped<T>), dim3(tiles * splitk), dim3(WG_THREADS), smem, s, a);
    CINEMA_CHECK_HIP(hipGetLastError());
    if (ng != nullptr) return ng_tail(nullptr, true, !weights_first);
  }
  (void)N;
  return 0;
}

extern "C" int cinema_siren_backward(const cinema_siren_desc *d, const float *params, const void *packed,
                                     const float *coords_t, const void *lat, const void *saved, const float *dout,
                                     int64_t N, int64_t Npad, void *gbuf, float *grad_params, float *dlat,
                                     float *dcoords, int32_t flags, void *stream) {
  CINEMA_REQUIRE(d && params && packed && coords_t && lat && saved && dout && gbuf && dlat && dcoords,
                 "cinema_siren_backward: null argument");
  CINEMA_REQUIRE((flags & CINEMA_BWD_SKIP_WEIGHT_GRADS) || grad_params, "cinema_siren_backward: grad_params is null");
  CINEMA_REQUIRE(N >= 1 && Npad >= N && Npad % 128 == 0, "cinema_siren_backward: need 1 <= N <= Npad, Npad %% 128 == 0");
  hipStream_t s = (hipStream_t)stream;
  return d->precision == CINEMA_PREC_BF16
             ? backward_impl<bf16_t>(d, params, packed, coords_t, lat, saved, dout, N, Npad, gbuf, grad_params, dlat,
                                     dcoords, flags, s)
             : backward_impl<float>(d, params, packed, coords_t, lat, saved, dout, N, Npad, gbuf, grad_params, dlat,
                                    dcoords, flags, s);
}

extern "C" int cinema_siren_backward_deferred(const cinema_siren_desc *d, const float *params, const void *packed,
                                              const float *coords_t, const void *lat, const void *saved, const float *dout,
                                              int64_t N, int64_t Npad, void *gbuf, float *grad_params, float *dlat,
                                              float *dcoords, int32_t flags, void *pending, void *stream) {
  CINEMA_REQUIRE(d && params && packed && coords_t && lat && saved && dout && gbuf && dlat && dcoords && pending,
                 "cinema_siren_backward_deferred: null argument");
  CINEMA_REQUIRE((flags & CINEMA_BWD_SKIP_WEIGHT_GRADS) || grad_params, "cinema_siren_backward: grad_params is null");
  CINEMA_REQUIRE(N >= 1 && Npad >= N && Npad % 128 == 0, "cinema_siren_backward: need 1 <= N <= Npad, Npad %% 128 == 0");
  static_assert(sizeof(Dw256Args) <= CINEMA_PENDING_BYTES, "CINEMA_PENDING_BYTES too small");
  hipStream_t s = (hipStream_t)stream;
  return d->precision == CINEMA_PREC_BF16
             ? backward_impl<bf16_t>(d, params, packed, coords_t, lat, saved, dout, N, Npad, gbuf, grad_params, dlat,
                                     dcoords, flags, s, (Dw256Args *)pending)
             : backward_impl<float>(d, params, packed, coords_t, lat, saved, dout, N, Npad, gbuf, grad_params, dlat,
                                    dcoords, flags, s, (Dw256Args *)pending);
}

// ----------------------------------------------------------------------------------------
// backward with the latent side reduced in node space (nodegrad.h): SIREN backward + K1 backward in one call
// ----------------------------------------------------------------------------------------
extern "C" int cinema_siren_backward_nodegrad(const cinema_siren_desc *d, const float *params, const void *packed,
                                              const float *coords_t, const void *saved, const float *dout, int64_t N,
                                              int64_t Npad, void *gbuf, float *grad_params, float *dcoords, int32_t flags,
                                              const float *coords, const int32_t *idx, const float *embed, int32_t tf_dim,
                                              int64_t S, int32_t C, int32_t lx, int32_t ly, int32_t lz, int32_t cond_dims,
                                              void *work, float *dlatents, float *dembed, float *dcond, void *stream) {
  CINEMA_REQUIRE(d && params && packed && coords_t && saved && dout && gbuf && dcoords && coords && idx && work && dlatents,
                 "cinema_siren_backward_nodegrad: null argument");
  CINEMA_REQUIRE(ng_eligible(d, lx, ly, lz) && fused_mode(d) == 64,
                 "cinema_siren_backward_nodegrad: needs a bf16 stack on the activation-stationary path (hidden <= 512) with a "
                 "modulated layer and a latent grid of at most %d nodes", NG_ONE_SLOT - 1);
  CINEMA_REQUIRE((flags & CINEMA_BWD_SKIP_WEIGHT_GRADS) || grad_params, "cinema_siren_backward_nodegrad: grad_params is null");
  CINEMA_REQUIRE(N >= 1 && Npad >= N && Npad % 128 == 0 && S >= 1, "cinema_siren_backward_nodegrad: need 1 <= N <= Npad, Npad %% 128 == 0, S >= 1");
  CINEMA_REQUIRE(C >= 1 && C + cond_dims == d->lat, "cinema_siren_backward_nodegrad: C + cond_dims = %d != lat_size %d", C + cond_dims, d->lat);
  CINEMA_REQUIRE(dembed == nullptr || (embed != nullptr && (tf_dim == 6 || tf_dim == 9 || tf_dim == 12)),
                 "cinema_siren_backward_nodegrad: dembed needs the embedding table (width 6, 9 or 12)");
  CINEMA_REQUIRE(dcond == nullptr || cond_dims > 0, "cinema_siren_backward_nodegrad: dcond without condition columns");
  const NgCtx ng = {work, coords, idx, embed, tf_dim, S, C, lx, ly, lz, cond_dims, dlatents, dembed, dcond};
  return backward_impl<bf16_t>(d, params, packed, coords_t, nullptr, saved, dout, N, Npad, gbuf, grad_params, nullptr, dcoords,
                               flags, (hipStream_t)stream, nullptr, nullptr, &ng);
}

// ----------------------------------------------------------------------------------------
// "nodes" path entry points (nodes.h): the chain kernels read the FiLM modulation from per-subject node tables.
// ----------------------------------------------------------------------------------------
extern "C" int cinema_siren_forward_nodes(const cinema_siren_desc *d, const float *params, const void *packed,
                                          const float *coords_t, const void *wrows, const void *tiles, const void *U,
                                          const void *V, int64_t S, int64_t N, int64_t Npad, void *saved, float *out,
                                          int32_t flags, void *stream) {
  CINEMA_REQUIRE(d && params && packed && coords_t && wrows && tiles && U && V && out, "cinema_siren_forward_nodes: null argument");
  CINEMA_REQUIRE(fused_mode(d) == 64 && d->pk_ns >= 0, "cinema_siren_forward_nodes: needs a bf16 stack on the activation-stationary path "
                 "(hidden <= 512)");
  CINEMA_REQUIRE(N >= 1 && Npad >= N && Npad % 128 == 0 && S >= 1, "cinema_siren_forward_nodes: need 1 <= N <= Npad, Npad %% 128 == 0, S >= 1");
  return nodes_forward(d, params, packed, coords_t, wrows, tiles, U, V, S, N, Npad, saved, out, (flags & CINEMA_NODES_ROUND_S) ? 1 : 0,
                       (hipStream_t)stream);
}

extern "C" int cinema_siren_backward_nodes(const cinema_siren_desc *d, const float *params, const void *packed,
                                           const float *coords_t, const void *lat, const void *wrows, const void *tiles,
                                           const void *U, int64_t S, const void *saved, const float *dout, int64_t N,
                                           int64_t Npad, void *gbuf, float *grad_params, float *dlat, float *dcoords,
                                           int32_t flags, void *pending, void *stream) {
  CINEMA_REQUIRE(d && params && packed && coords_t && lat && wrows && tiles && U && saved && dout && gbuf && dlat && dcoords,
                 "cinema_siren_backward_nodes: null argument");
  CINEMA_REQUIRE(fused_mode(d) == 64 && d->pk_ns >= 0, "cinema_siren_backward_nodes: needs a bf16 stack on the activation-stationary path");
  CINEMA_REQUIRE((flags & CINEMA_BWD_SKIP_WEIGHT_GRADS) || grad_params, "cinema_siren_backward_nodes: grad_params is null");
  CINEMA_REQUIRE(N >= 1 && Npad >= N && Npad % 128 == 0 && S >= 1, "cinema_siren_backward_nodes: need 1 <= N <= Npad, Npad %% 128 == 0, S >= 1");
  const NodesCtx nc = {wrows, tiles, U, S, (flags & CINEMA_NODES_ROUND_S) ? 1 : 0};
  return backward_impl<bf16_t>(d, params, packed, coords_t, lat, saved, dout, N, Npad, gbuf, grad_params, dlat, dcoords,
                               flags & ~CINEMA_NODES_ROUND_S, (hipStream_t)stream, (Dw256Args *)pending, &nc);
}

// ----------------------------------------------------------------------------------------
// stand-alone SineLayer.forward / backward (models/siren.py:26-36) over the layer-by-layer kernels
// ----------------------------------------------------------------------------------------
namespace {
struct SineLayerPlan {
  int H, HP, K, KP, L, LP; bool small, mod; long Npad; size_t esz;
  // element offsets (of T) into the work buffer, then byte offsets of the fp32 regions
  long W, WT, Wls, Wlt, WlsT, WltT, eye, X, LAT, Y, C, Z, S, G, DZ, DS, DT, t_end;
  size_t xs, dxp, dlp, bytes;
};

bool sine_layer_plan(const cinema_sine_layer_desc *d, int64_t N, SineLayerPlan &p) {
  if (!d || N < 1 || N > (1L << 27) || d->in_feat < 1 || d->out_feat < 16 || d->out_feat % 16 || d->lat_feat < 0) return false;
  if (d->in_feat > 8192 || d->out_feat > 8192 || d->lat_feat > 8192) return false;
  if (d->precision != CINEMA_PREC_F32 && d->precision != CINEMA_PREC_BF16) return false;
  p.H = d->out_feat; p.HP = round_up(p.H, 64);
  p.K = d->in_feat; p.small = p.K <= 3; p.KP = round_up(p.K, 64);
  p.L = d->lat_feat; p.mod = p.L > 0; p.LP = p.mod ? round_up(p.L, 64) : 0;
  p.Npad = (N + 127) / 128 * 128;
  p.esz = d->precision == CINEMA_PREC_BF16 ? 2 : 4;
  long o = 0;
  auto take = [&](long n) { const long at = o; o += (n + 127) / 128 * 128; return at; };
  const long HP = p.HP, KP = p.KP, LP = p.LP, NP = p.Npad;
  p.W = take(p.small ? 0 : HP * KP); p.WT = take(p.small ? 0 : KP * HP);
  p.Wls = take(HP * LP); p.Wlt = take(HP * LP); p.WlsT = take(LP * HP); p.WltT = take(LP * HP);
  p.eye = take(HP * HP);
  p.X = take(p.small ? 0 : NP * KP); p.LAT = take(NP * LP);
  p.Y = take(NP * HP); p.C = take(NP * HP); p.Z = take(p.mod ? NP * HP : 0); p.S = take(p.mod ? NP * HP : 0);
  p.G = take(NP * HP); p.DZ = take(NP * HP); p.DS = take(p.mod ? NP * HP : 0); p.DT = take(p.mod ? NP * HP : 0);
  p.t_end = o;
  size_t b = (size_t)o * p.esz;
  auto takeb = [&](size_t n) { const size_t at = b; b += (n + 255) / 256 * 256; return at; };
  p.xs = takeb(p.small ? (size_t)NP * p.K * 4 : 0);
  p.dxp = takeb((size_t)NP * (p.small ? p.K : KP) * 4);
  p.dlp = takeb((size_t)NP * LP * 4);
  p.bytes = b;
  return true;
}

template <typename T>
int sine_layer_forward_impl(const cinema_sine_layer_desc *d, const SineLayerPlan &p, const float *x, const float *latent,
                            const float *W, const float *b, const float *Wl, const float *bl, int64_t N, void *work,
                            float *out, hipStream_t s) {
  T *wk = (T *)work;
  PackArgs a;
  a.packed = work;
  int nj = 0;
  auto add = [&](const float *src, long dst, long rows, int cols, int ld, long rowsP, int colsP, int tr) {
    PackJob j = {src, dst, (int)rows, cols, ld, (int)rowsP, colsP, tr};
    a.job[nj++] = j;
  };
  if (!p.small) {
    add(W, p.W, p.H, p.K, p.K, p.HP, p.KP, 0);          // [HP x KP]: forward operand
    add(W, p.WT, p.H, p.K, p.K, p.KP, p.HP, 1);         // [KP x HP]: d input = dz W
    add(x, p.X, N, p.K, p.K, p.Npad, p.KP, 0);
  }
  if (p.mod) {
    add(Wl, p.Wls, p.H, p.L, p.L, p.HP, p.LP, 0);
    add(Wl + (size_t)p.H * p.L, p.Wlt, p.H, p.L, p.L, p.HP, p.LP, 0);
    add(Wl, p.WlsT, p.H, p.L, p.L, p.LP, p.HP, 1);
    add(Wl + (size_t)p.H * p.L, p.WltT, p.H, p.L, p.L, p.LP, p.HP, 1);
    add(latent, p.LAT, N, p.L, p.L, p.Npad, p.LP, 0);
  }
  if (nj) {
    a.njobs = nj;
    hipLaunchKernelGGL((k_pack<T>), dim3(64, nj), dim3(256), 0, s, a);
    CINEMA_CHECK_HIP(hipGetLastError());
  }
  float *xs = (float *)((char *)work + p.xs);
  if (p.small) {
    CINEMA_CHECK_HIP(hipMemsetAsync(xs, 0, (size_t)p.Npad * p.K * 4, s));
    CINEMA_CHECK_HIP(hipMemcpyAsync(xs, x, (size_t)N * p.K * 4, hipMemcpyDeviceToDevice, s));
  }
  FwdArgs f = {};
  f.bias = b; f.H = p.H; f.HP = p.HP; f.ldo = p.HP; f.tiles_f = (p.HP + 127) / 128; f.omega = d->omega;
  f.Y = wk + p.Y; f.C = wk + p.C; f.rowsW = p.HP;
  if (p.mod) {
    f.LAT = wk + p.LAT; f.LP = p.LP; f.Wls = wk + p.Wls; f.Wlt = wk + p.Wlt; f.bias_l = bl;
    f.Z = wk + p.Z; f.S = wk + p.S;
  }
  const int grid = f.tiles_f * (int)(p.Npad / 128);
  int rc;
  if (p.small) {
    f.Xs = xs; f.ldxs = p.K; f.ks = p.K; f.Ws = W; f.ws_sf = p.K; f.ws_sk = 1;
    rc = p.mod ? launch_fwd<T, true, true>(f, grid, s) : launch_fwd<T, false, true>(f, grid, s);
  } else {
    f.X = wk + p.X; f.ldx = p.KP; f.W = wk + p.W; f.K = p.KP;
    rc = p.mod ? launch_fwd<T, true, false>(f, grid, s) : launch_fwd<T, false, false>(f, grid, s);
  }
  if (rc) return rc;
  const unsigned ug = (unsigned)std::min<long>((N * p.H + 255) / 256, 4096);
  hipLaunchKernelGGL((k_unpad<T>), dim3(ug), dim3(256), 0, s, (const T *)(wk + p.Y), p.HP, out, (long)N, p.H);
  CINEMA_CHECK_HIP(hipGetLastError());
  return 0;
}

template <typename T>
int sine_layer_backward_impl(const cinema_sine_layer_desc *d, const SineLayerPlan &p, const float *W, int64_t N, void *work,
                             const float *dout, float *dx, float *dlatent, float *dW, float *db, float *dWl, float *dbl,
                             hipStream_t s) {
  T *wk = (T *)work;
  float *xs = (float *)((char *)work + p.xs), *dxp = (float *)((char *)work + p.dxp), *dlp = (float *)((char *)work + p.dlp);
  CINEMA_CHECK_HIP(hipMemsetAsync(dW, 0, (size_t)p.H * p.K * 4, s));
  CINEMA_CHECK_HIP(hipMemsetAsync(db, 0, (size_t)p.H * 4, s));
  if (p.mod) {
    CINEMA_CHECK_HIP(hipMemsetAsync(dWl, 0, (size_t)2 * p.H * p.L * 4, s));
    CINEMA_CHECK_HIP(hipMemsetAsync(dbl, 0, (size_t)2 * p.H * 4, s));
  }
  if (p.small) CINEMA_CHECK_HIP(hipMemsetAsync(dxp, 0, (size_t)p.Npad * p.K * 4, s));
  {
    PackArgs a;
    a.packed = work;
    PackJob j = {dout, p.G, (int)N, p.H, p.H, (int)p.Npad, p.HP, 0};
    a.job[0] = j;
    a.njobs = 1;
    hipLaunchKernelGGL((k_pack<T>), dim3(64, 1), dim3(256), 0, s, a);
    CINEMA_CHECK_HIP(hipGetLastError());
    hipLaunchKernelGGL((k_eye<T>), dim3(256), dim3(256), 0, s, wk + p.eye, p.HP);
    CINEMA_CHECK_HIP(hipGetLastError());
  }
  BwdArgs a = {};
  a.G = wk + p.G; a.ldg = p.HP; a.WT = wk + p.eye; a.Kred = p.HP; a.rowsWT = p.HP;     // dy itself: identity operand
  a.C = wk + p.C; a.ldo = p.HP; a.HP = p.HP; a.H = p.H; a.omega = d->omega; a.tiles_f = (p.HP + 127) / 128;
  a.DZ = wk + p.DZ; a.db = db;
  if (p.mod) { a.Z = wk + p.Z; a.S = wk + p.S; a.DS = wk + p.DS; a.DT = wk + p.DT; a.dbl = dbl; }
  if (p.small) { a.coords = xs; a.in_dim = p.K; a.W0 = W; a.dW0 = dW; a.dcoords = dxp; }
  const int grid = a.tiles_f * (int)(p.Npad / 128);
  int rc;
  if (p.small) rc = p.mod ? launch_bwd<T, true, false, true>(a, grid, s) : launch_bwd<T, false, false, true>(a, grid, s);
  else rc = p.mod ? launch_bwd<T, true, false, false>(a, grid, s) : launch_bwd<T, false, false, false>(a, grid, s);
  if (rc) return rc;
  auto dlat_gemm = [&](const NtSeg *segs, int ns, int cols, int colsP, float *outp) -> int {
    DlatArgs g = {};
    for (int q = 0; q < ns; ++q) g.seg[q] = segs[q];
    g.nseg = ns; g.ks = 0; g.lat = cols; g.LP = colsP; g.ldg = p.HP; g.out = outp;
    g.tiles_f = (colsP + 127) / 128;
    constexpr int smem = Engine<T>::NT_SMEM;
    CINEMA_SMEM_ATTR(k_dlat<T>, smem);
    hipLaunchKernelGGL((k_dlat<T>), dim3(g.tiles_f * (int)(p.Npad / 128)), dim3(WG_THREADS), smem, s, g);
    CINEMA_CHECK_HIP(hipGetLastError());
    return 0;
  };
  auto unpad = [&](const float *src, int ld, float *dst, int cols) -> int {
    const unsigned ug = (unsigned)std::min<long>((N * cols + 255) / 256, 4096);
    hipLaunchKernelGGL((k_unpad<float>), dim3(ug), dim3(256), 0, s, src, ld, dst, (long)N, cols);
    CINEMA_CHECK_HIP(hipGetLastError());
    return 0;
  };
  if (p.small) {
    if (int r2 = unpad(dxp, p.K, dx, p.K)) return r2;
  } else {
    NtSeg sg = {wk + p.WT, wk + p.DZ, p.HP};
    if (int r2 = dlat_gemm(&sg, 1, p.K, p.KP, dxp)) return r2;
    if (int r2 = unpad(dxp, p.KP, dx, p.K)) return r2;
  }
  if (p.mod) {
    NtSeg sg[2] = {{wk + p.WlsT, wk + p.DS, p.HP}, {wk + p.WltT, wk + p.DT, p.HP}};
    if (int r2 = dlat_gemm(sg, 2, p.L, p.LP, dlp)) return r2;
    if (int r2 = unpad(dlp, p.LP, dlatent, p.L)) return r2;
  }
  DwArgs w = {};
  int np = 0, tiles = 0;
  auto add = [&](const void *P, const void *Q, float *o, int ldq, int ldo, int K) {
    DwProb pr = {P, Q, o, p.HP, ldq, ldo, p.H, K, (K + 127) / 128};
    w.p[np] = pr;
    w.tile_start[np] = tiles;
    tiles += ((p.H + 127) / 128) * pr.tk;
    ++np;
  };
  if (!p.small) add(wk + p.DZ, wk + p.X, dW, p.KP, p.K, p.K);
  if (p.mod) {
    add(wk + p.DS, wk + p.LAT, dWl, p.LP, p.L, p.L);
    add(wk + p.DT, wk + p.LAT, dWl + (size_t)p.H * p.L, p.LP, p.L, p.L);
  }
  if (np) {
    w.tile_start[np] = tiles; w.nprob = np; w.Npad = (int)p.Npad;
    const int ntile = (int)(p.Npad / Engine<T>::BN);
    w.splitk = std::min(std::max(1, 512 / std::max(tiles, 1)), std::max(1, ntile / 4));
    constexpr int smem = Engine<T>::TN_SMEM;
    CINEMA_SMEM_ATTR(k_dw_grouped<T>, smem);
    hipLaunchKernelGGL((k_dw_grouped<T>), dim3(tiles * w.splitk), dim3(WG_THREADS), smem, s, w);
    CINEMA_CHECK_HIP(hipGetLastError());
  }
  return 0;
}
}  // namespace

extern "C" int64_t cinema_sine_layer_work_bytes(const cinema_sine_layer_desc *d, int64_t N) {
  SineLayerPlan p;
  if (!sine_layer_plan(d, N, p)) {
    cinema_set_error("cinema_sine_layer_work_bytes: need 1 <= N <= 2^27, in_feat >= 1, out_feat a multiple of 16, lat_feat >= 0 (all <= 8192), a valid precision");
    return -1;
  }
  return (int64_t)p.bytes;
}

extern "C" int cinema_sine_layer_forward(const cinema_sine_layer_desc *d, const float *x, const float *latent, const float *W,
                                         const float *b, const float *Wl, const float *bl, int64_t N, void *work, float *out,
                                         void *stream) {
  SineLayerPlan p;
  CINEMA_REQUIRE(sine_layer_plan(d, N, p), "cinema_sine_layer_forward: bad descriptor or N (see cinema_sine_layer_work_bytes)");
  CINEMA_REQUIRE(x && W && b && work && out, "cinema_sine_layer_forward: null argument");
  CINEMA_REQUIRE(!p.mod || (latent && Wl && bl), "cinema_sine_layer_forward: a modulated layer needs latent, linear_lats weight and bias");
  hipStream_t s = (hipStream_t)stream;
  return d->precision == CINEMA_PREC_BF16 ? sine_layer_forward_impl<bf16_t>(d, p, x, latent, W, b, Wl, bl, N, work, out, s)
                                          : sine_layer_forward_impl<float>(d, p, x, latent, W, b, Wl, bl, N, work, out, s);
}

extern "C" int cinema_sine_layer_backward(const cinema_sine_layer_desc *d, const float *W, int64_t N, void *work,
                                          const float *dout, float *dx, float *dlatent, float *dW, float *db, float *dWl,
                                          float *dbl, void *stream) {
  SineLayerPlan p;
  CINEMA_REQUIRE(sine_layer_plan(d, N, p), "cinema_sine_layer_backward: bad descriptor or N (see cinema_sine_layer_work_bytes)");
  CINEMA_REQUIRE(W && work && dout && dx && dW && db, "cinema_sine_layer_backward: null argument");
  CINEMA_REQUIRE(!p.mod || (dlatent && dWl && dbl), "cinema_sine_layer_backward: a modulated layer needs dlatent, dWl and dbl");
  hipStream_t s = (hipStream_t)stream;
  return d->precision == CINEMA_PREC_BF16 ? sine_layer_backward_impl<bf16_t>(d, p, W, N, work, dout, dx, dlatent, dW, db, dWl, dbl, s)
                                          : sine_layer_backward_impl<float>(d, p, W, N, work, dout, dx, dlatent, dW, db, dWl, dbl, s);
}

// ----------------------------------------------------------------------------------------
// Modulator with cnn_kernel_size = k > 0 (models/inr_decoder.py:109-123): Conv3d(C, C, k, padding='same') over the
// latent table (S, C, lx, ly, lz).  The table is tiny in space (G = lx*ly*lz nodes) and wide in channels, so the
// convolution is three GEMMs over an im2col image, all through the split-K weight-gradient engine (k_dw_grouped):
//   forward  outT[co][(s,n)]   = sum_{(tap,ci)} W2T[(tap,ci)][co] * colT[(tap,ci)][(s,n)]        colT = im2col(in, +)
//   d weight dW2[co][(tap,ci)] = sum_{(s,n)}    doutR[(s,n)][co]  * col[(s,n)][(tap,ci)]         col  = im2col(in, +)
//   d input  dinT[ci][(s,m)]   = sum_{(tap,co)} W3T[(tap,co)][ci] * dcolT[(tap,co)][(s,m)]       dcolT = im2col(dout, -)
// with col element ((s,n), (tap,c)) = x[s][c][n +- offset(tap)] or 0 outside the grid.
// ----------------------------------------------------------------------------------------
namespace {
struct ConvGeom { int S, C, lx, ly, lz, k, G, K3; long R, KK, KKp; };

__device__ __forceinline__ int conv_neighbour(const ConvGeom &g, int n, int tap, int sign) {
  const int p = (g.k - 1) >> 1;      // padding='same': (k - 1) / 2 in front, the rest behind (torch; equal for odd k)
  const int tz = tap % g.k, ty = (tap / g.k) % g.k, tx = tap / (g.k * g.k);
  const int iz = n % g.lz, iy = (n / g.lz) % g.ly, ix = n / (g.lz * g.ly);
  const int jx = ix + sign * (tx - p), jy = iy + sign * (ty - p), jz = iz + sign * (tz - p);
  if (jx < 0 || jx >= g.lx || jy < 0 || jy >= g.ly || jz < 0 || jz >= g.lz) return -1;
  return (jx * g.ly + jy) * g.lz + jz;
}

// TRANSPOSED = false: col [R][KKp] (channel fastest); true: colT [KKp][R] ((s, n) fastest).  Pads are written as zeros.
// A thread makes eight consecutive destination elements (C % 8 == 0, R % 128 == 0): all eight gathers are requested before
// the first conversion, one 16-byte (bf16) or two (fp32) stores -- one 2-byte store per thread took 62 us per call at the
// cfg-D table.  `zero` (n_zero floats): the split-K GEMM that follows accumulates with atomics; its output is cleared here
// instead of by a fill launch of its own.
template <typename T, bool TRANSPOSED>
__global__ __launch_bounds__(256) void k_conv_im2col(const float *x, ConvGeom g, int sign, T *dst, float *zero, long n_zero) {
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n_zero / 4; e += (long)gridDim.x * 256) {
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    *reinterpret_cast<f32x4 *>(zero + 4 * e) = z;
  }
  const long total8 = g.R * g.KKp / 8;
  const long SG = (long)g.S * g.G;
  for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < total8; q += (long)gridDim.x * 256) {
    float v[8];
    bool ok[8];
    if (TRANSPOSED) {
      const long kc = q / (g.R / 8), row0 = (q - kc * (g.R / 8)) * 8;       // eight consecutive (s, n) rows of one (tap, c)
      const int tap = (int)(kc / g.C), c = (int)(kc % g.C);
      // the tap's offset and the first row's node coordinates once, then a carry chain (ten integer divisions per element
      // made this variant VALU-bound: 55 us)
      const int p = (g.k - 1) >> 1;
      const int oz = sign * (tap % g.k - p), oy = sign * ((tap / g.k) % g.k - p), ox = sign * (tap / (g.k * g.k) - p);
      int s = (int)(row0 / g.G), n = (int)(row0 % g.G);
      int iz = n % g.lz, iy = (n / g.lz) % g.ly, ix = n / (g.lz * g.ly);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int jx = ix + ox, jy = iy + oy, jz = iz + oz;
        ok[j] = row0 + j < SG && kc < g.KK && jx >= 0 && jx < g.lx && jy >= 0 && jy < g.ly && jz >= 0 && jz < g.lz;
        const int nb = (jx * g.ly + jy) * g.lz + jz;
        v[j] = x[((long)(ok[j] ? s : 0) * g.C + (ok[j] ? c : 0)) * g.G + (ok[j] ? nb : 0)];
        if (++iz == g.lz) { iz = 0; if (++iy == g.ly) { iy = 0; if (++ix == g.lx) { ix = 0; ++s; } } }
      }
    } else {
      const long row = q / (g.KKp / 8), kc0 = (q - row * (g.KKp / 8)) * 8;    // eight consecutive channels of one (row, tap)
      const int s = (int)(row / g.G), n = (int)(row % g.G), tap = (int)(kc0 / g.C), c0 = (int)(kc0 % g.C);
      const int nb = (row < SG && kc0 < g.KK) ? conv_neighbour(g, n, tap, sign) : -1;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        ok[j] = nb >= 0;
        v[j] = x[((long)(ok[j] ? s : 0) * g.C + (ok[j] ? c0 + j : 0)) * g.G + (ok[j] ? nb : 0)];
      }
    }
    T o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (T)(ok[j] ? v[j] : 0.f);
    T *out = dst + q * 8;
    *reinterpret_cast<u32x4 *>(out) = *reinterpret_cast<const u32x4 *>(o);
    if constexpr (sizeof(T) == 4) *reinterpret_cast<u32x4 *>(out + 4) = *reinterpret_cast<const u32x4 *>(o + 4);
  }
}

// W [co][ci][K3] -> dst [(tap, a)][b] with (a, b) = (ci, co) (forward operand) or (co, ci) (d input operand); pads zero
template <typename T>
__global__ __launch_bounds__(256) void k_conv_pack_w(const float *W, ConvGeom g, int for_dinput, T *dst) {
  const long total = g.KKp * g.C;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const long kc = e / g.C;
    const int b = (int)(e % g.C);
    float v = 0.f;
    if (kc < g.KK) {
      const int tap = (int)(kc / g.C), a = (int)(kc % g.C);
      const int co = for_dinput ? a : b, ci = for_dinput ? b : a;
      v = W[((long)co * g.C + ci) * g.K3 + tap];
    }
    dst[e] = (T)v;
  }
}

// x [S][C][G] fp32 -> rows [R][C] T ((s, n) rows, channel fastest); pad rows zero
template <typename T>
__global__ __launch_bounds__(256) void k_conv_rows(const float *x, ConvGeom g, T *dst) {
  const long total = g.R * g.C;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const long row = e / g.C;
    const int c = (int)(e % g.C);
    float v = 0.f;
    if (row < (long)g.S * g.G) v = x[((row / g.G) * g.C + c) * g.G + row % g.G];
    dst[e] = (T)v;
  }
}

// yT [C][R] fp32 (+ bias[c]) -> y [S][C][G]
__global__ __launch_bounds__(256) void k_conv_untranspose(const float *yT, const float *bias, ConvGeom g, float *y) {
  const long total = (long)g.S * g.C * g.G;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int n = (int)(e % g.G), c = (int)((e / g.G) % g.C);
    const long s = e / ((long)g.G * g.C);
    y[e] = yT[(long)c * g.R + s * g.G + n] + (bias ? bias[c] : 0.f);
  }
}

// dW2 [C][KKp] ((tap, ci) columns) -> dW [co][ci][K3]; db[co] = sum of dout over subjects and nodes (one block per co)
__global__ __launch_bounds__(256) void k_conv_wgrad_finish(const float *dW2, const float *dout, ConvGeom g, float *dW, float *db) {
  const int co = blockIdx.x;
  for (int e = threadIdx.x; e < g.C * g.K3; e += 256) {
    const int ci = e / g.K3, tap = e % g.K3;
    dW[((long)co * g.C + ci) * g.K3 + tap] = dW2[(long)co * g.KKp + (long)tap * g.C + ci];
  }
  float acc = 0.f;
  for (long e = threadIdx.x; e < (long)g.S * g.G; e += 256) acc += dout[((e / g.G) * g.C + co) * g.G + e % g.G];
  __shared__ float red[4];
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) db[co] = red[0] + red[1] + red[2] + red[3];
}

bool conv_geom(int64_t S, int C, int lx, int ly, int lz, int k, ConvGeom &g) {
  if (S < 1 || C < 8 || C % 8 || C > 4096 || lx < 1 || ly < 1 || lz < 1 || k < 1 || k > 15) return false;
  if ((long)lx * ly * lz > 4096 || S * lx * ly * lz > (1L << 24)) return false;     // (s, n) rows are int in the GEMM engine
  g.S = (int)S; g.C = C; g.lx = lx; g.ly = ly; g.lz = lz; g.k = k; g.G = lx * ly * lz; g.K3 = k * k * k;
  g.R = ((long)S * g.G + 127) / 128 * 128;
  g.KK = (long)g.K3 * C;
  g.KKp = (g.KK + 127) / 128 * 128;
  return true;
}

struct ConvWork { size_t colT, col, wT, rows, outT, dW2, bytes; };
ConvWork conv_work(const ConvGeom &g, int precision) {
  const size_t e = precision == CINEMA_PREC_BF16 ? 2 : 4;
  ConvWork w;
  size_t o = 0;
  auto take = [&](size_t n) { const size_t at = o; o += (n + 255) / 256 * 256; return at; };
  w.colT = take((size_t)g.KKp * g.R * e);
  w.col = take((size_t)g.R * g.KKp * e);
  w.wT = take((size_t)g.KKp * g.C * e);
  w.rows = take((size_t)g.R * g.C * e);
  w.outT = take((size_t)g.C * g.R * 4);
  w.dW2 = take((size_t)g.C * g.KKp * 4);
  w.bytes = o;
  return w;
}

// out[M x K] (zeroed here) = sum over `nrows` rows of P[n][0..M) * Q[n][0..K): one k_dw_grouped problem, split over rows
template <typename T>
int conv_gemm(const void *P, int ldp, const void *Q, int ldq, float *out, int ldo, int M, int K, long nrows, hipStream_t s,
              bool out_cleared = false) {
  if (!out_cleared) CINEMA_CHECK_HIP(hipMemsetAsync(out, 0, (size_t)M * ldo * sizeof(float), s));
  DwArgs w = {};
  DwProb pr = {P, Q, out, ldp, ldq, ldo, M, K, (K + 127) / 128};
  w.p[0] = pr;
  const int tiles = ((M + 127) / 128) * pr.tk;
  w.tile_start[0] = 0; w.tile_start[1] = tiles; w.nprob = 1; w.Npad = (int)nrows;
  const int ntile = (int)(nrows / Engine<T>::BN);
  w.splitk = std::min(std::max(1, 1024 / std::max(tiles, 1)), std::max(1, ntile / 4));
  constexpr int smem = Engine<T>::TN_SMEM;
  CINEMA_SMEM_ATTR(k_dw_grouped<T>, smem);
  hipLaunchKernelGGL((k_dw_grouped<T>), dim3(tiles * w.splitk), dim3(WG_THREADS), smem, s, w);
  CINEMA_CHECK_HIP(hipGetLastError());
  return 0;
}

inline unsigned conv_grid(long total) { return (unsigned)std::min<long>((total + 255) / 256, 8192); }

template <typename T>
int conv_forward_impl(const ConvGeom &g, const ConvWork &w, const float *in, const float *W, const float *b, void *work,
                      float *out, hipStream_t s) {
  char *wk = (char *)work;
  T *colT = (T *)(wk + w.colT), *wT = (T *)(wk + w.wT);
  float *outT = (float *)(wk + w.outT);
  hipLaunchKernelGGL((k_conv_im2col<T, true>), dim3(conv_grid(g.R * g.KKp / 8)), dim3(256), 0, s, in, g, 1, colT, outT, (long)g.C * g.R);
  hipLaunchKernelGGL((k_conv_pack_w<T>), dim3(conv_grid(g.KKp * g.C)), dim3(256), 0, s, W, g, 0, wT);
  CINEMA_CHECK_HIP(hipGetLastError());
  if (int rc = conv_gemm<T>(wT, g.C, colT, (int)g.R, outT, (int)g.R, g.C, g.S * g.G, g.KKp, s, true)) return rc;
  hipLaunchKernelGGL(k_conv_untranspose, dim3(conv_grid((long)g.S * g.C * g.G)), dim3(256), 0, s, outT, b, g, out);
  CINEMA_CHECK_HIP(hipGetLastError());
  return 0;
}

template <typename T>
int conv_backward_impl(const ConvGeom &g, const ConvWork &w, const float *in, const float *W, const float *dout, void *work,
                       float *din, float *dW, float *db, hipStream_t s) {
  char *wk = (char *)work;
  T *colT = (T *)(wk + w.colT), *col = (T *)(wk + w.col), *wT = (T *)(wk + w.wT), *rows = (T *)(wk + w.rows);
  float *outT = (float *)(wk + w.outT), *dW2 = (float *)(wk + w.dW2);
  // d weight (+ d bias)
  hipLaunchKernelGGL((k_conv_im2col<T, false>), dim3(conv_grid(g.R * g.KKp / 8)), dim3(256), 0, s, in, g, 1, col, dW2, (long)g.C * g.KKp);
  hipLaunchKernelGGL((k_conv_rows<T>), dim3(conv_grid(g.R * g.C)), dim3(256), 0, s, dout, g, rows);
  CINEMA_CHECK_HIP(hipGetLastError());
  if (int rc = conv_gemm<T>(rows, g.C, col, (int)g.KKp, dW2, (int)g.KKp, g.C, (int)g.KK, g.R, s, true)) return rc;
  hipLaunchKernelGGL(k_conv_wgrad_finish, dim3(g.C), dim3(256), 0, s, dW2, dout, g, dW, db);
  CINEMA_CHECK_HIP(hipGetLastError());
  // d input: the same GEMM shape as the forward on the mirrored image of dout
  hipLaunchKernelGGL((k_conv_im2col<T, true>), dim3(conv_grid(g.R * g.KKp / 8)), dim3(256), 0, s, dout, g, -1, colT, outT, (long)g.C * g.R);
  hipLaunchKernelGGL((k_conv_pack_w<T>), dim3(conv_grid(g.KKp * g.C)), dim3(256), 0, s, W, g, 1, wT);
  CINEMA_CHECK_HIP(hipGetLastError());
  if (int rc = conv_gemm<T>(wT, g.C, colT, (int)g.R, outT, (int)g.R, g.C, g.S * g.G, g.KKp, s, true)) return rc;
  hipLaunchKernelGGL(k_conv_untranspose, dim3(conv_grid((long)g.S * g.C * g.G)), dim3(256), 0, s, outT, (const float *)nullptr, g, din);
  CINEMA_CHECK_HIP(hipGetLastError());
  return 0;
}
}  // namespace

extern "C" int64_t cinema_conv_mod_work_bytes(int64_t S, int32_t C, int32_t lx, int32_t ly, int32_t lz, int32_t k, int32_t precision) {
  ConvGeom g;
  if (!conv_geom(S, C, lx, ly, lz, k, g) || (precision != CINEMA_PREC_F32 && precision != CINEMA_PREC_BF16)) {
    cinema_set_error("cinema_conv_mod_work_bytes: need S >= 1, C a multiple of 8 (<= 4096), a kernel size <= 15, at most 4096 nodes per subject and 2^24 in all, a valid precision");
    return -1;
  }
  return (int64_t)conv_work(g, precision).bytes;
}

extern "C" int cinema_conv_mod_forward(const float *in, const float *W, const float *b, int64_t S, int32_t C, int32_t lx,
                                       int32_t ly, int32_t lz, int32_t k, int32_t precision, void *work, float *out,
                                       void *stream) {
  ConvGeom g;
  CINEMA_REQUIRE(conv_geom(S, C, lx, ly, lz, k, g), "cinema_conv_mod_forward: bad shape (see cinema_conv_mod_work_bytes)");
  CINEMA_REQUIRE(precision == CINEMA_PREC_F32 || precision == CINEMA_PREC_BF16, "cinema_conv_mod_forward: precision %d", (int)precision);
  CINEMA_REQUIRE(in && W && b && work && out, "cinema_conv_mod_forward: null argument");
  const ConvWork w = conv_work(g, precision);
  return precision == CINEMA_PREC_BF16 ? conv_forward_impl<bf16_t>(g, w, in, W, b, work, out, (hipStream_t)stream)
                                       : conv_forward_impl<float>(g, w, in, W, b, work, out, (hipStream_t)stream);
}

extern "C" int cinema_conv_mod_backward(const float *in, const float *W, const float *dout, int64_t S, int32_t C, int32_t lx,
                                        int32_t ly, int32_t lz, int32_t k, int32_t precision, void *work, float *din,
                                        float *dW, float *db, void *stream) {
  ConvGeom g;
  CINEMA_REQUIRE(conv_geom(S, C, lx, ly, lz, k, g), "cinema_conv_mod_backward: bad shape (see cinema_conv_mod_work_bytes)");
  CINEMA_REQUIRE(precision == CINEMA_PREC_F32 || precision == CINEMA_PREC_BF16, "cinema_conv_mod_backward: precision %d", (int)precision);
  CINEMA_REQUIRE(in && W && dout && work && din && dW && db, "cinema_conv_mod_backward: null argument");
  const ConvWork w = conv_work(g, precision);
  return precision == CINEMA_PREC_BF16 ? conv_backward_impl<bf16_t>(g, w, in, W, dout, work, din, dW, db, (hipStream_t)stream)
                                       : conv_backward_impl<float>(g, w, in, W, dout, work, din, dW, db, (hipStream_t)stream);
}
