// Weight-gradient problem table and the split reduction of k_dw256, shared between siren.hip (which fills the table
// and launches k_dw256) and trilerp.hip (which can run the reduction as a block range beside the K1 coordinate-gradient
// kernel: the two are independent, k_bwd_coord_reduce).  Included INSIDE the anonymous namespace of each translation unit.
#pragma once

// rows / nsplit / wg0 / slab0 are read by the 256 x 256 kernels only (k_dw256, dw256_reduce_body): a problem reduces over
// its own `rows` sample (or node) rows, cut into `nsplit` ranges; its workgroups are [wg0, wg0 + tiles * nsplit) and its
// partial tiles start at slab index slab0 (tile-major, split innermost).  k_dw_grouped uses the common splitk / Npad.
struct DwProb { const void *P; const void *Q; float *out; int ldp, ldq, ldo, M, K, tk; int rows, nsplit, wg0, slab0; };
struct DwArgs {
  DwProb p[3 * CINEMA_MAX_LAYERS + 2];
  int tile_start[3 * CINEMA_MAX_LAYERS + 3];
  int nprob, splitk, Npad;
};
struct Dw256Args { DwArgs g; float *slab; };

// out[h][k] = sum over the splits of one tile; workgroup (256 threads) = 8 rows of one tile, thread = 8 columns;
// vbid = 32 * tile + row block
__device__ __forceinline__ void dw256_reduce_body(const Dw256Args &a, int vbid) {
  const int tg = vbid >> 5, rb = vbid & 31;
  int pi = 0;
  while (pi + 1 < a.g.nprob && tg >= a.g.tile_start[pi + 1]) ++pi;
  const DwProb pr = a.g.p[pi];
  const int t = tg - a.g.tile_start[pi];
  const int h0 = (t / pr.tk) * 256, c0 = (t % pr.tk) * 256;
  const int row = rb * 8 + (threadIdx.x >> 5), col = (threadIdx.x & 31) * 8;
  if (h0 + row >= pr.M || c0 + col >= pr.K) return;
  f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;
  const float *src = a.slab + ((size_t)pr.slab0 + (size_t)t * pr.nsplit) * (256 * 256) + row * 256 + col;
  // all splits are requested before the first one is added (a rolled loop pays one memory round trip per split); the
  // sum keeps the split order
  int sp = 0;
  for (; sp + 4 <= pr.nsplit; sp += 4) {
    f32x4 v[8];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      v[2 * u] = *reinterpret_cast<const f32x4 *>(src + (size_t)(sp + u) * (256 * 256));
      v[2 * u + 1] = *reinterpret_cast<const f32x4 *>(src + (size_t)(sp + u) * (256 * 256) + 4);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) { s0 += v[2 * u]; s1 += v[2 * u + 1]; }
  }
  for (; sp < pr.nsplit; ++sp) {
    s0 += *reinterpret_cast<const f32x4 *>(src + (size_t)sp * (256 * 256));
    s1 += *reinterpret_cast<const f32x4 *>(src + (size_t)sp * (256 * 256) + 4);
  }
  float *out = pr.out + (size_t)(h0 + row) * pr.ldo + c0 + col;
  const int nv = pr.K - c0 - col;
#pragma unroll
  for (int j = 0; j < 8; ++j)
    if (j < nv) out[j] = j < 4 ? s0[j & 3] : s1[j & 3];
}

// dOut [Npad x out_dim] fp32 -> [Npad x 16] T (zero padded) so that the final_linear weight gradient joins the grouped
// weight-gradient launch as two more problems (P = dOut16, Q = x_L and Q = lat); also reduces the bias gradient.
// Any workgroup size that is a multiple of 64 up to 1024; vbid / vgrid as in step_device.h.
struct FinalBwdArgs { const float *dout; int out_dim; float *dbf; int Npad; };
template <typename T>
__device__ __forceinline__ void dout_prep_body(const FinalBwdArgs &a, T *dout16, unsigned vbid, unsigned vgrid) {
  __shared__ float red[16][16];
  const int nthr = blockDim.x;
  float colsum[16];
#pragma unroll
  for (int o = 0; o < 16; ++o) colsum[o] = 0.f;
  for (int n = vbid * nthr + threadIdx.x; n < a.Npad; n += vgrid * nthr) {
    float v[16];
#pragma unroll
    for (int o = 0; o < 16; ++o) {
      v[o] = o < a.out_dim ? a.dout[(size_t)n * a.out_dim + o] : 0.f;
      colsum[o] += v[o];
    }
    store16(dout16 + (size_t)n * 16, v);
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwave = nthr >> 6;
#pragma unroll
  for (int o = 0; o < 16; ++o) {
    float s = colsum[o];
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m, 64);
    if (lane == 0) red[wave][o] = s;
  }
  __syncthreads();
  if ((int)threadIdx.x < a.out_dim && a.dbf != nullptr) {
    float t = 0.f;
    for (int w = 0; w < nwave; ++w) t += red[w][threadIdx.x];
    atomicAdd(a.dbf + threadIdx.x, t);
  }
}
