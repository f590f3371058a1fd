// Weight-gradient problem table and the split reduction of k_dw256, shared between siren.hip (which fills the table
// and launches k_dw256) and trilerp.hip (which can run the reduction as a block range beside the K1 coordinate-gradient
// kernel: the two are independent, k_bwd_coord_reduce).  Included INSIDE the anonymous namespace of each translation unit.
#pragma once

struct DwProb { const void *P; const void *Q; float *out; int ldp, ldq, ldo, M, K, tk; };
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
  const float *src = a.slab + (size_t)tg * a.g.splitk * (256 * 256) + row * 256 + col;
  for (int sp = 0; sp < a.g.splitk; ++sp) {
    s0 += *reinterpret_cast<const f32x4 *>(src + (size_t)sp * (256 * 256));
    s1 += *reinterpret_cast<const f32x4 *>(src + (size_t)sp * (256 * 256) + 4);
  }
  float *out = pr.out + (size_t)(h0 + row) * pr.ldo + c0 + col;
  const int nv = pr.K - c0 - col;
#pragma unroll
  for (int j = 0; j < 8; ++j)
    if (j < nv) out[j] = j < 4 ? s0[j & 3] : s1[j & 3];
}
