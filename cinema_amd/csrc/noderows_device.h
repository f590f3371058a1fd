// Node rows of the node-space backward (nodegrad.h): T [S][32][LP] bf16 and their centred twins Tc.  Shared between the
// one-launch step head (fused.hip) and nodegrad.hip's own kernel; included INSIDE the anonymous namespace of each.
//   T  row (s, slot < G) = [table[s][:, slot], cond[s], 0...]          (rows of the other slots are zero)
//   Tc row (s, slot < G) = [table[s][:, slot] - mean_slot table[s][:, .], 0...]
// The mean over the G nodes is taken out in fp32 BEFORE the rows are rounded to bf16: d coords only sees differences
// between the nodes of a cell (the derivative weights of a sample sum to zero), so the rounding error must scale with
// the spread of a subject's nodes, not with their common part.  Conditions are per subject: their centred columns are zero.
// Workgroup (256 threads) = (subject, 32 columns): the 32 x G block of the table is contiguous -- read coalesced into
// LDS, column means by 32 threads, then thread = (slot, 8 columns) writes 16 B of each row.
#pragma once

struct NodeRowsCArgs { const float *latents; const float *cond; long S; int C, G, cond_dims, LP; bf16_t *T, *Tc; };
#define NODE_ROWS_LDS_BYTES ((32 * 32 + 32) * 4)
__device__ __forceinline__ void node_rows_centred_body(const NodeRowsCArgs &a, unsigned vbid, char *pool) {
  float *stage = reinterpret_cast<float *>(pool);   // [column][G], G <= 31  (the caller's __shared__ pool)
  float *mean = stage + 32 * 32;                    // [32]
  const int nchunk = a.LP >> 5, tid = threadIdx.x;
  const long s = vbid / nchunk;
  const int c0 = (int)(vbid % nchunk) * 32;
  if (s >= a.S) return;
  const int nch = a.C - c0 < 0 ? 0 : (a.C - c0 > 32 ? 32 : a.C - c0);      // table columns in this chunk
  const float *src = a.latents + (s * a.C + c0) * a.G;
  for (int e = tid; e < nch * a.G; e += 256) stage[e] = src[e];
  __syncthreads();
  if (tid < 32) {
    float m = 0.f;
    if (tid < nch) {
      for (int k = 0; k < a.G; ++k) m += stage[tid * a.G + k];
      m /= (float)a.G;
    }
    mean[tid] = m;
  }
  __syncthreads();
  if (tid < 128) {
    const int slot = tid >> 2, ch = tid & 3;
    bf16x8 o, oc;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int cl = ch * 8 + e, c = c0 + cl;
      float v = 0.f, vc = 0.f;
      if (slot < a.G) {
        if (cl < nch) { v = stage[cl * a.G + slot]; vc = v - mean[cl]; }
        else if (c >= a.C && c < a.C + a.cond_dims) v = a.cond[s * a.cond_dims + (c - a.C)];
      }
      o[e] = (bf16_t)v; oc[e] = (bf16_t)vc;
    }
    const long at = (s * 32 + slot) * a.LP + c0 + ch * 8;
    *reinterpret_cast<bf16x8 *>(a.T + at) = o;
    *reinterpret_cast<bf16x8 *>(a.Tc + at) = oc;
  }
}
