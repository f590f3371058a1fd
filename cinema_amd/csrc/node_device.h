// Device bodies of the node-space path shared between node.hip (stand-alone kernels) and the one-launch step head of
// fused.hip.  Included INSIDE the anonymous namespace of a translation unit, after nodes.h and k1_device.h.
// (vbid, vgrid) = index / count of the 256-thread workgroups doing the job.
#pragma once

// Per sample: registered coordinate (same arithmetic and roundings as the K1 kernels: affine_from_embed + affine_row)
// and the row of 32 fp16 interpolation weights in fragment position order (nodes.h).  Rows >= N are zero rows.
// LX = 0: run-time grid.
struct NodeWeightArgs {
  const float *coords; const int32_t *idx; const float *embed; int tf_dim; long N, rows, S; int lx, ly, lz;
  float *coords_t; f16_t *wrows;
};
template <int LX, int LY, int LZ>
__device__ __forceinline__ void node_weights_body(const NodeWeightArgs &a, unsigned vbid, unsigned vgrid) {
  const int lx = LX ? LX : a.lx, ly = LX ? LY : a.ly, lz = LX ? LZ : a.lz;
  for (long n = (long)vbid * 256 + threadIdx.x; n < a.rows; n += (long)vgrid * 256) {
    float wv[NODE_SLOTS];
#pragma unroll
    for (int p = 0; p < NODE_SLOTS; ++p) wv[p] = 0.f;
    float t0 = 0.f, t1 = 0.f, t2 = 0.f;
    if (n < a.N) {
      const float cx = a.coords[n * 3], cy = a.coords[n * 3 + 1], cz = a.coords[n * 3 + 2];
      long s = a.idx[n];
      s = s < 0 ? 0 : (s >= a.S ? a.S - 1 : s);
      t0 = cx; t1 = cy; t2 = cz;
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
      // coordinate column 0 walks the LAST grid axis (lz), column 1 ly, column 2 lx
      const Axis az = axis_nodes(t0, lz), ay = axis_nodes(t1, ly), ax = axis_nodes(t2, lx);
      int node[8]; float w[8];
      corner_setup(ax, ay, az, ly, lz, node, w);
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int pos = node_pos_of_slot(node[q]);
#pragma unroll
        for (int p = 0; p < NODE_SLOTS; ++p) wv[p] += p == pos ? w[q] : 0.f;    // a clamped axis repeats a node with weight 0
      }
      wv[NODE_ONE_SLOT] = 1.f;                                                  // position of slot 31 is 31
    }
    a.coords_t[n * 3] = t0; a.coords_t[n * 3 + 1] = t1; a.coords_t[n * 3 + 2] = t2;
    f16x8 *dst = reinterpret_cast<f16x8 *>(a.wrows + n * NODE_SLOTS);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      f16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (f16_t)wv[8 * c + e];
      dst[c] = o;
    }
  }
}

// Node rows T [S][32][LP] bf16 from the latent table in its API layout (S, C, G) and the per-subject conditions:
// row (s, slot < G) = [table[s][:, slot], cond[s], 0...], rows of the other slots are zero.  Thread = 8 columns.
struct NodeRowArgs { const float *latents; const float *cond; long S; int C, G, cond_dims, LP; bf16_t *T; };
__device__ __forceinline__ void node_rows_body(const NodeRowArgs &a, unsigned vbid, unsigned vgrid) {
  const int cpr = a.LP >> 3;
  const long total = a.S * NODE_SLOTS * cpr;
  for (long q = (long)vbid * 256 + threadIdx.x; q < total; q += (long)vgrid * 256) {
    const int ch = (int)(q % cpr);
    const long r = q / cpr;
    const int slot = (int)(r % NODE_SLOTS);
    const long s = r / NODE_SLOTS;
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = ch * 8 + e;
      float v = 0.f;
      if (slot < a.G) {
        if (c < a.C) v = a.latents[(s * a.C + c) * a.G + slot];
        else if (c < a.C + a.cond_dims) v = a.cond[s * a.cond_dims + (c - a.C)];
      }
      o[e] = (bf16_t)v;
    }
    *reinterpret_cast<bf16x8 *>(a.T + r * a.LP + ch * 8) = o;
  }
}
