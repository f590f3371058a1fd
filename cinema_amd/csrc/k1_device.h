// Device code of the K1 forward shared by trilerp.hip and fused.hip (the step-head launch): 3x3 helpers, the affine of
// an embedding row, interpolation nodes / weights and the grouped-batch forward body.  Included INSIDE the anonymous
// namespace of each translation unit.
#pragma once

struct M3 {
  float m[3][3];
};
__device__ __forceinline__ M3 m3_identity() {
  M3 r;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) r.m[i][j] = i == j ? 1.f : 0.f;
  return r;
}
__device__ __forceinline__ M3 m3_mul(const M3 &a, const M3 &b) {
  M3 r;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) r.m[i][j] = a.m[i][0] * b.m[0][j] + a.m[i][1] * b.m[1][j] + a.m[i][2] * b.m[2][j];
  return r;
}
__device__ __forceinline__ M3 m3_t(const M3 &a) {
  M3 r;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) r.m[i][j] = a.m[j][i];
  return r;
}
__device__ __forceinline__ float m3_dot(const M3 &a, const M3 &b) {
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) s += a.m[i][j] * b.m[i][j];
  return s;
}
__device__ __forceinline__ M3 rot_z(float c, float s) {
  M3 r = m3_identity();
  r.m[0][0] = c; r.m[0][1] = -s; r.m[1][0] = s; r.m[1][1] = c;
  return r;
}
__device__ __forceinline__ M3 rot_x(float c, float s) {
  M3 r = m3_identity();
  r.m[1][1] = c; r.m[1][2] = -s; r.m[2][1] = s; r.m[2][2] = c;
  return r;
}
// derivative matrices d/da of rot_z / rot_x
__device__ __forceinline__ M3 drot_z(float c, float s) {
  M3 r;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) r.m[i][j] = 0.f;
  r.m[0][0] = -s; r.m[0][1] = -c; r.m[1][0] = c; r.m[1][1] = -s;
  return r;
}
__device__ __forceinline__ M3 drot_x(float c, float s) {
  M3 r;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) r.m[i][j] = 0.f;
  r.m[1][1] = -s; r.m[1][2] = -c; r.m[2][1] = c; r.m[2][2] = -s;
  return r;
}
__device__ __forceinline__ void shear_mats(const float *e, M3 &sx, M3 &sy, M3 &sz) {
  const float hx = e[9], hy = e[10], hz = e[11];
  sx = m3_identity(); sy = m3_identity(); sz = m3_identity();
  sx.m[0][1] = hy; sx.m[0][2] = hz;
  sy.m[1][0] = hx; sy.m[1][2] = hz;
  sz.m[2][0] = hx; sz.m[2][1] = hy;
}

// A = Rz(e0) Rx(e1) Rz(e2) [diag(1+e6..8)] [Sx Sy Sz],  t = e3..5        (utils.py:126-166)
// Not inlined on purpose: how the products of a rotation entry get fused (and hence its last bits) depends on the code
// around an inlined copy, and the kernels that build a subject's affine (k_affine_from_embed, k_trilerp_fwd_sorted) must
// agree to the bit.  One compiled body, called from both.
struct Embed12 { float v[12]; };
__device__ __noinline__ M3 affine_from_embed(Embed12 ev, int dim) {
  const float *e = ev.v;
  float s1, c1, s2, c2, s3, c3;
  sincosf(e[0], &s1, &c1);
  sincosf(e[1], &s2, &c2);
  sincosf(e[2], &s3, &c3);
  M3 A = m3_mul(m3_mul(rot_z(c1, s1), rot_x(c2, s2)), rot_z(c3, s3));
  if (dim >= 9) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) A.m[i][j] *= 1.f + e[6 + j];
  }
  if (dim == 12) {
    M3 sx, sy, sz;
    shear_mats(e, sx, sy, sz);
    A = m3_mul(m3_mul(m3_mul(A, sx), sy), sz);
  }
  return A;
}

// one row of c' = A c + t with the roundings spelled out (left to the compiler, one kernel fuses the products and
// another one vectorises them into unfused packed multiplies: up to 64 ulp apart where the terms cancel)
__device__ __forceinline__ float affine_row(float a0, float a1, float a2, float t, float cx, float cy, float cz) {
  return __builtin_fmaf(a2, cz, __builtin_fmaf(a1, cy, a0 * cx)) + t;
}

// d embed from d coords_t:  coords_t = A c + t
// gradient of the loss w.r.t. one embedding row e from G = dL/dA (3x3) and d = dL/dt: every component is LINEAR in
// (G, d), so a subject's table row needs only the sums of d c^T and d over its samples (k_tf_table_bwd)
__device__ __forceinline__ void embed_grad(const float (&e)[12], int dim, const M3 &G, const float (&d)[3], float (&g)[12]) {
  float s1, c1, s2, c2, s3, c3;
  sincosf(e[0], &s1, &c1);
  sincosf(e[1], &s2, &c2);
  sincosf(e[2], &s3, &c3);
  const M3 Rz1 = rot_z(c1, s1), Rx2 = rot_x(c2, s2), Rz3 = rot_z(c3, s3);
  const M3 R = m3_mul(m3_mul(Rz1, Rx2), Rz3);
  M3 D = m3_identity(), Sh = m3_identity(), sx = m3_identity(), sy = m3_identity(), sz = m3_identity();
  if (dim >= 9) { D.m[0][0] = 1.f + e[6]; D.m[1][1] = 1.f + e[7]; D.m[2][2] = 1.f + e[8]; }
  if (dim == 12) { shear_mats(e, sx, sy, sz); Sh = m3_mul(m3_mul(sx, sy), sz); }
  // rotation angles: <G (D Sh)^T, dR/dtheta>
  const M3 GR = m3_mul(G, m3_t(m3_mul(D, Sh)));
  g[0] = m3_dot(GR, m3_mul(m3_mul(drot_z(c1, s1), Rx2), Rz3));
  g[1] = m3_dot(GR, m3_mul(m3_mul(Rz1, drot_x(c2, s2)), Rz3));
  g[2] = m3_dot(GR, m3_mul(m3_mul(Rz1, Rx2), drot_z(c3, s3)));
  g[3] = d[0]; g[4] = d[1]; g[5] = d[2];
  if (dim >= 9) {
    const M3 GD = m3_mul(m3_mul(m3_t(R), G), m3_t(Sh));
    g[6] = GD.m[0][0]; g[7] = GD.m[1][1]; g[8] = GD.m[2][2];
  }
  if (dim == 12) {
    const M3 GS = m3_mul(m3_t(m3_mul(R, D)), G);                 // dL/dSh
    const M3 gsx = m3_mul(GS, m3_t(m3_mul(sy, sz)));
    const M3 gsy = m3_mul(m3_mul(m3_t(sx), GS), m3_t(sz));
    const M3 gsz = m3_mul(m3_t(m3_mul(sx, sy)), GS);
    g[9] = gsy.m[1][0] + gsz.m[2][0];
    g[10] = gsx.m[0][1] + gsz.m[2][1];
    g[11] = gsx.m[0][2] + gsy.m[1][2];
  }
}

struct Axis {
  int i0, i1;
  float w0, w1;
  float dmul;   // d u / d x if the clip is inactive, else 0
};
__device__ __forceinline__ Axis axis_nodes(float x, int size) {
  Axis a;
  float u = (x + 1.f) * 0.5f * (float)(size - 1);
  const float hi = (float)(size - 1);
  a.dmul = (u > 0.f && u < hi) ? 0.5f * hi : 0.f;
  u = fminf(fmaxf(u, 0.f), hi);
  const float f = floorf(u);
  a.i0 = (int)f;
  a.w1 = u - f;
  a.w0 = 1.f - a.w1;
  a.i1 = a.i0 + 1;
  if (a.i1 > size - 1) { a.i1 = a.i0; a.w1 = 0.f; }
  return a;
}

struct TriArgs {
  const float *coords; const int32_t *idx; const float *affine; int affine_per_sample;
  const float *shadow; const float *cond;
  long N, Npad, S; int C, lx, ly, lz, cond_dims, LP;
  float *coords_t; void *lat_out;
};

__device__ __forceinline__ void corner_setup(const Axis &ax, const Axis &ay, const Axis &az, int ly, int lz,
                                             int (&node)[8], float (&w)[8]) {
  int q = 0;
#pragma unroll
  for (int bx = 0; bx < 2; ++bx)
#pragma unroll
    for (int by = 0; by < 2; ++by)
#pragma unroll
      for (int bz = 0; bz < 2; ++bz) {
        const int ix = bx ? ax.i1 : ax.i0, iy = by ? ay.i1 : ay.i0, iz = bz ? az.i1 : az.i0;
        node[q] = (ix * ly + iy) * lz + iz;
        w[q] = (bx ? ax.w1 : ax.w0) * (by ? ay.w1 : ay.w0) * (bz ? az.w1 : az.w0);
        ++q;
      }
}

// ----------------------------------------------------------------------------------------
// Forward for a batch grouped by subject (any order is computed correctly; a grouped one is fast), table-mode embedding,
// compile-time grid, C <= 256: a workgroup owns 64 consecutive samples, which in a grouped batch belong to one subject
// (else the chunk is walked segment by segment).  The subject's table goes from the API layout (C, G) straight into LDS
// as [node][channel] (no node-major shadow, no separate affine launch on the critical path: a sample's thread builds
// its subject's affine itself), the 8 corner rows of a sample are LDS reads, a lane owns 4 consecutive channels.
// Same arithmetic in the same order as k_affine_from_embed + k_trilerp_fwd: bit-identical outputs.
// ----------------------------------------------------------------------------------------
struct TriFwdSortedArgs {
  const float *coords; const int32_t *idx; const float *embed; int tf_dim; const float *latents; const float *cond;
  long N, Npad, S; int C, cond_dims, LP;
  float *coords_t; void *lat_out;
  int cond_by_subject;   // cond is a per-subject table [S x cond_dims] (row = idx[n]) instead of per-sample rows [N x cond_dims]
};
#define TFS_CHUNK 64
#define TFS_THREADS 256
#define TFS_CMAX 256

// LDS of the body (the caller's __shared__ pool, 16-byte aligned): inside k_step_head every block range carves its arrays
// from ONE pool -- as static arrays of each body they add up, and the sum decides how many workgroups a CU holds
#define TFS_LDS_BYTES(G) ((G) * TFS_CMAX * 4 + TFS_CHUNK * 8 * 4 * 2 + TFS_CHUNK * 4)
template <typename T, int LX, int LY, int LZ>
__device__ __forceinline__ void trilerp_fwd_sorted_body(const TriFwdSortedArgs &a, const int vbid, char *pool) {
  constexpr int G = LX * LY * LZ;
  float *tab = reinterpret_cast<float *>(pool);           // [node][channel]
  float *cw = tab + G * TFS_CMAX;                         // corner weights [TFS_CHUNK * 8]
  int *cn = reinterpret_cast<int *>(cw + TFS_CHUNK * 8);  // corner nodes [TFS_CHUNK * 8]
  int *sid = cn + TFS_CHUNK * 8;                          // [TFS_CHUNK]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const long base = (long)vbid * TFS_CHUNK;
  const int C = a.C;
  T *out = (T *)a.lat_out;
  const int nc = (int)(a.N - base < TFS_CHUNK ? (a.N - base > 0 ? a.N - base : 0) : TFS_CHUNK);   // real samples of the chunk
  auto clamp_s = [&](long sj) { return (int)(sj < 0 ? 0 : (sj >= a.S ? a.S - 1 : sj)); };
  auto table_request = [&](int s, float (&tv)[G]) {      // thread = channel: G strided reads of the subject's 27 KB block
    const float *tb = a.latents + ((long)s * C + (tid < C ? tid : 0)) * G;
#pragma unroll
    for (int k = 0; k < G; ++k) tv[k] = tb[k];
  };
  auto table_commit = [&](const float (&tv)[G]) {
#pragma unroll
    for (int k = 0; k < G; ++k) tab[k * TFS_CMAX + tid] = tid < C ? tv[k] : 0.f;
  };
  float tv[G];
  if (nc > 0) table_request(clamp_s(a.idx[base]), tv);
  if (tid < TFS_CHUNK) {
    const int t = tid;
    const long n = base + t;
    if (t < nc) {
      const float cx = a.coords[n * 3], cy = a.coords[n * 3 + 1], cz = a.coords[n * 3 + 2];
      const int s = clamp_s(a.idx[n]);
      sid[t] = s;
      float t0 = cx, t1 = cy, t2 = cz;
      if (a.embed != nullptr) {
        Embed12 ev;
        float (&e)[12] = ev.v;
#pragma unroll
        for (int k = 0; k < 12; ++k) e[k] = k < a.tf_dim ? a.embed[(long)s * a.tf_dim + k] : 0.f;
        const M3 A = affine_from_embed(ev, a.tf_dim);
        t0 = affine_row(A.m[0][0], A.m[0][1], A.m[0][2], e[3], cx, cy, cz);
        t1 = affine_row(A.m[1][0], A.m[1][1], A.m[1][2], e[4], cx, cy, cz);
        t2 = affine_row(A.m[2][0], A.m[2][1], A.m[2][2], e[5], cx, cy, cz);
      }
      a.coords_t[n * 3] = t0; a.coords_t[n * 3 + 1] = t1; a.coords_t[n * 3 + 2] = t2;
      const Axis az = axis_nodes(t0, LZ), ay = axis_nodes(t1, LY), ax = axis_nodes(t2, LX);
      int node[8]; float w[8];
      corner_setup(ax, ay, az, LY, LZ, node, w);
#pragma unroll
      for (int q = 0; q < 8; ++q) { cn[t * 8 + q] = node[q]; cw[t * 8 + q] = w[q]; }
    } else {
      sid[t] = -1;
      if (n < a.Npad) { a.coords_t[n * 3] = 0.f; a.coords_t[n * 3 + 1] = 0.f; a.coords_t[n * 3 + 2] = 0.f; }
    }
  }
  // pad rows of the chunk (n >= N): zero
  for (int t = nc + wave; t < TFS_CHUNK; t += TFS_THREADS / 64) {
    const long n = base + t;
    if (n < a.Npad)
      for (int j = lane; j < a.LP; j += 64) out[n * a.LP + j] = (T)0.f;
  }
  if (nc == 0) return;
  table_commit(tv);
  __syncthreads();
  int seg_lo = 0;
  while (seg_lo < nc) {                                 // one turn per subject present in the chunk (workgroup-uniform)
    const int s = sid[seg_lo];
    // the run of samples of s that starts at seg_lo (an unsorted batch just makes short runs)
    const unsigned long long same = __ballot(lane >= seg_lo && lane < nc && sid[lane] == s) >> seg_lo;
    const int run = ~same ? __builtin_ctzll(~same) : 64;
    const int seg_hi = seg_lo + run;
    if (seg_lo > 0) {
      __syncthreads();                                  // every wave is done with the previous table
      table_request(s, tv);
      table_commit(tv);
      __syncthreads();
    }
    const int ulo = seg_lo > wave * 16 ? seg_lo : wave * 16;
    const int uhi = seg_hi < wave * 16 + 16 ? seg_hi : wave * 16 + 16;
#pragma unroll 2
    for (int t = ulo; t < uhi; ++t) {
      const long n = base + t;
      T *orow = out + n * a.LP;
      if (4 * lane < C) {
        const int4 nd0 = *reinterpret_cast<const int4 *>(cn + t * 8), nd1 = *reinterpret_cast<const int4 *>(cn + t * 8 + 4);
        const float4 w0 = *reinterpret_cast<const float4 *>(cw + t * 8), w1 = *reinterpret_cast<const float4 *>(cw + t * 8 + 4);
        const int nd[8] = {nd0.x, nd0.y, nd0.z, nd0.w, nd1.x, nd1.y, nd1.z, nd1.w};
        const float w[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
        float v[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const float4 r = *reinterpret_cast<const float4 *>(tab + nd[q] * TFS_CMAX + 4 * lane);
          v[0] = __builtin_fmaf(w[q], r.x, v[0]); v[1] = __builtin_fmaf(w[q], r.y, v[1]);
          v[2] = __builtin_fmaf(w[q], r.z, v[2]); v[3] = __builtin_fmaf(w[q], r.w, v[3]);
        }
        if constexpr (sizeof(T) == 2) {
          typedef __attribute__((ext_vector_type(4))) T tx4;
          tx4 o = {(T)v[0], (T)v[1], (T)v[2], (T)v[3]};
          *reinterpret_cast<tx4 *>(orow + 4 * lane) = o;
        } else {
          float4 o = {v[0], v[1], v[2], v[3]};
          *reinterpret_cast<float4 *>(orow + 4 * lane) = o;
        }
      }
      for (int j = lane; j < a.cond_dims; j += 64) orow[C + j] = (T)a.cond[(a.cond_by_subject ? (long)s : n) * a.cond_dims + j];
      for (int j = C + a.cond_dims + lane; j < a.LP; j += 64) orow[j] = (T)0.f;
    }
    seg_lo = seg_hi;
  }
}

template <typename T, int LX, int LY, int LZ>
__global__ __launch_bounds__(TFS_THREADS) void k_trilerp_fwd_sorted(TriFwdSortedArgs a) {
  __shared__ __attribute__((aligned(16))) char pool[TFS_LDS_BYTES(LX * LY * LZ)];
  trilerp_fwd_sorted_body<T, LX, LY, LZ>(a, (int)blockIdx.x, pool);
}

