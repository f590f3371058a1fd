// Device bodies of the node-table preparation of the node-space backward (nodegrad.h): the interpolation weights of a
// batch and the centred node projections.  Shared between nodegrad.hip (its own launch, cinema_nodegrad_prepare) and
// fused.hip, where the weights ride in the step head and the projections as tail workgroups of the forward launch.
// Included INSIDE the anonymous namespace of each translation unit, after k1_device.h and gemm_engine.h.
#pragma once

typedef short s16x4_t __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bf16x4 cvt4(const f32x4 &v) {
  bf16x4 o = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
  return o;
}

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
  bf16_t *wrow; int32_t *axt;    // the same weights row-major [Npad][32]; interpolation axes [Npad][8 words]
};
#define NG_WB 4                   // 16-row blocks per workgroup: one round of dependent loads (idx -> embed) serves 64 rows
template <int NT>        // threads of the workgroup: 512 (k_ng_prepare) or 256 (k_step_head)
__device__ __forceinline__ void ng_weights_body(const NgWeightArgs &a, const long wg, float (*ct)[4]) {
  const int tid = threadIdx.x;
  const long blk0 = wg * NG_WB;
  if (blk0 >= a.nblk) {
    for (long t = tid; t <= a.S; t += NT) a.offs[t] = ng_lower(a.idx, a.N, t, a.S);
    return;
  }
  if (tid < 16 * NG_WB) {
    const long n = blk0 * 16 + tid;
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
  const int i = tid & 15;
#pragma unroll
  for (int q = 0; q < NG_WB; ++q) {
    const long blk = blk0 + q;
    if (blk >= a.nblk) break;
    const int r = 16 * q + i;
#pragma unroll
    for (int sh = 0; sh < 512 / NT; ++sh) {
      const int slot = (tid >> 4) + sh * (NT / 16);
      float w = 0.f;
      if (ct[r][3] != 0.f) w = slot == NG_ONE_SLOT ? 1.f : slot_weight(ct[r][0], ct[r][1], ct[r][2], a.lx, a.ly, a.lz, slot);
      a.wt[(blk * NG_SLOTS + slot) * 16 + i] = (bf16_t)w;
      a.wrow[(blk * 16 + i) * NG_SLOTS + slot] = (bf16_t)w;
    }
  }
  static_assert(NT >= 48 * NG_WB, "axis rows need 48 threads per 16-row block");
  if (tid < 48 * NG_WB) {  // (row, coordinate column): column 0 walks lz, 1 ly, 2 lx
    const int r = tid / 3, ax = tid - 3 * r;
    if (blk0 * 16 + r < (long)a.nblk * 16) {
      const Axis q = axis_nodes(ct[r][ax], ax == 0 ? a.lz : (ax == 1 ? a.ly : a.lx));
      int32_t *o = a.axt + (blk0 * 16 + r) * 8 + 2 * ax;    // {i0 | i1 << 8 | (clip inactive) << 16, w0}; words 6, 7 unused
      o[0] = q.i0 | (q.i1 << 8) | (q.dmul != 0.f ? 1 << 16 : 0);
      o[1] = __float_as_int(q.w0);
      if (ax == 0) { o[6] = 0; o[7] = 0; }
    }
  }
}

// ----------------------------------------------------------------------------------------
// U'[s][slot][f] = sum_c Wl[f][c] * Tc[s][slot][c]   and V' through final_linear[:, H:]   (Tc: noderows_device.h).
// Workgroup (8 waves) = (4 subjects, modulated layer, eighth of the 2 HP features); C[m = feature][n = slot].  The 128
// centred rows are one contiguous piece of Tc: copied into a swizzled LDS image; a wave requests all weight fragments of
// a 16-feature tile before its first product (one L2 round trip per tile) and every fragment serves the eight products of
// its four subjects.  (One subject per workgroup re-read the weights S times, and centring the rows in here -- 24
// workgroups per subject, strided fp32 loads of the API layout -- cost 17 us at cfg-D.)
// ----------------------------------------------------------------------------------------
struct NgProjArgs {
  const bf16_t *Tc; long S; int LP, HP, F, n_mod;
  const bf16_t *packed; long pk_wls[CINEMA_MAX_LAYERS], pk_wlt[CINEMA_MAX_LAYERS]; long pk_wf;
  int layer_of[CINEMA_MAX_LAYERS];
  bf16_t *Up, *Vp;
};
#define NG_PROJ_Q 8                               // feature parts per (subject group, layer)
#define NG_PROJ_SG 4                              // subjects per workgroup: every weight fragment serves 8 products
__device__ __forceinline__ void ng_proj_tile(const NgProjArgs &a, const lds_char *tc, const bf16_t *wrow, int ksteps, int i, int g,
                                             f32x4 (&acc)[NG_PROJ_SG][2]) {
  const int rowb = a.LP * 2, img = NG_SLOTS * rowb;
  for (int k0 = 0; k0 < ksteps; k0 += 8) {
    bf16x8 af[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) af[u] = *reinterpret_cast<const bf16x8 *>(wrow + 32 * (k0 + u < ksteps ? k0 + u : ksteps - 1));
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (k0 + u < ksteps) {
#pragma unroll
        for (int q = 0; q < NG_PROJ_SG; ++q)
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) {
            const int r = 16 * nt + i;
            const bf16x8 bfr = *(const LDS_AS bf16x8 *)(tc + q * img + r * rowb + (((4 * (k0 + u) + g) ^ (r & 7)) << 4));
            acc[q][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[u], bfr, acc[q][nt], 0, 0, 0);
          }
      }
    }
  }
}
__device__ __forceinline__ void ng_project_body(const NgProjArgs &a, int b, lds_char *tc) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 15, g = lane >> 4;
  const int part = b % NG_PROJ_Q; b /= NG_PROJ_Q;
  const int jm = b % a.n_mod;
  const long s0 = (long)(b / a.n_mod) * NG_PROJ_SG;
  const int rowb = a.LP * 2;
  {
    // rows s0 * 32 .. + 127 of Tc (rows behind S * 32 exist and are zero: Rp is a multiple of 128)
    const int cpr = a.LP >> 3, total = NG_PROJ_SG * NG_SLOTS * cpr;
    const bf16_t *src = a.Tc + s0 * NG_SLOTS * a.LP;
    for (int e0 = tid; e0 < total; e0 += 512 * 8) {
      u32x4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int e = e0 + 512 * u;
        v[u] = *reinterpret_cast<const u32x4 *>(src + (long)(e < total ? e : tid) * 8);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int e = e0 + 512 * u;
        if (e < total) {
          const int r = e / cpr, j = e - r * cpr;           // r = 32 q + slot: the images of the four subjects are adjacent
          *(LDS_AS u32x4 *)(tc + r * rowb + ((j ^ (r & 7)) << 4)) = v[u];
        }
      }
    }
  }
  __syncthreads();
  const int ksteps = a.LP >> 5;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  const int layer = a.layer_of[jm];
  // 16-feature tiles of this workgroup: its share of the 2 HP / 16 (scale half, then shift half)
  const int total_mt = a.HP >> 3, per = (total_mt + NG_PROJ_Q - 1) / NG_PROJ_Q;
  const int mt_end = min(total_mt, (part + 1) * per);
  for (int mt = part * per + wave; mt < mt_end; mt += 8) {
    const int f = mt * 16;                                // first feature of the tile in [0, 2 HP)
    const int half = f >= a.HP ? 1 : 0, hf = f - half * a.HP;
    const bf16_t *wrow = a.packed + (half ? a.pk_wlt[layer] : a.pk_wls[layer]) + (long)(hf + i) * a.LP + 8 * g;
    f32x4 acc[NG_PROJ_SG][2];
#pragma unroll
    for (int q = 0; q < NG_PROJ_SG; ++q) { acc[q][0] = zero4; acc[q][1] = zero4; }
    ng_proj_tile(a, tc, wrow, ksteps, i, g, acc);
    // lane (i = slot, g): features f + 4g .. + 3
#pragma unroll
    for (int q = 0; q < NG_PROJ_SG; ++q)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
        if (s0 + q < a.S)
          *reinterpret_cast<bf16x4 *>(a.Up + ((s0 + q) * NG_SLOTS + 16 * nt + i) * a.F + (long)jm * 2 * a.HP + f + 4 * g) = cvt4(acc[q][nt]);
  }
  if (jm == 0 && part == NG_PROJ_Q - 1 && wave == 7) {
    // V'[slot][o] = sum_c Wf[o][H + c] * Tc[slot][c]   (padded [16][HP + LP] copy of final_linear)
    const bf16_t *wrow = a.packed + a.pk_wf + (long)i * (a.HP + a.LP) + a.HP + 8 * g;
    f32x4 acc[NG_PROJ_SG][2];
#pragma unroll
    for (int q = 0; q < NG_PROJ_SG; ++q) { acc[q][0] = zero4; acc[q][1] = zero4; }
    ng_proj_tile(a, tc, wrow, ksteps, i, g, acc);
#pragma unroll
    for (int q = 0; q < NG_PROJ_SG; ++q)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
        if (s0 + q < a.S)
          *reinterpret_cast<bf16x4 *>(a.Vp + ((s0 + q) * NG_SLOTS + 16 * nt + i) * 16 + 4 * g) = cvt4(acc[q][nt]);
  }
}


// ---- host side: the two bodies' arguments from the work buffer's plan ----
inline NgWeightArgs ng_weight_args(const NgPlan &p, void *work, const float *coords, const int32_t *idx, const float *embed,
                                   int tf_dim, long N, int lx, int ly, int lz) {
  char *w = (char *)work;
  return NgWeightArgs{coords, idx, embed, tf_dim, N, p.S, lx, ly, lz, (bf16_t *)(w + p.wt), (int32_t *)(w + p.offs),
                      (int)(p.Npad / 16), (bf16_t *)(w + p.wrow), (int32_t *)(w + p.axt)};
}
inline int ng_weight_grid(const NgPlan &p) { return (int)((p.Npad / 16 + NG_WB - 1) / NG_WB + 1); }
inline NgProjArgs ng_proj_args(const cinema_siren_desc *d, const NgPlan &p, void *work, const void *packed) {
  char *w = (char *)work;
  NgProjArgs a = {};
  a.Tc = (const bf16_t *)(w + p.Tc); a.S = p.S; a.LP = d->LP; a.HP = d->HP; a.F = p.F; a.n_mod = p.n_mod;
  a.packed = (const bf16_t *)packed; a.pk_wf = d->pk_wf;
  for (int j = 0; j < p.n_mod; ++j) {
    a.layer_of[j] = p.layer_of[j];
    a.pk_wls[p.layer_of[j]] = d->pk_wls[p.layer_of[j]];
    a.pk_wlt[p.layer_of[j]] = d->pk_wlt[p.layer_of[j]];
  }
  a.Up = (bf16_t *)(w + p.Up); a.Vp = (bf16_t *)(w + p.Vp);
  return a;
}
inline long ng_proj_grid(const NgPlan &p) { return (long)((p.S + NG_PROJ_SG - 1) / NG_PROJ_SG) * p.n_mod * NG_PROJ_Q; }
inline int ng_proj_smem(const cinema_siren_desc *d) { return NG_PROJ_SG * NG_SLOTS * d->LP * 2; }
