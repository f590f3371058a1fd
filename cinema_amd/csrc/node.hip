// Node-space side of the "nodes" path (see nodes.h): per-sample interpolation weights, per-subject node rows and their
// projections through linear_lats / final_linear.
// Reference arithmetic: models/inr_decoder.py:64-85 (transform, [idcs_df] gather, grid_sample), models/siren.py:32
// (linear_lats) and :61 (final_linear on cat[x, latent]).
#include "nodes.h"
#include "gemm_engine.h"
#include <algorithm>

namespace {

#include "k1_device.h"
#include "node_device.h"

__global__ __launch_bounds__(256) void k_node_weights(NodeWeightArgs a) { node_weights_body<3, 3, 3>(a, blockIdx.x, gridDim.x); }
__global__ __launch_bounds__(256) void k_node_weights_g(NodeWeightArgs a) { node_weights_body<0, 0, 0>(a, blockIdx.x, gridDim.x); }
__global__ __launch_bounds__(256) void k_node_rows(NodeRowArgs a) { node_rows_body(a, blockIdx.x, gridDim.x); }

// ----------------------------------------------------------------------------------------
// U = Wl * T^T per (modulated layer, 64-feature slab, subject), written as fp16 MFMA A fragments (nodes.h).
// Workgroup = 4 waves = one (layer, slab) and NODE_PROJ_SUBJ subjects; the two 64-row halves (scale, shift) of the
// packed bf16 linear_lats copy are staged once in LDS (XOR-swizzled 16-B chunks), the node rows come from L2.
// C[m = slot][n = feature]: a lane ends up with slots 4g..4g+3 of both 16-slot tiles = its 8 positions of the fragment.
// ----------------------------------------------------------------------------------------
#define NODE_PROJ_SUBJ 8
struct NodeProjArgs {
  const bf16_t *packed; long pk_wls[CINEMA_MAX_LAYERS], pk_wlt[CINEMA_MAX_LAYERS]; long pk_wf;
  const float *params; long off_bl[CINEMA_MAX_LAYERS];
  int layer_of[CINEMA_MAX_LAYERS];      // modulated ordinal j -> layer
  int n_mod, nslab, H, HP, LP, out_dim;
  const bf16_t *T; long S;
  char *U, *V;
  int g_u;                              // workgroups [0, g_u) project U, the rest V
};

__global__ __launch_bounds__(256) void k_node_project(NodeProjArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  lds_char *ws = (lds_char *)smem;                         // [128 rows][LP] bf16, swizzled
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 15, g = lane >> 4;
  const int rowb = a.LP * 2, cpr = a.LP >> 3, ksteps = a.LP >> 5;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  if ((int)blockIdx.x >= a.g_u) {
    // V: one wave per subject; B rows = final_linear rows (lat part of the padded [16][HP + LP] copy)
    const long s = ((long)blockIdx.x - a.g_u) * 4 + wave;
    if (s >= a.S) return;
    f32x4 acc[2] = {zero4, zero4};
    const bf16_t *wrow = a.packed + a.pk_wf + (long)i * (a.HP + a.LP) + a.HP;
    for (int kk = 0; kk < ksteps; ++kk) {
      const bf16x8 b = *reinterpret_cast<const bf16x8 *>(wrow + 32 * kk + 8 * g);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const bf16x8 af = *reinterpret_cast<const bf16x8 *>(a.T + ((s * NODE_SLOTS + 16 * mt + i) * a.LP + 32 * kk + 8 * g));
        acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, b, acc[mt], 0, 0, 0);
      }
    }
    f16x8 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) { o[e] = (f16_t)acc[0][e]; o[4 + e] = (f16_t)acc[1][e]; }
    *reinterpret_cast<f16x8 *>(a.V + s * 1024 + lane * 16) = o;
    return;
  }
  const int ngrp = (int)((a.S + NODE_PROJ_SUBJ - 1) / NODE_PROJ_SUBJ);
  const int task = blockIdx.x / ngrp, grp = blockIdx.x - task * ngrp;
  const int j = task / a.nslab, slab = task - j * a.nslab;
  const int layer = a.layer_of[j];
  // stage the two 64-row halves of this slab
  for (int q = tid; q < 128 * cpr; q += 256) {
    const int row = q / cpr, c = q - row * cpr;
    const int h = row >> 6, fl = row & 63;
    const bf16_t *src = a.packed + (h ? a.pk_wlt[layer] : a.pk_wls[layer]) + (long)(slab * 64 + fl) * a.LP + c * 8;
    const int key = (((fl >> 3) & 1) << 2) | (fl & 3);      // 3 bits: stays inside an aligned group of 8 chunks
    *(LDS_AS u32x4 *)(ws + row * rowb + ((c ^ key) << 4)) = *reinterpret_cast<const u32x4 *>(src);
  }
  __syncthreads();
  for (int q = wave; q < NODE_PROJ_SUBJ; q += 4) {
    const long s = (long)grp * NODE_PROJ_SUBJ + q;
    if (s >= a.S) break;
    f32x4 acc[2][4][2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int ft = 0; ft < 4; ++ft) { acc[h][ft][0] = zero4; acc[h][ft][1] = zero4; }
    const bf16_t *trow = a.T + (s * NODE_SLOTS + i) * a.LP + 8 * g;
    for (int kk = 0; kk < ksteps; ++kk) {
      bf16x8 af[2];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) af[mt] = *reinterpret_cast<const bf16x8 *>(trow + (long)16 * mt * a.LP + 32 * kk);
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int ft = 0; ft < 4; ++ft) {
          const int fl = 32 * (ft >> 1) + 8 * (i >> 2) + 4 * (ft & 1) + (i & 3);      // feature map of nodes.h
          // key of row fl: ((fl >> 3) & 1) << 2 | (fl & 3) = i & 7
          const bf16x8 b = *(const LDS_AS bf16x8 *)(ws + (h * 64 + fl) * rowb + (((4 * kk + g) ^ (i & 7)) << 4));
#pragma unroll
          for (int mt = 0; mt < 2; ++mt) acc[h][ft][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[mt], b, acc[h][ft][mt], 0, 0, 0);
        }
    }
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int ft = 0; ft < 4; ++ft) {
        f16x8 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) { o[e] = (f16_t)acc[h][ft][0][e]; o[4 + e] = (f16_t)acc[h][ft][1][e]; }
        if (g == 3) {                                  // position 31 = the constant-1 slot: the linear_lats bias
          const int f = slab * 64 + 32 * (ft >> 1) + 8 * (i >> 2) + 4 * (ft & 1) + (i & 3);
          o[7] = (f16_t)(f < a.H ? a.params[a.off_bl[layer] + (long)h * a.H + f] : 0.f);
        }
        *reinterpret_cast<f16x8 *>(a.U + (((long)(j * a.nslab + slab) * a.S + s) * 2 + h) * 4096 + ft * 1024 + lane * 16) = o;
      }
  }
}

// Sample tiles of a batch grouped by subject (idx non-decreasing; the caller's precondition): every subject's row range
// is cut into tiles of at most 16 rows, tiles[t] = {subject, first row, rows, 0}, in row order; *n_tiles = their number
// (<= rows / 16 + subjects present).  One workgroup.  idx < 0 counts as subject 0, idx >= S as S - 1 (as everywhere).
// Rows [N, rows) -- the padding of the row buffers -- are appended to subject S - 1 so that they hold finite values too.
#define NODE_TILES_MAX_S 8192
__global__ __launch_bounds__(1024) void k_node_tiles(const int32_t *idx, long N, long rows, long S, int *out) {
  __shared__ int off[NODE_TILES_MAX_S + 1];
  __shared__ int tb[NODE_TILES_MAX_S + 1];
  __shared__ int part[1024];
  __shared__ int carry;
  const int tid = threadIdx.x;
  int4 *tiles = reinterpret_cast<int4 *>(out + 4);
  for (long s = tid; s <= S; s += 1024) {
    long lo = 0, hi = N;
    if (s == 0) hi = 0;
    else if (s == S) lo = hi = rows;       // the pad rows [N, rows) (zero weights, zero gradients) ride with the last subject
    while (lo < hi) {
      const long mid = (lo + hi) >> 1;
      if (idx[mid] < s) lo = mid + 1; else hi = mid;
    }
    off[s] = (int)lo;
  }
  if (tid == 0) carry = 0;
  __syncthreads();
  for (long b = 0; b < S; b += 1024) {
    const long s = b + tid;
    const int v = s < S ? (off[s + 1] - off[s] + 15) >> 4 : 0;
    part[tid] = v;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
      const int t = tid >= d ? part[tid - d] : 0;
      __syncthreads();
      part[tid] += t;
      __syncthreads();
    }
    if (s < S) tb[s] = carry + part[tid] - v;
    __syncthreads();
    if (tid == 0) carry += part[1023];
    __syncthreads();
  }
  const int T = carry;
  if (tid == 0) { tb[S] = T; out[0] = T; out[1] = out[2] = out[3] = 0; }
  __syncthreads();
  for (int t = tid; t < T; t += 1024) {
    long lo = 0, hi = S - 1;                   // the subject with tb[s] <= t < tb[s + 1]: the last one with tb[s] <= t
    while (lo < hi) {
      const long mid = (lo + hi + 1) >> 1;
      if (tb[mid] <= t) lo = mid; else hi = mid - 1;
    }
    const long s = lo;
    const int k = t - tb[s];
    const int cnt = off[s + 1] - off[s];
    const int4 e = {(int)s, off[s] + 16 * k, cnt - 16 * k < 16 ? cnt - 16 * k : 16, 0};
    tiles[t] = e;
  }
}

}  // namespace

bool nodes_eligible(const cinema_siren_desc *d, int lx, int ly, int lz) {
  return d->precision == CINEMA_PREC_BF16 && d->pk_fs >= 0 && lx >= 1 && ly >= 1 && lz >= 1 && lx * ly * lz < NODE_ONE_SLOT;
}

extern "C" int cinema_nodes_sizes(const cinema_siren_desc *d, int64_t S, int64_t *t_elems, int64_t *u_bytes, int64_t *v_bytes) {
  CINEMA_REQUIRE(d && S >= 1, "cinema_nodes_sizes: bad argument");
  int n_mod = 0;
  for (int i = 0; i < d->n_layers; ++i) n_mod += (d->modulated_mask >> i) & 1u;
  if (t_elems) *t_elems = S * NODE_SLOTS * d->LP;
  if (u_bytes) *u_bytes = node_u_bytes(n_mod, d->HP / 64, S);
  if (v_bytes) *v_bytes = node_v_bytes(S);
  return 0;
}

extern "C" int cinema_node_weights(const float *coords, const int32_t *idx, const float *embed, int32_t tf_dim, int64_t N,
                                   int64_t rows, int64_t S, int32_t lx, int32_t ly, int32_t lz, float *coords_t, void *wrows,
                                   void *stream) {
  CINEMA_REQUIRE(coords && idx && coords_t && wrows, "cinema_node_weights: null argument");
  CINEMA_REQUIRE(N >= 1 && rows >= N && S >= 1, "cinema_node_weights: bad sizes N=%ld rows=%ld S=%ld", (long)N, (long)rows, (long)S);
  CINEMA_REQUIRE(lx >= 1 && ly >= 1 && lz >= 1 && lx * ly * lz < NODE_ONE_SLOT, "cinema_node_weights: grid %d x %d x %d has more than %d nodes",
                 lx, ly, lz, NODE_ONE_SLOT - 1);
  CINEMA_REQUIRE(embed == nullptr || tf_dim == 6 || tf_dim == 9 || tf_dim == 12, "tf embedding width %d unsupported (6, 9 or 12)", tf_dim);
  NodeWeightArgs a = {coords, idx, embed, tf_dim, (long)N, (long)rows, (long)S, lx, ly, lz, coords_t, (f16_t *)wrows};
  const unsigned grid = (unsigned)((rows + 255) / 256);
  if (lx == 3 && ly == 3 && lz == 3) hipLaunchKernelGGL(k_node_weights, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(k_node_weights_g, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
  CINEMA_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int cinema_node_tiles(const int32_t *idx, int64_t N, int64_t rows, int64_t S, void *tiles, void *stream) {
  CINEMA_REQUIRE(idx && tiles && N >= 1 && rows >= N && S >= 1, "cinema_node_tiles: bad argument");
  CINEMA_REQUIRE(S <= NODE_TILES_MAX_S, "cinema_node_tiles: %ld subjects exceed the planner's table (%d)", (long)S, NODE_TILES_MAX_S);
  hipLaunchKernelGGL(k_node_tiles, dim3(1), dim3(1024), 0, (hipStream_t)stream, idx, (long)N, (long)rows, (long)S, (int *)tiles);
  CINEMA_CHECK_HIP(hipGetLastError());
  return 0;
}

int node_project_args(const cinema_siren_desc *d, const float *params, const void *packed, const void *T, int64_t S, void *U,
                      void *V, NodeProjArgs &a, int &grid, int &smem) {
  a = NodeProjArgs{};
  a.packed = (const bf16_t *)packed;
  int nm = 0;
  for (int i = 0; i < d->n_layers; ++i)
    if ((d->modulated_mask >> i) & 1u) {
      a.pk_wls[i] = d->pk_wls[i]; a.pk_wlt[i] = d->pk_wlt[i]; a.off_bl[i] = d->off_bl[i];
      a.layer_of[nm++] = i;
    }
  a.pk_wf = d->pk_wf; a.params = params;
  a.n_mod = nm; a.nslab = d->HP / 64; a.H = d->hidden; a.HP = d->HP; a.LP = d->LP; a.out_dim = d->out_dim;
  a.T = (const bf16_t *)T; a.S = (long)S; a.U = (char *)U; a.V = (char *)V;
  const long ngrp = (S + NODE_PROJ_SUBJ - 1) / NODE_PROJ_SUBJ;
  const long gu = (long)nm * a.nslab * ngrp, gv = (S + 3) / 4;
  CINEMA_REQUIRE(gu + gv < (1L << 30), "cinema_node_tables: too many subjects (%ld)", (long)S);
  a.g_u = (int)gu;
  grid = (int)(gu + gv);
  smem = 128 * d->LP * 2;
  CINEMA_REQUIRE(smem <= 160 * 1024, "cinema_node_tables: lat_size %d too wide for the LDS staging", d->lat);
  return 0;
}

extern "C" int cinema_node_tables(const cinema_siren_desc *d, const float *params, const void *packed, const float *latents,
                                  const float *cond_subject, int64_t S, int32_t C, int32_t lx, int32_t ly, int32_t lz,
                                  int32_t cond_dims, void *T, void *U, void *V, void *stream) {
  CINEMA_REQUIRE(d && params && packed && latents && T && U && V, "cinema_node_tables: null argument");
  CINEMA_REQUIRE(nodes_eligible(d, lx, ly, lz), "cinema_node_tables: needs a bf16 stack on the activation-stationary path and a "
                 "latent grid of at most %d nodes (got %d x %d x %d)", NODE_ONE_SLOT - 1, lx, ly, lz);
  CINEMA_REQUIRE(S >= 1 && C >= 1 && C + cond_dims == d->lat, "cinema_node_tables: C + cond_dims = %d != lat_size %d", C + cond_dims, d->lat);
  CINEMA_REQUIRE(cond_dims == 0 || cond_subject != nullptr, "cond_dims %d but cond_subject is null", cond_dims);
  hipStream_t s = (hipStream_t)stream;
  NodeRowArgs r = {latents, cond_subject, (long)S, C, lx * ly * lz, cond_dims, d->LP, (bf16_t *)T};
  const long total = (long)S * NODE_SLOTS * (d->LP / 8);
  hipLaunchKernelGGL(k_node_rows, dim3((unsigned)std::min<long>((total + 255) / 256, 4096)), dim3(256), 0, s, r);
  CINEMA_CHECK_HIP(hipGetLastError());
  NodeProjArgs a;
  int grid = 0, smem = 0;
  if (int rc = node_project_args(d, params, packed, T, S, U, V, a, grid, smem)) return rc;
  CINEMA_SMEM_ATTR(k_node_project, smem);
  hipLaunchKernelGGL(k_node_project, dim3((unsigned)grid), dim3(256), smem, s, a);
  CINEMA_CHECK_HIP(hipGetLastError());
  return 0;
}
