// K1: per-subject affine registration + trilinear interpolation of the spatial latent grid.
// Reference arithmetic: utils.py:126-166 (embed2affine / euler2rot),
// models/inr_decoder.py:64-71 (transform), :33 ([idcs_df] gather), :73-85 (grid_sample 5-D,
// bilinear, align_corners=True, padding_mode='border').
#include "common.h"
#include <algorithm>

namespace {

#include "k1_device.h"
#include "step_device.h"
#include "dw_device.h"


__global__ void k_affine_from_embed(const float *embed, long rows, int dim, float *affine) {
  const long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  Embed12 ev;
  float (&e)[12] = ev.v;
#pragma unroll
  for (int k = 0; k < 12; ++k) e[k] = k < dim ? embed[r * dim + k] : 0.f;
  const M3 A = affine_from_embed(ev, dim);
  float *o = affine + r * 12;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) o[3 * i + j] = A.m[i][j];
  o[9] = e[3]; o[10] = e[4]; o[11] = e[5];
}

__global__ void k_affine_bwd(const float *coords, const int32_t *idx, const float *embed, int dim, int per_sample,
                             const float *dct, long N, long S, float *dembed) {
  const long n = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  long row = n;
  if (!per_sample) {
    long s = idx[n];
    row = s < 0 ? 0 : (s >= S ? S - 1 : s);
  }
  float e[12];
  for (int k = 0; k < 12; ++k) e[k] = k < dim ? embed[row * dim + k] : 0.f;
  const float c[3] = {coords[n * 3], coords[n * 3 + 1], coords[n * 3 + 2]};
  const float d[3] = {dct[n * 3], dct[n * 3 + 1], dct[n * 3 + 2]};
  M3 G;  // dL/dA
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) G.m[i][j] = d[i] * c[j];
  float g[12];
  embed_grad(e, dim, G, d, g);
  // one row per sample (the table mode has its own kernel, k_tf_table_bwd)
  for (int k = 0; k < dim; ++k) dembed[n * dim + k] = g[k];
}

__global__ __launch_bounds__(256) void k_node_major(const float *lat, long S, int C, int G, float *shadow) {
  __shared__ __attribute__((aligned(16))) char pool[NODE_MAJOR_LDS_BYTES];
  node_major_body(lat, S, C, G, shadow, blockIdx.x, pool);
}


template <typename T>
__global__ __launch_bounds__(256) void k_trilerp_fwd(TriArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  T *out = (T *)a.lat_out;
  const int G = a.lx * a.ly * a.lz;
  for (long n = (long)blockIdx.x * 4 + wave; n < a.Npad; n += (long)gridDim.x * 4) {
    T *orow = out + n * a.LP;
    if (n >= a.N) {
      if (lane < 3) a.coords_t[n * 3 + lane] = 0.f;
      for (int j = lane; j < a.LP; j += 64) orow[j] = (T)0.f;
      continue;
    }
    const float cx = a.coords[n * 3], cy = a.coords[n * 3 + 1], cz = a.coords[n * 3 + 2];
    long s = a.idx[n];
    s = s < 0 ? 0 : (s >= a.S ? a.S - 1 : s);
    float t0 = cx, t1 = cy, t2 = cz;
    if (a.affine != nullptr) {
      const float *A = a.affine + (a.affine_per_sample ? n : s) * 12;
      t0 = affine_row(A[0], A[1], A[2], A[9], cx, cy, cz);
      t1 = affine_row(A[3], A[4], A[5], A[10], cx, cy, cz);
      t2 = affine_row(A[6], A[7], A[8], A[11], cx, cy, cz);
    }
    if (lane == 0) { a.coords_t[n * 3] = t0; a.coords_t[n * 3 + 1] = t1; a.coords_t[n * 3 + 2] = t2; }
    // coordinate column 0 walks the LAST grid axis (lz), column 1 ly, column 2 lx
    const Axis az = axis_nodes(t0, a.lz), ay = axis_nodes(t1, a.ly), ax = axis_nodes(t2, a.lx);
    int node[8]; float w[8];
    corner_setup(ax, ay, az, a.ly, a.lz, node, w);
    const float *base = a.shadow + s * (long)G * a.C;
    for (int c = lane; c < a.C; c += 64) {
      float v = 0.f;
#pragma unroll
      for (int q = 0; q < 8; ++q) v = __builtin_fmaf(w[q], base[(long)node[q] * a.C + c], v);   // fused, as in the sorted kernel
      orow[c] = (T)v;
    }
    for (int j = lane; j < a.cond_dims; j += 64) orow[a.C + j] = (T)a.cond[n * a.cond_dims + j];
    for (int j = a.C + a.cond_dims + lane; j < a.LP; j += 64) orow[j] = (T)0.f;
  }
}

// gradient w.r.t. the registered coordinates through the interpolation + d cond
struct TriBwdArgs {
  const int32_t *idx; const float *shadow; const float *coords_t; const float *dlat;
  long N, S; int C, lx, ly, lz, cond_dims, LP;
  float *dcoords_t; float *dcond;
  int32_t *sorted_offsets;   // non-NULL: idx is non-decreasing, write offsets[0..S] (first sample of every subject)
};

__device__ __forceinline__ void trilerp_bwd_coord_body(const TriBwdArgs &a, const unsigned vbid, const unsigned vgrid) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int G = a.lx * a.ly * a.lz;
  if (a.sorted_offsets != nullptr) {
    // batch already grouped by subject: the per-subject ranges are a lower_bound per subject (no counting sort)
    for (long t = (long)vbid * 256 + threadIdx.x; t <= a.S; t += (long)vgrid * 256) {
      long lo = 0, hi = a.N;
      while (lo < hi) {
        const long mid = (lo + hi) >> 1;
        if (a.idx[mid] < t) lo = mid + 1; else hi = mid;
      }
      a.sorted_offsets[t] = (int32_t)(t == a.S ? a.N : lo);
    }
  }
  for (long n = (long)vbid * 4 + wave; n < a.N; n += (long)vgrid * 4) {
    const float t0 = a.coords_t[n * 3], t1 = a.coords_t[n * 3 + 1], t2 = a.coords_t[n * 3 + 2];
    long s = a.idx[n];
    s = s < 0 ? 0 : (s >= a.S ? a.S - 1 : s);
    const Axis az = axis_nodes(t0, a.lz), ay = axis_nodes(t1, a.ly), ax = axis_nodes(t2, a.lx);
    const float *base = a.shadow + s * (long)G * a.C;
    const float *grow = a.dlat + n * a.LP;
    float gx = 0.f, gy = 0.f, gz = 0.f;
    if (ax.dmul != 0.f || ay.dmul != 0.f || az.dmul != 0.f) {
      for (int c = lane; c < a.C; c += 64) {
        const float go = grow[c];
        float v[8];
        int q = 0;
#pragma unroll
        for (int bx = 0; bx < 2; ++bx)
#pragma unroll
          for (int by = 0; by < 2; ++by)
#pragma unroll
            for (int bz = 0; bz < 2; ++bz) {
              const int ix = bx ? ax.i1 : ax.i0, iy = by ? ay.i1 : ay.i0, iz = bz ? az.i1 : az.i0;
              v[q++] = base[(long)((ix * a.ly + iy) * a.lz + iz) * a.C + c];
            }
        // v index = bx*4 + by*2 + bz
        const float dx = (ay.w0 * (az.w0 * (v[4] - v[0]) + az.w1 * (v[5] - v[1])) +
                          ay.w1 * (az.w0 * (v[6] - v[2]) + az.w1 * (v[7] - v[3])));
        const float dy = (ax.w0 * (az.w0 * (v[2] - v[0]) + az.w1 * (v[3] - v[1])) +
                          ax.w1 * (az.w0 * (v[6] - v[4]) + az.w1 * (v[7] - v[5])));
        const float dz = (ax.w0 * (ay.w0 * (v[1] - v[0]) + ay.w1 * (v[3] - v[2])) +
                          ax.w1 * (ay.w0 * (v[5] - v[4]) + ay.w1 * (v[7] - v[6])));
        gx += go * dx; gy += go * dy; gz += go * dz;
      }
#pragma unroll
      for (int m = 32; m >= 1; m >>= 1) {
        gx += __shfl_xor(gx, m, 64);
        gy += __shfl_xor(gy, m, 64);
        gz += __shfl_xor(gz, m, 64);
      }
      if (lane == 0) {
        a.dcoords_t[n * 3] += gz * az.dmul;       // column 0 <-> lz axis
        a.dcoords_t[n * 3 + 1] += gy * ay.dmul;
        a.dcoords_t[n * 3 + 2] += gx * ax.dmul;
      }
    }
    if (a.dcond != nullptr)
      for (int j = lane; j < a.cond_dims; j += 64) a.dcond[n * a.cond_dims + j] = grow[a.C + j];
  }
}

__global__ __launch_bounds__(256) void k_trilerp_bwd_coord(TriBwdArgs a) { trilerp_bwd_coord_body(a, blockIdx.x, gridDim.x); }

// The coordinate gradient and the split reduction of the weight-gradient tiles (k_dw256's slabs -> flat gradient) are
// independent and both short: as block ranges of one launch they overlap (29 + 16 us in a row otherwise).
__global__ __launch_bounds__(256) void k_bwd_coord_reduce(TriBwdArgs a, Dw256Args r, int g_coord, int g_red) {
  const int b = blockIdx.x;                      // workgroup-uniform; the streaming job first, the latency-bound one fills in
  if (b < g_red) dw256_reduce_body(r, b);
  else trilerp_bwd_coord_body(a, b - g_red, g_coord);
}


// ---- grouping of the samples by subject (counting sort) ----
// Both passes privatise the per-subject counters in LDS (one global atomic per non-empty bin per block)
// when the subject count fits; each block owns the contiguous sample range [beg, end).
#define GROUP_LDS_BINS 4096
__device__ __forceinline__ long clamp_subject(long s, long S) { return s < 0 ? 0 : (s >= S ? S - 1 : s); }

__global__ __launch_bounds__(256) void k_hist(const int32_t *idx, long N, long S, int32_t *counts) {
  __shared__ int32_t bins[GROUP_LDS_BINS];
  const long per = (N + gridDim.x - 1) / gridDim.x;
  const long beg = blockIdx.x * per, end = beg + per < N ? beg + per : N;
  if (S <= GROUP_LDS_BINS) {
    for (int b = threadIdx.x; b < S; b += blockDim.x) bins[b] = 0;
    __syncthreads();
    for (long n = beg + threadIdx.x; n < end; n += blockDim.x) atomicAdd(bins + clamp_subject(idx[n], S), 1);
    __syncthreads();
    for (int b = threadIdx.x; b < S; b += blockDim.x)
      if (bins[b]) atomicAdd(counts + b, bins[b]);
  } else {
    for (long n = beg + threadIdx.x; n < end; n += blockDim.x) atomicAdd(counts + clamp_subject(idx[n], S), 1);
  }
}
__global__ void k_scan(int32_t *counts, long S, int32_t *offsets) {
  // single block; serial over chunks of blockDim with a running carry (S is small: #subjects)
  __shared__ int32_t buf[1024];
  __shared__ int32_t carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (long b = 0; b < S; b += blockDim.x) {
    const long s = b + threadIdx.x;
    const int32_t v = s < S ? counts[s] : 0;
    buf[threadIdx.x] = v;
    __syncthreads();
    for (int d = 1; d < (int)blockDim.x; d <<= 1) {
      const int32_t t = threadIdx.x >= (unsigned)d ? buf[threadIdx.x - d] : 0;
      __syncthreads();
      buf[threadIdx.x] += t;
      __syncthreads();
    }
    if (s < S) {
      offsets[s] = carry + buf[threadIdx.x] - v;
      counts[s] = 0;   // becomes the fill cursor
    }
    __syncthreads();
    if (threadIdx.x == 0) carry += buf[blockDim.x - 1];
    __syncthreads();
  }
  if (threadIdx.x == 0) offsets[S] = carry;
}
__global__ __launch_bounds__(256) void k_fill(const int32_t *idx, long N, long S, const int32_t *offsets, int32_t *cursor,
                                              int32_t *perm) {
  __shared__ int32_t bins[GROUP_LDS_BINS];   // pass 1: block-local counts, then: next free slot of the block's range
  const long per = (N + gridDim.x - 1) / gridDim.x;
  const long beg = blockIdx.x * per, end = beg + per < N ? beg + per : N;
  if (S <= GROUP_LDS_BINS) {
    for (int b = threadIdx.x; b < S; b += blockDim.x) bins[b] = 0;
    __syncthreads();
    for (long n = beg + threadIdx.x; n < end; n += blockDim.x) atomicAdd(bins + clamp_subject(idx[n], S), 1);
    __syncthreads();
    for (int b = threadIdx.x; b < S; b += blockDim.x) {
      const int c = bins[b];
      bins[b] = c ? offsets[b] + atomicAdd(cursor + b, c) : 0;
    }
    __syncthreads();
    for (long n = beg + threadIdx.x; n < end; n += blockDim.x) {
      const int pos = atomicAdd(bins + clamp_subject(idx[n], S), 1);
      perm[pos] = (int32_t)n;
    }
  } else {
    for (long n = beg + threadIdx.x; n < end; n += blockDim.x) {
      const long s = clamp_subject(idx[n], S);
      perm[offsets[s] + atomicAdd(cursor + s, 1)] = (int32_t)n;
    }
  }
}

// The same grouping in ONE launch of one workgroup, for the batch sizes of a training step (N <= GROUP_SMALL_MAX,
// S <= GROUP_SMALL_BINS): every wave keeps a private histogram in LDS (a shared one serialises: a step has ~400
// samples per subject), the scan over (subject, wave) turns the counts into per-wave fill cursors, and the fill is
// again wave-private.  No memset / scan / fill launches.
#define GROUP_SMALL_MAX 32768
#define GROUP_SMALL_BINS 512
__global__ __launch_bounds__(1024) void k_group_small(const int32_t *idx, long N, long S, int32_t *offsets, int32_t *perm) {
  __shared__ int32_t bins[16 * GROUP_SMALL_BINS];   // [wave][subject]
  __shared__ int32_t part[1024];
  const int tid = threadIdx.x, wave = tid >> 6;
  int32_t *mybins = bins + wave * GROUP_SMALL_BINS;
  for (int b = tid; b < 16 * GROUP_SMALL_BINS; b += 1024) bins[b] = 0;
  // the subject ids of this thread's samples are fetched in one batch (independent loads in flight together) and kept
  int32_t mine[GROUP_SMALL_MAX / 1024];
#pragma unroll
  for (int k = 0; k < GROUP_SMALL_MAX / 1024; ++k) {
    const long n = tid + 1024L * k;
    mine[k] = n < N ? (int32_t)clamp_subject(idx[n], S) : -1;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < GROUP_SMALL_MAX / 1024; ++k)
    if (mine[k] >= 0) atomicAdd(mybins + mine[k], 1);
  __syncthreads();
  // per subject: exclusive prefix over the 16 waves; subject totals go through a block scan
  int total = 0;
  if (tid < S) {
    for (int w = 0; w < 16; ++w) {
      const int c = bins[w * GROUP_SMALL_BINS + tid];
      bins[w * GROUP_SMALL_BINS + tid] = total;
      total += c;
    }
  }
  part[tid] = total;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {
    const int t = tid >= d ? part[tid - d] : 0;
    __syncthreads();
    part[tid] += t;
    __syncthreads();
  }
  if (tid < S) {
    const int off = part[tid] - total;
    offsets[tid] = off;
    for (int w = 0; w < 16; ++w) bins[w * GROUP_SMALL_BINS + tid] += off;   // fill cursor of (wave, subject)
  }
  if (tid == 1023) offsets[S] = part[1023];
  __syncthreads();
#pragma unroll
  for (int k = 0; k < GROUP_SMALL_MAX / 1024; ++k)
    if (mine[k] >= 0) perm[atomicAdd(mybins + mine[k], 1)] = (int32_t)(tid + 1024 * k);
}

// Table mode, one block per subject: sum d c^T and d over the subject's samples (embed_grad is linear in them), then
// ONE evaluation of the embedding gradient per subject.  Replaces the per-sample rows + their per-subject reduction.
// (any workgroup size up to 1024: the body also runs as one extra blockIdx.y slice of k_lat_reduce_dense)
__device__ __forceinline__ void tf_table_bwd_body(const int32_t *offsets, const int32_t *perm, const float *coords,
                                                  const float *dct, const float *embed, int dim, float *dembed, long s) {
  __shared__ float red[16][12];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwave = (blockDim.x + 63) >> 6;
  float acc[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) acc[k] = 0.f;
#pragma unroll 2
  for (int p = offsets[s] + tid; p < offsets[s + 1]; p += blockDim.x) {
    const long n = perm != nullptr ? perm[p] : p;
    const float c[3] = {coords[n * 3], coords[n * 3 + 1], coords[n * 3 + 2]};
    const float d[3] = {dct[n * 3], dct[n * 3 + 1], dct[n * 3 + 2]};
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
      for (int j = 0; j < 3; ++j) acc[i * 3 + j] += d[i] * c[j];
      acc[9 + i] += d[i];
    }
  }
#pragma unroll
  for (int k = 0; k < 12; ++k) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) acc[k] += __shfl_xor(acc[k], m, 64);
    if (lane == 0) red[wave][k] = acc[k];
  }
  __syncthreads();
  if (tid == 0) {
    M3 G;
    float d[3], e[12], g[12], t[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) {
      t[k] = 0.f;
      for (int w = 0; w < nwave; ++w) t[k] += red[w][k];
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
      for (int j = 0; j < 3; ++j) G.m[i][j] = t[i * 3 + j];
      d[i] = t[9 + i];
    }
    for (int k = 0; k < 12; ++k) e[k] = k < dim ? embed[s * dim + k] : 0.f;
    embed_grad(e, dim, G, d, g);
    for (int k = 0; k < dim; ++k) dembed[s * dim + k] = g[k];
  }
}

__global__ __launch_bounds__(256) void k_tf_table_bwd(const int32_t *offsets, const int32_t *perm, const float *coords,
                                                      const float *dct, const float *embed, int dim, float *dembed) {
  tf_table_bwd_body(offsets, perm, coords, dct, embed, dim, dembed, blockIdx.x);
}


// d latents: one workgroup per (subject, 64-channel slice); the subject's samples are summed in LDS
struct LatRedArgs {
  const int32_t *offsets; const int32_t *perm; const float *coords_t; const float *dlat;
  int C, lx, ly, lz, LP; float *dlatents;
  // table-mode embedding gradient riding along as slice blockIdx.y == gridDim.y - 1 of k_lat_reduce_dense (tf_dembed != NULL)
  const float *tf_coords; const float *tf_dct; const float *tf_embed; int tf_dim; float *tf_dembed;
};
__global__ __launch_bounds__(256) void k_lat_reduce(LatRedArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float *acc = reinterpret_cast<float *>(smem_raw);   // [4][G][64]
  const int G = a.lx * a.ly * a.lz;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long s = blockIdx.x;
  const int c0 = blockIdx.y * 64;
  for (int e = threadIdx.x; e < 4 * G * 64; e += 256) acc[e] = 0.f;
  __syncthreads();
  float *mine = acc + wave * G * 64;
  const int beg = a.offsets[s], end = a.offsets[s + 1];
  const int c = c0 + lane;
  // generic grid: wave-private LDS table, plain read-modify-write (LDS float atomics serialise per lane: 3x slower)
  for (int p = beg + wave; p < end; p += 4) {
    const long n = a.perm != nullptr ? a.perm[p] : p;
    const Axis az = axis_nodes(a.coords_t[n * 3], a.lz), ay = axis_nodes(a.coords_t[n * 3 + 1], a.ly),
               ax = axis_nodes(a.coords_t[n * 3 + 2], a.lx);
    int node[8]; float w[8];
    corner_setup(ax, ay, az, a.ly, a.lz, node, w);
    const float v = c < a.C ? a.dlat[n * a.LP + c] : 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) mine[node[q] * 64 + lane] += w[q] * v;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 64 * G; e += 256) {
    const int cc = e / G, nd = e - cc * G;
    if (c0 + cc < a.C) {
      const int o = nd * 64 + cc;
      a.dlatents[(s * a.C + c0 + cc) * G + nd] = acc[o] + acc[G * 64 + o] + acc[2 * G * 64 + o] + acc[3 * G * 64 + o];
    }
  }
}

// Same reduction for a compile-time grid (the reference's 3x3x3 default, and 1x1x1): every lane owns one channel and
// keeps ALL G node sums in registers; the per-sample weights are expanded to dense per-axis vectors so that the
// accumulator index is static (G FMAs per sample, no LDS traffic in the loop).
template <int LX, int LY, int LZ>
__device__ __forceinline__ void dense_axis(const Axis &a, float (&w)[LX > LY ? (LX > LZ ? LX : LZ) : (LY > LZ ? LY : LZ)], int L) {
#pragma unroll
  for (int i = 0; i < (LX > LY ? (LX > LZ ? LX : LZ) : (LY > LZ ? LY : LZ)); ++i)
    w[i] = i < L ? ((i == a.i0 ? a.w0 : 0.f) + ((i == a.i1 && a.i1 != a.i0) ? a.w1 : 0.f)) : 0.f;
}

#define LATRED_WAVES 4     // 512 threads; LDS 27.6 KB (sums) + 28.7 KB (weights): two workgroups per CU, one round at cfg-D
#define LATRED_CHUNK 256
template <int LX, int LY, int LZ>
__global__ __launch_bounds__(128 * LATRED_WAVES) void k_lat_reduce_dense(LatRedArgs a) {
  constexpr int G = LX * LY * LZ;
  constexpr int GS = (G + 3) / 4 * 4;           // weight row stride: float4 broadcasts
  constexpr int LM = LX > LY ? (LX > LZ ? LX : LZ) : (LY > LZ ? LY : LZ);
  // 2 * LATRED_WAVES waves walk the subject's samples; the upper half folds its sums into the lower half's LDS slots.
  // The G dense interpolation weights of a sample are the same for every channel: one thread per sample computes them
  // into LDS (chunks of LATRED_CHUNK samples), the walking waves read them back as broadcasts.
  __shared__ float part[LATRED_WAVES][G][64];
  __shared__ __attribute__((aligned(16))) float wts[LATRED_CHUNK][GS];
  __shared__ int sn[LATRED_CHUNK];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long s = blockIdx.x;
  if (a.tf_dembed != nullptr && blockIdx.y == gridDim.y - 1) {        // workgroup-uniform
    tf_table_bwd_body(a.offsets, a.perm, a.tf_coords, a.tf_dct, a.tf_embed, a.tf_dim, a.tf_dembed, s);
    return;
  }
  const int c0 = blockIdx.y * 64, c = c0 + lane;
  const int beg = a.offsets[s], end = a.offsets[s + 1];
  float acc[G];
#pragma unroll
  for (int k = 0; k < G; ++k) acc[k] = 0.f;
  for (int cb = beg; cb < end; cb += LATRED_CHUNK) {
    const int cnt = end - cb < LATRED_CHUNK ? end - cb : LATRED_CHUNK;
    __syncthreads();                              // the previous chunk's weights have been read
    if ((int)threadIdx.x < cnt) {
      const int p = cb + threadIdx.x;
      const long n = a.perm != nullptr ? a.perm[p] : p;
      const Axis az = axis_nodes(a.coords_t[n * 3], LZ), ay = axis_nodes(a.coords_t[n * 3 + 1], LY),
                 ax = axis_nodes(a.coords_t[n * 3 + 2], LX);
      float wx[LM], wy[LM], wz[LM];
      dense_axis<LX, LY, LZ>(ax, wx, LX);
      dense_axis<LX, LY, LZ>(ay, wy, LY);
      dense_axis<LX, LY, LZ>(az, wz, LZ);
      sn[threadIdx.x] = (int)n;
#pragma unroll
      for (int ix = 0; ix < LX; ++ix)
#pragma unroll
        for (int iy = 0; iy < LY; ++iy)
#pragma unroll
          for (int iz = 0; iz < LZ; ++iz) wts[threadIdx.x][(ix * LY + iy) * LZ + iz] = wx[ix] * wy[iy] * wz[iz];
    }
    __syncthreads();
    // all d-latent values of this wave's samples are requested first (one latency for the chunk), then folded in
    constexpr int PER = LATRED_CHUNK / (2 * LATRED_WAVES);
    float v[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      const int t = wave + 2 * LATRED_WAVES * j;
      v[j] = (t < cnt && c < a.C) ? a.dlat[(long)sn[t] * a.LP + c] : 0.f;
    }
#pragma unroll 2
    for (int j = 0; j < PER; ++j) {
      const int t = wave + 2 * LATRED_WAVES * j;
      if (t >= cnt) break;
      float w[GS];
#pragma unroll
      for (int k4 = 0; k4 < GS / 4; ++k4) {
        const float4 q = *reinterpret_cast<const float4 *>(&wts[t][4 * k4]);
        w[4 * k4] = q.x; w[4 * k4 + 1] = q.y; w[4 * k4 + 2] = q.z; w[4 * k4 + 3] = q.w;
      }
#pragma unroll
      for (int k = 0; k < G; ++k) acc[k] = __builtin_fmaf(w[k], v[j], acc[k]);
    }
  }
  if (wave < LATRED_WAVES) {
#pragma unroll
    for (int k = 0; k < G; ++k) part[wave][k][lane] = acc[k];
  }
  __syncthreads();
  if (wave >= LATRED_WAVES) {
#pragma unroll
    for (int k = 0; k < G; ++k) part[wave - LATRED_WAVES][k][lane] += acc[k];
  }
  __syncthreads();
  // thread = (node, channel) with the channel fastest: conflict-free LDS reads
  for (int e = threadIdx.x; e < 64 * G; e += 128 * LATRED_WAVES) {
    const int nd = e >> 6, cc = e & 63;
    if (c0 + cc < a.C) {
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < LATRED_WAVES; ++w) t += part[w][nd][cc];
      a.dlatents[(s * a.C + c0 + cc) * G + nd] = t;
    }
  }
}

}  // namespace

extern "C" int cinema_affine_from_embedding(const float *embed, int64_t rows, int32_t tf_dim, float *affine, void *stream) {
  CINEMA_REQUIRE(embed && affine, "cinema_affine_from_embedding: null argument");
  CINEMA_REQUIRE(tf_dim == 6 || tf_dim == 9 || tf_dim == 12, "tf embedding width %d unsupported (6, 9 or 12)", tf_dim);
  if (rows <= 0) return 0;
  hipLaunchKernelGGL(k_affine_from_embed, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, (hipStream_t)stream, embed,
                     (long)rows, tf_dim, affine);
  CINEMA_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int cinema_latent_to_node_major(const float *latents, int64_t S, int32_t C, int32_t G, float *shadow, void *stream) {
  CINEMA_REQUIRE(latents && shadow && S >= 1 && C >= 1 && G >= 1, "cinema_latent_to_node_major: bad argument");
  const long nwg = (long)((G + 31) / 32) * ((C + 31) / 32) * S;
  CINEMA_REQUIRE(nwg < (1L << 31), "latent table too large for one launch (%ld tiles)", nwg);
  hipLaunchKernelGGL(k_node_major, dim3((unsigned)nwg), dim3(256), 0, (hipStream_t)stream, latents, (long)S, C, G, shadow);
  CINEMA_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int cinema_affine_trilerp_fwd(const float *coords, const int32_t *idx, const float *affine,
                                         int32_t affine_per_sample, const float *shadow, const float *cond, int64_t N,
                                         int64_t Npad, int64_t S, int32_t C, int32_t lx, int32_t ly, int32_t lz,
                                         int32_t cond_dims, int32_t precision, float *coords_t, void *lat_out, int32_t LP,
                                         void *stream) {
  CINEMA_REQUIRE(coords && idx && shadow && coords_t && lat_out, "cinema_affine_trilerp_fwd: null argument");
  CINEMA_REQUIRE(N >= 1 && Npad >= N && S >= 1, "cinema_affine_trilerp_fwd: bad sizes N=%ld Npad=%ld S=%ld", (long)N, (long)Npad, (long)S);
  CINEMA_REQUIRE(lx >= 1 && ly >= 1 && lz >= 1 && C >= 1, "bad latent grid %d x %d x %d x %d", C, lx, ly, lz);
  CINEMA_REQUIRE(cond_dims == 0 || cond != nullptr, "cond_dims %d but cond is null", cond_dims);
  CINEMA_REQUIRE(LP >= C + cond_dims, "LP %d < C + cond_dims = %d", LP, C + cond_dims);
  TriArgs a = {coords, idx, affine, affine_per_sample, shadow, cond, (long)N, (long)Npad, (long)S, C, lx, ly, lz, cond_dims, LP,
               coords_t, lat_out};
  const unsigned grid = (unsigned)std::min<long>((Npad + 3) / 4, 8192);
  if (precision == CINEMA_PREC_BF16) hipLaunchKernelGGL(k_trilerp_fwd<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(k_trilerp_fwd<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
  CINEMA_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int cinema_affine_trilerp_fwd_sorted(const float *coords, const int32_t *idx, const float *embed, int32_t tf_dim,
                                                const float *latents, const float *cond, int64_t N, int64_t Npad, int64_t S,
                                                int32_t C, int32_t lx, int32_t ly, int32_t lz, int32_t cond_dims,
                                                int32_t cond_by_subject, int32_t precision, float *coords_t, void *lat_out,
                                                int32_t LP, void *stream) {
  CINEMA_REQUIRE(coords && idx && latents && coords_t && lat_out, "cinema_affine_trilerp_fwd_sorted: null argument");
  CINEMA_REQUIRE(N >= 1 && Npad >= N && S >= 1, "cinema_affine_trilerp_fwd_sorted: bad sizes N=%ld Npad=%ld S=%ld", (long)N, (long)Npad, (long)S);
  CINEMA_REQUIRE(lx == 3 && ly == 3 && lz == 3 && C >= 4 && C <= TFS_CMAX && C % 4 == 0 && LP % 4 == 0,
                 "cinema_affine_trilerp_fwd_sorted: needs the 3x3x3 grid, C <= %d, C %% 4 == 0 (got %d x %d x %d x %d); use "
                 "cinema_affine_trilerp_fwd", TFS_CMAX, C, lx, ly, lz);
  CINEMA_REQUIRE(embed == nullptr || tf_dim == 6 || tf_dim == 9 || tf_dim == 12, "tf embedding width %d unsupported (6, 9 or 12)", tf_dim);
  CINEMA_REQUIRE(cond_dims == 0 || cond != nullptr, "cond_dims %d but cond is null", cond_dims);
  CINEMA_REQUIRE(LP >= C + cond_dims, "LP %d < C + cond_dims = %d", LP, C + cond_dims);
  TriFwdSortedArgs a = {coords, idx, embed, tf_dim, latents, cond, (long)N, (long)Npad, (long)S, C, cond_dims, LP, coords_t, lat_out,
                        cond_by_subject ? 1 : 0};
  const unsigned grid = (unsigned)((Npad + TFS_CHUNK - 1) / TFS_CHUNK);
  if (precision == CINEMA_PREC_BF16)
    hipLaunchKernelGGL((k_trilerp_fwd_sorted<bf16_t, 3, 3, 3>), dim3(grid), dim3(TFS_THREADS), 0, (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL((k_trilerp_fwd_sorted<float, 3, 3, 3>), dim3(grid), dim3(TFS_THREADS), 0, (hipStream_t)stream, a);
  CINEMA_CHECK_HIP(hipGetLastError());
  return 0;
}

static int trilerp_bwd_impl(bool sorted, const float *coords, const int32_t *idx, const float *embed, int32_t tf_dim,
                            int32_t embed_per_sample, const float *affine, const float *shadow, const float *coords_t,
                            const float *dlat, float *dcoords_t, int64_t N, int64_t Npad, int64_t S, int32_t C, int32_t lx,
                            int32_t ly, int32_t lz, int32_t cond_dims, int32_t LP, float *dlatents, float *dembed,
                            float *dcond, int32_t *work, void *stream, const Dw256Args *pending = nullptr) {
  (void)affine; (void)Npad;
  CINEMA_REQUIRE(coords && idx && coords_t && dcoords_t, "cinema_affine_trilerp_bwd: null argument");
  CINEMA_REQUIRE(dlat == nullptr || (shadow && work), "cinema_affine_trilerp_bwd: dlat given without shadow/work");
  CINEMA_REQUIRE(dlatents == nullptr || dlat != nullptr, "cinema_affine_trilerp_bwd: dlatents requested without dlat");
  CINEMA_REQUIRE(N >= 1 && S >= 1, "cinema_affine_trilerp_bwd: bad sizes");
  hipStream_t s = (hipStream_t)stream;
  const int G = lx * ly * lz;
  if (dlat != nullptr) {
    TriBwdArgs a = {idx, shadow, coords_t, dlat, (long)N, (long)S, C, lx, ly, lz, cond_dims, LP, dcoords_t, dcond,
                    sorted ? work + S : nullptr};
    const unsigned grid = (unsigned)std::min<long>((N + 3) / 4, 8192);
    if (pending != nullptr && pending->g.nprob > 0) {
      const int t256 = pending->g.tile_start[pending->g.nprob];
      hipLaunchKernelGGL(k_bwd_coord_reduce, dim3(grid + 32u * t256), dim3(256), 0, s, a, *pending, (int)grid, 32 * t256);
    } else {
      hipLaunchKernelGGL(k_trilerp_bwd_coord, dim3(grid), dim3(256), 0, s, a);
    }
    CINEMA_CHECK_HIP(hipGetLastError());
  } else if (pending != nullptr && pending->g.nprob > 0) {
    hipLaunchKernelGGL(k_bwd_coord_reduce, dim3(32u * pending->g.tile_start[pending->g.nprob]), dim3(256), 0, s, TriBwdArgs{}, *pending, 0,
                       32 * pending->g.tile_start[pending->g.nprob]);
    CINEMA_CHECK_HIP(hipGetLastError());
  }
  const bool table_tf = dembed != nullptr && !embed_per_sample;
  int32_t *counts = work, *offsets = work + S, *perm = sorted ? nullptr : work + 2 * S + 2;
  CINEMA_REQUIRE(!sorted || (work != nullptr && dlat != nullptr),
                 "cinema_affine_trilerp_bwd_sorted: needs dlat and the work buffer (offsets are written by the first kernel)");
  if (!sorted && (dlatents != nullptr || table_tf)) {
    CINEMA_REQUIRE(work != nullptr, "cinema_affine_trilerp_bwd: work buffer required");
    if (N <= GROUP_SMALL_MAX && S <= GROUP_SMALL_BINS) {
      hipLaunchKernelGGL(k_group_small, dim3(1), dim3(1024), 0, s, idx, (long)N, (long)S, offsets, perm);
    } else {
      CINEMA_CHECK_HIP(hipMemsetAsync(counts, 0, sizeof(int32_t) * (size_t)S, s));
      const unsigned g1 = (unsigned)std::min<long>((N + 1023) / 1024, 512);
      hipLaunchKernelGGL(k_hist, dim3(g1), dim3(256), 0, s, idx, (long)N, (long)S, counts);
      hipLaunchKernelGGL(k_scan, dim3(1), dim3(1024), 0, s, counts, (long)S, offsets);
      hipLaunchKernelGGL(k_fill, dim3(g1), dim3(256), 0, s, idx, (long)N, (long)S, offsets, counts, perm);
    }
    CINEMA_CHECK_HIP(hipGetLastError());
  }
  const bool dense_grid = (lx == ly && ly == lz && lx >= 1 && lx <= 3);
  const bool tf_rides = table_tf && dlatents != nullptr && dense_grid;    // one launch: extra slice of k_lat_reduce_dense
  if (dembed != nullptr) {
    CINEMA_REQUIRE(embed != nullptr, "dembed requested but embed is null");
    CINEMA_REQUIRE(tf_dim == 6 || tf_dim == 9 || tf_dim == 12, "tf embedding width %d unsupported (6, 9 or 12)", tf_dim);
  }
  if (dembed != nullptr && !tf_rides) {
    if (table_tf)
      hipLaunchKernelGGL(k_tf_table_bwd, dim3((unsigned)S), dim3(256), 0, s, offsets, perm, coords, dcoords_t, embed, tf_dim,
                         dembed);
    else
      hipLaunchKernelGGL(k_affine_bwd, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, coords, idx, embed, tf_dim,
                         embed_per_sample, dcoords_t, (long)N, (long)S, dembed);
    CINEMA_CHECK_HIP(hipGetLastError());
  }
  if (dlatents != nullptr) {
    const int smem = 4 * G * 64 * (int)sizeof(float);
    CINEMA_REQUIRE(smem <= 64 * 1024, "latent grid of %d nodes too large for the LDS reduction (max 64)", G);
    CINEMA_REQUIRE(S <= 0x7fffffff && (C + 63) / 64 <= 65535, "grid too large");
    LatRedArgs a = {offsets, perm, coords_t, dlat, C, lx, ly, lz, LP, dlatents, coords, dcoords_t, embed, tf_dim,
                    tf_rides ? dembed : nullptr};
    const dim3 grid((unsigned)S, (C + 63) / 64 + (tf_rides ? 1 : 0));
    if (lx == 3 && ly == 3 && lz == 3) hipLaunchKernelGGL((k_lat_reduce_dense<3, 3, 3>), grid, dim3(128 * LATRED_WAVES), 0, s, a);
    else if (lx == 1 && ly == 1 && lz == 1) hipLaunchKernelGGL((k_lat_reduce_dense<1, 1, 1>), grid, dim3(128 * LATRED_WAVES), 0, s, a);
    else if (lx == 2 && ly == 2 && lz == 2) hipLaunchKernelGGL((k_lat_reduce_dense<2, 2, 2>), grid, dim3(128 * LATRED_WAVES), 0, s, a);
    else hipLaunchKernelGGL(k_lat_reduce, grid, dim3(256), smem, s, a);
    CINEMA_CHECK_HIP(hipGetLastError());
  }
  return 0;
}

extern "C" int cinema_affine_trilerp_bwd(const float *coords, const int32_t *idx, const float *embed, int32_t tf_dim,
                                         int32_t embed_per_sample, const float *affine, const float *shadow,
                                         const float *coords_t, const float *dlat, float *dcoords_t, int64_t N, int64_t Npad,
                                         int64_t S, int32_t C, int32_t lx, int32_t ly, int32_t lz, int32_t cond_dims, int32_t LP,
                                         float *dlatents, float *dembed, float *dcond, int32_t *work, void *stream) {
  return trilerp_bwd_impl(false, coords, idx, embed, tf_dim, embed_per_sample, affine, shadow, coords_t, dlat, dcoords_t, N,
                          Npad, S, C, lx, ly, lz, cond_dims, LP, dlatents, dembed, dcond, work, stream);
}

extern "C" int cinema_affine_trilerp_bwd_sorted(const float *coords, const int32_t *idx, const float *embed, int32_t tf_dim,
                                                int32_t embed_per_sample, const float *affine, const float *shadow,
                                                const float *coords_t, const float *dlat, float *dcoords_t, int64_t N,
                                                int64_t Npad, int64_t S, int32_t C, int32_t lx, int32_t ly, int32_t lz,
                                                int32_t cond_dims, int32_t LP, float *dlatents, float *dembed, float *dcond,
                                                int32_t *work, void *stream) {
  return trilerp_bwd_impl(true, coords, idx, embed, tf_dim, embed_per_sample, affine, shadow, coords_t, dlat, dcoords_t, N,
                          Npad, S, C, lx, ly, lz, cond_dims, LP, dlatents, dembed, dcond, work, stream);
}

extern "C" int cinema_affine_trilerp_bwd_sorted_pending(const float *coords, const int32_t *idx, const float *embed,
                                                        int32_t tf_dim, int32_t embed_per_sample, const float *affine,
                                                        const float *shadow, const float *coords_t, const float *dlat,
                                                        float *dcoords_t, int64_t N, int64_t Npad, int64_t S, int32_t C,
                                                        int32_t lx, int32_t ly, int32_t lz, int32_t cond_dims, int32_t LP,
                                                        float *dlatents, float *dembed, float *dcond, int32_t *work,
                                                        const void *pending, void *stream) {
  CINEMA_REQUIRE(pending != nullptr, "cinema_affine_trilerp_bwd_sorted_pending: pending is null");
  return trilerp_bwd_impl(true, coords, idx, embed, tf_dim, embed_per_sample, affine, shadow, coords_t, dlat, dcoords_t, N,
                          Npad, S, C, lx, ly, lz, cond_dims, LP, dlatents, dembed, dcond, work, stream,
                          (const Dw256Args *)pending);
}

// ----------------------------------------------------------------------------------------
// cinema_affine_trilerp_bwd for callers that cannot allocate (HIP graph capture, serving loops, harnesses that forbid
// hipMalloc): the grouping scratch (`work` of the call above) is a file-scope device array instead of a caller buffer.
// Every GPU the library runs on holds its own copy (3.3 MiB) until the process ends; the address is looked up once per
// device, with that device current, and only that pointer -- never the array's name -- is handed to kernels / memsets.
// ----------------------------------------------------------------------------------------
#define NOALLOC_WORDS (2L * CINEMA_NOALLOC_MAX_S + 2 + 13L * CINEMA_NOALLOC_MAX_N)
static __device__ __align__(256) int32_t scratch[NOALLOC_WORDS];

static int noalloc_scratch(int32_t **out) {
  static std::atomic<int32_t *> ptr_of[CINEMA_MAX_DEVICES];
  const int dv = cinema_dev();
  int32_t *p = ptr_of[dv].load(std::memory_order_acquire);
  if (p == nullptr) {
    void *sym = nullptr;
    CINEMA_CHECK_HIP(hipGetSymbolAddress(&sym, HIP_SYMBOL(scratch)));
    CINEMA_REQUIRE(sym != nullptr, "cinema_affine_trilerp_bwd_noalloc: the scratch symbol has no address on device %d", dv);
    p = (int32_t *)sym;
    ptr_of[dv].store(p, std::memory_order_release);
  }
  *out = p;
  return 0;
}

extern "C" int cinema_affine_trilerp_bwd_noalloc(const float *coords, const int32_t *idx, const float *embed, int32_t tf_dim,
                                                 int32_t embed_per_sample, const float *affine, const float *shadow,
                                                 const float *coords_t, const float *dlat, float *dcoords_t, int64_t N,
                                                 int64_t Npad, int64_t S, int32_t C, int32_t lx, int32_t ly, int32_t lz,
                                                 int32_t cond_dims, int32_t LP, float *dlatents, float *dembed, float *dcond,
                                                 void *stream) {
  CINEMA_REQUIRE(N >= 1 && N <= CINEMA_NOALLOC_MAX_N && S >= 1 && S <= CINEMA_NOALLOC_MAX_S,
                 "cinema_affine_trilerp_bwd_noalloc: N = %ld, S = %ld exceed the static scratch (N <= %d, S <= %d); use "
                 "cinema_affine_trilerp_bwd with a caller-owned work buffer", (long)N, (long)S, CINEMA_NOALLOC_MAX_N,
                 CINEMA_NOALLOC_MAX_S);
  int32_t *work = nullptr;
  if (int rc = noalloc_scratch(&work)) return rc;
  // nothing in the scratch has to start at zero that the kernels do not write themselves: the one-launch grouping
  // (k_group_small) fills offsets[0..S] and perm[0..N) completely, the three-launch one clears its counters with a
  // hipMemsetAsync on `stream` first; the previous call's contents beyond that are never read.
  return trilerp_bwd_impl(false, coords, idx, embed, tf_dim, embed_per_sample, affine, shadow, coords_t, dlat, dcoords_t, N,
                          Npad, S, C, lx, ly, lz, cond_dims, LP, dlatents, dembed, dcond, work, stream);
}

// ----------------------------------------------------------------------------------------
// gradient of a per-subject condition table from the per-sample d cond rows of a batch sorted by subject
// ----------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void k_cond_table_bwd(const int32_t *idx, const float *dcond, long N, int cond_dims, float *out) {
  const int s = blockIdx.x;
  // [lo, hi) = rows of subject s (idx non-decreasing): two lower bounds
  auto lower = [&](int key) { long a = 0, b = N; while (a < b) { const long m = (a + b) >> 1; if (idx[m] < key) a = m + 1; else b = m; } return a; };
  const long lo = lower(s), hi = lower(s + 1);
  __shared__ float red[4];
  for (int j = 0; j < cond_dims; ++j) {
    float acc = 0.f;
    for (long n = lo + threadIdx.x; n < hi; n += 256) acc += dcond[n * cond_dims + j];
#pragma unroll
    for (int k = 32; k >= 1; k >>= 1) acc += __shfl_xor(acc, k, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) out[(long)s * cond_dims + j] = red[0] + red[1] + red[2] + red[3];
    __syncthreads();
  }
}
}  // namespace

extern "C" int cinema_cond_table_bwd(const int32_t *idx_sorted, const float *dcond, int64_t N, int64_t S, int32_t cond_dims,
                                     float *dtable, void *stream) {
  CINEMA_REQUIRE(idx_sorted && dcond && dtable && N >= 1 && S >= 1 && cond_dims >= 1, "cinema_cond_table_bwd: bad argument");
  hipLaunchKernelGGL(k_cond_table_bwd, dim3((unsigned)S), dim3(256), 0, (hipStream_t)stream, idx_sorted, dcond, (long)N, cond_dims, dtable);
  CINEMA_CHECK_HIP(hipGetLastError());
  return 0;
}
