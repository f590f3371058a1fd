// K3: ground-truth sampling fused with the loss reduction and dLoss/dOutput, plus the
// dense-inference epilogue.
// Reference arithmetic: utils.py:77-104 (Criterion), data_loading/dataset.py:87-89 (integer
// voxel gather of every modality), models/inr_decoder.py:50-57 (inference post-processing).
#include "common.h"
#include <algorithm>

namespace {

#include "step_device.h"

__global__ void k_class_weight_sum(const float *target, int tcols, const int64_t *voxel, const int32_t *subj,
                                   const uint8_t *labels, long V, long N, int n_cls, const float *class_w, float *sums) {
  float acc = 0.f;
  for (long n = (long)blockIdx.x * blockDim.x + threadIdx.x; n < N; n += (long)gridDim.x * blockDim.x) {
    int y = target ? (int)target[n * tcols + tcols - 1] : (int)labels[(long)subj[n] * V + voxel[n]];
    y = y < 0 ? 0 : (y >= n_cls ? n_cls - 1 : y);
    acc += class_w[y];
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) atomicAdd(sums + 2, acc);
}

__global__ __launch_bounds__(256) void k_step_prepare(StepPrepareArgs a) {
  __shared__ __attribute__((aligned(16))) char pool[STEP_PREPARE_LDS_BYTES];
  step_prepare_body(a, blockIdx.x, gridDim.x, pool);
}

struct LossArgs {
  const float *out; const float *target; const int64_t *voxel; const int32_t *subj; const float *intens;
  const uint8_t *labels; long V, N, Npad; int sr_dims, n_cls, metric; const float *class_w;
  float sr_scale, seg_scale; float *sums; float *dout; float *target_out; int wsum_on_device;
  const float *norm;   // optional fp32[2]: {samples, class-weight sum} that normalise the means (data-parallel ranks), else N / sums[2]
};

template <int MAXC>
__global__ __launch_bounds__(256) void k_sample_loss(LossArgs a) {
  const int od = a.sr_dims + a.n_cls;
  const int tcols = a.sr_dims + 1;
  float acc_sr = 0.f, acc_ce = 0.f, acc_w = 0.f;
  const float wtot = a.norm ? a.norm[1] : (a.wsum_on_device ? a.sums[2] : (float)a.N);
  const float inv_sr = a.sr_scale / ((a.norm ? a.norm[0] : (float)a.N) * (float)a.sr_dims);
  const float inv_w = wtot > 0.f ? a.seg_scale / wtot : 0.f;
  for (long n = (long)blockIdx.x * blockDim.x + threadIdx.x; n < a.Npad; n += (long)gridDim.x * blockDim.x) {
    float *drow = a.dout + n * od;
    if (n >= a.N) {
      for (int j = 0; j < od; ++j) drow[j] = 0.f;
      continue;
    }
    const float *orow = a.out + n * od;
    int y;
    long vbase = 0;
    if (a.target) {
      y = (int)a.target[n * tcols + a.sr_dims];
    } else {
      vbase = (long)a.subj[n] * a.V + a.voxel[n];
      y = (int)a.labels[vbase];
    }
    for (int m = 0; m < a.sr_dims; ++m) {
      const float t = a.target ? a.target[n * tcols + m] : a.intens[vbase * a.sr_dims + m];
      if (a.target_out) a.target_out[n * tcols + m] = t;
      const float d = orow[m] - t;
      if (a.metric == 0) {
        acc_sr += d * d;
        drow[m] = 2.f * d * inv_sr;
      } else {
        acc_sr += fabsf(d);
        drow[m] = (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) * inv_sr;
      }
    }
    if (a.target_out) a.target_out[n * tcols + a.sr_dims] = (float)y;
    if (a.seg_scale > 0.f) {
      // a label outside [0, n_cls) makes torch's CrossEntropyLoss of the reference abort (device assert); here it
      // poisons the sample's loss term and gradient row with NaN so that it cannot pass unnoticed
      const bool bad_label = y < 0 || y >= a.n_cls;
      y = y < 0 ? 0 : (y >= a.n_cls ? a.n_cls - 1 : y);
      float lg[MAXC];
      float mx = -INFINITY;
#pragma unroll
      for (int j = 0; j < MAXC; ++j) {
        lg[j] = j < a.n_cls ? orow[a.sr_dims + j] : -INFINITY;
        mx = fmaxf(mx, lg[j]);
      }
      float se = 0.f, ly = 0.f;
#pragma unroll
      for (int j = 0; j < MAXC; ++j) {
        const float e = j < a.n_cls ? expf(lg[j] - mx) : 0.f;
        se += e;
        lg[j] = e;
      }
      ly = orow[a.sr_dims + y];
      const float lse = mx + logf(se);
      const float w = a.class_w ? a.class_w[y] : 1.f;
      acc_ce += bad_label ? __builtin_nanf("") : w * (lse - ly);
      acc_w += w;
      const float k = bad_label ? __builtin_nanf("") : w * inv_w, inv_se = 1.f / se;
#pragma unroll
      for (int j = 0; j < MAXC; ++j)
        if (j < a.n_cls) drow[a.sr_dims + j] = k * (lg[j] * inv_se - (j == y ? 1.f : 0.f));
    } else {
      for (int j = 0; j < a.n_cls; ++j) drow[a.sr_dims + j] = 0.f;
    }
  }
  __shared__ float red[3][4];
  acc_sr = wave_sum(acc_sr);
  acc_ce = wave_sum(acc_ce);
  acc_w = wave_sum(acc_w);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { red[0][wave] = acc_sr; red[1][wave] = acc_ce; red[2][wave] = acc_w; }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicAdd(a.sums + 0, red[0][0] + red[0][1] + red[0][2] + red[0][3]);
    atomicAdd(a.sums + 1, red[1][0] + red[1][1] + red[1][2] + red[1][3]);
    atomicAdd(a.sums + 3, red[2][0] + red[2][1] + red[2][2] + red[2][3]);
  }
}

// ---- dense-inference epilogue ----
__global__ void k_infer_minmax(const float *out, long G, int od, int sr, float *minmax) {
  // values are clamped to [0,1] first, so unsigned-int ordering equals float ordering
  float mn[8], mx[8];
  for (int m = 0; m < 8; ++m) { mn[m] = 1.f; mx[m] = 0.f; }
  for (long n = (long)blockIdx.x * blockDim.x + threadIdx.x; n < G; n += (long)gridDim.x * blockDim.x)
    for (int m = 0; m < sr; ++m) {
      const float v = fminf(fmaxf(out[n * od + m], 0.f), 1.f);
      mn[m] = fminf(mn[m], v);
      mx[m] = fmaxf(mx[m], v);
    }
  for (int m = 0; m < sr; ++m) {
    float a = mn[m], b = mx[m];
#pragma unroll
    for (int k = 32; k >= 1; k >>= 1) {
      a = fminf(a, __shfl_xor(a, k, 64));
      b = fmaxf(b, __shfl_xor(b, k, 64));
    }
    if ((threadIdx.x & 63) == 0) {
      atomicMin(reinterpret_cast<unsigned int *>(minmax) + m, __float_as_uint(a));
      atomicMax(reinterpret_cast<unsigned int *>(minmax) + sr + m, __float_as_uint(b));
    }
  }
}

template <int MAXC>
__global__ void k_infer_finish(const float *out, long G, int od, int sr, const float *minmax, float *vol) {
  const int ncls = od - sr, vc = sr + 1 + ncls;
  for (long n = (long)blockIdx.x * blockDim.x + threadIdx.x; n < G; n += (long)gridDim.x * blockDim.x) {
    const float *o = out + n * od;
    float *v = vol + n * vc;
    for (int m = 0; m < sr; ++m) {
      const float x = fminf(fmaxf(o[m], 0.f), 1.f);
      v[m] = (x - minmax[m]) / (minmax[sr + m] - minmax[m]);   // 0/0 -> NaN like the reference
    }
    float lg[MAXC], mx = -INFINITY;
    int arg = 0;
#pragma unroll
    for (int j = 0; j < MAXC; ++j) {
      lg[j] = j < ncls ? o[sr + j] : -INFINITY;
      if (lg[j] > mx) { mx = lg[j]; arg = j; }   // first maximum wins, like torch.argmax
    }
    float se = 0.f;
#pragma unroll
    for (int j = 0; j < MAXC; ++j) { lg[j] = j < ncls ? expf(lg[j] - mx) : 0.f; se += lg[j]; }
    v[sr] = (float)arg;
    const float inv = 1.f / se;
#pragma unroll
    for (int j = 0; j < MAXC; ++j)
      if (j < ncls) v[sr + 1 + j] = lg[j] * inv;
  }
}

// chunked form of the two kernels above: one pass per chunk of network outputs writes the clamped intensities, argmax and
// softmax straight into the volume rows and folds the chunk into the running min / max; k_infer_rescale then maps the
// intensity channels of the finished volume with the whole-volume min / max (same formula, same rounding).
template <int MAXC>
__global__ __launch_bounds__(256) void k_infer_chunk(const float *out, long G, int od, int sr, float *minmax, float *vol) {
  // 256 rows per workgroup pass: the rows' od floats come in and their od + 1 floats go out as ONE contiguous run each
  // (coalesced 16-byte accesses through LDS; a thread per row straight on global memory touches 48 lines per load)
  __shared__ __attribute__((aligned(16))) float sin_[256 * 24], sout[256 * 25];
  const int ncls = od - sr, vc = sr + 1 + ncls;
  float mn[8], mxv[8];
  for (int m = 0; m < 8; ++m) { mn[m] = 1.f; mxv[m] = 0.f; }
  for (long r0 = (long)blockIdx.x * 256; r0 < G; r0 += (long)gridDim.x * 256) {
    const int rows = (int)(G - r0 < 256 ? G - r0 : 256);
    const float *src = out + r0 * od;
    const int nin = rows * od, nout = rows * vc;
    __syncthreads();
    if ((((size_t)src) & 15) == 0) {
      for (int e = threadIdx.x * 4; e < nin; e += 1024) {
        if (e + 4 <= nin) *reinterpret_cast<float4 *>(sin_ + e) = *reinterpret_cast<const float4 *>(src + e);
        else for (int q = e; q < nin; ++q) sin_[q] = src[q];
      }
    } else {
      for (int e = threadIdx.x; e < nin; e += 256) sin_[e] = src[e];
    }
    __syncthreads();
    if ((int)threadIdx.x < rows) {
      const float *o = sin_ + threadIdx.x * od;
      float *v = sout + threadIdx.x * vc;
      for (int m = 0; m < sr; ++m) {
        const float x = fminf(fmaxf(o[m], 0.f), 1.f);
        v[m] = x;
        mn[m] = fminf(mn[m], x);
        mxv[m] = fmaxf(mxv[m], x);
      }
      float lg[MAXC], mx = -INFINITY;
      int arg = 0;
#pragma unroll
      for (int j = 0; j < MAXC; ++j) {
        lg[j] = j < ncls ? o[sr + j] : -INFINITY;
        if (lg[j] > mx) { mx = lg[j]; arg = j; }   // first maximum wins, like torch.argmax
      }
      float se = 0.f;
#pragma unroll
      for (int j = 0; j < MAXC; ++j) { lg[j] = j < ncls ? expf(lg[j] - mx) : 0.f; se += lg[j]; }
      v[sr] = (float)arg;
      const float inv = 1.f / se;
#pragma unroll
      for (int j = 0; j < MAXC; ++j)
        if (j < ncls) v[sr + 1 + j] = lg[j] * inv;
    }
    __syncthreads();
    float *dst = vol + r0 * vc;
    if ((((size_t)dst) & 15) == 0) {
      for (int e = threadIdx.x * 4; e < nout; e += 1024) {
        if (e + 4 <= nout) *reinterpret_cast<float4 *>(dst + e) = *reinterpret_cast<const float4 *>(sout + e);
        else for (int q = e; q < nout; ++q) dst[q] = sout[q];
      }
    } else {
      for (int e = threadIdx.x; e < nout; e += 256) dst[e] = sout[e];
    }
  }
  // one atomic per workgroup and value (same-address atomics serialise: one per wave cost 0.3 ms per 2^19 rows)
  __shared__ float red[2][8][4];
  for (int m = 0; m < sr; ++m) {
    float a = mn[m], b = mxv[m];
#pragma unroll
    for (int k = 32; k >= 1; k >>= 1) {
      a = fminf(a, __shfl_xor(a, k, 64));
      b = fmaxf(b, __shfl_xor(b, k, 64));
    }
    if ((threadIdx.x & 63) == 0) { red[0][m][threadIdx.x >> 6] = a; red[1][m][threadIdx.x >> 6] = b; }
  }
  __syncthreads();
  if ((int)threadIdx.x < sr) {
    const int m = threadIdx.x;
    const float a = fminf(fminf(red[0][m][0], red[0][m][1]), fminf(red[0][m][2], red[0][m][3]));
    const float b = fmaxf(fmaxf(red[1][m][0], red[1][m][1]), fmaxf(red[1][m][2], red[1][m][3]));
    atomicMin(reinterpret_cast<unsigned int *>(minmax) + m, __float_as_uint(a));
    atomicMax(reinterpret_cast<unsigned int *>(minmax) + sr + m, __float_as_uint(b));
  }
}

__global__ __launch_bounds__(256) void k_infer_rescale(float *vol, long G, int vc, int sr, const float *minmax) {
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < G * sr; e += (long)gridDim.x * blockDim.x) {
    const long n = e / sr;
    const int m = (int)(e - n * sr);
    float *v = vol + n * vc + m;
    *v = (*v - minmax[m]) / (minmax[sr + m] - minmax[m]);   // 0/0 -> NaN like the reference
  }
}

// coordinates of a run of the dense evaluation grid, 'ij' (x slowest) flattening of three axis vectors
__global__ __launch_bounds__(256) void k_grid_coords(const float *ax, const float *ay, const float *az, long ny, long nz,
                                                     long start, long count, float *coords) {
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < count; e += (long)gridDim.x * blockDim.x) {
    const long f = start + e, i = f / (ny * nz), r = f - i * ny * nz, j = r / nz, k = r - j * nz;
    coords[3 * e + 0] = ax[i];
    coords[3 * e + 1] = ay[j];
    coords[3 * e + 2] = az[k];
  }
}

// ---- connected components of a 3-d mask (6-connectivity), labels = 1 + smallest linear index of the component ----
__global__ __launch_bounds__(256) void k_cc_init(const uint8_t *mask, long V, int32_t *lab) {
  for (long v = (long)blockIdx.x * 256 + threadIdx.x; v < V; v += (long)gridDim.x * 256) lab[v] = mask[v] ? (int32_t)(v + 1) : 0;
}

// one sweep: take the smallest label among a voxel and its 6 neighbours, then follow the label chain (a label names a
// voxel of the same component whose own label is smaller or equal) a few hops.  Labels only ever decrease and only to
// labels of the same component, so in-place races cost sweeps, never correctness; *changed counts the updates.
__global__ __launch_bounds__(256) void k_cc_sweep(int32_t *lab, int X, int Y, int Z, int32_t *changed) {
  const long V = (long)X * Y * Z;
  int any = 0;
  for (long v = (long)blockIdx.x * 256 + threadIdx.x; v < V; v += (long)gridDim.x * 256) {
    const int32_t own = lab[v];
    if (own == 0) continue;
    const int z = (int)(v % Z), y = (int)((v / Z) % Y), x = (int)(v / ((long)Z * Y));
    int32_t m = own;
    auto look = [&](long u) { const int32_t l = lab[u]; if (l != 0 && l < m) m = l; };
    if (z > 0) look(v - 1);
    if (z + 1 < Z) look(v + 1);
    if (y > 0) look(v - Z);
    if (y + 1 < Y) look(v + Z);
    if (x > 0) look(v - (long)Z * Y);
    if (x + 1 < X) look(v + (long)Z * Y);
#pragma unroll
    for (int hop = 0; hop < 4; ++hop) {
      const int32_t up = lab[m - 1];
      if (up != 0 && up < m) m = up;
    }
    if (m < own) { atomicMin(lab + v, m); atomicMin(lab + (own - 1), m); any = 1; }
  }
  if (__ballot(any) != 0ull && (threadIdx.x & 63) == 0) atomicAdd(changed, 1);
}

}  // namespace

static int sample_loss_impl(bool prepared, const float *out, const float *target, const int64_t *voxel, const int32_t *subj,
                            const float *intens, const uint8_t *labels, int64_t V, int64_t N, int64_t Npad,
                            int32_t sr_dims, int32_t n_cls, int32_t loss_metric, const float *class_w,
                            float sr_scale, float seg_scale, float *sums, float *dout, float *target_out,
                            void *stream, const float *norm = nullptr) {
  CINEMA_REQUIRE(out && sums && dout, "cinema_sample_loss_fwd_bwd: null argument");
  CINEMA_REQUIRE(target || (voxel && subj && intens && labels), "cinema_sample_loss_fwd_bwd: need a dense target or volumes");
  CINEMA_REQUIRE(N >= 1 && Npad >= N, "cinema_sample_loss_fwd_bwd: bad sizes");
  CINEMA_REQUIRE(sr_dims >= 1 && sr_dims <= 8 && n_cls >= 1 && n_cls <= 16, "unsupported out_dim split %d + %d", sr_dims, n_cls);
  CINEMA_REQUIRE(loss_metric == 0 || loss_metric == 1, "loss_metric must be 0 (mse) or 1 (l1)");
  hipStream_t s = (hipStream_t)stream;
  if (!prepared) CINEMA_CHECK_HIP(hipMemsetAsync(sums, 0, 4 * sizeof(float), s));
  const unsigned grid = (unsigned)std::min<long>((Npad + 255) / 256, 2048);
  const int wdev = (class_w != nullptr && seg_scale > 0.f) ? 1 : 0;
  if (wdev && !prepared) {
    hipLaunchKernelGGL(k_class_weight_sum, dim3(grid), dim3(256), 0, s, target, sr_dims + 1, voxel, subj, labels, (long)V,
                       (long)N, n_cls, class_w, sums);
    CINEMA_CHECK_HIP(hipGetLastError());
  }
  LossArgs a = {out, target, voxel, subj, intens, labels, (long)V, (long)N, (long)Npad, sr_dims, n_cls, loss_metric, class_w,
                sr_scale, seg_scale, sums, dout, target_out, wdev, norm};
  hipLaunchKernelGGL(k_sample_loss<16>, dim3(grid), dim3(256), 0, s, a);
  CINEMA_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int cinema_sample_loss_fwd_bwd(const float *out, const float *target, const int64_t *voxel, const int32_t *subj,
                                          const float *intens, const uint8_t *labels, int64_t V, int64_t N, int64_t Npad,
                                          int32_t sr_dims, int32_t n_cls, int32_t loss_metric, const float *class_w,
                                          float sr_scale, float seg_scale, float *sums, float *dout, float *target_out,
                                          void *stream) {
  return sample_loss_impl(false, out, target, voxel, subj, intens, labels, V, N, Npad, sr_dims, n_cls, loss_metric, class_w,
                          sr_scale, seg_scale, sums, dout, target_out, stream);
}

extern "C" int cinema_sample_loss_fwd_bwd_prepared(const float *out, const float *target, const int64_t *voxel,
                                                   const int32_t *subj, const float *intens, const uint8_t *labels, int64_t V,
                                                   int64_t N, int64_t Npad, int32_t sr_dims, int32_t n_cls, int32_t loss_metric,
                                                   const float *class_w, float sr_scale, float seg_scale, float *sums,
                                                   float *dout, float *target_out, void *stream) {
  return sample_loss_impl(true, out, target, voxel, subj, intens, labels, V, N, Npad, sr_dims, n_cls, loss_metric, class_w,
                          sr_scale, seg_scale, sums, dout, target_out, stream);
}

extern "C" int cinema_sample_loss_fwd_bwd_normed(const float *out, const float *target, const int64_t *voxel,
                                                 const int32_t *subj, const float *intens, const uint8_t *labels, int64_t V,
                                                 int64_t N, int64_t Npad, int32_t sr_dims, int32_t n_cls, int32_t loss_metric,
                                                 const float *class_w, float sr_scale, float seg_scale, float *sums,
                                                 float *dout, float *target_out, const float *norm, void *stream) {
  CINEMA_REQUIRE(norm != nullptr, "cinema_sample_loss_fwd_bwd_normed: norm is NULL");
  return sample_loss_impl(true, out, target, voxel, subj, intens, labels, V, N, Npad, sr_dims, n_cls, loss_metric, class_w,
                          sr_scale, seg_scale, sums, dout, target_out, stream, norm);
}

extern "C" int cinema_step_prepare(float *dcoords, int64_t n_dcoords, float *sums, const int64_t *voxel, const int32_t *subj,
                                   const uint8_t *labels, int64_t V, int64_t N, int32_t n_cls, const float *class_w,
                                   float *scratch, void *stream) {
  CINEMA_REQUIRE(sums != nullptr && scratch != nullptr && n_dcoords >= 0 && (dcoords != nullptr || n_dcoords == 0),
                 "cinema_step_prepare: bad argument");
  CINEMA_REQUIRE(class_w == nullptr || (voxel && subj && labels && N >= 1 && n_cls >= 1),
                 "cinema_step_prepare: class_w needs the label source");
  const long work = std::max<long>(class_w != nullptr ? (long)N : 0, (long)n_dcoords);
  const unsigned grid = (unsigned)std::max<long>(1, std::min<long>((work + 255) / 256, PREP_MAX_GRID));
  StepPrepareArgs a = {dcoords, (long)n_dcoords, sums, voxel, subj, labels, (long)V, (long)N, n_cls, class_w, scratch};
  hipLaunchKernelGGL(k_step_prepare, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
  CINEMA_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int cinema_infer_minmax(const float *out, int64_t G, int32_t out_dim, int32_t sr_dims, float *minmax, void *stream) {
  CINEMA_REQUIRE(out && minmax && G >= 1 && sr_dims >= 1 && sr_dims <= 8 && out_dim > sr_dims, "cinema_infer_minmax: bad argument");
  const unsigned grid = (unsigned)std::min<long>((G + 255) / 256, 4096);
  hipLaunchKernelGGL(k_infer_minmax, dim3(grid), dim3(256), 0, (hipStream_t)stream, out, (long)G, out_dim, sr_dims, minmax);
  CINEMA_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int cinema_infer_finish(const float *out, int64_t G, int32_t out_dim, int32_t sr_dims, const float *minmax,
                                   float *vol, void *stream) {
  CINEMA_REQUIRE(out && minmax && vol && G >= 1 && sr_dims >= 1 && out_dim > sr_dims && out_dim - sr_dims <= 16,
                 "cinema_infer_finish: bad argument");
  const unsigned grid = (unsigned)std::min<long>((G + 255) / 256, 8192);
  hipLaunchKernelGGL(k_infer_finish<16>, dim3(grid), dim3(256), 0, (hipStream_t)stream, out, (long)G, out_dim, sr_dims, minmax, vol);
  CINEMA_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int cinema_infer_chunk(const float *out, int64_t G, int32_t out_dim, int32_t sr_dims, float *minmax, float *vol,
                                  void *stream) {
  CINEMA_REQUIRE(out && minmax && vol && G >= 1 && sr_dims >= 1 && sr_dims <= 8 && out_dim > sr_dims && out_dim - sr_dims <= 16,
                 "cinema_infer_chunk: bad argument");
  const unsigned grid = (unsigned)std::min<long>((G + 255) / 256, 1024);
  hipLaunchKernelGGL(k_infer_chunk<16>, dim3(grid), dim3(256), 0, (hipStream_t)stream, out, (long)G, out_dim, sr_dims, minmax, vol);
  CINEMA_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int cinema_infer_rescale(float *vol, int64_t G, int32_t out_dim, int32_t sr_dims, const float *minmax, void *stream) {
  CINEMA_REQUIRE(vol && minmax && G >= 1 && sr_dims >= 1 && sr_dims <= 8 && out_dim > sr_dims, "cinema_infer_rescale: bad argument");
  const unsigned grid = (unsigned)std::min<long>((G * sr_dims + 255) / 256, 8192);
  hipLaunchKernelGGL(k_infer_rescale, dim3(grid), dim3(256), 0, (hipStream_t)stream, vol, (long)G, out_dim + 1, sr_dims, minmax);
  CINEMA_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int cinema_grid_coords(const float *ax, const float *ay, const float *az, int64_t nx, int64_t ny, int64_t nz,
                                  int64_t start, int64_t count, float *coords, void *stream) {
  CINEMA_REQUIRE(ax && ay && az && coords && nx >= 1 && ny >= 1 && nz >= 1, "cinema_grid_coords: bad argument");
  CINEMA_REQUIRE(start >= 0 && count >= 1 && start + count <= nx * ny * nz, "cinema_grid_coords: run [%ld, %ld) leaves the grid",
                 (long)start, (long)(start + count));
  const unsigned grid = (unsigned)std::min<long>((count + 255) / 256, 4096);
  hipLaunchKernelGGL(k_grid_coords, dim3(grid), dim3(256), 0, (hipStream_t)stream, ax, ay, az, (long)ny, (long)nz, (long)start,
                     (long)count, coords);
  CINEMA_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int cinema_cc_label(const uint8_t *mask, int32_t X, int32_t Y, int32_t Z, int32_t *labels, int32_t *changed,
                               int32_t sweeps, int32_t init, void *stream) {
  CINEMA_REQUIRE(mask && labels && changed && X >= 1 && Y >= 1 && Z >= 1 && sweeps >= 0 && (long)X * Y * Z < (1L << 31) - 1,
                 "cinema_cc_label: bad argument");
  const long V = (long)X * Y * Z;
  const unsigned grid = (unsigned)std::min<long>((V + 255) / 256, 8192);
  hipStream_t s = (hipStream_t)stream;
  if (init) hipLaunchKernelGGL(k_cc_init, dim3(grid), dim3(256), 0, s, mask, V, labels);
  CINEMA_CHECK_HIP(hipMemsetAsync(changed, 0, sizeof(int32_t), s));
  for (int k = 0; k < sweeps; ++k) hipLaunchKernelGGL(k_cc_sweep, dim3(grid), dim3(256), 0, s, labels, X, Y, Z, changed);
  CINEMA_CHECK_HIP(hipGetLastError());
  return 0;
}
