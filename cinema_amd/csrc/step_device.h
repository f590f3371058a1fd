// Device bodies shared between their own kernels (trilerp.hip, loss.hip, siren.hip) and the one-launch step head of
// fused.hip.  Included INSIDE the anonymous namespace of each translation unit.  (vbid, vgrid) = index / count of the
// workgroups doing the job; every body is written for 256-thread workgroups.
#pragma once

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}

// latent table (S, C, G) -> node-major shadow (S, G, C): one workgroup per (subject, 32 x 32 tile) through LDS;
// vbid = tile_g + tiles_g * (tile_c + tiles_c * subject)
#define NODE_MAJOR_LDS_BYTES (32 * 33 * 4)
__device__ __forceinline__ void node_major_body(const float *lat, long S, int C, int G, float *shadow, unsigned vbid, char *pool) {
  float (*tile)[33] = reinterpret_cast<float (*)[33]>(pool);      // [32][33] in the caller's __shared__ pool
  const unsigned tg = (G + 31) / 32, tc = (C + 31) / 32;
  const long s = vbid / (tg * tc);
  const int c0 = (int)((vbid / tg) % tc) * 32, g0 = (int)(vbid % tg) * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 256 threads: 32 x 8
  if (s >= S) return;
  for (int r = ty; r < 32; r += 8) {
    const int c = c0 + r, g = g0 + tx;
    tile[r][tx] = (c < C && g < G) ? lat[(s * C + c) * G + g] : 0.f;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int g = g0 + r, c = c0 + tx;
    if (g < G && c < C) shadow[(s * G + g) * C + c] = tile[tx][r];
  }
}

// Per-step clearing + class-weight normaliser: every workgroup zeroes its share of dcoords and leaves the partial sum of
// w[label] over its samples in scratch; the workgroup that finishes last (a counter in scratch, which it resets for the
// next call) adds the partials in workgroup order and writes sums = {0, 0, sum w, 0}.  No zero-then-accumulate ordering
// between launches, no float atomics: the same bits every run.
#define PREP_MAX_GRID 256
struct StepPrepareArgs {
  float *dcoords; long n_dcoords; float *sums; const int64_t *voxel; const int32_t *subj; const uint8_t *labels; long V, N;
  int n_cls; const float *class_w; float *scratch;
};
#define STEP_PREPARE_LDS_BYTES 32
__device__ __forceinline__ void step_prepare_body(const StepPrepareArgs &a, unsigned vbid, unsigned vgrid, char *pool) {
  float *red = reinterpret_cast<float *>(pool);                   // [4] + one flag, in the caller's __shared__ pool
  int &last = *reinterpret_cast<int *>(pool + 16);
  const int tid = threadIdx.x;
  float acc = 0.f;
  if (a.class_w != nullptr) {
    for (long n = (long)vbid * 256 + tid; n < a.N; n += (long)vgrid * 256) {
      int y = (int)a.labels[(long)a.subj[n] * a.V + a.voxel[n]];
      y = y < 0 ? 0 : (y >= a.n_cls ? a.n_cls - 1 : y);
      acc += a.class_w[y];
    }
  }
  for (long e = (long)vbid * 256 + tid; e < a.n_dcoords; e += (long)vgrid * 256) a.dcoords[e] = 0.f;
  acc = wave_sum(acc);
  if ((tid & 63) == 0) red[tid >> 6] = acc;
  __syncthreads();
  unsigned int *counter = reinterpret_cast<unsigned int *>(a.scratch + PREP_MAX_GRID);
  if (tid == 0) {
    a.scratch[vbid] = red[0] + red[1] + red[2] + red[3];
    __threadfence();
    last = atomicAdd(counter, 1u) == vgrid - 1;
  }
  __syncthreads();
  if (last && tid < 64) {
    __threadfence();
    float t = 0.f;
    for (int b = tid; b < (int)vgrid; b += 64) t += __builtin_nontemporal_load(a.scratch + b);
    t = wave_sum(t);
    if (tid == 0) {
      a.sums[0] = 0.f; a.sums[1] = 0.f; a.sums[2] = t; a.sums[3] = 0.f;
      *counter = 0u;
    }
  }
}

// fp32 master weights -> zero-padded T copies (optionally transposed); 64 workgroups per job: vbid = wg + 64 * job
struct PackJob { const float *src; long dst; int rows, cols, ld, rowsP, colsP, transpose; };
template <typename T>
__device__ __forceinline__ void pack_job_body(const PackJob &j, void *packed, unsigned wg, unsigned nwg) {
  T *dst = (T *)packed + j.dst;
  const long total = (long)j.rowsP * j.colsP;
  for (long e = (long)wg * 256 + threadIdx.x; e < total; e += (long)nwg * 256) {
    const int r = (int)(e / j.colsP), cc = (int)(e - (long)r * j.colsP);
    // destination element (r, cc) of a [rowsP x colsP] matrix; source is (r, cc) or (cc, r)
    const int sr = j.transpose ? cc : r, sc = j.transpose ? r : cc;
    const float v = (sr < j.rows && sc < j.cols) ? j.src[(size_t)sr * j.ld + sc] : 0.f;
    dst[e] = (T)v;
  }
}
