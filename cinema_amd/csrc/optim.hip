// Fused AdamW over a few large flat tensors in ONE launch (reference: torch.optim.AdamW as configured in
// build_atlas.py:432-453: latents, transformations, MLP parameters, each with its own learning rate).
// torch's multi-tensor kernel gives a single 2 M-element tensor only ~32 workgroups; this one is grid-strided
// with 16-byte accesses.
#include "common.h"
#include <algorithm>
#include <cmath>

namespace {

struct AdamArgs {
  cinema_adamw_group g[CINEMA_ADAMW_MAX_GROUPS];
  float bc1[CINEMA_ADAMW_MAX_GROUPS], bc2_rsqrt[CINEMA_ADAMW_MAX_GROUPS];
  int n_groups, clear_grads;
};

__device__ __forceinline__ void adam_one(float &p, float g, float &m, float &v, const cinema_adamw_group &c, float step_size,
                                         float bc2_rsqrt) {
  // torch.optim.AdamW (decoupled weight decay, no amsgrad, maximize=False)
  p *= 1.f - c.lr * c.weight_decay;
  m = c.beta1 * m + (1.f - c.beta1) * g;
  v = c.beta2 * v + (1.f - c.beta2) * g * g;
  const float denom = sqrtf(v) * bc2_rsqrt + c.eps;
  p -= step_size * (m / denom);
}

__global__ __launch_bounds__(256) void k_adamw(AdamArgs a) {
  const cinema_adamw_group c = a.g[blockIdx.y];
  const float step_size = c.lr / a.bc1[blockIdx.y];
  const float r2 = a.bc2_rsqrt[blockIdx.y];
  const long n4 = c.n >> 2;
  const bool aligned = ((((uintptr_t)c.p | (uintptr_t)c.g | (uintptr_t)c.m | (uintptr_t)c.v) & 15) == 0);
  long done = 0;
  if (aligned) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
      f32x4 p = reinterpret_cast<f32x4 *>(c.p)[i];
      const f32x4 g = reinterpret_cast<const f32x4 *>(c.g)[i];
      f32x4 m = reinterpret_cast<f32x4 *>(c.m)[i];
      f32x4 v = reinterpret_cast<f32x4 *>(c.v)[i];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float pk = p[k], mk = m[k], vk = v[k];
        adam_one(pk, g[k], mk, vk, c, step_size, r2);
        p[k] = pk; m[k] = mk; v[k] = vk;
      }
      reinterpret_cast<f32x4 *>(c.p)[i] = p;
      reinterpret_cast<f32x4 *>(c.m)[i] = m;
      reinterpret_cast<f32x4 *>(c.v)[i] = v;
      if (a.clear_grads) { const f32x4 z = {0.f, 0.f, 0.f, 0.f}; reinterpret_cast<f32x4 *>(const_cast<float *>(c.g))[i] = z; }
    }
    done = n4 << 2;
  }
  for (long i = done + (long)blockIdx.x * blockDim.x + threadIdx.x; i < c.n; i += (long)gridDim.x * blockDim.x) {
    adam_one(c.p[i], c.g[i], c.m[i], c.v[i], c, step_size, r2);
    if (a.clear_grads) const_cast<float *>(c.g)[i] = 0.f;
  }
}

}  // namespace

static int adamw_impl(const cinema_adamw_group *groups, int32_t n_groups, int clear_grads, void *stream) {
  CINEMA_REQUIRE(groups != nullptr && n_groups >= 1 && n_groups <= CINEMA_ADAMW_MAX_GROUPS,
                 "cinema_adamw_step: need 1..%d groups", CINEMA_ADAMW_MAX_GROUPS);
  AdamArgs a = {};
  long nmax = 0;
  for (int i = 0; i < n_groups; ++i) {
    const cinema_adamw_group &g = groups[i];
    CINEMA_REQUIRE(g.p && g.g && g.m && g.v && g.n >= 0 && g.step >= 1, "cinema_adamw_step: bad group %d", i);
    a.g[i] = g;
    a.bc1[i] = (float)(1.0 - std::pow((double)g.beta1, (double)g.step));
    a.bc2_rsqrt[i] = (float)(1.0 / std::sqrt(1.0 - std::pow((double)g.beta2, (double)g.step)));
    nmax = std::max<long>(nmax, g.n);
  }
  a.n_groups = n_groups;
  a.clear_grads = clear_grads;
  if (nmax == 0) return 0;
  const unsigned gx = (unsigned)std::min<long>((nmax / 4 + 255) / 256 + 1, 2048);
  hipLaunchKernelGGL(k_adamw, dim3(gx, n_groups), dim3(256), 0, (hipStream_t)stream, a);
  CINEMA_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int cinema_adamw_step(const cinema_adamw_group *groups, int32_t n_groups, void *stream) {
  return adamw_impl(groups, n_groups, 0, stream);
}

extern "C" int cinema_adamw_step_clear(const cinema_adamw_group *groups, int32_t n_groups, void *stream) {
  return adamw_impl(groups, n_groups, 1, stream);
}
