// Device helpers of the activation-stationary chain kernels (fused.hip, fused_nodes.hip): the per-wave weight ring, the
// k32 step, LDS feature vectors, packing.  Included INSIDE the anonymous namespace of each translation unit, after
// fused.h and gemm_engine.h.
#pragma once

// ----------------------------------------------------------------------------------------
// device: weight ring + segment runner
// ----------------------------------------------------------------------------------------
// Addresses are kept as wave-uniform 64-bit bases (SGPRs) + 32-bit per-lane byte offsets so that the only VGPRs
// alive across the layer loop are the accumulators, the ring and a handful of offsets.
struct WRing {
  bf16x8 s[FUSED_RING][4];
  const char *wp;  // uniform: next step to fetch
};

__device__ __forceinline__ void ring_fetch(bf16x8 (&slot)[4], const char *p, unsigned lane16) {
#pragma unroll
  for (int ft = 0; ft < 4; ++ft) slot[ft] = *reinterpret_cast<const bf16x8 *>(p + ft * 1024 + lane16);
}

__device__ __forceinline__ void ring_init(WRing &r, const char *stream, unsigned lane16) {
#pragma unroll
  for (int k = 0; k < FUSED_RING; ++k) ring_fetch(r.s[k], stream + k * 4096, lane16);
  r.wp = stream + FUSED_RING * 4096;
}

// 16-byte chunk c of row `row` of an LDS tile sits at chunk position c ^ (row & 15) (rows are multiples of 256 B):
// the 16 lanes of every ds_read_b128 lane group then cover all 64 banks exactly once.
#define FUSED_STEP(SLOT, KSTEP)                                                                              \
  {                                                                                                          \
    const int off_ = ((((KSTEP) << 2) + gk) ^ i) << 4;                                                       \
    bf16x8 b_[NST];                                                                                          \
    _Pragma("unroll") for (int st = 0; st < NST; ++st)                                                       \
      b_[st] = *(const LDS_AS bf16x8 *)(tile + rowoff + st * 16 * rowbytes + off_);                          \
    _Pragma("unroll") for (int ft = 0; ft < 4; ++ft)                                                         \
      _Pragma("unroll") for (int st = 0; st < NST; ++st)                                                     \
        acc[ft][st] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(r.s[SLOT][ft], b_[st], acc[ft][st], 0, 0, 0);  \
    ring_fetch(r.s[SLOT], r.wp, lane16);                                                                     \
    r.wp += 4096;                                                                                            \
    __builtin_amdgcn_sched_barrier(0); /* keep the refill here: hipcc otherwise sinks all refills of a group */ \
  }

// acc[f][n] += sum_k A_stream[f][k] * tile[n][k] over nsteps k32 steps of the wave's weight stream.  Every segment
// is padded to a multiple of FUSED_RING steps (zero weights against zeroed LDS columns), so the ring phase is always
// 0 at a segment boundary and the slots are addressed statically (a run-time slot index makes hipcc keep every slot
// twice and spill; with static slots its own counted vmcnt(15..12) waits come out exact).
template <int NST>
__device__ __forceinline__ void run_seg(f32x4 (&acc)[4][NST], WRing &r, const lds_char *tile, int rowbytes, int nsteps,
                                        int i, int gk, unsigned lane16) {
  const int rowoff = i * rowbytes;
  for (int k = 0; k < nsteps; k += 4) {
    FUSED_STEP(0, k) FUSED_STEP(1, k + 1) FUSED_STEP(2, k + 2) FUSED_STEP(3, k + 3)
  }
}
static_assert(FUSED_RING == 4, "run_seg is written for a 4-step ring");

template <int NST>
__device__ __forceinline__ void fill_acc(f32x4 (&acc)[4][NST], const f32x4 (&v)[4]) {
#pragma unroll
  for (int ft = 0; ft < 4; ++ft)
#pragma unroll
    for (int st = 0; st < NST; ++st) acc[ft][st] = v[ft];
}

// v[q] = bias[off + fb + 4q .. +3] from the LDS copy of the biases (each vector zero-padded to HP)
__device__ __forceinline__ void feature_vec(const lds_char *bias, int off, unsigned fb, f32x4 (&v)[4]) {
#pragma unroll
  for (int q = 0; q < 4; ++q) v[q] = *(const LDS_AS f32x4 *)(bias + ((off + fb) * 4u + q * 16u));
}

__device__ __forceinline__ void pack16(const float (&v)[16], bf16x8 &lo, bf16x8 &hi) {
#pragma unroll
  for (int j = 0; j < 8; ++j) { lo[j] = (bf16_t)v[j]; hi[j] = (bf16_t)v[8 + j]; }
}


// OCP fp8 (e4m3) copies of the tensors only the backward's epilogue reads (c, z, s): 16 values = one 16-byte vector.
// The values are scaled into the middle of the format's range first (the backward folds the inverse into its constants).
__device__ __forceinline__ u32x4 pack16_fp8(const float (&v)[16], float scale) {
  u32x4 o;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    int w = 0;
    w = __builtin_amdgcn_cvt_pk_fp8_f32(v[4 * q] * scale, v[4 * q + 1] * scale, w, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(v[4 * q + 2] * scale, v[4 * q + 3] * scale, w, true);
    o[q] = (unsigned)w;
  }
  return o;
}
__device__ __forceinline__ void unpack16_fp8(const u32x4 &p, float (&v)[16]) {
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const f32x2_t lo = __builtin_amdgcn_cvt_pk_f32_fp8((int)p[q], false), hi = __builtin_amdgcn_cvt_pk_f32_fp8((int)p[q], true);
    v[4 * q] = lo[0]; v[4 * q + 1] = lo[1]; v[4 * q + 2] = hi[0]; v[4 * q + 3] = hi[1];
  }
}
#define FP8_SCALE_C 256.f
#define FP8_SCALE_ZS 16.f
