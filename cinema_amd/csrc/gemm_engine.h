// LDS-tiled MFMA mainloops shared by the SIREN kernels (gfx950).
//
// Orientation.  Every "NT" product is computed as C[f][n] = sum_k Wt[f][k] * X[n][k] with the
// WEIGHTS as the MFMA A operand (rows = output features) and the ACTIVATIONS as the B operand
// (columns = samples).  With the 16x16 accumulator map (col = lane & 15, row = 4*(lane>>4)+reg)
// and the feature rows of each 64-row slab permuted on their way into LDS, a lane ends up
// holding 16 CONSECUTIVE output features of one sample (acc[ft][st][r] <-> feature
// 16*(lane>>4) + 4*ft + r, sample 16*st + (lane&15)), so epilogues read and write saved
// activations with 16-byte vectors instead of 2-byte scalars.
//
// Workgroup = 256 threads = 4 waves (2 x 2), tile = 128 features x 128 samples, each wave
// 64 x 64 = 4 x 4 MFMA tiles.  Double-buffered LDS, register-staged global loads issued one
// k-tile ahead (issue early, write late), one barrier per k-tile.
//   bf16: v_mfma_f32_16x16x32_bf16, k-tile 64, 128-B LDS rows with a 16-B-chunk XOR swizzle
//         (conflict-free ds_read_b128).
//   f32 : v_mfma_f32_16x16x4_f32 (exact f32), k-tile 16, 80-B LDS rows.
//
// The "TN" product C[h][k] = sum_n P[n][h] * Q[n][k] (weight gradients, reduction over the
// samples) keeps both operands in their natural [n][feature] layout in LDS; bf16 fragments
// come from ds_read_b64_tr_b16 (hardware transpose), f32 fragments are one element per lane.
#pragma once
#include "common.h"

#define WG_THREADS 256

// LDS pointers carry their address space explicitly: a generic char* selected at run time (double
// buffering) makes hipcc fall back to flat_load for the fragment reads, which also drains vmcnt.
#define LDS_AS __attribute__((address_space(3)))
typedef LDS_AS char lds_char;

template <typename T>
struct Engine;

// feature row (0..127 within the workgroup tile) -> LDS row: swap the two 2-bit fields
// (f>>4)&3 and (f>>2)&3 inside each 64-row slab.
__device__ __forceinline__ int perm_feature_row(int f) {
  return (f & 0x43) | (((f >> 4) & 3) << 2) | (((f >> 2) & 3) << 4);
}

// ----------------------------------------------------------------------------------------
// bf16
// ----------------------------------------------------------------------------------------
template <>
struct Engine<bf16_t> {
  static constexpr int BK = 64;                   // NT k-tile (elements)
  static constexpr int NT_TILE = 128 * 128;       // bytes per operand tile (128 rows x 128 B)
  static constexpr int NT_SMEM = 4 * NT_TILE;     // 2 operands x 2 buffers
  static constexpr int BN = 64;                   // TN n-tile (samples)
  static constexpr int TN_TILE = 64 * 256;        // 64 rows x 256 B (128 columns)
  static constexpr int TN_SMEM = 4 * TN_TILE;

  struct Stage { u32x4 a[4], b[4]; };

  // A: weights [rowsA x ld] (row-clamped), B: activations [>=128 rows x ld]
  __device__ static __forceinline__ void nt_load(Stage &s, const bf16_t *A, int ldA, int rowsA, const bf16_t *B,
                                                 int ldB, int k0, int tid) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int q = tid + WG_THREADS * it;
      const int row = q >> 3, ch = q & 7;
      const int ra = row < rowsA ? row : rowsA - 1;
      s.a[it] = *reinterpret_cast<const u32x4 *>(A + (size_t)ra * ldA + k0 + ch * 8);
      s.b[it] = *reinterpret_cast<const u32x4 *>(B + (size_t)row * ldB + k0 + ch * 8);
    }
  }
  __device__ static __forceinline__ void nt_store(const Stage &s, lds_char *sA, lds_char *sB, int tid) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int q = tid + WG_THREADS * it;
      const int row = q >> 3, ch = q & 7;
      const int pr = perm_feature_row(row);
      *(LDS_AS u32x4 *)(sA + pr * 128 + ((ch ^ ((pr >> 1) & 7)) << 4)) = s.a[it];
      *(LDS_AS u32x4 *)(sB + row * 128 + ((ch ^ ((row >> 1) & 7)) << 4)) = s.b[it];
    }
  }
  __device__ static __forceinline__ void nt_compute(f32x4 (&acc)[4][4], const lds_char *sA, const lds_char *sB, int wf,
                                                    int wn, int lane) {
    const int i = lane & 15, gk = lane >> 4;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8 a[4], b[4];
      const int ch = kk * 4 + gk;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int ra = wf * 64 + t * 16 + i;
        a[t] = *(const LDS_AS bf16x8 *)(sA + ra * 128 + ((ch ^ ((ra >> 1) & 7)) << 4));
        const int rb = wn * 64 + t * 16 + i;
        b[t] = *(const LDS_AS bf16x8 *)(sB + rb * 128 + ((ch ^ ((rb >> 1) & 7)) << 4));
      }
#pragma unroll
      for (int ft = 0; ft < 4; ++ft)
#pragma unroll
        for (int st = 0; st < 4; ++st)
          acc[ft][st] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[ft], b[st], acc[ft][st], 0, 0, 0);
    }
  }

  // ---- TN ----
  // P: [n][colsP] (ld ldP), tile columns h0..h0+127 (column-chunk clamped to the row), same for Q.
  __device__ static __forceinline__ void tn_load(Stage &s, const bf16_t *P, int ldP, int h0, const bf16_t *Q, int ldQ,
                                                 int c0, int n0, int tid) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int q = tid + WG_THREADS * it;
      const int row = q >> 4, ch = q & 15;
      int hp = h0 + ch * 8; if (hp + 8 > ldP) hp = 0;
      int cq = c0 + ch * 8; if (cq + 8 > ldQ) cq = 0;
      s.a[it] = *reinterpret_cast<const u32x4 *>(P + (size_t)(n0 + row) * ldP + hp);
      s.b[it] = *reinterpret_cast<const u32x4 *>(Q + (size_t)(n0 + row) * ldQ + cq);
    }
  }
  __device__ static __forceinline__ int tn_off(int row, int colbyte) {
    const int key = (row & 3) | (((row >> 3) & 1) << 2);
    return row * 256 + ((((colbyte >> 4) ^ (key << 1)) << 4) | (colbyte & 15));
  }
  __device__ static __forceinline__ void tn_store(const Stage &s, lds_char *sP, lds_char *sQ, int tid) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int q = tid + WG_THREADS * it;
      const int row = q >> 4, ch = q & 15;
      const int off = tn_off(row, ch * 16);
      *(LDS_AS u32x4 *)(sP + off) = s.a[it];
      *(LDS_AS u32x4 *)(sQ + off) = s.b[it];
    }
  }
  __device__ static __forceinline__ bf16x8 tr_frag(const lds_char *tile, int nbase, int colbase, int lane) {
    // lanes 16g..16g+15 fetch rows nbase+8g+{0..7} of 16 columns; lane 4q+p addresses row q, columns 4p..4p+3
    const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
    const int r0 = nbase + 8 * g + q;
    typedef LDS_AS s16x4 lds_s16x4;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (lds_s16x4 *)(tile + tn_off(r0, (colbase + 4 * p) * 2)));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (lds_s16x4 *)(tile + tn_off(r0 + 4, (colbase + 4 * p) * 2)));
    s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
  }
  // nht: number of 16-row output tiles of this wave that hold valid rows (wave-uniform; narrow problems skip the rest)
  __device__ static __forceinline__ void tn_compute(f32x4 (&acc)[4][4], const lds_char *sP, const lds_char *sQ, int wh, int wk,
                                                    int lane, int nht) {
    if (nht <= 0) return;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8 a[4], b[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        a[t] = tr_frag(sP, kk * 32, wh * 64 + t * 16, lane);
        b[t] = tr_frag(sQ, kk * 32, wk * 64 + t * 16, lane);
      }
#pragma unroll
      for (int ht = 0; ht < 4; ++ht)
        if (ht < nht) {
#pragma unroll
          for (int kt = 0; kt < 4; ++kt)
            acc[ht][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[ht], b[kt], acc[ht][kt], 0, 0, 0);
        }
    }
  }
};

// ----------------------------------------------------------------------------------------
// f32 (exact): one element per lane per fragment
// ----------------------------------------------------------------------------------------
template <>
struct Engine<float> {
  static constexpr int BK = 16;
  static constexpr int NT_ROW = 20;                       // floats per LDS row (16 + pad)
  static constexpr int NT_TILE = 128 * NT_ROW * 4;
  static constexpr int NT_SMEM = 4 * NT_TILE;
  static constexpr int BN = 16;
  static constexpr int TN_ROW = 144;                      // floats per LDS row (128 + pad)
  static constexpr int TN_TILE = 16 * TN_ROW * 4;
  static constexpr int TN_SMEM = 4 * TN_TILE;

  struct Stage { f32x4 a[2], b[2]; };

  __device__ static __forceinline__ void nt_load(Stage &s, const float *A, int ldA, int rowsA, const float *B,
                                                 int ldB, int k0, int tid) {
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int q = tid + WG_THREADS * it;
      const int row = q >> 2, ch = q & 3;
      const int ra = row < rowsA ? row : rowsA - 1;
      s.a[it] = *reinterpret_cast<const f32x4 *>(A + (size_t)ra * ldA + k0 + ch * 4);
      s.b[it] = *reinterpret_cast<const f32x4 *>(B + (size_t)row * ldB + k0 + ch * 4);
    }
  }
  __device__ static __forceinline__ void nt_store(const Stage &s, lds_char *sA, lds_char *sB, int tid) {
    LDS_AS float *fA = (LDS_AS float *)sA, *fB = (LDS_AS float *)sB;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int q = tid + WG_THREADS * it;
      const int row = q >> 2, ch = q & 3;
      *(LDS_AS f32x4 *)(fA + perm_feature_row(row) * NT_ROW + ch * 4) = s.a[it];
      *(LDS_AS f32x4 *)(fB + row * NT_ROW + ch * 4) = s.b[it];
    }
  }
  __device__ static __forceinline__ void nt_compute(f32x4 (&acc)[4][4], const lds_char *sA, const lds_char *sB, int wf,
                                                    int wn, int lane) {
    const LDS_AS float *fA = (const LDS_AS float *)sA, *fB = (const LDS_AS float *)sB;
    const int i = lane & 15, gk = lane >> 4;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      float a[4], b[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        a[t] = fA[(wf * 64 + t * 16 + i) * NT_ROW + kk * 4 + gk];
        b[t] = fB[(wn * 64 + t * 16 + i) * NT_ROW + kk * 4 + gk];
      }
#pragma unroll
      for (int ft = 0; ft < 4; ++ft)
#pragma unroll
        for (int st = 0; st < 4; ++st)
          acc[ft][st] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ft], b[st], acc[ft][st], 0, 0, 0);
    }
  }

  __device__ static __forceinline__ void tn_load(Stage &s, const float *P, int ldP, int h0, const float *Q, int ldQ,
                                                 int c0, int n0, int tid) {
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int q = tid + WG_THREADS * it;
      const int row = q >> 5, ch = q & 31;
      int hp = h0 + ch * 4; if (hp + 4 > ldP) hp = 0;
      int cq = c0 + ch * 4; if (cq + 4 > ldQ) cq = 0;
      s.a[it] = *reinterpret_cast<const f32x4 *>(P + (size_t)(n0 + row) * ldP + hp);
      s.b[it] = *reinterpret_cast<const f32x4 *>(Q + (size_t)(n0 + row) * ldQ + cq);
    }
  }
  __device__ static __forceinline__ void tn_store(const Stage &s, lds_char *sP, lds_char *sQ, int tid) {
    LDS_AS float *fP = (LDS_AS float *)sP, *fQ = (LDS_AS float *)sQ;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int q = tid + WG_THREADS * it;
      const int row = q >> 5, ch = q & 31;
      *(LDS_AS f32x4 *)(fP + row * TN_ROW + ch * 4) = s.a[it];
      *(LDS_AS f32x4 *)(fQ + row * TN_ROW + ch * 4) = s.b[it];
    }
  }
  __device__ static __forceinline__ void tn_compute(f32x4 (&acc)[4][4], const lds_char *sP, const lds_char *sQ, int wh, int wk,
                                                    int lane, int nht) {
    if (nht <= 0) return;
    const LDS_AS float *fP = (const LDS_AS float *)sP, *fQ = (const LDS_AS float *)sQ;
    const int i = lane & 15, gk = lane >> 4;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      float a[4], b[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        a[t] = fP[(kk * 4 + gk) * TN_ROW + wh * 64 + t * 16 + i];
        b[t] = fQ[(kk * 4 + gk) * TN_ROW + wk * 64 + t * 16 + i];
      }
#pragma unroll
      for (int ht = 0; ht < 4; ++ht)
        if (ht < nht) {
#pragma unroll
          for (int kt = 0; kt < 4; ++kt)
            acc[ht][kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ht], b[kt], acc[ht][kt], 0, 0, 0);
        }
    }
  }
};

// acc += Wt[f0.., :K] * X[n0.., :K]^T over K (a multiple of Engine::BK).  All 256 threads call it.
template <typename T>
__device__ __forceinline__ void nt_mainloop(f32x4 (&acc)[4][4], const T *A, int ldA, int rowsA, const T *B, int ldB,
                                            int K, char *smem) {
  using E = Engine<T>;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wf = wave >> 1, wn = wave & 1;
  lds_char *base = (lds_char *)smem;   // buffer b: A tile at b*2*TILE, B tile right behind it
  typename E::Stage st;
  const int nt = K / E::BK;
  __syncthreads();  // LDS may still be read by a previous mainloop of this workgroup
  E::nt_load(st, A, ldA, rowsA, B, ldB, 0, tid);
  E::nt_store(st, base, base + E::NT_TILE, tid);
  __syncthreads();
  for (int t = 0; t < nt; ++t) {
    const int cur = t & 1;
    if (t + 1 < nt) E::nt_load(st, A, ldA, rowsA, B, ldB, (t + 1) * E::BK, tid);
    E::nt_compute(acc, base + cur * 2 * E::NT_TILE, base + cur * 2 * E::NT_TILE + E::NT_TILE, wf, wn, lane);
    if (t + 1 < nt) E::nt_store(st, base + (cur ^ 1) * 2 * E::NT_TILE, base + (cur ^ 1) * 2 * E::NT_TILE + E::NT_TILE, tid);
    __syncthreads();
  }
}

// acc += sum over segments s of A_s[f0.., :K_s] * B_s[n0.., :K_s]^T: ONE pipeline over the k-tiles of all segments
// (two register stages, loads issued two tiles ahead, crossing segment boundaries), so a chain of short GEMMs that
// accumulate into the same tile pays one prologue instead of one per segment.
struct NtSeg { const void *A; const void *B; int K; };   // A: packed [rowsA x K] (ld = K), B: [rows x ldB]

template <typename T>
__device__ __forceinline__ void nt_mainloop_segs(f32x4 (&acc)[4][4], const NtSeg *segs, int nseg, int f0, int rowsA,
                                                 long n0, int ldB, char *smem) {
  using E = Engine<T>;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wf = wave >> 1, wn = wave & 1;
  lds_char *base = (lds_char *)smem;
  typename E::Stage s0, s1;
  int nt = 0;
  for (int s = 0; s < nseg; ++s) nt += segs[s].K / E::BK;
  if (nt == 0) return;
  int ls = 0, lk = 0;   // cursor of the next tile to load
  auto load_next = [&](typename E::Stage &st) {
    const NtSeg sg = segs[ls];
    E::nt_load(st, (const T *)sg.A + (size_t)f0 * sg.K, sg.K, rowsA - f0, (const T *)sg.B + (size_t)n0 * ldB, ldB, lk, tid);
    lk += E::BK;
    if (lk >= sg.K) { lk = 0; ++ls; }
  };
  __syncthreads();  // LDS may still be read by a previous mainloop of this workgroup
  load_next(s0);
  if (nt > 1) load_next(s1);
  E::nt_store(s0, base, base + E::NT_TILE, tid);
  __syncthreads();
#define NT_ITER(SFREE, SNEXT, CUR)                                                                            \
  {                                                                                                           \
    if (t + 2 < nt) load_next(SFREE);                                                                         \
    E::nt_compute(acc, base + (CUR) * 2 * E::NT_TILE, base + (CUR) * 2 * E::NT_TILE + E::NT_TILE, wf, wn, lane); \
    if (t + 1 < nt)                                                                                           \
      E::nt_store(SNEXT, base + ((CUR) ^ 1) * 2 * E::NT_TILE, base + ((CUR) ^ 1) * 2 * E::NT_TILE + E::NT_TILE, tid); \
    __syncthreads();                                                                                          \
  }
  int t = 0;
  while (true) {
    NT_ITER(s0, s1, 0)
    if (++t >= nt) break;
    NT_ITER(s1, s0, 1)
    if (++t >= nt) break;
  }
#undef NT_ITER
}

// acc += P[nbeg..nend, h0..]^T * Q[nbeg..nend, c0..]; nbeg/nend multiples of Engine::BN.
// Two register stages: the loads of n-tile t+2 are issued while tile t is being multiplied and tile t+1 waits in the
// other stage, so every global load has two full iterations to land (one stage leaves these kernels latency-bound:
// about one n-tile per HBM round trip and workgroup).
template <typename T>
__device__ __forceinline__ void tn_mainloop(f32x4 (&acc)[4][4], const T *P, int ldP, int h0, const T *Q, int ldQ,
                                            int c0, int nbeg, int nend, int nht, char *smem) {
  using E = Engine<T>;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wh = wave >> 1, wk = wave & 1;
  lds_char *base = (lds_char *)smem;
  typename E::Stage s0, s1;
  if (nbeg >= nend) return;  // uniform per workgroup
  const int last = nend - E::BN;
  E::tn_load(s0, P, ldP, h0, Q, ldQ, c0, nbeg, tid);
  E::tn_load(s1, P, ldP, h0, Q, ldQ, c0, nbeg + E::BN <= last ? nbeg + E::BN : last, tid);
  E::tn_store(s0, base, base + E::TN_TILE, tid);
  __syncthreads();
  // tile t sits in LDS buffer (t & 1); stage s_(t & 1) is free, stage s_((t + 1) & 1) holds tile t + 1
#define TN_ITER(SFREE, SNEXT, CUR)                                                                            \
  {                                                                                                           \
    const int n2_ = n0 + 2 * E::BN;                                                                           \
    E::tn_load(SFREE, P, ldP, h0, Q, ldQ, c0, n2_ <= last ? n2_ : last, tid);                                 \
    E::tn_compute(acc, base + (CUR) * 2 * E::TN_TILE, base + (CUR) * 2 * E::TN_TILE + E::TN_TILE, wh, wk, lane, nht); \
    if (n0 + E::BN < nend)                                                                                    \
      E::tn_store(SNEXT, base + ((CUR) ^ 1) * 2 * E::TN_TILE, base + ((CUR) ^ 1) * 2 * E::TN_TILE + E::TN_TILE, tid); \
    __syncthreads();                                                                                          \
  }
  int n0 = nbeg;
  while (true) {
    TN_ITER(s0, s1, 0)
    n0 += E::BN;
    if (n0 >= nend) break;
    TN_ITER(s1, s0, 1)
    n0 += E::BN;
    if (n0 >= nend) break;
  }
#undef TN_ITER
}
