// Device helpers of the wide chain kernels (fused_wide.hip, fused_nodes.hip): a wave takes its 64 output features in two
// passes of 32 (one fragment pair each), the weight ring holds eight steps of two fragments.  Included INSIDE the anonymous
// namespace of each translation unit, after fused.h and gemm_engine.h.
#pragma once

#define WIDE_RING 8               // steps (of two fragments = 2 KB per wave) in flight
#define DRAIN_VMEM() __builtin_amdgcn_s_waitcnt(0x0F70)   // s_waitcnt vmcnt(0) only: gfx9 encoding

struct WRing2 { bf16x8 s[WIDE_RING][2]; const char *wp; };

__device__ __forceinline__ void ring2_fetch(bf16x8 (&slot)[2], const char *p, unsigned lane16) {
  slot[0] = *reinterpret_cast<const bf16x8 *>(p + lane16);
  slot[1] = *reinterpret_cast<const bf16x8 *>(p + 1024 + lane16);
}

#define WIDE_STEP(SLOT, KSTEP)                                                                               \
  {                                                                                                          \
    int gi_s = gi;                                                                                           \
    asm volatile("" : "+v"(gi_s)); /* per-step addresses are formed here, not in front of the pass (spilled there) */ \
    const int off_ = (((KSTEP) << 2) ^ gi_s) << 4;   /* = (((KSTEP << 2) + gk) ^ i) << 4: one xor with a constant */ \
    bf16x8 b_[NST];                                                                                          \
    _Pragma("unroll") for (int st = 0; st < NST; ++st)                                                       \
      b_[st] = *(const LDS_AS bf16x8 *)(tile + rowoff + st * 16 * rowbytes + off_);                          \
    _Pragma("unroll") for (int ft = 0; ft < 2; ++ft)                                                         \
      _Pragma("unroll") for (int st = 0; st < NST; ++st)                                                     \
        acc[ft][st] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(r.s[SLOT][ft], b_[st], acc[ft][st], 0, 0, 0);  \
    ring2_fetch(r.s[SLOT], r.wp, lane16);                                                                    \
    r.wp += 4096;                                                                                            \
    __builtin_amdgcn_sched_barrier(0);                                                                       \
  }
#define WIDE_GROUP(K) WIDE_STEP(0, K) WIDE_STEP(1, K + 1) WIDE_STEP(2, K + 2) WIDE_STEP(3, K + 3) \
                      WIDE_STEP(4, K + 4) WIDE_STEP(5, K + 5) WIDE_STEP(6, K + 6) WIDE_STEP(7, K + 7)

// One pass: acc[q][n] += sum_k A[fragment pair][k] * tile[n][k] over nsteps (a multiple of 8) k32 steps.  The ring holds
// the pass's first eight steps on entry; the refills of its last group are the first eight steps of the NEXT pass
// (`next`: the other fragment pair of the same segment, or the first pair of the next segment), so the ring phase is 0 at
// every pass boundary and the slots are addressed statically.  The first group is peeled: the epilogue in front of it
// drained the counter, hipcc then knows the eight resident steps need no wait, and that epilogue's stores have eight
// steps of MFMA time to retire before the loop's first counted wait (fused_nodes.hip, run_seg_peeled).
template <int NST, int NSTEPS>
__device__ __forceinline__ void run_pass(f32x4 (&acc)[2][NST], WRing2 &r, const lds_char *tile, int rowbytes, int i, int gk,
                                         unsigned lane16, const char *start, const char *next) {
  static_assert(NSTEPS == 8 || NSTEPS == 16, "run_pass is written out for 8 or 16 steps (HP = 256 or 512)");
  const int rowoff = i * rowbytes;
  const int gi = gk ^ i;
  __builtin_amdgcn_sched_barrier(0);
  // straight-line code, no loop header: hipcc's counted vmcnt waits come out exact in every group (as a loop of one
  // iteration the second group opened with vmcnt(0), i.e. with the latency of the refill just issued)
  if constexpr (NSTEPS == 16) {
    r.wp = start + WIDE_RING * 4096;
    { WIDE_GROUP(0) }
    r.wp = next;
    { WIDE_GROUP(8) }
  } else {
    r.wp = next;
    { WIDE_GROUP(0) }
  }
}

