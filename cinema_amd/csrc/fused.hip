// Activation-stationary SIREN forward for gfx950 (bf16 MFMA, fp32 accumulate).
// Reference arithmetic: models/siren.py:26-36 (SineLayer.forward), :59-61 (Siren.forward).
//
// One workgroup = 8 waves owns FUSED_TS = 64 samples for the WHOLE stack.  The layer input x [64 x HP] and the
// interpolated latent rows [64 x LP] live in LDS for the lifetime of the workgroup, so activations are never
// re-read from HBM between layers; the saved tensors the backward needs (y, c, and z, s of modulated layers)
// are streamed out once.  Wave w owns output features [64w, 64w+64) of every layer: its MFMA A fragments come
// straight from global memory (L2) through a register ring fed by one contiguous per-wave weight stream
// (fused.h), its B fragments are conflict-free ds_read_b128 from the shared x / lat tiles.
#include "fused.h"
#include "nodegrad.h"
#include "gemm_engine.h"
#include <algorithm>
#include <cstdlib>

namespace {

#include "k1_device.h"
#include "step_device.h"
#include "noderows_device.h"
#include "ngprep_device.h"
#include "dw_device.h"

inline int round_up_i(int x, int m) { return (x + m - 1) / m * m; }
inline int ksteps(int k) { return round_up_i(k / 32, FUSED_RING); }   // k32 steps of a segment, padded to the ring

// ----------------------------------------------------------------------------------------
// stream plans (host)
// ----------------------------------------------------------------------------------------
struct Seg { long src; int ld, rows, cols, transpose, nsteps, step0, pad_; };
#define FUSED_MAX_SEGS 64
struct StreamPlan { Seg seg[FUSED_MAX_SEGS]; int nseg, steps; };

void plan_add(StreamPlan &p, long src, int ld, int rows, int cols, int tr, int nsteps) {
  Seg s = {src, ld, rows, cols, tr, nsteps, p.steps, 0};
  p.seg[p.nseg++] = s;
  p.steps += nsteps;
}

// forward: per layer [s = Wl_scale lat] [z = W x (i > 0)] [t = Wl_shift lat]; nodes path: [z = W x] only
void plan_fwd_stream(const cinema_siren_desc *d, StreamPlan &p, bool nodes = false) {
  const int H = d->hidden, L = d->lat;
  p.nseg = 0; p.steps = 0;
  for (int i = 0; i < d->n_layers; ++i) {
    const bool mod = !nodes && ((d->modulated_mask >> i) & 1u);
    if (mod) plan_add(p, d->off_wl[i], L, H, L, 0, ksteps(d->LP));
    if (i > 0) plan_add(p, d->off_w[i], H, H, H, 0, ksteps(d->HP));
    if (mod) plan_add(p, d->off_wl[i] + (long)H * L, L, H, L, 0, ksteps(d->LP));
  }
}

// backward chain: dy_i = W_{i+1}^T dz_{i+1}, i = n_layers-2 .. 0
void plan_bwd_stream(const cinema_siren_desc *d, StreamPlan &p) {
  const int H = d->hidden;
  p.nseg = 0; p.steps = 0;
  for (int i = d->n_layers - 2; i >= 0; --i) plan_add(p, d->off_w[i + 1], H, H, H, 1, ksteps(d->HP));
}

// two streams (forward, backward chain) packed by one launch: segments [0, nseg0) belong to stream 0
struct PackStreamArgs { Seg seg[FUSED_MAX_SEGS]; int nseg0, nseg1, steps0, steps1, nslab; const float *params; bf16_t *dst0, *dst1;
                        int pairmap; };  // bit 0 / bit 1: stream 0 / 1 in the pair-major feature map of fused_nodes.hip; bit 2: stream 1 in fused_wide.hip's map
                                         // (a lane's fragments 2p, 2p+1 hold 8 consecutive features)

// (vbid, vgrid): the workgroup's index and count among the workgroups doing this job -- the kernels below run either
// as launches of their own or as block ranges of k_step_head
__device__ __forceinline__ void pack_stream_body(const PackStreamArgs &a, unsigned vbid, unsigned vgrid) {
  // 32-bit indices (the host checks the range)
  const unsigned n0 = (unsigned)a.nslab * a.steps0 * 256u, nchunk = n0 + (unsigned)a.nslab * a.steps1 * 256u;
  for (unsigned qq = vbid * 256u + threadIdx.x; qq < nchunk; qq += vgrid * 256u) {
    const bool second = qq >= n0;
    const unsigned q = second ? qq - n0 : qq;
    const int steps = second ? a.steps1 : a.steps0, sbeg = second ? a.nseg0 : 0, send = second ? a.nseg0 + a.nseg1 : a.nseg0;
    const int lane = (int)(q & 63), ft = (int)((q >> 6) & 3);
    const unsigned sq = q >> 8;
    const int slab = (int)(sq / (unsigned)steps), step = (int)(sq - (unsigned)slab * (unsigned)steps);
    int si = sbeg;
    while (si + 1 < send && step >= a.seg[si + 1].step0) ++si;
    const Seg sg = a.seg[si];
    const int kk = step - sg.step0, i = lane & 15, gk = lane >> 4;
    const int feat = (second && (a.pairmap & 4)) ? 128 * (slab >> 1) + 64 * (ft >> 1) + 32 * (slab & 1) + 8 * (i >> 2) + 4 * (ft & 1) + (i & 3)
                     : ((a.pairmap >> (second ? 1 : 0)) & 1) ? slab * 64 + 32 * (ft >> 1) + 8 * (i >> 2) + 4 * (ft & 1) + (i & 3)
                               : slab * 64 + 16 * (i >> 2) + 4 * ft + (i & 3);
    const int k0 = 32 * kk + 8 * gk;
    const float *src = a.params + sg.src;
    bf16x8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = k0 + j;
      float x = 0.f;
      if (feat < sg.rows && k < sg.cols) x = sg.transpose ? src[(long)k * sg.ld + feat] : src[(long)feat * sg.ld + k];
      v[j] = (bf16_t)x;
    }
    *reinterpret_cast<bf16x8 *>((second ? a.dst1 : a.dst0) + (size_t)q * 8) = v;
  }
}
__global__ __launch_bounds__(256) void k_pack_stream(PackStreamArgs a) { pack_stream_body(a, blockIdx.x, gridDim.x); }

// fp32 side block in exactly the layout the forward kernel keeps in LDS (zero padded to HP per vector):
//   per layer [b_l] [bl_scale, bl_shift if modulated] ... then w0 as [HP][4] (k = 0..2, pad), then (backward only)
//   final_linear.weight[:, :H] as [16][HP]
struct PackSideArgs {
  const float *params; float *dst;
  long off_b[CINEMA_MAX_LAYERS], off_bl[CINEMA_MAX_LAYERS], off_w0;
  int lb_b[CINEMA_MAX_LAYERS], lb_bl[CINEMA_MAX_LAYERS], lb_total;
  unsigned mod_mask; int n_layers, H, HP, in_dim;
  long off_wf; int out_dim, lat;
  bf16_t *dst_wf; int LP;      // final_linear padded to [16][HP + LP] bf16 (read by the forward's last layer)
};

__device__ __forceinline__ void pack_side_body(const PackSideArgs &a, unsigned vbid, unsigned vgrid) {
  const int total = a.lb_total + 4 * a.HP + 16 * a.HP;
  const int total_wf = 16 * (a.HP + a.LP);
  for (int e = vbid * 256u + threadIdx.x; e < total + total_wf; e += vgrid * 256u) {
    float x = 0.f;
    if (e >= total) {
      const int q = e - total, o = q / (a.HP + a.LP), col = q - o * (a.HP + a.LP);
      if (o < a.out_dim) {
        if (col < a.HP) { if (col < a.H) x = a.params[a.off_wf + (long)o * (a.H + a.lat) + col]; }
        else if (col - a.HP < a.lat) x = a.params[a.off_wf + (long)o * (a.H + a.lat) + a.H + (col - a.HP)];
      }
      a.dst_wf[q] = (bf16_t)x;
      continue;
    }
    if (e >= a.lb_total + 4 * a.HP) {          // final_linear.weight[:, :H] as [16][HP] (backward, top of the chain)
      const int q = e - a.lb_total - 4 * a.HP, o = q / a.HP, f = q - o * a.HP;
      if (o < a.out_dim && f < a.H) x = a.params[a.off_wf + (long)o * (a.H + a.lat) + f];
    } else if (e >= a.lb_total) {
      const int q = e - a.lb_total, f = q >> 2, k = q & 3;
      if (f < a.H && k < a.in_dim) x = a.params[a.off_w0 + (long)f * a.in_dim + k];
    } else {
      int l = 0;
      while (l + 1 < a.n_layers && e >= a.lb_b[l + 1]) ++l;
      const int q = e - a.lb_b[l];
      const int part = q / a.HP, f = q - part * a.HP;      // 0: b, 1: bl scale, 2: bl shift
      if (f < a.H) x = part == 0 ? a.params[a.off_b[l] + f] : a.params[a.off_bl[l] + (long)(part - 1) * a.H + f];
    }
    a.dst[e] = x;
  }
}
__global__ __launch_bounds__(256) void k_pack_side(PackSideArgs a) { pack_side_body(a, blockIdx.x, gridDim.x); }

// The head of a training step in ONE launch.  The grouped-batch K1 forward, the repacking of the weights, the node-major
// shadow of the latent table and the per-step clearing are independent, but as separate kernels they either share a
// stream (23 + 17 us in a row for the two the forward waits for) or need a second stream, whose join costs 6-10 us of
// idle queue.  Block ranges: step_prepare (the longest chain of dependent loads; optional, g_pr = 0) | K1 forward chunks |
// backward-only transposes (64 workgroups each) | weight streams | side block | node rows and interpolation weights of the
// node-space backward (optional, g_nr = g_ngw = 0) | shadow tiles (optional, g_nm = 0).
#define HEAD_MAX_JOBS 16
struct StepHeadArgs {
  PackStreamArgs ps; PackSideArgs sd; TriFwdSortedArgs k1;
  PackJob job[HEAD_MAX_JOBS]; void *packed;
  const float *nm_lat; float *nm_shadow; long nm_S; int nm_C, nm_G;
  StepPrepareArgs pr; NodeRowsCArgs nr; NgWeightArgs ngw;
  int g_k1, g_ps, g_sd, g_job, g_nm, g_pr, g_nr, g_ngw;
  int skip;      // developer timing (CINEMA_HEAD_SKIP): block ranges switched off, results are wrong then
};
__global__ __launch_bounds__(256) void k_step_head(StepHeadArgs a) {
  // ONE LDS pool for whichever body this workgroup runs (the K1 forward's 32 000 B decide): as static arrays of the bodies
  // they added up to 41.5 KB = three workgroups per CU for 4 100 workgroups; now five
  __shared__ __attribute__((aligned(16))) char pool[TFS_LDS_BYTES(27)];
  static_assert(TFS_LDS_BYTES(27) >= NODE_MAJOR_LDS_BYTES && TFS_LDS_BYTES(27) >= NODE_ROWS_LDS_BYTES &&
                TFS_LDS_BYTES(27) >= 16 * NG_WB * 16 && TFS_LDS_BYTES(27) >= STEP_PREPARE_LDS_BYTES, "head LDS pool");
  int b = blockIdx.x;                            // workgroup-uniform dispatch; the longest dependent-load chains first
  if (b < a.g_pr) { if (!(a.skip & 1)) step_prepare_body(a.pr, b, a.g_pr, pool); return; }
  b -= a.g_pr;
  if (b < a.g_k1) { if (!(a.skip & 2)) trilerp_fwd_sorted_body<bf16_t, 3, 3, 3>(a.k1, b, pool); return; }
  b -= a.g_k1;
  if (b < a.g_ngw) {                             // dependent loads (idx -> embedding row) like K1: early
    if (!(a.skip & 4)) ng_weights_body<256>(a.ngw, b, reinterpret_cast<float (*)[4]>(pool));
    return;
  }
  b -= a.g_ngw;
  if (b < a.g_job) { if (!(a.skip & 8)) pack_job_body<bf16_t>(a.job[b >> 6], a.packed, b & 63, 64); return; }
  b -= a.g_job;
  if (b < a.g_ps) { if (!(a.skip & 16)) pack_stream_body(a.ps, b, a.g_ps); return; }
  b -= a.g_ps;
  if (b < a.g_sd) { if (!(a.skip & 32)) pack_side_body(a.sd, b, a.g_sd); return; }
  b -= a.g_sd;
  if (b < a.g_nr) { if (!(a.skip & 64)) node_rows_centred_body(a.nr, b, pool); return; }
  b -= a.g_nr;
  node_major_body(a.nm_lat, a.nm_S, a.nm_C, a.nm_G, a.nm_shadow, b, pool);
}

#include "fused_dev.h"

// optional phase trace of one wave (tools/trace_fused.py): 100 MHz timestamps
__device__ long long g_fused_trace[128];
#ifdef FUSED_TRACE_BUILD   // perturbs register allocation: only in the libcinema_hip_trace.so developer build
#define FUSED_TRACE(ID) \
  if (a.trace_block == (int)blockIdx.x && (int)threadIdx.x == a.trace_wave * 64) g_fused_trace[ID] = (long long)wall_clock64();
#else
#define FUSED_TRACE(ID)
#endif

// ----------------------------------------------------------------------------------------
// fused forward
// ----------------------------------------------------------------------------------------
struct FusedFwdArgs {
  int g4;                            // workgroups [0, g4) own 64 samples, the rest 48 (fused_launch_plan)
  const bf16_t *stream; long slab_elems;
  const float *params;
  long off_b[CINEMA_MAX_LAYERS], off_bl[CINEMA_MAX_LAYERS], off_w0;
  unsigned mod_mask; int n_layers, in_dim, H, HP, LP, out_dim;
  float om_first, om_hidden;
  const float *coords_t; const bf16_t *lat;
  bf16_t *saved; long slot;
  int sv_y[CINEMA_MAX_LAYERS], sv_c[CINEMA_MAX_LAYERS], sv_z[CINEMA_MAX_LAYERS], sv_s[CINEMA_MAX_LAYERS];
  const bf16_t *wf; const float *bf; float *out;
  int xrow, lrow, xsteps, lsteps;
  int lb_b[CINEMA_MAX_LAYERS], lb_bl[CINEMA_MAX_LAYERS], lb_total;   // LDS side block: float offsets
  const float *side;
  int trace_block, trace_wave;
  NgProjArgs proj; int g_chain;      // tail workgroups [g_chain, grid): node projections of the node-space backward
};

// NST = sample tiles of 16 per workgroup: 4 (64 samples) or 3 (48: the second, shorter round of a batch that fills the
// chip between one and two times -- its workgroups pay the same weight stream but three quarters of the epilogues)
template <int NST, bool FP8>
__device__ __forceinline__ void fused_fwd_body(const FusedFwdArgs &a, const long n0) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  lds_char *xs = (lds_char *)smem;
  lds_char *ls = xs + FUSED_TS * a.xrow;
  lds_char *bs = ls + FUSED_TS * a.lrow;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform by construction: keep it scalar
  const int i = lane & 15, g = lane >> 4, gk = g;
  const int slab = wave;
  const bool active = slab * 64 < a.HP;
  const unsigned fb = slab * 64 + 16 * g;
  const unsigned lane16 = lane * 16;
  // byte offset of (sample i of sample tile 0, feature fb) inside a [64 x HP] bf16 tile; tile st adds st*16*HP*2
  const unsigned toff = (i * a.HP + fb) * 2u;
  const unsigned tstep = 16u * a.HP * 2u;
  const bool keep = a.saved != nullptr;   // NULL: inference, nothing is kept for a backward

  FUSED_TRACE(0)
  WRing r;
  if (active) ring_init(r, (const char *)(a.stream + (long)slab * a.slab_elems), lane16);
  lds_char *cs = bs + (a.lb_total + 4 * a.HP) * 4;      // registered coordinates of the tile, [64][4] fp32
  {
    // latent rows -> LDS (swizzled 16-B chunks); the k-padding chunks meet zero weights and must not hold NaN patterns
    const int cpr = a.lrow >> 4, cdat = a.LP >> 3;
    const char *lt = (const char *)(a.lat + n0 * a.LP);
    const u32x4 z4 = {0u, 0u, 0u, 0u};
    for (int q = tid; q < NST * 16 * cpr; q += 512) {
      const int row = q / cpr, c = q - row * cpr;
      u32x4 v = z4;
      if (c < cdat) v = *reinterpret_cast<const u32x4 *>(lt + (unsigned)(row * a.LP + c * 8) * 2u);
      *(LDS_AS u32x4 *)(ls + row * a.lrow + ((c ^ (row & 15)) << 4)) = v;
    }
    if (a.xrow != a.HP * 2)
      for (int q = tid; q < FUSED_TS * a.xrow / 16; q += 512) *(LDS_AS u32x4 *)(xs + q * 16) = z4;
    // biases + first-layer weights: one coalesced copy of the block k_pack_side prepared
    const int nb4 = (a.lb_total + 4 * a.HP) / 4;
    for (int q = tid; q < nb4; q += 512) *(LDS_AS f32x4 *)(bs + q * 16) = *reinterpret_cast<const f32x4 *>(a.side + q * 4);
    if (tid < NST * 16) {
      const float *cp = a.coords_t + (n0 + tid) * a.in_dim;
      f32x4 c = {cp[0], a.in_dim > 1 ? cp[1] : 0.f, a.in_dim > 2 ? cp[2] : 0.f, 0.f};
      *(LDS_AS f32x4 *)(cs + tid * 16) = c;
    }
  }
  __syncthreads();
  FUSED_TRACE(1)

  f32x4 acc[4][NST];
#pragma nounroll
  for (int l = 0; l < a.n_layers; ++l) {
    const bool mod = (a.mod_mask >> l) & 1u;
    const float omega = l == 0 ? a.om_first : a.om_hidden;
    // re-derive the per-lane offsets from opaque copies in every iteration: otherwise LICM hoists ~100 registers of
    // address arithmetic out of the layer loop and spills them
    unsigned toff_l = toff, fb_l = fb;
    int i_l = i;
    asm volatile("" : "+v"(toff_l), "+v"(fb_l), "+v"(i_l));
    if (active) {
      f32x4 v[4];
      // modulated layer: scale branch first; s is stored (the backward needs it) and kept as packed bf16 --
      // the phase below uses exactly the value the backward will read
      bf16x8 sp[NST][2];
      if (mod) {
        feature_vec(bs, a.lb_bl[l], fb_l, v);
        fill_acc(acc, v);
        run_seg(acc, r, ls, a.lrow, a.lsteps, i_l, gk, lane16);
        FUSED_TRACE(2 + 8 * l)
        char *S = (char *)(a.saved + a.slot * a.sv_s[l] + n0 * a.HP);
#pragma unroll
        for (int st = 0; st < NST; ++st) {
          float sv[16];
#pragma unroll
          for (int j = 0; j < 16; ++j) sv[j] = acc[j >> 2][st][j & 3];
          pack16(sv, sp[st][0], sp[st][1]);
          if (keep) {
            if constexpr (FP8) {
              *reinterpret_cast<u32x4 *>(S + ((toff_l + st * tstep) >> 1)) = pack16_fp8(sv, FP8_SCALE_ZS);
            } else {
              *reinterpret_cast<bf16x8 *>(S + (toff_l + st * tstep)) = sp[st][0];
              *reinterpret_cast<bf16x8 *>(S + (toff_l + st * tstep + 16u)) = sp[st][1];
            }
          }
        }
      }
      feature_vec(bs, a.lb_b[l], fb_l, v);
      fill_acc(acc, v);
      if (l == 0) {
        // first layer: K = in_dim <= 3 on the VALU in fp32, operands from LDS
        f32x4 c[NST];
#pragma unroll
        for (int st = 0; st < NST; ++st) c[st] = *(const LDS_AS f32x4 *)(cs + (16 * st + i_l) * 16);
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          const f32x4 w = *(const LDS_AS f32x4 *)(bs + (a.lb_total + (fb_l + j) * 4) * 4);
#pragma unroll
          for (int st = 0; st < NST; ++st) acc[j >> 2][st][j & 3] += w[0] * c[st][0] + w[1] * c[st][1] + w[2] * c[st][2];
        }
      } else {
        run_seg(acc, r, xs, a.xrow, a.xsteps, i_l, gk, lane16);
      }
      FUSED_TRACE(3 + 8 * l)
      if (mod) {
        f32x4 bt[4];
        feature_vec(bs, a.lb_bl[l] + a.HP, fb_l, bt);
        char *Z = (char *)(a.saved + a.slot * a.sv_z[l] + n0 * a.HP);
#pragma unroll
        for (int st = 0; st < NST; ++st) {
          float zv[16];
#pragma unroll
          for (int j = 0; j < 16; ++j) zv[j] = acc[j >> 2][st][j & 3];
          bf16x8 zlo, zhi;
          pack16(zv, zlo, zhi);
          if (keep) {
            if constexpr (FP8) {
              *reinterpret_cast<u32x4 *>(Z + ((toff_l + st * tstep) >> 1)) = pack16_fp8(zv, FP8_SCALE_ZS);
            } else {
              *reinterpret_cast<bf16x8 *>(Z + (toff_l + st * tstep)) = zlo;
              *reinterpret_cast<bf16x8 *>(Z + (toff_l + st * tstep + 16u)) = zhi;
            }
          }
#pragma unroll
          for (int j = 0; j < 16; ++j) {
            const float sj = (float)(j < 8 ? sp[st][0][j & 7] : sp[st][1][j & 7]);
            acc[j >> 2][st][j & 3] = omega * zv[j] * sj + bt[j >> 2][j & 3];
          }
        }
        run_seg(acc, r, ls, a.lrow, a.lsteps, i_l, gk, lane16);
      }
      FUSED_TRACE(4 + 8 * l)
    }
    __syncthreads();  // every wave has finished reading x
    FUSED_TRACE(5 + 8 * l)
    if (active) {
      char *Y = (char *)(a.saved + a.slot * a.sv_y[l] + n0 * a.HP);
      char *C = (char *)(a.saved + a.slot * a.sv_c[l] + n0 * a.HP);
      const int c0 = fb_l >> 3;
#pragma unroll
      for (int st = 0; st < NST; ++st) {
        float yv[16], cv[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          float ph = acc[j >> 2][st][j & 3];
          if (!mod) ph *= omega;
          yv[j] = __sinf(ph);
          cv[j] = __cosf(ph);
        }
        bf16x8 ylo, yhi, clo, chi;
        pack16(yv, ylo, yhi);
        pack16(cv, clo, chi);
        if (keep) {
          *reinterpret_cast<bf16x8 *>(Y + (toff_l + st * tstep)) = ylo;
          *reinterpret_cast<bf16x8 *>(Y + (toff_l + st * tstep + 16u)) = yhi;
          if constexpr (FP8) {
            *reinterpret_cast<u32x4 *>(C + ((toff_l + st * tstep) >> 1)) = pack16_fp8(cv, FP8_SCALE_C);
          } else {
            *reinterpret_cast<bf16x8 *>(C + (toff_l + st * tstep)) = clo;
            *reinterpret_cast<bf16x8 *>(C + (toff_l + st * tstep + 16u)) = chi;
          }
        }
        lds_char *xr = xs + (16 * st + i_l) * a.xrow;
        *(LDS_AS bf16x8 *)(xr + ((c0 ^ i_l) << 4)) = ylo;
        *(LDS_AS bf16x8 *)(xr + (((c0 + 1) ^ i_l) << 4)) = yhi;
      }
    }
    FUSED_TRACE(6 + 8 * l)
    __syncthreads();  // x of the next layer is complete
    FUSED_TRACE(7 + 8 * l)
  }

  // final linear (siren.py:61): out[n][o] = bf[o] + sum_k [x_L | lat][n][k] * Wf[o][k]; wave w < 4 takes sample tile w
  if (wave < NST) {
    const int st = wave, KT = a.HP + a.LP;
    f32x4 o4 = {0.f, 0.f, 0.f, 0.f};
    const char *wrow = (const char *)a.wf;
    const unsigned woff = (i * KT + 8 * gk) * 2u;
    const lds_char *xrp = xs + (16 * st + i) * a.xrow;
    const lds_char *lrp = ls + (16 * st + i) * a.lrow;
    // all A fragments of a group are fetched before its first MFMA (26 dependent L2 round trips otherwise)
    const int nx = a.HP >> 5, nsteps = nx + (a.LP >> 5);
    for (int k0 = 0; k0 < nsteps; k0 += 8) {
      bf16x8 w[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int kk = k0 + k < nsteps ? k0 + k : nsteps - 1;
        w[k] = *reinterpret_cast<const bf16x8 *>(wrow + (woff + 64u * kk));     // Wf row is [x part | lat part], k contiguous
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int kk = k0 + k;
        if (kk < nsteps) {
          const lds_char *src = kk < nx ? xrp + (((((kk) << 2) + gk) ^ i) << 4) : lrp + (((((kk - nx) << 2) + gk) ^ i) << 4);
          const bf16x8 b = *(const LDS_AS bf16x8 *)src;
          o4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[k], b, o4, 0, 0, 0);
        }
      }
    }
    float *ot = a.out + (n0 + 16 * st) * a.out_dim;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int o = 4 * g + rr;
      if (o < a.out_dim) ot[(unsigned)(i * a.out_dim + o)] = o4[rr] + a.bf[o];
    }
  }
  FUSED_TRACE(127)
}

template <bool FP8>
__global__ __launch_bounds__(512) void k_fused_fwd(FusedFwdArgs a) {
  if ((int)blockIdx.x >= a.g_chain) {
    // behind the chain workgroups: U' / V' of the backward's node reductions (they read the packed weights and the centred
    // node rows of the step head only; these workgroups land on the CUs the shorter second round leaves idle)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    ng_project_body(a.proj, (int)blockIdx.x - a.g_chain, (lds_char *)smem);
    return;
  }
  if ((int)blockIdx.x < a.g4) fused_fwd_body<4, FP8>(a, 64L * blockIdx.x);
  else fused_fwd_body<3, FP8>(a, 64L * a.g4 + 48L * ((int)blockIdx.x - a.g4));
}


// ----------------------------------------------------------------------------------------
// fused backward chain: the mirror of k_fused_fwd.  The gradient tile g = dz_{l+1} [64 x HP] stays in LDS from the
// top of the stack to the first layer; per layer dy = W_{l+1}^T g streams the transposed weights, the epilogue applies
// the sine / FiLM derivatives with the saved c (z, s), streams dz (ds, dt) out for the weight-gradient and d-latent
// GEMMs, reduces the bias gradients and leaves dz in LDS as the next g.
// ----------------------------------------------------------------------------------------
struct FusedBwdArgs {
  FinalBwdArgs fin; bf16_t *dout16; int g_chain, g_fin;   // tail workgroups: dOut -> [Npad x 16] bf16 + final bias gradient
  int g4;                            // workgroups [0, g4) own 64 samples, the rest 48 (fused_launch_plan)
  const bf16_t *stream; long slab_elems;
  const float *side_w0;              // [HP][4] first-layer weights, then [16][HP] final_linear.weight[:, :H]
  const float *dout; const float *coords_t;
  const bf16_t *saved; long slot; int sv_c[CINEMA_MAX_LAYERS], sv_z[CINEMA_MAX_LAYERS], sv_s[CINEMA_MAX_LAYERS];
  bf16_t *gbuf; int gb_z[CINEMA_MAX_LAYERS], gb_s[CINEMA_MAX_LAYERS], gb_t[CINEMA_MAX_LAYERS];
  float *gp; long off_b[CINEMA_MAX_LAYERS], off_bl[CINEMA_MAX_LAYERS], off_w0; int wgrads;
  int latb;                          // wgrads and the linear_lats bias gradients are wanted from this kernel
  float *dcoords;
  unsigned mod_mask; int n_layers, in_dim, H, HP, out_dim, xrow, xsteps;
  float om_first, om_hidden;
};

template <bool MOD, bool FIRST, int NST, bool FP8, bool LATB>
__device__ __forceinline__ void bwd_epilogue(const FusedBwdArgs &a, f32x4 (&acc)[4][NST], int l, float omega, long n0,
                                             unsigned toff_l, unsigned tstep, unsigned fb_l, int i_l, int g, int lane,
                                             lds_char *gs, const lds_char *w0s, const lds_char *cs) {
      const char *C = (const char *)(a.saved + a.slot * a.sv_c[l] + n0 * a.HP);
      const char *Z = MOD ? (const char *)(a.saved + a.slot * a.sv_z[l] + n0 * a.HP) : C;
      const char *S = MOD ? (const char *)(a.saved + a.slot * a.sv_s[l] + n0 * a.HP) : C;
      char *DZ = (char *)(a.gbuf + a.slot * a.gb_z[l] + n0 * a.HP);
      char *DS = MOD ? (char *)(a.gbuf + a.slot * a.gb_s[l] + n0 * a.HP) : DZ;
      char *DT = MOD ? (char *)(a.gbuf + a.slot * a.gb_t[l] + n0 * a.HP) : DZ;
      const int c0 = fb_l >> 3;
      // column sums (bias / first-layer weight gradients): each sample tile is transposed-reduced over its 16 lanes
      // right away, so only one scalar per tensor stays live across the tiles instead of 16
      float tz = 0.f, ts = 0.f, tt = 0.f, tw[3] = {0.f, 0.f, 0.f};
#pragma unroll
      for (int st = 0; st < NST; ++st) {
        const unsigned o = toff_l + st * tstep;
        float cv[16], dz[16];
        // fp8 copies: the scales are folded into the constants below (c_inv: c, om_zs: omega for the z / s products)
        constexpr float c_inv = FP8 ? 1.f / FP8_SCALE_C : 1.f;
        const float om_zs = FP8 ? omega * (1.f / FP8_SCALE_ZS) : omega;
        if constexpr (FP8) {
          unpack16_fp8(*reinterpret_cast<const u32x4 *>(C + (o >> 1)), cv);
        } else {
          const bf16x8 lo = *reinterpret_cast<const bf16x8 *>(C + o), hi = *reinterpret_cast<const bf16x8 *>(C + o + 16u);
#pragma unroll
          for (int j = 0; j < 8; ++j) { cv[j] = (float)lo[j]; cv[8 + j] = (float)hi[j]; }
        }
        if constexpr (MOD) {
          float ds[16], dt[16];
          if constexpr (FP8) {
            // the packed words stay packed until their four values are used: three decoded tensors would hold 48 registers
            const u32x4 zp = *reinterpret_cast<const u32x4 *>(Z + (o >> 1)), sp = *reinterpret_cast<const u32x4 *>(S + (o >> 1));
            typedef float f32x2_t __attribute__((ext_vector_type(2)));
#pragma unroll
            for (int h = 0; h < 8; ++h) {
              const f32x2_t z2 = (h & 1) ? __builtin_amdgcn_cvt_pk_f32_fp8((int)zp[h >> 1], true)
                                         : __builtin_amdgcn_cvt_pk_f32_fp8((int)zp[h >> 1], false);
              const f32x2_t s2 = (h & 1) ? __builtin_amdgcn_cvt_pk_f32_fp8((int)sp[h >> 1], true)
                                         : __builtin_amdgcn_cvt_pk_f32_fp8((int)sp[h >> 1], false);
#pragma unroll
              for (int e = 0; e < 2; ++e) {
                const int j = 2 * h + e;
                const float gq = acc[j >> 2][st][j & 3] * cv[j] * c_inv;
                dt[j] = gq;
                dz[j] = om_zs * gq * s2[e];
                ds[j] = om_zs * gq * z2[e];
              }
            }
          } else {
            const bf16x8 zlo = *reinterpret_cast<const bf16x8 *>(Z + o), zhi = *reinterpret_cast<const bf16x8 *>(Z + o + 16u);
            const bf16x8 slo = *reinterpret_cast<const bf16x8 *>(S + o), shi = *reinterpret_cast<const bf16x8 *>(S + o + 16u);
#pragma unroll
            for (int j = 0; j < 16; ++j) {
              const float zj = (float)(j < 8 ? zlo[j & 7] : zhi[j & 7]), sj = (float)(j < 8 ? slo[j & 7] : shi[j & 7]);
              const float gq = acc[j >> 2][st][j & 3] * cv[j];
              dt[j] = gq;
              dz[j] = omega * gq * sj;
              ds[j] = omega * gq * zj;
            }
          }
          if constexpr (LATB) {      // compiled out when the node-space reductions take the linear_lats bias sums
            ts += reduce16_scatter(ds, lane);
            tt += reduce16_scatter(dt, lane);
          }
          bf16x8 lo, hi;
          pack16(ds, lo, hi);
          *reinterpret_cast<bf16x8 *>(DS + o) = lo;
          *reinterpret_cast<bf16x8 *>(DS + o + 16u) = hi;
          pack16(dt, lo, hi);
          *reinterpret_cast<bf16x8 *>(DT + o) = lo;
          *reinterpret_cast<bf16x8 *>(DT + o + 16u) = hi;
        } else {
#pragma unroll
          for (int j = 0; j < 16; ++j) dz[j] = (omega * c_inv) * acc[j >> 2][st][j & 3] * cv[j];
        }
        bf16x8 zl, zh;
        pack16(dz, zl, zh);
        *reinterpret_cast<bf16x8 *>(DZ + o) = zl;
        *reinterpret_cast<bf16x8 *>(DZ + o + 16u) = zh;
        if constexpr (!FIRST) {
          lds_char *xr = gs + (16 * st + i_l) * a.xrow;
          *(LDS_AS bf16x8 *)(xr + ((c0 ^ i_l) << 4)) = zl;
          *(LDS_AS bf16x8 *)(xr + (((c0 + 1) ^ i_l) << 4)) = zh;
        } else {
          // first layer: d coords = dz W0, dW0 = dz^T coords
          const f32x4 xc = *(const LDS_AS f32x4 *)(cs + (16 * st + i_l) * 16);
          float dc[3] = {0.f, 0.f, 0.f};
#pragma unroll
          for (int j = 0; j < 16; ++j) {
            const f32x4 w = *(const LDS_AS f32x4 *)(w0s + (fb_l + j) * 16);
#pragma unroll
            for (int k = 0; k < 3; ++k) dc[k] += dz[j] * w[k];
          }
          if (a.wgrads) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
              float p[16];
#pragma unroll
              for (int j = 0; j < 16; ++j) p[j] = dz[j] * xc[k];
              tw[k] += reduce16_scatter(p, lane);
            }
          }
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            dc[k] += __shfl_xor(dc[k], 16, 64);
            dc[k] += __shfl_xor(dc[k], 32, 64);
          }
          if (g == 0) {
#pragma unroll
            for (int k = 0; k < 3; ++k)
              if (k < a.in_dim) atomicAdd(a.dcoords + (n0 + 16 * st + i_l) * a.in_dim + k, dc[k]);
          }
        }
        if (a.wgrads) tz += reduce16_scatter(dz, lane);
        __builtin_amdgcn_sched_barrier(0);   // one sample tile at a time: bounds the live temporaries
      }
      // column sums over the 64 samples of this wave -> bias (and first-layer weight) gradients
      if (a.wgrads) {
        const int fi = fb_l + reduce16_index(i_l);
        if (fi < a.H) atomicAdd(a.gp + a.off_b[l] + fi, tz);
        if constexpr (MOD) {
          if (fi < a.H && LATB) {
            atomicAdd(a.gp + a.off_bl[l] + fi, ts);
            atomicAdd(a.gp + a.off_bl[l] + a.H + fi, tt);
          }
        }
        if constexpr (FIRST) {
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            if (k < a.in_dim && fi < a.H) atomicAdd(a.gp + a.off_w0 + (long)fi * a.in_dim + k, tw[k]);
          }
        }
      }
}

template <int NST, bool FP8, bool LATB>
__device__ __forceinline__ void fused_bwd_body(const FusedBwdArgs &a, const long n0) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  lds_char *gs = (lds_char *)smem;                       // g tile [64][xrow] bf16, swizzled
  lds_char *w0s = gs + FUSED_TS * a.xrow;                // [HP][4] fp32
  lds_char *wfs = w0s + a.HP * 16;                       // [16][HP] fp32
  lds_char *ds_ = wfs + 16 * a.HP * 4;                   // dOut tile [64][16] fp32
  lds_char *cs = ds_ + FUSED_TS * 64;                    // coords tile [64][4] fp32
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 15, g = lane >> 4, gk = g;
  const int slab = wave;
  const bool active = slab * 64 < a.HP;
  const unsigned fb = slab * 64 + 16 * g;
  const unsigned lane16 = lane * 16;
  const unsigned toff = (i * a.HP + fb) * 2u;
  const unsigned tstep = 16u * a.HP * 2u;

  WRing r;
  if (active && a.n_layers > 1) ring_init(r, (const char *)(a.stream + (long)slab * a.slab_elems), lane16);
  {
    const int nb4 = 20 * a.HP / 4;
    for (int q = tid; q < nb4; q += 512) *(LDS_AS f32x4 *)(w0s + q * 16) = *reinterpret_cast<const f32x4 *>(a.side_w0 + q * 4);
    for (int q = tid; q < NST * 16 * 16; q += 512) {
      const int row = q >> 4, o = q & 15;
      ((LDS_AS float *)ds_)[q] = o < a.out_dim ? a.dout[(n0 + row) * a.out_dim + o] : 0.f;
    }
    if (tid < NST * 16) {
      const float *cp = a.coords_t + (n0 + tid) * a.in_dim;
      f32x4 c = {cp[0], a.in_dim > 1 ? cp[1] : 0.f, a.in_dim > 2 ? cp[2] : 0.f, 0.f};
      *(LDS_AS f32x4 *)(cs + tid * 16) = c;
    }
    if (a.xrow != a.HP * 2) {
      const u32x4 z4 = {0u, 0u, 0u, 0u};
      for (int q = tid; q < FUSED_TS * a.xrow / 16; q += 512) *(LDS_AS u32x4 *)(gs + q * 16) = z4;
    }
  }
  __syncthreads();

  f32x4 acc[4][NST];
#pragma nounroll
  for (int l = a.n_layers - 1; l >= 0; --l) {
    const bool mod = (a.mod_mask >> l) & 1u;
    const float omega = l == 0 ? a.om_first : a.om_hidden;
    unsigned toff_l = toff, fb_l = fb;
    int i_l = i;
    asm volatile("" : "+v"(toff_l), "+v"(fb_l), "+v"(i_l));   // see k_fused_fwd
    if (active) {
      const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ft = 0; ft < 4; ++ft)
#pragma unroll
        for (int st = 0; st < NST; ++st) acc[ft][st] = zero4;
      if (l == a.n_layers - 1) {
        // top of the chain: dy[n][f] = sum_o dOut[n][o] * Wf[o][f] on the VALU
        for (int o = 0; o < a.out_dim; ++o) {
          float d[NST];
#pragma unroll
          for (int st = 0; st < NST; ++st) d[st] = ((const LDS_AS float *)ds_)[(16 * st + i_l) * 16 + o];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x4 w = *(const LDS_AS f32x4 *)(wfs + ((o * a.HP + fb_l) * 4u + q * 16u));
#pragma unroll
            for (int st = 0; st < NST; ++st) acc[q][st] += w * d[st];
          }
        }
      } else {
        run_seg(acc, r, gs, a.xrow, a.xsteps, i_l, gk, lane16);
      }
    }
    __syncthreads();  // every wave has finished reading g
    if (active) {
      if (l == 0) {
        if (mod) bwd_epilogue<true, true, NST, FP8, LATB>(a, acc, l, omega, n0, toff_l, tstep, fb_l, i_l, g, lane, gs, w0s, cs);
        else bwd_epilogue<false, true, NST, FP8, LATB>(a, acc, l, omega, n0, toff_l, tstep, fb_l, i_l, g, lane, gs, w0s, cs);
      } else {
        if (mod) bwd_epilogue<true, false, NST, FP8, LATB>(a, acc, l, omega, n0, toff_l, tstep, fb_l, i_l, g, lane, gs, w0s, cs);
        else bwd_epilogue<false, false, NST, FP8, LATB>(a, acc, l, omega, n0, toff_l, tstep, fb_l, i_l, g, lane, gs, w0s, cs);
      }
    }
    __syncthreads();  // g of the next (lower) layer is complete
  }
}

template <bool FP8, bool LATB>
__global__ __launch_bounds__(512) void k_fused_bwd(FusedBwdArgs a) {
  // behind the chain workgroups: the dOut preparation of the weight-gradient launch (reads dOut only; these workgroups
  // land on the CUs the shorter second round leaves idle)
  if ((int)blockIdx.x >= a.g_chain) { dout_prep_body<bf16_t>(a.fin, a.dout16, blockIdx.x - a.g_chain, a.g_fin); return; }
  if ((int)blockIdx.x < a.g4) fused_bwd_body<4, FP8, LATB>(a, 64L * blockIdx.x);
  else fused_bwd_body<3, FP8, LATB>(a, 64L * a.g4 + 48L * ((int)blockIdx.x - a.g4));
}

}  // namespace

// ========================================================================================
// host side
// ========================================================================================
bool fused_saved_fp8();
static int fused_bias_floats(const cinema_siren_desc *d) {
  int n = 0;
  for (int i = 0; i < d->n_layers; ++i) n += d->HP * (((d->modulated_mask >> i) & 1u) ? 3 : 1);
  return n;
}

static int fused_lds_bytes(const cinema_siren_desc *d) {
  return FUSED_TS * 64 * (ksteps(d->HP) + ksteps(d->LP)) + 4 * (fused_bias_floats(d) + 4 * d->HP) + FUSED_TS * 16;
}

bool fused_eligible(const cinema_siren_desc *d) {
  return d->precision == CINEMA_PREC_BF16 && d->HP <= FUSED_MAX_HP && fused_lds_bytes(d) <= 160 * 1024;
}

void fused_plan(cinema_siren_desc *d, long *pk) {
  d->pk_fs = d->pk_bs = d->pk_side = d->pk_ns = d->pk_nb = -1;
  d->fs_steps = d->bs_steps = d->ns_steps = d->nb_steps = 0;
  if (!fused_eligible(d)) return;
  StreamPlan p;
  const long nslab = d->HP / 64;
  plan_fwd_stream(d, p);
  d->pk_fs = *pk; d->fs_steps = p.steps;
  *pk += nslab * p.steps * FUSED_STEP_ELEMS;
  plan_bwd_stream(d, p);
  d->pk_bs = *pk; d->bs_steps = p.steps;
  *pk += nslab * p.steps * FUSED_STEP_ELEMS;
  *pk += (long)FUSED_RING * FUSED_STEP_ELEMS;  // zero pad: the rings prefetch past the end of the last slab
  // nodes path: the same two streams without the linear_lats segments
  plan_fwd_stream(d, p, true);
  d->pk_ns = *pk; d->ns_steps = p.steps;
  *pk += nslab * p.steps * FUSED_STEP_ELEMS;
  plan_bwd_stream(d, p);
  d->pk_nb = *pk; d->nb_steps = p.steps;
  *pk += nslab * p.steps * FUSED_STEP_ELEMS;
  *pk += (long)FUSED_RING * FUSED_STEP_ELEMS;
  d->pk_side = *pk;                            // fp32 side block (biases, first-layer weights) in a T-typed buffer
  *pk += 2L * (fused_bias_floats(d) + 4 * d->HP + 16 * d->HP);
}

static int fused_pack_args(const cinema_siren_desc *d, const float *params, void *packed, PackStreamArgs &a, PackSideArgs &b,
                           int &g_ps, int &g_sd, bool nodes = false) {
  StreamPlan p0, p1;
  plan_fwd_stream(d, p0, nodes);
  plan_bwd_stream(d, p1);
  CINEMA_REQUIRE(p0.nseg + p1.nseg <= FUSED_MAX_SEGS, "fused_pack: %d segments exceed the table", p0.nseg + p1.nseg);
  for (int k = 0; k < p0.nseg; ++k) a.seg[k] = p0.seg[k];
  for (int k = 0; k < p1.nseg; ++k) a.seg[p0.nseg + k] = p1.seg[k];
  a.nseg0 = p0.nseg; a.nseg1 = p1.nseg; a.steps0 = p0.steps; a.steps1 = p1.steps; a.nslab = d->HP / 64;
  a.params = params; a.pairmap = nodes ? 3 : (fused_bwd_wide(d) ? 4 : 0);
  a.dst0 = (bf16_t *)packed + (nodes ? d->pk_ns : d->pk_fs); a.dst1 = (bf16_t *)packed + (nodes ? d->pk_nb : d->pk_bs);
  const long nchunk = (long)a.nslab * (a.steps0 + a.steps1) * 256;
  CINEMA_REQUIRE(nchunk < (1L << 31), "fused_pack: weight streams too large for 32-bit chunk indices");
  g_ps = (int)std::min<long>((nchunk + 255) / 256, 4096);
  b = PackSideArgs{};
  b.params = params; b.dst = (float *)((bf16_t *)packed + d->pk_side);
  for (int i = 0; i < d->n_layers; ++i) {
    b.off_b[i] = d->off_b[i]; b.off_bl[i] = d->off_bl[i];
    b.lb_b[i] = b.lb_total; b.lb_total += d->HP;
    b.lb_bl[i] = b.lb_total;
    if ((d->modulated_mask >> i) & 1u) b.lb_total += 2 * d->HP;
  }
  b.off_w0 = d->off_w[0];
  b.mod_mask = d->modulated_mask; b.n_layers = d->n_layers; b.H = d->hidden; b.HP = d->HP; b.in_dim = d->in_dim;
  b.off_wf = d->off_wf; b.out_dim = d->out_dim; b.lat = d->lat;
  b.dst_wf = (bf16_t *)packed + d->pk_wf; b.LP = d->LP;
  g_sd = (b.lb_total + 20 * d->HP + 16 * (d->HP + d->LP) + 255) / 256;
  return 0;
}

static int fused_pack_impl(const cinema_siren_desc *d, const float *params, void *packed, hipStream_t s, bool nodes);
int fused_pack(const cinema_siren_desc *d, const float *params, void *packed, hipStream_t s) { return fused_pack_impl(d, params, packed, s, false); }
int fused_pack_nodes(const cinema_siren_desc *d, const float *params, void *packed, hipStream_t s) { return fused_pack_impl(d, params, packed, s, true); }

static int fused_pack_impl(const cinema_siren_desc *d, const float *params, void *packed, hipStream_t s, bool nodes) {
  if (d->pk_fs < 0) return 0;
  PackStreamArgs a;
  PackSideArgs b;
  int g_ps = 0, g_sd = 0;
  if (int rc = fused_pack_args(d, params, packed, a, b, g_ps, g_sd, nodes)) return rc;
  if (g_ps > 0) {
    hipLaunchKernelGGL(k_pack_stream, dim3(g_ps), dim3(256), 0, s, a);
    CINEMA_CHECK_HIP(hipGetLastError());
  }
  hipLaunchKernelGGL(k_pack_side, dim3(g_sd), dim3(256), 0, s, b);
  CINEMA_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int cinema_step_head(const cinema_siren_desc *d, const float *params, void *packed, const float *coords,
                                const int32_t *idx, const float *embed, int32_t tf_dim, const float *latents, const float *cond,
                                int64_t N, int64_t Npad, int64_t S, int32_t C, int32_t lx, int32_t ly, int32_t lz,
                                int32_t cond_dims, int32_t cond_by_subject, float *coords_t, void *lat_out, float *shadow,
                                void *node_work, float *dcoords, int64_t n_dcoords, float *sums, const int64_t *voxel, const uint8_t *labels,
                                int64_t V, int32_t n_cls, const float *class_w, float *scratch, void *stream) {
  CINEMA_REQUIRE(d && params && packed && coords && idx && latents && coords_t && lat_out, "cinema_step_head: null argument");
  CINEMA_REQUIRE(fused_eligible(d) && d->pk_fs >= 0, "cinema_step_head: needs the activation-stationary bf16 path (hidden <= 512)");
  CINEMA_REQUIRE(N >= 1 && Npad >= N && S >= 1, "cinema_step_head: bad sizes N=%ld Npad=%ld S=%ld", (long)N, (long)Npad, (long)S);
  CINEMA_REQUIRE(lx == 3 && ly == 3 && lz == 3 && C >= 4 && C <= TFS_CMAX && C % 4 == 0 && d->LP % 4 == 0,
                 "cinema_step_head: needs the 3x3x3 grid, C <= %d, C %% 4 == 0 (got %d x %d x %d x %d)", TFS_CMAX, C, lx, ly, lz);
  CINEMA_REQUIRE(embed == nullptr || tf_dim == 6 || tf_dim == 9 || tf_dim == 12, "tf embedding width %d unsupported (6, 9 or 12)", tf_dim);
  CINEMA_REQUIRE(cond_dims == 0 || cond != nullptr, "cond_dims %d but cond is null", cond_dims);
  CINEMA_REQUIRE(C + cond_dims == d->lat, "cinema_step_head: C + cond_dims = %d != lat_size %d", C + cond_dims, d->lat);
  StepHeadArgs a = {};
  if (int rc = fused_pack_args(d, params, packed, a.ps, a.sd, a.g_ps, a.g_sd)) return rc;
  a.k1 = TriFwdSortedArgs{coords, idx, embed, tf_dim, latents, cond, (long)N, (long)Npad, (long)S, C, cond_dims, d->LP, coords_t,
                          lat_out, cond_by_subject ? 1 : 0};
  a.g_k1 = (int)((Npad + TFS_CHUNK - 1) / TFS_CHUNK);
  // backward-only repacking: the transposed linear_lats copies of the d-latent GEMM (cinema_siren_pack_part(.., 2))
  int nj = 0;
  const int H = d->hidden, L = d->lat;
  for (int i = 0; i < d->n_layers; ++i)
    if ((d->modulated_mask >> i) & 1u) {
      CINEMA_REQUIRE(nj + 4 <= HEAD_MAX_JOBS, "cinema_step_head: more than %d modulated layers", HEAD_MAX_JOBS / 4);
      const float *wl = params + d->off_wl[i];
      a.job[nj++] = PackJob{wl, d->pk_wlsT[i], H, L, L, d->LP, d->HP, 1};
      a.job[nj++] = PackJob{wl + (size_t)H * L, d->pk_wltT[i], H, L, L, d->LP, d->HP, 1};
      // row-major copies: the node projections of cinema_nodegrad_prepare (cinema_siren_pack_part(.., 8))
      a.job[nj++] = PackJob{wl, d->pk_wls[i], H, L, L, d->HP, d->LP, 0};
      a.job[nj++] = PackJob{wl + (size_t)H * L, d->pk_wlt[i], H, L, L, d->HP, d->LP, 0};
    }
  a.packed = packed;
  a.g_job = 64 * nj;
  if (shadow != nullptr) {
    const int G = lx * ly * lz;
    a.nm_lat = latents; a.nm_shadow = shadow; a.nm_S = (long)S; a.nm_C = C; a.nm_G = G;
    const long nwg = (long)((G + 31) / 32) * ((C + 31) / 32) * S;
    CINEMA_REQUIRE(nwg < (1L << 24), "cinema_step_head: latent table too large (%ld tiles)", nwg);
    a.g_nm = (int)nwg;
  }
  if (node_work != nullptr) {                  // T and Tc of the node-space backward (cinema_nodegrad_prepare, flags & 1)
    CINEMA_REQUIRE(cond_by_subject || cond_dims == 0, "cinema_step_head: node rows need per-subject conditions");
    CINEMA_REQUIRE(ng_eligible(d, lx, ly, lz) && d->LP % 32 == 0, "cinema_step_head: node rows need a stack cinema_nodegrad_eligible accepts");
    CINEMA_REQUIRE(Npad % 128 == 0, "cinema_step_head: node tables need Npad %% 128 == 0");
    const NgPlan p = ng_plan(d, S, Npad);      // the same plan cinema_nodegrad_work_bytes(d, S, Npad) sized the buffer with
    a.nr = NodeRowsCArgs{latents, cond, (long)S, C, lx * ly * lz, cond_dims, d->LP, (bf16_t *)((char *)node_work + p.T),
                         (bf16_t *)((char *)node_work + p.Tc)};
    a.g_nr = (int)(S * (d->LP / 32));
    a.ngw = ng_weight_args(p, node_work, coords, idx, embed, tf_dim, (long)N, lx, ly, lz);
    a.g_ngw = ng_weight_grid(p);
  }
  if (sums != nullptr) {
    CINEMA_REQUIRE(scratch != nullptr && n_dcoords >= 0 && (dcoords != nullptr || n_dcoords == 0), "cinema_step_head: bad prepare arguments");
    CINEMA_REQUIRE(class_w == nullptr || (voxel && labels && n_cls >= 1), "cinema_step_head: class_w needs the label source");
    const long work = std::max<long>(class_w != nullptr ? (long)N : 0, (long)n_dcoords);
    a.g_pr = (int)std::max<long>(1, std::min<long>((work + 255) / 256, PREP_MAX_GRID));
    a.pr = StepPrepareArgs{dcoords, (long)n_dcoords, sums, voxel, idx, labels, (long)V, (long)N, n_cls, class_w, scratch};
  }
  static const int head_skip = getenv("CINEMA_HEAD_SKIP") ? atoi(getenv("CINEMA_HEAD_SKIP")) : 0;
  a.skip = head_skip;
  if (head_skip) fprintf(stderr, "[head] grids: prepare %d k1 %d weights %d jobs %d stream %d side %d rows %d shadow %d\n", a.g_pr, a.g_k1, a.g_ngw, a.g_job, a.g_ps, a.g_sd, a.g_nr, a.g_nm);
  hipLaunchKernelGGL(k_step_head, dim3((unsigned)(a.g_k1 + a.g_ps + a.g_sd + a.g_job + a.g_nr + a.g_ngw + a.g_nm + a.g_pr)), dim3(256), 0,
                     (hipStream_t)stream, a);
  CINEMA_CHECK_HIP(hipGetLastError());
  return 0;
}

// Experiment, OFF by default (CINEMA_SAVED_FP8=1 turns it on; read once per process): the tensors only the backward's
// epilogue reads (c, z, s) kept as OCP fp8 -- half the bytes of 12 of the forward's 18 saved tensors, forward 222 -> 183 us,
// backward chain 277 -> 256 us at cfg-D.  It does NOT hold the parity bar: the 2^-4 rounding enters every dz element and
// the sums over features / samples have random signs, so it does not average out relative to the gradient
// (test_g5_decoder_golden[cfgT-bf16]: 5 % L2 on the first layer's weight gradient against the 1 % bound).
bool fused_saved_fp8() {
  static const bool on = [] { const char *e = getenv("CINEMA_SAVED_FP8"); return e && e[0] == '1'; }();
  return on;
}

// developer hook (not part of the C ABI in include/): choose the workgroup / wave whose phase timestamps the next
// forward launches record, and read them back
static int g_trace_block = -1, g_trace_wave = 0;
extern "C" int cinema_debug_fused_trace(int block, int wave, long long *host_out128) {
  g_trace_block = block; g_trace_wave = wave;
  if (host_out128) CINEMA_CHECK_HIP(hipMemcpyFromSymbol(host_out128, HIP_SYMBOL(g_fused_trace), sizeof(long long) * 128));
  return 0;
}

// Launch plan of the chain kernels: Npad rows as g4 workgroups of 64 samples followed by g3 workgroups of 48, ONE launch.
// One workgroup per CU is resident and a workgroup's time is its (sample-count independent) weight stream plus the
// epilogues of its samples.  A batch that fills the chip between one and two times (cfg-D: 392 tiles of 64 on 256 CUs)
// pays two weight streams per CU either way; with 48-sample tiles behind the first chip filling the second stream
// carries three quarters of the epilogue traffic: forward 234 -> 222 us, backward chain 291 -> 271 us.  (As two
// launches it LOSES: a lone first round of 254 workgroups in lockstep takes 180 us in the backward -- all epilogues hit
// HBM at once -- where the mixed launch lets early finishers start the short tiles.  A delayed start of every other
// workgroup, meant to desynchronise the phases, changed nothing.)  CINEMA_FUSED_SPLIT=0 keeps 64-sample tiles only.
static void fused_launch_plan(int64_t Npad, int *g4, int *g3) {
  static const bool enabled = [] { const char *e = getenv("CINEMA_FUSED_SPLIT"); return !(e && e[0] == '0'); }();
  const int ncu = cinema_cu_count();
  const long units = Npad / 16;                      // Npad is a multiple of 64
  *g4 = (int)(units / 4); *g3 = 0;
  if (!enabled || units <= 4L * ncu) return;
  for (int c = ncu; c > ncu - 3 && c > 0; --c) {
    const long rest = units - 4L * c;
    if (rest > 0 && rest % 3 == 0 && rest / 3 <= ncu) { *g4 = c; *g3 = (int)(rest / 3); return; }
  }
}

int fused_forward(const cinema_siren_desc *d, const float *params, const void *packed, const float *coords_t,
                  const void *lat, int64_t N, int64_t Npad, void *saved, float *out, hipStream_t s, void *node_work,
                  int64_t node_S) {
  CINEMA_REQUIRE(fused_eligible(d) && d->pk_fs >= 0, "fused_forward: descriptor not planned for the fused path");
  CINEMA_REQUIRE(Npad % FUSED_TS == 0, "fused_forward: Npad %ld not a multiple of %d", (long)Npad, FUSED_TS);
  FusedFwdArgs a = {};
  a.stream = (const bf16_t *)packed + d->pk_fs;
  a.slab_elems = (long)d->fs_steps * FUSED_STEP_ELEMS;
  a.params = params;
  for (int i = 0; i < d->n_layers; ++i) {
    a.off_b[i] = d->off_b[i]; a.off_bl[i] = d->off_bl[i];
    a.sv_y[i] = d->sv_y[i]; a.sv_c[i] = d->sv_c[i]; a.sv_z[i] = d->sv_z[i]; a.sv_s[i] = d->sv_s[i];
    a.lb_b[i] = a.lb_total; a.lb_total += d->HP;
    a.lb_bl[i] = a.lb_total;
    if ((d->modulated_mask >> i) & 1u) a.lb_total += 2 * d->HP;
  }
  a.trace_block = g_trace_block; a.trace_wave = g_trace_wave;
  a.side = (const float *)((const bf16_t *)packed + d->pk_side);
  a.off_w0 = d->off_w[0];
  a.mod_mask = d->modulated_mask; a.n_layers = d->n_layers; a.in_dim = d->in_dim;
  a.H = d->hidden; a.HP = d->HP; a.LP = d->LP; a.out_dim = d->out_dim;
  a.om_first = d->omega_first; a.om_hidden = d->omega_hidden;
  a.coords_t = coords_t; a.lat = (const bf16_t *)lat;
  a.saved = (bf16_t *)saved; a.slot = (long)Npad * d->HP;
  a.wf = (const bf16_t *)packed + d->pk_wf; a.bf = params + d->off_bf; a.out = out;
  a.xsteps = ksteps(d->HP); a.lsteps = ksteps(d->LP);
  a.xrow = 64 * a.xsteps; a.lrow = 64 * a.lsteps;
  const int smem = fused_lds_bytes(d);
  int g3;
  fused_launch_plan(Npad, &a.g4, &g3);
  a.g_chain = a.g4 + g3;
  long g_proj = 0;
  if (node_work != nullptr) {
    CINEMA_REQUIRE(Npad % 128 == 0 && node_S >= 1, "fused_forward: node projections need Npad %% 128 == 0 and S >= 1");
    const NgPlan p = ng_plan(d, node_S, Npad);
    CINEMA_REQUIRE(ng_proj_smem(d) <= smem, "fused_forward: node projections need %d bytes of LDS, the forward has %d", ng_proj_smem(d), smem);
    a.proj = ng_proj_args(d, p, node_work, packed);
    g_proj = ng_proj_grid(p);
    CINEMA_REQUIRE(a.g_chain + g_proj < (1L << 30), "fused_forward: too many subjects");
  }
  if (fused_saved_fp8()) {
    CINEMA_SMEM_ATTR(k_fused_fwd<true>, smem);
    hipLaunchKernelGGL(k_fused_fwd<true>, dim3((unsigned)(a.g_chain + g_proj)), dim3(512), smem, s, a);
  } else {
    CINEMA_SMEM_ATTR(k_fused_fwd<false>, smem);
    hipLaunchKernelGGL(k_fused_fwd<false>, dim3((unsigned)(a.g_chain + g_proj)), dim3(512), smem, s, a);
  }
  CINEMA_CHECK_HIP(hipGetLastError());
  (void)N;
  return 0;
}

int fused_backward_chain(const cinema_siren_desc *d, const void *packed, const float *coords_t, const void *saved,
                         const float *dout, int64_t Npad, void *gbuf, float *gp, float *dcoords, bool wgrads,
                         hipStream_t s, void *dout16, float *dbf, bool lat_bias, float *colsum_rep, int rep_w) {
  CINEMA_REQUIRE(fused_eligible(d) && d->pk_bs >= 0, "fused_backward_chain: descriptor not planned for the fused path");
  if (fused_bwd_wide(d) && !fused_saved_fp8())     // the stream is packed for the wide kernel then (fused_pack_args)
    return wide_backward_chain(d, packed, coords_t, saved, dout, Npad, gbuf, gp, dcoords, wgrads, s, dout16, dbf, lat_bias, colsum_rep, rep_w);
  CINEMA_REQUIRE(Npad % FUSED_TS == 0, "fused_backward_chain: Npad %ld not a multiple of %d", (long)Npad, FUSED_TS);
  FusedBwdArgs a = {};
  a.stream = (const bf16_t *)packed + d->pk_bs;
  a.slab_elems = (long)d->bs_steps * FUSED_STEP_ELEMS;
  a.side_w0 = (const float *)((const bf16_t *)packed + d->pk_side) + fused_bias_floats(d);
  a.dout = dout; a.coords_t = coords_t;
  a.saved = (const bf16_t *)saved; a.slot = (long)Npad * d->HP; a.gbuf = (bf16_t *)gbuf;
  for (int i = 0; i < d->n_layers; ++i) {
    a.sv_c[i] = d->sv_c[i]; a.sv_z[i] = d->sv_z[i]; a.sv_s[i] = d->sv_s[i];
    a.gb_z[i] = d->gb_z[i]; a.gb_s[i] = d->gb_s[i]; a.gb_t[i] = d->gb_t[i];
    a.off_b[i] = d->off_b[i]; a.off_bl[i] = d->off_bl[i];
  }
  a.off_w0 = d->off_w[0];
  a.gp = gp; a.wgrads = wgrads ? 1 : 0; a.latb = (wgrads && lat_bias) ? 1 : 0; a.dcoords = dcoords;
  a.mod_mask = d->modulated_mask; a.n_layers = d->n_layers; a.in_dim = d->in_dim; a.H = d->hidden; a.HP = d->HP;
  a.out_dim = d->out_dim; a.xsteps = ksteps(d->HP); a.xrow = 64 * a.xsteps;
  a.om_first = d->omega_first; a.om_hidden = d->omega_hidden;
  const int smem = FUSED_TS * a.xrow + 20 * d->HP * 4 + FUSED_TS * 64 + FUSED_TS * 16;
  CINEMA_REQUIRE(smem <= 160 * 1024, "fused_backward_chain: LDS budget exceeded (%d bytes)", smem);
  int g3;
  fused_launch_plan(Npad, &a.g4, &g3);
  a.g_chain = a.g4 + g3;
  if (dout16 != nullptr) {
    a.fin = FinalBwdArgs{dout, d->out_dim, dbf, (int)Npad};
    a.dout16 = (bf16_t *)dout16;
    a.g_fin = (int)std::min<int64_t>((Npad + 511) / 512, 128);
  }
  const dim3 grid((unsigned)(a.g_chain + a.g_fin));
#define FUSED_BWD_LAUNCH(FP8, LATB)                                                            \
  do {                                                                                         \
    CINEMA_SMEM_ATTR((k_fused_bwd<FP8, LATB>), smem);                                          \
    hipLaunchKernelGGL((k_fused_bwd<FP8, LATB>), grid, dim3(512), smem, s, a);                 \
  } while (0)
  if (fused_saved_fp8()) {
    if (a.latb) FUSED_BWD_LAUNCH(true, true); else FUSED_BWD_LAUNCH(true, false);
  } else {
    if (a.latb) FUSED_BWD_LAUNCH(false, true); else FUSED_BWD_LAUNCH(false, false);
  }
#undef FUSED_BWD_LAUNCH
  CINEMA_CHECK_HIP(hipGetLastError());
  return 0;
}
