// Activation-stationary ("fused") SIREN kernels, bf16 only: internal interface between fused.hip and siren.hip.
#pragma once
#include "common.h"

// Weight streams (cinema_siren_desc::pk_fs / pk_bs): the MFMA A fragments of every GEMM of the stack, in the exact
// order one wave consumes them, so that a wave's weight traffic is ONE contiguous, fully coalesced stream:
//   [slab = 64 output features][k32 step][fragment ft = 0..3][lane = 0..63][8 x bf16]
// lane (i = lane & 15, gk = lane >> 4) of fragment ft holds W[slab*64 + 16*(i>>2) + 4*ft + (i&3)][32*step + 8*gk + 0..7],
// which makes accumulator register r of fragment ft in lane (i, g) the output feature slab*64 + 16*g + 4*ft + r of
// sample i: 16 consecutive features per lane, as in gemm_engine.h.
#define FUSED_STEP_ELEMS 2048     // bf16 elements per k32 step of one slab (4 KB)
#define FUSED_RING 4              // steps a wave keeps in flight
#define FUSED_MAX_HP 512          // 8 waves x 64 features
#define FUSED_TS 64               // samples per workgroup

bool fused_eligible(const cinema_siren_desc *d);
void fused_plan(cinema_siren_desc *d, long *pk);                   // appends the stream regions to the packed layout
int fused_pack(const cinema_siren_desc *d, const float *params, void *packed, hipStream_t s);
int fused_forward(const cinema_siren_desc *d, const float *params, const void *packed, const float *coords_t,
                  const void *lat, int64_t N, int64_t Npad, void *saved, float *out, hipStream_t s, void *node_work = nullptr,
                  int64_t node_S = 0);
// node_work != NULL: the node projections of the node-space backward (nodegrad.h) ride as tail workgroups of the launch
int fused_backward_chain(const cinema_siren_desc *d, const void *packed, const float *coords_t, const void *saved,
                         const float *dout, int64_t Npad, void *gbuf, float *grad_params, float *dcoords, bool wgrads,
                         hipStream_t s, void *dout16 = nullptr, float *dbf = nullptr, bool lat_bias = true,
                         float *colsum_rep = nullptr, int rep_w = 0);
// colsum_rep != NULL (wide kernel only): the b_l and first-layer weight gradient sums go into the replicas of nodegrad.h
// lat_bias = false: the linear_lats bias gradients are left to the caller (node gradients take them from dU)
// fused_wide.hip: the same backward chain for workgroups of up to 112 samples (two accumulator passes per wave); it reads
// the backward stream in the pair-major feature map, so packing and launch both ask fused_bwd_wide(d)
bool fused_bwd_wide(const cinema_siren_desc *d);
int wide_backward_chain(const cinema_siren_desc *d, const void *packed, const float *coords_t, const void *saved,
                        const float *dout, int64_t Npad, void *gbuf, float *grad_params, float *dcoords, bool wgrads,
                        hipStream_t s, void *dout16, float *dbf, bool lat_bias, float *colsum_rep, int rep_w);

// "nodes" path (fused_nodes.hip, node.hip; layout in nodes.h): the same chain kernels with the FiLM modulation read from
// per-subject node tables.  Its forward stream (pk_ns) holds the main weights only.
int nodes_forward(const cinema_siren_desc *d, const float *params, const void *packed, const float *coords_t, const void *wrows,
                  const void *tiles, const void *U, const void *V, int64_t S, int64_t N, int64_t Npad, void *saved, float *out,
                  int round_s, hipStream_t s);
int nodes_backward_chain(const cinema_siren_desc *d, const void *packed, const float *coords_t, const void *wrows,
                         const void *tiles, const void *U, int64_t S, const void *saved, const float *dout, int64_t N,
                         int64_t Npad, void *gbuf, float *grad_params, float *dcoords, bool wgrads, int round_s, hipStream_t s,
                         void *dout16 = nullptr, float *dbf = nullptr);
int fused_pack_nodes(const cinema_siren_desc *d, const float *params, void *packed, hipStream_t s);
