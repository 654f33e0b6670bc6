// Kernel arguments of the halo-resident implicit GEMMs (igemm5.hip: igemm5 / igemm6; igemm8.hip), filled by op_igemm5.
#pragma once
#include "common.h"

struct KP5 {
  const char* in;
  const char* w;      // [4 phases][cout][4*cin] (K order: tap, channel)
  char* out;
  const char* zero;
  const float* bias;
  float* partials;
  unsigned long long* stat_acc; int stat_pg, stat_reps;   // IgemmArgs::stat_acc
  int Hs, Ws, n;      // the small grid (MODE 1: input, MODE 0: output)
  int TH, TW;         // patch of the small grid, TH*TW = 256; TW a power of two
  int tiles_x, tiles_per_img, mtiles;
  int cin, ldin, coffin;
  int cout, ldout, coffout;
  int nchunk;         // cin / 64
  int relu_in, relu_cend, act_out;
  int ntiles;
  const char* mask; int ldmask, coffmask; float mask_slope;   // fused activation backward (IgemmArgs::mask)
  const char* add; int ldadd, coffadd;
  const unsigned long long* mask_bits;   // IgemmArgs::mask_bits (igemm8, MODE 3 only; else null)
  // IgemmArgs::c1w_* (igemm8, MODE 3 with mask_bits only): the weight gradient of the single-channel layer below, from the tile
  const float* c1w_img; float* c1w_part; float c1w_scale; int c1w_skip_out;
  // fused BatchNorm-backward reduction (IgemmArgs::bwd_*)
  const char* bwd_x; int bwd_ldx;
  const float* bwd_scale; const float* bwd_shift; const float* bwd_mean; const float* bwd_inv; int bwd_stride;
  float bwd_slope;
  int bwd_c0, bwd_c;   // IgemmArgs::bwd_c0 / bwd_c (igemm8 only; igemm6: 0 / cout)
  unsigned long long* bwd_acc; int bwd_reps; int bwd_pg_tiles;   // bwd_pg_tiles: M tiles per BatchNorm population (0: one population)
  int pool;           // MODE 2: the epilogue stores the 2x2 max pool of the tile (IgemmArgs::pool2)
  int dbg_epi;        // builds with -DGI_ABLATION only (GI_EPI_DBG): 1 = all tiles store into one 64 KiB window (no HBM write burst)
};

