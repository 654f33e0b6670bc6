// igemm8: the halo-resident implicit GEMM of igemm6 (igemm5.hip; Conv2d 4x4 / s2 / p1 of lib/models/networks.py:285-318 as a
// stride-2 gather, MODE 0, or as sub-pixel phases, MODE 1 / 3) re-cut so that TWO workgroups share a CU.
//
// Why. igemm6 runs one 8-wave workgroup per CU (200 VGPRs x 8 waves, 144 KiB of LDS). Its per-tile fixed costs - launch and
// offset set-up 2.9 us, the wait for the first halo and weight slices 3.1 us, the epilogue 2 - 5.75 us (measured by ablation,
// DESIGN.md 4.1g) - are serial with the K loop because nothing else is resident: on the layers with a 16-step K loop (d2, u2,
// the critic's conv2 and its input gradient) they are 40 - 55 % of a tile's time, and on every layer the eight waves reach
// their barrier, their fragment reads and their MFMAs in lockstep (two waves per SIMD running the same program:
// guides/MI355X_MICROARCH.md, "Two waves per SIMD", item 9). A persistent form of igemm6 was measured slower (registers).
//
// What changes. A workgroup is FOUR waves with the same 256 x 128 output tile: every wave owns 64 pixels x all 128 columns
// (accumulators 128 registers; 12 fragment reads per 32 MFMAs = 0.375 per MFMA instead of 0.5), the K step is 32 channels:
//   LDS per workgroup: halo 2 x 320 rows x 64 B = 40 KiB | weight ring 4 stages x 128 rows x 64 B = 32 KiB  -> 72 KiB,
// so two workgroups (2 x 4 waves = the same two waves per SIMD) are resident per CU and the hardware overlaps one's
// prologue / epilogue / barrier waits with the other's MFMAs. Per wave and per MFMA the LDS-DMA issue rate (2 weight pieces
// + 1.25 halo pieces per 32 MFMAs), the bytes per MAC (26 KiB per 256 x 128 x 64 MACs) and the wait / barrier schedule are
// igemm6's: a step (tap t of channel chunk c) waits for everything but the pieces issued during the previous step, passes
// ONE workgroup barrier, and issues the weight slice of step s + 3 and (taps 0, 1) the next chunk's halo between its MFMAs.
// Rows are 64 bytes (32 channels); a 1 KiB LDS-DMA piece is 16 rows; 16-byte chunk q of row R sits at physical chunk
// q ^ ((R >> 1) & 3) (applied to the per-lane DMA source offset): `ds_read_b128` of 16 consecutive rows is conflict-free
// for every alignment of the first row (enumerated over the instruction's four lane groups).
// Fragment schedule of a step (32 MFMAs = 4 pixel tiles x 8 column tiles, column half H0 = tiles 0..3, H1 = tiles 4..7):
//   in registers at the barrier: the pixel fragments a[0..3] and the column fragments b[0..3] of this step (read during the
//   previous step's H1: the data of a step lands one step early); during H0: b[4..7] (MODE 3 also the four pixel fragments
//   of px 1, whose halo rows are one column to the right); during H1: a[0..3], b[0..3] of the NEXT step.
// MODE 3 (64 output channels per N tile): columns 0..63 = px 0, 64..127 = px 1 of the same 64 channels, as in igemm6.
// K order per accumulator: 32-channel chunks ascending, taps 0..3 inside a chunk (igemm6: 64-channel chunks, per tap both
// 32-channel halves) - the two kernels agree to fp32 rounding, not bit for bit.
#include <stdlib.h>

#include "common.h"
#include "stat_acc.h"
#include "halo_args.h"

namespace {

__device__ __forceinline__ void blds16(__amdgpu_buffer_rsrc_t rs, unsigned voff, int soff, char* lds_wave_base) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds_wave_base, 16, voff, soff, 0, 0);
}

// ---- epilogue (contract of igemm3 / epilogue5): bias / activation, per-tile column statistics, the tile staged through LDS
//      and stored with 16-byte rows; optional fused activation backward and BatchNorm-backward reduction. 256 threads, wave w
//      holds rows 64 w .. 64 w + 63 x 128 columns. -----------------------------------------------------------------------------
// BITS: the instantiation carries the sign-word form of the fused activation backward (IgemmArgs::mask_bits). Only the dual-px
// kernel without the input ReLU (the input-gradient GEMMs of the second layers) does: the forward's <3, relu> instantiation sits at
// 256 VGPRs and spills with any more epilogue code (measured: u2 77.5 -> 101 us with 116 bytes of scratch).
template <int MODE, int BN, bool BITS>
__device__ __forceinline__ void epilogue8(const KP5& p, f4_t (&acc)[4][BN / 16], char* smem, int tid, int lane, int wave, int mt_idx,
                                          int ph, int py, int px, int n0, int img, int y0, int x0, int lgTW) {
  constexpr bool DUAL = MODE == 3;
  constexpr bool PH = MODE == 1 || MODE == 3;
  constexpr int BM = 256, MT = 4, NT = BN / 16;
  const int lr = lane & 15, lq = lane >> 4;
  auto out_pixel = [&](int m) -> int {
    const int ty_l = m >> lgTW, tx_l = m & (p.TW - 1);
    if constexpr (PH) return (img * 2 * p.Hs + 2 * (y0 + ty_l) + py) * (2 * p.Ws) + 2 * (x0 + tx_l) + px;   // DUAL: + 1 for px 1, below
    else return (img * p.Hs + y0 + ty_l) * p.Ws + x0 + tx_l;
  };
  constexpr int SLD = BN + 8;
  half_t* stg = (half_t*)smem;
  float* red = (float*)(smem + (int64_t)BM * SLD * 2);   // [4 waves][BN][2]
  const bool stats = p.partials || p.stat_acc;
  // (IgemmArgs::c1w_*, stage below) the image values of this wave's four K blocks are requested first: they arrive while the
  // accumulators are staged
  float c1w_x[BITS ? 4 : 1][8];
  if constexpr (BITS) {
    if (p.c1w_part) {
      const int Himg = 4 * p.Hs, Wimg = 4 * p.Ws;
      const int tap = lane & 15, kq = lane >> 4, ky = tap >> 2, kx = tap & 3;
      const float* ximg = p.c1w_img + (int64_t)img * Himg * Wimg;
#pragma unroll
      for (int bi = 0; bi < 4; ++bi) {
        const int b = wave + 4 * bi;
        const int P = b & 1, mrow = (b >> 1) * 32 + 8 * kq;
        const int iy = 4 * (y0 + (mrow >> lgTW)) + 2 * py - 1 + ky;
        const bool yok = iy >= 0 && iy < Himg;
        const float* rowp = ximg + (int64_t)(yok ? iy : 0) * Wimg;
        const int ix0 = 4 * (x0 + (mrow & (p.TW - 1))) + 2 * P - 1 + kx;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int ix = ix0 + 4 * j;
          const bool ok = yok && ix >= 0 && ix < Wimg;
          const float a = rowp[ok ? ix : 0];
          c1w_x[bi][j] = ok ? a : 0.f;
        }
      }
    }
  }
  if constexpr (BITS) {
    if (p.mask_bits) {
      // fused activation backward from sign words (IgemmArgs::mask_bits): the two px phases of a row are adjacent output pixels,
      // i.e. 16 consecutive bytes; element (column nt * 16 + 4 lq + r) is bit (nt & 1) * 16 + 4 lq + r of word (nt >> 1). The
      // slope goes to the fp32 accumulator; a second gradient (`add`) is added in the copy-out loop below.
      u4_t mb[MT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) mb[mt] = *(const u4_t*)(p.mask_bits + out_pixel(wave * 64 + mt * 16 + lr));
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int ch = nt * 16 + 4 * lq;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const int t = (int)(mb[mt][nt >> 1] >> ((nt & 1) * 16 + 4 * lq));
          h4_t o;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float v = acc[mt][nt][r];
            const int pos = (t << (31 - r)) >> 31;                         // -1 where the bit is set
            o[r] = (half_t)__builtin_bit_cast(float, (__builtin_bit_cast(int, v) & pos) | (__builtin_bit_cast(int, v * p.mask_slope) & ~pos));
          }
          *(h4_t*)(stg + (wave * 64 + mt * 16 + lr) * SLD + ch) = o;
        }
      }
    }
  }
  // the accumulator loop with the activation as a compile-time constant (common.h), and - MODE 0 / 1, which have the registers for
  // the extra code paths (the other modes spill: MODE 2 128 bytes of scratch, MODE 3 112) - the statistics and bias flags as well
  auto acc_loop = [&](auto ACTc, auto st, auto bi) {     // st / bi: std::true_type / std::false_type, or bool
    constexpr int ACT = decltype(ACTc)::value;
    constexpr bool CT = !std::is_same<decltype(st), bool>::value;   // run-time flags: the sums run unconditionally (as before)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int ch = nt * 16 + 4 * lq;                   // column of the tile (DUAL: px * 64 + channel)
      float bs[4] = {0.f, 0.f, 0.f, 0.f};
      if (bi) {
#pragma unroll
        for (int r = 0; r < 4; ++r) bs[r] = p.bias[n0 + (DUAL ? (ch & 63) : ch) + r];
      }
      float s[4] = {0.f, 0.f, 0.f, 0.f}, q[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        h4_t o;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = acc[mt][nt][r];
          if constexpr (CT) {
            if (bi) v += bs[r];
            if (st) { s[r] += v; q[r] += v * v; }
          } else {
            v += bs[r];
            s[r] += v;
            q[r] += v * v;
          }
          o[r] = (half_t)gi_act_c<ACT>(v);
        }
        *(h4_t*)(stg + (wave * 64 + mt * 16 + lr) * SLD + ch) = o;
      }
      if (st) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { s[r] = gi_row16_sum(s[r]); q[r] = gi_row16_sum(q[r]); }
        if (lr == 0) {
#pragma unroll
          for (int r = 0; r < 4; ++r) { red[(wave * BN + ch + r) * 2] = s[r]; red[(wave * BN + ch + r) * 2 + 1] = q[r]; }
        }
      }
    }
  };
  if (!(BITS && p.mask_bits))
  gi_with_act(p.act_out, [&](auto ACTc) {
    if constexpr (MODE == 0 || MODE == 1) {
      gi_with_bool(stats, [&](auto STc) { gi_with_bool(p.bias != nullptr, [&](auto BIc) { acc_loop(ACTc, STc, BIc); }); });
    } else {
      acc_loop(ACTc, stats, p.bias != nullptr);
    }
  });
  __syncthreads();
  if ((p.partials || p.stat_acc) && tid < BN) {
    float s = 0.f, q = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) { s += red[(i * BN + tid) * 2]; q += red[(i * BN + tid) * 2 + 1]; }
    const int phr = MODE == 1 ? ph : (DUAL ? py * 2 + (tid >> 6) : 0);       // sub-pixel phase of this column
    const int col = DUAL ? (tid & 63) : tid;
    if (p.stat_acc) {   // a patch lies inside one image, i.e. inside one BatchNorm population
      const int grp = (p.stat_pg > 0 && mt_idx * BM >= p.stat_pg) ? 1 : 0, rep = (mt_idx + phr) & (p.stat_reps - 1);
      gi_stat_add(p.stat_acc, p.cout, rep, grp, 0, n0 + col, s);
      gi_stat_add(p.stat_acc, p.cout, rep, grp, 1, n0 + col, q);
    } else {
      const int64_t trow = (int64_t)mt_idx + (int64_t)p.mtiles * phr;
      p.partials[(trow * 2 + 0) * p.cout + n0 + col] = s;
      p.partials[(trow * 2 + 1) * p.cout + n0 + col] = q;
    }
  }
  constexpr int CPRO = BN / 8;          // 16-byte chunks per row
  const int oc = tid % CPRO;
  const bool bwd = !DUAL && p.bwd_acc != nullptr && n0 >= p.bwd_c0 && n0 < p.bwd_c0 + p.bwd_c;   // (this N tile lies in the column range)
  const int bn0 = n0 - p.bwd_c0;   // the tile's first channel of the BatchNorm layer
  float bsc[8], bsh[8], bmu[8], biv[8], bs[8], bsx[8];
  if (bwd) {
    const int go = (p.bwd_pg_tiles > 0 && mt_idx >= p.bwd_pg_tiles) ? p.bwd_stride : 0;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int ch = go + bn0 + oc * 8 + e;
      bsc[e] = p.bwd_scale[ch]; bsh[e] = p.bwd_shift[ch]; bmu[e] = p.bwd_mean[ch]; biv[e] = p.bwd_inv[ch];
      bs[e] = bsx[e] = 0.f;
    }
  }
  if constexpr (MODE == 2) {
    if (p.pool) {     // 2x2 / stride-2 max pool of the staged (activated) 8 x 32 patch: 4 x 16 output pixels of (Hs/2) x (Ws/2)
      constexpr int RPQ = 256 / CPRO;
#pragma unroll
      for (int k = 0; k < 64 / RPQ; ++k) {
        const int r = tid / CPRO + k * RPQ;                 // pooled pixel of the patch: (r >> 4, r & 15)
        const int m = (2 * (r >> 4)) * 32 + 2 * (r & 15);   // its top-left source row (TW = 32)
        const char* sp = (const char*)stg + ((int64_t)m * SLD + oc * 8) * 2;
        const h8_t a = *(const h8_t*)sp, b = *(const h8_t*)(sp + SLD * 2), c = *(const h8_t*)(sp + 32 * SLD * 2), d = *(const h8_t*)(sp + 33 * SLD * 2);
        const h8_t v = __builtin_elementwise_max(__builtin_elementwise_max(a, b), __builtin_elementwise_max(c, d));
        const int64_t opx = ((int64_t)img * (p.Hs >> 1) + (y0 >> 1) + (r >> 4)) * (p.Ws >> 1) + (x0 >> 1) + (r & 15);
        *(h8_t*)(p.out + (opx * p.ldout + n0 + oc * 8) * 2) = v;
      }
      return;
    }
  }
  // sign words + second gradient (the generator's d2 input gradient): its own copy loop, four rows in flight, no run-time tests - in
  // the generic loop below hipcc puts every row's loads behind branches with an s_waitcnt vmcnt(0) after each pair (eight dependent
  // round trips per pass; with the byte load among them this GEMM went 66 -> 80 us). The slope is in already (accumulator stage):
  // the chunk's 8 sign bits become four 2 x 16-bit lane masks and the masked gradient is added with v_pk_add_f16.
  constexpr int RP = 256 / CPRO;        // rows per pass of the workgroup
  const int och = n0 + (DUAL ? (oc & 7) : oc) * 8;
  bool copied = false;
  if constexpr (BITS) {
    if (p.c1w_part && p.c1w_skip_out && !p.add) copied = true;   // the result feeds the fused weight gradient below and nothing else
    if (p.mask_bits && p.add) {
      constexpr int NB = 4;
#pragma unroll 1
      for (int h = 0; h < BM / (RP * NB); ++h) {
        int opxs[NB], mbyte[NB];
        u4_t ad[NB];
#pragma unroll
        for (int k = 0; k < NB; ++k) {
          const int r = tid / CPRO + (h * NB + k) * RP;
          opxs[k] = out_pixel(r) + (oc >> 3);
          mbyte[k] = ((const unsigned char*)p.mask_bits)[(int64_t)opxs[k] * 8 + (oc & 7)];
          ad[k] = *(const u4_t*)(p.add + ((int64_t)opxs[k] * p.ldadd + p.coffadd + och) * 2);
        }
#pragma unroll
        for (int k = 0; k < NB; ++k) {
          const int r = tid / CPRO + (h * NB + k) * RP;
          u4_t v = *(const u4_t*)((const char*)stg + ((int64_t)r * SLD + oc * 8) * 2);
          const int b = mbyte[k];
          u4_t a8 = ad[k];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const unsigned lo = (unsigned)((b << (31 - 2 * i)) >> 31) & 0xFFFFu, hi = (unsigned)((b << (30 - 2 * i)) >> 31) & 0xFFFF0000u;
            a8[i] &= lo | hi;
          }
          v = __builtin_bit_cast(u4_t, __builtin_bit_cast(h8_t, v) + __builtin_bit_cast(h8_t, a8));
          if (p.c1w_part) *(u4_t*)((char*)stg + ((int64_t)r * SLD + oc * 8) * 2) = v;   // the finished gradient, for the stage below
          if (!p.c1w_skip_out) *(u4_t*)(p.out + ((int64_t)opxs[k] * p.ldout + p.coffout + och) * 2) = v;
        }
      }
      copied = true;
    }
  }
  // the BatchNorm-backward sums (critic conv3 / conv4 input gradients): the same - their own copy loop, four rows' loads of the
  // layer's raw output in flight
  if constexpr (!DUAL && MODE != 2) {
    if (bwd && !p.mask) {
      constexpr int NB = 4;
#pragma unroll 1
      for (int h = 0; h < BM / (RP * NB); ++h) {
        int opxs[NB];
        u4_t xs[NB];
#pragma unroll
        for (int k = 0; k < NB; ++k) {
          opxs[k] = out_pixel(tid / CPRO + (h * NB + k) * RP);
          xs[k] = *(const u4_t*)(p.bwd_x + ((int64_t)opxs[k] * p.bwd_ldx + och - p.bwd_c0) * 2);
        }
#pragma unroll
        for (int k = 0; k < NB; ++k) {
          const int r = tid / CPRO + (h * NB + k) * RP;
          const u4_t v = *(const u4_t*)((const char*)stg + ((int64_t)r * SLD + oc * 8) * 2);
          const h8_t xv = __builtin_bit_cast(h8_t, xs[k]);
          const h8_t gv = __builtin_bit_cast(h8_t, v);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float xf = (float)xv[e];
            const float dz = (float)gv[e] * (fmaf(xf, bsc[e], bsh[e]) > 0.f ? 1.f : p.bwd_slope);
            bs[e] += dz;
            bsx[e] = fmaf(dz, (xf - bmu[e]) * biv[e], bsx[e]);
          }
          *(u4_t*)(p.out + ((int64_t)opxs[k] * p.ldout + p.coffout + och) * 2) = v;
        }
      }
      copied = true;
    }
  }
  // a thread copies 16 rows (one 16-byte chunk each), eight at a time: all global loads of the eight rows (fused mask, second
  // gradient, BatchNorm input) go out before the first store (as epilogue5)
  constexpr int NR = 8;
#pragma unroll 1
  for (int h = 0; h < (copied ? 0 : BM / (RP * NR)); ++h) {
    int opxs[NR];
    u4_t mk[NR], ad[NR], xs[NR];
#pragma unroll
    for (int k = 0; k < NR; ++k) {
      const int r = tid / CPRO + (h * NR + k) * RP;
      opxs[k] = out_pixel(r) + (DUAL ? (oc >> 3) : 0);
      const int64_t opx = opxs[k];
#ifdef GI_ABLATION
      if (p.mask && (p.dbg_epi & 2)) mk[k] = *(const u4_t*)(p.mask + (((opx * p.ldmask + p.coffmask + och) * 2) & 0xFFF0));   // mask from a 64 KiB window
      else
#endif
      if (BITS && p.mask_bits) {}
      else if (p.mask) mk[k] = *(const u4_t*)(p.mask + (opx * p.ldmask + p.coffmask + och) * 2);
      if (!(BITS && p.mask_bits) && p.mask && p.add) ad[k] = *(const u4_t*)(p.add + (opx * p.ldadd + p.coffadd + och) * 2);
      if (bwd) xs[k] = *(const u4_t*)(p.bwd_x + (opx * p.bwd_ldx + och - p.bwd_c0) * 2);
    }
#pragma unroll
    for (int k = 0; k < NR; ++k) {
      const int r = tid / CPRO + (h * NR + k) * RP;
      const int64_t opx = opxs[k];
      const int64_t o = opx * p.ldout + p.coffout + och;
      u4_t v = *(const u4_t*)((const char*)stg + ((int64_t)r * SLD + oc * 8) * 2);
      if (bwd) {
        const h8_t xv = __builtin_bit_cast(h8_t, xs[k]);
        const h8_t gv = __builtin_bit_cast(h8_t, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float xf = (float)xv[e];
          const float dz = (float)gv[e] * (fmaf(xf, bsc[e], bsh[e]) > 0.f ? 1.f : p.bwd_slope);
          bs[e] += dz;
          bsx[e] = fmaf(dz, (xf - bmu[e]) * biv[e], bsx[e]);
        }
      }
      if (BITS && p.mask_bits) {
      } else if (p.mask) {   // same arithmetic as the separate pass: fp16 value -> fp32 * slope -> fp16
        const h8_t m = __builtin_bit_cast(h8_t, mk[k]);
        h8_t hv = __builtin_bit_cast(h8_t, v);
        if (p.add) {
          const h8_t a8 = __builtin_bit_cast(h8_t, ad[k]);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const bool pos = (float)m[e] > 0.f;
            const float g = (float)hv[e] + (pos ? (float)a8[e] : 0.f);
            hv[e] = (half_t)(pos ? g : g * p.mask_slope);
          }
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) hv[e] = (float)m[e] > 0.f ? hv[e] : (half_t)((float)hv[e] * p.mask_slope);
        }
        v = __builtin_bit_cast(u4_t, hv);
      }
#ifdef GI_ABLATION
      if (p.dbg_epi & 1) { *(u4_t*)(p.out + ((o * 2) & 0xFFF0)) = v; continue; }   // all tiles store into one 64 KiB window
#endif
      *(u4_t*)(p.out + o * 2) = v;
    }
  }
  if constexpr (BITS) {
    if (p.c1w_part) {
      // Weight gradient of the single-channel layer below (IgemmArgs::c1w_*): the tile in LDS is dz1 for 256 x 2 output pixels
      // (Y, X) = (2 (y0 + ty) + py, 2 (x0 + tx) + P) of the layer's 64-channel map. D[tap][c] = A[tap][pixel] * B[pixel][c] on
      // v_mfma_f32_16x16x32_f16 as in c1_wgrad_mfma_kernel (c1.hip): K = 32 consecutive tile rows of one px phase P; A = the image
      // values img[2Y - 1 + ky][2X - 1 + kx] gathered from global memory (lane: tap = lane & 15, its 8 pixels are 4 image columns
      // apart), B = the staged rows read with the transposing ds_read_b64_tr_b16. Wave w takes K blocks w, w + 4, .. of the 16; the
      // four waves' sums are added through LDS and the workgroup stores its 64 x 16 partial (* c1w_scale) for the fixed-order pass.
      __syncthreads();                                   // every row of the tile is final (second gradient written back above)
      const int tap = lane & 15, kq = lane >> 4;
      const int tq = tap >> 2, tp = tap & 3;             // (transposing read: row tq / + 4 of the lane group's 8, columns 4 tp ..)
      f4_t wacc[4];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) wacc[nt] = f4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int bi = 0; bi < 4; ++bi) {
        const int b = wave + 4 * bi;
        const int P = b & 1, m0 = (b >> 1) * 32;
        h8_t af;
#pragma unroll
        for (int j = 0; j < 8; ++j) af[j] = (half_t)c1w_x[bi][j];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
          const char* b_lo = (const char*)stg + ((m0 + 8 * kq + tq) * SLD + P * 64 + nt * 16 + 4 * tp) * 2;
          const char* b_hi = b_lo + 4 * SLD * 2;
          fp16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_t*)b_lo);
          fp16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_t*)b_hi);
          const h4_t l4 = __builtin_bit_cast(h4_t, lo), h4 = __builtin_bit_cast(h4_t, hi);
          const h8_t bf = h8_t{l4[0], l4[1], l4[2], l4[3], h4[0], h4[1], h4[2], h4[3]};
          wacc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, bf, wacc[nt], 0, 0, 0);
        }
      }
      __syncthreads();                                   // the staged tile has been read by every wave
      float* wred = (float*)smem;                        // [4 waves][64 channels][16 taps]
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) wred[wave * 1024 + (nt * 16 + (lane & 15)) * 16 + 4 * (lane >> 4) + r] = wacc[nt][r];
      __syncthreads();
      float* part = p.c1w_part + (int64_t)blockIdx.x * 1024;
      for (int i = tid; i < 1024; i += 256) part[i] = ((wred[i] + wred[1024 + i]) + (wred[2048 + i] + wred[3072 + i])) * p.c1w_scale;
    }
  }
  if (bwd) {
    // lanes oc, oc + 16, oc + 32, oc + 48 of a wave hold the same channels: fold them, then the 4 waves through LDS.
    // (row order per thread differs from epilogue5's - 16 rows of a 4-wave workgroup instead of 8 of an 8-wave one - so
    //  these fp32 sums agree with igemm6's to rounding, not bit for bit; the accumulators they go to are exact)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
#pragma unroll
      for (int off = CPRO; off < 64; off <<= 1) { bs[e] += __shfl_xor(bs[e], off); bsx[e] += __shfl_xor(bsx[e], off); }
    }
    __syncthreads();      // the staged tile has been read by every thread
    float* fold = (float*)smem;   // [4 waves][BN][2]
    if (lane < CPRO) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { fold[(wave * BN + oc * 8 + e) * 2] = bs[e]; fold[(wave * BN + oc * 8 + e) * 2 + 1] = bsx[e]; }
    }
    __syncthreads();
    if (tid < BN) {
      float s = 0.f, q = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) { s += fold[(w * BN + tid) * 2]; q += fold[(w * BN + tid) * 2 + 1]; }
      const int grp = (p.bwd_pg_tiles > 0 && mt_idx >= p.bwd_pg_tiles) ? 1 : 0, rep = (mt_idx + (MODE == 1 ? ph : 0)) & (p.bwd_reps - 1);
      gi_stat_add(p.bwd_acc, p.bwd_c, rep, grp, 0, bn0 + tid, s);
      gi_stat_add(p.bwd_acc, p.bwd_c, rep, grp, 1, bn0 + tid, q);
    }
  }
}

// DBG (builds with -DGI_ABLATION only, WRONG results, timing experiments): 1 no LDS-DMA in the loop, 2 no MFMA, 4 no fragment
// reads, 8 no ReLU on the fragments, 16 no per-step barrier, 32 no epilogue
template <int MODE, bool RELU, int DBG = 0, int BN = 128>
__global__ void __launch_bounds__(256, 2) igemm8_kernel(KP5 p) {
  static_assert(BN == 128 || (BN == 64 && MODE != 3), "64-column tiles: not with both px phases in one tile");
  static_assert(MODE >= 0 && MODE <= 3, "modes 0 (stride-2 gather), 1 (phase), 2 (3x3 / s1), 3 (both px phases)");
  static_assert(!(MODE == 2 && RELU), "the 3x3 mode has no fused input ReLU");
  constexpr bool DUAL = MODE == 3;
  constexpr bool PH = MODE == 1 || MODE == 3;
  constexpr int NQ = MODE == 0 ? 4 : 1;                 // halo groups (parity classes) per channel chunk
  constexpr int NTAP = MODE == 2 ? 9 : 4;               // taps (steps) per halo group
  constexpr int PADX = (DUAL || MODE == 2) ? 2 : 1, PADY = MODE == 2 ? 2 : 1;
  constexpr int BK = 32, NW = 4;
  constexpr int AJ = MODE == 2 ? 6 : 5;                 // halo pieces (16 rows x 64 B) per wave and group
  // MODE 2: the 10 x 34 halo is 340 rows = 22 pieces; waves 2, 3 issue their fifth piece twice so that every wave counts the
  // same number of loads (piece index = j * 4 + wave there, wave * AJ + j in the 4-tap modes)
  constexpr int APIECES = MODE == 2 ? 22 : AJ * NW;
  constexpr int AHEAD = 3;                              // a weight slice is issued 3 steps before its step: ring of 4
  constexpr int A_BYTES = APIECES * 1024;               // 20 (22) KiB per buffer
  constexpr int B_BYTES = BN * 64;                      // 8 KiB per stage
  constexpr int A_OFF = 0, B_OFF = 2 * A_BYTES;         // + 4 ring stages: 72 (76) KiB
  constexpr int BJ = (BN / 16) / NW;                    // weight-slice pieces per wave and step (2; 1 with 64 columns)
  constexpr int MT = 4, NT = BN / 16, NH = NT / 2;      // NH: column tiles per half step
  constexpr int SP = NT == 8 ? 2 : 1;                   // a fragment read behind every SP-th MFMA
  constexpr unsigned OOB = 0x80000000u;                 // beyond any tensor (sizes are checked < 2^31 bytes): reads as zeros
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // ---- XCD-aware tile order (as igemm5 / igemm6)
  const int nyz = p.ntiles * (DUAL ? 2 : (PH ? 4 : 1));
  const int bid = blockIdx.x;
  const int xcd = bid & 7, local = bid >> 3;
  const int mt_idx = (local / nyz) * 8 + xcd;
  if (mt_idx >= p.mtiles) return;
  const int yz = local % nyz;
  const int nt_idx = yz % p.ntiles;
  const int ph = yz / p.ntiles;
  const int py = DUAL ? ph : (ph >> 1), px = DUAL ? 0 : (ph & 1);
  const int n0 = nt_idx * (DUAL ? 64 : BN);
  const int img = mt_idx / p.tiles_per_img, trem = mt_idx % p.tiles_per_img;
  const int y0 = (trem / p.tiles_x) * p.TH, x0 = (trem % p.tiles_x) * p.TW;
  const int HC = p.TW + PADX, HR = p.TH + PADY;
  const int Ktot2 = (PH ? 4 : (MODE == 2 ? 9 : 16)) * p.cin * 2;   // bytes per weight row
  const int64_t phase_bytes = (int64_t)p.cout * Ktot2;
  const int Win = 2 * p.Ws, Hin = 2 * p.Hs;             // mode 0: the large (input) grid

  const int64_t in_bytes = (int64_t)p.n * (MODE == 0 ? 4 : 1) * p.Hs * p.Ws * p.ldin * 2;
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, (int)in_bytes, 0x00020000);
  const char* wbase = p.w + (MODE == 1 ? ph * phase_bytes : (DUAL ? (py * 2) * phase_bytes : 0));
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)wbase, 0, (int)(DUAL ? 2 * phase_bytes : phase_bytes), 0x00020000);

  // ---- per-lane piece offsets (loop invariant): lane l of a piece = row l >> 2, LDS chunk l & 3 = logical chunk (l & 3) ^ s(row)
  const int lrow = lane >> 2, lc = lane & 3;
  unsigned voffA[NQ][AJ];
  auto piece_of = [&](int j) -> int {                    // halo piece index of this wave's j-th piece
    if constexpr (MODE == 2) { const int P = j * NW + wave; return P < APIECES ? P : P - NW; }
    else return wave * AJ + j;
  };
#pragma unroll
  for (int j = 0; j < AJ; ++j) {
    const int r = piece_of(j) * 16 + lrow;
    const int hr = r / HC, hc = r - hr * HC;
    const int cb = (lc ^ ((r >> 1) & 3)) * 16;
    if constexpr (MODE != 0) {
      const int iy = y0 + py - 1 + hr, ix = x0 + (MODE == 1 ? px : 0) - 1 + hc;   // DUAL: columns x0-1 .. x0+TW
      const bool ok = hr < HR && iy >= 0 && iy < p.Hs && ix >= 0 && ix < p.Ws;
      voffA[0][j] = ok ? (unsigned)((((img * p.Hs + iy) * p.Ws + ix) * p.ldin + p.coffin) * 2 + cb) : OOB;
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) {   // halo (hr,hc) of parity class (qy,qx) is input pixel (2*(y0+hr) - qy, 2*(x0+hc) - qx)
        const int iy = 2 * (y0 + hr) - (q >> 1), ix = 2 * (x0 + hc) - (q & 1);
        const bool ok = hr < HR && iy >= 0 && iy < Hin && ix >= 0 && ix < Win;
        voffA[q][j] = ok ? (unsigned)((((img * Hin + iy) * Win + ix) * p.ldin + p.coffin) * 2 + cb) : OOB;
      }
    }
  }
  unsigned voffB[BJ];
#pragma unroll
  for (int j = 0; j < BJ; ++j) {
    const int rb = (wave * BJ + j) * 16 + lrow;                     // row of the B tile (0..127)
    const int cb = (lc ^ ((rb >> 1) & 3)) * 16;
    if constexpr (DUAL) voffB[j] = (unsigned)((rb >> 6) * phase_bytes + (int64_t)(n0 + (rb & 63)) * Ktot2 + cb);   // rows 64.. = px 1
    else voffB[j] = (unsigned)((int64_t)(n0 + rb) * Ktot2 + cb);
  }

  // ---- fragment read offsets (loop invariant) -----------------------------------------------------------------------------
  const int lr = lane & 15, lq = lane >> 4;
  const int lgTW = 31 - __builtin_clz(p.TW);
  // (pixel tile, tap) inside an A buffer; DUAL: px 0. MODE 2 keeps the halo row of tap (0,0) only (36 addresses would not fit
  // the register budget) and derives a tap's address at its read: row + ky * HC + kx, then the chunk swizzle (5 VALU)
  int rdA[MT][MODE == 2 ? 1 : NTAP];
  // DUAL, px 1: its halo rows lie one column to the right, so tap (ty, tx = 1) of px 1 reads the rows of px 0's tap (ty, tx = 0)
  // (rdA[mt][2 ty]); only its tx = 0 taps (column tx_l + 2) need addresses of their own: [pixel tile][ty]
  int rdA2[DUAL ? MT : 1][2];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = wave * 64 + mt * 16 + lr;
    const int ty_l = m >> lgTW, tx_l = m & (p.TW - 1);
    if constexpr (MODE == 2) {
      rdA[mt][0] = ty_l * HC + tx_l;               // a ROW here, not a byte address
    } else {
#pragma unroll
      for (int tap = 0; tap < NTAP; ++tap) {
        const int R = PH ? (ty_l + 1 - (tap >> 1)) * HC + (tx_l + 1 - (tap & 1)) : (ty_l + (tap >> 1)) * HC + (tx_l + (tap & 1));
        rdA[mt][tap] = A_OFF + R * 64 + ((lq ^ ((R >> 1) & 3)) << 4);
        if constexpr (DUAL) {
          if ((tap & 1) == 0) rdA2[mt][tap >> 1] = A_OFF + (R + 1) * 64 + ((lq ^ (((R + 1) >> 1) & 3)) << 4);
        }
      }
    }
  }
  auto addrA = [&](int mt, auto TAP) -> int {       // byte address of (pixel tile, tap) inside an A buffer
    constexpr int tap = decltype(TAP)::value;
    if constexpr (MODE == 2) {
      const int R = rdA[mt][0] + (tap / 3) * HC + (tap % 3);
      return A_OFF + R * 64 + ((lq ^ ((R >> 1) & 3)) << 4);
    } else {
      return rdA[mt][tap];
    }
  };
  auto rd_px1 = [&](int mt, auto TAP) -> int {      // address of px 1's pixel fragment (DUAL)
    constexpr int tap = decltype(TAP)::value;
    if constexpr ((tap & 1) != 0) return rdA[mt][tap & ~1];
    else return rdA2[mt][tap >> 1];
  };
  const int rdB = B_OFF + lr * 64 + ((lq ^ ((lr >> 1) & 3)) << 4);   // + column tile * 1024 + stage * B_BYTES

  f4_t acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f4_t{0.f, 0.f, 0.f, 0.f};

  const int nchunk = p.cin / BK;
  const int ngroups = nchunk * NQ, nsteps = ngroups * NTAP;
  const int relu_cend = p.relu_in ? p.relu_cend : 0;

  // RELU kernels: the decoder's ReLU on a pixel fragment right before its first use. rmin = 0 for the skip half of a concat
  // buffer, 0x8000 (the smallest 16-bit integer: no change) for the rest - no branch either way
  auto relu_a = [&](h8_t& v, int rmin) {
    if constexpr (RELU && (DBG & 8) == 0) {
      typedef short s8_t __attribute__((ext_vector_type(8)));
      const short m = (short)rmin;
      const s8_t lo = {m, m, m, m, m, m, m, m};
      v = __builtin_bit_cast(h8_t, __builtin_elementwise_max(__builtin_bit_cast(s8_t, v), lo));
    }
  };
  auto rmin_of = [&](int chunk) -> int { return chunk * BK < relu_cend ? 0 : -32768; };
  auto ldsr = [&](int off) -> h8_t {          // one fragment read
    if constexpr ((DBG & 4) != 0) { h8_t v = {1, 1, 1, 1, 1, 1, 1, 1}; asm volatile("" : "+v"(v)); return v; }
    else return *(const h8_t*)(smem + off);
  };

  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  using I3 = std::integral_constant<int, 3>;

  // weight slice of (chunk, tap) into ring stage `stage` (4-tap modes: stage = tap)
  auto issue_b = [&](unsigned (&vb)[BJ], int chunk, auto Q, auto TAP, auto J, int stage) {
    constexpr int q = decltype(Q)::value, tap = decltype(TAP)::value, j = decltype(J)::value;
    int koff;
    if constexpr (MODE != 0) {
      koff = tap * p.cin + chunk * BK;
    } else {
      constexpr int ky = (q >> 1) ? 2 * (tap >> 1) : 2 * (tap >> 1) + 1, kx = (q & 1) ? 2 * (tap & 1) : 2 * (tap & 1) + 1;
      koff = (ky * 4 + kx) * p.cin + chunk * BK;
    }
    blds16(rsB, vb[j], koff * 2, smem + B_OFF + stage * B_BYTES + (wave * BJ + j) * 1024);
  };

  // ---- prologue: halo of group 0, weight slices of steps 0 .. AHEAD - 1 (nsteps >= 4) ---------------------------------------
  static_for<AJ>([&](auto J) { blds16(rsA, voffA[0][decltype(J)::value], 0, smem + A_OFF + piece_of(decltype(J)::value) * 1024); });
  static_for<BJ>([&](auto J) { issue_b(voffB, 0, I0{}, I0{}, J, 0); });
  static_for<BJ>([&](auto J) { issue_b(voffB, 0, I0{}, I1{}, J, 1); });
  static_for<BJ>([&](auto J) { issue_b(voffB, 0, I0{}, I2{}, J, 2); });
  // fragment registers: two pixel-fragment sets and the eight column fragments
  //   not DUAL: fa[tap & 1] = this step's pixels (both halves), fa[(tap & 1) ^ 1] receives the next step's during H1
  //   DUAL:     fa[0] = px 0 rows (H0; refilled with the next step's during H1), fa[1] = px 1 rows (read during H0, used in H1)
  h8_t fa[2][MT], fb[NT];
  if constexpr (BJ == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");     // all but the last slice issued (BJ pieces)
  else asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
#pragma unroll
  for (int i = 0; i < MT; ++i) fa[0][i] = *(const h8_t*)(smem + addrA(i, I0{}));
#pragma unroll
  for (int i = 0; i < NH; ++i) fb[i] = *(const h8_t*)(smem + rdB + i * 1024);

  // One halo group = four steps of straight-line code (igemm6's structure): every step issues the same number of LDS-DMA pieces
  // (beyond the end of the K loop with out-of-range offsets), so the counted waits are immediates. GI = group index modulo 2
  // (modes 1, 3) or 4 (mode 0): halo buffer GI & 1, parity class GI (mode 0).
  // c = sequential group index; Q / QN = parity class of this group / of the next one (MODE 0), chunk / chunk_n their channel chunks
  auto group = [&](int c, auto GIc, auto Qc, auto QNc, int chunk, int chunk_n) {
    constexpr int GI = decltype(GIc)::value;
    constexpr int BUF = GI & 1, Q = decltype(Qc)::value, QN = decltype(QNc)::value;
    constexpr int PAR0 = (MODE == 2) ? (GI & 1) : 0;             // nine taps per group: the fragment-set parity alternates per group
    using QT = std::integral_constant<int, Q>;
    using QNT = std::integral_constant<int, QN>;
    const int rmin = rmin_of(chunk);
    const bool next_a = c + 1 < ngroups;
    unsigned va[AJ];                                            // the next group's halo pieces (none after the last group)
#pragma unroll
    for (int j = 0; j < AJ; ++j) va[j] = next_a ? voffA[QN][j] : OOB;
    static_for<NTAP>([&](auto TAPc) {
      constexpr int tap = decltype(TAPc)::value;
      const int s = c * NTAP + tap;
      // in flight may stay: the pieces issued during step s-1 = a weight slice (BJ) + its halo pieces (3 in tap 0, 2 in tap 1;
      // MODE 2: one in each of taps 0..5)
      constexpr int ptap = (tap + NTAP - 1) % NTAP;
      constexpr int nwait = BJ + (MODE == 2 ? (ptap < AJ ? 1 : 0) : (ptap == 0 ? 3 : (ptap == 1 ? 2 : 0)));
      if constexpr (nwait == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
      else if constexpr (nwait == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      else if constexpr (nwait == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
      else if constexpr (nwait == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
      if constexpr ((DBG & 16) == 0) __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      // weight slice of step s + 3: 4-tap modes (group c, tap 3) during tap 0, else (group c + 1, tap - 1); MODE 2 (chunk c,
      // tap + 3) or (chunk c + 1, tap - 6); out of range past the end. Ring stage of a step = step & 3: a constant per tap in the
      // 4-tap modes, a scalar in MODE 2 (nine taps per group)
      const int stg = MODE == 2 ? (s & 3) : tap, nstg = MODE == 2 ? ((s + 1) & 3) : ((tap + 1) & 3), istg = MODE == 2 ? ((s + 3) & 3) : ((tap + 3) & 3);
      unsigned vb[BJ];
#pragma unroll
      for (int j = 0; j < BJ; ++j) vb[j] = s + AHEAD < nsteps ? voffB[j] : OOB;
      auto issue_piece = [&](auto IDX) {
        constexpr int idx = decltype(IDX)::value;      // MFMA counter of this step, 0 .. 31
        constexpr int NM = MT * NT;
        if constexpr ((DBG & 1) != 0) return;
        if constexpr (MODE == 2) {     // halo of chunk c+1: piece `tap` during taps 0..5
          if constexpr (tap < AJ && idx == NM / 4) blds16(rsA, va[tap], chunk_n * (BK * 2), smem + A_OFF + (BUF ^ 1) * A_BYTES + piece_of(tap) * 1024);
        } else if constexpr (tap == 0) {      // halo of group c+1: pieces 0,1,2 during tap 0, pieces 3,4 during tap 1
          if constexpr (idx == NM / 4) blds16(rsA, va[0], chunk_n * (BK * 2), smem + A_OFF + (BUF ^ 1) * A_BYTES + (wave * AJ + 0) * 1024);
          if constexpr (idx == NM / 4 + 2) blds16(rsA, va[1], chunk_n * (BK * 2), smem + A_OFF + (BUF ^ 1) * A_BYTES + (wave * AJ + 1) * 1024);
          if constexpr (idx == NM / 4 + 4) blds16(rsA, va[2], chunk_n * (BK * 2), smem + A_OFF + (BUF ^ 1) * A_BYTES + (wave * AJ + 2) * 1024);
        } else if constexpr (tap == 1) {
          if constexpr (idx == NM / 4) blds16(rsA, va[3], chunk_n * (BK * 2), smem + A_OFF + (BUF ^ 1) * A_BYTES + (wave * AJ + 3) * 1024);
          if constexpr (idx == NM / 4 + 3) blds16(rsA, va[4], chunk_n * (BK * 2), smem + A_OFF + (BUF ^ 1) * A_BYTES + (wave * AJ + 4) * 1024);
        }
        static_for<BJ>([&](auto Jc) {
          constexpr int j = decltype(Jc)::value;
          if constexpr (idx == NM - 2 - (BJ - 1 - j) * 2) {
            if constexpr (MODE == 2) {
              if constexpr (tap + AHEAD < NTAP) issue_b(vb, chunk, QT{}, std::integral_constant<int, tap + AHEAD>{}, Jc, istg);
              else issue_b(vb, chunk_n, QNT{}, std::integral_constant<int, tap + AHEAD - NTAP>{}, Jc, istg);
            } else {
              if constexpr (tap == 0) issue_b(vb, chunk, QT{}, I3{}, Jc, istg);
              else issue_b(vb, chunk_n, QNT{}, std::integral_constant<int, tap - 1>{}, Jc, istg);
            }
          }
        });
      };
      constexpr int CUR = DUAL ? 0 : ((PAR0 + tap) & 1);   // pixel fragments of H0 (not DUAL: of the whole step)
      const int rdBs = rdB + stg * B_BYTES, rdBn = rdB + nstg * B_BYTES;   // column fragments of this / the next step
      constexpr int OTH = DUAL ? 1 : (CUR ^ 1);        // DUAL: px 1 rows of this step; else: the next step's set
      // ---- H0: column tiles 0..3. One fragment read behind every second MFMA:
      //   not DUAL: b[4..7] of this step;  DUAL: a1[0], b[4..7], a1[1..3] of this step
      static_for<MT * NH>([&](auto IDX) {
        constexpr int idx = decltype(IDX)::value, mt = idx / NH, nt = idx % NH;
        if constexpr (nt == 0) relu_a(fa[CUR][mt], rmin);
        if constexpr ((DBG & 2) != 0) { asm volatile("" :: "v"(fa[CUR][mt])); asm volatile("" :: "v"(fb[nt])); }
        else acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[nt], fa[CUR][mt], acc[mt][nt], 0, 0, 0);   // D^T: rows = channels
        if constexpr (idx % SP == 0) {
          constexpr int k = idx / SP;                  // read slot 0 .. 7
          if constexpr (!DUAL) {
            if constexpr (k < NH) fb[NH + k] = ldsr(rdBs + (NH + k) * 1024);
          } else {
            if constexpr (k == 0) fa[1][0] = ldsr(rd_px1(0, TAPc) + BUF * A_BYTES);
            else if constexpr (k <= NH) fb[NH + k - 1] = ldsr(rdBs + (NH + k - 1) * 1024);
            else fa[1][k - NH] = ldsr(rd_px1(k - NH, TAPc) + BUF * A_BYTES);
          }
        }
        issue_piece(IDX);
        __builtin_amdgcn_sched_barrier(0);
      });
      // ---- H1: column tiles 4..7; behind every second MFMA one fragment of the NEXT step (a[0], b[0..3], a[1..3]); that
      //      step's data landed one step early (after the last step: stale bytes, never used)
      static_for<MT * NH>([&](auto IDX) {
        constexpr int idx = decltype(IDX)::value, mt = idx / NH, nt = NH + idx % NH;
        constexpr int HS = DUAL ? 1 : CUR;             // pixel fragments of H1
        if constexpr (DUAL && nt == NH) relu_a(fa[1][mt], rmin);
        if constexpr ((DBG & 2) != 0) { asm volatile("" :: "v"(fa[HS][mt])); asm volatile("" :: "v"(fb[nt])); }
        else acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[nt], fa[HS][mt], acc[mt][nt], 0, 0, 0);
        if constexpr (idx % SP == 0) {
          constexpr int k = idx / SP;
          constexpr int NXT = DUAL ? 0 : OTH;          // DUAL: fa[0] is free after H0
          constexpr int ntap = (tap + 1) % NTAP;
          constexpr int nbuf = tap < NTAP - 1 ? BUF : (BUF ^ 1);
          using NT_ = std::integral_constant<int, ntap>;
          // (not DUAL: fa[CUR] stays live through H1, the next step's pixels go to the other set;
          //  b[0..3] are free after H0)
          if constexpr (k == 0) fa[NXT][0] = ldsr(addrA(0, NT_{}) + nbuf * A_BYTES);
          else if constexpr (k <= NH) fb[k - 1] = ldsr(rdBn + (k - 1) * 1024);
          else if constexpr (k < NH + MT) fa[NXT][k - NH] = ldsr(addrA(k - NH, NT_{}) + nbuf * A_BYTES);
        }
        issue_piece(std::integral_constant<int, idx + MT * NH>{});
        __builtin_amdgcn_sched_barrier(0);
      });
    });
  };

  if constexpr (MODE == 0) {
    // Group order: per PAIR of 32-channel chunks (one 128-byte line of a pixel row with 64-channel granularity) the four parity classes,
    // each with both chunks back to back. A 64-byte request makes the memory side fetch the whole 128-byte line (measured:
    // tools/micro/stream_bw.hip, half-line reads run at half the useful bandwidth); with the chunk-major order of the first version the
    // line's other half was requested 16 steps later, after the line had left L2 - the critic's conv2 read its input 2.2 times.
    for (int cp = 0; 2 * cp < nchunk; ++cp) {     // nchunk is even (cin % 64 == 0, checked by the host)
      const int c = cp * 8, a = 2 * cp, b = 2 * cp + 1;
      group(c + 0, I0{}, I0{}, I0{}, a, b); group(c + 1, I1{}, I0{}, I1{}, b, a);
      group(c + 2, I0{}, I1{}, I1{}, a, b); group(c + 3, I1{}, I1{}, I2{}, b, a);
      group(c + 4, I0{}, I2{}, I2{}, a, b); group(c + 5, I1{}, I2{}, I3{}, b, a);
      group(c + 6, I0{}, I3{}, I3{}, a, b); group(c + 7, I1{}, I3{}, I0{}, b, a + 2);
    }
  } else {
    int c = 0;
    for (; c + 1 < ngroups; c += 2) { group(c, I0{}, I0{}, I0{}, c, c + 1); group(c + 1, I1{}, I0{}, I0{}, c + 1, c + 2); }
    if (c < ngroups) group(c, I0{}, I0{}, I0{}, c, c + 1);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  asm volatile("" ::: "memory");
  if constexpr ((DBG & 32) != 0) {   // no epilogue: keep the accumulators alive with a store that never happens
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) t += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (t == 12345.678f) p.out[0] = 1;
    return;
  }
  epilogue8<MODE, BN, (MODE == 3 && !RELU)>(p, acc, smem, tid, lane, wave, mt_idx, ph, py, px, n0, img, y0, x0, lgTW);
}

}  // namespace

// Launched by op_igemm5 (igemm5.hip) with the kernel arguments it has prepared; grid as igemm6's.
int op_igemm8_launch(hipStream_t st, int mode, bool dual, bool relu, int grid, const KP5& kp, int bn) {
  if (mode == 2 && bn == 64) {      // VGG conv1_2: 64 output channels
    // (64-column tiles were also measured on the 4-tap layers whose 128-column grid is one workgroup per CU - d3, u4, critic conv4:
    //  twice the workgroups, two per CU - and ran 6 - 10 % SLOWER than igemm6 there: twice the halo DMA per MAC; not instantiated)
    static GiDevOnce attr64;
    const int lds64 = 2 * 22528 + 4 * 4096;
    if (attr64.first()) { GI_HIP(hipFuncSetAttribute((const void*)igemm8_kernel<2, false, 0, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, lds64)); }
    hipLaunchKernelGGL((igemm8_kernel<2, false, 0, 64>), dim3(grid), dim3(256), lds64, st, kp);
    gi_note_kernel("igemm8<2,64>");
    GI_LAUNCH_CHECK();
    return GI_OK;
  }
  const int LDS = (mode == 2 ? 2 * 22528 : 2 * 20480) + 4 * 8192;   // 4-tap modes: = the epilogue's 256 x 136 halves + 4 x 128 x 2 floats
  static GiDevOnce attr[8];
  const int v = mode == 2 ? 6 : (dual ? 2 : mode) * 2 + (relu ? 1 : 0);
#define GI_K8(MODE_, RELU_, NAME_) do { \
    if (attr[v].first()) { GI_HIP(hipFuncSetAttribute((const void*)igemm8_kernel<MODE_, RELU_>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS)); } \
    hipLaunchKernelGGL((igemm8_kernel<MODE_, RELU_>), dim3(grid), dim3(256), LDS, st, kp); gi_note_kernel(NAME_); } while (0)
#ifdef GI_ABLATION   // timing-only ablation kernels compute wrong results: compiled only with `build.sh -DGI_ABLATION`
  { const char* e = getenv("GI_IGEMM8_DBG"); const int dbg = e ? atoi(e) : 0;
    if (dbg && v == 3) {
#define GI_K8D(D_) do { GI_HIP(hipFuncSetAttribute((const void*)igemm8_kernel<1, true, D_>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS)); \
      hipLaunchKernelGGL((igemm8_kernel<1, true, D_>), dim3(grid), dim3(256), LDS, st, kp); } while (0)
      switch (dbg) {
        case 1: GI_K8D(1); break; case 2: GI_K8D(2); break; case 4: GI_K8D(4); break; case 8: GI_K8D(8); break; case 16: GI_K8D(16); break;
        case 32: GI_K8D(32); break; case 5: GI_K8D(5); break; case 7: GI_K8D(7); break; case 6: GI_K8D(6); break; case 3: GI_K8D(3); break;
        case 39: GI_K8D(39); break; case 13: GI_K8D(13); break; default: GI_K8D(15); break;
      }
#undef GI_K8D
      GI_LAUNCH_CHECK();
      return GI_OK;
    }
    // the gather mode (stride-2 convolution: the critic's conv2, the generator's d2), GI_IGEMM8_DBG0
    const char* e0 = getenv("GI_IGEMM8_DBG0"); const int dbg0 = e0 ? atoi(e0) : 0;
    if (dbg0 && v == 0) {
#define GI_K8D0(D_) do { GI_HIP(hipFuncSetAttribute((const void*)igemm8_kernel<0, false, D_>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS)); \
      hipLaunchKernelGGL((igemm8_kernel<0, false, D_>), dim3(grid), dim3(256), LDS, st, kp); } while (0)
      switch (dbg0) {
        case 1: GI_K8D0(1); break; case 2: GI_K8D0(2); break; case 4: GI_K8D0(4); break; case 16: GI_K8D0(16); break; case 32: GI_K8D0(32); break;
        case 5: GI_K8D0(5); break; case 7: GI_K8D0(7); break; default: GI_K8D0(39); break;
      }
#undef GI_K8D0
      GI_LAUNCH_CHECK();
      return GI_OK;
    } }
#endif
  switch (v) {
    case 0: GI_K8(0, false, "igemm8<0>"); break;
    case 1: GI_K8(0, true, "igemm8<0,relu>"); break;
    case 2: GI_K8(1, false, "igemm8<1>"); break;
    case 3: GI_K8(1, true, "igemm8<1,relu>"); break;
    case 4: GI_K8(3, false, "igemm8<3>"); break;
    case 5: GI_K8(3, true, "igemm8<3,relu>"); break;
    default: GI_K8(2, false, "igemm8<2>"); break;
  }
#undef GI_K8
  GI_LAUNCH_CHECK();
  return GI_OK;
}
