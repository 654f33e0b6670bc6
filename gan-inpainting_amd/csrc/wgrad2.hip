// Weight-gradient GEMM of the 4x4 / stride-2 / pad-1 family, second generation (gfx950, fp16):
//
//   dW[a][ky][kx][b] += scale * sum_{n,y,x} S[n,y,x,a] * L[n,2y-1+ky,2x-1+kx,b]
//
// wgrad.hip gathers the L operand tap by tap through registers (32 KiB of loads per 128x128x64 MAC step). Here a
// workgroup owns 128 S channels x 64 L channels x the FOUR kx taps of one ky (a 128 x 256 tile of dW, 8 waves =
// 2 a-halves x 4 taps, 64 x 64 each). The K loop walks pixel tiles of 4 rows x 16 pixels of the small grid; per
// tile the S pixels (64 x 128 channels) and ONE halo of the large tensor - for every tile row the 34 pixels
// 2*x0-1 .. 2*x0+32 of row 2y-1+ky - arrive by LDS-DMA (three-stage ring, 5 pieces per wave) and all four taps
// read it with a stride of two rows: tap kx of pixel (ry,rx) is halo row ry*34 + 2*rx + kx. 33 KiB of loads per
// 128x256x64 MAC step, i.e. half of wgrad.hip per MAC, no register staging, no ds_write.
// Both operands are pixel-major (K-major): fragments come from ds_read_b64_tr_b16. LDS rows are 128 B with the
// 16-byte chunk c of row r at physical chunk c ^ (r & 7) (applied to the DMA source addresses): the four rows of
// a transposing read (consecutive for S, stride 2 for the halo) then touch eight distinct chunks.
// Pixel ranges are split over workgroups as in wgrad.hip: partial tiles to scratch + fixed-order reduction, or
// direct / atomic accumulation.
// NKY = 2 (default): a workgroup owns the taps ky and ky + 2 (same row parity): the halo grows from 4 to 5 rows (rows of
// 36 pixels so that the second tap set sits at a constant LDS distance, XOR 64 in the swizzle), each wave multiplies its
// S fragments with both tap sets (64 x 128 per wave, 128 accumulator registers). 38 KiB of loads per 128 x 512 x 64 MAC
// step instead of 33 KiB per 128 x 256 x 64: the kernel is bound by L2 -> LDS traffic (553 MB per launch on the critic's
// layers at 64 images, 4.9 TB/s), which this cuts to 312 MB.
#include <stdlib.h>

#include "common.h"

const char* gi_igemm3_zero_page(int dev);   // igemm3.hip

namespace {

struct WP2 {
  const char* S;
  const char* L;
  const char* zero;
  float* dW;
  float* part;        // non-null: split ks writes part[ks][ca][16*cb]
  int direct;         // single split: dW += acc without atomics
  int n, Hs, Ws, lgWs, lgHs;
  int ca, ldS, coffS;
  int cb, ldL, coffL;
  int relu_S;
  float scale;
  int ntile;          // pixel tiles = n * (Hs/4) * (Ws/16)
  int tiles_per_split;
};

__device__ __forceinline__ void glds16w(const char* src, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
__device__ __forceinline__ h8_t tr16x2(const char* lo_p, const char* hi_p) {
  fp16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_t*)lo_p);
  fp16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_t*)hi_p);
  h4_t l4 = __builtin_bit_cast(h4_t, lo), h4 = __builtin_bit_cast(h4_t, hi);
  return h8_t{l4[0], l4[1], l4[2], l4[3], h4[0], h4[1], h4[2], h4[3]};
}
__device__ __forceinline__ h8_t relu8(h8_t v) {
  typedef short s8_t __attribute__((ext_vector_type(8)));
  s8_t h = __builtin_bit_cast(s8_t, v);
  const s8_t z = {0, 0, 0, 0, 0, 0, 0, 0};
  h = __builtin_elementwise_max(h, z);
  return __builtin_bit_cast(h8_t, h);
}

// grid.x = (ca/128) * (cb/64) * (4 / NKY) ; grid.y = splits
template <int NKY>
__global__ void __launch_bounds__(512, NKY == 1 ? 2 : 1) wgrad2_kernel(WP2 p) {
  constexpr int HST = NKY == 2 ? 36 : 34;    // halo row stride in pixels
  constexpr int HROWS = NKY == 2 ? 5 : 4;
  constexpr int S_BYTES = 2 * 64 * 128;      // two 64-channel segments x 64 pixels x 128 B
  constexpr int L_ROWS = 192, L_BYTES = L_ROWS * 128;   // 4 x 34 = 136 halo rows used, 24 pieces of 8 rows
  constexpr int STAGE = S_BYTES + L_BYTES;   // 40 KiB
  constexpr int SJ = 2, LJ = 3;              // pieces per wave and tile
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wa = wave >> 2, kx = wave & 3;

  const int nbt = p.cb >> 6;
  int bx = blockIdx.x;
  const int ky = bx & (4 / NKY - 1); bx /= (4 / NKY);   // NKY = 2: taps ky and ky + 2
  const int b0 = (bx % nbt) * 64, a0 = (bx / nbt) * 128;
  const int ks = blockIdx.y;
  const int t0 = ks * p.tiles_per_split, t1 = min(p.ntile, t0 + p.tiles_per_split);
  const int HL = 2 * p.Hs, WL = 2 * p.Ws;
  const int txs = p.Ws >> 4, tys = p.Hs >> 2;          // pixel tiles per row / column of an image (powers of two)
  const int lg_txs = p.lgWs - 4, lg_tys = p.lgHs - 2;

  // ---- per-lane DMA source descriptions ------------------------------------------------------------------
  const int lrow = lane >> 3;
  const int lchunk = (lane & 7) ^ (lrow & 7);
  int64_t s_off[SJ];          // byte offset of this lane's S row relative to the tile's first pixel
  int64_t l_off[LJ];          // byte offset of this lane's halo row relative to the halo's first pixel
  int l_hr[LJ], l_hx[LJ];     // halo coordinates (hr >= 4: padding rows of the 192-row buffer)
#pragma unroll
  for (int j = 0; j < SJ; ++j) {
    const int ps = wave * SJ + j;                        // 0..15: segment ps >> 3, pixel rows (ps & 7) * 8 ..
    const int seg = ps >> 3, k = (ps & 7) * 8 + lrow;
    const int ry = k >> 4, rx = k & 15;
    s_off[j] = ((int64_t)(ry * p.Ws + rx) * p.ldS + p.coffS + a0 + seg * 64) * 2 + lchunk * 16;
  }
#pragma unroll
  for (int j = 0; j < LJ; ++j) {
    const int hrow = (wave * LJ + j) * 8 + lrow;         // 0..191
    const int hr = hrow / HST, hx = hrow - hr * HST;
    l_hr[j] = hr; l_hx[j] = hx;
    l_off[j] = ((int64_t)(2 * hr * WL + hx) * p.ldL + p.coffL + b0) * 2 + lchunk * 16;
  }
  // LDS-DMA pieces of pixel tile t into ring stage `stage`: pieces 0..SJ-1 = S rows, SJ..SJ+LJ-1 = halo rows
  const char* sbase = nullptr;
  const char* lbase = nullptr;
  int Y0 = 0, X0 = 0;
  auto tile_bases = [&](int t) {
    // tile t -> image n, tile row yb, tile column xb (all powers of two)
    const int xb = t & (txs - 1), yb = (t >> lg_txs) & (tys - 1), nn = t >> (lg_txs + lg_tys);
    const int y0 = yb * 4, x0 = xb * 16;
    sbase = p.S + (int64_t)((nn * p.Hs + y0) * p.Ws + x0) * p.ldS * 2;
    Y0 = 2 * y0 - 1 + ky; X0 = 2 * x0 - 1;               // first halo pixel (may lie outside)
    lbase = p.L + ((int64_t)(nn * HL + Y0) * WL + X0) * p.ldL * 2;
  };
  auto issue_piece = [&](int stage, auto PIECE) {        // after tile_bases(t)
    constexpr int j = decltype(PIECE)::value;
    char* dst = smem + stage * STAGE;
    if constexpr (j < SJ) {
      glds16w(sbase + s_off[j], dst + (wave * SJ + j) * 1024);
    } else {
      constexpr int jl = j - SJ;
      const int Y = Y0 + 2 * l_hr[jl], X = X0 + l_hx[jl];
      const bool ok = l_hr[jl] < HROWS && l_hx[jl] < 34 && Y >= 0 && Y < HL && X >= 0 && X < WL;
      glds16w(ok ? lbase + l_off[jl] : p.zero + lchunk * 16, dst + S_BYTES + (wave * LJ + jl) * 1024);
    }
  };
  auto issue = [&](int t, int stage) {
    tile_bases(t);
    static_for<SJ + LJ>([&](auto J) { issue_piece(stage, J); });
  };

  // ---- fragment read offsets (per lane, tile independent) -------------------------------------------------
  // ds_read_b64_tr_b16: per 16-lane group a 4(k) x 16(channel) block; lane 4q+pp supplies the address of block row
  // q, channels 4pp..4pp+3. lo / hi = k rows 8g+q / 8g+4+q of the 32-pixel k-step.
  const int g = lane >> 4, i16 = lane & 15, q = i16 >> 2, pp = i16 & 3;
  int aoff[2][2][4];   // [kstep][lo/hi][mt] byte offsets into the S part of a stage
  int boff[2][2][4];   // [kstep][lo/hi][nt] byte offsets into the L part
#pragma unroll
  for (int ksx = 0; ksx < 2; ++ksx)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int k = ksx * 32 + 8 * g + 4 * h + q;        // pixel of the tile
      const int srow = wa * 64 + k;                      // S row: segment wa (channels wa*64 ..), pixel k
      const int ry = k >> 4, rx = k & 15;
      const int lrw = ry * HST + 2 * rx + kx;            // halo row of tap kx
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int chunk = t * 2 + (pp >> 1);             // 16 channels of tile t start at byte t*32; this lane's 4 at + pp*8
        aoff[ksx][h][t] = srow * 128 + ((chunk ^ (srow & 7)) << 4) + (pp & 1) * 8;
        boff[ksx][h][t] = S_BYTES + lrw * 128 + ((chunk ^ (lrw & 7)) << 4) + (pp & 1) * 8;
      }
    }

  f4_t acc[NKY][4][4];
#pragma unroll
  for (int s2 = 0; s2 < NKY; ++s2)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[s2][i][j] = f4_t{0.f, 0.f, 0.f, 0.f};

  const int nt_tiles = t1 - t0;
  if (nt_tiles > 0) {
    issue(t0, 0);
    if (nt_tiles > 1) issue(t0 + 1, 1);
    int stage = 0;
    for (int i = 0; i < nt_tiles; ++i) {
      if (i + 1 < nt_tiles) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      const char* sb = smem + stage * STAGE;
      int st2 = stage + 2;
      if (st2 >= 3) st2 -= 3;
      const bool more = i + 2 < nt_tiles;
      if (more) tile_bases(t0 + i + 2);
#pragma unroll
      for (int ksx = 0; ksx < 2; ++ksx) {
        h8_t af[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) af[t] = tr16x2(sb + aoff[ksx][0][t], sb + aoff[ksx][1][t]);
        if (p.relu_S) {
#pragma unroll
          for (int t = 0; t < 4; ++t) af[t] = relu8(af[t]);
        }
#pragma unroll
        for (int s2 = 0; s2 < NKY; ++s2) {
          h8_t bf[4];
          // tap ky + 2 of pixel row ry is halo row ry + 1: 36 LDS rows further, i.e. (row & 7) ^ 4 in the swizzle
#pragma unroll
          for (int t = 0; t < 4; ++t)
            bf[t] = s2 ? tr16x2(sb + (boff[ksx][0][t] ^ 64) + HST * 128, sb + (boff[ksx][1][t] ^ 64) + HST * 128)
                       : tr16x2(sb + boff[ksx][0][t], sb + boff[ksx][1][t]);
#pragma unroll
          for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
              acc[s2][mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[mt], bf[nt], acc[s2][mt][nt], 0, 0, 0);
              // the five pieces of the next-but-one tile, spread over the step's MFMAs
              const int idx = (ksx * NKY + s2) * 16 + mt * 4 + nt;
              if (more) {
                static_for<SJ + LJ>([&](auto Q) {
                  constexpr int qq = decltype(Q)::value;
                  if (idx == NKY * (3 + 6 * qq)) issue_piece(st2, Q);
                });
              }
            }
        }
      }
      ++stage;
      if (stage == 3) stage = 0;
    }
  }

  // ---- epilogue: acc[mt][nt][r] = dW[a0 + wa*64 + mt*16 + (lane>>4)*4 + r][(ky*4+kx)*cb + b0 + nt*16 + (lane&15)]
  const int64_t ldw = (int64_t)16 * p.cb;
#pragma unroll
  for (int s2 = 0; s2 < NKY; ++s2) {
    const int colb = ((ky + 2 * s2) * 4 + kx) * p.cb + b0 + (lane & 15);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = a0 + wa * 64 + mt * 16 + (lane >> 4) * 4 + r;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
          const int64_t o = (int64_t)row * ldw + colb + nt * 16;
          const float v = acc[s2][mt][nt][r] * p.scale;
          if (p.part) p.part[(int64_t)ks * p.ca * ldw + o] = v;
          else if (p.direct) p.dW[o] += v;
          else atomicAdd(p.dW + o, v);
        }
      }
  }
}

// wgrad3: the two-ky tile of wgrad2_kernel<2> with the issue stream of igemm6 (round 2):
//   * LDS-DMA through buffer descriptors (32-bit per-lane offsets, out-of-range offset = hardware zero fill: no zero
//     page, no 64-bit address arithmetic per piece);
//   * ring stages laid out as [S0 S1 S2 | L0 L1 L2] so that every fragment read is `ds_read_b64_tr_b16 v, vaddr offset:imm`
//     with loop-invariant vaddr (the stage and the second tap set's +36 rows are immediates < 64 KiB);
//   * both tiles in flight have landed at a step's barrier (a tile is issued two steps ahead: a whole step of slack), so the
//     fragments of the NEXT tile's first block are read during the current tile's last block: no read burst behind the
//     barrier; inside a tile the B fragments of block i+1 are read during block i (double buffer), the A fragments of the
//     second k-half replace the first ones in place as their last MFMA has issued; one read pair every other MFMA,
//     `sched_barrier` after every MFMA keeps the order.
// The LDS-DMA piece is issued from inline assembly: behind the builtin the compiler's wait-count pass puts `s_waitcnt vmcnt(0)`
// in front of EVERY `ds_read_b64_tr_b16` that follows (the transposing-read intrinsic carries no alias information, so each read
// "may alias" the pending LDS write; plain `ds_read_b128` of igemm6 is not affected) - 133 such waits in wgrad2_kernel<2>, i.e.
// no DMA / MFMA overlap at all. The kernels order DMA and reads themselves (counted vmcnt + barrier per tile). The compiler
// knows of no outstanding vector-memory operation of its own inside the K loop, and where it does (prologue, epilogue) it can
// only over-wait.
typedef int rsrc4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ rsrc4_t make_rsrc4(const void* base, unsigned bytes) {     // raw buffer: stride 0, 32-bit offsets, zeros out of range
  const uint64_t b = (uint64_t)(uintptr_t)base;
  return rsrc4_t{(int)(unsigned)b, (int)(unsigned)(b >> 32) & 0xFFFF, (int)bytes, 0x00020000};
}
__device__ __forceinline__ void blds16w(rsrc4_t r4, unsigned voff, int soff, char* lds_wave_base) {
  const unsigned lds = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds_wave_base);
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" : : "s"(lds), "v"(voff), "s"(r4), "s"(soff) : "memory", "m0");
}

// LGTW: the pixel tile is 64 / TW rows x TW = 2^LGTW pixels: 4 x 16 (small grids at least 16 wide) or 8 x 8 (8 x 8 maps: the generator's
// level 5 at 256 x 256)
template <int LGTW>
__global__ void __launch_bounds__(512, 1) wgrad3_kernel(WP2 p) {
  constexpr int TW = 1 << LGTW, TH = 64 / TW;
  constexpr int HW = 2 * TW + 2;                   // halo pixels per row
  constexpr int HST = HW + 2, HROWS = TH + 1;      // row stride 36 / 20: a multiple of 4 that is not one of 8 (second tap set: XOR 64)
  static_assert(HST % 8 == 4 && (HROWS - 1) * HST + HW <= 192, "halo layout");
  constexpr int S_STAGE = 2 * 64 * 128;            // 16 KiB: two 64-channel segments x 64 pixels
  constexpr int L_STAGE = 192 * 128;               // 24 KiB: 5 x 36 halo rows used
  constexpr int L_BASE = 3 * S_STAGE;
  constexpr int SJ = 2, LJ = 3;
  constexpr unsigned OOB = 0x80000000u;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wa = wave >> 2, kx = wave & 3;
  const int nbt = p.cb >> 6;
  int bx = blockIdx.x;
  const int ky = bx & 1; bx >>= 1;                 // taps ky and ky + 2
  const int b0 = (bx % nbt) * 64, a0 = (bx / nbt) * 128;
  const int ks = blockIdx.y;
  const int t0 = ks * p.tiles_per_split, t1 = min(p.ntile, t0 + p.tiles_per_split);
  const int HL = 2 * p.Hs, WL = 2 * p.Ws;
  const int txs = p.Ws >> LGTW, tys = p.Hs / TH;
  const int lg_txs = p.lgWs - LGTW, lg_tys = p.lgHs - (6 - LGTW);

  const rsrc4_t rsS = make_rsrc4(p.S, (unsigned)((int64_t)p.n * p.Hs * p.Ws * p.ldS * 2));
  const rsrc4_t rsL = make_rsrc4(p.L, (unsigned)((int64_t)p.n * HL * WL * p.ldL * 2));

  // ---- per-lane DMA offsets (bytes, relative to the tile's first S pixel / first halo pixel) -------------------------------
  const int lrow = lane >> 3;
  const int lchunk = (lane & 7) ^ (lrow & 7);
  int s_voff[SJ], l_voff[LJ], l_hr[LJ], l_hx[LJ];
#pragma unroll
  for (int j = 0; j < SJ; ++j) {
    const int ps = wave * SJ + j;
    const int seg = ps >> 3, k = (ps & 7) * 8 + lrow;
    const int ry = k >> LGTW, rx = k & (TW - 1);
    s_voff[j] = ((ry * p.Ws + rx) * p.ldS + p.coffS + a0 + seg * 64) * 2 + lchunk * 16;
  }
#pragma unroll
  for (int j = 0; j < LJ; ++j) {
    const int hrow = (wave * LJ + j) * 8 + lrow;
    const int hr = hrow / HST, hx = hrow - hr * HST;
    l_hr[j] = (hr < HROWS && hx < HW) ? hr : 1 << 20;     // rows of the buffer no tap reads: always zero-filled
    l_hx[j] = hx;
    l_voff[j] = ((2 * hr * WL + hx) * p.ldL + p.coffL + b0) * 2 + lchunk * 16;
  }
  // tiles beyond the split's range are issued all the same, with out-of-range offsets (zeros into a stage nobody reads any
  // more): every step issues the same number of pieces, so the counted waits are immediates and no branch sits between MFMAs
  auto issue_piece = [&](int t, auto STG, auto PIECE) {
    constexpr int stage = decltype(STG)::value, j = decltype(PIECE)::value;
    const bool live = t < t1;
    const int xb = t & (txs - 1), yb = (t >> lg_txs) & (tys - 1), nn = t >> (lg_txs + lg_tys);
    const int y0 = yb * TH, x0 = xb * TW;
    if constexpr (j < SJ) {
      const int sb = live ? ((nn * p.Hs + y0) * p.Ws + x0) * p.ldS * 2 : 0;   // wave-uniform
      blds16w(rsS, live ? (unsigned)s_voff[j] : OOB, sb, smem + stage * S_STAGE + (wave * SJ + j) * 1024);
    } else {
      constexpr int jl = j - SJ;
      const int Y0 = 2 * y0 - 1 + ky, X0 = 2 * x0 - 1;                    // first halo pixel (may lie outside)
      const int Y = Y0 + 2 * l_hr[jl], X = X0 + l_hx[jl];
      const bool ok = live && Y >= 0 && Y < HL && X >= 0 && X < WL;
      const int lb = ((nn * HL + Y0) * WL + X0) * p.ldL * 2;               // may be negative; the sum is not when ok
      blds16w(rsL, ok ? (unsigned)(l_voff[jl] + lb) : OOB, 0, smem + L_BASE + stage * L_STAGE + (wave * LJ + jl) * 1024);
    }
  };

  // ---- fragment read addresses (per lane, loop invariant; stage / tap set = immediates) ---------------------------------------
  const int g = lane >> 4, i16 = lane & 15, q = i16 >> 2, pp = i16 & 3;
  int aoff[2][2][4], boff[2][2][4];
#pragma unroll
  for (int ksx = 0; ksx < 2; ++ksx)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int k = ksx * 32 + 8 * g + 4 * h + q;
      const int srow = wa * 64 + k;
      const int ry = k >> LGTW, rx = k & (TW - 1);
      const int lrw = ry * HST + 2 * rx + kx;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int chunk = t * 2 + (pp >> 1);
        aoff[ksx][h][t] = srow * 128 + ((chunk ^ (srow & 7)) << 4) + (pp & 1) * 8;
        boff[ksx][h][t] = L_BASE + lrw * 128 + ((chunk ^ (lrw & 7)) << 4) + (pp & 1) * 8;
      }
    }
  auto tr = [&](const char* lo_p, const char* hi_p) -> h8_t { return tr16x2(lo_p, hi_p); };
  // A fragment t of k-half KSX, B fragment t of (k-half KSX, tap set S2), from ring stage STG
  auto read_a = [&](auto STG, auto KSX, int t) -> h8_t {
    constexpr int stage = decltype(STG)::value, ksx = decltype(KSX)::value;
    return tr(smem + aoff[ksx][0][t] + stage * S_STAGE, smem + aoff[ksx][1][t] + stage * S_STAGE);
  };
  auto read_b = [&](auto STG, auto KSX, auto S2, int t) -> h8_t {
    constexpr int stage = decltype(STG)::value, ksx = decltype(KSX)::value, s2 = decltype(S2)::value;
    if constexpr (s2 == 0) return tr(smem + boff[ksx][0][t] + stage * L_STAGE, smem + boff[ksx][1][t] + stage * L_STAGE);
    else return tr(smem + (boff[ksx][0][t] ^ 64) + stage * L_STAGE + HST * 128, smem + (boff[ksx][1][t] ^ 64) + stage * L_STAGE + HST * 128);
  };

  f4_t acc[2][4][4];
#pragma unroll
  for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[s2][i][j] = f4_t{0.f, 0.f, 0.f, 0.f};

  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  typedef short s8_t __attribute__((ext_vector_type(8)));
  const short rm = p.relu_S ? (short)0 : (short)-32768;       // ReLU on S as an integer max (no branch): rm = smallest int16 = no change
  const s8_t rmin = {rm, rm, rm, rm, rm, rm, rm, rm};
  auto relu_a = [&](h8_t& v) { v = __builtin_bit_cast(h8_t, __builtin_elementwise_max(__builtin_bit_cast(s8_t, v), rmin)); };

  const int nt_tiles = t1 - t0;
  h8_t af[4], bfA[4], bfB[4];
  if (nt_tiles > 0) {
    static_for<SJ + LJ>([&](auto J) { issue_piece(t0, I0{}, J); });
    static_for<SJ + LJ>([&](auto J) { issue_piece(t0 + 1, I1{}, J); });
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#pragma unroll
    for (int t = 0; t < 4; ++t) { af[t] = read_a(I0{}, I0{}, t); bfA[t] = read_b(I0{}, I0{}, I0{}, t); }
  }
  // one pixel tile from stage STG; its first fragments (af, bfA) are in registers. NXT = stage of the next tile, ISS = stage the
  // tile two ahead is issued into (= the stage of the previous tile: every wave has passed the previous tile's barrier, after
  // which that stage is not read again).
  auto tile = [&](int i, auto STG, auto NXT, auto ISS) {
    const int tnext2 = t0 + i + 2;
    // 64 MFMAs in four blocks (k-half, tap set); idx = running MFMA number; pieces of tile i + 2 at MFMAs 6, 18, 30, 42, 54
    auto slot = [&](auto IDX) {
      constexpr int idx = decltype(IDX)::value;
      static_for<SJ + LJ>([&](auto Q) {
        constexpr int qq = decltype(Q)::value;
        if constexpr (idx == 6 + 12 * qq) issue_piece(tnext2, ISS, Q);
      });
    };
    // block 0: (k0, set 0) on af, bfA; reads bfB <- (k0, set 1)
    static_for<16>([&](auto IDX) {
      constexpr int idx = decltype(IDX)::value, mt = idx >> 2, nt = idx & 3;
      if constexpr (nt == 0) relu_a(af[mt]);
      acc[0][mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[mt], bfA[nt], acc[0][mt][nt], 0, 0, 0);
      if constexpr ((idx & 3) == 1) bfB[idx >> 2] = read_b(STG, I0{}, I1{}, idx >> 2);
      slot(IDX);
      __builtin_amdgcn_sched_barrier(0);
    });
    // block 1: (k0, set 1) on af, bfB; reads bfA <- (k1, set 0) and, as each row of MFMAs has issued, af[mt] <- k1
    static_for<16>([&](auto IDX) {
      constexpr int idx = decltype(IDX)::value, mt = idx >> 2, nt = idx & 3;
      acc[1][mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[mt], bfB[nt], acc[1][mt][nt], 0, 0, 0);
      if constexpr ((idx & 3) == 1) bfA[idx >> 2] = read_b(STG, I1{}, I0{}, idx >> 2);
      if constexpr (nt == 3) af[mt] = read_a(STG, I1{}, mt);
      slot(std::integral_constant<int, idx + 16>{});
      __builtin_amdgcn_sched_barrier(0);
    });
    // block 2: (k1, set 0) on af, bfA; reads bfB <- (k1, set 1)
    static_for<16>([&](auto IDX) {
      constexpr int idx = decltype(IDX)::value, mt = idx >> 2, nt = idx & 3;
      if constexpr (nt == 0) relu_a(af[mt]);
      acc[0][mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[mt], bfA[nt], acc[0][mt][nt], 0, 0, 0);
      if constexpr ((idx & 3) == 1) bfB[idx >> 2] = read_b(STG, I1{}, I1{}, idx >> 2);
      slot(std::integral_constant<int, idx + 32>{});
      __builtin_amdgcn_sched_barrier(0);
    });
    // the next tile (issued during the previous one: a whole step ago) must have landed in every wave's share before block 3
    // reads its first fragments: all but the 4 pieces of tile i + 2 issued so far
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    // block 3: (k1, set 1) on af, bfB; reads the NEXT tile's first fragments
    static_for<16>([&](auto IDX) {
      constexpr int idx = decltype(IDX)::value, mt = idx >> 2, nt = idx & 3;
      acc[1][mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[mt], bfB[nt], acc[1][mt][nt], 0, 0, 0);
      if constexpr ((idx & 3) == 1) bfA[idx >> 2] = read_b(NXT, I0{}, I0{}, idx >> 2);
      if constexpr (nt == 3) af[mt] = read_a(NXT, I0{}, mt);
      slot(std::integral_constant<int, idx + 48>{});
      __builtin_amdgcn_sched_barrier(0);
    });
  };
  for (int i = 0; i < nt_tiles; i += 3) {
    tile(i, I0{}, I1{}, I2{});
    if (i + 1 < nt_tiles) tile(i + 1, I1{}, I2{}, I0{});
    if (i + 2 < nt_tiles) tile(i + 2, I2{}, I0{}, I1{});
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  // ---- epilogue ----------------------------------------------------------------------------------------------------------------
  const int64_t ldw = (int64_t)16 * p.cb;
  // (the store mode is the same for every element: chosen once, not tested per element inside the unrolled loops - it was two scalar
  //  branches per element, 813 branches in this kernel's 4069 instructions. Measured and dropped in round 4: the partial tiles through a
  //  wave-private LDS slab and out as 16-byte stores of whole 256-byte rows - 32 store instructions per lane instead of 128 four-byte ones -
  //  critic conv2 55.5 -> 53.8 us, conv3 57.5 -> 65.0, generator u3 83.9 -> 87.2, u2 83.4 -> 80.5 on one box: the stores were not the limit)
  auto emit = [&](auto MODE) {
    constexpr int mode = decltype(MODE)::value;   // 0 partial tile, 1 direct accumulate, 2 float atomics
    float* base = mode == 0 ? p.part + (int64_t)ks * p.ca * ldw : p.dW;
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      const int colb = ((ky + 2 * s2) * 4 + kx) * p.cb + b0 + (lane & 15);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = a0 + wa * 64 + mt * 16 + (lane >> 4) * 4 + r;
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) {
            const int64_t o = (int64_t)row * ldw + colb + nt * 16;
            const float v = acc[s2][mt][nt][r] * p.scale;
            if constexpr (mode == 0) base[o] = v;
            else if constexpr (mode == 1) base[o] += v;
            else atomicAdd(base + o, v);
          }
        }
    }
  };
  if (p.part) emit(std::integral_constant<int, 0>{});
  else if (p.direct) emit(std::integral_constant<int, 1>{});
  else emit(std::integral_constant<int, 2>{});
}

// dW[i] += sum_k part[k][i] in a fixed order (as in wgrad.hip)
__global__ void __launch_bounds__(256) wgrad2_reduce_kernel(const float* __restrict__ part, float* __restrict__ dW, int64_t count4, int split) {
  __shared__ f4_t red[3][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int64_t i = (int64_t)blockIdx.x * 64 + tx;
  const int per = (split + 3) / 4, k0 = ty * per, k1 = min(split, k0 + per);
  f4_t s = f4_t{0.f, 0.f, 0.f, 0.f};
  if (i < count4)
    s = gi_ordered_sum_f4((const f4_t*)part + i, count4, k0, k1, s);   // eight loads in flight, the same order of additions
  if (ty > 0) red[ty - 1][tx] = s;
  __syncthreads();
  if (ty == 0 && i < count4) {
    s += red[0][tx];
    s += red[1][tx];
    s += red[2][tx];
    f4_t d = *(f4_t*)(dW + i * 4);
    d += s;
    *(f4_t*)(dW + i * 4) = d;
  }
}

constexpr int wgrad2_nky() { return 2; }   // ky tap rows per workgroup (the one-row form of round 1 moved 1.8x the bytes L2 -> LDS)

int wgrad2_split(int n, int Hs, int Ws, int ca, int cb, int* tiles_per_split) {
  const int ntile = n * Hs * Ws / 64;      // 4 x 16 or 8 x 8 pixel tiles
  const int blocks = (ca / 128) * (cb / 64) * (4 / wgrad2_nky());
  int split = (256 + blocks - 1) / blocks;             // one 8-wave workgroup per CU
  if (split > ntile / 8) split = ntile / 8;
  if (split < 1) split = 1;
  const int tps = (ntile + split - 1) / split;
  if (tiles_per_split) *tiles_per_split = tps;
  return (ntile + tps - 1) / tps;
}

// 8 x 8 pixel tiles (small grids 8 wide): served by wgrad3 only
bool wgrad3_square_ok(int n, int Hs, int Ws, int ca, int cb) {
  return ca % 128 == 0 && cb % 64 == 0 && gi_is_pow2(Hs) && Ws == 8 && Hs >= 8 && n >= 1;
}
bool wgrad2_ok(int n, int Hs, int Ws, int ca, int cb) {
  return ca % 128 == 0 && cb % 64 == 0 && gi_is_pow2(Hs) && gi_is_pow2(Ws) && Ws >= 16 && Hs >= 4 && n >= 1;
}

}  // namespace

int64_t op_wgrad2_scratch_bytes(int n, int Hs, int Ws, int ca, int cb) {
  if (!wgrad2_ok(n, Hs, Ws, ca, cb) && !wgrad3_square_ok(n, Hs, Ws, ca, cb)) return 0;
  const int split = wgrad2_split(n, Hs, Ws, ca, cb, nullptr);
  return split > 1 ? (int64_t)split * ca * 16 * cb * 4 : 0;
}

// fp16 only. GI_ERR_UNSUPPORTED: shape not served (the caller uses wgrad.hip).
int op_wgrad2(hipStream_t st, const WgradArgs& a) {
  const bool square = !wgrad2_ok(a.n, a.Hs, a.Ws, a.ca, a.cb) && wgrad3_square_ok(a.n, a.Hs, a.Ws, a.ca, a.cb);
  if (!wgrad2_ok(a.n, a.Hs, a.Ws, a.ca, a.cb) && !square) return GI_ERR_UNSUPPORTED;
  if (a.ldS % 8 != 0 || a.coffS % 8 != 0 || a.ldL % 8 != 0 || a.coffL % 8 != 0) return GI_ERR_UNSUPPORTED;
  if ((int64_t)a.n * a.Hs * a.Ws * a.ldS >= (1ll << 31) || (int64_t)a.n * 4 * a.Hs * a.Ws * a.ldL >= (1ll << 31)) return GI_ERR_UNSUPPORTED;
  int dev = 0;
  GI_HIP(hipGetDevice(&dev));
  const char* zero = gi_igemm3_zero_page(dev);
  if (!zero) return GI_ERR_HIP;
  WP2 p;
  p.S = (const char*)a.S; p.L = (const char*)a.L; p.zero = zero; p.dW = a.dW;
  p.n = a.n; p.Hs = a.Hs; p.Ws = a.Ws; p.lgWs = gi_ilog2(a.Ws); p.lgHs = gi_ilog2(a.Hs);
  p.ca = a.ca; p.ldS = a.ldS; p.coffS = a.coffS; p.cb = a.cb; p.ldL = a.ldL; p.coffL = a.coffL;
  p.relu_S = a.relu_S; p.scale = a.scale;
  p.ntile = a.n * a.Hs * a.Ws / 64;
  const int split = wgrad2_split(a.n, a.Hs, a.Ws, a.ca, a.cb, &p.tiles_per_split);
  const int64_t out_floats = (int64_t)a.ca * 16 * a.cb;
  p.part = nullptr;
  p.direct = split == 1 ? 1 : 0;
  if (split > 1 && a.scratch && a.scratch_bytes >= (int64_t)split * out_floats * 4) p.part = a.scratch;
  constexpr int LDS = 3 * (2 * 64 * 128 + 192 * 128);
  static GiDevOnce attr;
  if (attr.first()) {
    GI_HIP(hipFuncSetAttribute((const void*)wgrad2_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    GI_HIP(hipFuncSetAttribute((const void*)wgrad3_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    GI_HIP(hipFuncSetAttribute((const void*)wgrad3_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
  }
  const dim3 grid((a.ca / 128) * (a.cb / 64) * (4 / wgrad2_nky()), split);
  const int pipe = gi_opt(GI_OPT_WGRAD3);   // GI_WGRAD3=0: the unpipelined two-ky kernel (also the fallback beyond 2^31-byte tensors)
  // wgrad3 addresses both tensors with 32-bit byte offsets (buffer descriptors; 2^31 marks out-of-range)
  const bool small32 = (int64_t)a.n * a.Hs * a.Ws * a.ldS * 2 < (1ll << 31) && (int64_t)a.n * 4 * a.Hs * a.Ws * a.ldL * 2 < (1ll << 31);
  if (square) {   // 8 x 8 maps: wgrad3 only, and only where the partial-tile pass stays small (u5: 42.7 + 23.2 us against 86.7 us
    // for wgrad.hip; with four splits - d5 - the 67 MB of partial tiles cost more than the faster K loop gains: 48.7 against 45.2)
    if (!(pipe && small32) || split > 2) return GI_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(wgrad3_kernel<3>, grid, dim3(512), LDS, st, p);
    gi_note_kernel("wgrad3<3>");
  } else if (pipe && small32) { hipLaunchKernelGGL(wgrad3_kernel<4>, grid, dim3(512), LDS, st, p); gi_note_kernel("wgrad3<4>"); }
  else { hipLaunchKernelGGL(wgrad2_kernel<2>, grid, dim3(512), LDS, st, p); gi_note_kernel("wgrad2<2>"); }
  GI_LAUNCH_CHECK();
  if (p.part) {
    const int64_t c4 = out_floats / 4;
    hipLaunchKernelGGL(wgrad2_reduce_kernel, dim3((unsigned)((c4 + 63) / 64)), dim3(256), 0, st, (const float*)p.part, a.dW, c4, split);
    GI_LAUNCH_CHECK();
  }
  return GI_OK;
}
