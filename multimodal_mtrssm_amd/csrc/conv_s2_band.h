// Conv2d(k = 3, s = 2, p = 1) forward with few input channels (<= 8 per position: the encoders' first two layers, frame +
// 2 coordinate channels -> 8 and 8 -> 16, default.yaml:31-60), included by conv.hip after conv_split.h.
//
// Both layers are HBM-bound by two orders of magnitude (1.4 / 3.8 GFLOP for 157 MB each); the VALU kernel
// (conv_gather_thin_kernel: one thread per output pixel, every tap a stride-2 4-byte gather) and the patch-staged split
// kernel (two barriers per 16-channel step) ran them at 1.6 TB/s.  Here the staged recipe of the weight-gradient kernels:
//   * a tile = 256 output pixels of one frame (a band of 256 / Wq output rows) = 2 * 256 / Wq + 1 source rows, loaded once
//     with coalesced 16-byte loads, four positions (all <= 8 channels) per thread, a whole tile ahead in registers;
//   * activated and split into two bf16 pieces once, written as ONE 16-byte LDS row per position and piece, the columns
//     de-interleaved by parity (even image columns | odd image columns) so that the 32 pixels of an MFMA column tile read
//     consecutive rows for every tap of the stride-2 window;
//   * a tap IS a k-half: k-block kb = taps 2 kb and 2 kb + 1 (lane half kl), 8 channels each; 5 k-blocks (tap 9 has zero
//     weights) x 3 products per 32-pixel unit, the weights (32 output-channel rows, of which 8 or 16 are real) in registers
//     for the whole launch;
//   * two workgroups per CU, each with two register sets of requests (this tile's and the next one's); a third workgroup
//     per CU (168 registers: spills in the loop) doubled the 3-channel layer's time.
// Same arithmetic as the other bf16x2 kernels (fp32 accumulation of the three products).
#pragma once
#include <type_traits>

namespace mtrssm {

using band_f4 = __attribute__((ext_vector_type(4))) float;
constexpr int kBandPx = 256;  // output pixels per tile

__host__ __device__ constexpr int band_row_bytes(int Wq) { return (2 * Wq + 2) * 16; }  // Wq + 1 even slots, Wq odd slots, 1 pad
__host__ __device__ constexpr int band_lds_bytes(int Wq) { return 2 * (2 * (kBandPx / Wq) + 1) * band_row_bytes(Wq); }

template <int C, int C2>  // input channels per frame / frame-independent coordinate channels (1 + 2 or 8 + 0)
__global__ __launch_bounds__(256, 2) void conv3x3s2_band_kernel(const GatherProblem pa, const GatherProblem pb) {
  const bool second = blockIdx.x >= (unsigned)pa.nx;  // workgroup-uniform
  const GatherProblem& P = second ? pb : pa;
  const MtrssmConvGeom g = P.g;
  const float* __restrict__ src = P.src;
  const float* __restrict__ src2 = P.src2;
  const float* __restrict__ bias = P.bias;
  float* __restrict__ out = P.out;
  const int wg = second ? (int)blockIdx.x - pa.nx : (int)blockIdx.x, nwg = P.nx;
  const int Ws = g.Ws, Wq = g.Wq;                 // host: Ws == 2 Wq, Hs == 2 Hq, Wq in {8, 16, 32}
  const int wsh = 31 - __builtin_clz(Ws), qsh = wsh - 1;
  const int BR = kBandPx >> qsh;                  // output rows per band (host: Hq % BR == 0)
  const int bands = g.Hq / BR, ntiles = g.N * bands;
  const int SR = 2 * BR + 1;                      // source rows per band (the first one is the row above the band)
  const int RP = band_row_bytes(Wq), IMG = SR * RP;
  constexpr int CT = C + C2;                      // host: g.C == C, g.C2 == C2
  static_assert(CT <= 8, "one 16-byte row per position");
  const int plane_s = g.Hs * Ws, plane_o = g.Hq * Wq;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, il = lane & 31, kl = lane >> 5;
  if (wg >= ntiles) return;
  extern __shared__ __attribute__((aligned(16))) unsigned char band_lds[];  // [2 pieces][SR rows][RP]

  // weights: A operands, row = output channel il, k-block kb = (tap 2 kb + kl) x channels 0..7
  bf16x8 a[5][2];
  {
    const size_t piece = (size_t)g.CoutPad * 9 * g.Cpad;  // host: CoutPad == 32, Cpad == 16
#pragma unroll
    for (int kb = 0; kb < 5; ++kb) {
      const int tap = 2 * kb + kl;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        u32x4 v = u32x4{0u, 0u, 0u, 0u};
        if (tap < 9) v = *reinterpret_cast<const u32x4*>(P.wq + s * piece + ((size_t)il * 9 + tap) * g.Cpad);
        a[kb][s] = __builtin_bit_cast(bf16x8, v);
      }
    }
  }
  for (int o = tid * 16; o < 2 * IMG; o += 256 * 16) *reinterpret_cast<u32x4*>(band_lds + o) = u32x4{0u, 0u, 0u, 0u};

  // a thread owns 4 consecutive positions of one row (one 16-byte request per channel): the band's 1024 positions are one such
  // group per thread; the row above the band (Ws / 4 groups x CT channels <= 128 requests) is ONE more request per thread,
  // thread -> (channel hc, group hg), unconditional like the others (idle threads repeat a valid address)
  const int hgroups = Ws >> 2;
  const int hc = tid / hgroups, hg = tid - hc * hgroups;
  const bool hreal = hc < CT;
  const int hcc = hreal ? hc : 0;
  band_f4 pv[2][CT], ph[2];  // [register set]
  // requests as buffer loads: one 32-bit offset per group, the frame / channel part of the address in scalar registers
  const __amdgpu_buffer_rsrc_t rs2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src2 ? src2 : src), 0, 0x7fffffff, 0x00020000);
  auto request = [&](int tile, auto set_tag) {
    constexpr int SET = decltype(set_tag)::value;
    const int n = tile / bands, band = tile - n * bands;
    const int sy0 = 2 * band * BR;  // the band's first own source row (the row above it is the halo request)
    const __amdgpu_buffer_rsrc_t rsf =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src + (size_t)n * C * plane_s), 0, 0x7fffffff, 0x00020000);
    const int off = (sy0 * Ws + tid * 4) * 4;
#pragma unroll
    for (int c = 0; c < CT; ++c)
      pv[SET][c] = __builtin_bit_cast(band_f4, __builtin_amdgcn_raw_buffer_load_b128(c < C ? rsf : rs2, off, (c < C ? c : c - C) * plane_s * 4, 0));
    const int hoff = (((sy0 > 0 ? sy0 - 1 : 0) * Ws + hg * 4) + (hcc < C ? hcc : hcc - C) * plane_s) * 4;  // (frame row -1: stored as zeros)
    if (C2 > 0) {
      const band_f4 hf = __builtin_bit_cast(band_f4, __builtin_amdgcn_raw_buffer_load_b128(rsf, hcc < C ? hoff : 0, 0, 0));
      const band_f4 h2 = __builtin_bit_cast(band_f4, __builtin_amdgcn_raw_buffer_load_b128(rs2, hcc < C ? 0 : hoff, 0, 0));
      ph[SET] = hcc < C ? hf : h2;
    } else {
      ph[SET] = __builtin_bit_cast(band_f4, __builtin_amdgcn_raw_buffer_load_b128(rsf, hoff, 0, 0));
    }
  };
  const int act = g.act;
  const int mode = g.pre_act == 0 ? 0 : (act == MTRSSM_ACT_ELU ? 1 : 2);  // (one uniform branch per tile, not per element)
  auto stage_as = [&](int tile, auto mode_tag, auto set_tag) {
    constexpr int MODE = decltype(mode_tag)::value, SET = decltype(set_tag)::value;
    const int band = tile % bands;
    auto activate = [&](float v) {
      if (MODE == 1) v = elu_fast(v);
      if (MODE == 2) v = act_fwd(v, act);
      return v;
    };
    {
      const int p = tid * 4, r = 1 + (p >> wsh), x = p & (Ws - 1);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        u16x8 q0, q1;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          float v = 0.f;
          if (c < CT) v = activate(pv[SET][c < CT ? c : 0][e]);
          unsigned short h[2];
          split_bf16<2>(v, h);
          q0[c] = h[0];
          q1[c] = h[1];
        }
        const int ic = x + e + 1;  // image column (0 = the zero column left of the frame)
        const int slot = (ic & 1) ? Wq + 1 + (ic >> 1) : (ic >> 1);
        unsigned char* d = band_lds + r * RP + slot * 16;
        *reinterpret_cast<u16x8*>(d) = q0;
        *reinterpret_cast<u16x8*>(d + IMG) = q1;
      }
    }
    if (hreal) {  // image row 0: the source row above the band, one channel of 4 positions per thread
      const bool real = band > 0;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        unsigned short h[2];
        split_bf16<2>(real ? activate(ph[SET][e]) : 0.f, h);
        const int ic = hg * 4 + e + 1;
        const int slot = (ic & 1) ? Wq + 1 + (ic >> 1) : (ic >> 1);
        unsigned short* d = reinterpret_cast<unsigned short*>(band_lds + slot * 16) + hc;
        d[0] = h[0];
        d[IMG / 2] = h[1];
      }
    }
  };
  auto stage = [&](int tile, auto set_tag) {
    if (mode == 0) stage_as(tile, std::integral_constant<int, 0>{}, set_tag);
    else if (mode == 1) stage_as(tile, std::integral_constant<int, 1>{}, set_tag);
    else stage_as(tile, std::integral_constant<int, 2>{}, set_tag);
  };

  // this wave's two units: pixel j = 32 u + il of the band -> (oy, ox); tap (ty, tx) reads image row 2 oy + ty, column 2 ox + tx
  unsigned baddr[2][5];
  int opix[2];
#pragma unroll
  for (int uu = 0; uu < 2; ++uu) {
    const int j = (2 * wave + uu) * 32 + il, oy = j >> qsh, ox = j & (Wq - 1);
    opix[uu] = oy * Wq + ox;
#pragma unroll
    for (int kb = 0; kb < 5; ++kb) {
      const int tap = 2 * kb + kl < 9 ? 2 * kb + kl : 8, ty = tap / 3, tx = tap - 3 * ty;
      const int slot = (tx & 1) ? Wq + 1 + ox : ox + (tx >> 1);
      baddr[uu][kb] = (unsigned)((2 * oy + ty) * RP + slot * 16);
    }
  }
  float bv[16];
  {
    const float* bsafe = bias ? bias : src;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = 4 * kl + (r & 3) + 8 * (r >> 2);
      const float b = bsafe[co < g.Cout ? co : 0];
      bv[r] = (bias && co < g.Cout) ? b : 0.f;
    }
  }

  // Two register sets of requests: the tile after this one is requested BEFORE this one is converted, so a workgroup has
  // requests in flight all the time (one set, requested after the conversion: 2.5 TB/s -- nothing was in flight while a
  // workgroup converted).  The requests stay unconditional (past the last tile they repeat it) so that the wait before a
  // conversion counts only the older set.
  auto compute = [&](int tile) {
    const int n = tile / bands, band = tile - n * bands;
    float* obase = out + (size_t)n * g.Cout * plane_o + (size_t)band * kBandPx;
    f32x16 acc[2];  // the two units' chains interleaved: a dependent MFMA waits out its predecessor's passes
#pragma unroll
    for (int uu = 0; uu < 2; ++uu)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[uu][r] = bv[r];
#pragma unroll
    for (int kb = 0; kb < 5; ++kb) {
      bf16x8 b0[2], b1[2];
#pragma unroll
      for (int uu = 0; uu < 2; ++uu) {
        b0[uu] = *reinterpret_cast<const bf16x8*>(band_lds + baddr[uu][kb]);
        b1[uu] = *reinterpret_cast<const bf16x8*>(band_lds + baddr[uu][kb] + IMG);
      }
#pragma unroll
      for (int uu = 0; uu < 2; ++uu) acc[uu] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[kb][0], b1[uu], acc[uu], 0, 0, 0);
#pragma unroll
      for (int uu = 0; uu < 2; ++uu) acc[uu] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[kb][1], b0[uu], acc[uu], 0, 0, 0);
#pragma unroll
      for (int uu = 0; uu < 2; ++uu) acc[uu] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[kb][0], b0[uu], acc[uu], 0, 0, 0);
    }
#pragma unroll
    for (int uu = 0; uu < 2; ++uu)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = 4 * kl + (r & 3) + 8 * (r >> 2);
        if (co < g.Cout) obase[(size_t)co * plane_o + opix[uu]] = acc[uu][r];
      }
  };
  const int last = wg + ((ntiles - 1 - wg) / nwg) * nwg;  // this workgroup's last tile
  const std::integral_constant<int, 0> s0{};
  const std::integral_constant<int, 1> s1{};
  request(wg, s0);
  __syncthreads();  // the zero fill
  for (int tile = wg; tile < ntiles; tile += 2 * nwg) {
    const int t1 = tile + nwg, t2 = tile + 2 * nwg;
    request(t1 < last ? t1 : last, s1);
    stage(tile, s0);
    lds_barrier();
    compute(tile);
    lds_barrier();  // every wave is done with the image before the next tile overwrites it
    if (t1 >= ntiles) break;  // workgroup-uniform
    request(t2 < last ? t2 : last, s0);
    stage(t1, s1);
    lds_barrier();
    compute(t1);
    lds_barrier();
  }
}


// ------------------------------------------------------------------------------------------------
// ConvTranspose2d(k = 4, s = 2, p = 1) forward with 16 input channels and ONE or two output channels (the decoders' last layer,
// default.yaml:70-74) on the staged recipe.  convt_k4s2_thin_kernel (VALU: 256 fmas + 144 LDS reads per thread for a 2 x 2
// output block) is issue-bound at 2.4 TB/s.  Here the four output parity classes of an input position are the rows of an MFMA
// tile: out[2 iy + qy][2 ix + qx] = sum over the 3 x 3 neighbourhood (dy, dx) and ci of A[q][(dy, dx)][ci] * act(x)[ci][iy + dy][ix + dx]
// with A[q][(dy, dx)][ci] = w[ci][co][qy + 1 - 2 dy][qx + 1 - 2 dx] where that tap exists (four of the nine per class), else 0:
// 5 k-steps (two neighbours of 16 channels each) x 3 products of v_mfma_f32_16x16x32_bf16 per 16 positions.  A tile = one frame (1024
// input positions: 64 x 16 or 32 x 32), staged once: 16-byte requests (wave w = channels 4 w .. 4 w + 3, a lane = four groups of
// four positions; two register sets: the next frame is requested before this one is converted), converted once into
// [piece][channel half][haloed position] 16-byte rows -- the 32 positions of a unit read consecutive rows for every neighbour,
// the zero border is written once per launch.  Rows 0 .. 4 Cout - 1 of the 16 x 16 accumulator are the 2 x 2 output blocks.
// ------------------------------------------------------------------------------------------------
struct ConvtBandProblem {
  const float* src;   // [N][16][Hs][Ws]
  const float* w;     // [16][Cout][4][4]
  const float* bias;  // [Cout] or null
  float* out;         // [N][Cout][2 Hs][2 Ws]
  int N, Hs, Ws, Cout, pre_act, act, nx;
};

__host__ __device__ constexpr int convt_band_lds_bytes(int Hs, int Ws) { return 2 * 2 * ((Hs + 2) * (Ws + 2) + 1) * 16; }

__global__ __launch_bounds__(256, 2) void convt4s2_band_kernel(const ConvtBandProblem pa, const ConvtBandProblem pb) {
  const bool second = blockIdx.x >= (unsigned)pa.nx;  // workgroup-uniform
  const ConvtBandProblem& P = second ? pb : pa;
  const int wg = second ? (int)blockIdx.x - pa.nx : (int)blockIdx.x, nwg = P.nx;
  constexpr int C = 16, PLANE = 1024;
  const int Ws = P.Ws, Hs = P.Hs, Cout = P.Cout;            // host: Hs * Ws == 1024, Ws in {16, 32}, Cout <= 2
  const int wsh = 31 - __builtin_clz(Ws);
  const int ntiles = P.N;
  const int PW = Ws + 2;                                    // haloed row: positions -1 .. Ws
  const int NPOS = (Hs + 2) * PW + 1;
  const int HALF = NPOS * 16, IMG = 2 * HALF;               // one channel half / one piece
  const int Wo = 2 * Ws;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (wg >= ntiles) return;
  extern __shared__ __attribute__((aligned(16))) unsigned char band_lds[];  // [2 pieces][2 channel halves][NPOS] x 16 B

  // A operands of v_mfma_f32_16x16x32_bf16: row = lane & 15 = class q = row & 3 of output channel row >> 2; a k-step s holds two
  // neighbours (dy, dx) = (t / 3 - 1, t % 3 - 1), t = 2 s and 2 s + 1, of 16 channels each: lane group g = lane >> 4 has
  // neighbour 2 s + (g >> 1), channels 8 (g & 1) .. + 7.  (The 32 x 32 x 16 shape spent 27 MFMAs of 32 cycles per 32 positions
  // on 4 live rows of 32; this one 15 of 16 cycles per 16 positions.)
  const int lg = lane >> 4, lrow = lane & 15;
  bf16x8 a[5][2];
  {
    const int q = lrow & 3, co = lrow >> 2, qy = q >> 1, qx = q & 1;
#pragma unroll
    for (int ss = 0; ss < 5; ++ss) {
      const int t = 2 * ss + (lg >> 1);
      const int dy = t / 3 - 1, dx = t % 3 - 1;
      const int ky = qy + 1 - 2 * dy, kx = qx + 1 - 2 * dx;
      const bool live = t < 9 && co < Cout && ky >= 0 && ky < 4 && kx >= 0 && kx < 4;
      const int wofs = live ? co * 16 + ky * 4 + kx : 0;   // (unconditional requests from a valid address, selected afterwards)
      u16x8 h0, h1;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float wv = P.w[(size_t)(8 * (lg & 1) + j) * Cout * 16 + wofs];
        const float v = live ? wv : 0.f;
        unsigned short hh[2];
        split_bf16<2>(v, hh);
        h0[j] = hh[0];
        h1[j] = hh[1];
      }
      a[ss][0] = __builtin_bit_cast(bf16x8, h0);
      a[ss][1] = __builtin_bit_cast(bf16x8, h1);
    }
  }
  for (int o = tid * 16; o < 2 * IMG; o += 256 * 16) *reinterpret_cast<u32x4*>(band_lds + o) = u32x4{0u, 0u, 0u, 0u};

  band_f4 pv[1][4][4];  // [position group l + 64 k][channel 4 wave + c] (ONE set: two would not leave two workgroups per CU their registers)
  auto request = [&](int tile, auto set_tag) {
    constexpr int SET = decltype(set_tag)::value;
    const __amdgpu_buffer_rsrc_t rsf =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(P.src + ((size_t)tile * C + 4 * wave) * PLANE), 0, 0x7fffffff, 0x00020000);
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int c = 0; c < 4; ++c)
        pv[SET][k][c] = __builtin_bit_cast(band_f4, __builtin_amdgcn_raw_buffer_load_b128(rsf, (lane + 64 * k) * 16, c * PLANE * 4, 0));
  };
  const int act = P.act;
  const int mode = P.pre_act == 0 ? 0 : (act == MTRSSM_ACT_ELU ? 1 : 2);
  const int hf = wave >> 1, cofs = (wave & 1) * 8;   // channels 4 wave .. + 3: half hf, byte offset cofs inside the 16-byte row
  auto stage_as = [&](auto mode_tag, auto set_tag) {
    constexpr int MODE = decltype(mode_tag)::value, SET = decltype(set_tag)::value;
    auto activate = [&](float v) {
      if (MODE == 1) v = elu_fast(v);
      if (MODE == 2) v = act_fwd(v, act);
      return v;
    };
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int p = (lane + 64 * k) * 4, r = p >> wsh, x = p & (Ws - 1);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        unsigned short h[4][2];
#pragma unroll
        for (int c = 0; c < 4; ++c) split_bf16<2>(activate(pv[SET][k][c][e]), h[c]);
        unsigned char* d = band_lds + hf * HALF + ((r + 1) * PW + x + e + 1) * 16 + cofs;
        *reinterpret_cast<uint2*>(d) = make_uint2((unsigned)h[0][0] | ((unsigned)h[1][0] << 16), (unsigned)h[2][0] | ((unsigned)h[3][0] << 16));
        *reinterpret_cast<uint2*>(d + IMG) = make_uint2((unsigned)h[0][1] | ((unsigned)h[1][1] << 16), (unsigned)h[2][1] | ((unsigned)h[3][1] << 16));
      }
    }
  };
  auto stage = [&](auto set_tag) {
    if (mode == 0) stage_as(std::integral_constant<int, 0>{}, set_tag);
    else if (mode == 1) stage_as(std::integral_constant<int, 1>{}, set_tag);
    else stage_as(std::integral_constant<int, 2>{}, set_tag);
  };

  float b0 = 0.f, b1 = 0.f;
  if (P.bias) {
    b0 = P.bias[0];
    b1 = Cout > 1 ? P.bias[1] : 0.f;
  }
  const float bme = lg == 0 ? b0 : (lg == 1 ? b1 : 0.f);   // accumulator registers 0..3 of lane group g = rows 4 g + r = the classes of output channel g
  // per-lane neighbour offsets of the five k-steps (neighbour 9 does not exist: its weights are zero, any valid row will do)
  int toff[5];
#pragma unroll
  for (int ss = 0; ss < 5; ++ss) {
    const int t = 2 * ss + (lg >> 1) < 9 ? 2 * ss + (lg >> 1) : 8;
    toff[ss] = ((t / 3 - 1) * PW + (t % 3 - 1)) * 16;
  }
  using f32x4v = __attribute__((ext_vector_type(4))) float;
  auto compute = [&](int tile) {
    float* const o = P.out + ((size_t)tile * Cout + (lg < Cout ? lg : 0)) * (4 * PLANE);
    // this wave's sixteen 16-position units, four at a time (four independent MFMA chains): position j = 16 u + (lane & 15)
#pragma unroll 1
    for (int up = 0; up < 4; ++up) {
      unsigned bb[4];
      int opx[4];
#pragma unroll
      for (int uu = 0; uu < 4; ++uu) {
        const int j = (16 * wave + 4 * up + uu) * 16 + lrow, iy = j >> wsh, ix = j & (Ws - 1);
        bb[uu] = (unsigned)((lg & 1) * HALF + ((iy + 1) * PW + ix + 1) * 16);
        opx[uu] = (2 * iy) * Wo + 2 * ix;
      }
      f32x4v acc[4];
#pragma unroll
      for (int uu = 0; uu < 4; ++uu) acc[uu] = f32x4v{bme, bme, bme, bme};
#pragma unroll
      for (int ss = 0; ss < 5; ++ss) {
        bf16x8 q0[4], q1[4];
#pragma unroll
        for (int uu = 0; uu < 4; ++uu) {
          q0[uu] = *reinterpret_cast<const bf16x8*>(band_lds + bb[uu] + toff[ss]);
          q1[uu] = *reinterpret_cast<const bf16x8*>(band_lds + bb[uu] + toff[ss] + IMG);
        }
#pragma unroll
        for (int uu = 0; uu < 4; ++uu) acc[uu] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[ss][0], q1[uu], acc[uu], 0, 0, 0);
#pragma unroll
        for (int uu = 0; uu < 4; ++uu) acc[uu] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[ss][1], q0[uu], acc[uu], 0, 0, 0);
#pragma unroll
        for (int uu = 0; uu < 4; ++uu) acc[uu] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[ss][0], q0[uu], acc[uu], 0, 0, 0);
      }
      if (lg < Cout) {
#pragma unroll
        for (int uu = 0; uu < 4; ++uu) {
          *reinterpret_cast<float2*>(o + opx[uu]) = make_float2(acc[uu][0], acc[uu][1]);
          *reinterpret_cast<float2*>(o + opx[uu] + Wo) = make_float2(acc[uu][2], acc[uu][3]);
        }
      }
    }
  };
  // Two workgroups per CU: one converts (VALU) while the other multiplies -- a single wave per SIMD issues both in turn (counters
  // of the one-workgroup form: 41 us of MFMA + 29 us of VALU + 30 us of waits per wave in a 102 us launch).  The next frame is
  // requested right after this one's conversion and flies under its products.
  const int last = wg + ((ntiles - 1 - wg) / nwg) * nwg;
  const std::integral_constant<int, 0> s0{};
  request(wg, s0);
  __syncthreads();  // the zero fill
  for (int tile = wg; tile < ntiles; tile += nwg) {
    stage(s0);
    const int nx = tile + nwg;
    request(nx < last ? nx : last, s0);
    lds_barrier();
    compute(tile);
    lds_barrier();
  }
}

// ------------------------------------------------------------------------------------------------
// Conv2d(k = 4, s = 2, p = 1) gather with 16 input and 32 output channels on frames of 1024 positions (64 x 16, 32 x 32): the
// backward-data of the decoders' second ConvTranspose layer (default.yaml:66-69; out *= act'(layer input)).  The patch-staged
// split kernel re-stages weights and patch per 16-channel step (two barriers a step) and ran it at 3.4 TB/s; here the frame is
// staged once exactly as in convt4s2_band_kernel ([piece][channel half][haloed position] 16-byte rows, zero border written once),
// a tap is a k-block (16 taps x 3 products per 32 output pixels, all 32 accumulator rows live), the weights are A operands in
// registers for the whole launch.  One workgroup per CU: every thread stages two (4 positions x 4 channels) items, a wave
// multiplies ONE (32-pixel unit, 32-channel tile) job of the frame; the next frame's requests fly under the products and the
// epilogue (act' operand + stores).  Second instance: 32 -> 64 channels on 256-position frames (the FIRST ConvTranspose layer's
// backward-data): four waves = 2 units x 2 channel tiles, 32 k-blocks, 256 weight registers per lane.
// ------------------------------------------------------------------------------------------------
template <int CIN, int COUT, int PLANE>   // <16, 32, 1024>: eight waves, one 32-pixel unit each; <32, 64, 256>: four waves = 2 units x 2 channel tiles
__global__ __launch_bounds__(64 * (PLANE / 128) * (COUT / 32), 1) void conv4s2_band_kernel(const GatherProblem pa, const GatherProblem pb) {
  constexpr int OPX = PLANE / 4, UNITS = OPX / 32, CTS = COUT / 32, NW = UNITS * CTS, NT = 64 * NW;
  constexpr int OCT = CIN / 8, GPT = CIN / 16, KB = 16 * GPT;     // 8-channel octets per position; 16-channel groups per tap; k-blocks
  constexpr int NITEM = (PLANE / 4) * (CIN / 4) / NT;             // (4 positions x 4 channels) items per thread
  static_assert(NITEM * NT == (PLANE / 4) * (CIN / 4) && NITEM >= 1, "whole items");
  const bool second = blockIdx.x >= (unsigned)pa.nx;  // workgroup-uniform
  const GatherProblem& P = second ? pb : pa;
  const MtrssmConvGeom g = P.g;
  const int wg = second ? (int)blockIdx.x - pa.nx : (int)blockIdx.x, nwg = P.nx;
  const int Ws = g.Ws, Hs = g.Hs, Wq = g.Wq;                // host: Hs * Ws == PLANE, Ws a power of two, Wq = Ws / 2, Hq = Hs / 2
  const int wsh = 31 - __builtin_clz(Ws), qsh = wsh - 1;
  const int ntiles = g.N;
  const int PW = Ws + 2;
  const int NPOS = (Hs + 2) * PW + 1;
  const int OSZ = NPOS * 16, IMG = OCT * OSZ;               // one octet plane / one piece
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, il = lane & 31, kl = lane >> 5;
  if (wg >= ntiles) return;
  extern __shared__ __attribute__((aligned(16))) unsigned char band_lds[];  // [2 pieces][OCT][NPOS] x 16 B
  const int unit = wave % UNITS, ct = wave / UNITS;

  // A operands: row il = output channel 32 ct + il, k-block kb = (tap kb / GPT, channel group kb % GPT), this lane's 8 k-values =
  // input channels 16 (kb % GPT) + 8 kl .. + 7
  bf16x8 a[KB][2];
  {
    const size_t piece = (size_t)g.CoutPad * 16 * g.Cpad;  // host: CoutPad == COUT, Cpad == CIN
#pragma unroll
    for (int kb = 0; kb < KB; ++kb)
#pragma unroll
      for (int s = 0; s < 2; ++s)
        a[kb][s] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(P.wq + s * piece + ((size_t)(32 * ct + il) * 16 + kb / GPT) * CIN +
                                                                                    16 * (kb % GPT) + 8 * kl));
  }
  for (int o = tid * 16; o < 2 * IMG; o += NT * 16) *reinterpret_cast<u32x4*>(band_lds + o) = u32x4{0u, 0u, 0u, 0u};

  band_f4 pv[NITEM][4];
  auto request = [&](int tile) {
    const __amdgpu_buffer_rsrc_t rsf =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(P.src + (size_t)tile * CIN * PLANE), 0, 0x7fffffff, 0x00020000);
#pragma unroll
    for (int k = 0; k < NITEM; ++k) {
      const int it = tid + NT * k, pg = it % (PLANE / 4), cq = it / (PLANE / 4);
#pragma unroll
      for (int c = 0; c < 4; ++c)
        pv[k][c] = __builtin_bit_cast(band_f4, __builtin_amdgcn_raw_buffer_load_b128(rsf, ((4 * cq + c) * PLANE + 4 * pg) * 4, 0, 0));
    }
  };
  const int act = g.act;
  const int mode = g.pre_act == 0 ? 0 : (act == MTRSSM_ACT_ELU ? 1 : 2);
  auto stage_as = [&](auto mode_tag) {
    constexpr int MODE = decltype(mode_tag)::value;
    auto activate = [&](float v) {
      if (MODE == 1) v = elu_fast(v);
      if (MODE == 2) v = act_fwd(v, act);
      return v;
    };
#pragma unroll
    for (int k = 0; k < NITEM; ++k) {
      const int it = tid + NT * k, pg = it % (PLANE / 4), cq = it / (PLANE / 4);
      const int p = pg * 4, r = p >> wsh, x = p & (Ws - 1);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        unsigned short h[4][2];
#pragma unroll
        for (int c = 0; c < 4; ++c) split_bf16<2>(activate(pv[k][c][e]), h[c]);
        unsigned char* d = band_lds + (cq >> 1) * OSZ + ((r + 1) * PW + x + e + 1) * 16 + (cq & 1) * 8;
        *reinterpret_cast<uint2*>(d) = make_uint2((unsigned)h[0][0] | ((unsigned)h[1][0] << 16), (unsigned)h[2][0] | ((unsigned)h[3][0] << 16));
        *reinterpret_cast<uint2*>(d + IMG) = make_uint2((unsigned)h[0][1] | ((unsigned)h[1][1] << 16), (unsigned)h[2][1] | ((unsigned)h[3][1] << 16));
      }
    }
  };
  auto stage = [&]() {
    if (mode == 0) stage_as(std::integral_constant<int, 0>{});
    else if (mode == 1) stage_as(std::integral_constant<int, 1>{});
    else stage_as(std::integral_constant<int, 2>{});
  };

  // this wave's unit: output pixel j = 32 unit + il -> (oy, ox); tap (ky, kx) reads source (2 oy - 1 + ky, 2 ox - 1 + kx) = image (2 oy + ky, 2 ox + kx)
  const int j = 32 * unit + il, oy = j >> qsh, ox = j & (Wq - 1);
  const unsigned bb = (unsigned)(kl * OSZ + ((2 * oy) * PW + 2 * ox) * 16);
  const float* __restrict__ actgrad = P.actgrad_in;
  const float* __restrict__ bias = P.bias;
  const bool act_elu = act == MTRSSM_ACT_ELU, act_relu = act == MTRSSM_ACT_RELU;
  float bv[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) bv[r] = bias ? bias[32 * ct + 4 * kl + (r & 3) + 8 * (r >> 2)] : 0.f;
  auto frag_off = [&](int kb) { return (unsigned)(((kb / GPT) >> 2) * PW + ((kb / GPT) & 3)) * 16u + (unsigned)(2 * (kb % GPT)) * (unsigned)OSZ; };
  auto compute = [&](int tile) {
    const size_t obase = (size_t)tile * COUT * OPX + (size_t)(32 * ct + 4 * kl) * OPX + j;
    float gv[16];
    if (actgrad) {
#pragma unroll
      for (int r = 0; r < 16; ++r) gv[r] = actgrad[obase + (size_t)((r & 3) + 8 * (r >> 2)) * OPX];
    }
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = bv[r];
    bf16x8 q0[2], q1[2];
    q0[0] = *reinterpret_cast<const bf16x8*>(band_lds + bb + frag_off(0));
    q1[0] = *reinterpret_cast<const bf16x8*>(band_lds + bb + frag_off(0) + IMG);
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      const int st = kb & 1;
      if (kb + 1 < KB) {
        q0[st ^ 1] = *reinterpret_cast<const bf16x8*>(band_lds + bb + frag_off(kb + 1));
        q1[st ^ 1] = *reinterpret_cast<const bf16x8*>(band_lds + bb + frag_off(kb + 1) + IMG);
      }
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[kb][0], q1[st], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[kb][1], q0[st], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[kb][0], q0[st], acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float v = acc[r];
      if (actgrad) {
        const float x = gv[r];
        float e = __expf(x);
        const float neg = act_elu ? e : (act_relu ? 0.f : 1.f);
        v *= x > 0.f ? 1.f : neg;
      }
      P.out[obase + (size_t)((r & 3) + 8 * (r >> 2)) * OPX] = v;
    }
  };
  const int last = wg + ((ntiles - 1 - wg) / nwg) * nwg;
  request(wg);
  __syncthreads();  // the zero fill
  for (int tile = wg; tile < ntiles; tile += nwg) {
    stage();
    const int nx = tile + nwg;
    request(nx < last ? nx : last);
    lds_barrier();
    compute(tile);
    lds_barrier();
  }
}

// ------------------------------------------------------------------------------------------------
// ConvTranspose2d(k = 4, s = 2, p = 1) with the four output parity classes as ROWS of the MFMA tile (the recipe of
// convt4s2_band_kernel, for the decoders' first two layers and the backward-data of the encoders' stride-2 convs): a tile row is
// (output channel, class) = 4 co + q, a column an INPUT position, K runs over the position's 3 x 3 neighbourhood x input
// channels (a class uses 4 of the 9 neighbours, the other weights are zero: 2.25 x the products, all 32 rows live).  A lane's
// four consecutive accumulator rows are the 2 x 2 output block of one channel at (2 iy, 2 ix): two 8-byte stores per channel,
// whole lines per wave -- convt_quad_resident_kernel (one WAVE per class) stores 4-byte words at stride 8 and ran at
// 2.5-3.2 TB/s.  The frame is staged once ([piece][channel octet][haloed position] 16-byte rows, zero border written once per
// launch), the weights are A operands in registers for the whole launch (gathered from the four packed 2 x 2 sub-kernels the
// quad entry already takes), a wave multiplies one row tile x two 32-position units (two independent chains, one A fragment).
// ------------------------------------------------------------------------------------------------
// Workgroups per CU: a workgroup's waves stage (VALU) and multiply (MFMA) in lockstep around its one image buffer; where the
// registers allow it, several workgroups per CU run those phases against each other (16 -> 8 with act': 100 -> 88 us; the
// 32 -> 16 shape as four waves x two jobs, which would fit twice, spills 100+ registers).
__host__ __device__ constexpr int convt_rows_wgs(int cin, int cout, int plane) {
  const int waves = (plane / 64) * (cout / 8), a_regs = 9 * (cin / 16) * 8;
  return a_regs > 160 ? 1 : (waves <= 2 ? 4 : (waves <= 4 ? 2 : 1));
}
template <int CIN, int COUT, int PLANE, bool EPI>
__global__ __launch_bounds__(64 * (PLANE / 64) * (COUT / 8), (convt_rows_wgs(CIN, COUT, PLANE) * (PLANE / 64) * (COUT / 8) + 3) / 4)  // (threads, waves per SIMD)
void convt4s2_rows_kernel(const QuadProblem pa, const QuadProblem pb) {
  constexpr int UP = PLANE / 64, CTS = COUT / 8, NW = UP * CTS, NT = 64 * NW;   // unit pairs x row tiles = waves
  constexpr int OCT = CIN / 8, GPT = CIN / 16, KB = 9 * GPT;
  constexpr int NITEM = (PLANE / 4) * (CIN / 4) / NT;             // (4 positions x 4 channels) items per thread
  static_assert(NITEM * NT == (PLANE / 4) * (CIN / 4) && NITEM >= 1 && UP >= 1 && CTS >= 1, "whole items");
  const bool second = blockIdx.x >= (unsigned)pa.nx;  // workgroup-uniform
  // (every field is selected by itself: a reference to "second ? pb : pa" made the compiler copy both structs to scratch here)
#define MTRSSM_PICK(f) (second ? pb.f : pa.f)
  struct {
    const float* src;
    const float* bias;
    const float* actgrad;
    float* out;
  } P{MTRSSM_PICK(src), MTRSSM_PICK(bias), MTRSSM_PICK(actgrad), MTRSSM_PICK(out)};
  const int wg = second ? (int)blockIdx.x - pa.nx : (int)blockIdx.x, nwg = MTRSSM_PICK(nx);
  const int Ws = MTRSSM_PICK(g[0].Ws), Hs = MTRSSM_PICK(g[0].Hs);   // host: Hs * Ws == PLANE, Ws a power of two
  const int wsh = 31 - __builtin_clz(Ws);
  const int ntiles = MTRSSM_PICK(g[0].N);
  const int g_act = MTRSSM_PICK(g[0].act), g_pre = MTRSSM_PICK(g[0].pre_act);
  const int PW = Ws + 2;
  const int NPOS = (Hs + 2) * PW + 1;
  const int OSZ = NPOS * 16, IMG = OCT * OSZ;               // one octet plane / one piece
  const int Wo = 2 * Ws;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, il = lane & 31, kl = lane >> 5;
  if (wg >= ntiles) return;
  extern __shared__ __attribute__((aligned(16))) unsigned char band_lds[];  // [2 pieces][OCT][NPOS] x 16 B
  const int up = wave % UP, ct = wave / UP;

  // A operands: row il of tile ct = (output channel co, class q); k-block kb = (neighbour t = kb / GPT, channel group kb % GPT);
  // class q reads neighbour (dy, dx) with tap (ty, tx) = (OFFY_q - dy, OFFX_q - dx) of its packed 2 x 2 sub-kernel
  bf16x8 a[KB][2];
  {
    const int R = 32 * ct + il, co = R >> 2, q = R & 3;
    const unsigned short* wq = q == 0 ? MTRSSM_PICK(wq[0]) : (q == 1 ? MTRSSM_PICK(wq[1]) : (q == 2 ? MTRSSM_PICK(wq[2]) : MTRSSM_PICK(wq[3])));
    const int offy = q == 0 ? MTRSSM_PICK(g[0].OFFY) : (q == 1 ? MTRSSM_PICK(g[1].OFFY) : (q == 2 ? MTRSSM_PICK(g[2].OFFY) : MTRSSM_PICK(g[3].OFFY)));
    const int offx = q == 0 ? MTRSSM_PICK(g[0].OFFX) : (q == 1 ? MTRSSM_PICK(g[1].OFFX) : (q == 2 ? MTRSSM_PICK(g[2].OFFX) : MTRSSM_PICK(g[3].OFFX)));
#undef MTRSSM_PICK
    const size_t piece = (size_t)32 * 4 * CIN;  // host: CoutPad == 32, Cpad == CIN
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      const int t = kb / GPT, dy = t / 3 - 1, dx = t % 3 - 1;
      const int ty = offy - dy, tx = offx - dx;
      const bool live = ty >= 0 && ty < 2 && tx >= 0 && tx < 2;
      const size_t wofs = ((size_t)co * 4 + (live ? ty * 2 + tx : 0)) * CIN + 16 * (kb % GPT) + 8 * kl;  // unconditional requests, selected afterwards
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(wq + s * piece + wofs);
        a[kb][s] = __builtin_bit_cast(bf16x8, live ? v : u32x4{0u, 0u, 0u, 0u});
      }
    }
  }
  for (int o = tid * 16; o < 2 * IMG; o += NT * 16) *reinterpret_cast<u32x4*>(band_lds + o) = u32x4{0u, 0u, 0u, 0u};

  band_f4 pv[NITEM][4];
  auto request = [&](int tile) {
    const __amdgpu_buffer_rsrc_t rsf =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(P.src + (size_t)tile * CIN * PLANE), 0, 0x7fffffff, 0x00020000);
#pragma unroll
    for (int k = 0; k < NITEM; ++k) {
      const int it = tid + NT * k, pg = it % (PLANE / 4), cq = it / (PLANE / 4);
#pragma unroll
      for (int c = 0; c < 4; ++c)
        pv[k][c] = __builtin_bit_cast(band_f4, __builtin_amdgcn_raw_buffer_load_b128(rsf, ((4 * cq + c) * PLANE + 4 * pg) * 4, 0, 0));
    }
  };
  const int act = g_act;
  const int mode = g_pre == 0 ? 0 : (act == MTRSSM_ACT_ELU ? 1 : 2);
  auto stage_as = [&](auto mode_tag) {
    constexpr int MODE = decltype(mode_tag)::value;
    auto activate = [&](float v) {
      if (MODE == 1) v = elu_fast(v);
      if (MODE == 2) v = act_fwd(v, act);
      return v;
    };
#pragma unroll
    for (int k = 0; k < NITEM; ++k) {
      const int it = tid + NT * k, pg = it % (PLANE / 4), cq = it / (PLANE / 4);
      const int p = pg * 4, r = p >> wsh, x = p & (Ws - 1);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        unsigned short h[4][2];
#pragma unroll
        for (int c = 0; c < 4; ++c) split_bf16<2>(activate(pv[k][c][e]), h[c]);
        unsigned char* d = band_lds + (cq >> 1) * OSZ + ((r + 1) * PW + x + e + 1) * 16 + (cq & 1) * 8;
        *reinterpret_cast<uint2*>(d) = make_uint2((unsigned)h[0][0] | ((unsigned)h[1][0] << 16), (unsigned)h[2][0] | ((unsigned)h[3][0] << 16));
        *reinterpret_cast<uint2*>(d + IMG) = make_uint2((unsigned)h[0][1] | ((unsigned)h[1][1] << 16), (unsigned)h[2][1] | ((unsigned)h[3][1] << 16));
      }
    }
  };
  auto stage = [&]() {
    if (mode == 0) stage_as(std::integral_constant<int, 0>{});
    else if (mode == 1) stage_as(std::integral_constant<int, 1>{});
    else stage_as(std::integral_constant<int, 2>{});
  };

  // this wave's two units: input position j = 64 up + 32 u + il -> (iy, ix); the lane's accumulator rows 4 i + q are the classes
  // of output channel 8 ct + kl + 2 i
  unsigned bb[2];
  int opx[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int j = 64 * up + 32 * u + il, iy = j >> wsh, ix = j & (Ws - 1);
    bb[u] = (unsigned)(kl * OSZ + ((iy + 1) * PW + ix + 1) * 16);
    opx[u] = (2 * iy) * Wo + 2 * ix;
  }
  const float* __restrict__ actgrad = P.actgrad;  // host: non-null exactly in the EPI instantiation
  const float* __restrict__ bias = P.bias;
  const bool act_elu = act == MTRSSM_ACT_ELU, act_relu = act == MTRSSM_ACT_RELU;
  float bv[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) bv[i] = bias ? bias[8 * ct + kl + 2 * i] : 0.f;
  auto frag_off = [&](int kb) {
    const int t = kb / GPT;
    return ((t / 3 - 1) * PW + (t % 3 - 1)) * 16 + (2 * (kb % GPT)) * OSZ;
  };
  auto compute = [&](int tile) {
    const size_t obase = ((size_t)tile * COUT + 8 * ct + kl) * (size_t)(4 * PLANE);
    float2 gv[EPI ? 2 : 1][EPI ? 8 : 1];
    if constexpr (EPI) {
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float* gp = actgrad + obase + (size_t)(2 * i) * (4 * PLANE) + opx[u];
          gv[u][2 * i] = *reinterpret_cast<const float2*>(gp);
          gv[u][2 * i + 1] = *reinterpret_cast<const float2*>(gp + Wo);
        }
    }
    f32x16 acc[2];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[u][r] = bv[r >> 2];
    bf16x8 q0[2][2], q1[2][2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      q0[0][u] = *reinterpret_cast<const bf16x8*>(band_lds + (int)bb[u] + frag_off(0));
      q1[0][u] = *reinterpret_cast<const bf16x8*>(band_lds + (int)bb[u] + frag_off(0) + IMG);
    }
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      const int st = kb & 1;
      if (kb + 1 < KB) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          q0[st ^ 1][u] = *reinterpret_cast<const bf16x8*>(band_lds + (int)bb[u] + frag_off(kb + 1));
          q1[st ^ 1][u] = *reinterpret_cast<const bf16x8*>(band_lds + (int)bb[u] + frag_off(kb + 1) + IMG);
        }
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[kb][0], q1[st][u], acc[u], 0, 0, 0);
#pragma unroll
      for (int u = 0; u < 2; ++u) acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[kb][1], q0[st][u], acc[u], 0, 0, 0);
#pragma unroll
      for (int u = 0; u < 2; ++u) acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[kb][0], q0[st][u], acc[u], 0, 0, 0);
    }
    auto egrad = [&](float x) {
      const float e = __expf(x);
      const float neg = act_elu ? e : (act_relu ? 0.f : 1.f);
      return x > 0.f ? 1.f : neg;
    };
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float2 top = make_float2(acc[u][4 * i], acc[u][4 * i + 1]), bot = make_float2(acc[u][4 * i + 2], acc[u][4 * i + 3]);
        if constexpr (EPI) {
          top.x *= egrad(gv[u][2 * i].x);
          top.y *= egrad(gv[u][2 * i].y);
          bot.x *= egrad(gv[u][2 * i + 1].x);
          bot.y *= egrad(gv[u][2 * i + 1].y);
        }
        float* op = P.out + obase + (size_t)(2 * i) * (4 * PLANE) + opx[u];
        *reinterpret_cast<float2*>(op) = top;
        *reinterpret_cast<float2*>(op + Wo) = bot;
      }
  };
  const int last = wg + ((ntiles - 1 - wg) / nwg) * nwg;
  request(wg);
  __syncthreads();  // the zero fill
  for (int tile = wg; tile < ntiles; tile += nwg) {
    stage();
    const int nx = tile + nwg;
    request(nx < last ? nx : last);
    lds_barrier();
    compute(tile);
    lds_barrier();
  }
}

template <int CIN, int PLANE>
__host__ __device__ constexpr size_t convt_rows_lds_bytes() {
  // haloed positions: the widest plane of the shape (PLANE = 64: 18 x 6, 256: 34 x 10, 1024: 66 x 18) + 1
  return (size_t)2 * (CIN / 8) * ((PLANE == 64 ? 108 : (PLANE == 256 ? 340 : 1188)) + 1) * 16;
}

}  // namespace mtrssm
