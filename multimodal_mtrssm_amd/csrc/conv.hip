// Conv encoder / decoder kernels for gfx950 (MI355X): fp32 MFMA implicit GEMM over all B*T frames.
//
// One gather-GEMM kernel serves Conv2d forward, Conv2d backward-data, ConvTranspose2d forward and
// ConvTranspose2d backward-data (NCHW, fp32):
//
//     out[n, co, oy*OS+QY, ox*OS+QX] = epi( bias[co] + sum_{tap=(ty,tx)} sum_c Wp[co][tap][c] * pre(src[n, c, sy, sx]) )
//     sy = oy*SS + ty*TS + OFFY,   sx = ox*SS + tx*TS + OFFX        (zero outside the source plane)
//
//   Conv2d forward          : SS = stride, TS = +1, OFF = -pad, OS = 1
//   transposed gather (s=1) : SS = 1, TS = -1, OFF = +pad
//   transposed gather (s>1) : one launch per output parity class (qy,qx): only the taps that can reach
//                             that class are visited (sub-pixel decomposition -> no multiplies by zero),
//                             SS = 1, TS = -1, OS = stride, QY = qy
//
// As a GEMM: D[co][pixel] = sum_k A[co][k] B[k][pixel], k = (tap, c).  Each wave owns 32 pixels x TCO
// channels and issues v_mfma_f32_32x32x2_f32 (exact fp32: bitwise an fma chain, 64 FLOP/clk/SIMD), the
// A operand = packed weights [co][k] and the B operand = the gathered patch [k][pixel], both staged
// through LDS in 16-channel chunks (double-buffered, one barrier per chunk).  Lanes = pixels on the
// output side, so every store is a coalesced 128-byte row segment of one channel plane.
//
// pre() fuses the activation that precedes the layer (the stacks are "act -> conv" everywhere, see
// oracle/ref_cnn.py), epi() fuses bias and, for the backward-data use, the multiplication by act'(x).
//
// The weight-gradient kernel is the transposed problem: dWp[co][tap][c] += sum_pixels A[co][pix] G[pix][c],
// reduction over all N*H*W pixels split across workgroups, partial tiles combined with fp32 atomics.
#include <map>
#include <mutex>
#include <unordered_map>
#include <utility>

#include "scan_common.h"

namespace mtrssm {

void set_error(const char* fmt, ...);
void set_last_kernel(const char* name);

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int kKC = 16;          // channels per K chunk
constexpr int kTP = 128;         // pixels per workgroup tile (4 waves x 32)
constexpr int kConvThreads = 256;

__device__ __forceinline__ float act_grad_from_in(float x, int act) {
  switch (act) {
    case MTRSSM_ACT_RELU: return x > 0.f ? 1.f : 0.f;
    case MTRSSM_ACT_ELU: return x > 0.f ? 1.f : expf(x);
    case MTRSSM_ACT_TANH: { const float t = tanhf(x); return 1.f - t * t; }
    default: return 1.f;
  }
}

// ------------------------------------------------------------------------------------------------
// gather-GEMM
// ------------------------------------------------------------------------------------------------
template <int NT>  // NT = number of 32-channel output tiles per workgroup (1 or 2)
__global__ __launch_bounds__(kConvThreads) void conv_gather_gemm_kernel(
    const MtrssmConvGeom g, const float* __restrict__ src, const float* __restrict__ src2, const float* __restrict__ wp,
    const float* __restrict__ bias, const float* __restrict__ actgrad_in, const float* __restrict__ add_in, float* __restrict__ out) {
  constexpr int TCO = 32 * NT;
  constexpr int LDW = kKC + 1;
  __shared__ float g_lds[2][kKC][kTP];
  __shared__ float w_lds[2][TCO][LDW];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int taps = g.KH * g.KW;
  const int ctot = g.C + g.C2;
  const int nchunk_c = g.Cpad / kKC;
  const int nchunks = taps * nchunk_c;
  const int plane_s = g.Hs * g.Ws;
  const long ptot = (long)g.N * g.Hq * g.Wq;
  const long p0 = (long)blockIdx.x * kTP;
  const int co0 = blockIdx.y * TCO;

  // --- staging roles: thread -> (pixel column, row group) for the patch; (row, 4-float column) for the weights
  const int pc = tid & (kTP - 1), rg = tid >> 7;
  const long pst = p0 + pc;
  const bool pvalid = pst < ptot;
  int sn = 0, soy = 0, sox = 0;
  if (pvalid) {
    sn = (int)(pst / (g.Hq * g.Wq));
    const int rem = (int)(pst - (long)sn * g.Hq * g.Wq);
    soy = rem / g.Wq;
    sox = rem - soy * g.Wq;
  }
  const float* src_n = src + (size_t)sn * g.C * plane_s;
  const int wrow = tid >> 2, wcol = (tid & 3) * 4;  // 64 rows x 16 floats per pass

  float greg[kKC / 2];
  float4 wv = make_float4(0.f, 0.f, 0.f, 0.f);

  auto load_chunk = [&](int chunk) {
    const int tap = chunk / nchunk_c, c0 = (chunk - tap * nchunk_c) * kKC;
    const int ty = tap / g.KW, tx = tap - ty * g.KW;
    const int sy = soy * g.SS + ty * g.TS + g.OFFY, sx = sox * g.SS + tx * g.TS + g.OFFX;
    const bool ok = pvalid && sy >= 0 && sy < g.Hs && sx >= 0 && sx < g.Ws;
    const int off = sy * g.Ws + sx;
#pragma unroll
    for (int i = 0; i < kKC / 2; ++i) {
      const int c = c0 + rg + 2 * i;
      float v = 0.f;
      if (ok && c < ctot) {
        v = c < g.C ? src_n[(size_t)c * plane_s + off] : src2[(size_t)(c - g.C) * plane_s + off];
        if (g.pre_act) v = act_fwd(v, g.act);
      }
      greg[i] = v;
    }
    if (wrow < TCO) {
      const float* wsrc = wp + ((size_t)(co0 + wrow) * taps + tap) * g.Cpad + c0 + wcol;
      wv = *reinterpret_cast<const float4*>(wsrc);
    }
  };
  auto store_chunk = [&](int buf) {
#pragma unroll
    for (int i = 0; i < kKC / 2; ++i) g_lds[buf][rg + 2 * i][pc] = greg[i];
    if (wrow < TCO) {
      float* d = &w_lds[buf][wrow][wcol];
      d[0] = wv.x; d[1] = wv.y; d[2] = wv.z; d[3] = wv.w;
    }
  };

  f32x16 acc[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  if (nchunks > 0) {
    load_chunk(0);
    store_chunk(0);
  }
  __syncthreads();
  const int kl = lane >> 5, il = lane & 31;
  for (int chunk = 0; chunk < nchunks; ++chunk) {
    const int buf = chunk & 1;
    if (chunk + 1 < nchunks) load_chunk(chunk + 1);
#pragma unroll
    for (int step = 0; step < kKC / 2; ++step) {
      const float b = g_lds[buf][2 * step + kl][wave * 32 + il];
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const float a = w_lds[buf][j * 32 + il][2 * step + kl];
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
      }
    }
    if (chunk + 1 < nchunks) store_chunk(buf ^ 1);
    __syncthreads();
  }

  // --- epilogue: lane = pixel, registers = channels
  const long pe = p0 + wave * 32 + il;
  if (pe < ptot) {
    const int n = (int)(pe / (g.Hq * g.Wq));
    const int rem = (int)(pe - (long)n * g.Hq * g.Wq);
    const int oy = rem / g.Wq, ox = rem - oy * g.Wq;
    const size_t plane_o = (size_t)g.Ho * g.Wo;
    const size_t base = (size_t)n * g.Cout * plane_o + (size_t)(oy * g.OS + g.QY) * g.Wo + (ox * g.OS + g.QX);
#pragma unroll
    for (int j = 0; j < NT; ++j) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + j * 32 + (r & 3) + 8 * (r >> 2) + 4 * kl;
        if (co < g.Cout) {
          float v = acc[j][r] + (bias ? bias[co] : 0.f);
          const size_t o = base + (size_t)co * plane_o;
          if (actgrad_in) v *= act_grad_from_in(actgrad_in[o], g.act);
          if (add_in) v += add_in[o];
          out[o] = v;
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// weight gradient:  dwp[co][tap][c] += sum_{n,y,x} preA(a[n,co,y,x]) * preG(src[n,c,y*SS+ty+OFFY, x*SS+tx+OFFX])
// grid: (c tiles of 32, taps, pixel splits); each wave reduces its own 32-pixel slices; tiles of TCO=64
// output rows are looped inside (co tiles), partial sums combined through LDS then one atomic per element.
// ------------------------------------------------------------------------------------------------
template <int NT>
__global__ __launch_bounds__(kConvThreads) void conv_weight_grad_kernel(
    const MtrssmConvGeom g, const float* __restrict__ a, const float* __restrict__ src, const float* __restrict__ src2,
    const int pre_act_a, float* __restrict__ dwp) {
  constexpr int TCO = 32 * NT;
  constexpr int LDP = 33;
  __shared__ float a_lds[4][TCO][LDP];
  __shared__ float g_lds[4][32][LDP];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int kl = lane >> 5, il = lane & 31;
  const int taps = g.KH * g.KW;
  const int ctot = g.C + g.C2;
  const int c0 = blockIdx.x * 32;
  const int tap = blockIdx.y;
  const int ty = tap / g.KW, tx = tap - ty * g.KW;
  const int plane_s = g.Hs * g.Ws, plane_a = g.Hq * g.Wq;
  const long ptot = (long)g.N * plane_a;
  const int nsplit = gridDim.z;
  // this workgroup's pixel range, in units of 128-pixel groups
  const long groups = (ptot + kTP - 1) / kTP;
  const long gper = (groups + nsplit - 1) / nsplit;
  const long gbeg = (long)blockIdx.z * gper;
  const long gend = gbeg + gper < groups ? gbeg + gper : groups;

  for (int cot = 0; cot < g.CoutPad; cot += TCO) {
    f32x16 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    for (long grp = gbeg; grp < gend; ++grp) {
      // each wave stages its own 32 pixels: lane -> (pixel il, row parity kl)
      const long p = grp * kTP + wave * 32 + il;
      const bool pv = p < ptot;
      int n = 0, y = 0, x = 0;
      if (pv) {
        n = (int)(p / plane_a);
        const int rem = (int)(p - (long)n * plane_a);
        y = rem / g.Wq;
        x = rem - y * g.Wq;
      }
      const int sy = y * g.SS + ty * g.TS + g.OFFY, sx = x * g.SS + tx * g.TS + g.OFFX;
      const bool ok = pv && sy >= 0 && sy < g.Hs && sx >= 0 && sx < g.Ws;
      const float* a_n = a + (size_t)n * g.Cout * plane_a + (size_t)y * g.Wq + x;
      const float* s_n = src + (size_t)n * g.C * plane_s + (size_t)sy * g.Ws + sx;
#pragma unroll 4
      for (int i = 0; i < TCO / 2; ++i) {
        const int co = cot + kl + 2 * i;
        float v = 0.f;
        if (pv && co < g.Cout) {
          v = a_n[(size_t)co * plane_a];
          if (pre_act_a) v = act_fwd(v, g.act);
        }
        a_lds[wave][kl + 2 * i][il] = v;
      }
#pragma unroll 4
      for (int i = 0; i < 16; ++i) {
        const int c = c0 + kl + 2 * i;
        float v = 0.f;
        if (ok && c < ctot) {
          v = c < g.C ? s_n[(size_t)c * plane_s] : src2[(size_t)(c - g.C) * plane_s + (size_t)sy * g.Ws + sx];
          if (g.pre_act) v = act_fwd(v, g.act);
        }
        g_lds[wave][kl + 2 * i][il] = v;
      }
      __syncthreads();
#pragma unroll
      for (int step = 0; step < 16; ++step) {
        const float b = g_lds[wave][il][2 * step + kl];  // B[k = pixel][j = c]
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const float av = a_lds[wave][j * 32 + il][2 * step + kl];  // A[i = co][k = pixel]
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b, acc[j], 0, 0, 0);
        }
      }
      __syncthreads();
    }

    // combine the four waves' partial tiles through LDS (reuse a_lds as [4][TCO*32] floats), then atomics
    float* red = &a_lds[0][0][0];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = j * 32 + (r & 3) + 8 * (r >> 2) + 4 * kl;  // co within tile
        red[(wave * TCO + row) * LDP + il] = acc[j][r];             // col = c within tile
      }
    __syncthreads();
    for (int e = tid; e < TCO * 32; e += kConvThreads) {
      const int row = e >> 5, col = e & 31;
      const float s = red[(0 * TCO + row) * LDP + col] + red[(1 * TCO + row) * LDP + col] + red[(2 * TCO + row) * LDP + col] +
                      red[(3 * TCO + row) * LDP + col];
      const int co = cot + row, c = c0 + col;
      if (co < g.CoutPad && c < g.Cpad && s != 0.f) atomicAdd(&dwp[((size_t)co * taps + tap) * g.Cpad + c], s);
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------
// weight gradient, patch-staged (v2).  Same result as conv_weight_grad_kernel, different data flow:
//   * a workgroup owns a contiguous range of 64-pixel GROUPS (whole rows of one frame, or whole small
//     frames) and ALL (tap, 32-channel) column tiles of dwp for one 32*NT-row block of output channels;
//   * per group it stages the A tile [32*NT x 64 pixels] and ONE zero-haloed source patch
//     [channels][rows with halo][cols with halo] in LDS; every tap reads the same patch at a shifted
//     offset, so each source element is fetched once instead of once per tap and each A element feeds
//     every column tile (18 for a 64-channel 3x3) instead of one;
//   * the four waves split the column tiles (<= kMaxQ each, accumulators in registers across the whole
//     pixel range), and finish with fp32 atomics shaped as 128-byte row segments.
// Applicable when 64 % Wq == 0 and groups tile frames exactly (host-checked); otherwise v1 runs.
// ------------------------------------------------------------------------------------------------
constexpr int kGP = 64;    // pixels per group
constexpr int kMaxQ = 5;   // column tiles per wave
constexpr int kLDA = kGP + 1;

struct PatchGeom {
  int rpg, rp, ipg, ph, pw, ps_raw, ps;
  __host__ __device__ PatchGeom(const MtrssmConvGeom& g, int gp) {
    rpg = gp / g.Wq;
    rp = rpg < g.Hq ? rpg : g.Hq;
    ipg = rpg / rp;
    ph = (rp - 1) * g.SS + g.KH;
    pw = (g.Wq - 1) * g.SS + g.KW;
    ps_raw = ipg * ph * pw;
    ps = ps_raw | 1;  // odd channel stride: conflict-free when lanes index channels
  }
};

template <int NT>
__global__ __launch_bounds__(2 * kConvThreads) void conv_weight_grad_patch_kernel(
    const MtrssmConvGeom g, const float* __restrict__ a, const float* __restrict__ src, const float* __restrict__ src2,
    const int pre_act_a, float* __restrict__ dwp, float* __restrict__ dbias) {
  // 8 waves: waves 0-3 issue MFMAs on LDS buffer `cur`, waves 4-7 stage the next group into the other buffer
  // (wave specialisation: each SIMD holds one compute wave and one loader wave), one barrier per group.
  constexpr int TCO = 32 * NT;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const PatchGeom pg(g, kGP);
  const int ctot = g.C + g.C2;
  const int buf_floats = TCO * kLDA + ctot * pg.ps;
  int* pixtab = reinterpret_cast<int*>(lds + 2 * (size_t)buf_floats);  // [kGP]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool loader = wave >= 4;
  const int lw = wave & 3;
  const int kl = lane >> 5, il = lane & 31;
  const int taps = g.KH * g.KW;
  const int nq = (taps * ctot + 31) / 32;
  const int co0 = blockIdx.y * TCO;
  const int plane_s = g.Hs * g.Ws, plane_a = g.Hq * g.Wq;
  const long ptot = (long)g.N * plane_a;
  const long groups = (ptot + kGP - 1) / kGP;
  const long gper = (groups + gridDim.x - 1) / gridDim.x;
  const long gbeg = (long)blockIdx.x * gper;
  const long gend = gbeg + gper < groups ? gbeg + gper : groups;

  if (tid < kGP) {
    const int row = tid / g.Wq, ox = tid - row * g.Wq;
    const int ip = row / pg.rp, lr = row - ip * pg.rp;
    pixtab[tid] = ip * pg.ph * pg.pw + lr * g.SS * pg.pw + ox * g.SS;
  }

  // ---- loader role: stage one group (A tile + zero-haloed patch) into buffer `buf`.
  // LDS-DMA (global_load_lds_dword: global -> LDS, no VGPR staging): a loader wave keeps ALL of its ~48 row loads of
  // the group in flight at once -- with register staging next to 160 accumulator registers it could hold 8, and the
  // MFMA waves spent half their time at the barrier waiting for it (SQ_WAIT_ANY 52 %, profiles/round1_notes.md).
  // One DMA instruction writes 64 consecutive floats at a wave-uniform LDS base: rows of the A tile (lane = pixel) and
  // 64-position runs of one patch channel (lane = patch position).  Lanes with nothing to fetch (padding halo, tail)
  // are exec-masked out of the DMA and store a zero instead; the fused activation runs in place once the data landed.
  auto dma = [&](const float* gsrc, float* ldst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)ldst, 4, 0, 0);
  };
  auto stage = [&](long grp, int buf) {
    float* a_lds = lds + (size_t)buf * buf_floats;
    float* patch = a_lds + TCO * kLDA;
    const long p0 = grp * kGP;
    {  // A tile: loader wave lw owns rows lw, lw+4, ...; lane = pixel
      const long p = p0 + lane;
      const bool pv = p < ptot;
      int n = 0, rem = 0;
      if (pv) { n = (int)(p / plane_a); rem = (int)(p - (long)n * plane_a); }
      const float* a_n = a + (size_t)n * g.Cout * plane_a + rem;
      for (int row = lw; row < TCO; row += 4) {
        const int co = co0 + row;
        if (co >= g.Cout) break;  // rows beyond Cout feed MFMA rows that are never stored
        float* dst = a_lds + row * kLDA;
        if (pv) dma(a_n + (size_t)co * plane_a, dst);
        else dst[lane] = 0.f;
      }
    }
    const int n0 = (int)(p0 / plane_a);
    const int r0 = (int)((p0 - (long)n0 * plane_a) / g.Wq);
    const int sy0 = r0 * g.SS + g.OFFY, sx0 = g.OFFX;
    const int phw = pg.ph * pg.pw;
    for (int rb = 0; rb < pg.ps_raw; rb += 64) {
      const int r = rb + lane;
      const bool rv = r < pg.ps_raw;
      const int ip = r / phw, q = r - ip * phw;
      const int pr = q / pg.pw, pcn = q - pr * pg.pw;
      const int n = n0 + ip, sy = sy0 + pr, sx = sx0 + pcn;
      const bool ok = rv && n < g.N && sy >= 0 && sy < g.Hs && sx >= 0 && sx < g.Ws;
      const size_t off = (size_t)sy * g.Ws + sx;
      const float* s_n = src + (size_t)n * g.C * plane_s + off;
      for (int c = lw; c < ctot; c += 4) {
        float* dst = patch + c * pg.ps + rb;
        if (ok) dma(c < g.C ? s_n + (size_t)c * plane_s : src2 + (size_t)(c - g.C) * plane_s + off, dst);
        else if (rv) dst[lane] = 0.f;
      }
    }
    if (g.pre_act || pre_act_a) {  // in-place activation once the DMA data has landed (this wave's own rows only)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (pre_act_a) {
        for (int row = lw; row < TCO && co0 + row < g.Cout; row += 4) {
          float* d = a_lds + row * kLDA + lane;
          *d = act_fwd(*d, g.act);
        }
      }
      if (g.pre_act) {
        for (int rb = 0; rb < pg.ps_raw; rb += 64) {
          if (rb + lane < pg.ps_raw) {
            for (int c = lw; c < ctot; c += 4) {
              float* d = patch + c * pg.ps + rb + lane;
              *d = act_fwd(*d, g.act);
            }
          }
        }
      }
    }
  };

  // ---- compute role state: this wave's column tiles.  Columns are the FLATTENED (tap, channel) index
  // f = tap * ctot + c, cut into tiles of 32 (tile q = lw + 4 s): a 5-channel 3x3 layer needs 2 tiles, not 9.
  int cbase[kMaxQ];
  int nsl = 0;
#pragma unroll
  for (int s = 0; s < kMaxQ; ++s) {
    const int q = lw + 4 * s;
    cbase[s] = -1;
    if (q < nq) {
      nsl = s + 1;
      const int f = q * 32 + il;
      if (f < taps * ctot) {
        const int tap = f / ctot, c = f - tap * ctot;
        const int ty = tap / g.KW, tx = tap - ty * g.KW;
        cbase[s] = c * pg.ps + ty * pg.pw + tx;
      }
    }
  }
  f32x16 acc[kMaxQ][NT];
#pragma unroll
  for (int s = 0; s < kMaxQ; ++s)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[s][j][r] = 0.f;
  float bsum[NT];  // bias gradient = row sums of the A tile (raw a, never pre-activated); kept by compute wave 0
#pragma unroll
  for (int j = 0; j < NT; ++j) bsum[j] = 0.f;
  const bool do_bias = dbias != nullptr && wave == 0;

  __syncthreads();  // pixtab
  int cur = 1;      // iteration gbeg-1 only stages group gbeg into buffer 0 (single call site of stage())
  for (long grp = gbeg - 1; grp < gend; ++grp) {
    if (loader) {
      if (grp + 1 < gend) stage(grp + 1, cur ^ 1);
    } else if (grp >= gbeg) {
      const float* a_lds = lds + (size_t)cur * buf_floats;
      const float* patch = a_lds + TCO * kLDA;
#pragma unroll 2
      for (int step = 0; step < kGP / 2; ++step) {
        const int pix = 2 * step + kl;
        const int poff = pixtab[pix];
        float av[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) av[j] = a_lds[(j * 32 + il) * kLDA + pix];
        if (do_bias) {
#pragma unroll
          for (int j = 0; j < NT; ++j) bsum[j] += av[j];
        }
#pragma unroll
        for (int s = 0; s < kMaxQ; ++s) {
          if (s < nsl) {
            const float b = cbase[s] >= 0 ? patch[cbase[s] + poff] : 0.f;
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[s][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], b, acc[s][j], 0, 0, 0);
          }
        }
      }
    }
    __syncthreads();
    cur ^= 1;
  }
  if (do_bias) {  // lanes il and il+32 hold the even / odd pixels of row il
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const float tot = bsum[j] + __shfl_xor(bsum[j], 32, kWave);
      const int co = co0 + j * 32 + il;
      if (kl == 0 && co < g.Cout) atomicAdd(&dbias[co], tot);
    }
  }

  if (!loader) {
#pragma unroll
    for (int s = 0; s < kMaxQ; ++s) {
      const int f = (lw + 4 * s) * 32 + il;
      if (lw + 4 * s < nq && f < taps * ctot) {
        const int tap = f / ctot, c = f - tap * ctot;
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int co = co0 + j * 32 + (r & 3) + 8 * (r >> 2) + 4 * kl;
            if (co < g.Cout) atomicAdd(&dwp[((size_t)co * taps + tap) * g.Cpad + c], acc[s][j][r]);
          }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// gather-GEMM, patch-staged (v2).  One workgroup = one 128-pixel group (whole rows of a frame, or whole
// small frames) x 32*NT output channels.  Per 16-channel chunk the zero-haloed source patch is staged ONCE
// (activation applied once per element) and every tap reads it at a shifted offset; v1 re-gathers and
// re-activates the same pixels for each of the KH*KW taps, which made it VALU/L1-bound.
// (A wave-specialised persistent variant -- 4 MFMA + 4 loader waves, one workgroup per CU, as the weight-gradient
// kernel uses -- was measured 2x SLOWER here: with only two accumulators per wave a single MFMA wave per SIMD does not
// keep the pipe fed; six co-resident workgroups of this kernel do.  rocprof: profiles/round1_notes.md.)
// ------------------------------------------------------------------------------------------------
template <int NT, int NP>  // NT 32-channel tiles x NP 32-pixel tiles per wave; workgroup tile = 128*NP pixels x 32*NT channels
__global__ __launch_bounds__(kConvThreads) void conv_gather_gemm_patch_kernel(
    const MtrssmConvGeom g, const float* __restrict__ src, const float* __restrict__ src2, const float* __restrict__ wp,
    const float* __restrict__ bias, const float* __restrict__ actgrad_in, const float* __restrict__ add_in, float* __restrict__ out) {
  constexpr int TCO = 32 * NT;
  constexpr int TPX = kTP * NP;
  constexpr int LDW = kKC + 1;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const PatchGeom pg(g, TPX);
  float* patch = lds;                                            // [kKC][ps]
  float* w_lds = patch + (size_t)kKC * pg.ps;                    // [2][TCO][LDW]
  int* rtab = reinterpret_cast<int*>(w_lds + 2 * TCO * LDW);     // [ps_raw]: (frame-in-group << 26) | plane offset, or -1

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int kl = lane >> 5, il = lane & 31;
  const int taps = g.KH * g.KW;
  const int ctot = g.C + g.C2;
  const int plane_s = g.Hs * g.Ws, plane_q = g.Hq * g.Wq;
  const long ptot = (long)g.N * plane_q;
  const long p0 = (long)blockIdx.x * TPX;
  const int co0 = blockIdx.y * TCO;
  const int n0 = (int)(p0 / plane_q);
  const int r0 = (int)((p0 - (long)n0 * plane_q) / g.Wq);

  // patch position -> source offset, once per workgroup
  {
    const int khm = g.TS > 0 ? 0 : g.KH - 1, kwm = g.TS > 0 ? 0 : g.KW - 1;
    const int sy0 = r0 * g.SS + g.OFFY - khm, sx0 = g.OFFX - kwm;
    const int phw = pg.ph * pg.pw;
    for (int r = tid; r < pg.ps_raw; r += kConvThreads) {
      const int ip = r / phw, q = r - ip * phw;
      const int pr = q / pg.pw, pcn = q - pr * pg.pw;
      const int sy = sy0 + pr, sx = sx0 + pcn;
      const bool ok = n0 + ip < g.N && sy >= 0 && sy < g.Hs && sx >= 0 && sx < g.Ws;
      rtab[r] = ok ? ((ip << 26) | (sy * g.Ws + sx)) : -1;
    }
  }
  // this lane's pixels -> patch offsets (B operand: lane il = pixel); wave w owns pixels [w*32*NP, (w+1)*32*NP)
  int pixoff[NP];
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int pix = (wave * NP + i) * 32 + il;
    const int row = pix / g.Wq, ox = pix - row * g.Wq;
    const int ip = row / pg.rp, lr = row - ip * pg.rp;
    pixoff[i] = ip * pg.ph * pg.pw + lr * g.SS * pg.pw + ox * g.SS + kl * pg.ps;
  }
  const int wrow = tid >> 2, wcol = (tid & 3) * 4;
  float4 wv = make_float4(0.f, 0.f, 0.f, 0.f);
  auto load_w = [&](int tap, int c0) {
    if (wrow < TCO) wv = *reinterpret_cast<const float4*>(wp + ((size_t)(co0 + wrow) * taps + tap) * g.Cpad + c0 + wcol);
  };
  auto store_w = [&](int buf) {
    if (wrow < TCO) {
      float* d = w_lds + ((size_t)buf * TCO + wrow) * LDW + wcol;
      d[0] = wv.x; d[1] = wv.y; d[2] = wv.z; d[3] = wv.w;
    }
  };

  f32x16 acc[NP][NT];
#pragma unroll
  for (int i = 0; i < NP; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const float* src_n0 = src + (size_t)n0 * g.C * plane_s;
  __syncthreads();  // rtab
  int wbuf = 0;
  for (int c0 = 0; c0 < g.Cpad && taps > 0; c0 += kKC) {
    // ---- stage the patch chunk: wave w -> channels c0 + w + 4u, lanes -> patch positions
    for (int rb = 0; rb < pg.ps_raw; rb += 64) {
      const int r = rb + lane;
      const bool rv = r < pg.ps_raw;
      const int t = rv ? rtab[r] : -1;
      const bool ok = t >= 0;
      const int ip = t >> 26, off = t & ((1 << 26) - 1);
      const float* s_n = src_n0 + (size_t)ip * g.C * plane_s + off;
      float v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int c = c0 + wave + 4 * u;
        v[u] = 0.f;
        if (ok && c < ctot) v[u] = c < g.C ? s_n[(size_t)c * plane_s] : src2[(size_t)(c - g.C) * plane_s + off];
      }
      if (rv) {
#pragma unroll
        for (int u = 0; u < 4; ++u) patch[(wave + 4 * u) * pg.ps + r] = (g.pre_act && ok) ? act_fwd(v[u], g.act) : v[u];
      }
    }
    load_w(0, c0);
    store_w(wbuf);
    __syncthreads();
    for (int tap = 0; tap < taps; ++tap) {
      if (tap + 1 < taps) load_w(tap + 1, c0);
      const int ty = tap / g.KW, tx = tap - ty * g.KW;
      const int tapoff = g.TS > 0 ? ty * pg.pw + tx : (g.KH - 1 - ty) * pg.pw + (g.KW - 1 - tx);
      const float* wb = w_lds + ((size_t)wbuf * TCO + il) * LDW + kl;
      // all operands of the tap first (LDS reads in flight), then the MFMAs: one wait per tap, not per MFMA pair
      float bb[NP][kKC / 2], aa[NT][kKC / 2];
#pragma unroll
      for (int step = 0; step < kKC / 2; ++step) {
#pragma unroll
        for (int i = 0; i < NP; ++i) bb[i][step] = patch[pixoff[i] + tapoff + 2 * step * pg.ps];
#pragma unroll
        for (int j = 0; j < NT; ++j) aa[j][step] = wb[j * 32 * LDW + 2 * step];
      }
#pragma unroll
      for (int step = 0; step < kKC / 2; ++step)
#pragma unroll
        for (int i = 0; i < NP; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(aa[j][step], bb[i][step], acc[i][j], 0, 0, 0);
      if (tap + 1 < taps) store_w(wbuf ^ 1);
      __syncthreads();
      wbuf ^= 1;
    }
  }

#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const long pe = p0 + (wave * NP + i) * 32 + il;
    if (pe < ptot) {
      const int n = (int)(pe / plane_q);
      const int rem = (int)(pe - (long)n * plane_q);
      const int oy = rem / g.Wq, ox = rem - oy * g.Wq;
      const size_t plane_o = (size_t)g.Ho * g.Wo;
      const size_t base = (size_t)n * g.Cout * plane_o + (size_t)(oy * g.OS + g.QY) * g.Wo + (ox * g.OS + g.QX);
#pragma unroll
      for (int j = 0; j < NT; ++j) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int co = co0 + j * 32 + (r & 3) + 8 * (r >> 2) + 4 * kl;
          if (co < g.Cout) {
            float v = acc[i][j][r] + (bias ? bias[co] : 0.f);
            const size_t o = base + (size_t)co * plane_o;
            if (actgrad_in) v *= act_grad_from_in(actgrad_in[o], g.act);
            if (add_in) v += add_in[o];
            out[o] = v;
          }
        }
      }
    }
  }
}

}  // namespace mtrssm
#include "conv_split.h"
#include "conv_resident.h"
#include "conv_s2_band.h"
#include "conv_wgrad_resident.h"

namespace mtrssm {

// ---- partial-set reduction of the staged weight-gradient kernels, immediate or batched -----------------------------
// Every staged weight-gradient kernel leaves S partial tile sets in the caller's workspace and a small kernel sums them into
// dwp (38 such launches of 6-15 us per train step, launch-bound).  With deferral on (mtrssm_conv_weight_grad_deferred) the
// sums are recorded instead and mtrssm_conv_weight_grad_reduce() runs them all in ONE launch: the jobs travel by value in the
// kernel arguments (a captured graph replays them), each workgroup finds its job by its block range.
static int launched(const char* who);
enum ReduceVariant { kRedRes64, kRedRes32, kRed1x1_64, kRed1x1_128, kRedThin, kRedThinT, kRedS2, kRedT4, kRedT4b, kRedS2c };
struct ReduceJob {
  const float4* part;
  float* dwp;
  float* dbias;
  int S, cpad, variant, gx, gy;   // gx x gy blocks (block lb of the job = (lb % gx, lb / gx))
  __host__ int variant_gy() const { return gy; }
};
constexpr int kReduceBatchMax = 48;
struct ReduceBatch {
  int count;
  int first[kReduceBatchMax + 1];   // first block of job j; first[count] = the grid
  ReduceJob job[kReduceBatchMax];
};
__device__ __forceinline__ void wgrad_reduce_dispatch(const ReduceJob& j, int bx, int by) {
  switch (j.variant) {  // block-uniform
    case kRedRes64: wgrad_reduce_partials_body<64>(j.part, j.S, j.cpad, j.dwp, j.dbias, bx, by); break;
    case kRedRes32: wgrad_reduce_partials_body<32>(j.part, j.S, j.cpad, j.dwp, j.dbias, bx, by); break;
    case kRed1x1_64: wgrad_reduce_partials1x1_body<64>(j.part, j.S, j.cpad, j.dwp, j.dbias, bx, by); break;
    case kRed1x1_128: wgrad_reduce_partials1x1_body<128>(j.part, j.S, j.cpad, j.dwp, j.dbias, bx, by); break;
    case kRedThin: wgrad_reduce_partials_thin_body(j.part, j.S, j.cpad, j.dwp, j.dbias, bx, by); break;
    case kRedThinT: wgrad_reduce_partials_thint_body(j.part, j.S, j.cpad, j.dwp, bx, by); break;
    case kRedS2: wgrad_reduce_partials_s2_body(j.part, j.S, j.cpad, j.dwp, j.dbias, bx, by); break;
    case kRedT4: wgrad_reduce_partials_t4_body(j.part, j.S, j.cpad, j.dwp, bx, by); break;
    case kRedT4b: wgrad_reduce_partials_t4b_body(j.part, j.S, j.cpad, j.dwp, bx, by); break;
    default: wgrad_reduce_partials_s2c_body(j.part, j.S, j.cpad, j.dwp, j.dbias, bx, by); break;
  }
}
__global__ __launch_bounds__(256) void wgrad_reduce_batch_kernel(const ReduceBatch b) {
  int j = 0;
  while (j + 1 < b.count && (int)blockIdx.x >= b.first[j + 1]) ++j;  // scalar: <= 48 steps
  const int lb = (int)blockIdx.x - b.first[j], gx = b.job[j].gx;
  wgrad_reduce_dispatch(b.job[j], lb % gx, lb / gx);
}

// jobs recorded per stream.  Two jobs of one batch never add into the same dwp / dbias words unless the SAME layer ran twice
// between two flushes; reduce_record flushes first in that case (the sums are plain read-modify-writes).
static thread_local bool tl_defer_reduce = false;
static std::mutex g_reduce_mutex;
// keyed by (device, stream): the null stream of two devices is the same handle
static std::map<std::pair<int, hipStream_t>, std::vector<ReduceJob>> g_reduce_jobs;
static std::pair<int, hipStream_t> reduce_key(hipStream_t stream) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  return {dev, stream};
}

static int reduce_flush_locked(std::vector<ReduceJob>& jobs, hipStream_t stream) {
  size_t at = 0;
  while (at < jobs.size()) {
    ReduceBatch b{};
    int blocks = 0;
    while (at < jobs.size() && b.count < kReduceBatchMax) {
      b.first[b.count] = blocks;
      b.job[b.count] = jobs[at];
      blocks += jobs[at].gx * jobs[at].variant_gy();
      ++b.count;
      ++at;
    }
    b.first[b.count] = blocks;
    set_last_kernel("mtrssm::wgrad_reduce_batch_kernel");
    hipLaunchKernelGGL(wgrad_reduce_batch_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, b);
  }
  jobs.clear();
  return launched("conv_weight_grad_reduce");
}

int conv_weight_grad_reduce_flush(hipStream_t stream) {
  std::lock_guard<std::mutex> lock(g_reduce_mutex);
  auto it = g_reduce_jobs.find(reduce_key(stream));
  if (it == g_reduce_jobs.end() || it->second.empty()) return MTRSSM_OK;
  return reduce_flush_locked(it->second, stream);
}

// the reduce launch of a weight-gradient kernel: now, or recorded for the batch
static int reduce_now_or_later(int variant, dim3 rgrid, const float* part, int S, int cpad, float* dwp, float* dbias, hipStream_t stream) {
  ReduceJob j{reinterpret_cast<const float4*>(part), dwp, dbias, S, cpad, variant, (int)rgrid.x, (int)rgrid.y};
  if (tl_defer_reduce) {
    std::lock_guard<std::mutex> lock(g_reduce_mutex);
    auto& jobs = g_reduce_jobs[reduce_key(stream)];
    for (const ReduceJob& o : jobs)
      if (o.dwp == dwp || o.part == j.part) {  // the same target (or workspace) again: keep the order
        if (int rc = reduce_flush_locked(jobs, stream)) return rc;
        break;
      }
    jobs.push_back(j);
    return MTRSSM_OK;
  }
  ReduceBatch b{};
  b.count = 1;
  b.first[0] = 0;
  b.first[1] = (int)(rgrid.x * rgrid.y);
  b.job[0] = j;
  hipLaunchKernelGGL(wgrad_reduce_batch_kernel, dim3(rgrid.x * rgrid.y), dim3(256), 0, stream, b);
  return MTRSSM_OK;
}

}  // namespace mtrssm
namespace mtrssm {

// ------------------------------------------------------------------------------------------------
// Thin layers (few channels on big planes: first encoder conv 3->8, last decoder deconv 16->1 ...):
// a 32x32 MFMA tile would be >90 % padding and the layers are HBM-bound anyway, so they run on the VALU.
// gather: one thread per output pixel, COT output channels in registers, weights broadcast from LDS.
// ------------------------------------------------------------------------------------------------
constexpr int kThinMaxK = 256;  // taps * channels

template <int COT>
__global__ __launch_bounds__(kConvThreads) void conv_gather_thin_kernel(
    const MtrssmConvGeom g, const float* __restrict__ src, const float* __restrict__ src2, const float* __restrict__ wp,
    const float* __restrict__ bias, const float* __restrict__ actgrad_in, const float* __restrict__ add_in, float* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) float w_lds[kThinMaxK * COT];
  const int tid = threadIdx.x;
  const int taps = g.KH * g.KW, ctot = g.C + g.C2;
  const int K = taps * ctot;
  for (int e = tid; e < K * COT; e += kConvThreads) {
    const int k = e / COT, co = e - k * COT;
    const int tap = k / ctot, c = k - tap * ctot;
    w_lds[e] = co < g.Cout ? wp[((size_t)co * taps + tap) * g.Cpad + c] : 0.f;
  }
  __syncthreads();
  const long ptot = (long)g.N * g.Hq * g.Wq;
  const long p = (long)blockIdx.x * kConvThreads + tid;
  if (p >= ptot) return;
  const int n = (int)(p / (g.Hq * g.Wq));
  const int rem = (int)(p - (long)n * g.Hq * g.Wq);
  const int oy = rem / g.Wq, ox = rem - oy * g.Wq;
  const int plane_s = g.Hs * g.Ws;
  const float* src_n = src + (size_t)n * g.C * plane_s;
  float acc[COT];
#pragma unroll
  for (int j = 0; j < COT; ++j) acc[j] = (bias && j < g.Cout) ? bias[j] : 0.f;
  for (int ty = 0; ty < g.KH; ++ty) {
    const int sy = oy * g.SS + ty * g.TS + g.OFFY;
    if (sy < 0 || sy >= g.Hs) continue;
    for (int tx = 0; tx < g.KW; ++tx) {
      const int sx = ox * g.SS + tx * g.TS + g.OFFX;
      if (sx < 0 || sx >= g.Ws) continue;
      const int off = sy * g.Ws + sx;
      const float* wk = w_lds + (size_t)(ty * g.KW + tx) * ctot * COT;
      for (int c = 0; c < ctot; ++c) {
        float v = c < g.C ? src_n[(size_t)c * plane_s + off] : src2[(size_t)(c - g.C) * plane_s + off];
        if (g.pre_act) v = g.act == MTRSSM_ACT_ELU ? elu_fast(v) : act_fwd(v, g.act);
#pragma unroll
        for (int j = 0; j < COT; ++j) acc[j] = fmaf(v, wk[c * COT + j], acc[j]);
      }
    }
  }
  const size_t plane_o = (size_t)g.Ho * g.Wo;
  const size_t base = (size_t)n * g.Cout * plane_o + (size_t)(oy * g.OS + g.QY) * g.Wo + (ox * g.OS + g.QX);
  // epilogue: all loads first (clamped channel, no branch around a load), then arithmetic, then stores -- "load, wait,
  // use" per channel serialised COT dependent round trips per thread
  float gv[COT], av[COT];
#pragma unroll
  for (int j = 0; j < COT; ++j) {
    const size_t o = base + (size_t)(j < g.Cout ? j : g.Cout - 1) * plane_o;
    gv[j] = actgrad_in ? actgrad_in[o] : 0.f;
    av[j] = add_in ? add_in[o] : 0.f;
  }
  if (actgrad_in) {
    if (g.act == MTRSSM_ACT_ELU) {
#pragma unroll
      for (int j = 0; j < COT; ++j) gv[j] = gv[j] > 0.f ? 1.f : __expf(gv[j]);
    } else {
#pragma unroll
      for (int j = 0; j < COT; ++j) gv[j] = act_grad_from_in(gv[j], g.act);
    }
  }
#pragma unroll
  for (int j = 0; j < COT; ++j) {
    float v = acc[j];
    if (actgrad_in) v *= gv[j];
    if (add_in) v += av[j];
    asm volatile("" : "+v"(v));
    acc[j] = v;
  }
#pragma unroll
  for (int j = 0; j < COT; ++j)
    if (j < g.Cout) out[base + (size_t)j * plane_o] = acc[j];
}

// ------------------------------------------------------------------------------------------------
// Conv2d(k = 4, s = 2, p = 1) gather from ONE input channel to 16 output channels: the backward-data of the decoders' last
// ConvTranspose layer (default.yaml:70-74; out *= act'(layer input)).  conv_gather_thin_kernel<16> spends 48 memory
// instructions and 64 LDS weight reads on a pixel's 256 FMAs (3.6 TB/s).  Here a thread owns TWO horizontally adjacent output
// pixels: their 4 x 6 source window is 16 loads (aligned pairs + the two edge columns), a tap's 16 weights are read once for
// both pixels, the act' operand and the output go as 8-byte pairs: 24 memory instructions and 32 LDS reads per pixel.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void conv4s2_c1_thin16_kernel(const MtrssmConvGeom g, const float* __restrict__ src, const float* __restrict__ wp,
                                                                const float* __restrict__ bias, const float* __restrict__ actgrad_in,
                                                                float* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) float w_lds[16 * 16];  // [tap][output channel]
  const int tid = threadIdx.x;
  w_lds[tid] = wp[((size_t)(tid & 15) * 16 + (tid >> 4)) * g.Cpad];  // host: 256 threads; wp [CoutPad][16 taps][Cpad], channel 0
  __syncthreads();
  const int Wq = g.Wq, Hq = g.Hq, Ws = g.Ws, Hs = g.Hs;  // host: Ws == 2 Wq, Hs == 2 Hq, Wq even
  const int hw = Wq >> 1, ppf = Hq * hw;
  const long pair = (long)blockIdx.x * 256 + tid;
  const int n = (int)(pair / ppf);
  if (n >= g.N) return;
  const int rem = (int)(pair - (long)n * ppf), oy = rem / hw, ox = 2 * (rem - oy * hw);
  const float* __restrict__ s = src + (size_t)n * Hs * Ws;
  // source rows 2 oy - 1 .. 2 oy + 2, columns 2 ox - 1 .. 2 ox + 4: unconditional loads from clamped addresses, selected afterwards
  float v[4][6];
  const bool lv = ox > 0, rv = 2 * ox + 4 < Ws;
#pragma unroll
  for (int ky = 0; ky < 4; ++ky) {
    const int sy = 2 * oy - 1 + ky;
    const bool ok = sy >= 0 && sy < Hs;
    const float* row = s + (size_t)(ok ? sy : 0) * Ws + 2 * ox;
    const float l = row[lv ? -1 : 0], r = row[rv ? 4 : 3];
    const float2 m0 = *reinterpret_cast<const float2*>(row), m1 = *reinterpret_cast<const float2*>(row + 2);
    v[ky][0] = ok && lv ? l : 0.f;
    v[ky][1] = ok ? m0.x : 0.f;
    v[ky][2] = ok ? m0.y : 0.f;
    v[ky][3] = ok ? m1.x : 0.f;
    v[ky][4] = ok ? m1.y : 0.f;
    v[ky][5] = ok && rv ? r : 0.f;
  }
  const size_t plane_o = (size_t)Hq * Wq;
  const size_t base = (size_t)n * 16 * plane_o + (size_t)oy * Wq + ox;
  float2 gv[16];
  {
    const float* gsafe = actgrad_in ? actgrad_in : out;  // any readable address of the same shape: the value is dropped
#pragma unroll
    for (int j = 0; j < 16; ++j) gv[j] = *reinterpret_cast<const float2*>(gsafe + base + (size_t)j * plane_o);
  }
  float a0[16], a1[16];
  {
    const float* bsafe = bias ? bias : w_lds;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const float b = bsafe[j];
      a0[j] = a1[j] = bias ? b : 0.f;
    }
  }
#pragma unroll
  for (int ky = 0; ky < 4; ++ky)
#pragma unroll
    for (int kx = 0; kx < 4; ++kx) {
      asm volatile("" ::: "memory");  // keeps a tap's four weight reads at the tap (hoisted to the top they are 256 registers)
      const float4* w4 = reinterpret_cast<const float4*>(w_lds + (ky * 4 + kx) * 16);
      const float p0 = v[ky][kx], p1 = v[ky][kx + 2];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 w = w4[q];
        a0[4 * q] = fmaf(p0, w.x, a0[4 * q]);
        a0[4 * q + 1] = fmaf(p0, w.y, a0[4 * q + 1]);
        a0[4 * q + 2] = fmaf(p0, w.z, a0[4 * q + 2]);
        a0[4 * q + 3] = fmaf(p0, w.w, a0[4 * q + 3]);
        a1[4 * q] = fmaf(p1, w.x, a1[4 * q]);
        a1[4 * q + 1] = fmaf(p1, w.y, a1[4 * q + 1]);
        a1[4 * q + 2] = fmaf(p1, w.z, a1[4 * q + 2]);
        a1[4 * q + 3] = fmaf(p1, w.w, a1[4 * q + 3]);
      }
    }
  // act' as lane-uniform selects (host: no Tanh): a branch per element splits the epilogue into hundreds of blocks
  const bool elu = g.act == MTRSSM_ACT_ELU, relu = g.act == MTRSSM_ACT_RELU, has_g = actgrad_in != nullptr;
  auto egrad = [&](float x) {
    float e = __expf(x);
    asm volatile("" : "+v"(e));
    const float neg = elu ? e : (relu ? 0.f : 1.f);
    const float m = x > 0.f ? 1.f : neg;
    return has_g ? m : 1.f;
  };
#pragma unroll
  for (int j = 0; j < 16; ++j)
    *reinterpret_cast<float2*>(out + base + (size_t)j * plane_o) = make_float2(a0[j] * egrad(gv[j].x), a1[j] * egrad(gv[j].y));
}

// ------------------------------------------------------------------------------------------------
// Last decoder layer: ConvTranspose2d(k = 4, s = 2, p = 1) to <= 2 output channels with the preceding activation
// fused.  The generic thin kernel runs one launch per output parity class and re-activates every source value once
// per tap (expm1f dominates its instruction stream).  Here a workgroup stages act(x) for an 8 x 32 tile of INPUT
// positions (+1 halo) in LDS once -- one activation per element -- and each thread produces the 2 x 2 output block
// of its input position from the 3 x 3 neighbourhood: 9 LDS reads + 16 fmas per channel, weights through the
// scalar cache (wave-uniform addresses).
// ------------------------------------------------------------------------------------------------
// (tile of TW x 256 / TW input positions: 32 x 8, or 16 x 16 for planes only 16 wide -- the audio decoder's 64 x 16 -- where half of
// a 32-wide tile's threads had no pixel: 176 -> ~100 us)
constexpr int kCtTW = 32, kCtTH = 8, kCtPS = (kCtTH + 2) * (kCtTW + 2) + 1;  // the larger of the two tiles' LDS planes

// Transposed gather with <= 8 output channels (backward-data of the encoders' second conv: dX[N, 8, 32, 32] from
// dY[N, 16, 16, 16], k = 3, s = 2, p = 1), ALL output parity classes in one pass: one thread per output pixel visits the taps
// of its own parity.  The general path launches one sub-grid per parity class (4 launches, each re-reading the whole source
// and storing every second pixel): 4 x 69 us per modality against one ~40 us pass here (HBM: source once, output once, fully
// coalesced).
//   out[n, c, iy, ix] = (bias[c] + sum_{o, ky, kx} w[o][c][ky][kx] pre(y)[n, o, (iy + p - ky) / s, (ix + p - kx) / s]) * act'(actgrad_in) + add_in
template <int COT, int S>  // S: the stride as a constant
__global__ __launch_bounds__(kConvThreads) void conv_tgather_thin_kernel(
    const int N, const int O, const int Hs, const int Ws, const int Cc, const int KH, const int KW, const int P, const int Ho,
    const int Wo, const float* __restrict__ y, const float* __restrict__ w, const float* __restrict__ bias, const int pre_act, const int act,
    const float* __restrict__ actgrad_in, const float* __restrict__ add_in, float* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) float w_lds[kThinMaxK * COT];
  const int tid = threadIdx.x;
  const int taps = KH * KW;
  for (int e = tid; e < taps * O * COT; e += kConvThreads) {
    const int j = e % COT, k = e / COT;
    const int oc = k % O, t = k / O;
    w_lds[e] = j < Cc ? w[((size_t)oc * Cc + j) * taps + t] : 0.f;
  }
  __syncthreads();
  // Pixels are enumerated parity class by parity class (host: Ho % S == 0, Wo % S == 0): the lanes of a wave share their taps
  // (uniform loops, weights by broadcast LDS reads).  A first version with one thread per pixel in plane order ran every wave
  // through all KH * KW taps at 25 % of its lanes: 291 us; the four per-class launches it replaces: 4 x 69 us.
  const int Hq = Ho / S, Wq = Wo / S, PQ = Hq * Wq;
  const long ptot = (long)N * S * S * PQ;
  const long p = (long)blockIdx.x * kConvThreads + tid;
  if (p >= ptot) return;
  const int cls = (int)(p / PQ), r = (int)(p - (long)cls * PQ);
  const int n = cls / (S * S), q = cls - n * (S * S);
  const int qy = q / S, qx = q - qy * S;
  const int jy = r / Wq, jx = r - jy * Wq;
  const int iy = jy * S + qy, ix = jx * S + qx;
  const int plane_s = Hs * Ws;
  const float* yn = y + (size_t)n * O * plane_s;
  float acc[COT];
#pragma unroll
  for (int j = 0; j < COT; ++j) acc[j] = (bias && j < Cc) ? bias[j] : 0.f;
  for (int ky = (qy + P) % S; ky < KH; ky += S) {
    const int sy = jy + (qy + P - ky) / S;  // exact: (qy + P - ky) is a multiple of S
    if (sy < 0 || sy >= Hs) continue;
    for (int kx = (qx + P) % S; kx < KW; kx += S) {
      const int sx = jx + (qx + P - kx) / S;
      if (sx < 0 || sx >= Ws) continue;
      const float* wk = w_lds + (size_t)(ky * KW + kx) * O * COT;
      const float* yp = yn + sy * Ws + sx;
      for (int oc0 = 0; oc0 < O; oc0 += 16) {  // sixteen channels' loads in flight before the first is used
        float v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = yp[(size_t)(oc0 + u < O ? oc0 + u : O - 1) * plane_s];
        if (pre_act) {
#pragma unroll
          for (int u = 0; u < 16; ++u) v[u] = act == MTRSSM_ACT_ELU ? elu_fast(v[u]) : act_fwd(v[u], act);
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          if (oc0 + u < O) {
#pragma unroll
            for (int j = 0; j < COT; ++j) acc[j] = fmaf(v[u], wk[(oc0 + u) * COT + j], acc[j]);
          }
        }
      }
    }
  }
  const size_t plane_o = (size_t)Ho * Wo;
  const size_t base = (size_t)n * Cc * plane_o + (size_t)iy * Wo + ix;
  float gv[COT], av[COT];
#pragma unroll
  for (int j = 0; j < COT; ++j) {
    const size_t o = base + (size_t)(j < Cc ? j : Cc - 1) * plane_o;
    gv[j] = actgrad_in ? actgrad_in[o] : 0.f;
    av[j] = add_in ? add_in[o] : 0.f;
  }
  if (actgrad_in) {
    if (act == MTRSSM_ACT_ELU) {
#pragma unroll
      for (int j = 0; j < COT; ++j) gv[j] = gv[j] > 0.f ? 1.f : __expf(gv[j]);
    } else {
#pragma unroll
      for (int j = 0; j < COT; ++j) gv[j] = act_grad_from_in(gv[j], act);
    }
  }
#pragma unroll
  for (int j = 0; j < COT; ++j) {
    float v = acc[j];
    if (actgrad_in) v *= gv[j];
    if (add_in) v += av[j];
    if (j < Cc) out[base + (size_t)j * plane_o] = v;
  }
}
template <int COT, int TW>
__global__ __launch_bounds__(kConvThreads) void convt_k4s2_thin_kernel(
    const int N, const int C, const int Hs, const int Ws, const int Cout, const float* __restrict__ src,
    const float* __restrict__ w, const float* __restrict__ bias, const int pre_act, const int act, float* __restrict__ out) {
  constexpr int kCtTW = TW, kCtTH = kConvThreads / TW, kCtPW = kCtTW + 2, kCtPS = (kCtTH + 2) * kCtPW + 1;  // (shadow the defaults)
  extern __shared__ __attribute__((aligned(16))) float lds[];  // [C][kCtPS]
  const int tid = threadIdx.x;
  const int x0 = blockIdx.x * kCtTW, y0 = blockIdx.y * kCtTH, n = blockIdx.z;
  const int plane = Hs * Ws;
  const float* src_n = src + (size_t)n * C * plane;
  for (int e = tid; e < (kCtTH + 2) * kCtPW; e += kConvThreads) {
    const int py = e / kCtPW, px = e - py * kCtPW;
    const int sy = y0 + py - 1, sx = x0 + px - 1;
    const bool ok = sy >= 0 && sy < Hs && sx >= 0 && sx < Ws;
    const int off = sy * Ws + sx;
    for (int c0 = 0; c0 < C; c0 += 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = (ok && c0 + u < C) ? src_n[(size_t)(c0 + u) * plane + off] : 0.f;
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (c0 + u < C) lds[(c0 + u) * kCtPS + e] = (pre_act && ok) ? act_fwd(v[u], act) : v[u];
    }
  }
  __syncthreads();
  const int ty = tid / kCtTW, tx = tid - ty * kCtTW;
  const int iy = y0 + ty, ix = x0 + tx;
  if (iy >= Hs || ix >= Ws) return;
  float o00[COT], o01[COT], o10[COT], o11[COT];
#pragma unroll
  for (int j = 0; j < COT; ++j) o00[j] = o01[j] = o10[j] = o11[j] = (bias && j < Cout) ? bias[j] : 0.f;
  const float* t = lds + (ty + 1) * kCtPW + (tx + 1);
  for (int c = 0; c < C; ++c, t += kCtPS) {
    const float v00 = t[-kCtPW - 1], v01 = t[-kCtPW], v02 = t[-kCtPW + 1];
    const float v10 = t[-1], v11 = t[0], v12 = t[1];
    const float v20 = t[kCtPW - 1], v21 = t[kCtPW], v22 = t[kCtPW + 1];
#pragma unroll
    for (int j = 0; j < COT; ++j) {
      if (j < Cout) {
        const float* q = w + ((size_t)c * Cout + j) * 16;  // [ky][kx], wave-uniform
        // y[2i+0] = x[i] w[ky=1] + x[i-1] w[ky=3] ; y[2i+1] = x[i+1] w[ky=0] + x[i] w[ky=2]   (same along x)
        o00[j] += v11 * q[5] + v10 * q[7] + v01 * q[13] + v00 * q[15];
        o01[j] += v12 * q[4] + v11 * q[6] + v02 * q[12] + v01 * q[14];
        o10[j] += v21 * q[1] + v20 * q[3] + v11 * q[9] + v10 * q[11];
        o11[j] += v22 * q[0] + v21 * q[2] + v12 * q[8] + v11 * q[10];
      }
    }
  }
  const int Wo = 2 * Ws;
  const size_t plane_o = (size_t)4 * plane;
#pragma unroll
  for (int j = 0; j < COT; ++j) {
    if (j < Cout) {
      float* o = out + ((size_t)n * Cout + j) * plane_o + (size_t)(2 * iy) * Wo + 2 * ix;
      *reinterpret_cast<float2*>(o) = make_float2(o00[j], o01[j]);
      *reinterpret_cast<float2*>(o + Wo) = make_float2(o10[j], o11[j]);
    }
  }
}

// weight gradient of a thin layer: same group / patch staging as the patch kernel, but each thread owns up to
// kThinOut output elements (co, tap, c) and reduces over the group's 64 pixels with VALU fmas.
constexpr int kThinOut = 5;

__global__ __launch_bounds__(kConvThreads) void conv_weight_grad_thin_kernel(
    const MtrssmConvGeom g, const float* __restrict__ a, const float* __restrict__ src, const float* __restrict__ src2,
    const int pre_act_a, float* __restrict__ dwp, float* __restrict__ dbias) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const PatchGeom pg(g, kGP);
  const int ctot = g.C + g.C2;
  // two staging buffers [Cout][kLDA] + [ctot][ps]: the LDS-DMA of group g+1 flies while group g is reduced
  const int buf_floats = g.Cout * kLDA + ctot * pg.ps;
  int* pixtab = reinterpret_cast<int*>(lds + 2 * (size_t)buf_floats);  // [kGP]
  int* ptab = pixtab + kGP;  // [ps_raw rounded up to 64]: group-invariant decode (frame << 20 | patch row << 10 | patch column)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int taps = g.KH * g.KW;
  const int n_out = g.Cout * taps * ctot;
  const int plane_s = g.Hs * g.Ws, plane_a = g.Hq * g.Wq;
  const long ptot = (long)g.N * plane_a;
  const long groups = (ptot + kGP - 1) / kGP;
  const long gper = (groups + gridDim.x - 1) / gridDim.x;
  const long gbeg = (long)blockIdx.x * gper;
  const long gend = gbeg + gper < groups ? gbeg + gper : groups;
  if (tid < kGP) {
    const int row = tid / g.Wq, ox = tid - row * g.Wq;
    const int ip = row / pg.rp, lr = row - ip * pg.rp;
    pixtab[tid] = ip * pg.ph * pg.pw + lr * g.SS * pg.pw + ox * g.SS;
  }
  {
    const int phw = pg.ph * pg.pw;
    for (int r = tid; r < ((pg.ps_raw + 63) & ~63); r += kConvThreads) {
      const int ip = r / phw, q = r - ip * phw;
      const int pr = q / pg.pw, pcn = q - pr * pg.pw;
      ptab[r] = r < pg.ps_raw ? (ip << 20) | (pr << 10) | pcn : -1;
    }
  }
  __syncthreads();
  const int iptot = (int)ptot;  // host checks N * Hq * Wq < 2^31
  int arow[kThinOut], pbase[kThinOut];
  float acc[kThinOut];
#pragma unroll
  for (int s = 0; s < kThinOut; ++s) {
    const int o = tid + s * kConvThreads;  // o = (co * taps + tap) * ctot + c
    acc[s] = 0.f;
    arow[s] = -1;
    pbase[s] = 0;
    if (o < n_out) {
      const int co = o / (taps * ctot), r = o - co * taps * ctot;
      const int tap = r / ctot, c = r - tap * ctot;
      const int ty = tap / g.KW, tx = tap - ty * g.KW;
      arow[s] = co * kLDA;
      pbase[s] = c * pg.ps + ty * pg.pw + tx;
    }
  }
  float bsum = 0.f;  // thread t < Cout: bias gradient of channel t
  auto dma = [&](const float* gsrc, float* ldst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)ldst, 4, 0, 0);
  };
  auto stage = [&](long grp, float* a_lds, float* patch) {
    const int p0 = (int)grp * kGP;
    {
      const int p = p0 + lane;
      const bool pv = p < iptot;
      int n = 0, rem = 0;
      if (pv) { n = p / plane_a; rem = p - n * plane_a; }
      const float* a_n = a + (size_t)n * g.Cout * plane_a + rem;
      for (int row = wave; row < g.Cout; row += 4) {
        float* dst = a_lds + row * kLDA;
        if (pv) dma(a_n + (size_t)row * plane_a, dst);
        else dst[lane] = 0.f;
      }
    }
    {
      const int n0 = p0 / plane_a;
      const int r0 = (p0 - n0 * plane_a) / g.Wq;
      const int sy0 = r0 * g.SS + g.OFFY, sx0 = g.OFFX;
      for (int rb = 0; rb < pg.ps_raw; rb += 64) {
        const int d = ptab[rb + lane];
        const bool rv = d >= 0;
        const int n = n0 + (d >> 20), sy = sy0 + ((d >> 10) & 1023), sx = sx0 + (d & 1023);
        const bool ok = rv && n < g.N && sy >= 0 && sy < g.Hs && sx >= 0 && sx < g.Ws;
        const size_t off = (size_t)sy * g.Ws + sx;
        const float* s_n = src + (size_t)n * g.C * plane_s + off;
        for (int c = wave; c < ctot; c += 4) {
          float* dst = patch + c * pg.ps + rb;
          if (ok) dma(c < g.C ? s_n + (size_t)c * plane_s : src2 + (size_t)(c - g.C) * plane_s + off, dst);
          else if (rv) dst[lane] = 0.f;
        }
      }
    }
  };
  // LDS-DMA staging (every row load of a group in flight at once), double-buffered: one barrier per group.
  if (gbeg < gend) stage(gbeg, lds, lds + (size_t)g.Cout * kLDA);
  int cur = 0;
  for (long grp = gbeg; grp < gend; ++grp) {
    float* a_lds = lds + (size_t)cur * buf_floats;
    float* patch = a_lds + (size_t)g.Cout * kLDA;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's share of group grp has landed
    if (pre_act_a) {  // in place, each wave its own rows
      for (int row = wave; row < g.Cout; row += 4) {
        float* d = a_lds + row * kLDA + lane;
        *d = act_fwd(*d, g.act);
      }
    }
    if (g.pre_act) {
      for (int rb = 0; rb < pg.ps_raw; rb += 64) {
        if (rb + lane < pg.ps_raw) {
          for (int c = wave; c < ctot; c += 4) {
            float* d = patch + c * pg.ps + rb + lane;
            *d = act_fwd(*d, g.act);
          }
        }
      }
    }
    __syncthreads();  // group grp complete in `cur`; every wave is done reducing group grp-1 from the other buffer
    if (grp + 1 < gend) stage(grp + 1, lds + (size_t)(cur ^ 1) * buf_floats, lds + (size_t)(cur ^ 1) * buf_floats + (size_t)g.Cout * kLDA);
    if (dbias && tid < g.Cout) {
      const float* ar = a_lds + tid * kLDA;
      float t = 0.f;
#pragma unroll 8
      for (int pix = 0; pix < kGP; ++pix) t += ar[pix];
      bsum += t;
    }
#pragma unroll
    for (int s = 0; s < kThinOut; ++s) {
      if (arow[s] >= 0) {
        const float* ar = a_lds + arow[s];
        const float* pb = patch + pbase[s];
        float t = 0.f;
#pragma unroll 8
        for (int pix = 0; pix < kGP; ++pix) t = fmaf(ar[pix], pb[pixtab[pix]], t);
        acc[s] += t;
      }
    }
    cur ^= 1;
  }
#pragma unroll
  for (int s = 0; s < kThinOut; ++s) {
    const int o = tid + s * kConvThreads;
    if (o < n_out) {
      const int co = o / (taps * ctot), r = o - co * taps * ctot;
      const int tap = r / ctot, c = r - tap * ctot;
      atomicAdd(&dwp[((size_t)co * taps + tap) * g.Cpad + c], acc[s]);
    }
  }
  if (dbias && tid < g.Cout) atomicAdd(&dbias[tid], bsum);
}

// per-channel sum over (N, H*W): out[c] += sum x[n, c, :]
__global__ void channel_sum_kernel(const float* __restrict__ x, int N, int C, int HW, float* __restrict__ out) {
  // block (c, split): frames [nbeg, nend) of channel c; 16-byte loads when the plane is a multiple of 4 floats
  __shared__ float red[4];
  const int c = blockIdx.x;
  const int fper = (N + gridDim.y - 1) / gridDim.y;
  const int nbeg = blockIdx.y * fper, nend = nbeg + fper < N ? nbeg + fper : N;
  float acc = 0.f;
  if ((HW & 3) == 0 && !((uintptr_t)x & 15)) {
    const int hw4 = HW >> 2;
    const int quads = (nend - nbeg) * hw4;  // < 2^31: host checks N * HW
    auto quad = [&](int q) {
      const int n = nbeg + q / hw4, r4 = q % hw4;
      return reinterpret_cast<const float4*>(x + ((size_t)n * C + c) * HW)[r4];
    };
    int q = threadIdx.x;
    const int bd = blockDim.x;
    for (; q + 3 * bd < quads; q += 4 * bd) {  // four quads in flight per thread
      const float4 a = quad(q), b = quad(q + bd), d = quad(q + 2 * bd), e = quad(q + 3 * bd);
      acc += (((a.x + a.y) + (a.z + a.w)) + ((b.x + b.y) + (b.z + b.w))) + (((d.x + d.y) + (d.z + d.w)) + ((e.x + e.y) + (e.z + e.w)));
    }
    for (; q < quads; q += bd) {
      const float4 v = quad(q);
      acc += (v.x + v.y) + (v.z + v.w);
    }
  } else {
    const int elems = (nend - nbeg) * HW;
    for (int i = threadIdx.x; i < elems; i += blockDim.x) {
      const int n = nbeg + i / HW, r = i % HW;
      acc += x[((size_t)n * C + c) * HW + r];
    }
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(&out[c], red[0] + red[1] + red[2] + red[3]);
}

// ------------------------------------------------------------------------------------------------
// host launchers
// ------------------------------------------------------------------------------------------------
static bool plane_fits_26bit(const MtrssmConvGeom* g) { return (long)g->Hs * g->Ws < (1L << 26); }

static int check_geom(const MtrssmConvGeom* g, const char* who) {
  if (!g || g->N <= 0 || g->C <= 0 || g->Hs <= 0 || g->Ws <= 0 || g->C2 < 0 || g->KH < 0 || g->KW < 0 || g->Hq <= 0 || g->Wq <= 0 ||
      g->Ho <= 0 || g->Wo <= 0 || g->Cout <= 0 || g->OS <= 0) {
    set_error("%s: bad geometry", who);
    return MTRSSM_EINVAL;
  }
  if (g->Cpad % kKC || g->Cpad < g->C + g->C2 || g->CoutPad % 32 || g->CoutPad < g->Cout) {
    set_error("%s: Cpad must be a multiple of 16 covering C+C2, CoutPad a multiple of 32 covering Cout", who);
    return MTRSSM_EINVAL;
  }
  if ((g->Hq - 1) * g->OS + g->QY >= g->Ho || (g->Wq - 1) * g->OS + g->QX >= g->Wo || g->QY < 0 || g->QX < 0) {
    set_error("%s: output sub-grid exceeds the output plane", who);
    return MTRSSM_EINVAL;
  }
  if (g->Cout > 32 && g->CoutPad % 64) {
    set_error("%s: CoutPad must be a multiple of 64 when Cout > 32", who);
    return MTRSSM_EINVAL;
  }
  if (g->act < MTRSSM_ACT_IDENTITY || g->act > MTRSSM_ACT_TANH) {
    set_error("%s: unknown activation id %d", who, g->act);
    return MTRSSM_EINVAL;
  }
  if (g->mfma_split < 0 || g->mfma_split > 3) {
    set_error("%s: mfma_split must be 0 (fp32 MFMA) or 1..3 bf16 pieces, got %d", who, g->mfma_split);
    return MTRSSM_EINVAL;
  }
  return MTRSSM_OK;
}

static int launched(const char* who) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s launch failed: %s", who, hipGetErrorString(e));
    return MTRSSM_ELAUNCH;
  }
  return MTRSSM_OK;
}

// ---- split-bf16 gather kernels: plan (which kernel, which tiling) and launch of one or two problems
struct SplitPlan {
  int kind = 0;  // 0: not covered, 1: conv1x1_split_kernel, 2: conv_gather_split_kernel, 3: conv3x3_resident_kernel
  int tco = 0, ny = 0, nx = 0, sp = 0, pit = 0, tgs = 0, ngroups = 0;
  int res = 0;  // kind 3: CIN * 1000 + Cout (the instantiation)
  int stream = 0;  // kind 1: CIN * 1000 + Cout when conv1x1_stream_kernel covers the shape (operands decide at launch)
  size_t lds = 0;
  bool same_kernel(const SplitPlan& o) const {
    if (kind == 3) return o.kind == 3 && res == o.res;
    return kind == o.kind && kind != 0 && tco == o.tco && ny == o.ny && sp == o.sp && pit == o.pit && tgs == o.tgs && ngroups == o.ngroups;
  }
};

// MTRSSM_CONV_RESIDENT=0: the patch-staged kernel for every layer (A/B runs of conv3x3_resident_kernel)
static bool resident_enabled() {
  static const bool on = [] { const char* e = getenv("MTRSSM_CONV_RESIDENT"); return !(e && e[0] == '0'); }();
  return on;
}

// MTRSSM_PATCH_768=0: 641..768-position patches (the k=4 s=2 backward-data gathers, 648 positions) back on the fp32 patch
// kernel.  Round 1 measured that kernel faster for them (283 us per modality); with audio + vision paired into one launch the
// split kernel takes 182 us for both (round 2), so it is the default now.
static bool patch_limit_768() {
  static const bool on = [] { const char* e = getenv("MTRSSM_PATCH_768"); return !(e && e[0] == '0'); }();
  return on;
}

// hipFuncSetAttribute is per device: the "done" flags of the launch macros are kept per device (a process that drives several
// GPUs -- not this package's one-process-per-GPU model, but nothing forbids it -- sets the attribute on each)
static int device_slot() {
  int dev = 0;
  (void)hipGetDevice(&dev);
  return dev & 63;
}

static int cu_count() {
  static const int n = [] {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
    return v;
  }();
  return n;
}

static SplitPlan plan_split(const MtrssmConvGeom* g, bool has_wq) {
  SplitPlan pl;
  if (!(g->mfma_split > 0 && has_wq) || !((g->TS == 1 || g->TS == -1) && g->KH * g->KW > 0 && plane_fits_26bit(g))) return pl;
  if (g->Wq > kTP || kTP % g->Wq) return pl;
  const PatchGeom pg(*g, kTP);
  const bool tiles = (g->Hq % pg.rpg == 0) || (pg.rpg % g->Hq == 0);
  if (!tiles || pg.ipg >= 32) return pl;
  const long ptot = (long)g->N * g->Hq * g->Wq;
  const int sp = g->mfma_split, taps = g->KH * g->KW;
  pl.tco = g->Cout > 32 ? 64 : 32;
  pl.ny = g->CoutPad / pl.tco;
  pl.nx = (int)((ptot + kTP - 1) / kTP);
  pl.sp = sp;
  if ((long)g->N * g->Cout * g->Ho * g->Wo >= (1L << 31)) return pl;
  // 3x3 / stride 1 / pad 1 layers of the residual stacks on 64-pixel planes (8x8, 16x4, 4x16): weights resident in registers (conv_resident.h)
  if (sp == 2 && resident_enabled() && g->KH == 3 && g->KW == 3 && g->SS == 1 && g->OS == 1 && g->QY == 0 && g->QX == 0 && g->C2 == 0 &&
      (g->Ws == 4 || g->Ws == 8 || g->Ws == 16) && g->Hs * g->Ws == 64 && g->Hq == g->Hs && g->Wq == g->Ws && g->Ho == g->Hs && g->Wo == g->Ws && g->Cpad == g->C && g->OFFY == -g->TS &&
      g->OFFX == -g->TS && (long)g->N * g->C * 64 < (1L << 31) && g->act != MTRSSM_ACT_TANH) {
    const int key = g->C * 1000 + g->Cout;
    const int fpt = key == 64064 || key == 32064 || key == 64032 ? 2 : 1;  // frames per tile: whole tiles only
    if ((key == 64064 || key == 64128 || key == 128064 || key == 32064 || key == 64032) && g->N % fpt == 0) {
      pl.kind = 3;
      pl.res = key;
      pl.nx = g->N;  // frames; launch_split turns them into workgroups
      pl.lds = key == 64032 ? res_lds_bytes<64, 1, 2>() : key == 64064 ? res_lds_bytes<64, 2, 1>() : key == 64128 ? res_lds_bytes<64, 4, 1>() : key == 128064 ? res_lds_bytes<128, 2, 2>()
                                                                                                     : res_lds_bytes<32, 2, 1>();
      return pl;
    }
  }
  if (taps == 1 && g->SS == 1 && g->OS == 1 && g->OFFY == 0 && g->OFFX == 0 && g->C2 == 0 && g->Hs == g->Hq && g->Ws == g->Wq &&
      g->Ho == g->Hq && g->Wo == g->Wq && g->Cpad % 64 == 0) {  // 1x1 layers: 64 channels per step
    pl.kind = 1;
    pl.lds = (size_t)sp * (kTP + pl.tco) * 128;
    const int key = g->C * 1000 + g->Cout;
    if (sp == 2 && resident_enabled() && (key == 64064 || key == 128064 || key == 64128) && g->Cpad == g->C && g->CoutPad == g->Cout &&
        (g->Hq * g->Wq) % 32 == 0 && g->act != MTRSSM_ACT_TANH && (long)g->N * g->C * g->Hq * g->Wq < (1L << 31))
      pl.stream = key;
    return pl;
  }
  auto lds_of = [&](int t) { return (size_t)sp * ((size_t)pg.ps_raw * kRowB + (size_t)t * pl.tco * kRowB); };
  int tgs = 1;  // largest divisor of taps within the register / LDS budget
  for (int t = 1; t <= split_tg(sp) && t <= taps; ++t)
    if (taps % t == 0 && lds_of(t) <= 80 * 1024) tgs = t;
  pl.tgs = tgs;
  pl.ngroups = taps / tgs;
  pl.lds = lds_of(tgs);
  pl.pit = pg.ps_raw <= 384 ? 3 : 6;
  if (pl.lds <= 80 * 1024 && pg.ps_raw <= (patch_limit_768() ? 768 : 640) && (long)pg.ipg * g->C * g->Hs * g->Ws < (1L << 31) &&
      (long)sp * g->CoutPad * taps * g->Cpad < (1L << 31))
    pl.kind = 2;
  return pl;
}

static GatherProblem make_problem(const MtrssmConvGeom* g, const SplitPlan& pl, const float* src, const float* src2, const unsigned short* wq,
                                  const float* bias, const float* actgrad_in, const float* add_in, float* out) {
  GatherProblem p;
  p.g = *g; p.src = src; p.src2 = src2; p.wq = wq; p.bias = bias; p.actgrad_in = actgrad_in; p.add_in = add_in; p.out = out;
  p.tg = pl.tgs; p.ngroups = pl.ngroups; p.nx = pl.nx;
  return p;
}

// launches pa (and pb when pb.nx > 0: same kernel, its workgroups appended to the grid)
static int launch_split(const SplitPlan& pl, size_t lds, const GatherProblem& pa, const GatherProblem& pb, hipStream_t stream) {
  if (pl.kind == 3) {
    // persistent workgroups, one per CU; a pair shares the CUs in proportion to its frames (each workgroup keeps ONE
    // problem's weights in registers)
    const bool epi_a = pa.actgrad_in || pa.add_in, epi_b = pb.actgrad_in || pb.add_in;  // template switch of the kernel
    if (pb.nx > 0 && epi_a != epi_b) {
      GatherProblem none{};
      none.nx = 0;
      if (int rc = launch_split(pl, lds, pa, none, stream)) return rc;
      return launch_split(pl, lds, pb, none, stream);
    }
    const int fpt = pl.res == 64064 || pl.res == 32064 || pl.res == 64032 ? 2 : 1;  // frames per tile
    const long ta = (pa.nx + fpt - 1) / fpt, tb = (pb.nx + fpt - 1) / fpt;
    const int ncu = cu_count();
    GatherProblem qa = pa, qb = pb;
    if (tb == 0) {
      qa.nx = (int)(ta < ncu ? ta : ncu);
      qb.nx = 0;
    } else {
      long na = (ncu * ta + (ta + tb) / 2) / (ta + tb);
      na = na < 1 ? 1 : (na > ncu - 1 ? ncu - 1 : na);
      qa.nx = (int)(na < ta ? na : ta);
      qb.nx = (int)(ncu - na < tb ? ncu - na : tb);
    }
    const dim3 rgrid((unsigned)(qa.nx + qb.nx));
#define MTRSSM_RES_LAUNCH_E(CIN_, NCT_, KS_, EPI_)                                                                   \
  {                                                                                                                   \
    static bool attr_done_dev[64] = {}; bool& attr_done = attr_done_dev[device_slot()];                                                                                    \
    const size_t rl = res_lds_bytes<CIN_, NCT_, KS_>();                                                               \
    if (!attr_done) {                                                                                                 \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_resident_kernel<CIN_, NCT_, KS_, EPI_>),        \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)rl);                                 \
      attr_done = true;                                                                                               \
    }                                                                                                                 \
    set_last_kernel("mtrssm::conv3x3_resident_kernel<" #CIN_ ", " #NCT_ ", " #KS_ ", " #EPI_ ">");                      \
    hipLaunchKernelGGL((conv3x3_resident_kernel<CIN_, NCT_, KS_, EPI_>), rgrid, dim3(kResThreads), rl, stream, qa, qb); \
    return launched("conv_gather_gemm(resident)");                                                                    \
  }
#define MTRSSM_RES_LAUNCH(CIN_, NCT_, KS_) \
  { if (epi_a) MTRSSM_RES_LAUNCH_E(CIN_, NCT_, KS_, true) else MTRSSM_RES_LAUNCH_E(CIN_, NCT_, KS_, false) }
    if (pl.res == 64064) MTRSSM_RES_LAUNCH(64, 2, 1)
    if (pl.res == 64128) MTRSSM_RES_LAUNCH(64, 4, 1)
    if (pl.res == 128064) MTRSSM_RES_LAUNCH(128, 2, 2)
    if (pl.res == 32064) MTRSSM_RES_LAUNCH(32, 2, 1)
    if (pl.res == 64032) MTRSSM_RES_LAUNCH(64, 1, 2)   // the first stack conv's backward-data: 32 output channels, K split over wave pairs, two frames per tile
#undef MTRSSM_RES_LAUNCH_E
#undef MTRSSM_RES_LAUNCH
    set_error("conv_gather_gemm: no resident kernel for this plan");
    return MTRSSM_EINVAL;
  }
  if (pl.kind == 1 && pl.stream != 0) {
    // conv1x1_stream_kernel: forward (bias + skip) or backward-data (act'(h)) of the residual blocks' 1x1 conv; anything else
    // (no operand, both operands) stays on conv1x1_split_kernel
    auto dir = [](const GatherProblem& q) { return q.add_in && !q.actgrad_in ? 1 : (q.actgrad_in && !q.add_in && !q.bias ? 2 : 0); };
    const int da = dir(pa), db = pb.nx > 0 ? dir(pb) : da;
    const bool same_shape = pb.nx == 0 || (pb.g.C == pa.g.C && pb.g.Cout == pa.g.Cout);
    if (da != 0 && da == db && same_shape) {
      const long ta = (long)pa.g.N * (pa.g.Hq * pa.g.Wq / 32), tb = pb.nx > 0 ? (long)pb.g.N * (pb.g.Hq * pb.g.Wq / 32) : 0;
      const long wa = (ta + 3) / 4, wb = (tb + 3) / 4;  // workgroups that would get at least one tile per wave
      const int ncu = cu_count();
      GatherProblem qa = pa, qb = pb;
      if (tb == 0) {
        qa.nx = (int)(wa < ncu ? wa : ncu);
        qb.nx = 0;
      } else {
        long na = (ncu * ta + (ta + tb) / 2) / (ta + tb);
        na = na < 1 ? 1 : (na > ncu - 1 ? ncu - 1 : na);
        qa.nx = (int)(na < wa ? na : wa);
        qb.nx = (int)(ncu - na < wb ? ncu - na : wb);
      }
      const dim3 sgrid((unsigned)(qa.nx + qb.nx));
#define MTRSSM_STREAM_LAUNCH(CIN_, COUT_, FWD_)                                                                      \
  {                                                                                                                   \
    set_last_kernel("mtrssm::conv1x1_stream_kernel<" #CIN_ ", " #COUT_ ", " #FWD_ ">");                                \
    hipLaunchKernelGGL((conv1x1_stream_kernel<CIN_, COUT_, FWD_>), sgrid, dim3(kResThreads), 0, stream, qa, qb);      \
    return launched("conv_gather_gemm(1x1 stream)");                                                                  \
  }
      if (pl.stream == 64064) { if (da == 1) MTRSSM_STREAM_LAUNCH(64, 64, true) else MTRSSM_STREAM_LAUNCH(64, 64, false) }
      if (pl.stream == 128064 && da == 1) MTRSSM_STREAM_LAUNCH(128, 64, true)
      if (pl.stream == 64128 && da == 2) MTRSSM_STREAM_LAUNCH(64, 128, false)
#undef MTRSSM_STREAM_LAUNCH
    }
  }
  const dim3 grid((unsigned)(pa.nx + pb.nx), pl.ny);
  const int sp = pl.sp;
#define MTRSSM_ATTR_ONCE(K_)                                                                                         \
  {                                                                                                                   \
    static bool attr_done_dev[64] = {}; bool& attr_done = attr_done_dev[device_slot()];                                                                                    \
    if (!attr_done) {                                                                                                 \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(K_), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024); \
      attr_done = true;                                                                                               \
    }                                                                                                                 \
  }
  if (pl.kind == 1) {
#define MTRSSM_1X1_LAUNCH(NT_, SP_)                                                                                  \
  {                                                                                                                   \
    MTRSSM_ATTR_ONCE((conv1x1_split_kernel<NT_, SP_>))                                                               \
    set_last_kernel("mtrssm::conv1x1_split_kernel<" #NT_ ", " #SP_ ">");                                               \
    hipLaunchKernelGGL((conv1x1_split_kernel<NT_, SP_>), grid, dim3(kConvThreads), lds, stream, pa, pb);              \
    return launched("conv_gather_gemm(1x1)");                                                                         \
  }
    if (pl.tco == 64) { if (sp == 3) MTRSSM_1X1_LAUNCH(2, 3) else if (sp == 2) MTRSSM_1X1_LAUNCH(2, 2) else MTRSSM_1X1_LAUNCH(2, 1) }
    else { if (sp == 3) MTRSSM_1X1_LAUNCH(1, 3) else if (sp == 2) MTRSSM_1X1_LAUNCH(1, 2) else MTRSSM_1X1_LAUNCH(1, 1) }
#undef MTRSSM_1X1_LAUNCH
  }
#define MTRSSM_SPLIT_LAUNCH(NT_, SP_, PIT_)                                                                          \
  {                                                                                                                   \
    MTRSSM_ATTR_ONCE((conv_gather_split_kernel<NT_, SP_, PIT_>))                                                     \
    set_last_kernel("mtrssm::conv_gather_split_kernel<" #NT_ ", " #SP_ ", " #PIT_ ">");                                \
    hipLaunchKernelGGL((conv_gather_split_kernel<NT_, SP_, PIT_>), grid, dim3(kConvThreads), lds, stream, pa, pb);    \
    return launched("conv_gather_gemm(split)");                                                                       \
  }
  const int pit = pl.pit;
  if (pl.tco == 64) {
    if (sp == 3) { if (pit == 3) MTRSSM_SPLIT_LAUNCH(2, 3, 3) else MTRSSM_SPLIT_LAUNCH(2, 3, 6) }
    else if (sp == 2) { if (pit == 3) MTRSSM_SPLIT_LAUNCH(2, 2, 3) else MTRSSM_SPLIT_LAUNCH(2, 2, 6) }
    else { if (pit == 3) MTRSSM_SPLIT_LAUNCH(2, 1, 3) else MTRSSM_SPLIT_LAUNCH(2, 1, 6) }
  } else {
    if (sp == 3) { if (pit == 3) MTRSSM_SPLIT_LAUNCH(1, 3, 3) else MTRSSM_SPLIT_LAUNCH(1, 3, 6) }
    else if (sp == 2) { if (pit == 3) MTRSSM_SPLIT_LAUNCH(1, 2, 3) else MTRSSM_SPLIT_LAUNCH(1, 2, 6) }
    else { if (pit == 3) MTRSSM_SPLIT_LAUNCH(1, 1, 3) else MTRSSM_SPLIT_LAUNCH(1, 1, 6) }
  }
#undef MTRSSM_SPLIT_LAUNCH
#undef MTRSSM_ATTR_ONCE
  set_error("conv_gather_gemm: no split kernel for this plan");
  return MTRSSM_EINVAL;
}

int debug_set_resident_profile(void* buf) {
  unsigned long long* p = static_cast<unsigned long long*>(buf);
  return hipMemcpyToSymbol(HIP_SYMBOL(g_res_prof), &p, sizeof(p)) == hipSuccess ? MTRSSM_OK : MTRSSM_ELAUNCH;
}

int pack_conv_weight_launch(const float* w, int O, int I, int KH, int KW, long so, long si, long sh, long sw, int OPad, int IPad,
                            int pieces, float* wp, unsigned short* wq, hipStream_t stream) {
  if (!w || !wp || O <= 0 || I <= 0 || KH < 0 || KW < 0 || OPad < O || IPad < I || pieces < 0 || pieces > 3 || (pieces > 0 && !wq)) {
    set_error("pack_conv_weight: bad argument");
    return MTRSSM_EINVAL;
  }
  const long total = (long)OPad * KH * KW * IPad;
  if (total == 0) return MTRSSM_OK;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 1024) blocks = 1024;
  { set_last_kernel("mtrssm::pack_conv_weight_kernel"); hipLaunchKernelGGL(pack_conv_weight_kernel, dim3(blocks), dim3(256), 0, stream, w, O, I, KH, KW, so, si, sh, sw, OPad, IPad, pieces, wp, wq); }
  return launched("pack_conv_weight");
}

int pack_conv_weights_launch(const int64_t* table, int count, int blocks_per_weight, hipStream_t stream) {
  if (count == 0) return MTRSSM_OK;
  if (!table || count < 0 || count > 65535 || blocks_per_weight < 1 || blocks_per_weight > 1024) {
    set_error("pack_conv_weights: bad argument");
    return MTRSSM_EINVAL;
  }
  static_assert(sizeof(long) == sizeof(int64_t), "descriptor words are 64-bit");
  { set_last_kernel("mtrssm::pack_conv_weights_kernel"); hipLaunchKernelGGL(pack_conv_weights_kernel, dim3(blocks_per_weight, count), dim3(256), 0, stream, reinterpret_cast<const long*>(table)); }
  return launched("pack_conv_weights");
}

// MTRSSM_CONV_S2_BAND=0: the thin / patch-staged kernels for the encoders' first two layers (A/B runs of conv3x3s2_band_kernel)
static bool s2_band_enabled() {
  static const bool on = [] { const char* e = getenv("MTRSSM_CONV_S2_BAND"); return !(e && e[0] == '0'); }();
  return on;
}
// conv3x3s2_band_kernel's shapes: Conv2d(k 3, s 2, p 1) forward, <= 8 input channels (coordinate channels included), <= 16 outputs
static bool s2_band_covers(const MtrssmConvGeom* g, bool has_wq, bool has_epilogue_operand) {
  return s2_band_enabled() && g->mfma_split == 2 && has_wq && !has_epilogue_operand && g->KH == 3 && g->KW == 3 && g->SS == 2 && g->TS == 1 &&
         g->OFFY == -1 && g->OFFX == -1 && g->OS == 1 && g->QY == 0 && g->QX == 0 && g->Hs == 2 * g->Hq && g->Ws == 2 * g->Wq &&
         g->Ho == g->Hq && g->Wo == g->Wq && (g->Wq == 8 || g->Wq == 16 || g->Wq == 32) && g->Hq % (kBandPx / g->Wq) == 0 &&
         ((g->C == 1 && g->C2 == 2) || (g->C == 8 && g->C2 == 0)) && g->Cout <= 16 && g->CoutPad == 32 && g->Cpad == 16 && g->act != MTRSSM_ACT_TANH &&
         (long)g->N * g->C * g->Hs * g->Ws < (1L << 31) && (long)g->N * g->Cout * g->Hq * g->Wq < (1L << 31);
}
static int launch_s2_band(const GatherProblem& pa, const GatherProblem* pb, hipStream_t stream) {
  const long ta = (long)pa.g.N * (pa.g.Hq * pa.g.Wq / kBandPx), tb = pb ? (long)pb->g.N * (pb->g.Hq * pb->g.Wq / kBandPx) : 0;
  const int slots = 2 * cu_count();  // two workgroups per CU
  GatherProblem qa = pa, qb{};
  if (pb) qb = *pb;
  if (tb == 0) {
    qa.nx = (int)(ta < slots ? ta : slots);
    qb.nx = 0;
  } else {
    long na = (slots * ta + (ta + tb) / 2) / (ta + tb);
    na = na < 1 ? 1 : (na > slots - 1 ? slots - 1 : na);
    qa.nx = (int)(na < ta ? na : ta);
    qb.nx = (int)(slots - na < tb ? slots - na : tb);
  }
  const int wmax = pb && pb->g.Wq < pa.g.Wq ? pb->g.Wq : pa.g.Wq;  // the narrower plane has the taller band: the larger image
  const size_t lds = (size_t)band_lds_bytes(wmax);
  if (pa.g.C == 1) {
    set_last_kernel("mtrssm::conv3x3s2_band_kernel<1, 2>");
    hipLaunchKernelGGL((conv3x3s2_band_kernel<1, 2>), dim3((unsigned)(qa.nx + qb.nx)), dim3(256), lds, stream, qa, qb);
  } else {
    set_last_kernel("mtrssm::conv3x3s2_band_kernel<8, 0>");
    hipLaunchKernelGGL((conv3x3s2_band_kernel<8, 0>), dim3((unsigned)(qa.nx + qb.nx)), dim3(256), lds, stream, qa, qb);
  }
  return launched("conv_gather_gemm(s2 band)");
}

// conv4s2_band_kernel's shapes: Conv2d(k 4, s 2, p 1) gather, 16 -> 32 channels on frames of 1024 positions and 32 -> 64 on frames of
// 256 (the backward-data of the decoders' second and first ConvTranspose layers); MTRSSM_CONV_S2_BAND=0 switches it off with the
// other band kernels
static int s2k4_band_covers(const MtrssmConvGeom* g, bool has_wq, bool has_add) {   // 0: no; 1: 16 -> 32 on 1024 positions; 2: 32 -> 64 on 256
  if (!(s2_band_enabled() && g->mfma_split == 2 && has_wq && !has_add && g->KH == 4 && g->KW == 4 && g->SS == 2 && g->TS == 1 &&
        g->OFFY == -1 && g->OFFX == -1 && g->OS == 1 && g->QY == 0 && g->QX == 0 && g->C2 == 0 && g->Hq * 2 == g->Hs && g->Wq * 2 == g->Ws &&
        g->Ho == g->Hq && g->Wo == g->Wq && g->act != MTRSSM_ACT_TANH && g->Cpad == g->C && g->CoutPad == g->Cout &&
        (long)g->N * g->C * g->Hs * g->Ws < (1L << 29)))
    return 0;
  if (g->C == 16 && g->Cout == 32 && g->Hs * g->Ws == 1024 && (g->Ws == 16 || g->Ws == 32)) return 1;
  if (g->C == 32 && g->Cout == 64 && g->Hs * g->Ws == 256 && (g->Ws == 8 || g->Ws == 16)) return 2;
  return 0;
}
static int launch_s2k4_band(const GatherProblem& pa, const GatherProblem* pb, hipStream_t stream) {
  const long ta = pa.g.N, tb = pb ? pb->g.N : 0;
  const int slots = cu_count();  // one 512-thread workgroup per CU
  GatherProblem qa = pa, qb{};
  if (pb) qb = *pb;
  if (tb == 0) {
    qa.nx = (int)(ta < slots ? ta : slots);
    qb.nx = 0;
  } else {
    long na = (slots * ta + (ta + tb) / 2) / (ta + tb);
    na = na < 1 ? 1 : (na > slots - 1 ? slots - 1 : na);
    qa.nx = (int)(na < ta ? na : ta);
    qb.nx = (int)(slots - na < tb ? slots - na : tb);
  }
  const int kind = s2k4_band_covers(&pa.g, true, false);
  const int oct = pa.g.C / 8;
  auto lds_of = [&](const MtrssmConvGeom& q) { return 2 * oct * ((q.Hs + 2) * (q.Ws + 2) + 1) * 16; };
  const int la = lds_of(pa.g), lb = pb ? lds_of(pb->g) : 0;
  const size_t lds = (size_t)(la > lb ? la : lb);
  static bool attr_done_dev[64][2] = {};
  bool& attr_done = attr_done_dev[device_slot()][kind - 1];
  if (kind == 1) {
    if (!attr_done) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv4s2_band_kernel<16, 32, 1024>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
      attr_done = true;
    }
    set_last_kernel("mtrssm::conv4s2_band_kernel<16, 32, 1024>");
    hipLaunchKernelGGL((conv4s2_band_kernel<16, 32, 1024>), dim3((unsigned)(qa.nx + qb.nx)), dim3(512), lds, stream, qa, qb);
  } else {
    if (!attr_done) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv4s2_band_kernel<32, 64, 256>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
      attr_done = true;
    }
    set_last_kernel("mtrssm::conv4s2_band_kernel<32, 64, 256>");
    hipLaunchKernelGGL((conv4s2_band_kernel<32, 64, 256>), dim3((unsigned)(qa.nx + qb.nx)), dim3(256), lds, stream, qa, qb);
  }
  return launched("conv_gather_gemm(s2 k4 band)");
}

int conv_gather_gemm_launch(const MtrssmConvGeom* g, const float* src, const float* src2, const float* wp, const unsigned short* wq,
                            const float* bias, const float* actgrad_in, const float* add_in, float* out, hipStream_t stream) {
  if (int rc = check_geom(g, "conv_gather_gemm")) return rc;
  if (!src || !wp || !out || (g->C2 > 0 && !src2)) { set_error("conv_gather_gemm: null pointer"); return MTRSSM_EINVAL; }
  if ((uintptr_t)wp & 15) { set_error("conv_gather_gemm: packed weights must be 16-byte aligned"); return MTRSSM_EINVAL; }
  const long ptot = (long)g->N * g->Hq * g->Wq;
  if (s2_band_covers(g, wq != nullptr, actgrad_in || add_in)) {
    GatherProblem p{};
    p.g = *g; p.src = src; p.src2 = src2; p.wq = wq; p.bias = bias; p.out = out;
    return launch_s2_band(p, nullptr, stream);
  }
  if (s2k4_band_covers(g, wq != nullptr, add_in != nullptr)) {
    GatherProblem p{};
    p.g = *g; p.src = src; p.wq = wq; p.bias = bias; p.actgrad_in = actgrad_in; p.out = out;
    return launch_s2k4_band(p, nullptr, stream);
  }
  // MTRSSM_THIN16_PAIRS=0: conv_gather_thin_kernel<16> for the one-channel k = 4 stride-2 gather too (A/B runs)
  static const bool thin16_pairs = [] { const char* e = getenv("MTRSSM_THIN16_PAIRS"); return !(e && e[0] == '0'); }();
  if (thin16_pairs && g->C == 1 && g->C2 == 0 && g->Cout == 16 && g->KH == 4 && g->KW == 4 && g->SS == 2 && g->TS == 1 && g->OFFY == -1 && g->OFFX == -1 &&
      g->OS == 1 && g->QY == 0 && g->QX == 0 && g->pre_act == 0 && !add_in && g->Hs == 2 * g->Hq && g->Ws == 2 * g->Wq && g->Ho == g->Hq &&
      g->Wo == g->Wq && g->Wq % 2 == 0 && g->Ws % 2 == 0 && g->act != MTRSSM_ACT_TANH && !((uintptr_t)src & 7) && !((uintptr_t)out & 7) && !((uintptr_t)actgrad_in & 7) &&
      kConvThreads == 256) {
    const long pairs = (long)g->N * g->Hq * (g->Wq / 2);
    set_last_kernel("mtrssm::conv4s2_c1_thin16_kernel");
    hipLaunchKernelGGL(conv4s2_c1_thin16_kernel, dim3((unsigned)((pairs + 255) / 256)), dim3(256), 0, stream, *g, src, wp, bias, actgrad_in, out);
    return launched("conv_gather_gemm(thin16 pairs)");
  }
  if ((g->Cout <= 8 || g->C + g->C2 <= 2) && g->Cout <= 16 && g->KH * g->KW * (g->C + g->C2) <= kThinMaxK) {  // thin layer: VALU kernel
    const dim3 grid((unsigned)((ptot + kConvThreads - 1) / kConvThreads));
    if (g->Cout <= 2)
      { set_last_kernel("mtrssm::conv_gather_thin_kernel<2>"); hipLaunchKernelGGL(conv_gather_thin_kernel<2>, grid, dim3(kConvThreads), 0, stream, *g, src, src2, wp, bias, actgrad_in, add_in, out); }
    else if (g->Cout <= 8)
      { set_last_kernel("mtrssm::conv_gather_thin_kernel<8>"); hipLaunchKernelGGL(conv_gather_thin_kernel<8>, grid, dim3(kConvThreads), 0, stream, *g, src, src2, wp, bias, actgrad_in, add_in, out); }
    else
      { set_last_kernel("mtrssm::conv_gather_thin_kernel<16>"); hipLaunchKernelGGL(conv_gather_thin_kernel<16>, grid, dim3(kConvThreads), 0, stream, *g, src, src2, wp, bias, actgrad_in, add_in, out); }
    return launched("conv_gather_gemm(thin)");
  }
  const int gx = (int)((ptot + kTP - 1) / kTP);
  if ((g->TS == 1 || g->TS == -1) && g->KH * g->KW > 0 && plane_fits_26bit(g)) {
    // Tile choice is occupancy-driven (measured, profiles/round1_notes.md): this kernel's MFMA pipe is fed by MANY
    // co-resident waves, not by big register tiles.  On the 64-channel 3x3 layers: 128 px x 64 ch workgroup tiles
    // (32 x 64 per wave) 177 us; 256 x 64 (64 x 64 per wave, NP = 2) 235 us; 128 x 32 (twice the workgroups) 176 us.
    const int tco = g->Cout > 32 ? 64 : 32;
    const int ny = g->CoutPad / tco;
    for (int np = 1; np >= 1; --np) {
      const int tpx = kTP * np;
      if (g->Wq > tpx || tpx % g->Wq) continue;
      const PatchGeom pg(*g, tpx);
      const bool tiles = (g->Hq % pg.rpg == 0) || (pg.rpg % g->Hq == 0);
      const size_t lds = ((size_t)kKC * pg.ps + 2 * (size_t)tco * (kKC + 1) + pg.ps_raw) * sizeof(float);
      const long nx = (ptot + tpx - 1) / tpx;
      if (!tiles || pg.ipg >= 32) continue;
      const dim3 grid((unsigned)nx, ny);
      {
        const SplitPlan pl = plan_split(g, wq != nullptr);
        if (pl.kind != 0) {
          GatherProblem none{};
          none.nx = 0;
          return launch_split(pl, pl.lds, make_problem(g, pl, src, src2, wq, bias, actgrad_in, add_in, out), none, stream);
        }
      }
      if (lds > 64 * 1024) continue;
      if (tco == 64) {
        { set_last_kernel("mtrssm::conv_gather_gemm_patch_kernel<2, 1>"); hipLaunchKernelGGL((conv_gather_gemm_patch_kernel<2, 1>), grid, dim3(kConvThreads), lds, stream, *g, src, src2, wp, bias, actgrad_in, add_in, out); }
      } else {
        { set_last_kernel("mtrssm::conv_gather_gemm_patch_kernel<1, 1>"); hipLaunchKernelGGL((conv_gather_gemm_patch_kernel<1, 1>), grid, dim3(kConvThreads), lds, stream, *g, src, src2, wp, bias, actgrad_in, add_in, out); }
      }
      return launched("conv_gather_gemm(patch)");
    }
  }
  if (g->Cout > 32) {
    dim3 grid(gx, g->CoutPad / 64);
    { set_last_kernel("mtrssm::conv_gather_gemm_kernel<2>"); hipLaunchKernelGGL(conv_gather_gemm_kernel<2>, grid, dim3(kConvThreads), 0, stream, *g, src, src2, wp, bias, actgrad_in, add_in, out); }
  } else {
    dim3 grid(gx, 1);
    { set_last_kernel("mtrssm::conv_gather_gemm_kernel<1>"); hipLaunchKernelGGL(conv_gather_gemm_kernel<1>, grid, dim3(kConvThreads), 0, stream, *g, src, src2, wp, bias, actgrad_in, add_in, out); }
  }
  return launched("conv_gather_gemm");
}

// Two gather problems in one launch when they map to the same split kernel (the audio and the vision branch of one layer);
// otherwise two launches.
// 1 when the two problems run the same split kernel and therefore go out as ONE grid (mtrssm_conv_gather_pair_merges)
int conv_gather_pair_merges(const MtrssmConvGeom* ga, const MtrssmConvGeom* gb, bool has_wq) {
  if (!ga || !gb || check_geom(ga, "conv_gather_gemm_pair") || check_geom(gb, "conv_gather_gemm_pair")) return 0;
  // (the band kernel's pair: epilogue operands are checked again at launch, where a pair with one falls back to two launches)
  if (s2_band_covers(ga, has_wq, false) && s2_band_covers(gb, has_wq, false) && ga->C == gb->C) return 1;
  if (s2k4_band_covers(ga, has_wq, false) && s2k4_band_covers(ga, has_wq, false) == s2k4_band_covers(gb, has_wq, false)) return 1;
  const bool thin_a = (ga->Cout <= 8 || ga->C + ga->C2 <= 2) && ga->Cout <= 16;
  const bool thin_b = (gb->Cout <= 8 || gb->C + gb->C2 <= 2) && gb->Cout <= 16;
  if (thin_a || thin_b) return 0;
  const SplitPlan pa = plan_split(ga, has_wq), pb = plan_split(gb, has_wq);
  return pa.same_kernel(pb) ? 1 : 0;
}

int conv_gather_gemm_pair_launch(const MtrssmConvGeom* ga, const float* srca, const float* src2a, const float* wpa, const unsigned short* wqa,
                                 const float* biasa, const float* actgrada, const float* adda, float* outa, const MtrssmConvGeom* gb,
                                 const float* srcb, const float* src2b, const float* wpb, const unsigned short* wqb, const float* biasb,
                                 const float* actgradb, const float* addb, float* outb, hipStream_t stream) {
  if (srca && srcb && outa && outb && ga && gb && !(ga->C2 > 0 && !src2a) && !(gb->C2 > 0 && !src2b) && !check_geom(ga, "conv_gather_gemm_pair") &&
      !check_geom(gb, "conv_gather_gemm_pair") && s2_band_covers(ga, wqa != nullptr, actgrada || adda) && s2_band_covers(gb, wqb != nullptr, actgradb || addb) && ga->C == gb->C) {
    GatherProblem p{}, q{};
    p.g = *ga; p.src = srca; p.src2 = src2a; p.wq = wqa; p.bias = biasa; p.out = outa;
    q.g = *gb; q.src = srcb; q.src2 = src2b; q.wq = wqb; q.bias = biasb; q.out = outb;
    return launch_s2_band(p, &q, stream);
  }
  if (srca && srcb && outa && outb && ga && gb && !check_geom(ga, "conv_gather_gemm_pair") && !check_geom(gb, "conv_gather_gemm_pair") &&
      s2k4_band_covers(ga, wqa != nullptr, adda != nullptr) &&
      s2k4_band_covers(ga, wqa != nullptr, adda != nullptr) == s2k4_band_covers(gb, wqb != nullptr, addb != nullptr)) {
    GatherProblem p{}, q{};
    p.g = *ga; p.src = srca; p.wq = wqa; p.bias = biasa; p.actgrad_in = actgrada; p.out = outa;
    q.g = *gb; q.src = srcb; q.wq = wqb; q.bias = biasb; q.actgrad_in = actgradb; q.out = outb;
    return launch_s2k4_band(p, &q, stream);
  }
  if (srca && srcb && outa && outb && !(ga && ga->C2 > 0 && !src2a) && !(gb && gb->C2 > 0 && !src2b) &&
      conv_gather_pair_merges(ga, gb, wqa != nullptr && wqb != nullptr) &&
      !(s2_band_covers(ga, true, false) && s2_band_covers(gb, true, false) && ga->C == gb->C) &&
      !(s2k4_band_covers(ga, true, false) && s2k4_band_covers(ga, true, false) == s2k4_band_covers(gb, true, false))) {
    const SplitPlan pa = plan_split(ga, true), pb = plan_split(gb, true);
    return launch_split(pa, pa.lds > pb.lds ? pa.lds : pb.lds, make_problem(ga, pa, srca, src2a, wqa, biasa, actgrada, adda, outa),
                        make_problem(gb, pb, srcb, src2b, wqb, biasb, actgradb, addb, outb), stream);
  }
  if (int rc = conv_gather_gemm_launch(ga, srca, src2a, wpa, wqa, biasa, actgrada, adda, outa, stream)) return rc;
  return conv_gather_gemm_launch(gb, srcb, src2b, wpb, wqb, biasb, actgradb, addb, outb, stream);
}

// MTRSSM_RESBLOCK_FUSE=0: the residual blocks' forward as two launches (3x3, then 1x1 + skip) instead of the fused kernel
static bool resblock_fuse_enabled() {
  static const bool on = [] { const char* e = getenv("MTRSSM_RESBLOCK_FUSE"); return !(e && e[0] == '0'); }();
  return on;
}

// Forward of a whole residual block  y = x + Conv1x1(act(h)),  h = b3 + Conv3x3(act(x))  in one launch
// (conv3x3_resident_kernel<..., FUSE>): g is the 3x3's geometry exactly as for conv_gather_gemm (pre_act = 1), w1 [C][Cout] and
// b1 [C] the 1x1 module's own fp32 parameters.  Returns the kernel key (> 0) when the shape has a fused instance.
int conv_residual_fwd_supported(const MtrssmConvGeom* g) {
  if (!g || !resblock_fuse_enabled() || g->pre_act == 0 || g->mfma_split != 2 || g->C2 != 0) return 0;
  const SplitPlan pl = plan_split(g, true);
  return pl.kind == 3 && (pl.res == 64128 || pl.res == 64064) ? pl.res : 0;
}

static int residual_fwd_one_grid(const GatherProblem& pa, const GatherProblem& pb, int key, hipStream_t stream) {
  const int fpt = key == 64064 ? 2 : 1;  // frames per tile (the plan takes whole tiles only)
  const long ta = pa.nx / fpt, tb = pb.nx / fpt;
  const int ncu = cu_count();
  GatherProblem qa = pa, qb = pb;
  if (tb == 0) {
    qa.nx = (int)(ta < ncu ? ta : ncu);
    qb.nx = 0;
  } else {
    long na = (ncu * ta + (ta + tb) / 2) / (ta + tb);
    na = na < 1 ? 1 : (na > ncu - 1 ? ncu - 1 : na);
    qa.nx = (int)(na < ta ? na : ta);
    qb.nx = (int)(ncu - na < tb ? ncu - na : tb);
  }
  const dim3 rgrid((unsigned)(qa.nx + qb.nx));
#define MTRSSM_FUSE_LAUNCH(NCT_)                                                                                        \
  {                                                                                                                    \
    static bool attr_done_dev[64] = {};                                                                                \
    bool& attr_done = attr_done_dev[device_slot()];                                                                    \
    const size_t rl = res_lds_bytes<64, NCT_, 1>() + res_fuse_bytes<64, NCT_>();                                       \
    if (!attr_done) {                                                                                                  \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_resident_kernel<64, NCT_, 1, false, true>),      \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)rl);                                  \
      attr_done = true;                                                                                                \
    }                                                                                                                  \
    set_last_kernel("mtrssm::conv3x3_resident_kernel<64, " #NCT_ ", 1, false, true>");                                  \
    hipLaunchKernelGGL((conv3x3_resident_kernel<64, NCT_, 1, false, true>), rgrid, dim3(kResThreads), rl, stream, qa, qb); \
    return launched("residual_block_fwd");                                                                             \
  }
  if (key == 64128) MTRSSM_FUSE_LAUNCH(4)
  if (key == 64064) MTRSSM_FUSE_LAUNCH(2)
#undef MTRSSM_FUSE_LAUNCH
  set_error("residual_block_fwd: no fused kernel for this shape");
  return MTRSSM_EINVAL;
}

int conv_residual_fwd_launch(const MtrssmConvGeom* ga, const float* xa, const unsigned short* wq3a, const float* b3a, const float* w1a,
                             const float* b1a, float* ha, float* ya, const MtrssmConvGeom* gb, const float* xb, const unsigned short* wq3b,
                             const float* b3b, const float* w1b, const float* b1b, float* hb, float* yb, hipStream_t stream) {
  if (!ga || !xa || !wq3a || !b3a || !w1a || !b1a || !ha || !ya) {
    set_error("residual_block_fwd: null argument");
    return MTRSSM_EINVAL;
  }
  if (gb && (!xb || !wq3b || !b3b || !w1b || !b1b || !hb || !yb)) {
    set_error("residual_block_fwd: null argument in the second problem");
    return MTRSSM_EINVAL;
  }
  if (int rc = check_geom(ga, "residual_block_fwd")) return rc;
  if (gb)
    if (int rc = check_geom(gb, "residual_block_fwd")) return rc;
  const int ka = conv_residual_fwd_supported(ga), kb = gb ? conv_residual_fwd_supported(gb) : ka;
  if (ka == 0 || kb == 0) {
    set_error("residual_block_fwd: shape without a fused kernel (query mtrssm_residual_block_fwd_supported first)");
    return MTRSSM_EINVAL;
  }
  auto problem = [](const MtrssmConvGeom* g, const float* x, const unsigned short* wq, const float* b3, const float* w1, const float* b1,
                    float* h, float* y) {
    GatherProblem p{};
    p.g = *g; p.src = x; p.wq = wq; p.bias = b3; p.out = h; p.w1 = w1; p.b1 = b1; p.out2 = y;
    p.nx = g->N;
    return p;
  };
  const GatherProblem pa = problem(ga, xa, wq3a, b3a, w1a, b1a, ha, ya);
  GatherProblem none{};
  none.nx = 0;
  if (!gb) return residual_fwd_one_grid(pa, none, ka, stream);
  const GatherProblem pb = problem(gb, xb, wq3b, b3b, w1b, b1b, hb, yb);
  if (ka == kb) return residual_fwd_one_grid(pa, pb, ka, stream);
  if (int rc = residual_fwd_one_grid(pa, none, ka, stream)) return rc;
  return residual_fwd_one_grid(pb, none, kb, stream);
}

int channel_sum_launch(const float* x, int N, int C, int HW, float* out, hipStream_t stream);

// MTRSSM_NO_DIRECT_WGRAD=1: the patch-staged kernels for every layer (A/B runs of the register-direct 3x3 kernel)
// MTRSSM_WGRAD_RESIDENT=0: the register-direct 3x3 kernel instead of the staged-input one (A/B runs)
static bool wgrad_resident_enabled() {
  static const bool on = [] { const char* e = getenv("MTRSSM_WGRAD_RESIDENT"); return !(e && e[0] == '0'); }();
  return on;
}
// MTRSSM_WGRAD_RESIDENT_C32=0: the 32-channel input layer on the register-direct kernel (A/B runs)
static bool wgrad_resident_c32() {
  static const bool on = [] { const char* e = getenv("MTRSSM_WGRAD_RESIDENT_C32"); return !(e && e[0] == '0'); }();
  return on;
}
// MTRSSM_WGRAD_1X1_STAGED=0: the register-direct 1x1 kernel instead of the staged one (A/B runs)
static bool wgrad_1x1_staged_enabled() {
  static const bool on = [] { const char* e = getenv("MTRSSM_WGRAD_1X1_STAGED"); return !(e && e[0] == '0'); }();
  return on;
}
// MTRSSM_WGRAD_S2_STAGED=0: the patch-staged kernel for the second encoder layer instead of the staged one (A/B runs)
static bool wgrad_s2_staged_enabled() {
  static const bool on = [] { const char* e = getenv("MTRSSM_WGRAD_S2_STAGED"); return !(e && e[0] == '0'); }();
  return on;
}
// MTRSSM_WGRAD_PARTIALS=0: the staged-input kernel adds its tiles to dwp by atomics instead of storing partial sets
static bool wgrad_partials_enabled() {
  static const bool on = [] { const char* e = getenv("MTRSSM_WGRAD_PARTIALS"); return !(e && e[0] == '0'); }();
  return on;
}
// Partial tile sets of the staged weight-gradient kernels live in the CALLER's workspace (the library allocates nothing).  In
// query mode the launch function reports the size and returns before any launch.
#define MTRSSM_WGRAD_PART(bytes_expr)                                                                                              \
  float* part = nullptr;                                                                                                         \
  {                                                                                                                              \
    const size_t need_ = (bytes_expr);                                                                                           \
    if (query) { *query = wgrad_partials_enabled() ? need_ : 0; return MTRSSM_OK; }                                              \
    if (wgrad_partials_enabled() && workspace && workspace_bytes >= need_) part = static_cast<float*>(workspace);                \
  }
#define MTRSSM_WGRAD_NO_PART \
  if (query) { *query = 0; return MTRSSM_OK; }
static bool no_direct_wgrad() {
  static const bool off = getenv("MTRSSM_NO_DIRECT_WGRAD") != nullptr;
  return off;
}

// workspace: caller-owned device memory for the partial tile sets of the staged kernels (mtrssm_conv_weight_grad_workspace_bytes);
// too small / NULL = the atomics form of the same kernels.  query != NULL: store the bytes this geometry's kernel wants (0 for
// the kernels without partial sets) and return WITHOUT launching anything.
int conv_weight_grad_launch(const MtrssmConvGeom* g, const float* a, const float* src, const float* src2, int pre_act_a,
                            float* dwp, float* dbias, void* workspace, size_t workspace_bytes, size_t* query, hipStream_t stream);
// the same launch with its partial-set reduction recorded for conv_weight_grad_reduce_flush (the workspace must stay untouched
// until then)
int conv_weight_grad_deferred_launch(const MtrssmConvGeom* g, const float* a, const float* src, const float* src2, int pre_act_a,
                                     float* dwp, float* dbias, void* workspace, size_t workspace_bytes, hipStream_t stream) {
  tl_defer_reduce = true;
  const int rc = conv_weight_grad_launch(g, a, src, src2, pre_act_a, dwp, dbias, workspace, workspace_bytes, nullptr, stream);
  tl_defer_reduce = false;
  return rc;
}

int conv_weight_grad_launch(const MtrssmConvGeom* g, const float* a, const float* src, const float* src2, int pre_act_a,
                            float* dwp, float* dbias, void* workspace, size_t workspace_bytes, size_t* query, hipStream_t stream) {
  if (int rc = check_geom(g, "conv_weight_grad")) return rc;
  if (!query && (!a || !src || !dwp || (g->C2 > 0 && !src2))) { set_error("conv_weight_grad: null pointer"); return MTRSSM_EINVAL; }
  if (workspace && ((uintptr_t)workspace & 255)) { set_error("conv_weight_grad: workspace must be 256-byte aligned"); return MTRSSM_EINVAL; }
  if (dbias && pre_act_a) { set_error("conv_weight_grad: the fused bias gradient sums the RAW a tensor; pass dbias only with pre_act_a = 0"); return MTRSSM_EINVAL; }
  if (g->OS != 1 || g->QY != 0 || g->QX != 0 || g->KH * g->KW <= 0) { set_error("conv_weight_grad: needs a plain (OS=1) geometry with taps"); return MTRSSM_EINVAL; }
  const long ptot = (long)g->N * g->Hq * g->Wq;
  const int taps = g->KH * g->KW;
  const int ctot = g->C + g->C2;
  if ((g->mfma_split == 1 || g->mfma_split == 2) && taps == 1 && g->SS == 1 && g->OFFY == 0 && g->OFFX == 0 && g->C2 == 0 && g->Hs == g->Hq &&
      g->Ws == g->Wq && g->Hq * g->Wq == 64 && (g->C == 64 || g->C == 128) && g->Cout % 64 == 0 && g->Cout <= 65535 * 64 && g->Cpad >= g->C &&
      !pre_act_a && (g->act == MTRSSM_ACT_IDENTITY || g->act == MTRSSM_ACT_ELU || g->act == MTRSSM_ACT_RELU) && !((uintptr_t)a & 15) &&
      !((uintptr_t)src & 15) && wgrad_1x1_staged_enabled()) {
    // 1x1 layers of the residual stacks on 64-pixel planes: operands staged once per frame (conv_wgrad_resident.h)
    const int cogroups = g->Cout / 64;
    int wgs = cu_count() / cogroups;
    if (wgs < 1) wgs = 1;
    if (wgs > g->N) wgs = g->N;
    const int per = (g->N + wgs - 1) / wgs;
    const dim3 grid((unsigned)((g->N + per - 1) / per), cogroups);
    MTRSSM_WGRAD_PART((size_t)grid.x * cogroups * kWg1x1SetFloats * sizeof(float))
#define MTRSSM_WG1_LAUNCH(SP_, C_)                                                                                              \
  {                                                                                                                             \
    static bool attr_done_dev[64] = {}; bool& attr_done = attr_done_dev[device_slot()]; /* once per instantiation: never while another stream may be running the kernel */       \
    constexpr int lds_b = wg1x1_lds_bytes<SP_, C_>();                                                                           \
    if (!attr_done) {                                                                                                           \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv1x1_wgrad_staged_kernel<SP_, C_>),                            \
                                hipFuncAttributeMaxDynamicSharedMemorySize, lds_b);                                             \
      attr_done = true;                                                                                                         \
    }                                                                                                                           \
    set_last_kernel("mtrssm::conv1x1_wgrad_staged_kernel<" #SP_ ", " #C_ ">");                                                  \
    hipLaunchKernelGGL((conv1x1_wgrad_staged_kernel<SP_, C_>), grid, dim3(512), lds_b, stream, *g, a, src, dwp, part, dbias,    \
                       per);                                                                                                    \
  }
    const int sp = g->mfma_split;
    if (g->C == 64) { if (sp == 2) MTRSSM_WG1_LAUNCH(2, 64) else MTRSSM_WG1_LAUNCH(1, 64) }
    else { if (sp == 2) MTRSSM_WG1_LAUNCH(2, 128) else MTRSSM_WG1_LAUNCH(1, 128) }
#undef MTRSSM_WG1_LAUNCH
    if (part) {
      const dim3 rgrid((unsigned)(2 * (g->C / 32) * 256 / 32 + (dbias ? 1 : 0)), cogroups);  // + the bias block
      if (int rc = reduce_now_or_later(g->C == 64 ? kRed1x1_64 : kRed1x1_128, rgrid, part, (int)grid.x, g->Cpad, dwp, dbias, stream)) return rc;
    }
    return launched("conv_weight_grad(1x1 staged)");
  }
  if (g->mfma_split >= 1 && taps == 1 && g->SS == 1 && g->OFFY == 0 && g->OFFX == 0 && g->C2 == 0 && g->Hs == g->Hq && g->Ws == g->Wq &&
      (g->Hq * g->Wq) % 16 == 0 && g->Cout <= 128 && g->C <= 128 && g->C >= 8 && !((uintptr_t)a & 15) && !((uintptr_t)src & 15) &&
      ptot < (1L << 31)) {
    // 1x1 layers: both operands straight from HBM into MFMA registers (conv_split.h: conv1x1_weight_grad_split_kernel)
    MTRSSM_WGRAD_NO_PART
    const int tiles_co = (g->Cout + 31) / 32, tiles_ci = (g->C + 31) / 32;
    const int total = (int)(ptot / 16);
    int splits = 256;  // one workgroup per CU (one wave per SIMD: the conversion VALU work fills it); more partial tiles = more atomics (36 / 44 / 50 us at 256 / 512 / 1024)
    if (splits > total) splits = total;
    const int per = (total + splits - 1) / splits;
    const dim3 grid((unsigned)((total + per - 1) / per));
    const dim3 block(64 * tiles_co * tiles_ci);
    const int sp = g->mfma_split;
    set_last_kernel(sp == 3 ? "mtrssm::conv1x1_weight_grad_split_kernel<3>" : sp == 2 ? "mtrssm::conv1x1_weight_grad_split_kernel<2>" : "mtrssm::conv1x1_weight_grad_split_kernel<1>");
    if (sp == 3) hipLaunchKernelGGL((conv1x1_weight_grad_split_kernel<3>), grid, block, 0, stream, *g, a, src, pre_act_a, dwp, dbias, tiles_ci, per);
    else if (sp == 2) hipLaunchKernelGGL((conv1x1_weight_grad_split_kernel<2>), grid, block, 0, stream, *g, a, src, pre_act_a, dwp, dbias, tiles_ci, per);
    else hipLaunchKernelGGL((conv1x1_weight_grad_split_kernel<1>), grid, block, 0, stream, *g, a, src, pre_act_a, dwp, dbias, tiles_ci, per);
    return launched("conv_weight_grad(1x1 split)");
  }
  if ((g->mfma_split == 1 || g->mfma_split == 2) && g->KH == 3 && g->KW == 3 && g->SS == 1 && g->TS == 1 && g->OFFY == -1 && g->OFFX == -1 &&
      g->C2 == 0 && g->Hs == g->Hq && g->Ws == g->Wq && (g->Wq == 8 || g->Wq == 4) && g->Hq * g->Wq == 64 && (g->C == 64 || (g->C == 32 && wgrad_resident_c32())) &&
      g->Cout % 64 == 0 && g->Cout <= 65535 * 64 && g->Cpad >= g->C && !pre_act_a &&
      (g->act == MTRSSM_ACT_IDENTITY || g->act == MTRSSM_ACT_ELU || g->act == MTRSSM_ACT_RELU) && !((uintptr_t)a & 15) &&
      !((uintptr_t)src & 15) && !((uintptr_t)dwp & 15) && wgrad_resident_enabled() && !no_direct_wgrad()) {
    // 3x3 layers of the residual stacks on 64-pixel planes: both operands staged once per frame (conv_wgrad_resident.h)
    const int cogroups = g->Cout / 64;
    int wgs = (g->C == 64 ? cu_count() : 2 * cu_count()) / cogroups;  // one wave per SIMD over the chip
    if (wgs < 1) wgs = 1;
    if (wgs > g->N) wgs = g->N;
    const int per = (g->N + wgs - 1) / wgs;
    const dim3 grid((unsigned)((g->N + per - 1) / per), cogroups);
    const size_t set_floats = (size_t)wgres_set_floats(g->C);
    MTRSSM_WGRAD_PART((size_t)grid.x * cogroups * set_floats * sizeof(float))
#define MTRSSM_WGRES_LAUNCH(SP_, C_, W_)                                                                                         \
  {                                                                                                                             \
    static bool attr_done_dev[64] = {}; bool& attr_done = attr_done_dev[device_slot()]; /* once per instantiation: never while another stream may be running the kernel */       \
    constexpr int lds_b = wgres_lds_bytes<SP_, C_, W_>();                                                                       \
    if (!attr_done) {                                                                                                           \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_wgrad_resident_kernel<SP_, C_, W_>),                       \
                                hipFuncAttributeMaxDynamicSharedMemorySize, lds_b);                                             \
      attr_done = true;                                                                                                         \
    }                                                                                                                           \
    set_last_kernel("mtrssm::conv3x3_wgrad_resident_kernel<" #SP_ ", " #C_ ", " #W_ ">");                                        \
    hipLaunchKernelGGL((conv3x3_wgrad_resident_kernel<SP_, C_, W_>), grid, dim3(64 * 2 * (C_ / 32)), lds_b, stream, *g, a, src,  \
                       dwp, part, dbias, per);                                                                                  \
  }
    const int sp = g->mfma_split;
    if (g->C == 64) {
      if (g->Wq == 8) { if (sp == 2) MTRSSM_WGRES_LAUNCH(2, 64, 8) else MTRSSM_WGRES_LAUNCH(1, 64, 8) }
      else { if (sp == 2) MTRSSM_WGRES_LAUNCH(2, 64, 4) else MTRSSM_WGRES_LAUNCH(1, 64, 4) }
    } else {
      if (g->Wq == 8) { if (sp == 2) MTRSSM_WGRES_LAUNCH(2, 32, 8) else MTRSSM_WGRES_LAUNCH(1, 32, 8) }
      else { if (sp == 2) MTRSSM_WGRES_LAUNCH(2, 32, 4) else MTRSSM_WGRES_LAUNCH(1, 32, 4) }
    }
#undef MTRSSM_WGRES_LAUNCH
    if (part) {
      const dim3 rgrid((unsigned)(wgres_tile_floats(g->C) / 4 / 32 + (dbias ? 1 : 0)), cogroups);  // + the bias block
      if (int rc = reduce_now_or_later(g->C == 64 ? kRedRes64 : kRedRes32, rgrid, part, (int)grid.x, g->Cpad, dwp, dbias, stream)) return rc;
    }
    return launched("conv_weight_grad(3x3 resident)");
  }
  if (g->mfma_split >= 1 && g->KH == 3 && g->KW == 3 && g->SS == 1 && g->TS == 1 && g->OFFY == -1 && g->OFFX == -1 && g->C2 == 0 &&
      g->Hs == g->Hq && g->Ws == g->Wq && (g->Wq == 8 || g->Wq == 4) && (g->Hq * g->Wq) % 16 == 0 && g->C <= 64 && g->C >= 8 &&
      g->Cout <= 65535 * 64 && !((uintptr_t)a & 15) && !((uintptr_t)src & 15) && ptot < (1L << 31) && !no_direct_wgrad()) {
    // 3x3 layers of the residual stacks: register-direct (conv_split.h: conv3x3_weight_grad_split_kernel)
    MTRSSM_WGRAD_NO_PART
    const int cogroups = (g->Cout + 63) / 64;
    const int tiles_co = g->Cout > 32 ? 2 : 1, tiles_ci = (g->C + 31) / 32;
    const int nt = tiles_co * tiles_ci;
    const int total = (int)(ptot / 16);
    int splits = 1024 / (nt * cogroups);  // one wave per SIMD over the chip
    if (splits < 1) splits = 1;
    if (splits > total) splits = total;
    const int per = (total + splits - 1) / splits;
    const dim3 grid((unsigned)((total + per - 1) / per), cogroups);
    const dim3 block(64 * nt);
    const int sp = g->mfma_split;
#define MTRSSM_W33_LAUNCH(SP_, W_)                                                                                               \
  {                                                                                                                             \
    set_last_kernel("mtrssm::conv3x3_weight_grad_split_kernel<" #SP_ ", " #W_ ">");                                              \
    hipLaunchKernelGGL((conv3x3_weight_grad_split_kernel<SP_, W_>), grid, block, 0, stream, *g, a, src, pre_act_a, dwp, dbias, tiles_ci, per); \
  }
    if (g->Wq == 8) { if (sp == 3) MTRSSM_W33_LAUNCH(3, 8) else if (sp == 2) MTRSSM_W33_LAUNCH(2, 8) else MTRSSM_W33_LAUNCH(1, 8) }
    else { if (sp == 3) MTRSSM_W33_LAUNCH(3, 4) else if (sp == 2) MTRSSM_W33_LAUNCH(2, 4) else MTRSSM_W33_LAUNCH(1, 4) }
#undef MTRSSM_W33_LAUNCH
    return launched("conv_weight_grad(3x3 split)");
  }
  if ((g->mfma_split == 1 || g->mfma_split == 2) && g->KH == 3 && g->KW == 3 && g->SS == 2 && g->TS == 1 && g->OFFY == -1 && g->OFFX == -1 &&
      g->C == 1 && g->C2 == 2 && g->Cout == 8 && g->Hq * g->Wq == 1024 && (g->Wq == 32 || g->Wq == 16) && g->Hs == 2 * g->Hq && g->Ws == 2 * g->Wq &&
      g->Cpad >= 3 && !pre_act_a && (g->act == MTRSSM_ACT_IDENTITY || g->act == MTRSSM_ACT_ELU || g->act == MTRSSM_ACT_RELU || !g->pre_act) &&
      !((uintptr_t)a & 15) && !((uintptr_t)src & 15) && !((uintptr_t)src2 & 15) && wgrad_s2_staged_enabled()) {
    // first encoder layer (3x3 / stride 2, frame channel + 2 coordinate channels -> 8, 1024-pixel output planes): staged
    int wgs = cu_count();
    if (wgs > g->N) wgs = g->N;
    const int per = (g->N + wgs - 1) / wgs;
    const dim3 grid((unsigned)((g->N + per - 1) / per));
    MTRSSM_WGRAD_PART((size_t)grid.x * kWgThinSetFloats * sizeof(float))
#define MTRSSM_WGTHIN_LAUNCH(SP_, W_)                                                                                            \
  {                                                                                                                             \
    static bool attr_done_dev[64] = {}; bool& attr_done = attr_done_dev[device_slot()]; /* once per instantiation: never while another stream may be running the kernel */       \
    constexpr int lds_b = wgthin_lds_bytes<SP_, W_>();                                                                          \
    if (!attr_done) {                                                                                                           \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3s2_thin_wgrad_staged_kernel<SP_, W_>),                      \
                                hipFuncAttributeMaxDynamicSharedMemorySize, lds_b);                                             \
      attr_done = true;                                                                                                         \
    }                                                                                                                           \
    set_last_kernel("mtrssm::conv3x3s2_thin_wgrad_staged_kernel<" #SP_ ", " #W_ ">");                                            \
    hipLaunchKernelGGL((conv3x3s2_thin_wgrad_staged_kernel<SP_, W_>), grid, dim3(512), lds_b, stream, *g, a, src, src2, dwp,     \
                       part, dbias, per);                                                                                       \
  }
    const int sp = g->mfma_split;
    if (g->Wq == 32) { if (sp == 2) MTRSSM_WGTHIN_LAUNCH(2, 32) else MTRSSM_WGTHIN_LAUNCH(1, 32) }
    else { if (sp == 2) MTRSSM_WGTHIN_LAUNCH(2, 16) else MTRSSM_WGTHIN_LAUNCH(1, 16) }
#undef MTRSSM_WGTHIN_LAUNCH
    if (part)
      if (int rc = reduce_now_or_later(kRedThin, dim3(dbias ? 9 : 8), part, (int)grid.x, g->Cpad, dwp, dbias, stream)) return rc;
    return launched("conv_weight_grad(3x3 s2 thin staged)");
  }
  if ((g->mfma_split == 1 || g->mfma_split == 2) && g->KH == 4 && g->KW == 4 && g->SS == 2 && g->TS == 1 && g->OFFY == -1 && g->OFFX == -1 &&
      g->C == 1 && g->C2 == 0 && g->Cout == 16 && g->Hq * g->Wq == 1024 && (g->Wq == 32 || g->Wq == 16) && g->Hs == 2 * g->Hq && g->Ws == 2 * g->Wq &&
      g->Cpad >= 1 && !g->pre_act && !dbias && (g->act == MTRSSM_ACT_IDENTITY || g->act == MTRSSM_ACT_ELU || g->act == MTRSSM_ACT_RELU || !pre_act_a) &&
      !((uintptr_t)a & 15) && !((uintptr_t)src & 15) && wgrad_s2_staged_enabled()) {
    // the decoders' last ConvTranspose2d (k = 4 / stride 2, 16 -> 1, 1024-pixel input planes): operands staged once per frame
    int wgs = cu_count();
    if (wgs > g->N) wgs = g->N;
    const int per = (g->N + wgs - 1) / wgs;
    const dim3 grid((unsigned)((g->N + per - 1) / per));
    MTRSSM_WGRAD_PART((size_t)grid.x * kWgThinTSetFloats * sizeof(float))
#define MTRSSM_WGTHINT_LAUNCH(SP_, W_)                                                                                           \
  {                                                                                                                             \
    static bool attr_done_dev[64] = {}; bool& attr_done = attr_done_dev[device_slot()]; /* once per instantiation: never while another stream may be running the kernel */       \
    constexpr int lds_b = wgthint_lds_bytes<SP_, W_>();                                                                         \
    if (!attr_done) {                                                                                                           \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(convt4s2_thin_wgrad_staged_kernel<SP_, W_>),                       \
                                hipFuncAttributeMaxDynamicSharedMemorySize, lds_b);                                             \
      attr_done = true;                                                                                                         \
    }                                                                                                                           \
    set_last_kernel("mtrssm::convt4s2_thin_wgrad_staged_kernel<" #SP_ ", " #W_ ">");                                             \
    hipLaunchKernelGGL((convt4s2_thin_wgrad_staged_kernel<SP_, W_>), grid, dim3(512), lds_b, stream, *g, a, src, pre_act_a, dwp, \
                       part, per);                                                                                              \
  }
    const int sp = g->mfma_split;
    if (g->Wq == 32) { if (sp == 2) MTRSSM_WGTHINT_LAUNCH(2, 32) else MTRSSM_WGTHINT_LAUNCH(1, 32) }
    else { if (sp == 2) MTRSSM_WGTHINT_LAUNCH(2, 16) else MTRSSM_WGTHINT_LAUNCH(1, 16) }
#undef MTRSSM_WGTHINT_LAUNCH
    if (part)
      if (int rc = reduce_now_or_later(kRedThinT, dim3(16), part, (int)grid.x, g->Cpad, dwp, nullptr, stream)) return rc;
    return launched("conv_weight_grad(k4 s2 thin staged)");
  }
  if (g->mfma_split >= 1 && g->Cout <= 32 && taps * ctot <= 32 && g->Wq >= 8 && (g->Wq & (g->Wq - 1)) == 0 && (g->Hq * g->Wq) % 16 == 0 &&
      !((uintptr_t)a & 15) && ptot < (1L << 31) && (long)g->N * g->C * g->Hs * g->Ws < (1L << 40) && !no_direct_wgrad()) {
    // thin strided layers: one MFMA tile, A straight from HBM, B gathered per lane (conv_split.h: conv_weight_grad_thin_split_kernel)
    MTRSSM_WGRAD_NO_PART
    int log2_wq = 0;
    while ((1 << log2_wq) < g->Wq) ++log2_wq;
    const int total = (int)(ptot / 16);
    int waves = 4096;  // 4 per SIMD: 17 loads in flight per lane each; 16 waves per workgroup share one set of atomics
    if (waves > total) waves = total;
    const int per = (total + waves - 1) / waves;
    waves = (total + per - 1) / per;
    const dim3 grid((unsigned)((waves + 15) / 16));
    const int sp = g->mfma_split;
    set_last_kernel(sp == 3 ? "mtrssm::conv_weight_grad_thin_split_kernel<3>" : sp == 2 ? "mtrssm::conv_weight_grad_thin_split_kernel<2>" : "mtrssm::conv_weight_grad_thin_split_kernel<1>");
    if (sp == 3) hipLaunchKernelGGL((conv_weight_grad_thin_split_kernel<3>), grid, dim3(1024), 0, stream, *g, a, src, src2, pre_act_a, dwp, dbias, per, log2_wq);
    else if (sp == 2) hipLaunchKernelGGL((conv_weight_grad_thin_split_kernel<2>), grid, dim3(1024), 0, stream, *g, a, src, src2, pre_act_a, dwp, dbias, per, log2_wq);
    else hipLaunchKernelGGL((conv_weight_grad_thin_split_kernel<1>), grid, dim3(1024), 0, stream, *g, a, src, src2, pre_act_a, dwp, dbias, per, log2_wq);
    return launched("conv_weight_grad(thin split)");
  }
  if ((g->mfma_split == 1 || g->mfma_split == 2) && g->KH == 3 && g->KW == 3 && g->SS == 2 && g->TS == 1 && g->OFFY == -1 && g->OFFX == -1 &&
      g->C2 == 0 && g->C == 8 && g->Cout == 16 && g->Hq * g->Wq == 256 && (g->Wq == 16 || g->Wq == 8) && g->Hs == 2 * g->Hq && g->Ws == 2 * g->Wq &&
      g->Cpad >= 8 && !pre_act_a && (g->act == MTRSSM_ACT_IDENTITY || g->act == MTRSSM_ACT_ELU || g->act == MTRSSM_ACT_RELU) &&
      !((uintptr_t)a & 15) && !((uintptr_t)src & 15) && wgrad_s2_staged_enabled()) {
    // second encoder layer (3x3 / stride 2, 8 -> 16, 256-pixel output planes): operands staged once per frame (conv_wgrad_resident.h)
    int wgs = cu_count();
    if (wgs > g->N) wgs = g->N;
    const int per = (g->N + wgs - 1) / wgs;
    const dim3 grid((unsigned)((g->N + per - 1) / per));
    MTRSSM_WGRAD_PART((size_t)grid.x * kWgS2SetFloats * sizeof(float))
#define MTRSSM_WGS2_LAUNCH(SP_, W_)                                                                                              \
  {                                                                                                                             \
    static bool attr_done_dev[64] = {}; bool& attr_done = attr_done_dev[device_slot()]; /* once per instantiation: never while another stream may be running the kernel */       \
    constexpr int lds_b = wgs2_lds_bytes<SP_, W_>();                                                                            \
    if (!attr_done) {                                                                                                           \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3s2_wgrad_staged_kernel<SP_, W_>),                           \
                                hipFuncAttributeMaxDynamicSharedMemorySize, lds_b);                                             \
      attr_done = true;                                                                                                         \
    }                                                                                                                           \
    set_last_kernel("mtrssm::conv3x3s2_wgrad_staged_kernel<" #SP_ ", " #W_ ">");                                                 \
    hipLaunchKernelGGL((conv3x3s2_wgrad_staged_kernel<SP_, W_>), grid, dim3(512), lds_b, stream, *g, a, src, dwp, part, dbias,   \
                       per);                                                                                                    \
  }
    const int sp = g->mfma_split;
    if (g->Wq == 16) { if (sp == 2) MTRSSM_WGS2_LAUNCH(2, 16) else MTRSSM_WGS2_LAUNCH(1, 16) }
    else { if (sp == 2) MTRSSM_WGS2_LAUNCH(2, 8) else MTRSSM_WGS2_LAUNCH(1, 8) }
#undef MTRSSM_WGS2_LAUNCH
    if (part)
      if (int rc = reduce_now_or_later(kRedS2, dim3(dbias ? 49 : 48), part, (int)grid.x, g->Cpad, dwp, dbias, stream)) return rc;
    return launched("conv_weight_grad(3x3 s2 staged)");
  }
  if ((g->mfma_split == 1 || g->mfma_split == 2) && g->KH == 4 && g->KW == 4 && g->SS == 2 && g->TS == 1 && g->OFFY == -1 && g->OFFX == -1 &&
      g->C2 == 0 && g->C == 16 && g->Cout == 32 && g->Hq * g->Wq == 256 && (g->Wq == 16 || g->Wq == 8) && g->Hs == 2 * g->Hq && g->Ws == 2 * g->Wq &&
      g->Cpad >= 16 && !g->pre_act && !dbias && (g->act == MTRSSM_ACT_IDENTITY || g->act == MTRSSM_ACT_ELU || g->act == MTRSSM_ACT_RELU || !pre_act_a) &&
      !((uintptr_t)a & 15) && !((uintptr_t)src & 15) && wgrad_s2_staged_enabled()) {
    // the decoders' second ConvTranspose2d (k = 4 / stride 2, 32 -> 16, 256-pixel input planes): operands staged once per frame
    int wgs = cu_count();
    if (wgs > g->N) wgs = g->N;
    const int per = (g->N + wgs - 1) / wgs;
    const dim3 grid((unsigned)((g->N + per - 1) / per));
    MTRSSM_WGRAD_PART((size_t)grid.x * kWgT4SetFloats * sizeof(float))
#define MTRSSM_WGT4_LAUNCH(SP_, W_)                                                                                              \
  {                                                                                                                             \
    static bool attr_done_dev[64] = {}; bool& attr_done = attr_done_dev[device_slot()]; /* once per instantiation: never while another stream may be running the kernel */       \
    constexpr int lds_b = wgt4_lds_bytes<SP_, W_>();                                                                            \
    if (!attr_done) {                                                                                                           \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(convt4s2_wgrad_staged_kernel<SP_, W_>),                            \
                                hipFuncAttributeMaxDynamicSharedMemorySize, lds_b);                                             \
      attr_done = true;                                                                                                         \
    }                                                                                                                           \
    set_last_kernel("mtrssm::convt4s2_wgrad_staged_kernel<" #SP_ ", " #W_ ">");                                                  \
    hipLaunchKernelGGL((convt4s2_wgrad_staged_kernel<SP_, W_>), grid, dim3(512), lds_b, stream, *g, a, src, pre_act_a, dwp,      \
                       part, per);                                                                                              \
  }
    const int sp = g->mfma_split;
    if (g->Wq == 16) { if (sp == 2) MTRSSM_WGT4_LAUNCH(2, 16) else MTRSSM_WGT4_LAUNCH(1, 16) }
    else { if (sp == 2) MTRSSM_WGT4_LAUNCH(2, 8) else MTRSSM_WGT4_LAUNCH(1, 8) }
#undef MTRSSM_WGT4_LAUNCH
    if (part)
      if (int rc = reduce_now_or_later(kRedT4, dim3(kWgT4SetFloats / 4 / 8), part, (int)grid.x, g->Cpad, dwp, nullptr, stream)) return rc;
    return launched("conv_weight_grad(k4 s2 staged)");
  }
  if ((g->mfma_split == 1 || g->mfma_split == 2) && g->KH == 4 && g->KW == 4 && g->SS == 2 && g->TS == 1 && g->OFFY == -1 && g->OFFX == -1 &&
      g->C2 == 0 && g->C == 32 && g->Cout == 64 && g->Hq * g->Wq == 64 && (g->Wq == 8 || g->Wq == 4) && g->Hs == 2 * g->Hq && g->Ws == 2 * g->Wq &&
      g->Cpad >= 32 && !g->pre_act && !dbias && (g->act == MTRSSM_ACT_IDENTITY || g->act == MTRSSM_ACT_ELU || g->act == MTRSSM_ACT_RELU || !pre_act_a) &&
      !((uintptr_t)a & 15) && !((uintptr_t)src & 15) && wgrad_s2_staged_enabled()) {
    // the decoders' first ConvTranspose2d (k = 4 / stride 2, 64 -> 32, 64-pixel input planes): operands staged once per frame
    int wgs = cu_count();
    if (wgs > g->N) wgs = g->N;
    const int per = (g->N + wgs - 1) / wgs;
    const dim3 grid((unsigned)((g->N + per - 1) / per));
    MTRSSM_WGRAD_PART((size_t)grid.x * kWgT4bSetFloats * sizeof(float))
#define MTRSSM_WGT4B_LAUNCH(SP_, W_)                                                                                             \
  {                                                                                                                             \
    static bool attr_done_dev[64] = {}; bool& attr_done = attr_done_dev[device_slot()]; /* once per instantiation: never while another stream may be running the kernel */       \
    constexpr int lds_b = wgt4b_lds_bytes<SP_, W_>();                                                                           \
    if (!attr_done) {                                                                                                           \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(convt4s2b_wgrad_staged_kernel<SP_, W_>),                           \
                                hipFuncAttributeMaxDynamicSharedMemorySize, lds_b);                                             \
      attr_done = true;                                                                                                         \
    }                                                                                                                           \
    set_last_kernel("mtrssm::convt4s2b_wgrad_staged_kernel<" #SP_ ", " #W_ ">");                                                 \
    hipLaunchKernelGGL((convt4s2b_wgrad_staged_kernel<SP_, W_>), grid, dim3(512), lds_b, stream, *g, a, src, pre_act_a, dwp,     \
                       part, per);                                                                                              \
  }
    const int sp = g->mfma_split;
    if (g->Wq == 8) { if (sp == 2) MTRSSM_WGT4B_LAUNCH(2, 8) else MTRSSM_WGT4B_LAUNCH(1, 8) }
    else { if (sp == 2) MTRSSM_WGT4B_LAUNCH(2, 4) else MTRSSM_WGT4B_LAUNCH(1, 4) }
#undef MTRSSM_WGT4B_LAUNCH
    if (part)
      if (int rc = reduce_now_or_later(kRedT4b, dim3(kWgT4bSetFloats / 4 / 8), part, (int)grid.x, g->Cpad, dwp, nullptr, stream)) return rc;
    return launched("conv_weight_grad(k4 s2 staged, 64 rows)");
  }
  if ((g->mfma_split == 1 || g->mfma_split == 2) && g->KH == 3 && g->KW == 3 && g->SS == 2 && g->TS == 1 && g->OFFY == -1 && g->OFFX == -1 &&
      g->C2 == 0 && g->C == 16 && g->Cout == 32 && g->Hq * g->Wq == 64 && (g->Wq == 8 || g->Wq == 4) && g->Hs == 2 * g->Hq && g->Ws == 2 * g->Wq &&
      g->Cpad >= 16 && !pre_act_a && (g->act == MTRSSM_ACT_IDENTITY || g->act == MTRSSM_ACT_ELU || g->act == MTRSSM_ACT_RELU) &&
      !((uintptr_t)a & 15) && !((uintptr_t)src & 15) && wgrad_s2_staged_enabled()) {
    // third encoder layer (3x3 / stride 2, 16 -> 32, 64-pixel output planes): operands staged once per frame
    int wgs = cu_count();
    if (wgs > g->N) wgs = g->N;
    const int per = (g->N + wgs - 1) / wgs;
    const dim3 grid((unsigned)((g->N + per - 1) / per));
    MTRSSM_WGRAD_PART((size_t)grid.x * kWgS2cSetFloats * sizeof(float))
#define MTRSSM_WGS2C_LAUNCH(SP_, W_)                                                                                             \
  {                                                                                                                             \
    set_last_kernel("mtrssm::conv3x3s2c_wgrad_staged_kernel<" #SP_ ", " #W_ ">");                                                \
    hipLaunchKernelGGL((conv3x3s2c_wgrad_staged_kernel<SP_, W_>), grid, dim3(512), wgs2c_lds_bytes<SP_>(), stream, *g, a, src,   \
                       dwp, part, dbias, per);                                                                                  \
  }
    const int sp = g->mfma_split;
    if (g->Wq == 8) { if (sp == 2) MTRSSM_WGS2C_LAUNCH(2, 8) else MTRSSM_WGS2C_LAUNCH(1, 8) }
    else { if (sp == 2) MTRSSM_WGS2C_LAUNCH(2, 4) else MTRSSM_WGS2C_LAUNCH(1, 4) }
#undef MTRSSM_WGS2C_LAUNCH
    if (part)
      if (int rc = reduce_now_or_later(kRedS2c, dim3(dbias ? 161 : 160), part, (int)grid.x, g->Cpad, dwp, dbias, stream)) return rc;
    return launched("conv_weight_grad(3x3 s2 staged, 16 -> 32)");
  }
  MTRSSM_WGRAD_NO_PART
  // ---- patch-staged kernel when the 64-pixel groups tile the frames exactly
  if (g->TS == 1 && g->Wq <= kGP && kGP % g->Wq == 0) {
    const PatchGeom pg(*g, kGP);
    const bool tiles = (g->Hq % pg.rpg == 0) || (pg.rpg % g->Hq == 0);
    const int nq = (taps * ctot + 31) / 32;
    const int tco = g->Cout > 32 ? 64 : 32;
    const size_t lds = (2 * ((size_t)tco * kLDA + (size_t)ctot * pg.ps) + kGP) * sizeof(float);
    const int n_out = g->Cout * taps * ctot;
    const size_t lds_thin = (2 * ((size_t)g->Cout * kLDA + (size_t)ctot * pg.ps) + kGP + ((pg.ps_raw + 63) & ~63)) * sizeof(float);
    const bool mfma_ok = nq <= 4 * kMaxQ && lds <= 150 * 1024;
    (void)mfma_ok;
    if (tiles && n_out <= kThinOut * kConvThreads && lds_thin <= 64 * 1024 && ptot < (1L << 31) && pg.ps_raw < 1024 && pg.ipg < 1024 &&
        !(g->mfma_split >= 1 && ctot >= 8 && (g->Hq * g->Wq) % 8 == 0)) {  // thin layer: staging-bound, VALU reduction is faster
      const long groups = (ptot + kGP - 1) / kGP;
      long splits = 2048;  // 8 small workgroups per CU: the kernel is staging-latency-bound, occupancy hides it
      if (splits > groups) splits = groups;
      { set_last_kernel("mtrssm::conv_weight_grad_thin_kernel"); hipLaunchKernelGGL(conv_weight_grad_thin_kernel, dim3((unsigned)splits), dim3(kConvThreads), lds_thin, stream, *g, a, src, src2, pre_act_a, dwp, dbias); }
      return launched("conv_weight_grad(thin)");
    }
    if (tiles && g->mfma_split >= 1 && (g->Hq * g->Wq) % 8 == 0 && ctot <= 128 && !((uintptr_t)a & 15)) {
      // split-bf16 operands (conv_split.h): same persistent structure, bf16 MFMA + transposed LDS reads
      const int sp = g->mfma_split;
      int cp2 = 16;
      while (cp2 < ctot) cp2 *= 2;
      const int nhalf = taps * (cp2 / 16);
      const int tco_s = g->Cout > 32 ? 64 : 32;
      const int nblk = (pg.ps_raw + 63) / 64;
      const size_t lds_1 = (size_t)sp * ((size_t)tco_s * kWgLdaB + (size_t)pg.ps_raw * cp2 * 2);
      // double-buffered when it fits; a single buffer (stage / compute alternate) for big multi-round patches
      const int nbuf = 2 * lds_1 + (size_t)nblk * 64 * sizeof(int) <= 156 * 1024 || nblk * (cp2 / 8) <= 16 ? 2 : 1;
      const size_t lds_s = nbuf * lds_1 + (size_t)nblk * 64 * sizeof(int);
      if ((nhalf + 1) / 2 <= 4 * kMaxQ && pg.ps_raw < 1024 && pg.ipg < 1024 && lds_s <= 156 * 1024 &&
          (long)pg.ipg * g->C * g->Hs * g->Ws < (1L << 31) && ptot < (1L << 31)) {
        const long groups = (ptot + kGP - 1) / kGP;
        const int cotiles = g->Cout > 32 ? g->CoutPad / 64 : 1;
        long splits = 256 / cotiles;
        if (splits > groups) splits = groups;
        if (splits < 1) splits = 1;
        dim3 grid((unsigned)splits, cotiles);
#define MTRSSM_WG_SPLIT_LAUNCH(NT_, SP_)                                                                                       \
  {                                                                                                                             \
    static bool attr_done_dev[64] = {}; bool& attr_done = attr_done_dev[device_slot()]; /* once per instantiation: never while another stream may be running the kernel */       \
    if (!attr_done) {                                                                                                           \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_weight_grad_split_kernel<NT_, SP_>),                         \
                                hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);                                        \
      attr_done = true;                                                                                                         \
    }                                                                                                                           \
    set_last_kernel("mtrssm::conv_weight_grad_split_kernel<" #NT_ ", " #SP_ ">");                                                \
    hipLaunchKernelGGL((conv_weight_grad_split_kernel<NT_, SP_>), grid, dim3(2 * kConvThreads), lds_s, stream, *g, a, src, src2, \
                       pre_act_a, dwp, dbias, cp2, nbuf);                                                                             \
  }
        if (g->Cout > 32) { if (sp == 3) MTRSSM_WG_SPLIT_LAUNCH(2, 3) else if (sp == 2) MTRSSM_WG_SPLIT_LAUNCH(2, 2) else MTRSSM_WG_SPLIT_LAUNCH(2, 1) }
        else { if (sp == 3) MTRSSM_WG_SPLIT_LAUNCH(1, 3) else if (sp == 2) MTRSSM_WG_SPLIT_LAUNCH(1, 2) else MTRSSM_WG_SPLIT_LAUNCH(1, 1) }
#undef MTRSSM_WG_SPLIT_LAUNCH
        return launched("conv_weight_grad(split)");
      }
    }
    if (tiles && nq <= 4 * kMaxQ && lds <= 150 * 1024) {
      const long groups = (ptot + kGP - 1) / kGP;
      const int cotiles = g->Cout > 32 ? g->CoutPad / 64 : 1;
      long splits = 256 / cotiles;  // one 8-wave workgroup per CU (loader waves hide the staging); few atomics
      if (splits > groups) splits = groups;
      if (splits < 1) splits = 1;
      dim3 grid((unsigned)splits, cotiles);
      if (g->Cout > 32) {
        if (lds > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_weight_grad_patch_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        { set_last_kernel("mtrssm::conv_weight_grad_patch_kernel<2>"); hipLaunchKernelGGL(conv_weight_grad_patch_kernel<2>, grid, dim3(2 * kConvThreads), lds, stream, *g, a, src, src2, pre_act_a, dwp, dbias); }
      } else {
        if (lds > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_weight_grad_patch_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        { set_last_kernel("mtrssm::conv_weight_grad_patch_kernel<1>"); hipLaunchKernelGGL(conv_weight_grad_patch_kernel<1>, grid, dim3(2 * kConvThreads), lds, stream, *g, a, src, src2, pre_act_a, dwp, dbias); }
      }
      return launched("conv_weight_grad(patch)");
    }
  }
  // ---- general kernel
  const long groups = (ptot + kTP - 1) / kTP;
  const int ctiles = g->Cpad / 32 + (g->Cpad % 32 ? 1 : 0);
  // enough pixel splits to fill the chip (~4 workgroups per CU) without drowning in atomics
  long want = 1024 / ((long)ctiles * taps);
  if (want < 1) want = 1;
  if (want > groups) want = groups;
  if (want > 512) want = 512;
  dim3 grid(ctiles, taps, (int)want);
  if (dbias) {  // the general kernel has no fused bias reduction
    if (int rc = channel_sum_launch(a, g->N, g->Cout, g->Hq * g->Wq, dbias, stream)) return rc;
  }
  if (g->Cout > 32)
    { set_last_kernel("mtrssm::conv_weight_grad_kernel<2>"); hipLaunchKernelGGL(conv_weight_grad_kernel<2>, grid, dim3(kConvThreads), 0, stream, *g, a, src, src2, pre_act_a, dwp); }
  else
    { set_last_kernel("mtrssm::conv_weight_grad_kernel<1>"); hipLaunchKernelGGL(conv_weight_grad_kernel<1>, grid, dim3(kConvThreads), 0, stream, *g, a, src, src2, pre_act_a, dwp); }
  return launched("conv_weight_grad");
}

// The decoders' last layer on the MFMA (conv_s2_band.h: convt4s2_band_kernel): 16 input channels, <= 2 output channels, frames
// of 1024 positions (64 x 16 or 32 x 32); two bf16 pieces per operand.  Returns MTRSSM_EINVAL outside these shapes (the caller
// takes mtrssm_convt_k4s2_thin then).
int convt_k4s2_band_supported(int N, int C, int Hs, int Ws, int Cout) {
  return N > 0 && C == 16 && Cout >= 1 && Cout <= 2 && Hs * Ws == 1024 && (Ws == 16 || Ws == 32) && (long)N * C * 1024 < (1L << 29) ? 1 : 0;
}
int convt_k4s2_band_launch(int N, int C, int Hs, int Ws, int Cout, const float* src, const float* w, const float* bias, int pre_act,
                           int act, float* out, hipStream_t stream) {
  if (!convt_k4s2_band_supported(N, C, Hs, Ws, Cout) || !src || !w || !out) {
    set_error("convt_k4s2_band: needs 16 input channels, 1-2 output channels and 1024-position frames (64 x 16 or 32 x 32)");
    return MTRSSM_EINVAL;
  }
  if (act < MTRSSM_ACT_IDENTITY || act > MTRSSM_ACT_TANH) { set_error("convt_k4s2_band: unknown activation id %d", act); return MTRSSM_EINVAL; }
  if (((uintptr_t)out & 7) || ((uintptr_t)src & 15)) { set_error("convt_k4s2_band: src must be 16-byte, out 8-byte aligned"); return MTRSSM_EINVAL; }
  ConvtBandProblem p{}, none{};
  p.src = src; p.w = w; p.bias = bias; p.out = out; p.N = N; p.Hs = Hs; p.Ws = Ws; p.Cout = Cout; p.pre_act = pre_act; p.act = act;
  const int slots = 2 * cu_count();   // two workgroups per CU
  p.nx = N < slots ? N : slots;
  none.nx = 0;
  const size_t lds = (size_t)convt_band_lds_bytes(Hs, Ws);
  static bool attr_done_dev[64] = {};
  bool& attr_done = attr_done_dev[device_slot()];
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(convt4s2_band_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    attr_done = true;
  }
  set_last_kernel("mtrssm::convt4s2_band_kernel");
  hipLaunchKernelGGL(convt4s2_band_kernel, dim3((unsigned)p.nx), dim3(256), lds, stream, p, none);
  return launched("convt_k4s2_band");
}

int convt_k4s2_thin_launch(int N, int C, int Hs, int Ws, int Cout, const float* src, const float* w, const float* bias, int pre_act,
                           int act, float* out, hipStream_t stream) {
  if (N <= 0 || C <= 0 || Hs <= 0 || Ws <= 0 || Cout <= 0 || Cout > 2 || !src || !w || !out) {
    set_error("convt_k4s2_thin: bad argument (needs 1 <= Cout <= 2)");
    return MTRSSM_EINVAL;
  }
  if (act < MTRSSM_ACT_IDENTITY || act > MTRSSM_ACT_TANH) { set_error("convt_k4s2_thin: unknown activation id %d", act); return MTRSSM_EINVAL; }
  if ((uintptr_t)out & 7) { set_error("convt_k4s2_thin: out must be 8-byte aligned"); return MTRSSM_EINVAL; }
  const size_t lds = (size_t)C * kCtPS * sizeof(float);
  if (lds > 64 * 1024) { set_error("convt_k4s2_thin: %d input channels do not fit the LDS tile", C); return MTRSSM_ELDS; }
  const int tw = Ws <= 16 ? 16 : 32, th = kConvThreads / tw;
  const dim3 grid((Ws + tw - 1) / tw, (Hs + th - 1) / th, N);
#define MTRSSM_CT_LAUNCH(COT_, TW_)                                                                                  \
  { set_last_kernel("mtrssm::convt_k4s2_thin_kernel<" #COT_ ", " #TW_ ">"); hipLaunchKernelGGL((convt_k4s2_thin_kernel<COT_, TW_>), grid, dim3(kConvThreads), lds, stream, N, C, Hs, Ws, Cout, src, w, bias, pre_act, act, out); }
  if (Cout == 1) { if (tw == 16) MTRSSM_CT_LAUNCH(1, 16) else MTRSSM_CT_LAUNCH(1, 32) }
  else { if (tw == 16) MTRSSM_CT_LAUNCH(2, 16) else MTRSSM_CT_LAUNCH(2, 32) }
#undef MTRSSM_CT_LAUNCH
  return launched("convt_k4s2_thin");
}

int conv_tgather_thin_launch(int N, int O, int Hs, int Ws, int Cc, int KH, int KW, int S, int P, int Ho, int Wo, const float* y, const float* w,
                             const float* bias, int pre_act, int act, const float* actgrad_in, const float* add_in, float* out,
                             hipStream_t stream) {
  if (N <= 0 || O <= 0 || Hs <= 0 || Ws <= 0 || Cc <= 0 || Cc > 8 || KH <= 0 || KW <= 0 || S <= 0 || P < 0 || Ho <= 0 || Wo <= 0 || !y || !w ||
      !out || KH * KW * O > kThinMaxK) {
    set_error("conv_tgather_thin: bad argument (needs 1 <= Cout <= 8 and taps * channels <= %d)", kThinMaxK);
    return MTRSSM_EINVAL;
  }
  if (act < MTRSSM_ACT_IDENTITY || act > MTRSSM_ACT_TANH) { set_error("conv_tgather_thin: unknown activation id %d", act); return MTRSSM_EINVAL; }
  const long ptot = (long)N * Ho * Wo;
  if (ptot >= (1L << 31) * (long)kConvThreads) { set_error("conv_tgather_thin: too many output pixels"); return MTRSSM_EINVAL; }
  const dim3 grid((unsigned)((ptot + kConvThreads - 1) / kConvThreads));
  if (S != 2 || Ho % 2 || Wo % 2) { set_error("conv_tgather_thin: stride 2 and even output planes only (stride %d, %d x %d)", S, Ho, Wo); return MTRSSM_EINVAL; }
  if (Cc <= 2)
    { set_last_kernel("mtrssm::conv_tgather_thin_kernel<2, 2>"); hipLaunchKernelGGL((conv_tgather_thin_kernel<2, 2>), grid, dim3(kConvThreads), 0, stream, N, O, Hs, Ws, Cc, KH, KW, P, Ho, Wo, y, w, bias, pre_act, act, actgrad_in, add_in, out); }
  else
    { set_last_kernel("mtrssm::conv_tgather_thin_kernel<8, 2>"); hipLaunchKernelGGL((conv_tgather_thin_kernel<8, 2>), grid, dim3(kConvThreads), 0, stream, N, O, Hs, Ws, Cc, KH, KW, P, Ho, Wo, y, w, bias, pre_act, act, actgrad_in, add_in, out); }
  return launched("conv_tgather_thin");
}

// All four output parity classes of a k = 4, s = 2, p = 1 ConvTranspose2d forward in one pass (conv_resident.h:
// convt_quad_resident_kernel); optionally two tensors (audio + vision) in one launch.  Returns MTRSSM_EINVAL when the layer is
// outside the kernel's shapes: ask conv_convt_quad_supported first.
static int quad_key(const MtrssmConvGeom* g4) {
  if (!g4) return 0;
  const MtrssmConvGeom& g0 = g4[0];
  for (int q = 0; q < 4; ++q) {
    const MtrssmConvGeom& g = g4[q];
    if (g.KH != 2 || g.KW != 2 || g.TS != -1 || g.SS != 1 || g.OS != 2 || g.C2 != 0 || g.mfma_split != 2 || g.QY != (q >> 1) || g.QX != (q & 1) ||
        g.C != g0.C || g.Cout != g0.Cout || g.Hs != g0.Hs || g.Ws != g0.Ws || g.N != g0.N || g.Ho != 2 * g.Hs || g.Wo != 2 * g.Ws ||
        g.Hq != g.Hs || g.Wq != g.Ws || g.Cpad != g.C || g.CoutPad != 32 || g.act == MTRSSM_ACT_TANH || g.OFFY < 0 || g.OFFY > 1 ||
        g.OFFX < 0 || g.OFFX > 1)
      return 0;
  }
  if (g0.Ws & (g0.Ws - 1)) return 0;
  if ((long)g0.N * g0.Cout * g0.Ho * g0.Wo >= (1L << 31) || (long)g0.N * g0.C * g0.Hs * g0.Ws >= (1L << 31)) return 0;
  const int plane = g0.Hs * g0.Ws;
  if (g0.C == 64 && g0.Cout == 32 && plane == 64) return 1;
  if (g0.C == 32 && g0.Cout == 16 && plane == 256) return 2;
  if (g0.C == 32 && g0.Cout == 16 && plane == 64) return 3;  // backward-data of the encoders' third conv (k = 3 zero-padded to 4)
  if (g0.C == 16 && g0.Cout == 8 && plane == 256) return 4;  // ... of their second conv (8 of the tile's 32 output rows in use: HBM-bound)
  return 0;
}

int conv_convt_quad_supported(const MtrssmConvGeom* g4) { return resident_enabled() ? quad_key(g4) : 0; }

int conv_convt_quad_launch(const MtrssmConvGeom* ga4, const float* srca, const unsigned short* const* wqa4, const float* biasa,
                           const float* actgrada, float* outa, const MtrssmConvGeom* gb4, const float* srcb,
                           const unsigned short* const* wqb4, const float* biasb, const float* actgradb, float* outb, hipStream_t stream) {
  const int key = quad_key(ga4);
  const bool epi = actgrada != nullptr;
  if (!key || !srca || !wqa4 || !outa || (gb4 && (quad_key(gb4) != key || !srcb || !wqb4 || !outb || (actgradb != nullptr) != epi)) ||
      (epi != (key >= 3))) {
    set_error("convt_quad: layer outside the kernel's shapes (k4 s2 p1: 64 -> 32 on 64-pixel planes, 32 -> 16 on 256-pixel planes; with act' operand: 32 -> 16 on 64-pixel planes, 16 -> 8 on 256-pixel planes; two bf16 pieces)");
    return MTRSSM_EINVAL;
  }
  QuadProblem qa{}, qb{};
  for (int q = 0; q < 4; ++q) {
    qa.g[q] = ga4[q]; qa.wq[q] = wqa4[q];
    if (!wqa4[q]) { set_error("convt_quad: null packed weights"); return MTRSSM_EINVAL; }
    if (gb4) { qb.g[q] = gb4[q]; qb.wq[q] = wqb4[q]; if (!wqb4[q]) { set_error("convt_quad: null packed weights"); return MTRSSM_EINVAL; } }
  }
  qa.src = srca; qa.bias = biasa; qa.actgrad = actgrada; qa.out = outa;
  qb.src = srcb; qb.bias = biasb; qb.actgrad = actgradb; qb.out = outb;
  const long ta = ga4[0].N, tb = gb4 ? gb4[0].N : 0;
  const int ncu = cu_count();
  if (tb == 0) {
    qa.nx = (int)(ta < ncu ? ta : ncu);
    qb.nx = 0;
  } else {
    long na = (ncu * ta + (ta + tb) / 2) / (ta + tb);
    na = na < 1 ? 1 : (na > ncu - 1 ? ncu - 1 : na);
    qa.nx = (int)(na < ta ? na : ta);
    qb.nx = (int)(ncu - na < tb ? ncu - na : tb);
  }
  const dim3 grid((unsigned)(qa.nx + qb.nx));
  // The classes as tile rows (convt4s2_rows_kernel) where that form is faster: 32 -> 16 on 256-position frames (198 -> 158 us
  // paired) and 16 -> 8 with the act' operand (144 -> 88 us); the 64-position frames stay on one wave per class (the rows form
  // multiplies 2.25 x as much: 124 -> 189 us, 80 -> 103 us).  MTRSSM_CONVT_ROWS=<digits>: the keys (1..4) that use it ("0": none).
  static const unsigned rows_mask = [] {
    const char* e = getenv("MTRSSM_CONVT_ROWS");
    if (!e) return (1u << 2) | (1u << 4);
    unsigned m = 0;
    for (; *e; ++e)
      if (*e >= '1' && *e <= '4') m |= 1u << (*e - '0');
    return m;
  }();
#define MTRSSM_ROWS_LAUNCH(CIN_, COUT_, PLANE_, EPI_)                                                                \
  {                                                                                                                   \
    static bool attr_done_dev[64] = {}; bool& attr_done = attr_done_dev[device_slot()];                               \
    const size_t ql = convt_rows_lds_bytes<CIN_, PLANE_>();                                                           \
    const int slots = ncu * convt_rows_wgs(CIN_, COUT_, PLANE_);   /* persistent workgroups: this kernel's share of a CU */ \
    if (tb == 0) {                                                                                                    \
      qa.nx = (int)(ta < slots ? ta : slots);                                                                         \
    } else {                                                                                                          \
      long na = (slots * ta + (ta + tb) / 2) / (ta + tb);                                                             \
      na = na < 1 ? 1 : (na > slots - 1 ? slots - 1 : na);                                                            \
      qa.nx = (int)(na < ta ? na : ta);                                                                               \
      qb.nx = (int)(slots - na < tb ? slots - na : tb);                                                               \
    }                                                                                                                 \
    const dim3 grid((unsigned)(qa.nx + qb.nx));                                                                       \
    if (!attr_done) {                                                                                                 \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(convt4s2_rows_kernel<CIN_, COUT_, PLANE_, EPI_>),       \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)ql);                                 \
      attr_done = true;                                                                                               \
    }                                                                                                                 \
    set_last_kernel("mtrssm::convt4s2_rows_kernel<" #CIN_ ", " #COUT_ ", " #PLANE_ ", " #EPI_ ">");                      \
    hipLaunchKernelGGL((convt4s2_rows_kernel<CIN_, COUT_, PLANE_, EPI_>), grid, dim3(64 * ((PLANE_ + 63) / 64) * (COUT_ / 8)), ql, stream, qa, qb); \
    return launched("convt_quad(rows)");                                                                              \
  }
  if (rows_mask & (1u << key)) {
    if (key == 1) MTRSSM_ROWS_LAUNCH(64, 32, 64, false)
    if (key == 2) MTRSSM_ROWS_LAUNCH(32, 16, 256, false)
    if (key == 3) MTRSSM_ROWS_LAUNCH(32, 16, 64, true)
    if (key == 4) MTRSSM_ROWS_LAUNCH(16, 8, 256, true)
  }
#undef MTRSSM_ROWS_LAUNCH
#define MTRSSM_QUAD_LAUNCH(CIN_, COUT_, PLANE_, EPI_)                                                                \
  {                                                                                                                   \
    static bool attr_done_dev[64] = {}; bool& attr_done = attr_done_dev[device_slot()];                                                                                    \
    const size_t ql = quad_lds_bytes<CIN_, PLANE_>();                                                                 \
    const int slots = ncu * quad_wgs(CIN_, PLANE_);                                                                         \
    if (tb == 0) {                                                                                                    \
      qa.nx = (int)(ta < slots ? ta : slots);                                                                         \
    } else {                                                                                                          \
      long na = (slots * ta + (ta + tb) / 2) / (ta + tb);                                                             \
      na = na < 1 ? 1 : (na > slots - 1 ? slots - 1 : na);                                                            \
      qa.nx = (int)(na < ta ? na : ta);                                                                               \
      qb.nx = (int)(slots - na < tb ? slots - na : tb);                                                               \
    }                                                                                                                 \
    const dim3 grid((unsigned)(qa.nx + qb.nx));                                                                       \
    if (!attr_done) {                                                                                                 \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(convt_quad_resident_kernel<CIN_, COUT_, PLANE_, EPI_>), \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)ql);                                 \
      attr_done = true;                                                                                               \
    }                                                                                                                 \
    set_last_kernel("mtrssm::convt_quad_resident_kernel<" #CIN_ ", " #COUT_ ", " #PLANE_ ", " #EPI_ ">");                \
    hipLaunchKernelGGL((convt_quad_resident_kernel<CIN_, COUT_, PLANE_, EPI_>), grid, dim3(kResThreads), ql, stream, qa, qb); \
    return launched("convt_quad");                                                                                    \
  }
  if (key == 1) MTRSSM_QUAD_LAUNCH(64, 32, 64, false)
  if (key == 2) MTRSSM_QUAD_LAUNCH(32, 16, 256, false)
  if (key == 4) MTRSSM_QUAD_LAUNCH(16, 8, 256, true)
  MTRSSM_QUAD_LAUNCH(32, 16, 64, true)
#undef MTRSSM_QUAD_LAUNCH
}

int channel_sum_launch(const float* x, int N, int C, int HW, float* out, hipStream_t stream) {
  if (!x || !out || N <= 0 || C <= 0 || HW <= 0) { set_error("channel_sum: bad argument"); return MTRSSM_EINVAL; }
  const long total = (long)N * HW;
  if (total >= (1L << 31)) { set_error("channel_sum: N * HW must be below 2^31"); return MTRSSM_EINVAL; }
  int splits = (int)((total + 16383) / 16384);  // ~2048 blocks over all channels: enough loads in flight to stream from HBM
  if (splits > 2048 / C) splits = 2048 / C;
  if (splits > 256) splits = 256;  // one atomic per workgroup on out[c]: ~25 ns each when they queue up on one word
  if (splits > N) splits = N;
  if (splits < 1) splits = 1;
  { set_last_kernel("mtrssm::channel_sum_kernel"); hipLaunchKernelGGL(channel_sum_kernel, dim3(C, splits), dim3(256), 0, stream, x, N, C, HW, out); }
  return launched("channel_sum");
}

}  // namespace mtrssm
