// Split-bf16 MFMA variants of the patch-staged conv kernels (included by conv.hip).
//
// v_mfma_f32_32x32x2_f32 runs at 1/16 of the bf16 MFMA rate (MI355X_MICROARCH.md "Matrix cores").  An fp32 value
// is the exact sum of three bf16 pieces (8 significant bits each, round-to-nearest): x = p0 + p1 + p2.  The product
// of two such sums keeps the six terms pi*qj with i + j <= 2 (the dropped ones are below 2^-24 of |x||y|), each term
// one v_mfma_f32_32x32x16_bf16 with fp32 accumulation: fp32-grade results at 6/16 of the fp32-MFMA cost, and 8x fewer
// LDS operand reads per FLOP (one ds_read_b128 carries 8 k-values).  SPLIT = 1 is plain bf16 operands.
// Activation pieces are made ONCE per element while staging (after the fused activation), never per tap; weight pieces
// once per layer call by pack_conv_weight_kernel.
//
// LDS images are channel-innermost: [position][16 channels] bf16 = 32-byte rows, so a tap shift moves whole rows and
// every operand read stays 16-byte aligned.  The two 16-byte halves of a row are swapped on odd 8-row groups
// (half ^ ((row >> 3) & 1)): the 16 lanes one ds_read_b128 cycle serves ({0-3,12-15,20-27} ...) then fall on 16
// different 16-byte slots of the 256-byte bank row for any run of consecutive rows.
#pragma once

namespace mtrssm {

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using u16x8 = __attribute__((ext_vector_type(8))) unsigned short;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;

constexpr int kRowB = 32;      // bytes per LDS row: 16 bf16 channels
// taps whose weights are resident in LDS at once: bounded by the register-prefetch budget (2 waves / SIMD need <= 256 VGPRs)
__host__ __device__ constexpr int split_tg(int split) { return split >= 2 ? 5 : 9; }

template <int SPLIT>
__device__ __forceinline__ void split_bf16(float x, unsigned short (&p)[SPLIT]) {
  float r = x;
#pragma unroll
  for (int s = 0; s < SPLIT; ++s) {
    const __bf16 h = (__bf16)r;
    p[s] = __builtin_bit_cast(unsigned short, h);
    r -= (float)h;  // exact in fp32
  }
}

// Branch-free activations for the staging loops (a libm expm1f per element costs more VALU time than the MFMAs it feeds).
// ELU: v_exp_f32 - 1 on the negative side: absolute error < 1.2e-7, the rounding grain of the O(1) activations it joins
// in the dot product.
__device__ __forceinline__ float elu_fast(float x) { return x > 0.f ? x : __expf(x) - 1.f; }
template <int N>
__device__ __forceinline__ void act_inplace(float (&v)[N], int act) {
  if (act == MTRSSM_ACT_ELU) {
#pragma unroll
    for (int u = 0; u < N; ++u) v[u] = elu_fast(v[u]);
  } else if (act == MTRSSM_ACT_RELU) {
#pragma unroll
    for (int u = 0; u < N; ++u) v[u] = v[u] > 0.f ? v[u] : 0.f;
  } else if (act == MTRSSM_ACT_TANH) {
#pragma unroll
    for (int u = 0; u < N; ++u) v[u] = tanhf(v[u]);
  }
}

__device__ __forceinline__ unsigned swz_row(int row, int half) { return (unsigned)row * kRowB + (unsigned)((half ^ ((row >> 3) & 1)) << 4); }

// w[o][i][ky][kx] (any element strides) -> wp fp32 [OPad][taps][IPad] zero padded, and (pieces > 0) the bf16 pieces
// wq [pieces][OPad][taps][IPad] of the same values.
__device__ __forceinline__ void pack_conv_weight_slice(const float* __restrict__ w, int O, int I, int KH, int KW, long so, long si,
                                                       long sh, long sw, int OPad, int IPad, int pieces, float* __restrict__ wp,
                                                       unsigned short* __restrict__ wq, long first, long step, int VH = 0, int VW = 0) {
  // VH x VW (0: all): the view holds only the leading VH x VW taps of the KH x KW grid, the others are zeros (a 3 x 3 kernel
  // seen as the 4 x 4 kernel of a transposed convolution: its parity-class sub-kernels have 2 x 2, 2 x 1, 1 x 2 and 1 x 1 taps)
  if (VH <= 0) VH = KH;
  if (VW <= 0) VW = KW;
  const int taps = KH * KW;
  const long total = (long)OPad * taps * IPad;
  for (long idx = first; idx < total; idx += step) {
    const int i = (int)(idx % IPad);
    const long ot = idx / IPad;
    const int tap = (int)(ot % taps), o = (int)(ot / taps);
    float v = 0.f;
    if (o < O && i < I && tap / KW < VH && tap % KW < VW) v = w[o * so + i * si + (tap / KW) * sh + (tap % KW) * sw];
    wp[idx] = v;
    float r = v;
    for (int s = 0; s < pieces; ++s) {
      const __bf16 h = (__bf16)r;
      wq[(long)s * total + idx] = __builtin_bit_cast(unsigned short, h);
      r -= (float)h;
    }
  }
}

__global__ void pack_conv_weight_kernel(const float* __restrict__ w, int O, int I, int KH, int KW, long so, long si, long sh, long sw,
                                        int OPad, int IPad, int pieces, float* __restrict__ wp, unsigned short* __restrict__ wq) {
  pack_conv_weight_slice(w, O, I, KH, KW, so, si, sh, sw, OPad, IPad, pieces, wp, wq, (long)blockIdx.x * blockDim.x + threadIdx.x,
                         (long)gridDim.x * blockDim.x);
}

// Every conv weight of a train step in ONE launch: blockIdx.y walks a table of MTRSSM_PACK_DESC_WORDS int64 words per weight
// (include/mtrssm.h: mtrssm_pack_conv_weights).
__global__ void pack_conv_weights_kernel(const long* __restrict__ table) {
  const long* d = table + (size_t)blockIdx.y * MTRSSM_PACK_DESC_WORDS;
  pack_conv_weight_slice(reinterpret_cast<const float*>(d[0]), (int)d[3], (int)d[4], (int)d[5], (int)d[6], d[7], d[8], d[9], d[10],
                         (int)d[11], (int)d[12], (int)d[13], reinterpret_cast<float*>(d[1]), reinterpret_cast<unsigned short*>(d[2]),
                         (long)blockIdx.x * blockDim.x + threadIdx.x, (long)gridDim.x * blockDim.x, (int)d[14], (int)d[15]);
}

// One gather problem of a launch.  A launch carries two (the audio and the vision branch run the same layer on different planes):
// the second one's workgroups follow the first one's in the grid, which turns two launches of 2.08 rounds each (3 rounds:
// the last one nearly empty) into one of 4.17 (5 rounds) -- measured 25 % on the 3x3 64->64 layer.  nx = 0: no second problem.
struct GatherProblem {
  MtrssmConvGeom g;
  const float* src;
  const float* src2;
  const unsigned short* wq;
  const float* bias;
  const float* actgrad_in;
  const float* add_in;
  float* out;
  int tg, ngroups, nx;
  // fused residual block (conv3x3_resident_kernel<..., FUSE>): the block's 1x1 weight [C][Cout] and bias (fp32, as the module
  // holds them) and the block's output; NULL elsewhere
  const float* w1;
  const float* b1;
  float* out2;
};

// Epilogue of the gather kernels: out = (acc + bias) * act'(actgrad_in) + add_in for one lane's pixel x 32*NT channels,
// one 32-channel tile (16 values per lane) at a time.  A tile's global loads are all issued first (clamped addresses, no
// branch around a load), then the arithmetic, then the stores: written as "load, wait, use" per element the 3 x 32
// dependent loads cost ~60 us of latency per workgroup.  load(j) can be issued early (tile 0 under the last MFMA chain,
// tile j+1 before tile j is finished).  Offsets are 32-bit (host checks the output has < 2^31 elements).
struct ConvEpilogue {
  unsigned base, plane_o;
  bool pv, full;
  int cb, cout;
  float bv[16], gv[16], av[16];  // the tile in flight

  __device__ __forceinline__ void init(const MtrssmConvGeom& g, long pe, long ptot, int plane_q, int co0, int tco, int kl) {
    pv = pe < ptot;
    const long pc = pv ? pe : ptot - 1;
    const int n = (int)(pc / plane_q);
    const int rem = (int)(pc - (long)n * plane_q);
    const int oy = rem / g.Wq, ox = rem - oy * g.Wq;
    plane_o = (unsigned)(g.Ho * g.Wo);
    base = (unsigned)n * (unsigned)g.Cout * plane_o + (unsigned)((oy * g.OS + g.QY) * g.Wo + (ox * g.OS + g.QX));
    full = co0 + tco <= g.Cout;  // workgroup-uniform: no channel clamping needed
    cb = co0 + 4 * kl;
    cout = g.Cout;
  }
  __device__ __forceinline__ int chan(int j, int r) const {
    const int co = cb + j * 32 + (r & 3) + 8 * (r >> 2);
    return full ? co : (co < cout ? co : cout - 1);
  }
  __device__ __forceinline__ void load(int j, const float* __restrict__ bias, const float* __restrict__ actgrad_in,
                                       const float* __restrict__ add_in) {
#pragma unroll
    for (int r = 0; r < 16; ++r) bv[r] = bias ? bias[chan(j, r)] : 0.f;
    if (actgrad_in) {
#pragma unroll
      for (int r = 0; r < 16; ++r) gv[r] = actgrad_in[base + (unsigned)chan(j, r) * plane_o];
    }
    if (add_in) {
#pragma unroll
      for (int r = 0; r < 16; ++r) av[r] = add_in[base + (unsigned)chan(j, r) * plane_o];
    }
  }
  // consumes the loaded tile into res[] (so that the next tile's load can be issued before the stores)
  __device__ __forceinline__ void combine(const f32x16& acc, int act, bool has_g, bool has_a, float (&res)[16]) {
    if (has_g) {
      if (act == MTRSSM_ACT_ELU) {
#pragma unroll
        for (int r = 0; r < 16; ++r) gv[r] = gv[r] > 0.f ? 1.f : __expf(gv[r]);
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) gv[r] = act_grad_from_in(gv[r], act);
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float v = acc[r] + bv[r];
      if (has_g) v *= gv[r];
      if (has_a) v += av[r];
      asm volatile("" : "+v"(v));  // materialise here: sunk into the store's branch, every store waits for vmcnt(0)
      res[r] = v;
    }
  }
  __device__ __forceinline__ void store(int j, const float (&res)[16], float* __restrict__ out) const {
    if (full && pv) {
#pragma unroll
      for (int r = 0; r < 16; ++r) out[base + (unsigned)chan(j, r) * plane_o] = res[r];
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = cb + j * 32 + (r & 3) + 8 * (r >> 2);
        if (pv && co < cout) out[base + (unsigned)co * plane_o] = res[r];
      }
    }
  }
};

// Workgroup tile = 128 pixels x 32*NT channels, 4 waves (32 pixels each).  Steps = (16-channel chunk) x (tap group of tg
// taps, tg | taps).  A layer is only ~2 tiles per resident workgroup slot, so the kernel time is the per-workgroup
// dependency chain times the number of rounds, not a throughput bound (measured with everything but the skeleton switched
// off: profiles/round1_notes.md); everything below shortens that chain:
//  * no index table / extra barrier: every thread derives its own patch positions' source offsets;
//  * weights (L2-resident) are fetched one step ahead, the patch (HBM) a whole chunk ahead, into registers; they go to LDS
//    between the two barriers of a step, activation + bf16 pieces applied once per element;
//  * the tap loop is software-pipelined (the next tap's fragments are read while this tap's MFMAs run);
//  * the epilogue's loads are issued before the last MFMA chain, in registers the prefetch no longer needs.
template <int NT, int SPLIT, int PIT>  // PIT: patch positions per thread pair = 128 * PIT >= ps_raw
__global__ __launch_bounds__(kConvThreads, 3) void conv_gather_split_kernel(const GatherProblem pa, const GatherProblem pb) {
  const bool second = blockIdx.x >= (unsigned)pa.nx;  // workgroup-uniform
  const GatherProblem P = second ? pb : pa;  // by value: scalar selects into SGPRs (a reference made every g.field a memory load)
  const MtrssmConvGeom g = P.g;
  const float* __restrict__ src = P.src;
  const float* __restrict__ src2 = P.src2;
  const unsigned short* __restrict__ wq = P.wq;
  const float* __restrict__ bias = P.bias;
  const float* __restrict__ actgrad_in = P.actgrad_in;
  const float* __restrict__ add_in = P.add_in;
  float* __restrict__ out = P.out;
  const int tg = P.tg, ngroups = P.ngroups;
  const unsigned bx = second ? blockIdx.x - (unsigned)pa.nx : blockIdx.x;
  constexpr int TCO = 32 * NT;
  constexpr int LOGT = NT == 2 ? 6 : 5;
  constexpr int WP = (split_tg(SPLIT) * SPLIT * TCO * 2 + kConvThreads - 1) / kConvThreads;  // 16-byte weight pieces per thread
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const PatchGeom pg(g, kTP);
  const int img = pg.ps_raw * kRowB;                          // one piece of the patch: [ps_raw][16 ch]
  unsigned char* patch = lds_raw;                             // [SPLIT][ps_raw][32 B]
  unsigned char* w_lds = patch + (size_t)SPLIT * img;         // [tg][SPLIT][TCO][32 B]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int kl = lane >> 5, il = lane & 31;
  const int taps = g.KH * g.KW;
  const int ctot = g.C + g.C2;
  const int plane_s = g.Hs * g.Ws, plane_q = g.Hq * g.Wq;
  const long ptot = (long)g.N * plane_q;
  // XCD-aware tile order: consecutive workgroup ids go round-robin over the 8 XCDs (each with its own L2); give every XCD
  // a contiguous run of tiles so that neighbouring row bands of a frame (which share halo rows) meet in one L2.
  const unsigned nb = (unsigned)P.nx, per_xcd = (nb + 7) / 8;
  unsigned bid = (bx & 7) * per_xcd + (bx >> 3);
  if ((nb & 7) != 0) bid = bx;  // ragged grids keep the plain order (the remap must stay a bijection)
  const long p0 = (long)bid * kTP;
  const int co0 = blockIdx.y * TCO;
  const int n0 = (int)(p0 / plane_q);
  const int r0 = (int)((p0 - (long)n0 * plane_q) / g.Wq);

  // ---- register prefetch state and the hoisted per-thread offsets
  float pv[PIT][8];
  u32x4 wv[WP];
  const int oct = __builtin_amdgcn_readfirstlane(wave & 1);
  const int rlane = (wave >> 1) * 64 + lane;
  const float* src_n0 = src + (size_t)n0 * g.C * plane_s;
  const int npieces = tg * SPLIT * TCO * 2;
  unsigned fo[PIT], f2[PIT];   // element offsets of this thread's patch positions in src (from src_n0) / src2
  bool okp[PIT];
  {
    const int khm = g.TS > 0 ? 0 : g.KH - 1, kwm = g.TS > 0 ? 0 : g.KW - 1;
    const int sy0 = r0 * g.SS + g.OFFY - khm, sx0 = g.OFFX - kwm;
    const int phw = pg.ph * pg.pw;
#pragma unroll
    for (int it = 0; it < PIT; ++it) {
      const int r = it * 128 + rlane;
      const int ip = r / phw, q = r - ip * phw;
      const int pr = q / pg.pw, pcn = q - pr * pg.pw;
      const int sy = sy0 + pr, sx = sx0 + pcn;
      okp[it] = r < pg.ps_raw && n0 + ip < g.N && sy >= 0 && sy < g.Hs && sx >= 0 && sx < g.Ws;
      f2[it] = okp[it] ? (unsigned)(sy * g.Ws + sx) : 0u;
      fo[it] = okp[it] ? (unsigned)ip * (unsigned)(g.C * plane_s) + f2[it] : 0u;
    }
  }
  unsigned wo[WP];  // element offset of weight piece i in wq for tap group 0, chunk 0; q = tid + 256 i over [t][s][row][half]
#pragma unroll
  for (int i = 0; i < WP; ++i) {
    int q = tid + kConvThreads * i;
    q = q < npieces ? q : npieces - 1;
    const int half = q & 1, row = (q >> 1) & (TCO - 1), st = q >> (1 + LOGT);
    const int t = st / SPLIT, s = st - t * SPLIT;
    wo[i] = (unsigned)s * (unsigned)(g.CoutPad * taps * g.Cpad) + (unsigned)(((co0 + row) * taps + t) * g.Cpad + half * 8);
  }

  // Loads are branch-free per lane (clamped addresses, masked at store time): a load inside a divergent branch makes the
  // compiler wait for it at the join, which serialises the prefetch.
  auto load_patch = [&](int c0) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      int c = c0 + oct * 8 + u;
      c = c < ctot ? c : ctot - 1;  // scalar
      if (c < g.C) {
        const float* bp = src_n0 + (size_t)c * plane_s;
#pragma unroll
        for (int it = 0; it < PIT; ++it) pv[it][u] = bp[fo[it]];
      } else {
        const float* bp = src2 + (size_t)(c - g.C) * plane_s;
#pragma unroll
        for (int it = 0; it < PIT; ++it) pv[it][u] = bp[f2[it]];
      }
    }
  };
  auto store_patch = [&](int c0) {
#pragma unroll
    for (int it = 0; it < PIT; ++it) {
      const int r = it * 128 + rlane;
      if (r < pg.ps_raw) {
        u16x8 q[SPLIT];
        float x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) x[u] = (okp[it] && c0 + oct * 8 + u < ctot) ? pv[it][u] : 0.f;
        if (g.pre_act) act_inplace<8>(x, g.act);  // act(0) = 0 for every supported activation
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          unsigned short p[SPLIT];
          split_bf16<SPLIT>(x[u], p);
#pragma unroll
          for (int s = 0; s < SPLIT; ++s) q[s][u] = p[s];
        }
        const unsigned a = swz_row(r, oct);
#pragma unroll
        for (int s = 0; s < SPLIT; ++s) *reinterpret_cast<u16x8*>(patch + (size_t)s * img + a) = q[s];
      }
    }
  };
  auto load_w = [&](int c0, int tap0) {
    const unsigned short* wb = wq + (size_t)tap0 * g.Cpad + c0;  // scalar
#pragma unroll
    for (int i = 0; i < WP; ++i) wv[i] = *reinterpret_cast<const u32x4*>(wb + wo[i]);
  };
  auto store_w = [&]() {
#pragma unroll
    for (int i = 0; i < WP; ++i) {
      const int q = tid + kConvThreads * i;
      const int half = q & 1, row = (q >> 1) & (TCO - 1);
      if (q < npieces) *reinterpret_cast<u32x4*>(w_lds + (size_t)(q >> 1) * kRowB + ((half ^ ((row >> 3) & 1)) << 4)) = wv[i];
    }
  };

  const int nsteps = taps > 0 ? (g.Cpad / kKC) * ngroups : 0;
  if (nsteps > 0) { load_patch(0); load_w(0, 0); }

  int pixpos;  // this lane's pixel -> patch position (B operand: lane il = pixel); wave w owns pixels [32 w, 32 w + 32)
  {
    const int pix = wave * 32 + il;
    const int row = pix / g.Wq, ox = pix - row * g.Wq;
    const int ip = row / pg.rp, lr = row - ip * pg.rp;
    pixpos = ip * pg.ph * pg.pw + lr * g.SS * pg.pw + ox * g.SS;
  }
  f32x16 acc[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  const unsigned wa = swz_row(il, kl);  // (j*32 + il) >> 3 has the parity of il >> 3
  constexpr bool kPipe = SPLIT == 1;  // register budget: three bf16 pieces leave no room for a second fragment set
  ConvEpilogue epi;
  epi.init(g, p0 + wave * 32 + il, ptot, plane_q, co0, TCO, kl);

  // one tap's operand fragments
  struct Frag { bf16x8 b[SPLIT], a[NT][SPLIT]; };
  auto read_frag = [&](Frag& f, int t, int tapoff) {
    const int pp = pixpos + tapoff;
    const unsigned pa = swz_row(pp, kl);
#pragma unroll
    for (int s = 0; s < SPLIT; ++s) f.b[s] = *reinterpret_cast<const bf16x8*>(patch + (size_t)s * img + pa);
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int s = 0; s < SPLIT; ++s)
        f.a[j][s] = *reinterpret_cast<const bf16x8*>(w_lds + ((size_t)(t * SPLIT + s) * TCO + j * 32) * kRowB + wa);
  };
  auto mfma_frag = [&](const Frag& f) {  // smallest terms first (i + j = SPLIT-1 ... 0)
#pragma unroll
    for (int ord = SPLIT - 1; ord >= 0; --ord)
#pragma unroll
      for (int sa = 0; sa <= ord; ++sa)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[j][sa], f.b[ord - sa], acc[j], 0, 0, 0);
  };

  int grp = 0, c0 = 0;
  for (int step = 0; step < nsteps; ++step) {
    const int tap0 = grp * tg;
    if (step > 0) lds_barrier();  // every wave is done with the previous step's LDS images
    store_w();
    {  // next step's weights (L2-resident): in flight under the patch conversion and this step's MFMAs
      int ngrp = grp + 1, nc0 = c0;
      if (ngrp == ngroups) { ngrp = 0; nc0 += kKC; }
      if (step + 1 < nsteps) load_w(nc0, ngrp * tg);
    }
    if (grp == 0) {
      store_patch(c0);
      // the NEXT chunk's patch (HBM): a whole chunk of steps ahead of its conversion, in the same registers
      if (c0 + kKC < g.Cpad) load_patch(c0 + kKC);
    }
    lds_barrier();
    int ty = tap0 / g.KW, tx = tap0 - ty * g.KW;
    auto next_tapoff = [&]() {
      const int o = g.TS > 0 ? ty * pg.pw + tx : (g.KH - 1 - ty) * pg.pw + (g.KW - 1 - tx);
      if (++tx == g.KW) { tx = 0; ++ty; }
      return o;
    };
    if (kPipe) {  // software-pipelined taps: fragments of tap t+1 are read while tap t's MFMAs run
      Frag f0, f1;
      read_frag(f0, 0, next_tapoff());
      int t = 0;
      for (; t + 2 <= tg; t += 2) {
        read_frag(f1, t + 1, next_tapoff());
        mfma_frag(f0);
        if (t + 2 < tg) read_frag(f0, t + 2, next_tapoff());
        mfma_frag(f1);
      }
      if (t < tg) mfma_frag(f0);
    } else {
      for (int t = 0; t < tg; ++t) {
        Frag f;
        read_frag(f, t, next_tapoff());
        mfma_frag(f);
      }
    }
    if (++grp == ngroups) { grp = 0; c0 += kKC; }
  }
  epi.load(0, bias, actgrad_in, add_in);
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    float res[16];
    epi.combine(acc[j], g.act, actgrad_in != nullptr, add_in != nullptr, res);
    if (j + 1 < NT) epi.load(j + 1, bias, actgrad_in, add_in);
    epi.store(j, res, out);
  }
}

// ------------------------------------------------------------------------------------------------
// 1x1 convolutions (the second conv of every residual block, forward and backward-data): no halo, one tap, K = Cin.
// The general kernel walks them in 16-channel chunks -- two barriers and a staging round per 2 * NT * products MFMAs,
// pure dependency-chain latency (55 us for 105 MB of traffic).  Here a step stages 64 channels at once: [128 pixels]
// [64 ch] and [TCO][64 ch] bf16 images, 128-byte rows whose 16-byte slots are XOR-swizzled with (row >> 1) & 7 so that
// the 16 lanes of a ds_read_b128 cycle fall on 16 distinct slots of the 256-byte bank window; one barrier pair per 64
// channels, four back-to-back k-blocks of MFMAs per step.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned swz128(int row, int slot) { return (unsigned)row * 128u + (unsigned)((slot ^ ((row >> 1) & 7)) << 4); }

template <int NT, int SPLIT>
__global__ __launch_bounds__(kConvThreads, 3) void conv1x1_split_kernel(const GatherProblem pa, const GatherProblem pb) {
  const bool second = blockIdx.x >= (unsigned)pa.nx;  // workgroup-uniform
  const GatherProblem P = second ? pb : pa;  // by value: scalar selects into SGPRs (a reference made every g.field a memory load)
  const MtrssmConvGeom g = P.g;
  const float* __restrict__ src = P.src;
  const unsigned short* __restrict__ wq = P.wq;
  const float* __restrict__ bias = P.bias;
  const float* __restrict__ actgrad_in = P.actgrad_in;
  const float* __restrict__ add_in = P.add_in;
  float* __restrict__ out = P.out;
  const unsigned bx = second ? blockIdx.x - (unsigned)pa.nx : blockIdx.x;
  constexpr int TCO = 32 * NT;
  constexpr int KCH = 64;                                   // channels per step
  constexpr int WPC = SPLIT * TCO * (KCH / 8) / kConvThreads;  // 16-byte weight pieces per thread per step
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  unsigned char* patch = lds_raw;                            // [SPLIT][128 px][128 B]
  unsigned char* w_lds = patch + SPLIT * kTP * 128;          // [SPLIT][TCO][128 B]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int kl = lane >> 5, il = lane & 31;
  const int plane = g.Hq * g.Wq;
  const long ptot = (long)g.N * plane;
  const unsigned nb = (unsigned)P.nx, per_xcd = (nb + 7) / 8;
  unsigned bid = (bx & 7) * per_xcd + (bx >> 3);
  if ((nb & 7) != 0) bid = bx;
  const long p0 = (long)bid * kTP;
  const int co0 = blockIdx.y * TCO;

  // staging role: position = tid & 127, channel half = tid >> 7 (32 of the step's 64 channels)
  const int spos = tid & 127, shalf = tid >> 7;
  long sp = p0 + spos;
  const bool sok = sp < ptot;
  sp = sok ? sp : ptot - 1;
  const int sn = (int)(sp / plane), srem = (int)(sp - (long)sn * plane);
  const float* sbase = src + ((size_t)sn * g.C) * plane + srem;

  f32x16 acc[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  ConvEpilogue epi;
  epi.init(g, p0 + wave * 32 + il, ptot, plane, co0, TCO, kl);
  const int prow = wave * 32 + il;  // this lane's pixel row in the patch image
  const size_t wq_plane = (size_t)g.CoutPad * g.Cpad;

  for (int c0 = 0; c0 < g.Cpad; c0 += KCH) {
    // ---- loads first: 32 channels of this thread's pixel, and its weight pieces
    float v[32];
#pragma unroll
    for (int u = 0; u < 32; ++u) {
      int c = c0 + shalf * 32 + u;
      c = c < g.C ? c : g.C - 1;  // wave-uniform clamp; masked below
      v[u] = sbase[(size_t)c * plane];
    }
    u32x4 wv[WPC];
#pragma unroll
    for (int i = 0; i < WPC; ++i) {
      const int q = tid + kConvThreads * i;           // over [s][row][slot]
      const int slot = q & 7, row = (q >> 3) & (TCO - 1), s = q / (8 * TCO);
      wv[i] = *reinterpret_cast<const u32x4*>(wq + s * wq_plane + (size_t)(co0 + row) * g.Cpad + c0 + slot * 8);
    }
    if (c0 > 0) lds_barrier();  // everyone is done with the previous step's images
#pragma unroll
    for (int o = 0; o < 4; ++o) {
      float x[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) x[u] = (sok && c0 + shalf * 32 + o * 8 + u < g.C) ? v[o * 8 + u] : 0.f;
      if (g.pre_act) act_inplace<8>(x, g.act);
      u16x8 qv[SPLIT];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        unsigned short pc[SPLIT];
        split_bf16<SPLIT>(x[u], pc);
#pragma unroll
        for (int s = 0; s < SPLIT; ++s) qv[s][u] = pc[s];
      }
      const unsigned ad = swz128(spos, shalf * 4 + o);
#pragma unroll
      for (int s = 0; s < SPLIT; ++s) *reinterpret_cast<u16x8*>(patch + s * kTP * 128 + ad) = qv[s];
    }
#pragma unroll
    for (int i = 0; i < WPC; ++i) {
      const int q = tid + kConvThreads * i;
      const int slot = q & 7, row = (q >> 3) & (TCO - 1), s = q / (8 * TCO);
      *reinterpret_cast<u32x4*>(w_lds + s * TCO * 128 + swz128(row, slot)) = wv[i];
    }
    lds_barrier();
    // ---- four k-blocks of 16 channels
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      bf16x8 b[SPLIT], a[NT][SPLIT];
#pragma unroll
      for (int s = 0; s < SPLIT; ++s) b[s] = *reinterpret_cast<const bf16x8*>(patch + s * kTP * 128 + swz128(prow, kb * 2 + kl));
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int s = 0; s < SPLIT; ++s)
          a[j][s] = *reinterpret_cast<const bf16x8*>(w_lds + s * TCO * 128 + swz128(j * 32 + il, kb * 2 + kl));
#pragma unroll
      for (int ord = SPLIT - 1; ord >= 0; --ord)
#pragma unroll
        for (int sa = 0; sa <= ord; ++sa)
#pragma unroll
          for (int j = 0; j < NT; ++j)
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[j][sa], b[ord - sa], acc[j], 0, 0, 0);
    }
  }
  epi.load(0, bias, actgrad_in, add_in);
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    float res[16];
    epi.combine(acc[j], g.act, actgrad_in != nullptr, add_in != nullptr, res);
    if (j + 1 < NT) epi.load(j + 1, bias, actgrad_in, add_in);
    epi.store(j, res, out);
  }
}

// ------------------------------------------------------------------------------------------------
// Weight gradient with split-bf16 operands: dWp[co][tap][c] += sum_pix A[co][pix] P[c][pos(pix) + tap].
// Persistent, one 8-wave workgroup per CU looping over 64-pixel groups (as conv_weight_grad_patch_kernel): waves 4-7
// stage group g+1 (global fp32 -> activation -> bf16 pieces -> LDS) while waves 0-3 run group g's MFMAs, one barrier per
// group.  k = pixels: the A operand is a pixel-innermost image [co][64 px] read with ds_read_b128; the B operand comes
// from the SAME channel-innermost patch image the gather kernel uses, [position][channels], through
// ds_read_b64_tr_b16 -- the hardware transpose delivers, for a 16-lane group, 4 positions x 16 channels column-major, so
// a lane gets 4 consecutive k of its channel and a tap is just a row offset (aligned for every tap).  An MFMA's 32
// columns are two independent 16-channel halves (lane groups {0,2} and {1,3}) that may belong to different taps: column
// tiles are cut from the flattened (tap, 16-channel block) list.
// Rows of the patch image are Cp2 * 2 bytes (Cp2 = channels rounded up to a power of two >= 16); 64-byte segments of
// 128- and 256-byte rows are XOR-swizzled with the position so that the 4 rows x 2 halves a 32-lane half reads per
// cycle fall on distinct banks.
// ------------------------------------------------------------------------------------------------
using s16x4 = __attribute__((ext_vector_type(4))) short;
constexpr int kWgLdaB = kGP * 2 + 16;  // bytes per A-image row: 64 px bf16 + 16 (odd multiple of 16: conflict-free b128)

__device__ __forceinline__ unsigned wg_swz(unsigned pos, int rowb) {
  return rowb == 128 ? ((pos >> 1) & 1u) << 6 : (rowb == 256 ? (pos & 3u) << 6 : 0u);
}

template <int NT, int SPLIT>
__global__ __launch_bounds__(2 * kConvThreads, 2) void conv_weight_grad_split_kernel(
    const MtrssmConvGeom g, const float* __restrict__ a, const float* __restrict__ src, const float* __restrict__ src2,
    const int pre_act_a, float* __restrict__ dwp, float* __restrict__ dbias, const int cp2, const int nbuf) {
  constexpr int TCO = 32 * NT;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const PatchGeom pg(g, kGP);
  const int rowb = cp2 * 2;
  const int a_bytes = TCO * kWgLdaB;        // one piece of the A image
  const int p_bytes = pg.ps_raw * rowb;     // one piece of the patch image
  const int buf_bytes = SPLIT * (a_bytes + p_bytes);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool producer = wave >= 4;
  const int lw = wave & 3, ptid = tid & (kConvThreads - 1);
  const int kl = lane >> 5, il = lane & 31;
  const int taps = g.KH * g.KW;
  const int ctot = g.C + g.C2;
  const int nc16 = cp2 >> 4;
  const int nhalf = taps * nc16;            // 16-column half tiles
  const int co0 = blockIdx.y * TCO;
  const int plane_s = g.Hs * g.Ws, plane_a = g.Hq * g.Wq;
  const long ptot = (long)g.N * plane_a;
  const long groups = (ptot + kGP - 1) / kGP;
  const long gper = (groups + gridDim.x - 1) / gridDim.x;
  const long gbeg = (long)blockIdx.x * gper;
  const long gend = gbeg + gper < groups ? gbeg + gper : groups;
  if (gbeg >= gend) return;  // workgroup-uniform

  // group-invariant decode of the patch positions: (frame-in-group << 20) | (patch row << 10) | patch column
  int* ptab = reinterpret_cast<int*>(lds_raw + (size_t)nbuf * buf_bytes);  // [nblk * 64]
  const int nblk = (pg.ps_raw + 63) >> 6;
  {
    const int phw = pg.ph * pg.pw;
    for (int pos = tid; pos < nblk * 64; pos += 2 * kConvThreads) {
      const int ip = pos / phw, q = pos - ip * phw;
      const int pr = q / pg.pw, pcn = q - pr * pg.pw;
      ptab[pos] = pos < pg.ps_raw ? (ip << 20) | (pr << 10) | pcn : -1;
    }
  }
  __syncthreads();

  if (producer) {
    const int noct = cp2 >> 3;
    const int ntask = nblk * noct;  // wave-tasks: 64 positions x 8 channels
    constexpr int RB = 4;           // wave-tasks whose loads are in flight together
    float bsum[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) bsum[i] = 0.f;
    const int igroups = (int)groups, iptot = (int)ptot;  // host checks N * Hq * Wq < 2^31

    // one group's staging data in registers
    struct Regs {
      float4 av[NT][2];
      float pv[RB][8];
      int posr[RB];  // pos | oct << 16 | ok << 31
    };
    auto load_a = [&](Regs& R, int grp) {  // A tile: 2 x 16 B per task
      const int p0 = grp * kGP;
#pragma unroll
      for (int i = 0; i < NT; ++i) {
        const int idx = ptid + kConvThreads * i;
        const int row = idx >> 3, o8 = idx & 7;
        int p = p0 + 8 * o8;
        p = p < iptot ? p : iptot - 8;  // plane_a % 8 == 0 (host-checked): an octet never leaves its frame
        const int n = p / plane_a;
        const int rem = p - n * plane_a;
        const int co = co0 + row < g.Cout ? co0 + row : g.Cout - 1;
        const float4* ap = reinterpret_cast<const float4*>(a + ((size_t)n * g.Cout + co) * plane_a + rem);
        R.av[i][0] = ap[0];
        R.av[i][1] = ap[1];
      }
    };
    auto load_p = [&](Regs& R, int grp, int wb) {  // RB wave-tasks of the patch starting at wave-task wb
      const int p0 = grp * kGP;
      const int n0 = p0 / plane_a;
      const int r0 = (p0 - n0 * plane_a) / g.Wq;
      const int sy0 = r0 * g.SS + g.OFFY, sx0 = g.OFFX;
      const float* src_n0 = src + (size_t)n0 * g.C * plane_s;
#pragma unroll
      for (int i = 0; i < RB; ++i) {
        int wt = wb + lw + 4 * i;
        wt = __builtin_amdgcn_readfirstlane(wt < ntask ? wt : ntask - 1);  // clamped: surplus tasks redo the last one
        const int oct = wt / nblk, pb = wt - oct * nblk;
        const int pos = pb * 64 + lane;
        const int d = ptab[pos];
        const int ip = d >> 20, pr = (d >> 10) & 1023, pcn = d & 1023;
        const int sy = sy0 + pr, sx = sx0 + pcn;
        const bool ok = d >= 0 && n0 + ip < g.N && sy >= 0 && sy < g.Hs && sx >= 0 && sx < g.Ws;
        R.posr[i] = pos | (oct << 16) | (ok ? (int)0x80000000 : 0);
        const unsigned f2 = ok ? (unsigned)(sy * g.Ws + sx) : 0u;
        const unsigned fo = ok ? (unsigned)ip * (unsigned)(g.C * plane_s) + f2 : 0u;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          int c = oct * 8 + u;
          c = c < ctot ? c : ctot - 1;  // scalar
          const bool own = c < g.C;
          const float* bp = own ? src_n0 + (size_t)c * plane_s : src2 + (size_t)(c - g.C) * plane_s;
          R.pv[i][u] = bp[own ? fo : f2];
        }
      }
    };
    auto conv_p = [&](const Regs& R, unsigned char* P) {
#pragma unroll
      for (int i = 0; i < RB; ++i) {
        const int pos = R.posr[i] & 0xffff, oct = (R.posr[i] >> 16) & 0x7fff;
        const bool ok = R.posr[i] < 0;
        float x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) x[u] = (ok && oct * 8 + u < ctot) ? R.pv[i][u] : 0.f;
        if (g.pre_act) act_inplace<8>(x, g.act);
        u16x8 q[SPLIT];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          unsigned short pc[SPLIT];
          split_bf16<SPLIT>(x[u], pc);
#pragma unroll
          for (int s = 0; s < SPLIT; ++s) q[s][u] = pc[s];
        }
        const unsigned ad = (unsigned)pos * rowb + (((unsigned)oct * 16) ^ wg_swz(pos, rowb));
        if (pos < pg.ps_raw) {
#pragma unroll
          for (int s = 0; s < SPLIT; ++s) *reinterpret_cast<u16x8*>(P + (size_t)s * p_bytes + ad) = q[s];
        }
      }
    };
    auto conv_a = [&](const Regs& R, int grp, unsigned char* A) {
      const int p0 = grp * kGP;
#pragma unroll
      for (int i = 0; i < NT; ++i) {
        const int idx = ptid + kConvThreads * i;
        const int row = idx >> 3, o8 = idx & 7;
        const bool ok = p0 + 8 * o8 < iptot && co0 + row < g.Cout;
        float x[8] = {R.av[i][0].x, R.av[i][0].y, R.av[i][0].z, R.av[i][0].w, R.av[i][1].x, R.av[i][1].y, R.av[i][1].z, R.av[i][1].w};
#pragma unroll
        for (int u = 0; u < 8; ++u) { x[u] = ok ? x[u] : 0.f; bsum[i] += x[u]; }
        if (pre_act_a) act_inplace<8>(x, g.act);
        u16x8 q[SPLIT];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          unsigned short pc[SPLIT];
          split_bf16<SPLIT>(x[u], pc);
#pragma unroll
          for (int s = 0; s < SPLIT; ++s) q[s][u] = pc[s];
        }
#pragma unroll
        for (int s = 0; s < SPLIT; ++s) *reinterpret_cast<u16x8*>(A + (size_t)s * a_bytes + row * kWgLdaB + o8 * 16) = q[s];
      }
    };

    const int ibeg = (int)gbeg, iend = (int)gend;
    (void)igroups;
    if (ntask <= 4 * RB) {
      // One round per group: the loads of group k+2 are issued BEFORE group k+1 is converted, so a full group of MFMA
      // time hides the HBM latency.  Loads are unconditional (group index clamped) so that the wait before the
      // conversion is a counted vmcnt, not vmcnt(0).
      Regs cur, nxt;
      load_a(nxt, ibeg);
      load_p(nxt, ibeg, 0);
#pragma unroll 1
      for (int grp = ibeg - 1; grp < iend; ++grp) {  // iteration ibeg-1 only stages group ibeg
        if (grp + 1 < iend) {
          cur = nxt;
          const int g2 = grp + 2 < iend ? grp + 2 : iend - 1;
          load_a(nxt, g2);
          load_p(nxt, g2, 0);
          unsigned char* A = lds_raw + (size_t)((grp + 1 - ibeg) & 1) * buf_bytes;
          conv_p(cur, A + (size_t)SPLIT * a_bytes);
          conv_a(cur, grp + 1, A);
        }
        lds_barrier();
      }
    } else {
      // nbuf == 1 (images too big to double-buffer: the k=4 s=2 decoder layers in three pieces): stage, barrier, the
      // consumers compute, barrier -- no overlap, still faster than the fp32 kernel.
#pragma unroll 1
      for (int grp = ibeg - 1; grp < iend; ++grp) {
        if (grp + 1 < iend) {
          unsigned char* A = lds_raw + (size_t)(nbuf == 2 ? (grp + 1 - ibeg) & 1 : 0) * buf_bytes;
          Regs R;
          load_a(R, grp + 1);
#pragma unroll 1
          for (int wb = 0; wb < ntask; wb += 4 * RB) {
            load_p(R, grp + 1, wb);
            conv_p(R, A + (size_t)SPLIT * a_bytes);
          }
          conv_a(R, grp + 1, A);  // its loads have landed under the patch work
        }
        lds_barrier();
        if (nbuf == 1 && grp + 1 < iend) lds_barrier();  // the consumers are done with the only buffer
      }
    }
    if (dbias != nullptr) {  // thread (row, o8): the 8 octet lanes of a row are consecutive lanes
#pragma unroll
      for (int i = 0; i < NT; ++i) {
        float t = bsum[i];
        t += __shfl_xor(t, 1, kWave);
        t += __shfl_xor(t, 2, kWave);
        t += __shfl_xor(t, 4, kWave);
        const int row = (ptid + kConvThreads * i) >> 3;
        if ((ptid & 7) == 0 && co0 + row < g.Cout) atomicAdd(&dbias[co0 + row], t);
      }
    }
    return;
  }

  // ---- consumers: this wave's column tiles q = lw + 4 s; lane group G = lane >> 4 serves half h = G & 1, k-rows 8 (G >> 1) ..
  const int G = lane >> 4, h = G & 1, qrow = (lane & 15) >> 2, pq = lane & 3;
  unsigned posq[4];  // packed (16 bits each): patch positions of pixels 16 kb + 8 (G >> 1) + 4 rd + qrow, rd = 0, 1
#pragma unroll
  for (int kb = 0; kb < 4; ++kb) {
    unsigned pk = 0;
#pragma unroll
    for (int rd = 0; rd < 2; ++rd) {
      const int pix = 16 * kb + 8 * (G >> 1) + 4 * rd + qrow;
      const int row = pix / g.Wq, ox = pix - row * g.Wq;
      const int ip = row / pg.rp, lr = row - ip * pg.rp;
      pk |= (unsigned)(ip * pg.ph * pg.pw + lr * g.SS * pg.pw + ox * g.SS) << (16 * rd);
    }
    posq[kb] = pk;
  }
  unsigned tcb[kMaxQ];  // packed per column tile: tap offset in positions << 16 | channel byte offset of this lane's 4 columns
  int nsl = 0;
#pragma unroll
  for (int s = 0; s < kMaxQ; ++s) {
    const int q = lw + 4 * s;
    tcb[s] = (unsigned)(pq * 8);
    if (2 * q < nhalf) {
      nsl = s + 1;
      const int hf = 2 * q + h;
      if (hf < nhalf) {
        const int tap = hf / nc16, c16 = hf - tap * nc16;
        const int ty = tap / g.KW, tx = tap - ty * g.KW;
        tcb[s] = ((unsigned)(ty * pg.pw + tx) << 16) | (unsigned)(c16 * 32 + pq * 8);
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

  lds_barrier();  // first group staged
  int cur = 0;
  for (long grp = gbeg; grp < gend; ++grp) {
    const unsigned char* A = lds_raw + (size_t)cur * buf_bytes;
    const unsigned char* P = A + (size_t)SPLIT * a_bytes;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      bf16x8 af[NT][SPLIT];
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int s = 0; s < SPLIT; ++s)
          af[j][s] = *reinterpret_cast<const bf16x8*>(A + (size_t)s * a_bytes + (j * 32 + il) * kWgLdaB + kb * 32 + kl * 16);
#pragma unroll
      for (int sl = 0; sl < kMaxQ; ++sl) {
        if (sl < nsl) {  // wave-uniform: EXEC stays all ones for the transposed reads
          unsigned pk = posq[kb];
          asm volatile("" : "+v"(pk));  // opaque per iteration: otherwise all 40 read addresses are hoisted out of the group loop and spill
          const unsigned to = tcb[sl] >> 16, cb = tcb[sl] & 0xffffu;
          const unsigned p0 = (pk & 0xffffu) + to, p1 = (pk >> 16) + to;
          const unsigned a0 = p0 * rowb + (cb ^ wg_swz(p0, rowb));
          const unsigned a1 = p1 * rowb + (cb ^ wg_swz(p1, rowb));
          bf16x8 bfr[SPLIT];
#pragma unroll
          for (int s = 0; s < SPLIT; ++s) {
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(P + (size_t)s * p_bytes + a0));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(P + (size_t)s * p_bytes + a1));
            u16x8 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] = (unsigned short)lo[e]; v[4 + e] = (unsigned short)hi[e]; }
            bfr[s] = __builtin_bit_cast(bf16x8, v);
          }
#pragma unroll
          for (int ord = SPLIT - 1; ord >= 0; --ord)
#pragma unroll
            for (int sa = 0; sa <= ord; ++sa)
#pragma unroll
              for (int j = 0; j < NT; ++j)
                acc[sl][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[j][sa], bfr[ord - sa], acc[sl][j], 0, 0, 0);
        }
      }
    }
    lds_barrier();
    if (nbuf == 2) cur ^= 1;
    else if (grp + 1 < gend) lds_barrier();  // single buffer: wait for the next group to be staged
  }

#pragma unroll
  for (int s = 0; s < kMaxQ; ++s) {
    const int hf = 2 * (lw + 4 * s) + (il >> 4);  // this lane's COLUMN half (il), not its staging half
    if (s < nsl && hf < nhalf) {
      const int tap = hf / nc16, c16 = hf - tap * nc16;
      const int c = c16 * 16 + (il & 15);
      if (c < ctot) {
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
// Weight gradient of the 1x1 layers: dW[co][ci] = sum_{n,px} preA(a[n,co,px]) pre(src[n,ci,px]) is a plain GEMM whose
// k index (pixels) is the contiguous one in BOTH operands, which is exactly the MFMA register layout: lane (row = lane & 31,
// k-group = lane >> 5) holds 8 consecutive pixels of one channel plane = two 16-byte global loads, converted to bf16 pieces
// in registers.  No LDS, no staging waves, no barriers.  One wave per 32x32 output tile (up to 4 x 4 per workgroup), every
// workgroup a slice of the pixel stream, one prefetched k-step (16 pixels) in flight per wave; partial tiles by fp32 atomics.
// Rows / columns beyond Cout / C read a clamped plane and are never stored.
// ------------------------------------------------------------------------------------------------
template <int SPLIT>
__global__ __launch_bounds__(1024) void conv1x1_weight_grad_split_kernel(
    const MtrssmConvGeom g, const float* __restrict__ a, const float* __restrict__ src, const int pre_act_a, float* __restrict__ dwp,
    float* __restrict__ dbias, const int tiles_ci, const int steps_per_wg) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int il = lane & 31, kl = lane >> 5;
  const int tco = wave / tiles_ci, tci = wave - tco * tiles_ci;
  const int plane = g.Hq * g.Wq, spp = plane >> 4;  // 16-pixel k-steps per frame (host: plane % 16 == 0)
  const int total = g.N * spp;
  const int s0 = blockIdx.x * steps_per_wg;
  const int s1 = s0 + steps_per_wg < total ? s0 + steps_per_wg : total;
  if (s0 >= s1) return;
  const int co = tco * 32 + il, ci = tci * 32 + il;
  const float* ap = a + (size_t)(co < g.Cout ? co : g.Cout - 1) * plane + 8 * kl;
  const float* bp = src + (size_t)(ci < g.C ? ci : g.C - 1) * plane + 8 * kl;
  const size_t fa = (size_t)g.Cout * plane, fb = (size_t)g.C * plane;

  struct Step { float4 a0, a1, b0, b1; };
  int n = s0 / spp, q = s0 - n * spp;  // frame and k-step within it of the NEXT load
  auto load = [&](Step& t) {
    const float4* pa = reinterpret_cast<const float4*>(ap + (size_t)n * fa + (q << 4));
    const float4* pb = reinterpret_cast<const float4*>(bp + (size_t)n * fb + (q << 4));
    t.a0 = pa[0]; t.a1 = pa[1]; t.b0 = pb[0]; t.b1 = pb[1];
  };
  auto advance = [&](int s) {  // (n, q) from step s to step s + 1, clamped to the last one (loads stay unconditional)
    if (s + 1 < s1 && ++q == spp) { q = 0; ++n; }
  };
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float bsum = 0.f;
  constexpr int D = 4;  // k-steps in flight per wave (4 x 16 B per lane each): one wave per SIMD has to cover the HBM latency alone
  Step buf[D];
  int ls = s0;  // step of the next load
#pragma unroll
  for (int d = 0; d < D; ++d) { load(buf[d]); advance(ls); ++ls; }
#pragma unroll 1
  for (int s = s0; s < s1; s += D) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const Step cur = buf[d];
      load(buf[d]);
      advance(ls);
      ++ls;
      if (s + d < s1) {  // wave-uniform
        float xa[8] = {cur.a0.x, cur.a0.y, cur.a0.z, cur.a0.w, cur.a1.x, cur.a1.y, cur.a1.z, cur.a1.w};
        float xb[8] = {cur.b0.x, cur.b0.y, cur.b0.z, cur.b0.w, cur.b1.x, cur.b1.y, cur.b1.z, cur.b1.w};
#pragma unroll
        for (int u = 0; u < 8; ++u) bsum += xa[u];
        if (pre_act_a) act_inplace<8>(xa, g.act);
        if (g.pre_act) act_inplace<8>(xb, g.act);
        u16x8 qa[SPLIT], qb[SPLIT];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          unsigned short pa[SPLIT], pb[SPLIT];
          split_bf16<SPLIT>(xa[u], pa);
          split_bf16<SPLIT>(xb[u], pb);
#pragma unroll
          for (int p = 0; p < SPLIT; ++p) { qa[p][u] = pa[p]; qb[p][u] = pb[p]; }
        }
#pragma unroll
        for (int ord = SPLIT - 1; ord >= 0; --ord)
#pragma unroll
          for (int sa = 0; sa <= ord; ++sa)
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, qa[sa]), __builtin_bit_cast(bf16x8, qb[ord - sa]), acc, 0, 0, 0);
      }
    }
  }
  if (ci < g.C) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = tco * 32 + (r & 3) + 8 * (r >> 2) + 4 * kl;
      if (row < g.Cout) atomicAdd(&dwp[(size_t)row * g.Cpad + ci], acc[r]);
    }
  }
  if (dbias != nullptr && tci == 0) {  // wave-uniform
    bsum += __shfl_xor(bsum, 32, kWave);
    if (kl == 0 && co < g.Cout) atomicAdd(&dbias[co], bsum);
  }
}

// ------------------------------------------------------------------------------------------------
// Weight gradient of the 3x3 / stride 1 / pad 1 layers on 8- or 4-pixel-wide planes (the residual stacks: 8x8 vision,
// 16x4 audio), the same register-direct scheme: a lane's 8 pixels are PXR = 8 / W whole plane rows of its channel, and every
// tap's shifted copy of them lies in a window of PXR + 2 rows that the lane loads itself (16-byte loads), activates,
// zeroes outside the frame and splits ONCE; the nine B fragments are then static selections of window elements.  One
// wave per (32 co) x (32 ci) x 9 taps (nine accumulator tiles), up to 2 x 2 waves per workgroup, blockIdx.y walks
// 64-channel groups of Cout; partial tiles by fp32 atomics in the dwp layout [co][tap][Cpad].
// ------------------------------------------------------------------------------------------------
template <int SPLIT, int W>
__global__ __launch_bounds__(kConvThreads, 1) void conv3x3_weight_grad_split_kernel(
    const MtrssmConvGeom g, const float* __restrict__ a, const float* __restrict__ src, const int pre_act_a, float* __restrict__ dwp,
    float* __restrict__ dbias, const int tiles_ci, const int steps_per_wg) {
  constexpr int PXR = 8 / W, R = PXR + 2, NV = R * W, RV = W / 4;  // window rows, values, 16-byte loads per row
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int il = lane & 31, kl = lane >> 5;
  // (two k-halves of the slice on two waves per SIMD, met in LDS, were measured slower: 135 vs 116 us -- 256 registers per
  // wave spill next to 144 accumulators)
  const int tco = wave / tiles_ci, tci = wave - tco * tiles_ci;
  const int H = g.Hq, plane = H * W, spp = plane >> 4;  // host: Wq == W, plane % 16 == 0
  const int total = g.N * spp;
  const int s0 = blockIdx.x * steps_per_wg;
  const int s1 = s0 + steps_per_wg < total ? s0 + steps_per_wg : total;
  if (s0 >= s1) return;
  const int co = blockIdx.y * 64 + tco * 32 + il, ci = tci * 32 + il;
  const float* ap = a + (size_t)(co < g.Cout ? co : g.Cout - 1) * plane + 8 * kl;
  const float* bp = src + (size_t)(ci < g.C ? ci : g.C - 1) * plane;
  const size_t fa = (size_t)g.Cout * plane, fb = (size_t)g.C * plane;

  struct Step { float4 a0, a1; float4 w[R * RV]; int y0; };
  int n = s0 / spp, q = s0 - n * spp;  // frame and k-step within it of the NEXT load
  auto load = [&](Step& t) {
    const float4* pa = reinterpret_cast<const float4*>(ap + (size_t)n * fa + (q << 4));
    t.a0 = pa[0]; t.a1 = pa[1];
    t.y0 = ((q << 4) + 8 * kl) / W;  // first plane row of this lane's pixels
    const float* fr = bp + (size_t)n * fb;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      int y = t.y0 - 1 + r;
      y = y < 0 ? 0 : (y >= H ? H - 1 : y);  // clamped; zeroed after the activation
      const float4* pr = reinterpret_cast<const float4*>(fr + y * W);
#pragma unroll
      for (int v = 0; v < RV; ++v) t.w[r * RV + v] = pr[v];
    }
  };
  auto advance = [&](int s) {  // (n, q) from step s to step s + 1, clamped to the last one (loads stay unconditional)
    if (s + 1 < s1 && ++q == spp) { q = 0; ++n; }
  };
  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float bsum = 0.f;
  Step cur, nxt;  // (two k-steps in flight instead of one: no change, 122 vs 120 us -- not bound by the load latency)
  load(cur);
  advance(s0);
#pragma unroll 1
  for (int s = s0; s < s1; ++s) {
    load(nxt);
    advance(s + 1);
    float xa[8] = {cur.a0.x, cur.a0.y, cur.a0.z, cur.a0.w, cur.a1.x, cur.a1.y, cur.a1.z, cur.a1.w};
#pragma unroll
    for (int u = 0; u < 8; ++u) bsum += xa[u];
    if (pre_act_a) act_inplace<8>(xa, g.act);
    u16x8 qa[SPLIT];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      unsigned short pa[SPLIT];
      split_bf16<SPLIT>(xa[u], pa);
#pragma unroll
      for (int p = 0; p < SPLIT; ++p) qa[p][u] = pa[p];
    }
    // the whole window is activated, zeroed outside the frame and split once; the nine fragments are static selections
    // (converting row by row inside the tap loop measured slower: 128 vs 116 us)
    float wv[NV];
#pragma unroll
    for (int i = 0; i < R * RV; ++i) { wv[4 * i] = cur.w[i].x; wv[4 * i + 1] = cur.w[i].y; wv[4 * i + 2] = cur.w[i].z; wv[4 * i + 3] = cur.w[i].w; }
    if (g.pre_act) act_inplace<NV>(wv, g.act);
    unsigned short pb[SPLIT][NV];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int y = cur.y0 - 1 + r;
      const bool in = y >= 0 && y < H;
#pragma unroll
      for (int x = 0; x < W; ++x) {
        unsigned short pc[SPLIT];
        split_bf16<SPLIT>(in ? wv[r * W + x] : 0.f, pc);
#pragma unroll
        for (int p = 0; p < SPLIT; ++p) pb[p][r * W + x] = pc[p];
      }
    }
    u16x8 qb[9][SPLIT];
#pragma unroll
    for (int ty = 0; ty < 3; ++ty)
#pragma unroll
      for (int tx = 0; tx < 3; ++tx)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int r = e / W + ty, x = e % W + tx - 1;  // window row / column of pixel e under this tap
#pragma unroll
          for (int p = 0; p < SPLIT; ++p)
            qb[ty * 3 + tx][p][e] = (x < 0 || x >= W) ? (unsigned short)0 : pb[p][r * W + (x < 0 ? 0 : (x >= W ? W - 1 : x))];
        }
    // product-major: consecutive MFMAs go to different accumulator tiles, so none waits for its predecessor
#pragma unroll
    for (int ord = SPLIT - 1; ord >= 0; --ord)
#pragma unroll
      for (int sa = 0; sa <= ord; ++sa)
#pragma unroll
        for (int t = 0; t < 9; ++t)
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, qa[sa]), __builtin_bit_cast(bf16x8, qb[t][ord - sa]), acc[t], 0, 0, 0);
    cur = nxt;
  }
  if (ci < g.C) {
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = blockIdx.y * 64 + tco * 32 + (r & 3) + 8 * (r >> 2) + 4 * kl;
        if (row < g.Cout) atomicAdd(&dwp[((size_t)row * 9 + t) * g.Cpad + ci], acc[t][r]);
      }
  }
  if (dbias != nullptr && tci == 0) {  // wave-uniform
    bsum += __shfl_xor(bsum, 32, kWave);
    if (kl == 0 && co < g.Cout) atomicAdd(&dbias[co], bsum);
  }
}

// ------------------------------------------------------------------------------------------------
// Weight gradient of the thin strided layers (first encoder conv 3 -> 8, last decoder deconv 16 -> 1: at most 32 output
// channels and at most 32 (tap, channel) columns, big planes): ONE 32x32 MFMA tile.  The A operand (`a`, contiguous in the
// summation index) goes straight from HBM into the MFMA layout as in the 1x1 kernel; lane `col` of the B operand gathers
// its own tap / channel: 8 pixels of one output row = 8 loads SS floats apart from one source row (neighbouring lanes hit
// the same lines).  Every wave owns a slice of the pixel stream; the 16 waves of a workgroup meet in one LDS tile first.
// ------------------------------------------------------------------------------------------------
template <int SPLIT>
__global__ __launch_bounds__(1024) void conv_weight_grad_thin_split_kernel(
    const MtrssmConvGeom g, const float* __restrict__ a, const float* __restrict__ src, const float* __restrict__ src2,
    const int pre_act_a, float* __restrict__ dwp, float* __restrict__ dbias, const int steps_per_wave, const int log2_wq) {
  const int lane = threadIdx.x & 63;
  const int il = lane & 31, kl = lane >> 5;
  const int plane_a = g.Hq * g.Wq, plane_s = g.Hs * g.Ws;
  const int total = (int)(((long)g.N * plane_a) >> 4);  // 16-pixel k-steps (host: plane_a % 16 == 0, Wq % 8 == 0, Wq = 2^log2_wq)
  const int gw = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  const int s0 = gw * steps_per_wave;
  const int s1 = s0 + steps_per_wave < total ? s0 + steps_per_wave : total;
  const bool active = s0 < s1;  // wave-uniform; idle waves still meet the barriers below
  // every wave's partial tile goes through ONE tile in LDS (ds_add_f32) so that a workgroup of 16 waves sends one set of
  // global atomics: they all land on the same <= 1024 addresses, and 4096 sets of them cost more than the whole pixel stream
  __shared__ float red[16 * 64 + 32];
  for (int i = threadIdx.x; i < 16 * 64 + 32; i += blockDim.x) red[i] = 0.f;
  __syncthreads();
  const int ctot = g.C + g.C2, taps = g.KH * g.KW, ncols = taps * ctot;
  const int row = il < g.Cout ? il : g.Cout - 1;
  const bool col_ok = il < ncols;
  const int col = col_ok ? il : 0;
  const int tap = col / ctot, c = col - tap * ctot;
  const int ty = tap / g.KW, tx = tap - ty * g.KW;
  const bool own = c < g.C;
  const float* cplane = own ? src + (size_t)c * plane_s : src2 + (size_t)(c - g.C) * plane_s;
  const size_t fa = (size_t)g.Cout * plane_a, fb = own ? (size_t)g.C * plane_s : 0;
  const float* ap = a + (size_t)row * plane_a + 8 * kl;

  struct Step { float4 a0, a1; float b[8]; unsigned ok; };
  long p = (long)s0 << 4;
  int n = (int)(p / plane_a), rem = (int)(p - (long)n * plane_a);  // frame and pixel within it of the NEXT load
  auto load = [&](Step& t) {
    const float4* pa = reinterpret_cast<const float4*>(ap + (size_t)n * fa + rem);
    t.a0 = pa[0]; t.a1 = pa[1];
    const int q = rem + 8 * kl;
    const int y = q >> log2_wq, x0 = q & (g.Wq - 1);
    const int sy = y * g.SS + ty * g.TS + g.OFFY;
    const bool rok = col_ok && sy >= 0 && sy < g.Hs;
    const float* rp = cplane + (size_t)n * fb + (size_t)(rok ? sy : 0) * g.Ws;
    unsigned ok = 0;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int sx = (x0 + e) * g.SS + tx * g.TS + g.OFFX;
      const bool in = rok && sx >= 0 && sx < g.Ws;
      ok |= in ? 1u << e : 0u;
      t.b[e] = rp[in ? sx : 0];
    }
    t.ok = ok;
  };
  auto advance = [&](int s) {  // to step s + 1, clamped to the last one (loads stay unconditional)
    if (s + 1 < s1) { rem += 16; if (rem >= plane_a) { rem -= plane_a; ++n; } }
  };
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float bsum = 0.f;
  Step cur, nxt;
  if (active) {
    load(cur);
    advance(s0);
  }
#pragma unroll 1
  for (int s = s0; s < s1; ++s) {
    load(nxt);
    advance(s + 1);
    float xa[8] = {cur.a0.x, cur.a0.y, cur.a0.z, cur.a0.w, cur.a1.x, cur.a1.y, cur.a1.z, cur.a1.w};
    float xb[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { bsum += xa[e]; xb[e] = cur.b[e]; }
    if (pre_act_a) act_inplace<8>(xa, g.act);
    if (g.pre_act) act_inplace<8>(xb, g.act);
    u16x8 qa[SPLIT], qb[SPLIT];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      unsigned short pa[SPLIT], pb[SPLIT];
      split_bf16<SPLIT>(xa[e], pa);
      split_bf16<SPLIT>((cur.ok >> e) & 1u ? xb[e] : 0.f, pb);  // zero padding applies to the activated tensor
#pragma unroll
      for (int k = 0; k < SPLIT; ++k) { qa[k][e] = pa[k]; qb[k][e] = pb[k]; }
    }
#pragma unroll
    for (int ord = SPLIT - 1; ord >= 0; --ord)
#pragma unroll
      for (int sa = 0; sa <= ord; ++sa)
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, qa[sa]), __builtin_bit_cast(bf16x8, qb[ord - sa]), acc, 0, 0, 0);
    cur = nxt;
  }
  if (active) {
#pragma unroll
    for (int r = 0; r < 16; ++r) atomicAdd(&red[r * 64 + lane], acc[r]);
    bsum += __shfl_xor(bsum, 32, kWave);
    if (kl == 0) atomicAdd(&red[16 * 64 + il], bsum);
  }
  __syncthreads();
  if (threadIdx.x < 64) {
    if (col_ok) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int orow = (r & 3) + 8 * (r >> 2) + 4 * kl;
        if (orow < g.Cout) atomicAdd(&dwp[((size_t)orow * taps + tap) * g.Cpad + c], red[r * 64 + lane]);
      }
    }
    if (dbias != nullptr && kl == 0 && il < g.Cout) atomicAdd(&dbias[il], red[16 * 64 + il]);
  }
}

}  // namespace mtrssm
