// Weight-resident 3x3 gather kernel for the residual stacks (included by conv.hip after conv_split.h).
//
// The residual stacks run 3x3 / stride 1 / pad 1 layers with 32..128 channels on 64-pixel planes (8x8 vision, 16x4 audio)
// over B*T = 3200 frames: a
// layer is thousands of small tiles against ONE small weight matrix (64 x 576 values).  conv_gather_split_kernel
// re-stages that matrix per tile and per 16-channel step through LDS (two barriers a step, ~12 steps a tile) and runs
// at the length of that dependency chain: the MFMA pipe is 20 % busy (profiles/round1_notes.md, round2_notes.md).
// Here a workgroup is persistent (one per CU, one wave per SIMD, 512 registers per lane) and keeps ITS share of the
// weights -- both bf16 pieces of a 32-output-channel x K slab per wave, 288 registers at K = 576 -- in registers as
// ready-made MFMA A operands for the whole launch.  Only the pixels stream: a tile's frames are activated, split into
// bf16 pieces and written ONCE into a channel-innermost haloed image [frame][(H + 2) x (W + 2) positions][CIN] in LDS (the halo
// is zeroed once per launch and never written again), every MFMA B operand is one ds_read_b128 from it (2/3 of a read
// per MFMA: MI355X_MICROARCH.md "Issued between MFMAs by one wave per SIMD"), the next tile's frames are in flight in
// registers under this tile's MFMAs (double-buffered image, one barrier per tile), and the epilogue operands of a tile
// are requested before its MFMA loop.
//
// Wave roles (4 waves): NCT output-channel tiles x NPG frames x KS halves of the input channels, NCT * NPG * KS = 4;
// every wave owns all 64 pixels (two 32-pixel MFMA tiles) of one frame.  KS = 2 (K = 1152) meets in LDS once per tile.
#pragma once

namespace mtrssm {

constexpr int kResThreads = 256;
constexpr int kResPos = 108;  // haloed positions per frame: 10 x 10 (8 x 8 planes) or 18 x 6 (16 x 4 / 4 x 16 planes)

// Development aid (tools/resident_probe.py): when set, thread 0 of workgroup 0 stamps s_memtime (100 MHz) at the phase
// boundaries of its first tiles into this buffer.  Null in normal operation.
__device__ unsigned long long* g_res_prof = nullptr;
#define MTRSSM_RES_STAMP(i)                                                          \
  do {                                                                               \
    if (prof && it_no < 6) prof[2 + it_no * 8 + (i)] = __builtin_readcyclecounter(); \
  } while (0)

// Image rows: one position, all CIN channels of one piece, padded by 16 bytes: the row pitch is an odd number of 16-byte
// slots, so the 16 lanes one ds_read_b128 cycle serves (consecutive positions, the same channel slot) fall on 16 different
// slots of the 256-byte bank window, and every address is (lane base) + (tap: a scalar) + (channel block, piece: immediates).
__host__ __device__ constexpr int res_row_bytes(int cin) { return cin * 2 + 16; }

// Weight staging (prologue): groups of output-channel rows go through LDS in rows of 9 * CIN bf16 + 16 bytes (an odd number
// of 16-byte slots: the 16 lanes of a ds_read_b128 cycle, 16 rows, fall on 16 different slots of the bank window).
__host__ __device__ constexpr int res_wrow_bytes(int cin) { return 9 * cin * 2 + 16; }
__host__ __device__ constexpr int res_group_rows(int cin) { return cin >= 128 ? 32 : 64; }

template <int CIN, int NCT, int KS>
__host__ __device__ constexpr size_t res_lds_bytes() {
  constexpr int NPG = 4 / (NCT * KS);
  constexpr size_t run = (size_t)4 * NPG * kResPos * res_row_bytes(CIN) + (size_t)NCT * 32 * sizeof(float) +
                         (KS == 2 ? (size_t)NCT * 2 * 16 * 64 * sizeof(float) : 0);
  constexpr size_t stage = (size_t)res_group_rows(CIN) * res_wrow_bytes(CIN);
  return run > stage ? run : stage;
}

template <int CIN, int NCT, int KS>
__global__ __launch_bounds__(kResThreads, 1) void conv3x3_resident_kernel(const GatherProblem pa, const GatherProblem pb) {
  static_assert(NCT * KS == 4 || NCT * KS == 2 || NCT * KS == 1, "4 waves");
  constexpr int NPG = 4 / (NCT * KS);  // frames per tile
  constexpr int CW = CIN / KS;         // input channels per wave
  constexpr int CB = CW / 16;          // 16-channel k-blocks per tap
  constexpr int KB = 9 * CB;           // k-blocks per wave
  constexpr int RB = res_row_bytes(CIN);  // bytes per image row (one position, all channels, one piece, padded)
  constexpr int IMG = NPG * kResPos * RB;  // one piece of one buffer
  constexpr int OCT = CIN / 8;         // 8-channel octets per position
  constexpr int NIT = NPG * OCT / 4;   // staging items (frame, octet) per wave
  static_assert(NIT >= 1 && CB >= 1, "shape");
  const bool second = blockIdx.x >= (unsigned)pa.nx;  // workgroup-uniform
  const GatherProblem P = second ? pb : pa;
  const MtrssmConvGeom g = P.g;
  const float* __restrict__ src = P.src;
  const unsigned short* __restrict__ wq = P.wq;
  const float* __restrict__ bias = P.bias;
  const float* __restrict__ actgrad_in = P.actgrad_in;
  const float* __restrict__ add_in = P.add_in;
  float* __restrict__ out = P.out;
  const int wg = second ? (int)blockIdx.x - pa.nx : (int)blockIdx.x, nwg = P.nx;
  const int nframes = g.N;
  const int ntiles = (nframes + NPG - 1) / NPG;
  if (wg >= ntiles) return;  // workgroup-uniform

  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  unsigned char* patch = lds_raw;                                            // [2 buffers][2 pieces][IMG]
  float* bias_lds = reinterpret_cast<float*>(lds_raw + (size_t)4 * IMG);     // [NCT * 32]
  float* red = bias_lds + NCT * 32;                                          // KS == 2: [NCT][2][16][64]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kl = lane >> 5, il = lane & 31;
  const int ct = KS == 2 ? (wave & 1) : (wave % NCT);
  const int kh = KS == 2 ? (wave >> 1) : 0;
  const int pgi = KS == 2 ? 0 : (wave / NCT);

  unsigned long long* prof = (blockIdx.x == 0 && tid == 0) ? g_res_prof : nullptr;
  if (prof) prof[0] = __builtin_readcyclecounter();
  // ---- this wave's weights: MFMA A operands (row = output channel il of tile ct, 8 k-values per lane) for every k-block.
  // Read straight from the packed layout a lane's 16 bytes sit 18 * CIN bytes from its neighbour's: every load touches 32
  // lines for 1 KB and the 72 loads of a wave re-touch each line four times -- 19 k cycles of L2 traffic per launch
  // (measured).  Instead the rows go through LDS: coalesced 16-byte chunks in, fragments out.
  bf16x8 a[KB][2];
  {
    constexpr int GR = res_group_rows(CIN), WROW = res_wrow_bytes(CIN), CPR = 9 * CIN * 2 / 16;  // chunks per row
    constexpr int NGRP = NCT * 32 / GR > 0 ? NCT * 32 / GR : 1;
    constexpr int ROWS = NCT * 32 < GR ? NCT * 32 : GR;
    const size_t piece = (size_t)g.CoutPad * 9 * g.Cpad;  // host: Cpad == CIN
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
      for (int grp = 0; grp < NGRP; ++grp) {
        const u32x4* gsrc = reinterpret_cast<const u32x4*>(wq + s * piece + (size_t)grp * ROWS * 9 * CIN);
        constexpr int NCH = ROWS * CPR / kResThreads;  // chunks per thread: all requested before the first is stored
        static_assert(ROWS * CPR % kResThreads == 0, "whole chunks per thread");
        u32x4 wv[NCH];
#pragma unroll
        for (int i = 0; i < NCH; ++i) wv[i] = gsrc[tid + i * kResThreads];
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
          const int c = tid + i * kResThreads, row = c / CPR, col = c - row * CPR;
          *reinterpret_cast<u32x4*>(lds_raw + row * WROW + col * 16) = wv[i];
        }
        __syncthreads();
        if (NGRP == 1 || (ct * 32) / ROWS == grp) {  // wave-uniform
          const unsigned char* wl = lds_raw + ((ct * 32) % ROWS + il) * WROW + (kh * CW + 8 * kl) * 2;
#pragma unroll
          for (int kb = 0; kb < KB; ++kb) {
            const int t = kb / CB, cb = kb % CB;
            a[kb][s] = *reinterpret_cast<const bf16x8*>(wl + (t * CIN + cb * 16) * 2);
          }
        }
        __syncthreads();
      }
    }
  }
  if (prof) prof[56] = __builtin_readcyclecounter();
  // ---- zero both images once (the halo stays zero for the whole launch)
  for (int o = tid * 16; o < 4 * IMG; o += kResThreads * 16) *reinterpret_cast<u32x4*>(patch + o) = u32x4{0u, 0u, 0u, 0u};
  // this lane's 16 accumulator rows are output channels cbase + (r & 3) + 8 * (r >> 2) (host: Cout == 32 * NCT, no ragged
  // channel tile); the bias waits in LDS for the epilogue
  const int cbase = ct * 32 + 4 * kl;
  if (tid < NCT * 32) bias_lds[tid] = bias ? bias[tid] : 0.f;
  // planes of 64 pixels, W = 4, 8 or 16 wide (host); haloed rows are W + 2 positions
  const int pw = g.Wq + 2, wsh = g.Wq == 8 ? 3 : (g.Wq == 4 ? 2 : 4);
  const int tstep = g.TS > 0 ? RB : -RB;  // forward gather / flipped (backward-data) gather: bytes per position step
  const int trow = pw * tstep;
  int pos0[2];
#pragma unroll
  for (int pt = 0; pt < 2; ++pt) {
    const int pixel = pt * 32 + il;
    pos0[pt] = (pgi * kResPos + ((pixel >> wsh) + 1) * pw + (pixel & (g.Wq - 1)) + 1) * RB + (kh * (CW / 8) + kl) * 16;  // byte offset
  }
  const int spos = ((lane >> wsh) + 1) * pw + (lane & (g.Wq - 1)) + 1;  // staging: lane = pixel of the frame

  float pv[NIT][8];
  auto stage_load = [&](int tile) {
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int q = wave * NIT + it, fi = q / OCT, o = q % OCT;
      int f = tile * NPG + fi;
      f = f < nframes ? f : nframes - 1;  // a frame past the end: any finite values, its outputs are never stored
      const float* bp = src + ((size_t)f * CIN + o * 8) * 64 + lane;
#pragma unroll
      for (int u = 0; u < 8; ++u) pv[it][u] = bp[u * 64];
    }
  };
  auto stage_store = [&](int buf) {
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int q = wave * NIT + it, fi = q / OCT, o = q % OCT;
      float x[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) x[u] = pv[it][u];
      if (g.pre_act) {  // host: act is Identity, ELU or ReLU (no libm call in this kernel)
        if (g.act == MTRSSM_ACT_ELU) {
#pragma unroll
          for (int u = 0; u < 8; ++u) x[u] = elu_fast(x[u]);
        } else if (g.act == MTRSSM_ACT_RELU) {
#pragma unroll
          for (int u = 0; u < 8; ++u) x[u] = x[u] > 0.f ? x[u] : 0.f;
        }
      }
      u16x8 qv[2];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        unsigned short p[2];
        split_bf16<2>(x[u], p);
        qv[0][u] = p[0];
        qv[1][u] = p[1];
      }
      unsigned char* dst = patch + (size_t)buf * 2 * IMG + (fi * kResPos + spos) * RB + o * 16;
      *reinterpret_cast<u16x8*>(dst) = qv[0];
      *reinterpret_cast<u16x8*>(dst + IMG) = qv[1];
    }
  };

  if (prof) prof[57] = __builtin_readcyclecounter();
  stage_load(wg);
  if (prof) prof[58] = __builtin_readcyclecounter();
  __syncthreads();  // zero fill done before the interior is written
  if (prof) prof[59] = __builtin_readcyclecounter();
  stage_store(0);
  __syncthreads();

  if (prof) prof[1] = __builtin_readcyclecounter();
  struct Frag { bf16x8 b[2][2]; };
  int it_no = 0;
  for (int tile = wg; tile < ntiles; tile += nwg, ++it_no) {
    const int buf = it_no & 1;
    MTRSSM_RES_STAMP(0);
    const unsigned char* img = patch + (size_t)buf * 2 * IMG;
    {
      const int next = tile + nwg;
      stage_load(next < ntiles ? next : tile);  // unconditional: no load inside a branch
    }
    // epilogue operands of this tile: requested now, consumed after the MFMA loop
    const int frame = tile * NPG + pgi;
    const bool fv = frame < nframes;
    const unsigned obase = (unsigned)(fv ? frame : nframes - 1) * (unsigned)(NCT * 32 * 64);
    constexpr int NE = KS == 2 ? 1 : 2;  // pixel tiles this wave finishes (KS == 2: tile kh, its partner the other one)
    // Epilogue operands (act'(x) input, skip gradient): ONE pixel tile's worth of registers.  Tile 0's are requested in
    // the middle of the MFMA loop (in the registers the staging has just freed); tile 1's lines are only touched there
    // so that its real loads, issued after tile 0 is combined, hit in L2.
    float gv[16], av[16];
    unsigned ob[NE];
#pragma unroll
    for (int e = 0; e < NE; ++e) ob[e] = obase + (unsigned)(cbase * 64 + (KS == 2 ? kh : e) * 32 + il);
    auto epi_load = [&](int e) {
      if (actgrad_in) {
#pragma unroll
        for (int r = 0; r < 16; ++r) gv[r] = actgrad_in[ob[e] + (unsigned)(((r & 3) + 8 * (r >> 2)) * 64)];
      }
      if (add_in) {
#pragma unroll
        for (int r = 0; r < 16; ++r) av[r] = add_in[ob[e] + (unsigned)(((r & 3) + 8 * (r >> 2)) * 64)];
      }
    };
    // one load touches all of tile 1's lines: 32 channel rows x 128 bytes of each operand = 64 lines, one per lane; its
    // value is never used, only kept (one register) until the loop is over so that the compiler counts the load
    float touched = 0.f;
    auto epi_touch = [&]() {
      const float* base = (lane < 32 || !add_in) ? actgrad_in : add_in;
      if (!base) base = add_in;
      if (base) touched = base[obase + (unsigned)((ct * 32 + (lane & 31)) * 64 + 32)];
    };
    f32x16 acc[2];
#pragma unroll
    for (int pt = 0; pt < 2; ++pt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[pt][r] = 0.f;

    auto read_frag = [&](Frag& f, int kb) {
      const int t = kb / CB, cb = kb % CB;
      const int toff = (t / 3 - 1) * trow + (t % 3 - 1) * tstep;  // scalar
#pragma unroll
      for (int pt = 0; pt < 2; ++pt) {
        const unsigned char* ad = img + (pos0[pt] + toff) + cb * 32;
        f.b[pt][0] = *reinterpret_cast<const bf16x8*>(ad);
        f.b[pt][1] = *reinterpret_cast<const bf16x8*>(ad + IMG);
      }
    };
    // The scheduler, left alone, sinks every ds_read_b128 of the unrolled loop in front of its MFMA (register pressure
    // beats latency there: 58 instead of 32 cycles per MFMA measured); the barriers pin "next block's reads, then this
    // block's MFMAs".
    // Fragments are requested TWO blocks ahead: the compiler waits with lgkmcnt(0) (scalar loads share the counter), i.e.
    // also for the newest reads, which by then are a whole block of MFMAs old.
    Frag fr[3];
    read_frag(fr[0], 0);
    read_frag(fr[1], 1);
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      Frag& cur = fr[kb % 3];
      if (kb + 2 < KB) read_frag(fr[(kb + 2) % 3], kb + 2);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int pt = 0; pt < 2; ++pt) {
        acc[pt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[kb][0], cur.b[pt][1], acc[pt], 0, 0, 0);
        acc[pt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[kb][1], cur.b[pt][0], acc[pt], 0, 0, 0);
        acc[pt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[kb][0], cur.b[pt][0], acc[pt], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (kb == KB / 2) {
        MTRSSM_RES_STAMP(1);
        stage_store(buf ^ 1);  // the next tile's image: conversions in the MFMAs' shadow
        epi_load(0);
        if (NE == 2) epi_touch();
        MTRSSM_RES_STAMP(2);
      }
    }
    MTRSSM_RES_STAMP(3);
    f32x16 fin[NE];
    if (KS == 2) {  // the two halves of K meet: each wave hands over the pixel tile it does not finish
      float* mine = red + ((size_t)(ct * 2 + kh) * 16) * 64 + lane;
      const float* theirs = red + ((size_t)(ct * 2 + (kh ^ 1)) * 16) * 64 + lane;
#pragma unroll
      for (int r = 0; r < 16; ++r) mine[r * 64] = kh == 0 ? acc[1][r] : acc[0][r];
      lds_barrier();
#pragma unroll
      for (int r = 0; r < 16; ++r) fin[0][r] = (kh == 0 ? acc[0][r] : acc[1][r]) + theirs[r * 64];
    } else {
#pragma unroll
      for (int e = 0; e < NE; ++e) fin[e] = acc[e];
    }
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      if (e > 0) {
        asm volatile("" ::"v"(touched));
        epi_load(e);
      }
      float res[16];
      if (actgrad_in) {  // act'(x) from the layer input x: one workgroup-uniform branch per tile, not per element
        if (g.act == MTRSSM_ACT_ELU) {
#pragma unroll
          for (int r = 0; r < 16; ++r) gv[r] = gv[r] > 0.f ? 1.f : __expf(gv[r]);
        } else if (g.act == MTRSSM_ACT_RELU) {
#pragma unroll
          for (int r = 0; r < 16; ++r) gv[r] = gv[r] > 0.f ? 1.f : 0.f;
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) gv[r] = 1.f;
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v = fin[e][r] + bias_lds[cbase + (r & 3) + 8 * (r >> 2)];
        if (actgrad_in) v *= gv[r];
        if (add_in) v += av[r];
        res[r] = v;
      }
      if (fv) {
#pragma unroll
        for (int r = 0; r < 16; ++r) out[ob[e] + (unsigned)(((r & 3) + 8 * (r >> 2)) * 64)] = res[r];
      }
    }
    MTRSSM_RES_STAMP(4);
    lds_barrier();  // the next image is complete; everyone is done reading this one (LDS only: a __syncthreads would also
                    // wait for this tile's stores and the next tile's frames in flight)
    MTRSSM_RES_STAMP(5);
  }
}

}  // namespace mtrssm
