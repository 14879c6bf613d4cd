// Weight-resident 3x3 gather kernel for the residual stacks (included by conv.hip after conv_split.h).
//
// The residual stacks run 3x3 / stride 1 / pad 1 layers with 32..128 channels on 64-pixel planes (8x8 vision, 16x4 audio)
// over B*T = 3200 frames: a layer is thousands of small tiles against ONE small weight matrix (64 x 576 values).
// conv_gather_split_kernel re-stages that matrix per tile and per 16-channel step through LDS (two barriers a step, ~12
// steps a tile) and runs at the length of that dependency chain: the MFMA pipe is 20 % busy (profiles/round1_notes.md,
// round2_notes.md).  Here a workgroup is persistent (one per CU, one wave per SIMD, 512 registers per lane) and keeps ITS
// share of the weights -- both bf16 pieces of a 32-output-channel x K slab per wave, 288 registers at K = 576, most of them
// pinned to AGPRs -- in registers as ready-made MFMA A operands for the whole launch.  Only the pixels stream: a tile's
// frames are activated, split into bf16 pieces and written ONCE into a channel-innermost haloed image
// [frame][(H + 2) x (W + 2) positions][CIN] in LDS (the halo is zeroed once per launch and never written again), and
// every MFMA B operand is one ds_read_b128 from it (2/3 of a read per MFMA: MI355X_MICROARCH.md "Issued between MFMAs
// by one wave per SIMD").
//
// Wave roles (4 waves): NCT output-channel tiles x NPG frames x KS halves of the input channels, NCT * NPG * KS = 4;
// every wave owns the 64 pixels of one frame and walks them as two UNITS of 32 (one MFMA column tile each).  While a
// unit's 3 * KB MFMAs run, the wave's other instructions ride in their shadow, k-block by k-block: the previous unit's
// epilogue rows (bias is the accumulator's start value; act'(x) and the skip gradient are requested a unit ahead), the
// requests for the next tile's frames (unit 0) and their conversion into the other image (unit 1).  One LDS-only
// barrier per tile.  KS = 2 (K = 1152) exchanges half of the rows through LDS after every unit.
// Measured (profiles/round2_notes.md): paired 64 -> 64 layer 200 -> 93 us forward, 122 us backward-data.
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
#define MTRSSM_SGB(mask_, n_) __builtin_amdgcn_sched_group_barrier(mask_, n_, 0)

// Image rows: one position, all CIN channels of one piece, padded by 16 bytes: the row pitch is an odd number of 16-byte
// slots, so the 16 lanes one ds_read_b128 cycle serves (consecutive positions, the same channel slot) fall on 16 different
// slots of the 256-byte bank window, and every address is (lane base) + (tap: a scalar) + (channel block, piece: immediates).
__host__ __device__ constexpr int res_row_bytes(int cin) { return cin * 2 + 16; }

// Weight staging (prologue): groups of output-channel rows go through LDS in rows of 9 * CIN bf16 + 16 bytes (an odd number
// of 16-byte slots: the 16 lanes of a ds_read_b128 cycle, 16 rows, fall on 16 different slots of the bank window).
__host__ __device__ constexpr int res_wrow_bytes(int cin) { return 9 * cin * 2 + 16; }
__host__ __device__ constexpr int res_group_rows(int cin) { return cin >= 128 ? 32 : 64; }

// B-operand reads and their waits are inline asm.  The compiler's own s_waitcnt for these reads came out as lgkmcnt(0) every
// third k-block, right behind the newest requests.  LDS operations return in order, so "at most 4 outstanding" releases the
// fragment requested two blocks ago whatever else (the staging writes) is in flight; res_wait ties the fragment registers
// to that wait.  (Measured: no change of the unit time -- the schedule just no longer depends on the compiler's choice.)
template <int IMGB>
__device__ __forceinline__ void res_read_pair(bf16x8& hi, bf16x8& lo, unsigned addr, int cb) {
  switch (cb) {
    case 0: asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:%3" : "=&v"(hi), "=&v"(lo) : "v"(addr), "n"(IMGB)); break;
    case 1: asm volatile("ds_read_b128 %0, %2 offset:32\n\tds_read_b128 %1, %2 offset:%3" : "=&v"(hi), "=&v"(lo) : "v"(addr), "n"(IMGB + 32)); break;
    case 2: asm volatile("ds_read_b128 %0, %2 offset:64\n\tds_read_b128 %1, %2 offset:%3" : "=&v"(hi), "=&v"(lo) : "v"(addr), "n"(IMGB + 64)); break;
    default: asm volatile("ds_read_b128 %0, %2 offset:96\n\tds_read_b128 %1, %2 offset:%3" : "=&v"(hi), "=&v"(lo) : "v"(addr), "n"(IMGB + 96)); break;
  }
}
template <int N>
__device__ __forceinline__ void res_wait(bf16x8& b0, bf16x8& b1) {
  asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(b0), "+v"(b1) : "n"(N));
}

template <int CIN, int NCT, int KS>
__host__ __device__ constexpr size_t res_lds_bytes() {
  constexpr int NPG = 4 / (NCT * KS);
  constexpr size_t run = (size_t)4 * NPG * kResPos * res_row_bytes(CIN) + (size_t)NCT * 32 * sizeof(float) +
                         (KS == 2 ? (size_t)NPG * NCT * 2 * 16 * 64 * sizeof(float) : 0);
  constexpr size_t stage = (size_t)res_group_rows(CIN) * res_wrow_bytes(CIN);
  return run > stage ? run : stage;
}

// FUSE: the whole residual block  y = x + W1 * act(b3 + W3 (*) act(x)) + b1  in this launch (task 3).  A unit's
// 32 x 32 accumulator (rows = 32 of the block's intermediate channels, columns = 32 pixels) IS the B operand of the 1x1's MFMAs
// once activated and split into pieces: a lane's 16 rows are 2 k-blocks of 8 values when the 1x1's K axis is taken in the
// order (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5) -- the A operands (rows of W1, prepared once per workgroup in LDS) are simply
// gathered in that order.  Every wave multiplies ITS 32 intermediate channels into all CIN output channels (12 more MFMAs on
// top of the unit's 108), the NCT partial sums meet through LDS, each wave finishes CIN / NCT channels: + b1 + x, stored to
// out2.  The intermediate is still stored to out (the backward pass reads it); it is no longer read back.
template <int CIN, int NCT>
__host__ __device__ constexpr size_t res_fuse_bytes() {
  // W1 fragments [NCT][CIN / 32 tiles x 2 k-blocks][2 pieces][64 lanes] of 16 bytes + partial sums [4 waves][CIN / 2 rows][64 lanes]
  // (NCT == 2: the 16 rows the partner wave finishes)
  return (size_t)NCT * (CIN / 32) * 2 * 2 * 64 * 16 + (size_t)4 * (NCT == 2 ? 16 : CIN / 2) * 64 * sizeof(float);
}

template <int CIN, int NCT, int KS, bool EPI, bool FUSE = false>  // EPI: the epilogue has operands (act'(x) input and / or skip gradient)
__global__ __launch_bounds__(kResThreads, 1) void conv3x3_resident_kernel(const GatherProblem pa, const GatherProblem pb) {
  static_assert(NCT * KS == 4 || NCT * KS == 2 || NCT * KS == 1, "4 waves");
  static_assert(!FUSE || (KS == 1 && !EPI && CIN % 32 == 0), "fused block: whole K per wave, no epilogue operands");
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
  const int ntiles = nframes / NPG;  // host: N % NPG == 0
  if (wg >= ntiles) return;  // workgroup-uniform

  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  unsigned char* patch = lds_raw;                                            // [2 buffers][2 pieces][IMG]
  float* red = reinterpret_cast<float*>(lds_raw + (size_t)4 * IMG);          // KS == 2: [NPG][NCT][2][8][64]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kl = lane >> 5, il = lane & 31;
  // wave -> (output-channel tile, half of K, frame of the tile): <128, 2, 2>: (w & 1, w >> 1, 0); <64, 1, 2>: (0, w & 1, w >> 1)
  const int ct = wave % NCT;
  const int kh = (wave / NCT) % KS;
  const int pgi = wave / (NCT * KS);

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
  // Register classes: VALU operands must be architectural VGPRs, MFMA operands may be AGPRs.  Left to itself the allocator
  // fills the VGPR half with these long-lived fragments first and then spills them to scratch when the loop's VALU values
  // need room.  Pinning most of them to AGPRs (240 = 256 minus the accumulator) leaves the VGPR half to the loop.
  constexpr int kPinned = (2 * KB < 60 ? 2 * KB : 60);
#pragma unroll
  for (int kb = 0; kb < KB; ++kb)
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
      if (kb * 2 + s2 < kPinned) asm volatile("" : "+a"(a[kb][s2]));
  if (prof) prof[56] = __builtin_readcyclecounter();
  // ---- zero both images once (the halo stays zero for the whole launch)
  for (int o = tid * 16; o < 4 * IMG; o += kResThreads * 16) *reinterpret_cast<u32x4*>(patch + o) = u32x4{0u, 0u, 0u, 0u};
  // ---- FUSE: the 1x1's A operands, in the k order of the accumulator rows (see the kernel's header)
  constexpr int NOT = CIN / 32;            // output-channel tiles of the 1x1 (its Cout == CIN: a residual block)
  constexpr int NF = NOT * 2;              // A fragments per wave: (tile, k-block of 16 intermediate channels)
  constexpr int RF = NOT * 16 / NCT;       // accumulator rows of the 1x1 this wave finishes per unit
  unsigned char* w1l = lds_raw + res_lds_bytes<CIN, NCT, KS>();                       // behind everything the 3x3 uses
  float* red2 = reinterpret_cast<float*>(w1l + (size_t)NCT * NF * 2 * 64 * 16);       // [wave][NOT * 16 rows][64 lanes]
  float* __restrict__ out2 = P.out2;
  const int co2 = (ct * RF / 16) * 32 + ((ct * RF) % 16 / 4) * 8 + 4 * kl;            // first of this lane's finished channels
  float b1v[FUSE ? RF : 1];
  if constexpr (FUSE) {
    const float* __restrict__ w1 = P.w1;
    const float* __restrict__ b1 = P.b1;
    constexpr int CMID = NCT * 32;
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      const int ot = f / 2, kb2 = f % 2;
      unsigned hi[4], lo[4];
#pragma unroll
      for (int e2 = 0; e2 < 4; ++e2) {
        unsigned short p0[2], p1[2];
        const int e = 2 * e2;
        const int k0 = ct * 32 + 4 * kl + (e & 3) + 8 * (2 * kb2 + (e >> 2));
        split_bf16<2>(w1[(size_t)(ot * 32 + il) * CMID + k0], p0);
        split_bf16<2>(w1[(size_t)(ot * 32 + il) * CMID + k0 + 1], p1);
        hi[e2] = (unsigned)p0[0] | ((unsigned)p1[0] << 16);
        lo[e2] = (unsigned)p0[1] | ((unsigned)p1[1] << 16);
      }
      unsigned char* dst = w1l + ((size_t)((ct * NF + f) * 2) * 64 + lane) * 16;
      *reinterpret_cast<u32x4*>(dst) = u32x4{hi[0], hi[1], hi[2], hi[3]};
      *reinterpret_cast<u32x4*>(dst + 64 * 16) = u32x4{lo[0], lo[1], lo[2], lo[3]};
    }
#pragma unroll
    for (int j = 0; j < RF; ++j) b1v[j] = b1[co2 + (j & 3) + 8 * (j >> 2)];
  }
  // this lane's 16 accumulator rows are output channels cbase + (r & 3) + 8 * (r >> 2) (host: Cout == 32 * NCT, no ragged
  // channel tile); the bias is the accumulators' start value (KS == 2: in the first half of K only)
  const int cbase = ct * 32 + 4 * kl;
  float bv[16];
  {
    const float* bsafe = bias ? bias : src;  // any readable address: the value is dropped below
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float b = bsafe[cbase + (r & 3) + 8 * (r >> 2)];
      bv[r] = (bias && kh == 0) ? b : 0.f;
    }
  }
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
  // activation switches as lane-uniform selects (no branch inside the MFMA loop: a branch ends a scheduling region)
  const bool act_elu = g.act == MTRSSM_ACT_ELU, act_relu = g.act == MTRSSM_ACT_RELU, pre = g.pre_act != 0;
  auto act_fwd_sel = [&](float x) {  // pre-activation of a staged value
    float e = __expf(x) - 1.f;
    asm volatile("" : "+v"(e));  // computed unconditionally: the compiler would branch around the exponential
    const float neg = act_elu ? e : (act_relu ? 0.f : x);
    return (x > 0.f || !pre) ? x : neg;
  };
  auto act_grad_sel = [&](float x) {  // act'(x) from the layer input
    float e = __expf(x);
    asm volatile("" : "+v"(e));
    const float neg = act_elu ? e : (act_relu ? 0.f : 1.f);
    return x > 0.f ? 1.f : neg;
  };
  auto act_mid_sel = [&](float x) {  // FUSE: the block's activation on the intermediate
    const float e = __expf(x) - 1.f;
    const float neg = act_elu ? e : (act_relu ? 0.f : x);
    return x > 0.f ? x : neg;
  };

  // ---- staging of a tile's frames: (frame, 8-channel octet) items per wave, lane = pixel
  float pv[NIT][8];
  auto stage_load_part = [&](int tile, int part, int nparts) {  // loads [part * NIT * 8 / nparts, ...) of the tile's items
    constexpr int TOT = NIT * 8;
    const int per = TOT / nparts;
    int opaque = 0;  // addresses are formed HERE: hoisted to the top of the tile they sit in registers for its whole length
    asm volatile("" : "+s"(opaque));
#pragma unroll
    for (int j = 0; j < TOT; ++j) {
      if (j / per != part) continue;
      const int it = j / 8, u = j % 8;
      const int q = wave * NIT + it, fi = q / OCT, o = q % OCT;
      pv[it][u] = src[((size_t)(tile * NPG + fi + opaque) * CIN + o * 8 + u) * 64 + lane];
    }
  };
  auto stage_store_chunk = [&](int buf, int c) {  // chunk c: channels 2 * (c % 4), +1 of item c / 4 -> 4 bytes per piece
    const int it = c / 4, u2 = c % 4;
    const int q = wave * NIT + it, fi = q / OCT, o = q % OCT;
    unsigned short p0[2], p1[2];
    split_bf16<2>(act_fwd_sel(pv[it][2 * u2]), p0);
    split_bf16<2>(act_fwd_sel(pv[it][2 * u2 + 1]), p1);
    int opaque = 0;
    asm volatile("" : "+s"(opaque));
    unsigned char* dst = patch + (size_t)buf * 2 * IMG + (fi * kResPos + spos + opaque) * RB + o * 16 + u2 * 4;
    *reinterpret_cast<unsigned*>(dst) = (unsigned)p0[0] | ((unsigned)p1[0] << 16);
    *reinterpret_cast<unsigned*>(dst + IMG) = (unsigned)p0[1] | ((unsigned)p1[1] << 16);
  };
  constexpr int NSC = NIT * 4;  // staging chunks per tile

  if (prof) prof[57] = __builtin_readcyclecounter();
  stage_load_part(wg, 0, 1);
  __syncthreads();  // zero fill done before the interior is written
#pragma unroll
  for (int c = 0; c < NSC; ++c) stage_store_chunk(0, c);
  __syncthreads();
  if (prof) prof[1] = __builtin_readcyclecounter();

  // ---- the pipeline.  A unit = (tile, pixel half h): 3 * KB MFMAs into the accumulator.  In their shadow (between MFMAs, in program
  // order: the wave issues in order) run the epilogue of the PREVIOUS unit (the other accumulator), the requests for this
  // unit's epilogue operands, and (h = 0) the requests / (h = 1) the conversion of the next tile's frames.  Every k-block is
  // one scheduling region: its three MFMAs alternate with at most ~6 VALU instructions each.
  constexpr int ROWS = KS == 2 ? 8 : 16;            // accumulator rows this wave finishes per unit (KS == 2: half of them)
  constexpr int RPE = (EPI && KB >= 36) ? 1 : 2;    // rows per epilogue region
  constexpr int NES = ROWS / RPE;                   // epilogue regions: k-blocks [1, 1 + NES)
  constexpr int NLD = EPI ? 4 : 0;                  // operand-request regions: k-blocks [0, NLD) of the unit they belong to
  constexpr int S0 = 1 + NES;                       // staging chunks: k-block 0 and [S0, KB) of unit h = 1
  constexpr int CPS = KB >= 36 ? 1 : 2;             // staging chunks per region
  static_assert((1 + (KB - S0)) * CPS >= NSC, "the staging chunks fit in one unit");
  static_assert(S0 <= KB, "the epilogue fits in one unit");
  // A unit's sums leave the AGPRs once, right after its last MFMA (180 cycles), and the epilogue rows in the next unit's
  // shadow work on that copy in architectural VGPRs: no v_accvgpr_read and no AGPR-sourced store between the MFMAs.
  f32x16 acc;
  float fin[ROWS];
  // epilogue operands of unit h live in set h: requested in the first k-blocks of their unit, consumed in the next unit's
  // epilogue rows a whole unit (> 2 us) later -- one set, requested after the previous rows were done, left ~1 us
  float gvs[2][16], avs[2][16], part[8];  // KS == 2: part = the partner wave's half sums of the previous unit
  // The first unit's epilogue slots have no previous unit: they store zeros where the SAME wave stores (tile wg, h = 1) one
  // unit later (same address, program order).  Host: whole tiles only (N % NPG == 0), so every frame exists.
  unsigned ob_prev = (unsigned)(wg * NPG + pgi) * (unsigned)(NCT * 32 * 64) + (unsigned)(cbase * 64 + il) + 32u;
#pragma unroll
  for (int r = 0; r < 16; ++r) gvs[0][r] = gvs[1][r] = avs[0][r] = avs[1][r] = 0.f;
#pragma unroll
  for (int r = 0; r < ROWS; ++r) fin[r] = 0.f;
#pragma unroll
  for (int r = 0; r < 8; ++r) part[r] = 0.f;
  const float* gsrc = EPI ? (actgrad_in ? actgrad_in : add_in) : nullptr;  // host: EPI launches have at least one operand
  const float* asrc = EPI ? (add_in ? add_in : actgrad_in) : nullptr;
  const bool has_g = actgrad_in != nullptr, has_a = add_in != nullptr;
  auto row_off = [](int r) { return (unsigned)(((r & 3) + 8 * (r >> 2)) * 64); };
  // finished value of row j of the previous unit (accumulator pp), stored when its frame exists
  auto epi_row = [&](int set, int j) {
    float v = fin[j];
    int r = j;
    if (KS == 2) v += part[j];
    if (EPI) {
      float gm = act_grad_sel(gvs[set][r]);
      asm volatile("" : "+v"(gm));
      gm = has_g ? gm : 1.f;
      const float am = has_a ? avs[set][r] : 0.f;
      v = v * gm + am;
    }
    return v;
  };

  struct Frag { bf16x8 b[2]; };
  int it_no = 0;
  for (int tile = wg; tile < ntiles; tile += nwg, ++it_no) {
    const int buf = it_no & 1;
    const unsigned char* img = patch + (size_t)buf * 2 * IMG;
    const int next = tile + nwg < ntiles ? tile + nwg : tile;
    const unsigned obase = (unsigned)(tile * NPG + pgi) * (unsigned)(NCT * 32 * 64) + (unsigned)(cbase * 64 + il);
    MTRSSM_RES_STAMP(0);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const unsigned ob_cur = obase + (unsigned)(h * 32);
      const unsigned ibase = (unsigned)(uintptr_t)img + (unsigned)pos0[h];  // LDS byte address (low half of the flat address)
      auto read_frag = [&](Frag& f, int kb) {
        const int t = kb / CB, cb = kb % CB;
        const int toff = (t / 3 - 1) * trow + (t % 3 - 1) * tstep;  // scalar
        res_read_pair<IMG>(f.b[0], f.b[1], ibase + (unsigned)toff, cb);
      };
      // fragments are requested two blocks ahead
      Frag fr[3];
      read_frag(fr[0], 0);
      read_frag(fr[1], 1);
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = bv[r];
      // FUSE: the skip values of the rows this wave finishes, requested a whole unit before their use
      const unsigned o2base = (unsigned)(tile * NPG + pgi) * (unsigned)(CIN * 64) + (unsigned)(co2 * 64 + il + h * 32);
      float xs[FUSE ? RF : 1];
      if constexpr (FUSE) {
#pragma unroll
        for (int j = 0; j < RF; ++j) xs[j] = src[o2base + row_off(j)];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) {
        Frag& cur = fr[kb % 3];
        if (kb + 2 < KB) {
          read_frag(fr[(kb + 2) % 3], kb + 2);
          res_wait<4>(cur.b[0], cur.b[1]);
        } else if (kb + 1 < KB) {
          res_wait<2>(cur.b[0], cur.b[1]);
        } else {
          res_wait<0>(cur.b[0], cur.b[1]);
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[kb][0], cur.b[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[kb][1], cur.b[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[kb][0], cur.b[0], acc, 0, 0, 0);
        // ---- the shadow work of this k-block
        if (kb >= 1 && kb < 1 + NES) {  // epilogue rows of the previous unit
          float res[RPE];
#pragma unroll
          for (int i = 0; i < RPE; ++i) res[i] = epi_row(h ^ 1, (kb - 1) * RPE + i);
#pragma unroll
          for (int i = 0; i < RPE; ++i) {
            const int j = (kb - 1) * RPE + i;
            out[ob_prev + row_off(j) + (KS == 2 ? (unsigned)(kh * 16 * 64) : 0u)] = res[i];
          }
        }
        if (EPI && kb < NLD) {  // this unit's epilogue operands, into set h
          // one base pointer per operand and half of the rows: the row offsets stay inside the 12-bit immediate (computed
          // as 32 separate 64-bit addresses they cost 64 registers for a whole tile)
          constexpr int RPL = ROWS / 4;
#pragma unroll
          for (int i = 0; i < RPL; ++i) {
            const int j = kb * RPL + i;
            const unsigned ob2 = ob_cur + (KS == 2 ? (unsigned)(kh * 16 * 64) : 0u) + (j >= 8 ? 16u * 64u : 0u);
            const float* gp = gsrc + ob2;
            const float* ap = asrc + ob2;
            asm volatile("" : "+v"(gp), "+v"(ap));
            gvs[h][j] = gp[row_off(j & 7)];
            avs[h][j] = ap[row_off(j & 7)];
          }
        }
        if (h == 0 && kb >= NLD && kb < NLD + 4) stage_load_part(next, kb - NLD, 4);
        if (h == 1) {
          const int slot = kb == 0 ? 0 : (kb >= S0 ? 1 + (kb - S0) : NSC);
#pragma unroll
          for (int i = 0; i < CPS; ++i)
            if (slot * CPS + i < NSC) stage_store_chunk(buf ^ 1, slot * CPS + i);
        }
        MTRSSM_SGB(0x008, 1);
        MTRSSM_SGB(0x002, 6);
        MTRSSM_SGB(0x008, 1);
        MTRSSM_SGB(0x002, 6);
        MTRSSM_SGB(0x008, 1);
        MTRSSM_SGB(0x002, 6);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (h == 0) MTRSSM_RES_STAMP(1);
      if (KS == 2) {  // the two halves of K meet: each wave hands over the rows it does not finish
        float* mine = red + ((size_t)((pgi * NCT + ct) * 2 + kh) * 8) * 64 + lane;
        const float* theirs = red + ((size_t)((pgi * NCT + ct) * 2 + (kh ^ 1)) * 8) * 64 + lane;
#pragma unroll
        for (int j = 0; j < 8; ++j) mine[j * 64] = kh == 0 ? acc[8 + j] : acc[j];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          fin[j] = kh == 0 ? acc[j] : acc[8 + j];
          asm volatile("" : "+v"(fin[j]));  // architectural VGPRs: a store sourcing an AGPR stalls like the read does
        }
        lds_barrier();
#pragma unroll
        for (int j = 0; j < 8; ++j) part[j] = theirs[j * 64];
      } else {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          fin[j] = acc[j];
          asm volatile("" : "+v"(fin[j]));  // architectural VGPRs: a store sourcing an AGPR stalls like the read does
        }
      }
      if constexpr (FUSE) {
        // ---- the block's 1x1 on this unit: B = pieces of act(intermediate), straight from the accumulator copy
        bf16x8 mb[2][2];
#pragma unroll
        for (int kb2 = 0; kb2 < 2; ++kb2) {
          unsigned hi[4], lo[4];
#pragma unroll
          for (int e2 = 0; e2 < 4; ++e2) {
            unsigned short p0[2], p1[2];
            split_bf16<2>(act_mid_sel(fin[8 * kb2 + 2 * e2]), p0);
            split_bf16<2>(act_mid_sel(fin[8 * kb2 + 2 * e2 + 1]), p1);
            hi[e2] = (unsigned)p0[0] | ((unsigned)p1[0] << 16);
            lo[e2] = (unsigned)p0[1] | ((unsigned)p1[1] << 16);
          }
          mb[kb2][0] = __builtin_bit_cast(bf16x8, u32x4{hi[0], hi[1], hi[2], hi[3]});
          mb[kb2][1] = __builtin_bit_cast(bf16x8, u32x4{lo[0], lo[1], lo[2], lo[3]});
        }
        f32x16 acc2[NOT];
#pragma unroll
        for (int ot = 0; ot < NOT; ++ot) {
#pragma unroll
          for (int r = 0; r < 16; ++r) acc2[ot][r] = 0.f;
#pragma unroll
          for (int kb2 = 0; kb2 < 2; ++kb2) {
            const unsigned char* wf = w1l + ((size_t)((ct * NF + ot * 2 + kb2) * 2) * 64 + lane) * 16;
            const bf16x8 a_hi = *reinterpret_cast<const bf16x8*>(wf);
            const bf16x8 a_lo = *reinterpret_cast<const bf16x8*>(wf + 64 * 16);
            acc2[ot] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi, mb[kb2][1], acc2[ot], 0, 0, 0);
            acc2[ot] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_lo, mb[kb2][0], acc2[ot], 0, 0, 0);
            acc2[ot] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi, mb[kb2][0], acc2[ot], 0, 0, 0);
          }
        }
        lds_barrier();  // the previous unit's partial sums have been read by everyone
        if constexpr (NCT == 2) {
          // two waves per frame, two output tiles: each keeps the tile it finishes and hands the other one over
          static_assert(NCT != 2 || NOT == 2, "one finished tile per wave");
          float* mine = red2 + (size_t)wave * 16 * 64 + lane;
#pragma unroll
          for (int r = 0; r < 16; ++r) mine[r * 64] = ct ? acc2[0][r] : acc2[NOT - 1][r];
          lds_barrier();
          const float* theirs = red2 + (size_t)(pgi * 2 + (ct ^ 1)) * 16 * 64 + lane;
#pragma unroll
          for (int j = 0; j < 16; ++j) {
            const float keep = ct ? acc2[NOT - 1][j] : acc2[0][j];
            out2[o2base + row_off(j)] = (keep + theirs[j * 64]) + (b1v[j] + xs[j]);
          }
        } else {
          float* mine = red2 + (size_t)wave * (NOT * 16) * 64 + lane;
#pragma unroll
          for (int ot = 0; ot < NOT; ++ot)
#pragma unroll
            for (int r = 0; r < 16; ++r) mine[(ot * 16 + r) * 64] = acc2[ot][r];
          lds_barrier();
          const float* theirs = red2 + ((size_t)(pgi * NCT) * (NOT * 16) + ct * RF) * 64 + lane;
#pragma unroll
          for (int j = 0; j < RF; ++j) {
            float v = b1v[j] + xs[j];
#pragma unroll
            for (int w = 0; w < NCT; ++w) v += theirs[((size_t)w * (NOT * 16) + j) * 64];
            out2[o2base + row_off(j)] = v;
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if (h == 0) MTRSSM_RES_STAMP(2);
      ob_prev = ob_cur;
    }
    MTRSSM_RES_STAMP(4);
    lds_barrier();  // the next image is complete; everyone is done reading this one (LDS only: a __syncthreads would also
                    // wait for this tile's stores and the next tile's frames in flight)
    MTRSSM_RES_STAMP(5);
  }
  // ---- drain: the last unit's epilogue (accumulator 1)
#pragma unroll
  for (int j = 0; j < ROWS; ++j) out[ob_prev + row_off(j) + (KS == 2 ? (unsigned)(kh * 16 * 64) : 0u)] = epi_row(1, j);
}


// ------------------------------------------------------------------------------------------------
// ConvTranspose2d k = 4, s = 2, p = 1 forward of the decoders (64 -> 32 on 64-pixel planes, 32 -> 16 on 256-pixel planes), ALL
// FOUR output parity classes in one pass over the source.
//
// The general path runs one gather launch per parity class (conv.py: _conv_transposed_gather): every class re-reads the whole
// source and writes a quarter of the output -- four ~100-us launches per layer for traffic worth ~60 us.  Here the
// structure of conv3x3_resident_kernel is reused with the four WAVES of a workgroup as the four parity classes: a frame's
// source is staged once into the haloed channel-innermost image, wave q keeps the 2 x 2-tap sub-kernel of class q (both bf16
// pieces, 32 output-channel rows x 4 taps x CIN) in registers and walks the frame's PLANE / 32 units of its class's output
// sub-grid, storing to rows / columns 2 j + q of the output plane.  The previous unit's rows are stored, the next frame is
// requested and converted between the MFMAs as there.
// ------------------------------------------------------------------------------------------------
struct QuadProblem {
  MtrssmConvGeom g[4];          // the parity classes' geometries (KH = KW = 2, TS = -1, OS = 2, QY / QX, OFFY / OFFX), same source
  const unsigned short* wq[4];  // their packed sub-kernels [pieces][CoutPad = 32][4 taps][Cpad = CIN]
  const float* src;
  const float* bias;
  const float* actgrad;         // EPI: out *= act'(actgrad[same index as out]) (backward-data of a strided conv), else NULL
  float* out;
  int nx;                       // workgroups of the launch that work on this tensor
};

template <int CIN, int PLANE>
__host__ __device__ constexpr size_t quad_lds_bytes() {
  return (size_t)4 * (PLANE == 64 ? 108 : 340) * res_row_bytes(CIN);  // haloed positions: 10x10 / 18x6, 18x18 / 34x10
}

// 32 -> 16 on 64-position frames: two workgroups per CU (registers and images fit twice: 80 -> 75 us; the 64 -> 32 instance
// at half its registers spills and runs 118 -> 123 us)
__host__ __device__ constexpr int quad_wgs(int cin, int plane) { return plane == 64 && cin <= 32 ? 2 : 1; }
template <int CIN, int COUT, int PLANE, bool EPI>
__global__ __launch_bounds__(kResThreads, quad_wgs(CIN, PLANE)) void convt_quad_resident_kernel(const QuadProblem qa, const QuadProblem qb) {
  constexpr int CB = CIN / 16, KB = 4 * CB;
  constexpr int RB = res_row_bytes(CIN);
  constexpr int NPOS = PLANE == 64 ? 108 : 340;
  constexpr int IMG = NPOS * RB;           // one piece of one buffer (one frame)
  constexpr int OCT = CIN / 8;
  constexpr int UN = PLANE / 32;           // 32-pixel units per frame and parity class
  constexpr int NIT = OCT * (PLANE / 64) / 4;  // staging items (8 channels x 64 pixels) per wave
  constexpr int NSC = NIT * 4;             // staging chunks (2 channels of an item)
  constexpr int ROWS = COUT / 2;           // accumulator rows that hold output channels (16 of 16, or 8 when COUT = 16)
  constexpr int RPE = 2, NES = ROWS / RPE; // epilogue rows per k-block, k-blocks [1, 1 + NES)
  constexpr int SPU = KB - NES;            // staging slots per unit: k-block 0 and [1 + NES, KB)
  constexpr int NLD = EPI ? 2 : 0;         // k-blocks [0, NLD) request the unit's act' operand (ROWS / NLD rows each)
  static_assert(UN >= 2 && 1 + NES <= KB && (UN / 2) * SPU >= NSC && NIT >= 1 && NLD <= KB, "schedule");
  const bool second = blockIdx.x >= (unsigned)qa.nx;  // workgroup-uniform
  const QuadProblem& Q = second ? qb : qa;
  const int wg = second ? (int)blockIdx.x - qa.nx : (int)blockIdx.x, nwg = Q.nx;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kl = lane >> 5, il = lane & 31;
  const MtrssmConvGeom g = Q.g[wave];  // this wave's parity class
  const unsigned short* __restrict__ wq = Q.wq[wave];
  const float* __restrict__ src = Q.src;
  const float* __restrict__ bias = Q.bias;
  const float* __restrict__ actgrad = Q.actgrad;  // host: non-null exactly in the EPI instantiation
  float* __restrict__ out = Q.out;
  const int ntiles = g.N;  // one frame per tile
  if (wg >= ntiles) return;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  unsigned char* patch = lds_raw;  // [2 buffers][2 pieces][IMG]

  bf16x8 a[KB][2];
  {
    const size_t piece = (size_t)g.CoutPad * 4 * g.Cpad;  // host: CoutPad == 32, Cpad == CIN
    const unsigned short* wrow = wq + (size_t)il * 4 * CIN + 8 * kl;
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      const int t = kb / CB, cb = kb % CB;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        a[kb][s] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(wrow + s * piece + t * CIN + cb * 16));
        asm volatile("" : "+a"(a[kb][s]));
      }
    }
  }
  for (int o = tid * 16; o < 4 * IMG; o += kResThreads * 16) *reinterpret_cast<u32x4*>(patch + o) = u32x4{0u, 0u, 0u, 0u};
  const int cbase = 4 * kl;
  float bv[16];
  {
    const float* bsafe = bias ? bias : src;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = cbase + (r & 3) + 8 * (r >> 2);
      const float b = bsafe[co < COUT ? co : 0];
      bv[r] = (bias && co < COUT) ? b : 0.f;
    }
  }
  const int Ws = g.Ws, pw = Ws + 2, wsh = 31 - __builtin_clz(Ws);  // host: Ws a power of two, Hs * Ws == PLANE
  const int plane_o = g.Ho * g.Wo;
  const bool act_elu = g.act == MTRSSM_ACT_ELU, act_relu = g.act == MTRSSM_ACT_RELU, pre = g.pre_act != 0;
  auto act_fwd_sel = [&](float x) {
    float e = __expf(x) - 1.f;
    asm volatile("" : "+v"(e));
    const float neg = act_elu ? e : (act_relu ? 0.f : x);
    return (x > 0.f || !pre) ? x : neg;
  };
  float pv[NIT][8];
  auto stage_load_part = [&](int tile, int part, int nparts) {
    constexpr int TOT = NIT * 8;
    const int per = TOT / nparts;
#pragma unroll
    for (int j = 0; j < TOT; ++j) {
      if (j / per != part) continue;
      const int it = j / 8, u = j % 8;
      const int q = wave * NIT + it, o = q % OCT, blk = q / OCT;
      pv[it][u] = src[((size_t)tile * CIN + o * 8 + u) * PLANE + blk * 64 + lane];
    }
  };
  auto stage_store_chunk = [&](int buf, int c) {
    const int it = c / 4, u2 = c % 4;
    const int q = wave * NIT + it, o = q % OCT, blk = q / OCT;
    const int p = blk * 64 + lane;
    const int spos = ((p >> wsh) + 1) * pw + (p & (Ws - 1)) + 1;
    unsigned short p0[2], p1[2];
    split_bf16<2>(act_fwd_sel(pv[it][2 * u2]), p0);
    split_bf16<2>(act_fwd_sel(pv[it][2 * u2 + 1]), p1);
    unsigned char* dst = patch + (size_t)buf * 2 * IMG + spos * RB + o * 16 + u2 * 4;
    *reinterpret_cast<unsigned*>(dst) = (unsigned)p0[0] | ((unsigned)p1[0] << 16);
    *reinterpret_cast<unsigned*>(dst + IMG) = (unsigned)p0[1] | ((unsigned)p1[1] << 16);
  };
  stage_load_part(wg, 0, 1);
  __syncthreads();
#pragma unroll
  for (int c = 0; c < NSC; ++c) stage_store_chunk(0, c);
  __syncthreads();

  // this lane's output pixel of unit u: sub-grid pixel j = 32 u + il -> (jy, jx); source position of tap (0, 0); output offset
  auto unit_pos = [&](int u) {
    const int j = u * 32 + il, jy = j >> wsh, jx = j & (Ws - 1);
    return ((jy + g.OFFY + 1) * pw + (jx + g.OFFX + 1)) * RB + kl * 16;
  };
  auto unit_out = [&](int tile, int u) {
    const int j = u * 32 + il, jy = j >> wsh, jx = j & (Ws - 1);
    return (unsigned)tile * (unsigned)(COUT * plane_o) + (unsigned)(cbase * plane_o) + (unsigned)((jy * 2 + g.QY) * g.Wo + jx * 2 + g.QX);
  };
  f32x16 acc;
  float fin[ROWS], gvs[2][ROWS];  // gvs[u & 1]: the act' operand of unit u, requested in its first k-blocks, used one unit later
#pragma unroll
  for (int r = 0; r < ROWS; ++r) fin[r] = gvs[0][r] = gvs[1][r] = 0.f;
  auto act_grad_sel = [&](float x) {
    float e = __expf(x);
    asm volatile("" : "+v"(e));
    const float neg = act_elu ? e : (act_relu ? 0.f : 1.f);
    return x > 0.f ? 1.f : neg;
  };
  // (the first unit's epilogue slots store zeros where the same wave stores unit 1 of its first frame one unit later)
  unsigned ob_prev = unit_out(wg, 1);
  struct Frag { bf16x8 b[2]; };
  int it_no = 0;
  for (int tile = wg; tile < ntiles; tile += nwg, ++it_no) {
    const int buf = it_no & 1;
    const unsigned char* img = patch + (size_t)buf * 2 * IMG;
    const int next = tile + nwg < ntiles ? tile + nwg : tile;
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const unsigned ob_cur = unit_out(tile, u);
      const unsigned ibase = (unsigned)(uintptr_t)img + (unsigned)unit_pos(u);
      auto read_frag = [&](Frag& f, int kb) {
        const int t = kb / CB, cb = kb % CB;
        const int toff = -((t / 2) * pw + (t % 2)) * RB;  // TS = -1: tap (ty, tx) reads (jy - ty + OFFY, jx - tx + OFFX)
        res_read_pair<IMG>(f.b[0], f.b[1], ibase + (unsigned)toff, cb);
      };
      Frag fr[3];
      read_frag(fr[0], 0);
      read_frag(fr[1], 1);
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = bv[r];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) {
        Frag& cur = fr[kb % 3];
        if (kb + 2 < KB) {
          read_frag(fr[(kb + 2) % 3], kb + 2);
          res_wait<4>(cur.b[0], cur.b[1]);
        } else if (kb + 1 < KB) {
          res_wait<2>(cur.b[0], cur.b[1]);
        } else {
          res_wait<0>(cur.b[0], cur.b[1]);
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[kb][0], cur.b[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[kb][1], cur.b[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[kb][0], cur.b[0], acc, 0, 0, 0);
        if (kb >= 1 && kb < 1 + NES) {  // rows of the previous unit
#pragma unroll
          for (int i = 0; i < RPE; ++i) {
            const int j = (kb - 1) * RPE + i;
            float v = fin[j];
            if (EPI) v *= act_grad_sel(gvs[(u + 1) & 1][j]);
            out[ob_prev + (unsigned)(((j & 3) + 8 * (j >> 2)) * plane_o)] = v;
          }
        }
        if (EPI && kb < NLD) {
          constexpr int RPL = ROWS / 2;
#pragma unroll
          for (int i = 0; i < RPL; ++i) {
            const int j = kb * RPL + i;
            gvs[u & 1][j] = actgrad[ob_cur + (unsigned)(((j & 3) + 8 * (j >> 2)) * plane_o)];
          }
        }
        if (u == 0 && kb < 4) stage_load_part(next, kb, 4);
        if (u >= UN / 2) {
          const int local = kb == 0 ? 0 : (kb >= 1 + NES ? 1 + (kb - 1 - NES) : SPU);
          const int slot = (u - UN / 2) * SPU + local;
          if (local < SPU && slot < NSC) stage_store_chunk(buf ^ 1, slot);
        }
        MTRSSM_SGB(0x008, 1);
        MTRSSM_SGB(0x002, 6);
        MTRSSM_SGB(0x008, 1);
        MTRSSM_SGB(0x002, 6);
        MTRSSM_SGB(0x008, 1);
        MTRSSM_SGB(0x002, 6);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int j = 0; j < ROWS; ++j) {
        fin[j] = acc[j];
        asm volatile("" : "+v"(fin[j]));
      }
      __builtin_amdgcn_sched_barrier(0);
      ob_prev = ob_cur;
    }
    lds_barrier();
  }
#pragma unroll
  for (int j = 0; j < ROWS; ++j) {
    float v = fin[j];
    if (EPI) v *= act_grad_sel(gvs[(UN - 1) & 1][j]);
    out[ob_prev + (unsigned)(((j & 3) + 8 * (j >> 2)) * plane_o)] = v;
  }
}

// ------------------------------------------------------------------------------------------------
// Streaming 1x1 kernel for the residual stacks' second conv (forward with bias + skip, backward-data with act'(h)).
//
// K = Cin is 64 or 128: 24..48 MFMAs per 32 pixels against 16..32 KB of HBM traffic -- a bandwidth problem
// (conv1x1_split_kernel: 1.9 TB/s through two LDS images, two barriers per 128 pixels).  Here nothing meets in LDS and
// no wave waits for another: lanes = 32 consecutive pixels of one plane (a 128-byte segment per channel), the lane's
// half kl of every 16-channel k-block = 8 loads that ARE the MFMA B operand once activated and split, the whole weight
// matrix sits in AGPRs as A operands (<= 128 registers), and a wave walks its own tiles with the next tile's pixels
// and this tile's epilogue operand in flight under the conversions and MFMAs of the current one.  One wave per SIMD,
// 96 KB of requests in flight per CU.
// ------------------------------------------------------------------------------------------------
template <int CIN, int COUT, bool FWD>  // FWD: out = acc + bias + add_in;  !FWD: out = acc * act'(actgrad_in)
__global__ __launch_bounds__(kResThreads, 1) void conv1x1_stream_kernel(const GatherProblem pa, const GatherProblem pb) {
  constexpr int CT = COUT / 32, CB = CIN / 16;
  const bool second = blockIdx.x >= (unsigned)pa.nx;  // workgroup-uniform
  const GatherProblem P = second ? pb : pa;
  const MtrssmConvGeom g = P.g;
  const float* __restrict__ src = P.src;
  const unsigned short* __restrict__ wq = P.wq;
  const float* __restrict__ bias = P.bias;
  const float* __restrict__ opnd = FWD ? P.add_in : P.actgrad_in;  // host: the one operand this variant takes is present
  float* __restrict__ out = P.out;
  const int wg = second ? (int)blockIdx.x - pa.nx : (int)blockIdx.x, nwg = P.nx;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int kl = lane >> 5, il = lane & 31;
  const int plane = g.Hq * g.Wq;           // host: plane % 32 == 0
  const int tpf = plane >> 5;              // 32-pixel tiles per frame
  const int ntiles = g.N * tpf;
  const int nwv = nwg * 4;
  int t = wg * 4 + wave;
  if (t >= ntiles) return;  // wave-uniform; no barrier in this kernel

  bf16x8 a[CT][CB][2];
  {
    const size_t piece = (size_t)g.CoutPad * g.Cpad;  // host: Cpad == CIN, CoutPad == COUT
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          a[ct][cb][s] = __builtin_bit_cast(
              bf16x8, *reinterpret_cast<const u32x4*>(wq + s * piece + (size_t)(ct * 32 + il) * CIN + cb * 16 + 8 * kl));
          asm volatile("" : "+a"(a[ct][cb][s]));  // AGPRs: the VGPR half belongs to the streams (conv3x3_resident_kernel)
        }
  }
  float bv[CT][16];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
    for (int r = 0; r < 16; ++r) bv[ct][r] = (FWD && bias) ? bias[ct * 32 + 4 * kl + (r & 3) + 8 * (r >> 2)] : 0.f;
  const bool act_elu = g.act == MTRSSM_ACT_ELU, act_relu = g.act == MTRSSM_ACT_RELU, pre = g.pre_act != 0;

  auto in_ptr = [&](int tile) {  // this lane's pixel of the tile, channel 8 * kl
    const int n = tile / tpf, hw = (tile - n * tpf) * 32 + il;
    return src + ((size_t)n * CIN + 8 * kl) * plane + hw;
  };
  auto io_off = [&](int tile) {  // element offset of this lane's pixel, output channel 4 * kl
    const int n = tile / tpf, hw = (tile - n * tpf) * 32 + il;
    return ((size_t)n * COUT + 4 * kl) * plane + hw;
  };
  float pv[CB][8], nx[CB][8];
  {
    const float* p = in_ptr(t);
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
      for (int u = 0; u < 8; ++u) pv[cb][u] = p[(size_t)(cb * 16 + u) * plane];
  }
  for (; t < ntiles; t += nwv) {
    const int tn = t + nwv < ntiles ? t + nwv : t;
    {  // the next tile's pixels
      const float* p = in_ptr(tn);
#pragma unroll
      for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int u = 0; u < 8; ++u) nx[cb][u] = p[(size_t)(cb * 16 + u) * plane];
    }
    const size_t oo = io_off(t);
    float op[CT][16];  // this tile's epilogue operand
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int r = 0; r < 16; ++r) op[ct][r] = opnd[oo + (size_t)(ct * 32 + (r & 3) + 8 * (r >> 2)) * plane];
    f32x16 acc[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ct][r] = bv[ct][r];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) {
      u16x8 qh, ql;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const float x = pv[cb][u];
        float e = __expf(x) - 1.f;
        const float neg = act_elu ? e : (act_relu ? 0.f : x);
        const float y = (x > 0.f || !pre) ? x : neg;
        unsigned short p2[2];
        split_bf16<2>(y, p2);
        qh[u] = p2[0];
        ql[u] = p2[1];
      }
      const bf16x8 bh = __builtin_bit_cast(bf16x8, qh), bl = __builtin_bit_cast(bf16x8, ql);
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ct][cb][0], bl, acc[ct], 0, 0, 0);
        acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ct][cb][1], bh, acc[ct], 0, 0, 0);
        acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ct][cb][0], bh, acc[ct], 0, 0, 0);
      }
    }
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v = acc[ct][r];
        if (FWD) {
          v += op[ct][r];
        } else {
          const float x = op[ct][r];
          const float e = __expf(x);
          const float neg = act_elu ? e : (act_relu ? 0.f : 1.f);
          v *= x > 0.f ? 1.f : neg;
        }
        out[oo + (size_t)(ct * 32 + (r & 3) + 8 * (r >> 2)) * plane] = v;
      }
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
      for (int u = 0; u < 8; ++u) pv[cb][u] = nx[cb][u];
  }
}

}  // namespace mtrssm
