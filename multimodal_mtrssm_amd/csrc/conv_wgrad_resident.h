// Staged weight-gradient kernels (included by conv.hip after conv_split.h): a workgroup per CU owns a run of frames, loads
// every frame of both operands once, converts it once, keeps it in LDS for the MFMAs and leaves ONE partial set for a
// reduce kernel.  In this file: the 3x3 / stride 1 layers of the residual stacks (first, with the reasoning), their 1x1
// layers, the encoders' three 3x3 / stride-2 convolutions and the decoders' three k = 4 / stride-2 transposed convolutions.
//
// Weight gradient of the 3x3 / stride 1 / pad 1 layers of the residual stacks on 64-pixel planes (8x8 vision, 16x4 audio)
// with BOTH operands staged once per frame.
//
//   dW[co][ty][tx][ci] = sum over frames n and pixels (y, x) of  a[n][co][y][x] * act(src)[n][ci][y + ty - 1][x + tx - 1]
//
// The register-direct kernel (conv_split.h: conv3x3_weight_grad_split_kernel) lets every lane load, activate and split its
// own operands: each src element is fetched 2 (co tiles) x 3 (window rows) times by 16-byte loads that touch 32 cache lines
// per instruction and is converted as often, each `a` element twice -- the kernel is VALU-bound (about 75 % of its issue
// slots are conversions), and ~60 us of every launch pass after its loop (9.4 M fp32 atomics of the tiles, and -- the larger
// part, see profiles/round2_notes.md -- the bias sums every workgroup adds to the same 64 addresses).  Here a workgroup
// owns a run of frames and
//   * loads every frame of `src` and of `a` ONCE, coalesced (consecutive lanes = consecutive 16 bytes), activates / splits
//     it once and writes it to LDS: `a` as a plain [co][pixel] bf16 image, src as a channel-major image with a zero halo
//     row between consecutive planes: a row tap (ty) is a 16-byte read at row offset ty, the column taps (tx) are funnel
//     shifts of that fragment in registers -- a lane's 8 pixels are whole plane rows, so the shifted row needs no
//     neighbour (three column-shifted copies in LDS instead made the LDS pipe the bottleneck: 2300 of its cycles per
//     frame against 3456 of MFMA work, and the two did not overlap);
//   * double-buffers those images: frame n + 1 is converted (VALU + ds_write_b64) while the MFMAs of frame n run, its
//     raw values requested a whole frame ahead, so no HBM latency is exposed;
//   * gives each wave nine accumulator tiles chosen so that it reads two image rows instead of three: the two waves of a
//     ci tile hold {co tile 0: taps 0-4, co tile 1: taps 0-3} and {co tile 1: taps 4-8, co tile 0: taps 5-8};
//   * writes its partial tiles with plain 16-byte stores, in accumulator order, to a scratch slice of its own (`part`),
//     summed and scattered into dwp by wgrad_reduce_partials_kernel right behind it in the stream: no atomics, run-to-run
//     deterministic.  With part == NULL (no scratch: first call inside a stream capture, or MTRSSM_WGRAD_PARTIALS=0) the
//     tiles leave by fp32 atomics in the dwp layout [co][tap][Cpad], as in the other kernels.
// One barrier per frame.
//
// LDS images: src piece [C][H + 1 rows][W] bf16 -- row 0 of every plane is the zero halo, shared with the plane above (the
// image ends with one more zero row); `a` piece [64][64 pixels + 8].  Both pitches are an odd number of 16-byte
// (W = 4: 8-byte, reads are two ds_read_b64 halves) slots, so the 32 channels one read covers fall on different banks.
#pragma once

namespace mtrssm {

template <int W>
__host__ __device__ constexpr int wgres_pitch() { return (64 / W + 1) * 2 * W; }  // 144 (W = 8), 136 (W = 4)
constexpr int kWgresAPitch = 144;  // bytes per co row of the `a` image: 64 bf16 + 16
template <int SPLIT, int C, int W>
__host__ __device__ constexpr int wgres_xbuf_bytes() { return SPLIT * C * wgres_pitch<W>() + 16; }
template <int SPLIT>
__host__ __device__ constexpr int wgres_abuf_bytes() { return SPLIT * 64 * kWgresAPitch; }
template <int SPLIT, int C, int W>
__host__ __device__ constexpr int wgres_lds_bytes() { return 2 * (wgres_xbuf_bytes<SPLIT, C, W>() + wgres_abuf_bytes<SPLIT>()); }
// floats of one workgroup's partial tile set: its waves' nine 32x32 tiles
__host__ __device__ constexpr int wgres_tile_floats(int c) { return 2 * (c / 32) * 9 * 1024; }
// ... followed by the workgroup's 64 bias-gradient sums (256 workgroups adding to the same 64 addresses took 10-25 us)
__host__ __device__ constexpr int wgres_set_floats(int c) { return wgres_tile_floats(c) + 64; }
constexpr int kWg1x1SetFloats = 8 * 1024 + 64;  // conv1x1_wgrad_staged_kernel: 8 waves x one tile, + 64 bias sums

using wg_bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
using wg_f32x2 = __attribute__((ext_vector_type(2))) float;
using wg_f32x4 = __attribute__((ext_vector_type(4))) float;  // a register quadruple inline asm can name (float4 is a struct)
// two values -> packed bf16 pieces (round to nearest even, as split_bf16): d[s] = piece s of (f0 low half, f1 high half)
template <int SPLIT>
__device__ __forceinline__ void wg_split_pair(const float f0, const float f1, unsigned (&d)[SPLIT]) {
  d[0] = __builtin_bit_cast(unsigned, __builtin_convertvector(wg_f32x2{f0, f1}, wg_bf16x2));
  if constexpr (SPLIT >= 2) {
    const float r0 = f0 - __builtin_bit_cast(float, d[0] << 16), r1 = f1 - __builtin_bit_cast(float, d[0] & 0xffff0000u);  // exact
    d[1] = __builtin_bit_cast(unsigned, __builtin_convertvector(wg_f32x2{r0, r1}, wg_bf16x2));
  }
}

template <int SPLIT, int C, int W>
__global__ __launch_bounds__(64 * 2 * (C / 32), 1) void conv3x3_wgrad_resident_kernel(
    const MtrssmConvGeom g, const float* __restrict__ a, const float* __restrict__ src, float* __restrict__ dwp,
    float* __restrict__ part, float* __restrict__ dbias, const int frames_per_wg) {
  static_assert(SPLIT == 1 || SPLIT == 2, "one or two bf16 pieces");
  constexpr int NW = 2 * (C / 32), NT = 64 * NW;
  constexpr int RB = 2 * W;  // bytes per image row
  constexpr int PITCH = wgres_pitch<W>(), COPYB = C * PITCH, XBUFB = wgres_xbuf_bytes<SPLIT, C, W>();
  constexpr int APIECEB = 64 * kWgresAPitch, ABUFB = wgres_abuf_bytes<SPLIT>();
  constexpr int XI = 4, AI = 1024 / NT;  // float4 items per thread and frame: src (C * 16 = 4 NT), a (64 co * 16)
  extern __shared__ __attribute__((aligned(16))) unsigned char wgres_lds[];
  unsigned char* const lds = wgres_lds;           // [src image 0][src image 1][a image 0][a image 1]
  unsigned char* const lds_a = wgres_lds + 2 * XBUFB;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int il = lane & 31, kl = lane >> 5;
  const int tci = wave >> 1, half = wave & 1;
  const int n0 = blockIdx.x * frames_per_wg;
  const int n1 = n0 + frames_per_wg < g.N ? n0 + frames_per_wg : g.N;
  if (n0 >= n1) return;  // whole workgroup
  const int nlast = n1 - 1;
  const int cob = blockIdx.y * 64;
  // development aid (tools/wgrad_probe.py): cycle stamps of workgroup 0 and of the middle workgroup
  unsigned long long* const prof = (tid == 0 && blockIdx.y == 0 && g_res_prof) ? (blockIdx.x == 0 ? g_res_prof + 32 : (blockIdx.x == gridDim.x / 2 ? g_res_prof + 40 : nullptr)) : nullptr;
  if (prof) prof[0] = __builtin_readcyclecounter();

  // activation switches as lane-uniform selects (no branch inside the MFMA loop: a branch ends a scheduling region)
  const bool act_elu = g.act == MTRSSM_ACT_ELU, act_relu = g.act == MTRSSM_ACT_RELU, pre = g.pre_act != 0;
  auto act_sel = [&](float x) __attribute__((always_inline)) {
    float e = __expf(x) - 1.f;
    asm volatile("" : "+v"(e));  // computed unconditionally: the compiler would branch around the exponential
    const float neg = act_elu ? e : (act_relu ? 0.f : x);
    return (x > 0.f || !pre) ? x : neg;
  };

  // ---- zero everything once: the halo rows are never written again
  for (int o = tid * 16; o < wgres_lds_bytes<SPLIT, C, W>(); o += NT * 16) *reinterpret_cast<uint4*>(lds + o) = make_uint4(0u, 0u, 0u, 0u);

  // ---- staging: item j of a thread = float4 number tid + NT * j of the frame (of the workgroup's 64 co of it for `a`):
  // 4 consecutive pixels of channel (tid >> 4) + j * NT / 16
  const float4* const xsrc = reinterpret_cast<const float4*>(src) + tid;
  const float4* const asrc = reinterpret_cast<const float4*>(a + (size_t)cob * 64) + tid;
  const size_t xfr = (size_t)C * 16, afr = (size_t)g.Cout * 16;  // float4 per frame
  const int xi = tid & 15;
  const int xrow = (xi * 4) / W, xcol = (xi * 4) % W;
  const unsigned wofs = (unsigned)((tid >> 4) * PITCH + (xrow + 1) * RB + xcol * 2);
  const unsigned wofs_a = (unsigned)((tid >> 4) * kWgresAPitch + xi * 8);
  auto stage_x = [&](const float4 v, const int j, const unsigned bufoff) __attribute__((always_inline)) {
    unsigned d0[SPLIT], d1[SPLIT];
    wg_split_pair<SPLIT>(act_sel(v.x), act_sel(v.y), d0);
    wg_split_pair<SPLIT>(act_sel(v.z), act_sel(v.w), d1);
#pragma unroll
    for (int p = 0; p < SPLIT; ++p)
      *reinterpret_cast<uint2*>(lds + bufoff + wofs + (unsigned)(j * (NT / 16) * PITCH) + (unsigned)(p * COPYB)) = make_uint2(d0[p], d1[p]);
  };
  float bsum[AI];  // bias gradient: this thread's share of channel (tid >> 4) + j * NT / 16
#pragma unroll
  for (int j = 0; j < AI; ++j) bsum[j] = 0.f;
  auto stage_a = [&](const float4 v, const int j, const unsigned bufoff, const float keep) __attribute__((always_inline)) {
    bsum[j] += keep * ((v.x + v.y) + (v.z + v.w));
    unsigned d0[SPLIT], d1[SPLIT];
    wg_split_pair<SPLIT>(v.x, v.y, d0);
    wg_split_pair<SPLIT>(v.z, v.w, d1);
#pragma unroll
    for (int p = 0; p < SPLIT; ++p)
      *reinterpret_cast<uint2*>(lds_a + bufoff + wofs_a + (unsigned)(j * (NT / 16) * kWgresAPitch) + (unsigned)(p * APIECEB)) = make_uint2(d0[p], d1[p]);
  };

  // ---- the wave's tiles.  Its B operands come from TWO image rows per k-step, the outer one (ty = 0 for half 0, ty = 2
  // for half 1) and the centre one (ty = 1); the column taps are funnel shifts of those rows in registers (a lane's 8
  // pixels are whole plane rows, so a shifted row needs no neighbour).  Five B slots:
  //   0: outer row, tx = 0    1: outer, tx = 1    2: outer, tx = 2    3: centre, tx = (half 0: 0, half 1: 2)    4: centre, tx = 1
  // nine accumulators = (coA, slots 0-4) and (coB, slots 0-3); coA = co tile `half`: half 0 holds {co 0: taps 0-4, co 1: taps
  // 0-3}, half 1 {co 1: taps 6 7 8 5 4, co 0: taps 6 7 8 5}.
  const int coA = half, coB = half ^ 1;  // 32-channel tile of the co group
  const int tyo = half ? 2 : 0;
  const unsigned lane_b = (unsigned)((tci * 32 + il) * PITCH + kl * 16);
  const unsigned lane_a0 = (unsigned)((coA * 32 + il) * kWgresAPitch + kl * 16), lane_a1 = (unsigned)((coB * 32 + il) * kWgresAPitch + kl * 16);
  // W = 8: a fragment is one 16-byte row, rows ty apart are 16 bytes apart.  W = 4: a fragment is two 8-byte rows starting
  // at row ty: the outer one is 16-byte aligned, the centre one straddles -- it is put together from the outer fragment and
  // one more 8-byte row (`ext`).
  const unsigned outer_off = lane_b + (unsigned)(tyo * RB);
  const unsigned ext_off = lane_b + (W == 8 ? 16u : (half ? 8u : 16u));

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  u32x4 qa[2][2][SPLIT];  // [set][co tile][piece]
  u32x4 ro[2][SPLIT], rc[2][SPLIT];  // [set][piece]: outer row fragment; centre row fragment (W = 4: .xy = the extra row)
  auto operands = [&](const int q, const int set, const unsigned xoff, const unsigned aoff) __attribute__((always_inline)) {
#pragma unroll
    for (int p = 0; p < SPLIT; ++p) {
      qa[set][0][p] = *reinterpret_cast<const u32x4*>(lds_a + aoff + lane_a0 + (unsigned)(p * APIECEB + q * 32));
      qa[set][1][p] = *reinterpret_cast<const u32x4*>(lds_a + aoff + lane_a1 + (unsigned)(p * APIECEB + q * 32));
      ro[set][p] = *reinterpret_cast<const u32x4*>(lds + xoff + outer_off + (unsigned)(p * COPYB + q * 32));
      if (W == 8) {
        rc[set][p] = *reinterpret_cast<const u32x4*>(lds + xoff + ext_off + (unsigned)(p * COPYB + q * 32));
      } else {
        const uint2 e = *reinterpret_cast<const uint2*>(lds + xoff + ext_off + (unsigned)(p * COPYB + q * 32));
        rc[set][p] = u32x4{e.x, e.y, 0u, 0u};
      }
    }
  };
  auto ab = [](const unsigned hi, const unsigned lo) __attribute__((always_inline)) { return __builtin_amdgcn_alignbit(hi, lo, 16); };
  auto shr1 = [&](const u32x4 f) __attribute__((always_inline)) {  // pixel x <- x - 1, zero into every row's first pixel
    return W == 8 ? u32x4{f.x << 16, ab(f.y, f.x), ab(f.z, f.y), ab(f.w, f.z)} : u32x4{f.x << 16, ab(f.y, f.x), f.z << 16, ab(f.w, f.z)};
  };
  auto shl1 = [&](const u32x4 f) __attribute__((always_inline)) {  // pixel x <- x + 1, zero into every row's last pixel
    return W == 8 ? u32x4{ab(f.y, f.x), ab(f.z, f.y), ab(f.w, f.z), f.w >> 16} : u32x4{ab(f.y, f.x), f.y >> 16, ab(f.w, f.z), f.w >> 16};
  };
  auto mfmas = [&](const int set) __attribute__((always_inline)) {
    u32x4 qb[5][SPLIT];
#pragma unroll
    for (int p = 0; p < SPLIT; ++p) {
      const u32x4 o = ro[set][p];
      u32x4 c = rc[set][p];
      if (W == 4) c = half ? u32x4{c.x, c.y, o.x, o.y} : u32x4{o.z, o.w, c.x, c.y};
      const u32x4 cl = shl1(c), cr = shr1(c);
      qb[0][p] = shr1(o);
      qb[1][p] = o;
      qb[2][p] = shl1(o);
      qb[3][p] = half ? cl : cr;
      qb[4][p] = c;
    }
    // product-major: consecutive MFMAs go to different accumulator tiles, so none waits for its predecessor
#pragma unroll
    for (int ord = SPLIT - 1; ord >= 0; --ord)
#pragma unroll
      for (int sa = 0; sa <= ord; ++sa)
#pragma unroll
        for (int j = 0; j < 9; ++j)
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, qa[set][j < 5 ? 0 : 1][sa]),
                                                           __builtin_bit_cast(bf16x8, qb[j < 5 ? j : j - 5][ord - sa]), acc[j], 0, 0, 0);
  };

  // ---- raw frames: two register sets.  Set s is requested early in a frame and staged under the NEXT one, so every use
  // is at least three k-steps behind its request, and the requests of a frame follow its first staged item: whatever
  // counter value the compiler waits for there, nothing younger than a frame is in flight.
  // The requests are inline asm: the compiler sinks its own loads to their uses, a frame later (and then waits for them
  // there).  As volatile asm they keep their place among the LDS operations; raw_wait is the matching counter wait.
  constexpr int NI = XI + AI;
  wg_f32x4 raw[2][NI];  // [set][x items, a items]
  auto raw_load = [&](const int rs, const int n) __attribute__((always_inline)) {
#pragma unroll
    for (int it = 0; it < NI; ++it) {
      const float4* const ptr = it < XI ? xsrc + (size_t)n * xfr + NT * it : asrc + (size_t)n * afr + NT * (it - XI);
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(raw[rs][it]) : "v"(ptr));
    }
  };
  auto raw_wait = [&](const int rs) __attribute__((always_inline)) {  // every request so far has landed
    if constexpr (NI == 8) {
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(raw[rs][0]), "+v"(raw[rs][1]), "+v"(raw[rs][2]), "+v"(raw[rs][3]), "+v"(raw[rs][4]), "+v"(raw[rs][5]), "+v"(raw[rs][6]), "+v"(raw[rs][7]));
    } else {
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(raw[rs][0]), "+v"(raw[rs][1]), "+v"(raw[rs][2]), "+v"(raw[rs][3]), "+v"(raw[rs][4]), "+v"(raw[rs][5]), "+v"(raw[rs][6]), "+v"(raw[rs][7]), "+v"(raw[rs][8]), "+v"(raw[rs][9]), "+v"(raw[rs][10]), "+v"(raw[rs][11]));
    }
  };
  auto stage = [&](const int rs, const int it, const unsigned xo, const unsigned ao, const float keep) __attribute__((always_inline)) {
    const float4 v = make_float4(raw[rs][it].x, raw[rs][it].y, raw[rs][it].z, raw[rs][it].w);
    if (it < XI) stage_x(v, it, xo); else stage_a(v, it - XI, ao, keep);
  };
  // staging order within a frame: x0 a0 x1 a1 x2 a2 x3 a3 [a4 ...]; item number of the k-th staged one
  auto item_of = [](const int k) __attribute__((always_inline)) { return k < 8 ? ((k & 1) ? XI + (k >> 1) : (k >> 1)) : XI + k - 4; };

  // ---- prologue: frame n0 staged into images 0, frame n0 + 1 requested into set 1
  raw_load(0, n0);
  raw_wait(0);
  raw_load(1, n0 + 1 < nlast ? n0 + 1 : nlast);
  lds_barrier();  // the zeroes are in place
#pragma unroll
  for (int k = 0; k < NI; ++k) stage(0, k, 0u, 0u, 1.f);
  lds_barrier();  // images 0 complete
  operands(0, 0, 0u, 0u);
  if (prof) prof[1] = __builtin_readcyclecounter();

  // One frame: par = parity of the frame within the workgroup's run (a literal at the call sites: image offsets and raw
  // sets are compile-time).  Images par hold frame n; raw set par ^ 1 holds frame n + 1, staged into images par ^ 1 under
  // k-steps 0-2; raw set par is free and receives frame n + 2.
  auto frame = [&](const int n, const int par) __attribute__((always_inline)) {
    const unsigned xoff = par ? (unsigned)XBUFB : 0u, xoth = (unsigned)XBUFB - xoff;
    const unsigned aoff = par ? (unsigned)ABUFB : 0u, aoth = (unsigned)ABUFB - aoff;
    const float keep = n + 1 < n1 ? 1.f : 0.f;  // the frame staged under the last one is a duplicate: keep it out of the bias sum
    const int nn2 = n + 2 < nlast ? n + 2 : nlast;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int set = q & 1;
      // operands of the NEXT k-step first: their LDS latency passes under this step's MFMAs
      if (q < 3) {
        operands(q + 1, set ^ 1, xoff, aoff);
      } else {
        lds_barrier();  // every wave has read these images and written the other ones
        operands(0, set ^ 1, xoth, aoth);
      }
      mfmas(set);
      if (q < 3) {
#pragma unroll
        for (int k = q * NI / 3; k < (q + 1) * NI / 3; ++k) {
          if (k == 0) raw_wait(par ^ 1);  // requested a frame ago, nothing younger in flight
          stage(par ^ 1, item_of(k), xoth, aoth, keep);
          if (k == 0) raw_load(par, nn2);
        }
      }
      // the fragment shifts, the staging and the next operands in the MFMAs' shadow: after every MFMA up to five VALU /
      // transcendental instructions, after every other one an LDS access (left to itself the scheduler issues the MFMAs in
      // runs: 5400 instead of 4600 cycles per frame, 6400 instead of 4900 on the 4-wide planes)
#pragma unroll
      for (int i = 0; i < 9 * (SPLIT == 2 ? 3 : 1); ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x402, 5, 0);
        if ((i & 1) == 0) __builtin_amdgcn_sched_group_barrier(0x080, 1, 0);
      }
    }
  };
#pragma unroll 1
  for (int n = n0; n < n1; n += 2) {
    frame(n, 0);
    if (n + 1 < n1) frame(n + 1, 1);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the clamped requests of the last frame
  if (prof) prof[2] = __builtin_readcyclecounter();

  if (part) {
    // accumulator order: float4 number ((wave * 9 + j) * 4 + r / 4) * 64 + lane of this workgroup's set -- every store
    // instruction writes 1 KiB in a row (wgrad_reduce_partials_kernel maps it back to dwp)
    float4* const ps = reinterpret_cast<float4*>(part) + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * (wgres_set_floats(C) / 4) + (size_t)wave * (9 * 4 * 64) + lane;
#pragma unroll
    for (int j = 0; j < 9; ++j)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) ps[(j * 4 + gq) * 64] = make_float4(acc[j][4 * gq], acc[j][4 * gq + 1], acc[j][4 * gq + 2], acc[j][4 * gq + 3]);
  } else {
    const int ci = tci * 32 + il;
#pragma unroll
    for (int j = 0; j < 9; ++j) {
      const int k = j < 5 ? j : j - 5;
      const int tap = k < 3 ? tyo * 3 + k : (k == 3 ? (half ? 5 : 3) : 4);
      const int cot = cob + (j < 5 ? coA : coB) * 32;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = cot + (r & 3) + 8 * (r >> 2) + 4 * kl;
        atomicAdd(&dwp[((size_t)row * 9 + tap) * g.Cpad + ci], acc[j][r]);
      }
    }
  }
  if (dbias != nullptr) {
#pragma unroll
    for (int j = 0; j < AI; ++j) {  // the 16 lanes of a DPP row share a channel
      float v = bsum[j];
      v += dpp_move<0xB1, 0xF>(0.f, v);   // quad_perm [1, 0, 3, 2]
      v += dpp_move<0x4E, 0xF>(0.f, v);   // quad_perm [2, 3, 0, 1]
      v += dpp_move<0x141, 0xF>(0.f, v);  // row_half_mirror
      v += dpp_move<0x140, 0xF>(0.f, v);  // row_mirror
      if (xi == 0) {
        const int ch = (tid >> 4) + j * (NT / 16);
        if (part) part[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * wgres_set_floats(C) + wgres_tile_floats(C) + ch] = v;
        else atomicAdd(&dbias[cob + ch], v);
      }
    }
  }
  if (prof) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    prof[3] = __builtin_readcyclecounter();
  }
}

// Sum of the S partial tile sets of every co group (part [cogroups][S][set floats], accumulator order) into dwp.  Thread =
// one float4 of the set (rows r .. r + 3 of one accumulator column) x one of 8 slice groups of S (four loads in flight), the
// groups met in LDS; 32 lanes read 512 contiguous bytes of a set (with 128-byte pieces 147 KB apart the 520 MB of these sets
// streamed at 3 TB/s).  dwp is read-modified-written without atomics: launches that accumulate into one dwp chunk are ordered
// by their stream.
template <int C>
__device__ __forceinline__ void wgrad_reduce_partials_body(const float4* __restrict__ part, const int S, const int cpad,
                                                                    float* __restrict__ dwp, float* __restrict__ dbias, const int bx, const int by) {
  constexpr int SET4 = wgres_set_floats(C) / 4, TILE4 = wgres_tile_floats(C) / 4;
  constexpr int NL = 32, NG = 8;  // 32 float4 columns (512 bytes of a set) x 8 slice groups of S per workgroup
  static_assert(TILE4 % NL == 0, "whole blocks");
  __shared__ float4 red[NG][NL];
  const int li = threadIdx.x & (NL - 1), sg = threadIdx.x / NL;
  const int f = bx * NL + li;  // host: grid.x * 32 == TILE4 (+ 32: the bias block, 16 float4 of it in use, launched when dbias != NULL)
  const float4* const p = part + (size_t)by * S * SET4 + (f < SET4 ? f : SET4 - 1);
  float4 acc4[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) acc4[u] = make_float4(0.f, 0.f, 0.f, 0.f);
  int s = sg;
  for (; s + 3 * NG < S; s += 4 * NG) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float4 v = p[(size_t)(s + NG * u) * SET4];
      acc4[u].x += v.x; acc4[u].y += v.y; acc4[u].z += v.z; acc4[u].w += v.w;
    }
  }
  for (; s < S; s += NG) {
    const float4 v = p[(size_t)s * SET4];
    acc4[0].x += v.x; acc4[0].y += v.y; acc4[0].z += v.z; acc4[0].w += v.w;
  }
  red[sg][li] = make_float4((acc4[0].x + acc4[1].x) + (acc4[2].x + acc4[3].x), (acc4[0].y + acc4[1].y) + (acc4[2].y + acc4[3].y),
                            (acc4[0].z + acc4[1].z) + (acc4[2].z + acc4[3].z), (acc4[0].w + acc4[1].w) + (acc4[2].w + acc4[3].w));
  __syncthreads();
  if (sg == 0) {
    float4 t = red[0][li];
#pragma unroll
    for (int k = 1; k < NG; ++k) { t.x += red[k][li].x; t.y += red[k][li].y; t.z += red[k][li].z; t.w += red[k][li].w; }
    if (f >= TILE4) {  // block-uniform: the bias sums of channels 4 (f - TILE4) .. + 3 of this co group (16 float4)
      if (f - TILE4 < 16) {
        float* const o = dbias + by * 64 + 4 * (f - TILE4);
        o[0] += t.x; o[1] += t.y; o[2] += t.z; o[3] += t.w;
      }
      return;
    }
    // f = ((wave * 9 + j) * 4 + gq) * 64 + lane  ->  rows cot + 8 gq + 4 kl + {0..3}, tap, ci (the kernel's tile assignment)
    const int lane = f & 63, gq = (f >> 6) & 3, wj = f >> 8, j = wj % 9, wave = wj / 9;
    const int il = lane & 31, kl = lane >> 5, tci = wave >> 1, half = wave & 1;
    const int k = j < 5 ? j : j - 5, tap = k < 3 ? (half ? 6 : 0) + k : (k == 3 ? (half ? 5 : 3) : 4);
    const int row = by * 64 + (j < 5 ? half : half ^ 1) * 32 + 8 * gq + 4 * kl;
    float* const o = dwp + ((size_t)row * 9 + tap) * cpad + tci * 32 + il;
    const size_t rs = (size_t)9 * cpad;
    o[0] += t.x; o[rs] += t.y; o[2 * rs] += t.z; o[3 * rs] += t.w;
  }
}
template <int C>
__global__ __launch_bounds__(256) void wgrad_reduce_partials_kernel(const float4* __restrict__ part, const int S, const int cpad,
                                                                    float* __restrict__ dwp, float* __restrict__ dbias) {
  wgrad_reduce_partials_body<C>(part, S, cpad, dwp, dbias, (int)blockIdx.x, (int)blockIdx.y);
}

// ------------------------------------------------------------------------------------------------
// Weight gradient of the 1x1 layers of the residual stacks on 64-pixel planes (64 -> 64 in the encoders, 128 -> 64 in the
// decoders), operands staged once per frame as in conv3x3_wgrad_resident_kernel:
//   dW[co][ci] = sum over frames n and pixels p of  a[n][co][p] * act(src)[n][ci][p]
// The register-direct kernel (conv_split.h: conv1x1_weight_grad_split_kernel) converts every `a` element once per ci tile
// and every src element once per co tile (2x - 4x) on one wave per SIMD, and a single wave issues a VALU instruction every
// ~5 cycles (tools/micro/mfma_shadow.hip): it is conversion-bound at about half the HBM rate.  Here a workgroup of EIGHT
// waves (two per SIMD issue twice the VALU instructions per cycle) owns a run of frames; every thread loads, activates
// and splits its share of each frame exactly once (coalesced 16-byte loads, requested TWO frames ahead through three
// register sets: one frame in flight per CU does not cover the HBM latency at this rate) and writes it to
// double-buffered [channel][64 pixels + 8] bf16 images; the MFMA work is small (6 or 12 per wave and frame): wave w holds
// tile w % NTILE and, with four tiles, half of the k-steps of every frame.  One barrier per frame; partial tiles stored
// (32 KiB per workgroup) and summed by wgrad_reduce_partials1x1_kernel, or by fp32 atomics when part == NULL.
// ------------------------------------------------------------------------------------------------
template <int SPLIT, int C>
__host__ __device__ constexpr int wg1x1_lds_bytes() { return 2 * SPLIT * (64 + C) * kWgresAPitch; }

template <int SPLIT, int C>
__global__ __launch_bounds__(512, 1) void conv1x1_wgrad_staged_kernel(
    const MtrssmConvGeom g, const float* __restrict__ a, const float* __restrict__ src, float* __restrict__ dwp,
    float* __restrict__ part, float* __restrict__ dbias, const int frames_per_wg) {
  static_assert(SPLIT == 1 || SPLIT == 2, "one or two bf16 pieces");
  static_assert(C == 64 || C == 128, "input channels");
  constexpr int NT = 512, P = kWgresAPitch;
  constexpr int XI = C * 16 / NT, AI = 2, NI = XI + AI;  // float4 items per thread and frame: src, a (64 co * 16)
  constexpr int NTILE = 2 * (C / 32), KS = 8 / NTILE;    // tiles per co group; waves per tile (each 4 / KS k-steps of a frame)
  constexpr int APB = 64 * P, XPB = C * P;               // bytes of one piece of the a / src image
  constexpr int BUFB = SPLIT * (APB + XPB);              // one buffer: [a pieces][src pieces]
  extern __shared__ __attribute__((aligned(16))) unsigned char wg1_lds[];
  unsigned char* const lds = wg1_lds;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int il = lane & 31, kl = lane >> 5;
  const int tile = wave % NTILE, kpart = wave / NTILE;
  const int tco = tile & 1, tci = tile >> 1;
  const int n0 = blockIdx.x * frames_per_wg;
  const int n1 = n0 + frames_per_wg < g.N ? n0 + frames_per_wg : g.N;
  if (n0 >= n1) return;  // whole workgroup
  const int nlast = n1 - 1;
  const int cob = blockIdx.y * 64;
  unsigned long long* const prof = (tid == 0 && blockIdx.y == 0 && g_res_prof) ? (blockIdx.x == 0 ? g_res_prof + 32 : (blockIdx.x == gridDim.x / 2 ? g_res_prof + 40 : nullptr)) : nullptr;
  if (prof) prof[0] = __builtin_readcyclecounter();

  const bool act_elu = g.act == MTRSSM_ACT_ELU, act_relu = g.act == MTRSSM_ACT_RELU, pre = g.pre_act != 0;
  auto act_sel = [&](float x) __attribute__((always_inline)) {
    float e = __expf(x) - 1.f;
    asm volatile("" : "+v"(e));  // computed unconditionally: the compiler would branch around the exponential
    const float neg = act_elu ? e : (act_relu ? 0.f : x);
    return (x > 0.f || !pre) ? x : neg;
  };

  // staging: item j = float4 number tid + 512 j of the frame: 4 consecutive pixels of channel (tid >> 4) + 32 j
  const float4* const xsrc = reinterpret_cast<const float4*>(src) + tid;
  const float4* const asrc = reinterpret_cast<const float4*>(a + (size_t)cob * 64) + tid;
  const size_t xfr = (size_t)C * 16, afr = (size_t)g.Cout * 16;  // float4 per frame
  const unsigned wofs = (unsigned)((tid >> 4) * P + (tid & 15) * 8);
  float bsum[AI];
#pragma unroll
  for (int j = 0; j < AI; ++j) bsum[j] = 0.f;
  wg_f32x4 raw[3][NI];  // [set][a items, x items]
  auto raw_load = [&](const int rs, const int n) __attribute__((always_inline)) {
#pragma unroll
    for (int it = 0; it < NI; ++it) {
      const float4* const ptr = it < AI ? asrc + (size_t)n * afr + NT * it : xsrc + (size_t)n * xfr + NT * (it - AI);
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(raw[rs][it]) : "v"(ptr));
    }
  };
  auto raw_wait = [&](const int rs) __attribute__((always_inline)) {  // the two younger sets may still be in flight
    if constexpr (NI == 4) asm volatile("s_waitcnt vmcnt(8)" : "+v"(raw[rs][0]), "+v"(raw[rs][1]), "+v"(raw[rs][2]), "+v"(raw[rs][3]));
    else asm volatile("s_waitcnt vmcnt(12)" : "+v"(raw[rs][0]), "+v"(raw[rs][1]), "+v"(raw[rs][2]), "+v"(raw[rs][3]), "+v"(raw[rs][4]), "+v"(raw[rs][5]));
  };
  auto stage = [&](const int rs, const unsigned bufoff) __attribute__((always_inline)) {
#pragma unroll
    for (int it = 0; it < NI; ++it) {
      const wg_f32x4 v = raw[rs][it];
      unsigned d0[SPLIT], d1[SPLIT];
      if (it < AI) {
        bsum[it] += (v.x + v.y) + (v.z + v.w);
        wg_split_pair<SPLIT>(v.x, v.y, d0);
        wg_split_pair<SPLIT>(v.z, v.w, d1);
      } else {
        wg_split_pair<SPLIT>(act_sel(v.x), act_sel(v.y), d0);
        wg_split_pair<SPLIT>(act_sel(v.z), act_sel(v.w), d1);
      }
      const unsigned base = it < AI ? (unsigned)(it * 32 * P) : (unsigned)(SPLIT * APB + (it - AI) * 32 * P);
      const unsigned piece = it < AI ? (unsigned)APB : (unsigned)XPB;
#pragma unroll
      for (int p = 0; p < SPLIT; ++p) *reinterpret_cast<uint2*>(lds + bufoff + base + wofs + p * piece) = make_uint2(d0[p], d1[p]);
    }
  };

  const unsigned lane_a = (unsigned)((tco * 32 + il) * P + kl * 16);
  const unsigned lane_b = (unsigned)(SPLIT * APB + (tci * 32 + il) * P + kl * 16);
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;

  raw_load(0, n0);
  raw_load(1, n0 + 1 < nlast ? n0 + 1 : nlast);
  if (prof) prof[1] = __builtin_readcyclecounter();
  // One frame: rs = (n - n0) % 3 and par = (n - n0) & 1 are literals at the call sites.
  auto frame = [&](const int n, const int rs, const int par) __attribute__((always_inline)) {
    const unsigned bufoff = par ? (unsigned)BUFB : 0u;
    raw_load((rs + 2) % 3, n + 2 < nlast ? n + 2 : nlast);  // its set held frame n - 1, staged a frame ago
    raw_wait(rs);
    stage(rs, bufoff);
    lds_barrier();  // images par complete; every wave is done reading images par ^ 1 (frame n - 1)
#pragma unroll
    for (int qq = 0; qq < 4 / KS; ++qq) {
      const int q = kpart * (4 / KS) + qq;
      u32x4 qa[SPLIT], qb[SPLIT];
#pragma unroll
      for (int p = 0; p < SPLIT; ++p) {
        qa[p] = *reinterpret_cast<const u32x4*>(lds + bufoff + lane_a + (unsigned)(p * APB) + (unsigned)(q * 32));
        qb[p] = *reinterpret_cast<const u32x4*>(lds + bufoff + lane_b + (unsigned)(p * XPB) + (unsigned)(q * 32));
      }
#pragma unroll
      for (int ord = SPLIT - 1; ord >= 0; --ord)
#pragma unroll
        for (int sa = 0; sa <= ord; ++sa)
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, qa[sa]), __builtin_bit_cast(bf16x8, qb[ord - sa]), acc, 0, 0, 0);
    }
  };
#pragma unroll 1
  for (int n = n0; n < n1; n += 6) {
    frame(n, 0, 0);
    if (n + 1 < n1) frame(n + 1, 1, 1);
    if (n + 2 < n1) frame(n + 2, 2, 0);
    if (n + 3 < n1) frame(n + 3, 0, 1);
    if (n + 4 < n1) frame(n + 4, 1, 0);
    if (n + 5 < n1) frame(n + 5, 2, 1);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the clamped requests of the last frames
  if (prof) prof[2] = __builtin_readcyclecounter();

  if (part) {
    // one tile per wave, accumulator order: float4 number (wave * 4 + r / 4) * 64 + lane of this workgroup's set (2048 atomic
    // adders per address took 9-27 us of a 41 us launch; wgrad_reduce_partials1x1_kernel sums the sets behind this kernel)
    float4* const ps = reinterpret_cast<float4*>(part) + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * (kWg1x1SetFloats / 4) + (size_t)wave * 256 + lane;
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) ps[gq * 64] = make_float4(acc[4 * gq], acc[4 * gq + 1], acc[4 * gq + 2], acc[4 * gq + 3]);
  } else {
    const int ci = tci * 32 + il;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = cob + tco * 32 + (r & 3) + 8 * (r >> 2) + 4 * kl;
      atomicAdd(&dwp[(size_t)row * g.Cpad + ci], acc[r]);
    }
  }
  if (dbias != nullptr) {
#pragma unroll
    for (int j = 0; j < AI; ++j) {  // the 16 lanes of a DPP row share a channel
      float v = bsum[j];
      v += dpp_move<0xB1, 0xF>(0.f, v);
      v += dpp_move<0x4E, 0xF>(0.f, v);
      v += dpp_move<0x141, 0xF>(0.f, v);
      v += dpp_move<0x140, 0xF>(0.f, v);
      if ((tid & 15) == 0) {
        const int ch = (tid >> 4) + j * 32;
        if (part) part[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * kWg1x1SetFloats + 8 * 1024 + ch] = v;
        else atomicAdd(&dbias[cob + ch], v);
      }
    }
  }
  if (prof) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    prof[3] = __builtin_readcyclecounter();
  }
}

// Partial tiles of conv1x1_wgrad_staged_kernel (part [cogroups][S][8 waves][256] float4, accumulator order) into dwp; a
// workgroup's waves w and w + NTILE hold the same tile (k-halves), so the kernel sums 8 / NTILE x S slices per tile.
template <int C>
__device__ __forceinline__ void wgrad_reduce_partials1x1_body(const float4* __restrict__ part, const int S, const int cpad,
                                                                       float* __restrict__ dwp, float* __restrict__ dbias, const int bx, const int by) {
  constexpr int NTILE = 2 * (C / 32), KS = 8 / NTILE, TILE4 = 256, SET4 = kWg1x1SetFloats / 4;
  constexpr int NL = 32, NG = 8;  // 32 float4 columns (512 bytes of a slice) x 8 slice groups per workgroup
  __shared__ float4 red[NG][NL];
  const int li = threadIdx.x & (NL - 1), sg = threadIdx.x / NL;
  const int f = bx * NL + li;  // float4 of the NTILE tiles; host: grid.x * 32 == NTILE * 256 (+ 32: the bias block, 16 float4 in use)
  const bool bias = f >= NTILE * TILE4;  // block-uniform
  const int tile = f / TILE4, ft = f - tile * TILE4;
  const float4* const p = part + (size_t)by * S * SET4 + (bias ? (size_t)(8 * TILE4 + (f - NTILE * TILE4 < 16 ? f - NTILE * TILE4 : 15)) : (size_t)tile * TILE4 + ft);
  float4 acc4[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) acc4[u] = make_float4(0.f, 0.f, 0.f, 0.f);
  const int ks = bias ? 1 : KS;
  const int slices = S * ks;  // slice i = (workgroup i / ks, k-half i % ks)
  int i = sg;
  for (; i + 3 * NG < slices; i += 4 * NG) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int ii = i + NG * u;
      const float4 v = p[(size_t)(ii / ks) * SET4 + (size_t)(ii % ks) * (NTILE * TILE4)];
      acc4[u].x += v.x; acc4[u].y += v.y; acc4[u].z += v.z; acc4[u].w += v.w;
    }
  }
  for (; i < slices; i += NG) {
    const float4 v = p[(size_t)(i / ks) * SET4 + (size_t)(i % ks) * (NTILE * TILE4)];
    acc4[0].x += v.x; acc4[0].y += v.y; acc4[0].z += v.z; acc4[0].w += v.w;
  }
  red[sg][li] = make_float4((acc4[0].x + acc4[1].x) + (acc4[2].x + acc4[3].x), (acc4[0].y + acc4[1].y) + (acc4[2].y + acc4[3].y),
                            (acc4[0].z + acc4[1].z) + (acc4[2].z + acc4[3].z), (acc4[0].w + acc4[1].w) + (acc4[2].w + acc4[3].w));
  __syncthreads();
  if (sg == 0) {
    float4 v = red[0][li];
#pragma unroll
    for (int k = 1; k < NG; ++k) { v.x += red[k][li].x; v.y += red[k][li].y; v.z += red[k][li].z; v.w += red[k][li].w; }
    if (bias) {
      if (f - NTILE * TILE4 < 16) {
        float* const o = dbias + by * 64 + 4 * (f - NTILE * TILE4);
        o[0] += v.x; o[1] += v.y; o[2] += v.z; o[3] += v.w;
      }
      return;
    }
    const int lane = ft & 63, gq = ft >> 6, il = lane & 31, kl = lane >> 5;
    const int tco = tile & 1, tci = tile >> 1;
    float* const o = dwp + (size_t)(by * 64 + tco * 32 + 8 * gq + 4 * kl) * cpad + tci * 32 + il;
    o[0] += v.x; o[cpad] += v.y; o[2 * (size_t)cpad] += v.z; o[3 * (size_t)cpad] += v.w;
  }
}
template <int C>
__global__ __launch_bounds__(256) void wgrad_reduce_partials1x1_kernel(const float4* __restrict__ part, const int S, const int cpad,
                                                                       float* __restrict__ dwp, float* __restrict__ dbias) {
  wgrad_reduce_partials1x1_body<C>(part, S, cpad, dwp, dbias, (int)blockIdx.x, (int)blockIdx.y);
}

// ------------------------------------------------------------------------------------------------
// Weight gradient of the second encoder layer (3x3 / stride 2 / pad 1, 8 -> 16 channels, 256-pixel output planes: 16x16
// vision, 32x8 audio), staged like the kernels above:
//   dW[co][ky][kx][ci] = sum over frames n and output pixels (oy, ox) of  a[n][co][oy][ox] * act(src)[n][ci][2 oy + ky - 1][2 ox + kx - 1]
// The patch-staged kernel (conv_split.h: conv_weight_grad_split_kernel<1, 2>) gathers its patches by 4-byte loads and runs
// at 0.13 of the HBM rate (235-246 us for 157 MB).  Here every frame of `a` (16 x 256) and of src (8 x 2Ho x 2Wo) is loaded
// once with coalesced 16-byte loads (eight waves, requests two frames ahead through three register sets), converted once and
// written to LDS; src DE-INTERLEAVED by column parity (even / odd image, one zero halo row on top of every plane), so that the
// 8 consecutive output pixels of a lane's B fragment are 8 consecutive elements: kx = 1 reads the even image at ox, kx = 2 the
// odd image at ox, kx = 0 the odd image at ox - 1 = a funnel shift of the kx = 2 fragment with one more element in front.
// MFMA columns are (tap, ci) pairs, 72 of them in three 32-column tiles; rows are the 16 co (the upper half of the 32-row
// tile is unused).  Wave w: column tile w & 3 (3: staging only), k-steps (w >> 2) * 8 .. + 8 of the frame's 16.  Partial set:
// float4 number (w * 2 + r / 4) * 64 + lane, r < 8, + 16 bias sums.
// ------------------------------------------------------------------------------------------------
constexpr int kWgS2SetFloats = 8 * 2 * 64 * 4 + 16;
template <int WO>
__host__ __device__ constexpr int wgs2_cip() { return (((256 / WO * 2 + 1) * (2 * WO)) / 16 | 1) * 16; }  // bytes per channel of one src image: odd in 16-byte slots
template <int SPLIT, int WO>
__host__ __device__ constexpr int wgs2_lds_bytes() { return 2 * SPLIT * (2 * 8 * wgs2_cip<WO>() + 16 * 528); }

template <int SPLIT, int WO>
__global__ __launch_bounds__(512, 1) void conv3x3s2_wgrad_staged_kernel(
    const MtrssmConvGeom g, const float* __restrict__ a, const float* __restrict__ src, float* __restrict__ dwp,
    float* __restrict__ part, float* __restrict__ dbias, const int frames_per_wg) {
  static_assert(SPLIT == 1 || SPLIT == 2, "one or two bf16 pieces");
  static_assert(WO == 16 || WO == 8, "output plane 16x16 or 32x8");
  constexpr int NT = 512, C = 8, CO = 16, HO = 256 / WO, HS = 2 * HO, WS = 2 * WO;
  constexpr int ROWB = WS;                 // bytes per row of one parity image: WS / 2 bf16
  constexpr int CIP = wgs2_cip<WO>();      // bytes per channel: (HS + 1) rows, padded
  constexpr int XCOPY = C * CIP;           // one parity image of one piece
  constexpr int AP = 528, APB = CO * AP;   // `a` image: [co][256 pixels + 8] bf16
  constexpr int BUFB = SPLIT * (2 * XCOPY + APB);  // one buffer: [piece][even image][odd image] then [piece][a image]
  constexpr int XI = C * HS * WS / 4 / NT, AI = CO * 256 / 4 / NT, NI = XI + AI;  // 4 + 2 float4 items per thread and frame
  extern __shared__ __attribute__((aligned(16))) unsigned char wgs2_lds[];
  unsigned char* const lds = wgs2_lds;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int il = lane & 31, kl = lane >> 5;
  const int ct = wave & 3, kh = wave >> 2;
  const int n0 = blockIdx.x * frames_per_wg;
  const int n1 = n0 + frames_per_wg < g.N ? n0 + frames_per_wg : g.N;
  if (n0 >= n1) return;  // whole workgroup
  const int nlast = n1 - 1;

  const bool act_elu = g.act == MTRSSM_ACT_ELU, act_relu = g.act == MTRSSM_ACT_RELU, pre = g.pre_act != 0;
  auto act_sel = [&](float x) __attribute__((always_inline)) {
    float e = __expf(x) - 1.f;
    asm volatile("" : "+v"(e));  // computed unconditionally: the compiler would branch around the exponential
    const float neg = act_elu ? e : (act_relu ? 0.f : x);
    return (x > 0.f || !pre) ? x : neg;
  };

  // zero both buffers once: the halo rows are never written again
  for (int o = tid * 16; o < 2 * BUFB; o += NT * 16) *reinterpret_cast<uint4*>(lds + o) = make_uint4(0u, 0u, 0u, 0u);

  // staging: src item j = float4 number tid + 512 j of the frame = 4 consecutive columns c0 .. c0 + 3 of one row of one
  // channel; `a` item j = 4 consecutive pixels of channel tid / 64 + 8 j
  const float4* const xsrc = reinterpret_cast<const float4*>(src) + tid;
  const float4* const asrc = reinterpret_cast<const float4*>(a) + tid;
  constexpr size_t xfr = (size_t)C * HS * WS / 4, afr = (size_t)CO * 256 / 4;  // float4 per frame
  float bsum[AI];
#pragma unroll
  for (int j = 0; j < AI; ++j) bsum[j] = 0.f;
  wg_f32x4 raw[3][NI];  // [set][a items, src items]
  auto raw_load = [&](const int rs, const int n) __attribute__((always_inline)) {
#pragma unroll
    for (int it = 0; it < NI; ++it) {
      const float4* const ptr = it < AI ? asrc + (size_t)n * afr + NT * it : xsrc + (size_t)n * xfr + NT * (it - AI);
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(raw[rs][it]) : "v"(ptr));
    }
  };
  auto raw_wait = [&](const int rs) __attribute__((always_inline)) {  // the two younger sets may still be in flight
    static_assert(NI == 6, "six items");
    asm volatile("s_waitcnt vmcnt(12)" : "+v"(raw[rs][0]), "+v"(raw[rs][1]), "+v"(raw[rs][2]), "+v"(raw[rs][3]), "+v"(raw[rs][4]), "+v"(raw[rs][5]));
  };
  auto stage = [&](const int rs, const unsigned bufoff) __attribute__((always_inline)) {
#pragma unroll
    for (int it = 0; it < NI; ++it) {
      const wg_f32x4 v = raw[rs][it];
      if (it < AI) {
        bsum[it] += (v.x + v.y) + (v.z + v.w);
        unsigned d0[SPLIT], d1[SPLIT];
        wg_split_pair<SPLIT>(v.x, v.y, d0);
        wg_split_pair<SPLIT>(v.z, v.w, d1);
        const int f = tid + NT * it;  // co = f / 64, pixel = 4 (f % 64)
        const unsigned o = bufoff + (unsigned)(SPLIT * 2 * XCOPY) + (unsigned)((f >> 6) * AP + (f & 63) * 8);
#pragma unroll
        for (int p = 0; p < SPLIT; ++p) *reinterpret_cast<uint2*>(lds + o + p * APB) = make_uint2(d0[p], d1[p]);
      } else {
        unsigned de[SPLIT], dd[SPLIT];  // even columns (c0, c0 + 2), odd columns (c0 + 1, c0 + 3)
        wg_split_pair<SPLIT>(act_sel(v.x), act_sel(v.z), de);
        wg_split_pair<SPLIT>(act_sel(v.y), act_sel(v.w), dd);
        const int f = tid + NT * (it - AI);
        const int ci = f / (HS * WS / 4), rem = f - ci * (HS * WS / 4), row = rem / (WS / 4), c4 = rem - row * (WS / 4);
        const unsigned o = bufoff + (unsigned)(ci * CIP + (row + 1) * ROWB + c4 * 4);  // columns c0 / 2, c0 / 2 + 1 of the parity images
#pragma unroll
        for (int p = 0; p < SPLIT; ++p) {
          *reinterpret_cast<unsigned*>(lds + o + p * 2 * XCOPY) = de[p];
          *reinterpret_cast<unsigned*>(lds + o + p * 2 * XCOPY + XCOPY) = dd[p];
        }
      }
    }
  };

  // the lane's column of the tile: (tap, ci); columns 72 .. 95 of tile 2 are padding (computed on tap 8, never stored)
  const int col = ct * 32 + il, colc = col < 72 ? col : 71;
  const int tap = colc >> 3, ci = colc & 7, ky = tap / 3, kx = tap - 3 * ky;
  // k-step s covers output pixels 16 s .. 16 s + 15, this lane 8 of them: WO = 16: row s, columns 8 kl ..; WO = 8: row 2 s + kl
  const unsigned lane_b = (unsigned)((kx == 1 ? 0 : XCOPY) + ci * CIP + ky * ROWB + (WO == 16 ? kl * 16 : kl * 2 * ROWB));
  constexpr unsigned kStepB = WO == 16 ? 2 * ROWB : 4 * ROWB;
  const unsigned lane_a = (unsigned)(SPLIT * 2 * XCOPY + (il & 15) * AP + kl * 16);
  const bool shifted = kx == 0;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;

  raw_load(0, n0);
  raw_load(1, n0 + 1 < nlast ? n0 + 1 : nlast);
  lds_barrier();  // the zeroes are in place
  // One frame: rs = (n - n0) % 3 and par = (n - n0) & 1 are literals at the call sites.
  auto frame = [&](const int n, const int rs, const int par) __attribute__((always_inline)) {
    const unsigned bufoff = par ? (unsigned)BUFB : 0u;
    raw_load((rs + 2) % 3, n + 2 < nlast ? n + 2 : nlast);  // its set held frame n - 1, staged a frame ago
    raw_wait(rs);
    stage(rs, bufoff);
    lds_barrier();  // images par complete; every wave is done reading images par ^ 1 (frame n - 1)
    if (ct < 3) {
#pragma unroll
      for (int ss = 0; ss < 8; ++ss) {
        const int s = kh * 8 + ss;
        u32x4 qa[SPLIT], qb[SPLIT];
#pragma unroll
        for (int p = 0; p < SPLIT; ++p) {
          qa[p] = *reinterpret_cast<const u32x4*>(lds + bufoff + lane_a + (unsigned)(p * APB) + (unsigned)(s * 32));
          const unsigned bo = bufoff + lane_b + (unsigned)(p * 2 * XCOPY) + (unsigned)s * kStepB;
          const u32x4 f = *reinterpret_cast<const u32x4*>(lds + bo);
          // kx = 0: the odd image one element to the left; the element in front of the fragment is the last one of the
          // row's first half (WO = 16, second half) or the zero left of the row
          unsigned prev = 0u;
          if (WO == 16) prev = kl ? (unsigned)*reinterpret_cast<const unsigned short*>(lds + bo - 2) : 0u;
          const u32x4 sh = u32x4{(f.x << 16) | prev, __builtin_amdgcn_alignbit(f.y, f.x, 16), __builtin_amdgcn_alignbit(f.z, f.y, 16), __builtin_amdgcn_alignbit(f.w, f.z, 16)};
          qb[p] = shifted ? sh : f;
        }
#pragma unroll
        for (int ord = SPLIT - 1; ord >= 0; --ord)
#pragma unroll
          for (int sa = 0; sa <= ord; ++sa)
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, qa[sa]), __builtin_bit_cast(bf16x8, qb[ord - sa]), acc, 0, 0, 0);
      }
    }
  };
#pragma unroll 1
  for (int n = n0; n < n1; n += 6) {
    frame(n, 0, 0);
    if (n + 1 < n1) frame(n + 1, 1, 1);
    if (n + 2 < n1) frame(n + 2, 2, 0);
    if (n + 3 < n1) frame(n + 3, 0, 1);
    if (n + 4 < n1) frame(n + 4, 1, 0);
    if (n + 5 < n1) frame(n + 5, 2, 1);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the clamped requests of the last frames

  if (part) {
    float* const set = part + (size_t)blockIdx.x * kWgS2SetFloats;
    float4* const ps = reinterpret_cast<float4*>(set) + (size_t)wave * 128 + lane;
    ps[0] = make_float4(acc[0], acc[1], acc[2], acc[3]);   // rows 4 kl + 0 .. 3
    ps[64] = make_float4(acc[4], acc[5], acc[6], acc[7]);  // rows 8 + 4 kl + 0 .. 3
    if (dbias != nullptr) {
#pragma unroll
      for (int j = 0; j < AI; ++j) {  // a wave's 64 lanes share channel wave + 8 j
        const float v = wave_sum(bsum[j]);
        if (lane == 0) set[8 * 2 * 64 * 4 + wave + 8 * j] = v;
      }
    }
  } else {
    if (ct < 3 && col < 72) {
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * kl;
        atomicAdd(&dwp[((size_t)row * 9 + tap) * g.Cpad + ci], acc[r]);
      }
    }
    if (dbias != nullptr) {
#pragma unroll
      for (int j = 0; j < AI; ++j) {
        const float v = wave_sum(bsum[j]);
        if (lane == 0) atomicAdd(&dbias[wave + 8 * j], v);
      }
    }
  }
}

// Partial sets of conv3x3s2_wgrad_staged_kernel into dwp / dbias: 72 columns x 16 rows, two k-halves per workgroup.
__device__ __forceinline__ void wgrad_reduce_partials_s2_body(const float4* __restrict__ part, const int S, const int cpad,
                                                                       float* __restrict__ dwp, float* __restrict__ dbias, const int bx, const int by) {
  constexpr int SET4 = kWgS2SetFloats / 4, NL = 8, NG = 32;
  __shared__ float4 red[NG][NL];
  const int li = threadIdx.x & (NL - 1), sg = threadIdx.x / NL;
  const int f = bx * NL + li;  // float4 (ct * 2 + half) * 64 + lane of the three column tiles: 384 of them, then 4 bias float4
  const bool bias = f >= 384;          // block-uniform (48 tile blocks, then one bias block of which 4 columns are used)
  if (bias && f >= 388) {              // padding columns of the bias block: nothing to read
    red[sg][li] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const int ct = f >> 7, rem = f & 127;
  float4 acc4[2];
  acc4[0] = acc4[1] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (!(bias && f >= 388)) {
    const int ks = bias ? 1 : 2;
    const int slices = S * ks;  // slice i = (workgroup i / ks, k-half i % ks): waves ct and ct + 4
    const float4* const p = part + (bias ? (size_t)(8 * 128 + (f - 384)) : (size_t)ct * 128 + rem);
    int i = sg;
    for (; i + NG < slices; i += 2 * NG) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int ii = i + NG * u;
        const float4 v = p[(size_t)(ii / ks) * SET4 + (size_t)(ii % ks) * (4 * 128)];
        acc4[u].x += v.x; acc4[u].y += v.y; acc4[u].z += v.z; acc4[u].w += v.w;
      }
    }
    for (; i < slices; i += NG) {
      const float4 v = p[(size_t)(i / ks) * SET4 + (size_t)(i % ks) * (4 * 128)];
      acc4[0].x += v.x; acc4[0].y += v.y; acc4[0].z += v.z; acc4[0].w += v.w;
    }
    red[sg][li] = make_float4(acc4[0].x + acc4[1].x, acc4[0].y + acc4[1].y, acc4[0].z + acc4[1].z, acc4[0].w + acc4[1].w);
  }
  __syncthreads();
  if (sg == 0) {
    float4 v = red[0][li];
#pragma unroll
    for (int k = 1; k < NG; ++k) { v.x += red[k][li].x; v.y += red[k][li].y; v.z += red[k][li].z; v.w += red[k][li].w; }
    if (bias) {
      if (f < 388) { float* const o = dbias + 4 * (f - 384); o[0] += v.x; o[1] += v.y; o[2] += v.z; o[3] += v.w; }
      return;
    }
    const int half = rem >> 6, lane = rem & 63, il = lane & 31, kl = lane >> 5;
    const int col = ct * 32 + il;
    if (col < 72) {
      const int tap = col >> 3, ci = col & 7, row = 8 * half + 4 * kl;
      float* const o = dwp + ((size_t)row * 9 + tap) * cpad + ci;
      const size_t rs = (size_t)9 * cpad;
      o[0] += v.x; o[rs] += v.y; o[2 * rs] += v.z; o[3 * rs] += v.w;
    }
  }
}
__global__ __launch_bounds__(256) void wgrad_reduce_partials_s2_kernel(const float4* __restrict__ part, const int S, const int cpad,
                                                                       float* __restrict__ dwp, float* __restrict__ dbias) {
  wgrad_reduce_partials_s2_body(part, S, cpad, dwp, dbias, (int)blockIdx.x, (int)blockIdx.y);
}

// ------------------------------------------------------------------------------------------------
// Weight gradient of the decoders' second ConvTranspose2d (k = 4, stride 2, pad 1, 32 -> 16 channels; input planes of 256
// pixels: 16x16 vision, 32x8 audio), the same recipe with the roles the transposed layer gives the operands:
//   dW[ci][ky][kx][co] = sum over frames n and INPUT pixels (y, x) of  act(a)[n][ci][y][x] * src[n][co][2 y - 1 + ky][2 x - 1 + kx]
// `a` is the layer input (32 channels, activated here), src the output gradient (16 channels, 2H x 2W, raw).  src is
// de-interleaved by column parity with a zero halo row above AND below every plane: kx = 0 reads the odd image at x - 1
// (funnel shift with the element in front), kx = 1 the even image at x, kx = 2 the odd image at x, kx = 3 the even image
// at x + 1 (funnel shift the other way with the element behind).  MFMA columns are (tap, co) pairs: 16 x 16 = 256 = one
// 32-column tile per wave; rows are the 32 ci.  One image buffer (102 KB: two barriers per frame), raw frames (96 KB per
// frame and workgroup) requested two frames ahead.  Partial set: float4 number (w * 4 + r / 4) * 64 + lane.
// ------------------------------------------------------------------------------------------------
constexpr int kWgT4SetFloats = 8 * 4 * 64 * 4;
template <int WO>
__host__ __device__ constexpr int wgt4_cip() { return (((256 / WO * 2 + 2) * (2 * WO)) / 16 | 1) * 16; }
template <int SPLIT, int WO>
__host__ __device__ constexpr int wgt4_lds_bytes() { return SPLIT * (2 * 16 * wgt4_cip<WO>() + 32 * 528); }

template <int SPLIT, int WO>
__global__ __launch_bounds__(512, 1) void convt4s2_wgrad_staged_kernel(
    const MtrssmConvGeom g, const float* __restrict__ a, const float* __restrict__ src, const int pre_act_a, float* __restrict__ dwp,
    float* __restrict__ part, const int frames_per_wg) {
  static_assert(SPLIT == 1 || SPLIT == 2, "one or two bf16 pieces");
  static_assert(WO == 16 || WO == 8, "input plane 16x16 or 32x8");
  constexpr int NT = 512, C = 16, CO = 32, HO = 256 / WO, HS = 2 * HO, WS = 2 * WO;
  constexpr int ROWB = WS;                 // bytes per row of one parity image: WS / 2 bf16
  constexpr int CIP = wgt4_cip<WO>();      // bytes per channel: (HS + 2) rows, padded
  constexpr int XCOPY = C * CIP;           // one parity image of one piece
  constexpr int AP = 528, APB = CO * AP;   // `a` image: [ci][256 pixels + 8] bf16
  constexpr int XI = C * HS * WS / 4 / NT, AI = CO * 256 / 4 / NT, NI = XI + AI;  // 8 + 4 float4 items per thread and frame
  extern __shared__ __attribute__((aligned(16))) unsigned char wgt4_lds[];
  unsigned char* const lds = wgt4_lds;     // [piece][even image][odd image] then [piece][a image]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int il = lane & 31, kl = lane >> 5;
  const int n0 = blockIdx.x * frames_per_wg;
  const int n1 = n0 + frames_per_wg < g.N ? n0 + frames_per_wg : g.N;
  if (n0 >= n1) return;  // whole workgroup
  const int nlast = n1 - 1;

  const bool act_elu = g.act == MTRSSM_ACT_ELU, act_relu = g.act == MTRSSM_ACT_RELU, pre = pre_act_a != 0;
  auto act_sel = [&](float x) __attribute__((always_inline)) {
    float e = __expf(x) - 1.f;
    asm volatile("" : "+v"(e));  // computed unconditionally: the compiler would branch around the exponential
    const float neg = act_elu ? e : (act_relu ? 0.f : x);
    return (x > 0.f || !pre) ? x : neg;
  };

  // zero the buffer once: the halo rows are never written again
  for (int o = tid * 16; o < wgt4_lds_bytes<SPLIT, WO>(); o += NT * 16) *reinterpret_cast<uint4*>(lds + o) = make_uint4(0u, 0u, 0u, 0u);

  const float4* const xsrc = reinterpret_cast<const float4*>(src) + tid;
  const float4* const asrc = reinterpret_cast<const float4*>(a) + tid;
  constexpr size_t xfr = (size_t)C * HS * WS / 4, afr = (size_t)CO * 256 / 4;  // float4 per frame
  wg_f32x4 raw[3][NI];  // [set][a items, src items]
  auto raw_load = [&](const int rs, const int n) __attribute__((always_inline)) {
#pragma unroll
    for (int it = 0; it < NI; ++it) {
      const float4* const ptr = it < AI ? asrc + (size_t)n * afr + NT * it : xsrc + (size_t)n * xfr + NT * (it - AI);
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(raw[rs][it]) : "v"(ptr));
    }
  };
  auto raw_wait = [&](const int rs) __attribute__((always_inline)) {  // the two younger sets may still be in flight
    static_assert(NI == 12, "twelve items");
    asm volatile("s_waitcnt vmcnt(24)" : "+v"(raw[rs][0]), "+v"(raw[rs][1]), "+v"(raw[rs][2]), "+v"(raw[rs][3]), "+v"(raw[rs][4]), "+v"(raw[rs][5]),
                 "+v"(raw[rs][6]), "+v"(raw[rs][7]), "+v"(raw[rs][8]), "+v"(raw[rs][9]), "+v"(raw[rs][10]), "+v"(raw[rs][11]));
  };
  auto stage = [&](const int rs) __attribute__((always_inline)) {
#pragma unroll
    for (int it = 0; it < NI; ++it) {
      const wg_f32x4 v = raw[rs][it];
      if (it < AI) {
        unsigned d0[SPLIT], d1[SPLIT];
        wg_split_pair<SPLIT>(act_sel(v.x), act_sel(v.y), d0);
        wg_split_pair<SPLIT>(act_sel(v.z), act_sel(v.w), d1);
        const int f = tid + NT * it;  // ci = f / 64, pixel = 4 (f % 64)
        const unsigned o = (unsigned)(SPLIT * 2 * XCOPY) + (unsigned)((f >> 6) * AP + (f & 63) * 8);
#pragma unroll
        for (int p = 0; p < SPLIT; ++p) *reinterpret_cast<uint2*>(lds + o + p * APB) = make_uint2(d0[p], d1[p]);
      } else {
        unsigned de[SPLIT], dd[SPLIT];  // even columns (c0, c0 + 2), odd columns (c0 + 1, c0 + 3)
        wg_split_pair<SPLIT>(v.x, v.z, de);
        wg_split_pair<SPLIT>(v.y, v.w, dd);
        const int f = tid + NT * (it - AI);
        const int co = f / (HS * WS / 4), rem = f - co * (HS * WS / 4), row = rem / (WS / 4), c4 = rem - row * (WS / 4);
        const unsigned o = (unsigned)(co * CIP + (row + 1) * ROWB + c4 * 4);
#pragma unroll
        for (int p = 0; p < SPLIT; ++p) {
          *reinterpret_cast<unsigned*>(lds + o + p * 2 * XCOPY) = de[p];
          *reinterpret_cast<unsigned*>(lds + o + p * 2 * XCOPY + XCOPY) = dd[p];
        }
      }
    }
  };

  // the lane's column: (tap, co) = (col >> 4, col & 15)
  const int col = wave * 32 + il;
  const int tap = col >> 4, co = col & 15, ky = tap >> 2, kx = tap & 3;
  const bool odd = kx == 0 || kx == 2;
  // k-step s covers input pixels 16 s .. 16 s + 15, this lane 8 of them: WO = 16: row s, columns 8 kl ..; WO = 8: row 2 s + kl
  const unsigned lane_b = (unsigned)((odd ? XCOPY : 0) + co * CIP + ky * ROWB + (WO == 16 ? kl * 16 : kl * 2 * ROWB));
  constexpr unsigned kStepB = WO == 16 ? 2 * ROWB : 4 * ROWB;
  const unsigned lane_a = (unsigned)(SPLIT * 2 * XCOPY + il * AP + kl * 16);
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;

  raw_load(0, n0);
  raw_load(1, n0 + 1 < nlast ? n0 + 1 : nlast);
  auto frame = [&](const int n, const int rs) __attribute__((always_inline)) {  // rs = (n - n0) % 3, a literal at the call sites
    raw_load((rs + 2) % 3, n + 2 < nlast ? n + 2 : nlast);  // its set held frame n - 1, staged a frame ago
    raw_wait(rs);
    lds_barrier();  // every wave is done reading the previous frame's images (first frame: the zeroes are in place)
    stage(rs);
    lds_barrier();  // images complete
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      u32x4 qa[SPLIT], qb[SPLIT];
#pragma unroll
      for (int p = 0; p < SPLIT; ++p) {
        qa[p] = *reinterpret_cast<const u32x4*>(lds + lane_a + (unsigned)(p * APB) + (unsigned)(s * 32));
        const unsigned bo = lane_b + (unsigned)(p * 2 * XCOPY) + (unsigned)s * kStepB;
        const u32x4 f = *reinterpret_cast<const u32x4*>(lds + bo);
        // the element in front of / behind the fragment: the neighbouring half of the row (WO = 16) or the zero beyond it
        unsigned prev = 0u, next = 0u;
        if (WO == 16) {
          prev = kl ? (unsigned)*reinterpret_cast<const unsigned short*>(lds + bo - 2) : 0u;
          next = kl ? 0u : (unsigned)*reinterpret_cast<const unsigned short*>(lds + bo + 16);
        }
        const u32x4 shr = u32x4{(f.x << 16) | prev, __builtin_amdgcn_alignbit(f.y, f.x, 16), __builtin_amdgcn_alignbit(f.z, f.y, 16), __builtin_amdgcn_alignbit(f.w, f.z, 16)};
        const u32x4 shl = u32x4{__builtin_amdgcn_alignbit(f.y, f.x, 16), __builtin_amdgcn_alignbit(f.z, f.y, 16), __builtin_amdgcn_alignbit(f.w, f.z, 16), (f.w >> 16) | (next << 16)};
        qb[p] = kx == 0 ? shr : (kx == 3 ? shl : f);
      }
#pragma unroll
      for (int ord = SPLIT - 1; ord >= 0; --ord)
#pragma unroll
        for (int sa = 0; sa <= ord; ++sa)
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, qa[sa]), __builtin_bit_cast(bf16x8, qb[ord - sa]), acc, 0, 0, 0);
    }
  };
#pragma unroll 1
  for (int n = n0; n < n1; n += 3) {
    frame(n, 0);
    if (n + 1 < n1) frame(n + 1, 1);
    if (n + 2 < n1) frame(n + 2, 2);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the clamped requests of the last frames

  if (part) {
    float4* const ps = reinterpret_cast<float4*>(part) + (size_t)blockIdx.x * (kWgT4SetFloats / 4) + (size_t)wave * 256 + lane;
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) ps[gq * 64] = make_float4(acc[4 * gq], acc[4 * gq + 1], acc[4 * gq + 2], acc[4 * gq + 3]);
  } else {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * kl;
      atomicAdd(&dwp[((size_t)row * 16 + tap) * g.Cpad + co], acc[r]);
    }
  }
}

// Partial sets of convt4s2_wgrad_staged_kernel into dwp.
__device__ __forceinline__ void wgrad_reduce_partials_t4_body(const float4* __restrict__ part, const int S, const int cpad,
                                                                       float* __restrict__ dwp, const int bx, const int by) {
  constexpr int SET4 = kWgT4SetFloats / 4, NL = 8, NG = 32;
  __shared__ float4 red[NG][NL];
  const int li = threadIdx.x & (NL - 1), sg = threadIdx.x / NL;
  const int f = bx * NL + li;  // host: grid.x * 8 == SET4
  const float4* const p = part + f;
  float4 acc4[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) acc4[u] = make_float4(0.f, 0.f, 0.f, 0.f);
  int s = sg;
  for (; s + 3 * NG < S; s += 4 * NG) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float4 v = p[(size_t)(s + NG * u) * SET4];
      acc4[u].x += v.x; acc4[u].y += v.y; acc4[u].z += v.z; acc4[u].w += v.w;
    }
  }
  for (; s < S; s += NG) {
    const float4 v = p[(size_t)s * SET4];
    acc4[0].x += v.x; acc4[0].y += v.y; acc4[0].z += v.z; acc4[0].w += v.w;
  }
  red[sg][li] = make_float4((acc4[0].x + acc4[1].x) + (acc4[2].x + acc4[3].x), (acc4[0].y + acc4[1].y) + (acc4[2].y + acc4[3].y),
                            (acc4[0].z + acc4[1].z) + (acc4[2].z + acc4[3].z), (acc4[0].w + acc4[1].w) + (acc4[2].w + acc4[3].w));
  __syncthreads();
  if (sg == 0) {
    float4 v = red[0][li];
#pragma unroll
    for (int k = 1; k < NG; ++k) { v.x += red[k][li].x; v.y += red[k][li].y; v.z += red[k][li].z; v.w += red[k][li].w; }
    const int lane = f & 63, gq = (f >> 6) & 3, wave = f >> 8, il = lane & 31, kl = lane >> 5;
    const int col = wave * 32 + il, tap = col >> 4, co = col & 15, row = 8 * gq + 4 * kl;
    float* const o = dwp + ((size_t)row * 16 + tap) * cpad + co;
    const size_t rs = (size_t)16 * cpad;
    o[0] += v.x; o[rs] += v.y; o[2 * rs] += v.z; o[3 * rs] += v.w;
  }
}
__global__ __launch_bounds__(256) void wgrad_reduce_partials_t4_kernel(const float4* __restrict__ part, const int S, const int cpad,
                                                                       float* __restrict__ dwp) {
  wgrad_reduce_partials_t4_body(part, S, cpad, dwp, (int)blockIdx.x, (int)blockIdx.y);
}

// ------------------------------------------------------------------------------------------------
// ... and of the decoders' FIRST ConvTranspose2d (k = 4, stride 2, pad 1, 64 -> 32 channels; input planes of 64 pixels: 8x8
// vision, 16x4 audio).  32 src channels per tap make every 32-column tile ONE tap with lane = src channel, so the column
// shift is wave-uniform: wave w holds taps 2 w (odd image: kx = 0 shifted by one with a zero in front, kx = 2 as it stands)
// and 2 w + 1 (even image: kx = 1 as it stands, kx = 3 shifted the other way) for both 32-row tiles of the 64 ci = four
// accumulator tiles.  A lane's 8 pixels are one whole row (8x8) or two whole rows (16x4: two 8-byte reads two image rows
// apart), so no neighbour element is needed.  Partial set: float4 number ((w * 4 + tile) * 4 + r / 4) * 64 + lane,
// tile = 2 (row tile) + (tap & 1).
// ------------------------------------------------------------------------------------------------
constexpr int kWgT4bSetFloats = 8 * 4 * 4 * 64 * 4;
template <int WO>
__host__ __device__ constexpr int wgt4b_cip() { return (((64 / WO * 2 + 2) * (2 * WO)) / 16 | 1) * 16; }
template <int SPLIT, int WO>
__host__ __device__ constexpr int wgt4b_lds_bytes() { return SPLIT * (2 * 32 * wgt4b_cip<WO>() + 64 * kWgresAPitch); }

template <int SPLIT, int WO>
__global__ __launch_bounds__(512, 1) void convt4s2b_wgrad_staged_kernel(
    const MtrssmConvGeom g, const float* __restrict__ a, const float* __restrict__ src, const int pre_act_a, float* __restrict__ dwp,
    float* __restrict__ part, const int frames_per_wg) {
  static_assert(SPLIT == 1 || SPLIT == 2, "one or two bf16 pieces");
  static_assert(WO == 8 || WO == 4, "input plane 8x8 or 16x4");
  constexpr int NT = 512, C = 32, CO = 64, HO = 64 / WO, HS = 2 * HO, WS = 2 * WO;
  constexpr int ROWB = WS;                 // bytes per row of one parity image: WS / 2 bf16
  constexpr int CIP = wgt4b_cip<WO>();     // bytes per channel: (HS + 2) rows, padded
  constexpr int XCOPY = C * CIP;           // one parity image of one piece
  constexpr int AP = kWgresAPitch, APB = CO * AP;  // `a` image: [ci][64 pixels + 8] bf16
  constexpr int XI = C * HS * WS / 4 / NT, AI = CO * 64 / 4 / NT, NI = XI + AI;  // 4 + 2 float4 items per thread and frame
  extern __shared__ __attribute__((aligned(16))) unsigned char wgt4b_lds[];
  unsigned char* const lds = wgt4b_lds;    // [piece][even image][odd image] then [piece][a image]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int il = lane & 31, kl = lane >> 5;
  const int n0 = blockIdx.x * frames_per_wg;
  const int n1 = n0 + frames_per_wg < g.N ? n0 + frames_per_wg : g.N;
  if (n0 >= n1) return;  // whole workgroup
  const int nlast = n1 - 1;

  const bool act_elu = g.act == MTRSSM_ACT_ELU, act_relu = g.act == MTRSSM_ACT_RELU, pre = pre_act_a != 0;
  auto act_sel = [&](float x) __attribute__((always_inline)) {
    float e = __expf(x) - 1.f;
    asm volatile("" : "+v"(e));  // computed unconditionally: the compiler would branch around the exponential
    const float neg = act_elu ? e : (act_relu ? 0.f : x);
    return (x > 0.f || !pre) ? x : neg;
  };

  for (int o = tid * 16; o < wgt4b_lds_bytes<SPLIT, WO>(); o += NT * 16) *reinterpret_cast<uint4*>(lds + o) = make_uint4(0u, 0u, 0u, 0u);

  const float4* const xsrc = reinterpret_cast<const float4*>(src) + tid;
  const float4* const asrc = reinterpret_cast<const float4*>(a) + tid;
  constexpr size_t xfr = (size_t)C * HS * WS / 4, afr = (size_t)CO * 64 / 4;  // float4 per frame
  wg_f32x4 raw[3][NI];  // [set][a items, src items]
  auto raw_load = [&](const int rs, const int n) __attribute__((always_inline)) {
#pragma unroll
    for (int it = 0; it < NI; ++it) {
      const float4* const ptr = it < AI ? asrc + (size_t)n * afr + NT * it : xsrc + (size_t)n * xfr + NT * (it - AI);
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(raw[rs][it]) : "v"(ptr));
    }
  };
  auto raw_wait = [&](const int rs) __attribute__((always_inline)) {  // the two younger sets may still be in flight
    static_assert(NI == 6, "six items");
    asm volatile("s_waitcnt vmcnt(12)" : "+v"(raw[rs][0]), "+v"(raw[rs][1]), "+v"(raw[rs][2]), "+v"(raw[rs][3]), "+v"(raw[rs][4]), "+v"(raw[rs][5]));
  };
  auto stage = [&](const int rs) __attribute__((always_inline)) {
#pragma unroll
    for (int it = 0; it < NI; ++it) {
      const wg_f32x4 v = raw[rs][it];
      if (it < AI) {
        unsigned d0[SPLIT], d1[SPLIT];
        wg_split_pair<SPLIT>(act_sel(v.x), act_sel(v.y), d0);
        wg_split_pair<SPLIT>(act_sel(v.z), act_sel(v.w), d1);
        const int f = tid + NT * it;  // ci = f / 16, pixel = 4 (f % 16)
        const unsigned o = (unsigned)(SPLIT * 2 * XCOPY) + (unsigned)((f >> 4) * AP + (f & 15) * 8);
#pragma unroll
        for (int p = 0; p < SPLIT; ++p) *reinterpret_cast<uint2*>(lds + o + p * APB) = make_uint2(d0[p], d1[p]);
      } else {
        unsigned de[SPLIT], dd[SPLIT];  // even columns (c0, c0 + 2), odd columns (c0 + 1, c0 + 3)
        wg_split_pair<SPLIT>(v.x, v.z, de);
        wg_split_pair<SPLIT>(v.y, v.w, dd);
        const int f = tid + NT * (it - AI);
        const int co = f / (HS * WS / 4), rem = f - co * (HS * WS / 4), row = rem / (WS / 4), c4 = rem - row * (WS / 4);
        const unsigned o = (unsigned)(co * CIP + (row + 1) * ROWB + c4 * 4);
#pragma unroll
        for (int p = 0; p < SPLIT; ++p) {
          *reinterpret_cast<unsigned*>(lds + o + p * 2 * XCOPY) = de[p];
          *reinterpret_cast<unsigned*>(lds + o + p * 2 * XCOPY + XCOPY) = dd[p];
        }
      }
    }
  };

  // taps 2 w (odd image) and 2 w + 1 (even image): ky = w >> 1 for both, kx = 2 (w & 1) and 2 (w & 1) + 1; lane = src channel
  const int ky = wave >> 1;
  const bool hi = (wave & 1) != 0;  // kx in {2, 3}: the odd-image tap as it stands, the even-image tap shifted towards x + 1
  // k-step s covers input pixels 16 s .. 16 s + 15, this lane 8 of them: WO = 8: row 2 s + kl; WO = 4: rows 4 s + 2 kl, + 1
  const unsigned lane_b = (unsigned)(il * CIP + ky * ROWB + kl * (WO == 8 ? 2 * ROWB : 4 * ROWB));
  constexpr unsigned kStepB = WO == 8 ? 4 * ROWB : 8 * ROWB;
  const unsigned lane_a = (unsigned)(SPLIT * 2 * XCOPY + il * AP + kl * 16);
  f32x16 acc[4];  // [row tile * 2 + (tap & 1)]
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  auto ab = [](const unsigned h, const unsigned l) __attribute__((always_inline)) { return __builtin_amdgcn_alignbit(h, l, 16); };
  auto shr1 = [&](const u32x4 f) __attribute__((always_inline)) {  // element x <- x - 1, zero into every row's first element
    return WO == 8 ? u32x4{f.x << 16, ab(f.y, f.x), ab(f.z, f.y), ab(f.w, f.z)} : u32x4{f.x << 16, ab(f.y, f.x), f.z << 16, ab(f.w, f.z)};
  };
  auto shl1 = [&](const u32x4 f) __attribute__((always_inline)) {  // element x <- x + 1, zero into every row's last element
    return WO == 8 ? u32x4{ab(f.y, f.x), ab(f.z, f.y), ab(f.w, f.z), f.w >> 16} : u32x4{ab(f.y, f.x), f.y >> 16, ab(f.w, f.z), f.w >> 16};
  };
  auto read_frag = [&](const unsigned off) __attribute__((always_inline)) {
    if (WO == 8) return *reinterpret_cast<const u32x4*>(lds + off);
    const uint2 r0 = *reinterpret_cast<const uint2*>(lds + off), r1 = *reinterpret_cast<const uint2*>(lds + off + 2 * ROWB);
    return u32x4{r0.x, r0.y, r1.x, r1.y};
  };

  raw_load(0, n0);
  raw_load(1, n0 + 1 < nlast ? n0 + 1 : nlast);
  auto frame = [&](const int n, const int rs) __attribute__((always_inline)) {  // rs = (n - n0) % 3, a literal at the call sites
    raw_load((rs + 2) % 3, n + 2 < nlast ? n + 2 : nlast);  // its set held frame n - 1, staged a frame ago
    raw_wait(rs);
    lds_barrier();  // every wave is done reading the previous frame's images (first frame: the zeroes are in place)
    stage(rs);
    lds_barrier();  // images complete
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      u32x4 qa[2][SPLIT], qb[2][SPLIT];
#pragma unroll
      for (int p = 0; p < SPLIT; ++p) {
        qa[0][p] = *reinterpret_cast<const u32x4*>(lds + lane_a + (unsigned)(p * APB) + (unsigned)(s * 32));
        qa[1][p] = *reinterpret_cast<const u32x4*>(lds + lane_a + (unsigned)(32 * AP + p * APB) + (unsigned)(s * 32));
        const unsigned bo = lane_b + (unsigned)(p * 2 * XCOPY) + (unsigned)s * kStepB;
        const u32x4 fe = read_frag(bo), fo = read_frag(bo + XCOPY);
        const u32x4 fos = shr1(fo), fes = shl1(fe);
        qb[0][p] = hi ? fo : fos;   // tap 2 w: kx = 2 : kx = 0
        qb[1][p] = hi ? fes : fe;   // tap 2 w + 1: kx = 3 : kx = 1
      }
#pragma unroll
      for (int ord = SPLIT - 1; ord >= 0; --ord)
#pragma unroll
        for (int sa = 0; sa <= ord; ++sa)
#pragma unroll
          for (int t = 0; t < 4; ++t)
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, qa[t >> 1][sa]), __builtin_bit_cast(bf16x8, qb[t & 1][ord - sa]), acc[t], 0, 0, 0);
    }
  };
#pragma unroll 1
  for (int n = n0; n < n1; n += 3) {
    frame(n, 0);
    if (n + 1 < n1) frame(n + 1, 1);
    if (n + 2 < n1) frame(n + 2, 2);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the clamped requests of the last frames

  if (part) {
    float4* const ps = reinterpret_cast<float4*>(part) + (size_t)blockIdx.x * (kWgT4bSetFloats / 4) + (size_t)wave * 1024 + lane;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) ps[(t * 4 + gq) * 64] = make_float4(acc[t][4 * gq], acc[t][4 * gq + 1], acc[t][4 * gq + 2], acc[t][4 * gq + 3]);
  } else {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (t >> 1) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kl;
        atomicAdd(&dwp[((size_t)row * 16 + 2 * wave + (t & 1)) * g.Cpad + il], acc[t][r]);
      }
  }
}

// Partial sets of convt4s2b_wgrad_staged_kernel into dwp.
__device__ __forceinline__ void wgrad_reduce_partials_t4b_body(const float4* __restrict__ part, const int S, const int cpad,
                                                                        float* __restrict__ dwp, const int bx, const int by) {
  constexpr int SET4 = kWgT4bSetFloats / 4, NL = 8, NG = 32;
  __shared__ float4 red[NG][NL];
  const int li = threadIdx.x & (NL - 1), sg = threadIdx.x / NL;
  const int f = bx * NL + li;  // host: grid.x * 8 == SET4
  const float4* const p = part + f;
  float4 acc4[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) acc4[u] = make_float4(0.f, 0.f, 0.f, 0.f);
  int s = sg;
  for (; s + 3 * NG < S; s += 4 * NG) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float4 v = p[(size_t)(s + NG * u) * SET4];
      acc4[u].x += v.x; acc4[u].y += v.y; acc4[u].z += v.z; acc4[u].w += v.w;
    }
  }
  for (; s < S; s += NG) {
    const float4 v = p[(size_t)s * SET4];
    acc4[0].x += v.x; acc4[0].y += v.y; acc4[0].z += v.z; acc4[0].w += v.w;
  }
  red[sg][li] = make_float4((acc4[0].x + acc4[1].x) + (acc4[2].x + acc4[3].x), (acc4[0].y + acc4[1].y) + (acc4[2].y + acc4[3].y),
                            (acc4[0].z + acc4[1].z) + (acc4[2].z + acc4[3].z), (acc4[0].w + acc4[1].w) + (acc4[2].w + acc4[3].w));
  __syncthreads();
  if (sg == 0) {
    float4 v = red[0][li];
#pragma unroll
    for (int k = 1; k < NG; ++k) { v.x += red[k][li].x; v.y += red[k][li].y; v.z += red[k][li].z; v.w += red[k][li].w; }
    const int lane = f & 63, gq = (f >> 6) & 3, t = (f >> 8) & 3, wave = f >> 10, il = lane & 31, kl = lane >> 5;
    const int tap = 2 * wave + (t & 1), row = (t >> 1) * 32 + 8 * gq + 4 * kl;
    float* const o = dwp + ((size_t)row * 16 + tap) * cpad + il;
    const size_t rs = (size_t)16 * cpad;
    o[0] += v.x; o[rs] += v.y; o[2 * rs] += v.z; o[3 * rs] += v.w;
  }
}
__global__ __launch_bounds__(256) void wgrad_reduce_partials_t4b_kernel(const float4* __restrict__ part, const int S, const int cpad,
                                                                        float* __restrict__ dwp) {
  wgrad_reduce_partials_t4b_body(part, S, cpad, dwp, (int)blockIdx.x, (int)blockIdx.y);
}

// ------------------------------------------------------------------------------------------------
// Weight gradient of the FIRST encoder layer (3x3 / stride 2 / pad 1; one frame channel + two frame-independent
// coordinate channels -> 8; 1024-pixel output planes: 32x32 vision, 64x16 audio), the recipe of
// conv3x3s2_wgrad_staged_kernel on a thin layer:
//   dW[co][ky][kx][c] = sum over frames n and output pixels of  a[n][co][oy][ox] * act(src ++ coords)[n][c][2 oy + ky - 1][2 ox + kx - 1]
// 27 (tap, channel) columns = one MFMA tile with 8 of its 32 rows used; all eight waves hold that tile for eight of the
// frame's 64 k-steps each (the reduce kernel adds the eight).  The coordinate channels are staged once per workgroup.
// Partial set: float4 number w * 64 + lane (rows 4 kl .. + 3 of column il), + 8 bias sums.
// ------------------------------------------------------------------------------------------------
constexpr int kWgThinSetFloats = 8 * 64 * 4 + 8;
template <int WO>
__host__ __device__ constexpr int wgthin_cip() { return (((1024 / WO * 2 + 1) * (2 * WO)) / 16 | 1) * 16; }
template <int SPLIT, int WO>
__host__ __device__ constexpr int wgthin_lds_bytes() { return SPLIT * (2 * 3 * wgthin_cip<WO>() + 8 * 2064) + 128; }

template <int SPLIT, int WO>
__global__ __launch_bounds__(512, 1) void conv3x3s2_thin_wgrad_staged_kernel(
    const MtrssmConvGeom g, const float* __restrict__ a, const float* __restrict__ src, const float* __restrict__ coords,
    float* __restrict__ dwp, float* __restrict__ part, float* __restrict__ dbias, const int frames_per_wg) {
  static_assert(SPLIT == 1 || SPLIT == 2, "one or two bf16 pieces");
  static_assert(WO == 32 || WO == 16, "output plane 32x32 or 64x16");
  constexpr int NT = 512, CH = 3, CO = 8, HO = 1024 / WO, HS = 2 * HO, WS = 2 * WO;
  constexpr int ROWB = WS;                 // bytes per row of one parity image: WS / 2 bf16
  constexpr int CIP = wgthin_cip<WO>();    // bytes per channel: (HS + 1) rows, padded
  constexpr int XCOPY = CH * CIP;          // one parity image of one piece: [frame channel, coordinate y, coordinate x]
  constexpr int AP = 2064, APB = CO * AP;  // `a` image: [co][1024 pixels + 8] bf16
  constexpr int XI = HS * WS / 4 / NT, AI = CO * 1024 / 4 / NT, NI = XI + AI;  // 2 + 4 float4 items per thread and frame
  constexpr int LDSB = SPLIT * (2 * XCOPY + APB);
  extern __shared__ __attribute__((aligned(16))) unsigned char wgthin_lds[];
  unsigned char* const lds = wgthin_lds;   // [piece][even image][odd image], [piece][a image], 32 floats of bias scratch
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int il = lane & 31, kl = lane >> 5;
  const int n0 = blockIdx.x * frames_per_wg;
  const int n1 = n0 + frames_per_wg < g.N ? n0 + frames_per_wg : g.N;
  if (n0 >= n1) return;  // whole workgroup
  const int nlast = n1 - 1;

  const bool act_elu = g.act == MTRSSM_ACT_ELU, act_relu = g.act == MTRSSM_ACT_RELU, pre = g.pre_act != 0;
  auto act_sel = [&](float x) __attribute__((always_inline)) {
    float e = __expf(x) - 1.f;
    asm volatile("" : "+v"(e));  // computed unconditionally: the compiler would branch around the exponential
    const float neg = act_elu ? e : (act_relu ? 0.f : x);
    return (x > 0.f || !pre) ? x : neg;
  };

  for (int o = tid * 16; o < wgthin_lds_bytes<SPLIT, WO>(); o += NT * 16) *reinterpret_cast<uint4*>(lds + o) = make_uint4(0u, 0u, 0u, 0u);

  // a src / coordinate float4 = 4 consecutive columns c0 .. c0 + 3 of one row: even columns to the even image, odd to the odd one
  auto stage_src = [&](const float4 v, const int ch, const int f) __attribute__((always_inline)) {
    unsigned de[SPLIT], dd[SPLIT];
    wg_split_pair<SPLIT>(act_sel(v.x), act_sel(v.z), de);
    wg_split_pair<SPLIT>(act_sel(v.y), act_sel(v.w), dd);
    const int row = f / (WS / 4), c4 = f - row * (WS / 4);
    const unsigned o = (unsigned)(ch * CIP + (row + 1) * ROWB + c4 * 4);
#pragma unroll
    for (int p = 0; p < SPLIT; ++p) {
      *reinterpret_cast<unsigned*>(lds + o + p * 2 * XCOPY) = de[p];
      *reinterpret_cast<unsigned*>(lds + o + p * 2 * XCOPY + XCOPY) = dd[p];
    }
  };

  const float4* const xsrc = reinterpret_cast<const float4*>(src) + tid;
  const float4* const asrc = reinterpret_cast<const float4*>(a) + tid;
  constexpr size_t xfr = (size_t)HS * WS / 4, afr = (size_t)CO * 1024 / 4;  // float4 per frame
  float bsum[AI];
#pragma unroll
  for (int j = 0; j < AI; ++j) bsum[j] = 0.f;
  wg_f32x4 raw[3][NI];  // [set][a items, src items]
  auto raw_load = [&](const int rs, const int n) __attribute__((always_inline)) {
#pragma unroll
    for (int it = 0; it < NI; ++it) {
      const float4* const ptr = it < AI ? asrc + (size_t)n * afr + NT * it : xsrc + (size_t)n * xfr + NT * (it - AI);
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(raw[rs][it]) : "v"(ptr));
    }
  };
  auto raw_wait = [&](const int rs) __attribute__((always_inline)) {  // the two younger sets may still be in flight
    static_assert(NI == 6, "six items");
    asm volatile("s_waitcnt vmcnt(12)" : "+v"(raw[rs][0]), "+v"(raw[rs][1]), "+v"(raw[rs][2]), "+v"(raw[rs][3]), "+v"(raw[rs][4]), "+v"(raw[rs][5]));
  };
  auto stage = [&](const int rs) __attribute__((always_inline)) {
#pragma unroll
    for (int it = 0; it < NI; ++it) {
      const wg_f32x4 v = raw[rs][it];
      if (it < AI) {
        bsum[it] += (v.x + v.y) + (v.z + v.w);
        unsigned d0[SPLIT], d1[SPLIT];
        wg_split_pair<SPLIT>(v.x, v.y, d0);
        wg_split_pair<SPLIT>(v.z, v.w, d1);
        const int f = tid + NT * it;  // co = f / 256, pixel = 4 (f % 256)
        const unsigned o = (unsigned)(SPLIT * 2 * XCOPY) + (unsigned)((f >> 8) * AP + (f & 255) * 8);
#pragma unroll
        for (int p = 0; p < SPLIT; ++p) *reinterpret_cast<uint2*>(lds + o + p * APB) = make_uint2(d0[p], d1[p]);
      } else {
        stage_src(make_float4(v.x, v.y, v.z, v.w), 0, tid + NT * (it - AI));
      }
    }
  };

  // the lane's column: (tap, channel) = (il / 3, il % 3); lanes 27 .. 31 are padding (computed on column 26, never stored)
  const int colc = il < 27 ? il : 26;
  const int tap = colc / 3, ch = colc - 3 * tap, ky = tap / 3, kx = tap - 3 * ky;
  const unsigned lane_b = (unsigned)((kx == 1 ? 0 : XCOPY) + ch * CIP + ky * ROWB + kl * 16);
  const unsigned lane_a = (unsigned)(SPLIT * 2 * XCOPY + (il & 7) * AP + kl * 16);
  const bool shifted = kx == 0;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;

  raw_load(0, n0);
  raw_load(1, n0 + 1 < nlast ? n0 + 1 : nlast);
  lds_barrier();  // the zeroes are in place
  // the coordinate channels, once: 2 x HS x WS floats = 4 float4 per thread
#pragma unroll
  for (int j = 0; j < 2 * XI; ++j) {
    const int f = tid + NT * j;  // channel f / (HS WS / 4)
    stage_src(reinterpret_cast<const float4*>(coords)[f], 1 + f / (HS * WS / 4), f % (HS * WS / 4));
  }
  auto frame = [&](const int n, const int rs) __attribute__((always_inline)) {  // rs = (n - n0) % 3, a literal at the call sites
    raw_load((rs + 2) % 3, n + 2 < nlast ? n + 2 : nlast);  // its set held frame n - 1, staged a frame ago
    raw_wait(rs);
    lds_barrier();  // every wave is done reading the previous frame's images
    stage(rs);
    lds_barrier();  // images complete
#pragma unroll
    for (int ss = 0; ss < 8; ++ss) {
      const int s = wave * 8 + ss;  // k-step: output pixels 16 s .. 16 s + 15; WO = 32: row s / 2, half s & 1; WO = 16: row s
      const unsigned soff = WO == 32 ? (unsigned)((s >> 1) * 2 * ROWB + (s & 1) * 32) : (unsigned)(s * 2 * ROWB);
      const bool has_prev = WO == 32 ? ((s & 1) | kl) != 0 : kl != 0;
      u32x4 qa[SPLIT], qb[SPLIT];
#pragma unroll
      for (int p = 0; p < SPLIT; ++p) {
        qa[p] = *reinterpret_cast<const u32x4*>(lds + lane_a + (unsigned)(p * APB) + (unsigned)(s * 32));
        const unsigned bo = lane_b + (unsigned)(p * 2 * XCOPY) + soff;
        const u32x4 f = *reinterpret_cast<const u32x4*>(lds + bo);
        const unsigned prev = has_prev ? (unsigned)*reinterpret_cast<const unsigned short*>(lds + bo - 2) : 0u;
        const u32x4 sh = u32x4{(f.x << 16) | prev, __builtin_amdgcn_alignbit(f.y, f.x, 16), __builtin_amdgcn_alignbit(f.z, f.y, 16), __builtin_amdgcn_alignbit(f.w, f.z, 16)};
        qb[p] = shifted ? sh : f;
      }
#pragma unroll
      for (int ord = SPLIT - 1; ord >= 0; --ord)
#pragma unroll
        for (int sa = 0; sa <= ord; ++sa)
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, qa[sa]), __builtin_bit_cast(bf16x8, qb[ord - sa]), acc, 0, 0, 0);
    }
  };
#pragma unroll 1
  for (int n = n0; n < n1; n += 3) {
    frame(n, 0);
    if (n + 1 < n1) frame(n + 1, 1);
    if (n + 2 < n1) frame(n + 2, 2);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the clamped requests of the last frames

  // bias sums: `a` item j of this thread belongs to channel tid / 256 + 2 j: four waves per channel, met in LDS
  float* const bred = reinterpret_cast<float*>(lds + LDSB);  // [item][wave]
  if (dbias != nullptr) {
    lds_barrier();  // the images are dead
#pragma unroll
    for (int j = 0; j < AI; ++j) {
      const float v = wave_sum(bsum[j]);
      if (lane == 0) bred[j * 8 + wave] = v;
    }
    lds_barrier();
  }
  if (part) {
    float* const set = part + (size_t)blockIdx.x * kWgThinSetFloats;
    reinterpret_cast<float4*>(set)[wave * 64 + lane] = make_float4(acc[0], acc[1], acc[2], acc[3]);  // rows 4 kl + 0 .. 3
    if (dbias != nullptr && tid < 8) {
      const int j = tid >> 1, w0 = 4 * (tid & 1);  // channel tid = w0 / 4 + 2 j
      set[8 * 64 * 4 + tid] = (bred[j * 8 + w0] + bred[j * 8 + w0 + 1]) + (bred[j * 8 + w0 + 2] + bred[j * 8 + w0 + 3]);
    }
  } else {
    if (il < 27) {
#pragma unroll
      for (int r = 0; r < 4; ++r) atomicAdd(&dwp[((size_t)(4 * kl + r) * 9 + tap) * g.Cpad + ch], acc[r]);
    }
    if (dbias != nullptr && tid < 8) {
      const int j = tid >> 1, w0 = 4 * (tid & 1);
      atomicAdd(&dbias[tid], (bred[j * 8 + w0] + bred[j * 8 + w0 + 1]) + (bred[j * 8 + w0 + 2] + bred[j * 8 + w0 + 3]));
    }
  }
}

// Partial sets of conv3x3s2_thin_wgrad_staged_kernel into dwp / dbias: one tile, eight k-parts per workgroup.
__device__ __forceinline__ void wgrad_reduce_partials_thin_body(const float4* __restrict__ part, const int S, const int cpad,
                                                                         float* __restrict__ dwp, float* __restrict__ dbias, const int bx, const int by) {
  constexpr int SET4 = kWgThinSetFloats / 4, NL = 8, NG = 32;
  __shared__ float4 red[NG][NL];
  const int li = threadIdx.x & (NL - 1), sg = threadIdx.x / NL;
  const int f = bx * NL + li;  // lane's float4 (64 of them: 8 blocks), then the bias block (2 float4 used)
  const bool bias = f >= 64;           // block-uniform
  float4 acc4[2];
  acc4[0] = acc4[1] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (!(bias && f >= 66)) {
    const int ks = bias ? 1 : 8;
    const int slices = S * ks;  // slice i = (workgroup i / ks, wave i % ks)
    const float4* const p = part + (bias ? (size_t)(8 * 64 + (f - 64)) : (size_t)f);
    int i = sg;
    for (; i + NG < slices; i += 2 * NG) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int ii = i + NG * u;
        const float4 v = p[(size_t)(ii / ks) * SET4 + (size_t)(ii % ks) * 64];
        acc4[u].x += v.x; acc4[u].y += v.y; acc4[u].z += v.z; acc4[u].w += v.w;
      }
    }
    for (; i < slices; i += NG) {
      const float4 v = p[(size_t)(i / ks) * SET4 + (size_t)(i % ks) * 64];
      acc4[0].x += v.x; acc4[0].y += v.y; acc4[0].z += v.z; acc4[0].w += v.w;
    }
  }
  red[sg][li] = make_float4(acc4[0].x + acc4[1].x, acc4[0].y + acc4[1].y, acc4[0].z + acc4[1].z, acc4[0].w + acc4[1].w);
  __syncthreads();
  if (sg == 0) {
    float4 v = red[0][li];
#pragma unroll
    for (int k = 1; k < NG; ++k) { v.x += red[k][li].x; v.y += red[k][li].y; v.z += red[k][li].z; v.w += red[k][li].w; }
    if (bias) {
      if (f < 66) { float* const o = dbias + 4 * (f - 64); o[0] += v.x; o[1] += v.y; o[2] += v.z; o[3] += v.w; }
      return;
    }
    const int il = f & 31, kl = f >> 5;
    if (il < 27) {
      const int tap = il / 3, ch = il - 3 * tap;
      float* const o = dwp + ((size_t)(4 * kl) * 9 + tap) * cpad + ch;
      const size_t rs = (size_t)9 * cpad;
      o[0] += v.x; o[rs] += v.y; o[2 * rs] += v.z; o[3 * rs] += v.w;
    }
  }
}
__global__ __launch_bounds__(256) void wgrad_reduce_partials_thin_kernel(const float4* __restrict__ part, const int S, const int cpad,
                                                                         float* __restrict__ dwp, float* __restrict__ dbias) {
  wgrad_reduce_partials_thin_body(part, S, cpad, dwp, dbias, (int)blockIdx.x, (int)blockIdx.y);
}

// ------------------------------------------------------------------------------------------------
// Weight gradient of the decoders' LAST ConvTranspose2d (k = 4, stride 2, pad 1, 16 -> 1 channel; input planes of 1024
// pixels: 32x32 vision, 64x16 audio): convt4s2_wgrad_staged_kernel on a thin layer.  `a` is the layer input (16 channels,
// activated here), src the one-channel output gradient; 16 (tap) columns and 16 rows = a quarter of one MFMA tile, held by
// all eight waves for eight of the frame's 64 k-steps each.  Partial set: float4 number (w * 2 + r / 4) * 64 + lane, r < 8.
// ------------------------------------------------------------------------------------------------
constexpr int kWgThinTSetFloats = 8 * 2 * 64 * 4;
template <int WO>
__host__ __device__ constexpr int wgthint_cip() { return (((1024 / WO * 2 + 2) * (2 * WO)) / 16 | 1) * 16; }
template <int SPLIT, int WO>
__host__ __device__ constexpr int wgthint_lds_bytes() { return SPLIT * (2 * wgthint_cip<WO>() + 16 * 2064); }

template <int SPLIT, int WO>
__global__ __launch_bounds__(512, 1) void convt4s2_thin_wgrad_staged_kernel(
    const MtrssmConvGeom g, const float* __restrict__ a, const float* __restrict__ src, const int pre_act_a, float* __restrict__ dwp,
    float* __restrict__ part, const int frames_per_wg) {
  static_assert(SPLIT == 1 || SPLIT == 2, "one or two bf16 pieces");
  static_assert(WO == 32 || WO == 16, "input plane 32x32 or 64x16");
  constexpr int NT = 512, CO = 16, HO = 1024 / WO, HS = 2 * HO, WS = 2 * WO;
  constexpr int ROWB = WS;                  // bytes per row of one parity image: WS / 2 bf16
  constexpr int XCOPY = wgthint_cip<WO>();  // one parity image of one piece: (HS + 2) rows, padded
  constexpr int AP = 2064, APB = CO * AP;   // `a` image: [ci][1024 pixels + 8] bf16
  constexpr int XI = HS * WS / 4 / NT, AI = CO * 1024 / 4 / NT, NI = XI + AI;  // 2 + 8 float4 items per thread and frame
  extern __shared__ __attribute__((aligned(16))) unsigned char wgthint_lds[];
  unsigned char* const lds = wgthint_lds;   // [piece][even image][odd image] then [piece][a image]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int il = lane & 31, kl = lane >> 5;
  const int n0 = blockIdx.x * frames_per_wg;
  const int n1 = n0 + frames_per_wg < g.N ? n0 + frames_per_wg : g.N;
  if (n0 >= n1) return;  // whole workgroup
  const int nlast = n1 - 1;

  const bool act_elu = g.act == MTRSSM_ACT_ELU, act_relu = g.act == MTRSSM_ACT_RELU, pre = pre_act_a != 0;
  auto act_sel = [&](float x) __attribute__((always_inline)) {
    float e = __expf(x) - 1.f;
    asm volatile("" : "+v"(e));  // computed unconditionally: the compiler would branch around the exponential
    const float neg = act_elu ? e : (act_relu ? 0.f : x);
    return (x > 0.f || !pre) ? x : neg;
  };

  for (int o = tid * 16; o < wgthint_lds_bytes<SPLIT, WO>(); o += NT * 16) *reinterpret_cast<uint4*>(lds + o) = make_uint4(0u, 0u, 0u, 0u);

  const float4* const xsrc = reinterpret_cast<const float4*>(src) + tid;
  const float4* const asrc = reinterpret_cast<const float4*>(a) + tid;
  constexpr size_t xfr = (size_t)HS * WS / 4, afr = (size_t)CO * 1024 / 4;  // float4 per frame
  wg_f32x4 raw[3][NI];  // [set][a items, src items]
  auto raw_load = [&](const int rs, const int n) __attribute__((always_inline)) {
#pragma unroll
    for (int it = 0; it < NI; ++it) {
      const float4* const ptr = it < AI ? asrc + (size_t)n * afr + NT * it : xsrc + (size_t)n * xfr + NT * (it - AI);
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(raw[rs][it]) : "v"(ptr));
    }
  };
  auto raw_wait = [&](const int rs) __attribute__((always_inline)) {  // the two younger sets may still be in flight
    static_assert(NI == 10, "ten items");
    asm volatile("s_waitcnt vmcnt(20)" : "+v"(raw[rs][0]), "+v"(raw[rs][1]), "+v"(raw[rs][2]), "+v"(raw[rs][3]), "+v"(raw[rs][4]), "+v"(raw[rs][5]),
                 "+v"(raw[rs][6]), "+v"(raw[rs][7]), "+v"(raw[rs][8]), "+v"(raw[rs][9]));
  };
  auto stage = [&](const int rs) __attribute__((always_inline)) {
#pragma unroll
    for (int it = 0; it < NI; ++it) {
      const wg_f32x4 v = raw[rs][it];
      if (it < AI) {
        unsigned d0[SPLIT], d1[SPLIT];
        wg_split_pair<SPLIT>(act_sel(v.x), act_sel(v.y), d0);
        wg_split_pair<SPLIT>(act_sel(v.z), act_sel(v.w), d1);
        const int f = tid + NT * it;  // ci = f / 256, pixel = 4 (f % 256)
        const unsigned o = (unsigned)(SPLIT * 2 * XCOPY) + (unsigned)((f >> 8) * AP + (f & 255) * 8);
#pragma unroll
        for (int p = 0; p < SPLIT; ++p) *reinterpret_cast<uint2*>(lds + o + p * APB) = make_uint2(d0[p], d1[p]);
      } else {
        unsigned de[SPLIT], dd[SPLIT];  // even columns (c0, c0 + 2), odd columns (c0 + 1, c0 + 3)
        wg_split_pair<SPLIT>(v.x, v.z, de);
        wg_split_pair<SPLIT>(v.y, v.w, dd);
        const int f = tid + NT * (it - AI);
        const int row = f / (WS / 4), c4 = f - row * (WS / 4);
        const unsigned o = (unsigned)((row + 1) * ROWB + c4 * 4);
#pragma unroll
        for (int p = 0; p < SPLIT; ++p) {
          *reinterpret_cast<unsigned*>(lds + o + p * 2 * XCOPY) = de[p];
          *reinterpret_cast<unsigned*>(lds + o + p * 2 * XCOPY + XCOPY) = dd[p];
        }
      }
    }
  };

  // the lane's column = tap il & 15 (lanes 16 .. 31 repeat the columns, never stored)
  const int tap = il & 15, ky = tap >> 2, kx = tap & 3;
  const bool odd = kx == 0 || kx == 2;
  const unsigned lane_b = (unsigned)((odd ? XCOPY : 0) + ky * ROWB + kl * 16);
  const unsigned lane_a = (unsigned)(SPLIT * 2 * XCOPY + (il & 15) * AP + kl * 16);
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;

  raw_load(0, n0);
  raw_load(1, n0 + 1 < nlast ? n0 + 1 : nlast);
  auto frame = [&](const int n, const int rs) __attribute__((always_inline)) {  // rs = (n - n0) % 3, a literal at the call sites
    raw_load((rs + 2) % 3, n + 2 < nlast ? n + 2 : nlast);  // its set held frame n - 1, staged a frame ago
    raw_wait(rs);
    lds_barrier();  // every wave is done reading the previous frame's images (first frame: the zeroes are in place)
    stage(rs);
    lds_barrier();  // images complete
#pragma unroll
    for (int ss = 0; ss < 8; ++ss) {
      const int s = wave * 8 + ss;  // k-step: input pixels 16 s .. 16 s + 15; WO = 32: row s / 2, half s & 1; WO = 16: row s
      const unsigned soff = WO == 32 ? (unsigned)((s >> 1) * 2 * ROWB + (s & 1) * 32) : (unsigned)(s * 2 * ROWB);
      const bool has_prev = WO == 32 ? ((s & 1) | kl) != 0 : kl != 0;
      const bool has_next = WO == 32 ? ((s & 1) & kl) == 0 : kl == 0;
      u32x4 qa[SPLIT], qb[SPLIT];
#pragma unroll
      for (int p = 0; p < SPLIT; ++p) {
        qa[p] = *reinterpret_cast<const u32x4*>(lds + lane_a + (unsigned)(p * APB) + (unsigned)(s * 32));
        const unsigned bo = lane_b + (unsigned)(p * 2 * XCOPY) + soff;
        const u32x4 f = *reinterpret_cast<const u32x4*>(lds + bo);
        const unsigned prev = has_prev ? (unsigned)*reinterpret_cast<const unsigned short*>(lds + bo - 2) : 0u;
        const unsigned next = has_next ? (unsigned)*reinterpret_cast<const unsigned short*>(lds + bo + 16) : 0u;
        const u32x4 shr = u32x4{(f.x << 16) | prev, __builtin_amdgcn_alignbit(f.y, f.x, 16), __builtin_amdgcn_alignbit(f.z, f.y, 16), __builtin_amdgcn_alignbit(f.w, f.z, 16)};
        const u32x4 shl = u32x4{__builtin_amdgcn_alignbit(f.y, f.x, 16), __builtin_amdgcn_alignbit(f.z, f.y, 16), __builtin_amdgcn_alignbit(f.w, f.z, 16), (f.w >> 16) | (next << 16)};
        qb[p] = kx == 0 ? shr : (kx == 3 ? shl : f);
      }
#pragma unroll
      for (int ord = SPLIT - 1; ord >= 0; --ord)
#pragma unroll
        for (int sa = 0; sa <= ord; ++sa)
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, qa[sa]), __builtin_bit_cast(bf16x8, qb[ord - sa]), acc, 0, 0, 0);
    }
  };
#pragma unroll 1
  for (int n = n0; n < n1; n += 3) {
    frame(n, 0);
    if (n + 1 < n1) frame(n + 1, 1);
    if (n + 2 < n1) frame(n + 2, 2);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the clamped requests of the last frames

  if (part) {
    float4* const ps = reinterpret_cast<float4*>(part) + (size_t)blockIdx.x * (kWgThinTSetFloats / 4) + (size_t)wave * 128 + lane;
    ps[0] = make_float4(acc[0], acc[1], acc[2], acc[3]);   // rows 4 kl + 0 .. 3
    ps[64] = make_float4(acc[4], acc[5], acc[6], acc[7]);  // rows 8 + 4 kl + 0 .. 3
  } else if (il < 16) {
#pragma unroll
    for (int r = 0; r < 8; ++r) atomicAdd(&dwp[((size_t)((r & 3) + 8 * (r >> 2) + 4 * kl) * 16 + tap) * g.Cpad], acc[r]);
  }
}

// Partial sets of convt4s2_thin_wgrad_staged_kernel into dwp: one tile, eight k-parts per workgroup.
__device__ __forceinline__ void wgrad_reduce_partials_thint_body(const float4* __restrict__ part, const int S, const int cpad,
                                                                          float* __restrict__ dwp, const int bx, const int by) {
  constexpr int SET4 = kWgThinTSetFloats / 4, NL = 8, NG = 32;
  __shared__ float4 red[NG][NL];
  const int li = threadIdx.x & (NL - 1), sg = threadIdx.x / NL;
  const int f = bx * NL + li;  // half * 64 + lane: 128 of them (host: 16 blocks)
  const float4* const p = part + f;
  float4 acc4[2];
  acc4[0] = acc4[1] = make_float4(0.f, 0.f, 0.f, 0.f);
  const int slices = S * 8;  // slice i = (workgroup i / 8, wave i % 8)
  int i = sg;
  for (; i + NG < slices; i += 2 * NG) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int ii = i + NG * u;
      const float4 v = p[(size_t)(ii >> 3) * SET4 + (size_t)(ii & 7) * 128];
      acc4[u].x += v.x; acc4[u].y += v.y; acc4[u].z += v.z; acc4[u].w += v.w;
    }
  }
  for (; i < slices; i += NG) {
    const float4 v = p[(size_t)(i >> 3) * SET4 + (size_t)(i & 7) * 128];
    acc4[0].x += v.x; acc4[0].y += v.y; acc4[0].z += v.z; acc4[0].w += v.w;
  }
  red[sg][li] = make_float4(acc4[0].x + acc4[1].x, acc4[0].y + acc4[1].y, acc4[0].z + acc4[1].z, acc4[0].w + acc4[1].w);
  __syncthreads();
  if (sg == 0) {
    float4 v = red[0][li];
#pragma unroll
    for (int k = 1; k < NG; ++k) { v.x += red[k][li].x; v.y += red[k][li].y; v.z += red[k][li].z; v.w += red[k][li].w; }
    const int half = f >> 6, lane = f & 63, il = lane & 31, kl = lane >> 5;
    if (il < 16) {
      float* const o = dwp + ((size_t)(8 * half + 4 * kl) * 16 + il) * cpad;
      const size_t rs = (size_t)16 * cpad;
      o[0] += v.x; o[rs] += v.y; o[2 * rs] += v.z; o[3 * rs] += v.w;
    }
  }
}
__global__ __launch_bounds__(256) void wgrad_reduce_partials_thint_kernel(const float4* __restrict__ part, const int S, const int cpad,
                                                                          float* __restrict__ dwp) {
  wgrad_reduce_partials_thint_body(part, S, cpad, dwp, (int)blockIdx.x, (int)blockIdx.y);
}

// ------------------------------------------------------------------------------------------------
// ... and of the THIRD encoder layer (3x3 / stride 2 / pad 1, 16 -> 32 channels, 64-pixel output planes: 8x8 vision, 16x4
// audio): 144 (tap, ci) columns in five 32-column tiles (waves 0 - 4, the rest only stage), all 32 rows, four k-steps per
// frame.  A lane's 8 output pixels are one whole row (8x8) or two whole rows (16x4: two 8-byte reads two image rows apart),
// so kx = 0 is a funnel shift with zeros.  Partial set: float4 number (w * 4 + r / 4) * 64 + lane, + 32 bias sums.
// ------------------------------------------------------------------------------------------------
constexpr int kWgS2cSetFloats = 8 * 4 * 64 * 4 + 32;
template <int SPLIT>
__host__ __device__ constexpr int wgs2c_lds_bytes() { return SPLIT * (2 * 16 * 272 + 32 * kWgresAPitch); }

template <int SPLIT, int WO>
__global__ __launch_bounds__(512, 1) void conv3x3s2c_wgrad_staged_kernel(
    const MtrssmConvGeom g, const float* __restrict__ a, const float* __restrict__ src, float* __restrict__ dwp,
    float* __restrict__ part, float* __restrict__ dbias, const int frames_per_wg) {
  static_assert(SPLIT == 1 || SPLIT == 2, "one or two bf16 pieces");
  static_assert(WO == 8 || WO == 4, "output plane 8x8 or 16x4");
  constexpr int NT = 512, C = 16, CO = 32, HO = 64 / WO, HS = 2 * HO, WS = 2 * WO;
  constexpr int ROWB = WS;                 // bytes per row of one parity image: WS / 2 bf16
  constexpr int CIP = 272;                 // bytes per channel: (HS + 1) rows (272 / 264 bytes), an odd number of 16-byte slots
  static_assert((HS + 1) * ROWB <= CIP, "channel pitch");
  constexpr int XCOPY = C * CIP;           // one parity image of one piece
  constexpr int AP = kWgresAPitch, APB = CO * AP;  // `a` image: [co][64 pixels + 8] bf16
  constexpr int XI = C * HS * WS / 4 / NT, AI = CO * 64 / 4 / NT, NI = XI + AI;  // 2 + 1 float4 items per thread and frame
  extern __shared__ __attribute__((aligned(16))) unsigned char wgs2c_lds[];
  unsigned char* const lds = wgs2c_lds;    // [piece][even image][odd image] then [piece][a image]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int il = lane & 31, kl = lane >> 5;
  const int n0 = blockIdx.x * frames_per_wg;
  const int n1 = n0 + frames_per_wg < g.N ? n0 + frames_per_wg : g.N;
  if (n0 >= n1) return;  // whole workgroup
  const int nlast = n1 - 1;

  const bool act_elu = g.act == MTRSSM_ACT_ELU, act_relu = g.act == MTRSSM_ACT_RELU, pre = g.pre_act != 0;
  auto act_sel = [&](float x) __attribute__((always_inline)) {
    float e = __expf(x) - 1.f;
    asm volatile("" : "+v"(e));  // computed unconditionally: the compiler would branch around the exponential
    const float neg = act_elu ? e : (act_relu ? 0.f : x);
    return (x > 0.f || !pre) ? x : neg;
  };

  for (int o = tid * 16; o < wgs2c_lds_bytes<SPLIT>(); o += NT * 16) *reinterpret_cast<uint4*>(lds + o) = make_uint4(0u, 0u, 0u, 0u);

  const float4* const xsrc = reinterpret_cast<const float4*>(src) + tid;
  const float4* const asrc = reinterpret_cast<const float4*>(a) + tid;
  constexpr size_t xfr = (size_t)C * HS * WS / 4, afr = (size_t)CO * 64 / 4;  // float4 per frame
  float bsum = 0.f;  // this thread's share of channel tid / 16
  wg_f32x4 raw[3][NI];  // [set][a item, src items]
  auto raw_load = [&](const int rs, const int n) __attribute__((always_inline)) {
#pragma unroll
    for (int it = 0; it < NI; ++it) {
      const float4* const ptr = it < AI ? asrc + (size_t)n * afr + NT * it : xsrc + (size_t)n * xfr + NT * (it - AI);
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(raw[rs][it]) : "v"(ptr));
    }
  };
  auto raw_wait = [&](const int rs) __attribute__((always_inline)) {  // the two younger sets may still be in flight
    static_assert(NI == 3, "three items");
    asm volatile("s_waitcnt vmcnt(6)" : "+v"(raw[rs][0]), "+v"(raw[rs][1]), "+v"(raw[rs][2]));
  };
  auto stage = [&](const int rs) __attribute__((always_inline)) {
#pragma unroll
    for (int it = 0; it < NI; ++it) {
      const wg_f32x4 v = raw[rs][it];
      if (it < AI) {
        bsum += (v.x + v.y) + (v.z + v.w);
        unsigned d0[SPLIT], d1[SPLIT];
        wg_split_pair<SPLIT>(v.x, v.y, d0);
        wg_split_pair<SPLIT>(v.z, v.w, d1);
        const unsigned o = (unsigned)(SPLIT * 2 * XCOPY) + (unsigned)((tid >> 4) * AP + (tid & 15) * 8);
#pragma unroll
        for (int p = 0; p < SPLIT; ++p) *reinterpret_cast<uint2*>(lds + o + p * APB) = make_uint2(d0[p], d1[p]);
      } else {
        unsigned de[SPLIT], dd[SPLIT];  // even columns (c0, c0 + 2), odd columns (c0 + 1, c0 + 3)
        wg_split_pair<SPLIT>(act_sel(v.x), act_sel(v.z), de);
        wg_split_pair<SPLIT>(act_sel(v.y), act_sel(v.w), dd);
        const int f = tid + NT * (it - AI);
        const int ci = f / (HS * WS / 4), rem = f - ci * (HS * WS / 4), row = rem / (WS / 4), c4 = rem - row * (WS / 4);
        const unsigned o = (unsigned)(ci * CIP + (row + 1) * ROWB + c4 * 4);
#pragma unroll
        for (int p = 0; p < SPLIT; ++p) {
          *reinterpret_cast<unsigned*>(lds + o + p * 2 * XCOPY) = de[p];
          *reinterpret_cast<unsigned*>(lds + o + p * 2 * XCOPY + XCOPY) = dd[p];
        }
      }
    }
  };

  // the lane's column: (tap, ci) = (col >> 4, col & 15); columns 144 .. 159 of tile 4 are padding (computed on column 143)
  const int col = wave * 32 + il, colc = col < 144 ? col : 143;
  const int tap = colc >> 4, ci = colc & 15, ky = tap / 3, kx = tap - 3 * ky;
  // k-step s covers output pixels 16 s .. 16 s + 15, this lane 8 of them: WO = 8: row 2 s + kl; WO = 4: rows 4 s + 2 kl, + 1
  const unsigned lane_b = (unsigned)((kx == 1 ? 0 : XCOPY) + ci * CIP + ky * ROWB + kl * (WO == 8 ? 2 * ROWB : 4 * ROWB));
  constexpr unsigned kStepB = WO == 8 ? 4 * ROWB : 8 * ROWB;
  const unsigned lane_a = (unsigned)(SPLIT * 2 * XCOPY + il * AP + kl * 16);
  const bool shifted = kx == 0;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;

  raw_load(0, n0);
  raw_load(1, n0 + 1 < nlast ? n0 + 1 : nlast);
  auto frame = [&](const int n, const int rs) __attribute__((always_inline)) {  // rs = (n - n0) % 3, a literal at the call sites
    raw_load((rs + 2) % 3, n + 2 < nlast ? n + 2 : nlast);  // its set held frame n - 1, staged a frame ago
    raw_wait(rs);
    lds_barrier();  // every wave is done reading the previous frame's images (first frame: the zeroes are in place)
    stage(rs);
    lds_barrier();  // images complete
    if (wave < 5) {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        u32x4 qa[SPLIT], qb[SPLIT];
#pragma unroll
        for (int p = 0; p < SPLIT; ++p) {
          qa[p] = *reinterpret_cast<const u32x4*>(lds + lane_a + (unsigned)(p * APB) + (unsigned)(s * 32));
          const unsigned bo = lane_b + (unsigned)(p * 2 * XCOPY) + (unsigned)s * kStepB;
          u32x4 f;
          if (WO == 8) {
            f = *reinterpret_cast<const u32x4*>(lds + bo);
          } else {
            const uint2 r0 = *reinterpret_cast<const uint2*>(lds + bo), r1 = *reinterpret_cast<const uint2*>(lds + bo + 2 * ROWB);
            f = u32x4{r0.x, r0.y, r1.x, r1.y};
          }
          const u32x4 sh = WO == 8 ? u32x4{f.x << 16, __builtin_amdgcn_alignbit(f.y, f.x, 16), __builtin_amdgcn_alignbit(f.z, f.y, 16), __builtin_amdgcn_alignbit(f.w, f.z, 16)}
                                   : u32x4{f.x << 16, __builtin_amdgcn_alignbit(f.y, f.x, 16), f.z << 16, __builtin_amdgcn_alignbit(f.w, f.z, 16)};
          qb[p] = shifted ? sh : f;
        }
#pragma unroll
        for (int ord = SPLIT - 1; ord >= 0; --ord)
#pragma unroll
          for (int sa = 0; sa <= ord; ++sa)
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, qa[sa]), __builtin_bit_cast(bf16x8, qb[ord - sa]), acc, 0, 0, 0);
      }
    }
  };
#pragma unroll 1
  for (int n = n0; n < n1; n += 3) {
    frame(n, 0);
    if (n + 1 < n1) frame(n + 1, 1);
    if (n + 2 < n1) frame(n + 2, 2);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the clamped requests of the last frames

  float bv = bsum;  // the 16 lanes of a DPP row share channel tid / 16
  bv += dpp_move<0xB1, 0xF>(0.f, bv);
  bv += dpp_move<0x4E, 0xF>(0.f, bv);
  bv += dpp_move<0x141, 0xF>(0.f, bv);
  bv += dpp_move<0x140, 0xF>(0.f, bv);
  if (part) {
    float* const set = part + (size_t)blockIdx.x * kWgS2cSetFloats;
    if (wave < 5) {
      float4* const ps = reinterpret_cast<float4*>(set) + (size_t)wave * 256 + lane;
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) ps[gq * 64] = make_float4(acc[4 * gq], acc[4 * gq + 1], acc[4 * gq + 2], acc[4 * gq + 3]);
    }
    if (dbias != nullptr && (tid & 15) == 0) set[8 * 4 * 64 * 4 + (tid >> 4)] = bv;
  } else {
    if (wave < 5 && col < 144) {
#pragma unroll
      for (int r = 0; r < 16; ++r) atomicAdd(&dwp[((size_t)((r & 3) + 8 * (r >> 2) + 4 * kl) * 9 + tap) * g.Cpad + ci], acc[r]);
    }
    if (dbias != nullptr && (tid & 15) == 0) atomicAdd(&dbias[tid >> 4], bv);
  }
}

// Partial sets of conv3x3s2c_wgrad_staged_kernel into dwp / dbias.
__device__ __forceinline__ void wgrad_reduce_partials_s2c_body(const float4* __restrict__ part, const int S, const int cpad,
                                                                        float* __restrict__ dwp, float* __restrict__ dbias, const int bx, const int by) {
  constexpr int SET4 = kWgS2cSetFloats / 4, NL = 8, NG = 32;
  __shared__ float4 red[NG][NL];
  const int li = threadIdx.x & (NL - 1), sg = threadIdx.x / NL;
  const int f = bx * NL + li;  // float4 of the five column tiles: 1280 (160 blocks), then 8 bias float4 (one block)
  const bool bias = f >= 1280;         // block-uniform
  const float4* const p = part + (bias ? (size_t)(8 * 256 + (f - 1280)) : (size_t)f);
  float4 acc4[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) acc4[u] = make_float4(0.f, 0.f, 0.f, 0.f);
  int s = sg;
  for (; s + 3 * NG < S; s += 4 * NG) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float4 v = p[(size_t)(s + NG * u) * SET4];
      acc4[u].x += v.x; acc4[u].y += v.y; acc4[u].z += v.z; acc4[u].w += v.w;
    }
  }
  for (; s < S; s += NG) {
    const float4 v = p[(size_t)s * SET4];
    acc4[0].x += v.x; acc4[0].y += v.y; acc4[0].z += v.z; acc4[0].w += v.w;
  }
  red[sg][li] = make_float4((acc4[0].x + acc4[1].x) + (acc4[2].x + acc4[3].x), (acc4[0].y + acc4[1].y) + (acc4[2].y + acc4[3].y),
                            (acc4[0].z + acc4[1].z) + (acc4[2].z + acc4[3].z), (acc4[0].w + acc4[1].w) + (acc4[2].w + acc4[3].w));
  __syncthreads();
  if (sg == 0) {
    float4 v = red[0][li];
#pragma unroll
    for (int k = 1; k < NG; ++k) { v.x += red[k][li].x; v.y += red[k][li].y; v.z += red[k][li].z; v.w += red[k][li].w; }
    if (bias) {
      float* const o = dbias + 4 * (f - 1280);
      o[0] += v.x; o[1] += v.y; o[2] += v.z; o[3] += v.w;
      return;
    }
    const int lane = f & 63, gq = (f >> 6) & 3, wave = f >> 8, il = lane & 31, kl = lane >> 5;
    const int col = wave * 32 + il;
    if (col < 144) {
      const int tap = col >> 4, ci = col & 15, row = 8 * gq + 4 * kl;
      float* const o = dwp + ((size_t)row * 9 + tap) * cpad + ci;
      const size_t rs = (size_t)9 * cpad;
      o[0] += v.x; o[rs] += v.y; o[2 * rs] += v.z; o[3 * rs] += v.w;
    }
  }
}
__global__ __launch_bounds__(256) void wgrad_reduce_partials_s2c_kernel(const float4* __restrict__ part, const int S, const int cpad,
                                                                        float* __restrict__ dwp, float* __restrict__ dbias) {
  wgrad_reduce_partials_s2c_body(part, S, cpad, dwp, dbias, (int)blockIdx.x, (int)blockIdx.y);
}

}  // namespace mtrssm
