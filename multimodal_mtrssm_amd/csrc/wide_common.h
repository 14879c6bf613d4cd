// Shared pieces of the chip-wide ("wide") scan kernels (mrssm_wide.hip): gfx950, wave64.
//
// Regime: D, H >= 256 (BASELINE configs[4] "Large": D = H = 1024, S = 128).  One CU re-streaming the step's 42 MB of weights
// per row (mrssm_scan.hip) takes 500 us per timestep there.  Here ALL workgroups of the chip (one per CU) work on the same
// tile of 32 batch rows: every matrix product of the step is cut by OUTPUT column into 16-column tiles, a tile's weights are
// streamed by exactly one CU per step (as bf16 pieces in the MFMA A-operand order, packed once per launch), the 32 batch rows
// are the MFMA N dimension (v_mfma_f32_16x16x32_bf16, two row tiles), and the products of consecutive layers meet through an
// exchange buffer in L2 / MALL followed by a grid-wide barrier (all workgroups resident: grid <= CU count).
//
// Arithmetic: an fp32 value is the sum of P bf16 pieces (P = 3: exact to 2^-24, six MFMA products per k-block -- fp32-grade,
// the default; P = 2: 16 significant bits, three products), fp32 accumulation in the MFMA.
#pragma once
#include "scan_common.h"

namespace mtrssm {

constexpr int kWT = 256;                 // threads per workgroup: one wave per SIMD (512 registers each), the K range of a tile split over them
constexpr int kWW = kWT / kWave;
constexpr int kWRows = 32;               // batch rows per pass: two 16-row MFMA tiles
constexpr unsigned kWideSpinLimit = 1u << 22;

using wbf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using wf32x4 = __attribute__((ext_vector_type(4))) float;

// ---- packed weights: matrix W(n, k), n < N outputs, k < K inputs  ->  [n / 16][k / 32][piece][lane] of 16 bytes:
// lane l holds W(16 nt + (l & 15), 32 ks + 8 (l >> 4) + j), j = 0..7 -- the A operand of v_mfma_f32_16x16x32_bf16, so a
// wave's fragment is ONE contiguous 1 KiB load.  Zero padded.
struct WidePackJob {
  const float* src;
  long sn, sk;       // element strides of n and k in src
  int N, K;          // extent of the block this job fills; element (n, k) of it = src[(n - nskip) sn + (k - kskip) sk] where
  int nskip, kskip;  //   n >= nskip and k >= kskip, zero elsewhere (a source that starts inside a tile / a k-block)
  int NT, KS;        // ceil(N / 16), ceil(K / 32): tiles / k-blocks of the block
  int nt0, ks0, KST; // position of the block in the destination matrix (tile, k-block) and that matrix's k-blocks per tile
  uint4* dst;
};
constexpr int kWideMaxJobs = 16;
struct WidePackJobs {
  WidePackJob j[kWideMaxJobs];
  int count;
};

__host__ __device__ inline size_t wide_pack_uint4(int N, int K, int P) {
  return (size_t)((N + 15) / 16) * ((K + 31) / 32) * P * 64;
}
// ---- exchange vectors: x[row < 32][k < K] as bf16 pieces in the MFMA B-operand order: [piece][k / 32][row][32 k] -- lane l
// of row tile rt reads the 16 bytes at uint4 index ((p KS + ks) 32 + 16 rt + (l & 15)) 4 + (l >> 4): 1 KiB contiguous per wave.
__host__ __device__ inline size_t wide_x_uint4(int K, int P, int rows = kWRows) { return (size_t)P * ((K + 31) / 32) * rows * 4; }

template <int P>
__device__ __forceinline__ void wide_split4(const float (&v)[4], uint2 (&o)[P]) {
  float r[4] = {v[0], v[1], v[2], v[3]};
#pragma unroll
  for (int p = 0; p < P; ++p) {
    unsigned short h[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const __bf16 b = (__bf16)r[i];
      h[i] = __builtin_bit_cast(unsigned short, b);
      r[i] -= (float)b;
    }
    o[p] = make_uint2((unsigned)h[0] | ((unsigned)h[1] << 16), (unsigned)h[2] | ((unsigned)h[3] << 16));
  }
}

// Write-through (sc1) 8-byte store: the exchange payload needs no release fence (cdna_hip_programming.md Guideline 16, R1).
__device__ __forceinline__ void wide_store_u2(void* p, uint2 v) {
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), ((unsigned long long)v.y << 32) | v.x, __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void wide_store_f2(float* p, float a, float b) {
  wide_store_u2(p, make_uint2(__float_as_uint(a), __float_as_uint(b)));
}

// four consecutive k (k % 4 == 0) of one row into an exchange vector
template <int P, int ROWS = kWRows>
__device__ __forceinline__ void wide_x_store4(uint4* xb, int KS, int row, int k, const float (&v)[4]) {
  uint2 o[P];
  wide_split4<P>(v, o);
  char* base = reinterpret_cast<char*>(xb);
#pragma unroll
  for (int p = 0; p < P; ++p)
    wide_store_u2(base + ((size_t)((p * KS + (k >> 5)) * ROWS + row) * 64 + (size_t)(k & 31) * 2), o[p]);
}

// acc[nt][rt] += W_tile[nt] (16 outputs x K) . x (K x 16 rows of row tile rt) over this wave's k-blocks [ks0, ks1).
// wt[nt]: the tile's fragment array ([ks][piece][lane]); xb: exchange vector.  A CU streams these operands from L2 / MALL at
// the rate its loads in flight allow (measured with two k-blocks in flight per wave: 61 GB/s per CU, the latency-bound
// "handed-off payload" rate of the guide), so the loop keeps NS - 1 k-blocks (NS (NT + 2) P KiB per wave) requested ahead of
// the MFMAs in a ring of register stages with static indices.  Products: piece pairs (a, b) with a + b < P, smallest first.
using wu32x4 = __attribute__((ext_vector_type(4))) unsigned;
// Exchange vectors are written by OTHER workgroups of the same launch (write-through stores): every load of them is an sc1
// buffer load (served by L2 / the fabric, never by this CU's L1), which is what lets the grid barrier go without an acquire
// fence (MI355X_MICROARCH.md, "Valid forms": sc1 payload stores drained before the arrival add, sc1 poll, sc1 loads).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t wide_rsrc(const void* base) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ float wide_load_f(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

template <int NT, int P, int NS, int RT = 2>
__device__ __forceinline__ void wide_mfma_stream(wf32x4 (&acc)[NT][RT], const uint4* const (&wt)[NT], const uint4* __restrict__ xbase, int KS,
                                                 int ks0, int ks1, int lane) {
  if (ks0 >= ks1) return;
  const int xlane = (lane & 15) * 4 + (lane >> 4);
  const __amdgpu_buffer_rsrc_t xb = wide_rsrc(xbase);
  uint4 a[NS][NT][P];
  wu32x4 b[NS][RT][P];
  auto load = [&](int st, int ks) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int p = 0; p < P; ++p) a[st][nt][p] = wt[nt][((size_t)ks * P + p) * 64 + lane];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int p = 0; p < P; ++p)
        b[st][rt][p] = __builtin_amdgcn_raw_buffer_load_b128(xb, (((p * KS + ks) * (16 * RT) + 16 * rt) * 4 + xlane) * 16, 0, 16);   // aux 16 = sc1
  };
  auto compute = [&](int st) {
#pragma unroll
    for (int ord = P - 1; ord >= 0; --ord)
#pragma unroll
      for (int pa = 0; pa <= ord; ++pa)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int rt = 0; rt < RT; ++rt)
            acc[nt][rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(wbf16x8, a[st][nt][pa]),
                                                                  __builtin_bit_cast(wbf16x8, b[st][rt][ord - pa]), acc[nt][rt], 0, 0, 0);
  };
#pragma unroll
  for (int s = 0; s < NS - 1; ++s)
    if (ks0 + s < ks1) load(s, ks0 + s);
  for (int ks = ks0; ks < ks1; ks += NS) {
#pragma unroll
    for (int s = 0; s < NS; ++s) {   // static stage indices: block ks + s sits in stage s
      if (ks + s < ks1) {
        if (ks + s + NS - 1 < ks1) load((s + NS - 1) % NS, ks + s + NS - 1);
        compute(s);
      }
    }
  }
}

// Cross-wave reduction through LDS: red[(wave NTOT + tile) RT + rt][lane] (float4 each).
template <int NTOT, int RT = 2>
__device__ __forceinline__ void wide_red_store(wf32x4* red, int wave, int tile, int lane, const wf32x4 (&acc)[RT]) {
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) red[((wave * NTOT + tile) * RT + rt) * kWave + lane] = acc[rt];
}
template <int NTOT, int RT = 2>
__device__ __forceinline__ wf32x4 wide_red_sum(const wf32x4* red, int tile, int rt, int slot) {
  wf32x4 s = red[((0 * NTOT + tile) * RT + rt) * kWave + slot];
#pragma unroll
  for (int w = 1; w < kWW; ++w) s += red[((w * NTOT + tile) * RT + rt) * kWave + slot];
  return s;
}

// Grid-wide barrier over `nblk` resident workgroups, two levels: the workgroups with equal blockIdx % 8 (one XCD under the
// observed round-robin placement: speed only, nothing depends on it) arrive at their group's counter; the last of a group
// arrives at the top counter; the last group releases everybody by storing the epoch into each group's generation word, which is
// what a group's workgroups poll (32 pollers per word instead of 256 on one: the flat form cost 8 us after a short phase).
// Every storing wave drains its write-through stores before its workgroup arrives; everything another workgroup wrote is read
// with sc1 loads afterwards, so no acquire fence is needed (`acquire` adds it: an A/B switch).  Words live on 128-byte lines of
// their own in the control block: [+128] top, [+256 + 128 g] arrivals of group g, [+1280 + 128 g] generation of group g.
// Bounded: a poll that gives up sets the sticky status word and the LDS abort flag; the caller leaves the kernel.
constexpr int kWideGroups = 8;
constexpr size_t kWideCtl = 2560;   // control block bytes: [0] sticky status word (never cleared by a launch), then the barrier words
struct WideBarrier {
  unsigned* ctl;      // control block + 128
  int* status;
  int* abort_flag;    // LDS
  unsigned epoch;
  unsigned group, group_size, groups;
  bool acquire;
  __device__ __forceinline__ void init(void* control, int* status_, int* abort_, int nblk, int blk, bool acquire_) {
    ctl = reinterpret_cast<unsigned*>(static_cast<char*>(control) + 128);
    status = status_;
    abort_flag = abort_;
    epoch = 0;
    group = blk % kWideGroups;
    group_size = (nblk - (int)group + kWideGroups - 1) / kWideGroups;
    groups = nblk < kWideGroups ? nblk : kWideGroups;
    acquire = acquire_;
  }
  __device__ __forceinline__ bool sync(int code) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    ++epoch;
    if (threadIdx.x == 0) {
      unsigned* const top = ctl;
      unsigned* const arrive = ctl + 32 * (1 + group);
      unsigned* const gen = ctl + 32 * (1 + kWideGroups + group);
      const unsigned a = __hip_atomic_fetch_add(arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (a + 1 == group_size * epoch) {
        const unsigned t = __hip_atomic_fetch_add(top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t + 1 == groups * epoch) {
          for (unsigned g = 0; g < groups; ++g)
            __hip_atomic_store(ctl + 32 * (1 + kWideGroups + g), epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      bool ok = true;
      for (unsigned spins = 0; __hip_atomic_load(gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch; ++spins) {
        if (spins > kWideSpinLimit) { ok = false; break; }
        __builtin_amdgcn_s_sleep(1);
      }
      if (!ok) {
        *abort_flag = 1;
        atomicExch(status, code);
      }
      if (acquire) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    }
    __syncthreads();
    return *abort_flag == 0;
  }
};

}  // namespace mtrssm
