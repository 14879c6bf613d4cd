// Large dense GEMMs on the bf16 MFMA with fp32-grade operands (included by gemm.hip; GemmArgs, act4-style helpers above).
//
// gemm_f32_kernel reaches 63-98 TFLOP/s on the big Linear layers (157 is the fp32 MFMA's peak): the encoder head / decoder stem
// of the conv stacks (3200 x 4096 x 256) and, at BASELINE configs[4] dims, every projection and weight gradient around the scan
// (3200 x 1024 x 1024 ... 3072 x 1024 x 3200).  Here every fp32 operand value becomes P bf16 pieces while its tile is staged
// (P = 3: exact to 2^-24, six v_mfma_f32_32x32x16_bf16 products per k-block -- fp32-grade, the default; P = 2: 16 significant
// bits, three products -- the conv kernels' arithmetic), fp32 accumulation: 6/16 resp. 3/16 of the fp32 MFMA's cycles per
// multiply-add.  What the earlier split kernel (gemm_split_kernel) lacked:
//   * 128 x TN tiles (TN = 128 or 64), four waves of 64 x TN/2: twice the MFMA work per staged operand value;
//   * full tiles only (M % 128 = N % TN = R % 32 = 0, 16-byte aligned rows): per-thread pointers advance by one k-step per
//     request, nothing is clamped or masked in the loop; other shapes stay on the older kernels;
//   * eight waves in two roles (below), two LDS buffers of [piece][row][32 k] bf16 images (80-byte pitch: conflict-free 16-byte
//     fragment reads), 2 x 61 KB at P = 3, one workgroup per CU; the two steps after the one being staged are in flight in
//     registers.
// Same contract and epilogues as gemm_f32_kernel (bias, act' of the data gradient, accumulate, column sums, split reduction by
// fp32 atomics with the last-arriver pass).
#pragma once

namespace mtrssm {

constexpr int kGM = 128;      // tile rows (M)
constexpr int kGK = 32;       // reduction extent per step
constexpr int kGPitch = 80;   // bytes per image row: 32 bf16 + 16 (an odd number of 16-byte slots)

template <int P>
__device__ __forceinline__ void tile_store_item(const float (&x)[8], unsigned char* img, int img_bytes, int act, int row, int k8) {
  float v[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) v[u] = x[u];
  if (act == MTRSSM_ACT_ELU) {
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = v[u] > 0.f ? v[u] : __expf(v[u]) - 1.f;
  } else if (act == MTRSSM_ACT_RELU) {
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = fmaxf(v[u], 0.f);
  } else if (act == MTRSSM_ACT_TANH) {
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = 1.f - 2.f / (__expf(2.f * v[u]) + 1.f);
  }
  unsigned char* d = img + row * kGPitch + k8 * 16;
#pragma unroll
  for (int p = 0; p < P; ++p) {
    u16x8 piece;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const __bf16 h = (__bf16)v[u];
      piece[u] = __builtin_bit_cast(unsigned short, h);
      v[u] -= (float)h;
    }
    *reinterpret_cast<u16x8*>(d + p * img_bytes) = piece;
  }
}

// ROWS rows x 32 k of one operand: ITEMS = ROWS / 64 items of 8 consecutive k per thread
template <bool RMAJOR, int ROWS>
struct TileOperand {
  static constexpr int ITEMS = ROWS / 64;
  const float* ptr[ITEMS];   // this thread's items at the NEXT step to request
  size_t bump;               // floats per k-step
  int tid;                   // 0..255 within the staging role
  __device__ __forceinline__ void init(const float* P, int ld, int row0, int k0, int tid_) {
    tid = tid_;
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
      const int i = tid + 256 * j;
      if (RMAJOR) ptr[j] = P + (size_t)(k0 + 8 * (i / ROWS)) * ld + row0 + (i % ROWS);      // memory [k][row]
      else ptr[j] = P + (size_t)(row0 + (i >> 2)) * ld + k0 + 8 * (i & 3);                   // memory [row][k]
    }
    bump = RMAJOR ? (size_t)kGK * ld : kGK;
  }
  // request this thread's items of the current step, then move on by one step -- unless `last` (the requests stay unconditional
  // so that the compiler's vmcnt bookkeeping is static: a request under a branch makes every later wait a vmcnt(0), i.e. the
  // staging waves then sleep through the full latency of the requests they have just issued, every step)
  __device__ __forceinline__ void load(float (&v)[ITEMS][8], int ld, bool last) {
    const size_t step = last ? 0 : bump;
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
      if (RMAJOR) {
#pragma unroll
        for (int u = 0; u < 8; ++u) v[j][u] = ptr[j][(size_t)u * ld];
      } else {
        const float4 q0 = *reinterpret_cast<const float4*>(ptr[j]);
        const float4 q1 = *reinterpret_cast<const float4*>(ptr[j] + 4);
        v[j][0] = q0.x; v[j][1] = q0.y; v[j][2] = q0.z; v[j][3] = q0.w;
        v[j][4] = q1.x; v[j][5] = q1.y; v[j][6] = q1.z; v[j][7] = q1.w;
      }
      ptr[j] += step;
    }
  }
  template <int P>
  __device__ __forceinline__ void store(const float (&v)[ITEMS][8], unsigned char* img, int act) {
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
      const int i = tid + 256 * j;
      tile_store_item<P>(v[j], img, ROWS * kGPitch, act, RMAJOR ? i % ROWS : i >> 2, RMAJOR ? i / ROWS : i & 3);
    }
  }
};

// Eight waves: waves 0-3 multiply (2 x 2, each 64 rows x TN / 2 columns of the tile), waves 4-7 stage -- request the step after
// next, convert the next one to bf16 pieces and write it into the OTHER of two LDS buffers -- so that a SIMD's matrix pipe (its
// multiplying wave) and its vector pipe (its staging wave) run side by side; one barrier per k-step.
// Measured on 3200 x 1024 x 4096 (P = 3 / P = 2, us): whole kernel 235 / 156; without the MFMAs 167 / 138; without the
// conversions 165 / 121; requests + barriers + fragment reads alone 64 / 58 (13.8 TB/s out of L2); MFMAs + fragment reads
// without staging 146 / 89 (the MFMAs alone would be 94 / 47).  So a k-step of 32 carries ~800 cycles of barrier + LDS latency
// on either role beside 768-1536 of MFMA; no single pipe is saturated (LDS array 1150 of 3700 cycles per step, a third of
// them bank conflicts of the 16-byte stores).  Both roles in the same four waves, one buffer, two workgroups per CU gave
// 125-165 TFLOP/s at P = 3 and 110-200 at P = 2; this split: 90-135 and 128-200.  Next: 256 x 128 tiles on twelve waves.
template <bool AR, bool BR, int P, int TN>
__global__ __launch_bounds__(512) void gemm_tile_kernel(const GemmArgs g) {
  constexpr int NC = TN / 64;                    // 32-column accumulators per multiplying wave
  constexpr int kImgA = kGM * kGPitch, kImgB = TN * kGPitch;   // bytes of one piece
  constexpr int kBuf = P * (kImgA + kImgB);
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];   // 2 kBuf bytes
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool stager = wave >= 4;
  const int wr = (wave >> 1) & 1, wc = wave & 1, il = lane & 31, kl = lane >> 5;
  // XCD-aware order of the tiles (1-D grid; workgroups b, b + 8, ... run on one XCD under the observed round-robin placement:
  // speed only): XCD x takes a CONTIGUOUS run of the (row tile, column tile, slice) sequence, column tiles fastest -- its
  // workgroups share the A' row panels of a few row tiles in that XCD's L2 instead of every XCD streaming all of A' and B'
  // through the fabric (measured before: 840 MB of operand traffic for 80 MB of operands, 4.2 TB/s = the whole kernel time).
  const int tiles_x = g.N / TN, tiles_y = g.M / kGM, total = tiles_x * tiles_y * g.splits;
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3, q8 = total >> 3, r8 = total & 7;
  const int seq = xcd * q8 + (xcd < r8 ? xcd : r8) + idx;        // bijective for any total
  const int bz = seq % g.splits, bx = (seq / g.splits) % tiles_x, by = seq / (g.splits * tiles_x);
  const int i0 = by * kGM, j0 = bx * TN;
  const int steps = g.R / kGK;
  const int per = (steps + g.splits - 1) / g.splits;
  const int s_lo = bz * per, s_hi = min(steps, s_lo + per);
  if (s_lo >= s_hi) return;
  const bool want_colsum = g.colsum && bx == 0;
  float csum = 0.f;

  f32x16 acc[2][NC];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[t][c][i] = 0.f;

  const int fa_off = (wr * 64 + il) * kGPitch + kl * 16;
  const int fb_off = P * kImgA + (wc * (TN / 2) + il) * kGPitch + kl * 16;
  auto compute = [&](const unsigned char* buf) {
    const unsigned char* fa = buf + fa_off;
    const unsigned char* fb = buf + fb_off;
    bf16x8 a[2][2][P], b[2][NC][P];   // both k-blocks' fragments requested up front: the second lands under the first's MFMAs
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int p = 0; p < P; ++p) {
#pragma unroll
        for (int t = 0; t < 2; ++t) a[kb][t][p] = *reinterpret_cast<const bf16x8*>(fa + p * kImgA + t * 32 * kGPitch + kb * 32);
#pragma unroll
        for (int c = 0; c < NC; ++c) b[kb][c][p] = *reinterpret_cast<const bf16x8*>(fb + p * kImgB + c * 32 * kGPitch + kb * 32);
      }
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int ord = P - 1; ord >= 0; --ord)   // smallest terms first
#pragma unroll
        for (int pa = 0; pa <= ord; ++pa)
#pragma unroll
          for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int c = 0; c < NC; ++c) acc[t][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[kb][t][pa], b[kb][c][ord - pa], acc[t][c], 0, 0, 0);
    }
    if (want_colsum && tid < kGM) {   // bias gradient: the staged A' row (the pieces sum to the fp32 value)
      const unsigned char* row = buf + tid * kGPitch;
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int p = 0; p < P; ++p) {
          const u16x8 h = *reinterpret_cast<const u16x8*>(row + q * 16 + p * kImgA);
#pragma unroll
          for (int u = 0; u < 8; ++u) csum += __uint_as_float((unsigned)h[u] << 16);
        }
    }
  };

  if (stager) {
    TileOperand<AR, kGM> opa;
    TileOperand<BR, TN> opb;
    opa.init(g.A, g.lda, i0, s_lo * kGK, tid - 256);
    opb.init(g.B, g.ldb, j0, s_lo * kGK, tid - 256);
    float ra[2][TileOperand<AR, kGM>::ITEMS][8], rb[2][TileOperand<BR, TN>::ITEMS][8];   // two register stages of requests
    // step `nxt` is the next one to request; past the last step the requests repeat it (never stored where it would be read)
    const int n = s_hi - s_lo;
    int nxt = 0;
    auto request = [&](int set) {
      const bool last = nxt + 1 >= n;
      opa.load(ra[set], g.lda, last);
      opb.load(rb[set], g.ldb, last);
      ++nxt;
    };
    auto stage = [&](int set, unsigned char* buf) {
      opa.template store<P>(ra[set], buf, g.act_a);
      opb.template store<P>(rb[set], buf + P * kImgA, g.act_b);
    };
    __builtin_amdgcn_s_setprio(2);   // the conversions are the longer chain of a k-step (P = 3: -7 %)
    request(0);
    request(1);
    stage(0, lds);
    request(0);
    lds_barrier();
    for (int i = 0; i < n; i += 2) {
      stage(1, lds + kBuf);   // step i + 1 into buffer 1 while step i is multiplied out of buffer 0
      request(1);
      lds_barrier();
      stage(0, lds);          // step i + 2 into buffer 0 while step i + 1 is multiplied out of buffer 1
      request(0);
      lds_barrier();
    }
    return;   // the epilogue belongs to the waves that hold the accumulators (a barrier counts live waves only)
  }
  lds_barrier();
  for (int i = 0; i < s_hi - s_lo; i += 2) {
    compute(lds);
    lds_barrier();
    if (i + 1 < s_hi - s_lo) compute(lds + kBuf);
    lds_barrier();
  }

  // epilogue: C/D layout of the 32x32 MFMA: column = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
  const bool atomic = g.splits > 1;
  const bool finalize = atomic && g.tickets != nullptr;
  const bool colsum_now = want_colsum && tid < kGM;
  if (atomic) {
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int j = j0 + wc * (TN / 2) + c * 32 + il;
      const float bias0 = (g.bias && !finalize && bz == 0) ? g.bias[j] : 0.f;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int ibase = i0 + wr * 64 + t * 32 + 4 * kl;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) atomicAdd(g.C + (size_t)(ibase + (reg & 3) + 8 * (reg >> 2)) * g.ldc + j, acc[t][c][reg] + bias0);
      }
    }
    if (!finalize) {
      if (colsum_now) atomicAdd(g.colsum + i0 + tid, csum);
      return;
    }
    // (no static __shared__ object beside the dynamic region: it would shift the region's base off its 16-byte alignment;
    //  the operand images are dead by now)
    volatile int* const last_flag = reinterpret_cast<volatile int*>(lds);
    __threadfence();
    __syncthreads();
    if (tid == 0) *last_flag = atomicAdd(g.tickets + by * tiles_x + bx, 1) == g.splits - 1;
    __syncthreads();
    if (!*last_flag) {
      if (colsum_now) atomicAdd(g.colsum + i0 + tid, csum);
      return;
    }
    __threadfence();
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int j = j0 + wc * (TN / 2) + c * 32 + il;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int ibase = i0 + wr * 64 + t * 32 + 4 * kl;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg)
          acc[t][c][reg] = __hip_atomic_load(g.C + (size_t)(ibase + (reg & 3) + 8 * (reg >> 2)) * g.ldc + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int j = j0 + wc * (TN / 2) + c * 32 + il;
    const float bias = g.bias ? g.bias[j] : 0.f;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int ibase = i0 + wr * 64 + t * 32 + 4 * kl;
      float zv[16], cv[16];
      if (g.zgrad) {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) zv[reg] = g.zgrad[(size_t)(ibase + (reg & 3) + 8 * (reg >> 2)) * g.ldz + j];
      }
      if (g.accumulate && !atomic) {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) cv[reg] = g.C[(size_t)(ibase + (reg & 3) + 8 * (reg >> 2)) * g.ldc + j];
      }
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        float v = acc[t][c][reg] + bias;
        if (g.act_out) v = act_fwd(v, g.act_out);
        if (g.zgrad) v *= act_grad_from_in(zv[reg], g.act_z);
        if (g.accumulate && !atomic) v += cv[reg];
        g.C[(size_t)(ibase + (reg & 3) + 8 * (reg >> 2)) * g.ldc + j] = v;
      }
    }
  }
  if (colsum_now) atomicAdd(g.colsum + i0 + tid, csum);
}

}  // namespace mtrssm
