// kh_gemm.hip — CuMatrixBase::AddMatMat (cudamatrix/cu-matrix.cc:947-982, which
// calls cuBLAS sgemm through cublas-wrappers.h:28-33) as a hand-written FP32
// MFMA GEMM for gfx950.
//
// Numerics: v_mfma_f32_32x32x2_f32 is exact f32 — each output element is one
// k-ordered fmaf chain (guide §3 "FP32-input MFMA").  K is never split across
// accumulators or workgroups, so the result is bit-identical to
//     acc = 0; for k in 0..K-1: acc = fmaf(a[i,k], b[k,j], acc)
// which is what the CPU oracle computes; the epilogue is
//     c = beta == 0 ? alpha*acc : fl(beta*c) + fl(alpha*acc).
//
// Tiling: 128x128 output tile per 256-thread workgroup (4 waves as 2x2, each
// wave 64x64 = 2x2 MFMA 32x32 tiles -> 64 accumulator VGPRs), BK = 16,
// double-buffered LDS filled from registers (global loads for tile t+1 are
// issued before the MFMAs of tile t).  LDS image is [k][m] (+4 pad) so a wave's
// operand fetch (lanes 0-31: k, lanes 32-63: k+1, consecutive m) is a
// conflict-free ds_read_b32.  Workgroup ids are remapped so each XCD (own L2)
// gets a contiguous run of tiles, ordered in groups of kGroupM row panels x all
// column panels (column-major inside a group: the tiles resident together share
// few A and B panels).
//
// What tools/gemm_lab.hip measured on the forward pass's shapes:
// the 64 stores of an interior tile are issued back to back (per-element bounds
// checks made hipcc put an s_waitcnt vmcnt(0) — which also waits for the previous
// STORE — before every store: 40 k of a wave's 225 k cycles); interior tiles load
// whole k-slabs without bounds checks through a scalar base + one 32-bit lane
// offset; the workgroups that start together on a CU get different s_setprio
// levels (the matrix pipe is shared by priority, then age).
#include "kh_common.h"

using namespace kh;

namespace {

constexpr int BM = 128, BN = 128, BK = 16, LDT = BM + 4;
constexpr int kThreads = 256;

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct GemmArgs {
  const float *A;  // element (i,k) at A[i*a_si + k*a_sk]
  const float *B;  // element (j,k) at B[j*b_sj + k*b_sk]
  float *C;
  const float *bias;  // optional [N]
  int M, N, K;
  long a_si, a_sk, b_sj, b_sk;
  int c_stride;
  float alpha, beta;
  int tiles_m, tiles_n;
  int lane_offsets_ok;  // 128 rows of A and of B span < 2^31 bytes: 32-bit lane offsets
};

constexpr int kGroupM = 8;

__device__ __forceinline__ int XcdRemap(int bid, int nwg) {
  const int cpx = nwg >> 3, rem = nwg & 7;
  const int xcd = bid & 7, local = bid >> 3;
  return xcd < rem ? xcd * (cpx + 1) + local
                   : rem * (cpx + 1) + (xcd - rem) * cpx + local;
}

// Loads 4 consecutive-k elements of one row of an operand tile.
template <bool VEC>
__device__ __forceinline__ float4 LoadRow4(const float *base, long s_row, long s_k,
                                           int row, int k, int rows, int K) {
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (row < rows) {
    if (VEC) {
      const float *p = base + row * s_row + k;
      if (k + 3 < K) {
        v = *reinterpret_cast<const float4 *>(p);
      } else {
        if (k < K) v.x = p[0];
        if (k + 1 < K) v.y = p[1];
        if (k + 2 < K) v.z = p[2];
      }
    } else {
      const float *p = base + row * s_row + k * s_k;
      if (k < K) v.x = p[0];
      if (k + 1 < K) v.y = p[s_k];
      if (k + 2 < K) v.z = p[2 * s_k];
      if (k + 3 < K) v.w = p[3 * s_k];
    }
  }
  return v;
}

template <bool VEC_A, bool VEC_B>
__global__ void __launch_bounds__(kThreads, 4) GemmKernel(GemmArgs g) {  // 4 waves per SIMD: <= 128 registers
  __shared__ float As[2][BK][LDT];
  __shared__ float Bs[2][BK][LDT];

  switch ((blockIdx.x >> 8) & 3) {  // 32 CUs per XCD: workgroups b, b + 256, ... start on one CU
    case 1: __builtin_amdgcn_s_setprio(1); break;
    case 2: __builtin_amdgcn_s_setprio(2); break;
    case 3: __builtin_amdgcn_s_setprio(3); break;
    default: break;
  }
  const int nwg = g.tiles_m * g.tiles_n;
  const int tile = XcdRemap(blockIdx.x, nwg);
  const int per_group = kGroupM * g.tiles_n;
  const int gid = tile / per_group, in_group = tile - gid * per_group;
  const int group_rows = min(g.tiles_m - gid * kGroupM, kGroupM);
  const int tn = in_group / group_rows, tm = gid * kGroupM + (in_group - tn * group_rows);
  const int m0 = tm * BM, n0 = tn * BN;

  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int lrow = t >> 2;        // 0..63
  const int lk = (t & 3) << 2;    // 0,4,8,12

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

  const int rowsA = g.M - m0, rowsB = g.N - n0;
  const float *Ab = g.A + static_cast<long>(m0) * g.a_si;
  const float *Bb = g.B + static_cast<long>(n0) * g.b_sj;

  float4 ra[2], rb[2];
  const bool interior = VEC_A && VEC_B && g.lane_offsets_ok && rowsA >= BM && rowsB >= BN;
  unsigned offa[2], offb[2];  // byte offsets of this lane's two rows of A and of B
#pragma unroll
  for (int i = 0; i < 2; i++) {
    offa[i] = (static_cast<unsigned>(lrow + 64 * i) * static_cast<unsigned>(g.a_si) + lk) * 4u;
    offb[i] = (static_cast<unsigned>(lrow + 64 * i) * static_cast<unsigned>(g.b_sj) + lk) * 4u;
  }
  auto load_tile = [&](int k0) {
    if (interior && k0 + BK <= g.K) {
      const char *pa = reinterpret_cast<const char *>(Ab + k0);
      const char *pb = reinterpret_cast<const char *>(Bb + k0);
#pragma unroll
      for (int i = 0; i < 2; i++) {
        ra[i] = *reinterpret_cast<const float4 *>(pa + offa[i]);
        rb[i] = *reinterpret_cast<const float4 *>(pb + offb[i]);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 2; i++) {
        ra[i] = LoadRow4<VEC_A>(Ab, g.a_si, g.a_sk, lrow + 64 * i, k0 + lk, rowsA, g.K);
        rb[i] = LoadRow4<VEC_B>(Bb, g.b_sj, g.b_sk, lrow + 64 * i, k0 + lk, rowsB, g.K);
      }
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const int m = lrow + 64 * i;
      As[buf][lk + 0][m] = ra[i].x;
      As[buf][lk + 1][m] = ra[i].y;
      As[buf][lk + 2][m] = ra[i].z;
      As[buf][lk + 3][m] = ra[i].w;
      Bs[buf][lk + 0][m] = rb[i].x;
      Bs[buf][lk + 1][m] = rb[i].y;
      Bs[buf][lk + 2][m] = rb[i].z;
      Bs[buf][lk + 3][m] = rb[i].w;
    }
  };

  const int nk = (g.K + BK - 1) / BK;
  load_tile(0);
  store_tile(0);
  __syncthreads();

  const int kk = lane >> 5, l31 = lane & 31;
  for (int kt = 0; kt < nk; kt++) {
    const int buf = kt & 1;
    if (kt + 1 < nk) load_tile((kt + 1) * BK);
#pragma unroll
    for (int s = 0; s < BK / 2; s++) {
      const int k = 2 * s + kk;
      float a0 = As[buf][k][wm * 64 + l31];
      float a1 = As[buf][k][wm * 64 + 32 + l31];
      float b0 = Bs[buf][k][wn * 64 + l31];
      float b1 = Bs[buf][k][wn * 64 + 32 + l31];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
    if (kt + 1 < nk) {
      store_tile(buf ^ 1);  // the other buffer was last read in iteration kt-1
      __syncthreads();
    }
  }

  // epilogue: lane holds column (lane&31), rows (r&3) + 8*(r>>2) + 4*(lane>>5)
  if (m0 + BM <= g.M && n0 + BN <= g.N && (g.bias || g.beta == 0.f)) {
    // interior tile, nothing read from C: the 64 stores follow one another with no wait between them
    // (wave-uniform row base + one per-lane offset)
    const int wu = __builtin_amdgcn_readfirstlane(t >> 6);
    const int col = n0 + (wu & 1) * 64 + l31;
    float bv0 = 0.f, bv1 = 0.f;
    if (g.bias) {
      bv0 = g.bias[col];
      bv1 = g.bias[col + 32];
    }
    float *cb = g.C + static_cast<size_t>(m0 + (wu >> 1) * 64) * g.c_stride + n0 + (wu & 1) * 64;
    const unsigned voff = static_cast<unsigned>(4 * kk) * g.c_stride + l31;
    if (g.bias) {
#pragma unroll
      for (int i = 0; i < 2; i++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
          float *cp = cb + static_cast<size_t>(i * 32 + (r & 3) + 8 * (r >> 2)) * g.c_stride;
#ifndef KH_GEMM_NO_NT
          // (non-temporal: a product this wide - the output layer's logits - is read once, by the next kernel, and should
          // not displace the weight panels in L2: forward pass 400 -> 395 ms)
          __builtin_nontemporal_store(acc[i][0][r] + bv0, &cp[voff]);
          __builtin_nontemporal_store(acc[i][1][r] + bv1, &cp[voff + 32]);
#else
          cp[voff] = acc[i][0][r] + bv0;
          cp[voff + 32] = acc[i][1][r] + bv1;
#endif
        }
    } else {
#pragma unroll
      for (int i = 0; i < 2; i++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
          float *cp = cb + static_cast<size_t>(i * 32 + (r & 3) + 8 * (r >> 2)) * g.c_stride;
          cp[voff] = g.alpha * acc[i][0][r];
          cp[voff + 32] = g.alpha * acc[i][1][r];
        }
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < 2; i++) {
#pragma unroll
    for (int j = 0; j < 2; j++) {
      const int col = n0 + wn * 64 + j * 32 + l31;
      if (col >= g.N) continue;
      const float bv = g.bias ? g.bias[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk;
        if (row >= g.M) continue;
        float *cp = g.C + static_cast<size_t>(row) * g.c_stride + col;
        const float a = acc[i][j][r];
        float out;
        if (g.bias) {
          out = a + bv;
        } else {
          const float prod = g.alpha * a;
          out = (g.beta == 0.f) ? prod : (g.beta * *cp + prod);
        }
        *cp = out;
      }
    }
  }
}


// ---- AffineComponent + PnormComponent (p = 2) in one kernel -------------------------------------------
// nnet-component.cc:1219-1224 (bias rows, AddMatMat) followed by :386-391 (GroupPnorm): the 3500-wide
// activations of a hidden layer are never written (0.19 ms of GroupPnorm2RowKernel per 60 k frames, and
// 64 KB of stores per tile, gone: the fused call takes the time of the plain product).  128 x 160 tile
// (160 = 16 groups of 10), 4 waves of 32 rows x 160 columns (1 x 5 MFMA tiles: one k-ordered fmaf chain
// per element like GemmKernel, so the sums below see the same activations), then the tile goes through
// the operand buffers 8 rows per wave at a time and one lane forms a (row, group) sum in
// GroupPnorm2RowKernel's order: s = 0; s += x_j * x_j (j ascending); sqrtf(s).
constexpr int PBM = 128, PBN = 160, PLA = PBM + 4, PLB = PBN + 4, PSTR = PBN + 4;

struct PnormArgs {
  GemmArgs g;  // C unused
  float *Y;
  int y_stride, group;
};

template <bool VEC>
__global__ void __launch_bounds__(kThreads, 4) GemmPnormKernel(PnormArgs pa) {
  const GemmArgs &g = pa.g;
  __shared__ float lds[2 * BK * (PLA + PLB)];
  auto As = [&](int buf, int k, int m) -> float & { return lds[(buf * BK + k) * PLA + m]; };
  auto Bs = [&](int buf, int k, int n) -> float & { return lds[2 * BK * PLA + (buf * BK + k) * PLB + n]; };
  const int nwg = g.tiles_m * g.tiles_n;
  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int lrow = t >> 2;
  const int lk = (t & 3) << 2;
  const int kk = lane >> 5, l31 = lane & 31;
  const int nk = (g.K + BK - 1) / BK;
  const int tile = XcdRemap(blockIdx.x, nwg);
  const int per_group = kGroupM * g.tiles_n;
  const int gid = tile / per_group, in_group = tile - gid * per_group;
  const int group_rows = min(g.tiles_m - gid * kGroupM, kGroupM);
  const int tn = in_group / group_rows, tm = gid * kGroupM + (in_group - tn * group_rows);
  const int m0 = tm * PBM, n0 = tn * PBN;
  const int rowsA = g.M - m0, rowsB = g.N - n0;
  const float *Ab = g.A + static_cast<long>(m0) * g.a_si;
  const float *Bb = g.B + static_cast<long>(n0) * g.b_sj;
  const bool interior = VEC && g.lane_offsets_ok && rowsA >= PBM && rowsB >= PBN;
  const bool third = wave < 2;  // rows 128..159 of the B tile: lrow < 32

  float4 ra[2], rb[3];
  unsigned offa[2], offb[3];
#pragma unroll
  for (int i = 0; i < 2; i++) offa[i] = (static_cast<unsigned>(lrow + 64 * i) * static_cast<unsigned>(g.a_si) + lk) * 4u;
#pragma unroll
  for (int i = 0; i < 3; i++) offb[i] = (static_cast<unsigned>(lrow + 64 * i) * static_cast<unsigned>(g.b_sj) + lk) * 4u;
  rb[2] = make_float4(0.f, 0.f, 0.f, 0.f);
  auto load_tile = [&](int k0) {
    if (interior && k0 + BK <= g.K) {
      const char *pa_ = reinterpret_cast<const char *>(Ab + k0);
      const char *pb_ = reinterpret_cast<const char *>(Bb + k0);
#pragma unroll
      for (int i = 0; i < 2; i++) ra[i] = *reinterpret_cast<const float4 *>(pa_ + offa[i]);
#pragma unroll
      for (int i = 0; i < 2; i++) rb[i] = *reinterpret_cast<const float4 *>(pb_ + offb[i]);
      if (third) rb[2] = *reinterpret_cast<const float4 *>(pb_ + offb[2]);
    } else {
#pragma unroll
      for (int i = 0; i < 2; i++) ra[i] = LoadRow4<VEC>(Ab, g.a_si, g.a_sk, lrow + 64 * i, k0 + lk, rowsA, g.K);
#pragma unroll
      for (int i = 0; i < 2; i++) rb[i] = LoadRow4<VEC>(Bb, g.b_sj, g.b_sk, lrow + 64 * i, k0 + lk, rowsB, g.K);
      if (third) rb[2] = LoadRow4<VEC>(Bb, g.b_sj, g.b_sk, lrow + 128, k0 + lk, rowsB, g.K);
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const int m = lrow + 64 * i;
      As(buf, lk + 0, m) = ra[i].x;
      As(buf, lk + 1, m) = ra[i].y;
      As(buf, lk + 2, m) = ra[i].z;
      As(buf, lk + 3, m) = ra[i].w;
    }
#pragma unroll
    for (int i = 0; i < 3; i++) {
      if (i == 2 && !third) break;
      const int m = lrow + 64 * i;
      Bs(buf, lk + 0, m) = rb[i].x;
      Bs(buf, lk + 1, m) = rb[i].y;
      Bs(buf, lk + 2, m) = rb[i].z;
      Bs(buf, lk + 3, m) = rb[i].w;
    }
  };

  f32x16 acc[5];
#pragma unroll
  for (int j = 0; j < 5; j++)
#pragma unroll
    for (int r = 0; r < 16; r++) acc[j][r] = 0.f;
  load_tile(0);
  store_tile(0);
  __syncthreads();
  for (int kt = 0; kt < nk; kt++) {
    const int buf = kt & 1;
    if (kt + 1 < nk) load_tile((kt + 1) * BK);
#pragma unroll
    for (int s = 0; s < BK / 2; s++) {
      const int k = 2 * s + kk;
      const float a = As(buf, k, wave * 32 + l31);
      float b[5];
#pragma unroll
      for (int j = 0; j < 5; j++) b[j] = Bs(buf, k, 32 * j + l31);
#pragma unroll
      for (int j = 0; j < 5; j++) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[j], acc[j], 0, 0, 0);
    }
    if (kt + 1 < nk) {
      store_tile(buf ^ 1);
      __syncthreads();
    }
  }
  __syncthreads();  // the operand buffers become the staging area
  float *S = lds + wave * (8 * PSTR);
  float bv[5];
#pragma unroll
  for (int j = 0; j < 5; j++) bv[j] = n0 + 32 * j + l31 < g.N ? g.bias[n0 + 32 * j + l31] : 0.f;
  const int groups_tile = PBN / pa.group;
  const int items = 8 * groups_tile;  // (row, group) pairs of one pass of a wave
  const int groups_total = g.N / pa.group;
#pragma unroll
  for (int q = 0; q < 4; q++) {  // rows 8q .. 8q+7 of the wave's 32: accumulator registers 4q .. 4q+3
#pragma unroll
    for (int j = 0; j < 5; j++)
#pragma unroll
      for (int r = 0; r < 4; r++) S[(r + 4 * kk) * PSTR + 32 * j + l31] = acc[j][4 * q + r] + bv[j];
    __syncthreads();
    for (int it = lane; it < items; it += 64) {
      const int row = it / groups_tile, grp = it - row * groups_tile;
      const float *x = S + row * PSTR + grp * pa.group;
      float sum = 0.f;
      for (int j = 0; j < pa.group; j++) sum += x[j] * x[j];
      const int gr = m0 + wave * 32 + 8 * q + row, gc = n0 / pa.group + grp;
      if (gr < g.M && gc < groups_total) pa.Y[static_cast<size_t>(gr) * pa.y_stride + gc] = sqrtf(sum);
    }
    __syncthreads();
  }
}

int LaunchGemm(GemmArgs g) {
  if (g.M == 0 || g.N == 0) return KH_OK;
  g.tiles_m = DivUp(g.M, BM);
  g.tiles_n = DivUp(g.N, BN);
  const int nwg = g.tiles_m * g.tiles_n;
  g.lane_offsets_ok = (g.a_si >= 0 && g.b_sj >= 0 && g.a_si < (1L << 22) && g.b_sj < (1L << 22)) ? 1 : 0;
  const bool va = g.a_sk == 1 && (g.a_si % 4 == 0) &&
                  (reinterpret_cast<uintptr_t>(g.A) % 16 == 0);
  const bool vb = g.b_sk == 1 && (g.b_sj % 4 == 0) &&
                  (reinterpret_cast<uintptr_t>(g.B) % 16 == 0);
  dim3 grid(nwg), block(kThreads);
  if (va && vb)
    hipLaunchKernelGGL((GemmKernel<true, true>), grid, block, 0, Stream(), g);
  else if (va)
    hipLaunchKernelGGL((GemmKernel<true, false>), grid, block, 0, Stream(), g);
  else if (vb)
    hipLaunchKernelGGL((GemmKernel<false, true>), grid, block, 0, Stream(), g);
  else
    hipLaunchKernelGGL((GemmKernel<false, false>), grid, block, 0, Stream(), g);
  KH_LAUNCH_CHECK();
  return KH_OK;
}

}  // namespace

extern "C" {

int kh_add_mat_mat(float alpha, const float *A, KhMatrixDim dA, int transA,
                   const float *B, KhMatrixDim dB, int transB, float beta,
                   float *C, KhMatrixDim dC) {
  int rc = EnsureDevice();
  if (rc) return rc;
  KH_CHECK_ARG(A && B && C);
  KH_CHECK_ARG(dA.stride >= dA.cols && dB.stride >= dB.cols && dC.stride >= dC.cols);
  const int m = transA ? dA.cols : dA.rows, k = transA ? dA.rows : dA.cols;
  const int n = transB ? dB.rows : dB.cols, kb = transB ? dB.cols : dB.rows;
  // KALDI_ASSERT of cu-matrix.cc:959-967
  KH_CHECK_ARG(m == dC.rows && n == dC.cols && k == kb);
  GemmArgs g;
  g.A = A;
  g.B = B;
  g.C = C;
  g.bias = nullptr;
  g.M = m;
  g.N = n;
  g.K = k;
  g.a_si = transA ? 1 : dA.stride;
  g.a_sk = transA ? dA.stride : 1;
  g.b_sj = transB ? dB.stride : 1;
  g.b_sk = transB ? 1 : dB.stride;
  g.c_stride = dC.stride;
  g.alpha = alpha;
  g.beta = beta;
  return LaunchGemm(g);
}

int kh_affine(const float *A, KhMatrixDim dA, const float *W, KhMatrixDim dW,
              const float *bias, float *C, KhMatrixDim dC) {
  int rc = EnsureDevice();
  if (rc) return rc;
  KH_CHECK_ARG(A && W && C && bias);
  KH_CHECK_ARG(dA.cols == dW.cols && dC.rows == dA.rows && dC.cols == dW.rows);
  KH_CHECK_ARG(dA.stride >= dA.cols && dW.stride >= dW.cols && dC.stride >= dC.cols);
  GemmArgs g;
  g.A = A;
  g.B = W;
  g.C = C;
  g.bias = bias;
  g.M = dA.rows;
  g.N = dW.rows;
  g.K = dA.cols;
  g.a_si = dA.stride;
  g.a_sk = 1;
  g.b_sj = dW.stride;
  g.b_sk = 1;
  g.c_stride = dC.stride;
  g.alpha = 1.f;
  g.beta = 0.f;
  return LaunchGemm(g);
}

int kh_affine_pnorm_supported(int group_size) { return group_size >= 1 && group_size <= PBN && PBN % group_size == 0; }

int kh_affine_pnorm(const float *A, KhMatrixDim dA, const float *W, KhMatrixDim dW, const float *bias, float *Y,
                    KhMatrixDim dY, int group_size) {
  int rc = EnsureDevice();
  if (rc) return rc;
  KH_CHECK_ARG(A && W && Y && bias);
  KH_CHECK_ARG(kh_affine_pnorm_supported(group_size));
  KH_CHECK_ARG(dA.cols == dW.cols && dY.rows == dA.rows && dW.rows == dY.cols * group_size);
  KH_CHECK_ARG(dA.stride >= dA.cols && dW.stride >= dW.cols && dY.stride >= dY.cols);
  if (dA.rows == 0 || dY.cols == 0) return KH_OK;
  PnormArgs pa;
  GemmArgs &g = pa.g;
  g.A = A;
  g.B = W;
  g.C = nullptr;
  g.bias = bias;
  g.M = dA.rows;
  g.N = dW.rows;
  g.K = dA.cols;
  g.a_si = dA.stride;
  g.a_sk = 1;
  g.b_sj = dW.stride;
  g.b_sk = 1;
  g.c_stride = 0;
  g.alpha = 1.f;
  g.beta = 0.f;
  g.tiles_m = DivUp(g.M, PBM);
  g.tiles_n = DivUp(g.N, PBN);
  g.lane_offsets_ok = (g.a_si < (1L << 22) && g.b_sj < (1L << 22)) ? 1 : 0;
  pa.Y = Y;
  pa.y_stride = dY.stride;
  pa.group = group_size;
  const bool vec = (g.a_si % 4 == 0) && (g.b_sj % 4 == 0) && (reinterpret_cast<uintptr_t>(A) % 16 == 0) &&
                   (reinterpret_cast<uintptr_t>(W) % 16 == 0);
  dim3 grid(g.tiles_m * g.tiles_n), block(kThreads);
  if (vec)
    hipLaunchKernelGGL((GemmPnormKernel<true>), grid, block, 0, Stream(), pa);
  else
    hipLaunchKernelGGL((GemmPnormKernel<false>), grid, block, 0, Stream(), pa);
  KH_LAUNCH_CHECK();
  return KH_OK;
}

}  // extern "C"
