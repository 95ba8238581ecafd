// gemm_lab2.hip — round 5: the operand image in LDS as [row][16 k] with the k of a row permuted so that ONE ds_read_b128
// gives a lane its operand of four consecutive v_mfma_f32_32x32x2_f32 steps (lanes 0-31: k = 8g + 0,2,4,6; lanes 32-63:
// k = 8g + 1,3,5,7), and a thread's float4 of global memory goes to LDS as two ds_write_b64 — against the production
// image [k][row + 4] (one ds_read_b32 per operand and step, four ds_write_b32 per float4).  16-byte chunks of a row are
// XOR-swizzled with (row >> 2) & 3: the b128 reads of 16 consecutive rows cover the 64 banks once.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math tools/gemm_lab2.hip -o tools/gemm_lab2
//   tools/gemm_lab2 [group_m] [reps]
// Every variant must give the bits of the first row (the production tiling): one k-ordered fmaf chain per element.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x)                                                          \
  do {                                                                 \
    hipError_t e_ = (x);                                               \
    if (e_ != hipSuccess) {                                            \
      printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      exit(1);                                                         \
    }                                                                  \
  } while (0)

namespace {

constexpr int BK = 16;
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct GemmArgs {
  const float *A, *B;
  float *C;
  const float *bias;
  int M, N, K;
  long a_si, b_sj;
  int c_stride;
  int tiles_m, tiles_n, group_m;
};

__device__ __forceinline__ int XcdRemap(int bid, int nwg) {
  const int cpx = nwg >> 3, rem = nwg & 7;
  const int xcd = bid & 7, local = bid >> 3;
  return xcd < rem ? xcd * (cpx + 1) + local : rem * (cpx + 1) + (xcd - rem) * cpx + local;
}

__device__ __forceinline__ float4 LoadRow4(const float *base, long s_row, int row, int k, int rows, int K) {
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (row < rows) {
    const float *p = base + row * s_row + k;
    if (k + 3 < K) {
      v = *reinterpret_cast<const float4 *>(p);
    } else {
      if (k < K) v.x = p[0];
      if (k + 1 < K) v.y = p[1];
      if (k + 2 < K) v.z = p[2];
    }
  }
  return v;
}

__device__ __forceinline__ void TileOf(const GemmArgs &g, int *tm, int *tn) {
  const int nwg = g.tiles_m * g.tiles_n;
  const int tile = XcdRemap(blockIdx.x, nwg);
  const int per = g.group_m * g.tiles_n;
  const int gid = tile / per, in = tile - gid * per;
  const int first_m = gid * g.group_m;
  const int gsz = min(g.tiles_m - first_m, g.group_m);
  *tn = in / gsz;
  *tm = first_m + (in - *tn * gsz);
}

__device__ __forceinline__ void Prio() {
  switch ((blockIdx.x >> 8) & 3) {
    case 1: __builtin_amdgcn_s_setprio(1); break;
    case 2: __builtin_amdgcn_s_setprio(2); break;
    case 3: __builtin_amdgcn_s_setprio(3); break;
    default: break;
  }
}

// ---- the production tiling (kh_gemm.hip GemmKernel): [k][row + 4] image, 2 x 2 waves of 2 x 2 MFMA tiles
__device__ long long g_clk[4];   // shader clock / 100 MHz wall clock over the life of one workgroup in the middle of the grid
__device__ long long g_life[2 * 65536];   // wall clock (100 MHz) at the start and the end of every workgroup of the base kernel
__global__ void __launch_bounds__(256, 4) GemmBase(GemmArgs g) {
  constexpr int BM = 128, BN = 128, NT = 256, LA = BM + 4;
  const bool probe = blockIdx.x == gridDim.x / 2 && threadIdx.x == 0;
  long long c0 = 0, w0 = 0;
  if (probe) { c0 = clock64(); w0 = wall_clock64(); }
  if (threadIdx.x == 0 && blockIdx.x < 65536) g_life[2 * blockIdx.x] = wall_clock64();
  __shared__ float As[2][BK][LA];
  __shared__ float Bs[2][BK][LA];
  Prio();
  int tm, tn;
  TileOf(g, &tm, &tn);
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wm = wave >> 1, wn = wave & 1;
  const int lrow = t >> 2, lk = (t & 3) << 2, kk = lane >> 5, l31 = lane & 31;
  const int nk = (g.K + BK - 1) / BK;
  const int m0 = tm * BM, n0 = tn * BN, rowsA = g.M - m0, rowsB = g.N - n0;
  const float *Ab = g.A + static_cast<long>(m0) * g.a_si, *Bb = g.B + static_cast<long>(n0) * g.b_sj;
  const bool full = rowsA >= BM && rowsB >= BN;
  float4 ra[2], rb[2];
  unsigned offa[2], offb[2];
#pragma unroll
  for (int i = 0; i < 2; i++) {
    offa[i] = (static_cast<unsigned>(lrow + NT / 4 * i) * static_cast<unsigned>(g.a_si) + lk) * 4u;
    offb[i] = (static_cast<unsigned>(lrow + NT / 4 * i) * static_cast<unsigned>(g.b_sj) + lk) * 4u;
  }
  auto load_tile = [&](int k0) {
    if (full && k0 + BK <= g.K) {
      const char *pa = reinterpret_cast<const char *>(Ab + k0), *pb = reinterpret_cast<const char *>(Bb + k0);
#pragma unroll
      for (int i = 0; i < 2; i++) { ra[i] = *reinterpret_cast<const float4 *>(pa + offa[i]); rb[i] = *reinterpret_cast<const float4 *>(pb + offb[i]); }
    } else {
#pragma unroll
      for (int i = 0; i < 2; i++) {
        ra[i] = LoadRow4(Ab, g.a_si, lrow + NT / 4 * i, k0 + lk, rowsA, g.K);
        rb[i] = LoadRow4(Bb, g.b_sj, lrow + NT / 4 * i, k0 + lk, rowsB, g.K);
      }
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const int m = lrow + NT / 4 * i;
      As[buf][lk + 0][m] = ra[i].x; As[buf][lk + 1][m] = ra[i].y; As[buf][lk + 2][m] = ra[i].z; As[buf][lk + 3][m] = ra[i].w;
      Bs[buf][lk + 0][m] = rb[i].x; Bs[buf][lk + 1][m] = rb[i].y; Bs[buf][lk + 2][m] = rb[i].z; Bs[buf][lk + 3][m] = rb[i].w;
    }
  };
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
  load_tile(0);
  store_tile(0);
  __syncthreads();
  for (int kt = 0; kt < nk; kt++) {
    const int buf = kt & 1;
    if (kt + 1 < nk) load_tile((kt + 1) * BK);
#pragma unroll
    for (int s = 0; s < BK / 2; s++) {
      const int k = 2 * s + kk;
      float a0 = As[buf][k][wm * 64 + l31], a1 = As[buf][k][wm * 64 + 32 + l31];
      float b0 = Bs[buf][k][wn * 64 + l31], b1 = Bs[buf][k][wn * 64 + 32 + l31];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
    if (kt + 1 < nk) {
      store_tile(buf ^ 1);
      __syncthreads();
    }
  }
  if (m0 + BM <= g.M && n0 + BN <= g.N) {
    const int wu = __builtin_amdgcn_readfirstlane(t >> 6);
    const int col = n0 + (wu & 1) * 64 + l31;
    const float bv0 = g.bias[col], bv1 = g.bias[col + 32];
    float *cb = g.C + static_cast<size_t>(m0 + (wu >> 1) * 64) * g.c_stride + n0 + (wu & 1) * 64;
    const unsigned voff = static_cast<unsigned>(4 * kk) * g.c_stride + l31;
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int r = 0; r < 16; r++) {
        float *cp = cb + static_cast<size_t>(i * 32 + (r & 3) + 8 * (r >> 2)) * g.c_stride;
        cp[voff] = acc[i][0][r] + bv0;
        cp[voff + 32] = acc[i][1][r] + bv1;
      }
    if (probe) { g_clk[0] = clock64() - c0; g_clk[1] = wall_clock64() - w0; }
    if (threadIdx.x == 0 && blockIdx.x < 65536) g_life[2 * blockIdx.x + 1] = wall_clock64();
    return;
  }
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++) {
      const int col = n0 + wn * 64 + j * 32 + l31;
      if (col >= g.N) continue;
      const float bv = g.bias[col];
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk;
        if (row >= g.M) continue;
        g.C[static_cast<size_t>(row) * g.c_stride + col] = acc[i][j][r] + bv;
      }
    }
  if (threadIdx.x == 0 && blockIdx.x < 65536) g_life[2 * blockIdx.x + 1] = wall_clock64();
}

__device__ int g_tile_counter;
__global__ void __launch_bounds__(256, 4) GemmPersist(GemmArgs g) {
  constexpr int BM = 128, BN = 128, NT = 256, LA = BM + 4;
  __shared__ float As[2][BK][LA];
  __shared__ float Bs[2][BK][LA];
  Prio();
  __shared__ int s_tile;
  const int nwg_ = g.tiles_m * g.tiles_n;
  for (;;) {
  __syncthreads();   // (the previous tile's readers of the operand buffers and of s_tile are done)
  if (threadIdx.x == 0) s_tile = atomicAdd(&g_tile_counter, 1);
  __syncthreads();
  const int tile_ = s_tile;
  if (tile_ >= nwg_) break;
  int tm, tn;
  {
    const int per = g.group_m * g.tiles_n;
    const int gid = tile_ / per, in = tile_ - gid * per;
    const int first_m = gid * g.group_m;
    const int gsz = min(g.tiles_m - first_m, g.group_m);
    tn = in / gsz;
    tm = first_m + (in - tn * gsz);
  }
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wm = wave >> 1, wn = wave & 1;
  const int lrow = t >> 2, lk = (t & 3) << 2, kk = lane >> 5, l31 = lane & 31;
  const int nk = (g.K + BK - 1) / BK;
  const int m0 = tm * BM, n0 = tn * BN, rowsA = g.M - m0, rowsB = g.N - n0;
  const float *Ab = g.A + static_cast<long>(m0) * g.a_si, *Bb = g.B + static_cast<long>(n0) * g.b_sj;
  const bool full = rowsA >= BM && rowsB >= BN;
  float4 ra[2], rb[2];
  unsigned offa[2], offb[2];
#pragma unroll
  for (int i = 0; i < 2; i++) {
    offa[i] = (static_cast<unsigned>(lrow + NT / 4 * i) * static_cast<unsigned>(g.a_si) + lk) * 4u;
    offb[i] = (static_cast<unsigned>(lrow + NT / 4 * i) * static_cast<unsigned>(g.b_sj) + lk) * 4u;
  }
  auto load_tile = [&](int k0) {
    if (full && k0 + BK <= g.K) {
      const char *pa = reinterpret_cast<const char *>(Ab + k0), *pb = reinterpret_cast<const char *>(Bb + k0);
#pragma unroll
      for (int i = 0; i < 2; i++) { ra[i] = *reinterpret_cast<const float4 *>(pa + offa[i]); rb[i] = *reinterpret_cast<const float4 *>(pb + offb[i]); }
    } else {
#pragma unroll
      for (int i = 0; i < 2; i++) {
        ra[i] = LoadRow4(Ab, g.a_si, lrow + NT / 4 * i, k0 + lk, rowsA, g.K);
        rb[i] = LoadRow4(Bb, g.b_sj, lrow + NT / 4 * i, k0 + lk, rowsB, g.K);
      }
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const int m = lrow + NT / 4 * i;
      As[buf][lk + 0][m] = ra[i].x; As[buf][lk + 1][m] = ra[i].y; As[buf][lk + 2][m] = ra[i].z; As[buf][lk + 3][m] = ra[i].w;
      Bs[buf][lk + 0][m] = rb[i].x; Bs[buf][lk + 1][m] = rb[i].y; Bs[buf][lk + 2][m] = rb[i].z; Bs[buf][lk + 3][m] = rb[i].w;
    }
  };
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
  load_tile(0);
  store_tile(0);
  __syncthreads();
  for (int kt = 0; kt < nk; kt++) {
    const int buf = kt & 1;
    if (kt + 1 < nk) load_tile((kt + 1) * BK);
#pragma unroll
    for (int s = 0; s < BK / 2; s++) {
      const int k = 2 * s + kk;
      float a0 = As[buf][k][wm * 64 + l31], a1 = As[buf][k][wm * 64 + 32 + l31];
      float b0 = Bs[buf][k][wn * 64 + l31], b1 = Bs[buf][k][wn * 64 + 32 + l31];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
    if (kt + 1 < nk) {
      store_tile(buf ^ 1);
      __syncthreads();
    }
  }
  if (m0 + BM <= g.M && n0 + BN <= g.N) {
    const int wu = __builtin_amdgcn_readfirstlane(t >> 6);
    const int col = n0 + (wu & 1) * 64 + l31;
    const float bv0 = g.bias[col], bv1 = g.bias[col + 32];
    float *cb = g.C + static_cast<size_t>(m0 + (wu >> 1) * 64) * g.c_stride + n0 + (wu & 1) * 64;
    const unsigned voff = static_cast<unsigned>(4 * kk) * g.c_stride + l31;
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int r = 0; r < 16; r++) {
        float *cp = cb + static_cast<size_t>(i * 32 + (r & 3) + 8 * (r >> 2)) * g.c_stride;
        cp[voff] = acc[i][0][r] + bv0;
        cp[voff + 32] = acc[i][1][r] + bv1;
      }
    continue;
  }
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++) {
      const int col = n0 + wn * 64 + j * 32 + l31;
      if (col >= g.N) continue;
      const float bv = g.bias[col];
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk;
        if (row >= g.M) continue;
        g.C[static_cast<size_t>(row) * g.c_stride + col] = acc[i][j][r] + bv;
      }
    }
  }
}


// ---- [row][16 k] image, b128 operand reads.  WM x WN waves, each TM x TN MFMA tiles of 32 x 32.
template <int WM, int WN, int TM, int TN, int OCC, bool PRIO>
__global__ void __launch_bounds__(64 * WM * WN, OCC) GemmV(GemmArgs g) {
  constexpr int BM = 32 * WM * TM, BN = 32 * WN * TN, NT = 64 * WM * WN;
  constexpr int RA = BM * 4 / NT, RB = BN * 4 / NT;
  static_assert(RA * NT == BM * 4 && RB * NT == BN * 4, "whole float4 loads per thread");
  __shared__ __attribute__((aligned(16))) float As[2][BM][BK];
  __shared__ __attribute__((aligned(16))) float Bs[2][BN][BK];
  if (PRIO) Prio();
  int tm, tn;
  TileOf(g, &tm, &tn);
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int lrow = t >> 2, quad = t & 3, lk = quad << 2, kk = lane >> 5, l31 = lane & 31;
  const int nk = (g.K + BK - 1) / BK;
  const int m0 = tm * BM, n0 = tn * BN, rowsA = g.M - m0, rowsB = g.N - n0;
  const float *Ab = g.A + static_cast<long>(m0) * g.a_si, *Bb = g.B + static_cast<long>(n0) * g.b_sj;
  const bool full = rowsA >= BM && rowsB >= BN;
  float4 ra[RA], rb[RB];
  unsigned offa[RA], offb[RB];
#pragma unroll
  for (int i = 0; i < RA; i++) offa[i] = (static_cast<unsigned>(lrow + NT / 4 * i) * static_cast<unsigned>(g.a_si) + lk) * 4u;
#pragma unroll
  for (int i = 0; i < RB; i++) offb[i] = (static_cast<unsigned>(lrow + NT / 4 * i) * static_cast<unsigned>(g.b_sj) + lk) * 4u;
  auto load_fast = [&](int k0) {
    const char *pa = reinterpret_cast<const char *>(Ab + k0), *pb = reinterpret_cast<const char *>(Bb + k0);
#pragma unroll
    for (int i = 0; i < RA; i++) ra[i] = *reinterpret_cast<const float4 *>(pa + offa[i]);
#pragma unroll
    for (int i = 0; i < RB; i++) rb[i] = *reinterpret_cast<const float4 *>(pb + offb[i]);
  };
  auto load_tile = [&](int k0) {
#pragma unroll
    for (int i = 0; i < RA; i++) ra[i] = LoadRow4(Ab, g.a_si, lrow + NT / 4 * i, k0 + lk, rowsA, g.K);
#pragma unroll
    for (int i = 0; i < RB; i++) rb[i] = LoadRow4(Bb, g.b_sj, lrow + NT / 4 * i, k0 + lk, rowsB, g.K);
  };
  // a thread's float4 = k 4 quad .. 4 quad + 3 of one row: (x, z) go to the even-k chunk of the row's 8-group, (y, w) to the odd one
  const int grp = quad >> 1, q2 = (quad & 1) * 2;
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int i = 0; i < RA; i++) {
      const int m = lrow + NT / 4 * i, sw = (m >> 2) & 3;
      float *pe = &As[buf][m][((grp * 2 + 0) ^ sw) * 4 + q2], *po = &As[buf][m][((grp * 2 + 1) ^ sw) * 4 + q2];
      pe[0] = ra[i].x; pe[1] = ra[i].z;   // (two dwords of one ds_write2_b32: no register pair to assemble)
      po[0] = ra[i].y; po[1] = ra[i].w;
    }
#pragma unroll
    for (int i = 0; i < RB; i++) {
      const int m = lrow + NT / 4 * i, sw = (m >> 2) & 3;
      float *pe = &Bs[buf][m][((grp * 2 + 0) ^ sw) * 4 + q2], *po = &Bs[buf][m][((grp * 2 + 1) ^ sw) * 4 + q2];
      pe[0] = rb[i].x; pe[1] = rb[i].z;
      po[0] = rb[i].y; po[1] = rb[i].w;
    }
  };
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; i++)
#pragma unroll
    for (int j = 0; j < TN; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
  load_tile(0);
  store_tile(0);
  __syncthreads();
  const int swl = (l31 >> 2) & 3;   // (a tile's first row is a multiple of 32)
  auto mma = [&](int buf) {
#pragma unroll
    for (int g2 = 0; g2 < 2; g2++) {
      const int ch = ((g2 * 2 + kk) ^ swl) * 4;
      float4 af[TM], bf[TN];
#pragma unroll
      for (int i = 0; i < TM; i++) af[i] = *reinterpret_cast<const float4 *>(&As[buf][(wm * TM + i) * 32 + l31][ch]);
#pragma unroll
      for (int j = 0; j < TN; j++) bf[j] = *reinterpret_cast<const float4 *>(&Bs[buf][(wn * TN + j) * 32 + l31][ch]);
#pragma unroll
      for (int s = 0; s < 4; s++)
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
          for (int j = 0; j < TN; j++) {
            const float a = s == 0 ? af[i].x : s == 1 ? af[i].y : s == 2 ? af[i].z : af[i].w;
            const float b = s == 0 ? bf[j].x : s == 1 ? bf[j].y : s == 2 ? bf[j].z : bf[j].w;
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i][j], 0, 0, 0);
          }
    }
  };
  int kt = 0;
  if (full) {   // whole slabs of an interior tile: a loop without a branch in it
    const int nfast = g.K / BK;
    for (; kt + 1 < nfast; kt++) {
      load_fast((kt + 1) * BK);
      __builtin_amdgcn_sched_barrier(0);   // (the loads first: left to itself the scheduler sinks them to the end of the MFMA run)
      mma(kt & 1);
      __builtin_amdgcn_sched_barrier(0);
      store_tile((kt & 1) ^ 1);
      __syncthreads();
    }
  }
  for (; kt < nk; kt++) {
    const int buf = kt & 1;
    if (kt + 1 < nk) load_tile((kt + 1) * BK);
    mma(buf);
    if (kt + 1 < nk) {
      store_tile(buf ^ 1);
      __syncthreads();
    }
  }
  if (m0 + BM <= g.M && n0 + BN <= g.N) {
    float bv[TN];
#pragma unroll
    for (int j = 0; j < TN; j++) bv[j] = g.bias[n0 + (wn * TN + j) * 32 + l31];
    float *cb = g.C + static_cast<size_t>(m0 + wm * TM * 32) * g.c_stride + n0 + wn * TN * 32;
    const unsigned voff = static_cast<unsigned>(4 * kk) * g.c_stride + l31;
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
      for (int r = 0; r < 16; r++) {
        float *cp = cb + static_cast<size_t>(i * 32 + (r & 3) + 8 * (r >> 2)) * g.c_stride;
#pragma unroll
        for (int j = 0; j < TN; j++) cp[voff + 32 * j] = acc[i][j][r] + bv[j];
      }
    return;
  }
#pragma unroll
  for (int i = 0; i < TM; i++)
#pragma unroll
    for (int j = 0; j < TN; j++) {
      const int col = n0 + (wn * TN + j) * 32 + l31;
      if (col >= g.N) continue;
      const float bv = g.bias[col];
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int row = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk;
        if (row >= g.M) continue;
        g.C[static_cast<size_t>(row) * g.c_stride + col] = acc[i][j][r] + bv;
      }
    }
}


// ---- round 6: LDS-DMA.  The operand slabs go from global memory STRAIGHT into LDS (global_load_lds_dwordx4: gfx950 moves
// 16 bytes per lane, the wave's 64 chunks land side by side at M0) - no VGPR staging, no ds_write, the loads are not on the
// VALU path at all.  Image [row][4 chunks of 4 k], the chunk at position p of row r holds k-chunk p ^ ((r >> 2) & 3) (the
// swizzle is applied to the GLOBAL address of the lane whose chunk lands there); a lane reads chunk c of its row with one
// ds_read_b128 and uses elements kk, 2 + kk for the MFMA steps 2c, 2c + 1 (lanes 0-31: even k, lanes 32-63: odd k - the
// production kernel's k order, the same fmaf chain per element).  Two slabs in flight (even / odd buffers are separate
// objects so that the compiler's LDS-DMA alias tracking does not wait for the slab in flight before reading the other).
// The last slab of a K that is not a multiple of 16 goes through registers (a chunk cannot be masked by element).
typedef __attribute__((address_space(3))) void *LdsPtr;
typedef const __attribute__((address_space(1))) void *GlbPtr;
template <int OCC, bool PRIO>
__global__ void __launch_bounds__(256, OCC) GemmD(GemmArgs g) {
  constexpr int BM = 128, BN = 128;
  __shared__ __attribute__((aligned(16))) float As0[BM][BK];
  __shared__ __attribute__((aligned(16))) float As1[BM][BK];
  __shared__ __attribute__((aligned(16))) float Bs0[BN][BK];
  __shared__ __attribute__((aligned(16))) float Bs1[BN][BK];
  if (PRIO) Prio();
  int tm, tn;
  TileOf(g, &tm, &tn);
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int kk = lane >> 5, l31 = lane & 31;
  const int m0 = tm * BM, n0 = tn * BN, rowsA = g.M - m0, rowsB = g.N - n0;
  const float *Ab = g.A + static_cast<long>(m0) * g.a_si, *Bb = g.B + static_cast<long>(n0) * g.b_sj;
  // the wave's two DMA instructions per operand and slab: rows wave * 32 + 16 i + (lane >> 2), position lane & 3
  unsigned offa[2], offb[2];
#pragma unroll
  for (int i = 0; i < 2; i++) {
    const int r = wave * 32 + 16 * i + (lane >> 2), c = (lane & 3) ^ ((r >> 2) & 3);
    offa[i] = (static_cast<unsigned>(min(r, rowsA - 1)) * static_cast<unsigned>(g.a_si) + 4 * c) * 4u;   // (rows past the edge: the last valid row, never stored)
    offb[i] = (static_cast<unsigned>(min(r, rowsB - 1)) * static_cast<unsigned>(g.b_sj) + 4 * c) * 4u;
  }
  auto dma = [&](float (*As)[BK], float (*Bs)[BK], int k0) {
    const char *pa = reinterpret_cast<const char *>(Ab + k0), *pb = reinterpret_cast<const char *>(Bb + k0);
#pragma unroll
    for (int i = 0; i < 2; i++) {
      __builtin_amdgcn_global_load_lds((GlbPtr)(pa + offa[i]), (LdsPtr)&As[wave * 32 + 16 * i][0], 16, 0, 0);
      __builtin_amdgcn_global_load_lds((GlbPtr)(pb + offb[i]), (LdsPtr)&Bs[wave * 32 + 16 * i][0], 16, 0, 0);
    }
  };
  // The last slab of a K that is not a multiple of 16: its chunks are fetched like any other (a chunk that would start
  // beyond the row's stride is taken from the row's last whole chunk instead - any valid address), then the elements
  // k >= K are ZEROED in LDS before the barrier that publishes the slab (the production kernel feeds zeros there too: the
  // same fmaf chain).  Thread t owns row t of the 256 rows of the two operand slabs.
  const int k_tail = g.K & (BK - 1);
  unsigned offa_t[2], offb_t[2];
#pragma unroll
  for (int i = 0; i < 2; i++) {
    const int r = wave * 32 + 16 * i + (lane >> 2), c = (lane & 3) ^ ((r >> 2) & 3);
    const int k0 = g.K - k_tail;
    const int ca = min(c, (static_cast<int>(g.a_si) - k0) / 4 - 1), cb = min(c, (static_cast<int>(g.b_sj) - k0) / 4 - 1);
    offa_t[i] = (static_cast<unsigned>(min(r, rowsA - 1)) * static_cast<unsigned>(g.a_si) + 4 * ca) * 4u;
    offb_t[i] = (static_cast<unsigned>(min(r, rowsB - 1)) * static_cast<unsigned>(g.b_sj) + 4 * cb) * 4u;
  }
  auto dma_tail = [&](float (*As)[BK], float (*Bs)[BK], int k0) {
    const char *pa = reinterpret_cast<const char *>(Ab + k0), *pb = reinterpret_cast<const char *>(Bb + k0);
#pragma unroll
    for (int i = 0; i < 2; i++) {
      __builtin_amdgcn_global_load_lds((GlbPtr)(pa + offa_t[i]), (LdsPtr)&As[wave * 32 + 16 * i][0], 16, 0, 0);
      __builtin_amdgcn_global_load_lds((GlbPtr)(pb + offb_t[i]), (LdsPtr)&Bs[wave * 32 + 16 * i][0], 16, 0, 0);
    }
  };
  // (a wave zeroes the rows its OWN two DMA instructions per operand wrote - the vmcnt it has just waited for covers no
  // other wave's transfers: the first version had thread t zero row t, and a slab landing AFTER the zeroing gave wrong
  // tiles now and then - "BITS DIFFER" in two of three runs of round 6's first lab pass)
  auto zero_tail = [&](float (*As)[BK], float (*Bs)[BK]) {
    float (*X)[BK] = lane < 32 ? As : Bs;
    const int r = wave * 32 + (lane & 31), sw = (r >> 2) & 3;
    for (int k = k_tail; k < BK; k++) X[r][(((k >> 2) ^ sw) << 2) | (k & 3)] = 0.f;
  };
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
  const int swl = (l31 >> 2) & 3;
  auto mma = [&](const float (*As)[BK], const float (*Bs)[BK]) {
#pragma unroll
    for (int c = 0; c < 4; c++) {
      const int ch = (c ^ swl) * 4;
      float4 af[2], bf[2];
#pragma unroll
      for (int i = 0; i < 2; i++) af[i] = *reinterpret_cast<const float4 *>(&As[(wm * 2 + i) * 32 + l31][ch]);
#pragma unroll
      for (int j = 0; j < 2; j++) bf[j] = *reinterpret_cast<const float4 *>(&Bs[(wn * 2 + j) * 32 + l31][ch]);
#pragma unroll
      for (int s = 0; s < 2; s++)
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
          for (int j = 0; j < 2; j++) {
            const float a = s == 0 ? (kk ? af[i].y : af[i].x) : (kk ? af[i].w : af[i].z);
            const float b = s == 0 ? (kk ? bf[j].y : bf[j].x) : (kk ? bf[j].w : bf[j].z);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i][j], 0, 0, 0);
          }
    }
  };
  const int nfull = g.K / BK, nk = (g.K + BK - 1) / BK;   // whole slabs, all slabs (nk - nfull = 0 or 1)
  // slab kt -> the buffers; returns after the loads are ISSUED
#define KH_ISSUE(AS, BS, KT) do { if ((KT) < nfull) dma(AS, BS, (KT) * BK); else dma_tail(AS, BS, (KT) * BK); } while (0)
  // ... the slab is complete and visible to every wave
#define KH_PUBLISH(AS, BS, KT) do { __builtin_amdgcn_s_waitcnt(0x0f70); if ((KT) >= nfull) zero_tail(AS, BS); __syncthreads(); } while (0)
  KH_ISSUE(As0, Bs0, 0);
  KH_PUBLISH(As0, Bs0, 0);
  for (int kt = 0;; kt += 2) {
    const bool more1 = kt + 1 < nk;
    if (more1) KH_ISSUE(As1, Bs1, kt + 1);
    mma(As0, Bs0);
    if (!more1) break;
    KH_PUBLISH(As1, Bs1, kt + 1);
    const bool more2 = kt + 2 < nk;
    if (more2) KH_ISSUE(As0, Bs0, kt + 2);
    mma(As1, Bs1);
    if (!more2) break;
    KH_PUBLISH(As0, Bs0, kt + 2);
  }
#undef KH_ISSUE
#undef KH_PUBLISH
  if (m0 + BM <= g.M && n0 + BN <= g.N) {
    float bv[2];
#pragma unroll
    for (int j = 0; j < 2; j++) bv[j] = g.bias[n0 + (wn * 2 + j) * 32 + l31];
    float *cb = g.C + static_cast<size_t>(m0 + wm * 64) * g.c_stride + n0 + wn * 64;
    const unsigned voff = static_cast<unsigned>(4 * kk) * g.c_stride + l31;
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int r = 0; r < 16; r++) {
        float *cp = cb + static_cast<size_t>(i * 32 + (r & 3) + 8 * (r >> 2)) * g.c_stride;
#pragma unroll
        for (int j = 0; j < 2; j++) cp[voff + 32 * j] = acc[i][j][r] + bv[j];
      }
    return;
  }
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++) {
      const int col = n0 + (wn * 2 + j) * 32 + l31;
      if (col >= g.N) continue;
      const float bv = g.bias[col];
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int row = m0 + (wm * 2 + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk;
        if (row >= g.M) continue;
        g.C[static_cast<size_t>(row) * g.c_stride + col] = acc[i][j][r] + bv;
      }
    }
}
// ---- the same with a ring of three slabs
template <int OCC, bool PRIO>
__global__ void __launch_bounds__(256, OCC) GemmD3(GemmArgs g) {
  constexpr int BM = 128, BN = 128;
  __shared__ __attribute__((aligned(16))) float As0[BM][BK];
  __shared__ __attribute__((aligned(16))) float As1[BM][BK];
  __shared__ __attribute__((aligned(16))) float Bs0[BN][BK];
  __shared__ __attribute__((aligned(16))) float Bs1[BN][BK];
  __shared__ __attribute__((aligned(16))) float As2[BM][BK];
  __shared__ __attribute__((aligned(16))) float Bs2[BN][BK];
  if (PRIO) Prio();
  int tm, tn;
  TileOf(g, &tm, &tn);
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int kk = lane >> 5, l31 = lane & 31;
  const int m0 = tm * BM, n0 = tn * BN, rowsA = g.M - m0, rowsB = g.N - n0;
  const float *Ab = g.A + static_cast<long>(m0) * g.a_si, *Bb = g.B + static_cast<long>(n0) * g.b_sj;
  // the wave's two DMA instructions per operand and slab: rows wave * 32 + 16 i + (lane >> 2), position lane & 3
  unsigned offa[2], offb[2];
#pragma unroll
  for (int i = 0; i < 2; i++) {
    const int r = wave * 32 + 16 * i + (lane >> 2), c = (lane & 3) ^ ((r >> 2) & 3);
    offa[i] = (static_cast<unsigned>(min(r, rowsA - 1)) * static_cast<unsigned>(g.a_si) + 4 * c) * 4u;   // (rows past the edge: the last valid row, never stored)
    offb[i] = (static_cast<unsigned>(min(r, rowsB - 1)) * static_cast<unsigned>(g.b_sj) + 4 * c) * 4u;
  }
  auto dma = [&](float (*As)[BK], float (*Bs)[BK], int k0) {
    const char *pa = reinterpret_cast<const char *>(Ab + k0), *pb = reinterpret_cast<const char *>(Bb + k0);
#pragma unroll
    for (int i = 0; i < 2; i++) {
      __builtin_amdgcn_global_load_lds((GlbPtr)(pa + offa[i]), (LdsPtr)&As[wave * 32 + 16 * i][0], 16, 0, 0);
      __builtin_amdgcn_global_load_lds((GlbPtr)(pb + offb[i]), (LdsPtr)&Bs[wave * 32 + 16 * i][0], 16, 0, 0);
    }
  };
  // The last slab of a K that is not a multiple of 16: its chunks are fetched like any other (a chunk that would start
  // beyond the row's stride is taken from the row's last whole chunk instead - any valid address), then the elements
  // k >= K are ZEROED in LDS before the barrier that publishes the slab (the production kernel feeds zeros there too: the
  // same fmaf chain).  Thread t owns row t of the 256 rows of the two operand slabs.
  const int k_tail = g.K & (BK - 1);
  unsigned offa_t[2], offb_t[2];
#pragma unroll
  for (int i = 0; i < 2; i++) {
    const int r = wave * 32 + 16 * i + (lane >> 2), c = (lane & 3) ^ ((r >> 2) & 3);
    const int k0 = g.K - k_tail;
    const int ca = min(c, (static_cast<int>(g.a_si) - k0) / 4 - 1), cb = min(c, (static_cast<int>(g.b_sj) - k0) / 4 - 1);
    offa_t[i] = (static_cast<unsigned>(min(r, rowsA - 1)) * static_cast<unsigned>(g.a_si) + 4 * ca) * 4u;
    offb_t[i] = (static_cast<unsigned>(min(r, rowsB - 1)) * static_cast<unsigned>(g.b_sj) + 4 * cb) * 4u;
  }
  auto dma_tail = [&](float (*As)[BK], float (*Bs)[BK], int k0) {
    const char *pa = reinterpret_cast<const char *>(Ab + k0), *pb = reinterpret_cast<const char *>(Bb + k0);
#pragma unroll
    for (int i = 0; i < 2; i++) {
      __builtin_amdgcn_global_load_lds((GlbPtr)(pa + offa_t[i]), (LdsPtr)&As[wave * 32 + 16 * i][0], 16, 0, 0);
      __builtin_amdgcn_global_load_lds((GlbPtr)(pb + offb_t[i]), (LdsPtr)&Bs[wave * 32 + 16 * i][0], 16, 0, 0);
    }
  };
  // (a wave zeroes the rows its OWN two DMA instructions per operand wrote - the vmcnt it has just waited for covers no
  // other wave's transfers: the first version had thread t zero row t, and a slab landing AFTER the zeroing gave wrong
  // tiles now and then - "BITS DIFFER" in two of three runs of round 6's first lab pass)
  auto zero_tail = [&](float (*As)[BK], float (*Bs)[BK]) {
    float (*X)[BK] = lane < 32 ? As : Bs;
    const int r = wave * 32 + (lane & 31), sw = (r >> 2) & 3;
    for (int k = k_tail; k < BK; k++) X[r][(((k >> 2) ^ sw) << 2) | (k & 3)] = 0.f;
  };
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
  const int swl = (l31 >> 2) & 3;
  auto mma = [&](const float (*As)[BK], const float (*Bs)[BK]) {
#pragma unroll
    for (int c = 0; c < 4; c++) {
      const int ch = (c ^ swl) * 4;
      float4 af[2], bf[2];
#pragma unroll
      for (int i = 0; i < 2; i++) af[i] = *reinterpret_cast<const float4 *>(&As[(wm * 2 + i) * 32 + l31][ch]);
#pragma unroll
      for (int j = 0; j < 2; j++) bf[j] = *reinterpret_cast<const float4 *>(&Bs[(wn * 2 + j) * 32 + l31][ch]);
#pragma unroll
      for (int s = 0; s < 2; s++)
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
          for (int j = 0; j < 2; j++) {
            const float a = s == 0 ? (kk ? af[i].y : af[i].x) : (kk ? af[i].w : af[i].z);
            const float b = s == 0 ? (kk ? bf[j].y : bf[j].x) : (kk ? bf[j].w : bf[j].z);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i][j], 0, 0, 0);
          }
    }
  };
  const int nfull = g.K / BK, nk = (g.K + BK - 1) / BK;   // whole slabs, all slabs (nk - nfull = 0 or 1)
  // Three slabs: the one being read, the one that must have landed by the next barrier, the one in flight across it.
  // A slab is published by a COUNTED vmcnt (all of this wave's transfers but the youngest slab's four) + a raw s_barrier
  // (__syncthreads would drain the slab in flight: cdna_hip_programming.md, "Pipelining across barriers"); slab kt + 2 is
  // issued behind the barrier that publishes slab kt, i.e. when every wave has finished reading slab kt - 1 from its buffer.
  // The steady loop issues unconditionally (a conditional issue makes hipcc put its own vmcnt(0) in front of the reads);
  // the last two slabs are drained behind it through plain pointers.
#define KH_ISSUE(AS, BS, KT) do { if ((KT) < nfull) dma(AS, BS, (KT) * BK); else dma_tail(AS, BS, (KT) * BK); } while (0)
#define KH_PUB4() do { __builtin_amdgcn_s_waitcnt(0x0f74); __builtin_amdgcn_s_waitcnt(0xc07f); __builtin_amdgcn_s_barrier(); } while (0)
  if (nk >= 3) {
    KH_ISSUE(As0, Bs0, 0);
    KH_ISSUE(As1, Bs1, 1);
    int first = 0;   // the drain's first buffer
    for (int kt = 0;; kt += 3) {
      KH_PUB4(); KH_ISSUE(As2, Bs2, kt + 2); mma(As0, Bs0); if (kt + 2 == nk - 1) { first = 1; break; }
      KH_PUB4(); KH_ISSUE(As0, Bs0, kt + 3); mma(As1, Bs1); if (kt + 3 == nk - 1) { first = 2; break; }
      KH_PUB4(); KH_ISSUE(As1, Bs1, kt + 4); mma(As2, Bs2); if (kt + 4 == nk - 1) { first = 0; break; }
    }
    float (*A1)[BK] = first == 0 ? As0 : first == 1 ? As1 : As2, (*B1)[BK] = first == 0 ? Bs0 : first == 1 ? Bs1 : Bs2;
    float (*A2)[BK] = first == 0 ? As1 : first == 1 ? As2 : As0, (*B2)[BK] = first == 0 ? Bs1 : first == 1 ? Bs2 : Bs0;
    KH_PUB4();
    mma(A1, B1);
    __builtin_amdgcn_s_waitcnt(0x0f70);
    if (nk > nfull) zero_tail(A2, B2);
    __syncthreads();
    mma(A2, B2);
  } else {
    for (int kt = 0; kt < nk; kt++) {
      KH_ISSUE(As0, Bs0, kt);
      __builtin_amdgcn_s_waitcnt(0x0f70);
      if (kt >= nfull) zero_tail(As0, Bs0);
      __syncthreads();
      mma(As0, Bs0);
      __syncthreads();
    }
  }
#undef KH_ISSUE
#undef KH_PUB4
  if (m0 + BM <= g.M && n0 + BN <= g.N) {
    float bv[2];
#pragma unroll
    for (int j = 0; j < 2; j++) bv[j] = g.bias[n0 + (wn * 2 + j) * 32 + l31];
    float *cb = g.C + static_cast<size_t>(m0 + wm * 64) * g.c_stride + n0 + wn * 64;
    const unsigned voff = static_cast<unsigned>(4 * kk) * g.c_stride + l31;
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int r = 0; r < 16; r++) {
        float *cp = cb + static_cast<size_t>(i * 32 + (r & 3) + 8 * (r >> 2)) * g.c_stride;
#pragma unroll
        for (int j = 0; j < 2; j++) cp[voff + 32 * j] = acc[i][j][r] + bv[j];
      }
    return;
  }
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++) {
      const int col = n0 + (wn * 2 + j) * 32 + l31;
      if (col >= g.N) continue;
      const float bv = g.bias[col];
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int row = m0 + (wm * 2 + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk;
        if (row >= g.M) continue;
        g.C[static_cast<size_t>(row) * g.c_stride + col] = acc[i][j][r] + bv;
      }
    }
}
template <int OCC, bool PRIO>
void LaunchD3(GemmArgs g) {
  g.tiles_m = (g.M + 127) / 128;
  g.tiles_n = (g.N + 127) / 128;
  hipLaunchKernelGGL((GemmD3<OCC, PRIO>), dim3(g.tiles_m * g.tiles_n), dim3(256), 0, 0, g);
}
template <int OCC, bool PRIO>
void LaunchD(GemmArgs g) {
  g.tiles_m = (g.M + 127) / 128;
  g.tiles_n = (g.N + 127) / 128;
  hipLaunchKernelGGL((GemmD<OCC, PRIO>), dim3(g.tiles_m * g.tiles_n), dim3(256), 0, 0, g);
}

void LaunchPersist(GemmArgs g) {
  g.tiles_m = (g.M + 127) / 128;
  g.tiles_n = (g.N + 127) / 128;
  int zero = 0;
  CK(hipMemcpyToSymbolAsync(HIP_SYMBOL(g_tile_counter), &zero, sizeof(int), 0, hipMemcpyHostToDevice, 0));
  hipLaunchKernelGGL(GemmPersist, dim3(1024), dim3(256), 0, 0, g);
}
void LaunchBase(GemmArgs g) {
  g.tiles_m = (g.M + 127) / 128;
  g.tiles_n = (g.N + 127) / 128;
  hipLaunchKernelGGL(GemmBase, dim3(g.tiles_m * g.tiles_n), dim3(256), 0, 0, g);
}
template <int WM, int WN, int TM, int TN, int OCC, bool PRIO>
void LaunchV(GemmArgs g) {
  constexpr int BM = 32 * WM * TM, BN = 32 * WN * TN;
  g.tiles_m = (g.M + BM - 1) / BM;
  g.tiles_n = (g.N + BN - 1) / BN;
  hipLaunchKernelGGL((GemmV<WM, WN, TM, TN, OCC, PRIO>), dim3(g.tiles_m * g.tiles_n), dim3(64 * WM * WN), 0, 0, g);
}

typedef void (*LaunchFn)(GemmArgs);
struct Var {
  const char *name;
  LaunchFn fn;
};

}  // namespace

int main(int argc, char **argv) {
  const int group_m = argc > 1 ? atoi(argv[1]) : 8;
  const int reps = argc > 2 ? atoi(argv[2]) : 10;
  const Var vars[] = {
      {"production tiling 128x128 [k][row]", LaunchBase},
      {"the same, 1024 persistent workgroups + tile counter", LaunchPersist},
      {"LDS-DMA 128x128, 2x2 waves of 2x2, occ 4, prio", LaunchD<4, true>},
      {"LDS-DMA 128x128, 2x2 waves of 2x2, occ 4", LaunchD<4, false>},
      {"LDS-DMA 128x128, 2x2 waves of 2x2, occ 3, prio", LaunchD<3, true>},
      {"LDS-DMA ring of 3, counted vmcnt, occ 3, prio", LaunchD3<3, true>},
      {"LDS-DMA ring of 3, counted vmcnt, occ 3", LaunchD3<3, false>},
      {"LDS-DMA ring of 3, counted vmcnt, occ 2, prio", LaunchD3<2, true>},
      {"b128 128x128, 2x2 waves of 2x2, occ 4, prio", LaunchV<2, 2, 2, 2, 4, true>},
      {"b128 128x128, 2x2 waves of 2x2, occ 4", LaunchV<2, 2, 2, 2, 4, false>},
      {"b128 128x128, 2x2 waves of 2x2, occ 3", LaunchV<2, 2, 2, 2, 3, true>},
      {"b128 128x128, 4x1 waves of 1x4, occ 4", LaunchV<4, 1, 1, 4, 4, true>},
      {"b128 256x128, 2x2 waves of 4x2, occ 2", LaunchV<2, 2, 4, 2, 2, true>},
      {"b128 128x256, 2x2 waves of 2x4, occ 2", LaunchV<2, 2, 2, 4, 2, true>},
      {"b128 256x128, 4x2 waves of 2x2, occ 2", LaunchV<4, 2, 2, 2, 2, true>},
      {"b128 256x256, 2x2 waves of 4x4, occ 1", LaunchV<2, 2, 4, 4, 1, false>},
      {"b128 256x256, 4x2 waves of 2x4, occ 1", LaunchV<4, 2, 2, 4, 1, false>},
      {"b128 192x128, 2x2 waves of 3x2, occ 3", LaunchV<2, 2, 3, 2, 3, true>},
  };
  const int shapes[][3] = {{60000, 3500, 350}, {60000, 12000, 350}};
  for (const auto &sh : shapes) {
    const int M = sh[0], N = sh[1], K = sh[2];
    const int lda = (K + 3) & ~3, ldc = (N + 3) & ~3;
    std::vector<float> hA(static_cast<size_t>(M) * lda), hB(static_cast<size_t>(N) * lda), hbias(N);
    uint32_t s = 12345u + K + N;
    auto rnd = [&]() {
      s = s * 1664525u + 1013904223u;
      return (static_cast<int>(s >> 8) & 0xffff) / 32768.0f - 1.0f;
    };
    for (auto &x : hA) x = rnd();
    for (auto &x : hB) x = rnd() * 0.05f;
    for (auto &x : hbias) x = rnd();
    float *dA, *dB, *dC, *dbias;
    const size_t c_bytes = static_cast<size_t>(M) * ldc * 4;
    CK(hipMalloc(&dA, hA.size() * 4));
    CK(hipMalloc(&dB, hB.size() * 4));
    CK(hipMalloc(&dbias, N * 4));
    CK(hipMalloc(&dC, c_bytes));
    CK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dbias, hbias.data(), N * 4, hipMemcpyHostToDevice));
    GemmArgs g;
    g.A = dA; g.B = dB; g.C = dC; g.bias = dbias;
    g.M = M; g.N = N; g.K = K; g.a_si = lda; g.b_sj = lda; g.c_stride = ldc;
    g.group_m = group_m;
    printf("== M %d N %d K %d (group_m %d)\n", M, N, K, group_m);
    const double flop = 2.0 * M * N * K;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    std::vector<float> h0(static_cast<size_t>(M) * ldc), h1(static_cast<size_t>(M) * ldc);
    bool first = true;
    for (const Var &v : vars) {
      int same = 1;
      CK(hipMemset(dC, 0xff, c_bytes));
      v.fn(g);
      CK(hipDeviceSynchronize());
      if (first) {
        CK(hipMemcpy(h0.data(), dC, c_bytes, hipMemcpyDeviceToHost));
        first = false;
      } else {
        CK(hipMemcpy(h1.data(), dC, c_bytes, hipMemcpyDeviceToHost));
        for (int i = 0; i < M && same; i++)
          if (memcmp(&h0[static_cast<size_t>(i) * ldc], &h1[static_cast<size_t>(i) * ldc], N * 4)) same = 0;
      }
      for (int r = 0; r < 4 * reps; r++) v.fn(g);
      CK(hipEventRecord(e0, 0));
      for (int r = 0; r < reps; r++) v.fn(g);
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      ms /= reps;
      printf("%-48s %8.3f ms  %6.1f TFLOP/s  %s\n", v.name, ms, flop / ms / 1e9, same ? "bits ok" : "BITS DIFFER");
      if (v.fn == LaunchBase) {
        long long hc[4] = {0, 0, 0, 0};
        CK(hipMemcpyFromSymbol(hc, HIP_SYMBOL(g_clk), sizeof(hc)));
        {
          const int nwg = ((M + 127) / 128) * ((N + 127) / 128);
          std::vector<long long> life(2 * static_cast<size_t>(std::min(nwg, 65536)));
          CK(hipMemcpyFromSymbol(life.data(), HIP_SYMBOL(g_life), life.size() * 8));
          long long t0 = life[0], t1 = life[1];
          double sum = 0, mx = 0;
          for (size_t i = 0; i < life.size() / 2; i++) {
            t0 = std::min(t0, life[2 * i]); t1 = std::max(t1, life[2 * i + 1]);
            const double d = static_cast<double>(life[2 * i + 1] - life[2 * i]);
            sum += d; mx = std::max(mx, d);
          }
          const double span = static_cast<double>(t1 - t0);
          // occupancy over time in 20 slices
          std::vector<double> occ(20, 0.0);
          for (size_t i = 0; i < life.size() / 2; i++)
            for (int k = 0; k < 20; k++) {
              const double a = t0 + span * k / 20, b = t0 + span * (k + 1) / 20;
              const double lo = std::max<double>(a, life[2 * i]), hi = std::min<double>(b, life[2 * i + 1]);
              if (hi > lo) occ[k] += (hi - lo) / (b - a);
            }
          printf("    %d workgroups: first start to last end %.1f us; lifetime mean %.1f us, max %.1f us; workgroups alive, by twentieth of the span:",
                 nwg, span / 100.0, sum / (life.size() / 2) / 100.0, mx / 100.0);
          for (int k = 0; k < 20; k++) printf(" %.0f", occ[k]);
          printf("\n");
        }
        if (hc[1] > 0) printf("    shader clock under this load: %.0f MHz (%lld cycles in %.1f us, one workgroup)\n", hc[0] / (hc[1] / 100.0), hc[0], hc[1] / 100.0);
      }
    }
    CK(hipFree(dA)); CK(hipFree(dB)); CK(hipFree(dC)); CK(hipFree(dbias));
  }
  return 0;
}
