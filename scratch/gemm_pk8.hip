// Micro-benchmark: split-f16 GEMM core with PRE-SPLIT operands (hi/lo f16 granules, "PK8") staged by LDS-DMA.
// 256 x 256 tile, 8 waves (2 M x 4 N, wave tile 128 x 64), K step = 32 pair-channels (128 B per row), two LDS stages.
// Question it answers: what does the deep structure buy over the 128x128 register-staged kernel (~285 TFLOP/s algorithmic)?
// build: hipcc --offload-arch=gfx950 -O3 -o gemm_pk8 gemm_pk8.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 256, BN = 256, ROWB = 128;            // bytes per row and K step
constexpr int STAGE = (BM + BN) * ROWB;                  // 64 KB

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
typedef __attribute__((address_space(3))) void* lds_ptr;

// A: [M][K4] bytes (K4 = K*4 bytes per row, granules of 32 B = 8 channels: hi 16 B | lo 16 B)
// B: [N][K4] bytes, same format
template <int SYNC>
__global__ __launch_bounds__(512) void gemm_pk8(const char* __restrict__ A, const char* __restrict__ B, float* __restrict__ C,
                                                int M, int N, int K4, int nk) {
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, kq = lane >> 5;
  const int wm = wave >> 2, wn = wave & 3;               // 2 x 4
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const __amdgpu_buffer_rsrc_t ra = make_rsrc(A + (long long)m0 * K4, (unsigned)(BM * K4));
  const __amdgpu_buffer_rsrc_t rb = make_rsrc(B + (long long)n0 * K4, (unsigned)(BN * K4));

  // DMA mapping: one wave instruction = 1 KB = 8 rows x 128 B; lane L -> row (L>>3), physical chunk (L&7) holding
  // logical chunk (L&7) ^ ((row>>1)&7).  A: 256 rows = 32 instructions, 4 per wave; B the same.
  int voff[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int row = (wave * 4 + j) * 8 + (lane >> 3);
    const int lc = (lane & 7) ^ ((row >> 1) & 7);
    voff[j] = row * K4 + lc * 16;
  }
  auto issue = [&](int stage, int k) {
    char* base = lds + stage * STAGE;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_ptr)(base + (wave * 4 + j) * 1024), 16, voff[j], k * ROWB, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_ptr)(base + BM * ROWB + (wave * 4 + j) * 1024), 16, voff[j], k * ROWB, 0, 0);
    }
  };

  f32x16 acc[4][2];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = (f32x16)(0.f);

  // fragment addresses: row r, logical chunk c -> byte r*128 + ((c ^ ((r>>1)&7)) * 16)
  int aoffs[4], boffs[2];
#pragma unroll
  for (int t = 0; t < 4; ++t) aoffs[t] = (wm * 128 + t * 32 + li) * ROWB;
#pragma unroll
  for (int t = 0; t < 2; ++t) boffs[t] = BM * ROWB + (wn * 64 + t * 32 + li) * ROWB;
  const int sw = (li >> 1) & 7;   // rows t*32 + li: (row>>1)&7 == (li>>1)&7 since 32 | t*32 and wm*128

  issue(0, 0);
  for (int k = 0; k < nk; ++k) {
    const int st = k & 1;
    if (k + 1 < nk) {
      issue(st ^ 1, k + 1);
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (SYNC) __syncthreads(); else __builtin_amdgcn_s_barrier();
    const char* base = lds + st * STAGE;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int ch = (2 * (2 * ks + kq)) ^ sw, cl = (2 * (2 * ks + kq) + 1) ^ sw;
      h8 ah[4], al[4], bh[2], bl[2];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        ah[t] = *(const h8*)(base + aoffs[t] + ch * 16);
        al[t] = *(const h8*)(base + aoffs[t] + cl * 16);
      }
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        bh[t] = *(const h8*)(base + boffs[t] + ch * 16);
        bl[t] = *(const h8*)(base + boffs[t] + cl * 16);
      }
#pragma unroll
      for (int tm = 0; tm < 4; ++tm)
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) {
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[tn], al[tm], acc[tm][tn], 0, 0, 0);
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl[tn], ah[tm], acc[tm][tn], 0, 0, 0);
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[tn], ah[tm], acc[tm][tn], 0, 0, 0);
        }
    }
    if (SYNC) __syncthreads(); else { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
  }
  // transposed product: lane = pixel li of the block, channels (r&3) + 8*(r>>2) + 4*kq
#pragma unroll
  for (int tm = 0; tm < 4; ++tm)
#pragma unroll
    for (int tn = 0; tn < 2; ++tn)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 128 + tm * 32 + li, n = n0 + wn * 64 + tn * 32 + (r & 3) + 8 * (r >> 2) + 4 * kq;
        C[(long long)m * N + n] = acc[tm][tn][r];
      }
}

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 76800, N = argc > 2 ? atoi(argv[2]) : 512, K = argc > 3 ? atoi(argv[3]) : 4608;
  const int K4 = K * 4, nk = K / 32;
  std::vector<_Float16> ha((size_t)M * K * 2), hb((size_t)N * K * 2);
  // small integers so that the result can be checked exactly: value = hi + lo with hi in {-2..2}, lo in {-1,0,1}/8
  for (size_t i = 0; i < ha.size(); ++i) ha[i] = (_Float16)((((i / 8) & 1) == 0) ? (float)((int)((i * 2654435761u) >> 29) - 3) : 0.125f * ((int)((i * 40503u) >> 30) - 1));
  for (size_t i = 0; i < hb.size(); ++i) hb[i] = (_Float16)((((i / 8) & 1) == 0) ? (float)((int)((i * 2246822519u) >> 29) - 4) : 0.125f * ((int)((i * 3266489917u) >> 30) - 2));
  char *dA, *dB; float* dC;
  hipMalloc(&dA, (size_t)M * K4); hipMalloc(&dB, (size_t)N * K4); hipMalloc(&dC, (size_t)M * N * 4);
  hipMemcpy(dA, ha.data(), (size_t)M * K4, hipMemcpyHostToDevice);
  hipMemcpy(dB, hb.data(), (size_t)N * K4, hipMemcpyHostToDevice);
  hipFuncSetAttribute((const void*)gemm_pk8<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE);
  hipFuncSetAttribute((const void*)gemm_pk8<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE);
  dim3 grid(M / BM, N / BN);
  for (int variant = 0; variant < 2; ++variant) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int it = 0; it < 2; ++it) {
      if (variant) hipLaunchKernelGGL(gemm_pk8<1>, grid, dim3(512), 2 * STAGE, 0, dA, dB, dC, M, N, K4, nk);
      else hipLaunchKernelGGL(gemm_pk8<0>, grid, dim3(512), 2 * STAGE, 0, dA, dB, dC, M, N, K4, nk);
    }
    hipDeviceSynchronize();
    hipEventRecord(e0);
    const int iters = 10;
    for (int it = 0; it < iters; ++it) {
      if (variant) hipLaunchKernelGGL(gemm_pk8<1>, grid, dim3(512), 2 * STAGE, 0, dA, dB, dC, M, N, K4, nk);
      else hipLaunchKernelGGL(gemm_pk8<0>, grid, dim3(512), 2 * STAGE, 0, dA, dB, dC, M, N, K4, nk);
    }
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= iters;
    printf("variant %s: M %d N %d K %d  %.3f ms  %.1f TFLOP/s algorithmic (x3 = %.0f f16 MFMA TFLOP/s)  err=%s\n", variant ? "syncthreads" : "raw barrier",
           M, N, K, ms, 2.0 * M * N * K / ms / 1e9, 6.0 * M * N * K / ms / 1e9, hipGetErrorString(hipGetLastError()));
  }
  // check a few entries on the host: C[m][n] = sum_k (ah*bh + al*bh + ah*bl)
  std::vector<float> hc((size_t)M * N);
  hipMemcpy(hc.data(), dC, hc.size() * 4, hipMemcpyDeviceToHost);
  double maxerr = 0;
  int shown = 0;
  for (int t = 0; t < 64; ++t) {
    const int m = (t * 7919) % M, n = (t * 104729) % N;
    double s = 0;
    for (int g = 0; g < K / 8; ++g)
      for (int j = 0; j < 8; ++j) {
        const double ah = (float)ha[((size_t)m * K / 8 + g) * 16 + j], al = (float)ha[((size_t)m * K / 8 + g) * 16 + 8 + j];
        const double bh = (float)hb[((size_t)n * K / 8 + g) * 16 + j], bl = (float)hb[((size_t)n * K / 8 + g) * 16 + 8 + j];
        s += ah * bh + al * bh + ah * bl;
      }
    const double e = fabs(s - hc[(size_t)m * N + n]);
    if (e > maxerr) maxerr = e;
    if (e > 1e-3 && shown < 8) { printf("  C[%d][%d] = %g, expected %g\n", m, n, hc[(size_t)m * N + n], s); ++shown; }
  }
  printf("max abs error on 64 sampled entries: %g\n", maxerr);
  return 0;
}
