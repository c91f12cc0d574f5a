// Scratch microbenchmark (not product): store-only throughput of the two epilogue shapes into a 128-byte slice of 512-byte pixels.
//   shape 0 (epi_row32): lane = channel n (32 lanes) x pixel xl + 4*lh; 16 dword stores per 32-pixel row block, pixel c_r = (r&3) + 8*(r>>2)
//   shape 1 (transposed product): lane = pixel (32 lanes) x 16-byte half ... 4 x 16-byte stores per lane cover its pixel's 4*16 B... (lh picks the 32-B pair)
#include <hip/hip_runtime.h>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
template <int SHAPE>
__global__ __launch_bounds__(256) void store_k(char* __restrict__ dst, long long nrow32, int pitch, int off, int spin) {
  const int lane = threadIdx.x & 63, li = lane & 31, lh = lane >> 5;
  const long long wave = ((long long)blockIdx.x * 256 + threadIdx.x) >> 6, nw = ((long long)gridDim.x * 256) >> 6;
  float a = (float)lane;
  for (long long rb = wave; rb < nrow32; rb += nw) {            // one 32-pixel row block per trip
    for (int k = 0; k < spin; ++k) a = a * 1.0001f + 0.5f;       // stand-in for the matrix work between epilogues
    const __amdgpu_buffer_rsrc_t r = make_rsrc(dst + rb * 32 * pitch, 32u * pitch);
    if (SHAPE == 0) {
      const int vo = (4 * lh) * pitch + off + li * 4;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int c = (q & 3) + 8 * (q >> 2);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, a + q), r, vo, c * pitch, 0);
      }
    } else {
      const int vo = li * pitch + off + lh * 16;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        u32x4 v = {__builtin_bit_cast(unsigned, a + q), 1u, 2u, 3u};
        __builtin_amdgcn_raw_buffer_store_b128(v, r, vo, q * 32, 0);
      }
    }
  }
}
extern "C" int store_bw(void* dst, long long nrow32, int pitch, int off, int shape, int spin, int grid, void* stream) {
  if (shape == 0) hipLaunchKernelGGL(store_k<0>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (char*)dst, nrow32, pitch, off, spin);
  else hipLaunchKernelGGL(store_k<1>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (char*)dst, nrow32, pitch, off, spin);
  return (int)hipGetLastError();
}
