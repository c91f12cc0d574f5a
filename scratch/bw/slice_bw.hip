// Scratch microbenchmark (not product): useful HBM bandwidth of reading / writing a 128-byte channel slice of wider pixels.
// copy_k: every lane moves 16 bytes; a pixel's slice is 128 B = 8 lanes; src pixel pitch sp bytes (offset so), dst pitch dp (offset do).
#include <hip/hip_runtime.h>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void copy_k(const char* __restrict__ src, char* __restrict__ dst, long long npix, int sp, int so, int dp, int dof, int sb, int mode) {
  const long long nit = npix * (sb / 16);
  const int per = sb / 16;
  for (long long it = (long long)blockIdx.x * 256 + threadIdx.x; it < nit; it += (long long)gridDim.x * 256) {
    const long long px = it / per; const int c = (int)(it - px * per);
    u32x4 v = {1u, 2u, 3u, 4u};
    if (mode & 1) v = *(const u32x4*)(src + px * sp + so + c * 16);
    if (mode & 2) *(u32x4*)(dst + px * dp + dof + c * 16) = v;
    else if (v[0] == 0x12345678u && v[1] == 0x9abcdef0u) dst[0] = 1;
  }
}
extern "C" int slice_copy(const void* src, void* dst, long long npix, int sp, int so, int dp, int dof, int sb, int mode, int grid, void* stream) {
  hipLaunchKernelGGL(copy_k, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const char*)src, (char*)dst, npix, sp, so, dp, dof, sb, mode);
  return (int)hipGetLastError();
}
