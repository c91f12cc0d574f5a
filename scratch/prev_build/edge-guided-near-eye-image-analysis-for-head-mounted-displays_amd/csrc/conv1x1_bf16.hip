// 1x1 convolution over up to EGNE_MAXSEG raw bf16 slices on v_mfma_f32_16x16x32_bf16 (fp32 accumulate, bf16 output): the concat-free
// 1x1 convolutions of ESF-Net (models/RITnet_v2.py:59-61 conv21 / conv31, :38-41 Transition_down behind its pooling, :85-86
// conv11 / conv21 of the up blocks) and their merged data gradients in training plans with bf16 activation storage.
//
// Streaming form: the tensor IS the MFMA operand.  With the product transposed (weights as the A operand) a lane's B operand of a
// 32-channel k-step is "8 consecutive channels of one pixel" = ONE 16-byte global load, the four k-groups of a pixel are four
// neighbouring lanes -- 64 contiguous bytes per pixel and instruction, which is all a 32-channel slice has --, so activations never
// touch LDS or the vector ALU on the way in; the weights of the workgroup's output channels stay in LDS for the whole launch.
// What the first version got wrong (measured: 1.8 TB/s on the merged data gradients, behind the exact-fp32 implicit GEMM): it stored
// each lane's 4 result channels as they come out of the MFMA -- 16 contiguous bytes per pixel and instruction, and the same for the
// accumulated residual.  The memory pipeline works per cache line touched, not per byte: here a wave's 32 x CW result tile goes
// through LDS once (fp32) and leaves as 16-byte vectors of 8 channels with the lanes of a pixel side by side: whole 128-byte lines
// per pixel for 64 output channels, residual read the same way.
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr unsigned OOB = 0x80000000u;
constexpr int KC = 4;              // k-steps (of 32 channels) requested together: 2 x 4 loads of 16 bytes per lane in flight
constexpr int MAXKS = 48;          // k-steps of a launch (table in the kernel arguments)

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}

// per k-step: slice index and first channel of the step inside the slice
struct KTab { unsigned char seg[MAXKS]; unsigned short c0[MAXKS]; };

// NB16: 16-channel output blocks of this workgroup (CW = 16 NB16 channels; blockIdx.y selects the group)
template <int NB16>
__global__ __launch_bounds__(256)
void conv1x1_bf16_kernel(const egne_conv_desc p, const egne_bf16* __restrict__ wfrag, int nks, int nb16_total, KTab tab, long long M) {
  static_assert(NB16 == 2 || NB16 == 4, "the pixel-major store pattern needs 8 CW to divide 64 lanes");
  constexpr int CW = 16 * NB16, LDP = CW + 4;                           // result tile row pitch (floats)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  egne_bf16* const lw = (egne_bf16*)smem;                               // [nks][NB16][64 lanes][8]
  float* const tile = (float*)(smem + (size_t)nks * NB16 * 1024) + (threadIdx.x >> 6) * 32 * LDP;     // this wave's [32 px][LDP]
  const int tid = threadIdx.x, lane = tid & 63;
  const int l15 = lane & 15, kg = lane >> 4;
  const int b0 = blockIdx.y * NB16;
  for (int it = tid; it < nks * NB16 * 64; it += 256) {                 // 16-byte items
    const int l = it & 63, r = it >> 6, j = r % NB16, ks = r / NB16;
    const bool ok = b0 + j < nb16_total;
    const u32x4 v = ok ? *(const u32x4*)(wfrag + (((long long)ks * nb16_total + b0 + j) * 64 + l) * 8) : u32x4{0u, 0u, 0u, 0u};
    *(u32x4*)&lw[(long long)it * 8] = v;
  }
  __syncthreads();
  const float slope = p.act == EGNE_ACT_RELU ? 0.f : (p.act == EGNE_ACT_LEAKY ? 0.01f : 1.f);
  egne_bf16* const outp = (egne_bf16*)p.out;
  const egne_bf16* const resp = (const egne_bf16*)p.residual;
  const int cw0 = 16 * b0;                                               // first output channel of this workgroup
  constexpr int G = CW / 8;                                              // lanes per pixel on the way out (8 channels each)
  constexpr int PPI = 64 / G;                                            // pixels per store instruction
  const long long ngroups = (M + 31) / 32;
  const long long wave_id = (long long)blockIdx.x * 4 + (tid >> 6), nwaves = (long long)gridDim.x * 4;
  for (long long g = wave_id; g < ngroups; g += nwaves) {
    const long long m0 = g * 32;
    const int rows = (int)(M - m0 < 32 ? M - m0 : 32);
    f32x4 acc[2][NB16];
#pragma unroll
    for (int a = 0; a < 2 * NB16; ++a) (&acc[0][0])[a] = (f32x4)(0.f);
    for (int k0 = 0; k0 < nks; k0 += KC) {
      u32x4 xb[KC][2];
#pragma unroll
      for (int u = 0; u < KC; ++u) {
        const int ks = k0 + u;
        const bool on = ks < nks;
        const egne_seg& sg = p.seg[on ? tab.seg[ks] : 0];
        const int c = (on ? tab.c0[ks] : 0) + 8 * kg;
        const __amdgpu_buffer_rsrc_t r = make_rsrc((const egne_bf16*)sg.ptr + m0 * sg.pix_stride, (unsigned)rows * (unsigned)sg.pix_stride * 2u);
#pragma unroll
        for (int ph = 0; ph < 2; ++ph) {      // rows past M fall outside the resource and read zeros
          const int off = (on && c < sg.Cp) ? ((16 * ph + l15) * (int)sg.pix_stride + sg.ch_off + c) * 2 : (int)OOB;
          xb[u][ph] = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
        }
      }
#pragma unroll
      for (int u = 0; u < KC; ++u) {
        if (k0 + u < nks) {
#pragma unroll
          for (int j = 0; j < NB16; ++j) {
            const egne_bf16x8 a = *(const egne_bf16x8*)&lw[(((k0 + u) * NB16 + j) * 64 + lane) * 8];
#pragma unroll
            for (int ph = 0; ph < 2; ++ph)
              acc[ph][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, __builtin_bit_cast(egne_bf16x8, xb[u][ph]), acc[ph][j], 0, 0, 0);
          }
        }
      }
    }
    // lane holds pixel 16 ph + l15, channels 16 j + 4 kg + r: through the wave's LDS tile into pixel-major 8-channel vectors
#pragma unroll
    for (int ph = 0; ph < 2; ++ph)
#pragma unroll
      for (int j = 0; j < NB16; ++j) {
        const int n = cw0 + 16 * j + 4 * kg;
        const f32x4 bv = (p.bias && n < p.Cout_store) ? *(const f32x4*)(p.bias + n) : (f32x4)(0.f);
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float t = acc[ph][j][e] + bv[e];
          v[e] = fmaxf(t, t * slope);
        }
        *(f32x4*)&tile[(16 * ph + l15) * LDP + 16 * j + 4 * kg] = v;
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // (one wave: its own LDS writes are visible to its own reads after this)
    const __amdgpu_buffer_rsrc_t rout = make_rsrc(outp + m0 * p.out_pix_stride, (unsigned)rows * (unsigned)p.out_pix_stride * 2u);
    const __amdgpu_buffer_rsrc_t rres = make_rsrc(resp ? resp + m0 * p.res_pix_stride : nullptr, resp ? (unsigned)rows * (unsigned)p.res_pix_stride * 2u : 0u);
#pragma unroll
    for (int i = 0; i < 32 / PPI; ++i) {
      const int px = i * PPI + lane / G, cg = lane % G;
      const int n = cw0 + 8 * cg;
      const bool nok = n < p.Cout_store;                     // Cout_store is a multiple of 8
      egne_fv<8> v;
      const f32x4 t0 = *(const f32x4*)&tile[px * LDP + 8 * cg], t1 = *(const f32x4*)&tile[px * LDP + 8 * cg + 4];
#pragma unroll
      for (int e = 0; e < 4; ++e) { v.v[e] = t0[e]; v.v[4 + e] = t1[e]; }
      if (resp) {
        const u32x4 rw = __builtin_amdgcn_raw_buffer_load_b128(rres, nok ? (px * (int)p.res_pix_stride + p.res_ch_off + n) * 2 : (int)OOB, 0, 0);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v.v[2 * e] += __builtin_bit_cast(float, rw[e] << 16);
          v.v[2 * e + 1] += __builtin_bit_cast(float, rw[e] & 0xffff0000u);
        }
      }
      const f32x4 lo = {v.v[0], v.v[1], v.v[2], v.v[3]}, hi = {v.v[4], v.v[5], v.v[6], v.v[7]};
      const egne_bf16x4 l4 = __builtin_convertvector(lo, egne_bf16x4), h4 = __builtin_convertvector(hi, egne_bf16x4);
      const egne_bf16x8 pk = {l4[0], l4[1], l4[2], l4[3], h4[0], h4[1], h4[2], h4[3]};
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, pk), rout,
                                             nok ? (px * (int)p.out_pix_stride + p.out_ch_off + n) * 2 : (int)OOB, 0, 0);
    }
  }
}

// flat fp32 pack [CoutP][Ktot] (egne_pack_conv_weight / egne_pack_conv_weight_dgrad with kh = kw = 1) -> bf16 fragments
// [k-step][CoutP/16][lane = kg*16 + n%16][8]: element j of lane (n, kg) = W[n][kofs(step) + 8 kg + j], zero beyond the slice
__global__ void pack_conv1x1_bf16_k(const float* __restrict__ wflat, int CoutP, int Ktot, int nks, KTab tab, const int* __restrict__ kofs,
                                    const int* __restrict__ segcp, egne_bf16* __restrict__ out) {
  const long long total = (long long)nks * CoutP * 32;
  const int nb16 = CoutP >> 4;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int j = (int)(i & 7), l = (int)((i >> 3) & 63);
    long long q = i >> 9;
    const int cb = (int)(q % nb16), ks = (int)(q / nb16);
    const int n = cb * 16 + (l & 15), c = tab.c0[ks] + 8 * (l >> 4) + j;
    const int s = tab.seg[ks];
    out[i] = (egne_bf16)(c < segcp[s] ? wflat[(long long)n * Ktot + kofs[s] + c] : 0.f);
  }
}

bool make_tab(const egne_conv_desc& d, KTab* tab, int* nks_) {
  int nks = 0;
  for (int s = 0; s < d.nseg; ++s)
    for (int c0 = 0; c0 < d.seg[s].Cp; c0 += 32) {
      if (nks >= MAXKS) return false;
      tab->seg[nks] = (unsigned char)s; tab->c0[nks] = (unsigned short)c0; ++nks;
    }
  *nks_ = nks;
  return true;
}

// 16-channel output blocks per workgroup and the LDS bytes that takes: all of them up to 64 channels, else pieces of 64 or 32
// (2 or 4: the pixel-major store pattern needs 8 CW to divide 64 lanes; CoutP is a multiple of 32, so nb16 is even)
int blocks_per_wg(int nb16) { return nb16 <= 4 ? nb16 : (nb16 % 4 == 0 ? 4 : 2); }
size_t lds_bytes(int nks, int nb) { return (size_t)nks * nb * 1024 + (size_t)4 * 32 * (16 * nb + 4) * sizeof(float); }

}  // namespace

// number of bf16 elements of the fragment pack of a descriptor (k-steps x CoutP x 32), or -1 if the launch is not supported
extern "C" int64_t egne_conv1x1_bf16_pack_elems(const egne_conv_desc* dp) {
  if (!dp) return -1;
  KTab tab; int nks = 0;
  if (!make_tab(*dp, &tab, &nks)) return -1;
  if (lds_bytes(nks, blocks_per_wg(dp->CoutP / 16)) > 120 * 1024) return -1;
  return (int64_t)nks * dp->CoutP * 32;
}

// wflat: device pointer to the fp32 pack [CoutP][Ktot] of the same descriptor; seginfo: device int32 [2 * nseg] = the K offset of
// every slice, then the padded channel count of every slice
extern "C" int egne_pack_conv1x1_bf16(const egne_conv_desc* dp, const float* wflat, const int32_t* seginfo, void* wfrag, void* stream) {
  EGNE_REQUIRE(dp && wflat && seginfo && wfrag, "pack_conv1x1_bf16: null pointer");
  const egne_conv_desc& d = *dp;
  KTab tab; int nks = 0;
  EGNE_REQUIRE(d.nseg >= 1 && d.nseg <= EGNE_MAXSEG && d.CoutP % 32 == 0 && make_tab(d, &tab, &nks), "pack_conv1x1_bf16: too many k-steps");
  long long total = (long long)nks * d.CoutP * 32, g = (total + 255) / 256;
  if (g > 2048) g = 2048;
  hipLaunchKernelGGL(pack_conv1x1_bf16_k, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, wflat, d.CoutP, d.Ktot, nks, tab, seginfo,
                     seginfo + d.nseg, (egne_bf16*)wfrag);
  return egne::check_launch("egne_pack_conv1x1_bf16");
}

extern "C" int egne_conv1x1_bf16_fwd(const egne_conv_desc* dp, const void* wfrag, void* stream) {
  EGNE_REQUIRE(dp && wfrag, "conv1x1_bf16: null pointer");
  const egne_conv_desc& d = *dp;
  EGNE_REQUIRE(d.dtype == 1, "conv1x1_bf16: the descriptor must say bf16 tensors (dtype 1)");
  EGNE_REQUIRE(d.kh == 1 && d.kw == 1 && d.stride == 1 && d.pad_h == 0 && d.pad_w == 0 && d.ngroups == 1 && d.Ho == d.H && d.Wo == d.W &&
               d.nseg >= 1 && d.nseg <= EGNE_MAXSEG && !d.post_scale && !d.stats_ws && !d.pool_out && !d.dyn_scale && !d.absmax_out,
               "conv1x1_bf16: geometry / options not supported");
  int ktot = 0;
  for (int s = 0; s < d.nseg; ++s) {
    const egne_seg& g = d.seg[s];
    EGNE_REQUIRE(g.ptr && !g.scale && !g.shift && g.Cp % 8 == 0 && g.ch_off % 8 == 0 && g.pix_stride % 8 == 0 && ((uintptr_t)g.ptr & 15) == 0 &&
                 g.ch_off + g.Cp <= g.pix_stride && g.pix_stride * 64 < (1ll << 31), "conv1x1_bf16: slice %d (raw, 16-byte groups of 8 channels)", s);
    ktot += g.Cp;
  }
  EGNE_REQUIRE(ktot == d.Ktot && d.CoutP % 32 == 0 && d.Cout_store % 8 == 0 && d.Cout_store <= d.CoutP && d.out && ((uintptr_t)d.out & 15) == 0 &&
               d.out_pix_stride % 8 == 0 && d.out_ch_off % 8 == 0 && d.out_ch_off + d.Cout_store <= d.out_pix_stride &&
               d.out_pix_stride * 64 < (1ll << 31) && (!d.bias || ((uintptr_t)d.bias & 15) == 0), "conv1x1_bf16: output (16-byte groups of 8 channels)");
  EGNE_REQUIRE(!d.residual || (((uintptr_t)d.residual & 15) == 0 && d.res_pix_stride % 8 == 0 && d.res_ch_off % 8 == 0 && d.res_pix_stride * 64 < (1ll << 31)),
               "conv1x1_bf16: residual alignment");
  KTab tab; int nks = 0;
  EGNE_REQUIRE(make_tab(d, &tab, &nks), "conv1x1_bf16: more than %d k-steps", MAXKS);
  const int nb16 = d.CoutP / 16, nb = blocks_per_wg(nb16);
  const size_t lds = lds_bytes(nks, nb);
  EGNE_REQUIRE(lds <= 120 * 1024, "conv1x1_bf16: %zu bytes of LDS per workgroup exceed the budget", lds);
  const long long M = (long long)d.B * d.H * d.W;
  const int gy = (nb16 + nb - 1) / nb;
  long long gx = ((M + 31) / 32 + 3) / 4;
  long long cap = 256ll * (lds <= 36 * 1024 ? 4 : (lds <= 76 * 1024 ? 2 : 1)) / gy;     // workgroups: as many as stay resident
  if (cap < 1) cap = 1;
  if (gx > cap) gx = cap;
  hipStream_t st = (hipStream_t)stream;
  auto go = [&](auto kern) -> int {
    static const bool raised = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024) == hipSuccess;
    if (!raised) return egne::fail(EGNE_ERR_LAUNCH, "conv1x1_bf16: cannot raise the dynamic LDS limit");
    hipLaunchKernelGGL(kern, dim3((unsigned)gx, gy), dim3(256), lds, st, d, (const egne_bf16*)wfrag, nks, nb16, tab, M);
    return egne::check_launch("egne_conv1x1_bf16_fwd");
  };
  if (nb == 2) return go(conv1x1_bf16_kernel<2>);
  return go(conv1x1_bf16_kernel<4>);
}
