// Implicit-GEMM convolution of the wide VGG-trunk layers on PLAIN f16 operands (egne_conv_desc.f16_products = 1: one
// v_mfma_f32_16x16x32_f16 per product, fp32 tensors in HBM, fp32 accumulate) -- the frozen edge network next to a training plan with
// bf16 activation storage (vgg16_c.py:72-88 under utils.py:646; train.py --prec 16).  Same tile as conv_f16x3_big.hip (256 pixels x
// BN output channels, 8 waves of 128 x 64, K step = 32 channels of one tap, weights as ready-made LDS images staged by LDS-DMA,
// transposed product, 16-byte stores) and the same accumulation order, so the two kernels agree bit for bit.
//
// Why a kernel of its own: with one product instead of three a K step holds 32 MFMAs per wave (1 024 cycles per SIMD), and the
// two-stage loop of conv_f16x3_big.hip -- request, wait, convert, barrier, read fragments, multiply -- leaves the matrix pipe idle
// for longer than that per step (733-770 TFLOP/s where the three-product form issues 1 245 TFLOP/s worth of MFMAs).  Here
//   * an LDS row is 64 bytes (hi halves only), a stage 32 KB, and FOUR stages form a ring: the activations of step t + 2 are
//     converted and written, and the weight image of step t + 3 requested, while step t is multiplied -- global loads have two
//     steps to arrive (counted `s_waitcnt vmcnt`, never 0 inside the loop);
//   * a stage is complete ONE barrier before its step, so the fragments of the next step's first half (and its weights) are read
//     into registers during the second half of this one: the MFMAs behind a barrier start at once;
//   * the 16-byte chunk c of row r sits at chunk c ^ g((r >> 2) & 3), g = (0, 2, 3, 1): conflict free for the lane groups of
//     ds_read_b128 in the 16 x 16 x 32 operand pattern (16 rows x 4 chunks) and for the ds_write_b64 of the conversion.
#include "common.h"
#include <type_traits>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void* lds_ptr;

namespace {

constexpr unsigned OOB = 0x80000000u;
constexpr int BM = 256, ROWB = 64, NSTG = 4;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}

// chunk swizzle key of row r
__host__ __device__ __forceinline__ constexpr int swz(int r) {
  const int q = (r >> 2) & 3;
  return (((q ^ (q >> 1)) & 1) << 1) | (q >> 1);
}

// NB = 32-channel blocks per wave along N (2 -> BN = 256 with 4 waves along N, 1 -> BN = 128)
// F16IN: the input slice is held as f16 (egne_seg.presplit = 2: halves of x * a_scale in channel order): the activations are staged like the
// weights -- by LDS-DMA, three steps ahead, no registers, no conversion (lane l of a 16-row instruction: row l >> 2, LDS chunk l & 3 <- the row's
// chunk (l & 3) ^ key(row); taps outside the image and rows past the batch carry the out-of-range offset and land as zeros)
// (the body is a __device__ function behind two thin kernels: with F16IN as a parameter of the KERNEL template hipcc's host pass dropped the
//  stubs of the F16IN = true instantiations without a diagnostic)
template <int NB, bool F16IN>
__device__ __forceinline__ void conv_f16_big1_body(const egne_conv_desc& p, const char* __restrict__ wimg, float a_scale, float out_scale) {
  constexpr int BN = 128 * NB;
  constexpr int STAGE = (BM + BN) * ROWB;
  constexpr int NTN = 2 * NB;                            // 16-channel blocks per wave
  constexpr int NVM = (F16IN ? 2 : 4) + NB;              // vector-memory instructions a wave issues per step
  constexpr int ESZ = F16IN ? 2 : 4;
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);       // (uniform: the LDS-DMA's base goes through M0 without a loop)
  const int wm = wave >> 2, wn = wave & 3;               // 2 x 4 waves; wave tile 128 x (32 * NB)
  const long long M = (long long)p.B * p.Ho * p.Wo;
  const long long m0 = (long long)blockIdx.x * BM;
  const int ntile = blockIdx.y;
  const int T = p.kh * p.kw;
  const egne_seg sg = p.seg[0];
  const int hw = p.Ho * p.Wo, frame_px = p.H * p.W;
  const int b0 = (int)(m0 / hw);
  const long long in_left = ((long long)p.B - b0) * frame_px * sg.pix_stride * ESZ;
  const __amdgpu_buffer_rsrc_t rin = make_rsrc((const char*)sg.ptr + (long long)b0 * frame_px * sg.pix_stride * ESZ,
                                               (unsigned)(in_left < 0x7fffffffll ? in_left : 0x7fffffffll));
  const int nchunk = sg.Cp >> 5;
  const int nsteps = T * nchunk;                         // even (launcher)
  // weight images: [ntile][step = chunk*T + tap][BN rows x 64 B]
  const __amdgpu_buffer_rsrc_t rw = make_rsrc(wimg + (long long)ntile * nsteps * (BN * ROWB), (unsigned)(nsteps * BN * ROWB));

  // ---- activation staging.  fp32 input: thread -> 4 items (row = (tid>>3) + 64*i, float4 column c4 = tid&7); f16 input: lane -> 2 DMA
  //      pieces (row = 32 wave + 16 i + (lane >> 2), source chunk (lane & 3) ^ key(row)) ----
  constexpr int NIT = F16IN ? 2 : 4;
  const int c4 = tid & 7;
  int roff[NIT];
  unsigned tapmask[NIT];
  {
    const int dil = p.dil[0];
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      const int row_i = F16IN ? 32 * wave + 16 * i + (lane >> 2) : (tid >> 3) + 64 * i;
      const long long m = m0 + row_i;
      const int b = (int)(m / hw);
      const int r = (int)(m - (long long)b * hw);
      const int oy = r / p.Wo, ox = r - oy * p.Wo;
      roff[i] = F16IN ? ((((b - b0) * p.H + oy) * p.W + ox) * (int)sg.pix_stride + sg.ch_off + (((lane & 3) ^ swz(row_i)) << 3)) * 2
                      : ((((b - b0) * p.H + oy) * p.W + ox) * (int)sg.pix_stride + sg.ch_off + c4 * 4) * 4;
      unsigned mk = 0;
      for (int ky = 0; ky < p.kh; ++ky)
        for (int kx = 0; kx < p.kw; ++kx) {
          const int iy = oy + (ky - p.pad_h) * dil, ix = ox + (kx - p.pad_w) * dil;
          if (m < M && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) mk |= 1u << (ky * p.kw + kx);
        }
      tapmask[i] = mk;
    }
  }
  // LDS destination of item i: row r, chunk (c4 >> 1) ^ key(r), 8 bytes at (c4 & 1) * 8; r >> 2 = (tid >> 5) + 16 i: one key for all i
  const int ldst0 = (tid >> 3) * ROWB + ((((c4 >> 1) ^ swz(tid >> 3))) << 4) + (c4 & 1) * 8;
  const int wvoff = lane * 16;

  u32x4 ra[2][F16IN ? 1 : 4];
  int ky_n = 0, kx_n = 0, c0_n = 0, tap_n = 0;          // coordinates of the step load_a / dma_a requests next
  auto advance = [&]() {
    if (++kx_n == p.kw) { kx_n = 0; ++ky_n; }
    if (ky_n == p.kh) { ky_n = 0; c0_n += 32; }
    tap_n = tap_n + 1 == T ? 0 : tap_n + 1;
  };
  auto load_a = [&](u32x4* dst, bool on) {
    const int dil = p.dil[0];
    const int tapoff = (((ky_n - p.pad_h) * p.W + (kx_n - p.pad_w)) * dil * (int)sg.pix_stride + c0_n) * 4;
#pragma unroll
    for (int i = 0; i < (F16IN ? 0 : 4); ++i) {
      const bool ok = on && ((tapmask[i] >> tap_n) & 1u);
      dst[i] = __builtin_amdgcn_raw_buffer_load_b128(rin, ok ? roff[i] + tapoff : (int)OOB, 0, 0);
    }
    advance();
  };
  // f16 input: the step's 256 rows x 64 B straight into ring slot `slot` (16 instructions of 1 KB, two per wave)
  auto dma_a = [&](int slot, bool on) {
    const int dil = p.dil[0];
    const int tapoff = (((ky_n - p.pad_h) * p.W + (kx_n - p.pad_w)) * dil * (int)sg.pix_stride + c0_n) * 2;
    char* base = lds + slot * STAGE + wave * 2048;
#pragma unroll
    for (int i = 0; i < (F16IN ? 2 : 0); ++i) {
      const bool ok = on && ((tapmask[i] >> tap_n) & 1u);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rin, (lds_ptr)(base + i * 1024), 16, ok ? roff[i] + tapoff : (int)OOB, 0, 0, 0);
    }
    advance();
  };
  // weight image of `step` into ring slot `slot`: BN rows x 64 B = BN / 16 wave instructions of 1 KB, NB per wave; already swizzled
  auto dma_b = [&](int slot, int step) {
    char* base = lds + slot * STAGE + BM * ROWB;
    const int s = step < nsteps ? step : nsteps - 1;     // (past the end: a valid image into a slot nobody reads)
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int blk = wave * NB + j;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr)(base + blk * 1024), 16, wvoff, s * (BN * ROWB) + blk * 1024, 0, 0);
    }
  };
  auto store_a = [&](const u32x4* src, int slot) {
    char* base = lds + slot * STAGE + ldst0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const f32x4 v = __builtin_bit_cast(f32x4, src[i]);
      const f32x2 t0 = {v[0] * a_scale, v[1] * a_scale}, t1 = {v[2] * a_scale, v[3] * a_scale};
      const h2 h0 = __builtin_convertvector(t0, h2), h1 = __builtin_convertvector(t1, h2);
      const h4 hi = {h0[0], h0[1], h1[0], h1[1]};
      *(h4*)(base + 64 * i * ROWB) = hi;
    }
  };

  f32x4 acc[8][NTN];
#pragma unroll
  for (int a = 0; a < 8; ++a)
#pragma unroll
    for (int b = 0; b < NTN; ++b) acc[a][b] = (f32x4)(0.f);

  const int lr = lane & 15, kg = lane >> 4;              // row inside a 16-row block, 8-channel group of the K step
  const int frag = lr * ROWB + ((kg ^ swz(lr)) << 4);    // (blocks start on multiples of 16 rows: the key is the lane's)
  const int a_lane = wm * 128 * ROWB + frag;
  const int b_lane = BM * ROWB + wn * 32 * NB * ROWB + frag;
  auto read_a = [&](h8* dst, int slot, int half) {
    const char* base = lds + slot * STAGE + a_lane + half * 4 * 16 * ROWB;
#pragma unroll
    for (int t = 0; t < 4; ++t) dst[t] = *(const h8*)(base + t * 16 * ROWB);
  };
  auto read_b = [&](h8* dst, int slot) {
    const char* base = lds + slot * STAGE + b_lane;
#pragma unroll
    for (int t = 0; t < NTN; ++t) dst[t] = *(const h8*)(base + t * 16 * ROWB);
  };
  auto mfma_half = [&](const h8* a, const h8* b, int half) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int tn = 0; tn < NTN; ++tn)
        acc[half * 4 + t][tn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[tn], a[t], acc[half * 4 + t][tn], 0, 0, 0);
  };

  h8 a0[4], a1[4], bf[2][NTN];
  if constexpr (F16IN) {
    // ---- prologue: steps 0, 1 and 2 requested; 0 and 1 have landed ----
    dma_a(0, true);
    dma_b(0, 0);
    dma_a(1, nsteps > 1);
    dma_b(1, 1);
    dma_a(2, nsteps > 2);
    dma_b(2, 2);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NVM) : "memory");
    __builtin_amdgcn_s_barrier();
    read_a(a0, 0, 0);
    read_b(bf[0], 0);
    auto step_body = [&](int step, auto pc) {
      constexpr int P = decltype(pc)::value;
      const int cur = step & 3, nxt = (step + 1) & 3;
      read_a(a1, cur, 1);
      mfma_half(a0, bf[P], 0);
      dma_a((step + 3) & 3, step + 3 < nsteps);          // slot last read in step - 1: every wave has passed the barrier since
      dma_b((step + 3) & 3, step + 3);
      read_a(a0, nxt, 0);                                // slot nxt is complete since the barrier in front of this step
      read_b(bf[P ^ 1], nxt);
      mfma_half(a1, bf[P], 1);
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NVM) : "memory");      // step + 2 has landed
      __builtin_amdgcn_s_barrier();
    };
    for (int step = 0; step < nsteps; step += 2) {
      step_body(step, std::integral_constant<int, 0>{});
      step_body(step + 1, std::integral_constant<int, 1>{});
    }
  } else {
  // ---- prologue: steps 0 and 1 into slots 0 and 1, step 2 requested ----
  load_a(ra[0], true);
  dma_b(0, 0);
  load_a(ra[1], nsteps > 1);
  dma_b(1, 1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  store_a(ra[0], 0);
  store_a(ra[1], 1);
  load_a(ra[0], nsteps > 2);
  dma_b(2, 2);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  read_a(a0, 0, 0);
  read_b(bf[0], 0);

  // one step; PC = its parity (register sets)
  auto step_body = [&](int step, auto pc) {
    constexpr int P = decltype(pc)::value;
    const int cur = step & 3, nxt = (step + 1) & 3;
    read_a(a1, cur, 1);
    mfma_half(a0, bf[P], 0);
    load_a(ra[P ^ 1], step + 3 < nsteps);                // activations of step + 3 (written at the end of step + 1)
    dma_b((step + 3) & 3, step + 3);                     // that slot was last read in step - 1: every wave has passed the barrier since
    read_a(a0, nxt, 0);                                  // slot nxt is complete since the barrier in front of this step
    read_b(bf[P ^ 1], nxt);
    mfma_half(a1, bf[P], 1);
    // everything requested before this step has arrived; the registers go THROUGH the statement so that the conversion stays behind it
    // (hipcc otherwise converts at the top of the step and waits there: one step of cover instead of two)
    asm volatile("s_waitcnt vmcnt(%4)" : "+v"(ra[P][0]), "+v"(ra[P][1]), "+v"(ra[P][2]), "+v"(ra[P][3]) : "n"(NVM) : "memory");
    store_a(ra[P], (step + 2) & 3);
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NVM) : "memory");
    __builtin_amdgcn_s_barrier();
  };
  for (int step = 0; step < nsteps; step += 2) {
    step_body(step, std::integral_constant<int, 0>{});
    step_body(step + 1, std::integral_constant<int, 1>{});
  }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // (requests past the end: nothing may land in LDS after the workgroup has gone)

  // ---- epilogue: transposed product, lane = pixel lr of the block, channels n = 16 * blk + 4 * kg + e (register e) ----
  const float slope = p.act == EGNE_ACT_RELU ? 0.f : (p.act == EGNE_ACT_LEAKY ? 0.01f : 1.f);
  const long long left = M - m0;
  // out_split = 2: the output is stored as f16 halves of v * out_split_scale in channel order (the F16IN form's input format)
  const bool o16 = p.out_split == 2;
  const int oesz = o16 ? 2 : 4;
  const __amdgpu_buffer_rsrc_t rout = make_rsrc((char*)p.out + m0 * p.out_pix_stride * oesz, (unsigned)((left < BM ? left : BM) * p.out_pix_stride * oesz));
  bool bad = false;
#pragma unroll
  for (int tn = 0; tn < NTN; ++tn) {
    const int n = ntile * BN + wn * 32 * NB + tn * 16 + 4 * kg;
    const bool nok = n < p.Cout_store;
    const f32x4 bv = (p.bias && nok) ? *(const f32x4*)(p.bias + n) : (f32x4)(0.f);
#pragma unroll
    for (int tm = 0; tm < 8; ++tm) {
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float t = acc[tm][tn][e] * out_scale + bv[e];
        v[e] = fmaxf(t, t * slope);
      }
      if (tn == 0) bad |= egne_nonfinite(v[0]);          // lane = pixel: one channel per pixel (common.h)
      const int row = wm * 128 + tm * 16 + lr;
      if (o16) {
        const f32x2 u0 = {v[0] * p.out_split_scale, v[1] * p.out_split_scale}, u1 = {v[2] * p.out_split_scale, v[3] * p.out_split_scale};
        const h2 h0 = __builtin_convertvector(u0, h2), h1 = __builtin_convertvector(u1, h2);
        if (tn == 0) bad |= egne_nonfinite((float)h0[0]);
        const u32x2 two = {__builtin_bit_cast(unsigned, h0), __builtin_bit_cast(unsigned, h1)};
        __builtin_amdgcn_raw_buffer_store_b64(two, rout, nok ? (row * (int)p.out_pix_stride + p.out_ch_off + n) * 2 : (int)OOB, 0, 0);
      } else {
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rout,
                                               nok ? (row * (int)p.out_pix_stride + p.out_ch_off + n) * 4 : (int)OOB, 0, 0);
      }
    }
  }
  egne_ovf_commit(bad, p.ovf_flag);
}

template <int NB>
__global__ __launch_bounds__(512) void conv_f16_big1_kernel(const egne_conv_desc p, const char* __restrict__ wimg, float a_scale, float out_scale) {
  conv_f16_big1_body<NB, false>(p, wimg, a_scale, out_scale);
}
template <int NB>
__global__ __launch_bounds__(512) void conv_f16_big1_h_kernel(const egne_conv_desc p, const char* __restrict__ wimg, float a_scale, float out_scale) {
  conv_f16_big1_body<NB, true>(p, wimg, a_scale, out_scale);
}

// OIHW fp32 -> LDS images [ntile][step = chunk*T + tap][BN rows][64 B]: row j = output channel ntile*BN + j, its 16-byte chunk c
// holds channels chunk*32 + 8 (c ^ key(j)) .. +7 as f16(w * wscale)
__global__ void pack_weight_f16img1_k(const float* __restrict__ w, int Cout, int Cin, int T, int BN, int ntiles, int nchunk,
                                      float wscale, _Float16* __restrict__ out) {
  const long long total = (long long)ntiles * nchunk * T * BN * 32;     // halfs
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int e = (int)(i & 7), pc = (int)((i >> 3) & 3);
    long long q = i >> 5;
    const int j = (int)(q % BN); q /= BN;
    const int step = (int)(q % (nchunk * T));
    const int nt = (int)(q / (nchunk * T));
    const int chunk = step / T, tap = step - chunk * T;
    const int g = pc ^ swz(j);
    const int n = nt * BN + j, ci = chunk * 32 + 8 * g + e;
    out[i] = (_Float16)((n < Cout && ci < Cin) ? w[((long long)n * Cin + ci) * T + tap] * wscale : 0.f);
  }
}

}  // namespace

extern "C" int egne_pack_conv_weight_f16img1(const float* w_oihw, int Cout, int Cin, int kh, int kw, int BN, int Ktot, float wscale,
                                             void* wimg, void* stream) {
  EGNE_REQUIRE(w_oihw && wimg && Cout > 0 && Cin > 0 && (BN == 128 || BN == 256) && Ktot >= Cin && Ktot % 32 == 0 && wscale > 0.f,
               "pack_f16img1: bad sizes Cout %d Cin %d BN %d Ktot %d", Cout, Cin, BN, Ktot);
  const int ntiles = (Cout + BN - 1) / BN, nchunk = Ktot / 32;
  long long total = (long long)ntiles * nchunk * kh * kw * BN * 32, g = (total + 255) / 256;
  if (g > 8192) g = 8192;
  hipLaunchKernelGGL(pack_weight_f16img1_k, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, w_oihw, Cout, Cin, kh * kw, BN, ntiles,
                     nchunk, wscale, (_Float16*)wimg);
  return egne::check_launch("egne_pack_conv_weight_f16img1");
}

extern "C" int egne_conv2d_f16_big1_supported(const egne_conv_desc* dp) {
  if (!dp) return 0;
  const egne_conv_desc& d = *dp;
  return d.f16_products == 1 && d.nseg == 1 && d.seg[0].Cp % 32 == 0 && ((d.seg[0].Cp / 32) * d.kh * d.kw) % 2 == 0 && d.kh * d.kw <= 32;
}

// Same descriptor as egne_conv2d_f16x3_big_fwd with d->f16_products = 1 and an even number of K steps (Cp / 32 * kh * kw); `wimg` from
// egne_pack_conv_weight_f16img1.
extern "C" int egne_conv2d_f16_big1_fwd(const egne_conv_desc* dp, const void* wimg, float a_scale, float w_scale, void* stream) {
  EGNE_REQUIRE(dp && wimg, "conv_f16_big1: null pointer");
  const egne_conv_desc& d = *dp;
  EGNE_REQUIRE(d.f16_products == 1 && d.nseg == 1 && d.ngroups == 1 && d.stride == 1 && d.pad_mode == 0 && !d.seg[0].scale && !d.seg[0].shift &&
               !d.residual && !d.post_scale && d.kh * d.kw <= 32, "conv_f16_big1: unsupported descriptor");
  const egne_seg& g = d.seg[0];
  EGNE_REQUIRE(g.ptr && g.Cp % 32 == 0 && g.Cp == d.Ktot && ((uintptr_t)g.ptr & 15) == 0 && g.ch_off % 4 == 0 && g.pix_stride % 4 == 0 &&
               ((g.Cp / 32) * d.kh * d.kw) % 2 == 0, "conv_f16_big1: input slice (Cp %d Ktot %d, an even number of K steps)", g.Cp, d.Ktot);
  const bool in16 = g.presplit == 2;
  EGNE_REQUIRE(g.presplit == 0 || (in16 && g.ch_off % 8 == 0 && g.pix_stride % 8 == 0), "conv_f16_big1: an f16 input slice (presplit = 2) starts on multiples of 8 halfs");
  EGNE_REQUIRE(d.out_split == 0 || (d.out_split == 2 && d.out_split_scale > 0.f), "conv_f16_big1: out_split is 0 or 2 (f16 output) with a positive scale");
  EGNE_REQUIRE(d.CoutP % 128 == 0 && d.Cout_store <= d.CoutP && d.Cout_store % 4 == 0 && d.out && ((uintptr_t)d.out & 15) == 0 &&
               d.out_ch_off % 4 == 0 && d.out_pix_stride % 4 == 0 && d.out_ch_off + d.Cout_store <= d.out_pix_stride &&
               d.out_pix_stride * 1024 < (1ll << 31) && (!d.bias || ((uintptr_t)d.bias & 15) == 0), "conv_f16_big1: output");
  const int dd = d.dil[0];
  EGNE_REQUIRE(dd >= 1 && d.H + 2 * d.pad_h * dd - dd * (d.kh - 1) == d.Ho && d.W + 2 * d.pad_w * dd - dd * (d.kw - 1) == d.Wo,
               "conv_f16_big1: output %dx%d inconsistent with geometry", d.Ho, d.Wo);
  EGNE_REQUIRE(((uintptr_t)wimg & 15) == 0 && a_scale > 0.f && w_scale > 0.f, "conv_f16_big1: weights / scales");
  const long long M = (long long)d.B * d.Ho * d.Wo;
  const float os = 1.0f / (a_scale * w_scale);
  hipStream_t st = (hipStream_t)stream;
#define EGNE_BIG1_GO(KERN, BNV)                                                                                                          \
  do {                                                                                                                                  \
    if (!egne::raise_lds((const void*)KERN, NSTG * (BM + BNV) * ROWB))                                                                  \
      return egne::fail(EGNE_ERR_LAUNCH, "conv_f16_big1: cannot raise the dynamic LDS limit");                                          \
    hipLaunchKernelGGL(KERN, dim3((unsigned)((M + BM - 1) / BM), (unsigned)(d.CoutP / BNV)), dim3(512), NSTG * (BM + BNV) * ROWB, st,   \
                       d, (const char*)wimg, a_scale, os);                                                                              \
  } while (0)
  if (d.CoutP % 256 == 0) { if (in16) EGNE_BIG1_GO(conv_f16_big1_h_kernel<2>, 256); else EGNE_BIG1_GO(conv_f16_big1_kernel<2>, 256); }
  else { if (in16) EGNE_BIG1_GO(conv_f16_big1_h_kernel<1>, 128); else EGNE_BIG1_GO(conv_f16_big1_kernel<1>, 128); }
#undef EGNE_BIG1_GO
  return egne::check_launch("egne_conv2d_f16_big1_fwd");
}
