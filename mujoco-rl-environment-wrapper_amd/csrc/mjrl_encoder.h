// The encoder of the reference's vision autoencoder (vision/autoencoder.py:12-18) behind the agent cameras:
//   Conv2D(32, 3x3, relu, stride 2, same) -> Conv2D(64, 3x3, relu, stride 2, same) -> Flatten -> Dense(latent, relu)
// on 64x64x3 images scaled to [0, 1] (vision/train.py:26), NHWC, TensorFlow "same" padding (stride 2 on an even size
// pads one row / column at the END only).  It is the one dense contraction next to the step path, so it runs on the
// matrix cores: v_mfma_f32_16x16x32_bf16, bf16 operands, fp32 accumulation, activations rounded to bf16 between the
// layers.  The 1/255 input scale is folded into the first layer's weights, so the pixels enter as exact bf16 integers.
//
//   mjrl_encoder_conv_kernel   one workgroup (4 waves) per image: conv1 as an implicit GEMM [32 ch] x [27 -> 32] x
//                              [1024 px] from the uint8 image staged in LDS, its output (32x32x32 bf16, 64 KB) kept in
//                              LDS; conv2 as 9 taps x ([64 ch] x [32] x [256 px]) reading 16-byte channel runs of that
//                              LDS image; output 16x16x64 bf16 in flatten order (h, w, c) to HBM.
//   mjrl_encoder_dense_kernel  [n_img] x [16384] x [latent]: 32 images x all latent tiles x one sixteenth of K per
//                              workgroup, the sixteenths of K joined by the last workgroup to finish; bias, relu, fp32 latents
//                              and, when asked, their scatter into the observation rows (float64).
// Weight fragments are packed on the host in the lane order of the MFMA operands (lane l holds W[n = l & 15]
// [k = 8 (l >> 4) + j], j = 0..7), so a lane's fragment is one 16-byte load.
#ifndef MJRL_ENCODER_H
#define MJRL_ENCODER_H

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace enc {

using frag_ab = __attribute__((ext_vector_type(8))) __bf16;
using frag_cd = __attribute__((ext_vector_type(4))) float;

enum { IMG = 64, C1 = 32, H1 = 32, C2 = 64, H2 = 16, FLAT = H2 * H2 * C2, K1 = 32 /* 27 padded */ };

__device__ __forceinline__ unsigned short bf16_bits(float f) {      // round to nearest even (inputs are finite)
  unsigned u = __float_as_uint(f);
  return (unsigned short)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}
__device__ __forceinline__ __bf16 bf16_of(float f) {
  unsigned short b = bf16_bits(f);
  __bf16 r;
  __builtin_memcpy(&r, &b, 2);
  return r;
}

// Round 3.  Both convolutions run with the roles of the MFMA operands swapped against round 2 -- the WEIGHTS are the A
// operand (16 output channels x 32 k), the PIXELS the B operand (32 k x 16 pixels) -- so that the result tile is
// D[channel][pixel] and a lane holds FOUR CONSECUTIVE CHANNELS of one pixel: 8 contiguous bytes of the NHWC activation
// tensor, one 8-byte store (round 2's D[pixel][channel] left a lane with 4 pixels of one channel: 2-byte scattered
// stores, 64 per lane in conv2's epilogue, and a transposition through LDS).  conv1's K axis is reordered so that a
// lane's 8 k-values are 8 CONTIGUOUS BYTES of the uint8 image (k = 8 ky + j, j = 0..7: byte j of the 9-byte run
// (kx, c) = (0,0)..(2,2) of image row 2 oy + ky starting at pixel 2 ox; k = 24 + ky: the run's ninth byte): one
// unaligned 8-byte LDS read and v_cvt_f32_ubyte / v_perm instead of 8 single-byte gathers with a division each.  The
// image sits in LDS with its rows padded by 8 zero bytes and one zero row behind it, which is TensorFlow's "same"
// padding for stride 2 on an even size (one row / column at the END).  bf16 roundings are v_cvt_pk_bf16_f32 (gfx950;
// round to nearest even, the same bits as the integer formula in bf16_bits).
enum { ROWB = IMG * 3 + 8, IMG_LDS = ((IMG + 1) * ROWB + 15) / 16 * 16 };
using pair_bf = __attribute__((ext_vector_type(2))) __bf16;
using pair_f = __attribute__((ext_vector_type(2))) float;
__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {
  pair_f v = {lo, hi};
  pair_bf r = __builtin_convertvector(v, pair_bf);
  unsigned u;
  __builtin_memcpy(&u, &r, 4);
  return u;
}
// two bytes (0..255) of a word as a pair of bf16: exact, the upper halves of their float images
__device__ __forceinline__ unsigned bytes_bf16(unsigned w, int lo_byte) {
  const float a = (float)((w >> (8 * lo_byte)) & 255u), b = (float)((w >> (8 * lo_byte + 8)) & 255u);
  return __builtin_amdgcn_perm(__float_as_uint(b), __float_as_uint(a), 0x07060302u);
}

// Where pixel p of conv1's output lives in LDS: neighbours swapped in every other pair.  conv2 reads pixels 2 col + kx
// across a row of 16 lanes -- all of one parity, i.e. all in the same 64-byte half of the 128-byte bank row, which
// halves the LDS rate of those reads; with the swap the lanes alternate between the halves.  (No room for padding: two
// workgroups share a CU's 160 KB only while the tile stays at 64 KB.)
__device__ __forceinline__ int a1_pixel(int p) { return p ^ ((p >> 1) & 1); }

// LDS: [0, IMG_LDS) the padded image (uint8), then conv1's output a1[pixel 0..1023][channel 0..31] (bf16, 64 KB)
// (CONV_WAVES = 8 waves per image: the kernel is a chain of dependent LDS / MFMA / conversion steps, not a stream of
// MFMAs -- 700 of them in 35 k cycles with four waves --, and the LDS tile limits a CU to two images at a time, so the
// only way to more waves in flight is more waves per image: half the tiles and half the rows each.)
enum { CONV_WAVES = 8, CONV_THREADS = 64 * CONV_WAVES, CONV_ROWS = H2 / CONV_WAVES };
// Both convolutions of one image from the padded uint8 image in LDS (all CONV_THREADS threads of the workgroup, after a
// barrier behind whoever filled the image: the staging loop of mjrl_encoder_conv_kernel).
__device__ __forceinline__ void conv_layers(const unsigned char* img, unsigned short* a1, int image, const frag_ab* __restrict__ w1p,
                                            const float* __restrict__ b1, const frag_ab* __restrict__ w2p,
                                            const float* __restrict__ b2, unsigned short* __restrict__ a2) {
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int col = lane & 15, grp = lane >> 4;
#ifndef MJRL_CONV_VARIANT
#define MJRL_CONV_VARIANT 0           // (experiments: 1 no conv2 taps, 2 no conv1 tiles, 3 no epilogue stores)
#endif
  // ---- conv1: tile = (output row oy, half of the row): 16 pixels x 32 channels, K = 27 in 32 slots
  const frag_ab wa = w1p[lane], wb = w1p[64 + lane];            // A operands: channels 0..15 / 16..31 of the (scaled) weights
  float bias1[2][4];
#pragma unroll
  for (int nt = 0; nt < 2; nt++)
#pragma unroll
    for (int r = 0; r < 4; r++) bias1[nt][r] = b1[16 * nt + 4 * grp + r];
  // (four tiles in flight: a tile is a chain LDS read -> byte conversion -> MFMA -> bias / relu / rounding -> LDS write of
  // some 500 cycles with two waves per SIMD to hide it; rolled up, conv1 -- 32 of the kernel's 176 MFMAs -- took 15 of its 38 us)
#pragma unroll 4
  for (int tile = wave; tile < (MJRL_CONV_VARIANT == 2 ? 0 : 2 * H1); tile += CONV_WAVES) {
    const int oy = tile >> 1, ox = (tile & 1) * 16 + col;
    // this lane's 8 k-values: bytes 0..7 of image row 2 oy + grp from pixel 2 ox on (grp < 3); the ninth bytes of the
    // three rows and five zeros (grp 3)
    // (the run starts at byte 6 ox of its row, 2-byte aligned: read as the two 8-byte-aligned words around it and
    // shifted into place -- an unaligned 8-byte LDS read stalls the LDS pipe for some 60 cycles, SQ_LDS_UNALIGNED_STALL)
    const int off = (2 * oy + (grp < 3 ? grp : 0)) * ROWB + 6 * ox, sh = off & 7;
    const uint2 q0 = *(const uint2*)(img + (off & ~7)), q1 = *(const uint2*)(img + (off & ~7) + 8);
    const unsigned s0 = sh & 4 ? q0.y : q0.x, s1 = sh & 4 ? q1.x : q0.y, s2 = sh & 4 ? q1.y : q1.x;
    unsigned lo = s0, hi = s1;
    if (sh & 2) { lo = __builtin_amdgcn_alignbyte(s1, s0, 2); hi = __builtin_amdgcn_alignbyte(s2, s1, 2); }
    const unsigned char* at = img + 2 * oy * ROWB + 6 * ox;
    const unsigned n0 = at[8], n1 = at[ROWB + 8], n2 = at[2 * ROWB + 8];
    unsigned f0 = bytes_bf16(lo, 0), f1 = bytes_bf16(lo, 2), f2 = bytes_bf16(hi, 0), f3 = bytes_bf16(hi, 2);
    if (grp == 3) {
      f0 = bytes_bf16(n0 | (n1 << 8), 0);
      f1 = bytes_bf16(n2, 0);
      f2 = f3 = 0u;
    }
    frag_ab px;
    {
      const unsigned w4[4] = {f0, f1, f2, f3};
      __builtin_memcpy(&px, w4, 16);
    }
    frag_cd acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, px, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb, px, acc1, 0, 0, 0);
    // D[channel 4 grp + r][pixel col]: four consecutive channels of the lane's pixel -> one 8-byte store per half
    unsigned short* dst = a1 + a1_pixel(oy * H1 + ox) * C1 + 4 * grp;
    const uint2 h0 = {pack_bf16(fmaxf(acc0[0] + bias1[0][0], 0.0f), fmaxf(acc0[1] + bias1[0][1], 0.0f)),
                      pack_bf16(fmaxf(acc0[2] + bias1[0][2], 0.0f), fmaxf(acc0[3] + bias1[0][3], 0.0f))};
    const uint2 h1 = {pack_bf16(fmaxf(acc1[0] + bias1[1][0], 0.0f), fmaxf(acc1[1] + bias1[1][1], 0.0f)),
                      pack_bf16(fmaxf(acc1[2] + bias1[1][2], 0.0f), fmaxf(acc1[3] + bias1[1][3], 0.0f))};
    *(uint2*)dst = h0;
    *(uint2*)(dst + 16) = h1;
  }
  __syncthreads();
  // ---- conv2: wave w owns CONV_ROWS output rows (a tile = one output row, 16 pixels) x 4 channel tiles
  frag_cd acc[CONV_ROWS][4];
#pragma unroll
  for (int rt = 0; rt < CONV_ROWS; rt++)
#pragma unroll
    for (int nt = 0; nt < 4; nt++) acc[rt][nt] = frag_cd{0, 0, 0, 0};
#pragma unroll
  for (int tap = 0; tap < (MJRL_CONV_VARIANT == 1 ? 0 : 9); tap++) {      // (unrolled: the next taps' weight fragments are in flight while this one multiplies)
    const int ky = tap / 3, kx = tap - 3 * ky;
    frag_ab w[4];
#pragma unroll
    for (int nt = 0; nt < 4; nt++) w[nt] = w2p[(tap * 4 + nt) * 64 + lane];
#pragma unroll
    for (int rt = 0; rt < CONV_ROWS; rt++) {
      const int oy = CONV_ROWS * wave + rt, iy = 2 * oy + ky, ix = 2 * col + kx;
      frag_ab px;
      if (iy < H1 && ix < H1) px = *(const frag_ab*)(a1 + a1_pixel(iy * H1 + ix) * C1 + 8 * grp);     // 8 channels, 16 bytes
      else {
#pragma unroll
        for (int j = 0; j < 8; j++) px[j] = bf16_of(0.0f);
      }
#pragma unroll
      for (int nt = 0; nt < 4; nt++) acc[rt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[nt], px, acc[rt][nt], 0, 0, 0);
    }
  }
  // Epilogue.  The flattened activation vector goes to HBM in the order that is cheap to write: 16-byte chunks of the
  // lane's 2 x 4 consecutive channels (tiles 2 p and 2 p + 1) of pixel (oy, col), chunk index ((oy * 2 + p) * 16 + col) *
  // 4 + grp -- a wave's store instruction writes 1 KB of consecutive bytes (the Keras order (h, w, c) made it 8-byte pieces
  // 128 bytes apart: 8 of the kernel's 38 us).  The dense layer's weight rows are permuted to match on the host
  // (a2_source, mjrl_encoder_load), so the latents are those of Flatten -> Dense.
  uint4* out = (uint4*)(a2 + (size_t)image * FLAT);
  float bias[4][4];
#pragma unroll
  for (int nt = 0; nt < 4; nt++)
#pragma unroll
    for (int r = 0; r < 4; r++) bias[nt][r] = b2[16 * nt + 4 * grp + r];
#pragma unroll
  for (int rt = 0; rt < CONV_ROWS; rt++) {
    const int oy = CONV_ROWS * wave + rt;
#pragma unroll
    for (int p = 0; p < 2; p++) {
      uint4 v;
      v.x = pack_bf16(fmaxf(acc[rt][2 * p][0] + bias[2 * p][0], 0.0f), fmaxf(acc[rt][2 * p][1] + bias[2 * p][1], 0.0f));
      v.y = pack_bf16(fmaxf(acc[rt][2 * p][2] + bias[2 * p][2], 0.0f), fmaxf(acc[rt][2 * p][3] + bias[2 * p][3], 0.0f));
      v.z = pack_bf16(fmaxf(acc[rt][2 * p + 1][0] + bias[2 * p + 1][0], 0.0f), fmaxf(acc[rt][2 * p + 1][1] + bias[2 * p + 1][1], 0.0f));
      v.w = pack_bf16(fmaxf(acc[rt][2 * p + 1][2] + bias[2 * p + 1][2], 0.0f), fmaxf(acc[rt][2 * p + 1][3] + bias[2 * p + 1][3], 0.0f));
      if (MJRL_CONV_VARIANT != 3 || v.x == 0x12345678u) out[((oy * 2 + p) * 16 + col) * 4 + grp] = v;
    }
  }
}
__global__ __launch_bounds__(CONV_THREADS) void mjrl_encoder_conv_kernel(const unsigned char* __restrict__ rgb, int n_img,
                                                                const frag_ab* __restrict__ w1p, const float* __restrict__ b1,
                                                                const frag_ab* __restrict__ w2p, const float* __restrict__ b2,
                                                                unsigned short* __restrict__ a2) {
  extern __shared__ unsigned char lds_raw[];
  unsigned char* img = lds_raw;
  unsigned short* a1 = (unsigned short*)(lds_raw + IMG_LDS);
  const int image = blockIdx.x;
  if (image >= n_img) return;
  const int tid = threadIdx.x;
  {  // stage the image: 64 rows of 192 bytes into rows of ROWB, 8 bytes per thread and pass; zero the padding
    const uint2* src = (const uint2*)(rgb + (size_t)image * IMG * IMG * 3);
    for (int i = tid; i < IMG * IMG * 3 / 8; i += CONV_THREADS) {
      const int r = i / 24, c = i - 24 * r;
      *(uint2*)(img + r * ROWB + 8 * c) = src[i];
    }
    if (tid < IMG) *(uint2*)(img + tid * ROWB + IMG * 3) = uint2{0u, 0u};
    if (tid < ROWB / 8) *(uint2*)(img + IMG * ROWB + 8 * tid) = uint2{0u, 0u};
  }
  __syncthreads();
  conv_layers(img, a1, image, w1p, b1, w2p, b2, a2);
}

// element e of the activation vector as the conv kernel writes it -> its index (h * 16 + w) * 64 + c in Keras' Flatten order
__host__ __device__ inline int a2_source(int e) {
  const int chunk = e >> 3, j = e & 7, grp = chunk & 3, col = (chunk >> 2) & 15, p = (chunk >> 6) & 1, oy = chunk >> 7;
  const int c = 16 * (2 * p + (j >> 2)) + 4 * grp + (j & 3);
  return (oy * H2 + col) * C2 + c;
}

// Dense layer [n_img] x [16384] x [latent], round 3.  grid (ceil(n_img / 32), DENSE_KSPLIT), 4 waves per workgroup: a
// workgroup multiplies 32 images (two activation fragments) by ALL latent tiles over one sixteenth of K -- every weight
// fragment it loads is used twice, every activation fragment n_tile times, and the activations are read exactly once
// (round 2: 16 images x 2 tiles per workgroup read the activations 4 times and the weights 64 times, 35 us).  Its four
// waves split that sixteenth of K again and add their tiles through LDS; the workgroup writes its partial tile to
// part[image][ksplit][column]; mjrl_encoder_dense_finish_kernel adds the partials in split order (the sum does not depend
// on who finishes when).  No memset, no floating-point atomics, no fences.
// (NT latent tiles per launch, at most DENSE_MAXT = 7 -- the LDS tile of the four waves' partial sums is 8 KB per latent
// tile --; a wider latent takes one launch per group of tiles, nt0 = the group's first tile.)
#ifndef MJRL_DENSE_KSPLIT
#define MJRL_DENSE_KSPLIT 16
#endif
#ifndef MJRL_DENSE_UNROLL
#define MJRL_DENSE_UNROLL 4
#endif
enum { DENSE_KSPLIT = MJRL_DENSE_KSPLIT, DENSE_WAVES = 4, DENSE_MAXT = 7 };
template <int NT>
__global__ __launch_bounds__(64 * DENSE_WAVES) void mjrl_encoder_dense_kernel(
    const unsigned short* __restrict__ a2, int n_img, const frag_ab* __restrict__ wdp, int n_tile, int nt0,
    float* __restrict__ part) {
  __shared__ float red[DENSE_WAVES][2 * NT][64][4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, col = lane & 15, grp = lane >> 4;
  const int img0 = blockIdx.x * 32, ks = blockIdx.y;
  const int ncol = n_tile * 16, c0 = nt0 * 16;       // columns of the partial buffer; first column of this launch's tiles
  constexpr int STEPS = FLAT / 32 / DENSE_KSPLIT / DENSE_WAVES;       // k-steps of 32 per wave: 8
  const int kk0 = (ks * DENSE_WAVES + wave) * STEPS;
  const frag_ab* arow[2];
#pragma unroll
  for (int u = 0; u < 2; u++) {
    const int im = img0 + 16 * u + col < n_img ? img0 + 16 * u + col : n_img - 1;    // (rows past the batch repeat the last image)
    arow[u] = (const frag_ab*)(a2 + (size_t)im * FLAT) + grp;                        // k = 32 kk + 8 grp + j
  }
  const frag_ab* bcol = wdp + (size_t)nt0 * 64 + lane;                               // fragment (kk, nt): wdp[(kk * n_tile + nt) * 64 + lane]
  frag_cd acc[2][NT];
#pragma unroll
  for (int u = 0; u < 2; u++)
#pragma unroll
    for (int t = 0; t < NT; t++) acc[u][t] = frag_cd{0, 0, 0, 0};
  // The fragments of DENSE_BLOCK k-steps are fetched together, ahead of the multiplications that use them: left to the
  // compiler, every k-step's nine loads were issued right before its MFMAs and the wave sat out a trip to L2 / HBM per
  // step with four waves per CU to cover it (SQ_WAIT_ANY 88 % of the wave cycles): 14.8 -> 13.5 us.  (The scheduling
  // fence keeps the loads of a block from being sunk back to their uses.)
  constexpr int DENSE_BLOCK = MJRL_DENSE_UNROLL;
  static_assert(STEPS % DENSE_BLOCK == 0, "k-steps per wave must be a multiple of the prefetch block");
  for (int kb = kk0; kb < kk0 + STEPS; kb += DENSE_BLOCK) {
    frag_ab a[DENSE_BLOCK][2], b[DENSE_BLOCK][NT];
#pragma unroll
    for (int i = 0; i < DENSE_BLOCK; i++) {
#pragma unroll
      for (int u = 0; u < 2; u++) a[i][u] = arow[u][(kb + i) * 4];
#pragma unroll
      for (int t = 0; t < NT; t++) b[i][t] = bcol[((size_t)(kb + i) * n_tile + t) * 64];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < DENSE_BLOCK; i++)
#pragma unroll
      for (int u = 0; u < 2; u++)
#pragma unroll
        for (int t = 0; t < NT; t++) acc[u][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][u], b[i][t], acc[u][t], 0, 0, 0);
  }
  // C layout: column (latent) = lane & 15, rows (images) = 4 (lane >> 4) + r
#pragma unroll
  for (int u = 0; u < 2; u++)
#pragma unroll
    for (int t = 0; t < NT; t++)
#pragma unroll
      for (int r = 0; r < 4; r++) red[wave][u * NT + t][lane][r] = acc[u][t][r];
  __syncthreads();
  // the workgroup's partial tile: wave w adds the four waves' tiles w, w + 4, ... and writes them out
  // part[image][K split][column]: an image's partials sit next to each other (planes per split put the sixteen reads of
  // a latent in the finish kernel on sixteen different pages)
  float* mine = part + (size_t)ks * ncol;
  for (int x = wave; x < 2 * NT; x += DENSE_WAVES) {
    const int u = x / NT, t = x - NT * u;
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int image = img0 + 16 * u + 4 * grp + r;
      float v = red[0][x][lane][r];
#pragma unroll
      for (int w = 1; w < DENSE_WAVES; w++) v += red[w][x][lane][r];
      if (image < n_img) mine[(size_t)image * DENSE_KSPLIT * ncol + c0 + 16 * t + col] = v;
    }
  }
}

// The second half of the dense layer: latent n of image i = bias + the sum of its DENSE_KSPLIT partials in split order,
// relu, the fp32 latents and, when asked, their scatter into the observation rows (float64).  A kernel of its own: the
// kernel boundary is what makes the partials of workgroups on other XCDs visible (a device-scope fence inside the first
// kernel -- the "last workgroup adds them up" scheme -- writes back the whole L2 of the XCD every time it is executed,
// 2 x 512 times per launch: 39 us against 22).
__global__ __launch_bounds__(256) void mjrl_encoder_dense_finish_kernel(const float* __restrict__ part, int n_img, int latent,
                                                                        int n_tile, const float* __restrict__ bd, int relu,
                                                                        float* __restrict__ out, double* __restrict__ obs,
                                                                        const int* __restrict__ img_obs_row) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n_img * latent) return;
  const int image = i / latent, n = i - image * latent, ncol = 16 * n_tile;
  float v = 0.0f;
#pragma unroll
  for (int s = 0; s < DENSE_KSPLIT; s++) v += part[((size_t)image * DENSE_KSPLIT + s) * ncol + n];
  v += bd[n];
  if (relu) v = fmaxf(v, 0.0f);
  if (out) out[(size_t)image * latent + n] = v;
  if (obs && img_obs_row) {
    const int at = img_obs_row[image];        // index of the first latent slot in the flat observation tensor, or -1
    if (at >= 0) obs[(size_t)at + n] = (double)v;
  }
}

}  // namespace enc

#endif
