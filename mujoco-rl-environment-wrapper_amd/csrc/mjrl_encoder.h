// The encoder of the reference's vision autoencoder (vision/autoencoder.py:12-18) behind the agent cameras:
//   Conv2D(32, 3x3, relu, stride 2, same) -> Conv2D(64, 3x3, relu, stride 2, same) -> Flatten -> Dense(latent, relu)
// on 64x64x3 images scaled to [0, 1] (vision/train.py:26), NHWC, TensorFlow "same" padding (stride 2 on an even size
// pads one row / column at the END only).  It is the one dense contraction next to the step path, so it runs on the
// matrix cores: v_mfma_f32_16x16x32_bf16, bf16 operands, fp32 accumulation, activations rounded to bf16 between the
// layers.  The 1/255 input scale is folded into the first layer's weights, so the pixels enter as exact bf16 integers.
//
//   mjrl_encoder_conv_kernel   one workgroup (4 waves) per image: conv1 as an implicit GEMM [1024 px] x [27 -> 32] x
//                              [32 ch] from the uint8 image staged in LDS, its output (32x32x32 bf16, 64 KB) kept in
//                              LDS; conv2 as 9 taps x ([256 px] x [32] x [64 ch]) reading 16-byte channel runs of that
//                              LDS image; output 16x16x64 bf16 in flatten order (h, w, c) to HBM.
//   mjrl_encoder_dense_kernel  [n_img] x [16384] x [latent]: a workgroup of 8 waves per 16 images x up to 128 latent
//                              columns, K split eight ways and summed through LDS; bias, relu, fp32 latents and, when
//                              asked, their scatter into the observation rows (float64).
// Weight fragments are packed on the host in the lane order of the MFMA operands (lane l holds B[k = 8 (l >> 4) + j]
// [col = l & 15], j = 0..7), so a lane's fragment is one 16-byte load.
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

// LDS: [0, 12288) the image (uint8), [12288, 12288 + 65536) conv1's output a1[pixel 0..1023][channel 0..31] (bf16)
__global__ __launch_bounds__(256) void mjrl_encoder_conv_kernel(const unsigned char* __restrict__ rgb, int n_img,
                                                                const frag_ab* __restrict__ w1p, const float* __restrict__ b1,
                                                                const frag_ab* __restrict__ w2p, const float* __restrict__ b2,
                                                                unsigned short* __restrict__ a2) {
  extern __shared__ unsigned char lds_raw[];
  unsigned char* img = lds_raw;
  unsigned short* a1 = (unsigned short*)(lds_raw + IMG * IMG * 3);
  const int image = blockIdx.x;
  if (image >= n_img) return;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int row = lane & 15, grp = lane >> 4;
  {  // stage the image: 12288 bytes, 16 bytes per thread and pass
    const uint4* src = (const uint4*)(rgb + (size_t)image * IMG * IMG * 3);
    uint4* dst = (uint4*)img;
    for (int i = tid; i < IMG * IMG * 3 / 16; i += 256) dst[i] = src[i];
  }
  __syncthreads();
  // ---- conv1: output pixel (oy, ox) = row tile (oy, half) x 16 columns of ox; k = (ky * 3 + kx) * 3 + c
  const frag_ab wa = w1p[lane], wb = w1p[64 + lane];            // the two channel tiles of the (scaled) weights
  const float bias1a = b1[row], bias1b = b1[16 + row];
  for (int tile = wave; tile < 2 * H1; tile += 4) {
    const int oy = tile >> 1, ox = (tile & 1) * 16 + row;
    frag_ab a;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const int k = 8 * grp + j, tap = k / 3, c = k - 3 * tap, ky = tap / 3, kx = tap - 3 * ky;
      const int iy = 2 * oy + ky, ix = 2 * ox + kx;
      const bool in = k < 27 && iy < IMG && ix < IMG;
      a[j] = bf16_of(in ? (float)img[(iy * IMG + ix) * 3 + c] : 0.0f);      // 0..255: exact in bf16
    }
    frag_cd acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, wa, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, wb, acc1, 0, 0, 0);
    // C layout: column (channel) = lane & 15, rows (pixels) = 4 (lane >> 4) + r
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int px = oy * H1 + (tile & 1) * 16 + 4 * grp + r;
      a1[px * C1 + row] = bf16_bits(fmaxf(acc0[r] + bias1a, 0.0f));
      a1[px * C1 + 16 + row] = bf16_bits(fmaxf(acc1[r] + bias1b, 0.0f));
    }
  }
  __syncthreads();
  // ---- conv2: wave w owns output rows 4w .. 4w+3 (a row tile = one output row, 16 pixels) x 4 channel tiles
  frag_cd acc[4][4];
#pragma unroll
  for (int rt = 0; rt < 4; rt++)
#pragma unroll
    for (int nt = 0; nt < 4; nt++) acc[rt][nt] = frag_cd{0, 0, 0, 0};
  for (int tap = 0; tap < 9; tap++) {
    const int ky = tap / 3, kx = tap - 3 * ky;
    frag_ab b[4];
#pragma unroll
    for (int nt = 0; nt < 4; nt++) b[nt] = w2p[(tap * 4 + nt) * 64 + lane];
#pragma unroll
    for (int rt = 0; rt < 4; rt++) {
      const int oy = 4 * wave + rt, iy = 2 * oy + ky, ix = 2 * row + kx;
      frag_ab a;
      if (iy < H1 && ix < H1) a = *(const frag_ab*)(a1 + (iy * H1 + ix) * C1 + 8 * grp);     // 8 channels, 16 bytes
      else {
#pragma unroll
        for (int j = 0; j < 8; j++) a[j] = bf16_of(0.0f);
      }
#pragma unroll
      for (int nt = 0; nt < 4; nt++) acc[rt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b[nt], acc[rt][nt], 0, 0, 0);
    }
  }
  // Epilogue through LDS: the accumulator layout (a lane holds 4 pixels of one channel) would make 64 two-byte stores per
  // lane; a1 is dead once every wave has left the tap loop, so the tile is laid out there in flatten order (h, w, c) and
  // goes to HBM as 16-byte stores.
  __syncthreads();
  unsigned short* tile = a1;
#pragma unroll
  for (int nt = 0; nt < 4; nt++) {
    const float bias = b2[16 * nt + row];
#pragma unroll
    for (int rt = 0; rt < 4; rt++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int oy = 4 * wave + rt, ox = 4 * grp + r;
        tile[(oy * H2 + ox) * C2 + 16 * nt + row] = bf16_bits(fmaxf(acc[rt][nt][r] + bias, 0.0f));
      }
  }
  __syncthreads();
  uint4* out = (uint4*)(a2 + (size_t)image * FLAT);
  const uint4* src = (const uint4*)tile;
  for (int i = tid; i < FLAT * 2 / 16; i += 256) out[i] = src[i];
}

// grid (ceil(n_img / 16), ceil(latent tiles / DENSE_TILES)), 8 waves per workgroup: a workgroup owns 16 images x up to
// DENSE_TILES x 16 latent columns, its waves split K = 16384 eight ways (64 k-steps each) and the partial tiles are
// summed through LDS in wave order.  An activation fragment is fetched once per k-step and multiplied into every latent
// tile of the group; the weight fragments of a k-step's tiles are adjacent in memory.  Measured on config 5 with the
// encoder (512 copies, latent 100 = 7 tiles): 1, 2, 4, 8 tiles per group give 0.291, 0.288, 0.324, 0.302 ms per step --
// fewer, fatter workgroups read the activations fewer times but leave CUs idle (64 image tiles is all there is); 2.
#ifndef MJRL_DENSE_TILES
#define MJRL_DENSE_TILES 2
#endif
enum { DENSE_WAVES = 8, DENSE_TILES = MJRL_DENSE_TILES };
__global__ __launch_bounds__(64 * DENSE_WAVES) void mjrl_encoder_dense_kernel(
    const unsigned short* __restrict__ a2, int n_img, const frag_ab* __restrict__ wdp, const float* __restrict__ bd,
    int latent, int n_tile, int relu, float* __restrict__ out, double* __restrict__ obs, const int* __restrict__ img_obs_row,
    int obs_dim) {
  __shared__ float part[DENSE_WAVES][DENSE_TILES][64][4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, row = lane & 15, grp = lane >> 4;
  const int img0 = blockIdx.x * 16, nt0 = blockIdx.y * DENSE_TILES;
  const int tiles = n_tile - nt0 < DENSE_TILES ? n_tile - nt0 : DENSE_TILES;      // latent tiles of this workgroup
  const int my_img = img0 + row < n_img ? img0 + row : n_img - 1;           // (rows past the batch repeat the last image)
  const frag_ab* arow = (const frag_ab*)(a2 + (size_t)my_img * FLAT) + grp;  // k = 32 kk + 8 grp + j
  const frag_ab* bcol = wdp + (size_t)nt0 * 64 + lane;                       // fragment (kk, nt): wdp[(kk * n_tile + nt) * 64 + lane]
  frag_cd acc[DENSE_TILES];
#pragma unroll
  for (int t = 0; t < DENSE_TILES; t++) acc[t] = frag_cd{0, 0, 0, 0};
  constexpr int STEPS = FLAT / 32 / DENSE_WAVES;
  const int k0 = wave * STEPS;
  // (no branch on `tiles` in the loop: a group with fewer than DENSE_TILES tiles multiplies its last tile again and drops
  // the result -- with conditional loads the compiler kept every fragment live on both paths and spilled)
  int toff[DENSE_TILES];
#pragma unroll
  for (int t = 0; t < DENSE_TILES; t++) toff[t] = (t < tiles ? t : tiles - 1) * 64;
  for (int kk = k0; kk < k0 + STEPS; kk += 2) {
    frag_ab a[2], b[2][DENSE_TILES];
#pragma unroll
    for (int u = 0; u < 2; u++) {
      a[u] = arow[(kk + u) * 4];
      const frag_ab* bk = bcol + (size_t)(kk + u) * n_tile * 64;
#pragma unroll
      for (int t = 0; t < DENSE_TILES; t++) b[u][t] = bk[toff[t]];
    }
#pragma unroll
    for (int u = 0; u < 2; u++)
#pragma unroll
      for (int t = 0; t < DENSE_TILES; t++) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[u], b[u][t], acc[t], 0, 0, 0);
  }
#pragma unroll
  for (int t = 0; t < DENSE_TILES; t++)
#pragma unroll
    for (int r = 0; r < 4; r++) part[wave][t][lane][r] = acc[t][r];
  __syncthreads();
  // the sums and the epilogue: wave w takes the latent tiles w, w + 8, ... of the group
  for (int t = wave; t < tiles; t += DENSE_WAVES) {
    const int n = 16 * (nt0 + t) + row;                                      // this lane's latent column
    if (n >= latent) continue;
    const float bias = bd[n];
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int image = img0 + 4 * grp + r;
      if (image >= n_img) continue;
      float v = part[0][t][lane][r];
#pragma unroll
      for (int w = 1; w < DENSE_WAVES; w++) v += part[w][t][lane][r];
      v += bias;
      if (relu) v = fmaxf(v, 0.0f);
      if (out) out[(size_t)image * latent + n] = v;
      if (obs && img_obs_row) {
        const int at = img_obs_row[image];        // index of the first latent slot in the flat observation tensor, or -1
        if (at >= 0) obs[(size_t)at + n] = (double)v;
      }
    }
  }
}

}  // namespace enc

#endif
