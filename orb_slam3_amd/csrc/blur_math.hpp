// Arithmetic of cv::GaussianBlur(7x7, sigma 2) in OpenCV's 8.8 fixed-point path, shared by the
// full-frame blur (k_image.hip: k_blur) and the per-keypoint patch blur (k_describe.hip: k_patch_describe):
//   h = sum k_i * src[x+i-3]  (u16, <= 65280),  dst = (sum k_j * h[y+j-3] + 32768) >> 16,  k = 18 34 48 56 48 34 18.
#pragma once
#include <hip/hip_runtime.h>

typedef unsigned short __attribute__((ext_vector_type(2))) u16x2;
__device__ __forceinline__ u16x2 as_u16x2(unsigned v) { return __builtin_bit_cast(u16x2, v); }
__device__ __forceinline__ unsigned as_u32(u16x2 v) { return __builtin_bit_cast(unsigned, v); }

// horizontal 7-tap of pixel P (0..3) of the middle dword of the 12-byte window {w0,w1,w2}: the pixel's first tap is byte
// 1 + P of the window, so the 7 taps are spread over two or three ALIGNED dwords; each takes one v_dot4_u32_u8 against
// the kernel (18 34 48 56 48 34 18) shifted to where its bytes fall -- no byte realignment.  <= 65280
// acc0: what the sum starts from (k_blur folds the vertical pass's rounding in here: the taps sum to 256, so 128 added
// to every row's h is the 32768 of the second pass, and h + 128 <= 65408 still fits 16 bits)
template <int P>
__device__ __forceinline__ unsigned hblur_px(const unsigned (&w)[3], unsigned acc0 = 0) {
  constexpr unsigned long long K = 18ull | (34ull << 8) | (48ull << 16) | (56ull << 24) | (48ull << 32) | (34ull << 40) |
                                   (18ull << 48);
  // byte j of the 12-byte window carries tap j - (1 + P): the kernel as a 96-bit constant shifted left by 1 + P bytes
  constexpr int SH = 8 * (1 + P);
  constexpr unsigned K0 = (unsigned)(K << SH), K1 = (unsigned)(SH == 32 ? K : (K >> (32 - SH))),
                     K2 = (unsigned)(SH == 32 ? (K >> 32) : (K >> (64 - SH)));
  unsigned acc = acc0;
  if (K0) acc = __builtin_amdgcn_udot4(w[0], K0, acc, false);
  acc = __builtin_amdgcn_udot4(w[1], K1, acc, false);
  if (K2) acc = __builtin_amdgcn_udot4(w[2], K2, acc, false);
  return acc;
}

// (mov_dpp with bound_ctrl: the lane without a source gets 0 and nothing has to be put into the destination first --
//  update_dpp(0, ...) costs a v_mov 0 per move)
__device__ __forceinline__ unsigned from_left(unsigned v) {   // lane i <- lane i - 1 (wave_shr:1)
  return (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0x138, 0xf, 0xf, true);
}
__device__ __forceinline__ unsigned from_right(unsigned v) {  // lane i <- lane i + 1 (wave_shl:1)
  return (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0x130, 0xf, 0xf, true);
}

// vertical 7-tap on ROW PAIRS: A..D are the pairs (row 2m, row 2m+1) in u16 halves that cover the 7 rows of an output row;
// `odd` = the output row's first tap is the second row of pair A.  (acc + 32768) >> 16 is byte 2 of the result (< 2^24)
__device__ __forceinline__ unsigned vblur_px(unsigned A, unsigned B, unsigned C, unsigned D, bool odd) {
  const u16x2 we0 = {18, 34}, we1 = {48, 56}, we2 = {48, 34}, we3 = {18, 0};   // even: rows 2m .. 2m+6
  const u16x2 wo0 = {0, 18}, wo1 = {34, 48}, wo2 = {56, 48}, wo3 = {34, 18};   // odd:  rows 2m+1 .. 2m+7
  unsigned acc = __builtin_amdgcn_udot2(as_u16x2(A), odd ? wo0 : we0, 32768u, false);
  acc = __builtin_amdgcn_udot2(as_u16x2(B), odd ? wo1 : we1, acc, false);
  acc = __builtin_amdgcn_udot2(as_u16x2(C), odd ? wo2 : we2, acc, false);
  acc = __builtin_amdgcn_udot2(as_u16x2(D), odd ? wo3 : we3, acc, false);
  return acc;
}
