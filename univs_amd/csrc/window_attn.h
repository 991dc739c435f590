// Shared by the window-attention kernels (window_attn.hip: exact-f32 MFMA; window_attn_f16.hip: fp16 operands).
#pragma once

namespace univs {

// Image mode: qkv / out are in TOKEN order [B, H*W, ...] and the kernel does the reference's
// pad -> roll(-shift) -> window_partition on the way in and window_reverse -> roll(+shift) -> crop on
// the way out (swin.py:252-284) by index arithmetic: window b = (image, wy, wx), position j = (py, px)
// maps to pixel ((wy*ws + py + shift) mod Hp, (wx*ws + px + shift) mod Wp); pixels beyond (H, W) are the
// zero padding, whose q/k/v are the qkv Linear's bias (Linear(0) = bias) and which are not written back.
struct WinImage {
  int H, W, ws, shift, nWx, Hp, Wp;
};

inline WinImage win_image(int H, int W, int ws, int shift) {
  WinImage wi;
  wi.H = H; wi.W = W; wi.ws = ws; wi.shift = shift;
  wi.Hp = (H + ws - 1) / ws * ws;
  wi.Wp = (W + ws - 1) / ws * ws;
  wi.nWx = wi.Wp / ws;
  return wi;
}

}  // namespace univs
