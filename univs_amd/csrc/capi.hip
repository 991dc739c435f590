// extern "C" entry points of libunivs_hip.so (declared in include/univs_hip.h).
// Argument validation + dispatch only; kernels live in the sibling .hip files.
#include <stdarg.h>
#include <string.h>

#include <mutex>

#include "common.h"
#include "config.h"

namespace univs {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// process-wide settings (include/univs_hip.h: UnivsConfig); a mutex-protected copy, handed out by value
static std::mutex g_cfg_mu;
static UnivsConfig g_cfg = {};
UnivsConfig config() {
  std::lock_guard<std::mutex> lock(g_cfg_mu);
  return g_cfg;
}
static thread_local int g_msda_last = 0;
static thread_local int g_msda_gen = 0;   // generation of the LDS-tiled kernel that ran last (0: none)

int msda_forward_generic_f32(const float*, const LevelTable&, const float*, const float*, int, int,
                             int, int, int, int, int, float*, hipStream_t);
int msda_forward_generic_f64(const double*, const LevelTable&, const double*, const double*, int,
                             int, int, int, int, int, int, double*, hipStream_t);
// returns 1 if the tiled kernel was launched, 0 if its preconditions do not hold, <0 on error
int msda_forward_tiled2_f32(const float*, const LevelTable&, const float*, const float*, int, int, int, int, int, int,
                            int, float*, hipStream_t);
int mask_decode_f32(const float*, const float*, int, int, int, long long, float*, hipStream_t);
int transpose_f32(const float*, float*, long long, int, int, long long, long long, const float*, const float*, float*, hipStream_t);
int linear_split_f32(const float*, const float*, const float*, const float*, float*, long long, int, int, int, hipStream_t, int = 0, int = 0,
                     const float* winv = nullptr);
int msda_forward_heads_f32(const float*, const LevelTable&, const float*, const float*, long long, int, int, int, int, int, int, int,
                           float*, hipStream_t);
int msda_forward_strips_f32(const float*, const LevelTable&, const float*, const float*, long long, int, int, int, int, int,
                            int, int, float*, hipStream_t);
int mask_decode_last_impl();
int mask_decode_attn_f32(const float*, const float*, int, int, int, long long, uint8_t*, unsigned*, unsigned,
                         hipStream_t);
int attn_mask_rows_reset(uint8_t*, const unsigned*, unsigned, long long, long long, hipStream_t);

int msda_backward_f32(const float*, const LevelTable&, const float*, const float*, const float*, int, int, int, int,
                      int, int, int, float*, float*, float*, hipStream_t);
int msda_prepare_f32(const float*, int, int, const float*, long long, const LevelTable&, int, int, int, int, int,
                     float*, float*, hipStream_t);
int bilinear_resample_f32(const float*, const float*, float*, long long, int, int, int, int, hipStream_t);
int upsample2x_add_f32(const float*, const float*, const float*, float*, long long, int, int, hipStream_t);
int normalize_pad_f32(const float*, float*, long long, int, int, int, int, int, const float*, const float*, hipStream_t);
int conv3x3_nhwc_f16x3_f32(const float*, const void*, const float*, float*, int, int, int, int, int, hipStream_t);
int small_chain_f32(const float*, int, const void* const*, const float* const*, const float* const*, const int*, const float*, const float*,
                    float, float*, float*, long long, int, hipStream_t);
int group_norm_affine_f32(const float*, const float*, const float*, int, int, long long, int, float, float*, long long, float*, hipStream_t);
int bilinear_pyramid3_f32(const float*, float*, float*, float*, long long, int, int, hipStream_t);
int layer_norm_f32(const float*, const float*, const float*, const float*, long long, int, float, float*, float*, const float*, float*,
                   long long, hipStream_t);
int patch_merge_norm_f32(const float*, const float*, const float*, int, int, int, int, float, float*, hipStream_t);
int group_norm_f32(const float*, const float*, const float*, int, int, long long, int, float, int, float*, long long,
                   float*, hipStream_t);
int masked_softmax_f32(float*, const unsigned char*, int, int, int, int, hipStream_t);
int proca_attention_f32(const float*, const float*, const float*, int, int, int, int, int, float, float*, hipStream_t);
int prompt_prefix_f32(const float*, const float*, int, int, int, int, int, float, float*, unsigned*, uint8_t*, int*, uint8_t*, int*, uint8_t*,
                      uint8_t*, hipStream_t);
int prompt_draw(const uint8_t*, const int*, const uint8_t*, const int*, const float*, const float*, const long long*, int, int, int, int, int,
                int, long long*, long long*, uint8_t*, float*, hipStream_t);
int prompt_point_pe_f32(const float*, const float*, const float*, const float*, float, int, int, int, float*, hipStream_t);
int token_mean_f32(const float*, const float*, int, int, int, int, float*, hipStream_t);
int mask_stats_f32(const float*, long long, int, long long, long long, int, int, int, int, float, float, float, int*, hipStream_t);
int prompt_tokens_f32(const float*, const long long*, const float*, const long long*, const float*, const float*, const long long*,
                      const uint8_t*, const uint8_t*, const float*, const long long*, int, int, int, int, int, int, int, float*, float*,
                      uint8_t*, hipStream_t);
int window_attention_image_f32(const float*, const float*, const float*, const float*, int, int, int, int, int, int,
                               int, float, float*, hipStream_t);
int presplit_f16x3(const float*, int, int, int, int, void*, float*, hipStream_t);
int decoder_memory_f32(const float*, const float*, const float*, const float*, float*, float*, int, int, int, hipStream_t);
int patch_embed4_f32(const float*, const float*, const float*, const float*, const float*, float, float*, int, int, int, int, hipStream_t);
int cross_attention_f32(const float*, const float*, const float*, const unsigned char*, const unsigned*, unsigned, int, int, int, int, int,
                        int, int, int, float, float*, float*, hipStream_t);
size_t cross_attention_workspace_floats(int, int, int, int);
int mlp_f16x3_f32(const float*, const void*, const float*, const float*, const void*, const float*, const float*, const float*,
                  const float*, const float*, float, const float*, const float*, float, const float*, long long, float*, float*, long long,
                  int, int, int, int, hipStream_t);
int linear_f16x3_stream_f32(const float*, const void*, const float*, const float*, const float*, float*, long long, int, int, int,
                            hipStream_t);
int linear_f16x3_tile_f32(const float*, const void*, const float*, const float*, const float*, float*, long long, int, int, int,
                          hipStream_t);
int small_linear_f32(const float*, const float*, const void*, const float*, const float*, int, int, const float*, const float*, const float*,
                     float, float*, long long, int, int, int, int, int, hipStream_t);
int conv3x3_f16x3_f32(const float*, const void*, const float*, float*, int, int, int, int, int, hipStream_t);
int conv1x1_f16x3_f32(const float*, const void*, const float*, const float*, float*, int, int, int, int, int, hipStream_t);
int window_attention_image_f16mma(const float*, const float*, const float*, const float*, int, int, int, int, int, int,
                                  int, float, int, float*, hipStream_t);
int window_attention_f32(const float*, const float*, const float*, int, int, int, int, int, float,
                         float*, hipStream_t);

static int make_levels(const int64_t* shapes, const int64_t* starts, int L, int S, LevelTable* lv,
                       const char* what) {
  if (L < 1 || L > UNIVS_MAX_LEVELS) {
    set_error("%s: num_levels=%d outside [1,%d]", what, L, UNIVS_MAX_LEVELS);
    return UNIVS_ERR_INVALID_ARGUMENT;
  }
  if (!shapes || !starts) {
    set_error("%s: spatial_shapes / level_start_index must be host pointers, got NULL", what);
    return UNIVS_ERR_INVALID_ARGUMENT;
  }
  // The table must describe the concatenation `value` holds (ms_deform_attn.py:95: the levels are flattened and
  // concatenated in order): start[l] is the running sum of H*W and the levels cover exactly S tokens.  A table that
  // merely fits inside S (e.g. the previous, smaller resolution's) would sample with the wrong geometry silently.
  int64_t run = 0;
  for (int l = 0; l < L; ++l) {
    const int64_t H = shapes[2 * l], W = shapes[2 * l + 1], st = starts[l];
    if (H <= 0 || W <= 0 || st != run || st + H * W > (int64_t)S) {
      set_error("%s: level %d (H=%lld, W=%lld, start=%lld) is inconsistent: expected start=%lld, value length S=%d",
                what, l, (long long)H, (long long)W, (long long)st, (long long)run, S);
      return UNIVS_ERR_INVALID_ARGUMENT;
    }
    run += H * W;
    lv->H[l] = (int)H;
    lv->W[l] = (int)W;
    lv->start[l] = (int)st;
  }
  if (run != (int64_t)S) {
    set_error("%s: the %d levels hold %lld tokens but value has S=%d", what, L, (long long)run, S);
    return UNIVS_ERR_INVALID_ARGUMENT;
  }
  for (int l = L; l < UNIVS_MAX_LEVELS; ++l) lv->H[l] = lv->W[l] = lv->start[l] = 0;
  return UNIVS_OK;
}

}  // namespace univs

using namespace univs;

extern "C" {

const char* univs_version(void) { return "univs_hip 0.1.0 gfx950"; }
const char* univs_last_error(void) { return g_err; }

int univs_configure(const UnivsConfig* cfg) {
  UnivsConfig c = {};
  if (cfg) {
    if (cfg->size < (int)(2 * sizeof(int)) || cfg->size > (int)sizeof(UnivsConfig)) {
      set_error("univs_configure: UnivsConfig.size=%d (this library: %d)", cfg->size, (int)sizeof(UnivsConfig));
      return UNIVS_ERR_INVALID_ARGUMENT;
    }
    memcpy(&c, cfg, (size_t)cfg->size);   // fields the caller does not know keep their defaults (0)
    if (c.msda_impl < 0 || c.msda_impl > 2 || c.mask_decode_impl < 0 || c.mask_decode_impl > 2 || c.msda_halo > 64 ||
        (c.mask_decode_ct != 0 && c.mask_decode_ct != 2 && c.mask_decode_ct != 4) ||
        (c.linear_terms != 0 && c.linear_terms != 3 && c.linear_terms != 6) || c.mask_decode_wave_tiles > 64) {
      set_error("univs_configure: msda_impl=%d mask_decode_impl=%d msda_halo=%d mask_decode_ct=%d linear_terms=%d out of range",
                c.msda_impl, c.mask_decode_impl, c.msda_halo, c.mask_decode_ct, c.linear_terms);
      return UNIVS_ERR_INVALID_ARGUMENT;
    }
  }
  c.size = (int)sizeof(UnivsConfig);
  std::lock_guard<std::mutex> lock(g_cfg_mu);
  g_cfg = c;
  return UNIVS_OK;
}

int univs_get_config(UnivsConfig* out) {
  if (!out) {
    set_error("univs_get_config: NULL");
    return UNIVS_ERR_INVALID_ARGUMENT;
  }
  *out = config();
  out->size = (int)sizeof(UnivsConfig);
  return UNIVS_OK;
}

int univs_msda_set_impl(int impl) {
  if (impl < 0 || impl > 2) {
    set_error("univs_msda_set_impl: impl=%d not in {0,1,2}", impl);
    return UNIVS_ERR_INVALID_ARGUMENT;
  }
  std::lock_guard<std::mutex> lock(g_cfg_mu);
  g_cfg.msda_impl = impl;
  return UNIVS_OK;
}

int univs_msda_last_impl(void) { return g_msda_last; }
int univs_msda_last_tiled_generation(void) { return g_msda_gen; }

int univs_linear_fused_f32(const float* x, const float* weight, const float* bias, const float* residual, long long M, int N,
                           int K, int act, float* y, void* stream) {
  clear_sticky_error();
  if (M < 0 || N < 0 || K < 1 || act < 0 || act > 2 || (act != 0 && residual)) {
    set_error("univs_linear_fused_f32: bad arguments M=%lld N=%d K=%d act=%d%s", M, N, K, act,
              (act != 0 && residual) ? " (an activation and a residual exclude each other)" : "");
    return UNIVS_ERR_INVALID_ARGUMENT;
  }
  if (M == 0 || N == 0) return UNIVS_OK;
  if (!x || !weight || !y) {
    set_error("univs_linear_fused_f32: NULL data pointer");
    return UNIVS_ERR_INVALID_ARGUMENT;
  }
  const int epi = residual ? 3 : act;
  const int rc = univs::linear_split_f32(x, weight, bias, residual, y, M, N, K, epi, static_cast<hipStream_t>(stream));
  if (rc == 1) return UNIVS_OK;
  if (rc == 0) {
    set_error("univs_linear_fused_f32: shape M=%lld N=%d K=%d (or alignment) is not covered", M, N, K);
    return UNIVS_ERR_NOT_IMPLEMENTED;
  }
  return rc;
}


int univs_presplit_weights_f32(const float* w, int N, int K, int conv, void* wp, float* winv, void* stream) {
  clear_sticky_error();
  if (N < 0 || K < 32 || K % 32 != 0 || conv < 0 || conv > 2 || (conv == 1 && K % 9 != 0)) {
    set_error("univs_presplit_weights_f32: bad arguments N=%d K=%d mode=%d (K a multiple of 32; mode 1: K = 9 Cin)", N, K, conv);
    return UNIVS_ERR_INVALID_ARGUMENT;
  }
  if (N == 0) return UNIVS_OK;
  if (!w || !wp || !winv || (reinterpret_cast<uintptr_t>(wp) & 15)) {
    set_error("univs_presplit_weights_f32: NULL or unaligned pointer");
    return UNIVS_ERR_INVALID_ARGUMENT;
  }
  return univs::presplit_f16x3(w, N, K, conv == 1 ? K / 9 : 0, conv == 2 ? 1 : 0, wp, winv, static_cast<hipStream_t>(stream));
}

int univs_conv1x1_presplit_f32(const float* x, const void* wp, const float* winv, const float* bias, int T, int Cin, int Cout, int H,
                               int W, float* y, void* stream) {
  clear_sticky_error();
  if (T < 0 || Cin < 1 || Cout < 0 || H < 0 || W < 0) {
    set_error("univs_conv1x1_presplit_f32: bad dimensions T=%d Cin=%d Cout=%d H=%d W=%d", T, Cin, Cout, H, W);
    return UNIVS_ERR_INVALID_ARGUMENT;
  }
  if (T == 0 || Cout == 0 || H == 0 || W == 0) return UNIVS_OK;
  if (!x || !wp || !winv || !y) {
    set_error("univs_conv1x1_presplit_f32: NULL data pointer");
    return UNIVS_ERR_INVALID_ARGUMENT;
  }
  const int rc = univs::conv1x1_f16x3_f32(x, wp, winv, bias, y, T, Cin, Cout, H, W, static_cast<hipStream_t>(stream));
  if (rc == UNIVS_ERR_NOT_IMPLEMENTED)
    set_error("univs_conv1x1_presplit_f32: T=%d Cin=%d Cout=%d H=%d W=%d not covered (Cin %% 96 or %% 128, Cout %% 16, >= 4096 pixels)", T,
              Cin, Cout, H, W);
  return rc;
}

long long univs_cross_attention_workspace(int L, int S, int N, int H) {
  if (L <= 0 || S <= 0 || N <= 0 || H <= 0) return 0;
  return (long long)univs::cross_attention_workspace_floats(L, S, N, H);
}

int univs_cross_attention_f32(const float* q, const float* k, const float* v, const uint8_t* mask, int L, int S, int N, int H, int head_dim,
                              int ldq, int ldk, int ldv, float scale, float* workspace, float* out, void* stream) {
  return univs_cross_attention_flagged_f32(q, k, v, mask, nullptr, 0u, L, S, N, H, head_dim, ldq, ldk, ldv, scale, workspace, out, stream);
}

int univs_cross_attention_flagged_f32(const float* q, const float* k, const float* v, const uint8_t* mask, const uint32_t* mask_row_flags,
                                      uint32_t mask_generation, int L, int S, int N, int H, int head_dim, int ldq, int ldk, int ldv,
                                      float scale, float* workspace, float* out, void* stream) {
  clear_sticky_error();
  if (L < 0 || S < 1 || N < 0 || H < 1 || head_dim < 1) {
    set_error("univs_cross_attention_f32: bad dimensions L=%d S=%d N=%d H=%d head_dim=%d", L, S, N, H, head_dim);
    return UNIVS_ERR_INVALID_ARGUMENT;
  }
  if (L == 0 || N == 0) return UNIVS_OK;
  if (!q || !k || !v || !workspace || !out || (mask_row_flags && !mask)) {
    set_error("univs_cross_attention_f32: NULL data pointer (row flags come with a mask)");
    return UNIVS_ERR_INVALID_ARGUMENT;
  }
  const int rc = univs::cross_attention_f32(q, k, v, mask, mask_row_flags, mask_generation, L, S, N, H, head_dim, ldq, ldk, ldv, scale,
                                            workspace, out, static_cast<hipStream_t>(stream));
  if (rc == UNIVS_ERR_NOT_IMPLEMENTED)
    set_error("univs_cross_attention_f32: L=%d S=%d N=%d H=%d head_dim=%d not covered (head_dim == 32, S >= 32, with a mask S %% 4 == 0, "
              "N * H <= 65535, 16-byte aligned pointers)", L, S, N, H, head_dim);
  return rc;
}

int univs_mlp_presplit_f32(const float* x, const void* w1p, const float* w1inv, const float* b1, const void* w2p, const float* w2inv,
                           const float* b2, const float* residual, const float* ln_weight, const float* ln_bias, float ln_eps,
                           const float* post_ln_weight, const float* post_ln_bias, float post_ln_eps, const float* post_add,
                           long long post_add_rows, float* y2, long long M, int C, int Hd, int act, float* y, void* stream) {
  return univs_mlp_presplit_v2_f32(x, w1p, w1inv, b1, w2p, w2inv, b2, residual, 0, ln_weight, ln_bias, ln_eps, post_ln_weight, post_ln_bias,
                                   post_ln_eps, post_add, post_add_rows, y2, M, C, Hd, act, y, stream);
}

int univs_mlp_presplit_v2_f32(const float* x, const void* w1p, const float* w1inv, const float* b1, const void* w2p, const float* w2inv,
                              const float* b2, const float* residual, int flags, const float* ln_weight,
                              const float* ln_bias, float ln_eps, const float* post_ln_weight, const float* post_ln_bias, float post_ln_eps,
                              const float* post_add, long long post_add_rows, float* y2, long long M, int C, int Hd, int act, float* y,
                              void* stream) {
  clear_sticky_error();
  if (M < 0 || C < 1 || Hd < 1 || (act != 1 && act != 2)) {
    set_error("univs_mlp_presplit_f32: bad arguments M=%lld C=%d Hd=%d act=%d (1 ReLU, 2 GELU)", M, C, Hd, act);
    return UNIVS_ERR_INVALID_ARGUMENT;
  }
  if (M == 0) return UNIVS_OK;
  if (!x || !w1p || !w1inv || !w2p || !w2inv || !y) {
    set_error("univs_mlp_presplit_f32: NULL data pointer");
    return UNIVS_ERR_INVALID_ARGUMENT;
  }
  if ((flags & ~3) || ((flags & 1) && (residual || !ln_weight || x == y)) || ((flags & 2) && (!post_ln_weight || !y2 || post_add || (flags & 1)))) {
    set_error("univs_mlp_presplit_f32: flags=%d: UNIVS_MLP_RESIDUAL_IS_NORMED_X needs ln_weight, no residual pointer and y distinct from x "
              "(the normalised rows are parked in y); UNIVS_MLP_DUAL_OUTPUT needs post_ln_weight and y2, no post_add, and excludes the other", flags);
    return UNIVS_ERR_INVALID_ARGUMENT;
  }
  const int rc = univs::mlp_f16x3_f32(x, w1p, w1inv, b1, w2p, w2inv, b2, residual, ln_weight, ln_bias, ln_eps, post_ln_weight, post_ln_bias,
                                      post_ln_eps, post_add, post_add_rows, y2, y, M, C, Hd, act, flags,
                                      static_cast<hipStream_t>(stream));
  if (rc == UNIVS_ERR_NOT_IMPLEMENTED)
    set_error("univs_mlp_presplit_f32: shape M=%lld C=%d Hd=%d (or alignment) is not covered (C in 96 / 128 / 192 / 256 / 384, Hd %% 32 == 0, "
              "M >= 2048)", M, C, Hd);
  return rc;
}

int univs_small_linear_presplit_f32(const float* x, const float* x_add, const void* wp, const float* winv, const float* bias, int n_w,
                                    int f_off, const float* residual, const float* ln_weight, const float* ln_bias, float ln_eps,
                                    long long M, int N, int K, int relu, int add_features, int out_T, float* y, void* stream) {
  clear_sticky_error();
  if (M < 0 || N < 0 || K < 1 || n_w < 1 || f_off < 0 || f_off + N > n_w) {
    set_error("univs_small_linear_presplit_f32: bad arguments M=%lld N=%d K=%d n_w=%d f_off=%d", M, N, K, n_w, f_off);
    return UNIVS_ERR_INVALID_ARGUMENT;
  }
  if (M == 0 || N == 0) return UNIVS_OK;
  if (!x || !wp || !winv || !y || (ln_bias && !ln_weight)) {
    set_error("univs_small_linear_presplit_f32: NULL data pointer");
    return UNIVS_ERR_INVALID_ARGUMENT;
  }
  const int rc = univs::small_linear_f32(x, x_add, wp, winv, bias, n_w, f_off, residual, ln_weight, ln_bias, ln_eps, y, M, N, K, relu,
                                         add_features, out_T, static_cast<hipStream_t>(stream));
  if (rc == UNIVS_ERR_NOT_IMPLEMENTED)
    set_error("univs_small_linear_presplit_f32: M=%lld N=%d K=%d not covered (K %% 32 == 0, N %% 16 == 0, f_off %% 4 == 0, with a LayerNorm "
              "N == 256, M <= 1 048 560, 16-byte aligned pointers)", M, N, K);
  return rc;
}

int univs_small_mlp_presplit_f32(const float* x, int stages, const void* const* wp, const float* const* winv, const float* const* bias,
                                 const int* relu, const float* in_ln_weight, const float* in_ln_bias, float in_ln_eps, float* x_normed,
                                 long long M, int out_T, float* y, void* stream) {
  clear_sticky_error();
  if (M < 0 || stages < 1 || stages > 3 || out_T < 0) {
    set_error("univs_small_mlp_presplit_f32: bad arguments M=%lld stages=%d out_T=%d", M, stages, out_T);
    return UNIVS_ERR_INVALID_ARGUMENT;
  }
  if (M == 0) return UNIVS_OK;
  if (!x || !y || !wp || !winv || !bias || !relu) {
    set_error("univs_small_mlp_presplit_f32: NULL pointer");
    return UNIVS_ERR_INVALID_ARGUMENT;
  }
  const int rc = univs::small_chain_f32(x, stages, wp, winv, bias, relu, in_ln_weight, in_ln_bias, in_ln_eps, x_normed, y, M, out_T,
                                        static_cast<hipStream_t>(stream));
  if (rc == UNIVS_ERR_NOT_IMPLEMENTED)
    set_error("univs_small_mlp_presplit_f32: M=%lld not covered (M <= 1 048 560, out_T | M, 16-byte aligned pointers, x_normed / bias only with a LayerNorm)", M);
  return rc;
}

int univs_linear_presplit_f32(const float* x, const void* wp, const float* winv, const float* bias, const float* residual,
                              long long M, int N, int K, int act, float* y, void* stream) {
  clear_sticky_error();
  if (M < 0 || N < 0 || K < 1 || act < 0 || act > 2 || (act != 0 && residual)) {
    set_error("univs_linear_presplit_f32: bad arguments M=%lld N=%d K=%d act=%d%s", M, N, K, act,
              (act != 0 && residual) ? " (an activation and a residual exclude each other)" : "");
    return UNIVS_ERR_INVALID_ARGUMENT;
  }
  if (M == 0 || N == 0) return UNIVS_OK;
  if (!x || !wp || !winv || !y) {
    set_error("univs_linear_presplit_f32: NULL data pointer");
    return UNIVS_ERR_INVALID_ARGUMENT;
  }
  // the two-dimensional tiling where it applies (gemm_f16x3_tile.hip; UnivsConfig.linear_ablate == 6 switches it off: A / B), else the
  // row-range x pass kernel -- bit-identical results
  int rc = UNIVS_ERR_NOT_IMPLEMENTED;
  if (config().linear_ablate != 6)
    rc = univs::linear_f16x3_tile_f32(x, wp, winv, bias, residual, y, M, N, K, residual ? 3 : act, static_cast<hipStream_t>(stream));
  if (rc == UNIVS_ERR_NOT_IMPLEMENTED)
    rc = univs::linear_f16x3_stream_f32(x, wp, winv, bias, residual, y, M, N, K, residual ? 3 : act, static_cast<hipStream_t>(stream));
  if (rc == UNIVS_ERR_NOT_IMPLEMENTED) set_error("univs_linear_presplit_f32: shape M=%lld N=%d K=%d (or alignment) is not covered", M, N, K);
  return rc;
}

int univs_linear_resident_presplit_f32(const float* x, const void* wp, const float* winv, const float* bias, const float* residual,
                                       long long M, int N, int K, int act, float* y, void* stream) {
  clear_sticky_error();
  if (M < 0 || N < 0 || K < 1 || act < 0 || act > 2 || (act != 0 && residual)) {
    set_error("univs_linear_resident_presplit_f32: bad arguments M=%lld N=%d K=%d act=%d%s", M, N, K, act,
              (act != 0 && residual) ? " (an activation and a residual exclude each other)" : "");
    return UNIVS_ERR_INVALID_ARGUMENT;
  }
  if (M == 0 || N == 0) return UNIVS_OK;
  if (!x || !wp || !winv || !y) {
    set_error("univs_linear_resident_presplit_f32: NULL data pointer");
    return UNIVS_ERR_INVALID_ARGUMENT;
  }
  const int rc = univs::linear_split_f32(x, static_cast<const float*>(wp), bias, residual, y, M, N, K, residual ? 3 : act,
                                         static_cast<hipStream_t>(stream), 0, 0, winv);
  if (rc == 1) return UNIVS_OK;
  if (rc == 0) {
    set_error("univs_linear_resident_presplit_f32: shape M=%lld N=%d K=%d (or alignment) is not covered", M, N, K);
    return UNIVS_ERR_NOT_IMPLEMENTED;
  }
  return rc;
}

int univs_linear_blocked_presplit_f32(const float* x, const void* wp, const float* winv, const float* bias, long long M, int N, int K,
                                      int rows_per_batch, int col_block, float* y, void* stream) {
  if (M < 0 || N < 1 || K < 1 || rows_per_batch < 1 || col_block < 4 || col_block % 4 != 0 || N % col_block != 0 ||
      (M % rows_per_batch) != 0) {
    set_error("univs_linear_blocked_presplit_f32: bad arguments M=%lld N=%d K=%d rows_per_batch=%d col_block=%d", M, N, K, rows_per_batch,
              col_block);
    return UNIVS_ERR_INVALID_ARGUMENT;
  }
  if (M == 0) return UNIVS_OK;
  clear_sticky_error();
  if (!x || !wp || !winv || !y) {
    set_error("univs_linear_blocked_presplit_f32: NULL data pointer");
    return UNIVS_ERR_INVALID_ARGUMENT;
  }
  const int rc = univs::linear_split_f32(x, static_cast<const float*>(wp), bias, nullptr, y, M, N, K, /*LS_EPI_BLOCKED=*/4,
                                         static_cast<hipStream_t>(stream), rows_per_batch, col_block, winv);
  if (rc > 0) return UNIVS_OK;
  if (rc == 0) {
    set_error("univs_linear_blocked_presplit_f32: shape M=%lld N=%d K=%d (or alignment) is not covered (K == 256, M >= 2048)", M, N, K);
    return UNIVS_ERR_NOT_IMPLEMENTED;
  }
  return rc;
}

int univs_conv3x3_presplit_f32(const float* x, const void* wp, const float* winv, int T, int Cin, int Cout, int H, int W,
                               float* y, void* stream) {
  clear_sticky_error();
  if (T < 0 || Cin < 1 || Cout < 0 || H < 0 || W < 0) {
    set_error("univs_conv3x3_presplit_f32: bad dimensions T=%d Cin=%d Cout=%d H=%d W=%d", T, Cin, Cout, H, W);
    return UNIVS_ERR_INVALID_ARGUMENT;
  }
  if (T == 0 || Cout == 0 || H == 0 || W == 0) return UNIVS_OK;
  if (!x || !wp || !winv || !y) {
    set_error("univs_conv3x3_presplit_f32: NULL data pointer");
    return UNIVS_ERR_INVALID_ARGUMENT;
  }
  const int rc = univs::conv3x3_f16x3_f32(x, wp, winv, y, T, Cin, Cout, H, W, static_cast<hipStream_t>(stream));
  if (rc == UNIVS_ERR_NOT_IMPLEMENTED)
    set_error("univs_conv3x3_presplit_f32: T=%d Cin=%d Cout=%d H=%d W=%d not covered (Cin %% 128, Cout %% 16, >= 4096 pixels)", T, Cin, Cout, H, W);
  return rc;
}

int univs_conv3x3_nhwc_presplit_f32(const float* x, const void* wp, const float* winv, int T, int Cin, int Cout, int H, int W,
                                    float* y, void* stream) {
  clear_sticky_error();
  if (T < 0 || Cin < 1 || Cout < 0 || H < 0 || W < 0) {
    set_error("univs_conv3x3_nhwc_presplit_f32: bad dimensions T=%d Cin=%d Cout=%d H=%d W=%d", T, Cin, Cout, H, W);
    return UNIVS_ERR_INVALID_ARGUMENT;
  }
  if (T == 0 || Cout == 0 || H == 0 || W == 0) return UNIVS_OK;
  if (!x || !wp || !winv || !y) {
    set_error("univs_conv3x3_nhwc_presplit_f32: NULL data pointer");
    return UNIVS_ERR_INVALID_ARGUMENT;
  }
  const int rc = univs::conv3x3_nhwc_f16x3_f32(x, wp, winv, y, T, Cin, Cout, H, W, static_cast<hipStream_t>(stream));
  if (rc == UNIVS_ERR_NOT_IMPLEMENTED)
    set_error("univs_conv3x3_nhwc_presplit_f32: T=%d Cin=%d Cout=%d H=%d W=%d not covered (Cin %% 128, Cout %% 16, >= 4096 pixels)", T, Cin, Cout, H, W);
  return rc;
}

int univs_patch_embed4_f32(const float* x, const float* weight, const float* bias, const float* ln_weight, const float* ln_bias, float ln_eps,
                           int T, int H, int W, int E, float* out, void* stream) {
  clear_sticky_error();
  if (T < 0 || H < 0 || W < 0 || E < 1) {
    set_error("univs_patch_embed4_f32: bad dimensions T=%d H=%d W=%d E=%d", T, H, W, E);
    return UNIVS_ERR_INVALID_ARGUMENT;
  }
  if (T == 0 || H == 0 || W == 0) return UNIVS_OK;
  if (!x || !weight || !out) {
    set_error("univs_patch_embed4_f32: NULL data pointer");
    return UNIVS_ERR_INVALID_ARGUMENT;
  }
  const int rc = univs::patch_embed4_f32(x, weight, bias, ln_weight, ln_bias, ln_eps, out, T, H, W, E, static_cast<hipStream_t>(stream));
  if (rc == UNIVS_ERR_NOT_IMPLEMENTED)
    set_error("univs_patch_embed4_f32: T=%d H=%d W=%d E=%d not covered (E in 96 / 128 / 192, H %% 4, W %% 4, 16-byte alignment)", T, H, W, E);
  return rc;
}

int univs_decoder_memory_f32(const float* x, const float* level_embed, const float* pos_yx, const float* pos_t, int T, int C, int HW,
                             float* memory, float* key, void* stream) {
  clear_sticky_error();
  if (T < 0 || C < 0 || HW < 0) {
    set_error("univs_decoder_memory_f32: bad dimensions T=%d C=%d HW=%d", T, C, HW);
    return UNIVS_ERR_INVALID_ARGUMENT;
  }
  if (T == 0 || C == 0 || HW == 0) return UNIVS_OK;
  if (!x || !level_embed || !pos_yx || !pos_t || !memory || !key) {
    set_error("univs_decoder_memory_f32: NULL data pointer");
    return UNIVS_ERR_INVALID_ARGUMENT;
  }
  const int rc = univs::decoder_memory_f32(x, level_embed, pos_yx, pos_t, memory, key, T, C, HW, static_cast<hipStream_t>(stream));
  if (rc == UNIVS_ERR_NOT_IMPLEMENTED) set_error("univs_decoder_memory_f32: T=%d C=%d HW=%d not covered (C %% 4, HW %% 4, 16-byte alignment)", T, C, HW);
  return rc;
}

int univs_transpose_f32(const float* x, long long B, int R, int C, float* out, void* stream) {
  return univs_transpose_strided_f32(x, B, R, C, 0, out, stream);
}

int univs_transpose_strided_f32(const float* x, long long B, int R, int C, long long in_batch_stride, float* out, void* stream) {
  return univs_transpose_ex_f32(x, B, R, C, in_batch_stride, nullptr, out, 0, nullptr, nullptr, stream);
}

int univs_transpose_ex_f32(const float* x, long long B, int R, int C, long long in_batch_stride, const float* row_affine, float* out,
                           long long out_batch_stride, const float* addend, float* out2, void* stream) {
  clear_sticky_error();
  if (B < 0 || R < 0 || C < 0 || in_batch_stride < 0 || (in_batch_stride != 0 && in_batch_stride < (long long)R * C) ||
      out_batch_stride < 0 || (out_batch_stride != 0 && out_batch_stride < (long long)R * C)) {
    set_error("univs_transpose_f32: bad dimensions B=%lld R=%d C=%d in_batch_stride=%lld out_batch_stride=%lld", B, R, C, in_batch_stride,
              out_batch_stride);
    return UNIVS_ERR_INVALID_ARGUMENT;
  }
  if (B == 0 || R == 0 || C == 0) return UNIVS_OK;
  if (!x || !out || (addend != nullptr) != (out2 != nullptr)) {
    set_error("univs_transpose_f32: NULL data pointer (addend and out2 come together)");
    return UNIVS_ERR_INVALID_ARGUMENT;
  }
  const int rc = univs::transpose_f32(x, out, B, R, C, in_batch_stride, out_batch_stride, row_affine, addend, out2,
                                      static_cast<hipStream_t>(stream));
  if (rc == UNIVS_ERR_NOT_IMPLEMENTED)
    set_error("univs_transpose_f32: R=%d C=%d B=%lld not covered (R %% 4, C %% 4, strides %% 4, B <= 65535, 16-byte alignment)", R, C, B);
  return rc;
}

int univs_mask_decode_set_impl(int impl) {
  clear_sticky_error();
  if (impl < 0 || impl > 2) {
    set_error("univs_mask_decode_set_impl: impl=%d not in {0,1,2}", impl);
    return UNIVS_ERR_INVALID_ARGUMENT;
  }
  {
    std::lock_guard<std::mutex> lock(g_cfg_mu);
    g_cfg.mask_decode_impl = impl;
  }
  return UNIVS_OK;
}

int univs_mask_decode_last_impl(void) { return univs::mask_decode_last_impl(); }

int univs_msda_forward_f32(const float* value, const int64_t* spatial_shapes,
                           const int64_t* level_start, const float* sampling_loc,
                           const float* attn_weight, int N, int S, int M, int D, int L, int Lq,
                           int P, float* out, void* stream) {
  if (N < 0 || S < 0 || M < 0 || D < 0 || Lq < 0 || P < 0) {
    set_error("univs_msda_forward_f32: negative dimension");
    return UNIVS_ERR_INVALID_ARGUMENT;
  }
  if ((long long)N * Lq * M * D == 0) return UNIVS_OK;  // empty output
  clear_sticky_error();
  if (!value || !sampling_loc || !attn_weight || !out) {
    set_error("univs_msda_forward_f32: NULL data pointer");
    return UNIVS_ERR_INVALID_ARGUMENT;
  }
  LevelTable lv;
  int rc = make_levels(spatial_shapes, level_start, L, S, &lv, "univs_msda_forward_f32");
  if (rc != UNIVS_OK) return rc;
  hipStream_t st = (hipStream_t)stream;
  g_msda_gen = 0;
  if (config().msda_impl != 1) {
    // the LDS-tiled kernel for the encoder geometry on the standard layouts (msda_tiled2.hip: D == 32, P == 4, 3 <= L <= 4,
    // Lq == S); returns 0 when its preconditions fail.  (The module path uses univs_msda_forward_strips_f32.)
    rc = msda_forward_tiled2_f32(value, lv, sampling_loc, attn_weight, N, S, M, D, L, Lq, P, out, st);
    if (rc != 0) {
      if (rc > 0) { g_msda_last = 2; g_msda_gen = 2; }
      return rc < 0 ? rc : UNIVS_OK;
    }
  }
  g_msda_last = 1;
  return msda_forward_generic_f32(value, lv, sampling_loc, attn_weight, N, S, M, D, L, Lq, P, out, st);
}

int univs_msda_forward_f64(const double* value, const int64_t* spatial_shapes,
                           const int64_t* level_start, const double* sampling_loc,
                           const double* attn_weight, int N, int S, int M, int D, int L, int Lq,
                           int P, double* out, void* stream) {
  if (N < 0 || S < 0 || M < 0 || D < 0 || Lq < 0 || P < 0) {
    set_error("univs_msda_forward_f64: negative dimension");
    return UNIVS_ERR_INVALID_ARGUMENT;
  }
  if ((long long)N * Lq * M * D == 0) return UNIVS_OK;
  clear_sticky_error();
  if (!value || !sampling_loc || !attn_weight || !out) {
    set_error("univs_msda_forward_f64: NULL data pointer");
    return UNIVS_ERR_INVALID_ARGUMENT;
  }
  LevelTable lv;
  int rc = make_levels(spatial_shapes, level_start, L, S, &lv, "univs_msda_forward_f64");
  if (rc != UNIVS_OK) return rc;
  return msda_forward_generic_f64(value, lv, sampling_loc, attn_weight, N, S, M, D, L, Lq, P, out,
                                  (hipStream_t)stream);
}

int univs_msda_backward_f32(const float* value, const int64_t* spatial_shapes, const int64_t* level_start,
                            const float* sampling_loc, const float* attn_weight, const float* grad_output, int N,
                            int S, int M, int D, int L, int Lq, int P, float* grad_value,
                            float* grad_sampling_loc, float* grad_attn_weight, void* stream) {
  if (N < 0 || S < 0 || M < 0 || D < 0 || Lq < 0 || P < 0 || L < 0) {
    set_error("univs_msda_backward_f32: negative dimension");
    return UNIVS_ERR_INVALID_ARGUMENT;
  }
  clear_sticky_error();
  if ((long long)N * S * M * D > 0 && !grad_value) {
    set_error("univs_msda_backward_f32: NULL grad_value");
    return UNIVS_ERR_INVALID_ARGUMENT;
  }
  const long long nsamp = (long long)N * Lq * M * L * P;
  if (nsamp > 0 && (!value || !sampling_loc || !attn_weight || !grad_output || !grad_sampling_loc || !grad_attn_weight)) {
    set_error("univs_msda_backward_f32: NULL data pointer");
    return UNIVS_ERR_INVALID_ARGUMENT;
  }
  LevelTable lv;
  int rc = make_levels(spatial_shapes, level_start, L, S, &lv, "univs_msda_backward_f32");
  if (rc != UNIVS_OK) return rc;
  if ((long long)N * S * M * D == 0) return UNIVS_OK;
  return msda_backward_f32(value, lv, sampling_loc, attn_weight, grad_output, N, S, M, D, L, Lq, P, grad_value,
                           grad_sampling_loc, grad_attn_weight, (hipStream_t)stream);
}

int univs_mask_decode_f32(const float* mask_embed, const float* mask_features, int T, int Q, int C,
                          int HW, float* out, void* stream) {
  if (T < 0 || Q < 0 || C <= 0 || HW < 0) {
    set_error("univs_mask_decode_f32: bad dimensions T=%d Q=%d C=%d HW=%d", T, Q, C, HW);
    return UNIVS_ERR_INVALID_ARGUMENT;
  }
  if ((long long)T * Q * HW == 0) return UNIVS_OK;
  clear_sticky_error();
  if (!mask_embed || !mask_features || !out) {
    set_error("univs_mask_decode_f32: NULL data pointer");
    return UNIVS_ERR_INVALID_ARGUMENT;
  }
  return mask_decode_f32(mask_embed, mask_features, T, Q, C, HW, out, (hipStream_t)stream);
}

int univs_mask_decode_attn_f32(const float* mask_embed, const float* feat_lowres, int T, int Q,
                               int C, int hw, uint8_t* attn_mask, uint32_t* row_any_ws,
                               void* stream) {
  if (T < 0 || Q < 0 || C <= 0 || hw < 0) {
    set_error("univs_mask_decode_attn_f32: bad dimensions T=%d Q=%d C=%d hw=%d", T, Q, C, hw);
    return UNIVS_ERR_INVALID_ARGUMENT;
  }
  if ((long long)T * Q * hw == 0) return UNIVS_OK;
  clear_sticky_error();
  if (!mask_embed || !feat_lowres || !attn_mask || !row_any_ws) {
    set_error("univs_mask_decode_attn_f32: NULL data pointer");
    return UNIVS_ERR_INVALID_ARGUMENT;
  }
  return mask_decode_attn_f32(mask_embed, feat_lowres, T, Q, C, hw, attn_mask, row_any_ws, 0u,
                              (hipStream_t)stream);
}

int univs_mask_decode_attn_deferred_f32(const float* mask_embed, const float* feat_lowres, int T, int Q, int C, int hw, uint8_t* attn_mask,
                                        uint32_t* row_flags, uint32_t generation, void* stream) {
  if (T < 0 || Q < 0 || C <= 0 || hw < 0 || generation == 0) {
    set_error("univs_mask_decode_attn_deferred_f32: bad arguments T=%d Q=%d C=%d hw=%d generation=%u (non-zero)", T, Q, C, hw, generation);
    return UNIVS_ERR_INVALID_ARGUMENT;
  }
  if ((long long)T * Q * hw == 0) return UNIVS_OK;
  clear_sticky_error();
  if (!mask_embed || !feat_lowres || !attn_mask || !row_flags) {
    set_error("univs_mask_decode_attn_deferred_f32: NULL data pointer");
    return UNIVS_ERR_INVALID_ARGUMENT;
  }
  return mask_decode_attn_f32(mask_embed, feat_lowres, T, Q, C, hw, attn_mask, row_flags, generation, (hipStream_t)stream);
}

int univs_attn_mask_rows_reset(uint8_t* attn_mask, const uint32_t* row_flags, uint32_t generation, long long rows, long long hw,
                               void* stream) {
  clear_sticky_error();
  if (rows < 0 || hw < 0 || rows > 0x7fffffffLL) {
    set_error("univs_attn_mask_rows_reset: bad dimensions rows=%lld hw=%lld", rows, hw);
    return UNIVS_ERR_INVALID_ARGUMENT;
  }
  if (rows * hw == 0) return UNIVS_OK;
  if (!attn_mask || !row_flags) {
    set_error("univs_attn_mask_rows_reset: NULL data pointer");
    return UNIVS_ERR_INVALID_ARGUMENT;
  }
  return attn_mask_rows_reset(attn_mask, row_flags, generation, rows, hw, (hipStream_t)stream);
}

int univs_window_attention_f32(const float* qkv, const float* bias, const float* shift_mask,
                               int B_, int nW, int Ntok, int nH, int hd, float scale, float* out,
                               void* stream) {
  if (B_ < 0 || Ntok <= 0 || nH <= 0 || hd <= 0 || (shift_mask && (nW <= 0 || B_ % nW != 0))) {
    set_error("univs_window_attention_f32: bad dimensions B_=%d nW=%d Ntok=%d nH=%d hd=%d", B_, nW,
              Ntok, nH, hd);
    return UNIVS_ERR_INVALID_ARGUMENT;
  }
  if (B_ == 0) return UNIVS_OK;
  clear_sticky_error();
  if (!qkv || !bias || !out) {
    set_error("univs_window_attention_f32: NULL data pointer");
    return UNIVS_ERR_INVALID_ARGUMENT;
  }
  return window_attention_f32(qkv, bias, shift_mask, B_, nW > 0 ? nW : 1, Ntok, nH, hd, scale, out,
                              (hipStream_t)stream);
}

int univs_bilinear_resample_f32(const float* in, const float* addend, float* out, long long planes, int Hin,
                                int Win, int Hout, int Wout, void* stream) {
  clear_sticky_error();
  if (planes < 0 || Hin < 1 || Win < 1 || Hout < 1 || Wout < 1) {
    set_error("univs_bilinear_resample_f32: bad dimensions planes=%lld in=%dx%d out=%dx%d", planes, Hin, Win, Hout, Wout);
    return UNIVS_ERR_INVALID_ARGUMENT;
  }
  if (planes == 0) return UNIVS_OK;
  if (!in || !out) {
    set_error("univs_bilinear_resample_f32: NULL data pointer");
    return UNIVS_ERR_INVALID_ARGUMENT;
  }
  return bilinear_resample_f32(in, addend, out, planes, Hin, Win, Hout, Wout, static_cast<hipStream_t>(stream));
}

int univs_normalize_pad_f32(const float* x, const float* mean, const float* std, long long T, int C, int H, int W, int Hp, int Wp, float* out,
                            void* stream) {
  clear_sticky_error();
  if (T < 0 || C < 1 || H < 1 || W < 1 || Hp < H || Wp < W) {
    set_error("univs_normalize_pad_f32: bad dimensions T=%lld C=%d %dx%d -> %dx%d", T, C, H, W, Hp, Wp);
    return UNIVS_ERR_INVALID_ARGUMENT;
  }
  if (T == 0) return UNIVS_OK;
  if (!x || !mean || !std || !out) {
    set_error("univs_normalize_pad_f32: NULL data pointer");
    return UNIVS_ERR_INVALID_ARGUMENT;
  }
  const int rc = normalize_pad_f32(x, out, T, C, H, W, Hp, Wp, mean, std, static_cast<hipStream_t>(stream));
  if (rc == UNIVS_ERR_NOT_IMPLEMENTED) set_error("univs_normalize_pad_f32: T * C = %lld planes not covered (<= 65535)", T * C);
  return rc;
}

int univs_upsample2x_add_f32(const float* in, const float* addend, const float* addend_affine, float* out, long long planes, int Hin,
                             int Win, void* stream) {
  clear_sticky_error();
  if (planes < 0 || Hin < 1 || Win < 1) {
    set_error("univs_upsample2x_add_f32: bad dimensions planes=%lld in=%dx%d", planes, Hin, Win);
    return UNIVS_ERR_INVALID_ARGUMENT;
  }
  if (planes == 0) return UNIVS_OK;
  if (!in || !addend || !out) {
    set_error("univs_upsample2x_add_f32: NULL data pointer");
    return UNIVS_ERR_INVALID_ARGUMENT;
  }
  const int rc = upsample2x_add_f32(in, addend, addend_affine, out, planes, Hin, Win, static_cast<hipStream_t>(stream));
  if (rc == UNIVS_ERR_NOT_IMPLEMENTED)
    set_error("univs_upsample2x_add_f32: %dx%d not covered (Win even; in 8-byte, addend / out 16-byte aligned)", Hin, Win);
  return rc;
}

int univs_group_norm_affine_f32(const float* x, const float* gamma, const float* beta, int N, int C, long long HW, int groups, float eps,
                                float* ws, long long ws_floats, float* affine, void* stream) {
  clear_sticky_error();
  if (N < 0 || C < 1 || HW < 0 || groups < 1 || C % groups != 0 || (long long)N * C > 0x7fffffffLL) {
    set_error("univs_group_norm_affine_f32: bad dimensions N=%d C=%d HW=%lld groups=%d", N, C, HW, groups);
    return UNIVS_ERR_INVALID_ARGUMENT;
  }
  if ((long long)N * HW == 0) return UNIVS_OK;
  if (!x || !gamma || !beta || !ws || !affine) {
    set_error("univs_group_norm_affine_f32: NULL data pointer");
    return UNIVS_ERR_INVALID_ARGUMENT;
  }
  return group_norm_affine_f32(x, gamma, beta, N, C, HW, groups, eps, ws, ws_floats, affine, static_cast<hipStream_t>(stream));
}

int univs_bilinear_pyramid3_f32(const float* in, long long planes, int H, int W, float* out2, float* out4, float* out8,
                                void* stream) {
  clear_sticky_error();
  if (planes < 0 || H < 8 || W < 8) {
    set_error("univs_bilinear_pyramid3_f32: bad dimensions planes=%lld in=%dx%d", planes, H, W);
    return UNIVS_ERR_INVALID_ARGUMENT;
  }
  if (planes == 0) return UNIVS_OK;
  if (!in || !out2 || !out4 || !out8) {
    set_error("univs_bilinear_pyramid3_f32: NULL data pointer");
    return UNIVS_ERR_INVALID_ARGUMENT;
  }
  const int rc = bilinear_pyramid3_f32(in, out2, out4, out8, planes, H, W, static_cast<hipStream_t>(stream));
  if (rc == UNIVS_ERR_NOT_IMPLEMENTED) set_error("univs_bilinear_pyramid3_f32: %dx%d is not a multiple of 8 (or unaligned pointers)", H, W);
  return rc;
}

int univs_layer_norm_add_f32(const float* x, const float* residual, const float* gamma, const float* beta, const float* addend,
                             long long addend_rows, long long rows, int C, float eps, float* sum_out, float* out, float* out2,
                             void* stream) {
  clear_sticky_error();
  if (rows < 0 || C < 1) {
    set_error("univs_layer_norm_f32: bad dimensions rows=%lld C=%d", rows, C);
    return UNIVS_ERR_INVALID_ARGUMENT;
  }
  if (rows == 0) return UNIVS_OK;
  if (!x || !gamma || !beta || !out || (sum_out && !residual) || ((addend != nullptr) != (out2 != nullptr)) ||
      (addend && (!residual || sum_out || addend_rows < 1 || rows % addend_rows != 0))) {
    set_error("univs_layer_norm_f32: NULL data pointer (sum_out needs a residual; addend / out2 come together, with a residual, "
              "without sum_out; rows must be a multiple of addend_rows)");
    return UNIVS_ERR_INVALID_ARGUMENT;
  }
  const int rc = layer_norm_f32(x, residual, gamma, beta, rows, C, eps, sum_out, out, addend, out2, addend ? addend_rows : 1,
                                static_cast<hipStream_t>(stream));
  if (rc == UNIVS_ERR_NOT_IMPLEMENTED) set_error("univs_layer_norm_f32: C=%d not supported (C %% 4 == 0, C <= 3072)", C);
  return rc;
}

int univs_layer_norm_f32(const float* x, const float* residual, const float* gamma, const float* beta,
                         long long rows, int C, float eps, float* sum_out, float* out, void* stream) {
  return univs_layer_norm_add_f32(x, residual, gamma, beta, nullptr, 1, rows, C, eps, sum_out, out, nullptr, stream);
}

int univs_patch_merge_norm_f32(const float* x, const float* gamma, const float* beta, int B, int H, int W, int C, float eps, float* out,
                               void* stream) {
  clear_sticky_error();
  if (B < 0 || H < 0 || W < 0 || C < 1) {
    set_error("univs_patch_merge_norm_f32: bad dimensions B=%d H=%d W=%d C=%d", B, H, W, C);
    return UNIVS_ERR_INVALID_ARGUMENT;
  }
  if ((long long)B * H * W == 0) return UNIVS_OK;
  if (!x || !gamma || !beta || !out || ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(gamma) | reinterpret_cast<uintptr_t>(beta) |
                                        reinterpret_cast<uintptr_t>(out)) & 15)) {
    set_error("univs_patch_merge_norm_f32: NULL or misaligned data pointer (16 bytes)");
    return UNIVS_ERR_INVALID_ARGUMENT;
  }
  const int rc = patch_merge_norm_f32(x, gamma, beta, B, H, W, C, eps, out, static_cast<hipStream_t>(stream));
  if (rc == UNIVS_ERR_NOT_IMPLEMENTED) set_error("univs_patch_merge_norm_f32: C=%d not supported (C %% 4 == 0, C <= 768)", C);
  return rc;
}

int univs_group_norm_f32(const float* x, const float* gamma, const float* beta, int N, int C, long long HW,
                         int groups, float eps, int relu, float* ws, long long ws_floats, float* out, void* stream) {
  clear_sticky_error();
  if (N < 0 || C < 1 || HW < 0 || groups < 1 || C % groups != 0 || (long long)N * C > 0x7fffffffLL) {
    set_error("univs_group_norm_f32: bad dimensions N=%d C=%d HW=%lld groups=%d", N, C, HW, groups);
    return UNIVS_ERR_INVALID_ARGUMENT;
  }
  if ((long long)N * HW == 0) return UNIVS_OK;
  if (!x || !gamma || !beta || !ws || !out) {
    set_error("univs_group_norm_f32: NULL data pointer");
    return UNIVS_ERR_INVALID_ARGUMENT;
  }
  return group_norm_f32(x, gamma, beta, N, C, HW, groups, eps, relu, ws, ws_floats, out, static_cast<hipStream_t>(stream));
}

int univs_masked_softmax_f32(float* scores, const uint8_t* mask, int N, int h, int L, int S, void* stream) {
  clear_sticky_error();
  if (N < 0 || h < 0 || L < 0 || S < 0) {
    set_error("univs_masked_softmax_f32: negative dimension");
    return UNIVS_ERR_INVALID_ARGUMENT;
  }
  if ((long long)N * h * L * S == 0) return UNIVS_OK;
  if (!scores) {
    set_error("univs_masked_softmax_f32: NULL scores");
    return UNIVS_ERR_INVALID_ARGUMENT;
  }
  return masked_softmax_f32(scores, mask, N, h, L, S, static_cast<hipStream_t>(stream));
}

int univs_proca_attention_f32(const float* qkv0, const float* kd, const float* vd, int Qp, int L, int T, int heads, int head_dim,
                              float scale, float* out, void* stream) {
  clear_sticky_error();
  if (Qp < 0 || L < 0 || T < 0 || heads < 1 || head_dim < 1) {
    set_error("univs_proca_attention_f32: bad dimensions Qp=%d L=%d T=%d heads=%d head_dim=%d", Qp, L, T, heads, head_dim);
    return UNIVS_ERR_INVALID_ARGUMENT;
  }
  if ((long long)Qp * T == 0) return UNIVS_OK;
  if (!qkv0 || !out || (L > 0 && (!kd || !vd))) {
    set_error("univs_proca_attention_f32: NULL data pointer");
    return UNIVS_ERR_INVALID_ARGUMENT;
  }
  const int rc = proca_attention_f32(qkv0, kd, vd, Qp, L, T, heads, head_dim, scale, out, static_cast<hipStream_t>(stream));
  if (rc > 0) return UNIVS_OK;
  if (rc == 0) {
    set_error("univs_proca_attention_f32: shape not covered (head_dim == 32, 1 + L <= 16384)");
    return UNIVS_ERR_NOT_IMPLEMENTED;
  }
  return rc;
}

int univs_prompt_prefix_f32(const float* masks, const float* boxes, int F, int n, int h, int w, int scale, float mask_thresh,
                            float* feat_masks, uint32_t* stats, uint8_t* sel, int32_t* rowcnt, uint8_t* fmb, int32_t* counts,
                            uint8_t* valid, uint8_t* visible, void* stream) {
  clear_sticky_error();
  if (F < 0 || n < 0 || h < 1 || w < 1 || scale < 1 || h % scale || w % scale || (long long)F * n > 65535) {
    set_error("univs_prompt_prefix_f32: bad dimensions F=%d n=%d h=%d w=%d scale=%d", F, n, h, w, scale);
    return UNIVS_ERR_INVALID_ARGUMENT;
  }
  if ((long long)F * n == 0) return UNIVS_OK;
  if (!masks || !boxes || !feat_masks || !stats || !sel || !rowcnt || !fmb || !counts || !valid || !visible) {
    set_error("univs_prompt_prefix_f32: NULL data pointer");
    return UNIVS_ERR_INVALID_ARGUMENT;
  }
  return prompt_prefix_f32(masks, boxes, F, n, h, w, scale, mask_thresh, feat_masks, stats, sel, rowcnt, fmb, counts, valid, visible,
                           static_cast<hipStream_t>(stream));
}

int univs_prompt_draw(const uint8_t* sel, const int32_t* rowcnt, const uint8_t* fmb, const int32_t* counts, const float* u,
                      const float* keys, const int64_t* tab, int F, int n, int h, int w, int HW, int R, int64_t* point_idx,
                      int64_t* dense_idx, uint8_t* empty, float* point_coords, void* stream) {
  clear_sticky_error();
  if (F < 0 || n < 0 || h < 1 || w < 1 || HW < 1 || R < 1 || (long long)F * n > 65535) {
    set_error("univs_prompt_draw: bad dimensions F=%d n=%d h=%d w=%d HW=%d R=%d", F, n, h, w, HW, R);
    return UNIVS_ERR_INVALID_ARGUMENT;
  }
  if ((long long)F * n == 0) return UNIVS_OK;
  if (!sel || !rowcnt || !fmb || !counts || !point_idx || !dense_idx || !empty || !point_coords) {
    set_error("univs_prompt_draw: NULL data pointer");
    return UNIVS_ERR_INVALID_ARGUMENT;
  }
  if ((tab != nullptr) == (u != nullptr || keys != nullptr) || (!tab && (!u || !keys))) {
    set_error("univs_prompt_draw: either (u, keys) or tab");
    return UNIVS_ERR_INVALID_ARGUMENT;
  }
  const int rc = prompt_draw(sel, rowcnt, fmb, counts, u, keys, reinterpret_cast<const long long*>(tab), F, n, h, w, HW, R,
                             reinterpret_cast<long long*>(point_idx), reinterpret_cast<long long*>(dense_idx), empty, point_coords,
                             static_cast<hipStream_t>(stream));
  if (rc > 0) return UNIVS_OK;
  if (rc == 0) {
    set_error("univs_prompt_draw: the keys of one entity do not fit the LDS (HW = %d)", HW);
    return UNIVS_ERR_NOT_IMPLEMENTED;
  }
  return rc;
}

int univs_prompt_tokens_f32(const float* feats, const int64_t* feats_strides, const float* pos, const int64_t* pos_strides,
                            const float* qfeat, const float* qpe, const int64_t* dense_idx, const uint8_t* empty, const uint8_t* valid,
                            const float* boxes, const int64_t* kf, int F, int n, int R, int T, int C, int h_img, int w_img, float* fd,
                            float* pd, uint8_t* attn, void* stream) {
  clear_sticky_error();
  if (F < 0 || n < 0 || R < 1 || T < 1 || C < 1 || h_img < 1 || w_img < 1 || (long long)F * T * n > 65535) {
    set_error("univs_prompt_tokens_f32: bad dimensions F=%d n=%d R=%d T=%d C=%d h_img=%d w_img=%d", F, n, R, T, C, h_img, w_img);
    return UNIVS_ERR_INVALID_ARGUMENT;
  }
  if ((long long)F * n == 0) return UNIVS_OK;
  if (!feats || !feats_strides || !pos || !pos_strides || !qfeat || !qpe || !dense_idx || !empty || !valid || !boxes || !kf || !fd || !pd ||
      !attn) {
    set_error("univs_prompt_tokens_f32: NULL data pointer");
    return UNIVS_ERR_INVALID_ARGUMENT;
  }
  return prompt_tokens_f32(feats, reinterpret_cast<const long long*>(feats_strides), pos, reinterpret_cast<const long long*>(pos_strides),
                           qfeat, qpe, reinterpret_cast<const long long*>(dense_idx), empty, valid, boxes,
                           reinterpret_cast<const long long*>(kf), F, n, R, T, C, h_img, w_img, fd, pd, attn,
                           static_cast<hipStream_t>(stream));
}

int univs_prompt_point_pe_f32(const float* xy, const float* z, const float* dim_t, const float* dim_tz, float scale, int F, int n, int Fq,
                              float* out, void* stream) {
  clear_sticky_error();
  if (F < 0 || n < 0 || Fq < 1) {
    set_error("univs_prompt_point_pe_f32: bad dimensions F=%d n=%d Fq=%d", F, n, Fq);
    return UNIVS_ERR_INVALID_ARGUMENT;
  }
  if ((long long)F * n == 0) return UNIVS_OK;
  if (!xy || !z || !dim_t || !dim_tz || !out) {
    set_error("univs_prompt_point_pe_f32: NULL data pointer");
    return UNIVS_ERR_INVALID_ARGUMENT;
  }
  return prompt_point_pe_f32(xy, z, dim_t, dim_tz, scale, F, n, Fq, out, static_cast<hipStream_t>(stream));
}

int univs_mask_stats_strided_f32(const float* x, long long outer, int inner, long long stride_outer, long long stride_inner, int H, int W,
                                 int h_valid, int w_valid, float t_hi, float t_lo, float t_box, int32_t* out, void* stream) {
  clear_sticky_error();
  if (outer < 0 || inner < 1 || H < 1 || W < 1 || h_valid < 0 || w_valid < 0 || h_valid > H || w_valid > W || stride_outer < 0 ||
      stride_inner < (long long)H * W) {
    set_error("univs_mask_stats_f32: bad dimensions outer=%lld inner=%d H=%d W=%d h_valid=%d w_valid=%d strides %lld / %lld", outer, inner, H, W,
              h_valid, w_valid, stride_outer, stride_inner);
    return UNIVS_ERR_INVALID_ARGUMENT;
  }
  if (outer == 0) return UNIVS_OK;
  if (!x || !out) {
    set_error("univs_mask_stats_f32: NULL data pointer");
    return UNIVS_ERR_INVALID_ARGUMENT;
  }
  const int rc = mask_stats_f32(x, outer * inner, inner, stride_outer, stride_inner, H, W, h_valid, w_valid, t_hi, t_lo, t_box, out,
                                static_cast<hipStream_t>(stream));
  if (rc == UNIVS_ERR_NOT_IMPLEMENTED) set_error("univs_mask_stats_f32: planes=%lld not covered (<= 65535)", outer * inner);
  return rc;
}

int univs_mask_stats_f32(const float* x, long long planes, int H, int W, int h_valid, int w_valid, float t_hi, float t_lo, float t_box,
                         int32_t* out, void* stream) {
  if (planes > 0x7fffffffLL) {
    clear_sticky_error();
    set_error("univs_mask_stats_f32: planes=%lld not covered (<= 65535)", planes);
    return UNIVS_ERR_NOT_IMPLEMENTED;
  }
  return univs_mask_stats_strided_f32(x, planes > 0 ? 1 : 0, (int)std::max<long long>(planes, 1), 0, (long long)H * W, H, W, h_valid, w_valid, t_hi,
                                      t_lo, t_box, out, stream);
}

int univs_token_mean_f32(const float* x, const float* add, int n, int L, int T, int C, float* out, void* stream) {
  clear_sticky_error();
  if (n < 0 || L < 0 || T < 0 || C < 1) {
    set_error("univs_token_mean_f32: bad dimensions n=%d L=%d T=%d C=%d", n, L, T, C);
    return UNIVS_ERR_INVALID_ARGUMENT;
  }
  if ((long long)n * T == 0) return UNIVS_OK;
  if (!x || !out) {
    set_error("univs_token_mean_f32: NULL data pointer");
    return UNIVS_ERR_INVALID_ARGUMENT;
  }
  const int rc = token_mean_f32(x, add, n, L, T, C, out, static_cast<hipStream_t>(stream));
  if (rc > 0) return UNIVS_OK;
  if (rc == 0) {
    set_error("univs_token_mean_f32: shape not covered (C <= 1024, L <= 15360)");
    return UNIVS_ERR_NOT_IMPLEMENTED;
  }
  return rc;
}

int univs_window_attention_image_f32(const float* qkv, const float* qkv_bias, const float* bias,
                                     const float* shift_mask, int B, int H, int W, int ws, int shift, int nH,
                                     int hd, float scale, float* out, void* stream) {
  clear_sticky_error();
  if (B < 0 || H < 1 || W < 1 || ws < 1 || shift < 0 || shift >= ws || nH < 1 || hd < 1) {
    set_error("univs_window_attention_image_f32: bad dimensions B=%d H=%d W=%d ws=%d shift=%d nH=%d hd=%d", B, H, W,
              ws, shift, nH, hd);
    return UNIVS_ERR_INVALID_ARGUMENT;
  }
  if (B == 0) return UNIVS_OK;
  if (!qkv || !bias || !out) {
    set_error("univs_window_attention_image_f32: NULL data pointer");
    return UNIVS_ERR_INVALID_ARGUMENT;
  }
  return window_attention_image_f32(qkv, qkv_bias, bias, shift_mask, B, H, W, ws, shift, nH, hd, scale, out,
                                    static_cast<hipStream_t>(stream));
}

int univs_window_attention_image_mma(const float* qkv, const float* qkv_bias, const float* bias,
                                     const float* shift_mask, int B, int H, int W, int ws, int shift, int nH,
                                     int hd, float scale, int mma, float* out, void* stream) {
  if (mma == UNIVS_MMA_F32)
    return univs_window_attention_image_f32(qkv, qkv_bias, bias, shift_mask, B, H, W, ws, shift, nH, hd, scale, out, stream);
  clear_sticky_error();
  if (mma != UNIVS_MMA_F16 && mma != UNIVS_MMA_F16X3) {
    set_error("univs_window_attention_image_mma: mma=%d (UNIVS_MMA_F32 = 0, UNIVS_MMA_F16 = 1 or UNIVS_MMA_F16X3 = 2)", mma);
    return UNIVS_ERR_INVALID_ARGUMENT;
  }
  if (B < 0 || H < 1 || W < 1 || ws < 1 || shift < 0 || shift >= ws || nH < 1 || hd < 1) {
    set_error("univs_window_attention_image_mma: bad dimensions B=%d H=%d W=%d ws=%d shift=%d nH=%d hd=%d", B, H, W,
              ws, shift, nH, hd);
    return UNIVS_ERR_INVALID_ARGUMENT;
  }
  if (B == 0) return UNIVS_OK;
  if (!qkv || !bias || !out) {
    set_error("univs_window_attention_image_mma: NULL data pointer");
    return UNIVS_ERR_INVALID_ARGUMENT;
  }
  const int rc = window_attention_image_f16mma(qkv, qkv_bias, bias, shift_mask, B, H, W, ws, shift, nH, hd, scale,
                                               mma == UNIVS_MMA_F16X3 ? 3 : 1, out, static_cast<hipStream_t>(stream));
  if (rc == UNIVS_ERR_NOT_IMPLEMENTED && mma == UNIVS_MMA_F16X3)      // windows beyond 9 x 9: the exact kernel (same accuracy class)
    return univs_window_attention_image_f32(qkv, qkv_bias, bias, shift_mask, B, H, W, ws, shift, nH, hd, scale, out, stream);
  return rc;
}

int univs_msda_prepare_f32(const float* proj, int row_stride, int n_off, const float* ref_points,
                           long long ref_batch_stride, const int64_t* spatial_shapes, int N, int Lq, int M, int L,
                           int P, float* loc, float* attn, void* stream) {
  clear_sticky_error();
  if (N < 0 || Lq < 0 || M < 1 || L < 1 || L > UNIVS_MAX_LEVELS || P < 1 || row_stride < M * L * P * 3 || n_off < M * L * P * 2 ||
      n_off + M * L * P > row_stride || ref_batch_stride < 0) {
    set_error("univs_msda_prepare_f32: bad dimensions N=%d Lq=%d M=%d L=%d P=%d row_stride=%d n_off=%d", N, Lq, M, L, P,
              row_stride, n_off);
    return UNIVS_ERR_INVALID_ARGUMENT;
  }
  if ((long long)N * Lq == 0) return UNIVS_OK;
  if (!proj || !ref_points || !spatial_shapes || !loc || !attn) {
    set_error("univs_msda_prepare_f32: NULL pointer");
    return UNIVS_ERR_INVALID_ARGUMENT;
  }
  LevelTable lv;
  for (int l = 0; l < UNIVS_MAX_LEVELS; ++l) {
    lv.H[l] = l < L ? (int)spatial_shapes[2 * l] : 0;
    lv.W[l] = l < L ? (int)spatial_shapes[2 * l + 1] : 0;
    lv.start[l] = 0;
    if (l < L && (lv.H[l] < 1 || lv.W[l] < 1)) {
      set_error("univs_msda_prepare_f32: level %d has an empty shape", l);
      return UNIVS_ERR_INVALID_ARGUMENT;
    }
  }
  const int rc = msda_prepare_f32(proj, row_stride, n_off, ref_points, ref_batch_stride, lv, N, Lq, M, L, P, loc, attn,
                                  static_cast<hipStream_t>(stream));
  if (rc == UNIVS_ERR_NOT_IMPLEMENTED) set_error("univs_msda_prepare_f32: (L=%d, P=%d) not instantiated (P == 4, L <= 4)", L, P);
  return rc;
}

int univs_linear_blocked_f32(const float* x, const float* weight, const float* bias, long long M, int N, int K,
                             int rows_per_batch, int col_block, float* y, void* stream) {
  if (M < 0 || N < 1 || K < 1 || rows_per_batch < 1 || col_block < 4 || col_block % 4 != 0 || N % col_block != 0 ||
      (M % rows_per_batch) != 0) {
    set_error("univs_linear_blocked_f32: bad arguments M=%lld N=%d K=%d rows_per_batch=%d col_block=%d", M, N, K, rows_per_batch,
              col_block);
    return UNIVS_ERR_INVALID_ARGUMENT;
  }
  if (M == 0) return UNIVS_OK;
  clear_sticky_error();
  if (!x || !weight || !y) {
    set_error("univs_linear_blocked_f32: NULL data pointer");
    return UNIVS_ERR_INVALID_ARGUMENT;
  }
  const int rc = univs::linear_split_f32(x, weight, bias, nullptr, y, M, N, K, /*LS_EPI_BLOCKED=*/4, static_cast<hipStream_t>(stream),
                                         rows_per_batch, col_block);
  if (rc > 0) return UNIVS_OK;
  if (rc == 0) {
    set_error("univs_linear_blocked_f32: shape M=%lld N=%d K=%d (or alignment) is not covered (K == 256, M >= 2048)", M, N, K);
    return UNIVS_ERR_NOT_IMPLEMENTED;
  }
  return rc;
}

int univs_msda_forward_strips_f32(const float* value_hm, const int64_t* spatial_shapes, const int64_t* level_start,
                                  const float* proj_hm, const float* ref_points, long long ref_batch_stride, int N, int S, int M,
                                  int D, int L, int Lq, int P, float* out, void* stream) {
  if (N < 0 || S < 0 || M < 1 || D < 0 || Lq < 0 || P < 1 || L < 1 || L > UNIVS_MAX_LEVELS || ref_batch_stride < 0) {
    set_error("univs_msda_forward_strips_f32: bad dimensions N=%d S=%d M=%d D=%d L=%d Lq=%d P=%d", N, S, M, D, L, Lq, P);
    return UNIVS_ERR_INVALID_ARGUMENT;
  }
  if ((long long)N * Lq * M * D == 0) return UNIVS_OK;
  clear_sticky_error();
  g_msda_gen = 0;
  if (!value_hm || !proj_hm || !ref_points || !out) {
    set_error("univs_msda_forward_strips_f32: NULL data pointer");
    return UNIVS_ERR_INVALID_ARGUMENT;
  }
  LevelTable lv;
  int rc = make_levels(spatial_shapes, level_start, L, S, &lv, "univs_msda_forward_strips_f32");
  if (rc != UNIVS_OK) return rc;
  if (config().msda_impl == 1) {   // the generic kernel was forced: it has no head-major variant, the caller takes the two-operator path
    set_error("univs_msda_forward_strips_f32: generic implementation forced (univs_msda_set_impl(1))");
    return UNIVS_ERR_NOT_IMPLEMENTED;
  }
  rc = msda_forward_strips_f32(value_hm, lv, proj_hm, ref_points, ref_batch_stride, N, S, M, D, L, Lq, P, out,
                               static_cast<hipStream_t>(stream));
  if (rc > 0) {
    g_msda_last = 2;
    g_msda_gen = 5;
    return UNIVS_OK;
  }
  if (rc == 0) {
    set_error("univs_msda_forward_strips_f32: geometry not covered (D == 32, P == 4, 1 <= L <= 4, Lq == S, windows within 80 KB of LDS)");
    return UNIVS_ERR_NOT_IMPLEMENTED;
  }
  return rc;
}

int univs_msda_forward_heads_f32(const float* value_hm, const int64_t* spatial_shapes, const int64_t* level_start,
                                  const float* proj_hm, const float* ref_points, long long ref_batch_stride, int N, int S, int M,
                                  int D, int L, int Lq, int P, float* out, void* stream) {
  if (N < 0 || S < 0 || M < 1 || D < 0 || Lq < 0 || P < 1 || L < 1 || L > UNIVS_MAX_LEVELS || ref_batch_stride < 0) {
    set_error("univs_msda_forward_heads_f32: bad dimensions N=%d S=%d M=%d D=%d L=%d Lq=%d P=%d", N, S, M, D, L, Lq, P);
    return UNIVS_ERR_INVALID_ARGUMENT;
  }
  if ((long long)N * Lq * M * D == 0) return UNIVS_OK;
  clear_sticky_error();
  g_msda_gen = 0;
  if (!value_hm || !proj_hm || !ref_points || !out) {
    set_error("univs_msda_forward_heads_f32: NULL data pointer");
    return UNIVS_ERR_INVALID_ARGUMENT;
  }
  LevelTable lv;
  int rc = make_levels(spatial_shapes, level_start, L, S, &lv, "univs_msda_forward_heads_f32");
  if (rc != UNIVS_OK) return rc;
  if (config().msda_impl == 1) {   // the generic kernel was forced: it has no head-major variant, the caller takes the two-operator path
    set_error("univs_msda_forward_heads_f32: generic implementation forced (univs_msda_set_impl(1))");
    return UNIVS_ERR_NOT_IMPLEMENTED;
  }
  rc = msda_forward_heads_f32(value_hm, lv, proj_hm, ref_points, ref_batch_stride, N, S, M, D, L, Lq, P, out,
                               static_cast<hipStream_t>(stream));
  if (rc > 0) {
    g_msda_last = 2;
    g_msda_gen = 6;
    return UNIVS_OK;
  }
  if (rc == 0) {
    set_error("univs_msda_forward_heads_f32: geometry not covered (D == 32, P == 4, 1 <= L <= 4, Lq == S, windows within 160 KB of LDS)");
    return UNIVS_ERR_NOT_IMPLEMENTED;
  }
  return rc;
}

}  // extern "C"
