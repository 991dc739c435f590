// LDS-tiled MSDA forward, sixth generation (gfx950): a FULL head per lane-sample, one 8-wave workgroup per CU, columns
// walked in lockstep by the workgroups of an XCD.
//
// Operator: the core of MSDeformAttn.forward (ops/modules/ms_deform_attn.py:100-121): softmax over the L*P logits,
// sampling locations = reference + offset / (W_l, H_l), and ms_deform_attn_forward (ops/src/ms_deform_attn.h:25-44; kernel
// ms_deform_im2col_cuda.cuh:242-304, bilinear helper :38-89) for the encoder geometry (Lq == S, D = 32, P = 4), fed with the
// RAW projections -- the location / weight tensors never exist.
//
// What generation 5 (msda_strips.hip: half a head per lane-sample, two workgroups per CU) measured: 600 vector instructions
// per wave-item of which 96 are the interpolation's multiply-adds -- the per-sample record (location -> window row / column ->
// four LDS addresses and weights), the softmax, the exact division and the point reduction are paid once per HALF head -- and
// 1.73 x the algorithmic bytes from HBM: two workgroups that own horizontally adjacent tiles fetch the halo columns they share
// at different times.  This generation changes both:
//
//   * a lane owns a sample of a FULL head: 32 channels = 128 bytes per pixel = one whole line of HBM and half a bank row of
//     LDS.  The record, the softmax, the division and the reduction serve twice the channels; a level costs 32 ds_read_b128
//     + 64 v_pk_fma_f32 + 28 address XORs + ~40 record instructions.  The windows of a 12 x 8 tile (R = 6) are 153 KB: ONE
//     workgroup of 8 waves per CU (two per SIMD, <= 256 VGPRs), which hide LDS latency inside the wave: the reads of a
//     corner are issued while the previous corner's multiply-adds run (two register sets of 32).
//   * plain row-major windows, rows circular (msda_heads_geom.h); bank-conflict-free because a lane reads the horizontal
//     corner whose column parity equals its lane bit 3 first and the eight chunks of a pixel in the order j ^ (lane & 7).
//   * SEGMENTS instead of one contiguous range per workgroup (s6_build_segments): the W workgroups of an XCD walk W adjacent
//     columns of tiles top to bottom at the same time, so the halo columns two neighbours share come from HBM once.
//   * value head-major as full heads [N][M][S][32] (value_proj's epilogue: univs_linear_blocked_f32(..., S, 32)); the
//     projections [N][M][S][P][3L] as in generation 5.
#include <cstring>
#include <memory>
#include <mutex>
#include <type_traits>

#include "msda_heads_geom.h"
#include "config.h"
#include "msda_dev.h"

// Timing experiments (python -m univs_amd.build --ablate heads_*; results are then WRONG): S6_ABLATE bit 0 = no gather stream
// (LDS reads + multiply-adds), bit 1 = no row movement in the steady state (loads + LDS commits), bit 2 = no point
// reduction / output stores, bit 3 = no sample records (and no stream).
#ifndef S6_ABLATE
#define S6_ABLATE 0
#endif
// S6_TRACE (python -m univs_amd.build --ablate heads_trace): lane 0 of every wave keeps ten s_memtime stamps per item in
// registers and stores them behind barrier B: g_s6_trace[workgroup][item serial < 48][wave][10], read back with
// univs_dbg_s6_trace (tools/msda_trace6.py).
#ifdef S6_TRACE
#define S6_TRACE_ITEMS 48
__device__ unsigned long long g_s6_trace[256 * S6_TRACE_ITEMS * 8 * 10];
#define S6_STAMP(K) { __builtin_amdgcn_sched_barrier(0); stamp[K] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
#else
#define S6_STAMP(K)
#endif

namespace univs {

typedef float s6v4u __attribute__((ext_vector_type(4), aligned(4)));   // a 16-byte load that is only 4-byte aligned
typedef unsigned s6u3 __attribute__((ext_vector_type(3)));   // a piece's three used dwords

struct S6Args {
  const float* vhm;    // value, head-major [N][M][S][32]
  const float* qhm;    // projections, head-major [N][M][S][P][3 L]: L offset pairs (x, y), then L logits; levels in SLOT order
  const float* ref;    // reference points [N or 1][S][2]: one per query, the same for every level (the encoder's pixel centres)
  long long ref_batch_stride;   // floats between frames; 0: one set for all frames
  float* out;          // [N][S][M * 32]
  int N, S, M;
};

template <int L>
__global__ __launch_bounds__(64 * S6_NW, 2) void msda_fwd_heads(S6Args a, S6Levels lv, const S6Tile* __restrict__ tiles,
                                                                 const S6Piece* __restrict__ pieces,
                                                                 const int* __restrict__ qtab, const S6Seg* __restrict__ segs,
                                                                 const int* __restrict__ seg_begin, unsigned lds_dummy) {
  constexpr int P = 4, DH = S6_DH;
  extern __shared__ __attribute__((aligned(1024))) char lds6[];
  const unsigned lds_base = (unsigned)(unsigned long long)(T3_LDS char*)lds6;

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int S = a.S, M = a.M;

  auto header = [&](int tile) __attribute__((always_inline)) { return reinterpret_cast<const int*>(tiles + tile)[lane & 15]; };
  enum { HD_P0 = 0, HD_P1 = S6_LMAX, HD_TOTAL = 2 * S6_LMAX, HD_NCOLD = 2 * S6_LMAX + 1, HD_NENTER = 2 * S6_LMAX + 2 };
  auto hfield = [&](int hdv, int idx) __attribute__((always_inline)) { return __builtin_amdgcn_readlane(hdv, idx); };

  // =========================== moving rows ===========================
  // A piece = 8 pixels of one row of one level: lane (pixel = lane >> 3, chunk = lane & 7) moves 16 bytes.  Its descriptor is two
  // dwords (msda_heads_geom.h: S6Piece): the pixel index with the load mask in the top byte, the LDS offset with the store mask
  // in the top byte -- one v_readlane per request and per commit.  Masked-out lanes load from an offset outside the buffer
  // (0, no memory traffic) and store into a 1-KB dummy region behind the windows: no branch, no exec-mask change.
  const int lpx = lane >> 3, lch = lane & 7;
  const unsigned lanepart_g = (unsigned)((lpx - S6_PX_BIAS) * (DH * 4) + lch * 16), lanepart_l = (unsigned)(lpx * 128 + lch * 16);
  const unsigned maskpos = 24u + (unsigned)lpx;
  const unsigned dummy_l = lds_dummy + (unsigned)lane * 16u;
  t3v4 wreg[S6_PC];
  static_assert(S6_PCAP <= 64 && S6_PCAP % S6_PC == 0, "a wave fetches its piece list with one load; whole passes");
  auto piece_list = [&](int tile, int which) __attribute__((always_inline)) {   // lane k: piece k of my list
    const S6Piece* p = pieces + ((long long)(tile * 2 + which) * S6_NW + wave) * S6_PCAP + min(lane, S6_PCAP - 1);
    return *reinterpret_cast<const t3u2*>(p);
  };
  auto value_rsrc = [&](unsigned hd) __attribute__((always_inline)) {
    // one buffer resource over this (frame, head)'s value pixels
    const unsigned long long pv = (unsigned long long)(a.vhm + (long long)hd * S * DH);
    const unsigned plo = __builtin_amdgcn_readfirstlane((unsigned)pv), phi = __builtin_amdgcn_readfirstlane((unsigned)(pv >> 32));
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<float*>(((unsigned long long)phi << 32) | plo), 0,
                                             (int)((long long)S * DH * 4), 0x00020000);
  };
  auto request = [&](__amdgpu_buffer_rsrc_t rsrc, unsigned pa) __attribute__((always_inline)) {   // pa: uniform
    const unsigned on = (unsigned)__builtin_amdgcn_sbfe((int)pa, maskpos, 1u);                      // all ones: my column is in the level
    const unsigned off = (((pa & 0xffffffu) << 7) + lanepart_g) | ~on;                              // (~on: far outside the buffer)
    return __builtin_bit_cast(t3v4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0));
  };
  auto commit = [&](unsigned pb, const t3v4& v) __attribute__((always_inline)) {                    // pb: uniform
    const unsigned on = (unsigned)__builtin_amdgcn_sbfe((int)pb, maskpos, 1u);                      // all ones: my column is in the pitch
    const unsigned adr = (((pb & 0xffffffu) + lanepart_l) & on) | (dummy_l & ~on);
    *(T3_LDS t3v4*)((T3_LDS char*)lds6 + adr) = v;
  };

  // =========================== gathering ===========================
  const int qi = lane & 15, pt = lane >> 4;           // my sample: query qi of the wave's 16, point pt
  const unsigned rot8 = (unsigned)lane & 7u;           // my chunk rotation
  const int qslot = wave * 16 + qi;                    // my query's index within an item

  auto my_query = [&](int tile) __attribute__((always_inline)) { return qtab[tile * S6_QCAP + qslot]; };
  float Hf[L], Wf[L];
#pragma unroll
  for (int kk = 0; kk < L; ++kk) { Hf[kk] = (float)lv.H[kk]; Wf[kk] = (float)lv.W[kk]; }

  struct Inputs { float x[L], y[L], a[L]; };
  struct RawInputs { float v[3 * L]; float2 rp; };
  struct Recs { unsigned ca[4 * L]; float cw[4 * L]; unsigned long long mm[L]; };
  // my (query, head, point)'s 3 L floats -- L offset pairs then L logits (slot order) -- and the query's reference point:
  // loads only; `finish_inputs` does the arithmetic an item later, when the loads have long arrived
  auto load_raw = [&](int n, int m, int qg, RawInputs& r) __attribute__((always_inline)) {
    // uniform 64-bit bases + 32-bit lane offsets (host-checked: the projections of one (frame, head) stay below 4 GB)
    const char* rowb = reinterpret_cast<const char*>(a.qhm + ((long long)n * M + m) * S * (P * 3 * L));
    const char* refb = reinterpret_cast<const char*>(a.ref + n * a.ref_batch_stride);
    const unsigned ro = ((unsigned)qg * P + (unsigned)pt) * (unsigned)(3 * L * 4);
    const float* row = reinterpret_cast<const float*>(rowb + ro);
    r.rp = *reinterpret_cast<const float2*>(refb + (unsigned)qg * 8u);
    if constexpr (L == 3) {
      const s6v4u r0 = *reinterpret_cast<const s6v4u*>(row), r1 = *reinterpret_cast<const s6v4u*>(row + 4);
      r.v[0] = r0.x; r.v[1] = r0.y; r.v[2] = r0.z; r.v[3] = r0.w;
      r.v[4] = r1.x; r.v[5] = r1.y; r.v[6] = r1.z; r.v[7] = r1.w;
      r.v[8] = row[8];
    } else {
#pragma unroll
      for (int i = 0; i < 3 * L; ++i) r.v[i] = row[i];
    }
  };
  auto finish_inputs = [&](const RawInputs& r, Inputs& iv) __attribute__((always_inline)) {
    float lg[L];
#pragma unroll
    for (int kk = 0; kk < L; ++kk) {
      lg[kk] = r.v[2 * L + kk];
      // offset / (W_l, H_l), IEEE-exact: q = a * RN(1 / b), corrected by the exact remainder (two FMAs; b is a small integer
      // and a is far from the ends of the exponent range, so the corrected quotient is the correctly rounded one --
      // tools/heads_emulate.cpp compares it with the division on every sample)
      const float qx = r.v[2 * kk] * lv.rW[kk], qy = r.v[2 * kk + 1] * lv.rH[kk];
      const float ox = fmaf(fmaf(-qx, Wf[kk], r.v[2 * kk]), lv.rW[kk], qx);
      const float oy = fmaf(fmaf(-qy, Hf[kk], r.v[2 * kk + 1]), lv.rH[kk], qy);
      iv.x[kk] = r.rp.x + ox;
      iv.y[kk] = r.rp.y + oy;
    }
    // softmax over the L * P logits of (query, head): the 4 points of a query sit in the 4 DPP rows
    auto all_rows = [&](float v, bool is_max) __attribute__((always_inline)) {
      t3u2 s1 = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
      const float a1 = __uint_as_float(s1.x), b1 = __uint_as_float(s1.y);
      const float r1 = is_max ? fmaxf(a1, b1) : a1 + b1;
      t3u2 s2 = __builtin_amdgcn_permlane16_swap(__float_as_uint(r1), __float_as_uint(r1), false, false);
      const float a2 = __uint_as_float(s2.x), b2 = __uint_as_float(s2.y);
      return is_max ? fmaxf(a2, b2) : a2 + b2;
    };
    float mx = lg[0];
#pragma unroll
    for (int kk = 1; kk < L; ++kk) mx = fmaxf(mx, lg[kk]);
    mx = all_rows(mx, true);
    float sum = 0.f;
#pragma unroll
    for (int kk = 0; kk < L; ++kk) {
      iv.a[kk] = __builtin_amdgcn_exp2f((lg[kk] - mx) * 1.44269504088896340736f);   // arguments <= 0: no range handling needed
      sum += iv.a[kk];
    }
    sum = all_rows(sum, false);
    const float rs = __builtin_amdgcn_rcpf(sum);   // sum in [1, L * P]
#pragma unroll
    for (int kk = 0; kk < L; ++kk) iv.a[kk] = iv.a[kk] * rs;
  };
  // my sample's records at every level (msda_heads_geom.h: s6_record, shared with the host emulator) and the rare path's
  // masks (evaluated HERE: a branch between reads and multiply-adds lets hipcc sink the latter into its successor)
  auto make_records = [&](const Inputs& in, int hdv, Recs& r) __attribute__((always_inline)) {
#pragma unroll
    for (int kk = 0; kk < L; ++kk) {
      const S6Rec rec = s6_record(in.x[kk], in.y[kk], in.a[kk], Hf[kk], Wf[kk], (unsigned)hfield(hdv, HD_P0 + kk),
                                  (unsigned)hfield(hdv, HD_P1 + kk), lv.nr[kk], lv.pitch[kk], lv.next_d[kk], lv.wrap_d[kk],
                                  lds_base + (unsigned)lv.reg[kk], (unsigned)lane & 15u);
#pragma unroll
      for (int k = 0; k < 4; ++k) { r.ca[4 * kk + k] = rec.a[k]; r.cw[4 * kk + k] = rec.w[k]; }
      r.mm[kk] = __ballot(!rec.inwin);
      if (r.mm[kk] != 0)
        r.mm[kk] = __ballot(!rec.inwin && in.a[kk] != 0.f && s6_inband(in.x[kk], in.y[kk], Hf[kk], Wf[kk]));
    }
  };
  // sum the 4 points (DPP rows) of every query and store: after the two swap rounds row r of the wave holds the finished
  // chunk slots 2 r and 2 r + 1 of each query = channel chunks (2 r) ^ rot8 and (2 r + 1) ^ rot8
  // (`live`: my query slot is one of the tile's queries.  The slots behind them repeat the tile's last query -- same inputs, but a
  // lane's corner ORDER depends on its lane bits, so a repeat may round differently: it computes and does not store, or two
  // runs could differ in the last bit by which lane stored last.)
  auto reduce_store = [&](const t3v4 (&acc)[8], int n, int m, int qg, bool live) __attribute__((always_inline)) {
    if constexpr (S6_ABLATE & 4) {
      if (acc[0].x == 1.2345f) a.out[lane] = acc[0].x + acc[1].y + acc[7].w;
    } else {
      float a32[32];
#pragma unroll
      for (int j = 0; j < 8; ++j) { a32[4 * j] = acc[j].x; a32[4 * j + 1] = acc[j].y; a32[4 * j + 2] = acc[j].z; a32[4 * j + 3] = acc[j].w; }
      float s16[16], t8[8];
#pragma unroll
      for (int f = 0; f < 16; ++f) {
        const t3u2 sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(a32[f]), __float_as_uint(a32[f + 16]), false, false);
        s16[f] = __uint_as_float(sw.x) + __uint_as_float(sw.y);
      }
#pragma unroll
      for (int f = 0; f < 8; ++f) {
        const t3u2 sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(s16[f]), __float_as_uint(s16[f + 8]), false, false);
        t8[f] = __uint_as_float(sw.x) + __uint_as_float(sw.y);
      }
      char* ob = reinterpret_cast<char*>(a.out + ((long long)n * S * M + m) * 32);                             // uniform
      const unsigned oo = (unsigned)qg * (unsigned)(M * 128) + ((((unsigned)pt << 1) ^ rot8) << 4);          // < 2^32 (host-checked)
      if (live) {
        *reinterpret_cast<t3v4*>(ob + oo) = (t3v4){t8[0], t8[1], t8[2], t8[3]};
        *reinterpret_cast<t3v4*>(ob + (oo ^ 16u)) = (t3v4){t8[4], t8[5], t8[6], t8[7]};
      }
    }
  };

  // The gather is ONE stream over the item's 4 L corners (level-major): the reads of corner c + 2 are issued behind the
  // multiply-adds of corner c, four chunks at a time, into the register set those multiply-adds just freed (two sets of 8 x 16
  // bytes).  A read statement takes the four accumulators its predecessors wrote as "+v" operands: the data dependence is
  // what keeps hipcc from sinking the multiply-adds below every read (legal, and then the wave needs 4 L register sets and its
  // LDS phase and FMA phase no longer overlap; __builtin_amdgcn_sched_barrier does not stop it, the order is already fixed when
  // the DAG is linearised).  No wait inside a read statement; S6_WAIT4 makes four chunks visible to the compiler.
#define S6_READ4(A, J0, D)                                                                                              \
  asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %5\n\tds_read_b128 %2, %6\n\tds_read_b128 %3, %7"            \
               : "=&v"(D[J0]), "=&v"(D[J0 + 1]), "=&v"(D[J0 + 2]), "=&v"(D[J0 + 3])                                     \
               : "v"((A) ^ (unsigned)((J0) << 4)), "v"((A) ^ (unsigned)((J0 + 1) << 4)),                                \
                 "v"((A) ^ (unsigned)((J0 + 2) << 4)), "v"((A) ^ (unsigned)((J0 + 3) << 4))                             \
               : "memory")
#define S6_READ4_DEP(A, J0, D, ACC)                                                                                     \
  asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %9\n\tds_read_b128 %2, %10\n\tds_read_b128 %3, %11"          \
               : "=&v"(D[J0]), "=&v"(D[J0 + 1]), "=&v"(D[J0 + 2]), "=&v"(D[J0 + 3]), "+v"(ACC[J0]), "+v"(ACC[J0 + 1]),  \
                 "+v"(ACC[J0 + 2]), "+v"(ACC[J0 + 3])                                                                   \
               : "v"((A) ^ (unsigned)((J0) << 4)), "v"((A) ^ (unsigned)((J0 + 1) << 4)),                                \
                 "v"((A) ^ (unsigned)((J0 + 2) << 4)), "v"((A) ^ (unsigned)((J0 + 3) << 4))                             \
               : "memory")
#define S6_WAIT4(N, J0, D)                                                                             \
  asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(D[J0]), "+v"(D[J0 + 1]), "+v"(D[J0 + 2]), "+v"(D[J0 + 3]) : : "memory")
#define S6_FMA4(J0, D, WGT)                                                                                           \
  {                                                                                                                   \
    const t3v2 w2_ = {WGT, WGT};                                                                                      \
    _Pragma("unroll") for (int j_ = J0; j_ < J0 + 4; ++j_) {                                                          \
      const t3v2 lo = __builtin_elementwise_fma(w2_, (t3v2){D[j_].x, D[j_].y}, (t3v2){acc[j_].x, acc[j_].y});         \
      const t3v2 hi = __builtin_elementwise_fma(w2_, (t3v2){D[j_].z, D[j_].w}, (t3v2){acc[j_].z, acc[j_].w});         \
      acc[j_] = (t3v4){lo.x, lo.y, hi.x, hi.y};                                                                       \
    }                                                                                                                 \
  }

  // ---- the workgroup's segments
#ifdef S6_TRACE
  int trace_serial = 0;
#endif
  const int si0 = __builtin_amdgcn_readfirstlane(seg_begin[blockIdx.x]), si1 = __builtin_amdgcn_readfirstlane(seg_begin[blockIdx.x + 1]);
#pragma unroll 1
  for (int si = si0; si < si1; ++si) {
    const S6Seg* sp = segs + si;
    const int count = __builtin_amdgcn_readfirstlane(sp->count);
    if (count <= 0) continue;   // uniform
    const unsigned hd = (unsigned)__builtin_amdgcn_readfirstlane(sp->plane);
    const int tile0 = __builtin_amdgcn_readfirstlane(sp->tile0);
    const int n = (int)(hd / (unsigned)M), m = (int)(hd - (unsigned)n * (unsigned)M);
    const int tlast = tile0 + count - 1;
    const __amdgpu_buffer_rsrc_t vrs = value_rsrc(hd);

    // ---- prologue: the whole windows of the segment's first tile (a cold start), the first two query lists, the first inputs.
    // ONE round trip: every piece of the list (S6_PCAP, no-op pieces included: they touch no memory) is requested before the
    // first is waited for -- the accumulators and gather registers are dead here --, the inputs travel beside them.
    if (si > si0) __syncthreads();   // nobody gathers from the previous segment's windows any more
    const t3u2 clist = piece_list(tile0, 1);
    int hdv = header(tile0);
    int qg_cur = my_query(tile0);
    int qg_nxt = my_query(min(tile0 + 1, tlast));
    t3u2 rows = piece_list(min(tile0 + 1, tlast), 0);   // the rows entering the next tile's windows
    if (count < 2) rows = (t3u2){0u, 0u};               // no next tile: no-op pieces
    Inputs in_cur;
    {
      t3v4 creg[S6_PCAP];
#pragma unroll
      for (int j = 0; j < S6_PCAP; ++j) creg[j] = request(vrs, __builtin_amdgcn_readlane(clist.x, j));
      RawInputs r0;
      load_raw(n, m, qg_cur, r0);
      finish_inputs(r0, in_cur);
#pragma unroll
      for (int j = 0; j < S6_PCAP; ++j) commit(__builtin_amdgcn_readlane(clist.y, j), creg[j]);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);   // (see the wait before the output stores)
    __syncthreads();

#pragma unroll 1
    for (int tile = tile0;; ++tile) {
      const bool has_next = tile < tlast;
      const int tnxt = min(tile + 1, tlast), tn2 = min(tile + 2, tlast);
#ifdef S6_TRACE
      unsigned long long stamp[10];
#endif
      S6_STAMP(0)
      // ---- 0. the NEXT item's inputs are requested here (its query list was fetched an item ago); the rows entering its
      // windows are requested inside the gather stream, its header and the lists of the item after it ahead of the last level
      RawInputs raw_nxt;
      load_raw(n, m, qg_nxt, raw_nxt);
      if constexpr (!(S6_ABLATE & 2) && (S6_ABLATE & 9)) {   // (no stream to spread them over)
#pragma unroll
        for (int j = 0; j < S6_PC; ++j) wreg[j] = request(vrs, __builtin_amdgcn_readlane(rows.x, j));
      }
      int hdv_nxt = 0, qg_n2 = 0;
      t3u2 rows_n2 = {0u, 0u};
      __builtin_amdgcn_sched_barrier(0);
      S6_STAMP(1)

      // ---- A. my sample's records at every level
      Recs rc;
      make_records(in_cur, hdv, rc);
      S6_STAMP(2)

      // ---- B. gather: 4 L corners x 8 chunks in one stream
      t3v4 acc[8];   // my sample's 32 channels, chunk slot j = channel chunk j ^ rot8; summed over corners and levels
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = (t3v4){0.f, 0.f, 0.f, 0.f};
      if constexpr (S6_ABLATE & 9) {   // (keep the records alive)
        unsigned keep = 0;
        if constexpr (!(S6_ABLATE & 8)) {
#pragma unroll
          for (int c = 0; c < 4 * L; ++c) keep ^= rc.ca[c] ^ __float_as_uint(rc.cw[c]);
        }
        acc[0].x = __uint_as_float(keep);
        hdv_nxt = header(tnxt);
        qg_n2 = my_query(tn2);
        rows_n2 = piece_list(tn2, 0);
      } else {
        constexpr int NC = 4 * L;
        t3v4 d0[8], d1[8];
        S6_READ4(rc.ca[0], 0, d0);
        S6_READ4(rc.ca[0], 4, d0);
        S6_READ4(rc.ca[1], 0, d1);
        S6_READ4(rc.ca[1], 4, d1);
#pragma unroll
        for (int c = 0; c < NC; c += 2) {
          if (c == NC - 4) {   // (ahead of the last level's corners: L2 hits)
            hdv_nxt = header(tnxt);
            qg_n2 = my_query(tn2);
            rows_n2 = piece_list(tn2, 0);
          }
          if constexpr (!(S6_ABLATE & 2)) {   // the rows entering the next tile's windows: S6_PC requests spread over the stream
            // (issued in one burst at the top, they -- 96 KB per CU with the inputs -- wait ~1.7k clocks for the vector-memory
            // queue while every SIMD idles)
            constexpr int per = (S6_PC + NC / 2 - 1) / (NC / 2);
#pragma unroll
            for (int j = (c / 2) * per; j < (c / 2 + 1) * per && j < S6_PC; ++j)
              wreg[j] = request(vrs, __builtin_amdgcn_readlane(rows.x, j));
          }
          if (c + 2 < NC) {   // 16 reads in flight before and after every step
            S6_WAIT4(12, 0, d0); S6_FMA4(0, d0, rc.cw[c]);     S6_READ4_DEP(rc.ca[c + 2], 0, d0, acc);
            S6_WAIT4(12, 4, d0); S6_FMA4(4, d0, rc.cw[c]);     S6_READ4_DEP(rc.ca[c + 2], 4, d0, acc);
            S6_WAIT4(12, 0, d1); S6_FMA4(0, d1, rc.cw[c + 1]); S6_READ4_DEP(rc.ca[c + 3], 0, d1, acc);
            S6_WAIT4(12, 4, d1); S6_FMA4(4, d1, rc.cw[c + 1]); S6_READ4_DEP(rc.ca[c + 3], 4, d1, acc);
          } else {
            S6_WAIT4(12, 0, d0); S6_FMA4(0, d0, rc.cw[c]);
            S6_WAIT4(8, 4, d0);  S6_FMA4(4, d0, rc.cw[c]);
            S6_WAIT4(4, 0, d1);  S6_FMA4(0, d1, rc.cw[c + 1]);
            S6_WAIT4(0, 4, d1);  S6_FMA4(4, d1, rc.cw[c + 1]);
          }
        }
      }
      S6_STAMP(3)

      // ---- C. rare: samples whose footprint leaves the tile's window -> the whole wave fetches the four corners from
      // global memory (lane = corner lane >> 4, channels 2 (lane & 15) and + 1), sums them over the corners and hands the 32
      // channels to the owning lane
#pragma unroll
      for (int kk = 0; kk < L; ++kk) {
        unsigned long long m1 = rc.mm[kk];
        if (m1 != 0) {
          const float* vl = a.vhm + ((long long)hd * S + lv.start[kk]) * DH + 2 * (lane & 15);
#pragma unroll 1
          while (m1) {
            const int bl = __builtin_ctzll(m1);
            m1 &= m1 - 1;
            const float sx = __shfl(in_cur.x[kk], bl, 64), sy = __shfl(in_cur.y[kk], bl, 64), sa = __shfl(in_cur.a[kk], bl, 64);
            const Footprint fp = footprint(lv.H[kk], lv.W[kk], sx, sy, sa);
            const int cr = lane >> 4;
            const int hc = (cr & 2) ? fp.h1 : fp.h0, wc = (cr & 1) ? fp.w1 : fp.w0;
            const float wgt = cr == 0 ? fp.w00 : cr == 1 ? fp.w01 : cr == 2 ? fp.w10 : fp.w11;
            const float2 v2 = *reinterpret_cast<const float2*>(vl + (long long)(hc * lv.W[kk] + wc) * DH);
            float tot[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {   // sum over the 4 corner rows: afterwards every row holds channels 2 (lane & 15) + e
              const float v = wgt * (e ? v2.y : v2.x);
              const t3u2 s1 = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
              const float r1 = __uint_as_float(s1.x) + __uint_as_float(s1.y);
              const t3u2 s2 = __builtin_amdgcn_permlane16_swap(__float_as_uint(r1), __float_as_uint(r1), false, false);
              tot[e] = __uint_as_float(s2.x) + __uint_as_float(s2.y);
            }
            const int orot = bl & 7;   // the owner's chunk rotation
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              const int cl = (j ^ orot) * 2;   // lane holding the first two channels of the owner's chunk slot j (uniform)
              float add[4];
#pragma unroll
              for (int e = 0; e < 4; ++e) add[e] = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(tot[e & 1]), cl + (e >> 1)));
              if (lane == bl) acc[j] += (t3v4){add[0], add[1], add[2], add[3]};
            }
          }
        }
      }
      // the next item's locations and weights from its raw projections (loaded at the top of this item)
      Inputs in_nxt;
      finish_inputs(raw_nxt, in_nxt);
      S6_STAMP(4)

      // Every load of this item -- the next tile's rows, inputs, header, the lists of the tile after it -- is waited for
      // HERE, before the output stores are issued, so that no later wait for one of them waits for the stores'
      // acknowledgements (thousands of clocks).
      __builtin_amdgcn_s_waitcnt(0x0F70);
      S6_STAMP(5)
      __syncthreads();   // A: nobody reads the rows that are about to be replaced any more
      S6_STAMP(6)
      // the LDS stores of the entering rows are issued first (branch-free) and drain -- 13 clocks of the store path each,
      // ~850 per item and CU -- under the point reduction's vector work
      if constexpr (!(S6_ABLATE & 2)) {
#pragma unroll
        for (int j = 0; j < S6_PC; ++j) commit(__builtin_amdgcn_readlane(rows.y, j), wreg[j]);
      }
      S6_STAMP(7)
      reduce_store(acc, n, m, qg_cur, qslot < hfield(hdv, HD_TOTAL));
      if constexpr (!(S6_ABLATE & 2)) {
        if (has_next) {
          const int passes = (hfield(hdv_nxt, HD_NENTER) + S6_PC - 1) / S6_PC;   // > 1 only for tall tiles
#pragma unroll 1
          for (int pass = 1; pass < passes; ++pass) {
#pragma unroll
            for (int j = 0; j < S6_PC; ++j) wreg[j] = request(vrs, __builtin_amdgcn_readlane(rows.x, pass * S6_PC + j));
            __builtin_amdgcn_s_waitcnt(0x0F70);
#pragma unroll
            for (int j = 0; j < S6_PC; ++j) commit(__builtin_amdgcn_readlane(rows.y, pass * S6_PC + j), wreg[j]);
          }
        }
      }
      S6_STAMP(8)
      __syncthreads();   // B: the next tile's rows are in place
      S6_STAMP(9)
#ifdef S6_TRACE
      if (lane == 0 && trace_serial < S6_TRACE_ITEMS && blockIdx.x < 256) {
#pragma unroll
        for (int k = 0; k < 10; ++k) g_s6_trace[(((long long)blockIdx.x * S6_TRACE_ITEMS + trace_serial) * 8 + wave) * 10 + k] = stamp[k];
      }
      ++trace_serial;
#endif
      if (!has_next) break;
      hdv = hdv_nxt;
      rows = (tile + 2 <= tlast) ? rows_n2 : (t3u2){0u, 0u};
      qg_cur = qg_nxt;
      qg_nxt = qg_n2;
      in_cur = in_nxt;
    }
  }
#undef S6_READ4
#undef S6_READ4_DEP
#undef S6_WAIT4
#undef S6_FMA4
}

// ---- host side: per-geometry tables, built once per (device, level shapes, tile parameters, planes, grid, policy); a small LRU
struct S6Key {
  int dev, L, TH, TW, R, planes, grid, policy;
  int H[UNIVS_MAX_LEVELS], W[UNIVS_MAX_LEVELS];
  bool operator==(const S6Key& o) const {
    if (dev != o.dev || L != o.L || TH != o.TH || TW != o.TW || R != o.R || planes != o.planes || grid != o.grid || policy != o.policy)
      return false;
    for (int l = 0; l < L; ++l)
      if (H[l] != o.H[l] || W[l] != o.W[l]) return false;
    return true;
  }
};
struct S6Geo {
  S6Key key;
  S6Levels lv;
  S6Tile* tiles = nullptr;     // device
  S6Piece* pieces = nullptr;   // device
  int* qtable = nullptr;       // device
  S6Seg* segs = nullptr;       // device
  int* seg_begin = nullptr;    // device
  int ntiles = 0;
  size_t lds = 0;
  bool ok = false;
  unsigned long long stamp = 0;
  GeoUse use;                  // per-stream last-launch events + the capture pin (msda_geometry.h: geo_mark_use)
};

static void s6_free(S6Geo* g) {
  if (!g) return;
  if (g->tiles) (void)hipFree(g->tiles);
  if (g->pieces) (void)hipFree(g->pieces);
  if (g->qtable) (void)hipFree(g->qtable);
  if (g->segs) (void)hipFree(g->segs);
  if (g->seg_begin) (void)hipFree(g->seg_begin);
  g->use.destroy();
  delete g;
}

template <class T>
static bool s6_upload(T** dst, const std::vector<T>& src) {
  return hipMalloc(reinterpret_cast<void**>(dst), src.size() * sizeof(T)) == hipSuccess &&
         hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice) == hipSuccess;
}

// Shared ownership + event-deferred frees: see msda_geometry.h (an evicted entry is freed once nobody holds it and the last
// launch that read its tables has completed; a cache miss during stream capture returns nullptr).
static std::shared_ptr<S6Geo> s6_geometry(const LevelTable& lv, int L, int fine, int TH, int TW, int R, int planes, int grid,
                                          int policy, hipStream_t st) {
  static std::mutex mu;
  static std::vector<std::shared_ptr<S6Geo>> cache, retired;
  static unsigned long long clock_ = 0;
  constexpr size_t CACHE_MAX = 24;
  S6Key key{};
  if (hipGetDevice(&key.dev) != hipSuccess) return nullptr;
  key.L = L; key.TH = TH; key.TW = TW; key.R = R; key.planes = planes; key.grid = grid; key.policy = policy;
  for (int l = 0; l < L; ++l) { key.H[l] = lv.H[l]; key.W[l] = lv.W[l]; }
  std::lock_guard<std::mutex> lock(mu);
  for (size_t i = 0; i < retired.size();)
    if (geo_idle(retired[i])) retired.erase(retired.begin() + i);
    else ++i;
  for (const auto& e : cache)
    if (e->key == key) { e->stamp = ++clock_; return e; }
  if (geo_capturing(st)) return nullptr;
  S6Host h;
  s6_build_host(lv, L, fine, TH, TW, R, h);
  S6Geo* g = new S6Geo();
  g->key = key; g->lv = h.lv; g->ntiles = h.ntiles; g->lds = h.lds; g->ok = h.ok && h.lds + 1024 <= (size_t)S6_LDS_MAX; g->stamp = ++clock_;
  if (g->ok) {
    std::vector<S6Seg> segs;
    std::vector<int> begin;
    // a cold start (the whole windows instead of the entering rows) priced at 1.5 tiles
    if (!s6_build_segments(planes, h.tiles_x, h.tiles_y, grid, policy, 1.5, segs, begin)) g->ok = false;
    if (g->ok && (!s6_upload(&g->tiles, h.tiles) || !s6_upload(&g->pieces, h.pieces) || !s6_upload(&g->qtable, h.qtab) ||
                  !s6_upload(&g->segs, segs) || !s6_upload(&g->seg_begin, begin))) {
      (void)hipGetLastError();
      s6_free(g);
      return nullptr;
    }
  }
  std::shared_ptr<S6Geo> sp(g, s6_free);
  if (cache.size() >= CACHE_MAX) {   // retire the least recently used geometry of THIS device (image datasets: many resolutions)
    size_t lru = cache.size();
    for (size_t i = 0; i < cache.size(); ++i)
      if (cache[i]->key.dev == key.dev && !geo_pinned(cache[i]) && (lru == cache.size() || cache[i]->stamp < cache[lru]->stamp)) lru = i;
    if (lru < cache.size()) {
      retired.push_back(cache[lru]);
      cache.erase(cache.begin() + lru);
    }
  }
  cache.push_back(sp);
  return sp;
}

template <int L>
static void launch_heads(unsigned grid, hipStream_t st, const std::shared_ptr<S6Geo>& g, const S6Args& a) {
  auto kfn = msda_fwd_heads<L>;
  const size_t lds = g->lds + 1024;   // + the dummy region masked-out lanes commit into
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(kfn, dim3(grid), dim3(64 * S6_NW), lds, st, a, g->lv, g->tiles, g->pieces, g->qtable, g->segs, g->seg_begin, (unsigned)g->lds);
}

// returns 1 if launched, 0 if preconditions do not hold (caller takes another path), <0 on error
int msda_forward_heads_f32(const float* vhm, const LevelTable& lv, const float* qhm, const float* ref,
                           long long ref_batch_stride, int N, int S, int M, int D, int L, int Lq, int P, float* out,
                           hipStream_t st) {
  if (D != 32 || P != 4 || L < 1 || L > 4 || Lq != S || M < 1) return 0;
  if ((long long)S * S6_DH * 4 >= (1LL << 31) || (long long)N * M >= (1LL << 30) || (long long)S * M * 128 >= (1LL << 32) ||
      (long long)S * P * 3 * L * 4 >= (1LL << 32))
    return 0;
  long long expect = 0;
  int fine = 0;
  for (int l = 0; l < L; ++l) {
    if (lv.start[l] != expect || lv.H[l] < 2 || lv.W[l] < 2) return 0;
    expect += (long long)lv.H[l] * lv.W[l];
    if ((long long)lv.H[l] * lv.W[l] > (long long)lv.H[fine] * lv.W[fine]) fine = l;
  }
  if (expect != S) return 0;

  const UnivsConfig cfg = config();
  const int R = cfg.msda_halo > 0 ? cfg.msda_halo : 6;
  if (R < 0 || R > 64 || cfg.msda_strip_w < 0 || cfg.msda_strip_h < 0) return 0;
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v < 1) {
      (void)hipGetLastError();
      v = 256;
    }
    n_cu = v;
  }
  const int policy = cfg.msda_sched == 1 ? 0 : 1;   // default: lockstep rounds; 1: contiguous ranges (A / B runs)
  // Tilings in order of preference; the first whose tables fit (windows + dummy region within one CU's LDS, <= S6_QCAP queries per
  // tile) runs.  16 x 6 (128 queries per tile at the benchmark pyramids: every lane of the 8 waves owns a sample; 1.9 x the level's
  // pixels fetched per column against 2.2 x at 12 x 8) measured 120 us against 126 us per launch at config 2
  // (profiles/r06_msda_heads_*).  A caller's msda_strip_w / msda_strip_h come first.
  const int cand[][2] = {{cfg.msda_strip_w > 0 ? cfg.msda_strip_w : (cfg.msda_strip_h > 0 ? 12 : 16), cfg.msda_strip_h > 0 ? cfg.msda_strip_h : (cfg.msda_strip_w > 0 ? 8 : 6)},
                         {12, 8}, {12, 6}, {12, 4}, {8, 4}, {8, 2}, {4, 2}};
  std::shared_ptr<S6Geo> g;
  for (const auto& c : cand) {
    const int TW = c[0], TH = c[1];
    const long long ntiles = (long long)((lv.H[fine] + TH - 1) / TH) * ((lv.W[fine] + TW - 1) / TW);
    const long long nb = (long long)N * M * ntiles;
    if (nb <= 0 || nb > 0x7fffffffLL) return 0;
    const int grid = (int)std::min<long long>(nb, std::max(cfg.msda_grid > 0 ? cfg.msda_grid : n_cu, 1));
    g = s6_geometry(lv, L, fine, TH, TW, R, N * M, grid, policy, st);
    if (!g) return 0;
    if (g->ok) break;
    g.reset();
  }
  if (!g) return 0;

  S6Args a{vhm, qhm, ref, ref_batch_stride, out, N, S, M};
  const unsigned grid = (unsigned)g->key.grid;
  switch (L) {
    case 1: launch_heads<1>(grid, st, g, a); break;
    case 2: launch_heads<2>(grid, st, g, a); break;
    case 3: launch_heads<3>(grid, st, g, a); break;
    default: launch_heads<4>(grid, st, g, a); break;
  }
  int rc = check_launch("msda_fwd_heads");
  geo_mark_use(g, st);
  return rc == UNIVS_OK ? 1 : rc;
}

}  // namespace univs

#ifdef S6_TRACE
extern "C" int univs_dbg_s6_trace(void* host, unsigned long long bytes) {   // (trace builds only: not in include/univs_hip.h)
  if (bytes > sizeof(g_s6_trace)) bytes = sizeof(g_s6_trace);
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_s6_trace), bytes, 0, hipMemcpyDeviceToHost);
}
#endif
