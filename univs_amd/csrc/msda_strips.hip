// LDS-tiled MSDA forward, fifth generation (gfx950): "strips" at HALF a head per workgroup, two workgroups per CU.
//
// Operator: the core of MSDeformAttn.forward (ops/modules/ms_deform_attn.py:100-121): softmax over the L*P logits,
// sampling locations = reference + offset / (W_l, H_l), and ms_deform_attn_forward (ops/src/ms_deform_attn.h:25-44; kernel
// ms_deform_im2col_cuda.cuh:242-304, bilinear helper :38-89) for the encoder geometry (Lq == S, D = 32, P = 4), fed with the
// RAW projections -- the location / weight tensors never exist.
//
// What generation 4 (resident row-circular windows, a lane owns a sample; removed in round 3) measured: the gather proper
// is 4.2k of a 10.5k-clock item; the rest is phase structure -- vector-memory requests, LDS commits and LDS gathers are
// pipe-bound phases separated by two workgroup barriers, and one workgroup of 160 KB per CU has nobody to overlap them
// with.  This generation keeps the strips and the lane-owns-sample gather and changes what a workgroup is:
//
//   * HALF A HEAD per workgroup (16 channels, 64 bytes per pixel): windows are half as large, TWO workgroups fit a CU
//     (<= 80 KB each) and run their phases independently -- while one waits at a barrier or for its rows, the other
//     gathers.  Items are (frame, head, half, tile); the two halves of a head are neighbours in the item sequence.
//   * HEAD-MAJOR operands, written that way by the producing Linears' epilogues (linear_split.hip, LS_EPI_BLOCKED):
//     value [N][M][2][S][16] -- a window row is one contiguous burst, a 16-pixel piece is one wave instruction -- and the
//     projections [N][M][S][P][3L] = per (query, head, point): L offset pairs, then L logits -- a lane's inputs are 12 L
//     contiguous bytes (three loads for L = 3) instead of six strided 4/8-byte loads into a 1152-byte row shared with the
//     seven other heads.
//   * LDS "super-pixels" (msda_strips_geom.h): 128 bytes = pixel x of two consecutive rows, super-rows circular.  The 16-byte slot of a read
//     within the 256-byte bank row is (x parity, row parity, chunk): a lane visits its four corners in the order
//     (dx, dy) = (fs ^ k0, ft ^ k1) with fs = x parity ^ lane bit 0, ft = row parity ^ lane bit 1, and chunk j ^ lane bits
//     2-3, so the 16 lanes of every ds_read_b128 group (their low four lane bits are all different) hit 16 different
//     slots whatever pixels they sample.
//   * the 4 points of a query sit in the 4 DPP rows of a wave; the 16 accumulators are summed over the rows with
//     8 v_permlane32_swap + 4 v_permlane16_swap, after which row r holds the finished chunk slot r of each query: one
//     16-byte store per lane.
//   * rows entering the windows for the next tile are loaded into registers while the current tile is gathered and
//     committed between two barriers at the end of the item; headers / query lists / piece lists are fetched with vector
//     loads ahead of time and read with v_readlane; samples whose footprint leaves the window are added from global
//     memory by the whole wave (as in generation 4).
#include <cstring>
#include <memory>
#include <mutex>
#include <type_traits>

#include "msda_strips_geom.h"
#include "config.h"
#include "msda_dev.h"

namespace univs {

typedef float s5v4u __attribute__((ext_vector_type(4), aligned(4)));   // a 16-byte load that is only 4-byte aligned
typedef unsigned s5u3 __attribute__((ext_vector_type(3)));   // a piece's three used dwords

struct S5Args {
  const float* vhm;    // value, head-major halves [N][M][2][S][16]
  const float* qhm;    // projections, head-major [N][M][S][P][3 L]: L offset pairs (x, y), then L logits; levels in SLOT order
                       // (largest level first: S5Levels.l[kk] is the caller's index of slot kk)
  const float* ref;    // reference points [N or 1][S][2]: one per query, the same for every level (the encoder's pixel centres)
  long long ref_batch_stride;   // floats between frames; 0: one set for all frames
  float* out;          // [N][S][M * 32]
  int N, S, M;
};

template <int L>
__global__ __launch_bounds__(64 * S5_NW, 4) void msda_fwd_strips(S5Args a, S5Levels lv, const S5Tile* __restrict__ tiles,
                                                                  const S5Piece* __restrict__ pieces,
                                                                  const int* __restrict__ qtab, int ntiles, unsigned nitems) {
  constexpr int P = 4, DH = S5_DH;
  extern __shared__ __attribute__((aligned(1024))) char lds5[];
  const unsigned lds_base = (unsigned)(unsigned long long)(T3_LDS char*)lds5;

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int S = a.S, M = a.M;

  // ---- this workgroup's range [g0, g1) of the sequence (frame, head, half, tile column, tile row); the workgroups of an
  // XCD are neighbours in the sequence (same head, adjacent columns / the other half: their operands share that XCD's L2)
  const unsigned nxcd = min(8u, gridDim.x);
  const unsigned xcd = blockIdx.x % nxcd, widx = blockIdx.x / nxcd;
  const unsigned gq = gridDim.x / nxcd, gr = gridDim.x % nxcd;
  const unsigned lw_ = xcd * gq + min(xcd, gr) + widx;
  const unsigned g0 = (unsigned)((unsigned long long)lw_ * nitems / gridDim.x);
  const unsigned g1 = (unsigned)((unsigned long long)(lw_ + 1) * nitems / gridDim.x);
  if (g0 >= g1) return;   // uniform, before any barrier

  struct Item {   // workgroup-uniform
    int tile, n, m, half;
    unsigned hd;   // (n * M + m) * 2 + half
  };
  auto make_item = [&](unsigned g) __attribute__((always_inline)) {
    g = min(g, g1 - 1);   // past the range: the last tile again
    const unsigned hd = __builtin_amdgcn_readfirstlane(g / (unsigned)ntiles);
    const unsigned nm = hd >> 1;
    const unsigned n = __builtin_amdgcn_readfirstlane(nm / (unsigned)M);
    Item it;
    it.hd = hd;
    it.n = (int)n;
    it.m = (int)(nm - n * (unsigned)M);
    it.half = (int)(hd & 1u);
    it.tile = (int)(g - hd * (unsigned)ntiles);
    return it;
  };
  auto header = [&](const Item& it) __attribute__((always_inline)) {
    return reinterpret_cast<const int*>(tiles + it.tile)[lane & 15];
  };
  enum { HD_P0 = 0, HD_P1 = S5_LMAX, HD_TOTAL = 2 * S5_LMAX, HD_NCOLD = 2 * S5_LMAX + 1, HD_NENTER_NEXT = 2 * S5_LMAX + 2 };
  auto hfield = [&](int hdv, int idx) __attribute__((always_inline)) { return __builtin_amdgcn_readlane(hdv, idx); };

  // =========================== moving rows ===========================
  // A piece = 16 pixels of one row of one level: lane (pixel = lane >> 2, chunk = lane & 3) moves 16 bytes.
  const int lpx = lane >> 2, lch = lane & 3;
  const unsigned lanepart_g = (unsigned)((lpx - S5_PX_BIAS) * (DH * 4) + lch * 16), lanepart_l = (unsigned)(lpx * 128 + lch * 16);
  const unsigned lanebit = 1u << lpx;
  t3v4 wreg[S5_PC];
  static_assert(S5_PCAP <= 64 && S5_PCAP % S5_PC == 0, "a wave fetches its piece list with one load; whole passes");
  auto piece_list = [&](const Item& it, int which) __attribute__((always_inline)) {   // lane k: piece k of my list
    const S5Piece* p = pieces + ((long long)(it.tile * 2 + which) * S5_NW + wave) * S5_PCAP + min(lane, S5_PCAP - 1);
    return *reinterpret_cast<const s5u3*>(p);
  };
  // Pass `pass` of a list: pieces [pass * S5_PC, +S5_PC).  No bounds: the host pads every list with no-op pieces (no column
  // inside the level: the load returns 0 without touching memory; no column inside the pitch: nothing is stored).
  auto load_rows = [&](const s5u3& list, const Item& it, int pass) __attribute__((always_inline)) {
    // one buffer resource over this (frame, head, half)'s value pixels; masked-out columns get an offset outside it -> 0
    const unsigned long long pv = (unsigned long long)(a.vhm + (long long)it.hd * S * DH);
    const unsigned plo = __builtin_amdgcn_readfirstlane((unsigned)pv), phi = __builtin_amdgcn_readfirstlane((unsigned)(pv >> 32));
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<float*>(((unsigned long long)phi << 32) | plo), 0, (int)((long long)S * DH * 4), 0x00020000);
#pragma unroll
    for (int j = 0; j < S5_PC; ++j) {
      const unsigned pa = __builtin_amdgcn_readlane(list.x, pass * S5_PC + j);
      const unsigned pc = __builtin_amdgcn_readlane(list.z, pass * S5_PC + j);
      const unsigned off = (pc & lanebit) ? (pa & 0xffffffu) * (unsigned)(DH * 4) + lanepart_g : 0x80000000u;
      wreg[j] = __builtin_bit_cast(t3v4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0));
    }
  };
  auto commit_rows = [&](const s5u3& list, int pass, auto steady) __attribute__((always_inline)) {
    // the rows have arrived (in the steady state they were waited for before the item's output stores were issued)
    if constexpr (!decltype(steady)::value) __builtin_amdgcn_s_waitcnt(0x0F70);
#pragma unroll
    for (int j = 0; j < S5_PC; ++j) {
      const unsigned pb = __builtin_amdgcn_readlane(list.y, pass * S5_PC + j);
      const unsigned pc = __builtin_amdgcn_readlane(list.z, pass * S5_PC + j);
      if ((pc >> 16) & lanebit) *(T3_LDS t3v4*)((T3_LDS char*)lds5 + lanepart_l + pb) = wreg[j];
    }
  };

  // =========================== gathering ===========================
  const int qi = lane & 15, pt = lane >> 4;           // my sample: query qi of the wave's 16, point pt
  const unsigned rot4 = ((unsigned)lane >> 2) & 3u;    // my chunk rotation
  const int qslot = wave * 16 + qi;                    // my query's index within an item

  auto my_query = [&](const Item& it) __attribute__((always_inline)) { return qtab[it.tile * S5_QCAP + qslot]; };
  // The levels' sizes as floats (uniform; scalar operands: the 128-register budget has no room for vector copies)
  float Hf[L], Wf[L];
#pragma unroll
  for (int kk = 0; kk < L; ++kk) { Hf[kk] = (float)lv.H[kk]; Wf[kk] = (float)lv.W[kk]; }

  struct Inputs { float x[L], y[L], a[L]; };
  struct RawInputs { float v[3 * L]; float2 rp; };
  // my (query, head, point)'s 3 L floats -- L offset pairs then L logits (slot order) -- and the query's reference point:
  // loads only; `finish_inputs` does the arithmetic an item later, when the loads have long arrived
  auto load_raw = [&](const Item& it, int qg, RawInputs& r) __attribute__((always_inline)) {
    // uniform 64-bit bases + 32-bit lane offsets (host-checked: the projections of one (frame, head) stay below 4 GB)
    const char* rowb = reinterpret_cast<const char*>(a.qhm + ((long long)it.n * M + it.m) * S * (P * 3 * L));
    const char* refb = reinterpret_cast<const char*>(a.ref + it.n * a.ref_batch_stride);
    const unsigned ro = ((unsigned)qg * P + (unsigned)pt) * (unsigned)(3 * L * 4);
    const float* row = reinterpret_cast<const float*>(rowb + ro);
    r.rp = *reinterpret_cast<const float2*>(refb + (unsigned)qg * 8u);
    if constexpr (L == 3) {
      const s5v4u r0 = *reinterpret_cast<const s5v4u*>(row), r1 = *reinterpret_cast<const s5v4u*>(row + 4);
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
      // tools/strips_emulate.cpp compares it with the division on every sample)
      const float qx = r.v[2 * kk] * lv.rW[kk], qy = r.v[2 * kk + 1] * lv.rH[kk];
      // (the remainders as single FMAs: hipcc packs the pair with (W, H) in swapped halves -- the form common.h describes)
      const float ox = fmaf(fnma_single(qx, Wf[kk], r.v[2 * kk]), lv.rW[kk], qx);
      const float oy = fmaf(fnma_single(qy, Hf[kk], r.v[2 * kk + 1]), lv.rH[kk], qy);
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

  // ---- prologue: the whole windows of the first tile (a cold start), the first two query lists, the first inputs
  Item cur = make_item(g0);
  int hdv = header(cur);
  int qg_cur = my_query(cur);
  int qg_nxt = my_query(make_item(g0 + 1));
  Inputs in_cur;
  {
    RawInputs r0;
    load_raw(cur, qg_cur, r0);
    finish_inputs(r0, in_cur);
  }
  {
    const s5u3 list = piece_list(cur, 1);
    const int n_cold = hfield(hdv, HD_NCOLD);
    const int passes = (n_cold + S5_PC - 1) / S5_PC;
#pragma unroll 1
    for (int pass = 0; pass < passes; ++pass) {
      load_rows(list, cur, pass);
      commit_rows(list, pass, std::false_type{});
    }
  }
  s5u3 rows = piece_list(make_item(g0 + 1), 0);   // the rows entering the next tile's windows
  __builtin_amdgcn_s_waitcnt(0x0F70);   // (see the wait before the output stores)
  __syncthreads();

#pragma unroll 1
  for (unsigned g = g0;; ++g) {
    const bool has_next = g + 1 < g1;
    const Item nxt = make_item(g + 1);
    // ---- 0. everything the NEXT item needs is requested here, ahead of this item's gathers (~2k clocks of cover): its
    // inputs (the query list was fetched an item ago), the rows entering its windows (first pass; the piece list was
    // fetched an item ago; after the last tile of the range the same rows are written once more: identical data, nobody
    // reads them), its header, and the lists of the item after it.  Issued behind a level's gather instead, the last
    // slice would be waited for at the barrier a few hundred clocks later.
    RawInputs raw_nxt;
    load_raw(nxt, qg_nxt, raw_nxt);
    load_rows(rows, nxt, 0);
    int hdv_nxt = 0, qg_n2 = 0;
    s5u3 rows_n2 = {0u, 0u, 0u};
    __builtin_amdgcn_sched_barrier(0);

    t3v4 acc[4];   // my sample's 16 channels, chunk slot j = channel chunk j ^ rot4; summed over the levels
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = (t3v4){0.f, 0.f, 0.f, 0.f};

#pragma unroll
    for (int kk = 0; kk < L; ++kk) {
      if (kk == L - 1) {   // (ahead of the last level's gather: ~700 clocks of cover, L2 hits)
        hdv_nxt = header(nxt);
        const Item nn = make_item(g + 2);
        qg_n2 = my_query(nn);
        rows_n2 = piece_list(nn, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      // ---- A. my sample's record at this level (msda_strips_geom.h: s5_record, shared with the host emulator)
      const S5Rec rec = s5_record(in_cur.x[kk], in_cur.y[kk], in_cur.a[kk], Hf[kk], Wf[kk], (unsigned)hfield(hdv, HD_P0 + kk),
                                  (unsigned)hfield(hdv, HD_P1 + kk), lv.nsr[kk], lv.pitch[kk], lv.next_d[kk], lv.wrap_d[kk],
                                  lds_base + (unsigned)lv.reg[kk], (unsigned)lane & 15u);

      // ---- B. gather: 4 corners x 4 chunks, two corners per trip.  The 4 reads of a corner are one asm statement: left to hipcc at
      // this register budget (four waves per SIMD) every read is waited for before the next is issued (same destination
      // registers), and fully unrolled it issues all 16 reads of the level first (64 data registers: spills).
#define S5_FMA4(D, WGT, J)                                                                        \
  {                                                                                               \
    const t3v2 w2_ = {WGT, WGT};                                                                  \
    const t3v2 lo = __builtin_elementwise_fma(w2_, (t3v2){D.x, D.y}, (t3v2){acc[J].x, acc[J].y}); \
    const t3v2 hi = __builtin_elementwise_fma(w2_, (t3v2){D.z, D.w}, (t3v2){acc[J].z, acc[J].w}); \
    acc[J] = (t3v4){lo.x, lo.y, hi.x, hi.y};                                                      \
  }
      {
        unsigned ca = rec.a[0], cb = rec.a[1];
        float wa = rec.w[0], wb = rec.w[1];
#pragma unroll 1
        for (int hh = 0; hh < 2; ++hh) {
#define S5_READ4(A, D0, D1, D2, D3)                                                                                      \
  asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %5\n\tds_read_b128 %2, %6\n\tds_read_b128 %3, %7\n\ts_waitcnt lgkmcnt(0)" \
               : "=&v"(D0), "=&v"(D1), "=&v"(D2), "=&v"(D3)                                                              \
               : "v"(A), "v"((A) ^ 16u), "v"((A) ^ 32u), "v"((A) ^ 48u)                                                   \
               : "memory")
          t3v4 d0, d1, d2, d3;
          S5_READ4(ca, d0, d1, d2, d3);
          S5_FMA4(d0, wa, 0) S5_FMA4(d1, wa, 1) S5_FMA4(d2, wa, 2) S5_FMA4(d3, wa, 3)
          S5_READ4(cb, d0, d1, d2, d3);
          S5_FMA4(d0, wb, 0) S5_FMA4(d1, wb, 1) S5_FMA4(d2, wb, 2) S5_FMA4(d3, wb, 3)
#undef S5_READ4
          ca = rec.a[2]; cb = rec.a[3]; wa = rec.w[2]; wb = rec.w[3];
        }
      }
#undef S5_FMA4

      // ---- C. rare: samples whose footprint leaves the tile's window -> the whole wave fetches the four corners from
      // global memory (lane = corner lane >> 4, channel lane & 15), sums them over the corners and hands the 16 channels
      // to the owning lane
      unsigned long long mm = __ballot(!rec.inwin);
      if (mm != 0) mm = __ballot(!rec.inwin && in_cur.a[kk] != 0.f && s5_inband(in_cur.x[kk], in_cur.y[kk], Hf[kk], Wf[kk]));
      if (mm != 0) {
        const float* vl = a.vhm + ((long long)cur.hd * S + lv.start[kk]) * DH + (lane & 15);
#pragma unroll 1
        while (mm) {
          const int bl = __builtin_ctzll(mm);
          mm &= mm - 1;
          const float sx = __shfl(in_cur.x[kk], bl, 64), sy = __shfl(in_cur.y[kk], bl, 64), sa = __shfl(in_cur.a[kk], bl, 64);
          const Footprint fp = footprint(lv.H[kk], lv.W[kk], sx, sy, sa);
          const int cr = lane >> 4;
          const int hc = (cr & 2) ? fp.h1 : fp.h0, wc = (cr & 1) ? fp.w1 : fp.w0;
          const float wgt = cr == 0 ? fp.w00 : cr == 1 ? fp.w01 : cr == 2 ? fp.w10 : fp.w11;
          const float v = wgt * vl[(long long)(hc * lv.W[kk] + wc) * DH];
          // sum over the 4 corner rows: afterwards every row holds channel (lane & 15)
          const t3u2 s1 = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
          const float r1 = __uint_as_float(s1.x) + __uint_as_float(s1.y);
          const t3u2 s2 = __builtin_amdgcn_permlane16_swap(__float_as_uint(r1), __float_as_uint(r1), false, false);
          const float tot = __uint_as_float(s2.x) + __uint_as_float(s2.y);
          const int orot = (bl >> 2) & 3;   // the owner's chunk rotation
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int cbase = (j ^ orot) * 4;   // first channel of the owner's chunk slot j (uniform)
            float add[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) add[e] = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(tot), cbase + e));
            if (lane == bl) acc[j] += (t3v4){add[0], add[1], add[2], add[3]};
          }
        }
      }
    }
    // the next item's locations and weights from its raw projections (loaded at the top of this item)
    Inputs in_nxt;
    finish_inputs(raw_nxt, in_nxt);

    __syncthreads();   // A: nobody reads the rows that are about to be replaced any more
    // Every load of this item -- the next tile's rows, inputs, header, the lists of the tile after it -- is waited for
    // HERE, before the output stores are issued, so that no later wait for one of them waits for the stores'
    // acknowledgements (thousands of clocks).
    __builtin_amdgcn_s_waitcnt(0x0F70);
    commit_rows(rows, 0, std::true_type{});
    {
      const int passes = (hfield(hdv, HD_NENTER_NEXT) + S5_PC - 1) / S5_PC;   // > 1 only at the top of a tile column
#pragma unroll 1
      for (int pass = 1; pass < passes; ++pass) {
        load_rows(rows, nxt, pass);
        commit_rows(rows, pass, std::false_type{});
      }
    }

    // ---- D. sum the 4 points (DPP rows) of every query and store: after the two swap rounds row r of the wave holds the
    // finished chunk slot r of each query = channel chunk r ^ rot4
    {
      float a16[16];
#pragma unroll
      for (int j = 0; j < 4; ++j) { a16[4 * j] = acc[j].x; a16[4 * j + 1] = acc[j].y; a16[4 * j + 2] = acc[j].z; a16[4 * j + 3] = acc[j].w; }
      float s8[8], t4[4];
#pragma unroll
      for (int f = 0; f < 8; ++f) {
        const t3u2 sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(a16[f]), __float_as_uint(a16[f + 8]), false, false);
        s8[f] = __uint_as_float(sw.x) + __uint_as_float(sw.y);
      }
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        const t3u2 sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(s8[f]), __float_as_uint(s8[f + 4]), false, false);
        t4[f] = __uint_as_float(sw.x) + __uint_as_float(sw.y);
      }
      char* ob = reinterpret_cast<char*>(a.out + ((long long)cur.n * S * M + cur.m) * 32 + cur.half * DH);   // uniform
      const unsigned oo = (unsigned)qg_cur * (unsigned)(M * 128) + (((unsigned)pt ^ rot4) << 4);               // < 2^32 (host-checked)
      *reinterpret_cast<t3v4*>(ob + oo) = (t3v4){t4[0], t4[1], t4[2], t4[3]};
    }
    __syncthreads();   // B: the next tile's rows are in place
    if (!has_next) break;
    cur = nxt;
    hdv = hdv_nxt;
    rows = rows_n2;
    qg_cur = qg_nxt;
    qg_nxt = qg_n2;
    in_cur = in_nxt;
  }
}

// ---- host side: per-geometry tables, built once per (device, level shapes, tile parameters); a small LRU
struct S5Key {
  int dev, L, TH, TW, R;
  int H[UNIVS_MAX_LEVELS], W[UNIVS_MAX_LEVELS];
  bool operator==(const S5Key& o) const {
    if (dev != o.dev || L != o.L || TH != o.TH || TW != o.TW || R != o.R) return false;
    for (int l = 0; l < L; ++l)
      if (H[l] != o.H[l] || W[l] != o.W[l]) return false;
    return true;
  }
};
struct S5Geo {
  S5Key key;
  S5Levels lv;
  S5Tile* tiles = nullptr;     // device
  S5Piece* pieces = nullptr;   // device
  int* qtable = nullptr;       // device
  int ntiles = 0;
  size_t lds = 0;
  bool ok = false;
  unsigned long long stamp = 0;
  GeoUse use;                  // per-stream last-launch events + the capture pin (msda_geometry.h: geo_mark_use)
};

static void s5_free(S5Geo* g) {
  if (!g) return;
  if (g->tiles) (void)hipFree(g->tiles);
  if (g->pieces) (void)hipFree(g->pieces);
  if (g->qtable) (void)hipFree(g->qtable);
  g->use.destroy();
  delete g;
}

// Shared ownership + event-deferred frees: see msda_geometry.h (an evicted entry is freed once nobody holds it and the last
// launch that read its tables has completed; a cache miss during stream capture returns nullptr).
static std::shared_ptr<S5Geo> s5_geometry(const LevelTable& lv, int L, int fine, int TH, int TW, int R, hipStream_t st) {
  static std::mutex mu;
  static std::vector<std::shared_ptr<S5Geo>> cache, retired;
  static unsigned long long clock_ = 0;
  constexpr size_t CACHE_MAX = 24;
  S5Key key{};
  if (hipGetDevice(&key.dev) != hipSuccess) return nullptr;
  key.L = L; key.TH = TH; key.TW = TW; key.R = R;
  for (int l = 0; l < L; ++l) { key.H[l] = lv.H[l]; key.W[l] = lv.W[l]; }
  std::lock_guard<std::mutex> lock(mu);
  for (size_t i = 0; i < retired.size();)
    if (geo_idle(retired[i])) retired.erase(retired.begin() + i);
    else ++i;
  for (const auto& e : cache)
    if (e->key == key) { e->stamp = ++clock_; return e; }
  if (geo_capturing(st)) return nullptr;
  S5Host h;
  s5_build_host(lv, L, fine, TH, TW, R, h);
  S5Geo* g = new S5Geo();
  g->key = key; g->lv = h.lv; g->ntiles = h.ntiles; g->lds = h.lds; g->ok = h.ok; g->stamp = ++clock_;
  if (h.ok) {
    if (hipMalloc(reinterpret_cast<void**>(&g->tiles), h.tiles.size() * sizeof(S5Tile)) != hipSuccess ||
        hipMemcpy(g->tiles, h.tiles.data(), h.tiles.size() * sizeof(S5Tile), hipMemcpyHostToDevice) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&g->pieces), h.pieces.size() * sizeof(S5Piece)) != hipSuccess ||
        hipMemcpy(g->pieces, h.pieces.data(), h.pieces.size() * sizeof(S5Piece), hipMemcpyHostToDevice) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&g->qtable), h.qtab.size() * sizeof(int)) != hipSuccess ||
        hipMemcpy(g->qtable, h.qtab.data(), h.qtab.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) {
      (void)hipGetLastError();
      s5_free(g);
      return nullptr;
    }
  }
  std::shared_ptr<S5Geo> sp(g, s5_free);
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
static void launch_strips(unsigned grid, unsigned nitems, hipStream_t st, const std::shared_ptr<S5Geo>& g, const S5Args& a) {
  auto kfn = msda_fwd_strips<L>;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)g->lds);
  hipLaunchKernelGGL(kfn, dim3(grid), dim3(64 * S5_NW), g->lds, st, a, g->lv, g->tiles, g->pieces, g->qtable, g->ntiles, nitems);
}

// returns 1 if launched, 0 if preconditions do not hold (caller takes another path), <0 on error
int msda_forward_strips_f32(const float* vhm, const LevelTable& lv, const float* qhm, const float* ref,
                            long long ref_batch_stride, int N, int S, int M, int D, int L, int Lq, int P, float* out,
                            hipStream_t st) {
  if (D != 32 || P != 4 || L < 1 || L > 4 || Lq != S || M < 1) return 0;
  if ((long long)S * S5_DH * 4 >= (1LL << 31) || (long long)N * M * 2 >= (1LL << 30) || (long long)S * M * 128 >= (1LL << 32) ||
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
  const int TW = cfg.msda_strip_w > 0 ? cfg.msda_strip_w : 12, R = cfg.msda_halo > 0 ? cfg.msda_halo : 6;
  int TH = cfg.msda_strip_h > 0 ? cfg.msda_strip_h : 8;
  if (TH < 1 || TW < 1 || R < 0 || R > 64) return 0;
  std::shared_ptr<S5Geo> g;
  for (; TH >= 2; TH -= 2) {   // the windows of two workgroups must fit one CU's LDS
    g = s5_geometry(lv, L, fine, TH, TW, R, st);
    if (!g) return 0;
    if (g->ok && g->lds <= (size_t)S5_LDS_MAX) break;
    g.reset();
  }
  if (!g) return 0;

  const long long nb = (long long)N * M * 2 * g->ntiles;
  if (nb <= 0 || nb > 0x7fffffffLL) return 0;
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v < 1) {
      (void)hipGetLastError();
      v = 256;
    }
    n_cu = v;
  }
  const unsigned grid = (unsigned)std::min<long long>(nb, std::max(cfg.msda_grid > 0 ? cfg.msda_grid : 4 * n_cu, 1));
  S5Args a{vhm, qhm, ref, ref_batch_stride, out, N, S, M};
  switch (L) {
    case 1: launch_strips<1>(grid, (unsigned)nb, st, g, a); break;
    case 2: launch_strips<2>(grid, (unsigned)nb, st, g, a); break;
    case 3: launch_strips<3>(grid, (unsigned)nb, st, g, a); break;
    default: launch_strips<4>(grid, (unsigned)nb, st, g, a); break;
  }
  int rc = check_launch("msda_fwd_strips");
  geo_mark_use(g, st);
  return rc == UNIVS_OK ? 1 : rc;
}

}  // namespace univs
