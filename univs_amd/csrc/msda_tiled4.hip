// LDS-tiled MSDA forward, fourth generation (gfx950): "strips" -- resident, row-circular windows; a lane owns a sample.
//
// Operator: ms_deform_attn_forward (ops/src/ms_deform_attn.h:25-44; kernel ms_deform_im2col_cuda.cuh:242-304,
// bilinear helper :38-89) for the encoder geometry (Lq == S, D = 32, P = 4).
//
// What the earlier generations measured (profiles/r02_msda_trace_v4.txt, r02_msda_trace_v5.txt; DESIGN.md "what actually
// bounds MSDA"): (1) every (tile, level) step restaged its whole window -- 826 MB of fills per launch for a 100-MB value
// tensor, 40 % of them L2 misses; (2) with the fills cut, the kernel is VALU-issue-bound: a gather that spreads one
// sample over 32 lanes needs the sample's address and weights broadcast to those lanes (DPP, half rate on gfx950), ~10
// issue clocks per sample and SIMD on top of ~5 for the sample's record.  This generation removes both:
//
//   * STRIPS.  A workgroup owns a contiguous range of the column-major tile sequence, i.e. it walks down a tile column,
//     and keeps ALL the levels' windows resident at once, each stored row-circularly: level row y lives in LDS row
//     y mod NR_l (window of NR_l rows x pitch_l pixels).  Moving one tile down invalidates only the rows that left the
//     window; only the rows that entered it are fetched (8 / 4 / 2 rows of the three levels for an 8-row tile): 2.2x
//     less fill traffic than restaging (370 MB per launch at the bench geometry).  LDS row NR_l duplicates LDS row 0,
//     so a sample's bottom corners are always exactly one pitch below its top corners.
//   * A LANE OWNS A SAMPLE.  Lane (point p = lane >> 4, query i = lane & 15) computes the record of ITS sample --
//     window address, four corner weights -- and gathers it alone: 4 corners x 8 ds_read_b128 (32 channels), 64 packed
//     FMAs into 32 private accumulators.  Nothing is broadcast.  A wave evaluation = 16 queries x 4 points at one level;
//     the 8 waves of a workgroup own 16 queries of the tile each and visit the L levels one after the other.
//     Bank conflicts: lanes read different pixels, so "every lane reads chunk j" would put 64 lanes on the 8 banks of
//     that chunk.  Instead lane l reads chunk j ^ ((l >> 1) & 7) and takes the left / right corner first according to
//     the pixel's parity ^ (l & 1): the 16 lanes of a pass then hit 16 distinct (pixel parity, chunk) pairs = all 64
//     banks (tools/probes/lds_gather_b128.hip: 214 B/clk/CU against 27 without the rotation).  The accumulators of a
//     lane therefore hold the channel chunks in a lane-specific order; the order is undone in the output addresses.
//   * the 4 points of a query sit in the 4 DPP rows of a wave; per item the 32 accumulators are summed over the rows
//     with 16 v_permlane32_swap + 8 v_permlane16_swap (+ 24 adds), after which row r of the wave holds 8 finished
//     channels of each query: two 16-byte stores per lane.
//   * rows entering the windows for the next tile are loaded into registers while the current tile is gathered (every
//     wave moves its share: (row, 8-pixel column block) pieces, one buffer_load_dwordx4 + one ds_write_b128 each) and
//     committed between two barriers at the end of the item.  At the top of a column the whole windows are replaced.
//   * everything a wave waits for is requested ahead: the tile's query list two items ahead, its sampling locations /
//     attention weights one item ahead, the entering rows one item ahead.
//   * samples whose footprint leaves the tile's window (halo 6: < 0.03 % at the bench geometry) are added from global
//     memory by the whole wave and handed to the owning lane (v_readlane).
//
// LDS: sum over the levels of (NR_l + 1) * pitch_l * 128 bytes (12 x 8 tiles, halo 6, three levels: 162,176 B).
#include <cstring>
#include <type_traits>

#include "msda_geometry.h"
#include "msda_tiled3_dev.h"

#ifdef UNIVS_MSDA_TRACE
// Debug builds only (tools/msda_trace3.py --gen 4): s_memtime stamps of the second item of every workgroup, wave 0:
// 0 item top, 2..4 level k evaluated (record + gather + misses + that level's requests), 5 at barrier A, 6 past it,
// 7 rows committed, 8 outputs reduced + stored and past barrier B; 28 / 29: wave 0 enters / leaves the kernel.
// The stamps of an item are kept in scalar registers and written after its last barrier: a store in the middle of the
// item would be waited for by the item's own vmcnt waits and distort what it measures.
__device__ unsigned long long g_msda_trace4[4096 * 32];
#define T4STAMP_DECL unsigned long long t4ts[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}
#define T4STAMP(cond, i)                                                                     \
  do {                                                                                       \
    __builtin_amdgcn_sched_barrier(0);                                                       \
    if (cond) t4ts[i] = __builtin_amdgcn_s_memtime();                                        \
    __builtin_amdgcn_sched_barrier(0);                                                       \
  } while (0)
#define T4STAMP_FLUSH(cond)                                                                  \
  do {                                                                                       \
    if ((cond) && (threadIdx.x & 63) == 0 && blockIdx.x < 4096)                              \
      for (int i_ = 0; i_ < 10; ++i_) g_msda_trace4[blockIdx.x * 32 + i_] = t4ts[i_];        \
  } while (0)
#define T4STAMP_NOW(cond, i)                                                                 \
  do {                                                                                       \
    if ((cond) && (threadIdx.x & 63) == 0 && blockIdx.x < 4096) g_msda_trace4[blockIdx.x * 32 + (i)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
extern "C" __attribute__((visibility("default"))) int univs_msda_trace4_read(unsigned long long* dst, int n) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_msda_trace4), sizeof(unsigned long long) * 32 * n);
}
#else
#define T4STAMP_DECL
#define T4STAMP(cond, i)
#define T4STAMP_FLUSH(cond)
#define T4STAMP_NOW(cond, i)
#endif

#ifdef UNIVS_MSDA_ABLATE_BUILD
#define T4_ABLATE(x) (x)
#else
#define T4_ABLATE(x) 0   /* UNIVS_MSDA_ABLATE needs -DUNIVS_MSDA_ABLATE_BUILD: a runtime condition around the stores costs waits */
#endif

namespace univs {

constexpr int T4_NW = 8;           // waves of a workgroup; wave w owns the queries [16 w, 16 w + 16) of an item
constexpr int T4_QCAP = 16 * T4_NW;
constexpr int T4_PC = 8;           // row pieces a wave stages in registers per pass (a regular tile needs 7)
constexpr int T4_PCAP = 40;        // row pieces per wave and list (whole windows of up to 3 x 24 rows x 4 column blocks)
constexpr int T4_ROWS_MAX = 24, T4_PITCH_MAX = 32;   // window caps
constexpr int T4_PX_BIAS = 16;

constexpr int T4_LMAX = 4;
// Per level slot (visiting order), the same for every tile: a kernel argument.
struct T4Levels {
  int H[T4_LMAX], W[T4_LMAX], start[T4_LMAX], l[T4_LMAX];
  int pitch[T4_LMAX], reg[T4_LMAX], nr[T4_LMAX];   // LDS row pitch (pixels), region byte offset, rows of the circular buffer
};
// Per tile (tile = tx * tiles_y + ty, column-major: consecutive tiles are vertical neighbours), workgroup-uniform.
struct T4Tile {
  int wx0[T4_LMAX], wy0[T4_LMAX], ww[T4_LMAX], wh[T4_LMAX];   // this tile's windows (with the zero ring)
  int rot[T4_LMAX];   // wy0 mod nr: LDS row of window row 0
  int total;                   // queries of the tile
  int n_cold;                  // pieces per wave of this tile's "whole windows" list
  int n_enter_next;            // pieces per wave of the NEXT tile's "entering rows" list (next in the sequence, wrapping)
  int pad[9];
};
static_assert(sizeof(T4Tile) == 128, "two scalar loads");
// One (row, 8-pixel column block) of one level's window: what one wave instruction moves (8 lanes x 16 B per pixel-head).
struct T4Piece {
  unsigned a;   // T4_PX_BIAS + pixel index (start + y * W + x) of the block's first pixel within the frame (24 bits) |
                // columns inside the level << 24
  unsigned b;   // byte offset of the block's first pixel in LDS (18 bits) | copy to the mirror row (LDS row nr) << 18 |
                // level slot << 19 | columns inside the window pitch << 21
};

template <int L, bool FUSED>
__global__ __launch_bounds__(64 * T4_NW) void msda_fwd_tiled4(const float* __restrict__ value, T4Levels lv,
                                                               const T4Tile* __restrict__ tiles,
                                                               const T4Piece* __restrict__ pieces,
                                                               const int* __restrict__ qtab, int ntiles, int ablate,
                                                               T3Inputs in, int N, int S, int M, float* __restrict__ out,
                                                               unsigned nitems) {
  const float* __restrict__ loc = in.loc;
  const float* __restrict__ attn = in.attn;
  ablate = T4_ABLATE(ablate);
  constexpr int D = 32, P = 4;
  extern __shared__ __attribute__((aligned(1024))) char lds4[];
  const unsigned lds_base = (unsigned)(unsigned long long)(T3_LDS char*)lds4;   // a multiple of 1024

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;

  // ---- this workgroup's range [g0, g1) of the sequence (frame, head, tile column, tile row); the workgroups of an XCD
  // are neighbours in the sequence (same head, adjacent columns: their halos overlap in that XCD's L2)
  const unsigned nxcd = min(8u, gridDim.x);
  const unsigned xcd = blockIdx.x % nxcd, widx = blockIdx.x / nxcd;
  const unsigned gq = gridDim.x / nxcd, gr = gridDim.x % nxcd;
  const unsigned lw_ = xcd * gq + min(xcd, gr) + widx;
  const unsigned g0 = (unsigned)((unsigned long long)lw_ * nitems / gridDim.x);
  const unsigned g1 = (unsigned)((unsigned long long)(lw_ + 1) * nitems / gridDim.x);
  if (g0 >= g1) return;   // uniform, before any barrier

  struct Item {   // workgroup-uniform
    long long nm;   // n * S * M + m
    int tile, n, m;
  };
  auto make_item = [&](unsigned g) __attribute__((always_inline)) {
    g = min(g, g1 - 1);   // past the range: the last tile again
    const unsigned hd = g / (unsigned)ntiles;
    const unsigned n = hd / (unsigned)M;
    // (pin the quotients to scalars or everything derived from them, buffer resources included, is treated as divergent)
    const unsigned nu = __builtin_amdgcn_readfirstlane(n), hu = __builtin_amdgcn_readfirstlane(hd);
    Item it;
    it.n = (int)nu;
    it.m = (int)(hu - nu * (unsigned)M);
    it.nm = (long long)nu * S * M + it.m;
    it.tile = (int)(g - hu * (unsigned)ntiles);
    return it;
  };
  // Tile headers and piece lists are fetched with VECTOR loads (lane k gets dword k / piece k) an item ahead and read
  // with v_readlane where needed: a scalar load would have to be waited for on the spot (~300 clocks each, a dozen per
  // item), and the scalar registers to hold them ahead of time do not exist.
  auto header = [&](const Item& it) __attribute__((always_inline)) {
    return reinterpret_cast<const int*>(tiles + it.tile)[lane & 31];
  };
  enum { HD_WX0 = 0, HD_WY0 = T4_LMAX, HD_WW = 2 * T4_LMAX, HD_WH = 3 * T4_LMAX, HD_ROT = 4 * T4_LMAX, HD_TOTAL = 5 * T4_LMAX,
         HD_NCOLD = 5 * T4_LMAX + 1, HD_NENTER_NEXT = 5 * T4_LMAX + 2 };
  auto hfield = [&](int hdv, int idx) __attribute__((always_inline)) { return __builtin_amdgcn_readlane(hdv, idx); };

  // =========================== moving rows ===========================
  // Host-built lists of (row, column block) pieces per (tile, wave): the rows ENTERING the windows when the strip
  // arrives from the tile above (list 0; the whole windows at the top of a column) and the WHOLE windows (list 1, for
  // the first tile of a workgroup's range).  A piece is one buffer_load_dwordx4 and one or two ds_write_b128.
  const int lane8 = tid & 7;
  const int pstride = M * D * 4;
  // (pixel indices are stored with a bias of T4_PX_BIAS: the first pixel of a block may lie left of the level)
  const unsigned lanepart_g = (unsigned)(((lane >> 3) - T4_PX_BIAS) * pstride + lane8 * 16), lanepart_l = (unsigned)((lane >> 3) * 128 + lane8 * 16);
  const int lanebit = 1 << (lane >> 3);
  t3v4 wreg[T4_PC];
  static_assert(T4_PCAP <= 64 && T4_PCAP % T4_PC == 0, "a wave fetches its piece list with one load; whole passes");
  typedef unsigned pcvec __attribute__((ext_vector_type(2)));
  auto piece_list = [&](const Item& it, int which) __attribute__((always_inline)) {   // lane k: piece k of my list
    const T4Piece* p = pieces + ((long long)(it.tile * 2 + which) * T4_NW + wave) * T4_PCAP + min(lane, T4_PCAP - 1);
    return *reinterpret_cast<const pcvec*>(p);
  };
  // pieces [pass * T4_PC, ...) of the list -> registers
  // Pass `pass` of a list: pieces [pass * T4_PC, +T4_PC).  No bounds: the host pads every list with no-op pieces (no
  // column inside the level: the load returns 0 without touching memory; no column inside the pitch: nothing is stored).
  // A condition around a load -- even a uniform one -- makes hipcc wait for ALL outstanding loads before it.
  auto load_rows = [&](const pcvec& list, const Item& it, int pass) __attribute__((always_inline)) {
    // one buffer resource over this (frame, head)'s value rows; masked-out columns / rows of the zero ring get an offset
    // outside it and read 0
    const unsigned long long pv = (unsigned long long)(value + it.nm * D);
    const unsigned plo = __builtin_amdgcn_readfirstlane((unsigned)pv), phi = __builtin_amdgcn_readfirstlane((unsigned)(pv >> 32));
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<float*>(((unsigned long long)phi << 32) | plo), 0, (int)(((long long)S - 1) * M * D * 4 + D * 4), 0x00020000);
#pragma unroll
    for (int j = 0; j < T4_PC; ++j) {
      const unsigned pa = __builtin_amdgcn_readlane(list.x, pass * T4_PC + j);
      const unsigned off = ((pa >> 24) & lanebit) ? (pa & 0xffffffu) * (unsigned)pstride + lanepart_g : 0x80000000u;
      wreg[j] = __builtin_bit_cast(t3v4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0));
    }
  };
  auto commit_rows = [&](const pcvec& list, int pass, auto steady) __attribute__((always_inline)) {
    // the rows have arrived (in the steady state they were waited for before the item's output stores were issued)
    if constexpr (!decltype(steady)::value) __builtin_amdgcn_s_waitcnt(0x0F70);
#pragma unroll
    for (int j = 0; j < T4_PC; ++j) {
      const unsigned pb = __builtin_amdgcn_readlane(list.y, pass * T4_PC + j);
      if ((pb >> 21) & lanebit) {
        T3_LDS char* dst = (T3_LDS char*)lds4 + lanepart_l + (pb & 0x3ffffu);
        *(T3_LDS t3v4*)dst = wreg[j];
        if (pb & (1u << 18)) {   // window row that lives in LDS row 0: also into the mirror row nr
          const int slot = (pb >> 19) & 3;
          int wrapb = lv.nr[0] * lv.pitch[0];
#pragma unroll
          for (int t = 1; t < L; ++t) wrapb = slot == t ? lv.nr[t] * lv.pitch[t] : wrapb;
          *(T3_LDS t3v4*)(dst + wrapb * 128) = wreg[j];
        }
      }
    }
  };

  // =========================== gathering ===========================
  const int qi = lane & 15, pt = lane >> 4;          // my sample: query qi of the wave's 16, point pt
  const unsigned rot8 = (unsigned)(lane >> 1) & 7u;   // my chunk rotation
  const int qslot = wave * 16 + qi;                   // my query's index within an item

  // (the host pads the tile's query list with its last query: no clamp, no dependent scalar load)
  auto my_query = [&](const Item& it) __attribute__((always_inline)) { return qtab[it.tile * T4_QCAP + qslot]; };
  struct Inputs { float x[L], y[L], a[L]; };
  auto load_inputs = [&](const Item& it, int qg, Inputs& iv) __attribute__((always_inline)) {
    if constexpr (!FUSED) {
      const char* lb = reinterpret_cast<const char*>(loc + it.nm * (L * P * 2));
      const char* ab = reinterpret_cast<const char*>(attn + it.nm * (L * P));
#pragma unroll
      for (int kk = 0; kk < L; ++kk) {
        const unsigned e = (unsigned)(qg * (M * L * P) + lv.l[kk] * P + pt);
        const float2 xy = *reinterpret_cast<const float2*>(lb + e * 8u);
        iv.x[kk] = xy.x; iv.y[kk] = xy.y;
        iv.a[kk] = *reinterpret_cast<const float*>(ab + e * 4u);
      }
    } else {
      // raw projections -> locations / weights (csrc/msda_prepare.hip's arithmetic; the softmax sums in a different
      // order: per point over the levels, then over the 4 points)
      const float* row = in.proj + ((long long)it.n * S + qg) * in.row_stride;
      const float* ref0 = in.ref + it.n * in.ref_batch_stride;
      float lg[L];
#pragma unroll
      for (int kk = 0; kk < L; ++kk) {
        const int l = lv.l[kk];
        const float2 off = *reinterpret_cast<const float2*>(row + it.m * (L * P * 2) + (l * P + pt) * 2);
        const float2 rp2 = *reinterpret_cast<const float2*>(ref0 + ((long long)qg * L + l) * 2);
        lg[kk] = row[in.n_off + it.m * (L * P) + l * P + pt];
        iv.x[kk] = rp2.x + off.x / (float)lv.W[kk];
        iv.y[kk] = rp2.y + off.y / (float)lv.H[kk];
      }
      // the 4 points of a query sit in the 4 DPP rows: all-rows max / sum with two swap rounds
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
        iv.a[kk] = expf(lg[kk] - mx);
        sum += iv.a[kk];
      }
      sum = all_rows(sum, false);
#pragma unroll
      for (int kk = 0; kk < L; ++kk) iv.a[kk] = iv.a[kk] / sum;
    }
  };

  // The levels' sizes as floats, held in VGPRs on purpose: uniform, but an SGPR source operand halves the issue rate of
  // the fp32 instructions that consume it (profiles/r02_gfx950_issue_costs.txt).
  float Hf[L], Wf[L];
#pragma unroll
  for (int kk = 0; kk < L; ++kk) {
    Hf[kk] = (float)lv.H[kk]; Wf[kk] = (float)lv.W[kk];
    asm volatile("" : "+v"(Hf[kk]), "+v"(Wf[kk]));
  }

  // ---- prologue: the whole windows of the first tile (a cold start), the first two query lists, the first inputs
  T4STAMP_NOW(wave == 0, 28);
  Item cur = make_item(g0);
  int hdv = header(cur);
  int qg_cur = my_query(cur);
  int qg_nxt = my_query(make_item(g0 + 1));
  Inputs in_cur;
  load_inputs(cur, qg_cur, in_cur);
  {
    const pcvec list = piece_list(cur, 1);
    const int n_cold = hfield(hdv, HD_NCOLD);
    const int passes = (n_cold + T4_PC - 1) / T4_PC;
#pragma unroll 1
    for (int pass = 0; pass < passes; ++pass) {
      load_rows(list, cur, pass);
      commit_rows(list, pass, std::false_type{});
    }
  }
  pcvec rows = piece_list(make_item(g0 + 1), 0);   // the rows entering the next tile's windows
  __builtin_amdgcn_s_waitcnt(0x0F70);   // (see the wait before the output stores)
  __syncthreads();

#pragma unroll 1
  for (unsigned g = g0;; ++g) {
    const bool has_next = g + 1 < g1;
    T4STAMP_DECL;
    T4STAMP(g == g0 + 1 && wave == 0, 0);
    const Item nxt = make_item(g + 1);
    // ---- 0. requests for the next items are spread over the level loop below: the vector-memory pipe takes >= 16 clocks
    // per wave instruction, 17 instructions x 8 waves = 2.2k clocks per item -- issued in one go at the top of the item
    // by all waves they were a phase of their own during which the LDS idled (profiles/r02_msda_trace_v5.txt); issued
    // between the levels they drain while the waves gather.
    Inputs in_nxt;
    int hdv_nxt = 0, qg_n2 = 0;
    pcvec rows_n2 = {0u, 0u};
    // (query slots past the tile's last query repeat that query -- the host pads the list -- and store the same values to
    // the same place: no "valid" predicate anywhere, in particular not around the stores, whose count the compiler's
    // waitcnt bookkeeping must know exactly or every later wait for a load also waits for them)

    t3v4 acc[8];   // my sample's 32 channels, chunk slot j = channel chunk j ^ rot8; summed over the levels
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = (t3v4){0.f, 0.f, 0.f, 0.f};

#pragma unroll
    for (int kk = 0; kk < L; ++kk) {
      // ---- A. my sample's record at this level (reference arithmetic: ms_deform_im2col_cuda.cuh:285-293 and :38-89; the
      // window includes the one-pixel zero ring around the level, so out-of-level corners simply read zeros)
      const float him = in_cur.y[kk] * Hf[kk] - 0.5f, wim = in_cur.x[kk] * Wf[kk] - 0.5f;
      const float hf = floorf(him), wf = floorf(wim);
      // the band (-1, H) x (-1, W) as |v - centre| < radius with centre = (H - 1) / 2, radius = (H + 1) / 2; false for
      // NaN / inf like the reference's four compares
      const bool inband = fabsf(fmaf(Hf[kk], -0.5f, him) + 0.5f) < fmaf(Hf[kk], 0.5f, 0.5f) &&
                          fabsf(fmaf(Wf[kk], -0.5f, wim) + 0.5f) < fmaf(Wf[kk], 0.5f, 0.5f);
      const int r0 = (int)hf - hfield(hdv, HD_WY0 + kk), c0 = (int)wf - hfield(hdv, HD_WX0 + kk);
      const bool inwin = (unsigned)r0 < (unsigned)(hfield(hdv, HD_WH + kk) - 1) && (unsigned)c0 < (unsigned)(hfield(hdv, HD_WW + kk) - 1);
      const bool use = inband && inwin;
      const bool miss = inband && !inwin && in_cur.a[kk] != 0.f;
      // (a sample that must not contribute still reads: it points at the window's first pixel, which is always staged;
      // its weights are exact zeros -- selects, not products, so that a NaN location contributes nothing)
      const float lh = use ? him - hf : 0.f, lw = use ? wim - wf : 0.f, aw = use ? in_cur.a[kk] : 0.f;
      int rl = (use ? r0 : 0) + hfield(hdv, HD_ROT + kk);
      rl -= rl >= lv.nr[kk] ? lv.nr[kk] : 0;
      const unsigned tl = lds_base + (unsigned)lv.reg[kk] + (unsigned)((rl * lv.pitch[kk] + (use ? c0 : 0)) * 128);
      const unsigned fs = ((tl >> 7) ^ (unsigned)lane) & 1u;   // which corner column I read first
      const unsigned a0 = (tl + fs * 128u) | (rot8 << 4), a1 = (tl + 128u - fs * 128u) | (rot8 << 4);
      const unsigned rowb = (unsigned)(lv.pitch[kk] * 128);
      const float f0 = fs ? lw : 1.f - lw;
      const float wb = aw * lh, wt = aw - wb;
      const float wt0 = wt * f0, wt1 = wt - wt0, wb0 = wb * f0, wb1 = wb - wb0;

      // ---- B. gather: 4 corners x 8 chunks
      if (!(ablate & 4)) {
#define T4_CORNER(ADDR, WGT)                                                                      \
  {                                                                                               \
    const t3v2 w2 = {WGT, WGT};                                                                   \
    _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                               \
      const t3v4 d = *(const T3_LDS t3v4*)(unsigned long long)((ADDR) ^ (unsigned)(j << 4));      \
      const t3v2 lo = __builtin_elementwise_fma(w2, (t3v2){d.x, d.y}, (t3v2){acc[j].x, acc[j].y}); \
      const t3v2 hi = __builtin_elementwise_fma(w2, (t3v2){d.z, d.w}, (t3v2){acc[j].z, acc[j].w}); \
      acc[j] = (t3v4){lo.x, lo.y, hi.x, hi.y};                                                    \
    }                                                                                             \
  }
        T4_CORNER(a0, wt0)
        T4_CORNER(a1, wt1)
        T4_CORNER(a0 + rowb, wb0)
        T4_CORNER(a1 + rowb, wb1)
#undef T4_CORNER
      }

      // ---- C. rare: samples whose footprint leaves the tile's window -> the whole wave fetches the four corners from
      // global memory (lane = corner lane >> 4, channels 2 (lane & 15) and + 1), sums them over the corners and hands
      // the 32 channels to the owning lane
      unsigned long long mm = __ballot(miss);
      if (mm != 0 && !(ablate & 8)) {
        const float* vl = value + (cur.nm + (long long)lv.start[kk] * M) * D + (lane & 15) * 2;
#pragma unroll 1
        while (mm) {
          const int bl = __builtin_ctzll(mm);
          mm &= mm - 1;
          const float sx = __shfl(in_cur.x[kk], bl, 64), sy = __shfl(in_cur.y[kk], bl, 64), sa = __shfl(in_cur.a[kk], bl, 64);
          const Footprint fp = footprint(lv.H[kk], lv.W[kk], sx, sy, sa);
          const int cr = lane >> 4;
          const int hc = (cr & 2) ? fp.h1 : fp.h0, wc = (cr & 1) ? fp.w1 : fp.w0;
          const float wgt = cr == 0 ? fp.w00 : cr == 1 ? fp.w01 : cr == 2 ? fp.w10 : fp.w11;
          const t3v2 gv = *reinterpret_cast<const t3v2*>(vl + (long long)(hc * lv.W[kk] + wc) * (M * D));
          const float vx = wgt * gv.x, vy = wgt * gv.y;
          // sum over the 4 corner rows: afterwards rows 0, 1 hold channel 2 k, rows 2, 3 channel 2 k + 1 (lane k of the row)
          const t3u2 s1 = __builtin_amdgcn_permlane32_swap(__float_as_uint(vx), __float_as_uint(vy), false, false);
          const float r1 = __uint_as_float(s1.x) + __uint_as_float(s1.y);
          const t3u2 s2 = __builtin_amdgcn_permlane16_swap(__float_as_uint(r1), __float_as_uint(r1), false, false);
          const float tot = __uint_as_float(s2.x) + __uint_as_float(s2.y);
          const int orot = (bl >> 1) & 7;   // the owner's chunk rotation
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const int cbase = (j ^ orot) * 4;   // first channel of the owner's chunk slot j (uniform)
            float add[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int ch_ = cbase + e;
              add[e] = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(tot), (ch_ & 1) * 32 + (ch_ >> 1)));
            }
            if (lane == bl) acc[j] += (t3v4){add[0], add[1], add[2], add[3]};
          }
        }
      }
      // ---- requests, one slice per level (fenced: the scheduler would hoist the loads to the top of the item)
      __builtin_amdgcn_sched_barrier(0);
      if (kk == 0) {
        load_inputs(nxt, qg_nxt, in_nxt);   // the next item's inputs (its query list was fetched an item ago)
      }
      if (kk == (L > 1 ? 1 : 0)) {
        // the rows entering the next tile's windows (first pass; their piece list was fetched an item ago).  After the
        // last tile of the range the same rows are written once more: identical data, and nobody reads them.
        load_rows(rows, nxt, 0);
      }
      if (kk == L - 1) {
        hdv_nxt = header(nxt);              // the next item's header; the lists of the item after it
        const Item nn = make_item(g + 2);
        qg_n2 = my_query(nn);
        rows_n2 = piece_list(nn, 0);
      }
      T4STAMP(g == g0 + 1 && wave == 0 && kk < 3, 2 + kk);
    }

    // ---- D. sum the 4 points (DPP rows) of every query and store (before the barrier: it overlaps with the other waves' last
    // gathers, and the stores are acknowledged by the time the next item waits for its loads): after the two swap rounds row r of the wave holds
    // the finished chunk slots 2 r and 2 r + 1 of each query = channel chunks (2 r) ^ rot8 and (2 r + 1) ^ rot8
    T4STAMP(g == g0 + 1 && wave == 0, 5);
    __syncthreads();   // A: nobody reads the rows that are about to be replaced any more
    T4STAMP(g == g0 + 1 && wave == 0, 6);
    // Every load of this item -- the next tile's rows, inputs, header, the lists of the tile after it -- is waited for
    // HERE, before the output stores are issued, so that no later wait for one of them (the compiler's are conservative
    // across the loop back-edge: vmcnt(0)) waits for the stores' acknowledgements, which take thousands of clocks.
    __builtin_amdgcn_s_waitcnt(0x0F70);
    commit_rows(rows, 0, std::true_type{});
    {
      const int passes = (hfield(hdv, HD_NENTER_NEXT) + T4_PC - 1) / T4_PC;   // > 1 only at the top of a tile column
#pragma unroll 1
      for (int pass = 1; pass < passes; ++pass) {
        load_rows(rows, nxt, pass);
        commit_rows(rows, pass, std::false_type{});
      }
    }
    T4STAMP(g == g0 + 1 && wave == 0, 7);

    if (!(ablate & 16)) {
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
      float* orow = out + (cur.nm + (long long)qg_cur * M) * D;
      const unsigned ca = (unsigned)(2 * pt) ^ rot8, cb2 = (unsigned)(2 * pt + 1) ^ rot8;
      *reinterpret_cast<t3v4*>(orow + ca * 4) = (t3v4){t8[0], t8[1], t8[2], t8[3]};
      *reinterpret_cast<t3v4*>(orow + cb2 * 4) = (t3v4){t8[4], t8[5], t8[6], t8[7]};
    }
    __syncthreads();   // B: the next tile's rows are in place
    T4STAMP(g == g0 + 1 && wave == 0, 8);
    T4STAMP_FLUSH(g == g0 + 1 && wave == 0);
    if (!has_next) break;
    cur = nxt;
    hdv = hdv_nxt;
    rows = rows_n2;
    qg_cur = qg_nxt;
    qg_nxt = qg_n2;
    in_cur = in_nxt;
  }
  T4STAMP_NOW(wave == 0, 29);
}

// ---- host side: per-geometry table, built once per (device, level shapes, tile parameters)
struct T4Key {
  int dev, L, TH, TW, R;
  int H[UNIVS_MAX_LEVELS], W[UNIVS_MAX_LEVELS];
  bool operator==(const T4Key& o) const {
    if (dev != o.dev || L != o.L || TH != o.TH || TW != o.TW || R != o.R) return false;
    for (int l = 0; l < L; ++l)
      if (H[l] != o.H[l] || W[l] != o.W[l]) return false;
    return true;
  }
};
struct T4Geo {
  T4Key key;
  T4Levels lv;
  T4Tile* tiles;     // device [ntiles]
  T4Piece* pieces;   // device [ntiles][2][T4_NW][T4_PCAP]: list 0 = entering rows, list 1 = whole windows
  int* qtable;       // device [ntiles][T4_QCAP]: global query index of the tile's i-th query (padded with the last one)
  int ntiles;
  long long qmax;    // max queries of a tile
  size_t lds;        // bytes of all the levels' circular windows
  bool ok;           // the piece lists fit
};

static inline int pos_mod(int a, int b) { return ((a % b) + b) % b; }

static const T4Geo* t4_geometry(const LevelTable& lv, int L, int fine, int TH, int TW, int R) {
  static std::mutex mu;
  static std::vector<T4Geo*> cache;
  T4Key key{};
  if (hipGetDevice(&key.dev) != hipSuccess) return nullptr;
  key.L = L; key.TH = TH; key.TW = TW; key.R = R;
  for (int l = 0; l < L; ++l) { key.H[l] = lv.H[l]; key.W[l] = lv.W[l]; }
  std::lock_guard<std::mutex> lock(mu);
  for (const T4Geo* e : cache)
    if (e->key == key) return e;

  const int tiles_y = (lv.H[fine] + TH - 1) / TH, tiles_x = (lv.W[fine] + TW - 1) / TW;
  std::vector<int4> ax((size_t)L * tiles_x), ay((size_t)L * tiles_y);
  int pitch[UNIVS_MAX_LEVELS] = {0, 0, 0, 0}, nr[UNIVS_MAX_LEVELS] = {0, 0, 0, 0};
  for (int l = 0; l < L; ++l) {
    // windows with the zero ring, capped; samples beyond go through the global fallback
    int mw = 2, mh = 2;
    for (int tx = 0; tx < tiles_x; ++tx) {
      int4& e = ax[(size_t)l * tiles_x + tx];
      axis_entry(tx, tiles_x, TW, lv.W[l], lv.W[fine], R, T4_PITCH_MAX, /*ring=*/1, e);
      mw = std::max(mw, e.w);
    }
    for (int ty = 0; ty < tiles_y; ++ty) {
      int4& e = ay[(size_t)l * tiles_y + ty];
      axis_entry(ty, tiles_y, TH, lv.H[l], lv.H[fine], R, T4_ROWS_MAX, /*ring=*/1, e);
      mh = std::max(mh, e.w);
    }
    pitch[l] = mw;
    nr[l] = mh;
  }
  int ord[UNIVS_MAX_LEVELS];
  for (int l = 0; l < L; ++l) ord[l] = l;
  std::sort(ord, ord + L, [&](int a, int b) { return (long long)lv.H[a] * lv.W[a] > (long long)lv.H[b] * lv.W[b]; });
  T4Geo* g = new T4Geo();
  g->key = key;
  g->ntiles = tiles_y * tiles_x;
  g->qmax = 0;
  g->ok = true;
  size_t lds = 0;
  for (int kk = 0; kk < L; ++kk) {
    const int l = ord[kk];
    g->lv.H[kk] = lv.H[l]; g->lv.W[kk] = lv.W[l]; g->lv.start[kk] = lv.start[l]; g->lv.l[kk] = l;
    g->lv.pitch[kk] = pitch[l]; g->lv.nr[kk] = nr[l]; g->lv.reg[kk] = (int)lds;
    lds += (size_t)(nr[l] + 1) * pitch[l] * 128;
  }
  g->lds = lds;
  std::vector<T4Tile> tiles((size_t)g->ntiles);
  std::vector<T4Piece> pcs((size_t)g->ntiles * 2 * T4_NW * T4_PCAP, T4Piece{0u, 0u});
  std::vector<int> n_enter((size_t)g->ntiles, 0);
  std::vector<int> qtab((size_t)g->ntiles * T4_QCAP, 0);
  for (int tx = 0; tx < tiles_x; ++tx)
    for (int ty = 0; ty < tiles_y; ++ty) {
      const size_t tile = (size_t)tx * tiles_y + ty;
      int pre[UNIVS_MAX_LEVELS + 1] = {0};
      for (int l = 0; l < L; ++l) pre[l + 1] = pre[l] + ax[(size_t)l * tiles_x + tx].y * ay[(size_t)l * tiles_y + ty].y;
      g->qmax = std::max<long long>(g->qmax, pre[L]);
      if (pre[L] >= 1 && pre[L] <= T4_QCAP) {
        int last = 0;
        for (int l = 0; l < L; ++l) {
          const int4 gx = ax[(size_t)l * tiles_x + tx], gy = ay[(size_t)l * tiles_y + ty];
          for (int i = 0; i < gx.y * gy.y; ++i)
            qtab[tile * T4_QCAP + pre[l] + i] = last = lv.start[l] + (gy.x + i / gx.y) * lv.W[l] + gx.x + i % gx.y;
        }
        for (int i = pre[L]; i < T4_QCAP; ++i) qtab[tile * T4_QCAP + i] = last;
      }
      T4Tile& t = tiles[tile];
      std::memset(&t, 0, sizeof(t));
      t.total = pre[L];
      for (int which = 0; which < 2; ++which) {   // 0: entering rows, 1: whole windows
        int count = 0;
        for (int kk = 0; kk < L; ++kk) {
          const int l = ord[kk];
          const int4 gx = ax[(size_t)l * tiles_x + tx], gy = ay[(size_t)l * tiles_y + ty];
          t.wx0[kk] = gx.z; t.wy0[kk] = gy.z; t.ww[kk] = gx.w; t.wh[kk] = gy.w; t.rot[kk] = pos_mod(gy.z, nr[l]);
          int y0 = gy.z, n = gy.w;
          if (which == 0 && ty > 0) {
            const int4 py = ay[(size_t)l * tiles_y + ty - 1];
            y0 = std::max(gy.z, py.z + py.w);
            n = std::max(0, gy.z + gy.w - y0);
          }
          for (int r = 0; r < n; ++r) {
            const int y = y0 + r, ym = pos_mod(y, nr[l]);
            for (int b8 = 0; b8 * 8 < pitch[l]; ++b8) {
              T4Piece pc;
              int px = lv.start[l] + y * lv.W[l] + gx.z + 8 * b8;
              const int ldsoff = g->lv.reg[kk] + (ym * pitch[l] + 8 * b8) * 128;
              int ldmask = 0, stmask = 0;
              for (int t8 = 0; t8 < 8; ++t8) {
                const int cx = 8 * b8 + t8, x = gx.z + cx;
                if (cx < pitch[l]) stmask |= 1 << t8;
                if (cx < pitch[l] && y >= 0 && y < lv.H[l] && x >= 0 && x < lv.W[l]) ldmask |= 1 << t8;
              }
              px = ldmask ? px + T4_PX_BIAS : 0;
              if (px < 0 || px >= (1 << 24) || ldsoff >= (1 << 18)) g->ok = false;
              pc.a = ((unsigned)px & 0xffffffu) | ((unsigned)ldmask << 24);
              pc.b = (unsigned)ldsoff | ((ym == 0 ? 1u : 0u) << 18) | ((unsigned)kk << 19) | ((unsigned)stmask << 21);
              const int w = count % T4_NW, j = count / T4_NW;
              if (j < T4_PCAP) pcs[((tile * 2 + which) * T4_NW + w) * T4_PCAP + j] = pc;
              else g->ok = false;
              ++count;
            }
          }
        }
        const int per_wave = (count + T4_NW - 1) / T4_NW;
        if (which == 0) n_enter[tile] = per_wave;
        else t.n_cold = per_wave;
      }
    }
  for (size_t tile = 0; tile < (size_t)g->ntiles; ++tile) tiles[tile].n_enter_next = n_enter[(tile + 1) % g->ntiles];
  if (hipMalloc(reinterpret_cast<void**>(&g->tiles), tiles.size() * sizeof(T4Tile)) != hipSuccess ||
      hipMemcpy(g->tiles, tiles.data(), tiles.size() * sizeof(T4Tile), hipMemcpyHostToDevice) != hipSuccess ||
      hipMalloc(reinterpret_cast<void**>(&g->pieces), pcs.size() * sizeof(T4Piece)) != hipSuccess ||
      hipMemcpy(g->pieces, pcs.data(), pcs.size() * sizeof(T4Piece), hipMemcpyHostToDevice) != hipSuccess ||
      hipMalloc(reinterpret_cast<void**>(&g->qtable), qtab.size() * sizeof(int)) != hipSuccess ||
      hipMemcpy(g->qtable, qtab.data(), qtab.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) {
    (void)hipGetLastError();
    delete g;
    return nullptr;
  }
  cache.push_back(g);
  return g;
}

template <int L, bool FUSED>
static void launch_tiled4(unsigned grid, unsigned nitems, hipStream_t st, const float* value, const T4Geo* g, int ablate,
                          const T3Inputs& in, int N, int S, int M, float* out) {
  auto kfn = msda_fwd_tiled4<L, FUSED>;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)g->lds);
  hipLaunchKernelGGL(kfn, dim3(grid), dim3(64 * T4_NW), g->lds, st, value, g->lv, g->tiles, g->pieces, g->qtable, g->ntiles, ablate,
                     in, N, S, M, out, nitems);
}

// returns 1 if launched, 0 if preconditions do not hold (caller tries the next implementation), <0 on error
static int t4_forward(const float* value, const LevelTable& lv, const T3Inputs& in, bool fused, int N, int S, int M, int D,
                      int L, int Lq, int P, float* out, hipStream_t st) {
  if (D != 32 || P != 4 || L < 1 || L > 4 || Lq != S || M < 1) return 0;
  if ((long long)S * M * D * 4 >= (1LL << 30) || (long long)S * M * L * P * 8 >= (1LL << 31)) return 0;
  if (fused && (long long)S * in.row_stride * 4 >= (1LL << 31)) return 0;
  long long expect = 0;
  int fine = 0;
  for (int l = 0; l < L; ++l) {
    if (lv.start[l] != expect || lv.H[l] < 2 || lv.W[l] < 2) return 0;
    expect += (long long)lv.H[l] * lv.W[l];
    if ((long long)lv.H[l] * lv.W[l] > (long long)lv.H[fine] * lv.W[fine]) fine = l;
  }
  if (expect != S) return 0;

  const int TH = env_int("UNIVS_MSDA_TILE4_H", 8), TW = env_int("UNIVS_MSDA_TILE4_W", 12);
  const int R = env_int("UNIVS_MSDA_HALO", 6);
  const int ablate = env_int("UNIVS_MSDA_ABLATE", 0);
  if (TH < 1 || TW < 1 || R < 0 || R > 64) return 0;
  const T4Geo* g = t4_geometry(lv, L, fine, TH, TW, R);
  if (!g || !g->ok || g->qmax > T4_QCAP || g->qmax < 1 || g->lds > 160 * 1024) return 0;

  const long long nb = (long long)N * M * g->ntiles;
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
  const unsigned grid = (unsigned)std::min<long long>(nb, std::max(env_int("UNIVS_MSDA_GRID", n_cu), 1));
#define T4_LAUNCH(LL)                                                                                   \
  if (fused) launch_tiled4<LL, true>(grid, (unsigned)nb, st, value, g, ablate, in, N, S, M, out);       \
  else launch_tiled4<LL, false>(grid, (unsigned)nb, st, value, g, ablate, in, N, S, M, out);
  switch (L) {
    case 1: T4_LAUNCH(1) break;
    case 2: T4_LAUNCH(2) break;
    case 3: T4_LAUNCH(3) break;
    default: T4_LAUNCH(4) break;
  }
#undef T4_LAUNCH
  int rc = check_launch("msda_fwd_tiled4");
  return rc == UNIVS_OK ? 1 : rc;
}

int msda_forward_tiled4_f32(const float* value, const LevelTable& lv, const float* loc, const float* attn, int N,
                            int S, int M, int D, int L, int Lq, int P, float* out, hipStream_t st) {
  T3Inputs in{};
  in.loc = loc;
  in.attn = attn;
  return t4_forward(value, lv, in, false, N, S, M, D, L, Lq, P, out, st);
}

int msda_forward_fused_tiled4_f32(const float* value, const LevelTable& lv, const float* proj, int row_stride, int n_off,
                                  const float* ref, long long ref_batch_stride, int N, int S, int M, int D, int L, int Lq,
                                  int P, float* out, hipStream_t st) {
  T3Inputs in{};
  in.proj = proj;
  in.ref = ref;
  in.row_stride = row_stride;
  in.n_off = n_off;
  in.ref_batch_stride = ref_batch_stride;
  return t4_forward(value, lv, in, true, N, S, M, D, L, Lq, P, out, st);
}

}  // namespace univs
