// LDS-tiled MSDA forward, third generation (gfx950): sample records that never leave the register file, gathers by
// DPP-row pairs, fill waves that only move windows.
//
// Operator: ms_deform_attn_forward (ops/src/ms_deform_attn.h:25-44; kernel ms_deform_im2col_cuda.cuh:242-304,
// bilinear helper :38-89) for the encoder geometry (Lq == S, D = 32, P = 4).  Decomposition as in msda_tiled2.hip:
// persistent workgroups (one per CU) walk (frame, tile, head) items inside their XCD's chunk; an item is a
// TW x TH tile of the finest level plus the queries of the coarser levels above it, visited level by level
// ("steps"); the window [tile box +- halo] of the level's value map for this head is staged in LDS, two regions
// alternate by step parity, one workgroup barrier per step.  What changed against the second generation, and why
// (profiles/r01_msda_pmc_v3.txt: 48 % of the wave cycles parked at barriers / waitcnts, 6 of 16 waves building
// records, 1/3 of the LDS read cycles spent on 16-byte sample records):
//
//   * Records stay in registers.  Every gather wave builds the records of the samples it gathers and hands them to
//     the gathering lanes with DPP `row_newbcast` folded into the consuming instruction:
//         v_add_u32_dpp   addr,  bcast_k(slot),   lane_offset        (x2: top / bottom row)
//         ds_read_b64     d,     addr                                (x2)
//         v_fmac_f32_dpp  acc,   bcast_k(weight), d                  (x4)
//     A DPP row (16 lanes) owns one corner COLUMN of a sample (left or right pixel: 32 channels x 2 rows, 8 bytes
//     per lane and row); rows 2i / 2i+1 of a wave work on the same sample, so the 32 lanes of one ds_read_b64
//     lane group read two horizontally adjacent pixels = 256 contiguous bytes = every LDS bank once (conflict-free
//     by construction: the bank is (addr / 4) mod 64 for b64).  Lane k of a row holds the record {slot, w_top,
//     w_bottom} of sample k of its row pair (4 queries x 4 points per level step) for ITS corner column.  No record
//     traffic through LDS, no record barrier, 3 VALU per sample and row pair.
//   * Windows are clipped to the level (no zero ring): out-of-level bilinear corners are folded into the corner
//     weights (csrc/msda_tiled3_record.h), so every staged pixel is real data.
//   * The fill waves (NP of them) do nothing but move windows: buffer loads of step s+2 into registers, ds_write of
//     step s+1, while the NG gather waves work on step s.  LDS-DMA (`global_load_lds_dwordx4`) was measured for the
//     fills first and rejected: ~13 B/clk/CU, i.e. an 84-KB window takes 2.6 us -- fine for an HBM-bound weight
//     stream, 4x too slow for halo windows that are 85 % L2 hits (profiles/r02_msda_kbench_v1.txt).
//   * Sampling locations / attention weights are prefetched TWO steps ahead (they are the HBM-latency-bound
//     stream of the operator: 36 % of its bytes, touched exactly once).
//   * Samples whose footprint leaves the staged window are rare (halo 6: < 0.03 % at the bench geometry); their
//     record weights are 0 and a wave-uniform slow path adds them from global memory.
//
// LDS holds windows only (two regions).
#include <type_traits>

#include "msda_geometry.h"
#include "msda_tiled3_record.h"
#include "msda_tiled3_dev.h"

#ifdef UNIVS_MSDA_TRACE
// Debug builds only (tools/msda_trace3.py): s_memtime stamps of the second item of every workgroup.
// slots 0..14: gather wave 0, per step k: 5k + {0 top, 1 records built, 2 gathers done, 3 at barrier, 4 past barrier};
// slots 16..27: fill wave 0, per step k: 16 + 4k + {0 top, 1 committed, 2 loads issued, 3 past barrier}
__device__ unsigned long long g_msda_trace3[4096 * 32];
#define T3STAMP(cond, i)                                                                                  \
  do {                                                                                                    \
    if ((cond) && (threadIdx.x & 63) == 0 && blockIdx.x < 4096) g_msda_trace3[blockIdx.x * 32 + (i)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
extern "C" __attribute__((visibility("default"))) int univs_msda_trace3_read(unsigned long long* dst, int n) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_msda_trace3), sizeof(unsigned long long) * 32 * n);
}
#else
#define T3STAMP(cond, i)
#endif

namespace univs {

constexpr int T3_NP = 4;          // fill waves: 32 copy octets (8 lanes x 16 B = one pixel-head) as an 8 x 4 grid
constexpr int T3_OX = 8, T3_OY = 4;
constexpr int T3_PITCH_MAX = 32;  // window width cap (pixels), a multiple of T3_OX
constexpr int T3_ROWS_MAX = 24;   // window height cap, a multiple of T3_OY

// One step of one tile, everything the kernel needs as workgroup-uniform scalars, precomputed on the host: the first
// version derived these per step on the scalar unit (selects over the level table, prefix sums, divisions) and was
// SALU-bound -- 15 waves x ~300 scalar instructions per step on the CU's single scalar pipe (profiles/r02_msda_kbench_v2.txt).
// Table index: (tile * 2 + item parity) * L + step; 64 bytes = one s_load_dwordx16.
struct T3Entry {
  int H, W, start, l;           // the level visited at this step
  int wx0, wy0, ww, wh;         // staged window (may include the zero ring around the level)
  int pitch, reg, qfirst, qcount; // LDS row pitch of the window (pixels, a multiple of 8); LDS byte offset of the region;
                                  // this level's queries within the item
  int qx0, qy0, qnx, total;     // query box origin / width; queries of the whole item
};
static_assert(sizeof(T3Entry) == 64, "one scalar load");

// NP fill waves (waves 0 .. NP-1), NG gather waves, NB record batches per gather wave and step (a batch = 2 row
// pairs x 4 queries x 4 points); 8 * NB * NG query slots per item.
template <int L, int NP, int NG, int NB, bool FUSED>
__global__ __launch_bounds__(64 * (NP + NG)) void msda_fwd_tiled3(const float* __restrict__ value,
                                                                   const T3Entry* __restrict__ tab,
                                                                   const int* __restrict__ qtab, int qcap, int ntiles,
                                                                   int ablate, T3Inputs in, int N, int S, int M,
                                                                   float* __restrict__ out, unsigned nitems) {
  const float* __restrict__ loc = in.loc;
  const float* __restrict__ attn = in.attn;
  static_assert(L >= 2, "inputs are prefetched two steps ahead: at most one item ahead needs L >= 2");
  constexpr int D = 32, P = 4;
  extern __shared__ __attribute__((aligned(1024))) char lds3[];
  const unsigned lds_base = (unsigned)(unsigned long long)(T3_LDS char*)lds3;

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;

  // ---- this workgroup's items (XCD-chunked, fixed stride; see msda_tiled.hip)
  const unsigned nxcd = min(8u, gridDim.x);
  const unsigned xcd = blockIdx.x % nxcd, widx = blockIdx.x / nxcd;
  const unsigned nw = gridDim.x / nxcd + (xcd < gridDim.x % nxcd ? 1u : 0u);
  const unsigned cq = nitems / nxcd, cr = nitems % nxcd;
  const unsigned cbase = xcd < cr ? xcd * (cq + 1) : cr * (cq + 1) + (xcd - cr) * cq;
  const unsigned csize = cq + (xcd < cr ? 1u : 0u);
  if (widx >= csize) return;   // uniform, before any barrier

  struct Item {   // workgroup-uniform
    long long nm;   // n * S * M + m
    int tile, n, m;
  };
  auto make_item = [&](unsigned idx) __attribute__((always_inline)) {
    const unsigned item = cbase + idx;
    const unsigned tm = item / (unsigned)M;
    const unsigned n = tm / (unsigned)ntiles;
    // (the divisions run on the vector ALU; pin the results to scalars -- they are workgroup-uniform -- or everything
    // derived from them, buffer resources included, is treated as divergent: waterfall loops around every load)
    const unsigned nu = __builtin_amdgcn_readfirstlane(n), tmu = __builtin_amdgcn_readfirstlane(tm);
    Item it;
    it.n = (int)nu;
    it.m = (int)(item - tmu * (unsigned)M);
    it.nm = (long long)nu * S * M + it.m;
    it.tile = (int)(tmu - nu * (unsigned)ntiles);
    return it;
  };
  // the step's scalars: one 64-byte scalar load
  auto entry = [&](const Item& it, int par, int kk) __attribute__((always_inline)) {
    return tab[__builtin_amdgcn_readfirstlane((it.tile * 2 + par) * L + kk)];
  };

  if (wave < NP) {
    // =========================== fill waves ===========================
    // 32 copy octets as an 8 x 4 grid that tiles the window: octet (ox, oy) stages the pixels (oy + 4 vy, ox + 8 vx).
    // The per-lane part of every address is fixed for a step and the (vy, vx) part is uniform, so a staged row costs
    // one full-rate add; there is no per-pixel row / column arithmetic (the first version spent ~20 issue clocks per
    // staged row on it).  LDS rows have a pitch that is a multiple of 8 pixels and the regions hold whole 4-row groups,
    // so the writes need no bounds either: pad pixels receive garbage nobody reads.
    static_assert(NP == T3_NP, "octet grid is laid out for 4 fill waves");
    constexpr int VX = T3_PITCH_MAX / T3_OX, VY = T3_ROWS_MAX / T3_OY;
    const int lane8 = tid & 7, oct = tid >> 3;
    const int ox = oct & (T3_OX - 1), oy = oct >> 3;
    t3v4 wreg[VY][VX];
    // One pass over the staged rows does BOTH jobs of a step: row group vy of the window held in registers (`p`, loaded
    // one step ago) goes to LDS, and the same registers are refilled with row group vy of the next window (`n`).  The LDS
    // write path (13 clk per ds_write_b128) and the address path of the loads (16 clk per 1-KiB buffer_load_dwordx4) then
    // run side by side; done one after the other they cost 1.2k + 1.5k clocks per step and made the fill waves the
    // critical path (profiles/r02_msda_trace_v4.txt).  do_commit / do_load are workgroup-uniform.
    // Commit and load are UNCONDITIONAL (only the prologue skips the commit, at compile time): behind a runtime condition
    // -- even a uniform one like "there is a next item" -- hipcc's waitcnt pass puts `s_waitcnt vmcnt(0)` in front of
    // every load and every write and the step's loads return one by one.  After the last item the fill waves therefore
    // stage one more window of the same tile into the region nobody reads any more.
    auto commit_and_load = [&](auto do_commit_c, const T3Entry& p, const Item& it, const T3Entry& n)
                               __attribute__((always_inline)) {
      constexpr bool do_commit = decltype(do_commit_c)::value;
      // (explicitly scalar: left to itself hipcc keeps this descriptor in VGPRs and wraps every load in a waterfall loop)
      const unsigned long long pv = (unsigned long long)(value + (it.nm + (long long)n.start * M) * D);
      const unsigned plo = __builtin_amdgcn_readfirstlane((unsigned)pv), phi = __builtin_amdgcn_readfirstlane((unsigned)(pv >> 32));
      const int nrec = __builtin_amdgcn_readfirstlane((int)(((long long)n.H * n.W - 1) * M * D * 4 + D * 4));
      const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
          reinterpret_cast<float*>(((unsigned long long)phi << 32) | plo), 0, nrec, 0x00020000);
      const int pstride = M * D * 4;
      // columns: out-of-level ones (the zero ring; pad columns past the level's edge) get an offset far outside the
      // resource, which makes the load return 0; rows outside the level fall out of it by themselves
      unsigned cb[VX];
#pragma unroll
      for (int vx = 0; vx < VX; ++vx) {
        const int gx = n.wx0 + ox + vx * T3_OX;
        cb[vx] = (unsigned)gx < (unsigned)n.W ? (unsigned)(gx * pstride + lane8 * 16) : 0xC0000000u;
      }
      unsigned rowoff = (unsigned)((n.wy0 + oy) * n.W * pstride);
      unsigned rowstep = (unsigned)(T3_OY * n.W * pstride);
      T3_LDS t3v4* win = (T3_LDS t3v4*)(T3_LDS char*)(lds3 + p.reg) + ((oy * p.pitch + ox) * 8 + lane8);
      int winstep = T3_OY * p.pitch * 8;   // in 16-byte units
      asm volatile("" : "+v"(rowstep), "+v"(winstep));   // VGPRs: an SGPR source operand halves the add's rate
      const int nvx_p = p.pitch / T3_OX, nvx_n = n.pitch / T3_OX;
      if (do_commit) __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the staged window has arrived (explicit, see msda_tiled.hip)
#pragma unroll
      for (int vy = 0; vy < VY; ++vy) {
        if (do_commit && vy * T3_OY < p.wh) {   // uniform.  No per-lane bound: regions hold whole row groups of whole column
#pragma unroll                                  // blocks, pad pixels receive garbage nobody reads
          for (int vx = 0; vx < VX; ++vx)
            if (vx == 0 || vx < nvx_p) win[vx * T3_OX * 8] = wreg[vy][vx];   // the column block is an immediate offset
          win += winstep;
        }
        if (vy * T3_OY < n.wh) {   // uniform
#pragma unroll
          for (int vx = 0; vx < VX; ++vx)
            if (vx == 0 || vx < nvx_n)
              wreg[vy][vx] = __builtin_bit_cast(t3v4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, cb[vx] + rowoff, 0, 0));
          rowoff += rowstep;
        }
      }
    };

    Item cur = make_item(widx);
    T3Entry geo_p = entry(cur, 0, 0);
    commit_and_load(std::false_type{}, geo_p, cur, geo_p);        // step 0 -> registers
    {
      const T3Entry g1 = entry(cur, 0, 1);
      commit_and_load(std::true_type{}, geo_p, cur, g1);          // step 0 -> LDS, step 1 -> registers (the step held staged)
      geo_p = g1;
    }
    __syncthreads();   // step 0 is ready for the gather waves

    int par = 0;
    [[maybe_unused]] int itn = 0;
#pragma unroll 1
    for (unsigned idx = widx;; idx += nw, ++itn) {
      const bool has_next = idx + nw < csize;
      const Item nxt = make_item(has_next ? idx + nw : idx);
#pragma unroll
      for (int kk = 0; kk < L; ++kk) {
        // while the gather waves work on step (cur, kk): commit step + 1 (staged) and load step + 2
        T3STAMP(itn == 1 && wave == 0 && kk < 3, 16 + 4 * kk);
        const T3Entry geo_n = (kk + 2 < L) ? entry(cur, par, (kk + 2) % L) : entry(nxt, par ^ 1, (kk + 2) % L);
        T3STAMP(itn == 1 && wave == 0 && kk < 3, 16 + 4 * kk + 1);
        commit_and_load(std::true_type{}, geo_p, (kk + 2 < L) ? cur : nxt, geo_n);
        T3STAMP(itn == 1 && wave == 0 && kk < 3, 16 + 4 * kk + 2);
        __syncthreads();   // the one barrier of the step
        T3STAMP(itn == 1 && wave == 0 && kk < 3, 16 + 4 * kk + 3);
        geo_p = geo_n;
      }
      if (!has_next) break;
      cur = nxt;
      par ^= 1;
    }
  } else {
    // =========================== gather waves ===========================
    const int gw = wave - NP;
    const int rp = lane >> 5;            // row pair of the wave
    const int side = (lane >> 4) & 1;    // corner column of my DPP row: 0 left, 1 right
    const int k = lane & 15;             // my record: sample k of my row pair = (query k >> 2, point k & 3)
    const int ch = (lane & 15) * 2;      // my two channels as a gathering lane
    const float side_sign = side ? 1.f : -1.f, side_one = side ? 0.f : 1.f;   // column factor = lw * sign + one
    int qslot[NB];                       // my query's index within an item, per batch
#pragma unroll
    for (int b = 0; b < NB; ++b) qslot[b] = ((gw * NB + b) * 2 + rp) * 4 + (k >> 2);

    // my queries of an item: global query index per batch, from the host-built per-tile list (levels in order, raster
    // inside the level's query box) -- one load instead of ~50 half-rate VALU per query to derive it
    struct Mine { int total; int qg[NB]; };
    auto my_queries = [&](const Item& it, int par) __attribute__((always_inline)) {
      Mine me;
      me.total = entry(it, par, 0).total;
      const int* ql = qtab + it.tile * qcap;
#pragma unroll
      for (int b = 0; b < NB; ++b) me.qg[b] = ql[min(qslot[b], me.total - 1)];
      return me;
    };
    // sampling location + attention weight of my (query, point) at every level of an item, in visiting order
    struct Inputs { float x[L][NB], y[L][NB], a[L][NB]; };
    auto load_inputs = [&](const Item& it, int par, const Mine& me, Inputs& iv) __attribute__((always_inline)) {
      if constexpr (!FUSED) {
        const char* lb = reinterpret_cast<const char*>(loc + it.nm * (L * P * 2));
        const char* ab = reinterpret_cast<const char*>(attn + it.nm * (L * P));
#pragma unroll
        for (int kk = 0; kk < L; ++kk) {
          const int l = entry(it, par, kk).l;
#pragma unroll
          for (int b = 0; b < NB; ++b) {
            const unsigned e = (unsigned)(me.qg[b] * (M * L * P) + l * P + (k & 3));
            const float2 xy = *reinterpret_cast<const float2*>(lb + e * 8u);
            iv.x[kk][b] = xy.x; iv.y[kk][b] = xy.y;
            iv.a[kk][b] = *reinterpret_cast<const float*>(ab + e * 4u);
          }
        }
      } else {
        // raw projections -> locations / weights (csrc/msda_prepare.hip's arithmetic; the softmax sums in a different
        // order: per point over the levels, then over the 4 points of the quad)
        const float* row0 = in.proj + (long long)it.n * S * in.row_stride;
        const float* ref0 = in.ref + it.n * in.ref_batch_stride;
        float lg[L][NB];
#pragma unroll
        for (int kk = 0; kk < L; ++kk) {
          const T3Entry e = entry(it, par, kk);
#pragma unroll
          for (int b = 0; b < NB; ++b) {
            const float* row = row0 + (long long)me.qg[b] * in.row_stride;
            const float2 off = *reinterpret_cast<const float2*>(row + it.m * (L * P * 2) + (e.l * P + (k & 3)) * 2);
            const float2 rp2 = *reinterpret_cast<const float2*>(ref0 + ((long long)me.qg[b] * L + e.l) * 2);
            lg[kk][b] = row[in.n_off + it.m * (L * P) + e.l * P + (k & 3)];
            iv.x[kk][b] = rp2.x + off.x / (float)e.W;
            iv.y[kk][b] = rp2.y + off.y / (float)e.H;
          }
        }
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          float mx = lg[0][b];
#pragma unroll
          for (int kk = 1; kk < L; ++kk) mx = fmaxf(mx, lg[kk][b]);
          // the 4 points of a query sit in one quad of lanes: quad_perm [1,0,3,2] and [2,3,0,1]
          mx = fmaxf(mx, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(mx), 0xB1, 0xf, 0xf, false)));
          mx = fmaxf(mx, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(mx), 0x4E, 0xf, 0xf, false)));
          float sum = 0.f;
#pragma unroll
          for (int kk = 0; kk < L; ++kk) {
            iv.a[kk][b] = expf(lg[kk][b] - mx);
            sum += iv.a[kk][b];
          }
          sum += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sum), 0xB1, 0xf, 0xf, false));
          sum += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sum), 0x4E, 0xf, 0xf, false));
#pragma unroll
          for (int kk = 0; kk < L; ++kk) iv.a[kk][b] = iv.a[kk][b] / sum;
        }
      }
    };

    // ---- prologue: inputs of the first item
    Item cur = make_item(widx);
    Mine me_cur = my_queries(cur, 0);
    Inputs in_cur;
    load_inputs(cur, 0, me_cur, in_cur);
    __syncthreads();   // step 0's window is staged

    int par = 0;
    [[maybe_unused]] int itn = 0;
#pragma unroll 1
    for (unsigned idx = widx;; idx += nw, ++itn) {
      const bool has_next = idx + nw < csize;
      const Item nxt = make_item(has_next ? idx + nw : idx);
      // the next item's queries and inputs: issued now, used from the next iteration on (a whole item of cover for the
      // one stream of this operator that is touched exactly once and comes from HBM)
      const Mine me_nxt = my_queries(nxt, par ^ 1);
      Inputs in_nxt;
      load_inputs(nxt, par ^ 1, me_nxt, in_nxt);

      t3v2 acc[NB][4];   // (channel 2k, channel 2k + 1) of my corner column, per batch and query
#pragma unroll
      for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) acc[b][jj] = (t3v2){0.f, 0.f};

#pragma unroll
      for (int kk = 0; kk < L; ++kk) {
        T3STAMP(itn == 1 && gw == 0 && kk < 3, 5 * kk);
        const T3Entry q = entry(cur, par, kk);
        // ---- A. records of this step for my corner column
        int slot[NB];
        float wT[NB], wB[NB];
        bool miss[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          const T3Record r = t3_record(in_cur.x[kk][b], in_cur.y[kk][b], in_cur.a[kk][b], side, side_sign, side_one,
                                       qslot[b] < q.total, q.H, q.W, q.wx0, q.wy0, q.ww, q.wh, q.pitch);
          miss[b] = r.miss;
          slot[b] = r.slot;
          wT[b] = r.wt;
          wB[b] = r.wb;
        }

        T3STAMP(itn == 1 && gw == 0 && kk < 3, 5 * kk + 1);
        // ---- B. gathers: 16 samples per row pair and batch.  Per sample and row: one DPP add (address of the top row),
        // one plain add (bottom row), ONE 64-bit DPP move (both weights), two packed FMAs -- on gfx950 every DPP form issues
        // at half rate, so the broadcasts are not folded into the FMAs (profiles/r02_gfx950_issue_costs.txt)
        const int off_t = (int)lds_base + q.reg + (lane & 15) * 8;
        int pitchv = q.pitch * (D * 4);
        asm volatile("" : "+v"(pitchv));   // a VGPR: an SGPR source operand halves the add's rate
        if (!(ablate & 4)) {
#pragma unroll
          for (int b = 0; b < NB; ++b) {
            if ((gw * NB + b) * 8 < q.total) {   // uniform
              const long long wpair = __builtin_bit_cast(long long, (t3v2){wT[b], wB[b]});   // (w_top, w_bottom): one 64-bit DPP per sample
#define T3_IT(K)                                                                                          \
  {                                                                                                       \
    const int a_t = t3_bcast<K>(slot[b]) + off_t;                                                         \
    const int a_b = a_t + pitchv;                                                                         \
    const t3v2 dt = *(const T3_LDS t3v2*)(unsigned long long)(unsigned)a_t;                               \
    const t3v2 db = *(const T3_LDS t3v2*)(unsigned long long)(unsigned)a_b;                               \
    const t3v2 w = __builtin_bit_cast(t3v2, t3_bcast64<K>(wpair));                                        \
    /* packed math on the pairs exactly as loaded (left to the SLP vectoriser the FMAs of DIFFERENT samples get */ \
    /* paired and every pair costs two v_mov to assemble)                                                       */ \
    acc[b][K >> 2] = __builtin_elementwise_fma((t3v2){w.x, w.x}, dt, acc[b][K >> 2]);                     \
    acc[b][K >> 2] = __builtin_elementwise_fma((t3v2){w.y, w.y}, db, acc[b][K >> 2]);                     \
  }
              T3_IT(0) T3_IT(1) T3_IT(2) T3_IT(3) T3_IT(4) T3_IT(5) T3_IT(6) T3_IT(7)
              T3_IT(8) T3_IT(9) T3_IT(10) T3_IT(11) T3_IT(12) T3_IT(13) T3_IT(14) T3_IT(15)
#undef T3_IT
            }
          }
        }
        T3STAMP(itn == 1 && gw == 0 && kk < 3, 5 * kk + 2);
        // rare: corner columns outside the staged window -> straight from global memory (wave-uniform loop)
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          unsigned long long mm = __ballot(miss[b]);
          if (mm != 0 && !(ablate & 8)) {
            const float* vl = value + (cur.nm + (long long)q.start * M) * D + ch;
#pragma unroll 1
            while (mm) {
              const int bl = __builtin_ctzll(mm);
              mm &= mm - 1;
              const float mx = __shfl(in_cur.x[kk][b], bl, 64), my = __shfl(in_cur.y[kk][b], bl, 64),
                          ma = __shfl(in_cur.a[kk][b], bl, 64);
              const Footprint fp = footprint(q.H, q.W, mx, my, ma);
              const int sd = (bl >> 4) & 1;
              const int wc = sd ? fp.w1 : fp.w0;
              const float w0 = sd ? fp.w01 : fp.w00, w1 = sd ? fp.w11 : fp.w10;
              if ((lane >> 4) == (bl >> 4)) {
                const t3v2 g0 = *reinterpret_cast<const t3v2*>(vl + (long long)(fp.h0 * q.W + wc) * (M * D));
                const t3v2 g1 = *reinterpret_cast<const t3v2*>(vl + (long long)(fp.h1 * q.W + wc) * (M * D));
                const float cx = fmaf(w0, g0.x, w1 * g1.x), cy = fmaf(w0, g0.y, w1 * g1.y);
                const int jb = (bl & 15) >> 2;
#pragma unroll
                for (int jj = 0; jj < 4; ++jj)
                  if (jb == jj) acc[b][jj] += (t3v2){cx, cy};
              }
            }
          }
        }

        // ---- C. last level: add the two corner columns (rows 2i / 2i+1) and store
        if (kk + 1 == L && !(ablate & 16)) {
#pragma unroll
          for (int b = 0; b < NB; ++b) {
            if ((gw * NB + b) * 8 < q.total) {   // uniform
              const int qb = ((gw * NB + b) * 2 + rp) * 4;
#define T3_OUT(JJ)                                                                                                 \
  {                                                                                                                \
    /* rows 2i / 2i+1 hold the two corner columns: swap the odd rows of x with the even rows of y, add -> the even  */ \
    /* rows hold channel 2k, the odd rows channel 2k+1 of the finished output: 32 lanes = 128 contiguous bytes       */ \
    const t3u2 sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc[b][JJ].x), __float_as_uint(acc[b][JJ].y), false, false); \
    const float tot = __uint_as_float(sw.x) + __uint_as_float(sw.y);                                               \
    const int qgj = t3_bcast<4 * JJ>(me_cur.qg[b]);                                                                \
    if (qb + JJ < q.total)                                                                                         \
      *reinterpret_cast<float*>(reinterpret_cast<char*>(out + cur.nm * D) + ((unsigned)(qgj * M * D + ch + side) * 4u)) = tot; \
  }
              T3_OUT(0) T3_OUT(1) T3_OUT(2) T3_OUT(3)
#undef T3_OUT
            }
          }
        }
        T3STAMP(itn == 1 && gw == 0 && kk < 3, 5 * kk + 3);
        __syncthreads();   // the one barrier of the step
        T3STAMP(itn == 1 && gw == 0 && kk < 3, 5 * kk + 4);
      }
      if (!has_next) break;
      cur = nxt;
      me_cur = me_nxt;
      in_cur = in_nxt;
      par ^= 1;
    }
  }
}

// ---- host side: per-geometry step table, built once per (device, level shapes, tile parameters)
struct T3Key {
  int dev, L, TH, TW, R;
  int H[UNIVS_MAX_LEVELS], W[UNIVS_MAX_LEVELS];
  bool operator==(const T3Key& o) const {
    if (dev != o.dev || L != o.L || TH != o.TH || TW != o.TW || R != o.R) return false;
    for (int l = 0; l < L; ++l)
      if (H[l] != o.H[l] || W[l] != o.W[l]) return false;
    return true;
  }
};
struct T3Geo {
  T3Key key;
  T3Entry* table;   // device
  int* qtable;      // device: [ntiles][qcap] global query index of the tile's i-th query
  int qcap;
  int ntiles;
  long long qmax;   // max queries of a tile
  size_t lds;       // bytes of the two window regions
};

static const T3Geo* t3_geometry(const LevelTable& lv, int L, int fine, int TH, int TW, int R) {
  static std::mutex mu;
  static std::vector<T3Geo*> cache;
  T3Key key{};
  if (hipGetDevice(&key.dev) != hipSuccess) return nullptr;
  key.L = L; key.TH = TH; key.TW = TW; key.R = R;
  for (int l = 0; l < L; ++l) { key.H[l] = lv.H[l]; key.W[l] = lv.W[l]; }
  std::lock_guard<std::mutex> lock(mu);
  for (const T3Geo* e : cache)
    if (e->key == key) return e;

  const int tiles_y = (lv.H[fine] + TH - 1) / TH, tiles_x = (lv.W[fine] + TW - 1) / TW;
  std::vector<int4> ax((size_t)L * tiles_x), ay((size_t)L * tiles_y);
  long long lvl_px[UNIVS_MAX_LEVELS] = {0, 0, 0, 0};
  int pitch[UNIVS_MAX_LEVELS] = {0, 0, 0, 0};
  for (int l = 0; l < L; ++l) {
    // windows with the zero ring (ring = 1), capped at the fill grid's extent: samples beyond go through the global
    // fallback, results unchanged
    int mw = 2, mh = 2;
    for (int tx = 0; tx < tiles_x; ++tx) {
      int4& e = ax[(size_t)l * tiles_x + tx];
      axis_entry(tx, tiles_x, TW, lv.W[l], lv.W[fine], R, T3_PITCH_MAX, /*ring=*/1, e);
      mw = std::max(mw, e.w);
    }
    for (int ty = 0; ty < tiles_y; ++ty) {
      int4& e = ay[(size_t)l * tiles_y + ty];
      axis_entry(ty, tiles_y, TH, lv.H[l], lv.H[fine], R, T3_ROWS_MAX, /*ring=*/1, e);
      mh = std::max(mh, e.w);
    }
    pitch[l] = (mw + T3_OX - 1) / T3_OX * T3_OX;
    lvl_px[l] = (long long)pitch[l] * ((mh + T3_OY - 1) / T3_OY * T3_OY);   // whole 4-row groups of whole 8-pixel blocks
  }
  // region plan: step parity picks the region; an odd level count makes odd items start in B, so they visit their
  // two largest windows in swapped order (the accumulators do not care)
  int by_size[UNIVS_MAX_LEVELS], ord[2][UNIVS_MAX_LEVELS], reg[2][UNIVS_MAX_LEVELS];
  for (int l = 0; l < L; ++l) by_size[l] = l;
  std::sort(by_size, by_size + L, [&](int a, int b) { return lvl_px[a] > lvl_px[b]; });
  long long capA = 0, capB = 0;
  for (int par = 0; par < 2; ++par) {
    for (int kk = 0; kk < L; ++kk) ord[par][kk] = by_size[kk];
    const int first_region = (par == 1 && (L & 1)) ? 1 : 0;
    if (first_region == 1) std::swap(ord[par][0], ord[par][1]);
    for (int kk = 0; kk < L; ++kk) {
      const int region = (first_region + kk) & 1;
      long long& cap = region ? capB : capA;
      cap = std::max(cap, lvl_px[ord[par][kk]]);
      reg[par][kk] = region;
    }
  }
  T3Geo* g = new T3Geo();
  g->key = key;
  g->ntiles = tiles_y * tiles_x;
  g->lds = (size_t)(capA + capB) * 128;
  g->qmax = 0;
  std::vector<T3Entry> tab((size_t)g->ntiles * 2 * L);
  g->qcap = 192;
  std::vector<int> qtab((size_t)g->ntiles * g->qcap, 0);
  for (int ty = 0; ty < tiles_y; ++ty)
    for (int tx = 0; tx < tiles_x; ++tx) {
      int pre[UNIVS_MAX_LEVELS + 1] = {0};
      for (int l = 0; l < L; ++l) pre[l + 1] = pre[l] + ax[(size_t)l * tiles_x + tx].y * ay[(size_t)l * tiles_y + ty].y;
      g->qmax = std::max<long long>(g->qmax, pre[L]);
      for (int l = 0; l < L && pre[L] <= g->qcap; ++l) {
        const int4 gx = ax[(size_t)l * tiles_x + tx], gy = ay[(size_t)l * tiles_y + ty];
        for (int i = 0; i < gx.y * gy.y; ++i)
          qtab[(size_t)(ty * tiles_x + tx) * g->qcap + pre[l] + i] = lv.start[l] + (gy.x + i / gx.y) * lv.W[l] + gx.x + i % gx.y;
      }
      for (int par = 0; par < 2; ++par)
        for (int kk = 0; kk < L; ++kk) {
          const int l = ord[par][kk];
          const int4 gx = ax[(size_t)l * tiles_x + tx], gy = ay[(size_t)l * tiles_y + ty];
          T3Entry& e = tab[((size_t)(ty * tiles_x + tx) * 2 + par) * L + kk];
          e.H = lv.H[l]; e.W = lv.W[l]; e.start = lv.start[l]; e.l = l;
          e.wx0 = gx.z; e.wy0 = gy.z; e.ww = gx.w; e.wh = gy.w;
          e.pitch = pitch[l]; e.reg = reg[par][kk] ? (int)(capA * 128) : 0;
          e.qfirst = pre[l]; e.qcount = gx.y * gy.y;
          e.qx0 = gx.x; e.qy0 = gy.x; e.qnx = std::max(gx.y, 1); e.total = pre[L];
        }
    }
  if (hipMalloc(reinterpret_cast<void**>(&g->table), tab.size() * sizeof(T3Entry)) != hipSuccess ||
      hipMemcpy(g->table, tab.data(), tab.size() * sizeof(T3Entry), hipMemcpyHostToDevice) != hipSuccess ||
      hipMalloc(reinterpret_cast<void**>(&g->qtable), qtab.size() * sizeof(int)) != hipSuccess ||
      hipMemcpy(g->qtable, qtab.data(), qtab.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) {
    (void)hipGetLastError();
    delete g;
    return nullptr;
  }
  cache.push_back(g);
  return g;
}

template <int L, int NP, int NG, int NB, bool FUSED>
static void launch_tiled3(unsigned grid, unsigned nitems, hipStream_t st, const float* value, const T3Geo* g, int ablate,
                          const T3Inputs& in, int N, int S, int M, float* out) {
  auto kfn = msda_fwd_tiled3<L, NP, NG, NB, FUSED>;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)g->lds);
  hipLaunchKernelGGL(kfn, dim3(grid), dim3(64 * (NP + NG)), g->lds, st, value, g->table, g->qtable, g->qcap, g->ntiles, ablate,
                     in, N, S, M, out, nitems);
}

// returns 1 if launched, 0 if preconditions do not hold (caller tries the next implementation), <0 on error
static int t3_forward(const float* value, const LevelTable& lv, const T3Inputs& in, bool fused, int N, int S, int M, int D,
                      int L, int Lq, int P, float* out, hipStream_t st) {
  if (D != 32 || P != 4 || L < 2 || L > 4 || Lq != S || M < 1) return 0;
  if ((long long)S * M * D * 4 >= (1LL << 31) || (long long)S * M * L * P * 8 >= (1LL << 31)) return 0;
  if (fused && (long long)S * in.row_stride * 4 >= (1LL << 31)) return 0;
  long long expect = 0;
  int fine = 0;
  for (int l = 0; l < L; ++l) {
    if (lv.start[l] != expect || lv.H[l] < 2 || lv.W[l] < 2) return 0;
    expect += (long long)lv.H[l] * lv.W[l];
    if ((long long)lv.H[l] * lv.W[l] > (long long)lv.H[fine] * lv.W[fine]) fine = l;
  }
  if (expect != S) return 0;

  const int TH = env_int("UNIVS_MSDA_TILE3_H", 8), TW = env_int("UNIVS_MSDA_TILE3_W", 16);
  const int R = env_int("UNIVS_MSDA_HALO", 6);
  const int variant = env_int("UNIVS_MSDA_T3_VARIANT", 1);   // 1: 4 fill + 11 gather waves x 2 batches; 0: 4 + 8 x 3
  const int ablate = env_int("UNIVS_MSDA_ABLATE", 0);
  if (TH < 1 || TW < 1 || R < 0 || R > 64) return 0;
  const T3Geo* g = t3_geometry(lv, L, fine, TH, TW, R);
  const int qcap = variant == 1 ? 11 * 16 : 8 * 24;
  if (!g || g->qmax > qcap || g->qmax < 1 || g->lds > 160 * 1024) return 0;

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
#define T3_LAUNCH2(LL, FF)                                                                              \
  if (variant == 1) launch_tiled3<LL, T3_NP, 11, 2, FF>(grid, (unsigned)nb, st, value, g, ablate, in, N, S, M, out); \
  else launch_tiled3<LL, T3_NP, 8, 3, FF>(grid, (unsigned)nb, st, value, g, ablate, in, N, S, M, out);
#define T3_LAUNCH(LL)              \
  if (fused) { T3_LAUNCH2(LL, true) } \
  else { T3_LAUNCH2(LL, false) }
  switch (L) {
    case 2: T3_LAUNCH(2) break;
    case 3: T3_LAUNCH(3) break;
    default: T3_LAUNCH(4) break;
  }
#undef T3_LAUNCH
#undef T3_LAUNCH2
  int rc = check_launch("msda_fwd_tiled3");
  return rc == UNIVS_OK ? 1 : rc;
}

int msda_forward_tiled3_f32(const float* value, const LevelTable& lv, const float* loc, const float* attn, int N,
                            int S, int M, int D, int L, int Lq, int P, float* out, hipStream_t st) {
  T3Inputs in{};
  in.loc = loc;
  in.attn = attn;
  return t3_forward(value, lv, in, false, N, S, M, D, L, Lq, P, out, st);
}

// MSDeformAttn core fed with the raw projections: msda_prepare (softmax + reference + offset / normaliser) happens
// inside the sampling kernel, the [N, Lq, M, L, P, 2] / [N, Lq, M, L, P] tensors never exist.
int msda_forward_fused_tiled3_f32(const float* value, const LevelTable& lv, const float* proj, int row_stride, int n_off,
                                  const float* ref, long long ref_batch_stride, int N, int S, int M, int D, int L, int Lq,
                                  int P, float* out, hipStream_t st) {
  T3Inputs in{};
  in.proj = proj;
  in.ref = ref;
  in.row_stride = row_stride;
  in.n_off = n_off;
  in.ref_batch_stride = ref_batch_stride;
  return t3_forward(value, lv, in, true, N, S, M, D, L, Lq, P, out, st);
}

}  // namespace univs
