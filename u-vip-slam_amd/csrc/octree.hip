// Kernel wrapper of the quad-tree selection (octree_core.hpp): one workgroup per (frame, level), node state carved out of
// dynamic LDS, candidate state in registers (HBM scratch words for very long lists).  Two instantiations: 256 threads, four
// workgroups per CU (throughput: large batches fill the chip with problems), and 1024 threads, one per CU (latency: a
// small batch leaves CUs idle, so each problem takes four times the lanes and keeps lists of up to 32 K candidates in
// registers).
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include "common.hpp"
#include "gauss_body.hpp"
#include "fast_geom.hpp"
#include "octree_pyramid.hpp"

namespace uvo {

static inline int oct_capacity(int N, int nIni) {
  int m = N > 4 * nIni ? N : 4 * nIni;
  return (m + 8 + 3) & ~3;  // a multiple of 4: every node table of the LDS carve then starts on a 16-byte boundary (rank_below reads keys four at a time)
}
// OctLaunchState::wide_max_problems: at most this many (frame, level) problems run as 1024-thread workgroups (one per CU: beyond
// 256 problems the chip is full either way, and four independent 256-thread problems per CU use it better).  Measured, 640x512 /
// 8 levels, kernel span: 60 vs 81 us at batch 1, 75 vs 125 us at batch 32; at batch 128 wide still has the shorter span (240 vs
// 290 us) but the pipelined throughput of configs[3] drops 5 %, and at batch 256 it is 10 % down.
static inline int pow2_ge(int v) {
  int p = 1;
  while (p < v) p <<= 1;
  return p;
}
// bytes of LDS for node capacity M and a count pyramid of pyr_words (layout must match the carve in k_octree).  The pyramid (+ its 16
// statistics words) shares its bytes with the two box tables: the closed form never touches boxes, and the fall-back starts over.
static inline size_t oct_box_region_bytes(int M, int pyr_words) {
  const size_t boxes = (size_t)8 * M * 2, pyr = (size_t)4 * (pyr_words + 16);
  return (boxes > pyr ? boxes : pyr) + 15 & ~(size_t)15;
}
static inline size_t oct_lds_bytes(int M, int Mp2, int pyr_words) {
  size_t b = 0;
  b += (size_t)16 * M * 2;                  // ccnt[4M], ccnt2[4M] (aliased at the end by best64[2M] / sort64[Mp2])
  b += oct_box_region_bytes(M, pyr_words);  // boxA, boxB | count pyramid
  b += (size_t)4 * M * 7;                   // cntA, cntB, procRank, nodeOfRank, baseOfRank, outKey, outPt
  b += (size_t)4 * Mp2;                     // sortbuf
  b += (size_t)4 * (32 + 16);               // part (one partial per wavefront on the device), sc
  return b;
}

template <int NT>
__device__ __forceinline__ void octree_body(uint8_t* lds, const int level, const int f, const LevelGeom* __restrict__ lv, int nlevels, int Mmax, int Mp2max, int pyr_words,
                                            int box_region, int lds_bytes, const FastLevels& FL, const uint32_t* __restrict__ cand_lo,
                                            int32_t* __restrict__ cursor, int32_t* __restrict__ fcount, int32_t* __restrict__ n_cell_list, uint8_t* cell_hi,
                                            uint32_t* __restrict__ cand_xy, uint32_t* __restrict__ cand_sc, int64_t cand_block, int32_t* __restrict__ cand_count,
                                            uint32_t* __restrict__ pstate, uint32_t* __restrict__ sel_xy, uint32_t* __restrict__ sel_sc, int sel_block,
                                            int32_t* __restrict__ sel_count, const uint16_t* __restrict__ oct_tab) {
#ifdef UVO_OCT_TRACE
  const unsigned long long t_begin = wall_clock64();
  struct Stamp {
    unsigned long long t0;
    int slot;
    __device__ ~Stamp() {
      if (threadIdx.x == 0 && slot < 2048) g_oct_blocks[2 * slot] = t0, g_oct_blocks[2 * slot + 1] = wall_clock64();
    }
  } stamp{t_begin, (int)(blockIdx.x * gridDim.y + blockIdx.y)};
#endif
  OCT_TRACE_MARK()  // kernel start
  const LevelGeom& g = lv[level];
  const int64_t co = f * cand_block + g.cand_off;
  // ---- candidates of this (frame, level).  k_fast_score has already put the NMS survivors that reach fastTh into the candidate array
  // (n_hi of them); the others (7 <= score < fastTh) wait in the level's low list for the per-cell vote, which needs every region of
  // a cell finished: FAST(cell, fastTh); if empty FAST(cell, 7) (src/ORBextractor.cc:792-799) -- a low survivor is a candidate iff its
  // cell holds no survivor >= fastTh.  One contiguous read of the low list, appended behind the first n_hi candidates. ----
  __shared__ int s_pcount, s_zero[NT / 64];  // candidates so far; per wavefront: cell flags found zero
  {
    const FastLevel fg = FL.l[level];
    int32_t* cur = cursor + 2 * ((int64_t)f * nlevels + level);
    const int n_hi = min(cur[0], g.cand_cap), n_lo = min(cur[1], g.cand_cap);
    if (threadIdx.x == 0) s_pcount = n_hi;
    uint8_t* hi = cell_hi + (int64_t)f * FL.flags_per_frame + fg.flag_base;
    const int lane = threadIdx.x & 63;
    // the level's cell flags go to LDS first (the node tables are not live yet: the staging shares their bytes); levels with more
    // cells than fit read them from memory
    uint8_t* s_flag = lds;
    const int n_flags = fg.nRows * fg.nCols;
    const bool flag_lds = n_flags <= lds_bytes;
    // (while they pass by: the cells of this (frame, level) without a survivor >= fastTh are counted -- the batch's fall-back share
    // steers the level's FAST mode, below)
    {
      int z = 0;
      for (int i = threadIdx.x; i < n_flags; i += NT) {
        const uint8_t fl = hi[i];
        if (flag_lds) s_flag[i] = fl;
        z += fl == 0 ? 1 : 0;
      }
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) z += __shfl_xor(z, o, 64);
      if (lane == 0) s_zero[threadIdx.x >> 6] = z;
    }
    __syncthreads();
    constexpr int LU = 8;  // low-list words in flight per thread
    for (int i0 = 0; i0 < n_lo; i0 += LU * NT) {
      uint32_t e[LU];
      bool emit[LU];
#pragma unroll
      for (int u = 0; u < LU; ++u) {
        const int i = i0 + u * NT + (int)threadIdx.x;
        e[u] = i < n_lo ? cand_lo[co + i] : 0u;
      }
#pragma unroll
      for (int u = 0; u < LU; ++u) {
        const int i = i0 + u * NT + (int)threadIdx.x;
        emit[u] = false;
        if (i < n_lo) {
          const int xr = (int)(e[u] & 0xfff), yr = (int)((e[u] >> 12) & 0xfff), sc = (int)(e[u] >> 24);
          int cj = (int)(__umul24((uint32_t)(xr - 3), fg.inv_wcell) >> 24), ci = (int)(__umul24((uint32_t)(yr - 3), fg.inv_hcell) >> 24);
          cj = cj > fg.nCols - 1 ? fg.nCols - 1 : cj;
          ci = ci > fg.nRows - 1 ? fg.nRows - 1 : ci;
          const uint8_t flag = flag_lds ? s_flag[ci * fg.nCols + cj] : hi[ci * fg.nCols + cj];
          emit[u] = !flag && sc >= 7;
        }
      }
#pragma unroll
      for (int u = 0; u < LU; ++u) {
        const uint64_t m = __ballot(emit[u]);
        if (m == 0) continue;
        int slot = 0;
        if (lane == 0) slot = atomicAdd(&s_pcount, (int)__popcll(m));
        slot = __shfl(slot, 0, 64);
        if (emit[u]) {
          const int pos = slot + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
          if (pos < g.cand_cap) {
            cand_xy[co + pos] = (e[u] & 0xfffu) | (((e[u] >> 12) & 0xfffu) << 16);
            cand_sc[co + pos] = e[u] >> 24;
          }
        }
      }
    }
    __syncthreads();
    // the flags and cursors of this (frame, level) have been consumed: leave them zero for the next batch (k_fast_score only ever
    // sets / advances them)
    for (int i = threadIdx.x; i < n_flags; i += NT) hi[i] = 0;
    if (threadIdx.x == 0) cur[0] = 0, cur[1] = 0;
    if (threadIdx.x == 0 && level == 0 && f == 0) *n_cell_list = 0;  // the consumed list of fall-back cells (k_fast_cells_list)
  }
  // The lane's adaptive FAST mode: a level streams either at fastTh, with the sparse literal-7 pass over the cells left empty
  // (k_fast_cells), or once at 7 with the vote above; both give the same candidates, the cheaper one depends on how many cells fall
  // back.  This problem's count of fall-back cells (its zero flags minus the grid positions that are no cell at all) goes to
  // fcount[frame][level]; k_assemble, next on the stream, sums the batch and re-decides every level (describe.hip: adapt_fast_mode).
  // (Summing here -- an atomic per workgroup plus a last-one-out test -- cost the kernel 14 %; an agent-scope fence, which writes
  // back and invalidates an XCD's L2 on gfx950, 60 %.)
  if (threadIdx.x == 0) {
    int z = -(FL.l[level].nRows * FL.l[level].nCols - g.n_cells);
#pragma unroll
    for (int w = 0; w < NT / 64; ++w) z += s_zero[w];
    fcount[f * nlevels + level] = z;
  }
  OCT_TRACE_MARK()  // end of the candidate gather
  int P = s_pcount;
  if (threadIdx.x == 0) cand_count[f * nlevels + level] = P;
  P = P > g.cand_cap ? g.cand_cap : P;
  int32_t* out_n = sel_count + f * nlevels + level;
  if (P == 0) {
    if (threadIdx.x == 0) *out_n = 0;
    return;
  }
  oct::Params pr;
  pr.P = P;
  pr.N = g.quota;
  pr.W = g.bw;
  pr.H = g.bh;
  pr.nIni = g.nIni;
  pr.hX = g.hX;
  pr.nCols = g.nCols, pr.nRows = g.nRows, pr.wCell = g.wCell, pr.hCell = g.hCell;
  pr.M = Mmax;
  pr.Mp2 = Mp2max;
  oct::Work w;
  uint8_t* p = lds;
  w.ccnt = reinterpret_cast<uint32_t*>(p), p += (size_t)16 * Mmax;
  w.ccnt2 = reinterpret_cast<uint32_t*>(p), p += (size_t)16 * Mmax;
  w.boxA = reinterpret_cast<oct::Box*>(p);
  w.boxB = w.boxA + Mmax;
  w.pyr = reinterpret_cast<uint32_t*>(p);  // same bytes as the boxes
  w.stat = reinterpret_cast<int*>(p) + pyr_words;
  p += box_region;
  w.cntA = reinterpret_cast<uint32_t*>(p), p += (size_t)4 * Mmax;
  w.cntB = reinterpret_cast<uint32_t*>(p), p += (size_t)4 * Mmax;
  w.procRank = reinterpret_cast<int32_t*>(p), p += (size_t)4 * Mmax;
  w.nodeOfRank = reinterpret_cast<uint32_t*>(p), p += (size_t)4 * Mmax;
  w.baseOfRank = reinterpret_cast<uint32_t*>(p), p += (size_t)4 * Mmax;
  w.outKey = reinterpret_cast<uint32_t*>(p), p += (size_t)4 * Mmax;
  w.outPt = reinterpret_cast<uint32_t*>(p), p += (size_t)4 * Mmax;
  w.sortbuf = reinterpret_cast<uint32_t*>(p), p += (size_t)4 * Mp2max;
  w.part = reinterpret_cast<uint32_t*>(p), p += (size_t)4 * 32;
  w.sc = reinterpret_cast<int*>(p), p += (size_t)4 * 16;
  // path tables of the closed form: the bytes of ccnt / ccnt2 behind the first 2 * Mmax words (`best` lives in front of them)
  w.tab = reinterpret_cast<uint16_t*>(w.ccnt + 2 * Mmax);
  w.tab_cap = 12 * Mmax;  // 24 * Mmax bytes of 2-byte entries
  w.tab_src = oct_tab != nullptr && g.oct_tab_off >= 0 ? oct_tab + g.oct_tab_off : nullptr;

  const int64_t so = (int64_t)f * sel_block + g.sel_off;
  // First the closed form over the count pyramid (no pass over the candidates per generation, no per-candidate state).  Trees deeper
  // than the pyramid (sparse or clustered candidate sets: short lists in practice) take the pass-per-generation form, candidate state
  // in 8 registers per thread or, for long lists, in the HBM scratch words.
  int n = oct::run_pyramid(pr, w, cand_xy + co, cand_sc + co, sel_xy + so, sel_sc + so, g.sel_cap);
  if (n < 0) {
    if (P <= 8 * NT)
      n = oct::run<8>(pr, w, cand_xy + co, cand_sc + co, pstate + co, sel_xy + so, sel_sc + so, g.sel_cap);
    else
      n = oct::run<0>(pr, w, cand_xy + co, cand_sc + co, pstate + co, sel_xy + so, sel_sc + so, g.sel_cap);
  }
  if (threadIdx.x == 0) *out_n = n;
}

template <int NT>
__global__ __launch_bounds__(NT, NT == 256 ? 4 : 1) void k_octree(const LevelGeom* __restrict__ lv, int nlevels, int Mmax, int Mp2max, int pyr_words, int box_region, int lds_bytes,
                                                        FastLevels FL, const uint32_t* __restrict__ cand_lo,
                                                        int32_t* __restrict__ cursor, int32_t* __restrict__ fcount, int32_t* __restrict__ n_cell_list, uint8_t* cell_hi,
                                                        uint32_t* __restrict__ cand_xy, uint32_t* __restrict__ cand_sc,
                                                        int64_t cand_block, int32_t* __restrict__ cand_count,
                                                        uint32_t* __restrict__ pstate, uint32_t* __restrict__ sel_xy,
                                                        uint32_t* __restrict__ sel_sc, int sel_block, int32_t* __restrict__ sel_count, const uint16_t* __restrict__ oct_tab) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  // level-major dispatch: the long level-0 problems start first
  octree_body<NT>(lds, (int)blockIdx.y, (int)blockIdx.x, lv, nlevels, Mmax, Mp2max, pyr_words, box_region, lds_bytes, FL, cand_lo, cursor, fcount, n_cell_list, cell_hi, cand_xy,
                  cand_sc, cand_block, cand_count, pstate, sel_xy, sel_sc, sel_block, sel_count, oct_tab);
}

// DistributeOctTree and GaussianBlur in ONE launch (src/ORBextractor.cc:1006-1287 and :942: neither reads what the other writes).  The
// quad-tree is a chain of dependent phases per (frame, level) -- two thousand workgroups that mostly wait -- the blur a streaming kernel
// that fills the chip: as two launches of one stream they run one after the other (and another lane's kernels hide little of it: every
// heavy kernel fills the chip by itself); as one grid -- the quad-tree problems first, level-major, the blur's
// workgroups behind them -- the blur streams through the issue slots the quad-tree leaves idle.
struct GaussArgs {
  const uint8_t* pyr;
  uint8_t* blur;
  int64_t pyr_block;
  int4 taps;
  int rows_per_seg, blocks_x, batch;
  Level0View l0;
  GaussPlans plans;
};
#ifndef UVO_OCT_EVERY
#define UVO_OCT_EVERY 3  // every third workgroup of the front of the grid is a quad-tree problem (measured against every second: the same)
#endif
static_assert(UVO_OCT_EVERY >= 2, "the interleave needs at least one blur workgroup between two quad-tree problems");
#ifndef UVO_OCT_GAUSS_OCC
#define UVO_OCT_GAUSS_OCC 5  // workgroups per CU the register allocation is held to: 96 VGPRs, five wavefronts per SIMD.  At 4 the SSE2 instantiation takes
                             // 100 registers and falls to four wavefronts per SIMD: 0.251 against 0.229 ms per 257-frame launch (profiles/r06_blur_occupancy_ab.txt)
#endif
template <bool SSE2>
__global__ __launch_bounds__(256, UVO_OCT_GAUSS_OCC) void k_octree_gauss(int n_oct, GaussArgs G, const LevelGeom* __restrict__ lv, int nlevels, int Mmax, int Mp2max, int pyr_words, int box_region,
                                                        int lds_bytes, FastLevels FL, const uint32_t* __restrict__ cand_lo, int32_t* __restrict__ cursor,
                                                        int32_t* __restrict__ fcount, int32_t* __restrict__ n_cell_list, uint8_t* cell_hi,
                                                        uint32_t* __restrict__ cand_xy, uint32_t* __restrict__ cand_sc, int64_t cand_block,
                                                        int32_t* __restrict__ cand_count, uint32_t* __restrict__ pstate, uint32_t* __restrict__ sel_xy,
                                                        uint32_t* __restrict__ sel_sc, int sel_block, int32_t* __restrict__ sel_count, const uint16_t* __restrict__ oct_tab) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  // Workgroups are dispatched in index order, and all of them reserve the same LDS: with the quad-tree problems in front they would take
  // every slot of every CU and the blur would start when they are done.  Interleaved -- every third workgroup a quad-tree problem (level-
  // major: the long level-0 problems first), the others the blur -- a CU holds both kinds from the start.
  int b = (int)blockIdx.x, o = -1;
  if (b < UVO_OCT_EVERY * n_oct) {
    if (b % UVO_OCT_EVERY == 0) o = b / UVO_OCT_EVERY;
    else b -= b / UVO_OCT_EVERY + 1;
  } else {
    b -= n_oct;
  }
  if (o >= 0) {
    octree_body<256>(lds, o / G.batch, o % G.batch, lv, nlevels, Mmax, Mp2max, pyr_words, box_region, lds_bytes, FL, cand_lo, cursor, fcount, n_cell_list, cell_hi, cand_xy, cand_sc,
                     cand_block, cand_count, pstate, sel_xy, sel_sc, sel_block, sel_count, oct_tab);
  } else {
    gauss7_body<SSE2>(b, G.blocks_x, G.batch, reinterpret_cast<uint32_t(*)[GS_TILE_DW]>(lds), G.pyr, G.blur, G.pyr_block, lv, nlevels, G.taps, G.rows_per_seg, G.l0, G.plans);
  }
}

// LDS the quad-tree kernel needs for this geometry (M = the largest node table, its power of two, the count pyramid)
static size_t octree_lds(const Geom& g, int& M, int& Mp2, int& pyr_words) {
  M = 0;
  for (int l = 0; l < g.nlevels; ++l) {
    const int m = oct_capacity(g.lv[l].quota, g.lv[l].nIni);
    M = m > M ? m : M;
  }
  Mp2 = pow2_ge(M);
  pyr_words = 0;
  for (int l = 0; l < g.nlevels; ++l) pyr_words = std::max(pyr_words, oct::pyramid_words(g.lv[l].nIni));
  return oct_lds_bytes(M, Mp2, pyr_words);
}

// Everything about the quad-tree launch that can fail, done BEFORE the batch's first kernel is enqueued: k_octree is also what zeroes
// the fill cursors, cell flags and the fall-back cell list for the lane's next batch, so the launch itself must not be skipped once the
// FAST kernels of the batch are in the stream.
int prepare_octree(const Geom& g) {
  int M, Mp2, pyr_words;
  const size_t lds = octree_lds(g, M, Mp2, pyr_words);
  if (lds > 64 * 1024) {
    // hipFuncSetAttribute SETS the function's limit on the current device -- it does not raise it -- so several handles on one device
    // (different nfeatures -> different sizes) must agree on the largest request: a per-device maximum, raised under a lock
    static std::mutex mu;
    static size_t dev_max[64] = {0};
    int dev = 0;
    UVO_HIP_CHECK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(mu);
    if (dev < 0 || dev >= 64 || lds > dev_max[dev]) {
      UVO_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_octree<OCT_THREADS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      UVO_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_octree<1024>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      if (dev >= 0 && dev < 64) dev_max[dev] = lds;
    }
  }
  return UVO_OK;
}

int launch_octree(hipStream_t s, OctLaunchState& st, const LevelGeom* d_lv, const Geom& g, const uint32_t* d_cand_lo, int32_t* d_cursor,
                   int32_t* d_fcount, int32_t* d_n_cell_list, uint8_t* d_cell_hi, uint32_t* d_cand_xy, uint32_t* d_cand_sc, int64_t cand_block, int32_t* d_cand_count, uint32_t* d_pstate, uint32_t* d_sel_xy, uint32_t* d_sel_sc,
                   int32_t* d_sel_count, int batch, const uint16_t* d_oct_tab) {
  int M, Mp2, pyr_words;
  const size_t lds = octree_lds(g, M, Mp2, pyr_words);  // (prepare_octree has raised the kernel's limit for it)
  // The kernel is latency bound (a few dozen dependent phases per problem): the per-candidate state lives in registers,
  // the node tables in LDS, and the grid is level-major so that the long level-0 problems are dispatched first.
  const bool wide = batch * g.nlevels <= st.wide_max_problems;
  const int threads = wide ? 1024 : OCT_THREADS;
#ifdef UVO_OCT_TRACE
  {
    static bool once = false;
    if (!once) {
      once = true;
      int nb = -1;
      hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(k_octree<OCT_THREADS>), OCT_THREADS, lds);
      hipFuncAttributes fa;
      hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(k_octree<OCT_THREADS>));
      fprintf(stderr, "[oct] M=%d Mp2=%d lds=%zu occupancy blocks/CU=%d regs=%d static_lds=%zu\n", M, Mp2, lds, nb, fa.numRegs, fa.sharedSizeBytes);
    }
  }
#endif
  if (wide)
    hipLaunchKernelGGL(k_octree<1024>, dim3(batch, g.nlevels), dim3(threads), lds, s, d_lv, g.nlevels, M, Mp2, pyr_words, (int)oct_box_region_bytes(M, pyr_words), (int)lds, fast_levels(g, batch), d_cand_lo,
                       d_cursor, d_fcount, d_n_cell_list, d_cell_hi, d_cand_xy, d_cand_sc, cand_block, d_cand_count, d_pstate, d_sel_xy, d_sel_sc, g.sel_block, d_sel_count, d_oct_tab);
  else
    hipLaunchKernelGGL(k_octree<OCT_THREADS>, dim3(batch, g.nlevels), dim3(threads), lds, s, d_lv, g.nlevels, M, Mp2, pyr_words, (int)oct_box_region_bytes(M, pyr_words), (int)lds, fast_levels(g, batch),
                       d_cand_lo, d_cursor, d_fcount, d_n_cell_list, d_cell_hi, d_cand_xy, d_cand_sc, cand_block, d_cand_count, d_pstate, d_sel_xy, d_sel_sc, g.sel_block,
                       d_sel_count, d_oct_tab);
  return UVO_OK;
}


int gauss7_rows_per_seg(int batch);
int gauss7_blocks_per_frame(const Geom& g, int rows_per_seg);

// true when the batch's quad-tree problems take the 256-thread form, i.e. when launch_octree_gauss applies (small batches run the
// quad-tree as 1024-thread workgroups, one per CU: the two kernels stay apart there)
bool octree_gauss_applies(const OctLaunchState& st, const Geom& g, int batch) {
  int M, Mp2, pyr_words;
  return batch * g.nlevels > st.wide_max_problems && octree_lds(g, M, Mp2, pyr_words) <= 64 * 1024 &&
         gauss7_blocks_per_frame(g, gauss7_rows_per_seg(batch)) >= (UVO_OCT_EVERY - 1) * g.nlevels;  // (the interleave needs UVO_OCT_EVERY - 1 blur workgroups per quad-tree problem)
}

void launch_octree_gauss(hipStream_t s, const LevelGeom* d_lv, const Geom& g, const uint8_t* d_pyr, uint8_t* d_blur, int64_t pyr_block, int4 taps, int sse2_rounding,
                         const uint32_t* d_cand_lo, int32_t* d_cursor, int32_t* d_fcount, int32_t* d_n_cell_list, uint8_t* d_cell_hi, uint32_t* d_cand_xy,
                         uint32_t* d_cand_sc, int64_t cand_block, int32_t* d_cand_count, uint32_t* d_pstate, uint32_t* d_sel_xy, uint32_t* d_sel_sc,
                         int32_t* d_sel_count, int batch, Level0View l0, const uint16_t* d_oct_tab) {
  int M, Mp2, pyr_words;
  const size_t lds = std::max(octree_lds(g, M, Mp2, pyr_words), (size_t)GS_LDS_BYTES);
  GaussArgs G;
  G.pyr = d_pyr, G.blur = d_blur, G.pyr_block = pyr_block, G.taps = taps, G.batch = batch, G.l0 = l0;
  G.rows_per_seg = gauss7_rows_per_seg(batch), G.blocks_x = gauss7_blocks_per_frame(g, G.rows_per_seg);
  G.plans = gauss7_plans(g, G.rows_per_seg);
  const int n_oct = batch * g.nlevels;
  const dim3 grid(n_oct + G.blocks_x * batch);
  if (sse2_rounding)
    hipLaunchKernelGGL(k_octree_gauss<true>, grid, dim3(256), lds, s, n_oct, G, d_lv, g.nlevels, M, Mp2, pyr_words, (int)oct_box_region_bytes(M, pyr_words), (int)lds,
                       fast_levels(g, batch), d_cand_lo, d_cursor, d_fcount, d_n_cell_list, d_cell_hi, d_cand_xy, d_cand_sc, cand_block, d_cand_count, d_pstate, d_sel_xy, d_sel_sc,
                       g.sel_block, d_sel_count, d_oct_tab);
  else
    hipLaunchKernelGGL(k_octree_gauss<false>, grid, dim3(256), lds, s, n_oct, G, d_lv, g.nlevels, M, Mp2, pyr_words, (int)oct_box_region_bytes(M, pyr_words), (int)lds,
                       fast_levels(g, batch), d_cand_lo, d_cursor, d_fcount, d_n_cell_list, d_cell_hi, d_cand_xy, d_cand_sc, cand_block, d_cand_count, d_pstate, d_sel_xy, d_sel_sc,
                       g.sel_block, d_sel_count, d_oct_tab);
}

// the levels' path tables as the closed form builds them (octree_pyramid.hpp), for the kernels to copy
void octree_fill_path_tables(const Geom& g, uint16_t* dst) {
  for (int l = 0; l < g.nlevels; ++l) {
    const LevelGeom& L = g.lv[l];
    const int G = oct::pyramid_depth(L.nIni);
    if (L.oct_tab_off < 0 || G == 0) continue;
    oct::Params pr;
    memset(&pr, 0, sizeof(pr));
    pr.W = L.bw, pr.H = L.bh, pr.nIni = L.nIni, pr.hX = L.hX;
    uint16_t* xs = dst + L.oct_tab_off;
    uint16_t* ys = xs + L.bw;
    for (int x = 0; x < L.bw; ++x) xs[x] = (uint16_t)oct::path_xbits(pr, G, x);
    for (int y = 0; y < L.bh; ++y) ys[y] = (uint16_t)oct::path_ybits(pr, G, y);
  }
}

}  // namespace uvo

#ifdef UVO_OCT_TRACE
extern "C" int uvo_debug_oct_blocks(unsigned long long* out) {
  hipDeviceSynchronize();
  hipMemcpyFromSymbol(out, HIP_SYMBOL(g_oct_blocks), sizeof(unsigned long long) * 4096);
  return 2048;
}
extern "C" int uvo_debug_oct_trace(unsigned long long* out, int cap) {
  int n = 0;
  hipDeviceSynchronize();
  hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_oct_trace_n), sizeof(int));
  n = n < cap ? n : cap;
  hipMemcpyFromSymbol(out, HIP_SYMBOL(g_oct_trace), sizeof(unsigned long long) * 2 * n);
  int zero = 0;
  hipMemcpyToSymbol(HIP_SYMBOL(g_oct_trace_n), &zero, sizeof(int));
  return n;
}
#endif
