// Quad-tree keypoint selection in closed form over a count pyramid -- the fast path of k_octree.
// Replaces ORBextractor::DistributeOctTree + ExtractorNode::DivideNode (src/ORBextractor.cc:1006-1287); same results as the
// pass-per-generation formulation in octree_core.hpp, which stays as the fall-back for trees deeper than the pyramid.
//
// DivideNode halves a box at ceil(size / 2), so a point's whole root-to-leaf path is a pure function of its coordinates and the
// root box, and the node it sits in at generation g is the length-g prefix of its path.  One pass over the points histograms the
// depth-G prefixes, sums of four give the counts of every shallower node, and then everything the full passes (:1061-1132) decide
// is arithmetic on those counts: a node of depth g exists iff it has points and its parent has more than one; the list size after
// pass g is (#expandable nodes of depth g) + (#single-point nodes of depth <= g); the careful phase starts at the first g with
// size + 3 * expandable > N (:1140).  No pass over the points and no workgroup barrier per generation.  The careful rounds
// (:1143-1204) work on the < N expandable nodes of one generation: sort by (size desc, newest first), cut at the first split that
// reaches N, children by creation rank -- node-level work only, child counts read from the pyramid.  A second pass over the points
// finds each point's final node by walking its prefixes and takes the per-node maximum response (:1208-1226).
//
// List order (tools/octree_pyramid_proto.py is the executable derivation, checked against the std::list restatement): children are
// pushed to the FRONT in the order n1..n4 while parents are visited front to back, so a generation made by full passes reads: last
// digit descending, the digits above alternating, the root like digit 1.  With the even digits complemented ("T bin") generation
// g is ascending in T for even g and descending for odd g: the position of a node in its generation is its bin (or the mirrored
// bin), and the output order (generation desc, position asc) needs one sort of <= N + 3 keys at the end.
//
// Written against the OCT_* phase macros of octree_core.hpp so that tests/emu/ runs the same body on the host.
#pragma once
#include "octree_core.hpp"

namespace uvo {
namespace oct {

constexpr uint32_t PYR_FINAL = 0x80000000u;  // pyramid word of a final node: flag | output slot
constexpr int PYR_MAX_DEPTH = 5;
constexpr int PYR_MAX_BINS = 4096;           // deepest level: nIni * 4^G bins; positions inside a generation fit 12 bits
constexpr int PYR_STAT_E = 0, PYR_STAT_F = 8;  // stat[]: expandable / single-point nodes per depth

OCT_FN int pyramid_depth(int nIni) { return nIni <= 4 ? 5 : (nIni <= 16 ? 4 : (nIni <= 64 ? 3 : 0)); }
OCT_FN int pyramid_words(int nIni) {
  const int G = pyramid_depth(nIni);
  int o = 0, n = nIni;
  for (int g = 0; g <= G; ++g) o += n, n *= 4;
  return o;
}

// T bin of the depth-G prefix of a point's path: root, then per DivideNode level the child digit (n1..n4 = 0..3: right +1, bottom +2),
// even levels complemented.  The box arithmetic is DivideNode's (:1233-1258): halfX = ceil((UR.x - UL.x) / 2), left iff x < UL.x + halfX.
OCT_FN uint32_t path_tbin(const Params& pr, int G, uint32_t xy) {
  const int x = (int)(xy & 0xffff), y = (int)(xy >> 16);
  const int r = (int)((float)x / pr.hX);
  int ulx = (int)(pr.hX * (float)r), urx = (int)(pr.hX * (float)(r + 1)), uly = 0, bry = pr.H;
  uint32_t b = (uint32_t)r;
#pragma unroll
  for (int k = 1; k <= PYR_MAX_DEPTH; ++k) {
    if (k > G) break;
    const int mx = ulx + half_ceil(urx - ulx), my = uly + half_ceil(bry - uly);
    const int dx = x >= mx ? 1 : 0, dy = y >= my ? 1 : 0;
    ulx = dx ? mx : ulx, urx = dx ? urx : mx;
    uly = dy ? my : uly, bry = dy ? bry : my;
    uint32_t d = (uint32_t)(dx + 2 * dy);
    if ((k & 1) == 0) d = 3u - d;
    b = b * 4u + d;
  }
  return b;
}

// The path is separable: the x splits of DivideNode depend on x (and the root, itself a function of x) only, the y splits on y only
// (every root spans the full height).  xs[x] = root and the x bits of the T bin, ys[y] = its y bits: tbin = xs[x] | ys[y].  Building
// the two tables costs the path arithmetic once per column and row (W + H of them) instead of once per candidate and pass.
OCT_FN uint32_t path_xbits(const Params& pr, int G, int x) {
  const int r = (int)((float)x / pr.hX);
  int ulx = (int)(pr.hX * (float)r), urx = (int)(pr.hX * (float)(r + 1));
  uint32_t b = (uint32_t)r;
#pragma unroll
  for (int k = 1; k <= PYR_MAX_DEPTH; ++k) {
    if (k > G) break;
    const int mx = ulx + half_ceil(urx - ulx);
    const int dx = x >= mx ? 1 : 0;
    ulx = dx ? mx : ulx, urx = dx ? urx : mx;
    b = b * 4u + (uint32_t)((k & 1) == 0 ? 1 - dx : dx);
  }
  return b;
}
OCT_FN uint32_t path_ybits(const Params& pr, int G, int y) {
  int uly = 0, bry = pr.H;
  uint32_t b = 0;
#pragma unroll
  for (int k = 1; k <= PYR_MAX_DEPTH; ++k) {
    if (k > G) break;
    const int my = uly + half_ceil(bry - uly);
    const int dy = y >= my ? 1 : 0;
    uly = dy ? my : uly, bry = dy ? bry : my;
    b = b * 4u + 2u * (uint32_t)((k & 1) == 0 ? 1 - dy : dy);
  }
  return b;
}

// adds a wave's worth of predicate counts to an LDS counter with one atomic per wavefront (host emulation: a plain add)
OCT_FN void stat_add(int* dst, bool pred) {
#if OCT_DEVICE
  const unsigned long long m = __ballot(pred);
  if (m != 0 && (threadIdx.x & 63) == (unsigned)(__ffsll((long long)m) - 1)) atomicAdd(dst, (int)__popcll(m));
#else
  if (pred) *dst += 1;
#endif
}

// adds every thread's pair of small counts (e | f << 16) to two LDS counters: one wavefront reduction, one atomic per counter and wavefront
// (every thread of a wavefront must call it; host emulation: plain adds)
OCT_FN void stat_add_pair(int* dst_e, int* dst_f, uint32_t ef) {
#if OCT_DEVICE
  // wavefront sum on the DPP path (no LDS round trips): pairs, quads, half rows, rows, then row 0 -> 1 and 2 -> 3, rows 0 - 1 -> 2 - 3;
  // lane 63 holds the total
  int v = (int)ef;
  v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, false);   // quad_perm:[1,0,3,2]
  v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, false);   // quad_perm:[2,3,0,1]
  v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, false);  // row_half_mirror
  v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, false);  // row_mirror
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);  // row_bcast:15 into rows 1 and 3
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);  // row_bcast:31 into rows 2 and 3
  ef = (uint32_t)v;
  if ((threadIdx.x & 63) == 63) {
    if (ef & 0xffffu) atomicAdd(dst_e, (int)(ef & 0xffffu));
    if (ef >> 16) atomicAdd(dst_f, (int)(ef >> 16));
  }
#else
  *dst_e += (int)(ef & 0xffffu);
  *dst_f += (int)(ef >> 16);
#endif
}
OCT_FN uint32_t stat_ef(uint32_t c) { return c > 1 ? 1u : (c == 1 ? 0x10000u : 0u); }  // a node with c points: expandable / single-point

// returns the number of selected points, or -1 when the tree is deeper than the pyramid (caller falls back to oct::run).
// No per-candidate state: both passes over the candidates recompute the path from the coordinates (a few dozen integer
// operations), so the kernel's register budget is set by the node-level phases alone.
OCT_FN int run_pyramid(const Params& pr, const Work& w, const uint32_t* cand_xy, const uint32_t* cand_score, uint32_t* sel_xy,
                       uint32_t* sel_score, int sel_cap) {
  const int P = pr.P, N = pr.N;
  const int G = pyramid_depth(pr.nIni);
  if (G == 0) return -1;
  int base[PYR_MAX_DEPTH + 2], nb[PYR_MAX_DEPTH + 2];
  {
    int o = 0, n = pr.nIni;
    for (int g = 0; g <= PYR_MAX_DEPTH; ++g) {
      base[g] = o, nb[g] = n;
      if (g <= G) o += n, n *= 4;
    }
    base[PYR_MAX_DEPTH + 1] = o, nb[PYR_MAX_DEPTH + 1] = 0;
  }
  const int total = base[G] + nb[G];
  int* sc = w.sc;
  uint32_t* pyr = w.pyr;
  int* stat = w.stat;
  uint32_t* keys_in = reinterpret_cast<uint32_t*>(w.procRank);  // expandable nodes of the next careful round: (0xFFFFF - count) << 12 | position
  uint32_t* Acur = w.cntA;   // careful-born generation: bin of the node at a list position
  uint32_t* Anext = w.cntB;
  // tables pay when there are clearly more candidates than columns + rows, and need room
  const bool use_tab = w.tab != nullptr && pr.W + pr.H <= w.tab_cap && P > 2 * (pr.W + pr.H);
  uint16_t* xs = w.tab;
  uint16_t* ys = w.tab + pr.W;
  auto tbin_of = [&](uint32_t xy) -> uint32_t {
    return use_tab ? ((uint32_t)xs[xy & 0xffff] | (uint32_t)ys[xy >> 16]) : path_tbin(pr, G, xy);
  };

  // ---- histogram of the depth-G prefixes ----
  OCT_PHASE_BEGIN
  for (int i = tid; i < total; i += OCT_NT) pyr[i] = 0;
  if (tid < 16) stat[tid] = 0;
  if (tid == 0) sc[SC_NOUT] = 0, sc[SC_NA] = 0;
  if (use_tab && w.tab_src != nullptr) {  // ready-made (octree_fill_path_tables): two entries per load, both ends 4-byte aligned
    const uint32_t* src = reinterpret_cast<const uint32_t*>(w.tab_src);
    uint32_t* dst = reinterpret_cast<uint32_t*>(w.tab);
    for (int i = tid; i < (pr.W + pr.H + 1) / 2; i += OCT_NT) dst[i] = src[i];
  } else if (use_tab) {
    for (int x = tid; x < pr.W; x += OCT_NT) xs[x] = (uint16_t)path_xbits(pr, G, x);
    for (int y = tid; y < pr.H; y += OCT_NT) ys[y] = (uint16_t)path_ybits(pr, G, y);
  }
  OCT_PHASE_END
  OCT_PHASE_BEGIN
  // (the candidates are read eight at a time, the next eight while these are counted: the pass is bound by the latency of these loads,
  // not by arithmetic; indices past the list are clamped into it so that every load is unconditional)
  {
    uint32_t cur[8], nxt[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) cur[u] = cand_xy[tid + u * OCT_NT < P ? tid + u * OCT_NT : P - 1];
    for (int p0 = tid; p0 < P; p0 += 8 * OCT_NT) {
      const int pn = p0 + 8 * OCT_NT;
#pragma unroll
      for (int u = 0; u < 8; ++u) nxt[u] = cand_xy[pn + u * OCT_NT < P ? pn + u * OCT_NT : P - 1];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (p0 + u * OCT_NT < P) OCT_ATOMIC_ADD(&pyr[base[G] + (int)tbin_of(cur[u])], 1u);
#pragma unroll
      for (int u = 0; u < 8; ++u) cur[u] = nxt[u];
    }
  }
  OCT_PHASE_END
  // ---- counts of the shallower nodes: sums of four, two levels per phase -- and, while the counts pass by, per depth how many nodes
  //      EXIST (they hold points and their parent holds more than one, i.e. was split) with more than one point / with exactly one:
  //      a phase sees every node it sums together with its parent's count ----
  for (int g = G; g > 0;) {
    const bool two = g >= 2;
    OCT_PHASE_BEGIN
    uint32_t ef_g = 0, ef_g1 = 0, ef_top = 0;  // (e | f << 16) of depth g, of depth g - 1, and of depth 0 when this phase reaches it
    if (two) {
      for (int b = tid; b < nb[g - 2]; b += OCT_NT) {
        uint32_t s = 0, s1v[4];
        for (int c = 0; c < 4; ++c) {
          const uint32_t* q = &pyr[base[g] + 16 * b + 4 * c];
          const uint32_t s1 = q[0] + q[1] + q[2] + q[3];
          pyr[base[g - 1] + 4 * b + c] = s1;
          ef_g += (s1 > 1 ? ~0u : 0u) & (stat_ef(q[0]) + stat_ef(q[1]) + stat_ef(q[2]) + stat_ef(q[3]));  // (branch free: the sixteen reads stay in flight together)
          s1v[c] = s1;
          s += s1;
        }
        pyr[base[g - 2] + b] = s;
        ef_g1 += (s > 1 ? ~0u : 0u) & (stat_ef(s1v[0]) + stat_ef(s1v[1]) + stat_ef(s1v[2]) + stat_ef(s1v[3]));
        ef_top += g == 2 ? stat_ef(s) : 0u;
      }
    } else {
      for (int b = tid; b < nb[g - 1]; b += OCT_NT) {
        const uint32_t* q = &pyr[base[g] + 4 * b];
        const uint32_t p = q[0] + q[1] + q[2] + q[3];
        pyr[base[g - 1] + b] = p;
        ef_g += (p > 1 ? ~0u : 0u) & (stat_ef(q[0]) + stat_ef(q[1]) + stat_ef(q[2]) + stat_ef(q[3]));
        ef_top += g == 1 ? stat_ef(p) : 0u;
      }
    }
    stat_add_pair(&stat[PYR_STAT_E + g], &stat[PYR_STAT_F + g], ef_g);
    if (two) stat_add_pair(&stat[PYR_STAT_E + g - 1], &stat[PYR_STAT_F + g - 1], ef_g1);
    if ((two && g == 2) || (!two && g == 1)) stat_add_pair(&stat[PYR_STAT_E], &stat[PYR_STAT_F], ef_top);
    OCT_PHASE_END
    g -= two ? 2 : 1;
  }
  // ---- the full passes, on counts alone (every thread walks the same few numbers) ----
  int g = 0;
  int size = stat[PYR_STAT_E] + stat[PYR_STAT_F];
  bool careful = false;
  for (;;) {
    const int E = stat[PYR_STAT_E + g];
    if (E == 0) break;        // nothing to split: the pass changes nothing (:1136-1139)
    if (g == G) return -1;    // the next generation lies below the pyramid
    const int prev = size;
    size = size - E + stat[PYR_STAT_E + g + 1] + stat[PYR_STAT_F + g + 1];
    ++g;
    if (size >= N || size == prev) break;
    if (size + 3 * stat[PYR_STAT_E + g] > N) {  // :1140
      careful = true;
      break;
    }
  }
  if (careful && g == G) return -1;  // a careful round reads the children's counts
  // ---- nodes of the pass-made generations: single-point nodes are final where they were born; the multi-point nodes of generation g
  //      are final too, or become the first careful round's candidates ----
  // (the levels lie one behind the other in the pyramid: ONE index space over all the bins of generations 0 .. g, so that the few bins of
  // the shallow levels do not each cost the first threads a dependent trip of their own)
  OCT_PHASE_BEGIN
  for (int i = tid; i < base[g] + nb[g]; i += OCT_NT) {
    int k = 0;
#pragma unroll
    for (int d = 1; d <= PYR_MAX_DEPTH; ++d) k += (d <= g && i >= base[d]) ? 1 : 0;
    const int b = i - base[k];
    const uint32_t c = pyr[i];
    if (c == 0 || (k > 0 && pyr[base[k - 1] + (b >> 2)] <= 1)) continue;
    const uint32_t pos = (uint32_t)((k & 1) ? nb[k] - 1 - b : b);
    if (c == 1 || (k == g && !careful)) {
      const int slot = OCT_ATOMIC_ADD(&sc[SC_NOUT], 1);
      w.outKey[slot] = ((uint32_t)(GEN_MAX - k) << 16) | pos;
      w.outPt[slot] = ((uint32_t)k << 16) | (uint32_t)b;
    } else if (k == g) {
      const int slot = OCT_ATOMIC_ADD(&sc[SC_NA], 1);
      keys_in[slot] = ((0xFFFFFu - (c > 0xFFFFFu ? 0xFFFFFu : c)) << 12) | pos;
    }
  }
  OCT_PHASE_END
  // ---- careful rounds (:1143-1204) ----
  bool first = true;
  while (careful) {
    const int nExp = oct_bcast(&sc[SC_NA]);
    if (nExp == 0) break;
    if (g + 1 > G) return -1;
    const int dch = g + 1;  // depth of the children
    int n2 = 1;
    while (n2 < nExp) n2 <<= 1;
    OCT_PHASE_BEGIN
    for (int i = tid; i < pr.Mp2; i += OCT_NT) w.sortbuf[i] = i < nExp ? keys_in[i] : 0xFFFFFFFFu;
    OCT_PHASE_END
    block_sort(w.sortbuf, nExp, n2, w.baseOfRank);
    // parents in processing order: their bin, and the net number of nodes each split adds
    OCT_PHASE_BEGIN
    for (int i = tid; i < nExp; i += OCT_NT) {
      const uint32_t pos = w.sortbuf[i] & 0xfffu;
      const uint32_t bin = first ? (uint32_t)((g & 1) ? nb[g] - 1 - (int)pos : (int)pos) : Acur[pos];
      w.nodeOfRank[i] = bin;
      const uint32_t* c = &pyr[base[dch] + 4 * (int)bin];
      w.baseOfRank[i] = (uint32_t)((c[0] > 0) + (c[1] > 0) + (c[2] > 0) + (c[3] > 0)) - 1u;
    }
    if (tid == 0) sc[SC_M] = nExp;
    OCT_PHASE_END
    (void)block_scan_excl(w.baseOfRank, nExp, w.part, sc);
    OCT_PHASE_BEGIN
    for (int i = tid; i < nExp; i += OCT_NT) {
      const uint32_t* c = &pyr[base[dch] + 4 * (int)w.nodeOfRank[i]];
      const int add = (int)((c[0] > 0) + (c[1] > 0) + (c[2] > 0) + (c[3] > 0)) - 1;
      if (size + (int)w.baseOfRank[i] + add >= N) OCT_ATOMIC_MIN(&sc[SC_M], i + 1);  // :1197-1198: stop at the first split reaching N
    }
    OCT_PHASE_END
    const int nProc = oct_bcast(&sc[SC_M]);
    OCT_PHASE_BEGIN
    for (int i = tid; i < nProc; i += OCT_NT) {
      const uint32_t* c = &pyr[base[dch] + 4 * (int)w.nodeOfRank[i]];
      w.baseOfRank[i] = (uint32_t)((c[0] > 0) + (c[1] > 0) + (c[2] > 0) + (c[3] > 0));
    }
    if (tid == 0) sc[SC_NA] = 0;
    OCT_PHASE_END
    const int T = (int)block_scan_excl(w.baseOfRank, nProc, w.part, sc);
    // children by creation rank (parent processing order, n1..n4), pushed to the front: position = T - 1 - rank
    OCT_PHASE_BEGIN
    for (int i = tid; i < nExp; i += OCT_NT) {
      const uint32_t bin = w.nodeOfRank[i];
      if (i < nProc) {
        uint32_t rank = w.baseOfRank[i];
        for (int d = 0; d < 4; ++d) {
          const uint32_t cb = 4u * bin + (uint32_t)((dch & 1) == 0 ? 3 - d : d);
          const uint32_t c = pyr[base[dch] + (int)cb];
          if (c == 0) continue;
          const uint32_t pos = (uint32_t)(T - 1) - rank;
          ++rank;
          if (c == 1) {
            const int slot = OCT_ATOMIC_ADD(&sc[SC_NOUT], 1);
            w.outKey[slot] = ((uint32_t)(GEN_MAX - dch) << 16) | pos;
            w.outPt[slot] = ((uint32_t)dch << 16) | cb;
          } else {
            const int slot = OCT_ATOMIC_ADD(&sc[SC_NA], 1);
            keys_in[slot] = ((0xFFFFFu - (c > 0xFFFFFu ? 0xFFFFFu : c)) << 12) | pos;
            Anext[pos] = cb;
          }
        }
      } else {  // not reached by the truncated round: stays where it is in generation g
        const int slot = OCT_ATOMIC_ADD(&sc[SC_NOUT], 1);
        w.outKey[slot] = ((uint32_t)(GEN_MAX - g) << 16) | (w.sortbuf[i] & 0xfffu);
        w.outPt[slot] = ((uint32_t)g << 16) | bin;
      }
    }
    OCT_PHASE_END
    const int prev = size;
    size = prev - nProc + T;
    g = dch;
    first = false;
    {
      uint32_t* t = Acur;
      Acur = Anext;
      Anext = t;
    }
    if (size >= N || size == prev) break;  // :1201-1202
  }
  // the multi-point nodes the last careful round created stay as they are
  if (careful) {
    const int nLeft = oct_bcast(&sc[SC_NA]);
    OCT_PHASE_BEGIN
    for (int i = tid; i < nLeft; i += OCT_NT) {
      const uint32_t pos = keys_in[i] & 0xfffu;
      const int slot = OCT_ATOMIC_ADD(&sc[SC_NOUT], 1);
      w.outKey[slot] = ((uint32_t)(GEN_MAX - g) << 16) | pos;
      w.outPt[slot] = ((uint32_t)g << 16) | Acur[pos];
    }
    OCT_PHASE_END
  }

  // ---- list order = (generation desc, position asc): the output slot of every final node is the RANK of its key among the (distinct) keys
  //      -- every node counts the keys below its own (broadcast reads) and takes that slot: no sorted array is ever written, and the path
  //      tables (which share bytes with the sort buffer of the fall-back form) stay as they are ----
  const int nOut = oct_bcast(&sc[SC_NOUT]);
  uint64_t* best = reinterpret_cast<uint64_t*>(w.ccnt);
  OCT_PHASE_BEGIN
  for (int i = tid; i < nOut; i += OCT_NT) {
    const uint32_t key = w.outKey[i], ref = w.outPt[i];
    const int rank = rank_below(w.outKey, nOut, key);
    pyr[base[ref >> 16] + (int)(ref & 0xffffu)] = PYR_FINAL | (uint32_t)rank;
    best[rank] = 0;
  }
  OCT_PHASE_END
  // ---- per final node the best response, first in candidate order on ties (:1208-1226).  The candidate order of the reference
  //      (cell-major, raster inside a cell) is unique per candidate and invertible, so the winner's coordinates come back out of the
  //      word that won: (score << 32) | ~order ----
  // (x - 3) / wCell and (y - 3) / hCell by a multiplication with ceil(2^24 / cell): exact for dividends below 4096 and the cell sizes
  // build_geom accepts (17 .. 66 pixels; fast_geom.hpp) -- an integer division is forty instructions, and this pass does two per candidate
  const bool by_mul = pr.W < 4096 && pr.H < 4096 && pr.wCell >= 17 && pr.wCell <= 66 && pr.hCell >= 17 && pr.hCell <= 66;
  const uint32_t inv_w = by_mul ? (0x1000000u + (uint32_t)pr.wCell - 1u) / (uint32_t)pr.wCell : 0u;
  const uint32_t inv_h = by_mul ? (0x1000000u + (uint32_t)pr.hCell - 1u) / (uint32_t)pr.hCell : 0u;
  // every depth-G bin learns the slot of the final node on its path (a path crosses exactly one; deepest first so that the shallowest
  // wins), in place of its own count -- nobody else reads that word, the shallower levels stay as they are: the pass over the candidates
  // then takes one read per candidate instead of one per depth
  OCT_PHASE_BEGIN
  for (int b = tid; b < nb[G]; b += OCT_NT) {
    uint32_t slot = 0xFFFFFFFFu;
#pragma unroll
    for (int d = PYR_MAX_DEPTH; d >= 0; --d) {
      if (d > G) continue;
      const uint32_t v = pyr[base[d] + (b >> (2 * (G - d)))];
      if (v & PYR_FINAL) slot = v & 0x7FFFFFFFu;
    }
    pyr[base[G] + b] = slot;
  }
  OCT_PHASE_END
  OCT_PHASE_BEGIN
  {
    uint32_t cx[8], cs[8], nx[8], ns[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int q = tid + u * OCT_NT < P ? tid + u * OCT_NT : P - 1;
      cx[u] = cand_xy[q], cs[u] = cand_score[q];
    }
    for (int p0 = tid; p0 < P; p0 += 8 * OCT_NT) {
      const int pn = p0 + 8 * OCT_NT;
#pragma unroll
      for (int u = 0; u < 8; ++u) {  // the next eight while these are placed (unconditional loads: indices clamped into the list)
        const int q = pn + u * OCT_NT < P ? pn + u * OCT_NT : P - 1;
        nx[u] = cand_xy[q], ns[u] = cand_score[q];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (p0 + u * OCT_NT >= P) continue;
        const uint32_t xy = cx[u];
        const uint32_t slot = pyr[base[G] + (int)tbin_of(xy)];
        const int x = (int)(xy & 0xffff), y = (int)(xy >> 16);
        int j = by_mul ? (int)(((uint32_t)(x - 3) * inv_w) >> 24) : (x - 3) / pr.wCell;  // (a candidate lies inside a cell's interior: x, y >= 3)
        int i = by_mul ? (int)(((uint32_t)(y - 3) * inv_h) >> 24) : (y - 3) / pr.hCell;
        j = j > pr.nCols - 1 ? pr.nCols - 1 : j;
        i = i > pr.nRows - 1 ? pr.nRows - 1 : i;
        const uint32_t ord = ((uint32_t)(i * pr.nCols + j) * 128u + (uint32_t)(y - i * pr.hCell)) * 128u + (uint32_t)(x - j * pr.wCell);
        if (slot != 0xFFFFFFFFu) OCT_ATOMIC_MAX64(&best[slot], ((uint64_t)cs[u] << 32) | (uint64_t)(0xFFFFFFFFu - ord));
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) cx[u] = nx[u], cs[u] = ns[u];
    }
  }
  OCT_PHASE_END
  OCT_PHASE_BEGIN
  for (int o = tid; o < nOut && o < sel_cap; o += OCT_NT) {
    const uint64_t v = best[o];
    const uint32_t ord = 0xFFFFFFFFu - (uint32_t)v;
    const int cell = (int)(ord >> 14), i = cell / pr.nCols, j = cell - i * pr.nCols;
    const uint32_t x = (ord & 127u) + (uint32_t)(j * pr.wCell), y = ((ord >> 7) & 127u) + (uint32_t)(i * pr.hCell);
    sel_xy[o] = x | (y << 16);
    sel_score[o] = (uint32_t)(v >> 32);
  }
  OCT_PHASE_END
  return nOut;
}

}  // namespace oct
}  // namespace uvo

