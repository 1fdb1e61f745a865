// Host execution of the quad-tree kernel body (u-vip-slam_amd/csrc/octree_core.hpp): the OCT_* phase macros run
// the 256 "threads" of each phase one after another, so the exact selection logic that the HIP kernel runs can
// be checked against the oracle on a machine without a GPU.  Test scaffolding only.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../u-vip-slam_amd/csrc/octree_pyramid.hpp"

// k_regs: candidates per thread held in "registers" (8 or 32, as the kernel instantiates), 0 = memory-resident state,
// -1 = pick like the kernel does.  algo: 0 = pass per generation (octree_core.hpp) only, 1 = closed form over the count pyramid
// (octree_pyramid.hpp) only -- returns -1 when the tree is deeper than the pyramid --, 2 = as the kernel: pyramid, else fall back.
extern "C" int emu_octree_algo(const uint32_t* cand_xy, const uint32_t* cand_score, int P, int N, int W, int H, int nCols, int nRows, int wCell,
                            int hCell, uint32_t* sel_xy, uint32_t* sel_score, int sel_cap, int k_regs, int algo) {
  using namespace uvo::oct;
  if (P == 0) return 0;
  Params pr;
  pr.P = P, pr.N = N, pr.W = W, pr.H = H;
  pr.nIni = (int)roundf((float)W / (float)H);
  pr.hX = (float)W / (float)pr.nIni;
  pr.nCols = nCols, pr.nRows = nRows, pr.wCell = wCell, pr.hCell = hCell;
  int M = (N > 4 * pr.nIni ? N : 4 * pr.nIni) + 8;
  int Mp2 = 1;
  while (Mp2 < M) Mp2 <<= 1;
  pr.M = M, pr.Mp2 = Mp2;
  std::vector<uint64_t> ccnt(2 * (size_t)M + Mp2), ccnt2(2 * (size_t)M + Mp2);  // 8-byte aligned, >= 4M u32
  std::vector<Box> boxA(M), boxB(M);
  std::vector<uint32_t> cntA(M), cntB(M), nodeOfRank(M), baseOfRank(M), sortbuf(Mp2), outKey(M), outPt(M), part(2 * OCT_THREADS);
  std::vector<int32_t> procRank(M);
  std::vector<int> sc(16, 0);
  std::vector<uint32_t> pstate(P), pyr(pyramid_words(pr.nIni) + 1);
  std::vector<uint16_t> tab(12 * (size_t)M);  // as on the device: 24 * M bytes
  std::vector<int> stat(16, 0);
  Work w;
  w.boxA = boxA.data(), w.boxB = boxB.data(), w.cntA = cntA.data(), w.cntB = cntB.data(), w.procRank = procRank.data();
  w.ccnt = reinterpret_cast<uint32_t*>(ccnt.data());
  w.ccnt2 = reinterpret_cast<uint32_t*>(ccnt2.data());
  w.nodeOfRank = nodeOfRank.data(), w.baseOfRank = baseOfRank.data(), w.sortbuf = sortbuf.data();
  w.outKey = outKey.data(), w.outPt = outPt.data(), w.part = part.data(), w.sc = sc.data();
  w.pyr = pyr.data(), w.stat = stat.data();
  w.tab = tab.data(), w.tab_cap = (int)tab.size(), w.tab_src = nullptr;
  if (k_regs < 0) k_regs = P <= 8 * OCT_THREADS ? 8 : (P <= 32 * OCT_THREADS ? 32 : 0);
  if (k_regs > 0 && P > k_regs * OCT_THREADS) return -1;
  int n = -1;
  if (algo >= 1) {
    n = run_pyramid(pr, w, cand_xy, cand_score, sel_xy, sel_score, sel_cap);
    if (n >= 0 || algo == 1) return n < 0 ? -2 : n;   // -2: deeper than the pyramid
  }
  if (k_regs == 8) return run<8>(pr, w, cand_xy, cand_score, pstate.data(), sel_xy, sel_score, sel_cap);
  if (k_regs == 32) return run<32>(pr, w, cand_xy, cand_score, pstate.data(), sel_xy, sel_score, sel_cap);
  return run<0>(pr, w, cand_xy, cand_score, pstate.data(), sel_xy, sel_score, sel_cap);
}

extern "C" int emu_octree_k(const uint32_t* cand_xy, const uint32_t* cand_score, int P, int N, int W, int H, int nCols, int nRows, int wCell,
                            int hCell, uint32_t* sel_xy, uint32_t* sel_score, int sel_cap, int k_regs) {
  return emu_octree_algo(cand_xy, cand_score, P, N, W, H, nCols, nRows, wCell, hCell, sel_xy, sel_score, sel_cap, k_regs, 2);
}

extern "C" int emu_octree(const uint32_t* cand_xy, const uint32_t* cand_score, int P, int N, int W, int H, int nCols, int nRows, int wCell,
                          int hCell, uint32_t* sel_xy, uint32_t* sel_score, int sel_cap) {
  return emu_octree_k(cand_xy, cand_score, P, N, W, H, nCols, nRows, wCell, hCell, sel_xy, sel_score, sel_cap, -1);
}
