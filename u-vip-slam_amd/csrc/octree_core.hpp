// Quad-tree keypoint selection, one workgroup per (frame, level).
// Replaces ORBextractor::DistributeOctTree + ExtractorNode::DivideNode (src/ORBextractor.cc:1006-1287).
//
// The reference is a sequential std::list algorithm.  It is reformulated here as a sequence of data-parallel
// *passes* over "generations" of nodes (tools/octree_proto.py is the executable derivation, checked against the
// literal std::list restatement the tests use as checker):
//   * a generation A is the set of nodes created by one pass, stored in the reference's list order
//     (front -> back = newest -> oldest); older nodes that survive are frozen single-point nodes, which always
//     sit behind A in the list, generation by generation;
//   * a full pass (:1061-1132) splits every multi-point node of A; children are created in (parent list order,
//     n1..n4) order and push_front'ed, so the new generation in list order is the reverse creation order;
//   * a careful round (:1143-1204) does the same but visits parents by (size desc, newest first) -- the declared
//     deterministic replacement of the reference's pointer-valued tie-break -- and stops at
//     the first split that reaches N nodes; parents it did not reach stay where they are;
//   * the result (:1208-1229) is one point per surviving node: max response, first in candidate order on ties,
//     emitted in list order = (generation desc, position asc).
// A pass is: per-point child digit + histogram (LDS atomics), a scan over <= N parents, and for careful rounds a
// bitonic sort of <= N keys.  Point state lives in HBM scratch (L2 resident), node state in LDS.
//
// k_octree tries the closed form over a count pyramid first (octree_pyramid.hpp, included by the users of this header: it needs no
// pass over the points per generation); this formulation handles trees deeper than the pyramid.
//
// The body is written against the OCT_* phase macros so that tests/emu/octree_emu.cpp can run the *same* logic on
// the CPU (threads of a phase executed one after another) against that checker; on the GPU a phase ends in a
// workgroup barrier.
#pragma once
#include <cstring>
#include <stdint.h>

#ifndef OCT_THREADS
#define OCT_THREADS 256
#endif

#if defined(UVO_OCT_TRACE) && defined(__HIPCC__)
__device__ unsigned long long g_oct_trace[2048];
__device__ int g_oct_trace_n;
__device__ unsigned long long g_oct_blocks[4096];
#endif

#if defined(__HIP_DEVICE_COMPILE__)
#define OCT_DEVICE 1
#define OCT_NT ((int)blockDim.x)  // workgroup size chosen at launch: 256 (latency) or 64 (one wavefront per problem, throughput)
#define OCT_FN __device__ __forceinline__
#define OCT_PHASE_BEGIN \
  {                     \
    const int tid = threadIdx.x;
#ifdef UVO_OCT_TRACE  // developer build only (UVO_EXTRA_FLAGS=-DUVO_OCT_TRACE): per-phase time stamps of one workgroup
#define OCT_TRACE_MARK()                                                                              \
  if (threadIdx.x == 0 && blockIdx.y == 0 && blockIdx.x == (gridDim.y > 1 ? gridDim.x / 2 : 0) && g_oct_trace_n < 1024) { /* k_octree: level 0 of the middle frame; k_octree_gauss: problem 0 */ \
    g_oct_trace[2 * g_oct_trace_n] = __LINE__;                                                        \
    g_oct_trace[2 * g_oct_trace_n + 1] = wall_clock64();                                              \
    ++g_oct_trace_n;                                                                                  \
  }
#else
#define OCT_TRACE_MARK()
#endif
#define OCT_PHASE_END \
  }                   \
  __syncthreads();    \
  OCT_TRACE_MARK()
#define OCT_ATOMIC_ADD(p, v) atomicAdd((p), (v))
#define OCT_ATOMIC_MIN(p, v) atomicMin((p), (v))
#define OCT_ATOMIC_MAX64(p, v) atomicMax((unsigned long long*)(p), (unsigned long long)(v))
#else
#define OCT_DEVICE 0
#define OCT_NT OCT_THREADS
#if defined(__HIPCC__)  // host pass of a HIP translation unit: the kernel body naming these functions is still parsed
#define OCT_FN __host__ __device__ static inline
#else
#define OCT_FN static inline
#endif
#define OCT_PHASE_BEGIN for (int tid = 0; tid < OCT_THREADS; ++tid) {
#define OCT_PHASE_END }
#define OCT_TRACE_MARK()
template <class T, class U>
static inline T oct_host_add(T* p, U v) {
  T o = *p;
  *p = (T)(o + v);
  return o;
}
#define OCT_ATOMIC_ADD(p, v) oct_host_add((p), (v))
#define OCT_ATOMIC_MIN(p, v)        \
  do {                              \
    if ((v) < *(p)) *(p) = (v);     \
  } while (0)
#define OCT_ATOMIC_MAX64(p, v)                            \
  do {                                                    \
    if ((uint64_t)(v) > *(p)) *(p) = (uint64_t)(v);       \
  } while (0)
#endif

namespace uvo {
namespace oct {

struct Params {
  int P;            // candidates
  int N;            // target node count (level quota)
  int W, H;         // detection window (maxBorder - minBorder)
  int nIni;
  float hX;
  int nCols, nRows, wCell, hCell;  // FAST cell grid, for the candidate-order key
  int M;            // node capacity of the LDS arrays
  int Mp2;          // power of two >= M
};

struct Box {
  uint16_t ulx, urx, uly, bry;
};

// LDS (or host) working set; all arrays sized by Params::M unless noted
struct Work {
  Box* boxA;
  Box* boxB;
  uint32_t* cntA;
  uint32_t* cntB;
  int32_t* procRank;     // per node of A: rank in processing order, or -1
  uint32_t* ccnt;        // [4*M] child histogram (two buffers, swapped every generation; aliased later by best64 / sort64)
  uint32_t* ccnt2;       // [4*M]
  uint32_t* nodeOfRank;  // processing rank -> node position in A
  uint32_t* baseOfRank;  // exclusive scan of non-empty child counts in processing order
  uint32_t* sortbuf;     // [Mp2]
  uint32_t* outKey;      // per emitted node: ((GEN_MAX - gen) << 16) | pos
  uint32_t* outPt;       // candidate index
  uint32_t* part;        // [OCT_THREADS] scan partials
  // shared scalars
  int* sc;               // [16]
  // count pyramid of the closed-form path (octree_pyramid.hpp)
  uint32_t* pyr;         // [pyramid_words(nIni)]
  int* stat;             // [16]
  uint16_t* tab;         // path tables of the closed-form path (x part, then y part), tab_cap entries; on the device they share the
  int tab_cap;           // bytes of ccnt / ccnt2 behind the first 2M words (the closed form writes nothing there: `best` lives in front of them)
  const uint16_t* tab_src;  // the same tables ready-made in memory (a constant of the level's geometry: the kernel copies them), or null: computed here
};

enum { SC_NA = 0, SC_NOUT, SC_NEXP, SC_NPROC, SC_T, SC_NTOEXP, SC_M, SC_TMP };

constexpr uint32_t ST_FROZEN = 0xFFFFFFFFu;
constexpr uint32_t ST_UNPROC = 0x80000000u;
constexpr int GEN_MAX = 60;

OCT_FN int oct_bcast(const int* p) {
  int v = *p;
#if OCT_DEVICE
  __syncthreads();
#endif
  return v;
}

// in-place exclusive scan of a[0..n) (n <= a few thousand); returns the total.
// Same result on both builds; the device build scans the per-thread partial sums with wavefront shuffles
// (2 workgroup barriers), the host build (test emulation) with a Kogge-Stone pass over the OCT_NT partials.
OCT_FN uint32_t block_scan_excl(uint32_t* a, int n, uint32_t* part, int* sc) {
  const int per = (n + OCT_NT - 1) / OCT_NT;
#if OCT_DEVICE
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  uint32_t s = 0;
  const int b = tid * per, e = (b + per < n) ? b + per : n;
  for (int i = b; i < e; ++i) s += a[i];
  uint32_t incl = s;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t v = (uint32_t)__shfl_up((int)incl, off, 64);
    if (lane >= off) incl += v;
  }
  if (lane == 63) part[wv] = incl;
  __syncthreads();
  uint32_t woff = 0, total = 0;
#pragma unroll
  for (int k = 0; k < OCT_NT / 64; ++k) {
    const uint32_t t = part[k];
    if (k < wv) woff += t;
    total += t;
  }
  uint32_t run = woff + incl - s;
  for (int i = b; i < e; ++i) {
    const uint32_t v = a[i];
    a[i] = run;
    run += v;
  }
  __syncthreads();
  OCT_TRACE_MARK()
  (void)sc;
  return total;
#else
  OCT_PHASE_BEGIN
  uint32_t s = 0;
  const int b = tid * per, e = (b + per < n) ? b + per : n;
  for (int i = b; i < e; ++i) s += a[i];
  part[tid] = s;
  OCT_PHASE_END
  for (int off = 1; off < OCT_THREADS; off <<= 1) {
    uint32_t* tmp = part + OCT_THREADS;
    OCT_PHASE_BEGIN
    tmp[tid] = part[tid] + (tid >= off ? part[tid - off] : 0u);
    OCT_PHASE_END
    OCT_PHASE_BEGIN
    part[tid] = tmp[tid];
    OCT_PHASE_END
  }
  OCT_PHASE_BEGIN
  uint32_t run = tid ? part[tid - 1] : 0u;
  const int b = tid * per, e = (b + per < n) ? b + per : n;
  for (int i = b; i < e; ++i) {
    const uint32_t v = a[i];
    a[i] = run;
    run += v;
  }
  if (tid == OCT_THREADS - 1) sc[SC_TMP] = (int)part[OCT_THREADS - 1];
  OCT_PHASE_END
  return (uint32_t)oct_bcast(&sc[SC_TMP]);
#endif
}

// ascending sort of the n distinct keys a[0..n); a[n..n2) must be padded with the maximum value (n2 = power of two).
// Device: rank sort (every key counts the keys below it -- LDS broadcast reads, one pass, two barriers);
// host emulation: bitonic network.  tmp holds n elements.
// number of keys of a[0..n) below x; a is 16-byte aligned (the node tables are: oct_capacity rounds M to a multiple of 4), so the keys come
// four per LDS read (every thread reads the same address: a broadcast)
struct alignas(16) Key4 {
  uint32_t x, y, z, w;
};
OCT_FN Key4 key4_at(const uint32_t* p) {
#if OCT_DEVICE
  return *reinterpret_cast<const Key4*>(p);
#else
  Key4 k;
  memcpy(&k, p, sizeof(k));
  return k;
#endif
}
OCT_FN int key4_below(const Key4& k, uint32_t x) { return (k.x < x ? 1 : 0) + (k.y < x ? 1 : 0) + (k.z < x ? 1 : 0) + (k.w < x ? 1 : 0); }
OCT_FN int rank_below(const uint32_t* a, int n, uint32_t x) {
  int rank = 0, j = 0;
  // sixteen keys per trip, their four reads in flight together: one wavefront per SIMD runs this, nothing else hides an LDS round trip
  for (; j + 16 <= n; j += 16) {
    const Key4 k0 = key4_at(a + j), k1 = key4_at(a + j + 4), k2 = key4_at(a + j + 8), k3 = key4_at(a + j + 12);
    rank += key4_below(k0, x) + key4_below(k1, x) + key4_below(k2, x) + key4_below(k3, x);
  }
  for (; j + 4 <= n; j += 4) rank += key4_below(key4_at(a + j), x);
  for (; j < n; ++j) rank += a[j] < x ? 1 : 0;
  return rank;
}
template <class T>
OCT_FN int rank_below(const T* a, int n, T x) {
  int rank = 0;
  for (int j = 0; j < n; ++j) rank += a[j] < x;
  return rank;
}
template <class T>
OCT_FN void block_sort(T* a, int n, int n2, T* tmp) {
#if OCT_DEVICE
  (void)n2;
  for (int i = threadIdx.x; i < n; i += OCT_NT) {
    const T x = a[i];
    tmp[rank_below(a, n, x)] = x;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += OCT_NT) a[i] = tmp[i];
  __syncthreads();
  OCT_TRACE_MARK()
#else
  (void)tmp;
  (void)n;
  for (int k = 2; k <= n2; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      OCT_PHASE_BEGIN
      for (int i = tid; i < n2; i += OCT_NT) {
        const int ixj = i ^ j;
        if (ixj > i) {
          const T x = a[i], y = a[ixj];
          const bool up = (i & k) == 0;
          if ((x > y) == up) {
            a[i] = y;
            a[ixj] = x;
          }
        }
      }
      OCT_PHASE_END
    }
  }
#endif
}

OCT_FN int half_ceil(int a) { return (a + 1) >> 1; }  // ceil(a/2.f) for a >= 0 (DivideNode :1233-1234)

// Per-candidate state.  K > 0: candidate p = tid + k * OCT_NT lives in slot k of the owning thread's register arrays (the
// kernel picks the smallest K with P <= K * OCT_NT), so a pass over the points touches no memory but the LDS node tables.
// K == 0: fallback for longer lists -- coordinates are re-read from cand_xy and the state word lives in pstate (HBM/L2).
// State word of a live point: node position | child digit << 16; ST_UNPROC | position; ST_FROZEN.
template <int K>
struct PointRegs {
#if OCT_DEVICE
  uint32_t xy[K > 0 ? K : 1], st[K > 0 ? K : 1];
#else
  uint32_t xy[OCT_THREADS][K > 0 ? K : 1], st[OCT_THREADS][K > 0 ? K : 1];
#endif
};
#if OCT_DEVICE
#define OCT_PR(arr, k) arr[k]
#else
#define OCT_PR(arr, k) arr[tid][k]
#endif
// loop over the calling thread's candidates: k = slot, p = candidate index
#define OCT_POINTS_BEGIN                                           \
  _Pragma("unroll") for (int k = 0; k < (K > 0 ? K : kRt); ++k) { \
    const int p = tid + k * OCT_NT;                                \
    if (p >= P) break;
#define OCT_POINTS_END }
#define PT_XY() (K > 0 ? OCT_PR(R.xy, k) : cand_xy[p])
#define PT_ST() (K > 0 ? OCT_PR(R.st, k) : pstate[p])
#define PT_SET(v)                \
  do {                           \
    if (K > 0)                   \
      OCT_PR(R.st, k) = (v);     \
    else                         \
      pstate[p] = (v);           \
  } while (0)

OCT_FN uint32_t child_digit(const Box& b, uint32_t xy) {
  const int x = (int)(xy & 0xffff), y = (int)(xy >> 16);
  return (uint32_t)((x < (int)b.ulx + half_ceil((int)b.urx - (int)b.ulx) ? 0 : 1) + (y < (int)b.uly + half_ceil((int)b.bry - (int)b.uly) ? 0 : 2));
}

// cand_xy / cand_score: candidate coordinates (x | y<<16, relative to minBorder) and FAST scores
// pstate: per-candidate scratch word (used by K == 0 only).  sel_*: output in list order.  returns number of selected points.
template <int K>
OCT_FN int run(const Params& pr, const Work& w, const uint32_t* cand_xy, const uint32_t* cand_score, uint32_t* pstate, uint32_t* sel_xy,
               uint32_t* sel_score, int sel_cap) {
  const int P = pr.P, N = pr.N;
  const int kRt = (P + OCT_NT - 1) / OCT_NT;
  (void)kRt;
  int* sc = w.sc;
  Box* A = w.boxA;
  Box* B = w.boxB;
  uint32_t* cA = w.cntA;
  uint32_t* cB = w.cntB;
  uint32_t* H = w.ccnt;    // child histogram of the current generation A (filled by the previous pass over the points)
  uint32_t* Hn = w.ccnt2;  // child histogram of the generation being built
  PointRegs<K> R;

  // ---- roots (:1010-1052) ----
  OCT_PHASE_BEGIN
  for (int i = tid; i < pr.nIni; i += OCT_NT) Hn[i] = 0;
  for (int i = tid; i < 4 * pr.M; i += OCT_NT) H[i] = 0;
  if (tid == 0) sc[SC_NOUT] = 0;
  OCT_PHASE_END
  OCT_PHASE_BEGIN
  // every candidate falls into one of very few roots (nIni = round(W/H)): count the first four per thread and add once,
  // instead of P atomics serialising on the same LDS word
  uint32_t loc[4] = {0, 0, 0, 0};
  OCT_POINTS_BEGIN
  const uint32_t xy = cand_xy[p];
  if (K > 0) OCT_PR(R.xy, k) = xy;
  const float x = (float)(xy & 0xffff);
  const int r = (int)(x / pr.hX);
  if (r < 4) {
#pragma unroll
    for (int q = 0; q < 4; ++q) loc[q] += r == q;
  } else {
    OCT_ATOMIC_ADD(&Hn[r], 1u);
  }
  PT_SET((uint32_t)r);
  OCT_POINTS_END
#pragma unroll
  for (int q = 0; q < 4; ++q)
    if (loc[q]) OCT_ATOMIC_ADD(&Hn[q], loc[q]);
  OCT_PHASE_END
  OCT_PHASE_BEGIN
  if (tid == 0) {
    int na = 0;
    for (int r = 0; r < pr.nIni; ++r) {
      if (Hn[r] == 0) {
        w.nodeOfRank[r] = 0;
        continue;
      }
      w.nodeOfRank[r] = (uint32_t)na;
      Box b;
      b.ulx = (uint16_t)(int)(pr.hX * (float)r);
      b.urx = (uint16_t)(int)(pr.hX * (float)(r + 1));
      b.uly = 0;
      b.bry = (uint16_t)pr.H;
      A[na] = b;
      cA[na] = Hn[r];
      ++na;
    }
    sc[SC_NA] = na;
  }
  OCT_PHASE_END
  // root assignment, fused with the child digit + histogram of generation 0 (DivideNode :1262-1276)
  OCT_PHASE_BEGIN
  uint32_t loc[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // children of root nodes 0 and 1, counted per thread (see above)
  OCT_POINTS_BEGIN
  const uint32_t a = w.nodeOfRank[PT_ST()];
  if (cA[a] == 1) {
    const int slot = OCT_ATOMIC_ADD(&sc[SC_NOUT], 1);
    w.outKey[slot] = ((uint32_t)(GEN_MAX - 0) << 16) | a;
    w.outPt[slot] = (uint32_t)p;
    PT_SET(ST_FROZEN);
  } else {
    const uint32_t d = child_digit(A[a], PT_XY());
    const uint32_t h = 4 * a + d;
    if (h < 8) {
#pragma unroll
      for (int q = 0; q < 8; ++q) loc[q] += h == (uint32_t)q;
    } else {
      OCT_ATOMIC_ADD(&H[h], 1u);
    }
    PT_SET(a | (d << 16));
  }
  OCT_POINTS_END
#pragma unroll
  for (int q = 0; q < 8; ++q)
    if (loc[q]) OCT_ATOMIC_ADD(&H[q], loc[q]);
  OCT_PHASE_END

  int na = oct_bcast(&sc[SC_NA]);
  int size = na;
  int gen = 0;
  bool careful = false;
  int na_prev = 0;          // node count of the previous generation (only meaningful after a truncated round)
  bool truncated = false;

  for (;;) {
    const int prev_size = size;
    // ---- which nodes of A are expandable; default processing order = list order ----
    OCT_PHASE_BEGIN
    for (int a = tid; a < na; a += OCT_NT) w.baseOfRank[a] = cA[a] > 1 ? 1u : 0u;
    for (int i = tid; i < 4 * pr.M; i += OCT_NT) Hn[i] = 0;
    OCT_PHASE_END
    const int nExp = (int)block_scan_excl(w.baseOfRank, na, w.part, sc);
    if (nExp == 0) break;
    OCT_PHASE_BEGIN
    for (int a = tid; a < na; a += OCT_NT) {
      if (cA[a] > 1) {
        const uint32_t r = w.baseOfRank[a];
        w.procRank[a] = (int32_t)r;
        w.nodeOfRank[r] = (uint32_t)a;
      } else {
        w.procRank[a] = -1;
      }
    }
    OCT_PHASE_END
    int nProc = nExp;
    if (careful) {
      // ---- order parents by (size desc, list position asc) and cut at the first split reaching N (:1151-1199) ----
      OCT_PHASE_BEGIN
      for (int i = tid; i < pr.Mp2; i += OCT_NT) {
        uint32_t key = 0xFFFFFFFFu;
        if (i < nExp) {
          const uint32_t a = w.nodeOfRank[i];
          uint32_t c = cA[a];
          c = c > 0xFFFFFu ? 0xFFFFFu : c;
          key = ((0xFFFFFu - c) << 12) | a;
        }
        w.sortbuf[i] = key;
      }
      OCT_PHASE_END
      int n2 = 1;
      while (n2 < nExp) n2 <<= 1;
      block_sort(w.sortbuf, nExp, n2, w.baseOfRank);
      OCT_PHASE_BEGIN
      for (int i = tid; i < nExp; i += OCT_NT) {
        const uint32_t a = w.sortbuf[i] & 0xfffu;
        w.nodeOfRank[i] = a;
        const uint32_t* c = &H[4 * a];
        w.baseOfRank[i] = (uint32_t)((c[0] > 0) + (c[1] > 0) + (c[2] > 0) + (c[3] > 0)) - 1u;  // net nodes added by this split
      }
      if (tid == 0) sc[SC_M] = nExp;
      OCT_PHASE_END
      (void)block_scan_excl(w.baseOfRank, nExp, w.part, sc);
      OCT_PHASE_BEGIN
      for (int i = tid; i < nExp; i += OCT_NT) {
        const uint32_t a = w.nodeOfRank[i];
        const uint32_t* c = &H[4 * a];
        const int add = (int)((c[0] > 0) + (c[1] > 0) + (c[2] > 0) + (c[3] > 0)) - 1;
        if (size + (int)w.baseOfRank[i] + add >= N) OCT_ATOMIC_MIN(&sc[SC_M], i + 1);
      }
      OCT_PHASE_END
      nProc = oct_bcast(&sc[SC_M]);
      OCT_PHASE_BEGIN
      for (int i = tid; i < nExp; i += OCT_NT) w.procRank[w.nodeOfRank[i]] = i < nProc ? i : -1;
      OCT_PHASE_END
    }
    // ---- creation rank of every child: exclusive scan of non-empty-child counts in processing order ----
    OCT_PHASE_BEGIN
    for (int i = tid; i < nProc; i += OCT_NT) {
      const uint32_t* c = &H[4 * w.nodeOfRank[i]];
      w.baseOfRank[i] = (uint32_t)((c[0] > 0) + (c[1] > 0) + (c[2] > 0) + (c[3] > 0));
    }
    if (tid == 0) sc[SC_NTOEXP] = 0;
    OCT_PHASE_END
    const int T = (int)block_scan_excl(w.baseOfRank, nProc, w.part, sc);
    // ---- build the new generation B in list order: position = T-1-creation rank ----
    OCT_PHASE_BEGIN
    for (int i = tid; i < nProc; i += OCT_NT) {
      const uint32_t a = w.nodeOfRank[i];
      const Box b = A[a];
      const int hx = half_ceil((int)b.urx - (int)b.ulx), hy = half_ceil((int)b.bry - (int)b.uly);
      uint32_t rank = w.baseOfRank[i];
      int nmulti = 0;
      for (int d = 0; d < 4; ++d) {
        const uint32_t c = H[4 * a + d];
        if (c == 0) continue;
        const int pos = T - 1 - (int)rank;
        Box nb;
        nb.ulx = (uint16_t)((d & 1) ? b.ulx + hx : b.ulx);
        nb.urx = (uint16_t)((d & 1) ? b.urx : b.ulx + hx);
        nb.uly = (uint16_t)((d & 2) ? b.uly + hy : b.uly);
        nb.bry = (uint16_t)((d & 2) ? b.bry : b.uly + hy);
        B[pos] = nb;
        cB[pos] = c;
        nmulti += c > 1;
        ++rank;
      }
      if (nmulti) OCT_ATOMIC_ADD(&sc[SC_NTOEXP], nmulti);
    }
    OCT_PHASE_END
    // ---- move points to their child node; freeze single-point children; points that stay live get their child digit
    //      inside the new node straight away (histogram of the next generation) ----
    const int newgen = gen + 1;
    OCT_PHASE_BEGIN
    // two sweeps so that the table look-ups of all of a thread's candidates can be in flight together: first the new node
    // position of every live point (reads only), then the freeze / next-digit step (atomics)
    OCT_POINTS_BEGIN
    const uint32_t st = PT_ST();
    if (st != ST_FROZEN) {
      const uint32_t a = st & 0xffffu;
      const int d = (int)(st >> 16);
      const int r = w.procRank[a];
      if (r < 0) {
        PT_SET(ST_UNPROC | a);  // parent not reached by a truncated careful round (always the last round)
      } else {
        const uint32_t* c = &H[4 * a];
        const int lower = (int)((d > 0 && c[0] > 0) + (d > 1 && c[1] > 0) + (d > 2 && c[2] > 0));
        PT_SET((uint32_t)(T - 1 - ((int)w.baseOfRank[r] + lower)));
      }
    }
    OCT_POINTS_END
    OCT_POINTS_BEGIN
    const uint32_t st = PT_ST();
    if (st != ST_FROZEN && !(st & ST_UNPROC)) {
      const int pos = (int)st;
      if (cB[pos] == 1) {
        const int slot = OCT_ATOMIC_ADD(&sc[SC_NOUT], 1);
        w.outKey[slot] = ((uint32_t)(GEN_MAX - newgen) << 16) | (uint32_t)pos;
        w.outPt[slot] = (uint32_t)p;
        PT_SET(ST_FROZEN);
      } else {
        const uint32_t nd = child_digit(B[pos], PT_XY());
        OCT_ATOMIC_ADD(&Hn[4 * pos + nd], 1u);
        PT_SET((uint32_t)pos | (nd << 16));
      }
    }
    OCT_POINTS_END
    OCT_PHASE_END
    const int nToExpand = oct_bcast(&sc[SC_NTOEXP]);
    size = prev_size - nProc + T;
    truncated = nProc < nExp;
    na_prev = na;
    na = T;
    gen = newgen;
    {
      Box* tb = A;
      A = B;
      B = tb;
      uint32_t* tc = cA;
      cA = cB;
      cB = tc;
      uint32_t* th = H;
      H = Hn;
      Hn = th;
    }
    if (size >= N || size == prev_size) break;  // :1136-1139 / :1201-1202
    if (!careful && size + 3 * nToExpand > N) careful = true;  // :1140
  }

  // ---- surviving multi-point nodes: best response, first in candidate order (:1208-1226) ----
  uint64_t* best = reinterpret_cast<uint64_t*>(w.ccnt);
  const int nslots = na + (truncated ? na_prev : 0);
  OCT_PHASE_BEGIN
  for (int i = tid; i < nslots; i += OCT_NT) best[i] = 0;
  OCT_PHASE_END
  // candidate order of the reference (cell-major, raster inside a cell), inverted so that "first" is the larger word; it is
  // unique per candidate, so the winner of a slot is recognised by this word alone
  auto point_ord = [&](uint32_t xy) -> uint32_t {
    const int x = (int)(xy & 0xffff), y = (int)(xy >> 16);
    int j = (x - 3) / pr.wCell, i = (y - 3) / pr.hCell;
    j = j > pr.nCols - 1 ? pr.nCols - 1 : j;
    i = i > pr.nRows - 1 ? pr.nRows - 1 : i;
    const uint32_t ord = ((uint32_t)(i * pr.nCols + j) * 128u + (uint32_t)(y - i * pr.hCell)) * 128u + (uint32_t)(x - j * pr.wCell);
    return 0xFFFFFFFFu - ord;
  };
  OCT_PHASE_BEGIN
  OCT_POINTS_BEGIN
  const uint32_t st = PT_ST();
  if (st != ST_FROZEN) {
    const int slot = (st & ST_UNPROC) ? na + (int)(st & 0xffffu) : (int)(st & 0xffffu);
    OCT_ATOMIC_MAX64(&best[slot], ((uint64_t)cand_score[p] << 32) | (uint64_t)point_ord(PT_XY()));
  }
  OCT_POINTS_END
  OCT_PHASE_END
  OCT_PHASE_BEGIN
  OCT_POINTS_BEGIN
  const uint32_t st = PT_ST();
  if (st != ST_FROZEN) {
    const bool un = (st & ST_UNPROC) != 0;
    const int slot = un ? na + (int)(st & 0xffffu) : (int)(st & 0xffffu);
    if ((uint32_t)best[slot] == point_ord(PT_XY())) {
      const int o = OCT_ATOMIC_ADD(&sc[SC_NOUT], 1);
      w.outKey[o] = ((uint32_t)(GEN_MAX - (un ? gen - 1 : gen)) << 16) | (st & 0xffffu);
      w.outPt[o] = (uint32_t)p;
    }
  }
  OCT_POINTS_END
  OCT_PHASE_END

  // ---- list order = (generation desc, position asc) ----
  const int nOut = oct_bcast(&sc[SC_NOUT]);
  uint64_t* srt = reinterpret_cast<uint64_t*>(w.ccnt2);
  int n2 = 1;
  while (n2 < nOut) n2 <<= 1;
  OCT_PHASE_BEGIN
  for (int i = tid; i < n2; i += OCT_NT) srt[i] = i < nOut ? (((uint64_t)w.outKey[i] << 32) | w.outPt[i]) : ~0ull;
  OCT_PHASE_END
  block_sort(srt, nOut, n2, reinterpret_cast<uint64_t*>(w.ccnt));  // both histograms are dead by now
  OCT_PHASE_BEGIN
  for (int i = tid; i < nOut && i < sel_cap; i += OCT_NT) {
    const uint32_t p = (uint32_t)(srt[i] & 0xffffffffu);
    sel_xy[i] = cand_xy[p];
    sel_score[i] = cand_score[p];
  }
  OCT_PHASE_END
  return nOut;
}

#undef OCT_POINTS_BEGIN
#undef OCT_POINTS_END
#undef PT_XY
#undef PT_ST
#undef PT_SET
#undef OCT_PR

}  // namespace oct
}  // namespace uvo
