// Generic search engine behind the ORBmatcher Search* / Fuse entry points (src/ORBmatcher.cc).
//
// Every one of those loops has the same shape: queries are visited in a fixed order; each query owns a candidate list
// (keypoints in a grid window, or the keypoints sharing its vocabulary node); the 256-bit Hamming distance to every
// candidate is taken; an acceptance rule picks at most one target; and (except in Fuse) a target taken by an earlier query
// is skipped by later ones.  Here:
//   k_win_cand<FILL> : candidate lists from grid windows -- FrameKTL::GetFeaturesInArea (src/FrameKTL.cc:359-424) /
//                      KeyFrame::GetFeaturesInArea (src/KeyFrame.cc:952-992), in their (ix, iy, insertion) order
//   k_group_dist     : distances (and the epipolar predicate of SearchForTriangulation) for caller-given candidate lists
//   k_match_resolve  : the order-dependent loop solved exactly as a fixed point -- owner[t] = lowest-index query whose
//                      accepted choice is target t; query i may not use t when owner[t] < i.  Query 0 is final after one
//                      sweep, query i after at most i+1, so the iteration ends in the sequential result.
//   k_rot_filter     : rotation-consistency histogram + ComputeThreeMaxima (src/ORBmatcher.cc:1748-1789)
// packed candidate: target index (16 bits) | distance (9 bits) << 16 | octave (6 bits) << 25 | predicate << 31
#include "common.hpp"
#include "uvo_math.hpp"

namespace uvo {

constexpr int GR_COLS = 64, GR_ROWS = 48;  // include/FrameKTL.h:45-46
constexpr int HISTO_LENGTH = 30;           // src/ORBmatcher.cc:42

struct WinFrame {
  const uvo_keypoint* kp;
  const uint8_t* desc;
  int n;
  int min_x, min_y;
  float inv_w, inv_h;
};
struct WinQuery {
  const float *x, *y, *r;
  const int32_t *min_level, *max_level;
  const uint8_t* valid;
  const uint8_t* desc;
  int n;
};

__device__ __forceinline__ int ham256(const uint8_t* a, const uint8_t* b) {
  const uint4* A = reinterpret_cast<const uint4*>(a);
  const uint4* B = reinterpret_cast<const uint4*>(b);
  const uint4 a0 = A[0], a1 = A[1], b0 = B[0], b1 = B[1];
  return __popc(a0.x ^ b0.x) + __popc(a0.y ^ b0.y) + __popc(a0.z ^ b0.z) + __popc(a0.w ^ b0.w) + __popc(a1.x ^ b1.x) + __popc(a1.y ^ b1.y) +
         __popc(a1.z ^ b1.z) + __popc(a1.w ^ b1.w);
}

// FILL = false: count candidates per query; FILL = true: write packed candidates at cand_start[i]
template <bool FILL>
__global__ __launch_bounds__(256) void k_win_cand(WinFrame F, WinQuery Q, const int32_t* __restrict__ cell_start,
                                                  const int32_t* __restrict__ cell_items, int32_t* __restrict__ cand_cnt,
                                                  const int32_t* __restrict__ cand_start, uint32_t* __restrict__ cand) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= Q.n) return;
  int n = 0;
  if (Q.valid[i]) {
    const float x = Q.x[i], y = Q.y[i], r = Q.r[i];
    const int minLevel = Q.min_level[i], maxLevel = Q.max_level[i];
    // level filter of FrameKTL::GetFeaturesInArea :384-390,:403-413
    const bool bCheckLevels = !(minLevel == -1 && maxLevel == -1);
    const bool bSameLevel = bCheckLevels && minLevel == maxLevel;
    int x0 = (int)floorf((x - (float)F.min_x - r) * F.inv_w);
    x0 = max(0, x0);
    int x1 = (int)ceilf((x - (float)F.min_x + r) * F.inv_w);
    x1 = min(GR_COLS - 1, x1);
    int y0 = (int)floorf((y - (float)F.min_y - r) * F.inv_h);
    y0 = max(0, y0);
    int y1 = (int)ceilf((y - (float)F.min_y + r) * F.inv_h);
    y1 = min(GR_ROWS - 1, y1);
    if (x0 < GR_COLS && x1 >= 0 && y0 < GR_ROWS && y1 >= 0) {
      const int o = FILL ? cand_start[i] : 0;
      uint4 q0 = make_uint4(0, 0, 0, 0), q1 = q0;
      if (FILL) {
        const uint4* QD = reinterpret_cast<const uint4*>(Q.desc + (int64_t)i * 32);
        q0 = QD[0], q1 = QD[1];
      }
      // a grid column's cells are consecutive CSR runs: walk the run of (ix, y0..y1) eight items at a time, index / key point /
      // descriptor loads of a batch issued together (same (ix, iy, insertion) order as the nested loops of GetFeaturesInArea)
      for (int ix = x0; ix <= x1; ++ix) {
        const int k_end = cell_start[ix * GR_ROWS + y1 + 1];
        for (int k0 = cell_start[ix * GR_ROWS + y0]; k0 < k_end; k0 += 8) {
          int idx[8], oct[8];
          float kx[8], ky[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) idx[u] = k0 + u < k_end ? cell_items[k0 + u] : -1;
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const uvo_keypoint* p = F.kp + (idx[u] >= 0 ? idx[u] : 0);
            kx[u] = p->x, ky[u] = p->y, oct[u] = p->octave;
          }
          bool take[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            bool t = idx[u] >= 0;
            if (bCheckLevels && !bSameLevel)
              t = t && !(oct[u] < minLevel || oct[u] > maxLevel);
            else if (bSameLevel)
              t = t && oct[u] == minLevel;
            take[u] = t && !(fabsf(kx[u] - x) > r || fabsf(ky[u] - y) > r);
          }
          if (FILL) {
            uint4 d0[8], d1[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
              const uint4* D = reinterpret_cast<const uint4*>(F.desc + (int64_t)(take[u] ? idx[u] : 0) * 32);
              d0[u] = D[0], d1[u] = D[1];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
              if (!take[u]) continue;
              const int d = __popc(q0.x ^ d0[u].x) + __popc(q0.y ^ d0[u].y) + __popc(q0.z ^ d0[u].z) + __popc(q0.w ^ d0[u].w) +
                            __popc(q1.x ^ d1[u].x) + __popc(q1.y ^ d1[u].y) + __popc(q1.z ^ d1[u].z) + __popc(q1.w ^ d1[u].w);
              cand[o + n] = (uint32_t)idx[u] | ((uint32_t)d << 16) | ((uint32_t)(oct[u] & 63) << 25) | 0x80000000u;
              ++n;
            }
          } else {
#pragma unroll
            for (int u = 0; u < 8; ++u) n += take[u] ? 1 : 0;
          }
        }
      }
    }
  }
  if (!FILL) cand_cnt[i] = n;
}

// Epipolar predicate of ORBmatcher::CheckDistEpipolarLine (src/ORBmatcher.cc:136-153); f12 row-major 3x3, fp32 in the
// reference's evaluation order, final comparison in double (3.84 is a double literal).
struct Epipolar {
  float f[9];
  const float* q_x;
  const float* q_y;
  const float* t_x;
  const float* t_y;
  const float* sigma2;  // per octave of the target key frame
  int enabled;
};

// one thread per candidate entry; query of an entry found by binary search in cand_start
__global__ __launch_bounds__(256) void k_group_dist(int nq, const int32_t* __restrict__ cand_start, const int32_t* __restrict__ cand_idx,
                                                    const uint8_t* __restrict__ qdesc, const uint8_t* __restrict__ tdesc,
                                                    const int32_t* __restrict__ tlevel, Epipolar E, uint32_t* __restrict__ cand) {
  const int total = cand_start[nq];
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  int lo = 0, hi = nq - 1;  // last query with cand_start[q] <= e
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (cand_start[mid] <= e)
      lo = mid;
    else
      hi = mid - 1;
  }
  const int q = lo, t = cand_idx[e];
  const int d = ham256(qdesc + (int64_t)q * 32, tdesc + (int64_t)t * 32);
  const int oct = tlevel ? tlevel[t] : 0;
  uint32_t ok = 1;
  if (E.enabled) {
    const float x1 = E.q_x[q], y1 = E.q_y[q], x2 = E.t_x[t], y2 = E.t_y[t];
    const float a = x1 * E.f[0] + y1 * E.f[3] + E.f[6];
    const float b = x1 * E.f[1] + y1 * E.f[4] + E.f[7];
    const float c = x1 * E.f[2] + y1 * E.f[5] + E.f[8];
    const float num = a * x2 + b * y2 + c;
    const float den = a * a + b * b;
    if (den == 0) {
      ok = 0;
    } else {
      const float dsqr = num * num / den;
      ok = (double)dsqr < 3.84 * (double)E.sigma2[oct] ? 1u : 0u;
    }
  }
  cand[e] = (uint32_t)t | ((uint32_t)d << 16) | ((uint32_t)(oct & 63) << 25) | (ok << 31);
}

struct MatchRule {
  int rule, max_dist;
  float nn_ratio;
  int exclusive;
};

// choice of query i given the current ownership; returns target index or -1, best distance through *dist
__device__ __forceinline__ int rule_choice(const MatchRule& R, int i, const int32_t* cand_start, const uint32_t* cand, const int32_t* owner,
                                           int* dist) {
  const int b = cand_start[i], e = cand_start[i + 1];
  *dist = -1;
  if (R.rule == UVO_RULE_TRIANGULATION) {
    // :893-935 -- free candidates with d <= TH_LOW, sorted by (d, idx2); walk while d <= round(2*best); first that passes
    // the epipolar test
    int best = 0x7fffffff;
    for (int c = b; c < e; ++c) {
      const uint32_t v = cand[c];
      if (R.exclusive && owner[v & 0xffffu] < i) continue;
      const int d = (int)((v >> 16) & 0x1ffu);
      if (d > R.max_dist) continue;
      best = d < best ? d : best;
    }
    if (best == 0x7fffffff) return -1;
    const int dist_th = 2 * best;
    uint32_t pick = 0xffffffffu;  // (d << 16 | idx): the sort order of vector<pair<int,size_t>>
    for (int c = b; c < e; ++c) {
      const uint32_t v = cand[c];
      if (!(v >> 31)) continue;
      if (R.exclusive && owner[v & 0xffffu] < i) continue;
      const int d = (int)((v >> 16) & 0x1ffu);
      if (d > R.max_dist || d > dist_th) continue;
      const uint32_t key = ((uint32_t)d << 16) | (v & 0xffffu);
      pick = key < pick ? key : pick;
    }
    if (pick == 0xffffffffu) return -1;
    *dist = (int)(pick >> 16);
    return (int)(pick & 0xffffu);
  }
  const int none = R.rule == UVO_RULE_BEST_RATIO_SAME_LEVEL ? 256 : 0x7fffffff;  // :77-81 vs INT_MAX elsewhere
  int bestDist = none, bestLevel = -1, bestDist2 = none, bestLevel2 = -1, bestIdx = -1;
  // eight candidates at a time: their words, then their owners, as independent loads in front of the ordered walk
  for (int c0 = b; c0 < e; c0 += 8) {
    uint32_t vv[8];
    int ow[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) vv[k] = c0 + k < e ? cand[c0 + k] : 0u;
#pragma unroll
    for (int k = 0; k < 8; ++k) ow[k] = (R.exclusive && c0 + k < e) ? owner[vv[k] & 0xffffu] : 0x7fffffff;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (c0 + k >= e) break;
      const uint32_t v = vv[k];
      const int idx = (int)(v & 0xffffu);
      if (R.exclusive && ow[k] < i) continue;  // taken before this query's turn (or blocked from the start)
      const int d = (int)((v >> 16) & 0x1ffu), oct = (int)((v >> 25) & 63u);
      if (d < bestDist) {
        bestDist2 = bestDist;
        bestDist = d;
        bestLevel2 = bestLevel;
        bestLevel = oct;
        bestIdx = idx;
      } else if (d < bestDist2) {
        bestLevel2 = oct;
        bestDist2 = d;
      }
    }
  }
  bool ok = false;
  switch (R.rule) {
    case UVO_RULE_BEST_RATIO_SAME_LEVEL:  // :114-123
      ok = bestDist <= R.max_dist && !(bestLevel == bestLevel2 && (float)bestDist > R.nn_ratio * (float)bestDist2);
      break;
    case UVO_RULE_BEST_ONLY:  // :1701, :1101
      ok = bestDist <= R.max_dist;
      break;
    case UVO_RULE_BEST_RATIO_LE:  // :216-218
      ok = bestDist <= R.max_dist && (float)bestDist < R.nn_ratio * (float)bestDist2;
      break;
    case UVO_RULE_BEST_RATIO_LT:  // :786-788
      ok = bestDist < R.max_dist && (float)bestDist < R.nn_ratio * (float)bestDist2;
      break;
    case UVO_RULE_BEST_RATIO_LEQ:  // WindowSearch :475, SearchByProjection(F1, F2, windowSize) :583 (bestDist2 = INT_MAX when there is no second)
      ok = (float)bestDist <= (float)bestDist2 * R.nn_ratio && bestDist <= R.max_dist;
      break;
    default:
      break;
  }
  if (!ok || bestIdx < 0) return -1;
  *dist = bestDist;
  return bestIdx;
}

// single workgroup.  blocked[t] != 0: target unavailable from the start.  match[i] = target or -1, mdist[i] = its distance.
__global__ __launch_bounds__(1024) void k_match_resolve(int nq, int nt, const int32_t* __restrict__ cand_start, const uint32_t* __restrict__ cand,
                                                        const uint8_t* __restrict__ blocked, MatchRule R, int32_t* owner,
                                                        int32_t* owner_next, int32_t* __restrict__ match,
                                                        int32_t* __restrict__ mdist, int32_t* __restrict__ n_matches) {
  __shared__ int s_changed, s_count;
  // ownership tables in LDS for up to 4096 targets (the atomics and the dependent reads of the walk stay on chip)
  __shared__ int32_t s_owner[4096], s_owner_next[4096];
  if (nt <= 4096) owner = s_owner, owner_next = s_owner_next;
  const int INF = 0x7fffffff;
  for (int k = threadIdx.x; k < nt; k += blockDim.x) owner[k] = (blocked && blocked[k]) ? -1 : INF;
  for (int i = threadIdx.x; i < nq; i += blockDim.x) match[i] = -2;
  __syncthreads();
  for (int iter = 0; iter <= nq; ++iter) {
    if (threadIdx.x == 0) s_changed = 0;
    for (int k = threadIdx.x; k < nt; k += blockDim.x) owner_next[k] = owner[k] < 0 ? -1 : INF;
    __syncthreads();
    for (int i = threadIdx.x; i < nq; i += blockDim.x) {
      int d;
      const int ch = rule_choice(R, i, cand_start, cand, owner, &d);
      if (ch != match[i]) {
        match[i] = ch;
        s_changed = 1;
      }
      mdist[i] = d;
      if (ch >= 0 && R.exclusive) atomicMin(&owner_next[ch], i);
    }
    __syncthreads();
    for (int k = threadIdx.x; k < nt; k += blockDim.x) owner[k] = owner_next[k];
    const int changed = s_changed;
    __syncthreads();
    if (!changed || !R.exclusive) break;
  }
  if (threadIdx.x == 0) s_count = 0;
  __syncthreads();
  int c = 0;
  for (int i = threadIdx.x; i < nq; i += blockDim.x) c += match[i] >= 0;
  if (c) atomicAdd(&s_count, c);
  __syncthreads();
  if (threadIdx.x == 0) *n_matches = s_count;
}

// SearchForInitialization (src/ORBmatcher.cc:598-713): no exclusivity -- a target keeps the distance it was last matched at
// (vMatchedDistance) and a later query takes it over when its own distance is strictly smaller; candidates whose current matched
// distance is <= the query's distance are skipped before best / second are formed (:635-636).  Solved as a fixed point like the
// exclusive rules: what query i sees at target t is min{ dist(j, t) : j < i accepted t } -- every accept lowers the target's distance,
// so that is the state the sequential loop is in when it reaches i.  The accepts of one sweep hang off their target as a linked list
// (head[t], nxt[i]); query 0 is final after one sweep, query i once all j < i are.  Outputs: match[i] = the target query i accepted
// (displaced or not: the rotation histogram counts every accept, :662-670), owner[t] = the last query that accepted t (vnMatches21).
__global__ __launch_bounds__(1024) void k_match_resolve_steal(int nq, int nt, const int32_t* __restrict__ cand_start, const uint32_t* __restrict__ cand,
                                                              int max_dist, float nn_ratio, int32_t* head, int32_t* nxt, int32_t* tmp_choice,
                                                              int32_t* tmp_dist, int32_t* match, int32_t* mdist) {
  __shared__ int s_changed;
  for (int t = threadIdx.x; t < nt; t += blockDim.x) head[t] = -1;
  for (int i = threadIdx.x; i < nq; i += blockDim.x) match[i] = -1, mdist[i] = -1, nxt[i] = -1;
  __syncthreads();
  for (int iter = 0; iter <= nq; ++iter) {
    if (threadIdx.x == 0) s_changed = 0;
    for (int i = threadIdx.x; i < nq; i += blockDim.x) {
      int bestDist = 0x7fffffff, bestDist2 = 0x7fffffff, bestIdx = -1;
      for (int c = cand_start[i]; c < cand_start[i + 1]; ++c) {
        const uint32_t v = cand[c];
        const int t = (int)(v & 0xffffu), d = (int)((v >> 16) & 0x1ffu);
        int md = 0x7fffffff;  // vMatchedDistance[t] as query i finds it
        for (int j = head[t]; j >= 0; j = nxt[j])
          if (j < i) md = min(md, mdist[j]);
        if (md <= d) continue;  // :635
        if (d < bestDist) {
          bestDist2 = bestDist, bestDist = d, bestIdx = t;
        } else if (d < bestDist2) {
          bestDist2 = d;
        }
      }
      const bool ok = bestIdx >= 0 && bestDist <= max_dist && (float)bestDist < (float)bestDist2 * nn_ratio;  // :649-651
      tmp_choice[i] = ok ? bestIdx : -1;
      tmp_dist[i] = ok ? bestDist : -1;
    }
    __syncthreads();
    for (int t = threadIdx.x; t < nt; t += blockDim.x) head[t] = -1;
    __syncthreads();
    for (int i = threadIdx.x; i < nq; i += blockDim.x) {
      const int ch = tmp_choice[i];
      if (ch != match[i] || tmp_dist[i] != mdist[i]) s_changed = 1;
      match[i] = ch, mdist[i] = tmp_dist[i];
      nxt[i] = ch >= 0 ? atomicExch(&head[ch], i) : -1;
    }
    __syncthreads();
    const int changed = s_changed;
    __syncthreads();
    if (!changed) break;
  }
  // vnMatches21[t] = the last query that accepted t
  for (int t = threadIdx.x; t < nt; t += blockDim.x) {
    int last = -1;
    for (int j = head[t]; j >= 0; j = nxt[j]) last = max(last, j);
    head[t] = last;
  }
}
// after the optional rotation filter: a query keeps its match only while it still holds the target (:653-657), count the survivors
__global__ __launch_bounds__(1024) void k_steal_finalize(int nq, const int32_t* __restrict__ holder, int32_t* match, int32_t* mdist, int32_t* n_matches) {
  __shared__ int s_count;
  if (threadIdx.x == 0) s_count = 0;
  __syncthreads();
  int c = 0;
  for (int i = threadIdx.x; i < nq; i += blockDim.x) {
    const int t = match[i];
    if (t >= 0 && holder[t] != i) match[i] = -1, mdist[i] = -1;
    c += match[i] >= 0;
  }
  if (c) atomicAdd(&s_count, c);
  __syncthreads();
  if (threadIdx.x == 0) *n_matches = s_count;
}

// Rotation consistency: rot = angle(query) - angle(target), +360 if negative, bin = round(rot / 30) (bin 30 -> 0); keep the
// three most populated bins (ComputeThreeMaxima), drop the matches of every other bin.  Single workgroup.
__global__ __launch_bounds__(1024) void k_rot_filter(int nq, const float* __restrict__ qangle, const float* __restrict__ tangle,
                                                     int32_t* __restrict__ match, int32_t* __restrict__ mdist, int32_t* __restrict__ n_matches) {
  __shared__ int s_hist[HISTO_LENGTH];
  __shared__ int s_keep[3];
  __shared__ int s_removed;
  if (threadIdx.x < HISTO_LENGTH) s_hist[threadIdx.x] = 0;
  if (threadIdx.x == 0) s_removed = 0;
  __syncthreads();
  const float factor = 1.0f / HISTO_LENGTH;
  auto bin_of = [&](int i) -> int {
    float rot = qangle[i] - tangle[match[i]];
    if (rot < 0.0) rot += 360.0f;
    // the reference asserts bin >= 0 && bin < HISTO_LENGTH (e.g. src/ORBmatcher.cc:237); angles outside [0, 360) or NaN would
    // break that: such a match lands in no bin (-2 is never a kept bin) and is dropped below
    if (!(rot >= 0.0f && rot < 360.0f * 1.05f)) return -2;
    int bin = (int)roundf(rot * factor);
    if (bin == HISTO_LENGTH) bin = 0;
    return bin >= 0 && bin < HISTO_LENGTH ? bin : -2;
  };
  for (int i = threadIdx.x; i < nq; i += blockDim.x)
    if (match[i] >= 0) {
      const int b = bin_of(i);
      if (b >= 0) atomicAdd(&s_hist[b], 1);
    }
  __syncthreads();
  if (threadIdx.x == 0) {
    int max1 = 0, max2 = 0, max3 = 0, ind1 = -1, ind2 = -1, ind3 = -1;
    for (int i = 0; i < HISTO_LENGTH; i++) {
      const int s = s_hist[i];
      if (s > max1) {
        max3 = max2;
        max2 = max1;
        max1 = s;
        ind3 = ind2;
        ind2 = ind1;
        ind1 = i;
      } else if (s > max2) {
        max3 = max2;
        max2 = s;
        ind3 = ind2;
        ind2 = i;
      } else if (s > max3) {
        max3 = s;
        ind3 = i;
      }
    }
    if (max2 < 0.1f * (float)max1) {
      ind2 = -1;
      ind3 = -1;
    } else if (max3 < 0.1f * (float)max1) {
      ind3 = -1;
    }
    s_keep[0] = ind1, s_keep[1] = ind2, s_keep[2] = ind3;
  }
  __syncthreads();
  int removed = 0;
  for (int i = threadIdx.x; i < nq; i += blockDim.x) {
    if (match[i] < 0) continue;
    const int b = bin_of(i);
    if (b != s_keep[0] && b != s_keep[1] && b != s_keep[2]) {
      match[i] = -1;
      mdist[i] = -1;
      ++removed;
    }
  }
  if (removed) atomicAdd(&s_removed, removed);
  __syncthreads();
  if (threadIdx.x == 0) *n_matches -= s_removed;
}

__global__ __launch_bounds__(1024) void k_scan_counts(const int32_t* __restrict__ in, int32_t* __restrict__ out, int n) {
  // single-workgroup exclusive scan, out[n] = total
  __shared__ int s_part[1024];
  const int per = (n + 1023) / 1024;
  const int b = threadIdx.x * per, e = min(b + per, n);
  int s = 0;
  for (int i = b; i < e; ++i) s += in[i];
  s_part[threadIdx.x] = s;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    const int v = threadIdx.x >= off ? s_part[threadIdx.x - off] : 0;
    __syncthreads();
    s_part[threadIdx.x] += v;
    __syncthreads();
  }
  int run = threadIdx.x ? s_part[threadIdx.x - 1] : 0;
  for (int i = b; i < e; ++i) {
    const int v = in[i];
    out[i] = run;
    run += v;
  }
  if (threadIdx.x == 1023) out[n] = s_part[1023];
}

// ---- projection prologues (uvo_project_points) ------------------------------------------------------------------------
struct ProjCam {
  float r[9], t[3], ow[3];
  float fx, fy, cx, cy, min_x, max_x, min_y, max_y;
};
// cv::Mat 3x3 * 3x1 + 3x1 in fp32: row sum in float, `+ t` through double (cv::gemm small-matrix path, alpha = beta = 1.0)
__device__ __forceinline__ float row_affine(const float* r, const float* p, float t) {
  const float t0 = r[0] * p[0] + r[1] * p[1] + r[2] * p[2];
  return (float)((double)t0 * 1.0 + (double)t * 1.0);
}
__device__ __forceinline__ int lower_bound_level(const float* sf, int n, float ratio) {  // std::lower_bound index
  int i = 0;
  while (i < n && sf[i] < ratio) ++i;
  return i;
}

__global__ __launch_bounds__(256) void k_project(int mode, ProjCam C, int n, const float* __restrict__ xyz, const float* __restrict__ normal,
                                                 const float* __restrict__ min_d, const float* __restrict__ max_d,
                                                 const float* __restrict__ max_raw, const uint8_t* __restrict__ usable, const float* __restrict__ sf, int nlevels, float log_sf,
                                                 float cos_limit, uint8_t* __restrict__ valid, float* __restrict__ out_u,
                                                 float* __restrict__ out_v, int32_t* __restrict__ out_level, float* __restrict__ out_cos) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint8_t ok = 0;
  float u = 0.f, v = 0.f, vc = 0.f;
  int lvl = 0;
  do {
    if (usable && !usable[i]) break;
    const float P[3] = {xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]};
    const float X = row_affine(C.r, P, C.t[0]), Y = row_affine(C.r + 3, P, C.t[1]), Z = row_affine(C.r + 6, P, C.t[2]);
    float ow[3] = {C.ow[0], C.ow[1], C.ow[2]};
    if (mode == UVO_PROJECT_KF_RELOC) {
      // Ow = -Rcw.t()*tcw (src/ORBmatcher.cc:1628): general gemm path, double accumulation, alpha = -1
      for (int c = 0; c < 3; ++c) {
        double s = 0.0;
        for (int k = 0; k < 3; ++k) s += (double)C.r[3 * k + c] * (double)C.t[k];
        ow[c] = (float)(s * -1.0);
      }
    }
    if (mode == UVO_PROJECT_PIXEL || mode == UVO_PROJECT_PIXEL_BOUNDED) {
      // SearchByProjection(F1, F2, windowSize) :541-550 (no test at all) / SearchByProjection(CurrentFrame, LastFrame, th) :1530-1545
      const float invz = (float)(1.0 / (double)Z);
      u = C.fx * X * invz + C.cx, v = C.fy * Y * invz + C.cy;
      if (mode == UVO_PROJECT_PIXEL_BOUNDED && (u < C.min_x || u > C.max_x || v < C.min_y || v > C.max_y)) break;
      ok = 1;
      break;
    }
    if (mode == UVO_PROJECT_FUSE) {
      if (Z < 0.0f) break;                 // :1040
      const float invz = 1 / Z;            // :1043
      const float x = X * invz, y = Y * invz;
      u = C.fx * x + C.cx, v = C.fy * y + C.cy;
      if (!(u >= C.min_x && u < C.max_x && v >= C.min_y && v < C.max_y)) break;  // KeyFrame::IsInImage
    } else {
      if (mode == UVO_PROJECT_FRUSTUM && Z < 0.0) break;  // src/FrameKTL.cc:313
      const float invz = (float)(1.0 / (double)Z);        // :317 / src/ORBmatcher.cc:1652
      u = C.fx * X * invz + C.cx, v = C.fy * Y * invz + C.cy;
      if (u < C.min_x || u > C.max_x) break;
      if (v < C.min_y || v > C.max_y) break;
    }
    const float maxDistance = max_d[i], minDistance = min_d[i];  // GetMax/MinDistanceInvariance(), src/MapPoint.cc:344-354
    const float PO[3] = {P[0] - ow[0], P[1] - ow[1], P[2] - ow[2]};
    double s2 = 0.0;
    for (int k = 0; k < 3; ++k) s2 += (double)PO[k] * (double)PO[k];  // cv::norm: double accumulator
    const float dist = (float)sqrt(s2);
    if (mode != UVO_PROJECT_KF_RELOC && (dist < minDistance || dist > maxDistance)) break;
    if (mode != UVO_PROJECT_KF_RELOC) {
      double dot = 0.0;
      for (int k = 0; k < 3; ++k) dot += (double)PO[k] * (double)normal[3 * i + k];  // cv::Mat::dot: double accumulator
      if (mode == UVO_PROJECT_FUSE) {
        if (dot < 0.5 * (double)dist) break;  // :1066
      } else {
        vc = (float)(dot / (double)dist);  // src/FrameKTL.cc:338
        if (vc < cos_limit) break;
      }
    }
    if (mode == UVO_PROJECT_FRUSTUM) {
      const float ratio = max_raw[i] / dist;  // MapPoint::PredictScale src/MapPoint.cc:378 (the raw mfMaxDistance)
      int nScale = (int)ceilf(uvo_logf(ratio) / log_sf);
      if (nScale < 0)
        nScale = 0;
      else if (nScale >= nlevels)
        nScale = nlevels - 1;
      lvl = nScale;
    } else {
      const float ratio = dist / minDistance;  // src/ORBmatcher.cc:1664 / :1070
      lvl = min(lower_bound_level(sf, nlevels, ratio), nlevels - 1);
    }
    ok = 1;
  } while (false);
  valid[i] = ok;
  out_u[i] = ok ? u : 0.f;
  out_v[i] = ok ? v : 0.f;
  out_level[i] = ok ? lvl : 0;
  if (out_cos) out_cos[i] = ok ? vc : 0.f;
}

void launch_project(hipStream_t s, int mode, const uvo_camera_pose& cam, int n, const float* d_xyz, const float* d_normal, const float* d_min,
                    const float* d_max, const float* d_max_raw, const uint8_t* d_usable, const float* d_sf, int nlevels, float log_sf, float cos_limit, uint8_t* d_valid,
                    float* d_u, float* d_v, int32_t* d_level, float* d_cos) {
  ProjCam C;
  for (int i = 0; i < 9; ++i) C.r[i] = cam.rcw[i];
  for (int i = 0; i < 3; ++i) C.t[i] = cam.tcw[i], C.ow[i] = cam.ow[i];
  C.fx = cam.fx, C.fy = cam.fy, C.cx = cam.cx, C.cy = cam.cy;
  C.min_x = cam.min_x, C.max_x = cam.max_x, C.min_y = cam.min_y, C.max_y = cam.max_y;
  hipLaunchKernelGGL(k_project, dim3((n + 255) / 256), dim3(256), 0, s, mode, C, n, d_xyz, d_normal, d_min, d_max, d_max_raw, d_usable, d_sf, nlevels, log_sf,
                     cos_limit, d_valid, d_u, d_v, d_level, d_cos);
}

// SearchBySim3 per-point prologue (src/ORBmatcher.cc:1323-1359 / :1403-1441): world -> own camera -> other camera -> pixel
struct Sim3Chain {
  float r_own[9], t_own[3], s_r[9], t[3];
  float fx, fy, cx, cy, min_x, max_x, min_y, max_y;
};
__global__ __launch_bounds__(256) void k_project_sim3(Sim3Chain C, int n, const float* __restrict__ xyz, const float* __restrict__ min_d,
                                                      const float* __restrict__ max_d, const uint8_t* __restrict__ usable,
                                                      const float* __restrict__ sf, int nlevels, uint8_t* __restrict__ valid,
                                                      float* __restrict__ out_u, float* __restrict__ out_v, int32_t* __restrict__ out_level) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint8_t ok = 0;
  float u = 0.f, v = 0.f;
  int lvl = 0;
  do {
    if (usable && !usable[i]) break;
    const float P[3] = {xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]};
    const float Q[3] = {row_affine(C.r_own, P, C.t_own[0]), row_affine(C.r_own + 3, P, C.t_own[1]), row_affine(C.r_own + 6, P, C.t_own[2])};
    const float X = row_affine(C.s_r, Q, C.t[0]), Y = row_affine(C.s_r + 3, Q, C.t[1]), Z = row_affine(C.s_r + 6, Q, C.t[2]);
    if (Z < 0.0f) break;                           // :1328
    const float invz = (float)(1.0 / (double)Z);   // :1331
    const float x = X * invz, y = Y * invz;
    u = C.fx * x + C.cx, v = C.fy * y + C.cy;
    if (!(u >= C.min_x && u < C.max_x && v >= C.min_y && v < C.max_y)) break;  // KeyFrame::IsInImage
    const double s2 = (double)X * (double)X + (double)Y * (double)Y + (double)Z * (double)Z;  // cv::norm(p3Dc2): double accumulator
    const float dist3D = (float)sqrt(s2);
    if (dist3D < min_d[i] || dist3D > max_d[i]) break;  // :1346
    const float ratio = dist3D / min_d[i];
    lvl = min(lower_bound_level(sf, nlevels, ratio), nlevels - 1);  // :1352-1353
    ok = 1;
  } while (false);
  valid[i] = ok;
  out_u[i] = ok ? u : 0.f;
  out_v[i] = ok ? v : 0.f;
  out_level[i] = ok ? lvl : 0;
}
void launch_project_sim3(hipStream_t s, const float* r_own, const float* t_own, const float* s_r, const float* t, const uvo_camera_pose& cam, int n,
                         const float* d_xyz, const float* d_min, const float* d_max, const uint8_t* d_usable, const float* d_sf, int nlevels,
                         uint8_t* d_valid, float* d_u, float* d_v, int32_t* d_level) {
  Sim3Chain C;
  for (int i = 0; i < 9; ++i) C.r_own[i] = r_own[i], C.s_r[i] = s_r[i];
  for (int i = 0; i < 3; ++i) C.t_own[i] = t_own[i], C.t[i] = t[i];
  C.fx = cam.fx, C.fy = cam.fy, C.cx = cam.cx, C.cy = cam.cy;
  C.min_x = cam.min_x, C.max_x = cam.max_x, C.min_y = cam.min_y, C.max_y = cam.max_y;
  hipLaunchKernelGGL(k_project_sim3, dim3((n + 255) / 256), dim3(256), 0, s, C, n, d_xyz, d_min, d_max, d_usable, d_sf, nlevels, d_valid, d_u, d_v,
                     d_level);
}

// haloc::Hash::getHash (src/hash.cpp:57-85): thread = one (projection, descriptor column) output, sequential fp32 accumulation
__global__ __launch_bounds__(64) void k_haloc(const float* __restrict__ proj, int num_proj, int proj_stride, const uint8_t* __restrict__ desc, int n,
                                              float* __restrict__ hash) {
  const int k = blockIdx.x * 64 + threadIdx.x;
  if (k >= num_proj * 32) return;
  const int i = k >> 5, c = k & 31;
  const float* r = proj + (int64_t)i * proj_stride;
  float desc_sum = 0.0f;
  for (int m = 0; m < n; ++m) desc_sum += r[m] * (float)desc[(int64_t)m * 32 + c];
  hash[k] = desc_sum / (float)n;
}
void launch_haloc(hipStream_t s, const float* d_proj, int num_proj, int proj_stride, const uint8_t* d_desc, int n, float* d_hash) {
  hipLaunchKernelGGL(k_haloc, dim3((num_proj * 32 + 63) / 64), dim3(64), 0, s, d_proj, num_proj, proj_stride, d_desc, n, d_hash);
}

// grid build shared with search.hip
void launch_grid_build(hipStream_t s, const uvo_keypoint* d_kp, const uint8_t* d_desc, int n, int min_x, int min_y, int max_x, int max_y,
                       int32_t* d_cell_start, int32_t* d_cell_items, int32_t* d_cell_of_kp);

void launch_win_count(hipStream_t s, const uvo_keypoint* d_kp, const uint8_t* d_desc, int n, int min_x, int min_y, int max_x, int max_y, int nq,
                      const float* d_qx, const float* d_qy, const float* d_qr, const int32_t* d_qmin, const int32_t* d_qmax,
                      const uint8_t* d_qvalid, const uint8_t* d_qdesc, int32_t* d_cell_start, int32_t* d_cell_items, int32_t* d_cell_of_kp,
                      int32_t* d_cand_cnt, int32_t* d_cand_start) {
  launch_grid_build(s, d_kp, d_desc, n, min_x, min_y, max_x, max_y, d_cell_start, d_cell_items, d_cell_of_kp);
  WinFrame F{d_kp, d_desc, n, min_x, min_y, (float)GR_COLS / (float)(max_x - min_x), (float)GR_ROWS / (float)(max_y - min_y)};
  WinQuery Q{d_qx, d_qy, d_qr, d_qmin, d_qmax, d_qvalid, d_qdesc, nq};
  hipLaunchKernelGGL(k_win_cand<false>, dim3((nq + 255) / 256), dim3(256), 0, s, F, Q, d_cell_start, d_cell_items, d_cand_cnt,
                     (const int32_t*)nullptr, (uint32_t*)nullptr);
  hipLaunchKernelGGL(k_scan_counts, dim3(1), dim3(1024), 0, s, d_cand_cnt, d_cand_start, nq);
}

void launch_win_fill(hipStream_t s, const uvo_keypoint* d_kp, const uint8_t* d_desc, int n, int min_x, int min_y, int max_x, int max_y, int nq,
                     const float* d_qx, const float* d_qy, const float* d_qr, const int32_t* d_qmin, const int32_t* d_qmax,
                     const uint8_t* d_qvalid, const uint8_t* d_qdesc, const int32_t* d_cell_start, const int32_t* d_cell_items,
                     const int32_t* d_cand_start, uint32_t* d_cand) {
  WinFrame F{d_kp, d_desc, n, min_x, min_y, (float)GR_COLS / (float)(max_x - min_x), (float)GR_ROWS / (float)(max_y - min_y)};
  WinQuery Q{d_qx, d_qy, d_qr, d_qmin, d_qmax, d_qvalid, d_qdesc, nq};
  hipLaunchKernelGGL(k_win_cand<true>, dim3((nq + 255) / 256), dim3(256), 0, s, F, Q, d_cell_start, d_cell_items, (int32_t*)nullptr,
                     d_cand_start, d_cand);
}

void launch_group_dist(hipStream_t s, int nq, int total, const int32_t* d_cand_start, const int32_t* d_cand_idx, const uint8_t* d_qdesc,
                       const uint8_t* d_tdesc, const int32_t* d_tlevel, const float* f12, const float* d_qx, const float* d_qy,
                       const float* d_tx, const float* d_ty, const float* d_sigma2, uint32_t* d_cand) {
  Epipolar E;
  E.enabled = f12 ? 1 : 0;
  for (int i = 0; i < 9; ++i) E.f[i] = f12 ? f12[i] : 0.f;
  E.q_x = d_qx, E.q_y = d_qy, E.t_x = d_tx, E.t_y = d_ty, E.sigma2 = d_sigma2;
  if (total > 0)
    hipLaunchKernelGGL(k_group_dist, dim3((total + 255) / 256), dim3(256), 0, s, nq, d_cand_start, d_cand_idx, d_qdesc, d_tdesc, d_tlevel, E,
                       d_cand);
}

// ---- batched forms for the LocalMapping thread (src/LocalMapping.cc:1058-1080, :1228-1236) ----
// k_group_dist over the queries of SEVERAL (key frame 1, key frame 2) pairs at once: entry e belongs to query q (binary search), query q
// to pair q_pair[q]; candidate indices are global (pair_base[p] + index in key frame 2 of pair p); every pair has its own fundamental
// matrix (f12 + 9 p) and sigma table (sigma2 + sig_stride p).  The packed word holds the index inside the pair's key frame.
__global__ __launch_bounds__(256) void k_group_dist_pairs(int nq, const int32_t* __restrict__ cand_start, const int32_t* __restrict__ cand_idx,
                                                          const uint8_t* __restrict__ qdesc, const uint8_t* __restrict__ tdesc,
                                                          const int32_t* __restrict__ tlevel, const int32_t* __restrict__ q_pair,
                                                          const int32_t* __restrict__ pair_base, const float* __restrict__ f12,
                                                          const float* __restrict__ q_x, const float* __restrict__ q_y, const float* __restrict__ t_x,
                                                          const float* __restrict__ t_y, const float* __restrict__ sigma2, int sig_stride,
                                                          uint32_t* __restrict__ cand) {
  const int total = cand_start[nq];
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  int lo = 0, hi = nq - 1;  // last query with cand_start[q] <= e
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (cand_start[mid] <= e)
      lo = mid;
    else
      hi = mid - 1;
  }
  const int q = lo, t = cand_idx[e], p = q_pair[q];
  const int d = ham256(qdesc + (int64_t)q * 32, tdesc + (int64_t)t * 32);
  const int oct = tlevel[t];
  const float* F = f12 + 9 * p;
  // ORBmatcher::CheckDistEpipolarLine (src/ORBmatcher.cc:136-153), as in k_group_dist
  const float x1 = q_x[q], y1 = q_y[q], x2 = t_x[t], y2 = t_y[t];
  const float a = x1 * F[0] + y1 * F[3] + F[6];
  const float b = x1 * F[1] + y1 * F[4] + F[7];
  const float c = x1 * F[2] + y1 * F[5] + F[8];
  const float num = a * x2 + b * y2 + c;
  const float den = a * a + b * b;
  uint32_t ok = 0;
  if (den != 0) {
    const float dsqr = num * num / den;
    ok = (double)dsqr < 3.84 * (double)sigma2[(int64_t)p * sig_stride + oct] ? 1u : 0u;
  }
  cand[e] = (uint32_t)(t - pair_base[p]) | ((uint32_t)d << 16) | ((uint32_t)(oct & 63) << 25) | (ok << 31);
}

void launch_group_dist_pairs(hipStream_t s, int nq, int total, const int32_t* d_cand_start, const int32_t* d_cand_idx, const uint8_t* d_qdesc,
                             const uint8_t* d_tdesc, const int32_t* d_tlevel, const int32_t* d_q_pair, const int32_t* d_pair_base, const float* d_f12,
                             const float* d_qx, const float* d_qy, const float* d_tx, const float* d_ty, const float* d_sigma2, int sig_stride,
                             uint32_t* d_cand) {
  if (total > 0)
    hipLaunchKernelGGL(k_group_dist_pairs, dim3((total + 255) / 256), dim3(256), 0, s, nq, d_cand_start, d_cand_idx, d_qdesc, d_tdesc, d_tlevel, d_q_pair,
                       d_pair_base, d_f12, d_qx, d_qy, d_tx, d_ty, d_sigma2, sig_stride, d_cand);
}

// Search core of ORBmatcher::Fuse (src/ORBmatcher.cc:1077-1101) for one target key frame, straight to the result: per projected map point
// the key points KeyFrame::GetFeaturesInArea(u, v, th * scale[level]) returns, in its (ix, iy, insertion) order, on levels
// [level - 1, level], the smallest descriptor distance (strict <: the first of equals wins), accepted at <= TH_LOW.  No candidate lists
// and no ownership (Fuse has no exclusivity), so nothing needs a host round trip between the targets of a batch.
__global__ __launch_bounds__(256) void k_fuse_walk(WinFrame F, int nmp, const uint8_t* __restrict__ valid, const float* __restrict__ qu,
                                                   const float* __restrict__ qv, const int32_t* __restrict__ qlevel, const uint8_t* __restrict__ mp_desc,
                                                   const float* __restrict__ sf, float th, const int32_t* __restrict__ cell_start,
                                                   const int32_t* __restrict__ cell_items, int32_t* __restrict__ best_idx, int32_t* __restrict__ best_dist) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nmp) return;
  int bestDist = 0x7fffffff, bestIdx = -1;
  if (valid[i]) {
    const float x = qu[i], y = qv[i];
    const int level = qlevel[i];
    const float r = th * sf[level];
    const int minLevel = level - 1, maxLevel = level;
    int x0 = (int)floorf((x - (float)F.min_x - r) * F.inv_w);
    x0 = max(0, x0);
    int x1 = (int)ceilf((x - (float)F.min_x + r) * F.inv_w);
    x1 = min(GR_COLS - 1, x1);
    int y0 = (int)floorf((y - (float)F.min_y - r) * F.inv_h);
    y0 = max(0, y0);
    int y1 = (int)ceilf((y - (float)F.min_y + r) * F.inv_h);
    y1 = min(GR_ROWS - 1, y1);
    if (x0 < GR_COLS && x1 >= 0 && y0 < GR_ROWS && y1 >= 0) {
      const uint4* QD = reinterpret_cast<const uint4*>(mp_desc + (int64_t)i * 32);
      const uint4 q0 = QD[0], q1 = QD[1];
      for (int ix = x0; ix <= x1; ++ix) {
        const int k_end = cell_start[ix * GR_ROWS + y1 + 1];
        for (int k = cell_start[ix * GR_ROWS + y0]; k < k_end; ++k) {
          const int idx = cell_items[k];
          const uvo_keypoint* kp = F.kp + idx;
          const int oct = kp->octave;
          if (oct < minLevel || oct > maxLevel) continue;
          if (fabsf(kp->x - x) > r || fabsf(kp->y - y) > r) continue;
          const uint4* D = reinterpret_cast<const uint4*>(F.desc + (int64_t)idx * 32);
          const uint4 d0 = D[0], d1 = D[1];
          const int d = __popc(q0.x ^ d0.x) + __popc(q0.y ^ d0.y) + __popc(q0.z ^ d0.z) + __popc(q0.w ^ d0.w) + __popc(q1.x ^ d1.x) +
                        __popc(q1.y ^ d1.y) + __popc(q1.z ^ d1.z) + __popc(q1.w ^ d1.w);
          if (d < bestDist) bestDist = d, bestIdx = idx;
        }
      }
    }
  }
  const bool ok = bestIdx >= 0 && bestDist <= 50;  // TH_LOW, src/ORBmatcher.cc:41,:1101
  best_idx[i] = ok ? bestIdx : -1;
  best_dist[i] = ok ? bestDist : -1;
}

void launch_fuse_walk(hipStream_t s, const uvo_keypoint* d_kp, const uint8_t* d_desc, int n, int min_x, int min_y, int max_x, int max_y, int nmp,
                      const uint8_t* d_valid, const float* d_u, const float* d_v, const int32_t* d_level, const uint8_t* d_mp_desc, const float* d_sf, float th,
                      int32_t* d_cell_start, int32_t* d_cell_items, int32_t* d_cell_of_kp, int32_t* d_best_idx, int32_t* d_best_dist) {
  launch_grid_build(s, d_kp, d_desc, n, min_x, min_y, max_x, max_y, d_cell_start, d_cell_items, d_cell_of_kp);
  WinFrame F{d_kp, d_desc, n, min_x, min_y, (float)GR_COLS / (float)(max_x - min_x), (float)GR_ROWS / (float)(max_y - min_y)};
  hipLaunchKernelGGL(k_fuse_walk, dim3((nmp + 255) / 256), dim3(256), 0, s, F, nmp, d_valid, d_u, d_v, d_level, d_mp_desc, d_sf, th, d_cell_start, d_cell_items,
                     d_best_idx, d_best_dist);
}

void launch_match_resolve(hipStream_t s, int nq, int nt, const int32_t* d_cand_start, const uint32_t* d_cand, const uint8_t* d_blocked, int rule,
                          int max_dist, float nn_ratio, int exclusive, int32_t* d_owner, int32_t* d_owner_next, int32_t* d_match,
                          int32_t* d_mdist, int32_t* d_n_matches) {
  MatchRule R{rule, max_dist, nn_ratio, exclusive};
  hipLaunchKernelGGL(k_match_resolve, dim3(1), dim3(1024), 0, s, nq, nt, d_cand_start, d_cand, d_blocked, R, d_owner, d_owner_next, d_match,
                     d_mdist, d_n_matches);
}

void launch_match_resolve_steal(hipStream_t s, int nq, int nt, const int32_t* d_cand_start, const uint32_t* d_cand, int max_dist, float nn_ratio,
                                int32_t* d_head, int32_t* d_nxt, int32_t* d_tmp, int32_t* d_match, int32_t* d_mdist) {
  hipLaunchKernelGGL(k_match_resolve_steal, dim3(1), dim3(1024), 0, s, nq, nt, d_cand_start, d_cand, max_dist, nn_ratio, d_head, d_nxt, d_tmp, d_tmp + nq,
                     d_match, d_mdist);
}
void launch_steal_finalize(hipStream_t s, int nq, const int32_t* d_holder, int32_t* d_match, int32_t* d_mdist, int32_t* d_n_matches) {
  hipLaunchKernelGGL(k_steal_finalize, dim3(1), dim3(1024), 0, s, nq, d_holder, d_match, d_mdist, d_n_matches);
}

void launch_rot_filter(hipStream_t s, int nq, const float* d_qangle, const float* d_tangle, int32_t* d_match, int32_t* d_mdist,
                       int32_t* d_n_matches) {
  hipLaunchKernelGGL(k_rot_filter, dim3(1), dim3(1024), 0, s, nq, d_qangle, d_tangle, d_match, d_mdist, d_n_matches);
}

}  // namespace uvo
