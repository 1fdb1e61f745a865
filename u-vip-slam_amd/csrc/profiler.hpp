// Per-launch device timing with HIP events on the stream the kernels are launched on.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

namespace uvo {

struct Profiler {
  struct Rec {
    const char* name;
    hipEvent_t a, b;
    uint64_t seq;  // enqueue order across every Profiler of the process (a handle's pipeline lanes each own one; the report folds them)
  };
  static uint64_t next_seq() {
    static std::atomic<uint64_t> c{0};
    return c.fetch_add(1, std::memory_order_relaxed);
  }
  bool on = false;
  std::string only;  // when not empty: time launches of this kernel only (two event records per launch are not free)
  std::vector<Rec> recs;
  bool wants(const char* name) const { return on && (only.empty() || only == name); }

  // events are kept and used again: creating a pair per launch keeps the host busy long enough to starve a stream of short kernels, and the
  // idle time in front of a launch then counts as its duration
  std::vector<hipEvent_t> pool;
  hipEvent_t take() {
    if (!pool.empty()) {
      hipEvent_t e = pool.back();
      pool.pop_back();
      return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
  }
  void clear() {
    for (auto& r : recs) pool.push_back(r.a), pool.push_back(r.b);
    recs.clear();
    while (pool.size() > 4096) (void)hipEventDestroy(pool.back()), pool.pop_back();
  }
  ~Profiler() {
    clear();
    for (hipEvent_t e : pool) (void)hipEventDestroy(e);
  }
  struct Scope {
    Profiler* p;
    hipStream_t s;
    Rec r;
    bool active;
    Scope(Profiler* p_, const char* name, hipStream_t s_) : p(p_), s(s_), active(p_->wants(name)) {
      r.name = name;
      r.a = r.b = nullptr;
      r.seq = 0;
      if (active) {
        r.seq = next_seq();
        r.a = p->take();
        r.b = p->take();
        (void)hipEventRecord(r.a, s);
      }
    }
    ~Scope() {
      if (active) {
        (void)hipEventRecord(r.b, s);
        p->recs.push_back(r);
      }
    }
  };
  // caller has synchronised the stream.  Rows: one per kernel name (summed duration, launches), followed -- when `spread` is set and the
  // name has at least two launches -- by the pseudo-rows "name:min" / "name:p50" / "name:max" (per-launch duration) and
  // "name:period_min" / "name:period_p50" / "name:period_max" (start-to-start time of consecutive launches in enqueue order, whatever
  // lane's stream they ran in) and "name:period2_*" (half the start-to-start time of launches TWO apart: with two pipeline lanes taking
  // the batches in turn that is one lane's period per step -- consecutive launches belong to different lanes, whose phase is free).
  int report(char* names, int names_cap, float* ms, int32_t* launches, int cap, bool spread = true) {
    std::stable_sort(recs.begin(), recs.end(), [](const Rec& x, const Rec& y) { return x.seq < y.seq; });
    std::vector<std::string> nm;
    std::vector<float> tt;
    std::vector<int> cc;
    std::vector<std::vector<float>> dur, per, per2;
    std::vector<const Rec*> last, last2;
    for (auto& r : recs) {
      float t = 0;
      if (hipEventElapsedTime(&t, r.a, r.b) != hipSuccess) continue;
      size_t k = 0;
      for (; k < nm.size(); ++k)
        if (nm[k] == r.name) break;
      if (k == nm.size())
        nm.push_back(r.name), tt.push_back(0.f), cc.push_back(0), dur.emplace_back(), per.emplace_back(), per2.emplace_back(), last.push_back(nullptr), last2.push_back(nullptr);
      tt[k] += t;
      cc[k] += 1;
      dur[k].push_back(t);
      float dp = 0;
      if (last[k] && hipEventElapsedTime(&dp, last[k]->a, r.a) == hipSuccess) per[k].push_back(dp);
      if (last2[k] && hipEventElapsedTime(&dp, last2[k]->a, r.a) == hipSuccess) per2[k].push_back(0.5f * dp);
      last2[k] = last[k];
      last[k] = &r;
    }
    std::string joined;
    int m = 0;
    auto row = [&](const std::string& name, float v, int n) {
      if (m >= cap) return;
      joined += name;
      joined += '\n';
      ms[m] = v;
      launches[m] = n;
      ++m;
    };
    for (size_t k = 0; k < nm.size(); ++k) row(nm[k], tt[k], cc[k]);
    if (spread) {
      auto stats = [&](const std::string& name, const char* what, std::vector<float>& v) {
        if (v.size() < 2) return;
        std::sort(v.begin(), v.end());
        row(name + ":" + what + "min", v.front(), (int)v.size());
        row(name + ":" + what + "p50", v[v.size() / 2], (int)v.size());
        row(name + ":" + what + "max", v.back(), (int)v.size());
      };
      for (size_t k = 0; k < nm.size(); ++k) stats(nm[k], "", dur[k]), stats(nm[k], "period_", per[k]), stats(nm[k], "period2_", per2[k]);
    }
    snprintf(names, names_cap, "%s", joined.c_str());
    clear();
    return m;
  }
};

}  // namespace uvo
