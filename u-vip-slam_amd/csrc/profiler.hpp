// Per-launch device timing with HIP events on the stream the kernels are launched on.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdio>
#include <string>
#include <vector>

namespace uvo {

struct Profiler {
  struct Rec {
    const char* name;
    hipEvent_t a, b;
  };
  bool on = false;
  std::string only;  // when not empty: time launches of this kernel only (two event records per launch are not free)
  std::vector<Rec> recs;
  bool wants(const char* name) const { return on && (only.empty() || only == name); }

  void clear() {
    for (auto& r : recs) {
      (void)hipEventDestroy(r.a);
      (void)hipEventDestroy(r.b);
    }
    recs.clear();
  }
  struct Scope {
    Profiler* p;
    hipStream_t s;
    Rec r;
    bool active;
    Scope(Profiler* p_, const char* name, hipStream_t s_) : p(p_), s(s_), active(p_->wants(name)) {
      r.name = name;
      r.a = r.b = nullptr;
      if (active) {
        (void)hipEventCreate(&r.a);
        (void)hipEventCreate(&r.b);
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
  // caller has synchronised the stream
  int report(char* names, int names_cap, float* ms, int32_t* launches, int cap) {
    std::vector<std::string> nm;
    std::vector<float> tt;
    std::vector<int> cc;
    for (auto& r : recs) {
      float t = 0;
      if (hipEventElapsedTime(&t, r.a, r.b) != hipSuccess) continue;
      size_t k = 0;
      for (; k < nm.size(); ++k)
        if (nm[k] == r.name) break;
      if (k == nm.size()) nm.push_back(r.name), tt.push_back(0.f), cc.push_back(0);
      tt[k] += t;
      cc[k] += 1;
    }
    std::string joined;
    int m = 0;
    for (size_t k = 0; k < nm.size() && (int)k < cap; ++k, ++m) {
      joined += nm[k];
      joined += '\n';
      ms[k] = tt[k];
      launches[k] = cc[k];
    }
    snprintf(names, names_cap, "%s", joined.c_str());
    clear();
    return m;
  }
};

}  // namespace uvo
