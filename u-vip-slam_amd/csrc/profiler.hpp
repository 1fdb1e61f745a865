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
  std::vector<Rec> recs;

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
    Scope(Profiler* p_, const char* name, hipStream_t s_) : p(p_), s(s_) {
      r.name = name;
      r.a = r.b = nullptr;
      if (p->on) {
        (void)hipEventCreate(&r.a);
        (void)hipEventCreate(&r.b);
        (void)hipEventRecord(r.a, s);
      }
    }
    ~Scope() {
      if (p->on) {
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
