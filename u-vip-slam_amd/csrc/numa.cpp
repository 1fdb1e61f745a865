// Host-side placement for the N-GPU host-to-host leg (call site src/Tracking.cc:946): a rank's frames leave the host over
// the PCIe link of ITS GPU, so the thread that touches them first (and stages / gathers them) belongs on the CPUs -- and the memory on the
// NUMA node -- that link hangs off.  Linux sysfs tells: /sys/bus/pci/devices/<bdf>/local_cpulist and .../numa_node.
// Everything degrades to a no-op where the files do not exist (containers, other kernels): placement is speed, never correctness.
#include <sched.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "common.hpp"

namespace uvo {

// "0-15,64-79" -> cpu_set_t; false when the text holds no CPU
static bool parse_cpulist(const char* txt, cpu_set_t* set) {
  CPU_ZERO(set);
  int n = 0;
  const char* p = txt;
  while (*p) {
    while (*p == ' ' || *p == ',' || *p == '\n' || *p == '\t') ++p;
    if (!*p) break;
    char* e = nullptr;
    long a = strtol(p, &e, 10);
    if (e == p || a < 0) return false;
    long b = a;
    p = e;
    if (*p == '-') {
      b = strtol(p + 1, &e, 10);
      if (e == p + 1 || b < a) return false;
      p = e;
    }
    for (long c = a; c <= b && c < CPU_SETSIZE; ++c) CPU_SET((int)c, set), ++n;
  }
  return n > 0;
}

static bool read_small_file(const char* path, char* buf, size_t cap) {
  FILE* f = fopen(path, "r");
  if (!f) return false;
  const size_t n = fread(buf, 1, cap - 1, f);
  fclose(f);
  buf[n] = 0;
  return n > 0;
}

// The CPUs this process may run on: the affinity of the thread that loaded the library, taken once, before anyone narrowed it (falls back
// to Cpus_allowed_list of /proc/self/status, then to the caller's current mask).
static cpu_set_t capture_process_cpus() {
  cpu_set_t m;
  CPU_ZERO(&m);
  if (sched_getaffinity(0, sizeof(m), &m) == 0 && CPU_COUNT(&m) > 0) return m;
  char txt[8192];
  if (read_small_file("/proc/self/status", txt, sizeof(txt))) {
    char* k = strstr(txt, "Cpus_allowed_list:");
    if (k) {
      k += 18;
      if (char* nl = strchr(k, '\n')) *nl = 0;  // the list ends with its line
      if (parse_cpulist(k, &m)) return m;
    }
  }
  CPU_ZERO(&m);
  return m;
}
static const cpu_set_t g_process_cpus = capture_process_cpus();  // static initialisation = library load, on the loading thread
static cpu_set_t process_cpus() {
  if (CPU_COUNT(&g_process_cpus) > 0) return g_process_cpus;
  cpu_set_t m;
  CPU_ZERO(&m);
  (void)sched_getaffinity(0, sizeof(m), &m);
  return m;
}

// Binds the calling thread to the CPUs of `cpulist_path` that the process is allowed to run on (a side effect on the CALLING thread's
// affinity, inherited by threads it creates afterwards; UVO_NUMA_BIND=0 switches it off: INTEGRATION.md section 5).  1: bound, 0: nothing to do (no such file, an empty
// list, no CPU of the list available to this process), < 0: error.
int bind_thread_to_cpulist_file(const char* cpulist_path) {
  char txt[4096];
  const char* off = getenv("UVO_NUMA_BIND");  // UVO_NUMA_BIND=0: the host keeps its own placement (numactl, a job scheduler's cpusets)
  if (off && off[0] == '0') return 0;
  if (!cpulist_path || !read_small_file(cpulist_path, txt, sizeof(txt))) return 0;
  cpu_set_t want, both;
  if (!parse_cpulist(txt, &want)) return 0;
  // Against the mask the PROCESS was given (captured when the library was loaded), not the caller's current one: threads inherit their
  // creator's affinity, so a caller that has already bound itself next to GPU 0 would hand every thread it starts a mask that no longer
  // holds the CPUs of a GPU on the other socket -- the intersection would be empty and the shard would stay on the wrong node.
  const cpu_set_t have = process_cpus();
  CPU_AND(&both, &want, &have);
  if (CPU_COUNT(&both) == 0) return 0;  // (a cgroup that excludes the device's CPUs: stay where we are)
  if (sched_setaffinity(0, sizeof(both), &both) != 0) return fail(UVO_E_HIP, "sched_setaffinity failed");
  return 1;
}

// sysfs directory of a device ordinal: /sys/bus/pci/devices/0000:c1:00.0
static bool device_sysfs_dir(int device, std::string& dir) {
  char bdf[64] = "";
  if (hipDeviceGetPCIBusId(bdf, (int)sizeof(bdf), device) != hipSuccess || !bdf[0]) return false;
  for (char* c = bdf; *c; ++c) *c = (char)tolower(*c);
  dir = std::string("/sys/bus/pci/devices/") + bdf;
  return true;
}

}  // namespace uvo

extern "C" {

int uvo_host_bind_to_cpulist_file(const char* cpulist_path) { return uvo::bind_thread_to_cpulist_file(cpulist_path); }

int uvo_host_bind_near_device(int device, int32_t* numa_node) {
  if (numa_node) *numa_node = -1;
  std::string dir;
  if (!uvo::device_sysfs_dir(device, dir)) return 0;
  char txt[64];
  if (numa_node && uvo::read_small_file((dir + "/numa_node").c_str(), txt, sizeof(txt))) *numa_node = (int32_t)atoi(txt);
  return uvo::bind_thread_to_cpulist_file((dir + "/local_cpulist").c_str());
}

}  // extern "C"
