// host_numa.h -- which host CPUs sit next to a GPU (sysfs only), for the pool's worker threads and page-locked buffers
// (one rank per GPU on a two-socket node; the Python twin for ranks that bind before any GPU call is
// ark_vrf_amd/numa.py):   /sys/bus/pci/devices/<bdf>/numa_node  ->  /sys/devices/system/node/node<k>/cpulist.
// AVRF_SYSFS_ROOT replaces "/" (tests); AVRF_POOL_NUMA=0 switches the binding off.
#pragma once
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>

namespace avrf {

inline bool read_small_file(const std::string &path, std::string &out) {
  FILE *f = fopen(path.c_str(), "r");
  if (!f) return false;
  char buf[4096]; const size_t n = fread(buf, 1, sizeof buf - 1, f); fclose(f);
  buf[n] = 0; out = buf; return true;
}
// "0-3,8,10-11" -> {0,1,2,3,8,10,11}
inline std::vector<int> parse_cpulist(const std::string &s) {
  std::vector<int> out;
  const char *p = s.c_str();
  while (*p) {
    while (*p == ',' || *p == ' ' || *p == '\n') p++;
    if (!*p) break;
    char *e; long a = strtol(p, &e, 10); if (e == p) break;
    long b = a; p = e;
    if (*p == '-') { p++; b = strtol(p, &e, 10); if (e == p) break; p = e; }
    for (long c = a; c <= b && c < 4096; c++) out.push_back((int)c);
  }
  return out;
}
// NUMA node (-1: unknown) and its CPUs of the PCI device "dddd:bb:dd.f" (lower case, as sysfs spells it)
inline int numa_cpus_of_pci(const char *bdf, const char *root, std::vector<int> &cpus) {
  cpus.clear();
  std::string r = root && *root ? root : "/", txt;
  if (r.back() != '/') r += '/';
  std::string b = bdf ? bdf : "";
  for (auto &ch : b) if (ch >= 'A' && ch <= 'F') ch = (char)(ch - 'A' + 'a');
  if (b.empty() || !read_small_file(r + "sys/bus/pci/devices/" + b + "/numa_node", txt)) return -1;
  const int node = atoi(txt.c_str());
  if (node < 0) return -1;
  if (read_small_file(r + "sys/devices/system/node/node" + std::to_string(node) + "/cpulist", txt)) cpus = parse_cpulist(txt);
  return node;
}
inline const char *sysfs_root() { const char *e = getenv("AVRF_SYSFS_ROOT"); return e && *e ? e : "/"; }
inline bool numa_binding_enabled() { const char *e = getenv("AVRF_POOL_NUMA"); return !(e && !strcmp(e, "0")); }
// the CPUs of `cpus` this thread may run on (its current affinity mask); empty when nothing is left or nothing would change
inline bool numa_mask(const std::vector<int> &cpus, cpu_set_t &set) {
  cpu_set_t cur; CPU_ZERO(&cur); CPU_ZERO(&set);
  if (sched_getaffinity(0, sizeof cur, &cur) != 0) return false;
  int kept = 0;
  for (int c : cpus) if (c >= 0 && c < CPU_SETSIZE && CPU_ISSET(c, &cur)) { CPU_SET(c, &set); kept++; }
  return kept > 0 && kept < CPU_COUNT(&cur);
}

}  // namespace avrf
