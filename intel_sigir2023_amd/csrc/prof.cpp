#include "prof.h"

#include <stdio.h>
#include <string.h>

#include <map>
#include <string>
#include <vector>

#include "../../include/intel_hip.h"

namespace {
struct Rec { std::string name; hipEvent_t a, b; double flops, bytes; bool closed; int stream; };
std::vector<hipStream_t> g_streams;      // first-seen order -> small stream index of the timeline
int stream_index(hipStream_t s) {
  for (size_t i = 0; i < g_streams.size(); ++i)
    if (g_streams[i] == s) return (int)i;
  g_streams.push_back(s);
  return (int)g_streams.size() - 1;
}
bool g_on = false;
std::vector<Rec> g_recs;
std::vector<hipEvent_t> g_pool;
std::string g_report;

hipEvent_t get_event() {
  if (!g_pool.empty()) {
    hipEvent_t e = g_pool.back();
    g_pool.pop_back();
    return e;
  }
  hipEvent_t e;
  (void)hipEventCreate(&e);
  return e;
}
}  // namespace

bool prof_enabled() { return g_on; }

ProfScope::ProfScope(const char* name, hipStream_t s, double flops, double bytes) : idx(-1), st(s) {
  if (!g_on) return;
  Rec r;
  r.name = name; r.a = get_event(); r.b = get_event(); r.flops = flops; r.bytes = bytes; r.closed = false; r.stream = stream_index(s);
  (void)hipEventRecord(r.a, s);
  g_recs.push_back(r);
  idx = (int)g_recs.size() - 1;
}
ProfScope::ProfScope(const char* name, hipStream_t s, double flops, double bytes, int m, int n, int k) : idx(-1), st(s) {
  if (!g_on) return;
  char buf[160];
  snprintf(buf, sizeof(buf), "%s[%dx%dx%d]", name, m, n, k);
  Rec r;
  r.name = buf; r.a = get_event(); r.b = get_event(); r.flops = flops; r.bytes = bytes; r.closed = false; r.stream = stream_index(s);
  (void)hipEventRecord(r.a, s);
  g_recs.push_back(r);
  idx = (int)g_recs.size() - 1;
}
ProfScope::~ProfScope() {
  if (idx < 0) return;
  (void)hipEventRecord(g_recs[idx].b, st);
  g_recs[idx].closed = true;
}

extern "C" void intel_prof_enable(int on) { g_on = on != 0; }

// Synchronises the device, aggregates and clears the records.  Returns a JSON object string:
// {"kernel": {"launches": n, "ms": total, "flops": f, "bytes": b}, ...}
extern "C" const char* intel_prof_collect(void) {
  (void)hipDeviceSynchronize();
  struct Agg { long n; double ms, flops, bytes; };
  std::map<std::string, Agg> agg;
  for (Rec& r : g_recs) {
    float ms = 0.f;
    if (r.closed && hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
      Agg& a = agg[r.name];
      a.n += 1; a.ms += ms; a.flops += r.flops; a.bytes += r.bytes;
    }
    g_pool.push_back(r.a);
    g_pool.push_back(r.b);
  }
  g_recs.clear();
  g_report = "{";
  bool first = true;
  char buf[512];
  for (auto& kv : agg) {
    snprintf(buf, sizeof(buf), "%s\"%s\": {\"launches\": %ld, \"ms\": %.6f, \"flops\": %.6e, \"bytes\": %.6e}", first ? "" : ", ",
             kv.first.c_str(), kv.second.n, kv.second.ms, kv.second.flops, kv.second.bytes);
    g_report += buf;
    first = false;
  }
  g_report += "}";
  return g_report.c_str();
}

// The same records as a timeline: [{"name": ..., "stream": k, "t0": ms, "t1": ms}, ...] relative to the first record, in launch
// order.  With the branches left on their own streams (intel_set_concurrency(ctx, 1), the default) this is the step as it
// really overlaps -- two event packets per kernel instead of a tracing profiler's per-launch host cost.  Synchronises, clears.
extern "C" const char* intel_prof_timeline(void) {
  (void)hipDeviceSynchronize();
  g_report = "[";
  char buf[512];
  bool first = true;
  for (Rec& r : g_recs) {
    float t0 = 0.f, t1 = 0.f;
    if (r.closed && hipEventElapsedTime(&t0, g_recs[0].a, r.a) == hipSuccess && hipEventElapsedTime(&t1, g_recs[0].a, r.b) == hipSuccess) {
      snprintf(buf, sizeof(buf), "%s{\"name\": \"%s\", \"stream\": %d, \"t0\": %.5f, \"t1\": %.5f}", first ? "" : ", ", r.name.c_str(), r.stream, t0, t1);
      g_report += buf;
      first = false;
    }
  }
  for (Rec& r : g_recs) {
    g_pool.push_back(r.a);
    g_pool.push_back(r.b);
  }
  g_recs.clear();
  g_report += "]";
  return g_report.c_str();
}
